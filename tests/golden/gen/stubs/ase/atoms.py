class Atoms:
    def __init__(self, *a, **kw):
        pass
