class SinglePointCalculator:
    def __init__(self, *a, **kw):
        pass
