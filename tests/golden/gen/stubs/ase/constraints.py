"""Import-only stand-in."""


class Filter:
    pass
