"""Import-only stand-in (ASE is not installed here): names theforce.calculator.active[_bcm] import."""
all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]


class Calculator:
    def __init__(self, *a, **k):
        self.results = {}
