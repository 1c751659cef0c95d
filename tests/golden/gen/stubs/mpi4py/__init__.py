"""Minimal single-process stand-in for mpi4py, used ONLY by tests/golden/gen/make_golden.py
to import the reference package in the build container (mpi4py is not installed there).
Our own code; nothing here ships to the product path."""


class _Comm:
    def Get_size(self):
        return 1

    def Get_rank(self):
        return 0

    def Bcast(self, a, root=0):
        pass

    def Allreduce(self, a, b, op=None):
        b[...] = a

    def Barrier(self):
        pass


class MPI:
    COMM_WORLD = _Comm()
    MAX = "max"
    SUM = "sum"
    IN_PLACE = "in_place"
