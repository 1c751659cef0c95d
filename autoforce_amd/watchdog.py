"""A dead-man's switch around calls that can block for ever when a peer rank has died: RCCL's communicator set-up
and a rank's first collective have no time-out of their own (a rank that failed before `ncclCommInitRank`, or never
reaches the step's all-reduce, leaves the others waiting inside the library).  The reference has the same exposure
under mpirun (calculator/active.py:562,601-602,770-777 block in MPI); there the launcher kills the job.

    with Watchdog("comm_init", seconds=120, rank=rank):
        engine.comm_init(uid, rank, world)

On expiry the watchdog thread reports which rank is stuck where and ends THIS process with exit code 3 (os._exit: a
blocked HIP / RCCL call cannot be interrupted from Python).  Never re-execs: a process that has touched the GPU must
not be replaced by another program; the launcher (torchrun, the driver) sees the non-zero exit and tears the job down.
"""
import os
import sys
import threading


class Watchdog:
    def __init__(self, what, seconds=None, rank=0, stream=None):
        # SGPR_WATCHDOG_S is a FLOOR (wait at least that long), not an override: a site that asks for more — the timed
        # region of a long benchmark — keeps its own limit; 0 switches every watchdog off
        env = os.environ.get("SGPR_WATCHDOG_S")
        self.what, self.rank = what, rank
        base = 120.0 if seconds is None else float(seconds)
        self.seconds = base if not env else (0.0 if float(env) <= 0 else max(float(env), base))
        self.stream = stream or sys.stderr
        self._timer = None

    def _fire(self):
        try:
            print(f"[sgpr watchdog] rank {self.rank}: still inside `{self.what}` after {self.seconds:.0f} s — a peer rank has "
                  f"probably failed or never arrived; exiting with code 3 (set SGPR_WATCHDOG_S to wait longer, "
                  f"NCCL_DEBUG=WARN for RCCL's own diagnostics)", file=self.stream, flush=True)
        finally:
            os._exit(3)

    def __enter__(self):
        if self.seconds > 0:
            self._timer = threading.Timer(self.seconds, self._fire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False
