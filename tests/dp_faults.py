"""Fault and delay injection for the data-parallel skeleton (framefusion_amd/dp.py), installed by tests only:
`bench.py --dp-hooks tests.dp_faults` / tests/dp_worker.py import this module, which replaces `dp.hooks`.  What happens is
chosen by the environment of the test that starts the job:
    FF_DP_FAIL_FIRST_ATTEMPT=1 [FF_DP_FAIL_RANK=r]   the first collective of the first attempt raises (on rank r / everywhere)
    FF_DP_SLOW_BARRIER_MS=n                          the last rank sleeps n ms inside every barrier"""
import os
import time


def install(dp):
    class Faults(dp.Hooks):
        def first_collective(self, rank, attempt):
            fail_rank = os.environ.get("FF_DP_FAIL_RANK")
            if (os.environ.get("FF_DP_FAIL_FIRST_ATTEMPT") == "1" and attempt == 0 and
                    (fail_rank is None or int(fail_rank) == rank)):
                raise RuntimeError("injected failure of the first collective (FF_DP_FAIL_FIRST_ATTEMPT=1)")

        def in_barrier(self, rank, world):
            slow = os.environ.get("FF_DP_SLOW_BARRIER_MS")
            if slow and rank == world - 1:
                time.sleep(float(slow) * 1e-3)

    dp.hooks = Faults()
