#!/usr/bin/env python3
"""GPU box (host CPUs only): how the oracle's batch runner scales over the box's host threads -- one process with T threads against P processes --
on a slice of a large-network configuration.  Decides where tests/golden/gen_full_size.py is worth running (round 6)."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def work(args):
    name, lo, hi, threads = args
    import full_size_digest as fsd
    import oracle_ctypes as oc
    from configs import CONFIGS
    c = CONFIGS[name]
    cfg = oc.make_config(math_mode=1, **fsd.oracle_kwargs(c))
    oc.run_batch(cfg, np.arange(lo + 1, hi + 1, dtype=np.uint64), c["max_clock"], threads=threads, history_cap=0)
    return hi - lo


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "c4live_16384x64_longtail_equivocators_fixed"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    print("host threads", os.cpu_count(), flush=True)
    for threads in (256, 128, 64, 32):
        t0 = time.time()
        work((name, 0, k, threads))
        print("%s: 1 process x %d threads, %d instances: %.1f s" % (name, threads, k, time.time() - t0), flush=True)
    for procs, threads in ((8, 32), (32, 8), (128, 2)):
        per = k // procs
        t0 = time.time()
        with mp.get_context("spawn").Pool(procs) as pool:
            pool.map(work, [(name, i * per, (i + 1) * per, threads) for i in range(procs)])
        print("%s: %d processes x %d threads, %d instances: %.1f s" % (name, procs, threads, k, time.time() - t0), flush=True)
