#!/usr/bin/env python3
"""One-off validation on the GPU box: the whole headline batch (65 536 x 4 nodes, max_clock 1000) on the HIP path against the
CPU oracle on all host cores -- commit counts, active rounds, State hashes of every node of every instance and the aggregate
counters.  (tests/ checks strided subsets; this checks everything once.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle_ctypes as oc  # noqa: E402
from librabft_simulator_amd import BatchSimulator, RandomDelay  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
seeds = np.arange(1, m + 1, dtype=np.uint64)
t = time.time()
res = BatchSimulator.new(seeds, 4, RandomDelay.new(10.0, 4.0)).loop_until(1000)
cc, ar, st = res.commit_counts, res.active_rounds, res.last_committed_states
print("gpu %.2f s" % (time.time() - t), flush=True)
t = time.time()
ref = oc.run_batch(oc.make_config(num_nodes=4, math_mode=1), seeds, 1000, threads=os.cpu_count(), history_cap=0)
print("oracle %.1f s on %d threads" % (time.time() - t, os.cpu_count()), flush=True)
ok = (cc == ref["commit_counts"]).all() and (ar == ref["active_rounds"]).all() and (st == ref["last_states"]).all()
c, rc = res.counters, ref["counters"]
ok = ok and all(c[k] == rc[k] for k in ("events", "rng_draws", "rounds", "commits", "events_scheduled"))
print("instances", m, "mismatching instances", int(((cc != ref["commit_counts"]) | (st != ref["last_states"])).any(axis=1).sum()), "counters equal",
      all(c[k] == rc[k] for k in ("events", "rng_draws", "rounds", "commits", "events_scheduled")))
print("FULL BATCH PARITY", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
