#!/usr/bin/env python3
"""Offline randomised differential sweep of the LARGE-network cooperative event loop (CPU only; test infrastructure): the kernel logic compiled for the host
(oracle/host_model.cpp: run_coop with 64 emulated lanes -- coop_bulk, the request / response runs) against the full-fidelity oracle on networks of 33..128
nodes drawn like tests/test_fuzz_model.py::draw_large_config, two seeds each, with short horizons mixed in (timers past the horizon).
    python tests/tools/fuzz_large_coop.py 0 150 [--reference-semantics]     # chunks [0, 150); ~3 minutes per 100 on two cores
--reference-semantics forces quirks & 2 (requests answered by the requester: the mode of the response runs).  Prints every configuration that differs or
faults (faults = 2 is the harness's snapshot capacity with quirks bit 0 and a short query period, not a difference)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_ctypes as oracle  # noqa: E402
from test_fuzz_model import draw_large_config  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
ref_sem = "--reference-semantics" in sys.argv
bad = tot = 0
for chunk in range(lo, hi):
    rng = np.random.default_rng((4242000 if ref_sem else 5252000) + chunk)
    kw, n, max_clock = draw_large_config(rng)
    if ref_sem:
        kw["quirks"] = kw.get("quirks", 0) & 2
    kw.pop("drop_per_million", None)            # (loss keeps the lane-per-network loop: coop() excludes it)
    if rng.random() < 0.5:
        max_clock = int(rng.choice([40, 77, 150, 300, 420]))
        if n > 66 and max_clock > 200:
            max_clock = 200
    seeds = rng.integers(1, 2 ** 62, 2, dtype=np.uint64)
    cfg = oracle.make_config(math_mode=1, **kw)
    a = oracle.run_batch(cfg, seeds, max_clock, threads=2, history_cap=64)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=2, history_cap=64, qcap=max(8192, 32 * n * n),
                                   scap=min(65535, 6 * n * n + 16 * n) if kw.get("quirks", 0) & 1 else 128 * n, bcap=512, lcap=512, ql=0, qheap=1, qcal=1,
                                   ring=int(rng.choice([128, 256, 512])), ring_topup=int(rng.choice([0, 4, 16])))
    ok = (all((a[k] == b[k]).all() for k in ("commit_counts", "active_rounds", "last_states", "histories")) and not b["faults"].any()
          and all(a["counters"][k] == b["counters"][k] for k in ("events", "rng_draws", "rounds", "commits", "events_scheduled")))
    tot += 1
    if not ok:
        bad += 1
        print("MISMATCH" if not b["faults"].any() else "FAULT", chunk, kw, max_clock, b["faults"].tolist(), flush=True)
print("large configurations on the cooperative loop:", tot, "differing or faulted:", bad)
