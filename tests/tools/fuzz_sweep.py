#!/usr/bin/env python3
"""Offline randomised differential sweep (CPU only; test infrastructure): the kernel logic compiled for the host against the
full-fidelity oracle on `chunks x 16` configurations drawn like tests/test_fuzz_model.py does, with other seeds.
    python tests/tools/fuzz_sweep.py 0 400      # chunks [0, 400): 6 400 configurations, ~4 minutes on 8 cores
Prints every configuration whose commit logs / States / rounds / counters differ or that raised a fault word (capacity
overflows of this harness show up as faults = 2: snapshot slots with quirks bit 0 and a very short query period)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_ctypes as oracle  # noqa: E402
from test_fuzz_model import draw_config, draw_large_config  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
base = int(sys.argv[3]) if len(sys.argv) > 3 else 77000
bad = total = 0
for chunk in range(lo, hi):
    rng = np.random.default_rng(base + chunk)
    for it in range(16):
        kw = draw_config(rng)
        n = kw["num_nodes"]
        max_clock = int(rng.choice([300, 600, 1000])) if n <= 16 else 250
        m = 6 if n <= 16 else 2
        seeds = rng.integers(1, 2 ** 62, m, dtype=np.uint64)
        cfg = oracle.make_config(math_mode=1, **kw)
        a = oracle.run_batch(cfg, seeds, max_clock, threads=os.cpu_count(), history_cap=96)
        special = any(k in kw for k in ("equivocate_every", "drop_per_million", "partition_size")) or (kw.get("quirks", 0) & 1)
        qheap = 1 if (n > 4 or rng.random() < 0.3) else 0
        qcal = 1 if (rng.random() < 0.5 and (qheap or special or n > 16)) else 0
        scap = 64 if (n <= 4 and not special and rng.random() < 0.5) else max(128, 128 * n)
        if (kw.get("quirks", 0) & 1) and scap != 64 and os.environ.get("LBFT_FUZZ_WIDE_SNAPSHOTS"):
            # the record exchange with a short query period (delta 5) keeps a query-all's responses of every node in flight: 7 of 4 800 configurations of chunks
            # 400..700 exhaust 128 n slots (fault word 2, no silent difference); with this switch the same draws get 32 n^2 + 128 n and must equal the oracle
            scap = min(65535, 32 * n * n + 128 * n)
        ql, fg = int(rng.choice([0, 3, 11, 48])), int(rng.random() < 0.2)
        b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=os.cpu_count(), history_cap=96, qcap=max(4096, 24 * n * n), scap=scap,
                                       bcap=1024, lcap=1024, ql=ql, qheap=qheap, qcal=qcal, force_generic=fg)
        ok = (all((a[k] == b[k]).all() for k in ("commit_counts", "active_rounds", "last_states", "histories")) and not b["faults"].any()
              and all(a["counters"][k] == b["counters"][k] for k in ("events", "rng_draws", "rounds", "commits", "events_scheduled")))
        total += 1
        if not ok:
            bad += 1
            print("MISMATCH chunk", chunk, "it", it, kw, "max_clock", max_clock, "scap", scap, "ql", ql, "force_generic", fg, "qheap", qheap,
                  "qcal", qcal, "faults", b["faults"].tolist(), flush=True)
# ... and, every fourth chunk, one large network (33..128 nodes) through the cooperative event loop (64 emulated lanes)
lbad = ltotal = 0
for chunk in range(lo, hi, 4):
    rng = np.random.default_rng(base + 500000 + chunk)
    kw, n, max_clock = draw_large_config(rng)
    seeds = rng.integers(1, 2 ** 62, 1, dtype=np.uint64)
    cfg = oracle.make_config(math_mode=1, **kw)
    a = oracle.run_batch(cfg, seeds, max_clock, threads=os.cpu_count(), history_cap=64)
    ring, topup = int(rng.choice([128, 256, 512])), int(rng.choice([0, 4, 16]))
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=os.cpu_count(), history_cap=64, qcap=max(8192, 32 * n * n),
                                   scap=min(65535, 6 * n * n + 16 * n) if kw.get("quirks", 0) & 1 else 128 * n, bcap=512, lcap=512, ql=0, qheap=1, qcal=1,
                                   ring=ring, ring_topup=topup)
    ok = (all((a[k] == b[k]).all() for k in ("commit_counts", "active_rounds", "last_states", "histories")) and not b["faults"].any()
          and all(a["counters"][k] == b["counters"][k] for k in ("events", "rng_draws", "rounds", "commits", "events_scheduled")))
    ltotal += 1
    if not ok:
        lbad += 1
        print("MISMATCH (large) chunk", chunk, kw, "max_clock", max_clock, "ring", ring, "topup", topup, "faults", b["faults"].tolist(), flush=True)
print("configurations", total, "mismatching or faulted", bad, "| large-network configurations (cooperative loop)", ltotal, "mismatching or faulted", lbad)
