#!/usr/bin/env python3
"""CPU (test infrastructure): the device results of a FULL-SIZE large-network configuration (tests/tools/full_size_export.py, run on the GPU box) against
the CPU oracle on EVERY instance -- commit counts, active rounds and the State hash of every node (a SipHash of its whole committed history), bit for bit --
in chunks, on the cores at hand (hours of CPU time that the GPU box's budget does not have to pay for).  Appends one line per chunk and a verdict to the report.
    python tests/tools/full_size_check.py gpurun_out/full_size/c5_8192x100_weighted_epochs.npz --threads 6 --report profiles/r05/full_size_c5_all_8192.txt"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("npz")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--count", type=int, default=0, help="instances to check (0 = all from --first)")
    ap.add_argument("--report", default=None)
    a = ap.parse_args()
    import oracle_ctypes as oc
    from configs import CONFIGS
    name = os.path.basename(a.npz)[:-4]
    c = CONFIGS[name]
    d = np.load(a.npz)
    m = c["instances"]
    assert d["commit_counts"].shape[0] == m
    kw = dict(num_nodes=c["nodes"], mean=10.0, variance=c.get("variance", 4.0), commands_per_epoch=c.get("commands_per_epoch", 30000), quirks=c.get("quirks", 0),
              rights_rotation=c.get("rights_rotation", 0), equivocate_every=c.get("equivocate_every", 0))
    if c.get("weights"):
        kw["voting_rights"] = c["weights"]
    cfg = oc.make_config(math_mode=1, **kw)
    last = m if a.count == 0 else min(m, a.first + a.count)
    out = open(a.report, "a") if a.report else sys.stdout
    out.write("%s: device (%s, %.1f ms) against oracle/lbft_oracle.cpp (math_mode 1), instances [%d, %d) of %d, seeds = index + 1, %d threads\n" % (
        name, str(d["kernel"]), float(d["kernel_ms"]), a.first, last, m, a.threads))
    out.flush()
    bad, t0 = 0, time.time()
    for lo in range(a.first, last, a.chunk):
        hi = min(last, lo + a.chunk)
        seeds = np.arange(lo + 1, hi + 1, dtype=np.uint64)
        ref = oc.run_batch(cfg, seeds, c["max_clock"], threads=a.threads)
        same = {k: bool((d[k][lo:hi] == ref[r]).all()) for k, r in (("commit_counts", "commit_counts"), ("active_rounds", "active_rounds"), ("last_states", "last_states"))}
        bad += 0 if all(same.values()) else 1
        out.write("  [%5d, %5d) %s  commits %d  %.0f s\n" % (lo, hi, "equal" if all(same.values()) else "MISMATCH " + json.dumps(same), int(ref["commit_counts"].sum()), time.time() - t0))
        out.flush()
    out.write("%s: %s -- %d instances x %d nodes compared in %.0f s\n" % (name, "ALL EQUAL" if bad == 0 else "%d chunk(s) with mismatches" % bad, last - a.first, c["nodes"], time.time() - t0))
    out.flush()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
