#!/usr/bin/env python3
"""GPU box: bit-exact comparison of variant builds of the library (LBFT_HIP_LIB) with a reference build on the whole bench batch
-- commit counts, active rounds, State hashes of every node, the aggregate counters -- plus a bounded-launch run (run_steps) of the
variant.  One subprocess per library (a library is loaded once per process); prints one line per variant.
    python tools/variant_parity.py liblbft_hip.so liblbft_hip_d1.so liblbft_hip_d2.so"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(out):
    sys.path.insert(0, ROOT)
    from librabft_simulator_amd import BatchSimulator, RandomDelay
    m = int(os.environ.get("INSTANCES", "65536"))
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    sim = BatchSimulator.new(seeds, 4, RandomDelay.new(10.0, 4.0))
    res = sim.loop_until(1000)
    c = res.counters
    # the same batch advanced in bounded launches (137 events per instance and launch) must end in the same state
    sim2 = BatchSimulator.new(seeds[:8192], 4, RandomDelay.new(10.0, 4.0), lanes_per_wavefront=int(os.environ.get("LPW2", "32")))  # (LPW2=0: the batch's own kernel)
    r2 = None
    for _ in range(100):
        left, r2 = sim2.run_steps(1000, 137)
        if left == 0:
            break
    np.savez(out, cc=res.commit_counts, ar=res.active_rounds, st=res.last_committed_states,
             cc2=r2.commit_counts, st2=r2.last_committed_states,
             ctr=np.array([sum(c["events"]), c["rounds"], c["commits"], c["rng_draws"], c["events_scheduled"], c["timers_folded"], c["node_updates"]], dtype=np.uint64))
    print(json.dumps({"kernel_class": sim.layout()["kernel_class"], "class2": sim2.layout()["kernel_class"], "ms": sim.last_run_ms()[1]}))


def main():
    if sys.argv[1] == "--one":
        return one(sys.argv[2])
    libs = sys.argv[1:]
    ref = None
    for k, name in enumerate(libs):
        path = os.path.join(ROOT, "librabft_simulator_amd", name)
        out = "/tmp/vp_%d.npz" % k
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", out], env=dict(os.environ, LBFT_HIP_LIB=path), capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(name, "FAILED", r.stderr[-600:])
            continue
        d = dict(np.load(out))
        if ref is None:
            ref = d
        same = {k2: bool((d[k2] == ref[k2]).all()) for k2 in ("cc", "ar", "st", "ctr")}
        same["steps_cc"] = bool((d["cc2"] == ref["cc"][:len(d["cc2"])]).all())
        same["steps_st"] = bool((d["st2"] == ref["st"][:len(d["st2"])]).all())
        print(name, r.stdout.strip(), "OK" if all(same.values()) else "MISMATCH", same, d["ctr"].tolist(), flush=True)


if __name__ == "__main__":
    main()
