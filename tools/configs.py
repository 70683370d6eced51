#!/usr/bin/env python3
"""Runs BASELINE.json's configurations 2-5 on one GPU (the HIP path through the C ABI) and prints one JSON line each:
kernel time, rounds/s, events/s, faults, queue high-water mark, HBM footprint.  Configurations 4 and 5 use this
framework's extensions (equivocating leaders; weighted voting rights with reference quirk semantics Q1/Q2), for which
the reference has no answer: oracle/lbft_oracle.cpp is their specification."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c2_1024x4_lognormal": dict(instances=1024, nodes=4, max_clock=1000),
    "c2_1024x4_uniform": dict(instances=1024, nodes=4, max_clock=1000, uniform=(5, 15)),
    "c3_65536x4": dict(instances=65536, nodes=4, max_clock=1000),
    "c4_16384x64_longtail_equivocators": dict(instances=16384, nodes=64, max_clock=300, variance=400.0, equivocate_every=5),
    "c5_8192x100_weighted_epochs": dict(instances=8192, nodes=100, max_clock=300, weights=[1 + (i % 4) for i in range(100)],
                                        commands_per_epoch=50),
    # config 5 at max_clock 300 never reaches its first epoch change (~8 commits); this variant crosses epochs: the "fixed"
    # protocol mode (quirks Q1 and Q2 fixed -- the reference semantics stall at the first change), an epoch every 10
    # commands, and the voting rights rotating by one node per epoch (rights_rotation, the epoch-reconfiguration extension)
    "c5b_1024x100_rotating_rights_epochs_fixed": dict(instances=1024, nodes=100, max_clock=600, weights=[1 + (i % 4) for i in range(100)],
                                                      commands_per_epoch=10, quirks=3, rights_rotation=1),
}


def run(name, scale=1.0, reps=1, lpw=0):
    import numpy as np
    from librabft_simulator_amd import BatchSimulator, NodeConfig, RandomDelay
    c = CONFIGS[name]
    m = max(int(c["instances"] * scale), 1)
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    delay = RandomDelay.uniform(*c["uniform"]) if "uniform" in c else RandomDelay.new(10.0, c.get("variance", 4.0))
    sim = BatchSimulator.new(seeds, c["nodes"], delay, NodeConfig(), commands_per_epoch=c.get("commands_per_epoch", 30000),
                             voting_rights=c.get("weights"), equivocate_every=c.get("equivocate_every", 0), lanes_per_wavefront=lpw, quirks=c.get("quirks", 0),
                             rights_rotation=c.get("rights_rotation", 0))
    best = None
    for _ in range(reps):
        sim.reset()
        res = sim.loop_until(c["max_clock"], allow_faults=True)
        ms = sim.last_run_ms()[1]
        best = ms if best is None else min(best, ms)
    k = res.counters
    out = {"config": name, "instances": m, "nodes": c["nodes"], "max_clock": c["max_clock"], "kernel_ms": best,
           "rounds_per_s": k["rounds"] / (best * 1e-3), "commits_per_s": k["commits"] / (best * 1e-3),
           "events_per_s": sum(k["events"]) / (best * 1e-3), "events": sum(k["events"]), "rounds": k["rounds"], "commits": k["commits"],
           "faulted_instances": k["faulted_instances"], "max_queue": k["max_queue"], "max_snapshots": k["max_snapshots"],
           "device_gb": sim.device_bytes() / 1e9, "layout": sim.layout()}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    # c5b is opt-in: about a minute of GPU time per run
    ap.add_argument("names", nargs="*", default=[n for n in CONFIGS if not n.startswith("c5b")])
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the configuration's instance count")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--lpw", type=int, default=0, help="lanes per wavefront carrying an instance (0 = auto)")
    a = ap.parse_args()
    for n in a.names:
        run(n, a.scale, a.reps, a.lpw)
