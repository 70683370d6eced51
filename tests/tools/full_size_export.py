#!/usr/bin/env python3
"""GPU box: run a full-size large-network configuration of tools/configs.py on the HIP path and save what the parity checks compare -- commit counts, active
rounds, epochs, per-node State hashes (a SipHash of the node's whole committed history) of EVERY instance and the aggregate counters -- to a compressed npz
under gpurun_out/, so that the bit-exact comparison with the CPU oracle over all instances can run elsewhere (tests/tools/full_size_check.py) without
spending GPU minutes on host work.
    python tests/tools/full_size_export.py c5_8192x100_weighted_epochs c5live_8192x100_rotating_rights_epochs_fixed [--out gpurun_out/full_size]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+")
    ap.add_argument("--out", default="gpurun_out/full_size")
    a = ap.parse_args()
    from configs import CONFIGS, kernel_name
    from librabft_simulator_amd import BatchSimulator, NodeConfig, RandomDelay
    os.makedirs(a.out, exist_ok=True)
    for name in a.names:
        c = CONFIGS[name]
        seeds = np.arange(1, c["instances"] + 1, dtype=np.uint64)
        delay = RandomDelay.uniform(*c["uniform"]) if "uniform" in c else RandomDelay.new(10.0, c.get("variance", 4.0))
        sim = BatchSimulator.new(seeds, c["nodes"], delay, NodeConfig(), commands_per_epoch=c.get("commands_per_epoch", 30000), voting_rights=c.get("weights"),
                                 equivocate_every=c.get("equivocate_every", 0), quirks=c.get("quirks", 0), rights_rotation=c.get("rights_rotation", 0))
        res = sim.loop_until(c["max_clock"])
        assert not res.faults.any()
        np.savez_compressed(os.path.join(a.out, name + ".npz"), commit_counts=res.commit_counts, active_rounds=res.active_rounds, epochs=res.epochs,
                            last_states=res.last_committed_states, counters=json.dumps(res.counters), kernel=kernel_name(sim.layout()),
                            kernel_ms=sim.last_run_ms()[1])
        print(name, kernel_name(sim.layout()), "%.1f ms" % sim.last_run_ms()[1], "events", sum(res.counters["events"]), flush=True)


if __name__ == "__main__":
    main()
