#!/usr/bin/env python3
"""Generates tests/golden/: (1) reference_goldens.json -- the known answers the REFERENCE's own tests hold for the hot
path, transcribed with their source locations (they are what pins the oracle); (2) oracle_fixtures.json -- outputs of
the pinned oracle on small seeded configurations, so that the HIP path can also be checked against committed vectors
(tests/test_golden_fixtures.py) independently of building the oracle.  Re-run after changing the oracle:
    python tests/golden/gen_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle_ctypes as oc  # noqa: E402

REFERENCE = {
    "source": "novifinancial/librabft_simulator",
    "simulated_run_3_nodes": {
        "cite": "librabft-v2/tests/simulated_run.rs:45-66", "seed": 52, "nodes": 3, "mean": 10.0, "variance": 4.0, "max_clock": 1000,
        "commits": [27, 27, 27], "last_committed_state": [11134312813757838303] * 3},
    "simulated_run_8_nodes": {
        "cite": "librabft-v2/tests/simulated_run.rs:68-94", "seed": 48, "nodes": 8, "mean": 10.0, "variance": 4.0, "max_clock": 1000,
        "commits": [28] * 7 + [30], "last_committed_state": [12785928431398617538] * 7 + [4890275890002623733]},
    "empty_ledger_state": {"cite": "README.md:27", "value": 13646096770106105413},
    "quorum_threshold": {"cite": "bft-lib/src/unit_tests/configuration_tests.rs:31-47", "n_to_threshold": {"1": 1, "2": 2, "3": 3, "4": 3, "5": 4, "6": 5}},
    "pick_author_hits": {"cite": "bft-lib/src/unit_tests/configuration_tests.rs:17-29", "weights": [1, 2, 5], "seeds": [20, 27], "sorted_hits": [1, 2, 5]},
}

FIXTURES = {
    "c1_3nodes_fixed10": (dict(num_nodes=3, mean=10.0, variance=0.0), 4, 2800),
    "c2_4nodes_lognormal": (dict(num_nodes=4), 16, 1000),
    "c2_4nodes_uniform": (dict(num_nodes=4, delay_model=1, uniform_lo=5, uniform_hi=15), 8, 1000),
    "n8": (dict(num_nodes=8), 4, 1000),
    "weighted_n5": (dict(num_nodes=5, voting_rights=[5, 1, 1, 2, 3]), 8, 1000),
    "epoch_change_cpe50": (dict(num_nodes=4, commands_per_epoch=50), 8, 3000),
    "q2fixed_cpe50": (dict(num_nodes=4, commands_per_epoch=50, quirks=2), 8, 3000),
    "long_tail_n4": (dict(num_nodes=4, mean=10.0, variance=400.0), 8, 2000),
    "n40": (dict(num_nodes=40), 1, 300),
    "equivocators_n7": (dict(num_nodes=7, equivocate_every=3), 8, 1000),
    "lossy_n4": (dict(num_nodes=4, drop_per_million=50000, partition_size=2, partition_start=300, partition_end=500), 8, 1500),
}


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "reference_goldens.json"), "w") as f:
        json.dump(REFERENCE, f, indent=1)
    fx = {}
    for name, (kw, m, max_clock) in FIXTURES.items():
        seeds = (np.arange(1, m + 1, dtype=np.uint64) * 7919 + 11).astype(np.uint64)
        r = oc.run_batch(oc.make_config(math_mode=0, **kw), seeds, max_clock, threads=4, history_cap=16)
        r1 = oc.run_batch(oc.make_config(math_mode=1, **kw), seeds, max_clock, threads=4, history_cap=16)
        assert (r["last_states"] == r1["last_states"]).all(), name  # host libm (as the reference) == lbft_math.h
        c = r["counters"]
        fx[name] = {"config": kw, "seeds": [int(s) for s in seeds], "max_clock": max_clock,
                    "commit_counts": r["commit_counts"].tolist(), "active_rounds": r["active_rounds"].tolist(),
                    "last_committed_state": [[int(v) for v in row] for row in r["last_states"]],
                    "first_commits": [[[[int(e["proposer"]), int(e["index"]), int(e["time"])] for e in r["histories"][i, n][:min(16, int(r["commit_counts"][i, n]))]]
                                       for n in range(kw["num_nodes"])] for i in range(m)],
                    "events": c["events"], "rng_draws": c["rng_draws"], "events_scheduled": c["events_scheduled"]}
    with open(os.path.join(out_dir, "oracle_fixtures.json"), "w") as f:
        json.dump(fx, f)
    print("wrote", out_dir, {k: len(v["seeds"]) for k, v in fx.items()})


if __name__ == "__main__":
    main()
