"""Pin the CPU oracle against every golden / known-answer the reference holds for the hot path.

Sources (paths in novifinancial/librabft_simulator):
  * librabft-v2/tests/simulated_run.rs:45-94   -- the two golden integration tests
  * bft-lib/src/unit_tests/configuration_tests.rs:7-47 -- quorum thresholds, pick_author hit counts
  * README.md:27 -- State of the empty ledger
  * SURVEY.md Appendix B -- secondary checkpoints derived from the same goldens
"""
import numpy as np
import pytest

GOLDEN_3 = dict(seed=52, nodes=3, commits=[27, 27, 27], states=[11134312813757838303] * 3)
GOLDEN_8 = dict(seed=48, nodes=8, commits=[28] * 7 + [30],
                states=[12785928431398617538] * 7 + [4890275890002623733])


@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("g", [GOLDEN_3, GOLDEN_8], ids=["3nodes_seed52", "8nodes_seed48"])
def test_simulated_run_goldens(oracle, g, math_mode):
    # make_simulator(): LogNormal(10, 4), delta 20, gamma 2, lambda 0.5, tci 100000, cpe 30000, T=1000
    sim = oracle.OracleSim(oracle.make_config(num_nodes=g["nodes"], math_mode=math_mode), g["seed"]).run_until(1000)
    assert sim.commit_counts() == g["commits"]
    assert sim.last_committed_states() == g["states"]
    assert sim.counters()["response_inserts"] == 0  # quirk Q1: responses never insert anything


def test_appendix_b_checkpoints_seed52(oracle):
    sim = oracle.OracleSim(oracle.make_config(num_nodes=3), 52).run_until(1000)
    assert sim.startup_times() == [10, 10, 8]
    assert sim.active_rounds() == [37, 37, 37]
    c = sim.counters()
    assert c["events"] == [356, 88, 88, 496] and c["rng_draws"] == 882
    h = sim.committed_history(0)
    first = [(int(e["proposer"]), int(e["index"]), int(e["time"])) for e in h[:6]]
    assert first == [(2, 0, 32), (2, 1, 52), (2, 2, 78), (2, 3, 136), (1, 2, 160), (1, 3, 183)]
    assert (int(h[-1]["proposer"]), int(h[-1]["index"]), int(h[-1]["time"])) == (2, 12, 893)
    assert [oracle.leader(3, r) for r in range(1, 13)] == [1, 2, 2, 2, 1, 2, 1, 1, 2, 1, 2, 0]


def test_appendix_b_checkpoints_seed48(oracle):
    sim = oracle.OracleSim(oracle.make_config(num_nodes=8), 48).run_until(1000)
    assert sim.startup_times() == [12, 10, 11, 14, 7, 8, 7, 13]
    assert sim.active_rounds() == [36] * 7 + [37]
    c = sim.counters()
    assert c["events"] == [2169, 1295, 1295, 3398] and c["rng_draws"] == 8870
    h = sim.committed_history(0)
    first = [(int(e["proposer"]), int(e["index"]), int(e["time"])) for e in h[:6]]
    assert first == [(6, 0, 35), (3, 0, 61), (5, 0, 95), (6, 1, 124), (6, 2, 143), (3, 1, 162)]
    assert (int(h[-1]["proposer"]), int(h[-1]["index"]), int(h[-1]["time"])) == (1, 2, 830)
    assert [oracle.leader(8, r) for r in range(1, 13)] == [7, 6, 3, 5, 6, 6, 3, 5, 3, 5, 3, 7]


def test_fixed_delay_config1(oracle):
    # BASELINE.json configs[0]: 3 nodes, --mean 10 --variance 0 (delay == 10)
    sim = oracle.OracleSim(oracle.make_config(num_nodes=3, mean=10.0, variance=0.0), 1).run_until(1000)
    assert sim.startup_times() == [11, 11, 11]
    assert sim.commit_counts() == [27, 26, 26]
    assert min(sim.active_rounds()) == 36
    h = sim.committed_history(0)
    first = [(int(e["proposer"]), int(e["index"]), int(e["time"])) for e in h[:5]]
    assert first == [(2, 0, 30), (2, 1, 51), (2, 2, 72), (2, 3, 132), (1, 2, 162)]


def test_epoch_change_stall_q1_q2(oracle):
    # n=4, seed 3, T=3000, commands_per_epoch=50
    ref = oracle.OracleSim(oracle.make_config(num_nodes=4, commands_per_epoch=50), 3).run_until(3000)
    assert ref.commit_counts() == [50, 50, 49, 49] and ref.epochs() == [1, 1, 0, 0]
    fixed = oracle.OracleSim(oracle.make_config(num_nodes=4, commands_per_epoch=50, quirks=2), 3).run_until(3000)
    assert fixed.commit_counts() == [100] * 4 and fixed.epochs() == [2] * 4


def test_siphash_and_xoshiro_kats(oracle):
    L = oracle.lib()
    assert L.lbft_oracle_siphash13((0).to_bytes(8, "little"), 8) == 13646096770106105413  # README.md:27
    assert L.lbft_oracle_siphash13((1).to_bytes(8, "little"), 8) == 2206609067086327257
    x = np.zeros(3, dtype=np.uint64)
    L.lbft_oracle_xoshiro_first(52, x.ctypes.data, 3)
    assert [int(v) for v in x] == [0x81E93AE64DFA3E63, 0x79F50E0B404F5D45, 0x0BEAE4ADBCB274AE]


def test_configuration_unit_tests(oracle):
    L = oracle.lib()
    # configuration_tests.rs:40-47
    assert [L.lbft_oracle_quorum_threshold(None, n) for n in range(1, 7)] == [1, 2, 3, 3, 4, 5]
    # configuration_tests.rs:17-29 : hit counts sorted == [1, 2, 5]
    w = np.array([1, 2, 5], dtype=np.uint64)
    picks = [L.lbft_oracle_pick_author(w.ctypes.data, 3, s) for s in range(20, 28)]
    assert picks == [1, 0, 1, 2, 1, 2, 1, 1]
    assert sorted(picks.count(a) for a in set(picks)) == [1, 2, 5]


def test_reference_overheads_change_nothing_but_the_cost(oracle):
    """lbft_oracle_config.reference_overheads: the oracle also serialises the whole NodeState after every processed event
    (save_node, simulator.rs:307-309) and clones the notification per receiver (:348-354), as the reference does -- bench.py's
    cpu_baseline uses it; results and RNG consumption are untouched."""
    import numpy as np
    seeds = np.arange(40, 56, dtype=np.uint64)
    for kw in (dict(num_nodes=3), dict(num_nodes=8, mean=10.0, variance=400.0), dict(num_nodes=5, quirks=3, commands_per_epoch=9)):
        a = oracle.run_batch(oracle.make_config(**kw), seeds, 1000, threads=4, history_cap=64)
        b = oracle.run_batch(oracle.make_config(reference_overheads=1, **kw), seeds, 1000, threads=4, history_cap=64)
        for key in ("commit_counts", "active_rounds", "last_states", "histories"):
            assert (a[key] == b[key]).all(), key
        for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
            assert a["counters"][key] == b["counters"][key], key
        assert a["counters"]["saved_bytes"] == 0 and b["counters"]["saved_bytes"] > 1000 * len(seeds)
