"""CPU-only differential test: the compact structural model that the HIP kernels execute
(librabft_simulator_amd/csrc/lbft_core.h, compiled for the host by oracle/host_model.cpp) must equal the
full-fidelity oracle -- commit logs, State values, active rounds, event counts by kind, RNG draws and
creation stamps -- over many seeds and configurations."""
import numpy as np
import pytest

CASES = {
    # BASELINE.json configs[0]: 3 nodes, fixed delay 10 (mean 10, variance 0), ~100 rounds
    "c1_3nodes_fixed10": (dict(num_nodes=3, mean=10.0, variance=0.0), 32, 2800),
    # configs[1] shape: 4 nodes f=1, LogNormal(10,4) (primary parity run)
    "c2_4nodes_lognormal": (dict(num_nodes=4), 256, 1000),
    "c2_4nodes_uniform": (dict(num_nodes=4, delay_model=1, uniform_lo=5, uniform_hi=15), 128, 1000),
    "golden_shapes_3": (dict(num_nodes=3), 64, 1000),
    "golden_shapes_8": (dict(num_nodes=8), 32, 1000),
    "n1": (dict(num_nodes=1), 4, 300),
    "n2": (dict(num_nodes=2), 16, 1000),
    "n16": (dict(num_nodes=16), 4, 400),
    "epoch_change_cpe50": (dict(num_nodes=4, commands_per_epoch=50), 64, 3000),
    "epoch_change_cpe5": (dict(num_nodes=3, commands_per_epoch=5), 64, 1500),
    "weighted": (dict(num_nodes=5, voting_rights=[5, 1, 1, 2, 3]), 64, 1000),
    "long_tail": (dict(num_nodes=4, mean=10.0, variance=400.0), 128, 2000),
    "params": (dict(num_nodes=5, gamma=1.5, lambda_=0.25, delta=5, target_commit_interval=80), 64, 1500),
    "long_run": (dict(num_nodes=4), 16, 8000),
}


# ql = event-queue slots held in the (emulated) LDS front: 0 = HBM rows only, 5 = almost everything
# spills across the LDS/HBM boundary, 48 = the device default (the front covers the high-water mark at n <= 4)
# scap = notification snapshot slots: <= 64 uses the register-resident free mask, larger the HBM free stack
@pytest.mark.parametrize("ql,scap", [(0, 512), (5, 64), (16, 64), (48, 64), (48, 512)])  # scap <= 256: packed one-word queue entries (class 0)
@pytest.mark.parametrize("name", sorted(CASES))
def test_host_model_equals_oracle(oracle, name, ql, scap):
    kw, m, max_clock = CASES[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    seeds = np.arange(1000, 1000 + m, dtype=np.uint64) * 7919
    a = oracle.run_batch(cfg, seeds, max_clock, threads=8, history_cap=512)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=8, history_cap=512, qcap=4096, scap=scap, bcap=1024,
                                   lcap=1024, ql=ql)
    assert not b["faults"].any()
    for key in ("commit_counts", "active_rounds", "last_states", "histories"):
        assert (a[key] == b[key]).all(), key
    ca, cb = a["counters"], b["counters"]
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert ca[key] == cb[key], key
    assert ca["response_inserts"] == 0  # Q1: the premise of payload-free requests/responses


# Networks above 32 nodes: multi-word author / node sets (extension rows), receiver lists in HBM rows and the
# binary-heap event queue (BASELINE.json configs 4 and 5 are 64 and 100 nodes).
LARGE = {
    "n20_list_rows": (dict(num_nodes=20), 4, 300, 0),
    "n33_two_mask_words": (dict(num_nodes=33), 2, 300, 1),
    "n40_heap": (dict(num_nodes=40), 2, 250, 1),
    "n64_long_tail": (dict(num_nodes=64, mean=10.0, variance=400.0), 1, 300, 1),
    "n100_weighted": (dict(num_nodes=100, voting_rights=[1 + (i % 4) for i in range(100)]), 1, 200, 1),
    "n4_heap_mode": (dict(num_nodes=4), 64, 1000, 1),
    "n8_heap_mode_timeouts": (dict(num_nodes=8, mean=10.0, variance=400.0), 16, 1500, 1),
    "n36_timeouts": (dict(num_nodes=36, mean=10.0, variance=900.0, delta=5), 2, 400, 1),
}


# Cooperative event loop of the large-network kernels (SimT::run_coop / coop_bulk): the host build emulates the 64 lanes of a
# wavefront (PL<T> arrays, LBFT_FOR_LANES), so the lane mapping, ballots and shuffles of the device code run here.  ring = ring of
# pre-generated RNG draws (entries), topup = draws the generator runs ahead per step.
COOP = {
    "n33_two_mask_words": (dict(num_nodes=33), 2, 300),
    "n33_uniform_delays": (dict(num_nodes=33, delay_model=1, uniform_lo=5, uniform_hi=15), 2, 300),
    "n40": (dict(num_nodes=40), 2, 250),
    "n64_long_tail": (dict(num_nodes=64, mean=10.0, variance=400.0), 2, 300),
    "n64_long_tail_equivocators": (dict(num_nodes=64, mean=10.0, variance=400.0, equivocate_every=5), 2, 300),
    "n65_first_two_pass_list": (dict(num_nodes=65, mean=10.0, variance=100.0), 1, 250),
    "n66": (dict(num_nodes=66), 1, 200),
    "n100_weighted": (dict(num_nodes=100, voting_rights=[1 + (i % 4) for i in range(100)]), 1, 200),
    "n128_max": (dict(num_nodes=128), 1, 120),
    "n36_timeouts": (dict(num_nodes=36, mean=10.0, variance=900.0, delta=5), 2, 400),
    "n64_q3_equivocators_live": (dict(num_nodes=64, mean=10.0, variance=400.0, equivocate_every=5, quirks=3), 2, 500),
    "n100_q3_rotating_rights_epochs": (dict(num_nodes=100, voting_rights=[1 + (i % 4) for i in range(100)], quirks=3, rights_rotation=1,
                                            commands_per_epoch=3), 1, 300),
    "n36_q2_cpe3": (dict(num_nodes=36, commands_per_epoch=3, quirks=2), 2, 300),
}


@pytest.mark.parametrize("ring,topup", [(512, 0), (128, 0), (512, 4), (256, 16)])
@pytest.mark.parametrize("name", sorted(COOP))
def test_cooperative_event_loop_equals_oracle(oracle, name, ring, topup):
    kw, m, max_clock = COOP[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    n = kw["num_nodes"]
    seeds = np.arange(7, 7 + m, dtype=np.uint64) * 31337
    a = oracle.run_batch(cfg, seeds, max_clock, threads=8, history_cap=128)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=8, history_cap=128, qcap=max(4096, 8 * n * n),
                                   scap=(n * n + 8 * n) if kw.get("quirks", 0) & 1 else 16 * n, bcap=512, lcap=512, ql=0, qheap=1, qcal=1,
                                   ring=ring, ring_topup=topup)
    assert not b["faults"].any()
    for key in ("commit_counts", "active_rounds", "last_states", "histories"):
        assert (a[key] == b[key]).all(), key
    ca, cb = a["counters"], b["counters"]
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert ca[key] == cb[key], key


def test_cooperative_event_loop_small_capacities_fault_cleanly(oracle):
    """Queue / snapshot overflow inside a bulk send raises the fault bit (no out-of-bounds slot, no hang)."""
    kw = dict(num_nodes=40)
    cfg = oracle.make_config(math_mode=1, **kw)
    seeds = np.arange(1, 3, dtype=np.uint64)
    b = oracle.hostmodel_run_batch(cfg, seeds, 200, threads=2, history_cap=8, qcap=300, scap=640, bcap=128, lcap=128, ql=0, qheap=1, qcal=1, ring=128)
    assert (b["faults"] & 1).all()   # F_QUEUE_OVERFLOW
    b = oracle.hostmodel_run_batch(cfg, seeds, 200, threads=2, history_cap=8, qcap=16 * 1600, scap=3, bcap=128, lcap=128, ql=0, qheap=1, qcal=1, ring=128)
    assert (b["faults"] & 2).all()   # F_SNAP_OVERFLOW


# qcal = 1: calendar event queue (one FIFO per (time, kind) bucket) instead of the binary heap
@pytest.mark.parametrize("qcal", [0, 1])
@pytest.mark.parametrize("name", sorted(LARGE))
def test_host_model_large_networks(oracle, name, qcal):
    kw, m, max_clock, qheap = LARGE[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    n = kw["num_nodes"]
    seeds = np.arange(7, 7 + m, dtype=np.uint64) * 31337
    a = oracle.run_batch(cfg, seeds, max_clock, threads=8, history_cap=128)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=8, history_cap=128, qcap=max(4096, 8 * n * n), scap=16 * n,
                                   bcap=512, lcap=512, ql=7 if n in (8, 40) else 0, qheap=qheap, qcal=qcal)
    assert not b["faults"].any()
    for key in ("commit_counts", "active_rounds", "last_states", "histories"):
        assert (a[key] == b[key]).all(), key
    ca, cb = a["counters"], b["counters"]
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert ca[key] == cb[key], key
    assert ca["response_inserts"] == 0


def test_generic_class_equals_specialised(oracle):
    """SimT<K_GENERIC> (everything decided at run time: what the init / read-back kernels instantiate) == SimT<K_SMALL / K_MID / K_LARGE>."""
    for kw, m, max_clock, qheap in (LARGE["n4_heap_mode"], LARGE["n33_two_mask_words"], (dict(num_nodes=4), 64, 1000, 0)):
        cfg = oracle.make_config(math_mode=1, **kw)
        seeds = np.arange(1, m + 1, dtype=np.uint64)
        n = kw["num_nodes"]
        caps = dict(qcap=max(4096, 8 * n * n), scap=16 * n, bcap=512, lcap=512, ql=9, qheap=qheap, history_cap=64)
        a = oracle.hostmodel_run_batch(cfg, seeds, max_clock, force_generic=0, **caps)
        b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, force_generic=1, **caps)
        for key in ("commit_counts", "active_rounds", "last_states", "histories"):
            assert (a[key] == b[key]).all(), key
        assert a["counters"] == b["counters"]


def test_capacity_overflow_raises_fault_not_garbage(oracle):
    cfg = oracle.make_config(num_nodes=4, math_mode=1)
    seeds = np.arange(1, 9, dtype=np.uint64)
    b = oracle.hostmodel_run_batch(cfg, seeds, 1000, qcap=16, scap=4, bcap=8, lcap=8)
    assert (b["faults"] != 0).all()


def test_strict_math_equals_libm_mode(oracle):
    # math_mode 0 (host libm, what the Rust reference calls) and 1 (lbft_math.h, what the GPU runs)
    seeds = np.arange(1, 257, dtype=np.uint64)
    a = oracle.run_batch(oracle.make_config(num_nodes=4, math_mode=0), seeds, 1000, threads=8, history_cap=64)
    b = oracle.run_batch(oracle.make_config(num_nodes=4, math_mode=1), seeds, 1000, threads=8, history_cap=64)
    assert (a["histories"] == b["histories"]).all() and (a["last_states"] == b["last_states"]).all()


def round_switch_rows(rs, mr):
    """rows of round_switches.txt from the [rcap][n] trace: rounds 0 .. max_round - 1, None = empty cell."""
    lo = np.iinfo(np.int64).min
    return [[None if v == lo else int(v) for v in rs[r]] for r in range(int(mr))]


@pytest.mark.parametrize("n,max_clock", [(3, 1000), (4, 2000), (8, 600), (5, 1500)])
def test_round_switch_trace_equals_data_writer(oracle, n, max_clock):
    """DataWriter (bft-lib/src/data_writer.rs): the model examines only the previous event's node and re-checks after a
    timer that had folded duplicates; the oracle scans every node at every popped event like the reference."""
    kw = dict(num_nodes=n) if n != 5 else dict(num_nodes=5, mean=10.0, variance=400.0)  # long tail: timeouts, skipped rounds
    cfg = oracle.make_config(math_mode=1, **kw)
    seeds = np.arange(50, 58, dtype=np.uint64)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=4, qcap=4096, scap=64, bcap=512, lcap=512, ql=11, rcap=400)
    assert not b["faults"].any()
    for i, seed in enumerate(seeds):
        sim = oracle.OracleSim(cfg, int(seed)).enable_data_writer()
        sim.run_until(max_clock)
        rows, messages = sim.round_switches()
        assert round_switch_rows(b["round_switches"][i], b["max_rounds"][i]) == rows
        assert len(rows) > 3


# Equivocating leaders (extension; oracle/lbft_oracle.h "Equivocators" is the specification)
EQUIV = {
    "n4_one_equivocator": (dict(num_nodes=4, equivocate_every=4), 64, 1000),
    "n4_all_equivocate": (dict(num_nodes=4, equivocate_every=1), 16, 600),
    "n7_every_third": (dict(num_nodes=7, equivocate_every=3), 32, 1000),
    "n16_every_fifth_long_tail": (dict(num_nodes=16, equivocate_every=5, mean=10.0, variance=400.0), 8, 600),
    "n40_every_fifth": (dict(num_nodes=40, equivocate_every=5), 2, 300),
    "n5_weighted_epochs": (dict(num_nodes=5, equivocate_every=2, voting_rights=[1, 3, 1, 2, 2], commands_per_epoch=7), 32, 1500),
}


@pytest.mark.parametrize("qcal", [0, 1])
@pytest.mark.parametrize("name", sorted(EQUIV))
def test_host_model_equivocators(oracle, name, qcal):
    kw, m, max_clock = EQUIV[name]
    n = kw["num_nodes"]
    cfg = oracle.make_config(math_mode=1, **kw)
    seeds = np.arange(300, 300 + m, dtype=np.uint64)
    a = oracle.run_batch(cfg, seeds, max_clock, threads=8, history_cap=128)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=8, history_cap=128, qcap=max(4096, 8 * n * n), scap=max(64, 16 * n),
                                   bcap=1024, lcap=512, ql=13, qheap=1 if n > 4 else 0, qcal=qcal)
    assert not b["faults"].any()
    for key in ("commit_counts", "active_rounds", "last_states", "histories"):
        assert (a[key] == b[key]).all(), key
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert a["counters"][key] == b["counters"][key], key
    # fuzzing property: with fewer than a third of the voting power equivocating, the logs of any two nodes of a
    # network are prefix-consistent (safety), whatever happens to liveness
    k = kw["equivocate_every"]
    rights = kw.get("voting_rights", [1] * n)
    if 3 * sum(rights[i] for i in range(0, n, k)) < sum(rights):
        h, cc = a["histories"], a["commit_counts"]
        for i in range(m):
            longest = h[i, int(cc[i].argmax())]
            for node in range(n):
                c = int(cc[i, node])
                assert (h[i, node, :min(c, 128)] == longest[:min(c, 128)]).all()


# Lossy network (extension; oracle/lbft_oracle.h "Lossy network" is the specification): random loss and a partition
LOSSY = {
    "drop_5_percent": (dict(num_nodes=4, drop_per_million=50000), 64, 1500),
    "drop_30_percent_n7": (dict(num_nodes=7, drop_per_million=300000), 32, 1500),
    "partition_2_2": (dict(num_nodes=4, partition_size=2, partition_start=200, partition_end=700), 64, 1500),
    "partition_minority_n7": (dict(num_nodes=7, partition_size=2, partition_start=100, partition_end=600), 32, 1200),
    "drop_and_partition_n40": (dict(num_nodes=40, drop_per_million=20000, partition_size=13, partition_start=50, partition_end=150), 2, 300),
    "drop_equivocators_long_tail": (dict(num_nodes=7, drop_per_million=100000, equivocate_every=4, mean=10.0, variance=400.0), 32, 1500),
}


@pytest.mark.parametrize("qcal", [0, 1])
@pytest.mark.parametrize("name", sorted(LOSSY))
def test_host_model_lossy_network(oracle, name, qcal):
    kw, m, max_clock = LOSSY[name]
    n = kw["num_nodes"]
    cfg = oracle.make_config(math_mode=1, **kw)
    seeds = np.arange(900, 900 + m, dtype=np.uint64)
    a = oracle.run_batch(cfg, seeds, max_clock, threads=8, history_cap=128)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=8, history_cap=128, qcap=max(4096, 8 * n * n), scap=max(64, 16 * n),
                                   bcap=1024, lcap=512, ql=13, qheap=1 if n > 4 else 0, qcal=qcal)
    assert not b["faults"].any()
    for key in ("commit_counts", "active_rounds", "last_states", "histories"):
        assert (a[key] == b[key]).all(), key
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert a["counters"][key] == b["counters"][key], key
    # safety under message loss: logs stay prefix-consistent
    h, cc = a["histories"], a["commit_counts"]
    for i in range(m):
        longest = h[i, int(cc[i].argmax())]
        for node in range(n):
            c = min(int(cc[i, node]), 128)
            assert (h[i, node, :c] == longest[:c]).all()
    if name == "partition_2_2":  # no quorum on either side while the cut lasts: commits stall at what was reached before it
        healthy = oracle.run_batch(oracle.make_config(math_mode=1, num_nodes=4), seeds, max_clock, threads=8)
        assert a["commit_counts"].max() <= 12 and healthy["commit_counts"].min() >= 40


# quirks bit 1: EpochId::previous() = id - 1 (reference quirk Q2 fixed): epoch changes no longer stall the network
Q2 = {
    "n4_cpe50": (dict(num_nodes=4, commands_per_epoch=50, quirks=2), 64, 3000),
    "n3_cpe5": (dict(num_nodes=3, commands_per_epoch=5, quirks=2), 64, 1500),
    "n7_weighted_cpe9": (dict(num_nodes=7, commands_per_epoch=9, quirks=2, voting_rights=[2, 1, 1, 3, 1, 2, 1]), 32, 2000),
    "n36_cpe3": (dict(num_nodes=36, commands_per_epoch=3, quirks=2), 2, 300),
}


@pytest.mark.parametrize("name", sorted(Q2))
def test_host_model_q2_fixed(oracle, name):
    kw, m, max_clock = Q2[name]
    n = kw["num_nodes"]
    cfg = oracle.make_config(math_mode=1, **kw)
    seeds = np.arange(3, 3 + m, dtype=np.uint64)
    a = oracle.run_batch(cfg, seeds, max_clock, threads=8, history_cap=256)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=8, history_cap=256, qcap=max(4096, 8 * n * n), scap=max(64, 16 * n),
                                   bcap=1024, lcap=1024, ql=13, qheap=1 if n > 4 else 0, qcal=1 if n > 4 and max_clock <= 2047 else 0)
    assert not b["faults"].any()
    for key in ("commit_counts", "active_rounds", "last_states", "histories"):
        assert (a[key] == b[key]).all(), key
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert a["counters"][key] == b["counters"][key], key
    if name == "n4_cpe50":
        # under reference semantics this network stalls at [50, 50, 49, 49]; with previous() fixed it stays live
        stalled = oracle.run_batch(oracle.make_config(math_mode=1, num_nodes=4, commands_per_epoch=50), seeds, max_clock, threads=8)
        assert stalled["commit_counts"].max() <= 50
        assert (a["commit_counts"].min(axis=1) >= 90).all()


# quirks bit 0: requests are answered by the peer with real payloads (reference quirk Q1 fixed); with bit 1 too = "fixed" mode
Q1 = {
    "q1_n4": (dict(num_nodes=4, quirks=1), 64, 1500),
    "q1_n4_long_tail": (dict(num_nodes=4, quirks=1, mean=10.0, variance=400.0), 64, 2500),
    "q3_n4_cpe50": (dict(num_nodes=4, quirks=3, commands_per_epoch=50), 64, 3000),
    "q1_n4_cpe50": (dict(num_nodes=4, quirks=1, commands_per_epoch=50), 64, 3000),
    "q3_n3_cpe5": (dict(num_nodes=3, quirks=3, commands_per_epoch=5), 64, 1500),
    "q3_n7_weighted_cpe9": (dict(num_nodes=7, quirks=3, commands_per_epoch=9, voting_rights=[2, 1, 1, 3, 1, 2, 1]), 32, 2000),
    "q1_n8_lossy": (dict(num_nodes=8, quirks=1, drop_per_million=150000), 16, 1500),
    "q3_n5_partition_heals": (dict(num_nodes=5, quirks=3, partition_size=2, partition_start=100, partition_end=400, commands_per_epoch=20), 32, 2500),
    "q3_n36_cpe3": (dict(num_nodes=36, quirks=3, commands_per_epoch=3), 2, 300),
    "q1_n7_equivocators": (dict(num_nodes=7, quirks=1, equivocate_every=3), 32, 1500),
}


@pytest.mark.parametrize("name", sorted(Q1))
def test_host_model_q1_fixed(oracle, name):
    kw, m, max_clock = Q1[name]
    n = kw["num_nodes"]
    cfg = oracle.make_config(math_mode=1, **kw)
    seeds = np.arange(40, 40 + m, dtype=np.uint64)
    a = oracle.run_batch(cfg, seeds, max_clock, threads=8, history_cap=256)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=8, history_cap=256, qcap=max(4096, 16 * n * n), scap=max(128, 128 * n),
                                   bcap=1024, lcap=1024, ql=13, qheap=1 if n > 4 else 0, qcal=1 if max_clock <= 2047 else 0)
    assert not b["faults"].any()
    for key in ("commit_counts", "active_rounds", "last_states", "histories"):
        assert (a[key] == b[key]).all(), key
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert a["counters"][key] == b["counters"][key], key
    if name == "q3_n5_partition_heals":
        # with both quirks fixed a healed partition catches up (the reference semantics never do, see LOSSY partition test)
        assert (a["commit_counts"].min(axis=1) >= 25).mean() > 0.8


# Epoch reconfiguration (extension): the voting rights of epoch e are voting_rights[(i + e * rights_rotation) % n]
ROT = {
    "rot1_n4_q3_cpe5": (dict(num_nodes=4, quirks=3, commands_per_epoch=5, voting_rights=[1, 2, 3, 4], rights_rotation=1), 64, 2500),
    "rot3_n7_q3_cpe9": (dict(num_nodes=7, quirks=3, commands_per_epoch=9, voting_rights=[2, 1, 1, 3, 1, 2, 1], rights_rotation=3), 32, 2000),
    "rot1_n4_q2_cpe7": (dict(num_nodes=4, quirks=2, commands_per_epoch=7, voting_rights=[5, 1, 1, 1], rights_rotation=1), 64, 2000),
    "rot2_n5_reference_quirks": (dict(num_nodes=5, quirks=0, commands_per_epoch=10, voting_rights=[1, 1, 2, 2, 3], rights_rotation=2), 32, 1500),
    "rot5_n36_q3_cpe3": (dict(num_nodes=36, quirks=3, commands_per_epoch=3, voting_rights=[1 + (i % 3) for i in range(36)], rights_rotation=5), 2, 300),
    "rot7_n100_q3_cpe2": (dict(num_nodes=100, quirks=3, commands_per_epoch=2, voting_rights=[1 + (i % 4) for i in range(100)], rights_rotation=7), 2, 260),
    "rot1_n4_q3_lossy": (dict(num_nodes=4, quirks=3, commands_per_epoch=6, voting_rights=[1, 2, 3, 4], rights_rotation=1, drop_per_million=100000), 32, 2500),
}


@pytest.mark.parametrize("name", sorted(ROT))
def test_host_model_rotating_voting_rights(oracle, name):
    kw, m, max_clock = ROT[name]
    n = kw["num_nodes"]
    cfg = oracle.make_config(math_mode=1, **kw)
    seeds = np.arange(7, 7 + m, dtype=np.uint64)
    a = oracle.run_batch(cfg, seeds, max_clock, threads=8, history_cap=512)
    b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=8, history_cap=512, qcap=max(4096, 16 * n * n), scap=max(128, 128 * n),
                                   bcap=2048, lcap=2048, ql=13, qheap=1 if n > 4 else 0, qcal=1 if n > 4 and max_clock <= 2047 else 0)
    assert not b["faults"].any()
    for key in ("commit_counts", "active_rounds", "last_states", "histories"):
        assert (a[key] == b[key]).all(), key
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert a["counters"][key] == b["counters"][key], key
    # the rotation really changes the run (other leaders, other quorums) ...
    static = oracle.run_batch(oracle.make_config(math_mode=1, **dict(kw, rights_rotation=0)), seeds, max_clock, threads=8, history_cap=512)
    if kw["quirks"] & 2:
        assert (static["last_states"] != a["last_states"]).any()
        # ... the networks cross many epochs ...
        if n <= 7:
            assert (a["commit_counts"].min(axis=1) >= 4 * kw["commands_per_epoch"]).mean() > 0.5
    # ... and safety holds: the committed histories of the nodes of a network are prefixes of one another
    cc, h = a["commit_counts"], a["histories"]
    for i in range(m):
        longest = h[i, int(np.argmax(cc[i]))]
        for node in range(n):
            c = min(int(cc[i, node]), 512)
            assert (h[i, node, :c] == longest[:c]).all()


def test_rotating_equal_rights_changes_nothing(oracle):
    seeds = np.arange(1, 33, dtype=np.uint64)
    base = oracle.run_batch(oracle.make_config(math_mode=1, num_nodes=4, quirks=3, commands_per_epoch=5), seeds, 1500, threads=8)
    rot = oracle.run_batch(oracle.make_config(math_mode=1, num_nodes=4, quirks=3, commands_per_epoch=5, rights_rotation=1), seeds, 1500, threads=8)
    assert (base["last_states"] == rot["last_states"]).all() and (base["commit_counts"] == rot["commit_counts"]).all()


# Byte-exact record hashing (first half): the Block_ / QuorumCertificate_ records behind a node's committed
# history, hashed like the reference's SmrContext::hash (SipHash-1-3 of "Name::" + BCS).  The oracle keeps real records in maps
# keyed by these hashes; the compact model recomputes them from its block pool (structural ids, recorded QC voter sets).
HASHES = {
    "golden_n3": (dict(num_nodes=3), 52, 1000),
    "n4": (dict(num_nodes=4), 7, 1000),
    "n7_weighted_epochs_q2": (dict(num_nodes=7, voting_rights=[2, 1, 1, 3, 1, 2, 1], commands_per_epoch=9, quirks=2), 3, 2000),
    "n4_rotating_rights_q3": (dict(num_nodes=4, commands_per_epoch=5, quirks=3, voting_rights=[1, 2, 3, 4], rights_rotation=1), 11, 1500),
    "n7_equivocators": (dict(num_nodes=7, equivocate_every=3), 5, 1000),
    "n40_long_tail": (dict(num_nodes=40, mean=10.0, variance=400.0), 9, 300),
    "n4_lossy_long_tail": (dict(num_nodes=4, mean=10.0, variance=400.0, drop_per_million=50000), 4, 1500),
    "n5_q1_partition": (dict(num_nodes=5, quirks=3, partition_size=2, partition_start=100, partition_end=400, commands_per_epoch=20), 8, 2000),
}


def assert_record_hashes_equal(oracle, cfg, seed, max_clock, n, device_hashes):
    """device_hashes(node) -> structured or [k][4] array for that node; compared with the oracle's own records."""
    sim = oracle.OracleSim(cfg, int(seed)).run_until(max_clock)
    total = 0
    for node in range(n):
        ref = sim.committed_record_hashes(node)
        got = device_hashes(node, len(ref))
        assert ref["has_qc"].all()
        assert (got[:, 0] == ref["block_hash"]).all(), node
        assert (got[:, 1] == ref["state"]).all(), node
        assert (got[:, 2] == ref["qc_hash"]).all(), node
        assert ((got[:, 3] & 0xffffffff) == ref["num_votes"]).all() and ((got[:, 3] >> 32) == 0).all(), node
        if len(ref):
            assert int(ref["state"][-1]) == sim.last_committed_states()[node]
        total += len(ref)
    return total


@pytest.mark.parametrize("name", sorted(HASHES))
def test_committed_record_hashes_equal_the_oracle(oracle, name):
    kw, seed, max_clock = HASHES[name]
    n = kw["num_nodes"]
    cfg = oracle.make_config(math_mode=1, **kw)
    b = oracle.hostmodel_run_batch(cfg, np.array([seed], dtype=np.uint64), max_clock, qcap=max(4096, 16 * n * n),
                                   scap=max(128, 128 * n) if kw.get("quirks", 0) & 1 else 64 if n <= 4 else 8 * n, bcap=1024, lcap=1024,
                                   ql=16 if n <= 4 else 0, qheap=1 if n > 4 else 0, hash_cap=512)
    assert not b["faults"].any()
    total = assert_record_hashes_equal(oracle, cfg, seed, max_clock, n, lambda node, c: b["record_hashes"][0, node, :c])
    assert total >= n // 2  # something was committed
