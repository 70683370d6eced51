"""Replays of two unit scenarios of the reference that round 2 left out (missing #4), against the oracle, the host
build of the kernel logic (CPU) and the HIP path (-m gpu):

* librabft-v2/src/unit_tests/node_tests.rs:9-76 -- a one-node network with 2 commands per epoch: save_node -> load_node -> equal
  (here: image -> reader -> writer -> the same bytes, tests/bincode_nodestate.py), then a block and its quorum certificate are
  inserted and `highest_quorum_certificate_hash` must be the hash of that certificate;
* bft-lib/src/unit_tests/simulated_context_tests.rs:79-129 -- fetch / compute / commit / discard and read_epoch_id at 2 commands per
  epoch: commands are (proposer, 0..) in fetch order, a state of k commands belongs to epoch k / 2, forks are discarded, and the
  committed history is exactly the committed chain."""
import numpy as np
import pytest

from bincode_nodestate import dump_node_state, node_state


def _sip13_words(words):
    def rotl(x, b):
        return ((x << b) | (x >> (64 - b))) & 0xFFFFFFFFFFFFFFFF
    v0, v1, v2, v3 = 0x736f6d6570736575, 0x646f72616e646f6d, 0x6c7967656e657261, 0x7465646279746573
    M = 0xFFFFFFFFFFFFFFFF

    def rnd(v0, v1, v2, v3):
        v0 = (v0 + v1) & M; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32)
        v2 = (v2 + v3) & M; v3 = rotl(v3, 16); v3 ^= v2
        v0 = (v0 + v3) & M; v3 = rotl(v3, 21); v3 ^= v0
        v2 = (v2 + v1) & M; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32)
        return v0, v1, v2, v3
    for m in words:
        m &= M
        v3 ^= m
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
        v0 ^= m
    b = ((len(words) * 8) & 0xFF) << 56
    v3 ^= b
    v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    v0 ^= b
    v2 ^= 0xFF
    for _ in range(3):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    return v0 ^ v1 ^ v2 ^ v3


def _check_node_tests_image(img):
    n = node_state(img)
    assert dump_node_state(n) == img                                      # save -> load -> save (node_tests.rs:17-21)
    rs = n["record_store"]
    if rs["quorum_certificates"]:                                         # node_tests.rs:69-75
        by_round = {q["round"]: (h, q) for h, q in rs["quorum_certificates"]}
        h, q = by_round[rs["highest_quorum_certificate_round"]]
        assert rs["highest_quorum_certificate_hash"] == h
        assert q["certified_block_hash"] in dict(rs["blocks"]) and q["votes"] == [(0, (0, q["votes"][0][1][1]))] and q["author"] == 0
    else:
        assert rs["highest_quorum_certificate_hash"] == rs["initial_hash"] and rs["highest_quorum_certificate_round"] == 0
    assert n["epoch_id"] == len(n["past_record_stores"])
    return n


NODE_TESTS_CFG = dict(num_nodes=1, commands_per_epoch=2)  # SimulatedContext::new(Author(0), 1, 2), node_tests.rs:11-15


def test_node_tests_save_load_round_trip_oracle_and_host_model(oracle):
    cfg = oracle.make_config(math_mode=1, **NODE_TESTS_CFG)
    seen_qc = False
    for t in (0, 12, 13, 14, 20, 60):
        img = oracle.OracleSim(cfg, 1).run_until(t).save_node(0)
        n = _check_node_tests_image(img)
        seen_qc |= bool(n["record_store"]["quorum_certificates"]) or n["epoch_id"] > 0
        if t == 0:  # make_initial_state (node.rs:87-114): nothing inserted yet
            assert n["record_store"]["blocks"] == [] and n["record_store"]["current_round"] == 1 and n["latest_voted_round"] == 0
        images = oracle.hostmodel_node_images(cfg, 1, t, qcap=256, scap=64, bcap=512, lcap=512, ql=0, keep_stores=1)
        assert images[0] == img
    assert seen_qc


@pytest.mark.gpu
def test_node_tests_save_load_round_trip_device(oracle):
    import librabft_simulator_amd as amd
    cfg = oracle.make_config(math_mode=1, **NODE_TESTS_CFG)
    for t in (0, 13, 60):
        img = oracle.OracleSim(cfg, 1).run_until(t).save_node(0)
        res = amd.BatchSimulator.new(np.array([1], dtype=np.uint64), 1, amd.RandomDelay.new(10.0, 4.0), commands_per_epoch=2, keep_retired_stores=True).loop_until(t)
        dev = res.save_node(0, 0)
        _check_node_tests_image(dev)
        assert dev == img


def _check_simulated_context_scenario(hist, cc, epochs, states):
    """hist [node][k] of (proposer, index, time); the properties simulated_context_tests.rs:79-129 asserts, on every node's ledger."""
    for node in range(len(cc)):
        h = [(int(p), int(i), int(t)) for p, i, t in hist[node][:cc[node]]]
        per_proposer = {}
        for p, i, t in h:                                                  # CommandFetcher::fetch: (author, 0), (author, 1), ... in order;
            assert i >= per_proposer.get(p, -1) + 1                        # a fork's command (c3 in the test) is fetched but never committed
            per_proposer[p] = i
        assert epochs[node] == cc[node] // 2                               # read_epoch_id = commands / max_commands_per_epoch
        words = [len(h)] + [w for e in h for w in e]
        assert states[node] == _sip13_words(words)                         # State = hash of the committed execution_history
    longest = max(range(len(cc)), key=lambda k: cc[k])
    for node in range(len(cc)):                                            # one chain: commit extends, discard drops the fork
        assert [tuple(x) for x in hist[node][:cc[node]]] == [tuple(x) for x in hist[longest][:cc[node]]]


SC_CFG = dict(num_nodes=2, commands_per_epoch=2, quirks=3)  # SimulatedContext::new(Author(0), 2, 2), simulated_context_tests.rs:81-85 (live across epochs: quirks 3)


def test_simulated_context_scenario_oracle_and_host_model(oracle):
    cfg = oracle.make_config(math_mode=1, **SC_CFG)
    seeds = np.arange(1, 9, dtype=np.uint64)
    ref = oracle.run_batch(cfg, seeds, 300, history_cap=64)
    hm = oracle.hostmodel_run_batch(cfg, seeds, 300, history_cap=64, qcap=512, scap=128, bcap=512, lcap=512, ql=0)
    assert (ref["commit_counts"] == hm["commit_counts"]).all() and (ref["last_states"] == hm["last_states"]).all()
    assert (ref["histories"] == hm["histories"]).all()
    assert ref["commit_counts"].min() >= 4                                 # several epoch changes at every node
    for i in range(len(seeds)):
        hist = [[(h["proposer"], h["index"], h["time"]) for h in ref["histories"][i, node]] for node in range(2)]
        _check_simulated_context_scenario(hist, ref["commit_counts"][i], ref["commit_counts"][i] // 2, [int(x) for x in ref["last_states"][i]])


@pytest.mark.gpu
def test_simulated_context_scenario_device(oracle):
    import librabft_simulator_amd as amd
    cfg = oracle.make_config(math_mode=1, **SC_CFG)
    seeds = np.arange(1, 9, dtype=np.uint64)
    ref = oracle.run_batch(cfg, seeds, 300, history_cap=64)
    res = amd.BatchSimulator.new(seeds, 2, amd.RandomDelay.new(10.0, 4.0), commands_per_epoch=2, quirks=3).loop_until(300)
    cc, hist = res.commit_counts, res.committed_histories(64)
    assert (cc == ref["commit_counts"]).all() and (res.last_committed_states == ref["last_states"]).all() and (hist == ref["histories"]).all()
    for i in range(len(seeds)):
        h = [[(x["proposer"], x["index"], x["time"]) for x in hist[i, node]] for node in range(2)]
        _check_simulated_context_scenario(h, cc[i], res.epochs[i], [int(x) for x in res.last_committed_states[i]])


def test_time_conversion_of_the_host_mirror():
    """bft-lib/src/unit_tests/simulator_tests.rs:6-12, on the Python mirror of GlobalTime / NodeTime (simulator.rs:120-126), and
    base_type_tests.rs:6-9's arithmetic on the wrapped integers."""
    from librabft_simulator_amd import Duration, GlobalTime, NodeTime
    x, start = GlobalTime(15), GlobalTime(3)
    assert x.to_node_time(start) == NodeTime(12) and isinstance(x.to_node_time(start), NodeTime)
    assert GlobalTime.from_node_time(NodeTime(12), start) == x
    assert GlobalTime(3) + Duration(4) == GlobalTime(7) and repr(GlobalTime(3) + Duration(4)) == "GlobalTime(7)"
