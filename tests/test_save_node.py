"""ConsensusNode::save_node (librabft-v2/src/node.rs:233-238): the bincode image of a node's whole NodeState.

The reference holds no byte-level known answer for it (its HashMaps serialise in per-process order): parity is
UNPINNED against the reference; what is checked is (a) the oracle's image parses as the struct layout of the reference's source
(tests/bincode_nodestate.py, written from the struct definitions) and agrees with the oracle's own node view, and (b) the
device rebuilds the identical image -- every record, hash and signature -- from its structural state (-m gpu)."""
import numpy as np
import pytest

from bincode_nodestate import node_state

CASES = {
    "golden_n3": (dict(num_nodes=3), 52, (1000, 137, 455)),
    "n4_long_tail_timeouts": (dict(num_nodes=4, mean=10.0, variance=400.0), 7, (1500, 333)),
    "n7_weighted": (dict(num_nodes=7, voting_rights=[2, 1, 1, 3, 1, 2, 1]), 1234567, (800,)),
    "n5_equivocators": (dict(num_nodes=5, equivocate_every=2), 99, (700,)),
    "n5_fast_pacemaker": (dict(num_nodes=5, delta=3, gamma=1.5, lambda_=0.25, target_commit_interval=80), 5, (600,)),
    "n40": (dict(num_nodes=40), 3, (250,)),
    "n64_long_tail": (dict(num_nodes=64, mean=10.0, variance=400.0), 11, (300,)),
    "n4_q1_lossy": (dict(num_nodes=4, quirks=1, drop_per_million=100000), 21, (900,)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_image_has_the_reference_layout(oracle, name):
    kw, seed, horizons = CASES[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    for t in horizons:
        sim = oracle.OracleSim(cfg, seed).run_until(t)
        for node in range(kw["num_nodes"]):
            img = sim.save_node(node)
            n = node_state(img)  # consumes the image exactly
            rs = n["record_store"]
            view = sim.node_view(node)
            assert rs["current_round"] == view["current_round"] and rs["highest_committed_round"] == view["highest_committed_round"]
            assert rs["highest_quorum_certificate_round"] == view["highest_quorum_certificate_round"]
            assert n["pacemaker"]["active_round"] == view["active_round"] and n["latest_voted_round"] == view["latest_voted_round"]
            assert len(rs["current_timeouts"]) == view["num_current_timeouts"] and len(rs["current_votes"]) == view["num_current_votes"]
            assert [k for k, _ in rs["blocks"]] == sorted(k for k, _ in rs["blocks"])  # canonical order
            for h, b in rs["blocks"]:
                assert b["signature"] == (b["author"], h)  # Signature(author, hash of the record) (simulated_context.rs:259-261)
            for h, q in rs["quorum_certificates"]:
                assert q["signature"] == (q["author"], h) and len(q["votes"]) >= 1
            assert n["past_record_stores"] == [] and n["epoch_id"] == 0
            assert sim.save_node(node) == img  # deterministic


@pytest.mark.parametrize("name", sorted(CASES))
def test_image_builder_on_the_compact_model_equals_oracle_image(oracle, name):
    """csrc/lbft_save_node.h -- the builder the product library runs on the rows it copies back from the GPU -- applied to the host
    model's state: every record, hash and signature of every node equals the oracle's image, byte for byte."""
    kw, seed, horizons = CASES[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    n = kw["num_nodes"]
    special = any(k in kw for k in ("equivocate_every", "drop_per_million")) or kw.get("quirks", 0) & 1
    for t in horizons:
        sim = oracle.OracleSim(cfg, seed).run_until(t)
        caps = dict(qcap=max(4096, 8 * n * n), scap=(n * n + 8 * n + 64) if kw.get("quirks", 0) & 1 else max(64, 16 * n), bcap=512, lcap=512,
                    ql=48 if n <= 16 and not special else 0, qheap=1 if n > 16 else 0, qcal=1 if n > 32 else 0, ring=256 if n > 32 else 0, tw=8 if n > 32 else 0)
        rt = []
        images = oracle.hostmodel_node_images(cfg, seed, t, roundtrip=rt, **caps)
        for node in range(n):
            a, b = sim.save_node(node), images[node]
            assert b is not None
            if a != b:
                assert node_state(b) == node_state(a), (name, t, node)  # (a readable diff first)
            assert a == b, (name, t, node)
        # ConsensusNode::load_node (node.rs:211-231) through the product library's loader (csrc/lbft_save_node.h load_node_image): every
        # node's NodeState scrubbed, the image loaded, saved again == the image; a node time before the image's own times is refused
        assert rt == [0] * n, (name, t, rt)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_device_image_equals_oracle_image(oracle, name):
    import librabft_simulator_amd as amd
    kw, seed, horizons = CASES[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    kwd = dict(kw)
    n = kwd.pop("num_nodes")
    delay = amd.RandomDelay.new(kwd.pop("mean", 10.0), kwd.pop("variance", 4.0))
    nc = amd.NodeConfig(kwd.pop("target_commit_interval", 100000), kwd.pop("delta", 20), kwd.pop("gamma", 2.0), kwd.pop("lambda_", 0.5))
    for t in horizons:
        sim = oracle.OracleSim(cfg, seed).run_until(t)
        res = amd.BatchSimulator.new(np.array([seed + 1, seed], dtype=np.uint64), n, delay, nc, **kwd).loop_until(t)
        for node in range(n):
            a, b = sim.save_node(node), res.save_node(1, node)
            if a != b:
                assert node_state(b) == node_state(a), (name, t, node)  # (a readable diff first)
            assert a == b, (name, t, node)


@pytest.mark.gpu
def test_device_image_follows_the_state_across_runs_of_one_batch(oracle):
    """The library keeps the last image it built (size query + fill = one build); anything that writes the state rows must drop it."""
    import librabft_simulator_amd as amd
    seed = 9
    cfg = oracle.make_config(math_mode=1, num_nodes=4)
    batch = amd.BatchSimulator.new(np.array([seed], dtype=np.uint64), 4, amd.RandomDelay.new(10.0, 4.0), amd.NodeConfig())
    seen = set()
    for t in (300, 700, 300):
        batch.reset()
        res = batch.loop_until(t)
        for node in (2, 2, 0):
            img = res.save_node(0, node)
            assert img == oracle.OracleSim(cfg, seed).run_until(t).save_node(node), (t, node)
            seen.add(img)
    assert len(seen) == 4


# Nodes that have changed epoch carry their retired record stores (past_record_stores, node.rs:43,233-238,331-348): c5live-shaped runs.
EPOCH_CASES = {
    "n4_q3_cpe5": (dict(num_nodes=4, commands_per_epoch=5, quirks=3), 5, (400, 1000)),
    "n4_reference_quirks_stall_cpe50": (dict(num_nodes=4, commands_per_epoch=50), 3, (3000,)),                    # epochs [1, 1, 0, 0] for ever
    "n7_rotating_rights_q3_cpe3": (dict(num_nodes=7, voting_rights=[2, 1, 1, 3, 1, 2, 1], commands_per_epoch=3, quirks=3, rights_rotation=1), 11, (700,)),
    "n1_cpe2": (dict(num_nodes=1, commands_per_epoch=2), 1, (0, 30, 200)),                                        # node_tests.rs:11-15's context
    "n40_weighted_rotating_q3_cpe3": (dict(num_nodes=40, voting_rights=[1 + (i % 4) for i in range(40)], commands_per_epoch=3, quirks=3, rights_rotation=1), 7, (450,)),
}


@pytest.mark.parametrize("name", sorted(EPOCH_CASES))
def test_image_builder_serves_nodes_that_changed_epoch(oracle, name):
    """Host model rows (retired stores archived: keep_stores) -> csrc/lbft_save_node.h == the oracle's image incl. every past record store."""
    kw, seed, horizons = EPOCH_CASES[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    n = kw["num_nodes"]
    for t in horizons:
        sim = oracle.OracleSim(cfg, seed).run_until(t)
        caps = dict(qcap=max(4096, 8 * n * n), scap=(n * n + 8 * n + 64), bcap=1024, lcap=1024, ql=0, qheap=1 if n > 16 else 0, qcal=1 if n > 32 else 0,
                    ring=256 if n > 32 else 0, tw=8 if n > 32 else 0, keep_stores=1)
        rt = []
        images = oracle.hostmodel_node_images(cfg, seed, t, roundtrip=rt, **caps)
        assert rt == [0] * n, (name, t, rt)  # save -> scrub -> load -> save incl. every retired store (load_node_image)
        epochs = []
        for node in range(n):
            a, b = sim.save_node(node), images[node]
            assert b is not None
            if a != b:
                assert node_state(b) == node_state(a), (name, t, node)
            assert a == b, (name, t, node)
            ns = node_state(a)
            assert [e for e, _ in ns["past_record_stores"]] == list(range(ns["epoch_id"]))  # one retired store per epoch left, ascending
            epochs.append(ns["epoch_id"])
        if t >= 400:
            assert max(epochs) >= 1, (name, t, epochs)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(EPOCH_CASES))
def test_device_image_of_nodes_that_changed_epoch_equals_oracle_image(oracle, name):
    """lbft_batch_keep_retired_stores + lbft_batch_save_node == the oracle, byte for byte, after epoch changes (replaces round 2's
    test_device_refuses_nodes_that_changed_epoch)."""
    import librabft_simulator_amd as amd
    kw, seed, horizons = EPOCH_CASES[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    kwd = dict(kw)
    n = kwd.pop("num_nodes")
    for t in horizons:
        sim = oracle.OracleSim(cfg, seed).run_until(t)
        res = amd.BatchSimulator.new(np.array([seed + 1, seed], dtype=np.uint64), n, amd.RandomDelay.new(10.0, 4.0), keep_retired_stores=True, **kwd).loop_until(t)
        for node in range(n):
            a, b = sim.save_node(node), res.save_node(1, node)
            if a != b:
                assert node_state(b) == node_state(a), (name, t, node)
            assert a == b, (name, t, node)
        # the archive changes nothing else: same logs as a batch without it
        plain = amd.BatchSimulator.new(np.array([seed + 1, seed], dtype=np.uint64), n, amd.RandomDelay.new(10.0, 4.0), **kwd).loop_until(t)
        assert (plain.commit_counts == res.commit_counts).all() and (plain.last_committed_states == res.last_committed_states).all()


@pytest.mark.gpu
def test_device_without_the_archive_refuses_nodes_that_changed_epoch():
    """Off by default (it costs num_nodes x epochs x a node's rows of device memory per instance): then a node past its first epoch
    change is refused with LBFT_ERR_UNSUPPORTED, never served a wrong image; a too small buffer is LBFT_ERR_INVALID (round-2 advisor)."""
    import ctypes
    import librabft_simulator_amd as amd
    from librabft_simulator_amd import _lib
    res = amd.BatchSimulator.new(np.array([5], dtype=np.uint64), 4, amd.RandomDelay.new(10.0, 4.0), commands_per_epoch=5, quirks=3).loop_until(1000)
    assert res.epochs.min() >= 1
    with pytest.raises(amd.LbftError) as e:
        res.save_node(0, 0)
    assert e.value.code == -3
    res0 = amd.BatchSimulator.new(np.array([5], dtype=np.uint64), 4, amd.RandomDelay.new(10.0, 4.0)).loop_until(300)
    ln = ctypes.c_size_t(0)
    small = np.zeros(16, dtype=np.uint8)
    rc = _lib.lib().lbft_batch_save_node(res0._sim._h, 0, 0, small.ctypes.data, small.size, ctypes.byref(ln))
    assert rc == -1 and ln.value > 16  # LBFT_ERR_INVALID, *len = the size needed


LOAD_CASES = {
    "n4_reference": (dict(num_nodes=4), 7, 1000, 400),
    "n4_epochs_cpe5_q3": (dict(num_nodes=4, commands_per_epoch=5, quirks=3), 5, 1000, 900),
    "n7_rotating_rights_q3_cpe3": (dict(num_nodes=7, voting_rights=[2, 1, 1, 3, 1, 2, 1], commands_per_epoch=3, quirks=3, rights_rotation=1), 11, 700, 1500),
    "n40_weighted_rotating_q3_cpe3": (dict(num_nodes=40, voting_rights=[1 + (i % 4) for i in range(40)], commands_per_epoch=3, quirks=3, rights_rotation=1), 7, 450, 30000),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(LOAD_CASES))
def test_device_load_node_round_trips_images_of_two_points_in_time(oracle, name):
    """lbft_batch_load_node = ConsensusNode::load_node (node.rs:211-231) on the device.  A batch is advanced in bounded launches; every node
    of instance 1 is saved mid-run (T1) and at the end (T2).  Loading the T1 image into the finished node and saving again returns the T1
    image byte for byte (a real rollback of the NodeState: rounds, certificates, votes, retired stores), loading T2 restores T2; the final
    image also equals the oracle's.  The reference's guard and the unknown-record refusal leave the node untouched."""
    import librabft_simulator_amd as amd
    kw, seed, max_clock, steps = LOAD_CASES[name]
    cfg = oracle.make_config(math_mode=1, **kw)
    kwd = dict(kw)
    n = kwd.pop("num_nodes")
    seeds = np.array([seed + 1, seed], dtype=np.uint64)
    sim = amd.BatchSimulator.new(seeds, n, amd.RandomDelay.new(10.0, 4.0), keep_retired_stores=True, **kwd)
    left, res = sim.run_steps(max_clock, steps)
    assert left > 0  # mid-run
    img1 = [sim.save_node(1, node) for node in range(n)]
    for _ in range(10000):
        left, res = sim.run_steps(max_clock, 4 * steps)
        if left == 0:
            break
    assert left == 0
    img2 = [sim.save_node(1, node) for node in range(n)]
    ora = oracle.OracleSim(cfg, seed).run_until(max_clock)
    assert [ora.save_node(node) for node in range(n)] == img2
    assert any(a != b for a, b in zip(img1, img2))
    far = 1 << 40
    for node in range(n):
        sim.load_node(1, node, img1[node], far)
        assert sim.save_node(1, node) == img1[node], (name, node)
        assert node_state(sim.save_node(1, node)) == node_state(img1[node])
    for node in range(n):  # (the other nodes are back at T1 while this one returns to T2: no cross-talk through the shared block pool)
        sim.load_node(1, node, img2[node], far)
        assert sim.save_node(1, node) == img2[node], (name, node)
    # instance 0 (another seed) never moved
    other = amd.BatchSimulator.new(seeds, n, amd.RandomDelay.new(10.0, 4.0), keep_retired_stores=True, **kwd).loop_until(max_clock)
    assert all(sim.save_node(0, node) == other.save_node(0, node) for node in range(n))
    # node.rs:219-228: a node time before the image's own times
    ns = node_state(img2[0])
    latest = max(ns["latest_query_all_time"], ns["tracker"]["latest_commit_time"], ns["pacemaker"]["active_round_start_time"])
    with pytest.raises(amd.LbftError) as e:
        sim.load_node(1, 0, img2[0], latest - 1)
    assert e.value.code == -4 and "future" in str(e.value)
    sim.load_node(1, 0, img2[0], latest)  # (equal is accepted: node_time >= previous_time)
    # an image of ANOTHER run names records this instance's pool does not hold; a truncated image is malformed
    with pytest.raises(amd.LbftError) as e:
        sim.load_node(1, 0, other.save_node(0, 0), far)
    assert e.value.code == -3
    with pytest.raises(amd.LbftError) as e:
        sim.load_node(1, 0, img2[0][:-5], far)
    assert e.value.code == -1
    # an image with retired record stores needs a batch that archives them (round-4 advisor: it used to load, dropping past_record_stores silently, and
    # the node then failed save_node): refused with LBFT_ERR_UNSUPPORTED, the node untouched
    if node_state(img2[0])["past_record_stores"]:
        plain = amd.BatchSimulator.new(seeds, n, amd.RandomDelay.new(10.0, 4.0), **kwd)
        plain.loop_until(max_clock)
        with pytest.raises(amd.LbftError) as e:
            plain.load_node(1, 0, img2[0], far)
        assert e.value.code == -3 and "keep_retired_stores" in str(e.value)
    assert [sim.save_node(1, node) for node in range(n)] == img2  # every refusal left the node untouched


@pytest.mark.gpu
def test_device_load_node_takes_the_oracle_image_and_the_run_goes_on(oracle):
    """A NodeState saved by ANOTHER implementation of the same run (the oracle = the reference's records and hashes) loads into the device
    node in a node-level session, and the node behaves as the saved one: its next update_node equals the oracle node's."""
    import librabft_simulator_amd as amd
    seed, t = 9, 600
    cfg = oracle.make_config(num_nodes=4, math_mode=1)
    ora = oracle.OracleSim(cfg, seed).run_until(t)
    res = amd.BatchSimulator.new(np.array([seed], dtype=np.uint64), 4, amd.RandomDelay.new(10.0, 4.0)).loop_until(t)
    for node in range(4):
        img = ora.save_node(node)
        res._sim.load_node(0, node, img, 1 << 40)
        assert res.save_node(0, node) == img
