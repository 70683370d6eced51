"""Node-level interface: the reference's trait surface (bft-lib/src/interfaces.rs:12-86) for single nodes, without
the event loop.

CPU part: the record-store / pacemaker scenarios of the reference's unit tests
(librabft-v2/src/unit_tests/record_store_tests.rs) replayed on the ORACLE through update_node /
create_notification / handle_notification, asserting what those tests assert (one vote per author, QC at quorum,
timeouts -> TC -> next round, no commit on non-contiguous rounds, 3-chain commit).  The reference builds records by
hand in a SharedRecordStore fixture; here they are produced by the nodes themselves, which pins the same rules.

GPU part (-m gpu): the identical scripted sessions on the device (lbft_node_* through the C ABI) must return the same
NodeUpdateActions and the same node views as the oracle after every single call.
"""
import numpy as np
import pytest


class OracleDriver:
    def __init__(self, oracle, n, **kw):
        self.sim = oracle.OracleSim(oracle.make_config(num_nodes=n, math_mode=1, **kw), 1)
        self.n = n

    def update(self, node, clock):
        return self.sim.node_update(node, clock)

    def notify(self, sender):
        return self.sim.node_create_notification(sender)

    def deliver(self, receiver, notification):
        return self.sim.node_handle_notification(receiver, notification)

    def view(self, node):
        return self.sim.node_view(node)

    def release(self, notification):
        pass

    # DataSyncNode::create_request / handle_request / handle_response
    def request(self, node):
        return self.sim.node_create_request(node)

    def respond(self, node, request):
        return self.sim.node_handle_request(node, request)

    def absorb(self, node, response, clock):
        self.sim.node_handle_response(node, response, clock)


class DeviceDriver:
    def __init__(self, amd, n, **kw):
        nc = amd.NodeConfig(kw.get("target_commit_interval", 100000), kw.get("delta", 20), kw.get("gamma", 2.0), kw.get("lambda_", 0.5))
        self.sim = amd.BatchSimulator.new(np.array([1, 2], dtype=np.uint64), n, amd.RandomDelay.new(10.0, 4.0), nc,
                                          commands_per_epoch=kw.get("commands_per_epoch", 30000),
                                          voting_rights=kw.get("voting_rights"), snapshot_capacity=64, quirks=kw.get("quirks", 0))
        self.nodes = self.sim.manual(100000)[1]  # instance 1 (instance 0 stays untouched)
        self.n = n

    def update(self, node, clock):
        return self.nodes[node].update_node(clock)

    def notify(self, sender):
        return self.nodes[sender].create_notification()

    def deliver(self, receiver, notification):
        return self.nodes[receiver].handle_notification(notification)

    def view(self, node):
        return self.nodes[node].view()

    def release(self, notification):
        self.sim.release_notification(1, notification)

    def request(self, node):
        return self.nodes[node].create_request()

    def respond(self, node, request):
        return self.nodes[node].handle_request(request)

    def absorb(self, node, response, clock):
        self.nodes[node].handle_response(response, clock)


def full_exchange(d, clock, trace, rounds=1, members=None):
    """Every node updates at `clock`; each node whose actions ask for it sends its notification to the named
    receivers (broadcast = everybody else) and those update again -- repeated until nothing is sent.
    `members`: only these nodes take part (the others neither run nor receive anything)."""
    members = list(range(d.n)) if members is None else list(members)
    pending = list(members)
    for _ in range(200):
        if not pending:
            break
        nxt = []
        for node in pending:
            a = d.update(node, clock)
            trace.append(("update", node, clock, a, d.view(node)))
            receivers = [r for r in members if r != node] if a["should_broadcast"] else [r for r in a["should_send"] if r != node and r in members]
            if receivers:
                note = d.notify(node)
                for r in receivers:
                    sync = d.deliver(r, note)
                    trace.append(("deliver", node, r, sync, d.view(r)))
                    if r not in nxt:
                        nxt.append(r)
                d.release(note)
            if a["next_scheduled_update"] <= clock and node not in nxt:
                nxt.append(node)
        pending = nxt
    return trace


def scenario_healthy_rounds(d):
    """Rounds advance through proposals, votes and QCs; commits follow the 3-chain rule."""
    trace = []
    for clock in range(0, 12):
        full_exchange(d, clock, trace)
    return trace


def scenario_timeouts(d, finish=True):
    """Nobody hears the leader: every node times out, timeouts are exchanged, a TC forms and the round advances."""
    trace = []
    for node in range(d.n):                       # enter round 1, but deliver nothing
        trace.append(("update", node, 0, d.update(node, 0), d.view(node)))
    v0 = d.view(0)
    deadline = 10 ** 6
    for node in range(d.n):
        a = d.update(node, 1)
        trace.append(("update", node, 1, a, d.view(node)))
        deadline = min(deadline, a["next_scheduled_update"])
    notes = []
    for node in range(d.n):                       # at the deadline every node creates its timeout and broadcasts
        a = d.update(node, deadline)
        trace.append(("timeout", node, deadline, a, d.view(node)))
        notes.append(d.notify(node))
    for sender in range(d.n):
        for r in range(d.n):
            if r != sender:
                trace.append(("deliver", sender, r, d.deliver(r, notes[sender]), d.view(r)))
    for note in notes:
        d.release(note)
    for node in range(d.n if finish else 0):      # (the new leader's proposal is not delivered)
        trace.append(("update", node, deadline + 1, d.update(node, deadline + 1), d.view(node)))
    return trace, v0, deadline


def test_oracle_healthy_rounds_commit_rule(oracle):
    # record_store_tests.rs:148-165 (QC at quorum), :236-292 (3-chain commit), :123-146 (one vote per author)
    d = OracleDriver(oracle, 4)
    trace = scenario_healthy_rounds(d)
    views = [d.view(n) for n in range(4)]
    assert all(v["highest_quorum_certificate_round"] >= 4 for v in views)
    # 3-chain: with contiguous rounds r, r+1, r+2 certified, round r commits: committed = hqc - 2
    assert all(v["highest_committed_round"] == v["highest_quorum_certificate_round"] - 2 for v in views)
    assert all(v["commit_count"] == v["highest_committed_round"] for v in views)
    # votes held for the current round never exceed one per author
    assert all(t[4]["num_current_votes"] <= 4 for t in trace)
    # a QC appears exactly when the leader holds a quorum (3 of 4) of votes: election leaves "ongoing"
    won = [t for t in trace if t[0] == "deliver" and t[4]["election"] != 0]
    assert won and all(t[4]["num_current_votes"] >= 3 for t in won)
    # locked round / latest voted round follow node.rs:256-276
    assert all(v["locked_round"] <= v["latest_voted_round"] for v in views)


def test_oracle_timeouts_form_tc_and_advance_round(oracle):
    # record_store_tests.rs:167-217: timeouts accumulate weight, the TC forms at quorum and moves to round + 1
    d = OracleDriver(oracle, 4)
    trace, v0, deadline = scenario_timeouts(d)
    assert v0["active_round"] == 1 and deadline == 20  # delta * 1^gamma
    for node in range(4):
        v = d.view(node)
        assert v["highest_timeout_certificate_round"] == 1 and v["has_timeout_certificate"] == 1
        assert v["current_round"] == 2 and v["active_round"] == 2
        assert v["highest_quorum_certificate_round"] == 0 and v["highest_committed_round"] == 0  # no QC => no commit
        assert v["num_current_timeouts"] == 0  # cleared on the round change (record_store.rs:207-219)
    # before the third timeout arrived no node had a TC
    partial = [t for t in trace if t[0] == "deliver" and t[4]["num_current_timeouts"] == 2]
    assert partial and all(t[4]["has_timeout_certificate"] == 0 for t in partial)


def scenario_timeout_then_healthy(d):
    trace, _, deadline = scenario_timeouts(d, finish=False)
    for clock in range(deadline + 1, deadline + 5):
        full_exchange(d, clock, trace)
    return trace


def test_oracle_commit_needs_three_contiguous_certified_rounds(oracle):
    # record_store_tests.rs:219-234 / :236-292: round 1 times out (never certified), so the first commit is round 2
    # and it happens only once rounds 2, 3 and 4 are certified; highest_committed_round always trails the QC by 2.
    d = OracleDriver(oracle, 4)
    trace = scenario_timeout_then_healthy(d)
    views = [t[4] for t in trace]
    assert max(v["highest_quorum_certificate_round"] for v in views) >= 6
    for v in views:
        if v["highest_quorum_certificate_round"] < 4:
            assert v["highest_committed_round"] == 0 and v["commit_count"] == 0
        else:
            assert v["highest_committed_round"] == v["highest_quorum_certificate_round"] - 2
            # round 1 produced no block to commit; the ledger follows on the node's next update_node (node.rs:313-350)
            assert v["commit_count"] <= v["highest_committed_round"] - 1
    for node in range(4):
        d.update(node, 10 ** 4)
        v = d.view(node)
        assert v["commit_count"] == v["highest_committed_round"] - 1


def scenario_catch_up(d, clocks=400):
    """Nodes 0-2 (a quorum of 4) run on their own; node 3 hears nothing.  Then node 3 asks node 0 for what it lacks
    (create_request -> handle_request -> handle_response, data_sync.rs:66-71,183-240) and catches up: the other half of the
    DataSyncNode trait.  Rounds led by node 3 end in timeouts of the others."""
    trace = []
    for clock in range(0, clocks):
        full_exchange(d, clock, trace, members=[0, 1, 2])
    behind = d.view(3)
    trace.append(("behind", 3, behind))
    req = d.request(3)
    resp = d.respond(0, req)
    d.absorb(3, resp, clocks)
    trace.append(("absorbed", 3, d.view(3)))
    trace.append(("update", 3, clocks, d.update(3, clocks), d.view(3)))
    d.release(req)
    d.release(resp)
    # a second exchange brings nothing new
    req = d.request(3)
    resp = d.respond(1, req)
    d.absorb(3, resp, clocks)
    trace.append(("absorbed again", 3, d.view(3)))
    d.release(req)
    d.release(resp)
    for clock in range(clocks, clocks + 4):  # and node 3 takes part from now on
        full_exchange(d, clock, trace)
    return trace


def test_oracle_lagging_node_catches_up_through_request_and_response(oracle):
    d = OracleDriver(oracle, 4, quirks=1)
    trace = scenario_catch_up(d)
    behind = [t for t in trace if t[0] == "behind"][0][2]
    caught = [t for t in trace if t[0] == "absorbed"][0][2]
    ahead = d.view(0)
    assert behind["highest_quorum_certificate_round"] == 0 and behind["commit_count"] == 0
    assert caught["highest_quorum_certificate_round"] >= 4 and caught["highest_committed_round"] >= 2
    again = [t for t in trace if t[0] == "absorbed again"][0][2]
    updated = [t for t in trace if t[0] == "update" and t[1] == 3][0][4]
    assert again == updated  # nothing new in the second response
    assert d.view(3)["commit_count"] >= ahead["commit_count"] - 3


SESSIONS = {
    "catch_up_4": (4, dict(quirks=1), lambda d: scenario_catch_up(d)),
    "catch_up_4_across_epochs": (4, dict(quirks=3, commands_per_epoch=3), lambda d: scenario_catch_up(d)),
    "catch_up_7_weighted": (7, dict(quirks=1, voting_rights=[3, 1, 1, 2, 1, 1, 2]), lambda d: scenario_catch_up(d, 500)),
    "healthy_4": (4, {}, lambda d: scenario_healthy_rounds(d)),
    "healthy_7_weighted": (7, dict(voting_rights=[3, 1, 1, 2, 1, 1, 2]), lambda d: scenario_healthy_rounds(d)),
    "timeouts_4": (4, {}, lambda d: scenario_timeouts(d)[0]),
    "timeout_then_healthy_4": (4, {}, lambda d: scenario_timeout_then_healthy(d)),
    "timeouts_5_fast_pacemaker": (5, dict(delta=3, gamma=1.5, lambda_=0.25), lambda d: scenario_timeouts(d)[0]),
    "healthy_40_nodes": (40, {}, lambda d: full_exchange(d, 0, []) + full_exchange(d, 1, [])),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SESSIONS))
def test_device_node_level_equals_oracle(oracle, name):
    import librabft_simulator_amd as amd
    n, kw, script = SESSIONS[name]
    a = script(OracleDriver(oracle, n, **kw))
    dev = DeviceDriver(amd, n, **kw)
    b = script(dev)
    assert len(a) == len(b) and len(a) > 10
    for x, y in zip(a, b):
        assert x == y, (x, y)
    # the node-level session left instance 0 untouched and its results are readable through the batch calls
    res = dev.sim.manual_finalize()
    assert res.commit_counts[0].sum() == 0
    assert [int(c) for c in res.commit_counts[1]] == [dev.view(i)["commit_count"] for i in range(n)]


@pytest.mark.gpu
def test_device_request_response_in_reference_mode_are_self_answered_and_insert_nothing(oracle):
    """quirks bit 0 clear = the reference simulator's routing (simulator.rs:441-466, quirk Q1): the whole DataSyncNode trait is
    callable; a request is answered by its own requester and the response changes nothing -- on the device and in the oracle."""
    import librabft_simulator_amd as amd
    n = 4
    dev, ora = DeviceDriver(amd, n), OracleDriver(oracle, n)
    for d in (dev, ora):
        trace = []
        for clock in range(0, 400):
            full_exchange(d, clock, trace, members=[0, 1, 2])
    before = [dev.view(i) for i in range(n)]
    assert before == [ora.view(i) for i in range(n)] and before[0]["highest_quorum_certificate_round"] >= 2
    for node in (0, 3):  # an up-to-date node and the one that heard nothing
        req = dev.request(node)
        resp = dev.respond(node, req)          # self-answered
        dev.absorb(node, resp, 400)
        oreq = ora.request(node)
        ora.absorb(node, ora.respond(node, oreq), 400)
        assert dev.view(node) == before[node] == ora.view(node)
        dev.release(req); dev.release(resp)
    req = dev.request(3)
    with pytest.raises(amd.LbftError) as e:    # a peer's answer needs the payloads of quirks bit 0
        dev.respond(0, req)
    assert e.value.code == -3


@pytest.mark.gpu
def test_batched_node_calls_equal_single_calls():
    """lbft_node_calls: the trait calls of MANY instances in one launch == the same calls made one by one (one launch + sync each).
    Scenario per instance: every node's first update (the leader of round 1 proposes), the proposer's notification delivered to the
    others, their updates (votes) -- with a different proposer-visible time per instance so that the instances differ."""
    import time
    import librabft_simulator_amd as amd
    from librabft_simulator_amd import _lib
    m, n = 512, 4

    def drive(batched):
        sim = amd.BatchSimulator.new(np.arange(1, m + 1, dtype=np.uint64), n, amd.RandomDelay.new(10.0, 4.0))
        nodes = sim.manual(100000)
        trace, dt = [], 0.0

        def many(calls):
            nonlocal dt
            t0 = time.perf_counter()
            if batched:
                r = sim.node_calls(calls)
            else:
                r = []
                for op, inst, node, peer, handle, t in calls:
                    h = nodes[inst][node]
                    if op == _lib.CALL_UPDATE_NODE:
                        r.append({"actions": h.update_node(t), "handle": 0, "should_sync": False})
                    elif op == _lib.CALL_CREATE_NOTIFICATION:
                        r.append({"actions": None, "handle": h.create_notification()[1], "should_sync": False})
                    elif op == _lib.CALL_HANDLE_NOTIFICATION:
                        r.append({"actions": None, "handle": 0, "should_sync": h.handle_notification((peer, handle))})
            dt += time.perf_counter() - t0
            return r
        for node in range(n):  # first update of every node, instance-specific clocks
            res = many([(_lib.CALL_UPDATE_NODE, i, node, 0, 0, 1 + i % 7) for i in range(m)])
            trace.append([(r["actions"]["next_scheduled_update"], tuple(r["actions"]["should_send"]), r["actions"]["should_broadcast"]) for r in res])
        leader = [next((k for k in range(n) if trace[k][i][2]), 0) for i in range(m)]  # who broadcast (the proposer), per instance
        notes = many([(_lib.CALL_CREATE_NOTIFICATION, i, leader[i], 0, 0, 0) for i in range(m)])
        for d in range(1, n):
            res = many([(_lib.CALL_HANDLE_NOTIFICATION, i, (leader[i] + d) % n, leader[i], notes[i]["handle"], 0) for i in range(m)])
            trace.append([r["should_sync"] for r in res])
            res = many([(_lib.CALL_UPDATE_NODE, i, (leader[i] + d) % n, 0, 0, 9 + i % 5) for i in range(m)])
            trace.append([(r["actions"]["next_scheduled_update"], tuple(r["actions"]["should_send"]), r["actions"]["should_broadcast"]) for r in res])
        views = [[nodes[i][k].view() for k in range(n)] for i in range(0, m, 37)]
        return trace, views, dt

    t_b, v_b, dt_b = drive(True)
    t_s, v_s, dt_s = drive(False)
    assert t_b == t_s and v_b == v_s
    assert any(x[2] for x in t_b[0] + t_b[1] + t_b[2] + t_b[3])            # somebody proposed
    assert any(len(x[1]) for row in t_b[4:] for x in row if isinstance(x, tuple))  # votes were sent to the proposer
    print("node-level calls: %d instances, batched %.1f ms, one by one %.1f ms" % (m, dt_b * 1e3, dt_s * 1e3))
    assert dt_b < dt_s  # ~16 launches against ~16 * 512


@pytest.mark.gpu
def test_batched_node_calls_reject_two_calls_on_one_instance():
    import librabft_simulator_amd as amd
    from librabft_simulator_amd import _lib
    sim = amd.BatchSimulator.new(np.arange(1, 5, dtype=np.uint64), 4, amd.RandomDelay.new(10.0, 4.0))
    sim.manual(1000)
    with pytest.raises(amd.LbftError) as e:
        sim.node_calls([(_lib.CALL_UPDATE_NODE, 1, 0, 0, 0, 1), (_lib.CALL_UPDATE_NODE, 1, 2, 0, 0, 1)])
    assert e.value.code == -1


@pytest.mark.gpu
def test_counters_allgather_reduce_through_rccl():
    """lbft_batch_counters_allgather_reduce: the run's one collective, natively (ONE ncclAllGather on the batch's stream, reduced locally).  One GPU
    here, so the communicator has one rank: the call goes through librccl and must return the batch's own counters -- under its round-4 name too."""
    import ctypes
    import librabft_simulator_amd as amd
    try:
        rccl = ctypes.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("librccl not available")
    uid = (ctypes.c_char * 128)()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    u = UniqueId()
    ctypes.memmove(ctypes.byref(u), uid, 128)
    res = amd.BatchSimulator.new(np.arange(1, 513, dtype=np.uint64), 4, amd.RandomDelay.new(10.0, 4.0)).loop_until(500)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, u, 0) == 0
    try:
        agg = res.counters_allgather_reduce(comm.value)
        assert res.counters_allreduce(comm.value) == agg  # (the alias: lbft_batch_counters_allreduce)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
    own = res.counters
    for k in ("events", "rounds", "commits", "rng_draws", "events_scheduled", "faulted_instances", "timers_folded", "node_updates", "max_queue", "max_snapshots", "max_blocks"):
        assert agg[k] == own[k], k
