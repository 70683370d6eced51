"""Host-side mirror of the reference's simulator interface for the hot path, on top of the C ABI.

Reference surface mirrored here (names and argument meaning kept):
  * ``RandomDelay::new(mean, variance)``                       bft-lib/src/simulator.rs:98-107
  * ``GlobalTime(i64)``                                        bft-lib/src/simulator.rs:35-37
  * ``NodeConfig{target_commit_interval, delta, gamma, lambda}`` librabft-v2/src/node.rs:76-81
  * ``Simulator::new(rng_seed, num_nodes, network_delay, context_factory)`` and
    ``Simulator::loop_until(max_clock, csv_path) -> Vec<&Context>``  bft-lib/src/simulator.rs:200-250,380-475
  * ``SimulatedContext::committed_history()`` / ``last_committed_state()``
                                                               bft-lib/src/simulated_context.rs:98-100,194-196
``BatchSimulator`` is the batched form (many independent ``Simulator``s, one GPU lane each); it is
what the HIP path natively executes.  Everything computes on the GPU through liblbft_hip.so; there
is no CPU path in this package.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import COMMIT_DTYPE, RECORD_HASH_DTYPE, LbftActions, LbftConfig, LbftCounters, LbftError, LbftNodeView, check

Command = namedtuple("Command", ["proposer", "index"])  # simulated_context.rs:31-35
Author = int


class NodeTime(int):
    """bft-lib/src/base_types.rs NodeTime(i64): a node's local clock (global time minus its startup time)."""

    def __repr__(self):
        return "NodeTime(%d)" % int(self)


class State(int):
    """simulated_context.rs:28-29: State(u64)."""

    def __repr__(self):
        return "State(%d)" % int(self)


class GlobalTime(int):
    """bft-lib/src/simulator.rs:35-37; the conversions of :120-126 (a node's startup time: ``BatchResult.startup_times``)."""

    def __repr__(self):
        return "GlobalTime(%d)" % int(self)

    def to_node_time(self, startup_time):
        return NodeTime(int(self) - int(startup_time))

    @classmethod
    def from_node_time(cls, node_time, startup_time):
        return cls(int(node_time) + int(startup_time))

    def __add__(self, duration):  # GlobalTime + Duration (simulator.rs:90-96)
        return GlobalTime(int(self) + int(duration))


class Duration(int):
    """bft-lib/src/base_types.rs Duration(i64)."""


class RandomDelay:
    """bft-lib/src/simulator.rs:39-43,98-107.  ``RandomDelay.new(mean, variance)`` is the reference's
    log-normal delay; ``RandomDelay.uniform(lo, hi)`` is this framework's integer-uniform extension."""

    def __init__(self, mean=10.0, variance=4.0, model=0, lo=0, hi=0):
        self.mean, self.variance, self.model, self.lo, self.hi = float(mean), float(variance), int(model), int(lo), int(hi)

    @classmethod
    def new(cls, mean, variance):
        return cls(mean, variance, 0)

    @classmethod
    def uniform(cls, lo, hi):
        return cls(10.0, 4.0, 1, lo, hi)


class NodeConfig:
    """librabft-v2/src/node.rs:76-81 (defaults = CLI defaults, librabft-v2/src/main.rs:111-134)."""

    def __init__(self, target_commit_interval=100000, delta=20, gamma=2.0, lambda_=0.5):
        self.target_commit_interval = int(target_commit_interval)
        self.delta = int(delta)
        self.gamma = float(gamma)
        self.lambda_ = float(lambda_)


def make_config(num_nodes, network_delay, node_config, commands_per_epoch=30000, voting_rights=None,
                queue_capacity=0, snapshot_capacity=0, block_capacity=0, log_capacity=0, equivocate_every=0,
                drop_per_million=0, partition=None, quirks=0, rights_rotation=0):
    cfg = LbftConfig()
    cfg.num_nodes = num_nodes
    cfg.delay_model = network_delay.model
    cfg.mean = network_delay.mean
    cfg.variance = network_delay.variance
    cfg.uniform_lo = network_delay.lo
    cfg.uniform_hi = network_delay.hi
    cfg.commands_per_epoch = commands_per_epoch
    cfg.target_commit_interval = node_config.target_commit_interval
    cfg.delta = node_config.delta
    cfg.gamma = node_config.gamma
    cfg.lambda_ = node_config.lambda_
    cfg.quirks = int(quirks)
    cfg.equivocate_every = int(equivocate_every)
    cfg.drop_per_million = int(drop_per_million)
    cfg.rights_rotation = int(rights_rotation)  # extension: epoch e uses voting_rights[(i + e * rights_rotation) % n]
    if partition is not None:  # (size of the first side, start, end): nodes [0, size) are cut off during [start, end)
        cfg.partition_size, cfg.partition_start, cfg.partition_end = (int(v) for v in partition)
    cfg.queue_capacity = queue_capacity
    cfg.snapshot_capacity = snapshot_capacity
    cfg.block_capacity = block_capacity
    cfg.log_capacity = log_capacity
    if voting_rights is not None:
        arr = (C.c_uint64 * num_nodes)(*[int(w) for w in voting_rights])
        cfg._keepalive = arr
        cfg.voting_rights = C.cast(arr, C.POINTER(C.c_uint64))
    return cfg


def write_data_files(path, rows, messages, num_nodes):
    """DataWriter::write_to_file (bft-lib/src/data_writer.rs:62-97): ``round_switches.txt`` (header ``node i``, one
    row per round below the highest round reached, empty cells for rounds a node never entered) and
    ``number_of_messages.txt`` in directory ``path`` (created if missing, as DataWriter::new does)."""
    import os
    if not os.path.exists(path):
        os.mkdir(path)
    with open(os.path.join(path, "round_switches.txt"), "w") as f:
        f.write(",".join("node %d" % i for i in range(num_nodes)) + "\n")
        for row in rows:
            f.write(",".join("" if v is None else str(v) for v in row) + "\n")
    with open(os.path.join(path, "number_of_messages.txt"), "w") as f:
        f.write("%d\n" % messages)


class BatchResult:
    """Results of ``BatchSimulator.loop_until`` (lazy device read-back through the C ABI)."""

    def __init__(self, sim):
        self._sim = sim
        self._cache = {}
        self._gen = getattr(sim, "_state_generation", 0)

    def _get(self, key, fn):
        # what was read back belongs to the device state it was read from: BatchSimulator.load_node rewrites a node's rows (pacemaker round,
        # epoch, certificates ...) after the run, and a result object created before it must not serve the pre-load values
        gen = getattr(self._sim, "_state_generation", 0)
        if gen != self._gen:
            self._cache.clear()
            self._gen = gen
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    @property
    def counters(self):
        def f():
            c = LbftCounters()
            check(_lib.lib().lbft_batch_counters(self._sim._h, C.byref(c)))
            return c.as_dict()
        return self._get("counters", f)

    def counters_allgather_reduce(self, nccl_comm):
        """The run's one collective, natively: ONE RCCL ncclAllGather of the fourteen counter words per rank over the caller's communicator
        (an ncclComm_t as an integer / ctypes pointer; every rank calls it), reduced locally -- sums and high-water marks.  Returns the
        aggregate as a dict."""
        c = LbftCounters()
        check(_lib.lib().lbft_batch_counters_allgather_reduce(self._sim._h, C.c_void_p(int(nccl_comm)), C.byref(c)))
        return c.as_dict()

    counters_allreduce = counters_allgather_reduce  # (the name through round 4)

    def _node_array(self, name, dtype):
        def f():
            out = np.zeros((self._sim.num_instances, self._sim.num_nodes), dtype=dtype)
            check(getattr(_lib.lib(), "lbft_batch_" + name)(self._sim._h, out.ctypes.data))
            return out
        return self._get(name, f)

    @property
    def commit_counts(self):
        """contexts.iter().map(|c| c.committed_history().len())  [instance, node]"""
        return self._node_array("commit_counts", np.uint32)

    @property
    def active_rounds(self):
        return self._node_array("active_rounds", np.uint64)

    @property
    def last_committed_states(self):
        return self._node_array("last_committed_states", np.uint64)

    @property
    def startup_times(self):
        return self._node_array("startup_times", np.int64)

    @property
    def epochs(self):
        return self._node_array("epochs", np.uint64)

    @property
    def faults(self):
        def f():
            out = np.zeros(self._sim.num_instances, dtype=np.uint32)
            check(_lib.lib().lbft_batch_faults(self._sim._h, out.ctypes.data))
            return out
        return self._get("faults", f)

    def committed_histories(self, cap_per_node=None):
        """[instance, node, k] structured array (proposer, index, time); entries past the node's
        commit count are zero."""
        if cap_per_node is None:
            cap_per_node = int(self.commit_counts.max()) if self._sim.num_instances else 0
        cap_per_node = max(int(cap_per_node), 1)
        out = np.zeros((self._sim.num_instances, self._sim.num_nodes, cap_per_node), dtype=COMMIT_DTYPE)
        check(_lib.lib().lbft_batch_committed_histories(self._sim._h, out.ctypes.data, cap_per_node))
        return out

    def committed_history(self, instance, node):
        n = int(self.commit_counts[instance, node])
        out = np.zeros(max(n, 1), dtype=COMMIT_DTYPE)
        ln = C.c_size_t()
        check(_lib.lib().lbft_batch_committed_history(self._sim._h, instance, node, out.ctypes.data, n, C.byref(ln)))
        return out[:n]

    def committed_record_hashes(self, instance, node):
        """(block_hash, state, qc_hash, num_votes, flags) of the Block_ / QuorumCertificate_ records behind
        committed_history(instance, node), hashed like the reference's SmrContext::hash (BCS + SipHash-1-3)."""
        n = int(self.commit_counts[instance, node])
        out = np.zeros(max(n, 1), dtype=RECORD_HASH_DTYPE)
        ln = C.c_size_t()
        check(_lib.lib().lbft_batch_committed_record_hashes(self._sim._h, instance, node, out.ctypes.data, n, C.byref(ln)))
        return out[:n]

    def save_node(self, instance, node):
        """ConsensusNode::save_node (librabft-v2/src/node.rs:233-238): bincode image of the node's NodeState (bytes), HashMaps in
        ascending key order; what the reference's load_node (node.rs:211-231) deserialises."""
        return self._sim.save_node(instance, node)

    def round_switches(self, instance=0, cap_rounds=None):
        """DataWriter output of one instance (bft-lib/src/data_writer.rs): (rows, number_of_messages) where
        rows[round][node] is the GlobalTime at which the node was first seen in that round or None (empty cell)."""
        cap = int(cap_rounds or 1 << 16)
        n = self._sim.num_nodes
        mr, msgs = C.c_uint64(), C.c_uint64()
        out = np.full((min(cap, 1 << 16), n), np.iinfo(np.int64).min, dtype=np.int64)
        check(_lib.lib().lbft_batch_round_switches(self._sim._h, int(instance), out.ctypes.data, out.shape[0], C.byref(mr), C.byref(msgs)))
        lo = np.iinfo(np.int64).min
        rows = [[None if v == lo else int(v) for v in out[r]] for r in range(min(int(mr.value), out.shape[0]))]
        return rows, int(msgs.value)

    def contexts(self, instance=0):
        """The ``Vec<&Context>`` that ``Simulator::loop_until`` returns, for one instance."""
        return [SimulatedContextView(self, instance, n) for n in range(self._sim.num_nodes)]


class SimulatedContextView:
    """Read-only view with the accessors the reference's callers use (main.rs:47-53,
    tests/simulated_run.rs:45-94)."""

    def __init__(self, result, instance, node):
        self._r, self._i, self._n = result, instance, node

    def committed_history(self):
        h = self._r.committed_history(self._i, self._n)
        return [(Command(int(e["proposer"]), int(e["index"])), int(e["time"])) for e in h]

    def last_committed_state(self):
        out = C.c_uint64()
        check(_lib.lib().lbft_batch_last_committed_state(self._r._sim._h, self._i, self._n, C.byref(out)))
        return State(int(out.value))


class BatchSimulator:
    """Many independent ``Simulator``s (bft-lib/src/simulator.rs:26-33) advanced in lockstep on one GPU."""

    def __init__(self, rng_seeds, num_nodes, network_delay, node_config=None, commands_per_epoch=30000,
                 voting_rights=None, device=0, queue_capacity=0, snapshot_capacity=0, block_capacity=0,
                 log_capacity=0, max_steps_per_launch=0, lanes_per_wavefront=0, lds_queue_slots=-1, equivocate_every=0, drop_per_million=0, partition=None,
                 calendar_queue=True, quirks=0, rights_rotation=0, keep_retired_stores=False):
        seeds = np.ascontiguousarray(rng_seeds, dtype=np.uint64)
        self.seeds = seeds
        self.num_instances = int(seeds.shape[0])
        self.num_nodes = int(num_nodes)
        self.device = int(device)
        self._cfg = make_config(num_nodes, network_delay, node_config or NodeConfig(), commands_per_epoch, voting_rights,
                                queue_capacity, snapshot_capacity, block_capacity, log_capacity, equivocate_every, drop_per_million, partition, quirks,
                                rights_rotation)
        self._h = C.c_void_p()
        check(_lib.lib().lbft_batch_create(C.byref(self._cfg), seeds.ctypes.data, self.num_instances, self.device,
                                           C.byref(self._h)))
        if max_steps_per_launch:
            check(_lib.lib().lbft_batch_set_max_steps(self._h, max_steps_per_launch))
        if lanes_per_wavefront:
            check(_lib.lib().lbft_batch_set_lanes_per_wavefront(self._h, lanes_per_wavefront))
        if not calendar_queue:
            check(_lib.lib().lbft_batch_set_calendar_queue(self._h, 0))
        if lds_queue_slots != -1:
            check(_lib.lib().lbft_batch_set_lds_queue_slots(self._h, lds_queue_slots))
        if keep_retired_stores:  # past_record_stores (node.rs:43) in full: save_node then also serves nodes that have changed epoch
            check(_lib.lib().lbft_batch_keep_retired_stores(self._h, 1))

    @classmethod
    def new(cls, rng_seeds, num_nodes, network_delay, node_config=None, **kw):
        return cls(rng_seeds, num_nodes, network_delay, node_config, **kw)

    def loop_until(self, max_clock, csv_path=None, allow_faults=False, round_trace=None):
        """Simulator::loop_until for every instance.  ``csv_path`` is the reference's ``Option<String>`` data-files
        directory (bft-lib/src/simulator.rs:380-381, data_writer.rs): when given, the round-switch trace is recorded on
        the device and ``round_switches.txt`` / ``number_of_messages.txt`` of instance 0 are written there in the
        reference's CSV format.  ``round_trace=N`` only records (N rounds per node) for ``BatchResult.round_switches``."""
        # The trace rows are allocated for EVERY instance of the batch (num_nodes x round_trace words each).  Left to this method, the
        # capacity starts from a realistic bound -- a round of >= 3 nodes takes two network hops, max_clock / 5 rounds is generous for
        # delays of mean >= 5 -- and only a run that overflows it (F_TRACE_OVERFLOW: tiny networks, near-zero delays) is repeated with
        # the worst case, one round per time unit (at most 65 536 rounds: longer horizons need an explicit round_trace).
        auto = csv_path is not None and round_trace is None
        worst = min(int(max_clock) + 64, 1 << 16)
        if auto:
            round_trace = min(int(max_clock) // 5 + 64, worst)
        if round_trace:
            check(_lib.lib().lbft_batch_enable_round_trace(self._h, int(round_trace)))
        self._mutated()
        rc = check(_lib.lib().lbft_batch_run_until(self._h, int(max_clock)), allow_fault=allow_faults or auto)
        if auto and rc == _lib.LBFT_ERR_FAULT:
            if round_trace < worst and (BatchResult(self).faults & _lib.LBFT_FAULT_TRACE_OVERFLOW).any():
                import warnings
                warnings.warn("loop_until(csv_path=...): a node passed the automatic round-trace capacity (%d rounds); the whole batch is run "
                              "AGAIN with the worst case (%d rounds per node = %d trace words per instance, every instance of the batch). "
                              "Pass round_trace=N to choose the capacity yourself." % (round_trace, worst, self.num_nodes * (worst + 1)))
                self.reset()
                check(_lib.lib().lbft_batch_enable_round_trace(self._h, worst))
                rc = check(_lib.lib().lbft_batch_run_until(self._h, int(max_clock)), allow_fault=True)
            if rc == _lib.LBFT_ERR_FAULT and not allow_faults:
                # (the text comes from the instances' fault words, not from lbft_last_error: that may be another call's)
                f = BatchResult(self).faults
                bits = int(np.bitwise_or.reduce(f)) if len(f) else 0
                names = [n for b, n in sorted(_lib.FAULT_NAMES.items()) if bits & b] or ["fault bits 0x%x" % bits]
                raise LbftError(rc, "%d instance(s) faulted: %s" % (int((f != 0).sum()), ", ".join(names)))
        res = BatchResult(self)
        if csv_path is not None:
            write_data_files(csv_path, *res.round_switches(0), self.num_nodes)
        return res

    def run_steps(self, max_clock, steps, allow_faults=False):
        """At most ``steps`` events per instance (first call = Simulator::new).  Returns (unfinished instances, BatchResult
        or None); the result is available once nothing is left to process."""
        left = C.c_uint64()
        self._mutated()
        check(_lib.lib().lbft_batch_run_steps(self._h, int(max_clock), int(steps), C.byref(left)), allow_fault=allow_faults)
        return int(left.value), (BatchResult(self) if left.value == 0 else None)

    def save_checkpoint(self, path):
        """Whole-batch checkpoint (the reference's save_node, node.rs:233-238, at batch granularity)."""
        nbytes = _lib.lib().lbft_batch_checkpoint_bytes(self._h)
        buf = np.zeros(nbytes, dtype=np.uint8)
        check(_lib.lib().lbft_batch_checkpoint_save(self._h, buf.ctypes.data, nbytes))
        buf.tofile(path)
        return nbytes

    def load_checkpoint(self, path):
        """load_node (node.rs:211-231) at batch granularity: into a batch created with the same configuration."""
        buf = np.fromfile(path, dtype=np.uint8)
        self._mutated()
        check(_lib.lib().lbft_batch_checkpoint_load(self._h, buf.ctypes.data, buf.size))

    def manual(self, max_clock=1000):
        """Node-level mode: initial node states only (NodeState::make_initial_state), no event loop.  Returns
        ``nodes[instance][author]`` -> NodeHandle."""
        self._mutated()
        check(_lib.lib().lbft_batch_manual_begin(self._h, int(max_clock)))
        return [[NodeHandle(self, i, n) for n in range(self.num_nodes)] for i in range(self.num_instances)]

    def node_calls(self, calls):
        """Many trait calls in ONE launch (lbft_node_calls): `calls` = iterable of (op, instance, node, peer, handle, node_time) with op
        one of _lib.CALL_*; every call on another instance.  Returns a list of dicts (actions / handle / should_sync per call)."""
        calls = list(calls)
        arr = (_lib.LbftNodeCall * len(calls))()
        for k, (op, inst, node, peer, handle, t) in enumerate(calls):
            arr[k] = _lib.LbftNodeCall(int(op), int(inst), int(node), int(peer), int(handle), 0, int(t))
        res = (_lib.LbftNodeResult * len(calls))()
        self._mutated()
        check(_lib.lib().lbft_node_calls(self._h, arr, len(calls), res))
        return [{"actions": r.actions.as_dict(), "handle": int(r.handle), "should_sync": bool(r.should_sync), "status": int(r.status)} for r in res]

    def release_notification(self, instance, notification):
        check(_lib.lib().lbft_node_release_notification(self._h, int(instance), notification[1]))

    def manual_finalize(self):
        """Makes commit counts / histories / States of a node-level session readable (BatchResult)."""
        self._mutated()
        check(_lib.lib().lbft_batch_manual_finalize(self._h), allow_fault=True)
        return BatchResult(self)

    def save_node(self, instance, node):
        ln = C.c_size_t()
        check(_lib.lib().lbft_batch_save_node(self._h, int(instance), int(node), None, 0, C.byref(ln)))
        buf = np.zeros(ln.value, dtype=np.uint8)
        check(_lib.lib().lbft_batch_save_node(self._h, int(instance), int(node), buf.ctypes.data, ln.value, C.byref(ln)))
        return buf.tobytes()

    def load_node(self, instance, node, image, node_time):
        """ConsensusNode::load_node (node.rs:211-231): the bincode NodeState ``image`` (bytes) into the device-resident node; raises
        LbftError (LBFT_ERR_STATE) for "saved state from the future" (``node_time`` before the image's own times), LBFT_ERR_UNSUPPORTED
        for an image naming records this instance's block pool does not hold.  The node is untouched when it raises."""
        buf = np.frombuffer(bytes(image), dtype=np.uint8)
        check(_lib.lib().lbft_batch_load_node(self._h, int(instance), int(node), buf.ctypes.data, len(buf), int(node_time)))
        self._mutated()

    def _mutated(self):
        """Every call that changes device state -- a run, a reset, a checkpoint / node load, a node-level call -- invalidates what BatchResult
        objects of this simulator read back before (they re-read on their next access; round-5 advisor: only load_node used to do this)."""
        self._state_generation = getattr(self, "_state_generation", 0) + 1

    def reset(self):
        self._mutated()
        check(_lib.lib().lbft_batch_reset(self._h))

    def stream_handle(self):
        return _lib.lib().lbft_batch_stream(self._h)

    def last_run_ms(self):
        a, b = C.c_float(), C.c_float()
        check(_lib.lib().lbft_batch_last_run_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def phase_cycles(self):
        """Diagnostic builds only (LBFT_PHASE_TIMERS): shader cycles per phase of the event loop."""
        out = np.zeros(32, dtype=np.uint64)
        check(_lib.lib().lbft_batch_phase_cycles(self._h, out.ctypes.data))
        return out

    def layout(self):
        """Struct sizes behind the roofline arithmetic (see include/lbft.h lbft_batch_layout)."""
        out = np.zeros(8, dtype=np.uint32)
        check(_lib.lib().lbft_batch_layout(self._h, out.ctypes.data))
        keys = ("node_bytes", "event_bytes", "snapshot_bytes", "block_bytes", "instance_bytes", "lds_queue_slots",
                "lanes_per_wavefront", "kernel_class")
        return dict(zip(keys, (int(v) for v in out)))

    def device_bytes(self):
        return int(_lib.lib().lbft_batch_device_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().lbft_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NodeHandle:
    """One node of one instance behind the reference's trait surface (bft-lib/src/interfaces.rs): ``ConsensusNode::
    update_node`` and ``DataSyncNode::{create_notification, handle_notification}``, each executed on the GPU by
    ``lbft_node_*``.  Obtained from ``BatchSimulator.manual(...)``; the caller owns time and message delivery (the role
    of ``Simulator::loop_until`` or of bft-driver's ``CoreDriver``)."""

    def __init__(self, sim, instance, node):
        self._sim, self.instance, self.author = sim, int(instance), int(node)

    def update_node(self, clock):
        """ConsensusNode::update_node(clock: NodeTime) -> NodeUpdateActions (librabft-v2/src/node.rs:240-304)."""
        a = LbftActions()
        self._sim._mutated()
        check(_lib.lib().lbft_node_update(self._sim._h, self.instance, self.author, int(clock), C.byref(a)))
        return a.as_dict()

    def create_notification(self):
        """DataSyncNode::create_notification (librabft-v2/src/data_sync.rs:82-111) -> opaque handle."""
        h = C.c_uint32()
        check(_lib.lib().lbft_node_create_notification(self._sim._h, self.instance, self.author, C.byref(h)))
        return (self.author, int(h.value))

    def handle_notification(self, notification):
        """DataSyncNode::handle_notification (data_sync.rs:113-177); True when the reference returns Some(request)."""
        sender, handle = notification
        sync = C.c_uint32()
        self._sim._mutated()
        check(_lib.lib().lbft_node_handle_notification(self._sim._h, self.instance, self.author, sender, handle, C.byref(sync)))
        return bool(sync.value)

    def create_request(self):
        """DataSyncNode::create_request (data_sync.rs:66-71,179-181) -> opaque handle (batches with quirks bit 0)."""
        h = C.c_uint32()
        check(_lib.lib().lbft_node_create_request(self._sim._h, self.instance, self.author, C.byref(h)))
        return (self.author, int(h.value))

    def handle_request(self, request):
        """DataSyncNode::handle_request (data_sync.rs:183-207): the records the requester lacks -> opaque response handle."""
        _, handle = request
        h = C.c_uint32()
        self._sim._mutated()
        check(_lib.lib().lbft_node_handle_request(self._sim._h, self.instance, self.author, handle, C.byref(h)))
        return (self.author, int(h.value))

    def handle_response(self, response, clock):
        """DataSyncNode::handle_response(response, clock) (data_sync.rs:209-240)."""
        peer, handle = response
        self._sim._mutated()
        check(_lib.lib().lbft_node_handle_response(self._sim._h, self.instance, self.author, peer, handle, int(clock)))

    def release(self, message):
        """Drops a notification / request / response handle."""
        check(_lib.lib().lbft_node_release_notification(self._sim._h, self.instance, message[1]))

    def save_node(self):
        """ConsensusNode::save_node (node.rs:233-238) -> the bincode image of this node's NodeState."""
        return self._sim.save_node(self.instance, self.author)

    def load_node(self, image, clock):
        """ConsensusNode::load_node (node.rs:211-231): restore this node's NodeState from a save_node image; `clock` = the node's time
        (the reference refuses "saved state from the future")."""
        self._sim.load_node(self.instance, self.author, image, clock)

    def view(self):
        v = LbftNodeView()
        check(_lib.lib().lbft_node_view_get(self._sim._h, self.instance, self.author, C.byref(v)))
        return v.as_dict()


class Simulator:
    """``Simulator::new(rng_seed, num_nodes, network_delay, context_factory)`` for one network.

    The reference's ``context_factory`` closure (librabft-v2/src/main.rs:23-34) only carries
    ``commands_per_epoch`` and the ``NodeConfig``; pass them directly."""

    def __init__(self, rng_seed, num_nodes, network_delay, node_config=None, commands_per_epoch=30000, **kw):
        self._batch = BatchSimulator([rng_seed], num_nodes, network_delay, node_config, commands_per_epoch, **kw)

    @classmethod
    def new(cls, rng_seed, num_nodes, network_delay, node_config=None, commands_per_epoch=30000, **kw):
        return cls(rng_seed, num_nodes, network_delay, node_config, commands_per_epoch, **kw)

    def loop_until(self, max_clock, csv_path=None):
        self.result = self._batch.loop_until(max_clock, csv_path)
        return self.result.contexts(0)
