"""ctypes binding of the CPU oracle (oracle/liblbft_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (librabft_simulator_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblbft_oracle.so")


class OracleConfig(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_uint32),
        ("delay_model", C.c_uint32),
        ("mean", C.c_double),
        ("variance", C.c_double),
        ("uniform_lo", C.c_int64),
        ("uniform_hi", C.c_int64),
        ("commands_per_epoch", C.c_uint64),
        ("target_commit_interval", C.c_int64),
        ("delta", C.c_int64),
        ("gamma", C.c_double),
        ("lambda_", C.c_double),
        ("quirks", C.c_uint32),
        ("equivocate_every", C.c_uint32),
        ("drop_per_million", C.c_uint32),
        ("partition_size", C.c_uint32),
        ("partition_start", C.c_int64),
        ("partition_end", C.c_int64),
        ("math_mode", C.c_uint32),
        ("voting_rights", C.POINTER(C.c_uint64)),
        ("rights_rotation", C.c_uint32),
        ("reference_overheads", C.c_uint32),
    ]


class OracleCounters(C.Structure):
    _fields_ = [
        ("events", C.c_uint64 * 4),
        ("rng_draws", C.c_uint64),
        ("rounds", C.c_uint64),
        ("commits", C.c_uint64),
        ("response_inserts", C.c_uint64),
        ("max_queue", C.c_uint64),
        ("events_scheduled", C.c_uint64),
        ("saved_bytes", C.c_uint64),
    ]

    def as_dict(self):
        return {
            "events": list(self.events),
            "rng_draws": self.rng_draws,
            "rounds": self.rounds,
            "commits": self.commits,
            "response_inserts": self.response_inserts,
            "max_queue": self.max_queue,
            "events_scheduled": self.events_scheduled,
            "saved_bytes": self.saved_bytes,
        }


COMMIT_DTYPE = np.dtype([("proposer", "<u8"), ("index", "<u8"), ("time", "<i8")])
RECORD_HASH_DTYPE = np.dtype([("block_hash", "<u8"), ("state", "<u8"), ("qc_hash", "<u8"), ("has_qc", "<u4"), ("num_votes", "<u4")])

class OracleActions(C.Structure):
    _fields_ = [("next_scheduled_update", C.c_int64), ("should_send", C.c_uint64 * 2), ("should_broadcast", C.c_uint32),
                ("should_query_all", C.c_uint32)]

    def as_dict(self):
        bits = int(self.should_send[0]) | (int(self.should_send[1]) << 64)
        return {"next_scheduled_update": int(self.next_scheduled_update),
                "should_send": [a for a in range(128) if (bits >> a) & 1],
                "should_broadcast": bool(self.should_broadcast), "should_query_all": bool(self.should_query_all)}


class OracleNodeView(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("epoch_id", "current_round", "highest_quorum_certificate_round",
                                           "highest_timeout_certificate_round", "highest_committed_round", "active_round",
                                           "latest_voted_round", "locked_round", "commit_count")] + \
               [(n, C.c_uint32) for n in ("active_leader", "election", "num_current_timeouts", "num_current_votes",
                                           "has_proposed_block", "has_timeout_certificate")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def build(force=False):
    """Compile the oracle with g++ (Makefile in this directory)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.lbft_oracle_create.argtypes = [C.POINTER(OracleConfig), C.c_uint64, C.POINTER(vp)]
        L.lbft_oracle_create.restype = C.c_int
        L.lbft_oracle_run_until.argtypes = [vp, C.c_int64]
        L.lbft_oracle_run_until.restype = C.c_int
        L.lbft_oracle_destroy.argtypes = [vp]
        L.lbft_oracle_destroy.restype = None
        L.lbft_oracle_commit_count.argtypes = [vp, C.c_uint32]
        L.lbft_oracle_commit_count.restype = C.c_size_t
        L.lbft_oracle_committed_history.argtypes = [vp, C.c_uint32, vp, C.c_size_t]
        L.lbft_oracle_committed_history.restype = C.c_size_t
        L.lbft_oracle_committed_record_hashes.argtypes = [vp, C.c_uint32, vp, C.c_size_t]
        L.lbft_oracle_committed_record_hashes.restype = C.c_size_t
        for name in ("last_committed_state", "active_round", "epoch"):
            f = getattr(L, "lbft_oracle_" + name)
            f.argtypes = [vp, C.c_uint32]
            f.restype = C.c_uint64
        L.lbft_oracle_startup_time.argtypes = [vp, C.c_uint32]
        L.lbft_oracle_startup_time.restype = C.c_int64
        L.lbft_oracle_counters_get.argtypes = [vp, C.POINTER(OracleCounters)]
        L.lbft_oracle_counters_get.restype = None
        L.lbft_oracle_last_error.argtypes = [vp]
        L.lbft_oracle_last_error.restype = C.c_char_p
        L.lbft_oracle_run_batch.argtypes = [
            C.POINTER(OracleConfig), vp, C.c_size_t, C.c_int64, C.c_uint32, vp, vp, vp, vp, C.c_size_t,
            C.POINTER(OracleCounters)]
        L.lbft_oracle_run_batch.restype = C.c_int
        L.lbft_oracle_node_update.argtypes = [vp, C.c_uint32, C.c_int64, C.POINTER(OracleActions)]
        L.lbft_oracle_node_update.restype = C.c_int
        L.lbft_oracle_node_create_notification.argtypes = [vp, C.c_uint32]
        L.lbft_oracle_node_create_notification.restype = C.c_int
        L.lbft_oracle_node_handle_notification.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        L.lbft_oracle_node_handle_notification.restype = C.c_int
        L.lbft_oracle_node_create_request.argtypes = [vp, C.c_uint32]
        L.lbft_oracle_node_create_request.restype = C.c_int
        L.lbft_oracle_node_handle_request.argtypes = [vp, C.c_uint32, C.c_int]
        L.lbft_oracle_node_handle_request.restype = C.c_int
        L.lbft_oracle_node_handle_response.argtypes = [vp, C.c_uint32, C.c_int, C.c_int64]
        L.lbft_oracle_node_handle_response.restype = C.c_int
        L.lbft_oracle_node_view_get.argtypes = [vp, C.c_uint32, C.POINTER(OracleNodeView)]
        L.lbft_oracle_node_view_get.restype = C.c_int
        L.lbft_oracle_enable_data_writer.argtypes = [vp]
        L.lbft_oracle_enable_data_writer.restype = None
        L.lbft_oracle_round_switches.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint64)]
        L.lbft_oracle_round_switches.restype = C.c_uint64
        L.lbft_oracle_siphash13.argtypes = [C.c_char_p, C.c_size_t]
        L.lbft_oracle_siphash13.restype = C.c_uint64
        L.lbft_oracle_xoshiro_first.argtypes = [C.c_uint64, vp, C.c_size_t]
        L.lbft_oracle_xoshiro_first.restype = None
        L.lbft_oracle_pick_author.argtypes = [vp, C.c_size_t, C.c_uint64]
        L.lbft_oracle_pick_author.restype = C.c_uint64
        L.lbft_oracle_leader.argtypes = [vp, C.c_size_t, C.c_uint64]
        L.lbft_oracle_leader.restype = C.c_uint64
        L.lbft_oracle_quorum_threshold.argtypes = [vp, C.c_size_t]
        L.lbft_oracle_quorum_threshold.restype = C.c_uint64
        L.lbft_oracle_sample_delays.argtypes = [C.POINTER(OracleConfig), C.c_uint64, vp, C.c_size_t]
        L.lbft_oracle_sample_delays.restype = None
        L.lbft_oracle_shuffle.argtypes = [C.c_uint64, vp, C.c_size_t]
        L.lbft_oracle_shuffle.restype = None
        L.lbft_oracle_exp_strict.argtypes = [C.c_double]
        L.lbft_oracle_exp_strict.restype = C.c_double
        L.lbft_oracle_log_strict.argtypes = [C.c_double]
        L.lbft_oracle_log_strict.restype = C.c_double
        L.lbft_oracle_save_node.argtypes = [vp, C.c_uint32, vp, C.c_size_t]
        L.lbft_oracle_save_node.restype = C.c_size_t
        L.lbft_oracle_exp_mismatches.argtypes = [vp, C.c_size_t]
        L.lbft_oracle_exp_mismatches.restype = C.c_size_t
        L.lbft_oracle_log_mismatches.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.lbft_oracle_log_mismatches.restype = C.c_size_t
        _lib = L
    return _lib


def make_config(num_nodes=3, mean=10.0, variance=4.0, delay_model=0, uniform_lo=5, uniform_hi=15,
                commands_per_epoch=30000, target_commit_interval=100000, delta=20, gamma=2.0,
                lambda_=0.5, quirks=0, math_mode=0, voting_rights=None, equivocate_every=0, drop_per_million=0, partition_size=0,
                partition_start=0, partition_end=0, rights_rotation=0, reference_overheads=0):
    """Defaults = the reference CLI defaults (librabft-v2/src/main.rs:73-140)."""
    cfg = OracleConfig()
    cfg.num_nodes = num_nodes
    cfg.delay_model = delay_model
    cfg.mean = mean
    cfg.variance = variance
    cfg.uniform_lo = uniform_lo
    cfg.uniform_hi = uniform_hi
    cfg.commands_per_epoch = commands_per_epoch
    cfg.target_commit_interval = target_commit_interval
    cfg.delta = delta
    cfg.gamma = gamma
    cfg.lambda_ = lambda_
    cfg.quirks = quirks
    cfg.math_mode = math_mode
    cfg.equivocate_every = equivocate_every
    cfg.drop_per_million = drop_per_million
    cfg.partition_size = partition_size
    cfg.partition_start = partition_start
    cfg.partition_end = partition_end
    cfg.rights_rotation = rights_rotation
    cfg.reference_overheads = reference_overheads
    if voting_rights is not None:
        arr = (C.c_uint64 * num_nodes)(*voting_rights)
        cfg._keepalive = arr
        cfg.voting_rights = C.cast(arr, C.POINTER(C.c_uint64))
    return cfg


class OracleSim:
    """One simulated network: Simulator::new + loop_until (bft-lib/src/simulator.rs:200-250,380-475)."""

    def __init__(self, cfg, seed):
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = lib().lbft_oracle_create(C.byref(cfg), seed, C.byref(self.h))
        if rc != 0:
            raise RuntimeError("lbft_oracle_create failed: %d" % rc)

    def run_until(self, max_clock):
        rc = lib().lbft_oracle_run_until(self.h, max_clock)
        if rc != 0:
            raise RuntimeError("oracle panic: %s" % lib().lbft_oracle_last_error(self.h).decode())
        return self

    def commit_counts(self):
        return [lib().lbft_oracle_commit_count(self.h, n) for n in range(self.cfg.num_nodes)]

    def committed_history(self, node):
        n = lib().lbft_oracle_commit_count(self.h, node)
        out = np.zeros(n, dtype=COMMIT_DTYPE)
        lib().lbft_oracle_committed_history(self.h, node, out.ctypes.data, n)
        return out

    def committed_record_hashes(self, node):
        """(block_hash, state, qc_hash, has_qc, num_votes) of the records behind committed_history(node)."""
        n = lib().lbft_oracle_commit_count(self.h, node)
        out = np.zeros(n, dtype=RECORD_HASH_DTYPE)
        lib().lbft_oracle_committed_record_hashes(self.h, node, out.ctypes.data, n)
        return out

    def save_node(self, node):
        """ConsensusNode::save_node (node.rs:233-238): the canonical-order bincode image of the node's NodeState."""
        n = lib().lbft_oracle_save_node(self.h, node, None, 0)
        buf = np.zeros(n, dtype=np.uint8)
        assert lib().lbft_oracle_save_node(self.h, node, buf.ctypes.data, n) == n
        return buf.tobytes()

    def last_committed_states(self):
        return [lib().lbft_oracle_last_committed_state(self.h, n) for n in range(self.cfg.num_nodes)]

    def active_rounds(self):
        return [lib().lbft_oracle_active_round(self.h, n) for n in range(self.cfg.num_nodes)]

    def epochs(self):
        return [lib().lbft_oracle_epoch(self.h, n) for n in range(self.cfg.num_nodes)]

    def startup_times(self):
        return [lib().lbft_oracle_startup_time(self.h, n) for n in range(self.cfg.num_nodes)]

    def counters(self):
        c = OracleCounters()
        lib().lbft_oracle_counters_get(self.h, C.byref(c))
        return c.as_dict()

    # ---- DataWriter (bft-lib/src/data_writer.rs) ----
    def enable_data_writer(self):
        lib().lbft_oracle_enable_data_writer(self.h)
        return self

    def round_switches(self, cap_rounds=4096):
        """(rows, messages): rows[round][node] = time or None, exactly the cells of round_switches.txt."""
        out = np.full((cap_rounds, self.cfg.num_nodes), np.iinfo(np.int64).min, dtype=np.int64)
        msgs = C.c_uint64()
        mr = int(lib().lbft_oracle_round_switches(self.h, out.ctypes.data, cap_rounds, C.byref(msgs)))
        rows = [[None if v == np.iinfo(np.int64).min else int(v) for v in out[r]] for r in range(min(mr, cap_rounds))]
        return rows, int(msgs.value)

    # ---- node-level interface (the reference's traits on one node; no event loop) ----
    def node_update(self, node, clock):
        a = OracleActions()
        rc = lib().lbft_oracle_node_update(self.h, node, clock, C.byref(a))
        if rc != 0:
            raise RuntimeError("oracle node_update failed: %d %s" % (rc, lib().lbft_oracle_last_error(self.h).decode()))
        return a.as_dict()

    def node_create_notification(self, node):
        h = lib().lbft_oracle_node_create_notification(self.h, node)
        if h < 0:
            raise RuntimeError("oracle create_notification failed: %d" % h)
        return h

    def node_handle_notification(self, receiver, handle):
        sync = C.c_uint32()
        rc = lib().lbft_oracle_node_handle_notification(self.h, receiver, handle, C.byref(sync))
        if rc != 0:
            raise RuntimeError("oracle handle_notification failed: %d %s" % (rc, lib().lbft_oracle_last_error(self.h).decode()))
        return bool(sync.value)

    def node_create_request(self, node):
        h = lib().lbft_oracle_node_create_request(self.h, node)
        if h < 0:
            raise RuntimeError("oracle create_request failed: %d" % h)
        return h

    def node_handle_request(self, node, request):
        h = lib().lbft_oracle_node_handle_request(self.h, node, request)
        if h < 0:
            raise RuntimeError("oracle handle_request failed: %d %s" % (h, lib().lbft_oracle_last_error(self.h).decode()))
        return h

    def node_handle_response(self, node, response, clock):
        rc = lib().lbft_oracle_node_handle_response(self.h, node, response, clock)
        if rc != 0:
            raise RuntimeError("oracle handle_response failed: %d %s" % (rc, lib().lbft_oracle_last_error(self.h).decode()))

    def node_view(self, node):
        v = OracleNodeView()
        lib().lbft_oracle_node_view_get(self.h, node, C.byref(v))
        return v.as_dict()

    def close(self):
        if self.h:
            lib().lbft_oracle_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_batch(cfg, seeds, max_clock, threads=1, history_cap=0):
    """Run len(seeds) independent instances; returns dict of numpy arrays + summed counters."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    m, nn = len(seeds), cfg.num_nodes
    commit_counts = np.zeros((m, nn), dtype=np.uint32)
    active_rounds = np.zeros((m, nn), dtype=np.uint64)
    last_states = np.zeros((m, nn), dtype=np.uint64)
    hist = np.zeros((m, nn, history_cap), dtype=COMMIT_DTYPE) if history_cap else None
    ctr = OracleCounters()
    rc = lib().lbft_oracle_run_batch(
        C.byref(cfg), seeds.ctypes.data, m, max_clock, threads, commit_counts.ctypes.data,
        active_rounds.ctypes.data, last_states.ctypes.data, hist.ctypes.data if hist is not None else None,
        history_cap, C.byref(ctr))
    if rc != 0:
        raise RuntimeError("oracle batch failed: %d" % rc)
    return {"commit_counts": commit_counts, "active_rounds": active_rounds, "last_states": last_states,
            "histories": hist, "counters": ctr.as_dict()}


def leader(num_nodes, rnd, weights=None):
    w = None
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.uint64)
    return lib().lbft_oracle_leader(w.ctypes.data if w is not None else None, num_nodes, rnd)


# ----------------------------------------------------------------------------------------------
# Host build of the kernel logic (oracle/host_model.cpp) -- CPU-only differential testing.
# ----------------------------------------------------------------------------------------------
class HostModelCaps(C.Structure):
    _fields_ = [("qcap", C.c_uint32), ("scap", C.c_uint32), ("bcap", C.c_uint32), ("lcap", C.c_uint32), ("ql", C.c_uint32), ("qheap", C.c_uint32), ("force_generic", C.c_uint32), ("rcap", C.c_uint32), ("qcal", C.c_uint32), ("ring", C.c_uint32), ("ring_topup", C.c_uint32), ("tw", C.c_uint32), ("keep_stores", C.c_uint32)]


_hm = None


def hostmodel_lib():
    global _hm
    if _hm is None:
        build()
        # LBFT_HOSTMODEL_LIB: an experimental build of the kernel logic (e.g. liblbft_hostmodel_coop0.so, `make -C oracle coop0`)
        path = os.environ.get("LBFT_HOSTMODEL_LIB") or os.path.join(_HERE, "liblbft_hostmodel.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = C.CDLL(path)
        vp = C.c_void_p
        L.lbft_hostmodel_run_batch.argtypes = [
            C.POINTER(OracleConfig), C.POINTER(HostModelCaps), vp, C.c_size_t, C.c_int64, C.c_uint32, vp, vp, vp,
            vp, C.c_size_t, C.POINTER(OracleCounters), vp, vp, vp, vp, vp, vp, C.c_size_t]
        L.lbft_hostmodel_run_batch.restype = C.c_int
        _hm = L
    return _hm


def hostmodel_node_images(cfg, seed, max_clock, roundtrip=None, **caps):
    """save_node images (bytes, or None when unsupported) of every node of ONE network run on the host model: the image builder of the
    product library (csrc/lbft_save_node.h) applied to the host model's state rows.  `roundtrip`: a list that receives, per node, the
    result of save -> scrub the node's NodeState -> load_node_image (the product library's loader) -> save: 0 = byte-identical."""
    L = hostmodel_lib()
    rt = None
    if roundtrip is not None:
        rt = (C.c_int * cfg.num_nodes)(*([-999] * cfg.num_nodes))
        L.lbft_hostmodel_roundtrip_node_images.argtypes = [C.c_void_p]
        L.lbft_hostmodel_roundtrip_node_images.restype = None
        L.lbft_hostmodel_roundtrip_node_images(C.addressof(rt))
    n, stride = cfg.num_nodes, 1 << 22
    buf = np.zeros(n * stride, dtype=np.uint8)
    lens = (C.c_size_t * n)()
    L.lbft_hostmodel_capture_node_images.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.lbft_hostmodel_capture_node_images.restype = None
    L.lbft_hostmodel_capture_node_images(buf.ctypes.data, stride, C.addressof(lens))
    res = hostmodel_run_batch(cfg, np.array([seed], dtype=np.uint64), max_clock, **caps)
    assert not res["faults"].any()
    out = []
    for k in range(n):
        ln = int(lens[k])
        out.append(None if ln == 2 ** 64 - 1 else buf[k * stride:k * stride + ln].tobytes())
    if roundtrip is not None:
        roundtrip[:] = [int(v) for v in rt]
    return out


def hostmodel_run_batch(cfg, seeds, max_clock, threads=1, history_cap=0, qcap=256, scap=128, bcap=256, lcap=256, ql=0, qheap=0, force_generic=0, rcap=0, qcal=0,
                        hash_cap=0, ring=0, ring_topup=0, tw=0, keep_stores=0):
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    m, nn = len(seeds), cfg.num_nodes
    caps = HostModelCaps(qcap, scap, bcap, lcap, ql, qheap, force_generic, rcap, qcal, ring, ring_topup, tw, keep_stores)
    rs = np.zeros((m, max(rcap, 1), nn), dtype=np.int64)
    mr = np.zeros(m, dtype=np.uint32)
    commit_counts = np.zeros((m, nn), dtype=np.uint32)
    active_rounds = np.zeros((m, nn), dtype=np.uint64)
    last_states = np.zeros((m, nn), dtype=np.uint64)
    hist = np.zeros((m, nn, history_cap), dtype=COMMIT_DTYPE) if history_cap else None
    faults = np.zeros(m, dtype=np.uint32)
    maxq = np.zeros(m, dtype=np.uint32)
    maxsnap = np.zeros(m, dtype=np.uint32)
    rh = np.zeros((m, nn, hash_cap, 4), dtype=np.uint64) if hash_cap else None  # (block hash, State, QC hash, votes | flags << 32)
    ctr = OracleCounters()
    rc = hostmodel_lib().lbft_hostmodel_run_batch(
        C.byref(cfg), C.byref(caps), seeds.ctypes.data, m, max_clock, threads, commit_counts.ctypes.data,
        active_rounds.ctypes.data, last_states.ctypes.data, hist.ctypes.data if hist is not None else None,
        history_cap, C.byref(ctr), faults.ctypes.data, maxq.ctypes.data, maxsnap.ctypes.data, rs.ctypes.data, mr.ctypes.data,
        rh.ctypes.data if rh is not None else None, hash_cap)
    if rc < 0:
        raise RuntimeError("host model failed: %d" % rc)
    return {"commit_counts": commit_counts, "active_rounds": active_rounds, "last_states": last_states,
            "histories": hist, "counters": ctr.as_dict(), "faults": faults, "maxq": maxq, "maxsnap": maxsnap,
            "round_switches": rs, "max_rounds": mr, "record_hashes": rh}
