"""ctypes binding of include/lbft.h (liblbft_hip.so).  Fails loudly when the HIP library is missing:
there is no CPU fallback in this package."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# LBFT_HIP_LIB selects another in-tree build of the same sources (e.g. the phase-timer diagnostic build)
LIB_PATH = os.environ.get("LBFT_HIP_LIB") or os.path.join(HERE, "liblbft_hip.so")

LBFT_OK = 0
LBFT_ERR_INVALID = -1
LBFT_ERR_HIP = -2
LBFT_ERR_UNSUPPORTED = -3
LBFT_ERR_STATE = -4
LBFT_ERR_FAULT = -5
LBFT_FAULT_TRACE_OVERFLOW = 1 << 11  # include/lbft.h: a node went past the round-switch trace's capacity

FAULT_NAMES = {
    1 << 0: "queue_overflow", 1 << 1: "snapshot_overflow", 1 << 2: "block_overflow", 1 << 3: "log_overflow",
    1 << 4: "ballot_overflow", 1 << 5: "duration_table", 1 << 6: "commit_unknown_state",
    1 << 7: "commit_not_successor", 1 << 8: "stamp_overflow", 1 << 9: "internal", 1 << 11: "trace_overflow", 1 << 12: "epoch_overflow",
}


class LbftConfig(C.Structure):
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
        ("voting_rights", C.POINTER(C.c_uint64)),
        ("queue_capacity", C.c_uint32),
        ("snapshot_capacity", C.c_uint32),
        ("block_capacity", C.c_uint32),
        ("log_capacity", C.c_uint32),
        ("drop_per_million", C.c_uint32),
        ("partition_size", C.c_uint32),
        ("partition_start", C.c_int64),
        ("partition_end", C.c_int64),
        ("rights_rotation", C.c_uint32),
    ]


class LbftCounters(C.Structure):
    _fields_ = [
        ("events", C.c_uint64 * 4),
        ("rng_draws", C.c_uint64),
        ("rounds", C.c_uint64),
        ("commits", C.c_uint64),
        ("events_scheduled", C.c_uint64),
        ("faulted_instances", C.c_uint64),
        ("max_queue", C.c_uint64),
        ("max_snapshots", C.c_uint64),
        ("max_blocks", C.c_uint64),
        ("launches", C.c_uint64),
        ("timers_folded", C.c_uint64),
        ("node_updates", C.c_uint64),
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_ if name != "events"}
        d["events"] = list(self.events)
        return d


class LbftActions(C.Structure):
    """NodeUpdateActions (bft-lib/src/interfaces.rs:12-21)."""
    _fields_ = [("next_scheduled_update", C.c_int64), ("should_send", C.c_uint64 * 2), ("should_broadcast", C.c_uint32),
                ("should_query_all", C.c_uint32)]

    def as_dict(self):
        bits = int(self.should_send[0]) | (int(self.should_send[1]) << 64)
        return {"next_scheduled_update": int(self.next_scheduled_update),
                "should_send": [a for a in range(128) if (bits >> a) & 1],
                "should_broadcast": bool(self.should_broadcast), "should_query_all": bool(self.should_query_all)}


class LbftNodeCall(C.Structure):
    """lbft_node_call: one trait call of a batch of calls (lbft_node_calls)."""
    _fields_ = [("op", C.c_uint32), ("instance", C.c_uint32), ("node", C.c_uint32), ("peer", C.c_uint32), ("handle", C.c_uint32),
                ("reserved", C.c_uint32), ("node_time", C.c_int64)]


class LbftNodeResult(C.Structure):
    _fields_ = [("actions", LbftActions), ("handle", C.c_uint32), ("should_sync", C.c_uint32), ("status", C.c_int32), ("reserved", C.c_uint32)]


CALL_UPDATE_NODE, CALL_CREATE_NOTIFICATION, CALL_HANDLE_NOTIFICATION, CALL_RELEASE_NOTIFICATION, CALL_CREATE_REQUEST, CALL_HANDLE_REQUEST, CALL_HANDLE_RESPONSE = range(7)


class LbftNodeView(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("epoch_id", "current_round", "highest_quorum_certificate_round",
                                           "highest_timeout_certificate_round", "highest_committed_round", "active_round",
                                           "latest_voted_round", "locked_round", "commit_count")] + \
               [(n, C.c_uint32) for n in ("active_leader", "election", "num_current_timeouts", "num_current_votes",
                                           "has_proposed_block", "has_timeout_certificate")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


COMMIT_DTYPE = np.dtype([("proposer", "<u8"), ("index", "<u8"), ("time", "<i8")])
RECORD_HASH_DTYPE = np.dtype([("block_hash", "<u8"), ("state", "<u8"), ("qc_hash", "<u8"), ("num_votes", "<u4"), ("flags", "<u4")])

# every symbol include/lbft.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "lbft_batch_create", "lbft_batch_run_until", "lbft_batch_reset", "lbft_batch_commit_counts",
    "lbft_batch_active_rounds", "lbft_batch_committed_history", "lbft_batch_committed_histories", "lbft_batch_committed_record_hashes",
    "lbft_batch_last_committed_states", "lbft_batch_last_committed_state", "lbft_batch_save_node", "lbft_batch_load_node", "lbft_batch_startup_times", "lbft_batch_epochs", "lbft_batch_counters",
    "lbft_batch_faults", "lbft_batch_destroy", "lbft_batch_stream", "lbft_batch_last_run_ms",
    "lbft_batch_device_bytes", "lbft_batch_set_max_steps", "lbft_batch_set_lanes_per_wavefront",
    "lbft_batch_set_lds_queue_slots", "lbft_batch_set_calendar_queue", "lbft_batch_phase_cycles", "lbft_batch_layout",
    "lbft_batch_run_steps", "lbft_batch_checkpoint_bytes", "lbft_batch_checkpoint_save", "lbft_batch_checkpoint_load",
    "lbft_batch_enable_round_trace", "lbft_batch_keep_retired_stores", "lbft_batch_counters_allreduce", "lbft_batch_counters_allgather_reduce", "lbft_node_calls", "lbft_batch_round_switches", "lbft_batch_manual_begin", "lbft_batch_manual_finalize", "lbft_node_update", "lbft_node_create_notification",
    "lbft_node_handle_notification", "lbft_node_release_notification", "lbft_node_create_request", "lbft_node_handle_request",
    "lbft_node_handle_response", "lbft_node_view_get", "lbft_device_leaders", "lbft_device_sample_delays",
    "lbft_device_exp_log", "lbft_last_error", "lbft_build_info",
]

_lib = None


class LbftError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lbft error %d: %s" % (code, msg))
        self.code = code


def _needed_hip_soname(path, tag_wanted=1):
    """DT_NEEDED entry of `path` that names the HIP runtime (e.g. "libamdhip64.so.7"; tag_wanted = 14: the file's own DT_SONAME), or
    None.  Pure-Python ELF walk that reads only the ELF header, the section headers and the .dynamic / .dynstr sections (seek + read:
    torch's libamdhip64.so is tens of MB).  A file that cannot be parsed raises ValueError: the caller must not guess."""
    import struct
    with open(path, "rb") as f:
        head = f.read(0x40)
        if len(head) < 0x40 or head[:4] != b"\x7fELF" or head[4] != 2 or head[5] != 1:
            raise ValueError("%s: not a little-endian 64-bit ELF file" % path)
        shoff, = struct.unpack_from("<Q", head, 0x28)
        shentsize, shnum, _ = struct.unpack_from("<HHH", head, 0x3A)
        if shentsize < 0x40 or shnum == 0 or shnum > 65535:
            raise ValueError("%s: no section headers" % path)
        f.seek(shoff)
        table = f.read(shentsize * shnum)
        secs = [struct.unpack_from("<IIQQQQII", table, i * shentsize) for i in range(shnum)]
        for _name, typ, _flags, _addr, off, size, link, _info in secs:
            if typ != 6:  # SHT_DYNAMIC
                continue
            if link >= shnum:
                raise ValueError("%s: .dynamic without a string table" % path)
            stroff, strsize = secs[link][4], secs[link][5]
            f.seek(off)
            dyn = f.read(size)
            f.seek(stroff)
            strtab = f.read(strsize)
            for k in range(len(dyn) // 16):
                tag, val = struct.unpack_from("<qQ", dyn, 16 * k)
                if tag == tag_wanted and val < len(strtab):  # DT_NEEDED = 1, DT_SONAME = 14
                    s = strtab[val:strtab.index(b"\0", val)].decode()
                    if s.startswith("libamdhip64"):
                        return s
    return None


from struct import error as struct_error  # noqa: E402


def _one_hip_runtime():
    """PyTorch-ROCm wheels ship their own libamdhip64.so.7 / libhsa-runtime64 under torch/lib.  A process must hold ONE ROCr runtime: if
    liblbft_hip.so pulls in /opt/rocm's first and torch is imported afterwards, torch loads its own copy next to it and finds no device
    (torch.cuda.is_available() turns False).  Loaded the other way round, liblbft_hip.so's libamdhip64.so.7 resolves to the copy already
    in the process.  So: when torch is installed but not imported yet, map torch's runtime first -- unless LBFT_NO_TORCH_HIP=1 says not
    to (a process that will never import torch), or torch's bundled runtime has another soname (major version) than the one
    liblbft_hip.so was linked against: binding to a foreign HIP runtime is worse than the system one, so /opt/rocm's is used then."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("LBFT_NO_TORCH_HIP", "") not in ("", "0"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    path = os.path.join(libdir, "libamdhip64.so")
    if not os.path.exists(path):
        return
    try:
        want, have = _needed_hip_soname(LIB_PATH), _needed_hip_soname(path, 14)
    except (OSError, ValueError, struct_error) as e:
        # (round-4 advisor: a parse failure used to fall through to "map torch's runtime whatever its version")  Unknown sonames = no
        # evidence that torch's copy is the runtime this library was linked against: leave the choice to the dynamic loader and say so.
        import warnings
        warnings.warn("librabft_simulator_amd: could not read the HIP runtime sonames (%s); torch's bundled libamdhip64 is NOT pre-loaded -- import torch "
                      "before this package if both are used in one process" % e)
        return
    if want and have and want != have:
        return  # torch bundles a runtime of another soname than this library needs: leave it to the loader (/opt/rocm)
    C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Load liblbft_hip.so (built in-tree by librabft_simulator_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m librabft_simulator_amd.build` (needs hipcc). "
            "There is no CPU fallback." % LIB_PATH)
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lbft_batch_create.argtypes = [C.POINTER(LbftConfig), vp, C.c_size_t, C.c_int, C.POINTER(vp)]
    L.lbft_batch_create.restype = C.c_int
    L.lbft_batch_run_until.argtypes = [vp, C.c_int64]
    L.lbft_batch_run_until.restype = C.c_int
    L.lbft_batch_reset.argtypes = [vp]
    L.lbft_batch_reset.restype = C.c_int
    for name in ("commit_counts", "active_rounds", "last_committed_states", "startup_times", "epochs", "faults"):
        f = getattr(L, "lbft_batch_" + name)
        f.argtypes = [vp, vp]
        f.restype = C.c_int
    L.lbft_batch_committed_history.argtypes = [vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lbft_batch_committed_history.restype = C.c_int
    L.lbft_batch_committed_record_hashes.argtypes = [vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lbft_batch_committed_record_hashes.restype = C.c_int
    L.lbft_batch_committed_histories.argtypes = [vp, vp, C.c_size_t]
    L.lbft_batch_committed_histories.restype = C.c_int
    L.lbft_batch_last_committed_state.argtypes = [vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64)]
    L.lbft_batch_last_committed_state.restype = C.c_int
    L.lbft_batch_save_node.argtypes = [vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lbft_batch_save_node.restype = C.c_int
    L.lbft_batch_load_node.argtypes = [vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, C.c_int64]
    L.lbft_batch_load_node.restype = C.c_int
    L.lbft_batch_counters.argtypes = [vp, C.POINTER(LbftCounters)]
    L.lbft_batch_counters.restype = C.c_int
    L.lbft_batch_destroy.argtypes = [vp]
    L.lbft_batch_destroy.restype = None
    L.lbft_batch_stream.argtypes = [vp]
    L.lbft_batch_stream.restype = vp
    L.lbft_batch_last_run_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.lbft_batch_last_run_ms.restype = C.c_int
    L.lbft_batch_device_bytes.argtypes = [vp]
    L.lbft_batch_device_bytes.restype = C.c_size_t
    L.lbft_batch_set_max_steps.argtypes = [vp, C.c_uint32]
    L.lbft_batch_set_max_steps.restype = C.c_int
    L.lbft_batch_set_lanes_per_wavefront.argtypes = [vp, C.c_uint32]
    L.lbft_batch_set_lanes_per_wavefront.restype = C.c_int
    L.lbft_batch_set_lds_queue_slots.argtypes = [vp, C.c_int32]
    L.lbft_batch_set_lds_queue_slots.restype = C.c_int
    L.lbft_batch_run_steps.argtypes = [vp, C.c_int64, C.c_uint32, C.POINTER(C.c_uint64)]
    L.lbft_batch_run_steps.restype = C.c_int
    L.lbft_batch_checkpoint_bytes.argtypes = [vp]
    L.lbft_batch_checkpoint_bytes.restype = C.c_size_t
    L.lbft_batch_checkpoint_save.argtypes = [vp, vp, C.c_size_t]
    L.lbft_batch_checkpoint_save.restype = C.c_int
    L.lbft_batch_checkpoint_load.argtypes = [vp, vp, C.c_size_t]
    L.lbft_batch_checkpoint_load.restype = C.c_int
    L.lbft_node_calls.argtypes = [vp, C.POINTER(LbftNodeCall), C.c_size_t, C.POINTER(LbftNodeResult)]
    L.lbft_node_calls.restype = C.c_int
    for fn in (L.lbft_batch_counters_allreduce, L.lbft_batch_counters_allgather_reduce):
        fn.argtypes = [vp, vp, C.POINTER(LbftCounters)]
        fn.restype = C.c_int
    L.lbft_batch_keep_retired_stores.argtypes = [vp, C.c_int]
    L.lbft_batch_keep_retired_stores.restype = C.c_int
    L.lbft_batch_enable_round_trace.argtypes = [vp, C.c_uint32]
    L.lbft_batch_enable_round_trace.restype = C.c_int
    L.lbft_batch_round_switches.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.lbft_batch_round_switches.restype = C.c_int
    L.lbft_batch_manual_begin.argtypes = [vp, C.c_int64]
    L.lbft_batch_manual_begin.restype = C.c_int
    L.lbft_batch_manual_finalize.argtypes = [vp]
    L.lbft_batch_manual_finalize.restype = C.c_int
    L.lbft_node_update.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_int64, C.POINTER(LbftActions)]
    L.lbft_node_update.restype = C.c_int
    L.lbft_node_create_notification.argtypes = [vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)]
    L.lbft_node_create_notification.restype = C.c_int
    L.lbft_node_handle_notification.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.lbft_node_handle_notification.restype = C.c_int
    L.lbft_node_release_notification.argtypes = [vp, C.c_size_t, C.c_uint32]
    L.lbft_node_release_notification.restype = C.c_int
    L.lbft_node_create_request.argtypes = [vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)]
    L.lbft_node_create_request.restype = C.c_int
    L.lbft_node_handle_request.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.lbft_node_handle_request.restype = C.c_int
    L.lbft_node_handle_response.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64]
    L.lbft_node_handle_response.restype = C.c_int
    L.lbft_node_view_get.argtypes = [vp, C.c_size_t, C.c_uint32, C.POINTER(LbftNodeView)]
    L.lbft_node_view_get.restype = C.c_int
    L.lbft_batch_set_calendar_queue.argtypes = [vp, C.c_int]
    L.lbft_batch_set_calendar_queue.restype = C.c_int
    L.lbft_batch_layout.argtypes = [vp, vp]
    L.lbft_batch_layout.restype = C.c_int
    L.lbft_batch_phase_cycles.argtypes = [vp, vp]
    L.lbft_batch_phase_cycles.restype = C.c_int
    L.lbft_device_leaders.argtypes = [C.c_int, vp, C.c_uint32, vp, C.c_uint32]
    L.lbft_device_leaders.restype = C.c_int
    L.lbft_device_sample_delays.argtypes = [C.c_int, C.POINTER(LbftConfig), C.c_uint64, vp, C.c_size_t]
    L.lbft_device_sample_delays.restype = C.c_int
    L.lbft_device_exp_log.argtypes = [C.c_int, vp, vp, vp, C.c_size_t]
    L.lbft_device_exp_log.restype = C.c_int
    L.lbft_last_error.argtypes = []
    L.lbft_last_error.restype = C.c_char_p
    L.lbft_build_info.argtypes = []
    L.lbft_build_info.restype = C.c_char_p
    _lib = L
    return L


def check(rc, allow_fault=False):
    if rc == LBFT_OK or (allow_fault and rc == LBFT_ERR_FAULT):
        return rc
    raise LbftError(rc, lib().lbft_last_error().decode(errors="replace"))
