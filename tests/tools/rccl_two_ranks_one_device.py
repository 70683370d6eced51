#!/usr/bin/env python3
"""GPU box with ONE device: the 2-rank path of the native collective as far as RCCL lets it go there.  Two processes (rank 0 / 1) on HIP device 0 exchange a
ncclUniqueId through a file and call ncclCommInitRank(world = 2).  RCCL refuses two ranks of one communicator on the same device (ncclInvalidUsage,
"Duplicate GPU detected") -- AFTER librccl was loaded, the id travelled and the bootstrap ring over 127.0.0.1 connected, i.e. everything of the multi-GPU
path that does not need a second GPU has then run.  Should a RCCL build accept it, the script goes on: lbft_batch_counters_allgather_reduce on both
ranks, checked against the sum of the two ranks' own counters.  Prints one JSON line: {"rank", "init_rc", "collective": null | "ok" | "mismatch"}.
    python tests/tools/rccl_two_ranks_one_device.py <rank> <uid file>"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, path = int(sys.argv[1]), sys.argv[2]
    import numpy as np
    import librabft_simulator_amd as amd
    from librabft_simulator_amd import _lib
    _lib.lib()  # the HIP library (and with it ONE HIP runtime: _lib._one_hip_runtime) before librccl brings its own dependencies
    rccl = ctypes.CDLL("librccl.so.1")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        with open(path + ".tmp", "wb") as f:
            f.write(ctypes.string_at(ctypes.byref(uid), 128))
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 60:
                raise SystemExit("rank 1: no unique id from rank 0")
            time.sleep(0.05)
        ctypes.memmove(ctypes.byref(uid), open(path, "rb").read(), 128)
    # each rank runs its own shard first (the library picks HIP device 0)
    seeds = np.arange(1 + 256 * rank, 257 + 256 * rank, dtype=np.uint64)
    res = amd.BatchSimulator.new(seeds, 4, amd.RandomDelay.new(10.0, 4.0)).loop_until(300)
    own = res.counters
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    comm = ctypes.c_void_p()
    rc = rccl.ncclCommInitRank(ctypes.byref(comm), 2, uid, rank)
    out = {"rank": rank, "init_rc": rc, "collective": None, "rounds": own["rounds"]}
    if rc == 0:
        try:
            agg = res.counters_allgather_reduce(comm.value)
            out["collective"] = "ok" if agg["rounds"] >= own["rounds"] and agg["events"][0] >= own["events"][0] else "mismatch"
            out["agg_rounds"] = agg["rounds"]
        finally:
            rccl.ncclCommDestroy(comm)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
