#!/usr/bin/env python3
"""Headline benchmark: simulated consensus rounds/sec on 65 536 x 4-node LibraBFTv2 instances per GPU.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it with
torch.distributed.run, one rank per GPU (RCCL) -- and when it is started WITHOUT a launcher (`python bench.py --gpus 8`:
no WORLD_SIZE in the environment) it re-executes itself under `python -m torch.distributed.run --nproc-per-node N`
(rendezvous on 127.0.0.1, a free port), so the command line alone decides the number of ranks; a launcher whose
WORLD_SIZE disagrees with --gpus is an error.  `n_gpus` in the output is the communicator's size.  A "step" is one pass of the hot path over one batch of
synthetic input: Simulator::new + loop_until(max_clock) for every instance of the batch (seeds already
resident in HBM), including the device-side reduction and the D2H copy of the counters.  Rank 0 prints
ONE JSON line.  Instances shard across ranks with no data-path collective (weak scaling: 65 536 instances
per GPU -> `value`); ONE collective (an all-gather of one counter row per rank, reduced locally) aggregates the throughput counters.
For N > 1 the same command also times BASELINE config 3 as it is named -- ONE 65 536-instance batch split over the N GPUs (strong
scaling) -- in a second leg and reports it as `config3_strong` in the same line; `scaling` names which of the two `value` is.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak


def parse():
    ap = argparse.ArgumentParser(allow_abbrev=False)  # (no prefix matching: the child runs of --measure-traffic rebuild the command line from exact tokens)
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: the launcher's WORLD_SIZE, else 1); without a launcher `--gpus N` starts N ranks itself")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=65536, help="instances per GPU (weak scaling: the default mode)")
    ap.add_argument("--total-instances", type=int, default=0,
                    help="strong scaling: this many instances in total, sharded over the ranks (BASELINE.json config 3 read as ONE 65 536-instance batch on 1..8 GPUs)")
    ap.add_argument("--no-config3-strong", action="store_true",
                    help="(N > 1, weak mode) skip the second timed leg: BASELINE config 3 read as ONE batch of --instances instances split over the ranks")
    ap.add_argument("--parity-instances", type=int, default=16384, help="instances of the timed batch (strided) re-run on the CPU oracle and compared (0 = no gate)")
    ap.add_argument("--nodes", type=int, default=4)
    ap.add_argument("--max-clock", type=int, default=1000)
    ap.add_argument("--base-seed", type=int, default=1)
    ap.add_argument("--lpw", type=int, default=0, help="lanes per wavefront carrying an instance (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses HIP device 0")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the oracle sample")
    ap.add_argument("--measure-traffic", action="store_true", default=None,
                    help="(N = 1, rocprofv3 on the box) collect FETCH_SIZE / WRITE_SIZE of the run kernel in two child runs of this command: roofline.traffic is "
                         "then MEASURED IN THIS RUN (roofline.traffic_measured_here) instead of replayed from the committed profile.  Default since round 6 for "
                         "the headline workload on one GPU; --no-measure-traffic skips it")
    ap.add_argument("--no-measure-traffic", dest="measure_traffic", action="store_false")
    ap.add_argument("--native-collective", action="store_true",
                    help="also aggregate the counters through the C ABI's own collective (lbft_batch_counters_allgather_reduce: ncclAllGather on a "
                         "communicator built with ncclCommInitRank, one device per rank) and check it against the torch.distributed aggregate")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run with N ranks on this node.  `--standalone` lets the
    launcher pick its own free rendezvous port (no bind-close-reuse race here); `--local-addr 127.0.0.1` because the container's hostname may not
    resolve.  HSA_ENABLE_IPC_MODE_LEGACY=0 is only set when the caller's environment has no opinion: the host driver of this pool supports dmabuf IPC
    only, and without it RCCL's buffer exchange between the ranks fails with `hipIpcGetMemHandle: invalid argument`."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def traffic_measured_here(argv):
    """--measure-traffic: FETCH_SIZE and WRITE_SIZE of the run kernel collected NOW, each in its own rocprofv3 child run of this command
    (--kernel-trace only, as gpurun requires) -- per launch, with MI355X_MICROARCH.md's gfx950 correction (FETCH_SIZE counts a 128-byte
    request as 64 bytes: doubled).  None when rocprofv3 is not on the box or a pass fails."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if not prof:
        return {"error": "rocprofv3 not found"}
    if os.environ.get("LBFT_BENCH_CHILD"):  # (a child of this very function never profiles again, whatever its command line says)
        return {"error": "nested --measure-traffic refused"}
    child = [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--measure-traffic"] + ["--no-cpu-baseline", "--parity-instances", "0", "--no-measure-traffic"]
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="lbft_pmc_", dir="/tmp")
        try:
            r = subprocess.run([prof, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--"] + child,
                               env=dict(os.environ, TMPDIR="/tmp", LBFT_BENCH_CHILD="1"), cwd="/tmp", capture_output=True, text=True, timeout=240)
            per = collections.defaultdict(float)
            for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if "lbft_k_run" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                            per[row["Dispatch_Id"]] += float(row["Counter_Value"])
            if r.returncode != 0 or not per:
                return {"error": "%s pass failed (rc %d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-300:])}
            vals[counter] = sum(per.values()) / len(per)
        except Exception as e:  # noqa: BLE001 -- a profiler hiccup must not cost the bench line
            return {"error": "%s pass: %r" % (counter, e)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"fetch_kb_raw": vals["FETCH_SIZE"], "write_kb_raw": vals["WRITE_SIZE"], "gb_corrected": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / 1e9,
            "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, one child run each"}


def algorithmic_bytes_per_event(layout, counters):
    """SURVEY.md 8(d) with this build's struct sizes (DESIGN.md "Roofline"), taken from the library itself
    (lbft_batch_layout): 2*S_node + S_evt*(1 + p) + S_notif*(w + r), charged to every reference-equivalent event."""
    ev = sum(counters["events"])
    s_node, s_evt, s_notif = layout["node_bytes"], layout["event_bytes"], layout["snapshot_bytes"]
    p = counters["events_scheduled"] / max(ev, 1)
    r = counters["events"][0] / max(ev, 1)
    return 2 * s_node + s_evt * (1 + p) + s_notif * (r + r)


def executed_bytes(layout, counters):
    """The same formula charged only where the device moves the rows: duplicate timers folded at scheduling time never reach
    the queue nor load a node row (they are counted in events[3] to keep the reference's totals); cancelled timers and
    requests load a node's rows but do not write them back.  pops = events - timers_folded; node rows are read once per pop
    and written once per update_node call; one queue entry written and read per pop; notification snapshots as in 8(d)."""
    ev = sum(counters["events"])
    pops = ev - counters.get("timers_folded", 0)
    s_node, s_evt, s_notif = layout["node_bytes"], layout["event_bytes"], layout["snapshot_bytes"]
    return pops * s_node + counters.get("node_updates", pops) * s_node + 2 * pops * s_evt + 2 * counters["events"][0] * s_notif, pops


def measured_traffic_gb():
    """HBM bytes per launch of lbft_k_run from the PMC passes (FETCH_SIZE / WRITE_SIZE, collected in their own
    rocprofv3 runs of this same command by tools/gpu_profile.sh and committed under profiles/).  FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-byte requests as 64 bytes); the value is
    therefore an upper bound for narrow accesses.  PMC counters cannot be collected from inside this process, so the
    profile is stamped with the hash of the kernels' machine code it was taken with (librabft_simulator_amd.build.kernel_hash):
    None when no profile of the CURRENT kernels is committed -- a stale number is never replayed."""
    path = os.path.join(ROOT, "profiles", "current", "pmc_traffic.json")
    try:
        from librabft_simulator_amd.build import source_hash
        with open(path) as f:
            t = json.load(f)
        if t.get("source_hash") != source_hash():
            return None
        return {"fetch_kb_raw": t["FETCH_SIZE"], "write_kb_raw": t["WRITE_SIZE"],
                "gb_corrected": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / 1e9, "profile": t.get("profile"),
                "source_hash": t.get("source_hash")}
    except Exception:
        return None


def measured_issue():
    """Instruction-issue figures of lbft_k_run per launch from the SQ passes of the same stamped profile (the bound that binds: the
    headline kernel is VALU-issue / divergence bound, not HBM bound): VALU instructions, the share of the chip's VALU issue slots they
    occupy (4 cycles per wavefront instruction on a 16-lane SIMD, 1 024 SIMDs), and the lanes active per VALU instruction."""
    path = os.path.join(ROOT, "profiles", "current", "pmc_issue.json")
    try:
        from librabft_simulator_amd.build import source_hash
        with open(path) as f:
            t = json.load(f)
        if t.get("source_hash") != source_hash():
            return None
        cycles = t["GRBM_GUI_ACTIVE"] / 8.0  # (summed over the 8 XCDs)
        return {"valu_insts": t["SQ_INSTS_VALU"], "salu_insts": t.get("SQ_INSTS_SALU"),
                "valu_busy_frac": t["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles),
                "lanes_per_valu_inst": t["SQ_THREAD_CYCLES_VALU"] / t["SQ_INSTS_VALU"],
                "wave_cycles_issuing_frac": t.get("SQ_ACTIVE_INST_ANY", 0) / max(t.get("SQ_WAVE_CYCLES", 1), 1) if t.get("SQ_ACTIVE_INST_ANY") else None,
                "wave_cycles_waiting_frac": t.get("SQ_WAIT_ANY", 0) / max(t.get("SQ_WAVE_CYCLES", 1), 1) if t.get("SQ_WAIT_ANY") else None,
                "profile": t.get("profile"), "source_hash": t.get("source_hash")}
    except Exception:
        return None


def native_collective(res, rank, world, local_rank):
    """The C ABI's own collective on N ranks: a RCCL communicator built here with ncclCommInitRank (the unique id travels through
    torch.distributed's object broadcast), then lbft_batch_counters_allgather_reduce = ONE ncclAllGather on the batch's stream."""
    import ctypes
    import torch.distributed as dist
    rccl = ctypes.CDLL("librccl.so.1")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    box = [None]
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        box[0] = bytes(ctypes.string_at(ctypes.byref(uid), 128))
    dist.broadcast_object_list(box, src=0)
    ctypes.memmove(ctypes.byref(uid), box[0], 128)
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    comm = ctypes.c_void_p()
    rc = rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
    if rc != 0:
        raise RuntimeError("ncclCommInitRank failed: %d" % rc)
    try:
        return res.counters_allgather_reduce(comm.value)
    finally:
        rccl.ncclCommDestroy(comm)


def cpu_baseline(args, nodes, max_clock):
    """The CPU oracle ("port": C++ restatement of the reference, not the Rust binary -- no Rust toolchain in this image) on the
    host cores, bounded sample of the same workload.  `value` is measured with the oracle also paying what the reference pays
    around the protocol logic on every processed event (reference_overheads: bincode of the whole NodeState = save_node,
    simulator.rs:307-309, and a deep clone of the notification per receiver, :348-354); the bare protocol logic is reported
    beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_ctypes as oc
    host_threads = os.cpu_count() or 1
    # The baseline runs at the thread count that serves it BEST, not at os.cpu_count(): the GPU box reports 256 host threads but gives this
    # container the throughput of ~32 cores (round 6, tests/tools/oracle_scaling.py: 512 large networks 20 s on 32 threads, 51 s on 256), so the
    # probe below times a few counts and `cores` is the one used.
    cfg0 = oc.make_config(num_nodes=nodes, math_mode=1)
    best = None
    for t in sorted({min(host_threads, k) for k in (16, 32, 64, 128, host_threads)}):
        probe = 16 * t
        t0 = time.perf_counter()
        oc.run_batch(cfg0, np.arange(args.base_seed, args.base_seed + probe, dtype=np.uint64), max_clock, threads=t)
        rate = probe / max(time.perf_counter() - t0, 1e-3)
        if best is None or rate > best[0] * 1.05:
            best = (rate, t)
    cores = best[1]

    def timed(reference_overheads, seconds):
        cfg = oc.make_config(num_nodes=nodes, math_mode=1, reference_overheads=reference_overheads)
        probe = 8 * cores
        t0 = time.perf_counter()
        oc.run_batch(cfg, np.arange(args.base_seed, args.base_seed + probe, dtype=np.uint64), max_clock, threads=cores)
        dt = max(time.perf_counter() - t0, 1e-3)
        m = int(min(max(probe, probe * seconds / dt), 65536))
        seeds = np.arange(args.base_seed, args.base_seed + m, dtype=np.uint64)
        t0 = time.perf_counter()
        r = oc.run_batch(cfg, seeds, max_clock, threads=cores)
        return r["counters"], time.perf_counter() - t0, m

    c, dt, m = timed(1, args.cpu_seconds)
    cb, dtb, mb = timed(0, args.cpu_seconds / 2)
    # BASELINE.json configs[0], the reference's own CPU-runnable case: 1 instance x 3 nodes, fixed delay 10 (mean 10, variance 0),
    # ~100 rounds (max_clock 2800), one thread
    c1cfg = oc.make_config(num_nodes=3, mean=10.0, variance=0.0, math_mode=1, reference_overheads=1)
    t1 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t1 < 0.5:
        r1 = oc.run_batch(c1cfg, np.array([args.base_seed + reps], dtype=np.uint64), 2800, threads=1)
        reps += 1
    c1 = r1["counters"]["rounds"] * reps / (time.perf_counter() - t1)
    # `value` = the port's bare protocol logic (what the port measures reliably).  The figure with the reference's per-event
    # save_node / clone costs emulated is an ESTIMATE of the Rust binary (the emulation sorts maps and appends bytewise, costlier than
    # bincode: round-2 advisor) and is reported beside it, never as the baseline.
    return {
        "c1_single_thread_rounds_per_s": c1,
        "value": cb["rounds"] / dtb, "unit": "rounds/s", "cores": cores, "host_threads_available": host_threads, "kind": "port",
        "sample": "%d instances x %d nodes, LogNormal(10,4), max_clock %d, %d host threads, %.1f s; C++ port of the reference's protocol logic "
                  "(hash maps keyed by BCS+SipHash record hashes, history clones, real payloads), without its per-event save_node" % (mb, nodes, max_clock, cores, dtb),
        "events_per_s": sum(cb["events"]) / dtb, "commits_per_s": cb["commits"] / dtb,
        "with_reference_overheads_estimate": {
            "value": c["rounds"] / dt, "events_per_s": sum(c["events"]) / dt,
            "sample": "%d instances, %.1f s, plus an emulation of bincode(NodeState) per event (%.0f MB) and a notification clone per receiver: an "
                      "upper bound of what the Rust reference pays, not measured against it" % (m, dt, c["saved_bytes"] / 1e6)},
    }


def parity_gate(args, sim, res, seeds, nodes, max_clock):
    """every number is accompanied by a parity gate.  A strided sample of the TIMED batch's instances is re-run
    on the CPU oracle (test infrastructure, allowed in this leg only) and compared with what the device produced in the timed
    region: commit counts, active rounds and last_committed_state (SipHash of the whole committed history) of every node."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_ctypes as oc
    n = len(seeds)
    k = min(args.parity_instances, n)
    idx = np.unique(np.linspace(0, n - 1, k).astype(np.int64))
    cfg = oc.make_config(num_nodes=nodes, math_mode=1)
    t0 = time.perf_counter()
    ref = oc.run_batch(cfg, np.ascontiguousarray(seeds[idx]), max_clock, threads=min(os.cpu_count() or 1, 32))
    dt = time.perf_counter() - t0
    cc, ar, st = res.commit_counts[idx], res.active_rounds[idx], res.last_committed_states[idx]
    bad = (cc != ref["commit_counts"]).any(axis=1) | (ar != ref["active_rounds"]).any(axis=1) | (st != ref["last_states"]).any(axis=1)
    return {"checked_instances": int(len(idx)), "checked_nodes": int(len(idx) * nodes), "mismatches": int(bad.sum()),
            "compared": "commit counts, active rounds, State hash of the committed history, every node",
            "states_xor": "%016x" % int(np.bitwise_xor.reduce(st.astype(np.uint64).ravel())),
            "oracle_states_xor": "%016x" % int(np.bitwise_xor.reduce(ref["last_states"].astype(np.uint64).ravel())),
            "oracle_seconds": dt, "oracle": "oracle/lbft_oracle.cpp, math_mode 1"}


def run_kernel_name(kc):
    """The run kernel a batch executes, from lbft_batch_layout's flag word (include/lbft.h)."""
    cls = kc & 255
    if cls == 0:
        return "lbft_k_run0u" if kc & 32768 else "lbft_k_run0s" if kc & 8192 else "lbft_k_run0q" if kc & 16384 else "lbft_k_run0"
    if kc & 1024:  # the two-wavefronts-per-SIMD kernels
        return ("lbft_k_run2q" if kc & 4096 else "lbft_k_run2l") if cls == 2 else "lbft_k_run1l"
    return "lbft_k_run<%d>" % cls


def main():
    args = parse()
    import numpy as np
    import torch
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ  # (a scheduler that merely exports WORLD_SIZE is not one)
    if args.gpus is None:  # left at its default: adopt the launcher's size (`torchrun --nproc-per-node N bench.py` with no --gpus)
        args.gpus = int(os.environ["WORLD_SIZE"]) if under_launcher else 1
    if args.gpus > 1 and not under_launcher:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); start it as `python bench.py --gpus N` "
                         "or under torch.distributed.run --nproc-per-node N with the same N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from librabft_simulator_amd import BatchSimulator, NodeConfig, RandomDelay
    from librabft_simulator_amd.distributed import shard_seeds
    # weak scaling (default): `--instances` per GPU; strong scaling: `--total-instances` sharded over the ranks.
    # seed_i = base_seed + global instance index in both modes
    strong = args.total_instances > 0
    total = args.total_instances if strong else args.instances * world
    seeds = shard_seeds(args.base_seed, total, rank, world)
    m = len(seeds)
    sim = BatchSimulator.new(seeds, args.nodes, RandomDelay.new(10.0, 4.0), NodeConfig(), device=local_rank, lanes_per_wavefront=args.lpw)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_leg(s):
        """W untimed warmup steps, then exactly K steps bracketed by barrier + synchronize on both sides."""
        r = None
        for _ in range(args.warmup):
            s.reset()
            r = s.loop_until(args.max_clock)
        kms = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s.reset()
            r = s.loop_until(args.max_clock)
            kms.append(s.last_run_ms()[1])  # hipEvents around the run kernel on the batch's own stream
        barrier()
        return r, kms, time.perf_counter() - t0

    res, kernel_ms, elapsed = timed_leg(sim)
    c = res.counters
    # BASELINE config 3 is ONE 65 536-instance batch on 1 -> 8 GPUs, i.e. the STRONG split; the driver's `--gpus N` line reports the weak batch as
    # `value` (per-GPU work fixed) and, from a second timed leg of the same command, the strong split beside it as `config3_strong`
    # (round-5 review): --instances instances in total, contiguous shards of 1/N each
    strong_leg = None
    if not strong and world > 1 and not args.no_config3_strong:
        seeds_s = shard_seeds(args.base_seed, args.instances, rank, world)
        sim_s = BatchSimulator.new(seeds_s, args.nodes, RandomDelay.new(10.0, 4.0), NodeConfig(), device=local_rank, lanes_per_wavefront=args.lpw)
        res_s, kms_s, elapsed_s = timed_leg(sim_s)
        cs = res_s.counters
        strong_leg = {"elapsed": elapsed_s, "sums": [cs["rounds"], cs["commits"], sum(cs["events"]), cs["faulted_instances"]], "instances_per_gpu": len(seeds_s),
                      "kernel_ms": float(np.mean(kms_s)), "kernel": run_kernel_name(sim_s.layout().get("kernel_class", 0))}
    # the single collective of the run: one all-gather of every rank's row (throughput counters + its timed regions);
    # sums / the slowest rank are reduced locally
    from librabft_simulator_amd.distributed import aggregate_counters
    agg = aggregate_counters(c, device=("cuda" if (dist is None or args.backend == "nccl") else None),
                             extra_max=[elapsed] + ([strong_leg["elapsed"]] if strong_leg else []), extra_sum=strong_leg["sums"] if strong_leg else [])
    rounds, commits, events, faulted = float(agg["rounds"]), float(agg["commits"]), float(sum(agg["events"])), float(agg["faulted_instances"])
    elapsed = float(agg["extra_max"][0])
    if dist is not None:
        world = dist.get_world_size()  # n_gpus = the communicator's size
    native = None
    if args.native_collective and dist is not None and args.backend == "nccl" and not args.single_device:
        nat = native_collective(res, rank, world, local_rank)
        keys = ("events", "rounds", "commits", "rng_draws", "events_scheduled", "faulted_instances", "timers_folded", "node_updates",
                "max_queue", "max_snapshots", "max_blocks")
        native = {"ranks": world, "matches_torch_aggregate": all(nat[k] == agg[k] for k in keys), "rounds": nat["rounds"]}
    if rank == 0:
        per_step = elapsed / args.steps
        k_ms = float(np.mean(kernel_ms))
        layout = sim.layout()
        bpe = algorithmic_bytes_per_event(layout, c)
        headline = (m, args.nodes, args.max_clock) == (65536, 4, 1000)
        traffic = measured_traffic_gb() if headline else None
        issue = measured_issue() if headline else None
        # PMC counters cannot be read from inside this process: the headline line on one GPU collects them in two rocprofv3 child runs of the same
        # command (bounded: 240 s each; a box without rocprofv3, or a failing pass, leaves the replayed value and says so)
        want_here = args.measure_traffic if args.measure_traffic is not None else (headline and world == 1 and not os.environ.get("LBFT_BENCH_CHILD"))
        here = traffic_measured_here(sys.argv[1:]) if (want_here and world == 1) else None
        here_ok = bool(here) and "gb_corrected" in here
        local_events = sum(c["events"])
        achieved = local_events * bpe / (k_ms * 1e-3) / 1e9
        ex_bytes, pops = executed_bytes(layout, c)
        achieved_ex = ex_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "simulated consensus rounds/sec (whole node), 65 536 x 4-node instances per GPU" if not strong else
                      "simulated consensus rounds/sec (whole node), %d x %d-node instances in total" % (total, args.nodes),
            "value": rounds / per_step, "unit": "rounds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "int32/u64 + f64 delay sampling", "data": "synthetic",
            "config": {"workload": "%d instances x %d nodes (f=1) %s, LogNormal(mean 10, variance 4) delays, max_clock %d, "
                                   "delta 20 gamma 2 lambda 0.5, seeds base+i; reference quirks Q1-Q6" % (
                                       total if strong else m, args.nodes, "in total" if strong else "per GPU", args.max_clock),
                       "instances_per_gpu": m, "total_instances": total, "nodes": args.nodes, "max_clock": args.max_clock,
                       "parallelism": "%s scaling: %s, %d ranks, contiguous instance shards, one counter all-gather" % (
                           "strong" if strong else "weak", "%d instances in total" % total if strong else "%d instances per GPU" % m, world)},
            "committed_blocks_per_s": commits / per_step,
            # what the device executes (queue pops) leads; the reference-equivalent total also counts the duplicate timers the device
            # folds at scheduling time (they never reach the queue)
            "queue_pops_per_s": (events - float(agg.get("timers_folded", 0))) / per_step,
            "events_per_s": events / per_step,
            "events_note": "queue_pops_per_s = events the device executes; events_per_s = reference-equivalent events (adds the duplicate timers folded at scheduling time)",
            "faulted_instances": faulted,
            # `achieved` / `frac`: the algorithmic bytes of what the device EXECUTES (folded duplicate timers never reach the queue;
            # cancelled timers and requests do not write node rows back) over the kernel's duration -- the honest fraction.  The
            # SURVEY 8(d) figure charged to every reference-equivalent event stands beside it as `frac_reference_equivalent`.
            "roofline": {"bound": "hbm", "achieved": achieved_ex, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_ex / HBM_PEAK_GBS,
                         "traffic": here["gb_corrected"] if here_ok else traffic["gb_corrected"] if traffic else None, "traffic_unit": "GB per launch",
                         # where `traffic` comes from: PMC counters cannot be read from inside this process, so the line REPLAYS the committed
                         # profile of the same kernels (stamped with the hash of their machine code; null when the stamp is stale)
                         "traffic_source": here["source"] if here_ok else
                                           ("committed profile %s (kernel hash %s): replayed, not measured in this run" % (traffic.get("profile"), traffic.get("source_hash")))
                                           if traffic else "none: no committed profile matches the built kernels",
                         "traffic_measured_here": here, "traffic_replayed_from_committed_profile": traffic["gb_corrected"] if traffic else None,
                         "traffic_detail": traffic, "kernel": run_kernel_name(layout.get("kernel_class", 0)), "kernel_ms": k_ms,
                         "algorithmic_gb_per_launch": ex_bytes / 1e9,
                         "queue_pops_per_launch": pops, "node_updates_per_launch": c.get("node_updates"),
                         "timers_folded_per_launch": c.get("timers_folded"), "queue_pops_per_s": pops / (k_ms * 1e-3),
                         "executed_le_traffic": (ex_bytes / 1e9 <= (here["gb_corrected"] if here_ok else traffic["gb_corrected"])) if (here_ok or traffic) else None,
                         "frac_reference_equivalent": achieved / HBM_PEAK_GBS,
                         "reference_equivalent": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "bytes_per_event": bpe,
                                                  "gb_per_launch": local_events * bpe / 1e9, "events_per_launch": local_events},
                         # (kept for readers of earlier rounds' lines: the same numbers under their old names)
                         "executed": {"achieved": achieved_ex, "frac": achieved_ex / HBM_PEAK_GBS, "gb_per_launch": ex_bytes / 1e9},
                         # the bound that binds ("report both ... events/s per CU"): instruction issue under divergence
                         "issue": dict(issue or {}, source=("committed profile %s (kernel hash %s): replayed, not measured in this run" % (issue.get("profile"), issue.get("source_hash")))
                                       if issue else "none: no committed profile matches the built kernels",
                                       pops_per_s_per_cu=pops / (k_ms * 1e-3) / 256.0,
                                       events_per_s_per_cu=local_events / (k_ms * 1e-3) / 256.0,
                                       note="VALU-issue / divergence bound: see DESIGN.md section 5; counters from the stamped PMC profile (null when stale)"),
                         "layout": layout},
        }
        # BASELINE config 3 (one batch of --instances instances, 1 -> N GPUs: strong scaling) beside the weak `value`; at N = 1 the two coincide
        if not strong:
            if strong_leg:
                ps = float(agg["extra_max"][1]) / args.steps
                sr, sc, se, sf = agg["extra_sum"]
                out["config3_strong"] = {"scaling": "strong", "total_instances": args.instances, "instances_per_gpu": strong_leg["instances_per_gpu"], "n_gpus": world,
                                         "ms_per_step": ps * 1e3, "value": sr / ps, "unit": "rounds/s", "committed_blocks_per_s": sc / ps, "events_per_s": se / ps,
                                         "faulted_instances": sf, "kernel": strong_leg["kernel"], "kernel_ms_rank0": strong_leg["kernel_ms"],
                                         "note": "the same %d seeds as ONE batch split over the ranks (contiguous shards), timed like `value`: W warmup + K steps between barriers, max over ranks" % args.instances}
            elif world == 1:
                out["config3_strong"] = {"scaling": "strong", "total_instances": m, "instances_per_gpu": m, "n_gpus": 1, "ms_per_step": per_step * 1e3, "value": rounds / per_step,
                                         "unit": "rounds/s", "note": "at one GPU the strong split of config 3 IS the weak batch: same numbers as `value`"}
        if native is not None:
            out["native_collective"] = native
        if args.parity_instances > 0:  # (rank 0's shard of the timed batch; oracle = test infrastructure, used only as the checker)
            out["parity"] = parity_gate(args, sim, res, seeds, args.nodes, args.max_clock)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, args.nodes, args.max_clock)
        print(json.dumps(out))
        if out["roofline"]["executed_le_traffic"] is False:  # the "algorithmic" bytes must be a lower bound of what crosses the L2 boundary
            sys.stderr.write("bench.py: executed algorithmic bytes %.2f GB exceed the measured traffic %.2f GB\n" % (ex_bytes / 1e9, out["roofline"]["traffic"]))
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(4)
        if native is not None and not native["matches_torch_aggregate"]:
            sys.stderr.write("bench.py: the native collective disagrees with the torch.distributed aggregate\n")
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(5)
        if out.get("parity", {}).get("mismatches", 0) != 0:
            sys.stderr.write("bench.py: PARITY GATE FAILED: %s\n" % json.dumps(out["parity"]))
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
