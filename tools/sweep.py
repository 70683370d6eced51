#!/usr/bin/env python3
"""GPU tuning sweep (run on the GPU box via gpurun): kernel time of the run kernel for the bench workload
over (library variant, lanes per wavefront, LDS queue slots), plus the phase-timer breakdown of diagnostic
builds.  Prints one JSON line per configuration."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = {0: "pop", 1: "node_load", 2: "timer_pre", 3: "snap_release", 4: "request", 5: "response", 6: "pacemaker",
          7: "timeout_propose", 8: "vote", 9: "new_qc", 10: "commits_tracker", 11: "sync_push", 12: "pna_timer",
          13: "pna_notify", 14: "pna_query", 15: "end_node", 16: "send_prep", 17: "store_drain", 18: "send_sample", 19: "send_push", 27: "commits", 28: "pre_blk_miss", 29: "blk_miss", 20: "hn_hcc", 21: "response_walk", 22: "hn_block",
          23: "hn_timeouts", 24: "hn_vote"}
COUNTS = {25: "blk_miss_sites_per_step", 26: "blk_get_sites_per_step"}


def one(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    from librabft_simulator_amd import BatchSimulator, NodeConfig, RandomDelay, lib
    seeds = np.arange(1, args.instances + 1, dtype=np.uint64)
    if args.same_seed:  # divergence-free upper bound: every lane of a wavefront simulates the same network
        seeds = np.full(args.instances, 12345, dtype=np.uint64)
    elif args.seed_groups:  # every group of `seed_groups` consecutive instances shares a seed
        seeds = (np.arange(args.instances, dtype=np.uint64) // np.uint64(args.seed_groups)) + np.uint64(1)
    sim = BatchSimulator.new(seeds, args.nodes, RandomDelay.new(10.0, 4.0), NodeConfig(), lanes_per_wavefront=args.lpw,
                             lds_queue_slots=args.ql)
    ms = []
    res = None
    for _ in range(args.reps + 1):
        sim.reset()
        res = sim.loop_until(args.max_clock)
        ms.append(sim.last_run_ms()[1])
    c = res.counters
    out = {"lib": os.path.basename(os.environ.get("LBFT_HIP_LIB", "liblbft_hip.so")), "info": lib().lbft_build_info().decode(),
           "lpw": args.lpw, "ql": args.ql, "instances": args.instances, "nodes": args.nodes, "max_clock": args.max_clock,
           "kernel_ms": min(ms[1:]), "kernel_ms_all": ms, "events": sum(c["events"]), "rounds": c["rounds"],
           "events_per_s": sum(c["events"]) / (min(ms[1:]) * 1e-3), "faulted": c["faulted_instances"], "max_queue": c["max_queue"]}
    try:
        pc = sim.phase_cycles()
        tot = float(sum(int(pc[k]) for k in range(30) if k not in COUNTS)) or 1.0
        out["phases"] = {PHASES.get(k, str(k)): round(int(pc[k]) / tot, 4) for k in range(30) if int(pc[k]) and k not in COUNTS}
        out["counts"] = {v: round(int(pc[k]) / max(int(pc[30]), 1), 3) for k, v in COUNTS.items()}
        out["cycles_per_wave_step"] = int(pc[31]) / max(int(pc[30]), 1)
        out["wave_steps"] = int(pc[30])
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one", action="store_true")
    ap.add_argument("--lpw", type=int, default=0)
    ap.add_argument("--ql", type=int, default=-1)
    ap.add_argument("--instances", type=int, default=65536)
    ap.add_argument("--nodes", type=int, default=4)
    ap.add_argument("--max-clock", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--same-seed", action="store_true")
    ap.add_argument("--seed-groups", type=int, default=0)
    ap.add_argument("--libs", default="liblbft_hip.so")
    ap.add_argument("--grid", default="64:-1,32:-1,16:-1,64:0,32:0")
    args = ap.parse_args()
    if args.one:
        return one(args)
    for libname in args.libs.split(","):
        path = os.path.join(ROOT, "librabft_simulator_amd", libname)
        if not os.path.exists(path):
            print(json.dumps({"lib": libname, "error": "missing"}), flush=True)
            continue
        for item in args.grid.split(","):
            lpw, ql = item.split(":")
            env = dict(os.environ, LBFT_HIP_LIB=path)
            cmd = [sys.executable, os.path.abspath(__file__), "--one", "--lpw", lpw, "--ql", ql, "--instances", str(args.instances),
                   "--nodes", str(args.nodes), "--max-clock", str(args.max_clock), "--reps", str(args.reps)] + (["--same-seed"] if args.same_seed else []) + (["--seed-groups", str(args.seed_groups)] if args.seed_groups else [])
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            sys.stdout.write(r.stdout)
            if r.returncode != 0:
                print(json.dumps({"lib": libname, "lpw": lpw, "ql": ql, "error": r.stderr[-400:]}), flush=True)


if __name__ == "__main__":
    main()
