#!/usr/bin/env python3
"""Runs BASELINE.json's configurations 2-5 on one GPU (the HIP path through the C ABI) and prints one JSON line each:
kernel time, rounds/s, events/s, faults, queue high-water mark, HBM footprint.  Configurations 4 and 5 use this
framework's extensions (equivocating leaders; weighted voting rights with reference quirk semantics Q1/Q2), for which
the reference has no answer: oracle/lbft_oracle.cpp is their specification."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c2_1024x4_lognormal": dict(instances=1024, nodes=4, max_clock=1000),
    "c2_1024x4_uniform": dict(instances=1024, nodes=4, max_clock=1000, uniform=(5, 15)),
    "c3_65536x4": dict(instances=65536, nodes=4, max_clock=1000),
    # what each of 8 GPUs runs when configuration 3 is read as ONE 65 536-instance batch sharded over the node (strong scaling)
    "c3shard_8192x4": dict(instances=8192, nodes=4, max_clock=1000),
    "c4_16384x64_longtail_equivocators": dict(instances=16384, nodes=64, max_clock=300, variance=400.0, equivocate_every=5),
    "c5_8192x100_weighted_epochs": dict(instances=8192, nodes=100, max_clock=300, weights=[1 + (i % 4) for i in range(100)],
                                        commands_per_epoch=50),
    # config 5 at max_clock 300 never reaches its first epoch change (~8 commits); this variant crosses epochs: the "fixed"
    # protocol mode (quirks Q1 and Q2 fixed -- the reference semantics stall at the first change), an epoch every 10
    # commands, and the voting rights rotating by one node per epoch (rights_rotation, the epoch-reconfiguration extension)
    "c5b_1024x100_rotating_rights_epochs_fixed": dict(instances=1024, nodes=100, max_clock=600, weights=[1 + (i % 4) for i in range(100)],
                                                      commands_per_epoch=10, quirks=3, rights_rotation=1),
    # Configurations 4 and 5 as SURVEY.md 8(d) wrote them are degenerate (round-1 verdict): with the reference's quirk Q1 a node that
    # misses a certificate can never catch up, so under long-tailed delays stragglers accumulate until fewer than 2f+1 nodes are
    # current and the network stops committing (C4: 0 commits by clock 300, 4 by clock 3000 at the best nodes); C5 never reaches
    # its first epoch change.  The "live" variants exercise what the configurations are named for, at full size, in the fixed
    # protocol mode (quirks = 3: requests answered by the peer, EpochId::previous() = id - 1):
    #   c4live: 13 of 64 nodes equivocating under LogNormal(10, 400) delays -- every node of every instance commits >= 20 blocks by clock 1000;
    #   c5live: weighted voting rights rotating by one node per epoch, an epoch every 3 commands -- >= 2 epoch changes per node by clock 450.
    "c4live_16384x64_longtail_equivocators_fixed": dict(instances=16384, nodes=64, max_clock=1000, variance=400.0, equivocate_every=5, quirks=3),
    "c5live_8192x100_rotating_rights_epochs_fixed": dict(instances=8192, nodes=100, max_clock=450, weights=[1 + (i % 4) for i in range(100)],
                                                         commands_per_epoch=3, quirks=3, rights_rotation=1),
    # BASELINE configuration 5 AS IT IS NAMED (round-5 review, missing #1): 8 192 x 100 nodes, voting rights 1 + (i mod 4), an epoch every 50 commits
    # with the reconfiguration actually firing -- the epoch switch of librabft-v2/src/node.rs:331-348 and the configuration read of
    # bft-lib/src/simulated_context.rs:199-216 (here: the rights rotate by one node per epoch, the epoch-reconfiguration extension) -- in the fixed
    # protocol mode (the reference semantics stall at the first change), run until EVERY node of EVERY instance has changed epoch
    # at least once: ~33 commits per 1 000 ticks, the first change around clock 1 600-1 900, clock 2 500 leaves a margin.
    "c5named_8192x100_weighted_epoch_every_50_commits": dict(instances=8192, nodes=100, max_clock=2500, weights=[1 + (i % 4) for i in range(100)],
                                                             commands_per_epoch=50, quirks=3, rights_rotation=1),
}
HBM_PEAK_GBS = 8000.0
# Share of a configuration's notifications / responses that the cooperative runs retire -- such an event writes its node's timer words, not its row (roofline()).
# Since the round's last session COUNTED ON THE DEVICE, every instance of the full-size run (liblbft_hip_runcount.so = -DLBFT_PHASE_TIMERS -DLBFT_RUN_COUNT, LDS counters
# per wavefront; tools/gpu_r06_run_counts.sh -> profiles/r06/run_counts_on_the_device.jsonl: retired events / the kind's events, same counters and commits as the
# product library in the same call).  The host build of the kernel logic (64-lane segments, one or two instances: tests/tools/run_stats.cpp ->
# profiles/r06/cooperative_runs_final_c5_c4_c4live_c5live_c5named.jsonl) had given 0.8283 / 0.8763, 0.9618 / 0.9952, 0.8870 / 0.8201, 0.9506 / 0.9941, 0.9678 / 0.9927.
RUN_SHARES = {
    "c4_16384x64_longtail_equivocators": {"notify": 83467987 / 100539014, "response": 119600086 / 136261613},                   # 0.8302 / 0.8777
    "c5_8192x100_weighted_epochs": {"notify": 540324867 / 563063241, "response": 575621584 / 578109344},                       # 0.9596 / 0.9957
    "c4live_16384x64_longtail_equivocators_fixed": {"notify": 512370218 / 574990270, "response": 337271225 / 411852790},       # 0.8911 / 0.8189
    "c5live_8192x100_rotating_rights_epochs_fixed": {"notify": 649765811 / 684296122, "response": 711763478 / 716057659},      # 0.9495 / 0.9940
    "c5named_8192x100_weighted_epoch_every_50_commits": {"notify": 5392660624 / 5581414146, "response": 6508032751 / 6556394091},  # 0.9662 / 0.9926
}
OPT_IN = ("c5b", "c4live", "c5live", "c5named")
# What pins each configuration's results (printed with every line): the reference itself only holds answers for 3- / 8-node
# LogNormal(10, 4) runs; everything else is "device == oracle", the oracle being the specification.
PARITY = {
    "c2_1024x4_lognormal": "oracle == reference goldens (3/8 nodes, LogNormal(10,4)); 4 nodes: oracle-as-spec, same code path",
    "c3_65536x4": "oracle == reference goldens (3/8 nodes, LogNormal(10,4)); 4 nodes: oracle-as-spec, same code path",
    "c3shard_8192x4": "oracle == reference goldens (3/8 nodes, LogNormal(10,4)); 4 nodes: oracle-as-spec, same code path",
}
PARITY_DEFAULT = "oracle-as-spec (no reference answer: extension / size the reference never ran)"
# ... and how much of each full-size batch the test suite compares with the oracle (tests/test_gpu_parity.py::test_full_size_*,
# test_full_batch_math_mode_0_all_262144_nodes; bench.py's gate re-runs 1 024 instances of every timed batch)
PARITY_SAMPLE = {
    "c2_1024x4_lognormal": "all 1 024 instances", "c2_1024x4_uniform": "all 1 024 instances",
    "c3_65536x4": "all 65 536 instances (262 144 nodes), math_mode 0; 1 024 in every bench line",
    "c3shard_8192x4": "the first 8 192 instances of the c3 check",
    "c4_16384x64_longtail_equivocators": "all 16 384 instances in every suite run (committed oracle digests, computed on two machines) + 64 live on the oracle",
    "c5_8192x100_weighted_epochs": "all 8 192 instances in every suite run (committed oracle digests, tests/golden/full_size_digests.npz) + 64 live on the oracle",
    "c4live_16384x64_longtail_equivocators_fixed": "all 16 384 instances in every suite run (committed oracle digests) + 128 live on the oracle",
    "c5live_8192x100_rotating_rights_epochs_fixed": "all 8 192 instances in every suite run (committed oracle digests) + 64 live on the oracle",
    "c5named_8192x100_weighted_epoch_every_50_commits": "the first 3 584 of 8 192 instances in every suite run (committed oracle digests: 49 core-seconds of oracle time per instance, computed on the GPU box's host and on the build container) + 16 live",
}
# (round 6: every instance the oracle has a digest for in tests/golden/full_size_digests.npz is compared in every suite run -- tests/full_size_digest.py)
# What a line measures, where that is not what its name suggests (printed with the line)
NOTES = {
    "c4_16384x64_longtail_equivocators": "DEGENERATE as SURVEY 8(d) wrote it: under reference quirk Q1 stragglers never catch up and the network stops committing "
                                         "(0 commits by clock 300): this line times a stalled protocol; c4live is the configuration exercising what it is named for",
    "c5_8192x100_weighted_epochs": "DEGENERATE as SURVEY 8(d) wrote it: clock 300 ends before the first epoch change (50 commands per epoch, ~8 commits): no "
                                   "reconfiguration is exercised; c5live is",
}


def kernel_name(layout):
    k = layout["kernel_class"]
    cls, lean = k & 255, bool(k & 1024)
    if cls == 0:
        return "lbft_k_run0u" if k & 32768 else "lbft_k_run0s" if k & 8192 else "lbft_k_run0q" if k & 16384 else "lbft_k_run0"
    if lean:
        return ("lbft_k_run2q" if k & 4096 else "lbft_k_run2l") if cls == 2 else "lbft_k_run1l"
    return "lbft_k_run<%d>" % cls


def measured_traffic(name):
    """HBM bytes per launch of this configuration's run kernel from its own PMC passes (tools/gpu_configs_profile.sh writes
    <dir>/<config>.pmc.json; LBFT_PMC_DIR points at that directory): 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes."""
    d = os.environ.get("LBFT_PMC_DIR")
    if not d:
        return None
    try:
        with open(os.path.join(d, name + ".pmc.json")) as f:
            t = json.load(f)
        return {"gb_per_launch": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / 1e9, "fetch_kb_raw": t["FETCH_SIZE"], "write_kb_raw": t["WRITE_SIZE"]}
    except Exception:
        return None


def roofline(layout, k, kernel_ms, name=None):
    """SURVEY.md 8(d) per configuration, both ways (bench.py: all reference-equivalent events / rows the device moves).  The sizes are what
    one event moves (lbft_batch_layout: node burst = fixed words + set extension words; hcbr words of large networks not counted), so the
    algorithmic bytes are a lower bound of the traffic; `traffic` (per-configuration PMC pass) must not be below `executed`."""
    ev = sum(k["events"])
    s_node, s_evt, s_notif = layout["node_bytes"], layout["event_bytes"], layout["snapshot_bytes"]
    p = k["events_scheduled"] / max(ev, 1)
    r = k["events"][0] / max(ev, 1)
    bpe = 2 * s_node + s_evt * (1 + p) + s_notif * 2 * r
    pops = ev - k.get("timers_folded", 0)
    # (round 6) outside kernel class 0, a request under the reference's quirk Q1 -- answered by the requester itself with a payload-free response -- no longer
    # fetches the node's rows: those pops move an event entry and nothing else
    cls = layout["kernel_class"] & 255
    q1 = bool(CONFIGS.get(name, {}).get("quirks", 0) & 1) if name else bool(layout["kernel_class"] & 4096)
    node_reads = pops - (k["events"][1] if (cls != 0 and not q1) else 0)
    # ... and a response under the reference's semantics inserts nothing: its update_node is a no-op on a settled node (the response runs prove it for 88-99.5 % of
    # them, coop_responses), what is written back is the node's four timer words (16 B), not its row
    upd = k.get("node_updates", pops)
    noop_resp = min(k["events"][2], upd) if (cls != 0 and not q1) else 0
    # ... (third session of round 6) and so do the events the response / notification runs retire in every mode: a run's event reads its node's row (and a
    # notification its snapshot's fixed words), proves that handler + update_node leave the node as it was -- or only add to its current timeouts / ballot --
    # and writes the four timer words (+ a delta of a few words, not counted: a lower bound).  The product kernels carry no counter for them; the shares are
    # counted on the device by the diagnostic build of the same logic (RUN_SHARES above)
    sh = RUN_SHARES.get(name)
    noop = min(int(sh["notify"] * k["events"][0] + sh["response"] * k["events"][2]), upd) if (sh and cls != 0) else noop_resp
    ex_rows = node_reads * s_node + (upd - noop_resp) * s_node + noop_resp * 16 + 2 * pops * s_evt + 2 * k["events"][0] * s_notif  # (the model before the runs of notifications)
    ex = node_reads * s_node + (upd - noop) * s_node + noop * 16 + 2 * pops * s_evt + 2 * k["events"][0] * s_notif
    sec = kernel_ms * 1e-3
    t = measured_traffic(name) if name else None
    # `frac` = the algorithmic bytes of what the device EXECUTES over the kernel time (as bench.py since round 4); the SURVEY 8(d) figure
    # charged to every reference-equivalent event stands beside it
    out = {"bound": "hbm", "kernel": kernel_name(layout), "kernel_ms": kernel_ms, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "algorithmic_gb_per_launch": ex / 1e9, "achieved": ex / sec / 1e9, "frac": ex / sec / 1e9 / HBM_PEAK_GBS,
           "queue_pops": pops, "node_row_reads": node_reads, "node_updates": k.get("node_updates"), "pops_per_s_per_cu": pops / sec / 256.0,
           "frac_reference_equivalent": ev * bpe / sec / 1e9 / HBM_PEAK_GBS,
           "reference_equivalent": {"bytes_per_event": bpe, "gb_per_launch": ev * bpe / 1e9, "achieved": ev * bpe / sec / 1e9,
                                    "frac": ev * bpe / sec / 1e9 / HBM_PEAK_GBS},
           "executed": {"gb_per_launch": ex / 1e9, "achieved": ex / sec / 1e9, "frac": ex / sec / 1e9 / HBM_PEAK_GBS, "queue_pops": pops,
                        "node_updates": k.get("node_updates"), "updates_retired_in_runs_writing_timer_words_only": noop,
                        "frac_if_every_update_wrote_its_row": ex_rows / sec / 1e9 / HBM_PEAK_GBS},
           "traffic": t["gb_per_launch"] if t else None, "traffic_unit": "GB per launch (2 x FETCH_SIZE + WRITE_SIZE)", "traffic_detail": t}
    if t:
        out["traffic_over_reference_equivalent"] = t["gb_per_launch"] / (ev * bpe / 1e9)
        out["traffic_over_executed"] = t["gb_per_launch"] / (ex / 1e9)
        out["executed_le_traffic"] = ex / 1e9 <= t["gb_per_launch"]
        out["traffic_frac_of_peak"] = t["gb_per_launch"] / sec / HBM_PEAK_GBS
    return out


def run(name, scale=1.0, reps=1, lpw=0):
    import numpy as np
    from librabft_simulator_amd import BatchSimulator, NodeConfig, RandomDelay
    c = CONFIGS[name]
    m = max(int(c["instances"] * scale), 1)
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    delay = RandomDelay.uniform(*c["uniform"]) if "uniform" in c else RandomDelay.new(10.0, c.get("variance", 4.0))
    sim = BatchSimulator.new(seeds, c["nodes"], delay, NodeConfig(), commands_per_epoch=c.get("commands_per_epoch", 30000),
                             voting_rights=c.get("weights"), equivocate_every=c.get("equivocate_every", 0), lanes_per_wavefront=lpw, quirks=c.get("quirks", 0),
                             rights_rotation=c.get("rights_rotation", 0))
    best = None
    for _ in range(reps):
        sim.reset()
        res = sim.loop_until(c["max_clock"], allow_faults=True)
        ms = sim.last_run_ms()[1]
        best = ms if best is None else min(best, ms)
    k = res.counters
    cc = res.commit_counts
    worst = cc.min(axis=1)
    liveness = {"min_node_commits": {"min": int(worst.min()), "median": float(np.median(worst)), "max": int(worst.max())},
                "instances_with_5_commits_at_every_node": float((worst >= 5).mean()),
                "epochs_min_max": [int(res.epochs.min()), int(res.epochs.max())]}
    out = {"config": name, "note": NOTES.get(name), "parity": PARITY.get(name, PARITY_DEFAULT), "parity_sample": PARITY_SAMPLE.get(name), "liveness": liveness, "roofline": roofline(sim.layout(), k, best, name), "instances": m, "nodes": c["nodes"], "max_clock": c["max_clock"], "kernel_ms": best,
           "rounds_per_s": k["rounds"] / (best * 1e-3), "commits_per_s": k["commits"] / (best * 1e-3),
           "events_per_s": sum(k["events"]) / (best * 1e-3), "events": sum(k["events"]),
           "events_by_kind": {"notify": k["events"][0], "request": k["events"][1], "response": k["events"][2], "timer": k["events"][3]}, "rounds": k["rounds"], "commits": k["commits"],
           "faulted_instances": k["faulted_instances"], "max_queue": k["max_queue"], "max_snapshots": k["max_snapshots"],
           "device_gb": sim.device_bytes() / 1e9, "layout": sim.layout()}
    try:  # diagnostic builds (LBFT_HIP_LIB=.../liblbft_hip_prof.so): share of wavefront cycles per phase of the event loop
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from sweep import COUNTS, PHASES
        pc = sim.phase_cycles()
        if os.environ.get("LBFT_RUN_COUNT"):  # liblbft_hip_runcount.so (-DLBFT_PHASE_TIMERS -DLBFT_RUN_COUNT): what the cooperative runs retire ON THE DEVICE
            ev = k["events"]
            out["runs_device"] = {"retired": {"request": int(pc[0]), "response": int(pc[1]), "notify": int(pc[2])},
                                  "runs": {"request": int(pc[3]), "response": int(pc[4]), "notify": int(pc[5])},
                                  "share_of_kind": {"request": int(pc[0]) / max(ev[1], 1), "response": int(pc[1]) / max(ev[2], 1), "notify": int(pc[2]) / max(ev[0], 1)},
                                  "events_per_run": {"request": int(pc[0]) / max(int(pc[3]), 1), "response": int(pc[1]) / max(int(pc[4]), 1), "notify": int(pc[2]) / max(int(pc[5]), 1)},
                                  "wave_steps": int(pc[30])}
            raise StopIteration
        tot = float(sum(int(pc[i]) for i in range(30) if i not in COUNTS)) or 1.0
        out["phases"] = {PHASES.get(i, str(i)): round(int(pc[i]) / tot, 4) for i in range(30) if int(pc[i]) and i not in COUNTS}
        out["cycles_per_wave_step"] = int(pc[31]) / max(int(pc[30]), 1)
        out["wave_steps"] = int(pc[30])
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    # c5b / c4live / c5live are opt-in: seconds to a minute of GPU time per run and up to ~100 GB of device state
    ap.add_argument("names", nargs="*", default=[n for n in CONFIGS if not n.startswith(OPT_IN)])
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the configuration's instance count")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--lpw", type=int, default=0, help="lanes per wavefront carrying an instance (0 = auto)")
    a = ap.parse_args()
    for n in a.names:
        run(n, a.scale, a.reps, a.lpw)
