#!/usr/bin/env python3
"""Round-5 review item 4: the HBM traffic of the large-network kernels accounted for BY STRUCTURE and reconciled with the counters.

    python tools/traffic_reconcile.py profiles/r06/mem_lines_final.jsonl profiles/r06/baseline_configs.jsonl

Left: tests/tools/mem_lines.cpp -- a host build of the kernel logic that records every access of the event loop to an instance's rows and counts, per
processed event, the DISTINCT 128-byte lines read and the distinct 64-byte halves written, per structure (the model: nothing survives in a cache between
two events of a network -- thousands of other networks run in between --, everything inside one event does).  Right: rocprofv3's FETCH_SIZE / WRITE_SIZE
of the same configuration at full size (tools/gpu_configs_profile.sh, separate --pmc passes; bytes = 2 x FETCH_SIZE KB resp. WRITE_SIZE KB as
MI355X_MICROARCH.md prescribes for gfx950).  Prints one line per configuration: model and measured GB, their ratio, the executed (algorithmic) bytes
beside them, and the three structures that cost the most lines."""
import json
import sys

SHORT = {"c4": "c4_16384x64_longtail_equivocators", "c5": "c5_8192x100_weighted_epochs", "c4live": "c4live_16384x64_longtail_equivocators_fixed",
         "c5live": "c5live_8192x100_rotating_rights_epochs_fixed", "c5named": "c5named_8192x100_weighted_epoch_every_50_commits"}


def main():
    model = {}
    for l in open(sys.argv[1]):
        d = json.loads(l)
        model[SHORT.get(d["config"], d["config"])] = d
    out = []
    for l in open(sys.argv[2]):
        d = json.loads(l)
        m = model.get(d["config"])
        r = d["roofline"]
        t = r.get("traffic_detail")
        if not m or not t:
            continue
        pops = r["queue_pops"]
        mf, mw = m["model_bytes_per_pop"]["fetch_128B_lines"] * pops / 1e9, m["model_bytes_per_pop"]["write_64B"] * pops / 1e9
        cf, cw = 2 * t["fetch_kb_raw"] * 1024 / 1e9, t["write_kb_raw"] * 1024 / 1e9
        top = sorted(m["by_structure"], key=lambda s: -(s["lines_read_per_pop"] * 128 + s["lines_written_per_pop"] * 64))[:3]
        row = {"config": d["config"], "kernel": r["kernel"], "kernel_ms": round(d["kernel_ms"], 1), "queue_pops": pops,
               "model_gb": {"fetched": round(mf, 1), "written": round(mw, 1), "total": round(mf + mw, 1)},
               "counters_gb": {"fetched_2x_FETCH_SIZE": round(cf, 1), "written_WRITE_SIZE": round(cw, 1), "total": round(cf + cw, 1)},
               "counters_over_model": {"fetched": round(cf / mf, 3), "written": round(cw / mw, 3), "total": round((cf + cw) / (mf + mw), 3)},
               "executed_gb": round(r["algorithmic_gb_per_launch"], 1), "traffic_over_executed": round((cf + cw) / r["algorithmic_gb_per_launch"], 2),
               "frac_executed": round(r["frac"], 4),
               "largest_structures_bytes_per_pop": {s["structure"]: round(s["lines_read_per_pop"] * 128 + s["lines_written_per_pop"] * 64) for s in top}}
        out.append(row)
        print(json.dumps(row))
    return 0 if out else 1


if __name__ == "__main__":
    sys.exit(main())
