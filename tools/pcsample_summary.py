#!/usr/bin/env python3
"""Aggregate rocprofv3 PC-sampling output (csv) of the run kernel: samples per source line (the innermost inlined location the
-gline-tables-only build attaches to each instruction), per instruction, and -- stochastic method -- per stall reason; plus the
average number of active lanes per sampled instruction (popcount of the exec mask).
    python tools/pcsample_summary.py <rocprofv3 output dir> <out.json>"""
import collections
import csv
import glob
import json
import sys

csv.field_size_limit(1 << 30)


def popcount(x):
    return bin(x).count("1")


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = {"files": [], "rows": 0}
    by_line = collections.Counter()
    by_line_lanes = collections.Counter()
    by_inst = collections.Counter()
    by_inst_lanes = collections.Counter()
    by_stall = collections.Counter()
    by_type = collections.Counter()
    issued = collections.Counter()
    by_line_stall = collections.defaultdict(collections.Counter)
    head = []
    for path in sorted(glob.glob(root + "/**/*pc_sampling*.csv", recursive=True)):
        res["files"].append(path)
        with open(path, newline="") as f:
            rd = csv.DictReader(f)
            res.setdefault("columns", rd.fieldnames)
            for row in rd:
                res["rows"] += 1
                if len(head) < 40:
                    head.append(row)
                inst = row.get("Instruction", "")
                com = row.get("Instruction_Comment", "")
                em = row.get("Exec_Mask", "0")
                try:
                    lanes = popcount(int(em, 0) if isinstance(em, str) and em.startswith("0x") else int(em))
                except ValueError:
                    lanes = 0
                key = com or "?"
                by_line[key] += 1
                by_line_lanes[key] += lanes
                ik = (com + " | " + inst)
                by_inst[ik] += 1
                by_inst_lanes[ik] += lanes
                st = row.get("Stall_Reason")
                if st is not None:
                    by_stall[st] += 1
                    by_line_stall[key][st] += 1
                ty = row.get("Instruction_Type")
                if ty is not None:
                    by_type[ty] += 1
                wi = row.get("Wave_Issued_Instruction")
                if wi is not None:
                    issued[wi] += 1
    n = max(res["rows"], 1)
    res["head"] = head
    res["stall"] = dict(by_stall.most_common())
    res["type"] = dict(by_type.most_common())
    res["issued"] = dict(issued)
    res["lines"] = [{"where": k, "samples": v, "frac": round(v / n, 5), "lanes": round(by_line_lanes[k] / v, 2),
                     "stall": dict(by_line_stall[k].most_common(4)) if k in by_line_stall else None} for k, v in by_line.most_common(400)]
    res["insts"] = [{"where": k, "samples": v, "frac": round(v / n, 5), "lanes": round(by_inst_lanes[k] / v, 2)} for k, v in by_inst.most_common(1500)]
    json.dump(res, open(out, "w"), indent=0)
    print("rows", res["rows"], "files", len(res["files"]), "columns", res.get("columns"))
    print("stall", res["stall"], "issued", res["issued"])
    for e in res["lines"][:40]:
        print(e["samples"], e["frac"], e["lanes"], e["where"][-90:])


if __name__ == "__main__":
    main()
