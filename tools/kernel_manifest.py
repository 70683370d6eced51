#!/usr/bin/env python3
"""The codegen manifest of the built library (no GPU): per kernel the bytes and sha256 of its machine code, VGPRs, SGPRs, spilled registers and scratch
bytes, plus the hipcc version and the library's kernel hash.  tests/golden/kernel_manifest.json is the committed copy; tests/test_abi.py fails when the
built library deviates from it -- performance here has depended on register-allocator luck (EXPERIMENTS.md round 5: deleting a dead struct member grew
lbft_k_run2l from 135 to 142 KB and cost 14 % on the device), so a change of any run kernel's machine code must be a deliberate, re-measured commit:
    python tools/kernel_manifest.py            # print the manifest of the built library and what differs from the committed one
    python tools/kernel_manifest.py --write    # regenerate tests/golden/kernel_manifest.json (same commit as the kernel edit, after measuring it)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
MANIFEST = os.path.join(ROOT, "tests", "golden", "kernel_manifest.json")


def short(name):
    """_Z12lbft_k_run0qN4lbft6ParamsEPjS1_ -> lbft_k_run0q; _Z10lbft_k_runILi2EEvN4... -> lbft_k_run<2>"""
    m = re.match(r"_Z\d+(lbft_k_[a-z_0-9]+?)(?:ILi(\d+)EE)?(?:v?N4lbft|P|j|N|v)", name)
    if not m:
        return name
    return m.group(1) + ("<%s>" % m.group(2) if m.group(2) else "")


def hipcc_version():
    try:
        out = subprocess.check_output(["/opt/rocm/bin/hipcc", "--version"], stderr=subprocess.STDOUT).decode()
        hip = re.search(r"HIP version:\s*(\S+)", out)
        clang = re.search(r"clang version\s*(\S+)", out)
        return "HIP %s / clang %s" % (hip.group(1) if hip else "?", clang.group(1) if clang else "?")
    except Exception as e:  # noqa
        return "unknown (%s)" % type(e).__name__


def manifest(so_path=None):
    from kernel_text import kernels
    from librabft_simulator_amd import build
    from test_abi import _kernel_metadata
    so_path = so_path or build.OUT
    text = kernels(so_path)
    meta = _kernel_metadata(so_path)
    out = {}
    for name, m in meta.items():
        if "lbft_k_" not in name:
            continue
        h, size = text.get(name, ("?", 0))
        out[short(name)] = {"text_bytes": size, "text_sha256_16": h, "vgprs": m["vgpr_count"], "sgprs": m.get("sgpr_count"), "spilled_vgprs": m["vgpr_spill_count"],
                            "scratch_bytes": m["private_segment_fixed_size"]}
    return {"hipcc": hipcc_version(), "kernel_hash": build.kernel_hash(so_path), "flags": " ".join(build.HIPCC_FLAGS), "kernels": dict(sorted(out.items()))}


def diff(a, b):
    """Human-readable differences between two manifests (committed a, built b)."""
    lines = []
    if a.get("hipcc") != b.get("hipcc"):
        lines.append("hipcc: %s -> %s" % (a.get("hipcc"), b.get("hipcc")))
    if a.get("flags") != b.get("flags"):
        lines.append("flags: %s -> %s" % (a.get("flags"), b.get("flags")))
    ka, kb = a.get("kernels", {}), b.get("kernels", {})
    for k in sorted(set(ka) | set(kb)):
        if ka.get(k) != kb.get(k):
            lines.append("%-24s %s -> %s" % (k, json.dumps(ka.get(k)), json.dumps(kb.get(k))))
    return lines


def main():
    m = manifest()
    if "--write" in sys.argv:
        with open(MANIFEST, "w") as f:
            json.dump(m, f, indent=1, sort_keys=True)
            f.write("\n")
        print("wrote", MANIFEST)
        return
    print(json.dumps(m, indent=1, sort_keys=True))
    if os.path.exists(MANIFEST):
        d = diff(json.load(open(MANIFEST)), m)
        print("\n".join(["differs from the committed manifest:"] + d) if d else "== committed manifest")


if __name__ == "__main__":
    main()
