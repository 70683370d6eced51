"""Build liblbft_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m librabft_simulator_amd.build` or `build()`; the .so is git-ignored but travels to the GPU
box with the gpurun snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "lbft_hip.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("lbft_core.h", "lbft_math.h", "lbft_tables.h", "lbft_save_node.h")] + [
    os.path.join(HERE, "..", "include", "lbft.h")]
OUT = os.path.join(HERE, "liblbft_hip.so")

# -ffp-contract=off: Rust never fuses; every fused multiply-add in lbft_math.h is explicit.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the HIP library (there is no CPU fallback)")


def source_hash():
    """sha256 (16 hex digits) of the kernel sources: profiles/current/pmc_traffic.json is stamped with it, and bench.py only
    reports that traffic while the stamp matches the sources the library was built from."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS[:4]:  # lbft_hip.hip, lbft_core.h, lbft_math.h, lbft_tables.h
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return OUT
    cmd = [hipcc_path()] + HIPCC_FLAGS + [SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


def build_variant(tag, defines, verbose=False):
    """Diagnostic / tuning builds of the same sources (e.g. ("prof", ["-DLBFT_PHASE_TIMERS"])): written next to
    the product library as liblbft_hip_<tag>.so and selected with LBFT_HIP_LIB=<path>."""
    out = os.path.join(HERE, "liblbft_hip_%s.so" % tag)
    cmd = [hipcc_path()] + HIPCC_FLAGS + list(defines) + [SRC, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
