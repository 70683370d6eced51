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


def kernel_hash(so_path=None):
    """sha256 (16 hex digits) of the gfx950 code object inside the built library: profiles/current/pmc_traffic.json is stamped
    with it, and bench.py only reports that traffic while the stamp matches the kernels it is running (host-side edits of the C
    ABI do not change it; any kernel edit does).  Pure-Python ELF / offload-bundle parsing: no binutils needed on the GPU box."""
    import hashlib
    import struct
    blob = open(so_path or OUT, "rb").read()
    assert blob[:4] == b"\x7fELF" and blob[4] == 2, "not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)

    def section(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = section(shstrndx)
    strtab = blob[stroff:stroff + strsize]
    fat = None
    for i in range(shnum):
        name, off, size = section(i)
        if strtab[name:strtab.index(b"\0", name)] == b".hip_fatbin":
            fat = blob[off:off + size]
    assert fat is not None and fat.startswith(b"__CLANG_OFFLOAD_BUNDLE__"), "no offload bundle in the library"
    n, = struct.unpack_from("<Q", fat, 24)
    pos = 32
    for _ in range(n):
        o, sz, ts = struct.unpack_from("<QQQ", fat, pos)
        pos += 24
        triple = fat[pos:pos + ts].decode()
        pos += ts
        if "gfx950" in triple:
            co = fat[o:o + sz]
            # the machine code only (.text of the code object): the object as a whole also carries a per-translation-unit id
            # (__hip_cuid_*) that changes with any edit of the file, host code included
            cshoff, = struct.unpack_from("<Q", co, 0x28)
            centsize, cnum, cstrndx = struct.unpack_from("<HHH", co, 0x3A)

            def csection(i):
                name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", co, cshoff + i * centsize)
                return name, off, size
            _, cstroff, cstrsize = csection(cstrndx)
            cstr = co[cstroff:cstroff + cstrsize]
            for i in range(cnum):
                name, off, size = csection(i)
                if cstr[name:cstr.index(b"\0", name)] == b".text":
                    return hashlib.sha256(co[off:off + size]).hexdigest()[:16]
            raise RuntimeError("code object without .text")
    raise RuntimeError("no gfx950 code object in the library")


def source_hash():
    """(kept as the stamp's name in profiles/ and bench.py) = kernel_hash() of the built library."""
    return kernel_hash()


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
