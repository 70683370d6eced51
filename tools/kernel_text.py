#!/usr/bin/env python3
"""Per-kernel fingerprints of the gfx950 machine code inside a built library (no GPU, no binutils): sha256 of every kernel's bytes in .text,
so that a source edit that is meant to leave a kernel alone (a deleted switch, a new sibling kernel) can be checked against the build before it:
    python tools/kernel_text.py [lib ...]            # one line per kernel
    python tools/kernel_text.py --diff old.so new.so # kernels whose machine code differs
(Calls to out-of-line device functions are PC-relative: a kernel whose distance to lbft_exp / lbft_log changes shows up as changed although its
instructions are the same; the instruction count printed beside the hash tells the two cases apart.)"""
import hashlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def code_object(path):
    blob = open(path, "rb").read()
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)

    def section(i):
        name, _t, _f, _a, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = section(shstrndx)
    strtab = blob[stroff:stroff + strsize]
    for i in range(shnum):
        name, off, size = section(i)
        if strtab[name:strtab.index(b"\0", name)] == b".hip_fatbin":
            fat = blob[off:off + size]
            n, = struct.unpack_from("<Q", fat, 24)
            pos = 32
            for _ in range(n):
                o, sz, ts = struct.unpack_from("<QQQ", fat, pos)
                pos += 24
                triple = fat[pos:pos + ts].decode()
                pos += ts
                if "gfx950" in triple:
                    return fat[o:o + sz]
    raise RuntimeError("no gfx950 code object in " + path)


def kernels(path):
    co = code_object(path)
    shoff, = struct.unpack_from("<Q", co, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", co, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _f, addr, off, size, link, _info, _al, entsize = struct.unpack_from("<IIQQQQIIQQ", co, shoff + i * shentsize)
        secs.append((name, typ, addr, off, size, link, entsize))
    out = {}
    for (name, typ, addr, off, size, link, entsize) in secs:
        if typ != 2:  # SHT_SYMTAB
            continue
        stroff = secs[link][3]
        for k in range(size // entsize):
            st_name, st_info, _o, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", co, off + k * entsize)
            if (st_info & 15) != 2 or st_size == 0 or st_shndx >= len(secs):  # STT_FUNC
                continue
            nm = co[stroff + st_name:co.index(b"\0", stroff + st_name)].decode()
            s_addr, s_off = secs[st_shndx][2], secs[st_shndx][3]
            body = co[s_off + st_value - s_addr:s_off + st_value - s_addr + st_size]
            out[nm] = (hashlib.sha256(body).hexdigest()[:16], st_size)
    return out


def main():
    args = sys.argv[1:]
    if args and args[0] == "--diff":
        a, b = kernels(args[1]), kernels(args[2])
        for nm in sorted(set(a) | set(b)):
            if a.get(nm) != b.get(nm):
                print("%-60s %s -> %s" % (nm[:60], a.get(nm), b.get(nm)))
        return
    for path in args or [os.path.join(ROOT, "librabft_simulator_amd", "liblbft_hip.so")]:
        for nm, (h, size) in sorted(kernels(path).items()):
            print("%-28s %-60s %s %7d B" % (os.path.basename(path), nm[:60], h, size))


if __name__ == "__main__":
    main()
