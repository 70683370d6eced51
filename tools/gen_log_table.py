#!/usr/bin/env python3
"""Regenerates the LBFT_LOG_DATA_INIT block of librabft_simulator_amd/csrc/lbft_math.h: the coefficients and the 128-entry table of
glibc's log() (struct __log_data of sysdeps/ieee754/dbl-64/e_log_data.c, i.e. ARM optimized-routines' log_data.c with LOG_TABLE_BITS 7),
read from the host's libm.so.6 -- the library whose results lbft_log must reproduce bit for bit.  The block is located by its first
table entry (invc = 0x1.734f0c3e0de9fp+0, logc = -0x1.7cc7f79e69000p-2); 18 doubles precede the table (ln2hi, ln2lo, 5 + 11 polynomial
coefficients).  Prints the block; lbft_math.h holds the output for glibc 2.35 (the same numbers since glibc 2.28)."""
import struct
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
blob = open(path, "rb").read()
first = struct.pack("<d", float.fromhex("0x1.734f0c3e0de9fp+0")) + struct.pack("<d", float.fromhex("-0x1.7cc7f79e69000p-2"))
at = blob.find(first)
assert at > 144, "glibc log table not found in " + path
vals = struct.unpack_from("<%dQ" % (18 + 256), blob, at - 144)
assert struct.unpack("<d", struct.pack("<Q", vals[7]))[0] == -0.5  # B[0]
print("#define LBFT_LOG_DATA_WORDS %d" % len(vals))
print("#define LBFT_LOG_DATA_INIT { \\")
for k in range(0, len(vals), 4):
    print("  " + ", ".join("0x%016xULL" % v for v in vals[k:k + 4]) + ", \\")
print("}")
