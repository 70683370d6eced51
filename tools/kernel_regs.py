#!/usr/bin/env python3
"""Register budget of the run kernels in a built library: VGPRs, spilled registers and scratch bytes per kernel, read from
the AMDGPU metadata note of the gfx950 code object (no GPU needed).  The register-pressure work of DESIGN.md section 4 is
this loop: build one kernel class alone with a piece of source disabled (seconds instead of minutes)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DLBFT_DEV_ONLY_CLASS=7 [-D...] \\
          librabft_simulator_amd/csrc/lbft_hip.hip -o /tmp/dev7.so
    python tools/kernel_regs.py /tmp/dev7.so
and read what the piece costs; then measure the candidates on the GPU (tools/gpu_variants.sh).
    python tools/kernel_regs.py                      # the product library"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_abi import _kernel_metadata
    libs = sys.argv[1:] or [os.path.join(ROOT, "librabft_simulator_amd", "liblbft_hip.so")]
    for path in libs:
        for name, v in sorted(_kernel_metadata(path).items()):
            if "lbft_k_run" in name and v["vgpr_count"]:
                print("%-28s %-44s vgprs %3d  spilled %3d  scratch %4d B" % (os.path.basename(path), name[:44], v["vgpr_count"],
                                                                             v["vgpr_spill_count"], v["private_segment_fixed_size"]))


if __name__ == "__main__":
    main()
