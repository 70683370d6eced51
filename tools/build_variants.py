#!/usr/bin/env python3
"""Builds the opt-in variant libraries of round 4 (all unmeasured when written) next to the product library, in parallel:
    python tools/build_variants.py [tag ...]          # default: all
then  gpurun --timeout 2400 -- 'bash tools/gpu_uni_ab.sh; bash tools/gpu_s4_ab.sh; bash tools/gpu_w3_ab.sh'
The product library (liblbft_hip.so) is not touched: its machine code stays the profiled one."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from librabft_simulator_amd import build  # noqa: E402

LEAN3 = ["-DLBFT_LEAN_NODE_LDS=1", "-DLBFT_BLK_CACHE_LEAN5=1", "-DLBFT_LEAN_AX=0"]
VARIANTS = {
    # one network per wavefront as wavefront-uniform (scalar) code: LBFT_UNI=1 selects it at run time (tools/gpu_uni_ab.sh)
    "uni": ["-DLBFT_WITH_UNI"],
    "uni2": ["-DLBFT_WITH_UNI", "-DLBFT_BLK_CACHE_UNI=2"],
    "uni1": ["-DLBFT_WITH_UNI", "-DLBFT_BLK_CACHE_UNI=1"],
    # the small-batch kernel at four wavefronts per SIMD (tools/gpu_s4_ab.sh)
    "s4": ["-DLBFT_SMALL_WAVES_PER_SIMD=4", "-DLBFT_SMALL_RUN_WAVES=16", "-DLBFT_SMALL_NODE_LDS=1", "-DLBFT_BLK_CACHE_SMALL=1"],
    # the large-network kernels at three / four wavefronts per SIMD (tools/gpu_w3_ab.sh)
    "w3": ["-DLBFT_LEAN2_WAVES_PER_SIMD=3", "-DLBFT_LEAN2_RUN_WAVES=12"] + LEAN3,
    "w4": ["-DLBFT_LEAN2_WAVES_PER_SIMD=4", "-DLBFT_LEAN2_RUN_WAVES=16"] + LEAN3,
}


def main():
    tags = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(max_workers=min(len(tags), os.cpu_count() or 1)) as ex:
        for tag, path in zip(tags, ex.map(lambda t: build.build_variant(t, VARIANTS[t]), tags)):
            print(tag, os.path.basename(path), build.kernel_hash(path), flush=True)


if __name__ == "__main__":
    main()
