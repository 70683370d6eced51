#!/bin/bash
# Runs on the GPU box (via gpurun): the first measurement of the THREE-wavefronts-per-SIMD build of the large-network kernels (lbft_k_run2l / lbft_k_run2q with the
# event's node in an LDS column, one cached block record, no staged author sets: 168 registers, 28 / 48 spilled dwords; built at the end of round 4, unmeasured).
# Before the call, on the CPU side:
#   python -c "from librabft_simulator_amd import build; build.build_variant('w3', ['-DLBFT_LEAN2_WAVES_PER_SIMD=3', '-DLBFT_LEAN2_RUN_WAVES=12', '-DLBFT_LEAN_NODE_LDS=1', '-DLBFT_BLK_CACHE_LEAN5=1', '-DLBFT_LEAN_AX=0'])"
#   (and, same flags with 4 / 16: liblbft_hip_w4.so -- FOUR wavefronts per SIMD at 128 registers, 68 / 96 spilled dwords, lanes per wavefront a power of two again)
# then   gpurun --timeout 1500 -- 'bash tools/gpu_w3_ab.sh'
# (1) parity of the variant: the large-network device tests + a sample of the full-size checks against the oracle; (2) timing of the four large configurations, product
# library against the variant (tools/gpu_variants.sh: with and without the two-wavefront kernels).
set -u
export TMPDIR=/tmp
O=gpurun_out/w3
mkdir -p $O
W=$PWD/librabft_simulator_amd/liblbft_hip_w3.so
LBFT_HIP_LIB=$W timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "gpu_equals_oracle or heap_queue or long_horizon or multi_launch" > $O/parity_w3.txt 2>&1
echo "rc=$?" >> $O/parity_w3.txt; tail -4 $O/parity_w3.txt
LBFT_HIP_LIB=$W LBFT_FULL_CHECK_FRACTION=0.25 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "full_size_config4 or full_size_config5" > $O/full_size_w3.txt 2>&1
echo "rc=$?" >> $O/full_size_w3.txt; tail -4 $O/full_size_w3.txt
[ -f librabft_simulator_amd/liblbft_hip_w4.so ] && V="prod w3 w4" || V="prod w3"
bash tools/gpu_variants.sh "$V" "c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed" | tee $O/variants.txt
