#!/bin/bash
# Runs on the GPU box (via gpurun): the first measurement of the FOUR-wavefronts-per-SIMD build of the small-batch kernel lbft_k_run0s (the event's node in an LDS
# column, one cached block record: 128 registers, 28 spilled dwords; batches of <= 32 768 networks spread over 4 096 wavefronts of 1 / 2 / 4 / 8 lanes; built at the end
# of round 4, unmeasured).  Before the call, on the CPU side:
#   python -c "from librabft_simulator_amd import build; build.build_variant('s4', ['-DLBFT_SMALL_WAVES_PER_SIMD=4', '-DLBFT_SMALL_RUN_WAVES=16', '-DLBFT_SMALL_NODE_LDS=1', '-DLBFT_BLK_CACHE_SMALL=1'])"
# then   gpurun --timeout 900 -- 'bash tools/gpu_s4_ab.sh'
# (1) the variant bit for bit against the product library on whole batches (tools/variant_parity.py) and against the oracle (the device tests of small batches);
# (2) timing: 1 024 .. 32 768 x 4 networks, product against variant (8 192 = one GPU's share of the headline batch on an 8-GPU node: 9.6 ms as shipped).
set -u
export TMPDIR=/tmp
O=gpurun_out/s4
mkdir -p $O
S=$PWD/librabft_simulator_amd/liblbft_hip_s4.so
for m in 4096 8192 16384 32768; do LPW2=0 INSTANCES=$m timeout 300 python tools/variant_parity.py liblbft_hip.so liblbft_hip_s4.so | tee -a $O/parity_vs_product.txt; done
LBFT_HIP_LIB=$S timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or gpu_equals_oracle or multi_launch or reset_reruns or zero_max_clock or checkpoint" > $O/parity_s4.txt 2>&1
echo "rc=$?" >> $O/parity_s4.txt; tail -4 $O/parity_s4.txt
for m in 1024 2048 4096 8192 16384 32768; do
  timeout 300 python tools/sweep.py --libs liblbft_hip.so,liblbft_hip_s4.so --grid 0:-1 --reps 3 --instances $m >> $O/sweep.jsonl 2>> $O/sweep.err
done
python - <<'PY'
import json
for line in open("gpurun_out/s4/sweep.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "instances", "kernel_ms", "kernel_ms_all", "events", "faulted", "error")})
PY
