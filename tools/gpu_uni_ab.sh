#!/bin/bash
# Runs on the GPU box (via gpurun): the first measurement of lbft_k_run0u (one network per wavefront as wavefront-uniform code on the scalar unit; built at
# the end of round 4, unmeasured).  Before the call, on the CPU side:
#   python -c "from librabft_simulator_amd import build; build.build_variant('uni', ['-DLBFT_WITH_UNI'])"
#   (optional: 'uni1' / 'uni2' with '-DLBFT_BLK_CACHE_UNI=1' / '=2' added -- fewer cached block records = less state parked in VGPR lanes: the kernel's static size
#    is 6.7 k / 9.3 k / 10.0 k instructions with 1 / 2 / 3 records; every miss is an L2 round trip on a lone wavefront)
# then   gpurun --timeout 600 -- 'bash tools/gpu_uni_ab.sh'
# (1) parity: the device tests whose batches have <= 2 048 networks run lbft_k_run0u under LBFT_UNI=1 and are compared with the oracle as always;
# (2) timing: 256 / 1 024 / 2 048 x 4 networks, product kernel (lbft_k_run0s) against the variant -- LBFT_UNI=1 is ignored by the product library.
set -u
export TMPDIR=/tmp
O=gpurun_out/uni
mkdir -p $O
U=$PWD/librabft_simulator_amd/liblbft_hip_uni.so
LBFT_HIP_LIB=$U LBFT_UNI=1 timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or gpu_equals_oracle or multi_launch or reset_reruns or zero_max_clock" > $O/parity_uni.txt 2>&1
echo "rc=$?" >> $O/parity_uni.txt; tail -4 $O/parity_uni.txt
for m in 256 1024 2048; do
  LBFT_UNI=1 timeout 300 python tools/sweep.py --libs liblbft_hip.so,liblbft_hip_uni.so,liblbft_hip_uni2.so,liblbft_hip_uni1.so --grid 0:-1 --reps 3 --instances $m >> $O/sweep.jsonl 2>> $O/sweep.err
done
python - <<'PY'
import json
for line in open("gpurun_out/uni/sweep.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "instances", "kernel_ms", "kernel_ms_all", "events", "faulted", "error")})
PY
