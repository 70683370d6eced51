#!/bin/bash
# round 3, call A: class-0 instance-major layout A/B + phase profiles (run on the GPU box via gpurun)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
L=$PWD/librabft_simulator_amd
# parity of the variant (class-0 cases + full-size property test + checkpoint)
LBFT_HIP_LIB=$L/liblbft_hip_imaja.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or gpu_equals_oracle or full_size_65536x4 or checkpoint or multi_launch_equals or reset_reruns" > $O/pytest_imaja.log 2>&1
tail -3 $O/pytest_imaja.log
timeout 900 python tools/sweep.py --libs liblbft_hip.so,liblbft_hip_imaj.so,liblbft_hip_imaja.so --grid 32:-1,64:-1 > $O/sweep.jsonl 2> $O/sweep.err
timeout 600 python tools/sweep.py --libs liblbft_hip_prof.so,liblbft_hip_profa.so --grid 32:-1 > $O/phases.jsonl 2>> $O/sweep.err
python - <<'PY'
import json
for f in ("gpurun_out/r03a/sweep.jsonl","gpurun_out/r03a/phases.jsonl"):
    for l in open(f):
        d=json.loads(l)
        print(d.get("lib"), d.get("lpw"), d.get("kernel_ms"), d.get("events"), d.get("rounds"), d.get("faulted"), d.get("error","")[:200])
        if "phases" in d: print("  ", d.get("cycles_per_wave_step"), json.dumps(d["phases"]))
PY
