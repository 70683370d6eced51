#!/bin/bash
# round 3: A/B of class-0 variants on the bench workload (run on the GPU box via gpurun)
#   bash tools/gpu_r03_b.sh <out tag> "<lib:lpw:ql> ..." [pytest-lib]
set -u
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
L=$PWD/librabft_simulator_amd
if [ -n "${3:-}" ]; then
  LBFT_HIP_LIB=$L/$3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or gpu_equals_oracle or full_size_65536x4 or checkpoint or multi_launch_equals or reset_reruns" > $O/pytest_variant.log 2>&1
  tail -3 $O/pytest_variant.log
fi
: > $O/sweep.jsonl
for item in $2; do
  lib=${item%%:*}; rest=${item#*:}
  timeout 300 python tools/sweep.py --libs $lib --grid $rest ${SWEEP_ARGS:-} >> $O/sweep.jsonl 2>> $O/sweep.err
done
python - $O/sweep.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print(d.get("lib"), "lpw", d.get("lpw"), "ql", d.get("ql"), "ms", d.get("kernel_ms") and round(d["kernel_ms"],3), d.get("events"), d.get("rounds"), "faulted", d.get("faulted"), d.get("error","")[:300])
    if "phases" in d: print("  steps/wave", d.get("wave_steps",0)/2048.0, "cyc/step", d.get("cycles_per_wave_step"), json.dumps(d["phases"]))
PY
