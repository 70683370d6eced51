#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, the tuning sweep (tools/sweep.py) for LIBS / GRID, and the phase-timer
# breakdown of the diagnostic build.  Example: gpurun -- 'GRID=0:-1,64:-1 bash tools/gpu_check.sh'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py --libs ${LIBS:-liblbft_hip.so} --grid ${GRID:-64:-1,32:-1} > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err
timeout 300 python tools/sweep.py --libs ${PLIBS:-liblbft_hip_prof.so} --grid ${PGRID:-0:-1} >> gpurun_out/sweep.jsonl 2>> gpurun_out/sweep.err
python - <<'PY'
import json
for line in open("gpurun_out/sweep.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "lpw", "ql", "kernel_ms", "events_per_s", "faulted", "error", "cycles_per_wave_step")})
    if "phases" in d: print(d["phases"]); print(d.get("counts"))
PY
