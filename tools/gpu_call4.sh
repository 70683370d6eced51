#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py --libs liblbft_hip.so --grid 64:-1 > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err
timeout 600 python tools/sweep.py --libs liblbft_hip.so --grid 0:-1 --instances 1024 >> gpurun_out/sweep.jsonl 2>> gpurun_out/sweep.err
timeout 600 python tools/sweep.py --libs liblbft_hip.so --grid 0:-1 --instances 4096 --nodes 8 >> gpurun_out/sweep.jsonl 2>> gpurun_out/sweep.err
timeout 900 python tools/sweep.py --libs liblbft_hip.so --grid 0:-1 --instances 1024 --nodes 64 --max-clock 300 --reps 1 >> gpurun_out/sweep.jsonl 2>> gpurun_out/sweep.err
timeout 900 python tools/sweep.py --libs liblbft_hip.so --grid 0:-1 --instances 512 --nodes 100 --max-clock 300 --reps 1 >> gpurun_out/sweep.jsonl 2>> gpurun_out/sweep.err
python - <<'PY'
import json
for line in open("gpurun_out/sweep.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "instances", "nodes", "max_clock", "lpw", "kernel_ms", "events", "events_per_s", "faulted", "max_queue", "error")})
PY
tail -3 gpurun_out/sweep.err
