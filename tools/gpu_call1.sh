#!/bin/bash
# GPU call: parity tests, tuning sweep, phase-timer breakdown, rocprof kernel stats.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py --libs liblbft_hip.so --grid 64:-1,32:-1,16:-1,64:0,32:0,64:32 > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err
timeout 300 python tools/sweep.py --libs liblbft_hip_prof.so --grid 64:-1,32:-1,32:0 >> gpurun_out/sweep.jsonl 2>> gpurun_out/sweep.err
cat gpurun_out/sweep.jsonl | cut -c1-900
timeout 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.json | cut -c1-1500
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o stats --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1
find gpurun_out/prof_stats -name '*kernel_stats*' | head -3 | xargs -r head -5
