#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py --libs liblbft_hip.so,liblbft_hip_rowmajor.so --grid 64:-1,32:-1,64:0 > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err
timeout 300 python tools/sweep.py --libs liblbft_hip_prof.so --grid 64:-1,32:-1 >> gpurun_out/sweep.jsonl 2>> gpurun_out/sweep.err
cat gpurun_out/sweep.jsonl | cut -c1-1100
