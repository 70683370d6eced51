#!/bin/bash
# round 4: the whole device suite, then the profiles of the bench workload and of BASELINE.json's configurations
set -u
export TMPDIR=/tmp
O=gpurun_out/r04full
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_full_suite.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_full_suite.txt; tail -5 $O/pytest_gpu_full_suite.txt
bash tools/gpu_profile.sh r04 > $O/profile.log 2>&1; tail -12 $O/profile.log
timeout 300 python tools/sweep.py --libs liblbft_hip_prof.so --grid 0:-1 --reps 2 > $O/phases_lbft_k_run0q.jsonl 2> $O/phases.err
timeout 300 python tools/sweep.py --libs liblbft_hip_prof.so --grid 0:-1 --reps 2 --instances 1024 > $O/phases_small_batches.jsonl 2>> $O/phases.err
timeout 300 python tools/sweep.py --libs liblbft_hip_prof.so --grid 0:-1 --reps 2 --instances 8192 >> $O/phases_small_batches.jsonl 2>> $O/phases.err
bash tools/gpu_configs_profile.sh r04 > $O/configs.log 2>&1; tail -30 $O/configs.log
