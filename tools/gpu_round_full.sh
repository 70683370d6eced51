#!/bin/bash
# A round's closing GPU call: the whole device suite, then the profiles of the bench workload (kernel stats, PMC traffic + issue, phases from
# liblbft_hip_prof.so = build_variant("prof", ["-DLBFT_PHASE_TIMERS"])) and of BASELINE.json's configurations.  ~12 minutes of GPU time.
#   gpurun --timeout 5400 -- 'bash tools/gpu_round_full.sh r05'     then copy gpurun_out/{profile,configs}_<tag>/ into profiles/<tag>/ and profiles/current/
set -u
export TMPDIR=/tmp
TAG=${1:-rXX}
O=gpurun_out/${TAG}full
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu_full_suite.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_full_suite.txt; tail -5 $O/pytest_gpu_full_suite.txt
bash tools/gpu_profile.sh $TAG > $O/profile.log 2>&1; tail -12 $O/profile.log
timeout 300 python tools/sweep.py --libs liblbft_hip_prof.so --grid 0:-1 --reps 2 > $O/phases_lbft_k_run0q.jsonl 2> $O/phases.err
timeout 300 python tools/sweep.py --libs liblbft_hip_prof.so --grid 0:-1 --reps 2 --instances 1024 > $O/phases_small_batches.jsonl 2>> $O/phases.err
timeout 300 python tools/sweep.py --libs liblbft_hip_prof.so --grid 0:-1 --reps 2 --instances 8192 >> $O/phases_small_batches.jsonl 2>> $O/phases.err
bash tools/gpu_configs_profile.sh $TAG > $O/configs.log 2>&1; tail -30 $O/configs.log
