#!/bin/bash
# Round 4's last GPU call (8.8 GPU-minutes left): (1) the divergence curve of the shipped headline kernel (the same 65 536 x 4 batch with 1, 2, 8, 32 instances
# per seed and with one seed: how much of the launch is the union of divergent paths), (2) the widened oracle samples of the full-size checks, most valuable
# first.  Everything is written as it completes: the call may be cut short by the budget clamp.
set -u
export TMPDIR=/tmp
O=gpurun_out/r04last
mkdir -p $O
for g in 0 2 8 32; do
  timeout 120 python tools/sweep.py --libs liblbft_hip.so --grid 0:-1 --reps 3 --seed-groups $g >> $O/divergence.jsonl 2>> $O/divergence.err
done
timeout 120 python tools/sweep.py --libs liblbft_hip.so --grid 0:-1 --reps 3 --same-seed >> $O/divergence.jsonl 2>> $O/divergence.err
cut -c1-330 $O/divergence.jsonl
for k in config5_live config4_live config5_8192x100_weighted; do
  timeout 500 python -m pytest tests/test_gpu_parity.py -q -k "$k" --durations=3 > $O/full_size_$k.txt 2>&1; echo "rc=$?" >> $O/full_size_$k.txt; tail -6 $O/full_size_$k.txt
done
