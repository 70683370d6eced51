#!/bin/bash
# Round 6 A/B of the notification runs: (a) c4 / c5 on a build whose notification runs are compiled in but never attempted (-DLBFT_NTF_MIN=64u): what the
# code's presence costs the ordinary steps; (b) BASELINE config 5 as named with and without the runs.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06q}
mkdir -p $O
one() { lib=$1; cfg=$2; reps=$3
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 400 python tools/configs.py $cfg --reps $reps 2>> $O/ab.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['config'][:12], 'ms', round(d['kernel_ms'], 1), d['roofline']['kernel'], 'events', d['events'], 'commits', d['commits'])" >> $O/ab.txt
}
for lib in liblbft_hip.so liblbft_hip_ntf64.so liblbft_hip_nontf.so; do
  one $lib c4_16384x64_longtail_equivocators 3
  one $lib c5_8192x100_weighted_epochs 3
done
for lib in liblbft_hip.so liblbft_hip_nontf.so; do one $lib c5named_8192x100_weighted_epoch_every_50_commits 1; done
cat $O/ab.txt
