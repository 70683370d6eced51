#!/bin/bash
# round 4: the LDS window of block records for the large-network kernels
set -u
export TMPDIR=/tmp
O=gpurun_out/r04e
mkdir -p $O
run() { tag=$1; lib=$2; shift 2; LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 900 python tools/configs.py --reps 2 "$@" 2>> $O/configs.err | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print('$tag', d['config'][:16], round(d['kernel_ms'], 2), 'faulted', d['faulted_instances'], 'rounds', d['rounds'], 'commits', d['commits'], 'events', d['events'], flush=True)
" | tee -a $O/configs.txt; }
C4=c4_16384x64_longtail_equivocators; C5=c5_8192x100_weighted_epochs; C4L=c4live_16384x64_longtail_equivocators_fixed; C5L=c5live_8192x100_rotating_rights_epochs_fixed
run prod liblbft_hip.so $C4 $C5
run l5w_3rec_win32 liblbft_hip_l5w.so $C4 $C5
run l5w1_1rec_win32 liblbft_hip_l5w1.so $C4 $C5
LBFT_BLK_WINDOW=0 run l5w1_1rec_nowin liblbft_hip_l5w1.so $C4 $C5
LBFT_BLK_WINDOW=64 run l5w1_1rec_win64 liblbft_hip_l5w1.so $C4 $C5
run prod liblbft_hip.so $C4L $C5L
run l7w_win32 liblbft_hip_l7w.so $C4L $C5L
LBFT_BLK_WINDOW=0 run l7w_nowin liblbft_hip_l7w.so $C4L $C5L
LBFT_BLK_WINDOW=16 run l7w_win16 liblbft_hip_l7w.so $C4L $C5L
LBFT_BLK_WINDOW=64 run l7w_win64 liblbft_hip_l7w.so $C4L $C5L
tail -3 $O/configs.err
