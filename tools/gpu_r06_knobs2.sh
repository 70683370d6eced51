#!/bin/bash
# Round 6, third session: ring size / top-up rate re-swept after the runs of all kinds (a large network now makes ~15 k loop iterations instead of 215 k:
# a top-up of 4 draws per iteration no longer feeds the bulk sends, whose leader then generates the draws one lane at a time).  LIBS = libraries to sweep.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06k2}
LIBS=${2:-"liblbft_hip.so"}
mkdir -p $O
one() {  # label, cfg
  label=$1; cfg=$2
  timeout 200 python tools/configs.py $cfg --reps 2 2>> $O/knobs.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$label', d['config'][:12], 'ms', round(d['kernel_ms'], 1), 'events', d['events'])" >> $O/knobs2.txt
}
for lib in $LIBS; do
  export LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib
  for cfg in c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed; do
    one "$lib default(ring512,topup4)" $cfg
    for t in 16 32 64 128; do LBFT_RING_TOPUP=$t one "$lib topup$t" $cfg; done
    for r in 1024 2048; do LBFT_RING=$r LBFT_RING_TOPUP=64 one "$lib ring$r,topup64" $cfg; done
  done
done
cat $O/knobs2.txt
