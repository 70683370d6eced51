#!/bin/bash
# Round 6, third session: second ring / top-up sweep around the new defaults (16 per iteration up to 64 nodes, 128 above; ring 512) on the final library.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06k3}
mkdir -p $O
one() { label=$1; cfg=$2
  timeout 200 python tools/configs.py $cfg --reps 2 2>> $O/knobs.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$label', d['config'][:12], 'ms', round(d['kernel_ms'], 1), 'events', d['events'])" >> $O/knobs3.txt
}
for cfg in c5_8192x100_weighted_epochs c5live_8192x100_rotating_rights_epochs_fixed; do
  one "default(ring512,topup128)" $cfg
  for t in 192 256; do LBFT_RING_TOPUP=$t one "topup$t" $cfg; done
  for t in 128 256 512; do LBFT_RING=1024 LBFT_RING_TOPUP=$t one "ring1024,topup$t" $cfg; done
done
for cfg in c4_16384x64_longtail_equivocators c4live_16384x64_longtail_equivocators_fixed; do
  one "default(ring512,topup16)" $cfg
  for t in 8 24; do LBFT_RING_TOPUP=$t one "topup$t" $cfg; done
  LBFT_RING=256 one "ring256,topup16" $cfg
done
cat $O/knobs3.txt
