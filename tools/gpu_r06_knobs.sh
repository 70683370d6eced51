#!/bin/bash
# Round 6: run-time knobs of the cooperative large-network kernels after the runs -- ring size / top-up rate of the pre-generated draws (environment),
# lanes per wavefront -- on the four large configurations, two repetitions each.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06k}
mkdir -p $O
one() {  # label, cfg, extra args...
  label=$1; cfg=$2; shift 2
  timeout 200 python tools/configs.py $cfg --reps 2 "$@" 2>> $O/knobs.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$label', d['config'][:12], 'lpw', d['layout']['lanes_per_wavefront'], 'ms', round(d['kernel_ms'], 1))" >> $O/knobs.txt
}
for cfg in c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed; do
  one default $cfg
  for t in 0 2 8 16; do LBFT_RING_TOPUP=$t one topup$t $cfg; done
  for r in 256 1024; do LBFT_RING=$r one ring$r $cfg; done
done
for cfg in c4_16384x64_longtail_equivocators c4live_16384x64_longtail_equivocators_fixed; do one lpw4 $cfg --lpw 4; one lpw16 $cfg --lpw 16; done
for cfg in c5_8192x100_weighted_epochs c5live_8192x100_rotating_rights_epochs_fixed; do one lpw2 $cfg --lpw 2; one lpw8 $cfg --lpw 8; done
cat $O/knobs.txt
