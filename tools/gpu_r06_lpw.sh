# Round 6, third session: lanes per wavefront re-swept on the final library (the runs changed what a wavefront's lanes do: the notification runs' segments are 64 / lpw lanes wide)
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06y}; mkdir -p $O
one() { cfg=$1; lpw=$2
  timeout 300 python tools/configs.py $cfg --reps 2 --lpw $lpw 2>> $O/lpw.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'][:12], 'lpw', $lpw, 'ms', round(d['kernel_ms'], 1), d['roofline']['kernel'], 'events', d['events'])" >> $O/lanes_per_wavefront_final.txt
}
for lpw in 2 4 8 16; do one c4_16384x64_longtail_equivocators $lpw; one c4live_16384x64_longtail_equivocators_fixed $lpw; done
for lpw in 2 4 8; do one c5_8192x100_weighted_epochs $lpw; one c5live_8192x100_rotating_rights_epochs_fixed $lpw; done
cat $O/lanes_per_wavefront_final.txt
