#!/bin/bash
# Round 6 A/B: LIBS (space-separated library file names under librabft_simulator_amd/) on the four large configurations, three repetitions each; the headline
# bench line (kernel time only) on the first and the last of them; then the large-network parity subset on the product library.
#   bash tools/gpu_r06_ab3.sh <tag> "<libs>" [pytest -k expression | none]
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06g}
LIBS=${2:-"liblbft_hip.so"}
KEXPR=${3:-"config4_16384 or config5_8192x100_weighted or config4_live or config5_live or test_gpu_equals_oracle or heap_queue or long_horizon or multi_launch or checkpoint or reset or fuzz or large"}
mkdir -p $O
for lib in $LIBS; do
  [ -f librabft_simulator_amd/$lib ] || continue
  for cfg in c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed; do
    LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 300 python tools/configs.py $cfg --reps 3 2>> $O/ab.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['config'][:12], 'ms', round(d['kernel_ms'], 1), d['roofline']['kernel'], 'frac', round(d['roofline']['frac'], 4), 'events', d['events'], 'commits', d['commits'])" >> $O/ab.txt
  done
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-measure-traffic --parity-instances 1024 2>> $O/ab.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', 'headline ms_per_step', round(d['ms_per_step'], 3), 'kernel_ms', round(d['roofline']['kernel_ms'], 3), d['roofline']['kernel'], 'parity mismatches', d['parity']['mismatches'])" >> $O/ab.txt
done
cat $O/ab.txt
if [ "$KEXPR" != "none" ]; then
  LBFT_FUZZ_GPU_LARGE_CHUNKS=12 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_model.py -m gpu -x -q --durations=5 -k "$KEXPR" > $O/pytest_subset.txt 2>&1; tail -9 $O/pytest_subset.txt
fi
