#!/bin/bash
# Round 6: response runs (coop_responses, LBFT_RSPRUN) A/B -- the product library against the variant built with -DLBFT_RSPRUN=0 (liblbft_hip_norsp.so: the
# machine code of the previous commit), c4 / c5 at full size, three repetitions each; then parity of the product library: the large-network device cases and the
# full-size fixture checks of c4 / c5 (every instance), the device fuzz of large networks.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06f}
mkdir -p $O
for lib in liblbft_hip.so liblbft_hip_norsp.so; do
  [ -f librabft_simulator_amd/$lib ] || continue
  for cfg in c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed; do
    LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 300 python tools/configs.py $cfg --reps 3 2>> $O/ab.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$lib', d['config'][:12], 'ms', round(d['kernel_ms'], 1), d['roofline']['kernel'], 'frac', round(d['roofline']['frac'], 4), 'events', d['events'], 'commits', d['commits'])" >> $O/ab.txt
  done
done
cat $O/ab.txt
LBFT_FUZZ_GPU_LARGE_CHUNKS=12 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_model.py -m gpu -x -q --durations=10 -k "config4_16384 or config5_8192x100_weighted or config4_live or config5_live or test_gpu_equals_oracle or heap_queue or long_horizon or multi_launch or checkpoint or reset or fuzz or large" > $O/pytest_subset.txt 2>&1; tail -15 $O/pytest_subset.txt
if [ -f librabft_simulator_amd/liblbft_hip_prof.so ]; then
  for cfg in c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs; do
    LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_prof.so timeout 300 python tools/configs.py $cfg >> $O/phases_after_response_runs.jsonl 2>> $O/phases.err
  done
fi
