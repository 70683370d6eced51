#!/bin/bash
# round 4, GPU call C: the immediate-offset / scalar-base build (registers freed in every kernel) and what fits on top of it
set -u
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
timeout 600 python tools/sweep.py --libs liblbft_hip.so,liblbft_hip_n5.so,liblbft_hip_n5c4.so,liblbft_hip_n5d1.so,liblbft_hip_n5x.so,liblbft_hip_n5c4x.so,liblbft_hip_n5c4d1.so,liblbft_hip_full5.so --grid 0:-1 --reps 3 > $O/sweep_q.jsonl 2> $O/sweep.err
python - <<'PY'
import json
for line in open("gpurun_out/r04c/sweep_q.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "instances", "kernel_ms", "kernel_ms_all", "faulted", "error")})
PY
timeout 400 python tools/variant_parity.py liblbft_hip.so liblbft_hip_n5.so liblbft_hip_n5c4x.so liblbft_hip_full5.so > $O/parity.txt 2>&1; cat $O/parity.txt
run() { lib=$1; shift; LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 600 python tools/configs.py --reps 2 "$@" 2>> $O/configs.err | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print('$lib', d['config'][:16], round(d['kernel_ms'], 2), 'faulted', d['faulted_instances'], 'rounds', d['rounds'], 'commits', d['commits'], 'events', d['events'], flush=True)
" | tee -a $O/configs.txt; }
run liblbft_hip.so c2_1024x4_lognormal c2_1024x4_uniform c3shard_8192x4 c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs
run liblbft_hip_full5.so c2_1024x4_lognormal c2_1024x4_uniform c3shard_8192x4 c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs
for v in l5c2 l5c3 l5bx; do run liblbft_hip_$v.so c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs; done
run liblbft_hip.so c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed
run liblbft_hip_full5.so c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed
run liblbft_hip_l7c2.so c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed
tail -5 $O/configs.err
