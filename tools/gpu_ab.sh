#!/bin/bash
# Runs on the GPU box (via gpurun): A/B timing of variant builds of the library on the bench workload + bit-exactness against the first one.
#   LIBS=liblbft_hip.so,liblbft_hip_x.so PLIBS="liblbft_hip.so liblbft_hip_x.so" [INSTANCES=32768] bash tools/gpu_ab.sh
# (variants: librabft_simulator_amd/build.py build_variant, or one kernel class alone with -DLBFT_DEV_ONLY_CLASS=k in seconds)
set -u
export TMPDIR=/tmp
O=gpurun_out/ab
mkdir -p $O; rm -f $O/*.jsonl
timeout 900 python tools/sweep.py --libs ${LIBS} --grid 0:-1 --reps 3 --instances ${INSTANCES:-65536} > $O/sweep.jsonl 2> $O/sweep.err
python - <<'PY'
import json
for line in open("gpurun_out/ab/sweep.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "instances", "kernel_ms", "kernel_ms_all", "faulted", "error")})
PY
[ -n "${PLIBS:-}" ] && timeout 600 python tools/variant_parity.py ${PLIBS} | tee $O/parity.txt
