#!/bin/bash
# round 4, GPU call A: drain / hoist variants of lbft_k_run0q (timing + bit-exactness against the round-3 kernel) and PC sampling
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 500 python tools/sweep.py --libs liblbft_hip.so,liblbft_hip_base.so,liblbft_hip_d1.so,liblbft_hip_d2.so,liblbft_hip_d1r.so,liblbft_hip_d1t.so,liblbft_hip_h.so,liblbft_hip_d1h.so --grid 0:-1 --reps 3 > $O/sweep.jsonl 2> $O/sweep.err
python - <<'PY'
import json
for line in open("gpurun_out/r04a/sweep.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "kernel_ms", "kernel_ms_all", "faulted", "error")})
PY
timeout 400 python tools/variant_parity.py liblbft_hip.so liblbft_hip_d1.so liblbft_hip_d2.so liblbft_hip_d1h.so > $O/parity.txt 2>&1; cat $O/parity.txt
bash tools/gpu_pcsample.sh r04a liblbft_hip_dbg.so "stochastic:cycles:65536 host_trap:time:50" 2>&1 | tail -150
