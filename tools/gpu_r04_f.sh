#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04f
mkdir -p $O
timeout 600 python tools/sweep.py --libs ${LIBS} --grid 0:-1 --reps 3 > $O/sweep_q.jsonl 2> $O/sweep.err
timeout 600 python tools/sweep.py --libs ${LIBS} --grid 0:-1 --reps 3 --instances 32768 >> $O/sweep_q.jsonl 2>> $O/sweep.err
python - <<'PY'
import json
for line in open("gpurun_out/r04f/sweep_q.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "instances", "kernel_ms", "kernel_ms_all", "faulted", "error")})
PY
timeout 400 python tools/variant_parity.py ${PLIBS} > $O/parity.txt 2>&1; cat $O/parity.txt
INSTANCES=32768 timeout 400 python tools/variant_parity.py ${PLIBS} > $O/parity16.txt 2>&1; cat $O/parity16.txt
