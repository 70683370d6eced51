#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r04g
mkdir -p $O; rm -f $O/*.jsonl
timeout 600 python tools/sweep.py --libs ${LIBS} --grid 0:-1 --reps 3 > $O/sweep_q.jsonl 2> $O/sweep.err
for inst in 1024 8192; do timeout 300 python tools/sweep.py --libs ${SLIBS} --grid 0:-1 --reps 4 --instances $inst >> $O/sweep_q.jsonl 2>> $O/sweep.err; done
python - <<'PY'
import json
for line in open("gpurun_out/r04g/sweep_q.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "instances", "kernel_ms", "kernel_ms_all", "faulted", "error")})
PY
timeout 400 python tools/variant_parity.py ${PLIBS} > $O/parity.txt 2>&1; cat $O/parity.txt
