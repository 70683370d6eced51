#!/bin/bash
# Runs on the GPU box (via gpurun): kernel time of the bench workload for LIBS (default: product + phase-timer build).
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 400 python tools/sweep.py --libs ${LIBS:-liblbft_hip.so,liblbft_hip_prof.so} --grid ${GRID:-0:-1} > gpurun_out/sweepq.jsonl 2> gpurun_out/sweepq.err
python - <<PY
import json
for line in open("gpurun_out/sweepq.jsonl"):
    d = json.loads(line)
    print({k: d.get(k) for k in ("lib", "lpw", "ql", "kernel_ms", "events", "faulted", "error", "cycles_per_wave_step", "max_queue")})
    if "phases" in d: print(d["phases"]); print(d.get("counts"))
PY
tail -3 gpurun_out/sweepq.err
