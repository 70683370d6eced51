set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03k
timeout 300 python tools/sweep.py --libs liblbft_hip_msgprof.so --grid 0:-1 > gpurun_out/r03k/msgprof.jsonl 2> gpurun_out/r03k/err.txt; tail -3 gpurun_out/r03k/err.txt
python - <<'PY'
import json
for l in open('gpurun_out/r03k/msgprof.jsonl'):
    d=json.loads(l); print(d['kernel_ms'], d.get('counts'), d.get('wave_steps'), d['events'])
PY
