#!/bin/bash
# round 3: full GPU suite + bench lines (weak default, strong 65536 total on one rank = same batch, small strong shards) on the GPU box
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r03full}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 10 --warmup 2 --total-instances 8192 --no-cpu-baseline > $O/bench_strong_8192.json 2>> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 2 --total-instances 1024 --no-cpu-baseline > $O/bench_strong_1024.json 2>> $O/bench.err
python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"], "exec %.3f" % d["roofline"]["executed"]["frac"], "parity", d.get("parity", {}).get("mismatches"), d.get("parity", {}).get("checked_instances"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
