set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03chk
timeout 1500 python -m pytest tests -x -q -m gpu -k "not full_size_config and not full_batch" > gpurun_out/r03chk/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03chk/pytest.log
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r03chk/bench.json 2> gpurun_out/r03chk/bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03chk/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.3f kernel_ms %.3f frac %.3f exec %.3f parity %s cpu %.4g" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["executed"]["frac"], d["parity"], d["cpu_baseline"]["value"]))
PY
