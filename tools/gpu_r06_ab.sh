#!/bin/bash
# Round 6, one GPU call: (a) the full-size configurations on the built library (results exported for the offline comparison), (b) the large-network parity
# subset, (c) lanes-per-wavefront sweep of the large-network kernels, (d) the headline workload at 64 networks per wavefront, one wavefront per SIMD
# (variant build -DLBFT_QUAD_STRIDE32=0: liblbft_hip_q64.so) against the product's 32 x two per SIMD, (e) the default bench line.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06d}
mkdir -p $O
(timeout 400 python tests/tools/full_size_export.py c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed --out $O/full_size > $O/export.log 2>&1; echo rc=$? >> $O/export.log); tail -6 $O/export.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_gpu_equals_oracle or heap_queue or long_horizon or multi_launch or checkpoint or reset" > $O/pytest_subset.txt 2>&1; tail -3 $O/pytest_subset.txt
for cfg in c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed; do
  for lpw in 2 4 8 16; do
    timeout 200 python tools/configs.py $cfg --lpw $lpw 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'][:10], 'lpw', d['layout']['lanes_per_wavefront'], 'ms', round(d['kernel_ms'], 1), d['roofline']['kernel'], 'GB', round(d['device_gb'], 1))" >> $O/lpw_sweep.txt
  done
done
cat $O/lpw_sweep.txt
if [ -f librabft_simulator_amd/liblbft_hip_q64.so ]; then
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_q64.so timeout 300 python bench.py --steps 10 --warmup 2 --lpw 64 --no-cpu-baseline --no-measure-traffic --parity-instances 2048 > $O/bench_q64_lpw64.json 2> $O/bench_q64.err
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_q64.so timeout 300 python bench.py --steps 10 --warmup 2 --lpw 32 --no-cpu-baseline --no-measure-traffic --parity-instances 2048 > $O/bench_q64_lpw32.json 2>> $O/bench_q64.err
  python - $O <<'PY'
import json, sys
for n in ("bench_q64_lpw64", "bench_q64_lpw32"):
    try:
        d = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        print(n, "ms_per_step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms"], 3), d["roofline"]["kernel"], "lpw", d["roofline"]["layout"]["lanes_per_wavefront"], "parity", d["parity"]["mismatches"])
    except Exception as e:
        print(n, "failed", e)
PY
fi
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err
python - $O <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("bench: value %.4g ms_per_step %.3f kernel_ms %.3f frac %.4f traffic %s (%s) cpu %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"], r["traffic_source"][:40], d.get("cpu_baseline", {}).get("value")))
print("cpu cores used", d.get("cpu_baseline", {}).get("cores"), "measured_here", r.get("traffic_measured_here"))
PY
