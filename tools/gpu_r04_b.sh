#!/bin/bash
# round 4, GPU call B: load_node / launch-contract device tests, small-batch drain variants, block-cache policy, phase profiles of the live configurations
set -u
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
timeout 900 python -m pytest tests/test_save_node.py tests/test_abi.py tests/test_bench_launch.py -m gpu -x -q > $O/pytest_save_node.log 2>&1; echo "rc=$?" >> $O/pytest_save_node.log; tail -5 $O/pytest_save_node.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_node_level.py -m gpu -x -q -k "bench_two_ranks or allreduce" > $O/pytest_bench.log 2>&1; echo "rc=$?" >> $O/pytest_bench.log; tail -5 $O/pytest_bench.log
for inst in 1024 8192; do
  timeout 300 python tools/sweep.py --libs liblbft_hip.so,liblbft_hip_s_base.so,liblbft_hip_s_d1.so,liblbft_hip_s_d2.so --grid 0:-1 --reps 4 --instances $inst >> $O/sweep_small.jsonl 2>> $O/sweep.err
done
timeout 300 python tools/sweep.py --libs liblbft_hip.so,liblbft_hip_fifo.so,liblbft_hip_fifo4.so --grid 0:-1 --reps 3 >> $O/sweep_q.jsonl 2>> $O/sweep.err
python - <<'PY'
import json
for f in ("gpurun_out/r04b/sweep_small.jsonl", "gpurun_out/r04b/sweep_q.jsonl"):
    for line in open(f):
        d = json.loads(line)
        print({k: d.get(k) for k in ("lib", "instances", "kernel_ms", "kernel_ms_all", "faulted", "error")})
PY
timeout 300 python tools/variant_parity.py liblbft_hip.so liblbft_hip_fifo.so > $O/parity.txt 2>&1; cat $O/parity.txt
for cfg in c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed c4_16384x64_longtail_equivocators; do
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_prof.so timeout 400 python tools/configs.py $cfg >> $O/phases_large.jsonl 2>> $O/phases.err
done
python - <<'PY'
import json
for line in open("gpurun_out/r04b/phases_large.jsonl"):
    d = json.loads(line)
    print(d["config"], round(d["kernel_ms"], 1), d.get("cycles_per_wave_step"), d.get("wave_steps"))
    print(sorted(d.get("phases", {}).items(), key=lambda kv: -kv[1])[:16])
PY
timeout 200 python bench.py --steps 3 --warmup 1 > $O/bench_line.json 2> $O/bench.err; tail -c 1500 $O/bench_line.json; tail -3 $O/bench.err
