set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03dpp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_model.py -x -q -m gpu -k "golden or gpu_equals_oracle or multi_launch or checkpoint or reset_reruns or zero_max or capacity or random" > gpurun_out/r03dpp/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03dpp/pytest.log
for lib in liblbft_hip.so liblbft_hip_prev.so; do for m in 1024 2048 4096; do
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 200 python tools/sweep.py --one --instances $m --lpw 0 --ql -1 --reps 3 2>>gpurun_out/r03dpp/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['instances'], 'lpw', d['lpw'], 'ms %.3f' % d['kernel_ms'], d['events'], 'faulted', d['faulted'])"
done; done | tee gpurun_out/r03dpp/dpp.txt
