set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03f
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or gpu_equals_oracle or checkpoint or multi_launch_equals or reset_reruns or c2 or small" > gpurun_out/r03f/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03f/pytest.log
timeout 600 python -m pytest tests/test_fuzz_model.py -x -q -m gpu -k "random_configurations_on_the_device" > gpurun_out/r03f/fuzz.log 2>&1; echo rc=$?; tail -2 gpurun_out/r03f/fuzz.log
for e in 0 1; do
  echo "LBFT_NO_LSTATE=$e"
  for m in 256 1024 2048; do
    LBFT_NO_LSTATE=$e timeout 300 python tools/sweep.py --libs liblbft_hip.so --grid 0:-1 --instances $m 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ',d['instances'], round(d['kernel_ms'],3), d['events'], d['rounds'], 'faulted', d['faulted'])"
  done
  LBFT_NO_LSTATE=$e timeout 300 python tools/configs.py c2_1024x4_lognormal c2_1024x4_uniform 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ',d['config'], round(d['kernel_ms'],3), d['roofline']['kernel'], d['events'])"
done
