set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "third_party or golden or gpu_equals_oracle or full_batch" > gpurun_out/r03log/pytest.log 2>&1; echo rc=$?; tail -4 gpurun_out/r03log/pytest.log
