set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03full2
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size" --durations=10 > gpurun_out/r03full2/pytest_fullsize.log 2>&1; echo rc=$?; tail -20 gpurun_out/r03full2/pytest_fullsize.log
