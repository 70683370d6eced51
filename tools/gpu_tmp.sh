set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03fuzz
( time LBFT_FUZZ_GPU_CHUNKS=60 timeout 1200 python -m pytest tests/test_fuzz_model.py -x -q -m gpu ) > gpurun_out/r03fuzz/device_fuzz.txt 2>&1; echo rc=$?; tail -6 gpurun_out/r03fuzz/device_fuzz.txt
