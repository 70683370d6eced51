set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03i
( time LBFT_FUZZ_GPU_LARGE_CHUNKS=40 timeout 400 python -m pytest tests/test_fuzz_model.py -q -m gpu -k "random_large_configurations_on_the_device" ) > gpurun_out/r03i/large_fuzz.txt 2>&1; echo rc=$?; tail -6 gpurun_out/r03i/large_fuzz.txt
