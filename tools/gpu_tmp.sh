set -u
export TMPDIR=/tmp
bash tools/gpu_r03_b.sh r03d "liblbft_hip_w0.so:0:-1 liblbft_hip.so:0:-1 liblbft_hip_w4c2.so:0:-1 liblbft_hip_w4c1.so:0:-1 liblbft_hip_w2c3.so:0:-1 liblbft_hip_w8c2.so:0:-1 liblbft_hip_w0.so:0:-1 liblbft_hip.so:0:-1" liblbft_hip.so
( time LBFT_FUZZ_GPU_QUAD_CHUNKS=10 timeout 600 python -m pytest tests/test_fuzz_model.py -q -m gpu -k "headline_network" ) > gpurun_out/r03d/quad_fuzz.txt 2>&1; echo rc=$?; tail -3 gpurun_out/r03d/quad_fuzz.txt
