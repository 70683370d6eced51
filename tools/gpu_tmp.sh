set -u
export TMPDIR=/tmp
bash tools/gpu_r03_b.sh r03h "liblbft_hip.so:0:-1 liblbft_hip_bcl4.so:0:-1 liblbft_hip_bcl3.so:0:-1 liblbft_hip_bcl5.so:0:-1 liblbft_hip_bcl2.so:0:-1 liblbft_hip.so:0:-1 liblbft_hip_bcl4.so:0:-1" liblbft_hip_bcl4.so
( LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_bcl4.so LBFT_FUZZ_GPU_QUAD_CHUNKS=10 timeout 600 python -m pytest tests/test_fuzz_model.py -q -m gpu -k "headline_network" ) > gpurun_out/r03h/quad_fuzz.txt 2>&1; echo rc=$?; tail -2 gpurun_out/r03h/quad_fuzz.txt
