set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
( time LBFT_FUZZ_GPU_QUAD_CHUNKS=20 timeout 900 python -m pytest tests/test_fuzz_model.py -q -m gpu -k "headline_network" ) > gpurun_out/r03c/quad_fuzz.txt 2>&1; echo rc=$?; tail -5 gpurun_out/r03c/quad_fuzz.txt
for k in "follows_the_state or round_switch_csv"; do
  timeout 600 python -m pytest tests/test_fuzz_model.py tests/test_save_node.py tests/test_gpu_parity.py -q -m gpu -k "$k" > gpurun_out/r03c/diag.txt 2>&1; echo "[$k] rc=$?"; tail -2 gpurun_out/r03c/diag.txt
done
