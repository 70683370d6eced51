# Round 6, last session, first call: (1) what the runs retire on the device (tools/gpu_r06_run_counts.sh), (2) the device fuzz widened once more on the final
# library with chunks no earlier run drew (LBFT_FUZZ_GPU_*_FIRST).
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06rc}; mkdir -p $O
[ -n "${SKIP_RUN_COUNTS:-}" ] || bash tools/gpu_r06_run_counts.sh ${1:-r06rc} > $O/run_counts.txt 2>&1; cat $O/run_counts.txt
LBFT_FUZZ_GPU_FIRST=${FUZZ_FIRST:-80} LBFT_FUZZ_GPU_CHUNKS=${FUZZ_CHUNKS:-120} LBFT_FUZZ_GPU_QUAD_FIRST=50 LBFT_FUZZ_GPU_QUAD_CHUNKS=75 LBFT_FUZZ_GPU_LARGE_FIRST=64 LBFT_FUZZ_GPU_LARGE_CHUNKS=64 \
  timeout 1100 python -m pytest tests/test_fuzz_model.py -m gpu -q > $O/device_fuzz_new_chunks.txt 2>&1; tail -4 $O/device_fuzz_new_chunks.txt
