set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03j
( LBFT_FUZZ_GPU_LARGE_CHUNKS=10 timeout 400 python -m pytest tests/test_fuzz_model.py -x -q -m gpu -k "random_large_configurations_on_the_device" ) > gpurun_out/r03j/large_fuzz.txt 2>&1; echo rc=$?; tail -2 gpurun_out/r03j/large_fuzz.txt
for e in 1 0 1 0; do
  echo "LBFT_CAL_BM_LDS=$e"
  LBFT_CAL_BM_LDS=$e timeout 600 python tools/configs.py c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ',d['config'], round(d['kernel_ms'],1), d['roofline']['kernel'], d['events'], d['rounds'], 'faulted', d['faulted_instances'])"
done
