set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03big
for rep in 1 2; do for lib in liblbft_hip.so liblbft_hip_prev.so; do
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 600 python tools/configs.py c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs --reps 2 2>>gpurun_out/r03big/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['config'][:10], 'ms %.2f' % d['kernel_ms'], d['events'], 'faulted', d['faulted_instances'], d['roofline']['kernel'])"
done; done | tee gpurun_out/r03big/ab.txt
