set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03w8
for lib in liblbft_hip.so liblbft_hip_w8.so; do for cfg in "65536 0" "16384 0" "4096 0" "32768 0"; do set -- $cfg
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib LBFT_NO_WINDOW=1 timeout 200 python tools/sweep.py --one --instances $1 --lpw $2 --ql -1 --reps 3 2>>gpurun_out/r03w8/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['instances'], 'lpw', d['lpw'], 'ms %.3f' % d['kernel_ms'], d['events'], 'faulted', d['faulted'])"
done; done | tee gpurun_out/r03w8/w8.txt
