#!/bin/bash
# round 3: small-batch regime (1 024 x 4 = BASELINE config 2; 8 192 x 4 = one GPU's shard of config 3 on 8 GPUs) with / without the LDS window
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r03small}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or gpu_equals_oracle or multi_launch or checkpoint or reset_reruns or zero_max" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for nw in 0 1; do for m in 1024 2048 4096 8192 16384; do
  LBFT_NO_WINDOW=$nw timeout 200 python tools/sweep.py --one --instances $m --lpw 0 --ql -1 --reps 3 2>>$O/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('no_window=$nw', d['instances'], 'lpw', d['lpw'], 'ms %.3f' % d['kernel_ms'], d['events'], d['rounds'], 'faulted', d['faulted'])"
done; done | tee $O/small.txt
