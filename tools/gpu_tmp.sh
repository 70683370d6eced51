set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03quad
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_model.py -x -q -m gpu -k "golden or gpu_equals_oracle or multi_launch or checkpoint or reset_reruns or zero_max or full_size_65536x4 or capacity or two_ranks_on_one or random" > gpurun_out/r03quad/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03quad/pytest.log
for nq in 0 1; do for m in 1024 4096 8192 16384 65536; do
  LBFT_NO_QUAD=$nq timeout 200 python tools/sweep.py --one --instances $m --lpw 0 --ql -1 --reps 3 2>>gpurun_out/r03quad/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('no_quad=$nq', d['instances'], 'lpw', d['lpw'], 'ms %.3f' % d['kernel_ms'], d['events'], 'faulted', d['faulted'])"
done; done | tee gpurun_out/r03quad/quad.txt
