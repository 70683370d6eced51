set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03quad2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_model.py tests/test_save_node.py -x -q -m gpu -k "not full_size_config and not full_batch" > gpurun_out/r03quad2/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03quad2/pytest.log
for m in 1024 8192 32768 65536; do
  timeout 200 python tools/sweep.py --one --instances $m --lpw 0 --ql -1 --reps 3 2>>gpurun_out/r03quad2/err.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['instances'], 'lpw', d['lpw'], 'ql', d['ql'], 'ms %.3f' % d['kernel_ms'], d['events'], 'faulted', d['faulted'])"
done | tee gpurun_out/r03quad2/quad.txt
