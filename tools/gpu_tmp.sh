set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03calls
timeout 900 python -m pytest tests/test_node_level.py -x -q -m gpu -s > gpurun_out/r03calls/pytest.log 2>&1; echo rc=$?; grep "node-level calls" gpurun_out/r03calls/pytest.log; tail -8 gpurun_out/r03calls/pytest.log
