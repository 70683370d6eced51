set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03f4b
timeout 1200 python -m pytest tests/test_save_node.py tests/test_reference_unit_scenarios.py tests/test_abi.py tests/test_node_level.py tests/test_fuzz_model.py -x -q -m gpu > gpurun_out/r03f4b/pytest.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03f4b/pytest.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size and not full_batch" > gpurun_out/r03f4b/pytest2.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03f4b/pytest2.log
