# scratch driver for one gpurun call: the whole device suite, then the smoke entry point
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/suite
( time timeout 1200 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/suite/pytest_gpu.txt 2>&1; echo rc=$?; tail -6 gpurun_out/suite/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
