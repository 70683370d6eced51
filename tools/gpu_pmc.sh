#!/bin/bash
# PMC passes for the run kernel (each pass on its own; --kernel-trace only).
set -u
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- $CMD > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM
pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass grbm GRBM_GUI_ACTIVE
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ.get("OUTDIR") or "gpurun_out/%s" % (os.environ.get("TAG") or "pmc")
PY
python tools/pmc_summary.py $OUT > $OUT/summary.json 2> $OUT/summary.err; cat $OUT/summary.json
