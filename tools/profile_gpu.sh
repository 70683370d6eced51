#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
STEPS=${2:-3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats.log 2>&1
# PMC passes, each on its own (no tracing domains besides kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc --output-format csv -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o pmc --output-format csv -- $CMD > $OUT/pmc_tcc.log 2>&1
find $OUT -name '*.csv' | head -50
