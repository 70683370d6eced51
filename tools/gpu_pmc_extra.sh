#!/bin/bash
# Runs on the GPU box (via gpurun): extra PMC passes (instruction cache, issue/wait breakdown) for the bench workload.
set -u
OUT=gpurun_out/pmc_extra
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_0-9]*\|SQ_[A-Z_0-9]*" | sort -u > $OUT/counters.txt
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- $CMD > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log; }
pass ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
pass lvl SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM
for p in ic wait lvl; do python tools/pmc_summary.py $OUT/$p lbft_k_run > $OUT/$p.json 2>> $OUT/summary.err; cat $OUT/$p.json; done
rm -rf $OUT/ic $OUT/wait $OUT/lvl
wc -l $OUT/counters.txt
