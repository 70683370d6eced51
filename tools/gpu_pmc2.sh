#!/bin/bash
set -u
OUT=gpurun_out/pmc_icache
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAVES_[A-Z_]*\|SQC_DCACHE[A-Z_]*" | sort -u > $OUT/available.txt
cat $OUT/available.txt | tr '\n' ' '
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- $CMD > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log; }
pass ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass ic2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass ic3 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
python tools/pmc_summary.py $OUT
