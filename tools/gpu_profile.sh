#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line, rocprofv3 kernel stats and the PMC passes (each in its own
# run, --kernel-trace only) for the bench workload.  Outputs under gpurun_out/profile_<tag>/; copy what should be
# judged into profiles/<tag>/ (and profiles/current/pmc_traffic.json for bench.py's roofline.traffic).
set -u
TAG=${1:-rXX}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 5 --warmup 1 > $OUT/bench_line.json 2> $OUT/bench.err
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-instances 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats.log 2>&1
cp $OUT/stats/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- $CMD > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
pass sq3 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass grbm GRBM_GUI_ACTIVE
pass ifetch SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
python tools/pmc_summary.py $OUT > $OUT/pmc_lbft_k_run.json 2> $OUT/pmc_summary.err
# roofline.traffic for bench.py: FETCH/WRITE per launch, stamped with the hash of the kernels' machine code (copy to profiles/current/)
python - "$OUT" "$TAG" <<'PY'
import json, sys
sys.path.insert(0, ".")
from librabft_simulator_amd.build import source_hash
out, tag = sys.argv[1], sys.argv[2]
p = json.load(open(out + "/pmc_lbft_k_run.json"))
json.dump({"FETCH_SIZE": p.get("FETCH_SIZE"), "WRITE_SIZE": p.get("WRITE_SIZE"), "unit": "KB per launch of the run kernel (rocprofv3 --pmc, separate passes)",
           "profile": "profiles/%s/pmc_lbft_k_run.json" % tag, "workload": "bench.py default: 65536 x 4 nodes, max_clock 1000",
           "source_hash": source_hash()}, open(out + "/pmc_traffic.json", "w"), indent=1)
# roofline.issue for bench.py: the SQ passes, stamped the same way (copy to profiles/current/)
keys = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE")
json.dump(dict({k: p.get(k) for k in keys}, unit="per launch of the run kernel (rocprofv3 --pmc, separate passes)",
               profile="profiles/%s/pmc_lbft_k_run.json" % tag, workload="bench.py default: 65536 x 4 nodes, max_clock 1000", source_hash=source_hash()),
          open(out + "/pmc_issue.json", "w"), indent=1)
PY
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/tcc $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/grbm $OUT/ifetch
cat $OUT/kernel_stats.csv | head -4; cat $OUT/pmc_lbft_k_run.json | head -40; tail -1 $OUT/bench_line.json | cut -c1-600
