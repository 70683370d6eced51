#!/bin/bash
# Runs on the GPU box (via gpurun): BASELINE.json's configurations 2-5 (+ the live variants of 4 and 5) through
# tools/configs.py, then rocprofv3 kernel stats and PMC passes (each in its own run, --kernel-trace only) for the
# large-network kernels.  Outputs under gpurun_out/configs_<tag>/; copy what should be judged into profiles/<tag>/.
#   bash tools/gpu_configs_profile.sh r02 "c4_16384x64_longtail_equivocators c4live_16384x64_longtail_equivocators_fixed"
set -u
TAG=${1:-rXX}
PROF_CONFIGS=${2:-"c2_1024x4_lognormal c2_1024x4_uniform c3_65536x4 c3shard_8192x4 c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed c5named_8192x100_weighted_epoch_every_50_commits"}
OUT=gpurun_out/configs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# (first pass: timings; the lines are rewritten with each configuration's own PMC traffic at the end)
timeout 1200 python tools/configs.py c2_1024x4_lognormal c2_1024x4_uniform c3_65536x4 c3shard_8192x4 c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs \
  c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed c5named_8192x100_weighted_epoch_every_50_commits > $OUT/baseline_configs.jsonl 2> $OUT/configs.err
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1] + "/baseline_configs.jsonl"):
    d = json.loads(line)
    print(d["config"], "ms", round(d["kernel_ms"], 2), "ev/s %.3g" % d["events_per_s"], "faulted", d["faulted_instances"], d["liveness"], "frac", round(d["roofline"]["frac"], 4),
          "frac_ref_equiv", round(d["roofline"]["frac_reference_equivalent"], 4), d["roofline"]["kernel"], "GB", round(d["device_gb"], 1))
PY
pass() { cfg=$1; name=$2; shift 2; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$cfg.$name -o pmc --output-format csv -- python tools/configs.py $cfg > $OUT/$cfg.$name.log 2>&1 || tail -3 $OUT/$cfg.$name.log; }
for cfg in $PROF_CONFIGS; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/$cfg.stats -o stats --output-format csv -- python tools/configs.py $cfg > $OUT/$cfg.stats.log 2>&1
  cp $OUT/$cfg.stats/*kernel_stats.csv $OUT/$cfg.kernel_stats.csv 2>/dev/null
  pass $cfg fetch FETCH_SIZE
  pass $cfg write WRITE_SIZE
  pass $cfg sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
  pass $cfg sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
  python tools/pmc_summary.py $OUT "lbft_k_run" $cfg > $OUT/$cfg.pmc.json 2>> $OUT/pmc_summary.err
  rm -rf $OUT/$cfg.stats $OUT/$cfg.fetch $OUT/$cfg.write $OUT/$cfg.sq2 $OUT/$cfg.sq1
  head -3 $OUT/$cfg.kernel_stats.csv; head -30 $OUT/$cfg.pmc.json
done
# second pass of the lines, now with `roofline.traffic` from each configuration's own FETCH / WRITE passes
LBFT_PMC_DIR=$OUT timeout 1200 python tools/configs.py $(echo c2_1024x4_lognormal c2_1024x4_uniform c3_65536x4 c3shard_8192x4 $PROF_CONFIGS | tr " " "\n" | awk '!s[$0]++' | tr "\n" " ") > $OUT/baseline_configs_with_traffic.jsonl 2>> $OUT/configs.err
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1] + "/baseline_configs_with_traffic.jsonl"):
    d = json.loads(line); r = d["roofline"]
    print(d["config"], "ms", round(d["kernel_ms"], 2), "frac", round(r["frac"], 4), "ref_equiv", round(r["frac_reference_equivalent"], 4), "traffic GB", r["traffic"], "t/ref", r.get("traffic_over_reference_equivalent"), "t/exec", r.get("traffic_over_executed"))
PY
