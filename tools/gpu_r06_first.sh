#!/bin/bash
# Round 6, the session's first GPU call: (a) in the background on the box's host threads, oracle digests of more c5named instances for the full-size
# fixture (tests/golden/gen_full_size.py); (b) the whole device suite with durations; (c) phase profiles of the large-network configurations on the
# diagnostic build (liblbft_hip_prof.so = build_variant("prof", ["-DLBFT_PHASE_TIMERS"])); (d) the default bench line.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06e}
N5=${2:-1024}
mkdir -p $O
# (blocking: what the suite's fixture checks need -- every c4 instance and the first 256 of c5named; then the rest of c5named in the background)
python tests/golden/gen_full_size.py c4_16384x64_longtail_equivocators c5named_8192x100_weighted_epoch_every_50_commits --count c5named=256 --threads 224 --chunk 256 \
   --merge tests/golden/full_size_digests.npz --out $O/full_size_digests_first.npz --log $O/gen_full_size.log > $O/gen_full_size.err 2>&1; echo "gen(first) rc=$?" >> $O/gen_full_size.log
cp $O/full_size_digests_first.npz tests/golden/full_size_digests.npz
timeout 1500 python -m pytest tests -m gpu -q --durations=30 > $O/pytest_gpu_full_suite.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_full_suite.txt; tail -5 $O/pytest_gpu_full_suite.txt
(nice -n 5 python tests/golden/gen_full_size.py c5named_8192x100_weighted_epoch_every_50_commits --count c5named=$N5 --threads 160 --chunk 256 --merge $O/full_size_digests_first.npz \
   --out $O/full_size_digests.npz --log $O/gen_full_size.log >> $O/gen_full_size.err 2>&1; echo "gen rc=$?" >> $O/gen_full_size.log) &
GEN=$!
if [ -f librabft_simulator_amd/liblbft_hip_prof.so ]; then
  for cfg in c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed; do
    LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_prof.so timeout 300 python tools/configs.py $cfg >> $O/phases_large_networks.jsonl 2>> $O/phases.err
  done
fi
timeout 900 python tools/configs.py c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed \
  c5live_8192x100_rotating_rights_epochs_fixed c5named_8192x100_weighted_epoch_every_50_commits > $O/large_configs.jsonl 2> $O/large_configs.err
python - $O <<'PY'
import json, sys
for f in ("phases_large_networks.jsonl", "large_configs.jsonl"):
    try:
        for l in open(sys.argv[1] + "/" + f):
            d = json.loads(l)
            print(f[:6], d["config"][:14], "ms", round(d["kernel_ms"], 1), "frac", round(d["roofline"]["frac"], 4), "GB", round(d["device_gb"], 1), "lpw", d["layout"]["lanes_per_wavefront"], d["liveness"]["epochs_min_max"])
            if "phases" in d:
                print("    ", " ".join("%s=%.3f" % p for p in sorted(d["phases"].items(), key=lambda x: -x[1])[:14]), "cyc/step", round(d.get("cycles_per_wave_step", 0)))
    except Exception as e:
        print(f, "failed", e)
PY
wait $GEN
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err
python - $O <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("bench: value %.4g ms_per_step %.3f kernel_ms %.3f frac %.4f traffic %s (%s) cpu %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"], str(r["traffic_source"])[:40], d.get("cpu_baseline", {}).get("value")))
PY
tail -4 $O/gen_full_size.log
