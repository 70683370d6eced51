# Round 6, last session: what the cooperative runs retire ON THE DEVICE (liblbft_hip_runcount.so = build_variant("runcount", ["-DLBFT_PHASE_TIMERS", "-DLBFT_RUN_COUNT"]):
# no timers, LDS counters per wavefront) for the five large configurations -> tools/configs.py::RUN_SHARES.  The results must equal the product library's
# (same counters, same commits: checked below against a product run of the same configuration).
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06rc}; mkdir -p $O
CFGS="${CFGS:-c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed c5named_8192x100_weighted_epoch_every_50_commits}"
for cfg in $CFGS; do
  LBFT_RUN_COUNT=1 LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_runcount.so timeout 600 python tools/configs.py $cfg >> $O/run_counts_device.jsonl 2>> $O/run_counts.err
  case $cfg in c5named*) ;; *) timeout 600 python tools/configs.py $cfg >> $O/product_same_call.jsonl 2>> $O/run_counts.err ;; esac
done
python - $O <<'PY'
import json, sys
prod = {}
try:
    for l in open(sys.argv[1] + "/product_same_call.jsonl"):
        d = json.loads(l); prod[d["config"]] = d
except FileNotFoundError:
    pass
for l in open(sys.argv[1] + "/run_counts_device.jsonl"):
    d = json.loads(l)
    r = d.get("runs_device")
    p = prod.get(d["config"])
    same = None if p is None else (p["events_by_kind"] == d["events_by_kind"] and p["commits"] == d["commits"] and p["rounds"] == d["rounds"])
    print(d["config"][:14], "ms", round(d["kernel_ms"]), "product ms", round(p["kernel_ms"]) if p else None, "same counters as product:", same)
    if r:
        print("    share", {k: round(v, 4) for k, v in r["share_of_kind"].items()}, "per run", {k: round(v, 1) for k, v in r["events_per_run"].items()}, "wave steps", r["wave_steps"])
PY
