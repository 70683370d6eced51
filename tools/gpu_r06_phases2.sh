# Round 6, third session: phase shares of the four large configurations on the phase-timer build (-DLBFT_PHASE_TIMERS; ~2 x slower: shares, not times)
# after the record exchange's response runs and the "active" notification runs.  Runs are charged to: request -> "request", response -> "response",
# notification -> "snap_release" (tools/sweep.py PHASES).
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06u}; mkdir -p $O
for cfg in c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs c4live_16384x64_longtail_equivocators_fixed c5live_8192x100_rotating_rights_epochs_fixed; do
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_prof.so timeout 300 python tools/configs.py $cfg >> $O/phases_after_active_notification_runs.jsonl 2>> $O/phases.err
done
python - $O <<'PY'
import json, sys
for l in open(sys.argv[1] + "/phases_after_active_notification_runs.jsonl"):
    d = json.loads(l)
    print(d["config"][:12], "prof ms", round(d["kernel_ms"]), "wave steps", d["wave_steps"], "cyc/step", round(d["cycles_per_wave_step"]))
    print("   ", " ".join("%s=%.3f" % p for p in sorted(d["phases"].items(), key=lambda x: -x[1])[:18]))
PY
