set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03live
LBFT_HIP_LIB=$PWD/librabft_simulator_amd/liblbft_hip_prof.so timeout 600 python tools/configs.py c4live_16384x64_longtail_equivocators_fixed c4_16384x64_longtail_equivocators --scale 0.5 > gpurun_out/r03live/phases.jsonl 2>gpurun_out/r03live/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r03live/phases.jsonl"):
    d=json.loads(l); tot=d.get("cycles_per_wave_step",0)
    print(d["config"][:20], "ms %.1f"%d["kernel_ms"], "cyc/step", tot)
    for k,v in sorted(d.get("phases",{}).items(), key=lambda kv:-kv[1])[:14]: print("   %-18s %5.1f%%  %6.0f cyc"%(k,100*v,v*tot))
PY
