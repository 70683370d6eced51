set -u
export TMPDIR=/tmp
bash tools/gpu_r03_b.sh r03e "liblbft_hip.so:0:-1 liblbft_hip_ilp.so:0:-1 liblbft_hip_mmc.so:0:-1 liblbft_hip_iilp.so:0:-1 liblbft_hip_prio.so:0:-1 liblbft_hip_relax.so:0:-1 liblbft_hip.so:0:-1"
cp gpurun_out/r03e/sweep.jsonl gpurun_out/r03e/sweep_headline.jsonl
SWEEP_ARGS="--instances 1024" bash tools/gpu_r03_b.sh r03e "liblbft_hip.so:0:-1 liblbft_hip_ilp.so:0:-1 liblbft_hip_mmc.so:0:-1 liblbft_hip_iilp.so:0:-1 liblbft_hip_prio.so:0:-1"
cp gpurun_out/r03e/sweep.jsonl gpurun_out/r03e/sweep_1024.jsonl
for lib in liblbft_hip.so liblbft_hip_ilp.so liblbft_hip_mmc.so liblbft_hip_prio.so; do
  echo $lib; LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 300 python tools/configs.py c4_16384x64_longtail_equivocators c3shard_8192x4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ',d['config'], round(d['kernel_ms'],2), d['events'])"
done
