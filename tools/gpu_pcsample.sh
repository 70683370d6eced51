#!/bin/bash
# Runs on the GPU box (via gpurun): PC sampling of the run kernel of the bench workload (rocprofv3 beta feature), every attempt under
# its own timeout.  The library sampled is a -gline-tables-only build (same machine code + line tables), so that samples aggregate per
# source line.  Outputs: gpurun_out/pcs_<tag>/{avail.txt,<method>.json,<method>.log}
#   bash tools/gpu_pcsample.sh r04 liblbft_hip_dbg.so "stochastic:cycles:65536 host_trap:time:50"
set -u
TAG=${1:-rXX}
LIB=${2:-liblbft_hip_dbg.so}
TRIES=${3:-"stochastic:cycles:65536 host_trap:time:50"}
OUT=gpurun_out/pcs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 60 rocprofv3 -L > $OUT/avail.txt 2>&1
grep -i -B2 -A12 "pc.sampl" $OUT/avail.txt | head -60
for t in $TRIES; do
  m=$(echo $t | cut -d: -f1); u=$(echo $t | cut -d: -f2); iv=$(echo $t | cut -d: -f3)
  d=/tmp/pcs_${m}_$iv
  rm -rf $d
  LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$LIB timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $u --pc-sampling-interval $iv \
    --kernel-trace -d $d -o pcs --output-format csv -- python tools/sweep.py --one --instances ${INSTANCES:-65536} --reps 2 > $OUT/${m}_$iv.log 2>&1
  echo "$t rc=$?"; tail -2 $OUT/${m}_$iv.log | cut -c1-300
  ls -la $d/* 2>/dev/null | head
  timeout 300 python tools/pcsample_summary.py $d $OUT/${m}_$iv.json 2>&1 | tail -60
  rm -rf $d
done
