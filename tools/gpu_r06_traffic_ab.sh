#!/bin/bash
# Round 6, third session: HBM traffic (2 x FETCH_SIZE + WRITE_SIZE of the run kernel, separate --pmc passes) of one configuration on several libraries.
#   bash tools/gpu_r06_traffic_ab.sh <tag> "<libs>" "<configs>"
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06tr}; LIBS=$2; CFGS=$3
mkdir -p $O
for lib in $LIBS; do for cfg in $CFGS; do
  export LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_tmp; timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_tmp -o pmc --output-format csv -- python tools/configs.py $cfg > $O/pmc.log 2>&1
    python - "$O/pmc_tmp" "$lib" "$cfg" "$c" >> $O/traffic_ab.txt <<'PY'
import csv, glob, sys
tot = 0.0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lbft_k_run" in r.get("Kernel_Name", "") and r.get("Counter_Name") == sys.argv[4]:
            tot += float(r["Counter_Value"])
print(sys.argv[2], sys.argv[3][:12], sys.argv[4], "KB", round(tot), "GB", round(tot * 1024 / 1e9, 1))
PY
  done
done; done
rm -rf $O/pmc_tmp
cat $O/traffic_ab.txt
