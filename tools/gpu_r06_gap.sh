# Round 6, last session, last call: the window [1024, 2048) of c5named -- what the build container's cores had computed by then (tools/_scratch/c5named_local.npz)
# merged, the rest computed on the box's host cores, ALL of the window compared with the device's result in the same process (--device).
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06gap}; mkdir -p $O
timeout ${SECS:-960} python tests/golden/gen_full_size.py c5named_8192x100_weighted_epoch_every_50_commits --merge tools/_scratch/c5named_local.npz --first 1024 --count c5named=2048 \
  --out $O/c5named_1024_2048.npz --threads 64 --chunk 128 --save-every 1 --device --log $O/c5named_digests.log > $O/c5named_digests.out 2>&1
tail -4 $O/c5named_digests.log
