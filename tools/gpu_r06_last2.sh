# Round 6, last session, third call: (1) the seven chunks of the widened device fuzz that stopped on a HARNESS limit (block / snapshot pool too small for the drawn
# network -- the device raised the fault word, the host build does the same at the same capacity and equals the oracle with room -- and the "at least half of the
# draws on the headline kernel" guard) re-run with the corrected harness; (2) more oracle digests of c5named on the box's host cores (window [FIRST, LAST)).
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06fz3}; mkdir -p $O
T=tests/test_fuzz_model.py
LBFT_FUZZ_GPU_FIRST=149 LBFT_FUZZ_GPU_CHUNKS=47 LBFT_FUZZ_GPU_QUAD_FIRST=66 LBFT_FUZZ_GPU_QUAD_CHUNKS=57 timeout 600 python -m pytest -q -m gpu \
  "$T::test_random_configurations_on_the_device_match_the_oracle[149]" "$T::test_random_configurations_on_the_device_match_the_oracle[176]" "$T::test_random_configurations_on_the_device_match_the_oracle[195]" \
  "$T::test_random_headline_network_configurations_on_the_device_match_the_oracle[66]" "$T::test_random_headline_network_configurations_on_the_device_match_the_oracle[70]" \
  "$T::test_random_headline_network_configurations_on_the_device_match_the_oracle[93]" "$T::test_random_headline_network_configurations_on_the_device_match_the_oracle[114]" \
  "$T::test_random_headline_network_configurations_on_the_device_match_the_oracle[122]" > $O/device_fuzz_rerun_of_the_harness_limited_chunks.txt 2>&1
tail -3 $O/device_fuzz_rerun_of_the_harness_limited_chunks.txt
FIRST=${FIRST:-2048}; LAST=${LAST:-2944}; SECS=${SECS:-2000}
timeout $SECS python tests/golden/gen_full_size.py c5named_8192x100_weighted_epoch_every_50_commits --first $FIRST --count c5named=$LAST --out $O/c5named_${FIRST}_${LAST}.npz \
  --threads ${THREADS:-64} --chunk ${CHUNK:-128} --save-every 1 --log $O/c5named_digests.log > $O/c5named_digests.out 2>&1
tail -4 $O/c5named_digests.log
