# Round 6, last session, closing call (the library's machine code is the one profiles/r06 and profiles/current describe -- hash ac9bd7b41025767f -- so the kernel
# stats / PMC passes are not repeated): smoke(), the whole device suite on the final tree (new fixture digests, corrected fuzz harness), the default bench line,
# and the bench's parity gate on ALL 65 536 instances of three other seed ranges.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r06close}; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_full_suite_last_session.txt 2>&1; tail -3 $O/pytest_gpu_full_suite_last_session.txt
timeout 900 python bench.py > $O/bench_line_last_session.json 2> $O/bench.err; python - $O <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_line_last_session.json").read().strip().splitlines()[-1])
print("bench", round(d["value"] / 1e6, 1), "M rounds/s", d["ms_per_step"], "ms", "kernel", d["roofline"]["kernel_ms"], "frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], "parity", d["parity"]["mismatches"], "of", d["parity"]["checked_instances"], "issue", d["roofline"]["issue"].get("source"))
PY
for seed in 1000003 250000001 4000000007; do
  timeout 600 python bench.py --steps 2 --warmup 1 --base-seed $seed --parity-instances 65536 --no-cpu-baseline --no-measure-traffic >> $O/headline_parity_other_seed_ranges.jsonl 2>> $O/bench.err
done
python - $O <<'PY'
import json, sys
for l in open(sys.argv[1] + "/headline_parity_other_seed_ranges.jsonl"):
    d = json.loads(l)
    print("seed range parity:", d["parity"]["checked_instances"], "instances,", d["parity"]["mismatches"], "mismatches, rounds/s", round(d["value"] / 1e6, 1), "M")
PY
# ... and, with what is left of the round's GPU time, more oracle digests of c5named on the box's host cores, each compared with the device's result in the same process (--device)
if [ -n "${DIGEST_FIRST:-}" ]; then
  timeout ${DIGEST_SECS:-1300} python tests/golden/gen_full_size.py c5named_8192x100_weighted_epoch_every_50_commits --first $DIGEST_FIRST --count c5named=$DIGEST_LAST --out $O/c5named_${DIGEST_FIRST}_${DIGEST_LAST}.npz \
    --threads 64 --chunk 128 --save-every 1 --device --log $O/c5named_digests.log > $O/c5named_digests.out 2>&1
  tail -3 $O/c5named_digests.log
fi
