#!/usr/bin/env python3
"""Generates tests/golden/full_size_digests.npz (test infrastructure; see tests/full_size_digest.py): the CPU oracle (oracle/lbft_oracle.cpp, math_mode 1)
run over every instance of the full-size configurations of tools/configs.py, one 64-bit digest per instance of (commit counts, active rounds, State
hashes) of all its nodes.  Hours of CPU time on 8 cores, minutes on the GPU box's 256 host threads -- which is where round 6 ran it
(`gpurun -- python tests/golden/gen_full_size.py --out gpurun_out/full_size_digests.npz --device`), never inside the test suite.

    python tests/golden/gen_full_size.py [names ...] [--count c5named...=1024] [--merge A.npz --merge B.npz] [--out PATH] [--device]
    python tests/golden/gen_full_size.py --none --merge A.npz --merge B.npz --out tests/golden/full_size_digests.npz     # only merge

--count NAME=K   only the first K instances of configuration NAME (prefix match) get a digest; the rest stay "not covered" (c5named: 49 core-seconds per
                 instance on the oracle)
--first K        start at instance K (the window [K, count): several machines share one configuration, their outputs are merged afterwards)
--merge PATH     start from an existing fixture: configurations / instances it already covers are kept, new ones added
--device         (GPU box) also run every configuration on the HIP path and report, per configuration, how many covered instances differ
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def save(path, table, meta):
    arrays = {"meta": np.array(json.dumps(meta))}
    for name, (dg, cov) in table.items():
        arrays[name] = dg
        arrays[name + "__covered"] = cov
    tmp = path + ".tmp.npz"
    np.savez(tmp, **arrays)
    os.replace(tmp, path)


def main():
    import full_size_digest as fsd
    import oracle_ctypes as oc
    from configs import CONFIGS
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=list(fsd.FULL_SIZE))
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--chunk", type=int, default=0, help="instances per oracle call (0 = 4 x threads)")
    ap.add_argument("--count", action="append", default=[], help="NAME=K: only the first K instances of NAME")
    ap.add_argument("--merge", action="append", default=[], help="existing fixture file(s) to start from (repeatable: the union of what they cover)")
    ap.add_argument("--out", default=fsd.FIXTURE)
    ap.add_argument("--device", action="store_true")
    ap.add_argument("--none", action="store_true", help="run nothing: merge the given fixtures and write the result")
    ap.add_argument("--first", type=int, default=0, help="start at this instance (with --count NAME=K: the window [first, K) -- several machines can share a configuration)")
    ap.add_argument("--save-every", type=int, default=8, help="write the fixture every N oracle calls (a run that may be cut off: 1)")
    ap.add_argument("--log", default=None)
    a = ap.parse_args()
    limits = {}
    for item in a.count:
        k, v = item.split("=")
        limits[k] = int(v)
    log = open(a.log, "a") if a.log else sys.stdout

    def say(s):
        log.write(s + "\n")
        log.flush()

    table, meta = {}, {"configs": {}, "digest": "blake2b-64(u32le commit_counts || u32le active_rounds || u64le last_states) per instance",
                       "oracle": "oracle/lbft_oracle.cpp, math_mode 1, seeds = instance index + 1"}
    for path in a.merge:
        if not os.path.exists(path):
            continue
        with np.load(path, allow_pickle=False) as z:
            m2 = json.loads(str(z["meta"]))
            for k, v in m2.get("configs", {}).items():
                if v.get("covered", 0) >= meta.get("configs", {}).get(k, {}).get("covered", 0):
                    meta.setdefault("configs", {})[k] = v
            for name in fsd.FULL_SIZE:
                if name in z.files:
                    dg2, cov2 = z[name].copy(), z[name + "__covered"].astype(bool)
                    if name in table:
                        dg1, cov1 = table[name]
                        both = cov1 & cov2
                        assert (dg1[both] == dg2[both]).all(), "the merged fixtures disagree on %s" % name
                        dg1[cov2] = dg2[cov2]
                        table[name] = (dg1, cov1 | cov2)
                    else:
                        table[name] = (dg2, cov2)
    try:
        meta["oracle_source_sha1"] = subprocess.check_output(["sha1sum", os.path.join(ROOT, "oracle", "lbft_oracle.cpp")]).decode().split()[0]
    except Exception:
        pass
    chunk = a.chunk or 4 * a.threads
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    if a.none:
        a.names = []
        for name, (dg, cov) in table.items():
            meta.setdefault("configs", {}).setdefault(name, {})["covered"] = int(cov.sum())
        save(a.out, table, meta)
        say("merged: " + ", ".join("%s %d" % (k, int(v[1].sum())) for k, v in table.items()))
    for name in a.names:
        c = CONFIGS[name]
        m = c["instances"]
        kw = fsd.oracle_kwargs(c)
        cfg = oc.make_config(math_mode=1, **kw)
        want = m
        for k, v in limits.items():
            if name.startswith(k):
                want = min(m, v)
        dg, cov = table.get(name, (np.zeros(m, dtype=np.uint64), np.zeros(m, dtype=bool)))
        dev = None
        if a.device:
            from librabft_simulator_amd import BatchSimulator, NodeConfig, RandomDelay
            seeds = np.arange(1, m + 1, dtype=np.uint64)
            delay = RandomDelay.uniform(*c["uniform"]) if "uniform" in c else RandomDelay.new(10.0, c.get("variance", 4.0))
            sim = BatchSimulator.new(seeds, c["nodes"], delay, NodeConfig(), commands_per_epoch=c.get("commands_per_epoch", 30000), voting_rights=c.get("weights"),
                                     equivocate_every=c.get("equivocate_every", 0), quirks=c.get("quirks", 0), rights_rotation=c.get("rights_rotation", 0))
            t0 = time.time()
            res = sim.loop_until(c["max_clock"])
            assert not res.faults.any()
            dev = fsd.digests(res.commit_counts, res.active_rounds, res.last_committed_states)
            ep = res.epochs
            say("%s: device %.1f ms (wall %.1f s), %.2f GB, events %d, commits/node min %d max %d, epochs min %d max %d" % (
                name, sim.last_run_ms()[1], time.time() - t0, sim.device_bytes() / 1e9, sum(res.counters["events"]), int(res.commit_counts.min()),
                int(res.commit_counts.max()), int(ep.min()), int(ep.max())))
            del sim, res
        t0, done, bad = time.time(), 0, 0
        for lo in range(min(a.first, want), want, chunk):
            hi = min(want, lo + chunk)
            todo = np.nonzero(~cov[lo:hi])[0] + lo
            if len(todo):
                ref = oc.run_batch(cfg, (todo + 1).astype(np.uint64), c["max_clock"], threads=a.threads, history_cap=0)
                dg[todo] = fsd.digests(ref["commit_counts"], ref["active_rounds"], ref["last_states"])
                cov[todo] = True
                done += len(todo)
            if dev is not None:
                bad += int((dev[lo:hi] != dg[lo:hi]).sum())
            if ((lo - min(a.first, want)) // chunk) % a.save_every == a.save_every - 1 or hi == want:
                say("  %s [%d, %d) oracle %.0f s%s" % (name, 0, hi, time.time() - t0, "" if dev is None else ", device mismatches so far %d" % bad))
                table[name] = (dg, cov)
                save(a.out, table, meta)
        table[name] = (dg, cov)
        meta["configs"][name] = {"config": {k: v for k, v in c.items() if k != "weights"}, "weights": "1 + (i mod 4)" if c.get("weights") else None,
                                 "covered": int(cov.sum()), "instances": m, "oracle_seconds": round(time.time() - t0, 1), "threads": a.threads}
        save(a.out, table, meta)
        say("%s: %d of %d instances covered (%d new, %.0f s on %d threads)%s" % (
            name, int(cov.sum()), m, done, time.time() - t0, a.threads,
            "" if dev is None else "; device == oracle on all covered instances" if bad == 0 else "; DEVICE DIFFERS on %d instances" % bad))


if __name__ == "__main__":
    main()
