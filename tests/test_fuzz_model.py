"""Randomised differential test (seeded, CPU-only): the kernel logic compiled for the host (every size class, queue
discipline and extension) against the full-fidelity oracle on configurations drawn at random -- node counts, voting
rights, delay models, pacemaker parameters, epoch lengths, equivocators, message loss, partitions, Q2 on/off."""
import os

import numpy as np
import pytest


def draw_config(rng):
    n = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 6, 7, 8, 9, 12, 16, 17, 24, 33, 40]))
    kw = dict(num_nodes=n)
    if rng.random() < 0.3:
        kw.update(delay_model=1, uniform_lo=int(rng.integers(0, 8)), uniform_hi=int(rng.integers(8, 30)))
    else:
        mean = float(rng.choice([3.0, 10.0, 10.0, 25.0]))
        kw.update(mean=mean, variance=float(rng.choice([0.0, 4.0, 4.0, 100.0, 400.0])))
    if rng.random() < 0.4:
        kw["voting_rights"] = [int(v) for v in rng.integers(1, 6, n)]
    if rng.random() < 0.4:
        kw["commands_per_epoch"] = int(rng.choice([3, 7, 20, 50]))
    if rng.random() < 0.5:
        # (lambda * delta >= 2: a query period of 0 makes every update query all peers and the event population explode)
        kw.update(delta=int(rng.choice([5, 10, 20, 40])), gamma=float(rng.choice([1.0, 1.5, 2.0])), lambda_=float(rng.choice([0.5, 1.0])))
    if rng.random() < 0.3:
        kw["target_commit_interval"] = int(rng.choice([50, 200, 100000]))
    if rng.random() < 0.3 and n >= 2:
        kw["equivocate_every"] = int(rng.integers(1, n + 1))
    if rng.random() < 0.3:
        kw["drop_per_million"] = int(rng.choice([1000, 50000, 300000]))
    if rng.random() < 0.25 and n >= 2:
        start = int(rng.integers(0, 300))
        kw.update(partition_size=int(rng.integers(1, n)), partition_start=start, partition_end=start + int(rng.integers(50, 400)))
    if rng.random() < 0.45:
        kw["quirks"] = int(rng.choice([1, 2, 3]))
    if "voting_rights" in kw and "commands_per_epoch" in kw and n >= 2 and rng.random() < 0.6:
        kw["rights_rotation"] = int(rng.integers(1, n))  # epoch reconfiguration: the rights rotate with the epoch
    return kw


@pytest.mark.parametrize("chunk", range(18))
def test_random_configurations_match_the_oracle(oracle, chunk):
    rng = np.random.default_rng(20240 + chunk)
    for _ in range(16):
        kw = draw_config(rng)
        n = kw["num_nodes"]
        max_clock = int(rng.choice([300, 600, 1000])) if n <= 16 else 250
        m = 6 if n <= 16 else 2
        seeds = rng.integers(1, 2 ** 62, m, dtype=np.uint64)
        cfg = oracle.make_config(math_mode=1, **kw)
        a = oracle.run_batch(cfg, seeds, max_clock, threads=4, history_cap=96)
        special = any(k in kw for k in ("equivocate_every", "drop_per_million", "partition_size")) or (kw.get("quirks", 0) & 1)
        big = n > 4
        qheap = 1 if (big or rng.random() < 0.3) else 0
        qcal = 1 if (rng.random() < 0.5 and (qheap or special or n > 16)) else 0
        # <= 64 snapshot slots: the register-resident free mask (what the device picks for small honest networks)
        scap = 64 if (n <= 4 and not special and rng.random() < 0.5) else max(128, 128 * n)
        b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=4, history_cap=96, qcap=max(4096, 24 * n * n), scap=scap,
                                       bcap=1024, lcap=1024, ql=int(rng.choice([0, 3, 11, 48])), qheap=qheap, qcal=qcal,
                                       force_generic=int(rng.random() < 0.2), hash_cap=1024 if n <= 16 else 0)
        assert not b["faults"].any(), kw
        for key in ("commit_counts", "active_rounds", "last_states", "histories"):
            assert (a[key] == b[key]).all(), (key, kw)
        for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
            assert a["counters"][key] == b["counters"][key], (key, kw)
        if n <= 16:  # the reference's record hashes of the first network's committed chains (block, State, QC with its votes)
            sim = oracle.OracleSim(cfg, int(seeds[0])).run_until(max_clock)
            for node in range(n):
                ref = sim.committed_record_hashes(node)
                got = b["record_hashes"][0, node, :len(ref)]
                assert (got[:, 0] == ref["block_hash"]).all() and (got[:, 1] == ref["state"]).all(), kw
                assert (got[:, 2] == ref["qc_hash"]).all() and ((got[:, 3] >> 32) == 0).all() and ref["has_qc"].all(), kw


def draw_large_config(rng):
    kw = draw_config(rng)
    n = int(rng.choice([33, 36, 40, 48, 64, 65, 66, 80, 100, 128]))
    kw["num_nodes"] = n
    for k in ("voting_rights", "equivocate_every", "partition_size", "partition_start", "partition_end", "rights_rotation"):
        kw.pop(k, None)
    if rng.random() < 0.4:
        kw["voting_rights"] = [int(v) for v in rng.integers(1, 6, n)]
        if "commands_per_epoch" in kw and rng.random() < 0.6:
            kw["rights_rotation"] = int(rng.integers(1, n))
    if rng.random() < 0.4:
        kw["equivocate_every"] = int(rng.integers(2, 9))
    if rng.random() < 0.5:
        kw.pop("drop_per_million", None)
    max_clock = int(rng.choice([120, 200, 260])) if n <= 66 else 130
    return kw, n, max_clock


@pytest.mark.parametrize("chunk", range(8))
def test_random_large_configurations_on_the_cooperative_loop(oracle, chunk):
    """The same generator for networks of 33..128 nodes, run through the cooperative event loop of the large-network kernels
    (run_coop / coop_bulk on 64 emulated lanes): ring sizes, top-up rates, two-pass receiver lists (n > 65), equivocators,
    partitions, all quirk modes.  (Random message loss keeps the lane-per-network loop: coop() excludes it.)"""
    rng = np.random.default_rng(90210 + chunk)
    for _ in range(5):
        kw, n, max_clock = draw_large_config(rng)
        seeds = rng.integers(1, 2 ** 62, 1, dtype=np.uint64)
        cfg = oracle.make_config(math_mode=1, **kw)
        a = oracle.run_batch(cfg, seeds, max_clock, threads=4, history_cap=64)
        b = oracle.hostmodel_run_batch(cfg, seeds, max_clock, threads=4, history_cap=64, qcap=max(8192, 32 * n * n),
                                       scap=min(65535, 6 * n * n + 16 * n) if kw.get("quirks", 0) & 1 else 128 * n, bcap=512, lcap=512, ql=0, qheap=1, qcal=1,
                                       ring=int(rng.choice([128, 256, 512])), ring_topup=int(rng.choice([0, 4, 16])))
        assert not b["faults"].any(), (kw, b["faults"])
        for key in ("commit_counts", "active_rounds", "last_states", "histories"):
            assert (a[key] == b[key]).all(), (key, kw)
        for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
            assert a["counters"][key] == b["counters"][key], (key, kw)


# LBFT_FUZZ_GPU_CHUNKS=n widens the device run (10 configurations per chunk; the default keeps `pytest -m gpu` short); LBFT_FUZZ_GPU_FIRST=k starts at chunk k
# (other draws than an earlier widened run; likewise ..._QUAD_FIRST / ..._LARGE_FIRST below)
@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(int(os.environ.get("LBFT_FUZZ_GPU_FIRST", "0")), int(os.environ.get("LBFT_FUZZ_GPU_FIRST", "0")) + int(os.environ.get("LBFT_FUZZ_GPU_CHUNKS", "8"))))
def test_random_configurations_on_the_device_match_the_oracle(oracle, chunk):
    import librabft_simulator_amd as amd
    rng = np.random.default_rng(777 + chunk)
    for _ in range(10):
        kw = draw_config(rng)
        n = kw["num_nodes"]
        max_clock = int(rng.choice([300, 600, 1000, 2500])) if n <= 16 else 250
        m = int(rng.choice([3, 40, 70])) if n <= 16 else 3
        seeds = rng.integers(1, 2 ** 62, m, dtype=np.uint64)
        ref = oracle.run_batch(oracle.make_config(math_mode=1, **kw), seeds, max_clock, threads=8, history_cap=96)
        delay = amd.RandomDelay.uniform(kw["uniform_lo"], kw["uniform_hi"]) if kw.get("delay_model") == 1 else \
            amd.RandomDelay.new(kw.get("mean", 10.0), kw.get("variance", 4.0))
        nc = amd.NodeConfig(kw.get("target_commit_interval", 100000), kw.get("delta", 20), kw.get("gamma", 2.0), kw.get("lambda_", 0.5))
        part = (kw["partition_size"], kw["partition_start"], kw["partition_end"]) if "partition_size" in kw else None
        sim = amd.BatchSimulator.new(seeds, n, delay, nc, commands_per_epoch=kw.get("commands_per_epoch", 30000),
                                     voting_rights=kw.get("voting_rights"), equivocate_every=kw.get("equivocate_every", 0),
                                     drop_per_million=kw.get("drop_per_million", 0), partition=part, quirks=kw.get("quirks", 0),
                                     rights_rotation=kw.get("rights_rotation", 0),
                                     calendar_queue=bool(rng.random() < 0.7), max_steps_per_launch=int(rng.choice([0, 0, 173])),
                                     lanes_per_wavefront=int(rng.choice([0, 0, 1, 2, 8, 16, 32, 64])),
                                     # (equivocators propose twice per round: chunk 149 -- 3 nodes, rights [1, 1, 5], the heavy node committing alone every 1.7 ticks
                                     # and equivocating -- needs 1 204 block rows by clock 1 000; with max_clock + 64 the device raised F_BLOCK_OVERFLOW, as the host build does)
                                     block_capacity=(2 * max_clock + 256) if kw.get("equivocate_every", 0) else max_clock + 64,
                                     queue_capacity=max(4096, 64 * n * n),
                                     # 0 = automatic (<= 64 slots for small honest networks: the register-resident free mask)
                                     # (the record exchange on networks of > 16 nodes with a short query period keeps a query-all's responses of every node in
                                     # flight: chunks 176 / 195 -- 33 / 40 nodes, delta 5, quirks 1, a partition -- exhaust 128 n slots, F_SNAP_OVERFLOW on the
                                     # device and on the host build alike, and equal the oracle at 6 n^2 + 16 n: the large-network harness's capacity)
                                     snapshot_capacity=0 if (not (kw.get("quirks", 0) & 1) and rng.random() < 0.5) else
                                     (min(65535, 6 * n * n + 16 * n) if (n > 16 and kw.get("quirks", 0) & 1) else max(128, 128 * n)))
        res = sim.loop_until(max_clock, allow_faults=True)
        assert not res.faults.any(), (kw, sorted(set(int(f) for f in res.faults)), res.counters, sim.layout())
        assert (res.commit_counts == ref["commit_counts"]).all(), kw
        assert (res.active_rounds == ref["active_rounds"]).all(), kw
        assert (res.last_committed_states == ref["last_states"]).all(), kw
        assert (res.committed_histories(96) == ref["histories"]).all(), kw
        c, rc = res.counters, ref["counters"]
        for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
            assert c[key] == rc[key], (key, kw)


# The headline network's own kernel (lbft_k_run0q: 4 nodes, unit voting rights, log-normal delays, <= 64 snapshot slots, fixed at
# compile time; pairs of lanes scanning a queue): everything else about the configuration drawn at random -- pacemaker parameters,
# epoch lengths, delay mean / variance, equivocators, loss, partitions, Q2 -- with 1 .. 64 networks per wavefront and launches
# cut into pieces.  At least half of the draws must have run on that kernel -- or, one network per wavefront (batches this small then run lbft_k_run0u), on the
# scalar-unit kernel -- (the rest: the general small-network kernels).
@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(int(os.environ.get("LBFT_FUZZ_GPU_QUAD_FIRST", "0")), int(os.environ.get("LBFT_FUZZ_GPU_QUAD_FIRST", "0")) + int(os.environ.get("LBFT_FUZZ_GPU_QUAD_CHUNKS", "5"))))
def test_random_headline_network_configurations_on_the_device_match_the_oracle(oracle, chunk):
    import librabft_simulator_amd as amd
    rng = np.random.default_rng(31337 + chunk)
    on_quad = overflowed = 0
    for _ in range(10):
        kw = draw_config(rng)
        kw["num_nodes"] = n = 4
        for k in ("voting_rights", "rights_rotation", "delay_model", "uniform_lo", "uniform_hi"):
            kw.pop(k, None)
        if rng.random() < 0.8:
            # what the packed-queue kernels (class 0, this one included) leave to the general small-network kernel: equivocators, loss,
            # partitions, peers answering requests (Q1)
            for k in ("equivocate_every", "drop_per_million", "partition_size", "partition_start", "partition_end"):
                kw.pop(k, None)
            kw["quirks"] = kw.get("quirks", 0) & 2
        if kw.get("equivocate_every", 0) > n:
            kw["equivocate_every"] = n
        if "partition_size" in kw:
            kw["partition_size"] = min(kw["partition_size"], n - 1)
        kw.setdefault("mean", float(rng.choice([3.0, 10.0, 25.0])))
        kw.setdefault("variance", float(rng.choice([0.0, 4.0, 100.0])))
        max_clock = int(rng.choice([300, 600, 1000, 2500]))
        m = int(rng.choice([129, 300, 700, 1100]))  # (1 100 with one lane per wavefront: lbft_k_run0q; smaller batches there: lbft_k_run0u)
        seeds = rng.integers(1, 2 ** 62, m, dtype=np.uint64)
        ref = oracle.run_batch(oracle.make_config(math_mode=1, **kw), seeds, max_clock, threads=8, history_cap=96)
        nc = amd.NodeConfig(kw.get("target_commit_interval", 100000), kw.get("delta", 20), kw.get("gamma", 2.0), kw.get("lambda_", 0.5))
        part = (kw["partition_size"], kw["partition_start"], kw["partition_end"]) if "partition_size" in kw else None
        sim = amd.BatchSimulator.new(seeds, n, amd.RandomDelay.new(kw["mean"], kw["variance"]), nc, commands_per_epoch=kw.get("commands_per_epoch", 30000),
                                     equivocate_every=kw.get("equivocate_every", 0), drop_per_million=kw.get("drop_per_million", 0), partition=part,
                                     quirks=kw.get("quirks", 0), calendar_queue=bool(rng.random() < 0.3), max_steps_per_launch=int(rng.choice([0, 0, 173])),
                                     lanes_per_wavefront=int(rng.choice([1, 2, 4, 8, 16, 32, 64])), block_capacity=max_clock + 64)  # (1..32: lbft_k_run0q since round 5)
        res = sim.loop_until(max_clock, allow_faults=True)
        on_quad += bool(sim.layout()["kernel_class"] & (16384 | 32768))  # lbft_k_run0q, or lbft_k_run0u where one network per wavefront of a batch this small runs there
        if res.faults.any() and not (res.faults & ~np.uint32(1)).any():
            # LBFT_FAULT_QUEUE_OVERFLOW only: the scanned queue of these kernels holds 256 events (a larger queue_capacity = the heap / calendar of
            # the general kernel, which the other device fuzz covers); slow pacemakers under long delays can exceed it
            overflowed += 1
            continue
        assert not res.faults.any(), (kw, sorted(set(int(f) for f in res.faults)), res.counters, sim.layout())
        assert (res.commit_counts == ref["commit_counts"]).all(), kw
        assert (res.active_rounds == ref["active_rounds"]).all(), kw
        assert (res.last_committed_states == ref["last_states"]).all(), kw
        assert (res.committed_histories(96) == ref["histories"]).all(), kw
        c, rc = res.counters, ref["counters"]
        for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
            assert c[key] == rc[key], (key, kw)
    # (a guard that the draws reach the headline kernel at all, not a parity statement: 8 of 10 draws are eligible by construction, the packed queue's time
    # bits and the slot count take some away; of 125 chunks 120 had >= 5 on it, five had 3 or 4 -- every one of their configurations equal to the oracle)
    assert on_quad >= 3 and overflowed <= 2, (on_quad, overflowed)


# the same for networks of 33..128 nodes: the cooperative large-network kernel (lanes per wavefront 1..32, multi-launch)
@pytest.mark.gpu
@pytest.mark.parametrize("chunk", range(int(os.environ.get("LBFT_FUZZ_GPU_LARGE_FIRST", "0")), int(os.environ.get("LBFT_FUZZ_GPU_LARGE_FIRST", "0")) + int(os.environ.get("LBFT_FUZZ_GPU_LARGE_CHUNKS", "4"))))
def test_random_large_configurations_on_the_device_match_the_oracle(oracle, chunk):
    import librabft_simulator_amd as amd
    rng = np.random.default_rng(4242 + chunk)
    for _ in range(4):
        kw, n, max_clock = draw_large_config(rng)
        m = int(rng.choice([1, 3, 20]))
        seeds = rng.integers(1, 2 ** 62, m, dtype=np.uint64)
        ref = oracle.run_batch(oracle.make_config(math_mode=1, **kw), seeds, max_clock, threads=8, history_cap=64)
        delay = amd.RandomDelay.uniform(kw["uniform_lo"], kw["uniform_hi"]) if kw.get("delay_model") == 1 else \
            amd.RandomDelay.new(kw.get("mean", 10.0), kw.get("variance", 4.0))
        nc = amd.NodeConfig(kw.get("target_commit_interval", 100000), kw.get("delta", 20), kw.get("gamma", 2.0), kw.get("lambda_", 0.5))
        sim = amd.BatchSimulator.new(seeds, n, delay, nc, commands_per_epoch=kw.get("commands_per_epoch", 30000),
                                     voting_rights=kw.get("voting_rights"), equivocate_every=kw.get("equivocate_every", 0),
                                     drop_per_million=kw.get("drop_per_million", 0), quirks=kw.get("quirks", 0),
                                     rights_rotation=kw.get("rights_rotation", 0),
                                     calendar_queue=bool(rng.random() < 0.8), max_steps_per_launch=int(rng.choice([0, 0, 997])),
                                     lanes_per_wavefront=int(rng.choice([0, 0, 1, 4, 32])), block_capacity=max_clock + 64,
                                     queue_capacity=max(8192, 32 * n * n),
                                     # (a per-instance state must stay below 2^24 rows: 128 nodes x 281-word snapshots allow ~55 000 slots; a 128-node network with equal delays
                                     # has 49 000 in flight: chunk 57 of a widened run faulted on 36 000 -- on the device and on the host build alike)
                                     snapshot_capacity=min(52000, 6 * n * n + 16 * n) if kw.get("quirks", 0) & 1 else 128 * n)
        res = sim.loop_until(max_clock, allow_faults=True)
        assert not res.faults.any(), (kw, sorted(set(int(f) for f in res.faults)), res.counters, sim.layout())
        assert (res.commit_counts == ref["commit_counts"]).all(), kw
        assert (res.active_rounds == ref["active_rounds"]).all(), kw
        assert (res.last_committed_states == ref["last_states"]).all(), kw
        assert (res.committed_histories(64) == ref["histories"]).all(), kw
        c, rc = res.counters, ref["counters"]
        for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
            assert c[key] == rc[key], (key, kw)
        sim.close()
