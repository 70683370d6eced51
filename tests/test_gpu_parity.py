"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes -> liblbft_hip.so), against
the CPU oracle on the same seeds, against the reference's golden vectors, and -- at BASELINE.json's
full sizes -- through size-independent properties.  Bit-exact: everything here is integer work."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# Host threads of the oracle spot checks.  Measured on the GPU box (round 6, tests/tools/oracle_scaling.py, 512 instances of c4live): 32 threads 20.2 s,
# 64: 24.0 s, 128: 35.3 s, 256 (= os.cpu_count() there): 51.2 s -- the box gives this container the throughput of ~32 cores, and every thread beyond
# them costs: the suite's oracle time fell by 2.5 x with the cap.
HOST_THREADS = min(__import__("os").cpu_count() or 8, 32)


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import librabft_simulator_amd as L
    L.lib()  # fails loudly if liblbft_hip.so is missing
    return L


def run_gpu(amd, kw, seeds, max_clock, **sim_kw):
    kw = dict(kw)
    n = kw.pop("num_nodes")
    if kw.get("delay_model", 0) == 1:
        delay = amd.RandomDelay.uniform(kw.get("uniform_lo", 5), kw.get("uniform_hi", 15))
    else:
        delay = amd.RandomDelay.new(kw.get("mean", 10.0), kw.get("variance", 4.0))
    nc = amd.NodeConfig(kw.get("target_commit_interval", 100000), kw.get("delta", 20), kw.get("gamma", 2.0),
                        kw.get("lambda_", 0.5))
    sim = amd.BatchSimulator.new(seeds, n, delay, nc, commands_per_epoch=kw.get("commands_per_epoch", 30000),
                                 voting_rights=kw.get("voting_rights"), equivocate_every=kw.get("equivocate_every", 0),
                                 drop_per_million=kw.get("drop_per_million", 0), quirks=kw.get("quirks", 0),
                                 rights_rotation=kw.get("rights_rotation", 0),
                                 partition=(kw["partition_size"], kw["partition_start"], kw["partition_end"]) if "partition_size" in kw else None,
                                 **sim_kw)
    return sim, sim.loop_until(max_clock)


def assert_equal_to_oracle(oracle, res, kw, seeds, max_clock, cap=256):
    ref = oracle.run_batch(oracle.make_config(math_mode=1, **kw), seeds, max_clock, threads=8, history_cap=cap)
    assert not res.faults.any()
    assert (res.commit_counts == ref["commit_counts"]).all()
    assert (res.active_rounds == ref["active_rounds"]).all()
    assert (res.last_committed_states == ref["last_states"]).all()
    assert (res.committed_histories(cap) == ref["histories"]).all()
    c, rc = res.counters, ref["counters"]
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert c[key] == rc[key], key
    return ref


# Which class-0 run kernel serves which batch (round 5's measured crossover, lbft_hip.hip quad_eligible / small_batch_kernel / uni_kernel), named through
# lbft_batch_layout's flag word -- bit 14 lbft_k_run0q, bit 13 lbft_k_run0s, bits 13 + 15 lbft_k_run0u -- and every such batch bit for bit against the oracle.
@pytest.mark.parametrize("m,uniform,kernel,lanes", [
    (512, False, "lbft_k_run0u", 1),     # below 1 024 networks: the scalar-unit kernel whatever the network (256 x 4: 4.77 ms against 7.24 on lbft_k_run0q)
    (1024, False, "lbft_k_run0q", 1),    # BASELINE config 2's size with log-normal delays: the headline kernel, one network per wavefront
    (1024, True, "lbft_k_run0u", 1),     # BASELINE config 2 (uniform delays): not the compile-time network -> the scalar-unit kernel (4.71 against 5.11 ms)
    (8192, False, "lbft_k_run0q", 4),    # one GPU's share of config 3 on an 8-GPU node (8.2 against 9.4 ms)
    (4096, True, "lbft_k_run0s", 2),     # uniform delays beyond one network per wavefront: the wavefront-wide pop (7.2 against 8.4 ms on lbft_k_run0)
])
def test_small_batches_run_on_the_measured_kernel_and_equal_the_oracle(amd, oracle, m, uniform, kernel, lanes):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    kw = dict(num_nodes=4, delay_model=1, uniform_lo=5, uniform_hi=15) if uniform else dict(num_nodes=4)
    seeds = np.arange(1, m + 1, dtype=np.uint64) * 7919
    sim, res = run_gpu(amd, kw, seeds, 1000)
    lay = sim.layout()
    assert bench.run_kernel_name(lay["kernel_class"]) == kernel and lay["lanes_per_wavefront"] == lanes, lay
    idx = np.unique(np.linspace(0, m - 1, 768).astype(np.int64))
    ref = oracle.run_batch(oracle.make_config(math_mode=1, **kw), seeds[idx], 1000, threads=HOST_THREADS, history_cap=64)
    assert not res.faults.any()
    assert (res.commit_counts[idx] == ref["commit_counts"]).all() and (res.active_rounds[idx] == ref["active_rounds"]).all()
    assert (res.last_committed_states[idx] == ref["last_states"]).all() and (res.committed_histories(64)[idx] == ref["histories"]).all()


def test_reference_golden_3_nodes(amd):
    # librabft-v2/tests/simulated_run.rs:45-66
    contexts = amd.Simulator.new(52, 3, amd.RandomDelay.new(10.0, 4.0)).loop_until(amd.GlobalTime(1000))
    assert [len(c.committed_history()) for c in contexts] == [27, 27, 27]
    assert [int(c.last_committed_state()) for c in contexts] == [11134312813757838303] * 3
    h = contexts[0].committed_history()
    assert [(c.proposer, c.index, t) for c, t in h[:3]] == [(2, 0, 32), (2, 1, 52), (2, 2, 78)]


def test_reference_golden_8_nodes(amd):
    # librabft-v2/tests/simulated_run.rs:68-94
    contexts = amd.Simulator.new(48, 8, amd.RandomDelay.new(10.0, 4.0)).loop_until(amd.GlobalTime(1000))
    assert [len(c.committed_history()) for c in contexts] == [28] * 7 + [30]
    assert [int(c.last_committed_state()) for c in contexts] == [12785928431398617538] * 7 + [4890275890002623733]


CASES = {
    "c1_3nodes_fixed10_100rounds": (dict(num_nodes=3, mean=10.0, variance=0.0), 64, 2800),
    "c2_1024x4_lognormal": (dict(num_nodes=4), 1024, 1000),
    "c2_1024x4_uniform": (dict(num_nodes=4, delay_model=1, uniform_lo=5, uniform_hi=15), 1024, 1000),
    "n8": (dict(num_nodes=8), 128, 1000),
    "n1": (dict(num_nodes=1), 5, 300),
    "n2_ragged_batch": (dict(num_nodes=2), 67, 1000),
    "n16": (dict(num_nodes=16), 16, 400),
    # > 16 nodes: receiver lists in HBM rows; > 32: multi-word node/author sets; heap event queue (configs 4 / 5 shapes)
    "n20": (dict(num_nodes=20), 8, 300),
    "n33": (dict(num_nodes=33), 4, 300),
    "n64_long_tail": (dict(num_nodes=64, mean=10.0, variance=400.0), 4, 300),
    "n100_weighted": (dict(num_nodes=100, voting_rights=[1 + (i % 4) for i in range(100)]), 2, 200),
    "n36_timeouts": (dict(num_nodes=36, mean=10.0, variance=900.0, delta=5), 4, 400),
    "n8_timeouts": (dict(num_nodes=8, mean=10.0, variance=400.0), 64, 1500),
    # equivocating leaders (extension; oracle/lbft_oracle.h "Equivocators" is the specification) -- config 4's shape
    "equiv_n4": (dict(num_nodes=4, equivocate_every=4), 256, 1000),
    "equiv_n7_every_third": (dict(num_nodes=7, equivocate_every=3), 64, 1000),
    "equiv_n64_long_tail_every_fifth": (dict(num_nodes=64, mean=10.0, variance=400.0, equivocate_every=5), 4, 300),
    # lossy network (extension; oracle/lbft_oracle.h "Lossy network"): random loss, partition
    "lossy_drop_5_percent": (dict(num_nodes=4, drop_per_million=50000), 256, 1500),
    "lossy_partition_2_2": (dict(num_nodes=4, partition_size=2, partition_start=200, partition_end=700), 128, 1500),
    "lossy_drop_partition_n40": (dict(num_nodes=40, drop_per_million=20000, partition_size=13, partition_start=50, partition_end=150), 4, 300),
    "lossy_drop_equivocators_long_tail": (dict(num_nodes=7, drop_per_million=100000, equivocate_every=4, mean=10.0, variance=400.0), 64, 1500),
    # quirks bit 1: EpochId::previous() = id - 1 (reference quirk Q2 fixed): epoch changes do not stall the network
    "q2fixed_n4_cpe50": (dict(num_nodes=4, commands_per_epoch=50, quirks=2), 128, 3000),
    "q2fixed_n7_weighted_cpe9": (dict(num_nodes=7, commands_per_epoch=9, quirks=2, voting_rights=[2, 1, 1, 3, 1, 2, 1]), 64, 2000),
    "q2fixed_n36_cpe3": (dict(num_nodes=36, commands_per_epoch=3, quirks=2), 4, 300),
    # quirks bit 0: requests answered by the peer with real payloads (reference quirk Q1 fixed); 3 = Q1 and Q2 fixed
    "q1fixed_n4_long_tail": (dict(num_nodes=4, quirks=1, mean=10.0, variance=400.0), 128, 2500),
    "q3fixed_n4_cpe50": (dict(num_nodes=4, quirks=3, commands_per_epoch=50), 128, 3000),
    "q3fixed_n7_weighted_cpe9": (dict(num_nodes=7, quirks=3, commands_per_epoch=9, voting_rights=[2, 1, 1, 3, 1, 2, 1]), 64, 2000),
    "q3fixed_n5_partition_heals": (dict(num_nodes=5, quirks=3, partition_size=2, partition_start=100, partition_end=400, commands_per_epoch=20), 64, 2500),
    "q3fixed_n36_cpe3": (dict(num_nodes=36, quirks=3, commands_per_epoch=3), 4, 300),
    "q1fixed_n7_equivocators_lossy": (dict(num_nodes=7, quirks=1, equivocate_every=3, drop_per_million=100000), 64, 1500),
    # epoch reconfiguration (extension): the voting rights of epoch e are voting_rights[(i + e * rights_rotation) % n]
    "rot1_n4_q3_cpe5": (dict(num_nodes=4, quirks=3, commands_per_epoch=5, voting_rights=[1, 2, 3, 4], rights_rotation=1), 128, 2500),
    "rot3_n7_q3_cpe9": (dict(num_nodes=7, quirks=3, commands_per_epoch=9, voting_rights=[2, 1, 1, 3, 1, 2, 1], rights_rotation=3), 64, 2000),
    "rot1_n4_q2_cpe7_class0": (dict(num_nodes=4, quirks=2, commands_per_epoch=7, voting_rights=[5, 1, 1, 1], rights_rotation=1), 128, 2000),
    "rot5_n36_q3_cpe3": (dict(num_nodes=36, quirks=3, commands_per_epoch=3, voting_rights=[1 + (i % 3) for i in range(36)], rights_rotation=5), 4, 300),
    "rot7_n100_q2_cpe2": (dict(num_nodes=100, quirks=2, commands_per_epoch=2, voting_rights=[1 + (i % 4) for i in range(100)], rights_rotation=7), 2, 200),
    "rot7_n100_q3_cpe2": (dict(num_nodes=100, quirks=3, commands_per_epoch=2, voting_rights=[1 + (i % 4) for i in range(100)], rights_rotation=7), 2, 260),
    "equiv_n5_weighted_epochs": (dict(num_nodes=5, equivocate_every=2, voting_rights=[1, 3, 1, 2, 2], commands_per_epoch=7), 64, 1500),
    "epoch_change_cpe50": (dict(num_nodes=4, commands_per_epoch=50), 128, 3000),
    "weighted": (dict(num_nodes=5, voting_rights=[5, 1, 1, 2, 3]), 128, 1000),
    "long_tail": (dict(num_nodes=4, mean=10.0, variance=400.0), 256, 2000),
    "params": (dict(num_nodes=5, gamma=1.5, lambda_=0.25, delta=5, target_commit_interval=80), 128, 1500),
    "long_run": (dict(num_nodes=4), 32, 8000),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_equals_oracle(amd, oracle, name):
    kw, m, max_clock = CASES[name]
    seeds = np.arange(1, m + 1, dtype=np.uint64) * 104729 + 17
    _, res = run_gpu(amd, kw, seeds, max_clock)
    assert_equal_to_oracle(oracle, res, kw, seeds, max_clock, cap=512 if max_clock > 3000 else 256)


@pytest.mark.parametrize("name", ["n8", "n64_long_tail", "equiv_n7_every_third", "lossy_drop_partition_n40"])
def test_heap_queue_equals_calendar_queue(amd, oracle, name):
    """Outside kernel class 0, networks whose queue capacity exceeds 256 events use a calendar when max_clock <= 2047 (default in
    CASES above) and a binary heap otherwise (smaller ones keep the LDS-fronted array); both must give the oracle's results."""
    kw, m, max_clock = CASES[name]
    seeds = np.arange(1, m + 1, dtype=np.uint64) * 104729 + 17
    sim, res = run_gpu(amd, kw, seeds, max_clock, calendar_queue=False)
    assert (sim.layout()["kernel_class"] >> 9) & 1 == 0
    assert_equal_to_oracle(oracle, res, kw, seeds, max_clock)
    sim2, _ = run_gpu(amd, kw, seeds, max_clock)
    assert (sim2.layout()["kernel_class"] >> 9) & 1 == 1


def test_long_horizon_64_nodes_calendar_equals_heap_equals_oracle(amd, oracle):
    """Horizons beyond the round-1 calendar (max_clock <= 2047): 64-node networks to clock 3000 on the calendar queue (now up to
    16383) and on the binary heap (what longer horizons and the round trace fall back to) -- both bit-exact against the oracle.
    Reference ordering: bft-lib/src/simulator.rs:141-169."""
    kw, seeds, max_clock = dict(num_nodes=64), np.arange(1, 5, dtype=np.uint64) * 7919, 3000
    sim, res = run_gpu(amd, kw, seeds, max_clock)
    assert (sim.layout()["kernel_class"] >> 9) & 1 == 1                 # calendar
    ref = assert_equal_to_oracle(oracle, res, kw, seeds, max_clock, cap=512)
    assert ref["commit_counts"].max() >= 80
    sim2, res2 = run_gpu(amd, kw, seeds, max_clock, calendar_queue=False)
    assert (sim2.layout()["kernel_class"] >> 9) & 1 == 0 and (sim2.layout()["kernel_class"] >> 8) & 1 == 1   # heap
    assert_equal_to_oracle(oracle, res2, kw, seeds, max_clock, cap=512)
    assert res2.counters["max_queue"] == res.counters["max_queue"]


def test_bench_two_ranks_on_one_device():
    """The N > 1 path of bench.py on the HIP path (until an 8-GPU node runs it): two ranks (gloo rendezvous, both on HIP
    device 0), instances sharded by distributed.shard_seeds, ONE all-gather of the counters.  The aggregate must be the
    whole-job figure: twice the per-rank instance count, rounds of both shards."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--single-device", "--instances", "4096",
           "--no-cpu-baseline"]
    # first WITHOUT an external launcher (`python bench.py --gpus 2 ...`: bench.py starts its own two ranks), then under the driver's launcher
    env_bare = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bare = subprocess.run([sys.executable] + cmd[cmd.index(os.path.join(root, "bench.py")):], env=env_bare, capture_output=True, text=True, timeout=600, cwd=root)
    assert bare.returncode == 0, bare.stderr[-2000:]
    db = json.loads([l for l in bare.stdout.splitlines() if l.startswith("{")][-1])
    assert db["n_gpus"] == 2 and db["scaling"] == "weak" and db["faulted_instances"] == 0 and db["parity"]["mismatches"] == 0
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["faulted_instances"] == 0
    assert round(d["value"] * d["ms_per_step"]) == round(db["value"] * db["ms_per_step"])
    # ... and the SAME line carries BASELINE config 3 as it is named: one batch of --instances instances split over the ranks (strong scaling), timed in a
    # second leg; its aggregate work is that of a single 4 096-instance batch
    for line_ in (d, db):
        s3 = line_["config3_strong"]
        assert s3["scaling"] == "strong" and s3["total_instances"] == 4096 and s3["instances_per_gpu"] == 2048 and s3["n_gpus"] == 2 and s3["faulted_instances"] == 0
    one4 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--instances", "4096", "--no-cpu-baseline"],
                          env=env_bare, capture_output=True, text=True, timeout=600, cwd=root)
    assert one4.returncode == 0, one4.stderr[-2000:]
    w4 = json.loads([l for l in one4.stdout.splitlines() if l.startswith("{")][-1])
    assert round(d["config3_strong"]["value"] * d["config3_strong"]["ms_per_step"]) == round(w4["value"] * w4["ms_per_step"])
    assert w4["config3_strong"]["value"] == w4["value"] and w4["config3_strong"]["n_gpus"] == 1  # one GPU: the strong split is the weak batch
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--instances", "8192", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    w = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    # same 8 192 seeds (base_seed + global index), sharded 2 x 4 096: identical aggregate work
    assert round(d["value"] * d["ms_per_step"]) == round(w["value"] * w["ms_per_step"])
    assert round(d["events_per_s"] * d["ms_per_step"]) == round(w["events_per_s"] * w["ms_per_step"])
    # strong scaling (--total-instances: ONE 8 192-instance batch sharded over the two ranks): the same aggregate again
    cmd_s = cmd[:cmd.index("--instances")] + ["--total-instances", "8192", "--no-cpu-baseline"]
    out_s = subprocess.run(cmd_s, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out_s.returncode == 0, out_s.stderr[-2000:]
    ds = json.loads([l for l in out_s.stdout.splitlines() if l.startswith("{")][-1])
    assert ds["n_gpus"] == 2 and ds["scaling"] == "strong" and ds["config"]["total_instances"] == 8192 and ds["config"]["instances_per_gpu"] == 4096
    assert round(ds["value"] * ds["ms_per_step"]) == round(w["value"] * w["ms_per_step"]) and ds["parity"]["mismatches"] == 0


def test_bench_two_ranks_nccl():
    """bench.py --gpus 2 over RCCL (backend "nccl", one device per rank): what the driver's scaling run launches.  Needs two GPUs; the
    one-GPU boxes of the test tier skip it (the native RCCL path is still executed there: tests/test_node_level.py::
    test_counters_allgather_reduce_through_rccl, and test_two_rccl_ranks_on_one_device_reach_the_duplicate_device_check below)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (found %d)" % torch.cuda.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    for extra, scaling in ((["--instances", "4096"], "weak"), (["--total-instances", "8192"], "strong")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--native-collective"] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["faulted_instances"] == 0 and d["parity"]["mismatches"] == 0
        assert d["config"]["total_instances"] == 8192
        # the C ABI's own collective on two ranks (ncclCommInitRank + lbft_batch_counters_allgather_reduce) == the torch.distributed aggregate
        assert d["native_collective"]["ranks"] == 2 and d["native_collective"]["matches_torch_aggregate"]


def test_two_rccl_ranks_on_one_device_reach_the_duplicate_device_check(tmp_path):
    """2-GPU readiness on a 1-GPU box (round-4 review item 8): two processes, both on HIP device 0, build a TWO-rank RCCL communicator for the native
    collective -- librccl loaded by hand, the ncclUniqueId handed over, the bootstrap over 127.0.0.1 connected -- and RCCL then refuses it, because two
    ranks of one communicator may not share a device (ncclInvalidUsage = 5, "Duplicate GPU detected"): that refusal is the expected result here and the
    furthest the native 2-rank path can go without a second GPU.  A RCCL build that accepts the communicator runs the collective
    as well (tests/tools/rccl_two_ranks_one_device.py) and must aggregate correctly."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "tools", "rccl_two_ranks_one_device.py")
    uid = str(tmp_path / "nccl_uid.bin")
    import ctypes
    try:  # (round-5 advisor: a box whose loader path has no librccl skips instead of failing; the sibling test in test_node_level.py guards the same load)
        ctypes.CDLL("librccl.so.1")
    except OSError as e:
        pytest.skip("librccl.so.1 is not on the loader path: %s" % e)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"))
    procs = [subprocess.Popen([sys.executable, script, str(r), uid], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = []
    try:
        # a rank that dies early leaves its peer waiting in ncclCommInitRank: poll both and stop the peer as soon as one exits non-zero
        import time as _time
        deadline = _time.time() + 240
        while any(p.poll() is None for p in procs) and _time.time() < deadline:
            if any(p.poll() not in (None, 0) for p in procs):
                break
            _time.sleep(0.2)
        for p in procs:
            if p.poll() is None and any(q.poll() not in (None, 0) for q in procs):
                p.kill()
        for p in procs:
            o, e = p.communicate(timeout=30)
            assert p.returncode == 0, e[-2000:]
            outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert sorted(d["rank"] for d in outs) == [0, 1]
    rcs = {d["init_rc"] for d in outs}
    assert rcs in ({5}, {0}), outs  # ncclInvalidUsage on both ranks (one device), or a communicator on both
    if rcs == {0}:
        assert all(d["collective"] == "ok" for d in outs) and outs[0]["agg_rounds"] == outs[1]["agg_rounds"] == outs[0]["rounds"] + outs[1]["rounds"], outs


def test_multi_launch_equals_single_launch(amd, oracle):
    kw, seeds = dict(num_nodes=4), np.arange(1, 257, dtype=np.uint64)
    _, res = run_gpu(amd, kw, seeds, 1000, max_steps_per_launch=97)
    assert res.counters["launches"] > 10
    assert_equal_to_oracle(oracle, res, kw, seeds, 1000)


@pytest.mark.parametrize("name,steps", [("rot3_n7_q3_cpe9", 5), ("rot5_n36_q3_cpe3", 211), ("rot7_n100_q3_cpe2", 1999)])
def test_multi_launch_with_responses_spanning_epochs(amd, oracle, name, steps):
    """Quirks bit 0 + epoch changes: a response whose records span several epochs is several steps of the event loop (its
    continuation lives in the instance's scalar rows), so launches bounded to a few steps end between them."""
    kw, m, max_clock = CASES[name]
    seeds = np.arange(3, 3 + m, dtype=np.uint64) * 7919
    _, res = run_gpu(amd, kw, seeds, max_clock, max_steps_per_launch=steps)
    assert res.counters["launches"] > 10
    assert_equal_to_oracle(oracle, res, kw, seeds, max_clock)


def test_reset_reruns_identically(amd):
    seeds = np.arange(1, 129, dtype=np.uint64)
    sim, r1 = run_gpu(amd, dict(num_nodes=4), seeds, 1000)
    s1, h1 = r1.last_committed_states.copy(), r1.committed_histories(64)
    sim.reset()
    r2 = sim.loop_until(1000)
    assert (r2.last_committed_states == s1).all() and (r2.committed_histories(64) == h1).all()


def test_capacity_overflow_sets_fault_flag(amd):
    seeds = np.arange(1, 65, dtype=np.uint64)
    sim = amd.BatchSimulator.new(seeds, 4, amd.RandomDelay.new(10.0, 4.0), queue_capacity=16, block_capacity=8)
    with pytest.raises(amd.LbftError) as e:
        sim.loop_until(1000)
    assert e.value.code == -5
    out = np.zeros(64, dtype=np.uint32)  # the handle stays valid; per-instance fault words are readable
    from librabft_simulator_amd import _lib
    _lib.check(_lib.lib().lbft_batch_faults(sim._h, out.ctypes.data))
    assert (out != 0).all()


def test_zero_max_clock_and_empty_histories(amd, oracle):
    seeds = np.arange(1, 9, dtype=np.uint64)
    _, res = run_gpu(amd, dict(num_nodes=4), seeds, 0)
    assert res.commit_counts.sum() == 0
    assert (res.last_committed_states == 13646096770106105413).all()  # State of the empty ledger (README.md:27)
    assert sum(res.counters["events"]) == 0


def test_device_third_party_arithmetic(amd, oracle):
    from librabft_simulator_amd import _lib
    L, OL = _lib.lib(), oracle.lib()
    # leaders (pacemaker.rs:100-109 -> configuration.rs:65-75), unit and weighted rights
    for n, w in ((3, None), (8, None), (5, [5, 1, 1, 2, 3])):
        out = np.zeros(600, dtype=np.uint8)
        wa = np.array(w, dtype=np.uint64) if w else None
        _lib.check(L.lbft_device_leaders(0, wa.ctypes.data if w else None, n, out.ctypes.data, 600))
        assert [int(v) for v in out] == [oracle.leader(n, r, w) for r in range(600)]
    assert [int(v) for v in out[:0]] == []
    # delay streams: ziggurat + exp, log-normal and uniform
    for kw in (dict(variance=4.0), dict(variance=400.0), dict(variance=0.0), dict(delay_model=1, uniform_lo=5, uniform_hi=15)):
        n = 100000
        a, b = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        OL.lbft_oracle_sample_delays(ctypes.byref(oracle.make_config(math_mode=1, **kw)), 99, a.ctypes.data, n)
        if kw.get("delay_model") == 1:
            delay = amd.RandomDelay.uniform(5, 15)
        else:
            delay = amd.RandomDelay.new(10.0, kw["variance"])
        from librabft_simulator_amd.simulator import make_config
        cfg = make_config(3, delay, amd.NodeConfig())
        _lib.check(L.lbft_device_sample_delays(0, ctypes.byref(cfg), 99, b.ctypes.data, n))
        assert (a == b).all(), kw
    # exp/log bit patterns
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-12, 12, 100000), rng.uniform(1e-12, 1.0, 100000)])
    e, l = np.zeros_like(x), np.zeros_like(x)
    _lib.check(L.lbft_device_exp_log(0, x.ctypes.data, e.ctypes.data, l.ctypes.data, len(x)))
    ee = np.array([OL.lbft_oracle_exp_strict(float(v)) for v in x])
    ll = np.array([OL.lbft_oracle_log_strict(float(v)) for v in x[100000:]])
    assert (e.view(np.uint64) == ee.view(np.uint64)).all()
    assert (l[100000:].view(np.uint64) == ll.view(np.uint64)).all()


def siphash13_words(words):
    """SipHash-1-3 (keys 0,0) over little-endian u64 words (numpy-free, small inputs only)."""
    mask = (1 << 64) - 1
    v = [0x736f6d6570736575, 0x646f72616e646f6d, 0x6c7967656e657261, 0x7465646279746573]

    def rotl(x, b):
        return ((x << b) | (x >> (64 - b))) & mask

    def rnd():
        v[0] = (v[0] + v[1]) & mask; v[1] = rotl(v[1], 13); v[1] ^= v[0]; v[0] = rotl(v[0], 32)
        v[2] = (v[2] + v[3]) & mask; v[3] = rotl(v[3], 16); v[3] ^= v[2]
        v[0] = (v[0] + v[3]) & mask; v[3] = rotl(v[3], 21); v[3] ^= v[0]
        v[2] = (v[2] + v[1]) & mask; v[1] = rotl(v[1], 17); v[1] ^= v[2]; v[2] = rotl(v[2], 32)

    for m in words:
        m &= mask
        v[3] ^= m; rnd(); v[0] ^= m
    b = ((len(words) * 8) << 56) & mask
    v[3] ^= b; rnd(); v[0] ^= b
    v[2] ^= 0xff
    rnd(); rnd(); rnd()
    return v[0] ^ v[1] ^ v[2] ^ v[3]


def test_full_size_65536x4_properties(amd, oracle):
    """BASELINE.json configs[2] on one GPU: size-independent properties + oracle spot checks."""
    m = 65536
    seeds = np.arange(1, m + 1, dtype=np.uint64)  # seed_i = base_seed + i, base_seed = 1
    _, res = run_gpu(amd, dict(num_nodes=4), seeds, 1000)
    assert not res.faults.any()
    cc = res.commit_counts
    cap = int(cc.max())
    hist = res.committed_histories(cap)
    # (1) safety: within an instance every node's log is a prefix of the longest log
    longest = cc.argmax(axis=1)
    ref_log = hist[np.arange(m), longest]
    k = np.arange(cap)[None, None, :]
    valid = k < cc[:, :, None]
    same = (hist == ref_log[:, None, :]) | ~valid
    assert same.all()
    # (2) per-proposer command indices are strictly increasing along a log; times are non-decreasing per proposer
    # (3) every committed entry's proposer is a node id
    assert (hist["proposer"][valid] < 4).all()
    # (4) the device State hash equals SipHash-1-3 of the exported history (checksum of checksums)
    states = res.last_committed_states
    rng = np.random.default_rng(0)
    for i in rng.integers(0, m, 200):
        for n in range(4):
            words = [int(cc[i, n])]
            for e in hist[i, n, :cc[i, n]]:
                words += [int(e["proposer"]), int(e["index"]), int(e["time"]) & ((1 << 64) - 1)]
            assert siphash13_words(words) == int(states[i, n])
    # (5) liveness of the healthy configuration: almost every instance commits
    assert (cc.min(axis=1) >= 20).mean() > 0.95
    # (6) every instance against the committed oracle digests + a live oracle sample with histories, bit-exact
    covered, _ = _fixture_check("c3_65536x4", amd, oracle, res, dict(num_nodes=4), seeds, 1000, live=512, hist=hist)
    assert covered == m


def test_round_switch_csv_equals_reference_data_writer(amd, oracle, tmp_path):
    """`--create_csv` (bft-lib/src/data_writer.rs): device trace == the oracle's DataWriter, and the files have the
    reference's layout (header `node i`, empty cells, one message count)."""
    # (the last two: rounds far shorter than the trace capacity loop_until starts from -- the run is repeated with the worst-case capacity)
    for n, max_clock, kw in ((3, 1000, {}), (4, 2000, {}), (8, 600, {}), (5, 1500, dict(mean=10.0, variance=400.0)), (2, 700, dict(mean=1.0, variance=0.0)),
                             (1, 500, dict(delta=4, gamma=1.0))):
        seeds = np.arange(50, 58, dtype=np.uint64)
        sim = amd.BatchSimulator.new(seeds, n, amd.RandomDelay.new(kw.get("mean", 10.0), kw.get("variance", 4.0)),
                                     amd.NodeConfig(100000, kw.get("delta", 20), kw.get("gamma", 2.0), 0.5))
        res = sim.loop_until(max_clock, csv_path=str(tmp_path / ("csv%d" % n)))
        cfg = oracle.make_config(num_nodes=n, math_mode=1, **kw)
        if n <= 2:
            assert res.active_rounds.max() > max_clock // 5 + 64  # (the first capacity did overflow)
        for i, seed in enumerate(seeds):
            o = oracle.OracleSim(cfg, int(seed)).enable_data_writer()
            o.run_until(max_clock)
            assert res.round_switches(i) == o.round_switches()
        lines = (tmp_path / ("csv%d" % n) / "round_switches.txt").read_text().splitlines()
        rows, messages = res.round_switches(0)
        assert lines[0] == ",".join("node %d" % k for k in range(n)) and len(lines) == 1 + len(rows)
        assert lines[1] == "," * (n - 1)  # round 0 is never entered
        assert (tmp_path / ("csv%d" % n) / "number_of_messages.txt").read_text().strip() == str(messages)
        # the trace does not change results
        ref = oracle.run_batch(cfg, seeds, max_clock, threads=4)
        assert (res.commit_counts == ref["commit_counts"]).all() and (res.last_committed_states == ref["last_states"]).all()


def test_checkpoint_resume_equals_uninterrupted_run(amd, oracle, tmp_path):
    """save_node / load_node at batch granularity: step, checkpoint, restore into a NEW batch, continue == one run."""
    kw, seeds, max_clock = dict(num_nodes=4), np.arange(1, 193, dtype=np.uint64), 1000
    sim = amd.BatchSimulator.new(seeds, 4, amd.RandomDelay.new(10.0, 4.0))
    left, res = sim.run_steps(max_clock, 250)
    assert left == len(seeds) and res is None
    left, res = sim.run_steps(max_clock, 400)
    assert left > 0
    path = str(tmp_path / "batch.ckpt")
    assert sim.save_checkpoint(path) > 192 * 16000
    sim.close()
    sim2 = amd.BatchSimulator.new(np.zeros(len(seeds), dtype=np.uint64), 4, amd.RandomDelay.new(10.0, 4.0))  # seeds live in the state
    sim2.load_checkpoint(path)
    launches = 0
    while True:
        left, res = sim2.run_steps(max_clock, 300)
        launches += 1
        if left == 0:
            break
    assert launches >= 2
    assert_equal_to_oracle(oracle, res, kw, seeds, max_clock)
    # a checkpoint does not load into a differently configured batch: node count, protocol mode (quirks), fault model (loss,
    # partition, equivocators), rotating rights, delay model -- everything that changes the meaning or the encoding of the state
    # words (a class-0 checkpoint holds packed one-word queue entries; a lossy / quirks batch runs another kernel class)
    others = [dict(num_nodes=5), dict(quirks=1), dict(quirks=2), dict(drop_per_million=1000), dict(partition=(2, 0, 100)), dict(equivocate_every=2),
              dict(voting_rights=[1, 2, 3, 4], rights_rotation=1), dict(commands_per_epoch=7), dict(variance=9.0)]
    for o in others:
        o = dict(o)
        n = o.pop("num_nodes", 4)
        delay = amd.RandomDelay.new(10.0, o.pop("variance", 4.0))
        sim3 = amd.BatchSimulator.new(seeds, n, delay, **o)
        with pytest.raises(amd.LbftError):
            sim3.load_checkpoint(path)
        res3 = sim3.loop_until(200)  # ... and the refused load left the batch usable with its own capacities
        assert not res3.faults.any()
        sim3.close()


def _prefix_consistent(cc, hist):
    """Safety: within an instance every node's committed log is a prefix of the longest one."""
    m = cc.shape[0]
    longest = hist[np.arange(m), cc.argmax(axis=1)]
    k = np.arange(hist.shape[2])[None, None, :]
    valid = k < cc[:, :, None]
    return bool(((hist == longest[:, None, :]) | ~valid).all())


def _fixture_check(name, amd, oracle, res, kw, seeds, max_clock, live=64, hist=None):
    """Full-size parity without oracle time in the suite (tests/full_size_digest.py): EVERY instance of the batch that the committed fixture
    tests/golden/full_size_digests.npz covers -- one digest per instance of (commit counts, active rounds, State hashes of all its nodes), computed
    offline from the CPU oracle by tests/golden/gen_full_size.py -- is compared with the device's results; a small LIVE oracle sample (`live`
    instances, strided over the covered ones) guards the fixture itself against oracle drift and also compares the histories entry by entry.
    Returns the number of instances compared through the fixture."""
    import full_size_digest as fsd
    table, _meta = fsd.load_fixture()
    assert name in table, "tests/golden/full_size_digests.npz has no entry for %s (tests/golden/gen_full_size.py)" % name
    covered, bad = fsd.compare(name, res.commit_counts, res.active_rounds, res.last_committed_states)
    assert covered >= 256, (name, covered)
    assert len(bad) == 0, "%s: %d of %d covered instances differ from the oracle's digests, first: %s" % (name, len(bad), covered, bad[:8])
    want, cov = table[name]
    pool = np.nonzero(cov)[0]
    idx = pool[np.unique(np.linspace(0, len(pool) - 1, min(live, len(pool))).astype(np.int64))]
    cap = hist.shape[2] if hist is not None else 0
    ref = oracle.run_batch(oracle.make_config(math_mode=1, **kw), seeds[idx], max_clock, threads=HOST_THREADS, history_cap=cap)
    assert (fsd.digests(ref["commit_counts"], ref["active_rounds"], ref["last_states"]) == want[idx]).all(), "the live oracle disagrees with the committed fixture: regenerate it"
    assert (res.commit_counts[idx] == ref["commit_counts"]).all() and (res.active_rounds[idx] == ref["active_rounds"]).all()
    assert (res.last_committed_states[idx] == ref["last_states"]).all()
    if hist is not None:
        assert (hist[idx] == ref["histories"]).all()
    return covered, ref


def test_full_size_config4_16384x64_equivocators_properties(amd, oracle):
    """BASELINE.json configs[3] as SURVEY.md 8(d) wrote it: 16 384 x 64 nodes (f = 21), long-tail delays, every fifth node
    equivocating, reference semantics, clock 300.  Degenerate -- nothing commits by then (and with reference quirk Q1 the
    network stalls after ~4 commits however long it runs: stragglers cannot catch up) -- kept as the second line beside
    test_full_size_config4_live_*.  Properties at full size + every instance against the oracle's committed digests (_fixture_check)."""
    m, n, max_clock = 16384, 64, 300
    kw = dict(num_nodes=n, mean=10.0, variance=400.0, equivocate_every=5)
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    sim, res = run_gpu(amd, kw, seeds, max_clock)
    assert sim.layout()["kernel_class"] & 0x14ff == 0x0402, sim.layout()  # lbft_k_run2l (lbft_batch_layout: class 2 | two-wavefront kernel << 10 | record exchange << 12)
    assert not res.faults.any()
    cc = res.commit_counts
    hist = res.committed_histories(max(int(cc.max()), 1))
    assert _prefix_consistent(cc, hist)                      # 13 < 64/3 equivocators cannot break safety
    ar = res.active_rounds
    assert (ar >= 1).all() and (ar.max(axis=1) - ar.min(axis=1) <= ar.max()).all()
    c = res.counters
    assert c["events"][1] == c["events"][2] or c["events"][1] >= c["events"][2]  # every processed response had a request
    # ALL 16 384 instances against the committed oracle digests (518 M events on the oracle, offline), 64 live
    covered, _ = _fixture_check("c4_16384x64_longtail_equivocators", amd, oracle, res, kw, seeds, max_clock, live=64, hist=hist)
    assert covered == m


def test_full_size_config5_8192x100_weighted_epochs_properties(amd, oracle):
    """BASELINE.json configs[4] as SURVEY.md 8(d) wrote it: 8 192 x 100 nodes, voting rights 1 + (i mod 4), an epoch every
    50 commands (reference semantics incl. quirks Q1/Q2), clock 300: never reaches an epoch change -- kept as the second
    line beside test_full_size_config5_live_*.  Properties at full size + every instance against the oracle's committed digests (_fixture_check)."""
    m, n, max_clock = 8192, 100, 300
    rights = [1 + (i % 4) for i in range(n)]
    kw = dict(num_nodes=n, voting_rights=rights, commands_per_epoch=50)
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    sim, res = run_gpu(amd, kw, seeds, max_clock)
    assert sim.layout()["kernel_class"] & 0x14ff == 0x0402, sim.layout()  # lbft_k_run2l (lbft_batch_layout: class 2 | two-wavefront kernel << 10 | record exchange << 12)
    assert not res.faults.any()
    cc = res.commit_counts
    hist = res.committed_histories(max(int(cc.max()), 1))
    assert _prefix_consistent(cc, hist)
    assert (hist["proposer"][np.arange(hist.shape[2])[None, None, :] < cc[:, :, None]] < n).all()
    assert (cc.min(axis=1) >= 1).mean() > 0.9               # the healthy weighted network commits
    assert (res.epochs == 0).all()                           # 50 commands are not reached by clock 300
    # every instance against the committed oracle digests (rounds 1-5 compared 512 per suite run: 1 024 cost 228 s of oracle time), 64 live
    covered, _ = _fixture_check("c5_8192x100_weighted_epochs", amd, oracle, res, kw, seeds, max_clock, live=64, hist=hist)
    assert covered == m


def test_full_size_config4_live_16384x64_equivocators_commit(amd, oracle):
    """Configuration 4 exercising what it is named for, at full size: 16 384 x 64 nodes, LogNormal(10, 400) delays, 13 of 64
    nodes (index % 5 == 0) equivocating, the fixed protocol mode (quirks = 3: lagging nodes catch up through real
    request / response payloads), clock 1000.  Every node of (almost) every instance commits >= 5 blocks, so the safety
    assertion runs over non-empty logs; 2 048 instances (the sample of profiles/r03/full_size_checks_2048.txt) are compared with the oracle bit for bit."""
    m, n, max_clock = 16384, 64, 1000
    kw = dict(num_nodes=n, mean=10.0, variance=400.0, equivocate_every=5, quirks=3)
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    sim, res = run_gpu(amd, kw, seeds, max_clock)
    assert sim.layout()["kernel_class"] & 0x14ff == 0x1402, sim.layout()  # lbft_k_run2q (lbft_batch_layout: class 2 | two-wavefront kernel << 10 | record exchange << 12)
    assert not res.faults.any()
    cc = res.commit_counts
    assert (cc.min(axis=1) >= 5).mean() >= 0.9, np.bincount(cc.min(axis=1))   # liveness despite f_byz = 13 < 64 / 3
    assert cc.min() > 0
    hist = res.committed_histories(int(cc.max()))
    assert _prefix_consistent(cc, hist)                                          # safety over non-empty logs
    valid = np.arange(hist.shape[2])[None, None, :] < cc[:, :, None]
    assert (hist["proposer"][valid] < n).all()
    # both blocks of an equivocating proposal can never be committed: a (proposer, index) pair appears once per log
    key = hist["proposer"].astype(np.int64) * (1 << 32) + hist["index"].astype(np.int64)
    first = np.sort(np.where(valid, key, -1 - np.arange(hist.shape[2])[None, None, :]), axis=2)
    assert (np.diff(first, axis=2) != 0).all()
    # ALL 16 384 instances against the committed oracle digests (rounds 3-5 compared 2 048 per suite run and never the whole batch), 128 live
    covered, ref = _fixture_check("c4live_16384x64_longtail_equivocators_fixed", amd, oracle, res, kw, seeds, max_clock, live=128, hist=hist)
    assert covered == m
    assert ref["counters"]["response_inserts"] > 0                               # stragglers did catch up through responses


def test_full_size_config5_live_8192x100_rotating_rights_epochs(amd, oracle):
    """Configuration 5 exercising what it is named for, at full size: 8 192 x 100 nodes, voting rights 1 + (i mod 4) rotating
    by one node per epoch (rights_rotation = 1), an epoch every 3 commands, the fixed protocol mode (quirks = 3), clock
    450: every node of every instance goes through >= 2 epoch changes (node.rs:331-348).  1 024 instances (2 048 with LBFT_FULL_CHECK_FRACTION=2) bit-exact against the oracle."""
    m, n, max_clock = 8192, 100, 450
    rights = [1 + (i % 4) for i in range(n)]
    kw = dict(num_nodes=n, voting_rights=rights, commands_per_epoch=3, quirks=3, rights_rotation=1)
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    sim, res = run_gpu(amd, kw, seeds, max_clock)
    assert sim.layout()["kernel_class"] & 0x14ff == 0x1402, sim.layout()  # lbft_k_run2q (lbft_batch_layout: class 2 | two-wavefront kernel << 10 | record exchange << 12)
    assert not res.faults.any()
    cc, ep = res.commit_counts, res.epochs
    assert (ep >= 2).all()                                   # >= 2 reconfigurations at every node of every instance
    assert (ep == cc // 3).all()                             # read_epoch_id = commands / commands_per_epoch (simulated_context.rs:199-207)
    hist = res.committed_histories(int(cc.max()))
    assert _prefix_consistent(cc, hist)                      # logs agree across the epochs
    covered, _ = _fixture_check("c5live_8192x100_rotating_rights_epochs_fixed", amd, oracle, res, kw, seeds, max_clock, live=64, hist=hist)
    assert covered == m


def test_full_size_config5_as_named_8192x100_reconfiguration_every_50_commits(amd, oracle):
    """BASELINE.json configs[4] AS IT IS NAMED (round-5 review, missing #1): 8 192 instances x 100 nodes, weighted voting rights 1 + (i mod 4), an
    epoch every 50 commits with the reconfiguration FIRING -- the epoch switch of librabft-v2/src/node.rs:331-348 and the configuration read of
    bft-lib/src/simulated_context.rs:199-216 (the rights rotate by one node per epoch) -- in the fixed protocol mode (quirks = 3: the reference's own
    semantics stall at the first change), run to clock 2 500: every node of every instance passes its first epoch change
    (~33 commits per 1 000 ticks).  30.8 G events, ~46 s on the device, 93 GB of state.  Properties at full size; bit-exact against the oracle's
    digests on every instance the fixture covers (49 core-seconds of oracle time per instance: tests/golden/gen_full_size.py --count) and on a
    live sample."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from configs import CONFIGS
    import full_size_digest as fsd
    name = "c5named_8192x100_weighted_epoch_every_50_commits"
    c = CONFIGS[name]
    m, n, max_clock = c["instances"], c["nodes"], c["max_clock"]
    assert (m, n, c["commands_per_epoch"]) == (8192, 100, 50)
    kw = fsd.oracle_kwargs(c)
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    sim, res = run_gpu(amd, kw, seeds, max_clock)
    assert sim.layout()["kernel_class"] & 0x14ff == 0x1402, sim.layout()  # lbft_k_run2q
    assert not res.faults.any()
    cc, ep = res.commit_counts, res.epochs
    assert (ep >= 1).all()                                   # the reconfiguration fired at EVERY node of EVERY instance
    assert (ep == cc // 50).all()                            # read_epoch_id = commands / commands_per_epoch (simulated_context.rs:199-207)
    assert cc.min() >= 50
    hist = res.committed_histories(int(cc.max()))
    assert _prefix_consistent(cc, hist)                      # logs agree across the epoch change
    assert (hist["proposer"][np.arange(hist.shape[2])[None, None, :] < cc[:, :, None]] < n).all()
    covered, _ = _fixture_check(name, amd, oracle, res, kw, seeds, max_clock, live=16, hist=hist)
    assert covered >= 3584                                   # (3 584 instances since the round's last session: 49 core-seconds of oracle time each, computed on the GPU box's host and the build container)


def test_full_batch_math_mode_0_all_262144_nodes(amd, oracle):
    """The whole headline batch (65 536 x 4 nodes, clock 1000, ~1.3e8 events) on the HIP path against the oracle in
    math_mode = 0 -- exp / log from the HOST libm, which is what the Rust reference calls -- on every host core: commit
    counts, active rounds and State hashes of all 262 144 nodes and the aggregate counters.  (The other tests use the
    oracle's math_mode = 1, i.e. the lbft_math.h the kernels share; tests/test_math.py bridges the two on 2e7 points.)"""
    m = 65536
    seeds = np.arange(1, m + 1, dtype=np.uint64)
    _, res = run_gpu(amd, dict(num_nodes=4), seeds, 1000)
    ref = oracle.run_batch(oracle.make_config(num_nodes=4, math_mode=0), seeds, 1000, threads=HOST_THREADS, history_cap=0)
    assert not res.faults.any()
    assert (res.commit_counts == ref["commit_counts"]).all()
    assert (res.active_rounds == ref["active_rounds"]).all()
    assert (res.last_committed_states == ref["last_states"]).all()   # SipHash of every node's whole committed history
    c, rc = res.counters, ref["counters"]
    for key in ("events", "rng_draws", "rounds", "commits", "events_scheduled"):
        assert c[key] == rc[key], key
    assert c["timers_folded"] < c["events"][3] and c["node_updates"] < sum(c["events"])


# Byte-exact record hashing (first half): lbft_batch_committed_record_hashes against the oracle's own records
RECORD_HASHES = {
    "golden_n3": (dict(num_nodes=3), 1000),
    "n4": (dict(num_nodes=4), 1000),
    "n7_weighted_epochs_q2": (dict(num_nodes=7, voting_rights=[2, 1, 1, 3, 1, 2, 1], commands_per_epoch=9, quirks=2), 2000),
    "n4_rotating_rights_q3": (dict(num_nodes=4, commands_per_epoch=5, quirks=3, voting_rights=[1, 2, 3, 4], rights_rotation=1), 1500),
    "n7_equivocators": (dict(num_nodes=7, equivocate_every=3), 1000),
    "n40_long_tail": (dict(num_nodes=40, mean=10.0, variance=400.0), 300),
}


@pytest.mark.parametrize("name", sorted(RECORD_HASHES))
def test_committed_record_hashes_equal_the_oracle(amd, oracle, name):
    kw, max_clock = RECORD_HASHES[name]
    n = kw["num_nodes"]
    seeds = np.array([52, 7, 1234567], dtype=np.uint64)
    _, res = run_gpu(amd, kw, seeds, max_clock)
    assert not res.faults.any()
    cfg = oracle.make_config(math_mode=1, **kw)
    for i, seed in enumerate(seeds):
        sim = oracle.OracleSim(cfg, int(seed)).run_until(max_clock)
        for node in range(n):
            ref = sim.committed_record_hashes(node)
            got = res.committed_record_hashes(i, node)
            assert len(got) == len(ref)
            assert (got["block_hash"] == ref["block_hash"]).all() and (got["state"] == ref["state"]).all()
            assert (got["qc_hash"] == ref["qc_hash"]).all() and (got["num_votes"] == ref["num_votes"]).all()
            assert not got["flags"].any() and ref["has_qc"].all()
            if len(ref):
                assert int(got["state"][-1]) == int(res.last_committed_states[i, node])
