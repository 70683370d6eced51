"""N > 1 path on CPU: world_size-2 gloo processes shard the instances exactly as bench.py does on
GPUs and all-reduce the counters.  The per-shard "simulation" is the CPU oracle here (tests may use it);
the point is the partition + the collective: the union of the shards must equal the unsharded batch."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_instances, base_seed, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import oracle_ctypes as oc
    from librabft_simulator_amd.distributed import aggregate_counters, shard_seeds
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    seeds = shard_seeds(base_seed, n_instances, rank, world)
    r = oc.run_batch(oc.make_config(num_nodes=4, math_mode=1), seeds, 600, threads=2)
    c = r["counters"]
    local = {"events": c["events"], "rng_draws": c["rng_draws"], "rounds": c["rounds"], "commits": c["commits"],
             "events_scheduled": c["events_scheduled"], "faulted_instances": 0, "max_queue": c["max_queue"],
             "max_snapshots": 0, "max_blocks": 0}
    # bench.py's second leg rides in the SAME collective: the timed regions (max over ranks) and the strong-scaling leg's sums
    total = aggregate_counters(local, extra_max=[1.5 + rank, 10.0 - rank], extra_sum=[c["rounds"], len(seeds)])
    ret[rank] = (len(seeds), int(seeds[0]), int(seeds[-1]), total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_counter_allreduce(oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n, base = 101, 7  # ragged split: 51 + 50
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n, base, ret), nprocs=2, join=True)
    assert ret[0][0] == 51 and ret[1][0] == 50
    assert ret[0][1] == base and ret[0][2] == base + 50 and ret[1][1] == base + 51 and ret[1][2] == base + 100
    assert ret[0][3] == ret[1][3]  # every rank holds the same aggregate
    whole = oracle.run_batch(oracle.make_config(num_nodes=4, math_mode=1), np.arange(base, base + n, dtype=np.uint64), 600,
                             threads=4)["counters"]
    agg = ret[0][3]
    for k in ("events", "rng_draws", "rounds", "commits", "events_scheduled", "max_queue"):
        assert agg[k] == whole[k], k
    assert agg["extra_max"] == [2.5, 10.0] and agg["extra_sum"] == [float(whole["rounds"]), float(n)]


def test_shard_range_covers_everything():
    from librabft_simulator_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
