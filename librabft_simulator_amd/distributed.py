"""Multi-GPU sharding of a batch: one process per GPU, instances partitioned in contiguous ranges,
no data-path collective (instances are independent: bft-lib/src/simulator.rs:26-33 owns one RNG, one
clock and one node set per Simulator).  The only collective is one all-reduce of the throughput
counters (RCCL over xGMI on GPUs -- backend "nccl" -- or gloo on CPU for the tests)."""
import numpy as np

COUNTER_KEYS = ("events0", "events1", "events2", "events3", "rng_draws", "rounds", "commits", "events_scheduled",
                "faulted_instances")
MAX_KEYS = ("max_queue", "max_snapshots", "max_blocks")


def shard_range(n_instances, rank, world_size):
    """Contiguous partition: instance i -> rank i // ceil(n / world)."""
    per = -(-n_instances // world_size)
    lo = min(rank * per, n_instances)
    hi = min(lo + per, n_instances)
    return lo, hi


def shard_seeds(base_seed, n_instances, rank, world_size):
    """seed_i = base_seed + i for the global instance index i; returns this rank's slice."""
    lo, hi = shard_range(n_instances, rank, world_size)
    return (np.arange(lo, hi, dtype=np.uint64) + np.uint64(base_seed)).astype(np.uint64)


def counters_to_vector(counters):
    ev = counters["events"]
    sums = [ev[0], ev[1], ev[2], ev[3]] + [counters[k] for k in COUNTER_KEYS[4:]]
    maxs = [counters.get(k, 0) for k in MAX_KEYS]
    return sums, maxs


def aggregate_counters(counters, group=None, device=None):
    """All-reduce the counters of every rank (sum; high-water marks with max).  Works with any
    initialised torch.distributed backend; returns a dict with the same keys as lbft_counters."""
    import torch
    import torch.distributed as dist
    sums, maxs = counters_to_vector(counters)
    if not (dist.is_available() and dist.is_initialized()):
        s, m = sums, maxs
    else:
        ts = torch.tensor(sums, dtype=torch.int64, device=device)
        tm = torch.tensor(maxs, dtype=torch.int64, device=device)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX, group=group)
        s, m = ts.tolist(), tm.tolist()
    out = {"events": [int(v) for v in s[:4]]}
    for k, v in zip(COUNTER_KEYS[4:], s[4:]):
        out[k] = int(v)
    for k, v in zip(MAX_KEYS, m):
        out[k] = int(v)
    return out
