"""Multi-GPU sharding of a batch: one process per GPU, instances partitioned in contiguous ranges,
no data-path collective (instances are independent: bft-lib/src/simulator.rs:26-33 owns one RNG, one
clock and one node set per Simulator).  The only collective is one all-gather of a row of
throughput counters per rank (RCCL over xGMI on GPUs -- backend "nccl" -- or gloo on CPU for the tests); sums and
maxima are reduced locally from the gathered rows."""
import numpy as np

COUNTER_KEYS = ("events0", "events1", "events2", "events3", "rng_draws", "rounds", "commits", "events_scheduled",
                "faulted_instances", "timers_folded", "node_updates")
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
    sums = [ev[0], ev[1], ev[2], ev[3]] + [counters.get(k, 0) for k in COUNTER_KEYS[4:]]
    maxs = [counters.get(k, 0) for k in MAX_KEYS]
    return sums, maxs


def gather_rows(row, group=None, device=None):
    """THE collective of a run: every rank contributes one small row of numbers (float64: exact for counters below 2^53)
    and receives all rows -- one all-gather; sums, maxima and the slowest rank's time are then reduced locally, so that
    nothing else crosses xGMI.  Returns a [world, len(row)] tensor on the CPU (a [1, len] tensor when torch.distributed
    is not initialised)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in row], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return t.reshape(1, -1).cpu()
    world = dist.get_world_size(group)
    out = torch.empty(world * t.numel(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.reshape(world, -1).cpu()


def aggregate_counters(counters, group=None, device=None, extra_max=(), extra_sum=()):
    """Aggregate the counters of every rank (sum; high-water marks with max) with ONE collective.  Works with any
    initialised torch.distributed backend; returns a dict with the same keys as lbft_counters.  `extra_max`: further
    values reduced with max in the same collective (bench.py: the timed region of the slowest rank), returned as
    out["extra_max"]; `extra_sum`: further values summed in the same collective (bench.py: the rounds / commits / events of the
    strong-scaling leg), returned as out["extra_sum"]."""
    sums, maxs = counters_to_vector(counters)
    rows = gather_rows(list(sums) + list(maxs) + list(extra_max) + list(extra_sum), group=group, device=device)
    ns, nm, nx = len(sums), len(maxs), len(extra_max)
    s = rows[:, :ns].sum(dim=0).tolist()
    m = rows[:, ns:ns + nm].max(dim=0).values.tolist()
    out = {"events": [int(v) for v in s[:4]]}
    for k, v in zip(COUNTER_KEYS[4:], s[4:]):
        out[k] = int(v)
    for k, v in zip(MAX_KEYS, m):
        out[k] = int(v)
    out["extra_max"] = rows[:, ns + nm:ns + nm + nx].max(dim=0).values.tolist() if nx else []
    out["extra_sum"] = rows[:, ns + nm + nx:].sum(dim=0).tolist() if len(extra_sum) else []
    return out
