"""Full-size parity without oracle time in the device suite (test infrastructure).

`tests/golden/full_size_digests.npz` holds, for every full-size configuration of tools/configs.py, ONE 64-bit digest per instance of what the
parity checks compare -- the commit count, the active round and the State hash (a SipHash of the node's whole committed history,
bft-lib/src/simulated_context.rs:51-55) of every node of the instance -- computed from the CPU ORACLE's results by tests/golden/gen_full_size.py
(run offline: hours of CPU time that no suite run pays again).  A device test hashes its own results the same way and compares every instance the
fixture covers; a small live oracle sample stays in each test as a drift guard for the fixture itself (an oracle edit that changes results shows up as
"live oracle != fixture", not as a device failure).

digest(i) = blake2b-64( u32le commit_counts[i, :] || u32le active_rounds[i, :] || u64le last_states[i, :] )
"""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "full_size_digests.npz")

# the six full-size configurations (names of tools/configs.py CONFIGS)
FULL_SIZE = (
    "c3_65536x4",
    "c4_16384x64_longtail_equivocators",
    "c4live_16384x64_longtail_equivocators_fixed",
    "c5_8192x100_weighted_epochs",
    "c5live_8192x100_rotating_rights_epochs_fixed",
    "c5named_8192x100_weighted_epoch_every_50_commits",
)


def oracle_kwargs(c):
    """tools/configs.py entry -> keyword arguments of oracle_ctypes.make_config / tests' run_gpu."""
    kw = dict(num_nodes=c["nodes"])
    if "uniform" in c:
        kw.update(delay_model=1, uniform_lo=c["uniform"][0], uniform_hi=c["uniform"][1])
    else:
        kw.update(mean=10.0, variance=c.get("variance", 4.0))
    for src, dst in (("commands_per_epoch", "commands_per_epoch"), ("quirks", "quirks"), ("rights_rotation", "rights_rotation"),
                     ("equivocate_every", "equivocate_every"), ("weights", "voting_rights")):
        if c.get(src):
            kw[dst] = c[src]
    return kw


def digests(commit_counts, active_rounds, last_states):
    """One u64 per instance (see the module docstring); the arrays are [instances, nodes]."""
    cc = np.ascontiguousarray(np.asarray(commit_counts).astype("<u4"))
    ar = np.ascontiguousarray(np.asarray(active_rounds).astype("<u4"))
    ls = np.ascontiguousarray(np.asarray(last_states).astype("<u8"))
    assert cc.shape == ar.shape == ls.shape and cc.ndim == 2
    out = np.empty(cc.shape[0], dtype=np.uint64)
    for i in range(cc.shape[0]):
        h = hashlib.blake2b(digest_size=8)
        h.update(cc[i].tobytes())
        h.update(ar[i].tobytes())
        h.update(ls[i].tobytes())
        out[i] = int.from_bytes(h.digest(), "little")
    return out


_cache = None


def load_fixture():
    """{config name: (digests u64[instances], covered bool[instances])}, meta dict; empty when the fixture file is absent."""
    global _cache
    if _cache is None:
        table, meta = {}, {}
        if os.path.exists(FIXTURE):
            with np.load(FIXTURE, allow_pickle=False) as z:
                meta = json.loads(str(z["meta"]))
                for name in FULL_SIZE:
                    if name in z.files:
                        table[name] = (z[name], z[name + "__covered"].astype(bool))
        _cache = (table, meta)
    return _cache


def compare(name, commit_counts, active_rounds, last_states):
    """Device results of full-size configuration `name` against the fixture: (instances compared, indices that differ)."""
    table, _ = load_fixture()
    assert name in table, "no fixture for %s in %s: run tests/golden/gen_full_size.py" % (name, FIXTURE)
    want, covered = table[name]
    assert len(want) == len(commit_counts), (len(want), len(commit_counts))
    got = digests(commit_counts, active_rounds, last_states)
    bad = np.nonzero(covered & (got != want))[0]
    return int(covered.sum()), bad
