"""CPU: the committed full-size fixture (tests/golden/full_size_digests.npz, tests/full_size_digest.py) is what the device suite compares EVERY instance of
the full-size configurations with.  Here: it is there, it covers what the suite asserts, and the oracle as built NOW still reproduces it on a few instances
per configuration (an oracle edit that changes results must regenerate the fixture in the same commit: tests/golden/gen_full_size.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import full_size_digest as fsd  # noqa: E402

# instances the CPU suite re-runs on the oracle (c5named costs ~49 core-seconds per instance: the device suite's live sample covers it)
LIVE = {"c3_65536x4": 256, "c4_16384x64_longtail_equivocators": 8, "c4live_16384x64_longtail_equivocators_fixed": 8, "c5_8192x100_weighted_epochs": 4,
        "c5live_8192x100_rotating_rights_epochs_fixed": 4, "c5named_8192x100_weighted_epoch_every_50_commits": 0}
# ... and what the device tests assert about the coverage
MIN_COVERED = {"c3_65536x4": 65536, "c4_16384x64_longtail_equivocators": 16384, "c4live_16384x64_longtail_equivocators_fixed": 16384,
               "c5_8192x100_weighted_epochs": 8192, "c5live_8192x100_rotating_rights_epochs_fixed": 8192, "c5named_8192x100_weighted_epoch_every_50_commits": 3584}


def test_fixture_covers_every_full_size_configuration():
    from configs import CONFIGS
    table, meta = fsd.load_fixture()
    assert os.path.getsize(fsd.FIXTURE) < 2 << 20          # a small fixture: 8 bytes per instance
    for name in fsd.FULL_SIZE:
        assert name in table, "no digests for %s: tests/golden/gen_full_size.py" % name
        dg, cov = table[name]
        assert len(dg) == len(cov) == CONFIGS[name]["instances"]
        assert int(cov.sum()) >= MIN_COVERED[name], (name, int(cov.sum()))
        # distinct seeds -> (nearly always) distinct runs: a constant or zero-filled column would be a broken fixture.  Not `==`: short integer-time
        # histories do coincide (c5live instances 1546 / 3311: 7 commits at every node with the same proposers and times; the stalled c4 networks)
        assert len(np.unique(dg[cov])) >= 0.95 * int(cov.sum()), (name, len(np.unique(dg[cov])))
        assert name in meta["configs"]


def test_digest_is_the_documented_function():
    import hashlib
    cc = np.array([[3, 4], [5, 6]], dtype=np.uint32)
    ar = np.array([[7, 8], [9, 10]], dtype=np.uint64)          # (the device hands active rounds over as u64: the digest takes their low 32 bits)
    ls = np.array([[11, 12], [13, 2 ** 63 + 5]], dtype=np.uint64)
    want = [int.from_bytes(hashlib.blake2b(cc[i].astype("<u4").tobytes() + ar[i].astype("<u4").tobytes() + ls[i].astype("<u8").tobytes(), digest_size=8).digest(), "little")
            for i in range(2)]
    assert [int(v) for v in fsd.digests(cc, ar, ls)] == want


@pytest.mark.parametrize("name", [n for n in fsd.FULL_SIZE if LIVE[n]])
def test_the_oracle_still_reproduces_the_fixture(oracle, name):
    from configs import CONFIGS
    table, _ = fsd.load_fixture()
    dg, cov = table[name]
    c = CONFIGS[name]
    pool = np.nonzero(cov)[0]
    idx = pool[np.unique(np.linspace(0, len(pool) - 1, LIVE[name]).astype(np.int64))]
    ref = oracle.run_batch(oracle.make_config(math_mode=1, **fsd.oracle_kwargs(c)), (idx + 1).astype(np.uint64), c["max_clock"], threads=min(os.cpu_count() or 1, 8), history_cap=0)
    assert (fsd.digests(ref["commit_counts"], ref["active_rounds"], ref["last_states"]) == dg[idx]).all()
