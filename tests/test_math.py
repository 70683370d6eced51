"""lbft_math.h (shared by the HIP kernels and the oracle's strict mode) against the host libm."""
import ctypes
import math

import numpy as np


def test_exp_bit_identical_to_libm(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-12, 12, 200000), rng.normal(2.3, 0.2, 200000), [0.0, 1e-300, -1e-20, 700.0, -740.0]])
    bad = sum(1 for x in xs if L.lbft_oracle_exp_strict(float(x)) != math.exp(float(x)) and abs(x) < 512)
    assert bad == 0


def test_exp_bit_identical_to_libm_on_2e7_points(oracle):
    """The bridge between the oracle's two arithmetic modes (and hence between the HIP path, which shares lbft_math.h, and the
    reference's glibc calls): lbft_exp == the host libm's exp, bit for bit, on 2*10^7 points -- the range the delay sampler
    uses (mu + sigma * N for the log-normal, -x^2/2 for the ziggurat wedge test) and a sweep of the whole main path |x| < 512
    (beyond it exp is deterministic on host and device but not correctly rounded; no delay model reaches it)."""
    L = oracle.lib()
    rng = np.random.default_rng(11)
    total = 0
    for lo, hi, n in ((-12.0, 12.0, 8_000_000), (-40.0, 0.0, 4_000_000), (0.0, 6.0, 6_000_000), (-511.0, 511.0, 2_000_000)):
        x = rng.uniform(lo, hi, n)
        assert L.lbft_oracle_exp_mismatches(x.ctypes.data, n) == 0, (lo, hi)
        total += n
    assert total == 20_000_000


def test_log_bit_identical_to_libm_on_3e7_points(oracle):
    """lbft_log (round 3: glibc's own algorithm with the fused multiply-adds of its FMA build) against the host libm's log, bit for bit:
    the ziggurat tail's domain (0, 1) as open01() draws it, the branch around 1, and magnitudes down to the subnormals.  (Round 2's
    fdlibm-based log was within one ulp; tools/gen_log_table.py + a C++ sweep of 1.1e8 points, profiles/r03/log_exactness.txt, had no
    difference either.)"""
    L = oracle.lib()
    rng = np.random.default_rng(12)
    bits = rng.integers(0, 2 ** 52, 10_000_000, dtype=np.uint64)
    open01 = (bits | np.uint64(1023 << 52)).view(np.float64) - (1.0 - 2.0 ** -53)   # rand 0.8 Open01
    x = np.concatenate([open01, rng.uniform(0.9375, 1.0647, 5_000_000), rng.uniform(1e-300, 1.0, 5_000_000), np.exp(rng.uniform(-700, 0, 5_000_000)),
                        np.exp(rng.uniform(0, 700, 5_000_000)), np.array([1.0, 0.5, 2.0 ** -1060, 5e-324, 0.9375, 1.0 - 2.0 ** -53])])
    ulp1 = ctypes.c_size_t()
    assert L.lbft_oracle_log_mismatches(x.ctypes.data, len(x), ctypes.byref(ulp1)) == 0
    assert ulp1.value == 0, ulp1.value


def test_fixed_delay_truncation_quirk_q5(oracle):
    L = oracle.lib()
    # `--mean m --variance 0`: delay = trunc(exp(ln m)) -> 10, 19, 49
    for mean, want in ((10.0, 10), (20.0, 19), (50.0, 49), (100.0, 100), (7.0, 6)):
        assert int(L.lbft_oracle_exp_strict(math.log(mean))) == want == int(math.exp(math.log(mean)))
        cfg = oracle.make_config(num_nodes=3, mean=mean, variance=0.0, math_mode=1)
        out = np.zeros(4, dtype=np.int64)
        L.lbft_oracle_sample_delays(ctypes.byref(cfg), 1, out.ctypes.data, 4)
        assert (out == want).all()


def test_log_equals_libm(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(2)
    for x in rng.uniform(1e-12, 1.0, 100000):
        assert L.lbft_oracle_log_strict(float(x)) == math.log(float(x))


def test_delay_streams_agree_between_math_modes(oracle):
    L = oracle.lib()
    n = 200000
    a, b = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    for var in (4.0, 400.0):
        L.lbft_oracle_sample_delays(ctypes.byref(oracle.make_config(variance=var, math_mode=0)), 7, a.ctypes.data, n)
        L.lbft_oracle_sample_delays(ctypes.byref(oracle.make_config(variance=var, math_mode=1)), 7, b.ctypes.data, n)
        assert (a == b).all()
        assert abs(a.mean() - 9.5) < 0.3  # E[trunc(LogNormal(10, var))] ~ 9.5
