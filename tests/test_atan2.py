"""The correctly rounded atan2 of the FX_STEP_QR kernels (fiksi_amd/csrc/fx_atan2.h, host build of the same source)
against two independent references: libquadmath's binary128 atan2q rounded to double (the oracle's
'correctly_rounded' mode) on a few million arguments, and mpmath at 200 bits on a few thousand. Also records why
the routine exists: the platform libm (what Rust's f64::atan2 calls, expressions.rs:393, :665) is within an ulp of
the correctly rounded value but not equal to it on every argument."""
import numpy as np


def _args(n, seed):
    rng = np.random.default_rng(seed)
    sets = []
    sets.append((rng.uniform(-10, 10, n), rng.uniform(-10, 10, n)))                                   # sketch-sized vectors
    e1, e2 = rng.integers(-40, 40, n), rng.integers(-40, 40, n)
    sets.append((np.ldexp(rng.uniform(-0.5, 0.5, n), e1), np.ldexp(rng.uniform(-0.5, 0.5, n), e2)))  # wide exponent range
    x = rng.uniform(0, 1, n)
    sets.append((x * (1 + rng.uniform(-0.5, 0.5, n) * 1e-6), x))                                       # near the diagonal
    sets.append((rng.uniform(-0.5, 0.5, n) * 1e-9, rng.uniform(-0.5, 0.5, n)))                        # near the axes
    a, r = rng.uniform(0, 2 * np.pi, n), np.ldexp(1 + rng.uniform(0, 1, n), rng.integers(-10, 10, n))
    sets.append((r * np.sin(a), r * np.cos(a)))                                                        # all octants
    sets.append((rng.integers(-1000, 1001, n) / 64.0, rng.integers(-1000, 1001, n) / 64.0))            # table points, zeros
    y = np.concatenate([s[0] for s in sets])
    x = np.concatenate([s[1] for s in sets])
    return y, x


def test_matches_binary128_atan2_bit_for_bit(fiksi, oracle):
    from fiksi_amd import abi

    y, x = _args(500_000, 1)
    mine = abi.atan2_cr(y, x)
    with oracle.atan2_mode("correctly_rounded"):
        ref = oracle.atan2(y, x)
    assert np.array_equal(mine.view(np.uint64), ref.view(np.uint64))
    # the platform libm: within an ulp everywhere, equal on most arguments only
    with oracle.atan2_mode("libm"):
        libm = oracle.atan2(y, x)
    assert np.max(np.abs(libm - mine) / np.maximum(np.spacing(np.abs(mine)), 5e-324)) <= 1.0
    frac = np.mean(libm.view(np.uint64) != mine.view(np.uint64))
    assert 0 < frac < 5e-3, frac  # glibc 2.35: about 7e-4


def test_matches_mpmath(fiksi):
    import mpmath
    from fiksi_amd import abi

    mpmath.mp.prec = 200
    y, x = _args(400, 2)
    mine = abi.atan2_cr(y, x)
    for yi, xi, mi in zip(y, x, mine):
        v = mpmath.atan2(mpmath.mpf(float(yi)), mpmath.mpf(float(xi)))
        if v == 0:
            assert mi == 0.0
            continue
        err = abs(mpmath.mpf(float(mi)) - v) / mpmath.mpf(float(np.spacing(abs(mi))))
        assert err <= 0.5 + 1e-9, (yi, xi, mi, float(err))


def test_special_values(fiksi):
    from fiksi_amd import abi

    y = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.inf, 1.0, np.nan, 1e-300, 1e300])
    x = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, np.inf, -np.inf, 1.0, np.inf, 1.0, 1e300, 1e-300])
    mine, ref = abi.atan2_cr(y, x), np.arctan2(y, x)
    ok = (mine.view(np.uint64) == ref.view(np.uint64)) | (np.isnan(mine) & np.isnan(ref))
    assert ok.all(), (mine, ref)
