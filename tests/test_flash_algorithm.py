"""The arithmetic of the prompt attention kernels (csrc/flash_prefill.hip), restated in numpy and held against the float64 softmax on the CPU:
the lazy reference maximum (it only moves when a row maximum grows by more than 2^8 in the exp2 domain; O and l are rescaled once when it
does), P rounded to fp16 before it multiplies V, fully masked tiles computed instead of skipped (reference point initialised to -1e30, not
-inf), and -- the 8-wave kernel -- two tile sets (even / odd key tiles) with their own (O, m, l), merged at the end.  The GPU tests hold the
kernels themselves against the oracle; this one pins WHY their algorithm is right, including the cases random data never reaches."""
import numpy as np
import pytest

C1 = (1.0 / np.sqrt(128.0)) * 1.4426950408889634


def _one_set(q, k, v, limit, tiles, m0=-1e30):
    """One wave's loop over its key tiles (64 keys each) for one query row: returns (O, m, l) at the lazy reference point."""
    m, l = np.float32(m0), np.float32(0.0)
    o = np.zeros(v.shape[1], dtype=np.float32)
    for t in tiles:
        keys = np.arange(t * 64, t * 64 + 64)
        kk = np.clip(keys, 0, k.shape[0] - 1)                              # rows past the last key re-read the last row
        s = (k[kk].astype(np.float32) @ q.astype(np.float32)).astype(np.float32)
        s = np.where(keys <= limit, s, -np.inf).astype(np.float32)
        mx = s.max()
        with np.errstate(invalid="ignore", over="ignore"):
            m_new = mx if (mx - m) * np.float32(C1) > 8.0 else m
            alpha = np.exp2((m - m_new) * np.float32(C1), dtype=np.float32)
            p = np.exp2(s * np.float32(C1) - m_new * np.float32(C1)).astype(np.float32)
        assert np.isfinite(alpha) and np.isfinite(p).all()
        assert p.max() <= 256.0 * 1.0001                                   # the fp16 range P is kept in
        l = l * alpha + p.sum(dtype=np.float32)
        o = o * alpha + p.astype(np.float16).astype(np.float32) @ v[kk].astype(np.float32)
        m = m_new
    return o, m, l


def _kernel_model(q, k, v, limit, two_sets):
    ntiles = limit // 64 + 1
    if not two_sets:
        o, m, l = _one_set(q, k, v, limit, range(ntiles))
        return o / l
    niter = (ntiles + 1) // 2                                                # the odd set may get a tile behind the last one: fully masked
    oa, ma, la = _one_set(q, k, v, limit, [2 * j for j in range(niter)])
    ob, mb, lb = _one_set(q, k, v, limit, [2 * j + 1 for j in range(niter)])
    m = max(ma, mb)
    fa, fb = np.exp2((ma - m) * np.float32(C1)), np.exp2((mb - m) * np.float32(C1))
    return (oa * fa + ob * fb) / (la * fa + lb * fb)


def _truth(q, k, v, limit):
    s = (k[:limit + 1].astype(np.float64) @ q.astype(np.float64)) / np.sqrt(128.0)
    p = np.exp(s - s.max())
    return (p / p.sum()) @ v[:limit + 1].astype(np.float64)


@pytest.mark.parametrize("two_sets", [False, True])
@pytest.mark.parametrize("limit", [0, 5, 63, 64, 127, 128, 200, 700])
def test_lazy_reference_maximum_and_set_merge(two_sets, limit):
    rng = np.random.default_rng(limit + 17 * two_sets)
    n = 768
    q = rng.standard_normal(128).astype(np.float16)
    k = rng.standard_normal((n, 128)).astype(np.float16)
    v = rng.standard_normal((n, 128)).astype(np.float16)
    got = _kernel_model(q, k, v, limit, two_sets)
    ref = _truth(q, k, v, limit)
    assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("two_sets", [False, True])
@pytest.mark.parametrize("spikes", [(3,), (70,), (130, 131), (500,), (64, 320, 699)])
def test_a_key_that_dominates_late_moves_the_reference_point_once(two_sets, spikes):
    """Keys whose score exceeds everything before them by far more than 2^8: in the first tile, in an odd tile, in both sets, in the last tile."""
    rng = np.random.default_rng(sum(spikes))
    n, limit = 768, 699
    q = rng.standard_normal(128).astype(np.float16)
    k = rng.standard_normal((n, 128)).astype(np.float16)
    v = rng.standard_normal((n, 128)).astype(np.float16)
    for i, s in enumerate(spikes):
        k[s] = ((2.0 + i) * q.astype(np.float32)).astype(np.float16)        # later spikes dominate earlier ones
    got = _kernel_model(q, k, v, limit, two_sets)
    ref = _truth(q, k, v, limit)
    assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    assert np.abs(got - v[spikes[-1]].astype(np.float64)).max() < 2e-2      # the last spike owns the row


def test_a_set_without_a_visible_key_contributes_nothing():
    """One key tile only (the odd set's tile lies behind the last key) and rows whose odd tile is entirely in their future: m stays at
    -1e30, l at 0, and the merge weights that set with exp2(-huge) = 0 instead of NaN (what an initial -inf would give)."""
    rng = np.random.default_rng(4)
    q = rng.standard_normal(128).astype(np.float16)
    k = rng.standard_normal((64, 128)).astype(np.float16)
    v = rng.standard_normal((64, 128)).astype(np.float16)
    o, m, l = _one_set(q, k, v, limit=40, tiles=[1])
    assert m == np.float32(-1e30) and l == 0.0 and not o.any()
    got = _kernel_model(q, k, v, 40, two_sets=True)
    assert np.isfinite(got).all() and np.abs(got - _truth(q, k, v, 40)).max() < 2e-3
    with np.errstate(invalid="ignore"):
        assert np.isnan(np.exp2((np.float32(-np.inf) - np.float32(-np.inf)) * np.float32(C1)))   # the NaN the finite start avoids
