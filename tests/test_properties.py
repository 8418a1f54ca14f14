"""Self-certifying property tests of the oracle and of the host-side numerics, on CPU: where no
reference vector exists, the mathematics certifies the answer (KKT conditions, the definition of an order statistic,
invariances the reference's pipeline has by construction)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import stain_oracle as so
from stainlib_amd import distributed as sd


def _random_M(rng, allow_negative_correlation):
    M = rng.rand(2, 3) + 0.05
    if allow_negative_correlation:
        M[1] *= rng.choice([-1.0, 1.0], size=3)            # general atoms: the active-set enumeration, not the min/max form
    return M / np.linalg.norm(M, axis=1, keepdims=True) * rng.uniform(0.5, 1.5, size=(2, 1))


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 2 ** 31 - 1), neg=st.booleans(), lam=st.sampled_from([0.0, 0.01, 0.1, 1.0]))
def test_two_atom_lasso_satisfies_kkt(seed, neg, lam):
    """stain_utils.py:69-78 (spams.lasso, pos=True): the optimum is unique, so the KKT residual certifies ANY solver."""
    rng = np.random.RandomState(seed)
    M = _random_M(rng, neg)
    OD = np.abs(rng.randn(400, 3)) * rng.uniform(0.01, 3.0)
    OD[:20] = 1e-6                                          # background rows
    C = so.lasso2_nonneg(OD, M, lam)
    assert (C >= 0).all()
    assert so.lasso_kkt_violation(OD, M, C, lam) < 1e-9


@settings(max_examples=200, deadline=None)
@given(n=st.integers(1, 5000), pct=st.floats(0.0, 100.0), seed=st.integers(0, 1000))
def test_percentile_position_is_numpy_linear_interpolation(n, pct, seed):
    x = np.sort(np.random.RandomState(seed).rand(n))
    k, g = sd.percentile_position(n, pct)
    assert 0 <= k <= n - 1
    assert sd.np_lerp(x[k], x[min(k + 1, n - 1)], g) == np.percentile(x, pct)


@settings(max_examples=200, deadline=None)
@given(v=st.floats(allow_nan=False, allow_infinity=False, width=32))
def test_ordered_uint32_key_round_trip_and_order(v):
    """The selection keys are binary32 values compared as uint32: the map must be a bijection that preserves <."""
    a = np.float32(v)
    b = np.nextafter(a, np.float32(np.inf), dtype=np.float32)
    oa, ob = int(_f2ord(a)), int(_f2ord(b))
    if a == 0.0 and b == 0.0:
        return
    assert (oa < ob) == (a < b) or a == b
    assert sd.ord_to_float(oa) == a or (a == 0.0 and sd.ord_to_float(oa) == 0.0)


def _f2ord(a):
    u = np.asarray(a, np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)


def test_macenko_matrix_is_invariant_under_pixel_permutation_and_tiling():
    """macenko_stain_extractor.py:18-44 only sees the multiset of tissue pixels: any pixel order, any image shape."""
    I = so.synth_tile(96, 128, 5)
    M = so.macenko_stain_matrix(I)
    perm = np.random.RandomState(0).permutation(96 * 128)
    J = I.reshape(-1, 3)[perm].reshape(48, 256, 3)
    np.testing.assert_allclose(so.macenko_stain_matrix(J), M, rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(M, axis=1), 1.0, atol=1e-15)       # unit-norm rows (:44)
    assert M[0, 0] > M[1, 0]                                                      # H first (:40-43)


def test_transform_keeps_shape_dtype_and_is_deterministic():
    n = so.ExtractiveStainNormalizer("macenko")
    n.fit(so.synth_tile(64, 64, 1001, so.M_TRUE_TGT))
    I = so.synth_tile(40, 72, 3)
    a, b = n.transform(I), n.transform(I.copy())
    assert a.shape == I.shape and a.dtype == np.uint8 and np.array_equal(a, b)


def test_vahadane_fixed_point_is_stationary():
    """No oracle exists for spams.trainDL: the converged dictionary must be a fixed point of the
    block-coordinate update -- one more sweep moves it by less than the tolerance -- with non-negative unit-norm rows."""
    I = so.synth_tile(48, 48, 9)
    info = {}
    M = so.vahadane_stain_matrix(I, max_sweeps=600, tol=1e-11, info=info)
    M2 = so.vahadane_stain_matrix(I, max_sweeps=info.get("sweeps", 600) + 1, tol=0.0) if "sweeps" in info else M
    np.testing.assert_allclose(M2, M, rtol=0, atol=1e-8)
    assert (M >= 0).all() and np.allclose(np.linalg.norm(M, axis=1), 1.0, atol=1e-12) and M[0, 0] >= M[1, 0]
