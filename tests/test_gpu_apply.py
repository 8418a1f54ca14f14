"""-m gpu: the OD + reconstruction pass (sl_normalize_apply) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import stain_oracle as so
from tests.gpu_util import oracle_fit_tile, to_dev, u8_parity

pytestmark = pytest.mark.gpu


def _case(h, w, seeds):
    from stainlib_amd import engine
    tiles = [so.synth_tile(h, w, s) for s in seeds]
    tgt = so.synth_tile(h, w, 1001, so.M_TRUE_TGT)
    Mt, maxCt = oracle_fit_tile(tgt)
    fits = [oracle_fit_tile(I) for I in tiles]
    M = np.stack([f[0] for f in fits])
    mc = np.stack([f[1] for f in fits])
    out, pre = engine.normalize_apply(to_dev(tiles), M, mc, Mt, maxCt, want_prequant=True)
    out2 = engine.normalize_apply(to_dev(tiles), M, mc, Mt, maxCt)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    out, pre = out.cpu().numpy(), pre.cpu().numpy()
    for i, I in enumerate(tiles):
        C = so.get_concentrations(I, M[i]) * (maxCt / mc[i])
        want_pre = (255 * np.exp(-C @ Mt)).reshape(I.shape)
        rel = np.abs(pre[i] - want_pre) / np.maximum(np.abs(want_pre), 1e-30)
        assert rel.max() < 1e-4, rel.max()          # north_star tolerance: 1e-4 relative on reconstructed RGB
        assert rel.max() < 2e-5                     # what binary32 actually delivers
        u8_parity(out[i], so.truncate_u8(want_pre))


@pytest.mark.parametrize("h,w", [(64, 64), (256, 256), (96, 130), (33, 47)])
def test_apply_vs_oracle(h, w):
    _case(h, w, [2, 3, 4])


def test_apply_golden_256(golden_dir):
    """Golden vector from the reference run: full uint8 output of fit(target)+transform(I)."""
    from stainlib_amd import engine
    g = np.load(f"{golden_dir}/macenko_256_s2.npz")
    I = so.synth_tile(256, 256, 2)
    out = engine.normalize_apply(to_dev([I]), g["M"][None], g["maxC"], g["M_target"], g["maxC_target"].reshape(2))
    u8_parity(out.cpu().numpy()[0], g["out"])


def test_apply_large_linearity():
    """Full-size property: out depends on pixels independently -> shuffling pixels commutes."""
    from stainlib_amd import engine
    I = so.synth_tile(1024, 1024, 5)
    M, mc = oracle_fit_tile(I[:256, :256])
    Mt, mct = oracle_fit_tile(so.synth_tile(256, 256, 1001, so.M_TRUE_TGT))
    perm = np.random.RandomState(0).permutation(1024 * 1024)
    J = I.reshape(-1, 3)[perm].reshape(I.shape)
    a = engine.normalize_apply(to_dev([I, J]), np.stack([M, M]), np.stack([mc, mc]), Mt, mct).cpu().numpy()
    assert np.array_equal(a[0].reshape(-1, 3)[perm], a[1].reshape(-1, 3))


def _general_case(M_src, M_tgt, tiles, maxC_src, maxC_tgt):
    """sl_normalize_apply with arbitrary matrices against the oracle: lasso by active-set enumeration, 255*exp(-C M_t),
    truncation toward zero THEN wrap modulo 256 (the reference's astype(uint8), normalizer.py:50, SURVEY a-G)."""
    from stainlib_amd import engine
    n = len(tiles)
    out, pre = engine.normalize_apply(to_dev(tiles), np.stack([M_src] * n), np.stack([maxC_src] * n), M_tgt, maxC_tgt, want_prequant=True)
    out, pre = out.cpu().numpy(), pre.cpu().numpy()
    stats = []
    for i, I in enumerate(tiles):
        C = so.get_concentrations(I, M_src) * (maxC_tgt / maxC_src)
        want_pre = (255 * np.exp(-C @ M_tgt)).reshape(I.shape)
        rel = np.abs(pre[i] - want_pre) / np.maximum(np.abs(want_pre), 1e-30)
        assert rel.max() < 2e-5, rel.max()
        want = so.truncate_u8(want_pre)
        d = out[i].astype(np.int16) - want.astype(np.int16)
        # a byte may differ by one level, or -- where the value crosses a multiple of 256 -- by 255 (0 <-> 255 wrap)
        assert np.isin(np.abs(d), (0, 1, 255)).all(), np.unique(np.abs(d))
        flips = int((d != 0).sum())
        print(f"general path tile {i}: {flips} of {d.size} bytes differ; values above 255: {int((want_pre >= 256).sum())}")
        assert flips <= max(4, int(1e-4 * d.size))
        stats.append((float(want_pre.max()), C))
    return stats


def test_general_path_negative_correlation_and_wrap():
    """The paths no H&E fixture reaches: a stain pair with NEGATIVE correlation (lasso2's g12 < 0
    active-set enumeration) and a target matrix with a negative entry (values pass 255: pack_trunc_general, wrap modulo
    256; FAST = false instantiations of k_apply)."""
    tiles = [so.synth_tile(96, 130, s) for s in (2, 3)]
    ragged = [so.synth_tile(33, 47, 4)]                                                      # unaligned loads / byte-wise tail
    M_neg = so.normalize_rows(np.array([[0.9, -0.3, 0.3], [-0.2, 0.95, 0.25]]))            # g12 < 0
    assert M_neg[0] @ M_neg[1] < 0
    M_tgt_pos = so.normalize_rows(so.M_TRUE_TGT)
    st = _general_case(M_neg, M_tgt_pos, tiles, np.array([1.7, 1.3]), np.array([1.5, 1.1]))
    # all four active sets occur under the negatively correlated pair
    C = st[0][1]
    assert ((C[:, 0] > 0) & (C[:, 1] > 0)).any() and ((C[:, 0] > 0) & (C[:, 1] == 0)).any() and ((C[:, 0] == 0) & (C[:, 1] > 0)).any()
    # target with a negative entry: exp(+...) pushes values past 255 -> the cast wraps
    M_tgt_neg = so.normalize_rows(np.array([[0.55, 0.80, -0.25], [0.10, 0.95, 0.20]]))
    M_src = so.normalize_rows(so.M_TRUE_SRC)
    st = _general_case(M_src, M_tgt_neg, tiles, np.array([1.6, 1.2]), np.array([2.4, 1.0]))
    assert max(s[0] for s in st) > 256.0, "the case must actually exceed 255"
    # both at once, also on a tile whose byte count is not a multiple of 4
    _general_case(M_neg, M_tgt_neg, tiles, np.array([1.7, 1.3]), np.array([2.0, 1.1]))
    _general_case(M_neg, M_tgt_neg, ragged, np.array([1.7, 1.3]), np.array([2.0, 1.1]))


def test_general_path_concentrations_and_stain_augment():
    """get_concentrations and StainAugmentor.pop under a negatively correlated pair (FAST = false in augment_sweep)."""
    from stainlib_amd import engine
    tiles = [so.synth_tile(80, 100, s) for s in (5, 6)]
    M_neg = so.normalize_rows(np.array([[0.9, -0.3, 0.3], [-0.2, 0.95, 0.25]]))
    Cg = engine.concentrations(to_dev(tiles), np.stack([M_neg, M_neg])).cpu().numpy()
    for i, I in enumerate(tiles):
        Co = so.get_concentrations(I, M_neg)
        np.testing.assert_allclose(Cg[i], Co, rtol=0, atol=5e-6)
        assert so.lasso_kkt_violation(so.rgb_to_od(I).reshape(-1, 3), M_neg, Cg[i].astype(np.float64), 0.01) < 5e-6
    ab = np.array([[1.15, 0.07, 0.9, -0.05], [0.85, -0.1, 1.1, 0.12]])
    for bg in (False, True):
        out = engine.stain_augment(to_dev(tiles), np.stack([M_neg, M_neg]), ab, augment_background=bg).cpu().numpy()
        for i, I in enumerate(tiles):
            a = so.StainAugmentor("macenko", augment_background=bg)
            a.image_shape, a.stain_matrix = I.shape, M_neg
            a.source_concentrations = so.get_concentrations(I, M_neg)
            a.tissue_mask = so.tissue_mask(I).ravel()
            u8_parity(out[i], a.pop_with([ab[i, 0], ab[i, 2]], [ab[i, 1], ab[i, 3]]), label=f"stain_augment g12<0 bg={bg}")
