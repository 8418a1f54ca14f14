"""-m gpu: Vahadane (sparse-NMF) path.  The reference's spams.trainDL call is wall-clock budgeted and
randomly initialised, so parity is defined against the CONVERGED optimum of the same objective (oracle:
full-batch block-coordinate descent, run to 1e-12).  SURVEY 8c allows 1e-5 on the unit-norm rows; the tests hold the kernel to
what it promises and delivers (round-3 review, item 5): the iteration's own stopping tolerance, 1e-7 by default (measured: 3e-9 to
7e-8; test_vahadane_error_stays_within_the_tolerance checks the promise "error <= 0.7 dl_tol" over five tolerances), and the north
star's 1e-4 of the bytes."""
import numpy as np
import pytest
import torch

from oracle import stain_oracle as so
from tests.gpu_util import to_dev, u8_parity

pytestmark = pytest.mark.gpu
V_ATOL = 1e-7


def _oracle_fit(I):
    info = {}
    M = so.vahadane_stain_matrix(I, max_sweeps=600, tol=1e-12, info=info)
    C = so.get_concentrations(I, M)
    return M, np.percentile(C, 99, axis=0), info


# both schedules: 1 = one launch per phase (automatic below 640 tiles), 2 = the persistent fused kernel
@pytest.mark.parametrize("schedule", [1, 2])
@pytest.mark.parametrize("h,w", [(96, 96), (128, 160), (33, 47), (32, 40)])   # (32,40): waves without pixels
def test_vahadane_fit_vs_converged_oracle(h, w, schedule):
    from stainlib_amd import engine
    tiles = [so.synth_tile(h, w, s) for s in (2, 3, 4)]
    p = engine.make_params(dl_tol=1e-9, schedule=schedule)
    M, mc, st, sweeps = engine.vahadane_fit(to_dev(tiles), params=p)
    M, mc, st, sweeps = M.cpu().numpy(), mc.cpu().numpy(), st.cpu().numpy(), sweeps.cpu().numpy()
    assert (st == 0).all() and (sweeps < 40).all() and (sweeps >= 2).all()
    for i, I in enumerate(tiles):
        Mo, mco, info = _oracle_fit(I)
        np.testing.assert_allclose(M[i], Mo, rtol=0, atol=V_ATOL)
        np.testing.assert_allclose(mc[i], mco, rtol=1e-4)
        np.testing.assert_allclose(np.linalg.norm(M[i], axis=1), 1.0, atol=1e-12)
        assert M[i][0, 0] >= M[i][1, 0]                      # H first (vahadane_stain_extractor.py:40)
        assert (M[i] >= 0).all()                             # posD
        # stationarity certificate: the objective at our D is not worse than at the oracle's converged D
        OD = so.rgb_to_od(I).reshape(-1, 3)[so.tissue_mask(I).ravel()]

        def obj(D):
            Cc = so.lasso2_nonneg(OD, D, 0.1)
            r = OD - Cc @ D
            return (0.5 * (r * r).sum(1) + 0.1 * Cc.sum(1)).mean()
        assert obj(M[i]) <= obj(Mo) + 1e-9


def test_vahadane_default_tolerance_and_transform():
    import stainlib_amd as sl
    from stainlib_amd import engine
    I = so.synth_tile(160, 160, 7)
    tgt = so.synth_tile(160, 160, 1001, so.M_TRUE_TGT)
    n = sl.VahadaneNormalizer()
    n.fit(tgt)
    Mo, mco, _ = _oracle_fit(tgt)
    np.testing.assert_allclose(n.stain_matrix_target, Mo, rtol=0, atol=V_ATOL)
    out = n.transform(I)
    on = so.ExtractiveStainNormalizer("vahadane")
    on.stain_matrix_target, on.maxC_target = Mo, mco.reshape(1, 2)
    Ms, mcs, _ = _oracle_fit(I)
    Cs = so.get_concentrations(I, Ms) * (on.maxC_target / mcs)
    want = so.truncate_u8(255 * np.exp(-Cs @ Mo)).reshape(I.shape)
    d = np.abs(out.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1 and (d != 0).sum() <= max(4, 1e-4 * d.size)   # the dictionary agrees to < 1e-7: a few bytes flip by one level
    M1 = sl.VahadaneStainExtractor.get_stain_matrix(I)
    np.testing.assert_allclose(M1, Ms, rtol=0, atol=V_ATOL)
    # batch: failed tile passes through, others unaffected; sweeps reported
    white = np.full((160, 160, 3), 255, np.uint8)
    o, M, mc, st = engine.vahadane_transform(to_dev([I, white, I]), Mo, mco)
    assert list(st.cpu().numpy()) == [0, 1, 0]
    assert np.array_equal(o[1].cpu().numpy(), white) and torch.equal(o[0], o[2])
    assert np.array_equal(o[0].cpu().numpy(), out)
    sa = sl.StainAugmentor("vahadane")
    sa.fit(I)
    np.testing.assert_allclose(sa.stain_matrix, Ms, rtol=0, atol=V_ATOL)


def test_vahadane_512_tile():
    from stainlib_amd import engine
    I = so.synth_tile(512, 512, 11)
    M, mc, st, sweeps = engine.vahadane_fit(to_dev([I]))
    Mo, mco, info = _oracle_fit(I)
    np.testing.assert_allclose(M.cpu().numpy()[0], Mo, rtol=0, atol=V_ATOL)
    assert int(sweeps[0]) < 30


def test_vahadane_1024_tiles_fit_and_transform():
    """BASELINE configs[2] tile size: 1024 x 1024, fit and transform, both schedules, against the converged oracle (whose
    dictionary tests/golden/vahadane_pin_*.npz pin to scikit-learn's DictionaryLearning)."""
    from stainlib_amd import engine
    tiles = [so.synth_tile(1024, 1024, s) for s in (21, 22)]
    tgt = so.synth_tile(1024, 1024, 1001, so.M_TRUE_TGT)
    Mto, mcto, _ = _oracle_fit(tgt)
    fits = [_oracle_fit(I) for I in tiles]
    outs = []
    for sched in (1, 2):
        p = engine.make_params(schedule=sched, dl_tol=1e-7)
        Mt, mct, st, _ = engine.vahadane_fit(to_dev([tgt]), params=p)
        np.testing.assert_allclose(Mt[0].cpu().numpy(), Mto, rtol=0, atol=V_ATOL)
        out, M, mc, st = engine.vahadane_transform(to_dev(tiles), Mt[0], mct[0], params=p)
        assert (st.cpu().numpy() == 0).all()
        for i, I in enumerate(tiles):
            np.testing.assert_allclose(M[i].cpu().numpy(), fits[i][0], rtol=0, atol=V_ATOL)
            print(f"vahadane 1024^2 schedule {sched} tile {i}: maxC rel err {np.abs(mc[i].cpu().numpy() / fits[i][1] - 1).max():.1e}")
            np.testing.assert_allclose(mc[i].cpu().numpy(), fits[i][1], rtol=2e-6)
            Cs = so.get_concentrations(I, fits[i][0]) * (mcto / fits[i][1])
            want = so.truncate_u8(255 * np.exp(-Cs @ Mto)).reshape(I.shape)
            d = np.abs(out[i].cpu().numpy().astype(np.int16) - want.astype(np.int16))
            flips = int((d != 0).sum())
            print(f"vahadane 1024^2 schedule {sched} tile {i}: {flips} of {d.size} bytes differ ({flips / d.size:.2e}); "
                  f"|dM| {np.abs(M[i].cpu().numpy() - fits[i][0]).max():.1e}")
            # the dictionary agrees with the oracle's to < 1e-7 (its own stopping tolerance), which moves every pixel a little:
            # measured 10-30 of 3.1 M bytes; the bar is twice that (round-5 review: it stood at the north star's 1e-4 N = 314)
            assert d.max() <= 1 and flips <= 64
        outs.append(out)
    assert (outs[0] != outs[1]).float().mean().item() < 1e-4


def test_vahadane_schedules_agree():
    """The per-phase schedule and the fused kernel run the same iteration (only the binary64 summation order differs);
    a tight tolerance forces more full sweeps than the fixed launches provide, so the straggler path runs too."""
    from stainlib_amd import engine
    rng = np.random.default_rng(5)
    tiles = [so.synth_tile(192, 256, 100 + s) for s in range(6)]
    tiles[3] = np.full((192, 256, 3), 255, np.uint8)                     # empty mask
    tiles[4] = rng.integers(0, 256, (192, 256, 3), dtype=np.uint8)       # no structure at all
    tgt = so.synth_tile(192, 256, 1001, so.M_TRUE_TGT)
    Mt, mct, _, _ = engine.vahadane_fit(to_dev([tgt]))
    for tol, cap in ((1e-7, 100), (1e-12, 100), (0.0, 7)):      # (tol 0 never settles: sweeps 5-7 run in the straggler kernel)
        res = []
        for schedule in (1, 2):
            p = engine.make_params(dl_tol=tol, dl_max_sweeps=cap, schedule=schedule)
            M, mc, st, sweeps = engine.vahadane_fit(to_dev(tiles), params=p)
            o, M2, mc2, st2 = engine.vahadane_transform(to_dev(tiles), Mt[0], mct[0], params=p)
            assert torch.equal(st, st2)
            np.testing.assert_allclose(M.cpu().numpy(), M2.cpu().numpy(), rtol=0, atol=1e-13, equal_nan=True)
            res.append((M.cpu().numpy(), mc.cpu().numpy(), st.cpu().numpy(), sweeps.cpu().numpy(), o.cpu().numpy()))
        a, b = res
        assert list(a[2]) == list(b[2]) and a[2][3] == 1
        np.testing.assert_allclose(a[0], b[0], rtol=0, atol=max(10 * tol, 1e-11), equal_nan=True)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-6, equal_nan=True)
        d = np.abs(a[4].astype(np.int16) - b[4].astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 1e-4
        if tol == 0.0:
            ok = a[2] == 0
            assert (a[3][ok] == 7).all() and (b[3][ok] == 7).all()         # more than the 4 full-sweep launches: the tail kernel ran


@pytest.mark.parametrize("max_sweeps", [2, 3, 7])   # (after a single sweep the two atoms are still nearly collinear: the codes are ill-conditioned)
def test_vahadane_sweep_budget_both_schedules(max_sweeps):
    """dl_max_sweeps caps the FULL sweeps (the reference budgets wall-clock time instead); both schedules stop at the
    same iterate, whether the cap falls inside the fixed launches or in the straggler kernel."""
    from stainlib_amd import engine
    tiles = [so.synth_tile(160, 224, 300 + s) for s in range(4)]
    res = []
    for schedule in (1, 2):
        p = engine.make_params(dl_tol=1e-14, dl_max_sweeps=max_sweeps, schedule=schedule)
        M, mc, st, sweeps = engine.vahadane_fit(to_dev(tiles), params=p)
        sw = sweeps.cpu().numpy()
        # (two or three sweeps cannot reach 1e-14; seven can: the iteration is Newton-like and lands on the fixed point of the
        #  binary32 bursts after four or five, where it stops on its own)
        assert (st.cpu().numpy() == 0).all() and ((sw == max_sweeps).all() if max_sweeps <= 3 else ((sw >= 4) & (sw <= max_sweeps)).all())
        swl = res[-1][2] if res else sw
        assert (sw == swl).all()
        res.append((M.cpu().numpy(), mc.cpu().numpy(), sw))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0, atol=1e-11)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-6)


def test_vahadane_error_stays_within_the_tolerance():
    """The iteration stops on an a-posteriori estimate of its distance to the fixed point (stats_kernels.hpp: dict_advance),
    not after a confirming sweep: against the oracle converged to 1e-12 the dictionary error must stay within a few dl_tol
    (plus the ~3e-8 floor of the binary32 class moments), for loose and tight tolerances, on tiles of two sizes."""
    from stainlib_amd import engine
    for size, seeds in ((192, (31, 32, 33, 34)), (512, (41, 42))):
        tiles = [so.synth_tile(size, size, s) for s in seeds]
        refs = [so.vahadane_stain_matrix(I, max_sweeps=600, tol=1e-12) for I in tiles]
        for tol in (1e-4, 1e-5, 1e-6, 1e-7, 1e-9):
            for schedule in (1, 2):
                M, mc, st, sweeps = engine.vahadane_fit(to_dev(tiles), params=engine.make_params(dl_tol=tol, schedule=schedule))
                err = max(float(np.abs(M[i].cpu().numpy() - refs[i]).max()) for i in range(len(tiles)))
                print(f"size {size} dl_tol {tol:.0e} schedule {schedule}: sweeps {sweeps.cpu().numpy().tolist()} max |dM| {err:.1e}")
                assert (st.cpu().numpy() == 0).all()
                assert err <= 3.0 * tol + 1e-7


def test_vahadane_on_spatially_smooth_tiles():
    """Nuclei, slow eosin gradients, a lumen, little noise (oracle.structured_tile 'blobs'): neighbouring pixels are strongly
    correlated, so the stratified sample the iteration starts from is a poorer stand-in than on i.i.d. tiles.  The result
    must not care: converged oracle within 1e-7, both schedules, and the Macenko path takes no exact fallback on them."""
    from stainlib_amd import engine
    tiles = [so.structured_tile("blobs", 256, 320, s) for s in (5, 6, 7)]
    fits = [_oracle_fit(I) for I in tiles]
    for schedule in (1, 2):
        M, mc, st, sweeps = engine.vahadane_fit(to_dev(tiles), params=engine.make_params(schedule=schedule))
        print("blobs: schedule", schedule, "sweeps", sweeps.cpu().numpy().tolist())
        assert (st.cpu().numpy() == 0).all() and (sweeps.cpu().numpy() <= 6).all()
        for i in range(len(tiles)):
            np.testing.assert_allclose(M[i].cpu().numpy(), fits[i][0], rtol=0, atol=V_ATOL)
            np.testing.assert_allclose(mc[i].cpu().numpy(), fits[i][1], rtol=1e-4)
    p = engine.make_params()
    fb = engine.attach_fallbacks(p, len(tiles), device="cuda")
    M, mc, st = engine.macenko_fit(to_dev(tiles), params=p)
    assert (st.cpu().numpy() == 0).all() and int(fb.sum()) == 0
    for i, I in enumerate(tiles):
        np.testing.assert_allclose(M[i].cpu().numpy(), so.macenko_stain_matrix(I), rtol=0, atol=2e-6)
