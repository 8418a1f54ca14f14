"""The oracle (oracle/stain_oracle.py) against the golden vectors captured from the reference.

CPU-only.  This is what pins the oracle: every array below was produced by the reference's
own code (tests/golden/make_golden.py); the oracle must reproduce it."""
import glob
import hashlib
import os

import numpy as np
import pytest

from oracle import stain_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


MACENKO = sorted(glob.glob(os.path.join(GOLDEN, "macenko_*.npz")))


@pytest.mark.parametrize("path", MACENKO, ids=[os.path.basename(p)[:-4] for p in MACENKO])
def test_macenko_stages(path):
    g = np.load(path)
    size, seed = int(g["size"]), int(g["seed"])
    I = so.synth_tile(size, size, seed)
    tgt = so.synth_tile(size, size, 1000 + seed, so.M_TRUE_TGT)
    assert sha(I) == str(g["input_sha"]) and sha(tgt) == str(g["target_sha"])
    if "input" in g.files:
        assert np.array_equal(I, g["input"]) and np.array_equal(tgt, g["target"])
    mask = so.tissue_mask(I)
    assert int(mask.sum()) == int(g["mask_count"])
    assert np.array_equal(np.packbits(mask.ravel()), g["mask_bits"])
    # np.log differs by <=1 ulp between the numpy that wrote the goldens (1.26) and this one
    np.testing.assert_allclose(so.rgb_to_od(I).reshape(-1, 3)[::97], g["od_sub"], rtol=4e-16, atol=0)
    d = {}
    M = so.macenko_stain_matrix(I, details=d)
    np.testing.assert_allclose(d["cov"], g["cov"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(d["V"], g["V"], rtol=0, atol=1e-12)
    np.testing.assert_allclose([d["minPhi"], d["maxPhi"]], g["phi_pct"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(M, g["M"], rtol=0, atol=1e-13)
    # lasso: oracle closed form vs the reference run with an independent CD solver
    C = so.get_concentrations(I, g["M"])
    np.testing.assert_allclose(C[::97], g["C_sub"], rtol=0, atol=1e-10)
    assert so.lasso_kkt_violation(so.rgb_to_od(I).reshape(-1, 3), g["M"], C, 0.01) < 1e-12
    np.testing.assert_allclose(np.percentile(C, 99, axis=0).reshape(1, 2), g["maxC"], rtol=1e-9)


@pytest.mark.parametrize("path", MACENKO, ids=[os.path.basename(p)[:-4] for p in MACENKO])
def test_macenko_fit_transform(path):
    g = np.load(path)
    size, seed = int(g["size"]), int(g["seed"])
    I = so.synth_tile(size, size, seed)
    tgt = so.synth_tile(size, size, 1000 + seed, so.M_TRUE_TGT)
    n = so.ExtractiveStainNormalizer("macenko")
    n.fit(tgt)
    np.testing.assert_allclose(n.stain_matrix_target, g["M_target"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(n.maxC_target, g["maxC_target"], rtol=1e-9)
    d = {}
    out = n.transform(I, details=d)
    assert out.dtype == np.uint8 and out.shape == I.shape
    np.testing.assert_allclose(d["prequant"].reshape(-1, 3)[::97], g["prequant_sub"], rtol=1e-8)
    # the lasso stand-in used for the goldens converges to ~1e-12, so a byte can flip only
    # if a pre-quantisation value sits within 1e-9 of an integer: demand bit-identity.
    assert np.array_equal(out, g["out"])
    n2 = so.ExtractiveStainNormalizer("macenko")
    n2.fit(I)
    assert np.array_equal(n2.transform(I), g["out_self"])


HED = sorted(glob.glob(os.path.join(GOLDEN, "hed_*.npz")))


@pytest.mark.parametrize("path", HED, ids=[os.path.basename(p)[:-4] for p in HED])
def test_hed(path):
    g = np.load(path)
    size, seed, npseed = int(g["size"]), int(g["seed"]), int(g["npseed"])
    I = so.synth_tile(size, size, seed)
    assert sha(I) == str(g["input_sha"])
    np.testing.assert_allclose(so.rgb2hed(I).reshape(-1, 3)[::97], g["hed_sub"], rtol=0, atol=1e-14)
    t = 0.03
    assert np.array_equal(so.hed_transform(I, [-t] * 3, [-t] * 3), g["out_unrandomized"])
    np.random.seed(npseed)
    s, b = so.hed_randomize(t)
    np.testing.assert_array_equal(s, g["sigmas"])
    np.testing.assert_array_equal(b, g["biases"])
    assert np.array_equal(so.hed_transform(I, s, b), g["out"])
    white = np.full((16, 16, 3), 255, np.uint8)
    assert bool(g["white_is_same_object"]) and so.hed_transform(white, s, b) is white
    dark = np.full((16, 16, 3), 3, np.uint8)
    assert bool(g["dark_is_same_object"]) and so.hed_transform(dark, s, b) is dark
    f = I[:32, :32].astype(np.float64) / 255.0
    np.testing.assert_allclose(so.hed_transform(f, s, b), g["out_float"], rtol=0, atol=1e-13)


SA = sorted(glob.glob(os.path.join(GOLDEN, "stainaug_*.npz")))


@pytest.mark.parametrize("path", SA, ids=[os.path.basename(p)[:-4] for p in SA])
def test_stain_augmentor(path):
    g = np.load(path)
    size, seed, npseed = int(g["size"]), int(g["seed"]), int(g["npseed"])
    I = so.synth_tile(size, size, seed)
    assert sha(I) == str(g["input_sha"])
    a = so.StainAugmentor("macenko", augment_background=bool(g["background"]))
    a.fit(I)
    np.testing.assert_allclose(a.stain_matrix, g["M"], rtol=0, atol=1e-13)
    np.random.seed(npseed)
    assert np.array_equal(a.pop(), g["out0"])
    assert np.array_equal(a.pop(), g["out1"])
    d = g["draws0"]
    assert np.array_equal(a.pop_with([d[0], d[2]], [d[1], d[3]]), g["out0"])


def test_error_contract():
    g = np.load(os.path.join(GOLDEN, "errors.npz"))
    assert bool(g["white_raises"]) and str(g["white_msg"]) == "Empty tissue mask computed"
    with pytest.raises(so.TissueMaskException, match="Empty tissue mask computed"):
        so.tissue_mask(np.full((8, 8, 3), 255, np.uint8))
    assert str(g["bad_method_msg"]) == "Method not recognized."
    with pytest.raises(Exception, match="Method not recognized."):
        so.ExtractiveStainNormalizer("reinhard")
    assert str(g["float_msg"]) == "Image should be RGB uint8."
    with pytest.raises(AssertionError, match="Image should be RGB uint8."):
        so.macenko_stain_matrix(np.zeros((8, 8, 3), np.float32))


def test_lab_threshold_index():
    # default threshold: L8 <= 203  <=>  Y-table index <= 1146
    assert so.y_index_threshold(0.8) == 1146
    rng = np.random.RandomState(0)
    I = rng.randint(0, 256, size=(64, 64, 3)).astype(np.uint8)
    assert np.array_equal(so.lab_l8(I) / 255.0 < 0.8, so.lab_y_index(I) <= 1146)
    assert so.lab_l8(np.full((1, 1, 3), 255, np.uint8))[0, 0] == 255
    assert so.lab_l8(np.zeros((1, 1, 3), np.uint8))[0, 0] == 0


def test_grayscale_augmentor_golden():
    g = np.load(os.path.join(GOLDEN, "grayscale_128_s2_np11.npz"))
    I = so.synth_tile(128, 128, int(g["seed"]))
    a = so.GrayscaleAugmentor()
    a.fit(I)
    np.random.seed(int(g["npseed"]))
    assert np.array_equal(a.pop(), g["out0"]) and np.array_equal(a.pop(), g["out1"])
    assert np.array_equal(a.pop_with(*g["draws0"]), g["out0"])

