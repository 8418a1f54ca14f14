#!/opt/conda/bin/python3.9
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and the conda python3.9 that
has numpy 1.26 / scikit-image 0.18.3 / scikit-learn 0.24.2):

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

The reference cannot be imported as-is: ``import cv2`` and ``import spams``
(stainlib/utils/stain_utils.py:2-3) name wheels that exist nowhere in this image.
Two stand-in modules are therefore injected into ``sys.modules`` *before* the import:

  * ``cv2.cvtColor(I, COLOR_RGB2LAB / COLOR_LAB2RGB)``, ``cv2.split/merge/meanStdDev`` -> the OpenCV
    8-bit fixed-point restatement in oracle/stain_oracle.py (the SAME code the oracle uses: the tissue
    mask, ReinhardStainNormalizer and LuminosityStandardizer are therefore NOT independently pinned --
    "parity unpinned"; tools/pin_cv2.py closes that gap wherever a real cv2 exists);
  * ``spams.lasso(mode=2, pos=True)``   -> scikit-learn's coordinate-descent
    ``Lasso(positive=True)`` run to 1e-14 -- an implementation INDEPENDENT of the
    oracle's closed form, so the goldens do pin the oracle's lasso (images above 256 x 256 use a
    vectorised cyclic coordinate descent written here, iterated until nothing moves: also independent
    of the closed form, and fast enough for a megapixel);
  * ``spams.trainDL``                   -> not used for the goldens written here.

Everything else -- convert_RGB_to_OD, np.cov/eigh, arctan2, percentiles, rescale,
255*exp(-C@M), the truncating cast, StainAugmentor.pop, HedLighterColorAugmenter
with the real scikit-image 0.18.3 -- is the reference's own code executing.
Only arrays (inputs/outputs) are written; no reference source is copied.
"""
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import stain_oracle as so  # noqa: E402  (only for the cv2 stand-in + tile generator)

import scipy.sparse  # noqa: E402
from sklearn.linear_model import Lasso  # noqa: E402


def _install_standins():
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_RGB2LAB = 45
    cv2.COLOR_LAB2RGB = 57

    def cvtColor(I, code):
        if code == cv2.COLOR_RGB2LAB:
            return so.rgb2lab_u8(np.asarray(I)[:, :, :3])
        assert code == cv2.COLOR_LAB2RGB
        return so.lab2rgb_u8(np.asarray(I))

    cv2.split = lambda I: [np.ascontiguousarray(I[:, :, k]) for k in range(I.shape[2])]
    cv2.merge = lambda chans: np.stack(chans, axis=-1)
    cv2.meanStdDev = so.mean_std_dev
    cv2.cvtColor = cvtColor
    sys.modules["cv2"] = cv2

    spams = types.ModuleType("spams")

    def lasso(X, D, mode, lambda1, pos):
        assert mode == 2 and pos
        X, D = np.asarray(X), np.asarray(D)
        if X.shape[1] > 256 * 256:
            # cyclic coordinate descent on min_a>=0 1/2|x - D a|^2 + lam 1'a, all columns at once, to a fixed point
            G = D.T @ D
            B = D.T @ X - lambda1
            A = np.zeros((D.shape[1], X.shape[1]))
            for it in range(100000):
                prev = A.copy()
                for j in range(D.shape[1]):
                    r = B[j] - sum(G[j, k] * A[k] for k in range(D.shape[1]) if k != j)
                    A[j] = np.maximum(r / G[j, j], 0.0)
                if np.abs(A - prev).max() < 1e-15:
                    break
            return scipy.sparse.csc_matrix(A)
        # sklearn minimises 1/(2 n) ||y - Xw||^2 + alpha ||w||_1 with n = 3 rows
        n = X.shape[0]
        est = Lasso(alpha=lambda1 / n, fit_intercept=False, positive=True, tol=1e-14,
                    max_iter=1000000, precompute=False)
        est.fit(D, X)
        return scipy.sparse.csc_matrix(est.coef_.T)

    def trainDL(**kw):
        raise RuntimeError("trainDL stand-in is not available for golden generation")

    spams.lasso = lasso
    spams.trainDL = trainDL
    sys.modules["spams"] = spams


_install_standins()
sys.path.insert(0, "/root/reference")
import stainlib  # noqa: E402
from stainlib.augmentation.augmenter import HedLighterColorAugmenter, StainAugmentor  # noqa: E402
from stainlib.extraction.macenko_stain_extractor import MacenkoStainExtractor  # noqa: E402
from stainlib.normalization.normalizer import ExtractiveStainNormalizer, ReinhardStainNormalizer  # noqa: E402
from stainlib.utils import stain_utils as su  # noqa: E402
from stainlib.utils.excepts import TissueMaskException  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def macenko_case(size, seed, kind=None):
    """kind = None: the i.i.d. generator; otherwise oracle.structured_tile(kind, ...).  Tiles above 256^2 keep the SHA-256 of
    the output and every 997th pixel instead of the whole image."""
    I = so.synth_tile(size, size, seed) if kind is None else so.structured_tile(kind, size, size, seed)
    tgt = so.synth_tile(size, size, 1000 + seed, so.M_TRUE_TGT)
    rec = {"size": size, "seed": seed, "input_sha": sha(I), "target_sha": sha(tgt), "kind": "" if kind is None else kind}
    if size <= 64:
        rec["input"] = I
        rec["target"] = tgt
    # --- stages, each through the reference's own functions -----------------
    mask = su.LuminosityThresholdTissueLocator.get_tissue_mask(I)
    rec["mask_count"] = int(mask.sum())
    rec["mask_bits"] = np.packbits(mask.ravel())
    OD = su.convert_RGB_to_OD(I).reshape((-1, 3))
    rec["od_sub"] = OD[::97]
    ODt = OD[mask.ravel()]
    cov = np.cov(ODt, rowvar=False)
    _, V = np.linalg.eigh(cov)
    V = V[:, [2, 1]]
    if V[0, 0] < 0:
        V[:, 0] *= -1
    if V[0, 1] < 0:
        V[:, 1] *= -1
    That = ODt @ V
    phi = np.arctan2(That[:, 1], That[:, 0])
    rec["cov"] = cov
    rec["V"] = V
    rec["phi_pct"] = np.array([np.percentile(phi, 1), np.percentile(phi, 99)])
    M = MacenkoStainExtractor.get_stain_matrix(I)
    rec["M"] = M
    C = su.get_concentrations(I, M)
    rec["C_sub"] = np.ascontiguousarray(C[::97])
    rec["maxC"] = np.percentile(C, 99, axis=0).reshape((1, 2))
    # --- fit / transform -----------------------------------------------------
    nrm = ExtractiveStainNormalizer("macenko")
    nrm.fit(tgt)
    rec["M_target"] = nrm.stain_matrix_target
    rec["maxC_target"] = nrm.maxC_target
    out = nrm.transform(I)
    if size <= 256:
        rec["out"] = out
    else:
        rec["out_sub997"] = out.reshape(-1, 3)[::997]
    rec["out_sha"] = sha(out)
    Cs = C * (nrm.maxC_target / rec["maxC"])
    pre = 255 * np.exp(-1 * np.dot(Cs, nrm.stain_matrix_target))
    rec["prequant_sub"] = pre[::97]
    # self-transform (fit on the tile itself)
    if size <= 256:
        nrm2 = ExtractiveStainNormalizer("macenko")
        nrm2.fit(I)
        rec["out_self"] = nrm2.transform(I)
    return rec


def hed_case(size, seed, npseed):
    I = so.synth_tile(size, size, seed)
    aug = HedLighterColorAugmenter()
    rec = {"size": size, "seed": seed, "npseed": npseed, "input_sha": sha(I)}
    big = size > 256
    keep = (lambda a: a.reshape(-1, 3)[::97]) if big else (lambda a: a)
    unr = aug.transform(I)                              # sigma = beta = -0.03 (augmenter.py:194-198)
    rec["out_unrandomized"] = keep(unr)
    rec["out_unrandomized_sha"] = sha(unr)
    np.random.seed(npseed)
    aug.randomize()
    rec["sigmas"] = np.array(aug._sigmas)
    rec["biases"] = np.array(aug._biases)
    o = aug.transform(I)
    rec["out"] = keep(o)
    rec["out_sha"] = sha(o)
    rec["hed_sub"] = __import__("skimage.color").color.rgb2hed(I).reshape(-1, 3)[::97]
    white = np.full((16, 16, 3), 255, np.uint8)
    rec["white_is_same_object"] = bool(aug.transform(white) is white)
    dark = np.full((16, 16, 3), 3, np.uint8)
    rec["dark_is_same_object"] = bool(aug.transform(dark) is dark)
    f = I[:32, :32].astype(np.float64) / 255.0
    rec["out_float"] = aug.transform(f)
    return rec


def stainaug_case(size, seed, npseed, background):
    I = so.synth_tile(size, size, seed)
    aug = StainAugmentor("macenko", augment_background=background)
    aug.fit(I)
    rec = {"size": size, "seed": seed, "npseed": npseed, "background": background,
           "input_sha": sha(I), "M": aug.stain_matrix}
    np.random.seed(npseed)
    st = np.random.get_state()
    rec["out0"] = aug.pop()
    rec["out1"] = aug.pop()
    np.random.set_state(st)
    draws = [np.random.uniform(0.8, 1.2), np.random.uniform(-0.2, 0.2),
             np.random.uniform(0.8, 1.2), np.random.uniform(-0.2, 0.2)]
    rec["draws0"] = np.array(draws)                     # alpha0, beta0, alpha1, beta1
    return rec


def grayscale_case(size, seed, npseed):
    from stainlib.augmentation.augmenter import GrayscaleAugmentor
    I = so.synth_tile(size, size, seed)
    aug = GrayscaleAugmentor()
    aug.fit(I)
    rec = {"size": size, "seed": seed, "npseed": npseed, "input_sha": sha(I)}
    np.random.seed(npseed)
    st = np.random.get_state()
    rec["out0"] = aug.pop()
    rec["out1"] = aug.pop()
    np.random.set_state(st)
    rec["draws0"] = np.array([np.random.uniform(0.8, 1.2), np.random.uniform(-0.2, 0.2)])   # alpha, beta of out0
    return rec


def reinhard_case(size, seed, kind=None):
    """ReinhardStainNormalizer / LuminosityStandardizer / LAB helpers (normalizer.py:54-94, stain_utils.py:50-67,146-194)
    through the reference's own code on top of the cv2 stand-in."""
    I = so.synth_tile(size, size, seed) if kind is None else so.structured_tile(kind, size, size, seed)
    tgt = so.synth_tile(size, size, 1000 + seed, so.M_TRUE_TGT)
    rec = {"size": size, "seed": seed, "kind": "" if kind is None else kind, "input_sha": sha(I), "target_sha": sha(tgt)}
    rec["standardized"] = su.standardize_brightness(I)
    I1, I2, I3 = su.lab_split(I)
    rec["lab_split_sub"] = np.stack([I1, I2, I3], axis=-1).reshape(-1, 3)[::97]
    means, stds = su.get_mean_std(I)
    rec["means"] = np.array([float(m) for m in means])
    rec["stds"] = np.array([float(v) for v in stds])
    rec["merge_back"] = su.merge_back(*su.lab_split(I))                      # the Lab round trip of the helpers
    nrm = ReinhardStainNormalizer()
    nrm.fit(tgt)
    rec["target_means"] = np.array([float(m) for m in nrm.target_means])
    rec["target_stds"] = np.array([float(v) for v in nrm.target_stds])
    rec["out"] = nrm.transform(I)
    rec["out_masked"] = nrm.transform(I, mask_background=True)
    rec["out_masked_06"] = nrm.transform(I, mask_background=True, luminosity_threshold=0.6)
    rec["lum_std"] = stainlib.LuminosityStandardizer.standardize(I)
    rec["lum_std_80"] = stainlib.LuminosityStandardizer.standardize(I, percentile=80)
    OD = np.random.RandomState(seed).uniform(0.0, 3.0, size=(32, 32, 3))
    rec["od_to_rgb"] = su.convert_OD_to_RGB(OD)
    return rec


def tissue_case(I, seed):
    """A REAL stained-tissue image (scikit-image's immunohistochemistry() sample, `ihc.png`: "no known copyright restrictions")
    through every class of the reference a user would put it through -- the notebook's usage (stainlib_augmentation.ipynb cells
    4-15).  The input array itself is stored: the GPU box has no scikit-image."""
    I = np.ascontiguousarray(I)
    h, w = I.shape[:2]
    tgt = so.synth_tile(128, 128, 1000 + seed, so.M_TRUE_TGT)
    rec = {"input": I, "seed": seed, "input_sha": sha(I), "target_sha": sha(tgt)}

    def keep(name, a):                                    # whole images: SHA-256 + every 13th pixel (the oracle reproduces them in full)
        rec[name + "_sha"] = sha(a)
        rec[name + "_sub13"] = np.ascontiguousarray(a.reshape(-1, 3)[::13])
    # --- Macenko stages + fit / transform (macenko_stain_extractor.py:16-44, normalizer.py:27-50)
    mask = su.LuminosityThresholdTissueLocator.get_tissue_mask(I)
    rec["mask_count"] = int(mask.sum())
    rec["mask_bits"] = np.packbits(mask.ravel())
    M = MacenkoStainExtractor.get_stain_matrix(I)
    rec["M"] = M
    C = su.get_concentrations(I, M)
    rec["C_sub"] = np.ascontiguousarray(C[::97])
    rec["maxC"] = np.percentile(C, 99, axis=0).reshape((1, 2))
    nrm = ExtractiveStainNormalizer("macenko")
    nrm.fit(tgt)
    rec["M_target"] = nrm.stain_matrix_target
    rec["maxC_target"] = nrm.maxC_target
    rec["out"] = nrm.transform(I)
    Cs = C * (nrm.maxC_target / rec["maxC"])
    rec["prequant_sub"] = (255 * np.exp(-1 * np.dot(Cs, nrm.stain_matrix_target)))[::97]
    nrm2 = ExtractiveStainNormalizer("macenko")
    nrm2.fit(I)                                           # the tissue as the TARGET, a synthetic tile as the source
    src = so.synth_tile(128, 128, seed)
    rec["out_as_target"] = nrm2.transform(src)                # (128 x 128: kept whole)
    # --- StainAugmentor (augmenter.py:334-449)
    for bg in (False, True):
        aug = StainAugmentor("macenko", augment_background=bg)
        aug.fit(I)
        np.random.seed(7 + seed)
        st = np.random.get_state()
        keep("aug_out0_bg%d" % bg, aug.pop())
        keep("aug_out1_bg%d" % bg, aug.pop())
        np.random.set_state(st)
        rec["aug_draws0"] = np.array([np.random.uniform(0.8, 1.2), np.random.uniform(-0.2, 0.2),
                                      np.random.uniform(0.8, 1.2), np.random.uniform(-0.2, 0.2)])
    # --- HedLighterColorAugmenter with the real scikit-image 0.18.3 (augmenter.py:276-331)
    hed = HedLighterColorAugmenter()
    keep("hed_out_unrandomized", hed.transform(I))
    np.random.seed(5 + seed)
    hed.randomize()
    rec["hed_sigmas"] = np.array(hed._sigmas)
    rec["hed_biases"] = np.array(hed._biases)
    keep("hed_out", hed.transform(I))
    # --- Reinhard, LuminosityStandardizer, GrayscaleAugmentor (normalizer.py:54-94, stain_utils.py:146-194, augmenter.py:20-60)
    #     (on top of the cv2 stand-in: see the module docstring)
    rn = ReinhardStainNormalizer()
    rn.fit(tgt)
    keep("reinhard_out", rn.transform(I))
    keep("reinhard_out_masked", rn.transform(I, mask_background=True))
    rn2 = ReinhardStainNormalizer()
    rn2.fit(I)
    rec["reinhard_means"] = np.array([float(m) for m in rn2.target_means])
    rec["reinhard_stds"] = np.array([float(v) for v in rn2.target_stds])
    keep("lum_std", stainlib.LuminosityStandardizer.standardize(I))
    from stainlib.augmentation.augmenter import GrayscaleAugmentor
    ga = GrayscaleAugmentor()
    ga.fit(I)
    np.random.seed(11 + seed)
    keep("gray_out0", ga.pop())
    return rec


def errors_case():
    rec = {}
    try:
        su.LuminosityThresholdTissueLocator.get_tissue_mask(np.full((8, 8, 3), 255, np.uint8))
        rec["white_raises"] = False
    except TissueMaskException as e:
        rec["white_raises"] = True
        rec["white_msg"] = str(e)
    try:
        ExtractiveStainNormalizer("reinhard")
        rec["bad_method_raises"] = False
    except Exception as e:  # noqa: BLE001
        rec["bad_method_raises"] = True
        rec["bad_method_msg"] = str(e)
    try:
        MacenkoStainExtractor.get_stain_matrix(np.zeros((8, 8, 3), np.float32))
        rec["float_raises"] = False
    except AssertionError as e:
        rec["float_raises"] = True
        rec["float_msg"] = str(e)
    rgba = np.full((8, 8, 4), 120, np.uint8)
    rec["rgba_passes_guard"] = bool(su.is_uint8_image(rgba))                  # stain_utils.py:126-144: channels unchecked
    try:
        MacenkoStainExtractor.get_stain_matrix(rgba)
        rec["rgba_fails_later"] = False
    except Exception as e:  # noqa: BLE001
        rec["rgba_fails_later"] = True
        rec["rgba_exception"] = type(e).__name__
    try:
        su.convert_OD_to_RGB(np.full((2, 2, 3), -0.5))
        rec["neg_od_raises"] = False
    except AssertionError as e:
        rec["neg_od_raises"] = True
        rec["neg_od_msg"] = str(e)
    return rec


def save(name, rec):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    print("reference:", stainlib.__file__)
    for size in (64, 256):
        for seed in (1, 2, 3):
            save("macenko_%d_s%d" % (size, seed), macenko_case(size, seed))
    save("macenko_1024_s1", macenko_case(1024, 1))                        # BASELINE configs[1] tile size
    for kind in ("white_bg", "palette12", "quantized"):
        save("macenko_128_%s_s4" % kind, macenko_case(128, 4, kind))
    save("macenko_256_blobs_s5", macenko_case(256, 5, "blobs"))           # smooth spatial structure (nuclei, gradients, lumen)
    save("hed_512_s4_np5", hed_case(512, 4, 5))                           # BASELINE configs[3] tile size
    save("hed_128_s2_np0", hed_case(128, 2, 0))
    save("hed_128_s3_np123", hed_case(128, 3, 123))
    save("stainaug_128_s2_np7", stainaug_case(128, 2, 7, False))
    save("stainaug_128_s3_np7_bg", stainaug_case(128, 3, 7, True))
    save("grayscale_128_s2_np11", grayscale_case(128, 2, 11))
    save("reinhard_128_s2", reinhard_case(128, 2))
    save("reinhard_128_white_bg_s4", reinhard_case(128, 4, "white_bg"))
    save("errors", errors_case())
    tissue_main()


def tissue_main():
    """Real stained tissue: the 512 x 512 sample image of scikit-image and an odd crop of it."""
    from skimage import data
    ihc = np.ascontiguousarray(data.immunohistochemistry()[:, :, :3])
    save("tissue_ihc_512", tissue_case(ihc, 1))
    save("tissue_ihc_crop_383x509", tissue_case(ihc[3:386, 2:511], 2))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tissue":
        tissue_main()
    else:
        main()
