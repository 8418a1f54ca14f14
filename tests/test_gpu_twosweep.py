"""-m gpu: the two-read-sweep schedule of the fused Macenko kernel (stats_twosweep.hpp; SlParams.two_sweep).  The candidates of all four
order statistics are collected in the moments sweep under eigenvectors estimated from a cluster sample, and the finish verifies the
estimate against the exact eigenvectors: results must be the three-sweep schedule's bit for bit whatever happens to the attempt --
accepted, declined in phase 0, or refused by the finish (forced here) -- and the oracle's / the reference's within the usual bars."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import stain_oracle as so
from tests.gpu_util import oracle_fit_tile, to_dev, u8_parity

pytestmark = pytest.mark.gpu

M_ATOL = 5e-7
MAXC_RTOL = 5e-7
DIRECT, OFF, NO_ESTIMATE, SHARE, PLANE, BRACKET, LISTS = 1, 0, -1, -2, -3, -4, -5


def _run(dev, Mt, mct, mode, **kw):
    from stainlib_amd import engine
    n = dev.shape[0]
    p = engine.make_params(schedule=2, two_sweep=mode, **kw)
    fb = engine.attach_fallbacks(p, n, device="cuda")
    rs = torch.zeros((n,), dtype=torch.int32, device="cuda")
    ts = torch.full((n,), 99, dtype=torch.int32, device="cuda")
    p.resweeps_out, p.twosweep_out = rs.data_ptr(), ts.data_ptr()
    out, M, mc, st = engine.macenko_transform(dev, Mt, mct, params=p)
    torch.cuda.synchronize()
    return dict(out=out, M=M, mc=mc, st=st, fb=fb, rs=rs, ts=ts.cpu().numpy())


def _same(a, b, label):
    assert torch.equal(a["out"], b["out"]), label + ": bytes"
    assert torch.equal(a["st"], b["st"]), label + ": status"
    for k in ("M", "mc"):
        assert torch.equal(torch.nan_to_num(a[k], nan=-7.0), torch.nan_to_num(b[k], nan=-7.0)), label + ": " + k


def _batch(h, w, n):
    tiles = [so.synth_tile(h, w, 300 + s) for s in range(n)]
    tiles[1] = so.structured_tile("white_bg", h, w, 5)
    tiles[2] = so.structured_tile("quantized", h, w, 6)
    tiles[3] = so.structured_tile("blobs", h, w, 7)
    tiles[4] = np.full((h, w, 3), 255, np.uint8)                                  # empty mask
    tiles[5] = so.structured_tile("palette12", h, w, 8)                           # heavy ties
    return tiles


# (503, 527): a pixel count that is not a multiple of four (the byte-wise tile accesses, a ragged last chunk) on a streaming-size tile;
# (128, 130): the smallest tiles that try at all are 16 Ki pixels; (96, 96) is below: never attempted
@pytest.mark.parametrize("h,w", [(128, 130), (256, 320), (503, 527), (1024, 1024), (96, 96)])
def test_every_two_sweep_mode_gives_the_three_sweep_results(h, w):
    n = 8 if h * w >= (1 << 19) else 12
    tiles = _batch(h, w, n)
    dev = to_dev(tiles)
    Mt, mct = oracle_fit_tile(so.synth_tile(128, 128, 1001, so.M_TRUE_TGT))
    ref = _run(dev, Mt, mct, 1)
    assert (ref["ts"] == OFF).all()
    st = ref["st"].cpu().numpy()
    assert st[4] == 1 and (st[[0, 1, 2, 3]] == 0).all()
    for mode in (0, 2, 3, 4):
        r = _run(dev, Mt, mct, mode)
        _same(ref, r, f"{h}x{w} two_sweep={mode}")
        print(f"{h}x{w} two_sweep={mode}: attempts {r['ts'].tolist()} resweeps {r['rs'].cpu().tolist()} fallbacks {r['fb'].cpu().tolist()}")
        if h * w < (1 << 14):
            assert (r["ts"] == OFF).all()
            continue
        if mode == 2:
            assert (r["ts"][[0, 1, 2]] == DIRECT).all(), r["ts"]                     # i.i.d., white background, quantised: the direct route
            assert r["ts"][4] in (NO_ESTIMATE, OFF)                                   # no tissue: no estimate
        if mode == 3:
            assert (r["ts"][st == 0] != DIRECT).all() and (r["ts"][[0, 1, 2]] == PLANE).all(), r["ts"]    # the forced failure drives the fallback
        if mode == 4:
            assert (r["ts"][[0, 1, 2]] != DIRECT).all(), r["ts"]                     # a plane tilted by 0.05: refused by the finish's own check
    # ... and they are the oracle's
    n_ = so.ExtractiveStainNormalizer("macenko")
    n_.stain_matrix_target, n_.maxC_target = Mt, mct.reshape(1, 2)
    r = _run(dev, Mt, mct, 2)
    out = r["out"].cpu().numpy()
    for i in (0, 1, 2, 3):
        Mo, mco = oracle_fit_tile(tiles[i])
        np.testing.assert_allclose(r["M"].cpu().numpy()[i], Mo, rtol=0, atol=M_ATOL)
        np.testing.assert_allclose(r["mc"].cpu().numpy()[i], mco, rtol=MAXC_RTOL)
        u8_parity(out[i], n_.transform(tiles[i]), label=f"two-sweep forced {h}x{w} tile {i}", src=tiles[i])


GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "macenko_*.npz")))


@pytest.mark.parametrize("path", [p for p in GOLD if int(np.load(p)["size"]) >= 128], ids=lambda p: os.path.basename(p)[:-4])
def test_forced_two_sweep_against_the_reference_goldens(path):
    """fit(target) + transform(I) with the two-sweep schedule forced, against what the reference itself produced."""
    from stainlib_amd import engine
    g = np.load(path)
    size, seed, kind = int(g["size"]), int(g["seed"]), str(g["kind"])
    I = so.synth_tile(size, size, seed) if not kind else so.structured_tile(kind, size, size, seed)
    tgt = so.synth_tile(size, size, 1000 + seed, so.M_TRUE_TGT)
    p = engine.make_params(schedule=2, two_sweep=2)
    ts = torch.full((1,), 99, dtype=torch.int32, device="cuda")
    p.twosweep_out = ts.data_ptr()
    Mt, mct, st = engine.macenko_fit(to_dev([tgt]), params=p)
    np.testing.assert_allclose(Mt.cpu().numpy()[0], g["M_target"], rtol=0, atol=M_ATOL)
    np.testing.assert_allclose(mct.cpu().numpy()[0], g["maxC_target"].reshape(2), rtol=MAXC_RTOL)
    out, M, mc, st = engine.macenko_transform(to_dev([I]), Mt[0], mct[0], params=p)
    assert int(st[0]) == 0
    print(os.path.basename(path), "two-sweep attempt:", int(ts[0]))
    np.testing.assert_allclose(M.cpu().numpy()[0], g["M"], rtol=0, atol=M_ATOL)
    np.testing.assert_allclose(mc.cpu().numpy()[0], g["maxC"].reshape(2), rtol=MAXC_RTOL)
    got = out.cpu().numpy()[0]
    if "out" in g.files:
        u8_parity(got, g["out"], label=os.path.basename(path), src=I)
    else:
        u8_parity(got.reshape(-1, 3)[::997], g["out_sub997"], label=os.path.basename(path) + " (1/997 of the reference output)")
    o3, _, _, _ = engine.macenko_transform(to_dev([I]), Mt[0], mct[0], params=engine.make_params(schedule=2, two_sweep=1))
    assert torch.equal(o3, out)


def test_real_tissue_and_a_full_grid_in_the_automatic_mode():
    """A batch that fills the resident grid (automatic mode: every workgroup tries, one whose tile declined backs off for three tiles):
    i.i.d. tiles and the real-tissue fixture mirror-tiled; the attempts are reported, nothing fails, and the bytes are the
    three-sweep schedule's."""
    from stainlib_amd import engine
    from tools.synth import synth_tiles
    props = torch.cuda.get_device_properties(0)
    n = 2 * props.multi_processor_count
    rgb = synth_tiles(n, 512, 512, seed=11)
    ihc = np.load(os.path.join(os.path.dirname(__file__), "golden", "tissue_ihc_512.npz"))["input"]
    rgb[7] = torch.from_numpy(np.ascontiguousarray(ihc)).cuda()
    rgb[n - 3] = torch.from_numpy(np.ascontiguousarray(ihc[::-1])).cuda()
    Mt, mct = oracle_fit_tile(so.synth_tile(128, 128, 1001, so.M_TRUE_TGT))
    ref = _run(rgb, Mt, mct, 1)
    auto = _run(rgb, Mt, mct, 0)
    _same(ref, auto, "automatic, full grid")
    codes, counts = np.unique(auto["ts"], return_counts=True)
    print("automatic mode, attempts:", dict(zip(codes.tolist(), counts.tolist())))
    assert set(codes.tolist()) <= {DIRECT, OFF, NO_ESTIMATE, SHARE, PLANE, BRACKET, LISTS}
    assert (auto["ts"] == DIRECT).sum() >= n // 4                                   # about half the grid tries; i.i.d. tiles are accepted
    forced = _run(rgb, Mt, mct, 2)
    _same(ref, forced, "forced, full grid")
    assert (forced["ts"] == DIRECT).sum() >= n - 8


def test_a_workgroup_backs_off_after_a_declined_attempt_and_results_do_not_change():
    """Automatic mode on a batch of several rounds whose tiles decline the route in phase 0 (crops of the real-tissue fixture: its third
    eigenvalue puts the tilt bound above the limit): a workgroup that saw a decline skips the attempt on its next tiles (kTsBackoff), so
    twosweep_out holds both "no estimate" and "not attempted" -- and bytes, statistics and status equal the three-sweep run's."""
    I = np.load(os.path.join(os.path.dirname(__file__), "golden", "tissue_ihc_512.npz"))["input"]
    rng = np.random.RandomState(4242)
    h, w = 160, 176
    n = 1600                                   # more than three rounds of the 512 resident workgroups
    tiles = []
    for _ in range(n):
        y, x = int(rng.randint(0, 512 - h)), int(rng.randint(0, 512 - w))
        tiles.append(np.ascontiguousarray(I[y:y + h, x:x + w]))
    dev = to_dev(tiles)
    Mt, mct = oracle_fit_tile(so.synth_tile(128, 128, 1001, so.M_TRUE_TGT))
    ref = _run(dev, Mt, mct, 1)
    got = _run(dev, Mt, mct, 0)
    _same(got, ref, "automatic mode with back-off")
    codes = {int(k): int((got["ts"] == k).sum()) for k in np.unique(got["ts"])}
    print("two-sweep attempts over", n, "tiles:", codes)
    assert codes.get(OFF, 0) > 0 and any(k < 0 for k in codes), codes      # some declined, some were skipped after a decline
    assert (ref["ts"] == OFF).all()
