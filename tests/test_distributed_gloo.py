"""CPU, world_size = 2, gloo: the N>1 host path (sharding, stats gather with uneven shards, slide statistics).
No GPU compute is involved: per-tile statistics are synthetic tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stainlib_amd import distributed as sd


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 100000):
        for world in (1, 2, 3, 8):
            spans = [sd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sd.shard_range(4, 2, 2)


def _table(n, seed=0):
    rng = np.random.RandomState(seed)
    M = rng.rand(n, 2, 3) + 0.1
    M /= np.linalg.norm(M, axis=2, keepdims=True)
    maxC = rng.rand(n, 2) + 1.0
    status = (rng.rand(n) < 0.2).astype(np.int32)
    status[0] = 0
    return torch.from_numpy(M), torch.from_numpy(maxC), torch.from_numpy(status)


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    M, maxC, status = _table(n)
    lo, hi = sd.shard_range(n, rank, world)
    Ma, ca, sa = sd.gather_tile_stats(M[lo:hi], maxC[lo:hi], status[lo:hi])
    Ms, cs = sd.slide_statistics(Ma, ca, sa)
    q.put((rank, Ma.numpy(), ca.numpy(), sa.numpy(), Ms.numpy(), cs.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n", [7, 10])           # 7: uneven shards (3 + 4)
def test_gather_and_slide_statistics_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    M, maxC, status = _table(n)
    Ms1, cs1 = sd.slide_statistics(M, maxC, status)          # the single-process answer
    for rank, Ma, ca, sa, Ms, cs in res:
        assert np.array_equal(Ma, M.numpy()) and np.array_equal(ca, maxC.numpy()) and np.array_equal(sa, status.numpy())
        assert np.array_equal(Ms, Ms1.numpy()) and np.array_equal(cs, cs1.numpy())   # identical on every rank
    np.testing.assert_allclose(np.linalg.norm(res[0][4], axis=1), 1.0, atol=1e-12)


def test_single_process_passthrough_and_errors():
    M, maxC, status = _table(5)
    Ma, ca, sa = sd.gather_tile_stats(M, maxC, status)
    assert torch.equal(Ma, M) and torch.equal(ca, maxC) and torch.equal(sa, status)
    with pytest.raises(ValueError):
        sd.slide_statistics(M, maxC, torch.ones(5, dtype=torch.int32))


# ---- pooled slide-level mode: the distributed exact order statistic (radix select over all-reduced histograms) ----
def _f2ord(a):
    u = np.asarray(a, np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)


def _keys(rank, world, n=5000, seed=3):
    rng = np.random.RandomState(seed)
    run = np.float32(2.0) + np.arange(40, dtype=np.float32) * np.spacing(np.float32(2.0))       # 40 consecutive binary32 values
    allk = np.concatenate([rng.randn(n).astype(np.float32), np.zeros(400, np.float32), np.float32([1.5] * 7), run])  # ties + zeros
    rng.shuffle(allk)
    lo, hi = sd.shard_range(len(allk), rank, world)
    return allk, allk[lo:hi]


def _rank_pairs_with_numpy_histograms(mine, ks, wide=False):
    """Stand-in for the device kernels: histograms / next-above of THIS rank's keys with numpy; target 0 = the keys,
    target 1 = their negation (so the two targets walk different prefixes in lockstep)."""
    o = [_f2ord(mine).astype(np.uint64), _f2ord(-mine).astype(np.uint64)]

    def hist_fn(prefixes, bits):
        rows = []
        for t in range(2):
            sel = o[t] if bits == 0 else o[t][(o[t] >> np.uint64(32 - bits)) == np.uint64(prefixes[t])]
            b = (sel >> np.uint64(24 - bits)) & np.uint64(255)
            rows.append(np.bincount(b.astype(np.int64), minlength=256).astype(np.int64))
        return torch.from_numpy(np.stack(rows))

    def next_above_fn(keys):
        out = []
        for t in range(2):
            g = o[t][o[t] > np.uint64(keys[t])]
            out.append(int(g.min()) if len(g) else 0xffffffff)
        return out
    def hist16_fn(prefixes):                       # the last two rounds in one: low 16 bits under a 16-bit prefix
        rows = []
        for t in range(2):
            sel = o[t][(o[t] >> np.uint64(16)) == np.uint64(prefixes[t])]
            rows.append(np.bincount((sel & np.uint64(0xffff)).astype(np.int64), minlength=65536).astype(np.int64))
        return torch.from_numpy(np.stack(rows))
    return sd.exact_rank_pairs(hist_fn, next_above_fn, ks, hist16_fn=hist16_fn if wide else None)


def _worker_rank_pair(rank, world, port, ks, q, wide=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, mine = _keys(rank, world)
    q.put((rank, [_rank_pairs_with_numpy_histograms(mine, (k, k2), wide) for k, k2 in ks]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wide", [False, True])      # True: 8 + 8 + 16 bits (three rounds), False: four 8-bit rounds
def test_exact_rank_pairs_world2_match_sorted_union(wide):
    ks = [(0, 5406), (53, 2699), (2500, 2500), (5399, 17), (10 ** 9, 0)]   # incl. inside the block of ties and beyond the end
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rank_pair, args=(r, 2, port, ks, q, wide)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    allk, _ = _keys(0, 1)
    srt = [np.sort(allk), np.sort(-allk)]
    assert res[0][1] == res[1][1]                            # identical on every rank
    for (k0, k1), pair in zip(ks, res[0][1]):
        for t, k in enumerate((k0, k1)):
            a, b, total = pair[t]
            kk = min(k, len(srt[t]) - 1)
            assert total == len(srt[t])
            assert sd.ord_to_float(a) == srt[t][kk] and sd.ord_to_float(b) == srt[t][min(kk + 1, len(srt[t]) - 1)]


def test_exact_rank_pairs_successor_inside_the_last_window():
    """Ranks inside a run of consecutive binary32 values: the k+1-th key comes from the last histogram round (no
    next_above sweep); checked against the sorted keys (single process: the rounds are the same without a group)."""
    allk, _ = _keys(0, 1)
    srt = [np.sort(allk), np.sort(-allk)]
    k2 = int(np.searchsorted(srt[0], np.float32(2.0)))
    calls = []
    o = [_f2ord(allk).astype(np.uint64), _f2ord(-allk).astype(np.uint64)]

    def hist_fn(prefixes, bits):
        rows = []
        for t in range(2):
            sel = o[t] if bits == 0 else o[t][(o[t] >> np.uint64(32 - bits)) == np.uint64(prefixes[t])]
            rows.append(np.bincount(((sel >> np.uint64(24 - bits)) & np.uint64(255)).astype(np.int64), minlength=256).astype(np.int64))
        return torch.from_numpy(np.stack(rows))

    def next_above_fn(keys):
        calls.append(tuple(keys))
        return [int(o[t][o[t] > np.uint64(keys[t])].min()) if (o[t] > np.uint64(keys[t])).any() else 0xffffffff for t in range(2)]
    n = len(allk)
    for k in (k2 + 3, k2 + 17, k2 + 38):
        kneg = n - 1 - (k + 1)                       # the same pair of values seen from the negated keys
        res = sd.exact_rank_pairs(hist_fn, next_above_fn, (k, kneg))
        for t, kk in enumerate((k, kneg)):
            a, b, total = res[t]
            assert total == n and sd.ord_to_float(a) == srt[t][kk] and sd.ord_to_float(b) == srt[t][kk + 1]
    assert calls == []                               # every successor sat inside the 256-key window


def _window_fns(mine):
    """numpy stand-ins for sl_slide_key_histogram_sampled (every 4th key as the sample) and sl_slide_key_window."""
    o = [_f2ord(mine).astype(np.uint64), _f2ord(-mine).astype(np.uint64)]
    smp = [x[::4] for x in o]

    def sample_hist_fn(prefixes, bits):
        rows = []
        for t in range(2):
            sel = smp[t] if bits == 0 else smp[t][(smp[t] >> np.uint64(32 - bits)) == np.uint64(prefixes[t])]
            rows.append(np.bincount(((sel >> np.uint64(24 - bits)) & np.uint64(255)).astype(np.int64), minlength=256).astype(np.int64))
        return torch.from_numpy(np.stack(rows))

    def window_fn(lo):
        out = np.zeros(2 * 65536 + 2, np.int64)
        for t in range(2):
            d = o[t].astype(np.int64) - int(lo[t])
            out[2 * 65536 + t] = int((d < 0).sum())
            out[t * 65536:(t + 1) * 65536] = np.bincount(d[(d >= 0) & (d < 65536)], minlength=65536)
        return torch.from_numpy(out)
    return sample_hist_fn, window_fn


def _worker_window(rank, world, port, ks, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, mine = _keys(rank, world)
    sf, wf = _window_fns(mine)
    q.put((rank, [sd.window_rank_pairs(sf, wf, kk, (n_total, n_total)) for kk in ks]))
    dist.barrier()
    dist.destroy_process_group()


def test_window_rank_pairs_world2():
    """One-sweep selection: exact when the window (centred on the sample estimate) holds ranks k and k+1, None otherwise
    -- identically on every rank.  The keys are N(0,1): 65536 consecutive binary32 values span ~0.8 % of a value, so
    central ranks of 5447 keys miss the window (None -> the caller's radix rounds) while the block of 400 zeros
    and the run of consecutive values around 2.0 are hit."""
    allk, _ = _keys(0, 1)
    n = len(allk)
    srt = [np.sort(allk), np.sort(-allk)]
    kz = int(np.searchsorted(srt[0], np.float32(0.0))) + 100          # inside the zeros (ties)
    k2 = int(np.searchsorted(srt[0], np.float32(2.0))) + 5            # inside the run of consecutive values
    ks = [(kz, n - 1 - kz - 1), (k2, n - 1 - k2 - 1), (10, 10)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_window, args=(r, 2, port, ks, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]
    hits = 0
    for kk, pair in zip(ks, res[0][1]):
        if pair is None:
            continue
        hits += 1
        for t, k in enumerate(kk):
            a, b, total = pair[t]
            assert total == n and sd.ord_to_float(a) == srt[t][k] and sd.ord_to_float(b) == srt[t][min(k + 1, n - 1)]
    assert hits >= 2 and res[0][1][2] is None or hits == 3


def test_percentile_position_and_lerp_follow_numpy():
    rng = np.random.RandomState(0)
    x = np.sort(rng.rand(1001))
    for pct in (1.0, 99.0, 50.0, 0.0, 100.0):
        k, g = sd.percentile_position(len(x), pct)
        assert sd.np_lerp(x[k], x[min(k + 1, len(x) - 1)], g) == np.percentile(x, pct)


# ---- the REAL PooledSlideStatistics.__call__ / SlideNormalizer.transform_shard on two gloo ranks ------------------------
# The engine's device sweeps (sl_tile_moments, sl_slide_key_*, sl_normalize_apply) are replaced by numpy stand-ins built
# on the oracle, with the same contracts as include/stainlib_hip.h; everything else -- the moment / pixel-count
# all-reduces, the sampled estimate, the window all-reduce, the radix fallback, the broadcast-free agreement of the ranks,
# the apply pass with the slide statistics -- is the product code of stainlib_amd/distributed.py executing.
def _install_numpy_engine(force_radix=False):
    from oracle import stain_oracle as so
    from stainlib_amd import _ffi, engine

    def f2ord(a):
        u = np.asarray(a, np.float32).view(np.uint32)
        return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint64)

    def keys(tiles, keyset, basis):
        """ordered-uint32 keys of this rank's pixels, per target: [k0, k1] (angle: tissue pixels only, one key set twice)"""
        T = tiles.numpy()
        od = np.concatenate([so.rgb_to_od(t).reshape(-1, 3) for t in T]).astype(np.float32)
        if keyset == _ffi.KEYSET_ANGLE:
            mask = np.concatenate([(so.lab_l8(t) / 255.0 < 0.8).ravel() for t in T])
            V = np.asarray(basis, np.float64).reshape(3, 2).astype(np.float32)
            th = od[mask] @ V
            x, y = th[:, 0], th[:, 1]
            d = np.abs(x) + np.abs(y)
            p = np.where(d > 0, y / np.where(d > 0, d, 1), 0).astype(np.float32)
            p = np.where(x < 0, np.where(y >= 0, 2.0, -2.0).astype(np.float32) - p, p).astype(np.float32)
            k = f2ord(p)
            return [k, k]
        C = so.lasso2_nonneg(od.astype(np.float64), np.asarray(basis, np.float64).reshape(2, 3), 0.01).astype(np.float32)
        return [f2ord(C[:, 0]), f2ord(C[:, 1])]

    def tile_moments(tiles, params=None, ws=None):
        rows = []
        for t in tiles.numpy():
            od = so.rgb_to_od(t).reshape(-1, 3)[(so.lab_l8(t) / 255.0 < 0.8).ravel()]
            S = od.T @ od
            rows.append([len(od), *od.sum(0), S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]])
        return torch.tensor(rows, dtype=torch.float64)

    def hist(tiles, keyset, basis, prefixes, bits, hist=None, params=None, every=1):
        ks = keys(tiles, keyset, basis)
        rows = []
        for t in range(2):
            o = ks[t][::every]
            sel = o if bits == 0 else o[(o >> np.uint64(32 - bits)) == np.uint64(prefixes[t])]
            rows.append(np.bincount(((sel >> np.uint64(24 - bits)) & np.uint64(255)).astype(np.int64), minlength=256))
        return torch.from_numpy(np.stack(rows).astype(np.int64))

    def hist16(tiles, keyset, basis, prefixes16, hist=None, params=None):
        ks = keys(tiles, keyset, basis)
        rows = [np.bincount((ks[t][(ks[t] >> np.uint64(16)) == np.uint64(prefixes16[t])] & np.uint64(0xffff)).astype(np.int64), minlength=65536)
                for t in range(2)]
        return torch.from_numpy(np.stack(rows).astype(np.int64))

    def window(tiles, keyset, basis, lo, params=None):
        if force_radix:                       # a window that sees nothing: the caller must fall back to the radix rounds
            return torch.zeros((2 * 65536 + 2,), dtype=torch.int64)
        ks = keys(tiles, keyset, basis)
        out = np.zeros(2 * 65536 + 2, np.int64)
        for t in range(2):
            d = ks[t].astype(np.int64) - int(lo[t])
            out[t * 65536:(t + 1) * 65536] = np.bincount(d[(d >= 0) & (d < 65536)], minlength=65536)
            out[2 * 65536 + t] = int((d < 0).sum())
        return torch.from_numpy(out)

    def next_above(tiles, keyset, basis, key_ords, params=None):
        ks = keys(tiles, keyset, basis)
        out = []
        for t in range(2):
            g = ks[t][ks[t] > np.uint64(key_ords[t])]
            out.append(int(g.min()) if len(g) else 0xffffffff)
        return out

    def normalize_apply(rgb, M_src, maxC_src, M_tgt, maxC_tgt, lasso_lambda=0.01, out=None, want_prequant=False):
        res = []
        for i, t in enumerate(rgb.numpy()):
            C = so.get_concentrations(t, np.asarray(M_src[i])) * (np.asarray(maxC_tgt).reshape(2) / np.asarray(maxC_src[i]))
            res.append(so.truncate_u8(255 * np.exp(-C @ np.asarray(M_tgt))).reshape(t.shape))
        return torch.from_numpy(np.stack(res))

    # ---- the device-driven steps (sl_pool_*): numpy restatements of the single-workgroup decision kernels of csrc/slide.hip on a
    # CPU float64 "state" tensor with the layout of include/stainlib_hip.h (SL_POOL_*) -- the orchestration in
    # PooledSlideStatistics.enqueue / finish is the product code
    K_T, K_NPX, K_VD, K_VF, K_K, K_G, K_TOT, K_KS, K_BELOW, K_PREFIX, K_WLO, K_RES = 10, 11, 12, 18, 24, 26, 28, 30, 32, 34, 36, 43

    def ord2f(o):
        o = int(o)
        bits = (o & 0x7fffffff) if (o & 0x80000000) else (~o & 0xffffffff)
        return float(np.array([bits], np.uint32).view(np.float32)[0])

    def pool_begin(mom11, state=None, params=None):
        m = mom11.numpy()
        st = torch.zeros((_ffi.POOL_STATE_DOUBLES,), dtype=torch.float64)
        T = m[0]
        st[K_T], st[K_NPX] = T, m[10]
        if T < 1:
            st[_ffi.POOL_STATUS] = _ffi.TILE_EMPTY_MASK
            return st
        mean = m[1:4] / T
        S2 = np.array([[m[4], m[5], m[6]], [m[5], m[7], m[8]], [m[6], m[8], m[9]]])
        _, V = np.linalg.eigh((S2 - T * np.outer(mean, mean)) / (T - 1.0))
        V = V[:, [2, 1]].copy()
        for i in range(2):
            if V[0, i] < 0:
                V[:, i] *= -1.0
        st[K_VD:K_VD + 6] = torch.from_numpy(V.reshape(6))
        st[K_VF:K_VF + 6] = torch.from_numpy(V.astype(np.float32).astype(np.float64).reshape(6))
        for t, pct in enumerate((1.0, 99.0)):
            k, g = sd.percentile_position(int(T), pct)
            st[K_K + t], st[K_G + t] = k, g
        return st

    def basis_of(state, keyset):
        return state[K_VF:K_VF + 6].numpy() if keyset == _ffi.KEYSET_ANGLE else state[_ffi.POOL_M:_ffi.POOL_M + 6].numpy()

    def pool_histogram(tiles, keyset, state, rnd, slog, hist_out, params=None):
        pre = [int(state[K_PREFIX + t]) for t in range(2)]
        hist_out += hist(tiles, keyset, basis_of(state, keyset), pre, 8 * rnd, every=1 << slog)
        return hist_out

    def pool_pick(state, keyset, rnd, h):
        N = float(state[K_T] if keyset == _ffi.KEYSET_ANGLE else state[K_NPX])
        hc = h.numpy()
        for t in range(2):
            if rnd == 0:
                tot = int(hc[t].sum())
                f = min(max(float(state[K_K + t]) / (N - 1.0) if N > 1 else 0.0, 0.0), 1.0)
                state[K_TOT + t], state[K_KS + t], state[K_BELOW + t], state[K_PREFIX + t] = tot, (np.floor(f * (tot - 1.0)) if tot else 0.0), 0.0, 0.0
                if tot == 0:
                    state[_ffi.POOL_MISS] = float(int(state[_ffi.POOL_MISS]) | (1 if keyset == _ffi.KEYSET_ANGLE else 2))
            want = int(state[K_KS + t] - state[K_BELOW + t])
            cum, b = 0, 0
            while b < 255 and not (cum + int(hc[t][b]) > want):
                cum += int(hc[t][b]); b += 1
            state[K_BELOW + t] += cum
            state[K_PREFIX + t] = float((int(state[K_PREFIX + t]) << 8) | b)
        if rnd == 2:
            for t in range(2):
                est = (int(state[K_PREFIX + t]) << 8) | 0x80
                state[K_WLO + t] = float(min(max(est - 32768, 0), 0xffffffff - 65535))
                state[K_PREFIX + t] = 0.0

    def pool_window(tiles, keyset, state, buf, params=None):
        buf += window(tiles, keyset, basis_of(state, keyset), [int(state[K_WLO]), int(state[K_WLO + 1])])
        return buf

    def pool_resolve(state, keyset, win, params=None):
        N = int(state[K_T] if keyset == _ffi.KEYSET_ANGLE else state[K_NPX])
        b = win.numpy()
        res = []
        for t in range(2):
            histo, below = b[t * 65536:(t + 1) * 65536], int(b[2 * 65536 + t])
            k = min(max(int(state[K_K + t]), 0), N - 1)
            k1 = min(k + 1, N - 1)
            if not (N >= 1 and below <= k and k1 < below + int(histo.sum())):
                state[_ffi.POOL_MISS] = float(int(state[_ffi.POOL_MISS]) | (1 if keyset == _ffi.KEYSET_ANGLE else 2))
                if keyset != _ffi.KEYSET_ANGLE:                  # like k_pool_resolve: an unusable state ends with NaN in (M, maxC)
                    state[_ffi.POOL_M:_ffi.POOL_M + 6] = float("nan")
                    state[_ffi.POOL_MAXC:_ffi.POOL_MAXC + 2] = float("nan")
                return
            cum = np.cumsum(histo)
            lo = int(state[K_WLO + t])
            res += [ord2f(lo + int(np.searchsorted(cum, k - below, side="right"))), ord2f(lo + int(np.searchsorted(cum, k1 - below, side="right")))]
        state[K_RES:K_RES + 4] = torch.tensor(res, dtype=torch.float64)
        if keyset == _ffi.KEYSET_ANGLE:
            import math

            def ang(p):
                if abs(p) <= 1.0:
                    return math.atan2(p, 1.0 - abs(p))
                pp = 2.0 - p if p > 0 else -2.0 - p
                return math.atan2(pp, -(1.0 - abs(pp)))
            V = state[K_VD:K_VD + 6].numpy().reshape(3, 2)
            phis = [sd.np_lerp(ang(res[0]), ang(res[1]), float(state[K_G])), sd.np_lerp(ang(res[2]), ang(res[3]), float(state[K_G + 1]))]
            v1, v2 = V @ np.array([math.cos(phis[0]), math.sin(phis[0])]), V @ np.array([math.cos(phis[1]), math.sin(phis[1])])
            M = np.array([v1, v2]) if v1[0] > v2[0] else np.array([v2, v1])
            M = M / np.linalg.norm(M, axis=1, keepdims=True)
            state[_ffi.POOL_M:_ffi.POOL_M + 6] = torch.from_numpy(M.reshape(6))
            k, g = sd.percentile_position(int(state[K_NPX]), 99.0)
            state[K_K], state[K_K + 1], state[K_G], state[K_G + 1] = k, k, g, g
        else:
            for t in range(2):
                state[_ffi.POOL_MAXC + t] = sd.np_lerp(res[2 * t], res[2 * t + 1], float(state[K_G + t]))
            if int(state[_ffi.POOL_MISS]) != 0 or int(state[_ffi.POOL_STATUS]) != 0:
                state[_ffi.POOL_M:_ffi.POOL_M + 6] = float("nan")
                state[_ffi.POOL_MAXC:_ffi.POOL_MAXC + 2] = float("nan")

    engine.pool_begin, engine.pool_histogram, engine.pool_pick = pool_begin, pool_histogram, pool_pick
    engine.pool_window, engine.pool_resolve = pool_window, pool_resolve
    engine.make_params = lambda **kw: None
    engine.tile_moments = tile_moments
    engine.slide_key_histogram = hist
    engine.slide_key_histogram_sampled = lambda tiles, keyset, basis, pre, bits, slog, params=None: hist(tiles, keyset, basis, pre, bits, every=1 << slog)
    engine.slide_key_histogram16 = hist16
    engine.slide_key_window = window
    engine.slide_key_next_above = next_above
    engine.normalize_apply = normalize_apply


def _slide_tiles(big=False):
    from oracle import stain_oracle as so
    if big:      # ~100 k pixels: neighbouring order statistics lie well inside one 65536-key window, both stages take the one-sweep path
        return [so.synth_tile(128, 128, 300 + s) for s in range(5)] + [so.structured_tile("white_bg", 128, 128, 7)]
    return [so.synth_tile(48, 64, 300 + s) for s in range(5)] + [so.structured_tile("white_bg", 64, 48, 7).transpose(1, 0, 2).copy()]


class _FittedTarget:                       # what SlideNormalizer needs of a fitted normalizer
    def __init__(self):
        from oracle import stain_oracle as so
        n = so.ExtractiveStainNormalizer("macenko")
        n.fit(so.synth_tile(64, 64, 1001, so.M_TRUE_TGT))
        self.stain_matrix_target, self.maxC_target = n.stain_matrix_target, n.maxC_target


def _pooled_worker(rank, world, port, force_radix, q, big=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_numpy_engine(force_radix)
    tiles = _slide_tiles(big)
    lo, hi = sd.shard_range(len(tiles), rank, world)              # 6 tiles -> 3 + 3; rank 1 holds the white-background tile
    mine = torch.from_numpy(np.stack(tiles[lo:hi]))
    stats = sd.PooledSlideStatistics()
    M, maxC = stats(mine, merged=False)                           # the three-sweep chain (the one-sweep chain: further down)
    out, M_s, mc_s, st = sd.SlideNormalizer(_FittedTarget(), mode="pooled", merged=False).transform_shard(mine)
    q.put((rank, M, maxC, list(stats.last_path), out.numpy(), M_s.numpy(), mc_s.numpy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("force_radix", [False, True])
def test_pooled_slide_statistics_real_call_on_two_gloo_ranks(force_radix):
    from oracle import stain_oracle as so
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pooled_worker, args=(r, 2, port, force_radix, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the single-rank run of the same code, and the reference's statistics of the concatenated slide
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_pooled_worker, args=(0, 1, port, force_radix, q1))
    p1.start()
    one = q1.get(timeout=300)
    p1.join(timeout=60)
    tall = np.concatenate(_slide_tiles(), axis=0)
    M_ref = so.macenko_stain_matrix(tall)
    c_ref = np.percentile(so.get_concentrations(tall, M_ref), 99, axis=0)
    for rank, M, maxC, path, out, M_s, mc_s in res:
        # (on a slide of 18 k pixels the ranks k and k+1 of the 99th percentile lie further apart than one window is wide:
        #  that stage legitimately falls back -- on every rank alike, since the decision is made on all-reduced counts)
        assert path == (["radix", "radix"] if force_radix else ["window", "radix"]) and path == one[3]
        assert np.array_equal(M, res[0][1]) and np.array_equal(maxC, res[0][2])    # the ranks agree to the bit (no broadcast needed)
        np.testing.assert_allclose(M, one[1], rtol=0, atol=1e-12)                  # = the single-rank run up to the order of the
        np.testing.assert_allclose(maxC, one[2], rtol=1e-12)                       #   moment sums (all-reduce of two partial sums)
        np.testing.assert_allclose(M, M_ref, rtol=0, atol=2e-6)                    # = the reference on the concatenated image
        np.testing.assert_allclose(maxC, c_ref, rtol=2e-6)
        assert np.array_equal(M_s, M) and np.array_equal(mc_s, maxC)
    both = np.concatenate([res[0][4], res[1][4]])                                  # the two shards' outputs = the single-rank output
    d = both.astype(np.int16) - one[4].astype(np.int16)
    assert np.abs(d).max() <= 1 and (d != 0).sum() <= 2                            # (1e-16 in M can flip a byte on a knife edge)


def test_device_driven_pooled_statistics_on_two_gloo_ranks():
    """The device-driven chain (PooledSlideStatistics.enqueue / finish: per step one all-reduce, decisions in the pool state, one
    read-back at the end) on two gloo ranks with a slide large enough for both windows to catch their ranks: the ranks agree to
    the bit, match the single-rank run and the reference's statistics of the concatenated slide, and no host-driven round ran."""
    from oracle import stain_oracle as so
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pooled_worker, args=(r, 2, port, False, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_pooled_worker, args=(0, 1, port, False, q1, True))
    p1.start()
    one = q1.get(timeout=600)
    p1.join(timeout=60)
    tall = np.concatenate(_slide_tiles(True), axis=0)
    M_ref = so.macenko_stain_matrix(tall)
    c_ref = np.percentile(so.get_concentrations(tall, M_ref), 99, axis=0)
    for rank, M, maxC, path, out, M_s, mc_s in res:
        assert path == ["window", "window"] and path == one[3]                     # the device-driven path settled both stages
        assert np.array_equal(M, res[0][1]) and np.array_equal(maxC, res[0][2])    # the ranks agree to the bit
        np.testing.assert_allclose(M, one[1], rtol=0, atol=1e-12)
        np.testing.assert_allclose(maxC, one[2], rtol=1e-12)
        np.testing.assert_allclose(M, M_ref, rtol=0, atol=2e-6)
        np.testing.assert_allclose(maxC, c_ref, rtol=2e-6)
        assert np.array_equal(M_s, M) and np.array_equal(mc_s, maxC)


def _density_worker(rank, world, port, q, pass_total):
    """Uneven shards (3 + 4 tiles of 1024^2 on two ranks): what sample density does each rank hand to the sampled histogram passes?"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stainlib_amd import _ffi, engine
    seen = []
    engine.make_params = lambda **kw: None
    engine.tile_moments = lambda tiles, params=None, ws=None: torch.zeros((tiles.shape[0], 10), dtype=torch.float64)
    engine.pool_begin = lambda mom, state=None, params=None: torch.zeros((_ffi.POOL_STATE_DOUBLES,), dtype=torch.float64)

    def pool_histogram(tiles, keyset, state, rnd, slog, hist_out, params=None):
        seen.append(int(slog))
        return hist_out
    engine.pool_histogram = pool_histogram
    engine.pool_pick = lambda state, keyset, rnd, h: None
    engine.pool_window = lambda tiles, keyset, state, buf, params=None: buf
    engine.pool_resolve = lambda state, keyset, win, params=None: None
    lo, hi = sd.shard_range(7, rank, world)
    mine = torch.empty((hi - lo, 1024, 1024, 3), dtype=torch.uint8)
    sd.PooledSlideStatistics().enqueue(mine, n_tiles_total=7 if pass_total else None)
    q.put((rank, hi - lo, seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pass_total", [False, True])
def test_uneven_shards_agree_on_the_sample_density(pass_total):
    """Round-3 advisor finding: the density was derived from each rank's OWN tile count; with 3 + 4 tiles of 1024^2 on two ranks
    (2 x 4 Mpx = 8.4 Mpx -> every other row; 2 x 3 Mpx = 6.3 Mpx -> every row) the all-reduced sample histograms mixed densities."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_density_worker, args=(r, 2, port, q, pass_total)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[1] for r in res) == [3, 4]
    assert res[0][2] == res[1][2] and len(res[0][2]) == 6 and len(set(res[0][2])) == 1
    assert res[0][2][0] == (0 if pass_total else 1)          # 7 Mpx -> every row; agreed-on 2 x 4 Mpx -> every other row


# ---- the ONE-SWEEP pooled chain (PooledSlideStatistics.enqueue_merged, sl_pool2_*) on two gloo ranks: the product orchestration with
# the numpy stand-ins of tests/pool2_standins.py (see there for what they restate and what they replace by harness knowledge) ----------
def _many_tiles():
    """48 tiles of 384 x 256 (4.7 Mpx: the chain's sample is one pixel in two), two of them background"""
    from oracle import stain_oracle as so
    return [so.synth_tile(384, 256, 500 + s) for s in range(46)] + [np.full((384, 256, 3), 255, np.uint8)] * 2


def _merged_worker(rank, world, port, q, break_it=False, tiles_fn="big"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import pool2_standins
    _install_numpy_engine(False)
    tiles = _slide_tiles(True) if tiles_fn == "big" else _many_tiles()
    pool2_standins.install(tiles, break_it)
    lo, hi = sd.shard_range(len(tiles), rank, world)
    mine = torch.from_numpy(np.stack(tiles[lo:hi]))
    stats = sd.PooledSlideStatistics()
    direct = stats.finish(stats.enqueue_merged(mine, n_tiles_total=len(tiles)))
    miss = stats.last_miss
    M, maxC = stats(mine, n_tiles_total=len(tiles))
    path = list(stats.last_path)
    out, M_s, mc_s, st = sd.SlideNormalizer(_FittedTarget(), mode="pooled").transform_shard(mine, n_tiles_total=len(tiles))
    q.put((rank, direct is not None, miss, M, maxC, path, out.numpy(), M_s.numpy(), mc_s.numpy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_merged(world, break_it=False, tiles_fn="big"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merged_worker, args=(r, world, port, q, break_it, tiles_fn)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("tiles_fn", ["big", "many"])
def test_one_sweep_pooled_chain_on_two_gloo_ranks(tiles_fn):
    """PooledSlideStatistics.enqueue_merged / SlideNormalizer.transform_shard (the product orchestration) on two gloo ranks: the ranks
    agree to the bit without a broadcast, match the single-rank run and the reference's statistics of the concatenated slide, and
    the one-sweep chain settled both stages ("big": the sample is every pixel; "many": one pixel in two)."""
    from oracle import stain_oracle as so
    res, one = _run_merged(2, tiles_fn=tiles_fn), _run_merged(1, tiles_fn=tiles_fn)[0]
    tiles = _slide_tiles(True) if tiles_fn == "big" else _many_tiles()
    tall = np.concatenate(tiles, axis=0)
    M_ref = so.macenko_stain_matrix(tall)
    c_ref = np.percentile(so.get_concentrations(tall, M_ref), 99, axis=0)
    for rank, direct, miss, M, maxC, path, out, M_s, mc_s in res:
        assert direct and miss == 0 and path == ["merged", "merged"] and path == one[5]
        assert np.array_equal(M, res[0][3]) and np.array_equal(maxC, res[0][4])         # the ranks agree to the bit
        np.testing.assert_allclose(M, one[3], rtol=0, atol=1e-10)                       # = the single-rank run up to the order of the moment sums
        np.testing.assert_allclose(maxC, one[4], rtol=1e-10)                            #   (numpy's pairwise sums over 4.7 M pixels: 1e-11)
        np.testing.assert_allclose(M, M_ref, rtol=0, atol=2e-6)                         # = the reference on the concatenated image
        np.testing.assert_allclose(maxC, c_ref, rtol=2e-6)
        assert np.array_equal(M_s, M) and np.array_equal(mc_s, maxC)
    both = np.concatenate([res[0][6], res[1][6]])
    d = both.astype(np.int16) - one[6].astype(np.int16)
    assert np.abs(d).max() <= 1 and (d != 0).sum() <= 2


def test_one_sweep_pooled_chain_reports_a_miss_and_the_caller_falls_back_on_two_gloo_ranks():
    """A candidate list that lacks pixels it should hold (the stand-in's sweep drops the end of the lower angular tail): the chain's
    checks catch it on every rank alike (a miss, NaN statistics, never a wrong number) and __call__ / transform_shard settle the slide
    with the three-sweep chain -- same numbers as the reference."""
    from oracle import stain_oracle as so
    res = _run_merged(2, break_it=True)
    tall = np.concatenate(_slide_tiles(True), axis=0)
    M_ref = so.macenko_stain_matrix(tall)
    for rank, direct, miss, M, maxC, path, out, M_s, mc_s in res:
        assert not direct and (miss & 1) and path == ["window", "window"]
        assert np.array_equal(M, res[0][3]) and np.array_equal(maxC, res[0][4])
        np.testing.assert_allclose(M, M_ref, rtol=0, atol=2e-6)
        assert np.array_equal(M_s, M)


def test_sample_density_of_the_one_sweep_chain_is_a_function_of_the_slide_alone():
    """sample_log2_for: the whole slide up to 4 Mpx, then a sample that grows like pixels^(2/3) (one 64-pixel sub-row in 2^s) -- a pure
    function of the agreed pixel count, so that every rank picks the same density without a collective; the forced density of the
    tests (PooledSlideStatistics.sample_log2) overrides it on the instance only."""
    from stainlib_amd.distributed import PooledSlideStatistics as S
    assert [S.sample_log2_for(n) for n in (1, 1 << 20, 1 << 22)] == [0, 0, 0]
    assert S.sample_log2_for((1 << 22) + 1) == S.sample_log2_for(5_000_000) == 3
    assert S.sample_log2_for(512 << 20) == 6 and S.sample_log2_for(12_500 << 20) == 7 and S.sample_log2_for(100_000 << 20) == 8
    last = 0
    for e in range(0, 41):
        s = S.sample_log2_for(1 << e)
        assert 0 <= s <= 12 and s >= last                         # monotone, within what sl_pool2_* accept
        last = s
    a, b = S(group=False), S(group=False)
    a.sample_log2 = 4
    assert b.sample_log2 is None and S(group=False).sample_log2 is None
