"""numpy stand-ins for the device steps of the ONE-SWEEP pooled chain (sl_pool2_*), for the world-size-2 gloo tests on the CPU.

The orchestration under test is the product's (stainlib_amd/distributed.py: PooledSlideStatistics.enqueue_merged, finish, __call__,
SlideNormalizer.transform_shard): the sample / bands / sweep / exact / step sequence with its all-reduces, the rank-independent
sample density, the agreement of the ranks without a broadcast, the fall-back to the three-sweep chain.  The stand-ins keep the
contracts of include/stainlib_hip.h.  What a stand-in cannot restate cheaply -- the PROOF that a pixel is plain under every plane and
stain matrix the sample leaves possible (csrc/slide_merged.hip, tested on the GPU against the oracle) -- it replaces by knowledge
only a test harness has: the exact statistics of the whole slide, computed up front from all the tiles.  A pixel is left off the
candidate list only when its EXACT key lies strictly inside the thresholds the chain goes on to check, so the contract "every pixel
not on the list is plain" holds by construction; the rank bookkeeping (ranks shifted by the number of unlisted pixels), the
three-level narrowing in the ordered binary32 domain and the final checks are restated as k_p2_exact / k_p2_step do them."""
import math

import numpy as np
import torch

from stainlib_amd import distributed as sd

NB, TW, TP, WB, WBITS, KB, KBITS = 8192, 256, 8, 2048, 11, 4096, 12   # grid bins, tail words (TP per slot), coarse window bins and key bits per level, single keys of a last level
K_T, K_NPX, K_VD, K_VF, K_K, K_G, K_TS, K_NS, K_SLOG, K_BRK, K_BR, K_WLO, K_RES, K_SH, K_DONE, K_LEVEL = \
    10, 11, 12, 18, 24, 26, 30, 31, 32, 60, 100, 240, 110, 244, 120, 121
K_VH, K_MH, K_L = 34, 140, 150                         # (K_MH, K_L: private to the stand-ins -- the sample's stain matrix, the thresholds)


def f2ord(a):
    u = np.asarray(a, np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint64)


def ord2f(o):
    o = int(o)
    bits = (o & 0x7fffffff) if (o & 0x80000000) else (~o & 0xffffffff)
    return float(np.array([bits], np.uint32).view(np.float32)[0])


def install(all_tiles, break_it=False):
    from oracle import stain_oracle as so
    from stainlib_amd import _ffi, engine

    def tissue(px):
        return (so.lab_l8(px.reshape(1, -1, 3)) / 255.0 < 0.8).ravel()

    def od_of(px):
        return so.rgb_to_od(px.reshape(1, -1, 3)).reshape(-1, 3)

    def eig2(m):
        T = m[0]
        mean = m[1:4] / T
        S2 = np.array([[m[4], m[5], m[6]], [m[5], m[7], m[8]], [m[6], m[8], m[9]]])
        _, V = np.linalg.eigh((S2 - T * np.outer(mean, mean)) / (T - 1.0))
        V = V[:, [2, 1]].copy()
        for i in range(2):
            if V[0, i] < 0:
                V[:, i] *= -1.0
        return V

    def angle_keys(od32, V):
        th = od32 @ np.asarray(V, np.float64).astype(np.float32)
        x, y = th[:, 0], th[:, 1]
        d = np.abs(x) + np.abs(y)
        p = np.where(d > 0, y / np.where(d > 0, d, 1), 0).astype(np.float32)
        return np.where(x < 0, np.where(y >= 0, 2.0, -2.0).astype(np.float32) - p, p).astype(np.float32)

    def conc_keys(od32, M):
        return so.lasso2_nonneg(od32.astype(np.float64), np.asarray(M, np.float64).reshape(2, 3), 0.01).astype(np.float32)

    def moments(od):
        S = od.T @ od
        return [float(len(od)), *od.sum(0), S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]

    def ang(p):
        if abs(p) <= 1.0:
            return math.atan2(p, 1.0 - abs(p))
        pp = 2.0 - p if p > 0 else -2.0 - p
        return math.atan2(pp, -(1.0 - abs(pp)))

    def matrix_from(V, pa0, pb0, g0, pa1, pb1, g1):
        phis = [sd.np_lerp(ang(pa0), ang(pb0), g0), sd.np_lerp(ang(pa1), ang(pb1), g1)]
        v1, v2 = V @ np.array([math.cos(phis[0]), math.sin(phis[0])]), V @ np.array([math.cos(phis[1]), math.sin(phis[1])])
        M = np.array([v1, v2]) if v1[0] > v2[0] else np.array([v2, v1])
        return M / np.linalg.norm(M, axis=1, keepdims=True)

    # what only the harness knows: the exact statistics of the whole slide
    tall = np.concatenate([t.reshape(-1, 3) for t in all_tiles])
    V_true = eig2(np.array(moments(od_of(tall)[tissue(tall)])))
    M_true = so.macenko_stain_matrix(np.concatenate(all_tiles, axis=0))

    def pool2_workspace(n, h, w, slog, device):
        return {}

    def pool2_sample(tiles, slog, ws, params=None):
        px = tiles.numpy().reshape(-1, 3)[::1 << slog]
        ws["sample"], ws["sample_tissue"] = px, tissue(px)
        return torch.tensor(moments(od_of(px)[ws["sample_tissue"]]) + [float(len(px)), 0, 0, 0, 0, 0], dtype=torch.float64)

    def pool2_begin(mom, slog, state=None, params=None):
        m = mom.numpy()
        st = torch.zeros((_ffi.POOL2_STATE_DOUBLES,), dtype=torch.float64)
        st[K_TS], st[K_NS], st[K_SLOG] = m[0], m[10], slog
        if m[0] < 256:
            st[_ffi.POOL2_WHY] = 1
            return st
        st[K_VH:K_VH + 6] = torch.from_numpy(eig2(m).reshape(6))
        return st

    def put(hist, rows, below, listed):
        """rows: two grids of NB bins (the sample) or four targets of KB slots (the candidates; None: the target shares its pair's)"""
        hist.zero_()
        stride = (2 * NB) // len(rows)
        for u, row in enumerate(rows):
            if row is not None:
                hist[TW + stride * u:TW + stride * (u + 1)] = torch.from_numpy(row.astype(np.int64))
                hist[u] = int(below[u])
        hist[4] = int(listed)
        return hist

    def pool2_hist(which, keyset, mode, shape, slog, state, ws, hist, params=None):
        assert mode == which
        if which == 0:                                       # the sample on a uniform grid
            od = od_of(ws["sample"]).astype(np.float32)
            if keyset == _ffi.KEYSET_ANGLE:
                k = angle_keys(od[ws["sample_tissue"]], state[K_VH:K_VH + 6].numpy().reshape(3, 2))
                b = np.floor((k.astype(np.float64) + 1.0) * (NB / 2)).astype(np.int64)
                ks, lo, bins = [k, k], [-1.0, -1.0], [b, b]
            else:
                C = conc_keys(od, state[K_MH:K_MH + 6].numpy())
                ks, lo = [C[:, 0], C[:, 1]], [1e-30, 1e-30]
                bins = [np.floor(C[:, t].astype(np.float64) * (NB / 16.0)).astype(np.int64) for t in range(2)]
            rows = [np.bincount(bins[t][(ks[t] >= lo[t]) & (bins[t] < NB)], minlength=NB) for t in range(2)]
            return put(hist, rows, [int((ks[t] < lo[t]).sum()) for t in range(2)], 0)
        if int(state[K_DONE]) & (1 << keyset):               # a settled key set: the pass does not run
            return hist.zero_()
        rgb, fa, fc = ws["cand"]
        sel = fa if keyset == _ffi.KEYSET_ANGLE else fc
        od = od_of(rgb[sel]).astype(np.float32)
        if keyset == _ffi.KEYSET_ANGLE:
            k = angle_keys(od, state[K_VF:K_VF + 6].numpy().reshape(3, 2))
            ks = [k, k]
        else:
            C = conc_keys(od, state[_ffi.POOL_M:_ffi.POOL_M + 6].numpy())
            ks = [C[:, 0], C[:, 1]]
        rows, below = [], []
        for u in range(4):                                   # target u: rank k (even) / k + 1 (odd) of order statistic u // 2
            o, wlo, sh = f2ord(ks[u // 2]).astype(np.int64), int(state[K_WLO + u]), int(state[K_SH + u])
            if u & 1 and (wlo, sh) == (int(state[K_WLO + u - 1]), int(state[K_SH + u - 1])):
                rows.append(None)                            # the pair's histogram serves both ranks
                below.append(0)
                continue
            b = (o - wlo) >> sh
            rows.append(np.bincount(b[(o >= wlo) & (b < (KB if sh == 0 else WB))], minlength=KB))    # single keys: KB of them; coarse bins: WB
            below.append(int((o < wlo).sum()))
        return put(hist, rows, below, int(sel.sum()))

    def rank_bins(h, below, ranks):
        cum = np.cumsum(h)
        out = []
        for r in ranks:
            r = int(r)
            out.append(-1 if r < below else (int(np.searchsorted(cum, r - below, side="right")) if r - below < cum[-1] else len(h)))
        return out

    def pool2_bands(state, keyset, hist):
        if int(state[_ffi.POOL2_WHY]):
            return
        h = hist.numpy()
        slog = int(state[K_SLOG])
        if keyset == _ffi.KEYSET_ANGLE:
            n = float(state[K_TS])
            ends = []
            for q in (0.01, 0.99):
                r, sdv = q * (n - 1.0), (math.sqrt(q * (1 - q) * n * 16.0) if slog else 0.0)
                lo_b, hi_b = rank_bins(h[TW:TW + NB], int(h[0]), [math.floor(r - 6 * sdv) - 1, math.ceil(r + 6 * sdv) + 1])
                if lo_b < 0 or hi_b >= NB:
                    state[_ffi.POOL2_WHY] = 3
                    return
                ends += [-1.0 + lo_b / (NB / 2), -1.0 + (hi_b + 1) / (NB / 2)]
            state[K_BRK:K_BRK + 4] = torch.tensor(ends, dtype=torch.float64)
            V = state[K_VH:K_VH + 6].numpy().reshape(3, 2)
            m0, m1 = 0.5 * (ends[0] + ends[1]), 0.5 * (ends[2] + ends[3])
            state[K_MH:K_MH + 6] = torch.from_numpy(matrix_from(V, m0, m0, 0.0, m1, m1, 0.0).reshape(6))
        else:
            n = float(state[K_NS])
            r, sdv = 0.99 * (n - 1.0), (math.sqrt(0.0099 * n * 16.0) if slog else 0.0)
            for t in range(2):
                lo_b, = rank_bins(h[TW + NB * t:TW + NB * (t + 1)], int(h[t]), [math.floor(r - 6 * sdv) - 1])
                if lo_b < 0:
                    state[_ffi.POOL2_WHY] = 5
                    return
                state[K_L + t] = 0.9 * lo_b / (NB / 16.0)        # a threshold safely below the 99th percentile under any nearby matrix

    def pool2_sweep(tiles, slog, state, ws, params=None):
        px = tiles.numpy().reshape(-1, 3)
        if int(state[_ffi.POOL2_WHY]):
            ws["cand"] = (px[:0], np.zeros(0, bool), np.zeros(0, bool))
            return torch.zeros((16,), dtype=torch.float64)
        ts = tissue(px)
        od = od_of(px)
        od32 = od.astype(np.float32)
        hi0, lo1 = float(state[K_BRK + 1]) + 0.01, float(state[K_BRK + 2]) - 0.01
        ka = angle_keys(od32, V_true)
        fa = ts & ~((ka > hi0 + 1e-5) & (ka < lo1 - 1e-5))                          # tissue not strictly inside the cone, by the EXACT keys
        C = conc_keys(od32, M_true)
        L = state[K_L:K_L + 2].numpy()
        fc = (C[:, 0] >= L[0] * (1 - 1e-5)) | (C[:, 1] >= L[1] * (1 - 1e-5))
        if break_it:                                             # a list that silently lacks candidates: the checks must catch it
            fa &= ka > hi0 + 0.02
        keep = fa | fc
        ws["cand"] = (px[keep], fa[keep], fc[keep])
        return torch.tensor(moments(od[ts]) + [0, 0, float(len(px)), 0, 0, 0], dtype=torch.float64)

    def set_window(state, t, lo, hi):
        olo, ohi = int(f2ord(np.float32(lo))), int(f2ord(np.float32(hi)))
        span, sh = max(ohi, olo) - olo + 1, 0
        while span > KB and ((span + (1 << sh) - 1) >> sh) > WB:
            sh += 1
        for j in range(2):                                   # both ranks of the pair start in the same window
            state[K_WLO + 2 * t + j], state[K_SH + 2 * t + j] = float(olo), float(sh)

    def poison(state):
        state[_ffi.POOL_M:_ffi.POOL_M + 6] = float("nan")
        state[_ffi.POOL_MAXC:_ffi.POOL_MAXC + 2] = float("nan")

    def pool2_exact(tot, state):
        m = tot.numpy()
        T = m[0]
        state[K_T], state[K_NPX] = T, m[12]
        miss = 32 if int(state[_ffi.POOL2_WHY]) else 0
        if T < 1:
            state[_ffi.POOL_STATUS] = _ffi.TILE_EMPTY_MASK
        else:
            V = eig2(m)
            state[K_VD:K_VD + 6] = torch.from_numpy(V.reshape(6))
            state[K_VF:K_VF + 6] = torch.from_numpy(V.astype(np.float32).astype(np.float64).reshape(6))
            for t, pct in enumerate((1.0, 99.0)):
                state[K_K + t], state[K_G + t] = sd.percentile_position(int(T), pct)
            if not miss:
                br1, br2 = float(state[K_BRK + 1]) + 0.01, float(state[K_BRK + 2]) - 0.01
                state[K_BR + 1], state[K_BR + 2] = br1, br2
                set_window(state, 0, -2.0, br1)
                set_window(state, 1, br2, 2.0)
        state[_ffi.POOL_MISS] = float(miss)
        state[K_DONE], state[K_LEVEL] = 0.0, 0.0
        if miss or int(state[_ffi.POOL_STATUS]):
            state[K_DONE] = 3.0
            poison(state)

    def pool2_step(state, keyset, hist):
        if int(state[K_DONE]) & (1 << keyset):
            return
        h = hist.numpy()
        bit = 1 if keyset == _ffi.KEYSET_ANGLE else 2
        N = int(state[K_T] if keyset == _ffi.KEYSET_ANGLE else state[K_NPX])
        listed = int(h[4])
        sub = [0 if keyset == _ffi.KEYSET_ANGLE else N - listed, N - listed]
        res, new, exact, bad = [0.0] * 4, [], True, False
        for u in range(4):
            t = u // 2
            k = min(max(int(state[K_K + t]), 0), N - 1)
            r = (min(k + 1, N - 1) if u & 1 else k) - sub[t]
            wlo, sh = int(state[K_WLO + u]), int(state[K_SH + u])
            uh = u - 1 if u & 1 and (wlo, sh) == (int(state[K_WLO + u - 1]), int(state[K_SH + u - 1])) else u
            nb = KB if sh == 0 else WB
            b, = rank_bins(h[TW + KB * uh:TW + KB * uh + nb], int(h[uh]), [r])
            if b < 0 or b >= nb or r < 0:
                bad = True
                new.append((wlo, sh))
            elif sh == 0:
                res[u] = ord2f(wlo + b)
                new.append((wlo, 0))
            else:
                exact = False
                new.append((wlo + (b << sh), sh - WBITS if sh > KBITS else 0))
        miss, level = int(state[_ffi.POOL_MISS]), int(state[K_LEVEL])
        if bad or (not exact and level >= 2):
            state[_ffi.POOL_MISS], state[K_DONE] = float(miss | bit), 3.0
            return poison(state)
        if not exact:
            for u in range(4):
                state[K_WLO + u], state[K_SH + u] = float(new[u][0]), float(new[u][1])
            state[K_LEVEL] = float(level + 1)
            return
        state[K_RES:K_RES + 4] = torch.tensor(res, dtype=torch.float64)
        if keyset == _ffi.KEYSET_ANGLE:
            if not (res[1] <= float(state[K_BR + 1]) and res[2] >= float(state[K_BR + 2])):
                miss |= bit
            V = state[K_VD:K_VD + 6].numpy().reshape(3, 2)
            M = matrix_from(V, res[0], res[1], float(state[K_G]), res[2], res[3], float(state[K_G + 1]))
            state[_ffi.POOL_M:_ffi.POOL_M + 6] = torch.from_numpy(M.reshape(6))
            k, g = sd.percentile_position(int(state[K_NPX]), 99.0)
            state[K_K], state[K_K + 1], state[K_G], state[K_G + 1] = k, k, g, g
            for t in range(2):
                set_window(state, t, float(state[K_L + t]), 64.0)
            state[K_LEVEL], state[_ffi.POOL_MISS], state[K_DONE] = 0.0, float(miss), (3.0 if miss else 1.0)
            if miss:
                poison(state)
        else:
            if not (res[0] >= float(state[K_L]) and res[2] >= float(state[K_L + 1])):
                miss |= bit
            for t in range(2):
                state[_ffi.POOL_MAXC + t] = sd.np_lerp(res[2 * t], res[2 * t + 1], float(state[K_G + t]))
            state[_ffi.POOL_MISS], state[K_DONE] = float(miss), 3.0
            if miss or int(state[_ffi.POOL_STATUS]):
                poison(state)

    sd.PooledSlideStatistics.one_call = False          # step by step also on one rank (sl_pool2_local is the device's)
    engine.pool2_workspace, engine.pool2_sample, engine.pool2_begin, engine.pool2_hist = pool2_workspace, pool2_sample, pool2_begin, pool2_hist
    engine.pool2_bands, engine.pool2_sweep, engine.pool2_exact, engine.pool2_step = pool2_bands, pool2_sweep, pool2_exact, pool2_step
