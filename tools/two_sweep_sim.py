"""CPU model (numpy, binary64) of the two-read-sweep Macenko schedule: what a sweep that knows only the
eigenvectors of a CLUSTER SAMPLE gathered before it must collect, and whether the finish can prove the rest plain once the exact
eigenvectors are known.  Mirrors the device's construction step by step (stats_twosweep.hpp) so that its constants can be tuned here:

  phase 0   sample = n_lines 128-byte lines spread over the tile, m pixels of each; eigenvectors V~ of the sample's tissue pixels;
            angular brackets from the sample's angles under V~ (rank -/+ z sigma, sigma inflated by sqrt(deff) for the clustering);
            the plain cone (aH, aL) as TWO HALF-SPACES in optical-density space: gH . od > kappa S, gL . od > kappa S with
            gH = V~ nH, gL = V~ nL, S = od_r + od_g + od_b -- a 3-vector per threshold, no reference to the in-plane basis;
            the box of stain matrices: 9 in-plane grid points x 5 tilts of the plane; a(M; x) <= a~ + eps (|a~1| + |a~2|) + rho + zeta S
  sweep 1   exact moments; angular candidates = tissue and not provably inside the cone; concentration candidates as in round 3
  finish    exact V; the half-space normals projected onto plane(V): residual <= kappa proves every uncollected tissue pixel inside
            the cone (aH', aL') of the EXACT keys; exact M; T, r, E of its affine map against the box centre's: <= eps, rho, zeta.
An in-plane rotation between V~ and V costs nothing (the half-spaces are 3-D objects); only the TILT of the plane enters kappa / zeta.

    python tools/two_sweep_sim.py [n_lines=2048] [m=8] [deff=2] [ztilt=5]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import stain_oracle as so  # noqa: E402

N_LINES = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M_PER = int(sys.argv[2]) if len(sys.argv) > 2 else 8
DEFF = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
ZTILT = float(sys.argv[4]) if len(sys.argv) > 4 else 5.0
DL = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
Z = 6.0
BOX_FRAC = 0.6
BOX_INFLATE = 1.25
LAM = 0.01


def eig_plane(OD):
    w, V = np.linalg.eigh(np.cov(OD, rowvar=False))
    Vp = V[:, [2, 1]].copy()
    for k in range(2):
        if Vp[0, k] < 0:
            Vp[:, k] *= -1
    return Vp, w[::-1], V[:, 0]


def cluster_sample(h, w, n_lines, m, seed=0):
    """pixel indices: n_lines lines of 128 bytes (42 pixels) at a hashed position inside equal blocks of the tile, m pixels of each"""
    P = h * w
    nl = (3 * P) // 128
    n_lines = min(n_lines, nl)
    blk = nl / n_lines
    rng = np.random.RandomState(seed)
    line = np.minimum((np.arange(n_lines) * blk + rng.uniform(0, blk, n_lines)).astype(np.int64), nl - 1)
    first_px = (line * 128 + 2) // 3
    step = max(1, 40 // m)
    off = rng.randint(0, max(1, 41 - step * (m - 1)), n_lines)
    px = first_px[:, None] + off[:, None] + step * np.arange(m)[None, :]
    return np.minimum(px.ravel(), P - 1)


def lasso_affine(M, lam=LAM):
    G = M @ M.T
    Gi = np.linalg.inv(G)
    return Gi @ M, -lam * Gi.sum(1), G[0, 1]          # W (2x3), k (2), g12


def stain_from(V, a0, a1):
    v1 = V @ np.array([np.cos(a0), np.sin(a0)])
    v2 = V @ np.array([np.cos(a1), np.sin(a1)])
    HE = np.array([v1, v2]) if v1[0] > v2[0] else np.array([v2, v1])
    return so.normalize_rows(HE)


def relate(Wa, ka, Wc, kc):
    """T, r, E with  Wa x + ka = T (Wc x + kc) + r + E x"""
    T = Wa @ Wc.T @ np.linalg.inv(Wc @ Wc.T)
    return T, ka - T @ kc, Wa - T @ Wc


def brackets(keys, q, z, deff):
    s = np.sort(keys)
    n = len(s)
    r = q * (n - 1)
    sd = np.sqrt(q * (1 - q) * n * deff)
    lo_r, hi_r = int(np.floor(r - z * sd)) - 1, int(np.ceil(r + z * sd)) + 1
    return (s[lo_r] if lo_r >= 0 else -np.inf), (s[hi_r] if hi_r <= n - 1 else np.inf)


def run(name, I, seed=0, verbose=True):
    h, w = I.shape[:2]
    P = h * w
    mask = so.tissue_mask(I).ravel()
    ODall = so.rgb_to_od(I).reshape(-1, 3)
    OD = ODall[mask]
    T_ = len(OD)
    V, lam_e, nrm = eig_plane(OD)
    # ---------------- phase 0
    spx = cluster_sample(h, w, N_LINES, M_PER, seed)
    smask = mask[spx]
    Xs = ODall[spx][smask]
    ns = len(Xs)
    Vs, ls, ns_nrm = eig_plane(Xs)
    # tilt of the exact plane against the sample's, and the a-priori scale
    tilt_act = np.abs(V - Vs @ (Vs.T @ V)).max()
    tau = ZTILT * np.sqrt(DEFF / ns) * max(np.sqrt(ls[2] * ls[1]) / (ls[1] - ls[2]), np.sqrt(ls[2] * ls[0]) / (ls[0] - ls[2]))
    tau = max(tau, 2e-4)
    # ... and from the sample's own fourth moments (what a few saturated pixels do to the plane)
    dc = Xs - Xs.mean(0)
    a1, a2, a3 = dc @ Vs[:, 0], dc @ Vs[:, 1], dc @ ns_nrm
    se = max(np.sqrt((a1 * a1 * a3 * a3).sum()) / (ns * (ls[0] - ls[2])), np.sqrt((a2 * a2 * a3 * a3).sum()) / (ns * (ls[1] - ls[2])))
    tau = max(tau, 4.0 * np.sqrt(DEFF) * se)
    phis = np.arctan2(Xs @ Vs[:, 1], Xs @ Vs[:, 0])
    lo0, hi0 = brackets(phis, 0.01, Z, DEFF)
    lo1, hi1 = brackets(phis, 0.99, Z, DEFF)
    dl = DL * tau
    aH, aL = hi0 + dl, lo1 - dl
    nH, nL = np.array([-np.sin(aH), np.cos(aH)]), np.array([np.sin(aL), -np.cos(aL)])
    gH, gL = Vs @ nH, Vs @ nL
    kappa1, kappa2 = tau, tau * tau
    S = OD.sum(1)
    zt_ = np.abs(OD @ ns_nrm)
    plain_ang = ((OD @ gH) > kappa1 * zt_ + kappa2 * S) & ((OD @ gL) > kappa1 * zt_ + kappa2 * S)
    n_ang = int((~plain_ang).sum())
    # today's candidates for comparison: exact V, stratified sample brackets
    phi = np.arctan2(OD @ V[:, 1], OD @ V[:, 0])
    rs = np.random.RandomState(3)
    sub = rs.choice(T_, size=max(16, int(16384 * T_ / P)), replace=False)
    t_lo0, t_hi0 = brackets(phi[sub], 0.01, Z, 1.0)
    t_lo1, t_hi1 = brackets(phi[sub], 0.99, Z, 1.0)
    n_ang_today = int(((phi <= t_hi0) | (phi >= t_lo1)).sum())
    # ---------------- finish, angular part
    pH, pL = V.T @ gH, V.T @ gL
    nx = nrm * np.sign(nrm @ ns_nrm)
    c_n = max(abs(nx @ gH), abs(nx @ gL))
    ok_ang = c_n <= kappa1 and c_n * np.abs(nx - ns_nrm).max() <= kappa2
    aH2, aL2 = np.arctan2(-pH[0], pH[1]), np.arctan2(pL[0], -pL[1])
    proof = bool(np.all((phi[plain_ang] > aH2) & (phi[plain_ang] < aL2)))
    ps = np.sort(phi)
    k0, k1 = int(np.floor(0.01 * (T_ - 1))), int(np.floor(0.99 * (T_ - 1)))
    rot = 0.5 * ((aH2 - aH) + (aL2 - aL))
    b0 = (lo0 + rot - dl, aH2)
    b1 = (aL2, hi1 + rot + dl)
    cover = b0[0] <= ps[k0] and ps[min(k0 + 1, T_ - 1)] <= b0[1] and b1[0] <= ps[k1] and ps[min(k1 + 1, T_ - 1)] <= b1[1]
    members = int(((phi >= b0[0]) & (phi <= b0[1])).sum() + ((phi >= b1[0]) & (phi <= b1[1])).sum())
    # ---------------- the box of stain matrices (phase 0) and the concentration candidates
    closed = np.isfinite([lo0, hi0, lo1, hi1]).all()
    res = dict(name=name, ns=ns, tilt=tilt_act, tau=tau, ok_ang=ok_ang, proof=proof, cover=cover, ang=n_ang / T_, ang_today=n_ang_today / T_,
               members=members / T_, closed=closed)
    if closed:
        m0, r0 = 0.5 * (lo0 + hi0), BOX_FRAC * 0.5 * (hi0 - lo0)
        m1, r1 = 0.5 * (lo1 + hi1), BOX_FRAC * 0.5 * (hi1 - lo1)
        Mc = stain_from(Vs, m0, m1)
        Wc, kc, g12c = lasso_affine(Mc)
        e = np.zeros(2); q = np.zeros(2); zt = np.zeros(2)
        nn = ns_nrm
        tilts = [np.zeros((3, 2))] + [s * tau * np.outer(nn, ek) for ek in (np.array([1.0, 0]), np.array([0, 1.0])) for s in (1, -1)]
        bad = False
        for dV in tilts:
            Vt, _ = np.linalg.qr(Vs + dV)
            Vt = Vt * np.sign((Vt * Vs).sum(0))
            for i0 in (-1, 0, 1):
                for i1 in (-1, 0, 1):
                    Mg = stain_from(Vt, m0 + i0 * r0, m1 + i1 * r1)
                    Wa, ka, g12 = lasso_affine(Mg)
                    Tm, rr, E = relate(Wa, ka, Wc, kc)
                    bad |= g12 < 0
                    for i in range(2):
                        e[i] = max(e[i], abs(Tm[i, i] - 1), abs(Tm[i, 1 - i]))
                        q[i] = max(q[i], abs(rr[i]))
                        zt[i] = max(zt[i], abs(Wa[i] @ nn))
        eps, rho, zeta = BOX_INFLATE * e + 1e-7, BOX_INFLATE * q + 4e-6, BOX_INFLATE * zt + 1e-7
        # sample concentration brackets under the centre
        Cs = so.lasso2_nonneg(ODall[spx], Mc, LAM)
        cl, ch = np.zeros(2), np.zeros(2)
        for i in range(2):
            cl[i], ch[i] = brackets(Cs[:, i], 0.99, Z, DEFF)
        ref = np.where(np.isfinite(ch), ch, 2 * cl + 1)
        zref = 4.0 * np.sqrt(ls[2])
        delta = np.array([eps[i] * (ref[i] + 1.5 * ref[1 - i]) + rho[i] + zeta[i] * zref for i in range(2)])
        Lb, Hb = cl - delta, ch + delta
        thr = Lb - rho
        at = ODall @ Wc.T + kc
        zall = np.abs(ODall @ nn)
        sa = np.abs(at).sum(1)
        flag = np.zeros(P, bool)
        for i in range(2):
            flag |= at[:, i] + eps[i] * sa + zeta[i] * zall >= thr[i]
        n_conc = int(flag.sum())
        # finish: exact M
        Mx = so.macenko_stain_matrix(I)
        Wa, ka, g12 = lasso_affine(Mx)
        Tm, rr, E = relate(Wa, ka, Wc, kc)
        ok_c = g12 >= 0 and not bad
        for i in range(2):
            ok_c &= max(abs(Tm[i, i] - 1), abs(Tm[i, 1 - i])) <= eps[i] and abs(rr[i]) + 4e-6 <= rho[i] and abs(Wa[i] @ nn) <= zeta[i]
        Cx = so.lasso2_nonneg(ODall, Mx, LAM)
        kc_ = int(np.floor(0.99 * (P - 1)))
        cover_c = True
        proof_c = True
        mem_c = 0
        for i in range(2):
            cs = np.sort(Cx[:, i])
            cover_c &= Lb[i] <= cs[kc_] and cs[min(kc_ + 1, P - 1)] <= Hb[i]
            proof_c &= bool(np.all(Cx[~flag, i] < Lb[i]))
            mem_c += int(((Cx[:, i] >= Lb[i]) & (Cx[:, i] <= Hb[i])).sum())
        # ---------------- the colour cube: share of pixels in cells that are not provably plain
        od = so.od_lut()
        lo_od, hi_od = od[np.arange(32) * 8 + 7], od[np.arange(32) * 8]            # od falls with the byte
        gam = np.array(so._srgb_gamma_tab_b(), dtype=np.float64)
        wts = np.array([871.0, 2929.0, 296.0])
        ylim = (so.y_index_threshold(0.8) + 1) * 4096 - 2048

        def bounds(coef, want_max):
            out = np.zeros((32, 32, 32))
            for ch_, c in enumerate(coef):
                a, b = c * lo_od, c * hi_od
                v = np.maximum(a, b) if want_max else np.minimum(a, b)
                out = out + v.reshape([32 if j == ch_ else 1 for j in range(3)])
            return out
        lum_min = sum((wts[c] * gam[np.arange(32) * 8]).reshape([32 if j == c else 1 for j in range(3)]) for c in range(3))
        no_tissue = lum_min >= ylim
        zabs = np.maximum(np.abs(bounds(nn, True)), np.abs(bounds(nn, False)))
        cone = (bounds(gH - kappa2, False) - kappa1 * zabs > 1e-5) & (bounds(gL - kappa2, False) - kappa1 * zabs > 1e-5)
        amax = [bounds(Wc[i], True) + kc[i] for i in range(2)]
        amin = [bounds(Wc[i], False) + kc[i] for i in range(2)]
        sab = np.maximum(np.abs(amax[0]), np.abs(amin[0])) + np.maximum(np.abs(amax[1]), np.abs(amin[1]))
        conc_plain = np.ones((32, 32, 32), bool)
        for i in range(2):
            conc_plain &= amax[i] + eps[i] * sab + zeta[i] * zabs < thr[i] - 1e-5
        cell_plain = (no_tissue | cone) & conc_plain
        flat = I.reshape(-1, 3)
        amb = ~cell_plain[flat[:, 0] >> 3, flat[:, 1] >> 3, flat[:, 2] >> 3]
        res.update(ok_c=bool(ok_c), cover_c=bool(cover_c), proof_c=proof_c, conc=n_conc / P, mem_c=mem_c / P, cube=float(amb.mean()),
                   eps=eps.max(), rho=rho.max(), zeta=zeta.max())
    if verbose:
        print(f"{name:10s} ns {ns:5d} sd3 {np.sqrt(ls[2]):.3f} tilt {tilt_act:.1e} (tau {tau:.1e}) ok_ang {ok_ang} proof {proof} cover {cover}  ang cand {100 * n_ang / T_:5.2f} % of tissue "
              f"(today {100 * n_ang_today / T_:5.2f}) members {100 * members / T_:5.2f} %", end="")
        if closed:
            print(f" | box ok {res['ok_c']} cover {res['cover_c']} proof {res['proof_c']} conc cand {100 * res['conc']:5.2f} % of P, members {100 * res['mem_c']:5.2f} %, "
                  f"eps {res['eps']:.3f} rho {res['rho']:.4f} zeta {res['zeta']:.4f} | cube ambiguous {100 * res['cube']:5.1f} %")
        else:
            print(" | brackets open")
    return res


def tiles():
    yield "iid", so.synth_tile(1024, 1024, 7)
    for kind in ("white_bg", "quantized", "blobs"):
        yield kind, so.structured_tile(kind, 1024, 1024, 21)
    ihc = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tissue_ihc_512.npz"))["input"]
    row = np.concatenate([ihc, ihc[:, ::-1]], axis=1)
    yield "ihc", np.ascontiguousarray(np.concatenate([row, row[::-1]], axis=0))
    yield "ihc512", ihc


if __name__ == "__main__":
    print(f"cluster sample {N_LINES} lines x {M_PER} px, deff {DEFF}, z_tilt {ZTILT}")
    for name, I in tiles():
        for seed in range(3):
            run(f"{name}/{seed}", I, seed)
