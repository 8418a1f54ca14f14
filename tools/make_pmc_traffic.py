#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the counter dumps collect_profiles.sh wrote (<dir>/<tag>_pmc_f.txt, _pmc_w.txt,
_pmc_phase_fetch_n64.txt, _pmc_phase_fetch_n512.txt): per-launch HBM-side traffic of the headline kernels.

    python tools/make_pmc_traffic.py r02 gpurun_out > profiles/r02_pmc_traffic.json
"""
import json
import re
import sys

tag, d = sys.argv[1], sys.argv[2]


def dump(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+([0-9.eE+-]+)", line)
            if m and cur is not None:
                out[cur][m.group(1)] = float(m.group(2))
    return out


def pick(table, needle, counter):
    for k, v in table.items():
        if needle in k and counter in v:
            return v[counter]
    return None


f, w = dump(f"{d}/{tag}_pmc_f.txt"), dump(f"{d}/{tag}_pmc_w.txt")
tiles, px = 512, 1048576
kernels = {}
# algorithmic bytes = SURVEY 8(d)'s compulsory traffic, 3 B/px read + 3 B/px written, for both kernels; the fused kernel's own schedule
# (3 read sweeps + 1 write = 12 B/px) is reported beside it
for name, needle, alg_bpp, sched_bpp in (("k_fused<macenko,transform>", "k_fused<0, true, true", 6, 12), ("k_apply", "k_apply<", 6, 6)):
    fr, wr = pick(f, needle, "FETCH_SIZE"), pick(w, needle, "WRITE_SIZE")
    if fr is None or wr is None:
        continue
    hbm = int(round((2.0 * fr + wr) * 1024))
    alg = alg_bpp * tiles * px
    kernels[name] = {"FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB": wr, "hbm_bytes_per_launch": hbm,
                     "bytes_per_pixel": round(hbm / (tiles * px), 2), "algorithmic_bytes_per_launch": alg,
                     "traffic_over_algorithmic": round(hbm / alg, 4), "schedule_bytes_per_pixel": sched_bpp,
                     "traffic_over_schedule_bytes": round(hbm / (sched_bpp * tiles * px), 4),
                     "read_over_tile_reads": round(2.0 * fr * 1024 / ((sched_bpp - 3) * tiles * px), 4),
                     "write_over_output": round(wr * 1024 / (3 * tiles * px), 4)}
# round 5: the fused kernel with the two-sweep route off / forced (run_fused_once.py 512 0 1 / 512 0 2)
two_sweep = {}
for ts, label in ((1, "three_sweep_every_tile (two_sweep=1)"), (2, "every_tile_tries (two_sweep=2)")):
    try:
        f2, w2 = dump(f"{d}/{tag}_pmc_f_ts{ts}.txt"), dump(f"{d}/{tag}_pmc_w_ts{ts}.txt")
    except OSError:
        continue
    fr, wr = pick(f2, "k_fused<0, true, true", "FETCH_SIZE"), pick(w2, "k_fused<0, true, true", "WRITE_SIZE")
    if fr is None or wr is None:
        continue
    hbm = int(round((2.0 * fr + wr) * 1024))
    two_sweep[label] = {"FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB": wr, "hbm_bytes_per_launch": hbm, "bytes_per_pixel": round(hbm / (tiles * px), 2),
                        "read_bytes_per_pixel": round(2.0 * fr * 1024 / (tiles * px), 2), "write_over_output": round(wr * 1024 / (3 * tiles * px), 4)}
# round 5: the fused FIT kernel (no output): everything it writes is candidate lists, member lists, the sample and scratch
fit = {}
try:
    ff, fw = dump(f"{d}/{tag}_pmc_fit_f.txt"), dump(f"{d}/{tag}_pmc_fit_w.txt")
    fr, wr = pick(ff, "k_fused<0, false, true", "FETCH_SIZE"), pick(fw, "k_fused<0, false, true", "WRITE_SIZE")
    if fr is not None and wr is not None:
        fit = {"FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB": wr, "read_bytes_per_pixel": round(2.0 * fr * 1024 / (tiles * px), 2),
               "write_bytes_per_pixel": round(wr * 1024 / (tiles * px), 3), "write_bytes_per_tile": int(wr * 1024 / tiles),
               "note": "k_fused<macenko, fit>: reads = the one tile sweep (3 B/px) + lists / sample / scratch read back; writes = lists + sample + scratch "
                       "(i.i.d. tiles: ~0.36 MB raw candidates + ~0.32 MB bracket members + 0.06 MB sample per tile; the rest is scratch)"}
except OSError:
    pass
phase = {}
for n in (64, 512):
    try:
        t = dump(f"{d}/{tag}_pmc_phase_fetch_n{n}.txt")
    except OSError:
        continue
    phase[f"n{n}"] = {k: pick(t, needle, "FETCH_SIZE") for k, needle in
                      (("k_moments", "k_moments<"), ("k_select<merged>", "k_select<2"), ("k_select<conc>", "k_select<1"), ("k_apply", "k_apply<"))}
doc = {
    "how": "rocprofv3 --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE on `python tools/run_fused_once.py 512` (512 tiles of "
           "1024x1024x3 uint8 per launch), MI355X, ROCm 7.2 (tools/collect_profiles.sh, tools/make_pmc_traffic.py). Counters are in KiB "
           "and are taken at the L2 (TCC) <-> fabric boundary: hits in the 256 MB Infinity Cache, which sits on the memory side of the "
           "fabric, are NOT subtracted -- the counter cannot tell them from HBM. gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md "
           "(HBM section): FETCH_SIZE counts the 128-B requests of wide coalesced streaming reads as 64 B, so it is doubled; the "
           "calibration is k_apply in this same file, whose read volume is known exactly (512 x 3 145 728 B = 1 572 864 KiB). "
           "WRITE_SIZE needs no correction.",
    "tiles_per_launch": tiles, "pixels_per_tile": px, "kernels": kernels,
    "fused_kernel_by_two_sweep_mode": dict(two_sweep, note="round 5 (stats_twosweep.hpp): `kernels` above is the DEFAULT (every workgroup tries the two-sweep route; on the "
                                           "i.i.d. tiles of this run every tile takes it); here the route off for every tile and forced for every tile"),
    "fused_fit_kernel": fit,
    "per_phase_schedule_fetch_KiB_raw": dict(phase, tile_KiB=3072, note="one launch per phase, FETCH_SIZE per kernel launch at 64 tiles (192 MB of "
                                             "tiles: fits the 256 MB Infinity Cache) and at 512 tiles (1.6 GB): the same raw KiB per tile in both -- "
                                             "the counter does not see Infinity Cache hits; whether residency buys TIME: *_phase_classes.txt, DESIGN.md 4.1"),
    "raw_counter_dumps": [f"profiles/{tag}_pmc_{x}.txt" for x in ("f", "w", "s1", "s2", "phase_fetch_n64", "phase_fetch_n512")],
}
print(json.dumps(doc, indent=1))
