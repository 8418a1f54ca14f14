/* Argument checks of the C ABI (include/stainlib_hip.h) on the HOST side, no GPU needed: every refused call must return its
 * error code before anything is launched or dereferenced.  Built and run under AddressSanitizer by `make -C stainlib_amd/csrc asan`
 * (tests/test_host_api.py::test_c_abi_argument_checks_under_asan).  All data pointers of the ABI are DEVICE pointers: the host side
 * never reads through them, so the non-null ones below are deliberately wild. */
#include <stdio.h>
#include <string.h>
#include "../include/stainlib_hip.h"

static int checks = 0, failed = 0;
#define EXPECT(expr, want) do { long long got_ = (long long)(expr); ++checks; \
    if (got_ != (long long)(want)) { ++failed; printf("FAIL line %d: %s = %lld, expected %lld\n", __LINE__, #expr, got_, (long long)(want)); } } while (0)

int main(void) {
    uint8_t* rgb = (uint8_t*)0x100000;  uint8_t* out = (uint8_t*)0x200000;
    double* d6 = (double*)0x300000;     double* d2 = (double*)0x300100;
    int32_t* st = (int32_t*)0x300200;   float* f = (float*)0x400000;
    void* ws = (void*)0x10000000;       /* 256-byte aligned */
    const int n = 4, h = 64, w = 48;
    SlParams p;
    sl_default_params(0);               /* must not crash */
    sl_default_params(&p);
    EXPECT(p.luminosity_threshold == 0.8 && p.angular_percentile == 99.0 && p.lasso_lambda == 0.01 && p.schedule == 0, 1);
    EXPECT(sl_version() > 0, 1);
    for (int code = -1010; code <= 4; ++code) { const char* s = sl_error_string(code); EXPECT(s != 0 && strlen(s) > 0 && strlen(s) < 96, 1); }
    /* workspace sizes: positive for every operator that needs one, 0 for nonsense */
    const int ops[] = {SL_OP_MACENKO_FIT, SL_OP_MACENKO_TRANSFORM, SL_OP_VAHADANE_FIT, SL_OP_VAHADANE_TRANSFORM, SL_OP_HED_AUGMENT,
                       SL_OP_TILE_MOMENTS, SL_OP_LAB_STATS};
    for (unsigned i = 0; i < sizeof(ops) / sizeof(ops[0]); ++i) {
        EXPECT(sl_workspace_bytes(ops[i], n, h, w) > 0, 1);
        EXPECT(sl_workspace_bytes(ops[i], n, h, w) % 256, 0);
        EXPECT(sl_workspace_bytes(ops[i], 0, h, w), 0);
        EXPECT(sl_workspace_bytes(ops[i], n, -1, w), 0);
    }
    EXPECT(sl_workspace_bytes(999, n, h, w), 0);
    EXPECT(sl_workspace_bytes(SL_OP_MACENKO_TRANSFORM, 2048, 1024, 1024) > sl_workspace_bytes(SL_OP_MACENKO_TRANSFORM, 512, 1024, 1024), 1);
    const size_t need = sl_workspace_bytes(SL_OP_MACENKO_TRANSFORM, n, h, w), needv = sl_workspace_bytes(SL_OP_VAHADANE_TRANSFORM, n, h, w);
    /* fit / transform: null input, no tiles, bad shapes, workspace missing / short / misaligned, missing output or target */
    EXPECT(sl_macenko_fit(0, n, h, w, &p, d6, d2, st, ws, need, 0), SL_ERR_BADARG);
    EXPECT(sl_macenko_fit(rgb, 0, h, w, &p, d6, d2, st, ws, need, 0), SL_ERR_BADARG);
    EXPECT(sl_macenko_fit(rgb, n, 0, w, &p, d6, d2, st, ws, need, 0), SL_ERR_BADARG);
    EXPECT(sl_macenko_fit(rgb, n, h, -5, 0, d6, d2, st, ws, need, 0), SL_ERR_BADARG);
    EXPECT(sl_macenko_fit(rgb, n, 65536, 65536, 0, d6, d2, st, ws, need, 0), SL_ERR_BADARG);       /* more than 2^30 pixels */
    EXPECT(sl_macenko_fit(rgb, n, h, w, 0, d6, d2, st, 0, need, 0), SL_ERR_WORKSPACE);
    EXPECT(sl_macenko_fit(rgb, n, h, w, 0, d6, d2, st, ws, need / 2, 0), SL_ERR_WORKSPACE);
    EXPECT(sl_macenko_fit(rgb, n, h, w, 0, d6, d2, st, (char*)ws + 8, need, 0), SL_ERR_WORKSPACE);
    EXPECT(sl_vahadane_fit(0, n, h, w, 0, d6, d2, st, st, ws, needv, 0), SL_ERR_BADARG);
    EXPECT(sl_vahadane_fit(rgb, n, h, w, 0, d6, d2, st, st, ws, 16, 0), SL_ERR_WORKSPACE);
    for (int sched = 0; sched <= 2; ++sched) {
        p.schedule = sched;
        EXPECT(sl_macenko_transform(0, out, n, h, w, &p, d6, d2, 0, 0, 0, ws, need, 0), SL_ERR_BADARG);
        EXPECT(sl_macenko_transform(rgb, 0, n, h, w, &p, d6, d2, 0, 0, 0, ws, need, 0), SL_ERR_BADARG);
        EXPECT(sl_macenko_transform(rgb, out, n, h, w, &p, 0, d2, 0, 0, 0, ws, need, 0), SL_ERR_BADARG);
        EXPECT(sl_macenko_transform(rgb, out, n, h, w, &p, d6, 0, 0, 0, 0, ws, need, 0), SL_ERR_BADARG);
        EXPECT(sl_macenko_transform(rgb, out, -1, h, w, &p, d6, d2, 0, 0, 0, ws, need, 0), SL_ERR_BADARG);
        /* what THIS call needs (the plan its SlParams select) is checked, not the maximum over every SlParams */
        const size_t mine = sl_workspace_bytes_for(SL_OP_MACENKO_TRANSFORM, n, h, w, &p);
        EXPECT(mine > 0 && mine <= need && mine % 256 == 0, 1);
        EXPECT(sl_macenko_transform(rgb, out, n, h, w, &p, d6, d2, 0, 0, 0, ws, mine - 256, 0), SL_ERR_WORKSPACE);
        EXPECT(sl_workspace_bytes_for(SL_OP_VAHADANE_TRANSFORM, n, h, w, &p) <= needv, 1);
        EXPECT(sl_vahadane_transform(rgb, out, n, h, w, &p, d6, d2, 0, 0, 0, 0, needv, 0), SL_ERR_WORKSPACE);
        EXPECT(sl_vahadane_transform(rgb, 0, n, h, w, &p, d6, d2, 0, 0, 0, ws, needv, 0), SL_ERR_BADARG);
    }
    p.schedule = 1;
    {   /* the one-launch-per-phase schedule has no angular candidate list: less than the fused one at the same batch */
        const size_t per_phase = sl_workspace_bytes_for(SL_OP_MACENKO_TRANSFORM, 512, 1024, 1024, &p);
        p.schedule = 2;
        EXPECT(per_phase < sl_workspace_bytes_for(SL_OP_MACENKO_TRANSFORM, 512, 1024, 1024, &p), 1);
        EXPECT(sl_workspace_bytes_for(SL_OP_MACENKO_TRANSFORM, 512, 1024, 1024, &p) <= sl_workspace_bytes(SL_OP_MACENKO_TRANSFORM, 512, 1024, 1024), 1);
    }
    p.schedule = 0;
    EXPECT(sl_workspace_bytes_for(SL_OP_MACENKO_TRANSFORM, n, h, w, 0), sl_workspace_bytes_for(SL_OP_MACENKO_TRANSFORM, n, h, w, &p));
    EXPECT(sl_workspace_bytes_for(SL_OP_MACENKO_TRANSFORM, 0, h, w, &p), 0);
    EXPECT(sl_workspace_bytes_for(SL_OP_HED_AUGMENT, n, h, w, 0), sl_workspace_bytes(SL_OP_HED_AUGMENT, n, h, w));
    {
        SlParams bad = p;
        bad.struct_size = 8;
        EXPECT(sl_workspace_bytes_for(SL_OP_MACENKO_TRANSFORM, n, h, w, &bad), 0);
    }
    /* SlParams.struct_size: set by sl_default_params; a struct of another size (a caller built against another header) is refused by
     * every entry point that takes one, before anything else is looked at */
    EXPECT(p.struct_size == sizeof(SlParams), 1);
    {
        const uint32_t sizes[] = {0u, (uint32_t)sizeof(SlParams) - 8u, (uint32_t)sizeof(SlParams) + 8u};
        for (unsigned i = 0; i < 3; ++i) {
            SlParams q = p;
            q.struct_size = sizes[i];
            EXPECT(sl_macenko_fit(rgb, n, h, w, &q, d6, d2, st, ws, need, 0), SL_ERR_BADARG);
            EXPECT(sl_macenko_transform(rgb, out, n, h, w, &q, d6, d2, 0, 0, 0, ws, need, 0), SL_ERR_BADARG);
            EXPECT(sl_vahadane_fit(rgb, n, h, w, &q, d6, d2, st, st, ws, needv, 0), SL_ERR_BADARG);
            EXPECT(sl_vahadane_transform(rgb, out, n, h, w, &q, d6, d2, 0, 0, 0, ws, needv, 0), SL_ERR_BADARG);
            EXPECT(sl_stain_augment(rgb, out, n, h, w, d6, d6, 0, &q, 0), SL_ERR_BADARG);
            EXPECT(sl_tile_moments(rgb, n, h, w, &q, d6, ws, sl_workspace_bytes(SL_OP_TILE_MOMENTS, n, h, w), 0), SL_ERR_BADARG);
            EXPECT(sl_pool_begin(d6, &q, d6, 0), SL_ERR_BADARG);
        }
    }
    /* single-pass operators */
    EXPECT(sl_normalize_apply(0, out, n, h, w, d6, d2, d6, d2, 0.01, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_normalize_apply(rgb, 0, n, h, w, d6, d2, d6, d2, 0.01, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_normalize_apply(rgb, out, n, h, w, 0, d2, d6, d2, 0.01, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_normalize_apply(rgb, out, 0, h, w, d6, d2, d6, d2, 0.01, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_hed_augment(0, out, n, h, w, d6, d6, 0.05, 0.95, SL_HED_SKIMAGE_018, st, ws, 4096, 0), SL_ERR_BADARG);
    EXPECT(sl_hed_augment(rgb, out, n, h, w, d6, d6, 0.05, 0.95, 9, st, ws, 4096, 0), SL_ERR_BADARG);             /* unknown skimage mode */
    EXPECT(sl_hed_augment(rgb, out, n, h, w, d6, d6, 0.05, 0.95, SL_HED_SKIMAGE_018, st, 0, 4096, 0), SL_ERR_WORKSPACE);
    EXPECT(sl_rgb_to_od(0, n, h, w, d6, 0), SL_ERR_BADARG);
    EXPECT(sl_rgb_to_od(rgb, n, h, w, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_od_to_rgb(0, 12, out, st, 0), SL_ERR_BADARG);
    EXPECT(sl_stain_augment(0, out, n, h, w, d6, d6, 0, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_stain_augment(rgb, out, n, h, w, 0, d6, 0, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_tissue_mask(0, n, h, w, 0.8, out, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_tissue_mask(rgb, n, 0, w, 0.8, out, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_concentrations(rgb, n, h, w, 0, 0.01, f, 0), SL_ERR_BADARG);
    EXPECT(sl_concentrations(rgb, n, h, w, d6, 0.01, 0, 0), SL_ERR_BADARG);
    EXPECT(sl_grayscale_augment(rgb, 0, n, h, w, d2, 0), SL_ERR_BADARG);
    EXPECT(sl_rgb_to_lab8(0, out, n, h, w, 0), SL_ERR_BADARG);
    EXPECT(sl_lab8_to_rgb(rgb, 0, n, h, w, 0), SL_ERR_BADARG);
    printf("%s: %d checks, %d failed\n", failed ? "FAILED" : "OK", checks, failed);
    return failed ? 1 : 0;
}
