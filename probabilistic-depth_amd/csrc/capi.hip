// C ABI of the hot path (include/pdepth.h): argument validation + kernel dispatch.
// No torch types, no allocation, no host synchronisation.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/pdepth.h"
#include "kernels.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_desc(const pdepth_sweep_desc* d, const pdepth_camera* cam, const char* who) {
    if (!d || !cam) return fail(PDEPTH_E_ARG, "%s: null descriptor", who);
    if (d->B <= 0 || d->V <= 0 || d->C <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0)
        return fail(PDEPTH_E_ARG, "%s: non-positive dimension B=%d V=%d C=%d D=%d H=%d W=%d", who,
                    d->B, d->V, d->C, d->D, d->H, d->W);
    if ((long long)d->H * d->W > (1ll << 30))
        return fail(PDEPTH_E_ARG, "%s: H*W too large", who);
    if (!cam->K || !cam->R || !cam->t || !cam->rays || !cam->cxcy)
        return fail(PDEPTH_E_ARG, "%s: null camera pointer", who);
    if (d->blas_mode != PDEPTH_BLAS_FMA && d->blas_mode != PDEPTH_BLAS_SEPARATE)
        return fail(PDEPTH_E_ARG, "%s: unknown blas_mode %d", who, d->blas_mode);
    const long long chw = (long long)d->C * d->H * d->W;
    if (d->src_vstride < chw || d->src_bstride < 0 || d->ref_bstride < 0)
        return fail(PDEPTH_E_ARG, "%s: bad strides", who);
    return PDEPTH_OK;
}

pdepth::SweepArgs make_args(const pdepth_sweep_desc* d, const pdepth_camera* cam, const float* ref,
                            const float* src, const float* d_candi) {
    pdepth::SweepArgs a{};
    a.ref = ref; a.src = src; a.packed_src = nullptr;
    a.K = cam->K; a.R = cam->R; a.t = cam->t; a.rays = cam->rays; a.cxcy = cam->cxcy;
    a.d_candi = d_candi;
    a.B = d->B; a.V = d->V; a.C = d->C; a.D = d->D; a.H = d->H; a.W = d->W;
    a.metric = d->metric; a.sigma = d->sigma; a.blas_mode = d->blas_mode;
    a.fast_div = d->algo != PDEPTH_ALGO_DIRECT;
    a.ref_bstride = d->ref_bstride; a.src_bstride = d->src_bstride; a.src_vstride = d->src_vstride;
    return a;
}

// Implementation behind ALGO_AUTO.  L2 metric, C <= 72, D <= 128, at most 8 views: the distance-form kernel on the matrix
// pipe (sweep_dist.hip).  Everything else that fits a packed layout (L1 has no such form; wider features; more planes): the
// LDS-tiled kernel (sweep_tiled.hip).  Lab builds (-DPDEPTH_LAB) read PDEPTH_SWEEP_IMPL once per process to put another
// implementation behind AUTO for A/B timing: corr | tiled | mfma | cells (the kernels of earlier rounds, kept as
// independent checks); the product library has no such switch.
enum { IMPL_DEFAULT = 0, IMPL_CELLS = 1, IMPL_TILED = 2, IMPL_MFMA = 3, IMPL_CORR = 4 };
int sweep_impl() {
#ifdef PDEPTH_LAB
    static const int impl = [] {
        const char* f = getenv("PDEPTH_SWEEP_IMPL");
        if (!f) return (int)IMPL_DEFAULT;
        if (f[0] == 'c' && f[1] == 'e') return (int)IMPL_CELLS;
        if (f[0] == 'c' && f[1] == 'o') return (int)IMPL_CORR;
        if (f[0] == 'm') return (int)IMPL_MFMA;
        return f[0] == 't' ? (int)IMPL_TILED : (int)IMPL_DEFAULT;
    }();
    return impl;
#else
    return IMPL_DEFAULT;
#endif
}

int launched(hipError_t e, const char* who) {
    if (e != hipSuccess) return fail(PDEPTH_E_LAUNCH, "%s: %s", who, hipGetErrorString(e));
    return PDEPTH_OK;
}

// The tiled kernel needs one int per 16x4 tile (flags of tiles left to the gather kernel) plus the
// channel-group-planar copy of the source views it stages from.
size_t tiled_ws_bytes(const pdepth_sweep_desc* d) {
    return pdepth::sweep_tiled_workspace_bytes(d->B, d->V, d->C, d->H, d->W);
}

// does ALGO_AUTO run on a packed copy of the source for this shape?
bool uses_packed_source(const pdepth_sweep_desc* d) {
    return d->algo != PDEPTH_ALGO_DIRECT && d->D <= pdepth::sweep_tiled_max_planes() && d->W <= 32767 && d->H <= 32767 &&
           (long long)((d->C + 3) / 4 + 2) * d->H * d->W * 16 < (1ll << 31);
}

pdepth::SweepArgs shape_args(const pdepth_sweep_desc* d) {
    pdepth::SweepArgs a{};
    a.B = d->B; a.V = d->V; a.C = d->C; a.D = d->D; a.H = d->H; a.W = d->W; a.metric = d->metric;
    return a;
}

// Which sweep kernel family -- and with it which staging layout of the source -- a descriptor selects.  A pure function of
// the descriptor: the packing entry points and the sweep that follows decide alike.
//   distance form (sweep_dist.hip; layout dist_layout.hpp): what AUTO runs for L2, D <= 128, C <= 72, V <= 8
bool uses_dist(const pdepth_sweep_desc* d) {
    if (!uses_packed_source(d) || d->metric != PDEPTH_METRIC_L2) return false;
    if (d->algo != PDEPTH_ALGO_DIST && !(d->algo == PDEPTH_ALGO_AUTO && sweep_impl() == IMPL_DEFAULT)) return false;
    return pdepth::sweep_dist_supports(shape_args(d));
}
//   correlation form on mean-centred features (sweep_corr.hip; centred channel-group-planar layout): on request
//   (round 4's default; since round 6 in lab builds only -- make LAB=1 --, like the cell-list and first matrix-pipe kernels)
bool uses_corr(const pdepth_sweep_desc* d) {
#ifndef PDEPTH_LAB
    (void)d;
    return false;
#else
    if (!uses_packed_source(d) || d->metric != PDEPTH_METRIC_L2) return false;
    if (d->algo != PDEPTH_ALGO_CORR && !(d->algo == PDEPTH_ALGO_AUTO && sweep_impl() == IMPL_CORR)) return false;
    return pdepth::sweep_corr_supports(shape_args(d));
#endif
}
int source_layout(const pdepth_sweep_desc* d) {
    if (!uses_packed_source(d)) return PDEPTH_LAYOUT_NONE;
    if (uses_dist(d)) return PDEPTH_LAYOUT_DIST16;
    return uses_corr(d) ? PDEPTH_LAYOUT_C4_CENTRED : PDEPTH_LAYOUT_C4;
}

// packed_ready: src is NULL and the workspace already holds the packed source (pdepth_pack_source_f32)
int sweep_common(const pdepth_sweep_desc* d, const pdepth_camera* cam, const float* ref,
                 const float* src, const float* d_candi, float* cost, float* logp, float* depth,
                 void* workspace, size_t workspace_bytes, void* stream, const char* who, bool packed_ready = false) {
    if (int rc = check_desc(d, cam, who)) return rc;
    if (!ref || (!src && !packed_ready) || !d_candi) return fail(PDEPTH_E_ARG, "%s: null input pointer", who);
    if (packed_ready && !uses_packed_source(d))
        return fail(PDEPTH_E_ARG, "%s: this shape / algorithm does not run on a packed source (use pdepth_sweep_dpv_f32)", who);
    if (!cost && !logp && !depth) return fail(PDEPTH_E_ARG, "%s: no output requested", who);
    if (d->metric != PDEPTH_METRIC_L2 && d->metric != PDEPTH_METRIC_L1)
        return fail(PDEPTH_E_ARG, "%s: undefined metric for feature distance (%d)", who, d->metric);
    if (d->algo < PDEPTH_ALGO_AUTO || d->algo > PDEPTH_ALGO_DIST)
        return fail(PDEPTH_E_ARG, "%s: unknown algo %d", who, d->algo);
#ifdef PDEPTH_LAB
    if (d->algo == PDEPTH_ALGO_CELLS && (d->metric != PDEPTH_METRIC_L2 || d->D > pdepth::sweep_cells_max_planes()))
        return fail(PDEPTH_E_ARG, "%s: PDEPTH_ALGO_CELLS needs the L2 metric and D <= %d", who, pdepth::sweep_cells_max_planes());
    if (d->algo == PDEPTH_ALGO_MFMA) {
        const pdepth::SweepArgs probe = make_args(d, cam, ref, src, d_candi);
        if (!pdepth::sweep_mfma_supports(probe))
            return fail(PDEPTH_E_ARG, "%s: PDEPTH_ALGO_MFMA needs the L2 metric, D <= 128 and C <= 72", who);
    }
#else
    if (d->algo == PDEPTH_ALGO_CELLS || d->algo == PDEPTH_ALGO_MFMA || d->algo == PDEPTH_ALGO_CORR)
        return fail(PDEPTH_E_ARG, "%s: PDEPTH_ALGO_CELLS / PDEPTH_ALGO_MFMA / PDEPTH_ALGO_CORR exist in lab builds only (make LAB=1)", who);
#endif
    if (d->algo == PDEPTH_ALGO_CORR && !uses_corr(d))
        return fail(PDEPTH_E_ARG, "%s: PDEPTH_ALGO_CORR needs the L2 metric, D <= 128 and C <= 72", who);
    if (d->algo == PDEPTH_ALGO_DIST && !uses_dist(d))
        return fail(PDEPTH_E_ARG, "%s: PDEPTH_ALGO_DIST needs the L2 metric, D <= 128, C <= 72 and at most 8 source views", who);
    if (d->algo == PDEPTH_ALGO_TILED_2 && d->D > 64)
        return fail(PDEPTH_E_ARG, "%s: PDEPTH_ALGO_TILED_2 needs D <= 64", who);
    if (d->algo >= PDEPTH_ALGO_TILED_1 && !uses_packed_source(d))
        return fail(PDEPTH_E_ARG, "%s: the selected implementation does not support this shape", who);
    if (!(d->sigma > 0.0f) && !(d->sigma < 0.0f))
        return fail(PDEPTH_E_ARG, "%s: sigma must be non-zero", who);
    if (d->D > pdepth::sweep_direct_max_planes(d->C))
        return fail(PDEPTH_E_ARG, "%s: D=%d exceeds the %d planes one launch supports at C=%d", who,
                    d->D, pdepth::sweep_direct_max_planes(d->C), d->C);
    pdepth::SweepArgs a = make_args(d, cam, ref, src, d_candi);
    a.cost_out = cost; a.logp_out = logp; a.depth_out = depth;
    // the tiled kernel addresses one view through a 32-bit buffer descriptor (C*H*W*4 bytes < 2^31)
    // (the tiled kernels also pack a footprint as two 16-bit coordinates)
    if (uses_packed_source(d)) {
        const size_t need = tiled_ws_bytes(d);
        if (!workspace || workspace_bytes < need)
            return fail(PDEPTH_E_WORKSPACE, "%s: ALGO_AUTO needs %zu bytes of workspace (got %zu); "
                        "query pdepth_sweep_workspace_bytes()", who, need, workspace_bytes);
        if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0)
            return fail(PDEPTH_E_WORKSPACE, "%s: workspace must be 256-byte aligned", who);
        if (d->algo == PDEPTH_ALGO_TILED_1)
            return launched(pdepth::launch_sweep_tiled_n1(a, workspace, (hipStream_t)stream, packed_ready), who);
        if (d->algo == PDEPTH_ALGO_TILED_2)
            return launched(pdepth::launch_sweep_tiled_n2(a, workspace, (hipStream_t)stream, packed_ready), who);
        if (uses_dist(d)) return launched(pdepth::launch_sweep_dist(a, workspace, (hipStream_t)stream, packed_ready), who);
#ifdef PDEPTH_LAB
        if (uses_corr(d)) return launched(pdepth::launch_sweep_corr(a, workspace, (hipStream_t)stream, packed_ready), who);
        if (d->algo == PDEPTH_ALGO_MFMA || (d->algo == PDEPTH_ALGO_AUTO && sweep_impl() == IMPL_MFMA && pdepth::sweep_mfma_supports(a)))
            return launched(pdepth::launch_sweep_mfma(a, workspace, (hipStream_t)stream, packed_ready), who);
        if (d->algo == PDEPTH_ALGO_CELLS || (d->algo == PDEPTH_ALGO_AUTO && d->metric == PDEPTH_METRIC_L2 &&
                                             d->D <= pdepth::sweep_cells_max_planes() && sweep_impl() == IMPL_CELLS))
            return launched(pdepth::launch_sweep_cells(a, workspace, (hipStream_t)stream, packed_ready), who);
#endif
        return launched(pdepth::launch_sweep_tiled(a, workspace, (hipStream_t)stream, packed_ready), who);
    }
    return launched(pdepth::launch_sweep_direct(a, (hipStream_t)stream), who);
}

}  // namespace

extern "C" {

int pdepth_abi_version(void) { return PDEPTH_ABI_VERSION; }
const char* pdepth_last_error(void) { return g_err; }

size_t pdepth_sweep_workspace_bytes(const pdepth_sweep_desc* desc) {
    if (!desc || desc->algo == PDEPTH_ALGO_DIRECT) return 0;
    if (desc->B <= 0 || desc->D <= 0 || desc->H <= 0 || desc->W <= 0) return 0;
    return tiled_ws_bytes(desc);
}

int pdepth_sweep_source_layout(const pdepth_sweep_desc* desc) {
    if (!desc || desc->B <= 0 || desc->V <= 0 || desc->C <= 0 || desc->D <= 0 || desc->H <= 0 || desc->W <= 0) return PDEPTH_LAYOUT_NONE;
    return source_layout(desc);
}

int pdepth_sweep_centres_source(const pdepth_sweep_desc* desc) {
    const int l = pdepth_sweep_source_layout(desc);
    return (l == PDEPTH_LAYOUT_C4_CENTRED || l == PDEPTH_LAYOUT_DIST16) ? 1 : 0;
}

int pdepth_sweep_cost_f32(const pdepth_sweep_desc* desc, const pdepth_camera* cam, const float* ref,
                          const float* src, const float* d_candi, float* cost, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (!cost) return fail(PDEPTH_E_ARG, "pdepth_sweep_cost_f32: null output");
    return sweep_common(desc, cam, ref, src, d_candi, cost, nullptr, nullptr, workspace, workspace_bytes,
                        stream, "pdepth_sweep_cost_f32");
}

int pdepth_sweep_dpv_f32(const pdepth_sweep_desc* desc, const pdepth_camera* cam, const float* ref,
                         const float* src, const float* d_candi, float* cost, float* logp,
                         float* depth, void* workspace, size_t workspace_bytes, void* stream) {
    return sweep_common(desc, cam, ref, src, d_candi, cost, logp, depth, workspace, workspace_bytes, stream,
                        "pdepth_sweep_dpv_f32");
}

int pdepth_pack_source_f32(const pdepth_sweep_desc* desc, const float* src, void* workspace, size_t workspace_bytes,
                           void* stream) {
    const char* who = "pdepth_pack_source_f32";
    if (!desc || !src) return fail(PDEPTH_E_ARG, "%s: null pointer", who);
    if (desc->B <= 0 || desc->V <= 0 || desc->C <= 0 || desc->D <= 0 || desc->H <= 0 || desc->W <= 0)
        return fail(PDEPTH_E_ARG, "%s: non-positive dimension", who);
    if (desc->src_vstride < (long long)desc->C * desc->H * desc->W || desc->src_bstride < 0)
        return fail(PDEPTH_E_ARG, "%s: bad strides", who);
    if (!uses_packed_source(desc))
        return fail(PDEPTH_E_ARG, "%s: this shape / algorithm does not run on a packed source", who);
    const size_t need = tiled_ws_bytes(desc);
    if (!workspace || workspace_bytes < need)
        return fail(PDEPTH_E_WORKSPACE, "%s: needs %zu bytes of workspace (got %zu)", who, need, workspace_bytes);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0)
        return fail(PDEPTH_E_WORKSPACE, "%s: workspace must be 256-byte aligned", who);
    pdepth::SweepArgs a{};
    a.src = src;
    a.B = desc->B; a.V = desc->V; a.C = desc->C; a.D = desc->D; a.H = desc->H; a.W = desc->W;
    a.src_bstride = desc->src_bstride; a.src_vstride = desc->src_vstride;
    if (uses_dist(desc)) return launched(pdepth::launch_pack_dist(a, workspace, (hipStream_t)stream), who);
    return launched(pdepth::launch_pack_c4(a, workspace, (hipStream_t)stream, uses_corr(desc)), who);
}

int pdepth_pack_views_f32(const pdepth_sweep_desc* desc, const float* feat, const float* rgb, int32_t pool_rate, float* ref_out,
                          void* workspace, size_t workspace_bytes, void* stream) {
    const char* who = "pdepth_pack_views_f32";
    if (!desc || !feat || !rgb || !ref_out) return fail(PDEPTH_E_ARG, "%s: null pointer", who);
    if (desc->B <= 0 || desc->V <= 0 || desc->C <= 3 || desc->D <= 0 || desc->H <= 0 || desc->W <= 0 || pool_rate < 1)
        return fail(PDEPTH_E_ARG, "%s: bad dimension (C must count the 3 pooled image channels)", who);
    if (!uses_packed_source(desc))
        return fail(PDEPTH_E_ARG, "%s: this shape / algorithm does not run on a packed source", who);
    if ((long long)desc->H * pool_rate * desc->W * pool_rate * 3 >= (1ll << 31))
        return fail(PDEPTH_E_ARG, "%s: image too large", who);
    const size_t need = tiled_ws_bytes(desc);
    if (!workspace || workspace_bytes < need)
        return fail(PDEPTH_E_WORKSPACE, "%s: needs %zu bytes of workspace (got %zu)", who, need, workspace_bytes);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0)
        return fail(PDEPTH_E_WORKSPACE, "%s: workspace must be 256-byte aligned", who);
    pdepth::SweepArgs a{};
    a.B = desc->B; a.V = desc->V; a.C = desc->C; a.D = desc->D; a.H = desc->H; a.W = desc->W;
    if (uses_dist(desc))
        return launched(pdepth::launch_pack_views_dist(a, feat, rgb, pool_rate, desc->H * pool_rate, desc->W * pool_rate, ref_out, workspace,
                                                       (hipStream_t)stream), who);
    return launched(pdepth::launch_pack_views(a, feat, rgb, pool_rate, desc->H * pool_rate, desc->W * pool_rate, ref_out, workspace,
                                              (hipStream_t)stream, uses_corr(desc)), who);
}

int pdepth_sweep_dpv_packed_f32(const pdepth_sweep_desc* desc, const pdepth_camera* cam, const float* ref,
                                const float* d_candi, float* cost, float* logp, float* depth, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return sweep_common(desc, cam, ref, nullptr, d_candi, cost, logp, depth, workspace, workspace_bytes, stream,
                        "pdepth_sweep_dpv_packed_f32", true);
}

int pdepth_dpv_reduce_f32(const float* logits, const float* d_candi, int32_t B, int32_t D, int32_t H,
                          int32_t W, float* logp, float* depth, void* stream) {
    if (!logits || !d_candi) return fail(PDEPTH_E_ARG, "pdepth_dpv_reduce_f32: null input");
    if (!logp && !depth) return fail(PDEPTH_E_ARG, "pdepth_dpv_reduce_f32: no output requested");
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0)
        return fail(PDEPTH_E_ARG, "pdepth_dpv_reduce_f32: non-positive dimension");
    return launched(pdepth::launch_dpv_reduce(logits, d_candi, B, D, H, W, logp, depth,
                                              (hipStream_t)stream), "pdepth_dpv_reduce_f32");
}

int pdepth_dpv_reduce_ex_f32(const float* logits, const float* addend, const float* d_candi, int32_t B, int32_t D,
                             int32_t H, int32_t W, float* logp, float* prob, float* depth, float* variance,
                             float* logp_quarter, void* stream) {
    const char* who = "pdepth_dpv_reduce_ex_f32";
    if (!logits || !d_candi) return fail(PDEPTH_E_ARG, "%s: null input", who);
    if (!logp && !prob && !depth && !variance && !logp_quarter) return fail(PDEPTH_E_ARG, "%s: no output requested", who);
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(PDEPTH_E_ARG, "%s: non-positive dimension", who);
    if (logp_quarter && (H < 4 || W < 4)) return fail(PDEPTH_E_ARG, "%s: quarter output needs H, W >= 4", who);
    if (prob == logits || variance == logits || (addend && (logp == addend || prob == addend)))
        return fail(PDEPTH_E_ARG, "%s: only logp may alias logits", who);
    return launched(pdepth::launch_dpv_reduce_ex(logits, addend, d_candi, B, D, H, W, logp, prob, depth, variance,
                                                 logp_quarter, (hipStream_t)stream), who);
}

int pdepth_dpv_expect_f32(const float* dpv, const float* d_candi, int32_t B, int32_t D, int32_t H,
                          int32_t W, int32_t bv_log, float* depth, void* stream) {
    if (!dpv || !d_candi || !depth) return fail(PDEPTH_E_ARG, "pdepth_dpv_expect_f32: null pointer");
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0)
        return fail(PDEPTH_E_ARG, "pdepth_dpv_expect_f32: non-positive dimension");
    return launched(pdepth::launch_dpv_expect(dpv, d_candi, B, D, H, W, bv_log, depth,
                                              (hipStream_t)stream), "pdepth_dpv_expect_f32");
}

int pdepth_dpv_moments_f32(const float* dpv, const float* d_candi, int32_t B, int32_t D, int32_t H,
                           int32_t W, int32_t bv_log, float* mean, float* variance, void* stream) {
    if (!dpv || !d_candi || !variance) return fail(PDEPTH_E_ARG, "pdepth_dpv_moments_f32: null pointer");
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0)
        return fail(PDEPTH_E_ARG, "pdepth_dpv_moments_f32: non-positive dimension");
    return launched(pdepth::launch_dpv_moments(dpv, d_candi, B, D, H, W, bv_log, mean, variance,
                                               (hipStream_t)stream), "pdepth_dpv_moments_f32");
}

int pdepth_warp_feature_f32(const pdepth_sweep_desc* desc, const pdepth_camera* cam, const float* src,
                            const float* d_candi, float* out, void* stream) {
    if (int rc = check_desc(desc, cam, "pdepth_warp_feature_f32")) return rc;
    if (!src || !d_candi || !out) return fail(PDEPTH_E_ARG, "pdepth_warp_feature_f32: null pointer");
    if (desc->C != desc->D)
        return fail(PDEPTH_E_ARG, "pdepth_warp_feature_f32: needs C == D (got C=%d D=%d)", desc->C, desc->D);
    pdepth::SweepArgs a = make_args(desc, cam, nullptr, src, d_candi);
    return launched(pdepth::launch_warp_feature(a, out, (hipStream_t)stream), "pdepth_warp_feature_f32");
}

int pdepth_sample_coords_f32(const pdepth_sweep_desc* desc, const pdepth_camera* cam,
                             const float* d_candi, float* ix, float* iy, void* stream) {
    if (!desc || !cam) return fail(PDEPTH_E_ARG, "pdepth_sample_coords_f32: null descriptor");
    pdepth_sweep_desc d = *desc;
    if (d.C <= 0) d.C = 1;
    d.src_vstride = (long long)d.C * d.H * d.W;
    if (int rc = check_desc(&d, cam, "pdepth_sample_coords_f32")) return rc;
    if (!d_candi || !ix || !iy) return fail(PDEPTH_E_ARG, "pdepth_sample_coords_f32: null pointer");
    pdepth::SweepArgs a = make_args(&d, cam, nullptr, nullptr, d_candi);
    return launched(pdepth::launch_sample_coords(a, ix, iy, (hipStream_t)stream), "pdepth_sample_coords_f32");
}

size_t pdepth_ufield_workspace_bytes(int32_t B, int32_t H, int32_t W) {
    return (B > 0 && H > 0 && W > 0) ? pdepth::ufield_workspace_bytes(B, H, W) : 0;
}

int pdepth_ufield_f32(const float* dpv, const float* d_candi, const float* intr, const float* mask, int32_t B, int32_t D,
                      int32_t H, int32_t W, int32_t bv_log, float unc_ang, float z_start, float z_end, float min_depth,
                      int32_t quash, float oob_depth, float* plane, float* depth_zero, void* workspace,
                      size_t workspace_bytes, void* stream) {
    const char* who = "pdepth_ufield_f32";
    if (!dpv || !d_candi || !intr || !plane || !depth_zero) return fail(PDEPTH_E_ARG, "%s: null pointer", who);
    if (B <= 0 || D <= 0 || H <= 1 || W <= 1) return fail(PDEPTH_E_ARG, "%s: bad dimension", who);
    const size_t need = pdepth::ufield_workspace_bytes(B, H, W);
    if (!workspace || workspace_bytes < need)
        return fail(PDEPTH_E_WORKSPACE, "%s: needs %zu bytes of workspace (got %zu)", who, need, workspace_bytes);
    return launched(pdepth::launch_ufield(dpv, d_candi, intr, mask, B, D, H, W, bv_log, unc_ang, z_start, z_end, min_depth,
                                          quash, oob_depth, plane, depth_zero, workspace, (hipStream_t)stream), who);
}

int pdepth_dpv_fuse_f32(const float* logp, const float* dmaps, const float* masks, const float* d_candi,
                        int32_t B, int32_t D, int32_t H, int32_t W, float var, float eps, float* fused,
                        float* logfused, void* stream) {
    if (!logp || !dmaps || !masks || !d_candi) return fail(PDEPTH_E_ARG, "pdepth_dpv_fuse_f32: null input");
    if (!fused && !logfused) return fail(PDEPTH_E_ARG, "pdepth_dpv_fuse_f32: no output requested");
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return fail(PDEPTH_E_ARG, "pdepth_dpv_fuse_f32: non-positive dimension");
    if (!(var > 0.0f)) return fail(PDEPTH_E_ARG, "pdepth_dpv_fuse_f32: var must be positive");
    return launched(pdepth::launch_dpv_fuse(logp, dmaps, masks, d_candi, B, D, H, W, var, eps, fused, logfused,
                                            (hipStream_t)stream), "pdepth_dpv_fuse_f32");
}

}  // extern "C" (reopened below)

namespace {
// kernel_size odd, strides >= 1, and no index of the reference's kernel outside its padded buffers
int check_corr(const char* who, int32_t B, int32_t C, int32_t H, int32_t W, int32_t pad, int32_t k, int32_t md, int32_t s1, int32_t s2,
               int* oH, int* oW) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(PDEPTH_E_ARG, "%s: non-positive dimension", who);
    if (pad < 0 || k < 1 || (k & 1) == 0 || md < 0 || s1 < 1 || s2 < 1)
        return fail(PDEPTH_E_ARG, "%s: bad configuration (pad %d, kernel %d, max_displacement %d, stride1 %d, stride2 %d)", who, pad, k, md, s1, s2);
    if (md - (md / s2) * s2 - (k - 1) / 2 < 0)
        return fail(PDEPTH_E_ARG, "%s: kernel_size %d with max_displacement %d / stride2 %d reads outside the padded input in the "
                    "reference kernel (needs (kernel_size-1)/2 <= max_displacement mod stride2)", who, k, md, s2);
    if (!pdepth::correlation_output_size(H, W, pad, k, md, s1, oH, oW))
        return fail(PDEPTH_E_ARG, "%s: empty output (pad %d too small for max_displacement %d, kernel %d)", who, pad, md, k);
    // (one thread per output / input element in a 1-D grid of 256-thread blocks: below the 2^31 - 1 block limit with room to spare)
    if ((long long)(2 * (md / s2) + 1) * (2 * (md / s2) + 1) * *oH * *oW * B >= (1ll << 38))
        return fail(PDEPTH_E_ARG, "%s: output too large for one launch (2^38 elements)", who);
    if ((long long)B * C * H * W >= (1ll << 38))
        return fail(PDEPTH_E_ARG, "%s: input too large for one launch (2^38 elements)", who);
    return PDEPTH_OK;
}
bool corr_fast_path(int32_t pad, int32_t k, int32_t md, int32_t s1, int32_t s2) {   // the configuration of pwclite.py:123-125 and kin
    return k == 1 && s1 == 1 && pad == md && md >= 1 && md % s2 == 0 && md / s2 <= pdepth::correlation_max_radius();
}
}  // namespace

extern "C" {

int pdepth_correlation_output_size(int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size, int32_t max_displacement, int32_t stride1,
                                   int32_t stride2, int32_t* out_channels, int32_t* out_height, int32_t* out_width) {
    int oH = 0, oW = 0;
    if (int rc = check_corr("pdepth_correlation_output_size", 1, 1, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oH, &oW))
        return rc;
    if (out_channels) *out_channels = (2 * (max_displacement / stride2) + 1) * (2 * (max_displacement / stride2) + 1);
    if (out_height) *out_height = oH;
    if (out_width) *out_width = oW;
    return PDEPTH_OK;
}

int pdepth_correlation_forward_f32(const float* input1, const float* input2, int32_t B, int32_t C, int32_t H,
                                   int32_t W, int32_t pad_size, int32_t kernel_size, int32_t max_displacement,
                                   int32_t stride1, int32_t stride2, int32_t corr_multiply, float* output,
                                   void* stream) {
    (void)corr_multiply;
    const char* who = "pdepth_correlation_forward_f32";
    if (!input1 || !input2 || !output) return fail(PDEPTH_E_ARG, "%s: null pointer", who);
    int oH, oW;
    if (int rc = check_corr(who, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oH, &oW)) return rc;
    if (corr_fast_path(pad_size, kernel_size, max_displacement, stride1, stride2))
        return launched(pdepth::launch_correlation_forward(input1, input2, B, C, H, W, max_displacement / stride2, stride2,
                                                           output, (hipStream_t)stream), who);
    return launched(pdepth::launch_correlation_general_forward(input1, input2, 0, B, C, H, W, pad_size, kernel_size, max_displacement,
                                                               stride1, stride2, output, (hipStream_t)stream), who);
}

int pdepth_correlation_backward_f32(const float* input1, const float* input2, const float* grad_output, int32_t B,
                                    int32_t C, int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size,
                                    int32_t max_displacement, int32_t stride1, int32_t stride2, int32_t corr_multiply,
                                    float* grad_input1, float* grad_input2, void* stream) {
    (void)corr_multiply;
    const char* who = "pdepth_correlation_backward_f32";
    if (!input1 || !input2 || !grad_output || (!grad_input1 && !grad_input2)) return fail(PDEPTH_E_ARG, "%s: null pointer", who);
    int oH, oW;
    if (int rc = check_corr(who, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oH, &oW)) return rc;
    if (corr_fast_path(pad_size, kernel_size, max_displacement, stride1, stride2))
        return launched(pdepth::launch_correlation_backward(input1, input2, grad_output, B, C, H, W, max_displacement / stride2,
                                                            stride2, grad_input1, grad_input2, (hipStream_t)stream), who);
    return launched(pdepth::launch_correlation_general_backward(input1, input2, grad_output, 0, B, C, H, W, pad_size, kernel_size,
                                                                max_displacement, stride1, stride2, grad_input1, grad_input2,
                                                                (hipStream_t)stream), who);
}

int pdepth_correlation_forward_f16(const void* input1, const void* input2, int32_t B, int32_t C, int32_t H, int32_t W, int32_t pad_size,
                                   int32_t kernel_size, int32_t max_displacement, int32_t stride1, int32_t stride2,
                                   int32_t corr_multiply, void* output, void* stream) {
    (void)corr_multiply;
    const char* who = "pdepth_correlation_forward_f16";
    if (!input1 || !input2 || !output) return fail(PDEPTH_E_ARG, "%s: null pointer", who);
    int oH, oW;
    if (int rc = check_corr(who, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oH, &oW)) return rc;
    return launched(pdepth::launch_correlation_general_forward(input1, input2, 1, B, C, H, W, pad_size, kernel_size, max_displacement,
                                                               stride1, stride2, output, (hipStream_t)stream), who);
}

int pdepth_correlation_backward_f16(const void* input1, const void* input2, const void* grad_output, int32_t B, int32_t C, int32_t H,
                                    int32_t W, int32_t pad_size, int32_t kernel_size, int32_t max_displacement, int32_t stride1,
                                    int32_t stride2, int32_t corr_multiply, void* grad_input1, void* grad_input2, void* stream) {
    (void)corr_multiply;
    const char* who = "pdepth_correlation_backward_f16";
    if (!input1 || !input2 || !grad_output || (!grad_input1 && !grad_input2)) return fail(PDEPTH_E_ARG, "%s: null pointer", who);
    int oH, oW;
    if (int rc = check_corr(who, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &oH, &oW)) return rc;
    return launched(pdepth::launch_correlation_general_backward(input1, input2, grad_output, 1, B, C, H, W, pad_size, kernel_size,
                                                                max_displacement, stride1, stride2, grad_input1, grad_input2,
                                                                (hipStream_t)stream), who);
}

int pdepth_inverse_warp_f32(const float* img, const float* depth, const float* Kinv, const float* proj, int32_t B,
                            int32_t C, int32_t H, int32_t W, int32_t mode, float* out, uint8_t* valid, void* stream) {
    if (!img || !depth || !Kinv || !proj || !out) return fail(PDEPTH_E_ARG, "pdepth_inverse_warp_f32: null pointer");
    if (B <= 0 || C <= 0 || H <= 1 || W <= 1) return fail(PDEPTH_E_ARG, "pdepth_inverse_warp_f32: bad dimension");
    if (mode != PDEPTH_SAMPLE_BILINEAR && mode != PDEPTH_SAMPLE_NEAREST) return fail(PDEPTH_E_ARG, "pdepth_inverse_warp_f32: unknown mode %d", mode);
    return launched(pdepth::launch_inverse_warp(img, depth, Kinv, proj, B, C, H, W, mode, out, valid, (hipStream_t)stream),
                    "pdepth_inverse_warp_f32");
}

int pdepth_inverse_warp_backward_f32(const float* img, const float* depth, const float* Kinv, const float* proj,
                                     const float* grad_out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t mode,
                                     float* grad_img, float* grad_point, void* stream) {
    if (!img || !depth || !Kinv || !proj || !grad_out || (!grad_img && !grad_point))
        return fail(PDEPTH_E_ARG, "pdepth_inverse_warp_backward_f32: null pointer");
    if (B <= 0 || C <= 0 || H <= 1 || W <= 1) return fail(PDEPTH_E_ARG, "pdepth_inverse_warp_backward_f32: bad dimension");
    if (mode != PDEPTH_SAMPLE_BILINEAR && mode != PDEPTH_SAMPLE_NEAREST)
        return fail(PDEPTH_E_ARG, "pdepth_inverse_warp_backward_f32: unknown mode %d", mode);
    return launched(pdepth::launch_inverse_warp_backward(img, depth, Kinv, proj, grad_out, B, C, H, W, mode, grad_img,
                                                         grad_point, (hipStream_t)stream), "pdepth_inverse_warp_backward_f32");
}

}  // extern "C"
