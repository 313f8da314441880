// Internal launch interface between the C ABI (capi.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace pdepth {

// Flattened arguments of one sweep launch (device pointers, element strides).
struct SweepArgs {
    const float* ref;
    const float* src;
    const float* K;
    const float* R;
    const float* t;
    const float* rays;
    const float* cxcy;
    const float* d_candi;
    float* cost_out;   // [B,D,H,W] or nullptr
    float* logp_out;   // [B,D,H,W] or nullptr
    float* depth_out;  // [B,H,W]   or nullptr
    int B, V, C, D, H, W;
    int metric;
    int blas_mode;
    int fast_div;      // 1: shared-reciprocal divide chain (geometry.hpp), 0: compiler's IEEE divides
    float sigma;
    long long ref_bstride, src_bstride, src_vstride;
    // the source views in the sweep kernels' staging layout ([B*V][C/4 + 2][H][W] float4, sweep_tiled.hip), set by the
    // launchers of the tiled / cell-list paths: the gather kernel reads it for the tiles handed to it when src == nullptr
    // (packed-source entry: the caller no longer has the NCHW source)
    const void* packed_src;
    // device-side choice between two sweep kernels launched back to back (pick.hpp): 0 = none (the kernel runs), PICK_SKIP_IF_SET
    // = leave at once if queue[PICK_SLOT] != 0, PICK_RUN_IF_SET = leave at once if it is 0.  In the pre-pass: != 0 = compute it.
    int pick;
};
// conditioning above which a batch item is routed to the gather kernel (sweep_dist.hip: "Conditioning"; sweep_pack.hip)
#ifndef PDEPTH_COND_LIMIT
#define PDEPTH_COND_LIMIT 4.0e-4f
#endif
// ... for the LDS-tiled kernel, whose correlation-form plane group is noisier than the distance form (soak 9107: 1.04e-4 /
// 1.09e-4 m, scaled, at 2.0e-4 / 2.3e-4 of this measure on unit-variance features, V = 3)
#ifndef PDEPTH_COND_LIMIT_TILED
#define PDEPTH_COND_LIMIT_TILED 1.5e-4f
#endif
constexpr int PICK_SLOT = 50, PICK_MFMA = 1;          // workspace int behind the tile flags (cleared with them)
// more of the 64 workspace ints behind the tile flags:
constexpr int NONCENTRED_SLOT = 51;        // set by the pre-pass of a NOT centred source whose channel offsets exceed the spread (sweep_pack.hip)
constexpr int CORR_DONE_SLOT = 52;         // sweep_corr.hip: workgroups that have left (the last one zeroes the queue counters)
constexpr int CORR_DIRECT_SLOT = 53;       // ... pixel blocks evaluated directly, this call so far / of the last finished call
constexpr int CORR_DIRECT_LAST_SLOT = 54;
constexpr int CORR_PACK_TIMEOUT_SLOT = 55;   // ... workgroups that gave up waiting for the in-kernel pack (never, unless the counters were corrupted)
// channel statistics of the source (workspace tail, sweep_pack.hip): per batch item mu[c] at +0, var[c] at +STATS_VAR, the
// squared offset that was NOT subtracted at +STATS_OFF, the largest sampled |x| at +STATS_AMAX, half the mean squared
// difference of samples STATS_LAG_PX texels apart at +STATS_LAG (the spread of a channel at the distance of a plane sweep:
// equal to var[c] for white features, smaller for smooth ones), and STATS_NFLAG ints at +STATS_FLAGS: [1] != 0 = the item was left to the gather kernel (sweep_dist.hip: routing); [0] != 0 = a feature
// of the item did not fit the fp16 range of the distance-form layout (pack_dist.hip)
constexpr int STATS_VAR = 80, STATS_OFF = 160, STATS_AMAX = 240, STATS_LAG = 320, STATS_FLAGS = 400, STATS_NFLAG = 16, STATS_READY = 416, STATS_STRIDE = 496;
// STATS_READY: one int per channel = the tag of the launch whose statistics the row holds (pack_dist.hip: the pack kernel computes the
// statistics in its first workgroups and every workgroup waits for its batch item's tags -- one pre-pass launch)
constexpr int STATS_LAG_PX = 16;
// which staging layout the packed-source region holds (written by the pack kernels, checked by the sweep kernels: a sweep on
// another family's layout fills its outputs with NaN instead of returning numbers computed from the wrong bytes)
constexpr int LAYOUT_SLOT = 56;
constexpr int LAYOUT_C4 = 1, LAYOUT_C4_CENTRED = 2, LAYOUT_DIST16 = 3;   // (0: nothing packed yet)
constexpr int DIST_DONE_SLOT = 57, DIST_DIRECT_SLOT = 58, DIST_DIRECT_LAST_SLOT = 59;   // sweep_dist.hip: as the CORR_ slots
// DIST_DIRECT_LAST_SLOT holds (nonce << 20) | count, DIST_NONCE_SLOT the nonce of the last call: a count whose nonce is another
// call's reads as 0 (where every workgroup runs one item there is no counter of finished workgroups to reset anything by: 2 048
// returning atomics on one address were a quarter of such a launch)
constexpr int DIST_NONCE_SLOT = 60;
constexpr int PICK_SKIP_IF_SET = 1, PICK_RUN_IF_SET = 2;

// First statement of a sweep kernel on a packed source: does the workspace hold the layout this kernel reads?  If not (a C
// caller swept a workspace packed for another kernel family: include/pdepth.h, pdepth_sweep_source_layout) every output of
// the call is filled with NaN by the whole grid and the kernel leaves: loud numbers instead of costs computed from the
// wrong bytes.  (The packing entry points write the tag; the Python binding refuses the mismatch before it gets here.)
__device__ __forceinline__ bool poison_on_foreign_layout(const SweepArgs& a, const int* __restrict__ queue, int expected) {
    if (queue[LAYOUT_SLOT] == expected) return false;
    const float nan = __builtin_nanf("");
    const size_t hw = (size_t)a.H * a.W, nvol = (size_t)a.B * a.D * hw, nmap = (size_t)a.B * hw;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < nvol; i += step) {
        if (a.cost_out) a.cost_out[i] = nan;
        if (a.logp_out) a.logp_out[i] = nan;
    }
    if (a.depth_out)
        for (size_t i = i0; i < nmap; i += step) a.depth_out[i] = nan;
    return true;
}
constexpr int PH_PRE = 1, PH_KERNEL = 2, PH_GATHER = 4, PH_ALL = 7;   // phases of a sweep launcher: pre-pass / flag clear, kernel, gather

// sweep_direct.hip
hipError_t launch_sweep_direct(const SweepArgs& a, hipStream_t stream);
// Workspace counter (an int of the 64 queue ints behind the tile flags) of the tiles handed to the gather kernel:
// every writer of a gather flag increments it, the gather kernel's blocks leave at once while it is zero.
constexpr int GATHER_COUNT_SLOT = 48;
hipError_t launch_sweep_direct_flagged(const SweepArgs& a, const int* tile_flags, const int* gather_count, int tiles_x,
                                       int tiles, hipStream_t stream, int flag_value = 1);
// ... the whole batch items b with item_flags[b * item_stride] != 0 (the distance-form kernel's routed items: STATS_FLAGS + 1)
hipError_t launch_sweep_direct_items(const SweepArgs& a, const int* item_flags, int item_stride, hipStream_t stream);  // runs the tiles whose flag == flag_value
int sweep_direct_max_planes(int C);

// sweep_tiled.hip
size_t sweep_tiled_workspace_bytes(int B, int V, int C, int H, int W);
int sweep_tiled_max_planes();
// packed_ready: the workspace already holds the packed source of exactly these views (pdepth_pack_source_f32)
hipError_t launch_sweep_tiled(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready = false, int phases = PH_ALL);     // picks a variant
hipError_t launch_sweep_tiled_n1(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready = false, int phases = PH_ALL);  // one 16x4 tile per block
hipError_t launch_sweep_tiled_n2(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready = false, int phases = PH_ALL);  // two tiles per block

// sweep_pack.hip: pre-pass of the packed-source kernels (channel statistics + packed source + Gram planes; clears flags and
// queue counters).  centre: subtract the channel means (sweep_corr.hip); else the plain layout (mu = 0)
hipError_t launch_pack_c4(const SweepArgs& a, void* workspace, hipStream_t stream, bool centre);
// ... of a call whose sweep kernel packs the source itself: statistics + workspace bookkeeping only
hipError_t launch_stats_only(const SweepArgs& a, void* workspace, hipStream_t stream);
bool sweep_ws_holds_pack_counters(int B, int H, int W);
int* sweep_ws_pack_counters(const SweepArgs& a, void* workspace);
hipError_t clear_sweep_flags(const SweepArgs& a, void* workspace, hipStream_t stream);
// encoder epilogue: cat(feat, avg_pool2d(rgb)) -> packed source views + NCHW reference view, in one pass (a.C = Cf + 3)
hipError_t launch_pack_views(const SweepArgs& a, const float* feat, const float* rgb, int rate, int img_h, int img_w, float* ref_out,
                             void* workspace, hipStream_t stream, bool centre);
int sweep_device_cus();
// workspace head shared by the packed-source kernels: tile flags (+ the 64 queue / counter ints behind them); its tail
size_t sweep_ws_flag_only_bytes(int B, int H, int W);
size_t sweep_ws_flag_bytes(int B, int H, int W);
size_t sweep_ws_stats_offset(int B, int V, int C, int H, int W);

// sweep_pack.hip: the channel statistics alone (mean-centring on), for the pack kernels of pack_dist.hip
hipError_t launch_feature_stats(const SweepArgs& a, float* stats, hipStream_t stream);
int sweep_resident_workgroups();   // 256-thread workgroups the device certainly holds at once (conservative)
hipError_t launch_view_stats(const SweepArgs& a, const float* feat, const float* rgb, int rate, int img_h, int img_w, float* stats, hipStream_t stream);
// pack_dist.hip: statistics + the source views in the distance-form kernel's layout (dist_layout.hpp)
// fuse_stats: the statistics inside the pack kernel (one launch) -- only where the sweep kernel follows in the same call: it clears the tags
hipError_t launch_pack_dist(const SweepArgs& a, void* workspace, hipStream_t stream, bool fuse_stats = false);
hipError_t launch_pack_views_dist(const SweepArgs& a, const float* feat, const float* rgb, int rate, int img_h, int img_w, float* ref_out,
                                  void* workspace, hipStream_t stream);
// sweep_dist.hip (L2 only): distance form sum_t w_t |s_t - r|^2 - Q on the matrix pipe (fp16 high / low parts), C <= 72, D <= 128
bool sweep_dist_supports(const SweepArgs& a);
hipError_t launch_sweep_dist(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready = false);

// sweep_corr.hip (L2 only): correlation form on mean-centred features, one workgroup per block of 16 pixels
bool sweep_corr_supports(const SweepArgs& a);
hipError_t launch_sweep_corr(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready = false);

// sweep_mfma.hip (L2 only; same workspace as the tiled kernel): the channel contraction on the matrix pipe
bool sweep_mfma_supports(const SweepArgs& a);
hipError_t launch_sweep_mfma(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready = false, int phases = PH_ALL);

// sweep_cells.hip (L2 only; same workspace as the tiled kernel)
int sweep_cells_max_planes();
hipError_t launch_sweep_cells(const SweepArgs& a, void* workspace, hipStream_t stream, bool packed_ready = false);

// sweep_cells_fast.hip: straight-line instantiation (D = 64 or 128), flags the tiles it leaves to the generic kernel
hipError_t launch_sweep_cells_fast(const SweepArgs& a, const float4* packed, int* flags, int* queue, int* redo_list,
                                   int tiles_x, int tiles, int n_cu, hipStream_t stream);

// dpv.hip
hipError_t launch_dpv_reduce(const float* logits, const float* d_candi, int B, int D, int H,
                             int W, float* logp, float* depth, hipStream_t stream);
hipError_t launch_dpv_reduce_ex(const float* logits, const float* addend, const float* d_candi, int B, int D, int H, int W,
                                float* logp, float* prob, float* depth, float* variance, float* quarter, hipStream_t stream);
hipError_t launch_dpv_expect(const float* dpv, const float* d_candi, int B, int D, int H, int W,
                             int bv_log, float* depth, hipStream_t stream);

// warp.hip
hipError_t launch_warp_feature(const SweepArgs& a, float* out, hipStream_t stream);
hipError_t launch_sample_coords(const SweepArgs& a, float* ix, float* iy, hipStream_t stream);

// ufield.hip
size_t ufield_workspace_bytes(int B, int H, int W);
hipError_t launch_ufield(const float* dpv, const float* d_candi, const float* intr, const float* mask, int B, int D, int H,
                         int W, int bv_log, float unc_ang, float zstart, float zend, float mind, int quash, float oob_depth,
                         float* plane, float* depth_zero, void* workspace, hipStream_t stream);

// extras.hip
hipError_t launch_dpv_moments(const float* dpv, const float* d_candi, int B, int D, int H, int W, int bv_log,
                              float* mean, float* var, hipStream_t stream);
hipError_t launch_dpv_fuse(const float* logp, const float* dmaps, const float* masks, const float* d_candi,
                           int B, int D, int H, int W, float var, float eps, float* fused, float* logfused,
                           hipStream_t stream);
hipError_t launch_correlation_forward(const float* x1, const float* x2, int B, int C, int H, int W, int radius,
                                      int stride2, float* out, hipStream_t stream);
int correlation_max_radius();
// correlation_general.hip: every configuration of the reference's kernel, fp32 or fp16 I/O (half != 0), fp32 accumulation
bool correlation_output_size(int H, int W, int pad, int k, int md, int s1, int* oH, int* oW);
hipError_t launch_correlation_general_forward(const void* x1, const void* x2, int half, int B, int C, int H, int W, int pad, int k, int md,
                                              int s1, int s2, void* out, hipStream_t stream);
hipError_t launch_correlation_general_backward(const void* x1, const void* x2, const void* go, int half, int B, int C, int H, int W, int pad,
                                               int k, int md, int s1, int s2, void* g1, void* g2, hipStream_t stream);
hipError_t launch_correlation_backward(const float* x1, const float* x2, const float* go, int B, int C, int H, int W,
                                       int radius, int stride2, float* g1, float* g2, hipStream_t stream);
hipError_t launch_inverse_warp(const float* img, const float* depth, const float* Kinv, const float* proj, int B,
                               int C, int H, int W, int mode, float* out, unsigned char* valid, hipStream_t stream);
hipError_t launch_inverse_warp_backward(const float* img, const float* depth, const float* Kinv, const float* proj,
                                        const float* grad_out, int B, int C, int H, int W, int mode, float* grad_img,
                                        float* grad_pc, hipStream_t stream);

}  // namespace pdepth
