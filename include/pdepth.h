/*
 * pdepth.h -- C ABI of the MI355X (gfx950) plane-sweep / DPV hot path.
 *
 * Drop-in boundary for soulslicer/probabilistic-depth.  The reference has no FFI on this
 * path: its boundary is a set of Python functions built from ATen ops.  Every entry point
 * below replaces one of them (reference file:line cited per function); the Python host
 * package binds this header with ctypes (probabilistic-depth_amd/_native.py) and
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless a stride argument says
 *     otherwise; strides are in ELEMENTS (floats);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); launches are
 *     asynchronous, no host synchronisation, no allocation -- callers own all buffers
 *     (mirrors the reference's native op: models/correlation_package/correlation_cuda.cc:36-42,76);
 *   - return value 0 = launched, non-zero = PDEPTH_E_* and nothing was launched; the text
 *     is available from pdepth_last_error() (thread local).  Mirrors the reference's
 *     "kernel launcher returns 0/1, wrapper raises" convention
 *     (models/correlation_package/correlation_cuda_kernel.cu:383-392, correlation_cuda.cc:81-83);
 *   - the sweep kernels order their tiles per XCD assuming the SPX partition mode of the MI355X
 *     (workgroup i is dispatched to XCD i mod 8, each XCD with its own L2); under another
 *     partition mode the results are the same and only L2 locality is lost;
 *   - arithmetic: inputs, outputs, sample positions, bilinear weights and every sum are fp32 (SURVEY.md section 8:
 *     d_candi is float64 on the host and cast to fp32 at use, warping/homography.py:115, utils/img_utils.py:58).
 *     ONE step of the default sweep kernel (PDEPTH_ALGO_AUTO / _DIST, L2 metric) is not an fp32 instruction: the channel
 *     contraction <s, r> of the centred features runs on the matrix pipe as products of fp16 high / low PAIRS
 *     (h = fp16(x'), l = fp16(x' - h): 22 bits per feature, products exact, fp32 accumulation;
 *     v_mfma_f32_16x16x32_f16) -- measured as accurate as the fp32 matrix instruction.
 *     BASELINE.json's north star describes the path as "no MFMA (a gather/reduce)": the gather and the reductions are
 *     vector code here too, the contraction over the channels is not (DESIGN.md section 1 says why).  A caller who
 *     wants the reference's own operation order end to end selects PDEPTH_ALGO_DIRECT.
 *   - values that are not numbers propagate as in the reference: a NaN / inf feature or depth candidate makes the costs
 *     it enters, and with them the pixel's log-DPV and depth, non-finite.  A finite feature beyond the fp16 range of the
 *     default kernel's scaled layout (several thousand times the sampled maximum of its batch item): pdepth_sweep_dpv_f32
 *     evaluates that batch item with the gather kernel on the NCHW tensor (right numbers), the packed entry -- which has
 *     no NCHW tensor -- makes the item's outputs NaN: never a clamped number.
 *   - routing (PDEPTH_ALGO_AUTO / _DIST through pdepth_sweep_cost_f32 / pdepth_sweep_dpv_f32): a batch item whose costs are
 *     so large against the candidates' range that two fp32 evaluations agree to 1e-4 m only if they round alike
 *     (V (2 sum_c var_c + |mu|^2) / sigma * (d_max - d_min) * 2^-23 > 4e-4; the headline workload: 5.6e-5), or whose
 *     features carry trends the centring cannot remove, is evaluated by the gather kernel (the reference's operation
 *     order, PDEPTH_ALGO_DIRECT's kernel) inside the same call: ONE more launch, whose blocks leave at once when no item
 *     is flagged.  The packed entry of the default kernel evaluates every item in the distance form (it has no fp32
 *     features to fall back to); the LDS-tiled kernel (L1, C > 72, D > 128, forced selectors) routes on both entries.
 */
#ifndef PDEPTH_H_
#define PDEPTH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDEPTH_ABI_VERSION 6   /* 5: PDEPTH_ALGO_DIST (what AUTO runs), pdepth_sweep_source_layout, layout tag in the workspace;
                                * 6: the distance-form layout is 320 bytes per texel at C = 67 (336 in v5): a workspace packed by a v5
                                *    library must be re-packed; PDEPTH_ALGO_CORR answers in lab builds only */

enum {
    PDEPTH_OK = 0,
    PDEPTH_E_ARG = 1,       /* bad shape / null pointer / unsupported value      */
    PDEPTH_E_LAUNCH = 2,    /* hipGetLastError() after launch was not success   */
    PDEPTH_E_WORKSPACE = 3  /* workspace too small for the requested algorithm  */
};

/* feature distance, warping/homography.py:128-133 ('L2' | 'L1', anything else raises) */
enum { PDEPTH_METRIC_L2 = 0, PDEPTH_METRIC_L1 = 1 };

/* Rounding of the host BLAS whose sgemm/sgemv the reference's CPU path calls for K@R, K@t and
 * (K@R)@rays (warping/homography.py:119-121).  One ulp of those terms moves the L2 cost by ~1e-3,
 * so "identical to the reference CPU path" depends on which MKL code path the host takes:
 *   FMA      (MKL on Intel AVX-512/AVX2): sgemm = fma chain k=0,1,2 (first product rounded alone);
 *                                         sgemv (K@t) = (p1 + p2) + p0, products rounded separately
 *   SEPARATE (MKL on AMD EPYC)          : sgemm and sgemv = (p0 + p1) + p2, products rounded separately
 * The Python host probes torch's CPU matmul once and passes the matching mode
 * (probabilistic-depth_amd/_native.py host_blas_mode()). */
enum { PDEPTH_BLAS_FMA = 0, PDEPTH_BLAS_SEPARATE = 1 };

/* algorithm selector for the sweep kernels (pdepth_sample_coords_f32 reports the positions of the same
 * selector: AUTO = explicit shared-reciprocal fma divide chain, DIRECT = compiler IEEE divides; the two are
 * bit-identical for every position within reach of the image, tests/test_hip_parity.py) */
enum {
    PDEPTH_ALGO_AUTO = 0,    /* fastest algorithm valid for the given geometry          */
    PDEPTH_ALGO_DIRECT = 1,  /* per-plane bilinear gather, reference op order (any pose) */
    /* implementation selectors (parity tests, A/B timing): what AUTO may pick, forced.  Same workspace as AUTO. */
    PDEPTH_ALGO_TILED_1 = 2, /* LDS-tiled band kernel, one 16x4 tile per block                          */
    PDEPTH_ALGO_TILED_2 = 3, /* LDS-tiled band kernel, two tiles per block (D <= 64)                    */
    PDEPTH_ALGO_CELLS = 4,   /* lab builds only (make LAB=1): cell-list kernels of round 2 (L2, D <= 128)           */
    PDEPTH_ALGO_MFMA = 5,    /* lab builds only: matrix-pipe kernel of round 3 (L2, D <= 128, C <= 72)              */
    PDEPTH_ALGO_CORR = 6,    /* LAB BUILDS ONLY since ABI 6 (PDEPTH_E_ARG otherwise): correlation form on mean-centred features, fp32 matrix instructions (L2 metric, D <= 128,
                                C <= 72; other inputs: PDEPTH_E_ARG): the default of ABI 4, kept as an independent check */
    PDEPTH_ALGO_DIST = 7     /* distance form sum_t w_t |s_t - r|^2 - Q on fp16 high / low parts, matrix pipe (L2 metric,
                                D <= 128, C <= 72, V <= 8; other inputs: PDEPTH_E_ARG): what AUTO runs on those shapes  */
};

/* staging layouts of the packed source (pdepth_sweep_source_layout) */
enum {
    PDEPTH_LAYOUT_NONE = 0,        /* the descriptor does not run on a packed source                                   */
    PDEPTH_LAYOUT_C4 = 1,          /* channel-group-planar float4 + Gram planes (LDS-tiled kernel)                     */
    PDEPTH_LAYOUT_C4_CENTRED = 2,  /* the same on mean-centred features (PDEPTH_ALGO_CORR)                             */
    PDEPTH_LAYOUT_DIST16 = 3       /* fp16 high / low planes in matrix-operand order + neighbour differences, ring of
                                      zero-feature texels (PDEPTH_ALGO_DIST; csrc/dist_layout.hpp)                     */
};

/* Geometry + layout of one batched sweep call. */
typedef struct pdepth_sweep_desc {
    int32_t B;           /* batch items (depth volumes)                                 */
    int32_t V;           /* source views per item (reference view excluded)             */
    int32_t C;           /* feature channels (67 = 64 learned + 3 rgb, models.py:518-520)*/
    int32_t D;           /* depth planes                                                */
    int32_t H, W;        /* sweep resolution                                            */
    int32_t metric;      /* PDEPTH_METRIC_*                                             */
    int32_t algo;        /* PDEPTH_ALGO_*                                               */
    int32_t blas_mode;   /* PDEPTH_BLAS_* : rounding of K@R, K@t, (K@R)@rays            */
    float sigma;         /* costV_sigma (cfg.var.sigma_soft_max, 10.0)                  */
    int64_t ref_bstride; /* elements between ref[b] and ref[b+1]   (>= C*H*W)           */
    int64_t src_bstride; /* elements between src[b,0] and src[b+1,0]                    */
    int64_t src_vstride; /* elements between src[b,v] and src[b,v+1] (>= C*H*W)         */
} pdepth_sweep_desc;

/* Camera of one batched sweep call (all device pointers). */
typedef struct pdepth_camera {
    const float *K;     /* [B,3,3]   intrinsic_M_cuda             (models.py:537)        */
    const float *R;     /* [B,V,3,3] src_cam_poses[b,v,:3,:3]     (models.py:530)        */
    const float *t;     /* [B,V,3]   src_cam_poses[b,v,:3,3]      (models.py:531)        */
    const float *rays;  /* [B,3,H*W] unit_ray_array_2D, col=y*W+x (models.py:539)        */
    const float *cxcy;  /* [B,2]     float32(intrinsic_M[0,2]), float32(intrinsic_M[1,2])
                                     (warping/homography.py:194)                         */
} pdepth_camera;

int pdepth_abi_version(void);
const char *pdepth_last_error(void);

/*
 * Plane-sweep cost volume.  Replaces est_swp_volume_v4 (warping/homography.py:98-135) +
 * _back_warp_homo_parallel (:170-198) + img_dis_L2_pard/img_dis_L1_pard (:80-86), batched
 * over the Python loop at models/models.py:528-550.
 *   ref  [B,C,H,W] (ref_bstride), src [B,V,C,H,W] (src_bstride/src_vstride), d_candi [D]
 *   cost [B,D,H,W] contiguous, fully overwritten:
 *   cost[b,k,y,x] = sum_v ( sum_c dist(warp_{v,k}(src)[c,y,x], ref[c,y,x]) / sigma )
 */
int pdepth_sweep_cost_f32(const pdepth_sweep_desc *desc, const pdepth_camera *cam,
                          const float *ref, const float *src, const float *d_candi,
                          float *cost, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Fused sweep + DPV reduction: cost -> log_softmax over D -> E[d].  Replaces the chain
 * est_swp_volume_v4 -> F.log_softmax(dim=1) (models/packnet.py:380-394) ->
 * dpv_to_depthmap(BV_log=True) (utils/img_utils.py:52-61) without writing the cost volume.
 *   cost  [B,D,H,W] or NULL     (un-normalised cost, as pdepth_sweep_cost_f32)
 *   logp  [B,D,H,W] or NULL     (log DPV)
 *   depth [B,H,W]   or NULL     (expected depth)
 * At least one output must be non-NULL.
 */
int pdepth_sweep_dpv_f32(const pdepth_sweep_desc *desc, const pdepth_camera *cam,
                         const float *ref, const float *src, const float *d_candi,
                         float *cost, float *logp, float *depth,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same fused sweep for a caller that keeps its source features in the kernels' staging layout -- e.g. the epilogue
 * of the feature encoder (models/models.py:518-534 is where the reference concatenates the 64 learned channels with
 * the RGB thumbnail) -- so that the re-layout is paid once per frame instead of once per sweep call (it is a fifth of the
 * time of pdepth_sweep_dpv_f32 on the headline shape):
 *   pdepth_pack_source_f32      : src [B,V,C,H,W] (strides from desc) -> workspace, in the layout the sweep kernel that
 *                                 `desc` selects reads (pdepth_sweep_source_layout(desc)):
 *                                   PDEPTH_LAYOUT_DIST16 (L2, C <= 72, D <= 128, V <= 8: what AUTO runs): per view an
 *                                     (H + 2) x wp(W) image (a ring of zero-feature texels; rows padded to a multiple of 8
 *                                     texels), planes of 16 bytes per texel: the centred, power-of-two scaled features as
 *                                     fp16 high and low parts in matrix-operand order, |x'|^2 as three fp16 pieces, and an
 *                                     fp32 record of the five squared neighbour differences of the cell (20 planes = 320
 *                                     bytes per texel at C = 67), stored texel-group-major: the planes of four consecutive
 *                                     texels of a row lie together (1 280 bytes at C = 67; csrc/dist_layout.hpp);
 *                                   PDEPTH_LAYOUT_C4 (L1 metric, C > 72, D > 128: the LDS-tiled kernel): float4 texels of 4
 *                                     channels, planes [ceil(C/4)][H][W], + 2 Gram planes;
 *   pdepth_sweep_dpv_packed_f32 : the sweep on a workspace packed by that call for a desc with the same B, V, C, H, W and
 *                                 the SAME pdepth_sweep_source_layout() -- a layout is tied to its kernel family: the
 *                                 metric and the algo selector may only change within it (any cameras, depth candidates,
 *                                 sigma, reference features).  Outputs as pdepth_sweep_dpv_f32.  A foreign layout is
 *                                 detected on the device (tag in the workspace): every output is filled with NaN.
 *                                 The packed entry of the default kernel is ONE launch.
 * PDEPTH_E_ARG if the shape does not run on a packed source (then use pdepth_sweep_dpv_f32).
 * Statistics.  The default kernel centres and scales with per-channel constants of the batch item, estimated from sampled
 * rows: pdepth_sweep_dpv_f32 and pdepth_pack_views_f32 sample every source view AND the reference view,
 * pdepth_pack_source_f32 (which is not given the reference) the source views.
 */
int pdepth_pack_source_f32(const pdepth_sweep_desc *desc, const float *src, void *workspace,
                           size_t workspace_bytes, void *stream);
/*
 * The encoder epilogue, fused with that re-layout: replaces
 *     feat_imgs_all = torch.cat((feat_imgs, F.avg_pool2d(rgb, dw_rate)), dim=1)     models/models.py:518-520, packnet.py:355-357
 *     feat_img_ref = feat_imgs_all[i, -1], feat_imgs_src = feat_imgs_all[i, :-1]    models/models.py:530-534
 * and pdepth_pack_source_f32 with ONE pass over the encoder output:
 *   feat [B*(V+1), C-3, H, W] (the encoder's feature maps, view V of every item = the reference view),
 *   rgb  [B*(V+1), 3, H*pool_rate, W*pool_rate] (the frames; pooled like F.avg_pool2d(kernel = stride = pool_rate)),
 *   -> workspace: the V source views of every item in the staging layout (for pdepth_sweep_dpv_packed_f32, same desc),
 *   -> ref_out [B, C, H, W]: the reference view's features, NCHW (what that entry takes as `ref`).
 * desc: B, V (source views), C (= encoder channels + 3), D, H, W, algo = PDEPTH_ALGO_AUTO.
 */
int pdepth_pack_views_f32(const pdepth_sweep_desc *desc, const float *feat, const float *rgb, int32_t pool_rate,
                          float *ref_out, void *workspace, size_t workspace_bytes, void *stream);
int pdepth_sweep_dpv_packed_f32(const pdepth_sweep_desc *desc, const pdepth_camera *cam, const float *ref,
                                const float *d_candi, float *cost, float *logp, float *depth,
                                void *workspace, size_t workspace_bytes, void *stream);

/* Bytes of scratch the two sweep entry points need for `desc` (0 for ALGO_DIRECT); the workspace
 * must be 256-byte aligned.  ALGO_AUTO without it returns PDEPTH_E_WORKSPACE.  It holds two ints per 16x4 tile (tile flags
 * and the list of left-over tiles of the LDS-tiled kernel; the default kernel keeps its eight per-XCD queue counters in the
 * first, 256 bytes apart), 64 counter / tag ints, the packed source views -- sized for the LARGEST of the layouts of the shape, so any pack fits any sweep:
 * max(B*V*(8*nchk(C)+4)*(H+2)*wp(W)*16 + 256*B*V, B*V*(ceil(C/4)+2)*H*W*16) bytes with nchk(C) = 0 | 1 | 2 chunks of 32
 * channels and wp(W) = W + 2 rounded up to a multiple of 8 -- and 496 floats of channel statistics per batch item.  Size it
 * once per shape and reuse it.  A call rewrites all of it: do not share one workspace between calls that may run
 * concurrently (different streams). */
size_t pdepth_sweep_workspace_bytes(const pdepth_sweep_desc *desc);

/* 1 if the packing entry points (pdepth_pack_source_f32, pdepth_pack_views_f32) subtract the channel means from the
 * packed source for `desc` -- i.e. the sweep it selects is the distance-form kernel (PDEPTH_ALGO_DIST, directly or through
 * PDEPTH_ALGO_AUTO; lab builds: also PDEPTH_ALGO_CORR) --, else 0.  A packed workspace must be swept with a descriptor for which this answer is the same (the centred and the
 * plain layout differ; the library cannot tell them apart from the host).  No reference counterpart: the reference
 * never re-lays its features (warping/homography.py:123-129 works on the NCHW tensors). */
int pdepth_sweep_centres_source(const pdepth_sweep_desc *desc);

/* Which staging layout (PDEPTH_LAYOUT_*) the packing entry points write for `desc`, i.e. which kernel family the sweep it
 * selects belongs to.  A packed workspace must be swept with a descriptor for which this answer is the same.  The pack
 * kernels also write the answer into the workspace, and a sweep kernel that finds another family's tag there fills its
 * outputs with NaN instead of returning numbers computed from the wrong bytes.  No reference counterpart (as above). */
int pdepth_sweep_source_layout(const pdepth_sweep_desc *desc);

/*
 * DPV reduction: logits [B,D,H,W] -> logp [B,D,H,W] (may alias logits, may be NULL) and
 * depth [B,H,W] (may be NULL).  Replaces F.log_softmax(x, dim=1) (models/models.py:560,637,
 * 694, :351) followed by dpv_to_depthmap(., BV_log=True) (utils/img_utils.py:52-61;
 * trainer/default_trainer.py:229-233) in one pass over the logits.
 */
int pdepth_dpv_reduce_f32(const float *logits, const float *d_candi, int32_t B, int32_t D,
                          int32_t H, int32_t W, float *logp, float *depth, void *stream);

/*
 * The same single pass with optional extras (any output may be NULL, not all): replaces, next to the lines above,
 *   addend       [B,D,H,W] or NULL : x = logits + addend before the softmax -- the feedback update
 *                                    F.log_softmax(BV_cur + BV_resi, dim=1)            (models/models.py:694)
 *   prob         [B,D,H,W]         : exp(logp), the decoder's input torch.exp(BV_cur_upd) (models/models.py:697, :651)
 *   variance     [B,H,W]           : sum_k (d_k - E[d])^2 p_k, the inline lines of the evaluation loop
 *                                                                        (trainer/default_trainer.py:333-336)
 *   logp_quarter [B,D,H/4,W/4]     : F.interpolate(logp, scale_factor=0.25, mode='nearest'), the prev_output of the
 *                                    next frame                          (trainer/default_trainer.py:221)
 * logp may alias logits; no other aliasing.
 */
int pdepth_dpv_reduce_ex_f32(const float *logits, const float *addend, const float *d_candi, int32_t B, int32_t D,
                             int32_t H, int32_t W, float *logp, float *prob, float *depth, float *variance,
                             float *logp_quarter, void *stream);

/*
 * Expectation only: depth[b,y,x] = sum_k d_k * (bv_log ? exp(dpv[b,k,y,x]) : dpv[b,k,y,x]).
 * Replaces dpv_to_depthmap (utils/img_utils.py:52-61), batched over B.
 */
int pdepth_dpv_expect_f32(const float *dpv, const float *d_candi, int32_t B, int32_t D,
                          int32_t H, int32_t W, int32_t bv_log, float *depth, void *stream);

/*
 * Mean and variance of the depth distribution: z = bv_log ? exp(dpv) : dpv, mean = sum_k d_k z_k,
 * variance[b,y,x] = sum_k (d_k - mean)^2 z_k.  Replaces the per-item torch ops of the evaluation loop
 * (trainer/default_trainer.py:333-336), batched over B.  mean may be NULL.
 */
int pdepth_dpv_moments_f32(const float *dpv, const float *d_candi, int32_t B, int32_t D,
                           int32_t H, int32_t W, int32_t bv_log, float *mean, float *variance, void *stream);

/*
 * Diagonal feature warp: out[b,v,i,y,x] = bilinear(src[b,v,i,:,:]) sampled with the
 * plane-i homography of view v.  Replaces warp_feature (warping/homography.py:137-168),
 * which warps all D x C planes and keeps [i,i]; requires C == D.
 *   desc->C must equal desc->D; desc->V counts ALL views passed (reference view included,
 *   models/models.py:616-617); src [B,V,D,H,W] via src_bstride/src_vstride; out contiguous.
 */
int pdepth_warp_feature_f32(const pdepth_sweep_desc *desc, const pdepth_camera *cam,
                            const float *src, const float *d_candi, float *out, void *stream);

/*
 * Sampling coordinates only (diagnostic, used by the parity tests to pin the coordinate
 * pipeline bit-for-bit): ix, iy [B,V,D,H,W] = un-normalised pixel coordinates handed to the
 * bilinear sampler (warping/homography.py:185-196 + ATen grid_sampler unnormalize).
 */
int pdepth_sample_coords_f32(const pdepth_sweep_desc *desc, const pdepth_camera *cam,
                             const float *d_candi, float *ix, float *iy, void *stream);

/*
 * Uncertainty-field collapse: replaces gen_ufield (utils/img_utils.py:268-358; called through compute_unc_field
 * :178-181 by the evaluation loop, trainer/default_trainer.py:243-244), batched over B.
 *   dpv [B,D,H,W] (log-DPV if bv_log, else probabilities), intr [B,3,3] full-resolution intrinsics, mask [B,H,W] or
 *   NULL (validity of the ground-truth volume).  unc_ang = rows the volume is shifted by (cfgx["unc_ang"]; 0 = no
 *   shift), [z_start, z_end] = height band (cfgx: unc_shift, unc_shift + unc_span), min_depth / quash as in the
 *   reference's branches (:269-290: cfgx and ILIM 3 / 1, KITTI 0 / 0), oob_depth = the depth the reference assigns to
 *   rows shifted in from outside (sum_k d_k for a log-DPV -- exp of the zero padding -- else 0).
 *   plane [B,D,W] = per column the mean depth distribution of the pixels in the band (NaN for columns without one, as
 *   in the reference), depth_zero [B,H,W] = E[d] masked to those pixels.  The min/max normalisation (:353-355) is a
 *   [D,W] post-process left to the host.  Workspace: pdepth_ufield_workspace_bytes (two [B,H,W] maps + [B,W]).
 */
size_t pdepth_ufield_workspace_bytes(int32_t B, int32_t H, int32_t W);
int pdepth_ufield_f32(const float *dpv, const float *d_candi, const float *intr, const float *mask, int32_t B,
                      int32_t D, int32_t H, int32_t W, int32_t bv_log, float unc_ang, float z_start, float z_end,
                      float min_depth, int32_t quash, float oob_depth, float *plane, float *depth_zero,
                      void *workspace, size_t workspace_bytes, void *stream);

/*
 * DPV Bayesian fusion of the upsample mode: replaces gen_dpv_withmask (utils/img_utils.py:360-375, with
 * gen_soft_label_torch :31-47 and gen_uniform :49-50) followed by the fuse / renormalise / clamp / log
 * lines of BaseModel.forward_int (models/models.py:666-672).
 *   logp [B,D,H,W] log-DPV of the network, dmaps [B,H,W] sparse depth, masks [B,H,W] validity (channel 0
 *   of the reference's [B,1,H,W]), var = 0.3 in the reference, eps = torch.finfo(float).eps.
 *   fused [B,D,H,W] (clamped probabilities) and logfused [B,D,H,W]; either may be NULL, not both.
 */
int pdepth_dpv_fuse_f32(const float *logp, const float *dmaps, const float *masks, const float *d_candi,
                        int32_t B, int32_t D, int32_t H, int32_t W, float var, float eps, float *fused,
                        float *logfused, void *stream);

/*
 * The reference's native correlation operator, forward and backward: replaces correlation_forward_cuda /
 * correlation_backward_cuda (models/correlation_package/correlation_cuda.cc:10-87, :89-167; kernels
 * correlation_cuda_kernel.cu:41-114, :116-300) with the reference's argument list (the rbot1 / rbot2 scratch tensors --
 * zero-padded NHWC repacks -- are not needed).  Every configuration the reference's kernel defines:
 *     kr = (kernel_size-1)/2, dr = max_displacement/stride2, ds = 2 dr + 1
 *     output [B, ds*ds, oH, oW], oH = ceil((H + 2 pad_size - 2 (kr + max_displacement)) / stride1) (correlation_cuda.cc:24-33;
 *     pdepth_correlation_output_size), channel (tj+dr)*ds + (ti+dr),
 *     out = 1/(k*k*C) sum_{j,i in [-kr,kr]} sum_c in1p[y1+j, x1+i] * in2p[y1 + tj*stride2 + j, x1 + ti*stride2 + i],
 *     (y1, x1) = (oy, ox) * stride1 + max_displacement in the coordinates of the zero-padded inputs.
 * kernel_size must be odd and (kernel_size-1)/2 <= max_displacement mod stride2 (beyond that the reference's kernel reads
 * outside its padded buffers); corr_multiply is accepted and ignored like in the reference kernel.  The configuration the
 * reference instantiates (kernel 1, stride1 1, pad_size = max_displacement <= 4*stride2: pwclite.py:123-125) runs
 * LDS-tiled fp32 kernels; everything else, and the fp16 entries (fp16 tensors, fp32 accumulation:
 * AT_DISPATCH_FLOATING_TYPES_AND_HALF at correlation_cuda_kernel.cu:352-369), a general gather kernel.
 *   forward : input1, input2 [B,C,H,W] -> output;   backward: grad_output [B,ds*ds,oH,oW] -> grad_input1, grad_input2
 *   [B,C,H,W] (either may be NULL).
 */
int pdepth_correlation_output_size(int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size, int32_t max_displacement,
                                   int32_t stride1, int32_t stride2, int32_t *out_channels, int32_t *out_height,
                                   int32_t *out_width);
int pdepth_correlation_forward_f32(const float *input1, const float *input2, int32_t B, int32_t C, int32_t H,
                                   int32_t W, int32_t pad_size, int32_t kernel_size, int32_t max_displacement,
                                   int32_t stride1, int32_t stride2, int32_t corr_multiply, float *output,
                                   void *stream);
int pdepth_correlation_backward_f32(const float *input1, const float *input2, const float *grad_output, int32_t B,
                                    int32_t C, int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size,
                                    int32_t max_displacement, int32_t stride1, int32_t stride2,
                                    int32_t corr_multiply, float *grad_input1, float *grad_input2, void *stream);
int pdepth_correlation_forward_f16(const void *input1, const void *input2, int32_t B, int32_t C, int32_t H, int32_t W,
                                   int32_t pad_size, int32_t kernel_size, int32_t max_displacement, int32_t stride1,
                                   int32_t stride2, int32_t corr_multiply, void *output, void *stream);
int pdepth_correlation_backward_f16(const void *input1, const void *input2, const void *grad_output, int32_t B,
                                    int32_t C, int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size,
                                    int32_t max_displacement, int32_t stride1, int32_t stride2, int32_t corr_multiply,
                                    void *grad_input1, void *grad_input2, void *stream);

/*
 * Depth-map driven inverse warp: replaces the per-pixel part of inverse_warp (utils/inverse_warp.py:174-210 with
 * pixel2cam :26-40 and cam2pixel :43-69), used by the training losses (losses/loss_blocks.py:116 with the default
 * 'bilinear', :151 with 'nearest').  The host supplies Kinv = intrinsics.inverse() [B,3,3] and
 * proj = intrinsics @ pose_mat [B,3,4] (:193-203); img [B,C,H,W], depth [B,H,W] ->
 * out [B,C,H,W] (zeros padding, grid_sample's default align_corners=False) and valid [B,H,W] as bytes
 * (|normalised coordinate| <= 1, :208), valid may be NULL.
 */
enum { PDEPTH_SAMPLE_BILINEAR = 0, PDEPTH_SAMPLE_NEAREST = 1 };
int pdepth_inverse_warp_f32(const float *img, const float *depth, const float *Kinv, const float *proj,
                            int32_t B, int32_t C, int32_t H, int32_t W, int32_t mode, float *out, uint8_t *valid,
                            void *stream);

/*
 * Backward of the same (what autograd does behind the reference's call, through F.grid_sample and the projection):
 *   grad_out [B,C,H,W] -> grad_img [B,C,H,W] (fully overwritten: cleared, then scattered with atomics like ATen's
 *   grid_sampler backward; may be NULL) and grad_point [B,3,H,W] = dL/d(X, Y, Z) of the projected point
 *   proj[:, :, :3] @ cam + proj[:, :, 3] of every pixel (may be NULL; zero in nearest mode).  The gradients of the
 *   depth map, the pose and the intrinsics follow from grad_point by the 3x3 / 3x4 products the host already owns
 *   (probabilistic-depth_amd/utils/inverse_warp.py does them in torch).
 */
int pdepth_inverse_warp_backward_f32(const float *img, const float *depth, const float *Kinv, const float *proj,
                                     const float *grad_out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t mode,
                                     float *grad_img, float *grad_point, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PDEPTH_H_ */
