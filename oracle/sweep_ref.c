/*
 * sweep_ref.c -- plain-C restatement of the plane-sweep + DPV path.  TEST INFRASTRUCTURE ONLY
 * (see oracle/ref_cpu.py for the rules: only tests/, smoke() and bench.py's cpu_baseline leg
 * may load it).
 *
 * Role: an independent "truth given the fp32 sample positions".  The sample coordinates are
 * evaluated in fp32 with exactly the rounding of the reference's CPU path (pinned in
 * tests/test_coords_pin.py); interpolation, distance, log-softmax and expectation are then
 * carried in double.  Comparing the HIP kernels and the torch oracle against it separates
 * "rounding noise of an fp32 evaluation order" from real errors.
 *
 * Follows (reference checkout):
 *   warping/homography.py:119-121  term1 = K t, term2 = (K R) rays
 *   warping/homography.py:185-196  P = term1 + term2 d; P/(Pz+1e-10); (u-cx)/cx
 *   ATen grid_sampler_2d (bilinear, zeros, align_corners=False)
 *   warping/homography.py:80-86    L2 / L1 distance, :129 division by sigma per view
 *   models/packnet.py:394          log_softmax over D
 *   utils/img_utils.py:52-61       E[d]
 */
#include <math.h>
#include <stddef.h>

/* 3-term dot product with the rounding of the host BLAS (include/pdepth.h PDEPTH_BLAS_*):
 * separate = 0: fma chain (MKL on Intel); separate = 1: (p0 + p1) + p2 (MKL on AMD). */
static float dot3_blas(int separate, float a0, float b0, float a1, float b1, float a2, float b2) {
    volatile float p0 = a0 * b0;
    if (separate) {
        volatile float p1 = a1 * b1, p2 = a2 * b2;
        volatile float s = p0 + p1;
        return s + p2;
    }
    return fmaf(a2, b2, fmaf(a1, b1, p0));
}

static void view_xform(int separate, const float *K, const float *R, const float *t, float *kr, float *kt) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            kr[i * 3 + j] = dot3_blas(separate, K[i * 3 + 0], R[0 * 3 + j], K[i * 3 + 1], R[1 * 3 + j],
                                      K[i * 3 + 2], R[2 * 3 + j]);
        {
            volatile float p0 = K[i * 3 + 0] * t[0], p1 = K[i * 3 + 1] * t[1], p2 = K[i * 3 + 2] * t[2];
            volatile float s = separate ? p0 + p1 : p1 + p2;
            kt[i] = separate ? s + p2 : s + p0;
        }
    }
}

static void sample_pos(int separate, const float *kr, const float *kt, float r0, float r1, float r2, float d,
                       float cx, float cy, int H, int W, float *ix, float *iy) {
    float a = dot3_blas(separate, kr[0], r0, kr[1], r1, kr[2], r2);
    float b = dot3_blas(separate, kr[3], r0, kr[4], r1, kr[5], r2);
    float c = dot3_blas(separate, kr[6], r0, kr[7], r1, kr[8], r2);
    volatile float ad = a * d, bd = b * d, cd = c * d; /* separate mul, then add (two ATen ops) */
    float px = kt[0] + ad, py = kt[1] + bd, pz = kt[2] + cd;
    float den = pz + 1e-10f;
    float u = px / den, v = py / den;
    float gx = (u - cx) / cx, gy = (v - cy) / cy;
    *ix = fmaf(gx + 1.0f, (float)W / 2.0f, -0.5f);
    *iy = fmaf(gy + 1.0f, (float)H / 2.0f, -0.5f);
}

/* One batch item.  ref [C,H,W], src [V,C,H,W], K [9], R [V,9], t [V,3], rays [3,HW], d [D].
 * Outputs (any may be NULL): cost [D,H,W], logp [D,H,W], depth [H,W]  -- all double. */
void pdo_sweep_dpv_f64(const float *ref, const float *src, const float *K, const float *R, const float *t,
                       const float *rays, float cx, float cy, const float *d_candi, int V, int C, int D,
                       int H, int W, double sigma, int metric, int blas_separate, double *cost, double *logp,
                       double *depth, double *scratch /* [D] per call, caller provided */) {
    const int HW = H * W;
    for (int p = 0; p < HW; ++p) {
        for (int k = 0; k < D; ++k) scratch[k] = 0.0;
        for (int v = 0; v < V; ++v) {
            float kr[9], kt[3];
            view_xform(blas_separate, K, R + v * 9, t + v * 3, kr, kt);
            const float *sv = src + (size_t)v * C * HW;
            for (int k = 0; k < D; ++k) {
                float ix, iy;
                sample_pos(blas_separate, kr, kt, rays[p], rays[HW + p], rays[2 * HW + p], d_candi[k], cx, cy, H, W, &ix, &iy);
                double acc = 0.0;
                if (ix == ix && iy == iy) {
                    float xf = floorf(ix), yf = floorf(iy);
                    float wx = ix - xf, ex = 1.0f - wx, ny = iy - yf, sy = 1.0f - ny; /* fp32 weights */
                    double nw = (double)sy * ex, ne = (double)sy * wx, sw = (double)ny * ex, se = (double)ny * wx;
                    double xc = xf < -2.f ? -2.f : (xf > W + 1 ? W + 1 : xf);
                    double yc = yf < -2.f ? -2.f : (yf > H + 1 ? H + 1 : yf);
                    int x0 = (int)xc, y0 = (int)yc;
                    int in00 = x0 >= 0 && x0 < W && y0 >= 0 && y0 < H;
                    int in01 = x0 + 1 >= 0 && x0 + 1 < W && y0 >= 0 && y0 < H;
                    int in10 = x0 >= 0 && x0 < W && y0 + 1 >= 0 && y0 + 1 < H;
                    int in11 = x0 + 1 >= 0 && x0 + 1 < W && y0 + 1 >= 0 && y0 + 1 < H;
                    for (int c = 0; c < C; ++c) {
                        const float *s = sv + (size_t)c * HW;
                        double val = 0.0;
                        if (in00) val += nw * s[y0 * W + x0];
                        if (in01) val += ne * s[y0 * W + x0 + 1];
                        if (in10) val += sw * s[(y0 + 1) * W + x0];
                        if (in11) val += se * s[(y0 + 1) * W + x0 + 1];
                        double diff = val - (double)ref[(size_t)c * HW + p];
                        acc += metric == 0 ? diff * diff : fabs(diff);
                    }
                } else {
                    acc = NAN;
                }
                scratch[k] += acc / sigma;
            }
        }
        if (cost)
            for (int k = 0; k < D; ++k) cost[(size_t)k * HW + p] = scratch[k];
        if (logp || depth) {
            double m = -INFINITY, s = 0.0, e = 0.0;
            for (int k = 0; k < D; ++k) m = scratch[k] > m ? scratch[k] : m;
            for (int k = 0; k < D; ++k) s += exp(scratch[k] - m);
            double ls = log(s);
            for (int k = 0; k < D; ++k) {
                double lp = scratch[k] - m - ls;
                if (logp) logp[(size_t)k * HW + p] = lp;
                e += (double)d_candi[k] * exp(lp);
            }
            if (depth) depth[p] = e;
        }
    }
}

/* logits [D,HW] (float) -> logp, depth in double. */
void pdo_dpv_reduce_f64(const float *logits, const float *d_candi, int D, int HW, double *logp, double *depth) {
    for (int p = 0; p < HW; ++p) {
        double m = -INFINITY, s = 0.0, e = 0.0;
        for (int k = 0; k < D; ++k) m = logits[(size_t)k * HW + p] > m ? logits[(size_t)k * HW + p] : m;
        for (int k = 0; k < D; ++k) s += exp((double)logits[(size_t)k * HW + p] - m);
        double ls = log(s);
        for (int k = 0; k < D; ++k) {
            double lp = (double)logits[(size_t)k * HW + p] - m - ls;
            if (logp) logp[(size_t)k * HW + p] = lp;
            e += (double)d_candi[k] * exp(lp);
        }
        if (depth) depth[p] = e;
    }
}
