// Numerics + layout probe for the distance-form sweep kernel (csrc/sweep_dist.hip):
//   Y[texel][pixel] = |s_texel - r_pixel|^2 over C = 67 channels, three ways:
//     (a) v_mfma_f32_16x16x32_f16 on fp16 hi/lo splits (7 instructions, specials folded: N_t, rr_n as fp16 pieces)
//     (b) v_mfma_f32_16x16x4_f32 chain for X, Y = N - 2X + rr in fp32 (what sweep_corr.hip's form costs in rounding)
//     (c) host fp64 on the fp32 inputs
// plus: A/B lane-map check with exact integer data, fp16 subnormal operands, and MFMA issue rate.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/mb_split16 tools/mb_split16.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int C = 67;

// one wave: lane (m = lane & 15, kq = lane >> 4).  A[m][k], B[k][n] given as [16][32] halfs row-major (A) and [16 n][32 k] (B: per pixel).
__global__ void k_layout(const _Float16* A, const _Float16* Bm, float* out) {
    const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[m * 32 + 8 * kq + j]; b[j] = Bm[m * 32 + 8 * kq + j]; }
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    // D[row = 4 kq + i][col = m]
    for (int i = 0; i < 4; ++i) out[(4 * kq + i) * 16 + m] = acc[i];
}

// split x into fp16 hi + lo
__device__ __host__ inline void split16(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}
__device__ __host__ inline void split3(float x, _Float16& a, _Float16& b, _Float16& c) {
    a = (_Float16)x; const float r1 = x - (float)a;
    b = (_Float16)r1; const float r2 = r1 - (float)b;
    c = (_Float16)r2;
}

// s [16 texels][C], r [16 pixels][C] fp32 (already centred + scaled).  Ya: split-fp16, Yb: fp32 MFMA chain.
__global__ void k_dist(const float* s, const float* r, float* Ya, float* Yb, int nrep) {
    const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
    for (int rep = 0; rep < nrep; ++rep) {
        const float* sr = s + (size_t)rep * 16 * C;
        const float* rr_ = r + (size_t)rep * 16 * C;
        // ---- (a) chunks: A0 = s_h[0..31], A1 = s_h[32..63], A2 = s_l[0..31], A3 = s_l[32..63], At = tail
        h8 A0, A1, A2, A3, At, Bh0, Bh1, Bl0, Bl1, Bt;
        float N = 0.f, RR = 0.f;
        for (int c = 0; c < C; ++c) { N = __builtin_fmaf(sr[m * C + c], sr[m * C + c], N); RR = __builtin_fmaf(rr_[m * C + c], rr_[m * C + c], RR); }
        for (int j = 0; j < 8; ++j) {
            _Float16 h, l;
            split16(sr[m * C + 8 * kq + j], h, l); A0[j] = h; A2[j] = l;
            split16(sr[m * C + 32 + 8 * kq + j], h, l); A1[j] = h; A3[j] = l;
            split16(rr_[m * C + 8 * kq + j], h, l); Bh0[j] = (_Float16)(-2.0f * (float)h); Bl0[j] = (_Float16)(-2.0f * (float)l);
            split16(rr_[m * C + 32 + 8 * kq + j], h, l); Bh1[j] = (_Float16)(-2.0f * (float)h); Bl1[j] = (_Float16)(-2.0f * (float)l);
            At[j] = (_Float16)0.f; Bt[j] = (_Float16)0.f;
        }
        // tail: slots 0..7 (kq 0): s_h[64+j] x r_h ; 8..15 (kq 1): s_l x r_h ; 16..23 (kq 2): s_h x r_l ; 24..31 (kq 3): specials
        for (int j = 0; j < 8; ++j) {
            const int c = 64 + j;
            _Float16 sh = (_Float16)0.f, sl = (_Float16)0.f, rh = (_Float16)0.f, rl = (_Float16)0.f;
            if (c < C) { split16(sr[m * C + c], sh, sl); split16(rr_[m * C + c], rh, rl); }
            if (kq == 0) { At[j] = sh; Bt[j] = (_Float16)(-2.0f * (float)rh); }
            if (kq == 1) { At[j] = sl; Bt[j] = (_Float16)(-2.0f * (float)rh); }
            if (kq == 2) { At[j] = sh; Bt[j] = (_Float16)(-2.0f * (float)rl); }
        }
        if (kq == 3) {
            _Float16 n1, n2, n3, q1, q2, q3;
            split3(N * (1.0f / 32.0f), n1, n2, n3);
            split3(RR * (1.0f / 32.0f), q1, q2, q3);
            At[0] = n1; At[1] = n2; At[2] = n3; Bt[0] = Bt[1] = Bt[2] = (_Float16)32.f;
            At[3] = At[4] = At[5] = (_Float16)32.f; Bt[3] = q1; Bt[4] = q2; Bt[5] = q3;
        }
        v4f acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, Bh0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, Bh1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, Bl0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, Bl1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2, Bh0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A3, Bh1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(At, Bt, acc, 0, 0, 0);
        for (int i = 0; i < 4; ++i) Ya[(size_t)rep * 256 + (4 * kq + i) * 16 + m] = acc[i];
        // ---- (b) fp32 chain: X[texel][pixel], 17 MFMAs of K = 4
        v4f x = {0, 0, 0, 0};
        for (int g = 0; g < 17; ++g) {
            const int c = 4 * g + kq;
            const float av = c < C ? sr[m * C + c] : 0.f, bv = c < C ? rr_[m * C + c] : 0.f;
            x = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, x, 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) {
            const int tex = 4 * kq + i;
            float Nt = 0.f;
            for (int c = 0; c < C; ++c) Nt = __builtin_fmaf(sr[tex * C + c], sr[tex * C + c], Nt);
            Yb[(size_t)rep * 256 + tex * 16 + m] = __builtin_fmaf(-2.0f, x[i], Nt) + RR;
        }
    }
}

__global__ void k_subnormal(float* out) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    // A[m][k=0] = 2^-20 (fp16 subnormal), B[0][n] = 1024  -> D = 2^-10 if subnormals are honoured, 0 if flushed
    if (lane < 16) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1024.f; }
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

template <int NACC>
__global__ void k_rate(float* out, int iters) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f - threadIdx.x * 0.002f); }
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double frand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double gauss() { return sqrt(-2.0 * log(frand())) * cos(6.283185307179586 * frand()); }

int main() {
    // ---- layout
    {
        std::vector<_Float16> A(16 * 32), B(16 * 32);
        for (int m = 0; m < 16; ++m) for (int k = 0; k < 32; ++k) { A[m * 32 + k] = (_Float16)(float)((m * 7 + k * 3) % 11 - 5); B[m * 32 + k] = (_Float16)(float)((m * 5 + k * k) % 13 - 6); }
        _Float16 *dA, *dB; float* dO;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dO, 256 * 4);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dO);
        std::vector<float> O(256); hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int row = 0; row < 16; ++row) for (int col = 0; col < 16; ++col) {
            double e = 0; for (int k = 0; k < 32; ++k) e += (double)(float)A[row * 32 + k] * (double)(float)B[col * 32 + k];
            if (O[row * 16 + col] != (float)e) ++bad;
        }
        printf("layout: D[row=texel m][col=pixel n] with A[m][8kq+j], B[n][8kq+j]: %d mismatches of 256\n", bad);
    }
    // ---- subnormals
    {
        float* dO; hipMalloc(&dO, 4); hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, dO);
        float o; hipMemcpy(&o, dO, 4, hipMemcpyDeviceToHost);
        printf("subnormal fp16 operand: 2^-20 * 1024 = %g (expected %g; 0 = flushed)\n", o, 9.765625e-4);
    }
    // ---- numerics: nrep blocks of 16 texels x 16 pixels; features N(0,1) * scale + common offset `off` * direction
    const int nrep = 64;
    for (int cfg = 0; cfg < 6; ++cfg) {
        const double scale = 32.0;                       // sigma -> 32 (the pack kernel's power-of-two scaling)
        const double off = cfg == 0 ? 0.0 : cfg == 1 ? 1.0 : cfg == 2 ? 2.0 : cfg == 3 ? 4.0 : cfg == 4 ? 8.0 : 0.0;   // common component, in sigma per channel
        const bool correlated = cfg == 5;                // s = r + 0.3 noise: small distances (peaked DPV)
        std::vector<float> s((size_t)nrep * 16 * C), r((size_t)nrep * 16 * C);
        srand(1234 + cfg);
        for (int rep = 0; rep < nrep; ++rep) {
            std::vector<double> common(C);
            for (int c = 0; c < C; ++c) common[c] = off * (frand() < 0.5 ? -1 : 1);
            for (int i = 0; i < 16; ++i) for (int c = 0; c < C; ++c) {
                const double rv = gauss() + common[c];
                r[((size_t)rep * 16 + i) * C + c] = (float)(scale * rv);
                s[((size_t)rep * 16 + i) * C + c] = (float)(scale * (correlated ? rv + 0.3 * gauss() : gauss() + common[c]));
            }
        }
        float *ds, *dr, *dYa, *dYb;
        hipMalloc(&ds, s.size() * 4); hipMalloc(&dr, r.size() * 4); hipMalloc(&dYa, nrep * 256 * 4); hipMalloc(&dYb, nrep * 256 * 4);
        hipMemcpy(ds, s.data(), s.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, r.data(), r.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_dist, dim3(1), dim3(64), 0, 0, ds, dr, dYa, dYb, nrep);
        std::vector<float> Ya(nrep * 256), Yb(nrep * 256);
        hipMemcpy(Ya.data(), dYa, Ya.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(Yb.data(), dYb, Yb.size() * 4, hipMemcpyDeviceToHost);
        double ea = 0, eb = 0, ed = 0, ymean = 0, ra = 0, rb = 0, rd = 0;
        for (int rep = 0; rep < nrep; ++rep) for (int t = 0; t < 16; ++t) for (int n = 0; n < 16; ++n) {
            double y = 0; float yd = 0.f;
            for (int c = 0; c < C; ++c) {
                const double d = (double)s[((size_t)rep * 16 + t) * C + c] - (double)r[((size_t)rep * 16 + n) * C + c];
                y += d * d;
                const float df = s[((size_t)rep * 16 + t) * C + c] - r[((size_t)rep * 16 + n) * C + c];   // the reference's own form in fp32
                yd = fmaf(df, df, yd);
            }
            const double da = fabs(Ya[rep * 256 + t * 16 + n] - y), db = fabs(Yb[rep * 256 + t * 16 + n] - y), dd = fabs((double)yd - y);
            ea = fmax(ea, da); eb = fmax(eb, db); ed = fmax(ed, dd); ra += da * da; rb += db * db; rd += dd * dd; ymean += y;
        }
        const double cnt = nrep * 256.0, u = scale * scale * 10.0;   // cost units: / scale^2 / sigma(10)
        printf("cfg %d (offset %.0f sigma%s): mean cost %.3f | cost error max / rms:  split-fp16 %.2e / %.2e   fp32-mfma N-2X+rr %.2e / %.2e   fp32 direct %.2e / %.2e\n",
               cfg, off, correlated ? ", correlated" : "", ymean / cnt / u, ea / u, sqrt(ra / cnt) / u, eb / u, sqrt(rb / cnt) / u, ed / u, sqrt(rd / cnt) / u);
        hipFree(ds); hipFree(dr); hipFree(dYa); hipFree(dYb);
    }
    // ---- issue rate
    {
        float* dO; hipMalloc(&dO, 1024 * 256 * 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const int iters = 20000;
        auto run = [&](auto kern, int nacc, const char* nm) {
            hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, dO, 10); hipDeviceSynchronize();
            hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, dO, iters); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double n_mfma = 1024.0 * 4 * iters * nacc;   // per wave
            printf("%s: %.3f ms, %.1f cycles per MFMA per SIMD at 2.4 GHz (1024 SIMDs)\n", nm, ms, ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
        };
        run(k_rate<1>, 1, "16x16x32 f16, 1 accumulator chain");
        run(k_rate<2>, 2, "16x16x32 f16, 2 accumulator chains");
    }
    return 0;
}
