// One texel of the packed source (sweep_pack.hip says what the layout is): shared by the pack kernel of the pre-pass and
// by the correlation-form sweep kernel, which packs batch item b + 1 while it sweeps item b (sweep_corr.hip).
// One thread per texel, channels in order (sequential fma: deterministic, the same bits on either path).
#pragma once
#include <hip/hip_runtime.h>

namespace pdepth {

// s = the texel's first channel (channel stride HW); hr / hd = the texel has a right / lower neighbour inside the image;
// mu = the constants subtracted per channel (LDS; CENTRE only, >= 4 * ceil(C / 4) entries, zeros beyond C);
// store(g, float4) writes plane g of the texel: g < ceil(C/4) the channel groups, then the two Gram planes.
template <bool CENTRE, typename Store>
__device__ __forceinline__ void pack_texel(const float* __restrict__ s, int C, int HW, int W, bool hr, bool hd, const float* mu, Store store) {
    const int ngrp = (C + 3) / 4;
    float n = 0.f, h = 0.f, vv = 0.f, d1 = 0.f, d2 = 0.f, mm = 0.f;
    // (branch-free: the neighbours beyond the image are loaded from the texel itself and then dropped)
    const int i01 = hr ? 1 : 0, i10 = hd ? W : 0;
    // a channel group = 16 loads (4 channels x the texel and its three neighbours); the next group's loads are issued before
    // the current one is used, so that 32 loads per thread are in flight (the pack is a stream: latency is all that it
    // can lose -- the compiler's own schedule waits for every channel's loads before it issues the next channel's)
    auto issue = [&](int g, float(&v)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = g * 4 + j;
            const float* sc = s + (size_t)min(c, C - 1) * HW;   // (channels beyond C: loaded from the last one, dropped below)
            v[4 * j + 0] = sc[0]; v[4 * j + 1] = sc[i01]; v[4 * j + 2] = sc[i10]; v[4 * j + 3] = sc[i01 + i10];
        }
    };
    auto finish = [&](int g, const float(&v)[16]) {
        float c4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = g * 4 + j;
            const float u = CENTRE ? mu[c] : 0.0f;   // (channels beyond C: 0, the statistics kernels pad with zeros)
            const bool in = c < C;   // uniform
            const float s00 = in ? v[4 * j + 0] - u : 0.f;
            const float s01 = in && hr ? v[4 * j + 1] - u : 0.f;
            const float s10 = in && hd ? v[4 * j + 2] - u : 0.f;
            const float s11 = in && hr && hd ? v[4 * j + 3] - u : 0.f;
            c4[j] = s00;
            n = __builtin_fmaf(s00, s00, n);
            h = __builtin_fmaf(s00, s01, h);
            vv = __builtin_fmaf(s00, s10, vv);
            d1 = __builtin_fmaf(s00, s11, d1);
            d2 = __builtin_fmaf(s01, s10, d2);
            mm = __builtin_fmaf(s00, u, mm);
        }
        store(g, make_float4(c4[0], c4[1], c4[2], c4[3]));
    };
    float va[16], vb[16];
    issue(0, va);
    for (int g = 0; g < ngrp; g += 2) {
        if (g + 1 < ngrp) issue(g + 1, vb);
        __builtin_amdgcn_sched_barrier(0);
        finish(g, va);
        if (g + 1 < ngrp) {
            if (g + 2 < ngrp) issue(g + 2, va);
            __builtin_amdgcn_sched_barrier(0);
            finish(g + 1, vb);
        }
    }
    store(ngrp, make_float4(n, h, vv, d1 + d2));
    store(ngrp + 1, make_float4(mm, 0.f, 0.f, 0.f));
}

}  // namespace pdepth
