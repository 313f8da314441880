// Build-time knobs of sweep_dist.hip (tools/variants_dist.sh sets them with -D for A/B builds; the defaults are the product).
#pragma once

#ifndef DIST_MAXB1
#define DIST_MAXB1 34      // blocks of 16 texels a pass can take, D <= 64 (22 until the end of round 5: the headline is indifferent, wide baselines evaluate fewer passes directly; three workgroups per CU either way)
#endif
#ifndef DIST_MAXB2
#define DIST_MAXB2 34      // ... D > 64 (the most that leaves three workgroups per CU: 35 -> two; config 5: 32 -> 34 = 6 809 -> 1 482 direct passes, 5.58 -> 5.03 ms per call)
#endif
#ifndef DIST_MAXB_NP2
#define DIST_MAXB_NP2 26   // ... of a pass over two pixel blocks (D <= 64): 75 KB of LDS per workgroup, two per CU
#endif
#ifndef DIST_NP2
#define DIST_NP2 0         // 1: whole tiles at D <= 64 run two pixel blocks per pass (built, parity-green, measured 25 % SLOWER: it
#endif                     //    saves 4 % of the vector instructions and leaves two workgroups per CU instead of three)
#ifndef DIST_OCC1
#define DIST_OCC1 3        // minimum waves per SIMD asked of the compiler at D <= 64: 167 registers with two texel operand sets, nothing spilled,
#endif                     //    three workgroups per CU (2: the allocator takes 178 -- two workgroups per CU)
#ifndef DIST_OCC2
#define DIST_OCC2 3        // ... at D > 64: without the bound the allocator takes 170 registers (two workgroups per CU: config 5 20 % slower),
#endif                     //    with it 165, nothing spilled
#ifndef DIST_SETS1
#define DIST_SETS1 2       // texel operand register sets of a wave (= its blocks in flight) at D <= 64 ...
#endif
#ifndef DIST_SETS2
#define DIST_SETS2 1       // ... and at D > 64 (2: 168 registers under the launch bound, 8 of them spilled)
#endif
#ifndef DIST_XPRIO
#define DIST_XPRIO 1       // wave priority in the matrix phase
#endif
#ifndef DIST_STORE_AUX
#define DIST_STORE_AUX 2   // nt: the outputs are written once and not read by this kernel (0, 1, 3: no difference measured)
#endif
#ifndef DIST_BANDS
#define DIST_BANDS 16      // bands of tile rows per image in the XCD partition: 16 (XCD q: half-bands q and 8 + q) | 8 (band q: 7 % fewer L2 misses, 5 % slower -- the XCDs' loads differ; profiles/r05_ab/xcd_bands_8_vs_16.txt)
#endif
#ifndef DIST_COL_ALT
#define DIST_COL_ALT 0     // columns of a band from both image borders inwards (1) | left to right (0: 1.2 % faster on the forward motion, 5 % fewer L2 misses)
#endif
// 1: a pass whose texel blocks do not fit LDS is retried as two passes of 32 planes before the direct evaluation.  Built,
// parity-green, off: config 5 (6 809 of 524 288 passes direct) 5.23 -> 5.13 ms per call with 10 registers spilled, 5.92 ms
// without spills (193 registers: two workgroups per CU); the headline 2-3 % slower (profiles/r05_ab/split_planes.txt)
#ifndef DIST_QSTRIDE
#define DIST_QSTRIDE 64    // ints between the queue counters of two XCDs
#endif
#ifndef DIST_ILV
#define DIST_ILV 1         // whole items are pixel block s of four vertically adjacent tiles (sweep_dist.hip: decode)
#endif
#ifndef DIST_ABL
#define DIST_ABL 0         // timing only (wrong results), bits: 1 = every sample position computed twice, 2 = no texel loads
#endif
#ifndef DIST_ABL_NOB3
#define DIST_ABL_NOB3 0    // timing only (wrong results): no barrier in front of the merge of the waves' softmax parts
#endif
// channel groups of the direct evaluation whose taps are in flight together (3 / 9: 168 registers under a launch bound of 3
// waves per SIMD, no scratch -- and config 5, 1.3 % of whose passes go this way, 8 % slower)
#ifndef DIST_DIRECT_UNROLL
#define DIST_DIRECT_UNROLL 1
#endif
#ifndef DIST_SPLIT_PLANES
#define DIST_SPLIT_PLANES 0
#endif
// tiles at the end of every XCD queue that are handed out as four single pixel blocks, in percent of the workgroups per XCD:
// the end of a launch is then a pass long instead of a tile long.  0: 0.408 / 0.377 ms, 100: 0.402 / 0.363, 200: 0.404 / 0.369,
// 400: 0.409 / 0.379 (profiles/r05_ab/queue_tail_as_single_blocks.txt)
#ifndef DIST_TAIL_PCT
#define DIST_TAIL_PCT 100
#endif
#ifndef DIST_STAGGER
#define DIST_STAGGER 2       // start-up stagger of the persistent workgroups, in sleeps of 6 400 cycles per CU slot (0: off; 1 / 3 / 4: less)
#endif
#ifndef DIST_SPI1_BELOW
#define DIST_SPI1_BELOW 2  // single pixel blocks as queue items below this many tiles per workgroup
#endif
#ifndef DIST_ONE_EACH_X
// no queue -- a workgroup per item, the hardware's dispatcher instead of the per-XCD counters -- up to this many items per
// resident workgroup.  Packed entry, us per call at 2 / 6 (profiles/r05_ab/workgroup_per_item_range.txt): B=4 64x128 60.7 / 45.5,
// B=8 64x128 84.6 / 78.5, B=1 128x256 58.9 / 45.9, B=4 128x256 119 / 102, B=1 256x512 185 / 159, B=2 256x512 229 / 214;
// beyond: B=3 256x512 (8 x) 294 persistent / 307, B=4 256x512 (10.7 x) 401 / 435
#define DIST_ONE_EACH_X 6
#endif
#ifndef DIST_GUARD_RATIO
#define DIST_GUARD_RATIO 1.7f   // the guard: energy of the centred features / their spread at a lag of 16 texels, and ...
#endif
#ifndef DIST_GUARD_ENERGY
#define DIST_GUARD_ENERGY 110.0f   // ... energy x 10 / sigma beyond which an item is evaluated directly (sweep_dist.hip)
#endif
#ifndef DIST_EXACT_EXP
#define DIST_EXACT_EXP 0   // 1: geometry.hpp's exp_nonpos (1.5 ulp, 12 instructions) instead of the hardware 2^x on the rounded product
#endif
#ifndef DIST_FORCE_DIRECT
#define DIST_FORCE_DIRECT -2   // test builds: -1 = every pass takes the direct evaluation, v >= 0 = the passes of view v
#endif

// The matrix instruction.  DIST_MFMA16: two K = 16 instructions (v_mfma_f32_16x16x16_f16) per K = 32 operand pair -- the probe
// that showed the packed-fp32 erratum (wave_util.hpp) to need v_mfma_f32_16x16x32_f16 in the other waves.
#ifdef DIST_MFMA16
#define DIST_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(a, a, 4, 5, 6, 7), __builtin_shufflevector(b, b, 4, 5, 6, 7), \
                               __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(a, a, 0, 1, 2, 3), __builtin_shufflevector(b, b, 0, 1, 2, 3), c, 0, 0, 0), 0, 0, 0)
#else
#define DIST_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif

// phase stamps (-DDIST_STAMPS): shader-clock cycles per phase, summed per wave, added into the queue ints 8..31 on the way out
// (every stamp is a scalar memory read and a wait for it: the phases stretch, their proportions are indicative only)
#ifdef DIST_STAMPS
#define DSTAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_t; stamp_t = t_; }
#elif defined(DIST_MARKS)   // static instruction census per phase (tools/dbg/isa_census.py)
#define DSTAMP(i) asm volatile("; MARK " #i ::: "memory");
#else
#define DSTAMP(i)
#endif
