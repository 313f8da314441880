// Build-time constants and diagnostics of sweep_corr.hip (one place for them: the kernel source itself carries no experiment
// switches).  A product build defines none of these names.
#pragma once

#ifndef CORR_OCC1   // waves per SIMD the register allocation must allow: D <= 64 / D <= 128
#define CORR_OCC1 4
#endif
#ifndef CORR_OCC2
#define CORR_OCC2 3
#endif
#ifndef CORR_STORE_AUX   // cache policy of the output stores: 2 = nt (written once, read by nobody in this launch: -2 % on the
#define CORR_STORE_AUX 2 // forward-motion pose, whose 8x2 pixel blocks store 32-byte runs)
#endif
#ifndef CORR_HALF_BANDS   // XCD q owns half-bands q and 8 + q of the image (balanced on a forward motion); 0: one band of rows per XCD
#define CORR_HALF_BANDS 1
#endif
#ifndef CORR_MS_ALWAYS    // fetch the <s', mu> records of a pass's slots even when none of its cells lies on the image border
#define CORR_MS_ALWAYS 0
#endif
#ifndef CORR_XPRIO        // wave priority while a wave is in the matrix phase (its loads go out first)
#define CORR_XPRIO 1
#endif
#ifndef CORR_ONE_EACH_X   // a workgroup per item, no queue, while the items are at most this many times the workgroups the chip holds at once
#define CORR_ONE_EACH_X 2
#endif
#ifndef CORR_SPI1_BELOW   // queue items are single pixel blocks (not 16x4 tiles of four) while there are fewer tiles than this many per workgroup
#define CORR_SPI1_BELOW 2
#endif
#ifndef CORR_PACK_AUX   // cache policy of the in-kernel pack's stores: 16 = sc1 (write-through to memory: the other XCDs read them)
#define CORR_PACK_AUX 16
#endif
#ifndef CORR_MAXB2   // ... at D > 64 (two groups of 64 planes per view, each a pass)
#define CORR_MAXB2 32
#endif
#ifndef CORR_MAXB1   // blocks of 16 texels a pass can take, D <= 64 (LDS: 4 workgroups per CU)
#define CORR_MAXB1 22
#endif

// Diagnostic build only (-DCORR_STAMPS, tools/dbg/corr_stamps.py): time per phase in 10 ns ticks, summed over waves, in the spare
// ints behind the queue counters.  No stamp exists in the product build.
#ifdef CORR_STAMPS
#define CSTAMP(idx) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); stamp_acc[idx] += now_ - stamp_t; stamp_t = now_; }
#else
#define CSTAMP(idx)
#endif

