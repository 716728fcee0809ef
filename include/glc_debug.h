/*
 * glc_debug.h - cross-check hooks of libglc_hip.so.  NOT part of the drop-in boundary (that is
 * include/glc.h): these entry points exist so that soak tools and tests can run one stream through
 * two independent implementations of the same arithmetic and demand identical bits.
 */
#ifndef GLC_DEBUG_H
#define GLC_DEBUG_H

#include "glc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Which inverse-transform kernel the decode entry points of `ctx` launch (imdct_block,
 * src/codec.rs:377-390):
 *   0  shipped: k_imdct_plan + k_imdct_apply - 8 frames of one channel per unit over the union of
 *      their indices; union records in global memory, coefficients fed from SGPRs, absent rows
 *      skipped one by one by scalar branches
 *   1  k_imdct_rows: one row per workgroup, no grouping (the simplest restatement)
 *   2  plan + apply without the skip (every row takes every union entry)
 *   3  plan + apply without the issue-priority schedule (waves of a SIMD finish one after the other)
 *   4  plan + apply skipping absent rows only in pairs (both rows of a pair lack the entry)
 *   5  the shipped kernels, units dealt so that the four which share a CU are consecutive frame groups of
 *      one channel (L1 reuse of table rows) instead of being ranked by work   (placement: speed only)
 *   6  the shipped kernels in the natural unit order with the channel rotated per round (round 2's placement)
 * All of them produce the same bits; tools/soak_decode.py checks that on random streams. */
int glc_debug_set_imdct_variant(glc_ctx *ctx, int variant);

/* Which forward-transform kernel takes the launches of 4096 rows or more of `ctx` (mdct_block,
 * src/codec.rs:359-375; shorter launches are dispatched by row count alone and are not affected):
 *   0  shipped: k_mdct_fwd_st - table values from SGPRs, a wave's lanes hold 256 rows -, 16 waves per
 *      workgroup when the launch's last round of 32 row tiles is full or more than half full, else 8
 *   1  k_mdct_fwd_dma: round 3's kernel (128 x 128 tile, both operands from LDS)
 *   2  k_mdct_fwd_st, 8 waves per workgroup for every launch
 *   3  k_mdct_fwd_st, 16 waves per workgroup for every launch
 *   4  shipped, and launches of 1793..2048 rows take k_mdct_fwd_sched (64 x 128 tile) - the kernel glc_encode gives
 *      its opening rounds, which run beside each other; a launch that has the chip to itself takes the 2 x 4 kernel
 * All of them produce the same bits (tests/test_gpu_parity.py). */
int glc_debug_set_mdct_variant(glc_ctx *ctx, int variant);

/* The shader clock the device HOLDS under load (measurement only; bench.py's roofline.clock_ghz_held).
 * `begin` starts one sleeping wave on a stream of its own that runs for `window_us` microseconds beside
 * whatever the caller queues meanwhile and reads the shader-cycle counter against the constant 100 MHz
 * counter; `end` waits for it and returns cycles / ticks x 0.1 GHz.  The caller keeps the device busy
 * with the kernel of interest for at least the window. */
int glc_debug_clock_probe_begin(glc_ctx *ctx, uint32_t window_us);
int glc_debug_clock_probe_end(glc_ctx *ctx, float *ghz);

#ifdef __cplusplus
}
#endif
#endif /* GLC_DEBUG_H */
