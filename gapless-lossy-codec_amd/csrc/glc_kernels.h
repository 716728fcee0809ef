// glc_kernels.h — launch interface of the gfx950 kernels (implemented in glc_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace glc {

// Device-resident constant tables of one context.
struct DeviceTables {
  const float *cos_t;      // [2048][1024]  T transposed (row i): forward MDCT operand
  const float *cos;        // [1024][2048]  T (row k): inverse MDCT operand
  const float *window;     // [2048]
  const float *indiv;      // [1024] 1/weights[k].max(0.1)
  const float *band_pf;    // [n_bands]
  const float *band_len;   // [n_bands]
  const uint16_t *band_of; // [1024]
  const uint32_t *edges;   // [n_bands+1]
  uint32_t n_bands;
  float norm, cf, noise_floor;
};

// View of interleaved PCM on the device (whole stream or a shard with halo).
struct PcmView {
  const float *p;      // element (t0*ch) of the stream
  uint64_t t0;         // first per-channel sample index present
  uint64_t t_count;    // per-channel samples present
  uint64_t n_samples;  // interleaved length of the WHOLE stream
  uint32_t ch;
};

// K1: windowed forward MDCT of rows [0, M) (row = (frame - frame_begin)*ch + c) -> coef[M][1024].
hipError_t launch_mdct_forward(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin,
                               uint32_t M, float *coef, hipStream_t s, int variant = 0, bool beside = false);
// K2: scale, masking thresholds, quantiser -> record header {scale,nnz} + dense i16 row.  For 1 / 2 /
// 4 channels the kernel also takes the raw-vs-compressed decision and writes the raw plane of raw
// frames (*decided = true: do not launch K3); `pcm` / `frame_begin` are what that needs.
hipError_t launch_quantize(const DeviceTables &t, const float *coef, uint32_t M, uint32_t ch, const PcmView &pcm,
                           uint64_t frame_begin, uint8_t *records, hipStream_t s, bool *decided);
// K3: per-frame raw-vs-compressed decision and raw fallback plane (channel counts K2 does not decide).
hipError_t launch_decide_raw(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin,
                             uint32_t n_frames, uint8_t *records, hipStream_t s);

// P1-P3: compact blob (glc_common.h CompactLayout) of M = n_frames*ch rows of records, written to
// `blob` on the device: header, per-frame raw flags, per-row scale and pair count, the ascending
// (u16 idx | i16 q << 16) pairs of every compressed row back to back, then the 2048-sample planes of
// raw-frame rows.  loc[M], blk[ceil(M/1024)], blk_raw[same], totals[2] are scratch.  The alignment
// padding of the fixed sections is NOT written here (the caller zeroes [0, o_pairs) first).
hipError_t launch_compact(const uint8_t *records, uint32_t M, uint32_t ch, uint64_t n_frames, uint32_t *loc,
                          uint64_t *blk, uint64_t *blk_raw, uint64_t *totals, uint8_t *blob, uint64_t o_israw,
                          uint64_t o_scale, uint64_t o_cnt, uint64_t o_pairs, hipStream_t s);

// D1: sparse dequant + inverse MDCT + window -> blocks[row][2048].
//   pairs: packed (u16 idx | i16 q << 16), canonical (ascending, unique, idx < 1024)
//   row_begin[M] / row_cnt[M]: pair range of a row; row_scale[M]; row_raw[M]: -1 or offset (in
//   i16) of the frame's raw_pcm vec in raw_pool, row_raw_len[M] its length.
struct DecodeRows {
  const uint32_t *pairs;
  const uint64_t *row_begin;
  const uint32_t *row_cnt;
  const float *row_scale;
  const int64_t *row_raw;
  const uint64_t *row_raw_len;
  const int16_t *raw_pool;
  uint32_t any_raw;  // the stream has rows of raw frames (their blocks are written by a kernel of their own)
};
// variant (include/glc_debug.h): 0 = shipped (k_imdct_plan + k_imdct_apply, absent row pairs skipped
// by scalar branches); 1 = one row per workgroup (the cross-check kernel); 2 = plan + apply without
// the skip; 3 = without the issue-priority schedule; 4 = skipping in row pairs only.  All but 1 need a workspace `plan` of imdct_plan_bytes(plan_groups) bytes,
// plan_groups >= ch (launches with more (frame group, channel) units go through it in batches).
// reuse_plan: the workspace still holds the plan records of exactly this launch (same rows, same
// row_begin and M, one batch) - only the apply kernel runs.
uint64_t imdct_plan_bytes(uint32_t groups);
hipError_t launch_imdct_rows(const DeviceTables &t, const DecodeRows &rows, uint32_t row_begin,
                             uint32_t M, uint32_t ch, float *blocks, hipStream_t s, int variant = 0,
                             void *plan = nullptr, uint32_t plan_groups = 0, bool reuse_plan = false);
// D2: overlap-add + interleave of hops [hop_begin, hop_end) into out (hop h = second half of
// frame h-1 + first half of frame h; hop n_frames is the bare overlap tail).  `blocks` holds
// frames blk_frame0, blk_frame0+1, ... (blk_frame0 may be -1: a zero "frame before the first").
hipError_t launch_overlap_add(const float *blocks, int64_t blk_frame0, uint64_t n_frames,
                              uint32_t ch, uint64_t hop_begin, uint64_t hop_end, float *out,
                              hipStream_t s);

// Measurement only (include/glc_debug.h): one sleeping wave that reports {shader cycles, 100 MHz ticks}
// over a window of `ticks_100mhz` reference ticks, beside whatever runs on the device meanwhile.
hipError_t launch_clock_probe(uint64_t ticks_100mhz, uint64_t *out, hipStream_t s);

}  // namespace glc
