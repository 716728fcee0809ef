// glc_kernels.hip — hand-written gfx950 (CDNA4) kernels of the codec hot path.
//
// Numerics contract: the reference transform is a dense 1024x2048 table
// contraction accumulated strictly in ascending index order with a separately rounded f32
// multiply and f32 add per term (rustc never fuses).  Every kernel below keeps that order and
// never lets the compiler contract a*b+c: the file is compiled with -ffp-contract=off and the
// accumulation is additionally written with __fmul_rn/__fadd_rn.  No v_fma/v_fmac/v_pk_fma may
// appear in the MDCT kernels (checked at build time by tools/check_isa.py).
//
// Reference loops replaced (file:line into /root/reference):
//   K1 k_mdct_fwd      src/codec.rs:476-481 (window) + :359-374 (mdct_block)
//   K2 k_quantize      :488 (scale) + :188-240 (thresholds) + :270-311 (quantiser)
//   K3 k_decide_raw    :496-502 (raw plane) + :505-540 (size estimate, decision)
//   D1 k_imdct_rows    :626-644 (raw frames), :651-665 (dequant), :377-390 (imdct), :672-675
//   D2 k_overlap_add   :688-705 (overlap-add + interleave), :722-729 (tail)
#include "glc_kernels.h"
#include "glc_mdct_fwd.hpp"

#pragma clang fp contract(off)

namespace glc {

namespace {

constexpr int kHopI = 1024;
constexpr int kFrameI = 2048;

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }

// One zero-padded, de-interleaved PCM sample: padded[c][frame*1024 + i] of src/codec.rs:433-447.
__device__ __forceinline__ float pcm_at(const PcmView &v, int64_t frame, uint32_t c, int i) {
  const int64_t t = frame * kHopI + i - kHopI / 2;  // 512 leading zeros
  if (t < 0) return 0.0f;
  const uint64_t idx = static_cast<uint64_t>(t) * v.ch + c;
  if (idx >= v.n_samples) return 0.0f;  // trailing padding
  const uint64_t rel = static_cast<uint64_t>(t) - v.t0;
  if (static_cast<uint64_t>(t) < v.t0 || rel >= v.t_count) return 0.0f;  // outside the shard
  return v.p[rel * v.ch + c];
}

__device__ __forceinline__ short sat_i16(float x) {
  // f32::clamp(-32768, 32767) then `as i16`: truncation, NaN -> 0 (src/codec.rs:301,501)
  if (x != x) return 0;
  x = fminf(fmaxf(x, -32768.0f), 32767.0f);
  return static_cast<short>(static_cast<int>(x));
}

// K1 (forward MDCT) lives in glc_mdct_fwd.hpp.

// ------------------------------------------------------------------------------------------
// K2: scale, masking thresholds, quantiser (+ the raw-vs-compressed decision when a frame's
// channels all sit in one wave).  A wavefront owns 4 consecutive frame-channel rows.
//   phase 1  16 coefficients per lane and row: max|c| by shuffle (order-free), squares to LDS
//   phase 2  band sums in the reference's ascending order (src/codec.rs:212-214): lane l sums
//            bands (l & 15), +16, +32, +48 of row l >> 4, so the long last band (683 bins at
//            48 kHz) of the 4 rows runs in 4 lanes side by side instead of one lane per wave
//   phase 3  thresholds, noise floor, quantiser; {scale, nnz} into the record header
//   phase 4  FUSED (1 / 2 / 4 channels: a frame's rows are 1 / 2 / 4 consecutive rows of this
//            wave): size estimate and decision of src/codec.rs:505-521; a compressed frame gets
//            its dense i16 rows, a raw frame the channel-planar windowed i16 plane (:496-502, Q1)
//            - K3 is not launched at all.  Other channel counts: dense rows here, decision in K3.
// ------------------------------------------------------------------------------------------
constexpr int kQRows = 4;  // rows per wave

template <bool FUSED>
__global__ __launch_bounds__(256) void k_quantize(DeviceTables tb, const float *__restrict__ coef,
                                                   unsigned M, unsigned ch, unsigned long long rec_bytes,
                                                   unsigned long long hdr_bytes, PcmView pcm, long long frame_begin,
                                                   unsigned char *__restrict__ records) {
  __shared__ __attribute__((aligned(16))) float ssq[4][kQRows][kHopI];  // 64 KiB
  __shared__ float sbase[4][kQRows][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned m0 = (blockIdx.x * 4 + w) * kQRows;

  // per-lane constants of the 16 bins this lane owns (the same bins in every row)
  float4 indiv4[4];
  unsigned short bo[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k0 = (lane + 64 * j) * 4;
    indiv4[j] = *reinterpret_cast<const float4 *>(tb.indiv + k0);
    const ushort4 b4 = *reinterpret_cast<const ushort4 *>(tb.band_of + k0);
    bo[j][0] = b4.x; bo[j][1] = b4.y; bo[j][2] = b4.z; bo[j][3] = b4.w;
  }

  float4 c4[kQRows][4];
  float scale[kQRows];
#pragma unroll
  for (int r = 0; r < kQRows; ++r) {
    const unsigned m = m0 + r;
    float amax = 0.0f;
    if (m < M) {
      const float4 *src = reinterpret_cast<const float4 *>(coef + static_cast<size_t>(m) * kHopI);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = src[lane + 64 * j];
        c4[r][j] = v;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        float4 sq;  // each square is one rounding, order-free; only the SUM is ordered
        sq.x = mul_rn(v.x, v.x); sq.y = mul_rn(v.y, v.y); sq.z = mul_rn(v.z, v.z); sq.w = mul_rn(v.w, v.w);
        *reinterpret_cast<float4 *>(&ssq[w][r][(lane + 64 * j) * 4]) = sq;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) c4[r][j] = float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    scale[r] = fmaxf(amax, 1e-10f);  // :488 (and global_max at :198, :278)
  }
  __syncthreads();

  {
    const int r = lane >> 4;
    if (m0 + r < M) {
      const float *sq = ssq[w][r];
      for (unsigned b = lane & 15; b < tb.n_bands; b += 16) {
        const unsigned lo = tb.edges[b], hi = tb.edges[b + 1];
        float ss = 0.0f;
        unsigned i = lo;
        // head: up to 3 bins until the index is 16-byte aligned
        for (; i < hi && (i & 3u); ++i) ss = add_rn(ss, sq[i]);
        // body: 16 bins per step as four ds_read_b128, the next step's reads in flight while the
        // current 16 adds (a dependent chain, the reference's order) execute; ping-pong registers
#define GLC_ADD4(V) ss = add_rn(ss, V.x); ss = add_rn(ss, V.y); ss = add_rn(ss, V.z); ss = add_rn(ss, V.w)
        if (i + 16 <= hi) {
          const float4 *q4 = reinterpret_cast<const float4 *>(sq);
          float4 a0 = q4[i >> 2], a1 = q4[(i >> 2) + 1], a2 = q4[(i >> 2) + 2], a3 = q4[(i >> 2) + 3];
          while (i + 32 <= hi) {
            const float4 b0 = q4[(i >> 2) + 4], b1 = q4[(i >> 2) + 5], b2 = q4[(i >> 2) + 6], b3 = q4[(i >> 2) + 7];
            GLC_ADD4(a0); GLC_ADD4(a1); GLC_ADD4(a2); GLC_ADD4(a3);
            i += 16;
            if (i + 32 <= hi) {
              a0 = q4[(i >> 2) + 4]; a1 = q4[(i >> 2) + 5]; a2 = q4[(i >> 2) + 6]; a3 = q4[(i >> 2) + 7];
              GLC_ADD4(b0); GLC_ADD4(b1); GLC_ADD4(b2); GLC_ADD4(b3);
              i += 16;
            } else {
              a0 = b0; a1 = b1; a2 = b2; a3 = b3;
            }
          }
          GLC_ADD4(a0); GLC_ADD4(a1); GLC_ADD4(a2); GLC_ADD4(a3);
          i += 16;
        }
#undef GLC_ADD4
        for (; i < hi; ++i) ss = add_rn(ss, sq[i]);  // tail: fewer than 16 bins
        const float energy = sqrtf(ss / tb.band_len[b]);                                  // :214-215
        sbase[w][r][b] = mul_rn(mul_rn(mul_rn(energy, 0.01f), tb.cf), tb.band_pf[b]);      // :223
      }
    }
  }
  __syncthreads();

  short4 pk[kQRows][4];
  unsigned nnz[kQRows];
#pragma unroll
  for (int r = 0; r < kQRows; ++r) {
    const unsigned m = m0 + r;
    nnz[r] = 0;
    if (m >= M) break;
    const float sc = scale[r];
    const float nfl = mul_rn(tb.noise_floor, sc);  // :277
    const float peak_gate = mul_rn(sc, 0.3f);      // global_max * 0.3, :232
    const float peak_cap = mul_rn(sc, 0.05f);      // global_max * 0.05, :234
    unsigned cnt = 0;
    // the 16 band bases of this lane's bins, fetched together: left to the compiler, each of the 16 LDS reads
    // sits directly in front of its use and is waited for there (64 exposed LDS round trips per wave)
    float sb[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) sb[j][e] = sbase[w][r][bo[j][e]];
    asm volatile("" : "+v"(sb[0][0]), "+v"(sb[0][1]), "+v"(sb[0][2]), "+v"(sb[0][3]), "+v"(sb[1][0]), "+v"(sb[1][1]),
                      "+v"(sb[1][2]), "+v"(sb[1][3]), "+v"(sb[2][0]), "+v"(sb[2][1]), "+v"(sb[2][2]), "+v"(sb[2][3]),
                      "+v"(sb[3][0]), "+v"(sb[3][1]), "+v"(sb[3][2]), "+v"(sb[3][3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float cv[4] = {c4[r][j].x, c4[r][j].y, c4[r][j].z, c4[r][j].w};
      const float iv[4] = {indiv4[j].x, indiv4[j].y, indiv4[j].z, indiv4[j].w};
      short qv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = fabsf(cv[e]);
        float t = mul_rn(sb[j][e], iv[e]);                           // :228-229
        if (a > peak_gate) t = fminf(t, peak_cap);                  // :232-235
        const float thr = mul_rn(t, sc);                            // :288
        short q = 0;
        if (a > nfl && a > thr) {                                   // :291
          const float normalized = cv[e] / sc;                      // :299 (IEEE divide)
          q = sat_i16(roundf(mul_rn(normalized, 32768.0f)));        // :300-301
        }
        qv[e] = q;
        cnt += (q != 0);
      }
      pk[r][j].x = qv[0]; pk[r][j].y = qv[1]; pk[r][j].z = qv[2]; pk[r][j].w = qv[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    nnz[r] = cnt;
    if (lane == 0) {
      const unsigned frame = m / ch, c = m % ch;
      unsigned char *rec = records + static_cast<size_t>(frame) * rec_bytes;
      *reinterpret_cast<float *>(rec + 8 + 8 * c) = sc;
      *reinterpret_cast<unsigned *>(rec + 8 + 8 * c + 4) = cnt;
    }
  }

  // phase 4: payload.  FUSED: the frame of row r spans rows r - r % ch .. + ch - 1 of this wave.
#pragma unroll
  for (int r = 0; r < kQRows; ++r) {
    const unsigned m = m0 + r;
    if (m >= M) break;
    const unsigned frame = m / ch, c = m % ch;
    unsigned char *rec = records + static_cast<size_t>(frame) * rec_bytes;
    bool use_raw = false;
    if (FUSED) {
      unsigned long long compressed = 8ull + 4ull * ch + 64ull;   // :513, :515
#pragma unroll
      for (int q = 0; q < kQRows; ++q)
        if (static_cast<unsigned>(q) / ch == static_cast<unsigned>(r) / ch) compressed += 8ull + 4ull * nnz[q];  // :507-511
      const unsigned long long raw_size = 2ull * kFrameI * ch;    // :518
      use_raw = static_cast<float>(compressed) >= mul_rn(static_cast<float>(raw_size), 0.85f);  // :521
      if (c == 0 && lane == 0) {
        *reinterpret_cast<unsigned *>(rec) = use_raw ? 1u : 0u;
        *reinterpret_cast<unsigned *>(rec + 4) = 0u;
      }
    }
    short *qrow = reinterpret_cast<short *>(rec + hdr_bytes) + static_cast<size_t>(c) * kFrameI;
    if (!use_raw) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<short4 *>(qrow + (lane + 64 * j) * 4) = pk[r][j];
    } else {
      // raw fallback plane of this row's channel, windowed once (:498-502)
      const long long fabs_ = frame_begin + frame;
      for (int i = lane; i < kFrameI; i += 64) {
        const float sw = mul_rn(pcm_at(pcm, fabs_, c, i), tb.window[i]);  // :500
        qrow[i] = sat_i16(mul_rn(sw, 32767.0f));                          // :501
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K3: one workgroup per frame: size estimate and raw-vs-compressed decision; raw frames get
// the channel-planar windowed i16 plane (quirk Q1) written over their payload.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decide_raw(DeviceTables tb, PcmView pcm, long long frame_begin,
                                                     unsigned n_frames, unsigned long long rec_bytes,
                                                     unsigned long long hdr_bytes,
                                                     unsigned char *__restrict__ records) {
  const unsigned fr = blockIdx.x;
  if (fr >= n_frames) return;
  unsigned char *rec = records + static_cast<size_t>(fr) * rec_bytes;
  const unsigned ch = pcm.ch;
  unsigned long long compressed = 0;
  for (unsigned c = 0; c < ch; ++c)
    compressed += 8ull + 4ull * *reinterpret_cast<const unsigned *>(rec + 8 + 8 * c + 4);  // :507-511
  compressed += 8ull + 4ull * ch;  // :513
  compressed += 64ull;             // :515
  const unsigned long long raw_size = 2ull * kFrameI * ch;  // :518
  const bool use_raw =
      static_cast<float>(compressed) >= mul_rn(static_cast<float>(raw_size), 0.85f);  // :521
  if (threadIdx.x == 0) {
    *reinterpret_cast<unsigned *>(rec) = use_raw ? 1u : 0u;
    *reinterpret_cast<unsigned *>(rec + 4) = 0u;
  }
  if (!use_raw) return;
  short *plane = reinterpret_cast<short *>(rec + hdr_bytes);
  const long long frame = frame_begin + fr;
  for (unsigned idx = threadIdx.x; idx < ch * kFrameI; idx += 256) {
    const unsigned c = idx / kFrameI, i = idx % kFrameI;
    const float s = mul_rn(pcm_at(pcm, frame, c, static_cast<int>(i)), tb.window[i]);  // :500
    plane[idx] = sat_i16(mul_rn(s, 32767.0f));                                         // :501
  }
}

// ------------------------------------------------------------------------------------------
// D1: one workgroup per frame-channel row, 8 outputs per lane.  out[i] = sum over k ascending
// of c[k]*T[k][i]; adding the +0.0 products of zero coefficients is the identity on the running
// sum (which is never -0.0), so iterating only the stored non-zeros in ascending k is
// bit-identical to the reference's dense loop and does nnz/1024 of the work.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_imdct_rows(DeviceTables tb, DecodeRows rows, unsigned row_begin,
                                                     unsigned M, unsigned ch, float *__restrict__ blocks) {
  __shared__ float s_val[kHopI];
  __shared__ unsigned short s_idx[kHopI];
  const unsigned r = blockIdx.x;
  if (r >= M) return;
  const unsigned m = row_begin + r;
  float *out = blocks + static_cast<size_t>(r) * kFrameI;
  const int tid = threadIdx.x;

  const long long raw_off = rows.row_raw[m];
  if (raw_off >= 0) {
    // raw frame: read as if interleaved (Q1), /32767, no window (Q2) — src/codec.rs:629-640
    const unsigned c = m % ch;
    const unsigned long long raw_len = rows.row_raw_len[m];
    const short *raw = rows.raw_pool + raw_off;
    for (int i = tid; i < kFrameI; i += 256) {
      const unsigned long long si = static_cast<unsigned long long>(i) * ch + c;
      float v = 0.0f;
      if (si < raw_len) v = static_cast<float>(raw[si]) / 32767.0f;
      out[i] = v;
    }
    return;
  }

  const unsigned long long p0 = rows.row_begin[m];
  const unsigned n = min(rows.row_cnt[m], static_cast<unsigned>(kHopI));  // canonical lists hold <= 1024
  const float scale = fmaxf(rows.row_scale[m], 1e-12f);  // :653
  for (unsigned j = tid; j < n; j += 256) {
    const unsigned pr = rows.pairs[p0 + j];
    const short q = static_cast<short>(pr >> 16);
    s_idx[j] = static_cast<unsigned short>(pr & 0xFFFFu);
    s_val[j] = mul_rn(static_cast<float>(q) / 32768.0f, scale);  // :663
  }
  __syncthreads();

  float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  const float *T = tb.cos + tid * 4;
  for (unsigned j = 0; j < n; ++j) {
    const float cv = s_val[j];
    const float *trow = T + static_cast<size_t>(s_idx[j]) * kFrameI;
    const float4 t0 = *reinterpret_cast<const float4 *>(trow);
    const float4 t1 = *reinterpret_cast<const float4 *>(trow + 1024);
    a0.x = add_rn(a0.x, mul_rn(cv, t0.x)); a0.y = add_rn(a0.y, mul_rn(cv, t0.y));
    a0.z = add_rn(a0.z, mul_rn(cv, t0.z)); a0.w = add_rn(a0.w, mul_rn(cv, t0.w));
    a1.x = add_rn(a1.x, mul_rn(cv, t1.x)); a1.y = add_rn(a1.y, mul_rn(cv, t1.y));
    a1.z = add_rn(a1.z, mul_rn(cv, t1.z)); a1.w = add_rn(a1.w, mul_rn(cv, t1.w));
  }
  const float4 w0 = *reinterpret_cast<const float4 *>(tb.window + tid * 4);
  const float4 w1 = *reinterpret_cast<const float4 *>(tb.window + 1024 + tid * 4);
  float4 o0, o1;  // out[i] = s*norm (:388) then *= window[i] (:674)
  o0.x = mul_rn(mul_rn(a0.x, tb.norm), w0.x); o0.y = mul_rn(mul_rn(a0.y, tb.norm), w0.y);
  o0.z = mul_rn(mul_rn(a0.z, tb.norm), w0.z); o0.w = mul_rn(mul_rn(a0.w, tb.norm), w0.w);
  o1.x = mul_rn(mul_rn(a1.x, tb.norm), w1.x); o1.y = mul_rn(mul_rn(a1.y, tb.norm), w1.y);
  o1.z = mul_rn(mul_rn(a1.z, tb.norm), w1.z); o1.w = mul_rn(mul_rn(a1.w, tb.norm), w1.w);
  *reinterpret_cast<float4 *>(out + tid * 4) = o0;
  *reinterpret_cast<float4 *>(out + 1024 + tid * 4) = o1;
}

// ------------------------------------------------------------------------------------------
// D1, SHIPPED: plan + apply.  A unit of work is 8 consecutive frames of ONE channel (rows f*ch + c,
// f = f0 .. f0+7) decoded over the UNION of their coefficient indices, so that a table row is read
// from L2 once for the group.  Consecutive frames of one channel are the rows that share indices:
// tonal material keeps its partials from frame to frame, while two channels may carry different
// instruments.  Per row the arithmetic is the reference's: its stored non-zeros applied in ascending
// k; a row that lacks an index of the union multiplies the table row by its +0.0 and adds the signed
// zero, which is the identity on a running sum that is never -0.0 (the same identity the sparse skip
// of the one-row kernel rests on).
//   k_imdct_plan   one workgroup per (group, channel): the rows are dequantised into LDS, the ascending
//                  union of their indices is built and written to global memory as 64-byte records
//                  {8 coefficients (+0.0 = absent), byte offset of the table row of entry j+2}, with a
//                  header {n_u, live rows, offsets of entries 0 and 1, dense flag, rows of raw frames}
//                  and the unit's work for k_imdct_order.  Nothing here depends on where the blocks
//                  go: the records of a launch stay valid for as long as the context holds the
//                  stream's rows, and a repeated decode skips this kernel (glc_api.hip launch_d1).
//   k_imdct_order  ranks the units of a launch by work and deals them over the CUs (speed only).
//   k_imdct_apply  no LDS, no barrier: each wave owns 8 rows x 512 outputs.  The record of the next
//                  entry arrives by scalar loads (s_load_dwordx8 + s_load_dword) a whole entry ahead;
//                  the coefficient pairs feed v_pk_mul_f32 straight from SGPRs (lane-broadcast by
//                  op_sel); the table row of the entry after next is in flight by
//                  global_load_dwordx4 from an SGPR base.  The vector ALU executes the 64 packed
//                  multiplies / adds of an entry and nothing else; a row pair whose two coefficients
//                  are both absent is skipped by a scalar branch (SKIP), so groups that share few
//                  indices degrade gracefully instead of needing a second code path.
// ------------------------------------------------------------------------------------------
typedef float d1x2 __attribute__((ext_vector_type(2)));
typedef float d1x4 __attribute__((ext_vector_type(4)));
typedef unsigned d1u8 __attribute__((ext_vector_type(8)));
typedef unsigned d1u2 __attribute__((ext_vector_type(2)));
constexpr unsigned kPlanRecDwords = 16;       // 64-byte records
constexpr unsigned kPlanHdrDwords = 8;        // {n_u, live rows, table offsets of the first <= 4 entries, dense flag, pad}
constexpr unsigned kPlanRecCap = kHopI + 8;   // per group: the union holds <= 1024 entries; the apply loop reads a few past

__global__ __launch_bounds__(256) void k_imdct_plan(DecodeRows rows, unsigned row_begin, unsigned n_frames, unsigned ch,
                                                     unsigned group_begin, unsigned ahead, unsigned *__restrict__ plan_hdr,
                                                     unsigned *__restrict__ plan_rec, unsigned *__restrict__ plan_work) {
  constexpr int G = 8;
  __shared__ __attribute__((aligned(16))) float s_c[kHopI * G];
  __shared__ unsigned s_mask[kHopI / 32];
  __shared__ unsigned short s_u[kHopI + 8];
  __shared__ unsigned s_wsum[4];
  const int tid = threadIdx.x;
  // blockIdx = (frame group - fg_begin) * ch + channel: the batch covers whole frame groups
  const unsigned c = blockIdx.x % ch;
  const unsigned fr0 = (group_begin + blockIdx.x / ch) * G;
  const int lane = tid & 63, w = tid >> 6;
  // The kernel is a chain of latencies, so everything independent is issued together: the metadata
  // of the 8 rows by 8 lanes at once (broadcast by shuffles afterwards) while the LDS is zeroed, then
  // the first 256 pairs of ALL rows before any of them is scattered.
  long long raw_l = -1;
  unsigned long long p0_l = 0;
  unsigned n_l = 0, valid_l = 0;
  float scale_l = 0.0f;
  if (lane < G && fr0 + lane < n_frames) {
    const unsigned m = row_begin + (fr0 + lane) * ch + c;
    raw_l = rows.row_raw[m];
    p0_l = rows.row_begin[m];
    n_l = min(rows.row_cnt[m], static_cast<unsigned>(kHopI));  // canonical lists hold <= 1024
    scale_l = fmaxf(rows.row_scale[m], 1e-12f);                  // :653
    valid_l = 1;
  }
  {
    d1x4 *z = reinterpret_cast<d1x4 *>(s_c);
    for (int i = tid; i < kHopI * G / 4; i += 256) z[i] = d1x4{0.f, 0.f, 0.f, 0.f};
  }
  if (tid < kHopI / 32) s_mask[tid] = 0u;
  unsigned live = 0, rawm = 0;
  unsigned long long p0[G];
  unsigned n[G];
  float scale[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const unsigned valid = __shfl(valid_l, g);
    const long long raw_off = __shfl(raw_l, g);
    p0[g] = __shfl(p0_l, g);
    n[g] = __shfl(n_l, g);
    scale[g] = __shfl(scale_l, g);
    if (valid && raw_off >= 0) rawm |= 1u << g;
    else if (valid) live |= 1u << g;
    if (!(live & (1u << g))) n[g] = 0;
  }
  unsigned pr[G];
#pragma unroll
  for (int g = 0; g < G; ++g) pr[g] = static_cast<unsigned>(tid) < n[g] ? rows.pairs[p0[g] + tid] : 0xFFFFu;
  __syncthreads();  // LDS zeroed
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const unsigned idx = pr[g] & 0xFFFFu;
    if (idx < static_cast<unsigned>(kHopI)) {  // 0xFFFF = no pair for this thread (also what a reader must ignore, :660)
      const short q = static_cast<short>(pr[g] >> 16);
      s_c[idx * G + g] = mul_rn(static_cast<float>(q) / 32768.0f, scale[g]);  // :663
      atomicOr(&s_mask[idx >> 5], 1u << (idx & 31));
    }
    for (unsigned j = tid + 256; j < n[g]; j += 256) {  // lists longer than 256 entries
      const unsigned p = rows.pairs[p0[g] + j];
      const unsigned k = p & 0xFFFFu;
      if (k < static_cast<unsigned>(kHopI)) {
        s_c[k * G + g] = mul_rn(static_cast<float>(static_cast<short>(p >> 16)) / 32768.0f, scale[g]);
        atomicOr(&s_mask[k >> 5], 1u << (k & 31));
      }
    }
  }
  __syncthreads();
  // ascending union list: thread t owns bins 4t..4t+3; exclusive scan of the per-thread counts
  const unsigned nib = (s_mask[tid >> 3] >> ((tid & 7) * 4)) & 0xFu;
  const unsigned cnt = __popc(nib);
  unsigned incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_wsum[w] = incl;
  __syncthreads();
  unsigned base = 0, n_u = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned v = s_wsum[i];
    if (i < w) base += v;
    n_u += v;
  }
  {
    unsigned pos = base + incl - cnt;
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (nib & (1u << b)) s_u[pos++] = static_cast<unsigned short>(tid * 4 + b);
  }
  __syncthreads();
  if (tid < 8) s_u[n_u + tid] = n_u ? s_u[n_u - 1] : static_cast<unsigned short>(0);  // run-ahead reads stay valid
  __syncthreads();
  unsigned *rec = plan_rec + static_cast<size_t>(blockIdx.x) * kPlanRecCap * kPlanRecDwords;
  for (unsigned j = tid; j < n_u; j += 256) {
    const unsigned k = s_u[j];
    d1x4 *dst = reinterpret_cast<d1x4 *>(rec + static_cast<size_t>(j) * kPlanRecDwords);
    dst[0] = *reinterpret_cast<const d1x4 *>(&s_c[k * G]);
    dst[1] = *reinterpret_cast<const d1x4 *>(&s_c[k * G + 4]);
    rec[static_cast<size_t>(j) * kPlanRecDwords + 8] = static_cast<unsigned>(s_u[j + ahead]) << 13;  // table row bytes = 8192
  }
  if (tid < 8) {
    unsigned *h = plan_hdr + static_cast<size_t>(blockIdx.x) * kPlanHdrDwords;
    // h[6]: 1 when the rows share most of their indices (union <= 2x the mean list), see k_imdct_apply
    unsigned total = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) total += n[g];
    const unsigned dense = n_u * G <= 2u * total ? 1u : 0u;
    // h[7]: rows of raw frames (written by k_imdct_raw_rows on every launch: the plan may outlive one, the blocks do not)
    h[tid] = tid == 0 ? n_u : tid == 1 ? live : tid < 2 + ahead ? static_cast<unsigned>(s_u[tid - 2]) << 13 : tid == 6 ? dense : tid == 7 ? rawm : 0u;
    // work of the unit for the placement below: packed operations it issues (stored non-zeros) plus
    // the per-entry overhead of walking its union (record fetch, table row)
    if (tid == 0) plan_work[blockIdx.x] = total + n_u + (rawm ? 64u : 0u);
  }
}

// Rows of raw frames of a launch (streams that have any): read as if interleaved (Q1), /32767, no window
// (Q2) - src/codec.rs:629-640.  One workgroup per row; rows of compressed frames return at once.  Its
// own kernel so that the apply kernel stays free of the division's fused expansion (tools/check_isa.py)
// and a kept plan (launch_d1) never leaves these blocks unwritten.
__global__ __launch_bounds__(256) void k_imdct_raw_rows(DecodeRows rows, unsigned row_begin, unsigned M, unsigned ch,
                                                         float *__restrict__ blocks) {
  const unsigned r = blockIdx.x;
  if (r >= M) return;
  const unsigned m = row_begin + r;
  const long long raw_off = rows.row_raw[m];
  if (raw_off < 0) return;
  const unsigned c = m % ch;
  const unsigned long long raw_len = rows.row_raw_len[m];
  const short *raw = rows.raw_pool + raw_off;
  float *out = blocks + static_cast<size_t>(r) * kFrameI;
  for (int i = threadIdx.x; i < kFrameI; i += 256) {
    const unsigned long long si = static_cast<unsigned long long>(i) * ch + c;
    float v = 0.0f;
    if (si < raw_len) v = static_cast<float>(raw[si]) / 32767.0f;
    out[i] = v;
  }
}

// Placement of the units of one launch (speed only).  All units of a launch of <= 1024 are resident at
// once, four to a CU, and the dispatcher deals an empty chip so that blocks b, b + 256, b + 512, b + 768
// share a CU (measured: tools/d1_tune.hip prints the hardware ids).  The kernel ends when its slowest CU
// does, so units are ranked by work (descending; ties by index, which makes the ranks a permutation)
// and dealt in a snake over the 256 CU slots: heaviest with lightest.  Units beyond the first 1024
// follow in descending order (they are dispatched as earlier ones finish: longest first).  A few
// microseconds, amortised over the decodes that reuse the plan (tools/d1_tune.hip: apply 80.4 -> 76.3 us
// at config 2).
constexpr unsigned kOrderMaxUnits = 4096;
// One WAVE per unit: its 64 lanes compare the unit's key with all n keys (each lane a strided share,
// read straight from L2), a wave reduction gives the rank.  (The first version - one THREAD per unit, every
// key compared from LDS - took 25 us for 1024 units: four workgroups, one wave per SIMD, 6 K vector
// instructions each at the single-wave issue rate.)
__global__ __launch_bounds__(256) void k_imdct_order(const unsigned *__restrict__ plan_work, unsigned n_units,
                                                      unsigned *__restrict__ order, unsigned ch, unsigned neighbours) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned u = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (u >= n_units) return;
  if (neighbours) {
    // (include/glc_debug.h variant 5, measurement only) the units that share a CU are consecutive frame
    // groups of one channel - similar unions, similar pace: they find each other's table rows in the CU's L1
    const unsigned n_fg = n_units / ch, rounds = n_units >> 8;
    const unsigned fg = u / ch, c = u - fg * ch;
    const unsigned v = c * n_fg + fg;  // channel-major
    if (lane == 0) order[(v % rounds) * 256u + v / rounds] = u;
    return;
  }
  const unsigned mine = plan_work[u];
  // rank = units with more work, or equal work and a lower index (ties by index make the ranks a permutation)
  unsigned cnt = 0;
  for (unsigned j = lane; j < n_units; j += 64u) {
    const unsigned k = plan_work[j];
    cnt += (k > mine || (k == mine && j < u)) ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  const unsigned rank = cnt;
  unsigned pos = rank;
  if (rank < 1024u) {
    const unsigned round = rank >> 8, p = rank & 255u;
    const unsigned in_round = min(256u, min(n_units, 1024u) - (round << 8));  // the last round may be partial
    pos = (round << 8) + ((round & 1u) ? in_round - 1u - p : p);
  }
  if (lane == 0) order[pos] = u;
}

// rows (r, r+1) x 8 columns, coefficient pair in SGPRs: the same instruction block as k1::mac2rows
__device__ __forceinline__ void d1_mac2rows_s(d1x2 (&c0)[4], d1x2 (&c1)[4], d1u2 a, d1x2 b0, d1x2 b1, d1x2 b2, d1x2 b3) {
  d1x2 t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "v_pk_mul_f32 %8, %16, %17 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %9, %16, %18 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %10, %16, %19 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %11, %16, %20 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %12, %16, %17 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %13, %16, %18 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %14, %16, %19 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %15, %16, %20 op_sel:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, %8\n\t"
      "v_pk_add_f32 %1, %1, %9\n\t"
      "v_pk_add_f32 %2, %2, %10\n\t"
      "v_pk_add_f32 %3, %3, %11\n\t"
      "v_pk_add_f32 %4, %4, %12\n\t"
      "v_pk_add_f32 %5, %5, %13\n\t"
      "v_pk_add_f32 %6, %6, %14\n\t"
      "v_pk_add_f32 %7, %7, %15"
      : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]),
        "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
      : "s"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

// The same block with a scalar branch around each row's half: a row that lacks the entry (+0.0, bits
// 0) costs two scalar instructions instead of 4 + 4 packed ones.  The branches live INSIDE the asm:
// written as C++ control flow around two asm blocks, hipcc re-allocates the 64 accumulators per arm
// and spills (128 VGPRs + scratch, 156 us instead of 108).
__device__ __forceinline__ void d1_mac2rows_fine_s(d1x2 (&c0)[4], d1x2 (&c1)[4], d1u2 a, d1x2 b0, d1x2 b1, d1x2 b2, d1x2 b3) {
  d1x2 t0, t1, t2, t3;
  asm volatile(
      "s_cmp_lg_u32 %13, 0\n\t"
      "s_cbranch_scc0 1f\n\t"
      "v_pk_mul_f32 %8, %12, %15 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %9, %12, %16 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %10, %12, %17 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %11, %12, %18 op_sel_hi:[0,1]\n\t"
      "v_pk_add_f32 %0, %0, %8\n\t"
      "v_pk_add_f32 %1, %1, %9\n\t"
      "v_pk_add_f32 %2, %2, %10\n\t"
      "v_pk_add_f32 %3, %3, %11\n"
      "1:\n\t"
      "s_cmp_lg_u32 %14, 0\n\t"
      "s_cbranch_scc0 2f\n\t"
      "v_pk_mul_f32 %8, %12, %15 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %9, %12, %16 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %10, %12, %17 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %11, %12, %18 op_sel:[1,0]\n\t"
      "v_pk_add_f32 %4, %4, %8\n\t"
      "v_pk_add_f32 %5, %5, %9\n\t"
      "v_pk_add_f32 %6, %6, %10\n\t"
      "v_pk_add_f32 %7, %7, %11\n"
      "2:"
      : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]),
        "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "s"(a), "s"(a.x), "s"(a.y), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
      : "scc");
}

// Two table rows per wave in registers: the one being applied and the next, in flight (four were
// measured and bought nothing: tools/d1_tune.hip, profiles/r02_d1_*).
template <bool SKIP, bool PRIO = true, bool FINE = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_imdct_apply(DeviceTables tb, const unsigned *__restrict__ plan_hdr, const unsigned *__restrict__ plan_rec,
                   const unsigned *__restrict__ order, unsigned n_frames, unsigned ch, unsigned group_begin, unsigned n_units,
                   float *__restrict__ blocks) {
  constexpr int G = 8, R = 2;
  // block -> unit (frame group, channel) of this batch.  (Speed only.)  All units of a launch of
  // <= 1024 are resident at once, four to a CU, and the dispatcher deals an empty chip so that blocks
  // b, b + 256, b + 512, b + 768 share a CU (measured: tools/d1_tune.hip prints the hardware ids).
  // Channels can differ in how many coefficients they keep (config 2: 985 against 1146 stored
  // non-zeros + union entries per unit), so the natural order - which hands a CU four units of ONE
  // channel whenever 256 % ch == 0 - leaves the slowest CU 14 % above the mean; rotating the channel
  // by the round (b / 256) of the frame group's first block gives every CU all channels (7 % above;
  // a full sort by work would reach 4 % but costs more than it returns: 10 us in the plan kernel).
  // ... or, when the launch has been ranked by work (k_imdct_order), the table says which unit this block takes.
  if (blockIdx.x >= n_units) return;
  unsigned fg, c;
  if (order) {
    const unsigned u = __builtin_amdgcn_readfirstlane(order[blockIdx.x]);
    fg = u / ch;
    c = u - fg * ch;
  } else {
    fg = blockIdx.x / ch;
    c = (blockIdx.x - fg * ch + ((fg * ch) >> 8)) % ch;
  }
  const unsigned local = fg * ch + c;
  const unsigned fr0 = (group_begin + fg) * G;
  const unsigned *hdr = plan_hdr + static_cast<size_t>(local) * kPlanHdrDwords;
  const unsigned n_u = __builtin_amdgcn_readfirstlane(hdr[0]);
  const unsigned live = __builtin_amdgcn_readfirstlane(hdr[1]);
  if (!live) return;
  // The four waves of a SIMD are arbitrated oldest-first, so left alone they finish one after the
  // other and the last one runs by itself, with nobody to fill its scalar and wait slots.  Each wave
  // of a dense unit therefore lowers its own issue priority as it gets through its union (3 in the
  // first quarter ... 0 in the last): whoever is furthest behind goes first, the waves of a SIMD
  // progress together and keep covering each other's stalls to the end (config 2: D1 106 -> 92 us
  // with the per-row skip in place, 80 -> 76 us with fully shared indices; debug variant 3 is the
  // kernel without it).  Only for units whose rows share most indices, flagged by the plan kernel.
  const unsigned dense_unit = PRIO ? __builtin_amdgcn_readfirstlane(hdr[6]) : 0u;
  const unsigned q1 = n_u >> 2, q2 = n_u >> 1, q3 = q1 + q2;
  unsigned prio_next = dense_unit ? q1 : 0xFFFFFFFFu, prio_level = 0;  // scalar state: one compare per 4 entries
  // units that are not dense keep the top priority throughout: they are the long ones (the broadband
  // frames at a stream's edges double their union), and in a launch where every unit is like that
  // equal priorities change nothing
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  const unsigned *rec = plan_rec + static_cast<size_t>(local) * kPlanRecCap * kPlanRecDwords;
  const unsigned col0 = static_cast<unsigned>(threadIdx.x) * 8u;  // 8 consecutive outputs per lane
  const unsigned lane_off = col0 * 4u;

  d1x2 acc[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int h = 0; h < 4; ++h) acc[g][h] = d1x2{0.f, 0.f};
  d1x4 t_lo[R], t_hi[R];
  d1u8 ca, cb, cc, cd;        // records of the pair being applied (ca, cb) and of the next pair (cc, cd)
  unsigned ka, kb, kc, kd;
  auto issue_tab = [&](d1x4 &lo, d1x4 &hi, unsigned koff) {
    const unsigned long long row = reinterpret_cast<unsigned long long>(tb.cos) + koff;  // scalar ALU
    asm volatile(
        "global_load_dwordx4 %0, %2, %3\n\t"
        "global_load_dwordx4 %1, %2, %3 offset:16"
        : "=&v"(lo), "=&v"(hi)
        : "v"(lane_off), "s"(row)
        : "memory");
  };
#pragma unroll
  for (int r = 0; r < R; ++r) issue_tab(t_lo[r], t_hi[r], __builtin_amdgcn_readfirstlane(hdr[2 + r]));  // entries 0, 1
  // Records are fetched TWO entries at a time, a whole pair of entries ahead: scalar loads return out
  // of order, so the only safe wait is lgkmcnt(0), which covers everything issued so far - fetching
  // every other entry doubles the time each fetch has before it is waited for (measured on the
  // config-2 batch: 107 -> 95 us, profiles/r02_d1_tune_real_rows.txt).  Scalar and vector operands
  // sit in SEPARATE asm statements: LLVM treats every output of an asm that has one VGPR output as
  // divergent, and a "divergent" row offset would be added on the vector ALU.
#define GLC_D1_FETCH2(C0, K0, C1, K1, J)                                                                          \
  do {                                                                                                          \
    const unsigned *nrec = rec + static_cast<size_t>(J) * kPlanRecDwords;                                        \
    asm volatile(                                                                                               \
        "s_load_dwordx8 %0, %4, 0x0\n\ts_load_dword %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_load_dword %3, %4, 0x60" \
        : "=&s"(C0), "=&s"(K0), "=&s"(C1), "=&s"(K1)                                                             \
        : "s"(nrec)                                                                                             \
        : "memory");                                                                                            \
  } while (0)
#define GLC_D1_WAIT2(C0, K0, C1, K1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(C0), "+s"(K0), "+s"(C1), "+s"(K1)::"memory")
  // One entry.  S: its table slot; CC: its 8 coefficients; KC: table offset of entry J + R, whose row
  // refills slot S.  Of the table loads only those of the R - 1 following entries may be in flight.
#define GLC_D1_PAIR(S, P, A0, A1)                                                                                \
  do {                                                                                                          \
    if (SKIP && FINE) {                                                                                          \
      if (P.x | P.y) d1_mac2rows_fine_s(acc[A0], acc[A1], P, t_lo[S].xy, t_lo[S].zw, t_hi[S].xy, t_hi[S].zw);      \
    } else if (!SKIP || (P.x | P.y)) {                                                                           \
      d1_mac2rows_s(acc[A0], acc[A1], P, t_lo[S].xy, t_lo[S].zw, t_hi[S].xy, t_hi[S].zw);                         \
    }                                                                                                           \
  } while (0)
#define GLC_D1_ENTRY(S, CC, KC)                                                                                  \
  do {                                                                                                          \
    asm volatile("s_waitcnt vmcnt(2)" : "+v"(t_lo[S]), "+v"(t_hi[S])::"memory");                                  \
    const d1u2 p0 = CC.s01, p1 = CC.s23, p2 = CC.s45, p3 = CC.s67;                                               \
    GLC_D1_PAIR(S, p0, 0, 1);                                                                                    \
    GLC_D1_PAIR(S, p1, 2, 3);                                                                                    \
    GLC_D1_PAIR(S, p2, 4, 5);                                                                                    \
    GLC_D1_PAIR(S, p3, 6, 7);                                                                                    \
    issue_tab(t_lo[S], t_hi[S], KC);                                                                             \
  } while (0)
  GLC_D1_FETCH2(ca, ka, cb, kb, 0);
  unsigned j = 0;
#pragma unroll 1
  for (; j + 4 <= n_u; j += 4) {
    if (j >= prio_next) {  // crossed a quarter of the union: three times per wave
      ++prio_level;
      if (prio_level == 1) __builtin_amdgcn_s_setprio(2);
      else if (prio_level == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      prio_next = prio_level == 1 ? q2 : prio_level == 2 ? q3 : 0xFFFFFFFFu;
    }
    GLC_D1_WAIT2(ca, ka, cb, kb);
    GLC_D1_FETCH2(cc, kc, cd, kd, j + 2);
    GLC_D1_ENTRY(0, ca, ka);
    GLC_D1_ENTRY(1, cb, kb);
    GLC_D1_WAIT2(cc, kc, cd, kd);
    GLC_D1_FETCH2(ca, ka, cb, kb, j + 4);
    GLC_D1_ENTRY(0, cc, kc);
    GLC_D1_ENTRY(1, cd, kd);
  }
  if (j < n_u) {  // 1..3 entries left; (ca, cb) hold entries j, j + 1
    GLC_D1_WAIT2(ca, ka, cb, kb);
    GLC_D1_FETCH2(cc, kc, cd, kd, j + 2);
    GLC_D1_ENTRY(0, ca, ka);
    if (j + 1 < n_u) {
      GLC_D1_ENTRY(1, cb, kb);
      if (j + 2 < n_u) {
        GLC_D1_WAIT2(cc, kc, cd, kd);
        GLC_D1_ENTRY(0, cc, kc);
      }
    }
  }
#undef GLC_D1_ENTRY
#undef GLC_D1_PAIR
#undef GLC_D1_FETCH2
#undef GLC_D1_WAIT2
  // drain the run-ahead loads before their registers are reused
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ca), "+s"(cb), "+s"(cc), "+s"(cd), "+s"(ka), "+s"(kb), "+s"(kc), "+s"(kd)::"memory");
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(t_lo[0]), "+v"(t_hi[0]), "+v"(t_lo[1]), "+v"(t_hi[1])::"memory");

  const d1x4 w0 = *reinterpret_cast<const d1x4 *>(tb.window + col0), w1 = *reinterpret_cast<const d1x4 *>(tb.window + col0 + 4);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (!(live & (1u << g))) continue;
    float *out = blocks + static_cast<size_t>((fr0 + g) * ch + c) * kFrameI + col0;
    d1x4 o0, o1;  // out[i] = s*norm (:388) then *= window[i] (:674)
    o0.x = mul_rn(mul_rn(acc[g][0].x, tb.norm), w0.x); o0.y = mul_rn(mul_rn(acc[g][0].y, tb.norm), w0.y);
    o0.z = mul_rn(mul_rn(acc[g][1].x, tb.norm), w0.z); o0.w = mul_rn(mul_rn(acc[g][1].y, tb.norm), w0.w);
    o1.x = mul_rn(mul_rn(acc[g][2].x, tb.norm), w1.x); o1.y = mul_rn(mul_rn(acc[g][2].y, tb.norm), w1.y);
    o1.z = mul_rn(mul_rn(acc[g][3].x, tb.norm), w1.z); o1.w = mul_rn(mul_rn(acc[g][3].y, tb.norm), w1.w);
    *reinterpret_cast<d1x4 *>(out) = o0;
    *reinterpret_cast<d1x4 *>(out + 4) = o1;
  }
}

// ------------------------------------------------------------------------------------------
// D2: overlap-add + interleave.  blocks holds frames [blk_frame0, ...) as [frame][ch][2048];
// hop h = second half of frame h-1 (+0.0 before the first frame) + first half of frame h; the
// hop after the last frame is the bare overlap tail (no add, src/codec.rs:722-729).
// ------------------------------------------------------------------------------------------
// One workgroup row (blockIdx.y) per hop, a float4 of interleaved output per thread: 32-bit index
// arithmetic (CH = 1 / 2 / 4 / 8: shifts; CH = 0: one 32-bit division per sample), coalesced 16-byte
// stores, each block plane read in runs of consecutive samples.  (Round 2's kernel walked the output
// with a 64-bit grid-stride index: two 64-bit divisions per sample - 20 us at config 2, as long as the
// 100 MB it moves take at Infinity-Cache speed.)
template <int CH, bool VEC>
__global__ __launch_bounds__(256) void k_overlap_add(const float *__restrict__ blocks, long long blk_frame0,
                                                      unsigned long long n_frames, unsigned ch,
                                                      unsigned long long hop_begin, float *__restrict__ out) {
  const unsigned per_hop = static_cast<unsigned>(kHopI) * ch;
  const unsigned o0 = (blockIdx.x * 256u + threadIdx.x) * 4u;  // first of this thread's 4 outputs inside the hop
  if (o0 >= per_hop) return;
  const unsigned long long h = hop_begin + blockIdx.y;
  const bool has_prev = h >= 1, has_cur = h < n_frames;
  // frame h-1 (second half) and frame h (first half) of the ring: [slot][ch][2048]
  const float *prev = blocks + (static_cast<size_t>(static_cast<long long>(h) - 1 - blk_frame0) * ch) * kFrameI + kHopI;
  const float *cur = blocks + (static_cast<size_t>(static_cast<long long>(h) - blk_frame0) * ch) * kFrameI;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned o = o0 + e;
    unsigned i, c;
    if constexpr (CH == 1) i = o, c = 0;
    else if constexpr (CH == 2) i = o >> 1, c = o & 1u;
    else if constexpr (CH == 4) i = o >> 2, c = o & 3u;
    else if constexpr (CH == 8) i = o >> 3, c = o & 7u;
    else i = o / ch, c = o - i * ch;
    const size_t at = static_cast<size_t>(c) * kFrameI + i;
    const float p = has_prev ? prev[at] : 0.0f;  // overlap starts as +0.0, :601
    v[e] = has_cur ? add_rn(p, cur[at]) : p;      // :695 / the bare tail, :727
  }
  float *dst = out + static_cast<size_t>(blockIdx.y) * per_hop + o0;
  if constexpr (VEC) {
    *reinterpret_cast<float4 *>(dst) = float4{v[0], v[1], v[2], v[3]};
  } else {  // a destination that is only 4-byte aligned (glc_decode_range_device takes any device pointer)
    dst[0] = v[0], dst[1] = v[1], dst[2] = v[2], dst[3] = v[3];
  }
}

// ------------------------------------------------------------------------------------------
// P1-P3: device-side compaction of frame records into the compact blob (glc_common.h
// CompactLayout), so that the host boundary and the multi-GPU gather move (u16 idx, i16 q) pairs
// instead of dense 1024-bin rows.
//   P1  per row: pairs it contributes (nnz, 0 for rows of raw frames); exclusive scan inside
//       blocks of 1024 rows; per-row scale and count, per-frame raw flag into the blob
//   P2  one workgroup: exclusive scans of the block totals (pairs, raw rows), then the blob header
//       they determine and the zeroed alignment gap in front of the raw section
//   P3  one wave per row: ballot + popcount prefix keeps ascending k (src/codec.rs:303-306);
//       planes of raw-frame rows go to the raw section, which starts behind the pairs
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_scan_rows(const unsigned char *__restrict__ records, unsigned M,
                                                         unsigned ch, unsigned long long rec_bytes,
                                                         unsigned *__restrict__ loc,
                                                         unsigned long long *__restrict__ blk,
                                                         unsigned long long *__restrict__ blk_raw,
                                                         float *__restrict__ scales, unsigned *__restrict__ cnt,
                                                         unsigned char *__restrict__ is_raw) {
  // one 32-bit word scans both counts: low 21 bits = pairs (<= 1024*1024 per block), high 11 =
  // rows of raw frames (<= 1024 per block)
  __shared__ unsigned s_part[256];
  const unsigned base = blockIdx.x * 1024u + threadIdx.x * 4u;
  unsigned v[4], sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned m = base + j;
    unsigned n = 0;
    if (m < M) {
      const unsigned frame = m / ch, c = m % ch;
      const unsigned char *rec = records + static_cast<size_t>(frame) * rec_bytes;
      const unsigned raw = *reinterpret_cast<const unsigned *>(rec);
      const unsigned nnz = min(*reinterpret_cast<const unsigned *>(rec + 8 + 8 * c + 4), static_cast<unsigned>(kHopI));
      n = raw ? (1u << 21) : nnz;
      scales[m] = *reinterpret_cast<const float *>(rec + 8 + 8 * c);
      cnt[m] = raw ? 0u : nnz;
      if (c == 0) is_raw[frame] = raw ? 1 : 0;
    }
    v[j] = sum;  // exclusive within the thread
    sum += n;
  }
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan of the 256 thread sums
    unsigned t = threadIdx.x >= static_cast<unsigned>(off) ? s_part[threadIdx.x - off] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += t;
    __syncthreads();
  }
  const unsigned before = threadIdx.x ? s_part[threadIdx.x - 1] : 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < M) loc[base + j] = before + v[j];
  if (threadIdx.x == 255) {
    blk[blockIdx.x] = s_part[255] & 0x1FFFFFu;
    blk_raw[blockIdx.x] = s_part[255] >> 21;
  }
}

// One workgroup: exclusive scans of the per-block pair counts and raw-row counts, their totals, and
// the blob header they determine (three dependent launches of a few microseconds each before).
__global__ __launch_bounds__(1024) void k_pack_scan_blocks(unsigned long long *__restrict__ blk,
                                                            unsigned long long *__restrict__ blk_raw, unsigned n,
                                                            unsigned long long *__restrict__ totals, unsigned ch,
                                                            unsigned long long n_frames, unsigned long long o_pairs,
                                                            unsigned char *__restrict__ blob) {
  __shared__ unsigned long long s[1024];
  unsigned long long sums[2];
#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    unsigned long long *v = which ? blk_raw : blk;
    unsigned long long carry = 0;
    for (unsigned b0 = 0; b0 < n; b0 += 1024) {
      const unsigned i = b0 + threadIdx.x;
      const unsigned long long mine = i < n ? v[i] : 0ull;
      s[threadIdx.x] = mine;
      __syncthreads();
      for (int off = 1; off < 1024; off <<= 1) {
        unsigned long long t = threadIdx.x >= static_cast<unsigned>(off) ? s[threadIdx.x - off] : 0ull;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
      }
      if (i < n) v[i] = carry + s[threadIdx.x] - mine;  // exclusive
      const unsigned long long chunk_total = s[1023];
      __syncthreads();
      carry += chunk_total;
    }
    sums[which] = carry;
  }
  const unsigned long long n_pairs = sums[0], n_raw_rows = sums[1];
  const unsigned long long pairs_end = o_pairs + 4ull * n_pairs;
  const unsigned long long raw_off = (pairs_end + 63ull) & ~63ull;
  if (threadIdx.x == 0) {
    totals[0] = n_pairs;
    totals[1] = n_raw_rows;
    unsigned long long *h = reinterpret_cast<unsigned long long *>(blob);
    h[0] = 0x42434C47ull | (static_cast<unsigned long long>(ch) << 32);  // magic, channels
    h[1] = n_frames;
    h[2] = n_pairs;
    h[3] = n_raw_rows;
    h[4] = raw_off + n_raw_rows * 4096ull;
    h[5] = h[6] = h[7] = 0ull;
  }
  if (threadIdx.x < 64 && pairs_end + threadIdx.x < raw_off) blob[pairs_end + threadIdx.x] = 0;  // deterministic padding
}

__global__ __launch_bounds__(256) void k_pack_rows(const unsigned char *__restrict__ records, unsigned M,
                                                    unsigned ch, unsigned long long rec_bytes,
                                                    unsigned long long hdr_bytes, const unsigned *__restrict__ loc,
                                                    const unsigned long long *__restrict__ blk,
                                                    const unsigned long long *__restrict__ blk_raw,
                                                    const unsigned long long *__restrict__ totals,
                                                    const unsigned *__restrict__ cnt, unsigned long long o_pairs,
                                                    unsigned char *__restrict__ blob) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned m = blockIdx.x * 4 + w;
  if (m >= M) return;
  const unsigned l = loc[m];
  const unsigned long long off = blk[m >> 10] + (l & 0x1FFFFFu);
  const unsigned frame = m / ch, c = m % ch;
  const unsigned char *rec = records + static_cast<size_t>(frame) * rec_bytes;
  const short *qrow = reinterpret_cast<const short *>(rec + hdr_bytes) + static_cast<size_t>(c) * kFrameI;
  if (*reinterpret_cast<const unsigned *>(rec)) {
    // row of a raw frame: its 2048-sample plane goes to the raw section (planar order == row order, Q1)
    const unsigned long long raw_off = (o_pairs + 4ull * totals[0] + 63ull) & ~63ull;
    const unsigned long long rrow = blk_raw[m >> 10] + (l >> 21);
    const short4 *src = reinterpret_cast<const short4 *>(qrow);
    short4 *dst = reinterpret_cast<short4 *>(blob + raw_off + rrow * (kFrameI * 2ull));
    for (int i = lane; i < kFrameI / 4; i += 64) dst[i] = src[i];
    return;
  }
  unsigned *dst = reinterpret_cast<unsigned *>(blob + o_pairs) + off;
  const unsigned room = cnt[m];  // a record whose nnz field disagrees with its row cannot write past its slot
  unsigned done = 0;
  for (int k0 = 0; k0 < kHopI; k0 += 64) {
    const short q = qrow[k0 + lane];
    const unsigned long long mask = __ballot(q != 0);
    if (q != 0) {
      const unsigned pos = done + __popcll(mask & ((1ull << lane) - 1ull));
      if (pos < room)
        dst[pos] = static_cast<unsigned>(k0 + lane) | (static_cast<unsigned>(static_cast<unsigned short>(q)) << 16);
    }
    done += __popcll(mask);
  }
  // fewer non-zeros than the nnz field claims: fill the rest of the slot (idx 0xFFFF is ignored by
  // every reader, src/codec.rs:660) so the blob never carries uninitialised bytes
  for (unsigned pos = done + lane; pos < room; pos += 64) dst[pos] = 0xFFFFu;
}

// ------------------------------------------------------------------------------------------
// Clock probe (include/glc_debug.h, measurement only): ONE wave that sleeps beside whatever else runs
// on the device and reads the shader-clock counter (s_memtime) against the constant 100 MHz counter
// (s_memrealtime) over `ticks_100mhz`: shader cycles / reference ticks x 100 MHz = the clock the chip
// held over that window (MI355X_MICROARCH.md, DVFS give-back item 6).  It executes a handful of scalar
// instructions per microsecond; nothing of the product reads what it writes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long ticks_100mhz, unsigned long long *__restrict__ out) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks_100mhz) {
    __builtin_amdgcn_s_sleep(32);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
}

}  // namespace

hipError_t launch_clock_probe(uint64_t ticks_100mhz, uint64_t *out, hipStream_t s) {
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, static_cast<unsigned long long>(ticks_100mhz),
                     reinterpret_cast<unsigned long long *>(out));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------- launchers

// Which kernel takes a launch of 3584 / 4096 rows or more (the rest of the dispatch is by row count alone):
//   st 16 waves  256 x 128 tiles, one 1024-thread workgroup per CU: a launch runs in rounds of 256 tiles;
//   st 8 waves   256 x 64 tiles, two 512-thread workgroups per CU: rounds of 512, and a CU with one
//                workgroup left finishes it in about 0.6 of a round.
// The 16-wave form is 2 % faster on full rounds (its staging and barrier cost nothing: k1_tune [abl]) and
// worse on a last round that is half empty - so it takes the launches whose last round of 32 row tiles is
// full or more than half full, the 8-wave form the rest (profiles/r03_k1_tune_st_*.txt).
namespace {
template <int NW>
hipError_t launch_st_ch(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M, float *coef,
                        hipStream_t s) {
  // PCM by one dwordx4 per lane and piece when the channel count divides the tile height, else one
  // dword per (row, sample).  PRIO: 8 waves - by quarter of the i loop (between a CU's two workgroups);
  // 16 waves - by distance from the last barrier (inside the workgroup).
  constexpr int P = NW == 8 ? 1 : 2;
  switch (pcm.ch) {
    case 1: return k1::launch_st<4, 1, P, 4, NW>(t, pcm, frame_begin, M, coef, s);
    case 2: return k1::launch_st<4, 2, P, 4, NW>(t, pcm, frame_begin, M, coef, s);
    case 4: return k1::launch_st<4, 4, P, 4, NW>(t, pcm, frame_begin, M, coef, s);
    case 8: return k1::launch_st<4, 8, P, 4, NW>(t, pcm, frame_begin, M, coef, s);
    default: return k1::launch_st<4, 0, P, 4, NW>(t, pcm, frame_begin, M, coef, s);
  }
}
hipError_t launch_dma_ch(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M, float *coef,
                         hipStream_t s) {
  switch (pcm.ch) {
    case 1: return k1::launch_dma<4, 1, 1>(t, pcm, frame_begin, M, coef, s);
    case 2: return k1::launch_dma<4, 2, 1>(t, pcm, frame_begin, M, coef, s);
    case 4: return k1::launch_dma<4, 4, 1>(t, pcm, frame_begin, M, coef, s);
    case 8: return k1::launch_dma<4, 8, 1>(t, pcm, frame_begin, M, coef, s);
    default: return k1::launch_dma<4, 0, 1>(t, pcm, frame_begin, M, coef, s);
  }
}
}  // namespace

hipError_t launch_mdct_forward(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin,
                               uint32_t M, float *coef, hipStream_t s, int variant, bool beside) {
  // Shapes measured with tools/k1_tune.hip.  The f32 VALU needs >= 4 waves per SIMD to approach its
  // issue rate, so every kernel for 4096 rows or more keeps 4 x 8 outputs per lane (32 accumulators) and
  // 16 waves per CU.
  // A clip of a few seconds is latency-bound by one wave's chain of 2048 dependent i-steps, not by
  // throughput, and the length of a step is the lane tile: cut it until every SIMD has a wave of its own
  // (glc_mdct_fwd.hpp k_mdct_fwd_small: 2 x 2 outputs per lane, then 2 x 4).  Measured against the 4 x 8
  // kernels (profiles/r03_k1_tune_short_clips.txt, r03_k1_tune_mid_sizes.txt): 172 rows 0.053-0.058 ms (4 x 8
  // tile: 0.19), 600 rows 0.10, 1024 rows 0.11, 2048 rows 0.20, 3072 rows 0.28; a launch of 256-row tiles costs
  // 0.28-0.34 ms however few rows it has (one workgroup's chain) and draws level at about 3500 rows - when
  // the channel count has a segment loader; with one dword per (row, sample) only at 4096.  (Rounds 1-3 used a
  // 64 x 128 kernel for 1793..4095 rows: 0.22 ms at 2048 rows, 0.34 at 3072 - slower than both neighbours when
  // a launch has the chip to itself; it keeps one job, below.)
  const bool seg = pcm.ch == 1 || pcm.ch == 2 || pcm.ch == 4 || pcm.ch == 8;
  if (M <= 640) return k1::launch_small<2>(t, pcm, frame_begin, M, coef, s);
  // beside: an opening round of glc_encode (2048 rows), which runs beside its neighbours' kernels on a second
  // stream.  Alone the 2 x 4 kernel is faster there (0.198 against 0.221 ms), but its 1024 workgroups fill every
  // CU four deep and two such launches get in each other's way; the 64 x 128 kernel of rounds 1-3 puts ONE
  // workgroup on each CU: glc_encode at config 2 1.00-1.02 ms against 1.05-1.08 (profiles/r03_encode_rounds_kernel.txt).
  if ((beside || variant == 4) && M > 1792 && M <= 2048) return k1::launch_sched<64, 128, 16, 4>(t, pcm, frame_begin, M, coef, s);
  if (M < (seg ? 3584u : 4096u)) return k1::launch_small<4>(t, pcm, frame_begin, M, coef, s);
  // variant (include/glc_debug.h glc_debug_set_mdct_variant): 0 = shipped, 1 = round 3's kernel, 2 / 3 = one form for every launch
  if (variant == 1) return launch_dma_ch(t, pcm, frame_begin, M, coef, s);
  if (variant == 2) return launch_st_ch<8>(t, pcm, frame_begin, M, coef, s);
  if (variant == 3) return launch_st_ch<16>(t, pcm, frame_begin, M, coef, s);
  const unsigned last_round = ((M + 255) / 256) % 32;  // row tiles in the last round of 32
  if (last_round == 0 || last_round > 16) return launch_st_ch<16>(t, pcm, frame_begin, M, coef, s);
  return launch_st_ch<8>(t, pcm, frame_begin, M, coef, s);
}

hipError_t launch_quantize(const DeviceTables &t, const float *coef, uint32_t M, uint32_t ch, const PcmView &pcm,
                           uint64_t frame_begin, uint8_t *records, hipStream_t s, bool *decided) {
  *decided = ch == 1 || ch == 2 || ch == 4;  // a frame's rows sit in one wave: K2 decides raw-vs-compressed itself
  if (M == 0) return hipSuccess;
  const unsigned long long hdr = ((8ull + 8ull * ch) + 15ull) & ~15ull;
  const unsigned long long rec = hdr + 2ull * kFrameI * ch;
  const dim3 grid((M + 4 * kQRows - 1) / (4 * kQRows));
  if (*decided)
    hipLaunchKernelGGL(k_quantize<true>, grid, dim3(256), 0, s, t, coef, M, ch, rec, hdr, pcm,
                       static_cast<long long>(frame_begin), records);
  else
    hipLaunchKernelGGL(k_quantize<false>, grid, dim3(256), 0, s, t, coef, M, ch, rec, hdr, pcm,
                       static_cast<long long>(frame_begin), records);
  return hipGetLastError();
}

hipError_t launch_decide_raw(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin,
                             uint32_t n_frames, uint8_t *records, hipStream_t s) {
  if (n_frames == 0) return hipSuccess;
  const unsigned long long hdr = ((8ull + 8ull * pcm.ch) + 15ull) & ~15ull;
  const unsigned long long rec = hdr + 2ull * kFrameI * pcm.ch;
  hipLaunchKernelGGL(k_decide_raw, dim3(n_frames), dim3(256), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), n_frames, rec, hdr, records);
  return hipGetLastError();
}

hipError_t launch_compact(const uint8_t *records, uint32_t M, uint32_t ch, uint64_t n_frames, uint32_t *loc,
                          uint64_t *blk, uint64_t *blk_raw, uint64_t *totals, uint8_t *blob, uint64_t o_israw,
                          uint64_t o_scale, uint64_t o_cnt, uint64_t o_pairs, hipStream_t s) {
  auto *t = reinterpret_cast<unsigned long long *>(totals);
  if (M == 0) {  // an empty range: the scans are over nothing, the header says so
    hipLaunchKernelGGL(k_pack_scan_blocks, dim3(1), dim3(1024), 0, s, reinterpret_cast<unsigned long long *>(blk),
                       reinterpret_cast<unsigned long long *>(blk_raw), 0u, t, ch, 0ull, static_cast<unsigned long long>(o_pairs), blob);
    return hipGetLastError();
  }
  const unsigned long long hdr = ((8ull + 8ull * ch) + 15ull) & ~15ull;
  const unsigned long long rec = hdr + 2ull * kFrameI * ch;
  const unsigned nblk = (M + 1023) / 1024;
  auto *b = reinterpret_cast<unsigned long long *>(blk);
  auto *br = reinterpret_cast<unsigned long long *>(blk_raw);
  hipLaunchKernelGGL(k_pack_scan_rows, dim3(nblk), dim3(256), 0, s, records, M, ch, rec, loc, b, br,
                     reinterpret_cast<float *>(blob + o_scale), reinterpret_cast<unsigned *>(blob + o_cnt),
                     blob + o_israw);
  hipLaunchKernelGGL(k_pack_scan_blocks, dim3(1), dim3(1024), 0, s, b, br, nblk, t, ch, static_cast<unsigned long long>(n_frames),
                     static_cast<unsigned long long>(o_pairs), blob);
  hipLaunchKernelGGL(k_pack_rows, dim3((M + 3) / 4), dim3(256), 0, s, records, M, ch, rec, hdr, loc, b, br, t,
                     reinterpret_cast<const unsigned *>(blob + o_cnt), static_cast<unsigned long long>(o_pairs), blob);
  return hipGetLastError();
}

uint64_t imdct_plan_bytes(uint32_t groups) {
  // headers | records | work keys | unit order
  return static_cast<uint64_t>(groups) * (4ull * kPlanHdrDwords + static_cast<uint64_t>(kPlanRecCap) * kPlanRecDwords * 4ull + 8ull);
}

hipError_t launch_imdct_rows(const DeviceTables &t, const DecodeRows &rows, uint32_t row_begin,
                             uint32_t M, uint32_t ch, float *blocks, hipStream_t s, int variant, void *plan,
                             uint32_t plan_groups, bool reuse_plan) {
  if (M == 0) return hipSuccess;
  // every caller decodes whole frames; the one-row kernel is the cross-check variant (glc_debug.h)
  if (variant == 1 || ch == 0 || M % ch != 0) {
    hipLaunchKernelGGL(k_imdct_rows, dim3(M), dim3(256), 0, s, t, rows, row_begin, M, ch, blocks);
    return hipGetLastError();
  }
  if (!plan || plan_groups < ch) return hipErrorInvalidValue;
  const uint32_t n_frames = M / ch;
  const uint32_t groups = (n_frames + 7) / 8;
  // plan + apply, in batches of whole frame groups (x all channels) through one workspace of
  // plan_groups (frame group, channel) units
  unsigned *hdr = static_cast<unsigned *>(plan);
  unsigned *rec = hdr + static_cast<size_t>(plan_groups) * kPlanHdrDwords;
  unsigned *work = rec + static_cast<size_t>(plan_groups) * kPlanRecCap * kPlanRecDwords;
  unsigned *order = work + plan_groups;
  const uint32_t fg_per_batch = plan_groups / ch;
  if (reuse_plan && groups > fg_per_batch) return hipErrorInvalidValue;  // only a one-batch launch leaves its plan behind
  if (rows.any_raw) hipLaunchKernelGGL(k_imdct_raw_rows, dim3(M), dim3(256), 0, s, rows, row_begin, M, ch, blocks);
  for (uint32_t fg0 = 0; fg0 < groups; fg0 += fg_per_batch) {
    const uint32_t n_fg = groups - fg0 < fg_per_batch ? groups - fg0 : fg_per_batch;
    const uint32_t n_units = n_fg * ch;
    const dim3 grid(n_units);
    // more units than CUs: rank them by work (fewer land one per CU whatever the order)
    const bool ranked = n_units > 256 && n_units <= kOrderMaxUnits;
    if (!reuse_plan) {
      hipLaunchKernelGGL(k_imdct_plan, grid, dim3(256), 0, s, rows, row_begin, n_frames, ch, fg0, 2u, hdr, rec, work);
      const unsigned neighbours = variant == 5 && n_units % 256 == 0 && n_units % ch == 0 ? 1u : 0u;
      if (ranked && variant != 6)
        hipLaunchKernelGGL(k_imdct_order, dim3((n_units + 3) / 4), dim3(256), 0, s, work, n_units, order, ch, neighbours);
    }
    const unsigned *ord = ranked && variant != 6 ? order : nullptr;
    if (variant == 2)
      hipLaunchKernelGGL(k_imdct_apply<false>, grid, dim3(256), 0, s, t, hdr, rec, ord, n_frames, ch, fg0, n_units, blocks);
    else if (variant == 4)
      hipLaunchKernelGGL((k_imdct_apply<true, true, false>), grid, dim3(256), 0, s, t, hdr, rec, ord, n_frames, ch, fg0, n_units, blocks);
    else if (variant == 3)
      hipLaunchKernelGGL((k_imdct_apply<true, false>), grid, dim3(256), 0, s, t, hdr, rec, ord, n_frames, ch, fg0, n_units, blocks);
    else
      hipLaunchKernelGGL(k_imdct_apply<true>, grid, dim3(256), 0, s, t, hdr, rec, ord, n_frames, ch, fg0, n_units, blocks);
  }
  return hipGetLastError();
}

hipError_t launch_overlap_add(const float *blocks, int64_t blk_frame0, uint64_t n_frames, uint32_t ch,
                              uint64_t hop_begin, uint64_t hop_end, float *out, hipStream_t s) {
  if (hop_end <= hop_begin) return hipSuccess;
  const unsigned per_hop = 1024u * ch;
  const unsigned bx = (per_hop / 4u + 255u) / 256u;  // float4s per hop over 256 threads
  // blockIdx.y is 16 bits wide: hops in slabs of 32768 (every caller stays far below: rounds of <= 4097 hops)
  for (uint64_t h0 = hop_begin; h0 < hop_end; h0 += 32768) {
    const unsigned nh = static_cast<unsigned>(hop_end - h0 < 32768 ? hop_end - h0 : 32768);
    const dim3 grid(bx, nh);
    float *o = out + (h0 - hop_begin) * per_hop;
    const long long f0 = static_cast<long long>(blk_frame0);
    const unsigned long long nf = n_frames, hb = h0;
    if (reinterpret_cast<uintptr_t>(o) & 15u) {
      hipLaunchKernelGGL((k_overlap_add<0, false>), grid, dim3(256), 0, s, blocks, f0, nf, ch, hb, o);
      continue;
    }
    switch (ch) {
      case 1: hipLaunchKernelGGL((k_overlap_add<1, true>), grid, dim3(256), 0, s, blocks, f0, nf, ch, hb, o); break;
      case 2: hipLaunchKernelGGL((k_overlap_add<2, true>), grid, dim3(256), 0, s, blocks, f0, nf, ch, hb, o); break;
      case 4: hipLaunchKernelGGL((k_overlap_add<4, true>), grid, dim3(256), 0, s, blocks, f0, nf, ch, hb, o); break;
      case 8: hipLaunchKernelGGL((k_overlap_add<8, true>), grid, dim3(256), 0, s, blocks, f0, nf, ch, hb, o); break;
      default: hipLaunchKernelGGL((k_overlap_add<0, true>), grid, dim3(256), 0, s, blocks, f0, nf, ch, hb, o); break;
    }
  }
  return hipGetLastError();
}

}  // namespace glc
