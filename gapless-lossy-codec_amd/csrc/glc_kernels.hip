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

// ------------------------------------------------------------------------------------------
// K1: C[m][k] = fl( fl( sum_{i ascending} fl( fl(x[m,i]*w[i]) * T[k][i] ) ) * norm )
//
// Exact-order SGEMM on the vector ALU: M = frame-channels, N = 1024, K = 2048, no split-K, one
// accumulator per output, multiply and add issued as separate instructions.  128x128 tile per
// 256-thread workgroup, 8x8 outputs per lane (as 2x2 groups of 4 so that every LDS read is a
// conflict-free ds_read_b128), A tile built on the fly from interleaved PCM (window applied
// while staging), T streamed from L2 through a double-buffered LDS ring.  blockIdx.x % 8 selects
// the coefficient tile so each XCD's L2 keeps one 1 MiB panel of T.
// ------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 16;

__global__ __launch_bounds__(256, 2) void k_mdct_fwd(DeviceTables tb, PcmView pcm,
                                                      long long frame_begin, unsigned M,
                                                      float *__restrict__ coef) {
  // i-major tiles: As[ii][row], Bs[ii][col]; every ds_read in the inner loop is a b128 whose
  // 16 lanes of a group cover one contiguous 256-B span (conflict-free).
  __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

  const int tid = threadIdx.x;
  const int n_tile = blockIdx.x & 7;  // 1024 / BN = 8 coefficient tiles
  const int m_tile = blockIdx.x >> 3;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int tx = tid & 15, ty = tid >> 4;

  // --- A operand: interleaved PCM read through a buffer descriptor whose hardware range check
  // supplies the encoder's zero padding (512 leading zeros, tail, shard edges): an element
  // before the descriptor base wraps to a huge unsigned offset, one past the end is >= the
  // record count; both load 0.0.  All descriptor inputs are blockIdx/kernarg scalars.
  const long long ch = pcm.ch;
  const long long f0 = frame_begin + m0 / pcm.ch;                        // first frame of the tile
  const long long e_first = (f0 * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch;
  long long e_end = static_cast<long long>(pcm.t_count) * ch;            // shard end (elements)
  const long long n_rel = static_cast<long long>(pcm.n_samples) - static_cast<long long>(pcm.t0) * ch;
  if (e_end > n_rel) e_end = n_rel;                                      // stream end
  long long e_base = e_first < 0 ? 0 : e_first;
  long long e_cnt = e_end - e_base;
  if (e_cnt < 0) e_cnt = 0;
  if (e_cnt > (1ll << 28)) e_cnt = 1ll << 28;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(pcm.p + e_base), 0, static_cast<int>(e_cnt * 4), 0x00020000);

  // this thread stages ONE row (r = tid % 128) and 8 of the 16 i of a stage (i = a_i + 2 j)
  const int a_r = tid & (BM - 1);
  const int a_i = tid >> 7;
  const unsigned a_row = m0 + a_r;
  unsigned a_off = 0x80000000u;  // out-of-range row: every load returns 0
  if (a_row < M) {
    const long long f = frame_begin + a_row / pcm.ch;
    const long long c = a_row % pcm.ch;
    const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch + c;
    a_off = static_cast<unsigned>((e_row - e_base + a_i * ch) * 4);  // may wrap: that IS the padding
  }
  const unsigned a_step = static_cast<unsigned>(2 * ch * 4);  // bytes between this thread's i's
  const float *w_ptr = tb.window + a_i;
  // B operand: float4 (row = idx/32, col4 = idx%32), idx = tid + 256 j, j = 0..1
  const float *b_ptr = tb.cos_t + n0 + static_cast<size_t>(tid >> 5) * kHopI + (tid & 31) * 4;

  float a_stage[8];
  float4 b_stage[2];

  auto load_stage = [&](int i0) {
    const unsigned off0 = a_off + static_cast<unsigned>(i0) * static_cast<unsigned>(ch * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off0 + j * a_step, 0, 0));
      a_stage[j] = mul_rn(x, w_ptr[i0 + 2 * j]);  // block[i] = slice[i] * window[i], src/codec.rs:480
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      b_stage[j] = *reinterpret_cast<const float4 *>(b_ptr + static_cast<size_t>(i0 + 8 * j) * kHopI);
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) As[buf][(a_i + 2 * j) * BM + a_r] = a_stage[j];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *reinterpret_cast<float4 *>(&Bs[buf][((tid >> 5) + 8 * j) * BN + (tid & 31) * 4]) = b_stage[j];
  };

  float acc[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[r][c] = 0.0f;  // `let mut s = 0.0f32`, :365

  load_stage(0);
  store_stage(0);
  __syncthreads();

  constexpr int kStages = kFrameI / BK;
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    const int buf = s & 1;
    // prefetch the next stage into registers (the last iteration re-fetches stage 0 into the
    // idle buffer: keeps the loop body branch-free)
    load_stage(((s + 1) & (kStages - 1)) * BK);

    const float *Ab = As[buf] + ty * 4;
    const float *Bb = Bs[buf] + tx * 4;
    float4 a0 = *reinterpret_cast<const float4 *>(&Ab[0]);
    float4 a1 = *reinterpret_cast<const float4 *>(&Ab[64]);
    float4 b0 = *reinterpret_cast<const float4 *>(&Bb[0]);
    float4 b1 = *reinterpret_cast<const float4 *>(&Bb[64]);
#pragma unroll 2
    for (int ii = 0; ii < BK; ++ii) {
      // next i-step's operands in flight while this one computes (the last step re-reads row 0)
      const int nx = (ii + 1) & (BK - 1);
      const float4 na0 = *reinterpret_cast<const float4 *>(&Ab[nx * BM]);
      const float4 na1 = *reinterpret_cast<const float4 *>(&Ab[nx * BM + 64]);
      const float4 nb0 = *reinterpret_cast<const float4 *>(&Bb[nx * BN]);
      const float4 nb1 = *reinterpret_cast<const float4 *>(&Bb[nx * BN + 64]);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[r][c] = add_rn(acc[r][c], mul_rn(av[r], bv[c]));  // :369
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }

    store_stage(buf ^ 1);
    __syncthreads();
  }

  // epilogue: out[k] = s * norm, :372
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const unsigned row = m0 + ((r < 4) ? (ty * 4 + r) : (64 + ty * 4 + (r - 4)));
    if (row >= M) continue;
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0;
    float4 o0, o1;
    o0.x = mul_rn(acc[r][0], tb.norm); o0.y = mul_rn(acc[r][1], tb.norm);
    o0.z = mul_rn(acc[r][2], tb.norm); o0.w = mul_rn(acc[r][3], tb.norm);
    o1.x = mul_rn(acc[r][4], tb.norm); o1.y = mul_rn(acc[r][5], tb.norm);
    o1.z = mul_rn(acc[r][6], tb.norm); o1.w = mul_rn(acc[r][7], tb.norm);
    *reinterpret_cast<float4 *>(dst + tx * 4) = o0;
    *reinterpret_cast<float4 *>(dst + 64 + tx * 4) = o1;
  }
}

// ------------------------------------------------------------------------------------------
// K2: one wavefront per frame-channel row.  scale = max|c| (order-free), per-band sequential
// sum of squares (one lane per critical band: the reference's summation order is kept), masking
// thresholds, noise floor, quantiser.  Emits the dense i16 row + {scale, nnz} into the record.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_quantize(DeviceTables tb, const float *__restrict__ coef,
                                                   unsigned M, unsigned ch, unsigned long long rec_bytes,
                                                   unsigned long long hdr_bytes,
                                                   unsigned char *__restrict__ records) {
  __shared__ __attribute__((aligned(16))) float srow[4][kHopI];
  __shared__ float sbase[4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned m = blockIdx.x * 4 + w;
  const bool live = m < M;

  float4 c4[4];
  float amax = 0.0f;
  if (live) {
    const float4 *src = reinterpret_cast<const float4 *>(coef + static_cast<size_t>(m) * kHopI);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c4[j] = src[lane + 64 * j];
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(c4[j].x), fabsf(c4[j].y)), fmaxf(fabsf(c4[j].z), fabsf(c4[j].w))));
      *reinterpret_cast<float4 *>(&srow[w][(lane + 64 * j) * 4]) = c4[j];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
  const float scale = fmaxf(amax, 1e-10f);  // :488 (and global_max at :198, :278)
  __syncthreads();

  if (live && lane < static_cast<int>(tb.n_bands)) {
    const unsigned lo = tb.edges[lane], hi = tb.edges[lane + 1];
    float ss = 0.0f;
    for (unsigned i = lo; i < hi; ++i) {
      const float v = srow[w][i];
      ss = add_rn(ss, mul_rn(v, v));  // :212-214, ascending i
    }
    const float energy = sqrtf(ss / tb.band_len[lane]);                       // :214-215
    const float base = mul_rn(mul_rn(mul_rn(energy, 0.01f), tb.cf), tb.band_pf[lane]);  // :223
    sbase[w][lane] = base;
  }
  __syncthreads();
  if (!live) return;

  const unsigned frame = m / ch, c = m % ch;
  unsigned char *rec = records + static_cast<size_t>(frame) * rec_bytes;
  short *qrow = reinterpret_cast<short *>(rec + hdr_bytes) + static_cast<size_t>(c) * kFrameI;

  const float nfl = mul_rn(tb.noise_floor, scale);  // :277
  const float peak_gate = mul_rn(scale, 0.3f);      // global_max * 0.3, :232
  const float peak_cap = mul_rn(scale, 0.05f);      // global_max * 0.05, :234
  unsigned cnt = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k0 = (lane + 64 * j) * 4;
    const float cv[4] = {c4[j].x, c4[j].y, c4[j].z, c4[j].w};
    short qv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      const float a = fabsf(cv[e]);
      float t = mul_rn(sbase[w][tb.band_of[k]], tb.indiv[k]);  // :228-229
      if (a > peak_gate) t = fminf(t, peak_cap);               // :232-235
      const float thr = mul_rn(t, scale);                      // :288
      short q = 0;
      if (a > nfl && a > thr) {                                // :291
        const float normalized = cv[e] / scale;                // :299 (IEEE divide)
        q = sat_i16(roundf(mul_rn(normalized, 32768.0f)));     // :300-301
      }
      qv[e] = q;
      cnt += (q != 0);
    }
    short4 pk;
    pk.x = qv[0]; pk.y = qv[1]; pk.z = qv[2]; pk.w = qv[3];
    *reinterpret_cast<short4 *>(qrow + k0) = pk;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (lane == 0) {
    *reinterpret_cast<float *>(rec + 8 + 8 * c) = scale;
    *reinterpret_cast<unsigned *>(rec + 8 + 8 * c + 4) = cnt;
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

  const unsigned long long p0 = rows.row_off[m], p1 = rows.row_off[m + 1];
  const unsigned n = static_cast<unsigned>(p1 - p0);
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
// D2: overlap-add + interleave.  blocks holds frames [blk_frame0, ...) as [frame][ch][2048];
// hop h = second half of frame h-1 (+0.0 before the first frame) + first half of frame h; the
// hop after the last frame is the bare overlap tail (no add, src/codec.rs:722-729).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_overlap_add(const float *__restrict__ blocks, long long blk_frame0,
                                                      unsigned long long n_frames, unsigned ch,
                                                      unsigned long long hop_begin,
                                                      unsigned long long n_out, float *__restrict__ out) {
  const unsigned long long per_hop = static_cast<unsigned long long>(kHopI) * ch;
  for (unsigned long long o = blockIdx.x * 256ull + threadIdx.x; o < n_out;
       o += static_cast<unsigned long long>(gridDim.x) * 256ull) {
    const unsigned long long h = hop_begin + o / per_hop;
    const unsigned rem = static_cast<unsigned>(o % per_hop);
    const unsigned i = rem / ch, c = rem % ch;
    float prev = 0.0f;  // overlap starts as +0.0, :601
    if (h >= 1) {
      const long long slot = static_cast<long long>(h) - 1 - blk_frame0;
      prev = blocks[(static_cast<size_t>(slot) * ch + c) * kFrameI + kHopI + i];
    }
    float v;
    if (h < n_frames) {
      const long long slot = static_cast<long long>(h) - blk_frame0;
      const float cur = blocks[(static_cast<size_t>(slot) * ch + c) * kFrameI + i];
      v = add_rn(prev, cur);  // :695
    } else {
      v = prev;  // :727
    }
    out[o] = v;
  }
}

}  // namespace

// ---------------------------------------------------------------------------- launchers

hipError_t launch_mdct_forward(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin,
                               uint32_t M, float *coef, hipStream_t s) {
  if (M == 0) return hipSuccess;
  const unsigned m_tiles = (M + BM - 1) / BM;
  hipLaunchKernelGGL(k_mdct_fwd, dim3(m_tiles * 8), dim3(256), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), M, coef);
  return hipGetLastError();
}

hipError_t launch_quantize(const DeviceTables &t, const float *coef, uint32_t M, uint32_t ch,
                           uint8_t *records, hipStream_t s) {
  if (M == 0) return hipSuccess;
  const unsigned long long hdr = ((8ull + 8ull * ch) + 15ull) & ~15ull;
  const unsigned long long rec = hdr + 2ull * kFrameI * ch;
  hipLaunchKernelGGL(k_quantize, dim3((M + 3) / 4), dim3(256), 0, s, t, coef, M, ch, rec, hdr, records);
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

hipError_t launch_imdct_rows(const DeviceTables &t, const DecodeRows &rows, uint32_t row_begin,
                             uint32_t M, uint32_t ch, float *blocks, hipStream_t s) {
  if (M == 0) return hipSuccess;
  hipLaunchKernelGGL(k_imdct_rows, dim3(M), dim3(256), 0, s, t, rows, row_begin, M, ch, blocks);
  return hipGetLastError();
}

hipError_t launch_overlap_add(const float *blocks, int64_t blk_frame0, uint64_t n_frames, uint32_t ch,
                              uint64_t hop_begin, uint64_t hop_end, float *out, hipStream_t s) {
  if (hop_end <= hop_begin) return hipSuccess;
  const unsigned long long n_out = (hop_end - hop_begin) * 1024ull * ch;
  unsigned long long blocks_needed = (n_out + 255) / 256;
  const unsigned grid = static_cast<unsigned>(blocks_needed < 8192 ? blocks_needed : 8192);
  hipLaunchKernelGGL(k_overlap_add, dim3(grid), dim3(256), 0, s, blocks,
                     static_cast<long long>(blk_frame0), static_cast<unsigned long long>(n_frames), ch,
                     static_cast<unsigned long long>(hop_begin), n_out, out);
  return hipGetLastError();
}

}  // namespace glc
