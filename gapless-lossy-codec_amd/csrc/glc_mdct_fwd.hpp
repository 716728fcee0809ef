// glc_mdct_fwd.hpp — K1, the windowed forward MDCT as an exact-order SGEMM on the vector ALU.
//
//   C[m][k] = fl( fl( sum_{i ascending} fl( fl(x[m,i]*w[i]) * T[k][i] ) ) * norm )
//
// replaces the window loop (src/codec.rs:476-481) and MdctTables::mdct_block (:359-374) of the
// reference for every frame-channel row m = (frame - frame_begin)*ch + c.  M = rows, N = 1024,
// K = 2048; the K loop is strictly ascending with ONE accumulator per output, multiply and add
// are separate instructions (v_pk_mul_f32 / v_pk_add_f32, never an FMA), there is no split-K.
//
// What is in this file - exactly the kernels libglc_hip.so instantiates (launch_mdct_forward in
// glc_kernels.hip); every other shape, schedule and ablation that was measured lives in
// tools/k1_variants.hpp beside the tuning harness tools/k1_tune.hip:
//   k_mdct_fwd_st      launches of >= 4096 rows (BASELINE config 2 = 8192): a wave's lanes hold 256 ROWS
//                      (4 each) and share 8 columns, so the table values are wave-uniform and come by scalar
//                      loads straight from the table into SGPRs; the windowed samples by one ds_read_b128
//                      per i-step from a 3-slot LDS ring; 4 i-steps per fetch, a group ahead.  256x128
//                      tile / 1024 threads (one workgroup per CU, priority by distance from the barrier) or
//                      256x64 / 512 (two per CU, priority by quarter of the i loop), chosen per launch by
//                      the fill of its last round of tiles; PCM tile by dwordx4 segments when the stream has
//                      1 / 2 / 4 / 8 channels (CH), one dword per (row, sample) otherwise (CH = 0).
//   k_mdct_fwd_dma     the kernel of rounds 2 / 3 for the same launches, now behind
//                      glc_debug_set_mdct_variant(ctx, 1): 128x128 tile, 512 threads, 4x8 outputs per lane
//                      with lanes <-> columns, both operands from LDS (table tile by LDS-DMA two stages
//                      ahead); 8-9 % slower on real samples because the chip holds less clock under it.
//   k_mdct_fwd_sched   the opening rounds of glc_encode (2048 rows each, running beside each other), as <64,128,16,4>:
//                      64x128 tile, 256 threads, ONE workgroup per CU; hand-scheduled inline-asm i-steps (step4), LDS
//                      operand prefetch, XCD-aware tile map, register staging, classic double buffer.  (Rounds 1-3:
//                      every launch of 1793..4095 rows.)
//   k_mdct_fwd_small   below 3584 rows (4096 for channel counts without a segment loader): 2x2 / 2x4 outputs per
//                      lane (<= 640 rows / above) on 32x32 / 32x64 tiles, 256 threads,
//                      hand-scheduled with a register ring of LDS operands and counted lgkmcnt waits
//                      (short clips: one wave's 2048-step chain is everything; the step is the lane tile).
//   k_mdct_fwd         the same tiling left to hipcc's scheduler: no longer instantiated by the library
//                      (round 2 used <32,64,32,4,4,4,2> for <= 512 rows); kept as the plainest statement
//                      of the loop and for tools/k1_tune.hip.
//   mac2rows           the 2-row x 8-column multiply/add block; also the entry step of D1 (k_imdct_chan).
#pragma once
#include <hip/hip_runtime.h>

#include "glc_kernels.h"

namespace glc {
namespace k1 {

constexpr int kHopI = 1024;
constexpr int kFrameI = 2048;

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }

// BM x BN output tile per workgroup, BK table rows per LDS stage, TM x TN outputs per lane
// (TM, TN in {4, 8}: one or two conflict-free ds_read_b128 per operand and i-step),
// UNROLL i-steps per loop body, MINW = min waves per SIMD for the register allocator.
template <int BM, int BN, int BK, int TM, int TN, int UNROLL, int MINW>
struct Cfg {
  static constexpr int kThreads = (BM / TM) * (BN / TN);
  static constexpr int kNtx = BN / TN;
  static constexpr int kNTiles = kHopI / BN;
  static constexpr int kAPer = BM * BK / kThreads;       // A elements staged per thread
  static constexpr int kAStride = kThreads / BM;          // i distance between them
  static constexpr int kBPer = BK * BN / 4 / kThreads;    // B float4 staged per thread
  static constexpr int kBRowsPer = kThreads / (BN / 4);   // table rows covered per float4 round
  static_assert(kThreads % BM == 0 && kAPer * kAStride == BK, "A staging shape");
  static_assert(kThreads % (BN / 4) == 0 && kBPer * kBRowsPer == BK, "B staging shape");
  static_assert((TM == 4 || TM == 8) && (TN == 4 || TN == 8), "lane tile");
  static_assert((BK & (BK - 1)) == 0 && BK % UNROLL == 0, "BK");
};

template <int BM, int BN, int BK, int TM, int TN, int UNROLL, int MINW>
__global__ __launch_bounds__((BM / TM) * (BN / TN), MINW) void k_mdct_fwd(DeviceTables tb, PcmView pcm,
                                                                         long long frame_begin,
                                                                         unsigned M,
                                                                         float *__restrict__ coef) {
  using C = Cfg<BM, BN, BK, TM, TN, UNROLL, MINW>;
  // i-major tiles: As[ii][row], Bs[ii][col]; every ds_read in the inner loop is a b128 whose
  // 16 lanes of a group cover one contiguous 256-B span (conflict-free).
  __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

  const int tid = threadIdx.x;
  // blockIdx.x % kNTiles picks the coefficient tile: blocks are dealt round-robin over the 8
  // XCDs, so each XCD's L2 keeps only 1024/BN/8 panels of T (speed only, never correctness).
  const int n_tile = blockIdx.x % C::kNTiles;
  const int m_tile = blockIdx.x / C::kNTiles;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int tx = tid % C::kNtx, ty = tid / C::kNtx;

  // --- A operand: interleaved PCM read through a buffer descriptor whose hardware range check
  // supplies the encoder's zero padding (512 leading zeros, tail, shard edges): an element
  // before the descriptor base wraps to a huge unsigned offset, one past the end is >= the
  // record count; both load 0.0.  All descriptor inputs are blockIdx/kernarg scalars.
  const long long ch = pcm.ch;
  const long long f0 = frame_begin + m0 / pcm.ch;  // first frame of the tile
  const long long e_first = (f0 * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch;
  long long e_end = static_cast<long long>(pcm.t_count) * ch;  // shard end (elements)
  const long long n_rel = static_cast<long long>(pcm.n_samples) - static_cast<long long>(pcm.t0) * ch;
  if (e_end > n_rel) e_end = n_rel;  // stream end
  const long long e_base = e_first < 0 ? 0 : e_first;
  long long e_cnt = e_end - e_base;
  if (e_cnt < 0) e_cnt = 0;
  if (e_cnt > (1ll << 28)) e_cnt = 1ll << 28;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(pcm.p + e_base), 0, static_cast<int>(e_cnt * 4), 0x00020000);

  // this thread stages ONE row (r = tid % BM) and kAPer of the BK i of a stage
  const int a_r = tid % BM;
  const int a_i = tid / BM;
  const unsigned a_row = m0 + a_r;
  unsigned a_off = 0x80000000u;  // out-of-range row: every load returns 0
  if (a_row < M) {
    const long long f = frame_begin + a_row / pcm.ch;
    const long long c = a_row % pcm.ch;
    const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch + c;
    a_off = static_cast<unsigned>((e_row - e_base + a_i * ch) * 4);  // may wrap: that IS the padding
  }
  const unsigned a_step = static_cast<unsigned>(C::kAStride * ch * 4);  // bytes between this thread's i's
  const float *w_ptr = tb.window + a_i;
  // B operand: float4 (row = idx / (BN/4), col4 = idx % (BN/4)), idx = tid + kThreads j
  const int b_r = tid / (BN / 4), b_c4 = tid % (BN / 4);
  const float *b_ptr = tb.cos_t + n0 + static_cast<size_t>(b_r) * kHopI + b_c4 * 4;

  float a_stage[C::kAPer];
  float4 b_stage[C::kBPer];

  auto load_stage = [&](int i0) {
    const unsigned off0 = a_off + static_cast<unsigned>(i0) * static_cast<unsigned>(ch * 4);
#pragma unroll
    for (int j = 0; j < C::kAPer; ++j) {
      const float x = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off0 + j * a_step, 0, 0));
      a_stage[j] = mul_rn(x, w_ptr[i0 + C::kAStride * j]);  // block[i] = slice[i]*window[i], :480
    }
#pragma unroll
    for (int j = 0; j < C::kBPer; ++j)
      b_stage[j] = *reinterpret_cast<const float4 *>(b_ptr + static_cast<size_t>(i0 + C::kBRowsPer * j) * kHopI);
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int j = 0; j < C::kAPer; ++j) As[buf][(a_i + C::kAStride * j) * BM + a_r] = a_stage[j];
#pragma unroll
    for (int j = 0; j < C::kBPer; ++j)
      *reinterpret_cast<float4 *>(&Bs[buf][(b_r + C::kBRowsPer * j) * BN + b_c4 * 4]) = b_stage[j];
  };

  float acc[TM][TN];
#pragma unroll
  for (int r = 0; r < TM; ++r)
#pragma unroll
    for (int c = 0; c < TN; ++c) acc[r][c] = 0.0f;  // `let mut s = 0.0f32`, :365

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
#pragma unroll UNROLL
    for (int ii = 0; ii < BK; ++ii) {
      float av[TM], bv[TN];
      {
        const float4 v = *reinterpret_cast<const float4 *>(&Ab[ii * BM]);
        av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
      }
      if constexpr (TM == 8) {
        const float4 v = *reinterpret_cast<const float4 *>(&Ab[ii * BM + BM / 2]);
        av[4] = v.x; av[5] = v.y; av[6] = v.z; av[7] = v.w;
      }
      {
        const float4 v = *reinterpret_cast<const float4 *>(&Bb[ii * BN]);
        bv[0] = v.x; bv[1] = v.y; bv[2] = v.z; bv[3] = v.w;
      }
      if constexpr (TN == 8) {
        const float4 v = *reinterpret_cast<const float4 *>(&Bb[ii * BN + BN / 2]);
        bv[4] = v.x; bv[5] = v.y; bv[6] = v.z; bv[7] = v.w;
      }
#pragma unroll
      for (int r = 0; r < TM; ++r)
#pragma unroll
        for (int c = 0; c < TN; ++c) acc[r][c] = add_rn(acc[r][c], mul_rn(av[r], bv[c]));  // :369
    }

    store_stage(buf ^ 1);
    __syncthreads();
  }

  // epilogue: out[k] = s * norm, :372
#pragma unroll
  for (int r = 0; r < TM; ++r) {
    const unsigned row = m0 + ((r < 4) ? (ty * 4 + r) : (BM / 2 + ty * 4 + (r - 4)));
    if (row >= M) continue;
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0;
    float4 o;
    o.x = mul_rn(acc[r][0], tb.norm); o.y = mul_rn(acc[r][1], tb.norm);
    o.z = mul_rn(acc[r][2], tb.norm); o.w = mul_rn(acc[r][3], tb.norm);
    *reinterpret_cast<float4 *>(dst + tx * 4) = o;
    if constexpr (TN == 8) {
      o.x = mul_rn(acc[r][4], tb.norm); o.y = mul_rn(acc[r][5], tb.norm);
      o.z = mul_rn(acc[r][6], tb.norm); o.w = mul_rn(acc[r][7], tb.norm);
      *reinterpret_cast<float4 *>(dst + BN / 2 + tx * 4) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Hand-scheduled kernels (4x8 outputs per lane).  Same arithmetic, but the instruction order
// is pinned with inline asm because hipcc (ROCm 7.2) (a) places every v_pk_add directly behind
// the v_pk_mul it depends on and (b) sinks the LDS reads of the next i-step below the current
// step's math, so a wave stalls on both.  Here each i-step issues its 6 ds_read_b64 for the NEXT
// step first, then 2 groups of {8 v_pk_mul, 8 v_pk_add} (dependent instructions 8 apart), then
// one s_waitcnt: register set X feeds even steps, set Y odd steps (ping-pong, no copies).
// ------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Operands {  // one i-step's A (4 rows) and B (8 columns) values of this lane
  f32x2 a0, a1, b0, b1, b2, b3;
};

template <int BM, int BN>
__device__ __forceinline__ void lds_fetch4(Operands &o, unsigned a_addr, unsigned b_addr, int ii) {
  // 4-row lane tile: rows ty*4 .. ty*4+3
  asm volatile(
      "ds_read_b64 %0, %6 offset:%c8\n\t"
      "ds_read_b64 %1, %6 offset:%c9\n\t"
      "ds_read_b64 %2, %7 offset:%c10\n\t"
      "ds_read_b64 %3, %7 offset:%c11\n\t"
      "ds_read_b64 %4, %7 offset:%c12\n\t"
      "ds_read_b64 %5, %7 offset:%c13"
      : "=&v"(o.a0), "=&v"(o.a1), "=&v"(o.b0), "=&v"(o.b1), "=&v"(o.b2), "=&v"(o.b3)
      : "v"(a_addr), "v"(b_addr), "i"(ii * BM * 4), "i"(ii * BM * 4 + 8), "i"(ii * BN * 4), "i"(ii * BN * 4 + 8),
        "i"(ii * BN * 4 + BN * 2), "i"(ii * BN * 4 + BN * 2 + 8)
      : "memory");
}

__device__ __forceinline__ void lds_wait4(Operands &o) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(o.a0), "+v"(o.a1), "+v"(o.b0), "+v"(o.b1), "+v"(o.b2), "+v"(o.b3)
               :
               : "memory");
}

// rows (r, r+1) x 8 columns: c[r][j] += a.lo * b_j ; c[r+1][j] += a.hi * b_j   (j = column pair)
__device__ __forceinline__ void mac2rows(f32x2 (&c0)[4], f32x2 (&c1)[4], f32x2 a, f32x2 b0, f32x2 b1,
                                         f32x2 b2, f32x2 b3) {
  f32x2 t0, t1, t2, t3, t4, t5, t6, t7;
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
      : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}

// One whole i-step of a 4x8 lane tile as ONE asm statement (no compiler pads between groups):
// issue the next step's six ds_read_b64 (into n), do this step's 32 multiplies and 32 adds from
// c (dependent instructions 8 apart), then wait for the reads.  r<row><colpair> = accumulators.
template <int BM, int BN, bool FETCH>
__device__ __forceinline__ void step4(f32x2 (&acc)[4][4], const Operands &c, Operands &n, unsigned a_addr,
                                      unsigned b_addr, int ii_next) {
  f32x2 t0, t1, t2, t3, t4, t5, t6, t7;
  // named operands keep the string readable: [r<row><colpair>]
  if constexpr (FETCH) {
    asm volatile(
        "ds_read_b64 %[na0], %[aa] offset:%c[oa0]\n\t"
        "ds_read_b64 %[na1], %[aa] offset:%c[oa1]\n\t"
        "ds_read_b64 %[nb0], %[ba] offset:%c[ob0]\n\t"
        "ds_read_b64 %[nb1], %[ba] offset:%c[ob1]\n\t"
        "ds_read_b64 %[nb2], %[ba] offset:%c[ob2]\n\t"
        "ds_read_b64 %[nb3], %[ba] offset:%c[ob3]\n\t"
        "v_pk_mul_f32 %[t0], %[ca0], %[cb0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t1], %[ca0], %[cb1] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t2], %[ca0], %[cb2] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t3], %[ca0], %[cb3] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t4], %[ca0], %[cb0] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t5], %[ca0], %[cb1] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t6], %[ca0], %[cb2] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t7], %[ca0], %[cb3] op_sel:[1,0]\n\t"
        "v_pk_add_f32 %[r00], %[r00], %[t0]\n\t"
        "v_pk_add_f32 %[r01], %[r01], %[t1]\n\t"
        "v_pk_add_f32 %[r02], %[r02], %[t2]\n\t"
        "v_pk_add_f32 %[r03], %[r03], %[t3]\n\t"
        "v_pk_add_f32 %[r10], %[r10], %[t4]\n\t"
        "v_pk_add_f32 %[r11], %[r11], %[t5]\n\t"
        "v_pk_add_f32 %[r12], %[r12], %[t6]\n\t"
        "v_pk_add_f32 %[r13], %[r13], %[t7]\n\t"
        "v_pk_mul_f32 %[t0], %[ca1], %[cb0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t1], %[ca1], %[cb1] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t2], %[ca1], %[cb2] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t3], %[ca1], %[cb3] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t4], %[ca1], %[cb0] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t5], %[ca1], %[cb1] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t6], %[ca1], %[cb2] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t7], %[ca1], %[cb3] op_sel:[1,0]\n\t"
        "v_pk_add_f32 %[r20], %[r20], %[t0]\n\t"
        "v_pk_add_f32 %[r21], %[r21], %[t1]\n\t"
        "v_pk_add_f32 %[r22], %[r22], %[t2]\n\t"
        "v_pk_add_f32 %[r23], %[r23], %[t3]\n\t"
        "v_pk_add_f32 %[r30], %[r30], %[t4]\n\t"
        "v_pk_add_f32 %[r31], %[r31], %[t5]\n\t"
        "v_pk_add_f32 %[r32], %[r32], %[t6]\n\t"
        "v_pk_add_f32 %[r33], %[r33], %[t7]\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [r00] "+v"(acc[0][0]), [r01] "+v"(acc[0][1]), [r02] "+v"(acc[0][2]), [r03] "+v"(acc[0][3]),
          [r10] "+v"(acc[1][0]), [r11] "+v"(acc[1][1]), [r12] "+v"(acc[1][2]), [r13] "+v"(acc[1][3]),
          [r20] "+v"(acc[2][0]), [r21] "+v"(acc[2][1]), [r22] "+v"(acc[2][2]), [r23] "+v"(acc[2][3]),
          [r30] "+v"(acc[3][0]), [r31] "+v"(acc[3][1]), [r32] "+v"(acc[3][2]), [r33] "+v"(acc[3][3]),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
          [t6] "=&v"(t6), [t7] "=&v"(t7), [na0] "=&v"(n.a0), [na1] "=&v"(n.a1), [nb0] "=&v"(n.b0),
          [nb1] "=&v"(n.b1), [nb2] "=&v"(n.b2), [nb3] "=&v"(n.b3)
        : [ca0] "v"(c.a0), [ca1] "v"(c.a1), [cb0] "v"(c.b0), [cb1] "v"(c.b1), [cb2] "v"(c.b2), [cb3] "v"(c.b3),
          [aa] "v"(a_addr), [ba] "v"(b_addr), [oa0] "i"(ii_next * BM * 4), [oa1] "i"(ii_next * BM * 4 + 8),
          [ob0] "i"(ii_next * BN * 4), [ob1] "i"(ii_next * BN * 4 + 8), [ob2] "i"(ii_next * BN * 4 + BN * 2),
          [ob3] "i"(ii_next * BN * 4 + BN * 2 + 8)
        : "memory");
  } else {
    asm volatile(
        "v_pk_mul_f32 %[t0], %[ca0], %[cb0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t1], %[ca0], %[cb1] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t2], %[ca0], %[cb2] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t3], %[ca0], %[cb3] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t4], %[ca0], %[cb0] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t5], %[ca0], %[cb1] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t6], %[ca0], %[cb2] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t7], %[ca0], %[cb3] op_sel:[1,0]\n\t"
        "v_pk_add_f32 %[r00], %[r00], %[t0]\n\t"
        "v_pk_add_f32 %[r01], %[r01], %[t1]\n\t"
        "v_pk_add_f32 %[r02], %[r02], %[t2]\n\t"
        "v_pk_add_f32 %[r03], %[r03], %[t3]\n\t"
        "v_pk_add_f32 %[r10], %[r10], %[t4]\n\t"
        "v_pk_add_f32 %[r11], %[r11], %[t5]\n\t"
        "v_pk_add_f32 %[r12], %[r12], %[t6]\n\t"
        "v_pk_add_f32 %[r13], %[r13], %[t7]\n\t"
        "v_pk_mul_f32 %[t0], %[ca1], %[cb0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t1], %[ca1], %[cb1] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t2], %[ca1], %[cb2] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t3], %[ca1], %[cb3] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[t4], %[ca1], %[cb0] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t5], %[ca1], %[cb1] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t6], %[ca1], %[cb2] op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %[t7], %[ca1], %[cb3] op_sel:[1,0]\n\t"
        "v_pk_add_f32 %[r20], %[r20], %[t0]\n\t"
        "v_pk_add_f32 %[r21], %[r21], %[t1]\n\t"
        "v_pk_add_f32 %[r22], %[r22], %[t2]\n\t"
        "v_pk_add_f32 %[r23], %[r23], %[t3]\n\t"
        "v_pk_add_f32 %[r30], %[r30], %[t4]\n\t"
        "v_pk_add_f32 %[r31], %[r31], %[t5]\n\t"
        "v_pk_add_f32 %[r32], %[r32], %[t6]\n\t"
        "v_pk_add_f32 %[r33], %[r33], %[t7]"
        : [r00] "+v"(acc[0][0]), [r01] "+v"(acc[0][1]), [r02] "+v"(acc[0][2]), [r03] "+v"(acc[0][3]),
          [r10] "+v"(acc[1][0]), [r11] "+v"(acc[1][1]), [r12] "+v"(acc[1][2]), [r13] "+v"(acc[1][3]),
          [r20] "+v"(acc[2][0]), [r21] "+v"(acc[2][1]), [r22] "+v"(acc[2][2]), [r23] "+v"(acc[2][3]),
          [r30] "+v"(acc[3][0]), [r31] "+v"(acc[3][1]), [r32] "+v"(acc[3][2]), [r33] "+v"(acc[3][3]),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
          [t6] "=&v"(t6), [t7] "=&v"(t7)
        : [ca0] "v"(c.a0), [ca1] "v"(c.a1), [cb0] "v"(c.b0), [cb1] "v"(c.b1), [cb2] "v"(c.b2), [cb3] "v"(c.b3));
  }
}

// 64x128 (BM x BN) tile, 4x8 outputs per lane, register staging, two LDS slots per operand tile.
template <int BM, int BN, int BK, int MINW>
__global__ __launch_bounds__((BM / 4) * (BN / 8)) __attribute__((amdgpu_waves_per_eu(MINW, MINW)))
void k_mdct_fwd_sched(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M,
                      float *__restrict__ coef) {
  constexpr int TM = 4;
  using C = Cfg<BM, BN, BK, TM, 8, 2, MINW>;
  __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

  const int tid = threadIdx.x;
  // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 says which blocks share an
  // L2).  Give XCD x the contiguous tile range [x*T/8, (x+1)*T/8) in (m_tile, n_tile) order: the
  // PCM rows of an m-tile are then fetched by ONE XCD instead of all eight, and the table rows of
  // a stage are shared in that XCD's L2 by all resident m-tiles, which sweep i together
  // (measured: 4x less L2 fill traffic).  Placement affects speed only, never results.
  static_assert(C::kNTiles == 8, "tile map assumes 8 coefficient tiles");
  const unsigned g = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int n_tile = g % C::kNTiles;
  const int m_tile = g / C::kNTiles;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int tx = tid % C::kNtx, ty = tid / C::kNtx;

  const long long ch = pcm.ch;
  const long long f0 = frame_begin + m0 / pcm.ch;
  const long long e_first = (f0 * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch;
  long long e_end = static_cast<long long>(pcm.t_count) * ch;
  const long long n_rel = static_cast<long long>(pcm.n_samples) - static_cast<long long>(pcm.t0) * ch;
  if (e_end > n_rel) e_end = n_rel;
  const long long e_base = e_first < 0 ? 0 : e_first;
  long long e_cnt = e_end - e_base;
  if (e_cnt < 0) e_cnt = 0;
  if (e_cnt > (1ll << 28)) e_cnt = 1ll << 28;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(pcm.p + e_base), 0, static_cast<int>(e_cnt * 4), 0x00020000);

  const int a_r = tid % BM;
  const int a_i = tid / BM;
  const unsigned a_row = m0 + a_r;
  unsigned a_off = 0x80000000u;
  if (a_row < M) {
    const long long f = frame_begin + a_row / pcm.ch;
    const long long c = a_row % pcm.ch;
    const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch + c;
    a_off = static_cast<unsigned>((e_row - e_base + a_i * ch) * 4);
  }
  const unsigned a_step = static_cast<unsigned>(C::kAStride * ch * 4);
  const float *w_ptr = tb.window + a_i;
  const int b_r = tid / (BN / 4), b_c4 = tid % (BN / 4);
  const float *b_ptr = tb.cos_t + n0 + static_cast<size_t>(b_r) * kHopI + b_c4 * 4;

  // staging registers: raw sample and window value are multiplied only when the stage is
  // written to LDS, so the wave waits for its global loads at the END of the stage
  float a_raw[C::kAPer], a_win[C::kAPer];
  f32x4 b_stage[C::kBPer];
  auto load_stage = [&](int i0) {
    const unsigned off0 = a_off + static_cast<unsigned>(i0) * static_cast<unsigned>(ch * 4);
#pragma unroll
    for (int j = 0; j < C::kAPer; ++j) {
      a_raw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off0 + j * a_step, 0, 0));
      a_win[j] = w_ptr[i0 + C::kAStride * j];
    }
#pragma unroll
    for (int j = 0; j < C::kBPer; ++j)
      b_stage[j] = *reinterpret_cast<const f32x4 *>(b_ptr + static_cast<size_t>(i0 + C::kBRowsPer * j) * kHopI);
  };
  auto store_stage = [&](int buf) {
    // pin the first use of the staged registers behind the stage's math (volatile asm
    // statements keep their order): hipcc otherwise hoists the multiply, and with it the
    // vmcnt wait, into the middle of the stage
#pragma unroll
    for (int j = 0; j < C::kAPer; ++j) {
      float r = a_raw[j], w = a_win[j];
      asm volatile("" : "+v"(r), "+v"(w));
      As[buf][(a_i + C::kAStride * j) * BM + a_r] = mul_rn(r, w);  // block[i] = slice[i]*window[i], :480
    }
#pragma unroll
    for (int j = 0; j < C::kBPer; ++j) {
      f32x4 b = b_stage[j];
      asm volatile("" : "+v"(b));
      *reinterpret_cast<f32x4 *>(&Bs[buf][(b_r + C::kBRowsPer * j) * BN + b_c4 * 4]) = b;
    }
  };

  f32x2 acc[TM][4];
#pragma unroll
  for (int r = 0; r < TM; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = f32x2{0.0f, 0.0f};

  load_stage(0);
  store_stage(0);
  __syncthreads();

  // LDS byte addresses of this lane's operand columns (low 32 bits of a generic LDS pointer)
  const unsigned a_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&As[0][ty * 4]));
  const unsigned b_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&Bs[0][tx * 4]));

  constexpr int kStages = kFrameI / BK;
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    const int buf = s & 1;
    load_stage(((s + 1) & (kStages - 1)) * BK);
    const unsigned a_addr = a_lds0 + buf * (BK * BM * 4);
    const unsigned b_addr = b_lds0 + buf * (BK * BN * 4);
    Operands X, Y;
    lds_fetch4<BM, BN>(X, a_addr, b_addr, 0);
    lds_wait4(X);
#pragma unroll
    for (int ii = 0; ii < BK; ii += 2) {
      step4<BM, BN, true>(acc, X, Y, a_addr, b_addr, ii + 1);
      if (ii + 2 < BK) step4<BM, BN, true>(acc, Y, X, a_addr, b_addr, ii + 2);
      else step4<BM, BN, false>(acc, Y, X, a_addr, b_addr, 0);
    }
    store_stage(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int r = 0; r < TM; ++r) {
    const unsigned row = m0 + ty * 4 + r;
    if (row >= M) continue;
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0;
    float4 o;
    o.x = mul_rn(acc[r][0].x, tb.norm); o.y = mul_rn(acc[r][0].y, tb.norm);
    o.z = mul_rn(acc[r][1].x, tb.norm); o.w = mul_rn(acc[r][1].y, tb.norm);
    *reinterpret_cast<float4 *>(dst + tx * 4) = o;
    o.x = mul_rn(acc[r][2].x, tb.norm); o.y = mul_rn(acc[r][2].y, tb.norm);
    o.z = mul_rn(acc[r][3].x, tb.norm); o.w = mul_rn(acc[r][3].y, tb.norm);
    *reinterpret_cast<float4 *>(dst + BN / 2 + tx * 4) = o;
  }
}

template <int BM, int BN, int BK, int MINW>
inline hipError_t launch_sched(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M,
                               float *coef, hipStream_t s) {
  using C = Cfg<BM, BN, BK, 4, 8, 2, MINW>;
  if (M == 0) return hipSuccess;
  const unsigned m_tiles = (M + BM - 1) / BM;
  hipLaunchKernelGGL((k_mdct_fwd_sched<BM, BN, BK, MINW>), dim3(m_tiles * C::kNTiles), dim3(C::kThreads), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), M, coef);
  return hipGetLastError();
}

#ifndef GLC_K1_A_STRIDE
#define GLC_K1_A_STRIDE 132  // floats between consecutive i rows of the A tile in LDS (128 = unpadded; tuning: k1_tune)
#endif
// ------------------------------------------------------------------------------------------
// LDS-DMA kernel (128x128 tile, 512 threads, 4x8 outputs per lane, 3-slot LDS ring).
// The table tile of stage s+2 is copied global -> LDS by `global_load_lds_dwordx4` while stages s
// and s+1 compute: two stages of latency budget, no VGPRs and no ds_write for the table.  The PCM
// tile (needs the window multiply) goes through registers one stage ahead, the window itself sits
// in LDS.  The PCM loads of the loop are inline asm (hipcc must not move their first use); one
// `s_waitcnt vmcnt(0)` per stage, in the middle of the stage, retires loads that are a stage old.
// Same arithmetic, same order.
// ------------------------------------------------------------------------------------------
// PRIO (speed only; which one ships is decided by tools/k1_tune.hip): issue priority as a schedule.  The
// two workgroups of a CU (2 + 2 waves on every SIMD) are arbitrated oldest-first, so left alone the
// older one takes every issue slot it can use and the younger one fills its gaps - and finishes the
// kernel alone, two waves to a SIMD.  A wave lowers its own priority as it gets through the i loop, so
// whichever workgroup is behind goes first:  1 = by quarter of the loop (3, 2, 1, 0);  2 / 3 / 4 = a
// cycle of four levels every 8 / 16 / 32 stages (the skew stays within a cycle).
// STAMP (tuning only): workgroup timeline into `stamps` - start, the four quarter points, end.
template <int MINW, int CH = 0, int PRIO = 0, bool STAMP = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(MINW, MINW)))
void k_mdct_fwd_dma(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M, float *__restrict__ coef,
                    unsigned long long *__restrict__ stamps) {
  // CH = 0: one PCM dword per (row, i) and lane - any channel count.  CH = 1 / 2 / 4 / 8 (the
  // stream's channel count, which then divides the tile height): a stage's 16 samples x CH channels
  // of one frame are 64 * CH contiguous bytes, fetched by 4 * CH lanes with one dwordx4 each - 16x
  // fewer cache-line touches in the texture addresser than the per-row loader.
  // 512 threads, 2 workgroups per CU (56 KiB LDS each).
  constexpr int BM = 128, BN = 128, BK = 16, TM = 4, RING = 3;
  // A rows are kAS floats apart in LDS, 4 more than the tile is wide: the segment loader's four ds_write_b32
  // per lane put lanes of DIFFERENT i at the same column, and with a stride of 128 floats all of them fall
  // into one bank (8-way for stereo, 16-way for 4 / 8 channels: SQ_LDS_BANK_CONFLICT was 22 % of the
  // LDS-array cycles, profiles/r03_k1_pmc_before_prio.txt); the padding spreads them (2-way, which costs nothing).
  constexpr int kAS = GLC_K1_A_STRIDE;
  constexpr int kThreads = 4 * BM;
  constexpr int kAPer = 4, kAStride = 4;
  constexpr bool kSeg = CH != 0;  // segment loader
  static_assert(!kSeg || CH == 1 || CH == 2 || CH == 4 || CH == 8, "segment loader shapes");
  constexpr int kDma = (BK * BN * 4) / (kThreads * 16);  // table-DMA instructions per thread and stage
  static_assert(kDma == 1, "one table-DMA instruction per thread and stage");
  __shared__ __attribute__((aligned(16))) float As[RING][BK * kAS];
  __shared__ __attribute__((aligned(16))) float Bs[RING][BK * BN];
  __shared__ __attribute__((aligned(16))) float Ws[kFrameI];

  const int tid = threadIdx.x;
  const unsigned g = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-aware tile map
  const int n_tile = g % 8;
  const int m_tile = g / 8;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int tx = tid % 16, ty = tid / 16;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  for (int i = tid; i < kFrameI; i += kThreads) Ws[i] = tb.window[i];

  const long long ch = pcm.ch;
  const long long f0 = frame_begin + m0 / pcm.ch;
  const long long e_first = (f0 * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch;
  long long e_end = static_cast<long long>(pcm.t_count) * ch;
  const long long n_rel = static_cast<long long>(pcm.n_samples) - static_cast<long long>(pcm.t0) * ch;
  if (e_end > n_rel) e_end = n_rel;
  const long long e_base = e_first < 0 ? 0 : e_first;
  long long e_cnt = e_end - e_base;
  if (e_cnt < 0) e_cnt = 0;
  if (e_cnt > (1ll << 28)) e_cnt = 1ll << 28;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(pcm.p + e_base), 0, static_cast<int>(e_cnt * 4), 0x00020000);

  const int a_r = tid % BM;
  const int a_i = tid / BM;  // 0..3: i = a_i + 4 j  (the same for every lane of a wave)
  const unsigned a_row = m0 + a_r;
  unsigned a_off = 0x80000000u;
  if (a_row < M) {
    const long long f = frame_begin + a_row / pcm.ch;
    const long long c = a_row % pcm.ch;
    const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch + c;
    a_off = static_cast<unsigned>((e_row - e_base + a_i * ch) * 4);
  }
  const unsigned a_step = static_cast<unsigned>(kAStride * ch * 4);
  const unsigned i_bytes = static_cast<unsigned>(ch * 4);
  // segment loader: lane -> (frame of the tile, 4 consecutive floats of its 16 x CH segment)
  constexpr int kSegCh = kSeg ? CH : 1;
  const int seg_fl = tid / (4 * kSegCh);       // frame within the tile
  const int seg_o = (tid % (4 * kSegCh)) * 4;  // first float of this lane inside the segment
  if constexpr (kSeg) {
    const unsigned row0 = m0 + seg_fl * CH;
    a_off = 0x80000000u;
    if (row0 < M) {
      const long long f = frame_begin + row0 / CH;
      const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * CH;
      a_off = static_cast<unsigned>((e_row - e_base + seg_o) * 4);
    }
  }
  // table DMA: wave w copies rows 2 w, 2 w + 1 of the stage's 16 x 128 tile (1 KiB, lane-linear)
  const float *b_src = tb.cos_t + n0 + static_cast<size_t>(2 * wave + (lane >> 5)) * kHopI + (lane & 31) * 4;

  float a_raw[kAPer];
  f32x4 a_seg = {0.f, 0.f, 0.f, 0.f};
  auto issue_a = [&](int i0) {  // asm: hipcc must not count these loads (see lds_fetch4)
    const unsigned o = a_off + static_cast<unsigned>(i0) * i_bytes;
    if constexpr (kSeg) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(a_seg) : "v"(o), "s"(a_rsrc) : "memory");
      return;
    }
    asm volatile(
        "buffer_load_dword %0, %4, %8, 0 offen\n\t"
        "buffer_load_dword %1, %5, %8, 0 offen\n\t"
        "buffer_load_dword %2, %6, %8, 0 offen\n\t"
        "buffer_load_dword %3, %7, %8, 0 offen"
        : "=&v"(a_raw[0]), "=&v"(a_raw[1]), "=&v"(a_raw[2]), "=&v"(a_raw[3])
        : "v"(o), "v"(o + a_step), "v"(o + 2 * a_step), "v"(o + 3 * a_step), "s"(a_rsrc)
        : "memory");
  };
  auto issue_b = [&](int i0, int slot) {
    __builtin_amdgcn_global_load_lds(b_src + static_cast<size_t>(i0) * kHopI, &Bs[slot][2 * wave * BN], 16, 0, 0);
  };
  auto store_a = [&](int i0, int slot) {
    if constexpr (kSeg) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = seg_o + j;  // float e of the segment: sample i = e / CH of channel e % CH
        const int ii = e / CH;
        As[slot][ii * kAS + seg_fl * CH + e % CH] = mul_rn(a_seg[j], Ws[i0 + ii]);  // :480
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < kAPer; ++j) {
      const int ii = a_i + kAStride * j;
      As[slot][ii * kAS + a_r] = mul_rn(a_raw[j], Ws[i0 + ii]);  // block[i] = slice[i]*window[i], :480
    }
  };
  auto wait_staged = [&]() {  // everything this wave has in flight: PCM registers + table DMA of the next stage
    if constexpr (kSeg)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_seg)::"memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_raw[0]), "+v"(a_raw[1]), "+v"(a_raw[2]), "+v"(a_raw[3])::"memory");
  };

  f32x2 acc[TM][4];
#pragma unroll
  for (int r = 0; r < TM; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = f32x2{0.0f, 0.0f};

  constexpr int kStages = kFrameI / BK;
  // prologue: stage 0 complete in slot 0, table of stage 1 in flight to slot 1, PCM of stage 1 in regs
  __syncthreads();  // Ws
  issue_a(0);
  issue_b(0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+v"(a_raw[0]), "+v"(a_raw[1]), "+v"(a_raw[2]), "+v"(a_raw[3]), "+v"(a_seg)::"memory");
  store_a(0, 0);
  issue_a(BK);
  issue_b(BK, 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  const unsigned a_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&As[0][ty * 4]));
  const unsigned b_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&Bs[0][tx * 4]));

  // Stage hand-off in the MIDDLE of a stage (round 2): the next stage's tiles are published by a
  // barrier after the first 8 i-steps, so the stage's last i-step prefetches the first operands of
  // the next stage and no wave starts a stage with an exposed LDS round trip.  The table DMA and the
  // PCM loads of stage s+2 are issued right after that barrier - every wave has then left stage s-1,
  // whose slot the DMA overwrites - and have 1.5 stages to land, so the mid-stage wait is a plain
  // vmcnt(0) on loads that are a whole stage old.  Slot use: stage s reads slot s % 3; As[(s+1) % 3]
  // is written before the barrier of stage s (last read in stage s-2), Bs[(s+2) % 3] after it.
  // Measured against the end-of-stage hand-off with a counted vmcnt(1) (round 1, kept as
  // k1x::k_mdct_fwd_dma in tools/k1_variants.hpp): the same time within run-to-run noise (0.58 ms,
  // profiles/r02_k1_tune_mid_stage.txt) - the hand-off is not where the idle issue slots come from -
  // so the simpler protocol (one plain wait on stage-old loads, no exposed fetch) is the one shipped.
  Operands X, Y;
  lds_fetch4<kAS, BN>(X, a_lds0, b_lds0, 0);
  lds_wait4(X);
  if constexpr (STAMP) {
    if (tid == 0) stamps[static_cast<size_t>(blockIdx.x) * 8] = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    if constexpr (PRIO >= 5) {
      // a ladder that gets finer towards the end (the tail of a launch is as long as its last rung):
      //   5: [0, 64) 3, [64, 96) 2, [96, 112) 1, [112, 128) 0      6: [0, 48) 3, [48, 96) 2, [96, 120) 1, [120, 128) 0
      constexpr int b1 = PRIO == 5 ? 64 : 48, b2 = 96, b3 = PRIO == 5 ? 112 : 120;
      if (s == b1) __builtin_amdgcn_s_setprio(2);
      else if (s == b2) __builtin_amdgcn_s_setprio(1);
      else if (s == b3) __builtin_amdgcn_s_setprio(0);
    } else if constexpr (PRIO != 0) {
      constexpr int kShift = PRIO == 1 ? 5 : PRIO == 2 ? 1 : PRIO == 3 ? 2 : 3;  // stages per level = 1 << kShift
      if ((s & ((1 << kShift) - 1)) == 0) {
        const int level = (s >> kShift) & 3;
        if (level == 0) __builtin_amdgcn_s_setprio(3);
        else if (level == 1) __builtin_amdgcn_s_setprio(2);
        else if (level == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
    }
    if constexpr (STAMP) {
      if (tid == 0 && (s & 31) == 0 && s) stamps[static_cast<size_t>(blockIdx.x) * 8 + (s >> 5)] = __builtin_amdgcn_s_memrealtime();
    }
    const int slot = s % 3, nslot = (s + 1) % 3;
    const unsigned a_addr = a_lds0 + slot * (BK * kAS * 4);
    const unsigned b_addr = b_lds0 + slot * (BK * BN * 4);
    const unsigned a_next = a_lds0 + nslot * (BK * kAS * 4);
    const unsigned b_next = b_lds0 + nslot * (BK * BN * 4);
#pragma unroll
    for (int ii = 0; ii < BK / 2; ii += 2) {
      step4<kAS, BN, true>(acc, X, Y, a_addr, b_addr, ii + 1);
      step4<kAS, BN, true>(acc, Y, X, a_addr, b_addr, ii + 2);
    }
    // stage s+1: its PCM registers and table DMA (both issued in the middle of stage s-1) have landed
    wait_staged();
    store_a(((s + 1) & (kStages - 1)) * BK, nslot);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's ds_writes of stage s+1 have landed
    __builtin_amdgcn_s_barrier();
    issue_b(((s + 2) & (kStages - 1)) * BK, (s + 2) % 3);  // the slot of stage s-1: every wave has left it
    issue_a(((s + 2) & (kStages - 1)) * BK);
#pragma unroll
    for (int ii = BK / 2; ii < BK; ii += 2) {
      step4<kAS, BN, true>(acc, X, Y, a_addr, b_addr, ii + 1);
      if (ii + 2 < BK) step4<kAS, BN, true>(acc, Y, X, a_addr, b_addr, ii + 2);
      else step4<kAS, BN, true>(acc, Y, X, a_next, b_next, 0);  // the first operands of stage s+1
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the wrap-around prefetches
  if constexpr (STAMP) {
    if (tid == 0) stamps[static_cast<size_t>(blockIdx.x) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
  }

#pragma unroll
  for (int r = 0; r < TM; ++r) {
    const unsigned row = m0 + ty * 4 + r;
    if (row >= M) continue;
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0;
    float4 o;
    o.x = mul_rn(acc[r][0].x, tb.norm); o.y = mul_rn(acc[r][0].y, tb.norm);
    o.z = mul_rn(acc[r][1].x, tb.norm); o.w = mul_rn(acc[r][1].y, tb.norm);
    *reinterpret_cast<float4 *>(dst + tx * 4) = o;
    o.x = mul_rn(acc[r][2].x, tb.norm); o.y = mul_rn(acc[r][2].y, tb.norm);
    o.z = mul_rn(acc[r][3].x, tb.norm); o.w = mul_rn(acc[r][3].y, tb.norm);
    *reinterpret_cast<float4 *>(dst + BN / 2 + tx * 4) = o;
  }
}

template <int MINW, int CH = 0, int PRIO = 0, bool STAMP = false>
inline hipError_t launch_dma(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M,
                             float *coef, hipStream_t s, unsigned long long *stamps = nullptr) {
  if (M == 0) return hipSuccess;
  if (CH != 0 && pcm.ch != static_cast<uint32_t>(CH)) return hipErrorInvalidValue;
  if (STAMP && !stamps) return hipErrorInvalidValue;
  const unsigned m_tiles = (M + 127) / 128;
  hipLaunchKernelGGL((k_mdct_fwd_dma<MINW, CH, PRIO, STAMP>), dim3(m_tiles * 8), dim3(512), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), M, coef, stamps);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// Scalar-table kernel (256 x 64 tile, 512 threads, 4 rows x 8 columns per lane, lanes <-> ROWS).
// The other kernels give a lane its own columns, so the table value is the per-lane operand and the
// sample the shared one - and a shared operand that is windowed on the fly cannot come from anywhere
// but LDS.  Turn the tile round: the 64 lanes of a wave hold 256 DIFFERENT rows and the SAME 8 columns.
// The 8 table values of an i-step are then wave-uniform and static - one s_load_dwordx8 straight from
// the table in global memory into SGPRs, no table tile in LDS at all - and the lane's 4 windowed samples
// are one ds_read_b128.  Per i-step and wave: 1 KiB out of LDS instead of 3 KiB, one operand of every
// v_pk_mul from the scalar file (tools/microbench_sgpr.hip: the same stream holds 3-5 % more clock
// that way).  Same products, same ascending-i adds, one accumulator per output.
// Scalar loads return out of order, so every wait is lgkmcnt(0); operands are therefore fetched TWO
// i-steps at a time, a whole pair ahead (as k_imdct_apply does with its records).
// ------------------------------------------------------------------------------------------
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

template <int D>
struct StOps {  // operands of D i-steps: 4 rows of this lane (VGPRs), 8 columns of this wave (SGPRs)
  f32x4 a[D];
  u32x8 b[D];
};

// II: i-step inside the stage (compile time: all offsets are immediates)
template <int II>
__device__ __forceinline__ void st_fetch_b(u32x8 &b, const unsigned *brow) {
  asm volatile("s_load_dwordx8 %0, %1, %c2" : "=&s"(b) : "s"(brow), "i"(II * kHopI * 4) : "memory");
}
template <int II, int AS>
__device__ __forceinline__ void st_fetch_a(f32x4 &a, unsigned a_addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(a) : "v"(a_addr), "i"(II * AS * 4) : "memory");
}
template <int II, int AS, int D>
__device__ __forceinline__ void st_fetch(StOps<D> &o, unsigned a_addr, const unsigned *brow) {
  static_assert(D == 2 || D == 4, "i-steps per fetch");
  st_fetch_b<II>(o.b[0], brow);
  st_fetch_b<II + 1>(o.b[1], brow);
  if constexpr (D == 4) {
    st_fetch_b<II + 2>(o.b[2], brow);
    st_fetch_b<II + 3>(o.b[3], brow);
  }
  st_fetch_a<II, AS>(o.a[0], a_addr);
  st_fetch_a<II + 1, AS>(o.a[1], a_addr);
  if constexpr (D == 4) {
    st_fetch_a<II + 2, AS>(o.a[2], a_addr);
    st_fetch_a<II + 3, AS>(o.a[3], a_addr);
  }
}

// scalar and vector results are tied in SEPARATE statements (an asm with one VGPR output makes all of
// its outputs divergent to LLVM, and the table values would be copied to VGPRs)
template <int D>
__device__ __forceinline__ void st_wait(StOps<D> &o) {
  if constexpr (D == 2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(o.b[0]), "+s"(o.b[1])::"memory");
    asm volatile("" : "+v"(o.a[0]), "+v"(o.a[1])::"memory");
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(o.b[0]), "+s"(o.b[1]), "+s"(o.b[2]), "+s"(o.b[3])::"memory");
    asm volatile("" : "+v"(o.a[0]), "+v"(o.a[1]), "+v"(o.a[2]), "+v"(o.a[3])::"memory");
  }
}

// rows (r, r+1) x 8 columns, table pairs in SGPRs: the instruction block of mac2rows
__device__ __forceinline__ void mac2rows_st(f32x2 (&c0)[4], f32x2 (&c1)[4], f32x2 a, u32x2 b0, u32x2 b1, u32x2 b2,
                                            u32x2 b3) {
  f32x2 t0, t1, t2, t3, t4, t5, t6, t7;
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
      : "v"(a), "s"(b0), "s"(b1), "s"(b2), "s"(b3));
}

// the fetch's i-steps: ascending i, for every output
template <int D>
__device__ __forceinline__ void st_mac(f32x2 (&acc)[4][4], const StOps<D> &o) {
#pragma unroll
  for (int d = 0; d < D; ++d) {
    mac2rows_st(acc[0], acc[1], o.a[d].xy, o.b[d].s01, o.b[d].s23, o.b[d].s45, o.b[d].s67);
    mac2rows_st(acc[2], acc[3], o.a[d].zw, o.b[d].s01, o.b[d].s23, o.b[d].s45, o.b[d].s67);
  }
}

// NW waves per workgroup (8: 256 x 64 tile, two workgroups per CU;  16: 256 x 128, one), BK i-steps per
// LDS stage, D i-steps per operand fetch.
// PRIO 1: priority by quarter of the i loop (between the two workgroups of a CU, as k_mdct_fwd_dma);
// PRIO 2: by distance from the last barrier (inside a workgroup: whoever is behind goes first).
// ABL (tuning only, wrong results): 1 = the table address does not advance, 2 = no staging and no barrier.
// STAMP (tuning only): workgroup timeline into `stamps` - entry, loop start, the quarter points, loop end, stores drained.
template <int MINW, int CH = 0, int PRIO = 0, int D = 2, int NW = 8, int BK = 16, int ABL = 0, bool STAMP = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(MINW, MINW)))
void k_mdct_fwd_st(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M, float *__restrict__ coef,
                   unsigned long long *__restrict__ stamps) {
  if constexpr (STAMP) {
    if (threadIdx.x == 0) stamps[static_cast<size_t>(blockIdx.x) * 8] = __builtin_amdgcn_s_memrealtime();
  }
  // CH as in k_mdct_fwd_dma: 0 = one PCM dword per (row, i) and lane; 1 / 2 / 4 / 8 = the stream's
  // channel count, BK samples x CH channels of a frame fetched as BK / 4 * CH dwordx4.
  constexpr int BM = 256, BN = 8 * NW, RING = 3;
  constexpr int kAS = BM + 4;  // the same bank spread as GLC_K1_A_STRIDE (260 = 132 = 4 mod 64)
  constexpr int kThreads = NW * 64;
  constexpr int kNTiles = kHopI / BN;
  constexpr bool kSeg = CH != 0;
  static_assert(!kSeg || CH == 1 || CH == 2 || CH == 4 || CH == 8, "segment loader shapes");
  static_assert((BK == 16 || BK == 32) && BK % (2 * D) == 0, "stage depth");
  constexpr int kIGroups = kThreads / BM;          // per-row loader: i = a_i + kIGroups * j
  constexpr int kAPer = BK / kIGroups;             //   dwords per lane and stage
  constexpr int kPieces = BM * BK / 4 / kThreads;  // segment loader: dwordx4 per lane and stage
  static_assert(kAPer == 4 || kAPer == 8, "per-row loader shapes");
  static_assert(kPieces == 1 || kPieces == 2, "segment loader shapes");
  __shared__ __attribute__((aligned(16))) float As[RING][BK * kAS];
  __shared__ __attribute__((aligned(16))) float Ws[kFrameI];

  const int tid = threadIdx.x;
  const unsigned g = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-aware tile map
  const int n_tile = g % kNTiles;
  const int m_tile = g / kNTiles;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

  for (int i = tid; i < kFrameI; i += kThreads) Ws[i] = tb.window[i];

  const long long ch = pcm.ch;
  const long long f0 = frame_begin + m0 / pcm.ch;
  const long long e_first = (f0 * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch;
  long long e_end = static_cast<long long>(pcm.t_count) * ch;
  const long long n_rel = static_cast<long long>(pcm.n_samples) - static_cast<long long>(pcm.t0) * ch;
  if (e_end > n_rel) e_end = n_rel;
  const long long e_base = e_first < 0 ? 0 : e_first;
  long long e_cnt = e_end - e_base;
  if (e_cnt < 0) e_cnt = 0;
  if (e_cnt > (1ll << 28)) e_cnt = 1ll << 28;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(pcm.p + e_base), 0, static_cast<int>(e_cnt * 4), 0x00020000);

  // per-row loader: lane -> (row a_r of the tile, i = a_i + kIGroups * j)
  const int a_r = tid % BM;
  const int a_i = tid / BM;
  unsigned a_off[2] = {0x80000000u, 0x80000000u};
  const unsigned a_step = static_cast<unsigned>(kIGroups * ch * 4);
  const unsigned i_bytes = static_cast<unsigned>(ch * 4);
  // segment loader: (lane, p) -> (frame seg_fl + p * kSegHalf of the tile, 4 consecutive floats of its BK x CH segment)
  constexpr int kSegCh = kSeg ? CH : 1;
  constexpr int kSegLanes = BK / 4 * kSegCh;  // lanes per frame segment
  constexpr int kSegHalf = kThreads / kSegLanes;
  const int seg_fl = tid / kSegLanes;
  const int seg_o = (tid % kSegLanes) * 4;
  if constexpr (kSeg) {
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const unsigned row0 = m0 + (seg_fl + p * kSegHalf) * CH;
      if (row0 < M) {
        const long long f = frame_begin + row0 / CH;
        const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * CH;
        a_off[p] = static_cast<unsigned>((e_row - e_base + seg_o) * 4);
      }
    }
  } else {
    const unsigned a_row = m0 + a_r;
    if (a_row < M) {
      const long long f = frame_begin + a_row / pcm.ch;
      const long long c = a_row % pcm.ch;
      const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch + c;
      a_off[0] = static_cast<unsigned>((e_row - e_base + a_i * ch) * 4);
    }
  }

  float a_raw[8];
  f32x4 a_seg[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  auto issue_a = [&](int i0) {  // asm: hipcc must not count these loads (see lds_fetch4)
    if constexpr (kSeg) {
      const unsigned o0 = a_off[0] + static_cast<unsigned>(i0) * i_bytes;
      if constexpr (kPieces == 1) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(a_seg[0]) : "v"(o0), "s"(a_rsrc) : "memory");
      } else {
        const unsigned o1 = a_off[1] + static_cast<unsigned>(i0) * i_bytes;
        asm volatile("buffer_load_dwordx4 %0, %2, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen"
                     : "=&v"(a_seg[0]), "=&v"(a_seg[1])
                     : "v"(o0), "v"(o1), "s"(a_rsrc)
                     : "memory");
      }
      return;
    }
    const unsigned o = a_off[0] + static_cast<unsigned>(i0) * i_bytes;
    asm volatile(
        "buffer_load_dword %0, %4, %8, 0 offen\n\t"
        "buffer_load_dword %1, %5, %8, 0 offen\n\t"
        "buffer_load_dword %2, %6, %8, 0 offen\n\t"
        "buffer_load_dword %3, %7, %8, 0 offen"
        : "=&v"(a_raw[0]), "=&v"(a_raw[1]), "=&v"(a_raw[2]), "=&v"(a_raw[3])
        : "v"(o), "v"(o + a_step), "v"(o + 2 * a_step), "v"(o + 3 * a_step), "s"(a_rsrc)
        : "memory");
    if constexpr (kAPer == 8)
      asm volatile(
          "buffer_load_dword %0, %4, %8, 0 offen\n\t"
          "buffer_load_dword %1, %5, %8, 0 offen\n\t"
          "buffer_load_dword %2, %6, %8, 0 offen\n\t"
          "buffer_load_dword %3, %7, %8, 0 offen"
          : "=&v"(a_raw[4]), "=&v"(a_raw[5]), "=&v"(a_raw[6]), "=&v"(a_raw[7])
          : "v"(o + 4 * a_step), "v"(o + 5 * a_step), "v"(o + 6 * a_step), "v"(o + 7 * a_step), "s"(a_rsrc)
          : "memory");
  };
  auto store_a = [&](int i0, int slot) {
    if constexpr (kSeg) {
#pragma unroll
      for (int p = 0; p < kPieces; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = seg_o + j;  // float e of the segment: sample i = e / CH of channel e % CH
          const int ii = e / CH;
          As[slot][ii * kAS + (seg_fl + p * kSegHalf) * CH + e % CH] = mul_rn(a_seg[p][j], Ws[i0 + ii]);  // :480
        }
      return;
    }
#pragma unroll
    for (int j = 0; j < kAPer; ++j) {
      const int ii = a_i + kIGroups * j;
      As[slot][ii * kAS + a_r] = mul_rn(a_raw[j], Ws[i0 + ii]);  // block[i] = slice[i]*window[i], :480
    }
  };
  auto wait_staged = [&]() {
    if constexpr (kSeg)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_seg[0]), "+v"(a_seg[1])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(a_raw[0]), "+v"(a_raw[1]), "+v"(a_raw[2]), "+v"(a_raw[3]), "+v"(a_raw[4]), "+v"(a_raw[5]),
                     "+v"(a_raw[6]), "+v"(a_raw[7])::"memory");
  };

  f32x2 acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = f32x2{0.0f, 0.0f};

  constexpr int kStages = kFrameI / BK;
  // prologue: stage 0 complete in slot 0, PCM of stage 1 in registers
  __syncthreads();  // Ws
  issue_a(0);
  wait_staged();
  store_a(0, 0);
  issue_a(BK);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  const unsigned a_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&As[0][lane * 4]));
  // this wave's 8 columns of table row 0 (wave-uniform: the scalar loads' base)
  const unsigned *b_base = reinterpret_cast<const unsigned *>(tb.cos_t) + n0 + 8 * wave;

  // Stage hand-off in the middle of a stage, as in k_mdct_fwd_dma: the samples of stage s+1 are
  // published by a barrier after the first BK / 2 i-steps, the stage's last fetch takes the first operands
  // of stage s+1, and the PCM loads of stage s+2 are issued behind the barrier.  Slot use: stage s reads
  // slot s % 3; As[(s+1) % 3] is written before the barrier of stage s (last read in stage s-2).
  StOps<D> X, Y;
  st_fetch<0, kAS>(X, a_lds0, b_base);
  st_wait(X);
  if constexpr (STAMP) {
    if (tid == 0) stamps[static_cast<size_t>(blockIdx.x) * 8 + 1] = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    if constexpr (STAMP) {
      if (tid == 0 && s && s % (kStages / 4) == 0) stamps[static_cast<size_t>(blockIdx.x) * 8 + 1 + s / (kStages / 4)] = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (PRIO == 1) {
      constexpr int kQuarter = kStages / 4;
      if (s % kQuarter == 0) {
        const int level = (s / kQuarter) & 3;
        if (level == 0) __builtin_amdgcn_s_setprio(3);
        else if (level == 1) __builtin_amdgcn_s_setprio(2);
        else if (level == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
    }
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);  // half-way between two barriers
    const int slot = (ABL & 2) ? 0 : s % 3, nslot = (ABL & 2) ? 0 : (s + 1) % 3;
    const unsigned a_addr = a_lds0 + slot * (BK * kAS * 4);
    const unsigned a_next = a_lds0 + nslot * (BK * kAS * 4);
    const unsigned *brow = (ABL & 1) ? b_base : b_base + static_cast<size_t>(s) * (BK * kHopI);
    const unsigned *brow_next = (ABL & 1) ? b_base : b_base + static_cast<size_t>((s + 1) & (kStages - 1)) * (BK * kHopI);
    // the hand-off: PCM registers of stage s+1 (issued in the middle of stage s-1) have landed -> LDS,
    // lgkmcnt(0) (this wave's ds_writes and the operands in flight), barrier, PCM loads of stage s+2
#define GLC_ST_HANDOFF(NEXT)                                      \
  do {                                                            \
    wait_staged();                                                \
    store_a(((s + 1) & (kStages - 1)) * BK, nslot);               \
    st_wait(NEXT);                                                \
    __builtin_amdgcn_s_barrier();                                 \
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);       \
    issue_a(((s + 2) & (kStages - 1)) * BK);                      \
  } while (0)
    // X holds the operands of i-steps [0, D) of the stage; groups alternate X, Y
#define GLC_ST_GROUP(CUR, NXT, II)                                                        \
  do {                                                                                    \
    if constexpr ((II) + D < BK) st_fetch<((II) + D) % BK, kAS>(NXT, a_addr, brow);        \
    else st_fetch<0, kAS>(NXT, a_next, brow_next);                                        \
    st_mac(acc, CUR);                                                                     \
    if constexpr ((II) + D == BK / 2 && !(ABL & 2)) GLC_ST_HANDOFF(NXT);                  \
    else st_wait(NXT);                                                                    \
  } while (0)
    GLC_ST_GROUP(X, Y, 0);
    GLC_ST_GROUP(Y, X, D);
    if constexpr (BK / D > 2) {
      GLC_ST_GROUP(X, Y, 2 * D);
      GLC_ST_GROUP(Y, X, 3 * D);
    }
    if constexpr (BK / D > 4) {
      GLC_ST_GROUP(X, Y, 4 * D);
      GLC_ST_GROUP(Y, X, 5 * D);
      GLC_ST_GROUP(X, Y, 6 * D);
      GLC_ST_GROUP(Y, X, 7 * D);
    }
    if constexpr (BK / D > 8) {
      GLC_ST_GROUP(X, Y, 8 * D);
      GLC_ST_GROUP(Y, X, 9 * D);
      GLC_ST_GROUP(X, Y, 10 * D);
      GLC_ST_GROUP(Y, X, 11 * D);
      GLC_ST_GROUP(X, Y, 12 * D);
      GLC_ST_GROUP(Y, X, 13 * D);
      GLC_ST_GROUP(X, Y, 14 * D);
      GLC_ST_GROUP(Y, X, 15 * D);
    }
#undef GLC_ST_GROUP
#undef GLC_ST_HANDOFF
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the wrap-around prefetch
  if constexpr (STAMP) {
    if (tid == 0) stamps[static_cast<size_t>(blockIdx.x) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
  }

#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned row = m0 + lane * 4 + r;
    if (row >= M) continue;
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0 + 8 * wave;
    float4 o;
    o.x = mul_rn(acc[r][0].x, tb.norm); o.y = mul_rn(acc[r][0].y, tb.norm);
    o.z = mul_rn(acc[r][1].x, tb.norm); o.w = mul_rn(acc[r][1].y, tb.norm);
    *reinterpret_cast<float4 *>(dst) = o;
    o.x = mul_rn(acc[r][2].x, tb.norm); o.y = mul_rn(acc[r][2].y, tb.norm);
    o.z = mul_rn(acc[r][3].x, tb.norm); o.w = mul_rn(acc[r][3].y, tb.norm);
    *reinterpret_cast<float4 *>(dst + 4) = o;
  }
  if constexpr (STAMP) {  // this wave's stores have left the CU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) stamps[static_cast<size_t>(blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int MINW, int CH = 0, int PRIO = 0, int D = 2, int NW = 8, int BK = 16, int ABL = 0, bool STAMP = false>
inline hipError_t launch_st(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M, float *coef,
                            hipStream_t s, unsigned long long *stamps = nullptr) {
  if (M == 0) return hipSuccess;
  if (CH != 0 && pcm.ch != static_cast<uint32_t>(CH)) return hipErrorInvalidValue;
  if (STAMP && !stamps) return hipErrorInvalidValue;
  const unsigned m_tiles = (M + 255) / 256;
  hipLaunchKernelGGL((k_mdct_fwd_st<MINW, CH, PRIO, D, NW, BK, ABL, STAMP>), dim3(m_tiles * (kHopI / (8 * NW))), dim3(NW * 64), 0,
                     s, t, pcm, static_cast<long long>(frame_begin), M, coef, stamps);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// Short clips (<= 1792 rows: BASELINE config 1 and every clip of the reference's own tests).  A launch
// this small cannot fill the chip; what it costs is ONE wave's chain of 2048 dependent i-steps, and the
// length of a step is the lane tile: 128 VALU cycles for 4 x 8 outputs, 16 for 2 x 2.  So the tile is
// cut until every SIMD of the chip has a wave of its own - 2 x 2 outputs per lane up to 256 rows, 2 x 4
// up to 512 (65 536 lanes either way) - in 256-thread workgroups (one wave per SIMD of a CU) on
// 32 x 32 / 32 x 64 tiles.  Same arithmetic, same order; hand-scheduled like the large kernels:
// operands come from LDS through a four-deep register ring with counted lgkmcnt waits (LDS returns in
// order), the next stage's global loads are in flight during the stage and only waited for at its end.
// (Round 2 left this range to hipcc on a 4 x 4 lane tile: 129 VGPRs + 80 B of scratch, 0.13 ms at 172 rows.)
// ------------------------------------------------------------------------------------------
#ifndef GLC_SMALL_PAIR
#define GLC_SMALL_PAIR 1  // k_mdct_fwd_small waits for its LDS operands once per two i-steps (tuning: k1_tune)
#endif
#ifndef GLC_SMALL_BK
#define GLC_SMALL_BK 32  // i-steps per LDS stage of k_mdct_fwd_small (64 measured 4 % slower: gpurun r3d2)
#endif
template <int TN>
struct SmallOps {  // operands of one i-step: a = 2 rows, b = TN columns
  f32x2 a;
  f32x2 b[TN / 2];
};

template <int TN, int BM, int BN>
__device__ __forceinline__ void small_fetch(SmallOps<TN> &o, unsigned a_addr, unsigned b_addr, int ii) {
  if constexpr (TN == 2) {
    asm volatile(
        "ds_read_b64 %0, %2 offset:%c4\n\t"
        "ds_read_b64 %1, %3 offset:%c5"
        : "=&v"(o.a), "=&v"(o.b[0])
        : "v"(a_addr), "v"(b_addr), "i"(ii * BM * 4), "i"(ii * BN * 4)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b64 %0, %3 offset:%c5\n\t"
        "ds_read_b64 %1, %4 offset:%c6\n\t"
        "ds_read_b64 %2, %4 offset:%c7"
        : "=&v"(o.a), "=&v"(o.b[0]), "=&v"(o.b[1])
        : "v"(a_addr), "v"(b_addr), "i"(ii * BM * 4), "i"(ii * BN * 4), "i"(ii * BN * 4 + 8)
        : "memory");
  }
}

// wait until at most PENDING of this wave's LDS reads are outstanding, i.e. `o` (older than those) has landed
template <int TN, int PENDING>
__device__ __forceinline__ void small_wait(SmallOps<TN> &o) {
  if constexpr (TN == 2)
    asm volatile("s_waitcnt lgkmcnt(%c2)" : "+v"(o.a), "+v"(o.b[0]) : "i"(PENDING) : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(%c3)" : "+v"(o.a), "+v"(o.b[0]), "+v"(o.b[1]) : "i"(PENDING) : "memory");
}

// c[r][j] += a_r * b_j: rows (a.x, a.y) x TN columns, multiply and add separately rounded
template <int TN>
__device__ __forceinline__ void small_mac(f32x2 (&acc)[2][TN / 2], const SmallOps<TN> &o) {
  if constexpr (TN == 2) {
    f32x2 t0, t1;
    asm volatile(
        "v_pk_mul_f32 %2, %4, %5 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %4, %5 op_sel:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\t"
        "v_pk_add_f32 %1, %1, %3"
        : "+v"(acc[0][0]), "+v"(acc[1][0]), "=&v"(t0), "=&v"(t1)
        : "v"(o.a), "v"(o.b[0]));
  } else {
    f32x2 t0, t1, t2, t3;
    asm volatile(
        "v_pk_mul_f32 %4, %8, %9 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %5, %8, %10 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %6, %8, %9 op_sel:[1,0]\n\t"
        "v_pk_mul_f32 %7, %8, %10 op_sel:[1,0]\n\t"
        "v_pk_add_f32 %0, %0, %4\n\t"
        "v_pk_add_f32 %1, %1, %5\n\t"
        "v_pk_add_f32 %2, %2, %6\n\t"
        "v_pk_add_f32 %3, %3, %7"
        : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(o.a), "v"(o.b[0]), "v"(o.b[1]));
  }
}

// i-steps II .. BK-1 of a stage, written out at compile time: wait for the oldest step of the ring
// (LDS returns in order: all but the younger steps' reads must have landed), use it, refill its slot
template <int TN, int BM, int BN, int BK, int D, int II>
__device__ __forceinline__ void small_steps(f32x2 (&acc)[2][TN / 2], SmallOps<TN> (&ring)[D], unsigned a_addr, unsigned b_addr) {
  if constexpr (II < BK) {
    constexpr int R = 1 + TN / 2;
#if GLC_SMALL_PAIR
    // two i-steps per wait: one s_waitcnt and one burst of LDS reads per 2 x 4 vector ops
    static_assert(BK % 2 == 0 && D % 2 == 0, "pairs");
    constexpr int younger = II + D <= BK ? D - 2 : BK - 2 - II;  // steps behind II + 1 still in flight
    static_assert(younger * R <= 15, "lgkmcnt is a 4-bit counter");
    small_wait<TN, younger * R>(ring[(II + 1) % D]);
    asm volatile("" : "+v"(ring[II % D].a), "+v"(ring[II % D].b[0]));  // (II landed before II + 1: LDS returns in order)
    if constexpr (TN == 4) asm volatile("" : "+v"(ring[II % D].b[1]));
    small_mac<TN>(acc, ring[II % D]);
    small_mac<TN>(acc, ring[(II + 1) % D]);
    if constexpr (II + D < BK) {
      small_fetch<TN, BM, BN>(ring[II % D], a_addr, b_addr, II + D);
      small_fetch<TN, BM, BN>(ring[(II + 1) % D], a_addr, b_addr, II + D + 1);
    }
    small_steps<TN, BM, BN, BK, D, II + 2>(acc, ring, a_addr, b_addr);
#else
    constexpr int younger = II + D <= BK ? D - 1 : BK - 1 - II;
    static_assert(younger * R <= 15, "lgkmcnt is a 4-bit counter");
    small_wait<TN, younger * R>(ring[II % D]);
    small_mac<TN>(acc, ring[II % D]);
    if constexpr (II + D < BK) small_fetch<TN, BM, BN>(ring[II % D], a_addr, b_addr, II + D);
    small_steps<TN, BM, BN, BK, D, II + 1>(acc, ring, a_addr, b_addr);
#endif
  }
}

template <int TN>
__global__ __launch_bounds__(256) void k_mdct_fwd_small(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M,
                                                         float *__restrict__ coef) {
  // D: i-steps of operands in flight (an LDS read takes ~250 cycles under no load, a 2 x 2 step 16 of VALU:
  // the ring is as deep as the 4-bit lgkmcnt allows - 7 x 2 or 5 x 3 younger reads)
  constexpr int BM = 32, BN = 16 * TN, BK = GLC_SMALL_BK, D = TN == 2 ? 8 : 6;
  constexpr int kNTiles = kHopI / BN;
  constexpr int kAStride = 8, kAPer = BK / kAStride;     // 256 threads = 32 rows x 8 i; i = a_i + 8 j
  constexpr int kBRowsPer = 256 / (BN / 4), kBPer = BK / kBRowsPer;
  static_assert(TN == 2 || TN == 4, "lane tile");
  __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

  const int tid = threadIdx.x;
  const int n_tile = blockIdx.x % kNTiles, m_tile = blockIdx.x / kNTiles;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int tx = tid % 16, ty = tid / 16;

  // A operand through a buffer descriptor whose range check is the encoder's zero padding (see k_mdct_fwd)
  const long long ch = pcm.ch;
  const long long f0 = frame_begin + m0 / pcm.ch;
  const long long e_first = (f0 * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch;
  long long e_end = static_cast<long long>(pcm.t_count) * ch;
  const long long n_rel = static_cast<long long>(pcm.n_samples) - static_cast<long long>(pcm.t0) * ch;
  if (e_end > n_rel) e_end = n_rel;
  const long long e_base = e_first < 0 ? 0 : e_first;
  long long e_cnt = e_end - e_base;
  if (e_cnt < 0) e_cnt = 0;
  if (e_cnt > (1ll << 28)) e_cnt = 1ll << 28;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(pcm.p + e_base), 0, static_cast<int>(e_cnt * 4), 0x00020000);
  const int a_r = tid % BM, a_i = tid / BM;
  const unsigned a_row = m0 + a_r;
  unsigned a_off = 0x80000000u;  // out-of-range row: every load returns 0
  if (a_row < M) {
    const long long f = frame_begin + a_row / pcm.ch;
    const long long c = a_row % pcm.ch;
    const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch + c;
    a_off = static_cast<unsigned>((e_row - e_base + a_i * ch) * 4);  // may wrap: that IS the padding
  }
  const unsigned a_step = static_cast<unsigned>(kAStride * ch * 4);
  const float *w_ptr = tb.window + a_i;
  const int b_r = tid / (BN / 4), b_c4 = tid % (BN / 4);
  const float *b_ptr = tb.cos_t + n0 + static_cast<size_t>(b_r) * kHopI + b_c4 * 4;

  float a_raw[kAPer], a_win[kAPer];
  f32x4 b_stage[kBPer];
  auto load_stage = [&](int i0) {
    const unsigned off0 = a_off + static_cast<unsigned>(i0) * static_cast<unsigned>(ch * 4);
#pragma unroll
    for (int j = 0; j < kAPer; ++j) {
      a_raw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off0 + j * a_step, 0, 0));
      a_win[j] = w_ptr[i0 + kAStride * j];
    }
#pragma unroll
    for (int j = 0; j < kBPer; ++j)
      b_stage[j] = *reinterpret_cast<const f32x4 *>(b_ptr + static_cast<size_t>(i0 + kBRowsPer * j) * kHopI);
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int j = 0; j < kAPer; ++j) {
      float r = a_raw[j], w = a_win[j];
      asm volatile("" : "+v"(r), "+v"(w));  // first use of the staged registers stays behind the stage's math
      As[buf][(a_i + kAStride * j) * BM + a_r] = mul_rn(r, w);  // block[i] = slice[i]*window[i], :480
    }
#pragma unroll
    for (int j = 0; j < kBPer; ++j) {
      f32x4 b = b_stage[j];
      asm volatile("" : "+v"(b));
      *reinterpret_cast<f32x4 *>(&Bs[buf][(b_r + kBRowsPer * j) * BN + b_c4 * 4]) = b;
    }
  };

  f32x2 acc[2][TN / 2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < TN / 2; ++c) acc[r][c] = f32x2{0.0f, 0.0f};  // `let mut s = 0.0f32`, :365

  load_stage(0);
  store_stage(0);
  __syncthreads();

  const unsigned a_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&As[0][ty * 2]));
  const unsigned b_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&Bs[0][tx * TN]));
  constexpr int kStages = kFrameI / BK;
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    const int buf = s & 1;
    load_stage(((s + 1) & (kStages - 1)) * BK);  // (the last iteration re-fetches stage 0 into the idle buffer)
    const unsigned a_addr = a_lds0 + buf * (BK * BM * 4);
    const unsigned b_addr = b_lds0 + buf * (BK * BN * 4);
    SmallOps<TN> ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) small_fetch<TN, BM, BN>(ring[d], a_addr, b_addr, d);
    small_steps<TN, BM, BN, BK, D, 0>(acc, ring, a_addr, b_addr);
    store_stage(buf ^ 1);
    __syncthreads();
  }

  // epilogue: out[k] = s * norm, :372
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const unsigned row = m0 + ty * 2 + r;
    if (row >= M) continue;
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0 + tx * TN;
    if constexpr (TN == 2) {
      float2 o;
      o.x = mul_rn(acc[r][0].x, tb.norm); o.y = mul_rn(acc[r][0].y, tb.norm);
      *reinterpret_cast<float2 *>(dst) = o;
    } else {
      float4 o;
      o.x = mul_rn(acc[r][0].x, tb.norm); o.y = mul_rn(acc[r][0].y, tb.norm);
      o.z = mul_rn(acc[r][1].x, tb.norm); o.w = mul_rn(acc[r][1].y, tb.norm);
      *reinterpret_cast<float4 *>(dst) = o;
    }
  }
}

template <int TN>
inline hipError_t launch_small(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M, float *coef,
                               hipStream_t s) {
  if (M == 0) return hipSuccess;
  const unsigned m_tiles = (M + 31) / 32;
  hipLaunchKernelGGL((k_mdct_fwd_small<TN>), dim3(m_tiles * (kHopI / (16 * TN))), dim3(256), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), M, coef);
  return hipGetLastError();
}

template <int BM, int BN, int BK, int TM, int TN, int UNROLL, int MINW>
inline hipError_t launch(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M,
                         float *coef, hipStream_t s) {
  using C = Cfg<BM, BN, BK, TM, TN, UNROLL, MINW>;
  if (M == 0) return hipSuccess;
  const unsigned m_tiles = (M + BM - 1) / BM;
  hipLaunchKernelGGL((k_mdct_fwd<BM, BN, BK, TM, TN, UNROLL, MINW>), dim3(m_tiles * C::kNTiles),
                     dim3(C::kThreads), 0, s, t, pcm, static_cast<long long>(frame_begin), M, coef);
  return hipGetLastError();
}

}  // namespace k1
}  // namespace glc
