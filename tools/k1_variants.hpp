// k1_variants.hpp — TUNING ONLY (namespace glc::k1x): every shape, schedule and ablation of the
// forward-MDCT kernel that was measured on the way to the shipped ones.  The library does not
// include this file; csrc/glc_mdct_fwd.hpp (namespace glc::k1) holds exactly the kernels
// libglc_hip.so instantiates.  tools/k1_tune.hip times these against the shipped kernels and
// checks all of them bit for bit against a naive kernel.
//
//   C[m][k] = fl( fl( sum_{i ascending} fl( fl(x[m,i]*w[i]) * T[k][i] ) ) * norm )
//
// replaces the window loop (src/codec.rs:476-481) and MdctTables::mdct_block (:359-374) of the
// reference for every frame-channel row m = (frame - frame_begin)*ch + c.  M = rows, N = 1024,
// K = 2048; the K loop is strictly ascending with ONE accumulator per output, multiply and add
// are separate instructions (v_pk_mul_f32 / v_pk_add_f32, never an FMA), there is no split-K.
//
// What is in this file
//   k_mdct_fwd_dma     SHIPPED for launches of >= 4096 rows (BASELINE config 2 = 8192): 128x128 tile,
//                      512 threads, 4x8 outputs per lane, table tile copied global -> LDS by LDS-DMA
//                      two stages ahead (3-slot ring), one counted vmcnt wait per stage; PCM tile by
//                      one dwordx4 per lane and stage when the stream has 1 / 2 / 4 / 8 channels
//                      (CH), one dword per (row, sample) otherwise.  BM = 64 (256 threads, window by
//                      scalar loads) is a measured alternative, not shipped.
//   k_mdct_fwd_sched   SHIPPED for 513..4095 rows as <64,128,16,4,0,4>.  Hand-scheduled inline-asm
//                      i-steps (step4 / mac2rows), LDS operand prefetch, XCD-aware tile map, register
//                      staging.  Other shapes and ABL / SCALAR / RING / WLDS are tuning knobs.
//   k_mdct_fwd         the same tiling left to hipcc's scheduler.  SHIPPED as <32,64,32,4,4,4,2> for
//                      launches of up to 512 rows (short clips: the latency of one workgroup's
//                      2048-step chain is everything, and smaller lane tiles shorten the step);
//                      larger shapes are tuning only (19-24 T MAC/s).
//   k_mdct_fwd_mx      tuning only: products by v_mfma_f32_32x32x1_2b_f32 with C = 0 (bit-identical to
//                      v_mul_f32, tools/mfma_probe.hip), accumulation by v_pk_add_f32.  Bit-exact,
//                      slower: the f32 matrix instruction runs on the vector ALU's multipliers.
// tools/k1_tune.hip times them against each other and checks every variant bit-for-bit against a
// naive kernel; profiles/r01_k1_tune_*.txt hold the numbers.
#pragma once
#include <hip/hip_runtime.h>

#include "glc_kernels.h"

namespace glc {
namespace k1x {

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
// Hand-scheduled variant (8x8 outputs per lane).  Same arithmetic, but the instruction order
// is pinned with inline asm because hipcc (ROCm 7.2) (a) places every v_pk_add directly behind
// the v_pk_mul it depends on and (b) sinks the LDS reads of the next i-step below the current
// step's math, so a wave stalls on both.  Here each i-step issues its 8 ds_read_b64 for the NEXT
// step first, then 4 groups of {8 v_pk_mul, 8 v_pk_add} (dependent instructions 8 apart), then
// one s_waitcnt: register set X feeds even steps, set Y odd steps (ping-pong, no copies).
// ------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Operands {  // one i-step's A (8 rows) and B (8 columns) values of this lane
  f32x2 a0, a1, a2, a3, b0, b1, b2, b3;
};

template <int BM, int BN>
__device__ __forceinline__ void lds_fetch4(Operands &o, unsigned a_addr, unsigned b_addr, int ii) {
  // 4-row lane tile: rows ty*4 .. ty*4+3 only (a2/a3 unused)
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

template <int BM, int BN>
__device__ __forceinline__ void lds_fetch(Operands &o, unsigned a_addr, unsigned b_addr, int ii) {
  // asm loads: hipcc does not count them; every use is behind lds_wait() (cdna guide §5.7 iii).
  // Outputs are early-clobber: the first ds_read must not land on the address registers the
  // later ones still need.
  asm volatile(
      "ds_read_b64 %0, %8 offset:%c10\n\t"
      "ds_read_b64 %1, %8 offset:%c11\n\t"
      "ds_read_b64 %2, %8 offset:%c12\n\t"
      "ds_read_b64 %3, %8 offset:%c13\n\t"
      "ds_read_b64 %4, %9 offset:%c14\n\t"
      "ds_read_b64 %5, %9 offset:%c15\n\t"
      "ds_read_b64 %6, %9 offset:%c16\n\t"
      "ds_read_b64 %7, %9 offset:%c17"
      : "=&v"(o.a0), "=&v"(o.a1), "=&v"(o.a2), "=&v"(o.a3), "=&v"(o.b0), "=&v"(o.b1), "=&v"(o.b2), "=&v"(o.b3)
      : "v"(a_addr), "v"(b_addr), "i"(ii * BM * 4), "i"(ii * BM * 4 + 8), "i"(ii * BM * 4 + BM * 2),
        "i"(ii * BM * 4 + BM * 2 + 8), "i"(ii * BN * 4), "i"(ii * BN * 4 + 8), "i"(ii * BN * 4 + BN * 2),
        "i"(ii * BN * 4 + BN * 2 + 8)
      : "memory");
}

__device__ __forceinline__ void lds_wait(Operands &o) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(o.a0), "+v"(o.a1), "+v"(o.a2), "+v"(o.a3), "+v"(o.b0), "+v"(o.b1), "+v"(o.b2), "+v"(o.b3)
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

#include "k1_step_scalar.inc"

template <int TM>
__device__ __forceinline__ void mac_step(f32x2 (&acc)[TM][4], const Operands &o) {
  mac2rows(acc[0], acc[1], o.a0, o.b0, o.b1, o.b2, o.b3);
  mac2rows(acc[2], acc[3], o.a1, o.b0, o.b1, o.b2, o.b3);
  if constexpr (TM == 8) {
    mac2rows(acc[4], acc[5], o.a2, o.b0, o.b1, o.b2, o.b3);
    mac2rows(acc[6], acc[7], o.a3, o.b0, o.b1, o.b2, o.b3);
  }
}

// ABL (tuning only, results are wrong when != 0): 1 = no staging/barrier inside the stage loop,
// 2 = additionally no LDS operand reads inside the loop (pure VALU stream), 3 = staging but no
// barrier, 4 = barrier but no staging.
// SCALAR = v_mul_f32/v_add_f32 stream (TM == 4 only) instead of the packed v_pk_* one: on
// gfx950 both forms have the same peak MAC rate, but the 2-cycle scalar ops reach it with fewer
// waves per SIMD (profiles/r01_microbench_valu_mfma.txt: A vs B).
// RING = LDS slots per operand tile: 2 = classic double buffer (stage s+1 is written at the end
// of stage s, so its ds_writes must land before the barrier); 3 = stage s+2 is written at the end
// of stage s into the slot nobody reads, the barrier publishes the writes of the PREVIOUS stage
// and needs no LDS wait in front of it.
// WLDS = the window (8 KiB) is copied to LDS once and read from there when a stage is written,
// instead of 4 global loads per thread and stage.
template <int BM, int BN, int BK, int MINW, int ABL = 0, int TM = 8, bool SCALAR = false, int RING = 2,
          bool WLDS = false>
__global__ __launch_bounds__((BM / TM) * (BN / 8)) __attribute__((amdgpu_waves_per_eu(MINW, MINW)))
void k_mdct_fwd_sched(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M,
                      float *__restrict__ coef) {
  using C = Cfg<BM, BN, BK, TM, 8, 2, MINW>;
  static_assert(!SCALAR || TM == 4, "scalar stream is written for the 4x8 lane tile");
  __shared__ __attribute__((aligned(16))) float As[RING][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[RING][BK * BN];
  __shared__ float Ws[WLDS ? kFrameI : 1];

  const int tid = threadIdx.x;
  if (WLDS)
    for (int i = tid; i < kFrameI; i += C::kThreads) Ws[i] = tb.window[i];
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
      if (!WLDS) a_win[j] = w_ptr[i0 + C::kAStride * j];
    }
#pragma unroll
    for (int j = 0; j < C::kBPer; ++j)
      b_stage[j] = *reinterpret_cast<const f32x4 *>(b_ptr + static_cast<size_t>(i0 + C::kBRowsPer * j) * kHopI);
  };
  auto store_stage = [&](int buf, int i0) {
    // pin the first use of the staged registers behind the stage's math (volatile asm
    // statements keep their order): hipcc otherwise hoists the multiply, and with it the
    // vmcnt wait, into the middle of the stage
#pragma unroll
    for (int j = 0; j < C::kAPer; ++j) {
      float r = a_raw[j], w = WLDS ? 0.0f : a_win[j];
      asm volatile("" : "+v"(r), "+v"(w));
      if (WLDS) w = Ws[i0 + a_i + C::kAStride * j];
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
  float accs[4][8];
#pragma unroll
  for (int r = 0; r < TM; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) accs[r][c] = 0.0f;
  auto fetch = [&](Operands &o, unsigned a_addr, unsigned b_addr, int ii) {
    if constexpr (TM == 8) lds_fetch<BM, BN>(o, a_addr, b_addr, ii);
    else lds_fetch4<BM, BN>(o, a_addr, b_addr, ii);
  };
  auto wait = [&](Operands &o) {
    if constexpr (TM == 8) lds_wait(o);
    else lds_wait4(o);
  };

  if (WLDS) __syncthreads();
  load_stage(0);
  store_stage(0, 0);
  if (RING == 3) {
    load_stage(BK);
    store_stage(1, BK);
  }
  __syncthreads();

  // LDS byte addresses of this lane's operand columns (low 32 bits of a generic LDS pointer)
  const unsigned a_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&As[0][ty * 4]));
  const unsigned b_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&Bs[0][tx * 4]));

  constexpr int kStages = kFrameI / BK;
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    const int buf = RING == 3 ? s % 3 : (s & 1);
    if (ABL == 0 || ABL == 3) load_stage(((s + RING - 1) & (kStages - 1)) * BK);
    const unsigned a_addr = a_lds0 + buf * (BK * BM * 4);
    const unsigned b_addr = b_lds0 + buf * (BK * BN * 4);
    Operands X, Y;
    fetch(X, a_addr, b_addr, 0);
    wait(X);
    if (ABL == 2) {
      fetch(Y, a_addr, b_addr, 1);
      wait(Y);
    }
    if constexpr (SCALAR && ABL != 2) {
#pragma unroll
      for (int ii = 0; ii < BK; ii += 2) {
        step4s<BM, BN, true>(accs, X, Y, a_addr, b_addr, ii + 1);
        if (ii + 2 < BK) step4s<BM, BN, true>(accs, Y, X, a_addr, b_addr, ii + 2);
        else step4s<BM, BN, false>(accs, Y, X, a_addr, b_addr, 0);
      }
    } else if constexpr (TM == 4 && ABL != 2) {
#pragma unroll
      for (int ii = 0; ii < BK; ii += 2) {
        step4<BM, BN, true>(acc, X, Y, a_addr, b_addr, ii + 1);
        if (ii + 2 < BK) step4<BM, BN, true>(acc, Y, X, a_addr, b_addr, ii + 2);
        else step4<BM, BN, false>(acc, Y, X, a_addr, b_addr, 0);
      }
    } else {
#pragma unroll
      for (int ii = 0; ii < BK; ii += 2) {
        if (ABL != 2) fetch(Y, a_addr, b_addr, ii + 1);
        mac_step<TM>(acc, X);
        if (ABL != 2) wait(Y);
        if (ABL != 2 && ii + 2 < BK) fetch(X, a_addr, b_addr, ii + 2);
        mac_step<TM>(acc, Y);
        if (ABL != 2 && ii + 2 < BK) wait(X);
      }
    }
    if (ABL == 3) store_stage(buf ^ 1, ((s + 1) & (kStages - 1)) * BK);
    if (ABL == 4) __syncthreads();
    if (ABL == 0) {
      if (RING == 3) {
        store_stage((s + 2) % 3, ((s + 2) & (kStages - 1)) * BK);
        __builtin_amdgcn_s_barrier();  // publishes the writes made one stage ago; no LDS wait
      } else {
        store_stage(buf ^ 1, ((s + 1) & (kStages - 1)) * BK);
        __syncthreads();
      }
    }
  }

#pragma unroll
  for (int r = 0; r < TM; ++r) {
    const unsigned row = m0 + ((r < 4) ? (ty * 4 + r) : (BM / 2 + ty * 4 + (r - 4)));
    if (row >= M) continue;
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0;
    if constexpr (SCALAR) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = f32x2{accs[r & 3][2 * c], accs[r & 3][2 * c + 1]};
    }
    float4 o;
    o.x = mul_rn(acc[r][0].x, tb.norm); o.y = mul_rn(acc[r][0].y, tb.norm);
    o.z = mul_rn(acc[r][1].x, tb.norm); o.w = mul_rn(acc[r][1].y, tb.norm);
    *reinterpret_cast<float4 *>(dst + tx * 4) = o;
    o.x = mul_rn(acc[r][2].x, tb.norm); o.y = mul_rn(acc[r][2].y, tb.norm);
    o.z = mul_rn(acc[r][3].x, tb.norm); o.w = mul_rn(acc[r][3].y, tb.norm);
    *reinterpret_cast<float4 *>(dst + BN / 2 + tx * 4) = o;
  }
}

template <int BM, int BN, int BK, int MINW, int ABL = 0, int TM = 8, bool SCALAR = false, int RING = 2,
          bool WLDS = false>
inline hipError_t launch_sched(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M,
                               float *coef, hipStream_t s) {
  using C = Cfg<BM, BN, BK, TM, 8, 2, MINW>;
  if (M == 0) return hipSuccess;
  const unsigned m_tiles = (M + BM - 1) / BM;
  hipLaunchKernelGGL((k_mdct_fwd_sched<BM, BN, BK, MINW, ABL, TM, SCALAR, RING, WLDS>), dim3(m_tiles * C::kNTiles),
                     dim3(C::kThreads), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), M, coef);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// LDS-DMA kernel (128x128 tile, 512 threads, 4x8 outputs per lane, 3-slot LDS ring).
// The table tile of stage s+2 is copied global -> LDS by `global_load_lds_dwordx4` while stages s
// and s+1 compute: two stages of latency budget, no VGPRs and no ds_write for the table.  The PCM
// tile (needs the window multiply) goes through registers one stage ahead, the window itself sits
// in LDS.  All vector-memory operations of the loop are inline asm so that ONE counted
// `s_waitcnt vmcnt(1)` per stage waits for the PCM loads and the table DMA of the NEXT stage while
// leaving the DMA of the stage after it in flight.  Same arithmetic, same order.
// ------------------------------------------------------------------------------------------
template <int MINW, int ABL = 0, int BM = 128, int CH = 0, int STAGGER = 0>
__global__ __launch_bounds__(4 * BM) __attribute__((amdgpu_waves_per_eu(MINW, MINW)))
void k_mdct_fwd_dma(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M, float *__restrict__ coef) {
  // CH = 0: one PCM dword per (row, i) and lane - any channel count.  CH = 1 / 2 / 4 / 8 (the
  // stream's channel count, which then divides the tile height): a stage's 16 samples x CH channels
  // of one frame are 64 * CH contiguous bytes, fetched by 4 * CH lanes with one dwordx4 each - 16x
  // fewer cache-line touches in the texture addresser than the per-row loader.
  // BM = 128: 512 threads, 2 workgroups per CU (56 KiB LDS each).  BM = 64: 256 threads, window
  // values by scalar loads instead of LDS (36 KiB), 4 workgroups per CU - each SIMD then holds one
  // wave of four DIFFERENT workgroups, whose barrier waits do not coincide.
  constexpr int BN = 128, BK = 16, TM = 4, RING = 3;
  constexpr int kThreads = 4 * BM, kWaves = kThreads / 64;
  constexpr int kAPer = 4, kAStride = 4;
  constexpr bool kWinLds = BM == 128;
  constexpr bool kSeg = CH != 0;  // segment loader
  static_assert(!kSeg || (BM == 128 && (CH == 1 || CH == 2 || CH == 4 || CH == 8)), "segment loader shapes");
  constexpr int kDma = (BK * BN * 4) / (kThreads * 16);  // table-DMA instructions per thread and stage
  static_assert(BM == 128 || BM == 64, "tile heights with a hand-written schedule");
  __shared__ __attribute__((aligned(16))) float As[RING][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[RING][BK * BN];
  __shared__ __attribute__((aligned(16))) float Ws[kWinLds ? kFrameI : 4];

  const int tid = threadIdx.x;
  const unsigned g = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-aware tile map
  const int n_tile = g % 8;
  const int m_tile = g / 8;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int tx = tid % 16, ty = tid / 16;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  // STAGGER (tuning): every other workgroup of an XCD starts STAGGER x 64 cycles late, so that the two
  // workgroups sharing a CU do not hit their per-stage barriers at the same moment
  if constexpr (STAGGER > 0) {
    if ((blockIdx.x >> 3) & 1) __builtin_amdgcn_s_sleep(STAGGER);
  }

  if constexpr (kWinLds)
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
  const int a_i_s = __builtin_amdgcn_readfirstlane(a_i);
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
  // table DMA: instruction d of wave w copies rows 2(d*kWaves + w), +1 of the stage's 16 x 128
  // tile (1 KiB, lane-linear)
  const float *b_src = tb.cos_t + n0 + static_cast<size_t>(2 * wave + (lane >> 5)) * kHopI + (lane & 31) * 4;

  float a_raw[kAPer];
  f32x4 a_seg = {0.f, 0.f, 0.f, 0.f};
  auto issue_a = [&](int i0) {  // asm: hipcc must not count these loads (see lds_fetch)
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
#pragma unroll
    for (int d = 0; d < kDma; ++d)
      __builtin_amdgcn_global_load_lds(b_src + static_cast<size_t>(i0 + 2 * d * kWaves) * kHopI,
                                       &Bs[slot][2 * (d * kWaves + wave) * BN], 16, 0, 0);
  };
  // window values of a stage: from LDS, or (BM = 64) four wave-uniform scalar loads.  hipcc would
  // use vector loads here (it cannot prove the table unclobbered across the asm blocks) and then
  // wait for vmcnt(0), DMA included, so the s_loads are written by hand; `pin_w` after an
  // lgkmcnt(0) wait is what makes their results visible to the compiler-scheduled consumers.
  float w_next[kAPer] = {0.f, 0.f, 0.f, 0.f}, w_far[kAPer] = {0.f, 0.f, 0.f, 0.f};
  auto load_w = [&](int i0) {
    if constexpr (!kWinLds) {
      const float *wp = tb.window + i0 + a_i_s;
      asm volatile(
          "s_load_dword %0, %4, 0x0\n\t"
          "s_load_dword %1, %4, 0x10\n\t"
          "s_load_dword %2, %4, 0x20\n\t"
          "s_load_dword %3, %4, 0x30"
          : "=&s"(w_far[0]), "=&s"(w_far[1]), "=&s"(w_far[2]), "=&s"(w_far[3])
          : "s"(wp)
          : "memory");
    }
  };
  auto pin_w = [&]() {  // call only behind an lgkmcnt(0) wait
    if constexpr (!kWinLds) {
      asm volatile("" : "+s"(w_far[0]), "+s"(w_far[1]), "+s"(w_far[2]), "+s"(w_far[3]));
#pragma unroll
      for (int j = 0; j < kAPer; ++j) w_next[j] = w_far[j];
    }
  };
  auto store_a = [&](int i0, int slot) {
    if constexpr (kSeg) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = seg_o + j;  // float e of the segment: sample i = e / CH of channel e % CH
        const int ii = e / CH;
        As[slot][ii * BM + seg_fl * CH + e % CH] = mul_rn(a_seg[j], Ws[i0 + ii]);  // :480
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < kAPer; ++j) {
      const int ii = a_i + kAStride * j;
      const float wv = kWinLds ? Ws[i0 + ii] : w_next[j];
      As[slot][ii * BM + a_r] = mul_rn(a_raw[j], wv);  // block[i] = slice[i]*window[i], :480
    }
  };
  auto wait_all_but_newest_dma = [&]() {
    if constexpr (kSeg)
      asm volatile("s_waitcnt vmcnt(1)" : "+v"(a_seg)::"memory");
    else if constexpr (kDma == 1)
      asm volatile("s_waitcnt vmcnt(1)" : "+v"(a_raw[0]), "+v"(a_raw[1]), "+v"(a_raw[2]), "+v"(a_raw[3])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(2)" : "+v"(a_raw[0]), "+v"(a_raw[1]), "+v"(a_raw[2]), "+v"(a_raw[3])::"memory");
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
  load_w(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+v"(a_raw[0]), "+v"(a_raw[1]), "+v"(a_raw[2]), "+v"(a_raw[3]), "+v"(a_seg)::"memory");
  pin_w();
  store_a(0, 0);
  issue_a(BK);
  issue_b(BK, 1);
  load_w(BK);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  pin_w();
  __syncthreads();

  const unsigned a_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&As[0][ty * 4]));
  const unsigned b_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&Bs[0][tx * 4]));

#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    const int slot = s % 3;
    // STAGGER < 0 (tuning): issue priority alternates between the two workgroups of a CU (blocks b and
    // b + 256) every -STAGGER stages, so that one of them runs ahead while the other fills its gaps
    if constexpr (STAGGER < 0) {
      if ((((s / (-STAGGER)) ^ static_cast<int>(blockIdx.x >> 8)) & 1) != 0) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    }
    // in flight on entry: PCM loads of stage s+1 (regs) and table DMA of stage s+1 (slot (s+1)%3)
    if (ABL == 0) issue_b(((s + 2) & (kStages - 1)) * BK, (s + 2) % 3);  // slot of stage s-1: free since the barrier
    const unsigned a_addr = a_lds0 + slot * (BK * BM * 4);
    const unsigned b_addr = b_lds0 + slot * (BK * BN * 4);
    Operands X, Y;
    if (ABL == 0 && s > 0) load_w(((s + 1) & (kStages - 1)) * BK);  // s == 0: loaded by the prologue
    lds_fetch4<BM, BN>(X, a_addr, b_addr, 0);
    lds_wait4(X);  // lgkmcnt(0): the scalar loads above are back as well
    if (ABL == 0 && s > 0) pin_w();
#pragma unroll
    for (int ii = 0; ii < BK; ii += 2) {
      step4<BM, BN, true>(acc, X, Y, a_addr, b_addr, ii + 1);
      if (ii + 2 < BK) step4<BM, BN, true>(acc, Y, X, a_addr, b_addr, ii + 2);
      else step4<BM, BN, false>(acc, Y, X, a_addr, b_addr, 0);
    }
    if (ABL == 0) {
      // everything but the youngest vector-memory ops (the DMA of stage s+2) has landed: the PCM
      // registers of stage s+1 and, older still, the table DMA of stage s+1
      wait_all_but_newest_dma();
      store_a(((s + 1) & (kStages - 1)) * BK, (s + 1) % 3);
      issue_a(((s + 2) & (kStages - 1)) * BK);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's ds_writes of stage s+1 have landed
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the wrap-around prefetches

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

template <int MINW, int ABL = 0, int BM = 128, int CH = 0, int STAGGER = 0>
inline hipError_t launch_dma(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M,
                             float *coef, hipStream_t s) {
  if (M == 0) return hipSuccess;
  if (CH != 0 && pcm.ch != static_cast<uint32_t>(CH)) return hipErrorInvalidValue;
  const unsigned m_tiles = (M + BM - 1) / BM;
  hipLaunchKernelGGL((k_mdct_fwd_dma<MINW, ABL, BM, CH, STAGGER>), dim3(m_tiles * 8), dim3(4 * BM), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), M, coef);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Matrix-pipe multiplier variant (tuning harness only).  `v_mfma_f32_32x32x1_2b_f32` with C = 0
// returns exactly the bits of v_mul_f32 (tools/mfma_probe.hip), so the products of one i-step of
// a 32-row x 64-column wave tile can come from ONE matrix instruction (two 32x32 outer products:
// the A operand is one x value per row, the B operand one table value per column) and only the
// separately rounded accumulation stays on the vector ALU: acc += P as v_pk_add_f32.  Same
// arithmetic, same order.  The f32 matrix pipe does not co-issue with the VALU, so
// this wins no issue slots; what it removes is operand traffic: 3 ds_read_b32 per 4096 MACs
// instead of 12 ds_read_b64, and the multiply's VGPR operand reads.
//   tile 128 x 128, 256 threads: wave w owns rows 64 (w / 2) .. +64 and columns 64 (w % 2) .. +64
// ------------------------------------------------------------------------------------------
typedef float f32x32 __attribute__((ext_vector_type(32)));

template <int BK, int MINW>
__global__ __launch_bounds__(256, MINW) void k_mdct_fwd_mx(DeviceTables tb, PcmView pcm, long long frame_begin,
                                                          unsigned M, float *__restrict__ coef) {
  constexpr int BM = 128, BN = 128, kThreads = 256;
  constexpr int kAPer = BM * BK / kThreads, kAStride = kThreads / BM;
  constexpr int kBPer = BK * BN / 4 / kThreads, kBRowsPer = kThreads / (BN / 4);
  __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

  const int tid = threadIdx.x;
  const unsigned g = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-aware tile map
  const int n_tile = g % 8, m_tile = g / 8;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int wave = tid >> 6, lane = tid & 63;
  const int r0 = (wave >> 1) * 64, c0 = (wave & 1) * 64;

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
  unsigned a_off = 0x80000000u;
  if (a_row < M) {
    const long long f = frame_begin + a_row / pcm.ch;
    const long long c = a_row % pcm.ch;
    const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * ch + c;
    a_off = static_cast<unsigned>((e_row - e_base + a_i * ch) * 4);
  }
  const unsigned a_step = static_cast<unsigned>(kAStride * ch * 4);
  const float *w_ptr = tb.window + a_i;
  const int b_r = tid / (BN / 4), b_c4 = tid % (BN / 4);
  const float *b_ptr = tb.cos_t + n0 + static_cast<size_t>(b_r) * kHopI + b_c4 * 4;

  float a_stage[kAPer];
  float4 b_stage[kBPer];
  auto load_stage = [&](int i0) {
    const unsigned off0 = a_off + static_cast<unsigned>(i0) * static_cast<unsigned>(ch * 4);
#pragma unroll
    for (int j = 0; j < kAPer; ++j) {
      const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, off0 + j * a_step, 0, 0));
      a_stage[j] = mul_rn(x, w_ptr[i0 + kAStride * j]);
    }
#pragma unroll
    for (int j = 0; j < kBPer; ++j)
      b_stage[j] = *reinterpret_cast<const float4 *>(b_ptr + static_cast<size_t>(i0 + kBRowsPer * j) * kHopI);
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int j = 0; j < kAPer; ++j) As[buf][(a_i + kAStride * j) * BM + a_r] = a_stage[j];
#pragma unroll
    for (int j = 0; j < kBPer; ++j)
      *reinterpret_cast<float4 *>(&Bs[buf][(b_r + kBRowsPer * j) * BN + b_c4 * 4]) = b_stage[j];
  };

  f32x32 acc0, acc1, zero;
#pragma unroll
  for (int j = 0; j < 32; ++j) acc0[j] = 0.0f, acc1[j] = 0.0f, zero[j] = 0.0f;

  load_stage(0);
  store_stage(0);
  __syncthreads();

  constexpr int kStages = kFrameI / BK;
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    const int buf = s & 1;
    load_stage(((s + 1) & (kStages - 1)) * BK);
    const float *Ab = As[buf] + r0 + (lane & 31);
    const float *Bb = Bs[buf] + c0 + lane;
#pragma unroll
    for (int ii = 0; ii < BK; ++ii) {
      const float a0 = Ab[ii * BM], a1 = Ab[ii * BM + 32], b = Bb[ii * BN];
      const f32x32 p0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a0, b, zero, 0, 0, 0);  // fl(a*b), one per output
      acc0 = acc0 + p0;                                                             // separately rounded add
      const f32x32 p1 = __builtin_amdgcn_mfma_f32_32x32x1f32(a1, b, zero, 0, 0, 0);
      acc1 = acc1 + p1;
    }
    store_stage(buf ^ 1);
    __syncthreads();
  }

  // vgpr j of block j / 16: column = lane % 32 + 32 (j / 16), row = 8 ((j % 16) / 4) + 4 (lane / 32) + j % 4
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const f32x32 &acc = half ? acc1 : acc0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int jj = j % 16;
      const unsigned row = m0 + r0 + half * 32 + 8 * (jj / 4) + 4 * (lane >> 5) + (jj % 4);
      if (row >= M) continue;
      coef[static_cast<size_t>(row) * kHopI + n0 + c0 + (lane & 31) + 32 * (j / 16)] = mul_rn(acc[j], tb.norm);
    }
  }
}

template <int BK, int MINW>
inline hipError_t launch_mx(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M, float *coef,
                            hipStream_t s) {
  if (M == 0) return hipSuccess;
  const unsigned m_tiles = (M + 127) / 128;
  hipLaunchKernelGGL((k_mdct_fwd_mx<BK, MINW>), dim3(m_tiles * 8), dim3(256), 0, s, t, pcm,
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

}  // namespace k1x
}  // namespace glc
