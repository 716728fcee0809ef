// k1_row_variant.hpp — TUNING ONLY: k_mdct_fwd_st's orientation with ONE row per lane, measured for launches
// below 4096 rows and NOT adopted (numbers: profiles/r03_k1_tune_row_variant.txt).  The library does not include
// this file; tools/k1_tune.hip does, after csrc/glc_mdct_fwd.hpp, whose namespace and helpers it continues.
//
// What the measurement says (MI355X, stereo, against the shipped kernels for the same row count):
//   172 rows   C=4: 0.055-0.058 ms   shipped 2 x 2 kernel: 0.053       no gain
//   600 rows   C=8: 0.083-0.095      shipped: 0.103                    -20 %
//   1024 rows  C=4: 0.123            shipped 2 x 4 kernel: 0.111       worse
//   1792 rows  C=8: 0.152            shipped: 0.187                    -19 %
//   3000 rows  C=8: 0.255            shipped 64 x 128 kernel: 0.309    -17 %
//   4094 rows  C=8: 0.304            k_mdct_fwd_st, 8 waves: 0.291     worse
// Why it does not deliver what its operand traffic promises: scalar loads only allow lgkmcnt(0), so there is
// never more than one group of operands in flight, and a group (32 table dwords: the SGPR file holds two) is
// 4 i-steps of C=8, 8 of C=4, 16 of C=2 - 32 packed operations, 128 cycles of arithmetic against a round trip of
// 250-500 cycles.  k_mdct_fwd_st has the same one group in flight but four waves per SIMD to hide it; a launch
// that puts one or two waves on a SIMD has not.  On top of that a group's time grows with the number of scalar
// loads in it when every load is a cache line of its own (table rows are 4 KiB apart): C=8 250 cycles per group
// of 4 loads, C=4 500 per 8, C=2 1670 per 16.  k_mdct_fwd_row8 checks that part: with a table copy laid out
// [column group][i][8] a group is two 64-byte loads - 1024 rows 0.118 ms instead of 0.147, 3000 rows 0.234 instead
// of 0.254, but 172 rows 0.059 (one round trip per 4 i-steps stays).  Not worth 8 MiB more per context.
#pragma once
#include "glc_mdct_fwd.hpp"

namespace glc {
namespace k1 {
// ------------------------------------------------------------------------------------------
// Row kernel (launches below 4096 rows): k_mdct_fwd_st's orientation with ONE row per lane.  A wave's
// lanes hold 64 rows and share C columns (2, 4 or 8): the table values of an i-step are C wave-uniform
// dwords by one scalar load, the lane's windowed sample comes from LDS - one ds_read_b128 per FOUR
// i-steps (the samples' tile is row-major: As[row][i]).  What a short launch lacks is work per SIMD, and
// what it then costs is (a) one wave's chain of 2048 dependent i-steps and (b) operand bandwidth per
// step: with both operands from LDS (k_mdct_fwd_small below: 2 x 2 outputs per lane) every wave-step
// pulls 1 KiB through the CU's 128-byte LDS port, 32 cycles for the four waves of a CU against 16 of
// arithmetic.  Here it is 256 bytes per wave-step, and the lane tile can be as narrow as the launch
// needs to put a wave on every SIMD.  256-thread workgroups (a wave per SIMD), 64 rows x 4 C columns.
// Same products, same ascending-i adds, one accumulator per output.
// Scalar loads return out of order: every wait is lgkmcnt(0), operands are fetched G = 32 / C i-steps at
// a time, a group ahead (two register sets).
// ------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int C> struct SrTab;  // the C table values of one i-step
template <> struct SrTab<2> { typedef u32x2 type; };
template <> struct SrTab<4> { typedef u32x4 type; };
template <> struct SrTab<8> { typedef u32x8 type; };

template <int C>
struct SrOps {  // operands of G = 32 / C i-steps: this lane's samples (4 steps per register quad), this wave's columns
  static constexpr int G = 32 / C;
  f32x4 a[G / 4];
  typename SrTab<C>::type b[G];
};

template <int C, int II>
__device__ __forceinline__ void sr_fetch_b(typename SrTab<C>::type &b, const unsigned *brow) {
  if constexpr (C == 2) asm volatile("s_load_dwordx2 %0, %1, %c2" : "=&s"(b) : "s"(brow), "i"(II * kHopI * 4) : "memory");
  if constexpr (C == 4) asm volatile("s_load_dwordx4 %0, %1, %c2" : "=&s"(b) : "s"(brow), "i"(II * kHopI * 4) : "memory");
  if constexpr (C == 8) asm volatile("s_load_dwordx8 %0, %1, %c2" : "=&s"(b) : "s"(brow), "i"(II * kHopI * 4) : "memory");
}
template <int C, int II, int K>
__device__ __forceinline__ void sr_fetch_bs(SrOps<C> &o, const unsigned *brow) {  // steps II + K .. II + G - 1
  if constexpr (K < SrOps<C>::G) {
    sr_fetch_b<C, II + K>(o.b[K], brow);
    sr_fetch_bs<C, II, K + 1>(o, brow);
  }
}
template <int C, int II>
__device__ __forceinline__ void sr_fetch(SrOps<C> &o, unsigned a_addr, const unsigned *brow) {
  sr_fetch_bs<C, II, 0>(o, brow);
  constexpr int Q = SrOps<C>::G / 4;
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(o.a[0]) : "v"(a_addr), "i"(II * 4) : "memory");
  if constexpr (Q > 1) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(o.a[1]) : "v"(a_addr), "i"(II * 4 + 16) : "memory");
  if constexpr (Q > 2) {
    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(o.a[2]) : "v"(a_addr), "i"(II * 4 + 32) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(o.a[3]) : "v"(a_addr), "i"(II * 4 + 48) : "memory");
  }
}
// scalar and vector results are tied in separate statements (see st_wait)
template <int C>
__device__ __forceinline__ void sr_wait(SrOps<C> &o) {
  if constexpr (C == 8)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(o.b[0]), "+s"(o.b[1]), "+s"(o.b[2]), "+s"(o.b[3])::"memory");
  if constexpr (C == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(o.b[0]), "+s"(o.b[1]), "+s"(o.b[2]), "+s"(o.b[3]), "+s"(o.b[4]), "+s"(o.b[5]), "+s"(o.b[6]), "+s"(o.b[7])::"memory");
  if constexpr (C == 2)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(o.b[0]), "+s"(o.b[1]), "+s"(o.b[2]), "+s"(o.b[3]), "+s"(o.b[4]), "+s"(o.b[5]), "+s"(o.b[6]), "+s"(o.b[7]),
                   "+s"(o.b[8]), "+s"(o.b[9]), "+s"(o.b[10]), "+s"(o.b[11]), "+s"(o.b[12]), "+s"(o.b[13]), "+s"(o.b[14]), "+s"(o.b[15])::"memory");
  constexpr int Q = SrOps<C>::G / 4;
  asm volatile("" : "+v"(o.a[0])::"memory");
  if constexpr (Q > 1) asm volatile("" : "+v"(o.a[1])::"memory");
  if constexpr (Q > 2) asm volatile("" : "+v"(o.a[2]), "+v"(o.a[3])::"memory");
}

// two i-steps of one row: acc_j += a.lo * b0_j, then acc_j += a.hi * b1_j (j = column pair), in that order
__device__ __forceinline__ void sr_mac2(f32x2 (&c)[4], f32x2 a, u32x8 b0, u32x8 b1) {
  f32x2 t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "v_pk_mul_f32 %4, %12, %13 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %5, %12, %14 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %6, %12, %15 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %7, %12, %16 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %8, %12, %17 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %9, %12, %18 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %10, %12, %19 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %11, %12, %20 op_sel:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, %4\n\t"
      "v_pk_add_f32 %1, %1, %5\n\t"
      "v_pk_add_f32 %2, %2, %6\n\t"
      "v_pk_add_f32 %3, %3, %7\n\t"
      "v_pk_add_f32 %0, %0, %8\n\t"
      "v_pk_add_f32 %1, %1, %9\n\t"
      "v_pk_add_f32 %2, %2, %10\n\t"
      "v_pk_add_f32 %3, %3, %11"
      : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5),
        "=&v"(t6), "=&v"(t7)
      : "v"(a), "s"(b0.s01), "s"(b0.s23), "s"(b0.s45), "s"(b0.s67), "s"(b1.s01), "s"(b1.s23), "s"(b1.s45), "s"(b1.s67));
}
__device__ __forceinline__ void sr_mac2(f32x2 (&c)[2], f32x2 a, u32x4 b0, u32x4 b1) {
  f32x2 t0, t1, t2, t3;
  asm volatile(
      "v_pk_mul_f32 %2, %6, %7 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %3, %6, %8 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %4, %6, %9 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %5, %6, %10 op_sel:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, %2\n\t"
      "v_pk_add_f32 %1, %1, %3\n\t"
      "v_pk_add_f32 %0, %0, %4\n\t"
      "v_pk_add_f32 %1, %1, %5"
      : "+v"(c[0]), "+v"(c[1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(a), "s"(b0.xy), "s"(b0.zw), "s"(b1.xy), "s"(b1.zw));
}
// C = 2: four i-steps at once (the products first, then the chain of four adds)
__device__ __forceinline__ void sr_mac4(f32x2 (&c)[1], f32x4 a, u32x2 b0, u32x2 b1, u32x2 b2, u32x2 b3) {
  f32x2 t0, t1, t2, t3;
  asm volatile(
      "v_pk_mul_f32 %1, %5, %7 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %2, %5, %8 op_sel:[1,0]\n\t"
      "v_pk_mul_f32 %3, %6, %9 op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %4, %6, %10 op_sel:[1,0]\n\t"
      "v_pk_add_f32 %0, %0, %1\n\t"
      "v_pk_add_f32 %0, %0, %2\n\t"
      "v_pk_add_f32 %0, %0, %3\n\t"
      "v_pk_add_f32 %0, %0, %4"
      : "+v"(c[0]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(a.xy), "v"(a.zw), "s"(b0), "s"(b1), "s"(b2), "s"(b3));
}
template <int C>
__device__ __forceinline__ void sr_mac(f32x2 (&acc)[C / 2], const SrOps<C> &o) {
#pragma unroll
  for (int q = 0; q < SrOps<C>::G / 4; ++q) {
    if constexpr (C == 2) {
      sr_mac4(acc, o.a[q], o.b[4 * q], o.b[4 * q + 1], o.b[4 * q + 2], o.b[4 * q + 3]);
    } else {
      sr_mac2(acc, o.a[q].xy, o.b[4 * q], o.b[4 * q + 1]);
      sr_mac2(acc, o.a[q].zw, o.b[4 * q + 2], o.b[4 * q + 3]);
    }
  }
}

template <int C, int CH = 0, int BK = 32>
__global__ __launch_bounds__(256) void k_mdct_fwd_row(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M,
                                                       float *__restrict__ coef) {
  // CH as in k_mdct_fwd_dma: 0 = one PCM dword per (row, i) and lane; 1 / 2 / 4 / 8 = the stream's channel
  // count, BK samples x CH channels of a frame fetched as BK / 4 * CH dwordx4.  The window value of a sample
  // is loaded beside it (no window copy in LDS: a short launch would pay for filling it).
  constexpr int BM = 64, BN = 4 * C, RING = 3, G = SrOps<C>::G;
  constexpr int kAS = BK + 4;  // floats between rows of the tile: ds_read_b128 of 16 consecutive lanes hits 64 different banks
  constexpr int kThreads = 256;
  constexpr int kNTiles = kHopI / BN;
  constexpr bool kSeg = CH != 0;
  static_assert(C == 2 || C == 4 || C == 8, "columns per wave");
  static_assert(!kSeg || CH == 1 || CH == 2 || CH == 4 || CH == 8, "segment loader shapes");
  static_assert((BK == 16 || BK == 32 || BK == 64) && BK % (2 * G) == 0, "stage depth");
  constexpr int kIGroups = kThreads / BM;          // per-row loader: i = a_i + 4 j
  constexpr int kAPer = BK / kIGroups;             //   dwords per lane and stage (4, 8 or 16)
  constexpr int kPieces = BM * BK / 4 / kThreads;  // segment loader: dwordx4 per lane and stage (1, 2 or 4)
  __shared__ __attribute__((aligned(16))) float As[RING][BM * kAS];

  const int tid = threadIdx.x;
  const int n_tile = blockIdx.x % kNTiles;
  const int m_tile = blockIdx.x / kNTiles;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;

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

  // per-row loader: lane -> (row a_r of the tile, i = a_i + 4 j)
  const int a_r = tid % BM;
  const int a_i = tid / BM;
  unsigned a_off[kSeg ? kPieces : 1];
#pragma unroll
  for (int p = 0; p < (kSeg ? kPieces : 1); ++p) a_off[p] = 0x80000000u;
  const unsigned a_step = static_cast<unsigned>(kIGroups * ch * 4);
  const unsigned i_bytes = static_cast<unsigned>(ch * 4);
  // segment loader: (lane, p) -> (frame seg_fl + p * kSegStep of the tile, 4 consecutive floats of its BK x CH segment)
  constexpr int kSegCh = kSeg ? CH : 1;
  constexpr int kSegLanes = BK / 4 * kSegCh;  // lanes per frame segment
  constexpr int kSegStep = kThreads / kSegLanes;
  const int seg_fl = tid / kSegLanes;
  const int seg_o = (tid % kSegLanes) * 4;
  if constexpr (kSeg) {
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const unsigned row0 = m0 + (seg_fl + p * kSegStep) * CH;
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

  // staged registers: raw samples and their window values (segment loader: the 4 floats of a piece are
  // samples seg_o / CH .. of 4 / CH .. 1 consecutive i; per-row loader: one sample each)
  constexpr int kWin = kSeg ? (CH == 1 ? 4 : CH == 2 ? 2 : 1) : 1;
  f32x4 a_seg[kSeg ? kPieces : 1];
  float w_seg[kWin];
  float a_raw[kSeg ? 1 : kAPer], a_win[kSeg ? 1 : kAPer];
  auto issue_a = [&](int i0) {
    if constexpr (kSeg) {
#pragma unroll
      for (int p = 0; p < kPieces; ++p) {
        const unsigned o = a_off[p] + static_cast<unsigned>(i0) * i_bytes;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(a_seg[p]) : "v"(o), "s"(a_rsrc) : "memory");
      }
      const float *w = tb.window + i0 + seg_o / CH;
#pragma unroll
      for (int k = 0; k < kWin; ++k) w_seg[k] = w[k];
    } else {
      const unsigned o = a_off[0] + static_cast<unsigned>(i0) * i_bytes;
#pragma unroll
      for (int j = 0; j < kAPer; ++j) {
        a_raw[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a_rsrc, o + j * a_step, 0, 0));
        a_win[j] = tb.window[i0 + a_i + kIGroups * j];
      }
    }
  };
  auto store_a = [&](int slot) {
    if constexpr (kSeg) {
#pragma unroll
      for (int p = 0; p < kPieces; ++p) {
        f32x4 v = a_seg[p];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory");  // (stage-old loads: see the hand-off)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = seg_o + j;  // float e of the segment: sample i = e / CH of channel e % CH
          const int ii = e / CH;
          As[slot][((seg_fl + p * kSegStep) * CH + e % CH) * kAS + ii] = mul_rn(v[j], w_seg[j / CH < kWin ? j / CH : 0]);  // :480
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < kAPer; ++j) {
        float r = a_raw[j], w = a_win[j];
        asm volatile("" : "+v"(r), "+v"(w));  // first use of the staged registers stays where the hand-off is
        As[slot][a_r * kAS + a_i + kIGroups * j] = mul_rn(r, w);  // block[i] = slice[i]*window[i], :480
      }
    }
  };

  f32x2 acc[C / 2];
#pragma unroll
  for (int c = 0; c < C / 2; ++c) acc[c] = f32x2{0.0f, 0.0f};

  constexpr int kStages = kFrameI / BK;
  // prologue: stage 0 complete in slot 0, PCM of stage 1 in registers
  issue_a(0);
  store_a(0);
  issue_a(BK);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  const unsigned a_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&As[0][lane * kAS]));
  // this wave's C columns of table row 0 (wave-uniform: the scalar loads' base)
  const unsigned *b_base = reinterpret_cast<const unsigned *>(tb.cos_t) + n0 + C * wave;

  // Stage hand-off in the middle of a stage (k_mdct_fwd_st's protocol): the samples of stage s+1 are published
  // by a barrier after the first BK / 2 i-steps, the stage's last fetch takes the first operands of stage
  // s+1, the PCM loads of stage s+2 are issued behind the barrier.  Stage s reads slot s % 3.
  SrOps<C> X, Y;
  sr_fetch<C, 0>(X, a_lds0, b_base);
  sr_wait(X);
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    const int slot = s % 3, nslot = (s + 1) % 3;
    const unsigned a_addr = a_lds0 + slot * (BM * kAS * 4);
    const unsigned a_next = a_lds0 + nslot * (BM * kAS * 4);
    const unsigned *brow = b_base + static_cast<size_t>(s) * (BK * kHopI);
    const unsigned *brow_next = b_base + static_cast<size_t>((s + 1) & (kStages - 1)) * (BK * kHopI);
#define GLC_SR_GROUP(CUR, NXT, II)                                                  \
  do {                                                                              \
    if constexpr ((II) + G < BK) sr_fetch<C, ((II) + G) % BK>(NXT, a_addr, brow);    \
    else sr_fetch<C, 0>(NXT, a_next, brow_next);                                    \
    sr_mac<C>(acc, CUR);                                                            \
    if constexpr ((II) + G == BK / 2) {                                             \
      store_a(nslot);                                                               \
      sr_wait(NXT);                                                                 \
      __builtin_amdgcn_s_barrier();                                                 \
      issue_a(((s + 2) & (kStages - 1)) * BK);                                      \
    } else {                                                                        \
      sr_wait(NXT);                                                                 \
    }                                                                               \
  } while (0)
    GLC_SR_GROUP(X, Y, 0);
    GLC_SR_GROUP(Y, X, G);
    if constexpr (BK / G > 2) {
      GLC_SR_GROUP(X, Y, 2 * G);
      GLC_SR_GROUP(Y, X, 3 * G);
    }
    if constexpr (BK / G > 4) {
      GLC_SR_GROUP(X, Y, 4 * G);
      GLC_SR_GROUP(Y, X, 5 * G);
      GLC_SR_GROUP(X, Y, 6 * G);
      GLC_SR_GROUP(Y, X, 7 * G);
    }
    if constexpr (BK / G > 8) {
      GLC_SR_GROUP(X, Y, 8 * G);
      GLC_SR_GROUP(Y, X, 9 * G);
      GLC_SR_GROUP(X, Y, 10 * G);
      GLC_SR_GROUP(Y, X, 11 * G);
      GLC_SR_GROUP(X, Y, 12 * G);
      GLC_SR_GROUP(Y, X, 13 * G);
      GLC_SR_GROUP(X, Y, 14 * G);
      GLC_SR_GROUP(Y, X, 15 * G);
    }
#undef GLC_SR_GROUP
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the wrap-around prefetch

  // epilogue: out[k] = s * norm, :372
  const unsigned row = m0 + lane;
  if (row < M) {
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0 + C * wave;
#pragma unroll
    for (int c = 0; c < C / 2; ++c) {
      float2 o;
      o.x = mul_rn(acc[c].x, tb.norm); o.y = mul_rn(acc[c].y, tb.norm);
      *reinterpret_cast<float2 *>(dst + 2 * c) = o;
    }
  }
}

// ---- the same kernel for C = 8 reading a table copy laid out [column group of 8][i][8] (tb.cos is borrowed for
// its address by the harness): the 8 values of two consecutive i-steps are ONE 64-byte scalar load.
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
struct SrOps8 {  // 4 i-steps
  f32x4 a;
  u32x16 b01, b23;
};
template <int II>
__device__ __forceinline__ void sr8_fetch(SrOps8 &o, unsigned a_addr, const unsigned *bgrp) {
  asm volatile("s_load_dwordx16 %0, %2, %c3\n\ts_load_dwordx16 %1, %2, %c4"
               : "=&s"(o.b01), "=&s"(o.b23)
               : "s"(bgrp), "i"(II * 32), "i"(II * 32 + 64)
               : "memory");
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(o.a) : "v"(a_addr), "i"(II * 4) : "memory");
}
__device__ __forceinline__ void sr8_wait(SrOps8 &o) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(o.b01), "+s"(o.b23)::"memory");
  asm volatile("" : "+v"(o.a)::"memory");
}
__device__ __forceinline__ void sr8_mac(f32x2 (&acc)[4], const SrOps8 &o) {
  sr_mac2(acc, o.a.xy, o.b01.lo, o.b01.hi);
  sr_mac2(acc, o.a.zw, o.b23.lo, o.b23.hi);
}

template <int CH = 0, int BK = 32>
__global__ __launch_bounds__(256) void k_mdct_fwd_row8(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M,
                                                        float *__restrict__ coef) {
  constexpr int C = 8, BM = 64, BN = 4 * C, RING = 3, G = 4;
  constexpr int kAS = BK + 4;
  constexpr int kThreads = 256;
  constexpr int kNTiles = kHopI / BN;
  static_assert(CH == 1 || CH == 2 || CH == 4 || CH == 8, "segment loader shapes (tuning variant: no per-row loader)");
  constexpr int kPieces = BM * BK / 4 / kThreads;
  __shared__ __attribute__((aligned(16))) float As[RING][BM * kAS];
  const int tid = threadIdx.x;
  const int n_tile = blockIdx.x % kNTiles;
  const int m_tile = blockIdx.x / kNTiles;
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
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
  unsigned a_off[kPieces];
  const unsigned i_bytes = static_cast<unsigned>(ch * 4);
  constexpr int kSegLanes = BK / 4 * CH;
  constexpr int kSegStep = kThreads / kSegLanes;
  const int seg_fl = tid / kSegLanes;
  const int seg_o = (tid % kSegLanes) * 4;
#pragma unroll
  for (int p = 0; p < kPieces; ++p) {
    a_off[p] = 0x80000000u;
    const unsigned row0 = m0 + (seg_fl + p * kSegStep) * CH;
    if (row0 < M) {
      const long long f = frame_begin + row0 / CH;
      const long long e_row = (f * kHopI - kHopI / 2 - static_cast<long long>(pcm.t0)) * CH;
      a_off[p] = static_cast<unsigned>((e_row - e_base + seg_o) * 4);
    }
  }
  constexpr int kWin = CH == 1 ? 4 : CH == 2 ? 2 : 1;
  f32x4 a_seg[kPieces];
  float w_seg[kWin];
  auto issue_a = [&](int i0) {
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      const unsigned o = a_off[p] + static_cast<unsigned>(i0) * i_bytes;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(a_seg[p]) : "v"(o), "s"(a_rsrc) : "memory");
    }
    const float *w = tb.window + i0 + seg_o / CH;
#pragma unroll
    for (int k = 0; k < kWin; ++k) w_seg[k] = w[k];
  };
  auto store_a = [&](int slot) {
#pragma unroll
    for (int p = 0; p < kPieces; ++p) {
      f32x4 v = a_seg[p];
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = seg_o + j;
        const int ii = e / CH;
        As[slot][((seg_fl + p * kSegStep) * CH + e % CH) * kAS + ii] = mul_rn(v[j], w_seg[j / CH < kWin ? j / CH : 0]);
      }
    }
  };
  f32x2 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x2{0.0f, 0.0f};
  constexpr int kStages = kFrameI / BK;
  issue_a(0);
  store_a(0);
  issue_a(BK);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned a_lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&As[0][lane * kAS]));
  // this wave's column group in the [group][i][8] copy (borrowed field: tb.cos)
  const unsigned *b_base = reinterpret_cast<const unsigned *>(tb.cos) + static_cast<size_t>(n0 / 8 + wave) * (kFrameI * 8);
  SrOps8 X, Y;
  sr8_fetch<0>(X, a_lds0, b_base);
  sr8_wait(X);
#pragma unroll 1
  for (int s = 0; s < kStages; ++s) {
    const int slot = s % 3, nslot = (s + 1) % 3;
    const unsigned a_addr = a_lds0 + slot * (BM * kAS * 4);
    const unsigned a_next = a_lds0 + nslot * (BM * kAS * 4);
    const unsigned *brow = b_base + static_cast<size_t>(s) * (BK * 8);
    const unsigned *brow_next = b_base + static_cast<size_t>((s + 1) & (kStages - 1)) * (BK * 8);
#define GLC_SR8_GROUP(CUR, NXT, II)                                             \
  do {                                                                          \
    if constexpr ((II) + G < BK) sr8_fetch<((II) + G) % BK>(NXT, a_addr, brow);  \
    else sr8_fetch<0>(NXT, a_next, brow_next);                                  \
    sr8_mac(acc, CUR);                                                          \
    if constexpr ((II) + G == BK / 2) {                                         \
      store_a(nslot);                                                           \
      sr8_wait(NXT);                                                            \
      __builtin_amdgcn_s_barrier();                                             \
      issue_a(((s + 2) & (kStages - 1)) * BK);                                  \
    } else {                                                                    \
      sr8_wait(NXT);                                                            \
    }                                                                           \
  } while (0)
    GLC_SR8_GROUP(X, Y, 0);
    GLC_SR8_GROUP(Y, X, 4);
    GLC_SR8_GROUP(X, Y, 8);
    GLC_SR8_GROUP(Y, X, 12);
    if constexpr (BK > 16) {
      GLC_SR8_GROUP(X, Y, 16);
      GLC_SR8_GROUP(Y, X, 20);
      GLC_SR8_GROUP(X, Y, 24);
      GLC_SR8_GROUP(Y, X, 28);
    }
#undef GLC_SR8_GROUP
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned row = m0 + lane;
  if (row < M) {
    float *dst = coef + static_cast<size_t>(row) * kHopI + n0 + C * wave;
#pragma unroll
    for (int c = 0; c < C / 2; ++c) {
      float2 o;
      o.x = mul_rn(acc[c].x, tb.norm); o.y = mul_rn(acc[c].y, tb.norm);
      *reinterpret_cast<float2 *>(dst + 2 * c) = o;
    }
  }
}
template <int CH = 0, int BK = 32>
inline hipError_t launch_row8(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M, float *coef,
                              hipStream_t s) {
  if (M == 0) return hipSuccess;
  if (pcm.ch != static_cast<uint32_t>(CH)) return hipErrorInvalidValue;
  const unsigned m_tiles = (M + 63) / 64;
  hipLaunchKernelGGL((k_mdct_fwd_row8<CH, BK>), dim3(m_tiles * (kHopI / 32)), dim3(256), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), M, coef);
  return hipGetLastError();
}

template <int C, int CH = 0, int BK = 32>
inline hipError_t launch_row(const DeviceTables &t, const PcmView &pcm, uint64_t frame_begin, uint32_t M, float *coef,
                             hipStream_t s) {
  if (M == 0) return hipSuccess;
  if (CH != 0 && pcm.ch != static_cast<uint32_t>(CH)) return hipErrorInvalidValue;
  const unsigned m_tiles = (M + 63) / 64;
  hipLaunchKernelGGL((k_mdct_fwd_row<C, CH, BK>), dim3(m_tiles * (kHopI / (4 * C))), dim3(256), 0, s, t, pcm,
                     static_cast<long long>(frame_begin), M, coef);
  return hipGetLastError();
}


}  // namespace k1
}  // namespace glc
