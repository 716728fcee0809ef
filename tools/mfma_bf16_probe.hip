// mfma_bf16_probe.hip - can the bf16 matrix pipe (which, unlike the f32 one, executes beside the vector ALU)
// produce fl(a * b) of two f32?  An f32 splits EXACTLY into three bf16 pieces (3 x 8 significand bits), so
// a * b is the sum of nine piece products, each exact in f32.  If v_mfma_f32_32x32x16_bf16 accumulated its
// K = 16 products exactly and rounded once, nine of its K slots would deliver the correctly rounded product
// with C = 0, and K1's multiplies could leave the vector ALU.  This probe feeds random
// normal operands (no subnormal pieces: moderate exponents) and counts how often the matrix result differs
// from v_mul_f32, for three orders of the nine terms in the K slots.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_bf16_probe.hip -o build/mfma_bf16_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf16_trunc(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
__device__ __forceinline__ __bf16 to_bf16_exact(float x) {  // x has <= 8 significant bits: the conversion is exact
  const unsigned short h = static_cast<unsigned short>(__float_as_uint(x) >> 16);
  __bf16 r;
  __builtin_memcpy(&r, &h, 2);
  return r;
}

// order: which (p, q) piece pair sits in K slot j (slots 0..7 in lane half 0, slot 8 = K index 8 in half 1)
__constant__ int kOrder[3][9][2] = {
    {{0, 0}, {0, 1}, {1, 0}, {0, 2}, {1, 1}, {2, 0}, {1, 2}, {2, 1}, {2, 2}},   // descending magnitude
    {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}},   // ascending magnitude
    {{0, 0}, {2, 2}, {0, 1}, {2, 1}, {1, 0}, {1, 2}, {0, 2}, {2, 0}, {1, 1}},   // mixed
};

__global__ void k_probe(const float *a, const float *b, int order, float *d_mfma, float *d_mul) {
  const int lane = threadIdx.x, half = lane >> 5, idx = lane & 31;
  float ap[3], bp[3];
  ap[0] = bf16_trunc(a[idx]);
  ap[1] = bf16_trunc(a[idx] - ap[0]);
  ap[2] = a[idx] - ap[0] - ap[1];
  bp[0] = bf16_trunc(b[idx]);
  bp[1] = bf16_trunc(b[idx] - bp[0]);
  bp[2] = b[idx] - bp[0] - bp[1];
  bf16x8 A, B;
  for (int j = 0; j < 8; ++j) {
    const int slot = half * 8 + j;  // K index of this lane's element j
    float av = 0.0f, bv = 0.0f;
    if (slot < 9) av = ap[kOrder[order][slot][0]], bv = bp[kOrder[order][slot][1]];
    A[j] = to_bf16_exact(av);
    B[j] = to_bf16_exact(bv);
  }
  f32x16 zero;
  for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
  const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, zero, 0, 0, 0);
  for (int j = 0; j < 16; ++j) {
    const int row = 8 * (j / 4) + 4 * half + j % 4, col = idx;
    d_mfma[row * 32 + col] = d[j];
    d_mul[row * 32 + col] = __fmul_rn(a[row], b[col]);
  }
}

int main() {
  uint32_t s = 2463534242u;
  auto rnd = [&] { s ^= s << 13, s ^= s >> 17, s ^= s << 5; return s; };
  float *da, *db, *dm, *dv;
  (void)hipMalloc(&da, 128), (void)hipMalloc(&db, 128), (void)hipMalloc(&dm, 4096), (void)hipMalloc(&dv, 4096);
  for (int order = 0; order < 3; ++order) {
    long long bad = 0, total = 0, off_by_more = 0;
    for (int round = 0; round < 400; ++round) {
      std::vector<float> a(32), b(32);
      for (int i = 0; i < 32; ++i) {
        const uint32_t ua = (rnd() & 0x807FFFFFu) | ((110u + rnd() % 30u) << 23), ub = (rnd() & 0x807FFFFFu) | ((110u + rnd() % 30u) << 23);
        std::memcpy(&a[i], &ua, 4), std::memcpy(&b[i], &ub, 4);
      }
      (void)hipMemcpy(da, a.data(), 128, hipMemcpyHostToDevice);
      (void)hipMemcpy(db, b.data(), 128, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, order, dm, dv);
      std::vector<uint32_t> m(1024), v(1024);
      (void)hipMemcpy(m.data(), dm, 4096, hipMemcpyDeviceToHost);
      (void)hipMemcpy(v.data(), dv, 4096, hipMemcpyDeviceToHost);
      for (int i = 0; i < 1024; ++i) {
        ++total;
        if (m[i] != v[i]) {
          ++bad;
          const long long d = static_cast<long long>(m[i] & 0x7FFFFFFFu) - static_cast<long long>(v[i] & 0x7FFFFFFFu);
          if (d > 1 || d < -1) ++off_by_more;
          if (bad <= 3) std::printf("  order %d: mfma %08x  v_mul %08x\n", order, m[i], v[i]);
        }
      }
    }
    std::printf("order %d: %lld of %lld products differ from v_mul_f32 (%.3f %%), %lld by more than 1 ulp\n", order, bad, total,
                100.0 * bad / total, off_by_more);
  }
  return 0;
}
