// mfma_probe.hip - is v_mfma_f32_32x32x1_2b_f32 with C = 0 the same function as v_mul_f32?
// One wave computes two 32x32 outer products of edge-case operands (normals, products that fall
// into the subnormal range, subnormal operands, zeros of both signs, infinities, NaNs with
// payloads) and compares every element with __fmul_rn bit for bit.  Needed before the matrix pipe
// could be considered as the multiplier of the exact-order transform.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_probe.hip -o build/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x32 __attribute__((ext_vector_type(32)));

__global__ void k_probe(const float *a, const float *b, float *d_mfma, float *d_mul) {
  const int lane = threadIdx.x;
  f32x32 zero;
  for (int i = 0; i < 32; ++i) zero[i] = 0.0f;
  const f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a[lane], b[lane], zero, 0, 0, 0);
  // layout: vgpr j of block blk = j / 16: col = lane % 32, row = 8 * ((j % 16) / 4) + 4 * (lane / 32) + j % 4
  for (int j = 0; j < 32; ++j) {
    const int blk = j / 16, jj = j % 16;
    const int row = 8 * (jj / 4) + 4 * (lane / 32) + jj % 4, col = lane % 32;
    const int idx = (blk * 32 + row) * 32 + col;
    d_mfma[idx] = d[j];
    d_mul[idx] = __fmul_rn(a[blk * 32 + row], b[blk * 32 + col]);
  }
}

static float bits(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

int main() {
  std::vector<float> a(64), b(64);
  uint32_t s = 12345;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return s; };
  const float specials[] = {0.0f, -0.0f, bits(0x7F800000u), bits(0xFF800000u), bits(0x7FC00000u), bits(0xFFC12345u),
                            bits(0x7F812345u), bits(0x00000001u), bits(0x807FFFFFu), bits(0x00800000u), 1.0f, -1.0f,
                            bits(0x7F7FFFFFu), bits(0x3F800001u), 1e-20f, -3e-25f};
  int total_bad = 0;
  for (int round = 0; round < 64; ++round) {
    for (int i = 0; i < 64; ++i) {
      const int kind = (round + i) % 4;
      if (kind == 0) a[i] = bits((rnd() & 0x007FFFFFu) | ((100u + rnd() % 60u) << 23) | (rnd() & 0x80000000u));  // normal
      else if (kind == 1) a[i] = bits((rnd() & 0x007FFFFFu) | ((40u + rnd() % 40u) << 23) | (rnd() & 0x80000000u));  // small
      else if (kind == 2) a[i] = bits(rnd() & 0x807FFFFFu);                                                       // subnormal
      else a[i] = specials[rnd() % 16];
      const int kb = (round * 3 + i) % 4;
      if (kb == 0) b[i] = bits((rnd() & 0x007FFFFFu) | ((100u + rnd() % 60u) << 23) | (rnd() & 0x80000000u));
      else if (kb == 1) b[i] = bits((rnd() & 0x007FFFFFu) | ((30u + rnd() % 60u) << 23) | (rnd() & 0x80000000u));
      else if (kb == 2) b[i] = bits(rnd() & 0x807FFFFFu);
      else b[i] = specials[rnd() % 16];
    }
    float *da, *db, *dm, *dv;
    (void)hipMalloc(&da, 256), (void)hipMalloc(&db, 256), (void)hipMalloc(&dm, 2048 * 4), (void)hipMalloc(&dv, 2048 * 4);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dm, dv);
    std::vector<uint32_t> m(2048), v(2048);
    hipMemcpy(m.data(), dm, 2048 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(v.data(), dv, 2048 * 4, hipMemcpyDeviceToHost);
    int bad = 0, bad_nan = 0, bad_sub = 0, bad_other = 0;
    for (int i = 0; i < 2048; ++i)
      if (m[i] != v[i]) {
        ++bad;
        const bool nan = (v[i] & 0x7FFFFFFFu) > 0x7F800000u;
        const bool sub = (v[i] & 0x7F800000u) == 0 || (m[i] & 0x7F800000u) == 0;
        if (nan) ++bad_nan;
        else if (sub) ++bad_sub;
        else {
          if (bad_other < 3 && total_bad < 20)
            std::printf("  other: mfma %08x mul %08x\n", m[i], v[i]);
          ++bad_other;
        }
      }
    if (bad) std::printf("round %2d: %4d differ (NaN results %d, subnormal/zero results %d, other %d)\n", round, bad, bad_nan, bad_sub, bad_other);
    total_bad += bad;
    hipFree(da), hipFree(db), hipFree(dm), hipFree(dv);
  }
  std::printf("mfma_probe: %d of %d products differ from v_mul_f32\n", total_bad, 64 * 2048);
  return 0;
}
