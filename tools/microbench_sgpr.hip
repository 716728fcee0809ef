// microbench_sgpr.hip - would K1 gain from taking its A operand (the windowed sample, the same for every
// lane of a row group) from SGPRs instead of VGPRs?  (VERDICT r2 #4.)  The bare instruction stream of an
// i-step - v_pk_mul_f32 with op_sel broadcast, v_pk_add_f32, dependent ops 8 apart, operands CHANGING from
// step to step (four rotating sets of random values, so the multipliers toggle like in the real kernel) -
// with the A pair in VGPRs and in SGPRs, for the 4 x 8 lane tile at 4 waves per SIMD (K1 at config 2) and the
// 8 x 8 tile at 2.  No memory traffic at all: this is the ceiling each form could reach, and the clock the
// chip holds under it (a sleeping probe wave: s_memtime against s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench_sgpr.hip -o build/microbench_sgpr
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e = (x);                                                          \
    if (e != hipSuccess) {                                                       \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

__global__ void k_clock_probe(unsigned long long ticks_100mhz, unsigned long long *out) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks_100mhz) {
    __builtin_amdgcn_s_sleep(32);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0, out[1] = r1 - r0;
}

// rows (r, r+1) x 8 columns from one A pair: 8 multiplies, 8 adds (the mac2rows block of K1 / D1)
#define MAC2ROWS(C0, C1, A, ACON, B0, B1, B2, B3)                                                      \
  do {                                                                                                 \
    f32x2 t0, t1, t2, t3, t4, t5, t6, t7;                                                               \
    asm volatile(                                                                                      \
        "v_pk_mul_f32 %8, %16, %17 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %9, %16, %18 op_sel_hi:[0,1]\n\t"   \
        "v_pk_mul_f32 %10, %16, %19 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %11, %16, %20 op_sel_hi:[0,1]\n\t" \
        "v_pk_mul_f32 %12, %16, %17 op_sel:[1,0]\n\tv_pk_mul_f32 %13, %16, %18 op_sel:[1,0]\n\t"       \
        "v_pk_mul_f32 %14, %16, %19 op_sel:[1,0]\n\tv_pk_mul_f32 %15, %16, %20 op_sel:[1,0]\n\t"       \
        "v_pk_add_f32 %0, %0, %8\n\tv_pk_add_f32 %1, %1, %9\n\tv_pk_add_f32 %2, %2, %10\n\t"           \
        "v_pk_add_f32 %3, %3, %11\n\tv_pk_add_f32 %4, %4, %12\n\tv_pk_add_f32 %5, %5, %13\n\t"         \
        "v_pk_add_f32 %6, %6, %14\n\tv_pk_add_f32 %7, %7, %15"                                         \
        : "+v"(C0[0]), "+v"(C0[1]), "+v"(C0[2]), "+v"(C0[3]), "+v"(C1[0]), "+v"(C1[1]), "+v"(C1[2]), "+v"(C1[3]),  \
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)        \
        : ACON(A), "v"(B0), "v"(B1), "v"(B2), "v"(B3));                                                \
  } while (0)

// ROWS = 4 (4 x 8 lane tile) or 8 (8 x 8); SGPR: the A pairs live in SGPRs
template <int ROWS, bool SGPR>
__global__ __launch_bounds__(256) void k_stream(float *out, const float *in, int iters) {
  constexpr int NP = ROWS / 2;  // A pairs per step
  f32x2 acc[ROWS][4];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = f32x2{0.f, 0.f};
  // four rotating operand sets (a step of the real kernel never sees the operands of the step before)
  f32x2 b[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) b[s][j] = f32x2{in[(threadIdx.x * 7 + s * 61 + j * 13) & 1023], in[(threadIdx.x * 11 + s * 37 + j * 29 + 5) & 1023]};
  f32x2 av[4][NP];
  u32x2 as[4][NP];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i0 = (blockIdx.x * 3 + s * 17 + p * 5) & 1023, i1 = (blockIdx.x * 5 + s * 23 + p * 7 + 1) & 1023;
      av[s][p] = f32x2{in[i0 + (SGPR ? 0 : (threadIdx.x >> 4) & 3)], in[i1]};
      as[s][p] = u32x2{static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(in[i0])))),
                       static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(in[i1]))))};
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if constexpr (SGPR) MAC2ROWS(acc[2 * p], acc[2 * p + 1], as[s][p], "s", b[s][0], b[s][1], b[s][2], b[s][3]);
        else MAC2ROWS(acc[2 * p], acc[2 * p + 1], av[s][p], "v", b[s][0], b[s][1], b[s][2], b[s][3]);
      }
  }
  f32x2 sum = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) sum += acc[r][c];
  out[blockIdx.x * 256 + threadIdx.x] = sum.x + sum.y;
}

int main() {
  float *in, *out;
  CHECK(hipMalloc(&in, 2048 * 4));
  CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float)));
  std::vector<float> h(2048);
  unsigned s = 12345u;
  for (auto &v : h) {
    s = s * 1664525u + 1013904223u;
    v = ((int)(s >> 8) % 20001 - 10000) * 1e-4f;  // random in [-1, 1]: products and sums toggle like audio x table
  }
  CHECK(hipMemcpy(in, h.data(), 2048 * 4, hipMemcpyHostToDevice));
  hipStream_t ps;
  CHECK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  unsigned long long *probe;
  CHECK(hipHostMalloc(reinterpret_cast<void **>(&probe), 64, hipHostMallocDefault));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 4000;
  auto run = [&](const char *name, auto kernel, int rows, int waves_per_simd) -> int {
    const int grid = 256 * waves_per_simd;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, in, iters);
    CHECK(hipDeviceSynchronize());
    float ms1 = 0;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, in, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms1, e0, e1));
    const int reps = 12;
    probe[0] = probe[1] = 0;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, ps, (unsigned long long)(0.6 * reps * ms1 * 1e-3 * 1e8), probe);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, in, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipStreamSynchronize(ps));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double macs = (double)rows * 8 * 4 * iters * 256.0 * grid;  // rows x 8 columns x 4 operand sets per iteration
    const double rate = macs / (ms * 1e-3), ghz = probe[1] ? (double)probe[0] / probe[1] * 0.1 : 0.0;
    printf("%-52s %d waves/SIMD  %7.3f ms  %6.2f T unfused-MAC/s at %.3f GHz held = %.3f of the issue bound at that clock\n", name,
           waves_per_simd, ms, rate * 1e-12, ghz, ghz > 0 ? rate / (16.0 * 1024.0 * ghz * 1e9) : 0.0);
    return 0;
  };
  for (int round = 0; round < 2; ++round) {
    run("4 x 8 tile, A pair in VGPRs (K1's form)", k_stream<4, false>, 4, 4);
    run("4 x 8 tile, A pair in SGPRs", k_stream<4, true>, 4, 4);
    run("8 x 8 tile, A pairs in VGPRs", k_stream<8, false>, 8, 2);
    run("8 x 8 tile, A pairs in SGPRs (D1's form)", k_stream<8, true>, 8, 2);
    run("4 x 8 tile, A pair in VGPRs, 2 waves per SIMD", k_stream<4, false>, 4, 2);
    run("4 x 8 tile, A pair in SGPRs, 2 waves per SIMD", k_stream<4, true>, 4, 2);
  }
  return 0;
}
