// microbench.hip — gfx950 issue-rate probes that decide the forward-MDCT kernel design.
//
// The codec's transform needs, per term, a separately rounded f32 multiply and f32 add
// (SURVEY.md F3).  Candidates for producing 1 "unfused MAC":
//   A  v_mul_f32 + v_add_f32                      (2 VALU lane-ops)
//   B  v_pk_mul_f32 + v_pk_add_f32                (2 packed VALU ops per 2 MACs)
//   C  v_mfma_f32_32x32x1_2b_f32 with C = 0 (the matrix pipe computes fl(a*b) exactly: one fma
//      with a zero addend is one rounding of the product) + v_pk_add_f32 on the VALU: the two
//      pipes run concurrently, so the multiplies leave the VALU entirely.
// Prints unfused GMAC/s for each at 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench.hip -o build/microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#pragma clang fp contract(off)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e = (x);                                                          \
    if (e != hipSuccess) {                                                       \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

// A: scalar mul + add, 32 independent accumulators
__global__ __launch_bounds__(256) void k_scalar(float *out, const float *in, int iters) {
  float acc[32];
  const float a0 = in[threadIdx.x & 63];
  float b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = in[64 + j + (threadIdx.x & 7)];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  float a = a0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t[j]) : "v"(a), "v"(b[j]));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[g * 8 + j]) : "v"(t[j]));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// B: packed mul + add, 32 independent accumulator pairs (64 MACs per pass)
__global__ __launch_bounds__(256) void k_packed(float *out, const float *in, int iters) {
  f32x2 acc[32];
  f32x2 a = {in[threadIdx.x & 63], in[(threadIdx.x + 1) & 63]};
  f32x2 b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = f32x2{in[64 + j], in[72 + j]};
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = f32x2{0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x2 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t[j]) : "v"(a), "v"(b[j]));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[g * 8 + j]) : "v"(t[j]));
    }
  }
  f32x2 s = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 32; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

// B2: as B but with the lane-broadcast op_sel forms the MDCT kernel uses
__global__ __launch_bounds__(256) void k_packed_bcast(float *out, const float *in, int iters) {
  f32x2 acc[32];
  f32x2 a = {in[threadIdx.x & 63], in[(threadIdx.x + 1) & 63]};
  f32x2 b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = f32x2{in[64 + j], in[72 + j]};
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = f32x2{0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x2 t[8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t[j]) : "v"(a), "v"(b[j]));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(t[4 + j]) : "v"(a), "v"(b[j]));
#pragma unroll
      for (int j = 0; j < 8; ++j)
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[g * 8 + j]) : "v"(t[j]));
    }
  }
  f32x2 s = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 32; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

// C: products on the matrix pipe (32x32x1, 2 blocks, C = 0), adds on the VALU
template <int PK>
__global__ __launch_bounds__(256) void k_mfma_add(float *out, const float *in, int iters) {
  f32x32 acc;
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  float a = in[threadIdx.x & 63];
  float b = in[64 + (threadIdx.x & 63)];
  f32x32 zero;
#pragma unroll
  for (int j = 0; j < 32; ++j) zero[j] = 0.f;
  for (int it = 0; it < iters; it += 2) {
    // two MFMAs in flight: their D registers are independent, the adds of one overlap the other
    f32x32 d0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0);
    f32x32 d1 = __builtin_amdgcn_mfma_f32_32x32x1f32(b, a, zero, 0, 0, 0);
    if (PK) {
#pragma unroll
      for (int j = 0; j < 32; j += 2) {
        f32x2 x = {acc[j], acc[j + 1]}, y = {d0[j], d0[j + 1]};
        x = x + y;
        acc[j] = x.x; acc[j + 1] = x.y;
      }
#pragma unroll
      for (int j = 0; j < 32; j += 2) {
        f32x2 x = {acc[j], acc[j + 1]}, y = {d1[j], d1[j + 1]};
        x = x + y;
        acc[j] = x.x; acc[j + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[j]) : "v"(d0[j]));
#pragma unroll
      for (int j = 0; j < 32; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[j]) : "v"(d1[j]));
    }
    a += 1.0f;  // keep operands changing
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// D: MFMA alone (issue rate of the f32 K=1 form)
__global__ __launch_bounds__(256) void k_mfma_only(float *out, const float *in, int iters) {
  f32x32 acc0, acc1;
#pragma unroll
  for (int j = 0; j < 32; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
  float a = in[threadIdx.x & 63];
  float b = in[64 + (threadIdx.x & 63)];
  for (int it = 0; it < iters; it += 2) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x1f32(b, a, acc1, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) s += acc0[j] + acc1[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
static int run(const char *name, F launch, double macs_per_thread_iter, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    const int grid = 256 * wg_per_cu;
    launch(grid, iters / 8);  // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0));
      launch(grid, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double macs = macs_per_thread_iter * iters * 256.0 * grid;
    const double rate = macs / (best * 1e-3);
    printf("%-28s waves/SIMD=%d  %8.3f ms  %8.2f T unfused-MAC/s  (%.2f MAC/clk/SIMD @2.4GHz)\n", name,
           wg_per_cu, best, rate * 1e-12, rate / (1024.0 * 2.4e9));
  }
  return 0;
}

int main() {
  float *in, *out;
  CHECK(hipMalloc(&in, 4096));
  CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float)));  // largest grid: 256 CUs x 8 WGs
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * (i % 97);
  CHECK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
  const int iters = 20000;
  run("A v_mul+v_add", [&](int g, int it) { hipLaunchKernelGGL(k_scalar, dim3(g), dim3(256), 0, 0, out, in, it); }, 32.0, iters);
  run("B v_pk_mul+v_pk_add", [&](int g, int it) { hipLaunchKernelGGL(k_packed, dim3(g), dim3(256), 0, 0, out, in, it); }, 64.0, iters);
  run("B2 pk with op_sel broadcast", [&](int g, int it) { hipLaunchKernelGGL(k_packed_bcast, dim3(g), dim3(256), 0, 0, out, in, it); }, 64.0, iters);
  run("C mfma32x32x1_2b + v_add", [&](int g, int it) { hipLaunchKernelGGL(k_mfma_add<0>, dim3(g), dim3(256), 0, 0, out, in, it); }, 32.0, iters);
  run("C' mfma32x32x1_2b + pk_add", [&](int g, int it) { hipLaunchKernelGGL(k_mfma_add<1>, dim3(g), dim3(256), 0, 0, out, in, it); }, 32.0, iters);
  run("D mfma32x32x1_2b only", [&](int g, int it) { hipLaunchKernelGGL(k_mfma_only, dim3(g), dim3(256), 0, 0, out, in, it); }, 32.0, iters);
  return 0;
}
