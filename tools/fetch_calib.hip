// fetch_calib.hip — known-byte-count microbenchmarks for the TCC fabric counters on gfx950.
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports half the bytes of a wide (16 B/lane)
// coalesced streaming read; other access widths are uncalibrated.  K1 reads PCM as 128-byte
// segments (8 lanes x buffer_load_dwordx4, consecutive segments 8 KiB apart) and the cosine table
// as 32-byte pieces by s_load_dwordx8 (k_mdct_fwd_st) - until round 3 as 512-byte rows by
// global_load_lds_dwordx4 (k_mdct_fwd_dma) -, and writes 32 bytes per lane and row; this tool moves a known
// number of bytes ONCE with exactly those access patterns (and the plain ones), one kernel per pattern, so that
//     rocprofv3 --pmc FETCH_SIZE -- build/fetch_calib      (and a second pass with WRITE_SIZE)
// gives counter-KB per true KB for each.  tools/pmc_traffic.py runs the passes and applies the factor.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o build/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define OK(x)                                                            \
  do {                                                                   \
    hipError_t e = (x);                                                  \
    if (e != hipSuccess) {                                               \
      std::printf("%s: %s\n", #x, hipGetErrorString(e));                 \
      std::exit(1);                                                      \
    }                                                                    \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t kBytes = 512ull << 20;  // every kernel moves exactly this much (beyond L2 and Infinity Cache)

// 4 B per lane, 256 B per wave-load, every byte once
__global__ __launch_bounds__(256) void calib_read_b32(const float *p, size_t n, float *sink) {
  float s = 0.f;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256ull) s += p[i];
  if (s == 123.456f) *sink = s;
}
// 16 B per lane, 1 KiB per wave-load
__global__ __launch_bounds__(256) void calib_read_b128(const f32x4 *p, size_t n4, float *sink) {
  float s = 0.f;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256ull) {
    const f32x4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) *sink = s;
}
// K1's PCM pattern: 8 lanes fetch one 128-byte segment with buffer_load_dwordx4; the 64 segments of a
// 512-thread workgroup are 8 KiB apart (one stereo frame hop); a workgroup walks 64 consecutive
// segments of each of its rows (= 8 KiB per row), so that every byte of its 512 KiB tile is read once.
__global__ __launch_bounds__(512) void calib_read_segments(const float *p, size_t n_bytes, float *sink) {
  const size_t tile = 64ull * 8192ull;  // bytes per workgroup
  const size_t base = blockIdx.x * tile;
  if (base + tile > n_bytes) return;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p) + base / 4, 0, static_cast<int>(tile), 0x00020000);
  const unsigned row = threadIdx.x / 8, piece = threadIdx.x % 8;
  float s = 0.f;
  for (unsigned seg = 0; seg < 64; ++seg) {
    const unsigned off = row * 8192u + seg * 128u + piece * 16u;
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(off), "s"(rsrc) : "memory");
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) *sink = s;
}
// K1's table pattern: a wave copies two 512-byte rows (4 KiB apart) per instruction straight into
// LDS with global_load_lds_dwordx4; a 512-thread workgroup covers 16 rows x 512 B per step and walks
// 8 column blocks x 128 steps... here simply every byte of a [rows][1024] float matrix once.
__global__ __launch_bounds__(512) void calib_read_lds_dma(const float *p, size_t n_rows, float *sink) {
  __shared__ __attribute__((aligned(16))) float buf[16 * 128];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float s = 0.f;
  // block b owns column block b % 8 (128 floats) of the rows [16 (b / 8) * 64, +16 * 64)
  const unsigned cb = blockIdx.x % 8;
  const size_t r0 = static_cast<size_t>(blockIdx.x / 8) * 1024;
  if (r0 + 1024 > n_rows) return;
  for (unsigned step = 0; step < 64; ++step) {
    const float *src = p + (r0 + step * 16 + 2 * wave + (lane >> 5)) * 1024 + cb * 128 + (lane & 31) * 4;
    __builtin_amdgcn_global_load_lds(src, &buf[2 * wave * 128], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    s += buf[threadIdx.x] + buf[threadIdx.x + 512] + buf[threadIdx.x + 1024] + buf[threadIdx.x + 1536];
    __syncthreads();
  }
  if (s == 123.456f) *sink = s;
}
// k_mdct_fwd_st's table pattern: a wave takes 8 consecutive floats of a 4 KiB row with one s_load_dwordx8,
// four rows per wait; the 16 waves of a workgroup cover 512 bytes of the row, 8 workgroups the row.
// Every byte of a [rows][1024] float matrix once.
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024) void calib_read_scalar(const float *p, size_t n_rows, float *sink) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned cb = blockIdx.x % 8;
  const size_t r0 = static_cast<size_t>(blockIdx.x / 8) * 1024;
  if (r0 + 1024 > n_rows) return;
  const unsigned *src = reinterpret_cast<const unsigned *>(p) + r0 * 1024 + cb * 128 + wave * 8;
  unsigned s = 0;
  for (unsigned r = 0; r < 1024; r += 4) {
    u32x8 a, b, c, d;
    const unsigned *q = src + static_cast<size_t>(r) * 1024;
    asm volatile(
        "s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x1000\n\ts_load_dwordx8 %2, %4, 0x2000\n\t"
        "s_load_dwordx8 %3, %4, 0x3000\n\ts_waitcnt lgkmcnt(0)"
        : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d)
        : "s"(q)
        : "memory");
    s += a.s0 ^ b.s1 ^ c.s2 ^ d.s3 ^ a.s7 ^ b.s6 ^ c.s5 ^ d.s4;
  }
  if (s == 0x12345678u) *sink = 1.0f;
}
// k_mdct_fwd_st's output pattern: a lane owns 4 rows (4 KiB apart) and writes 32 bytes of each as two
// 16-byte stores; the 16 waves of a workgroup fill 512 bytes of every row, 8 workgroups the row.
__global__ __launch_bounds__(1024) void calib_write_rows(float *p, size_t n_rows) {
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned cb = blockIdx.x % 8;
  const size_t r0 = static_cast<size_t>(blockIdx.x / 8) * 256;
  if (r0 + 256 > n_rows) return;
  for (unsigned r = 0; r < 4; ++r) {
    float *dst = p + (r0 + lane * 4 + r) * 1024 + cb * 128 + wave * 8;
    *reinterpret_cast<f32x4 *>(dst) = f32x4{1.f, 2.f, 3.f, 4.f};
    *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{5.f, 6.f, 7.f, 8.f};
  }
}
__global__ __launch_bounds__(256) void calib_write_b32(float *p, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256ull) p[i] = 1.0f;
}
__global__ __launch_bounds__(256) void calib_write_b128(f32x4 *p, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256ull)
    p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

int main() {
  float *d = nullptr, *sink = nullptr;
  OK(hipMalloc(&d, kBytes));
  OK(hipMalloc(&sink, 4));
  OK(hipMemset(d, 0, kBytes));
  OK(hipDeviceSynchronize());
  const size_t n = kBytes / 4;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_read_b32, dim3(4096), dim3(256), 0, 0, d, n, sink);
    hipLaunchKernelGGL(calib_read_b128, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const f32x4 *>(d), n / 4, sink);
    hipLaunchKernelGGL(calib_read_segments, dim3(kBytes / (64 * 8192)), dim3(512), 0, 0, d, kBytes, sink);
    hipLaunchKernelGGL(calib_read_lds_dma, dim3((n / 1024 / 1024) * 8), dim3(512), 0, 0, d, n / 1024, sink);
    hipLaunchKernelGGL(calib_read_scalar, dim3((n / 1024 / 1024) * 8), dim3(1024), 0, 0, d, n / 1024, sink);
    hipLaunchKernelGGL(calib_write_rows, dim3((n / 1024 / 256) * 8), dim3(1024), 0, 0, d, n / 1024);
    hipLaunchKernelGGL(calib_write_b32, dim3(4096), dim3(256), 0, 0, d, n);
    hipLaunchKernelGGL(calib_write_b128, dim3(4096), dim3(256), 0, 0, reinterpret_cast<f32x4 *>(d), n / 4);
    OK(hipDeviceSynchronize());
  }
  std::printf("fetch_calib: every kernel moved %zu bytes, 3 repetitions\n", kBytes);
  return 0;
}
