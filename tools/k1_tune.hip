// k1_tune.hip — times tile-shape variants of the forward-MDCT kernel (glc_mdct_fwd.hpp) against
// each other on the GPU and checks every variant bit-for-bit against a naive one-output-per-lane
// kernel that accumulates in the reference's order.  Development tool, not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I gapless-lossy-codec_amd/csrc -I tools \
//        tools/k1_tune.hip -o build/k1_tune
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "glc_mdct_fwd.hpp"  // namespace glc::k1: the kernels the library ships
#include "k1_row_variant.hpp"  // k_mdct_fwd_row: one row per lane (measured, not adopted)
#include "k1_variants.hpp"   // namespace glc::k1x: every other shape / schedule / ablation (tuning only)

#pragma clang fp contract(off)

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

using namespace glc;

// naive reference: one (row, k) per thread, i ascending, separate mul and add
__global__ void k_naive(DeviceTables tb, PcmView pcm, long long frame_begin, unsigned M, float *coef) {
  const unsigned k = blockIdx.x * 256 + threadIdx.x;
  const unsigned m = blockIdx.y;
  if (m >= M || k >= 1024) return;
  const long long f = frame_begin + m / pcm.ch;
  const unsigned c = m % pcm.ch;
  float s = 0.f;
  for (int i = 0; i < 2048; ++i) {
    const long long t = f * 1024 + i - 512;
    float x = 0.f;
    if (t >= 0) {
      const unsigned long long idx = (unsigned long long)t * pcm.ch + c;
      if (idx < pcm.n_samples && (unsigned long long)t >= pcm.t0 && (unsigned long long)t - pcm.t0 < pcm.t_count)
        x = pcm.p[((unsigned long long)t - pcm.t0) * pcm.ch + c];
    }
    const float b = __fmul_rn(x, tb.window[i]);
    s = __fadd_rn(s, __fmul_rn(b, tb.cos_t[(size_t)i * 1024 + k]));
  }
  coef[(size_t)m * 1024 + k] = __fmul_rn(s, tb.norm);
}

// launch_dma has a defaulted trailing argument (the timeline buffer): adapt it to the harness' signature
template <int MINW, int CH, int PRIO>
static hipError_t dma_prio(const DeviceTables &t, const PcmView &pcm, uint64_t f0, uint32_t M, float *coef, hipStream_t s) {
  return k1::launch_dma<MINW, CH, PRIO>(t, pcm, f0, M, coef, s);
}
// ... and launch_st
template <int... Ps>
static hipError_t st_v(const DeviceTables &t, const PcmView &pcm, uint64_t f0, uint32_t M, float *coef, hipStream_t s) {
  return k1::launch_st<Ps...>(t, pcm, f0, M, coef, s);
}
template <int MINW, int CH = 0>
static hipError_t dma_shipped(const DeviceTables &t, const PcmView &pcm, uint64_t f0, uint32_t M, float *coef, hipStream_t s) {
  return k1::launch_dma<MINW, CH>(t, pcm, f0, M, coef, s);
}

struct Variant {
  std::string name;
  std::function<hipError_t(const DeviceTables &, const PcmView &, uint64_t, uint32_t, float *, hipStream_t)> fn;
};

#define V(BM, BN, BK, TM, TN, UN, MW) \
  Variant { #BM "x" #BN " bk" #BK " t" #TM "x" #TN " u" #UN " w" #MW, k1x::launch<BM, BN, BK, TM, TN, UN, MW> }

__global__ void k_fill_random(float *p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    unsigned x = seed ^ (unsigned)(i * 2654435761ull);
    x ^= x << 13, x ^= x >> 17, x ^= x << 5;
    p[i] = ((int)(x % 20001u) - 10000) * 1e-4f * 0.3f;
  }
}

// one sleeping wave beside the kernel under test: shader cycles against the 100 MHz reference = the clock the
// chip holds under that kernel (MI355X_MICROARCH.md, DVFS give-back item 6)
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

__global__ void k_count_diff(const unsigned *a, const unsigned *b, size_t n, unsigned long long *bad) {
  unsigned long long local = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) local += a[i] != b[i];
  if (local) atomicAdd(bad, local);
}

int main(int argc, char **argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 4096;
  const unsigned ch = argc > 2 ? atoi(argv[2]) : 2;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const uint64_t L = (uint64_t)frames * 1024;
  const uint64_t n_samples = L * ch;
  const uint32_t M = frames * ch;

  std::vector<float> h_cos_t((size_t)2048 * 1024), h_win(2048), h_pcm(n_samples);
  srand(1);
  for (auto &v : h_cos_t) v = cosf((float)(rand() % 100000) * 0.001f);
  for (int i = 0; i < 2048; ++i) h_win[i] = sinf(3.14159265f * (i + 0.5f) / 2048.f);
  // argv[4]: "zero" = all-zero PCM (least switching activity), "chord" = smooth tonal data, default random
  const std::string fill = argc > 4 ? argv[4] : "random";
  for (size_t i = 0; i < h_pcm.size(); ++i) {
    if (fill == "zero") h_pcm[i] = 0.0f;
    else if (fill == "chord") h_pcm[i] = 0.05f * (sinf(0.013f * (float)(i / ch)) + sinf(0.171f * (float)(i / ch)) + sinf(0.0007f * (float)(i / ch)));
    else h_pcm[i] = ((rand() % 20001) - 10000) * 1e-4f * 0.3f;
  }

  float *d_cos_t, *d_win, *d_pcm, *d_ref, *d_out;
  CHECK(hipMalloc(&d_cos_t, h_cos_t.size() * 4));
  CHECK(hipMalloc(&d_win, 2048 * 4));
  CHECK(hipMalloc(&d_pcm, n_samples * 4));
  CHECK(hipMalloc(&d_ref, (size_t)M * 1024 * 4));
  CHECK(hipMalloc(&d_out, (size_t)M * 1024 * 4));
  CHECK(hipMemcpy(d_cos_t, h_cos_t.data(), h_cos_t.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_win, h_win.data(), 2048 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_pcm, h_pcm.data(), n_samples * 4, hipMemcpyHostToDevice));

  // [column group of 8][i][8] copy of the table for k_mdct_fwd_row8 (tools/k1_row_variant.hpp), through the
  // field the forward kernels do not use
  std::vector<float> h_t8(h_cos_t.size());
  for (size_t i = 0; i < 2048; ++i)
    for (size_t k = 0; k < 1024; ++k) h_t8[((k / 8) * 2048 + i) * 8 + k % 8] = h_cos_t[i * 1024 + k];
  float *d_t8;
  CHECK(hipMalloc(&d_t8, h_t8.size() * 4));
  CHECK(hipMemcpy(d_t8, h_t8.data(), h_t8.size() * 4, hipMemcpyHostToDevice));
  DeviceTables tb{};
  tb.cos = d_t8;
  tb.cos_t = d_cos_t;
  tb.window = d_win;
  tb.norm = 0.044194173f;
  PcmView pcm{d_pcm, 0, L, n_samples, ch};

  if (fill == "soak") {
    // soak: `reps` rounds of fresh random PCM through the shipped kernels, every output compared on
    // the device with the naive kernel's (a rare hazard in the hand-written LDS ring / counted waits
    // would show up as a mismatch in some round); prints one progress line per 100 rounds
    unsigned long long *d_bad;
    CHECK(hipMalloc(&d_bad, 8));
    CHECK(hipMemset(d_bad, 0, 8));
    unsigned long long total_bad = 0;
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_fill_random, dim3(1024), dim3(256), 0, 0, d_pcm, n_samples, 0x9E3779B9u * (unsigned)(r + 1));
      hipLaunchKernelGGL(k_naive, dim3(4, M), dim3(256), 0, 0, tb, pcm, 0ll, M, d_ref);
      // the instantiations the library ships (glc_kernels.hip launch_mdct_forward): k_mdct_fwd_st with 16 and
      // with 8 waves per workgroup, the channel count's segment loader and the per-row loader in turn; below
      // 4096 rows the short-clip / 64 x 128 kernels the library would take for that row count
      hipError_t launched;
      if (M <= 640) launched = k1::launch_small<2>(tb, pcm, 0, M, d_out, 0);
      else if (M > 1792 && M <= 2048 && (r & 1)) launched = k1::launch_sched<64, 128, 16, 4>(tb, pcm, 0, M, d_out, 0);  // glc_encode's opening rounds
      else if (M < 3584 || (M < 4096 && (r & 4))) launched = k1::launch_small<4>(tb, pcm, 0, M, d_out, 0);
      else {
        const bool seg = (r & 2) == 0 && (ch == 1 || ch == 2 || ch == 4 || ch == 8);
        if (r & 1) {
          launched = !seg ? k1::launch_st<4, 0, 1, 4, 8>(tb, pcm, 0, M, d_out, 0)
                   : ch == 1 ? k1::launch_st<4, 1, 1, 4, 8>(tb, pcm, 0, M, d_out, 0)
                   : ch == 2 ? k1::launch_st<4, 2, 1, 4, 8>(tb, pcm, 0, M, d_out, 0)
                   : ch == 4 ? k1::launch_st<4, 4, 1, 4, 8>(tb, pcm, 0, M, d_out, 0)
                             : k1::launch_st<4, 8, 1, 4, 8>(tb, pcm, 0, M, d_out, 0);
        } else {
          launched = !seg ? k1::launch_st<4, 0, 2, 4, 16>(tb, pcm, 0, M, d_out, 0)
                   : ch == 1 ? k1::launch_st<4, 1, 2, 4, 16>(tb, pcm, 0, M, d_out, 0)
                   : ch == 2 ? k1::launch_st<4, 2, 2, 4, 16>(tb, pcm, 0, M, d_out, 0)
                   : ch == 4 ? k1::launch_st<4, 4, 2, 4, 16>(tb, pcm, 0, M, d_out, 0)
                             : k1::launch_st<4, 8, 2, 4, 16>(tb, pcm, 0, M, d_out, 0);
        }
      }
      CHECK(launched);
      hipLaunchKernelGGL(k_count_diff, dim3(1024), dim3(256), 0, 0, reinterpret_cast<const unsigned *>(d_ref),
                         reinterpret_cast<const unsigned *>(d_out), (size_t)M * 1024, d_bad);
      if (r % 100 == 99 || r == reps - 1) {
        CHECK(hipMemcpy(&total_bad, d_bad, 8, hipMemcpyDeviceToHost));
        printf("soak round %d: %llu mismatching coefficients so far\n", r + 1, total_bad);
        fflush(stdout);
      }
    }
    return total_bad ? 1 : 0;
  }

  if (fill == "timeline" || (argc > 5 && std::string(argv[5]) == "timeline")) {
    // workgroup timeline of the dma kernel (stereo): when do the two workgroups of a CU (blocks b, b + 256)
    // pass the quarter points of the i loop, with and without the priority schedule?
    if (ch != 2 || M < 4096) return printf("timeline: needs stereo and >= 4096 rows\n"), 1;
    const unsigned nblk = (M + 127) / 128 * 8;
    unsigned long long *d_st;
    CHECK(hipMalloc(&d_st, size_t(nblk) * 64));
    std::vector<unsigned long long> st(size_t(nblk) * 8);
    auto show = [&](const char *name, auto launch) {
      for (int i = 0; i < 20; ++i) CHECK(launch());  // warm clocks
      CHECK(hipMemset(d_st, 0, size_t(nblk) * 64));
      CHECK(launch());
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      for (unsigned b = 0; b < nblk; ++b) t0 = std::min(t0, st[size_t(b) * 8]);
      printf("%s\n", name);
      for (int q = 0; q <= 4; ++q) {  // 0 = loop start, 1..3 = quarter points, 4 = loop end
        std::vector<double> a, d;
        for (unsigned b = 0; b < nblk; ++b) a.push_back((st[size_t(b) * 8 + q] - t0) * 0.01);
        for (unsigned b = 0; b + 256 < nblk; ++b) d.push_back(std::fabs(a[b] - a[b + 256]));
        std::vector<double> sa = a;
        std::sort(sa.begin(), sa.end());
        std::sort(d.begin(), d.end());
        printf("  %-13s us: min %7.1f p10 %7.1f p50 %7.1f p90 %7.1f max %7.1f | partners (b, b + 256) apart: p50 %6.1f p90 %6.1f max %6.1f\n",
               q == 0 ? "loop start" : q == 4 ? "loop end" : q == 1 ? "1/4 of loop" : q == 2 ? "1/2 of loop" : "3/4 of loop",
               sa.front(), sa[sa.size() / 10], sa[sa.size() / 2], sa[sa.size() * 9 / 10], sa.back(),
               d.empty() ? 0.0 : d[d.size() / 2], d.empty() ? 0.0 : d[d.size() * 9 / 10], d.empty() ? 0.0 : d.back());
      }
      fflush(stdout);
    };
    // k_mdct_fwd_st stamps: 0 entry, 1 loop start, 2..4 quarter points, 5 loop end, 6 stores drained
    auto show_st = [&](const char *name, unsigned n_st, auto launch) {
      for (int i = 0; i < 200; ++i) CHECK(launch());  // warm clocks; the stamps kept are those of the LAST launch
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      for (unsigned b = 0; b < n_st; ++b) t0 = std::min(t0, st[size_t(b) * 8]);
      printf("%s (%u workgroups; microseconds after the first workgroup's first instruction)\n", name, n_st);
      static const char *what[7] = {"entry", "loop start", "1/4 of loop", "1/2 of loop", "3/4 of loop", "loop end", "stores drained"};
      for (int q = 0; q <= 6; ++q) {
        std::vector<double> a;
        for (unsigned b = 0; b < n_st; ++b) a.push_back((st[size_t(b) * 8 + q] - t0) * 0.01);
        std::sort(a.begin(), a.end());
        printf("  %-15s us: min %7.1f p10 %7.1f p50 %7.1f p90 %7.1f max %7.1f\n", what[q], a.front(), a[a.size() / 10],
               a[a.size() / 2], a[a.size() * 9 / 10], a.back());
      }
      // the dispatcher deals consecutive workgroups round-robin over the 8 XCDs: loop end by XCD
      printf("  loop end by XCD (mean / max us):");
      for (unsigned x = 0; x < 8; ++x) {
        double sum = 0, mx = 0;
        unsigned n = 0;
        for (unsigned b = x; b < n_st; b += 8, ++n) {
          const double v = (st[size_t(b) * 8 + 5] - t0) * 0.01;
          sum += v;
          mx = std::max(mx, v);
        }
        printf("  %u: %.1f / %.1f", x, sum / n, mx);
      }
      printf("\n");
      fflush(stdout);
    };
    show_st("k_mdct_fwd_st, 16 waves per workgroup, PRIO 2 (shipped at config 2)", (M + 255) / 256 * 8,
            [&] { return k1::launch_st<4, 2, 2, 4, 16, 16, 0, true>(tb, pcm, 0, M, d_out, 0, d_st); });
    show_st("k_mdct_fwd_st, 8 waves per workgroup, PRIO 1", (M + 255) / 256 * 16,
            [&] { return k1::launch_st<4, 2, 1, 4, 8, 16, 0, true>(tb, pcm, 0, M, d_out, 0, d_st); });
    show_st("k_mdct_fwd_st, 16 waves per workgroup, PRIO 2 (again)", (M + 255) / 256 * 8,
            [&] { return k1::launch_st<4, 2, 2, 4, 16, 16, 0, true>(tb, pcm, 0, M, d_out, 0, d_st); });
    show("k_mdct_fwd_dma PRIO 0 (round 2)", [&] { return k1::launch_dma<4, 2, 0, true>(tb, pcm, 0, M, d_out, 0, d_st); });
    show("k_mdct_fwd_dma PRIO 1 (by quarter; round 3, first half)", [&] { return k1::launch_dma<4, 2, 1, true>(tb, pcm, 0, M, d_out, 0, d_st); });
    show("k_mdct_fwd_dma PRIO 3 (cycle of 16 stages)", [&] { return k1::launch_dma<4, 2, 3, true>(tb, pcm, 0, M, d_out, 0, d_st); });
    return 0;
  }

  hipLaunchKernelGGL(k_naive, dim3(4, M), dim3(256), 0, 0, tb, pcm, 0ll, M, d_ref);
  CHECK(hipDeviceSynchronize());
  std::vector<uint32_t> ref((size_t)M * 1024), out((size_t)M * 1024);
  CHECK(hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost));

  // hand-scheduled kernels (shipped shapes first), their ablations, then hipcc-scheduled shapes
  std::vector<Variant> vs = {
      // (the first line of a run is measured on a cold device: it is repeated further down)
      Variant{"warm-up: SHIPPED dma, segment loader", dma_shipped<4, 2>},
      // the three kernels libglc_hip.so ships (csrc/glc_mdct_fwd.hpp)
      Variant{"SHIPPED dma 128x128 512thr, per-row PCM loader (>= 4096 rows, other channel counts)", dma_shipped<4>},
      Variant{"SHIPPED dma 128x128 512thr, dwordx4 segment loader (>= 4096 rows, stereo)", dma_shipped<4, 2>},
      Variant{"SHIPPED dma PRIO 1, the channel count's own loader (1 / 4 / 8: segments, else per row)",
              [](const DeviceTables &t, const PcmView &p, uint64_t f0, uint32_t M, float *c, hipStream_t s) {
                switch (p.ch) {
                  case 1: return k1::launch_dma<4, 1, 1>(t, p, f0, M, c, s);
                  case 2: return k1::launch_dma<4, 2, 1>(t, p, f0, M, c, s);
                  case 4: return k1::launch_dma<4, 4, 1>(t, p, f0, M, c, s);
                  case 8: return k1::launch_dma<4, 8, 1>(t, p, f0, M, c, s);
                  default: return k1::launch_dma<4, 0, 1>(t, p, f0, M, c, s);
                }
              }},
      // k_mdct_fwd_row (tools/k1_row_variant.hpp): one row per lane, C columns per wave from SGPRs; measured for
      // launches below 4096 rows and not adopted (K1_FILTER='[row]' build/k1_tune <frames> 2 20)
      Variant{"[row] C=2 bk32, segment loader (stereo)", k1::launch_row<2, 2, 32>},
      Variant{"[row] C=4 bk32, segment loader (stereo)", k1::launch_row<4, 2, 32>},
      Variant{"[row] C=8 bk32, segment loader (stereo)", k1::launch_row<8, 2, 32>},
      Variant{"[row] C=2 bk64, segment loader (stereo)", k1::launch_row<2, 2, 64>},
      Variant{"[row] C=4 bk64, segment loader (stereo)", k1::launch_row<4, 2, 64>},
      Variant{"[row] C=8 bk64, segment loader (stereo)", k1::launch_row<8, 2, 64>},
      Variant{"[row] C=4 bk16, segment loader (stereo)", k1::launch_row<4, 2, 16>},
      Variant{"[row] C=8 bk16, segment loader (stereo)", k1::launch_row<8, 2, 16>},
      Variant{"[row] C=8 bk16, table copy [group][i][8]: two 64-byte scalar loads per 4 i-steps", k1::launch_row8<2, 16>},
      Variant{"[row] C=8 bk32, table copy [group][i][8]", k1::launch_row8<2, 32>},
      Variant{"[row] C=4 bk32, per-row loader", k1::launch_row<4, 0, 32>},
      Variant{"[row] C=8 bk32, per-row loader", k1::launch_row<8, 0, 32>},
      Variant{"[row] SHIPPED small 2x2 (<= 640 rows)", k1::launch_small<2>},
      Variant{"[row] SHIPPED small 2x4 (641..3583 rows)", k1::launch_small<4>},
      Variant{"[row] SHIPPED sched 64x128 (the opening rounds of glc_encode; 1793..4095 rows until round 3)", k1::launch_sched<64, 128, 16, 4>},
      Variant{"[row] st 8 waves (>= 4096 rows)", st_v<4, 2, 1, 4>},
      // k_mdct_fwd_st: the table from SGPRs, lanes <-> rows (K1_FILTER='[cand]' K1_ROUNDS=4 compares interleaved)
      Variant{"[cand] dma 128x128 PRIO 1 (shipped until round 3), the channel count's own loader",
              [](const DeviceTables &t, const PcmView &p, uint64_t f0, uint32_t M, float *c, hipStream_t s) {
                switch (p.ch) {
                  case 1: return k1::launch_dma<4, 1, 1>(t, p, f0, M, c, s);
                  case 2: return k1::launch_dma<4, 2, 1>(t, p, f0, M, c, s);
                  case 4: return k1::launch_dma<4, 4, 1>(t, p, f0, M, c, s);
                  case 8: return k1::launch_dma<4, 8, 1>(t, p, f0, M, c, s);
                  default: return k1::launch_dma<4, 0, 1>(t, p, f0, M, c, s);
                }
              }},
      Variant{"[cand] st 8 waves 256x64 bk16 d4 PRIO 1, the channel count's own loader (SHIPPED, 4096..8191 rows)",
              [](const DeviceTables &t, const PcmView &p, uint64_t f0, uint32_t M, float *c, hipStream_t s) {
                switch (p.ch) {
                  case 1: return k1::launch_st<4, 1, 1, 4>(t, p, f0, M, c, s);
                  case 2: return k1::launch_st<4, 2, 1, 4>(t, p, f0, M, c, s);
                  case 4: return k1::launch_st<4, 4, 1, 4>(t, p, f0, M, c, s);
                  case 8: return k1::launch_st<4, 8, 1, 4>(t, p, f0, M, c, s);
                  default: return k1::launch_st<4, 0, 1, 4>(t, p, f0, M, c, s);
                }
              }},
      Variant{"[cand] st 16 waves 256x128 bk16 d4 PRIO 2, the channel count's own loader (SHIPPED, >= 8192 rows)",
              [](const DeviceTables &t, const PcmView &p, uint64_t f0, uint32_t M, float *c, hipStream_t s) {
                switch (p.ch) {
                  case 1: return k1::launch_st<4, 1, 2, 4, 16>(t, p, f0, M, c, s);
                  case 2: return k1::launch_st<4, 2, 2, 4, 16>(t, p, f0, M, c, s);
                  case 4: return k1::launch_st<4, 4, 2, 4, 16>(t, p, f0, M, c, s);
                  case 8: return k1::launch_st<4, 8, 2, 4, 16>(t, p, f0, M, c, s);
                  default: return k1::launch_st<4, 0, 2, 4, 16>(t, p, f0, M, c, s);
                }
              }},
      // (2 rows per lane = 8 waves per SIMD was measured with a TR parameter that is not kept: 0.588-0.611 ms
      // against 0.547-0.554 for the shipped forms in the same process, profiles/r03_k1_tune_st_two_rows.txt; with
      // 4 i-steps per fetch hipcc ran out of scalar registers there and spilled table values into VGPR lanes
      // BETWEEN the scalar load and its s_waitcnt - stale values, wrong results, a memory fault: the case
      // tools/check_isa.py refuses for the shipped kernels)
      Variant{"[cand] st 8 waves bk16 d2 PRIO 1", st_v<4, 2, 1>},
      Variant{"[cand] st 16 waves bk32 d4 PRIO 2", st_v<4, 2, 2, 4, 16, 32>},
      Variant{"[st] 8 waves bk16 d2 PRIO 0", st_v<4, 2, 0>},
      Variant{"[st] 16 waves bk16 d2 PRIO 0", st_v<4, 2, 0, 2, 16, 16>},
      Variant{"[st] 16 waves bk16 d4 PRIO 0", st_v<4, 2, 0, 4, 16, 16>},
      Variant{"[st] 16 waves bk32 d2 PRIO 2", st_v<4, 2, 2, 2, 16, 32>},
      // ablations (results are wrong by construction)
      Variant{"[abl] st 16 waves bk16 d4 PRIO 2: table address does not advance (scalar cache hits)", st_v<4, 2, 2, 4, 16, 16, 1>},
      Variant{"[abl] st 16 waves bk16 d4 PRIO 2: no staging, no barrier", st_v<4, 2, 2, 4, 16, 16, 2>},
      Variant{"[abl] st 16 waves bk16 d4 PRIO 2: neither", st_v<4, 2, 2, 4, 16, 16, 3>},
      Variant{"[abl] st 8 waves bk16 d4 PRIO 1: table address does not advance", st_v<4, 2, 1, 4, 8, 16, 1>},
      Variant{"[abl] st 8 waves bk16 d4 PRIO 1: no staging, no barrier", st_v<4, 2, 1, 4, 8, 16, 2>},
      Variant{"[abl] st 8 waves bk16 d4 PRIO 1: neither", st_v<4, 2, 1, 4, 8, 16, 3>},
      Variant{"SHIPPED sched 64x128 256thr (the opening rounds of glc_encode; 1793..4095 rows until round 3)", k1::launch_sched<64, 128, 16, 4>},
      Variant{"SHIPPED small 32x32 t2x2 256thr, hand-scheduled (<= 640 rows)", k1::launch_small<2>},
      Variant{"SHIPPED small 32x64 t2x4 256thr, hand-scheduled (641..3583 rows)", k1::launch_small<4>},
      Variant{"round 2: hipcc-scheduled 32x64 t4x4 (was shipped for <= 512 rows)", k1::launch<32, 64, 32, 4, 4, 4, 2>},
      // issue priority as a schedule (glc_mdct_fwd.hpp PRIO): whichever workgroup of a CU is behind goes first
      Variant{"dma segment loader, PRIO 1: priority by quarter of the loop (3, 2, 1, 0)", dma_prio<4, 2, 1>},
      Variant{"dma segment loader, PRIO 2: four levels cycling every 8 stages", dma_prio<4, 2, 2>},
      Variant{"dma segment loader, PRIO 3: four levels cycling every 16 stages", dma_prio<4, 2, 3>},
      Variant{"dma segment loader, PRIO 4: four levels cycling every 32 stages", dma_prio<4, 2, 4>},
      Variant{"dma segment loader, PRIO 5: rungs of 64, 32, 16, 16 stages", dma_prio<4, 2, 5>},
      Variant{"dma segment loader, PRIO 6: rungs of 48, 48, 24, 8 stages", dma_prio<4, 2, 6>},
      Variant{"SHIPPED dma, segment loader (PRIO 0, again)", dma_shipped<4, 2>},
      Variant{"dma segment loader, PRIO 1 (again)", dma_prio<4, 2, 1>},
      Variant{"dma segment loader, PRIO 3 (again)", dma_prio<4, 2, 3>},
      Variant{"dma segment loader, PRIO 5 (again)", dma_prio<4, 2, 5>},
      Variant{"dma segment loader, PRIO 6 (again)", dma_prio<4, 2, 6>},
      Variant{"dma segment loader, PRIO 1 (third time)", dma_prio<4, 2, 1>},
      // tuning variants (tools/k1_variants.hpp)
      Variant{"dma segment loader, round-1 protocol: end-of-stage hand-off, counted vmcnt(1)", k1x::launch_dma<4, 0, 128, 2, 0>},
      Variant{"dma segment loader, issue priority alternates between a CU's two workgroups every stage", k1x::launch_dma<4, 0, 128, 2, -1>},
      Variant{"dma segment loader, ... every 2 stages", k1x::launch_dma<4, 0, 128, 2, -2>},
      Variant{"dma segment loader, ... every 8 stages", k1x::launch_dma<4, 0, 128, 2, -8>},
      Variant{"dma segment loader, no stagger (again)", k1x::launch_dma<4, 0, 128, 2, 0>},
      Variant{"SHIPPED dma, segment loader (again: mid-stage hand-off)", dma_shipped<4, 2>},
      Variant{"dma segment loader, end-of-stage hand-off (k1x, again)", k1x::launch_dma<4, 0, 128, 2, 0>},
      Variant{"SHIPPED dma, segment loader (third time)", dma_shipped<4, 2>},
      Variant{"dma segment loader, end-of-stage hand-off (k1x, third time)", k1x::launch_dma<4, 0, 128, 2, 0>},
      Variant{"dma segment loader, odd workgroups start 16 x 64 cycles late", k1x::launch_dma<4, 0, 128, 2, 16>},
      Variant{"dma segment loader, odd workgroups start 32 x 64 cycles late", k1x::launch_dma<4, 0, 128, 2, 32>},
      Variant{"dma segment loader, odd workgroups start 64 x 64 cycles late", k1x::launch_dma<4, 0, 128, 2, 64>},
      Variant{"dma segment loader, odd workgroups start 100 x 64 cycles late", k1x::launch_dma<4, 0, 128, 2, 100>},
      Variant{"sched pk 128x128 bk16 t4x8 w4 512thr (shipped, M < 16384)", k1x::launch_sched<128, 128, 16, 4, 0, 4>},
      Variant{"sched pk 128x128 bk16 t8x8 w3 256thr (shipped, M >= 16384)", k1x::launch_sched<128, 128, 16, 3, 0, 8>},
      Variant{"sched pk 128x128 bk16 t4x8 w4 512thr + window in LDS", k1x::launch_sched<128, 128, 16, 4, 0, 4, false, 2, true>},
      Variant{"sched pk 128x128 bk16 t8x8 w3 256thr + window in LDS", k1x::launch_sched<128, 128, 16, 3, 0, 8, false, 2, true>},
      Variant{"dma   pk 128x128 bk16 t4x8 w4 512thr ring3 (table by LDS-DMA)", k1x::launch_dma<4>},
      Variant{"dma   pk 128x128 ... w3", k1x::launch_dma<3>},
      Variant{"dma   pk 128x128 + segment loader (one dwordx4 per lane and stage)", k1x::launch_dma<4, 0, 128, 2>},
      Variant{"mx    128x128 bk16 256thr: products by v_mfma_f32_32x32x1_2b (C = 0), adds by VALU, w2", k1x::launch_mx<16, 2>},
      Variant{"mx    128x128 bk16 ... w3", k1x::launch_mx<16, 3>},
      Variant{"mx    128x128 bk16 ... w4", k1x::launch_mx<16, 4>},
      Variant{"mx    128x128 bk32 ... w2", k1x::launch_mx<32, 2>},
      Variant{"dma   pk  64x128 bk16 t4x8 w4 256thr ring3, window by scalar loads", k1x::launch_dma<4, 0, 64>},
      Variant{"dma   pk  64x128 ... w5", k1x::launch_dma<5, 0, 64>},
      Variant{"  512thr ABL1 (no staging, no barrier)", k1x::launch_sched<128, 128, 16, 4, 1, 4>},
      Variant{"  512thr ABL2 (pure VALU stream)", k1x::launch_sched<128, 128, 16, 4, 2, 4>},
      Variant{"  512thr ABL3 (staging, no barrier)", k1x::launch_sched<128, 128, 16, 4, 3, 4>},
      Variant{"  512thr ABL4 (barrier, no staging)", k1x::launch_sched<128, 128, 16, 4, 4, 4>},
      Variant{"sched pk  64x128 bk16 t4x8 w4 256thr", k1x::launch_sched<64, 128, 16, 4, 0, 4>},
      Variant{"sched pk  64x128 bk16 t4x8 w4 ring3", k1x::launch_sched<64, 128, 16, 4, 0, 4, false, 3>},
      Variant{"sched sc  64x128 bk16 t4x8 w4 (scalar v_mul/v_add)", k1x::launch_sched<64, 128, 16, 4, 0, 4, true>},
      Variant{"sched pk  64x128 bk8  t4x8 w4", k1x::launch_sched<64, 128, 8, 4, 0, 4>},
      Variant{"sched pk  64x128 bk32 t4x8 w4", k1x::launch_sched<64, 128, 32, 4, 0, 4>},
      V(128, 128, 16, 8, 8, 2, 2), V(128, 128, 8, 8, 8, 2, 2), V(64, 128, 16, 4, 8, 2, 4),
      V(128, 128, 16, 4, 8, 2, 2), V(64, 64, 16, 4, 4, 4, 8),
      // short clips (run with e.g. `k1_tune 86 2`): latency of the 2048-step chain, not throughput
      V(16, 64, 16, 4, 4, 2, 1), V(16, 64, 32, 4, 4, 2, 1), V(32, 64, 16, 4, 4, 2, 1), V(16, 128, 16, 4, 4, 2, 1),
      V(32, 128, 32, 4, 4, 2, 1), V(16, 64, 64, 4, 4, 4, 1), V(32, 64, 32, 4, 4, 4, 2),
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipStream_t probe_stream;
  CHECK(hipStreamCreateWithFlags(&probe_stream, hipStreamNonBlocking));
  unsigned long long *h_probe;
  CHECK(hipHostMalloc(reinterpret_cast<void **>(&h_probe), 64, hipHostMallocDefault));
  const double macs = (double)M * 1024.0 * 2048.0;
  // K1_FILTER=<substring>: only the variants whose name contains it;  K1_ROUNDS=<n>: the whole list n times
  // (the clock the chip holds drifts by a few per cent from one measurement to the next: compare interleaved)
  const char *filter = getenv("K1_FILTER");
  const int rounds = getenv("K1_ROUNDS") ? atoi(getenv("K1_ROUNDS")) : 1;
  for (int round = 0; round < rounds; ++round)
  for (auto &v : vs) {
    if (filter && v.name.find(filter) == std::string::npos) continue;
    CHECK(hipMemset(d_out, 0xFF, (size_t)M * 1024 * 4));
    {
      const hipError_t first = v.fn(tb, pcm, 0, M, d_out, 0);
      if (first == hipErrorInvalidValue) {  // a variant built for another channel count
        (void)hipGetLastError();
        continue;
      }
      CHECK(first);
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != out[i];
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r) {
      CHECK(hipEventRecord(e0));
      CHECK(v.fn(tb, pcm, 0, M, d_out, 0));
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      sum += ms;
    }
    // second pass: the clock held under this kernel (probe window = 60 % of `reps` launches) and the time in it
    h_probe[0] = h_probe[1] = 0;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, probe_stream, (unsigned long long)(0.6 * reps * best * 1e-3 * 1e8), h_probe);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) CHECK(v.fn(tb, pcm, 0, M, d_out, 0));
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipStreamSynchronize(probe_stream));
    float ms_loop;
    CHECK(hipEventElapsedTime(&ms_loop, e0, e1));
    const double ghz = h_probe[1] ? (double)h_probe[0] / (double)h_probe[1] * 0.1 : 0.0;
    const double per_launch = ms_loop / reps;
    // 16 unfused MAC / clk / SIMD is the vector ALU's issue bound; 1024 SIMDs
    const double util = ghz > 0 ? macs / (per_launch * 1e-3) / (16.0 * 1024.0 * ghz * 1e9) : 0.0;
    printf("%-28s  best %7.3f ms  avg %7.3f ms  %6.2f T unfused-MAC/s  mismatches %zu | back to back %7.3f ms at %.3f GHz held = %.3f of the issue bound at that clock\n",
           v.name.c_str(), best, sum / reps, macs / (best * 1e-3) * 1e-12, bad, per_launch, ghz, util);
    fflush(stdout);
  }
  return 0;
}
