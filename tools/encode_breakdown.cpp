// encode_breakdown.cpp — where does a host-buffer glc_encode call spend its time?  
// Times the pieces through the public C ABI on BASELINE config 2 (4096 frames x 1024, stereo chord):
// the upload alone, the device encode on resident samples, the device compaction + download + host
// indexing (glc_frames_from_device_records), and the whole glc_encode.  Best of N, warm.
// Build: make -C gapless-lossy-codec_amd/csrc tools        Usage: build/encode_breakdown [frames = 4096] [ch = 2]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "glc.h"
#include "glc_debug.h"

#define OK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                \
      std::exit(1);                                                      \
    }                                                                    \
  } while (0)
#define GL(x)                                                            \
  do {                                                                   \
    int r_ = (x);                                                        \
    if (r_ != 0) {                                                       \
      std::printf("%s -> %d: %s\n", #x, r_, glc_last_error(nullptr));    \
      std::exit(1);                                                      \
    }                                                                    \
  } while (0)

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
  const uint64_t frames = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096;
  const uint16_t ch = argc > 2 ? static_cast<uint16_t>(std::atoi(argv[2])) : 2;
  const uint64_t per_ch = frames * 1024, n = per_ch * ch;
  std::vector<float> pcm(n);
  // the bench's own batch when tools/dump_d1_rows.py has written it (same sparsity as bench.py), else a stand-in chord
  bool from_file = false;
  if (FILE *fp = std::fopen("build/chord_cfg2.f32", "rb")) {
    from_file = frames == 4096 && ch == 2 && std::fread(pcm.data(), 4, n, fp) == n;
    std::fclose(fp);
  }
  if (!from_file)
    for (uint64_t t = 0; t < per_ch; ++t)
      for (uint16_t c = 0; c < ch; ++c) {
        double v = 0;
        for (int h = 0; h < 16; ++h) v += std::sin(2 * M_PI * (110.0 * (h + 1) + 7 * c) * t / 48000.0 + h) / 16;
        pcm[t * ch + c] = static_cast<float>(0.7 * v);
      }
  std::printf("input: %s\n", from_file ? "build/chord_cfg2.f32 (the bench's batch)" : "stand-in chord");
  glc_ctx *ctx = nullptr;
  GL(glc_ctx_create(0, 48000, &ctx));
  if (const char *v = std::getenv("GLC_MDCT_VARIANT")) {  // include/glc_debug.h: which forward-transform kernel (A / B runs)
    GL(glc_debug_set_mdct_variant(ctx, std::atoi(v)));
    std::printf("forward-transform variant %s\n", v);
  }
  hipStream_t s = static_cast<hipStream_t>(glc_ctx_stream(ctx));
  glc_plan plan;
  GL(glc_plan_encode(n, ch, &plan));
  float *d_pcm = nullptr;
  void *d_rec = nullptr;
  OK(hipMalloc(&d_pcm, n * 4));
  OK(hipMalloc(&d_rec, plan.n_frames * glc_record_bytes(ch)));
  // fn returns the milliseconds it wants counted
  auto best = [&](const char *name, int reps, auto fn) {
    double b = 1e30;
    for (int i = 0; i < reps; ++i) b = std::min(b, fn());
    std::printf("%-74s %8.3f ms\n", name, b);
  };
  // warm the device clocks and every lazily created buffer
  for (int i = 0; i < 30; ++i) {
    glc_frames *F = nullptr;
    GL(glc_encode(ctx, pcm.data(), n, ch, &F));
    glc_frames_free(F);
  }
  best("upload: hipMemcpy H2D of the whole stream from pageable memory", 20, [&] {
    const double t0 = now_ms();
    OK(hipMemcpy(d_pcm, pcm.data(), n * 4, hipMemcpyHostToDevice));
    return now_ms() - t0;
  });
  best("device encode on resident samples (K1 + K2), queued + synchronised", 20, [&] {
    const double t0 = now_ms();
    GL(glc_encode_range_device(ctx, d_pcm, 0, per_ch, n, ch, 0, plan.n_frames, d_rec, nullptr));
    OK(hipStreamSynchronize(s));
    return now_ms() - t0;
  });
  best("glc_frames_from_device_records (compaction + download + host index)", 20, [&] {
    glc_frames *F = nullptr;
    const double t0 = now_ms();
    GL(glc_frames_from_device_records(ctx, d_rec, plan.n_frames, n, ch, &F));
    const double t = now_ms() - t0;
    glc_frames_free(F);
    return t;
  });
  best("glc_frames_free of that result", 20, [&] {
    glc_frames *F = nullptr;
    GL(glc_frames_from_device_records(ctx, d_rec, plan.n_frames, n, ch, &F));
    const double t0 = now_ms();
    glc_frames_free(F);
    return now_ms() - t0;
  });
  best("glc_encode: host buffer in, EncodedAudio out", 30, [&] {
    glc_frames *F = nullptr;
    const double t0 = now_ms();
    GL(glc_encode(ctx, pcm.data(), n, ch, &F));
    const double t = now_ms() - t0;
    glc_frames_free(F);
    return t;
  });
  {  // ... and from a buffer the runtime has never seen (a caller that loads a file, encodes it once, frees it):
     // the pages are pinned for the first time on the way up
    double b = 1e30;
    for (int i = 0; i < 6; ++i) {
      std::vector<float> fresh(pcm.size() + 1024 * (i + 1));  // a different size each time: a new mapping, not a recycled one
      std::copy(pcm.begin(), pcm.end(), fresh.begin());
      glc_frames *F = nullptr;
      const double t0 = now_ms();
      GL(glc_encode(ctx, fresh.data(), n, ch, &F));
      b = std::min(b, now_ms() - t0);
      glc_frames_free(F);
    }
    std::printf("%-74s %8.3f ms\n", "glc_encode from a freshly allocated host buffer", b);
  }
  OK(hipFree(d_pcm));
  OK(hipFree(d_rec));
  glc_ctx_destroy(ctx);
  return 0;
}
