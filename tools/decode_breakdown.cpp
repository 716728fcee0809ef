// decode_breakdown.cpp — the inverse transform (D1) and the whole device decode timed through the
// C ABI without Python in the process, on BASELINE config 2 (4096 frames x 1024, stereo chord); the
// like-for-like partner of tools/bench_d1.py and of the `L` family of tools/d1_tune.hip.
// Build: make -C gapless-lossy-codec_amd/csrc tools      Usage: build/decode_breakdown [variants, e.g. 0,4,0,4] [reps = 100] [frames = 4096]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "glc.h"
#include "glc_debug.h"

#define OK(x)                                                         \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) {                                           \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                   \
    }                                                                 \
  } while (0)
#define GL(x)                                                         \
  do {                                                                \
    int r_ = (x);                                                     \
    if (r_ != 0) {                                                    \
      std::printf("%s -> %d: %s\n", #x, r_, glc_last_error(nullptr)); \
      std::exit(1);                                                   \
    }                                                                 \
  } while (0)

int main(int argc, char **argv) {
  const char *variants = argc > 1 ? argv[1] : "0,4,0,4";
  const int reps = argc > 2 ? std::atoi(argv[2]) : 100;
  const uint64_t frames = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 4096;
  const uint16_t ch = 2;
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
  glc_ctx *enc = nullptr, *dec = nullptr;
  GL(glc_ctx_create(0, 48000, &enc));
  GL(glc_ctx_create(0, 48000, &dec));
  glc_frames *F = nullptr;
  GL(glc_encode(enc, pcm.data(), n, ch, &F));
  float *d_blk = nullptr, *d_out = nullptr;
  OK(hipMalloc(&d_blk, frames * ch * 2048 * sizeof(float)));
  OK(hipMalloc(&d_out, (frames + 1) * 1024 * ch * sizeof(float)));
  for (int i = 0; i < 300; ++i) GL(glc_imdct_device(dec, F, 0, frames, d_blk));  // clocks up
  GL(glc_ctx_synchronize(dec));
  for (const char *p = variants; *p;) {
    const int v = std::atoi(p);
    GL(glc_debug_set_imdct_variant(dec, v));
    for (int i = 0; i < 20; ++i) GL(glc_imdct_device(dec, F, 0, frames, d_blk));
    float ms = 0;
    GL(glc_ctx_timer_begin(dec));
    for (int i = 0; i < reps; ++i) GL(glc_imdct_device(dec, F, 0, frames, d_blk));
    GL(glc_ctx_timer_end(dec, &ms));
    std::printf("D1 debug variant %d: %7.1f us per launch (%llu stereo frames)\n", v, ms / reps * 1e3, static_cast<unsigned long long>(frames));
    while (*p && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  GL(glc_debug_set_imdct_variant(dec, 0));
  {
    float ms = 0;
    for (int i = 0; i < 20; ++i) GL(glc_decode_device(dec, F, d_out, (frames + 1) * 1024 * ch, nullptr, nullptr));
    GL(glc_ctx_timer_begin(dec));
    for (int i = 0; i < reps; ++i) GL(glc_decode_device(dec, F, d_out, (frames + 1) * 1024 * ch, nullptr, nullptr));
    GL(glc_ctx_timer_end(dec, &ms));
    std::printf("glc_decode_device (D1 + D2), rows resident: %7.1f us per call\n", ms / reps * 1e3);
  }
  {  // a NEW stream every call on a warm context: row preparation + upload + kernels (what a decoder sees in service)
    std::vector<float> pcm2(pcm.begin() + 2048, pcm.end());
    pcm2.resize(pcm.size(), 0.0f);
    glc_frames *F2 = nullptr;
    GL(glc_encode(enc, pcm2.data(), n, ch, &F2));
    double best = 1e30;
    for (int i = 0; i < 20; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      GL(glc_decode_device(dec, (i & 1) ? F2 : F, d_out, (frames + 1) * 1024 * ch, nullptr, nullptr));
      GL(glc_ctx_synchronize(dec));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (i >= 2) best = std::min(best, ms);
    }
    std::printf("glc_decode_device of a stream the context has not seen (warm context): %.3f ms per call, synchronised\n", best);
    glc_frames_free(F2);
  }
  {  // host boundary: Decoder::decode from an EncodedAudio in host memory to PCM in host memory
    std::vector<float> out((frames + 1) * 1024 * ch);
    uint64_t n_out = 0;
    double best = 1e30, first = 0;
    glc_ctx *hd = nullptr;
    GL(glc_ctx_create(0, 48000, &hd));
    for (int i = 0; i < 12; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      GL(glc_decode(hd, F, out.data(), out.size(), &n_out));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (i == 0) first = ms;
      else best = std::min(best, ms);
    }
    std::printf("glc_decode, EncodedAudio in host memory -> %llu samples in host memory: first call %.3f ms (row preparation + upload), then %.3f ms\n",
                static_cast<unsigned long long>(n_out), first, best);
    glc_ctx_destroy(hd);
  }
  glc_frames_free(F);
  OK(hipFree(d_blk));
  OK(hipFree(d_out));
  glc_ctx_destroy(enc);
  glc_ctx_destroy(dec);
  return 0;
}
