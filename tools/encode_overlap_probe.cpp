// encode_overlap_probe.cpp — do transform launches on two streams overlap?  (sizing the opening rounds of
// glc_encode and the chunk alternation of glc_encode_range_device)  n launches of
// `frames` stereo frames each, device-resident, through two contexts' streams: all on one, alternating
// between the two, and as one call over the whole range.
// Build: make -C gapless-lossy-codec_amd/csrc tools      Usage: build/encode_overlap_probe [frames = 1024] [launches = 4]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "glc.h"
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1);} } while (0)
#define GL(x) do { int r_ = (x); if (r_ != 0) { std::printf("%s -> %d: %s\n", #x, r_, glc_last_error(nullptr)); std::exit(1);} } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const uint64_t piece = argc > 1 ? std::atoi(argv[1]) : 1024;  // frames per launch
  const int n_pieces = argc > 2 ? std::atoi(argv[2]) : 4;
  const uint16_t ch = 2;
  const uint64_t frames = piece * n_pieces, per_ch = frames * 1024, n = per_ch * ch;
  std::vector<float> pcm(n);
  for (uint64_t i = 0; i < n; ++i) pcm[i] = 0.3f * static_cast<float>((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f - 0.15f;
  glc_ctx *c[2];
  GL(glc_ctx_create(0, 48000, &c[0]));
  GL(glc_ctx_create(0, 48000, &c[1]));
  float *d_pcm; void *d_rec;
  OK(hipMalloc(&d_pcm, n * 4)); OK(hipMalloc(&d_rec, frames * glc_record_bytes(ch)));
  OK(hipMemcpy(d_pcm, pcm.data(), n * 4, hipMemcpyHostToDevice));
  uint8_t *rec = static_cast<uint8_t *>(d_rec);
  auto run = [&](bool two) {
    for (int i = 0; i < n_pieces; ++i) {
      glc_ctx *x = c[two ? (i & 1) : 0];
      GL(glc_encode_range_device(x, d_pcm, 0, per_ch, n, ch, i * piece, (i + 1) * piece, rec + i * piece * glc_record_bytes(ch), nullptr));
    }
    GL(glc_ctx_synchronize(c[0])); GL(glc_ctx_synchronize(c[1]));
  };
  for (int i = 0; i < 200; ++i) run(true);
  for (int rep = 0; rep < 2; ++rep)
    for (int two = 0; two < 2; ++two) {
      double best = 1e9;
      for (int i = 0; i < 30; ++i) { const double t0 = now_ms(); run(two); best = std::min(best, now_ms() - t0); }
      std::printf("%d launches of %llu stereo frames on %s: %.3f ms\n", n_pieces, (unsigned long long)piece, two ? "two streams alternately" : "one stream", best);
    }
  GL(glc_encode_range_device(c[0], d_pcm, 0, per_ch, n, ch, 0, frames, rec, nullptr)); GL(glc_ctx_synchronize(c[0]));
  double best = 1e9;
  for (int i = 0; i < 30; ++i) { const double t0 = now_ms(); GL(glc_encode_range_device(c[0], d_pcm, 0, per_ch, n, ch, 0, frames, rec, nullptr)); GL(glc_ctx_synchronize(c[0])); best = std::min(best, now_ms() - t0); }
  std::printf("one launch of all %llu frames: %.3f ms\n", (unsigned long long)frames, best);
  return 0;
}
