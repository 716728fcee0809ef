// encode_throughput.cpp — host-boundary THROUGHPUT of glc_encode when several calls are in flight.
// A context is one call at a time (like `&mut self`), but distinct contexts run concurrently: T host
// threads, each with its own context and its own copy of the BASELINE config-2 batch, encode in a loop;
// the aggregate rate says how much of one call's serial tail (last round's transform, compaction,
// download) other calls' uploads can hide.  PCIe moves 4 B per sample: ~14 G samples/s at 56 GB/s.
// Build: make -C gapless-lossy-codec_amd/csrc tools      Usage: build/encode_throughput [threads = 3] [frames = 4096] [ch = 2] [calls = 200]
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "glc.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int max_threads = argc > 1 ? std::atoi(argv[1]) : 3;
  const uint64_t frames = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 4096;
  const uint16_t ch = argc > 3 ? static_cast<uint16_t>(std::atoi(argv[3])) : 2;
  const int calls = argc > 4 ? std::atoi(argv[4]) : 200;
  const uint64_t per_ch = frames * 1024, n = per_ch * ch;
  std::vector<float> base(n);
  for (uint64_t t = 0; t < per_ch; ++t)
    for (uint16_t c = 0; c < ch; ++c) {
      double v = 0;
      for (int h = 0; h < 16; ++h) v += std::sin(2 * M_PI * (110.0 * (h + 1) + 7 * c) * t / 48000.0 + h) / 16;
      base[t * ch + c] = static_cast<float>(0.7 * v);
    }
  for (int T = 1; T <= max_threads; ++T) {
    std::vector<std::vector<float>> pcm(T, base);  // a buffer per thread, as separate files would be
    std::vector<glc_ctx *> ctx(T, nullptr);
    for (int t = 0; t < T; ++t)
      if (glc_ctx_create(0, 48000, &ctx[t]) != 0) return std::printf("glc_ctx_create: %s\n", glc_last_error(nullptr)), 1;
    std::atomic<int> failed{0};
    auto work = [&](int t, int reps) {
      for (int i = 0; i < reps; ++i) {
        glc_frames *F = nullptr;
        if (glc_encode(ctx[t], pcm[t].data(), n, ch, &F) != 0) failed = 1;
        glc_frames_free(F);
      }
    };
    {  // warm: clocks, buffers, helper threads
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back(work, t, 30);
      for (auto &x : th) x.join();
    }
    const double t0 = now_s();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t, calls);
    for (auto &x : th) x.join();
    const double dt = now_s() - t0;
    std::printf("%d context(s) x %d calls of %llu frames x %u ch: %.3f ms per call in flight, %.3f ms per call aggregate = %.0f Msamples/s%s\n", T, calls,
                (unsigned long long)frames, ch, dt / calls * 1e3, dt / (calls * T) * 1e3, double(n) * calls * T / dt / 1e6,
                failed ? "  (FAILED calls!)" : "");
    for (int t = 0; t < T; ++t) glc_ctx_destroy(ctx[t]);
  }
  return 0;
}
