// h2d_probe.cpp — what does it cost to get a caller-owned (pageable) PCM buffer onto the device?
// Decides the staging strategy of glc_encode / glc_decode: plain hipMemcpy from pageable
// memory, hipHostRegister + async copy + unregister, or a copy through a pinned ring.
// Build: hipcc -O2 tools/h2d_probe.cpp -o build/h2d_probe        Usage: build/h2d_probe [MiB = 32]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define OK(x)                                                                          \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      std::printf("%s: %s\n", #x, hipGetErrorString(e));                               \
      std::exit(1);                                                                    \
    }                                                                                  \
  } while (0)

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
  const size_t mib = argc > 1 ? std::atoi(argv[1]) : 32;
  const size_t n = mib << 20;
  void *d = nullptr;
  OK(hipMalloc(&d, n));
  std::vector<char> page(n, 1);  // pageable, touched
  void *pin = nullptr;
  OK(hipHostMalloc(&pin, n, hipHostMallocDefault));
  std::memset(pin, 2, n);
  hipStream_t s;
  OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  size_t bytes_moved = n;  // what the GB/s column divides by
  auto best = [&](const char *name, auto fn) {
    double b = 1e30;
    for (int i = 0; i < 7; ++i) {
      const double t0 = now_ms();
      fn();
      const double t = now_ms() - t0;
      if (t < b) b = t;
    }
    std::printf("%-58s %8.3f ms  %7.1f GB/s\n", name, b, bytes_moved / (b * 1e-3) / 1e9);
  };
  best("H2D hipMemcpy from pageable", [&] { OK(hipMemcpy(d, page.data(), n, hipMemcpyHostToDevice)); });
  best("H2D hipMemcpyAsync from pageable + sync", [&] {
    OK(hipMemcpyAsync(d, page.data(), n, hipMemcpyHostToDevice, s));
    OK(hipStreamSynchronize(s));
  });
  best("H2D hipMemcpyAsync from pinned + sync", [&] {
    OK(hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, s));
    OK(hipStreamSynchronize(s));
  });
  best("hipHostRegister + H2D async + sync + hipHostUnregister", [&] {
    OK(hipHostRegister(page.data(), n, hipHostRegisterDefault));
    OK(hipMemcpyAsync(d, page.data(), n, hipMemcpyHostToDevice, s));
    OK(hipStreamSynchronize(s));
    OK(hipHostUnregister(page.data()));
  });
  best("  hipHostRegister alone (+ unregister)", [&] {
    OK(hipHostRegister(page.data(), n, hipHostRegisterDefault));
    OK(hipHostUnregister(page.data()));
  });
  best("memcpy pageable -> pinned (1 thread)", [&] { std::memcpy(pin, page.data(), n); });
  for (int nt : {2, 4, 8}) {
    char name[96];
    std::snprintf(name, sizeof name, "memcpy pageable -> pinned (%d threads)", nt);
    best(name, [&] {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
          const size_t a = n * t / nt, b = n * (t + 1) / nt;
          std::memcpy(static_cast<char *>(pin) + a, page.data() + a, b - a);
        });
      for (auto &x : th) x.join();
    });
  }
  best("staged: 4 chunks memcpy(4 thr) -> pinned -> H2D async, pipelined", [&] {
    const int chunks = 4, nt = 4;
    for (int c = 0; c < chunks; ++c) {
      const size_t a = n * c / chunks, b = n * (c + 1) / chunks;
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
          const size_t x = a + (b - a) * t / nt, y = a + (b - a) * (t + 1) / nt;
          std::memcpy(static_cast<char *>(pin) + x, page.data() + x, y - x);
        });
      for (auto &x : th) x.join();
      OK(hipMemcpyAsync(static_cast<char *>(d) + a, static_cast<char *>(pin) + a, b - a, hipMemcpyHostToDevice, s));
    }
    OK(hipStreamSynchronize(s));
  });
  // The figures above are best-of-7 on ONE buffer: the runtime remembers the pages it has pinned for a
  // copy (and a registration), so repeats are cheap.  A caller that hands over a new buffer for every
  // file pays for the pinning once per buffer:
  {
    double t_copy = 1e30, t_reg = 1e30;
    for (int i = 0; i < 5; ++i) {
      std::vector<char> fresh(n, static_cast<char>(i + 3));  // touched, never seen by the runtime
      double t0 = now_ms();
      OK(hipMemcpy(d, fresh.data(), n, hipMemcpyHostToDevice));
      t_copy = std::min(t_copy, now_ms() - t0);
      std::vector<char> fresh2(n, static_cast<char>(i + 5));
      t0 = now_ms();
      OK(hipHostRegister(fresh2.data(), n, hipHostRegisterDefault));
      t_reg = std::min(t_reg, now_ms() - t0);
      OK(hipHostUnregister(fresh2.data()));
    }
    std::printf("%-58s %8.3f ms  %7.1f GB/s\n", "H2D hipMemcpy from a pageable buffer never seen before", t_copy, n / (t_copy * 1e-3) / 1e9);
    std::printf("%-58s %8.3f ms\n", "hipHostRegister of a buffer never seen before", t_reg);
  }
  bytes_moved = 4 << 20;
  best("D2H hipMemcpy to pageable (4 MiB)", [&] { OK(hipMemcpy(page.data(), d, 4 << 20, hipMemcpyDeviceToHost)); });
  best("D2H hipMemcpyAsync to pinned + sync (4 MiB)", [&] {
    OK(hipMemcpyAsync(pin, d, 4 << 20, hipMemcpyDeviceToHost, s));
    OK(hipStreamSynchronize(s));
  });
  return 0;
}
