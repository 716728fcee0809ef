// glc_multi_gpu.cpp — the multi-GPU host of SURVEY 8(e) without any Python: ONE process, one
// glc_ctx per visible device, frame-range shards, and the single RCCL gather of the compact blobs
// to device 0.  This is the shape a Rust (or C / C++) host of the reference would copy: the
// library itself never owns a communicator (include/glc.h has no RCCL types), it hands out device
// pointers and byte counts and the host moves them.
//
//   encode (frames mode, default)   the rayon loop of Encoder::encode (src/codec.rs:462) split into
//       contiguous frame ranges, one per device; each device reads its own PCM slice + 1024-sample
//       halo (supplied at upload, no device exchange), writes fixed-size records, compacts them
//       (glc_compact_device_records) and sends the blob to device 0 (ncclSend / ncclRecv in one group
//       = a gather with per-rank sizes); the root copies the blobs to the host and assembles
//       EncodedAudio (glc_frames_from_compact).  The .glc bytes are compared with a single-device
//       glc_encode of the same stream.
//   decode (frames mode)            the un-trimmed output has n_frames + 1 hops; device r decodes a contiguous
//       hop range with glc_decode_range_device, the PCM pieces go to device 0 (same send/recv group)
//       and are compared bit for bit with the single-device decode.
//   --streams                       BASELINE config 4's other sharding: stream s <-> device s, every
//       device encodes a whole stream of its own, same gather, one EncodedAudio per stream.
//
// Build: make -C gapless-lossy-codec_amd/csrc tools   (hipcc, links libglc_hip.so and librccl.so)
// Usage: build/glc_multi_gpu [--streams] [--frames N (per device, default 4096)] [--devices D]
// With one visible device it degrades to a single shard (the collective is then empty) - that is
// what runs on a one-GPU development box; exit code 0 = bytes identical.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "glc.h"

#define HIP_OK(x)                                                                       \
  do {                                                                                  \
    hipError_t e__ = (x);                                                               \
    if (e__ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      std::exit(2);                                                                     \
    }                                                                                   \
  } while (0)
#define NCCL_OK(x)                                                                       \
  do {                                                                                   \
    ncclResult_t r__ = (x);                                                              \
    if (r__ != ncclSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #x, ncclGetErrorString(r__)); \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)
#define GLC_OK_(x, ctx)                                                                  \
  do {                                                                                   \
    int r__ = (x);                                                                       \
    if (r__ != GLC_OK) {                                                                 \
      std::fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, r__, glc_last_error(ctx)); \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)

namespace {

constexpr uint32_t kSampleRate = 48000;
constexpr uint16_t kChannels = 2;

// deterministic tonal test signal (per-channel chords), generated in f64 and rounded to f32
std::vector<float> make_pcm(uint64_t per_channel, unsigned seed) {
  std::vector<float> x(per_channel * kChannels);
  unsigned s = 12345u + 977u * seed;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return (s >> 8) * (1.0 / 16777216.0);
  };
  for (uint16_t c = 0; c < kChannels; ++c) {
    double f[12], p[12];
    for (int k = 0; k < 12; ++k) f[k] = 80.0 + 7900.0 * rnd(), p[k] = 6.283185307179586 * rnd();
    for (uint64_t t = 0; t < per_channel; ++t) {
      double v = 0;
      for (int k = 0; k < 12; ++k) v += 0.06 * std::sin(6.283185307179586 * f[k] * (double(t) / kSampleRate) + p[k]);
      x[t * kChannels + c] = static_cast<float>(v);
    }
  }
  return x;
}

struct Shard {
  uint64_t f0, f1, t0, t_count;
};

// contiguous balanced frame ranges + the PCM span (with halo) each one reads, clipped to the stream
std::vector<Shard> plan_shards(uint64_t n_frames, uint64_t per_channel, int world) {
  std::vector<Shard> out;
  const uint64_t base = n_frames / world, extra = n_frames % world;
  uint64_t f = 0;
  for (int r = 0; r < world; ++r) {
    const uint64_t n = base + (static_cast<uint64_t>(r) < extra ? 1 : 0);
    Shard s{f, f + n, 0, 0};
    if (n) {
      const uint64_t lo = s.f0 * GLC_HOP_SIZE > GLC_HOP_SIZE / 2 ? s.f0 * GLC_HOP_SIZE - GLC_HOP_SIZE / 2 : 0;
      uint64_t hi = (s.f1 - 1) * GLC_HOP_SIZE + GLC_FRAME_SIZE - GLC_HOP_SIZE / 2;
      if (hi > per_channel) hi = per_channel;
      s.t0 = lo;
      s.t_count = hi > lo ? hi - lo : 0;
    }
    out.push_back(s);
    f += n;
  }
  return out;
}

std::vector<uint8_t> serialize(const glc_frames *f) {
  std::vector<uint8_t> b(glc_serialized_size(f));
  uint64_t w = 0;
  GLC_OK_(glc_serialize(f, b.data(), b.size(), &w), nullptr);
  return b;
}

struct Dev {
  int id = 0;
  glc_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;  // the context's stream: encode, compaction and the send are ordered on it
  float *d_pcm = nullptr;
  void *d_rec = nullptr, *d_blob = nullptr;
  uint64_t blob_cap = 0;
  glc_compact_info info{};
};

}  // namespace

int main(int argc, char **argv) {
  bool streams = false;
  uint64_t frames_per_dev = 4096;
  int want_devices = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--streams") streams = true;
    else if (a == "--frames" && i + 1 < argc) frames_per_dev = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--devices" && i + 1 < argc) want_devices = std::atoi(argv[++i]);
    else {
      std::fprintf(stderr, "usage: %s [--streams] [--frames N] [--devices D]\n", argv[0]);
      return 2;
    }
  }
  int n_dev = 0;
  HIP_OK(hipGetDeviceCount(&n_dev));
  if (n_dev <= 0) {
    std::fprintf(stderr, "no HIP device: this tool (like the library) has no CPU path\n");
    return 2;
  }
  if (want_devices > 0 && want_devices < n_dev) n_dev = want_devices;
  const int world = n_dev;
  std::printf("glc_multi_gpu: %d device(s), %s mode, %llu frames per device, %u Hz x %u ch\n", world,
              streams ? "stream-per-device" : "frame-shard", static_cast<unsigned long long>(frames_per_dev),
              kSampleRate, kChannels);

  // ---- one context per device, one communicator per device (single process) -------------------
  std::vector<Dev> dev(world);
  std::vector<int> ids(world);
  for (int r = 0; r < world; ++r) ids[r] = r;
  std::vector<ncclComm_t> comm(world);
  NCCL_OK(ncclCommInitAll(comm.data(), world, ids.data()));
  for (int r = 0; r < world; ++r) {
    dev[r].id = r;
    GLC_OK_(glc_ctx_create(r, kSampleRate, &dev[r].ctx), nullptr);
    dev[r].stream = static_cast<hipStream_t>(glc_ctx_stream(dev[r].ctx));
  }

  // ---- the job ---------------------------------------------------------------------------------
  // frames mode: ONE stream of world * frames_per_dev frames; streams mode: `world` streams
  const uint64_t n_streams = streams ? world : 1;
  const uint64_t frames_per_stream = streams ? frames_per_dev : frames_per_dev * world;
  const uint64_t per_channel = frames_per_stream * GLC_HOP_SIZE;  // gives exactly that many frames
  const uint64_t n_samples = per_channel * kChannels;
  glc_plan plan;
  GLC_OK_(glc_plan_encode(n_samples, kChannels, &plan), nullptr);
  if (plan.n_frames != frames_per_stream) {
    std::fprintf(stderr, "internal: plan gives %llu frames\n", static_cast<unsigned long long>(plan.n_frames));
    return 2;
  }
  std::vector<std::vector<float>> pcm(n_streams);
  for (uint64_t s = 0; s < n_streams; ++s) pcm[s] = make_pcm(per_channel, static_cast<unsigned>(s));
  const std::vector<Shard> shards =
      streams ? std::vector<Shard>(world, Shard{0, plan.n_frames, 0, per_channel}) : plan_shards(plan.n_frames, per_channel, world);
  const uint64_t rec_bytes = glc_record_bytes(kChannels);

  for (int r = 0; r < world; ++r) {
    Dev &d = dev[r];
    const Shard &sh = shards[r];
    const std::vector<float> &src = pcm[streams ? r : 0];
    HIP_OK(hipSetDevice(d.id));
    const uint64_t nf = sh.f1 - sh.f0;
    d.blob_cap = glc_compact_bound(kChannels, nf);
    HIP_OK(hipMalloc(&d.d_pcm, std::max<uint64_t>(sh.t_count * kChannels, 1) * sizeof(float)));
    HIP_OK(hipMalloc(&d.d_rec, std::max<uint64_t>(nf * rec_bytes, 1)));
    HIP_OK(hipMalloc(&d.d_blob, d.blob_cap));
    // the shard's own PCM slice + halo goes up; nothing else of the stream is on this device
    HIP_OK(hipMemcpy(d.d_pcm, src.data() + sh.t0 * kChannels, sh.t_count * kChannels * sizeof(float), hipMemcpyHostToDevice));
  }

  const auto t_begin = std::chrono::steady_clock::now();
  // encode: queued on every device's stream, no host synchronisation in between
  for (int r = 0; r < world; ++r) {
    Dev &d = dev[r];
    const Shard &sh = shards[r];
    if (sh.f1 > sh.f0)
      GLC_OK_(glc_encode_range_device(d.ctx, d.d_pcm, sh.t0, sh.t_count, n_samples, kChannels, sh.f0, sh.f1, d.d_rec, nullptr),
              d.ctx);
  }
  // compact on every device (each call synchronises its own stream and returns the blob size)
  for (int r = 0; r < world; ++r) {
    Dev &d = dev[r];
    GLC_OK_(glc_compact_device_records(d.ctx, d.d_rec, shards[r].f1 - shards[r].f0, kChannels, d.d_blob, d.blob_cap, &d.info),
            d.ctx);
  }
  // the single collective: every other device sends its blob to device 0
  std::vector<uint64_t> off(world, 0);
  uint64_t root_bytes = 0;
  for (int r = 0; r < world; ++r) {
    off[r] = root_bytes;
    root_bytes += (dev[r].info.bytes + 255) & ~255ull;
  }
  uint8_t *d_root = nullptr;
  HIP_OK(hipSetDevice(dev[0].id));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_root), root_bytes));
  HIP_OK(hipMemcpyAsync(d_root, dev[0].d_blob, dev[0].info.bytes, hipMemcpyDeviceToDevice, dev[0].stream));
  uint64_t gathered = 0;
  NCCL_OK(ncclGroupStart());
  for (int r = 1; r < world; ++r) {
    // each call on the device its communicator and stream belong to
    HIP_OK(hipSetDevice(dev[r].id));
    NCCL_OK(ncclSend(dev[r].d_blob, dev[r].info.bytes, ncclUint8, 0, comm[r], dev[r].stream));
    HIP_OK(hipSetDevice(dev[0].id));
    NCCL_OK(ncclRecv(d_root + off[r], dev[r].info.bytes, ncclUint8, r, comm[0], dev[0].stream));
    gathered += dev[r].info.bytes;
  }
  NCCL_OK(ncclGroupEnd());
  // A first multi-rank failure must name its rank: wait for every rank's stream with a deadline
  // instead of blocking in one hipStreamSynchronize for ever.
  {
    const double limit_s = std::getenv("GLC_GATHER_TIMEOUT_S") ? std::atof(std::getenv("GLC_GATHER_TIMEOUT_S")) : 120.0;
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<char> done(world, 0);
    int left = world;
    while (left > 0) {
      for (int r = 0; r < world; ++r) {
        if (done[r]) continue;
        HIP_OK(hipSetDevice(dev[r].id));
        const hipError_t q = hipStreamQuery(dev[r].stream);
        if (q == hipSuccess) {
          done[r] = 1;
          --left;
        } else if (q != hipErrorNotReady) {
          std::fprintf(stderr, "glc_multi_gpu: rank %d (device %d): %s while gathering %llu bytes\n", r, dev[r].id,
                       hipGetErrorString(q), static_cast<unsigned long long>(dev[r].info.bytes));
          return 3;
        }
      }
      if (left > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > limit_s) {
        for (int r = 0; r < world; ++r)
          if (!done[r])
            std::fprintf(stderr, "glc_multi_gpu: rank %d (device %d) has not finished its %s of %llu bytes after %.0f s\n", r,
                         dev[r].id, r == 0 ? "receives" : "send", static_cast<unsigned long long>(dev[r].info.bytes), limit_s);
        return 3;
      }
    }
    HIP_OK(hipSetDevice(dev[0].id));
  }
  // root: blobs to the host (pinned), assemble
  uint8_t *h_root = nullptr;
  HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&h_root), root_bytes, hipHostMallocDefault));
  HIP_OK(hipMemcpyAsync(h_root, d_root, root_bytes, hipMemcpyDeviceToHost, dev[0].stream));
  HIP_OK(hipStreamSynchronize(dev[0].stream));
  for (int r = 1; r < world; ++r) {
    HIP_OK(hipSetDevice(dev[r].id));
    HIP_OK(hipStreamSynchronize(dev[r].stream));
  }
  std::vector<glc_frames *> result;
  if (streams) {
    for (int r = 0; r < world; ++r) {
      const void *b[1] = {h_root + off[r]};
      const uint64_t sz[1] = {dev[r].info.bytes};
      glc_frames *f = nullptr;
      GLC_OK_(glc_frames_from_compact(kSampleRate, n_samples, kChannels, b, sz, 1, &f), nullptr);
      result.push_back(f);
    }
  } else {
    std::vector<const void *> b(world);
    std::vector<uint64_t> sz(world);
    for (int r = 0; r < world; ++r) b[r] = h_root + off[r], sz[r] = dev[r].info.bytes;
    glc_frames *f = nullptr;
    GLC_OK_(glc_frames_from_compact(kSampleRate, n_samples, kChannels, b.data(), sz.data(), static_cast<uint32_t>(world), &f), nullptr);
    result.push_back(f);
  }
  const double job_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();

  // ---- check: the same streams through the single-device host call -------------------------------
  int bad = 0;
  uint64_t glc_total = 0;
  for (size_t s = 0; s < result.size(); ++s) {
    glc_frames *single = nullptr;
    GLC_OK_(glc_encode(dev[0].ctx, pcm[s].data(), n_samples, kChannels, &single), dev[0].ctx);
    const std::vector<uint8_t> a = serialize(result[s]), b = serialize(single);
    glc_total += a.size();
    if (a != b) {
      ++bad;
      std::fprintf(stderr, "stream %zu: gathered .glc (%zu bytes) differs from the single-device encode (%zu bytes)\n", s,
                   a.size(), b.size());
    }
    glc_frames_free(single);
    glc_frames_free(result[s]);
  }
  // ---- decode, sharded the same way (frames mode): device r decodes a contiguous range of the
  // n_frames + 1 output hops with glc_decode_range_device (the library recomputes the one halo frame
  // the overlap-add carries, src/codec.rs:701-705), the PCM pieces go to device 0 in rank order -
  // again one send/recv group - and must equal the single-device decode bit for bit.
  if (!streams) {
    glc_frames *whole = nullptr;
    GLC_OK_(glc_encode(dev[0].ctx, pcm[0].data(), n_samples, kChannels, &whole), dev[0].ctx);
    const uint64_t hops = plan.n_frames + 1, per_hop = uint64_t(GLC_HOP_SIZE) * kChannels;
    std::vector<uint64_t> h0(world + 1, 0);
    for (int r = 0; r < world; ++r) h0[r + 1] = h0[r] + hops / world + (uint64_t(r) < hops % world ? 1 : 0);
    float *d_all = nullptr, *d_ref = nullptr;
    HIP_OK(hipSetDevice(dev[0].id));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_all), hops * per_hop * sizeof(float)));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_ref), hops * per_hop * sizeof(float)));
    std::vector<float *> d_part(world, nullptr);
    for (int r = 0; r < world; ++r) {
      const uint64_t n = (h0[r + 1] - h0[r]) * per_hop;
      HIP_OK(hipSetDevice(dev[r].id));
      d_part[r] = r == 0 ? d_all : nullptr;
      if (r) HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_part[r]), std::max<uint64_t>(n, 1) * sizeof(float)));
      GLC_OK_(glc_decode_range_device(dev[r].ctx, whole, h0[r], h0[r + 1], d_part[r], n), dev[r].ctx);
    }
    NCCL_OK(ncclGroupStart());
    for (int r = 1; r < world; ++r) {
      const uint64_t n = (h0[r + 1] - h0[r]) * per_hop;
      NCCL_OK(ncclSend(d_part[r], n, ncclFloat, 0, comm[r], dev[r].stream));
      NCCL_OK(ncclRecv(d_all + h0[r] * per_hop, n, ncclFloat, r, comm[0], dev[0].stream));
    }
    NCCL_OK(ncclGroupEnd());
    for (int r = 0; r < world; ++r) {
      HIP_OK(hipSetDevice(dev[r].id));
      HIP_OK(hipStreamSynchronize(dev[r].stream));
    }
    uint64_t start = 0, n_out = 0;
    GLC_OK_(glc_decode_device(dev[0].ctx, whole, d_ref, hops * per_hop, &start, &n_out), dev[0].ctx);
    GLC_OK_(glc_ctx_synchronize(dev[0].ctx), dev[0].ctx);
    std::vector<float> a(hops * per_hop), b(hops * per_hop);
    HIP_OK(hipSetDevice(dev[0].id));
    HIP_OK(hipMemcpy(a.data(), d_all, a.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(b.data(), d_ref, b.size() * sizeof(float), hipMemcpyDeviceToHost));
    const bool same = std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
    std::printf("decode: %llu hops in %d shard(s), gathered PCM %s the single-device decode\n",
                static_cast<unsigned long long>(hops), world, same ? "is bit-identical to" : "DIFFERS from");
    if (!same) ++bad;
    for (int r = 1; r < world; ++r) {
      HIP_OK(hipSetDevice(dev[r].id));
      HIP_OK(hipFree(d_part[r]));
    }
    HIP_OK(hipSetDevice(dev[0].id));
    HIP_OK(hipFree(d_all));
    HIP_OK(hipFree(d_ref));
    glc_frames_free(whole);
  }
  const double msamples = double(n_samples) * n_streams / 1e6;
  std::printf("encode + compaction + gather + assembly: %.3f ms wall for %.1f Msamples (%.0f Msamples/s, first call: includes "
              "communicator warm-up)\n", job_ms, msamples, msamples / (job_ms * 1e-3));
  std::printf("gathered to device 0 over RCCL: %llu bytes in %d blob(s) (dense records would be %llu bytes); .glc total %llu bytes\n",
              static_cast<unsigned long long>(gathered), world - 1,
              static_cast<unsigned long long>(rec_bytes * frames_per_dev * (world - 1)), static_cast<unsigned long long>(glc_total));
  std::printf("%s\n", bad ? "MISMATCH" : "OK: every assembled stream is byte-identical to the single-device encode");

  HIP_OK(hipHostFree(h_root));
  HIP_OK(hipSetDevice(dev[0].id));
  HIP_OK(hipFree(d_root));
  for (int r = 0; r < world; ++r) {
    HIP_OK(hipSetDevice(dev[r].id));
    HIP_OK(hipFree(dev[r].d_pcm));
    HIP_OK(hipFree(dev[r].d_rec));
    HIP_OK(hipFree(dev[r].d_blob));
    glc_ctx_destroy(dev[r].ctx);
    NCCL_OK(ncclCommDestroy(comm[r]));
  }
  return bad ? 1 : 0;
}
