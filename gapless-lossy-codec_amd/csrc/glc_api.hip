// glc_api.hip — context, encode/decode drivers and the extern "C" surface (include/glc.h).
//
// Host-side counterpart of Encoder::new / encode (src/codec.rs:406-565) and Decoder::new /
// decode_streaming / decode (src/codec.rs:581-768).  There is no CPU compute path in this
// library: every entry point that transforms audio launches the gfx950 kernels and fails with
// GLC_ENODEV / GLC_EHIP when that is impossible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <cstdlib>
#include <vector>

#include "../../include/glc_debug.h"
#include "glc_common.h"
#include "glc_kernels.h"

namespace glc {

static std::mutex g_err_mu;
static std::string g_err;

void set_global_error(const std::string &msg) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_err = msg;
}

}  // namespace glc

// Device buffer that grows on demand (never shrinks while the context lives).
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Pinned host staging buffer (grows on demand): device <-> host copies through it run at PCIe
// speed and truly asynchronously, which pageable memory does not give.
struct HostBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// A parked host thread that runs one job at a time (glc_encode's uploader and launcher): started on
// first use and kept for the life of the context, because a fresh std::thread costs ~50 us before its
// first HIP call returns (thread start + the runtime's per-thread state) - 5 % of a config-2 call.
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, idle = true, quit = false;
  void submit(std::function<void()> j) {  // may throw std::system_error / std::bad_alloc on first use
    if (!th.joinable())
      th = std::thread([this] {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
          cv.wait(lk, [&] { return has_job || quit; });
          if (quit) return;
          std::function<void()> j2 = std::move(job);
          has_job = false;
          lk.unlock();
          j2();  // jobs do not throw (glc_encode wraps them)
          lk.lock();
          idle = true;
          cv.notify_all();
        }
      });
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j);
    has_job = true;
    idle = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return idle; });
  }
  ~Worker() {
    if (!th.joinable()) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      cv.notify_all();
    }
    th.join();
  }
};

struct glc_ctx {
  int device = 0;
  uint32_t sample_rate = 0;
  hipStream_t stream = nullptr;      // stream in use
  hipStream_t own_stream = nullptr;  // the context's private stream
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;  // glc_ctx_timer_*
  hipStream_t copy_stream = nullptr;  // glc_encode: uploads run ahead of the kernels on this one
  hipEvent_t ev_copy = nullptr;
  hipStream_t stream_b = nullptr;     // odd rounds / chunks of an encode are transformed here, beside the even ones on `stream`
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // glc_encode_range_device: second stream after the caller's work, caller's stream after both
  DevBuf coef_b;                      // ... with a coefficient workspace of their own
  hipStream_t down_stream = nullptr;  // glc_encode: round i is compacted and its blob comes down while round i+1 is transformed
  std::vector<hipEvent_t> ev_round;   // one per round: records written
  std::unique_ptr<Worker> enc_up, enc_launch;  // glc_encode's helper threads (multi-round streams only)
  glc::HostTables host;
  glc::DeviceTables dev{};
  DevBuf tables;     // all constant tables in one allocation
  DevBuf coef;       // MDCT coefficient workspace [rows][1024]
  DevBuf pcm;        // staging for host-boundary encode / decode output
  DevBuf records;    // staging for host-boundary encode
  DevBuf blocks;     // decode: windowed IMDCT blocks [(chunk+1)][ch][2048]
  DevBuf dec_meta;   // decode: pairs / offsets / scales / raw pool
  DevBuf dec_plan;   // decode: union records of the (group, channel) units of one D1 launch (+ the unit order)
  uint32_t dec_plan_groups = 0;
  // the launch whose plan records dec_plan still holds: a repeated launch of the same rows of the same
  // stream skips k_imdct_plan (valid only for launches of one batch, plan_M != 0)
  uint64_t plan_uid = 0;
  uint32_t plan_row_begin = 0, plan_M = 0;
  DevBuf pack_meta;  // compaction scratch: loc, blk, blk_raw, totals
  DevBuf pack_blob;  // compaction: the compact blob of glc_encode / glc_frames_from_device_records
  HostBuf host_stage;  // pinned: the blob on its way to the host
  std::string err;
  // decode session (decode_prepare / round_launch): device-resident sparse rows + position
  glc::DecodeRows dec_rows{};
  uint64_t dec_uid = 0;  // glc_frames::uid whose rows dec_meta holds (0: none)
  // header of that stream (what glc_decode needs besides the rows) and a fingerprint of its pools: a
  // caller-supplied stream id that comes back with different counts is treated as a new stream
  uint32_t dec_delay = 0;
  uint64_t dec_orig_len = 0, dec_n_pairs = 0, dec_n_raw = 0;
  int d1_variant = 0;    // include/glc_debug.h: which inverse-transform kernel / path to launch
  int k1_variant = 0;    // include/glc_debug.h: which forward-transform kernel takes launches of >= 4096 rows
  hipStream_t probe_stream = nullptr;  // include/glc_debug.h clock probe
  HostBuf probe_out;
  uint32_t dec_ch = 0;
  uint64_t dec_frames = 0, dec_next = 0;
  bool stream_open = false;  // glc_decode_stream_begin called, last chunk not yet delivered
  // streaming session: chunk i is copied to the host while chunk i+1 is already being decoded
  DevBuf stream_out;                           // two chunk-sized output buffers
  hipEvent_t ev_dec[2] = {nullptr, nullptr};   // "kernels of the chunk in buffer b are done"
  int stream_buf = 0;                          // buffer holding the chunk the next call delivers
  uint64_t stream_frames = 0;                  // its frame count
  bool stream_last = false;                    // ... and whether it is the last one (carries the tail)
};

namespace {

constexpr uint64_t kEncodeChunkFrames = 4096;  // frames per K1/K2/K3 round: coef stays MALL-sized
// ... of at least 8192 rows: a 128 x 128-tile launch of 4096 rows is one workgroup per CU, two waves per
// SIMD (mono, k1_tune: 27.1 T MAC/s at 4096 rows, 30.8 T at 8192)
inline uint64_t encode_chunk_frames(uint32_t ch) { return ch == 1 ? 2 * kEncodeChunkFrames : kEncodeChunkFrames; }
constexpr uint64_t kDecodeChunkFrames = 4096;

constexpr uint32_t kPlanGroups = 2048;  // (group, channel) units per D1 batch: 135 MB of workspace

// glc_encode's helper threads run stages that report through fail(): they must not write ctx->err
// (one std::string, two writers) - each helper points this at a string of its own for the duration of
// its job, and the calling thread stores the winning message into ctx->err once, at the end.
thread_local std::string *t_err_sink = nullptr;

int fail(glc_ctx *ctx, int code, const std::string &msg) {
  if (t_err_sink) {
    *t_err_sink = msg;
    return code;
  }
  if (ctx) ctx->err = msg;
  glc::set_global_error(msg);
  return code;
}

int hip_fail(glc_ctx *ctx, hipError_t e, const char *what) {
  return fail(ctx, e == hipErrorOutOfMemory ? GLC_ENOMEM : GLC_EHIP,
              std::string(what) + ": " + hipGetErrorString(e));
}

#define GLC_HIP(ctx, call)                                   \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) return hip_fail(ctx, e__, #call); \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

const char *glc_last_error(const glc_ctx *ctx) {
  if (ctx) return ctx->err.c_str();
  static thread_local std::string copy;
  {
    std::lock_guard<std::mutex> lk(glc::g_err_mu);
    copy = glc::g_err;
  }
  return copy.c_str();
}

int glc_ctx_create(int device, uint32_t sample_rate, glc_ctx **out) {
  if (!out) return GLC_EINVAL;
  *out = nullptr;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    return fail(nullptr, GLC_ENODEV,
                std::string("glc_ctx_create: no HIP device (") + hipGetErrorString(e) +
                    "); this library has no CPU fallback");
  if (device < 0 || device >= n_dev)
    return fail(nullptr, GLC_EINVAL, "glc_ctx_create: device index out of range");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(nullptr, GLC_ENODEV,
                std::string("glc_ctx_create: device is ") + prop.gcnArchName +
                    ", kernels are built for gfx950 only");

  std::unique_ptr<glc_ctx> ctx(new (std::nothrow) glc_ctx);
  if (!ctx) return GLC_ENOMEM;
  ctx->device = device;
  ctx->sample_rate = sample_rate;
  glc::build_host_tables(sample_rate, ctx->host);

  DeviceGuard guard(device);
  GLC_HIP(nullptr, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
  ctx->stream = ctx->own_stream;

  // one allocation, 256-B aligned sub-buffers
  const glc::HostTables &h = ctx->host;
  const size_t nb = h.edges.size() - 1;
  size_t off = 0;
  auto place = [&](size_t bytes) {
    size_t at = off;
    off = align_up(off + bytes, 256);
    return at;
  };
  const size_t o_cos_t = place(h.cos_table_t.size() * 4);
  const size_t o_cos = place(h.cos_table.size() * 4);
  const size_t o_win = place(h.window.size() * 4);
  const size_t o_indiv = place(h.indiv.size() * 4);
  const size_t o_pf = place(64 * 4);
  const size_t o_len = place(64 * 4);
  const size_t o_bof = place(h.band_of.size() * 2);
  const size_t o_edges = place(65 * 4);
  e = ctx->tables.reserve(off);
  if (e != hipSuccess) {
    (void)hipStreamDestroy(ctx->own_stream);
    return hip_fail(nullptr, e, "hipMalloc(tables)");
  }
  uint8_t *base = static_cast<uint8_t *>(ctx->tables.p);
  auto up = [&](size_t o, const void *src, size_t bytes) {
    return hipMemcpy(base + o, src, bytes, hipMemcpyHostToDevice);
  };
  std::vector<float> pf(64, 0.f), len(64, 1.f);
  std::vector<uint32_t> edges(65, glc::kHop);
  std::copy(h.band_pf.begin(), h.band_pf.end(), pf.begin());
  std::copy(h.band_len.begin(), h.band_len.end(), len.begin());
  std::copy(h.edges.begin(), h.edges.end(), edges.begin());
  hipError_t es[8] = {up(o_cos_t, h.cos_table_t.data(), h.cos_table_t.size() * 4),
                      up(o_cos, h.cos_table.data(), h.cos_table.size() * 4),
                      up(o_win, h.window.data(), h.window.size() * 4),
                      up(o_indiv, h.indiv.data(), h.indiv.size() * 4),
                      up(o_pf, pf.data(), 64 * 4),
                      up(o_len, len.data(), 64 * 4),
                      up(o_bof, h.band_of.data(), h.band_of.size() * 2),
                      up(o_edges, edges.data(), 65 * 4)};
  for (hipError_t x : es)
    if (x != hipSuccess) {
      ctx->tables.release();
      (void)hipStreamDestroy(ctx->own_stream);
      return hip_fail(nullptr, x, "hipMemcpy(tables)");
    }
  glc::DeviceTables &d = ctx->dev;
  d.cos_t = reinterpret_cast<const float *>(base + o_cos_t);
  d.cos = reinterpret_cast<const float *>(base + o_cos);
  d.window = reinterpret_cast<const float *>(base + o_win);
  d.indiv = reinterpret_cast<const float *>(base + o_indiv);
  d.band_pf = reinterpret_cast<const float *>(base + o_pf);
  d.band_len = reinterpret_cast<const float *>(base + o_len);
  d.band_of = reinterpret_cast<const uint16_t *>(base + o_bof);
  d.edges = reinterpret_cast<const uint32_t *>(base + o_edges);
  d.n_bands = static_cast<uint32_t>(nb);
  d.norm = h.norm;
  d.cf = h.cf;
  d.noise_floor = h.noise_floor;
  *out = ctx.release();
  return GLC_OK;
}

void glc_ctx_destroy(glc_ctx *ctx) {
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  ctx->enc_up.reset();      // joins the helper threads (idle: glc_encode waits for them before it returns)
  ctx->enc_launch.reset();
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
  if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
  if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  for (hipEvent_t e : ctx->ev_dec)
    if (e) (void)hipEventDestroy(e);
  ctx->stream_out.release();
  for (hipEvent_t e : ctx->ev_round)
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream_b) (void)hipStreamDestroy(ctx->stream_b);
  if (ctx->probe_stream) (void)hipStreamDestroy(ctx->probe_stream);
  ctx->probe_out.release();
  ctx->coef_b.release();
  if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  ctx->tables.release();
  ctx->coef.release();
  ctx->pcm.release();
  ctx->records.release();
  ctx->blocks.release();
  ctx->dec_meta.release();
  ctx->dec_plan.release();
  ctx->pack_meta.release();
  ctx->pack_blob.release();
  ctx->host_stage.release();
  delete ctx;
}

void *glc_ctx_stream(glc_ctx *ctx) { return ctx ? ctx->stream : nullptr; }
int glc_ctx_device(const glc_ctx *ctx) { return ctx ? ctx->device : -1; }

int glc_ctx_set_stream(glc_ctx *ctx, void *hip_stream) {
  if (!ctx) return GLC_EINVAL;
  DeviceGuard guard(ctx->device);
  GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return GLC_OK;
}

int glc_ctx_synchronize(glc_ctx *ctx) {
  if (!ctx) return GLC_EINVAL;
  DeviceGuard guard(ctx->device);
  GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GLC_OK;
}

int glc_ctx_timer_begin(glc_ctx *ctx) {
  if (!ctx) return GLC_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!ctx->ev_begin) GLC_HIP(ctx, hipEventCreate(&ctx->ev_begin));
  if (!ctx->ev_end) GLC_HIP(ctx, hipEventCreate(&ctx->ev_end));
  GLC_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
  return GLC_OK;
}

int glc_ctx_timer_end(glc_ctx *ctx, float *elapsed_ms) {
  if (!ctx || !elapsed_ms || !ctx->ev_begin) return fail(ctx, GLC_EINVAL, "glc_ctx_timer_end: no timer running");
  DeviceGuard guard(ctx->device);
  GLC_HIP(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
  GLC_HIP(ctx, hipEventSynchronize(ctx->ev_end));
  GLC_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_begin, ctx->ev_end));
  return GLC_OK;
}

int glc_ctx_tables(const glc_ctx *ctx, float *cos_table, float *window, float *norm,
                   float *weights, uint32_t *band_edges, uint32_t *n_edges) {
  if (!ctx) return GLC_EINVAL;
  const glc::HostTables &h = ctx->host;
  if (cos_table) std::memcpy(cos_table, h.cos_table.data(), h.cos_table.size() * 4);
  if (window) std::memcpy(window, h.window.data(), h.window.size() * 4);
  if (norm) *norm = h.norm;
  if (weights) std::memcpy(weights, h.weights.data(), h.weights.size() * 4);
  if (band_edges) std::memcpy(band_edges, h.edges.data(), h.edges.size() * 4);
  if (n_edges) *n_edges = static_cast<uint32_t>(h.edges.size());
  return GLC_OK;
}

// ------------------------------------------------------------------------------ encode

// glc_encode_range_device on a given stream with a given coefficient workspace (glc_encode runs
// alternate rounds on two streams)
static int encode_range_on(glc_ctx *ctx, hipStream_t stream, DevBuf &coef_ws, const float *d_pcm, uint64_t t0, uint64_t t_count,
                           uint64_t n_samples, uint16_t channels, uint64_t frame_begin, uint64_t frame_end,
                           void *d_records, float *d_coeffs, bool alternate_ok = false, bool beside = false) {
  if (!ctx || !d_pcm || !d_records) return fail(ctx, GLC_EINVAL, "glc_encode_range_device: null argument");
  const glc_plan plan = glc::plan_encode(n_samples, channels);
  if (plan.n_frames == 0)
    return fail(ctx, GLC_EINVAL, "glc_encode_range_device: the reference encoder panics on this input");
  if (frame_begin > frame_end || frame_end > plan.n_frames)
    return fail(ctx, GLC_EINVAL, "glc_encode_range_device: frame range out of bounds");
  const uint32_t ch = channels;
  if (d_coeffs && (frame_end - frame_begin) * ch > 0xFFFFFFFFull)
    return fail(ctx, GLC_EINVAL, "glc_encode_range_device: range too large for one coefficient tap");
  // The shard must hold every real sample the frame range reads:
  // per-channel t in [1024*f0 - 512, 1024*(f1-1) - 512 + 2048) clipped to the stream.
  if (frame_end > frame_begin) {
    const int64_t need_lo = std::max<int64_t>(0, static_cast<int64_t>(frame_begin) * glc::kHop - glc::kHop / 2);
    const int64_t stream_len = static_cast<int64_t>(plan.per_channel);
    const int64_t need_hi = std::min<int64_t>(stream_len, static_cast<int64_t>(frame_end - 1) * glc::kHop - glc::kHop / 2 + glc::kFrame);
    if (need_hi > need_lo &&
        (static_cast<int64_t>(t0) > need_lo || static_cast<int64_t>(t0 + t_count) < need_hi))
      return fail(ctx, GLC_EINVAL, "glc_encode_range_device: PCM shard does not cover the frame range (halo missing)");
  }
  DeviceGuard guard(ctx->device);
  const uint64_t rec = glc::record_bytes(ch);
  glc::PcmView view{d_pcm, t0, t_count, n_samples, ch};
  uint8_t *recs = static_cast<uint8_t *>(d_records);
  const uint64_t chunk = d_coeffs ? (frame_end - frame_begin ? frame_end - frame_begin : 1) : encode_chunk_frames(ch);
  if (!d_coeffs) {
    const uint64_t rows = std::min<uint64_t>(chunk, frame_end - frame_begin) * ch;
    GLC_HIP(ctx, coef_ws.reserve(std::max<size_t>(rows, 1) * glc::kHop * sizeof(float)));
  }
  // A range of several chunks alternates between the caller's stream and a second one (its own
  // coefficient workspace): chunk c's quantiser then runs beside chunk c+1's transform instead of in
  // front of it (six 4096-frame chunks: 3.78 -> 3.68 ms).  Fork and join by events, so the caller sees
  // plain stream order: everything queued before the call is finished before the second stream
  // starts, and the caller's stream continues only when both are done.
  const bool alternate = alternate_ok && !d_coeffs && frame_end - frame_begin > chunk;
  if (alternate) {
    if (!ctx->stream_b) GLC_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_b, hipStreamNonBlocking));
    if (!ctx->ev_fork) GLC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    if (!ctx->ev_join) GLC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    GLC_HIP(ctx, ctx->coef_b.reserve(static_cast<size_t>(chunk) * ch * glc::kHop * sizeof(float)));
    GLC_HIP(ctx, hipEventRecord(ctx->ev_fork, stream));
    GLC_HIP(ctx, hipStreamWaitEvent(ctx->stream_b, ctx->ev_fork, 0));
  }
  uint64_t c_idx = 0;
  for (uint64_t f = frame_begin; f < frame_end; f += chunk, ++c_idx) {
    const uint64_t nf = std::min<uint64_t>(chunk, frame_end - f);
    const uint32_t M = static_cast<uint32_t>(nf * ch);
    const bool odd = alternate && (c_idx & 1);
    hipStream_t st = odd ? ctx->stream_b : stream;
    float *coef = d_coeffs ? d_coeffs + (f - frame_begin) * ch * glc::kHop : static_cast<float *>(odd ? ctx->coef_b.p : coef_ws.p);
    uint8_t *r = recs + (f - frame_begin) * rec;
    GLC_HIP(ctx, glc::launch_mdct_forward(ctx->dev, view, f, M, coef, st, ctx->k1_variant, beside));
    bool decided = false;
    GLC_HIP(ctx, glc::launch_quantize(ctx->dev, coef, M, ch, view, f, r, st, &decided));
    if (!decided) GLC_HIP(ctx, glc::launch_decide_raw(ctx->dev, view, f, static_cast<uint32_t>(nf), r, st));
  }
  if (alternate) {
    GLC_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream_b));
    GLC_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_join, 0));
  }
  return GLC_OK;
}

int glc_encode_range_device(glc_ctx *ctx, const float *d_pcm, uint64_t t0, uint64_t t_count,
                            uint64_t n_samples, uint16_t channels, uint64_t frame_begin,
                            uint64_t frame_end, void *d_records, float *d_coeffs) {
  if (!ctx) return GLC_EINVAL;
  return encode_range_on(ctx, ctx->stream, ctx->coef, d_pcm, t0, t_count, n_samples, channels, frame_begin, frame_end,
                         d_records, d_coeffs, /*alternate_ok=*/true);
}

int glc_mdct_forward_device(glc_ctx *ctx, const float *d_pcm, uint64_t t0, uint64_t t_count,
                            uint64_t n_samples, uint16_t channels, uint64_t frame_begin,
                            uint64_t frame_end, float *d_coeffs) {
  if (!ctx || !d_pcm || !d_coeffs) return fail(ctx, GLC_EINVAL, "glc_mdct_forward_device: null argument");
  const glc_plan plan = glc::plan_encode(n_samples, channels);
  if (plan.n_frames == 0 || frame_begin > frame_end || frame_end > plan.n_frames)
    return fail(ctx, GLC_EINVAL, "glc_mdct_forward_device: bad stream length or frame range");
  const uint64_t rows = (frame_end - frame_begin) * channels;
  if (rows > 0xFFFFFFFFull) return fail(ctx, GLC_EINVAL, "glc_mdct_forward_device: range too large");
  DeviceGuard guard(ctx->device);
  glc::PcmView view{d_pcm, t0, t_count, n_samples, channels};
  GLC_HIP(ctx, glc::launch_mdct_forward(ctx->dev, view, frame_begin, static_cast<uint32_t>(rows), d_coeffs,
                                        ctx->stream, ctx->k1_variant));
  return GLC_OK;
}

// Queue the compaction of `n_frames` records into `d_blob` (capacity checked by the caller) on
// the context's stream; nothing is synchronised.
// scratch of one compaction of M rows: loc[M] | blk[nblk] | blk_raw[nblk] | totals, each 256-byte aligned
static size_t compact_scratch_bytes(uint64_t M) {
  const size_t nblk = (static_cast<size_t>(M) + 1023) / 1024;
  return align_up(static_cast<size_t>(M) * 4, 256) + 2 * align_up(nblk * 8, 256) + 256;
}

static int compact_launch(glc_ctx *ctx, const void *d_records, uint64_t n_frames, uint32_t ch, void *d_blob,
                          hipStream_t stream) {
  DevBuf &pm = ctx->pack_meta;
  const uint64_t M64 = n_frames * ch;
  if (M64 > 0xFFFFFFFFull) return fail(ctx, GLC_EINVAL, "compaction: frame range too long");
  const uint32_t M = static_cast<uint32_t>(M64);
  const glc::CompactLayout l = glc::compact_layout(ch, n_frames);
  const size_t nblk = (static_cast<size_t>(M) + 1023) / 1024;
  size_t off = 0;
  auto place = [&](size_t bytes) {
    size_t at = off;
    off = align_up(off + bytes, 256);
    return at;
  };
  const size_t o_loc = place(static_cast<size_t>(M) * 4), o_blk = place(nblk * 8), o_blkr = place(nblk * 8);
  const size_t o_tot = place(16);
  GLC_HIP(ctx, pm.reserve(std::max<size_t>(off, compact_scratch_bytes(M))));
  uint8_t *mb = static_cast<uint8_t *>(pm.p);
  uint8_t *blob = static_cast<uint8_t *>(d_blob);
  GLC_HIP(ctx, hipMemsetAsync(blob, 0, l.o_pairs, stream));  // header + section padding: deterministic bytes
  GLC_HIP(ctx, glc::launch_compact(static_cast<const uint8_t *>(d_records), M, ch, n_frames,
                                   reinterpret_cast<uint32_t *>(mb + o_loc), reinterpret_cast<uint64_t *>(mb + o_blk),
                                   reinterpret_cast<uint64_t *>(mb + o_blkr), reinterpret_cast<uint64_t *>(mb + o_tot),
                                   blob, l.o_israw, l.o_scale, l.o_cnt, l.o_pairs, stream));
  return GLC_OK;
}

int glc_compact_device_records(glc_ctx *ctx, const void *d_records, uint64_t n_frames, uint16_t channels,
                               void *d_blob, uint64_t cap, glc_compact_info *info) {
  if (!ctx || !d_blob || !info || (!d_records && n_frames)) return fail(ctx, GLC_EINVAL, "glc_compact_device_records: null argument");
  if (channels == 0) return fail(ctx, GLC_EINVAL, "glc_compact_device_records: channels == 0");
  const glc::CompactLayout l = glc::compact_layout(channels, n_frames);
  if (cap < l.bound) return fail(ctx, GLC_EINVAL, "glc_compact_device_records: blob buffer smaller than glc_compact_bound()");
  DeviceGuard guard(ctx->device);
  int rc = compact_launch(ctx, d_records, n_frames, channels, d_blob, ctx->stream);
  if (rc != GLC_OK) return rc;
  glc::CompactHeader h;
  GLC_HIP(ctx, hipMemcpyAsync(&h, d_blob, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h.magic != glc::kCompactMagic || h.bytes > l.bound)
    return fail(ctx, GLC_EHIP, "glc_compact_device_records: the device wrote an inconsistent header");
  info->n_frames = h.n_frames;
  info->n_pairs = h.n_pairs;
  info->n_raw_rows = h.n_raw_rows;
  info->bytes = h.bytes;
  return GLC_OK;
}

int glc_frames_from_device_records(glc_ctx *ctx, const void *d_records, uint64_t n_frames,
                                   uint64_t n_samples, uint16_t channels, glc_frames **out) {
  if (!ctx || !d_records || !out) return fail(ctx, GLC_EINVAL, "glc_frames_from_device_records: null argument");
  *out = nullptr;
  const glc_plan plan = glc::plan_encode(n_samples, channels);
  if (plan.n_frames == 0 || plan.n_frames != n_frames)
    return fail(ctx, GLC_EINVAL, "glc_frames_from_device_records: record count does not match the stream length");
  DeviceGuard guard(ctx->device);
  const glc::CompactLayout l = glc::compact_layout(channels, n_frames);
  GLC_HIP(ctx, ctx->pack_blob.reserve(l.bound));
  glc_compact_info info;
  int rc = glc_compact_device_records(ctx, d_records, n_frames, channels, ctx->pack_blob.p, ctx->pack_blob.cap, &info);
  if (rc != GLC_OK) return rc;
  // Only the bitstream's payload crosses PCIe, and it lands where it stays: the (u16, i16) pairs
  // and the raw planes are copied straight into the EncodedAudio's own pools, the small per-frame /
  // per-row metadata through pinned staging.
  const uint32_t ch = channels;
  const uint64_t raw_off = glc::compact_raw_offset(l, info.n_pairs);
  std::unique_ptr<glc_frames> F(new (std::nothrow) glc_frames);
  if (!F) return fail(ctx, GLC_ENOMEM, "glc_frames_from_device_records: host allocation failed");
  try {
    F->sample_rate = ctx->sample_rate;
    F->channels = channels;
    F->total_samples = n_samples;           // src/codec.rs:423,555
    F->encoder_delay = plan.encoder_delay;  // :547
    F->padding = plan.padding;              // :546
    F->original_length = n_samples;         // :562
    F->n_frames = n_frames;
    F->list_begin.assign(n_frames + 1, 0);
    F->scale_begin.assign(n_frames + 1, 0);
    F->raw_begin.assign(n_frames + 1, 0);
    F->raw_tag.resize(n_frames);
    F->pairs.resize(info.n_pairs);
    F->raw.resize(info.n_raw_rows * glc::kFrame);
    const uint64_t n_comp_rows = n_frames * ch - info.n_raw_rows;
    F->list_off.reserve(n_comp_rows + 1);
    F->list_off.push_back(0);
    F->scales.reserve(n_comp_rows);
  } catch (const std::bad_alloc &) {
    return fail(ctx, GLC_ENOMEM, "glc_frames_from_device_records: host allocation failed");
  }
  GLC_HIP(ctx, ctx->host_stage.reserve(l.o_pairs));
  const uint8_t *blob = static_cast<const uint8_t *>(ctx->pack_blob.p);
  GLC_HIP(ctx, hipMemcpyAsync(ctx->host_stage.p, blob, l.o_pairs, hipMemcpyDeviceToHost, ctx->stream));
  if (info.n_pairs)
    GLC_HIP(ctx, hipMemcpyAsync(F->pairs.data(), blob + l.o_pairs, info.n_pairs * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (info.n_raw_rows)
    GLC_HIP(ctx, hipMemcpyAsync(F->raw.data(), blob + raw_off, info.n_raw_rows * glc::kFrame * 2, hipMemcpyDeviceToHost,
                                ctx->stream));
  GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  glc::CompactHeader h;
  std::memcpy(&h, ctx->host_stage.p, sizeof h);
  bool canonical = true;
  try {
    rc = glc::index_compact_meta(F.get(), ch, h, static_cast<const uint8_t *>(ctx->host_stage.p), 0, 0, 0, /*trusted=*/true,
                                 &canonical);
  } catch (const std::bad_alloc &) {
    return fail(ctx, GLC_ENOMEM, "glc_frames_from_device_records: host allocation failed");
  }
  if (rc != GLC_OK) return fail(ctx, rc, std::string("glc_frames_from_device_records: ") + glc_last_error(nullptr));
  F->lists_canonical = true;  // ballot-packed in ascending k
  *out = F.release();
  return GLC_OK;
}

int glc_encode(glc_ctx *ctx, const float *pcm, uint64_t n_samples, uint16_t channels,
               glc_frames **out) {
  if (!out) return fail(ctx, GLC_EINVAL, "glc_encode: null argument");
  return glc_encode_hooked(ctx, pcm, n_samples, channels, nullptr, nullptr, out);
}

int glc_encode_hooked(glc_ctx *ctx, const float *pcm, uint64_t n_samples, uint16_t channels,
                      glc_frames_hook hook, void *hook_user, glc_frames **out) {
  if (!ctx || !pcm || (!out && !hook)) return fail(ctx, GLC_EINVAL, "glc_encode: null argument");
  if (out) *out = nullptr;
  const glc_plan plan = glc::plan_encode(n_samples, channels);
  if (plan.n_frames == 0)
    return fail(ctx, GLC_EINVAL,
                "glc_encode: the reference encoder panics on this input (channels == 0, <= 512 "
                "samples per channel, or ragged channels)");
  DeviceGuard guard(ctx->device);
  const uint32_t ch = channels;
  const uint64_t rec = glc::record_bytes(ch);
  const uint64_t t_count = (n_samples + ch - 1) / ch;
  // A pipeline over rounds of at most kEncodeChunkFrames frames.  Every stage has its own stream AND
  // its own host thread, because two of the stages block their caller: a copy from or to pageable
  // memory returns when it is done.
  //   uploader thread   round i+1's samples go up, back to back                      (copy_stream)
  //   launcher thread   round i is transformed and quantised as soon as it is up     (stream / stream_b)
  //   this thread       round i-1 is compacted, its blob comes down - metadata, then the payload
  //                     straight into the EncodedAudio's pools - and is indexed       (down_stream)
  // PCIe is full duplex and the compaction kernels are small, so a call costs
  // max(upload, kernels) + the first upload + the last round's compaction and download instead of
  // their sum.  One rule keeps the stages from blocking each other: NOTHING IS QUEUED BEHIND AN
  // UNFINISHED DEPENDENCY.  A process has a handful of hardware queues for all its streams, and a
  // queue is in order: a `hipStreamWaitEvent` that has to wait parks every stream that shares its
  // queue (seen in the trace: the compaction's wait for the quantiser held the NEXT upload's event
  // back, and with it the next transform), and the copy engines execute in submission order.  So a
  // stage waits on the host (thread hand-off, hipEventSynchronize) and only then queues its work.
  // Rounds: the upload nothing hides is the first round's, and the kernels nothing hides are the last
  // round's, so a stream opens with four rounds of 2048 rows (0.15 ms of PCIe each for stereo) before
  // it goes on in rounds of kEncodeChunkFrames.  A 2048-row launch alone leaves the chip half empty
  // (its time is one workgroup's 2048-step chain, 0.22 ms whatever the row count), so consecutive
  // rounds run on TWO streams with a coefficient workspace each: the next round's transform fills in
  // beside this one's, and a round's quantiser runs beside the next transform (four 1024-frame stereo
  // launches: 0.87 ms on one stream, 0.67 ms alternating, 0.59 ms as one launch; two 2048-frame ones
  // 0.66 / 0.60).  BASELINE config 2 (4096 stereo frames) is the four opening rounds; a stream of
  // one round runs on this thread alone.
  // (at most 2048 rows: the size the round kernel of launch_mdct_forward - `beside` - is dealt one workgroup per CU for)
  const uint64_t piece = std::max<uint64_t>(1, 2048 / ch);  // frames of an opening round
  const uint64_t opening = 4 * piece;
  struct Round {
    uint64_t f0, nf, blob_off, hi;  // hi: interleaved samples that must be on the device before its kernels run
    glc::CompactLayout l;
  };
  std::vector<Round> rounds;
  std::unique_ptr<glc_frames> F;
  uint64_t max_nf = 0;
  try {
    uint64_t blob_off = 0;
    for (uint64_t f = 0, nf = 0; f < plan.n_frames; f += nf) {
      const uint64_t left = plan.n_frames - f;
      nf = std::min<uint64_t>(f < opening ? piece : encode_chunk_frames(ch), left);
      if (left - nf < piece / 2) nf = left;  // no round for a remainder of less than half a piece
      // (Measured and dropped, round 3: cutting the END of a stream in halves - .. 2048, 1024, 512, 512 frames -
      // so that the last, unhidden transform is a short one: 1.045 ms against 1.02 at config 2; a round
      // costs the launcher and the collector more than the shorter tail returns.)
      // frames [f, f+nf) read per-channel samples below 1024*(f+nf-1) - 512 + 2048
      const uint64_t hi_t = std::min<uint64_t>(t_count, (f + nf - 1) * glc::kHop + glc::kFrame - glc::kHop / 2);
      Round r{f, nf, blob_off, std::min<uint64_t>(n_samples, hi_t * ch), glc::compact_layout(ch, nf)};
      blob_off += align_up(r.l.bound, 256);
      max_nf = std::max(max_nf, nf);
      rounds.push_back(r);
    }
    F.reset(new glc_frames);
    F->sample_rate = ctx->sample_rate;
    F->channels = channels;
    F->total_samples = n_samples;           // src/codec.rs:423,555
    F->encoder_delay = plan.encoder_delay;  // :547
    F->padding = plan.padding;              // :546
    F->original_length = n_samples;         // :562
    F->n_frames = plan.n_frames;
    F->list_begin.assign(plan.n_frames + 1, 0);
    F->scale_begin.assign(plan.n_frames + 1, 0);
    F->raw_begin.assign(plan.n_frames + 1, 0);
    F->raw_tag.resize(plan.n_frames);
    F->list_off.reserve(plan.n_frames * ch + 1);  // never reallocated: a hook may be reading it while later rounds append
    F->list_off.push_back(0);
    F->scales.reserve(plan.n_frames * ch);
  } catch (const std::bad_alloc &) {
    return fail(ctx, GLC_ENOMEM, "glc_encode: host allocation failed");
  }
  const size_t n_rounds = rounds.size();
  // pinned landing zone of a round's blob: metadata + its expected pairs (at most the whole blob, at most 64 MiB)
  const size_t stage_cap = std::max<size_t>(std::min<size_t>(align_up(glc::compact_layout(ch, max_nf).bound, 256), size_t(64) << 20),
                                            align_up(glc::compact_layout(ch, max_nf).o_pairs, 256));
  GLC_HIP(ctx, ctx->pcm.reserve(static_cast<size_t>(t_count) * ch * sizeof(float)));
  GLC_HIP(ctx, ctx->records.reserve(static_cast<size_t>(plan.n_frames) * rec));
  GLC_HIP(ctx, ctx->pack_blob.reserve(rounds.back().blob_off + align_up(rounds.back().l.bound, 256)));
  GLC_HIP(ctx, ctx->host_stage.reserve(stage_cap));
  // the per-round workspaces at their largest now: growing one mid-pipeline would free it under queued work
  GLC_HIP(ctx, ctx->coef.reserve(static_cast<size_t>(max_nf) * ch * glc::kHop * sizeof(float)));
  if (n_rounds > 1) GLC_HIP(ctx, ctx->coef_b.reserve(static_cast<size_t>(max_nf) * ch * glc::kHop * sizeof(float)));
  GLC_HIP(ctx, ctx->pack_meta.reserve(compact_scratch_bytes(max_nf * ch)));
  if (!ctx->stream_b) GLC_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_b, hipStreamNonBlocking));
  if (!ctx->copy_stream) GLC_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (!ctx->down_stream) GLC_HIP(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
  while (ctx->ev_round.size() < n_rounds) {
    hipEvent_t e = nullptr;
    GLC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    try {
      ctx->ev_round.push_back(e);
    } catch (const std::bad_alloc &) {
      (void)hipEventDestroy(e);
      return fail(ctx, GLC_ENOMEM, "glc_encode: host allocation failed");
    }
  }
  GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // earlier work may still read the staging buffers
  float *d_pcm = static_cast<float *>(ctx->pcm.p);
  uint8_t *d_blob = static_cast<uint8_t *>(ctx->pack_blob.p);
  uint8_t *h_meta = static_cast<uint8_t *>(ctx->host_stage.p);
  hipEvent_t *ev_rec = ctx->ev_round.data();

  // progress shared by the three threads; an error anywhere stops all of them
  struct Progress {
    std::mutex mu;
    std::condition_variable cv;
    size_t uploaded = 0, queued = 0;  // rounds whose samples are on the device / whose records event has been recorded
    int rc = GLC_OK;
    std::string msg;
    void set_error(int code, const std::string &m) {
      std::lock_guard<std::mutex> lk(mu);
      if (rc == GLC_OK) rc = code, msg = m;
      cv.notify_all();
    }
    // wait until *counter > i; false when another stage has failed
    bool wait_for(const size_t &counter, size_t i) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return counter > i || rc != GLC_OK; });
      return rc == GLC_OK;
    }
    void advance(size_t &counter) {
      std::lock_guard<std::mutex> lk(mu);
      ++counter;
      cv.notify_all();
    }
  } prog;
  auto hip_msg = [](const char *what, hipError_t e) { return std::string(what) + ": " + hipGetErrorString(e); };
  std::atomic<double> pairs_per_frame{0.0};  // stored pairs per frame so far, + 25 % (sizes the next round's first copy)

  auto upload = [&] {  // stage 1
    DeviceGuard g(ctx->device);
    uint64_t copied = 0;
    for (size_t i = 0; i < n_rounds; ++i) {
      const Round &r = rounds[i];
      hipError_t e = hipSuccess;
      if (r.hi > copied)
        e = hipMemcpyAsync(d_pcm + copied, pcm + copied, (r.hi - copied) * sizeof(float), hipMemcpyHostToDevice, ctx->copy_stream);
      copied = std::max(copied, r.hi);
      // a copy from pageable memory has completed when the call returns; should the runtime ever
      // queue it instead, this is where it is waited for (the launcher queues nothing behind it)
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
      if (e != hipSuccess) return prog.set_error(e == hipErrorOutOfMemory ? GLC_ENOMEM : GLC_EHIP, hip_msg("glc_encode: upload", e));
      prog.advance(prog.uploaded);
      { std::lock_guard<std::mutex> lk(prog.mu); if (prog.rc != GLC_OK) return; }
    }
  };
  auto launch = [&] {  // stage 2
    std::string my_err;  // failures of this stage land here, not in ctx->err (another thread's to write)
    struct SinkGuard {
      std::string *prev;
      explicit SinkGuard(std::string *s) : prev(t_err_sink) { t_err_sink = s; }
      ~SinkGuard() { t_err_sink = prev; }
    } sink(&my_err);
    DeviceGuard g(ctx->device);
    for (size_t i = 0; i < n_rounds; ++i) {
      const Round &r = rounds[i];
      if (!prog.wait_for(prog.uploaded, i)) return;
      hipStream_t cs = (i & 1) ? ctx->stream_b : ctx->stream;
      hipError_t e = hipSuccess;
      uint8_t *recs = static_cast<uint8_t *>(ctx->records.p) + r.f0 * rec;
      int rc = encode_range_on(ctx, cs, (i & 1) ? ctx->coef_b : ctx->coef, d_pcm, 0, t_count, n_samples, channels, r.f0,
                               r.f0 + r.nf, recs, nullptr, /*alternate_ok=*/false, /*beside=*/true);
      if (rc != GLC_OK) return prog.set_error(rc, my_err);
      e = hipEventRecord(ev_rec[i], cs);
      if (e != hipSuccess) return prog.set_error(GLC_EHIP, hip_msg("glc_encode: queueing a round", e));
      prog.advance(prog.queued);
    }
  };
  uint64_t p_used = 0, r_used = 0;  // the pools are grown ahead of need (below): what of them is filled
  // glc_encode_hooked: the hook runs on THIS thread (the collecting one), in the time it would otherwise
  // spend blocked waiting for the device: while round i's records are not ready, the frames of the
  // rounds already collected are handed out in small ranges, the event polled in between; whatever is
  // left when the last round has been collected follows at the end.  (A hook thread of the library's
  // own was built first and measured 2.46 ms against 1.49 for the config-2 call: what a hook does is
  // allocate - the host's nested vectors - and memory allocated on a helper thread and freed by the
  // caller lives in a secondary malloc arena that is trimmed and re-faulted on every call.)
  uint64_t delivered = 0, complete = 0;  // frames handed to the hook / complete in the EncodedAudio
  constexpr uint64_t kHookSlice = 64;    // frames per call while the device is being polled (a few microseconds of host work)
  auto deliver = [&](uint64_t upto) -> bool {  // frames [delivered, upto), upto <= complete
    glc_frames_view v;
    (void)glc_frames_get_view(F.get(), &v);
    v.n_frames = upto;
    v.n_lists = F->list_begin[upto];
    v.n_pairs = F->list_off[v.n_lists];
    v.n_scales = F->scale_begin[upto];
    v.n_raw = F->raw_begin[upto];
    const int hrc = hook(hook_user, &v, delivered, upto);
    delivered = upto;
    if (hrc != 0) prog.set_error(GLC_EINVAL, "glc_encode_hooked: the hook asked to stop");
    return hrc == 0;
  };
  // block until `st` has drained - handing out finished frames meanwhile (hooked encodes)
  auto drain = [&](hipStream_t st) -> hipError_t {
    if (hook)
      while (delivered < complete && hipStreamQuery(st) == hipErrorNotReady)
        if (!deliver(std::min(complete, delivered + kHookSlice))) break;
    return hipStreamSynchronize(st);
  };
  auto grow_pairs = [&](size_t want) {
    if (F->pairs.size() < want) F->pairs.resize(want);
  };
  auto grow_raw = [&](size_t want) {
    if (F->raw.size() < want) F->raw.resize(want);
  };
  auto collect = [&] {  // stage 3
    for (size_t i = 0; i < n_rounds; ++i) {
      const Round &r = rounds[i];
      if (i > 0) {
        // Growing a pool zero-fills it: do that for this round while its kernels still run, from the
        // density of the stream so far (+ 25 %); a round that turns out denser grows again below.
        const double per_frame = static_cast<double>(p_used) / static_cast<double>(r.f0) * 1.25;
        const uint64_t want = p_used + static_cast<uint64_t>(per_frame * static_cast<double>(r.nf)) + 1024;
        grow_pairs(want);
      }
      if (hook) {  // hand out finished frames while this round's records are not ready
        while (delivered < complete) {
          bool ready;
          {
            std::lock_guard<std::mutex> lk(prog.mu);
            if (prog.rc != GLC_OK) return;
            ready = prog.queued > i;
          }
          if (ready && hipEventQuery(ev_rec[i]) == hipSuccess) break;
          if (!deliver(std::min(complete, delivered + kHookSlice))) return;
        }
      }
      if (!prog.wait_for(prog.queued, i)) return;
      hipError_t e = hipEventSynchronize(ev_rec[i]);  // the round's records are written: compact them now
      if (e != hipSuccess) return prog.set_error(GLC_EHIP, hip_msg("glc_encode: waiting for a round", e));
      const uint8_t *blob = d_blob + r.blob_off;
      uint8_t *hm = h_meta;
      {
        uint8_t *recs = static_cast<uint8_t *>(ctx->records.p) + r.f0 * rec;
        const int crc = compact_launch(ctx, recs, r.nf, ch, d_blob + r.blob_off, ctx->down_stream);
        if (crc != GLC_OK) return prog.set_error(crc, "glc_encode: compaction launch failed");
      }
      // One copy fetches the metadata (header, raw flags, scale factors, list lengths: they say how
      // long the payload is) AND as much of the pair section behind it as this round is expected to
      // fill, from the density of the stream so far (+ 25 %), into pinned memory; a round that holds
      // more, or raw planes, fetches the rest straight into the pools once the header is known.
      // (Queueing the last round's compaction and download behind its quantiser on the round's own
      // stream, to save the launch latency, measured 70 us SLOWER at config 2.)
      uint64_t guess_pairs = 0;
      if (i > 0) guess_pairs = static_cast<uint64_t>(pairs_per_frame.load(std::memory_order_relaxed) * static_cast<double>(r.nf)) + 1024;
      // ... for short rounds, where the second round trip is what costs; a long round's payload (3.7 MB
      // for 4096 stereo frames) goes straight into the pools - the host copy out of the pinned buffer
      // would cost more than the round trip (one hour of stereo: 61 ms with it, 54 without)
      if (guess_pairs * 4 > (size_t(3) << 19)) guess_pairs = 0;
      const uint64_t first_bytes = std::min<uint64_t>(r.l.o_pairs + guess_pairs * 4, std::min<uint64_t>(r.l.bound, stage_cap));
      e = hipMemcpyAsync(hm, blob, first_bytes, hipMemcpyDeviceToHost, ctx->down_stream);
      if (e == hipSuccess) e = drain(ctx->down_stream);
      if (e != hipSuccess) return prog.set_error(GLC_EHIP, hip_msg("glc_encode: download", e));
      glc::CompactHeader h;
      std::memcpy(&h, hm, sizeof h);
      if (h.magic != glc::kCompactMagic || h.n_frames != r.nf || h.channels != ch || h.bytes > r.l.bound ||
          h.n_pairs > r.nf * ch * glc::kHop || h.n_raw_rows > r.nf * ch)
        return prog.set_error(GLC_EHIP, "glc_encode: the device wrote an inconsistent compact header");
      const uint64_t p_at = p_used, r_at = r_used;
      if (i == 0 && n_rounds > 1) {  // reserve the pools once from the first round's density (+ 30 %)
        const double scale = static_cast<double>(plan.n_frames) / static_cast<double>(r.nf) * 1.3;
        F->pairs.reserve(static_cast<size_t>(static_cast<double>(h.n_pairs) * scale) + 4096);
        if (h.n_raw_rows) F->raw.reserve(static_cast<size_t>(static_cast<double>(h.n_raw_rows * glc::kFrame) * scale));
      }
      grow_pairs(p_at + h.n_pairs);
      grow_raw(r_at + h.n_raw_rows * glc::kFrame);
      p_used += h.n_pairs;
      r_used += h.n_raw_rows * glc::kFrame;
      pairs_per_frame.store(static_cast<double>(p_used) / static_cast<double>(r.f0 + r.nf) * 1.25, std::memory_order_relaxed);
      const uint64_t have = std::min<uint64_t>(h.n_pairs, (first_bytes - r.l.o_pairs) / 4);  // pairs already on the host
      if (h.n_pairs > have)
        e = hipMemcpyAsync(F->pairs.data() + p_at + have, blob + r.l.o_pairs + have * 4, (h.n_pairs - have) * 4, hipMemcpyDeviceToHost,
                           ctx->down_stream);
      if (e == hipSuccess && h.n_raw_rows)
        e = hipMemcpyAsync(F->raw.data() + r_at, blob + glc::compact_raw_offset(r.l, h.n_pairs), h.n_raw_rows * glc::kFrame * 2,
                           hipMemcpyDeviceToHost, ctx->down_stream);
      if (have) std::memcpy(F->pairs.data() + p_at, hm + r.l.o_pairs, have * 4);
      bool canonical = true;
      int rc = GLC_OK;
      if (e == hipSuccess) rc = glc::index_compact_meta(F.get(), ch, h, hm, r.f0, p_at, r_at, /*trusted=*/true, &canonical);
      // the pools may move when the next round grows them, and h_meta is reused: all of it has to have landed
      if (h.n_pairs > have || h.n_raw_rows) {
        const hipError_t e2 = drain(ctx->down_stream);
        if (e == hipSuccess) e = e2;
      }
      if (e != hipSuccess) return prog.set_error(GLC_EHIP, hip_msg("glc_encode: download", e));
      if (rc != GLC_OK) return prog.set_error(rc, std::string("glc_encode: ") + glc_last_error(nullptr));
      complete = r.f0 + r.nf;
    }
    if (hook && delivered < complete) (void)deliver(complete);
  };

  int rc = GLC_OK;
  std::string msg;
  try {
    if (n_rounds == 1) {
      upload();
      launch();
      collect();
    } else {
      if (!ctx->enc_up) ctx->enc_up.reset(new Worker);
      if (!ctx->enc_launch) ctx->enc_launch.reset(new Worker);
      auto guarded = [&prog](const std::function<void()> &f) {
        return [&prog, f] {
          try {
            f();
          } catch (...) {
            try {
              prog.set_error(GLC_ENOMEM, "glc_encode: host allocation failed");
            } catch (...) {
            }
          }
        };
      };
      int started = 0;
      try {
        ctx->enc_up->submit(guarded(upload));
        started = 1;
        ctx->enc_launch->submit(guarded(launch));
        started = 2;
      } catch (...) {  // a helper could not be started: stop the others, report
        prog.set_error(GLC_ENOMEM, "glc_encode: cannot start a helper thread");
        if (started >= 1) ctx->enc_up->wait();
        if (started >= 2) ctx->enc_launch->wait();
        throw;
      }
      try {
        collect();
      } catch (...) {
        prog.set_error(GLC_ENOMEM, "glc_encode: host allocation failed");
      }
      ctx->enc_up->wait();
      ctx->enc_launch->wait();
    }
    std::lock_guard<std::mutex> lk(prog.mu);
    rc = prog.rc;
    msg = prog.msg;
  } catch (const std::bad_alloc &) {
    rc = GLC_ENOMEM, msg = "glc_encode: host allocation failed";
  } catch (...) {  // std::system_error from starting a helper thread; nothing may cross the C ABI
    rc = GLC_ENOMEM, msg = "glc_encode: cannot start a helper thread";
  }
  if (rc != GLC_OK) {
    // nothing may still be in flight into the caller's or the result's memory when this returns
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream_b);
    (void)hipStreamSynchronize(ctx->down_stream);
    return fail(ctx, rc, msg);
  }
  try {
    F->pairs.resize(p_used);  // shrinks: the pools were grown ahead of need
    F->raw.resize(r_used);
  } catch (const std::bad_alloc &) {
    return fail(ctx, GLC_ENOMEM, "glc_encode: host allocation failed");
  }
  F->lists_canonical = true;  // ballot-packed in ascending k
  if (out) *out = F.release();
  return GLC_OK;
}

// ------------------------------------------------------------------------------ decode

namespace {

// Upload the sparse representation of `in` and reset the overlap state: after this the stream
// can be decoded front to back in rounds (round_launch).  Sparse lists are used as stored when
// canonical (strictly ascending, idx < 1024 — what the encoder emits); other lists are
// canonicalised on the host with the reference's dense-array semantics (last write wins,
// idx >= 1024 ignored, src/codec.rs:659-665) and appended behind the stored pairs.
int decode_prepare_impl(glc_ctx *ctx, const glc_frames *in);

int decode_prepare(glc_ctx *ctx, const glc_frames *in) {
  try {  // no C++ exception may cross the C ABI
    return decode_prepare_impl(ctx, in);
  } catch (const std::bad_alloc &) {
    return fail(ctx, GLC_ENOMEM, "glc_decode: host allocation failed");
  }
}

int decode_prepare_impl(glc_ctx *ctx, const glc_frames *in) {
  const uint32_t ch = in->channels;
  if (ch == 0) return fail(ctx, GLC_EINVAL, "glc_decode: header.channels == 0");
  const uint64_t nf = in->n_frames;
  const uint64_t M = nf * ch;
  if (M > 0xFFFFFFFFull) return fail(ctx, GLC_EINVAL, "glc_decode: stream too long");
  if (ctx->d1_variant != 1) {
    // D1's plan workspace: one launch of this stream's (8-frame group, channel) units, at most kPlanGroups
    // per batch (66 KB per unit: a 12-frame clip needs 4 units, not the 135 MB of a full batch)
    DeviceGuard guard_plan(ctx->device);
    const uint64_t units = ((nf + 7) / 8) * ch;
    const uint32_t want = static_cast<uint32_t>(std::max<uint64_t>(ch, std::min<uint64_t>(std::max<uint32_t>(kPlanGroups, ch), units)));
    if (want > ctx->dec_plan_groups) {
      GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // queued launches may still read the old workspace
      GLC_HIP(ctx, ctx->dec_plan.reserve(glc::imdct_plan_bytes(want)));
      ctx->dec_plan_groups = want;
      ctx->plan_uid = 0;
    }
  }
  // The sparse rows of this stream are still on the device from an earlier call (a glc_frames is
  // immutable and its uid is unique in the process, or a caller-supplied identity of the content;
  // the pool sizes are compared as well, so that a recycled id does not silently decode old rows).
  if (ctx->dec_uid != 0 && ctx->dec_uid == in->uid && ctx->dec_frames == nf && ctx->dec_ch == ch &&
      ctx->dec_n_pairs == in->pairs.size() && ctx->dec_n_raw == in->raw.size() && ctx->dec_delay == in->encoder_delay &&
      ctx->dec_orig_len == in->original_length) {
    ctx->dec_next = 0;
    return GLC_OK;
  }
  ctx->dec_uid = 0;
  ctx->plan_uid = 0;

  // The five per-row arrays are built in ONE host block in the layout they have on the device, so that
  // they go up in one copy (each copy from pageable memory costs ~20 us before it moves a byte: eight
  // of them were a third of what a context spends on a stream it has not seen).
  const size_t Mr = std::max<size_t>(M, 1);
  const size_t h_begin = 0, h_cnt = align_up(h_begin + Mr * 8, 256), h_scale = align_up(h_cnt + Mr * 4, 256),
               h_raw = align_up(h_scale + Mr * 4, 256), h_rawlen = align_up(h_raw + Mr * 8, 256),
               h_end = align_up(h_rawlen + Mr * 8, 256);
  std::vector<uint64_t> host_rows(h_end / 8, 0);  // 8-byte aligned storage
  uint8_t *hb = reinterpret_cast<uint8_t *>(host_rows.data());
  uint64_t *row_begin = reinterpret_cast<uint64_t *>(hb + h_begin);
  uint32_t *row_cnt = reinterpret_cast<uint32_t *>(hb + h_cnt);
  float *row_scale = reinterpret_cast<float *>(hb + h_scale);
  int64_t *row_raw = reinterpret_cast<int64_t *>(hb + h_raw);
  uint64_t *row_raw_len = reinterpret_cast<uint64_t *>(hb + h_rawlen);
  for (uint64_t m = 0; m < M; ++m) row_raw[m] = -1;
  uint32_t any_raw = 0;
  std::vector<uint32_t> extra;  // canonicalised copies of non-canonical lists
  std::vector<int32_t> dense;
  const uint64_t n_stored = in->pairs.size();
  for (uint64_t f = 0; f < nf; ++f) {
    if (in->raw_tag[f]) {
      any_raw = 1;
      for (uint32_t c = 0; c < ch; ++c) {
        row_raw[f * ch + c] = static_cast<int64_t>(in->raw_begin[f]);
        row_raw_len[f * ch + c] = in->raw_begin[f + 1] - in->raw_begin[f];
      }
      continue;
    }
    const uint64_t l0 = in->list_begin[f], nl = in->list_begin[f + 1] - l0;
    const uint64_t s0 = in->scale_begin[f], ns = in->scale_begin[f + 1] - s0;
    if (nl < ch || ns < ch)  // the reference indexes [ch] out of bounds and panics, :652-653
      return fail(ctx, GLC_EFORMAT, "glc_decode: frame has fewer channel vectors than header.channels");
    for (uint32_t c = 0; c < ch; ++c) {
      const uint64_t a = in->list_off[l0 + c], b = in->list_off[l0 + c + 1];
      bool canonical = true;
      if (!in->lists_canonical && b > a) {
        // strictly ascending indices below 1024: branch-free over the whole list, so that the compiler
        // vectorises it (streams built by glc_frames_from_parts / _gather / glc_deserialize come through
        // here on their first decode: 0.9 M pairs at config 2)
        const uint32_t *pp = in->pairs.data() + a;
        const uint64_t n_p = b - a;
        uint32_t bad = (pp[0] & 0xFFFFu) >= glc::kHop ? 1u : 0u;
        for (uint64_t j = 1; j < n_p; ++j) {
          const uint32_t k = pp[j] & 0xFFFFu, kp = pp[j - 1] & 0xFFFFu;
          bad |= (k <= kp ? 1u : 0u) | (k >= glc::kHop ? 1u : 0u);
        }
        canonical = bad == 0;
      }
      const uint64_t m = f * ch + c;
      if (canonical) {
        row_begin[m] = a;
        row_cnt[m] = static_cast<uint32_t>(b - a);
      } else {
        dense.assign(glc::kHop, INT32_MIN);
        for (uint64_t j = a; j < b; ++j) {
          const uint32_t k = in->pairs[j] & 0xFFFFu;
          if (k < glc::kHop) dense[k] = static_cast<int16_t>(in->pairs[j] >> 16);
        }
        row_begin[m] = n_stored + extra.size();
        // stored zeros stay: 0 * scale is NaN when the scale is infinite (src/codec.rs:663), and
        // the result must not depend on whether the list happened to be in canonical order
        for (uint32_t k = 0; k < glc::kHop; ++k)
          if (dense[k] != INT32_MIN)
            extra.push_back(k | (static_cast<uint32_t>(static_cast<uint16_t>(dense[k])) << 16));
        row_cnt[m] = static_cast<uint32_t>(n_stored + extra.size() - row_begin[m]);
      }
      row_scale[m] = in->scales[s0 + c];
    }
  }

  DeviceGuard guard(ctx->device);
  size_t off = 0;
  auto place = [&](size_t bytes) {
    size_t at = off;
    off = align_up(off + bytes, 256);
    return at;
  };
  const size_t o_pairs = place(std::max<size_t>(n_stored + extra.size(), 1) * 4);
  const size_t o_rows = place(h_end);  // the block above, as it is
  const size_t o_begin = o_rows + h_begin, o_cnt = o_rows + h_cnt, o_scale = o_rows + h_scale, o_raw = o_rows + h_raw,
               o_rawlen = o_rows + h_rawlen;
  const size_t o_pool = place(std::max<size_t>(in->raw.size(), 1) * 2);
  GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a previous session may still read dec_meta
  GLC_HIP(ctx, ctx->dec_meta.reserve(off));
  uint8_t *mb = static_cast<uint8_t *>(ctx->dec_meta.p);
  auto up = [&](size_t o, const void *src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    return hipMemcpyAsync(mb + o, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  };
  GLC_HIP(ctx, up(o_pairs, in->pairs.data(), n_stored * 4));
  GLC_HIP(ctx, up(o_pairs + n_stored * 4, extra.data(), extra.size() * 4));
  GLC_HIP(ctx, up(o_rows, hb, h_end));
  GLC_HIP(ctx, up(o_pool, in->raw.data(), in->raw.size() * 2));
  GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host vectors above go out of scope
  ctx->dec_rows = glc::DecodeRows{reinterpret_cast<const uint32_t *>(mb + o_pairs),
                                  reinterpret_cast<const uint64_t *>(mb + o_begin),
                                  reinterpret_cast<const uint32_t *>(mb + o_cnt),
                                  reinterpret_cast<const float *>(mb + o_scale),
                                  reinterpret_cast<const int64_t *>(mb + o_raw),
                                  reinterpret_cast<const uint64_t *>(mb + o_rawlen),
                                  reinterpret_cast<const int16_t *>(mb + o_pool), any_raw};
  ctx->dec_ch = ch;
  ctx->dec_frames = nf;
  ctx->dec_next = 0;
  ctx->dec_uid = in->uid;
  ctx->dec_delay = in->encoder_delay;
  ctx->dec_orig_len = in->original_length;
  ctx->dec_n_pairs = in->pairs.size();
  ctx->dec_n_raw = in->raw.size();
  return GLC_OK;
}

// D1 for rows [row_begin, row_begin + M) of the prepared stream.  A launch that fits one batch of the
// plan workspace leaves its plan records (and the unit order) behind; the same launch of the same
// stream again - a repeated decode - skips k_imdct_plan.
int launch_d1(glc_ctx *ctx, uint32_t row_begin, uint32_t M, float *blocks) {
  const uint32_t ch = ctx->dec_ch;
  const bool planned = ctx->d1_variant != 1 && ch != 0 && M % ch == 0;
  const bool one_batch = planned && ((M / ch + 7) / 8) * ch <= ctx->dec_plan_groups;
  const bool reuse = one_batch && ctx->plan_uid == ctx->dec_uid && ctx->plan_uid != 0 && ctx->plan_row_begin == row_begin &&
                     ctx->plan_M == M;
  GLC_HIP(ctx, glc::launch_imdct_rows(ctx->dev, ctx->dec_rows, row_begin, M, ch, blocks, ctx->stream, ctx->d1_variant,
                                      ctx->dec_plan.p, ctx->dec_plan.p ? ctx->dec_plan_groups : 0, reuse));
  if (planned) {
    ctx->plan_uid = one_batch ? ctx->dec_uid : 0;
    ctx->plan_row_begin = row_begin;
    ctx->plan_M = M;
  }
  return GLC_OK;
}

// Queue the kernels of the next round of at most `round_frames` frames of the prepared session into
// `dout` (hop dec_next at dout[0]) and mark their completion with `ev`.  A round that reaches the
// last frame also produces the bare overlap tail (src/codec.rs:722-729); `flush_at_full` selects
// the reference's streaming rule (a chunk is flushed once it holds >= 500 frames, :708-717, so a
// stream of exactly k*500 frames ends with a tail-only chunk).  Slot 0 of the block ring carries
// the previous round's last frame for the overlap-add.
int round_launch(glc_ctx *ctx, uint64_t round_frames, bool flush_at_full, float *dout, hipEvent_t ev,
                 uint64_t *frames_out, bool *last_out) {
  const uint32_t ch = ctx->dec_ch;
  const uint64_t nf = ctx->dec_frames, f0 = ctx->dec_next;
  const uint64_t left = nf - f0;
  const bool last = flush_at_full ? left < round_frames : left <= round_frames;
  const uint64_t n = last ? left : round_frames;
  const size_t slot = static_cast<size_t>(ch) * glc::kFrame;  // floats per frame
  DeviceGuard guard(ctx->device);
  float *blocks = static_cast<float *>(ctx->blocks.p);
  // (overlap = 0.0 before the first frame, :601: the overlap-add never reads slot 0 for hop 0)
  if (n) {
    const int rc = launch_d1(ctx, static_cast<uint32_t>(f0 * ch), static_cast<uint32_t>(n * ch), blocks + slot);
    if (rc != GLC_OK) return rc;
  }
  GLC_HIP(ctx, glc::launch_overlap_add(blocks, static_cast<int64_t>(f0) - 1, nf, ch, f0, f0 + n + (last ? 1 : 0), dout,
                                       ctx->stream));
  if (!last)
    GLC_HIP(ctx, hipMemcpyAsync(blocks, blocks + n * slot, slot * sizeof(float), hipMemcpyDeviceToDevice,
                                ctx->stream));
  GLC_HIP(ctx, hipEventRecord(ev, ctx->stream));
  ctx->dec_next = f0 + n;
  *frames_out = n;
  *last_out = last;
  return GLC_OK;
}

int ensure_copy_objects(glc_ctx *ctx) {
  DeviceGuard guard(ctx->device);
  if (!ctx->copy_stream) GLC_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  for (hipEvent_t &e : ctx->ev_dec)
    if (!e) GLC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return GLC_OK;
}

// Hops [hop_begin, hop_end) of the un-trimmed stream (hop h = second half of frame h-1 + first
// half of frame h; hop n_frames is the bare tail) into device memory at d_out, after
// decode_prepare.  A range that does not start at 0 recomputes frame hop_begin-1 as its halo: the
// only state the overlap-add carries (src/codec.rs:701-705), which is what lets GPUs decode
// disjoint hop ranges of one stream independently.
int decode_hops_prepared(glc_ctx *ctx, uint64_t hop_begin, uint64_t hop_end, float *d_out) {
  if (hop_end <= hop_begin) return GLC_OK;
  const uint32_t ch = ctx->dec_ch;
  const uint64_t nf = ctx->dec_frames;
  DeviceGuard guard(ctx->device);
  const uint64_t f_end = std::min(hop_end, nf);  // frames [hop_begin, f_end) are decoded for their own hops
  const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(kDecodeChunkFrames, f_end > hop_begin ? f_end - hop_begin : 1));
  const size_t slot = static_cast<size_t>(ch) * glc::kFrame;
  GLC_HIP(ctx, ctx->blocks.reserve((chunk + 1) * slot * sizeof(float)));
  float *blocks = static_cast<float *>(ctx->blocks.p);
  if (hop_begin != 0) {  // (hop 0 has no frame before it: overlap = 0.0, :601, and slot 0 is never read for it)
    const int rc = launch_d1(ctx, static_cast<uint32_t>((hop_begin - 1) * ch), ch, blocks);
    if (rc != GLC_OK) return rc;
  }
  uint64_t f0 = hop_begin;
  do {
    const uint64_t nchunk = f0 < f_end ? std::min(chunk, f_end - f0) : 0;
    const bool tail = f0 + nchunk == nf && hop_end == nf + 1;  // this round also emits the bare overlap tail
    if (nchunk) {
      const int rc = launch_d1(ctx, static_cast<uint32_t>(f0 * ch), static_cast<uint32_t>(nchunk * ch), blocks + slot);
      if (rc != GLC_OK) return rc;
    }
    GLC_HIP(ctx, glc::launch_overlap_add(blocks, static_cast<int64_t>(f0) - 1, nf, ch, f0, f0 + nchunk + (tail ? 1 : 0),
                                         d_out + (f0 - hop_begin) * glc::kHop * ch, ctx->stream));
    f0 += nchunk + (tail ? 1 : 0);
    if (f0 < hop_end)
      GLC_HIP(ctx, hipMemcpyAsync(blocks, blocks + nchunk * slot, slot * sizeof(float), hipMemcpyDeviceToDevice,
                                  ctx->stream));
  } while (f0 < hop_end);
  return GLC_OK;
}

}  // namespace

// Decoder::decode of the prepared session into host memory (src/codec.rs:744-768).
static int decode_prepared_to_host(glc_ctx *ctx, float *pcm_out, uint64_t cap, uint64_t *n_out, const char *who) {
  const uint64_t n_frames = ctx->dec_frames;
  const uint32_t channels = ctx->dec_ch;
  // gapless trim, src/codec.rs:756-765 (delay counted in INTERLEAVED samples, quirk Q3)
  const uint64_t all = (n_frames + 1) * static_cast<uint64_t>(glc::kHop) * channels;
  uint64_t start = 0, n = all;
  if (n > ctx->dec_delay) {
    start = ctx->dec_delay;
    n -= ctx->dec_delay;
  }
  if (n > ctx->dec_orig_len) n = ctx->dec_orig_len;
  if (n_out) *n_out = n;
  if (cap < n) return fail(ctx, GLC_EINVAL, std::string(who) + ": output buffer too small");
  ctx->dec_next = 0;
  // Rounds of 4096 frames through two device output buffers: the kernels of round r+1 are queued
  // before round r is copied out (on the copy stream, behind that round's event), so the D2H of
  // one round overlaps the decode of the next.  The block ring is sized once: slot 0 carries state.
  const uint64_t round = std::max<uint64_t>(1, std::min<uint64_t>(kDecodeChunkFrames, n_frames));
  const uint64_t per_hop = static_cast<uint64_t>(glc::kHop) * channels;
  const size_t bufcap = static_cast<size_t>(round + 1) * per_hop;  // floats per output buffer
  {
    DeviceGuard guard(ctx->device);
    GLC_HIP(ctx, ctx->blocks.reserve((round + 1) * static_cast<size_t>(channels) * glc::kFrame * sizeof(float)));
    GLC_HIP(ctx, ctx->pcm.reserve(2 * bufcap * sizeof(float)));
  }
  int rc = ensure_copy_objects(ctx);
  if (rc != GLC_OK) return rc;
  DeviceGuard guard(ctx->device);
  float *stage = static_cast<float *>(ctx->pcm.p);
  int buf = 0;
  uint64_t f0 = 0, frames = 0;
  bool last = false;
  rc = round_launch(ctx, round, false, stage, ctx->ev_dec[0], &frames, &last);
  if (rc != GLC_OK) return rc;
  for (;;) {
    const uint64_t cur_f0 = f0, cur_frames = frames;
    const bool cur_last = last;
    if (!cur_last) {
      f0 += frames;
      rc = round_launch(ctx, round, false, stage + static_cast<size_t>(buf ^ 1) * bufcap, ctx->ev_dec[buf ^ 1], &frames, &last);
      if (rc != GLC_OK) return rc;
    }
    const uint64_t c_lo = cur_f0 * per_hop, c_hi = (cur_f0 + cur_frames + (cur_last ? 1 : 0)) * per_hop;
    const uint64_t lo = std::max(c_lo, start), hi = std::min(c_hi, start + n);
    // a stream of a single round (short clips) has nothing to overlap: copy in stream order
    hipStream_t cs = (cur_last && cur_f0 == 0) ? ctx->stream : ctx->copy_stream;
    if (cs != ctx->stream) GLC_HIP(ctx, hipStreamWaitEvent(cs, ctx->ev_dec[buf], 0));
    if (hi > lo)
      GLC_HIP(ctx, hipMemcpyAsync(pcm_out + (lo - start), stage + static_cast<size_t>(buf) * bufcap + (lo - c_lo),
                                  (hi - lo) * sizeof(float), hipMemcpyDeviceToHost, cs));
    GLC_HIP(ctx, hipStreamSynchronize(cs));
    if (cur_last) break;
    buf ^= 1;
  }
  return GLC_OK;
}

int glc_decode(glc_ctx *ctx, const glc_frames *in, float *pcm_out, uint64_t cap, uint64_t *n_out) {
  if (!ctx || !in || (!pcm_out && cap)) return fail(ctx, GLC_EINVAL, "glc_decode: null argument");
  ctx->stream_open = false;
  if (n_out) *n_out = glc_decoded_len(in);
  if (cap < glc_decoded_len(in)) return fail(ctx, GLC_EINVAL, "glc_decode: output buffer too small");
  const int rc = decode_prepare(ctx, in);
  if (rc != GLC_OK) return rc;
  return decode_prepared_to_host(ctx, pcm_out, cap, n_out, "glc_decode");
}

uint64_t glc_ctx_resident_stream(const glc_ctx *ctx) { return ctx ? ctx->dec_uid : 0; }

int glc_decode_resident(glc_ctx *ctx, uint64_t stream_id, float *pcm_out, uint64_t cap, uint64_t *n_out) {
  if (!ctx || (!pcm_out && cap)) return fail(ctx, GLC_EINVAL, "glc_decode_resident: null argument");
  if (stream_id == 0 || ctx->dec_uid != stream_id)
    return fail(ctx, GLC_EINVAL, "glc_decode_resident: that stream is not resident on this context");
  ctx->stream_open = false;
  return decode_prepared_to_host(ctx, pcm_out, cap, n_out, "glc_decode_resident");
}

int glc_decode_device(glc_ctx *ctx, const glc_frames *in, float *d_all, uint64_t cap_all, uint64_t *start,
                      uint64_t *n_out) {
  if (!ctx || !in || !d_all) return fail(ctx, GLC_EINVAL, "glc_decode_device: null argument");
  ctx->stream_open = false;
  const uint32_t ch = in->channels;
  const uint64_t all = (in->n_frames + 1) * static_cast<uint64_t>(glc::kHop) * ch;
  if (cap_all < all) return fail(ctx, GLC_EINVAL, "glc_decode_device: output buffer too small");
  uint64_t s0 = 0, n = all;
  if (n > in->encoder_delay) {
    s0 = in->encoder_delay;
    n -= in->encoder_delay;
  }
  if (n > in->original_length) n = in->original_length;
  if (start) *start = s0;
  if (n_out) *n_out = n;
  int rc = decode_prepare(ctx, in);
  if (rc != GLC_OK) return rc;
  return decode_hops_prepared(ctx, 0, in->n_frames + 1, d_all);
}

int glc_decode_range_device(glc_ctx *ctx, const glc_frames *in, uint64_t hop_begin, uint64_t hop_end,
                            float *d_out, uint64_t cap) {
  if (!ctx || !in || !d_out) return fail(ctx, GLC_EINVAL, "glc_decode_range_device: null argument");
  ctx->stream_open = false;
  if (hop_begin > hop_end || hop_end > in->n_frames + 1)
    return fail(ctx, GLC_EINVAL, "glc_decode_range_device: hop range out of bounds");
  if (cap < (hop_end - hop_begin) * glc::kHop * in->channels)
    return fail(ctx, GLC_EINVAL, "glc_decode_range_device: output buffer too small");
  int rc = decode_prepare(ctx, in);
  if (rc != GLC_OK) return rc;
  return decode_hops_prepared(ctx, hop_begin, hop_end, d_out);
}

int glc_imdct_device(glc_ctx *ctx, const glc_frames *in, uint64_t frame_begin, uint64_t frame_end,
                     float *d_blocks) {
  if (!ctx || !in || !d_blocks) return fail(ctx, GLC_EINVAL, "glc_imdct_device: null argument");
  ctx->stream_open = false;
  if (frame_begin > frame_end || frame_end > in->n_frames)
    return fail(ctx, GLC_EINVAL, "glc_imdct_device: frame range out of bounds");
  int rc = decode_prepare(ctx, in);
  if (rc != GLC_OK) return rc;
  DeviceGuard guard(ctx->device);
  const uint32_t ch = ctx->dec_ch;
  return launch_d1(ctx, static_cast<uint32_t>(frame_begin * ch), static_cast<uint32_t>((frame_end - frame_begin) * ch), d_blocks);
}

int glc_debug_set_imdct_variant(glc_ctx *ctx, int variant) {
  if (!ctx || variant < 0 || variant > 6) return fail(ctx, GLC_EINVAL, "glc_debug_set_imdct_variant: variant must be 0..6");
  if (variant != ctx->d1_variant) ctx->plan_uid = 0;  // variants 5 / 6 deal the units differently: the kept order is not theirs
  ctx->d1_variant = variant;
  return GLC_OK;
}

int glc_debug_set_mdct_variant(glc_ctx *ctx, int variant) {
  if (!ctx || variant < 0 || variant > 4) return fail(ctx, GLC_EINVAL, "glc_debug_set_mdct_variant: variant must be 0..4");
  ctx->k1_variant = variant;
  return GLC_OK;
}

int glc_debug_clock_probe_begin(glc_ctx *ctx, uint32_t window_us) {
  if (!ctx || window_us == 0) return fail(ctx, GLC_EINVAL, "glc_debug_clock_probe_begin: bad argument");
  DeviceGuard guard(ctx->device);
  if (!ctx->probe_stream) GLC_HIP(ctx, hipStreamCreateWithFlags(&ctx->probe_stream, hipStreamNonBlocking));
  GLC_HIP(ctx, ctx->probe_out.reserve(64));  // pinned: the wave writes its two counters straight to the host
  std::memset(ctx->probe_out.p, 0, 16);
  GLC_HIP(ctx, glc::launch_clock_probe(static_cast<uint64_t>(window_us) * 100ull, static_cast<uint64_t *>(ctx->probe_out.p),
                                       ctx->probe_stream));
  return GLC_OK;
}

int glc_debug_clock_probe_end(glc_ctx *ctx, float *ghz) {
  if (!ctx || !ghz || !ctx->probe_stream) return fail(ctx, GLC_EINVAL, "glc_debug_clock_probe_end: no probe running");
  DeviceGuard guard(ctx->device);
  GLC_HIP(ctx, hipStreamSynchronize(ctx->probe_stream));
  const uint64_t *o = static_cast<const uint64_t *>(ctx->probe_out.p);
  if (o[1] == 0) return fail(ctx, GLC_EHIP, "glc_debug_clock_probe_end: the probe reported nothing");
  *ghz = static_cast<float>(static_cast<double>(o[0]) / static_cast<double>(o[1]) * 0.1);
  return GLC_OK;
}

int glc_decode_stream_begin(glc_ctx *ctx, const glc_frames *in) {
  if (!ctx || !in) return fail(ctx, GLC_EINVAL, "glc_decode_stream_begin: null argument");
  ctx->stream_open = false;
  int rc = decode_prepare(ctx, in);
  if (rc != GLC_OK) return rc;
  {
    DeviceGuard guard(ctx->device);
    const size_t chunk = static_cast<size_t>(GLC_FRAMES_PER_CHUNK) + 1;
    GLC_HIP(ctx, ctx->blocks.reserve(chunk * in->channels * glc::kFrame * sizeof(float)));
    GLC_HIP(ctx, ctx->stream_out.reserve(2 * chunk * glc::kHop * in->channels * sizeof(float)));
  }
  rc = ensure_copy_objects(ctx);
  if (rc != GLC_OK) return rc;
  ctx->stream_buf = 0;
  // the first chunk is on its way before the caller asks for it
  rc = round_launch(ctx, GLC_FRAMES_PER_CHUNK, true, static_cast<float *>(ctx->stream_out.p), ctx->ev_dec[0],
                    &ctx->stream_frames, &ctx->stream_last);
  if (rc != GLC_OK) return rc;
  ctx->stream_open = true;
  return GLC_OK;
}

int glc_decode_stream_next(glc_ctx *ctx, float *chunk, uint64_t cap, uint64_t *n_out, int *is_last) {
  if (!ctx || !n_out || !is_last) return fail(ctx, GLC_EINVAL, "glc_decode_stream_next: null argument");
  if (!ctx->stream_open) return fail(ctx, GLC_EINVAL, "glc_decode_stream_next: no stream open");
  const int buf = ctx->stream_buf;
  const bool last = ctx->stream_last;
  const uint64_t per_hop = static_cast<uint64_t>(glc::kHop) * ctx->dec_ch;
  const uint64_t n = (ctx->stream_frames + (last ? 1 : 0)) * per_hop;
  *n_out = n;
  *is_last = last ? 1 : 0;
  if (cap < n || (!chunk && n)) return fail(ctx, GLC_EINVAL, "glc_decode_stream_next: chunk buffer too small");
  const size_t bufcap = (static_cast<size_t>(GLC_FRAMES_PER_CHUNK) + 1) * per_hop;
  DeviceGuard guard(ctx->device);
  // double buffering: the kernels of the following chunk are queued first, then this chunk is
  // copied out on the copy stream as soon as its own kernels have finished
  if (!last) {
    const int rc = round_launch(ctx, GLC_FRAMES_PER_CHUNK, true,
                                static_cast<float *>(ctx->stream_out.p) + static_cast<size_t>(buf ^ 1) * bufcap,
                                ctx->ev_dec[buf ^ 1], &ctx->stream_frames, &ctx->stream_last);
    if (rc != GLC_OK) return rc;
  }
  GLC_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_dec[buf], 0));
  if (n)
    GLC_HIP(ctx, hipMemcpyAsync(chunk, static_cast<const float *>(ctx->stream_out.p) + static_cast<size_t>(buf) * bufcap,
                                n * sizeof(float), hipMemcpyDeviceToHost, ctx->copy_stream));
  GLC_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
  ctx->stream_buf = buf ^ 1;
  if (last) ctx->stream_open = false;
  return GLC_OK;
}

}  // extern "C"
