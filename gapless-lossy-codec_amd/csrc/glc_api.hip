// glc_api.hip — context, encode/decode drivers and the extern "C" surface (include/glc.h).
//
// Host-side counterpart of Encoder::new / encode (src/codec.rs:406-565) and Decoder::new /
// decode_streaming / decode (src/codec.rs:581-768).  There is no CPU compute path in this
// library: every entry point that transforms audio launches the gfx950 kernels and fails with
// GLC_ENODEV / GLC_EHIP when that is impossible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

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

struct glc_ctx {
  int device = 0;
  uint32_t sample_rate = 0;
  hipStream_t stream = nullptr;      // stream in use
  hipStream_t own_stream = nullptr;  // the context's private stream
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;  // glc_ctx_timer_*
  glc::HostTables host;
  glc::DeviceTables dev{};
  DevBuf tables;     // all constant tables in one allocation
  DevBuf coef;       // MDCT coefficient workspace [rows][1024]
  DevBuf pcm;        // staging for host-boundary encode / decode output
  DevBuf records;    // staging for host-boundary encode
  DevBuf blocks;     // decode: windowed IMDCT blocks [(chunk+1)][ch][2048]
  DevBuf dec_meta;   // decode: pairs / offsets / scales / raw pool
  std::string err;
  // streaming decode state (glc_decode_stream_*)
  std::vector<float> stream_pcm;
  uint64_t stream_pos = 0;
  uint32_t stream_ch = 0;
  bool stream_open = false;
};

namespace {

constexpr uint64_t kEncodeChunkFrames = 4096;  // rows per K1/K2/K3 round: coef stays MALL-sized
constexpr uint64_t kDecodeChunkFrames = 4096;

int fail(glc_ctx *ctx, int code, const std::string &msg) {
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
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
  if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  ctx->tables.release();
  ctx->coef.release();
  ctx->pcm.release();
  ctx->records.release();
  ctx->blocks.release();
  ctx->dec_meta.release();
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

int glc_encode_range_device(glc_ctx *ctx, const float *d_pcm, uint64_t t0, uint64_t t_count,
                            uint64_t n_samples, uint16_t channels, uint64_t frame_begin,
                            uint64_t frame_end, void *d_records, float *d_coeffs) {
  if (!ctx || !d_pcm || !d_records) return fail(ctx, GLC_EINVAL, "glc_encode_range_device: null argument");
  const glc_plan plan = glc::plan_encode(n_samples, channels);
  if (plan.n_frames == 0)
    return fail(ctx, GLC_EINVAL, "glc_encode_range_device: the reference encoder panics on this input");
  if (frame_begin > frame_end || frame_end > plan.n_frames)
    return fail(ctx, GLC_EINVAL, "glc_encode_range_device: frame range out of bounds");
  const uint32_t ch = channels;
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
  const uint64_t chunk = d_coeffs ? (frame_end - frame_begin ? frame_end - frame_begin : 1) : kEncodeChunkFrames;
  if (!d_coeffs) {
    const uint64_t rows = std::min<uint64_t>(chunk, frame_end - frame_begin) * ch;
    GLC_HIP(ctx, ctx->coef.reserve(std::max<size_t>(rows, 1) * glc::kHop * sizeof(float)));
  }
  for (uint64_t f = frame_begin; f < frame_end; f += chunk) {
    const uint64_t nf = std::min<uint64_t>(chunk, frame_end - f);
    const uint32_t M = static_cast<uint32_t>(nf * ch);
    float *coef = d_coeffs ? d_coeffs + (f - frame_begin) * ch * glc::kHop : static_cast<float *>(ctx->coef.p);
    uint8_t *r = recs + (f - frame_begin) * rec;
    GLC_HIP(ctx, glc::launch_mdct_forward(ctx->dev, view, f, M, coef, ctx->stream));
    GLC_HIP(ctx, glc::launch_quantize(ctx->dev, coef, M, ch, r, ctx->stream));
    GLC_HIP(ctx, glc::launch_decide_raw(ctx->dev, view, f, static_cast<uint32_t>(nf), r, ctx->stream));
  }
  return GLC_OK;
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
                                        ctx->stream));
  return GLC_OK;
}

int glc_encode(glc_ctx *ctx, const float *pcm, uint64_t n_samples, uint16_t channels,
               glc_frames **out) {
  if (!ctx || !pcm || !out) return fail(ctx, GLC_EINVAL, "glc_encode: null argument");
  *out = nullptr;
  const glc_plan plan = glc::plan_encode(n_samples, channels);
  if (plan.n_frames == 0)
    return fail(ctx, GLC_EINVAL,
                "glc_encode: the reference encoder panics on this input (channels == 0, <= 512 "
                "samples per channel, or ragged channels)");
  DeviceGuard guard(ctx->device);
  const uint32_t ch = channels;
  const uint64_t rec = glc::record_bytes(ch);
  const uint64_t t_count = (n_samples + ch - 1) / ch;
  GLC_HIP(ctx, ctx->pcm.reserve(static_cast<size_t>(t_count) * ch * sizeof(float)));
  GLC_HIP(ctx, ctx->records.reserve(static_cast<size_t>(plan.n_frames) * rec));
  GLC_HIP(ctx, hipMemcpyAsync(ctx->pcm.p, pcm, n_samples * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  int rc = glc_encode_range_device(ctx, static_cast<const float *>(ctx->pcm.p), 0, t_count, n_samples,
                                   channels, 0, plan.n_frames, ctx->records.p, nullptr);
  if (rc != GLC_OK) return rc;
  std::vector<uint8_t> host;
  try {
    host.resize(static_cast<size_t>(plan.n_frames) * rec);
  } catch (const std::bad_alloc &) {
    return fail(ctx, GLC_ENOMEM, "glc_encode: host allocation failed");
  }
  GLC_HIP(ctx, hipMemcpyAsync(host.data(), ctx->records.p, host.size(), hipMemcpyDeviceToHost, ctx->stream));
  GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  rc = glc_frames_from_records(ctx->sample_rate, n_samples, channels, host.data(), plan.n_frames, out);
  if (rc != GLC_OK) ctx->err = glc_last_error(nullptr);
  return rc;
}

// ------------------------------------------------------------------------------ decode

namespace {

// Decode everything to an un-trimmed interleaved buffer of (n_frames+1)*1024*ch samples on the
// host (what decode_streaming emits in total, src/codec.rs:688-732).
int decode_all(glc_ctx *ctx, const glc_frames *in, std::vector<float> &all) {
  const uint32_t ch = in->channels;
  if (ch == 0) return fail(ctx, GLC_EINVAL, "glc_decode: header.channels == 0");
  const uint64_t nf = in->n_frames;
  const uint64_t M = nf * ch;

  // Per-row metadata.  Sparse lists are used as stored when canonical (strictly ascending,
  // idx < 1024 — what the encoder emits); otherwise they are canonicalised on the host with
  // the reference's dense-array semantics (last write wins, idx >= 1024 ignored, :659-665).
  std::vector<uint64_t> row_off(M + 1, 0), row_raw_len(M, 0);
  std::vector<int64_t> row_raw(M, -1);
  std::vector<float> row_scale(M, 0.f);
  std::vector<uint32_t> pairs;
  pairs.reserve(in->pairs.size());
  std::vector<int32_t> dense;
  for (uint64_t f = 0; f < nf; ++f) {
    if (in->raw_tag[f]) {
      for (uint32_t c = 0; c < ch; ++c) {
        row_raw[f * ch + c] = static_cast<int64_t>(in->raw_begin[f]);
        row_raw_len[f * ch + c] = in->raw_begin[f + 1] - in->raw_begin[f];
        row_off[f * ch + c + 1] = pairs.size();
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
      int32_t last = -1;
      for (uint64_t j = a; j < b; ++j) {
        const int32_t k = static_cast<int32_t>(in->pairs[j] & 0xFFFFu);
        if (k <= last || k >= static_cast<int32_t>(glc::kHop)) {
          canonical = false;
          break;
        }
        last = k;
      }
      if (canonical) {
        pairs.insert(pairs.end(), in->pairs.begin() + a, in->pairs.begin() + b);
      } else {
        dense.assign(glc::kHop, INT32_MIN);
        for (uint64_t j = a; j < b; ++j) {
          const uint32_t k = in->pairs[j] & 0xFFFFu;
          if (k < glc::kHop) dense[k] = static_cast<int16_t>(in->pairs[j] >> 16);
        }
        // a stored q == 0 dequantises to +/-0.0 and contributes nothing to the running sum
        for (uint32_t k = 0; k < glc::kHop; ++k)
          if (dense[k] != INT32_MIN && dense[k] != 0)
            pairs.push_back(k | (static_cast<uint32_t>(static_cast<uint16_t>(dense[k])) << 16));
      }
      row_scale[f * ch + c] = in->scales[s0 + c];
      row_off[f * ch + c + 1] = pairs.size();
    }
  }

  DeviceGuard guard(ctx->device);
  // upload metadata
  size_t off = 0;
  auto place = [&](size_t bytes) {
    size_t at = off;
    off = align_up(off + bytes, 256);
    return at;
  };
  const size_t o_pairs = place(std::max<size_t>(pairs.size(), 1) * 4);
  const size_t o_off = place((M + 1) * 8);
  const size_t o_scale = place(std::max<size_t>(M, 1) * 4);
  const size_t o_raw = place(std::max<size_t>(M, 1) * 8);
  const size_t o_rawlen = place(std::max<size_t>(M, 1) * 8);
  const size_t o_pool = place(std::max<size_t>(in->raw.size(), 1) * 2);
  GLC_HIP(ctx, ctx->dec_meta.reserve(off));
  uint8_t *mb = static_cast<uint8_t *>(ctx->dec_meta.p);
  auto up = [&](size_t o, const void *src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    return hipMemcpyAsync(mb + o, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  };
  GLC_HIP(ctx, up(o_pairs, pairs.data(), pairs.size() * 4));
  GLC_HIP(ctx, up(o_off, row_off.data(), (M + 1) * 8));
  GLC_HIP(ctx, up(o_scale, row_scale.data(), M * 4));
  GLC_HIP(ctx, up(o_raw, row_raw.data(), M * 8));
  GLC_HIP(ctx, up(o_rawlen, row_raw_len.data(), M * 8));
  GLC_HIP(ctx, up(o_pool, in->raw.data(), in->raw.size() * 2));
  glc::DecodeRows rows{reinterpret_cast<const uint32_t *>(mb + o_pairs),
                       reinterpret_cast<const uint64_t *>(mb + o_off),
                       reinterpret_cast<const float *>(mb + o_scale),
                       reinterpret_cast<const int64_t *>(mb + o_raw),
                       reinterpret_cast<const uint64_t *>(mb + o_rawlen),
                       reinterpret_cast<const int16_t *>(mb + o_pool)};

  // Chunked: blocks buffer = slot 0 (frame before the chunk) + up to kDecodeChunkFrames frames.
  const uint64_t total = (nf + 1) * glc::kHop * ch;
  try {
    all.resize(total);
  } catch (const std::bad_alloc &) {
    return fail(ctx, GLC_ENOMEM, "glc_decode: host allocation failed");
  }
  const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(kDecodeChunkFrames, nf));
  const size_t slot = static_cast<size_t>(ch) * glc::kFrame;  // floats per frame
  GLC_HIP(ctx, ctx->blocks.reserve((chunk + 1) * slot * sizeof(float)));
  GLC_HIP(ctx, ctx->pcm.reserve((chunk + 1) * glc::kHop * ch * sizeof(float)));
  float *blocks = static_cast<float *>(ctx->blocks.p);
  float *dout = static_cast<float *>(ctx->pcm.p);
  GLC_HIP(ctx, hipMemsetAsync(blocks, 0, slot * sizeof(float), ctx->stream));  // overlap = 0.0, :601
  if (nf == 0) {
    // no frames: the output is the 1024*ch zeros of the initial overlap, :722-729
    std::fill(all.begin(), all.end(), 0.0f);
    GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLC_OK;
  }
  for (uint64_t f0 = 0; f0 < nf; f0 += chunk) {
    const uint64_t n = std::min(chunk, nf - f0);
    GLC_HIP(ctx, glc::launch_imdct_rows(ctx->dev, rows, static_cast<uint32_t>(f0 * ch),
                                        static_cast<uint32_t>(n * ch), ch, blocks + slot, ctx->stream));
    const bool last = f0 + n == nf;
    const uint64_t hop_end = f0 + n + (last ? 1 : 0);  // the final chunk also emits the tail hop
    GLC_HIP(ctx, glc::launch_overlap_add(blocks, static_cast<int64_t>(f0) - 1, nf, ch, f0, hop_end,
                                         dout, ctx->stream));
    GLC_HIP(ctx, hipMemcpyAsync(all.data() + f0 * glc::kHop * ch, dout,
                                (hop_end - f0) * glc::kHop * ch * sizeof(float), hipMemcpyDeviceToHost,
                                ctx->stream));
    if (!last)  // carry the chunk's last frame into slot 0
      GLC_HIP(ctx, hipMemcpyAsync(blocks, blocks + n * slot, slot * sizeof(float),
                                  hipMemcpyDeviceToDevice, ctx->stream));
    GLC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `all` is pageable: keep the copy ordered
  }
  return GLC_OK;
}

}  // namespace

int glc_decode(glc_ctx *ctx, const glc_frames *in, float *pcm_out, uint64_t cap, uint64_t *n_out) {
  if (!ctx || !in || (!pcm_out && cap)) return fail(ctx, GLC_EINVAL, "glc_decode: null argument");
  const uint64_t want = glc_decoded_len(in);
  if (n_out) *n_out = want;
  if (cap < want) return fail(ctx, GLC_EINVAL, "glc_decode: output buffer too small");
  std::vector<float> all;
  int rc = decode_all(ctx, in, all);
  if (rc != GLC_OK) return rc;
  // gapless trim, src/codec.rs:756-765 (delay counted in INTERLEAVED samples, quirk Q3)
  uint64_t start = 0, n = all.size();
  if (n > in->encoder_delay) {
    start = in->encoder_delay;
    n -= in->encoder_delay;
  }
  if (n > in->original_length) n = in->original_length;
  if (n) std::memcpy(pcm_out, all.data() + start, n * sizeof(float));
  if (n_out) *n_out = n;
  return GLC_OK;
}

int glc_decode_stream_begin(glc_ctx *ctx, const glc_frames *in) {
  if (!ctx || !in) return fail(ctx, GLC_EINVAL, "glc_decode_stream_begin: null argument");
  ctx->stream_open = false;
  int rc = decode_all(ctx, in, ctx->stream_pcm);
  if (rc != GLC_OK) return rc;
  ctx->stream_pos = 0;
  ctx->stream_ch = in->channels;
  ctx->stream_open = true;
  return GLC_OK;
}

int glc_decode_stream_next(glc_ctx *ctx, float *chunk, uint64_t cap, uint64_t *n_out, int *is_last) {
  if (!ctx || !n_out || !is_last) return fail(ctx, GLC_EINVAL, "glc_decode_stream_next: null argument");
  if (!ctx->stream_open) return fail(ctx, GLC_EINVAL, "glc_decode_stream_next: no stream open");
  // src/codec.rs:708-717: a chunk is flushed once it holds >= 500 frames; the remainder plus the
  // overlap tail forms the last chunk (:722-732).
  const uint64_t per_chunk = static_cast<uint64_t>(GLC_FRAMES_PER_CHUNK) * glc::kHop * ctx->stream_ch;
  const uint64_t total = ctx->stream_pcm.size();
  const uint64_t body = total - static_cast<uint64_t>(glc::kHop) * ctx->stream_ch;  // frames' hops
  uint64_t n;
  bool last;
  if (body - ctx->stream_pos >= per_chunk && ctx->stream_pos < body) {
    n = per_chunk;
    last = false;
  } else {
    n = total - ctx->stream_pos;
    last = true;
  }
  *n_out = n;
  *is_last = last ? 1 : 0;
  if (cap < n || (!chunk && n)) return fail(ctx, GLC_EINVAL, "glc_decode_stream_next: chunk buffer too small");
  if (n) std::memcpy(chunk, ctx->stream_pcm.data() + ctx->stream_pos, n * sizeof(float));
  ctx->stream_pos += n;
  if (last) {
    ctx->stream_open = false;
    std::vector<float>().swap(ctx->stream_pcm);
  }
  return GLC_OK;
}

}  // extern "C"
