// glc.hpp — header-only C++ mirror of the reference's public codec API over the C ABI (glc.h).
//
// Same names, argument meaning and results as the Rust items in src/codec.rs of
// ajcm474/gapless-lossy-codec v0.5.0 (file:line cited per member); `anyhow::Result` becomes an
// exception (glc::Error, carrying the glc_status code).  Inputs the reference panics on throw
// glc::Error(GLC_EINVAL).  All arithmetic happens in libglc_hip.so on a gfx950 device.
#pragma once
#include <cstdint>
#include <exception>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "glc.h"

namespace glc {

constexpr unsigned FRAME_SIZE = GLC_FRAME_SIZE;              // src/codec.rs:15
constexpr unsigned HOP_SIZE = GLC_HOP_SIZE;                  // src/codec.rs:16
constexpr unsigned FRAMES_PER_CHUNK = GLC_FRAMES_PER_CHUNK;  // src/codec.rs:18

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

namespace detail {
inline void check(int rc, const glc_ctx *ctx = nullptr) {
  if (rc != GLC_OK) {
    const char *m = glc_last_error(ctx);
    if (!m || !*m) m = glc_last_error(nullptr);
    throw Error(rc, m ? m : "glc error");
  }
}
}  // namespace detail

struct AudioHeader {  // src/codec.rs:39-45
  uint32_t sample_rate;
  uint16_t channels;
  uint64_t total_samples;
};
struct GaplessInfo {  // src/codec.rs:47-53
  uint32_t encoder_delay, padding;
  uint64_t original_length;
};
struct EncodedFrame {  // src/codec.rs:56-69
  std::vector<std::vector<std::pair<uint16_t, int16_t>>> sparse_coeffs_per_channel;
  std::vector<float> scale_factors;
  bool has_raw_pcm = false;  // Option<Vec<i16>>
  std::vector<int16_t> raw_pcm;
};
struct AudioChunk {  // src/codec.rs:81-85
  std::vector<float> samples;
  bool is_last;
};

// EncodedAudio (src/codec.rs:31-37): owns the library-side object; frames are read on demand.
class EncodedAudio {
 public:
  EncodedAudio() = default;
  explicit EncodedAudio(glc_frames *h) : h_(h) {}
  EncodedAudio(EncodedAudio &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  EncodedAudio &operator=(EncodedAudio &&o) noexcept {
    if (this != &o) {
      glc_frames_free(h_);
      h_ = o.h_;
      o.h_ = nullptr;
    }
    return *this;
  }
  EncodedAudio(const EncodedAudio &) = delete;
  EncodedAudio &operator=(const EncodedAudio &) = delete;
  ~EncodedAudio() { glc_frames_free(h_); }

  AudioHeader header() const {
    const glc_info i = info();
    return {i.sample_rate, i.channels, i.total_samples};
  }
  GaplessInfo gapless_info() const {
    const glc_info i = info();
    return {i.encoder_delay, i.padding, i.original_length};
  }
  uint64_t n_frames() const { return info().n_frames; }
  EncodedFrame frame(uint64_t f) const {
    EncodedFrame out;
    if (glc_frame_is_raw(h_, f) == 1) {
      uint64_t n = 0;
      detail::check(glc_frame_raw(h_, f, nullptr, 0, &n));
      out.has_raw_pcm = true;
      out.raw_pcm.resize(n);
      detail::check(glc_frame_raw(h_, f, out.raw_pcm.data(), n, &n));
    }
    for (uint32_t c = 0;; ++c) {
      uint32_t n = 0;
      if (glc_frame_sparse(h_, f, c, nullptr, nullptr, 0, &n) != GLC_OK) break;
      std::vector<uint16_t> idx(n);
      std::vector<int16_t> q(n);
      detail::check(glc_frame_sparse(h_, f, c, idx.data(), q.data(), n, &n));
      out.sparse_coeffs_per_channel.emplace_back();
      for (uint32_t j = 0; j < n; ++j) out.sparse_coeffs_per_channel.back().emplace_back(idx[j], q[j]);
    }
    for (uint32_t c = 0;; ++c) {
      float s;
      if (glc_frame_scale(h_, f, c, &s) != GLC_OK) break;
      out.scale_factors.push_back(s);
    }
    return out;
  }
  // bincode::serialize / deserialize of the struct (what save_encoded / load_encoded write)
  std::vector<uint8_t> to_bytes() const {
    std::vector<uint8_t> b(glc_serialized_size(h_));
    uint64_t w = 0;
    detail::check(glc_serialize(h_, b.data(), b.size(), &w));
    return b;
  }
  static EncodedAudio from_bytes(const uint8_t *p, uint64_t n) {
    glc_frames *h = nullptr;
    detail::check(glc_deserialize(p, n, &h));
    return EncodedAudio(h);
  }
  const glc_frames *handle() const { return h_; }

  // ---- the structured bridge (glc.h glc_frames_view / glc_frames_gather): EncodedAudio.frames as nested
  // vectors in one pass over the flat pools, and back by one pointer per vector - no byte stream
  static EncodedFrame frame_from_view(const glc_frames_view &v, uint64_t f) {
    EncodedFrame out;
    const uint64_t l0 = v.list_begin[f], l1 = v.list_begin[f + 1];
    out.sparse_coeffs_per_channel.resize(l1 - l0);
    for (uint64_t l = l0; l < l1; ++l) {
      auto &dst = out.sparse_coeffs_per_channel[l - l0];
      const uint64_t a = v.list_off[l], b = v.list_off[l + 1];
      dst.reserve(b - a);
      for (uint64_t j = a; j < b; ++j) dst.emplace_back(static_cast<uint16_t>(v.pairs[j] & 0xFFFFu), static_cast<int16_t>(v.pairs[j] >> 16));
    }
    if (v.scale_begin[f + 1] > v.scale_begin[f]) out.scale_factors.assign(v.scales + v.scale_begin[f], v.scales + v.scale_begin[f + 1]);
    out.has_raw_pcm = v.raw_tag[f] != 0;
    if (out.has_raw_pcm && v.raw_begin[f + 1] > v.raw_begin[f]) out.raw_pcm.assign(v.raw + v.raw_begin[f], v.raw + v.raw_begin[f + 1]);
    return out;
  }
  std::vector<EncodedFrame> frames() const {  // EncodedAudio.frames, src/codec.rs:35
    glc_frames_view v;
    detail::check(glc_frames_get_view(h_, &v));
    std::vector<EncodedFrame> out;
    out.reserve(v.n_frames);
    for (uint64_t f = 0; f < v.n_frames; ++f) out.push_back(frame_from_view(v, f));
    return out;
  }
  // EncodedAudio { header, frames, gapless_info } from the host's own nested vectors.  `stream_id`: 0, or the
  // caller's identity (< 2^63) of this stream's content - a Decoder that still holds it decodes it again
  // without uploading anything (glc.h glc_frames_from_parts).
  static EncodedAudio from_frames(const AudioHeader &h, const std::vector<EncodedFrame> &frames, const GaplessInfo &g,
                                  uint64_t stream_id = 0) {
    static_assert(sizeof(std::pair<uint16_t, int16_t>) == 4, "(u16, i16) pairs are the library's packed pairs");
    static const int16_t none[1] = {0};
    const size_t nf = frames.size();
    std::vector<uint32_t> lists_per(nf), list_len, scales_per(nf);
    std::vector<const void *> list_ptr;
    std::vector<const float *> scale_ptr(nf);
    std::vector<const int16_t *> raw_ptr(nf);
    std::vector<uint64_t> raw_len(nf);
    for (size_t f = 0; f < nf; ++f) {
      const EncodedFrame &fr = frames[f];
      lists_per[f] = static_cast<uint32_t>(fr.sparse_coeffs_per_channel.size());
      for (const auto &l : fr.sparse_coeffs_per_channel) list_ptr.push_back(l.data()), list_len.push_back(static_cast<uint32_t>(l.size()));
      scales_per[f] = static_cast<uint32_t>(fr.scale_factors.size());
      scale_ptr[f] = fr.scale_factors.data();
      raw_ptr[f] = fr.has_raw_pcm ? (fr.raw_pcm.empty() ? none : fr.raw_pcm.data()) : nullptr;
      raw_len[f] = fr.has_raw_pcm ? fr.raw_pcm.size() : 0;
    }
    glc_frames_gather gg{};
    gg.sample_rate = h.sample_rate, gg.channels = h.channels, gg.total_samples = h.total_samples;
    gg.encoder_delay = g.encoder_delay, gg.padding = g.padding, gg.original_length = g.original_length;
    gg.n_frames = nf;
    gg.lists_per_frame = lists_per.data(), gg.list_ptr = list_ptr.data(), gg.list_len = list_len.data();
    gg.scales_per_frame = scales_per.data(), gg.scale_ptr = scale_ptr.data(), gg.raw_ptr = raw_ptr.data(), gg.raw_len = raw_len.data();
    glc_frames *out = nullptr;
    detail::check(glc_frames_from_gather(&gg, stream_id, &out));
    return EncodedAudio(out);
  }
  uint64_t stream_id() const { return glc_frames_stream_id(h_); }

 private:
  glc_info info() const {
    glc_info i{};
    detail::check(glc_frames_info(h_, &i));
    return i;
  }
  glc_frames *h_ = nullptr;
};

class Encoder {
 public:
  // Encoder::new(sample_rate: u32) — src/codec.rs:406
  explicit Encoder(uint32_t sample_rate, int device = 0) { detail::check(glc_ctx_create(device, sample_rate, &ctx_)); }
  ~Encoder() { glc_ctx_destroy(ctx_); }
  Encoder(const Encoder &) = delete;
  Encoder &operator=(const Encoder &) = delete;
  // encode(&mut self, samples: &[f32], channels: u16) -> Result<EncodedAudio> — src/codec.rs:421
  EncodedAudio encode(const float *samples, uint64_t n_samples, uint16_t channels) {
    glc_frames *h = nullptr;
    detail::check(glc_encode(ctx_, samples, n_samples, channels, &h), ctx_);
    return EncodedAudio(h);
  }
  EncodedAudio encode(const std::vector<float> &samples, uint16_t channels) {
    return encode(samples.data(), samples.size(), channels);
  }
  // The same call handing the frames out as they arrive on the host, while the device still works on later
  // ones (glc_encode_hooked): `on_frames(first_frame, frames)` is called on this thread with ascending,
  // contiguous ranges - where an application fills its own Vec<EncodedFrame> under the encode instead of after it.
  EncodedAudio encode(const float *samples, uint64_t n_samples, uint16_t channels,
                      const std::function<void(uint64_t, std::vector<EncodedFrame> &&)> &on_frames) {
    struct Ctx {
      const std::function<void(uint64_t, std::vector<EncodedFrame> &&)> *fn;
      std::exception_ptr error;
    } c{&on_frames, nullptr};
    auto tramp = [](void *user, const glc_frames_view *v, uint64_t f0, uint64_t f1) -> int {
      Ctx &cx = *static_cast<Ctx *>(user);
      try {  // no exception may cross the C frames of the library
        std::vector<EncodedFrame> part;
        part.reserve(f1 - f0);
        for (uint64_t f = f0; f < f1; ++f) part.push_back(EncodedAudio::frame_from_view(*v, f));
        (*cx.fn)(f0, std::move(part));
        return 0;
      } catch (...) {
        cx.error = std::current_exception();
        return 1;
      }
    };
    glc_frames *h = nullptr;
    const int rc = glc_encode_hooked(ctx_, samples, n_samples, channels, tramp, &c, &h);
    if (c.error) std::rethrow_exception(c.error);
    detail::check(rc, ctx_);
    return EncodedAudio(h);
  }
  // One shard of an encode on device-resident PCM (multi-GPU hosts): frames [frame_begin, frame_end)
  // from the PCM slice [t0, t0 + t_count) per channel, fixed-size records to d_records
  // (glc_record_bytes(channels) each).  Queued on the context's stream; see glc_encode_range_device.
  void encode_range_device(const float *d_pcm, uint64_t t0, uint64_t t_count, uint64_t n_samples, uint16_t channels,
                           uint64_t frame_begin, uint64_t frame_end, void *d_records) {
    detail::check(glc_encode_range_device(ctx_, d_pcm, t0, t_count, n_samples, channels, frame_begin, frame_end,
                                          d_records, nullptr), ctx_);
  }
  // Assemble EncodedAudio from the records of all shards, gathered in frame order on this device.
  EncodedAudio frames_from_device_records(const void *d_records, uint64_t n_frames, uint64_t n_samples, uint16_t channels) {
    glc_frames *h = nullptr;
    detail::check(glc_frames_from_device_records(ctx_, d_records, n_frames, n_samples, channels, &h), ctx_);
    return EncodedAudio(h);
  }
  void synchronize() { detail::check(glc_ctx_synchronize(ctx_), ctx_); }
  glc_ctx *ctx() { return ctx_; }

 private:
  glc_ctx *ctx_ = nullptr;
};

class Decoder {
 public:
  // Decoder::new(channels: usize, sample_rate: u32) — src/codec.rs:581; `channels` is accepted and
  // ignored like the reference (the stream's header decides)
  Decoder(size_t /*channels*/, uint32_t sample_rate, int device = 0) {
    detail::check(glc_ctx_create(device, sample_rate, &ctx_));
  }
  ~Decoder() { glc_ctx_destroy(ctx_); }
  Decoder(const Decoder &) = delete;
  Decoder &operator=(const Decoder &) = delete;
  // decode(&mut self, &EncodedAudio, _) -> Result<Vec<f32>> — src/codec.rs:744
  std::vector<float> decode(const EncodedAudio &encoded) {
    std::vector<float> out(glc_decoded_len(encoded.handle()));
    uint64_t n = 0;
    detail::check(glc_decode(ctx_, encoded.handle(), out.data(), out.size(), &n), ctx_);
    out.resize(n);
    return out;
  }
  // The stream whose sparse rows this Decoder still holds on the device (0: none), and its decode without
  // an EncodedAudio (glc.h glc_decode_resident): for callers that recognise a stream by the id they gave it
  uint64_t resident_stream() const { return glc_ctx_resident_stream(ctx_); }
  std::vector<float> decode_resident(uint64_t stream_id, uint64_t decoded_len) {
    std::vector<float> out(decoded_len);
    uint64_t n = 0;
    detail::check(glc_decode_resident(ctx_, stream_id, out.data(), out.size(), &n), ctx_);
    out.resize(n);
    return out;
  }
  // One shard of a decode into device memory: hops [hop_begin, hop_end) of the un-trimmed stream
  // (n_frames + 1 hops of 1024 * channels samples); see glc_decode_range_device.
  void decode_range_device(const EncodedAudio &encoded, uint64_t hop_begin, uint64_t hop_end, float *d_out, uint64_t cap) {
    detail::check(glc_decode_range_device(ctx_, encoded.handle(), hop_begin, hop_end, d_out, cap), ctx_);
  }
  void synchronize() { detail::check(glc_ctx_synchronize(ctx_), ctx_); }
  // decode_streaming(&mut self, Arc<EncodedAudio>, _) -> Receiver<AudioChunk> — src/codec.rs:595:
  // the receiver becomes a callback invoked once per chunk, in order, the last one with is_last
  void decode_streaming(const EncodedAudio &encoded, const std::function<void(AudioChunk &&)> &on_chunk) {
    detail::check(glc_decode_stream_begin(ctx_, encoded.handle()), ctx_);
    const uint64_t cap = static_cast<uint64_t>(FRAMES_PER_CHUNK) * HOP_SIZE * encoded.header().channels;
    for (;;) {
      AudioChunk c{std::vector<float>(cap), false};
      uint64_t n = 0;
      int last = 0;
      detail::check(glc_decode_stream_next(ctx_, c.samples.data(), cap, &n, &last), ctx_);
      c.samples.resize(n);
      c.is_last = last != 0;
      on_chunk(std::move(c));
      if (last) return;
    }
  }

 private:
  glc_ctx *ctx_ = nullptr;
};

// save_encoded / load_encoded — src/codec.rs:774-786
inline void save_encoded(const EncodedAudio &e, const std::string &path) { detail::check(glc_save(e.handle(), path.c_str())); }
inline EncodedAudio load_encoded(const std::string &path) {
  glc_frames *h = nullptr;
  detail::check(glc_load(path.c_str(), &h));
  return EncodedAudio(h);
}


// ---- src/audio.rs + src/flac.rs twins (host only) -----------------------------------------------
struct LoadedAudio {  // the (Vec<f32>, u32, u16) tuple of load_audio_file_lossless
  std::vector<float> samples;
  uint32_t sample_rate;
  uint16_t channels;
};
namespace detail {
inline LoadedAudio take(float *p, uint64_t n, uint32_t sr, uint16_t ch) {
  LoadedAudio a{std::vector<float>(p, p + n), sr, ch};
  glc_free(p);
  return a;
}
inline std::string lower_ext(const std::string &path) {
  const size_t slash = path.find_last_of('/'), dot = path.find_last_of('.');
  if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
  std::string e = path.substr(dot + 1);
  for (char &c : e) c = static_cast<char>(c >= 'A' && c <= 'Z' ? c - 'A' + 'a' : c);
  return e;
}
}  // namespace detail
// load_audio_file_lossless — src/audio.rs:19-36
inline LoadedAudio load_audio_file_lossless(const std::string &path) {
  const std::string ext = detail::lower_ext(path);
  if (ext.empty()) throw Error(GLC_EINVAL, "No file extension");
  float *p = nullptr;
  uint64_t n = 0;
  uint32_t sr = 0;
  uint16_t ch = 0;
  if (ext == "wav") detail::check(glc_wav_load(path.c_str(), &p, &n, &sr, &ch));
  else if (ext == "flac") detail::check(glc_flac_load(path.c_str(), &p, &n, &sr, &ch));
  else throw Error(GLC_EINVAL, "Unsupported file format: " + ext);
  return detail::take(p, n, sr, ch);
}
// export_to_wav — src/audio.rs:100-132
inline void export_to_wav(const std::string &path, const std::vector<float> &s, uint32_t sample_rate, uint16_t channels) {
  detail::check(glc_wav_save16(path.c_str(), s.data(), s.size(), sample_rate, channels));
}
// encode_flac_with_level / encode_flac — src/flac.rs:947-1063
inline std::vector<uint8_t> encode_flac_with_level(const std::vector<float> &s, uint32_t sample_rate, uint16_t channels,
                                                   uint8_t compression_level) {
  uint8_t *p = nullptr;
  uint64_t n = 0;
  detail::check(glc_flac_encode(s.data(), s.size(), sample_rate, channels, compression_level, &p, &n));
  std::vector<uint8_t> out(p, p + n);
  glc_free(p);
  return out;
}
inline std::vector<uint8_t> encode_flac(const std::vector<float> &s, uint32_t sample_rate, uint16_t channels) {
  return encode_flac_with_level(s, sample_rate, channels, 5);
}
// export_to_flac_with_level / export_to_flac — src/flac.rs:1066-1087, src/audio.rs:87-96
inline void export_to_flac_with_level(const std::string &path, const std::vector<float> &s, uint32_t sample_rate,
                                      uint16_t channels, uint8_t compression_level) {
  detail::check(glc_flac_save(path.c_str(), s.data(), s.size(), sample_rate, channels, compression_level));
}
inline void export_to_flac(const std::string &path, const std::vector<float> &s, uint32_t sample_rate, uint16_t channels) {
  export_to_flac_with_level(path, s, sample_rate, channels, 5);
}

}  // namespace glc
