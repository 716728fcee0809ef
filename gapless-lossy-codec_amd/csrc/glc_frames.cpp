// glc_frames.cpp — EncodedAudio on the host: assembly from device records, the .glc container
// (bincode 1.x default options, as produced by src/codec.rs:774-786) and accessors.
//
// Wire format (from the serde derive order of the structs at src/codec.rs:31-69; bincode 1.x:
// little-endian fixed-width integers, u64 sequence lengths, one-byte Option tag):
//   u32 sample_rate | u16 channels | u64 total_samples
//   u64 n_frames, per frame:
//     u64 n_lists, per list: u64 n_pairs, n_pairs x (u16 index, i16 value)
//     u64 n_scales, n_scales x f32
//     u8 tag, if tag == 1: u64 n_raw, n_raw x i16
//   u32 encoder_delay | u32 padding | u64 original_length
#include <atomic>
#include <cstdio>
#include <memory>
#include <new>

#include "glc_common.h"

namespace {

struct Writer {
  uint8_t *p;
  uint64_t pos = 0;
  template <class T>
  void put(T v) {
    std::memcpy(p + pos, &v, sizeof(T));
    pos += sizeof(T);
  }
  void bytes(const void *src, uint64_t n) {
    if (n) std::memcpy(p + pos, src, n);
    pos += n;
  }
};

struct Reader {
  const uint8_t *p;
  uint64_t len;
  uint64_t pos = 0;
  bool ok = true;
  template <class T>
  T get() {
    T v{};
    if (!ok || len - pos < sizeof(T)) {
      ok = false;
      return v;
    }
    std::memcpy(&v, p + pos, sizeof(T));
    pos += sizeof(T);
    return v;
  }
  // sequence length followed by `n * elem` bytes must still fit in the buffer
  uint64_t seq_len(uint64_t elem) {
    uint64_t n = get<uint64_t>();
    if (ok && n > (len - pos) / elem) ok = false;
    return ok ? n : 0;
  }
};

}  // namespace

namespace glc {

uint64_t next_frames_uid() {
  static std::atomic<uint64_t> counter{0};
  // library-made identities live in the upper half of the id space; the lower half belongs to the
  // caller-supplied stream ids of glc_frames_from_parts / _gather
  return (counter.fetch_add(1, std::memory_order_relaxed) + 1) | (1ull << 63);
}

// Index vectors of EncodedAudio for the frames of ONE compact blob whose payload (pairs, raw planes)
// already sits in F->pairs / F->raw at p_at / r_at: per-frame raw tags, list offsets, scales.  Appends
// to F->list_off (which starts as {0}: every list adds its end offset, so the vector is complete after
// every blob - glc_encode_hooked hands out views of a stream that is still growing) and F->scales.  `meta`
// points at the blob's first o_pairs bytes (header + raw flags + scales + counts).  Untrusted blobs
// are checked for canonical (strictly ascending, < 1024) lists on the way.
int index_compact_meta(glc_frames *F, uint32_t ch, const CompactHeader &h, const uint8_t *meta, uint64_t f_at,
                       uint64_t p_at, uint64_t r_at, bool trusted, bool *canonical) {
  const CompactLayout l = compact_layout(ch, h.n_frames);
  const uint8_t *israw = meta + l.o_israw;
  const float *scale = reinterpret_cast<const float *>(meta + l.o_scale);
  const uint32_t *cnt = reinterpret_cast<const uint32_t *>(meta + l.o_cnt);
  const uint32_t *pairs = F->pairs.data() + p_at;
  uint64_t p_in = 0, raw_rows_in = 0;
  for (uint64_t f = 0; f < h.n_frames; ++f) {
    const uint64_t fo = f_at + f;
    F->raw_tag[fo] = israw[f] ? 1 : 0;
    if (israw[f]) {
      raw_rows_in += ch;
      for (uint32_t c = 0; c < ch; ++c)
        if (cnt[f * ch + c] != 0) {
          set_global_error("glc_frames_from_compact: raw frame with a sparse list");
          return GLC_EFORMAT;
        }
    } else {
      for (uint32_t c = 0; c < ch; ++c) {
        const uint32_t n = cnt[f * ch + c];
        if (n > kHop || p_in + n > h.n_pairs) {
          set_global_error("glc_frames_from_compact: corrupt blob (row counts exceed the pair pool)");
          return GLC_EFORMAT;
        }
        if (!trusted && *canonical) {
          int32_t last = -1;
          for (uint32_t j = 0; j < n; ++j) {
            const int32_t k = static_cast<int32_t>(pairs[p_in + j] & 0xFFFFu);
            if (k <= last || k >= static_cast<int32_t>(kHop)) {
              *canonical = false;
              break;
            }
            last = k;
          }
        }
        p_in += n;
        F->list_off.push_back(p_at + p_in);  // list_off = {0, end of list 0, end of list 1, ...}: the caller pushed the 0
        F->scales.push_back(scale[f * ch + c]);
      }
    }
    F->list_begin[fo + 1] = F->list_off.size() - 1;  // lists so far (list_off carries a leading 0)
    F->scale_begin[fo + 1] = F->scales.size();
    F->raw_begin[fo + 1] = r_at + raw_rows_in * kFrame;
  }
  if (p_in != h.n_pairs || raw_rows_in != h.n_raw_rows) {
    set_global_error("glc_frames_from_compact: corrupt blob (section totals disagree with the header)");
    return GLC_EFORMAT;
  }
  return GLC_OK;
}

int frames_from_compact(uint32_t sample_rate, uint64_t n_samples, uint16_t channels, const void *const *blobs,
                        const uint64_t *blob_bytes, uint32_t n_blobs, bool trusted, glc_frames **out) {
  if (!out || (n_blobs && (!blobs || !blob_bytes))) return GLC_EINVAL;
  *out = nullptr;
  const glc_plan plan = plan_encode(n_samples, channels);
  if (plan.n_frames == 0) {
    set_global_error("glc_frames_from_compact: the reference encoder panics on this stream length");
    return GLC_EINVAL;
  }
  const uint32_t ch = channels;
  // pass 1: validate every blob against its own header and the stream
  uint64_t n_frames = 0, n_pairs = 0, n_raw_rows = 0;
  std::vector<CompactHeader> hdrs(n_blobs);
  for (uint32_t b = 0; b < n_blobs; ++b) {
    if (!blobs[b] || blob_bytes[b] < sizeof(CompactHeader)) {
      set_global_error("glc_frames_from_compact: blob shorter than its header");
      return GLC_EFORMAT;
    }
    if (reinterpret_cast<uintptr_t>(blobs[b]) % 8 != 0) {  // the sections are read in place as u32 / f32 / u64
      set_global_error("glc_frames_from_compact: blob must be 8-byte aligned");
      return GLC_EINVAL;
    }
    CompactHeader &h = hdrs[b];
    std::memcpy(&h, blobs[b], sizeof h);
    if (h.magic != kCompactMagic || h.channels != ch || h.n_frames > plan.n_frames) {
      set_global_error("glc_frames_from_compact: bad magic, channel count or frame count");
      return GLC_EFORMAT;
    }
    const CompactLayout l = compact_layout(ch, h.n_frames);
    const uint64_t M = h.n_frames * ch;
    if (h.n_pairs > M * kHop || h.n_raw_rows > M || h.n_raw_rows % ch != 0) {
      set_global_error("glc_frames_from_compact: corrupt blob (pair / raw-row counts)");
      return GLC_EFORMAT;
    }
    const uint64_t need = compact_raw_offset(l, h.n_pairs) + h.n_raw_rows * kFrame * 2;
    if (h.bytes != need || blob_bytes[b] < need) {
      set_global_error("glc_frames_from_compact: blob size does not match its header");
      return GLC_EFORMAT;
    }
    n_frames += h.n_frames;
    n_pairs += h.n_pairs;
    n_raw_rows += h.n_raw_rows;
  }
  if (n_frames != plan.n_frames) {
    set_global_error("glc_frames_from_compact: the blobs do not add up to the stream's frame count");
    return GLC_EINVAL;
  }
  std::unique_ptr<glc_frames> F(new (std::nothrow) glc_frames);
  if (!F) return GLC_ENOMEM;
  F->sample_rate = sample_rate;
  F->channels = channels;
  F->total_samples = n_samples;           // src/codec.rs:423,555
  F->encoder_delay = plan.encoder_delay;  // :547
  F->padding = plan.padding;              // :546
  F->original_length = n_samples;         // :562
  F->n_frames = n_frames;
  bool canonical = true;
  try {
    F->list_begin.assign(n_frames + 1, 0);
    F->scale_begin.assign(n_frames + 1, 0);
    F->raw_begin.assign(n_frames + 1, 0);
    F->raw_tag.resize(n_frames);
    F->pairs.resize(n_pairs);
    F->raw.resize(n_raw_rows * kFrame);
    const uint64_t n_comp_rows = n_frames * ch - n_raw_rows;
    F->list_off.clear();
    F->list_off.reserve(n_comp_rows + 1);
    F->list_off.push_back(0);
    F->scales.clear();
    F->scales.reserve(n_comp_rows);
    uint64_t f_at = 0, p_at = 0, r_at = 0;  // frames / pairs / raw samples placed so far
    for (uint32_t b = 0; b < n_blobs; ++b) {
      const CompactHeader &h = hdrs[b];
      const CompactLayout l = compact_layout(ch, h.n_frames);
      const uint8_t *base = static_cast<const uint8_t *>(blobs[b]);
      const uint32_t *pairs = reinterpret_cast<const uint32_t *>(base + l.o_pairs);
      const int16_t *raw = reinterpret_cast<const int16_t *>(base + compact_raw_offset(l, h.n_pairs));
      if (h.n_pairs) std::memcpy(F->pairs.data() + p_at, pairs, h.n_pairs * 4);
      if (h.n_raw_rows) std::memcpy(F->raw.data() + r_at, raw, h.n_raw_rows * kFrame * 2);
      const int rc = index_compact_meta(F.get(), ch, h, base, f_at, p_at, r_at, trusted, &canonical);
      if (rc != GLC_OK) return rc;
      f_at += h.n_frames;
      p_at += h.n_pairs;
      r_at += h.n_raw_rows * kFrame;
    }
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  F->lists_canonical = canonical;
  *out = F.release();
  return GLC_OK;
}

}  // namespace glc

extern "C" {

uint64_t glc_compact_bound(uint16_t channels, uint64_t n_frames) {
  return glc::compact_layout(channels, n_frames).bound;
}

int glc_compact_records(const void *records, uint64_t n_frames, uint16_t channels, void *blob, uint64_t cap,
                        glc_compact_info *info) {
  if (!blob || !info || channels == 0 || (!records && n_frames)) return GLC_EINVAL;
  const uint32_t ch = channels;
  const glc::CompactLayout l = glc::compact_layout(ch, n_frames);
  if (cap < l.bound) {
    glc::set_global_error("glc_compact_records: blob buffer smaller than glc_compact_bound()");
    return GLC_EINVAL;
  }
  const uint64_t rec = glc::record_bytes(ch), hdr = glc::record_header_bytes(ch);
  uint8_t *out = static_cast<uint8_t *>(blob);
  std::memset(out, 0, l.o_pairs);
  uint8_t *israw = out + l.o_israw;
  float *scale = reinterpret_cast<float *>(out + l.o_scale);
  uint32_t *cnt = reinterpret_cast<uint32_t *>(out + l.o_cnt);
  uint32_t *pairs = reinterpret_cast<uint32_t *>(out + l.o_pairs);
  const uint8_t *base = static_cast<const uint8_t *>(records);
  uint64_t n_pairs = 0, n_raw_rows = 0;
  for (uint64_t f = 0; f < n_frames; ++f) {
    const uint8_t *r = base + f * rec;
    uint32_t raw;
    std::memcpy(&raw, r, 4);
    israw[f] = raw ? 1 : 0;
    for (uint32_t c = 0; c < ch; ++c) {
      uint32_t nnz;
      std::memcpy(&scale[f * ch + c], r + 8 + 8 * c, 4);
      std::memcpy(&nnz, r + 8 + 8 * c + 4, 4);
      if (nnz > glc::kHop) nnz = glc::kHop;
      cnt[f * ch + c] = raw ? 0 : nnz;
      if (raw) {
        ++n_raw_rows;
        continue;
      }
      const int16_t *row = reinterpret_cast<const int16_t *>(r + hdr) + static_cast<size_t>(c) * glc::kFrame;
      uint32_t done = 0;
      for (uint32_t k = 0; k < glc::kHop && done < nnz; ++k)
        if (row[k] != 0) pairs[n_pairs + done++] = k | (static_cast<uint32_t>(static_cast<uint16_t>(row[k])) << 16);
      for (; done < nnz; ++done) pairs[n_pairs + done] = 0xFFFFu;  // same filler as the device kernel
      n_pairs += nnz;
    }
  }
  const uint64_t raw_off = glc::compact_raw_offset(l, n_pairs);
  std::memset(out + l.o_pairs + 4 * n_pairs, 0, raw_off - (l.o_pairs + 4 * n_pairs));
  int16_t *rawp = reinterpret_cast<int16_t *>(out + raw_off);
  uint64_t rr = 0;
  for (uint64_t f = 0; f < n_frames; ++f) {
    if (!israw[f]) continue;
    std::memcpy(rawp + rr * glc::kFrame, base + f * rec + hdr, sizeof(int16_t) * glc::kFrame * ch);
    rr += ch;
  }
  glc::CompactHeader h{};
  h.magic = glc::kCompactMagic;
  h.channels = ch;
  h.n_frames = n_frames;
  h.n_pairs = n_pairs;
  h.n_raw_rows = n_raw_rows;
  h.bytes = raw_off + n_raw_rows * glc::kFrame * 2;
  std::memcpy(out, &h, sizeof h);
  info->n_frames = n_frames;
  info->n_pairs = n_pairs;
  info->n_raw_rows = n_raw_rows;
  info->bytes = h.bytes;
  return GLC_OK;
}

int glc_frames_from_compact(uint32_t sample_rate, uint64_t n_samples, uint16_t channels, const void *const *blobs,
                            const uint64_t *blob_bytes, uint32_t n_blobs, glc_frames **out) {
  return glc::frames_from_compact(sample_rate, n_samples, channels, blobs, blob_bytes, n_blobs, false, out);
}

uint64_t glc_record_bytes(uint16_t channels) { return glc::record_bytes(channels); }

int glc_plan_encode(uint64_t n_samples, uint16_t channels, glc_plan *out) {
  if (!out) return GLC_EINVAL;
  *out = glc::plan_encode(n_samples, channels);
  if (out->n_frames == 0) {
    glc::set_global_error("glc_plan_encode: the reference encoder panics on this input "
                          "(channels == 0, <= 512 samples per channel, or ragged channels)");
    return GLC_EINVAL;
  }
  return GLC_OK;
}

int glc_frames_from_records(uint32_t sample_rate, uint64_t n_samples, uint16_t channels,
                            const void *records, uint64_t n_frames, glc_frames **out) {
  if (!records || !out) return GLC_EINVAL;
  const glc_plan plan = glc::plan_encode(n_samples, channels);
  if (plan.n_frames == 0 || plan.n_frames != n_frames) {
    glc::set_global_error("glc_frames_from_records: record count does not match the stream length");
    return GLC_EINVAL;
  }
  std::unique_ptr<glc_frames> F(new (std::nothrow) glc_frames);
  if (!F) return GLC_ENOMEM;
  const uint32_t ch = channels;
  const uint64_t rec = glc::record_bytes(ch), hdr = glc::record_header_bytes(ch);
  F->sample_rate = sample_rate;
  F->channels = channels;
  F->total_samples = n_samples;           // src/codec.rs:423,555
  F->encoder_delay = plan.encoder_delay;  // :547
  F->padding = plan.padding;              // :546
  F->original_length = n_samples;         // :562
  F->n_frames = n_frames;
  try {
    F->list_begin.assign(n_frames + 1, 0);
    F->scale_begin.assign(n_frames + 1, 0);
    F->raw_begin.assign(n_frames + 1, 0);
    F->raw_tag.assign(n_frames, 0);
    // pass 1: sizes
    uint64_t n_lists = 0, n_pairs = 0, n_scales = 0, n_raw = 0;
    const uint8_t *base = static_cast<const uint8_t *>(records);
    for (uint64_t f = 0; f < n_frames; ++f) {
      const uint8_t *r = base + f * rec;
      uint32_t is_raw;
      std::memcpy(&is_raw, r, 4);
      F->raw_tag[f] = is_raw ? 1 : 0;
      if (is_raw) {
        n_raw += static_cast<uint64_t>(glc::kFrame) * ch;  // FRAME_SIZE per channel, :469
      } else {
        n_lists += ch;
        n_scales += ch;
        for (uint32_t c = 0; c < ch; ++c) {
          uint32_t nnz;
          std::memcpy(&nnz, r + 8 + 8 * c + 4, 4);
          if (nnz > glc::kHop) {
            glc::set_global_error("glc_frames_from_records: corrupt record (nnz > 1024)");
            return GLC_EINVAL;
          }
          n_pairs += nnz;
        }
      }
      F->list_begin[f + 1] = n_lists;
      F->scale_begin[f + 1] = n_scales;
      F->raw_begin[f + 1] = n_raw;
    }
    F->list_off.assign(n_lists + 1, 0);
    F->pairs.resize(n_pairs);
    F->scales.resize(n_scales);
    F->raw.resize(n_raw);
    // pass 2: payload.  Ascending-k scan of the dense row == push order at :285-308.
    uint64_t li = 0, pi = 0, si = 0, ri = 0;
    for (uint64_t f = 0; f < n_frames; ++f) {
      const uint8_t *r = base + f * rec;
      const int16_t *payload = reinterpret_cast<const int16_t *>(r + hdr);
      if (F->raw_tag[f]) {
        std::memcpy(&F->raw[ri], payload, sizeof(int16_t) * glc::kFrame * ch);  // planar, Q1
        ri += static_cast<uint64_t>(glc::kFrame) * ch;
        continue;
      }
      for (uint32_t c = 0; c < ch; ++c) {
        float scale;
        uint32_t nnz;
        std::memcpy(&scale, r + 8 + 8 * c, 4);
        std::memcpy(&nnz, r + 8 + 8 * c + 4, 4);
        F->scales[si++] = scale;
        const int16_t *row = payload + static_cast<size_t>(c) * glc::kFrame;
        const uint64_t start = pi;
        for (uint32_t k = 0; k < glc::kHop && pi - start < nnz; ++k) {
          const int16_t q = row[k];
          if (q != 0) F->pairs[pi++] = static_cast<uint32_t>(k) | (static_cast<uint32_t>(static_cast<uint16_t>(q)) << 16);
        }
        if (pi - start != nnz) {
          glc::set_global_error("glc_frames_from_records: corrupt record (nnz mismatch)");
          return GLC_EINVAL;
        }
        F->list_off[++li] = pi;
      }
    }
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  F->lists_canonical = true;  // built by an ascending scan of dense rows
  *out = F.release();
  return GLC_OK;
}

uint64_t glc_serialized_size(const glc_frames *f) {
  if (!f) return 0;
  uint64_t n = 4 + 2 + 8 + 8;
  n += f->n_frames * (8 + 8 + 1);
  n += (f->list_off.size() - 1) * 8;
  n += f->pairs.size() * 4 + f->scales.size() * 4 + f->raw.size() * 2;
  for (uint64_t i = 0; i < f->n_frames; ++i) n += f->raw_tag[i] ? 8 : 0;
  n += 4 + 4 + 8;
  return n;
}

int glc_serialize(const glc_frames *f, uint8_t *buf, uint64_t cap, uint64_t *written) {
  if (!f || !buf) return GLC_EINVAL;
  const uint64_t need = glc_serialized_size(f);
  if (cap < need) {
    glc::set_global_error("glc_serialize: buffer too small");
    return GLC_EINVAL;
  }
  Writer w{buf};
  w.put<uint32_t>(f->sample_rate);
  w.put<uint16_t>(f->channels);
  w.put<uint64_t>(f->total_samples);
  w.put<uint64_t>(f->n_frames);
  for (uint64_t i = 0; i < f->n_frames; ++i) {
    const uint64_t l0 = f->list_begin[i], l1 = f->list_begin[i + 1];
    w.put<uint64_t>(l1 - l0);
    for (uint64_t l = l0; l < l1; ++l) {
      const uint64_t a = f->list_off[l], b = f->list_off[l + 1];
      w.put<uint64_t>(b - a);
      w.bytes(f->pairs.data() + a, (b - a) * 4);  // (u16 LE, i16 LE) == packed u32 LE
    }
    const uint64_t s0 = f->scale_begin[i], s1 = f->scale_begin[i + 1];
    w.put<uint64_t>(s1 - s0);
    w.bytes(f->scales.data() + s0, (s1 - s0) * 4);
    w.put<uint8_t>(f->raw_tag[i]);
    if (f->raw_tag[i]) {
      const uint64_t r0 = f->raw_begin[i], r1 = f->raw_begin[i + 1];
      w.put<uint64_t>(r1 - r0);
      w.bytes(f->raw.data() + r0, (r1 - r0) * 2);
    }
  }
  w.put<uint32_t>(f->encoder_delay);
  w.put<uint32_t>(f->padding);
  w.put<uint64_t>(f->original_length);
  if (written) *written = w.pos;
  return w.pos == need ? GLC_OK : GLC_EFORMAT;
}

int glc_deserialize(const uint8_t *buf, uint64_t len, glc_frames **out) {
  if (!buf || !out) return GLC_EINVAL;
  std::unique_ptr<glc_frames> F(new (std::nothrow) glc_frames);
  if (!F) return GLC_ENOMEM;
  Reader r{buf, len};
  F->sample_rate = r.get<uint32_t>();
  F->channels = r.get<uint16_t>();
  F->total_samples = r.get<uint64_t>();
  const uint64_t nf = r.seq_len(8 + 8 + 1);  // every frame costs at least 17 bytes
  if (!r.ok) {
    glc::set_global_error("glc_deserialize: truncated header");
    return GLC_EFORMAT;
  }
  try {
    F->n_frames = nf;
    F->list_begin.assign(nf + 1, 0);
    F->scale_begin.assign(nf + 1, 0);
    F->raw_begin.assign(nf + 1, 0);
    F->raw_tag.assign(nf, 0);
    F->list_off.assign(1, 0);
    for (uint64_t i = 0; i < nf && r.ok; ++i) {
      const uint64_t nl = r.seq_len(8);
      for (uint64_t l = 0; l < nl && r.ok; ++l) {
        const uint64_t np = r.seq_len(4);
        if (!r.ok) break;
        const size_t at = F->pairs.size();
        F->pairs.resize(at + np);
        if (np) std::memcpy(F->pairs.data() + at, buf + r.pos, np * 4);
        r.pos += np * 4;
        F->list_off.push_back(F->pairs.size());
      }
      F->list_begin[i + 1] = F->list_off.size() - 1;
      const uint64_t ns = r.seq_len(4);
      if (!r.ok) break;
      const size_t sat = F->scales.size();
      F->scales.resize(sat + ns);
      if (ns) std::memcpy(F->scales.data() + sat, buf + r.pos, ns * 4);
      r.pos += ns * 4;
      F->scale_begin[i + 1] = F->scales.size();
      const uint8_t tag = r.get<uint8_t>();
      if (tag > 1) r.ok = false;  // bincode rejects any other Option tag
      if (!r.ok) break;
      F->raw_tag[i] = tag;
      if (tag) {
        const uint64_t nr = r.seq_len(2);
        if (!r.ok) break;
        const size_t rat = F->raw.size();
        F->raw.resize(rat + nr);
        if (nr) std::memcpy(F->raw.data() + rat, buf + r.pos, nr * 2);
        r.pos += nr * 2;
      }
      F->raw_begin[i + 1] = F->raw.size();
    }
    F->encoder_delay = r.get<uint32_t>();
    F->padding = r.get<uint32_t>();
    F->original_length = r.get<uint64_t>();
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  if (!r.ok) {
    glc::set_global_error("glc_deserialize: malformed or truncated .glc stream");
    return GLC_EFORMAT;
  }
  // bincode::deserialize (not deserialize_from) ignores trailing bytes; so do we.
  *out = F.release();
  return GLC_OK;
}

int glc_save(const glc_frames *f, const char *path) {
  if (!f || !path) return GLC_EINVAL;
  const uint64_t n = glc_serialized_size(f);
  std::vector<uint8_t> buf;
  try {  // no C++ exception may cross the C ABI
    buf.resize(n);
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  uint64_t w = 0;
  int rc = glc_serialize(f, buf.data(), n, &w);
  if (rc != GLC_OK) return rc;
  FILE *fp = std::fopen(path, "wb");
  if (!fp) {
    glc::set_global_error(std::string("glc_save: cannot open ") + path);
    return GLC_EIO;
  }
  const size_t put = std::fwrite(buf.data(), 1, w, fp);
  const int cl = std::fclose(fp);
  if (put != w || cl != 0) {
    glc::set_global_error(std::string("glc_save: short write to ") + path);
    return GLC_EIO;
  }
  return GLC_OK;
}

int glc_load(const char *path, glc_frames **out) {
  if (!path || !out) return GLC_EINVAL;
  FILE *fp = std::fopen(path, "rb");
  if (!fp) {
    glc::set_global_error(std::string("glc_load: cannot open ") + path);
    return GLC_EIO;
  }
  std::vector<uint8_t> buf;
  uint8_t tmp[1 << 16];
  size_t got;
  try {
    while ((got = std::fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + got);
  } catch (const std::bad_alloc &) {
    std::fclose(fp);
    return GLC_ENOMEM;
  }
  std::fclose(fp);
  return glc_deserialize(buf.data(), buf.size(), out);
}

void glc_frames_free(glc_frames *f) { delete f; }

int glc_frames_info(const glc_frames *f, glc_info *out) {
  if (!f || !out) return GLC_EINVAL;
  std::memset(out, 0, sizeof *out);
  out->sample_rate = f->sample_rate;
  out->channels = f->channels;
  out->total_samples = f->total_samples;
  out->encoder_delay = f->encoder_delay;
  out->padding = f->padding;
  out->original_length = f->original_length;
  out->n_frames = f->n_frames;
  for (uint64_t i = 0; i < f->n_frames; ++i) out->n_raw_frames += f->raw_tag[i];
  out->total_nnz = f->pairs.size();
  return GLC_OK;
}

int glc_frame_is_raw(const glc_frames *f, uint64_t frame) {
  if (!f || frame >= f->n_frames) return GLC_EINVAL;
  return f->raw_tag[frame];
}

int glc_frame_sparse(const glc_frames *f, uint64_t frame, uint32_t channel, uint16_t *idx,
                     int16_t *q, uint32_t cap, uint32_t *n) {
  if (!f || frame >= f->n_frames) return GLC_EINVAL;
  const uint64_t l = f->list_begin[frame] + channel;
  if (l >= f->list_begin[frame + 1]) return GLC_EINVAL;
  const uint64_t a = f->list_off[l], b = f->list_off[l + 1];
  if (n) *n = static_cast<uint32_t>(b - a);
  for (uint64_t j = a; j < b && j - a < cap; ++j) {
    if (idx) idx[j - a] = static_cast<uint16_t>(f->pairs[j] & 0xFFFFu);
    if (q) q[j - a] = static_cast<int16_t>(f->pairs[j] >> 16);
  }
  return GLC_OK;
}

int glc_frame_scale(const glc_frames *f, uint64_t frame, uint32_t channel, float *scale) {
  if (!f || !scale || frame >= f->n_frames) return GLC_EINVAL;
  const uint64_t s = f->scale_begin[frame] + channel;
  if (s >= f->scale_begin[frame + 1]) return GLC_EINVAL;
  *scale = f->scales[s];
  return GLC_OK;
}

int glc_frame_raw(const glc_frames *f, uint64_t frame, int16_t *pcm, uint64_t cap, uint64_t *n) {
  if (!f || frame >= f->n_frames || !f->raw_tag[frame]) return GLC_EINVAL;
  const uint64_t a = f->raw_begin[frame], b = f->raw_begin[frame + 1];
  if (n) *n = b - a;
  const uint64_t take = (b - a) < cap ? (b - a) : cap;
  if (pcm && take) std::memcpy(pcm, f->raw.data() + a, sizeof(int16_t) * take);  // (Some(vec![]) in an empty pool: no null memcpy)
  return GLC_OK;
}

// ---- the structured bridge: EncodedAudio as flat arrays (include/glc.h) ----------------------

int glc_frames_get_view(const glc_frames *f, glc_frames_view *out) {
  if (!f || !out) return GLC_EINVAL;
  std::memset(out, 0, sizeof *out);
  out->sample_rate = f->sample_rate;
  out->channels = f->channels;
  out->total_samples = f->total_samples;
  out->encoder_delay = f->encoder_delay;
  out->padding = f->padding;
  out->original_length = f->original_length;
  out->n_frames = f->n_frames;
  out->n_lists = f->list_off.empty() ? 0 : f->list_off.size() - 1;
  out->n_pairs = f->pairs.size();
  out->n_scales = f->scales.size();
  out->n_raw = f->raw.size();
  out->list_begin = f->list_begin.data();
  out->list_off = f->list_off.data();
  out->pairs = f->pairs.data();
  out->scale_begin = f->scale_begin.data();
  out->scales = f->scales.data();
  out->raw_tag = f->raw_tag.data();
  out->raw_begin = f->raw_begin.data();
  out->raw = f->raw.data();
  return GLC_OK;
}

namespace {

// offsets[0 .. n] must start at 0, never decrease and end at `total`
bool offsets_ok(const uint64_t *off, uint64_t n, uint64_t total) {
  if (!off || off[0] != 0 || off[n] != total) return false;
  for (uint64_t i = 0; i < n; ++i)
    if (off[i] > off[i + 1]) return false;
  return true;
}

int assign_uid(glc_frames *F, uint64_t stream_id) {
  if (stream_id >> 63) {
    glc::set_global_error("stream_id must be below 2^63 (the upper half identifies library-made objects)");
    return GLC_EINVAL;
  }
  if (stream_id) F->uid = stream_id;
  return GLC_OK;
}

}  // namespace

int glc_frames_from_parts(const glc_frames_view *p, uint64_t stream_id, glc_frames **out) {
  if (!p || !out) return GLC_EINVAL;
  *out = nullptr;
  const uint64_t nf = p->n_frames;
  // every count must be addressable before anything is read through it
  if (nf > (1ull << 40) || p->n_lists > (1ull << 44) || p->n_pairs > (1ull << 46) || p->n_scales > (1ull << 44) ||
      p->n_raw > (1ull << 46)) {
    glc::set_global_error("glc_frames_from_parts: absurd counts");
    return GLC_EFORMAT;
  }
  if (!p->list_begin || !p->list_off || !p->scale_begin || !p->raw_begin || (nf && !p->raw_tag) ||
      (p->n_pairs && !p->pairs) || (p->n_scales && !p->scales) || (p->n_raw && !p->raw)) {
    glc::set_global_error("glc_frames_from_parts: null array");
    return GLC_EINVAL;
  }
  if (!offsets_ok(p->list_begin, nf, p->n_lists) || !offsets_ok(p->list_off, p->n_lists, p->n_pairs) ||
      !offsets_ok(p->scale_begin, nf, p->n_scales) || !offsets_ok(p->raw_begin, nf, p->n_raw)) {
    glc::set_global_error("glc_frames_from_parts: offsets are not monotonic or do not span their pool");
    return GLC_EFORMAT;
  }
  for (uint64_t f = 0; f < nf; ++f)
    if (p->raw_tag[f] > 1 || (!p->raw_tag[f] && p->raw_begin[f + 1] != p->raw_begin[f])) {
      glc::set_global_error("glc_frames_from_parts: raw_tag is not 0 / 1, or a frame without raw_pcm owns raw samples");
      return GLC_EFORMAT;
    }
  std::unique_ptr<glc_frames> F(new (std::nothrow) glc_frames);
  if (!F) return GLC_ENOMEM;
  const int rc = assign_uid(F.get(), stream_id);
  if (rc != GLC_OK) return rc;
  try {
    F->sample_rate = p->sample_rate;
    F->channels = p->channels;
    F->total_samples = p->total_samples;
    F->encoder_delay = p->encoder_delay;
    F->padding = p->padding;
    F->original_length = p->original_length;
    F->n_frames = nf;
    F->list_begin.assign(p->list_begin, p->list_begin + nf + 1);
    F->list_off.assign(p->list_off, p->list_off + p->n_lists + 1);
    F->pairs.assign(p->pairs, p->pairs + p->n_pairs);
    F->scale_begin.assign(p->scale_begin, p->scale_begin + nf + 1);
    F->scales.assign(p->scales, p->scales + p->n_scales);
    F->raw_tag.assign(p->raw_tag, p->raw_tag + nf);
    F->raw_begin.assign(p->raw_begin, p->raw_begin + nf + 1);
    F->raw.assign(p->raw, p->raw + p->n_raw);
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  F->lists_canonical = false;  // checked at decode
  *out = F.release();
  return GLC_OK;
}

int glc_frames_from_gather(const glc_frames_gather *g, uint64_t stream_id, glc_frames **out) {
  if (!g || !out) return GLC_EINVAL;
  *out = nullptr;
  const uint64_t nf = g->n_frames;
  if (nf > (1ull << 40)) {
    glc::set_global_error("glc_frames_from_gather: absurd frame count");
    return GLC_EFORMAT;
  }
  if (nf && (!g->lists_per_frame || !g->scales_per_frame || !g->scale_ptr || !g->raw_ptr || !g->raw_len)) {
    glc::set_global_error("glc_frames_from_gather: null array");
    return GLC_EINVAL;
  }
  std::unique_ptr<glc_frames> F(new (std::nothrow) glc_frames);
  if (!F) return GLC_ENOMEM;
  const int rc = assign_uid(F.get(), stream_id);
  if (rc != GLC_OK) return rc;
  try {
    F->sample_rate = g->sample_rate;
    F->channels = g->channels;
    F->total_samples = g->total_samples;
    F->encoder_delay = g->encoder_delay;
    F->padding = g->padding;
    F->original_length = g->original_length;
    F->n_frames = nf;
    F->list_begin.assign(nf + 1, 0);
    F->scale_begin.assign(nf + 1, 0);
    F->raw_begin.assign(nf + 1, 0);
    F->raw_tag.assign(nf, 0);
    // pass 1: sizes
    uint64_t n_lists = 0, n_pairs = 0, n_scales = 0, n_raw = 0;
    for (uint64_t f = 0; f < nf; ++f) {
      const uint32_t nl = g->lists_per_frame[f];
      if (nl && (!g->list_ptr || !g->list_len)) {
        glc::set_global_error("glc_frames_from_gather: null list array");
        return GLC_EINVAL;
      }
      for (uint32_t l = 0; l < nl; ++l) {
        if (g->list_len[n_lists + l] && !g->list_ptr[n_lists + l]) {
          glc::set_global_error("glc_frames_from_gather: null list pointer with a non-zero length");
          return GLC_EINVAL;
        }
        n_pairs += g->list_len[n_lists + l];
      }
      n_lists += nl;
      if (g->scales_per_frame[f] && !g->scale_ptr[f]) {
        glc::set_global_error("glc_frames_from_gather: null scale pointer with a non-zero length");
        return GLC_EINVAL;
      }
      n_scales += g->scales_per_frame[f];
      if (g->raw_ptr[f]) {
        F->raw_tag[f] = 1;
        n_raw += g->raw_len[f];
      }
      F->list_begin[f + 1] = n_lists;
      F->scale_begin[f + 1] = n_scales;
      F->raw_begin[f + 1] = n_raw;
    }
    F->list_off.resize(n_lists + 1);
    F->pairs.resize(n_pairs);
    F->scales.resize(n_scales);
    F->raw.resize(n_raw);
    // pass 2: payload, one copy per vector
    uint64_t li = 0, pi = 0;
    for (uint64_t f = 0; f < nf; ++f) {
      const uint32_t nl = g->lists_per_frame[f];
      for (uint32_t l = 0; l < nl; ++l, ++li) {
        F->list_off[li] = pi;
        const uint32_t n = g->list_len[li];
        if (n) std::memcpy(F->pairs.data() + pi, g->list_ptr[li], static_cast<size_t>(n) * 4);
        pi += n;
      }
      const uint32_t ns = g->scales_per_frame[f];
      if (ns) std::memcpy(F->scales.data() + F->scale_begin[f], g->scale_ptr[f], static_cast<size_t>(ns) * 4);
      if (g->raw_ptr[f] && g->raw_len[f])
        std::memcpy(F->raw.data() + F->raw_begin[f], g->raw_ptr[f], g->raw_len[f] * 2);
    }
    F->list_off[n_lists] = pi;
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  F->lists_canonical = false;  // checked at decode
  *out = F.release();
  return GLC_OK;
}

uint64_t glc_frames_stream_id(const glc_frames *f) { return f ? f->uid : 0; }

uint64_t glc_decoded_len(const glc_frames *f) {
  if (!f) return 0;
  // src/codec.rs:756-765
  uint64_t all = (f->n_frames + 1) * static_cast<uint64_t>(glc::kHop) * f->channels;
  if (all > f->encoder_delay) all -= f->encoder_delay;
  if (all > f->original_length) all = f->original_length;
  return all;
}

const char *glc_version(void) { return "glc-mi355x 0.1 (reference gapless-lossy-codec 0.5.0)"; }

}  // extern "C"
