// host_fuzz.cpp — AddressSanitizer + UBSan harness for the host-side parsers and writers of the
// library (no device code): the .glc container and the structured bridge (glc_frames.cpp), the WAV twin
// (glc_wav.cpp) and the FLAC twin (glc_flac.cpp).  GPU sanitizers are not available on the pool, host ones are, and
// these are the functions that read files a user did not write.
//
// Build (tests/test_host.py does this):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude \
//       tools/host_fuzz.cpp gapless-lossy-codec_amd/csrc/{glc_frames,glc_tables,glc_wav,glc_flac}.cpp \
//       -lpthread -o build/host_fuzz
// Run: build/host_fuzz <seconds> <tmp dir>     exit 0 = no finding, no property violated
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "glc.h"

namespace glc {
void set_global_error(const std::string &) {}  // the real one lives next to the device code
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static uint64_t below(uint64_t n) { return n ? rnd() % n : 0; }

#define REQUIRE(cond)                                                        \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "host_fuzz: property violated at %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

static void mutate(std::vector<uint8_t> &b) {
  if (b.empty()) return;
  switch (below(6)) {
    case 0: b[below(b.size())] ^= static_cast<uint8_t>(1u << below(8)); break;
    case 1: b[below(b.size())] = static_cast<uint8_t>(rnd()); break;
    case 2: b.resize(below(b.size())); break;
    case 3: {  // overwrite a little-endian u64 (a bincode length) with something absurd
      if (b.size() >= 8) {
        const uint64_t v = below(4) ? rnd() >> below(64) : ~0ull - below(16);
        std::memcpy(&b[below(b.size() - 7)], &v, 8);
      }
      break;
    }
    case 4: b.insert(b.begin() + below(b.size()), static_cast<uint8_t>(rnd())); break;
    default: b.erase(b.begin() + below(b.size())); break;
  }
}

// ---- .glc container ---------------------------------------------------------------------------
static std::vector<uint8_t> make_glc(unsigned ch, unsigned n_frames) {
  // records as the device writes them: header {is_raw, scale, nnz per channel} + dense i16 payload
  const uint64_t rb = glc_record_bytes(static_cast<uint16_t>(ch));
  const uint64_t hdr = rb - 4096ull * ch;
  std::vector<uint8_t> rec(rb * n_frames, 0);
  for (unsigned f = 0; f < n_frames; ++f) {
    uint8_t *r = rec.data() + rb * f;
    const bool raw = below(5) == 0;
    uint32_t is_raw = raw;
    std::memcpy(r, &is_raw, 4);
    int16_t *pay = reinterpret_cast<int16_t *>(r + hdr);
    for (unsigned c = 0; c < ch; ++c) {
      float scale = static_cast<float>(below(1000)) / 999.0f;
      uint32_t nnz = 0;
      if (raw) {
        for (unsigned i = 0; i < 2048; ++i) pay[c * 2048 + i] = static_cast<int16_t>(rnd());
      } else {
        for (unsigned k = 0; k < 1024; ++k)
          if (below(8) == 0) pay[c * 2048 + k] = static_cast<int16_t>(rnd() | 1), ++nnz;
      }
      std::memcpy(r + 8 + 8 * c, &scale, 4);
      std::memcpy(r + 8 + 8 * c + 4, &nnz, 4);
    }
  }
  const uint64_t n_samples = static_cast<uint64_t>(n_frames) * 1024 * ch;  // gives exactly n_frames
  glc_frames *fr = nullptr;
  std::vector<uint8_t> out;
  if (glc_frames_from_records(44100, n_samples, static_cast<uint16_t>(ch), rec.data(), n_frames, &fr) != GLC_OK) return out;
  out.resize(glc_serialized_size(fr));
  uint64_t w = 0;
  REQUIRE(glc_serialize(fr, out.data(), out.size(), &w) == GLC_OK && w == out.size());
  glc_frames_free(fr);
  return out;
}

static void touch(const glc_frames *fr) {
  glc_info info;
  REQUIRE(glc_frames_info(fr, &info) == GLC_OK);
  (void)glc_decoded_len(fr);
  const uint64_t nf = info.n_frames < 64 ? info.n_frames : 64;
  std::vector<uint16_t> idx(2048);
  std::vector<int16_t> q(2048), raw(1 << 16);
  for (uint64_t f = 0; f < nf; ++f) {
    (void)glc_frame_is_raw(fr, f);
    for (uint32_t c = 0; c < 4; ++c) {
      uint32_t n = 0;
      float s;
      (void)glc_frame_sparse(fr, f, c, idx.data(), q.data(), 2048, &n);
      (void)glc_frame_scale(fr, f, c, &s);
    }
    uint64_t n = 0;
    (void)glc_frame_raw(fr, f, raw.data(), raw.size(), &n);
  }
}

static void fuzz_glc(const std::string &tmp) {
  std::vector<uint8_t> good = make_glc(1 + static_cast<unsigned>(below(3)), 1 + static_cast<unsigned>(below(6)));
  if (good.empty()) return;
  {  // an accepted stream re-serialises to the same bytes
    glc_frames *fr = nullptr;
    REQUIRE(glc_deserialize(good.data(), good.size(), &fr) == GLC_OK);
    std::vector<uint8_t> again(glc_serialized_size(fr));
    uint64_t w = 0;
    REQUIRE(glc_serialize(fr, again.data(), again.size(), &w) == GLC_OK && again == good);
    touch(fr);
    const std::string path = tmp + "/fuzz.glc";
    REQUIRE(glc_save(fr, path.c_str()) == GLC_OK);
    glc_frames *back = nullptr;
    REQUIRE(glc_load(path.c_str(), &back) == GLC_OK);
    glc_frames_free(back);
    glc_frames_free(fr);
  }
  for (int round = 0; round < 40; ++round) {
    std::vector<uint8_t> bad = good;
    for (uint64_t m = 1 + below(3); m; --m) mutate(bad);
    glc_frames *fr = nullptr;
    const int rc = glc_deserialize(bad.data(), bad.size(), &fr);
    if (rc == GLC_OK) {  // whatever parses must round-trip byte for byte and be safe to walk
      std::vector<uint8_t> again(glc_serialized_size(fr));
      uint64_t w = 0;
      REQUIRE(glc_serialize(fr, again.data(), again.size(), &w) == GLC_OK);
      REQUIRE(again.size() <= bad.size() && std::memcmp(again.data(), bad.data(), again.size()) == 0);
      touch(fr);
      glc_frames_free(fr);
    } else {
      REQUIRE(fr == nullptr && (rc == GLC_EFORMAT || rc == GLC_ENOMEM || rc == GLC_EINVAL));
    }
  }
}

// ---- the structured bridge: flat view, from_parts / from_gather on hostile arrays ------------------
static void fuzz_bridge() {
  std::vector<uint8_t> good = make_glc(1 + static_cast<unsigned>(below(3)), 1 + static_cast<unsigned>(below(6)));
  if (good.empty()) return;
  glc_frames *fr = nullptr;
  REQUIRE(glc_deserialize(good.data(), good.size(), &fr) == GLC_OK);
  glc_frames_view v;
  REQUIRE(glc_frames_get_view(fr, &v) == GLC_OK);
  // own copies of every array, so that a mutation cannot touch the library's object
  std::vector<uint64_t> list_begin(v.list_begin, v.list_begin + v.n_frames + 1), list_off(v.list_off, v.list_off + v.n_lists + 1),
      scale_begin(v.scale_begin, v.scale_begin + v.n_frames + 1), raw_begin(v.raw_begin, v.raw_begin + v.n_frames + 1);
  std::vector<uint32_t> pairs(v.pairs, v.pairs + v.n_pairs);
  std::vector<float> scales(v.scales, v.scales + v.n_scales);
  std::vector<uint8_t> raw_tag(v.raw_tag, v.raw_tag + v.n_frames);
  std::vector<int16_t> raw(v.raw, v.raw + v.n_raw);
  glc_frames_view p = v;
  auto point = [&] {
    p.list_begin = list_begin.data(), p.list_off = list_off.data(), p.pairs = pairs.data(), p.scale_begin = scale_begin.data();
    p.scales = scales.data(), p.raw_tag = raw_tag.data(), p.raw_begin = raw_begin.data(), p.raw = raw.data();
  };
  point();
  {  // the unmodified parts rebuild the same stream, byte for byte, under a caller's id
    glc_frames *back = nullptr;
    REQUIRE(glc_frames_from_parts(&p, 77, &back) == GLC_OK && glc_frames_stream_id(back) == 77);
    std::vector<uint8_t> again(glc_serialized_size(back));
    uint64_t w = 0;
    REQUIRE(glc_serialize(back, again.data(), again.size(), &w) == GLC_OK && again == good);
    glc_frames_free(back);
    REQUIRE(glc_frames_from_parts(&p, 1ull << 63, &back) == GLC_EINVAL && back == nullptr);
  }
  for (int round = 0; round < 30; ++round) {
    // hostile offsets / counts / tags: the constructor validates before it reads through them
    glc_frames_view q = p;
    std::vector<uint64_t> lb = list_begin, lo = list_off, sb = scale_begin, rb = raw_begin;
    std::vector<uint8_t> rt = raw_tag;
    auto wreck = [&](std::vector<uint64_t> &a) {
      if (a.empty()) return;
      uint64_t &x = a[below(a.size())];
      switch (below(4)) {
        case 0: x += 1 + below(5); break;
        case 1: x = x ? x - 1 : 7; break;
        case 2: x = rnd() >> below(64); break;
        default: x = ~0ull - below(9); break;
      }
    };
    switch (below(8)) {
      case 0: wreck(lb); break;
      case 1: wreck(lo); break;
      case 2: wreck(sb); break;
      case 3: wreck(rb); break;
      case 4: if (!rt.empty()) rt[below(rt.size())] = static_cast<uint8_t>(below(4)); break;
      case 5: q.n_pairs += below(2) ? 1 : ~0ull >> below(40); break;
      case 6: q.n_lists = below(2) ? q.n_lists + 1 : q.n_lists ? q.n_lists - 1 : 3; break;
      default: q.n_raw = below(2) ? q.n_raw + 2 : q.n_raw / 2; break;
    }
    // (counts larger than the arrays are only legal to TRY when the offsets veto them first: n_lists / n_pairs /
    //  n_raw are checked against the closing offsets, which live inside the arrays we own)
    if (q.n_lists > p.n_lists) q.n_lists = p.n_lists;   // list_off has n_lists + 1 entries: reading beyond is the caller's bug, not input
    q.list_begin = lb.data(), q.list_off = lo.data(), q.scale_begin = sb.data(), q.raw_begin = rb.data(), q.raw_tag = rt.data();
    glc_frames *out = nullptr;
    const int rc = glc_frames_from_parts(&q, 0, &out);
    if (rc == GLC_OK) {  // whatever is accepted is a well-formed stream: it serialises, re-parses and can be walked
      std::vector<uint8_t> bytes(glc_serialized_size(out));
      uint64_t w = 0;
      REQUIRE(glc_serialize(out, bytes.data(), bytes.size(), &w) == GLC_OK);
      glc_frames *re = nullptr;
      REQUIRE(glc_deserialize(bytes.data(), bytes.size(), &re) == GLC_OK);
      touch(re);
      glc_frames_free(re);
      touch(out);
      glc_frames_free(out);
    } else {
      REQUIRE(out == nullptr && (rc == GLC_EFORMAT || rc == GLC_EINVAL || rc == GLC_ENOMEM));
    }
  }
  {  // the gather form from per-vector copies: same bytes; null vectors with non-zero lengths are refused
    std::vector<std::vector<uint32_t>> lists;
    std::vector<std::vector<float>> sc(v.n_frames);
    std::vector<std::vector<int16_t>> rw(v.n_frames);
    std::vector<uint32_t> lists_per(v.n_frames), list_len, scales_per(v.n_frames);
    std::vector<const void *> list_ptr;
    std::vector<const float *> scale_ptr(v.n_frames);
    std::vector<const int16_t *> raw_ptr(v.n_frames);
    std::vector<uint64_t> raw_len(v.n_frames);
    for (uint64_t f = 0; f < v.n_frames; ++f) {
      lists_per[f] = static_cast<uint32_t>(list_begin[f + 1] - list_begin[f]);
      for (uint64_t l = list_begin[f]; l < list_begin[f + 1]; ++l) lists.emplace_back(pairs.begin() + list_off[l], pairs.begin() + list_off[l + 1]);
      sc[f].assign(scales.begin() + scale_begin[f], scales.begin() + scale_begin[f + 1]);
      scales_per[f] = static_cast<uint32_t>(sc[f].size());
      if (raw_tag[f]) rw[f].assign(raw.begin() + raw_begin[f], raw.begin() + raw_begin[f + 1]);
    }
    static const int16_t none[1] = {0};
    for (auto &l : lists) list_ptr.push_back(l.data()), list_len.push_back(static_cast<uint32_t>(l.size()));
    for (uint64_t f = 0; f < v.n_frames; ++f) {
      scale_ptr[f] = sc[f].data();
      raw_ptr[f] = raw_tag[f] ? (rw[f].empty() ? none : rw[f].data()) : nullptr;
      raw_len[f] = rw[f].size();
    }
    glc_frames_gather g{};
    g.sample_rate = v.sample_rate, g.channels = v.channels, g.total_samples = v.total_samples, g.encoder_delay = v.encoder_delay;
    g.padding = v.padding, g.original_length = v.original_length, g.n_frames = v.n_frames;
    g.lists_per_frame = lists_per.data(), g.list_ptr = list_ptr.data(), g.list_len = list_len.data();
    g.scales_per_frame = scales_per.data(), g.scale_ptr = scale_ptr.data(), g.raw_ptr = raw_ptr.data(), g.raw_len = raw_len.data();
    glc_frames *back = nullptr;
    REQUIRE(glc_frames_from_gather(&g, 5, &back) == GLC_OK);
    std::vector<uint8_t> again(glc_serialized_size(back));
    uint64_t w = 0;
    REQUIRE(glc_serialize(back, again.data(), again.size(), &w) == GLC_OK && again == good);
    glc_frames_free(back);
    if (!list_ptr.empty() && list_len[0]) {
      list_ptr[0] = nullptr;
      REQUIRE(glc_frames_from_gather(&g, 0, &back) == GLC_EINVAL && back == nullptr);
    }
  }
  glc_frames_free(fr);
}

// ---- WAV ----------------------------------------------------------------------------------------
static void fuzz_wav(const std::string &tmp) {
  const std::string path = tmp + "/fuzz.wav";
  const unsigned ch = 1 + static_cast<unsigned>(below(3));
  std::vector<float> pcm(ch * (1 + below(300)));
  for (float &v : pcm) v = static_cast<float>(static_cast<int64_t>(below(70000)) - 35000) / 32767.0f;
  REQUIRE(glc_wav_save16(path.c_str(), pcm.data(), pcm.size(), 8000 + static_cast<uint32_t>(below(90000)), static_cast<uint16_t>(ch)) == GLC_OK);
  std::vector<uint8_t> good;
  {
    FILE *fp = std::fopen(path.c_str(), "rb");
    REQUIRE(fp);
    uint8_t tmpb[4096];
    size_t got;
    while ((got = std::fread(tmpb, 1, sizeof tmpb, fp)) > 0) good.insert(good.end(), tmpb, tmpb + got);
    std::fclose(fp);
  }
  float *p = nullptr;
  uint64_t n = 0;
  uint32_t sr = 0;
  uint16_t c = 0;
  REQUIRE(glc_wav_load(path.c_str(), &p, &n, &sr, &c) == GLC_OK && n == pcm.size() && c == ch);
  glc_free(p);
  for (int round = 0; round < 30; ++round) {
    std::vector<uint8_t> bad = good;
    for (uint64_t m = 1 + below(3); m; --m) {
      if (below(2) && bad.size() > 44) bad[below(44)] = static_cast<uint8_t>(rnd());  // aim at the header
      else mutate(bad);
    }
    FILE *fp = std::fopen(path.c_str(), "wb");
    REQUIRE(fp);
    std::fwrite(bad.data(), 1, bad.size(), fp);
    std::fclose(fp);
    p = nullptr;
    if (glc_wav_load(path.c_str(), &p, &n, &sr, &c) == GLC_OK) {
      REQUIRE(p != nullptr && c != 0);
      volatile float sink = n ? p[n - 1] : 0.0f;  // the whole buffer must be addressable
      (void)sink;
      glc_free(p);
    }
  }
}

// ---- FLAC ---------------------------------------------------------------------------------------
static void fuzz_flac() {
  const unsigned ch = 1 + static_cast<unsigned>(below(8));
  const uint64_t n = ch * (16 + below(9000)) + below(ch);
  std::vector<float> pcm(n);
  const int kind = static_cast<int>(below(4));
  float walk = 0.0f;
  for (float &v : pcm) {
    const float r = static_cast<float>(static_cast<int64_t>(below(2001)) - 1000) / 1000.0f;
    if (kind == 0) v = r * 1.5f;                          // clips
    else if (kind == 1) v = (walk += r * 0.01f);          // smooth
    else if (kind == 2) v = below(500) ? 0.0f : r * 4.0f; // impulses: very long Rice zero runs
    else v = r * 1e-3f;
  }
  const uint8_t level = static_cast<uint8_t>(below(9));
  uint8_t *buf = nullptr;
  uint64_t len = 0;
  REQUIRE(glc_flac_encode(pcm.data(), n, 8000 + static_cast<uint32_t>(below(190000)), static_cast<uint16_t>(ch), level, &buf, &len) == GLC_OK);
  const uint64_t total = n / ch;
  const uint64_t block = level <= 2 ? 1152 : 4096;
  const unsigned order = level == 0 ? 0 : level == 1 ? 1 : level == 2 ? 2 : level <= 4 ? 3 : 4;
  const uint64_t last = total > block ? (total % block ? total % block : block) : total;
  float *out = nullptr;
  uint64_t n_out = 0;
  uint32_t sr = 0;
  uint16_t c = 0;
  const int rc = glc_flac_decode(buf, len, &out, &n_out, &sr, &c);
  if (order && last == order) {
    REQUIRE(rc != GLC_OK);  // the reference's unparseable final frame, reproduced on purpose
  } else {
    REQUIRE(rc == GLC_OK && n_out == total * ch && c == ch);
    for (uint64_t i = 0; i < n_out; ++i) {
      float v = pcm[i] * 32767.0f;
      v = v != v ? 0.0f : v < -32768.0f ? -32768.0f : v > 32767.0f ? 32767.0f : v;
      REQUIRE(out[i] == static_cast<float>(static_cast<int16_t>(v)) / 32768.0f);
    }
    glc_free(out);
  }
  std::vector<uint8_t> good(buf, buf + len);
  glc_free(buf);
  for (int round = 0; round < 30; ++round) {
    std::vector<uint8_t> bad = good;
    for (uint64_t m = 1 + below(3); m; --m) mutate(bad);
    out = nullptr;
    if (glc_flac_decode(bad.data(), bad.size(), &out, &n_out, &sr, &c) == GLC_OK) {
      volatile float sink = n_out ? out[n_out - 1] : 0.0f;
      (void)sink;
      glc_free(out);
    }
  }
  {  // pure noise behind a valid marker
    std::vector<uint8_t> junk(4 + below(400));
    for (uint8_t &b : junk) b = static_cast<uint8_t>(rnd());
    if (below(2) && junk.size() >= 4) std::memcpy(junk.data(), "fLaC", 4);
    out = nullptr;
    if (glc_flac_decode(junk.data(), junk.size(), &out, &n_out, &sr, &c) == GLC_OK) glc_free(out);
  }
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 5.0;
  const std::string tmp = argc > 2 ? argv[2] : "/tmp";
  if (argc > 3) rng_state ^= std::strtoull(argv[3], nullptr, 0);
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t rounds = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    fuzz_glc(tmp);
    fuzz_bridge();
    fuzz_wav(tmp);
    fuzz_flac();
    ++rounds;
  }
  std::printf("host_fuzz: %llu rounds clean\n", static_cast<unsigned long long>(rounds));
  return 0;
}
