// glc_flac.cpp — FLAC file-I/O twin of the reference (host only, integer work, no device):
//   * the encoder restates the reference's own pure-Rust encoder, src/flac.rs:8-1087 (fixed
//     predictors + partitioned Rice, 16-bit, independent channels), so that `glc -d file.glc`
//     writes the same .flac bytes;
//   * the decoder stands in for the third-party `claxon` crate the reference reads .flac input
//     with (src/audio.rs:68-85); it follows RFC 9639 (all subframe types, stereo decorrelation,
//     both Rice code books, wasted bits) and checks both CRCs.
// Frames of a FLAC stream are byte-aligned and independent, so the encoder fans them out over
// host threads and concatenates; the STREAMINFO MD5 is a serial chain and runs beside them.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "glc_common.h"

namespace {

// ---------------------------------------------------------------- bit output (BitWriter, flac.rs:321-424)
struct BitOut {
  std::vector<uint8_t> bytes;  // storage; [0, len) is written
  size_t len = 0;
  uint64_t acc = 0;  // pending bits, right-aligned (bits above n are stale)
  unsigned n = 0;    // number of pending bits (< 32 between calls)

  // make room for `extra` more bytes beyond what is pending
  void room(size_t extra) {
    if (len + extra + 16 > bytes.size()) bytes.resize(std::max(bytes.size() * 2, len + extra + 16));
  }
  // low `bits` bits of value, most significant first (write_bits, flac.rs:340-380)
  void put(uint64_t value, unsigned bits) {
    room(16);
    while (bits > 32) {
      put32(static_cast<uint32_t>(value >> (bits - 32)), 32);
      bits -= 32;
    }
    put32(static_cast<uint32_t>(value), bits);
  }
  // the caller has made room (8 bytes per call suffice)
  void put32(uint32_t value, unsigned bits) {
    if (!bits) return;
    const uint64_t v = bits == 32 ? value : (value & ((1u << bits) - 1u));
    acc = (acc << bits) | v;
    n += bits;
    if (n >= 32) {
      const uint32_t w = static_cast<uint32_t>(acc >> (n - 32));
      uint8_t *d = bytes.data() + len;
      d[0] = static_cast<uint8_t>(w >> 24), d[1] = static_cast<uint8_t>(w >> 16);
      d[2] = static_cast<uint8_t>(w >> 8), d[3] = static_cast<uint8_t>(w);
      len += 4;
      n -= 32;
    }
  }
  // `count` zero bits (the run of write_unary, flac.rs:395-403); room for count/8 + 8 bytes made here
  void zeros(uint64_t count) {
    if (count >= 64) {
      room(static_cast<size_t>(count >> 3) + 16);
      const unsigned fill = (32 - n) & 31;  // complete the pending word
      put32(0, fill);
      count -= fill;
      std::memset(bytes.data() + len, 0, static_cast<size_t>(count >> 3));
      len += static_cast<size_t>(count >> 3);
      count &= 7;
    }
    while (count > 32) put32(0, 32), count -= 32;
    put32(0, static_cast<unsigned>(count));
  }
  // pending bits out, zero-padded to a byte boundary (byte_align, flac.rs:405-413)
  void align() {
    room(8);
    if (n & 7) put32(0, 8 - (n & 7));
    while (n) {
      bytes[len++] = static_cast<uint8_t>(acc >> (n - 8));
      n -= 8;
    }
  }
};

// ---------------------------------------------------------------- checksums
uint8_t crc8(const uint8_t *p, size_t len) {  // polynomial 0x07, flac.rs:19-51
  static uint8_t table[256];
  static bool ready = [] {
    for (int i = 0; i < 256; ++i) {
      uint8_t c = static_cast<uint8_t>(i);
      for (int b = 0; b < 8; ++b) c = static_cast<uint8_t>((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
      table[i] = c;
    }
    return true;
  }();
  (void)ready;
  uint8_t c = 0;
  for (size_t i = 0; i < len; ++i) c = table[c ^ p[i]];
  return c;
}

uint16_t crc16(const uint8_t *p, size_t len) {  // polynomial 0x8005, flac.rs:54-80
  static uint16_t table[256];
  static bool ready = [] {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = static_cast<uint16_t>(i << 8);
      for (int b = 0; b < 8; ++b) c = static_cast<uint16_t>((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
      table[i] = c;
    }
    return true;
  }();
  (void)ready;
  uint16_t c = 0;
  for (size_t i = 0; i < len; ++i) c = static_cast<uint16_t>((c << 8) ^ table[(c >> 8) ^ p[i]]);
  return c;
}

// RFC 1321 over the little-endian bytes of the interleaved i16 stream (compute_md5,
// flac.rs:305-318; the reference's MD5Context :83-302 is the standard algorithm)
struct Md5 {
  uint32_t s[4] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u};
  uint64_t bits = 0;
  uint8_t buf[64];
  unsigned fill = 0;

  static uint32_t rol(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
  void block(const uint8_t *p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
        0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
        0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
        0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
        0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
        0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
        0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int R[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t x[16];
    for (int i = 0; i < 16; ++i)
      x[i] = p[4 * i] | (p[4 * i + 1] << 8) | (p[4 * i + 2] << 16) | (static_cast<uint32_t>(p[4 * i + 3]) << 24);
    uint32_t a = s[0], b = s[1], c = s[2], d = s[3];
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) f = (b & c) | (~b & d), g = i;
      else if (i < 32) f = (b & d) | (c & ~d), g = (5 * i + 1) & 15;
      else if (i < 48) f = b ^ c ^ d, g = (3 * i + 5) & 15;
      else f = c ^ (b | ~d), g = (7 * i) & 15;
      const uint32_t t = d;
      d = c;
      c = b;
      b = b + rol(a + f + K[i] + x[g], R[i]);
      a = t;
    }
    s[0] += a, s[1] += b, s[2] += c, s[3] += d;
  }
  void update(const uint8_t *p, size_t len) {
    bits += static_cast<uint64_t>(len) << 3;
    if (fill) {
      const size_t take = std::min<size_t>(64 - fill, len);
      std::memcpy(buf + fill, p, take);
      fill += static_cast<unsigned>(take), p += take, len -= take;
      if (fill < 64) return;
      block(buf);
      fill = 0;
    }
    for (; len >= 64; p += 64, len -= 64) block(p);
    if (len) std::memcpy(buf, p, len), fill = static_cast<unsigned>(len);
  }
  void finish(uint8_t out[16]) {
    const uint64_t total = bits;
    static const uint8_t pad[64] = {0x80};
    update(pad, fill < 56 ? 56 - fill : 120 - fill);
    uint8_t lenb[8];
    for (int i = 0; i < 8; ++i) lenb[i] = static_cast<uint8_t>(total >> (8 * i));
    update(lenb, 8);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) out[4 * i + j] = static_cast<uint8_t>(s[i] >> (8 * j));
  }
};

// ---------------------------------------------------------------- encoder (flac.rs:427-1053)
void put_utf8(BitOut &w, uint64_t v) {  // write_utf8_number, flac.rs:427-478
  if (v < 0x80) {
    w.put(v, 8);
    return;
  }
  int extra;
  uint8_t lead;
  if (v < 0x800) extra = 1, lead = 0xC0 | ((v >> 6) & 0x1F);
  else if (v < 0x10000) extra = 2, lead = 0xE0 | ((v >> 12) & 0x0F);
  else if (v < 0x200000) extra = 3, lead = 0xF0 | ((v >> 18) & 0x07);
  else if (v < 0x4000000) extra = 4, lead = 0xF8 | ((v >> 24) & 0x03);
  else if (v < 0x80000000ull) extra = 5, lead = 0xFC | ((v >> 30) & 0x01);
  else extra = 6, lead = 0xFE;
  w.put(lead, 8);
  for (int i = extra - 1; i >= 0; --i) w.put(0x80 | ((v >> (6 * i)) & 0x3F), 8);
}

// calculate_rice_parameter, flac.rs:515-552: floor(log2(mean |r|)), capped at 14 (the later
// "adjust" branch can never fire: mean >= 2^param by construction)
uint32_t rice_parameter(const int32_t *r, size_t len) {
  if (!len) return 0;
  uint64_t sum = 0;
  for (size_t i = 0; i < len; ++i) sum += static_cast<uint32_t>(r[i] < 0 ? -static_cast<int64_t>(r[i]) : r[i]);
  uint64_t mean = sum / len;
  uint32_t param = 0;
  while (mean > 1 && param < 14) mean >>= 1, ++param;
  return param;
}

unsigned trailing_zeros(size_t v) {
  unsigned n = 0;
  while (v && !(v & 1)) v >>= 1, ++n;
  return v ? n : 64;
}

// encode_subframe + apply_fixed_predictor + encode_residual + encode_rice_partition
// (flac.rs:481-512, :555-745) for one channel of one block
void put_subframe(BitOut &w, const int32_t *s, size_t block, unsigned level, std::vector<int32_t> &res) {
  static const size_t kOrder[9] = {0, 1, 2, 3, 3, 4, 4, 4, 4};
  size_t order = kOrder[level];
  if (block < order) order = 0;
  w.put(0, 1);
  w.put(order ? (0x08u | order) : 0x01u, 6);  // fixed predictor of that order, or verbatim
  w.put(0, 1);                                // no wasted bits
  if (!order) {
    w.room(2 * block + 32);
    for (size_t i = 0; i < block; ++i) w.put32(static_cast<uint32_t>(s[i]), 16);
    return;
  }
  for (size_t i = 0; i < order; ++i) w.put(static_cast<uint32_t>(s[i]), 16);
  const size_t nres = block - order;
  res.resize(nres);
  for (size_t i = order; i < block; ++i) {
    int32_t pred;
    switch (order) {
      case 1: pred = s[i - 1]; break;
      case 2: pred = 2 * s[i - 1] - s[i - 2]; break;
      case 3: pred = 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
      default: pred = 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
    }
    res[i - order] = s[i] - pred;
  }
  // partition order by level, limited by the block's power-of-two factor (flac.rs:590-608)
  const unsigned cap = level == 0 ? 0u : level <= 2 ? 2u : level <= 5 ? 4u : 6u;
  unsigned porder = std::min(cap, std::min(trailing_zeros(block), 8u));
  while (porder > 0) {
    const size_t per = block >> porder;
    if (per > order && per >= 4) break;
    --porder;
  }
  w.put(0, 2);  // Rice code book with 4-bit parameters
  w.put(porder, 4);
  const size_t per = block >> porder;
  size_t at = 0;
  for (size_t p = 0; p < (size_t{1} << porder); ++p) {
    const size_t cnt = p == 0 ? per - order : per;
    if (!cnt) continue;  // block == order: nothing is written, not even a parameter (flac.rs:632-635)
    const int32_t *r = res.data() + at;
    at += cnt;
    const uint32_t k = rice_parameter(r, cnt);
    w.put(k, 4);
    w.room(cnt * 12 + 32);  // <= 12 bytes per sample while its zero run stays below 64 bits
    for (size_t i = 0; i < cnt; ++i) {
      const uint32_t folded = r[i] >= 0 ? static_cast<uint32_t>(r[i]) << 1
                                        : (static_cast<uint32_t>(-(r[i] + 1)) << 1) | 1u;
      const uint32_t msb = folded >> k;
      w.zeros(msb);
      if (msb >= 64) w.room((cnt - i) * 12 + 32);  // a long run made only its own room
      w.put32((1u << k) | (folded & ((1u << k) - 1u)), k + 1);  // the stop bit, then k low bits
    }
  }
}

// encode_frame, flac.rs:748-905
void put_frame(BitOut &w, const int16_t *pcm, size_t block, unsigned ch, uint32_t sample_rate, uint32_t frame_no,
               unsigned level, std::vector<int32_t> &plane, std::vector<int32_t> &res) {
  w.align();  // frames start on a byte boundary; this only flushes the previous frame's CRC
  const size_t start = w.len;
  w.put(0x3FFE, 14);
  w.put(0, 2);  // reserved, fixed block size
  unsigned bcode;
  switch (block) {
    case 192: bcode = 1; break;
    case 576: bcode = 2; break;
    case 1152: bcode = 3; break;
    case 2304: bcode = 4; break;
    case 4608: bcode = 5; break;
    case 256: bcode = 8; break;
    case 512: bcode = 9; break;
    case 1024: bcode = 10; break;
    case 2048: bcode = 11; break;
    case 4096: bcode = 12; break;
    case 8192: bcode = 13; break;
    case 16384: bcode = 14; break;
    case 32768: bcode = 15; break;
    default: bcode = block < 256 ? 6 : 7;
  }
  w.put(bcode, 4);
  unsigned rcode;
  switch (sample_rate) {
    case 88200: rcode = 1; break;
    case 176400: rcode = 2; break;
    case 192000: rcode = 3; break;
    case 8000: rcode = 4; break;
    case 16000: rcode = 5; break;
    case 22050: rcode = 6; break;
    case 24000: rcode = 7; break;
    case 32000: rcode = 8; break;
    case 44100: rcode = 9; break;
    case 48000: rcode = 10; break;
    case 96000: rcode = 11; break;
    default: rcode = 0;  // "get from STREAMINFO"
  }
  w.put(rcode, 4);
  w.put(ch == 1 ? 0u : ch == 2 ? 1u : ch - 1u, 4);  // independent channels (flac.rs:821-833)
  w.put(4, 3);                                     // 16 bits per sample
  w.put(0, 1);
  put_utf8(w, frame_no);
  if (bcode == 6) w.put(block - 1, 8);
  else if (bcode == 7) w.put(block - 1, 16);
  w.align();  // the header is a whole number of bytes: flush them so that the CRC can see them
  w.put(crc8(w.bytes.data() + start, w.len - start), 8);
  plane.resize(block);
  for (unsigned c = 0; c < ch; ++c) {
    for (size_t i = 0; i < block; ++i) plane[i] = pcm[i * ch + c];
    put_subframe(w, plane.data(), block, level, res);
  }
  w.align();
  w.put(crc16(w.bytes.data() + start, w.len - start), 16);
}

int16_t to_i16(float s) {  // (s * 32767.0).clamp(-32768.0, 32767.0) as i16, flac.rs:955-958
  float v = s * 32767.0f;
  if (v != v) return 0;  // NaN passes through clamp and casts to 0
  if (v < -32768.0f) v = -32768.0f;
  if (v > 32767.0f) v = 32767.0f;
  return static_cast<int16_t>(v);
}

// -> the stream as consecutive pieces (STREAMINFO, then one run of frames per worker)
int flac_encode(const float *samples, uint64_t n, uint32_t sample_rate, uint16_t channels, unsigned level,
                std::vector<BitOut> &parts) {
  if (channels == 0) {
    glc::set_global_error("glc_flac_encode: channels == 0 (the reference divides by zero here)");
    return GLC_EINVAL;
  }
  const uint64_t total = n / channels;
  if (total < 16) {
    glc::set_global_error("FLAC requires at least 16 samples per channel, got " + std::to_string(total));
    return GLC_EINVAL;
  }
  if (level > 8) {
    glc::set_global_error("Invalid compression level " + std::to_string(level) + ", must be 0-8");
    return GLC_EINVAL;
  }
  const size_t block = static_cast<size_t>(std::max<uint64_t>(std::min<uint64_t>(level <= 2 ? 1152 : 4096, total), 16));
  const uint64_t n_frames = (total + block - 1) / block;
  std::vector<int16_t> pcm(n);

  unsigned hw = std::thread::hardware_concurrency();
  const unsigned n_thr = static_cast<unsigned>(std::max<uint64_t>(1, std::min<uint64_t>({hw ? hw : 1u, 32u, n_frames / 8 + 1})));
  // f32 -> i16 over all n samples (a trailing partial sample-frame is hashed but not framed,
  // flac.rs:960, :1021-1030), fanned out like the frames below
  auto convert = [&](uint64_t a, uint64_t b) {
    for (uint64_t i = a; i < b; ++i) pcm[i] = to_i16(samples[i]);
  };
  parts.assign(n_thr, BitOut());
  auto frames = [&](unsigned t) {
    const uint64_t f0 = n_frames * t / n_thr, f1 = n_frames * (t + 1) / n_thr;
    std::vector<int32_t> plane, res;
    BitOut &w = parts[t];
    w.bytes.resize(static_cast<size_t>((f1 - f0) * block * channels * 3 / 2) + 4096);
    for (uint64_t f = f0; f < f1; ++f) {
      const uint64_t first = f * block;
      const size_t cur = static_cast<size_t>(std::min<uint64_t>(block, total - first));
      put_frame(w, pcm.data() + first * channels, cur, channels, sample_rate, static_cast<uint32_t>(f), level, plane, res);
    }
    w.align();
  };
  {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_thr; ++t) pool.emplace_back(convert, n * t / n_thr, n * (t + 1) / n_thr);
    convert(0, n / n_thr);
    for (auto &th : pool) th.join();
  }
  uint8_t md5[16];
  {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_thr; ++t) pool.emplace_back(frames, t);
    Md5 h;  // serial by nature; runs on the calling thread beside the frame workers
    h.update(reinterpret_cast<const uint8_t *>(pcm.data()), static_cast<size_t>(n) * 2);
    h.finish(md5);
    for (auto &th : pool) th.join();
  }
  BitOut head;
  head.put(0x664C6143u, 32);  // "fLaC"
  // write_streaminfo, flac.rs:908-944
  head.put(1, 1);
  head.put(0, 7);
  head.put(34, 24);
  head.put(static_cast<uint16_t>(block), 16);
  head.put(static_cast<uint16_t>(block), 16);
  head.put(0, 24);
  head.put(0, 24);
  head.put(sample_rate, 20);
  head.put(static_cast<uint16_t>(channels - 1), 3);
  head.put(15, 5);
  head.put(total, 36);
  for (uint8_t b : md5) head.put(b, 8);
  head.align();
  parts.insert(parts.begin(), std::move(head));
  return GLC_OK;
}

// ---------------------------------------------------------------- decoder (≙ claxon, RFC 9639)
struct BitIn {
  const uint8_t *p;
  size_t len, pos = 0;  // pos in bits
  bool bad = false;
  // the next 57..64 bits, left-aligned; bytes past the end read as zero
  uint64_t window() const {
    const size_t at = pos >> 3;
    uint64_t w = 0;
    if (at + 8 <= len) {
      for (int i = 0; i < 8; ++i) w = (w << 8) | p[at + i];
    } else {
      for (int i = 0; i < 8; ++i) w = (w << 8) | (at + i < len ? p[at + i] : 0u);
    }
    return w << (pos & 7);
  }
  void advance(size_t bits) {
    pos += bits;
    if (pos > len * 8) bad = true, pos = len * 8;
  }
  uint64_t get(unsigned bits) {
    if (!bits) return 0;
    if (bits > 32) {
      const uint64_t hi = get(bits - 32);
      return (hi << 32) | get(32);
    }
    const uint64_t v = window() >> (64 - bits);
    advance(bits);
    return v;
  }
  int64_t gets(unsigned bits) {
    if (!bits) return 0;
    const uint64_t v = get(bits);
    const uint64_t sign = 1ull << (bits - 1);
    return static_cast<int64_t>((v ^ sign)) - static_cast<int64_t>(sign);
  }
  uint64_t unary() {  // number of zeros before the next one
    uint64_t z = 0;
    for (;;) {
      const uint64_t w = window();
      if (w) {
        const unsigned lead = static_cast<unsigned>(__builtin_clzll(w));
        advance(lead + 1);
        return z + lead;
      }
      const unsigned valid = 64 - static_cast<unsigned>(pos & 7);
      z += valid;
      advance(valid);
      if (bad) return 0;
    }
  }
};

// Two's-complement wrap-around arithmetic: a damaged stream can make the predictors run away, and
// that must stay defined behaviour (the frame's CRC-16 rejects it afterwards).
inline int64_t wadd(int64_t a, int64_t b) { return static_cast<int64_t>(static_cast<uint64_t>(a) + static_cast<uint64_t>(b)); }
inline int64_t wsub(int64_t a, int64_t b) { return static_cast<int64_t>(static_cast<uint64_t>(a) - static_cast<uint64_t>(b)); }
inline int64_t wmul(int64_t a, int64_t b) { return static_cast<int64_t>(static_cast<uint64_t>(a) * static_cast<uint64_t>(b)); }

const char *read_residual(BitIn &r, int64_t *s, size_t block, size_t order) {
  const unsigned method = static_cast<unsigned>(r.get(2));
  if (method > 1) return "reserved residual coding method";
  const unsigned pbits = method ? 5 : 4, esc = method ? 31 : 15;
  const unsigned porder = static_cast<unsigned>(r.get(4));
  if ((block >> porder) << porder != block) return "block size not divisible by the partition count";
  const size_t per = block >> porder;
  if (per < order) return "partition shorter than the predictor order";
  size_t i = order;
  for (size_t p = 0; p < (size_t{1} << porder); ++p) {
    const size_t cnt = p == 0 ? per - order : per;
    const unsigned k = static_cast<unsigned>(r.get(pbits));
    if (k == esc) {
      const unsigned raw = static_cast<unsigned>(r.get(5));
      for (size_t j = 0; j < cnt; ++j) s[i++] = r.gets(raw);
    } else {
      for (size_t j = 0; j < cnt; ++j) {
        const uint64_t folded = (r.unary() << k) | r.get(k);
        s[i++] = static_cast<int64_t>(folded >> 1) ^ -static_cast<int64_t>(folded & 1);
      }
    }
    if (r.bad) return "truncated residual";
  }
  return nullptr;
}

const char *read_subframe(BitIn &r, int64_t *s, size_t block, unsigned bps) {
  if (r.get(1)) return "subframe padding bit set";
  const unsigned type = static_cast<unsigned>(r.get(6));
  unsigned wasted = 0;
  if (r.get(1)) {
    wasted = static_cast<unsigned>(r.unary()) + 1;
    if (wasted >= bps) return "wasted bits exceed the sample size";
    bps -= wasted;
  }
  if (type == 0) {
    const int64_t v = r.gets(bps);
    for (size_t i = 0; i < block; ++i) s[i] = v;
  } else if (type == 1) {
    for (size_t i = 0; i < block; ++i) s[i] = r.gets(bps);
  } else if (type >= 8 && type <= 12) {
    const size_t order = type - 8;
    if (order > block) return "fixed predictor order exceeds the block size";
    for (size_t i = 0; i < order; ++i) s[i] = r.gets(bps);
    if (const char *e = read_residual(r, s, block, order)) return e;
    for (size_t i = order; i < block; ++i) {
      int64_t pred = 0;
      switch (order) {
        case 1: pred = s[i - 1]; break;
        case 2: pred = wsub(wmul(2, s[i - 1]), s[i - 2]); break;
        case 3: pred = wadd(wsub(wmul(3, s[i - 1]), wmul(3, s[i - 2])), s[i - 3]); break;
        case 4: pred = wsub(wadd(wsub(wmul(4, s[i - 1]), wmul(6, s[i - 2])), wmul(4, s[i - 3])), s[i - 4]); break;
        default: break;
      }
      s[i] = wadd(s[i], pred);
    }
  } else if (type >= 32) {
    const size_t order = (type & 31) + 1;
    if (order > block) return "LPC order exceeds the block size";
    for (size_t i = 0; i < order; ++i) s[i] = r.gets(bps);
    const unsigned prec = static_cast<unsigned>(r.get(4)) + 1;
    if (prec == 16) return "reserved LPC precision";
    const int64_t shift = r.gets(5);
    if (shift < 0) return "negative LPC shift";
    int64_t coef[32];
    for (size_t j = 0; j < order; ++j) coef[j] = r.gets(prec);
    if (const char *e = read_residual(r, s, block, order)) return e;
    for (size_t i = order; i < block; ++i) {
      int64_t acc = 0;
      for (size_t j = 0; j < order; ++j) acc = wadd(acc, wmul(coef[j], s[i - 1 - j]));
      s[i] = wadd(s[i], acc >> shift);
    }
  } else {
    return "reserved subframe type";
  }
  if (r.bad) return "truncated subframe";
  if (wasted)
    for (size_t i = 0; i < block; ++i) s[i] = static_cast<int64_t>(static_cast<uint64_t>(s[i]) << wasted);
  return nullptr;
}

struct Span {
  const uint8_t *p;
  size_t n;
  size_t size() const { return n; }
  const uint8_t *data() const { return p; }
  uint8_t operator[](size_t i) const { return p[i]; }
};

int flac_decode(const Span f, std::vector<float> &out, uint32_t &sample_rate, uint16_t &channels) {
  auto fail = [](const std::string &m) {
    glc::set_global_error("glc_flac_load: " + m);
    return GLC_EFORMAT;
  };
  if (f.size() < 8 || std::memcmp(f.data(), "fLaC", 4)) return fail("not a FLAC stream");
  size_t pos = 4;
  bool have_info = false, last = false;
  unsigned info_bps = 0, info_ch = 0;
  uint32_t info_rate = 0;
  uint64_t info_total = 0;
  while (!last) {
    if (pos + 4 > f.size()) return fail("truncated metadata");
    last = f[pos] & 0x80;
    const unsigned type = f[pos] & 0x7F;
    const size_t len = (f[pos + 1] << 16) | (f[pos + 2] << 8) | f[pos + 3];
    pos += 4;
    if (pos + len > f.size()) return fail("truncated metadata block");
    if (type == 0) {
      if (len < 34) return fail("short STREAMINFO");
      BitIn r{f.data() + pos, len};
      r.get(16), r.get(16), r.get(24), r.get(24);
      info_rate = static_cast<uint32_t>(r.get(20));
      info_ch = static_cast<unsigned>(r.get(3)) + 1;
      info_bps = static_cast<unsigned>(r.get(5)) + 1;
      info_total = r.get(36);
      have_info = true;
    }
    pos += len;
  }
  if (!have_info) return fail("no STREAMINFO block");
  if (info_bps < 4) return fail("unsupported sample size");
  sample_rate = info_rate;
  channels = static_cast<uint16_t>(info_ch);
  // `(1 << (bits_per_sample - 1)) as f32`, audio.rs:72
  // `(1 << (info.bits_per_sample - 1)) as f32`, audio.rs:72, on an i32 literal: 32-bit streams divide by
  // i32::MIN = -2147483648.0 (polarity inverted, quirk Q11, kept)
  const float scale = info_bps == 32 ? -2147483648.0f : static_cast<float>(1ull << (info_bps - 1));
  out.clear();
  // STREAMINFO's sample count is a hint from the file, not a promise: reserve no more than the
  // remaining bytes could plausibly hold (the vector still grows if constant frames beat that)
  if (info_total) out.reserve(static_cast<size_t>(std::min<uint64_t>(info_total * info_ch, (f.size() - pos) * 8ull)));
  std::vector<int64_t> plane;
  while (pos < f.size()) {
    BitIn r{f.data() + pos, f.size() - pos};
    if (r.get(14) != 0x3FFE) return fail("lost frame sync");
    if (r.get(1)) return fail("reserved header bit set");
    r.get(1);  // blocking strategy: only changes the meaning of the coded number
    const unsigned bcode = static_cast<unsigned>(r.get(4)), rcode = static_cast<unsigned>(r.get(4));
    const unsigned ccode = static_cast<unsigned>(r.get(4)), scode = static_cast<unsigned>(r.get(3));
    if (r.get(1)) return fail("reserved header bit set");
    unsigned lead = static_cast<unsigned>(r.get(8)), extra = 0;
    if (lead & 0x80) {
      while (lead & (0x80 >> (extra + 1))) ++extra;
      ++extra;
      if (extra == 1 || extra > 7) return fail("bad coded frame number");
      for (unsigned i = 1; i < extra; ++i)
        if ((r.get(8) & 0xC0) != 0x80) return fail("bad coded frame number");
    }
    size_t block;
    if (bcode == 0) return fail("reserved block size code");
    else if (bcode == 1) block = 192;
    else if (bcode <= 5) block = size_t{576} << (bcode - 2);
    else if (bcode == 6) block = static_cast<size_t>(r.get(8)) + 1;
    else if (bcode == 7) block = static_cast<size_t>(r.get(16)) + 1;
    else block = size_t{256} << (bcode - 8);
    if (rcode == 12) r.get(8);
    else if (rcode == 13 || rcode == 14) r.get(16);
    else if (rcode == 15) return fail("invalid sample rate code");
    static const unsigned kBps[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    unsigned bps = scode == 0 ? info_bps : kBps[scode];
    if (!bps) return fail("reserved sample size code");
    if (bps != info_bps) return fail("sample size changes mid-stream");
    unsigned nch;
    if (ccode < 8) nch = ccode + 1;
    else if (ccode <= 10) nch = 2;
    else return fail("reserved channel assignment");
    if (nch != info_ch) return fail("channel count changes mid-stream");
    if (r.bad) return fail("truncated frame header");
    const size_t hdr = r.pos >> 3;
    if (crc8(f.data() + pos, hdr) != r.get(8) || r.bad) return fail("frame header CRC-8 mismatch");
    plane.assign(block * nch, 0);
    for (unsigned c = 0; c < nch; ++c) {
      const bool side = (ccode == 8 && c == 1) || (ccode == 9 && c == 0) || (ccode == 10 && c == 1);
      if (const char *e = read_subframe(r, plane.data() + c * block, block, bps + (side ? 1 : 0))) return fail(e);
    }
    r.pos = (r.pos + 7) & ~size_t{7};
    const size_t body = r.pos >> 3;
    if (crc16(f.data() + pos, body) != r.get(16) || r.bad) return fail("frame CRC-16 mismatch");
    int64_t *a = plane.data(), *b = plane.data() + block;
    if (ccode == 8) {
      for (size_t i = 0; i < block; ++i) b[i] = wsub(a[i], b[i]);
    } else if (ccode == 9) {
      for (size_t i = 0; i < block; ++i) a[i] = wadd(a[i], b[i]);
    } else if (ccode == 10) {
      for (size_t i = 0; i < block; ++i) {
        const int64_t mid = wmul(a[i], 2) | (b[i] & 1), sd = b[i];
        a[i] = wadd(mid, sd) >> 1;
        b[i] = wsub(mid, sd) >> 1;
      }
    }
    const size_t base = out.size();
    out.resize(base + block * nch);
    for (size_t i = 0; i < block; ++i)
      for (unsigned c = 0; c < nch; ++c)
        out[base + i * nch + c] = static_cast<float>(static_cast<int32_t>(plane[c * block + i])) / scale;  // audio.rs:79
    pos += r.pos >> 3;
  }
  return GLC_OK;
}

bool read_file(const char *path, std::vector<uint8_t> &buf) {
  FILE *fp = std::fopen(path, "rb");
  if (!fp) return false;
  uint8_t tmp[1 << 16];
  size_t got;
  while ((got = std::fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + got);
  std::fclose(fp);
  return true;
}

}  // namespace

extern "C" {

int glc_flac_encode(const float *samples, uint64_t n_samples, uint32_t sample_rate, uint16_t channels,
                    uint8_t level, uint8_t **out, uint64_t *out_len) {
  if ((!samples && n_samples) || !out || !out_len) return GLC_EINVAL;
  try {  // no C++ exception may cross the C ABI
    std::vector<BitOut> parts;
    const int rc = flac_encode(samples, n_samples, sample_rate, channels, level, parts);
    if (rc != GLC_OK) return rc;
    size_t total = 0;
    for (const BitOut &p : parts) total += p.len;
    uint8_t *buf = static_cast<uint8_t *>(std::malloc(total ? total : 1));
    if (!buf) return GLC_ENOMEM;
    size_t at = 0;
    for (const BitOut &p : parts) std::memcpy(buf + at, p.bytes.data(), p.len), at += p.len;
    *out = buf;
    *out_len = total;
    return GLC_OK;
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  } catch (const std::exception &e) {
    glc::set_global_error(std::string("glc_flac_encode: ") + e.what());
    return GLC_EINVAL;
  }
}

int glc_flac_save(const char *path, const float *samples, uint64_t n_samples, uint32_t sample_rate,
                  uint16_t channels, uint8_t level) {
  if (!path || (!samples && n_samples)) return GLC_EINVAL;
  try {
    std::vector<BitOut> parts;
    const int rc = flac_encode(samples, n_samples, sample_rate, channels, level, parts);
    if (rc != GLC_OK) return rc;
    FILE *fp = std::fopen(path, "wb");
    bool ok = fp != nullptr;
    for (const BitOut &p : parts) ok = ok && std::fwrite(p.bytes.data(), 1, p.len, fp) == p.len;
    if (fp && std::fclose(fp) != 0) ok = false;
    if (!ok) {
      glc::set_global_error(std::string("glc_flac_save: cannot write ") + path);
      return GLC_EIO;
    }
    return GLC_OK;
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  } catch (const std::exception &e) {
    glc::set_global_error(std::string("glc_flac_save: ") + e.what());
    return GLC_EINVAL;
  }
}

int glc_flac_decode(const uint8_t *buf, uint64_t len, float **samples, uint64_t *n_samples, uint32_t *sample_rate,
                    uint16_t *channels) {
  if (!buf || !samples || !n_samples || !sample_rate || !channels) return GLC_EINVAL;
  try {
    std::vector<float> pcm;
    const int rc = flac_decode(Span{buf, static_cast<size_t>(len)}, pcm, *sample_rate, *channels);
    if (rc != GLC_OK) return rc;
    float *o = static_cast<float *>(std::malloc((pcm.size() ? pcm.size() : 1) * sizeof(float)));
    if (!o) return GLC_ENOMEM;
    if (!pcm.empty()) std::memcpy(o, pcm.data(), pcm.size() * sizeof(float));
    *samples = o;
    *n_samples = pcm.size();
    return GLC_OK;
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  } catch (const std::exception &e) {
    glc::set_global_error(std::string("glc_flac_decode: ") + e.what());
    return GLC_EFORMAT;
  }
}

int glc_flac_load(const char *path, float **samples, uint64_t *n_samples, uint32_t *sample_rate, uint16_t *channels) {
  if (!path) return GLC_EINVAL;
  std::vector<uint8_t> f;
  bool opened = false;
  try {
    opened = read_file(path, f);
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  if (!opened) {
    glc::set_global_error(std::string("glc_flac_load: cannot open ") + path);
    return GLC_EIO;
  }
  return glc_flac_decode(f.data(), f.size(), samples, n_samples, sample_rate, channels);
}

}  // extern "C"
