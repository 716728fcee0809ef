// glc_wav.cpp — WAV reader / 16-bit WAV writer: the file-I/O twin of the reference's
// src/audio.rs (load_wav :39-64 via hound, export_to_wav :100-132).  Host only.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "glc_common.h"

namespace {

bool read_all(const char *path, std::vector<uint8_t> &buf) {
  FILE *fp = std::fopen(path, "rb");
  if (!fp) return false;
  uint8_t tmp[1 << 16];
  size_t got;
  while ((got = std::fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + got);
  std::fclose(fp);
  return true;
}

uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
uint16_t le16(const uint8_t *p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

}  // namespace

extern "C" {

void glc_free(void *p) { std::free(p); }

int glc_wav_load(const char *path, float **samples, uint64_t *n_samples, uint32_t *sample_rate,
                 uint16_t *channels) {
  if (!path || !samples || !n_samples || !sample_rate || !channels) return GLC_EINVAL;
  std::vector<uint8_t> f;
  bool opened = false;
  try {  // no C++ exception may cross the C ABI
    opened = read_all(path, f);
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  if (!opened) {
    glc::set_global_error(std::string("glc_wav_load: cannot open ") + path);
    return GLC_EIO;
  }
  if (f.size() < 12 || std::memcmp(f.data(), "RIFF", 4) || std::memcmp(f.data() + 8, "WAVE", 4)) {
    glc::set_global_error("glc_wav_load: not a RIFF/WAVE file");
    return GLC_EFORMAT;
  }
  uint16_t fmt = 0, ch = 0, bits = 0, block = 0;
  uint32_t sr = 0;
  const uint8_t *data = nullptr;
  uint64_t data_len = 0;
  size_t pos = 12;
  while (pos + 8 <= f.size()) {
    const uint32_t len = le32(&f[pos + 4]);
    const uint8_t *body = &f[pos + 8];
    const size_t avail = f.size() - pos - 8;
    if (!std::memcmp(&f[pos], "fmt ", 4) && len >= 16 && avail >= 16) {
      fmt = le16(body);
      ch = le16(body + 2);
      sr = le32(body + 4);
      block = le16(body + 12);
      bits = le16(body + 14);
      if (fmt == 0xFFFE && len >= 40 && avail >= 40) fmt = le16(body + 24);  // EXTENSIBLE sub-format
    } else if (!std::memcmp(&f[pos], "data", 4)) {
      data = body;
      data_len = len <= avail ? len : avail;  // tolerate a truncated / streaming length
      break;
    }
    pos += 8 + static_cast<size_t>(len) + (len & 1);
  }
  const bool is_float = fmt == 3 && bits == 32;
  const bool is_int = fmt == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32);
  if (!data || ch == 0 || (!is_float && !is_int) || block != ch * (bits / 8)) {
    glc::set_global_error("glc_wav_load: unsupported or malformed WAV (need PCM 8/16/24/32 or float 32)");
    return GLC_EFORMAT;
  }
  const uint32_t bps = bits / 8;
  const uint64_t n = data_len / bps;
  float *out = static_cast<float *>(std::malloc((n ? n : 1) * sizeof(float)));
  if (!out) return GLC_ENOMEM;
  // `(1 << (bits - 1)) as f32`, audio.rs:55: the literal is an i32, so for 32-bit samples the shift
  // lands on the sign bit and the divisor is i32::MIN = -2147483648.0 - the reference inverts the
  // polarity of 32-bit integer WAV files (quirk Q11, kept)
  const float max = bits == 32 ? -2147483648.0f : static_cast<float>(1u << (bits - 1));
  for (uint64_t i = 0; i < n; ++i) {
    const uint8_t *p = data + i * bps;
    if (is_float) {
      std::memcpy(&out[i], p, 4);
    } else {
      int32_t s;
      if (bits == 8) s = static_cast<int32_t>(p[0]) - 128;  // WAV 8-bit is unsigned
      else if (bits == 16) s = static_cast<int16_t>(le16(p));
      else if (bits == 24) s = (static_cast<int32_t>((p[0] | (p[1] << 8) | (static_cast<uint32_t>(p[2]) << 16)) << 8)) >> 8;
      else s = static_cast<int32_t>(le32(p));
      out[i] = static_cast<float>(s) / max;  // `s? as f32 / max`, audio.rs:58
    }
  }
  *samples = out;
  *n_samples = n;
  *sample_rate = sr;
  *channels = ch;
  return GLC_OK;
}

int glc_wav_save16(const char *path, const float *samples, uint64_t n_samples, uint32_t sample_rate,
                   uint16_t channels) {
  if (!path || (!samples && n_samples) || channels == 0) return GLC_EINVAL;
  const uint64_t data_bytes = n_samples * 2;
  if (data_bytes > 0xFFFFFFFFull - 36) {
    glc::set_global_error("glc_wav_save16: stream too long for a RIFF file");
    return GLC_EINVAL;
  }
  std::vector<uint8_t> buf;
  try {
    buf.resize(44 + data_bytes);
  } catch (const std::bad_alloc &) {
    return GLC_ENOMEM;
  }
  auto put32 = [&](size_t at, uint32_t v) { for (int i = 0; i < 4; ++i) buf[at + i] = (v >> (8 * i)) & 0xFF; };
  auto put16 = [&](size_t at, uint16_t v) { buf[at] = v & 0xFF; buf[at + 1] = v >> 8; };
  std::memcpy(&buf[0], "RIFF", 4);
  put32(4, static_cast<uint32_t>(36 + data_bytes));
  std::memcpy(&buf[8], "WAVEfmt ", 8);
  put32(16, 16);
  put16(20, 1);
  put16(22, channels);
  put32(24, sample_rate);
  put32(28, sample_rate * channels * 2);
  put16(32, static_cast<uint16_t>(channels * 2));
  put16(34, 16);
  std::memcpy(&buf[36], "data", 4);
  put32(40, static_cast<uint32_t>(data_bytes));
  for (uint64_t i = 0; i < n_samples; ++i) {
    float v = samples[i] * 32767.0f;  // convert_f32_to_i16, audio.rs:11-16
    int16_t q;
    if (v != v) q = 0;
    else {
      if (v < -32768.0f) v = -32768.0f;
      if (v > 32767.0f) v = 32767.0f;
      q = static_cast<int16_t>(v);
    }
    put16(44 + 2 * i, static_cast<uint16_t>(q));
  }
  FILE *fp = std::fopen(path, "wb");
  if (!fp) {
    glc::set_global_error(std::string("glc_wav_save16: cannot open ") + path);
    return GLC_EIO;
  }
  const size_t put = std::fwrite(buf.data(), 1, buf.size(), fp);
  if (std::fclose(fp) != 0 || put != buf.size()) {
    glc::set_global_error("glc_wav_save16: short write");
    return GLC_EIO;
  }
  return GLC_OK;
}

}  // extern "C"
