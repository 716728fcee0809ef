// glc_tables.cpp — host-side generation of the codec's constant tables.
//
// Replaces MdctTables::new (src/codec.rs:326-356) and PerceptualWeights::new /
// compute_critical_bands (src/codec.rs:102-183) of the reference.  The tables are part of the
// codec's definition: the cosine table is evaluated at f32-rounded angles of up to 8037 rad with
// the system libm's cosf (which is not correctly rounded), so a device-side or "exact" cosine
// would define a different transform.  They are therefore built once on the
// host, in binary32 with contraction off, and uploaded.
#include <cmath>

#include "glc_common.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace glc {

namespace {
constexpr float kPi = 3.14159265358979323846f;  // std::f32::consts::PI

// Piecewise perceptual weight of one bin, src/codec.rs:107-132.
float bin_weight(uint32_t k, float sr) {
  const float n = static_cast<float>(kHop);
  const float hz = (static_cast<float>(k) / (2.0f * n)) * sr;
  float w;
  if (hz < 100.0f)
    w = 0.3f + (hz / 100.0f) * 0.4f;
  else if (hz < 200.0f)
    w = 0.7f + ((hz - 100.0f) / 100.0f) * 0.3f;
  else if (hz < 5000.0f)
    w = 1.0f;
  else if (hz < 10000.0f)
    w = 1.0f - ((hz - 5000.0f) / 5000.0f) * 0.3f;
  else
    w = 0.7f - std::fmin((hz - 10000.0f) / 12000.0f, 1.0f) * 0.5f;
  return std::fmax(w, 0.2f);
}
}  // namespace

void build_host_tables(uint32_t sample_rate, HostTables &t) {
  t.sample_rate = sample_rate;
  const float n = static_cast<float>(kHop);

  // --- MdctTables::new ---------------------------------------------------------------
  t.cos_table.resize(static_cast<size_t>(kHop) * kFrame);
  t.cos_table_t.resize(static_cast<size_t>(kHop) * kFrame);
  const float pi_over_n = kPi / n;
  std::vector<float> pre(kFrame);  // (PI/n) * (i + 0.5 + n/2), rounded after each op
  for (uint32_t i = 0; i < kFrame; ++i) {
    const float shifted = (static_cast<float>(i) + 0.5f) + n / 2.0f;
    pre[i] = pi_over_n * shifted;
  }
  for (uint32_t k = 0; k < kHop; ++k) {
    const float kk = static_cast<float>(k) + 0.5f;
    float *row = &t.cos_table[static_cast<size_t>(k) * kFrame];
    for (uint32_t i = 0; i < kFrame; ++i) {
      const float angle = pre[i] * kk;
      const float c = cosf(angle);
      row[i] = c;
      t.cos_table_t[static_cast<size_t>(i) * kHop + k] = c;
    }
  }
  t.window.resize(kFrame);
  for (uint32_t i = 0; i < kFrame; ++i) {
    const float x = kPi * (static_cast<float>(i) + 0.5f);
    t.window[i] = sinf(x / static_cast<float>(kFrame));
  }
  t.norm = sqrtf(2.0f / n);

  // --- PerceptualWeights::new --------------------------------------------------------
  const float sr = static_cast<float>(sample_rate);
  t.weights.resize(kHop);
  for (uint32_t k = 0; k < kHop; ++k) t.weights[k] = bin_weight(k, sr);

  t.edges.clear();
  t.edges.push_back(0);
  const float nyquist = sr / 2.0f;
  for (float f = 0.0f; f < nyquist && t.edges.size() < 50;) {
    const uint32_t bin = static_cast<uint32_t>((f / nyquist) * n);
    if (bin > t.edges.back() && bin < kHop) t.edges.push_back(bin);
    f += f < 500.0f ? 50.0f : f < 2000.0f ? 100.0f : f < 8000.0f ? 250.0f : 500.0f;
  }
  t.edges.push_back(kHop);

  // --- frame-invariant parts of compute_masking_thresholds (src/codec.rs:218-228) -----
  const size_t nb = t.edges.size() - 1;
  t.band_pf.assign(nb, 0.f);
  t.band_len.assign(nb, 1.f);
  t.band_of.assign(kHop, 0);
  for (size_t b = 0; b < nb; ++b) {
    const uint32_t lo = t.edges[b], hi = t.edges[b + 1];
    float acc = 0.0f;
    for (uint32_t i = lo; i < hi; ++i) {
      acc = acc + t.weights[i];
      t.band_of[i] = static_cast<uint16_t>(b);
    }
    const float len = static_cast<float>(hi - lo);
    const float avg = acc / len;
    t.band_len[b] = len;
    t.band_pf[b] = 1.0f / std::fmax(avg, 0.1f);
  }
  t.indiv.resize(kHop);
  for (uint32_t k = 0; k < kHop; ++k) t.indiv[k] = 1.0f / std::fmax(t.weights[k], 0.1f);
  t.cf = std::fmax(1.0f - kQuality, 0.01f);
  t.noise_floor = powf(10.0f, kNoiseFloorDb / 20.0f);  // src/codec.rs:277
}

glc_plan plan_encode(uint64_t n_samples, uint16_t channels) {
  glc_plan p{};
  if (channels == 0) return p;  // `i % ch` panics, src/codec.rs:430
  const uint64_t ch = channels;
  auto per_channel = [&](uint64_t c) { return n_samples > c ? (n_samples - c + ch - 1) / ch : 0; };
  auto padded = [](uint64_t len) { return ((kHop / 2 + len + kHop - 1) / kHop) * kHop + kHop / 2; };
  const uint64_t l0 = per_channel(0);
  const uint64_t p0 = padded(l0);
  const uint64_t nf = p0 < kFrame ? 1 : (p0 - kFrame) / kHop + 1;  // :449-455
  const uint64_t last_end = (nf - 1) * kHop + kFrame;              // slice end, :474
  for (uint64_t c = 0; c < ch; ++c)
    if (padded(per_channel(c)) < last_end) return p;  // reference panics: slice out of range
  p.n_frames = nf;
  p.padded_len = p0;
  p.per_channel = l0;
  p.encoder_delay = kHop / 2;                                  // :547
  p.padding = static_cast<uint32_t>(p0 - l0 - kHop / 2);       // :546
  return p;
}

}  // namespace glc
