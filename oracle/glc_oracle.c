/*
 * glc_oracle.c — CPU restatement of /root/reference/src/codec.rs (TEST INFRASTRUCTURE ONLY).
 * See glc_oracle.h for the role of this file and the "PARITY UNPINNED" statement.
 *
 * Every float expression below is written so that gcc -O2 -ffp-contract=off evaluates it in
 * IEEE binary32 in exactly the association the Rust source uses (rustc never contracts or
 * re-associates).  Transcendentals (cosf/sinf/powf/log2f/sqrtf) come from the system libm,
 * which is what Rust's f32 methods call on Linux.
 */
#define _GNU_SOURCE
#include "glc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define N_HOP GLO_HOP_SIZE
#define N_FRAME GLO_FRAME_SIZE

/* src/codec.rs:15-29 */
static const float kNoiseFloorDb = -48.0f;
static const float kQuality = 0.7f;
static const uint32_t kMinBits = 8, kMaxBits = 16;
static const float kCompressionThreshold = 0.85f;
static const float kPi = 3.14159265358979323846f; /* std::f32::consts::PI */

/* ---------------------------------------------------------------- tables */

void glo_tables(float *table, float *window, float *norm) {
  /* src/codec.rs:331-338: angle = PI / n * (i + 0.5 + n/2) * (k + 0.5), left to right in f32 */
  const float n = (float)N_HOP;
  for (uint32_t k = 0; k < N_HOP; ++k) {
    for (uint32_t i = 0; i < N_FRAME; ++i) {
      float a = kPi / n;
      float b = ((float)i + 0.5f) + n / 2.0f;
      float ab = a * b;
      float angle = ab * ((float)k + 0.5f);
      table[(size_t)k * N_FRAME + i] = cosf(angle);
    }
  }
  /* src/codec.rs:342-344 */
  for (uint32_t i = 0; i < N_FRAME; ++i) {
    float num = kPi * ((float)i + 0.5f);
    window[i] = sinf(num / (float)N_FRAME);
  }
  /* src/codec.rs:347 */
  *norm = sqrtf(2.0f / n);
}

void glo_mdct_block(const float *table, float norm, const float *block, float *out) {
  /* src/codec.rs:363-373: strictly sequential, multiply then add */
  for (uint32_t k = 0; k < N_HOP; ++k) {
    const float *tb = table + (size_t)k * N_FRAME;
    float s = 0.0f;
    for (uint32_t i = 0; i < N_FRAME; ++i) {
      float p = block[i] * tb[i];
      s = s + p;
    }
    out[k] = s * norm;
  }
}

void glo_imdct_block(const float *table, float norm, const float *coeffs, float *out) {
  /* src/codec.rs:380-389 */
  for (uint32_t i = 0; i < N_FRAME; ++i) {
    float s = 0.0f;
    for (uint32_t k = 0; k < N_HOP; ++k) {
      float p = coeffs[k] * table[(size_t)k * N_FRAME + i];
      s = s + p;
    }
    out[i] = s * norm;
  }
}

/* ---------------------------------------------------------------- perceptual model */

uint32_t glo_perceptual(uint32_t sample_rate, float *weights, uint32_t *edges) {
  const float n = (float)N_HOP;
  const float sr = (float)sample_rate;
  /* src/codec.rs:104-133 */
  for (uint32_t k = 0; k < N_HOP; ++k) {
    float norm_freq = (float)k / (2.0f * n);
    float f = norm_freq * sr;
    float w;
    if (f < 100.0f) {
      w = 0.3f + (f / 100.0f) * 0.4f;
    } else if (f < 200.0f) {
      w = 0.7f + ((f - 100.0f) / 100.0f) * 0.3f;
    } else if (f < 5000.0f) {
      w = 1.0f;
    } else if (f < 10000.0f) {
      w = 1.0f - ((f - 5000.0f) / 5000.0f) * 0.3f;
    } else {
      w = 0.7f - fminf((f - 10000.0f) / 12000.0f, 1.0f) * 0.5f;
    }
    weights[k] = fmaxf(w, 0.2f);
  }
  /* src/codec.rs:146-183 */
  uint32_t nb = 0;
  edges[nb++] = 0;
  const float nyq = sr / 2.0f;
  float freq = 0.0f;
  while (freq < nyq && nb < 50) {
    float r = (freq / nyq) * n;
    uint32_t bin = (uint32_t)r; /* `as usize`: truncation, r >= 0 */
    if (bin > edges[nb - 1] && bin < N_HOP) edges[nb++] = bin;
    if (freq < 500.0f)
      freq += 50.0f;
    else if (freq < 2000.0f)
      freq += 100.0f;
    else if (freq < 8000.0f)
      freq += 250.0f;
    else
      freq += 500.0f;
  }
  edges[nb++] = N_HOP;
  return nb;
}

static float abs_max_floor(const float *c, float floor_) {
  /* iter().map(abs).fold(0.0, f32::max).max(floor) — src/codec.rs:198,278,488 */
  float m = 0.0f;
  for (uint32_t k = 0; k < N_HOP; ++k) m = fmaxf(m, fabsf(c[k]));
  return fmaxf(m, floor_);
}

void glo_thresholds(const float *coeffs, const float *weights, const uint32_t *edges,
                    uint32_t n_edges, float *thr) {
  memset(thr, 0, sizeof(float) * N_HOP);
  const float gmax = abs_max_floor(coeffs, 1e-10f);
  for (uint32_t b = 0; b + 1 < n_edges; ++b) {
    uint32_t start = edges[b];
    uint32_t end = edges[b + 1] < N_HOP ? edges[b + 1] : N_HOP;
    if (start >= end) continue;
    const float len = (float)(end - start);
    /* src/codec.rs:212-215.  (Rust's f32 Sum starts at -0.0 on recent toolchains; adding a
     * square (never -0.0... +0.0 or positive) to either zero gives the same bits.) */
    float ss = 0.0f;
    for (uint32_t i = start; i < end; ++i) {
      float sq = coeffs[i] * coeffs[i];
      ss = ss + sq;
    }
    float energy = sqrtf(ss / len);
    /* :218 */
    float ws = 0.0f;
    for (uint32_t i = start; i < end; ++i) ws = ws + weights[i];
    float avg_w = ws / len;
    /* :221-223 */
    float cf = fmaxf(1.0f - kQuality, 0.01f);
    float pf = 1.0f / fmaxf(avg_w, 0.1f);
    float base = ((energy * 0.01f) * cf) * pf;
    /* :226-236 */
    for (uint32_t i = start; i < end; ++i) {
      float indiv = 1.0f / fmaxf(weights[i], 0.1f);
      float t = base * indiv;
      if (fabsf(coeffs[i]) > gmax * 0.3f) t = fminf(t, gmax * 0.05f);
      thr[i] = t;
    }
  }
}

/* src/codec.rs:243-267 — kept for fidelity; cannot return 0 once abs_val > threshold. */
static uint32_t quant_bits_fast(float abs_val, float threshold, float gmax) {
  if (abs_val <= threshold) return 0;
  float importance = fmaxf(log2f(abs_val / threshold), 0.0f);
  float rel = abs_val / gmax;
  float score = importance * 0.3f + rel * 0.7f;
  float scaled = score * (float)(kMaxBits - kMinBits);
  /* Rust `as u32` saturates (NaN -> 0); release-mode `+` wraps. */
  uint32_t add;
  if (!(scaled > 0.0f))
    add = 0;
  else if (scaled >= 4294967296.0f)
    add = 0xFFFFFFFFu;
  else
    add = (uint32_t)scaled;
  uint32_t bits = kMinBits + add;
  if (bits < kMinBits) bits = kMinBits;
  if (bits > kMaxBits) bits = kMaxBits;
  return bits;
}

static int16_t sat_i16(float v) {
  /* clamp(-32768, 32767) then `as i16` (trunc; NaN -> 0) — src/codec.rs:301,501 */
  if (v != v) return 0;
  if (v < -32768.0f) v = -32768.0f;
  if (v > 32767.0f) v = 32767.0f;
  return (int16_t)v;
}

uint32_t glo_compress(const float *coeffs, float scale, const float *thr, uint16_t *idx,
                      int16_t *q) {
  /* src/codec.rs:277-308 */
  const float nfl = powf(10.0f, kNoiseFloorDb / 20.0f) * scale;
  const float gmax = abs_max_floor(coeffs, 1e-10f);
  const float max_q = 32768.0f;
  uint32_t nnz = 0;
  for (uint32_t k = 0; k < N_HOP; ++k) {
    float c = coeffs[k];
    float a = fabsf(c);
    float t = thr[k] * scale;
    if (a > nfl && a > t) {
      if (quant_bits_fast(a, t, gmax) == 0) continue;
      float normalized = c / scale;
      float quantized = roundf(normalized * max_q);
      int16_t qi = sat_i16(quantized);
      if (qi != 0) {
        idx[nnz] = (uint16_t)k;
        q[nnz] = qi;
        ++nnz;
      }
    }
  }
  return nnz;
}

/* ---------------------------------------------------------------- encode driver */

uint64_t glo_num_frames(uint64_t n_samples, uint16_t channels) {
  if (channels == 0) return 0; /* `i % ch` panics, :430 */
  const uint64_t ch = channels;
  /* per-channel lengths after `per_chan[i % ch].push` (:428-431) */
  uint64_t l0 = (n_samples + ch - 1) / ch;
  uint64_t p0 = ((512 + l0 + 1023) / 1024) * 1024 + 512; /* :438-445 */
  uint64_t nf = p0 < N_FRAME ? 1 : (p0 - N_FRAME) / N_HOP + 1;
  uint64_t need = (nf - 1) * N_HOP + N_FRAME; /* slice end at :474 */
  for (uint64_t c = 0; c < ch; ++c) {
    uint64_t lc = n_samples > c ? (n_samples - c + ch - 1) / ch : 0;
    uint64_t pc = ((512 + lc + 1023) / 1024) * 1024 + 512;
    if (pc < need) return 0; /* slice out of range -> panic */
  }
  return nf;
}

typedef struct enc_frame {
  uint8_t is_raw;
  uint32_t *nnz;   /* [ch] */
  float *scale;    /* [ch] */
  uint16_t **idx;  /* [ch][nnz] */
  int16_t **q;     /* [ch][nnz] */
  int16_t *raw;    /* [ch*2048] planar, Q1 */
} enc_frame;

typedef struct enc_job {
  uint32_t ch;
  uint64_t f0, nf;
  float **padded;
  uint64_t pad_base; /* padded[c][0] is element pad_base of the channel's padded array (0: whole array) */
  uint64_t tap_f0;   /* taps row = (frame - tap_f0)*ch + c */
  const float *table, *window, *weights;
  const uint32_t *edges;
  uint32_t n_edges;
  float norm;
  enc_frame *frames; /* NULL in timing mode */
  const glo_taps *taps;
  atomic_ullong next;
  atomic_ullong sink;
} enc_job;

static void encode_one_frame(enc_job *J, uint64_t fi, float *block, float *coeffs, float *thr,
                             uint16_t *tidx, int16_t *tq, int16_t *raw_tmp, uint32_t *nnz_tmp,
                             float *scale_tmp, uint16_t *idx_all, int16_t *q_all) {
  const uint32_t ch = J->ch;
  for (uint32_t c = 0; c < ch; ++c) {
    const float *slice = J->padded[c] + (fi * N_HOP - J->pad_base); /* :473-474 */
    for (uint32_t i = 0; i < N_FRAME; ++i) block[i] = slice[i] * J->window[i]; /* :476-481 */
    glo_mdct_block(J->table, J->norm, block, coeffs);                         /* :485 */
    float max_val = abs_max_floor(coeffs, 1e-10f);                            /* :488 */
    scale_tmp[c] = max_val;
    glo_thresholds(coeffs, J->weights, J->edges, J->n_edges, thr);            /* :492 */
    uint32_t n = glo_compress(coeffs, max_val, thr, tidx, tq);                /* :493 */
    nnz_tmp[c] = n;
    memcpy(idx_all + (size_t)c * N_HOP, tidx, n * sizeof(uint16_t));
    memcpy(q_all + (size_t)c * N_HOP, tq, n * sizeof(int16_t));
    for (uint32_t i = 0; i < N_FRAME; ++i) { /* :498-502, channel-planar (Q1) */
      float s = slice[i] * J->window[i];
      raw_tmp[(size_t)c * N_FRAME + i] = sat_i16(s * 32767.0f);
    }
    if (J->taps) {
      size_t m = (size_t)(fi - J->tap_f0) * ch + c;
      if (J->taps->coeffs) memcpy(J->taps->coeffs + m * N_HOP, coeffs, sizeof(float) * N_HOP);
      if (J->taps->scales) J->taps->scales[m] = max_val;
      if (J->taps->nnz) J->taps->nnz[m] = n;
      if (J->taps->dense_q) {
        int16_t *d = J->taps->dense_q + m * N_HOP;
        memset(d, 0, sizeof(int16_t) * N_HOP);
        for (uint32_t j = 0; j < n; ++j) d[tidx[j]] = tq[j];
      }
    }
  }
  /* :505-521 */
  size_t compressed = 0;
  for (uint32_t c = 0; c < ch; ++c) compressed += 8 + (size_t)nnz_tmp[c] * 4;
  compressed += 8 + (size_t)ch * 4;
  compressed += 64;
  size_t raw_size = (size_t)N_FRAME * ch * 2;
  int use_raw = (float)compressed >= ((float)raw_size * kCompressionThreshold);
  if (J->taps && J->taps->is_raw) J->taps->is_raw[fi - J->tap_f0] = (uint8_t)use_raw;

  if (!J->frames) { /* timing mode: fold results so nothing is dead */
    unsigned long long acc = (unsigned long long)use_raw;
    for (uint32_t c = 0; c < ch; ++c) acc += nnz_tmp[c] + (unsigned)raw_tmp[(size_t)c * N_FRAME + 7];
    atomic_fetch_add(&J->sink, acc);
    return;
  }
  enc_frame *F = &J->frames[fi - J->f0];
  F->is_raw = (uint8_t)use_raw;
  /* scale / nnz are kept for raw frames too: the reference computes them before it decides
   * (:488-493) and the device records carry them in every frame's header */
  F->nnz = (uint32_t *)malloc(sizeof(uint32_t) * ch);
  F->scale = (float *)malloc(sizeof(float) * ch);
  for (uint32_t c = 0; c < ch; ++c) {
    F->nnz[c] = nnz_tmp[c];
    F->scale[c] = scale_tmp[c];
  }
  if (use_raw) {
    F->raw = (int16_t *)malloc(sizeof(int16_t) * N_FRAME * ch);
    memcpy(F->raw, raw_tmp, sizeof(int16_t) * N_FRAME * ch);
  } else {
    F->idx = (uint16_t **)malloc(sizeof(uint16_t *) * ch);
    F->q = (int16_t **)malloc(sizeof(int16_t *) * ch);
    for (uint32_t c = 0; c < ch; ++c) {
      uint32_t n = nnz_tmp[c];
      F->idx[c] = (uint16_t *)malloc(sizeof(uint16_t) * (n ? n : 1));
      F->q[c] = (int16_t *)malloc(sizeof(int16_t) * (n ? n : 1));
      memcpy(F->idx[c], idx_all + (size_t)c * N_HOP, n * sizeof(uint16_t));
      memcpy(F->q[c], q_all + (size_t)c * N_HOP, n * sizeof(int16_t));
    }
  }
}

static void *encode_worker(void *arg) {
  enc_job *J = (enc_job *)arg;
  const uint32_t ch = J->ch;
  float *block = (float *)malloc(sizeof(float) * N_FRAME);
  float *coeffs = (float *)malloc(sizeof(float) * N_HOP);
  float *thr = (float *)malloc(sizeof(float) * N_HOP);
  uint16_t *tidx = (uint16_t *)malloc(sizeof(uint16_t) * N_HOP);
  int16_t *tq = (int16_t *)malloc(sizeof(int16_t) * N_HOP);
  int16_t *raw_tmp = (int16_t *)malloc(sizeof(int16_t) * N_FRAME * ch);
  uint32_t *nnz_tmp = (uint32_t *)malloc(sizeof(uint32_t) * ch);
  float *scale_tmp = (float *)malloc(sizeof(float) * ch);
  uint16_t *idx_all = (uint16_t *)malloc(sizeof(uint16_t) * N_HOP * ch);
  int16_t *q_all = (int16_t *)malloc(sizeof(int16_t) * N_HOP * ch);
  for (;;) {
    unsigned long long k = atomic_fetch_add(&J->next, 1ull);
    if (k >= J->nf) break;
    encode_one_frame(J, J->f0 + k, block, coeffs, thr, tidx, tq, raw_tmp, nnz_tmp, scale_tmp,
                     idx_all, q_all);
  }
  free(block); free(coeffs); free(thr); free(tidx); free(tq); free(raw_tmp);
  free(nnz_tmp); free(scale_tmp); free(idx_all); free(q_all);
  return NULL;
}

static int resolve_threads(int n) {
  if (n > 0) return n;
  long c = sysconf(_SC_NPROCESSORS_ONLN);
  return c > 0 ? (int)c : 1;
}

static void run_workers(void *(*fn)(void *), void *arg, int n_threads) {
  n_threads = resolve_threads(n_threads);
  if (n_threads == 1) {
    fn(arg);
    return;
  }
  pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
  for (int i = 0; i < n_threads; ++i) pthread_create(&t[i], NULL, fn, arg);
  for (int i = 0; i < n_threads; ++i) pthread_join(t[i], NULL);
  free(t);
}

/* deinterleave + pad, src/codec.rs:426-447; returns per-channel arrays and padded_len[0]. */
static float **build_padded(const float *pcm, uint64_t n, uint32_t ch, uint64_t *p0_out,
                            uint64_t *l0_out) {
  float **padded = (float **)malloc(sizeof(float *) * ch);
  for (uint32_t c = 0; c < ch; ++c) {
    uint64_t lc = n > c ? (n - c + ch - 1) / ch : 0;
    uint64_t pc = ((512 + lc + 1023) / 1024) * 1024 + 512;
    padded[c] = (float *)calloc(pc, sizeof(float));
    for (uint64_t t = 0; t < lc; ++t) padded[c][512 + t] = pcm[t * ch + c];
    if (c == 0) {
      *p0_out = pc;
      *l0_out = lc;
    }
  }
  return padded;
}

typedef struct wbuf {
  uint8_t *p;
  uint64_t len, cap;
} wbuf;
static void wb_put(wbuf *w, const void *src, uint64_t n) {
  if (w->len + n > w->cap) {
    uint64_t nc = w->cap ? w->cap * 2 : 4096;
    while (nc < w->len + n) nc *= 2;
    w->p = (uint8_t *)realloc(w->p, nc);
    w->cap = nc;
  }
  memcpy(w->p + w->len, src, n);
  w->len += n;
}
static void wb_u8(wbuf *w, uint8_t v) { wb_put(w, &v, 1); }
static void wb_u16(wbuf *w, uint16_t v) { wb_put(w, &v, 2); }
static void wb_u32(wbuf *w, uint32_t v) { wb_put(w, &v, 4); }
static void wb_u64(wbuf *w, uint64_t v) { wb_put(w, &v, 8); }

int glo_encode(uint32_t sample_rate, const float *pcm, uint64_t n_samples, uint16_t channels,
               int n_threads, uint8_t **out_bytes, uint64_t *out_len, const glo_taps *taps) {
  uint64_t nf = glo_num_frames(n_samples, channels);
  if (nf == 0) return -1;
  const uint32_t ch = channels;
  float *table = (float *)malloc(sizeof(float) * N_HOP * N_FRAME);
  float window[N_FRAME], weights[N_HOP], norm;
  uint32_t edges[GLO_MAX_BANDS];
  glo_tables(table, window, &norm);
  uint32_t n_edges = glo_perceptual(sample_rate, weights, edges);

  uint64_t p0 = 0, l0 = 0;
  float **padded = build_padded(pcm, n_samples, ch, &p0, &l0);

  enc_job J;
  memset(&J, 0, sizeof J);
  J.ch = ch; J.f0 = 0; J.nf = nf; J.padded = padded; J.table = table; J.window = window;
  J.weights = weights; J.edges = edges; J.n_edges = n_edges; J.norm = norm; J.taps = taps;
  J.frames = (enc_frame *)calloc(nf, sizeof(enc_frame));
  atomic_init(&J.next, 0);
  atomic_init(&J.sink, 0);
  run_workers(encode_worker, &J, n_threads);

  /* bincode 1.x default config (src/codec.rs:776): LE fixed ints, u64 lengths, u8 Option tag.
   * Field order = struct order at src/codec.rs:31-69. */
  wbuf w = {0};
  wb_u32(&w, sample_rate);
  wb_u16(&w, channels);
  wb_u64(&w, n_samples); /* total_samples, :423 */
  wb_u64(&w, nf);
  for (uint64_t f = 0; f < nf; ++f) {
    enc_frame *F = &J.frames[f];
    if (F->is_raw) {
      wb_u64(&w, 0); /* sparse_coeffs_per_channel: empty */
      wb_u64(&w, 0); /* scale_factors: empty */
      wb_u8(&w, 1);
      wb_u64(&w, (uint64_t)N_FRAME * ch);
      wb_put(&w, F->raw, sizeof(int16_t) * N_FRAME * ch);
      free(F->raw); free(F->nnz); free(F->scale);
    } else {
      wb_u64(&w, ch);
      for (uint32_t c = 0; c < ch; ++c) {
        wb_u64(&w, F->nnz[c]);
        for (uint32_t j = 0; j < F->nnz[c]; ++j) {
          wb_u16(&w, F->idx[c][j]);
          wb_u16(&w, (uint16_t)F->q[c][j]);
        }
        free(F->idx[c]);
        free(F->q[c]);
      }
      wb_u64(&w, ch);
      for (uint32_t c = 0; c < ch; ++c) wb_put(&w, &F->scale[c], 4);
      wb_u8(&w, 0);
      free(F->idx); free(F->q); free(F->nnz); free(F->scale);
    }
  }
  /* :544-547 */
  wb_u32(&w, 512u);
  wb_u32(&w, (uint32_t)(p0 - l0 - 512));
  wb_u64(&w, n_samples);

  for (uint32_t c = 0; c < ch; ++c) free(padded[c]);
  free(padded); free(J.frames); free(table);
  *out_bytes = w.p;
  *out_len = w.len;
  return 0;
}

double glo_time_encode_frames(uint32_t sample_rate, const float *pcm, uint64_t n_samples,
                              uint16_t channels, uint64_t f0, uint64_t n_frames, int n_threads) {
  uint64_t nf = glo_num_frames(n_samples, channels);
  if (nf == 0 || f0 >= nf) return -1.0;
  if (f0 + n_frames > nf) n_frames = nf - f0;
  const uint32_t ch = channels;
  float *table = (float *)malloc(sizeof(float) * N_HOP * N_FRAME);
  float window[N_FRAME], weights[N_HOP], norm;
  uint32_t edges[GLO_MAX_BANDS];
  glo_tables(table, window, &norm);
  uint32_t n_edges = glo_perceptual(sample_rate, weights, edges);
  uint64_t p0 = 0, l0 = 0;
  float **padded = build_padded(pcm, n_samples, ch, &p0, &l0);
  enc_job J;
  memset(&J, 0, sizeof J);
  J.ch = ch; J.f0 = f0; J.nf = n_frames; J.padded = padded; J.table = table; J.window = window;
  J.weights = weights; J.edges = edges; J.n_edges = n_edges; J.norm = norm;
  atomic_init(&J.next, 0);
  atomic_init(&J.sink, 0);
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  run_workers(encode_worker, &J, n_threads);
  clock_gettime(CLOCK_MONOTONIC, &b);
  for (uint32_t c = 0; c < ch; ++c) free(padded[c]);
  free(padded); free(table);
  if (atomic_load(&J.sink) == 0xFFFFFFFFFFFFFFFFull) return -2.0;
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

/* Frames [f0, f1) of a stream of n_samples interleaved samples, computed from ONE SHARD of its
 * PCM: `shard` holds per-channel samples [t0, t0 + t_count) (interleaved, shard[0] = sample t0 of
 * channel 0).  Samples of the stream that lie outside the shard are poisoned with NaN, so a frame
 * range whose halo is missing cannot silently agree with anything; samples outside the stream are
 * the encoder's zero padding (src/codec.rs:433-447).  Output: the fixed-size frame records of the
 * device path (include/glc.h glc_record_bytes): u32 is_raw | u32 0 | ch x {f32 scale, u32 nnz} |
 * pad to 16 | i16 payload[ch][2048] (dense quantised row in [0, 1024) or the planar raw plane).
 * Only the span of padded samples the range reads is materialised, so windows at the far end of
 * hour-long streams cost nothing. */
int glo_encode_range_records(uint32_t sample_rate, const float *shard, uint64_t t0, uint64_t t_count,
                             uint64_t n_samples, uint16_t channels, uint64_t f0, uint64_t f1,
                             int n_threads, uint8_t *records, const glo_taps *taps) {
  uint64_t nf = glo_num_frames(n_samples, channels);
  if (nf == 0 || f0 > f1 || f1 > nf) return -1;
  if (f0 == f1) return 0;
  const uint32_t ch = channels;
  float *table = (float *)malloc(sizeof(float) * N_HOP * N_FRAME);
  float window[N_FRAME], weights[N_HOP], norm;
  uint32_t edges[GLO_MAX_BANDS];
  glo_tables(table, window, &norm);
  uint32_t n_edges = glo_perceptual(sample_rate, weights, edges);
  const uint64_t base = f0 * N_HOP, span = (f1 - 1 - f0) * N_HOP + N_FRAME;
  float **padded = (float **)malloc(sizeof(float *) * ch);
  for (uint32_t c = 0; c < ch; ++c) {
    uint64_t lc = n_samples > c ? (n_samples - c + ch - 1) / ch : 0; /* per_chan[c].len() */
    padded[c] = (float *)calloc(span, sizeof(float));
    for (uint64_t j = 0; j < span; ++j) {
      uint64_t pi = base + j; /* index into padded[c]; real sample t = pi - 512 */
      if (pi < 512) continue;
      uint64_t t = pi - 512;
      if (t >= lc) continue; /* trailing padding */
      if (t < t0 || t - t0 >= t_count) padded[c][j] = NAN; /* inside the stream, outside the shard */
      else padded[c][j] = shard[(t - t0) * ch + c];
    }
  }
  enc_job J;
  memset(&J, 0, sizeof J);
  J.ch = ch; J.f0 = f0; J.nf = f1 - f0; J.padded = padded; J.pad_base = base; J.tap_f0 = f0;
  J.table = table; J.window = window; J.weights = weights; J.edges = edges; J.n_edges = n_edges;
  J.norm = norm; J.taps = taps;
  J.frames = (enc_frame *)calloc(f1 - f0, sizeof(enc_frame));
  atomic_init(&J.next, 0);
  atomic_init(&J.sink, 0);
  run_workers(encode_worker, &J, n_threads);
  const uint64_t hdr = ((8ull + 8ull * ch) + 15ull) & ~15ull, rec = hdr + 2ull * N_FRAME * ch;
  for (uint64_t k = 0; k < f1 - f0; ++k) {
    enc_frame *F = &J.frames[k];
    uint8_t *r = records + k * rec;
    memset(r, 0, rec);
    uint32_t is_raw = F->is_raw;
    memcpy(r, &is_raw, 4);
    int16_t *pay = (int16_t *)(r + hdr);
    for (uint32_t c = 0; c < ch; ++c) {
      memcpy(r + 8 + 8 * c, &F->scale[c], 4);
      memcpy(r + 8 + 8 * c + 4, &F->nnz[c], 4);
    }
    if (F->is_raw) {
      memcpy(pay, F->raw, sizeof(int16_t) * N_FRAME * ch);
      free(F->raw); free(F->nnz); free(F->scale);
    } else {
      for (uint32_t c = 0; c < ch; ++c) {
        for (uint32_t j = 0; j < F->nnz[c]; ++j) pay[(size_t)c * N_FRAME + F->idx[c][j]] = F->q[c][j];
        free(F->idx[c]);
        free(F->q[c]);
      }
      free(F->idx); free(F->q); free(F->nnz); free(F->scale);
    }
  }
  for (uint32_t c = 0; c < ch; ++c) free(padded[c]);
  free(padded); free(J.frames); free(table);
  return 0;
}

/* ---------------------------------------------------------------- decode */

typedef struct rbuf {
  const uint8_t *p;
  uint64_t len, pos;
  int bad;
} rbuf;
static void rb_get(rbuf *r, void *dst, uint64_t n) {
  if (r->bad || n > r->len - r->pos) {
    r->bad = 1;
    memset(dst, 0, n);
    return;
  }
  memcpy(dst, r->p + r->pos, n);
  r->pos += n;
}
static uint64_t rb_u64(rbuf *r) { uint64_t v; rb_get(r, &v, 8); return v; }
static uint32_t rb_u32(rbuf *r) { uint32_t v; rb_get(r, &v, 4); return v; }
static uint16_t rb_u16(rbuf *r) { uint16_t v; rb_get(r, &v, 2); return v; }
static uint8_t rb_u8(rbuf *r) { uint8_t v; rb_get(r, &v, 1); return v; }

typedef struct dec_frame {
  uint8_t is_raw;
  uint64_t n_ch;         /* sparse_coeffs_per_channel.len() */
  uint64_t *nnz;         /* [n_ch] */
  const uint8_t **pairs; /* [n_ch] -> packed (u16,i16) in the byte stream */
  uint64_t n_scales;
  const uint8_t *scales;
  uint64_t raw_len;
  const uint8_t *raw;
} dec_frame;

typedef struct dec_job {
  uint32_t ch;
  const dec_frame *frames;
  uint64_t f0, nf;
  const float *table, *window;
  float norm;
  float *blocks; /* [nf][ch][2048] */
  atomic_ullong next;
  int bad;
} dec_job;

static void *decode_worker(void *arg) {
  dec_job *J = (dec_job *)arg;
  const uint32_t ch = J->ch;
  float coeffs[N_HOP];
  for (;;) {
    unsigned long long k = atomic_fetch_add(&J->next, 1ull);
    if (k >= J->nf) break;
    const dec_frame *F = &J->frames[J->f0 + k];
    float *out = J->blocks + (size_t)k * ch * N_FRAME;
    if (F->is_raw) {
      /* src/codec.rs:626-644: read as if interleaved (Q1), no window (Q2) */
      for (uint32_t c = 0; c < ch; ++c) {
        float *blk = out + (size_t)c * N_FRAME;
        for (uint32_t i = 0; i < N_FRAME; ++i) {
          uint64_t si = (uint64_t)i * ch + c;
          float v = 0.0f;
          if (si < F->raw_len) {
            int16_t s;
            memcpy(&s, F->raw + si * 2, 2);
            v = (float)s / 32767.0f;
          }
          blk[i] = v;
        }
      }
    } else {
      for (uint32_t c = 0; c < ch; ++c) {
        /* :651-665 */
        for (uint32_t i = 0; i < N_HOP; ++i) coeffs[i] = 0.0f;
        float scale;
        memcpy(&scale, F->scales + (size_t)c * 4, 4);
        scale = fmaxf(scale, 1e-12f);
        const uint8_t *pp = F->pairs[c];
        for (uint64_t j = 0; j < F->nnz[c]; ++j) {
          uint16_t index;
          int16_t qv;
          memcpy(&index, pp + j * 4, 2);
          memcpy(&qv, pp + j * 4 + 2, 2);
          if (index < N_HOP) coeffs[index] = ((float)qv / 32768.0f) * scale;
        }
        float *blk = out + (size_t)c * N_FRAME;
        glo_imdct_block(J->table, J->norm, coeffs, blk);            /* :669 */
        for (uint32_t i = 0; i < N_FRAME; ++i) blk[i] *= J->window[i]; /* :672-675 */
      }
    }
  }
  return NULL;
}

int glo_decode(const uint8_t *bytes, uint64_t len, int n_threads, float **out_pcm,
               uint64_t *out_n, uint32_t *sample_rate, uint16_t *channels) {
  rbuf r = {bytes, len, 0, 0};
  uint32_t sr = rb_u32(&r);
  uint16_t chs = rb_u16(&r);
  (void)rb_u64(&r); /* total_samples */
  uint64_t nf = rb_u64(&r);
  if (r.bad || nf > len) return -1;
  const uint32_t ch = chs;
  dec_frame *frames = (dec_frame *)calloc(nf ? nf : 1, sizeof(dec_frame));
  int bad = 0;
  for (uint64_t f = 0; f < nf && !bad; ++f) {
    dec_frame *F = &frames[f];
    F->n_ch = rb_u64(&r);
    if (r.bad || F->n_ch > len) { bad = 1; break; }
    F->nnz = (uint64_t *)calloc(F->n_ch ? F->n_ch : 1, sizeof(uint64_t));
    F->pairs = (const uint8_t **)calloc(F->n_ch ? F->n_ch : 1, sizeof(uint8_t *));
    for (uint64_t c = 0; c < F->n_ch; ++c) {
      uint64_t n = rb_u64(&r);
      if (r.bad || n > (len - r.pos) / 4) { bad = 1; break; }
      F->nnz[c] = n;
      F->pairs[c] = bytes + r.pos;
      r.pos += n * 4;
    }
    if (bad) break;
    F->n_scales = rb_u64(&r);
    if (r.bad || F->n_scales > (len - r.pos) / 4) { bad = 1; break; }
    F->scales = bytes + r.pos;
    r.pos += F->n_scales * 4;
    uint8_t tag = rb_u8(&r);
    if (r.bad || tag > 1) { bad = 1; break; }
    F->is_raw = tag;
    if (tag) {
      F->raw_len = rb_u64(&r);
      if (r.bad || F->raw_len > (len - r.pos) / 2) { bad = 1; break; }
      F->raw = bytes + r.pos;
      r.pos += F->raw_len * 2;
    } else if (F->n_ch < ch || F->n_scales < ch) {
      bad = 1; /* reference would index out of bounds (:652-653) */
    }
  }
  uint32_t enc_delay = rb_u32(&r);
  (void)rb_u32(&r);
  uint64_t orig_len = rb_u64(&r);
  if (r.bad || r.pos != len) bad = 1;
  if (bad || ch == 0) {
    for (uint64_t f = 0; f < nf; ++f) { free(frames[f].nnz); free((void *)frames[f].pairs); }
    free(frames);
    return -1;
  }

  float *table = (float *)malloc(sizeof(float) * N_HOP * N_FRAME);
  float window[N_FRAME], norm;
  glo_tables(table, window, &norm);

  /* all = (nf + 1) * 1024 * ch samples, :688-729 */
  uint64_t total = (nf + 1) * (uint64_t)N_HOP * ch;
  float *all = (float *)malloc(sizeof(float) * (total ? total : 1));
  float *overlap = (float *)calloc((size_t)ch * N_HOP, sizeof(float));
  const uint64_t BATCH = 256;
  float *blocks = (float *)malloc(sizeof(float) * BATCH * ch * N_FRAME);
  uint64_t wpos = 0;
  for (uint64_t f0 = 0; f0 < nf; f0 += BATCH) {
    uint64_t nb = nf - f0 < BATCH ? nf - f0 : BATCH;
    dec_job J;
    memset(&J, 0, sizeof J);
    J.ch = ch; J.frames = frames; J.f0 = f0; J.nf = nb; J.table = table; J.window = window;
    J.norm = norm; J.blocks = blocks;
    atomic_init(&J.next, 0);
    run_workers(decode_worker, &J, n_threads);
    for (uint64_t k = 0; k < nb; ++k) {
      const float *blk = blocks + (size_t)k * ch * N_FRAME;
      for (uint32_t i = 0; i < N_HOP; ++i)
        for (uint32_t c = 0; c < ch; ++c)
          all[wpos++] = overlap[(size_t)c * N_HOP + i] + blk[(size_t)c * N_FRAME + i]; /* :695 */
      for (uint32_t c = 0; c < ch; ++c)
        memcpy(overlap + (size_t)c * N_HOP, blk + (size_t)c * N_FRAME + N_HOP,
               sizeof(float) * N_HOP); /* :701-705 */
    }
  }
  for (uint32_t i = 0; i < N_HOP; ++i) /* :723-729 */
    for (uint32_t c = 0; c < ch; ++c) all[wpos++] = overlap[(size_t)c * N_HOP + i];

  /* :756-765 */
  uint64_t n_all = wpos, start = 0;
  if (n_all > enc_delay) { start = enc_delay; n_all -= enc_delay; }
  if (n_all > orig_len) n_all = orig_len;
  float *out = (float *)malloc(sizeof(float) * (n_all ? n_all : 1));
  memcpy(out, all + start, sizeof(float) * n_all);

  free(all); free(overlap); free(blocks); free(table);
  for (uint64_t f = 0; f < nf; ++f) { free(frames[f].nnz); free((void *)frames[f].pairs); }
  free(frames);
  *out_pcm = out;
  *out_n = n_all;
  if (sample_rate) *sample_rate = sr;
  if (channels) *channels = chs;
  return 0;
}

void glo_free(void *p) { free(p); }

/* ---------------------------------------------------------------- test signals
 * Restatement of the generators in the reference's tests/utils.rs:5-114 (inputs only). */

static uint64_t gen_total(uint32_t sr, float dur) { return (uint64_t)((float)sr * dur); }

uint64_t glo_gen_tone(int kind, float f0, float f1, uint32_t sr, uint16_t ch, float dur,
                      float *out) {
  /* kind 0 sine (:5-22), 1 square (:25-43), 2 sawtooth (:46-64), 3 sweep f0->f1 (:67-86) */
  uint64_t total = gen_total(sr, dur);
  if (!out) return total * ch;
  for (uint64_t i = 0; i < total; ++i) {
    float t = (float)i / (float)sr;
    float s;
    if (kind == 0) {
      s = sinf(((2.0f * kPi) * f0) * t) * 0.5f;
    } else if (kind == 1) {
      float phase = ((2.0f * kPi) * f0) * t;
      s = sinf(phase) >= 0.0f ? 0.3f : -0.3f;
    } else if (kind == 2) {
      float phase = fmodf(((2.0f * kPi) * f0) * t, 2.0f * kPi);
      s = ((phase / kPi) - 1.0f) * 0.3f;
    } else {
      float progress = t / dur;
      float freq = f0 + (f1 - f0) * progress;
      s = sinf(((2.0f * kPi) * freq) * t) * 0.3f;
    }
    for (uint16_t c = 0; c < ch; ++c) out[i * ch + c] = s;
  }
  return total * ch;
}

uint64_t glo_gen_noise(uint32_t sr, uint16_t ch, float dur, uint64_t seed, float *out) {
  /* tests/utils.rs:89-114: u64 LCG, (state as f32 / u64::MAX as f32 - 0.5) * 0.6 */
  uint64_t total = gen_total(sr, dur);
  if (!out) return total * ch;
  uint64_t state = seed;
  const float denom = 18446744073709551615.0f; /* u64::MAX as f32 == 2^64 */
  for (uint64_t i = 0; i < total * ch; ++i) {
    state = state * 1664525ull + 1013904223ull;
    float normalized = (float)state / denom;
    out[i] = (normalized - 0.5f) * 0.6f;
  }
  return total * ch;
}
