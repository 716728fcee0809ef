/*
 * glc_oracle.h — CPU restatement of the reference codec hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X path.  It restates, in plain C, the algorithm of
 * /root/reference/src/codec.rs (v0.5.0).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product library (libglc_hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference's own tests (the tests directory) hold no golden vectors or byte-level
 * expectations for this path, and no Rust toolchain exists in the build image, so the oracle is
 * pinned only by (a) line-by-line reading of src/codec.rs, (b) an independent numpy float32
 * restatement (oracle/glc_oracle_np.py) that must agree bit-for-bit, and (c) the reference's
 * property tests (lengths, SNR bounds, sparsity, size ratios) restated in tests/.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  Contraction MUST stay
 * off: rustc never fuses `s += a * b`.
 */
#ifndef GLC_ORACLE_H
#define GLC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLO_FRAME_SIZE 2048u /* src/codec.rs:15 */
#define GLO_HOP_SIZE 1024u   /* src/codec.rs:16 */
#define GLO_MAX_BANDS 51u    /* src/codec.rs:154,181: at most 50 edges + final n */

/* src/codec.rs:326-356  MdctTables::new — T[k*2048+i], window[2048], norm. */
void glo_tables(float *table, float *window, float *norm);
/* src/codec.rs:359-374 */
void glo_mdct_block(const float *table, float norm, const float *block, float *out);
/* src/codec.rs:377-390 */
void glo_imdct_block(const float *table, float norm, const float *coeffs, float *out);
/* src/codec.rs:102-183  PerceptualWeights::new: weights[1024], edges[<=51]; returns #edges. */
uint32_t glo_perceptual(uint32_t sample_rate, float *weights, uint32_t *edges);
/* src/codec.rs:188-240 */
void glo_thresholds(const float *coeffs, const float *weights, const uint32_t *edges,
                    uint32_t n_edges, float *thr);
/* src/codec.rs:270-311 (incl. the dead bits helper :243-267); returns nnz. */
uint32_t glo_compress(const float *coeffs, float scale, const float *thr, uint16_t *idx,
                      int16_t *q);

/* Optional taps for parity tests; any pointer may be NULL.
 *   coeffs  [n_frames*ch*1024] f32   MDCT output (after norm), row m = frame*ch + c
 *   scales  [n_frames*ch]      f32   max|c| clamped to 1e-10 (computed even for raw frames)
 *   nnz     [n_frames*ch]      u32   kept coefficients (computed even for raw frames)
 *   is_raw  [n_frames]         u8    per-frame fallback decision
 *   dense_q [n_frames*ch*1024] i16   quantised value per bin (0 = dropped)
 */
typedef struct glo_taps {
  float *coeffs;
  float *scales;
  uint32_t *nnz;
  uint8_t *is_raw;
  int16_t *dense_q;
} glo_taps;

/* Number of frames Encoder::encode produces (src/codec.rs:433-455); 0 if the reference
 * would panic (Q6: <= 512 samples per channel, channels == 0, ragged channels that
 * under-run the slice at :474). */
uint64_t glo_num_frames(uint64_t n_samples, uint16_t channels);

/* src/codec.rs:421-565 + :774-779: encode interleaved PCM and bincode-serialise.
 * Returns 0 and a malloc'd byte buffer (release with glo_free), or -1 where the reference
 * would panic.  n_threads <= 0 → all online cores (rayon's default, :462). */
int glo_encode(uint32_t sample_rate, const float *pcm, uint64_t n_samples, uint16_t channels,
               int n_threads, uint8_t **out_bytes, uint64_t *out_len, const glo_taps *taps);

/* Frames [f0, f1) of a stream of n_samples interleaved samples computed from one shard of its PCM
 * (per-channel samples [t0, t0+t_count), interleaved): the body of the rayon loop :462-541 per
 * frame, written as the device path's fixed-size frame records (include/glc.h).  Stream samples
 * outside the shard are poisoned with NaN (a missing halo cannot go unnoticed).  taps rows are
 * relative to f0.  Returns 0, or -1 for a bad range / a stream the reference panics on. */
int glo_encode_range_records(uint32_t sample_rate, const float *shard, uint64_t t0, uint64_t t_count,
                             uint64_t n_samples, uint16_t channels, uint64_t f0, uint64_t f1,
                             int n_threads, uint8_t *records, const glo_taps *taps);

/* src/codec.rs:781-786 + :744-768 (+ decode_streaming :595-741): parse .glc bytes and decode.
 * Returns 0 and malloc'd interleaved f32 (glo_free), -1 on malformed input. */
int glo_decode(const uint8_t *bytes, uint64_t len, int n_threads, float **out_pcm,
               uint64_t *out_n, uint32_t *sample_rate, uint16_t *channels);

/* Time only the transform+quantiser of `n_frames` frames starting at frame `f0` of the given
 * input (no serialisation): the cpu_baseline leg of bench.py.  Returns seconds (wall). */
double glo_time_encode_frames(uint32_t sample_rate, const float *pcm, uint64_t n_samples,
                              uint16_t channels, uint64_t f0, uint64_t n_frames, int n_threads);

/* Test-signal generators restating the reference's tests/utils.rs:5-114 (inputs only).
 * out == NULL returns the sample count.  kind: 0 sine, 1 square, 2 sawtooth, 3 sweep f0->f1. */
uint64_t glo_gen_tone(int kind, float f0, float f1, uint32_t sr, uint16_t ch, float dur,
                      float *out);
uint64_t glo_gen_noise(uint32_t sr, uint16_t ch, float dur, uint64_t seed, float *out);

void glo_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
