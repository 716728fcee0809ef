/*
 * glc.h — C ABI of the MI355X-native MDCT / quantiser hot path of the gapless lossy codec.
 *
 * Drop-in boundary for the reference crate `gapless_lossy_codec` (v0.5.0).  The reference has no
 * FFI seam today: `Encoder` / `Decoder` are concrete Rust structs in src/codec.rs.  Each entry
 * point below names the reference item it replaces (file:line into the reference tree); a thin
 * Rust shim keeps the Rust signatures and forwards here.
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types; all integers little-endian host order
 *   - return 0 (GLC_OK) on success, a negative glc_status otherwise; the text of the last
 *     failure is available from glc_last_error()
 *   - a glc_ctx is NOT thread-safe (≙ `&mut self`, src/codec.rs:421,744); distinct contexts may
 *     be used from distinct threads
 *   - inputs the reference would panic on (channels == 0, <= 512 samples per
 *     channel, ragged channel lengths that under-run a frame) return GLC_EINVAL instead
 *   - there is NO CPU fallback: every compute entry point fails with GLC_ENODEV / GLC_EHIP when
 *     no gfx950 device is usable
 */
#ifndef GLC_H
#define GLC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLC_FRAME_SIZE 2048u /* src/codec.rs:15 FRAME_SIZE */
#define GLC_HOP_SIZE 1024u   /* src/codec.rs:16 HOP_SIZE  */
#define GLC_FRAMES_PER_CHUNK 500u /* src/codec.rs:18 */

typedef enum glc_status {
  GLC_OK = 0,
  GLC_EINVAL = -1,  /* bad argument / input the reference panics on */
  GLC_EHIP = -2,    /* HIP runtime or kernel failure */
  GLC_ENOMEM = -3,  /* host or device allocation failed */
  GLC_EFORMAT = -4, /* malformed .glc byte stream (bincode error in load_encoded) */
  GLC_ENODEV = -5,  /* no usable gfx950 device */
  GLC_EIO = -6      /* file I/O error (save_encoded / load_encoded) */
} glc_status;

/* ≙ Encoder / Decoder (src/codec.rs:396-402, :571-577): MDCT table, window, perceptual model,
 * device workspaces and the HIP stream the kernels run on. */
typedef struct glc_ctx glc_ctx;
/* ≙ EncodedAudio (src/codec.rs:31-37), owned by the library. */
typedef struct glc_frames glc_frames;

/* ≙ AudioHeader + GaplessInfo (src/codec.rs:39-53) + summary counts. */
typedef struct glc_info {
  uint32_t sample_rate;
  uint16_t channels;
  uint16_t reserved;
  uint64_t total_samples;
  uint32_t encoder_delay;
  uint32_t padding;
  uint64_t original_length;
  uint64_t n_frames;
  uint64_t n_raw_frames; /* frames carrying raw_pcm: Some(..) */
  uint64_t total_nnz;    /* sum of sparse list lengths */
} glc_info;

/* Result of the padding arithmetic of Encoder::encode (src/codec.rs:433-455, :543-547). */
typedef struct glc_plan {
  uint64_t n_frames;     /* 0 if the reference would panic */
  uint64_t padded_len;   /* per-channel length of padded[0] */
  uint64_t per_channel;  /* per_chan[0].len() */
  uint32_t encoder_delay;
  uint32_t padding;
} glc_plan;

/* ---- context -------------------------------------------------------------------------- */

/* Encoder::new(sample_rate) src/codec.rs:406-418 and Decoder::new(_, sample_rate) :581-592.
 * Builds MdctTables (:326-356) and PerceptualWeights (:102-183) on the host with the system
 * libm, uploads them to HIP device `device`, creates the stream and workspaces. */
int glc_ctx_create(int device, uint32_t sample_rate, glc_ctx **out);
void glc_ctx_destroy(glc_ctx *ctx);
/* Message of the last failure on `ctx` (or of the last context-less failure if ctx == NULL). */
const char *glc_last_error(const glc_ctx *ctx);
/* The HIP stream (hipStream_t) every kernel of this context is launched on.  It is a stream of the context's own,
 * created non-blocking: it does NOT wait for work a caller has queued elsewhere (not even on the legacy default
 * stream).  A device buffer that another stream is still filling or reading must be synchronised with by the
 * caller - or the context put on that stream with glc_ctx_set_stream - before an entry point is given it. */
void *glc_ctx_stream(glc_ctx *ctx);
int glc_ctx_device(const glc_ctx *ctx);
/* Run this context's kernels on a caller-owned hipStream_t instead (e.g. the stream a host
 * framework times with its own events).  The caller keeps ownership; NULL restores the
 * context's private stream. */
int glc_ctx_set_stream(glc_ctx *ctx, void *hip_stream);
/* Block until all work queued on the context's stream has finished. */
int glc_ctx_synchronize(glc_ctx *ctx);
/* Device-side stopwatch on the context's stream: `begin` records a hipEvent, `end` records a
 * second one, waits for it and returns the elapsed milliseconds between the two — the time the
 * kernels queued in between spent on that stream. */
int glc_ctx_timer_begin(glc_ctx *ctx);
int glc_ctx_timer_end(glc_ctx *ctx, float *elapsed_ms);

/* ---- encode --------------------------------------------------------------------------- */

/* Padding / frame-count arithmetic only (host, no device): src/codec.rs:433-455. */
int glc_plan_encode(uint64_t n_samples, uint16_t channels, glc_plan *out);

/* Encoder::encode(&mut self, samples: &[f32], channels: u16) -> Result<EncodedAudio>
 * src/codec.rs:421-565.  `pcm` is interleaved host memory, borrowed for the call. */
int glc_encode(glc_ctx *ctx, const float *pcm, uint64_t n_samples, uint16_t channels,
               glc_frames **out);

/* Fixed-size per-frame record the device path emits (one per frame):
 *   u32 is_raw, u32 reserved, then per channel {f32 scale, u32 nnz}, padded to 16 B,
 *   then int16 payload[channels][2048]: compressed frames hold the dense quantised row in
 *   payload[c][0..1023]; raw frames hold the channel-planar windowed i16 block (Q1). */
uint64_t glc_record_bytes(uint16_t channels);

/* Device-resident frame-range encode: the body of the rayon loop at src/codec.rs:462-541 for
 * frames [frame_begin, frame_end) of a stream of `n_samples` interleaved samples.
 *   d_pcm      device pointer to interleaved f32 PCM covering per-channel sample indices
 *              [t0, t0 + t_count) of the stream (a shard with its halo, or the whole stream
 *              with t0 = 0); samples outside the stream are the encoder's zero padding
 *   d_records  device buffer of (frame_end - frame_begin) * glc_record_bytes(channels) bytes
 *   d_coeffs   optional device buffer [(frame_end-frame_begin)*channels][1024] f32 receiving
 *              the MDCT coefficients (parity tap); NULL to use the context workspace
 * Work is queued on glc_ctx_stream(ctx) and NOT synchronised. */
int glc_encode_range_device(glc_ctx *ctx, const float *d_pcm, uint64_t t0, uint64_t t_count,
                            uint64_t n_samples, uint16_t channels, uint64_t frame_begin,
                            uint64_t frame_end, void *d_records, float *d_coeffs);

/* The transform alone: window (src/codec.rs:476-481) + MdctTables::mdct_block (:359-374) for every
 * frame-channel of [frame_begin, frame_end) -> d_coeffs[(frame-frame_begin)*channels + c][1024].
 * Same arguments as glc_encode_range_device; used for kernel-level timing and the parity tap. */
int glc_mdct_forward_device(glc_ctx *ctx, const float *d_pcm, uint64_t t0, uint64_t t_count,
                            uint64_t n_samples, uint16_t channels, uint64_t frame_begin,
                            uint64_t frame_end, float *d_coeffs);

/* Host assembly of EncodedAudio from `n_frames` consecutive records (all shards concatenated in
 * frame order): builds the sparse (u16, i16) lists in ascending k (src/codec.rs:303-306) and the
 * header / gapless info (:543-564). */
int glc_frames_from_records(uint32_t sample_rate, uint64_t n_samples, uint16_t channels,
                            const void *records, uint64_t n_frames, glc_frames **out);

/* Same result as glc_frames_from_records, from records still on the device (this context's
 * device; e.g. right after glc_encode_range_device, or on the gather root): the sparse lists are
 * compacted on the device (scan + ballot pack, ascending k) and only the bitstream's payload —
 * (u16, i16) pairs, scale factors, raw planes of raw frames — crosses to the host.  Synchronises. */
int glc_frames_from_device_records(glc_ctx *ctx, const void *d_records, uint64_t n_frames,
                                   uint64_t n_samples, uint16_t channels, glc_frames **out);

/* Compact form of a contiguous frame range: one self-describing blob holding exactly the
 * bitstream's payload for those frames (per-frame raw flags, per-row scale factor and list length,
 * the (u16 index, i16 value) lists of src/codec.rs:303-306 back to back, raw_pcm planes of raw
 * frames) - about 1/8 of the fixed-size records on tonal material.  It is what crosses PCIe after an
 * encode and what a multi-GPU job gathers to its root: every rank compacts its own frame
 * range, the blobs travel (RCCL send/recv or gather, sizes from glc_compact_info.bytes), and the root
 * assembles EncodedAudio from the blobs in frame order.  Layout: DESIGN.md section 3. */
typedef struct glc_compact_info {
  uint64_t n_frames;
  uint64_t n_pairs;    /* total sparse-list entries */
  uint64_t n_raw_rows; /* frame-channels that belong to raw frames */
  uint64_t bytes;      /* size of the blob (what has to travel) */
} glc_compact_info;
/* Capacity a blob buffer needs for any `n_frames` frames of `channels` channels (worst case). */
uint64_t glc_compact_bound(uint16_t channels, uint64_t n_frames);
/* Device-side compaction (scan + ballot pack, ascending k) of `n_frames` records at d_records into
 * the device buffer d_blob (cap >= glc_compact_bound).  Runs on the context's stream and
 * synchronises it (the sizes come back through `info`).  n_frames may be 0 (an empty shard). */
int glc_compact_device_records(glc_ctx *ctx, const void *d_records, uint64_t n_frames, uint16_t channels,
                               void *d_blob, uint64_t cap, glc_compact_info *info);
/* Host twin of the same packing, for records that are already in host memory. */
int glc_compact_records(const void *records, uint64_t n_frames, uint16_t channels, void *blob, uint64_t cap,
                        glc_compact_info *info);
/* Host assembly of EncodedAudio (header + gapless info as src/codec.rs:543-564) from `n_blobs`
 * host-resident blobs (each 8-byte aligned) that together cover the stream's frames in order.  Every
 * count in a blob is validated against its size; GLC_EFORMAT on inconsistency. */
int glc_frames_from_compact(uint32_t sample_rate, uint64_t n_samples, uint16_t channels,
                            const void *const *blobs, const uint64_t *blob_bytes, uint32_t n_blobs,
                            glc_frames **out);

/* ---- decode --------------------------------------------------------------------------- */

/* Length Decoder::decode will return: min(original_length, (n_frames+1)*1024*ch - delay). */
uint64_t glc_decoded_len(const glc_frames *in);

/* Decoder::decode(&mut self, &EncodedAudio, _) -> Result<Vec<f32>>  src/codec.rs:744-768
 * (including decode_streaming :595-741, overlap-add and the gapless trim). */
int glc_decode(glc_ctx *ctx, const glc_frames *in, float *pcm_out, uint64_t cap,
               uint64_t *n_out);

/* Device-resident decode: the whole un-trimmed stream decode_streaming emits ((n_frames + 1) *
 * 1024 * channels samples, src/codec.rs:688-732) is written to the device buffer d_all (capacity
 * cap_all samples); *start / *n_out give the gapless-trimmed window Decoder::decode would return
 * (:756-765) inside it.  Work is queued on glc_ctx_stream(ctx) and NOT synchronised, except for
 * the one-off upload of the sparse rows: a context keeps the rows of the last stream it decoded on
 * the device, so decoding the same glc_frames object again (any glc_decode_* entry point) uploads
 * nothing. */
int glc_decode_device(glc_ctx *ctx, const glc_frames *in, float *d_all, uint64_t cap_all,
                      uint64_t *start, uint64_t *n_out);

/* One shard of the same decode (multi-GPU): hops [hop_begin, hop_end) of the
 * un-trimmed stream, hop_end <= n_frames + 1, written to d_out (hop hop_begin at d_out[0], capacity
 * cap samples).  Hop h is the second half of frame h-1 plus the first half of frame h
 * (src/codec.rs:688-705), hop n_frames the bare overlap tail (:722-729); a range that does not
 * start at 0 recomputes frame hop_begin-1 as its halo, so disjoint ranges decoded on different
 * GPUs concatenate to exactly the whole-stream output.  Queued on the context's stream. */
int glc_decode_range_device(glc_ctx *ctx, const glc_frames *in, uint64_t hop_begin,
                            uint64_t hop_end, float *d_out, uint64_t cap);

/* The inverse transform alone: dequantisation (src/codec.rs:651-665) + MdctTables::imdct_block
 * (:377-390) + window (:672-675) - or the raw-frame path (:626-644) - for every frame-channel of
 * [frame_begin, frame_end) -> d_blocks[(frame-frame_begin)*channels + c][2048] f32, before any
 * overlap-add.  Used for kernel-level timing and as the parity tap of the decode side (the
 * counterpart of glc_mdct_forward_device).  Queued on the context's stream, not synchronised. */
int glc_imdct_device(glc_ctx *ctx, const glc_frames *in, uint64_t frame_begin, uint64_t frame_end,
                     float *d_blocks);

/* Decoder::decode_streaming src/codec.rs:595-741: un-trimmed output delivered in chunks of at
 * least FRAMES_PER_CHUNK*1024*channels samples (AudioChunk, :81-85).  `begin` decodes on the
 * device; `next` copies the next chunk (returns its size through n_out, sets *is_last on the
 * final chunk, which carries the overlap tail :722-732). */
int glc_decode_stream_begin(glc_ctx *ctx, const glc_frames *in);
int glc_decode_stream_next(glc_ctx *ctx, float *chunk, uint64_t cap, uint64_t *n_out,
                           int *is_last);

/* ---- container (.glc = bincode 1.x of EncodedAudio) ------------------------------------- */

/* save_encoded / load_encoded src/codec.rs:774-786. */
uint64_t glc_serialized_size(const glc_frames *f);
int glc_serialize(const glc_frames *f, uint8_t *buf, uint64_t cap, uint64_t *written);
int glc_deserialize(const uint8_t *buf, uint64_t len, glc_frames **out);
int glc_save(const glc_frames *f, const char *path);
int glc_load(const char *path, glc_frames **out);
void glc_frames_free(glc_frames *f);

/* ---- EncodedAudio accessors ----------------------------------------------------------- */

int glc_frames_info(const glc_frames *f, glc_info *out);
/* EncodedFrame.raw_pcm.is_some() */
int glc_frame_is_raw(const glc_frames *f, uint64_t frame);
/* EncodedFrame.sparse_coeffs_per_channel[channel]: copies up to cap pairs, returns the list
 * length through n (idx/q may be NULL to query). */
int glc_frame_sparse(const glc_frames *f, uint64_t frame, uint32_t channel, uint16_t *idx,
                     int16_t *q, uint32_t cap, uint32_t *n);
/* EncodedFrame.scale_factors[channel] */
int glc_frame_scale(const glc_frames *f, uint64_t frame, uint32_t channel, float *scale);
/* EncodedFrame.raw_pcm: copies up to cap samples, returns the length through n. */
int glc_frame_raw(const glc_frames *f, uint64_t frame, int16_t *pcm, uint64_t cap, uint64_t *n);

/* ---- EncodedAudio as flat arrays: the structured bridge to the reference's nested Vecs ------- */

/* EncodedAudio (src/codec.rs:31-69) as the flat pools a glc_frames keeps, so that a host fills or
 * reads `Vec<EncodedFrame>` with one slice copy per list and no byte stream in between.  Every Vec of
 * the schema keeps its own length (any well-formed .glc round-trips):
 *   frame f: sparse_coeffs_per_channel = lists  list_begin[f] .. list_begin[f+1]   (indices into list_off)
 *            list l                    = pairs  list_off[l]   .. list_off[l+1]     ((u16 index, i16 value),
 *                                        4 bytes each: index in the low half, the layout of src/codec.rs:303-306
 *                                        written little-endian - what bincode emits for Vec<(u16, i16)>)
 *            scale_factors             = scales scale_begin[f] .. scale_begin[f+1]
 *            raw_pcm                   = None if raw_tag[f] == 0, else raw raw_begin[f] .. raw_begin[f+1]
 * As a VIEW (glc_frames_get_view) the pointers are borrowed from the glc_frames and stay valid until it
 * is freed.  As PARTS (glc_frames_from_parts) they are the caller's arrays, copied by the call. */
typedef struct glc_frames_view {
  uint32_t sample_rate;     /* AudioHeader, src/codec.rs:39-46 */
  uint16_t channels;
  uint16_t reserved;
  uint64_t total_samples;
  uint32_t encoder_delay;   /* GaplessInfo, src/codec.rs:48-53 */
  uint32_t padding;
  uint64_t original_length;
  uint64_t n_frames;
  uint64_t n_lists;         /* all sparse lists of the stream */
  uint64_t n_pairs;
  uint64_t n_scales;
  uint64_t n_raw;           /* i16 samples in the raw pool */
  const uint64_t *list_begin;   /* [n_frames + 1] */
  const uint64_t *list_off;     /* [n_lists + 1]  */
  const uint32_t *pairs;        /* [n_pairs]      */
  const uint64_t *scale_begin;  /* [n_frames + 1] */
  const float *scales;          /* [n_scales]     */
  const uint8_t *raw_tag;       /* [n_frames]     */
  const uint64_t *raw_begin;    /* [n_frames + 1] */
  const int16_t *raw;           /* [n_raw]        */
} glc_frames_view;

/* Borrow the pools of `f` (≙ reading the EncodedAudio that Encoder::encode returned, src/codec.rs:421). */
int glc_frames_get_view(const glc_frames *f, glc_frames_view *out);

/* Build an EncodedAudio from flat arrays (≙ the `&EncodedAudio` Decoder::decode takes, src/codec.rs:744).
 * Every offset is validated (monotonic, inside its pool); GLC_EFORMAT otherwise.  `stream_id`: 0, or a
 * caller-chosen identity < 2^63 of this stream's CONTENT - the caller promises that two objects built
 * with the same non-zero id hold the same stream.  A context recognises the id of the stream whose
 * sparse rows (and inverse-transform plan) it still holds on the device and decodes it again without
 * preparing or uploading anything (glc_ctx_resident_stream). */
int glc_frames_from_parts(const glc_frames_view *parts, uint64_t stream_id, glc_frames **out);

/* The same constructor for a host whose lists live in separate allocations (the reference's
 * Vec<Vec<(u16, i16)>>): pointer + length per list, per-frame scale and raw vectors by pointer; the
 * payload is copied once, straight into the pools.  list l of frame f is lists[Σ_{g<f} lists_per_frame[g] + l]. */
typedef struct glc_frames_gather {
  uint32_t sample_rate;
  uint16_t channels;
  uint16_t reserved;
  uint64_t total_samples;
  uint32_t encoder_delay;
  uint32_t padding;
  uint64_t original_length;
  uint64_t n_frames;
  const uint32_t *lists_per_frame;   /* [n_frames]                sparse_coeffs_per_channel.len() */
  const void *const *list_ptr;       /* [Σ lists_per_frame]       pointer to the list's 4-byte pairs */
  const uint32_t *list_len;          /* [Σ lists_per_frame]       pairs in the list */
  const uint32_t *scales_per_frame;  /* [n_frames]                scale_factors.len() */
  const float *const *scale_ptr;     /* [n_frames] */
  const int16_t *const *raw_ptr;     /* [n_frames]                NULL = raw_pcm: None */
  const uint64_t *raw_len;           /* [n_frames] */
} glc_frames_gather;
int glc_frames_from_gather(const glc_frames_gather *g, uint64_t stream_id, glc_frames **out);

/* Identity of a glc_frames: the stream_id it was built with (glc_frames_from_parts / _gather), or a
 * process-unique number >= 2^63 for objects the library built itself. */
uint64_t glc_frames_stream_id(const glc_frames *f);
/* Identity of the stream whose sparse rows `ctx` holds on the device (0: none). */
uint64_t glc_ctx_resident_stream(const glc_ctx *ctx);
/* Decoder::decode of the resident stream, without a glc_frames: for a host that has recognised the
 * stream by its id and so need not flatten its nested vectors again.  GLC_EINVAL if `stream_id` is not
 * the resident stream (then build the glc_frames and call glc_decode). */
int glc_decode_resident(glc_ctx *ctx, uint64_t stream_id, float *pcm_out, uint64_t cap, uint64_t *n_out);

/* Encoder::encode with a hook: `fn(user, view, frame_begin, frame_end)` is called each time the frames
 * [frame_begin, frame_end) have arrived on the host (ranges ascend and tile [0, n_frames)), while the
 * device still works on later frames - where a host builds its nested EncodedFrame vectors, hidden
 * behind the rest of the encode instead of after it.  The calls are made on the calling thread, in the
 * time it would otherwise spend blocked on the device (ranges of a few dozen frames while it polls,
 * the remainder at the end).  `view` covers frames [0, frame_end) and is valid only during the call.
 * A non-zero return aborts the encode (GLC_EINVAL).  `out` may be NULL when the hook has taken
 * everything it needs. */
typedef int (*glc_frames_hook)(void *user, const glc_frames_view *view, uint64_t frame_begin, uint64_t frame_end);
int glc_encode_hooked(glc_ctx *ctx, const float *pcm, uint64_t n_samples, uint16_t channels,
                      glc_frames_hook fn, void *user, glc_frames **out);

/* ---- WAV file I/O twin (src/audio.rs; host only, no device) ---------------------------------- */

/* load_wav src/audio.rs:39-64: RIFF/WAVE PCM (8/16/24/32-bit integer -> s / 2^(bits-1), 8-bit is
 * unsigned in the file) and 32-bit IEEE float (passed through), incl. WAVE_FORMAT_EXTENSIBLE.
 * Returns a malloc'd interleaved buffer; release it with glc_free. */
int glc_wav_load(const char *path, float **samples, uint64_t *n_samples, uint32_t *sample_rate,
                 uint16_t *channels);
/* export_to_wav src/audio.rs:100-132: 16-bit PCM, (s * 32767).clamp(-32768, 32767) as i16. */
int glc_wav_save16(const char *path, const float *samples, uint64_t n_samples, uint32_t sample_rate,
                   uint16_t channels);
void glc_free(void *p);

/* ---- FLAC file I/O twin (src/flac.rs, src/audio.rs:68-96; host only, no device) ----------- */

/* encode_flac_with_level src/flac.rs:947-1053 (encode_flac :1056-1063 is level 5): the reference's
 * own encoder, byte for byte - 16-bit, (s * 32767).clamp(-32768, 32767) as i16, block 1152
 * (levels 0-2) or 4096, verbatim (level 0) or fixed predictor of order 1/2/3/3/4.. by level,
 * partitioned Rice with 4-bit parameters, independent channels, STREAMINFO with the MD5 of the
 * samples.  GLC_EINVAL for < 16 samples per channel, level > 8 (the reference's two Err cases)
 * and channels == 0 (a division by zero there).  *out is malloc'd; release it with glc_free. */
int glc_flac_encode(const float *samples, uint64_t n_samples, uint32_t sample_rate,
                    uint16_t channels, uint8_t level, uint8_t **out, uint64_t *out_len);
/* export_to_flac_with_level src/flac.rs:1066-1077 (export_to_flac :1080-1087 is level 5). */
int glc_flac_save(const char *path, const float *samples, uint64_t n_samples, uint32_t sample_rate,
                  uint16_t channels, uint8_t level);
/* load_flac src/audio.rs:68-85 (the reference delegates to the claxon crate): any RFC 9639
 * stream - constant / verbatim / fixed / LPC subframes, left-side / side-right / mid-side stereo,
 * both Rice code books, wasted bits - with both CRCs checked; samples become s / 2^(bits-1).
 * Returns a malloc'd interleaved buffer; release it with glc_free. */
int glc_flac_load(const char *path, float **samples, uint64_t *n_samples, uint32_t *sample_rate,
                  uint16_t *channels);
int glc_flac_decode(const uint8_t *buf, uint64_t len, float **samples, uint64_t *n_samples,
                    uint32_t *sample_rate, uint16_t *channels);

/* ---- tables (for inspection / parity tests) ---------------------------------------------- */

/* Copies of the host tables of a context: MdctTables.cos_table [1024*2048] (row k), window
 * [2048], norm; PerceptualWeights.weights [1024] and critical_bands (<= 51 edges).
 * Any pointer may be NULL. */
int glc_ctx_tables(const glc_ctx *ctx, float *cos_table, float *window, float *norm,
                   float *weights, uint32_t *band_edges, uint32_t *n_edges);

const char *glc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GLC_H */
