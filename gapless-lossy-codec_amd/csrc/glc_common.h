// glc_common.h — shared host-side definitions for the MI355X codec hot path.
// Constants mirror /root/reference/src/codec.rs:15-29.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/glc.h"

namespace glc {

constexpr uint32_t kFrame = GLC_FRAME_SIZE;  // FRAME_SIZE, src/codec.rs:15
constexpr uint32_t kHop = GLC_HOP_SIZE;      // HOP_SIZE,   src/codec.rs:16
constexpr uint32_t kMaxEdges = 51;           // <= 50 band edges + final n, src/codec.rs:154,181
constexpr float kNoiseFloorDb = -48.0f;      // src/codec.rs:22
constexpr float kQuality = 0.7f;             // src/codec.rs:23
constexpr float kCompressionThreshold = 0.85f;  // src/codec.rs:29

// Host tables: MdctTables (src/codec.rs:316-356) + PerceptualWeights (:93-184) + the
// per-bin / per-band constants the device quantiser needs (all derived with the same f32
// expressions the reference evaluates per frame, hoisted because they are frame-invariant).
struct HostTables {
  std::vector<float> cos_table;    // [1024][2048], row k  (reference layout)
  std::vector<float> cos_table_t;  // [2048][1024], row i  (device layout for the forward MDCT)
  std::vector<float> window;       // [2048]
  float norm = 0.f;
  std::vector<float> weights;      // [1024]
  std::vector<uint32_t> edges;     // band edges
  // derived, frame-invariant pieces of compute_masking_thresholds (:218-228):
  std::vector<float> band_pf;      // per band: (1-q).max(.01) and 1/avg_w.max(.1) stay separate
  std::vector<float> band_len;     // (end-start) as f32
  std::vector<float> indiv;        // per bin: 1.0 / weights[i].max(0.1)
  std::vector<uint16_t> band_of;   // per bin: band index
  float cf = 0.f;                  // (1.0 - QUALITY_FACTOR).max(0.01)
  float noise_floor = 0.f;         // 10f32.powf(NOISE_FLOOR_DB / 20.0)
  uint32_t sample_rate = 0;
};

void build_host_tables(uint32_t sample_rate, HostTables &t);

// Padding arithmetic of Encoder::encode, src/codec.rs:433-455.
glc_plan plan_encode(uint64_t n_samples, uint16_t channels);

// Fixed-size device record (see include/glc.h glc_record_bytes).
inline uint64_t record_header_bytes(uint32_t ch) { return ((8ull + 8ull * ch) + 15ull) & ~15ull; }
inline uint64_t record_bytes(uint32_t ch) {
  return record_header_bytes(ch) + 2ull * kFrame * ch;
}

// Compact ("bitstream payload") form of a contiguous frame range, one self-describing blob:
//   header (64 B) | is_raw u8[n_frames] | scale f32[M] | cnt u32[M] | pairs u32[n_pairs] |
//   raw i16[n_raw_rows][2048]          M = n_frames * channels, every section 64-byte aligned
// cnt[m] is the sparse-list length of row m (0 for the rows of raw frames); pairs are the lists
// back to back in row order, (u16 idx | i16 q << 16) ascending in idx; raw holds the 2048-sample
// planes of the rows of raw frames in row order (channel-planar per frame, quirk Q1).  This is what
// crosses PCIe after an encode and what a multi-GPU job gathers: ~1/8 of the fixed-size records.
constexpr uint32_t kCompactMagic = 0x42434C47u;  // "GLCB"
struct CompactHeader {
  uint32_t magic;
  uint32_t channels;
  uint64_t n_frames;
  uint64_t n_pairs;
  uint64_t n_raw_rows;
  uint64_t bytes;  // whole blob, header included
  uint64_t reserved[3];
};
static_assert(sizeof(CompactHeader) == 64, "compact header is 64 bytes");
inline uint64_t align64(uint64_t v) { return (v + 63ull) & ~63ull; }
struct CompactLayout {
  uint64_t o_israw, o_scale, o_cnt, o_pairs;  // byte offsets of the fixed sections
  uint64_t bound;                             // worst-case blob size for this range
};
inline CompactLayout compact_layout(uint32_t ch, uint64_t n_frames) {
  const uint64_t M = n_frames * ch;
  CompactLayout l;
  l.o_israw = sizeof(CompactHeader);
  l.o_scale = l.o_israw + align64(n_frames);
  l.o_cnt = l.o_scale + align64(4 * M);
  l.o_pairs = l.o_cnt + align64(4 * M);
  // a row is either compressed (<= 1024 pairs = 4096 B) or raw (2048 i16 = 4096 B)
  l.bound = l.o_pairs + 4096ull * M + 64ull;
  return l;
}
inline uint64_t compact_raw_offset(const CompactLayout &l, uint64_t n_pairs) { return align64(l.o_pairs + 4 * n_pairs); }

void set_global_error(const std::string &msg);

// Host assembly of EncodedAudio from compact blobs in frame order (glc_frames_from_compact).
// `trusted`: the blobs were produced by this process's own pack kernels (lists known canonical).
int index_compact_meta(glc_frames *F, uint32_t ch, const CompactHeader &h, const uint8_t *meta, uint64_t f_at,
                       uint64_t p_at, uint64_t r_at, bool trusted, bool *canonical);
int frames_from_compact(uint32_t sample_rate, uint64_t n_samples, uint16_t channels, const void *const *blobs,
                        const uint64_t *blob_bytes, uint32_t n_blobs, bool trusted, glc_frames **out);

}  // namespace glc

namespace glc {
uint64_t next_frames_uid();  // never 0
}

// EncodedAudio (src/codec.rs:31-69) in a flat, general form: every Vec of the schema keeps its
// own length so that any well-formed bincode stream round-trips byte-for-byte.
struct glc_frames {
  uint32_t sample_rate = 0;
  uint16_t channels = 0;
  uint64_t total_samples = 0;
  uint32_t encoder_delay = 0;
  uint32_t padding = 0;
  uint64_t original_length = 0;
  uint64_t n_frames = 0;
  // per frame
  std::vector<uint64_t> list_begin;   // [n_frames+1] -> index into list_off (sparse lists)
  std::vector<uint64_t> scale_begin;  // [n_frames+1] -> index into scales
  std::vector<uint8_t> raw_tag;       // [n_frames]   Option tag
  std::vector<uint64_t> raw_begin;    // [n_frames+1] -> index into raw
  // pools
  std::vector<uint64_t> list_off;  // [n_lists+1] -> index into pairs
  std::vector<uint32_t> pairs;     // (u16 idx) | (u16 q << 16), stream order
  std::vector<float> scales;
  std::vector<int16_t> raw;
  // true when every sparse list is known to be strictly ascending with idx < 1024 (streams
  // assembled from this library's own records); streams read from bytes are checked at decode
  bool lists_canonical = false;
  // process-unique identity of this (immutable) object: lets a context recognise a stream whose
  // sparse rows it already holds on the device (glc_decode_* called again on the same frames)
  uint64_t uid = glc::next_frames_uid();
};
