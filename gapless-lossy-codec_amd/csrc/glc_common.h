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

void set_global_error(const std::string &msg);

}  // namespace glc

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
};
