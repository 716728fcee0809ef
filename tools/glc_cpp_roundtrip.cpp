// glc_cpp_roundtrip.cpp — exercises include/glc.hpp (the C++ mirror of the reference's Encoder /
// Decoder API) the way tests/integration_tests.rs drives the Rust API: encode interleaved f32 PCM,
// save, load, decode both ways, and leave the artefacts on disk for the caller to compare.
//   glc_cpp_roundtrip <in.f32> <sample_rate> <channels> <out.glc> <out.f32>
// Build: g++ -O2 -std=c++17 -Iinclude tools/glc_cpp_roundtrip.cpp -Lgapless-lossy-codec_amd -lglc_hip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include "glc.hpp"

int main(int argc, char **argv) {
  if (argc != 6) {
    std::fprintf(stderr, "usage: %s in.f32 sample_rate channels out.glc out.f32\n", argv[0]);
    return 2;
  }
  try {
    std::ifstream in(argv[1], std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<float> pcm(raw.size() / sizeof(float));
    std::memcpy(pcm.data(), raw.data(), pcm.size() * sizeof(float));
    const uint32_t sr = static_cast<uint32_t>(std::atoi(argv[2]));
    const uint16_t ch = static_cast<uint16_t>(std::atoi(argv[3]));

    glc::Encoder encoder(sr);
    glc::EncodedAudio encoded = encoder.encode(pcm, ch);
    glc::save_encoded(encoded, argv[4]);

    glc::EncodedAudio loaded = glc::load_encoded(argv[4]);
    glc::Decoder decoder(ch, sr);
    std::vector<float> whole = decoder.decode(loaded);

    std::vector<float> streamed;
    size_t chunks = 0;
    bool saw_last = false;
    decoder.decode_streaming(loaded, [&](glc::AudioChunk &&c) {
      if (saw_last) throw glc::Error(GLC_EINVAL, "chunk after is_last");
      streamed.insert(streamed.end(), c.samples.begin(), c.samples.end());
      saw_last = c.is_last;
      ++chunks;
    });
    // Decoder::decode is the stream with the first encoder_delay values drained and the rest cut
    // to original_length (src/codec.rs:755-765)
    const glc::GaplessInfo gi = loaded.gapless_info();
    if (!saw_last || streamed.size() < gi.encoder_delay + whole.size() || whole.size() != gi.original_length ||
        std::memcmp(streamed.data() + gi.encoder_delay, whole.data(), whole.size() * sizeof(float)) != 0) {
      std::fprintf(stderr, "streaming decode differs from decode\n");
      return 1;
    }
    // structure walk through the accessors (EncodedFrame view)
    uint64_t nnz = 0, raw_frames = 0;
    for (uint64_t f = 0; f < loaded.n_frames(); ++f) {
      const glc::EncodedFrame fr = loaded.frame(f);
      raw_frames += fr.has_raw_pcm;
      for (const auto &l : fr.sparse_coeffs_per_channel) nnz += l.size();
    }
    // the structured bridge through the C++ mirror: frames() == frame(f) for every f; the nested vectors go
    // back in by pointer (from_frames) and serialise to the same bytes; the hooked encode hands out the same
    // frames in ascending contiguous ranges; a stream id makes the second decode a resident one
    {
      const std::vector<glc::EncodedFrame> all = loaded.frames();
      if (all.size() != loaded.n_frames()) return std::fprintf(stderr, "frames(): wrong count\n"), 1;
      for (uint64_t f = 0; f < loaded.n_frames(); ++f) {
        const glc::EncodedFrame one = loaded.frame(f);
        if (one.sparse_coeffs_per_channel != all[f].sparse_coeffs_per_channel || one.scale_factors != all[f].scale_factors ||
            one.has_raw_pcm != all[f].has_raw_pcm || one.raw_pcm != all[f].raw_pcm)
          return std::fprintf(stderr, "frames() differs from frame(%llu)\n", (unsigned long long)f), 1;
      }
      const glc::EncodedAudio again = glc::EncodedAudio::from_frames(loaded.header(), all, loaded.gapless_info(), 4711);
      if (again.to_bytes() != loaded.to_bytes() || again.stream_id() != 4711) return std::fprintf(stderr, "from_frames: bytes differ\n"), 1;
      std::vector<glc::EncodedFrame> hooked;
      uint64_t next = 0;
      bool ordered = true;
      const glc::EncodedAudio e2 = encoder.encode(pcm.data(), pcm.size(), ch, [&](uint64_t f0, std::vector<glc::EncodedFrame> &&part) {
        ordered = ordered && f0 == next;
        next = f0 + part.size();
        for (auto &fr : part) hooked.push_back(std::move(fr));
      });
      if (!ordered || hooked.size() != all.size() || e2.to_bytes() != loaded.to_bytes())
        return std::fprintf(stderr, "hooked encode: ranges or bytes differ\n"), 1;
      for (size_t f = 0; f < all.size(); ++f)
        if (hooked[f].sparse_coeffs_per_channel != all[f].sparse_coeffs_per_channel || hooked[f].scale_factors != all[f].scale_factors ||
            hooked[f].raw_pcm != all[f].raw_pcm)
          return std::fprintf(stderr, "hooked encode: frame %zu differs\n", f), 1;
      glc::Decoder d2(ch, sr);
      const std::vector<float> first = d2.decode(again);
      if (d2.resident_stream() != 4711) return std::fprintf(stderr, "stream id not resident\n"), 1;
      const std::vector<float> second = d2.decode_resident(4711, first.size());
      if (first != whole || second != whole) return std::fprintf(stderr, "bridge decodes differ\n"), 1;
      try {
        encoder.encode(pcm.data(), pcm.size(), ch, [](uint64_t, std::vector<glc::EncodedFrame> &&) { throw std::runtime_error("stop"); });
        return std::fprintf(stderr, "a throwing hook was swallowed\n"), 1;
      } catch (const std::runtime_error &) {
      }
    }
    // src/audio.rs + src/flac.rs twins through the C++ mirror: export the decoded samples to FLAC and
    // WAV next to the output file and read both back (tests/test_export.rs)
    const std::string flac_path = std::string(argv[5]) + ".flac", wav_path = std::string(argv[5]) + ".wav";
    glc::export_to_flac(flac_path, whole, sr, ch);
    glc::export_to_wav(wav_path, whole, sr, ch);
    const glc::LoadedAudio from_flac = glc::load_audio_file_lossless(flac_path);
    const glc::LoadedAudio from_wav = glc::load_audio_file_lossless(wav_path);
    // FLAC frames whole sample-frames only (a ragged tail is hashed but not framed, src/flac.rs:960)
    const size_t framed = whole.size() / ch * ch;
    if (from_flac.sample_rate != sr || from_flac.channels != ch || from_flac.samples.size() != framed ||
        from_wav.samples.size() < framed ||
        std::memcmp(from_wav.samples.data(), from_flac.samples.data(), framed * sizeof(float)) != 0) {
      std::fprintf(stderr, "FLAC / WAV export does not read back\n");
      return 1;
    }
    try {
      glc::load_audio_file_lossless(std::string(argv[5]) + ".mp3");
      return 1;
    } catch (const glc::Error &e) {
      if (e.code != GLC_EINVAL) return 1;
    }
    std::ofstream out(argv[5], std::ios::binary);
    out.write(reinterpret_cast<const char *>(whole.data()), static_cast<std::streamsize>(whole.size() * sizeof(float)));
    const glc::AudioHeader h = loaded.header();
    const glc::GaplessInfo g = loaded.gapless_info();
    std::printf("{\"sample_rate\": %u, \"channels\": %u, \"total_samples\": %llu, \"encoder_delay\": %u, "
                "\"padding\": %u, \"original_length\": %llu, \"n_frames\": %llu, \"raw_frames\": %llu, "
                "\"nnz\": %llu, \"decoded\": %zu, \"chunks\": %zu}\n",
                h.sample_rate, h.channels, (unsigned long long)h.total_samples, g.encoder_delay, g.padding,
                (unsigned long long)g.original_length, (unsigned long long)loaded.n_frames(),
                (unsigned long long)raw_frames, (unsigned long long)nnz, whole.size(), chunks);
    // error behaviour: the reference panics on channels == 0; here it is an exception
    try {
      encoder.encode(pcm, 0);
      std::fprintf(stderr, "channels == 0 accepted\n");
      return 1;
    } catch (const glc::Error &e) {
      if (e.code != GLC_EINVAL) return 1;
    }
    return 0;
  } catch (const glc::Error &e) {
    std::fprintf(stderr, "glc::Error %d: %s\n", e.code, e.what());
    return 1;
  }
}
