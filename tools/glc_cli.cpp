// glc_cli.cpp — command-line twin of the reference's `glc` binary (src/main.rs) for the paths
// this repository implements: `glc file.wav|file.flac ...` encodes to .glc (encode_file,
// src/main.rs:21-52) and `glc -d file.glc ... [--wav] [--flac-level N]` decodes to FLAC (default)
// or 16-bit WAV (decode_file, :55-113; argument handling :354-583).  It uses only the C ABI of
// libglc_hip.so.  Playback (-p) and the GUI stay with the reference.
// Build: g++ -O2 -std=c++17 -Iinclude tools/glc_cli.cpp -Lgapless-lossy-codec_amd -lglc_hip \
//        -Wl,-rpath,'$ORIGIN/../gapless-lossy-codec_amd' -o build/glc
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "glc.h"

static std::string with_ext(const std::string &path, const char *ext) {
  const size_t slash = path.find_last_of('/');
  const size_t dot = path.find_last_of('.');
  const std::string stem = (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ? path.substr(0, dot) : path;
  return stem + "." + ext;
}

static std::string lower_ext(const std::string &path) {
  const size_t slash = path.find_last_of('/');
  const size_t dot = path.find_last_of('.');
  if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
  std::string e = path.substr(dot + 1);
  for (char &c : e) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return e;
}

static bool exists(const std::string &p) {
  FILE *f = std::fopen(p.c_str(), "rb");
  if (f) std::fclose(f);
  return f != nullptr;
}

static std::string file_name(const std::string &path) {
  const size_t slash = path.find_last_of('/');
  return slash == std::string::npos ? path : path.substr(slash + 1);
}

static long file_size(const std::string &p) {
  FILE *f = std::fopen(p.c_str(), "rb");
  if (!f) return -1;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fclose(f);
  return n;
}

static int encode_file(const std::string &in) {
  std::printf("Loading: \"%s\"\n", file_name(in).c_str());
  float *pcm = nullptr;
  uint64_t n = 0;
  uint32_t sr = 0;
  uint16_t ch = 0;
  // load_audio_file_lossless, src/audio.rs:19-36: by lower-cased extension
  const int lrc = lower_ext(in) == "flac" ? glc_flac_load(in.c_str(), &pcm, &n, &sr, &ch)
                                          : glc_wav_load(in.c_str(), &pcm, &n, &sr, &ch);
  if (lrc != GLC_OK) {
    std::fprintf(stderr, "Error encoding file: %s\n", glc_last_error(nullptr));
    return 1;
  }
  std::printf("Encoding: %u Hz, %u channels, %llu samples\n", sr, ch, static_cast<unsigned long long>(n));
  glc_ctx *ctx = nullptr;
  glc_frames *fr = nullptr;
  int rc = glc_ctx_create(0, sr, &ctx);
  if (rc == GLC_OK) rc = glc_encode(ctx, pcm, n, ch, &fr);
  glc_free(pcm);
  if (rc != GLC_OK) {
    std::fprintf(stderr, "Error encoding file: %s\n", glc_last_error(ctx));
    glc_ctx_destroy(ctx);
    return 1;
  }
  const std::string out = with_ext(in, "glc");
  rc = glc_save(fr, out.c_str());
  glc_frames_free(fr);
  glc_ctx_destroy(ctx);
  if (rc != GLC_OK) {
    std::fprintf(stderr, "Error encoding file: %s\n", glc_last_error(nullptr));
    return 1;
  }
  const long a = file_size(in), b = file_size(out);
  std::printf("Saved: \"%s\" (%ld bytes, %.1f%% of original)\n", file_name(out).c_str(), b, 100.0 * b / a);
  return 0;
}

static int decode_file(const std::string &in, bool wav, unsigned flac_level) {
  std::printf("Loading: \"%s\"\n", file_name(in).c_str());
  glc_frames *fr = nullptr;
  if (glc_load(in.c_str(), &fr) != GLC_OK) {
    std::fprintf(stderr, "Error decoding file: %s\n", glc_last_error(nullptr));
    return 1;
  }
  glc_info info;
  glc_frames_info(fr, &info);
  std::printf("Decoding: %u Hz, %u channels\n", info.sample_rate, info.channels);
  glc_ctx *ctx = nullptr;
  int rc = glc_ctx_create(0, info.sample_rate, &ctx);
  std::vector<float> pcm(glc_decoded_len(fr));
  uint64_t n = 0;
  if (rc == GLC_OK) rc = glc_decode(ctx, fr, pcm.data(), pcm.size(), &n);
  glc_frames_free(fr);
  if (rc != GLC_OK) {
    std::fprintf(stderr, "Error decoding file: %s\n", glc_last_error(ctx));
    glc_ctx_destroy(ctx);
    return 1;
  }
  glc_ctx_destroy(ctx);
  std::printf("Decoded %llu samples\n", static_cast<unsigned long long>(n));
  const std::string out = with_ext(in, wav ? "wav" : "flac");
  rc = wav ? glc_wav_save16(out.c_str(), pcm.data(), n, info.sample_rate, info.channels)
           : glc_flac_save(out.c_str(), pcm.data(), n, info.sample_rate, info.channels, static_cast<uint8_t>(flac_level));
  if (rc != GLC_OK) {
    std::fprintf(stderr, "Error decoding file: %s\n", glc_last_error(nullptr));
    return 1;
  }
  if (wav) std::printf("Saved: \"%s\" (WAV)\n", file_name(out).c_str());
  else std::printf("Saved: \"%s\" (FLAC, level %u)\n", file_name(out).c_str(), flac_level);
  return 0;
}

static void print_usage() {
  std::fprintf(stderr,
               "Usage:\n"
               "  glc <file.wav|file.flac> ...                    Encode audio files to .glc (MI355X)\n"
               "  glc -d <file.glc> ... [--wav] [--flac-level N]  Decode .glc files\n"
               "\n"
               "Options:\n"
               "  -d, --decode       Decode .glc files to FLAC (default) or WAV\n"
               "      --wav          Output WAV format instead of FLAC\n"
               "      --flac-level   Set FLAC compression level 0-8 (default: 5)\n"
               "\n"
               "Playback (-p) and the GUI are not part of this build; use the reference binary.\n");
}

int main(int argc, char **argv) {
  if (argc < 2) {  // the reference launches its GUI here, or prints the usage without the ui feature
    print_usage();
    return 1;
  }
  const std::string first = argv[1];
  if (first == "-h" || first == "--help") {
    print_usage();
    return 1;
  }
  if (first == "-p" || first == "--play") {
    std::fprintf(stderr, "Error: playback stays with the reference binary\n");
    return 1;
  }
  bool failed = false;
  if (first == "-d" || first == "--decode") {  // src/main.rs:364-456
    if (argc < 3) {
      std::fprintf(stderr, "Error: -d requires at least one .glc file\n");
      print_usage();
      return 1;
    }
    bool wav = false;
    unsigned level = 5;
    std::vector<std::string> files;
    for (int i = 2; i < argc; ++i) {
      const std::string a = argv[i];
      if (a == "--wav") {
        wav = true;
      } else if (a == "--flac-level") {
        if (i + 1 >= argc) {
          std::fprintf(stderr, "Error: --flac-level requires a value (0-8)\n");
          return 1;
        }
        const std::string v = argv[++i];  // `parse::<u8>()`: optional '+', digits, <= 255
        size_t k = (!v.empty() && v[0] == '+') ? 1 : 0;
        bool ok = k < v.size() && v.size() - k <= 3;
        for (size_t j = k; ok && j < v.size(); ++j) ok = std::isdigit(static_cast<unsigned char>(v[j])) != 0;
        const unsigned long parsed = ok ? std::strtoul(v.c_str() + k, nullptr, 10) : 256;
        if (parsed > 255) {
          std::fprintf(stderr, "Error: Invalid FLAC level, must be 0-8\n");
          return 1;
        }
        if (parsed > 8) {
          std::fprintf(stderr, "Error: FLAC level must be 0-8\n");
          return 1;
        }
        level = static_cast<unsigned>(parsed);
      } else if (!exists(a)) {
        std::fprintf(stderr, "Error: File not found: \"%s\"\n", a.c_str());
        failed = true;
      } else if (lower_ext(a) != "glc") {
        std::fprintf(stderr, "Error: Not a .glc file: \"%s\"\n", a.c_str());
        failed = true;
      } else {
        files.push_back(a);
      }
    }
    if (files.empty()) {
      std::fprintf(stderr, "Error: No valid .glc files to decode\n");
      return 1;
    }
    for (const std::string &f : files) failed |= decode_file(f, wav, level) != 0;
    return failed ? 1 : 0;
  }
  for (int i = 1; i < argc; ++i) {  // src/main.rs:546-581: keep going, exit 1 if any file failed
    const std::string a = argv[i];
    if (!exists(a)) {
      std::fprintf(stderr, "Error: File not found: \"%s\"\n", a.c_str());
      failed = true;
    } else if (lower_ext(a) != "wav" && lower_ext(a) != "flac") {
      std::fprintf(stderr, "Error: Unsupported file type: \"%s\"\nSupported formats: WAV, FLAC\n", a.c_str());
      failed = true;
    } else {
      failed |= encode_file(a) != 0;
    }
  }
  return failed ? 1 : 0;
}
