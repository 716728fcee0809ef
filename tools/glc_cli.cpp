// glc_cli.cpp — command-line twin of the reference's `glc` binary (src/main.rs) for the paths
// this repository implements: `glc file.wav ...` encodes to .glc (encode_file, src/main.rs:21-52)
// and `glc -d --wav file.glc ...` decodes to 16-bit WAV (decode_file, :55-113).  It uses only the
// C ABI of libglc_hip.so.  FLAC input/output, playback and the GUI stay with the reference.
// Build: g++ -O2 -std=c++17 -Iinclude tools/glc_cli.cpp -Lgapless-lossy-codec_amd -lglc_hip \
//        -Wl,-rpath,'$ORIGIN/../gapless-lossy-codec_amd' -o build/glc
#include <cstdio>
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
  if (glc_wav_load(in.c_str(), &pcm, &n, &sr, &ch) != GLC_OK) {
    std::fprintf(stderr, "Error: %s\n", glc_last_error(nullptr));
    return 1;
  }
  std::printf("Encoding: %u Hz, %u channels, %llu samples\n", sr, ch, static_cast<unsigned long long>(n));
  glc_ctx *ctx = nullptr;
  glc_frames *fr = nullptr;
  int rc = glc_ctx_create(0, sr, &ctx);
  if (rc == GLC_OK) rc = glc_encode(ctx, pcm, n, ch, &fr);
  glc_free(pcm);
  if (rc != GLC_OK) {
    std::fprintf(stderr, "Error: %s\n", glc_last_error(ctx));
    glc_ctx_destroy(ctx);
    return 1;
  }
  const std::string out = with_ext(in, "glc");
  rc = glc_save(fr, out.c_str());
  glc_frames_free(fr);
  glc_ctx_destroy(ctx);
  if (rc != GLC_OK) {
    std::fprintf(stderr, "Error: %s\n", glc_last_error(nullptr));
    return 1;
  }
  const long a = file_size(in), b = file_size(out);
  std::printf("Saved: \"%s\" (%ld bytes, %.1f%% of original)\n", file_name(out).c_str(), b, 100.0 * b / a);
  return 0;
}

static int decode_file(const std::string &in) {
  std::printf("Loading: \"%s\"\n", file_name(in).c_str());
  glc_frames *fr = nullptr;
  if (glc_load(in.c_str(), &fr) != GLC_OK) {
    std::fprintf(stderr, "Error: %s\n", glc_last_error(nullptr));
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
    std::fprintf(stderr, "Error: %s\n", glc_last_error(ctx));
    glc_ctx_destroy(ctx);
    return 1;
  }
  glc_ctx_destroy(ctx);
  std::printf("Decoded %llu samples\n", static_cast<unsigned long long>(n));
  const std::string out = with_ext(in, "wav");
  if (glc_wav_save16(out.c_str(), pcm.data(), n, info.sample_rate, info.channels) != GLC_OK) {
    std::fprintf(stderr, "Error: %s\n", glc_last_error(nullptr));
    return 1;
  }
  std::printf("Saved: \"%s\" (WAV)\n", file_name(out).c_str());
  return 0;
}

int main(int argc, char **argv) {
  bool decode = false, wav = false;
  std::vector<std::string> files;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-d" || a == "--decode") decode = true;
    else if (a == "--wav") wav = true;
    else if (a == "-h" || a == "--help") { files.clear(); break; }
    else if (a == "-p" || a == "--play" || a == "--ffplay" || a == "--flac-level") {
      std::fprintf(stderr, "%s: playback and FLAC export stay with the reference binary\n", a.c_str());
      return 2;
    } else files.push_back(a);
  }
  if (files.empty()) {
    std::fprintf(stderr, "usage: glc file.wav [...]        encode to .glc (MI355X)\n"
                         "       glc -d --wav file.glc [...] decode to 16-bit WAV\n");
    return 2;
  }
  if (decode && !wav) {
    std::fprintf(stderr, "glc -d: only --wav output is implemented here (FLAC export stays with the reference)\n");
    return 2;
  }
  int failed = 0;  // like src/main.rs:546-581: keep going, exit 1 if any file failed
  for (const std::string &f : files) failed += decode ? decode_file(f) : encode_file(f);
  return failed ? 1 : 0;
}
