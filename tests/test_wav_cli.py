"""WAV I/O twin (src/audio.rs) on CPU, cross-checked with Python's own `wave` module, and the
CLI twin of `glc` end-to-end on the GPU."""
import os
import struct
import subprocess
import wave

import numpy as np
import pytest

import glc_amd
from conftest import ROOT, gen_noise, gen_tone
from oracle import oracle as O

CLI = os.path.join(ROOT, "build", "glc")


def _write_wav(path, fmt, bits, ch, sr, raw: bytes, extensible=False):
    block = ch * bits // 8
    if extensible:
        sub = struct.pack("<H", fmt) + bytes.fromhex("000000001000800000aa00389b71")
        body = struct.pack("<HHIIHHHHI", 0xFFFE, ch, sr, sr * block, block, bits, 22, bits, 0) + sub
    else:
        body = struct.pack("<HHIIHH", fmt, ch, sr, sr * block, block, bits)
    chunks = b"fmt " + struct.pack("<I", len(body)) + body
    chunks += b"LIST" + struct.pack("<I", 5) + b"junk!" + b"\x00"          # odd-sized chunk + pad
    chunks += b"data" + struct.pack("<I", len(raw)) + raw
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def test_load_wav_int_formats(tmp_path):
    rng = np.random.RandomState(3)
    for bits, dtype in [(16, np.int16), (32, np.int32)]:
        v = rng.randint(np.iinfo(dtype).min, np.iinfo(dtype).max, 4000).astype(dtype)
        p = tmp_path / f"i{bits}.wav"
        _write_wav(p, 1, bits, 2, 44100, v.tobytes(), extensible=(bits == 32))
        x, sr, ch = glc_amd.load_wav(p)
        assert (sr, ch) == (44100, 2)
        # audio.rs:54-58: `(1 << (bits - 1)) as f32` on an i32 literal - at 32 bits that is i32::MIN, so the
        # reference divides by -2147483648.0 and inverts the polarity of 32-bit files (quirk Q11)
        div = np.float32(-2147483648.0) if bits == 32 else np.float32(2 ** (bits - 1))
        assert np.array_equal(x, v.astype(np.float32) / div)
    v8 = rng.randint(0, 256, 999).astype(np.uint8)
    _write_wav(tmp_path / "u8.wav", 1, 8, 1, 8000, v8.tobytes())
    x, sr, ch = glc_amd.load_wav(tmp_path / "u8.wav")
    assert np.array_equal(x, (v8.astype(np.int32) - 128).astype(np.float32) / np.float32(128))
    v24 = rng.randint(-2 ** 23, 2 ** 23, 3000).astype(np.int32)
    raw = b"".join(int(s).to_bytes(3, "little", signed=True) for s in v24)
    _write_wav(tmp_path / "i24.wav", 1, 24, 3, 96000, raw)
    x, sr, ch = glc_amd.load_wav(tmp_path / "i24.wav")
    assert (sr, ch) == (96000, 3) and np.array_equal(x, v24.astype(np.float32) / np.float32(2 ** 23))
    f = rng.uniform(-1, 1, 2000).astype(np.float32)
    _write_wav(tmp_path / "f32.wav", 3, 32, 2, 48000, f.tobytes())
    x, sr, ch = glc_amd.load_wav(tmp_path / "f32.wav")
    assert np.array_equal(x.view(np.uint32), f.view(np.uint32))


def test_load_wav_written_by_python_wave(tmp_path):
    v = (np.sin(np.arange(5000) * 0.01) * 20000).astype(np.int16)
    with wave.open(str(tmp_path / "py.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(22050); w.writeframes(v.tobytes())
    x, sr, ch = glc_amd.load_wav(tmp_path / "py.wav")
    assert (sr, ch) == (22050, 1) and np.array_equal(x, v.astype(np.float32) / np.float32(32768))


def test_export_to_wav_matches_reference_rounding(tmp_path):
    x = np.concatenate([np.linspace(-1.2, 1.2, 4001).astype(np.float32),
                        np.float32([0.0, -0.0, 1.0, -1.0, 0.99998474, 3.05e-5, -3.05e-5, np.nan])])
    glc_amd.export_to_wav(tmp_path / "o.wav", x, 48000, 1)
    with wave.open(str(tmp_path / "o.wav"), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 48000, x.size)
        got = np.frombuffer(w.readframes(x.size), np.int16)
    with np.errstate(invalid="ignore"):
        v = x * np.float32(32767.0)                       # (sample * 32767.0).clamp(..) as i16, audio.rs:11-16
        want = np.where(np.isnan(v), 0, np.trunc(np.clip(v, -32768.0, 32767.0))).astype(np.int16)
    assert np.array_equal(got, want)


def test_wav_errors(tmp_path):
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.load_wav(tmp_path / "missing.wav")
    assert e.value.code == -6
    (tmp_path / "bad.wav").write_bytes(b"RIFFxxxxWAVEnope")
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.load_wav(tmp_path / "bad.wav")
    assert e.value.code == -4
    _write_wav(tmp_path / "adpcm.wav", 2, 4, 1, 8000, b"\x00" * 64)
    with pytest.raises(glc_amd.GlcError):
        glc_amd.load_wav(tmp_path / "adpcm.wav")


@pytest.mark.gpu
def test_cli_encode_decode_matches_oracle(tmp_path):
    """`glc song.wav` then `glc -d --wav song.glc` (src/main.rs:21-113) through the CLI twin."""
    assert os.path.exists(CLI), "build/glc missing: run __graft_entry__.build()"
    sr, ch = 44100, 2
    x = np.concatenate([gen_tone("sine", 440.0, sr, ch, 0.6), gen_noise(sr, ch, 0.1, 5)])
    x16 = np.trunc(np.clip(x * np.float32(32767.0), -32768, 32767)).astype(np.int16)
    wav = tmp_path / "song.wav"
    with wave.open(str(wav), "wb") as w:
        w.setnchannels(ch); w.setsampwidth(2); w.setframerate(sr); w.writeframes(x16.tobytes())
    r = subprocess.run([CLI, str(wav)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Encoding: 44100 Hz, 2 channels, %d samples" % x16.size in r.stdout
    pcm = x16.astype(np.float32) / np.float32(32768)          # what load_wav hands the encoder
    ref = O.encode(pcm, sr, ch)
    assert (tmp_path / "song.glc").read_bytes() == ref.glc
    os.remove(wav)
    r = subprocess.run([CLI, "-d", "--wav", str(tmp_path / "song.glc")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    dref, _, _ = O.decode(ref.glc)
    want = np.trunc(np.clip(dref * np.float32(32767.0), -32768, 32767)).astype(np.int16)
    with wave.open(str(wav), "rb") as w:
        assert w.getnframes() * ch == dref.size
        got = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    assert np.array_equal(got, want)
    # error paths: a failed file does not stop the others, exit code 1 (src/main.rs:546-581)
    r = subprocess.run([CLI, str(tmp_path / "nope.wav"), str(wav)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Saved" in r.stdout
    r = subprocess.run([CLI, "-d", str(tmp_path / "nope.glc"), str(wav)], capture_output=True, text=True)
    assert r.returncode == 1 and "File not found" in r.stderr and "Not a .glc file" in r.stderr


def test_cpp_mirror_header_compiles_standalone(tmp_path):
    """include/glc.hpp is self-contained C++17 (no HIP headers, no torch) and warning-free."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "glc.hpp"\nint main() { return sizeof(glc::Encoder) + sizeof(glc::Decoder) == 0; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ch,family", [(1, "tone"), (2, "mixed"), (2, "noise")])
def test_cpp_mirror_roundtrip_matches_oracle(tmp_path, ch, family):
    """glc::Encoder / glc::Decoder / save_encoded / load_encoded / decode_streaming (include/glc.hpp,
    the C++ twin of src/codec.rs's public API) in a process with no Python and no torch."""
    import json
    exe = os.path.join(ROOT, "build", "glc_cpp_roundtrip")
    assert os.path.exists(exe), "build/glc_cpp_roundtrip missing: run __graft_entry__.build()"
    sr = 44100
    if family == "tone":
        x = gen_tone("sine", 440.0, sr, ch, 1.3)
    elif family == "noise":
        x = gen_noise(sr, ch, 0.4, 9)                     # raw-PCM fallback frames
    else:
        x = np.concatenate([gen_tone("square", 220.0, sr, ch, 12.0), gen_noise(sr, ch, 0.2, 3)])  # > 1 chunk
    (tmp_path / "in.f32").write_bytes(np.ascontiguousarray(x, np.float32).tobytes())
    r = subprocess.run([exe, str(tmp_path / "in.f32"), str(sr), str(ch), str(tmp_path / "o.glc"),
                        str(tmp_path / "o.f32")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout)
    ref = O.encode(x, sr, ch)
    assert (tmp_path / "o.glc").read_bytes() == ref.glc
    dref, _, _ = O.decode(ref.glc)
    got = np.frombuffer((tmp_path / "o.f32").read_bytes(), np.float32)
    assert got.size == dref.size == info["decoded"] and np.array_equal(got.view(np.uint32), dref.view(np.uint32))
    assert info["n_frames"] == ref.n_frames and info["original_length"] == x.size
    assert info["chunks"] == -(-ref.n_frames // 500)
    from oracle import flac_oracle as F
    assert (tmp_path / "o.f32.flac").read_bytes() == F.encode_flac_with_level(dref, sr, ch, 5)
    if family == "noise":
        assert info["raw_frames"] > 0
