"""FLAC export / import twin (src/flac.rs, src/audio.rs:19-36,68-96).  Host-only
integer work, so everything except the CLI legs runs without a GPU.

* the encoder is compared byte for byte with the numpy restatement of src/flac.rs in
  oracle/flac_oracle.py, and every stream is also read back by that module's independent RFC 9639
  decoder (all CRCs + MD5) — the role `claxon` plays in the reference's tests/test_flac.rs;
* the decoder (`glc_flac_load`, the stand-in for claxon) is checked on streams from tests/flac_synth.py
  that use the parts of the format the reference's encoder never emits.

Parity status of this row: unpinned by the reference (no .flac fixture, no known-answer test there).
"""
import hashlib
import os
import subprocess
import wave

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import flac_synth as S
import glc_amd
from conftest import ROOT, gen_noise, gen_tone
from oracle import flac_oracle as F

CLI = os.path.join(ROOT, "build", "glc")


def _check_stream(x, sr, ch, level=5, decodable=True):
    x = np.asarray(x, np.float32)
    got = glc_amd.encode_flac_with_level(x, sr, ch, level)
    want = F.encode_flac_with_level(x, sr, ch, level)
    assert got == want, "bytes differ from the restatement of src/flac.rs"
    i16 = F.to_i16(x)
    whole = i16[:(i16.size // ch) * ch]
    if decodable:
        pcm, rate, nch, bps = F.decode_flac(got, verify_md5=(whole.size == i16.size))
        assert (rate, nch, bps) == (sr, ch, 16) and np.array_equal(pcm, whole.astype(np.int64))
        y, rate2, nch2 = glc_amd.decode_flac(got)
        assert (rate2, nch2) == (sr, ch)
        assert np.array_equal(y, whole.astype(np.float32) / np.float32(32768.0))   # audio.rs:72-80
    return got


# ---------------------------------------------------------------- the reference's own tests, tests/test_flac.rs
def _ref_test_signal(tmp_path, name, samples, sr, ch):
    """tests/test_flac.rs:4-52: export_to_flac + export_to_wav, load back, metadata and RMS bound."""
    samples = np.asarray(samples, np.float32)
    fp, wp = tmp_path / f"{name}.flac", tmp_path / f"{name}.wav"
    glc_amd.export_to_flac(fp, samples, sr, ch)
    glc_amd.export_to_wav(wp, samples, sr, ch)
    loaded, rate, nch = glc_amd.load_audio_file_lossless(fp)
    assert rate == sr and nch == ch and loaded.size == samples.size
    rms = np.sqrt(np.mean((samples - loaded) ** 2, dtype=np.float32))
    assert rms < 0.0001
    assert fp.read_bytes() == _check_stream(samples, sr, ch)
    return os.path.getsize(fp), os.path.getsize(wp)


def test_flac_silence(tmp_path):
    f, w = _ref_test_signal(tmp_path, "silence", np.zeros(1000), 44100, 1)
    assert f < w


def test_flac_dc_offset(tmp_path):
    _ref_test_signal(tmp_path, "dc", np.full(1000, 0.5), 44100, 1)


def test_flac_sine_wave(tmp_path):
    t = np.arange(4410, dtype=np.float32) / np.float32(44100.0)
    _ref_test_signal(tmp_path, "sine", np.sin(np.float32(2.0 * np.pi) * np.float32(440.0) * t) * np.float32(0.8), 44100, 1)


def test_flac_white_noise(tmp_path):
    seed, out = 12345, []
    for _ in range(8820):                                    # tests/test_flac.rs:83-91
        seed = (seed * 1103515245 + 12345) & 0xFFFFFFFF
        out.append(np.float32(((seed >> 16) & 0x7FFF)) / np.float32(32768.0) * np.float32(2.0) - np.float32(1.0))
    _ref_test_signal(tmp_path, "noise", out, 44100, 1)


def test_flac_stereo(tmp_path):
    t = np.arange(4410, dtype=np.float32) / np.float32(44100.0)
    lr = np.stack([np.sin(np.float32(2 * np.pi * 440.0) * t) * 0.5, np.sin(np.float32(2 * np.pi * 880.0) * t) * 0.5], 1)
    _ref_test_signal(tmp_path, "stereo", lr.reshape(-1), 44100, 2)


def test_flac_sample_rates(tmp_path):
    _ref_test_signal(tmp_path, "48khz", np.zeros(4800), 48000, 1)
    _ref_test_signal(tmp_path, "96khz", np.zeros(9600), 96000, 1)


def test_flac_minimum_size(tmp_path):
    _ref_test_signal(tmp_path, "small", np.arange(16, dtype=np.float32) / 16.0 * 2.0 - 1.0, 8000, 1)


def test_flac_compression_levels(tmp_path):
    t = np.arange(1000, dtype=np.float32) / np.float32(44100.0)
    x = (np.sin(np.float32(2 * np.pi * 440.0) * t) * 0.5).astype(np.float32)
    sizes = []
    for level in range(9):
        p = tmp_path / f"test_level_{level}.flac"
        glc_amd.export_to_flac_with_level(p, x, 44100, 1, level)
        loaded, _, _ = glc_amd.load_audio_file_lossless(p)
        assert loaded.size == x.size
        assert p.read_bytes() == _check_stream(x, 44100, 1, level)
        sizes.append(os.path.getsize(p))
    assert sizes[0] == max(sizes)            # verbatim is the largest


# ---------------------------------------------------------------- byte identity over inputs / levels / layouts
def _impulses(n):
    x = np.zeros(n, np.float32)
    x[5::997] = 1.0
    x[300::1409] = -1.0
    return x


FAMILIES = {
    "tone": lambda sr, ch, n: gen_tone("sine", 440.0, sr, ch, n / sr)[:n * ch],
    "square": lambda sr, ch, n: gen_tone("square", 220.0, sr, ch, n / sr)[:n * ch],
    "noise": lambda sr, ch, n: gen_noise(sr, ch, n / sr, 7)[:n * ch],
    "loud": lambda sr, ch, n: (gen_noise(sr, ch, n / sr, 3)[:n * ch] * 4.0).astype(np.float32),   # clips both ways
    "impulse": lambda sr, ch, n: _impulses(n * ch),                       # Rice zero runs of ~2^19 bits
    "quiet": lambda sr, ch, n: (gen_noise(sr, ch, n / sr, 4)[:n * ch] * 1e-4).astype(np.float32),
}


@pytest.mark.parametrize("level", range(9))
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_encoder_bytes_match_restatement(family, level):
    for sr, ch, n in ((44100, 1, 5000), (48000, 2, 9000), (96000, 6, 4200)):
        _check_stream(FAMILIES[family](sr, ch, n), sr, ch, level)


@pytest.mark.parametrize("sr", [8000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 176400, 192000,
                                11025, 12345, 384000, 655350, 1048575])
def test_every_sample_rate_code(sr):
    _check_stream(gen_tone("sine", 100.0, 8000, 1, 0.05), sr, 1)


@pytest.mark.parametrize("n", [16, 17, 191, 192, 255, 256, 257, 576, 1024, 1152, 2048, 2304, 4095, 4096, 4097,
                               4096 + 16, 4096 + 255, 4096 + 256, 4096 + 4095, 8192, 4608, 12288 + 5])
@pytest.mark.parametrize("level", [2, 5, 8])
def test_block_size_codes_and_last_block(n, level):
    x = gen_noise(44100, 2, n / 44100 + 0.01, n)[:2 * n] * 0.25
    _check_stream(x, 44100, 2, level)


def test_multichannel_up_to_eight_and_header_wrap():
    for ch in (3, 4, 5, 7, 8):
        _check_stream(gen_noise(48000, ch, 0.05, ch), 48000, ch)
    # channels > 8 do not fit STREAMINFO's 3 bits / the frame header's 4; the reference writes the
    # wrapped fields anyway (flac.rs:935, :831) and so does the twin (an undecodable stream)
    _check_stream(gen_noise(48000, 11, 0.05, 1), 48000, 11, decodable=False)


def test_trailing_partial_sample_frame_is_hashed_but_not_framed():
    """flac.rs:960 (total = len / channels), :1021-1030 (loop stops at remaining/channels == 0), but
    compute_md5 (:1001) hashes every converted sample."""
    x = gen_noise(44100, 1, 0.2, 9)[:5001]
    got = _check_stream(x, 44100, 2)
    i16 = F.to_i16(x)
    assert got[26:42] == hashlib.md5(i16.tobytes()).digest()
    assert got[26:42] != hashlib.md5(i16[:5000].tobytes()).digest()
    y, _, _ = glc_amd.decode_flac(got)
    assert y.size == 5000


def test_last_block_equal_to_predictor_order_quirk():
    """A final block of exactly `order` samples gets a residual section with no Rice parameter
    (flac.rs:632-635 `continue`), which no decoder can parse; the twin reproduces the bytes."""
    for level, tail in ((5, 4), (3, 3), (8, 4)):
        x = gen_noise(44100, 1, 0.2, 5)[:4096 + tail]
        got = _check_stream(x, 44100, 1, level, decodable=False)
        with pytest.raises(F.FlacError):
            F.decode_flac(got)
        with pytest.raises(glc_amd.GlcError) as e:
            glc_amd.decode_flac(got)
        assert e.value.code == -4
    for level, tail in ((5, 3), (5, 5), (3, 2), (1, 1), (2, 2), (2, 1)):   # neighbours are fine
        _check_stream(gen_noise(44100, 1, 0.2, 5)[:4096 + tail] if level > 2 else
                      gen_noise(44100, 1, 0.2, 5)[:1152 + tail], 44100, 1, level,
                      decodable=not (level == 1 and tail == 1) and not (level == 2 and tail == 2))


def test_non_finite_and_out_of_range_samples():
    x = gen_tone("sine", 300.0, 44100, 2, 0.1)
    x[5], x[6], x[7], x[8], x[9] = np.nan, np.inf, -np.inf, 1.0, -1.0
    x[10], x[11] = np.float32(32767.5 / 32767.0), np.float32(-1.00004)
    got = _check_stream(x, 44100, 2)
    y, _, _ = glc_amd.decode_flac(got)
    assert y[5] == 0 and y[6] == np.float32(32767 / 32768) and y[7] == -1.0 and y[9] == np.float32(-32767 / 32768)


def test_errors_match_the_reference():
    x = np.zeros(64, np.float32)
    for n, ch in ((15, 1), (31, 2), (0, 1)):
        with pytest.raises(glc_amd.GlcError, match=f"FLAC requires at least 16 samples per channel, got {n // ch}"):
            glc_amd.encode_flac(x[:n], 44100, ch)
        with pytest.raises(ValueError):
            F.encode_flac_with_level(x[:n], 44100, ch, 5)
    with pytest.raises(glc_amd.GlcError, match="Invalid compression level 9, must be 0-8"):
        glc_amd.encode_flac_with_level(x, 44100, 1, 9)
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.encode_flac(x, 44100, 0)
    assert e.value.code == -1
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.export_to_flac("/nonexistent-dir/x.flac", x, 44100, 1)
    assert e.value.code == -6
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.load_flac("/nonexistent-dir/x.flac")
    assert e.value.code == -6


def test_threaded_encoder_is_deterministic_and_matches():
    """Enough frames (> 8 per worker) for the frame fan-out to use several threads."""
    sr, ch = 48000, 2
    x = np.concatenate([gen_tone("sine", 440.0, sr, ch, 4.0), gen_noise(sr, ch, 3.0, 11) * 0.3])
    a = _check_stream(x, sr, ch, 5)
    assert a == glc_amd.encode_flac(x, sr, ch)
    _check_stream(x[: 2 * 300000], sr, ch, 1)


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.integers(16, 9000), st.integers(1, 8), st.integers(0, 8), st.integers(0, 2 ** 32 - 1),
       st.sampled_from([8000, 44100, 48000, 96000, 7]), st.sampled_from([1e-3, 0.1, 1.0, 3.0]))
def test_fuzz_encoder(n, ch, level, seed, sr, gain):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n * ch) * gain).astype(np.float32)
    if seed & 1:
        x = np.cumsum(x).astype(np.float32) * np.float32(0.05)     # smooth: small residuals, long zero runs
    order = {0: 0, 1: 1, 2: 2, 3: 3, 4: 3}.get(level, 4)
    last = n % (1152 if level <= 2 else 4096) if n > (1152 if level <= 2 else 4096) else n
    _check_stream(x, sr, ch, level, decodable=not (order and last == order))


# ---------------------------------------------------------------- MD5Context (flac.rs:83-302) is plain MD5
@pytest.mark.parametrize("n", [0, 1, 27, 28, 29, 31, 32, 33, 59, 60, 61, 63, 64, 65, 200])
def test_reference_md5_structure_is_standard_md5(n):
    i16 = (np.arange(n) * 977 % 65536 - 32768).astype(np.int16)
    assert F.compute_md5_ref(i16) == hashlib.md5(i16.tobytes()).digest()


def test_encoder_with_reference_structured_md5():
    x = gen_tone("sine", 440.0, 8000, 1, 0.05)
    assert F.encode_flac_with_level(x, 8000, 1, 5, md5=F.compute_md5_ref) == glc_amd.encode_flac(x, 8000, 1)


# ---------------------------------------------------------------- decoder (≙ claxon) on the rest of RFC 9639
def _pcm(n, ch, bps, seed, smooth=True):
    rng = np.random.default_rng(seed)
    top = (1 << (bps - 1)) - 1
    if smooth:
        t = np.arange(n)[:, None] / 40.0 + np.arange(ch)[None, :]
        v = np.sin(t) * top * 0.6 + rng.integers(-3, 4, (n, ch))
    else:
        v = rng.integers(-top - 1, top + 1, (n, ch))
    return np.clip(np.rint(v), -top - 1, top).astype(np.int64).reshape(-1)


def _check_decode(data, pcm, ch, bps, rate):
    # audio.rs:72 `(1 << (bits_per_sample - 1)) as f32` on an i32 literal: i32::MIN at 32 bits (quirk Q11)
    div = np.float32(-2147483648.0) if bps == 32 else np.float32(1 << (bps - 1))
    want = (pcm.astype(np.int32).astype(np.float32)) / div
    y, r, c = glc_amd.decode_flac(data)
    assert (r, c) == (rate, ch) and np.array_equal(y, want)
    p2, r2, c2, b2 = F.decode_flac(data)
    assert (r2, c2, b2) == (rate, ch, bps) and np.array_equal(p2, pcm)


@pytest.mark.parametrize("bps", [8, 12, 16, 20, 24, 32])
def test_decoder_sample_sizes_and_subframe_kinds(bps):
    n, ch, rate = 3 * 576, 2, 44100
    pcm = _pcm(n, ch, bps, bps)
    pcm.reshape(-1, ch)[576:1152, 1] = 5                      # a constant block for channel 1
    kinds = [
        [dict(kind="fixed", order=2, porder=3), dict(kind="verbatim")],
        [dict(kind="lpc", order=3, coefs=[1500, -900, 200], precision=12, shift=10, porder=2, method=1),
         dict(kind="constant")],
        [dict(kind="fixed", order=4, porder=0, escape_parts=(0,)), dict(kind="fixed", order=0, porder=6, method=1)],
    ]
    data = S.stream(pcm, ch, bps, rate, [576] * 3, lambda i: kinds[i])
    _check_decode(data, pcm, ch, bps, rate)


@pytest.mark.parametrize("assignment", ["left_side", "side_right", "mid_side"])
@pytest.mark.parametrize("bps", [16, 24, 32])
def test_decoder_stereo_decorrelation(assignment, bps):
    pcm = _pcm(1024, 2, bps, 3, smooth=(bps != 16))
    if bps == 16:                                              # extreme values: side needs the 17th bit
        pcm[:8] = [32767, -32768, -32768, 32767, 32767, 32767, -32768, -32768]
    spec = [dict(kind="fixed", order=1, porder=2), dict(kind="fixed", order=1, porder=2)]
    data = S.stream(pcm, 2, bps, 48000, [512, 512], lambda i: spec, lambda i: assignment if i else "independent")
    _check_decode(data, pcm, 2, bps, 48000)


def test_decoder_lpc_orders_precisions_shifts():
    rng = np.random.default_rng(5)
    pcm = _pcm(4 * 256, 1, 16, 8)
    specs = []
    for order, precision, shift in ((1, 2, 0), (8, 15, 14), (12, 9, 7), (32, 12, 12)):
        top = 1 << (precision - 1)
        specs.append([dict(kind="lpc", order=order, coefs=rng.integers(-top, top, order).tolist(), precision=precision,
                           shift=shift, porder=1)])
    # wild coefficients blow the residual up; keep them tame for the two big orders
    specs[2][0]["coefs"] = (np.array(specs[2][0]["coefs"]) // 16).tolist()
    specs[3][0]["coefs"] = (np.array(specs[3][0]["coefs"]) // 64).tolist()
    specs[1][0]["coefs"] = (np.array(specs[1][0]["coefs"]) // 16).tolist()
    data = S.stream(pcm, 1, 16, 32000, [256] * 4, lambda i: specs[i])
    _check_decode(data, pcm, 1, 16, 32000)


def test_decoder_wasted_bits_variable_blocks_uncommon_codes_and_metadata():
    ch, bps, rate = 2, 16, 50000                              # rate code 12 (kHz in 8 bits)
    blocks = [100, 256, 4608, 1000, 16, 257]                  # 8-bit, table, table, 16-bit, 8-bit, 16-bit codes
    pcm = _pcm(sum(blocks), ch, bps, 21)
    pcm.reshape(-1, ch)[:, 0] &= ~7                           # three wasted bits on channel 0
    spec = [dict(kind="fixed", order=2, wasted=3), dict(kind="lpc", order=2, coefs=[60, -29], precision=7, shift=5)]
    meta = [(4, b"\x00" * 40), (1, b"\x00" * 13)]             # VORBIS_COMMENT-typed blob, PADDING
    data = S.stream(pcm, ch, bps, rate, blocks, lambda i: spec, variable=True, extra_metadata=meta, total_known=False,
                    bps_from_info=True)
    _check_decode(data, pcm, ch, bps, rate)
    for rate2 in (44101, 441000, 22050):                      # codes 13 (Hz, 16 bits), 14 (Hz/10), table
        data = S.stream(pcm[:2 * 356], ch, bps, rate2, [100, 256], lambda i: spec)
        _check_decode(data, pcm[:2 * 356], ch, bps, rate2)
    for nch in (1, 3, 8):
        p = _pcm(192, nch, 16, nch)
        data = S.stream(p, nch, 16, 8000, [192], lambda i: [dict(kind="fixed", order=3)] * nch)
        _check_decode(data, p, nch, 16, 8000)


def test_decoder_rejects_damage():
    x = gen_tone("sine", 440.0, 44100, 2, 0.3)
    good = glc_amd.encode_flac(x, 44100, 2)
    glc_amd.decode_flac(good)
    rng = np.random.default_rng(0)
    for _ in range(60):                                       # any flipped byte after STREAMINFO trips a CRC / syntax check
        bad = bytearray(good)
        at = int(rng.integers(42, len(good)))
        bad[at] ^= 1 << int(rng.integers(0, 8))
        with pytest.raises(glc_amd.GlcError) as e:
            glc_amd.decode_flac(bytes(bad))
        assert e.value.code == -4
    assert glc_amd.decode_flac(good[:42])[0].size == 0        # metadata only: no frames, no samples
    for cut in (0, 3, 4, 20, 41, 43, 60, len(good) - 1, len(good) - 2):
        with pytest.raises(glc_amd.GlcError):
            glc_amd.decode_flac(good[:cut])
    with pytest.raises(glc_amd.GlcError):
        glc_amd.decode_flac(b"RIFF" + good[4:])
    with pytest.raises(glc_amd.GlcError):                     # no STREAMINFO
        glc_amd.decode_flac(b"fLaC" + bytes([0x81, 0, 0, 4]) + b"\0" * 4)


@settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck))
@given(st.binary(min_size=0, max_size=300), st.booleans())
def test_fuzz_decoder_never_crashes(blob, with_header):
    head = glc_amd.encode_flac(np.zeros(16, np.float32), 44100, 1)[:42] if with_header else b""
    try:
        y, sr, ch = glc_amd.decode_flac(head + blob)
        assert y.size % max(ch, 1) == 0
    except glc_amd.GlcError as e:
        assert e.code in (-4, -1)


# ---------------------------------------------------------------- audio::load_audio_file_lossless + the CLI legs
def test_load_audio_file_lossless_dispatch(tmp_path):
    x = gen_tone("sine", 440.0, 44100, 2, 0.1)
    glc_amd.export_to_flac(tmp_path / "a.FLAC", x, 44100, 2)
    glc_amd.export_to_wav(tmp_path / "a.WaV", x, 44100, 2)
    f, sr, ch = glc_amd.load_audio_file_lossless(tmp_path / "a.FLAC")
    w, sr2, ch2 = glc_amd.load_audio_file_lossless(tmp_path / "a.WaV")
    assert (sr, ch) == (sr2, ch2) == (44100, 2)
    assert np.array_equal(f * np.float32(32768), w * np.float32(32768))   # same i16 behind both
    with pytest.raises(glc_amd.GlcError, match="Unsupported file format: mp3"):
        glc_amd.load_audio_file_lossless(tmp_path / "a.mp3")
    with pytest.raises(glc_amd.GlcError, match="No file extension"):
        glc_amd.load_audio_file_lossless(tmp_path / "noext")


@pytest.mark.gpu
def test_cli_decodes_to_flac_by_default_and_encodes_flac_input(tmp_path):
    """`glc -d song.glc [--flac-level N]` (src/main.rs:375-440, :83-94) and `glc song.flac`
    (:21-52 through load_audio_file_lossless)."""
    from oracle import oracle as O
    assert os.path.exists(CLI), "build/glc missing: run __graft_entry__.build()"
    sr, ch = 44100, 2
    x = np.concatenate([gen_tone("sine", 440.0, sr, ch, 0.7), gen_noise(sr, ch, 0.1, 5)])
    glc_amd.export_to_flac(tmp_path / "song.flac", x, sr, ch)
    r = subprocess.run([CLI, str(tmp_path / "song.flac")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    pcm = F.to_i16(x).astype(np.float32) / np.float32(32768)             # what load_flac hands the encoder
    ref = O.encode(pcm, sr, ch)
    assert (tmp_path / "song.glc").read_bytes() == ref.glc
    os.remove(tmp_path / "song.flac")
    dref, _, _ = O.decode(ref.glc)
    for args, level in (([], 5), (["--flac-level", "8"], 8), (["--flac-level", "0"], 0)):
        r = subprocess.run([CLI, "-d", str(tmp_path / "song.glc")] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert f'Saved: "song.flac" (FLAC, level {level})' in r.stdout
        assert (tmp_path / "song.flac").read_bytes() == F.encode_flac_with_level(dref, sr, ch, level)
    r = subprocess.run([CLI, "-d", str(tmp_path / "song.glc"), "--flac-level", "9"], capture_output=True, text=True)
    assert r.returncode == 1 and "FLAC level must be 0-8" in r.stderr


# ---------------------------------------------------------------- the reference's tests/test_export.rs
def _export_roundtrip(tmp_path, clips, sr, ch, ext):
    """tests/test_export.rs:14-165: encode -> decode -> export (FLAC with the flac-export feature,
    WAV without) -> load_audio_file_lossless; plus the bytes against the oracle."""
    from oracle import oracle as O
    enc, dec = glc_amd.Encoder(sr), glc_amd.Decoder(ch, sr)
    decoded, want = [], []
    for x in clips:
        decoded.append(dec.decode(enc.encode(x, ch)))
        want.append(O.decode(O.encode(x, sr, ch).glc)[0])
    enc.close(); dec.close()
    allp, allw = np.concatenate(decoded), np.concatenate(want)
    assert np.array_equal(allp.view(np.uint32), allw.view(np.uint32))
    path = tmp_path / f"test_export.{ext}"
    (glc_amd.export_to_flac if ext == "flac" else glc_amd.export_to_wav)(path, allp, sr, ch)
    assert path.exists()
    loaded, rate, nch = glc_amd.load_audio_file_lossless(path)
    assert rate == sr and nch == ch and loaded.size == allp.size
    assert np.array_equal(loaded, F.to_i16(allw).astype(np.float32) / np.float32(32768.0))
    if ext == "flac":
        assert path.read_bytes() == F.encode_flac_with_level(allw, sr, ch, 5)


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["flac", "wav"])
def test_export_basic(tmp_path, ext):
    _export_roundtrip(tmp_path, [gen_tone("sine", 440.0, 44100, 2, 2.0)], 44100, 2, ext)


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["flac", "wav"])
def test_export_mono(tmp_path, ext):
    _export_roundtrip(tmp_path, [gen_tone("sine", 1000.0, 48000, 1, 1.5)], 48000, 1, ext)


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["flac", "wav"])
def test_export_gapless_playlist(tmp_path, ext):
    clips = [gen_tone("sine", f, 44100, 2, 1.0) for f in (440.0, 880.0, 1320.0)]
    _export_roundtrip(tmp_path, clips, 44100, 2, ext)


# ---------------------------------------------------------------- committed fixtures (tests/golden/make_golden_flac.py)
def test_flac_golden_fixtures():
    """The frozen oracle bytes: the product reproduces them, the oracle still does, and the
    product's decoder reads them back to the generating signal."""
    import json
    import importlib.util
    gdir = os.path.join(ROOT, "tests", "golden")
    spec = importlib.util.spec_from_file_location("make_golden_flac", os.path.join(gdir, "make_golden_flac.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    cases = json.load(open(os.path.join(gdir, "flac_golden.json")))
    assert len(cases) == len(mk.CASES)
    for c in cases:
        x = mk.signal(c["signal"])
        assert hashlib.sha256(x.tobytes()).hexdigest() == c["input_sha256"]
        want = open(os.path.join(gdir, c["file"]), "rb").read()
        assert len(want) == c["length"] and hashlib.sha256(want).hexdigest() == c["sha256"]
        assert glc_amd.encode_flac_with_level(x, c["sample_rate"], c["channels"], c["level"]) == want
        assert F.encode_flac_with_level(x, c["sample_rate"], c["channels"], c["level"]) == want
        y, sr, ch = glc_amd.load_flac(os.path.join(gdir, c["file"]))
        assert (sr, ch) == (c["sample_rate"], c["channels"])
        assert np.array_equal(y, F.to_i16(x).astype(np.float32) / np.float32(32768.0))
