"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  The bar is bit equality: MDCT coefficients (f32 bits), scale factors,
quantised integers, raw decisions, the .glc byte stream, and decoded PCM (f32 bits)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import glc_amd
from conftest import ROOT, calculate_snr, gen_chord, gen_noise, gen_tone, parse_glc
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU (no CPU fallback exists)"
    return torch


def device_encode(torch, x, sr, ch, f0=None, f1=None, t0=0, t_count=None, want_coeffs=True):
    """Run glc_encode_range_device on torch-owned device memory; returns (records, coeffs)."""
    plan = glc_amd.plan_encode(x.size, ch)
    f0 = 0 if f0 is None else f0
    f1 = plan.n_frames if f1 is None else f1
    L = -(-x.size // ch)
    t_count = L - t0 if t_count is None else t_count
    lo, hi = t0 * ch, min((t0 + t_count) * ch, x.size)
    d_pcm = torch.from_numpy(x[lo:hi].copy()).cuda()
    rec = glc_amd.lib.glc_record_bytes(ch)
    d_rec = torch.zeros((f1 - f0) * rec, dtype=torch.uint8, device="cuda")
    d_coef = torch.zeros(((f1 - f0) * ch, 1024), dtype=torch.float32, device="cuda") if want_coeffs else None
    enc = glc_amd.Encoder(sr)
    torch.cuda.synchronize()
    enc.encode_range_device(d_pcm.data_ptr(), t0, t_count, x.size, ch, f0, f1, d_rec.data_ptr(),
                            d_coef.data_ptr() if want_coeffs else 0)
    enc.synchronize()
    return d_rec.cpu().numpy(), (d_coef.cpu().numpy() if want_coeffs else None)


def split_records(recs, ch):
    hdr = ((8 + 8 * ch) + 15) // 16 * 16
    rec = hdr + 4096 * ch
    r = recs.reshape(-1, rec)
    is_raw = r[:, 0:4].copy().view(np.uint32)[:, 0]
    meta = r[:, 8:8 + 8 * ch].copy().view(np.uint32).reshape(-1, ch, 2)
    scale = meta[:, :, 0].copy().view(np.float32).reshape(-1)
    nnz = meta[:, :, 1].reshape(-1)
    payload = r[:, hdr:].copy().view(np.int16).reshape(-1, ch, 2048)
    return is_raw, scale, nnz, payload


def test_context_tables_equal_oracle(torch_cuda):
    enc = glc_amd.Encoder(48000)
    T, w, n, wt, edges = enc.tables()
    To, wo, no = O.tables()
    assert np.array_equal(bits(T), bits(To)) and np.array_equal(bits(w), bits(wo)) and n == no
    wo2, eo = O.perceptual(48000)
    assert np.array_equal(bits(wt), bits(wo2)) and np.array_equal(edges, eo)


CASES = [
    ("cfg1_sine_44k_stereo", lambda: gen_tone("sine", 440.0, 44100, 2, 2.0), 44100, 2),
    ("noise_44k_stereo", lambda: gen_noise(44100, 2, 0.5, 12345), 44100, 2),
    ("sweep_48k_mono", lambda: gen_tone("sweep", 100.0, 48000, 1, 1.0, 10000.0), 48000, 1),
    ("square_44k_mono", lambda: gen_tone("square", 1000.0, 44100, 1, 0.5), 44100, 1),
    ("chord_48k_stereo", lambda: gen_chord(48000, 2, 48000), 48000, 2),
    ("chord_96k_6ch", lambda: gen_chord(96000, 6, 20000), 96000, 6),
    ("chord_192k_8ch", lambda: gen_chord(192000, 8, 16000), 192000, 8),
    ("mixed_44k_3ch", lambda: np.concatenate([gen_chord(44100, 3, 9000), gen_noise(44100, 3, 0.2, 5)]), 44100, 3),
    ("ragged_stereo", lambda: gen_chord(44100, 2, 5000)[:-1], 44100, 2),
    ("quiet_48k_stereo", lambda: gen_chord(48000, 2, 6000, amp=1e-12), 48000, 2),
    ("silence_48k_stereo", lambda: np.zeros(2 * 5000, np.float32), 48000, 2),
    ("min_len_mono", lambda: gen_chord(48000, 1, 513), 48000, 1),
    ("denormal_stereo", lambda: _special("denormal"), 44100, 2),
    ("huge_and_clipping_mono", lambda: _special("huge"), 48000, 1),
    ("nan_inf_stereo", lambda: _special("nan"), 48000, 2),
    ("impulses_5ch", lambda: _special("impulse"), 96000, 5),
]


def _special(kind):
    """Inputs outside the nominal [-1, 1] PCM range: the GPU path must still agree bit for bit."""
    rng = np.random.RandomState(11)
    if kind == "denormal":   # f32 subnormals in, subnormal products and sums (denorm mode must be IEEE)
        x = gen_chord(44100, 2, 6000, amp=0.2)
        x[2000:9000] = (rng.uniform(-1, 1, 7000) * 1e-39).astype(np.float32)
        x[9000:9800] = np.float32(1e-45) * rng.randint(-3, 4, 800).astype(np.float32)
        return x
    if kind == "huge":       # far beyond full scale: quantiser clamps, raw plane saturates
        x = gen_chord(48000, 1, 7000, amp=0.3)
        x[1000:1400] *= np.float32(1e6)
        x[3000:3003] = np.float32([3.0e38, -3.0e38, 1.0e30])
        return x
    if kind == "nan":        # NaN / Inf propagate exactly like the reference's f32 arithmetic
        x = gen_chord(48000, 2, 5000, amp=0.2)
        x[1501] = np.nan
        x[4002] = np.inf
        x[7003] = -np.inf
        return x
    x = np.zeros(4000 * 5, np.float32)   # isolated unit impulses, one per channel
    for c in range(5):
        x[(700 + 411 * c) * 5 + c] = 1.0 if c % 2 == 0 else -1.0
    return x


@pytest.mark.parametrize("name,make,sr,ch", CASES, ids=[c[0] for c in CASES])
def test_encode_stages_bit_exact(torch_cuda, name, make, sr, ch):
    x = make()
    ref = O.encode(x, sr, ch, taps=True)
    recs, coef = device_encode(torch_cuda, x, sr, ch)
    # K1: MDCT coefficients, f32 bit equality (tolerance: 0 ulp)
    assert coef.shape == ref.coeffs.shape
    assert np.array_equal(bits(coef), bits(ref.coeffs)), \
        f"{(bits(coef) != bits(ref.coeffs)).sum()} coefficient words differ"
    # K2/K3: scale, nnz, raw decision, dense q / raw plane
    is_raw, scale, nnz, payload = split_records(recs, ch)
    assert np.array_equal(is_raw.astype(np.uint8), ref.is_raw)
    assert np.array_equal(bits(scale), bits(ref.scales))
    assert np.array_equal(nnz, ref.nnz)
    g = parse_glc(ref.glc)
    for f in range(ref.n_frames):
        for c in range(ch):
            if ref.is_raw[f]:
                assert np.array_equal(payload[f, c], g["frames"][f]["raw"][c * 2048:(c + 1) * 2048])
            else:
                assert np.array_equal(payload[f, c, :1024], ref.dense_q[f * ch + c])
    # assembled byte stream
    out = glc_amd.EncodedAudio.from_records(sr, x.size, ch, recs)
    assert out.to_bytes() == ref.glc


@pytest.mark.parametrize("name,make,sr,ch", CASES, ids=[c[0] for c in CASES])
def test_encode_decode_through_host_api(torch_cuda, name, make, sr, ch):
    x = make()
    ref = O.encode(x, sr, ch)
    enc = glc_amd.Encoder(sr).encode(x, ch)           # Encoder::encode
    assert enc.to_bytes() == ref.glc                  # .glc identical to the CPU reference path
    dec = glc_amd.Decoder(1, sr).decode(enc)          # Decoder::new ignores `channels` (Q4)
    dref, _, _ = O.decode(ref.glc)
    assert dec.size == x.size == dref.size            # tests/test_codec.rs length equality
    assert np.array_equal(bits(dec), bits(dref))      # decoded PCM, f32 bit equality
    chunks = list(glc_amd.Decoder(ch, sr).decode_streaming(enc))
    assert chunks[-1].is_last and not any(c.is_last for c in chunks[:-1])
    allv = np.concatenate([c.samples for c in chunks])
    assert allv.size == (ref.n_frames + 1) * 1024 * ch
    assert np.array_equal(bits(allv[512:512 + dec.size]), bits(dec))


def test_golden_fixtures(torch_cuda):
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        gold = json.load(fh)
    for case in gold["cases"]:
        g = case["generator"]
        x = gen_noise(case["sample_rate"], case["channels"], g["dur"], g["seed"]) if g["kind"] == "noise" \
            else gen_tone(g["kind"], g["f0"], case["sample_rate"], case["channels"], g["dur"], g.get("f1", 0.0))
        assert hashlib.sha256(x.tobytes()).hexdigest() == case["input_sha256"]
        enc = glc_amd.Encoder(case["sample_rate"]).encode(x, case["channels"])
        data = enc.to_bytes()
        with open(os.path.join(ROOT, "tests", "golden", case["glc_file"]), "rb") as fh:
            assert data == fh.read()
        dec = glc_amd.Decoder(case["channels"], case["sample_rate"]).decode(enc)
        assert hashlib.sha256(dec.tobytes()).hexdigest() == case["decoded_sha256"]


def test_shard_with_halo_equals_whole_stream(torch_cuda):
    """Frame-range shards (own PCM slice + halo) produce the same records as the whole stream."""
    sr, ch = 48000, 2
    x = np.concatenate([gen_chord(sr, ch, 30000), gen_noise(sr, ch, 0.2, 9), gen_chord(sr, ch, 9000, seed=3)])
    whole, _ = device_encode(torch_cuda, x, sr, ch, want_coeffs=False)
    plan = glc_amd.plan_encode(x.size, ch)
    rec = glc_amd.lib.glc_record_bytes(ch)
    for world in (2, 3, 5):
        parts = []
        for s in glc_amd.shard.plan_shards(plan.n_frames, plan.per_channel, world):
            r, _ = device_encode(torch_cuda, x, sr, ch, s.frame_begin, s.frame_end, s.t0, s.t_count, want_coeffs=False)
            assert r.size == s.n_frames * rec
            parts.append(r)
        assert np.array_equal(np.concatenate(parts), whole)
    # a shard without its halo is rejected, not silently zero-padded
    s = glc_amd.shard.plan_shards(plan.n_frames, plan.per_channel, 2)[1]
    with pytest.raises(glc_amd.GlcError):
        device_encode(torch_cuda, x, sr, ch, s.frame_begin, s.frame_end, s.t0 + 600, s.t_count - 600, want_coeffs=False)


def test_decode_of_noncanonical_lists(torch_cuda):
    """Hostile-but-well-formed streams: duplicate / descending / out-of-range indices follow the
    reference's dense-array semantics (last write wins, idx >= 1024 ignored, src/codec.rs:659-665)."""
    import struct
    sr, ch = 44100, 1
    body = b""
    lists = [[(5, 100), (3, -200), (5, 300), (2000, 77), (1023, -32768), (0, 0)], [], [(7, 1)]]
    for l in lists:
        body += struct.pack("<Q", 1) + struct.pack("<Q", len(l)) + b"".join(struct.pack("<Hh", i, q) for i, q in l)
        body += struct.pack("<Qf", 1, 0.25) + b"\x00"
    data = struct.pack("<IHQQ", sr, ch, 3000, len(lists)) + body + struct.pack("<IIQ", 512, 0, 3000)
    ref, _, _ = O.decode(data)
    enc = glc_amd.EncodedAudio.from_bytes(data)
    assert enc.to_bytes() == data
    dec = glc_amd.Decoder(ch, sr).decode(enc)
    assert np.array_equal(bits(dec), bits(ref))


def test_reference_properties_on_gpu_path(torch_cuda):
    """tests/test_codec.rs / test_comprehensive.rs style properties through the product API."""
    for kind, f, sr, ch, dur, bound in [("sine", 440.0, 44100, 1, 2.0, -10.0), ("square", 1000.0, 44100, 1, 2.0, -15.0),
                                        ("sawtooth", 440.0, 44100, 1, 2.0, -10.0), ("sine", 440.0, 48000, 2, 1.0, -10.0)]:
        x = gen_tone(kind, f, sr, ch, dur)
        enc = glc_amd.Encoder(sr).encode(x, ch)
        dec = glc_amd.Decoder(ch, sr).decode(enc)
        assert dec.size == x.size and calculate_snr(x, dec) > bound
    # gapless: three files decode to lengths that sum exactly (tests/test_codec.rs:140-170)
    total = 0
    for f in (440.0, 550.0, 660.0):
        x = gen_tone("sine", f, 44100, 2, 0.5)
        enc = glc_amd.Encoder(44100).encode(x, 2)
        total += glc_amd.Decoder(2, 44100).decode(enc).size
    assert total == 3 * gen_tone("sine", 440.0, 44100, 2, 0.5).size


def test_degenerate_inputs_return_errors(torch_cuda):
    e = glc_amd.Encoder(44100)
    for x, ch in [(np.zeros(512, np.float32), 1), (np.zeros(1024, np.float32), 2), (np.zeros(100, np.float32), 0)]:
        with pytest.raises(glc_amd.GlcError) as err:
            e.encode(x, ch)
        assert err.value.code == -1


def test_cfg2_full_size_properties(torch_cuda):
    """BASELINE config 2 at full size (4096 frames x 1024, 48 kHz stereo): the oracle is too slow
    for all of it, so check (a) a 64-frame window bit-exactly against the oracle run on that
    window's own PCM slice, and (b) size-independent properties over the whole batch."""
    sr, ch = 48000, 2
    L = 4096 * 1024
    x = gen_chord(sr, ch, L)
    recs, _ = device_encode(torch_cuda, x, sr, ch, want_coeffs=False)
    is_raw, scale, nnz, payload = split_records(recs, ch)
    assert is_raw.size == 4096 and not is_raw.any()
    assert (nnz > 0).all() and (nnz < 512).all()
    # every kept coefficient is non-zero, nnz equals the count, the largest |q| is 32767/32768
    q = payload[:, :, :1024]
    assert np.array_equal((q != 0).sum(axis=2).reshape(-1), nnz)
    assert (np.abs(q.astype(np.int32)).max(axis=2) >= 32767).all()
    # (a) a sub-stream starting at per-channel sample 1024*f0 has its frame j equal to frame
    # f0 + j of the big stream for every j >= 1 that does not touch the sub-stream's end
    f0, nsub = 1000, 66
    sub = x[1024 * f0 * ch:(1024 * (f0 + nsub) + 512) * ch]
    ref = O.encode(sub, sr, ch, taps=True)
    assert ref.n_frames == nsub
    for j in range(1, nsub - 1):
        for c in range(ch):
            mb, ms = (f0 + j) * ch + c, j * ch + c
            assert scale[mb].view(np.uint32) == ref.scales[ms].view(np.uint32)
            assert nnz[mb] == ref.nnz[ms]
            assert np.array_equal(q[f0 + j, c], ref.dense_q[ms])
    # (b) encode -> decode -> length and SNR on the whole batch through the host API
    enc = glc_amd.EncodedAudio.from_records(sr, x.size, ch, recs)
    dec = glc_amd.Decoder(ch, sr).decode(enc)
    assert dec.size == x.size
    assert calculate_snr(x[:400000], dec[:400000]) > -10.0


# ---------------------------------------------------------------------------------------
# BASELINE configs 3-5 at full size.  The oracle is too slow for whole streams, so each test
# pins a prefix / window bit-exactly against the oracle and checks size-independent properties
# (exact length, periodicity of a tiled input, chunk-boundary continuity) on the whole stream.
# ---------------------------------------------------------------------------------------

def _tile(seg, ch, reps):
    return np.tile(seg.reshape(-1, ch), (reps, 1)).reshape(-1)


def test_cfg3_ten_minutes_roundtrip(torch_cuda):
    """config 3: encode+decode roundtrip, 10 min 48 kHz stereo (overlap-add gapless check)."""
    sr, ch = 48000, 2
    seg = gen_chord(sr, ch, 480000)              # 10 s
    x = _tile(seg, ch, 60)                       # 10 min = 57 600 000 samples
    assert x.size == 57_600_000 and glc_amd.plan_encode(x.size, ch).n_frames == 28125
    enc = glc_amd.Encoder(sr).encode(x, ch)
    info = enc.info()
    assert info.n_frames == 28125 and info.n_raw_frames == 0
    dec = glc_amd.Decoder(ch, sr).decode(enc)
    assert dec.size == x.size                    # gapless length equality
    # prefix pinned against the oracle: the first 10 s of the stream
    n_pref = 400 * 1024 * ch
    ref = O.encode(x[:n_pref + 2048 * ch], sr, ch, taps=True)
    for f in range(0, 398, 37):
        a, b = enc.frames[f], parse_glc(ref.glc)["frames"][f]
        assert np.array_equal(np.float32(a.scale_factors).view(np.uint32), b["scales"].view(np.uint32))
        for c in range(ch):
            assert a.sparse_coeffs_per_channel[c] == list(zip(b["lists"][c][0].tolist(), b["lists"][c][1].tolist()))
    dref, _, _ = O.decode(ref.glc)
    assert np.array_equal(bits(dec[:390 * 1024 * ch]), bits(dref[:390 * 1024 * ch]))
    # overlap-add continuity across the library's 4096-frame decode chunks and 500-frame
    # streaming chunks: decoding a window that straddles the boundary alone gives the same bits
    assert calculate_snr(x[:2_000_000], dec[:2_000_000]) > -10.0
    for boundary in (4096, 8192, 500, 24576):
        lo, hi = boundary - 3, boundary + 3
        sub = x[lo * 1024 * ch:(hi * 1024 + 512) * ch]
        sub_dec = glc_amd.Decoder(ch, sr).decode(glc_amd.Encoder(sr).encode(sub, ch))
        # sub-stream hop j (j >= 2) equals hop lo + j of the big stream; account for the 512-sample
        # interleaved delay trim (Q3) on both sides
        a = dec[((lo + 2) * 1024) * ch:((hi - 1) * 1024) * ch]
        b = sub_dec[(2 * 1024) * ch:((hi - lo - 1) * 1024) * ch]
        assert np.array_equal(bits(a), bits(b))


def _blocks_of_frame(fr, ch):
    """Windowed IMDCT blocks [ch][2048] of one EncodedFrame, by the oracle's imdct_block
    (src/codec.rs:626-675): what the overlap-add consumes."""
    _, w, _ = O.tables()
    out = np.zeros((ch, 2048), np.float32)
    for c in range(ch):
        if fr.raw_pcm is not None:
            idx = np.arange(2048) * ch + c
            ok = idx < fr.raw_pcm.size
            out[c, ok] = fr.raw_pcm[idx[ok]].astype(np.float32) / np.float32(32767.0)
            continue
        coeffs = np.zeros(1024, np.float32)
        scale = np.maximum(np.float32(fr.scale_factors[c]), np.float32(1e-12))
        for k, q in fr.sparse_coeffs_per_channel[c]:
            if k < 1024:
                coeffs[k] = (np.float32(q) / np.float32(32768.0)) * scale
        out[c] = O.imdct_block(coeffs) * w
    return out


def test_cfg4_one_hour_96k_stereo_true_size(torch_cuda):
    """BASELINE config 4, one rank's share AT TRUE SIZE: 1 h of 96 kHz stereo = 691 200 000 samples,
    337 500 frames (2.76 GB of f32 PCM, 2.77 GB of records).  The stream is a 60 s segment tiled 60
    times; 60 s x 96 kHz = 5625 frames exactly, so records and decoded PCM repeat with period 5625
    frames - a size-independent property checked over the whole hour - and a 64-frame window near
    frame 337 000 (per-channel sample offsets beyond 2^28, byte offsets beyond 2 GiB: 64-bit index
    arithmetic and the buffer-descriptor rebasing of K1) is pinned against the oracle, for the
    encoder's records and for the decoded PCM."""
    sr, ch = 96000, 2
    seg = gen_chord(sr, ch, 60 * sr, n_tones=8)
    reps = 60
    x = _tile(seg, ch, reps)
    assert x.size == 691_200_000
    plan = glc_amd.plan_encode(x.size, ch)
    nf = plan.n_frames
    assert nf == 337_500 == 5625 * reps
    rec = glc_amd.lib.glc_record_bytes(ch)
    d_pcm = torch_cuda.from_numpy(x).cuda()
    d_rec = torch_cuda.zeros(nf * rec, dtype=torch_cuda.uint8, device="cuda")
    enc = glc_amd.Encoder(sr)
    torch_cuda.cuda.synchronize()
    enc.encode_range_device(d_pcm.data_ptr(), 0, plan.per_channel, x.size, ch, 0, nf, d_rec.data_ptr())
    enc.synchronize()
    r = d_rec.view(nf, rec)
    for k in range(1, reps):   # periodicity over the whole hour, compared on the device
        assert torch_cuda.equal(r[1:5624], r[k * 5625 + 1:k * 5625 + 5624]), f"period {k}"
    # far-end window pinned against the oracle, computed from that window's own PCM slice
    f0, f1 = 337_000, 337_064
    t0, t1 = f0 * 1024 - 512, min(plan.per_channel, (f1 - 1) * 1024 - 512 + 2048)
    assert t0 * ch * 4 > 2 ** 31
    want, _ = O.encode_range_records(x[t0 * ch:t1 * ch], t0, t1 - t0, x.size, sr, ch, f0, f1)
    got = r[f0:f1].reshape(-1).cpu().numpy()
    assert np.array_equal(got, want)
    # and the stream's last frames (zero padding past the end of a > 2^31-element buffer)
    want, _ = O.encode_range_records(x[(nf - 9) * 1024 * ch:], (nf - 9) * 1024, plan.per_channel - (nf - 9) * 1024,
                                     x.size, sr, ch, nf - 8, nf)
    assert np.array_equal(r[nf - 8:nf].reshape(-1).cpu().numpy(), want)
    # EncodedAudio through the device compaction (scan offsets beyond 2^32 bytes of records)
    ea = enc.frames_from_device_records(d_rec.data_ptr(), nf, x.size, ch)
    info = ea.info()
    assert info.n_frames == nf and info.n_raw_frames == 0 and info.total_nnz > nf
    del d_rec, r, d_pcm
    # decode: exact length, periodicity, and the far-end window against blocks from the oracle's IMDCT
    dec = glc_amd.Decoder(ch, sr).decode(ea)
    assert dec.size == x.size
    per = 5625 * 1024 * ch
    for k in (1, 7, 31, 57):   # (the 60th period ends in the stream's tail)
        assert np.array_equal(bits(dec[per:2 * per]), bits(dec[(k + 1) * per:(k + 2) * per])), f"decoded period {k}"
    blocks = {f: _blocks_of_frame(ea.frames[f], ch) for f in range(f0, f0 + 13)}
    for h in range(f0 + 1, f0 + 13):   # hop h = second half of frame h-1 + first half of frame h (:688-705)
        hop = (blocks[h - 1][:, 1024:] + blocks[h][:, :1024]).T.reshape(-1)      # interleaved
        a = h * 1024 * ch - 512                                                   # delay trim in interleaved units (Q3)
        assert np.array_equal(bits(dec[a:a + 1024 * ch]), bits(hop)), f"decoded hop {h}"
    assert calculate_snr(x[:400000], dec[:400000]) > -10.0


def test_multi_gpu_host_tool(torch_cuda):
    """tools/glc_multi_gpu.cpp: one process, one context per visible device, RCCL gather of the
    compact blobs to device 0, bytes compared with the single-device encode inside the tool (with one
    visible device the collective is empty).  Both shardings of SURVEY 8e."""
    import subprocess
    exe = os.path.join(ROOT, "build", "glc_multi_gpu")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build_tools()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for args in (["--frames", "700"], ["--streams", "--frames", "300"]):
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "OK: every assembled stream is byte-identical" in p.stdout
        if "--streams" not in args:   # the sharded decode leg runs in frames mode
            assert "gathered PCM is bit-identical to the single-device decode" in p.stdout


def test_cfg5_eight_channel_192k(torch_cuda):
    """config 5: 7.1-channel 192 kHz high-res (8 channels): channel x frame tiling stress."""
    sr, ch = 192000, 8
    seg = gen_chord(sr, ch, 192000, n_tones=6)   # 1 s
    x = _tile(seg, ch, 20)                       # 20 s = 30 720 000 samples, 3750 frames x 8 ch
    plan = glc_amd.plan_encode(x.size, ch)
    assert plan.n_frames == 3750
    enc = glc_amd.Encoder(sr).encode(x, ch)
    assert enc.info().n_frames == 3750
    ref = O.encode(x[:(40 * 1024 + 512) * ch], sr, ch)
    g = parse_glc(ref.glc)
    for f in range(0, 38, 5):
        a, b = enc.frames[f], g["frames"][f]
        assert np.array_equal(np.float32(a.scale_factors).view(np.uint32), b["scales"].view(np.uint32))
        for c in range(ch):
            assert a.sparse_coeffs_per_channel[c] == list(zip(b["lists"][c][0].tolist(), b["lists"][c][1].tolist()))
    dec = glc_amd.Decoder(ch, sr).decode(enc)
    assert dec.size == x.size
    dref, _, _ = O.decode(ref.glc)
    assert np.array_equal(bits(dec[:30 * 1024 * ch]), bits(dref[:30 * 1024 * ch]))


def test_device_side_compaction_equals_host_assembly(torch_cuda):
    """glc_frames_from_device_records (scan + ballot pack on the GPU) == glc_frames_from_records
    (host scan of the dense rows), on a stream mixing compressed and raw frames."""
    sr, ch = 44100, 3
    x = np.concatenate([gen_chord(sr, ch, 9000), gen_noise(sr, ch, 0.2, 5), gen_chord(sr, ch, 3000, seed=2),
                        gen_noise(sr, ch, 0.05, 6)])
    plan = glc_amd.plan_encode(x.size, ch)
    rec = glc_amd.lib.glc_record_bytes(ch)
    d_pcm = torch_cuda.from_numpy(x).cuda()
    d_rec = torch_cuda.zeros(plan.n_frames * rec, dtype=torch_cuda.uint8, device="cuda")
    enc = glc_amd.Encoder(sr)
    torch_cuda.cuda.synchronize()
    enc.encode_range_device(d_pcm.data_ptr(), 0, x.size // ch, x.size, ch, 0, plan.n_frames, d_rec.data_ptr())
    enc.synchronize()
    a = enc.frames_from_device_records(d_rec.data_ptr(), plan.n_frames, x.size, ch)
    b = glc_amd.EncodedAudio.from_records(sr, x.size, ch, d_rec.cpu().numpy())
    assert 0 < a.info().n_raw_frames < plan.n_frames
    assert a.to_bytes() == b.to_bytes() == O.encode(x, sr, ch).glc


@pytest.mark.parametrize("n_frames", [499, 500, 501, 1000, 1203])
def test_streaming_chunk_boundaries(torch_cuda, n_frames):
    """decode_streaming chunking (src/codec.rs:708-732): a chunk per 500 frames, the remainder plus
    the overlap tail in the last chunk; each chunk decoded on demand, all bits equal to decode()."""
    sr, ch = 48000, 2
    x = gen_chord(sr, ch, n_frames * 1024, n_tones=4)
    assert glc_amd.plan_encode(x.size, ch).n_frames == n_frames
    enc = glc_amd.Encoder(sr).encode(x, ch)
    dec = glc_amd.Decoder(ch, sr)
    whole = dec.decode(enc)
    chunks = list(dec.decode_streaming(enc))
    sizes = [c.samples.size // (1024 * ch) for c in chunks]
    full, rem = divmod(n_frames, 500)
    assert sizes == [500] * full + [rem + 1]
    assert [c.is_last for c in chunks] == [False] * full + [True]
    allv = np.concatenate([c.samples for c in chunks])
    assert np.array_equal(bits(allv[512:512 + whole.size]), bits(whole))
    # a second decode on the same Decoder (fresh session) gives the same bits
    assert np.array_equal(bits(dec.decode(enc)), bits(whole))


def test_decode_degenerate_streams(torch_cuda):
    """Streams no encoder emits but the container allows: zero frames; more channel vectors than
    header.channels (extra ones ignored, src/codec.rs:648-653); original_length / delay larger than
    the decoded stream."""
    import struct
    for sr, ch, frames, delay, orig in [(48000, 2, [], 512, 10), (44100, 1, [], 5000, 7),
                                        (44100, 1, [[[(3, 1000)], [(9, -7)]]], 0, 10 ** 9)]:
        body = b""
        for lists in frames:
            body += struct.pack("<Q", len(lists))
            for l in lists:
                body += struct.pack("<Q", len(l)) + b"".join(struct.pack("<Hh", i, q) for i, q in l)
            body += struct.pack("<Q", len(lists)) + b"".join(struct.pack("<f", 0.5) for _ in lists) + b"\x00"
        data = struct.pack("<IHQQ", sr, ch, orig, len(frames)) + body + struct.pack("<IIQ", delay, 0, orig)
        ref, _, _ = O.decode(data)
        dec = glc_amd.Decoder(ch, sr).decode(glc_amd.EncodedAudio.from_bytes(data))
        assert dec.size == ref.size and np.array_equal(bits(dec), bits(ref))
    # fewer channel vectors than header.channels: the reference panics; here GLC_EFORMAT
    data = struct.pack("<IHQQ", 44100, 2, 100, 1) + struct.pack("<QQ", 1, 0) + struct.pack("<Qf", 1, 0.5) + b"\x00" \
        + struct.pack("<IIQ", 512, 0, 100)
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.Decoder(2, 44100).decode(glc_amd.EncodedAudio.from_bytes(data))
    assert e.value.code == -4


def test_device_resident_decode(torch_cuda):
    """glc_decode_device writes the un-trimmed stream into caller-owned device memory; its
    trimmed window equals Decoder::decode bit for bit (multi-chunk: > 4096 frames)."""
    sr, ch = 48000, 2
    x = np.concatenate([gen_chord(sr, ch, 4500 * 1024, n_tones=5), gen_noise(sr, ch, 0.3, 4)])
    enc = glc_amd.Encoder(sr).encode(x, ch)
    nf = enc.info().n_frames
    assert nf > 4096 and 0 < enc.info().n_raw_frames < nf
    dec = glc_amd.Decoder(ch, sr)
    host = dec.decode(enc)
    d_all = torch_cuda.empty((nf + 1) * 1024 * ch, dtype=torch_cuda.float32, device="cuda")
    torch_cuda.cuda.synchronize()
    start, n = dec.decode_device(enc, d_all.data_ptr(), d_all.numel())
    dec.synchronize()
    assert (start, n) == (512, x.size)
    assert np.array_equal(bits(d_all.cpu().numpy()[start:start + n]), bits(host))
    with pytest.raises(glc_amd.GlcError):
        dec.decode_device(enc, d_all.data_ptr(), d_all.numel() - 1)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_decode_equals_whole_stream(torch_cuda, world):
    """SURVEY 8e, decode side: disjoint hop ranges decoded independently (each recomputes one
    halo frame for the overlap-add) concatenate to the whole-stream output bit for bit - across
    raw frames, the 4096-frame round boundary, the stream start (+0.0 overlap) and the bare tail."""
    sr, ch = 48000, 2
    x = np.concatenate([gen_chord(sr, ch, 4200 * 1024, n_tones=5), gen_noise(sr, ch, 0.2, 4),
                        gen_chord(sr, ch, 100 * 1024, n_tones=3, seed=9)])
    enc = glc_amd.Encoder(sr).encode(x, ch)
    nf = enc.info().n_frames
    dec = glc_amd.Decoder(ch, sr)
    whole = torch_cuda.empty((nf + 1) * 1024 * ch, dtype=torch_cuda.float32, device="cuda")
    dec.decode_device(enc, whole.data_ptr(), whole.numel())
    dec.synchronize()
    want = whole.cpu().numpy()
    parts = []
    for r in glc_amd.shard.hop_ranges(nf, world):
        d = torch_cuda.full((max(len(r), 1) * 1024 * ch,), float("nan"), dtype=torch_cuda.float32, device="cuda")
        torch_cuda.cuda.synchronize()
        dec.decode_range_device(enc, r.start, r.stop, d.data_ptr(), d.numel())
        dec.synchronize()
        parts.append(d.cpu().numpy()[:len(r) * 1024 * ch])
    got = np.concatenate(parts)
    assert got.size == want.size and np.array_equal(bits(got), bits(want))
    # odd cuts: a single hop in the middle, the tail alone, an empty range
    for a, b in ((nf // 2, nf // 2 + 1), (nf, nf + 1), (7, 7), (4095, 4098), (0, 1)):
        d = torch_cuda.empty(max(b - a, 1) * 1024 * ch, dtype=torch_cuda.float32, device="cuda")
        torch_cuda.cuda.synchronize()
        dec.decode_range_device(enc, a, b, d.data_ptr(), d.numel())
        dec.synchronize()
        assert np.array_equal(bits(d.cpu().numpy()[:(b - a) * 1024 * ch]), bits(want[a * 1024 * ch:b * 1024 * ch]))
    with pytest.raises(glc_amd.GlcError):
        dec.decode_range_device(enc, 0, nf + 2, whole.data_ptr(), whole.numel())
    with pytest.raises(glc_amd.GlcError):
        dec.decode_range_device(enc, 0, 10, whole.data_ptr(), 10 * 1024 * ch - 1)


def test_progress_messages(torch_cuda):
    """Progress (src/codec.rs:71-79) as the streaming decoder sends it (:609, :713, :736)."""
    sr, ch = 44100, 1
    x = gen_chord(sr, ch, 1100 * 1024, n_tones=3)
    enc = glc_amd.Encoder(sr).encode(x, ch)
    seen = []
    dec = glc_amd.Decoder(ch, sr)
    out = dec.decode(enc, lambda kind, value: seen.append((kind, value)))
    assert np.array_equal(bits(out), bits(dec.decode(enc)))
    kinds = [k for k, _ in seen]
    assert kinds[0] == "Status" and kinds[-1] == "Complete" and kinds.count("Decoding") == 2
    assert seen[0][1] == "Starting streaming decode of 1100 frames"
    assert all(0.0 < v <= 100.0 for k, v in seen if k == "Decoding")


@pytest.mark.parametrize("ch", [7, 16, 33])
def test_unusual_channel_counts(torch_cuda, ch):
    """Row = frame * ch + c for any channel count (tiles start mid-frame when ch does not divide 128)."""
    sr = 48000
    x = gen_chord(sr, ch, 3000 + 17 * ch, n_tones=3)
    ref = O.encode(x, sr, ch)
    enc = glc_amd.Encoder(sr).encode(x, ch)
    assert enc.to_bytes() == ref.glc
    dec = glc_amd.Decoder(ch, sr).decode(enc)
    dref, _, _ = O.decode(ref.glc)
    assert np.array_equal(bits(dec), bits(dref))


def test_fuzz_random_configurations(torch_cuda):
    """40 seeded random configurations (sample rate, channels, length incl. ragged, content mix,
    amplitude from 1e-6 to clipping): `.glc` bytes and decoded PCM bits equal the oracle's."""
    rng = np.random.default_rng(20260104)
    rates = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 12345]
    encs, decs = {}, {}
    for case in range(40):
        sr = int(rng.choice(rates))
        ch = int(rng.choice([1, 1, 2, 2, 2, 3, 5, 6, 8]))
        n_per = int(rng.integers(513, 40000))
        n = n_per * ch - int(rng.integers(0, ch))                  # sometimes ragged
        if (n + ch - 1) // ch <= 512:
            n = 513 * ch
        kind = int(rng.integers(0, 5))
        t = np.arange(n_per + 1, dtype=np.float64)[:, None]
        if kind == 0:      # tones
            f = rng.uniform(30, sr / 2.2, (1, ch))
            x = np.sin(2 * np.pi * f * t / sr) * 0.5
        elif kind == 1:    # noise (raw path)
            x = rng.standard_normal((n_per + 1, ch)) * 0.3
        elif kind == 2:    # tone, then noise, then silence
            x = np.sin(2 * np.pi * 440.0 * t / sr) * np.ones((1, ch))
            x[n_per // 3:2 * n_per // 3] = rng.standard_normal((2 * n_per // 3 - n_per // 3, ch)) * 0.2
            x[2 * n_per // 3:] = 0.0
        elif kind == 3:    # sparse clicks
            x = np.zeros((n_per + 1, ch))
            x[rng.integers(0, n_per, 40), rng.integers(0, ch, 40)] = rng.uniform(-1, 1, 40)
        else:              # chirp with a DC offset
            x = np.sin(2 * np.pi * (50.0 + 0.2 * t) * t / sr) * 0.4 + 0.1
        amp = float(rng.choice([1e-6, 1e-3, 0.3, 1.0, 3.0]))
        x = (x * amp).astype(np.float32).reshape(-1)[:n]
        ref = O.encode(x, sr, ch)
        enc_ctx = encs.setdefault(sr, glc_amd.Encoder(sr))
        dec_ctx = decs.setdefault(sr, glc_amd.Decoder(ch, sr))
        enc = enc_ctx.encode(x, ch)
        assert enc.to_bytes() == ref.glc, (case, sr, ch, n, kind, amp)
        dref, _, _ = O.decode(ref.glc)
        assert np.array_equal(bits(dec_ctx.decode(enc)), bits(dref)), (case, sr, ch, n, kind, amp)
    for c in list(encs.values()) + list(decs.values()):
        c.close()


def test_distinct_contexts_run_concurrently(torch_cuda):
    """`&mut self` makes one Encoder / Decoder single-threaded, but distinct instances may be used
    from different threads at once (rayon callers): four host threads, each with its own contexts
    and its own input, produce the oracle's bytes while running interleaved on one device."""
    import threading
    sr = 44100
    inputs = [(gen_tone("sine", 300.0 + 100 * i, sr, 1 + i % 2, 0.6 + 0.1 * i), 1 + i % 2) for i in range(4)]
    refs = [O.encode(x, sr, ch) for x, ch in inputs]
    drefs = [O.decode(r.glc)[0] for r in refs]
    errors = []

    def work(i):
        try:
            x, ch = inputs[i]
            enc, dec = glc_amd.Encoder(sr), glc_amd.Decoder(ch, sr)
            for _ in range(8):
                ea = enc.encode(x, ch)
                assert ea.to_bytes() == refs[i].glc
                assert np.array_equal(bits(dec.decode(ea)), bits(drefs[i]))
                chunks = [c.samples for c in dec.decode_streaming(ea)]
                assert np.array_equal(bits(np.concatenate(chunks)[512:512 + x.size]), bits(drefs[i]))
            enc.close(); dec.close()
        except BaseException as e:  # noqa: BLE001 - reported to the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("ch", [1, 2, 4, 8, 3])
def test_long_ragged_streams_and_shards(torch_cuda, ch):
    """Launches of >= 4096 rows take the LDS-DMA transform, with the dwordx4 segment loader when
    the channel count is 1 / 2 / 4 / 8 and the per-row loader otherwise (3): ragged end (the last
    sample frame is incomplete, the last frames run into the zero padding - partially out-of-range
    dwordx4 loads), stream start, and shards whose buffer begins mid-stream, all against the oracle."""
    sr = 48000
    frames = 4096 // ch + 37
    n = frames * 1024 * ch - 300 * ch - (ch - 1)                   # ragged: not a whole sample frame
    rng = np.random.default_rng(ch)
    t = np.arange(frames * 1024, dtype=np.float64)[:, None]
    x = (np.sin(2 * np.pi * rng.uniform(60, 9000, (1, ch)) * t / sr) * 0.4).astype(np.float32)
    x[5000:9000] = rng.standard_normal((4000, ch)).astype(np.float32) * 0.3     # some raw frames
    x[-3000:] += rng.standard_normal((3000, ch)).astype(np.float32) * 0.2       # activity in the last frames
    x = x.reshape(-1)[:n]
    ref = O.encode(x, sr, ch)
    enc = glc_amd.Encoder(sr).encode(x, ch)
    assert enc.info().n_frames * ch >= 4096
    assert enc.to_bytes() == ref.glc
    whole, _ = device_encode(torch_cuda, x, sr, ch, want_coeffs=False)
    plan = glc_amd.plan_encode(x.size, ch)
    rec = glc_amd.lib.glc_record_bytes(ch)
    # two shards, the first one large enough for the DMA transform; the second starts mid-stream
    cut = plan.n_frames - 20
    lo_hi = [(0, cut), (cut, plan.n_frames)]
    parts = []
    for f0, f1 in lo_hi:
        t0 = max(0, f0 * 1024 - 512)
        t1 = min(plan.per_channel, (f1 - 1) * 1024 - 512 + 2048)
        r, _ = device_encode(torch_cuda, x, sr, ch, f0, f1, t0, t1 - t0, want_coeffs=False)
        assert r.size == (f1 - f0) * rec
        parts.append(r)
    assert np.array_equal(np.concatenate(parts), whole)
    # and a big shard that itself starts mid-stream (buffer base shifted, halo in front)
    f0, f1 = 9, plan.n_frames
    t0 = f0 * 1024 - 512
    r, _ = device_encode(torch_cuda, x, sr, ch, f0, f1, t0, plan.per_channel - t0, want_coeffs=False)
    assert np.array_equal(r, whole[f0 * rec:])
    # device PCM that is only 4-byte aligned (the dwordx4 loads must not assume more)
    d = torch_cuda.zeros(x.size + 1, dtype=torch_cuda.float32, device="cuda")
    d[1:] = torch_cuda.from_numpy(x).cuda()
    d_rec = torch_cuda.zeros(plan.n_frames * rec, dtype=torch_cuda.uint8, device="cuda")
    e = glc_amd.Encoder(sr)
    torch_cuda.cuda.synchronize()
    e.encode_range_device(d.data_ptr() + 4, 0, plan.per_channel, x.size, ch, 0, plan.n_frames, d_rec.data_ptr(), 0)
    e.synchronize()
    assert np.array_equal(d_rec.cpu().numpy(), whole)


# ---------------------------------------------------------------------------------------
# Direct coefficient-bit parity of the hand-scheduled transforms: the launches the
# benchmark and long streams actually take - k_mdct_fwd_small<4> at the top of its range (up to 3583 /
# 4095 rows), k_mdct_fwd_st in both workgroup shapes and round 3's k_mdct_fwd_dma (ch 1 / 2 / 4 / 8 =
# dwordx4 segment loader, 3 = per-row loader) - compared f32 bit for f32 bit with the oracle's mdct_block on windows of
# frames: the stream start (leading zero padding), a noise burst, the ragged end (partially
# out-of-range loads), and the same through a shard whose device buffer starts mid-stream.
# Reference: src/codec.rs:476-485, :359-374.
# ---------------------------------------------------------------------------------------

def _k1_stream(ch, frames, seed):
    sr = 48000
    rng = np.random.default_rng(seed)
    t = np.arange(frames * 1024, dtype=np.float64)[:, None]
    x = (np.sin(2 * np.pi * rng.uniform(60, 9000, (1, ch)) * t / sr) * 0.4).astype(np.float32)
    mid = (frames // 2) * 1024
    x[mid:mid + 6000] = rng.standard_normal((6000, ch)).astype(np.float32) * 0.3   # noise burst
    x[-3000:] += rng.standard_normal((3000, ch)).astype(np.float32) * 0.2          # activity in the last frames
    n = frames * 1024 * ch - 300 * ch - (ch - 1)                                    # ragged: not a whole sample frame
    return x.reshape(-1)[:n], sr


def _set_mdct_variant(ctx, v):
    glc_amd.lib.glc_debug_set_mdct_variant.restype = C.c_int
    glc_amd.lib.glc_debug_set_mdct_variant.argtypes = [C.c_void_p, C.c_int]
    assert glc_amd.lib.glc_debug_set_mdct_variant(ctx._h, v) == 0


# which kernel a launch reaches: by its row count (small4: the 2 x 4 short-clip kernel, up to 3583 rows - 4095
# when the channel count has no segment loader; low: the first row counts of k_mdct_fwd_st), or pinned through
# include/glc_debug.h (1 = dma: round 3's first kernel, 2 / 3 = k_mdct_fwd_st with 8 / 16 waves per workgroup);
# "shipped" = the dispatch itself at 8192 + rows (16 waves: the last round of row tiles is more than half full)
_K1_KERNELS = {"small4": (2300, 0), "small4-top": (3500, 0), "low": (3584 + 24, 0), "dma": (4096 + 333, 1),
               "st8": (4096 + 333, 2), "st16": (4096 + 333, 3), "shipped": (8192 + 4400, 0)}
_K1_BY_ROWS = ("small4", "small4-top", "low")


@pytest.mark.parametrize("ch", [1, 2, 4, 8, 3])
@pytest.mark.parametrize("kernel", list(_K1_KERNELS))
def test_k1_large_launch_coefficient_bits(torch_cuda, ch, kernel):
    rows_wanted, variant = _K1_KERNELS[kernel]
    frames = -(-rows_wanted // ch) + 3
    x, sr = _k1_stream(ch, frames, 100 + ch)
    plan = glc_amd.plan_encode(x.size, ch)
    nf, L = plan.n_frames, plan.per_channel
    enc = glc_amd.Encoder(sr)
    _set_mdct_variant(enc, variant)
    W = 5  # frames per oracle window

    def launch_and_check(f0, t0, t_count, windows):
        rows = (nf - f0) * ch
        assert rows <= 4095 if kernel in _K1_BY_ROWS else rows >= 4096
        lo, hi = t0 * ch, min((t0 + t_count) * ch, x.size)
        d_pcm = torch_cuda.from_numpy(x[lo:hi].copy()).cuda()
        d_coef = torch_cuda.full((rows, 1024), float("nan"), dtype=torch_cuda.float32, device="cuda")
        torch_cuda.cuda.synchronize()
        enc.mdct_forward_device(d_pcm.data_ptr(), t0, t_count, x.size, ch, f0, nf, d_coef.data_ptr())
        enc.synchronize()
        coef = d_coef.cpu().numpy()
        assert not np.isnan(coef).any()
        for a in windows:
            b = min(a + W, nf)
            _, ref = O.encode_range_records(x[lo:hi], t0, t_count, x.size, sr, ch, a, b, taps=True)
            got = coef[(a - f0) * ch:(b - f0) * ch]
            assert np.array_equal(bits(got), bits(ref.coeffs)), \
                f"{kernel} ch={ch}: {(bits(got) != bits(ref.coeffs)).sum()} coefficient words differ in frames [{a},{b})"
        if kernel not in _K1_BY_ROWS:  # every row of the launch: the other kernels for this size give the same words
            for other in (1, 2, 3):
                if other == variant:
                    continue
                _set_mdct_variant(enc, other)
                d_coef.fill_(float("nan"))
                torch_cuda.cuda.synchronize()
                enc.mdct_forward_device(d_pcm.data_ptr(), t0, t_count, x.size, ch, f0, nf, d_coef.data_ptr())
                enc.synchronize()
                assert np.array_equal(bits(d_coef.cpu().numpy()), bits(coef)), f"{kernel} ch={ch}: variant {other} differs"
            _set_mdct_variant(enc, variant)

    # whole stream on the device: start, noise burst, tile boundaries (rows 128k, 256k), ragged end
    launch_and_check(0, 0, L, [0, nf // 2 - 2, (128 * 3) // ch, (256 * 5) // ch - 2, nf - W])
    # a shard: frames [f0, nf) from a buffer that holds only [1024 f0 - 512, L)
    f0 = 2 if kernel in _K1_BY_ROWS else 3
    if (nf - f0) * ch >= (1793 if kernel in _K1_BY_ROWS else 4096):
        launch_and_check(f0, f0 * 1024 - 512, L - (f0 * 1024 - 512), [f0, nf // 2 - 1, nf - W])


def test_k1_round_kernel_of_the_host_pipeline(torch_cuda):
    """The kernel glc_encode gives its opening rounds (k_mdct_fwd_sched: launches of 1793..2048 rows that run
    beside each other), pinned through include/glc_debug.h variant 4: every coefficient word of a 2048-row
    and of a ragged 1800-row launch equals the 2 x 4 kernel's, windows of it equal the oracle's."""
    for ch, frames in ((2, 1024), (1, 2048), (4, 450), (3, 640)):
        x, sr = _k1_stream(ch, frames, 900 + ch)
        plan = glc_amd.plan_encode(x.size, ch)
        nf = plan.n_frames
        assert 1792 < nf * ch <= 2048
        d_pcm = torch_cuda.from_numpy(x).cuda()
        enc = glc_amd.Encoder(sr)
        out = []
        for variant in (4, 0):
            _set_mdct_variant(enc, variant)
            d_coef = torch_cuda.full((nf * ch, 1024), float("nan"), dtype=torch_cuda.float32, device="cuda")
            torch_cuda.cuda.synchronize()
            enc.mdct_forward_device(d_pcm.data_ptr(), 0, plan.per_channel, x.size, ch, 0, nf, d_coef.data_ptr())
            enc.synchronize()
            out.append(d_coef.cpu().numpy())
        assert np.array_equal(bits(out[0]), bits(out[1]))
        for a in (0, nf // 2, nf - 4):
            _, ref = O.encode_range_records(x, 0, plan.per_channel, x.size, sr, ch, a, a + 4, taps=True)
            assert np.array_equal(bits(out[0][a * ch:(a + 4) * ch]), bits(ref.coeffs))


def test_k1_small_launch_tile_edges(torch_cuda):
    """The short-clip transform (k_mdct_fwd_small: 2 x 2 outputs per lane up to 640 rows, 2 x 4 above)
    across its 32-row tile edges, with partial last tiles, ragged ends, and at both ends of each range
    (3583 rows with a segment loader, 4095 without)."""
    for ch, frames in ((1, 70), (2, 255), (3, 170), (8, 64), (2, 320), (1, 641), (2, 500), (3, 597), (8, 224), (5, 300),
                       (1, 3583), (7, 585)):
        x, sr = _k1_stream(ch, frames, 7 * ch)
        plan = glc_amd.plan_encode(x.size, ch)
        nf = plan.n_frames
        assert nf * ch <= (3583 if ch in (1, 2, 4, 8) else 4095)
        d_pcm = torch_cuda.from_numpy(x).cuda()
        d_coef = torch_cuda.zeros((nf * ch, 1024), dtype=torch_cuda.float32, device="cuda")
        enc = glc_amd.Encoder(sr)
        torch_cuda.cuda.synchronize()
        enc.mdct_forward_device(d_pcm.data_ptr(), 0, plan.per_channel, x.size, ch, 0, nf, d_coef.data_ptr())
        enc.synchronize()
        coef = d_coef.cpu().numpy()
        for a in (0, nf // 2, nf - 4):
            _, ref = O.encode_range_records(x, 0, plan.per_channel, x.size, sr, ch, a, a + 4, taps=True)
            assert np.array_equal(bits(coef[a * ch:(a + 4) * ch]), bits(ref.coeffs))


# ---------------------------------------------------------------------------------------
# Decode-side tap: glc_imdct_device (dequant + imdct_block + window, raw frames) against the oracle's
# imdct_block, f32 bit equality, before any overlap-add.  Reference: src/codec.rs:626-675, :377-390.
# ---------------------------------------------------------------------------------------

def _ref_blocks(glc_bytes, frames):
    g = parse_glc(glc_bytes)
    ch = g["channels"]
    _, w, _ = O.tables()
    out = np.zeros((len(frames) * ch, 2048), np.float32)
    for j, f in enumerate(frames):
        fr = g["frames"][f]
        for c in range(ch):
            if fr["raw"] is not None:
                raw = fr["raw"]
                idx = np.arange(2048) * ch + c
                ok = idx < raw.size
                v = np.zeros(2048, np.float32)
                v[ok] = raw[idx[ok]].astype(np.float32) / np.float32(32767.0)      # :638
                out[j * ch + c] = v
                continue
            k, q = fr["lists"][c]
            coeffs = np.zeros(1024, np.float32)
            scale = np.maximum(fr["scales"][c], np.float32(1e-12))                 # :653
            for kk, qq in zip(k.tolist(), q.tolist()):
                if kk < 1024:
                    coeffs[kk] = (np.float32(qq) / np.float32(32768.0)) * scale    # :663
            out[j * ch + c] = O.imdct_block(coeffs) * w                            # :669, :674
    return out


@pytest.mark.parametrize("ch,n_per", [(1, 40000), (2, 30000), (3, 9000), (6, 8000)])
def test_imdct_tap_bit_exact(torch_cuda, ch, n_per):
    sr = 48000
    x = np.concatenate([gen_chord(sr, ch, n_per, n_tones=9), gen_noise(sr, ch, 0.08, 5), gen_chord(sr, ch, 5000, seed=3)])
    ref = O.encode(x, sr, ch)
    ea = glc_amd.EncodedAudio.from_bytes(ref.glc)
    nf = ea.info().n_frames
    assert 0 < ea.info().n_raw_frames < nf
    dec = glc_amd.Decoder(ch, sr)
    for f0, f1 in ((0, nf), (3, nf - 2), (5, 6)):
        d_blk = torch_cuda.full(((f1 - f0) * ch, 2048), float("nan"), dtype=torch_cuda.float32, device="cuda")
        torch_cuda.cuda.synchronize()
        dec.imdct_device(ea, f0, f1, d_blk.data_ptr())
        dec.synchronize()
        got = d_blk.cpu().numpy()
        frames = sorted(set([f0, f0 + 1, (f0 + f1) // 2, f1 - 1] + list(range(f0, min(f1, f0 + 12)))))
        frames = [f for f in frames if f0 <= f < f1]
        want = _ref_blocks(ref.glc, frames)
        for j, f in enumerate(frames):
            a = got[(f - f0) * ch:(f - f0 + 1) * ch]
            assert np.array_equal(bits(a), bits(want[j * ch:(j + 1) * ch])), (ch, f0, f1, f)


def test_decode_reuses_resident_rows(torch_cuda):
    """A context keeps the sparse rows of the last stream it decoded on the device: decoding the same
    EncodedAudio again (any entry point) uploads nothing and gives the same bits; another stream in
    between replaces them."""
    sr, ch = 44100, 2
    xa = gen_chord(sr, ch, 30000, n_tones=6)
    xb = np.concatenate([gen_chord(sr, ch, 12000, seed=5), gen_noise(sr, ch, 0.1, 8)])
    ea, eb = glc_amd.Encoder(sr).encode(xa, ch), glc_amd.Encoder(sr).encode(xb, ch)
    ra, rb = O.decode(ea.to_bytes())[0], O.decode(eb.to_bytes())[0]
    dec = glc_amd.Decoder(ch, sr)
    for _ in range(2):
        assert np.array_equal(bits(dec.decode(ea)), bits(ra))
        assert np.array_equal(bits(dec.decode(ea)), bits(ra))
        chunks = np.concatenate([c.samples for c in dec.decode_streaming(ea)])
        assert np.array_equal(bits(chunks[512:512 + ra.size]), bits(ra))
        assert np.array_equal(bits(dec.decode(eb)), bits(rb))
    # a stream reloaded from bytes is a different object with the same content
    ea2 = glc_amd.EncodedAudio.from_bytes(ea.to_bytes())
    assert np.array_equal(bits(dec.decode(ea2)), bits(ra))
    del ea2
    assert np.array_equal(bits(dec.decode(ea)), bits(ra))


def test_infinite_scale_with_stored_zero(torch_cuda):
    """A stored q == 0 under an infinite scale dequantises to NaN (0 * inf, src/codec.rs:663) and
    poisons the frame - whatever order the list is in (canonical lists go to the device as stored,
    others through the host canonicalisation)."""
    import struct
    sr, ch = 44100, 1
    inf = float("inf")
    for lists, scales in [([[(3, 0), (9, 100)], [(4, 7)]], [inf, 0.25]),        # ascending: canonical path
                          ([[(9, 100), (3, 0)], [(4, 7)]], [inf, 0.25]),        # descending: host canonicalisation
                          ([[(3, 0), (3, 0), (1, 5)], [(4, 7)]], [inf, inf]),   # duplicate + infinite everywhere
                          ([[(3, 0)], [(2000, 0), (5, 0)]], [0.5, inf])]:       # finite scale: zero stays harmless
        body = b""
        for l, sc in zip(lists, scales):
            body += struct.pack("<Q", 1) + struct.pack("<Q", len(l)) + b"".join(struct.pack("<Hh", i, q) for i, q in l)
            body += struct.pack("<Qf", 1, sc) + b"\x00"
        data = struct.pack("<IHQQ", sr, ch, 3000, len(lists)) + body + struct.pack("<IIQ", 512, 0, 3000)
        ref, _, _ = O.decode(data)
        dec = glc_amd.Decoder(ch, sr).decode(glc_amd.EncodedAudio.from_bytes(data))
        assert np.array_equal(bits(dec), bits(ref)), (lists, scales)


def test_device_compaction_blob_equals_host_twin(torch_cuda):
    """glc_compact_device_records == glc_compact_records byte for byte (deterministic padding), on
    compressed + raw frames, for several channel counts, an empty range and a too-small buffer."""
    for sr, ch in ((44100, 1), (48000, 2), (44100, 3), (96000, 8)):
        x = np.concatenate([gen_chord(sr, ch, 9000), gen_noise(sr, ch, 0.15, 5), gen_chord(sr, ch, 3000, seed=2)])
        plan = glc_amd.plan_encode(x.size, ch)
        recs, _ = device_encode(torch_cuda, x, sr, ch, want_coeffs=False)
        d_rec = torch_cuda.from_numpy(recs).cuda()
        cap = glc_amd.compact_bound(ch, plan.n_frames)
        d_blob = torch_cuda.full((cap,), 0xAB, dtype=torch_cuda.uint8, device="cuda")
        enc = glc_amd.Encoder(sr)
        torch_cuda.cuda.synchronize()
        info = enc.compact_device_records(d_rec.data_ptr(), plan.n_frames, ch, d_blob.data_ptr(), cap)
        host = glc_amd.compact_records(recs, ch)
        assert info.bytes == host.size and info.n_frames == plan.n_frames and info.n_raw_rows > 0
        assert np.array_equal(d_blob.cpu().numpy()[:info.bytes], host)
        out = glc_amd.EncodedAudio.from_compact(sr, x.size, ch, [d_blob.cpu().numpy()[:info.bytes]])
        assert out.to_bytes() == O.encode(x, sr, ch).glc
        # two shards compacted separately assemble to the same stream
        cut = plan.n_frames // 2 + 1
        rb = glc_amd.lib.glc_record_bytes(ch)
        blobs = []
        for a, b in ((0, cut), (cut, plan.n_frames)):
            i2 = enc.compact_device_records(d_rec.data_ptr() + a * rb, b - a, ch, d_blob.data_ptr(), cap)
            blobs.append(d_blob.cpu().numpy()[:i2.bytes].copy())
        assert glc_amd.EncodedAudio.from_compact(sr, x.size, ch, blobs).to_bytes() == out.to_bytes()
        # empty range; undersized buffer
        i0 = enc.compact_device_records(0, 0, ch, d_blob.data_ptr(), cap)
        assert (i0.n_frames, i0.n_pairs, i0.n_raw_rows) == (0, 0, 0) and i0.bytes == glc_amd.compact_records(recs[:0], ch).size
        with pytest.raises(glc_amd.GlcError):
            enc.compact_device_records(d_rec.data_ptr(), plan.n_frames, ch, d_blob.data_ptr(), cap - 1)


def _random_sparse_stream(rng, sr, ch, nf, max_nnz=160, raw_share=0.05):
    """EncodedAudio with random sparse rows (a share of common indices per channel) and some raw
    frames, built from fixed-size records - decode-side test material the encoder would never emit."""
    rec = glc_amd.lib.glc_record_bytes(ch)
    hdr = rec - 4096 * ch
    M = nf * ch
    nnz = rng.integers(0, max_nnz + 1, M)
    common = [rng.permutation(1024) for _ in range(ch)]
    buf = np.zeros((nf, rec), np.uint8)
    pay = buf[:, hdr:].view(np.int16).reshape(nf, ch, 2048)
    for c in range(ch):
        for f in range(nf):
            n = int(nnz[f * ch + c])
            idx = common[c][:n] if rng.random() < 0.8 else rng.permutation(1024)[:n]
            pay[f, c, idx] = rng.integers(1, 20000, n).astype(np.int16) * rng.choice(np.array([-1, 1], np.int16), n)
    raw_frame = rng.random(nf) < raw_share
    pay[raw_frame] = rng.integers(-32768, 32768, (int(raw_frame.sum()), ch, 2048)).astype(np.int16)
    meta = buf[:, 8:8 + 8 * ch].view(np.uint32).reshape(nf, ch, 2)
    meta[:, :, 0] = rng.uniform(1e-3, 1.0, (nf, ch)).astype(np.float32).view(np.uint32)
    meta[:, :, 1] = (pay[:, :, :1024] != 0).sum(2)
    buf[:, 0:4].view(np.uint32)[:, 0] = raw_frame
    return glc_amd.EncodedAudio.from_records(sr, nf * 1024 * ch, ch, buf.reshape(-1))


def _set_imdct_variant(ctx, v):
    glc_amd.lib.glc_debug_set_imdct_variant.restype = C.c_int
    glc_amd.lib.glc_debug_set_imdct_variant.argtypes = [C.c_void_p, C.c_int]
    assert glc_amd.lib.glc_debug_set_imdct_variant(ctx._h, v) == 0


@pytest.mark.parametrize("ch,nf", [(7, 2400), (300, 21), (1, 4100)])
def test_d1_plan_batches_and_cross_check_kernels(torch_cuda, ch, nf):
    """The shipped inverse transform (plan + apply) against the two cross-check forms of
    include/glc_debug.h - one row per workgroup, and plan + apply without the scalar skip - bit for bit,
    on streams whose (frame group, channel) units exceed one plan batch of 2048 units (7 ch x 300
    groups, 300 ch x 3 groups with 6 groups per batch) and on a mono stream longer than one decode round."""
    rng = np.random.default_rng(ch * 1000 + nf)
    ea = _random_sparse_stream(rng, 48000, ch, nf)
    dec = glc_amd.Decoder(ch, 48000)
    outs = []
    for v in (0, 1, 2):
        _set_imdct_variant(dec, v)
        d = torch_cuda.full(((nf + 1) * 1024 * ch,), float("nan"), dtype=torch_cuda.float32, device="cuda")
        torch_cuda.cuda.synchronize()
        dec.decode_device(ea, d.data_ptr(), d.numel())
        dec.synchronize()
        outs.append(d.cpu().numpy().view(np.uint32))
    _set_imdct_variant(dec, 0)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # and the same bits through the host API (rounds + D2H pipelining)
    host = dec.decode(ea)
    assert np.array_equal(bits(host), outs[0][512:512 + host.size])


@pytest.mark.parametrize("ch", [1, 2, 4, 3])
def test_raw_frames_in_a_later_round(torch_cuda, ch):
    """Streams longer than one 4096-frame encode round with raw (noise) frames inside the SECOND round:
    the quantiser takes the raw decision itself for 1 / 2 / 4 channels (the frame's rows sit in one
    wave) and must window the right absolute frame; 3 channels go through k_decide_raw.  Records of a
    window around the burst equal the oracle's, byte for byte."""
    sr = 48000
    nf = 4096 + 120
    rng = np.random.default_rng(77 + ch)
    t = np.arange(nf * 1024, dtype=np.float64)[:, None]
    x = (np.sin(2 * np.pi * rng.uniform(100, 6000, (1, ch)) * t / sr) * 0.3).astype(np.float32)
    lo, hi = (4096 + 30) * 1024, (4096 + 60) * 1024
    x[lo:hi] = rng.standard_normal((hi - lo, ch)).astype(np.float32) * 0.3
    x = x.reshape(-1)
    plan = glc_amd.plan_encode(x.size, ch)
    assert plan.n_frames == nf
    recs, _ = device_encode(torch_cuda, x, sr, ch, want_coeffs=False)
    rec = glc_amd.lib.glc_record_bytes(ch)
    f0, f1 = 4096 + 20, 4096 + 70
    t0, t1 = f0 * 1024 - 512, (f1 - 1) * 1024 - 512 + 2048
    want, taps = O.encode_range_records(x[t0 * ch:t1 * ch], t0, t1 - t0, x.size, sr, ch, f0, f1, taps=True)
    assert 10 < int(taps.is_raw.sum()) < f1 - f0
    assert np.array_equal(recs[f0 * rec:f1 * rec], want)
    # and the host API agrees with the device records
    enc = glc_amd.Encoder(sr).encode(x, ch)
    assert enc.to_bytes() == glc_amd.EncodedAudio.from_records(sr, x.size, ch, recs).to_bytes()
    assert enc.info().n_raw_frames == int(split_records(recs, ch)[0].sum())


@pytest.mark.parametrize("ch,shape", [(2, "sparse-then-dense"), (2, "dense-then-sparse"), (1, "raw-then-tonal"),
                                       (3, "sparse-then-dense"), (2, "three-rounds-raw-last")])
def test_host_encode_pipeline_across_density_changes(torch_cuda, ch, shape):
    """glc_encode runs upload / kernels / download as a pipeline over rounds and grows the EncodedAudio's
    pools ahead of need from the density of the rounds it has seen: streams whose later rounds are much
    denser (the estimate is too small), much sparser (the pools shrink at the end), or raw (the other
    pool) must give exactly the bytes of the unpipelined device path - one frames_from_device_records over
    the whole range - and, on windows across round boundaries, the oracle's records."""
    sr = 48000
    piece = max(1, 2048 // ch)               # frames of an opening round; four of them, then 4096 (csrc/glc_api.hip)
    first = 4 * piece
    rounds = 3 if shape.startswith("three") else 2
    nf = first + 4096 * (rounds - 1) - 37    # ragged last round
    rng = np.random.default_rng(1234 + ch + len(shape))
    total = nf * 1024

    def tones(n, a, b):  # rows [a, b) of an n-tone chord per channel
        t = np.arange(a, b, dtype=np.float64)[:, None]
        out = np.zeros((b - a, ch), np.float32)
        for i in range(n):
            out += (np.sin(2 * np.pi * rng.uniform(60, 15000, (1, ch)) * t / sr + i) * (0.5 / n)).astype(np.float32)
        return out

    cut = first * 1024
    x = np.empty((total, ch), np.float32)
    if shape == "sparse-then-dense":
        x[:cut], x[cut:] = tones(2, 0, cut), tones(40, cut, total)
    elif shape == "dense-then-sparse":
        x[:cut], x[cut:] = tones(40, 0, cut), tones(1, cut, total) * 0.2
    elif shape == "raw-then-tonal":
        x[:cut], x[cut:] = rng.standard_normal((cut, ch)).astype(np.float32) * 0.3, tones(8, cut, total)
    else:
        c2 = (first + 4096) * 1024
        x[:c2], x[c2:] = tones(6, 0, c2), rng.standard_normal((total - c2, ch)).astype(np.float32) * 0.3
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    plan = glc_amd.plan_encode(x.size, ch)
    assert plan.n_frames == nf
    enc = glc_amd.Encoder(sr)
    got = enc.encode(x, ch)
    recs, _ = device_encode(torch_cuda, x, sr, ch, want_coeffs=False)
    assert got.to_bytes() == glc_amd.EncodedAudio.from_records(sr, x.size, ch, recs).to_bytes()
    # twice through the same context: the second call reuses every buffer, event and stream of the first
    assert enc.encode(x, ch).to_bytes() == got.to_bytes()
    rec = glc_amd.lib.glc_record_bytes(ch)
    for edge in (piece, first):              # across a boundary between opening rounds, and into the first long round
        f0, f1 = edge - 12, edge + 12
        t0, t1 = f0 * 1024 - 512, (f1 - 1) * 1024 - 512 + 2048
        want, _ = O.encode_range_records(x[t0 * ch:t1 * ch], t0, t1 - t0, x.size, sr, ch, f0, f1)
        assert np.array_equal(recs[f0 * rec:f1 * rec], want)
    info = got.info()
    if "raw" in shape:
        assert 0 < info.n_raw_frames < nf


def test_multi_chunk_range_keeps_the_callers_stream_order(torch_cuda):
    """glc_encode_range_device over several 4096-frame chunks forks onto a second stream and joins again:
    on the CALLER's stream it must still behave like one in-order operation.  The samples are produced
    by work queued on that stream right before the call (a device-to-device copy behind a long
    fill), the records are consumed by work queued right after it (a copy into another buffer), with no
    host synchronisation in between; the result equals the synchronised single-stream path."""
    torch = torch_cuda
    sr, ch, nf = 48000, 2, 3 * 4096 + 100
    rng = np.random.default_rng(99)
    t = np.arange(nf * 1024, dtype=np.float64)[:, None]
    x = (np.sin(2 * np.pi * rng.uniform(100, 9000, (1, ch)) * t / sr) * 0.4).astype(np.float32)
    x[5000 * 1024:5040 * 1024] = rng.standard_normal((40 * 1024, ch)).astype(np.float32) * 0.3  # raw frames in chunk 1
    x = x.reshape(-1)
    want, _ = device_encode(torch, x, sr, ch, want_coeffs=False)
    rec = glc_amd.lib.glc_record_bytes(ch)
    s = torch.cuda.Stream()
    enc = glc_amd.Encoder(sr)
    enc.set_stream(s.cuda_stream)
    src = torch.from_numpy(x).cuda()
    d_pcm = torch.zeros_like(src)
    d_rec = torch.zeros(nf * rec, dtype=torch.uint8, device="cuda")
    out = torch.zeros_like(d_rec)
    junk = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        junk.fill_(1)                      # keeps the stream busy, so the copy below is still pending ...
        d_pcm.copy_(src, non_blocking=True)
        enc.encode_range_device(d_pcm.data_ptr(), 0, x.size // ch, x.size, ch, 0, nf, d_rec.data_ptr())  # ... when this is queued
        out.copy_(d_rec, non_blocking=True)  # and this must see all three chunks' records
    s.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    enc.set_stream(0)


def test_pipelined_encode_contexts_come_and_go_and_run_side_by_side(torch_cuda):
    """Multi-round glc_encode keeps two helper threads parked in its context: contexts that are created,
    used for a few pipelined calls and destroyed in a loop must neither leak them nor hang on the way out,
    and three host threads, each with its own context and its own many-round stream (one of them on a
    stream torch owns), must all produce the bytes of the unpipelined device path."""
    import threading
    sr = 48000
    streams = []
    for i, ch in enumerate((2, 5, 1)):
        piece = max(1, 2048 // ch)
        nf = 4 * piece + 4096 + 300 + 17 * i   # four opening rounds, one full round, a ragged one
        rng = np.random.default_rng(500 + i)
        t = np.arange(nf * 1024, dtype=np.float64)[:, None]
        x = (np.sin(2 * np.pi * rng.uniform(80, 7000, (1, ch)) * t / sr) * 0.35).astype(np.float32)
        x[(piece + 3) * 1024:(piece + 9) * 1024] = rng.standard_normal((6 * 1024, ch)).astype(np.float32) * 0.3
        x = x.reshape(-1)
        recs, _ = device_encode(torch_cuda, x, sr, ch, want_coeffs=False)
        streams.append((x, ch, glc_amd.EncodedAudio.from_records(sr, x.size, ch, recs).to_bytes()))
    for _ in range(6):  # contexts come and go
        enc = glc_amd.Encoder(sr)
        for x, ch, want in streams[:2]:
            assert enc.encode(x, ch).to_bytes() == want
        enc.close()
    errors = []

    def work(i):
        try:
            x, ch, want = streams[i]
            enc = glc_amd.Encoder(sr)
            s = None
            if i == 0:
                s = torch_cuda.cuda.Stream()
                enc.set_stream(s.cuda_stream)
            for _ in range(5):
                assert enc.encode(x, ch).to_bytes() == want
            if s is not None:
                enc.set_stream(0)
            enc.close()
        except BaseException as e:  # noqa: BLE001 - reported to the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_torch_and_library_share_one_hip_runtime(torch_cuda):
    """is a torch stream handle passed to glc_ctx_set_stream a foreign pointer?  No: with
    torch imported first (glc_amd/_lib.py does) the loader resolves libglc_hip.so's libamdhip64.so.7 to
    the runtime torch bundles (same SONAME), so exactly one HIP runtime is mapped after both sides have
    initialised the device, and an event recorded by torch on its stream orders work the library queues."""
    from glc_amd._lib import hip_runtimes_mapped
    torch = torch_cuda
    enc = glc_amd.Encoder(48000)
    rts = hip_runtimes_mapped()
    assert len(rts) == 1, rts
    assert glc_amd.lib.glc_ctx_device(enc._h) == torch.cuda.current_device()
    s = torch.cuda.Stream()
    enc.set_stream(s.cuda_stream)   # refused (GlcError) if a second runtime were mapped
    assert glc_amd.lib.glc_ctx_stream(enc._h) == s.cuda_stream
    enc.set_stream(0)


def test_mono_stream_across_its_8192_frame_chunks(torch_cuda):
    """A mono launch of 4096 rows would leave every CU one workgroup (two waves per SIMD), so mono streams
    are encoded in chunks of 8192 frames: a stream of two chunks + a ragged rest, through glc_encode (rounds),
    glc_encode_range_device (chunks alternating between two streams) and a shard that starts inside chunk 2,
    against the oracle."""
    torch = torch_cuda
    sr, ch, frames = 48000, 1, 2 * 8192 + 301
    rng = np.random.default_rng(2024)
    t = np.arange(frames * 1024, dtype=np.float64)
    x = (0.3 * np.sin(2 * np.pi * 441.0 * t / sr) + 0.1 * np.sin(2 * np.pi * 5003.0 * t / sr)).astype(np.float32)
    for f in (8190, 8191, 8192, 16383, 16384):           # noise bursts (raw frames) on both sides of the chunk edges
        x[f * 1024:(f + 1) * 1024] = rng.standard_normal(1024).astype(np.float32) * 0.3
    x = x[:-77]
    ref = O.encode(x, sr, ch, taps=True)
    assert ref.n_frames > 2 * 8192 and ref.is_raw.sum() >= 4
    assert glc_amd.Encoder(sr).encode(x, ch).to_bytes() == ref.glc
    whole, _ = device_encode(torch, x, sr, ch, want_coeffs=False)
    assert glc_amd.EncodedAudio.from_records(sr, x.size, ch, whole).to_bytes() == ref.glc
    plan = glc_amd.plan_encode(x.size, ch)
    rec = glc_amd.lib.glc_record_bytes(ch)
    f0 = 8192 + 5
    t0 = f0 * 1024 - 512
    part, _ = device_encode(torch, x, sr, ch, f0, plan.n_frames, t0, plan.per_channel - t0, want_coeffs=False)
    assert np.array_equal(part, whole[f0 * rec:])
