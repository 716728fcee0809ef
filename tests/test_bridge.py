"""The structured bridge of include/glc.h (glc_frames_view / _from_parts / _from_gather, stream ids,
glc_encode_hooked, glc_decode_resident): how a host whose EncodedAudio is the reference's nested
vectors (src/codec.rs:31-69) crosses the C ABI without a byte stream in between.

CPU part: the constructors and the view round-trip the oracle's streams byte for byte and reject
inconsistent arrays.  GPU part (-m gpu): encode -> view -> nested -> gather -> decode equals the oracle
bit for bit; the hook sees the stream range by range; a stream id keeps the rows (and the
inverse-transform plan) resident across calls - including across different output buffers and raw frames."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

import glc_amd
from glc_amd import EncodedAudio
from conftest import ROOT, gen_chord, gen_noise, gen_tone
from oracle import oracle as O


def nested_from_parts(p):
    """EncodedAudio.parts() -> [(lists, scales, raw)] per frame: the reference's Vec<EncodedFrame>."""
    fr = []
    for i in range(p["n_frames"]):
        l0, l1 = int(p["list_begin"][i]), int(p["list_begin"][i + 1])
        lists = [p["pairs"][int(p["list_off"][l]):int(p["list_off"][l + 1])].copy() for l in range(l0, l1)]
        sc = p["scales"][int(p["scale_begin"][i]):int(p["scale_begin"][i + 1])].copy()
        raw = p["raw"][int(p["raw_begin"][i]):int(p["raw_begin"][i + 1])].copy() if p["raw_tag"][i] else None
        fr.append((lists, sc, raw))
    return fr


def oracle_streams():
    yield "cfg1", O.encode(gen_tone("sine", 440.0, 44100, 2, 2.0), 44100, 2).glc
    yield "noise_raw", O.encode(gen_noise(44100, 2, 0.25, 12345), 44100, 2).glc
    yield "mixed_3ch", O.encode(np.concatenate([gen_chord(44100, 3, 9000), gen_noise(44100, 3, 0.2, 5)]), 44100, 3).glc
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.glc"))):
        yield os.path.basename(f), open(f, "rb").read()


def test_view_parts_gather_roundtrip_bytes():
    for name, blob in oracle_streams():
        ea = EncodedAudio.from_bytes(blob)
        p = ea.parts()
        assert p["n_frames"] == ea.info().n_frames and p["n_pairs"] == ea.info().total_nnz
        assert p["list_begin"][-1] == p["n_lists"] and p["list_off"][-1] == p["n_pairs"]
        assert EncodedAudio.from_parts(p).to_bytes() == blob, name
        assert EncodedAudio.from_nested(ea.header, nested_from_parts(p), ea.gapless_info).to_bytes() == blob, name


def test_view_layout_is_the_bincode_pair_layout():
    """pairs[j] = idx | q << 16, little-endian: the 4 bytes bincode writes for one (u16, i16)."""
    ref = O.encode(gen_tone("sine", 440.0, 44100, 2, 2.0), 44100, 2)
    ea = EncodedAudio.from_bytes(ref.glc)
    p = ea.parts()
    fr = ea.frames[3]
    l = int(p["list_begin"][3])
    a, b = int(p["list_off"][l]), int(p["list_off"][l + 1])
    got = [(int(v) & 0xFFFF, int(np.int16(np.uint16(int(v) >> 16)))) for v in p["pairs"][a:b]]
    assert got == fr.sparse_coeffs_per_channel[0] and len(got) > 0


def test_stream_ids():
    blob = O.encode(gen_tone("sine", 440.0, 44100, 2, 0.5), 44100, 2).glc
    ea = EncodedAudio.from_bytes(blob)
    assert ea.stream_id >> 63 == 1, "library-made objects live in the upper half of the id space"
    assert EncodedAudio.from_bytes(blob).stream_id != ea.stream_id
    p = ea.parts()
    assert EncodedAudio.from_parts(p, 1234).stream_id == 1234
    assert EncodedAudio.from_parts(p).stream_id >> 63 == 1
    with pytest.raises(glc_amd.GlcError):
        EncodedAudio.from_parts(p, 1 << 63)


def test_from_parts_rejects_inconsistent_arrays():
    blob = O.encode(np.concatenate([gen_chord(44100, 2, 6000), gen_noise(44100, 2, 0.2, 5)]), 44100, 2).glc
    good = EncodedAudio.from_bytes(blob).parts()
    assert good["n_raw"] > 0 and good["n_pairs"] > 0

    def broken(**kw):
        p = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in good.items()}
        for k, f in kw.items():
            p[k] = f(p[k])
        return p

    def bump(i, d):
        def f(a):
            a = a.copy()
            a[i] += np.uint64(d) if d > 0 else np.uint64(0)
            if d < 0:
                a[i] -= np.uint64(-d)
            return a
        return f
    cases = [
        broken(list_off=bump(3, 10_000_000)),          # offset beyond the pool / not monotonic
        broken(list_off=bump(-1, 1)),                  # closing offset != n_pairs
        broken(list_begin=bump(2, 5)),                 # frame claims lists of the next frames, then goes back
        broken(scale_begin=bump(-1, -1)),              # does not span the pool
        broken(raw_begin=bump(0, 1)),                  # does not start at 0
        broken(raw_tag=lambda a: np.where(a == 1, 2, a).astype(np.uint8)),   # Option tag 2
        broken(raw_tag=lambda a: np.zeros_like(a)),    # raw samples owned by frames without raw_pcm
        broken(n_pairs=lambda n: n + 1),               # count disagrees with the offsets
    ]
    for i, p in enumerate(cases):
        with pytest.raises(glc_amd.GlcError) as e:
            EncodedAudio.from_parts(p)
        assert e.value.code in (-1, -4), (i, e.value)
    # None pointers with non-zero counts
    v = glc_amd._lib.GlcFramesView()
    v.n_frames, v.n_pairs = 1, 4
    out = C.c_void_p()
    assert glc_amd.lib.glc_frames_from_parts(C.byref(v), 0, C.byref(out)) == -1
    assert glc_amd.lib.glc_frames_from_parts(None, 0, C.byref(out)) == -1


def test_gather_accepts_general_streams():
    """Any well-formed EncodedAudio: more lists than channels, empty lists, Some(vec![]), no scales."""
    hdr = glc_amd.AudioHeader(44100, 2, 12345)
    gap = glc_amd.GaplessInfo(512, 7, 12345)
    frames = [([np.array([1 | (5 << 16), 9 | (0xFFFB << 16)], np.uint32), np.empty(0, np.uint32), np.array([3], np.uint32)],
               np.array([0.5, 0.25, 1.0], np.float32), None),
              ([], np.empty(0, np.float32), np.empty(0, np.int16)),
              ([np.empty(0, np.uint32)], np.array([1e-10], np.float32), np.arange(-3, 4, dtype=np.int16))]
    ea = EncodedAudio.from_nested(hdr, frames, gap, stream_id=99)
    blob = ea.to_bytes()
    back = EncodedAudio.from_bytes(blob)
    assert back.to_bytes() == blob and ea.stream_id == 99
    p = ea.parts()
    assert p["n_frames"] == 3 and p["n_lists"] == 4 and p["n_pairs"] == 3 and p["n_raw"] == 7
    assert list(p["raw_tag"]) == [0, 1, 1] and list(p["raw_begin"]) == [0, 0, 0, 7]
    assert EncodedAudio.from_parts(p).to_bytes() == blob


# --------------------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU (no CPU fallback exists)"
    return torch


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


GPU_CASES = [
    ("cfg1_sine_44k_stereo", lambda: gen_tone("sine", 440.0, 44100, 2, 2.0), 44100, 2),
    ("noise_raw_44k_stereo", lambda: gen_noise(44100, 2, 0.4, 12345), 44100, 2),
    ("mixed_3ch", lambda: np.concatenate([gen_chord(44100, 3, 30000), gen_noise(44100, 3, 0.3, 5), gen_chord(44100, 3, 9000)]), 44100, 3),
    ("chord_48k_stereo_5_rounds", lambda: gen_chord(48000, 2, 1024 * 4700), 48000, 2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,sr,ch", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_shim_equivalent_roundtrip_equals_oracle(torch_cuda, name, make, sr, ch):
    """Encoder::encode -> nested EncodedAudio -> Decoder::decode through the bridge, against the oracle."""
    x = make()
    ref = O.encode(x, sr, ch)
    dref, _, _ = O.decode(ref.glc)
    enc = glc_amd.Encoder(sr)
    ea = enc.encode(x, ch)
    p = ea.parts()
    assert EncodedAudio.from_parts(p).to_bytes() == ref.glc
    nested = nested_from_parts(p)
    dec = glc_amd.Decoder(ch, sr)
    for build in (lambda: EncodedAudio.from_nested(ea.header, nested, ea.gapless_info),
                  lambda: EncodedAudio.from_parts(p)):
        e2 = build()
        assert e2.to_bytes() == ref.glc
        out = dec.decode(e2)
        assert out.size == dref.size and np.array_equal(bits(out), bits(dref))


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,sr,ch", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_encode_hooked_delivers_the_stream_in_ascending_ranges(torch_cuda, name, make, sr, ch):
    x = make()
    ref = O.encode(x, sr, ch)
    enc = glc_amd.Encoder(sr)
    seen, nested = [], []

    def hook(parts, f0, f1):
        seen.append((f0, f1))
        assert parts["n_frames"] == f1 and parts["list_begin"].size == f1 + 1, (parts["n_frames"], f0, f1)
        sub = nested_from_parts(parts)
        nested.extend(sub[f0:f1])
        return 0
    ea = enc.encode_hooked(x, ch, hook)
    n_frames = ea.info().n_frames
    assert seen[0][0] == 0 and seen[-1][1] == n_frames and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    # (how many ranges a stream of several rounds arrives in depends on how far the device is ahead of the
    # host: a cold first call delivers everything at the end, a warm one in slices of a few dozen frames
    # while it waits - tools/bridge_bench.cpp checks that case)
    for _ in range(3):
        seen.clear(), nested.clear()
        ea = enc.encode_hooked(x, ch, hook)
        assert seen[0][0] == 0 and seen[-1][1] == n_frames and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
        assert EncodedAudio.from_nested(ea.header, nested, ea.gapless_info).to_bytes() == ref.glc
    assert ea.to_bytes() == ref.glc
    assert EncodedAudio.from_nested(ea.header, nested, ea.gapless_info).to_bytes() == ref.glc
    # a hook that says stop: GLC_EINVAL, and the context stays usable
    with pytest.raises(glc_amd.GlcError) as e:
        enc.encode_hooked(x, ch, lambda parts, f0, f1: 1)
    assert e.value.code == -1
    assert enc.encode(x, ch).to_bytes() == ref.glc
    # ... and an exception inside the hook comes back as that exception
    with pytest.raises(ZeroDivisionError):
        enc.encode_hooked(x, ch, lambda parts, f0, f1: 1 // 0)
    assert enc.encode(x, ch).to_bytes() == ref.glc


@pytest.mark.gpu
def test_stream_id_keeps_rows_and_plan_resident(torch_cuda):
    """Two objects built with one stream id are one stream to the context: the second decode prepares
    and uploads nothing (and skips the plan kernel); glc_decode_resident needs no object at all.
    Every result equals the oracle's bits - also into a different output buffer, with raw frames in the
    stream (their blocks are written by the apply kernel, not by the skipped plan kernel)."""
    torch = torch_cuda
    sr, ch = 44100, 2
    x = np.concatenate([gen_chord(sr, ch, 40000), gen_noise(sr, ch, 0.3, 5), gen_chord(sr, ch, 20000, seed=3)])
    ref = O.encode(x, sr, ch, taps=True)
    assert 0 < ref.is_raw.sum() < ref.n_frames
    dref, _, _ = O.decode(ref.glc)
    p = EncodedAudio.from_bytes(ref.glc).parts()
    dec = glc_amd.Decoder(ch, sr)
    assert dec.resident_stream() == 0
    a = EncodedAudio.from_parts(p, 4242)
    out = dec.decode(a)
    assert np.array_equal(bits(out), bits(dref)) and dec.resident_stream() == 4242
    del a
    b = EncodedAudio.from_parts(p, 4242)  # a new object, the same content under the same id
    for _ in range(3):
        assert np.array_equal(bits(dec.decode(b)), bits(dref))
    buf = np.full(dref.size + 5, np.nan, np.float32)
    got = dec.decode_resident(4242, buf)
    assert got.size == dref.size and np.array_equal(bits(got), bits(dref)) and np.isnan(buf[dref.size:]).all()
    # the IMDCT tap into two DIFFERENT device buffers: a skipped plan kernel must not leave raw rows unwritten
    nf = ref.n_frames
    blocks_ref = None
    for i in range(3):
        d_blk = torch.full((nf * ch, 2048), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the decoder on its own: without this the NaNs can land last
        dec.imdct_device(b, 0, nf, d_blk.data_ptr())
        dec.synchronize()
        h = d_blk.cpu().numpy()
        assert not np.isnan(h[ref.is_raw.repeat(ch) == 1]).any(), "raw rows written on every launch"
        if blocks_ref is None:
            blocks_ref = h
        assert np.array_equal(bits(h), bits(blocks_ref))
    # a different range in between invalidates the kept plan; the full range afterwards is planned again
    d_part = torch.empty((10 * ch, 2048), dtype=torch.float32, device="cuda")
    dec.imdct_device(b, 5, 15, d_part.data_ptr())
    dec.synchronize()
    assert np.array_equal(bits(d_part.cpu().numpy()), bits(blocks_ref[5 * ch:15 * ch]))
    assert np.array_equal(bits(dec.decode(b)), bits(dref))
    # not resident -> EINVAL; another stream replaces the resident one; a recycled id with other content is re-prepared
    with pytest.raises(glc_amd.GlcError):
        dec.decode_resident(777, buf)
    y = gen_tone("sine", 440.0, sr, ch, 1.0)
    oy = O.encode(y, sr, ch)
    dy, _, _ = O.decode(oy.glc)
    c = EncodedAudio.from_parts(EncodedAudio.from_bytes(oy.glc).parts(), 4242)
    assert np.array_equal(bits(dec.decode(c)), bits(dy)), "same id, different sizes: treated as a new stream"
    with pytest.raises(glc_amd.GlcError):
        dec.decode_resident(0, buf)


@pytest.mark.gpu
def test_repeated_decode_paths_stay_bit_exact(torch_cuda):
    """Every decode entry point, repeated and interleaved on one context (plan kept / dropped / rebuilt)."""
    torch = torch_cuda
    sr, ch = 48000, 2
    x = gen_chord(sr, ch, 1024 * 700 + 333)
    ref = O.encode(x, sr, ch)
    dref, _, _ = O.decode(ref.glc)
    ea = EncodedAudio.from_bytes(ref.glc)
    dec = glc_amd.Decoder(ch, sr)
    nf = ref.n_frames
    d_all = torch.empty((nf + 1) * 1024 * ch, dtype=torch.float32, device="cuda")
    for round_ in range(3):
        s, n = dec.decode_device(ea, d_all.data_ptr(), d_all.numel())
        dec.synchronize()
        assert np.array_equal(bits(d_all.cpu().numpy()[s:s + n]), bits(dref))
        assert np.array_equal(bits(dec.decode(ea)), bits(dref))
        chunks = [c.samples for c in dec.decode_streaming(ea)]
        allv = np.concatenate(chunks)[512:][:dref.size]
        assert np.array_equal(bits(allv), bits(dref))
        for variant in (1, 2, 3, 4, 0):
            glc_amd.lib.glc_debug_set_imdct_variant.restype = C.c_int
            glc_amd.lib.glc_debug_set_imdct_variant.argtypes = [C.c_void_p, C.c_int]
            assert glc_amd.lib.glc_debug_set_imdct_variant(dec._h, variant) == 0
            assert np.array_equal(bits(dec.decode(ea)), bits(dref)), variant


@pytest.mark.gpu
def test_bridge_bench_tool_agrees_with_itself(torch_cuda):
    """tools/bridge_bench.cpp at BASELINE config 1 and a 3-channel stream with raw frames absent/present:
    the driver exits non-zero if any path disagrees."""
    exe = os.path.join(ROOT, "build", "bridge_bench")
    assert os.path.exists(exe), "build/bridge_bench missing (make -C gapless-lossy-codec_amd/csrc tools)"
    for args in (["86", "2", "44100"], ["300", "3", "48000"]):
        r = subprocess.run([exe] + args, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and '"ok": true' in r.stdout, r.stdout + r.stderr
