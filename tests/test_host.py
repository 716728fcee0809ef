"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol of
include/glc.h, the .glc container reader/writer round-trips the oracle's bytes, record assembly
reproduces the oracle's byte stream, and compute entry points fail loudly without a GPU.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import glc_amd
from conftest import gen_noise, gen_tone, parse_glc, records_from_taps, ROOT
from oracle import oracle as O


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "glc.h")).read()
    declared = set(re.findall(r"\b(glc_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"glc_status"}
    assert declared, "no declarations parsed"
    assert declared == set(glc_amd.SIGNATURES), declared ^ set(glc_amd.SIGNATURES)
    for name in declared:
        assert hasattr(glc_amd.lib, name)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.Encoder(44100)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


@pytest.mark.parametrize("n,ch", [(176400, 2), (88200, 2), (8388608, 2), (57600000, 2), (513, 1),
                                  (1537, 1), (1538, 1), (48000 * 8, 8), (1027, 2), (3075, 3), (7001, 6)])
def test_plan_matches_oracle(n, ch):
    p = glc_amd.plan_encode(n, ch)
    assert p.n_frames == O.num_frames(n, ch) > 0
    assert p.encoder_delay == 512
    l0 = -(-n // ch)
    assert p.per_channel == l0 and p.padding == p.padded_len - l0 - 512


@pytest.mark.parametrize("n,ch", [(0, 1), (512, 1), (1024, 2), (100, 0), (5, 7)])
def test_plan_rejects_inputs_the_reference_panics_on(n, ch):
    assert O.num_frames(n, ch) == 0
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.plan_encode(n, ch)
    assert e.value.code == -1


def test_ragged_channel_lengths_follow_reference():
    # n % ch != 0: per_chan[c] lengths differ by one (src/codec.rs:428-431); the reference only
    # panics when a shorter channel's padded length under-runs the last frame
    for n, ch in [(2049, 2), (2050, 3), (3 * 1536 + 1, 3), (2 * 1536 + 1, 2)]:
        nf = O.num_frames(n, ch)
        if nf == 0:
            with pytest.raises(glc_amd.GlcError):
                glc_amd.plan_encode(n, ch)
        else:
            assert glc_amd.plan_encode(n, ch).n_frames == nf
    assert O.num_frames(2 * 1536 + 1, 2) == 0  # ch0 has 1537 -> 2 frames, ch1 1536 -> too short


def _golden_bytes():
    g = os.path.join(ROOT, "tests", "golden")
    return [open(os.path.join(g, f), "rb").read() for f in sorted(os.listdir(g)) if f.endswith(".glc")]


def test_container_roundtrip_is_byte_identical():
    for data in _golden_bytes():
        enc = glc_amd.EncodedAudio.from_bytes(data)
        assert enc.to_bytes() == data
        ref = parse_glc(data)
        i = enc.info()
        assert (i.sample_rate, i.channels, i.total_samples) == (ref["sample_rate"], ref["channels"], ref["total_samples"])
        assert (i.encoder_delay, i.padding, i.original_length) == (ref["encoder_delay"], ref["padding"], ref["original_length"])
        assert i.n_frames == len(ref["frames"])
        assert i.n_raw_frames == sum(f["raw"] is not None for f in ref["frames"])
        for k in (0, len(ref["frames"]) // 2, len(ref["frames"]) - 1):
            fr, rf = enc.frames[k], ref["frames"][k]
            if rf["raw"] is not None:
                assert np.array_equal(fr.raw_pcm, rf["raw"]) and not fr.scale_factors
            else:
                assert fr.raw_pcm is None
                assert np.array_equal(np.float32(fr.scale_factors).view(np.uint32), rf["scales"].view(np.uint32))
                for c, (idx, q) in enumerate(rf["lists"]):
                    assert fr.sparse_coeffs_per_channel[c] == list(zip(idx.tolist(), q.tolist()))


def test_save_load(tmp_path):
    data = _golden_bytes()[0]
    enc = glc_amd.EncodedAudio.from_bytes(data)
    p = tmp_path / "a.glc"
    glc_amd.save_encoded(enc, p)
    assert p.read_bytes() == data
    assert glc_amd.load_encoded(p).to_bytes() == data
    with pytest.raises(glc_amd.GlcError) as e:
        glc_amd.load_encoded(tmp_path / "missing.glc")
    assert e.value.code == -6


def test_container_rejects_truncated_and_hostile_streams():
    data = _golden_bytes()[0]
    for cut in (0, 5, 13, 21, 22, 30, 100, len(data) // 2, len(data) - 1):
        with pytest.raises(glc_amd.GlcError) as e:
            glc_amd.EncodedAudio.from_bytes(data[:cut])
        assert e.value.code == -4
    # absurd lengths must not allocate: frame count, list length, raw length
    bad = bytearray(data)
    bad[14:22] = (2 ** 62).to_bytes(8, "little")
    with pytest.raises(glc_amd.GlcError):
        glc_amd.EncodedAudio.from_bytes(bytes(bad))
    bad = bytearray(data)
    bad[30:38] = (2 ** 40).to_bytes(8, "little")  # first sparse list length
    with pytest.raises(glc_amd.GlcError):
        glc_amd.EncodedAudio.from_bytes(bytes(bad))
    # Option tag other than 0/1 is a bincode error
    g = parse_glc(data)
    first = g["frames"][0]
    tag_pos = 22 + 8 + sum(8 + 4 * len(i) for i, _ in first["lists"]) + 8 + 4 * len(first["scales"])
    assert data[tag_pos] == 0
    bad = bytearray(data)
    bad[tag_pos] = 2
    with pytest.raises(glc_amd.GlcError):
        glc_amd.EncodedAudio.from_bytes(bytes(bad))
    # trailing bytes are accepted, like bincode::deserialize
    assert glc_amd.EncodedAudio.from_bytes(data + b"xyz").to_bytes() == data


CASES = [("sine", lambda: gen_tone("sine", 440.0, 44100, 2, 1.0), 44100, 2),
         ("noise", lambda: gen_noise(44100, 2, 0.25, 12345), 44100, 2),
         ("sweep", lambda: gen_tone("sweep", 100.0, 48000, 1, 0.5, 10000.0), 48000, 1)]


@pytest.mark.parametrize("name,make,sr,ch", CASES, ids=[c[0] for c in CASES])
def test_record_assembly_reproduces_oracle_bytes(name, make, sr, ch):
    x = make()
    enc = O.encode(x, sr, ch, taps=True)
    ref = parse_glc(enc.glc)
    raw_rows = {}
    for f, fr in enumerate(ref["frames"]):
        if fr["raw"] is not None:
            for c in range(ch):
                raw_rows[f * ch + c] = fr["raw"][c * 2048:(c + 1) * 2048]
    recs = records_from_taps(enc, ch, raw_rows)
    assert recs.size == enc.n_frames * glc_amd.lib.glc_record_bytes(ch)
    out = glc_amd.EncodedAudio.from_records(sr, x.size, ch, recs)
    assert out.to_bytes() == enc.glc
    # corrupt nnz -> rejected, not mis-assembled
    if not enc.is_raw.all():
        f = int(np.argmin(enc.is_raw))
        bad = recs.copy()
        off = f * glc_amd.lib.glc_record_bytes(ch) + 12
        bad[off:off + 4] = np.frombuffer(np.uint32(enc.nnz[f * ch] + 1).tobytes(), np.uint8)
        with pytest.raises(glc_amd.GlcError):
            glc_amd.EncodedAudio.from_records(sr, x.size, ch, bad)
    with pytest.raises(glc_amd.GlcError):
        glc_amd.EncodedAudio.from_records(sr, x.size, ch, recs[:-glc_amd.lib.glc_record_bytes(ch)])


def test_decoded_len():
    for data in _golden_bytes():
        enc = glc_amd.EncodedAudio.from_bytes(data)
        dec, _, _ = O.decode(data)
        assert glc_amd.lib.glc_decoded_len(enc._h) == dec.size


def test_container_fuzz_never_crashes():
    """f1: the reader trusts no length field.  Random mutations of a valid stream either fail with
    GLC_EFORMAT or yield a stream that re-serialises to a fixed point (serialise(deserialise(x))
    is stable), and never crash or allocate absurdly."""
    from hypothesis import given, settings, strategies as st
    base = _golden_bytes()[0]

    @settings(max_examples=300, deadline=None)
    @given(st.lists(st.tuples(st.integers(0, len(base) - 1), st.integers(0, 255)), min_size=1, max_size=8),
           st.integers(0, len(base)))
    def run(muts, cut):
        b = bytearray(base)
        for pos, val in muts:
            b[pos] = val
        data = bytes(b[:cut]) if cut % 3 == 0 else bytes(b)
        try:
            enc = glc_amd.EncodedAudio.from_bytes(data)
        except glc_amd.GlcError as e:
            assert e.code == -4
            return
        again = enc.to_bytes()
        assert glc_amd.EncodedAudio.from_bytes(again).to_bytes() == again
        assert len(again) <= len(data)

    run()


def test_host_parsers_under_address_and_ub_sanitizers(tmp_path):
    """tools/host_fuzz.cpp: the container, WAV and FLAC readers / writers built with
    -fsanitize=address,undefined (host code only; GPU sanitizers do not exist on the pool) and
    driven with mutated files plus their round-trip properties for a few seconds."""
    import subprocess
    csrc = os.path.join(ROOT, "gapless-lossy-codec_amd", "csrc")
    exe = tmp_path / "host_fuzz"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "host_fuzz.cpp")]
    cmd += [os.path.join(csrc, f) for f in ("glc_frames.cpp", "glc_tables.cpp", "glc_wav.cpp", "glc_flac.cpp")]
    cmd += ["-lpthread", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), "6", str(tmp_path), "0x5eed"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "rounds clean" in r.stdout, r.stderr[-3000:]


def test_one_hip_runtime_is_mapped():
    """torch's wheel bundles a libamdhip64 with the system ROCm's SONAME; glc_amd imports torch first, so
    the loader gives libglc_hip.so the runtime torch mapped: ONE runtime, stream handles may cross
    (tests/test_gpu_parity.py passes torch streams to glc_ctx_set_stream).  The other import order maps
    two; shown in a child process."""
    import subprocess
    import sys
    from glc_amd._lib import hip_runtimes_mapped
    assert len(hip_runtimes_mapped()) == 1, hip_runtimes_mapped()
    code = ("import ctypes, sys; sys.path.insert(0, %r); ctypes.CDLL(%r); import torch; import re; "
            "print(len({m.group(1) for l in open('/proc/self/maps') for m in [re.search(r'(/\\S*libamdhip64[^/\\s]*)', l)] if m}))"
            % (ROOT, glc_amd.LIB_PATH))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "2", out.stdout + out.stderr
