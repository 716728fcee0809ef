"""CPU tests: the C oracle against the independent numpy restatement and against the
properties the reference's own tests assert (tests/*.rs hold no golden vectors: SURVEY.md §4).
Parity of the oracle with the reference is therefore UNPINNED beyond these checks."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT, calculate_snr, gen_noise, gen_tone, parse_glc
from oracle import glc_oracle_np as NP
from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_tables_match_numpy_restatement():
    T, w, n = O.tables()
    T2, w2, n2 = NP.tables_cached()
    assert np.array_equal(bits(T), bits(T2))
    assert np.array_equal(bits(w), bits(w2))
    assert n == n2
    assert float(T.max()) <= 1.0 and T.shape == (1024, 2048)


def test_table_is_not_an_exact_cosine():
    # the libm-rounded table differs from the rounded f64 cosine (1 ulp cases),
    # so the libm is part of the codec definition.  Pin the count for this image's glibc.
    T, _, _ = O.tables()
    k = np.arange(1024, dtype=np.float32)[:, None] + np.float32(0.5)
    i = np.arange(2048, dtype=np.float32)
    pre = (np.float32(np.pi) / np.float32(1024)) * ((i + np.float32(0.5)) + np.float32(512))
    ang = (pre.astype(np.float32)[None, :] * k).astype(np.float32)
    exact = np.cos(ang.astype(np.float64)).astype(np.float32)
    assert float(ang.max()) > 8000.0
    assert np.abs(T.astype(np.float64) - exact).max() < 1.3e-7
    assert (bits(T) != bits(exact)).sum() > 0


@pytest.mark.parametrize("sr", [8000, 22050, 44100, 48000, 96000, 192000])
def test_perceptual_tables(sr):
    w, e = O.perceptual(sr)
    w2, e2 = NP.perceptual(sr)
    assert np.array_equal(bits(w), bits(w2)) and np.array_equal(e, e2)
    assert e[0] == 0 and e[-1] == 1024 and len(e) <= 51 and np.all(np.diff(e.astype(int)) > 0)
    assert w.min() >= 0.2 and w.max() <= 1.0


def test_band_counts_match_survey_probe():
    # last band 653 bins @44.1k, 683 @48k, 854 @96k, 912 @192k; 51 edges at all four
    for sr, last in [(44100, 653), (48000, 683), (96000, 854), (192000, 912)]:
        _, e = O.perceptual(sr)
        assert len(e) == 51 and 1024 - int(e[-2]) == last


CASES = [
    ("sine_44k_stereo", lambda: gen_tone("sine", 440.0, 44100, 2, 2.0), 44100, 2),   # BASELINE cfg 1
    ("noise_44k_stereo", lambda: gen_noise(44100, 2, 0.5, 12345), 44100, 2),
    ("sweep_48k_mono", lambda: gen_tone("sweep", 100.0, 48000, 1, 1.0, 10000.0), 48000, 1),
    ("square_44k_mono", lambda: gen_tone("square", 1000.0, 44100, 1, 0.5), 44100, 1),
    ("denormals", lambda: _special("denormal"), 44100, 2),
    ("huge_values", lambda: _special("huge"), 48000, 1),
    ("nan_inf", lambda: _special("nan"), 48000, 2),
    ("impulses_5ch", lambda: _special("impulse"), 96000, 5),
]


def _special(kind):
    from test_gpu_parity import _special as sp
    return sp(kind)


@pytest.mark.parametrize("name,make,sr,ch", CASES, ids=[c[0] for c in CASES])
def test_c_oracle_equals_numpy_restatement(name, make, sr, ch):
    x = make()
    a = O.encode(x, sr, ch, taps=True)
    b = NP.encode(x, sr, ch)
    assert np.array_equal(bits(a.coeffs), bits(b["coeffs"]))
    assert np.array_equal(bits(a.scales), bits(b["scales"]))
    assert np.array_equal(a.dense_q, b["dense_q"])
    assert np.array_equal(a.nnz, b["nnz"])
    assert np.array_equal(a.is_raw, b["is_raw"])
    assert a.glc == b["glc"]
    d1, sr1, ch1 = O.decode(a.glc)
    d2 = NP.decode(a.glc)
    assert (sr1, ch1) == (sr, ch)
    assert np.array_equal(bits(d1), bits(d2))
    assert d1.size == x.size


def test_structural_known_answers():
    # derivable from src/codec.rs alone
    assert O.num_frames(176400, 2) == 86          # 2 s @ 44.1 kHz stereo
    assert O.num_frames(88200, 2) == 43
    assert O.num_frames(8388608, 2) == 4096       # BASELINE cfg 2
    assert O.num_frames(57600000, 2) == 28125     # cfg 3
    assert O.num_frames(691200000, 2) == 337500   # one cfg 4 stream
    assert O.num_frames(512 * 2, 2) == 0          # <= 512 per channel: reference panics
    assert O.num_frames(513 * 2, 2) == 1
    assert O.num_frames(100, 0) == 0
    x = gen_tone("sine", 440.0, 44100, 2, 2.0)
    g = parse_glc(O.encode(x, 44100, 2).glc)
    assert g["encoder_delay"] == 512 and g["padding"] == 888 and g["original_length"] == 176400
    assert len(g["frames"]) == 86 and g["total_samples"] == 176400


def test_reference_property_tests_hold_for_oracle():
    """The assertions of the reference's tests/test_codec.rs, test_simple.rs,
    test_compression_ratio.rs restated (length equality, SNR bounds, sparsity)."""
    for kind, f, sr, ch, dur, bound in [("sine", 440.0, 44100, 1, 2.0, -10.0),     # test_codec.rs:8-24
                                        ("square", 1000.0, 44100, 1, 2.0, -15.0),  # :27-44
                                        ("sawtooth", 440.0, 44100, 1, 2.0, -10.0), # :47-64
                                        ("sine", 440.0, 48000, 1, 1.0, -10.0),
                                        ("sine", 440.0, 44100, 2, 2.0, -10.0)]:    # :89-108
        x = gen_tone(kind, f, sr, ch, dur)
        enc = O.encode(x, sr, ch)
        dec, _, _ = O.decode(enc.glc)
        assert dec.size == x.size
        assert calculate_snr(x, dec) > bound
    # test_compression_ratio.rs:33 — fewer than 50 % of coefficients kept for a 440 Hz mono sine
    x = gen_tone("sine", 440.0, 44100, 1, 2.0)
    enc = O.encode(x, 44100, 1, taps=True)
    assert enc.nnz.sum() / (enc.n_frames * 1024) < 0.5


def test_white_noise_takes_raw_fallback():
    x = gen_noise(44100, 2, 1.0, 12345)
    enc = O.encode(x, 44100, 2, taps=True)
    assert enc.is_raw.all() and enc.n_frames == 43        # SURVEY §4 probe: 43/43 raw
    g = parse_glc(enc.glc)
    assert all(f["raw"] is not None and f["raw"].size == 2048 * 2 and not f["lists"] for f in g["frames"])


def test_golden_fixture_matches_oracle():
    """tests/golden/*.json were produced by tests/golden/make_golden.py from the oracle (the
    reference holds no vectors and cannot be built here); they freeze today's behaviour so that a
    libm / compiler drift on another box is detected."""
    with open(os.path.join(GOLDEN, "golden.json")) as fh:
        gold = json.load(fh)
    T, w, n = O.tables()
    assert hashlib.sha256(T.tobytes()).hexdigest() == gold["tables"]["cos_table_sha256"]
    assert hashlib.sha256(w.tobytes()).hexdigest() == gold["tables"]["window_sha256"]
    for case in gold["cases"]:
        x = _golden_input(case)
        enc = O.encode(x, case["sample_rate"], case["channels"], taps=True)
        assert len(enc.glc) == case["glc_len"]
        assert hashlib.sha256(enc.glc).hexdigest() == case["glc_sha256"]
        assert enc.nnz.tolist() == case["nnz"]
        assert [int(v) for v in bits(enc.scales)] == case["scale_bits"]
        dec, _, _ = O.decode(enc.glc)
        assert hashlib.sha256(dec.tobytes()).hexdigest() == case["decoded_sha256"]
        with open(os.path.join(GOLDEN, case["glc_file"]), "rb") as fh:
            assert fh.read() == enc.glc


def _golden_input(case):
    g = case["generator"]
    if g["kind"] == "noise":
        return gen_noise(case["sample_rate"], case["channels"], g["dur"], g["seed"])
    return gen_tone(g["kind"], g["f0"], case["sample_rate"], case["channels"], g["dur"], g.get("f1", 0.0))


def test_faster_transforms_change_the_bitstream():
    """SURVEY F2 / F3, reproducible from the repo (tools/f2_evidence.py; numbers quoted in DESIGN.md
    section 2): the north_star's butterfly / FFT MDCT converges to the TRUE cosine transform, the
    reference evaluates an f32 table of cosf(f32-rounded angle) with separately rounded multiply and
    add in ascending order (src/codec.rs:326-338, :359-374).  Feeding the oracle's own scale /
    threshold / quantiser (src/codec.rs:188-311) with the true-cosine coefficients, with an FMA chain
    over the reference's table, or with the smallest re-association (two accumulators) changes
    scale-factor bits - which the .glc stores verbatim - and quantised integers.  None of them can
    produce the reference's bytes; that is why K1 is an exact-order contraction."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import f2_evidence as F2
    seen = {}
    for name, x, sr, ch in F2.signals():
        seen[name] = F2.compare(x, sr, ch)
    sine, noise = seen["cfg1_sine440_44k_stereo_2s"], seen["lcg_noise_44k_stereo_1s"]
    # the ideal transform: every scale factor differs, on both signals; broadband content loses most integers
    assert sine["true_cosine_f64"]["scale_bits_differing"] == 1.0 and noise["true_cosine_f64"]["scale_bits_differing"] == 1.0
    assert sine["true_cosine_f64"]["quantised_ints_differing"] > 0.005
    assert noise["true_cosine_f64"]["quantised_ints_differing"] > 0.5 and noise["true_cosine_f64"]["max_abs_q_delta"] > 9
    # same table, fused or re-associated: still not the reference's stream
    for v in ("fused_f32", "split_k2_f32"):
        for s in (sine, noise):
            assert s[v]["scale_bits_differing"] > 0.25, (v, s[v])
            assert s[v]["quantised_ints_differing"] > 0.0, (v, s[v])
