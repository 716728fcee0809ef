#!/usr/bin/env python3
"""Why the shipped transform is a literal table contraction and not a butterfly / FFT MDCT (SURVEY
F2 / F3): every faster evaluation of the MDCT changes the quantised integers and
the scale-factor bits of the .glc stream, so it is a different codec, not a faster one.

The reference computes (src/codec.rs:326-338, :359-374)
    T[k][i] = cosf(fl(fl(fl(PI/1024) * (i + 0.5 + 512)) * (k + 0.5)))     angle rounded to f32 (up to 8037 rad)
    out[k]  = fl(fl(sum over ascending i of fl(b[i] * T[k][i])) * norm)    separately rounded mul and add

This script (numpy + the CPU oracle, no GPU) feeds the oracle's own windowed blocks through three
other evaluations of "the same" transform and then through the oracle's scale / masking-threshold /
quantiser code, and counts what changes against the oracle:
    true_cosine_f64   exact cosine of the exact angle, f64 accumulation, rounded to f32 once at the end -
                      what an ideal butterfly MDCT converges to (any f32 FFT is further away)
    fused_f32         the reference's table and order, but s = fma(b[i], T[k][i], s) (one rounding per term)
    split_k2_f32      the reference's table, unfused, two interleaved accumulators added at the end
                      (the smallest possible re-association: what a split-K or tree reduction does)
Prints one JSON line per (signal, variant); tests/test_oracle.py asserts the fractions are non-zero.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402  (test infrastructure; this script is not on the product path)

F32 = np.float32
HOP, FRAME = 1024, 2048


def windowed_blocks(x, ch):
    """The blocks Encoder::encode hands to mdct_block (src/codec.rs:426-481): deinterleave, 512 leading
    zeros, zero tail, block[i] = fl(slice[i] * window[i]).  Rows ordered (frame, channel)."""
    _, w, _ = O.tables()
    nf = O.num_frames(x.size, ch)
    per = [x[c::ch] for c in range(ch)]
    L = len(per[0])
    padded_len = (nf + 1) * HOP + HOP  # enough zeros behind the data for every frame
    rows = np.zeros((nf * ch, FRAME), F32)
    for c in range(ch):
        p = np.zeros(max(padded_len, 512 + L + FRAME), F32)
        p[512:512 + len(per[c])] = per[c]
        for f in range(nf):
            rows[f * ch + c] = (p[f * HOP:f * HOP + FRAME] * w).astype(F32)
    return rows


def variant_coeffs(rows, which):
    T, _, norm = O.tables()          # T[k][i] f32, the reference's table
    if which == "reference_order_f32":   # sanity: must reproduce the oracle's coefficients bit for bit
        s = np.zeros((rows.shape[0], HOP), F32)
        for i in range(FRAME):
            s = (s + (rows[:, i, None] * T[None, :, i]).astype(F32)).astype(F32)
        return (s * norm).astype(F32)
    if which == "true_cosine_f64":
        i = np.arange(FRAME, dtype=np.float64)[None, :]
        k = np.arange(HOP, dtype=np.float64)[:, None]
        Tt = np.cos(np.pi / 1024.0 * (i + 0.5 + 512.0) * (k + 0.5))
        s = rows.astype(np.float64) @ Tt.T
        return (s * np.sqrt(2.0 / 1024.0)).astype(F32)
    if which == "fused_f32":
        # fma through f64: the product of two f32 is exact in f64; the f64 sum is rounded once more to
        # f32 (double rounding can differ from a hardware fma in ~1e-9 of the terms - irrelevant here)
        s = np.zeros((rows.shape[0], HOP), F32)
        T64 = T.astype(np.float64)
        for i in range(FRAME):
            s = (rows[:, i, None].astype(np.float64) * T64[None, :, i] + s.astype(np.float64)).astype(F32)
        return (s * norm).astype(F32)
    if which == "split_k2_f32":
        s0 = np.zeros((rows.shape[0], HOP), F32)
        s1 = np.zeros((rows.shape[0], HOP), F32)
        for i in range(0, FRAME, 2):
            s0 = (s0 + (rows[:, i, None] * T[None, :, i]).astype(F32)).astype(F32)
            s1 = (s1 + (rows[:, i + 1, None] * T[None, :, i + 1]).astype(F32)).astype(F32)
        return ((s0 + s1).astype(F32) * norm).astype(F32)
    raise ValueError(which)


def quantise(coeffs, sr):
    """The oracle's own scale / thresholds / quantiser on given coefficients -> (scale bits, dense q)."""
    weights, edges = O.perceptual(sr)
    M = coeffs.shape[0]
    scales = np.empty(M, F32)
    dense = np.zeros((M, HOP), np.int16)
    for m in range(M):
        c = coeffs[m]
        scale = F32(max(F32(np.max(np.abs(c))), F32(1e-10)))  # :488 fold(0, max).max(1e-10)
        thr = O.thresholds(c, weights, edges)
        idx, q = O.compress(c, scale, thr)
        scales[m] = scale
        dense[m, idx] = q
    return scales, dense


def compare(x, sr, ch, variants=("true_cosine_f64", "fused_f32", "split_k2_f32")):
    ref = O.encode(x, sr, ch, taps=True)
    rows = windowed_blocks(x, ch)
    out = {}
    # the harness itself: the oracle's order in numpy reproduces the oracle's coefficients exactly
    chk = variant_coeffs(rows[:8], "reference_order_f32")
    assert np.array_equal(chk.view(np.uint32), ref.coeffs[:8].view(np.uint32)), "harness does not reproduce the oracle"
    s_ref, q_ref = quantise(ref.coeffs, sr)
    assert np.array_equal(s_ref.view(np.uint32), ref.scales.view(np.uint32)) and np.array_equal(q_ref, ref.dense_q)
    for v in variants:
        c = variant_coeffs(rows, v)
        s, q = quantise(c, sr)
        stored = (q != 0) | (q_ref != 0)
        diff = (q != q_ref) & stored
        ulp = np.abs(c.view(np.int32).astype(np.int64) - ref.coeffs.view(np.int32).astype(np.int64))
        out[v] = {
            "rows": int(c.shape[0]),
            "coeff_words_differing": float((c.view(np.uint32) != ref.coeffs.view(np.uint32)).mean()),
            "median_coeff_ulp": float(np.median(ulp)),
            "scale_bits_differing": float((s.view(np.uint32) != s_ref.view(np.uint32)).mean()),
            "stored_positions": int(stored.sum()),
            "quantised_ints_differing": float(diff.sum() / max(1, stored.sum())),
            "max_abs_q_delta": int(np.abs(q.astype(np.int32) - q_ref.astype(np.int32))[stored].max()) if stored.any() else 0,
            "kept_set_differs_rows": float(((q != 0) != (q_ref != 0)).any(axis=1).mean()),
        }
    return out


def signals():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import gen_noise, gen_tone
    yield "cfg1_sine440_44k_stereo_2s", gen_tone("sine", 440.0, 44100, 2, 2.0), 44100, 2
    yield "lcg_noise_44k_stereo_1s", gen_noise(44100, 2, 1.0, 12345), 44100, 2


def main():
    for name, x, sr, ch in signals():
        for v, r in compare(x, sr, ch).items():
            print(json.dumps({"signal": name, "variant": v, **r}))


if __name__ == "__main__":
    main()
