"""End-to-end soak against the CPU oracle (checker only): seeded random configurations (sample rate,
channels, ragged lengths, content mix from silence to clipping noise) through Encoder::encode and
Decoder::decode; `.glc` bytes and decoded PCM bits must equal the oracle's every time.  With the word
`pipeline` every case is a many-channel stream of two to four encode rounds, so that glc_encode runs
as its three-thread pipeline (one round runs on the calling thread alone).
Usage: python tools/soak_encode.py [cases] [pipeline]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import glc_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pipeline = len(sys.argv) > 2 and sys.argv[2] == "pipeline"
rates = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000, 12345]
encs, decs = {}, {}
bad = 0
t0 = time.time()
for case in range(cases):
    rng = np.random.default_rng(777000 + case)
    sr = int(rng.choice(rates))
    ch = int(rng.choice([1, 1, 2, 2, 2, 3, 4, 5, 6, 8]))
    n_per = int(rng.integers(513, 30000)) if rng.random() < 0.9 else int(rng.integers(30000, 600000))
    if pipeline:  # opening rounds of 2048 // ch frames, later rounds 4096: 2 .. 4 rounds
        ch = int(rng.choice([8, 12, 16, 24]))
        first = -(-4096 // ch)
        frames = 2 * first + int(rng.integers(1, 300)) if rng.random() < 0.8 else first + 4096 + int(rng.integers(1, 600))
        n_per = frames * 1024 - int(rng.integers(0, 1024))
    n = max(n_per * ch - int(rng.integers(0, ch)), 513 * ch)
    t = np.arange(n_per + 1, dtype=np.float64)[:, None]
    kind = int(rng.integers(0, 5 if pipeline else 6))  # kind 5 is too costly to generate at pipeline sizes
    if kind == 0:
        x = np.sin(2 * np.pi * rng.uniform(30, sr / 2.2, (1, ch)) * t / sr) * 0.5
    elif kind == 1:
        x = rng.standard_normal((n_per + 1, ch)) * 0.3
    elif kind == 2:
        x = np.sin(2 * np.pi * 440.0 * t / sr) * np.ones((1, ch))
        a, b = n_per // 3, 2 * n_per // 3
        x[a:b] = rng.standard_normal((b - a, ch)) * 0.2
        x[b:] = 0.0
    elif kind == 3:
        x = np.zeros((n_per + 1, ch))
        x[rng.integers(0, n_per, 40), rng.integers(0, ch, 40)] = rng.uniform(-1, 1, 40)
    elif kind == 4:
        x = np.sin(2 * np.pi * (50.0 + 0.2 * t) * t / sr) * 0.4 + 0.1
    else:
        x = sum(np.sin(2 * np.pi * f * t / sr + p) for f, p in zip(rng.uniform(50, sr / 2.5, 12), rng.uniform(0, 6.28, 12))) \
            * np.ones((1, ch)) * 0.05
    amp = float(rng.choice([1e-6, 1e-3, 0.3, 1.0, 3.0]))
    x = (x * amp).astype(np.float32).reshape(-1)[:n]
    enc = encs.setdefault(sr, glc_amd.Encoder(sr))
    dec = decs.setdefault(sr, glc_amd.Decoder(ch, sr))
    try:
        ref = O.encode(x, sr, ch)
    except ValueError:   # an input the reference panics on: the product must refuse it too
        try:
            enc.encode(x, ch)
            bad += 1
            print(f"case {case}: product accepted an input the reference panics on", flush=True)
        except glc_amd.GlcError as e:
            assert e.code == -1
            panics = globals().get("panics", 0) + 1
        continue
    ea = enc.encode(x, ch)
    ok = ea.to_bytes() == ref.glc
    if ok:
        dref, _, _ = O.decode(ref.glc)
        ok = np.array_equal(dec.decode(ea).view(np.uint32), dref.view(np.uint32))
    if not ok:
        bad += 1
        print(f"case {case}: MISMATCH sr={sr} ch={ch} n={n} kind={kind} amp={amp}", flush=True)
    if case % (5 if pipeline else 50) == (4 if pipeline else 49) or case == cases - 1:
        print(f"encode soak case {case + 1}: {bad} mismatching cases so far, {globals().get('panics', 0)} inputs refused by both "
              f"({time.time() - t0:.0f} s)", flush=True)
sys.exit(1 if bad else 0)
