"""Soak of the structured bridge (include/glc.h): random streams - 1..8 channels, 1..9000 frames, tonal /
noisy / mixed content so that raw frames come and go - are encoded with glc_encode_hooked (the hook
rebuilds the reference's nested EncodedFrame vectors range by range), rebuilt through
glc_frames_from_gather and glc_frames_from_parts, decoded first-sight, by stream id and by
glc_decode_resident; everything must equal the plain glc_encode / glc_decode of the same input (bytes
and f32 bits).  GPU against GPU: the oracle is not involved (tests/test_bridge.py pins the bridge to it).
Usage: python tools/soak_bridge.py [cases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import glc_amd  # noqa: E402
from glc_amd import EncodedAudio  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
t0 = time.time()
encs, decs = {}, {}
for case in range(cases):
    rng = np.random.default_rng(424200 + case)
    sr = int(rng.choice([22050, 44100, 48000, 96000]))
    ch = int(rng.choice([1, 1, 2, 2, 2, 3, 4, 6, 8]))
    frames = int(rng.integers(1, 400)) if rng.random() < 0.6 else int(rng.integers(400, 9000 // ch + 400))
    n_per = frames * 1024 + int(rng.integers(0, 1024))
    t = np.arange(n_per, dtype=np.float64)[:, None]
    x = (np.sin(2 * np.pi * rng.uniform(60, 9000, (1, ch)) * t / sr) * rng.uniform(0.05, 0.5)).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):      # noise bursts -> raw frames
        a = int(rng.integers(0, n_per))
        b = min(n_per, a + int(rng.integers(500, 20000)))
        x[a:b] = rng.standard_normal((b - a, ch)).astype(np.float32) * 0.3
    x = x.reshape(-1)
    if n_per <= 512:
        continue
    enc = encs.setdefault(sr, glc_amd.Encoder(sr))
    dec = decs.setdefault(sr, glc_amd.Decoder(ch, sr))
    plain = enc.encode(x, ch)
    want = plain.to_bytes()
    pcm = dec.decode(plain).copy()
    seen, nested = [], []

    def hook(parts, f0, f1):
        seen.append((f0, f1))
        for i in range(f0, f1):
            l0, l1 = int(parts["list_begin"][i]), int(parts["list_begin"][i + 1])
            lists = [parts["pairs"][int(parts["list_off"][l]):int(parts["list_off"][l + 1])].copy() for l in range(l0, l1)]
            sc = parts["scales"][int(parts["scale_begin"][i]):int(parts["scale_begin"][i + 1])].copy()
            raw = parts["raw"][int(parts["raw_begin"][i]):int(parts["raw_begin"][i + 1])].copy() if parts["raw_tag"][i] else None
            nested.append((lists, sc, raw))
        return 0
    hooked = enc.encode_hooked(x, ch, hook)
    ok = hooked.to_bytes() == want
    nf = plain.info().n_frames
    ok &= seen[0][0] == 0 and seen[-1][1] == nf and all(a[1] == b[0] for a, b in zip(seen, seen[1:])) and len(nested) == nf
    sid = 1 + case
    g = EncodedAudio.from_nested(plain.header, nested, plain.gapless_info, stream_id=sid)
    ok &= g.to_bytes() == want
    ok &= np.array_equal(dec.decode(g).view(np.uint32), pcm.view(np.uint32)) and dec.resident_stream() == sid
    p = EncodedAudio.from_parts(plain.parts(), sid)
    ok &= p.to_bytes() == want
    ok &= np.array_equal(dec.decode(p).view(np.uint32), pcm.view(np.uint32))          # same id: resident rows + kept plan
    buf = np.empty(pcm.size, np.float32)
    ok &= np.array_equal(dec.decode_resident(sid, buf).view(np.uint32), pcm.view(np.uint32))
    if not ok:
        bad += 1
        print(f"case {case}: sr {sr} ch {ch} frames {nf}: MISMATCH", flush=True)
    if case % 25 == 24 or case == cases - 1:
        print(f"bridge soak case {case + 1}: {bad} mismatching cases so far ({time.time() - t0:.0f} s)", flush=True)
sys.exit(1 if bad else 0)
