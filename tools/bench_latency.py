"""Latency of the host-boundary calls on short clips (the reference's own test sizes)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import glc_amd  # noqa: E402
from conftest import gen_tone  # noqa: E402

for dur in (0.1, 0.5, 2.0, 10.0):
    sr, ch = 44100, 2
    x = gen_tone("sine", 440.0, sr, ch, dur)
    enc, dec = glc_amd.Encoder(sr), glc_amd.Decoder(ch, sr)
    ea = enc.encode(x, ch)
    dec.decode(ea)
    te, td = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        ea = enc.encode(x, ch)
        t1 = time.perf_counter()
        dec.decode(ea)
        t2 = time.perf_counter()
        te.append(t1 - t0)
        td.append(t2 - t1)
    print(f"{dur:5.1f} s clip ({ea.info().n_frames:4d} frames): encode {1e3 * min(te):7.3f} ms  decode {1e3 * min(td):7.3f} ms  "
          f"(x realtime: {dur / min(te):9.0f} / {dur / min(td):9.0f})")
