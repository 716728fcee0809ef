"""Host-boundary timing of BASELINE config 3 (10 min 48 kHz stereo): glc_encode and glc_decode
wall time including H2D / D2H and host assembly (never the headline `value`)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import glc_amd  # noqa: E402
from conftest import gen_chord  # noqa: E402

sr, ch = 48000, 2
seg = gen_chord(sr, ch, 480000)
x = np.tile(seg.reshape(-1, ch), (60, 1)).reshape(-1)
enc = glc_amd.Encoder(sr)
dec = glc_amd.Decoder(ch, sr)
reuse = np.zeros(x.size, np.float32)  # a destination whose pages exist already
for rep in range(6):
    t0 = time.perf_counter()
    ea = enc.encode(x, ch)
    t1 = time.perf_counter()
    out = dec.decode(ea) if rep < 3 else dec.decode(ea, out=reuse)   # fresh pages vs reused buffer
    t2 = time.perf_counter()
    data = ea.to_bytes()
    t3 = time.perf_counter()
    print(f"rep {rep}: encode {1e3*(t1-t0):8.1f} ms ({x.size/(t1-t0)/1e6:8.1f} Msamples/s)  "
          f"decode {1e3*(t2-t1):8.1f} ms ({x.size/(t2-t1)/1e6:8.1f} Msamples/s)  serialize {1e3*(t3-t2):6.1f} ms  "
          f".glc {len(data)/1e6:.1f} MB  nnz/frame-ch {ea.info().total_nnz/(ea.info().n_frames*ch):.1f}")
assert out.size == x.size
