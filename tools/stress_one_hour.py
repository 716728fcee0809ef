"""One-off stress (BASELINE config 4, one rank's share at TRUE size): a 1-hour 96 kHz stereo stream
(691 200 000 samples, 337 500 frames) through Encoder::encode and Decoder::decode on one GPU.
The stream tiles a 60 s segment (= 5625 frames exactly), so frames repeat with period 5625:
a size-independent check of every frame beyond the first period; the first period is what
tests/test_gpu_parity.py::test_cfg4_long_96k_stream_shard pins against the oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import glc_amd  # noqa: E402
from conftest import gen_chord  # noqa: E402

sr, ch, minutes = 96000, 2, int(os.environ.get("GLC_STRESS_MINUTES", "60"))
seg = gen_chord(sr, ch, 60 * sr, n_tones=8)
x = np.tile(seg.reshape(-1, ch), (minutes, 1)).reshape(-1)
print(f"{x.size} samples ({x.nbytes / 1e9:.2f} GB), expecting {5625 * minutes} frames", flush=True)
enc = glc_amd.Encoder(sr)
t0 = time.perf_counter()
ea = enc.encode(x, ch)
t1 = time.perf_counter()
info = ea.info()
assert info.n_frames == 5625 * minutes and info.n_raw_frames == 0, (info.n_frames, info.n_raw_frames)
print(f"encode {t1 - t0:.2f} s ({x.size / (t1 - t0) / 1e6:.0f} Msamples/s incl. PCIe), nnz/row "
      f"{info.total_nnz / (info.n_frames * ch):.1f}", flush=True)
# periodicity: frame f and f + 5625 carry identical lists and scales (interior frames)
for f in (1, 17, 2812, 5623):
    a = ea.frames[f]
    for k in (1, minutes // 2, minutes - 1):
        b = ea.frames[f + 5625 * k]
        assert a.sparse_coeffs_per_channel == b.sparse_coeffs_per_channel and a.scale_factors == b.scale_factors
dec = glc_amd.Decoder(ch, sr)
t2 = time.perf_counter()
y = dec.decode(ea)
t3 = time.perf_counter()
assert y.size == x.size
print(f"decode {t3 - t2:.2f} s ({x.size / (t3 - t2) / 1e6:.0f} Msamples/s incl. PCIe)", flush=True)
per = 5625 * 1024 * ch
for k in (1, minutes // 2, minutes - 2):
    assert np.array_equal(y[per + 4096:2 * per - 4096].view(np.uint32),
                          y[(k + 1) * per + 4096:(k + 2) * per - 4096].view(np.uint32))
e = x[10000:2_000_000] - y[10000:2_000_000]
print("snr dB", 10 * np.log10((x[10000:2_000_000] ** 2).sum() / (e ** 2).sum()), "OK", flush=True)
