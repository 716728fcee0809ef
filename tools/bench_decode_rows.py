"""D1 (sparse IMDCT) timing on synthetic streams whose rows are correlated to a chosen degree:
`share` = fraction of a row's coefficient indices that it has in common with every other row (the
rest are drawn independently per row).  share = 1 is stationary tonal material, share = 0 the worst
case for the grouped kernel.  Usage: [GLC_D1_GROUP=0|2|4|8] python tools/bench_decode_rows.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import glc_amd  # noqa: E402

sr, ch, nf, nnz = 48000, 2, 4096, 114
rec = glc_amd.lib.glc_record_bytes(ch)
hdr = rec - 4096 * ch
rng = np.random.default_rng(3)
dec = glc_amd.Decoder(ch, sr)
for share in (1.0, 0.75, 0.5, 0.0):
    common = rng.choice(1024, int(round(nnz * share)), replace=False)
    rest = np.setdiff1d(np.arange(1024), common)
    buf = np.zeros((nf, rec), np.uint8)
    for f in range(nf):
        for c in range(ch):
            idx = np.concatenate([common, rng.choice(rest, nnz - common.size, replace=False)])
            pay = buf[f, hdr + c * 4096: hdr + (c + 1) * 4096].view(np.int16)
            pay[idx] = rng.integers(1, 3000, idx.size) * rng.choice([-1, 1], idx.size)
            buf[f, 8 + 8 * c:12 + 8 * c] = np.frombuffer(np.float32(0.3).tobytes(), np.uint8)
            buf[f, 12 + 8 * c:16 + 8 * c] = np.frombuffer(np.uint32(nnz).tobytes(), np.uint8)
    ea = glc_amd.EncodedAudio.from_records(sr, nf * 1024 * ch, ch, buf.reshape(-1))
    d_all = torch.empty((nf + 1) * 1024 * ch, dtype=torch.float32, device="cuda")
    for _ in range(3):
        dec.decode_device(ea, d_all.data_ptr(), d_all.numel())
    dec.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dec.decode_device(ea, d_all.data_ptr(), d_all.numel())
    dec.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"group {os.environ.get('GLC_D1_GROUP', 'default')}: shared indices {share:4.2f}  decode_device {ms:6.3f} ms per {nf} frames "
          f"(checksum {float(d_all.double().abs().sum()):.6e})", flush=True)
