"""D1 (sparse IMDCT) timing on synthetic streams whose rows are correlated to a chosen degree:
`share` = fraction of a row's coefficient indices that it has in common with every other row (the
rest are drawn independently per row).  share = 1 is stationary tonal material, share = 0 the worst
case for the grouped kernels.  Every kernel variant of include/glc_debug.h is timed (D1 alone, HIP
events).  Usage: python tools/bench_decode_rows.py [variants, default 0,4,1]"""
import ctypes as C
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
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 2, 1]
glc_amd.lib.glc_debug_set_imdct_variant.restype = C.c_int
glc_amd.lib.glc_debug_set_imdct_variant.argtypes = [C.c_void_p, C.c_int]
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
    d_blk = torch.empty((nf * ch, 2048), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    out = []
    for _ in range(300):  # the device clocked down while the host built the stream
        dec.imdct_device(ea, 0, nf, d_blk.data_ptr())
    for v in variants:
        assert glc_amd.lib.glc_debug_set_imdct_variant(dec._h, v) == 0
        for _ in range(30):
            dec.imdct_device(ea, 0, nf, d_blk.data_ptr())
        dec.timer_begin()
        for _ in range(20):
            dec.imdct_device(ea, 0, nf, d_blk.data_ptr())
        out.append(f"variant {v}: {dec.timer_end() / 20 * 1e3:7.1f} us")
    print(f"shared indices {share:4.2f}  D1 per {nf} stereo frames (nnz/row {nnz}):  " + "   ".join(out), flush=True)
