"""D1 (dequant + inverse MDCT + window) alone on the BASELINE config-2 batch: 4096 frames of the
bench's 48 kHz stereo chord, encoded once, then glc_imdct_device timed with HIP events for every
kernel variant of include/glc_debug.h.  Also the command to put under rocprofv3 (--kernel-trace /
--pmc) for the decode kernel.  Usage: python tools/bench_d1.py [reps] [variants, e.g. 0,2,3,1]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the workload generator)
import glc_amd  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2, 1]
glc_amd.lib.glc_debug_set_imdct_variant.restype = C.c_int
glc_amd.lib.glc_debug_set_imdct_variant.argtypes = [C.c_void_p, C.c_int]
SR, CH, NF = bench.SR, bench.CH, bench.FRAMES_PER_GPU
x = bench.chord(np, 0, NF * 1024)
ea = glc_amd.Encoder(SR).encode(x, CH)
nnz = ea.info().total_nnz
dec = glc_amd.Decoder(CH, SR)
d_blk = torch.empty((NF * CH, 2048), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(300):  # clocks up: the device idles while the host encodes and uploads
    dec.imdct_device(ea, 0, NF, d_blk.data_ptr())
ref = None
for v in variants:
    assert glc_amd.lib.glc_debug_set_imdct_variant(dec._h, v) == 0
    for _ in range(100):  # ... and again after the read-back of the previous variant's output
        dec.imdct_device(ea, 0, NF, d_blk.data_ptr())
    dec.timer_begin()
    for _ in range(reps):
        dec.imdct_device(ea, 0, NF, d_blk.data_ptr())
    ms = dec.timer_end() / reps
    dec.synchronize()
    h = d_blk.cpu().numpy().view(np.uint32)
    same = True if ref is None else bool(np.array_equal(ref, h))
    ref = h if ref is None else ref
    tf = nnz * 2048 * 2 / (ms * 1e-3) / 1e12
    print(f"variant {v}: {ms * 1e3:8.1f} us  {tf:6.2f} TFLOP/s = {tf / 78.65:5.3f} of the unfused ceiling  "
          f"(nnz/row {nnz / (NF * CH):.1f})  bits equal to first variant: {same}", flush=True)
