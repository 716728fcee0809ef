"""Soak of the decode kernels: random sparse streams (random list lengths 0..400, a random share of
common indices, random raw frames, 1..3 channels) are decoded three ways - the shipped plan + apply
kernels (absent row pairs skipped), the same without the skip, and the
one-row kernel (include/glc_debug.h) - and all outputs must be bit-identical.
Usage: python tools/soak_decode.py [rounds]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import glc_amd  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
glc_amd.lib.glc_debug_set_imdct_variant.restype = C.c_int
glc_amd.lib.glc_debug_set_imdct_variant.argtypes = [C.c_void_p, C.c_int]
dec = glc_amd.Decoder(2, 48000)
bad = 0
t0 = time.time()
for r in range(rounds):
    rng = np.random.default_rng(1000 + r)
    ch = int(rng.choice([1, 2, 2, 2, 3]))
    nf = int(rng.choice([2048, 2048, 1027, 333]))
    M = nf * ch
    rec = glc_amd.lib.glc_record_bytes(ch)
    hdr = rec - 4096 * ch
    nnz = rng.integers(0, 401, M)
    nnz[rng.random(M) < 0.05] = 0
    order = np.argsort(rng.random((M, 1024)), axis=1)                 # a random permutation of the bins per row
    common = rng.permutation(1024)
    share = rng.random()
    use_common = rng.random(M) < share                                # these rows take the same leading bins
    order[use_common] = common
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(1024)[None, :].repeat(M, 0), axis=1)
    keep = rank < nnz[:, None]
    q = rng.integers(1, 20000, (M, 1024)).astype(np.int16) * rng.choice(np.array([-1, 1], np.int16), (M, 1024))
    q[~keep] = 0
    raw_frame = rng.random(nf) < 0.1
    buf = np.zeros((nf, rec), np.uint8)
    pay = buf[:, hdr:].view(np.int16).reshape(nf, ch, 2048)
    pay[:, :, :1024] = q.reshape(nf, ch, 1024)
    pay[raw_frame] = rng.integers(-32768, 32768, (int(raw_frame.sum()), ch, 2048)).astype(np.int16)
    meta = buf[:, 8:8 + 8 * ch].view(np.uint32).reshape(nf, ch, 2)
    meta[:, :, 0] = rng.uniform(1e-3, 1.0, (nf, ch)).astype(np.float32).view(np.uint32)
    meta[:, :, 1] = keep.sum(1).reshape(nf, ch)
    buf[:, 0:4].view(np.uint32)[:, 0] = raw_frame
    ea = glc_amd.EncodedAudio.from_records(48000, nf * 1024 * ch, ch, buf.reshape(-1))
    outs = []
    for variant in (0, 1, 2, 3, 4, 0):  # the last pass reuses the kept plan into a fresh (NaN) buffer
        assert glc_amd.lib.glc_debug_set_imdct_variant(dec._h, variant) == 0
        d = torch.full(((nf + 1) * 1024 * ch,), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        dec.decode_device(ea, d.data_ptr(), d.numel())
        dec.synchronize()
        outs.append(d.cpu().numpy().view(np.uint32))
    if not all(np.array_equal(outs[0], o) for o in outs[1:]):
        bad += 1
        print(f"round {r} (ch {ch}, {nf} frames): outputs differ in "
              f"{[int((outs[0] != o).sum()) for o in outs[1:]]} samples (one-row / no-skip / no-priority)", flush=True)
    if r % 10 == 9 or r == rounds - 1:
        print(f"decode soak round {r + 1}: {bad} differing streams so far ({time.time() - t0:.0f} s)", flush=True)
glc_amd.lib.glc_debug_set_imdct_variant(dec._h, 0)
sys.exit(1 if bad else 0)
