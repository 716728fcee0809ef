"""Writes the sparse rows of the BASELINE config-2 batch (4096 frames of the bench's 48 kHz stereo
chord, encoded on the GPU) to build/d1_rows.bin for tools/d1_tune.hip: u32 {n_frames, channels, rows,
n_pairs}, u64 begin[rows], u32 cnt[rows], f32 scale[rows], u32 pairs[n_pairs]; and the batch itself
(interleaved f32) to build/chord_cfg2.f32 for the C++ drivers tools/encode_breakdown.cpp / decode_breakdown.cpp.
Usage: python tools/dump_d1_rows.py [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import glc_amd  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else bench.FRAMES_PER_GPU
SR, CH = bench.SR, bench.CH
x = bench.chord(np, 0, nf * 1024)
enc = glc_amd.Encoder(SR)
rec = glc_amd.lib.glc_record_bytes(CH)
d_pcm = torch.from_numpy(x).cuda()
d_rec = torch.empty(nf * rec, dtype=torch.uint8, device="cuda")
cap = glc_amd.compact_bound(CH, nf)
d_blob = torch.empty(cap, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
enc.encode_range_device(d_pcm.data_ptr(), 0, nf * 1024, x.size, CH, 0, nf, d_rec.data_ptr())
info = enc.compact_device_records(d_rec.data_ptr(), nf, CH, d_blob.data_ptr(), cap)
blob = d_blob[:info.bytes].cpu().numpy()
M = nf * CH
a64 = lambda v: (v + 63) // 64 * 64
o_scale = 64 + a64(nf)
o_cnt = o_scale + a64(4 * M)
o_pairs = o_cnt + a64(4 * M)
scale = blob[o_scale:o_scale + 4 * M].view(np.float32)
cnt = blob[o_cnt:o_cnt + 4 * M].view(np.uint32)
pairs = blob[o_pairs:o_pairs + 4 * info.n_pairs].view(np.uint32)
begin = np.concatenate([[0], np.cumsum(cnt[:-1], dtype=np.uint64)]).astype(np.uint64)
os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
x.tofile(os.path.join(ROOT, "build", "chord_cfg2.f32"))  # the batch itself, for tools/{en,de}code_breakdown.cpp
with open(os.path.join(ROOT, "build", "d1_rows.bin"), "wb") as fh:
    fh.write(np.array([nf, CH, M, info.n_pairs], np.uint32).tobytes())
    fh.write(begin.tobytes())
    fh.write(cnt.tobytes())
    fh.write(scale.tobytes())
    fh.write(pairs.tobytes())
print(f"build/d1_rows.bin: {nf} frames x {CH} ch, {info.n_pairs} pairs ({info.n_pairs / M:.1f} per row), raw rows {info.n_raw_rows}")
