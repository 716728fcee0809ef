"""Frame-range sharding of one encode across ranks, and the single gather that ends it.

Frames are independent (the rayon loop at src/codec.rs:462): rank r encodes the contiguous frame
range [f0_r, f1_r) from the slice of interleaved PCM that range reads — per-channel samples
[1024*f0 - 512, 1024*f1 + 512) clipped to the stream, i.e. the range plus a halo shared with its
neighbours, supplied at upload time (no device exchange).  Every rank emits fixed-size frame
records, compacts them on its own device (glc_compact_device_records: the bitstream's payload,
about 1/8 of the records) and ONE gather of the compact blobs to rank 0 is the only data-path
collective (`gather_compact`; RCCL over xGMI when the process group is `nccl`, gloo in the CPU
tests; the blobs differ in size, so a 8-byte-per-rank size exchange precedes the payload).  Rank 0
assembles EncodedAudio with glc_frames_from_compact.  `gather_records` moves the dense records
instead (kept for hosts that want them).  The same gather serves the other sharding of SURVEY 8e,
one whole stream per rank (`assemble_streams`).  Decode shards over hop ranges (`hop_ranges`), each
rank recomputing one halo frame.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

HOP = 1024
FRAME = 2048


@dataclass(frozen=True)
class Shard:
    rank: int
    frame_begin: int
    frame_end: int
    t0: int        # first per-channel sample index the shard holds
    t_count: int   # per-channel samples the shard holds

    @property
    def n_frames(self) -> int:
        return self.frame_end - self.frame_begin


def frame_ranges(n_frames: int, world: int) -> List[range]:
    """Contiguous, balanced partition: the first n_frames % world ranks get one extra frame."""
    base, extra = divmod(n_frames, world)
    out, f = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append(range(f, f + n))
        f += n
    return out


def hop_ranges(n_frames: int, world: int) -> List[range]:
    """Decode side: the un-trimmed output has n_frames + 1 hops of 1024 samples per channel (the
    last one is the overlap tail); rank r decodes a contiguous hop range with
    glc_decode_range_device, which recomputes the one halo frame its overlap-add needs, and the
    ranks' PCM concatenates in rank order (one gather of PCM, no other exchange)."""
    return frame_ranges(n_frames + 1, world)


def plan_shards(n_frames: int, per_channel: int, world: int) -> List[Shard]:
    """Shard r = frames + the PCM span (with halo) they read, clipped to [0, per_channel)."""
    shards = []
    for r, fr in enumerate(frame_ranges(n_frames, world)):
        if len(fr) == 0:
            shards.append(Shard(r, fr.start, fr.stop, 0, 0))
            continue
        lo = max(0, fr.start * HOP - HOP // 2)
        hi = min(per_channel, (fr.stop - 1) * HOP - HOP // 2 + FRAME)
        hi = max(hi, lo)
        shards.append(Shard(r, fr.start, fr.stop, lo, hi - lo))
    return shards


def gather_records(local_records, shards: List[Shard], record_bytes: int, group=None, dst: int = 0):
    """The single collective: gather every rank's record bytes (a flat uint8 torch tensor, on
    the GPU for nccl / on the CPU for gloo) to `dst`.  Shards may differ by one frame, so every
    rank pads to the largest shard; `dst` strips the padding and returns the records of all
    frames in frame order as one flat tensor (None on other ranks)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    assert world == len(shards)
    max_bytes = max(s.n_frames for s in shards) * record_bytes
    send = local_records.reshape(-1)
    assert send.dtype == torch.uint8 and send.numel() == shards[rank].n_frames * record_bytes
    if send.numel() < max_bytes:
        pad = torch.zeros(max_bytes - send.numel(), dtype=torch.uint8, device=send.device)
        send = torch.cat([send, pad])
    recv: Optional[list] = None
    if rank == dst:
        recv = [torch.empty(max_bytes, dtype=torch.uint8, device=send.device) for _ in range(world)]
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][: shards[r].n_frames * record_bytes] for r in range(world)])


def gather_compact(local_blob, group=None, dst: int = 0):
    """The single data-path collective, compact form: gather every rank's compact blob (a flat
    uint8 torch tensor holding exactly info.bytes bytes; on the GPU for nccl, on the CPU for gloo)
    to `dst`.  Blob sizes differ per rank, so the ranks first exchange their sizes (one 8-byte
    all_gather), pad to the largest and gather; `dst` returns the list of blobs in rank order
    (uint8 tensors, padding stripped), other ranks None."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    send = local_blob.reshape(-1)
    assert send.dtype == torch.uint8
    size = torch.tensor([send.numel()], dtype=torch.int64, device=send.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=send.device) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    max_bytes = max(sizes)
    if send.numel() < max_bytes:
        send = torch.cat([send, torch.zeros(max_bytes - send.numel(), dtype=torch.uint8, device=send.device)])
    recv: Optional[list] = None
    if rank == dst:
        recv = [torch.empty(max_bytes, dtype=torch.uint8, device=send.device) for _ in range(world)]
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [recv[r][: sizes[r]] for r in range(world)]


def assemble_frame_shards(blobs, sample_rate: int, n_samples: int, channels: int):
    """Root side of the frame-range sharding: the ranks' blobs, in rank (= frame) order, are ONE
    stream.  `blobs`: uint8 tensors / arrays / bytes as returned by gather_compact."""
    from .codec import EncodedAudio
    return EncodedAudio.from_compact(sample_rate, n_samples, channels, [_host_bytes(b) for b in blobs])


def assemble_streams(blobs, sample_rate: int, n_samples_per_stream, channels: int):
    """Root side of the stream-per-rank sharding (BASELINE config 4: stream s <-> rank
    s): every blob is a whole stream of its own.  -> list of EncodedAudio in rank order."""
    from .codec import EncodedAudio
    if isinstance(n_samples_per_stream, int):
        n_samples_per_stream = [n_samples_per_stream] * len(blobs)
    return [EncodedAudio.from_compact(sample_rate, n, channels, [_host_bytes(b)])
            for b, n in zip(blobs, n_samples_per_stream)]


def _host_bytes(b):
    if hasattr(b, "detach"):  # torch tensor (possibly on the GPU)
        return b.detach().cpu().numpy()
    return b
