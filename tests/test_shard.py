"""Multi-rank path on CPU: frame-range sharding + the single gather (gloo, world_size 2 and 3).
Each rank's records are produced by the ORACLE here (this is a test of the product's sharding,
gather and assembly logic, not of the kernels — those are covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import glc_amd
from conftest import gen_tone, gen_noise, parse_glc, records_from_taps
from glc_amd import shard
from oracle import oracle as O


def test_frame_ranges_partition():
    for nf in (0, 1, 7, 86, 4096, 337500):
        for w in (1, 2, 3, 8):
            rs = shard.frame_ranges(nf, w)
            assert len(rs) == w and rs[0].start == 0 and rs[-1].stop == nf
            assert all(a.stop == b.start for a, b in zip(rs, rs[1:]))
            assert max(len(r) for r in rs) - min(len(r) for r in rs) <= 1


def test_shard_halo_covers_every_sample_read():
    L = 88200
    nf = O.num_frames(L * 2, 2)
    for w in (1, 2, 4, 8):
        for s in shard.plan_shards(nf, L, w):
            if s.n_frames == 0:
                continue
            lo = max(0, s.frame_begin * 1024 - 512)
            hi = min(L, (s.frame_end - 1) * 1024 - 512 + 2048)
            assert s.t0 <= lo and s.t0 + s.t_count >= hi
            assert s.t_count <= s.n_frames * 1024 + 1024  # range + one hop of halo


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, x, sr, ch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = glc_amd.plan_encode(x.size, ch)
        shards = shard.plan_shards(plan.n_frames, plan.per_channel, world)
        me = shards[rank]
        # stand-in for the device encode of this shard: oracle on the FULL stream, keep own frames
        enc = O.encode(x, sr, ch, taps=True, n_threads=1)
        ref = parse_glc(enc.glc)
        raw_rows = {}
        for f, fr in enumerate(ref["frames"]):
            if fr["raw"] is not None:
                for c in range(ch):
                    raw_rows[f * ch + c] = fr["raw"][c * 2048:(c + 1) * 2048]
        rec = glc_amd.lib.glc_record_bytes(ch)
        allrec = records_from_taps(enc, ch, raw_rows).reshape(-1, rec)
        local = torch.from_numpy(allrec[me.frame_begin:me.frame_end].copy().reshape(-1))
        got = shard.gather_records(local, shards, rec)
        if rank == 0:
            out = glc_amd.EncodedAudio.from_records(sr, x.size, ch, got.numpy())
            q.put(out.to_bytes() == enc.glc)
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_encode_gathers_to_identical_bytes(world):
    x = np.concatenate([gen_tone("sine", 440.0, 44100, 2, 0.4), gen_noise(44100, 2, 0.1, 3)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, x, 44100, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_hop_ranges_cover_the_untrimmed_stream():
    """Decode shards: n_frames + 1 hops, contiguous, balanced, in rank order."""
    from glc_amd import shard
    for nf in (0, 1, 5, 86, 4096, 28125):
        for world in (1, 2, 3, 8):
            rs = shard.hop_ranges(nf, world)
            assert len(rs) == world and rs[0].start == 0 and rs[-1].stop == nf + 1
            assert all(a.stop == b.start for a, b in zip(rs, rs[1:]))
            assert max(map(len, rs)) - min(map(len, rs)) <= 1
