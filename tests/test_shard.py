"""Multi-rank path on CPU: frame-range sharding + the single gather (gloo, world_size 2 / 3 / 5 / 8),
and the stream-per-rank variant.  Each rank's records are produced by the ORACLE from that rank's
own PCM slice (this is a test of the product's sharding, halo, gather and assembly logic, not of
the kernels — those are covered by the -m gpu tests)."""
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


def _worker(rank, world, port, x, sr, ch, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = glc_amd.plan_encode(x.size, ch)
        shards = shard.plan_shards(plan.n_frames, plan.per_channel, world)
        me = shards[rank]
        # Stand-in for the device encode of this shard: the oracle on THIS RANK'S PCM SLICE ONLY
        # (per-channel samples [t0, t0 + t_count) = its frame range + halo).  Everything else of the
        # stream is NaN-poisoned inside the oracle, so a wrong t0 / t_count / halo changes the bytes.
        mine = x[me.t0 * ch:(me.t0 + me.t_count) * ch].copy()
        recs, _ = O.encode_range_records(mine, me.t0, me.t_count, x.size, sr, ch, me.frame_begin, me.frame_end,
                                         n_threads=1)
        rec = glc_amd.lib.glc_record_bytes(ch)
        assert recs.size == me.n_frames * rec
        if mode == "records":
            got = shard.gather_records(torch.from_numpy(recs), shards, rec)
            if rank == 0:
                out = glc_amd.EncodedAudio.from_records(sr, x.size, ch, got.numpy())
                q.put(out.to_bytes())
            else:
                assert got is None
        else:
            blob = glc_amd.compact_records(recs, ch)
            got = shard.gather_compact(torch.from_numpy(blob))
            if rank == 0:
                assert len(got) == world
                out = shard.assemble_frame_shards(got, sr, x.size, ch)
                q.put(out.to_bytes())
                q.put(sum(int(g.numel()) for g in got))
            else:
                assert got is None
    finally:
        dist.destroy_process_group()


def _run(world, x, sr, ch, mode, n_results=1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, x, sr, ch, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(n_results)]
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,mode", [(2, "records"), (3, "compact"), (8, "compact")])
def test_sharded_encode_gathers_to_identical_bytes(world, mode):
    """Every rank encodes its frame range from its own PCM slice + halo; the gathered pieces
    assemble to the bytes of the whole-stream encode (tonal and raw frames, ragged split)."""
    x = np.concatenate([gen_tone("sine", 440.0, 44100, 2, 0.4), gen_noise(44100, 2, 0.1, 3)])
    want = O.encode(x, 44100, 2).glc
    res = _run(world, x, 44100, 2, mode, 1 if mode == "records" else 2)
    assert res[0] == want
    if mode == "compact":  # the compact gather moves the payload, not 4 KiB per frame-channel
        nf = glc_amd.plan_encode(x.size, 2).n_frames
        assert res[1] < nf * glc_amd.lib.glc_record_bytes(2) // 2


def test_more_ranks_than_frames():
    """Empty shards (world > n_frames): ranks without frames contribute an empty blob."""
    x = gen_tone("sine", 300.0, 48000, 1, 0.05)          # 2400 samples -> 2 frames
    assert glc_amd.plan_encode(x.size, 1).n_frames == 2
    res = _run(5, x, 48000, 1, "compact", 2)
    assert res[0] == O.encode(x, 48000, 1).glc


def _stream_worker(rank, world, port, sr, ch, q):
    """Stream-per-rank sharding (BASELINE config 4): rank s encodes stream s whole."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = _stream(rank, sr, ch)
        nf = glc_amd.plan_encode(x.size, ch).n_frames
        recs, _ = O.encode_range_records(x, 0, -(-x.size // ch), x.size, sr, ch, 0, nf, n_threads=1)
        got = shard.gather_compact(torch.from_numpy(glc_amd.compact_records(recs, ch)))
        if rank == 0:
            sizes = [_stream(r, sr, ch).size for r in range(world)]
            for ea in shard.assemble_streams(got, sr, sizes, ch):
                q.put(ea.to_bytes())
    finally:
        dist.destroy_process_group()


def _stream(rank, sr, ch):
    return gen_tone("sine", 220.0 * (rank + 1), sr, ch, 0.2 + 0.05 * rank) if rank != 1 else gen_noise(sr, ch, 0.15, 7)


def test_stream_per_rank_gather():
    world, sr, ch = 3, 48000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, sr, ch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r] == O.encode(_stream(r, sr, ch), sr, ch).glc


def test_compact_blob_validation():
    """glc_frames_from_compact rejects blobs whose counts disagree with their size or the stream."""
    sr, ch = 44100, 2
    x = np.concatenate([gen_tone("sine", 440.0, sr, ch, 0.2), gen_noise(sr, ch, 0.05, 3)])
    nf = glc_amd.plan_encode(x.size, ch).n_frames
    recs, _ = O.encode_range_records(x, 0, x.size // ch, x.size, sr, ch, 0, nf)
    blob = glc_amd.compact_records(recs, ch)
    assert glc_amd.EncodedAudio.from_compact(sr, x.size, ch, [blob]).to_bytes() == O.encode(x, sr, ch).glc
    # equal to the assembly from the dense records
    assert glc_amd.EncodedAudio.from_records(sr, x.size, ch, recs).to_bytes() == O.encode(x, sr, ch).glc
    bad = []
    b = blob.copy(); b[0] ^= 1; bad.append([b])                                    # magic
    b = blob.copy(); b[4] = 3; bad.append([b])                                     # channels
    b = blob.copy(); b[16:24] = np.frombuffer(np.uint64(10 ** 9).tobytes(), np.uint8); bad.append([b])  # n_pairs
    bad.append([blob[:-1]])                                                         # truncated
    bad.append([blob[:40]])                                                         # shorter than the header
    bad.append([blob, blob])                                                        # too many frames
    bad.append([])                                                                  # none at all
    lay_cnt = 64 + (nf + 63) // 64 * 64 + (4 * nf * ch + 63) // 64 * 64
    b = blob.copy(); b[lay_cnt:lay_cnt + 4] = np.frombuffer(np.uint32(2000).tobytes(), np.uint8); bad.append([b])  # cnt > 1024
    for blobs in bad:
        with pytest.raises(glc_amd.GlcError):
            glc_amd.EncodedAudio.from_compact(sr, x.size, ch, blobs)
    # a non-ascending list is legal container content: kept as stored, flagged for the decoder
    lay_pairs = lay_cnt + (4 * nf * ch + 63) // 64 * 64
    b = blob.copy()
    p = b[lay_pairs:lay_pairs + 8].copy()
    b[lay_pairs:lay_pairs + 4], b[lay_pairs + 4:lay_pairs + 8] = p[4:8], p[0:4]
    assert glc_amd.EncodedAudio.from_compact(sr, x.size, ch, [b]).to_bytes() != O.encode(x, sr, ch).glc


def test_hop_ranges_cover_the_untrimmed_stream():
    """Decode shards: n_frames + 1 hops, contiguous, balanced, in rank order."""
    from glc_amd import shard
    for nf in (0, 1, 5, 86, 4096, 28125):
        for world in (1, 2, 3, 8):
            rs = shard.hop_ranges(nf, world)
            assert len(rs) == world and rs[0].start == 0 and rs[-1].stop == nf + 1
            assert all(a.stop == b.start for a, b in zip(rs, rs[1:]))
            assert max(map(len, rs)) - min(map(len, rs)) <= 1
