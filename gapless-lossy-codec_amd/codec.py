"""Host-side mirror of the reference's codec API (src/codec.rs) over the C ABI (include/glc.h).

Names, argument meaning and results follow the reference crate:

    Encoder::new(sample_rate)                      src/codec.rs:406   -> Encoder(sample_rate)
    Encoder::encode(&samples, channels)            src/codec.rs:421   -> Encoder.encode(samples, channels)
    Decoder::new(channels, sample_rate)            src/codec.rs:581   -> Decoder(channels, sample_rate)
    Decoder::decode(&encoded, progress)            src/codec.rs:744   -> Decoder.decode(encoded)
    Decoder::decode_streaming(encoded, progress)   src/codec.rs:595   -> Decoder.decode_streaming(encoded)
    save_encoded / load_encoded                    src/codec.rs:774-786

The reference panics on degenerate input; here those cases raise GlcError with
code GLC_EINVAL.  All arithmetic happens in libglc_hip.so on a gfx950 device; this module holds no
numerics and no fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import (FRAMES_HOOK, GLC_EINVAL, GlcCompactInfo, GlcError, GlcFramesGather, GlcFramesView, GlcInfo, GlcPlan,
                   check, lib)

FRAME_SIZE = 2048        # src/codec.rs:15
HOP_SIZE = 1024          # src/codec.rs:16
FRAMES_PER_CHUNK = 500   # src/codec.rs:18


@dataclass
class AudioHeader:  # src/codec.rs:39-45
    sample_rate: int
    channels: int
    total_samples: int


@dataclass
class GaplessInfo:  # src/codec.rs:47-53
    encoder_delay: int
    padding: int
    original_length: int


@dataclass
class EncodedFrame:  # src/codec.rs:56-69
    sparse_coeffs_per_channel: List[List[Tuple[int, int]]]
    scale_factors: List[float]
    raw_pcm: Optional[np.ndarray]


@dataclass
class AudioChunk:  # src/codec.rs:81-85
    samples: np.ndarray
    is_last: bool


class _Frames(Sequence):
    """Lazy `Vec<EncodedFrame>` view over a glc_frames handle."""

    def __init__(self, owner: "EncodedAudio"):
        self._o = owner

    def __len__(self) -> int:
        return self._o.info().n_frames

    def __getitem__(self, f):
        if isinstance(f, slice):
            return [self[i] for i in range(*f.indices(len(self)))]
        n = len(self)
        if f < 0:
            f += n
        if not 0 <= f < n:
            raise IndexError(f)
        h = self._o._h
        if lib.glc_frame_is_raw(h, f) == 1:
            ln = C.c_uint64()
            check(lib.glc_frame_raw(h, f, None, 0, C.byref(ln)))
            raw = np.empty(ln.value, np.int16)
            check(lib.glc_frame_raw(h, f, raw.ctypes.data_as(C.c_void_p), ln.value, C.byref(ln)))
            return EncodedFrame([], [], raw)
        lists, scales = [], []
        c = 0
        while True:
            n_pairs = C.c_uint32()
            if lib.glc_frame_sparse(h, f, c, None, None, 0, C.byref(n_pairs)) != 0:
                break
            idx = np.empty(n_pairs.value, np.uint16)
            q = np.empty(n_pairs.value, np.int16)
            check(lib.glc_frame_sparse(h, f, c, idx.ctypes.data_as(C.c_void_p),
                                       q.ctypes.data_as(C.c_void_p), n_pairs.value, C.byref(n_pairs)))
            lists.append(list(zip(idx.tolist(), q.tolist())))
            c += 1
        c = 0
        while True:
            s = C.c_float()
            if lib.glc_frame_scale(h, f, c, C.byref(s)) != 0:
                break
            scales.append(s.value)
            c += 1
        return EncodedFrame(lists, scales, None)


class EncodedAudio:
    """src/codec.rs:31-37; owns a library-side glc_frames object."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.glc_frames_free(h)

    def info(self) -> GlcInfo:
        i = GlcInfo()
        check(lib.glc_frames_info(self._h, C.byref(i)))
        return i

    @property
    def header(self) -> AudioHeader:
        i = self.info()
        return AudioHeader(i.sample_rate, i.channels, i.total_samples)

    @property
    def gapless_info(self) -> GaplessInfo:
        i = self.info()
        return GaplessInfo(i.encoder_delay, i.padding, i.original_length)

    @property
    def frames(self) -> _Frames:
        return _Frames(self)

    def to_bytes(self) -> bytes:
        """bincode::serialize(encoded) — the .glc byte stream (src/codec.rs:776)."""
        n = lib.glc_serialized_size(self._h)
        buf = (C.c_uint8 * n)()
        w = C.c_uint64()
        check(lib.glc_serialize(self._h, buf, n, C.byref(w)))
        return bytes(buf)

    @staticmethod
    def from_bytes(data: bytes) -> "EncodedAudio":
        """bincode::deserialize (src/codec.rs:784)."""
        arr = np.frombuffer(data, np.uint8)
        out = C.c_void_p()
        check(lib.glc_deserialize(arr.ctypes.data_as(C.c_void_p), arr.size, C.byref(out)))
        return EncodedAudio(out.value)

    # ---- the structured bridge (include/glc.h glc_frames_view): EncodedAudio as flat arrays ----------
    _PART_TYPES = (("list_begin", np.uint64), ("list_off", np.uint64), ("pairs", np.uint32), ("scale_begin", np.uint64),
                   ("scales", np.float32), ("raw_tag", np.uint8), ("raw_begin", np.uint64), ("raw", np.int16))

    @property
    def stream_id(self) -> int:
        return int(lib.glc_frames_stream_id(self._h))

    def parts(self) -> dict:
        """Copies of the flat pools behind this EncodedAudio (glc_frames_get_view): header fields plus
        list_begin / list_off / pairs / scale_begin / scales / raw_tag / raw_begin / raw as numpy arrays."""
        v = GlcFramesView()
        check(lib.glc_frames_get_view(self._h, C.byref(v)))
        return _view_to_dict(v)

    @staticmethod
    def from_parts(parts: dict, stream_id: int = 0) -> "EncodedAudio":
        """glc_frames_from_parts: the inverse of parts(); offsets are validated by the library."""
        keep = {k: np.ascontiguousarray(parts[k], t).reshape(-1) for k, t in EncodedAudio._PART_TYPES}
        v = GlcFramesView()
        for k in ("sample_rate", "channels", "total_samples", "encoder_delay", "padding", "original_length"):
            setattr(v, k, int(parts[k]))
        v.n_frames = int(parts.get("n_frames", keep["raw_tag"].size))
        v.n_lists = int(parts.get("n_lists", max(keep["list_off"].size, 1) - 1))
        v.n_pairs = int(parts.get("n_pairs", keep["pairs"].size))
        v.n_scales = int(parts.get("n_scales", keep["scales"].size))
        v.n_raw = int(parts.get("n_raw", keep["raw"].size))
        for k, _ in EncodedAudio._PART_TYPES:
            setattr(v, k, keep[k].ctypes.data if keep[k].size else None)
        out = C.c_void_p()
        check(lib.glc_frames_from_parts(C.byref(v), stream_id, C.byref(out)))
        return EncodedAudio(out.value)

    @staticmethod
    def from_nested(header: AudioHeader, frames, gapless: GaplessInfo, stream_id: int = 0) -> "EncodedAudio":
        """glc_frames_from_gather from the reference's nested shape: `frames` is a sequence of
        (sparse_coeffs_per_channel: list of uint32 arrays of packed (idx | q << 16) pairs,
         scale_factors: float32 array, raw_pcm: int16 array or None) - one pointer per vector crosses."""
        lists_per, list_ptr, list_len, scales_per, scale_ptr, raw_ptr, raw_len, keep = [], [], [], [], [], [], [], []
        for lists, scales, raw in frames:
            lists_per.append(len(lists))
            for l in lists:
                a = np.ascontiguousarray(l, np.uint32).reshape(-1)
                keep.append(a)
                list_ptr.append(a.ctypes.data if a.size else 0)
                list_len.append(a.size)
            sc = np.ascontiguousarray(scales, np.float32).reshape(-1)
            keep.append(sc)
            scales_per.append(sc.size)
            scale_ptr.append(sc.ctypes.data if sc.size else 0)
            if raw is None:
                raw_ptr.append(0)
                raw_len.append(0)
            else:
                r = np.ascontiguousarray(raw, np.int16).reshape(-1)
                n_r = r.size
                if n_r == 0:
                    r = np.zeros(1, np.int16)  # Some(vec![]): a non-null pointer with length 0
                keep.append(r)
                raw_ptr.append(r.ctypes.data)
                raw_len.append(n_r)
        arr = lambda x, t: np.ascontiguousarray(np.array(x, dtype=t))
        a_lp, a_ll, a_sp = arr(lists_per, np.uint32), arr(list_len, np.uint32), arr(scales_per, np.uint32)
        a_ptr, a_sptr, a_rptr, a_rl = arr(list_ptr, np.uint64), arr(scale_ptr, np.uint64), arr(raw_ptr, np.uint64), arr(raw_len, np.uint64)
        g = GlcFramesGather()
        g.sample_rate, g.channels, g.total_samples = header.sample_rate, header.channels, header.total_samples
        g.encoder_delay, g.padding, g.original_length = gapless.encoder_delay, gapless.padding, gapless.original_length
        g.n_frames = len(lists_per)
        for k, a in (("lists_per_frame", a_lp), ("list_ptr", a_ptr), ("list_len", a_ll), ("scales_per_frame", a_sp),
                     ("scale_ptr", a_sptr), ("raw_ptr", a_rptr), ("raw_len", a_rl)):
            setattr(g, k, a.ctypes.data if a.size else None)
        out = C.c_void_p()
        check(lib.glc_frames_from_gather(C.byref(g), stream_id, C.byref(out)))
        return EncodedAudio(out.value)

    @staticmethod
    def from_records(sample_rate: int, n_samples: int, channels: int, records: np.ndarray) -> "EncodedAudio":
        """Assemble from the device path's fixed-size frame records (all shards, frame order)."""
        records = np.ascontiguousarray(records, np.uint8).reshape(-1)
        rec = lib.glc_record_bytes(channels)
        if rec == 0 or records.size % rec:
            raise GlcError(GLC_EINVAL, "record buffer is not a whole number of records")
        out = C.c_void_p()
        check(lib.glc_frames_from_records(sample_rate, n_samples, channels,
                                          records.ctypes.data_as(C.c_void_p), records.size // rec,
                                          C.byref(out)))
        return EncodedAudio(out.value)


    @staticmethod
    def from_compact(sample_rate: int, n_samples: int, channels: int, blobs) -> "EncodedAudio":
        """Assemble from compact blobs (one per shard, frame order): glc_frames_from_compact.
        Each blob is a bytes-like / uint8 array as produced by Encoder.compact_device_records or
        compact_records."""
        arrs = [np.ascontiguousarray(np.frombuffer(b, np.uint8) if isinstance(b, (bytes, bytearray, memoryview))
                                     else b, np.uint8).reshape(-1) for b in blobs]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        sizes = (C.c_uint64 * max(n, 1))(*[a.size for a in arrs])
        out = C.c_void_p()
        check(lib.glc_frames_from_compact(sample_rate, n_samples, channels, ptrs, sizes, n, C.byref(out)))
        return EncodedAudio(out.value)


def _view_to_dict(v: GlcFramesView) -> dict:
    def take(ptr, n, t):
        if not n:
            return np.empty(0, t)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(t))), shape=(n,)).copy()
    d = {k: int(getattr(v, k)) for k in ("sample_rate", "channels", "total_samples", "encoder_delay", "padding",
                                         "original_length", "n_frames", "n_lists", "n_pairs", "n_scales", "n_raw")}
    nf = d["n_frames"]
    d["list_begin"] = take(v.list_begin, nf + 1, np.uint64)
    d["list_off"] = take(v.list_off, d["n_lists"] + 1, np.uint64)
    d["pairs"] = take(v.pairs, d["n_pairs"], np.uint32)
    d["scale_begin"] = take(v.scale_begin, nf + 1, np.uint64)
    d["scales"] = take(v.scales, d["n_scales"], np.float32)
    d["raw_tag"] = take(v.raw_tag, nf, np.uint8)
    d["raw_begin"] = take(v.raw_begin, nf + 1, np.uint64)
    d["raw"] = take(v.raw, d["n_raw"], np.int16)
    return d


def compact_bound(channels: int, n_frames: int) -> int:
    """Capacity a compact-blob buffer needs for n_frames frames (glc_compact_bound)."""
    return int(lib.glc_compact_bound(channels, n_frames))


def compact_records(records: np.ndarray, channels: int) -> np.ndarray:
    """Host twin of the device compaction (glc_compact_records): records -> compact blob bytes."""
    records = np.ascontiguousarray(records, np.uint8).reshape(-1)
    rec = lib.glc_record_bytes(channels)
    if rec == 0 or records.size % rec:
        raise GlcError(GLC_EINVAL, "record buffer is not a whole number of records")
    nf = records.size // rec
    blob = np.empty(compact_bound(channels, nf), np.uint8)
    info = GlcCompactInfo()
    check(lib.glc_compact_records(records.ctypes.data_as(C.c_void_p), nf, channels, blob.ctypes.data_as(C.c_void_p),
                                  blob.size, C.byref(info)))
    return blob[:info.bytes].copy()


def plan_encode(n_samples: int, channels: int) -> GlcPlan:
    """Frame count / padding of Encoder::encode (src/codec.rs:433-455); raises where it panics."""
    p = GlcPlan()
    check(lib.glc_plan_encode(n_samples, channels, C.byref(p)))
    return p


class _Ctx:
    def __init__(self, sample_rate: int, device: int):
        h = C.c_void_p()
        check(lib.glc_ctx_create(device, sample_rate, C.byref(h)))
        self._h = h
        self.sample_rate = sample_rate
        self.device = device

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.glc_ctx_destroy(h)

    def close(self):
        self.__del__()

    def tables(self):
        """(cos_table[1024,2048], window[2048], norm, weights[1024], band_edges)."""
        T = np.empty((HOP_SIZE, FRAME_SIZE), np.float32)
        w = np.empty(FRAME_SIZE, np.float32)
        wt = np.empty(HOP_SIZE, np.float32)
        e = np.zeros(64, np.uint32)
        n = C.c_float()
        ne = C.c_uint32()
        check(lib.glc_ctx_tables(self._h, T.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                 C.byref(n), wt.ctypes.data_as(C.c_void_p),
                                 e.ctypes.data_as(C.c_void_p), C.byref(ne)), self._h)
        return T, w, np.float32(n.value), wt, e[:ne.value].copy()

    def set_stream(self, raw_stream: int) -> None:
        """Run this context's kernels on a caller-owned hipStream_t (0 restores the private stream).
        A handle is only meaningful inside the HIP runtime that created it: refused when the process
        has mapped two (libglc_hip.so loaded before torch, see _lib.py)."""
        if raw_stream:
            from ._lib import hip_runtimes_mapped
            rts = hip_runtimes_mapped()
            if len(rts) > 1:
                raise GlcError(GLC_EINVAL, f"two HIP runtimes are mapped ({rts}): a foreign hipStream_t must not cross; "
                                           "import torch before glc_amd")
        check(lib.glc_ctx_set_stream(self._h, C.c_void_p(raw_stream)), self._h)

    def synchronize(self) -> None:
        check(lib.glc_ctx_synchronize(self._h), self._h)

    def timer_begin(self) -> None:
        """Record a HIP event on the context's stream (device-side stopwatch)."""
        check(lib.glc_ctx_timer_begin(self._h), self._h)

    def timer_end(self) -> float:
        """Record a second event, wait for it, return elapsed milliseconds."""
        ms = C.c_float()
        check(lib.glc_ctx_timer_end(self._h, C.byref(ms)), self._h)
        return ms.value


class Encoder(_Ctx):
    """Encoder::new(sample_rate) — src/codec.rs:406."""

    def __init__(self, sample_rate: int, device: int = 0):
        super().__init__(sample_rate, device)

    def encode(self, samples, channels: int) -> EncodedAudio:
        """Encoder::encode(&mut self, samples: &[f32], channels: u16) — src/codec.rs:421."""
        pcm = np.ascontiguousarray(samples, np.float32).reshape(-1)
        out = C.c_void_p()
        check(lib.glc_encode(self._h, pcm.ctypes.data_as(C.c_void_p), pcm.size, channels, C.byref(out)),
              self._h)
        return EncodedAudio(out.value)

    def encode_hooked(self, samples, channels: int, hook) -> EncodedAudio:
        """glc_encode_hooked: `hook(parts, frame_begin, frame_end)` is called each time a range of frames
        has arrived on the host (while the device works on later ones); `parts` is the dict of
        EncodedAudio.parts() for frames [0, frame_end).  A truthy return aborts the encode."""
        pcm = np.ascontiguousarray(samples, np.float32).reshape(-1)
        failure = []

        def tramp(_user, view, f0, f1):
            try:
                return 1 if hook(_view_to_dict(view.contents), int(f0), int(f1)) else 0
            except BaseException as e:  # no exception may cross the C frames
                failure.append(e)
                return 1
        cb = FRAMES_HOOK(tramp)
        out = C.c_void_p()
        rc = lib.glc_encode_hooked(self._h, pcm.ctypes.data_as(C.c_void_p), pcm.size, channels, cb, None, C.byref(out))
        if failure:
            raise failure[0]
        check(rc, self._h)
        return EncodedAudio(out.value)

    def encode_range_device(self, d_pcm: int, t0: int, t_count: int, n_samples: int, channels: int,
                            frame_begin: int, frame_end: int, d_records: int, d_coeffs: int = 0) -> None:
        """Device-resident frame range (body of the rayon loop, src/codec.rs:462-541).  Pointers
        are raw device addresses; work is queued on the context's stream, not synchronised."""
        check(lib.glc_encode_range_device(self._h, C.c_void_p(d_pcm), t0, t_count, n_samples, channels,
                                          frame_begin, frame_end, C.c_void_p(d_records),
                                          C.c_void_p(d_coeffs) if d_coeffs else None), self._h)

    def frames_from_device_records(self, d_records: int, n_frames: int, n_samples: int, channels: int) -> EncodedAudio:
        """EncodedAudio from records still on this context's device (device-side compaction)."""
        out = C.c_void_p()
        check(lib.glc_frames_from_device_records(self._h, C.c_void_p(d_records), n_frames, n_samples, channels,
                                                 C.byref(out)), self._h)
        return EncodedAudio(out.value)

    def compact_device_records(self, d_records: int, n_frames: int, channels: int, d_blob: int, cap: int) -> GlcCompactInfo:
        """Pack n_frames device records into the compact blob at device address d_blob
        (glc_compact_device_records); synchronises; returns the sizes (info.bytes travel)."""
        info = GlcCompactInfo()
        check(lib.glc_compact_device_records(self._h, C.c_void_p(d_records) if d_records else None, n_frames, channels,
                                             C.c_void_p(d_blob), cap, C.byref(info)), self._h)
        return info

    def mdct_forward_device(self, d_pcm: int, t0: int, t_count: int, n_samples: int, channels: int,
                            frame_begin: int, frame_end: int, d_coeffs: int) -> None:
        """Window + mdct_block only (src/codec.rs:476-485) for a frame range, device-resident."""
        check(lib.glc_mdct_forward_device(self._h, C.c_void_p(d_pcm), t0, t_count, n_samples, channels,
                                          frame_begin, frame_end, C.c_void_p(d_coeffs)), self._h)


class Decoder(_Ctx):
    """Decoder::new(channels, sample_rate) — src/codec.rs:581.  `channels` is accepted and
    ignored exactly like the reference (quirk Q4): the stream's header decides."""

    def __init__(self, channels: int, sample_rate: int, device: int = 0):
        super().__init__(sample_rate, device)
        self.channels = channels

    @staticmethod
    def _progress(sender, kind: str, value) -> None:
        """Progress (src/codec.rs:71-79) delivered to an optional callable(kind, value) in place of
        the crossbeam Sender: kinds 'Status', 'Decoding', 'Complete' as sent at :609, :713, :736."""
        if sender is not None:
            sender(kind, value)

    def decode(self, encoded: EncodedAudio, progress_sender=None, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Decoder::decode — src/codec.rs:744-768 (overlap-add, gapless trim).  `out` (optional, not
        in the reference): a float32 array of at least total_samples to decode into, for callers
        that reuse a buffer - pages that were touched before take the D2H copies much faster."""
        if progress_sender is not None:  # the reference decodes through decode_streaming (:747)
            chunks = [c.samples for c in self.decode_streaming(encoded, progress_sender)]
            allv = np.concatenate(chunks) if chunks else np.empty(0, np.float32)
            g = encoded.gapless_info
            if allv.size > g.encoder_delay:
                allv = allv[g.encoder_delay:]
            return allv[:g.original_length].copy()
        n = lib.glc_decoded_len(encoded._h)
        if out is None:
            out = np.empty(n, np.float32)
        elif out.dtype != np.float32 or not out.flags.c_contiguous or out.size < n:
            raise GlcError(GLC_EINVAL, "out must be a C-contiguous float32 array of at least total_samples")
        got = C.c_uint64()
        check(lib.glc_decode(self._h, encoded._h, out.ctypes.data_as(C.c_void_p), n, C.byref(got)),
              self._h)
        return out[:got.value]

    def resident_stream(self) -> int:
        """Identity of the stream whose sparse rows this context holds on the device (0: none)."""
        return int(lib.glc_ctx_resident_stream(self._h))

    def decode_resident(self, stream_id: int, out: np.ndarray) -> np.ndarray:
        """glc_decode_resident: Decoder::decode of the resident stream without an EncodedAudio."""
        got = C.c_uint64()
        check(lib.glc_decode_resident(self._h, stream_id, out.ctypes.data_as(C.c_void_p), out.size, C.byref(got)), self._h)
        return out[:got.value]

    def decode_device(self, encoded: EncodedAudio, d_all: int, cap_all: int):
        """Device-resident decode of the whole un-trimmed stream into device memory at address
        d_all; returns (start, n): the window Decoder::decode would return (src/codec.rs:756-765).
        Queued on the context's stream, not synchronised."""
        start = C.c_uint64()
        n = C.c_uint64()
        check(lib.glc_decode_device(self._h, encoded._h, C.c_void_p(d_all), cap_all, C.byref(start), C.byref(n)),
              self._h)
        return start.value, n.value

    def decode_range_device(self, encoded: EncodedAudio, hop_begin: int, hop_end: int, d_out: int, cap: int) -> None:
        """One shard of the decode: hops [hop_begin, hop_end) of the un-trimmed stream into device
        memory at address d_out (glc_decode_range_device).  Queued, not synchronised."""
        check(lib.glc_decode_range_device(self._h, encoded._h, hop_begin, hop_end, C.c_void_p(d_out), cap), self._h)

    def imdct_device(self, encoded: EncodedAudio, frame_begin: int, frame_end: int, d_blocks: int) -> None:
        """Dequant + imdct_block + window (src/codec.rs:651-675) alone for a frame range:
        d_blocks[(frame - frame_begin) * ch + c][2048] on the device (glc_imdct_device)."""
        check(lib.glc_imdct_device(self._h, encoded._h, frame_begin, frame_end, C.c_void_p(d_blocks)), self._h)

    def decode_streaming(self, encoded: EncodedAudio, progress_sender=None) -> Iterator[AudioChunk]:
        """Decoder::decode_streaming — src/codec.rs:595-741: yields AudioChunk until is_last."""
        import time as _time
        t0 = _time.perf_counter()
        total_frames = encoded.info().n_frames
        self._progress(progress_sender, "Status", f"Starting streaming decode of {total_frames} frames")
        check(lib.glc_decode_stream_begin(self._h, encoded._h), self._h)
        ch = encoded.header.channels
        cap = FRAMES_PER_CHUNK * HOP_SIZE * ch  # the last chunk is < 500 frames + the tail hop
        done = 0
        while True:
            buf = np.empty(cap, np.float32)
            n = C.c_uint64()
            last = C.c_int()
            check(lib.glc_decode_stream_next(self._h, buf.ctypes.data_as(C.c_void_p), cap, C.byref(n),
                                             C.byref(last)), self._h)
            done += FRAMES_PER_CHUNK
            if not last.value and total_frames:
                self._progress(progress_sender, "Decoding", min(done, total_frames) / total_frames * 100.0)
            yield AudioChunk(buf[:n.value].copy(), bool(last.value))
            if last.value:
                self._progress(progress_sender, "Complete",
                               f"Decoded {total_frames} frames in {_time.perf_counter() - t0:.2f}s")
                return


def save_encoded(encoded: EncodedAudio, path) -> None:
    """src/codec.rs:774-779"""
    check(lib.glc_save(encoded._h, str(path).encode()))


def load_encoded(path) -> EncodedAudio:
    """src/codec.rs:781-786"""
    out = C.c_void_p()
    check(lib.glc_load(str(path).encode(), C.byref(out)))
    return EncodedAudio(out.value)


def load_wav(path):
    """audio::load_wav (src/audio.rs:39-64) -> (samples f32 interleaved, sample_rate, channels)."""
    ptr = C.c_void_p()
    n = C.c_uint64()
    sr = C.c_uint32()
    ch = C.c_uint16()
    check(lib.glc_wav_load(str(path).encode(), C.byref(ptr), C.byref(n), C.byref(sr), C.byref(ch)))
    try:
        out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n.value,)).copy() \
            if n.value else np.empty(0, np.float32)
    finally:
        lib.glc_free(ptr)
    return out, sr.value, ch.value


def export_to_wav(path, samples, sample_rate: int, channels: int) -> None:
    """audio::export_to_wav (src/audio.rs:100-132): 16-bit PCM."""
    s = np.ascontiguousarray(samples, np.float32).reshape(-1)
    check(lib.glc_wav_save16(str(path).encode(), s.ctypes.data_as(C.c_void_p), s.size, sample_rate, channels))


def _take_f32(ptr, n):
    try:
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,)).copy() if n \
            else np.empty(0, np.float32)
    finally:
        lib.glc_free(ptr)


def encode_flac_with_level(samples, sample_rate: int, channels: int, compression_level: int) -> bytes:
    """flac::encode_flac_with_level (src/flac.rs:947-1053): the reference's own FLAC encoder."""
    s = np.ascontiguousarray(samples, np.float32).reshape(-1)
    if not 0 <= int(compression_level) <= 255 or not 0 <= int(channels) <= 0xFFFF:
        raise GlcError(-1, "compression_level is a u8 and channels a u16 in the reference")
    ptr = C.c_void_p()
    n = C.c_uint64()
    check(lib.glc_flac_encode(s.ctypes.data_as(C.c_void_p), s.size, sample_rate, channels, compression_level,
                              C.byref(ptr), C.byref(n)))
    try:
        return C.string_at(ptr, n.value)
    finally:
        lib.glc_free(ptr)


def encode_flac(samples, sample_rate: int, channels: int) -> bytes:
    """flac::encode_flac (src/flac.rs:1056-1063): compression level 5."""
    return encode_flac_with_level(samples, sample_rate, channels, 5)


def export_to_flac_with_level(path, samples, sample_rate: int, channels: int, compression_level: int) -> None:
    """flac::export_to_flac_with_level (src/flac.rs:1066-1077)."""
    s = np.ascontiguousarray(samples, np.float32).reshape(-1)
    check(lib.glc_flac_save(str(path).encode(), s.ctypes.data_as(C.c_void_p), s.size, sample_rate, channels,
                            compression_level))


def export_to_flac(path, samples, sample_rate: int, channels: int) -> None:
    """audio::export_to_flac (src/audio.rs:87-96) = flac::export_to_flac (src/flac.rs:1080-1087)."""
    export_to_flac_with_level(path, samples, sample_rate, channels, 5)


def load_flac(path):
    """audio::load_flac (src/audio.rs:68-85) -> (samples f32 interleaved, sample_rate, channels)."""
    ptr = C.c_void_p()
    n = C.c_uint64()
    sr = C.c_uint32()
    ch = C.c_uint16()
    check(lib.glc_flac_load(str(path).encode(), C.byref(ptr), C.byref(n), C.byref(sr), C.byref(ch)))
    return _take_f32(ptr, n.value), sr.value, ch.value


def decode_flac(data: bytes):
    """In-memory form of load_flac."""
    ptr = C.c_void_p()
    n = C.c_uint64()
    sr = C.c_uint32()
    ch = C.c_uint16()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else (C.c_uint8 * 1)()
    check(lib.glc_flac_decode(C.cast(buf, C.c_void_p), len(data), C.byref(ptr), C.byref(n), C.byref(sr), C.byref(ch)))
    return _take_f32(ptr, n.value), sr.value, ch.value


def load_audio_file_lossless(path):
    """audio::load_audio_file_lossless (src/audio.rs:19-36): dispatch on the lower-cased extension."""
    import os
    ext = os.path.splitext(str(path))[1]
    if not ext:
        raise GlcError(-1, "No file extension")
    ext = ext[1:].lower()
    if ext == "wav":
        return load_wav(path)
    if ext == "flac":
        return load_flac(path)
    raise GlcError(-1, f"Unsupported file format: {ext}")
