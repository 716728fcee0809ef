"""ctypes front-end of the CPU oracle (oracle/libglc_oracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The oracle restates /root/reference/src/codec.rs; parity is UNPINNED by the reference (it ships
no golden vectors) — see oracle/glc_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libglc_oracle.so")

HOP = 1024
FRAME = 2048


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "glc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libglc_oracle.so"])
    return _SO


class _Taps(C.Structure):
    _fields_ = [
        ("coeffs", C.c_void_p),
        ("scales", C.c_void_p),
        ("nnz", C.c_void_p),
        ("is_raw", C.c_void_p),
        ("dense_q", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.glo_tables.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.glo_mdct_block.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.glo_imdct_block.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.glo_perceptual.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.glo_perceptual.restype = C.c_uint32
        L.glo_thresholds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.glo_compress.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.glo_compress.restype = C.c_uint32
        L.glo_num_frames.argtypes = [C.c_uint64, C.c_uint16]
        L.glo_num_frames.restype = C.c_uint64
        L.glo_encode.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint16, C.c_int,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(_Taps)]
        L.glo_encode.restype = C.c_int
        L.glo_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint16)]
        L.glo_decode.restype = C.c_int
        L.glo_time_encode_frames.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint16,
                                             C.c_uint64, C.c_uint64, C.c_int]
        L.glo_time_encode_frames.restype = C.c_double
        L.glo_encode_range_records.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                               C.c_uint16, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p,
                                               C.POINTER(_Taps)]
        L.glo_encode_range_records.restype = C.c_int
        L.glo_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


_tables_cache = None


def tables():
    """(T[1024,2048] f32, window[2048] f32, norm f32) — src/codec.rs:326-356."""
    global _tables_cache
    if _tables_cache is None:
        T = np.empty((HOP, FRAME), np.float32)
        w = np.empty(FRAME, np.float32)
        n = C.c_float()
        lib().glo_tables(_p(T), _p(w), C.byref(n))
        _tables_cache = (T, w, np.float32(n.value))
    return _tables_cache


def mdct_block(block: np.ndarray) -> np.ndarray:
    T, _, n = tables()
    block = np.ascontiguousarray(block, np.float32)
    out = np.empty(HOP, np.float32)
    lib().glo_mdct_block(_p(T), C.c_float(n), _p(block), _p(out))
    return out


def imdct_block(coeffs: np.ndarray) -> np.ndarray:
    T, _, n = tables()
    coeffs = np.ascontiguousarray(coeffs, np.float32)
    out = np.empty(FRAME, np.float32)
    lib().glo_imdct_block(_p(T), C.c_float(n), _p(coeffs), _p(out))
    return out


def perceptual(sample_rate: int):
    w = np.empty(HOP, np.float32)
    e = np.zeros(51, np.uint32)
    nb = lib().glo_perceptual(sample_rate, _p(w), _p(e))
    return w, e[:nb].copy()


def thresholds(coeffs, weights, edges) -> np.ndarray:
    coeffs = np.ascontiguousarray(coeffs, np.float32)
    edges = np.ascontiguousarray(edges, np.uint32)
    thr = np.empty(HOP, np.float32)
    lib().glo_thresholds(_p(coeffs), _p(weights), _p(edges), len(edges), _p(thr))
    return thr


def compress(coeffs, scale, thr):
    coeffs = np.ascontiguousarray(coeffs, np.float32)
    idx = np.empty(HOP, np.uint16)
    q = np.empty(HOP, np.int16)
    n = lib().glo_compress(_p(coeffs), C.c_float(scale), _p(thr), _p(idx), _p(q))
    return idx[:n].copy(), q[:n].copy()


def num_frames(n_samples: int, channels: int) -> int:
    return int(lib().glo_num_frames(n_samples, channels))


@dataclass
class OracleEncode:
    glc: bytes
    n_frames: int
    coeffs: np.ndarray | None = None   # [F*ch, 1024] f32
    scales: np.ndarray | None = None   # [F*ch] f32
    nnz: np.ndarray | None = None      # [F*ch] u32
    is_raw: np.ndarray | None = None   # [F] u8
    dense_q: np.ndarray | None = None  # [F*ch, 1024] i16


def encode(pcm: np.ndarray, sample_rate: int, channels: int, taps: bool = False,
           n_threads: int = 0) -> OracleEncode:
    pcm = np.ascontiguousarray(pcm, np.float32)
    nf = num_frames(pcm.size, channels)
    if nf == 0:
        raise ValueError("reference would panic on this input (SURVEY Q6)")
    t = _Taps()
    res = OracleEncode(b"", nf)
    if taps:
        M = nf * channels
        res.coeffs = np.empty((M, HOP), np.float32)
        res.scales = np.empty(M, np.float32)
        res.nnz = np.empty(M, np.uint32)
        res.is_raw = np.empty(nf, np.uint8)
        res.dense_q = np.empty((M, HOP), np.int16)
        t.coeffs, t.scales, t.nnz = _p(res.coeffs), _p(res.scales), _p(res.nnz)
        t.is_raw, t.dense_q = _p(res.is_raw), _p(res.dense_q)
    out = C.c_void_p()
    n = C.c_uint64()
    rc = lib().glo_encode(sample_rate, _p(pcm), pcm.size, channels, n_threads, C.byref(out),
                          C.byref(n), C.byref(t) if taps else None)
    if rc != 0:
        raise ValueError("oracle encode failed")
    res.glc = C.string_at(out, n.value)
    lib().glo_free(out)
    return res


def record_bytes(channels: int) -> int:
    return ((8 + 8 * channels) + 15) // 16 * 16 + 2 * FRAME * channels


def encode_range_records(shard: np.ndarray, t0: int, t_count: int, n_samples: int, sample_rate: int,
                         channels: int, f0: int, f1: int, taps: bool = False, n_threads: int = 0):
    """Frames [f0, f1) of a stream from ONE shard of its PCM (per-channel samples [t0, t0+t_count)),
    as the device path's fixed-size records; stream samples outside the shard are NaN-poisoned.
    -> (records uint8, OracleEncode taps with rows relative to f0, or None)."""
    shard = np.ascontiguousarray(shard, np.float32).reshape(-1)
    assert shard.size >= min(t_count * channels, max(0, n_samples - t0 * channels)), "shard shorter than it claims"
    nf = f1 - f0
    rec = np.zeros(nf * record_bytes(channels), np.uint8)
    t = _Taps()
    res = None
    if taps:
        M = nf * channels
        res = OracleEncode(b"", nf, np.empty((M, HOP), np.float32), np.empty(M, np.float32),
                           np.empty(M, np.uint32), np.empty(nf, np.uint8), np.empty((M, HOP), np.int16))
        t.coeffs, t.scales, t.nnz = _p(res.coeffs), _p(res.scales), _p(res.nnz)
        t.is_raw, t.dense_q = _p(res.is_raw), _p(res.dense_q)
    # the C side reads shard[(t - t0)*ch + c] only for samples inside the stream
    need = max(0, min(t_count * channels, n_samples - t0 * channels))
    buf = shard if shard.size >= t_count * channels else np.concatenate(
        [shard, np.zeros(t_count * channels - shard.size, np.float32)])
    del need
    rc = lib().glo_encode_range_records(sample_rate, _p(buf), t0, t_count, n_samples, channels, f0, f1,
                                        n_threads, _p(rec), C.byref(t) if taps else None)
    if rc != 0:
        raise ValueError("oracle range encode failed (bad range or a stream the reference panics on)")
    return rec, res


def decode(glc: bytes, n_threads: int = 0):
    """-> (pcm f32 interleaved, sample_rate, channels) — src/codec.rs:744-768."""
    buf = np.frombuffer(glc, np.uint8)
    out = C.c_void_p()
    n = C.c_uint64()
    sr = C.c_uint32()
    ch = C.c_uint16()
    rc = lib().glo_decode(_p(buf), buf.size, n_threads, C.byref(out), C.byref(n), C.byref(sr),
                          C.byref(ch))
    if rc != 0:
        raise ValueError("malformed .glc")
    pcm = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_float)), shape=(n.value,)).copy() \
        if n.value else np.empty(0, np.float32)
    lib().glo_free(out)
    return pcm, sr.value, ch.value


def time_encode_frames(pcm: np.ndarray, sample_rate: int, channels: int, f0: int, n_frames: int,
                       n_threads: int = 0) -> float:
    pcm = np.ascontiguousarray(pcm, np.float32)
    return float(lib().glo_time_encode_frames(sample_rate, _p(pcm), pcm.size, channels, f0,
                                              n_frames, n_threads))
