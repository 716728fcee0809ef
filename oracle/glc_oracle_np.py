"""Second, independent restatement of /root/reference/src/codec.rs in numpy float32.

TEST INFRASTRUCTURE ONLY (same rules as oracle/oracle.py).  Its job is to pin the C oracle:
tests/test_oracle.py requires the two to agree bit-for-bit on MDCT coefficients, scale factors,
quantised integers, raw decisions, .glc bytes and decoded PCM.  It shares no code with the C
oracle: numpy float32 elementwise ops are IEEE binary32 with no fusion, loops over the
accumulation index keep the reference's summation order, and the bincode layout is written with
`struct`.  Transcendentals: the f32 angles are formed in numpy and passed one by one to the
system libm's cosf/sinf through ctypes — Rust's f32::cos/sin call the same libm on Linux, and
glibc's cosf is NOT correctly rounded (1.3 % of the table differs by 1 ulp from the rounded
f64 cosine), so the libm is part of the reference's definition.
"""
from __future__ import annotations

import ctypes as _C
import struct

import numpy as np

_libm = _C.CDLL("libm.so.6")
_libm.cosf.restype = _C.c_float
_libm.cosf.argtypes = [_C.c_float]
_libm.sinf.restype = _C.c_float
_libm.sinf.argtypes = [_C.c_float]


def _map_f32(fn, a: np.ndarray) -> np.ndarray:
    return np.array([fn(x) for x in a.reshape(-1).tolist()], np.float32).reshape(a.shape)


F32 = np.float32
HOP, FRAME = 1024, 2048
PI = F32(np.pi)


def tables():
    """src/codec.rs:326-356"""
    n = F32(HOP)
    i = np.arange(FRAME, dtype=F32)
    k = np.arange(HOP, dtype=F32)
    a = PI / n
    b = (i + F32(0.5)) + n / F32(2.0)
    ab = (a * b).astype(F32)
    angle = (ab[None, :] * (k[:, None] + F32(0.5))).astype(F32)
    T = _map_f32(_libm.cosf, angle)
    num = (PI * (i + F32(0.5))).astype(F32)
    w = _map_f32(_libm.sinf, (num / F32(FRAME)).astype(F32))
    norm = np.sqrt(F32(2.0) / n).astype(F32)
    return T, w, norm


_cache = None


def tables_cached():
    global _cache
    if _cache is None:
        _cache = tables()
    return _cache


def perceptual(sr: int):
    """src/codec.rs:102-183"""
    n = F32(HOP)
    srf = F32(sr)
    k = np.arange(HOP, dtype=F32)
    f = ((k / (F32(2.0) * n)) * srf).astype(F32)
    w = np.empty(HOP, F32)
    for j in range(HOP):
        x = f[j]
        if x < 100.0:
            v = F32(0.3) + (x / F32(100.0)) * F32(0.4)
        elif x < 200.0:
            v = F32(0.7) + ((x - F32(100.0)) / F32(100.0)) * F32(0.3)
        elif x < 5000.0:
            v = F32(1.0)
        elif x < 10000.0:
            v = F32(1.0) - ((x - F32(5000.0)) / F32(5000.0)) * F32(0.3)
        else:
            v = F32(0.7) - min((x - F32(10000.0)) / F32(12000.0), F32(1.0)) * F32(0.5)
        w[j] = max(F32(v), F32(0.2))
    edges = [0]
    nyq = srf / F32(2.0)
    freq = F32(0.0)
    while freq < nyq and len(edges) < 50:
        b = int((freq / nyq) * n)
        if b > edges[-1] and b < HOP:
            edges.append(b)
        if freq < 500.0:
            freq = F32(freq + F32(50.0))
        elif freq < 2000.0:
            freq = F32(freq + F32(100.0))
        elif freq < 8000.0:
            freq = F32(freq + F32(250.0))
        else:
            freq = F32(freq + F32(500.0))
    edges.append(HOP)
    return w, np.array(edges, np.uint32)


def num_frames(n_samples: int, ch: int) -> int:
    l0 = -(-n_samples // ch)
    r = -(-(512 + l0) // 1024)
    return r - 1


def windowed_rows(pcm: np.ndarray, ch: int, w: np.ndarray):
    """rows m = frame*ch + c of slice*window (src/codec.rs:426-481)"""
    n = pcm.size
    assert n % ch == 0
    L = n // ch
    P = -(-(512 + L) // 1024) * 1024 + 512
    nf = (P - FRAME) // HOP + 1
    padded = np.zeros((ch, P), F32)
    padded[:, 512:512 + L] = pcm.reshape(L, ch).T
    rows = np.empty((nf * ch, FRAME), F32)
    for f in range(nf):
        rows[f * ch:(f + 1) * ch] = padded[:, f * HOP:f * HOP + FRAME] * w[None, :]
    return rows, nf, P, L


def mdct_rows(rows: np.ndarray, T: np.ndarray, norm) -> np.ndarray:
    """src/codec.rs:359-374 for every row at once; i ascending, mul then add."""
    s = np.zeros((rows.shape[0], HOP), F32)
    Tt = np.ascontiguousarray(T.T)
    for i in range(FRAME):
        s += rows[:, i:i + 1] * Tt[i][None, :]
    return (s * norm).astype(F32)


def imdct_rows(coeffs: np.ndarray, T: np.ndarray, norm) -> np.ndarray:
    """src/codec.rs:377-390; k ascending over ALL k (zeros included)."""
    s = np.zeros((coeffs.shape[0], FRAME), F32)
    for k in range(HOP):
        s += coeffs[:, k:k + 1] * T[k][None, :]
    return (s * norm).astype(F32)


def thresholds_rows(c: np.ndarray, w: np.ndarray, edges: np.ndarray):
    """src/codec.rs:188-240 (vectorised over rows, sequential inside each band)."""
    M = c.shape[0]
    thr = np.zeros((M, HOP), F32)
    gmax = np.fmax(np.fmax.reduce(np.abs(c), axis=1, initial=F32(0.0)), F32(1e-10)).astype(F32)  # f32::max ignores NaN
    cf = max(F32(1.0) - F32(0.7), F32(0.01))
    for b in range(len(edges) - 1):
        s, e = int(edges[b]), min(int(edges[b + 1]), HOP)
        if s >= e:
            continue
        ln = F32(e - s)
        ss = np.zeros(M, F32)
        ws = F32(0.0)
        for i in range(s, e):
            ss = ss + c[:, i] * c[:, i]
            ws = F32(ws + w[i])
        energy = np.sqrt(ss / ln).astype(F32)
        avg_w = F32(ws / ln)
        pf = F32(1.0) / max(avg_w, F32(0.1))
        base = ((energy * F32(0.01)) * cf) * pf
        for i in range(s, e):
            indiv = F32(1.0) / max(w[i], F32(0.1))
            t = (base * indiv).astype(F32)
            peak = np.abs(c[:, i]) > gmax * F32(0.3)
            t = np.where(peak, np.fmin(t, gmax * F32(0.05)), t)  # f32::min ignores NaN
            thr[:, i] = t
    return thr, gmax


def quantise_rows(c: np.ndarray, scale: np.ndarray, thr: np.ndarray) -> np.ndarray:
    """src/codec.rs:270-311 -> dense i16 (0 = dropped)."""
    nfl_c = F32(0.003981071058660746)  # 10f32.powf(-2.4f32); bits 0x3b8273a5 (glibc, MPFR agree)
    nfl = (nfl_c * scale).astype(F32)[:, None]
    a = np.abs(c)
    t = (thr * scale[:, None]).astype(F32)
    keep = (a > nfl) & (a > t)
    normalized = (c / scale[:, None]).astype(F32)
    x = (normalized * F32(32768.0)).astype(F32).astype(np.float64)
    r = np.trunc(x + np.copysign(0.5, x))  # f32::round, half away from zero
    r = np.clip(r, -32768.0, 32767.0)
    q = np.where(keep, r, 0.0).astype(np.int16)
    return q


def encode(pcm: np.ndarray, sr: int, ch: int):
    """src/codec.rs:421-565 + bincode (:774-779).  Returns dict of taps + 'glc' bytes."""
    pcm = np.ascontiguousarray(pcm, F32)
    T, w, norm = tables_cached()
    weights, edges = perceptual(sr)
    rows, nf, P, L = windowed_rows(pcm, ch, w)
    c = mdct_rows(rows, T, norm)
    scale = np.fmax(np.fmax.reduce(np.abs(c), axis=1, initial=F32(0.0)), F32(1e-10)).astype(F32)  # :488, NaN-ignoring
    thr, _ = thresholds_rows(c, weights, edges)
    q = quantise_rows(c, scale, thr)
    nnz = (q != 0).sum(axis=1).astype(np.uint32)
    with np.errstate(invalid="ignore"):
        rv = np.clip((rows * F32(32767.0)).astype(F32), -32768.0, 32767.0)
    raw = np.where(np.isnan(rv), F32(0.0), rv).astype(np.int16)  # `NaN as i16` == 0 in Rust
    is_raw = np.zeros(nf, np.uint8)
    out = [struct.pack("<IHQ", sr, ch, pcm.size), struct.pack("<Q", nf)]
    for f in range(nf):
        sl = slice(f * ch, (f + 1) * ch)
        compressed = int((8 + 4 * nnz[sl].astype(np.int64)).sum()) + 8 + 4 * ch + 64
        raw_size = FRAME * ch * 2
        if F32(compressed) >= F32(raw_size) * F32(0.85):
            is_raw[f] = 1
            out.append(struct.pack("<QQBQ", 0, 0, 1, FRAME * ch))
            out.append(raw[sl].tobytes())  # channel-planar (Q1)
        else:
            out.append(struct.pack("<Q", ch))
            for m in range(f * ch, (f + 1) * ch):
                idx = np.nonzero(q[m])[0]
                out.append(struct.pack("<Q", idx.size))
                pairs = np.empty((idx.size, 2), np.uint16)
                pairs[:, 0] = idx
                pairs[:, 1] = q[m, idx].view(np.uint16)
                out.append(pairs.tobytes())
            out.append(struct.pack("<Q", ch))
            out.append(scale[sl].tobytes())
            out.append(b"\x00")
    out.append(struct.pack("<IIQ", 512, P - L - 512, pcm.size))
    return dict(glc=b"".join(out), n_frames=nf, coeffs=c, scales=scale, nnz=nnz, dense_q=q,
                is_raw=is_raw, thr=thr)


def decode(glc: bytes) -> np.ndarray:
    """src/codec.rs:781-786, :595-768"""
    T, w, norm = tables_cached()
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, glc, pos)
        pos += struct.calcsize("<" + fmt)
        return v

    sr, ch, total = rd("IHQ")
    (nf,) = rd("Q")
    blocks = np.zeros((nf, ch, FRAME), F32)
    for f in range(nf):
        (ncv,) = rd("Q")
        lists = []
        for _ in range(ncv):
            (n,) = rd("Q")
            a = np.frombuffer(glc, np.uint16, n * 2, pos).reshape(n, 2)
            pos += 4 * n
            lists.append(a)
        (ns,) = rd("Q")
        scales = np.frombuffer(glc, F32, ns, pos)
        pos += 4 * ns
        (tag,) = rd("B")
        if tag:
            (rl,) = rd("Q")
            rawv = np.frombuffer(glc, np.int16, rl, pos)
            pos += 2 * rl
            for c in range(ch):
                si = np.arange(FRAME) * ch + c
                ok = si < rl
                v = np.zeros(FRAME, F32)
                v[ok] = rawv[si[ok]].astype(F32) / F32(32767.0)
                blocks[f, c] = v
        else:
            co = np.zeros((ch, HOP), F32)
            for c in range(ch):
                s = max(F32(scales[c]), F32(1e-12))
                qs = lists[c][:, 1].copy().view(np.int16)
                for index, qv in zip(lists[c][:, 0].tolist(), qs.tolist()):
                    if index < HOP:
                        co[c, index] = (F32(qv) / F32(32768.0)) * s
            blocks[f] = imdct_rows(co, T, norm) * w[None, :]
    delay, _pad, orig = rd("IIQ")
    assert pos == len(glc)
    allv = np.empty(((nf + 1) * HOP, ch), F32)
    overlap = np.zeros((ch, HOP), F32)
    for f in range(nf):
        allv[f * HOP:(f + 1) * HOP] = (overlap + blocks[f, :, :HOP]).T
        overlap = blocks[f, :, HOP:].copy()
    allv[nf * HOP:] = overlap.T
    flat = allv.reshape(-1)
    if flat.size > delay:
        flat = flat[delay:]
    if flat.size > orig:
        flat = flat[:orig]
    return flat.copy()
