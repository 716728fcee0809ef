"""CPU oracle for the FLAC export / import row.  TEST INFRASTRUCTURE ONLY: nothing
under gapless-lossy-codec_amd/ may import this module.

Two independent pieces:

* ``encode_flac_with_level`` — a numpy restatement of the reference's own encoder,
  /root/reference/src/flac.rs (each function cites the lines it follows).  The product's
  ``glc_flac_encode`` must produce the same bytes.
* ``decode_flac`` — a pure-Python FLAC decoder written from RFC 9639 alone (it shares no code with
  the product's decoder).  It checks every CRC-8 / CRC-16 and the STREAMINFO MD5 (hashlib), so a
  stream it accepts is a valid FLAC stream that decodes to the expected PCM; that is what stands in
  for `claxon`, the crate the reference's tests read their own output back with
  (tests/test_flac.rs:24).

Parity status: **unpinned by the reference** — it ships no .flac fixture and no known-answer test
(tests/test_flac.rs asserts only lengths and an RMS bound), and no Rust toolchain exists here to
run it.  What pins the bytes is the source itself (integer arithmetic only) plus RFC 9639 validity.
"""
from __future__ import annotations

import hashlib

import numpy as np

MAX_RICE_PARAM_4BIT = 14  # flac.rs:12


# ------------------------------------------------------------------ checksums (flac.rs:19-80)
def _crc_table(width: int, poly: int):
    top, mask = 1 << (width - 1), (1 << width) - 1
    out = []
    for i in range(256):
        c = i << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
        out.append(c)
    return out


_CRC8 = _crc_table(8, 0x07)
_CRC16 = _crc_table(16, 0x8005)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c = _CRC8[c ^ b]
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c = ((c << 8) ^ _CRC16[(c >> 8) ^ b]) & 0xFFFF
    return c


# ------------------------------------------------------------------ MD5Context (flac.rs:83-302)
class RefMD5:
    """The reference's hand-written MD5, restated with its own buffering logic (update :102-150,
    finalize :276-301) so that the claim "it is standard MD5" is itself a test (vs hashlib)."""

    _S = [7, 12, 17, 22] * 4 + [5, 9, 14, 20] * 4 + [4, 11, 16, 23] * 4 + [6, 10, 15, 21] * 4
    _K = [int(abs(np.sin(np.float64(i + 1))) * 2.0 ** 32) & 0xFFFFFFFF for i in range(64)]

    def __init__(self):
        self.state = [0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476]
        self.count = [0, 0]
        self.buffer = bytearray(64)

    def update(self, data: bytes):
        n = len(data)
        index = (self.count[0] >> 3) & 0x3F
        add = (n << 3) & 0xFFFFFFFF
        self.count[0] = (self.count[0] + add) & 0xFFFFFFFF
        if self.count[0] < add:
            self.count[1] = (self.count[1] + 1) & 0xFFFFFFFF
        self.count[1] = (self.count[1] + ((n & 0xFFFFFFFF) >> 29)) & 0xFFFFFFFF
        part = 64 - index
        at = 0
        if n >= part:
            self.buffer[index:index + part] = data[:part]
            self.transform(bytes(self.buffer))
            i = part
            while i + 63 < n:
                self.transform(data[i:i + 64])
                i += 64
            at = i
        if at < n:
            where = index if at == 0 else 0
            self.buffer[where:where + n - at] = data[at:]

    def transform(self, block: bytes):
        a, b, c, d = self.state
        x = [int.from_bytes(block[4 * i:4 * i + 4], "little") for i in range(16)]
        for i in range(64):
            if i < 16:
                f, g = (b & c) | (~b & d), i
            elif i < 32:
                f, g = (b & d) | (c & ~d), (5 * i + 1) % 16
            elif i < 48:
                f, g = b ^ c ^ d, (3 * i + 5) % 16
            else:
                f, g = c ^ (b | ~d), (7 * i) % 16
            t = (a + f + self._K[i] + x[g]) & 0xFFFFFFFF
            a, d, c = d, c, b
            b = (b + ((t << self._S[i]) | (t >> (32 - self._S[i])))) & 0xFFFFFFFF
        self.state = [(s + v) & 0xFFFFFFFF for s, v in zip(self.state, (a, b, c, d))]

    def finalize(self) -> bytes:
        bits = self.count[0].to_bytes(4, "little") + self.count[1].to_bytes(4, "little")
        index = (self.count[0] >> 3) & 0x3F
        pad = 56 - index if index < 56 else 120 - index
        self.update(b"\x80" + bytes(pad - 1))
        self.update(bits)
        return b"".join(s.to_bytes(4, "little") for s in self.state)


def compute_md5_ref(i16: np.ndarray) -> bytes:
    """compute_md5, flac.rs:305-318: two bytes per update call (small inputs only: pure Python)."""
    ctx = RefMD5()
    raw = np.ascontiguousarray(i16, "<i2").tobytes()
    for i in range(0, len(raw), 2):
        ctx.update(raw[i:i + 2])
    return ctx.finalize()


# ------------------------------------------------------------------ bit assembly
def _field(value: int, nbits: int) -> np.ndarray:
    """write_bits, flac.rs:340-380: the low `nbits` bits of value, most significant first."""
    value &= (1 << nbits) - 1
    return np.array([(value >> (nbits - 1 - i)) & 1 for i in range(nbits)], np.uint8)


def _fields16(values: np.ndarray) -> np.ndarray:
    """A run of 16-bit two's-complement samples (`sample as u64` then 16 bits, flac.rs:727,735)."""
    v = values.astype(np.int64) & 0xFFFF
    return ((v[:, None] >> np.arange(15, -1, -1)) & 1).astype(np.uint8).reshape(-1)


def _utf8_number(v: int) -> np.ndarray:  # write_utf8_number, flac.rs:427-478
    if v < 0x80:
        by = [v]
    else:
        extra = next((e for e in range(1, 6) if v < 1 << (5 * e + 6)), 6)  # 0x800, 0x10000, 0x200000, ...
        lead = (0xFF << (7 - extra)) & 0xFF
        by = [lead | ((v >> (6 * extra)) & ((1 << (6 - extra)) - 1))]
        by += [0x80 | ((v >> (6 * i)) & 0x3F) for i in range(extra - 1, -1, -1)]
    return np.concatenate([_field(b, 8) for b in by])


def calculate_rice_parameter(residual: np.ndarray) -> int:  # flac.rs:515-552
    if residual.size == 0:
        return 0
    mean = int(np.abs(residual.astype(np.int64)).sum()) // residual.size
    if mean == 0:
        return 0
    param, test = 0, mean
    while test > 0 and param < MAX_RICE_PARAM_4BIT:
        test >>= 1
        if test > 0:
            param += 1
    if param > 0 and mean < (1 << (param - 1)):
        param -= 1
    return min(param, MAX_RICE_PARAM_4BIT)


def _rice_partition(residual: np.ndarray, k: int) -> np.ndarray:  # encode_rice_partition, flac.rs:555-584
    r = residual.astype(np.int64)
    folded = np.where(r >= 0, r << 1, ((-(r + 1)) << 1) | 1)
    msb, lsb = folded >> k, folded & ((1 << k) - 1)
    ends = np.cumsum(msb + 1 + k)
    starts = ends - (msb + 1 + k)
    bits = np.zeros(int(ends[-1]) if r.size else 0, np.uint8)
    bits[starts + msb] = 1  # `msb` zeros, then the one (write_unary, flac.rs:395-403)
    for j in range(k):
        bits[starts + msb + 1 + j] = (lsb >> (k - 1 - j)) & 1
    return bits


def _trailing_zeros(v: int) -> int:
    return (v & -v).bit_length() - 1 if v else 64


def _residual(res: np.ndarray, order: int, block: int, level: int) -> list:  # encode_residual, flac.rs:587-684
    tz = min(_trailing_zeros(block), 8)
    porder = 0 if level == 0 else min(2, tz) if level <= 2 else min(4, tz) if level <= 5 else min(6, tz)
    while porder > 0:
        per = block >> porder
        if per > order and per >= 4:
            break
        porder -= 1
    out = [_field(0, 2), _field(porder, 4)]
    per = block >> porder
    at = 0
    for p in range(1 << porder):
        cnt = per - order if p == 0 else per
        if cnt == 0:
            continue
        part = res[at:at + cnt]
        at += cnt
        k = calculate_rice_parameter(part)  # never exceeds 14, so the escape branch :643-672 is dead
        out += [_field(k, 4), _rice_partition(part, k)]
    return out


def _subframe(s: np.ndarray, level: int) -> list:  # encode_subframe :687-745, apply_fixed_predictor :481-512
    block = s.size
    order = {0: 0, 1: 1, 2: 2, 3: 3, 4: 3}.get(level, 4)
    if block < order:
        order = 0
    out = [_field(0, 1), _field(0b000001 if order == 0 else 0b001000 | order, 6), _field(0, 1)]
    if order == 0:
        return out + [_fields16(s)]
    out.append(_fields16(s[:order]))
    x = s.astype(np.int64)
    res = x.copy()
    for _ in range(order):  # order-th finite difference == the reference's closed-form predictors
        res = np.concatenate([[0], np.diff(res)])
    return out + _residual(res[order:], order, block, level)


_BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12,
                8192: 13, 16384: 14, 32768: 15}
_RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9,
               48000: 10, 96000: 11}


def _frame(pcm: np.ndarray, channels: int, sample_rate: int, frame_number: int, level: int) -> bytes:
    """encode_frame, flac.rs:748-905; pcm is this block's interleaved i16."""
    block = pcm.size // channels
    bcode = _BLOCK_CODES.get(block, 6 if block < 256 else 7)
    head = [_field(0x3FFE, 14), _field(0, 1), _field(0, 1), _field(bcode, 4),
            _field(_RATE_CODES.get(sample_rate, 0), 4),
            _field(0 if channels == 1 else 1 if channels == 2 else channels - 1, 4), _field(0b100, 3), _field(0, 1),
            _utf8_number(frame_number)]
    if bcode == 6:
        head.append(_field((block - 1) & 0xFF, 8))
    elif bcode == 7:
        head.append(_field(block - 1, 16))
    hb = np.packbits(np.concatenate(head)).tobytes()
    parts = [np.unpackbits(np.frombuffer(hb + bytes([crc8(hb)]), np.uint8))]
    planes = pcm.reshape(block, channels)
    for c in range(channels):
        parts += _subframe(planes[:, c], level)
    body = np.packbits(np.concatenate(parts)).tobytes()  # packbits zero-pads: byte_align, flac.rs:405-413
    return body + crc16(body).to_bytes(2, "big")


def to_i16(samples: np.ndarray) -> np.ndarray:
    """`(s * 32767.0).clamp(-32768.0, 32767.0) as i16`, flac.rs:955-958 (NaN casts to 0)."""
    v = np.asarray(samples, np.float32) * np.float32(32767.0)
    v = np.where(np.isnan(v), np.float32(0), np.clip(v, np.float32(-32768.0), np.float32(32767.0)))
    return np.trunc(v).astype(np.int16)


def encode_flac_with_level(samples, sample_rate: int, channels: int, level: int, md5=None) -> bytes:
    """flac.rs:947-1053.  `md5` lets a test substitute the reference-structured MD5 for hashlib."""
    i16 = to_i16(np.asarray(samples, np.float32).reshape(-1))
    total = i16.size // channels
    if total < 16:
        raise ValueError(f"FLAC requires at least 16 samples per channel, got {total}")
    if level > 8:
        raise ValueError(f"Invalid compression level {level}, must be 0-8")
    block = max(min(1152 if level <= 2 else 4096, total), 16)
    digest = md5(i16) if md5 else hashlib.md5(i16.astype("<i2").tobytes()).digest()
    info = np.concatenate([_field(1, 1), _field(0, 7), _field(34, 24), _field(block, 16), _field(block, 16),
                           _field(0, 24), _field(0, 24), _field(sample_rate, 20), _field(channels - 1, 3),
                           _field(15, 5), _field(total, 36)])
    out = [b"fLaC", np.packbits(info).tobytes(), digest]
    at, number = 0, 0
    while at < i16.size:  # flac.rs:1021-1050
        cur = min(block, (i16.size - at) // channels)
        if cur == 0:
            break
        out.append(_frame(i16[at:at + cur * channels], channels, sample_rate, number, level))
        at += cur * channels
        number += 1
    return b"".join(out)


# ------------------------------------------------------------------ independent decoder (RFC 9639)
class FlacError(ValueError):
    pass


class _Bits:
    def __init__(self, data: bytes):
        self.s = bin(int.from_bytes(b"\x01" + data, "big"))[3:]  # leading 1 keeps the zero prefix
        self.pos = 0

    def u(self, n: int) -> int:
        if n == 0:
            return 0
        if self.pos + n > len(self.s):
            raise FlacError("out of data")
        v = int(self.s[self.pos:self.pos + n], 2)
        self.pos += n
        return v

    def i(self, n: int) -> int:
        v = self.u(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self) -> int:
        j = self.s.find("1", self.pos)
        if j < 0:
            raise FlacError("out of data")
        z = j - self.pos
        self.pos = j + 1
        return z


def _dec_residual(r: _Bits, block: int, order: int) -> list:
    method = r.u(2)
    if method > 1:
        raise FlacError("reserved residual method")
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    porder = r.u(4)
    if block % (1 << porder):
        raise FlacError("bad partition order")
    per = block >> porder
    out = []
    for p in range(1 << porder):
        cnt = per - order if p == 0 else per
        if cnt < 0:
            raise FlacError("partition shorter than predictor order")
        k = r.u(pbits)
        if k == esc:
            raw = r.u(5)
            out += [r.i(raw) for _ in range(cnt)]
        else:
            for _ in range(cnt):
                folded = (r.unary() << k) | r.u(k)
                out.append((folded >> 1) ^ -(folded & 1))
    return out


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _dec_subframe(r: _Bits, block: int, bps: int) -> list:
    if r.u(1):
        raise FlacError("padding bit")
    kind = r.u(6)
    wasted = 0
    if r.u(1):
        wasted = r.unary() + 1
        bps -= wasted
    if kind == 0:
        s = [r.i(bps)] * block
    elif kind == 1:
        s = [r.i(bps) for _ in range(block)]
    elif 8 <= kind <= 12 or kind >= 32:
        if kind >= 32:
            order = (kind & 31) + 1
            s = [r.i(bps) for _ in range(order)]
            prec = r.u(4) + 1
            if prec == 16:
                raise FlacError("reserved precision")
            shift = r.i(5)
            if shift < 0:
                raise FlacError("negative shift")
            coef = [r.i(prec) for _ in range(order)]
        else:
            order = kind - 8
            s = [r.i(bps) for _ in range(order)]
            coef, shift = _FIXED[order], 0
        for e in _dec_residual(r, block, order):
            pred = sum(c * s[-1 - j] for j, c in enumerate(coef)) >> shift
            s.append(e + pred)
    else:
        raise FlacError("reserved subframe type")
    return [v << wasted for v in s] if wasted else s


def decode_flac(data: bytes, verify_md5: bool = True):
    """-> (int samples interleaved as np.int64, sample_rate, channels, bits_per_sample).  Raises
    FlacError on any CRC / MD5 / syntax violation."""
    if data[:4] != b"fLaC":
        raise FlacError("no fLaC marker")
    pos, last, info = 4, False, None
    while not last:
        last, kind = bool(data[pos] & 0x80), data[pos] & 0x7F
        n = int.from_bytes(data[pos + 1:pos + 4], "big")
        if kind == 0:
            r = _Bits(data[pos + 4:pos + 4 + n])
            info = dict(min_block=r.u(16), max_block=r.u(16), min_frame=r.u(24), max_frame=r.u(24), rate=r.u(20),
                        ch=r.u(3) + 1, bps=r.u(5) + 1, total=r.u(36), md5=data[pos + 4 + 18:pos + 4 + 34])
        pos += 4 + n
    if info is None:
        raise FlacError("no STREAMINFO")
    ch, bps = info["ch"], info["bps"]
    planes = [[] for _ in range(ch)]
    expect = 0
    while pos < len(data):
        r = _Bits(data[pos:pos + 16])
        if r.u(14) != 0x3FFE or r.u(1):
            raise FlacError("lost sync")
        variable = r.u(1)
        bcode, rcode, ccode, scode = r.u(4), r.u(4), r.u(4), r.u(3)
        if r.u(1):
            raise FlacError("reserved bit")
        lead = r.u(8)
        nb = 0 if lead < 0x80 else 8 - (lead ^ 0xFF).bit_length() - 1
        number = lead if lead < 0x80 else lead & ((1 << (6 - nb)) - 1)
        for _ in range(nb):
            cont = r.u(8)
            if cont & 0xC0 != 0x80:
                raise FlacError("bad coded number")
            number = (number << 6) | (cont & 0x3F)
        if bcode == 0:
            raise FlacError("reserved block code")
        block = 192 if bcode == 1 else 576 << (bcode - 2) if bcode <= 5 else r.u(8) + 1 if bcode == 6 \
            else r.u(16) + 1 if bcode == 7 else 256 << (bcode - 8)
        if rcode == 12:
            r.u(8)
        elif rcode in (13, 14):
            r.u(16)
        elif rcode == 15:
            raise FlacError("bad rate code")
        hdr = r.pos // 8
        if crc8(data[pos:pos + hdr]) != data[pos + hdr]:
            raise FlacError("CRC-8")
        if not variable and number != expect:
            raise FlacError(f"frame number {number}, expected {expect}")
        expect += 1
        fb = {0: bps, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}.get(scode)
        if fb != bps:
            raise FlacError("sample size")
        nch = ccode + 1 if ccode < 8 else 2
        if ccode > 10 or nch != ch:
            raise FlacError("channel assignment")
        r = _Bits(data[pos + hdr + 1:])
        sub = []
        for c in range(nch):
            side = (ccode == 8 and c == 1) or (ccode == 9 and c == 0) or (ccode == 10 and c == 1)
            sub.append(_dec_subframe(r, block, bps + (1 if side else 0)))
        body = hdr + 1 + (r.pos + 7) // 8
        if crc16(data[pos:pos + body]) != int.from_bytes(data[pos + body:pos + body + 2], "big"):
            raise FlacError("CRC-16")
        if ccode == 8:
            sub[1] = [a - b for a, b in zip(sub[0], sub[1])]
        elif ccode == 9:
            sub[0] = [a + b for a, b in zip(sub[0], sub[1])]
        elif ccode == 10:
            mid = [(m << 1) | (s & 1) for m, s in zip(sub[0], sub[1])]
            sub = [[(m + s) >> 1 for m, s in zip(mid, sub[1])], [(m - s) >> 1 for m, s in zip(mid, sub[1])]]
        for c in range(nch):
            planes[c] += sub[c]
        pos += body + 2
    pcm = np.array(planes, np.int64).T.reshape(-1)
    if info["total"] and info["total"] * ch != pcm.size:
        raise FlacError("sample count differs from STREAMINFO")
    if verify_md5 and any(info["md5"]):
        width = (bps + 7) // 8
        raw = b"".join(int(v).to_bytes(width, "little", signed=True) for v in pcm) if width != 2 \
            else pcm.astype("<i2").tobytes()
        if hashlib.md5(raw).digest() != info["md5"]:
            raise FlacError("MD5")
    return pcm, info["rate"], ch, bps
