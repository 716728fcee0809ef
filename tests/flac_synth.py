"""A small FLAC *writer* for tests only: it produces streams that use the parts of RFC 9639 the
reference's own encoder never emits (LPC and constant subframes, left-side / side-right / mid-side
stereo, the 5-bit Rice code book, escaped partitions, wasted bits, 8..32-bit samples, variable
block sizes, uncommon block-size / sample-rate codes, extra metadata blocks), so that the
product's decoder (`glc_flac_load`, the stand-in for claxon) can be checked against known PCM and
against the independent Python decoder in oracle/flac_oracle.py."""
import hashlib

import numpy as np

from oracle.flac_oracle import crc8, crc16


class W:
    def __init__(self):
        self.s = []

    def u(self, v, n):
        if n:
            self.s.append(format(int(v) & ((1 << n) - 1), "0%db" % n))

    def unary(self, z):
        self.s.append("0" * int(z) + "1")

    def align(self):
        n = sum(map(len, self.s)) % 8
        if n:
            self.s.append("0" * (8 - n))

    def bytes(self):
        bits = "".join(self.s)
        assert len(bits) % 8 == 0
        return int(bits, 2).to_bytes(len(bits) // 8, "big") if bits else b""


def _utf8(v):
    if v < 0x80:
        return bytes([v])
    extra = next((e for e in range(1, 6) if v < 1 << (5 * e + 6)), 6)
    lead = (0xFF << (7 - extra)) & 0xFF
    return bytes([lead | ((v >> (6 * extra)) & ((1 << (6 - extra)) - 1))] +
                 [0x80 | ((v >> (6 * i)) & 0x3F) for i in range(extra - 1, -1, -1)])


def _residual(w, res, order, block, method, porder, escape_parts=()):
    w.u(method, 2)
    w.u(porder, 4)
    per = block >> porder
    at = 0
    for p in range(1 << porder):
        cnt = per - order if p == 0 else per
        part = [int(v) for v in res[at:at + cnt]]
        at += cnt
        pbits = 5 if method else 4
        if p in escape_parts:
            w.u((1 << pbits) - 1, pbits)
            width = max([1] + [(v if v >= 0 else ~v).bit_length() + 1 for v in part])
            if not any(part):
                width = 0
            w.u(width, 5)
            for v in part:
                w.u(v, width)
            continue
        mean = sum(abs(v) for v in part) // max(1, cnt)
        k = min(max(mean.bit_length() - 1, 0), (1 << pbits) - 2)
        w.u(k, pbits)
        for v in part:
            folded = (v << 1) if v >= 0 else ((-v - 1) << 1) | 1
            w.unary(folded >> k)
            w.u(folded, k)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(w, s, bps, spec):
    """spec: dict(kind='constant'|'verbatim'|'fixed'|'lpc', order, coefs, precision, shift, method,
    porder, escape_parts, wasted)"""
    s = [int(v) for v in s]
    wasted = spec.get("wasted", 0)
    kind = spec["kind"]
    w.u(0, 1)
    order = spec.get("order", 0)
    w.u({"constant": 0, "verbatim": 1}.get(kind, (8 + order) if kind == "fixed" else (32 + order - 1)), 6)
    if wasted:
        w.u(1, 1)
        w.unary(wasted - 1)
        assert all(v % (1 << wasted) == 0 for v in s)
        s = [v >> wasted for v in s]
        bps -= wasted
    else:
        w.u(0, 1)
    if kind == "constant":
        assert len(set(s)) == 1
        w.u(s[0], bps)
        return
    if kind == "verbatim":
        for v in s:
            w.u(v, bps)
        return
    for v in s[:order]:
        w.u(v, bps)
    if kind == "fixed":
        coefs, shift = FIXED[order], 0
    else:
        coefs, shift = spec["coefs"], spec["shift"]
        w.u(spec["precision"] - 1, 4)
        w.u(shift, 5)
        for c in coefs:
            w.u(c, spec["precision"])
    res = [s[i] - (sum(c * s[i - 1 - j] for j, c in enumerate(coefs)) >> shift) for i in range(order, len(s))]
    _residual(w, res, order, len(s), spec.get("method", 0), spec.get("porder", 0), spec.get("escape_parts", ()))


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13,
               16384: 14, 32768: 15}
RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10,
              96000: 11}
BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def frame(planes, bps, rate, number, specs, assignment="independent", variable=False, bps_from_info=False):
    """planes: list (per channel) of int lists of one block; specs: one subframe spec per channel."""
    nch, block = len(planes), len(planes[0])
    w = W()
    w.u(0x3FFE, 14)
    w.u(0, 1)
    w.u(1 if variable else 0, 1)
    bcode = BLOCK_CODES.get(block, 6 if block <= 256 else 7)
    w.u(bcode, 4)
    if rate in RATE_CODES:
        rcode = RATE_CODES[rate]
    elif rate % 1000 == 0 and rate < 256000:
        rcode = 12
    elif rate < 65536:
        rcode = 13
    elif rate % 10 == 0:
        rcode = 14
    else:
        rcode = 0
    w.u(rcode, 4)
    code = {"independent": nch - 1, "left_side": 8, "side_right": 9, "mid_side": 10}[assignment]
    w.u(code, 4)
    w.u(0 if bps_from_info else BPS_CODES[bps], 3)
    w.u(0, 1)
    head = w.bytes() + _utf8(number)
    if bcode == 6:
        head += bytes([block - 1])
    elif bcode == 7:
        head += (block - 1).to_bytes(2, "big")
    if rcode == 12:
        head += bytes([rate // 1000])
    elif rcode == 13:
        head += rate.to_bytes(2, "big")
    elif rcode == 14:
        head += (rate // 10).to_bytes(2, "big")
    head += bytes([crc8(head)])
    chans = [list(map(int, p)) for p in planes]
    widths = [bps] * nch
    if assignment == "left_side":
        chans = [chans[0], [a - b for a, b in zip(chans[0], chans[1])]]
        widths = [bps, bps + 1]
    elif assignment == "side_right":
        chans = [[a - b for a, b in zip(chans[0], chans[1])], chans[1]]
        widths = [bps + 1, bps]
    elif assignment == "mid_side":
        chans = [[(a + b) >> 1 for a, b in zip(chans[0], chans[1])], [a - b for a, b in zip(chans[0], chans[1])]]
        widths = [bps, bps + 1]
    w = W()
    for c in range(nch):
        _subframe(w, chans[c], widths[c], specs[c])
    w.align()
    body = head + w.bytes()
    return body + crc16(body).to_bytes(2, "big")


def stream(pcm, channels, bps, rate, blocks, specs_for, assignment_for=lambda i: "independent", variable=False,
           extra_metadata=(), total_known=True, bps_from_info=False):
    """pcm: interleaved ints; blocks: list of block sizes covering the stream."""
    pcm = np.asarray(pcm, np.int64).reshape(-1, channels)
    width = (bps + 7) // 8
    md5 = hashlib.md5(b"".join(int(v).to_bytes(width, "little", signed=True) for v in pcm.reshape(-1))).digest()
    w = W()
    w.u(min(blocks), 16)
    w.u(max(blocks), 16)
    w.u(0, 24)
    w.u(0, 24)
    w.u(rate, 20)
    w.u(channels - 1, 3)
    w.u(bps - 1, 5)
    w.u(pcm.shape[0] if total_known else 0, 36)
    info = w.bytes() + md5
    out = [b"fLaC", bytes([0x00 if extra_metadata else 0x80]) + len(info).to_bytes(3, "big") + info]
    for j, (kind, payload) in enumerate(extra_metadata):
        last = 0x80 if j == len(extra_metadata) - 1 else 0
        out.append(bytes([last | kind]) + len(payload).to_bytes(3, "big") + payload)
    at = 0
    for i, b in enumerate(blocks):
        planes = [pcm[at:at + b, c].tolist() for c in range(channels)]
        out.append(frame(planes, bps, rate, at if variable else i, specs_for(i), assignment_for(i), variable,
                         bps_from_info))
        at += b
    assert at == pcm.shape[0]
    return b"".join(out)
