"""Generates tests/golden/flac_golden.json + *.flac from the FLAC oracle (oracle/flac_oracle.py, the
numpy restatement of the reference's src/flac.rs).  The reference ships no .flac fixture and cannot
be run here, so these freeze the ORACLE's bytes; every file is also accepted (CRC-8, CRC-16, MD5)
by that module's independent RFC 9639 decoder.  Inputs are the reference's own test signals
(tests/test_flac.rs).  Run from the repo root:  python tests/golden/make_golden_flac.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import flac_oracle as F  # noqa: E402


def signal(kind):
    f32 = np.float32
    if kind == "sine_4410":       # tests/test_flac.rs:69-78
        t = np.arange(4410, dtype=f32) / f32(44100.0)
        return np.sin(f32(2.0 * np.pi) * f32(440.0) * t).astype(f32) * f32(0.8)
    if kind == "lcg_noise_8820":  # tests/test_flac.rs:81-94
        seed, out = 12345, []
        for _ in range(8820):
            seed = (seed * 1103515245 + 12345) & 0xFFFFFFFF
            out.append(f32(((seed >> 16) & 0x7FFF)) / f32(32768.0) * f32(2.0) - f32(1.0))
        return np.array(out, f32)
    if kind == "stereo_4410":     # tests/test_flac.rs:97-110
        t = np.arange(4410, dtype=f32) / f32(44100.0)
        lr = np.stack([np.sin(f32(2 * np.pi * 440.0) * t) * f32(0.5), np.sin(f32(2 * np.pi * 880.0) * t) * f32(0.5)], 1)
        return lr.astype(f32).reshape(-1)
    if kind == "ramp_16":         # tests/test_flac.rs:123-133
        return (np.arange(16, dtype=f32) / f32(16.0) * f32(2.0) - f32(1.0)).astype(f32)
    raise ValueError(kind)


CASES = [("sine_4410", 44100, 1, 5), ("lcg_noise_8820", 44100, 1, 5), ("stereo_4410", 44100, 2, 5),
         ("ramp_16", 8000, 1, 5), ("sine_4410", 44100, 1, 0), ("stereo_4410", 44100, 2, 2), ("lcg_noise_8820", 44100, 1, 8)]


def main():
    gold = []
    for kind, sr, ch, level in CASES:
        x = signal(kind)
        data = F.encode_flac_with_level(x, sr, ch, level)
        pcm, r, c, bps = F.decode_flac(data)
        assert (r, c, bps) == (sr, ch, 16) and np.array_equal(pcm, F.to_i16(x).astype(np.int64))
        fn = f"{kind}_{sr}_{ch}ch_l{level}.flac"
        with open(os.path.join(HERE, fn), "wb") as fh:
            fh.write(data)
        gold.append(dict(signal=kind, sample_rate=sr, channels=ch, level=level, file=fn, length=len(data),
                         sha256=hashlib.sha256(data).hexdigest(), input_sha256=hashlib.sha256(x.tobytes()).hexdigest()))
    with open(os.path.join(HERE, "flac_golden.json"), "w") as fh:
        json.dump(gold, fh, indent=1)
    print("wrote", len(gold), "FLAC cases")


if __name__ == "__main__":
    main()
