"""Generates tests/golden/golden.json + *.glc from the CPU oracle.

The reference (Rust) cannot be compiled in this image and its own tests hold no golden vectors,
so these fixtures freeze the ORACLE's output (cross-checked bit-for-bit against the independent
numpy restatement in tests/test_oracle.py).  They are data: inputs are named by generator
parameters, outputs are hashes, small integer lists and the .glc bytes.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from conftest import gen_noise, gen_tone  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [
    dict(name="cfg1_sine440_44k_stereo_2s", sample_rate=44100, channels=2,
         generator=dict(kind="sine", f0=440.0, dur=2.0)),
    dict(name="noise_44k_stereo_0p25s", sample_rate=44100, channels=2,
         generator=dict(kind="noise", seed=12345, dur=0.25)),
    dict(name="sweep_48k_mono_1s", sample_rate=48000, channels=1,
         generator=dict(kind="sweep", f0=100.0, f1=10000.0, dur=1.0)),
]


def main():
    T, w, n = O.tables()
    gold = dict(tables=dict(cos_table_sha256=hashlib.sha256(T.tobytes()).hexdigest(),
                            window_sha256=hashlib.sha256(w.tobytes()).hexdigest(),
                            norm_bits=int(np.float32(n).view(np.uint32))), cases=[])
    for c in CASES:
        g = c["generator"]
        if g["kind"] == "noise":
            x = gen_noise(c["sample_rate"], c["channels"], g["dur"], g["seed"])
        else:
            x = gen_tone(g["kind"], g["f0"], c["sample_rate"], c["channels"], g["dur"], g.get("f1", 0.0))
        enc = O.encode(x, c["sample_rate"], c["channels"], taps=True)
        dec, _, _ = O.decode(enc.glc)
        fn = c["name"] + ".glc"
        with open(os.path.join(HERE, fn), "wb") as fh:
            fh.write(enc.glc)
        gold["cases"].append(dict(c, n_samples=int(x.size), n_frames=int(enc.n_frames),
                                  input_sha256=hashlib.sha256(x.tobytes()).hexdigest(),
                                  glc_file=fn, glc_len=len(enc.glc),
                                  glc_sha256=hashlib.sha256(enc.glc).hexdigest(),
                                  nnz=enc.nnz.tolist(), is_raw=enc.is_raw.tolist(),
                                  scale_bits=[int(v) for v in enc.scales.view(np.uint32)],
                                  decoded_sha256=hashlib.sha256(dec.tobytes()).hexdigest()))
    with open(os.path.join(HERE, "golden.json"), "w") as fh:
        json.dump(gold, fh, indent=1)
    print("wrote", len(gold["cases"]), "cases")


if __name__ == "__main__":
    main()
