"""HBM-side traffic of the encode kernels from PMC counters, calibrated on known byte counts.

Runs (on the GPU box) separate `rocprofv3 --pmc` passes - FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950 (MI355X_MICROARCH.md, PMC slots) - over
  1. build/fetch_calib: kernels that move exactly 512 MiB each with the access patterns K1 uses
     (128-byte PCM segments by buffer_load_dwordx4, table rows by global_load_lds_dwordx4) and two
     plain ones -> counter bytes per true byte for every pattern;
  2. `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline`: per-launch counter means of K1 / K2 / D1.
and writes profiles/k1_traffic.json (what bench.py reports as roofline.traffic, with the commit it was
taken at) plus the per-kernel CSV means under profiles/<tag>_pmc/.

Usage (from the repo root, inside gpurun):  python3 tools/pmc_traffic.py --commit <sha> [--tag r02]
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K1 = "k_mdct_fwd_st"
ALGO_K1_BYTES = 8388608 * 4 + 8 * 1024 * 1024 + 8192 * 1024 * 4   # PCM once + table once + coefficient write
PCM_ONCE = 8388608 * 4 * 129 / 128   # every PCM byte once per row tile of 128 frames (+ the 1024-sample halo)


def run_pass(counters, cmd, outdir):
    shutil.rmtree(outdir, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    full = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", outdir, "--"] + cmd
    p = subprocess.run(full, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode != 0:
        sys.stderr.write(p.stdout.decode("utf-8", "replace")[-2000:])
        raise SystemExit(f"pass {counters} failed")
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(outdir, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean_tail(v):
    """Mean over the second half of the dispatches (the first ones see cold caches / clocks)."""
    v = v[len(v) // 2:] if len(v) > 3 else v
    return sum(v) / len(v)


def pick(acc, needle, counter):
    for k, c in acc.items():
        if needle in k and counter in c:
            return mean_tail(c[counter]), len(c[counter])
    return None, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--commit", default="unknown")
    ap.add_argument("--tag", default="r02")
    args = ap.parse_args()
    scratch = os.path.join(ROOT, "gpurun_out", f"{args.tag}_pmc")
    calib = [os.path.join(ROOT, "build", "fetch_calib")]
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--lean"]
    true_kb = 512 * 1024.0

    cf = run_pass(["FETCH_SIZE"], calib, os.path.join(scratch, "calib_fetch"))
    cw = run_pass(["WRITE_SIZE"], calib, os.path.join(scratch, "calib_write"))
    factors = {}
    for name, pattern in (("calib_read_b32", "4 B/lane coalesced global_load_dword"),
                          ("calib_read_b128", "16 B/lane coalesced global_load_dwordx4"),
                          ("calib_read_segments", "K1 PCM: 128-B segments, 8 lanes x buffer_load_dwordx4, 8 KiB apart"),
                          ("calib_read_lds_dma", "k_mdct_fwd_dma's table: 512-B rows by global_load_lds_dwordx4"),
                          ("calib_read_scalar", "K1 table: 32-B pieces of 4-KiB rows by s_load_dwordx8")):
        v, n = pick(cf, name, "FETCH_SIZE")
        factors[name] = {"pattern": pattern, "FETCH_SIZE_KB": v, "dispatches": n,
                         "counter_bytes_per_true_byte": None if v is None else round(v / true_kb, 4)}
    for name, pattern in (("calib_write_b32", "4 B/lane coalesced store"), ("calib_write_b128", "16 B/lane coalesced store"),
                          ("calib_write_rows", "K1 output: 2 x 16 B per lane and row, a lane's rows 4 KiB apart")):
        v, n = pick(cw, name, "WRITE_SIZE")
        factors[name] = {"pattern": pattern, "WRITE_SIZE_KB": v, "dispatches": n,
                         "counter_bytes_per_true_byte": None if v is None else round(v / true_kb, 4)}

    bf = run_pass(["FETCH_SIZE"], bench, os.path.join(scratch, "bench_fetch"))
    bw = run_pass(["WRITE_SIZE"], bench, os.path.join(scratch, "bench_write"))
    bh = run_pass(["TCC_HIT_sum", "TCC_MISS_sum"], bench, os.path.join(scratch, "bench_tcc"))
    kernels = {}
    for label, needle in (("K1", K1), ("K2", "k_quantize")):
        f, n = pick(bf, needle, "FETCH_SIZE")
        w, _ = pick(bw, needle, "WRITE_SIZE")
        h, _ = pick(bh, needle, "TCC_HIT_sum")
        m, _ = pick(bh, needle, "TCC_MISS_sum")
        kernels[label] = {"kernel": needle, "dispatches": n, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
                          "TCC_HIT_sum": h, "TCC_MISS_sum": m}
    # K1 reads with two patterns that the counter weighs differently (FETCH_SIZE reports half the bytes of the
    # dwordx4 segment reads and all the bytes of the scalar loads): one raw figure, two unknowns.  The PCM
    # side is pinned from the algorithm - with the XCD-aware tile map the 8 workgroups that share a row tile
    # sit on one XCD, so a PCM byte misses that L2 once: PCM_ONCE bytes - and the rest of the raw figure is the table.
    fs = [factors[k]["counter_bytes_per_true_byte"] for k in ("calib_read_segments", "calib_read_scalar")]
    fw = factors["calib_write_rows"]["counter_bytes_per_true_byte"]
    k1 = kernels["K1"]
    out = {
        "kernel": "glc::k1::k_mdct_fwd_st<4, 2, 2, 4, 16, 16, 0, false> (256x128 tile, 1024 threads, table values by scalar loads, PCM by dwordx4 segments)",
        "commit": args.commit,
        "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 10, corrected by the "
                  "factors tools/fetch_calib.hip measures for K1's own access patterns in the same session",
        "calibration": factors,
        "kernels": kernels,
        "algorithmic_bytes_per_launch": ALGO_K1_BYTES,
    }
    if None not in fs and fw and k1["FETCH_SIZE_KB_raw"] is not None:
        raw = k1["FETCH_SIZE_KB_raw"] * 1024.0
        pcm_once = PCM_ONCE
        table = max(0.0, raw - fs[0] * pcm_once) / fs[1]
        rd = pcm_once + table
        wr = k1["WRITE_SIZE_KB_raw"] * 1024.0 / fw
        out["read_correction"] = {"segments": fs[0], "scalar": fs[1],
                                  "pcm_bytes_assumed": int(pcm_once), "table_bytes": int(table),
                                  "bounds_if_one_pattern": [int(raw / max(fs)), int(raw / min(fs))]}
        out["write_correction"] = fw
        out["read_bytes_per_launch"] = int(rd)
        out["write_bytes_per_launch"] = int(wr)
        out["hbm_bytes_per_launch"] = int(rd + wr)
        out["vs_algorithmic"] = round((rd + wr) / ALGO_K1_BYTES, 3)
        if k1["TCC_HIT_sum"] is not None:
            out["l2_hit_rate"] = round(k1["TCC_HIT_sum"] / (k1["TCC_HIT_sum"] + k1["TCC_MISS_sum"]), 4)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "k1_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
