"""Memory- and issue-side counters of the inverse transform's apply kernel (D1) at BASELINE config 2 ->
gpurun_out/<tag>_d1_pmc.txt.  Separate `rocprofv3 --pmc` passes over `python3 tools/bench_d1.py 30 0`.
Usage (repo root, inside gpurun):  python3 tools/pmc_d1.py [--tag r03]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_traffic import pick, run_pass  # noqa: E402

K = "k_imdct_apply"
PASSES = [
    ["GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_SMEM"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    args = ap.parse_args()
    scratch = os.path.join(ROOT, "gpurun_out", f"{args.tag}_d1_pmc")
    cmd = ["python3", os.path.join(ROOT, "tools", "bench_d1.py"), "30", "0"]
    got = {}
    lines = [f"# rocprofv3 --pmc ({len(PASSES)} separate passes) -- python3 tools/bench_d1.py 30 0",
             f"# kernel {K}<true, true, true> on the kept plan, 8192 rows (BASELINE config 2 chord, nnz/row 114); second half of the dispatches",
             "# pass counter n mean"]
    for i, counters in enumerate(PASSES, 1):
        try:
            acc = run_pass(counters, cmd, os.path.join(scratch, f"pass{i}"))
        except SystemExit as e:
            lines.append(f"{i} FAILED {counters}: {e}")
            continue
        for c in counters:
            v, n = pick(acc, K, c)
            if v is not None:
                got[c] = v
                lines.append(f"{i} {c} {n} {v:.0f}")
    lines.append("# derived")
    if "GRBM_GUI_ACTIVE" in got:
        cyc = got["GRBM_GUI_ACTIVE"] / 8.0
        lines.append(f"cycles_per_launch {cyc:.0f}")
        if "SQ_ACTIVE_INST_VALU" in got:
            lines.append(f"valu_busy_frac_of_simd_cycles {got['SQ_ACTIVE_INST_VALU'] * 4.0 / (cyc * 1024.0):.4f}")
    if "TCC_REQ_sum" in got:
        lines.append(f"l2_request_bytes_per_launch_if_128B {got['TCC_REQ_sum'] * 128:.0f}   (table rows: 8 KiB per unit and union entry)")
    if "TCC_HIT_sum" in got and "TCC_MISS_sum" in got:
        lines.append(f"l2_hit_rate {got['TCC_HIT_sum'] / (got['TCC_HIT_sum'] + got['TCC_MISS_sum']):.4f}")
    if "FETCH_SIZE" in got:
        lines.append(f"l2_miss_read_bytes_per_launch {got['FETCH_SIZE'] * 1024 * 2:.0f}   (FETCH_SIZE KB x 2: gfx950 counts half)")
    if "WRITE_SIZE" in got:
        lines.append(f"write_bytes_per_launch {got['WRITE_SIZE'] * 1024:.0f}")
    out = os.path.join(ROOT, "gpurun_out", f"{args.tag}_d1_pmc.txt")
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
