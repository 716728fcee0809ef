"""Issue-side counters of the shipped forward transform (K1) at BASELINE config 2 -> profiles/<tag>_k1_pmc.txt.

Separate `rocprofv3 --pmc` passes (a pass holds a handful of SQ counters; no tracing options beside
--pmc: the pool refuses that combination) over `python3 bench.py --steps 5 --warmup 1 --lean`, the
program itself right behind `--`.  Per pass the means over the second half of K1's dispatches, then the
derived figures the roofline discussion needs:
    clock held            GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
    VALU busy             SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024 SIMDs)   (SQ_ACTIVE_* count quad-cycles summed over SIMDs)
    packed-math share     the 2 x 16 x 2048 x (8192 x 1024 / 64 / 32) v_pk_mul / v_pk_add the arithmetic needs / SQ_INSTS_VALU
    wave parked / stalled SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES
Usage (from the repo root, inside gpurun):  python3 tools/pmc_k1.py [--tag r03]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_traffic import K1, pick, run_pass  # noqa: E402

PASSES = [
    ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAVES"],
    ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"],
    ["SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS"],
    ["SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD"],
    ["SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VMEM_WR", "SQ_WAIT_INST_ANY"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--kernel-ms", type=float, default=None, help="K1 launch time of an un-profiled run (bench line), for the clock")
    args = ap.parse_args()
    scratch = os.path.join(ROOT, "gpurun_out", f"{args.tag}_k1_pmc")
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--lean"]
    got = {}
    lines = [f"# rocprofv3 --pmc ({len(PASSES)} separate passes) -- python3 bench.py --steps 5 --warmup 1 --lean",
             f"# kernel {K1} (the shipped instance for stereo), 8192 rows x 1024 coefficients (BASELINE config 2, chord input); second half of the dispatches",
             "# SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over all SIMDs; GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES in cycles summed over 8 XCDs / 32 SEs",
             "# pass counter n mean"]
    for i, counters in enumerate(PASSES, 1):
        try:
            acc = run_pass(counters, bench, os.path.join(scratch, f"pass{i}"))
        except SystemExit as e:
            lines.append(f"{i} FAILED {counters}: {e}")
            continue
        for c in counters:
            v, n = pick(acc, K1, c)
            if v is not None:
                got.setdefault(c, v)
                lines.append(f"{i} {c} {n} {v:.0f}")
    d = lambda k: got.get(k)
    lines.append("# derived")
    if d("GRBM_GUI_ACTIVE"):
        cyc = d("GRBM_GUI_ACTIVE") / 8.0
        lines.append(f"cycles_per_launch {cyc:.0f}")
        if args.kernel_ms:
            lines.append(f"clock_ghz_held {cyc / (args.kernel_ms * 1e6):.3f}   (cycles / {args.kernel_ms} ms of the un-profiled bench line)")
        if d("SQ_ACTIVE_INST_VALU"):
            lines.append(f"valu_busy_frac_of_simd_cycles {d('SQ_ACTIVE_INST_VALU') * 4.0 / (cyc * 1024.0):.4f}")
        need = 2 * 2048 * 16 * (8192 * 1024 // 64 // 32)  # packed mul + add wave-instructions the arithmetic needs
        if d("SQ_INSTS_VALU"):
            lines.append(f"needed_packed_math_wave_insts {need}  share_of_SQ_INSTS_VALU {need / d('SQ_INSTS_VALU'):.4f}")
            lines.append(f"issue_bound_cycles_per_launch {d('SQ_INSTS_VALU') * 4.0 / 1024.0:.0f}  (every VALU wave-instruction holds its SIMD 4 cycles)")
    if d("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if d(k):
                lines.append(f"{k}_over_wave_cycles {d(k) / d('SQ_WAVE_CYCLES'):.4f}")
    if d("SQ_LDS_IDX_ACTIVE") and d("GRBM_GUI_ACTIVE"):
        lines.append(f"lds_array_busy_frac {d('SQ_LDS_IDX_ACTIVE') / (d('GRBM_GUI_ACTIVE') / 8.0 * 256.0):.4f}  (LDS-array cycles over cycles x 256 CUs)")
    out = os.path.join(ROOT, "gpurun_out", f"{args.tag}_k1_pmc.txt")
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
