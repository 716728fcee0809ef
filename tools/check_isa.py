"""Build-time guard of the numerics contract: the transform kernels must not contain a fused
multiply-add (v_fma*, v_fmac*, v_pk_fma*, v_mad*, v_mac*, v_dot*) or an accumulating MFMA — the
reference accumulates with a separately rounded multiply and add.

  k_mdct_fwd_sched / k_mdct_fwd_dma / k_mdct_fwd_st / k_mdct_fwd_small / k_imdct_apply  strict: no fused op at all (no division or sqrt inside)
  k_imdct_rows / k_imdct_plan   fused ops allowed only inside hipcc's correctly-rounded f32 division expansion
                (v_div_scale ... v_div_fixup), which the raw-frame path `i16 / 32767.0` needs
The quantiser / decision / overlap-add kernels are not scanned: their IEEE divide, sqrt and
64-bit index division expand to FMA-based sequences by design; their arithmetic is pinned by the
bit-exact parity tests instead."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gapless-lossy-codec_amd", "csrc")
ISA = os.path.join(ROOT, "build", "isa", "glc_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
FORBIDDEN = re.compile(r"^\s+(v_fma\w*|v_fmac\w*|v_pk_fma\w*|v_mad_\w*f32|v_mac\w*|v_dot\w*|v_mfma\w*)\b")
KERNELS = ("k_mdct_fwd_sched", "k_mdct_fwd_dma", "k_mdct_fwd_st", "k_mdct_fwd_small", "k_imdct_rows", "k_imdct_plan", "k_imdct_apply")
DIV_WINDOW = {"k_imdct_rows", "k_imdct_plan"}


def main() -> int:
    subprocess.check_call(["make", "-C", CSRC, "-s", "isa"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cur, bad, seen, in_div = None, [], set(), False
    first_ds = {}   # kernel -> True once its first ds_read has been seen
    for line in open(ISA):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = next((k for k in KERNELS if k in m.group(1)), None)
            in_div = False
            first_ds = {}
            if cur:
                seen.add(cur)
        elif cur and cur in DIV_WINDOW and "v_div_scale_f32" in line:
            in_div = True
        elif cur and "v_div_fixup_f32" in line:
            in_div = False
        elif cur and FORBIDDEN.match(line) and not in_div:
            bad.append((cur, line.strip()))
        # k_mdct_fwd_small waits for its LDS operands with COUNTED lgkmcnt (LDS returns in order): a scalar
        # load in flight at the same time would make those counts meaningless (SMEM returns out of order)
        if cur == "k_mdct_fwd_small":
            if re.match(r"^\s+ds_read", line):
                first_ds[cur] = True
            elif first_ds.get(cur) and re.match(r"^\s+(s_load|s_buffer_load)", line):
                bad.append((cur, "scalar load beside counted LDS waits: " + line.strip()))
        asm_async = cur and (cur.startswith("k_mdct_fwd") or cur == "k_imdct_apply")  # loads issued and waited for in separate asm statements
        if asm_async and re.match(r"^\s+(scratch_|buffer_store.*offen.*s\[0:3\]|buffer_load.*off.*s\[0:3\])", line):
            bad.append((cur, "register spill: " + line.strip()))
        # k_mdct_fwd_st / k_imdct_apply keep table values / records in SGPRs that an asm statement loads and a LATER
        # one waits for: a spill into VGPR lanes between the two copies stale values (it happened in a tuning
        # variant: wrong results and a memory fault), besides putting v_writelane / v_readlane into the inner loop
        if asm_async and re.match(r"^\s+(v_writelane|v_readlane)", line):
            bad.append((cur, "scalar register spill: " + line.strip()))
        if "s_endpgm" in line:
            cur = None
    missing = set(KERNELS) - seen
    if missing:
        print("check_isa: kernels not found in ISA:", sorted(missing))
        return 1
    if bad:
        for k, l in bad:
            print(f"check_isa: fused op in {k}: {l}")
        return 1
    print(f"check_isa: {len(seen)} kernels clean (no fused multiply-add)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
