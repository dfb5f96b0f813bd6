#!/usr/bin/env python3
"""Mix-weighted VALU issue ceiling of the hot kernels, from their disassembly and the measured issue table.

    python tools/valu_ceiling.py            # writes profiles/valu_ceiling.json

profiles/r04/valu_issue_table.json (tools/exp/valu_probe.hip, measured on MI355X) says what one wave64 instruction of each
opcode costs a SIMD: 4.1 cycles for every packed, three-operand, DPP, compare, convert and 24-bit-multiply instruction and
for anything with an SGPR operand; 2.2 cycles -- with two or more waves on the SIMD, and only in runs of such instructions --
for the two-operand 32-bit logic / shift-right / add / sub / mov, the unpacked 16-bit VOP2 ops, v_fma/mul/add_f32 and
v_bitop3_b32; 8.1 for v_rcp_f32, v_swap_b32, v_permlane32_swap, v_min3_u16 / v_fma_f16.  In a stream that alternates
2-cycle and 4-cycle instructions the 2-cycle ones cost 4 as well (mix_and_pkmin_1to1: 4.0 per instruction).

For each kernel this script recompiles its source with the Makefile's flags and --save-temps, counts the VALU opcodes of the
kernel's body and prices them twice:
  * `ceiling_all4`  -- every instruction at the measured 4-cycle rate (545 G wave-inst/s chip-wide): the rate a kernel reaches
                       when its cheap instructions never pair up;
  * `ceiling_mix`   -- every instruction at its own row of the table (the optimistic bound).
bench.py divides a kernel's SQ_INSTS_VALU / duration by these."""
from __future__ import annotations

import json
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "vision_slam_frontend_amd" / "csrc"
sys.path.insert(0, str(ROOT))
from vision_slam_frontend_amd.buildinfo import kernel_source_hash  # noqa: E402
TABLE = ROOT / "profiles" / "r04" / "valu_issue_table.json"
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-result"]
KERNELS = {  # stage -> (source, substring of the mangled kernel name whose body is counted)
    "fast_score_nms": ("k_fast.hip", "fast_march_resident_kernelILb0ELb1E"),
    "fast_score_nms_grid": ("k_fast.hip", "fast_march_kernelILb0ELb1E"),
    "gauss_blur7": ("k_blur.hip", "blur_mma_kernel"),
    "select_harris_angle": ("k_select.hip", "orb_select_kernelILi256E"),
    "orb_describe": ("k_describe.hip", "orb_orient_describe_kernelILi4E"),
    "pyramid_resize": ("k_pyramid.hip", "resize_strip_kernelILi8E"),
    "hamming_knn2": ("k_match.hip", "knn2_fp4_kernelILb0E"),
}
TWO_CYCLE = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_add_u32",
             "v_sub_u32", "v_subrev_u32", "v_min_u16", "v_max_u16", "v_min_i16", "v_max_i16", "v_add_u16", "v_sub_u16",
             "v_mul_lo_u16", "v_lshlrev_b16", "v_lshrrev_b16", "v_ashrrev_i16", "v_fma_f32", "v_fmac_f32", "v_mul_f32",
             "v_add_f32", "v_sub_f32", "v_min_f16", "v_max_f16", "v_add_f16", "v_mul_f16", "v_bitop3_b32"}
EIGHT_CYCLE = {"v_rcp_f32", "v_swap_b32", "v_permlane32_swap_b32", "v_min3_u16", "v_fma_f16", "v_min3_f16", "v_rsq_f32",
               "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_div_scale_f64",
               "v_mul_lo_u32", "v_mul_hi_u32"}  # (v_mul_lo_u32 measured 4.1; kept at 4 below -- see cost())


def kernel_bodies(asm: str):
    """name -> list of instruction lines of every function in a --save-temps .s file."""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None:
            if re.match(r"^\s+s_endpgm", line):
                cur = None
            elif re.match(r"^\s+[a-z_0-9]+", line) and not line.strip().startswith((";", ".")):
                out[cur].append(line.strip())
    return out


def classify(line: str, cyc2: float, cyc4: float, cyc8: float):
    op = line.split()[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if not base.startswith("v_") or base.startswith(("v_mfma", "v_smfmac", "v_accvgpr")):
        return None
    uses_sgpr = bool(re.search(r"(?<![a-z])(s\d+|s\[\d+:\d+\]|vcc|exec)(?![a-z0-9])", line.split(None, 1)[1] if " " in line else ""))
    is_dpp = op.endswith("_dpp") or bool(re.search(r"row_|wave_|quad_perm|bound_ctrl", line))
    if base in ("v_mul_lo_u32", "v_mul_hi_u32"):
        return base, cyc4
    if base in EIGHT_CYCLE:
        return base, cyc8
    if base in TWO_CYCLE and not uses_sgpr and not is_dpp:
        return base, cyc2
    return base, cyc4


def main():
    table = json.loads(TABLE.read_text())
    ops = {o["op"]: o for o in table["ops"]}
    cyc4 = ops["pk_min_i16"]["w4"]["cycles_per_inst_per_simd"]
    cyc2 = ops["and_b32"]["w4"]["cycles_per_inst_per_simd"]
    cyc8 = ops["rcp_f32"]["w4"]["cycles_per_inst_per_simd"]
    rate4 = ops["pk_min_i16"]["w4"]["g_wave_inst_per_s"]   # measured chip-wide, at the clock the chip holds under that load
    clock_ghz = rate4 * cyc4 / (table["cus"] * 4)
    out = {"table": str(TABLE.relative_to(ROOT)), "source_hash": kernel_source_hash(), "cycles": {"two": cyc2, "four": cyc4, "eight": cyc8},
           "rate_all4_g_wave_inst_per_s": rate4, "clock_ghz_under_valu_load": clock_ghz, "kernels": {}}
    cache = {}
    for stage, (src, needle) in KERNELS.items():
        if src not in cache:
            with tempfile.TemporaryDirectory() as td:
                extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if src == "k_match.hip" else []
                subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "--save-temps", "-c", str(CSRC / src), "-o",
                                       str(Path(td) / "x.o")], cwd=td, stderr=subprocess.DEVNULL)
                s = next(Path(td).glob("*gfx950.s")).read_text()
            cache[src] = kernel_bodies(s)
        bodies = [(n, b) for n, b in cache[src].items() if needle in n]
        if not bodies:
            print("no kernel matching", needle, "in", src, file=sys.stderr)
            continue
        name, body = max(bodies, key=lambda nb: len(nb[1]))
        hist, cycles, n = Counter(), 0.0, 0
        for line in body:
            c = classify(line, cyc2, cyc4, cyc8)
            if c is None:
                continue
            hist[c[0]] += 1
            cycles += c[1]
            n += 1
        two = sum(v for k, v in hist.items() if k in TWO_CYCLE)
        out["kernels"][stage] = {
            "kernel": name, "source": src, "valu_instructions_static": n,
            "two_cycle_class_static": two, "mean_cycles_mix": cycles / max(n, 1),
            "ceiling_all4_g_wave_inst_per_s": rate4,
            "ceiling_mix_g_wave_inst_per_s": rate4 * cyc4 / (cycles / max(n, 1)),
            "top_opcodes": dict(hist.most_common(14)),
        }
    dst = ROOT / "profiles" / "valu_ceiling.json"  # (beside traffic.json: both follow the sources, not a round)
    dst.write_text(json.dumps(out, indent=1) + "\n")
    for k, v in out["kernels"].items():
        print("%-22s %5d VALU (%4d two-cycle class)  mean %.2f cyc  ceiling %.0f .. %.0f G wave-inst/s" % (
            k, v["valu_instructions_static"], v["two_cycle_class_static"], v["mean_cycles_mix"],
            v["ceiling_all4_g_wave_inst_per_s"], v["ceiling_mix_g_wave_inst_per_s"]))


if __name__ == "__main__":
    main()
