#!/usr/bin/env python3
"""Static VALU issue-cost estimate of a gfx950 kernel from its assembly (hipcc -S --cuda-device-only).

The per-instruction costs are the ones measured by profiles/micro/valu_rates{,2,3}.hip on an MI355X (cycles per wave64 instruction
per SIMD with four waves resident per SIMD, round 3):

  fast   2.5  v_fma/fmac/fmaak/fmamk/mul/add/sub_f32, v_and/or/xor_b32, v_add/sub_u32, v_lshrrev_b32, v_mov_b32, v_cndmask_b32
              -- only when every source is a VGPR, an inline constant or a literal
  slow   4.2  the same opcodes with an SGPR source, and every other VALU opcode (min/max/med3, cvt, floor/fract/rndne, bfe,
              lshl_add, and_or, add3, perm, alignbit, lshlrev, mul24/mad24, mul_lo/hi, v_cmp, SDWA and DPP forms)
  packed 4.8  v_pk_*_f32 (two results: no cheaper per result than the fast scalar form)
  trans  8.1  v_exp/log/rcp/rsq/sqrt/sin/cos_f32

Usage: isa_cost.py file.s kernel_name_substring [--blocks]
Prints the whole-function count per class and, with --blocks, one line per basic block so the hot loop can be read off.
"""
import re
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_mov_b32",
        "v_cndmask_b32"}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
COST = {"fast": 2.5, "slow": 4.2, "packed": 4.8, "trans": 8.1}
SGPR = re.compile(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|ttmp\d+)(?![a-z0-9_])")


def classify(op, operands):
    base = op
    for suffix in ("_e32", "_e64", "_sdwa", "_dpp", "_e64_dpp"):
        if base.endswith(suffix):
            base = base[: -len(suffix)]
    if base.startswith("v_pk_"):
        return "packed"
    if base in TRANS:
        return "trans"
    if op.endswith("_sdwa") or op.endswith("_dpp") or "row_" in operands or "wave_" in operands or "quad_perm" in operands:
        return "slow"
    if base in FAST:
        srcs = operands.split(",")[1:]
        if base == "v_cndmask_b32":
            srcs = srcs[:2]  # the mask is an SGPR pair by construction; measured at the fast rate after a v_cmp
        if any(SGPR.search(s) for s in srcs):
            return "slow"
        return "fast"
    return "slow"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    per_block = "--blocks" in sys.argv
    inside = False
    blocks = []
    cur = None
    for line in open(path):
        s = line.split(";")[0].strip()
        if not inside:
            if s.endswith(":") and needle in s and not s.startswith("."):
                inside = True
                cur = {"name": s[:-1][:60], "fast": 0, "slow": 0, "packed": 0, "trans": 0, "salu": 0, "lds": 0, "vmem": 0, "slow_ops": {}}
                blocks.append(cur)
            continue
        if s.startswith(".Lfunc_end"):
            break
        if s.startswith(".LBB") and s.split()[0].endswith(":"):
            cur = {"name": s.split()[0][:-1], "fast": 0, "slow": 0, "packed": 0, "trans": 0, "salu": 0, "lds": 0, "vmem": 0, "slow_ops": {}}
            blocks.append(cur)
            continue
        if not s or s.startswith("."):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        operands = parts[1] if len(parts) > 1 else ""
        if op.startswith("v_"):
            c = classify(op, operands)
            cur[c] += 1
            if c == "slow":
                key = op + ("(s)" if op.split("_e")[0] in FAST or op in FAST else "")
                cur["slow_ops"][key] = cur["slow_ops"].get(key, 0) + 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            cur["vmem"] += 1
    if not blocks:
        sys.exit("kernel not found")
    total = {k: sum(b[k] for b in blocks) for k in ("fast", "slow", "packed", "trans", "salu", "lds", "vmem")}
    slow_ops = {}
    for b in blocks:
        for k, v in b["slow_ops"].items():
            slow_ops[k] = slow_ops.get(k, 0) + v

    def cycles(b):
        return sum(b[k] * COST[k] for k in COST)

    if per_block:
        for b in blocks:
            n = b["fast"] + b["slow"] + b["packed"] + b["trans"]
            if n + b["lds"] + b["vmem"] < 8:
                continue
            print(f'{b["name"]:28s} valu {n:5d} (fast {b["fast"]:4d} slow {b["slow"]:4d} pk {b["packed"]:4d} trans {b["trans"]:3d}) '
                  f'salu {b["salu"]:4d} lds {b["lds"]:4d} vmem {b["vmem"]:3d}  issue cycles {cycles(b):7.0f}')
    n = total["fast"] + total["slow"] + total["packed"] + total["trans"]
    print(f'{blocks[0]["name"]}: valu {n} (fast {total["fast"]} slow {total["slow"]} packed {total["packed"]} trans {total["trans"]}) '
          f'salu {total["salu"]} lds {total["lds"]} vmem {total["vmem"]}  issue cycles {cycles(total):.0f}')
    top = sorted(slow_ops.items(), key=lambda kv: -kv[1])[:24]
    print("slow-class opcodes:", ", ".join(f"{k} {v}" for k, v in top))


if __name__ == "__main__":
    main()
