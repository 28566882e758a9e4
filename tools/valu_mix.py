#!/usr/bin/env python3
"""Instruction mix of the main filter kernel, for the mix-weighted VALU ceiling (`roofline.valu_quoted` of bench.py).

tools/valu_peak.hip measures how many cycles a SIMD needs per wave64 instruction for every VALU opcode the kernel uses
(profiles/r03_valu_peak.json): two classes, ~2.2 cycles ("full rate": plain and / or / xor / add / sub / mov ...) and ~4.1
("half rate": shifts, three-operand ops, packed 16-bit ops, compares ...).  The kernel's ceiling in wave-instructions/s is
1024 SIMDs x clock / (mean cycles per instruction of ITS mix).  The mix has to be the DYNAMIC one; there is no counter for it,
so it is estimated from the disassembly: every VALU instruction is weighted by W^(loop depth), loops found from backward
branches of the kernel's assembly.  W is not known either (trip counts are data dependent: 8 characters per verification
run, up to 16 entries per bucket, 2 - 9 placement passes per family): the estimate is given for W = 4, 8 and 16 and the spread is
the error bar.  Usage:

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o /tmp/filter_e2.s sctagger_amd/csrc/filter_e2.hip
    python tools/valu_mix.py /tmp/filter_e2.s profiles/r03_valu_peak.json > profiles/r03_valu_mix.json
"""
import collections
import json
import re
import sys

MAX_DEPTH = 4   # family passes > bucket entries / batches > automaton steps > (inner bookkeeping); deeper nesting is layout, not loops
KERNEL = "_Z8k_filterILi2EjLi0ELi6ELi6ELi64ELb0ELb1ElLb0EEv10FilterArgsIT7_E"   # k_filter<2, unsigned int, 0, 6, 6, 64, false, true, long, false>


def kernel_lines(path, sym):
    out, on = [], False
    for line in open(path):
        if line.startswith(sym + ":"):
            on = True
            continue
        if on:
            if line.startswith(".Lfunc_end"):
                break
            out.append(line.rstrip("\n"))
    if not out:
        raise SystemExit("kernel %s not found in %s" % (sym, path))
    return out


def base_op(mn):
    """v_lshrrev_b32_e32 -> v_lshrrev_b32; sdwa / dpp forms keep their suffix (they issue like the VOP3 class)."""
    return re.sub(r"_(e32|e64)$", "", mn)


def main():
    asm, rates_path = sys.argv[1], sys.argv[2]
    rates = {}
    for r in json.load(open(rates_path))["rows"]:
        if r["waves_per_simd"] == 8:     # cycles per wave-instruction per SIMD = 1024 SIMDs x 2.4 GHz / chip rate
            rates[r["op"]] = 1024 * 2.4e9 / r["wave_insts_per_s_chip"]
    lines = kernel_lines(asm, KERNEL)
    # label positions and backward branches -> loop intervals [target, branch]
    label_at = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    latch = {}      # loop header -> its last backward branch (several latches of one loop are one loop)
    for i, l in enumerate(lines):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            latch[label_at[m.group(1)]] = max(latch.get(label_at[m.group(1)], 0), i)
    loops = sorted(latch.items())
    depth = [0] * len(lines)
    for a, b in loops:
        for i in range(a, b + 1):
            depth[i] = min(depth[i] + 1, MAX_DEPTH)
    ops = []
    for i, l in enumerate(lines):
        m = re.match(r"^\s+(v_\w+)", l)
        if m:
            ops.append((base_op(m.group(1)), depth[i]))
    static = collections.Counter(o for o, _ in ops)

    def cls(op):
        """cycles per instruction: measured if the opcode was measured, else by family"""
        if op in rates:     # (v_cndmask_b32 measured 22.9 cycles in the harness's back-to-back VCC form: priced as half rate)
            return (rates[op] if rates[op] < 6.0 else 4.2), True
        for k, v in rates.items():          # same mnemonic stem (v_cmp_gt_u32 for every v_cmp_*, v_lshrrev_b32 for its sdwa form ...)
            if op.split("_")[1] == k.split("_")[1] and ("sdwa" in op) == ("sdwa" in k):
                return v, False
        if "sdwa" in op or "dpp" in op:
            return rates.get("v_lshrrev_b32_sdwa", 4.2), False
        return 4.2, False                   # unmeasured: priced as half rate (pessimistic for the ceiling = optimistic for nobody)

    out = {"kernel": "k_filter<2, unsigned int, 0, 6, 6, 64, false, true, long, false>", "static_valu_instructions": len(ops),
           "loops_found": len(loops), "max_loop_depth": max(depth) if depth else 0, "by_weight": {}}
    measured_share = sum(c for o, c in static.items() if cls(o)[1]) / float(len(ops))
    out["static_share_of_instructions_whose_opcode_was_measured"] = round(measured_share, 3)
    for W in (1, 4, 8, 16):
        tot = cyc = full = 0.0
        for o, d in ops:
            w = float(W) ** d
            c, _ = cls(o)
            tot += w
            cyc += w * c
            full += w if c < 3.0 else 0.0
        out["by_weight"][str(W)] = {"full_rate_share": round(full / tot, 4), "mean_cycles_per_instruction": round(cyc / tot, 4)}
    mid = out["by_weight"]["8"]
    out["valu_full_rate_share"] = mid["full_rate_share"]
    out["mean_cycles_per_instruction"] = mid["mean_cycles_per_instruction"]
    out["mean_cycles_range"] = [min(v["mean_cycles_per_instruction"] for k, v in out["by_weight"].items() if k != "1"),
                                max(v["mean_cycles_per_instruction"] for k, v in out["by_weight"].items() if k != "1")]
    out["top_static_opcodes"] = [[o, c, cls(o)[0]] for o, c in static.most_common(25)]
    out["method"] = ("static disassembly of the kernel, every VALU instruction weighted by W^(loop depth) (loops = backward branches), W = 8 quoted, "
                     "4 and 16 as the error bar; cycles per opcode from tools/valu_peak.hip at 8 waves per SIMD")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
