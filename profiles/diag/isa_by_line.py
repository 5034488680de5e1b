#!/usr/bin/env python3
"""Attributes the instructions of one kernel in a hipcc -S -gline-tables-only listing to source lines.

    python profiles/diag/isa_by_line.py pt_kernels.s <kernel-symbol-substring> [file:first-last ...]

Prints, per requested source range (default: every line), vector / scalar / memory / LDS instruction counts and the
vector instructions by issue class (full rate, half rate, transcendental) as measured by micro/valu_mix.hip.
"""
import collections
import re
import sys

HALF = re.compile(r"^v_(min|max|med3|lshl|add3|lshl_add|lshl_or|or3|and_or|bfe|bfi|bcnt|ffb|mbcnt|cvt|cmp|cmpx|mul_lo|mul_hi|mad_u64|mad_i64|readlane|readfirstlane|writelane|pk_|perm|alignbit|sad|lshr|xad|add_lshl|cndmask_b32_e64|ldexp|frexp|fract|trunc|floor|rndne|ceil|div_|fmac_f64|lshlrev_b64|lshrrev_b64|ashrrev_i64|mov_b64|mul_u32_u24|mad_u32_u24|mul_i32_i24)")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)")


def classify(op, operands):
    if op.startswith("v_"):
        if TRANS.match(op):
            return "v_trans"
        if HALF.match(op) or "dpp" in operands or re.search(r"\bs\d+|\bs\[", operands.split(",", 1)[1] if "," in operands else ""):
            return "v_half"
        return "v_full"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
            return "s_wait"
        if op.startswith(("s_cbranch", "s_branch")):
            return "s_branch"
        return "s_alu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    ranges = []
    for a in sys.argv[3:]:
        f, r = a.split(":")
        lo, hi = r.split("-")
        ranges.append((f, int(lo), int(hi)))
    files = {}
    counts = collections.defaultdict(collections.Counter)
    ops = collections.defaultdict(collections.Counter)
    inside = False
    cur = None
    for line in open(path):
        s = line.strip()
        m = re.match(r"\.file\s+(\d+)\s+(?:\"[^\"]*\"\s+)?\"([^\"]+)\"", s)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
            continue
        if re.match(r"^_Z\S*:", s):
            inside = sym in s.split(":")[0]
            continue
        if not inside:
            continue
        if s.startswith(".Lfunc_end"):
            inside = False
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        operands = parts[1].split(";")[0] if len(parts) > 1 else ""
        c = classify(op, operands)
        counts[cur][c] += 1
        ops[cur][op] += 1
    if not ranges:
        for k in sorted(counts, key=lambda k: (str(k[0]), k[1]) if k else ("", 0)):
            print(k, dict(counts[k]))
        return
    for f, lo, hi in ranges:
        tot = collections.Counter()
        o = collections.Counter()
        for k, c in counts.items():
            if k and k[0] == f and lo <= k[1] <= hi:
                tot.update(c)
                o.update(ops[k])
        v = tot["v_full"] + tot["v_half"] + tot["v_trans"]
        cyc = 2.5 * tot["v_full"] + 4.2 * tot["v_half"] + 8.2 * tot["v_trans"] + 2.2 * (tot["s_alu"] + tot["s_branch"])
        print(f"{f}:{lo}-{hi}: vector {v} (full {tot['v_full']}, half {tot['v_half']}, trans {tot['v_trans']}), scalar {tot['s_alu']} + {tot['s_branch']} branches, "
              f"waits {tot['s_wait']}, vmem {tot['vmem']}, lds {tot['lds']}; weighted cycles {cyc:.0f}")
        print("   ", ", ".join(f"{k} {n}" for k, n in o.most_common(40)))


if __name__ == "__main__":
    main()
