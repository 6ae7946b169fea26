#!/usr/bin/env python3
"""ISA audit of the stage kernels: per-kernel static instruction mix from the hipcc --save-temps
assembly (opcode classes that matter for the fp64 issue floor), VGPRs, scratch, occupancy.

usage: tools/isa_audit.py file.s [kernel-name-substring ...]
Static counts (every instruction once, loops not weighted); the dynamic counts are the SQ
counters in profiles/*valu_counters.txt.
"""
import collections
import re
import subprocess
import sys

CLASSES = [
    ("div_scale", r"^v_div_scale_f64"), ("div_fmas", r"^v_div_fmas_f64"), ("div_fixup", r"^v_div_fixup_f64"),
    ("rcp_f64", r"^v_rcp_f64"), ("rsq_f64", r"^v_rsq_f64"), ("sqrt_f64", r"^v_sqrt_f64"),
    ("fma_f64", r"^v_fma_f64"), ("mul_f64", r"^v_mul_f64"), ("add_f64", r"^v_add_f64"),
    ("minmax_f64", r"^v_(min|max)_f64"), ("cmp_f64", r"^v_cmpx?_\w+_f64"), ("ldexp/frexp/class", r"^v_(ldexp|frexp|cmp_class|trig)"),
    ("cndmask_b32", r"^v_cndmask_b32"), ("mov_b32", r"^v_mov_b32"), ("accvgpr", r"^v_accvgpr"),
    ("int32 valu", r"^v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|bfi|min_[iu]|max_[iu]|cmp\w*_[iu]\d|readlane|readfirstlane|writelane|perm|alignbit|lshlrev|lshrrev|ashrrev|add_co|addc_co|add3|lshl_add|mbcnt)"),
    ("dpp/permute", r"(ds_bpermute|ds_permute|_dpp|v_permlane|ds_swizzle)"),
    ("ds_read", r"^ds_read"), ("ds_write", r"^ds_write"),
    ("global_load", r"^(global|buffer|flat)_load"), ("global_store", r"^(global|buffer|flat)_store"), ("atomic", r"atomic"),
    ("scratch", r"^scratch_"), ("s_waitcnt", r"^s_waitcnt"), ("s_barrier", r"^s_barrier"), ("s_cbranch", r"^s_cbranch"),
]


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


def main():
    path = sys.argv[1]
    want = sys.argv[2:]
    kern = None
    counts = {}
    meta = collections.defaultdict(dict)
    order = []
    for line in open(path, errors="replace"):
        s = line.strip()
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", s)
        if m and not s.startswith(".L"):
            kern = m.group(1)
            counts[kern] = collections.Counter()
            order.append(kern)
            continue
        if kern is None:
            continue
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            pass
        m = re.match(r"^;\s*(NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize|codeLenInByte):\s*(\d+)", s)
        if m:
            meta[kern][m.group(1)] = int(m.group(2))
            continue
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        op = s.split()[0]
        c = counts[kern]
        if op.startswith("v_"):
            c["VALU total"] += 1
        elif op.startswith("s_"):
            c["SALU/ctl total"] += 1
        hit = False
        for name, pat in CLASSES:
            if re.search(pat, op):
                c[name] += 1
                hit = True
                break
        if not hit and op.startswith("v_"):
            c["other valu"] += 1
            c["other:" + op] += 1
    for k in order:
        d = demangle(k)
        short = re.sub(r"\(.*", "", d).replace("void ", "")
        if want and not any(w.replace(" ", "") in short.replace(" ", "") for w in want):
            continue
        c = counts[k]
        if c["VALU total"] == 0:
            continue
        mt = meta[k]
        print("== %s" % short[:110])
        print("   vgpr %s agpr %s scratch %s B  sgpr %s  lds %s B  occupancy %s  code %s B" % (
            mt.get("NumVgprs"), mt.get("NumAgprs"), mt.get("ScratchSize"), mt.get("NumSgprs"), mt.get("LDSByteSize"),
            mt.get("Occupancy"), mt.get("codeLenInByte")))
        names = ["VALU total"] + [n for n, _ in CLASSES] + ["other valu", "SALU/ctl total"]
        print("   " + "  ".join("%s=%d" % (n, c[n]) for n in names if c[n]))
        oth = sorted(((n[6:], v) for n, v in c.items() if n.startswith("other:")), key=lambda t: -t[1])[:12]
        if oth:
            print("   other: " + " ".join("%s=%d" % t for t in oth))


if __name__ == "__main__":
    main()
