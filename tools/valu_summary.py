#!/usr/bin/env python
"""Turn the counter CSV of tools/pmc_valu.sh into profiles-style JSON (stamped with the library / source hashes):
per kernel SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_* averaged per launch, and the number of
launches per stage (the workload runs 2 RK2 cycles = 4 stages).  usage: valu_summary.py <dir> <tag> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    d, tag, out = sys.argv[1:4]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    import bench
    kernels = {}
    for k, c in acc.items():
        if not k.startswith("akmi::k_") or "init" in k or "calib" in k or k.startswith("akmi::k_newdt"):
            continue
        n = len(c["SQ_INSTS_VALU"])
        kernels[k] = {"launches": n, "launches_per_stage": n/4.0,
                      "insts_valu_per_launch": sum(c["SQ_INSTS_VALU"])/n,
                      "active_inst_valu_per_launch": sum(c["SQ_ACTIVE_INST_VALU"])/n,
                      "wave_cycles_per_launch": sum(c["SQ_WAVE_CYCLES"])/n,
                      "wait_inst_any_per_launch": sum(c["SQ_WAIT_INST_ANY"])/n,
                      "wait_any_per_launch": sum(c["SQ_WAIT_ANY"])/n,
                      "grbm_gui_active_per_launch": sum(c["GRBM_GUI_ACTIVE"])/n}
    json.dump({"tag": tag, "lib_sha16": bench.lib_sha16(), "src_sha16": bench.src_sha16(),
               "workload": "tools/pmc_workload.py: 2 RK2 cycles of orszag_tang 256^3 (4 stages)",
               "units": "SQ_* in wave-instructions / quad-cycles summed over the chip, GRBM_GUI_ACTIVE summed over 8 XCDs",
               "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
