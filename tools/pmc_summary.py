"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output per kernel (average per
launch), calibrated on akmi::k_calib_copy whose true traffic is known (512 MiB each way)."""
import csv
import glob
import json
import os
import sys


def load(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            n, s = acc.get(k, (0, 0.0))
            acc[k] = (n + 1, s + float(r["Counter_Value"]))
    return {k: s/n for k, (n, s) in acc.items()}, {k: n for k, (n, s) in acc.items()}


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, nw = load(sys.argv[2], "WRITE_SIZE")
    true_bytes = 64*1024*1024*8.0
    cal = "akmi::k_calib_copy"
    # counters are reported in KiB-like units of 1024 B by the guide; calibrate instead of assuming
    fscale = true_bytes/fetch[cal] if cal in fetch and fetch[cal] > 0 else None
    wscale = true_bytes/write[cal] if cal in write and write[cal] > 0 else None
    out = {"calibration": {"kernel": cal, "true_bytes_each_way": true_bytes,
                           "raw_FETCH_SIZE": fetch.get(cal), "raw_WRITE_SIZE": write.get(cal),
                           "bytes_per_FETCH_unit": fscale, "bytes_per_WRITE_unit": wscale,
                           "note": "guide: nominal unit 1024 B; gfx950 FETCH_SIZE reads 1/2 of a "
                                   "wide coalesced stream -> expect ~2048 B per unit for reads"},
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        rd = fetch.get(k, 0.0)*(fscale or 1024.0)
        wr = write.get(k, 0.0)*(wscale or 1024.0)
        out["kernels"][k] = {"launches": nf.get(k, nw.get(k, 0)), "read_bytes_per_launch": rd,
                             "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    # which build of the library these counters belong to (bench.py refuses another build's numbers)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.environ.get("AKMI_LIB") or os.path.join(root, "athenak_amd", "lib", "libakmi.so")
    out["lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    sys.path.insert(0, root)
    import bench
    out["src_sha16"] = bench.src_sha16()
    if len(sys.argv) > 3:
        out["tag"] = sys.argv[3]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
