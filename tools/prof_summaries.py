#!/usr/bin/env python3
"""Condenses rocprofv3 csv output into the small summaries kept under profiles/ (tools/gpu_visit.sh).
  prof_summaries.py trace <kernel_trace.csv> <out.csv>   this library's dispatches with their durations
  prof_summaries.py pmc   <counter_collection.csv> <out.csv>   per (kernel, counter): dispatches and sum"""
import collections
import csv
import sys


def trace(src, dst):
    rows = [r for r in csv.DictReader(open(src)) if "edlib_amd" in r["Kernel_Name"]]
    keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X", "LDS_Block_Size",
            "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size"]
    keep = [k for k in keep if rows and k in rows[0]]
    w = csv.writer(open(dst, "w"))
    w.writerow(keep + ["Duration_ms"])
    for r in rows:
        w.writerow([r[k] for k in keep] + ["%.4f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)])


def pmc(src, dst):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(src)):
        a = acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    w = csv.writer(open(dst, "w"))
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum"])
    for (k, c), (n, v) in acc.items():
        w.writerow([k, c, n, "%.6g" % v])
        if "edlib_amd" in k:
            print("  %-70s %-22s n=%d sum=%.6g" % (k[:70], c, n, v))


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
