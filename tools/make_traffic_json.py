#!/usr/bin/env python3
"""profiles/<R>_pmc_fetch_c2_1M.csv + <R>_pmc_write_c2_1M.csv (tools/gpu_profiles.sh) -> profiles/hbm_traffic.json,
which bench.py copies into `roofline.traffic` TOGETHER WITH the commit the counters were taken at (it is a
measurement of that commit, not of the timed run).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE on
gfx950 reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md §HBM; confirmed here on
pack_target_2bit_kernel: 2458 KB reported for its 5,000,000-byte read)."""
import csv, json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 "
                 "--no-cpu-baseline --no-e2e` (tools/gpu_profiles.sh); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024", "configs": {}}
for cfg, tag, units in (("2", "c2_1M", 1000000),):
    items = {}
    for ctr, fn in (("fetch_kb", "%s_pmc_fetch_%s.csv" % (R, tag)), ("write_kb", "%s_pmc_write_%s.csv" % (R, tag))):
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", fn))):
            if "edlib_amd" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("edlib_amd::", "")
            items.setdefault(k, {"fetch_kb": 0.0, "write_kb": 0.0, "dispatches": 0})
            items[k][ctr] += float(r["Sum"]); items[k]["dispatches"] = int(r["Dispatches"])
    total = sum(2 * v["fetch_kb"] + v["write_kb"] for v in items.values()) * 1024
    for v in items.values():
        v["bytes"] = int((2 * v["fetch_kb"] + v["write_kb"]) * 1024)
    out["configs"][cfg] = {"units": units, "commit": commit, "bytes_per_step": int(total), "per_kernel": items}
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v["bytes_per_step"] for k, v in out["configs"].items()}))
