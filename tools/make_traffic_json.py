#!/usr/bin/env python3
"""<dir>/<R>_pmc_fetch_c{2,4,5}.csv + <R>_pmc_write_c{2,4,5}.csv (tools/gpu_visit.sh traffic) -> profiles/hbm_traffic.json,
which bench.py copies into `roofline.traffic` TOGETHER WITH the commit the counters were taken at (a measurement of
that commit, not of the timed run).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE on gfx950 reports
half the bytes of a wide coalesced read (MI355X_MICROARCH.md §HBM; confirmed here on pack_target_2bit_kernel:
2458 KB reported for its 5,000,000-byte read).  One bench step per pass (--steps 1 --warmup 0).
Round 5: the FETCH_SIZE pass (three TCC slots) did not come back within its limit on this pool once the ring32 kernels
were in the step (twice, 90 s and 240 s; the WRITE_SIZE pass and every test are unaffected), so the fetch pass takes the
counter FETCH_SIZE is derived from, TCC_EA0_RDREQ_sum (one slot): FETCH_SIZE [KB] = RDREQ x 64 / 1024 (the guide's formula),
i.e. the same number by another route; a file whose Counter_Name is TCC_EA0_RDREQ_sum is converted here."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edlib_amd.parallel import sources_sha           # one digest over the product's sources + bench.py: what the counters were taken on
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
R = sys.argv[2] if len(sys.argv) > 2 else "r03"
commit = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else subprocess.run(
    ["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
dst = os.path.join(ROOT, "profiles", "hbm_traffic.json")
try:
    out = json.load(open(dst))
except Exception:
    out = {"configs": {}}
out["method"] = ("rocprofv3 --pmc FETCH_SIZE (round 5: TCC_EA0_RDREQ_sum x 64 B, the counter it derives from) / --pmc WRITE_SIZE in separate passes over `bench.py --config N --steps 1 --warmup 0 "
                 "--no-cpu-baseline --no-e2e --no-secondary` (tools/gpu_visit.sh traffic); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024")
for cfg, units in (("2", 1000000), ("4", 100000), ("5", 10000)):
    items, ok = {}, True
    for ctr, fn in (("fetch_kb", "%s_pmc_fetch_c%s.csv" % (R, cfg)), ("write_kb", "%s_pmc_write_c%s.csv" % (R, cfg))):
        path = os.path.join(src, fn)
        if not os.path.exists(path):
            ok = False
            break
        for r in csv.DictReader(open(path)):
            if "edlib_amd" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("edlib_amd::", "")
            items.setdefault(k, {"fetch_kb": 0.0, "write_kb": 0.0, "dispatches": 0})
            v = float(r["Sum"])
            if r.get("Counter_Name", "").startswith("TCC_EA0_RDREQ"):
                v = v * 64.0 / 1024.0                        # requests -> the KB FETCH_SIZE would report
            items[k][ctr] += v
            items[k]["dispatches"] = int(r["Dispatches"])
    if not ok or not items:
        continue
    total = sum(2 * v["fetch_kb"] + v["write_kb"] for v in items.values()) * 1024
    for v in items.values():
        v["bytes"] = int((2 * v["fetch_kb"] + v["write_kb"]) * 1024)
    out["configs"][cfg] = {"units": units, "commit": commit, "sources_sha": sources_sha(ROOT), "bytes_per_step": int(total), "per_kernel": items}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v["bytes_per_step"] for k, v in out["configs"].items()}))
