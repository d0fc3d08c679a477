#!/usr/bin/env python3
"""profiles/hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
HBM bytes per step of the scan kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the scan
launches of one step (MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reads exactly half of a wide
coalesced stream -- confirmed here on pack_target_2bit_kernel: 2449 KB reported for a 5,000,000-byte
16-B-per-lane read -- and WRITE_SIZE matched the known 1.25 MB / 24 MB stores of the pack and Peq
kernels; the scan kernel's own loads are 4 B per lane and scalar, so the factor 2 is an upper bound)."""
import csv, json, sys, collections

def total(path, counter, kernel_substr):
    acc = 0.0; n = 0
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
            acc += float(row["Counter_Value"]); n += 1
    return acc, n

fetch_csv, write_csv, steps, reads, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
f, nf = total(fetch_csv, "FETCH_SIZE", "scan_reads")
w, nw = total(write_csv, "WRITE_SIZE", "scan_reads")
per_step = (2 * f + w) * 1024 / steps
json.dump({"bytes_per_launch": int(per_step), "unit": "HBM bytes per step (all scan launches of one step)",
           "fetch_size_kb_per_step": f / steps, "write_size_kb_per_step": w / steps,
           "scan_launches_per_step": nf / steps, "reads": reads,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes on `bench.py --reads %d --steps %d --warmup 0 "
                     "--no-cpu-baseline`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count correction)" % (reads, steps)},
          open(out, "w"), indent=1)
print(open(out).read())
