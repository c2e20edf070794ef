#!/usr/bin/env python3
"""tools/microbench_fetch_shape.hip under `rocprofv3 --pmc FETCH_SIZE` -> profiles/<tag>_fetch_shapes.json: per load shape
the bytes the kernel requested (its own stdout), the counter's raw reading (median over the three launches, KiB -> bytes) and
their ratio -- the factor a FETCH_SIZE reading of that shape is multiplied by to become bytes.

    python tools/summarize_fetch_shapes.py <rocprof dir> <stdout of the microbenchmark> <out.json>
"""
import collections
import csv
import glob
import json
import os
import re
import sys

prof_dir, log, out_path = sys.argv[1:4]
req = {}
for line in open(log):
    m = re.match(r"(shape_\w+)\s+requested_bytes (\d+)\s+ms ([\d.]+)\s+GB/s ([\d.]+)\s+\| (.*)", line)
    if m:
        req[m.group(1)] = {"requested_bytes": int(m.group(2)), "ms": float(m.group(3)), "requested_gb_per_s": float(m.group(4)),
                           "shape": m.group(5).strip()}
vals = collections.defaultdict(list)
path = glob.glob(os.path.join(prof_dir, "**", "*_counter_collection.csv"), recursive=True)[0]
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] == "FETCH_SIZE":
        vals[r["Kernel_Name"].split("(")[0].strip()].append(float(r["Counter_Value"]) * 1024.0)
out = {"source": "tools/microbench_fetch_shape.hip under rocprofv3 --kernel-trace --pmc FETCH_SIZE (2 GiB buffer, every byte "
                 "requested once per kernel; median of three launches)",
       "unit": "bytes", "shapes": {}}
for name, r in req.items():
    v = sorted(vals.get(name, []))
    if not v:
        continue
    raw = v[len(v) // 2]
    out["shapes"][name] = dict(r, fetch_raw=raw, requested_over_raw=r["requested_bytes"] / raw, launches=len(v))
json.dump(out, open(out_path, "w"), indent=1)
for name, r in out["shapes"].items():
    print("%-18s requested %6.3f GB  FETCH_SIZE %6.3f GB  requested/raw %.3f  %.1f GB/s  %s" %
          (name, r["requested_bytes"] / 1e9, r["fetch_raw"] / 1e9, r["requested_over_raw"], r["requested_gb_per_s"], r["shape"]))
