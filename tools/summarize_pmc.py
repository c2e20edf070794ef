#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of bench.py into profiles/<tag>_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01

Counters are per dispatch, in KiB.  On gfx950 FETCH_SIZE reads half the bytes of wide coalesced
streaming reads (MI355X_MICROARCH.md, section HBM), so the corrected figure doubles it; WRITE_SIZE
is taken as is.  bench.py attaches the corrected per-launch bytes of the dominant kernel as
roofline.traffic when the workload matches the one profiled."""
import collections
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, tag = sys.argv[1:4]


def collect(d, counter):
    path = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    vals = collections.defaultdict(list)
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "amv::" not in r["Kernel_Name"]:
            continue
        # one row per kernel AND launch shape: a template may run at several batch sizes in one process (the
        # bench's untimed 10 000-frame extra), and a median over all of them would describe none
        name = "%s grid=%s" % (r["Kernel_Name"].split("(")[0].replace("void ", "").strip(), r["Grid_Size"])
        vals[name].append(float(r["Counter_Value"]))
        rows.append((name, r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Counter_Value"]))
    with open("%s_pmc_%s.csv" % (tag, counter.lower()), "w") as f:
        f.write("kernel,grid,workgroup,lds,vgpr,%s_KiB\n" % counter)
        for row in rows:
            f.write(",".join(map(str, row)) + "\n")
    return {k: sorted(v)[len(v) // 2] for k, v in vals.items()}


fetch = collect(fetch_dir, "FETCH_SIZE")
write = collect(write_dir, "WRITE_SIZE")
import subprocess
try:
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except OSError:
    head = ""
out = {"unit": "bytes per launch (median over the launches of one grid size)", "fetch_correction": "FETCH_SIZE x2 (gfx950 wide-read undercount)",
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (tools/profile_round.sh)", "head": head, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fr, wr = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    out["kernels"][k] = {"fetch_raw": fr, "write": wr, "hbm_corrected": 2 * fr + wr}
json.dump(out, open(tag + "_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
