#!/usr/bin/env python3
"""Per-kernel medians of the SQ counter passes of tools/pmc_sq.sh, keyed on kernel AND grid size (a template may run
at several batch sizes in one process).

    python tools/summarize_sq.py gpurun_out/<tag>_pmc1 gpurun_out/<tag>_pmc2 gpurun_out/<tag>_pmc3 > profiles/<tag>_sq.json
"""
import collections
import csv
import glob
import json
import os
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "amv::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            key = "%s grid=%s wg=%s" % (name, r["Grid_Size"], r["Workgroup_Size"])
            vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(vals.items()):
    m = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    if m.get("SQ_WAVES"):
        m["valu_per_wave"] = m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"]
        m["lds_per_wave"] = m.get("SQ_INSTS_LDS", 0) / m["SQ_WAVES"]
    if m.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY"):
            if c in m:
                m[c + "/WAVE_CYCLES"] = m[c] / m["SQ_WAVE_CYCLES"]
    if m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
    out[k] = m
json.dump(out, sys.stdout, indent=1)
