#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of bench.py into profiles/<tag>_traffic.json.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01

Counters are per dispatch, in KiB.  On gfx950 FETCH_SIZE tallies every fabric read request at 64 bytes
(MI355X_MICROARCH.md, section HBM, states it for 16-byte-per-lane streaming reads); the calibration of round 6
(tools/microbench_fetch_shape.hip, profiles/r06_fetch_shapes.json) shows that EVERY read request of every load shape
these kernels use -- 4 B per lane, 16 B per lane, the ADPCM staging fetch's 64-byte row pieces, the one-lane entropy
kernel's 32-byte requests, the reconstruction's 4-byte records -- is a 128-byte line (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ),
and that >= 98.9 % of the requests of each of the bench's own kernels are: FETCH_SIZE x 2 IS the bytes that crossed the
fabric, for every kernel.  Each kernel's row names the microbenchmark shape its loads have; WRITE_SIZE is taken as is.
bench.py attaches the corrected per-launch bytes of the dominant kernel as roofline.traffic when the workload matches
the one profiled."""
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
# which of the calibrated load shapes (profiles/r06_fetch_shapes.json) a kernel's dominant loads have
SHAPE_OF = (("amv_adpcm_guess", "shape_c_paced"), ("amv_adpcm_sweep", "shape_c_paced"), ("amv_adpcm_decode", "shape_b"),
            ("amv_adpcm", "shape_a"), ("amv_huffman_fast", "shape_d_paced"), ("amv_huffman_sync", "shape_d_paced"),
            ("amv_huffman_kernel", "shape_d"), ("amv_unstuff", "shape_b"), ("amv_reconstruct", "shape_e"),
            ("amv_encode_frame", "shape_b"), ("amv_", "shape_a"))
try:
    CAL = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r06_fetch_shapes.json")))
except (OSError, ValueError):
    CAL = {"shapes": {}, "bench_kernels": {"kernels": {}}}


def correction(kernel):
    """{factor, shape, measured}: the factor is 2 for every shape (all fabric reads are 128-byte lines tallied at 64); the
    shape's row says how many times its lines cross the fabric per requested byte"""
    short = kernel.split("::")[-1]
    shape = next(sh for key, sh in SHAPE_OF if short.startswith(key))
    row = CAL["shapes"].get(shape, {})
    own = next((v for k, v in CAL["bench_kernels"]["kernels"].items() if k.split(" grid=")[0].split("<")[0] == kernel.split(" grid=")[0].split("<")[0]), None)
    return {"factor": 2.0, "microbenchmark_row": shape, "fabric_bytes_over_fetch_raw": row.get("fabric_bytes_over_fetch_raw"),
            "fabric_bytes_over_requested_for_this_shape": row.get("fabric_bytes_over_requested"),
            "share_of_128B_requests_in_this_kernel": own["share_128B"] if own else None}


out = {"unit": "bytes per launch (median over the launches of one grid size)",
       "fetch_correction": "FETCH_SIZE x2 for every kernel: every fabric read request is a 128-byte line tallied at 64 bytes "
                           "(calibrated per load shape, profiles/r06_fetch_shapes.json; per kernel below)",
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (tools/profile_round.sh)", "head": head, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fr, wr = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    out["kernels"][k] = {"fetch_raw": fr, "write": wr, "hbm_corrected": 2 * fr + wr, "fetch_correction": correction(k)}
json.dump(out, open(tag + "_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
