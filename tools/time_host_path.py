#!/usr/bin/env python3
"""PCIe-inclusive decode rate: amvhip_decode_batch with host buffers (chunks in, frames out), the figure
DESIGN.md quotes beside the device-resident headline.  Pageable numpy buffers, as a C host would pass."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

n, w, h = 10000, 160, 120
pkg = entry.load_package()
orc = entry.load_oracle()
ctx = pkg.Context(0)
blob, offs, lens = orc.synth_stream(0xA11CE, 0, n, w, h, threads=16)
out = np.zeros((n, h, ctx.stride(w)), np.uint8)
st = np.zeros(n, np.int32)
for _ in range(2):
    ctx.decode_batch(blob, blob.size, offs, lens, n, w, h, 0, out, st)
t = time.perf_counter()
reps = 5
for _ in range(reps):
    ctx.decode_batch(blob, blob.size, offs, lens, n, w, h, 0, out, st)
dt = (time.perf_counter() - t) / reps
assert (st == 0).all()
print(json.dumps({"host_buffers_frames_per_s": n / dt, "ms_per_batch": dt * 1e3, "bytes_out_per_batch": out.nbytes,
                  "effective_GBps": (out.nbytes + blob.size) / dt / 1e9}))
