#!/usr/bin/env python3
"""Kernel-level timing of the encode path for experiments (not the bench): synthetic WxH frames made on the device,
amvhip_encode_batch_dev timed through the library's own profiling hooks.  AMVHIP_LIB selects a library build (variant
builds of tools/build_variant.py may write wrong bytes: nothing is checked here)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8000)
ap.add_argument("--width", type=int, default=320)
ap.add_argument("--height", type=int, default=240)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
pkg = entry.load_package()
ctx = pkg.Context(0)
dev = "cuda:0"
w, h, n = a.width, a.height, a.frames
s = torch.cuda.current_stream().cuda_stream
rgb = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
ctx.synth_frames_dev(0xA11CE, 0, n, w, h, rgb, s)
cap = max(1 << 20, n * w * h)
blob = torch.zeros(cap, dtype=torch.uint8, device=dev)
offs = torch.zeros(n, dtype=torch.int64, device=dev)
lens = torch.zeros(n, dtype=torch.int32, device=dev)
for _ in range(2):
    ctx.encode_batch_dev(rgb, w * 3, 0, n, w, h, 0, blob, cap, offs, lens, s)
torch.cuda.synchronize()
ctx.prof_enable(True)
ctx.prof_reset()
for _ in range(a.steps):
    ctx.encode_batch_dev(rgb, w * 3, 0, n, w, h, 0, blob, cap, offs, lens, s)
torch.cuda.synchronize()
out = {"lib": os.environ.get("AMVHIP_LIB", "default"), "mean_chunk": float(lens.sum().item()) / n}
for k in (pkg.K_FDCT, pkg.K_PACK, pkg.K_PACK_SERIAL, pkg.K_COMPACT):
    cnt, ms = ctx.prof_read(k)
    if cnt:
        out[ctx.kernel_name(k)] = round(ms / a.steps, 4)
print(json.dumps(out))
