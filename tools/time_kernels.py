#!/usr/bin/env python3
"""Kernel-level timing of the decode path for experiments (not the bench): builds the synthetic
160x120 stream on the device, then times the entropy stage and the reconstruction stage separately
with HIP events through the library's own profiling hooks.  AMVHIP_LIB selects a library build."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=10000)
ap.add_argument("--width", type=int, default=160)
ap.add_argument("--height", type=int, default=120)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--serial", action="store_true")
ap.add_argument("--noise", action="store_true", help="white-noise frames (every coefficient of every block non-zero)")
ap.add_argument("--mixed", type=int, default=0, help="every k-th frame white noise")
ap.add_argument("--same", type=int, default=-1, help="every frame of the batch is a copy of this frame (no spread in sync rounds)")
ap.add_argument("--trace", default="", help="write the per-wave lines of the several-lanes-per-frame launch (amvhip_entropy_trace) to this file")
a = ap.parse_args()
pkg = entry.load_package()
ctx = pkg.Context(0)
dev = "cuda:0"
w, h, n = a.width, a.height, a.frames
s = torch.cuda.current_stream().cuda_stream
cap = max(1 << 20, n * w * h * (2 if a.noise or a.mixed else 1))
blob = torch.zeros(cap, dtype=torch.uint8, device=dev)
offs = torch.zeros(n, dtype=torch.int64, device=dev)
lens = torch.zeros(n, dtype=torch.int32, device=dev)
pos = 0
toffs = torch.zeros(2000, dtype=torch.int64, device=dev)
rgb = torch.empty((2000, h, w, 3), dtype=torch.uint8, device=dev)
for lo in range(0, n, 2000):
    cnt = min(2000, n - lo)
    ctx.synth_frames_dev(0xA11CE, lo, cnt, w, h, rgb, s)
    if a.noise:
        torch.cuda.synchronize()
        rgb[:cnt] = torch.randint(0, 256, (cnt, h, w, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
    if a.mixed:
        torch.cuda.synchronize()
        k = len(range(a.mixed - 1, cnt, a.mixed))
        rgb[a.mixed - 1:cnt:a.mixed] = torch.randint(0, 256, (k, h, w, 3), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
    ctx.encode_batch_dev(rgb, w * 3, 0, cnt, w, h, 0, blob[pos:], cap - pos, toffs, lens[lo:], s)
    torch.cuda.synchronize()
    offs[lo:lo + cnt] = toffs[:cnt] + pos
    pos = (int(offs[lo + cnt - 1]) + int(lens[lo + cnt - 1]) + 3) & ~3
if a.same >= 0:
    offs[:] = offs[a.same].clone()
    lens[:] = lens[a.same].clone()
nblk = ((w + 15) // 16) * ((h + 15) // 16) * 6
st = torch.empty(n, dtype=torch.int32, device=dev)
ok = torch.empty(n, dtype=torch.int32, device=dev)
out = torch.empty((n, h, ctx.stride(w)), dtype=torch.uint8, device=dev)
if a.serial:
    ctx.set_entropy_mode(pkg.ENTROPY_SERIAL)
for it in range(a.steps + 2):
    if it == 2:
        ctx.prof_enable(True)
        ctx.prof_reset()
    ctx.decode_batch_dev(blob, cap, offs, lens, n, w, h, 0, out, st, s)
torch.cuda.synchronize()
res = {"lib": os.path.basename(pkg.LIB_PATH), "frames": n, "size": [w, h], "bad": int((st != 0).sum())}
for k in (pkg.K_UNSTUFF, pkg.K_HUFFMAN, pkg.K_HUFFMAN_SERIAL, pkg.K_RECON):
    cnt, ms = ctx.prof_read(k)
    res[ctx.kernel_name(k)] = round(ms / a.steps, 4)   # per step (a step may launch a kernel more than once)
ctx.entropy_stats(True)
ctx.decode_batch_dev(blob, cap, offs, lens, n, w, h, 0, out, st, s)
es = ctx.entropy_stats(False)
res["sync"] = {"mean_rounds": round(es["rounds"] / max(es["frames"], 1), 2), "max_rounds": es["max_rounds"],
               "kclk_per_wave": {k: round(v / 1e3, 1) for k, v in es["clocks_per_wave"].items()}}
if a.trace:
    import numpy as np
    tr = ctx.entropy_trace(16384)
    tr = tr[tr[:, 1] != 0]
    if len(tr):
        t0 = int(tr[:, 0].min())
        dur = (tr[:, 1] - tr[:, 0]).astype(np.int64) / 100.0          # us (100 MHz)
        beg = (tr[:, 0].astype(np.int64) - t0) / 100.0
        end = beg + dur
        rounds = (tr[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
        share = (tr[:, 6] >> np.uint64(32)).astype(np.int64)
        order = np.argsort(-end)
        res["trace"] = {"waves": int(len(tr)), "launch_us": float(end.max()), "begin_us_p50_max": [float(np.median(beg)), float(beg.max())],
                        "duration_us_p50_p99_max": [float(np.percentile(dur, q)) for q in (50, 99, 100)],
                        "rounds_hist": {int(k): int((rounds == k).sum()) for k in np.unique(rounds)},
                        "last_to_end": [{"end_us": float(end[i]), "begin_us": float(beg[i]), "rounds": int(rounds[i]), "share_bits": int(share[i]),
                                         "kclk": [int(tr[i, 2 + q]) // 1000 for q in range(4)]} for i in order[:12]]}
        np.save(a.trace, tr)
print(json.dumps(res))
