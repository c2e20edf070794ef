#!/usr/bin/env python3
"""Kernel-level timing of the chained IMA-ADPCM encode for experiments (not the bench; no correctness gate, so that
variant builds that leave work out can be timed).  AMVHIP_LIB selects a library build; run under rocprofv3 --stats for
the per-kernel split."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=200000)
ap.add_argument("--samples", type=int, default=1378)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
pkg = entry.load_package()
ctx = pkg.Context(0)
dev = "cuda:0"
n, spf = a.chunks, a.samples
s = torch.cuda.current_stream().cuda_stream
pcm = torch.empty(n * spf, dtype=torch.int16, device=dev)
ctx.synth_audio_dev(0xA11CE, 0, n * spf, pcm, s)
pcm_offs = torch.arange(n, dtype=torch.int64, device=dev) * spf
nsamp = torch.full((n,), spf, dtype=torch.int32, device=dev)
clen = 8 + spf // 2
offs = torch.arange(n, dtype=torch.int64, device=dev) * clen
chunks = torch.zeros(n * clen + 16, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for it in range(a.steps + 2):
    if it == 2:
        ctx.prof_enable(True)
        ctx.prof_reset()
    ctx.adpcm_encode_batch_dev(pcm, pcm_offs, nsamp, n, None, chunks, offs, s)
torch.cuda.synchronize()
launches, ms = ctx.prof_read(pkg.K_ADPCM_ENC)
stats = ctx.adpcm_chain_stats()
print(json.dumps({"lib": os.path.basename(pkg.LIB_PATH), "chunks": n, "encode_ms": ms / max(launches, 1), "sweeps_env": os.environ.get("AMVHIP_ADPCM_SWEEPS"),
                  "exhaustive": stats["exhaustive"], "recoded": stats["recoded"][:12],
                  "checksum": int(chunks.to(torch.int64).sum().item())}))
