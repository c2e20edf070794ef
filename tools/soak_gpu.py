#!/usr/bin/env python3
"""Soak runs of the HIP path against the CPU oracle, longer and wider than the test suite's cases (run on the GPU box:
`gpurun -- python tools/soak_gpu.py adpcm 0 150`).  Each seed draws its own geometry / content / damage; a run prints one
line per mismatch and a summary, and exits non-zero if anything differed.

  adpcm  LO HI   chained IMA-ADPCM encode of ragged streams (a few hundred to a few thousand chunks, some empty or short),
                 three times each, every byte against the oracle's sequential encode.  Round 4: two holes in the index
                 chain's list handling passed every test and showed up here within forty streams.
  decode LO HI   batches of random size and content (synthetic, noise, flat, noisy synthetic), a third of the chunks damaged
                 or cut, every entropy lane count, both output modes: pixels and statuses against the oracle.
  ffmpeg LO HI   the same kind of batches through AMVHIP_FLAG_FFMPEG (planar YUVJ420P, the fork's own arithmetic and flip; odd
                 sizes too), every second seed with AMVHIP_FLAG_FFMPEG_KEEP over a buffer of random bytes: every byte against
                 the oracle's restatement -- whole blocks in front of a chunk's first error written, nothing else touched.
  encode LO HI   random geometries, strides, channel orders, quantiser biases and content: chunks against the oracle's.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from conftest import SEED  # noqa: E402


def soak_adpcm(pkg, orc, lo, hi):
    ctx = pkg.Context(0)
    bad = repaired = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(200, 6000))
        sizes = [1378 if rng.integers(0, 10) else 2 * int(rng.integers(0, 700)) for _ in range(n)]
        pcm_offs = np.cumsum([0] + sizes).astype(np.uint64)
        pcm = orc.synth_audio(SEED + seed, 777, int(pcm_offs[-1]) + 2)
        offs = np.cumsum([0] + [8 + s // 2 for s in sizes]).astype(np.uint64)
        want, idx = [], 0
        for i in range(n):
            seg = pcm[int(pcm_offs[i]):int(pcm_offs[i + 1])]
            if seg.size:
                chunk, idx = orc.adpcm_encode_chunk(seg, idx)
            else:
                chunk = bytes([0, 0, idx, 0, 0, 0, 0, 0])
            want.append(chunk)
        want = b"".join(want)
        for rep in range(3):
            blob = np.full(int(offs[-1]), 0xEE, np.uint8)
            ctx.adpcm_encode_batch(pcm, pcm.size, pcm_offs[:-1].copy(), np.array(sizes, np.uint32), n, None, blob, blob.size, offs[:-1].copy())
            if ctx.adpcm_chain_stats()["exhaustive"]:      # the chain's check (or its settle kernel) sent the stream down the slow route
                repaired += 1
                print("exhaustive route taken: adpcm seed", seed, "rep", rep, "chunks", n, ctx.adpcm_chain_stats(), flush=True)
            if blob.tobytes() != want:
                bad += 1
                d = np.nonzero(blob != np.frombuffer(want, np.uint8))[0]
                ch = int(np.searchsorted(offs, d[0], side="right") - 1)
                print("MISMATCH adpcm seed", seed, "rep", rep, "chunks", n, "bytes", d.size, "first in chunk", ch, "at", int(d[0] - offs[ch]),
                      ctx.adpcm_chain_stats(), flush=True)
    print("adpcm: streams that took the exhaustive route (none expected on ordinary audio):", repaired)
    return bad


def soak_decode(pkg, orc, lo, hi):
    import test_gpu_parity as T
    bad = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        w, h, n = 2 * int(rng.integers(8, 220)), 2 * int(rng.integers(8, 160)), int(rng.integers(3, 90))
        chunks = []
        for t in range(n):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                src = orc.synth_frame(SEED, 1000 * seed + t, w, h)
            elif kind == 1:
                src = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            elif kind == 2:
                src = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
            else:
                src = (orc.synth_frame(SEED, t, w, h).astype(np.int32) + rng.integers(-40, 41, (h, w, 3))).clip(0, 255).astype(np.uint8)
            c = bytearray(orc.encode_frame(src, w, h, qbias=int(rng.integers(0, 2)) * 128))
            if rng.random() < 0.35 and len(c) > 8:
                for _ in range(int(rng.integers(1, 5))):
                    c[int(rng.integers(2, len(c) - 2))] ^= 1 << int(rng.integers(0, 8))
            if rng.random() < 0.1:
                c = c[: int(rng.integers(2, len(c)))]
            chunks.append(bytes(c))
        flags = seed & 1
        want, wst = T._oracle_decode(orc, chunks, w, h, flags)
        for lanes in ("1", "2", "16", None):
            if lanes:
                os.environ["AMVHIP_SYNC_LANES"] = lanes
            else:
                os.environ.pop("AMVHIP_SYNC_LANES", None)
            ctx = pkg.Context(0)
            got, st = T._gpu_decode(ctx, chunks, w, h, flags, pad_front=int(rng.integers(0, 4)))
            ctx.close()
            if not ((st == wst).all() and (got == want).all()):
                bad += 1
                print("MISMATCH decode seed", seed, w, h, n, "lanes", lanes, flush=True)
    os.environ.pop("AMVHIP_SYNC_LANES", None)
    return bad


def soak_ffmpeg(pkg, orc, lo, hi):
    import torch
    import test_gpu_parity as T
    bad = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        w, h, n = int(rng.integers(16, 400)), int(rng.integers(16, 300)), int(rng.integers(3, 60))
        ew, eh = w + (w & 1), h + (h & 1)            # (the encoder takes even sizes; the decoder is told the odd one)
        chunks = []
        for t in range(n):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                src = orc.synth_frame(SEED, 1000 * seed + t, ew, eh)
            elif kind == 1:
                src = rng.integers(0, 256, (eh, ew, 3)).astype(np.uint8)
            else:
                src = (orc.synth_frame(SEED, t, ew, eh).astype(np.int32) + rng.integers(-40, 41, (eh, ew, 3))).clip(0, 255).astype(np.uint8)
            c = bytearray(orc.encode_frame(src, ew, eh, qbias=int(rng.integers(0, 2)) * 128))
            if rng.random() < 0.4 and len(c) > 12:
                for _ in range(int(rng.integers(1, 5))):
                    c[int(rng.integers(4, len(c) - 2))] ^= 1 << int(rng.integers(0, 8))
            if rng.random() < 0.15:
                c = c[: int(rng.integers(2, len(c)))]
            chunks.append(bytes(c))
        keep = bool(seed & 1)
        fb = orc.decode_frame_ffmpeg(chunks[0], w, h)[0].size
        before = rng.integers(0, 256, (n, fb), dtype=np.uint8)
        want, wst = [], []
        for i, c in enumerate(chunks):
            o, st, _ = orc.decode_frame_ffmpeg_keep(c, w, h, before[i]) if keep else orc.decode_frame_ffmpeg(c, w, h)
            want.append(o)
            wst.append(st)
        want, wst = np.stack(want), np.array(wst, np.int32)
        blob, offs, lens, nbytes = T._blob_of(chunks, int(rng.integers(0, 4)))
        flags = pkg.FLAG_FFMPEG | (pkg.FLAG_FFMPEG_KEEP if keep else 0)
        for lanes in ("1", "4", "64", None):
            if lanes:
                os.environ["AMVHIP_SYNC_LANES"] = lanes
            else:
                os.environ.pop("AMVHIP_SYNC_LANES", None)
            ctx = pkg.Context(0)
            d_out = torch.from_numpy(before.copy()).to("cuda:0")
            d_st = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
            ctx.decode_batch_dev(T._t(blob), nbytes, T._t(offs), T._t(lens), n, w, h, flags, d_out, d_st, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ctx.close()
            if not ((d_st.cpu().numpy() == wst).all() and (d_out.cpu().numpy() == want).all()):
                bad += 1
                print("MISMATCH ffmpeg seed", seed, w, h, n, "keep" if keep else "plain", "lanes", lanes, flush=True)
    os.environ.pop("AMVHIP_SYNC_LANES", None)
    return bad


def soak_encode(pkg, orc, lo, hi):
    import torch
    import test_gpu_parity as T
    ctx = pkg.Context(0)
    bad = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        w, h, n = 2 * int(rng.integers(4, 330)), 2 * int(rng.integers(4, 250)), int(rng.integers(1, 40))
        n = max(1, min(n, 6_000_000 // (w * h)))
        bgr, qbias = int(rng.integers(0, 2)), 128 * int(rng.integers(0, 2))
        stride = w * 3 + int(rng.integers(0, 3)) * 5
        kind = int(rng.integers(0, 5))
        if kind == 0:
            pix = rng.integers(0, 256, (n, h, w, 3)).astype(np.uint8)
        elif kind == 1:
            pix = np.stack([orc.synth_frame(SEED, 31 * seed + t, w, h) for t in range(n)])
        elif kind == 2:
            pix = np.full((n, h, w, 3), int(rng.integers(0, 256)), np.uint8)
        elif kind == 3:
            pix = np.stack([orc.synth_frame(SEED, 7 * seed + t, w, h) for t in range(n)]).astype(np.int32)
            pix = (pix + rng.integers(-60, 61, pix.shape)).clip(0, 255).astype(np.uint8)
        else:                                       # a flat picture with islands of noise
            pix = np.full((n, h, w, 3), 128, np.uint8)
            for _ in range(int(rng.integers(1, 30))):
                y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
                patch = pix[int(rng.integers(0, n)), y:y + 8, x:x + 8]
                patch[...] = rng.integers(0, 256, patch.shape)
        src = np.zeros((n, h, stride), np.uint8)
        src[:, :, : w * 3] = pix.reshape(n, h, w * 3)
        src[:, :, w * 3:] = 0xEE
        cap = ctx.encode_bound(w, h) * n
        d_blob = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
        d_offs = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        d_lens = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        ctx.encode_batch_dev(T._t(src), stride, bgr, n, w, h, qbias, d_blob, cap, d_offs, d_lens)
        torch.cuda.synchronize()
        blob, offs, lens = d_blob.cpu().numpy(), d_offs.cpu().numpy(), d_lens.cpu().numpy()
        for i in range(n):
            want = orc.encode_frame(np.ascontiguousarray(pix[i]), w, h, bgr=bool(bgr), qbias=qbias)
            if int(lens[i]) != len(want) or blob[int(offs[i]):int(offs[i]) + len(want)].tobytes() != want:
                bad += 1
                print("MISMATCH encode seed", seed, w, h, n, "frame", i, "kind", kind, flush=True)
                break
    return bad


def main():
    if len(sys.argv) != 4 or sys.argv[1] not in ("adpcm", "decode", "ffmpeg", "encode"):
        raise SystemExit(__doc__)
    what, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    pkg = entry.build()
    orc = entry.load_oracle()
    orc.lib()
    bad = {"adpcm": soak_adpcm, "decode": soak_decode, "ffmpeg": soak_ffmpeg, "encode": soak_encode}[what](pkg, orc, lo, hi)
    print("soak", what, "seeds", lo, "..", hi - 1, "mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
