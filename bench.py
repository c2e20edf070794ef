#!/usr/bin/env python3
"""bench.py -- AMV decode throughput on MI355X (BASELINE.json: 160x120 decode, bit-exact vs amvlib).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the decode hot path (entropy kernel + reconstruction kernel, through the
C ABI entry amvhip_decode_batch_dev) over one batch: a synthetic 160x120 AMV stream of --frames
frames per GPU (default 10 000), compressed chunks already resident in HBM
when the timed region starts, decoded BGR24 frames left in HBM.  Frames shard by contiguous range,
one process per GPU, no collective inside the codec path ("scaling": "weak": every GPU decodes
its own --frames frames).  The stream is made on the device, outside the timed region, by the
library's own generator + encoder (both proven byte-identical to the CPU oracle by the tests and
spot-checked again here).

Extra objects on the JSON line:
  roofline     the dominant kernel against the HBM roof: algorithmic bytes of the path per launch
               (sum of chunk bytes + 3*W*H per frame) / that kernel's mean duration
               measured with HIP events on the launch stream inside the timed region.
  cpu_baseline the CPU oracle (a port of amvlib's algorithm; the reference itself cannot travel to
               the GPU box) decoding a bounded sample of the same stream on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SEED = 0xA11CE
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU per step")
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--height", type=int, default=120)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="frames of the stream the CPU baseline decodes")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # RCCL over xGMI
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the decode path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    pkg = entry.load_package()
    sh = entry._load(entry.PKG_NAME + ".sharding", os.path.join(entry.PKG_DIR, "sharding.py"))
    ctx = pkg.Context(local)
    w, h, n = args.width, args.height, args.frames
    stream = torch.cuda.current_stream().cuda_stream
    fb = ctx.frame_bytes(w, h)

    # ---- the stream of this rank: frames [rank*n, (rank+1)*n) of the seeded source, encoded on the device
    first = rank * n
    cap = max(1 << 20, n * w * h)             # ~0.2 B/pixel in practice; checked below
    d_blob = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_offs = torch.zeros(n, dtype=torch.int64, device=dev)
    d_lens = torch.zeros(n, dtype=torch.int32, device=dev)
    chunk = 2000                              # encode in slices to bound the RGB staging buffer
    pos = 0
    d_rgb = torch.empty((min(chunk, n), h, w, 3), dtype=torch.uint8, device=dev)
    t_offs = torch.zeros(chunk, dtype=torch.int64, device=dev)
    for lo in range(0, n, chunk):
        cnt = min(chunk, n - lo)
        ctx.synth_frames_dev(SEED, first + lo, cnt, w, h, d_rgb, stream)
        ctx.encode_batch_dev(d_rgb, w * 3, 0, cnt, w, h, pkg.QBIAS_AMV, d_blob[pos:], cap - pos, t_offs, d_lens[lo:], stream)
        torch.cuda.synchronize()
        d_offs[lo:lo + cnt] = t_offs[:cnt] + pos
        pos = int(d_offs[lo + cnt - 1].item()) + int(d_lens[lo + cnt - 1].item())
        pos = (pos + 3) & ~3                  # keep every slice's base 4-byte aligned
        if pos > cap:
            raise SystemExit("synthetic stream overflowed its buffer")
    del d_rgb
    stream_bytes = int(d_lens.sum().item())
    d_out = torch.empty((n, h, ctx.stride(w)), dtype=torch.uint8, device=dev)
    d_st = torch.empty(n, dtype=torch.int32, device=dev)

    def step():
        ctx.decode_batch_dev(d_blob, cap, d_offs, d_lens, n, w, h, 0, d_out, d_st, stream)

    # ---- correctness gate before any timing: sample frames against the CPU oracle (checker only)
    step()
    torch.cuda.synchronize()
    if int((d_st != 0).sum().item()) != 0:
        raise SystemExit("decode reported errors on the synthetic stream")
    orc = entry.load_oracle()
    offs_h, lens_h = d_offs.cpu().numpy(), d_lens.cpu().numpy()
    for i in sorted({0, 1, n // 2, n - 1}):
        ch = d_blob[int(offs_h[i]):int(offs_h[i]) + int(lens_h[i])].cpu().numpy().tobytes()
        want, st, _ = orc.decode_frame(ch, w, h)
        if st != 0 or not (d_out[i].cpu().numpy() == want).all():
            raise SystemExit("HIP decode differs from the oracle at frame %d" % (first + i))
        if ch != orc.encode_frame(orc.synth_frame(SEED, first + i, w, h), w, h):
            raise SystemExit("device-made stream differs from the oracle's encoder at frame %d" % (first + i))

    # ---- timed region
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.prof_enable(True)
    ctx.prof_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    elapsed = sh.max_over_ranks(elapsed, dev)
    total_frames = sh.sum_over_ranks(float(n * args.steps), dev)

    ctx.entropy_stats(True)          # one extra, untimed step: how many synchronisation rounds the frames needed
    step()
    sync = ctx.entropy_stats(False)

    kern = {}
    for k in (pkg.K_UNSTUFF, pkg.K_HUFFMAN, pkg.K_HUFFMAN_SERIAL, pkg.K_RECON):
        launches, ms = ctx.prof_read(k)
        kern[ctx.kernel_name(k)] = {"launches": launches, "avg_ms": ms / max(launches, 1)}
    dom = max(kern, key=lambda name: kern[name]["avg_ms"])
    algo_bytes = stream_bytes + n * 3 * w * h                 # per launch (= per step, per GPU)
    achieved = algo_bytes / (kern[dom]["avg_ms"] * 1e-3) / 1e9 if kern[dom]["avg_ms"] > 0 else 0.0

    # HBM bytes per launch of the dominant kernel from the PMC passes of this same command
    # (tools/summarize_pmc.py -> profiles/*_traffic.json); only quoted for the workload that was profiled
    traffic = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if (w, h, n) == (160, 120, 10000):
            for kname, rec in prof["kernels"].items():
                if kname.split("<")[0].endswith(dom) and rec["hbm_corrected"] > 1e6:
                    traffic = rec["hbm_corrected"]
    except (OSError, ValueError, KeyError):
        pass

    result = {
        "metric": "AMV frames/sec/GPU (160x120 decode, bit-exact)" if (w, h) == (160, 120) else "AMV frames/sec/GPU (%dx%d decode, bit-exact)" % (w, h),
        "value": total_frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": "%dx%d AMV decode, %d-frame synthetic stream per GPU, chunks resident in HBM" % (w, h, n),
                   "frames_per_gpu": n, "mean_chunk_bytes": stream_bytes / n, "parallelism": "frame-range x%d" % world,
                   "per_gpu_frames_per_s": total_frames / elapsed / world},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": algo_bytes, "kernels": kern,
                     "path_achieved": algo_bytes / (elapsed / args.steps) / 1e9,
                     "entropy_sync_rounds": {"mean": sync["rounds"] / max(sync["frames"], 1), "max": sync["max_rounds"]}},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        m = min(args.cpu_sample, n)
        blob_h = d_blob[: int(offs_h[m - 1]) + int(lens_h[m - 1]) + 16].cpu().numpy()
        o64, l32 = offs_h[:m].astype(np.uint64), lens_h[:m].astype(np.uint32)
        # a one-GPU box owns a 16-core share of the host whatever cpu_count says
        cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("AMV_BENCH_CORES", "16"))))
        t = time.perf_counter()
        _, st1 = orc.decode_batch(blob_h, o64, l32, w, h, 0, threads=1)
        t1 = time.perf_counter() - t
        reps, tn = 0, 0.0
        while tn < 5.0 and reps < 64:         # bounded: a few seconds of all-core work
            t = time.perf_counter()
            orc.decode_batch(blob_h, o64, l32, w, h, 0, threads=cores)
            tn += time.perf_counter() - t
            reps += 1
        assert (st1 == 0).all()
        result["cpu_baseline"] = {"value": m * reps / tn, "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": "first %d frames of the same stream, CPU oracle (amvlib algorithm restated in C), "
                                            "frame-sharded over %d OpenMP threads, %d passes" % (m, cores, reps),
                                  "single_thread_value": m / t1}
    if rank == 0:
        print(json.dumps(result))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
