#!/usr/bin/env python3
"""bench.py -- AMV codec throughput on MI355X.  Default: BASELINE.json's metric (160x120 decode,
bit-exact vs amvlib) on its configs[1].

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (starts the N ranks itself, one child process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W     (--gpus must equal WORLD_SIZE)

Exit codes: 0 clean; 2 more GPUs asked for than visible; 3 a leg beside the headline hung (the line is written first);
4 one raised a device / RCCL error, on this rank or on a peer (the line is written first).

The line of a run with ranks is as complete as the single-GPU one: `cpu_baseline` (rank 0 runs the CPU path, the others wait
in a barrier outside every timed region), configs[3] as stated (`strong10k_*`), and every other BASELINE config run by every
rank on frames of its own -- rates summed over the ranks, times the slowest rank's -- with `rccl_ranks`, `run_status` and
`strong10k_status` in front (CONFIG_HEAD_RANKS).

One step = one pass of the hot path, through the C ABI, over one batch that is already resident in
HBM when the timed region starts; results stay in HBM.  Frames shard by contiguous range, one
process per GPU, no collective inside the codec path ("scaling": "weak": every GPU works on its own
--frames frames).  Synthetic sources are made on the device, outside the timed region, by the
library's generator (+ encoder), both proven byte-identical to the CPU oracle by the tests and
spot-checked again here before anything is timed.

  --workload decode      (default) --frames (160 000) chunks of WxH (160x120) -> BGR24 frames  configs[1], [3]
  --workload encode      RGB24 frames of WxH (320x240) -> chunks; the round trip through the
                         bit-exact decoder is PSNR-checked against the source                configs[2]
  --workload coresident  WxH (320x240) video decode on one HIP stream with IMA-ADPCM decode +
                         encode of the frames' audio chunks on a second stream               configs[4]
  --workload adpcm       IMA-ADPCM chunks -> PCM -> chunks, audio alone
  --workload amvlib      the drop-in surface, one frame per call: AmvReadNextFrame / AmvVideoDecode /
                         AmvAudioDecode from a C host over AMV files (PCIe inclusive; not the headline)

Extra objects on the JSON line:
  roofline     the dominant kernel against the HBM roof: algorithmic bytes of the path per launch
               (video: chunk bytes + 3*W*H per frame; audio: chunk bytes + 2 per sample)
               / that kernel's mean duration measured with HIP events on the launch stream inside the
               timed region.
  cpu_baseline the CPU oracle (a port of the reference's algorithm; the reference itself cannot
               travel to the GPU box) on a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SEED = 0xA11CE
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SAMPLES_PER_FRAME = 1378  # 22050 Hz mono at 16 fps: the audio chunk that travels with one video frame
# frames per GPU per step of the decode workload.  The path is throughput-bound only when the chip is full:
# 10 000 frames of 160x120 are 2 500 waves of the entropy kernel (10 per CU) and the step then lasts as long
# as one wave's serial chain; DESIGN.md section 9 has the sweep (`--frames 10000` gives the 10 000-frame figure).
DECODE_FRAMES = 160000
# what the strong-scaling leg (a few seconds of work) gets before the line leaves without it
STRONG_LEG_SECONDS = float(os.environ.get("AMV_BENCH_STRONG_SECONDS", "240"))


EXIT_FAILED = 4         # a leg raised a device / RCCL error: the line is written, the run is not clean
EXIT_HUNG = 3           # a leg did not end: the line is written by the watchdog, the run is not clean
EXIT_USAGE = 2          # --gpus asks for more devices than there are
LAUNCH_GRACE_SECONDS = float(os.environ.get("AMV_BENCH_LAUNCH_GRACE", "60"))   # launch_ranks: after one rank failed
# A rehearsal of the N-rank flow where there is ONE GPU: AMV_BENCH_BACKEND=gloo (torch.distributed's gloo moves device tensors
# too) and AMV_BENCH_ONE_DEVICE=1 (every rank on device 0) run the real ranks, the real kernels, the real exchange and the real
# record -- everything but RCCL and xGMI.  The line says so (config.rehearsal); it is never a scaling figure.
BACKEND = os.environ.get("AMV_BENCH_BACKEND", "nccl")
ONE_DEVICE = os.environ.get("AMV_BENCH_ONE_DEVICE", "") not in ("", "0")
LAUNCH_LIMIT_SECONDS = float(os.environ.get("AMV_BENCH_LAUNCH_LIMIT", "3600"))  # launch_ranks: the whole run, whatever it waits for
# what a leg beside the headline (a few seconds of work each) gets before the line leaves without it
LEG_SECONDS = float(os.environ.get("AMV_BENCH_LEG_SECONDS", "300"))
FLAG_KEY = "amv_bench_failed"   # the ranks' side channel: a key of the rendezvous store (class Peers)


class Env:
    json_fd = None
    failed = None
    strong_phase = ""       # where the guarded leg of this rank is (the strong leg's phases, a secondary leg's name)
    verdict = None          # the guarded leg's recorder: verdict(status) writes the leg's status scalars into result["config"]
    result = None           # the line as far as it has got (what a watchdog or the peers' flag writes out)
    peers = None
    world = 1
    rank = 0
    dist = False
    line_written = False    # emit_line: the one JSON line has gone out


def leave(code):
    """every exit that does not unwind (a communicator may be in no state to be torn down): flush what Python holds first"""
    try:
        sys.stdout.flush()
        sys.stderr.flush()
    except (OSError, ValueError):
        pass
    os._exit(code)


LINE_LOCK = threading.Lock()


def emit_line(E, result):
    """ONE JSON line per run, whichever thread gets to write it (the main thread at the end, a leg's watchdog, the peers'
    watcher): the first caller writes, later ones find it written.  A watchdog or the watcher may come while the main thread
    is adding keys to the dictionaries -- json.dumps then raises "dictionary changed size during iteration"; it is tried again
    a few times (the main thread is about to block in a collective or has), and what goes out in the worst case is the
    line's scalars with the reason in run_status."""
    with LINE_LOCK:
        if getattr(E, "line_written", False):
            return
        E.line_written = True
        payload = None
        for _ in range(8):
            try:
                result["config"] = ordered_config(result["config"])
                payload = json.dumps(result)
                break
            except RuntimeError:
                time.sleep(0.05)
        if payload is None:
            payload = json.dumps({k: v for k, v in list(result.items()) if isinstance(v, (int, float, str, bool, type(None)))}
                                 | {"config": {"run_status": "error: the line was being written to while it had to leave"}})
        if E.rank == 0 and E.json_fd is not None:
            os.write(E.json_fd, (payload + "\n").encode())


def write_line_now(E, status=None):
    """the line as far as it has got, from whichever thread has to write it (rank 0 only; the others write nothing)"""
    if E.result is None:
        return
    if status is not None:
        if E.verdict is not None:
            E.verdict(status)
        else:
            E.result["config"]["run_status"] = status
    emit_line(E, E.result)


class Peers:
    """The ranks' side channel: one key of the rendezvous store (TCP on the host, nothing to do with the GPUs or RCCL).
    A rank whose leg raised sets the key BEFORE it leaves; every rank polls it from a thread, so that a rank blocked in a
    collective its failed peer will never join does not wait for its own watchdog -- and, under a launcher that ends the
    survivors as soon as one rank has failed (torch.distributed.run does), rank 0 has written the line by then."""

    def __init__(self, E, store, period=0.5):
        import threading
        self.E, self.store, self.period = E, store, period
        self.stop = threading.Event()
        threading.Thread(target=self.watch, daemon=True).start()

    def raise_flag(self, text, linger=3.0):
        """tell the others, then give their pollers `linger` seconds to act on it before this rank's exit ends the run"""
        try:
            self.store.set(FLAG_KEY, "rank %d: %s" % (self.E.rank, text))
        except Exception:
            return
        if self.E.world > 1:
            time.sleep(linger)

    def watch(self):
        E = self.E
        while not self.stop.wait(self.period):
            try:
                if not self.store.check([FLAG_KEY]):
                    continue
                text = self.store.get(FLAG_KEY).decode(errors="replace")
            except Exception:
                return                                   # the store is gone (its host left): nothing more to learn here
            if text.startswith("rank %d:" % E.rank):
                return                                   # this rank's own flag: it is leaving by itself
            write_line_now(E, "error: peer failed (%s) while this rank was in %s" % (text[:300], E.strong_phase or "the headline"))
            os.write(2, ("bench.py: rank %d: a peer failed (%s); leaving with code %d\n" % (E.rank, text[:300], EXIT_FAILED)).encode())
            leave(EXIT_FAILED)


def balanced_src_share(world, n_total=10000, frame_bytes=57600, link_gb_s=153.0):
    """The share of configs[3]'s stream the rank that holds it keeps so that its own decode ends with the others' decode + send
    (DESIGN.md section 10's table, solved for s): a rank's decode of n of these frames takes 0.27 ms + 0.044 us x n on one
    MI355X (0.31 / 0.37 / 0.49 / 0.73 ms at 1 250 / 2 500 / 5 000 / 10 000 frames, tools/time_kernels.py + the launches' gaps),
    a sender's frames then cross its ONE xGMI link (~153 GB/s).  A model from single-GPU times: what it predicts is what the
    second strong leg exists to measure."""
    if world < 2:
        return None
    ms = lambda n: 0.27 + 0.044e-3 * n
    cost = lambda s: max(ms(s * n_total), ms((1 - s) * n_total / (world - 1)) + (1 - s) * n_total / (world - 1) * frame_bytes / (link_gb_s * 1e6))
    return min((i / 100.0 for i in range(101)), key=cost)


def share_flat(strong, status):
    """the second strong leg (the source keeps balanced_src_share of the stream), as flat scalars of config"""
    return {"strong10k_srcshare": (strong or {}).get("src_share"), "strong10k_srcshare_status": status,
            "strong10k_srcshare_ms": (strong or {}).get("ms_per_step"), "strong10k_srcshare_fps": (strong or {}).get("frames_per_s")}


def strong_flat(strong, status):
    """configs[3] as stated, as FLAT scalars of `config` (a record that keeps scalars only still carries them, and a hang or
    an error of the exchange is a value one can read, not a missing key)"""
    ph = (strong or {}).get("phase_ms_max_over_ranks", {})
    return {"rccl_ranks": (strong or {}).get("rccl_ranks", dist.get_world_size() if dist.is_initialized() else 0),
            "strong10k_status": status,
            "strong10k_ms": (strong or {}).get("ms_per_step"),
            "strong10k_fps": (strong or {}).get("frames_per_s"),
            "strong10k_scatter_ms": ph.get("scatter"), "strong10k_decode_ms": ph.get("decode"),
            "strong10k_gather_ms": ph.get("gather")}


def make_video_stream(E, first, n, w, h, noise_every=0):
    """frames [first, first+n) of the seeded source, encoded on the device -> blob, cap, offs, lens, bytes.
    noise_every = k: every k-th frame (frame number % k == k - 1) is white noise instead -- every coefficient of every block
    non-zero, a chunk 2.7 times the stream's mean (the "mixed" stream: what per-frame workspace is for)"""
    ctx, pkg, dev, stream = E.ctx, E.pkg, E.dev, E.stream
    cap = max(1 << 20, n * w * h * (2 if noise_every else 1))             # ~0.2 B/pixel in practice; checked below
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED)
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
        if noise_every:
            torch.cuda.synchronize()
            idx = torch.arange(first + lo, first + lo + cnt, device=dev)
            pick = (idx % noise_every) == noise_every - 1
            k = int(pick.sum().item())
            if k:
                d_rgb[:cnt][pick] = torch.randint(0, 256, (k, h, w, 3), dtype=torch.uint8, device=dev, generator=gen)
            torch.cuda.synchronize()
        ctx.encode_batch_dev(d_rgb, w * 3, 0, cnt, w, h, pkg.QBIAS_AMV, d_blob[pos:], cap - pos, t_offs, d_lens[lo:], stream)
        torch.cuda.synchronize()
        d_offs[lo:lo + cnt] = t_offs[:cnt] + pos
        pos = int(d_offs[lo + cnt - 1].item()) + int(d_lens[lo + cnt - 1].item())
        pos = (pos + 3) & ~3                  # keep every slice's base 4-byte aligned
        if pos > cap:
            raise SystemExit("synthetic stream overflowed its buffer")
    return d_blob, pos, d_offs, d_lens, int(d_lens.sum().item())   # pos: the bytes of the blob the stream occupies


def timed(E, step, steps, warmup, finish=None):
    """finish: what a pipelined step loop still owes after its last step (inside the timed region)"""
    for _ in range(warmup):
        step()
    if finish:
        finish()
    torch.cuda.synchronize()
    for c in [E.ctx] + getattr(E, "extra_ctx", []):
        c.prof_enable(True)
        c.prof_reset()
    if E.world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if finish:
        finish()
    torch.cuda.synchronize()
    if E.world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for c in [E.ctx] + getattr(E, "extra_ctx", []):
        c.prof_enable(False)
    return E.sh.max_over_ranks(elapsed, E.dev)


def kernel_times(E, ids, ctx=None, steps=1):
    """per kernel: launches in the timed region, milliseconds per STEP (a step may launch a kernel more than once: the
    decode path's fall-back rounds, the ADPCM chain's sweeps) and per launch"""
    ctx = ctx or E.ctx
    kern = {}
    for k in ids:
        launches, ms = ctx.prof_read(k)
        if launches:
            kern[ctx.kernel_name(k)] = {"launches": launches, "ms_per_step": ms / max(steps, 1), "ms_per_launch": ms / launches}
    return kern


TRAFFIC_ROUND = "r06"   # profiles/<round>_traffic*.json: this round's PMC passes (tools/profile_round.sh)


def profiled_traffic(tag, dom):
    """HBM bytes per launch of the dominant kernel from the PMC passes of this same command (tools/summarize_pmc.py ->
    profiles/r02_traffic*.json, one row per kernel and grid size; the timed batch is the largest grid) and where the
    figure comes from: it is a profile of the commit named there, not a measurement of this run."""
    path = os.path.join(ROOT, "profiles", "%s_traffic%s.json" % (TRAFFIC_ROUND, tag))
    try:
        prof = json.load(open(path))
        hits = [rec["hbm_corrected"] for kname, rec in prof["kernels"].items()
                if kname.split(" grid=")[0].split("<")[0].endswith(dom) and rec["hbm_corrected"] > 1e6]
        if hits:
            return max(hits), {"file": os.path.relpath(path, ROOT), "head": prof.get("head", ""),
                               "method": "rocprofv3 --pmc FETCH_SIZE (x2: every fabric read is a 128-byte line tallied at 64, calibrated per load shape in profiles/r06_fetch_shapes.json) + WRITE_SIZE, separate passes"}
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def profiled_path_traffic(tag, names, min_bytes=1e8):
    """the same passes, summed over every kernel of the path (the largest grid of each): what the path moves per step"""
    path = os.path.join(ROOT, "profiles", "%s_traffic%s.json" % (TRAFFIC_ROUND, tag))
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None
    total = 0.0
    for name in names:
        hits = [rec["hbm_corrected"] for kname, rec in prof["kernels"].items()
                if kname.split(" grid=")[0].split("<")[0].endswith(name) and rec["hbm_corrected"] > min_bytes]
        total += max(hits) if hits else 0.0
    return total or None


def profiled_adpcm_traffic(dom):
    """the ADPCM workload's kernels in the same kind of passes (profiles/<round>_traffic_adpcm.json): the decode kernel's
    bytes per launch, or -- the chained encode is a dozen launches -- the sum over the chain's kernels, every launch of a step"""
    path = os.path.join(ROOT, "profiles", "%s_traffic_adpcm.json" % TRAFFIC_ROUND)
    try:
        prof = json.load(open(path))
        rows = {k: v for k, v in prof["kernels"].items() if "adpcm" in k}
        if dom.startswith("amv_adpcm_decode"):
            total = max(v["hbm_corrected"] for k, v in rows.items() if "adpcm_decode" in k)
        else:
            total = sum(v["hbm_corrected"] * v.get("launches_per_step", 1) for k, v in rows.items() if "adpcm_decode" not in k)
        return total, {"file": os.path.relpath(path, ROOT), "head": prof.get("head", ""),
                       "method": "rocprofv3 --pmc FETCH_SIZE (x2: every fabric read is a 128-byte line tallied at 64, profiles/r06_fetch_shapes.json) + WRITE_SIZE, separate passes; "
                                 "one row per kernel and grid size, summed over the chain's kernels"}
    except (OSError, ValueError, KeyError):
        return None, None


def cpu_share():
    """what this process may use of the host: the CPUs of its affinity mask (what `nproc` prints) and the CPU quota of its
    cgroup (cpu.max = "quota period": a one-GPU box of this pool sees all 256 host CPUs and is given 16 CPUs' worth of
    time), read where they are -- nothing assumed"""
    avail = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if f[0] == "max" else float(f[0]) / float(f[1])
            else:
                q = float(f[0])
                quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return avail, quota


def cpu_cores():
    """threads of the CPU leg: the affinity mask, capped by the cgroup's CPU quota when there is one (more threads than
    that only take turns); AMV_BENCH_CORES overrides"""
    avail, quota = cpu_share()
    n = avail if quota is None else max(1, min(avail, int(math.ceil(quota))))
    cap = os.environ.get("AMV_BENCH_CORES")
    return max(1, min(avail, int(cap))) if cap else n


def cpu_leg(E, args):
    """does THIS rank run the CPU path beside the line?  Rank 0 does, whatever the world size: the host's cores are one
    resource, the figure is per host, and the other ranks wait for it in cpu_leg_done()'s barrier -- outside every timed
    region (timed() brackets its own with barriers)"""
    return E.rank == 0 and not args.no_cpu_baseline


def cpu_leg_done(E, args):
    if E.world > 1 and not args.no_cpu_baseline:
        dist.barrier()


def cpu_note():
    avail, quota = cpu_share()
    return {"affinity_cpus": avail, "cgroup_cpu_quota": quota, "host_cpus_online": os.cpu_count()}


def base_result(E, args, metric, unit, units_per_step_per_gpu, elapsed):
    total = E.sh.sum_over_ranks(float(units_per_step_per_gpu * args.steps), E.dev)
    return {
        "metric": metric, "value": total / elapsed, "unit": unit, "n_gpus": E.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
    }


def roofline(kern, algo_bytes, elapsed_per_step, traffic, extra=None):
    """achieved / frac: the path's algorithmic bytes per launch over the dominant kernel's time, as the bench contract
    defines them; path_achieved / path_frac: the same bytes over the whole step (every kernel of the path), the
    figure to read the path by"""
    dom = max(kern, key=lambda name: kern[name]["ms_per_step"])
    achieved = algo_bytes / (kern[dom]["ms_per_step"] * 1e-3) / 1e9
    tr, src = traffic(dom) if callable(traffic) else (traffic, None)
    path = algo_bytes / elapsed_per_step / 1e9
    r = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src,
         "algorithmic_bytes_per_launch": algo_bytes, "kernels": kern,
         "path_achieved": path, "path_frac": path / HBM_PEAK_GBS}
    if extra:
        r.update(extra)
    return r


# ---------------------------------------------------------------------------------------------
def run_decode(E, args):
    ctx, pkg, dev, stream = E.ctx, E.pkg, E.dev, E.stream
    w, h, n = args.width or 160, args.height or 120, args.frames or DECODE_FRAMES
    first = E.rank * n
    kind = getattr(args, "stream", "synthetic") or "synthetic"
    orc = entry.load_oracle()
    if kind == "amv1":
        # the reference's own clip (C-AMVDecoder/bin/AMV1.amv, 128x96, 252 frames, chunks of 1 410 - 2 924 bytes), its chunks
        # laid out back to back over and over until there are n of them: a real clip's spread of chunk sizes, device-resident
        info, amv_chunks, _ = orc.parse_amv(open(os.path.join(ROOT, "tests", "golden", "AMV1.amv"), "rb").read())
        w, h = info["width"], info["height"]
        once = np.frombuffer(b"".join(c + b"\0" * (-len(c) % 4) for c in amv_chunks), np.uint8)
        lens1 = np.array([len(c) for c in amv_chunks], np.int64)
        offs1 = np.concatenate([[0], np.cumsum((lens1 + 3) & ~3)[:-1]])
        reps = (n + len(lens1) - 1) // len(lens1)
        d_blob = torch.from_numpy(once.copy()).to(dev).repeat(reps)
        cap = int(d_blob.numel())
        d_offs = (torch.from_numpy(offs1).to(dev)[None, :] + torch.arange(reps, device=dev)[:, None] * once.size).reshape(-1)[:n].contiguous()
        d_lens = torch.from_numpy(lens1.astype(np.int32)).to(dev).repeat(reps)[:n].contiguous()
        stream_bytes = int(d_lens.sum().item())
    else:
        d_blob, cap, d_offs, d_lens, stream_bytes = make_video_stream(E, first, n, w, h, 16 if kind == "mixed" else 0)
    d_out = torch.empty((n, h, ctx.stride(w)), dtype=torch.uint8, device=dev)
    d_st = torch.empty(n, dtype=torch.int32, device=dev)

    def step():
        ctx.decode_batch_dev(d_blob, cap, d_offs, d_lens, n, w, h, 0, d_out, d_st, stream)

    # correctness gate before any timing: sample frames against the CPU oracle (checker only)
    step()
    torch.cuda.synchronize()
    if int((d_st != 0).sum().item()) != 0:
        raise SystemExit("decode reported errors on the synthetic stream")
    offs_h, lens_h = d_offs.cpu().numpy(), d_lens.cpu().numpy()
    for i in sorted({0, 1, 15, n // 2, n - 1} & set(range(n))):
        ch = d_blob[int(offs_h[i]):int(offs_h[i]) + int(lens_h[i])].cpu().numpy().tobytes()
        want, st, _ = orc.decode_frame(ch, w, h)
        if st != 0 or not (d_out[i].cpu().numpy() == want).all():
            raise SystemExit("HIP decode differs from the oracle at frame %d" % (first + i))
        if kind == "amv1":
            if ch != amv_chunks[i % len(amv_chunks)]:
                raise SystemExit("looped clip: frame %d is not the clip's chunk" % i)
        elif not (kind == "mixed" and (first + i) % 16 == 15) and ch != orc.encode_frame(orc.synth_frame(SEED, first + i, w, h), w, h):
            raise SystemExit("device-made stream differs from the oracle's encoder at frame %d" % (first + i))

    # ... and EVERY frame of the batch against a second decode of it with the other entropy kernel (the serial one-lane
    # walk, amv_huffman_kernel + dense coefficient lines): two independent routes to the same bytes, compared on the
    # device, a slice of the batch at a time so that the gate holds one slice of reference frames, not a second batch
    piece = max(1, min(n, (1 << 30) // (h * ctx.stride(w))))
    d_ref = torch.empty((piece, h, ctx.stride(w)), dtype=torch.uint8, device=dev)
    d_st2 = torch.empty(piece, dtype=torch.int32, device=dev)
    ctx.set_entropy_mode(pkg.ENTROPY_SERIAL)
    same = True
    for lo in range(0, n, piece):
        k = min(piece, n - lo)
        ctx.decode_batch_dev(d_blob, cap, d_offs[lo:lo + k], d_lens[lo:lo + k], k, w, h, 0, d_ref, d_st2, stream)
        torch.cuda.synchronize()
        same = same and torch.equal(d_ref[:k], d_out[lo:lo + k]) and int((d_st2[:k] != 0).sum().item()) == 0
    ctx.set_entropy_mode(pkg.ENTROPY_AUTO)
    if not same:
        raise SystemExit("the batch decodes differently through the serial entropy kernel")
    del d_ref, d_st2

    ids = (pkg.K_UNSTUFF, pkg.K_HUFFMAN, pkg.K_HUFFMAN_SERIAL, pkg.K_RECON)
    one_call = None
    if not args.pipelined:
        elapsed = timed(E, step, args.steps, args.warmup)
        kern = kernel_times(E, ids, steps=args.steps)
    else:
        # The step loop of a caller with the next batch in hand (a player or transcoder going through windows of frames):
        # batch k + 1 is submitted before batch k is collected, so its entropy stage runs beside batch k's reconstruction.
        # Every batch is decoded in full into buffers of its own (two sets, used in turn); K steps = K batches, the
        # pipeline's fill and drain are inside the timed region.
        few = max(3, args.steps // 4)
        t1 = timed(E, step, few, 1)                    # for comparison: the one-call form, a batch at a time
        one_call = {"ms_per_step": 1e3 * t1 / few, "steps": few, "kernels": kernel_times(E, ids, steps=few)}
        outs = [(d_out, d_st), (torch.empty_like(d_out), torch.empty_like(d_st))]
        fly = {"k": 0, "open": 0}

        def step_pipelined():
            o, s = outs[fly["k"] & 1]
            ctx.decode_submit_dev(d_blob, cap, d_offs, d_lens, n, w, h, 0, o, s, stream)
            if fly["open"]:
                ctx.decode_collect_dev(stream)         # the batch before this one
            fly["open"] = 1
            fly["k"] += 1

        def finish():
            if fly["open"]:
                ctx.decode_collect_dev(stream)
            fly["open"] = 0

        for o, s in outs:
            o.fill_(0x5A)
            s.fill_(-1)
        elapsed = timed(E, step_pipelined, args.steps, args.warmup, finish)
        kern = kernel_times(E, ids, steps=args.steps)
        # both buffer sets hold the batch, bit for bit what the one-call form (checked against the oracle above) wrote
        if int((outs[1][1] != 0).sum().item()) or int((outs[0][1] != 0).sum().item()) or not torch.equal(outs[0][0], outs[1][0]):
            raise SystemExit("pipelined decode: the two buffer sets differ")
        step()
        torch.cuda.synchronize()
        if not torch.equal(d_out, outs[1][0]):
            raise SystemExit("pipelined decode differs from the one-call form")
        del outs
    workspace = ctx.decode_workspace_per_frame()     # device bytes the context holds per frame of this batch
    ctx.entropy_stats(True)          # one extra, untimed step: how many synchronisation rounds the frames needed
    step()
    sync = ctx.entropy_stats(False)

    result = base_result(E, args, "AMV frames/sec/GPU (%dx%d decode, bit-exact)" % (w, h), "frames/s", n, elapsed)
    result["config"] = {"workload": "%dx%d AMV decode, %d-frame %s stream per GPU, chunks resident in HBM"
                                    % (w, h, n, {"synthetic": "synthetic", "mixed": "mixed (synthetic + noise)", "amv1": "AMV1.amv looped"}[kind]),
                        "frames_per_gpu": n, "mean_chunk_bytes": stream_bytes / n, "parallelism": "frame-range x%d" % E.world,
                        "per_gpu_frames_per_s": result["value"] / E.world,
                        "gate": "every frame == a second decode through the serial entropy kernel (device compare); frames 0, 1, "
                                "n/2, n-1 == the CPU oracle; their chunks == the oracle's encoder"}
    result["config"]["decode_workspace_bytes_per_frame"] = workspace
    result["config"]["handed_to_serial"] = int(sync["handed_to_serial"])   # frames of the batch the parallel entropy kernels gave up on
    result["config"]["stream"] = {"synthetic": "seeded synthetic source (BASELINE.md section 4), every frame",
                                  "mixed": "the synthetic source with every 16th frame replaced by white noise: chunk sizes from %d to %d bytes"
                                           % (int(d_lens.min().item()), int(d_lens.max().item())),
                                  "amv1": "the reference's clip AMV1.amv (128x96, 252 frames, chunks of %d - %d bytes) repeated"
                                          % (int(d_lens.min().item()), int(d_lens.max().item()))}[kind]
    if one_call:
        result["config"]["calls"] = ("amvhip_decode_submit_dev / _collect_dev, batch k+1 submitted before batch k is collected "
                                     "(entropy stage of one batch beside the reconstruction of the one before), two sets of output buffers")
        result["config"]["one_call_form"] = one_call   # amvhip_decode_batch_dev, one batch at a time, same stream
    result["roofline"] = roofline(
        kern, stream_bytes + n * 3 * w * h, elapsed / args.steps,
        (lambda dom: profiled_traffic("", dom)) if (w, h, n) == (160, 120, DECODE_FRAMES) else
        ((lambda dom: profiled_traffic("_decode320", dom)) if (w, h, n) == (320, 240, 128000) else None),
        {"entropy_sync_rounds": {"mean": sync["rounds"] / max(sync["frames"], 1), "max": sync["max_rounds"]}})

    if (w, h, n) == (160, 120, DECODE_FRAMES):
        pt = profiled_path_traffic("", ("amv_unstuff_kernel", "amv_huffman_fast_kernel", "amv_reconstruct_kernel"))
        if pt:
            result["roofline"]["path_traffic"] = pt            # all three decode kernels, counter bytes per step
            result["config"]["decode_traffic_ratio"] = pt / (stream_bytes + n * 3 * w * h)

    if E.dist and getattr(args, "strong_leg", True):   # config 4 as BASELINE.json states it, beside the weak-scaling line above
        result["config"]["scaling_modes"] = {"weak": "value / ms_per_step of this line: every GPU decodes its own %d frames" % n,
                                             "strong": "config4_strong_10k (flat: strong10k_*)"}
        guarded_strong(E, result, lambda: run_strong(E, args, w, h))
        if E.failed:
            return result       # this rank is out of step with the others: no further collective (the CPU leg's barrier)
        # ... and once more with the rank that holds the stream keeping the share that balances its own decode against the
        # others' decode + send (it pays no link for its own frames): the measurement of DESIGN section 10's table
        s = balanced_src_share(E.world, frame_bytes=h * ctx.stride(w))
        if s is not None and not os.environ.get("AMV_BENCH_SRC_SHARE"):
            guarded_strong(E, result, lambda: run_strong(E, args, w, h, share=s), flat=share_flat,
                           key="config4_strong_10k_src_share", what="strong-scaling leg, source share %.2f" % s)
            E.sh.configure(src_share=None)
            if E.failed:
                return result

    if cpu_leg(E, args):
        m = min(args.cpu_sample, n)
        blob_h = d_blob[: int(offs_h[m - 1]) + int(lens_h[m - 1]) + 16].cpu().numpy()
        o64, l32 = offs_h[:m].astype(np.uint64), lens_h[:m].astype(np.uint32)
        cores = cpu_cores()
        t = time.perf_counter()
        _, st1 = orc.decode_batch(blob_h, o64, l32, w, h, 0, threads=1)
        t1 = time.perf_counter() - t
        reps, tn = 0, 0.0
        while tn < args.cpu_seconds and reps < 64:         # bounded: a few seconds of all-core work
            t = time.perf_counter()
            orc.decode_batch(blob_h, o64, l32, w, h, 0, threads=cores)
            tn += time.perf_counter() - t
            reps += 1
        assert (st1 == 0).all()
        result["cpu_baseline"] = {"value": m * reps / tn, "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": "first %d frames of the same stream, CPU oracle (amvlib algorithm restated in C), "
                                            "frame-sharded over %d OpenMP threads, %d passes" % (m, cores, reps),
                                  "single_thread_value": m / t1, **cpu_note()}
    cpu_leg_done(E, args)
    return result


# ---------------------------------------------------------------------------------------------
def guarded(E, result, body, seconds, verdict, what):
    """One leg beside the headline, under N ranks: whatever happens to it, the line in `result` goes out -- and the run then
    ENDS NON-ZERO: a hang (a send nobody receives, a barrier a failed peer never joins) or a device / RCCL error is a
    finding, never a clean run.  verdict(status) writes the leg's flat status scalars into result["config"]
    ("ok" | "error: ..." | "hung in <phase> on rank r").  body() -> the leg's value.  An exception is recorded, E.failed set,
    the peers told (class Peers) and None returned (write_line_and_leave exits with EXIT_FAILED behind the line); no end
    within `seconds`: this rank's watchdog writes the line (rank 0) and leaves with EXIT_HUNG.  SystemExit -- a gate's
    verdict -- passes through."""
    import threading
    done = threading.Event()
    E.result, E.verdict = result, verdict
    E.strong_phase = E.strong_phase or what

    def bail():
        if done.wait(seconds):
            return
        write_line_now(E, "hung in %s on rank %d: no end after %g s" % (E.strong_phase, E.rank, seconds))
        os.write(2, ("bench.py: rank %d: %s hung in phase %r; line written, leaving with code %d\n"
                     % (E.rank, what, E.strong_phase, EXIT_HUNG)).encode())
        leave(EXIT_HUNG)

    threading.Thread(target=bail, daemon=True).start()
    value = None
    try:
        value = body()
        status = "ok"
    except Exception as e:                       # (SystemExit is not an Exception)
        status = "error: %s in %s: %s" % (type(e).__name__, E.strong_phase, str(e)[:300])
        E.failed = status                        # the line goes out, then write_line_and_leave() leaves non-zero
    done.set()
    verdict(status)
    E.verdict = None
    E.strong_phase = ""
    return value


def guarded_strong(E, result, body, seconds=None, flat=None, key="config4_strong_10k", what="strong-scaling leg"):
    """The exchange has never met more than one GPU: configs[3] as stated under guarded()'s contract.
    The verdict is a flat scalar, strong10k_status = "ok" | "error: ..." | "hung in <phase> on rank r" (flat: the function
    that makes the leg's flat scalars, strong_flat unless given; key: where the leg's dictionary goes)."""
    E.strong_phase = "setup"
    box = {}
    flat = flat or strong_flat

    def verdict(status):
        result["config"].update(flat(box.get("strong") if status == "ok" else None, status))

    def run():
        box["strong"] = body()
        return box["strong"]

    strong = guarded(E, result, run, STRONG_LEG_SECONDS if seconds is None else seconds, verdict, what)
    result["config"][key] = strong if strong is not None else {"failed": E.failed}


def write_line_and_leave(E, result):
    """rank 0 writes the one JSON line; a run with a failed leg then leaves at once with EXIT_FAILED (the line is out, the
    peers are told; the communicator may be in no state to be torn down).  Returns normally otherwise."""
    if E.dist:
        result["config"].setdefault("run_status", E.failed or "ok")     # every leg beside the headline, in one scalar
    sys.stdout.flush()
    emit_line(E, result)
    if E.failed:
        os.write(2, ("bench.py: rank %d: %s -- leaving with code %d\n" % (E.rank, E.failed, EXIT_FAILED)).encode())
        if E.peers:
            E.peers.raise_flag(E.failed)
        leave(EXIT_FAILED)


def run_strong(E, args, w, h, n_total=10000, share=None):
    """BASELINE.json configs[3] as it is stated: ONE 10 000-frame 160x120 stream, held by rank 0 in HBM, frame-sharded
    over the ranks: scatter-v of the chunks over RCCL, per-rank decode through the C ABI, gather of the BGR frames back
    to rank 0 -- timed end to end ("scaling": "strong").  Runs next to the weak-scaling line, never instead of it."""
    ctx, dev, stream, sh = E.ctx, E.dev, E.stream, E.sh
    d_blob = d_offs = d_lens = None
    # AMV_BENCH_SRC_SHARE: the fraction of the stream rank 0 -- which holds it and pays no link for its own frames -- keeps
    # (sharding.configure; default: equal ranges).  Unmeasured on more than one GPU: a knob, not a claim.  share=: the second
    # leg's, from balanced_src_share.
    share = share if share is not None else (os.environ.get("AMV_BENCH_SRC_SHARE") or None)
    sh.configure(src_share=float(share) if share is not None else None)
    E.strong_phase = "making the stream"
    if E.rank == 0:
        d_blob, cap, d_offs, d_lens, stream_bytes = make_video_stream(E, 0, n_total, w, h)
    maxf = max(hi - lo for lo, hi in (sh.frame_range(n_total, r, E.world) for r in range(E.world)))
    shape = (h, ctx.stride(w))
    # rank 0 decodes straight into its slice of the gathered buffer (`into`), the others into a buffer of their own that
    # the send reads; the gathered buffer is allocated once and handed back in every step (out=)
    d_out = torch.empty((maxf,) + shape, dtype=torch.uint8, device=dev) if E.rank != 0 else None
    d_full = torch.empty((n_total,) + shape, dtype=torch.uint8, device=dev) if E.rank == 0 else None
    d_st = torch.empty(maxf, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    first_of = sh.frame_range(n_total, E.rank, E.world)[0]
    # sub-batches: the send of one posted while the next decodes pays when the decode is bound by throughput; a rank's
    # share of THIS stream (1 250 frames at 8 ranks) is one wave deep, and two launches of half the frames take twice as long
    k = 2 if maxf >= 16384 else 1

    def decode(my_blob, my_offs, my_lens, first, into=None):
        cnt = int(my_lens.numel())
        dst = into if into is not None else d_out[first - first_of: first - first_of + cnt]
        if cnt:
            ctx.decode_batch_dev(my_blob, int(my_blob.numel()), my_offs, my_lens, cnt, w, h, 0, dst, d_st, stream)
            bad.add_((d_st[:cnt] != 0).sum())
        return dst

    def clock():
        torch.cuda.synchronize()
        return time.perf_counter()

    def phase_clock():
        # between the phases of a step: what THIS rank's stream has been given (the scatter's and the gather's transfers are
        # waited for on it), not the device -- rank 0's receives are posted before its own decode and are still in flight,
        # on RCCL's stream, when that decode ends
        torch.cuda.current_stream().synchronize()
        return time.perf_counter()

    def one_step(timed_clock=None):
        return sh.strong_step(d_blob, d_offs, d_lens, n_total, dev, decode, clock=timed_clock, frame_shape=shape, k=k, out=d_full)

    # gate: the gathered frames equal one GPU decoding the whole stream by itself, and the oracle on a sample
    E.strong_phase = "first exchange (scatter / decode / gather)"
    full, _ = one_step()
    torch.cuda.synchronize()
    ok = 1
    if E.rank == 0:
        ref = torch.empty((n_total, h, ctx.stride(w)), dtype=torch.uint8, device=dev)
        st = torch.empty(n_total, dtype=torch.int32, device=dev)
        ctx.decode_batch_dev(d_blob, cap, d_offs, d_lens, n_total, w, h, 0, ref, st, stream)
        torch.cuda.synchronize()
        ok = int(torch.equal(full, ref) and int((st != 0).sum().item()) == 0)
        orc = entry.load_oracle()
        offs_h, lens_h = d_offs.cpu().numpy(), d_lens.cpu().numpy()
        for i in sorted({0, min(n_total // E.world, n_total - 1), n_total // 2, n_total - 1}):
            ch = d_blob[int(offs_h[i]):int(offs_h[i]) + int(lens_h[i])].cpu().numpy().tobytes()
            ok &= int((full[i].cpu().numpy() == orc.decode_frame(ch, w, h)[0]).all())
        del ref
    E.strong_phase = "gate all-reduce"
    ok = int(sh.sum_over_ranks(float(ok), dev)) == E.world and int(sh.sum_over_ranks(float(bad.item()), dev)) == 0
    if not ok:
        raise SystemExit("strong-scaling leg: gathered frames differ from the single-GPU decode / the oracle")
    del full

    steps = max(3, min(args.steps, 10))
    phases = {"scatter": 0.0, "decode": 0.0, "gather": 0.0}
    E.strong_phase = "timed exchange"
    for _ in range(2):
        one_step()
    dist.barrier()
    t0 = clock()
    for _ in range(steps):
        _, ph = one_step(phase_clock)
        for name in phases:
            phases[name] += ph[name]
    dist.barrier()
    elapsed = sh.max_over_ranks(clock() - t0, dev)
    phases = {name: sh.max_over_ranks(v, dev) / steps * 1e3 for name, v in phases.items()}
    return {"scaling": "strong", "workload": "one %d-frame %dx%d stream on rank 0 -> scatter-v of chunks (RCCL) -> per-rank decode "
                                            "-> gather of BGR frames to rank 0, end to end" % (n_total, w, h),
            "frames": n_total, "steps": steps, "ms_per_step": elapsed / steps * 1e3, "frames_per_s": n_total * steps / elapsed,
            "phase_ms_max_over_ranks": phases, "backend": dist.get_backend(), "n_gpus": E.world, "sub_batches": k,
            "rccl_ranks": dist.get_world_size(),          # the communicator's size as the backend reports it
            "src_share": float(share) if share is not None else None, "frames_by_rank": [b - a for a, b in (sh.frame_range(n_total, r, E.world) for r in range(E.world))],
            "exchange": "source sends slices of its blob (point to point, one grouped call), frames are received straight into "
                        "slices of one buffer on rank 0, whose own range is decoded in place",
            "gathered_bytes_per_step": n_total * h * ctx.stride(w)}


# ---------------------------------------------------------------------------------------------
def run_encode(E, args):
    ctx, pkg, dev, stream = E.ctx, E.pkg, E.dev, E.stream
    w, h, n = args.width or 320, args.height or 240, args.frames or 8000
    if args.psnr_floor is None:
        args.psnr_floor = 26.6 if (w, h) == (320, 240) else 25.4
    first = E.rank * n
    d_rgb = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
    ctx.synth_frames_dev(SEED, first, n, w, h, d_rgb, stream)
    cap = max(1 << 20, n * w * h)
    d_blob = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_offs = torch.zeros(n, dtype=torch.int64, device=dev)
    d_lens = torch.zeros(n, dtype=torch.int32, device=dev)

    def step():
        ctx.encode_batch_dev(d_rgb, w * 3, 0, n, w, h, pkg.QBIAS_AMV, d_blob, cap, d_offs, d_lens, stream)

    # gate: chunks byte-identical to the oracle's encoder on sample frames; whole batch round-trips through
    # the bit-exact decoder with PSNR against the source above the stated floor
    step()
    torch.cuda.synchronize()
    stream_bytes = int(d_lens.sum().item())
    if int(d_offs[-1].item()) + int(d_lens[-1].item()) > cap:
        raise SystemExit("encoded stream overflowed its buffer")
    orc = entry.load_oracle()
    offs_h, lens_h = d_offs.cpu().numpy(), d_lens.cpu().numpy()
    for i in sorted({0, 1, n // 2, n - 1}):
        ch = d_blob[int(offs_h[i]):int(offs_h[i]) + int(lens_h[i])].cpu().numpy().tobytes()
        if ch != orc.encode_frame(d_rgb[i].cpu().numpy(), w, h):
            raise SystemExit("HIP encode differs from the oracle's encoder at frame %d" % (first + i))
    d_out = torch.empty((n, h, ctx.stride(w)), dtype=torch.uint8, device=dev)
    d_st = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.decode_batch_dev(d_blob, cap, d_offs, d_lens, n, w, h, 0, d_out, d_st, stream)
    torch.cuda.synchronize()
    if int((d_st != 0).sum().item()) != 0:
        raise SystemExit("round trip: the decoder rejected an encoded frame")
    sq = 0.0
    for lo in range(0, n, 256):                # decoder output is BGR, rows padded to the stride
        a = d_out[lo:lo + 256, :, : w * 3].reshape(-1, h, w, 3).flip(-1).to(torch.int32)
        sq += float(((a - d_rgb[lo:lo + 256].to(torch.int32)) ** 2).sum().item())
    psnr = 10.0 * math.log10(255.0 ** 2 / (sq / (n * h * w * 3)))
    one = orc.psnr(d_out[0, :, : w * 3].reshape(h, w, 3).flip(-1).cpu().numpy(), d_rgb[0].cpu().numpy())
    if psnr < args.psnr_floor:
        raise SystemExit("round-trip PSNR %.2f dB is below the %.1f dB floor" % (psnr, args.psnr_floor))

    elapsed = timed(E, step, args.steps, args.warmup)
    kern = kernel_times(E, (pkg.K_FDCT, pkg.K_PACK, pkg.K_PACK_SERIAL, pkg.K_COMPACT), steps=args.steps)
    result = base_result(E, args, "AMV frames/sec/GPU (%dx%d encode, PSNR-checked round trip)" % (w, h), "frames/s", n, elapsed)
    result["config"] = {"workload": "%dx%d AMV encode (rgb24 -> yuvj420p, fdct, quantise, Huffman), %d synthetic frames per GPU "
                                    "resident in HBM" % (w, h, n),
                        "frames_per_gpu": n, "mean_chunk_bytes": stream_bytes / n, "parallelism": "frame-range x%d" % E.world,
                        "round_trip_psnr_db": psnr, "psnr_floor_db": args.psnr_floor, "frame0_psnr_db_cpu_checker": one,
                        "bit_exact_vs_cpu_encoder": True}
    result["roofline"] = roofline(kern, stream_bytes + n * 3 * w * h, elapsed / args.steps,
                                  (lambda dom: profiled_traffic("_encode", dom)) if (w, h, n) == (320, 240, 8000) else None)

    if cpu_leg(E, args):
        import concurrent.futures
        m = min(256, n)
        src = d_rgb[:m].cpu().numpy()
        t = time.perf_counter()
        for i in range(m):
            orc.encode_frame(src[i], w, h)
        t1 = time.perf_counter() - t
        cores = cpu_cores()
        with concurrent.futures.ThreadPoolExecutor(cores) as ex:      # the ctypes call drops the interpreter lock
            t = time.perf_counter()
            list(ex.map(lambda i: orc.encode_frame(src[i % m], w, h), range(4 * m)))
            tn = time.perf_counter() - t
        result["cpu_baseline"] = {"value": 4 * m / tn, "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": "first %d frames of the same source x4, CPU oracle (the reference build's encoder "
                                            "algorithm restated in C), one frame per call over %d threads" % (m, cores),
                                  "single_thread_value": m / t1}
    cpu_leg_done(E, args)
    return result


# ---------------------------------------------------------------------------------------------
def make_audio(E, first_chunk, n):
    """n audio chunks of SAMPLES_PER_FRAME samples: pcm, pcm_offs, nsamp, chunk offs/lens"""
    dev = E.dev
    spf = SAMPLES_PER_FRAME
    d_pcm = torch.empty(n * spf, dtype=torch.int16, device=dev)
    E.ctx.synth_audio_dev(SEED, first_chunk * spf, n * spf, d_pcm, E.stream)
    d_pcm_offs = torch.arange(n, dtype=torch.int64, device=dev) * spf
    d_nsamp = torch.full((n,), spf, dtype=torch.int32, device=dev)
    clen = 8 + spf // 2
    d_offs = torch.arange(n, dtype=torch.int64, device=dev) * clen
    d_lens = torch.full((n,), clen, dtype=torch.int32, device=dev)
    d_chunks = torch.zeros(n * clen + 16, dtype=torch.uint8, device=dev)
    return d_pcm, d_pcm_offs, d_nsamp, d_chunks, d_offs, d_lens, clen


def audio_gate(E, A, n):
    """encode on the device == oracle's encoder (index carried) on the first chunks, and EVERY chunk == a second encode through
    the exhaustive 89-start route (other kernels, nothing guessed; device compare); decode on the device == oracle's decoder"""
    d_pcm, d_pcm_offs, d_nsamp, d_chunks, d_offs, d_lens, clen = A
    orc = entry.load_oracle()
    spf = SAMPLES_PER_FRAME
    old = os.environ.get("AMVHIP_ADPCM_SWEEPS")
    os.environ["AMVHIP_ADPCM_SWEEPS"] = "map"          # (read when a context is made)
    try:
        ref = E.pkg.Context(E.dev.index)
    finally:
        if old is None:
            os.environ.pop("AMVHIP_ADPCM_SWEEPS", None)
        else:
            os.environ["AMVHIP_ADPCM_SWEEPS"] = old
    try:
        d_want = torch.zeros_like(d_chunks)
        ref.adpcm_encode_batch_dev(d_pcm, d_pcm_offs, d_nsamp, n, None, d_want, d_offs, E.stream)
        torch.cuda.synchronize()
        if not torch.equal(d_want[: n * clen], d_chunks[: n * clen]):
            raise SystemExit("HIP ADPCM encode: the chained route and the exhaustive route differ")
        del d_want
    finally:
        ref.close()
    m = min(n, 64)
    pcm = d_pcm[: m * spf].cpu().numpy()
    got = d_chunks[: m * clen].cpu().numpy()
    idx = 0
    for i in range(m):
        want, idx = orc.adpcm_encode_chunk(pcm[i * spf:(i + 1) * spf], idx)
        if got[i * clen:(i + 1) * clen].tobytes() != want:
            raise SystemExit("HIP ADPCM encode differs from the oracle at chunk %d" % i)
    return orc


def run_adpcm(E, args, with_video=False):
    ctx, pkg, dev, stream = E.ctx, E.pkg, E.dev, E.stream
    spf = SAMPLES_PER_FRAME
    if with_video:
        w, h, n = args.width or 320, args.height or 240, args.frames or 16000
        first = E.rank * n
        d_blob, cap, d_voffs, d_vlens, stream_bytes = make_video_stream(E, first, n, w, h)
        d_out = torch.empty((n, h, ctx.stride(w)), dtype=torch.uint8, device=dev)
        d_st = torch.empty(n, dtype=torch.int32, device=dev)
        na = n
    else:
        na = args.frames or 200000
        first = E.rank * na
    A = make_audio(E, first, na)
    d_pcm, d_pcm_offs, d_nsamp, d_chunks, d_offs, d_lens, clen = A
    d_pcm2 = torch.zeros(na * spf + 8, dtype=torch.int16, device=dev)
    torch.cuda.synchronize()                   # the sources were made on the main stream
    side = torch.cuda.Stream(device=dev) if with_video else None
    astream = side.cuda_stream if with_video else stream
    actx = pkg.Context(dev.index) if with_video else ctx    # one context per stream that is meant to overlap
    if with_video:
        E.extra_ctx = [actx]

    def audio():
        actx.adpcm_encode_batch_dev(d_pcm, d_pcm_offs, d_nsamp, na, None, d_chunks, d_offs, astream)
        actx.adpcm_decode_batch_dev(d_chunks, na * clen, d_offs, d_lens, na, d_pcm2, d_pcm_offs, None, astream)

    def step():
        if with_video:
            ctx.decode_batch_dev(d_blob, cap, d_voffs, d_vlens, n, w, h, 0, d_out, d_st, stream)
        audio()

    step()
    torch.cuda.synchronize()
    orc = audio_gate(E, A, na)
    got = d_pcm2[: spf * min(na, 64)].cpu().numpy()
    chunks_h = d_chunks[: clen * min(na, 64)].cpu().numpy()
    for i in range(min(na, 64)):
        want, _ = orc.adpcm_decode_chunk(chunks_h[i * clen:(i + 1) * clen])
        if not (got[i * spf:(i + 1) * spf] == want[:spf]).all():
            raise SystemExit("HIP ADPCM decode differs from the oracle at chunk %d" % i)
    if with_video:
        if int((d_st != 0).sum().item()) != 0:
            raise SystemExit("decode reported errors on the synthetic stream")
        offs_h, lens_h = d_voffs.cpu().numpy(), d_vlens.cpu().numpy()
        for i in sorted({0, n - 1}):
            ch = d_blob[int(offs_h[i]):int(offs_h[i]) + int(lens_h[i])].cpu().numpy().tobytes()
            want, st, _ = orc.decode_frame(ch, w, h)
            if st != 0 or not (d_out[i].cpu().numpy() == want).all():
                raise SystemExit("HIP decode differs from the oracle at frame %d" % (first + i))

    elapsed = timed(E, step, args.steps, args.warmup)
    audio_bytes = na * (clen + 2 * spf)
    if with_video:
        kern = kernel_times(E, (pkg.K_UNSTUFF, pkg.K_HUFFMAN, pkg.K_HUFFMAN_SERIAL, pkg.K_RECON), steps=args.steps)
        audio_kern = kernel_times(E, (pkg.K_ADPCM_DEC, pkg.K_ADPCM_ENC), actx, steps=args.steps)   # on the second stream, overlapped with the video kernels
        result = base_result(E, args, "AMV frames/sec/GPU (%dx%d decode with co-resident IMA-ADPCM, bit-exact)" % (w, h),
                             "frames/s", n, elapsed)
        result["config"] = {"workload": "%dx%d AMV decode of %d frames per GPU on one HIP stream, IMA-ADPCM encode + decode of "
                                        "the frames' %d-sample audio chunks on a second stream" % (w, h, n, spf),
                            "frames_per_gpu": n, "mean_chunk_bytes": stream_bytes / n, "parallelism": "frame-range x%d" % E.world,
                            "adpcm_samples_per_s": 2.0 * na * spf * args.steps * E.world / elapsed}
        result["roofline"] = roofline(kern, stream_bytes + n * 3 * w * h, elapsed / args.steps, None,
                                      {"co_resident_audio_kernels": audio_kern, "audio_algorithmic_bytes_per_step": audio_bytes})
    else:
        kern = kernel_times(E, (pkg.K_ADPCM_DEC, pkg.K_ADPCM_ENC), steps=args.steps)
        result = base_result(E, args, "IMA-ADPCM samples/sec/GPU (encode + decode, bit-exact)", "samples/s", 2 * na * spf, elapsed)
        result["dtype"] = "int32"
        result["config"] = {"workload": "%d AMV audio chunks of %d samples per GPU: PCM -> ADPCM (step index carried through "
                                        "the stream, as the reference encoder does) -> PCM" % (na, spf),
                            "chunks_per_gpu": na, "parallelism": "chunk-range x%d" % E.world}
        result["roofline"] = roofline(kern, audio_bytes, elapsed / args.steps, profiled_adpcm_traffic if na == 200000 else None)

    if cpu_leg(E, args) and not with_video:
        m = min(na, 4096)
        pcm = d_pcm[: m * spf].cpu().numpy()
        ch = d_chunks[: m * clen].cpu().numpy()
        t = time.perf_counter()
        idx = 0
        for i in range(m):
            _, idx = orc.adpcm_encode_chunk(pcm[i * spf:(i + 1) * spf], idx)
            orc.adpcm_decode_chunk(ch[i * clen:(i + 1) * clen])
        t1 = time.perf_counter() - t
        result["cpu_baseline"] = {"value": 2 * m * spf / t1, "unit": "samples/s", "cores": 1, "kind": "port",
                                  "sample": "first %d chunks of the same audio, CPU oracle encode + decode, one thread "
                                            "(includes the ctypes call per chunk)" % m}
    if cpu_leg(E, args) and with_video:
        # the CPU path for the same unit of work: one 320x240 frame decoded + its audio chunk encoded and decoded
        m = min(n, args.cpu_sample, 512)
        blob_h = d_blob[: int(offs_h[m - 1]) + int(lens_h[m - 1]) + 16].cpu().numpy()
        o64, l32 = offs_h[:m].astype(np.uint64), lens_h[:m].astype(np.uint32)
        pcm = d_pcm[: m * spf].cpu().numpy()
        ch = d_chunks[: m * clen].cpu().numpy()
        cores = cpu_cores()
        t = time.perf_counter()
        orc.decode_batch(blob_h, o64, l32, w, h, 0, threads=1)
        tv1 = time.perf_counter() - t
        t = time.perf_counter()
        orc.decode_batch(blob_h, o64, l32, w, h, 0, threads=cores)
        tvn = time.perf_counter() - t
        t = time.perf_counter()
        idx = 0
        for i in range(m):
            _, idx = orc.adpcm_encode_chunk(pcm[i * spf:(i + 1) * spf], idx)
            orc.adpcm_decode_chunk(ch[i * clen:(i + 1) * clen])
        ta = time.perf_counter() - t
        result["cpu_baseline"] = {"value": m / (tvn + ta), "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": "first %d frames: CPU oracle video decode frame-sharded over %d threads, then the "
                                            "frames' audio chunks encoded (index carried: a serial chain) and decoded on one "
                                            "thread" % (m, cores),
                                  "single_thread_value": m / (tv1 + ta)}
    cpu_leg_done(E, args)
    if with_video:
        E.extra_ctx = []
        actx.close()
    return result


# ---------------------------------------------------------------------------------------------
def run_amvlib(E, args):
    """The drop-in surface itself: the player loop of the reference (AMVDecoderDlg.cpp FillBuffer / AmvLibTest.cpp) --
    AmvReadNextFrame, AmvVideoDecode, AmvAudioDecode, one frame per call -- run by a plain C host
    (tests/c/amvlib_host.c, compiled here with gcc against include/amvhip.h) over the reference's own clip and over a
    muxed synthetic 160x120 file; beside it the CPU oracle decoding the same chunks on one thread."""
    import subprocess
    import tempfile
    pkg, ctx, dev = E.pkg, E.ctx, E.dev
    orc = entry.load_oracle()
    lib = pkg.load_library()
    tmp = tempfile.mkdtemp(prefix="amvlib_bench_")
    exe = os.path.join(tmp, "amvlib_host")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "amvlib_host.c"),
                    "-L", libdir, "-l:" + os.path.basename(pkg.LIB_PATH), "-Wl,-rpath," + libdir, "-o", exe], check=True)
    # the synthetic file: device-made 160x120 chunks + the audio that travels with them, through the library's muxer
    w, h, n, spf = 160, 120, args.frames or 4000, SAMPLES_PER_FRAME
    d_blob, cap, d_offs, d_lens, _ = make_video_stream(E, 0, n, w, h)
    A = make_audio(E, 0, n)
    d_pcm, d_pcm_offs, d_nsamp, d_chunks, d_aoffs, d_alens, clen = A
    ctx.adpcm_encode_batch_dev(d_pcm, d_pcm_offs, d_nsamp, n, None, d_chunks, d_aoffs, E.stream)
    torch.cuda.synchronize()
    blob, offs, lens, ach = d_blob.cpu().numpy(), d_offs.cpu().numpy(), d_lens.cpu().numpy(), d_chunks.cpu().numpy()
    synth = os.path.join(tmp, "synth160.amv")
    m = lib.amvhip_mux_open(synth.encode(), w, h, 16, 22050, 200000, 64000)
    for i in range(n):
        v = blob[int(offs[i]):int(offs[i]) + int(lens[i])]
        a = ach[i * clen:(i + 1) * clen]
        assert lib.amvhip_mux_write_frame(m, v.ctypes.data, v.size, a.ctypes.data, a.size) == 0
    assert lib.amvhip_mux_close(m) == 0

    def play(path, passes, readahead=None):
        env = dict(os.environ)
        if readahead is not None:
            env["AMVHIP_READAHEAD"] = str(readahead)
        out = subprocess.run([exe, path, "--bench", str(passes)], check=True, capture_output=True, text=True, env=env).stdout
        f = dict(line.split(": ", 1) for line in out.strip().splitlines())
        return int(f["bench frames"]) / float(f["bench seconds"])

    amv1 = os.path.join(ROOT, "tests", "golden", "AMV1.amv")
    rates = {"AMV1.amv 128x96 (252 frames x 40 passes)": play(amv1, 40),
             "synthetic 160x120 (%d frames x 4 passes)" % n: play(synth, 4),
             "synthetic 160x120, AMVHIP_READAHEAD=1 (one frame per GPU round trip)": play(synth, 1, readahead=1) if n <= 4000 else None}
    # gate: the file decodes to the oracle's frames through the same surface (first frames; the tests cover the rest)
    amv = lib.AmvOpen(synth.encode())
    for i in range(3):
        assert lib.AmvReadNextFrame(amv) == 0 and lib.AmvVideoDecode(amv) == 0
        got = np.frombuffer(__import__("ctypes").string_at(amv.contents.videobuf.fbmpdat, w * h * 3), np.uint8)
        ch = blob[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes()
        if not (got == orc.decode_frame(ch, w, h)[0].ravel()).all():
            raise SystemExit("amvlib surface differs from the oracle at frame %d" % i)
    lib.AmvClose(amv)

    value = rates["synthetic 160x120 (%d frames x 4 passes)" % n]
    result = {"metric": "AMV frames/sec through the amvlib call surface (AmvReadNextFrame + AmvVideoDecode + AmvAudioDecode, "
                        "160x120, bit-exact)", "value": value, "unit": "frames/s", "n_gpus": 1, "steps": 4, "warmup": 1,
              "ms_per_step": n / value * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
              "data": "synthetic", "config": {"workload": "amvlib player loop over a muxed %d-frame 160x120 AMV file, one frame per "
                                                          "call, host buffers (PCIe inclusive), read-ahead windows behind the calls" % n,
                                              "frames_per_s": rates}}
    m = min(n, 1024)
    t = time.perf_counter()
    orc.decode_batch(blob, offs[:m].astype(np.uint64), lens[:m].astype(np.uint32), w, h, 0, threads=1)
    for i in range(m):
        orc.adpcm_decode_chunk(ach[i * clen:(i + 1) * clen])
    t1 = time.perf_counter() - t
    result["cpu_baseline"] = {"value": m / t1, "unit": "frames/s", "cores": 1, "kind": "port",
                              "sample": "first %d frames of the same file: CPU oracle video + audio decode, one thread (the reference's "
                                        "own loop is single-threaded)" % m}
    return result


def run_secondary(E, args, result):
    """The other BASELINE.json configs beside the headline, each with its own gate and timed loop (fewer steps), so that
    the one line the driver records carries all five: 320x240 decode, the 10 000-frame 160x120 stream (configs[3]'s
    stream on one GPU), 320x240 encode (configs[2]), video decode with co-resident ADPCM (configs[4]), ADPCM alone.
    Under N ranks every rank runs every leg on frames of its own (weak, like the headline: rates summed over the ranks,
    times the slowest rank's -- base_result / timed do that for any world size), rank 0 runs the CPU path beside each while
    the others wait in a barrier, and each leg is guarded (guarded()): an error or a hang of one is `run_status` on the line
    and a non-zero exit, and the legs after it do not run (this rank is out of step with its peers).  Every finished leg is
    flattened into result["config"] at once, so that a line written early carries what there is."""
    import copy
    out = {}
    result["config"]["secondary"] = out
    # batch sizes: what fills the chip with ONE entropy lane per frame (320x240: 32 000 / 64 000 / 128 000 frames per step
    # decode at 5.9 / 6.6 / 7.7 M frames/s -- a small batch is cut into several speculative lanes per frame); the
    # 10 000-frame line is there to show the other regime
    plan = (("decode_320x240", run_decode, {"width": 320, "height": 240, "frames": 128000}),
            ("decode_160x120_10k_stream", run_decode, {"frames": 10000}),
            ("decode_160x120_mixed", run_decode, {"stream": "mixed"}),
            ("decode_amv1_looped", run_decode, {"stream": "amv1", "frames": 200000}),
            ("encode_320x240", run_encode, {}),
            ("coresident_320x240_adpcm", lambda e, a: run_adpcm(e, a, with_video=True), {"frames": 64000}),
            ("adpcm", lambda e, a: run_adpcm(e, a, with_video=False), {}))

    def verdict(status):
        if status != "ok":
            result["config"]["run_status"] = status

    for name, fn, over in plan:
        a = copy.copy(args)
        a.steps, a.warmup = min(args.steps, 10), min(args.warmup, 2)
        a.cpu_sample, a.cpu_seconds = 512, 1.0     # bounded CPU legs: every entry carries the CPU path beside it
        a.strong_leg = False                       # configs[3] as stated rides on the headline only
        for k, v in over.items():
            setattr(a, k, v)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        if E.dist:
            E.strong_phase = "leg " + name
            r = guarded(E, result, lambda: fn(E, a), LEG_SECONDS, verdict, "leg " + name)
            E.extra_ctx = []
            if r is None:
                break
        else:
            try:
                r = fn(E, a)
            except torch.OutOfMemoryError as e:    # a smaller part than the 288 GB this is sized for: say so, keep the line
                E.extra_ctx = []
                out[name] = {"skipped": "out of device memory: %s" % str(e).splitlines()[0]}
                result["config"].update(flat_secondary({name: out[name]}))
                continue
            E.extra_ctx = []
        roof = r["roofline"]
        entry_ = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "steps": r["steps"], "ms_per_step": r["ms_per_step"],
                  "workload": r["config"]["workload"], "kernels": roof["kernels"],
                  "roofline": {k: roof[k] for k in ("kernel", "achieved", "frac", "path_achieved", "path_frac", "traffic", "traffic_source",
                                                    "algorithmic_bytes_per_launch")},
                  "cpu_baseline": r.get("cpu_baseline"), "wall_s": None}
        for k in ("mean_chunk_bytes", "round_trip_psnr_db", "psnr_floor_db", "bit_exact_vs_cpu_encoder", "adpcm_samples_per_s", "gate",
                  "handed_to_serial", "stream"):
            if k in r["config"]:
                entry_[k] = r["config"][k]
        if "co_resident_audio_kernels" in roof:
            entry_["co_resident_audio_kernels"] = roof["co_resident_audio_kernels"]
        entry_["wall_s"] = time.perf_counter() - t0
        out[name] = entry_
        result["config"].update(flat_secondary({name: entry_}))
    result["config"]["secondary"] = result["config"].pop("secondary")     # the nested form behind the flat scalars
    return out


def flat_secondary(sec):
    """the secondary entries once more as flat scalars of `config`: a record that keeps scalars only still carries every
    BASELINE config's rate, roofline fraction and the CPU path beside it (CONFIG_HEAD orders them; cores, single-thread
    figures and path fractions stay in config.secondary)"""
    names = {"decode_320x240": "c320_decode", "decode_160x120_10k_stream": "stream10k", "encode_320x240": "enc320",
             "coresident_320x240_adpcm": "coresident", "adpcm": "adpcm", "decode_160x120_mixed": "mixed160",
             "decode_amv1_looped": "amv1"}
    flat = {}
    for name, e in sec.items():
        p = names.get(name, name)
        if "skipped" in e:
            flat[p + "_skipped"] = e["skipped"]
            continue
        unit = "sps" if e["unit"].startswith("samples") else "fps"
        flat["%s_%s" % (p, unit)] = e["value"]
        flat[p + "_ms"] = e["ms_per_step"]
        flat[p + "_frac"] = e["roofline"]["frac"]
        cb = e.get("cpu_baseline")
        if cb:
            flat["%s_cpu_%s" % (p, unit)] = cb["value"]
        if "handed_to_serial" in e:
            flat[p + "_handed_to_serial"] = e["handed_to_serial"]
        if "adpcm_samples_per_s" in e:
            flat[p + "_adpcm_sps"] = e["adpcm_samples_per_s"]
    return flat


# What a record that keeps only the first scalars of `config` must still carry, in this order: the workload, then per
# BASELINE config its rate, its roofline fraction and the CPU path beside it (configs[1] is the line itself: value, roofline,
# cpu_baseline), then the headline's own diagnostics.
CONFIG_HEAD = ("workload", "parallelism",
               "c320_decode_fps", "c320_decode_frac", "c320_decode_cpu_fps",
               "enc320_fps", "enc320_frac", "enc320_cpu_fps",
               "stream10k_fps", "stream10k_frac", "stream10k_cpu_fps",
               "coresident_fps", "coresident_frac", "coresident_cpu_fps", "coresident_adpcm_sps",
               "adpcm_sps", "adpcm_frac", "adpcm_cpu_sps",
               "decode_traffic_ratio", "handed_to_serial", "mixed160_fps", "mixed160_handed_to_serial",
               "frames_per_gpu", "mean_chunk_bytes")
# ... and the line of a run with ranks (a communicator: WORLD_SIZE > 1, or --strong): the communicator's size and the verdicts
# first (the strong leg's and, in one scalar, every other leg's), configs[3] as stated, then the same sixteen -- all five
# BASELINE configs inside the first 24 -- and the strong leg's phases behind them.
CONFIG_HEAD_RANKS = ("workload", "rccl_ranks", "run_status", "strong10k_status", "strong10k_ms", "strong10k_fps") + CONFIG_HEAD[2:18] + (
    "strong10k_decode_ms", "strong10k_gather_ms", "strong10k_scatter_ms", "strong10k_srcshare", "strong10k_srcshare_status",
    "strong10k_srcshare_ms", "strong10k_srcshare_fps", "parallelism") + CONFIG_HEAD[18:]


def ordered_config(cfg):
    """`cfg` with the head's keys first (those that exist), everything else behind them in the order it came"""
    order = CONFIG_HEAD_RANKS if "rccl_ranks" in cfg or "strong10k_status" in cfg else CONFIG_HEAD
    head = {k: cfg[k] for k in order if k in cfg}
    head.update((k, v) for k, v in cfg.items() if k not in head)
    return head


# ---------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks -- one CHILD process per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run sets them -- relay rank 0's one JSON line
    and leave with the worst exit code.  Like torch.distributed.run's agent, THIS process hosts the rendezvous store (a
    TCPStore on a port the system picks: nothing is probed and re-bound later, and the store -- the ranks' side channel,
    class Peers -- outlives any rank) and the ranks join it as clients (TORCHELASTIC_USE_AGENT_STORE).  This process never
    touches the GPU (counting devices does not initialise HIP on this image) and never re-execs itself.  The whole run has
    a wall-clock limit (AMV_BENCH_LAUNCH_LIMIT): ranks that hang together -- in the rendezvous, in a barrier -- are ended by
    their PIDs and the run leaves with EXIT_HUNG.  A run whose rank 0 printed no line is not clean, whatever the exit codes.
    AMV_BENCH_CHILD replaces the child command (the launcher's test: tests/test_bench_launcher.py)."""
    import shlex
    import subprocess
    import threading
    n = args.gpus
    if not os.environ.get("AMV_BENCH_CHILD") and not ONE_DEVICE:
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d: %d devices needed, %d visible -- not measuring fewer GPUs under that name\n" % (n, n, have))
            return EXIT_USAGE
    import datetime
    store = dist.TCPStore("127.0.0.1", int(os.environ.get("MASTER_PORT", "0")), None, True,
                          datetime.timedelta(seconds=300), wait_for_workers=False)
    port = store.port
    child = shlex.split(os.environ["AMV_BENCH_CHILD"]) if os.environ.get("AMV_BENCH_CHILD") else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_USE_AGENT_STORE="True")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen(child + argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    # rank 0's stdout is the line (read by a thread, so that the other ranks are watched meanwhile); everything else the
    # children print is already on stderr
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    started = time.time()
    deadline = None
    codes = [None] * n

    def end(r, why, code):
        sys.stderr.write("bench.py: rank %d still running %s: killing pid %d\n" % (r, why, procs[r].pid))
        procs[r].kill()                                              # this exact child, nothing by pattern
        procs[r].wait()
        codes[r] = code

    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
        if any(c not in (None, 0) for c in codes) and deadline is None:
            deadline = time.time() + LAUNCH_GRACE_SECONDS      # a rank failed: the others get this long to end by themselves
        if deadline is not None and time.time() > deadline:
            for r in range(n):
                if codes[r] is None:
                    end(r, "%g s after another rank failed" % LAUNCH_GRACE_SECONDS, EXIT_HUNG)
        if time.time() - started > LAUNCH_LIMIT_SECONDS:
            for r in range(n):
                if codes[r] is None:
                    end(r, "after the run's limit of %g s (AMV_BENCH_LAUNCH_LIMIT)" % LAUNCH_LIMIT_SECONDS, EXIT_HUNG)
        time.sleep(0.05)
    reader.join(10)
    line = got[0] if got else b""
    sys.stdout.buffer.write(line)
    sys.stdout.flush()
    worst = 0
    for r, c in enumerate(codes):
        if c != 0:
            sys.stderr.write("bench.py: rank %d left with code %d\n" % (r, c))
            worst = max(worst, c if c > 0 else 128 - c)
    if not line.strip() and worst == 0:
        sys.stderr.write("bench.py: every rank left with code 0 but rank 0 wrote no line (or its pipe is still held open): "
                         "not a clean run\n")
        worst = EXIT_FAILED
    return worst


def run_workload(E, args):
    """everything between "the ranks, the device and the context are there" and "the line goes out" """
    if args.workload == "decode":
        result = run_decode(E, args)
        plain = not (args.frames or args.width or args.height or args.pipelined or args.stream != "synthetic")
        if plain and not args.no_secondary and not E.failed:
            run_secondary(E, args, result)
    elif args.workload == "encode":
        result = run_encode(E, args)
    elif args.workload == "coresident":
        result = run_adpcm(E, args, with_video=True)
    elif args.workload == "amvlib":
        result = run_amvlib(E, args)
    else:
        result = run_adpcm(E, args, with_video=False)
    return result


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("decode", "encode", "coresident", "adpcm", "amvlib"), default="decode")
    ap.add_argument("--frames", type=int, default=None, help="frames (audio chunks) per GPU per step; default per workload")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--psnr-floor", type=float, default=None,
                    help="encode: round-trip PSNR the batch must reach (dB); default = the floor frozen from the CPU "
                         "encoder in tests/test_oracle_pin.py::test_encode_round_trip_quality (26.6 at 320x240, 25.4 below: "
                         "amvlib's colour matrix is not the inverse of the encoder's, which bounds the figure)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelined", action="store_true",
                    help="decode workload: time the amvhip_decode_submit_dev / _collect_dev loop (batch k+1 submitted before "
                         "batch k is collected) instead of amvhip_decode_batch_dev, a batch at a time")
    ap.add_argument("--strong", action="store_true",
                    help="decode: also run configs[3] as stated (one 10 000-frame stream scattered from rank 0, decoded, gathered "
                         "back) and report it under config.config4_strong_10k; always on when WORLD_SIZE > 1")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="frames of the stream the CPU baseline decodes")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="bound on the all-core part of the CPU baseline")
    ap.add_argument("--stream", choices=("synthetic", "mixed", "amv1"), default="synthetic",
                    help="decode: the seeded synthetic source; the same with every 16th frame noise (chunk sizes spread over 20x); "
                         "the reference's clip AMV1.amv (128x96) repeated")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default line only: skip the other BASELINE configs (config.secondary)")
    return ap.parse_args(argv)


def main():
    args = parse_args()

    # --gpus is the number of ranks.  Under torch.distributed.run (or any launcher that sets WORLD_SIZE) it must agree with
    # the environment; a plain `python bench.py --gpus N` starts the N ranks itself -- BEFORE anything touches the GPU.
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: refusing to report one under the other's name"
                             % (args.gpus, os.environ["WORLD_SIZE"]))
    elif args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    elif args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")

    E = Env()
    E.world = int(os.environ.get("WORLD_SIZE", "1"))
    E.rank = int(os.environ.get("RANK", "0"))
    local = 0 if ONE_DEVICE else int(os.environ.get("LOCAL_RANK", "0"))
    # the one JSON line must be alone on stdout: RCCL prints a version banner there when a communicator comes up, so
    # descriptor 1 points at stderr for the length of the run and the line goes out through the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    E.json_fd = json_fd
    os.dup2(2, 1)
    E.dist = E.world > 1 or args.strong
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d of %d: %d devices needed, %d visible (the codec path has no CPU fallback)"
                         % (E.rank, E.world, local + 1, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the codec path has no CPU fallback")
    torch.cuda.set_device(local)
    if E.dist:          # --strong with one process: the same RCCL code path, world size 1 (a rehearsal on a one-GPU box)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if BACKEND == "nccl":
            dist.init_process_group("nccl", rank=E.rank, world_size=E.world,    # RCCL over xGMI
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(BACKEND, rank=E.rank, world_size=E.world)   # (a rehearsal: see BACKEND above)
        E.peers = Peers(E, dist.distributed_c10d._get_default_store())
    E.dev = torch.device("cuda", local)
    E.pkg = entry.load_package()
    E.sh = entry._load(entry.PKG_NAME + ".sharding", os.path.join(entry.PKG_DIR, "sharding.py"))
    E.ctx = E.pkg.Context(local)
    E.stream = torch.cuda.current_stream().cuda_stream

    try:
        result = run_workload(E, args)
        if E.dist and (BACKEND != "nccl" or ONE_DEVICE):
            result["config"]["rehearsal"] = "backend %s%s: the ranks, kernels, exchange and record of an N-rank run without RCCL / xGMI -- not a scaling figure" % (
                BACKEND, ", every rank on device 0" if ONE_DEVICE else "")
    except BaseException as e:          # a gate's SystemExit included: the peers must not wait for a rank that has left
        if E.peers and not isinstance(e, KeyboardInterrupt):
            E.peers.raise_flag("%s: %s" % (type(e).__name__, str(e)[:300]))
        raise
    write_line_and_leave(E, result)
    E.ctx.close()
    if E.dist:
        E.peers.stop.set()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
