"""Frame sharding of an AMV stream over the GPUs of one node (one process per GPU).

Every video chunk and every audio chunk decodes on its own (intra-only video, each audio chunk
carries predictor + step index), so a stream shards by contiguous frame range with no exchange
inside the codec path.  The only collectives are the ones either side of it: rank 0 hands each rank
its slice of the compressed stream (scatter-v), and fixed-size decoded frames come back (gather).
torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the
CPU tests.  At ~0.6 GB/s of payload per GPU these moves are three orders of magnitude below one
xGMI link, so a plain scatter/gather is the right shape (no ring tuning, no overlap machinery).
"""
import numpy as np
import torch
import torch.distributed as dist


def frame_range(n_total, rank, world):
    """contiguous slice [lo, hi) of rank `rank`: r*n/G .. (r+1)*n/G"""
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


def shard_tables(offs, lens, rank, world):
    """(lo, hi, byte_lo, byte_hi) of this rank's chunks inside the stream's blob"""
    lo, hi = frame_range(len(lens), rank, world)
    if hi == lo:
        return lo, hi, 0, 0
    return lo, hi, int(offs[lo]), int(offs[hi - 1]) + int(lens[hi - 1])


def _as_tensor(x, dtype):
    """numpy array or torch tensor (host or device) -> torch tensor of `dtype`, no copy when possible"""
    if isinstance(x, np.ndarray):
        if x.dtype == np.uint64:
            x = x.view(np.int64)
        elif x.dtype == np.uint32:
            x = x.view(np.int32)
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x if x.dtype == dtype else x.to(dtype)


def scatter_stream(blob, offs, lens, device, src=0):
    """rank `src` holds the stream (blob uint8, offs, lens) as numpy arrays or as torch tensors (host or
    already on `device`: then every move is device to device); every rank returns its own
    (blob tensor on `device`, offs int64 rebased to 0, lens int32, first_frame).  Other ranks pass None.
    One broadcast of the split table (4 numbers per rank), then one scatter each for bytes, offsets, lengths."""
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = torch.zeros(2 + 4 * world, dtype=torch.int64)
    if rank == src:
        blob, offs, lens = _as_tensor(blob, torch.uint8), _as_tensor(offs, torch.int64), _as_tensor(lens, torch.int32)
        n = int(lens.numel())
        o_h, l_h = offs.cpu().numpy(), lens.cpu().numpy()          # the split table is made on the host
        rows = [shard_tables(o_h, l_h, r, world) for r in range(world)]
        meta[0], meta[1] = n, max(1, max(b1 - b0 for _, _, b0, b1 in rows))
        meta[2:] = torch.tensor(rows, dtype=torch.int64).flatten()
    meta = meta.to(device)
    dist.broadcast(meta, src)
    m = meta.cpu().tolist()
    n, maxb = m[0], m[1]
    lo, hi, b0, b1 = m[2 + 4 * rank: 6 + 4 * rank]
    maxf = max(1, max(m[3 + 4 * r] - m[2 + 4 * r] for r in range(world)))
    my_blob = torch.zeros(maxb, dtype=torch.uint8, device=device)
    my_offs = torch.zeros(maxf, dtype=torch.int64, device=device)
    my_lens = torch.zeros(maxf, dtype=torch.int32, device=device)
    if rank == src:
        blob, offs, lens = blob.to(device), offs.to(device), lens.to(device)
        bl, ol, ll = [], [], []
        for r in range(world):
            rlo, rhi, rb0, rb1 = m[2 + 4 * r: 6 + 4 * r]
            b = torch.zeros(maxb, dtype=torch.uint8, device=device)
            b[: rb1 - rb0] = blob[rb0:rb1]
            o = torch.zeros(maxf, dtype=torch.int64, device=device)
            o[: rhi - rlo] = offs[rlo:rhi] - rb0
            ln = torch.zeros(maxf, dtype=torch.int32, device=device)
            ln[: rhi - rlo] = lens[rlo:rhi]
            bl.append(b); ol.append(o); ll.append(ln)
        dist.scatter(my_blob, bl, src)
        dist.scatter(my_offs, ol, src)
        dist.scatter(my_lens, ll, src)
    else:
        dist.scatter(my_blob, None, src)
        dist.scatter(my_offs, None, src)
        dist.scatter(my_lens, None, src)
    return my_blob[: max(b1 - b0, 0)], my_offs[: hi - lo], my_lens[: hi - lo], lo


def gather_frames(local, n_total, dst=0):
    """local: [n_local, ...] uint8 frames of this rank's range; rank `dst` gets [n_total, ...]"""
    rank, world = dist.get_rank(), dist.get_world_size()
    maxf = max(frame_range(n_total, r, world)[1] - frame_range(n_total, r, world)[0] for r in range(world))
    pad = torch.zeros((maxf,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if rank == dst:
        parts = [torch.zeros_like(pad) for _ in range(world)]
        dist.gather(pad, parts, dst)
        out = []
        for r in range(world):
            lo, hi = frame_range(n_total, r, world)
            out.append(parts[r][: hi - lo])
        return torch.cat(out)
    dist.gather(pad, None, dst)
    return None


def max_over_ranks(seconds, device):
    """the slowest rank's time (what the bench contract divides by)"""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def strong_step(blob, offs, lens, n_total, device, decode, src=0, clock=None):
    """BASELINE.json configs[3] as stated: ONE stream of n_total frames held by rank `src` is frame-sharded over the
    ranks -- scatter-v of the compressed chunks, decode(my_blob, my_offs, my_lens, first_frame) -> [n_local, ...]
    uint8 frames on `device`, gather of the frames back to `src`.  Returns (frames on src | None, seconds per phase)
    with clock() read after each phase (pass a function that synchronises the device first when timing a GPU)."""
    clock = clock or (lambda: 0.0)
    t0 = clock()
    my_blob, my_offs, my_lens, first = scatter_stream(blob, offs, lens, device, src)
    t1 = clock()
    local = decode(my_blob, my_offs, my_lens, first)
    t2 = clock()
    full = gather_frames(local, n_total, src)
    t3 = clock()
    return full, {"scatter": t1 - t0, "decode": t2 - t1, "gather": t3 - t2}
