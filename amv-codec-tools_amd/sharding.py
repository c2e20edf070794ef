"""Frame sharding of an AMV stream over the GPUs of one node (one process per GPU).

Every video chunk and every audio chunk decodes on its own (intra-only video, each audio chunk
carries predictor + step index), so a stream shards by contiguous frame range with no exchange
inside the codec path.  The only moves are the ones either side of it: rank 0 hands each rank its
slice of the compressed stream (scatter-v), and fixed-size decoded frames come back (gather).
torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the
CPU tests.

Shape of the exchange (round 4; the round-3 form built zero-padded copies on both sides -- four passes
over the frames, more time than the decode at world size one):

* nothing is padded and nothing is copied that does not cross a link: the source SENDS SLICES of its own
  blob / offset / length tensors (point to point, one grouped call: `batch_isend_irecv`) and keeps its own
  range as views; the destination RECEIVES STRAIGHT INTO slices of one [n_total, ...] buffer, and its own
  rank's decoder writes into its slice of that buffer in place (`into=`);
* the split table (4 numbers per rank) is made on the device that holds the stream -- two gathers over
  offs / lens -- and broadcast; every rank reads it back once (the receive sizes are data);
* a rank's range can go out in k sub-batches, the send of sub-batch j posted (on RCCL's own stream) while
  sub-batch j + 1 decodes.  xGMI is point to point -- seven links of ~153 GB/s per GPU, one per peer --
  so the gather into rank 0 runs on seven links at once and is bound by each sender's ONE link:
  BASELINE configs[3] at 8 ranks moves 7 x 72 MB = 504 MB into rank 0, 72 MB / 153 GB/s = 0.47 ms per
  link in parallel, against ~0.31 ms for the decode of a rank's 1 250 frames (a launch one wave deep);
  DESIGN.md section 10 has the arithmetic.  No 1 -> 8 run has been measured;
* the destination posts its receives BEFORE it decodes its own range (`FrameGather.expect`): RCCL's stream then waits for
  nothing of the decode, and a destination that keeps a larger range (`configure`) decodes it beside the transfers.
"""
import numpy as np
import torch
import torch.distributed as dist


# How a stream is cut.  Equal contiguous ranges unless `configure` gives the rank that holds the stream a
# larger one: that rank pays no link for its own frames -- its range is decoded in place in the gathered buffer -- while
# every other rank decodes its frames and THEN sends them over its ONE xGMI link to the source (~153 GB/s), and a rank's
# decode of configs[3]'s share grows slowly with its frames (a launch one generation deep: 0.31 ms for 1 250 frames, ~0.37 for
# 2 500, ~0.49 for 5 000).  With 8 ranks and the source keeping a share s of a 10 000-frame 160x120 stream a sender needs
# decode + (1 - s) / 7 x 576 MB / 153 GB/s: s = 1/8 -> 0.31 + 0.47 ms, s = 1/2 -> 0.29 + 0.27 ms beside the source's own
# 0.49 ms for 5 000 frames -- ~0.61 ms end to end against 0.83 with equal ranges and 0.73 on one GPU.  DESIGN.md section 10
# has the table; NOTHING of it has been measured on more than one GPU.  A knob, not a claim.
_SPLIT = {"src": 0, "src_share": None}


def configure(src_share=None, src=0):
    """src_share: the fraction of a stream's frames rank `src` keeps (None: equal ranges); the others share the rest equally"""
    if src_share is not None and not 0.0 <= float(src_share) <= 1.0:
        raise ValueError("src_share must be inside [0, 1]")
    _SPLIT["src"], _SPLIT["src_share"] = int(src), None if src_share is None else float(src_share)
    _INDEX_CACHE.clear()


def range_bounds(n_total, world):
    """b[0 .. world]: rank r owns frames [b[r], b[r + 1]) -- contiguous, in rank order, covering [0, n_total)"""
    share, src = _SPLIT["src_share"], _SPLIT["src"]
    if share is None or world == 1 or not 0 <= src < world:
        return [(r * n_total) // world for r in range(world + 1)]
    mine = min(n_total, int(round(share * n_total)))
    rest, others = n_total - mine, world - 1
    sizes = [((k + 1) * rest) // others - (k * rest) // others for k in range(others)]
    sizes.insert(src, mine)
    b = [0]
    for sz in sizes:
        b.append(b[-1] + sz)
    return b


def frame_range(n_total, rank, world):
    """contiguous slice [lo, hi) of rank `rank`: r*n/G .. (r+1)*n/G, or what `configure` made of it"""
    if _SPLIT["src_share"] is None:
        return (rank * n_total) // world, ((rank + 1) * n_total) // world
    b = range_bounds(n_total, world)
    return b[rank], b[rank + 1]


def shard_tables(offs, lens, rank, world):
    """(lo, hi, byte_lo, byte_hi) of this rank's chunks inside the stream's blob (host arrays)"""
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


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _host_transport(tensors):
    """gloo given device tensors (bench.py's one-GPU rehearsal of the N-rank flow): it reads and writes them from the host,
    knowing nothing of streams -- so the device must have finished what produces a tensor before it is posted, and the
    host must have finished a receive before a kernel reads it.  RCCL orders its transfers on streams: nothing to do."""
    return (dist.is_initialized() and dist.get_backend() == "gloo" and any(t is not None and t.is_cuda for t in tensors))


def _exchange(ops):
    """one grouped point-to-point call (RCCL: one fused launch); returns when the transfers are done or, on a device,
    ordered on the current stream"""
    if not ops:
        return
    host = _host_transport([op.tensor for op in ops])
    if host:
        torch.cuda.synchronize()
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if host:
        torch.cuda.synchronize()


_INDEX_CACHE = {}


def _range_ends(n, world, device):
    """device indices [lo_0 .. lo_{w-1}, last_0 .. last_{w-1}] of the ranks' first and last chunks (made once per stream
    length: building them is a host-to-device copy)"""
    key = (n, world, str(device))
    if key not in _INDEX_CACHE:
        los = [min(frame_range(n, r, world)[0], max(n - 1, 0)) for r in range(world)]
        lasts = [max(frame_range(n, r, world)[1] - 1, 0) for r in range(world)]
        if len(_INDEX_CACHE) > 64:
            _INDEX_CACHE.clear()
        _INDEX_CACHE[key] = torch.tensor(los + lasts, dtype=torch.int64, device=device)
    return _INDEX_CACHE[key]


def split_probe(offs, lens, world):
    """[3 * world + 1] int64 on the tensors' device: offs at every rank's first chunk, offs and lens at its last one, and
    the chunk count -- all the split table needs from the stream, picked by two gathers, no host round trip"""
    n = int(lens.numel())
    if n == 0:
        return torch.zeros(3 * world + 1, dtype=torch.int64, device=offs.device)
    idx = _range_ends(n, world, offs.device)
    return torch.cat([offs.index_select(0, idx), lens.index_select(0, idx[world:]).to(torch.int64),
                      torch.full((1,), n, dtype=torch.int64, device=offs.device)])


def split_table(probe, world):
    """host side of it: rows (lo, hi, byte_lo, byte_hi) per rank from split_probe's numbers"""
    p = probe.tolist() if hasattr(probe, "tolist") else list(probe)
    n = p[3 * world]
    rows = []
    for r in range(world):
        lo, hi = frame_range(n, r, world)
        rows.append((lo, hi, 0, 0) if hi == lo else (lo, hi, p[r], p[world + r] + p[2 * world + r]))
    return rows


def scatter_stream(blob, offs, lens, device, src=0):
    """rank `src` holds the stream (blob uint8, offs, lens) as numpy arrays or as torch tensors (host or already on
    `device`: then every move is device to device); every rank returns its own
    (blob tensor on `device`, offs int64 rebased to 0, lens int32, first_frame).  Other ranks pass None.
    One broadcast of the numbers the split table is made of, then one grouped point-to-point call: the source sends
    slices of its own tensors (its own range stays a view), the others receive exactly their bytes."""
    rank, world = _world()
    probe = torch.zeros(3 * world + 1, dtype=torch.int64, device=device)
    if rank == src:
        blob = _as_tensor(blob, torch.uint8).to(device)
        offs, lens = _as_tensor(offs, torch.int64).to(device), _as_tensor(lens, torch.int32).to(device)
        probe = split_probe(offs, lens, world)
    if world > 1:
        dist.broadcast(probe, src)
    m = split_table(probe.cpu(), world)             # the one read-back: receive sizes are data
    lo, hi, b0, b1 = m[rank]
    if rank == src:
        ops = []
        for r in range(world):
            rlo, rhi, rb0, rb1 = m[r]
            if r == src or rhi == rlo:
                continue
            ops += [dist.P2POp(dist.isend, blob[rb0:rb1], r), dist.P2POp(dist.isend, offs[rlo:rhi], r),
                    dist.P2POp(dist.isend, lens[rlo:rhi], r)]
        _exchange(ops)
        # the source's own range stays a view of its blob -- from the 4-byte boundary at or below its first chunk, because
        # amvhip_decode_batch_dev wants a 4-byte aligned blob (include/amvhip.h) and a chunk may start anywhere
        base = b0 & ~3 if blob.data_ptr() % 4 == 0 else b0
        mine = blob[base:b1]
        if mine.numel() and mine.data_ptr() % 4:
            mine = mine.clone()                     # a blob that was itself handed over at an odd address: one copy
        return mine, (offs[lo:hi] - base if base else offs[lo:hi]), lens[lo:hi], lo
    my_blob = torch.empty(max(b1 - b0, 0), dtype=torch.uint8, device=device)
    my_offs = torch.empty(hi - lo, dtype=torch.int64, device=device)
    my_lens = torch.empty(hi - lo, dtype=torch.int32, device=device)
    if hi > lo:
        _exchange([dist.P2POp(dist.irecv, my_blob, src), dist.P2POp(dist.irecv, my_offs, src), dist.P2POp(dist.irecv, my_lens, src)])
        my_offs -= b0
    return my_blob, my_offs, my_lens, lo


def sub_ranges(lo, hi, k):
    """[lo, hi) cut into k contiguous pieces (the last ones may be empty)"""
    n = hi - lo
    return [(lo + (j * n) // k, lo + ((j + 1) * n) // k) for j in range(k)]


class FrameGather:
    """Fixed-size decoded frames back to rank `dst`, straight into slices of ONE [n_total, ...] buffer there.

    g = FrameGather(n_total, frame_shape, dtype, device, dst, k)
    for j, (a, b) in enumerate(g.pieces):          # this rank's sub-batches, global frame numbers
        g.expect(j)                                # on dst: the receives of sub-batch j, posted before its own decode
        frames = decode(..., into=g.slot(j))       # on dst: a view of the buffer, decode writes in place; elsewhere None
        g.post(j, frames)                          # elsewhere: send sub-batch j; returns at once on a device
    full = g.finish()                              # on dst: the buffer, every range in place; elsewhere None
    """

    def __init__(self, n_total, frame_shape, dtype, device, dst=0, k=1, out=None):
        self.rank, self.world = _world()
        self.n_total, self.dst, self.k = n_total, dst, max(1, int(k))
        self.pieces = sub_ranges(*frame_range(n_total, self.rank, self.world), self.k)
        self.works = []
        self.expected = set()
        self.out = None
        if self.rank == dst:
            shape = (n_total,) + tuple(frame_shape)
            self.out = out if out is not None and tuple(out.shape) == shape else torch.empty(shape, dtype=dtype, device=device)

    def slot(self, j):
        if self.out is None:
            return None
        a, b = self.pieces[j]
        return self.out[a:b]

    def expect(self, j):
        """on dst: post the receives of every other rank's sub-batch j (once; they land in slices no decode of this rank
        writes).  Posted before this rank's own decode, they wait for nothing of it."""
        if self.rank != self.dst or j in self.expected:
            return
        self.expected.add(j)
        ops = []
        for r in range(self.world):
            if r == self.dst:
                continue
            ra, rb = sub_ranges(*frame_range(self.n_total, r, self.world), self.k)[j]
            if rb > ra:
                ops.append(dist.P2POp(dist.irecv, self.out[ra:rb], r))
        if ops:
            self.works += dist.batch_isend_irecv(ops)

    def post(self, j, frames):
        a, b = self.pieces[j]
        if self.rank == self.dst:
            if b > a and (frames.data_ptr() != self.out[a:b].data_ptr()):
                self.out[a:b].copy_(frames)         # a decoder that did not take `into`
            self.expect(j)
        elif b > a:
            self.keep = getattr(self, "keep", []) + [frames]        # alive until the send has gone
            if _host_transport([frames]):
                torch.cuda.synchronize()                            # (gloo reads the frames from the host: the decode must be done)
            self.works += dist.batch_isend_irecv([dist.P2POp(dist.isend, frames, self.dst)])

    def finish(self):
        for w in self.works:
            w.wait()
        if self.works and _host_transport([self.out] + getattr(self, "keep", [])):
            torch.cuda.synchronize()
        self.works, self.keep, self.expected = [], [], set()
        return self.out


def gather_frames(local, n_total, dst=0, out=None):
    """local: [n_local, ...] uint8 frames of this rank's range; rank `dst` gets [n_total, ...] (one buffer, every range
    received in place; `out` reuses a buffer of that shape)"""
    g = FrameGather(n_total, tuple(local.shape[1:]), local.dtype, local.device, dst, 1, out)
    g.post(0, local)
    return g.finish()


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


def strong_step(blob, offs, lens, n_total, device, decode, src=0, clock=None, frame_shape=None, dtype=torch.uint8, k=1, out=None):
    """BASELINE.json configs[3] as stated: ONE stream of n_total frames held by rank `src` is frame-sharded over the
    ranks -- scatter-v of the compressed chunks, decode(my_blob, my_offs, my_lens, first_frame, into) -> [n_local, ...]
    frames on `device` (`into`: on `src` the slice of the result the frames belong in -- write there and return it --
    elsewhere None), the frames back to `src` in k sub-batches, each sent while the next one decodes.
    Returns (frames on src | None, seconds per phase) with clock() read after each phase (pass a function that
    synchronises the device first when timing a GPU; with k > 1 `decode` includes the sends it overlaps)."""
    clock = clock or (lambda: 0.0)
    t0 = clock()
    my_blob, my_offs, my_lens, first = scatter_stream(blob, offs, lens, device, src)
    t1 = clock()
    g = None
    for j in range(max(1, int(k))):
        if g is None:
            lo, hi = sub_ranges(first, first + int(my_lens.numel()), max(1, int(k)))[j]
        else:
            lo, hi = g.pieces[j]
        a, b = lo - first, hi - first
        if g is None and frame_shape is not None:
            g = FrameGather(n_total, frame_shape, dtype, device, src, k, out)
        into = g.slot(j) if g is not None else None
        if g is not None:
            g.expect(j)
        frames = decode(my_blob, my_offs[a:b], my_lens[a:b], lo, into)
        if g is None:                               # frame shape learnt from the first decode
            g = FrameGather(n_total, tuple(frames.shape[1:]), frames.dtype, device, src, k, out)
        if j == max(1, int(k)) - 1:
            t2 = clock()
        g.post(j, frames)
    full = g.finish()
    t3 = clock()
    return full, {"scatter": t1 - t0, "decode": t2 - t1, "gather": t3 - t2}
