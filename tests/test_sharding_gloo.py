"""The N>1 path on CPU: world_size 2 (and 3, uneven split) over gloo.  The frame-range scatter /
gather that bench.py and a multi-GPU host use is exercised end to end; the per-rank decode is the
oracle here because this container has no GPU (on the GPU box the same ranges go to the HIP path)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SEED

W, H, N = 160, 120, 21


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, src_share=None):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = entry.load_oracle()
        sh = entry._load("amv_codec_tools_amd.sharding", os.path.join(entry.PKG_DIR, "sharding.py"))
        if src_share is not None:
            sh.configure(src_share=src_share)       # rank 0, which holds the stream, keeps this fraction of the frames
        dev = torch.device("cpu")
        blob = offs = lens = None
        if rank == 0:
            blob, offs, lens = orc.synth_stream(SEED, 0, N, W, H)
        lo, hi = sh.frame_range(N, rank, world)

        calls = []

        def decode(my_blob, my_offs, my_lens, first, into=None):
            k = int(my_lens.numel())
            assert lo <= first and first + k <= hi
            calls.append((first, k, into is not None))
            b = my_blob.numpy()
            frames = np.zeros((k, H, orc.stride(W)), np.uint8)
            for i in range(k):
                o, ln = int(my_offs[i]), int(my_lens[i])
                out, st, _ = orc.decode_frame(b[o:o + ln].tobytes(), W, H)
                assert st == 0
                frames[i] = out
            if into is not None:                      # rank 0: the slice of the gathered buffer these frames belong in
                assert tuple(into.shape) == frames.shape
                into.copy_(torch.from_numpy(frames))
                return into
            return torch.from_numpy(frames)

        shape = (H, orc.stride(W))
        # the move bench.py times for BASELINE configs[3]: scatter-v -> per-rank decode -> gather; the stream is handed
        # over as numpy arrays the first time and as torch tensors (what bench.py holds) the second
        full, phases = sh.strong_step(blob, offs, lens, N, dev, decode, frame_shape=shape)
        assert set(phases) == {"scatter", "decode", "gather"}
        assert calls == [(lo, hi - lo, rank == 0)]    # one call for the whole range; in place on the gathering rank only
        if rank == 0:
            tb, to, tl = torch.from_numpy(blob), torch.from_numpy(offs.view(np.int64)), torch.from_numpy(lens.view(np.int32))
            full2, _ = sh.strong_step(tb, to, tl, N, dev, decode, frame_shape=shape, out=torch.empty_like(full))
            assert torch.equal(full, full2)
            # the frame shape learnt from the first decode, and the range sent in three sub-batches
            full3, _ = sh.strong_step(tb, to, tl, N, dev, decode)
            del calls[:]
            full4, _ = sh.strong_step(tb, to, tl, N, dev, decode, frame_shape=shape, k=3)
            assert torch.equal(full, full3) and torch.equal(full, full4)
        else:
            sh.strong_step(None, None, None, N, dev, decode, frame_shape=shape)
            sh.strong_step(None, None, None, N, dev, decode)
            del calls[:]
            sh.strong_step(None, None, None, N, dev, decode, frame_shape=shape, k=3)
        assert [c[:2] for c in calls] == [(a, b - a) for a, b in sh.sub_ranges(lo, hi, 3)]
        # the scattered slices are exactly the rank's bytes: nothing padded
        my_blob, my_offs, my_lens, first = sh.scatter_stream(blob, offs, lens, dev)
        assert first == lo and my_lens.numel() == hi - lo
        if hi > lo:
            assert int(my_offs[0]) == 0 and my_blob.numel() == int(my_offs[-1]) + int(my_lens[-1])
        # the stream held by rank 1, its chunks packed back to back (chunk starts at any byte): the source's own range is a
        # view of its blob, and a view the decode ABI accepts -- 4-byte aligned (amvhip_decode_batch_dev rejects others)
        pb = po = pl = None
        if rank == 1:
            b0_, o0_, l0_ = orc.synth_stream(SEED, 0, N, W, H)
            po = np.concatenate([[1], 1 + np.cumsum(l0_.astype(np.int64))[:-1]]).astype(np.uint64)    # odd starts
            pb = np.zeros(int(po[-1]) + int(l0_[-1]) + 8, np.uint8)
            for i in range(N):
                pb[int(po[i]):int(po[i]) + int(l0_[i])] = b0_[int(o0_[i]):int(o0_[i]) + int(l0_[i])]
            pl = l0_
        if src_share is not None:
            sh.configure(src_share=src_share, src=1)  # (the larger range moves to the rank that holds this stream)
            lo, hi = sh.frame_range(N, rank, world)
            my_blob, my_offs, my_lens, first = sh.scatter_stream(blob, offs, lens, dev)   # rank 0's stream under the same cut
            assert first == lo and my_lens.numel() == hi - lo
        s_blob, s_offs, s_lens, s_first = sh.scatter_stream(pb, po, pl, dev, src=1)
        assert s_first == lo and s_lens.numel() == hi - lo
        if hi > lo:
            assert s_blob.data_ptr() % 4 == 0 and 0 <= int(s_offs[0]) < 4
            assert s_blob.numel() == int(s_offs[-1]) + int(s_lens[-1])
        assert torch.equal(decode(s_blob, s_offs, s_lens, s_first), decode(my_blob, my_offs, my_lens, first))
        # gather_frames: the plain form (a decoder that knows nothing of `into`)
        mine = decode(my_blob, my_offs, my_lens, first)
        got = sh.gather_frames(mine, N)
        assert (got is None) == (rank != 0)
        if rank == 0:
            assert torch.equal(got, full)
        slow = sh.max_over_ranks(float(rank + 1), dev)
        total = sh.sum_over_ranks(float(hi - lo), dev)
        assert slow == float(world) and total == float(N)
        if rank == 0:
            want, _ = orc.decode_batch(blob, offs, lens, W, H)
            q.put(bool((full.numpy() == want).all()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,src_share", [(2, None), (3, None), (3, 0.5), (2, 0.9), (3, 0.0)])
def test_scatter_decode_gather(world, src_share):
    """src_share: the rank that holds the stream keeps that fraction of the frames (sharding.configure) -- half of them among
    three ranks, nine tenths among two, none at all (a source that only distributes): the same exchange, the same frames"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, src_share)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_frame_ranges_cover_everything():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    sh = entry._load("amv_codec_tools_amd.sharding", os.path.join(entry.PKG_DIR, "sharding.py"))
    for n in (0, 1, 7, 8, 10000, 10001):
        for g in (1, 2, 3, 4, 8):
            rs = [sh.frame_range(n, r, g) for r in range(g)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(g - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
    # a larger range for the rank that holds the stream: still contiguous, in rank order, covering everything; the others equal
    try:
        for n in (0, 1, 7, 10000, 10001):
            for g in (2, 3, 8):
                for src in (0, 1, g - 1):
                    for share in (0.0, 0.125, 0.25, 0.5, 1.0):
                        sh.configure(src_share=share, src=src)
                        rs = [sh.frame_range(n, r, g) for r in range(g)]
                        assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(g - 1))
                        assert rs[src][1] - rs[src][0] == min(n, int(round(share * n)))
                        rest = [b - a for r, (a, b) in enumerate(rs) if r != src]
                        assert max(rest) - min(rest) <= 1
                        assert sh.range_bounds(n, g) == [rs[0][0]] + [b for _, b in rs]
    finally:
        sh.configure()
    assert sh.frame_range(10, 1, 2) == (5, 10)
