"""Parity of the HIP path against the CPU oracle, through the C ABI, on a real MI355X.
Bit-exact everywhere (integer / byte work).  One process, one context."""
import ctypes

import numpy as np
import pytest

from conftest import SEED
from test_oracle_pin import AMVLIB_HASH, AMVLIB_HASH_FIXED_ZZ

pytestmark = pytest.mark.gpu


def _t(arr, dev="cuda:0"):
    import torch
    a = np.ascontiguousarray(arr)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    elif a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).to(dev)


def _blob_of(chunks, pad_front=0):
    """pack chunks back to back (optionally starting at an odd offset to exercise unaligned reads)"""
    offs, lens, parts, pos = [], [], [b"\xaa" * pad_front], pad_front
    for c in chunks:
        offs.append(pos)
        lens.append(len(c))
        parts.append(bytes(c))
        pos += len(c)
    blob = np.frombuffer(b"".join(parts) + b"\0" * 8, np.uint8).copy()
    return blob, np.array(offs, np.uint64), np.array(lens, np.uint32), pos


def _gpu_decode(ctx, chunks, w, h, flags=0, pad_front=0):
    import torch
    blob, offs, lens, nbytes = _blob_of(chunks, pad_front)
    n = len(chunks)
    d_out = torch.full((n, h, ctx.stride(w)), 0x5A, dtype=torch.uint8, device="cuda:0")
    d_st = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    ctx.decode_batch_dev(_t(blob), nbytes, _t(offs), _t(lens), n, w, h, flags, d_out, d_st,
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), d_st.cpu().numpy()


def _oracle_decode(orc, chunks, w, h, flags=0):
    outs, sts = [], []
    for c in chunks:
        o, st, _ = orc.decode_frame(c, w, h, flags)
        outs.append(o)
        sts.append(st)
    return np.stack(outs), np.array(sts, np.int32)


def _synth_chunks(orc, n, w, h, first=0, qbias=0):
    return [orc.encode_frame(orc.synth_frame(SEED, first + t, w, h), w, h, qbias=qbias) for t in range(n)]


# ---------------------------------------------------------------------------------- decode

@pytest.mark.parametrize("flags,want", [(0, AMVLIB_HASH), (1, AMVLIB_HASH_FIXED_ZZ)])
def test_decode_amv1_matches_amvlib_hash(ctx, orc, amv1, flags, want):
    """the reference's own clip: all 252 frames, hashed the way the survey hashed amvlib's output"""
    out, st = _gpu_decode(ctx, amv1["video"], 128, 96, flags)
    assert (st == 0).all()
    h = orc.SURVEY_FNV_SEED
    for f in out:
        h = orc.fnv1a64(h, f)
    assert h == want


@pytest.mark.parametrize("w,h,n,pad", [(160, 120, 70, 0), (160, 120, 5, 3), (320, 240, 9, 1), (176, 144, 3, 2),
                                       (16, 16, 2, 0), (130, 98, 3, 0), (336, 32, 2, 0), (8, 8, 1, 0)])
def test_decode_matches_oracle(ctx, orc, w, h, n, pad):
    """ragged batch sizes (tail lanes), sizes with partial MCUs / padded rows, unaligned chunk starts"""
    we, he = w + (w & 1), h + (h & 1)
    chunks = []
    for t in range(n):
        src = orc.synth_frame(SEED, t, we, he)
        chunks.append(orc.encode_frame(src, we, he))
    if (we, he) != (w, h):   # odd sizes: decode the even-size stream at the odd size the header would claim
        pytest.skip("encoder needs even sizes")
    for flags in (0, 1):
        got, st = _gpu_decode(ctx, chunks, w, h, flags, pad)
        want, wst = _oracle_decode(orc, chunks, w, h, flags)
        assert (st == wst).all() and (wst == 0).all()
        assert (got == want).all()


def test_decode_odd_width_row_padding(ctx, orc):
    """width 130: rows are padded to 392 bytes and the pad bytes stay zero (AmvJpeg.c:1524, AMVDec.c:283)"""
    chunks = _synth_chunks(orc, 3, 144, 96)      # 9 MCUs wide; claim 130 px: the last MCU is cut at the right edge
    got, st = _gpu_decode(ctx, chunks, 130, 90)
    want, wst = _oracle_decode(orc, chunks, 130, 90)
    assert got.shape[2] == 392 and (st == wst).all() and (got == want).all()


def test_decode_empty_batch(ctx):
    assert ctx.decode_batch_dev(None, 0, None, None, 0, 160, 120, 0, None, None) == 0


def test_decode_error_frames_match_oracle(ctx, orc):
    """corrupt, truncated and garbage chunks: status bits and every output byte equal the oracle's
    (MCUs before the error are stored, the rest stays zero)"""
    w, h = 160, 120
    rng = np.random.default_rng(5)
    good = _synth_chunks(orc, 6, w, h)
    chunks = list(good)
    chunks.append(good[0][: len(good[0]) // 3])                         # truncated
    chunks.append(good[1][:2])                                          # header only
    chunks.append(b"")                                                  # empty
    b = bytearray(good[2]); b[400:440] = b"\xff\x00" * 20; chunks.append(bytes(b))   # run of ones: FORMAT
    for k in range(12):                                                 # random corruption
        b = bytearray(good[k % 6])
        for _ in range(1 + k):
            b[int(rng.integers(2, len(b) - 2))] = int(rng.integers(0, 256))
        chunks.append(bytes(b))
    for k in range(6):                                                  # pure noise
        chunks.append(b"\xff\xd8" + rng.integers(0, 256, 3000).astype(np.uint8).tobytes())
    chunks.append(b"\xff\xd8" + b"\xff" * 500)                          # all FF
    got, st = _gpu_decode(ctx, chunks, w, h, 0, pad_front=1)
    want, wst = _oracle_decode(orc, chunks, w, h)
    assert (st == wst).all(), (st, wst)
    assert (got == want).all()
    assert (wst != 0).sum() >= 8 and (wst & orc.ST_FORMAT).any() and (wst & orc.ST_TRUNCATED).any()


def test_decode_both_entropy_kernels(ctx, pkg, orc, amv1):
    """the one-lane-per-frame kernel (SERIAL) and the wave-per-frame kernel (AUTO) give the same
    bytes and statuses; chunks too large for the LDS window and long FF runs take the hand-back path"""
    w, h = 160, 120
    rng = np.random.default_rng(17)
    chunks = _synth_chunks(orc, 9, w, h, first=900)
    chunks.append(b"\xff\xd8" + rng.integers(0, 255, 9000).astype(np.uint8).tobytes())      # > LDS window
    chunks.append(b"\xff\xd8" + bytes(rng.integers(0, 256, 20000).astype(np.uint8)))       # > LDS window, with FFs
    chunks.append(chunks[0][:1000] + b"\xff" * 9 + chunks[0][1000:])                         # long FF run
    chunks.append(chunks[1][:-2])                                                           # EOI missing
    chunks.append(chunks[2] + b"\x00" * 5000)                                               # trailing junk
    big = orc.encode_frame(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), w, h)          # noise: ~20 kB chunk
    assert len(big) > 9000
    chunks.append(big)
    want, wst = _oracle_decode(orc, chunks, w, h)
    try:
        for mode in (pkg.ENTROPY_SERIAL, pkg.ENTROPY_AUTO):
            ctx.set_entropy_mode(mode)
            got, st = _gpu_decode(ctx, chunks, w, h, 0, pad_front=3)
            assert (st == wst).all(), (mode, st, wst)
            assert (got == want).all(), mode
            out, st = _gpu_decode(ctx, amv1["video"][:70], 128, 96)
            hsh = orc.FNV_BASIS
            ref = orc.FNV_BASIS
            for k in range(70):
                hsh = orc.fnv1a64(hsh, out[k])
                ref = orc.fnv1a64(ref, orc.decode_frame(amv1["video"][k], 128, 96)[0])
            assert hsh == ref and (st == 0).all()
    finally:
        ctx.set_entropy_mode(pkg.ENTROPY_AUTO)


def test_huffman_stage_matches_oracle(ctx, orc):
    import torch
    w, h, n = 160, 120, 67
    chunks = _synth_chunks(orc, n, w, h, first=40)
    blob, offs, lens, nbytes = _blob_of(chunks, 2)
    nblk = orc.nmcu(w, h) * 6
    d_coef = torch.full((n, nblk, 64), 77, dtype=torch.int16, device="cuda:0")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda:0")
    d_ok = torch.empty(n, dtype=torch.int32, device="cuda:0")
    ctx.huffman_decode_dev(_t(blob), nbytes, _t(offs), _t(lens), n, w, h, d_coef, d_st, d_ok)
    torch.cuda.synchronize()
    coef = d_coef.cpu().numpy()
    assert (d_st.cpu().numpy() == 0).all() and (d_ok.cpu().numpy() == 80).all()
    for i, c in enumerate(chunks):
        want = orc.decode_frame(c, w, h, 0, want_coef=True)[3]
        assert (coef[i] == want).all(), i


def test_reconstruct_stage_matches_oracle(ctx, orc):
    """dequant + IDCT + colour from arbitrary coefficients, including values no encoder produces"""
    import torch
    w, h, n = 160, 120, 4
    nm = orc.nmcu(w, h)
    rng = np.random.default_rng(9)
    coef = np.zeros((n, nm * 6, 64), np.int16)
    coef[0] = rng.integers(-40, 41, coef[0].shape)
    coef[1, :, 0] = rng.integers(-1024, 1024, nm * 6)                     # DC only
    coef[2] = rng.integers(-1023, 1024, coef[2].shape) * (rng.random(coef[2].shape) < 0.1)
    coef[3] = rng.integers(-32768, 32768, coef[3].shape)                  # saturating: iclp / wrap behaviour
    ok = np.array([nm, nm, nm, nm - 7], np.uint32)
    d_out = torch.zeros((n, h, ctx.stride(w)), dtype=torch.uint8, device="cuda:0")
    for flags in (0, 1):
        ctx.reconstruct_dev(_t(coef), _t(ok), n, w, h, flags, d_out)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        px = np.zeros(64, np.int32)
        for f in range(n):
            want = np.zeros((h, orc.stride(w)), np.uint8)
            for m in range(int(ok[f])):
                blocks = []
                for k in range(6):
                    orc.lib().amvo_dequant_idct_block(np.ascontiguousarray(coef[f, m * 6 + k]).ctypes.data, min(k - 3, 2) if k >= 4 else 0, flags, px.ctypes.data)
                    blocks.append(px.copy())
                my, mx = divmod(m, 10)
                for i in range(16):
                    row = my * 16 + i
                    if row >= h:
                        break
                    for j in range(16):
                        y = blocks[(i >> 3) * 2 + (j >> 3)][(i & 7) * 8 + (j & 7)]
                        u = blocks[4][(i >> 1) * 8 + (j >> 1)]
                        v = blocks[5][(i >> 1) * 8 + (j >> 1)]
                        bgr = np.zeros(3, np.uint8)
                        orc.lib().amvo_yuv_to_bgr(int(y), int(u), int(v), bgr.ctypes.data)
                        want[h - 1 - row, (mx * 16 + j) * 3:(mx * 16 + j) * 3 + 3] = bgr
            assert (got[f] == want).all(), (flags, f)


def test_decode_host_buffers_and_aliases(ctx, pkg, orc):
    w, h = 160, 120
    chunks = _synth_chunks(orc, 3, w, h, first=300)
    blob, offs, lens, nbytes = _blob_of(chunks)
    out = np.zeros((3, h, ctx.stride(w)), np.uint8)
    st = np.full(3, -1, np.int32)
    ctx.decode_batch(blob, nbytes, offs, lens, 3, w, h, 0, out, st)
    want, _ = _oracle_decode(orc, chunks, w, h)
    assert (st == 0).all() and (out == want).all()
    one = np.zeros((h, w * 3), np.uint8)
    assert pkg.load_library().decode_amv_frame(chunks[1], len(chunks[1]), w, h, one.ctypes.data) == 0
    assert (one == want[1]).all()
    assert pkg.load_library().decode_amv_frame(chunks[1][:200], 200, w, h, one.ctypes.data) == -1


def test_async_decode_calls_in_flight(ctx, pkg, orc):
    """amvhip_decode_batch_async, six calls in a row before one amvhip_sync: every call's frames come back on the copy
    stream out of one of two staging buffers used in turn, so the kernels of call k + 2 must wait for the copy of call k
    (events), and a grown staging buffer must not be freed under a copy.  Different content, sizes growing and
    shrinking from call to call (so that the staging buffers are re-allocated in mid-flight), page-locked and pageable
    destinations; then an encode through the same context's other host entry point, and a synchronous decode."""
    import torch
    lib = pkg.load_library()
    w, h = 160, 120
    base = _synth_chunks(orc, 8, w, h, first=500)
    want1 = [orc.decode_frame(c, w, h)[0] for c in base]
    counts = [3, 40, 7, 120, 1, 64]
    calls = []
    for k, n in enumerate(counts):
        chunks = [base[(k + i) % 8] for i in range(n)]
        blob, offs, lens, nbytes = _blob_of(chunks, pad_front=0)
        out = torch.empty((n, h, ctx.stride(w)), dtype=torch.uint8)
        out = out.pin_memory() if k % 2 == 0 else out
        out.fill_(0x5A)
        st = np.full(n, -1, np.int32)
        calls.append((chunks, blob, offs, lens, nbytes, out, st))
    for chunks, blob, offs, lens, nbytes, out, st in calls:
        rc = lib.amvhip_decode_batch_async(ctx.h, blob.ctypes.data, nbytes, offs.ctypes.data, lens.ctypes.data, len(chunks), w, h, 0,
                                           out.data_ptr(), st.ctypes.data)
        assert rc == 0, lib.amvhip_last_error(ctx.h)
    assert lib.amvhip_sync(ctx.h) == 0
    for k, (chunks, blob, offs, lens, nbytes, out, st) in enumerate(calls):
        got = out.numpy()
        assert (st == 0).all(), k
        for i in range(len(chunks)):
            assert (got[i] == want1[(k + i) % 8]).all(), (k, i)
    # the context's other host entry points still work behind that (they share its upload staging)
    src = np.stack([orc.synth_frame(SEED, 500 + t, w, h) for t in range(4)])
    eb = np.zeros(ctx.encode_bound(w, h) * 4, np.uint8)
    eo, el = np.zeros(4, np.uint64), np.zeros(4, np.uint32)
    ctx.encode_batch(src, w * 3, 0, 4, w, h, 0, eb, eb.size, eo, el)
    for t in range(4):
        assert eb[int(eo[t]):int(eo[t]) + int(el[t])].tobytes() == base[t], t
    out = np.zeros((8, h, ctx.stride(w)), np.uint8)
    st = np.full(8, -1, np.int32)
    blob, offs, lens, nbytes = _blob_of(base)
    ctx.decode_batch(blob, nbytes, offs, lens, 8, w, h, 0, out, st)
    assert (st == 0).all() and (out == np.stack(want1)).all()


# ---------------------------------------------------------------------------------- encode

@pytest.mark.parametrize("w,h,n,bgr,qbias", [(320, 240, 5, 0, 0), (160, 120, 66, 0, 0), (160, 120, 3, 1, 128),
                                             (176, 144, 2, 0, 128), (16, 16, 3, 0, 0), (336, 48, 2, 1, 0)])
def test_encode_matches_oracle(ctx, orc, w, h, n, bgr, qbias):
    """coefficient stage and final chunks are byte-identical to the oracle's encoder"""
    import torch
    src = np.stack([orc.synth_frame(SEED, 11 * t, w, h) for t in range(n)])
    if bgr:
        src = np.ascontiguousarray(src[..., ::-1])
    nblk = orc.nmcu(w, h) * 6
    d_src = _t(src)
    d_coef = torch.zeros((n, nblk, 64), dtype=torch.int16, device="cuda:0")
    ctx.encode_coefs_dev(d_src, w * 3, bgr, n, w, h, qbias, d_coef)
    cap = ctx.encode_bound(w, h) * n
    d_blob = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    d_offs = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    d_lens = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    ctx.encode_batch_dev(d_src, w * 3, bgr, n, w, h, qbias, d_blob, cap, d_offs, d_lens)
    torch.cuda.synchronize()
    coef, blob, offs, lens = d_coef.cpu().numpy(), d_blob.cpu().numpy(), d_offs.cpu().numpy(), d_lens.cpu().numpy()
    pos = 0
    for i in range(n):
        want_chunk, want_coef = orc.encode_frame(src[i], w, h, bgr=bool(bgr), qbias=qbias, want_coef=True)
        assert (coef[i] == want_coef).all(), i
        assert int(offs[i]) == pos and int(lens[i]) == len(want_chunk)
        assert blob[pos:pos + len(want_chunk)].tobytes() == want_chunk, i
        pos += len(want_chunk)


def test_encode_extreme_content(ctx, pkg, orc):
    """flat, saturated and pure-noise frames: long zero runs (ZRL), large coefficients, many FF bytes"""
    import torch
    w, h = 160, 120
    rng = np.random.default_rng(3)
    frames = [np.zeros((h, w, 3), np.uint8), np.full((h, w, 3), 255, np.uint8),
              rng.integers(0, 256, (h, w, 3)).astype(np.uint8),
              (rng.integers(0, 2, (h, w, 1)) * 255).astype(np.uint8).repeat(3, 2),
              np.indices((h, w)).sum(0).astype(np.uint8)[..., None].repeat(3, 2)]
    src = np.stack(frames)
    n = len(frames)
    cap = ctx.encode_bound(w, h) * n
    blob = np.zeros(cap, np.uint8)
    offs = np.zeros(n, np.uint64)
    lens = np.zeros(n, np.uint32)
    try:
        # the wave-per-frame coder (the noise frames overflow its LDS window and are handed back to the
        # lane-per-frame coder) and the lane-per-frame coder alone
        for mode in (pkg.ENTROPY_AUTO, pkg.ENTROPY_SERIAL):
            ctx.set_entropy_mode(mode)
            blob[:] = 0
            ctx.encode_batch(src, w * 3, 0, n, w, h, 0, blob, cap, offs, lens)
            for i in range(n):
                want = orc.encode_frame(src[i], w, h)
                assert blob[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes() == want, (mode, i)
    finally:
        ctx.set_entropy_mode(pkg.ENTROPY_AUTO)
    got, st = _gpu_decode(ctx, [blob[int(o):int(o) + int(l)].tobytes() for o, l in zip(offs, lens)], w, h, 1)
    assert (st == 0).all()


def test_encode_rejects_bad_arguments(ctx, pkg):
    import torch
    d = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(pkg.AmvHipError):
        ctx.encode_coefs_dev(d, 15 * 3, 0, 1, 15, 16, 0, d)        # odd width
    with pytest.raises(pkg.AmvHipError):
        ctx.encode_coefs_dev(d, 8, 0, 1, 16, 16, 0, d)             # stride < 3*w
    one = np.zeros(16, np.uint8)
    assert pkg.load_library().encode_amv_frame(one.ctypes.data, 3, 1, 1, 0, one.ctypes.data, 16) == -1


def test_round_trip_through_aliases(ctx, pkg, orc):
    lib = pkg.load_library()
    w, h = 320, 240
    src = orc.synth_frame(SEED, 77, w, h)
    buf = np.zeros(ctx.encode_bound(w, h), np.uint8)
    n = lib.encode_amv_frame(src.ctypes.data, w * 3, w, h, 0, buf.ctypes.data, buf.size)
    assert n > 0 and buf[:n].tobytes() == orc.encode_frame(src, w, h)
    out = np.zeros((h, w * 3), np.uint8)
    assert lib.decode_amv_frame(buf.ctypes.data, n, w, h, out.ctypes.data) == 0
    assert orc.psnr(src, out.reshape(h, w, 3)[:, :, ::-1]) > 26.0   # see test_encode_round_trip_quality


# ---------------------------------------------------------------------------------- ADPCM

def test_adpcm_decode_matches_oracle_and_reference_clip(ctx, orc, amv1):
    chunks = list(amv1["audio"]) + [b"", b"\x00" * 8, amv1["audio"][0][:9], bytes([0, 0, 200, 0, 0, 0, 0, 0]) + bytes(range(256))]
    blob, offs, lens, nbytes = _blob_of(chunks, 1)
    pcm_offs = np.cumsum([0] + [2 * max(len(c) - 8, 0) for c in chunks]).astype(np.uint64)
    pcm = np.full(int(pcm_offs[-1]) + 4, 0x5A5A, np.int16)
    fin = np.zeros((len(chunks), 2), np.int32)
    ctx.adpcm_decode_batch(blob, nbytes, offs, lens, len(chunks), pcm, pcm.size, pcm_offs[:-1].copy(), fin)
    for i, c in enumerate(chunks):
        want, _ = orc.adpcm_decode_chunk(c)
        got = pcm[int(pcm_offs[i]):int(pcm_offs[i + 1])]
        assert (got == want).all(), i
        if want.size:
            assert fin[i, 0] == want[-1]
    assert (pcm[int(pcm_offs[-1]):] == 0x5A5A).all()


def test_adpcm_encode_matches_oracle(ctx, orc):
    """reference behaviour (step_index carried through the stream) and the independent-chunk form"""
    rng = np.random.default_rng(21)
    n = 37
    sizes = [1378 + 2 * int(rng.integers(-3, 4)) for _ in range(n)]
    sizes[5] = 2
    sizes[6] = 0
    pcm_offs = np.cumsum([0] + sizes).astype(np.uint64)
    pcm = orc.synth_audio(SEED, 0, int(pcm_offs[-1]) + 2)
    pcm[3000:4000] = rng.integers(-32768, 32768, 1000)                 # loud noise: drives the index to 88
    pcm[9000:11000] = 0                                                # silence: drives it to 0
    offs = np.cumsum([0] + [8 + s // 2 for s in sizes]).astype(np.uint64)
    nsamp = np.array(sizes, np.uint32)
    blob = np.zeros(int(offs[-1]), np.uint8)
    ctx.adpcm_encode_batch(pcm, pcm.size, pcm_offs[:-1].copy(), nsamp, n, None, blob, blob.size, offs[:-1].copy())
    idx, starts = 0, []
    for i in range(n):
        starts.append(idx)
        seg = pcm[int(pcm_offs[i]):int(pcm_offs[i + 1])]
        want, idx = orc.adpcm_encode_chunk(seg if seg.size else np.zeros(1, np.int16), idx) if seg.size else (bytes(8), idx)
        if seg.size:
            assert blob[int(offs[i]):int(offs[i + 1])].tobytes() == want, i
    assert len(set(starts)) > 5
    step_in = rng.integers(0, 89, n).astype(np.int32)
    blob2 = np.zeros_like(blob)
    ctx.adpcm_encode_batch(pcm, pcm.size, pcm_offs[:-1].copy(), nsamp, n, step_in, blob2, blob2.size, offs[:-1].copy())
    for i in range(n):
        seg = pcm[int(pcm_offs[i]):int(pcm_offs[i + 1])]
        if seg.size:
            assert blob2[int(offs[i]):int(offs[i + 1])].tobytes() == orc.adpcm_encode_chunk(seg, int(step_in[i]))[0], i


def test_amvlib_adpcm_entry_points(ctx, pkg, orc, amv1):
    lib = pkg.load_library()
    a = amv1["audio"][3]
    c = pkg.ADPCMContext()
    c.channel = 1
    c.status[0].predictor = int(np.frombuffer(a[:2], "<i2")[0])
    c.status[0].step_index = a[2]
    n = len(a) - 8
    pcm = np.zeros(2 * (n + 4), np.int16)
    dl = ctypes.c_int(0)
    buf = np.frombuffer(a[8:], np.uint8).copy()
    rc = lib.AdpcmImaDecodeFrame(ctypes.byref(c), pcm.ctypes.data, ctypes.byref(dl), buf.ctypes.data, n)
    want, _ = orc.adpcm_decode_chunk(a)
    assert rc == (n + 3) // 4 * 4 and dl.value == 4 * rc          # the reference's 4-byte stride (AdpcmIma.c:225-241)
    assert (pcm[:want.size] == want).all()
    assert lib.AdpcmImaDecodeFrame(ctypes.byref(c), None, ctypes.byref(dl), buf.ctypes.data, n) == -1
    assert lib.AdpcmImaDecodeFrame(ctypes.byref(c), pcm.ctypes.data, ctypes.byref(dl), buf.ctypes.data, 0) == -1
    # amvlib's own (WAV-layout) encoder
    fs = 8 * 40
    x = (orc.synth_audio(SEED, 5000, fs + 1) * 3).astype(np.int16)
    e = pkg.ADPCMContext()
    e.status[0].step_index = 17
    out = np.zeros(4 + fs // 2, np.uint8)
    m = lib.AdpcmImaEncodeFrame(ctypes.byref(e), 1, fs, out.ctypes.data, out.size, x.ctypes.data)
    want = np.zeros_like(out)
    st = np.array([0, 17], np.int32)
    assert m == orc.lib().amvo_adpcm_wav_encode_frame(x.ctypes.data, fs, st.ctypes.data, want.ctypes.data)
    assert (out == want).all() and (e.status[0].prev_sample, e.status[0].step_index) == (st[0], st[1])


# ---------------------------------------------------------------------------------- amvlib surface

def test_amvlib_player_loop(ctx, pkg, orc, amv1):
    """the loop the reference's player runs (AMVDecoderDlg.cpp FillBuffer): read, video, audio"""
    lib = pkg.load_library()
    amv = lib.AmvOpen(amv1["path"].encode())
    assert amv
    d = amv.contents
    for k in range(6):
        assert lib.AmvReadNextFrame(amv) == 0
        assert lib.AmvVideoDecode(amv) == 0 and d.videobuf.len == 128 * 96 * 3
        got = np.frombuffer(ctypes.string_at(d.videobuf.fbmpdat, d.videobuf.len), np.uint8)
        assert (got == orc.decode_frame(amv1["video"][k], 128, 96)[0].ravel()).all()
        assert lib.AmvAudioDecode(amv) == 0
        want, _ = orc.adpcm_decode_chunk(amv1["audio"][k])
        pcm = np.frombuffer(ctypes.string_at(d.audiobuf.audiodata, d.audiobuf.len), np.int16)
        assert (pcm[:want.size] == want).all()
    info = pkg.AMVInfo()
    info.dwWidth, info.dwHeight = 128, 96
    fb = pkg.FRAMEBUFF()
    chunk = np.frombuffer(amv1["video"][9], np.uint8).copy()
    fb.videobuff = chunk.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
    fb.videobufflen = chunk.size
    vb = pkg.VIDEOBUFF()
    out = np.zeros(128 * 96 * 3, np.uint8)
    vb.fbmpdat = out.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
    vb.len = out.size
    lib.PrepareForVideoDecode(ctypes.byref(info))
    assert lib.AmvJpegDecode(ctypes.byref(info), ctypes.byref(fb), ctypes.byref(vb)) == 0
    assert (out == orc.decode_frame(amv1["video"][9], 128, 96)[0].ravel()).all()
    assert lib.AmvJpegDecode(None, ctypes.byref(fb), ctypes.byref(vb)) == -1
    lib.AmvClose(amv)


# ---------------------------------------------------------------------------------- sources / full size

def test_synthetic_sources_match_cpu_generator(ctx, orc):
    import torch
    for w, h, first, n in ((160, 120, 0, 3), (320, 240, 250, 2), (16, 16, 4000, 2)):
        d = torch.zeros((n, h, w, 3), dtype=torch.uint8, device="cuda:0")
        ctx.synth_frames_dev(SEED, first, n, w, h, d)
        got = d.cpu().numpy()
        for t in range(n):
            assert (got[t] == orc.synth_frame(SEED, first + t, w, h)).all()
    d = torch.zeros(5000, dtype=torch.int16, device="cuda:0")
    ctx.synth_audio_dev(SEED, 123456789, 5000, d)
    assert (d.cpu().numpy() == orc.synth_audio(SEED, 123456789, 5000)).all()


def test_full_size_stream_properties(ctx, orc):
    """BASELINE.json's 10 000-frame 160x120 stream, checked through size-independent properties:
    every frame decodes clean; decoding is independent of how the stream is batched (a checksum of
    per-frame checksums agrees between one 10k batch and ragged sub-batches); a sample of frames is
    bit-exact against the oracle; encode(GPU) of the same sources reproduces the chunk bytes."""
    import torch
    w, h, n = 160, 120, 10000
    dev = "cuda:0"
    s = torch.cuda.current_stream().cuda_stream
    d_src = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
    ctx.synth_frames_dev(SEED, 0, n, w, h, d_src, s)
    cap = 6000 * n
    d_blob = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_offs = torch.zeros(n, dtype=torch.int64, device=dev)
    d_lens = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.encode_batch_dev(d_src, w * 3, 0, n, w, h, 0, d_blob, cap, d_offs, d_lens, s)
    d_out = torch.empty((n, h, w * 3), dtype=torch.uint8, device=dev)
    d_st = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.decode_batch_dev(d_blob, cap, d_offs, d_lens, n, w, h, 0, d_out, d_st, s)
    torch.cuda.synchronize()
    lens, offs = d_lens.cpu().numpy(), d_offs.cpu().numpy()
    assert int(offs[-1]) + int(lens[-1]) <= cap and (d_st == 0).all()
    assert 0.12 < lens.mean() / (w * h) < 0.25                       # ~0.2 B/pixel, BASELINE.md section 4
    weights = torch.arange(1, h * w * 3 + 1, dtype=torch.int64, device=dev)
    sums = (d_out.view(n, -1).to(torch.int64) * weights).sum(1)
    # ragged re-batching: 1 + 63 + 64 + 65 + 4000 + rest
    d_out2 = torch.empty_like(d_out)
    pos = 0
    for cnt in (1, 63, 64, 65, 4000, n - 4193):
        ctx.decode_batch_dev(d_blob, cap, d_offs[pos:], d_lens[pos:], cnt, w, h, 0, d_out2[pos:], d_st[pos:], s)
        pos += cnt
    torch.cuda.synchronize()
    sums2 = (d_out2.view(n, -1).to(torch.int64) * weights).sum(1)
    assert torch.equal(sums, sums2) and (d_st == 0).all()
    blob = d_blob.cpu().numpy()
    for i in (0, 1, 63, 64, 4999, 9998, 9999):
        chunk = blob[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes()
        assert chunk == orc.encode_frame(orc.synth_frame(SEED, i, w, h), w, h)
        assert (d_out[i].cpu().numpy() == orc.decode_frame(chunk, w, h)[0]).all()


@pytest.mark.parametrize("w,h,n", [(160, 120, 6000), (160, 120, 12500), (160, 120, 30000), (320, 240, 6000), (320, 240, 12500)])
def test_batches_in_the_lane_tables_windows(ctx, pkg, orc, w, h, n):
    """batch sizes between the boundaries round 6 moved in huffman_sync_lanes' table (generations of 13 waves per unit;
    two lanes for short frames beyond eight lanes' residency): the table's choice decodes every frame to the bytes the
    serial entropy kernel gives -- compared frame by frame on the device --, every status clean, and a sample of frames is
    the oracle's"""
    import torch
    dev = "cuda:0"
    s = torch.cuda.current_stream().cuda_stream
    pool = 2000                                            # distinct frames, repeated through the batch at a stride
    d_src = torch.empty((pool, h, w, 3), dtype=torch.uint8, device=dev)
    ctx.synth_frames_dev(SEED, 0, pool, w, h, d_src, s)
    # (the batch's chunks overlap in the blob -- every pool chunk is used several times -- and the workspace is sized from the
    # blob's bytes, include/amvhip.h: the blob is given room for the sum of the lengths, or frames would be handed to the serial kernel)
    cap = pool * w * h + n * (w * h // 4)
    d_blob = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_poffs = torch.zeros(pool, dtype=torch.int64, device=dev)
    d_plens = torch.zeros(pool, dtype=torch.int32, device=dev)
    ctx.encode_batch_dev(d_src, w * 3, 0, pool, w, h, 0, d_blob, cap, d_poffs, d_plens, s)
    torch.cuda.synchronize()
    del d_src
    pick = (torch.arange(n, device=dev) * 7) % pool
    d_offs, d_lens = d_poffs[pick].contiguous(), d_plens[pick].contiguous()
    stride = ctx.stride(w)
    d_out = torch.empty((n, h, stride), dtype=torch.uint8, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ctx.decode_batch_dev(d_blob, cap, d_offs, d_lens, n, w, h, 0, d_out, d_st, s)
    torch.cuda.synchronize()
    assert int((d_st != 0).sum()) == 0 and ctx.entropy_stats(False)["handed_to_serial"] == 0
    # the serial kernel's bytes of the pool's frames, once; every frame of the batch against its pool frame's
    ctx.set_entropy_mode(pkg.ENTROPY_SERIAL)
    try:
        d_ref = torch.empty((pool, h, stride), dtype=torch.uint8, device=dev)
        d_rst = torch.full((pool,), -1, dtype=torch.int32, device=dev)
        ctx.decode_batch_dev(d_blob, cap, d_poffs, d_plens, pool, w, h, 0, d_ref, d_rst, s)
        torch.cuda.synchronize()
    finally:
        ctx.set_entropy_mode(pkg.ENTROPY_AUTO)
    assert int((d_rst != 0).sum()) == 0
    for lo in range(0, n, 4000):
        hi = min(n, lo + 4000)
        assert torch.equal(d_out[lo:hi], d_ref[pick[lo:hi]]), (w, h, n, lo)
    offs, lens = d_poffs.cpu().numpy(), d_plens.cpu().numpy()
    for i in (0, n // 2, n - 1):
        p = int(pick[i])
        chunk = d_blob[int(offs[p]):int(offs[p]) + int(lens[p])].cpu().numpy().tobytes()
        assert (d_out[i].cpu().numpy() == orc.decode_frame(chunk, w, h)[0]).all()


def test_adpcm_encode_long_stream_index_chain(ctx, orc):
    """the step index is carried through 700 chunks (three 256-chunk workgroup maps composed): every chunk
    equals the oracle's sequential encode, and decoding the device's chunks on the device equals the oracle's decode"""
    rng = np.random.default_rng(5)
    n = 700
    sizes = [1378 if k % 7 else 2 * int(rng.integers(1, 900)) for k in range(n)]
    pcm_offs = np.cumsum([0] + sizes).astype(np.uint64)
    pcm = orc.synth_audio(SEED, 12345, int(pcm_offs[-1]) + 2)
    pcm[200000:230000] = rng.integers(-30000, 30000, 30000)            # a loud stretch: the index climbs
    pcm[500000:520000] = 0                                             # silence: it falls back to 0
    offs = np.cumsum([0] + [8 + s // 2 for s in sizes]).astype(np.uint64)
    nsamp = np.array(sizes, np.uint32)
    blob = np.zeros(int(offs[-1]), np.uint8)
    ctx.adpcm_encode_batch(pcm, pcm.size, pcm_offs[:-1].copy(), nsamp, n, None, blob, blob.size, offs[:-1].copy())
    idx, seen = 0, set()
    for i in range(n):
        seen.add(idx)
        want, idx = orc.adpcm_encode_chunk(pcm[int(pcm_offs[i]):int(pcm_offs[i + 1])], idx)
        assert blob[int(offs[i]):int(offs[i + 1])].tobytes() == want, i
    assert len(seen) > 20
    lens = np.diff(offs).astype(np.uint32)
    out = np.zeros(int(pcm_offs[-1]) + 8, np.int16)
    fin = np.zeros((n, 2), np.int32)
    ctx.adpcm_decode_batch(blob, blob.size, offs[:-1].copy(), lens, n, out, out.size, pcm_offs[:-1].copy(), fin)
    for i in (0, 1, 255, 256, 257, 511, 512, 699):
        want, _ = orc.adpcm_decode_chunk(blob[int(offs[i]):int(offs[i + 1])])
        assert (out[int(pcm_offs[i]):int(pcm_offs[i + 1])] == want[: sizes[i]]).all(), i


def test_decode_fuzz_geometries_and_damage(ctx, orc):
    """random even geometries (partial MCUs on both edges, more than one segment per MCU row, frames larger
    than the synchronising kernel's LDS window next to tiny ones), random content and random damage:
    status and every output byte equal the oracle's"""
    rng = np.random.default_rng(77)
    for case in range(8):
        w = 2 * int(rng.integers(8, 200))
        h = 2 * int(rng.integers(8, 150))
        n = int(rng.integers(3, 40))
        chunks = []
        for t in range(n):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                src = orc.synth_frame(SEED, 1000 * case + t, w, h)
            elif kind == 1:
                src = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)                      # noise: long chunks
            elif kind == 2:
                src = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)              # flat: tiny chunks
            else:
                src = (orc.synth_frame(SEED, t, w, h).astype(np.int32) + rng.integers(-40, 41, (h, w, 3))).clip(0, 255).astype(np.uint8)
            c = bytearray(orc.encode_frame(src, w, h, qbias=int(rng.integers(0, 2)) * 128))
            if rng.random() < 0.3 and len(c) > 8:                                          # damage
                for _ in range(int(rng.integers(1, 4))):
                    c[int(rng.integers(2, len(c) - 2))] ^= 1 << int(rng.integers(0, 8))
            if rng.random() < 0.1:
                c = c[: int(rng.integers(2, len(c)))]                                      # truncation
            chunks.append(bytes(c))
        flags = case & 1
        got, st = _gpu_decode(ctx, chunks, w, h, flags, pad_front=int(rng.integers(0, 4)))
        want, wst = _oracle_decode(orc, chunks, w, h, flags)
        assert (st == wst).all(), (case, w, h, st, wst)
        assert (got == want).all(), (case, w, h)


def test_reconstruction_launch_in_parts_and_extreme_shapes(ctx, pkg, orc, monkeypatch):
    """the reconstruction's one-dimensional launch (amv_block_load.h: PieceMap): shapes whose piece counts are not
    powers of two (13 segments per MCU row; 40 MCU rows = 10 row groups; 9 rows = the five-row workgroups), and the
    same batches launched in parts of 3 frames (what a batch too large for the reciprocal division gets) -- every
    byte as the oracle's, in both output modes"""
    rng = np.random.default_rng(5)
    for w, h, n in ((2048, 16, 7), (16, 640, 7), (176, 144, 11), (320, 240, 10), (160, 120, 9)):
        chunks = [orc.encode_frame(orc.synth_frame(SEED, 3 * t, w, h), w, h) for t in range(n)]
        c = bytearray(chunks[n // 2])
        c[len(c) // 2] ^= 0x10                                    # one damaged frame
        chunks[n // 2] = bytes(c)
        for most in (None, "3"):
            if most:
                monkeypatch.setenv("AMVHIP_RECON_MOST", most)
            else:
                monkeypatch.delenv("AMVHIP_RECON_MOST", raising=False)
            got, st = _gpu_decode(ctx, chunks, w, h)              # (the knob is read at every launch)
            want, wst = _oracle_decode(orc, chunks, w, h)
            assert (st == wst).all() and (got == want).all(), (w, h, most)
            got, st = _gpu_decode_ffmpeg(ctx, pkg, chunks, w, h)
            want, wst = _oracle_decode_ffmpeg(orc, chunks, w, h)
            assert (st == wst).all() and (got == want).all(), (w, h, most, "compat")
    monkeypatch.delenv("AMVHIP_RECON_MOST", raising=False)


def test_encode_fuzz_geometries(ctx, orc):
    """random even geometries (right and bottom edge replication, widths that are not 0 mod 4 or mod 16),
    padded source rows, both channel orders and both quantiser biases: chunks byte-identical to the oracle's"""
    import torch
    rng = np.random.default_rng(99)
    sizes = [(18, 10), (30, 34), (150, 98), (162, 122)] + [(2 * int(rng.integers(4, 180)), 2 * int(rng.integers(4, 130))) for _ in range(6)]
    for case, (w, h) in enumerate(sizes):
        n = int(rng.integers(2, 12))
        bgr = case & 1
        qbias = 128 * ((case >> 1) & 1)
        stride = w * 3 + int(rng.integers(0, 3)) * 5
        src = np.zeros((n, h, stride), np.uint8)
        pix = rng.integers(0, 256, (n, h, w, 3)).astype(np.uint8) if case % 3 == 0 else \
            np.stack([orc.synth_frame(SEED, 31 * case + t, w, h) for t in range(n)])
        src[:, :, : w * 3] = pix.reshape(n, h, w * 3)
        src[:, :, w * 3:] = 0xEE                                              # must never be read as pixels
        cap = ctx.encode_bound(w, h) * n
        d_blob = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
        d_offs = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        d_lens = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        ctx.encode_batch_dev(_t(src), stride, bgr, n, w, h, qbias, d_blob, cap, d_offs, d_lens)
        torch.cuda.synchronize()
        blob, offs, lens = d_blob.cpu().numpy(), d_offs.cpu().numpy(), d_lens.cpu().numpy()
        for i in range(n):
            want = orc.encode_frame(np.ascontiguousarray(pix[i]), w, h, bgr=bool(bgr), qbias=qbias)
            assert int(lens[i]) == len(want) and blob[int(offs[i]):int(offs[i]) + len(want)].tobytes() == want, (w, h, i)


def test_amvlib_export_helpers(ctx, pkg, orc, amv1, tmp_path):
    """AmvCreateWavFileFromAmvFile (PCM: every chunk decoded on the GPU) and AmvConvertJpegFileToBmpFile
    (a still written by AmvCreateJpegFileFromFrameBuffer comes back as the decoded frame in a 24-bit BMP)"""
    lib = pkg.load_library()
    dec = lib.AmvOpen(amv1["path"].encode())
    assert dec
    wav = str(tmp_path / "pcm.wav")
    assert lib.AmvCreateWavFileFromAmvFile(dec, 0, wav.encode()) == 0
    w = open(wav, "rb").read()
    # what the reference's loop writes: audiobuf.len bytes of every AmvAudioDecode (AMVDec.c:512-519); each
    # starts with the chunk's defined samples (the oracle's), cf. test_amvlib_player_loop
    parts = []
    assert lib.AmvRewindFrameStart(dec) == 0
    for a in amv1["audio"]:
        assert lib.AmvReadNextFrame(dec) == 0 and lib.AmvAudioDecode(dec) == 0
        part = ctypes.string_at(dec.contents.audiobuf.audiodata, dec.contents.audiobuf.len)
        want = orc.adpcm_decode_chunk(a)[0].astype("<i2").tobytes()
        assert part[: len(want)] == want
        parts.append(part)
    pcm = b"".join(parts)
    le = lambda o, k=4: int.from_bytes(w[o:o + k], "little")
    assert w[:4] == b"RIFF" and le(4) == len(pcm) + 38 and w[8:16] == b"WAVEfmt " and le(16) == 18
    assert (le(20, 2), le(22, 2), le(24), le(28), le(32, 2), le(34, 2)) == (1, 1, 16000, 32000, 2, 16)
    assert w[38:42] == b"data" and le(42) == len(pcm) and w[46:] == pcm
    # stills -> BMP
    for k in (0, 100):
        assert lib.AmvRewindFrameStart(dec) == 0
        for _ in range(k + 1):
            assert lib.AmvReadNextFrame(dec) == 0
        base = str(tmp_path / ("f%d" % k))
        assert lib.AmvCreateJpegFileFromFrameBuffer(dec, base.encode()) == 0
        jpg = base + "-amvjpg_%06d_.jpg" % dec.contents.framebuf.framenum   # the counter runs on across rewinds (AMVDec.c:233,253)
        bmp = base + ".bmp"
        assert lib.AmvConvertJpegFileToBmpFile(jpg.encode(), bmp.encode()) == 0
        bmp2 = bmp + ".by_its_amvjpeg_name.bmp"                    # AmvJpeg.h:94: the same function under the name AmvJpeg.c gives it
        assert lib.ConvertJpegFileToBmpFile(jpg.encode(), bmp2.encode()) == 0 and open(bmp2, "rb").read() == open(bmp, "rb").read()
        b = open(bmp, "rb").read()
        want, st, _ = orc.decode_frame(amv1["video"][k], 128, 96)
        assert st == 0 and b[:2] == b"BM" and int.from_bytes(b[2:6], "little") == len(b) == 54 + want.size
        assert int.from_bytes(b[10:14], "little") == 54 and int.from_bytes(b[14:18], "little") == 40
        assert (int.from_bytes(b[18:22], "little"), int.from_bytes(b[22:26], "little"), b[26], b[28]) == (128, 96, 1, 24)
        assert b[54:] == want.tobytes()
    assert lib.AmvConvertJpegFileToBmpFile(amv1["path"].encode(), (str(tmp_path / "x.bmp")).encode()) == -1   # not such a JPEG
    lib.AmvClose(dec)


@pytest.mark.parametrize("w,h,n", [(640, 480, 3), (1024, 768, 2)])
def test_decode_large_frames(ctx, orc, w, h, n):
    """640x480: long streams through the per-lane windows (hundreds of refills per lane); 1024x768: more blocks
    than a record's block field holds, so the whole batch takes the lane-per-frame kernel"""
    chunks = _synth_chunks(orc, n, w, h)
    damaged = bytearray(chunks[-1])
    damaged[len(damaged) // 2] ^= 0x10
    chunks.append(bytes(damaged))
    got, st = _gpu_decode(ctx, chunks, w, h)
    want, wst = _oracle_decode(orc, chunks, w, h)
    assert (st == wst).all() and (wst[:n] == 0).all()
    assert (got == want).all()


def test_c_host_links_and_matches_amvlib(pkg, amv1, tmp_path):
    """a plain C program in the shape of the reference's AmvLibTest.cpp, compiled with gcc against include/amvhip.h and
    linked with libamvhip.so, decodes the reference clip to the very hash amvlib produced"""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "amvlib_host")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "amvlib_host.c"),
                    "-L", libdir, "-l:" + os.path.basename(pkg.LIB_PATH), "-Wl,-rpath," + libdir, "-o", exe], check=True)
    wav = str(tmp_path / "out.wav")
    out = subprocess.run([exe, amv1["path"], wav], check=True, capture_output=True, text=True, timeout=300).stdout
    fields = dict(line.split(": ", 1) for line in out.strip().splitlines())
    assert fields["size"] == "128 x 96" and fields["speed"] == "12 frames/s" and fields["total frames"] == "252"
    assert fields["decoded frames"] == "252" and fields["pcm bytes"] == "672504"        # the byte count the survey recorded
    assert fields["video chunk bytes"] == str(sum(map(len, amv1["video"])))
    assert int(fields["video fnv1a64"], 16) == AMVLIB_HASH
    assert os.path.getsize(wav) == 52 + sum(len(a) - 8 for a in amv1["audio"])


def test_coresident_video_and_adpcm_streams(pkg, orc):
    """BASELINE.json configs[4]: 320x240 video decode on HIP stream A while IMA-ADPCM encode (step index carried)
    + decode of the frames' audio chunks run on stream B -- two contexts (one per stream, as include/amvhip.h
    prescribes for work that overlaps), enqueued back to back with no synchronisation in between, twice over so
    that the second round's kernels meet the first round's on the device.  EVERY frame, chunk and sample is
    compared with the oracle."""
    import torch
    dev = "cuda:0"
    w, h, n, spf = 320, 240, 192, 1378
    vctx, actx = pkg.Context(0), pkg.Context(0)
    blob, offs, lens = orc.synth_stream(SEED, 500, n, w, h, threads=8)
    want_frames, want_st = orc.decode_batch(blob, offs, lens, w, h, 0, threads=8)
    pcm = orc.synth_audio(SEED, 777, n * spf)
    clen = 8 + spf // 2
    want_chunks, idx = [], 0
    for i in range(n):
        c, idx = orc.adpcm_encode_chunk(pcm[i * spf:(i + 1) * spf], idx)
        want_chunks.append(c)
    want_pcm = np.concatenate([orc.adpcm_decode_chunk(c)[0][:spf] for c in want_chunks])

    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    d_blob, d_offs, d_lens = _t(blob), _t(offs), _t(lens)
    d_pcm = _t(pcm)
    d_pcm_offs = torch.arange(n, dtype=torch.int64, device=dev) * spf
    d_nsamp = torch.full((n,), spf, dtype=torch.int32, device=dev)
    d_aoffs = torch.arange(n, dtype=torch.int64, device=dev) * clen
    d_alens = torch.full((n,), clen, dtype=torch.int32, device=dev)
    outs = []
    torch.cuda.synchronize()
    for rnd in range(2):
        d_out = torch.full((n, h, vctx.stride(w)), 0x5A, dtype=torch.uint8, device=dev)
        d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
        d_chunks = torch.zeros(n * clen + 16, dtype=torch.uint8, device=dev)
        d_pcm2 = torch.zeros(n * spf + 8, dtype=torch.int16, device=dev)
        outs.append((d_out, d_st, d_chunks, d_pcm2))
    torch.cuda.synchronize()
    for d_out, d_st, d_chunks, d_pcm2 in outs:        # no host synchronisation inside this loop
        vctx.decode_batch_dev(d_blob, blob.size, d_offs, d_lens, n, w, h, 0, d_out, d_st, sa.cuda_stream)
        actx.adpcm_encode_batch_dev(d_pcm, d_pcm_offs, d_nsamp, n, None, d_chunks, d_aoffs, sb.cuda_stream)
        actx.adpcm_decode_batch_dev(d_chunks, n * clen, d_aoffs, d_alens, n, d_pcm2, d_pcm_offs, None, sb.cuda_stream)
    torch.cuda.synchronize()
    for d_out, d_st, d_chunks, d_pcm2 in outs:
        assert (d_st.cpu().numpy() == want_st).all() and (want_st == 0).all()
        assert (d_out.cpu().numpy() == want_frames).all()
        assert d_chunks[: n * clen].cpu().numpy().tobytes() == b"".join(want_chunks)
        assert (d_pcm2[: n * spf].cpu().numpy() == want_pcm).all()
    vctx.close()
    actx.close()


# ---------------------------------------------------------------------------------- FFmpeg-compat mode

def _gpu_decode_ffmpeg(ctx, pkg, chunks, w, h, pad_front=0):
    import torch
    blob, offs, lens, nbytes = _blob_of(chunks, pad_front)
    n = len(chunks)
    fb = ctx.yuv420_frame_bytes(w, h)
    d_out = torch.full((n, fb), 0x5A, dtype=torch.uint8, device="cuda:0")
    d_st = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    ctx.decode_batch_dev(_t(blob), nbytes, _t(offs), _t(lens), n, w, h, pkg.FLAG_FFMPEG, d_out, d_st,
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), d_st.cpu().numpy()


def _oracle_decode_ffmpeg(orc, chunks, w, h):
    outs, sts = [], []
    for c in chunks:
        o, st, _ = orc.decode_frame_ffmpeg(c, w, h)
        outs.append(o)
        sts.append(st)
    return np.stack(outs), np.array(sts, np.int32)


def test_ffmpeg_compat_decode_amv1(ctx, pkg, orc, amv1):
    """rows a14/a15: AMVHIP_FLAG_FFMPEG on all 252 chunks of the reference's fixture == the oracle's restatement
    of the patched FFmpeg's amv_decoder (whose IDCT and tables are pinned by the reference's own objects)"""
    got, st = _gpu_decode_ffmpeg(ctx, pkg, amv1["video"], 128, 96, pad_front=1)
    want, want_st = _oracle_decode_ffmpeg(orc, amv1["video"], 128, 96)
    assert (st == 0).all() and (want_st == 0).all()
    assert (got == want).all()


def test_ffmpeg_keep_leaves_what_mjpegdec_leaves(pkg, orc, amv1):
    """AMVHIP_FLAG_FFMPEG_KEEP (mjpegdec.c:699-716: mjpeg_decode_scan returns at the failing block, the blocks before it are
    in the picture, the rest of the picture is as it was): damaged and truncated chunks of the reference's clip and of two
    synthetic geometries (a height whose bottom rows the flip formula never reaches among them) decoded over a buffer the
    caller filled -- every byte as the oracle's restatement leaves it: blocks in front of the failing one written, those
    of the failing MCU included, every other byte the caller's; undamaged frames beside them in the same batch.  Through
    the parallel entropy kernels (several lane counts), the serial one, and the host-buffer entry point; the flag without
    AMVHIP_FLAG_FFMPEG is refused."""
    import os
    import torch
    rng = np.random.default_rng(4242)
    cases = [(128, 96, list(amv1["video"][:60]))]
    for w, h in ((160, 120), (46, 30)):
        cases.append((w, h, [orc.encode_frame(orc.synth_frame(SEED, 7 * t, w + (w & 1), h + (h & 1)), w + (w & 1), h + (h & 1)) for t in range(30)]))
    keep = {k: os.environ.get(k) for k in ("AMVHIP_SYNC_LANES",)}
    ctxs = {}
    try:
        for lanes in (None, "1", "4", "64"):
            if lanes is None:
                os.environ.pop("AMVHIP_SYNC_LANES", None)
            else:
                os.environ["AMVHIP_SYNC_LANES"] = lanes
            ctxs[lanes] = pkg.Context(0)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        for w, h, chunks in cases:
            damaged = []
            for i, c in enumerate(chunks):
                c = bytearray(c)
                if i % 3 != 0 and len(c) > 40:                      # flipped bits somewhere in the scan (one alone usually resynchronises)
                    for _ in range(3):
                        c[int(rng.integers(8, len(c) - 4))] ^= 1 << int(rng.integers(0, 8))
                if i % 3 == 2:                                       # ... and the chunk cut short
                    c = c[: int(rng.integers(6, len(c) - 2))]
                damaged.append(bytes(c))
            fb = ctxs[None].yuv420_frame_bytes(w, h)
            before = rng.integers(0, 256, (len(damaged), fb), dtype=np.uint8)
            want, want_st, part = [], [], 0
            for i, c in enumerate(damaged):
                o, st, blocks = orc.decode_frame_ffmpeg_keep(c, w, h, before[i])
                want.append(o)
                want_st.append(st)
                part += 1 if (st and blocks % 6) else 0
                if not st:                                           # an undamaged frame: the plain mode's bytes wherever it writes
                    plain = orc.decode_frame_ffmpeg(c, w, h)[0]
                    changed = o != before[i]
                    assert (o[changed] == plain[changed]).all()
            want, want_st = np.stack(want), np.array(want_st, np.int32)
            assert (want_st != 0).sum() >= 3 and part >= 1, (w, h, want_st, part)   # some fail inside an MCU
            blob, offs, lens, nbytes = _blob_of(damaged, 1)
            for key, c in ctxs.items():
                for mode in ((pkg.ENTROPY_AUTO, pkg.ENTROPY_SERIAL) if key is None else (pkg.ENTROPY_AUTO,)):
                    c.set_entropy_mode(mode)
                    d_out = torch.from_numpy(before.copy()).to("cuda:0")
                    d_st = torch.full((len(damaged),), -1, dtype=torch.int32, device="cuda:0")
                    c.decode_batch_dev(_t(blob), nbytes, _t(offs), _t(lens), len(damaged), w, h, pkg.FLAG_FFMPEG | pkg.FLAG_FFMPEG_KEEP,
                                       d_out, d_st, torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    assert (d_st.cpu().numpy() == want_st).all(), (w, h, key, mode)
                    got = d_out.cpu().numpy()
                    bad = np.nonzero((got != want).any(axis=1))[0]
                    assert bad.size == 0, (w, h, key, mode, bad[:5])
                    c.set_entropy_mode(pkg.ENTROPY_AUTO)
            # the host-buffer form
            out = before.copy()
            st = np.full(len(damaged), -1, np.int32)
            ctxs[None].decode_batch(blob, nbytes, offs, lens, len(damaged), w, h, pkg.FLAG_FFMPEG | pkg.FLAG_FFMPEG_KEEP, out, st)
            assert (st == want_st).all() and (out == want).all()     # (the caller's frames go up to the staging buffer first)
        with pytest.raises(Exception):
            d_out = torch.zeros((1, ctxs[None].frame_bytes(128, 96)), dtype=torch.uint8, device="cuda:0")
            d_st = torch.zeros(1, dtype=torch.int32, device="cuda:0")
            ctxs[None].decode_batch_dev(_t(blob), nbytes, _t(offs), _t(lens), 1, 128, 96, pkg.FLAG_FFMPEG_KEEP, d_out, d_st,
                                        torch.cuda.current_stream().cuda_stream)
    finally:
        for c in ctxs.values():
            c.close()


@pytest.mark.parametrize("w,h,n", [(160, 120, 40), (320, 240, 6), (176, 144, 3), (130, 98, 3), (16, 16, 2), (336, 32, 2),
                                   (46, 30, 3), (33, 31, 3), (640, 480, 2)])
def test_ffmpeg_compat_decode_matches_oracle(ctx, pkg, orc, w, h, n):
    """synthetic clips incl. partial MCUs, heights the flip formula of mjpegdec.c:672-677 shifts (98) or leaves
    partly unwritten (30, 31), odd sizes (chroma planes of (w+1)/2 x (h+1)/2), several segments per MCU row"""
    ew, eh = w + (w & 1), h + (h & 1)           # the oracle's encoder wants even sizes; decode the same scan as w x h
    chunks = [orc.encode_frame(orc.synth_frame(SEED, 3 + t, ew, eh), ew, eh) for t in range(n)]
    rng = np.random.default_rng(w * h)
    noise = rng.integers(0, 256, (eh, ew, 3), dtype=np.uint8)
    chunks.append(orc.encode_frame(noise, ew, eh))
    if (ew + 15) // 16 != (w + 15) // 16 or (eh + 15) // 16 != (h + 15) // 16:
        pytest.skip("geometry changes the MCU grid")
    got, st = _gpu_decode_ffmpeg(ctx, pkg, chunks, w, h)
    want, want_st = _oracle_decode_ffmpeg(orc, chunks, w, h)
    assert (st == want_st).all() and (want_st == 0).all()
    assert (got == want).all()


def test_ffmpeg_compat_errors_saturation_and_kernels(ctx, pkg, orc, amv1):
    """damaged chunks (statuses and the zeroed remainder as in the amvlib mode), coefficients that wrap the int16
    stores of the reference's DCTELEM blocks, both entropy kernels, the host-buffer entry point"""
    import torch
    w, h = 160, 120
    good = [orc.encode_frame(orc.synth_frame(SEED, t, w, h), w, h) for t in range(6)]
    rng = np.random.default_rng(11)
    chunks = list(good)
    for c in good[:4]:
        b = bytearray(c)
        for _ in range(3):
            b[int(rng.integers(2, len(b) - 2))] ^= 1 << int(rng.integers(0, 8))
        chunks.append(bytes(b))
    chunks += [good[0][: len(good[0]) // 2], good[1][:40], b"\xff\xd8\xff\xd9", b"\xff\xd8" + b"\xff\x00" * 400 + b"\xff\xd9",
               b"\xff\xd8" + bytes(rng.integers(0, 255, 3000, dtype=np.uint8)) + b"\xff\xd9"]
    want, want_st = _oracle_decode_ffmpeg(orc, chunks, w, h)
    assert (want_st != 0).sum() >= 4
    for mode in (pkg.ENTROPY_AUTO, pkg.ENTROPY_SERIAL):
        ctx.set_entropy_mode(mode)
        try:
            got, st = _gpu_decode_ffmpeg(ctx, pkg, chunks, w, h, pad_front=3)
        finally:
            ctx.set_entropy_mode(pkg.ENTROPY_AUTO)
        assert (st == want_st).all()
        assert (got == want).all()
    # host buffers
    blob, offs, lens, nbytes = _blob_of(chunks)
    out = np.zeros((len(chunks), ctx.yuv420_frame_bytes(w, h)), np.uint8)
    st = np.zeros(len(chunks), np.int32)
    ctx.decode_batch(blob, nbytes, offs, lens, len(chunks), w, h, pkg.FLAG_FFMPEG, out, st)
    assert (out == want).all() and (st == want_st).all()
    # stage access with hand-made coefficients: extremes wrap exactly as the reference's int16 blocks do
    nm = 80
    coef = rng.integers(-2047, 2048, (3, nm * 6, 64)).astype(np.int16)
    coef[1] = (coef[1] * (rng.random(coef[1].shape) < 0.05)).astype(np.int16)     # sparse: DC-only rows take the shortcut
    coef[2, :, 1:] = 0                                                            # DC only
    d_out = torch.zeros((3, ctx.yuv420_frame_bytes(w, h)), dtype=torch.uint8, device="cuda:0")
    ctx.reconstruct_dev(_t(coef), _t(np.full(3, nm, np.uint32)), 3, w, h, pkg.FLAG_FFMPEG, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    L = orc.lib()
    for f in range(3):
        Y, Cb, Cr = orc.yuv_planes(got[f], w, h)
        for m in (0, 7, 33, 79):
            my, mx = divmod(m, 10)
            for k in range(6):
                blk = np.zeros(64, np.int16)
                px = np.zeros(64, np.uint8)
                L.amvo_ffmpeg_dequant_block(coef[f, m * 6 + k].ctypes.data, 0 if k < 4 else k - 3, blk.ctypes.data)
                L.amvo_simple_idct_put(px.ctypes.data, 8, blk.ctypes.data)
                plane, v = (Y, 2) if k < 4 else ((Cb, 1) if k == 4 else (Cr, 1))
                by, bx = (2 * my + (k >> 1), 2 * mx + (k & 1)) if k < 4 else (my, mx)
                start = v * (8 * 8 - ((h // 2) & 7)) - 1
                for i in range(8):
                    p = start - (8 * by + i)
                    if 0 <= p < plane.shape[0]:
                        assert (plane[p, 8 * bx: 8 * bx + 8] == px[8 * i: 8 * i + 8]).all(), (f, m, k, i)


# ---------------------------------------------------------------------------------- FFmpeg AVCodec plugin surface

def test_encode_yuv420_entry_matches_rgb_path(ctx, orc):
    """the YUVJ420P-input encode entry (what amv_encoder takes, mjpegenc.c:493): the oracle's rgb24_to_yuvj420p of a
    source followed by this entry gives the oracle encoder's chunk of the RGB source, byte for byte -- sizes with
    partial MCUs, padded plane strides, device and host forms"""
    import torch
    L = orc.lib()
    for w, h, n in ((160, 120, 5), (320, 240, 3), (130, 98, 3), (16, 16, 2)):
        cw, ch = w // 2, h // 2
        ys, cs = w + 24, cw + 8                                    # padded rows
        Y = np.full((n, h, ys), 0xEE, np.uint8)
        Cb = np.full((n, ch, cs), 0xEE, np.uint8)
        Cr = np.full((n, ch, cs), 0xEE, np.uint8)
        want = []
        for t in range(n):
            src = orc.synth_frame(SEED, 40 + t, w, h)
            y, cb, cr = np.zeros((h, w), np.uint8), np.zeros((ch, cw), np.uint8), np.zeros((ch, cw), np.uint8)
            L.amvo_rgb24_to_yuvj420p(src.ctypes.data, w * 3, w, h, 0, y.ctypes.data, cb.ctypes.data, cr.ctypes.data)
            Y[t, :, :w], Cb[t, :, :cw], Cr[t, :, :cw] = y, cb, cr
            want.append(orc.encode_frame(src, w, h))
        cap = ctx.encode_bound(w, h) * n
        d_blob = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
        d_offs = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        d_lens = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        ctx.encode_yuv420_batch_dev(_t(Y), _t(Cb), _t(Cr), ys, cs, h * ys, ch * cs, n, w, h, 0, d_blob, cap, d_offs, d_lens)
        torch.cuda.synchronize()
        blob, offs, lens = d_blob.cpu().numpy(), d_offs.cpu().numpy(), d_lens.cpu().numpy()
        for t in range(n):
            assert blob[int(offs[t]):int(offs[t]) + int(lens[t])].tobytes() == want[t], (w, h, t)
        blob2, offs2, lens2 = np.zeros(cap, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
        ctx.encode_yuv420_batch(Y, Cb, Cr, ys, cs, h * ys, ch * cs, n, w, h, 0, blob2, cap, offs2, lens2)
        assert (lens2 == lens.astype(np.uint32)).all() and blob2[: int(offs2[-1] + lens2[-1])].tobytes() == b"".join(want)


def test_encode_reports_blob_overflow(ctx, pkg, orc):
    """a blob smaller than the chunks need: the device form marks the chunks it could not write with length 0,
    the host form returns AMVHIP_ERR_SPACE"""
    import torch
    w, h, n = 160, 120, 6
    src = np.stack([orc.synth_frame(SEED, t, w, h) for t in range(n)])
    want = [orc.encode_frame(src[t], w, h) for t in range(n)]
    cap = len(want[0]) + len(want[1]) + len(want[2]) + 10        # room for three chunks and a bit
    d_blob = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    d_offs = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    d_lens = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    ctx.encode_batch_dev(_t(src), w * 3, 0, n, w, h, 0, d_blob, cap, d_offs, d_lens)
    torch.cuda.synchronize()
    lens, offs, blob = d_lens.cpu().numpy(), d_offs.cpu().numpy(), d_blob.cpu().numpy()
    assert [int(x) for x in lens] == [len(want[0]), len(want[1]), len(want[2]), 0, 0, 0]
    assert [int(x) for x in offs] == list(np.cumsum([0] + [len(c) for c in want[:-1]]))
    assert blob[: sum(map(len, want[:3]))].tobytes() == b"".join(want[:3])
    with pytest.raises(pkg.AmvHipError, match="-4"):
        ctx.encode_batch(src, w * 3, 0, n, w, h, 0, np.zeros(cap, np.uint8), cap, np.zeros(n, np.uint64), np.zeros(n, np.uint32))


def test_audio_framing_helpers_and_single_chunk_encoder(ctx, pkg, orc):
    """adpcm.c:469-477 framing and amvenc.c:276-281 frame_size in the product == the oracle's restatement; the
    one-chunk encoder hands the step index in and out exactly as the sequential reference does"""
    lib = pkg.load_library()
    assert lib.amvhip_amv_audio_frame_size(22050, 1, 16) == 1378 and lib.amvhip_amv_audio_frame_size(16000, 1, 12) == 1333
    L = orc.lib()
    for fs, rate in ((1378, 22050), (1471, 22050), (1333, 16000), (735, 22050)):
        e1, w1, e2, w2 = ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint32(0), ctypes.c_uint64(0)
        for _ in range(200):
            assert lib.amvhip_amv_audio_pairs(fs, rate, ctypes.byref(e1), ctypes.byref(w1)) == \
                L.amvo_adpcm_amv_pairs(fs, rate, ctypes.byref(e2), ctypes.byref(w2))
            assert (e1.value, w1.value) == (e2.value, w2.value)
    pcm = orc.synth_audio(SEED, 5000, 30 * 1378)
    idx, want_idx = ctypes.c_int32(0), 0
    for i in range(30):
        seg = np.ascontiguousarray(pcm[i * 1378:(i + 1) * 1378])
        out = np.zeros(8 + 689, np.uint8)
        n = lib.amvhip_adpcm_encode_frame(ctx.h, seg.ctypes.data, 1378, ctypes.byref(idx), out.ctypes.data, out.size)
        want, want_idx = orc.adpcm_encode_chunk(seg, want_idx)
        assert n == len(want) and out.tobytes() == want and idx.value == want_idx, i


def test_avcodec_plugin_through_the_struct(ctx, pkg, orc, amv1, tmp_path):
    """rows b2 / f2: a C program (tests/c/lavc_host.c, compiled against the reference's avcodec.h) calls
    amv_decoder, amv_encoder, adpcm_ima_amv_decoder and adpcm_ima_amv_encoder of libamvhip_lavc.so through
    `struct AVCodec` and must reproduce the batch ABI's and the oracle's bytes"""
    import os
    import struct
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "c", "_bin", "lavc_host")
    plugin = os.path.join(os.path.dirname(pkg.LIB_PATH), "libamvhip_lavc.so")
    assert os.path.exists(exe) and os.path.exists(plugin), "built by amv-codec-tools_amd/build.py where /root/reference exists"
    out = subprocess.run([exe, amv1["path"], str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    f = dict(line.split(": ", 1) for line in out.stdout.strip().splitlines() if ": " in line)
    assert f["decoded frames"] == "252" and f["pix_fmt is yuvj420p"] == "1" and f["emu_edge rejected"] == "1"
    assert f["get_buffer calls"] == "252" and f["release_buffer calls"] == "252" and f["11025 Hz rejected"] == "1"
    rd = lambda name: open(os.path.join(str(tmp_path), name), "rb").read()
    # decode leg: video == FFmpeg-compat batch decode == oracle; audio == oracle
    got_v = np.frombuffer(rd("dec_video.yuv"), np.uint8).reshape(252, -1)
    batch, st = _gpu_decode_ffmpeg(ctx, pkg, amv1["video"], 128, 96)
    assert (st == 0).all() and (got_v == batch).all()
    assert (got_v[::25] == _oracle_decode_ffmpeg(orc, amv1["video"][::25], 128, 96)[0]).all()
    want_a = np.concatenate([orc.adpcm_decode_chunk(a)[0] for a in amv1["audio"]])
    assert (np.frombuffer(rd("dec_audio.pcm"), np.int16) == want_a).all()
    # video encode leg: == batch ABI on the same planes; every chunk decodes clean on the amvlib-pinned oracle
    w, h, n = 160, 120, 24
    src = np.frombuffer(rd("enc_src.yuv"), np.uint8).reshape(n, -1)
    Y = np.ascontiguousarray(src[:, : w * h]).reshape(n, h, w)
    Cb = np.ascontiguousarray(src[:, w * h: w * h + w * h // 4]).reshape(n, h // 2, w // 2)
    Cr = np.ascontiguousarray(src[:, w * h + w * h // 4:]).reshape(n, h // 2, w // 2)
    cap = ctx.encode_bound(w, h) * n
    blob, offs, lens = np.zeros(cap, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    ctx.encode_yuv420_batch(Y, Cb, Cr, w, w // 2, w * h, w * h // 4, n, w, h, 0, blob, cap, offs, lens)
    raw, pos = rd("enc_video.bin"), 0
    for t in range(n):
        (ln,) = struct.unpack_from("<I", raw, pos)
        chunk = raw[pos + 4: pos + 4 + ln]
        pos += 4 + ln
        assert chunk == blob[int(offs[t]):int(offs[t]) + int(lens[t])].tobytes(), t
        dec, dst, _ = orc.decode_frame(chunk, w, h)
        assert dst == 0
        if t == 0:   # luma of the decoded picture is close to the source plane (different colour matrices: loose bound)
            yy = (dec[:, : w * 3].reshape(h, w, 3).astype(np.int32) * np.array([117, 601, 306])).sum(2) >> 10
            assert np.abs(yy - Y[0].astype(np.int32)).mean() < 12
    assert pos == len(raw)
    # ... and with PIX_FMT_YUVJ422P (the other entry of pix_fmts): the chunk of the 4:2:0 picture the chroma rows average to
    assert f["pix_fmts"] == "1 1 1" and f["encoded 422 frames"] == "6"
    n2 = 6
    src = np.frombuffer(rd("enc_src422.yuv"), np.uint8).reshape(n2, -1)
    Y = np.ascontiguousarray(src[:, : w * h]).reshape(n2, h, w)
    C = src[:, w * h:].reshape(n2, 2, h, w // 2).astype(np.uint16)
    C420 = ((C[:, :, 0::2] + C[:, :, 1::2] + 1) >> 1).astype(np.uint8)
    cap = ctx.encode_bound(w, h) * n2
    blob, offs, lens = np.zeros(cap, np.uint8), np.zeros(n2, np.uint64), np.zeros(n2, np.uint32)
    ctx.encode_yuv420_batch(Y, np.ascontiguousarray(C420[:, 0]), np.ascontiguousarray(C420[:, 1]), w, w // 2, w * h, w * h // 4, n2, w, h, 0,
                            blob, cap, offs, lens)
    raw, pos = rd("enc_video422.bin"), 0
    for t in range(n2):
        (ln,) = struct.unpack_from("<I", raw, pos)
        assert raw[pos + 4: pos + 4 + ln] == blob[int(offs[t]):int(offs[t]) + int(lens[t])].tobytes(), t
        assert orc.decode_frame(raw[pos + 4: pos + 4 + ln], w, h)[1] == 0
        pos += 4 + ln
    assert pos == len(raw)
    # audio encode leg: both passes == the oracle's sequential encoder fed by the oracle's framing
    pcm = np.frombuffer(rd("enc_pcm.raw"), np.int16)
    raw, pos = rd("enc_audio.bin"), 0
    L = orc.lib()
    for fs in (1378, 1471):
        e, wr, p, idx, sizes = ctypes.c_uint32(0), ctypes.c_uint64(0), 0, 0, set()
        for i in range(40):
            npairs = L.amvo_adpcm_amv_pairs(fs, 22050, ctypes.byref(e), ctypes.byref(wr))
            want, idx = orc.adpcm_encode_chunk(np.ascontiguousarray(pcm[p:p + 2 * npairs]), idx)
            (ln,) = struct.unpack_from("<I", raw, pos)
            assert raw[pos + 4: pos + 4 + ln] == want, (fs, i)
            pos += 4 + ln
            p += 2 * npairs
            sizes.add(npairs)
        assert len(sizes) >= 2          # the odd-sample carry / the 1 Hz resync really changed chunk sizes
    assert pos == len(raw)


def test_amvlib_readahead_semantics(ctx, pkg, orc, amv1, tmp_path):
    """the read-ahead window behind AmvReadNextFrame changes nothing a caller can see: every frame and every PCM
    chunk of the clip (window boundaries included) equals the oracle; rewinding mid-window, decoding a frame twice,
    editing framebuf (falls back to the single-chunk path), a file cut in the middle of a chunk"""
    import os
    lib = pkg.load_library()
    want_v = [orc.decode_frame(c, 128, 96)[0].ravel() for c in amv1["video"]]

    def frame(d):
        return np.frombuffer(ctypes.string_at(d.videobuf.fbmpdat, d.videobuf.len), np.uint8)

    for window in ("7", "256", "1"):
        os.environ["AMVHIP_READAHEAD"] = window
        try:
            amv = lib.AmvOpen(amv1["path"].encode())
            d = amv.contents
            held_v = held_a = None
            for k in range(252):
                assert lib.AmvReadNextFrame(amv) == 0 and d.framebuf.framenum == k + 1
                # what the decode calls handed out for the frame before (pointers into the read-ahead windows since round
                # 4) is the caller's until its NEXT decode call of that kind: reading on -- window boundaries included --
                # has not touched it
                if held_v is not None:
                    assert ctypes.string_at(held_v[0], len(held_v[1])) == held_v[1], (window, k)
                    assert ctypes.string_at(held_a[0], len(held_a[1])) == held_a[1], (window, k)
                assert ctypes.string_at(d.framebuf.videobuff, d.framebuf.videobufflen) == amv1["video"][k]
                assert ctypes.string_at(d.framebuf.audiobuff, d.framebuf.audiobufflen) == amv1["audio"][k]
                assert lib.AmvVideoDecode(amv) == 0 and (frame(d) == want_v[k]).all(), (window, k)
                assert lib.AmvAudioDecode(amv) == 0
                a = amv1["audio"][k]
                n4 = (len(a) - 8 + 3) & ~3
                assert d.audiobuf.len == 4 * n4
                pcm = np.frombuffer(ctypes.string_at(d.audiobuf.audiodata, d.audiobuf.len), np.int16)
                assert (pcm == orc.adpcm_decode_chunk(a + b"\0" * (n4 - (len(a) - 8)))[0]).all(), (window, k)
                if k == 100:   # again, and with the chunk edited in place: no stale result
                    assert lib.AmvVideoDecode(amv) == 0 and (frame(d) == want_v[k]).all()
                    other = amv1["video"][5]
                    ctypes.memmove(d.framebuf.videobuff, other, min(len(other), d.framebuf.videobufflen))
                    keep = d.framebuf.videobufflen
                    d.framebuf.videobufflen = min(len(other), keep)
                    rc = lib.AmvVideoDecode(amv)
                    ref, st, _ = orc.decode_frame(other[: d.framebuf.videobufflen], 128, 96)
                    assert (rc == 0) == (st == 0) and (frame(d) == ref.ravel()).all()
                    d.framebuf.videobufflen = keep
                held_v = (ctypes.cast(d.videobuf.fbmpdat, ctypes.c_void_p).value, ctypes.string_at(d.videobuf.fbmpdat, d.videobuf.len))
                held_a = (ctypes.cast(d.audiobuf.audiodata, ctypes.c_void_p).value, ctypes.string_at(d.audiobuf.audiodata, d.audiobuf.len))
            assert lib.AmvReadNextFrame(amv) == 0 and d.framebuf.framenum == -1          # AMV_END_
            assert lib.AmvRewindFrameStart(amv) == 0
            for k in range(10):
                assert lib.AmvReadNextFrame(amv) == 0 and lib.AmvVideoDecode(amv) == 0 and (frame(d) == want_v[k]).all()
            assert lib.AmvRewindFrameStart(amv) == 0                                       # mid-window
            assert lib.AmvReadNextFrame(amv) == 0 and lib.AmvVideoDecode(amv) == 0 and (frame(d) == want_v[0]).all()
            lib.AmvClose(amv)
        finally:
            os.environ.pop("AMVHIP_READAHEAD", None)
    # truncated file: frames before the cut decode, the cut frame fails without moving the position
    data = amv1["data"]
    cut = data.find(amv1["video"][20]) + 100
    p = str(tmp_path / "cut.amv")
    open(p, "wb").write(data[:cut])
    amv = lib.AmvOpen(p.encode())
    for k in range(20):
        assert lib.AmvReadNextFrame(amv) == 0 and lib.AmvVideoDecode(amv) == 0 and (frame(amv.contents) == want_v[k]).all()
    pos = amv.contents.fileseekpos
    assert lib.AmvReadNextFrame(amv) == -1 and amv.contents.fileseekpos == pos
    lib.AmvClose(amv)


def test_decode_fallback_rounds(ctx, pkg, orc):
    """frames that go through the serial kernel are reconstructed a round of dense lines at a time (the context keeps
    lines for 16 384 frames or a quarter of the batch, not for all of it): 20 000 small frames, more than half of them
    handed back by the unstuffer (a run of FF bytes longer than its look-back), and the whole batch again in
    AMVHIP_ENTROPY_SERIAL mode -- every frame and status against the oracle, in both output modes"""
    w, h, n = 32, 16, 20000
    rng = np.random.default_rng(99)
    base = [orc.encode_frame(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), w, h) for _ in range(40)]
    chunks = []
    for i in range(n):
        c = base[i % 40]
        if i % 9 < 5:   # a long run of FF FF ... in the scan: invalid, but every decoder must agree on what it does
            cut = 10 + (i % 17)
            c = c[:cut] + b"\xff" * 40 + c[cut:]   # (past the unstuffer's 32-byte look-back: these frames go to the serial kernel)
        chunks.append(c)
    uniq = {}
    for c in set(chunks):
        uniq[c] = (orc.decode_frame(c, w, h)[:2], orc.decode_frame_ffmpeg(c, w, h)[:2])
    for flags, pick in ((0, 0), (pkg.FLAG_FFMPEG, 1)):
        for mode in (pkg.ENTROPY_AUTO, pkg.ENTROPY_SERIAL):
            ctx.set_entropy_mode(mode)
            try:
                got, st = (_gpu_decode(ctx, chunks, w, h) if not flags else _gpu_decode_ffmpeg(ctx, pkg, chunks, w, h))
            finally:
                ctx.set_entropy_mode(pkg.ENTROPY_AUTO)
            for i in range(n):
                want, wst = uniq[chunks[i]][pick]
                assert st[i] == wst and (got[i].ravel() == want.ravel()).all(), (flags, mode, i)
    fresh = pkg.Context(0)   # buffers only grow: a context of its own shows what this batch needs
    _gpu_decode(fresh, chunks, w, h)
    assert 0 < fresh.decode_workspace_per_frame() < 16 * 1024
    # the frames the parallel kernels handed to the serial one are reported: the 5 of every 9 with the FF
    # run and no others -- the noise frames, every coefficient non-zero, have had record space of their own since round 4
    assert fresh.entropy_stats(False)["handed_to_serial"] == sum(1 for i in range(n) if i % 9 < 5)
    fresh.close()


def test_resample_matches_oracle(ctx, orc):
    """the rescaler in front of the encoder (imgresample.c img_resample, pinned by the reference's own object in
    tests/test_oracle_pin.py) on padded device planes, and the one-call rescale + encode == rescale, then encode"""
    import torch
    lib_call = ctx.lib.amvhip_resample_yuv420_dev
    rng = np.random.default_rng(17)
    for iw, ih, ow, oh, n in ((640, 480, 160, 120, 3), (320, 240, 160, 120, 4), (176, 144, 160, 120, 2), (160, 120, 320, 240, 2),
                              (130, 98, 64, 48, 3), (162, 122, 160, 120, 2)):
        icw, ich, ocw, och = iw // 2, ih // 2, ow // 2, oh // 2
        ys, cs = iw + 16, icw + 8
        Y = rng.integers(0, 256, (n, ih, ys), dtype=np.uint8)
        Cb = rng.integers(0, 256, (n, ich, cs), dtype=np.uint8)
        Cr = rng.integers(0, 256, (n, ich, cs), dtype=np.uint8)
        dY, dCb, dCr = _t(Y), _t(Cb), _t(Cr)
        oys, ocs = ow + 4, ocw + 4
        oY = torch.full((n, oh, oys), 0x5A, dtype=torch.uint8, device="cuda:0")
        oCb = torch.full((n, och, ocs), 0x5A, dtype=torch.uint8, device="cuda:0")
        oCr = torch.full((n, och, ocs), 0x5A, dtype=torch.uint8, device="cuda:0")
        rc = lib_call(ctx.h, dY.data_ptr(), dCb.data_ptr(), dCr.data_ptr(), ys, cs, ih * ys, ich * cs, iw, ih,
                      oY.data_ptr(), oCb.data_ptr(), oCr.data_ptr(), oys, ocs, oh * oys, och * ocs, ow, oh, n, None)
        assert rc == 0
        torch.cuda.synchronize()
        gY, gCb, gCr = oY.cpu().numpy(), oCb.cpu().numpy(), oCr.cpu().numpy()
        for t in range(n):
            tight = np.concatenate([Y[t, :, :iw].ravel(), Cb[t, :, :icw].ravel(), Cr[t, :, :icw].ravel()])
            want = orc.img_resample_yuv420(tight, iw, ih, ow, oh)
            assert (gY[t, :, :ow].ravel() == want[: ow * oh]).all(), (iw, ih, t)
            assert (gCb[t, :, :ocw].ravel() == want[ow * oh: ow * oh + ocw * och]).all()
            assert (gCr[t, :, :ocw].ravel() == want[ow * oh + ocw * och:]).all()
        assert (gY[:, :, ow:] == 0x5A).all() and (gCb[:, :, ocw:] == 0x5A).all()      # the padding is nobody's
        if (ow, oh) == (160, 120):   # one call: rescale + encode == encode of the rescaled planes
            cap = ctx.encode_bound(ow, oh) * n
            b1, o1, l1 = (torch.zeros(cap, dtype=torch.uint8, device="cuda:0"), torch.zeros(n, dtype=torch.int64, device="cuda:0"),
                          torch.zeros(n, dtype=torch.int32, device="cuda:0"))
            b2, o2, l2 = torch.zeros_like(b1), torch.zeros_like(o1), torch.zeros_like(l1)
            assert ctx.lib.amvhip_encode_yuv420_scaled_batch_dev(ctx.h, dY.data_ptr(), dCb.data_ptr(), dCr.data_ptr(), ys, cs, ih * ys,
                                                                 ich * cs, iw, ih, n, ow, oh, 0, b1.data_ptr(), cap, o1.data_ptr(),
                                                                 l1.data_ptr(), None) == 0
            ctx.encode_yuv420_batch_dev(oY, oCb, oCr, oys, ocs, oh * oys, och * ocs, n, ow, oh, 0, b2, cap, o2, l2)
            torch.cuda.synchronize()
            assert torch.equal(l1, l2) and torch.equal(o1, o2) and int(l1.min()) > 4
            end = int(o1[-1]) + int(l1[-1])
            assert torch.equal(b1[:end], b2[:end])


def test_adpcm_trellis_matches_oracle(ctx, pkg, orc):
    """the reference's `-trellis N` search (adpcm.c:287-443): every frontier size, chunks of different lengths (incl. the
    128-sample freeze boundary +-1, tiny and empty ones), loud / silent / clipping content, start indices 0..88; the
    end index of every chunk; the decoded error of the set is below the plain quantiser's from a beam of 8 on"""
    import torch
    lib = pkg.load_library()
    rng = np.random.default_rng(31)
    sizes = [1378, 1380, 128, 130, 126, 256, 258, 2, 0, 4, 640, 1378, 2048, 1376]
    offs_s = np.cumsum([0] + sizes).astype(np.uint64)
    pcm = orc.synth_audio(SEED, 4242, int(offs_s[-1]) + 2)
    pcm[1400:2700] = rng.integers(-32768, 32768, 1300)          # clipping noise
    pcm[3000:3600] = 0                                          # silence
    pcm[5000:7000] = (rng.integers(-3000, 3000, 2000)).astype(np.int16)
    n = len(sizes)
    step_in = np.array([0, 88, 40, 7, 60, 33, 1, 50, 12, 87, 20, 45, 70, 5], np.int32)
    coffs = np.cumsum([0] + [8 + s // 2 for s in sizes]).astype(np.uint64)
    d_pcm, d_po, d_ns, d_si, d_co = _t(pcm), _t(offs_s[:-1].copy()), _t(np.array(sizes, np.uint32)), _t(step_in), _t(coffs[:-1].copy())
    for trellis in (1, 2, 3, 4, 5):
        d_blob = torch.zeros(int(coffs[-1]) + 8, dtype=torch.uint8, device="cuda:0")
        d_so = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
        assert lib.amvhip_adpcm_encode_trellis_batch_dev(ctx.h, d_pcm.data_ptr(), d_po.data_ptr(), d_ns.data_ptr(), n, d_si.data_ptr(),
                                                         trellis, d_blob.data_ptr(), d_co.data_ptr(), d_so.data_ptr(), None) == 0
        torch.cuda.synchronize()
        blob, so = d_blob.cpu().numpy(), d_so.cpu().numpy()
        tot_t = tot_p = 0.0
        for i, sz in enumerate(sizes):
            seg = pcm[int(offs_s[i]):int(offs_s[i]) + sz]
            got = blob[int(coffs[i]):int(coffs[i + 1])].tobytes()
            if sz == 0:
                assert got[:8] == bytes([0, 0, int(step_in[i]), 0, 0, 0, 0, 0]) and so[i] == step_in[i]
                continue
            want, idx = orc.adpcm_encode_chunk_trellis(seg, int(step_in[i]), trellis)
            assert got == want and so[i] == idx, (trellis, i)
            plain, _ = orc.adpcm_encode_chunk(seg, int(step_in[i]))
            err = lambda c: float(((orc.adpcm_decode_chunk(c)[0][:sz].astype(np.float64) - seg) ** 2).sum())
            tot_t += err(got)
            tot_p += err(plain)
        assert trellis < 3 or tot_t < tot_p       # a beam of 8 or more beats the plain quantiser on the whole set
    # frame form: index in and out, a stream of chunks
    idx, want_idx = ctypes.c_int32(0), 0
    for i in range(8):
        seg = np.ascontiguousarray(pcm[i * 700: i * 700 + 700])
        out = np.zeros(8 + 350, np.uint8)
        m = lib.amvhip_adpcm_encode_frame_trellis(ctx.h, seg.ctypes.data, 700, ctypes.byref(idx), 3, out.ctypes.data, out.size)
        want, want_idx = orc.adpcm_encode_chunk_trellis(seg, want_idx, 3)
        assert m == len(want) and out.tobytes() == want and idx.value == want_idx
    assert lib.amvhip_adpcm_encode_frame_trellis(ctx.h, pcm.ctypes.data, 700, ctypes.byref(idx), 6, out.ctypes.data, out.size) == pkg.ERR_ARG


def test_one_lane_per_frame_walk(pkg, orc, amv1):
    """the entropy kernel a chip-filling batch gets (one lane per frame, amv_huffman_fast_kernel) on batches small enough
    to check every byte: a context created with AMVHIP_SYNC_LANES=1 decodes the reference's clip, geometries with one,
    two and "ten plus one" MCU-row segments per row, damaged / truncated / garbage chunks (the stride that raised the
    flag is walked again: status, first bad MCU), frames with more coefficients than the record space holds and frames
    the unstuffer hands to the serial kernel -- status and pixels equal the oracle's, in both output modes"""
    import os
    old = os.environ.get("AMVHIP_SYNC_LANES")
    os.environ["AMVHIP_SYNC_LANES"] = "1"
    try:
        one = pkg.Context(0)
    finally:
        if old is None:
            os.environ.pop("AMVHIP_SYNC_LANES", None)
        else:
            os.environ["AMVHIP_SYNC_LANES"] = old
    try:
        rng = np.random.default_rng(2024)
        got, st = _gpu_decode(one, amv1["video"], amv1["info"]["width"], amv1["info"]["height"])
        want, wst = _oracle_decode(orc, amv1["video"], amv1["info"]["width"], amv1["info"]["height"])
        assert (st == wst).all() and (got == want).all()
        for w, h, n in ((160, 120, 130), (320, 240, 7), (176, 144, 9), (16, 16, 70), (336, 32, 5), (130, 98, 66)):
            chunks = _synth_chunks(orc, n, w, h)
            noise = orc.encode_frame(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), w, h)   # fills the record space
            flat = orc.encode_frame(np.full((h, w, 3), 128, np.uint8), w, h)                    # DC + EOB only
            chunks += [noise, flat, noise[: len(noise) // 2], flat[:3], b"", b"\xff\xd8" + b"\xff" * 300]
            for k in range(10):   # damage at random places, early and late
                b = bytearray(chunks[k % n])
                for _ in range(1 + k % 3):
                    b[int(rng.integers(2, len(b) - 2))] ^= 1 << int(rng.integers(0, 8))
                chunks.append(bytes(b))
            c = chunks[1]
            chunks.append(c[:12] + b"\xff" * 40 + c[12:])   # a run of FF bytes past the unstuffer's look-back: serial kernel
            chunks.append(c[:12] + b"\xff" * 5 + c[12:])    # a short run: the unstuffer's general rule
            for flags in (0, 1):
                got, st = _gpu_decode(one, chunks, w, h, flags, pad_front=flags)
                want, wst = _oracle_decode(orc, chunks, w, h, flags)
                assert (st == wst).all(), (w, h, flags, st, wst)
                assert (got == want).all(), (w, h, flags)
            assert (wst != 0).sum() >= 4
            got, st = _gpu_decode_ffmpeg(one, pkg, chunks, w, h)
            want, wst = _oracle_decode_ffmpeg(orc, chunks, w, h)
            assert (st == wst).all() and (got == want).all(), (w, h)
    finally:
        one.close()


def test_decode_submit_collect_pipeline(pkg, orc, amv1):
    """amvhip_decode_submit_dev / _collect_dev: batches of different sizes, geometries and flags (damaged chunks among
    them), the next one submitted before the last one is collected, every batch with buffers of its own -- status and
    pixels equal the oracle's; a third submit without a collect is refused; another entry point of the context in
    between waits for what is in flight; an empty batch goes through"""
    import torch
    ctx = pkg.Context(0)
    try:
        rng = np.random.default_rng(31)
        stream = torch.cuda.Stream()
        sid = stream.cuda_stream
        jobs = []
        for k, (w, h, n, flags) in enumerate(((160, 120, 90, 0), (320, 240, 6, 1), (16, 16, 300, 0), (176, 144, 5, 0), (160, 120, 33, 0),
                                              (130, 98, 4, 1))):
            chunks = _synth_chunks(orc, n, w, h, first=10 * k)
            b = bytearray(chunks[n // 2]); b[len(b) // 2] ^= 0x40; chunks[n // 2] = bytes(b)       # damage
            chunks[-1] = chunks[-1][: len(chunks[-1]) // 2]                                        # truncation
            blob, offs, lens, nbytes = _blob_of(chunks, pad_front=k % 3)
            jobs.append({"w": w, "h": h, "n": n, "flags": flags, "chunks": chunks, "nbytes": nbytes, "blob": _t(blob), "offs": _t(offs),
                         "lens": _t(lens), "out": torch.full((n, h, ctx.stride(w)), 0x5A, dtype=torch.uint8, device="cuda:0"),
                         "st": torch.full((n,), -1, dtype=torch.int32, device="cuda:0")})
        torch.cuda.synchronize()

        def submit(j):
            ctx.decode_submit_dev(j["blob"], j["nbytes"], j["offs"], j["lens"], j["n"], j["w"], j["h"], j["flags"], j["out"], j["st"], sid)

        submit(jobs[0])
        for k in range(1, len(jobs)):
            submit(jobs[k])               # the entropy stage of batch k beside the reconstruction of batch k - 1
            if k == 2:
                with pytest.raises(pkg.AmvHipError):
                    submit(jobs[0])       # two in flight
            ctx.decode_collect_dev(sid)   # batch k - 1
        ctx.decode_collect_dev(sid)
        with pytest.raises(pkg.AmvHipError):
            ctx.decode_collect_dev(sid)   # nothing left
        stream.synchronize()
        for j in jobs:
            want, wst = _oracle_decode(orc, j["chunks"], j["w"], j["h"], j["flags"])
            assert (j["st"].cpu().numpy() == wst).all(), (j["w"], j["h"])
            assert (j["out"].cpu().numpy() == want).all(), (j["w"], j["h"])
        # a batch in flight, then the one-call form on the same context (which waits for it), then an empty batch
        for j in jobs[:2]:
            j["out"].fill_(0x5A); j["st"].fill_(-1)
        submit(jobs[0])
        j = jobs[1]
        ctx.decode_batch_dev(j["blob"], j["nbytes"], j["offs"], j["lens"], j["n"], j["w"], j["h"], j["flags"], j["out"], j["st"], sid)
        ctx.decode_submit_dev(j["blob"], 0, j["offs"], j["lens"], 0, j["w"], j["h"], 0, j["out"], j["st"], sid)
        ctx.decode_collect_dev(sid)
        ctx.decode_collect_dev(sid)
        stream.synchronize()
        for j in jobs[:2]:
            want, wst = _oracle_decode(orc, j["chunks"], j["w"], j["h"], j["flags"])
            assert (j["st"].cpu().numpy() == wst).all() and (j["out"].cpu().numpy() == want).all()
    finally:
        ctx.close()


def test_unstuffer_boundaries(ctx, orc):
    """FF bytes where the unstuffer's tiles (1 KB), lanes (16 bytes) and words meet: a valid chunk with FF 00, FF FF 00,
    FF FF FF 00 and a lone FF spliced in around those places, for every alignment of the chunk in the blob; chunks that
    end in FF, at a tile boundary, one byte short of it -- status and pixels equal the oracle's"""
    w, h = 160, 120
    rng = np.random.default_rng(404)
    base = orc.encode_frame(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), w, h)   # noise: a long chunk (> 4 tiles)
    assert len(base) > 4200
    chunks = []
    for splice in (b"\xff\x00", b"\xff\xff\x00", b"\xff\xff\xff\x00", b"\xff"):
        for at in (2, 3, 13, 14, 15, 16, 17, 18, 1021, 1022, 1023, 1024, 1025, 1026, 1027, 2047, 2048, 2049, 4094, 4095, 4096):
            chunks.append(base[:at] + splice + base[at:])
    for cut in (1024, 1025, 1026, 1027, 1028, 2050, 4097, 4098):
        chunks.append(base[:cut])
        chunks.append(base[:cut - 1] + b"\xff")
        chunks.append(base[:cut - 2] + b"\xff\xff")
    for pad in range(4):
        got, st = _gpu_decode(ctx, chunks, w, h, 0, pad_front=pad)
        want, wst = _oracle_decode(orc, chunks, w, h)
        assert (st == wst).all(), (pad, np.nonzero(st != wst)[0][:8])
        assert (got == want).all(), pad


def test_entropy_trace_has_a_line_per_wave(pkg, orc):
    """amvhip_entropy_trace (round 6: what a small batch's entropy launch lasts as long as): with gathering
    on, a decode through the several-lanes kernel leaves one line per task -- begin before end on the constant-rate clock,
    phase clocks that add up to less than the task's duration at any plausible shader clock, rounds within the loop's bound,
    the lane count and share length of the launch -- and the pixels are the oracle's whether gathering is on or not"""
    import os
    import torch
    w, h, n = 160, 120, 400
    chunks = _synth_chunks(orc, n, w, h)
    old = os.environ.get("AMVHIP_SYNC_LANES")
    os.environ["AMVHIP_SYNC_LANES"] = "16"
    try:
        c = pkg.Context(0)
    finally:
        if old is None:
            os.environ.pop("AMVHIP_SYNC_LANES", None)
        else:
            os.environ["AMVHIP_SYNC_LANES"] = old
    try:
        plain, st0 = _gpu_decode(c, chunks, w, h)
        c.entropy_stats(True)
        traced, st1 = _gpu_decode(c, chunks, w, h)
        tr = c.entropy_trace(4096)
        stats = c.entropy_stats(False)
        assert (st0 == 0).all() and (st1 == 0).all() and (plain == traced).all()
        want = np.stack([orc.decode_frame(ch, w, h)[0] for ch in chunks[:8]])
        assert (traced[:8] == want).all()
        tr = tr[tr[:, 1] != 0]
        assert len(tr) == (n + 3) // 4 == stats["waves"]                       # four frames per wave at sixteen lanes
        dur = (tr[:, 1] - tr[:, 0]).astype(np.int64)                           # 100 MHz ticks
        assert (dur > 0).all() and (dur < 100 * 1000 * 50).all()               # under 50 ms each
        phases = tr[:, 2:6].astype(np.int64).sum(axis=1)                       # shader clocks
        assert (phases > 0).all() and (phases < dur * 40).all()                # a shader clock under 4 GHz
        rounds = (tr[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
        share = (tr[:, 6] >> np.uint64(32)).astype(np.int64)
        assert (rounds >= 1).all() and (rounds <= 17).all() and rounds.max() == stats["max_rounds"]
        bits = np.array([8 * (len(ch) - 4) for ch in chunks])
        assert share.min() >= 64 and share.max() <= ((bits.max() + 15) // 16 + 31) // 32 * 32
        assert ((tr[:, 7] & np.uint64(0xffff)) == 16).all()
    finally:
        c.close()


def test_every_lane_count_gives_the_same_bytes(pkg, orc):
    """4 000 chunks, a quarter of them damaged (bit flips, truncation, runs of FF spliced in), decoded with 1, 2, 4, 8, 16, 32
    and 64 lanes per frame (the one-lane kernel and every instantiation of the several-lanes one, whose lanes remember their
    walks and find finality by a prefix scan since round 5: 64 lanes on these frames means shares of 440 bits, a dozen
    rounds and full memos): identical statuses and pixels, and a sample equals the oracle's"""
    import os
    import torch
    w, h, n = 160, 120, 4000
    rng = np.random.default_rng(7)
    base = _synth_chunks(orc, 100, w, h)
    chunks = []
    for i in range(n):
        c = bytearray(base[i % 100])
        r = rng.random()
        if r < 0.15:
            for _ in range(int(rng.integers(1, 4))):
                c[int(rng.integers(2, len(c) - 2))] ^= 1 << int(rng.integers(0, 8))
        elif r < 0.20:
            c = c[: int(rng.integers(2, len(c)))]
        elif r < 0.25:
            p = int(rng.integers(2, len(c) - 2))
            c[p:p] = b"\xff" * int(rng.integers(1, 30))
        chunks.append(bytes(c))
    old = os.environ.get("AMVHIP_SYNC_LANES")
    res = {}
    try:
        for lanes in ("1", "2", "4", "8", "16", "32", "64"):
            os.environ["AMVHIP_SYNC_LANES"] = lanes
            one = pkg.Context(0)
            try:
                res[lanes] = _gpu_decode(one, chunks, w, h, 0, pad_front=int(lanes) % 4)
            finally:
                one.close()
    finally:
        if old is None:
            os.environ.pop("AMVHIP_SYNC_LANES", None)
        else:
            os.environ["AMVHIP_SYNC_LANES"] = old
    ref_out, ref_st = res["1"]
    assert (ref_st != 0).sum() > n // 10
    for lanes, (out, st) in res.items():
        assert (st == ref_st).all() and (out == ref_out).all(), lanes
    for i in rng.integers(0, n, 120):
        want, wst, _ = orc.decode_frame(chunks[int(i)], w, h)
        assert wst == ref_st[int(i)] and (ref_out[int(i)] == want).all(), int(i)


def test_kernels_match_reference_produced_outputs(ctx, pkg, orc, amv1):
    """tests/golden/reference_outputs.json (hashes of what real amvlib, the reference's patched FFmpeg and its adpcm_ima_amv
    encoder produced): the HIP path reproduces every one of them -- amvlib decode at the headline 160x120 (half-used last
    MCU row), 320x240 and four odd geometries; the FFmpeg-compat decode of all of AMV1.amv and of a synthetic clip; the
    ADPCM encoder with the step index carried, plain and -trellis 3"""
    import json
    import os
    from conftest import GOLDEN
    ref = json.load(open(os.path.join(GOLDEN, "reference_outputs.json")))
    seed, basis = ref["seed"], int(ref["fnv_basis"], 16)
    step = ref["amvlib_decode"]["frame_step"]
    for c in ref["amvlib_decode"]["cases"]:
        w, h = c["w"], c["h"]
        chunks = [orc.encode_frame(orc.synth_frame(seed, step * t, w, h), w, h) for t in range(c["n"])]
        out, st = _gpu_decode(ctx, chunks, w, h)
        assert (st == 0).all()
        hh = basis
        for f in out:
            hh = orc.fnv1a64(hh, f)
        assert "%016x" % hh == c["fnv"], (w, h)
    ff = ref["ffmpeg_decode"]
    s = ff["synth_160x120"]
    synth = [orc.encode_frame(orc.synth_frame(seed, s["frame_step"] * t, s["w"], s["h"]), s["w"], s["h"]) for t in range(s["n"])]
    for chunks, w, h, want in ((amv1["video"], 128, 96, ff["amv1_all_252_frames"]), (synth, s["w"], s["h"], s["fnv"])):
        out, st = _gpu_decode_ffmpeg(ctx, pkg, chunks, w, h)
        assert (st == 0).all()
        hh = basis
        for f in out:
            hh = orc.fnv1a64(hh, f)
        assert "%016x" % hh == want, (w, h)
    # the audio encoder as the plugin drives it: one chunk per call, the step index handed from call to call
    lib = pkg.load_library()
    a = ref["adpcm_ima_amv_encode"]
    fs, k = a["frame_size"], a["chunks"]
    pcm = orc.synth_audio(seed, 0, k * fs)
    for key in ("plain", "trellis3"):
        idx, hh = ctypes.c_int32(0), basis
        for i in range(k):
            seg = np.ascontiguousarray(pcm[i * fs: (i + 1) * fs])
            out = np.zeros(8 + fs // 2, np.uint8)
            if key == "plain":
                m = lib.amvhip_adpcm_encode_frame(ctx.h, seg.ctypes.data, fs, ctypes.byref(idx), out.ctypes.data, out.size)
            else:
                m = lib.amvhip_adpcm_encode_frame_trellis(ctx.h, seg.ctypes.data, fs, ctypes.byref(idx), a[key]["trellis"],
                                                          out.ctypes.data, out.size)
            assert m == out.size
            hh = orc.fnv1a64(hh, out)
        assert ("%016x" % hh, idx.value) == (a[key]["fnv"], a[key]["end_index"]), key
    # and the batch form with the index carried on the device (no start indices given)
    offs_s = (np.arange(k) * fs).astype(np.uint64)
    coffs = (np.arange(k) * (8 + fs // 2)).astype(np.uint64)
    blob = np.zeros(k * (8 + fs // 2), np.uint8)
    ctx.adpcm_encode_batch(pcm, pcm.size, offs_s, np.full(k, fs, np.uint32), k, None, blob, blob.size, coffs)
    hh = basis
    for i in range(k):
        hh = orc.fnv1a64(hh, blob[i * (8 + fs // 2): (i + 1) * (8 + fs // 2)])
    assert "%016x" % hh == a["plain"]["fnv"]


def _with_env(pkg, name, value):
    """a context created while the environment variable holds `value` (the knobs are read at creation)"""
    import os
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return pkg.Context(0)
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def test_adpcm_index_chain_routes(pkg, orc):
    """The step-index chain (adpcm.c:461-498) resolved by guessed starts + sweeps, by the settling workgroup alone, by the
    exhaustive 89-start map, and -- on a stream built so that every chunk's end depends on its start (hundreds of
    two-to-eight-sample chunks: each sweep settles one more chunk) -- by the exhaustive route the device falls back to on
    its own.  Every route writes the bytes of the oracle's sequential encode."""
    rng = np.random.default_rng(99)
    ordinary = [1378] * 900 + [2 * int(rng.integers(0, 700)) for _ in range(100)]
    rng.shuffle(ordinary)
    cases = []
    for kind, sizes in (("audio", ordinary), ("walk", [2] * 1500), ("audio", [1378] * 40), ("walk", [2] * 70), ("audio", [1378])):
        n = len(sizes)
        pcm_offs = np.cumsum([0] + sizes).astype(np.uint64)
        pcm = orc.synth_audio(SEED, 777, int(pcm_offs[-1]) + 2)
        if kind == "walk":
            # two-sample chunks {a, a} (index - 2) and {a, a + 20000} (index - 1 + 8): a walk kept inside 20..60, where
            # nothing clamps and every chunk maps start s to s + constant -- the end depends on the start all along
            idx = 0
            for i in range(n):
                up = idx < 20 or (idx < 60 and rng.integers(0, 4) == 0)
                pcm[2 * i] = -10000
                pcm[2 * i + 1] = 10000 if up else -10000
                idx += 7 if up else -2
        offs = np.cumsum([0] + [8 + s // 2 for s in sizes]).astype(np.uint64)
        want, idx, starts = [], 0, []
        for i in range(n):
            starts.append(idx)
            seg = pcm[int(pcm_offs[i]):int(pcm_offs[i + 1])]
            if seg.size:
                chunk, idx = orc.adpcm_encode_chunk(seg, idx)
            else:
                chunk = bytes([0, 0, idx, 0, 0, 0, 0, 0])
            want.append(chunk)
        if kind == "walk":
            assert 15 <= min(starts[8:]) and max(starts) <= 70 and len(set(starts)) > 20, (min(starts[8:]), max(starts))
        cases.append((sizes, pcm, pcm_offs, offs, b"".join(want), len(set(starts))))
    seen_exhaustive = seen_settled = seen_unsettled = 0
    for knob in (None, "0", "3", "map", "nosettle"):
        ctx = pkg.Context(0) if knob is None else _with_env(pkg, "AMVHIP_ADPCM_SWEEPS", knob)
        try:
            for k, (sizes, pcm, pcm_offs, offs, want, _) in enumerate(cases):
                n = len(sizes)
                blob = np.full(int(offs[-1]), 0xEE, np.uint8)
                ctx.adpcm_encode_batch(pcm, pcm.size, pcm_offs[:-1].copy(), np.array(sizes, np.uint32), n, None, blob, blob.size,
                                       offs[:-1].copy())
                assert blob.tobytes() == want, (knob, k)
                if knob == "nosettle":
                    # the chain stopped after its launched sweeps: its own check (every chunk's start == its predecessor's
                    # end) must have noticed wherever something was left, and the exhaustive route have written the bytes
                    st = ctx.adpcm_chain_stats()
                    if k in (0, 1):
                        assert st["exhaustive"], (knob, k, st)
                        seen_unsettled += 1
                elif knob != "map":
                    st = ctx.adpcm_chain_stats()
                    if k == 1:
                        assert st["exhaustive"], (knob, st)                  # 1 500 dependent chunks: no sweep count settles them
                        seen_exhaustive += 1
                    if k == 0 and knob is None:
                        assert not st["exhaustive"] and len(st["recoded"]) >= 2 and st["recoded"][0] < n, st
                        assert all(a >= b for a, b in zip(st["recoded"], st["recoded"][1:])), st
                        seen_settled += 1
                    if k in (2, 4):
                        assert not st["exhaustive"], (knob, k, st)
        finally:
            ctx.close()
    assert seen_exhaustive == 3 and seen_settled == 1 and seen_unsettled == 2


def test_adpcm_index_chain_many_streams(ctx, orc):
    """The sweeps' hand-offs are between workgroups that run at the same time (the front sweep writes predictions about
    chunks other workgroups are coding): thirty ragged streams of a few hundred to a few thousand chunks, each coded
    three times, every byte against the oracle's sequential encode.  (A first list that missed one chunk in 4 000 -- a
    shuffle inside a branch -- passed every other test of this file.)  (Stream lengths chosen so that the front sweep gets
    its look-ahead lists of one to three hundred heads.)"""
    rng = np.random.default_rng(20260404)
    for it in range(30):
        n = int(rng.integers(300, 5000))
        sizes = [1378 if rng.integers(0, 10) else 2 * int(rng.integers(0, 700)) for _ in range(n)]
        pcm_offs = np.cumsum([0] + sizes).astype(np.uint64)
        pcm = orc.synth_audio(SEED + it, 4242, int(pcm_offs[-1]) + 2)
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
            assert blob.tobytes() == want, (it, rep, n, ctx.adpcm_chain_stats())
        # the property that makes a stream the sequential encoder's, read off the BYTES alone (adpcm.c:461-498 carries
        # step_index from chunk to chunk): the index in a chunk's header is where the DECODER's index stands at the end of
        # the chunk before -- checked with the device's own decoder, which knows nothing of the encoder's bookkeeping
        lens = np.array([8 + s // 2 for s in sizes], np.uint32)
        out_offs = np.cumsum([0] + [2 * (int(l) - 8) for l in lens]).astype(np.uint64)
        dec = np.zeros(int(out_offs[-1]) + 8, np.int16)
        fin = np.full((n, 2), -7, np.int32)
        ctx.adpcm_decode_batch(blob, blob.size, offs[:-1].copy(), lens, n, dec, dec.size, out_offs[:-1].copy(), fin)
        head = blob[offs[:-1].astype(np.int64) + 2].astype(np.int32)
        assert head[0] == 0
        carried = 0
        for i in range(n):
            assert head[i] == carried, (it, i)
            if sizes[i]:
                carried = int(fin[i, 1])


def test_adpcm_full_size_chain_against_exhaustive_route(pkg, orc):
    """bench.py's ADPCM workload at full size (200 000 chunks of 1 378 samples, index carried through all of them): the chain
    of guessed starts, sweeps, front sweep and settling rounds writes the bytes the exhaustive 89-start route writes (other
    kernels, no guessing), every start index in the headers continues its predecessor's decode-side end index, the first 300
    chunks are the oracle's sequential encode, and decoding returns PCM of the right length for every chunk."""
    import torch
    dev = "cuda:0"
    n, spf = 200000, 1378
    clen = 8 + spf // 2
    s = torch.cuda.current_stream().cuda_stream
    ctx = pkg.Context(0)
    ref = _with_env(pkg, "AMVHIP_ADPCM_SWEEPS", "map")
    try:
        pcm = torch.empty(n * spf, dtype=torch.int16, device=dev)
        ctx.synth_audio_dev(SEED, 0, n * spf, pcm, s)
        pcm_offs = torch.arange(n, dtype=torch.int64, device=dev) * spf
        nsamp = torch.full((n,), spf, dtype=torch.int32, device=dev)
        offs = torch.arange(n, dtype=torch.int64, device=dev) * clen
        got = torch.full((n * clen + 16,), 0xEE, dtype=torch.uint8, device=dev)
        want = torch.full((n * clen + 16,), 0xDD, dtype=torch.uint8, device=dev)
        for rep in range(3):
            got[: n * clen] = 0xEE
            ctx.adpcm_encode_batch_dev(pcm, pcm_offs, nsamp, n, None, got, offs, s)
            if rep == 0:
                ref.adpcm_encode_batch_dev(pcm, pcm_offs, nsamp, n, None, want, offs, s)
            torch.cuda.synchronize()
            assert torch.equal(got[: n * clen], want[: n * clen]), rep
        st = ctx.adpcm_chain_stats()
        assert not st["exhaustive"] and len(st["recoded"]) >= 4, st
        # the chain as the headers state it: chunk i + 1 starts where decoding chunk i leaves the step index
        lens = torch.full((n,), clen, dtype=torch.int32, device=dev)
        back = torch.zeros(n * spf + 8, dtype=torch.int16, device=dev)
        fin = torch.zeros((n, 2), dtype=torch.int32, device=dev)
        ctx.adpcm_decode_batch_dev(got, n * clen, offs, lens, n, back, pcm_offs, fin, s)
        torch.cuda.synchronize()
        heads = got[: n * clen].view(n, clen)
        assert int(heads[0, 2]) == 0 and torch.equal(heads[1:, 2].to(torch.int32), fin[:-1, 1])
        assert len(torch.unique(heads[:, 2])) > 20
        h_pcm, h_got = pcm[: 300 * spf].cpu().numpy(), got[: 300 * clen].cpu().numpy()
        idx = 0
        for i in range(300):
            chunk, idx = orc.adpcm_encode_chunk(h_pcm[i * spf:(i + 1) * spf], idx)
            assert h_got[i * clen:(i + 1) * clen].tobytes() == chunk, i
    finally:
        ctx.close()
        ref.close()


def test_encode_frame_kernel_paths(ctx, pkg, orc):
    """amv_encode_frame_kernel's less travelled paths, every chunk against the oracle's encoder: runs of symbols that
    overflow a lane's scratch and are coded a second time straight into the round's bit string (a few noisy blocks in a
    flat picture); frames handed back to the two-stage route from a batch larger than one fall-back round (so that rounds
    with base > 0 run), among frames that are not; three segments per MCU row (336 wide) and an uneven split (176 wide: 6 + 5 MCUs), with the MJPEG quantiser bias."""
    import torch
    rng = np.random.default_rng(41)
    # (a) flat frames with islands of noise, 320x240 (two segments per row)
    w, h = 320, 240
    frames = []
    for t in range(6):
        f = orc.synth_frame(SEED, 11 * t, w, h)
        f[:, :] = f[h // 2, w // 2]                              # flat
        for _ in range(1 + t):
            y, x = int(rng.integers(0, h - 8)) & ~7, int(rng.integers(0, w - 8)) & ~7
            f[y:y + 8, x:x + 8] = rng.integers(0, 256, (8, 8, 3))
        frames.append(f)
    frames.append(rng.integers(0, 256, (h, w, 3)).astype(np.uint8))   # and one the window cannot hold
    src = np.stack(frames)
    n = len(frames)
    cap = ctx.encode_bound(w, h) * n
    blob, offs, lens = np.zeros(cap, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    ctx.encode_batch(src, w * 3, 0, n, w, h, 0, blob, cap, offs, lens)
    fat = 0
    for i in range(n):
        want, coef = orc.encode_frame(src[i], w, h, want_coef=True)
        assert blob[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes() == want, i
        fat += int(((coef[:, 1:] != 0).sum(axis=1) > 40).sum())
    assert fat >= 10                                             # blocks of > 40 coefficients: > 256 bits each
    # (b) 1 100 small frames, every 97th one noise: two fall-back rounds, hand-backs in both
    w, h, n = 48, 32, 1100
    src = np.stack([orc.synth_frame(SEED, t, w, h) if t % 97 else rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for t in range(n)])
    cap = ctx.encode_bound(w, h) * n
    blob, offs, lens = np.zeros(cap, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    ctx.encode_batch(src, w * 3, 0, n, w, h, 0, blob, cap, offs, lens)
    for i in list(range(0, n, 97)) + [1, 2, 500, 1023, 1024, 1025, 1067, 1099]:
        assert blob[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes() == orc.encode_frame(src[i], w, h), i
    assert int(offs[-1]) + int(lens[-1]) == int(lens.astype(np.int64).sum())
    # (c) 336 x 32: 21 MCU columns = segments of 7 + 7 + 7; 176 x 144: 11 = 6 + 5
    for w, h in ((336, 32), (176, 144)):
        src = np.stack([orc.synth_frame(SEED, 5 * t, w, h) for t in range(3)])
        n = 3
        cap = ctx.encode_bound(w, h) * n
        blob, offs, lens = np.zeros(cap, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
        ctx.encode_batch(src, w * 3, 0, n, w, h, 128, blob, cap, offs, lens)
        for i in range(n):
            assert blob[int(offs[i]):int(offs[i]) + int(lens[i])].tobytes() == orc.encode_frame(src[i], w, h, qbias=128), (w, h, i)


def test_amvlib_reader_has_no_size_limit(ctx, pkg, orc, tmp_path):
    """the read-ahead window's buffers are sized for ordinary AMV streams; a frame that does not fit an empty window --
    a noise picture (its chunk is several times 0.2 byte per pixel) and an audio chunk of more than 8 KB, as the FIRST
    frame of a window and as a later one, with every window size including AMVHIP_READAHEAD=1 -- is read and decoded
    like any other (the reference's reader mallocs whatever the chunk header says, AMVDec.c:196-231)"""
    import os
    lib = pkg.load_library()
    w, h = 160, 120
    rng = np.random.default_rng(17)
    noise = orc.encode_frame(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), w, h)
    noise = noise[:-2] + bytes(20000) + noise[-2:]                                 # what follows the last MCU is not read (AmvJpeg.c:1244-1287)
    assert len(noise) > w * h // 2 + 4096 and orc.decode_frame(noise, w, h)[1] == 0   # fatter than a window slot, and valid
    plain = [orc.encode_frame(orc.synth_frame(SEED, t, w, h), w, h) for t in range(6)]
    long_pcm = orc.synth_audio(SEED, 0, 20000)
    big_audio, _ = orc.adpcm_encode_chunk(long_pcm, 0)                             # 10 008 bytes
    small_audio, _ = orc.adpcm_encode_chunk(long_pcm[:1378], 0)
    video = [noise, plain[0], plain[1], noise, plain[2], plain[3]]
    audio = [big_audio, small_audio, big_audio, small_audio, small_audio, big_audio]
    path = str(tmp_path / "fat.amv").encode()
    m = lib.amvhip_mux_open(path, w, h, 16, 22050, 200000, 64000)
    for v, a in zip(video, audio):
        vb, abuf = np.frombuffer(v, np.uint8), np.frombuffer(a, np.uint8)
        assert lib.amvhip_mux_write_frame(m, vb.ctypes.data, vb.size, abuf.ctypes.data, abuf.size) == 0
    assert lib.amvhip_mux_close(m) == 0
    for window in ("1", "2", "256"):
        os.environ["AMVHIP_READAHEAD"] = window
        try:
            amv = lib.AmvOpen(path)
            d = amv.contents
            for k, (v, a) in enumerate(zip(video, audio)):
                assert lib.AmvReadNextFrame(amv) == 0, (window, k)
                assert ctypes.string_at(d.framebuf.videobuff, d.framebuf.videobufflen) == v
                assert ctypes.string_at(d.framebuf.audiobuff, d.framebuf.audiobufflen) == a
                assert lib.AmvVideoDecode(amv) == 0
                got = np.frombuffer(ctypes.string_at(d.videobuf.fbmpdat, d.videobuf.len), np.uint8)
                assert (got == orc.decode_frame(v, w, h)[0].ravel()).all(), (window, k)
                assert lib.AmvAudioDecode(amv) == 0
                n4 = (len(a) - 8 + 3) & ~3
                pcm = np.frombuffer(ctypes.string_at(d.audiobuf.audiodata, d.audiobuf.len), np.int16)
                assert d.audiobuf.len == 4 * n4 and (pcm == orc.adpcm_decode_chunk(a + b"\0" * (n4 - (len(a) - 8)))[0]).all(), (window, k)
            assert lib.AmvReadNextFrame(amv) == 0 and d.framebuf.framenum == -1
            lib.AmvClose(amv)
        finally:
            os.environ.pop("AMVHIP_READAHEAD", None)


def test_c_host_shards_a_stream_over_contexts(pkg, amv1, tmp_path):
    """tests/c/shard_host.c: one process, one amvhip context + HIP stream per device (and, on a box with fewer devices
    than contexts, several independent contexts per device), contiguous frame ranges, decoded frames gathered device to
    device into one buffer on device 0 -- equal to a single-context decode and to the hash amvlib itself produced.
    More than one DEVICE is exercised only where the box has them (this pool's test boxes have one)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "shard_host")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.run(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "c", "shard_host.c"), "-L", libdir, "-l:" + os.path.basename(pkg.LIB_PATH),
                    "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    for contexts in (None, 2, 5):
        cmd = [exe, amv1["path"]] + ([str(contexts)] if contexts else [])
        out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=300).stdout
        f = dict(line.split(": ", 1) for line in out.strip().splitlines())
        assert f["frames"] == "252" and f["failed frames"] == "0" and f["match"] == "yes", out
        assert int(f["gathered fnv1a64"], 16) == AMVLIB_HASH and int(f["single fnv1a64"], 16) == AMVLIB_HASH
        want = contexts if contexts else int(f["devices"])
        assert int(f["contexts"]) == want
        ranges = [f["context %d" % r] for r in range(want)]
        assert ("frames 0..%d gather" % (252 // want)) in ranges[0] and "..252 gather" in ranges[-1]
        # every context says how its frames reached device 0: in place, over the peer link, or staged by the runtime
        assert ranges[0].endswith("gather: same device")
        assert all(r.endswith(("same device", "peer link (access enabled)", "staged by the runtime (no peer access)")) for r in ranges)


def test_amvlib_adpcm_stereo_decode(ctx, pkg, orc):
    """AdpcmImaDecodeFrame with channel == 2 (AdpcmIma.c:222-237): of every 8 input bytes the first 4 are the left
    channel's nibbles, the last 4 the right's, each channel with a state of its own, samples interleaved L R; the
    8-byte stride reads up to 7 bytes past buf_size (taken as zero).  Against what the REFERENCE's own AdpcmIma.c object made
    of the same seeded inputs (tests/golden/ref_adpcm_stereo.json, written here from oracle/_ref by
    tests/golden/make_ref_golden.py: the compiled reference is not loaded on the GPU box), and against the oracle's mono
    decoder channel by channel; the context's state carries from call to call."""
    import json
    import os
    from conftest import ROOT
    sys_path_golden = os.path.join(ROOT, "tests", "golden")
    fixture = json.load(open(os.path.join(sys_path_golden, "ref_adpcm_stereo.json")))
    lib = pkg.load_library()
    rng = np.random.default_rng(fixture["seed"])
    for case in fixture["cases"]:
        size = case["size"]
        buf = rng.integers(0, 256, size, dtype=np.uint8)
        n8 = (size + 7) & ~7
        mine = pkg.ADPCMContext()
        mine.channel = 2
        for ch in (0, 1):
            mine.status[ch].predictor, mine.status[ch].step_index = fixture["start"][ch]
        for call in range(2):                                  # the second call starts from the state the first left
            got = np.zeros(2 * n8 + 16, np.int16)
            dl = ctypes.c_int(0)
            start = [(mine.status[ch].predictor, mine.status[ch].step_index) for ch in (0, 1)]
            rc = lib.AdpcmImaDecodeFrame(ctypes.byref(mine), got.ctypes.data, ctypes.byref(dl), buf.ctypes.data, size)
            assert rc == n8 and dl.value == 4 * n8
            padded = np.concatenate([buf, np.zeros(n8 - size, np.uint8)]).reshape(-1, 8)
            want = np.zeros(2 * n8, np.int16)
            ends = []
            for ch in (0, 1):                                  # channel ch = a mono AMV chunk of its own
                pred, idx = start[ch]
                chunk = int(pred).to_bytes(2, "little", signed=True) + bytes([idx, 0, 0, 0, 0, 0]) + padded[:, 4 * ch: 4 * ch + 4].tobytes()
                pcm, _ = orc.adpcm_decode_chunk(chunk)
                want[ch::2] = pcm[:n8]
                ends.append(int(pcm[n8 - 1]))
            assert (got[: 2 * n8] == want).all(), (size, call)
            assert [mine.status[0].predictor, mine.status[1].predictor] == ends
            theirs = case["calls"][call]                       # the reference's own run of this call
            assert theirs["rc"] == rc and theirs["declen"] == dl.value, (size, call)
            assert "%016x" % orc.fnv1a64(orc.FNV_BASIS, got[: 2 * n8].view(np.uint8)) == theirs["pcm_fnv"], (size, call)
            assert [[mine.status[ch].predictor, mine.status[ch].step_index] for ch in (0, 1)] == theirs["end"], (size, call)


def test_encode_yuv422_entry(ctx, orc):
    """amvhip_encode_yuv422_batch(_dev): planar YUVJ422P in (chroma planes w/2 x h, padded strides) against the ORACLE's
    restatement of the rule -- amvo_yuv422_to_420 (chroma row r = (row 2r + row 2r+1 + 1) >> 1) followed by the oracle's
    plane encoder, which test_oracle_pin.py ties to the oracle's RGB encoder -- byte for byte: the fixture geometries with
    partial MCUs (rows below the picture repeat the edge row), device and host forms, the frames the one-kernel encoder
    hands back (noise), 255 + 254 and 0 + 1 chroma pairs (the +1 of the rounding).  The YUVJ420P entry on the
    oracle-averaged planes gives the same bytes (what the round-3 form of this test compared with)."""
    import torch
    rng = np.random.default_rng(422)
    for w, h, n in ((160, 120, 4), (130, 98, 3), (16, 16, 2), (320, 240, 2), (176, 144, 2), (336, 32, 2)):
        cw = w // 2
        ys, cs = w + 8, cw + 24
        Y = rng.integers(0, 256, (n, h, ys), dtype=np.uint8)
        C2 = rng.integers(0, 256, (2, n, h, cs), dtype=np.uint8)
        smooth = (np.add.outer(np.arange(h) * 2, np.arange(cs) * 3) & 255).astype(np.uint8)
        C2[:, : n - 1] = smooth                                   # all but the last frame: codable content; the last: noise
        C2[0, 0, 0::2, : cw // 2], C2[0, 0, 1::2, : cw // 2] = 255, 254   # pairs whose average needs the rounding: 255 + 254 -> 255
        C2[1, 0, 0::2, : cw // 2], C2[1, 0, 1::2, : cw // 2] = 0, 1       # 0 + 1 -> 1
        Y[: n - 1] = (np.add.outer(np.arange(h) * 3, np.arange(ys)) & 255).astype(np.uint8)
        want = [orc.encode_frame_yuv(Y[t], C2[0, t], C2[1, t], w, h) for t in range(n)]
        cap = ctx.encode_bound(w, h) * n
        blob, offs, lens = np.zeros(cap, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
        ctx.encode_yuv422_batch(Y, np.ascontiguousarray(C2[0]), np.ascontiguousarray(C2[1]), ys, cs, h * ys, h * cs, n, w, h, 0, blob, cap, offs, lens)
        for t in range(n):
            assert blob[int(offs[t]):int(offs[t]) + int(lens[t])].tobytes() == want[t], (w, h, t)
        d_blob = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
        d_offs = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        d_lens = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        ctx.encode_yuv422_batch_dev(_t(Y), _t(C2[0]), _t(C2[1]), ys, cs, h * ys, h * cs, n, w, h, 0, d_blob, cap, d_offs, d_lens)
        torch.cuda.synchronize()
        got, go, gl = d_blob.cpu().numpy(), d_offs.cpu().numpy(), d_lens.cpu().numpy()
        for t in range(n):
            assert got[int(go[t]):int(go[t]) + int(gl[t])].tobytes() == want[t], (w, h, t)
        # the 4:2:0 entry on the oracle's averaged planes: the same chunks
        C0 = np.zeros((2, n, h // 2, cs), np.uint8)
        L = orc.lib()
        for c in range(2):
            for t in range(n):
                src = np.ascontiguousarray(C2[c, t])
                L.amvo_yuv422_to_420(src.ctypes.data, cs, cw, h, C0[c, t].ctypes.data, cs)
        b0, o0, l0 = np.zeros(cap, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
        ctx.encode_yuv420_batch(Y, np.ascontiguousarray(C0[0]), np.ascontiguousarray(C0[1]), ys, cs, h * ys, (h // 2) * cs, n, w, h, 0, b0, cap, o0, l0)
        for t in range(n):
            assert b0[int(o0[t]):int(o0[t]) + int(l0[t])].tobytes() == want[t], (w, h, t)
            assert orc.decode_frame(want[t], w, h)[1] == 0          # and every chunk is a valid AMV frame


def test_mixed_stream_keeps_the_parallel_kernels(pkg, orc):
    """Record space is per frame, from the frame's own chunk length (round 4; one stride for all, sized from the batch's
    MEAN chunk, before): in a stream of light frames with a heavy one every sixteenth -- flat and slowly varying pictures,
    noise in between, whose chunks are many times the mean -- no frame is handed to the one-lane serial kernel
    (amvhip_entropy_stats), and every byte and status equals the oracle's.  Both parallel entropy kernels: the
    speculative lanes a small batch gets and the one-lane-per-frame kernel of a chip-filling batch (AMVHIP_SYNC_LANES=1).
    A frame's record space is two words per byte of its chunk + two per block (amvhip_api.hip: `add_rec`, `hi_rec`), never
    more than a frame with every coefficient non-zero could fill -- white noise stays on the parallel kernels too --
    and chunks that overlap in the blob (their lengths add up to more than the blob holds) are decoded all the same."""
    import os
    import torch
    w, h = 160, 120
    rng = np.random.default_rng(416)
    light = [orc.encode_frame(np.full((h, w, 3), v, np.uint8), w, h) for v in (0, 90, 128, 255)]
    ramp = np.add.outer(np.arange(h), np.arange(w))[:, :, None].repeat(3, 2)
    light += [orc.encode_frame(((ramp * k) // 8 & 255).astype(np.uint8), w, h) for k in (1, 2, 3)]
    heavy = [orc.encode_frame(rng.integers(96, 160, (h, w, 3)).astype(np.uint8), w, h) for _ in range(3)]      # mild noise: many small coefficients
    assert min(len(c) for c in heavy) > 8 * max(len(c) for c in light[:4])
    chunks = [(heavy[(i // 16) % 3] if i % 16 == 15 else light[i % len(light)]) for i in range(640)]
    uniq = {c: orc.decode_frame(c, w, h) for c in set(chunks)}
    want = np.stack([uniq[c][0] for c in chunks])
    wst = np.array([uniq[c][1] for c in chunks], np.int32)
    assert (wst == 0).all()
    old = os.environ.get("AMVHIP_SYNC_LANES")
    ctxs = []
    try:
        for lanes in (None, "1"):
            if lanes:
                os.environ["AMVHIP_SYNC_LANES"] = lanes
            ctxs.append(pkg.Context(0))
    finally:
        if old is None:
            os.environ.pop("AMVHIP_SYNC_LANES", None)
        else:
            os.environ["AMVHIP_SYNC_LANES"] = old
    try:
        for c in ctxs:
            got, st = _gpu_decode(c, chunks, w, h)
            assert (st == wst).all() and (got == want).all()
            assert c.entropy_stats(False)["handed_to_serial"] == 0
            # the noisiest picture there is: more records than any frame is given -> the serial kernel, same bytes
            loud = orc.encode_frame(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), w, h)
            mix = chunks[:40] + [loud] + chunks[40:80]
            got, st = _gpu_decode(c, mix, w, h)
            ref, rst, _ = orc.decode_frame(loud, w, h)
            assert (got[40] == ref).all() and st[40] == rst and (got[:40] == want[:40]).all() and (got[41:] == want[40:80]).all()
            # every frame the SAME chunk of the blob: the lengths add up to 300 times what the blob holds
            one = heavy[0]
            blob = np.frombuffer(one + b"\0" * 16, np.uint8).copy()
            n = 300
            d_out = torch.zeros((n, h, c.stride(w)), dtype=torch.uint8, device="cuda:0")
            d_st = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
            c.decode_batch_dev(_t(blob), len(one), _t(np.zeros(n, np.uint64)), _t(np.full(n, len(one), np.uint32)), n, w, h, 0, d_out, d_st,
                               torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert (d_st.cpu().numpy() == 0).all() and (d_out.cpu().numpy() == uniq[one][0][None]).all()
    finally:
        for c in ctxs:
            c.close()


def test_heavy_frames_get_several_lanes_in_a_one_lane_batch(pkg, orc):
    """A batch that gets ONE entropy lane per frame (here forced: AMVHIP_SYNC_LANES=1; on a full-size batch by its size)
    splits itself on the device: frames whose chunk is over twice the batch's mean go to the several-lanes kernel, the
    others stay one lane per frame -- two frame lists, two launches, one record layout; the reconstruction reads every
    frame's lane count from the frame.  Every byte and status equals the oracle's: a stream with a noise frame every
    sixteenth, the same with the heavy frames in a block at either end, frames just under and just over the threshold,
    a batch with nothing heavy in it, a batch of one, damaged heavy and light frames -- and the split reports what it
    did (amvhip_decode_split_stats).  AMVHIP_SPLIT=0 / 8 / 64: no split, other lane counts: the same bytes."""
    import os
    w, h = 160, 120
    rng = np.random.default_rng(20251)
    light = _synth_chunks(orc, 6, w, h)
    noise = [orc.encode_frame(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), w, h) for _ in range(3)]
    mild = [orc.encode_frame(np.clip(orc.synth_frame(SEED, t, w, h).astype(np.int32) + rng.integers(-a, a + 1, (h, w, 3)), 0, 255).astype(np.uint8), w, h)
            for t, a in ((0, 8), (1, 16), (2, 24), (3, 40))]          # between the two: chunks of 1.2 .. 2.5 times a light one
    damaged = []
    for c in (noise[0], light[0]):
        b = bytearray(c)
        b[len(b) // 2] ^= 0x5a
        b[len(b) // 2 + 1] = 0xff
        damaged.append(bytes(b))
        damaged.append(c[: len(c) // 3])
    memo = {}

    def want(c):
        if c not in memo:
            memo[c] = orc.decode_frame(c, w, h)
        return memo[c]

    batches = {
        "every sixteenth": [noise[(i // 16) % 3] if i % 16 == 15 else light[i % 6] for i in range(400)],
        "heavy first": [noise[i % 3] for i in range(70)] + [light[i % 6] for i in range(500)],
        "heavy last": [light[i % 6] for i in range(500)] + [noise[i % 3] for i in range(70)],
        "around the line": [(light + mild)[i % 10] for i in range(300)] + [noise[0]] * 5,
        "nothing heavy": [light[i % 6] for i in range(200)],
        "one frame": [noise[1]],
        "damaged": [light[i % 6] for i in range(150)] + damaged + [noise[2]] * 3 + damaged[::-1],
    }
    old = {k: os.environ.get(k) for k in ("AMVHIP_SYNC_LANES", "AMVHIP_SPLIT")}
    ctxs = {}
    try:
        os.environ["AMVHIP_SYNC_LANES"] = "1"
        for split in (None, "0", "8", "64"):
            if split is None:
                os.environ.pop("AMVHIP_SPLIT", None)
            else:
                os.environ["AMVHIP_SPLIT"] = split
            ctxs[split] = pkg.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        for name, chunks in batches.items():
            exp = np.stack([want(c)[0] for c in chunks])
            est = np.array([want(c)[1] for c in chunks], np.int32)
            mean16 = sum((len(c) + 47) >> 4 for c in chunks) / len(chunks)
            n_heavy = sum(((len(c) + 47) >> 4) > 2 * mean16 for c in chunks)       # the kernel's own rule (amv_split_kernel)
            for split, c in ctxs.items():
                got, st = _gpu_decode(c, chunks, w, h)
                assert (st == est).all(), (name, split, np.nonzero(st != est)[0][:5])
                assert (got == exp).all(), (name, split, np.nonzero((got != exp).reshape(len(chunks), -1).any(1))[0][:5])
                sp = c.decode_split_stats()
                if split == "0":
                    assert sp == {"heavy": 0, "light": 0}, (name, sp)
                else:
                    assert sp == {"heavy": n_heavy, "light": len(chunks) - n_heavy}, (name, split, sp)
            if name in ("every sixteenth", "heavy first", "heavy last", "around the line", "damaged"):
                assert 0 < n_heavy < len(chunks), name
            if name in ("nothing heavy", "one frame"):
                assert n_heavy == 0, name
        # a batch decoded with several lanes per frame anyway makes no split
        plain = pkg.Context(0)
        try:
            got, st = _gpu_decode(plain, batches["every sixteenth"], w, h)
            assert plain.decode_split_stats() == {"heavy": 0, "light": 0}
            assert (got == np.stack([want(c)[0] for c in batches["every sixteenth"]])).all()
        finally:
            plain.close()
    finally:
        for c in ctxs.values():
            c.close()


def test_decode_lengths_that_overflow_the_layout(pkg, orc):
    """Chunk lengths are whatever the caller wrote: 20 consecutive frames claim ~4 GB each (a 32-bit prefix over one
    workgroup's 256 lengths would wrap after 16 of them, and a frame's workspace window would start below its
    predecessor's).  The layout saturates instead: those frames (clamped to the blob, as include/amvhip.h promises) and
    the ones behind them take the serial kernel, every frame decodes to what the oracle makes of the bytes from its offset
    to the end of the blob, and the frames in front of them are untouched.  Both parallel entropy kernels, and BOTH layout
    routes -- the one-workgroup kernel a batch of up to 16 384 frames gets (round 6) and the three launches of larger ones
    (AMVHIP_LAYOUT=large) -- which must lay out the same lines: the same frames handed to the serial kernel, the same bytes.
    The huge lengths sit inside one block of 256 frames, and across the boundary of two."""
    import os
    import torch
    w, h = 160, 120
    chunks = _synth_chunks(orc, 12, w, h)
    n = 600
    seq = [chunks[i % len(chunks)] for i in range(n)]
    blob, offs, lens, nbytes = _blob_of(seq)
    want = np.stack([orc.decode_frame(c, w, h)[0] for c in chunks])
    keep = {k: os.environ.get(k) for k in ("AMVHIP_SYNC_LANES", "AMVHIP_LAYOUT")}
    ctxs = {}
    try:
        for lanes in (None, "1"):
            for layout in (None, "large"):
                for k, v in (("AMVHIP_SYNC_LANES", lanes), ("AMVHIP_LAYOUT", layout)):
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                ctxs[(lanes, layout)] = pkg.Context(0)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        for huge in (range(100, 120), range(250, 270), range(0, 0)):
            lens2 = lens.copy()
            for i in huge:
                lens2[i] = 0xfffffff0 - (i & 3)
            want_huge = {i: orc.decode_frame(blob[int(offs[i]):nbytes].tobytes(), w, h) for i in huge}
            handed = {}
            for key, c in ctxs.items():
                d_out = torch.full((n, h, c.stride(w)), 0x5A, dtype=torch.uint8, device="cuda:0")
                d_st = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
                c.decode_batch_dev(_t(blob), nbytes, _t(offs), _t(lens2), n, w, h, 0, d_out, d_st, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                handed[key] = c.entropy_stats(False)["handed_to_serial"]
                got, st = d_out.cpu().numpy(), d_st.cpu().numpy()
                for i in range(n):
                    if i in want_huge:
                        assert st[i] == want_huge[i][1] and (got[i] == want_huge[i][0]).all(), (key, i)
                    else:
                        assert st[i] == 0 and (got[i] == want[i % len(chunks)]).all(), (key, i)
            for lanes in (None, "1"):          # the two routes laid out the same windows: the same frames ran out of them
                assert handed[(lanes, None)] == handed[(lanes, "large")], (list(huge)[:1], handed)
            assert (handed[(None, None)] > 0) == (len(huge) > 0), handed
    finally:
        for c in ctxs.values():
            c.close()


def test_adpcm_decode_large_ragged_batch(ctx, orc, amv1):
    """the batch decode against the oracle sample for sample on 5 000 chunks of every length from none to 2 000 nibble
    bytes -- shorter than one tile, an exact multiple, ragged tails --, start indices 0..88 and one beyond the table
    (clamped), predictors at both rails, random nibbles and runs that saturate predictor and index, chunks wherever they
    fall in the blob, empty / header-only / truncated ones in between (they write nothing), and the end state of every
    chunk; the same batch cut to 100 chunks.  (Written for a lane-per-chunk decode kernel that round 4 measured and did not
    keep -- DESIGN.md section 7; it stays as the widest check of amv_adpcm_decode_kernel.)"""
    rng = np.random.default_rng(40404)
    n = 5000
    chunks = []
    for i in range(n):
        nb = int(rng.integers(0, 2001)) if i % 7 else (0, 1, 63, 64, 65, 127, 128, 689, 1024)[(i // 7) % 9]
        pred = (-32768, 32767, 0, int(rng.integers(-32768, 32768)))[i % 4]
        idx = int(rng.integers(0, 90)) if i % 11 else 200
        body = rng.integers(0, 256, nb, dtype=np.uint8)
        if i % 5 == 0:
            body[:] = rng.choice([0x77, 0xff, 0x0f, 0x88, 0x00], nb)          # runs that saturate predictor and index
        hdr = np.array([pred & 0xff, (pred >> 8) & 0xff, idx, 0, (2 * nb) & 0xff, ((2 * nb) >> 8) & 0xff, 0, 0], np.uint8)
        c = hdr.tobytes() + body.tobytes()
        if i % 97 == 0:
            c = (b"", c[:8], c[:5])[(i // 97) % 3]                            # nothing to decode: the kernel leaves the PCM alone
        chunks.append(c)
    chunks += list(amv1["audio"][:64])
    n = len(chunks)
    blob, offs, lens, nbytes = _blob_of(chunks, 1)
    pcm_offs = np.cumsum([0] + [2 * max(len(c) - 8, 0) for c in chunks]).astype(np.uint64)
    want = [orc.adpcm_decode_chunk(c)[0] for c in chunks]
    for count in (n, 100):
        pcm = np.full(int(pcm_offs[count]) + 8, 0x5A5A, np.int16)
        fin = np.full((count, 2), -7, np.int32)
        ctx.adpcm_decode_batch(blob, nbytes, offs[:count].copy(), lens[:count].copy(), count, pcm, pcm.size, pcm_offs[:count].copy(), fin)
        for i in range(count):
            got = pcm[int(pcm_offs[i]):int(pcm_offs[i + 1])]
            assert (got == want[i]).all(), (count, i, len(chunks[i]))
            if want[i].size:
                assert fin[i, 0] == want[i][-1], (count, i)
        assert (pcm[int(pcm_offs[count]):] == 0x5A5A).all()
