"""Pins of the CPU oracle (no GPU).  The oracle is trusted by the GPU parity tests only because
these pass:
  * video decode  -- the reference's own fixture AMV1.amv must hash to the values amvlib itself
                     produced, quirk on and off;
  * audio decode, amvlib's WAV encoder, forward DCT -- bit-equal to oracle/_ref/libamvref.so,
                     which is compiled from the reference's AdpcmIma.c and jfdctint.c;
  * entropy coder -- oracle/_ref/libavcref.so holds the reference's OWN mjpegenc.c / mjpeg.c (ff_mjpeg_encode_mb,
                     stuffing, escape_FF, trailer) behind oracle/ref_harness.c: (i) re-coding the coefficients the
                     oracle's decoder recovers from every chunk of the real AMV1.amv with the reference's coder gives
                     the device-made chunk back byte for byte -- the oracle's Huffman DECODER is the inverse of the
                     reference's coder; (ii) the oracle's ENCODER writes the bytes the reference's coder writes for the
                     same coefficients;
  * committed golden vectors of the synthetic clips (tests/golden/synth_golden.json).
"""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, SEED

AMVLIB_HASH = 0xADC922C6366237B5        # amvlib over all 252 frames of AMV1.amv
AMVLIB_HASH_FIXED_ZZ = 0xA3F28348069FCE7C  # same with Zig_Zag[3][4]=31
AMV1_PCM_BYTES = 672504                 # decoded audio of AMV1.amv


def test_amv1_header(amv1):
    # reference AmvLibTest prints these: 128x96, 12 fps, 16 kHz mono, 252+252 chunks
    assert (amv1["info"]["width"], amv1["info"]["height"], amv1["info"]["fps"]) == (128, 96, 12)
    assert amv1["info"]["sample_rate"] == 16000 and amv1["info"]["us_per_frame"] == 83333
    assert len(amv1["video"]) == 252 and len(amv1["audio"]) == 252
    sizes = [len(v) for v in amv1["video"]]
    assert (min(sizes), max(sizes)) == (1410, 2924) and sum(sizes) == 606456


@pytest.mark.parametrize("flags,want", [(0, AMVLIB_HASH), (1, AMVLIB_HASH_FIXED_ZZ)])
def test_video_decode_pinned_by_amvlib_hash(orc, amv1, flags, want):
    h = orc.SURVEY_FNV_SEED
    for chunk in amv1["video"]:
        out, st, ok = orc.decode_frame(chunk, 128, 96, flags)
        assert st == 0 and ok == 48
        h = orc.fnv1a64(h, out)
    assert h == want


def test_audio_decode_matches_reference_build(orc, amv1):
    R = orc.ref()
    total = 0
    for a in amv1["audio"]:
        mine, hdr = orc.adpcm_decode_chunk(a)
        total += mine.size * 2
        assert mine.size == 2 * (len(a) - 8) and 0 <= mine.size - hdr <= 1   # header counts samples, payload is whole bytes
        if R is None:
            continue
        c = orc.RefADPCMContext()
        c.channel = 1
        c.status[0].predictor = int(np.frombuffer(a[:2], "<i2")[0])
        c.status[0].step_index = a[2]
        pcm = np.zeros(4096, np.int16)
        dl = ctypes.c_int(0)
        buf = (ctypes.c_ubyte * (len(a) + 8)).from_buffer_copy(a + b"\0" * 8)
        R.AdpcmImaDecodeFrame(ctypes.byref(c), pcm.ctypes.data, ctypes.byref(dl), ctypes.byref(buf, 8), len(a) - 8)
        assert (pcm[:mine.size] == mine).all()
    assert total == AMV1_PCM_BYTES


def test_fdct_matches_reference_build(orc):
    R = orc.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(7)
    for it in range(5000):
        lo, hi = ((-128, 128), (-255, 256), (0, 256))[it % 3]
        a = rng.integers(lo, hi, 64).astype(np.int16)
        b = a.copy()
        R.ff_jpeg_fdct_islow(a.ctypes.data)
        orc.lib().amvo_fdct_islow(b.ctypes.data)
        assert (a == b).all()


def test_fdct_outputs_fit_dctelem(orc):
    """What lets the device's transform_block (csrc/amv_encode_common.h) leave out DCTELEM's 16-bit truncation
    (dsputil.h:38) and use 24-bit multiply-adds in the column pass: for samples of -128..127 -- all the planes ever hold --
    every output of jfdctint.c's row pass is inside +-16385 and every output of the column pass inside +-8193.  The passes
    are restated here in exact integer arithmetic (checked against the oracle's amvo_fdct_islow, itself pinned by the
    reference's jfdctint.c), the bound is the sum of |weight| over a pass's eight inputs, and the inputs that reach it --
    every output's own sign pattern at full swing -- go through both passes."""
    def one_pass(d, row):
        shift, up = (13 - 4, 4) if row else (13 + 4, None)
        desc = lambda v, n: (v + (1 << (n - 1))) >> n
        t0, t7, t1, t6 = d[0] + d[7], d[0] - d[7], d[1] + d[6], d[1] - d[6]
        t2, t5, t3, t4 = d[2] + d[5], d[2] - d[5], d[3] + d[4], d[3] - d[4]
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        o = [0] * 8
        o[0] = (t10 + t11) << up if row else desc(t10 + t11, 4)
        o[4] = (t10 - t11) << up if row else desc(t10 - t11, 4)
        z1 = (t12 + t13) * 4433
        o[2], o[6] = desc(z1 + t13 * 6270, shift), desc(z1 - t12 * 15137, shift)
        z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
        z5 = (z3 + z4) * 9633
        a4, a5, a6, a7 = t4 * 2446, t5 * 16819, t6 * 25172, t7 * 12299
        z1, z2, z3, z4 = z1 * -7373, z2 * -20995, z3 * -16069 + z5, z4 * -3196 + z5
        o[7], o[5], o[3], o[1] = desc(a4 + z1 + z3, shift), desc(a5 + z2 + z4, shift), desc(a6 + z2 + z3, shift), desc(a7 + z1 + z4, shift)
        return o

    def fdct2(block):                      # jfdctint.c:262-: rows, then columns
        rows = [one_pass([int(v) for v in block[r]], True) for r in range(8)]
        cols = [one_pass([rows[r][c] for r in range(8)], False) for c in range(8)]
        return rows, [[cols[c][r] for c in range(8)] for r in range(8)]

    rng = np.random.default_rng(12)
    for it in range(300):                  # the restatement is the oracle's transform
        a = rng.integers(-128, 128, (8, 8)).astype(np.int16)
        b = a.copy().reshape(64)
        orc.lib().amvo_fdct_islow(b.ctypes.data)
        assert (np.array(fdct2(a)[1]).reshape(64) == b).all()
    # weights of a pass: the response to unit inputs, exact in rational arithmetic = the integer pass without its rounding
    big = 1 << 20
    w = np.array([one_pass([big if i == k else 0 for i in range(8)], True) for k in range(8)], dtype=np.float64) / big   # [input][output]
    worst_row = worst_col = 0
    for u in range(8):
        for v in range(8):
            # the block that drives output (v, u) hardest: every sample at the end of its range that the output's sign pattern asks for
            pat = np.sign(np.outer(w[:, v], w[:, u]))
            for lo_hi in ((-128, 127), (127, -128)):
                blk = np.where(pat >= 0, lo_hi[1], lo_hi[0])
                rows, out = fdct2(blk)
                worst_row = max(worst_row, max(abs(x) for r in rows for x in r))
                worst_col = max(worst_col, max(abs(x) for r in out for x in r))
    for it in range(2000):                 # ... and two-level blocks at random
        rows, out = fdct2(rng.choice([-128, 127], (8, 8)))
        worst_row = max(worst_row, max(abs(x) for r in rows for x in r))
        worst_col = max(worst_col, max(abs(x) for r in out for x in r))
    assert 16000 < worst_row <= 16385 and 8000 < worst_col <= 8193, (worst_row, worst_col)
    bound_row = max(np.abs(w[:, k]).sum() * 128 for k in range(8))
    assert bound_row <= 16385, bound_row


def test_wav_layout_encoder_matches_reference_build(orc):
    R = orc.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(11)
    for it in range(40):
        fs = 8 * int(rng.integers(1, 200))
        x = (rng.integers(-32768, 32768, fs + 1) if it % 2 else orc.synth_audio(SEED, it * 999, fs + 1) * 4).astype(np.int16)
        idx = int(rng.integers(0, 89))
        c = orc.RefADPCMContext()
        c.status[0].step_index = idx
        want = np.zeros(4 + fs // 2 + 16, np.uint8)
        n = R.AdpcmImaEncodeFrame(ctypes.byref(c), 1, fs, want.ctypes.data, want.size, x.ctypes.data)
        got = np.zeros_like(want)
        st = np.array([0, idx], np.int32)
        m = orc.lib().amvo_adpcm_wav_encode_frame(x.ctypes.data, fs, st.ctypes.data, got.ctypes.data)
        assert n == m == 4 + fs // 2 and (want[:n] == got[:n]).all()
        assert (st[0], st[1]) == (c.status[0].prev_sample, c.status[0].step_index)


def test_idct_dc_shortcuts_are_identities(orc):
    """amvlib's all-AC-zero shortcuts (AmvJpeg.c:1087-1092,1134-1140) equal the full butterflies, so
    the HIP kernel may drop them: the oracle keeps the shortcuts, here they are compared with a
    numpy restatement that never takes them (exhaustive over every dequantised DC value)."""
    W1, W2, W3, W5, W6, W7 = 2841, 2676, 2408, 1609, 1108, 565

    def full(b, col):
        b = b.astype(np.int64)
        up, bias, rnd, dn, out = (256, 8192, 4, 3, 14) if col else (2048, 128, 0, 0, 8)
        x0 = b[0] * up + bias; x1 = b[4] * up; x2, x3, x4, x5, x6, x7 = b[6], b[2], b[1], b[7], b[5], b[3]
        x8 = W7 * (x4 + x5) + rnd; x4 = (x8 + (W1 - W7) * x4) >> dn; x5 = (x8 - (W1 + W7) * x5) >> dn
        x8 = W3 * (x6 + x7) + rnd; x6 = (x8 - (W3 - W5) * x6) >> dn; x7 = (x8 - (W3 + W5) * x7) >> dn
        x8 = x0 + x1; x0 = x0 - x1; x1 = W6 * (x3 + x2) + rnd
        x2 = (x1 - (W2 + W6) * x2) >> dn; x3 = (x1 + (W2 - W6) * x3) >> dn
        x1 = x4 + x6; x4 = x4 - x6; x6 = x5 + x7; x5 = x5 - x7
        x7 = x8 + x3; x8 = x8 - x3; x3 = x0 + x2; x0 = x0 - x2
        x2 = (181 * (x4 + x5) + 128) >> 8; x4 = (181 * (x4 - x5) + 128) >> 8
        return np.stack([x7 + x1, x3 + x2, x0 + x4, x8 + x6, x8 - x6, x0 - x4, x3 - x2, x7 - x1]) >> out

    dcs = np.arange(-2048 * 9, 2048 * 9 + 1, dtype=np.int64)      # every DC * step the tables allow
    z = np.zeros((8, dcs.size), np.int64); z[0] = dcs
    rows = full(z, False)                                            # row pass of a DC-only block
    assert (rows == dcs * 8).all()
    cols = np.clip(full(np.concatenate([rows[:1], np.zeros((7, dcs.size), np.int64)]), True), -256, 255)
    blk = np.zeros(64, np.int32)
    for i in range(0, dcs.size, 97):                                 # and the oracle agrees
        blk[:] = 0; blk[0] = dcs[i]
        orc.lib().amvo_idct_block(blk.ctypes.data)
        assert (blk == cols[0, i]).all()


def test_error_semantics(orc):
    w, h = 160, 120
    chunk = orc.encode_frame(orc.synth_frame(SEED, 3, w, h), w, h)
    good, st, ok = orc.decode_frame(chunk, w, h)
    assert st == 0 and ok == 80
    # truncated: zero-extended, flagged, still every MCU attempted
    out, st, ok = orc.decode_frame(chunk[:len(chunk) // 2], w, h)
    assert st & orc.ST_TRUNCATED
    # a run of ones is not a Huffman code: FORMAT, decoding stops, later MCUs stay zero (AMVDec.c:283)
    bad = bytearray(chunk); bad[300:340] = b"\xff\x00" * 20
    out, st, ok = orc.decode_frame(bytes(bad), w, h)
    assert st & orc.ST_FORMAT and ok < 80
    row0_mcu = ok // 10
    assert not out[: h - 16 * (row0_mcu + 1)].any()                  # rows of undecoded MCU rows (stored bottom-up)
    assert (out[h - 16 * row0_mcu:] == good[h - 16 * row0_mcu:]).all()


def test_encode_round_trip_quality(orc):
    """encode -> bit-exact decoder -> PSNR against the RGB source.  amvlib's colour matrix
    (AmvJpeg.c:808-810) is not the inverse of FFmpeg's RGB->YCbCr (colorspace.h), which bounds the
    figure; the thresholds are the measured values of the oracle minus 0.3 dB."""
    for (w, h), lo_amv, lo_mjpeg in (((160, 120), 25.4, 25.9), ((320, 240), 26.6, 27.3)):
        ps = {0: [], 128: []}
        for t in (0, 50, 190):
            src = orc.synth_frame(SEED, t, w, h)
            for qb in ps:
                out, st, ok = orc.decode_frame(orc.encode_frame(src, w, h, qbias=qb), w, h, orc.FLAG_ZIGZAG_FIXED)
                assert st == 0
                ps[qb].append(orc.psnr(src, out[:, : w * 3].reshape(h, w, 3)[:, :, ::-1]))
        assert np.mean(ps[0]) > lo_amv and np.mean(ps[128]) > lo_mjpeg, (w, h, np.mean(ps[0]), np.mean(ps[128]))


def test_adpcm_round_trip_and_framing(orc):
    pcm = orc.synth_audio(SEED, 0, 1378 * 4)
    idx, dec = 0, []
    for k in range(4):
        chunk, idx = orc.adpcm_encode_chunk(pcm[1378 * k: 1378 * (k + 1)], idx)
        assert len(chunk) == 8 + 689 and int.from_bytes(chunk[4:8], "little") == 1378
        dec.append(orc.adpcm_decode_chunk(chunk)[0])
    dec = np.concatenate(dec).astype(np.float64)
    err = dec - pcm
    snr = 10 * np.log10((pcm.astype(np.float64) ** 2).sum() / (err ** 2).sum())
    assert snr > 20, snr
    # adpcm.c:469-477 framing: 22050 Hz / 16 fps -> frame_size 1378, pairs 689 with the 1 Hz resync
    extra, written = ctypes.c_uint32(0), ctypes.c_uint64(0)
    pairs = [orc.lib().amvo_adpcm_amv_pairs(1378, 22050, ctypes.byref(extra), ctypes.byref(written)) for _ in range(32)]
    assert pairs[0] == 689 and sum(pairs) * 2 == written.value and written.value % 22050 == 0


def test_synthetic_golden_vectors(orc):
    """committed fixtures: tests/golden/synth_golden.json (made by tests/golden/make_golden.py)"""
    gold = json.load(open(os.path.join(GOLDEN, "synth_golden.json")))
    for case in gold["video"]:
        w, h, t = case["w"], case["h"], case["frame"]
        src = orc.synth_frame(gold["seed"], t, w, h)
        assert "%016x" % orc.fnv1a64(orc.FNV_BASIS, src) == case["rgb_fnv"]
        chunk = orc.encode_frame(src, w, h, qbias=case["qbias"])
        assert len(chunk) == case["chunk_len"] and "%016x" % orc.fnv1a64(orc.FNV_BASIS, np.frombuffer(chunk, np.uint8)) == case["chunk_fnv"]
        out, st, ok = orc.decode_frame(chunk, w, h, case["flags"])
        assert st == 0 and "%016x" % orc.fnv1a64(orc.FNV_BASIS, out) == case["bgr_fnv"]
    a = gold["audio"]
    pcm = orc.synth_audio(gold["seed"], a["first"], a["n"])
    chunk, idx = orc.adpcm_encode_chunk(pcm, a["step_in"])
    assert "%016x" % orc.fnv1a64(orc.FNV_BASIS, pcm) == a["pcm_fnv"] and idx == a["step_out"]
    assert "%016x" % orc.fnv1a64(orc.FNV_BASIS, np.frombuffer(chunk, np.uint8)) == a["chunk_fnv"]


def _need_avcref(orc):
    R = orc.avcref()
    if R is None:
        pytest.skip("oracle/_ref/libavcref.so is built only where /root/reference exists")
    return R


def test_entropy_decoder_inverts_reference_coder_on_amv1(orc, amv1):
    """All 252 device-made chunks of the reference's fixture: decode with the oracle (coefficients, DC already
    accumulated = what encode_block takes with its predictors at 0), re-code with the reference's own
    ff_mjpeg_encode_mb + ff_mjpeg_encode_picture_trailer (mjpegenc.c:437-450,345-355): byte-identical."""
    _need_avcref(orc)
    for i, chunk in enumerate(amv1["video"]):
        _, st, ok, coef = orc.decode_frame(chunk, 128, 96, want_coef=True)
        assert st == 0 and ok == 48
        assert chunk[:2] == b"\xff\xd8" and orc.ref_mjpeg_encode_scan(coef) == chunk[2:], i


def test_entropy_coder_matches_reference_build(orc):
    """rows a20 / a21: the oracle encoder's scan bytes == the reference's entropy coder on the oracle's own
    quantised blocks -- synthetic clips at both sizes and biases, partial MCUs, noise (many FF bytes, ZRL runs,
    11-bit magnitudes), flat and saturated frames"""
    _need_avcref(orc)
    rng = np.random.default_rng(77)
    cases = []
    for w, h, n in ((160, 120, 6), (320, 240, 3), (176, 144, 2), (130, 98, 2), (16, 16, 2)):
        cases += [(orc.synth_frame(SEED, 31 * t, w, h), w, h) for t in range(n)]
    for w, h in ((160, 120), (48, 32)):
        cases.append((rng.integers(0, 256, (h, w, 3), dtype=np.uint8), w, h))                       # noise
        cases.append((np.full((h, w, 3), 255, np.uint8), w, h))                                      # white
        cases.append((np.zeros((h, w, 3), np.uint8), w, h))                                          # black
        chk = ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8)               # 1-pixel checker
        cases.append((np.repeat(chk[:, :, None], 3, 2), w, h))
    escaped = 0
    for src, w, h in cases:
        for qbias in (0, 128):
            chunk, coef = orc.encode_frame(src, w, h, qbias=qbias, want_coef=True)
            assert chunk[:2] == b"\xff\xd8" and chunk[-2:] == b"\xff\xd9"
            assert orc.ref_mjpeg_encode_scan(coef) == chunk[2:], (w, h, qbias)
            escaped += chunk.count(b"\xff\x00")
    assert escaped > 50


def test_huffman_tables_match_reference_build(orc):
    """the K.3 specifications compiled into the oracle decode what mjpeg.c:62-127 specifies, and the canonical
    code assignment equals ff_mjpeg_build_huffman_codes (mjpeg.c:129-147): checked by coding single symbols"""
    R = _need_avcref(orc)
    for t in range(4):
        bits = np.zeros(17, np.uint8)
        vals = np.zeros(256, np.uint8)
        n = R.amvref_mjpeg_huffman_spec(t, bits.ctypes.data, vals.ctypes.data)
        assert n == (12 if t < 2 else 162) and int(bits[1:].sum()) == n
        size = np.zeros(256, np.uint8)
        code = np.zeros(256, np.uint16)
        R.amvref_mjpeg_huffman_codes(t, size.ctypes.data, code.ctypes.data)
        assert int((size > 0).sum()) == n and size.max() <= 16
        # Kraft sum of a complete-but-one prefix code (JPEG reserves the all-ones code)
        assert sum(2.0 ** -int(s) for s in size if s) == 1.0 - 2.0 ** -int(size.max())


def test_simple_idct_matches_reference_build(orc):
    """row a15: the oracle's simple_idct restatement against the reference's own simple_idct.c object --
    random dense blocks, sparse blocks (the DC-only row shortcut of idctRowCondDC is NOT the general formula),
    dequantised AMV coefficients, and full-range int16 values where the int16 stores between the passes wrap"""
    R = _need_avcref(orc)
    L = orc.lib()
    rng = np.random.default_rng(3)
    blocks = [rng.integers(-2048, 2048, 64), rng.integers(-32768, 32768, 64), np.zeros(64, np.int64)]
    for mag in (8, 64, 512, 2048, 6000, 16384, 32767):              # dense: row results beyond int16 wrap in the stores
        blocks += [rng.integers(-mag, mag + 1, 64) for _ in range(60)]
    for _ in range(300):
        b = np.zeros(64, np.int64)
        k = int(rng.integers(1, 12))
        b[rng.integers(0, 64, k)] = rng.integers(-1500, 1500, k)
        blocks.append(b)
    for dc in (-32768, -4096, -1, 0, 1, 1023, 1024, 2047, 2048, 4095, 32767):      # DC only: every row takes the shortcut
        b = np.zeros(64, np.int64)
        b[0] = dc
        blocks.append(b)
        b = b.copy()
        b[8] = 7                                                                   # row 1 has only its first value
        blocks.append(b)
    for _ in range(200):
        blocks.append(rng.integers(-300, 300, 64) * (rng.random(64) < 0.3))
    arr = np.array(blocks, np.int16)
    want = arr.copy()
    R.amvref_simple_idct(want.ctypes.data, len(want))
    got = arr.copy()
    for i in range(len(got)):
        L.amvo_simple_idct(got[i].ctypes.data)
    assert (got == want).all()
    # the put form = the same sums clipped to 0..255
    for i in range(0, len(arr), 7):
        blk = arr[i].copy()
        px = np.zeros(64, np.uint8)
        L.amvo_simple_idct_put(px.ctypes.data, 8, blk.ctypes.data)
        assert (px == np.clip(want[i], 0, 255)).all()


def test_q60_tables_match_reference_header(orc):
    """the quantiser tables sp5xdec.c:60-61 puts in front of an AMV scan (sp5x_quant_table[10], [11])"""
    R = _need_avcref(orc)
    for chroma in (0, 1):
        a, b = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
        orc.lib().amvo_q60_table(chroma, a.ctypes.data)
        R.amvref_sp5x_quant(chroma, b.ctypes.data)
        assert (a == b).all()
    # the wrapper's frame header says 4:2:0 with component ids 1,2,3, DC/AC tables 0 for Y and 1 for C (sp5x.h:27-51)
    seg = np.zeros(64, np.uint8)
    n = R.amvref_sp5x_segment(1, seg.ctypes.data, 64)
    assert bytes(seg[:n])[:5] == b"\xff\xc0\x00\x11\x08" and bytes(seg[9:n]) == bytes([3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])
    n = R.amvref_sp5x_segment(2, seg.ctypes.data, 64)
    assert bytes(seg[:n]) == bytes([0xff, 0xda, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0])


def test_ffmpeg_compat_decode_properties(orc, amv1):
    """whole-frame FFmpeg-compat decode (not buildable from the reference here: mjpegdec.c needs configure's
    ENABLE_* switches), checked through its parts and its geometry: every block of a frame equals
    reference simple_idct (clipped) of the dequantised coefficients, placed where mjpegdec.c:672-677 puts it."""
    R = _need_avcref(orc)
    L = orc.lib()
    for chunk, w, h in [(amv1["video"][0], 128, 96), (amv1["video"][200], 128, 96),
                        (orc.encode_frame(orc.synth_frame(SEED, 9, 160, 120), 160, 120), 160, 120),
                        (orc.encode_frame(orc.synth_frame(SEED, 4, 130, 98), 130, 98), 130, 98)]:
        buf, st, ok = orc.decode_frame_ffmpeg(chunk, w, h)
        _, st2, ok2, coef = orc.decode_frame(chunk, w, h, want_coef=True)
        assert st == st2 == 0 and ok == ok2
        Y, Cb, Cr = orc.yuv_planes(buf, w, h)
        mcw, mch = (w + 15) // 16, (h + 15) // 16
        for plane, v, comp in ((Y, 2, 0), (Cb, 1, 1), (Cr, 1, 2)):
            full = np.zeros((8 * mch * v, 8 * mcw * v), np.uint8)               # the padded, unflipped component
            for m in range(mcw * mch):
                my, mx = divmod(m, mcw)
                for k in ((0, 1, 2, 3) if comp == 0 else (3 + comp,)):
                    blk = np.zeros(64, np.int16)
                    L.amvo_ffmpeg_dequant_block(coef[m * 6 + k].ctypes.data, comp, blk.ctypes.data)
                    R.amvref_simple_idct(blk.ctypes.data, 1)
                    by, bx = (2 * my + (k >> 1), 2 * mx + (k & 1)) if comp == 0 else (my, mx)
                    full[8 * by: 8 * by + 8, 8 * bx: 8 * bx + 8] = np.clip(blk, 0, 255).reshape(8, 8)
            start = v * (8 * mch - ((h // 2) & 7)) - 1
            for p in range(plane.shape[0]):
                r = start - p
                want = full[r, : plane.shape[1]] if 0 <= r < full.shape[0] else np.zeros(plane.shape[1], np.uint8)
                assert (plane[p] == want).all(), (w, h, comp, p)


def test_ffmpeg_keep_restatement_properties(orc, amv1):
    """amvo_decode_frame_ffmpeg_keep (AMVHIP_FLAG_FFMPEG_KEEP; mjpegdec.c:699-716) -- RESTATEMENT ONLY, no reference build
    pins this slice -- against the plain FFmpeg-compat decode and the geometry of mjpegdec.c:672-677: (1) an undamaged chunk
    writes exactly the plain mode's bytes wherever a block lands and nothing else; (2) a damaged chunk fails where the plain
    mode fails (same status, blocks_ok // 6 == nmcu_ok), every whole MCU in front holds the plain mode's bytes, the failing
    MCU's blocks in front of the failing one hold dequantised + simple_idct'ed coefficients at their place, and every other
    byte is the buffer's own."""
    L = orc.lib()
    rng = np.random.default_rng(99)
    seen_partial = 0
    for w, h, chunks in ((128, 96, amv1["video"][:60]), (46, 30, [orc.encode_frame(orc.synth_frame(SEED, t, 46, 30), 46, 30) for t in range(24)])):
        mcw, mch = (w + 15) // 16, (h + 15) // 16
        cw, chh = (w + 1) // 2, (h + 1) // 2
        fb = w * h + 2 * cw * chh

        def block_bytes(block):                 # frame-buffer positions of block number `block` (MCU-major, Y0..Y3 Cb Cr)
            m, k = divmod(block, 6)
            my, mx = divmod(m, mcw)
            comp = 0 if k < 4 else k - 3
            v = 2 if comp == 0 else 1
            pw, ph = (w, h) if comp == 0 else (cw, chh)
            base = 0 if comp == 0 else w * h + (comp - 1) * cw * chh
            sy = (2 * my + (k >> 1) if comp == 0 else my) * 8
            sx = (2 * mx + (k & 1) if comp == 0 else mx) * 8
            start = v * (8 * mch - ((h // 2) & 7)) - 1
            pos = {}
            for i in range(8):
                p = start - (sy + i)
                if 0 <= p < ph:
                    for j in range(8):
                        if sx + j < pw:
                            pos[base + p * pw + sx + j] = (i, j)
            return pos, comp

        for t, chunk in enumerate(chunks):
            c = bytearray(chunk)
            if t % 3 != 0:                          # (one flipped bit usually decodes to the end: the codes resynchronise)
                for _ in range(3):
                    c[int(rng.integers(8, len(c) - 4))] ^= 1 << int(rng.integers(0, 8))
            if t % 3 == 2:
                c = c[: int(rng.integers(6, len(c) - 2))]
            c = bytes(c)
            before = rng.integers(0, 256, fb, dtype=np.uint8)
            got, st, blocks = orc.decode_frame_ffmpeg_keep(c, w, h, before)
            plain, pst, nmcu = orc.decode_frame_ffmpeg(c, w, h)
            assert st == pst and blocks // 6 == nmcu and (st != 0 or blocks == 6 * mcw * mch)
            expect = before.copy()
            coef, est = orc.entropy_blocks(c, 6 * mcw * mch)          # the entropy stage alone, block by block
            assert est == st and len(coef) == blocks
            for blk in range(blocks):
                pos, comp = block_bytes(blk)
                seen_partial += blk >= 6 * nmcu
                deq = np.zeros(64, np.int16)
                L.amvo_ffmpeg_dequant_block(np.ascontiguousarray(coef[blk]).ctypes.data, comp, deq.ctypes.data)
                px = np.zeros(64, np.uint8)
                L.amvo_simple_idct_put(px.ctypes.data, 8, deq.ctypes.data)
                for q, (i, j) in pos.items():
                    expect[q] = px[8 * i + j]
                    if blk < 6 * nmcu:
                        assert plain[q] == px[8 * i + j]                # a whole MCU in front: the plain mode's bytes
            assert (got == expect).all(), (w, h, t, st, blocks)
    assert seen_partial >= 3


def test_oracle_under_sanitizers(amv1):
    """the restatement, built with -fsanitize=address,undefined, over every chunk of the reference
    clip, truncated / damaged / random chunks, wrong geometries and the encoders -- where the reference itself reads out
    of bounds (AdpcmIma.c:225-237, AmvJpeg.c:967-969,1167) the oracle must define the outcome without doing so"""
    import subprocess
    here = os.path.join(os.path.dirname(GOLDEN), "..", "oracle")
    subprocess.run(["make", "-C", here, "-s", "sanitize"], check=True)
    out = subprocess.run([os.path.join(here, "sanitize_main"), amv1["path"]], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sanitized: 252 video chunks, 252 audio chunks" in out.stdout


def test_img_resample_matches_reference_build(orc):
    """the oracle's restatement of img_resample (the sws_scale shim's rescaler) and of the filter
    builder behind it against the reference's own imgresample.c / resample2.c objects: down- and up-scaling, odd
    ratios, sizes whose chroma planes round down, flat and noisy content"""
    R = _need_avcref(orc)
    rng = np.random.default_rng(8)
    for iw, ih, ow, oh in ((640, 480, 160, 120), (320, 240, 160, 120), (352, 288, 160, 120), (176, 144, 160, 120),
                           (160, 120, 320, 240), (130, 98, 64, 48), (160, 120, 160, 96), (162, 122, 160, 120), (64, 48, 66, 50)):
        for kind in range(3):
            n = orc.yuv420_bytes(iw, ih)
            src = (rng.integers(0, 256, n, dtype=np.uint8) if kind == 0 else
                   np.full(n, 255 if kind == 1 else 0, np.uint8))
            if kind == 2:
                src[: iw * ih] = (np.add.outer(np.arange(ih) * 3, np.arange(iw) * 5) & 255).astype(np.uint8).ravel()
            mine = orc.img_resample_yuv420(src, iw, ih, ow, oh)
            ref = np.zeros_like(mine)
            assert R.amvref_img_resample(src.ctypes.data, iw, ih, ref.ctypes.data, ow, oh) == 0
            assert (mine == ref).all(), (iw, ih, ow, oh, kind)


def test_rgb_to_yuvj420p_matches_reference_build(orc):
    """row a17: the oracle's rgb24_to_yuvj420p against the reference's own (a static function of imgconvert.c, reached by
    compiling that file into the harness): random, extreme and gradient content, several even sizes, padded source rows"""
    R = _need_avcref(orc)
    L = orc.lib()
    rng = np.random.default_rng(12)
    for w, h in ((160, 120), (320, 240), (2, 2), (18, 6), (130, 98)):
        for kind in range(4):
            stride = w * 3 + (0 if kind % 2 == 0 else 7)
            src = np.zeros((h, stride), np.uint8)
            if kind == 0:
                src[:, : w * 3] = rng.integers(0, 256, (h, w * 3), dtype=np.uint8)
            elif kind == 1:
                src[:] = 255
            elif kind == 2:
                src[:, : w * 3] = (np.add.outer(np.arange(h) * 7, np.arange(w * 3) * 3) & 255).astype(np.uint8)
            else:
                src[:, : w * 3] = rng.integers(0, 2, (h, w * 3), dtype=np.uint8) * 255
            got = [np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
            want = [np.zeros_like(a) for a in got]
            L.amvo_rgb24_to_yuvj420p(src.ctypes.data, stride, w, h, 0, *[a.ctypes.data for a in got])
            R.amvref_rgb24_to_yuvj420p(src.ctypes.data, stride, w, h, *[a.ctypes.data for a in want])
            for a, b in zip(got, want):
                assert (a == b).all(), (w, h, kind)


def _reference_outputs():
    return json.load(open(os.path.join(GOLDEN, "reference_outputs.json")))


def test_oracle_matches_reference_produced_outputs(orc, amv1):
    """tests/golden/reference_outputs.json: hashes of what the REAL reference code produced (amvlib's AmvJpegDecode, the
    patched FFmpeg's amv decoder, its adpcm_ima_amv encoder plain and -trellis 3) on inputs this repository can rebuild
    from its seed -- the pin for the four slices DESIGN.md section 2 could not build here (amvlib dequant / IDCT / colour /
    StoreBuffer at the headline geometries, the FFmpeg-compat decode as a whole frame, the ADPCM-AMV encoder, its trellis)"""
    ref = _reference_outputs()
    seed, basis = ref["seed"], int(ref["fnv_basis"], 16)
    assert seed == SEED and basis == orc.FNV_BASIS
    step = ref["amvlib_decode"]["frame_step"]
    for c in ref["amvlib_decode"]["cases"]:
        w, h, hh = c["w"], c["h"], basis
        for t in range(c["n"]):
            out, st, _ = orc.decode_frame(orc.encode_frame(orc.synth_frame(seed, step * t, w, h), w, h), w, h)
            assert st == 0
            hh = orc.fnv1a64(hh, out)
        assert "%016x" % hh == c["fnv"], (w, h)
    ff = ref["ffmpeg_decode"]
    hh = basis
    for chunk in amv1["video"]:
        out, st, _ = orc.decode_frame_ffmpeg(chunk, 128, 96)
        assert st == 0
        hh = orc.fnv1a64(hh, out)
    assert "%016x" % hh == ff["amv1_all_252_frames"]
    s = ff["synth_160x120"]
    hh = basis
    for t in range(s["n"]):
        chunk = orc.encode_frame(orc.synth_frame(seed, s["frame_step"] * t, s["w"], s["h"]), s["w"], s["h"])
        hh = orc.fnv1a64(hh, orc.decode_frame_ffmpeg(chunk, s["w"], s["h"])[0])
    assert "%016x" % hh == s["fnv"]
    a = ref["adpcm_ima_amv_encode"]
    fs, k = a["frame_size"], a["chunks"]
    pcm = orc.synth_audio(seed, 0, k * fs)
    for key in ("plain", "trellis3"):
        idx, hh = 0, basis
        for i in range(k):
            x = pcm[i * fs: (i + 1) * fs]
            chunk, idx = orc.adpcm_encode_chunk(x, idx) if key == "plain" else orc.adpcm_encode_chunk_trellis(x, idx, a[key]["trellis"])
            hh = orc.fnv1a64(hh, np.frombuffer(chunk, np.uint8))
        assert ("%016x" % hh, idx) == (a[key]["fnv"], a[key]["end_index"]), key


def test_plane_encoders_are_the_rgb_encoder_behind_its_conversion(orc):
    """The oracle's plane-level encoders (round 4: what the YUVJ420P / YUVJ422P entries of the product are compared with):
    amvo_encode_frame_yuv420 on the oracle's own rgb24_to_yuvj420p planes -- itself pinned to the reference's imgconvert.c --
    gives amvo_encode_frame's chunk byte for byte, with padded plane rows; amvo_yuv422_to_420 is the stated rule
    ((a + b + 1) >> 1 per sample over row pairs) on exhaustive pairs; the 4:2:2 encoder is that rule in front of the
    4:2:0 one."""
    L = orc.lib()
    rng = np.random.default_rng(7)
    for w, h in ((160, 120), (130, 98), (16, 16), (336, 32)):
        cw, ch = w // 2, h // 2
        src = orc.synth_frame(SEED, 9, w, h)
        y, cb, cr = np.zeros((h, w + 8), np.uint8), np.zeros((ch, cw + 24), np.uint8), np.zeros((ch, cw + 24), np.uint8)
        yt, bt, rt = np.zeros((h, w), np.uint8), np.zeros((ch, cw), np.uint8), np.zeros((ch, cw), np.uint8)
        L.amvo_rgb24_to_yuvj420p(src.ctypes.data, w * 3, w, h, 0, yt.ctypes.data, bt.ctypes.data, rt.ctypes.data)
        y[:, :w], cb[:, :cw], cr[:, :cw] = yt, bt, rt
        y[:, w:], cb[:, cw:], cr[:, cw:] = 0xEE, 0xEE, 0xEE         # the padding is never read
        assert orc.encode_frame_yuv(y, cb, cr, w, h) == orc.encode_frame(src, w, h), (w, h)
        c2 = rng.integers(0, 256, (2, h, cw + 5), dtype=np.uint8)
        avg = ((c2[:, 0::2].astype(np.uint16) + c2[:, 1::2] + 1) >> 1).astype(np.uint8)
        assert orc.encode_frame_yuv(y, c2[0], c2[1], w, h) == orc.encode_frame_yuv(y, avg[0], avg[1], w, h), (w, h)
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    pairs = np.ascontiguousarray(np.stack([a.ravel(), b.ravel()]))      # two rows of 65 536 samples: every (a, b)
    out = np.zeros((1, 65536), np.uint8)
    L.amvo_yuv422_to_420(pairs.ctypes.data, 65536, 65536, 2, out.ctypes.data, 65536)
    assert (out[0] == ((a.ravel().astype(np.uint16) + b.ravel() + 1) >> 1)).all()


def test_reference_stereo_fixture_is_what_the_reference_build_makes(orc):
    """tests/golden/ref_adpcm_stereo.json (what the GPU test compares AdpcmImaDecodeFrame's stereo path with, so that the
    compiled reference object need not be loaded on the GPU box) is regenerated here from oracle/_ref and must equal the
    committed file; skipped where the reference tree is not at hand."""
    import importlib.util
    if orc.ref() is None:
        pytest.skip("oracle/_ref is not built here")
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("make_ref_golden", os.path.join(ROOT, "tests", "golden", "make_ref_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.compute() == json.load(open(os.path.join(ROOT, "tests", "golden", "ref_adpcm_stereo.json")))
