"""ctypes binding of the CPU oracle (oracle/libamvoracle.so) and, when present, of the reference
build oracle/_ref/libamvref.so.  TEST INFRASTRUCTURE ONLY: imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product package.
"""
import ctypes
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libamvoracle.so")
REF = os.path.join(HERE, "_ref", "libamvref.so")
AVCREF = os.path.join(HERE, "_ref", "libavcref.so")

ST_FORMAT, ST_OVERRUN, ST_TRUNCATED = 1, 2, 4
FLAG_ZIGZAG_FIXED = 1
FNV_BASIS = 0xCBF29CE484222325
# the seed the survey's harness actually used for its chained FNV-1a-64 (SURVEY.md section 8c:
# "seed 1469598103934665603"; it is the standard basis with its last decimal digit missing)
SURVEY_FNV_SEED = 1469598103934665603

_vp, _u32, _u64, _int = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int


def _stale(target, sources):
    return not os.path.exists(target) or os.path.getmtime(target) < max(os.path.getmtime(os.path.join(HERE, f)) for f in sources)


def build(force=False):
    """make -C oracle (the restatement always; _ref only where /root/reference exists).  Several processes may come here
    at once (the ranks of a multi-GPU bench run each load the checker): the make runs under a file lock, and whoever
    waited for it looks again before running it a second time."""
    import fcntl
    want_lib = force or _stale(LIB, ("amv_oracle.c", "amv_oracle.h", "Makefile"))
    want_ref = os.path.isdir("/root/reference") and (force or not os.path.exists(REF) or
                                                     _stale(AVCREF, ("ref_harness.c", "ref_harness_imgconvert.c", "Makefile")))
    if not (want_lib or want_ref):
        return
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or _stale(LIB, ("amv_oracle.c", "amv_oracle.h", "Makefile")):
            subprocess.run(["make", "-C", HERE, "-s", "libamvoracle.so"], check=True)
        if os.path.isdir("/root/reference") and (force or not os.path.exists(REF) or
                                                 _stale(AVCREF, ("ref_harness.c", "ref_harness_imgconvert.c", "Makefile"))):
            subprocess.run(["make", "-C", HERE, "-s", "ref"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        sig = {
            "amvo_stride": (_u32, [_u32]),
            "amvo_mcus_per_row": (_u32, [_u32]),
            "amvo_mcu_rows": (_u32, [_u32]),
            "amvo_decode_frame": (_int, [_vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
            "amvo_dequant_idct_block": (None, [_vp, _int, _u32, _vp]),
            "amvo_idct_block": (None, [_vp]),
            "amvo_yuv_to_bgr": (None, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp]),
            "amvo_q60_table": (None, [_int, _vp]),
            "amvo_simple_idct": (None, [_vp]),
            "amvo_simple_idct_put": (None, [_vp, _int, _vp]),
            "amvo_ffmpeg_dequant_block": (None, [_vp, _int, _vp]),
            "amvo_yuv420_frame_bytes": (_u32, [_u32, _u32]),
            "amvo_decode_frame_ffmpeg": (_int, [_vp, _u32, _u32, _u32, _vp, _vp, _vp]),
            "amvo_decode_frame_ffmpeg_keep": (_int, [_vp, _u32, _u32, _u32, _vp, _vp, _vp]),
            "amvo_entropy_blocks": (_u32, [_vp, _u32, _u32, _vp, _vp]),
            "amvo_build_resample_filter": (None, [_vp, _int, _int]),
            "amvo_img_resample_yuv420": (None, [_vp, _int, _int, _vp, _int, _int]),
            "amvo_adpcm_decode_chunk": (_int, [_vp, _u32, _vp, _vp]),
            "amvo_adpcm_encode_chunk": (_int, [_vp, _u32, ctypes.POINTER(_int), _vp]),
            "amvo_adpcm_encode_chunk_trellis": (_int, [_vp, _u32, ctypes.POINTER(_int), _int, _vp]),
            "amvo_adpcm_wav_encode_frame": (_int, [_vp, _int, _vp, _vp]),
            "amvo_adpcm_amv_pairs": (_u32, [_u32, _u32, ctypes.POINTER(_u32), ctypes.POINTER(_u64)]),
            "amvo_rgb24_to_yuvj420p": (None, [_vp, _u32, _u32, _u32, _int, _vp, _vp, _vp]),
            "amvo_fdct_islow": (None, [_vp]),
            "amvo_quantize_block": (None, [_vp, _int, _u32, _vp]),
            "amvo_encode_frame": (_int, [_vp, _u32, _u32, _u32, _int, _u32, _vp, _vp]),
            "amvo_encode_bound": (_u32, [_u32, _u32]),
            "amvo_encode_frame_yuv420": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp]),
            "amvo_encode_frame_yuv422": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp]),
            "amvo_yuv422_to_420": (None, [_vp, _u32, _u32, _u32, _vp, _u32]),
            "amvo_synth_frame": (None, [_u32, _u32, _u32, _u32, _vp]),
            "amvo_synth_audio": (None, [_u32, _u64, _u32, _vp]),
            "amvo_fnv1a64": (_u64, [_u64, _vp, ctypes.c_size_t]),
            "amvo_psnr": (ctypes.c_double, [_vp, _vp, ctypes.c_size_t]),
            "amvo_decode_batch": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _int]),
            "amvo_synth_encode_batch": (_int, [_u32, _u32, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp, _int]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


_ref = None


def ref():
    """oracle/_ref/libamvref.so (reference AdpcmIma.c + jfdctint.c) or None when not built"""
    global _ref
    if _ref is None and os.path.exists(REF):
        R = ctypes.CDLL(REF)
        R.AdpcmImaDecodeFrame.restype = _int
        R.AdpcmImaDecodeFrame.argtypes = [_vp, _vp, _vp, _vp, _int]
        R.AdpcmImaEncodeFrame.restype = _int
        R.AdpcmImaEncodeFrame.argtypes = [_vp, _int, _int, _vp, _int, _vp]
        R.ff_jpeg_fdct_islow.restype = None
        R.ff_jpeg_fdct_islow.argtypes = [_vp]
        _ref = R
    return _ref


_avcref = None


def avcref():
    """oracle/_ref/libavcref.so: the reference's own mjpegenc.c / mjpeg.c / simple_idct.c / sp5x.h behind
    ref_harness.c, or None when not built"""
    global _avcref
    if _avcref is None and os.path.exists(AVCREF):
        R = ctypes.CDLL(AVCREF)
        R.amvref_mjpeg_encode_scan.restype = _int
        R.amvref_mjpeg_encode_scan.argtypes = [_vp, _int, _vp, _int]
        R.amvref_simple_idct.restype = None
        R.amvref_simple_idct.argtypes = [_vp, _int]
        R.amvref_sp5x_quant.restype = None
        R.amvref_sp5x_quant.argtypes = [_int, _vp]
        R.amvref_sp5x_segment.restype = _int
        R.amvref_sp5x_segment.argtypes = [_int, _vp, _int]
        R.amvref_mjpeg_huffman_spec.restype = _int
        R.amvref_mjpeg_huffman_spec.argtypes = [_int, _vp, _vp]
        R.amvref_rgb24_to_yuvj420p.restype = None
        R.amvref_rgb24_to_yuvj420p.argtypes = [_vp, _int, _int, _int, _vp, _vp, _vp]
        R.amvref_img_resample.restype = _int
        R.amvref_img_resample.argtypes = [_vp, _int, _int, _vp, _int, _int]
        R.amvref_mjpeg_huffman_codes.restype = None
        R.amvref_mjpeg_huffman_codes.argtypes = [_int, _vp, _vp]
        _avcref = R
    return _avcref


def ref_mjpeg_encode_scan(coef):
    """coef [nmcu*6, 64] int16 (zig-zag order, not predicted) -> the bytes the REFERENCE's entropy coder +
    picture trailer write (scan, padding, FF escaping, EOI)"""
    coef = np.ascontiguousarray(coef, np.int16)
    nm = coef.shape[0] // 6
    buf = np.zeros(nm * 6 * 64 * 8 + 64, np.uint8)
    n = avcref().amvref_mjpeg_encode_scan(coef.ctypes.data, nm, buf.ctypes.data, buf.size)
    if n < 0:
        raise RuntimeError("reference entropy coder failed")
    return bytes(buf[:n])


class RefADPCMChannelStatus(ctypes.Structure):  # reference AdpcmIma.h:11-18
    _fields_ = [("predictor", ctypes.c_int), ("step_index", ctypes.c_short), ("step", ctypes.c_int),
                ("prev_sample", ctypes.c_int)]


class RefADPCMContext(ctypes.Structure):  # reference AdpcmIma.h:20-25
    _fields_ = [("channel", ctypes.c_int), ("status", RefADPCMChannelStatus * 2), ("sample_buffer", ctypes.c_short * 32)]


# ---- convenience wrappers (numpy in / numpy out) -------------------------------------------

def stride(w):
    return lib().amvo_stride(w)


def nmcu(w, h):
    return lib().amvo_mcus_per_row(w) * lib().amvo_mcu_rows(h)


def decode_frame(chunk, w, h, flags=0, want_coef=False):
    """-> (bgr[h, stride] uint8, status, nmcu_ok[, coef[nmcu*6, 64] int16])"""
    L = lib()
    chunk = bytes(chunk)
    out = np.zeros((h, L.amvo_stride(w)), np.uint8)
    coef = np.zeros((nmcu(w, h) * 6, 64), np.int16) if want_coef else None
    ok, st = _u32(), _u32()
    L.amvo_decode_frame(chunk, len(chunk), w, h, flags, out.ctypes.data, coef.ctypes.data if want_coef else None,
                        ctypes.byref(ok), ctypes.byref(st))
    return (out, st.value, ok.value, coef) if want_coef else (out, st.value, ok.value)


def decode_frame_ffmpeg(chunk, w, h):
    """FFmpeg-compat mode -> (yuvj420p bytes [w*h + 2*cw*ch] uint8, status, nmcu_ok)"""
    L = lib()
    chunk = bytes(chunk)
    out = np.zeros(L.amvo_yuv420_frame_bytes(w, h), np.uint8)
    ok, st = _u32(), _u32()
    L.amvo_decode_frame_ffmpeg(chunk, len(chunk), w, h, out.ctypes.data, ctypes.byref(ok), ctypes.byref(st))
    return out, st.value, ok.value


def entropy_blocks(chunk, nblocks):
    """the entropy stage alone -> (coef [done, 64] int16 of the whole blocks before the first error, status)"""
    L = lib()
    chunk = bytes(chunk)
    coef = np.zeros((nblocks, 64), np.int16)
    st = _u32()
    done = L.amvo_entropy_blocks(chunk, len(chunk), nblocks, coef.ctypes.data, ctypes.byref(st))
    return coef[:done], st.value


def decode_frame_ffmpeg_keep(chunk, w, h, before):
    """AMVHIP_FLAG_FFMPEG_KEEP: `before` (a YUVJ420P frame buffer) with every whole block ahead of the chunk's first error
    put into it, nothing else touched -> (buffer, status, whole blocks decoded)"""
    L = lib()
    chunk = bytes(chunk)
    out = np.array(before, dtype=np.uint8, copy=True).reshape(-1)
    assert out.size == L.amvo_yuv420_frame_bytes(w, h)
    ok, st = _u32(), _u32()
    L.amvo_decode_frame_ffmpeg_keep(chunk, len(chunk), w, h, out.ctypes.data, ctypes.byref(ok), ctypes.byref(st))
    return out, st.value, ok.value


def yuv_planes(buf, w, h):
    """split a YUVJ420P frame buffer -> (Y [h, w], Cb [ch, cw], Cr [ch, cw])"""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    return (buf[: w * h].reshape(h, w), buf[w * h: w * h + cw * ch].reshape(ch, cw),
            buf[w * h + cw * ch: w * h + 2 * cw * ch].reshape(ch, cw))


def yuv420_bytes(w, h):
    return w * h + 2 * (w // 2) * (h // 2)


def img_resample_yuv420(frame, iw, ih, ow, oh):
    """tight YUV420P frame (uint8 [iw*ih + 2*(iw/2)*(ih/2)]) -> the same at ow x oh (reference img_resample)"""
    frame = np.ascontiguousarray(frame, np.uint8)
    out = np.zeros(yuv420_bytes(ow, oh), np.uint8)
    lib().amvo_img_resample_yuv420(frame.ctypes.data, iw, ih, out.ctypes.data, ow, oh)
    return out


def encode_frame(pix, w, h, bgr=False, qbias=0, want_coef=False):
    """pix: [h, w, 3] uint8 contiguous -> chunk bytes[, coef]"""
    L = lib()
    pix = np.ascontiguousarray(pix, np.uint8)
    buf = np.zeros(L.amvo_encode_bound(w, h), np.uint8)
    coef = np.zeros((nmcu(w, h) * 6, 64), np.int16) if want_coef else None
    n = L.amvo_encode_frame(pix.ctypes.data, w * 3, w, h, 1 if bgr else 0, qbias, buf.ctypes.data,
                            coef.ctypes.data if want_coef else None)
    if n < 0:
        raise ValueError("amvo_encode_frame rejected %dx%d" % (w, h))
    return (bytes(buf[:n]), coef) if want_coef else bytes(buf[:n])


def encode_frame_yuv(y, cb, cr, w, h, qbias=0):
    """planes as 2-D uint8 arrays (rows may be wider than the picture: the row pitch is the array's) -> chunk bytes.
    cb / cr with h/2 rows: YUVJ420P (amvo_encode_frame_yuv420); with h rows: YUVJ422P by the product's averaging rule
    (amvo_encode_frame_yuv422)"""
    L = lib()
    y, cb, cr = (np.ascontiguousarray(p, np.uint8) for p in (y, cb, cr))
    assert cb.shape == cr.shape and y.shape[0] == h and cb.shape[0] in (h // 2, h)
    buf = np.zeros(L.amvo_encode_bound(w, h), np.uint8)
    fn = L.amvo_encode_frame_yuv420 if cb.shape[0] == h // 2 else L.amvo_encode_frame_yuv422
    n = fn(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, y.shape[1], cb.shape[1], w, h, qbias, buf.ctypes.data)
    if n < 0:
        raise ValueError("encode_frame_yuv rejected %dx%d" % (w, h))
    return bytes(buf[:n])


def synth_frame(seed, t, w, h):
    rgb = np.zeros((h, w, 3), np.uint8)
    lib().amvo_synth_frame(seed, t, w, h, rgb.ctypes.data)
    return rgb


def synth_audio(seed, first, n):
    pcm = np.zeros(n, np.int16)
    lib().amvo_synth_audio(seed, first, n, pcm.ctypes.data)
    return pcm


def adpcm_decode_chunk(chunk):
    chunk = bytes(chunk)
    pcm = np.zeros(max(2 * (len(chunk) - 8), 0) + 8, np.int16)
    hdr = _u32()
    n = lib().amvo_adpcm_decode_chunk(chunk, len(chunk), pcm.ctypes.data, ctypes.byref(hdr))
    return (pcm[:n].copy() if n > 0 else pcm[:0]), hdr.value


def adpcm_encode_chunk(samples, step_index):
    samples = np.ascontiguousarray(samples, np.int16)
    out = np.zeros(8 + len(samples) // 2, np.uint8)
    si = _int(step_index)
    n = lib().amvo_adpcm_encode_chunk(samples.ctypes.data, len(samples), ctypes.byref(si), out.ctypes.data)
    return bytes(out[:n]), si.value


def adpcm_encode_chunk_trellis(samples, step_index, trellis):
    """-> (chunk bytes, step index after) with the reference's trellis search, frontier 2**trellis"""
    samples = np.ascontiguousarray(samples, np.int16)
    out = np.zeros(8 + samples.size // 2 + 8, np.uint8)
    idx = ctypes.c_int(step_index)
    n = lib().amvo_adpcm_encode_chunk_trellis(samples.ctypes.data, samples.size, ctypes.byref(idx), trellis, out.ctypes.data)
    if n < 0:
        raise ValueError("trellis must be 1..5")
    return bytes(out[:n]), idx.value


def fnv1a64(h, arr):
    arr = np.ascontiguousarray(arr)
    return lib().amvo_fnv1a64(h, arr.ctypes.data, arr.nbytes)


def psnr(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().amvo_psnr(a.ctypes.data, b.ctypes.data, a.size)


def synth_stream(seed, first, n, w, h, qbias=0, threads=1):
    """seeded synthetic clip encoded by the oracle's encoder -> (blob uint8, offs uint64, lens uint32)"""
    L = lib()
    cap = int(L.amvo_encode_bound(w, h)) * n
    cap = min(cap, max(1 << 20, n * w * h))  # real chunks are ~0.2 B/pixel
    blob = np.zeros(cap, np.uint8)
    offs = np.zeros(n, np.uint64)
    lens = np.zeros(n, np.uint32)
    rc = L.amvo_synth_encode_batch(seed, first, n, w, h, qbias, blob.ctypes.data, cap, offs.ctypes.data,
                                   lens.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("synthetic stream did not fit its buffer")
    total = int(offs[-1] + lens[-1]) if n else 0
    return blob[:total + 16].copy(), offs, lens


def decode_batch(blob, offs, lens, w, h, flags=0, threads=1):
    L = lib()
    n = len(lens)
    out = np.zeros((n, h, L.amvo_stride(w)), np.uint8)
    st = np.zeros(n, np.int32)
    L.amvo_decode_batch(blob.ctypes.data, offs.ctypes.data, lens.ctypes.data, n, w, h, flags, out.ctypes.data,
                        st.ctypes.data, threads)
    return out, st


def parse_amv(data):
    """walk an AMV file: -> (info dict, [video chunks], [audio chunks])  (reference AMVDec.c:150-238)"""
    info = {"us_per_frame": struct.unpack_from("<I", data, 32)[0], "width": struct.unpack_from("<I", data, 64)[0],
            "height": struct.unpack_from("<I", data, 68)[0], "fps": struct.unpack_from("<I", data, 72)[0],
            "sample_rate": struct.unpack_from("<I", data, 288)[0]}
    p = data.find(b"movi") + 4
    vids, auds = [], []
    while p + 8 <= len(data) and data[p:p + 4] != b"AMV_":
        n = struct.unpack_from("<I", data, p + 4)[0]
        (vids if data[p:p + 4] == b"00dc" else auds).append(data[p + 8:p + 8 + n])
        p += 8 + n
    return info, vids, auds
