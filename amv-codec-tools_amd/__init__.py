"""amv-codec-tools_amd -- MI355X-native AMV codec hot path (video decode/encode, IMA ADPCM).

The product is libamvhip.so (HIP kernels for gfx950 behind a C ABI, include/amvhip.h).  This
module is the thin ctypes binding tests and bench.py use; it contains no codec arithmetic and
no CPU fallback: importing it without the built library raises.

The directory name is not a Python identifier; load it with `__graft_entry__.load_package()`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMVHIP_LIB") or os.path.join(_HERE, "libamvhip.so")   # override: kernel experiments

OK, ERR_ARG, ERR_DEVICE, ERR_NOMEM, ERR_SPACE = 0, -1, -2, -3, -4
ST_FORMAT, ST_OVERRUN, ST_TRUNCATED = 1, 2, 4
FLAG_ZIGZAG_FIXED = 1
FLAG_FFMPEG = 2
FLAG_FFMPEG_KEEP = 4
QBIAS_AMV, QBIAS_MJPEG = 0, 128
K_HUFFMAN, K_RECON, K_FDCT, K_PACK, K_ADPCM_DEC, K_ADPCM_ENC, K_SYNTH, K_HUFFMAN_SERIAL, K_UNSTUFF, K_PACK_SERIAL, K_COMPACT = range(11)
ENTROPY_AUTO, ENTROPY_SERIAL = 0, 1

_vp, _u8p = ctypes.c_void_p, ctypes.c_void_p
_u32, _u64, _i32, _int = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int


class AMVInfo(ctypes.Structure):  # include/amvhip.h, reference AMVDec.h:29-47
    _fields_ = [("dwMicroSecPerFrame", ctypes.c_uint), ("dwWidth", ctypes.c_uint), ("dwHeight", ctypes.c_uint),
                ("dwSpeed", ctypes.c_uint), ("dwTimeSec", ctypes.c_uint), ("dwTimeMin", ctypes.c_uint),
                ("dwTimeHour", ctypes.c_uint), ("wFormatTag", ctypes.c_ushort), ("nChannels", ctypes.c_ushort),
                ("nSamplesPerSec", ctypes.c_uint), ("nAvgBytesPerSec", ctypes.c_uint),
                ("nBlockAlign", ctypes.c_ushort), ("wBitsPerSample", ctypes.c_ushort),
                ("cbSize", ctypes.c_ushort), ("wSamplesPerBlock", ctypes.c_ushort)]


class FRAMEBUFF(ctypes.Structure):  # AMVDec.h:50-57
    _fields_ = [("videobuff", ctypes.POINTER(ctypes.c_ubyte)), ("audiobuff", ctypes.POINTER(ctypes.c_ubyte)),
                ("videobufflen", ctypes.c_uint), ("audiobufflen", ctypes.c_uint), ("framenum", ctypes.c_int)]


class VIDEOBUFF(ctypes.Structure):  # AMVDec.h:59-63
    _fields_ = [("fbmpdat", ctypes.POINTER(ctypes.c_ubyte)), ("len", ctypes.c_uint)]


class AUDIOBUFF(ctypes.Structure):  # AMVDec.h:67-71
    _fields_ = [("audiodata", ctypes.POINTER(ctypes.c_short)), ("len", ctypes.c_uint)]


class AMVDecoder(ctypes.Structure):  # AMVDec.h:74-91
    _fields_ = [("amvfilename", ctypes.c_char_p), ("opened", ctypes.c_int), ("dataseekpos", ctypes.c_long),
                ("fileseekpos", ctypes.c_long), ("amvinfo", AMVInfo), ("currentframe", ctypes.c_uint),
                ("totalframe", ctypes.c_uint), ("framebuf", FRAMEBUFF), ("videobuf", VIDEOBUFF),
                ("audiobuf", AUDIOBUFF)]


class ADPCMChannelStatus(ctypes.Structure):  # AdpcmIma.h:11-18
    _fields_ = [("predictor", ctypes.c_int), ("step_index", ctypes.c_short), ("step", ctypes.c_int),
                ("prev_sample", ctypes.c_int)]


class ADPCMContext(ctypes.Structure):  # AdpcmIma.h:20-25
    _fields_ = [("channel", ctypes.c_int), ("status", ADPCMChannelStatus * 2), ("sample_buffer", ctypes.c_short * 32)]


# every symbol include/amvhip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "PrepareForVideoDecode": (None, [ctypes.POINTER(AMVInfo)]),
    "AmvJpegDecode": (_int, [ctypes.POINTER(AMVInfo), ctypes.POINTER(FRAMEBUFF), ctypes.POINTER(VIDEOBUFF)]),
    "AdpcmImaDecodeFrame": (_int, [ctypes.POINTER(ADPCMContext), _vp, ctypes.POINTER(_int), _vp, _int]),
    "AdpcmImaEncodeFrame": (_int, [ctypes.POINTER(ADPCMContext), _int, _int, _vp, _int, _vp]),
    "AmvOpen": (ctypes.POINTER(AMVDecoder), [ctypes.c_char_p]),
    "AmvClose": (None, [ctypes.POINTER(AMVDecoder)]),
    "AmvReadNextFrame": (_int, [ctypes.POINTER(AMVDecoder)]),
    "AmvRewindFrameStart": (_int, [ctypes.POINTER(AMVDecoder)]),
    "AmvVideoDecode": (_int, [ctypes.POINTER(AMVDecoder)]),
    "AmvAudioDecode": (_int, [ctypes.POINTER(AMVDecoder)]),
    "AmvJpegPutHeader": (None, [_vp, ctypes.c_ushort, ctypes.c_ushort]),
    "AmvCreateJpegFileFromFrameBuffer": (_int, [ctypes.POINTER(AMVDecoder), ctypes.c_char_p]),
    "AmvCreateJpegFileFromBuffer": (_int, [ctypes.POINTER(AMVInfo), ctypes.POINTER(FRAMEBUFF), ctypes.c_char_p]),
    "AmvConvertJpegFileToBmpFile": (_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "ConvertJpegFileToBmpFile": (_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "AmvCreateWavFileFromAmvFile": (_int, [ctypes.POINTER(AMVDecoder), _int, ctypes.c_char_p]),
    "decode_amv_frame": (_int, [_vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, _vp]),
    "encode_amv_frame": (_int, [_vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, _int, _vp, ctypes.c_uint]),
    "amvhip_create": (_int, [ctypes.POINTER(_vp), _int]),
    "amvhip_destroy": (None, [_vp]),
    "amvhip_last_error": (ctypes.c_char_p, [_vp]),
    "amvhip_device": (_int, [_vp]),
    "amvhip_stride": (_u32, [_u32]),
    "amvhip_frame_bytes": (_u64, [_u32, _u32]),
    "amvhip_yuv420_frame_bytes": (_u64, [_u32, _u32]),
    "amvhip_encode_bound": (_u32, [_u32, _u32]),
    "amvhip_jpeg_header": (_u32, [ctypes.c_ushort, ctypes.c_ushort, _vp, _u32]),
    "amvhip_decode_batch_dev": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "amvhip_decode_submit_dev": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "amvhip_decode_collect_dev": (_int, [_vp, _vp]),
    "amvhip_decode_workspace_per_frame": (ctypes.c_double, [_vp]),
    "amvhip_decode_batch": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "amvhip_decode_batch_async": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "amvhip_sync": (_int, [_vp]),
    "amvhip_host_alloc": (_int, [_vp, ctypes.POINTER(_vp), ctypes.c_size_t]),
    "amvhip_host_free": (None, [_vp, _vp]),
    "amvhip_huffman_decode_dev": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "amvhip_reconstruct_dev": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "amvhip_encode_batch_dev": (_int, [_vp, _vp, _u32, _int, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp, _vp]),
    "amvhip_encode_batch": (_int, [_vp, _vp, _u32, _int, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp]),
    "amvhip_encode_yuv420_batch_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp, _vp]),
    "amvhip_encode_yuv422_batch_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp, _vp]),
    "amvhip_encode_yuv420_batch": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp]),
    "amvhip_encode_yuv422_batch": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp]),
    "amvhip_resample_yuv420_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _u32, _u32, _u32, _vp]),
    "amvhip_encode_yuv420_scaled_batch_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u64, _u64, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _u64, _vp, _vp, _vp]),
    "amvhip_encode_coefs_dev": (_int, [_vp, _vp, _u32, _int, _u32, _u32, _u32, _u32, _vp, _vp]),
    "amvhip_adpcm_decode_batch_dev": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "amvhip_adpcm_encode_batch_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "amvhip_adpcm_decode_batch": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _vp, _u64, _vp, _vp]),
    "amvhip_adpcm_decode_batch_async": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _vp, _u64, _vp, _vp]),
    "amvhip_adpcm_encode_batch": (_int, [_vp, _vp, _u64, _vp, _vp, _u32, _vp, _vp, _u64, _vp]),
    "amvhip_adpcm_encode_frame": (_int, [_vp, _vp, _u32, ctypes.POINTER(_i32), _vp, _u32]),
    "amvhip_adpcm_encode_trellis_batch_dev": (_int, [_vp, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _vp, _vp, _vp]),
    "amvhip_adpcm_encode_frame_trellis": (_int, [_vp, _vp, _u32, ctypes.POINTER(_i32), _u32, _vp, _u32]),
    "amvhip_amv_audio_pairs": (_u32, [_u32, _u32, ctypes.POINTER(_u32), ctypes.POINTER(_u64)]),
    "amvhip_amv_audio_frame_size": (_u32, [_u32, _u32, _u32]),
    "amvhip_adpcm_wav_encode_frame": (_int, [_vp, _vp, _int, _vp, _vp, _int]),
    "amvhip_synth_frames_dev": (_int, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "amvhip_synth_audio_dev": (_int, [_vp, _u32, _u64, _u64, _vp, _vp]),
    "amvhip_set_entropy_mode": (_int, [_vp, _int]),
    "amvhip_entropy_stats": (_int, [_vp, _int, _vp]),
    "amvhip_entropy_trace": (_int, [_vp, _vp, _u32]),
    "amvhip_decode_split_stats": (_int, [_vp, _vp]),
    "amvhip_adpcm_chain_stats": (_int, [_vp, _vp]),
    "amvhip_adpcm_quotient_table": (None, [_vp]),
    "amvhip_prof_enable": (None, [_vp, _int]),
    "amvhip_prof_reset": (None, [_vp]),
    "amvhip_prof_read": (_int, [_vp, _int, ctypes.POINTER(_u64), ctypes.POINTER(ctypes.c_double)]),
    "amvhip_kernel_name": (ctypes.c_char_p, [_int]),
    # container writer (host C)
    "amvhip_mux_open": (_vp, [ctypes.c_char_p, _u32, _u32, _u32, _u32, _u32, _u32]),
    "amvhip_mux_write_frame": (_int, [_vp, _u8p, _u32, _u8p, _u32]),
    "amvhip_mux_close": (_int, [_vp]),
}

_lib = None


def load_library():
    """dlopen libamvhip.so and type every entry point.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libamvhip.so is not built (%s). Run `python amv-codec-tools_amd/build.py`; there is no CPU fallback."
                % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64 and the tensors this binding is
        # handed live in it.  If libamvhip.so were loaded first it would bring in the system runtime, and
        # the second runtime to initialise finds no usable device.  So torch (when present) goes first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class AmvHipError(RuntimeError):
    pass


def _ptr(x):
    """device/host address of a torch tensor, numpy array, bytes object or int; None -> NULL.  The caller keeps x alive
    for the duration of the call; other buffer types (bytearray, memoryview ...) are refused: wrap them in numpy
    (np.frombuffer) so that the address is that of the caller's own memory, never of a temporary copy."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        return x.ctypes.data
    if isinstance(x, bytes):      # immutable, owned by the caller: input arguments only
        return ctypes.cast(ctypes.c_char_p(x), ctypes.c_void_p).value
    raise TypeError("pass a torch tensor, numpy array, bytes or int address, not %s" % type(x).__name__)


class Context:
    """One amvhip_ctx.  Methods mirror the batch entry points of include/amvhip.h one to one;
    arguments are torch tensors / numpy arrays (their addresses are passed through)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.amvhip_create(ctypes.byref(h), int(device))
        if rc != OK:
            raise AmvHipError("amvhip_create(device=%d) failed with %d: no usable HIP device" % (device, rc))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.amvhip_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc, what):
        if rc < 0:
            raise AmvHipError("%s failed (%d): %s" % (what, rc, self.lib.amvhip_last_error(self.h).decode()))
        return rc

    # geometry
    def stride(self, w):
        return self.lib.amvhip_stride(w)

    def frame_bytes(self, w, h):
        return self.lib.amvhip_frame_bytes(w, h)

    def yuv420_frame_bytes(self, w, h):
        return self.lib.amvhip_yuv420_frame_bytes(w, h)

    def encode_bound(self, w, h):
        return self.lib.amvhip_encode_bound(w, h)

    # device-resident
    def decode_batch_dev(self, blob, blob_bytes, offs, lens, n, w, h, flags, out, status, stream=None):
        return self._check(self.lib.amvhip_decode_batch_dev(self.h, _ptr(blob), blob_bytes, _ptr(offs), _ptr(lens), n,
                                                            w, h, flags, _ptr(out), _ptr(status), stream), "decode_batch_dev")

    def decode_submit_dev(self, blob, blob_bytes, offs, lens, n, w, h, flags, out, status, stream=None):
        """queue a batch on the context's own streams; `stream` marks where the inputs are ready and does not wait"""
        return self._check(self.lib.amvhip_decode_submit_dev(self.h, _ptr(blob), blob_bytes, _ptr(offs), _ptr(lens), n,
                                                             w, h, flags, _ptr(out), _ptr(status), stream), "decode_submit_dev")

    def decode_collect_dev(self, stream=None):
        """`stream` waits for the oldest submitted batch"""
        return self._check(self.lib.amvhip_decode_collect_dev(self.h, stream), "decode_collect_dev")

    def decode_workspace_per_frame(self):
        return self.lib.amvhip_decode_workspace_per_frame(self.h)

    def huffman_decode_dev(self, blob, blob_bytes, offs, lens, n, w, h, coef, status, nmcu_ok, stream=None):
        return self._check(self.lib.amvhip_huffman_decode_dev(self.h, _ptr(blob), blob_bytes, _ptr(offs), _ptr(lens), n,
                                                              w, h, _ptr(coef), _ptr(status), _ptr(nmcu_ok), stream),
                           "huffman_decode_dev")

    def reconstruct_dev(self, coef, nmcu_ok, n, w, h, flags, out, stream=None):
        return self._check(self.lib.amvhip_reconstruct_dev(self.h, _ptr(coef), _ptr(nmcu_ok), n, w, h, flags,
                                                           _ptr(out), stream), "reconstruct_dev")

    def encode_batch_dev(self, pix, pix_stride, is_bgr, n, w, h, qbias, blob, blob_cap, offs, lens, stream=None):
        return self._check(self.lib.amvhip_encode_batch_dev(self.h, _ptr(pix), pix_stride, is_bgr, n, w, h, qbias,
                                                            _ptr(blob), blob_cap, _ptr(offs), _ptr(lens), stream),
                           "encode_batch_dev")

    def encode_yuv420_batch_dev(self, y, cb, cr, y_stride, c_stride, y_frame, c_frame, n, w, h, qbias, blob, blob_cap, offs,
                                lens, stream=None):
        return self._check(self.lib.amvhip_encode_yuv420_batch_dev(self.h, _ptr(y), _ptr(cb), _ptr(cr), y_stride, c_stride,
                                                                   y_frame, c_frame, n, w, h, qbias, _ptr(blob), blob_cap,
                                                                   _ptr(offs), _ptr(lens), stream), "encode_yuv420_batch_dev")

    def encode_yuv422_batch_dev(self, y, cb, cr, y_stride, c_stride, y_frame, c_frame, n, w, h, qbias, blob, blob_cap, offs,
                                lens, stream=None):
        return self._check(self.lib.amvhip_encode_yuv422_batch_dev(self.h, _ptr(y), _ptr(cb), _ptr(cr), y_stride, c_stride,
                                                                   y_frame, c_frame, n, w, h, qbias, _ptr(blob), blob_cap,
                                                                   _ptr(offs), _ptr(lens), stream), "encode_yuv422_batch_dev")

    def encode_yuv420_batch(self, y, cb, cr, y_stride, c_stride, y_frame, c_frame, n, w, h, qbias, blob, blob_cap, offs, lens):
        return self._check(self.lib.amvhip_encode_yuv420_batch(self.h, _ptr(y), _ptr(cb), _ptr(cr), y_stride, c_stride,
                                                               y_frame, c_frame, n, w, h, qbias, _ptr(blob), blob_cap,
                                                               _ptr(offs), _ptr(lens)), "encode_yuv420_batch")

    def encode_yuv422_batch(self, y, cb, cr, y_stride, c_stride, y_frame, c_frame, n, w, h, qbias, blob, blob_cap, offs, lens):
        return self._check(self.lib.amvhip_encode_yuv422_batch(self.h, _ptr(y), _ptr(cb), _ptr(cr), y_stride, c_stride,
                                                               y_frame, c_frame, n, w, h, qbias, _ptr(blob), blob_cap,
                                                               _ptr(offs), _ptr(lens)), "encode_yuv422_batch")

    def encode_coefs_dev(self, pix, pix_stride, is_bgr, n, w, h, qbias, coef, stream=None):
        return self._check(self.lib.amvhip_encode_coefs_dev(self.h, _ptr(pix), pix_stride, is_bgr, n, w, h, qbias,
                                                            _ptr(coef), stream), "encode_coefs_dev")

    def adpcm_decode_batch_dev(self, blob, blob_bytes, offs, lens, n, pcm, pcm_offs, final_state=None, stream=None):
        return self._check(self.lib.amvhip_adpcm_decode_batch_dev(self.h, _ptr(blob), blob_bytes, _ptr(offs), _ptr(lens),
                                                                  n, _ptr(pcm), _ptr(pcm_offs), _ptr(final_state), stream),
                           "adpcm_decode_batch_dev")

    def adpcm_encode_batch_dev(self, pcm, pcm_offs, nsamp, n, step_in, blob, offs, stream=None):
        return self._check(self.lib.amvhip_adpcm_encode_batch_dev(self.h, _ptr(pcm), _ptr(pcm_offs), _ptr(nsamp), n,
                                                                  _ptr(step_in), _ptr(blob), _ptr(offs), stream),
                           "adpcm_encode_batch_dev")

    def synth_frames_dev(self, seed, first, n, w, h, rgb, stream=None):
        return self._check(self.lib.amvhip_synth_frames_dev(self.h, seed, first, n, w, h, _ptr(rgb), stream), "synth_frames_dev")

    def synth_audio_dev(self, seed, first, n, pcm, stream=None):
        return self._check(self.lib.amvhip_synth_audio_dev(self.h, seed, first, n, _ptr(pcm), stream), "synth_audio_dev")

    # host buffers
    def decode_batch(self, blob, blob_bytes, offs, lens, n, w, h, flags, out, status):
        return self._check(self.lib.amvhip_decode_batch(self.h, _ptr(blob), blob_bytes, _ptr(offs), _ptr(lens), n, w, h,
                                                        flags, _ptr(out), _ptr(status)), "decode_batch")

    def encode_batch(self, pix, pix_stride, is_bgr, n, w, h, qbias, blob, blob_cap, offs, lens):
        return self._check(self.lib.amvhip_encode_batch(self.h, _ptr(pix), pix_stride, is_bgr, n, w, h, qbias, _ptr(blob),
                                                        blob_cap, _ptr(offs), _ptr(lens)), "encode_batch")

    def adpcm_decode_batch(self, blob, blob_bytes, offs, lens, n, pcm, pcm_samples, pcm_offs, final_state=None):
        return self._check(self.lib.amvhip_adpcm_decode_batch(self.h, _ptr(blob), blob_bytes, _ptr(offs), _ptr(lens), n,
                                                              _ptr(pcm), pcm_samples, _ptr(pcm_offs), _ptr(final_state)),
                           "adpcm_decode_batch")

    def adpcm_encode_batch(self, pcm, pcm_samples, pcm_offs, nsamp, n, step_in, blob, blob_bytes, offs):
        return self._check(self.lib.amvhip_adpcm_encode_batch(self.h, _ptr(pcm), pcm_samples, _ptr(pcm_offs), _ptr(nsamp),
                                                              n, _ptr(step_in), _ptr(blob), blob_bytes, _ptr(offs)),
                           "adpcm_encode_batch")

    def set_entropy_mode(self, mode):
        return self._check(self.lib.amvhip_set_entropy_mode(self.h, mode), "set_entropy_mode")

    def entropy_stats(self, enable=True):
        out = (ctypes.c_uint64 * 10)()
        self._check(self.lib.amvhip_entropy_stats(self.h, 1 if enable else 0, out), "entropy_stats")
        waves = max(out[9], 1)
        return {"frames": out[0], "rounds": out[1], "max_rounds": out[2], "handed_to_serial": out[3], "waves": out[9],
                "clocks_per_wave": {"zero": out[4] / waves, "first_walk": out[5] / waves, "sync_rounds": out[6] / waves,
                                    "write": out[7] / waves, "dc": out[8] / waves}}

    def entropy_trace(self, tasks):
        """per task (wave) of the last several-lanes-per-frame launch with gathering on: [tasks, 8] uint64 (include/amvhip.h)"""
        import numpy as np
        out = np.zeros((tasks, 8), np.uint64)
        got = self.lib.amvhip_entropy_trace(self.h, out.ctypes.data, tasks)
        if got < 0:
            self._check(got, "entropy_trace")
        return out[:got]

    def decode_split_stats(self):
        """last decode call: {"heavy": frames that got several entropy lanes, "light": frames that got one} (0, 0: no split)"""
        out = (ctypes.c_uint32 * 2)()
        self._check(self.lib.amvhip_decode_split_stats(self.h, out), "decode_split_stats")
        return {"heavy": int(out[0]), "light": int(out[1])}

    def adpcm_chain_stats(self):
        """last chained ADPCM encode: {"exhaustive": bool, "recoded": [chunks coded again in sweep 1, 2, ...]}"""
        out = (ctypes.c_uint32 * 64)()
        self._check(self.lib.amvhip_adpcm_chain_stats(self.h, out), "adpcm_chain_stats")
        rec = list(out[1:63])
        while rec and rec[-1] == 0:
            rec.pop()
        return {"exhaustive": bool(out[0]), "recoded": rec}

    # timing
    def prof_enable(self, on=True):
        self.lib.amvhip_prof_enable(self.h, 1 if on else 0)

    def prof_reset(self):
        self.lib.amvhip_prof_reset(self.h)

    def prof_read(self, kernel):
        n, ms = ctypes.c_uint64(), ctypes.c_double()
        self.lib.amvhip_prof_read(self.h, kernel, ctypes.byref(n), ctypes.byref(ms))
        return n.value, ms.value

    def kernel_name(self, kernel):
        return self.lib.amvhip_kernel_name(kernel).decode()
