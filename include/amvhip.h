/*
 * amvhip.h -- C ABI of libamvhip.so, the MI355X (gfx950) AMV codec hot path.
 *
 * Two surfaces, both plain C (no torch / C++ types in any signature):
 *
 *  1. The amvlib call surface of the reference (tomvanbraeckel/amv-codec-tools),
 *     same names, argument meaning, struct layouts and return codes, so that a host
 *     built against C-AMVDecoder/amvlib/{AMVDec.h,AmvJpeg.h,AdpcmIma.h} links against
 *     this library instead of AmvLib.dll / the amvlib objects.  Each declaration cites
 *     the reference interface it replaces (paths relative to the reference tree).
 *
 *  2. A batch surface (new): many independent chunks per call, device-resident
 *     inputs/outputs, asynchronous on a caller-supplied HIP stream.  This is what the
 *     AMVmuxer / a player loop should call when it has more than one frame in hand;
 *     INTEGRATION.md shows the binding.
 *
 * Every entry point runs on the GPU.  There is no CPU fallback: if no HIP device is
 * usable the calls fail (amvlib surface: -1 / NULL, batch surface: AMVHIP_ERR_DEVICE).
 */
#ifndef AMVHIP_H
#define AMVHIP_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* =====================================================================================
 * 1. amvlib surface
 * =================================================================================== */

/* C-AMVDecoder/amvlib/AMVDec.h:29-47 */
typedef struct _amv_important_header_data {
    unsigned int dwMicroSecPerFrame;
    unsigned int dwWidth;
    unsigned int dwHeight;
    unsigned int dwSpeed;
    unsigned int dwTimeSec;
    unsigned int dwTimeMin;
    unsigned int dwTimeHour;
    unsigned short wFormatTag;
    unsigned short nChannels;
    unsigned int nSamplesPerSec;
    unsigned int nAvgBytesPerSec;
    unsigned short nBlockAlign;
    unsigned short wBitsPerSample;
    unsigned short cbSize;
    unsigned short wSamplesPerBlock;
} AMVInfo;

/* AMVDec.h:50-57 */
typedef struct _frame_buffer_struct {
    unsigned char *videobuff;
    unsigned char *audiobuff;
    unsigned int videobufflen;
    unsigned int audiobufflen;
    int framenum;
} FRAMEBUFF;

/* AMVDec.h:59-63 */
typedef struct _video_buffer_struct {
    unsigned char *fbmpdat;
    unsigned int len;
} VIDEOBUFF;

/* AMVDec.h:65-71 */
#define AUDIO_FILE_TYPE_PCM 0
#define AUDIO_FILE_TYPE_ADPCM_IMA 1
typedef struct _audio_buffer_struct {
    short *audiodata;
    unsigned int len;
} AUDIOBUFF;

/* AMVDec.h:74-91 */
typedef struct _amv_decode_struct {
    char *amvfilename;
    int opened;
    long dataseekpos;
    long fileseekpos;
    AMVInfo amvinfo;
    unsigned int currentframe;
    unsigned int totalframe;
    FRAMEBUFF framebuf;
    VIDEOBUFF videobuf;
    AUDIOBUFF audiobuf;
} AMVDecoder;

/* C-AMVDecoder/amvlib/AdpcmIma.h:11-25 */
typedef struct ADPCMChannelStatus {
    int predictor;
    short int step_index;
    int step;
    int prev_sample;
} ADPCMChannelStatus;

typedef struct ADPCMContext {
    int channel;
    ADPCMChannelStatus status[2];
    short sample_buffer[32];
} ADPCMContext;

/* AmvJpeg.h:95-96 (AmvJpeg.c:1396,1515).  PrepareForVideoDecode rebuilt constant tables per
 * frame in the reference; here the tables live in the code object and the call only
 * validates its argument.  AmvJpegDecode: video->fbmpdat must hold
 * ((dwWidth*24+31)/32)*4*dwHeight bytes; returns 0 ok, -1 on any decode error. */
void PrepareForVideoDecode(AMVInfo *info);
int AmvJpegDecode(AMVInfo *info, FRAMEBUFF *inbuff, VIDEOBUFF *video);

/* AdpcmIma.h:28-32 (AdpcmIma.c:206,92).  Decode: AMV nibble order (high nibble first); c->channel == 2 follows
 * the reference's stereo branch (of every 8 bytes, 4 are the left channel's and 4 the right's; samples leave
 * interleaved, each channel with the predictor and step index of c->status[ch]);
 * returns bytes consumed (>0) or -1.  Encode: the reference's IMA-WAV-layout routine, which
 * nothing in the reference calls; kept for ABI completeness. */
int AdpcmImaDecodeFrame(ADPCMContext *c, void *data, int *data_size, unsigned char *buf, int buf_size);
int AdpcmImaEncodeFrame(ADPCMContext *c, int channels, int frame_size, unsigned char *frame,
                        int buf_size, void *data);

/* AMVDec.h:94-109 (AMVDec.c:15,131,150,240,259,288,358,384).  The container reader is
 * restated LP64-safe (the reference's AMVHeader.h uses `unsigned long` DWORDs). */
AMVDecoder *AmvOpen(const char *amvname);
void AmvClose(AMVDecoder *amv);
int AmvReadNextFrame(AMVDecoder *amv);
int AmvRewindFrameStart(AMVDecoder *amv);
int AmvVideoDecode(AMVDecoder *amv);
int AmvAudioDecode(AMVDecoder *amv);

/* AMVDec.h:102-107 and AmvJpeg.h:93-94 (AMVDec.c:342-547, AmvJpeg.c:315-414,1289-1393): the export
 * helpers.  JPEG stills are the chunk's scan behind a standard JFIF header with amvlib's tables;
 * AmvConvertJpegFileToBmpFile reads back exactly such files (the reference's is a general baseline
 * JPEG reader; files with other tables or sampling factors return -1 here) and decodes on the GPU.
 * AmvCreateWavFileFromAmvFile: type AUDIO_FILE_TYPE_PCM decodes every chunk (GPU), AUDIO_FILE_TYPE_ADPCM_IMA
 * copies the nibbles.  0 on success, -1 on error, like the reference. */
void AmvJpegPutHeader(FILE *fp, unsigned short height, unsigned short width);
int AmvCreateJpegFileFromFrameBuffer(AMVDecoder *amv, const char *dirname);
int AmvCreateJpegFileFromBuffer(AMVInfo *amvinfo, FRAMEBUFF *framebuf, const char *filename);
int AmvConvertJpegFileToBmpFile(const char *jpgname, const char *bmpname);
int ConvertJpegFileToBmpFile(const char *jpgname, const char *bmpname);   /* AmvJpeg.h:94: the same function under its AmvJpeg.c name */
int AmvCreateWavFileFromAmvFile(AMVDecoder *amv, int type, const char *wavfile);

/* the names BASELINE.json's north_star uses; thin aliases of the single-frame paths.
 * decode: chunk -> BGR24 (amvlib layout); encode: RGB24/BGR24 top-down -> chunk, returns length. */
int decode_amv_frame(const unsigned char *chunk, unsigned int len, unsigned int width,
                     unsigned int height, unsigned char *bgr_out);
int encode_amv_frame(const unsigned char *pixels, unsigned int stride, unsigned int width,
                     unsigned int height, int is_bgr, unsigned char *chunk_out, unsigned int cap);

/* =====================================================================================
 * 2. batch surface
 * =================================================================================== */

typedef struct amvhip_ctx amvhip_ctx;

#define AMVHIP_MAX_DIM 16384 /* widest / tallest picture the batch ABI takes (AMVHIP_ERR_ARG beyond; the amvlib surface: -1) */
#define AMVHIP_OK 0
#define AMVHIP_ERR_ARG -1     /* null pointer, zero/odd size, ... */
#define AMVHIP_ERR_DEVICE -2  /* no usable HIP device / HIP call failed (see amvhip_last_error) */
#define AMVHIP_ERR_NOMEM -3
#define AMVHIP_ERR_SPACE -4   /* output capacity too small */

/* per-frame decode status bits written to status[i] (0 = frame ok) */
#define AMVHIP_ST_FORMAT 1u    /* invalid Huffman code (amvlib: FUNC_FORMAT_ERROR, AmvJpeg.c:887) */
#define AMVHIP_ST_OVERRUN 2u   /* coefficient index ran past 63 (amvlib writes out of bounds there) */
#define AMVHIP_ST_TRUNCATED 4u /* scan needs more bits than the chunk holds */

/* decode flags */
#define AMVHIP_FLAG_ZIGZAG_FIXED 1u /* standard zig-zag instead of amvlib's table (AmvJpeg.c:138) */
/* FFmpeg-compat mode: the output of the patched FFmpeg's amv_decoder instead of amvlib's -- sp5x "Q60" quantiser
 * tables (libavcodec/sp5xdec.c:40,60-61, sp5x.h:187-194), standard zig-zag (dsputil.c:50-59), last_dc = 1024 on
 * dequantised DC (mjpegdec.c:805,388-390), simple_idct_put (simple_idct.c:390-398) and planar YUVJ420P output
 * flipped with the formula of mjpegdec.c:672-677.  d_out then holds n * amvhip_yuv420_frame_bytes(w,h) bytes:
 * per frame the Y plane (w*h), then Cb and Cr ((w+1)/2 x (h+1)/2 each), rows tight.  The two reference decoders
 * do NOT agree with each other; amvlib's output is the default and the headline target. */
#define AMVHIP_FLAG_FFMPEG 2u
/* ... and, with it, what the patched FFmpeg leaves in the picture when a chunk is damaged: mjpeg_decode_scan returns at
 * the block whose decode_block fails (mjpegdec.c:699-706) with every block before it already put into the picture
 * (:708-716) and everything else as the picture buffer was.  So: every whole block before the frame's first error is
 * written -- the blocks of the failing MCU in front of the failing one included -- and every other byte of the frame in
 * d_out stays as the caller had it (no plane row is cleared either, not even for an undamaged frame: a host that wants
 * FFmpeg's "shows what it has" hands in the picture before).  Without this flag AMVHIP_FLAG_FFMPEG zero-fills from the
 * failing MCU on, as amvlib does (AMVDec.c:283).  WHERE a chunk fails is this library's strict decoder's verdict
 * (status bits above) in both modes -- mjpegdec.c's own (:384 "error dc", :420 "error count", no truncation check: it
 * reads what lies behind the chunk) is not a function of the chunk alone.  AMVHIP_ERR_ARG without AMVHIP_FLAG_FFMPEG. */
#define AMVHIP_FLAG_FFMPEG_KEEP 4u

/* encode quantiser bias in 1/256 of a step: 0 = the reference's AMV setting
 * (mpegvideo_enc.c:492-496), 128 = its MJPEG setting (round to nearest, :488-490). */
#define AMVHIP_QBIAS_AMV 0
#define AMVHIP_QBIAS_MJPEG 128

int amvhip_create(amvhip_ctx **ctx, int device);
void amvhip_destroy(amvhip_ctx *ctx);
const char *amvhip_last_error(const amvhip_ctx *ctx);
int amvhip_device(const amvhip_ctx *ctx);

/* geometry helpers (AmvJpeg.c:420,1524: stride; :1276-1284: MCU grid) */
uint32_t amvhip_stride(uint32_t width);
uint64_t amvhip_frame_bytes(uint32_t width, uint32_t height);
uint64_t amvhip_yuv420_frame_bytes(uint32_t width, uint32_t height);   /* YUVJ420P frame, tight planes (mjpegdec.c:312) */
uint32_t amvhip_encode_bound(uint32_t width, uint32_t height);
/* the bytes AmvJpegPutHeader writes (SOI ... SOS) for a picture size; returns their number (623) and
 * copies them to out when cap allows.  Host only. */
uint32_t amvhip_jpeg_header(uint16_t height, uint16_t width, uint8_t *out, uint32_t cap);

/*
 * Video decode, device-resident.  Replaces a loop of AmvVideoDecode/AmvJpegDecode calls
 * (AMVDec.c:259-286, AmvJpeg.c:1515-1539) over n independent chunks.
 *   d_blob     : all chunks back to back (each "FF D8" scan "FF D9"); the pointer must be 4-byte aligned (AMVHIP_ERR_ARG
 *                otherwise; chunks themselves may start at any byte)
 *   blob_bytes : the bytes of d_blob the chunks OCCUPY (the end of the last chunk), not the capacity of a larger
 *                buffer: chunks are bounds-checked against it, and the two workspaces between the decode stages -- the
 *                unstuffed scans, the coefficient records -- are laid out per frame from d_lens[i] (on the device) inside a
 *                total sized from blob_bytes here: 1x + 48 bytes per frame for the scans, 8x + (8 per block + 380) bytes per
 *                frame for the records.  Chunks that OVERLAP in the blob (their lengths add up to more than blob_bytes) are
 *                legal and decode correctly, but the layout runs out and the frames behind that point are decoded by the
 *                one-lane-per-frame kernel: same bytes, slower; so is a frame with more than two records per chunk byte
 *                (amvhip_entropy_stats reports how many frames of the last call took that route)
 *   d_offs[i]  : byte offset of chunk i in d_blob;  d_lens[i]: its length
 *   d_out      : n * amvhip_frame_bytes(w,h), 4-byte aligned; frame i is BGR24, row 0 = top, rows padded to
 *                amvhip_stride(w); pixels of MCUs after a failing one are zero (AMVDec.c:283)
 *   d_status   : n int32, AMVHIP_ST_* bits
 *   stream     : hipStream_t (NULL = default stream).  Asynchronous: returns after enqueue.
 * Workspace is owned by ctx and grown on demand (a hipMalloc on first use of a larger
 * batch; none in steady state).  Calls on one context reuse that workspace, so they must be
 * ordered on the device: keep a context on one stream at a time (the lock inside only orders the
 * enqueueing) and create one context per stream for work that is meant to overlap.
 */
int amvhip_decode_batch_dev(amvhip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_bytes,
                            const uint64_t *d_offs, const uint32_t *d_lens, uint32_t n,
                            uint32_t width, uint32_t height, uint32_t flags,
                            uint8_t *d_out, int32_t *d_status, void *stream);

/*
 * The same decode in two halves, for a caller that has the next batch in hand before it needs the last one's pixels
 * (a player or transcoder working through windows of frames): the entropy stage of batch k+1 then runs beside the
 * reconstruction of batch k -- the first waits on memory, the second on the vector ALUs.
 *   amvhip_decode_submit_dev : arguments as amvhip_decode_batch_dev.  The inputs must be ready where `stream` stands
 *       at the call (that point is recorded); the work is queued on streams the context owns and `stream` is NOT made
 *       to wait.  d_blob/d_offs/d_lens must stay untouched and d_out/d_status belong to the batch until it has been
 *       collected and `stream` has passed that point: a batch submitted meanwhile needs buffers of its own.
 *   amvhip_decode_collect_dev: `stream` waits for the oldest batch submitted and not yet collected; work queued on
 *       `stream` afterwards sees its d_out and d_status.
 * At most two batches between submit and collect (a third submit returns AMVHIP_ERR_ARG): what the entropy stage hands
 * to the reconstruction exists twice in the context.  Any other entry point of the context first waits for the batches
 * submitted so far.  Results are those of amvhip_decode_batch_dev, byte for byte.
 */
int amvhip_decode_submit_dev(amvhip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_bytes,
                             const uint64_t *d_offs, const uint32_t *d_lens, uint32_t n,
                             uint32_t width, uint32_t height, uint32_t flags,
                             uint8_t *d_out, int32_t *d_status, void *stream);
int amvhip_decode_collect_dev(amvhip_ctx *ctx, void *stream);

/* device workspace the context held for the last amvhip_decode_batch_dev call, in bytes per frame of that call */
double amvhip_decode_workspace_per_frame(const amvhip_ctx *ctx);

/* Same with host buffers: H2D, decode, D2H, synchronous. */
int amvhip_decode_batch(amvhip_ctx *ctx, const uint8_t *blob, uint64_t blob_bytes,
                        const uint64_t *offs, const uint32_t *lens, uint32_t n,
                        uint32_t width, uint32_t height, uint32_t flags,
                        uint8_t *out, int32_t *status);

/*
 * Asynchronous host-buffer forms, for a host that wants the GPU working while it does something else (the
 * read-ahead behind AmvReadNextFrame uses them): copies and kernels are queued on a stream the context owns and
 * the call returns; amvhip_sync waits for everything queued so far.  blob/offs/lens must stay untouched, and
 * out/status unread, until then.  Page-locked buffers (amvhip_host_alloc) make the copies truly asynchronous;
 * pageable ones work but block.  The host-buffer entry points of one context queue their uploads and kernels on that
 * one stream, in order; the decoded frames of amvhip_decode_batch_async come back on a second stream of the context
 * (so that the copy of one call runs beside the upload and the kernels of the next).  The results of an *_async call
 * are therefore complete only after amvhip_sync, which waits for both streams -- not after some other blocking
 * host-buffer call (the blocking encode and ADPCM entry points synchronise the first stream alone).
 */
int amvhip_decode_batch_async(amvhip_ctx *ctx, const uint8_t *blob, uint64_t blob_bytes,
                              const uint64_t *offs, const uint32_t *lens, uint32_t n,
                              uint32_t width, uint32_t height, uint32_t flags,
                              uint8_t *out, int32_t *status);
int amvhip_sync(amvhip_ctx *ctx);
int amvhip_host_alloc(amvhip_ctx *ctx, void **p, size_t bytes);
void amvhip_host_free(amvhip_ctx *ctx, void *p);

/* Entropy-stage kernel choice.  AUTO: several lanes per frame (self-synchronising parallel Huffman
 * decode); frames whose chunk does not fit its workspace window fall to the one-lane-per-frame kernel.
 * SERIAL: always the one-lane-per-frame kernel.  Results are identical; the switch exists so the
 * parity tests cover both kernels. */
#define AMVHIP_ENTROPY_AUTO 0
#define AMVHIP_ENTROPY_SERIAL 1
int amvhip_set_entropy_mode(amvhip_ctx *ctx, int mode);
/* Diagnostics of the synchronising entropy kernel: synchronises the device, returns the counters
 * gathered since the last call in out[] = {frames, sum of synchronisation rounds, max rounds, frames the LAST decode
 * call handed to the one-lane-per-frame kernel (always reported, whether gathering is on or not),
 * then shader clocks summed over waves for: coefficient zeroing, first walk, synchronisation
 * rounds, writing pass, DC pass, and the number of waves}, clears them and switches gathering on
 * or off (off by default; costs a few atomics per frame). */
int amvhip_entropy_stats(amvhip_ctx *ctx, int enable, uint64_t out[10]);
/* ... and, per task (wave) of the last launch of the several-lanes-per-frame kernel while gathering was on, a line of
 * eight words: {constant-rate clock (100 MHz) at the task's begin, at its end, shader clocks of the first walk, the
 * synchronisation rounds, the strict pass, the DC pass, rounds of the wave's worst frame | share length in bits << 32,
 * workgroup << 32 | wave << 16 | lanes per frame}.  out: 8 * tasks words; returns the lines copied (< 0: error).  Lines behind
 * the launch's tasks are zero when no launch since gathering went on had more (amvhip_entropy_stats clears them then).
 * What a launch one generation of waves deep lasts as long as: its slowest task (tools/time_kernels.py --trace). */
int amvhip_entropy_trace(amvhip_ctx *ctx, uint64_t *out, uint32_t tasks);
/* A batch large enough to give every frame ONE entropy lane (a wave's 64 frames then finish together) gives the frames
 * whose chunk is over twice the batch's mean chunk several lanes each instead -- the split is made on the device, from
 * d_lens.  out[] = {frames of the LAST decode call that went the several-lanes way, frames that went one lane per frame};
 * {0, 0} when the call made no split (a smaller batch: every frame has several lanes anyway).  Synchronises the device.
 * No analogue in the reference (its decoder takes one frame at a time, AMVDec.c:259). */
int amvhip_decode_split_stats(amvhip_ctx *ctx, uint32_t out[2]);

/* Stage access for parity tests: entropy stage only.  d_coef: n * nmcu*6*64 int16,
 * DC-predicted quantised coefficients in bitstream order (amvlib MCUBuffer,
 * AmvJpeg.c:1200-1223); d_nmcu_ok: MCUs decoded before the first error. */
int amvhip_huffman_decode_dev(amvhip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_bytes,
                              const uint64_t *d_offs, const uint32_t *d_lens, uint32_t n,
                              uint32_t width, uint32_t height,
                              int16_t *d_coef, int32_t *d_status, uint32_t *d_nmcu_ok, void *stream);
/* Stage access: dequantise + IDCT + colour + store from coefficients
 * (IQtIZzBlock/Fast_IDCT/GetYUV/StoreBuffer, AmvJpeg.c:1010-1059,754-840). */
int amvhip_reconstruct_dev(amvhip_ctx *ctx, const int16_t *d_coef, const uint32_t *d_nmcu_ok,
                           uint32_t n, uint32_t width, uint32_t height, uint32_t flags,
                           uint8_t *d_out, void *stream);

/*
 * Video encode, device-resident.  Replaces amv_encode_picture + the RGB front end
 * (mjpegenc.c:454-472, imgconvert_template.h:654, mpegvideo_enc.c:3647, mjpegenc.c:379-450,
 * :282-355) over n frames.
 *   d_pix      : n frames, RGB24 (is_bgr=0) or BGR24 (is_bgr=1), top-down, pix_stride bytes/row,
 *                frame i at d_pix + i*pix_stride*height; width and height must be even
 *   d_blob     : output of blob_cap bytes; chunks are written back to back in frame order.
 *                n * amvhip_encode_bound(w,h) always suffices (real streams run at ~0.2 byte per pixel).
 *                A chunk that would end past blob_cap is NOT written and its d_lens entry is 0 (d_offs still
 *                says where it would have started): a consumer checks d_lens[i] != 0.  The host-buffer forms
 *                return AMVHIP_ERR_SPACE instead.
 *   d_offs/d_lens : n entries written
 */
int amvhip_encode_batch_dev(amvhip_ctx *ctx, const uint8_t *d_pix, uint32_t pix_stride, int is_bgr,
                            uint32_t n, uint32_t width, uint32_t height, uint32_t qbias,
                            uint8_t *d_blob, uint64_t blob_cap, uint64_t *d_offs, uint32_t *d_lens,
                            void *stream);
int amvhip_encode_batch(amvhip_ctx *ctx, const uint8_t *pix, uint32_t pix_stride, int is_bgr,
                        uint32_t n, uint32_t width, uint32_t height, uint32_t qbias,
                        uint8_t *blob, uint64_t blob_cap, uint64_t *offs, uint32_t *lens);
/*
 * The same from planar YUVJ420P, the pixel format the reference's amv_encoder declares and takes
 * (mjpegenc.c:485-494 pix_fmts; amv_encode_picture :454-472 flips it by negative linesize; get_pixels
 * mpegvideo_enc.c:1539-1549): no colour conversion, otherwise the path above.  Frame i's planes start at
 * d_y + i*y_frame_stride and d_cb/d_cr + i*c_frame_stride (bytes), rows y_stride / c_stride apart; chroma planes
 * are (w/2) x (h/2).  rgb24_to_yuvj420p followed by this entry equals amvhip_encode_batch_dev bit for bit.
 */
int amvhip_encode_yuv420_batch_dev(amvhip_ctx *ctx, const uint8_t *d_y, const uint8_t *d_cb, const uint8_t *d_cr,
                                   uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                                   uint64_t c_frame_stride, uint32_t n, uint32_t width, uint32_t height,
                                   uint32_t qbias, uint8_t *d_blob, uint64_t blob_cap, uint64_t *d_offs,
                                   uint32_t *d_lens, void *stream);
int amvhip_encode_yuv420_batch(amvhip_ctx *ctx, const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                               uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                               uint64_t c_frame_stride, uint32_t n, uint32_t width, uint32_t height,
                               uint32_t qbias, uint8_t *blob, uint64_t blob_cap, uint64_t *offs, uint32_t *lens);
/*
 * ... and from planar YUVJ422P, the other pixel format amv_encoder declares (mjpegenc.c:493; chroma planes (w/2) x h,
 * mpegvideo_enc.c:534-543 h/v sampling 2x2 : 1x2).  The reference would code such a picture as MCUs of eight blocks
 * (ff_mjpeg_encode_mb, mjpegenc.c:437-450, the CHROMA_420 test failing) -- a scan no AMV decoder can read: the container
 * has no frame header, and both amvlib (AmvJpeg.c:1406-1420) and the reference's own amv decoder (sp5xdec.c:51-91, a
 * fixed 4:2:0 SOF) take six blocks per MCU.  Here the two chroma rows over every 4:2:0 sample are averaged ((a + b + 1)
 * >> 1) and the picture is coded as above: a valid AMV chunk.  Arguments as amvhip_encode_yuv420_batch(_dev).
 */
int amvhip_encode_yuv422_batch_dev(amvhip_ctx *ctx, const uint8_t *d_y, const uint8_t *d_cb, const uint8_t *d_cr,
                                   uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                                   uint64_t c_frame_stride, uint32_t n, uint32_t width, uint32_t height,
                                   uint32_t qbias, uint8_t *d_blob, uint64_t blob_cap, uint64_t *d_offs,
                                   uint32_t *d_lens, void *stream);
int amvhip_encode_yuv422_batch(amvhip_ctx *ctx, const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                               uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                               uint64_t c_frame_stride, uint32_t n, uint32_t width, uint32_t height,
                               uint32_t qbias, uint8_t *blob, uint64_t blob_cap, uint64_t *offs, uint32_t *lens);
/*
 * The picture rescaler in front of the encoder: img_resample (libavcodec/imgresample.c:474-495), the arithmetic of
 * the sws_scale shim (:599) ffmpeg.c:757 runs when the source is not the target size (AMVmuxer/Makefile:15-17 asks for
 * -s 160x120): four-tap, 16-phase polyphase filter built as av_build_filter does (resample2.c:93-140), horizontal pass
 * into bytes, vertical pass over four such lines, edges repeated; the luma increments and filters also serve the
 * chroma planes, which are (w >> 1) x (h >> 1).  YUV420P frames in, YUV420P frames out, plane layout as in
 * amvhip_encode_yuv420_batch_dev; at most 65535 frames per call.  Byte-identical to the reference's routine.
 */
int amvhip_resample_yuv420_dev(amvhip_ctx *ctx, const uint8_t *d_src_y, const uint8_t *d_src_cb, const uint8_t *d_src_cr,
                               uint32_t src_y_stride, uint32_t src_c_stride, uint64_t src_y_frame_stride,
                               uint64_t src_c_frame_stride, uint32_t src_width, uint32_t src_height,
                               uint8_t *d_dst_y, uint8_t *d_dst_cb, uint8_t *d_dst_cr,
                               uint32_t dst_y_stride, uint32_t dst_c_stride, uint64_t dst_y_frame_stride,
                               uint64_t dst_c_frame_stride, uint32_t dst_width, uint32_t dst_height, uint32_t n, void *stream);
/* rescale to width x height, then encode: one call for ffmpeg.c's sws_scale + avcodec_encode_video pair (:757-814) */
int amvhip_encode_yuv420_scaled_batch_dev(amvhip_ctx *ctx, const uint8_t *d_y, const uint8_t *d_cb, const uint8_t *d_cr,
                                          uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                                          uint64_t c_frame_stride, uint32_t src_width, uint32_t src_height, uint32_t n,
                                          uint32_t width, uint32_t height, uint32_t qbias, uint8_t *d_blob,
                                          uint64_t blob_cap, uint64_t *d_offs, uint32_t *d_lens, void *stream);
/* Stage access: quantised coefficients (zig-zag order, not predicted), n*nmcu*6*64 int16 */
int amvhip_encode_coefs_dev(amvhip_ctx *ctx, const uint8_t *d_pix, uint32_t pix_stride, int is_bgr,
                            uint32_t n, uint32_t width, uint32_t height, uint32_t qbias,
                            int16_t *d_coef, void *stream);

/*
 * IMA ADPCM, AMV chunk layout (AMVDec.c:312-320 + AdpcmIma.c:206-242; adpcm.c:461-498).
 * decode: chunk i = {s16 predictor, u8 step_index, u8 0, u32 nsamples, nibbles}; writes
 *         2*(len-8) samples at d_pcm + d_pcm_offs[i] (offsets in samples).
 * encode: chunk i takes d_nsamp[i] (even) samples from d_pcm + d_pcm_offs[i] and writes
 *         8 + nsamp/2 bytes at d_blob + d_offs[i].  step_index handling:
 *         d_step_in != NULL : chunk i starts from d_step_in[i] (independent chunks);
 *         d_step_in == NULL : the reference's behaviour, step_index carried from chunk i-1
 *                             (chunk 0 starts at 0); chunks of one call form one stream.  The chain is
 *                             resolved on the device (guessed starts, chunks that guessed wrong coded
 *                             again; a stream that does not settle takes an exhaustive 89-start map
 *                             instead), never by a serial pass and never by waiting for the stream.
 */
int amvhip_adpcm_decode_batch_dev(amvhip_ctx *ctx, const uint8_t *d_blob, uint64_t blob_bytes,
                                  const uint64_t *d_offs, const uint32_t *d_lens, uint32_t n,
                                  int16_t *d_pcm, const uint64_t *d_pcm_offs,
                                  int32_t *d_final_state /* optional: n x {predictor, step_index} */,
                                  void *stream);
int amvhip_adpcm_encode_batch_dev(amvhip_ctx *ctx, const int16_t *d_pcm, const uint64_t *d_pcm_offs,
                                  const uint32_t *d_nsamp, uint32_t n, const int32_t *d_step_in,
                                  uint8_t *d_blob, const uint64_t *d_offs, void *stream);
/* The 89 factors r[i] with which the encode kernels take min(7, |delta| * 4 / step[i]) as trunc((float)|delta| * r[i])
 * (host arithmetic, no device needed): published so that the exactness of that shortcut can be checked off the device. */
void amvhip_adpcm_quotient_table(float out[89]);
/* Diagnostic of the last amvhip_adpcm_encode_batch_dev call with d_step_in == NULL (waits for the device):
 * out[0] = 1 if the stream took the exhaustive route, out[1..] = chunks coded again in sweep 1, 2, ... (0-terminated,
 * at most 60).  Returns AMVHIP_OK, or AMVHIP_ERR_ARG when no such call has been made. */
int amvhip_adpcm_chain_stats(amvhip_ctx *ctx, uint32_t out[64]);
/* host-buffer forms (H2D, kernel, D2H, synchronous).  pcm_samples / blob_bytes are the sizes of
 * the whole pcm / blob arrays the offsets index into. */
int amvhip_adpcm_decode_batch(amvhip_ctx *ctx, const uint8_t *blob, uint64_t blob_bytes,
                              const uint64_t *offs, const uint32_t *lens, uint32_t n,
                              int16_t *pcm, uint64_t pcm_samples, const uint64_t *pcm_offs,
                              int32_t *final_state);
int amvhip_adpcm_decode_batch_async(amvhip_ctx *ctx, const uint8_t *blob, uint64_t blob_bytes,
                                    const uint64_t *offs, const uint32_t *lens, uint32_t n,
                                    int16_t *pcm, uint64_t pcm_samples, const uint64_t *pcm_offs,
                                    int32_t *final_state);   /* see amvhip_decode_batch_async */
int amvhip_adpcm_encode_batch(amvhip_ctx *ctx, const int16_t *pcm, uint64_t pcm_samples,
                              const uint64_t *pcm_offs, const uint32_t *nsamp, uint32_t n,
                              const int32_t *step_in, uint8_t *blob, uint64_t blob_bytes,
                              const uint64_t *offs);
/* One AMV audio chunk, host buffers, the step index handed in and out (what adpcm_encode_frame keeps in its
 * context between calls, adpcm.c:461-498).  nsamp even and > 0; writes 8 + nsamp/2 bytes, returns that count. */
int amvhip_adpcm_encode_frame(amvhip_ctx *ctx, const int16_t *samples, uint32_t nsamp, int32_t *step_index,
                              uint8_t *chunk, uint32_t cap);
/* The reference's `-trellis N` quality mode (adpcm_compress_trellis, adpcm.c:287-443, as the AMV case calls it :482-487):
 * a beam search over the 2^N best decoder states instead of the plain quantiser, 1 <= N <= 5; same chunk layout, lower
 * error.  The batch form takes independent chunks (d_step_in required, d_step_out optional: the index each chunk ends
 * on); the frame form hands the index in and out like amvhip_adpcm_encode_frame. */
int amvhip_adpcm_encode_trellis_batch_dev(amvhip_ctx *ctx, const int16_t *d_pcm, const uint64_t *d_pcm_offs,
                                          const uint32_t *d_nsamp, uint32_t n, const int32_t *d_step_in, uint32_t trellis,
                                          uint8_t *d_blob, const uint64_t *d_offs, int32_t *d_step_out, void *stream);
int amvhip_adpcm_encode_frame_trellis(amvhip_ctx *ctx, const int16_t *samples, uint32_t nsamp, int32_t *step_index,
                                      uint32_t trellis, uint8_t *chunk, uint32_t cap);
/* The framing the reference's AMV audio encoder and muxer apply around the kernel (host arithmetic only):
 * amvhip_amv_audio_pairs      adpcm.c:469-477,497: sample pairs of the next chunk for a nominal frame_size (odd sizes
 *                             alternate, a chunk that would straddle a whole second is stretched to end on it);
 *                             *extra (0/1) and *samples_written are the stream state, both start at 0.
 * amvhip_amv_audio_frame_size amvenc.c:276-281: frame_size = sample_rate * time_base (22050/16 -> 1378). */
uint32_t amvhip_amv_audio_pairs(uint32_t frame_size, uint32_t sample_rate, uint32_t *extra, uint64_t *samples_written);
uint32_t amvhip_amv_audio_frame_size(uint32_t sample_rate, uint32_t tb_num, uint32_t tb_den);
/* amvlib's IMA-WAV-layout encoder (AdpcmIma.c:43-160), one mono frame, host buffers; backs
 * AdpcmImaEncodeFrame.  state = {prev_sample (out), step_index (in/out)}; returns bytes written. */
int amvhip_adpcm_wav_encode_frame(amvhip_ctx *ctx, const int16_t *samples, int frame_size,
                                  int32_t state[2], uint8_t *frame, int buf_size);

/* Seeded synthetic sources of BASELINE.md section 4, generated on the device
 * (integer-only; byte-identical to the CPU generator used by the parity tests). */
int amvhip_synth_frames_dev(amvhip_ctx *ctx, uint32_t seed, uint32_t first_frame, uint32_t n,
                            uint32_t width, uint32_t height, uint8_t *d_rgb, void *stream);
int amvhip_synth_audio_dev(amvhip_ctx *ctx, uint32_t seed, uint64_t first_sample, uint64_t n,
                           int16_t *d_pcm, void *stream);

/*
 * AMV container writer (host C, no device work): the muxer half of the path, with the layout
 * AMVmuxer/ffmpeg/libavformat/amvenc.c writes -- 304-byte header (amvh :128-177, two strl lists
 * :181-262), "movi" at 0x138, unpadded 00dc/01wb chunks (:317-321) in strict video/audio alternation
 * (:378-406), counters and duration patched at close (:72-114), "AMV_END_" trailer (:332).  The reader
 * half is AmvOpen / AmvReadNextFrame above.  bit rates: what FFmpeg's codec contexts would hold
 * (its defaults are 200000 and 64000); they only fill the two byte-rate fields.
 * open: NULL on error; write_frame / close: 0 or -1.
 */
typedef struct amvhip_muxer amvhip_muxer;
amvhip_muxer *amvhip_mux_open(const char *path, uint32_t width, uint32_t height, uint32_t fps,
                              uint32_t sample_rate, uint32_t video_bit_rate, uint32_t audio_bit_rate);
int amvhip_mux_write_frame(amvhip_muxer *m, const uint8_t *video, uint32_t video_len, const uint8_t *audio,
                           uint32_t audio_len);
int amvhip_mux_close(amvhip_muxer *m);

/*
 * Kernel timing with HIP events recorded on the launch stream around every kernel
 * launch (bench.py's roofline leg).  Off by default.  amvhip_prof_read synchronises the
 * events recorded since the last reset and returns launches / summed milliseconds.
 */
#define AMVHIP_K_HUFFMAN 0
#define AMVHIP_K_RECON 1
#define AMVHIP_K_FDCT 2
#define AMVHIP_K_PACK 3
#define AMVHIP_K_ADPCM_DEC 4
#define AMVHIP_K_ADPCM_ENC 5       /* map + chain + encode kernels, timed as one (the map dominates) */
#define AMVHIP_K_SYNTH 6
#define AMVHIP_K_HUFFMAN_SERIAL 7
#define AMVHIP_K_UNSTUFF 8
#define AMVHIP_K_PACK_SERIAL 9
#define AMVHIP_K_COMPACT 10   /* amv_scan_kernel + amv_gather_kernel, timed as one */
#define AMVHIP_K_COUNT 12
void amvhip_prof_enable(amvhip_ctx *ctx, int on);
void amvhip_prof_reset(amvhip_ctx *ctx);
int amvhip_prof_read(amvhip_ctx *ctx, int kernel, uint64_t *launches, double *total_ms);
const char *amvhip_kernel_name(int kernel);

#ifdef __cplusplus
}
#endif
#endif /* AMVHIP_H */
