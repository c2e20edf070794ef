/*
 * amv_oracle.h -- CPU restatement of the AMV codec hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity checker for the HIP path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library (libamvhip.so)
 * never links, loads or calls anything in this directory.
 *
 * It is a plain-C, single-threaded, bit-serial restatement of what the reference
 * (tomvanbraeckel/amv-codec-tools) computes; every function cites the reference
 * file:line it follows.  Reference paths are relative to /root/reference.
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   - video decode: FNV-1a-64 over all 252 decoded frames of the reference's own
 *     fixture C-AMVDecoder/bin/AMV1.amv must equal adc922c6366237b5 (amvlib run
 *     recorded in SURVEY.md section 8c / Appendix A step 5); with the zig-zag quirk
 *     disabled it must equal a3f28348069fce7c (same source).
 *   - audio decode, forward DCT: checked against oracle/_ref/libamvref.so, which is
 *     compiled directly from the reference's AdpcmIma.c and jfdctint.c.
 *   - AmvJpeg.c itself needs <windows.h>, which this image lacks, so it is NOT built
 *     here (no stand-in headers are written).
 */
#ifndef AMV_ORACLE_H
#define AMV_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* decode status bits (0 = ok).  amvlib itself only knows 0 / -1 (AmvJpeg.c:1531-1538). */
#define AMVO_ST_FORMAT    1u /* Huffman code longer than 16 bits (AmvJpeg.c:887-888)          */
#define AMVO_ST_OVERRUN   2u /* run/size pushed the coefficient index past 63.  The reference  */
                             /* writes out of bounds there (AmvJpeg.c:967-969, UB); defined    */
                             /* here as an error that stops the frame like FORMAT does.        */
#define AMVO_ST_TRUNCATED 4u /* more bits consumed than the chunk holds.  The reference reads  */
                             /* past the buffer (UB); defined here as zero-extension + flag.   */

/* decode flags */
#define AMVO_FLAG_ZIGZAG_FIXED 1u /* use the correct zig-zag (31 at [3][4]) instead of amvlib's 37 */

/* row stride of the BGR24 output: AmvJpeg.c:420,1524  WIDTHBYTES(w*24) */
uint32_t amvo_stride(uint32_t w);
/* MCUs per frame: AmvJpeg.c:1276-1284 loop bounds */
uint32_t amvo_mcus_per_row(uint32_t w);
uint32_t amvo_mcu_rows(uint32_t h);

/*
 * Decode one video chunk ("FF D8" + scan + "FF D9") exactly as AmvJpegDecode does
 * (AmvJpeg.c:1515-1539 -> Decode :1244 -> DecodeMCUBlock/HufBlock/DecodeElement,
 * IQtIZzBlock, Fast_IDCT, GetYUV, StoreBuffer).
 *   out      : amvo_stride(w)*h bytes, zero-filled here first (AMVDec.c:283), BGR24.
 *   coef_out : optional, nmcu*6*64 int16, DC-predicted quantised coefficients in
 *              bitstream (zig-zag) order, block order Y0 Y1 Y2 Y3 U V per MCU
 *              (the contents of MCUBuffer, AmvJpeg.c:1200-1223); zero-filled first.
 *   nmcu_ok  : optional, number of MCUs fully decoded and stored.
 *   status   : optional, AMVO_ST_* bits.
 * returns 0 if status == 0 else -1 (AmvJpeg.c:1531-1538).
 */
int amvo_decode_frame(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h,
                      uint32_t flags, uint8_t *out, int16_t *coef_out,
                      uint32_t *nmcu_ok, uint32_t *status);

/* dequantise + (quirky) de-zig-zag + IDCT + level offset for one block
 * (IQtIZzBlock, AmvJpeg.c:1010-1048).  comp: 0 = Y, 1 = U, 2 = V. */
void amvo_dequant_idct_block(const int16_t coef[64], int comp, uint32_t flags, int32_t out[64]);
/* Fast_IDCT on 64 int32 in place (AmvJpeg.c:1050-1059,1082-1175) */
void amvo_idct_block(int32_t blk[64]);
/* StoreBuffer's per-pixel conversion (AmvJpeg.c:805-831); bgr[0..2] = B,G,R */
void amvo_yuv_to_bgr(int32_t y, int32_t u, int32_t v, uint8_t bgr[3]);

/* ---- FFmpeg-compat video decode (the patched FFmpeg's amv_decoder) ------ */
/* sp5x "Q60" tables in zig-zag order (sp5x.h:187-194 = sp5x_quant_table[10], [11]; sp5xdec.c:40,60-61) */
void amvo_q60_table(int chroma, uint8_t out[64]);
/* simple_idct (simple_idct.c:410-419) in place on 64 int16; pinned against the reference's own object */
void amvo_simple_idct(int16_t blk[64]);
/* simple_idct_put (:390-398): rows in place, columns clipped to 0..255 into dest */
void amvo_simple_idct_put(uint8_t *dest, int line_size, int16_t blk[64]);
/* decode_block's dequantisation (mjpegdec.c:388-390,417,424; last_dc = 1024 :805) of a scan-order block whose
 * DC is the running sum of differences; out in natural order, int16 like DCTELEM.  comp 0 = luma table */
void amvo_ffmpeg_dequant_block(const int16_t coef[64], int comp, int16_t out[64]);
uint32_t amvo_yuv420_frame_bytes(uint32_t w, uint32_t h);
/* whole frame: YUVJ420P planes (Y, Cb, Cr; tight rows), flipped per mjpegdec.c:672-677 */
int amvo_decode_frame_ffmpeg(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h,
                             uint8_t *out, uint32_t *nmcu_ok, uint32_t *status);
/* the entropy stage alone: coef[b][64] (scan order, DC predicted) of every whole block before the first error; returns their number */
uint32_t amvo_entropy_blocks(const uint8_t *chunk, uint32_t len, uint32_t nblocks, int16_t *coef, uint32_t *status);
/* AMVHIP_FLAG_FFMPEG_KEEP: blocks before the first error written, `out` otherwise untouched; blocks_ok = whole blocks decoded */
int amvo_decode_frame_ffmpeg_keep(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h,
                                  uint8_t *out, uint32_t *blocks_ok, uint32_t *status);

/* ---- picture rescale (the sws_scale shim of libavcodec/imgresample.c) ---------------- */
/* av_build_filter (resample2.c:93-140) as img_resample_full_init calls it (imgresample.c:468-471): 16 phases x 4 taps */
void amvo_build_resample_filter(int16_t filter[64], int out_size, int in_size);
/* component_resample (imgresample.c:341-405) of one plane */
void amvo_resample_plane(const uint8_t *in, int iwrap, int iw, int ih, uint8_t *out, int owrap, int ow, int oh,
                         int h_incr, int v_incr, const int16_t *hf, const int16_t *vf);
/* img_resample (:474-495): tight YUV420P frame iw x ih -> ow x oh (chroma planes (w >> 1) x (h >> 1)) */
void amvo_img_resample_yuv420(const uint8_t *in, int iw, int ih, uint8_t *out, int ow, int oh);

/* ---- IMA ADPCM (AMV layout) ------------------------------------------------------ */
/* AmvAudioDecode header parse (AMVDec.c:312-320) + AdpcmImaDecodeFrame (AdpcmIma.c:206-242)
 * with AdpcmImaExpandNibble (:170-204).  Writes 2*(len-8) samples (the defined part,
 * SURVEY.md 3.2); returns that count, or -1 if len <= 8.  *nsamples_hdr = le32 at +4. */
int amvo_adpcm_decode_chunk(const uint8_t *chunk, uint32_t len, int16_t *pcm, uint32_t *nsamples_hdr);

/* FFmpeg AMV encoder: adpcm_encode_frame CODEC_ID_ADPCM_IMA_AMV (adpcm.c:461-498) with
 * adpcm_ima_compress_sample (:219-227).  nsamp must be even.  *step_index is read and
 * updated (it persists across chunks in the reference).
 * Writes 8 + nsamp/2 bytes; returns that count. */
int amvo_adpcm_encode_chunk(const int16_t *samples, uint32_t nsamp, int *step_index, uint8_t *out);

/* The same with the reference's trellis search (adpcm.c:287-443, IMA branch; `-trellis N` of the ffmpeg CLI), frontier
 * 2^trellis, 1 <= trellis <= 5.  Unpinned restatement (adpcm.c's encoders are behind a configure switch).  -1 on a bad
 * trellis value. */
int amvo_adpcm_encode_chunk_trellis(const int16_t *samples, uint32_t nsamp, int *step_index, int trellis, uint8_t *out);

/* Per-chunk sample-pair count n of the FFmpeg AMV framing (adpcm.c:469-477):
 * odd frame_size carry and the once-per-second resync.  State is carried in
 * extra (0/1) and samples_written. */
uint32_t amvo_adpcm_amv_pairs(uint32_t frame_size, uint32_t sample_rate,
                              uint32_t *extra, uint64_t *samples_written);

/* amvlib's own encoder (AdpcmIma.c:43-160): IMA-WAV block layout, mono.  Nothing in the
 * reference calls it; restated because the symbol is part of the ABI.
 * samples: 1 + 8*(frame_size/8) values; state[0] = prev_sample (out), state[1] = step_index
 * (in/out).  Writes 4 + 4*(frame_size/8) bytes; returns that count. */
int amvo_adpcm_wav_encode_frame(const int16_t *samples, int frame_size, int32_t state[2], uint8_t *frame);

/* ---- video encode (own bitstream writer following the reference algorithm) -------- */
/* rgb24_to_yuvj420p (imgconvert_template.h:654-..., colorspace.h:30-97); w,h even.
 * bgr != 0 swaps the byte order of the source (the reference instantiates both, imgconvert.c:1660,1686) */
void amvo_rgb24_to_yuvj420p(const uint8_t *rgb, uint32_t src_stride, uint32_t w, uint32_t h,
                            int bgr, uint8_t *y, uint8_t *cb, uint8_t *cr);
/* ff_jpeg_fdct_islow (jfdctint.c:184-343), in place on 64 int16 */
void amvo_fdct_islow(int16_t blk[64]);
/* dct_quantize_c restricted to AMV (mpegvideo_enc.c:3647-3724, bias 0 :492-496,
 * qmat :80-91 with qscale 8) but with amvlib's quant tables and a true -128 level
 * shift.  in: fdct output (natural order, x8 scale);
 * out: quantised coefficients in zig-zag order.  comp 0 = luma table, else chroma.
 * qbias: intra_quant_bias in 1/256 steps: 0 = the reference's AMV setting (:492-496),
 * 128 = its MJPEG setting (:488-490). */
void amvo_quantize_block(const int16_t dct[64], int comp, uint32_t qbias, int16_t zz[64]);
/* Encode one frame: colour conversion, vertical flip (mjpegenc.c:454-472), fdct, quantise,
 * Huffman (mjpegenc.c:357-435), stuffing + FF escaping + EOI (:282-355), SOI only (:201-204).
 * src: RGB24 (bgr=0) or BGR24 (bgr=1), top-down, stride given.  out must hold
 * amvo_encode_bound(w,h) bytes.  coef_out optional (nmcu*6*64, zig-zag order, not predicted).
 * returns chunk length, or -1 on bad size. */
int amvo_encode_frame(const uint8_t *src, uint32_t src_stride, uint32_t w, uint32_t h, int bgr,
                      uint32_t qbias, uint8_t *out, int16_t *coef_out);
uint32_t amvo_encode_bound(uint32_t w, uint32_t h);
/* the same from planes: YUVJ420P as amv_encoder takes it (mjpegenc.c:493), and YUVJ422P by the product's rule (the two chroma
 * rows over a 4:2:0 sample averaged, rounding up -- amvo_yuv422_to_420; the reference's own 4:2:2 scan has eight blocks per
 * MCU, mjpegenc.c:437-450, which no AMV decoder reads).  out: amvo_encode_bound(w,h) bytes; returns the chunk's length. */
int amvo_encode_frame_yuv420(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint32_t y_stride, uint32_t c_stride,
                             uint32_t w, uint32_t h, uint32_t qbias, uint8_t *out);
void amvo_yuv422_to_420(const uint8_t *c422, uint32_t stride422, uint32_t cw, uint32_t h, uint8_t *c420, uint32_t stride420);
int amvo_encode_frame_yuv422(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint32_t y_stride, uint32_t c_stride,
                             uint32_t w, uint32_t h, uint32_t qbias, uint8_t *out);

/* ---- synthetic sources, integer-only, seeded ---------------- */
void amvo_synth_frame(uint32_t seed, uint32_t frame, uint32_t w, uint32_t h, uint8_t *rgb /* w*h*3, RGB24 */);
void amvo_synth_audio(uint32_t seed, uint64_t first_sample, uint32_t n, int16_t *pcm);

/* ---- helpers -------------------------------------------------------------------- */
uint64_t amvo_fnv1a64(uint64_t h, const uint8_t *p, size_t n); /* start with AMVO_FNV_BASIS */
#define AMVO_FNV_BASIS 0xcbf29ce484222325ull
double amvo_psnr(const uint8_t *a, const uint8_t *b, size_t n);

/* batch drivers used by bench.py's cpu_baseline leg and by fixtures (optionally OpenMP) */
int amvo_decode_batch(const uint8_t *blob, const uint64_t *offs, const uint32_t *lens, uint32_t n,
                      uint32_t w, uint32_t h, uint32_t flags, uint8_t *out, int32_t *status, int threads);
int amvo_synth_encode_batch(uint32_t seed, uint32_t first_frame, uint32_t n, uint32_t w, uint32_t h,
                            uint32_t qbias, uint8_t *blob, uint64_t cap, uint64_t *offs, uint32_t *lens, int threads);

#ifdef __cplusplus
}
#endif
#endif
