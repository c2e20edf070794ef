/*
 * ref_harness_imgconvert.c -- entry point INTO the reference's colour conversion.  TEST INFRASTRUCTURE ONLY.
 *
 * rgb24_to_yuvj420p (libavcodec/imgconvert_template.h:654-..., instantiated by imgconvert.c:1686-1707 with the macros of
 * colorspace.h:30-97) is a static function: the only way to call the reference's own code is to
 * compile imgconvert.c as part of this translation unit, from where it lies (the #include below names the reference's
 * file; nothing is copied).  Everything else of imgconvert.c -- img_convert, the other converters, their use of
 * ff_cropTbl / avcodec_check_dimensions from files that need ./configure -- is unreachable from the one exported
 * function and is dropped by the link (--gc-sections, checked with -z defs).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "imgconvert.c"

__attribute__((visibility("default")))
void amvref_rgb24_to_yuvj420p(const uint8_t *rgb, int stride, int w, int h, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    AVPicture src, dst;
    memset(&src, 0, sizeof src);
    memset(&dst, 0, sizeof dst);
    src.data[0] = (uint8_t *)rgb;
    src.linesize[0] = stride;
    dst.data[0] = y;
    dst.data[1] = cb;
    dst.data[2] = cr;
    dst.linesize[0] = w;
    dst.linesize[1] = dst.linesize[2] = w / 2;
    rgb24_to_yuvj420p(&dst, &src, w, h);
}
