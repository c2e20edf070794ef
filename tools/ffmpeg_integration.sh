#!/bin/bash
# The AVCodec plugin (amv-codec-tools_amd/host/amvhip_lavc.c) linked into the reference's OWN ffmpeg binary and run
# through its `make test` recipe (AMVmuxer/Makefile:15-17) -- on a machine WITHOUT a GPU: the device half is
# tests/c/host_stub.c ("zeros of the right size"), so pixels and nibbles mean nothing; what this proves is the
# reference-side integration of INTEGRATION.md section 3:
#   * the patch a maintainer applies (three table renames + one Makefile line) is enough;
#   * the fork links with no duplicate and no missing symbol -- once against the stub device, once against the real
#     libamvhip.so (whose codec open then fails here: there is no CPU fallback);
#   * allcodecs.c:64,255's REGISTER_ENCDEC pick up the plugin's tables (`ffmpeg -formats` lists amv / adpcm_ima_amv, and
#     the tables in the binary are the plugin's: the reference's own are renamed *_cpu and unregistered);
#   * ffmpeg.c:814,1083's call path and the muxer's frame_size hack (libavformat/amvenc.c:276-281) meet the plugin:
#     `ffmpeg -i in.avi -f amv -r 16 -s 160x120 -ac 1 -ar 22050 out.amv` and the decode back both complete.
# The caller (tests/test_abi_and_host.py::test_plugin_inside_the_reference_ffmpeg) then walks out.amv the way
# compare_amv.c:29-97 walks a file.
#
# Nothing of the reference enters this repository or travels anywhere: the tree is copied to a temporary directory,
# patched and built THERE.  Exit code 77 = the reference tree is not here (the GPU box): skipped.
#
# usage: tools/ffmpeg_integration.sh <workdir>      (outputs: <workdir>/out.amv, back.avi, formats.txt, *.log)
set -e
REF=${AMV_REFERENCE:-/root/reference}/AMVmuxer/ffmpeg
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:?usage: ffmpeg_integration.sh <workdir>}
[ -d "$REF" ] || { echo "ffmpeg_integration: $REF is not here: skipped"; exit 77; }
mkdir -p "$W"
W=$(cd "$W" && pwd)
FF=$W/ffmpeg
rm -rf "$FF"
cp -r "$REF" "$FF"
chmod -R u+w "$FF"
cd "$FF"

# ---- the maintainer's patch ---------------------------------------------------------
# 1. the fork's own tables step aside (they stay in their objects under another name; nothing registers them)
sed -i 's/^AVCodec amv_decoder = {/AVCodec amv_decoder_cpu = {/' libavcodec/sp5xdec.c
sed -i 's/^AVCodec amv_encoder = {/AVCodec amv_encoder_cpu = {/' libavcodec/mjpegenc.c
sed -i 's/^ADPCM_CODEC(CODEC_ID_ADPCM_IMA_AMV, adpcm_ima_amv);/ADPCM_CODEC(CODEC_ID_ADPCM_IMA_AMV, adpcm_ima_amv_cpu);/' libavcodec/adpcm.c
grep -q 'amv_decoder_cpu' libavcodec/sp5xdec.c && grep -q 'amv_encoder_cpu' libavcodec/mjpegenc.c && grep -q 'adpcm_ima_amv_cpu' libavcodec/adpcm.c
# 2. the plugin object joins libavcodec (common.mak: `$(AR) rc $@ $^ $(EXTRAOBJS)`); its device half comes from EXTRALIBS
echo 'EXTRAOBJS += $(AMVHIP_OBJS)' >> libavcodec/Makefile

# ---- configure as SURVEY.md section 8c says --------------------------------------------------------------------
./configure --disable-mmx --disable-network --disable-zlib --disable-vhook --disable-ffserver --disable-ffplay \
    --disable-debug --extra-cflags="-fgnu89-inline -w" > "$W/configure.log" 2>&1
echo 'EXTRALIBS += $(AMVHIP_LIBS)' >> config.mak       # 3. ... and what the plugin needs at link time (-lamvhip)

# ---- the plugin, against the fork's own avcodec.h where the COPY lies ------------------------------------------
mkdir -p "$W/obj"
gcc -O2 -fPIC -Wall -Wextra -Wno-unused-parameter -std=gnu11 -I"$FF/libavcodec" -I"$FF/libavutil" -I"$FF" \
    -c "$ROOT/amv-codec-tools_amd/host/amvhip_lavc.c" -o "$W/obj/amvhip_lavc.o"
for f in amvlib_compat amv_container; do
    gcc -O2 -fPIC -Wall -Wextra -std=gnu11 -c "$ROOT/amv-codec-tools_amd/host/$f.c" -o "$W/obj/$f.o"
done
gcc -O2 -fPIC -Wall -Wextra -std=gnu11 -I"$ROOT/include" -c "$ROOT/tests/c/host_stub.c" -o "$W/obj/host_stub.o"

# ---- build A: the device half = the stub (the run below) --------------------------------------------------------
JOBS=${AMV_FFMPEG_JOBS:-8}
make -j"$JOBS" AMVHIP_OBJS="$W/obj/amvhip_lavc.o $W/obj/amvlib_compat.o $W/obj/amv_container.o $W/obj/host_stub.o" \
    AMVHIP_LIBS="-lpthread" > "$W/make.log" 2>&1 || { tail -30 "$W/make.log"; echo "ffmpeg_integration: build failed"; exit 1; }
if grep -qi "multiple definition\|undefined reference" "$W/make.log"; then
    echo "ffmpeg_integration: duplicate or missing symbols"; grep -i "multiple definition\|undefined reference" "$W/make.log" | head; exit 1
fi
cp ffmpeg "$W/ffmpeg_stub"

# whose tables are in the binary?  the plugin's four (defined once), the fork's renamed
nm ffmpeg_g | grep -E ' [DdBbRr] (amv_decoder|amv_encoder|adpcm_ima_amv_decoder|adpcm_ima_amv_encoder)(_cpu)?$' | sort -k3 > "$W/tables.txt"
nm ffmpeg_g | grep -cE ' [Tt] amvhip_video_decode_frame$' > /dev/null

# ---- build B (link only): the device half = the real libamvhip.so ------------------------------------------------
if [ -f "$ROOT/amv-codec-tools_amd/libamvhip.so" ]; then
    rm -f libavcodec/libavcodec.a ffmpeg ffmpeg_g
    make -j"$JOBS" AMVHIP_OBJS="$W/obj/amvhip_lavc.o" \
        AMVHIP_LIBS="-L$ROOT/amv-codec-tools_amd -l:libamvhip.so -Wl,-rpath,$ROOT/amv-codec-tools_amd" > "$W/make_real.log" 2>&1 \
        || { tail -30 "$W/make_real.log"; echo "ffmpeg_integration: link against libamvhip.so failed"; exit 1; }
    cp ffmpeg "$W/ffmpeg_real"
fi

# ---- the recipe of AMVmuxer/Makefile:15-17 ----------------------------------------------------------------------
cd "$W"
export HOST_STUB_MODE=zero
./ffmpeg_stub -formats > formats.txt 2>&1
# a synthetic source the fork's own muxers hold: 2.5 s of 352x288 at 25 fps + stereo 44.1 kHz PCM, in an AVI
python3 - <<'EOF'
import numpy as np
rng = np.random.default_rng(0xA11CE)
w, h, n = 352, 288, 63
with open("in.yuv", "wb") as f:
    for t in range(n):
        y = ((np.arange(w)[None, :] + 2 * t) ^ (np.arange(h)[:, None] + t)).astype(np.uint8)
        f.write(y.tobytes()); f.write(np.full((h // 2, w // 2), 96 + t, np.uint8).tobytes()); f.write(np.full((h // 2, w // 2), 160 - t, np.uint8).tobytes())
t = np.arange(int(44100 * n / 25))
pcm = (8000 * np.sin(t * 0.05) + rng.integers(-300, 300, t.size)).astype("<i2")
np.stack([pcm, pcm[::-1]], 1).tofile("in.pcm")
EOF
./ffmpeg_stub -f rawvideo -pix_fmt yuv420p -s 352x288 -r 25 -i in.yuv -f s16le -ar 44100 -ac 2 -i in.pcm \
    -vcodec mpeg4 -b 2000k -acodec pcm_s16le -y in.avi > make_input.log 2>&1
./ffmpeg_stub -i in.avi -f amv -r 16 -s 160x120 -ac 1 -ar 22050 -y out.amv > encode.log 2>&1
./ffmpeg_stub -i out.amv -y back.avi > decode.log 2>&1
if [ -x ./ffmpeg_real ]; then
    ./ffmpeg_real -formats > formats_real.txt 2>&1
    # no GPU here: the real library's amvhip_create fails, the plugin's init fails, ffmpeg says so -- and never falls back
    ./ffmpeg_real -i in.avi -f amv -r 16 -s 160x120 -ac 1 -ar 22050 -y out_real.amv > encode_real.log 2>&1 && echo "opened" > real_opened.txt || true
fi
rm -f in.yuv in.pcm
echo "ffmpeg_integration: ok"
