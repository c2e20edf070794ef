"""Build libamvhip.so (HIP kernels + C ABI + amvlib call surface) in-tree for gfx950, and -- where the reference
tree is present -- libamvhip_lavc.so, the FFmpeg `AVCodec` plugin surface (host/amvhip_lavc.c), which is compiled
against the reference's own libavcodec/avcodec.h where it lies (the binding a maintainer adds to the patched FFmpeg;
on the GPU box, which has no /root/reference, the prebuilt library that travelled with the snapshot is used).

    python amv-codec-tools_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Objects go to amv-codec-tools_amd/build/, the library to
amv-codec-tools_amd/libamvhip.so (git-ignored, but it travels with the gpurun snapshot).
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "libamvhip.so")
LAVC_OUT = os.path.join(HERE, "libamvhip_lavc.so")
LAVC_HOST = os.path.join(ROOT, "tests", "c", "_bin", "lavc_host")     # the C program of tests/c/lavc_host.c
REF_FFMPEG = "/root/reference/AMVmuxer/ffmpeg"
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"

HIP_SOURCES = ["csrc/amv_decode.hip", "csrc/amv_decode_sync.hip", "csrc/amv_reconstruct.hip", "csrc/amv_reconstruct_ff.hip", "csrc/amv_encode.hip", "csrc/amv_encode_par.hip", "csrc/amv_resample.hip", "csrc/amv_adpcm.hip", "csrc/amv_synth.hip", "csrc/amvhip_api.hip"]
C_SOURCES = ["host/amvlib_compat.c", "host/amv_container.c"]
HEADERS = ["csrc/amv_tables.h", "csrc/amv_kernels.h", "csrc/amv_block_load.h", "csrc/amv_piece_map.h", "csrc/amv_encode_common.h", "../include/amvhip.h"]

# -fwrapv: the codec's integer pipeline is defined on two's-complement wrap (see amv_decode.hip)
HIPFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fwrapv", "-fno-strict-aliasing", f"--offload-arch={ARCH}",
            "-Wall", "-Wno-unused-function"]
CFLAGS = ["-O2", "-fPIC", "-Wall", "-Wextra", "-std=gnu11"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libamvhip.so cannot be built (there is no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    p = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), p.stdout, p.stderr))
    return p.stderr


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src in HIP_SOURCES:
        o = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(o)
        if force or _stale(o, [os.path.join(HERE, src)] + hdrs):
            jobs.append([hipcc] + HIPFLAGS + ["-c", src, "-o", o])
    for src in C_SOURCES:
        o = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(o)
        if force or _stale(o, [os.path.join(HERE, src)] + hdrs):
            jobs.append(["gcc"] + CFLAGS + ["-c", src, "-o", o])
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
            for warn in ex.map(_run, jobs):
                if verbose and warn:
                    sys.stderr.write(warn)
    if jobs or force or _stale(OUT, objs):
        _run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs + ["-lpthread"])
    build_lavc(force)
    return OUT


def build_lavc(force=False):
    """the AVCodec plugin + the C host that drives it, against the reference's avcodec.h (only where it exists)"""
    if not os.path.isdir(REF_FFMPEG):
        return None
    inc = ["-I" + os.path.join(REF_FFMPEG, "libavcodec"), "-I" + os.path.join(REF_FFMPEG, "libavutil")]
    src = os.path.join(HERE, "host", "amvhip_lavc.c")
    hdr = os.path.join(ROOT, "include", "amvhip.h")
    if force or _stale(LAVC_OUT, [src, hdr, OUT, os.path.abspath(__file__)]):
        _run(["gcc", "-O2", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-std=gnu11", "-shared"] + inc +
             [src, "-o", LAVC_OUT, "-L" + HERE, "-l:libamvhip.so", "-Wl,-rpath,$ORIGIN", "-Wl,-z,defs"])
    host_src = os.path.join(ROOT, "tests", "c", "lavc_host.c")
    if force or _stale(LAVC_HOST, [host_src, hdr, LAVC_OUT]):
        os.makedirs(os.path.dirname(LAVC_HOST), exist_ok=True)
        _run(["gcc", "-O2", "-Wall", "-std=gnu11"] + inc + ["-I" + os.path.join(ROOT, "include"), host_src, "-o", LAVC_HOST,
              "-L" + HERE, "-l:libamvhip_lavc.so", "-l:libamvhip.so", "-Wl,-rpath,$ORIGIN/../../../amv-codec-tools_amd"])
    return LAVC_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
