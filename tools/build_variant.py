#!/usr/bin/env python3
"""Kernel experiments: build amv-codec-tools_amd/libamvhip_<name>.so from the current objects with ONE HIP
source replaced by a patched copy (select it at run time with AMVHIP_LIB=...).

    python tools/build_variant.py <name> csrc/amv_decode_sync.hip /tmp/patched.hip
"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "amv-codec-tools_amd")
spec = importlib.util.spec_from_file_location("amv_build", os.path.join(PKG, "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

name, src, patched = sys.argv[1:4]
b.build()
tmp = os.path.join(PKG, "csrc", "_variant_%s.hip" % name)
obj = "/tmp/_variant_%s.o" % name
open(tmp, "w").write(open(patched).read())
try:
    subprocess.check_call([b._hipcc()] + b.HIPFLAGS + ["-c", tmp, "-o", obj], cwd=PKG)
finally:
    os.remove(tmp)
objs = [os.path.join(b.OBJ, os.path.basename(s) + ".o") for s in b.HIP_SOURCES + b.C_SOURCES if s != src]
out = os.path.join(PKG, "libamvhip_%s.so" % name)
subprocess.check_call([b._hipcc(), "-shared", "-fPIC", "--offload-arch=%s" % b.ARCH, "-o", out] + objs + [obj, "-lpthread"])
print(out)
