#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/collect_profiles.py r02
"""
import glob
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
for name in ("bench", "bench_encode", "bench_coresident", "bench_adpcm", "bench_amvlib", "bench_decode320", "bench_decode10k",
             "bench_strong_world1", "bench_mixed", "bench_amv1", "bench_torchrun_world1", "bench_decode1250"):
    p = os.path.join(src, "%s_%s.json" % (tag, name))
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "%s_%s.json" % (tag, name)))
for name in ("rehearsal_two_ranks_one_gpu_gloo", "rehearsal_two_ranks_one_gpu_gloo_torchrun"):
    p = os.path.join(src, "%s_%s.json" % (tag, name))
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "%s_%s.json" % (tag, name)))
for kind in ("stats", "stats_encode", "stats_coresident", "stats_adpcm"):
    hits = glob.glob(os.path.join(src, "%s_%s" % (tag, kind), "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, kind.replace("stats", "bench").replace("bench_", "bench_") if kind != "stats" else "bench")))
for suffix in ("", "_encode", "_decode320", "_adpcm"):
    f, w = os.path.join(src, "%s_pmc_fetch%s" % (tag, suffix)), os.path.join(src, "%s_pmc_write%s" % (tag, suffix))
    if os.path.isdir(f) and os.path.isdir(w):
        subprocess.run([sys.executable, os.path.join(root, "tools", "summarize_pmc.py"), f, w, os.path.join(dst, tag + suffix)], check=True,
                       stdout=subprocess.DEVNULL)
        # summarize_pmc names its outputs <prefix>_traffic.json / <prefix>_pmc_*.csv
        if suffix:
            os.replace(os.path.join(dst, tag + suffix + "_traffic.json"), os.path.join(dst, "%s_traffic%s.json" % (tag, suffix)))
sq = [os.path.join(src, "%ssq_pmc%d" % (tag, i)) for i in (1, 2, 3)]
if all(os.path.isdir(d) for d in sq):
    with open(os.path.join(dst, "%s_sq_counters.json" % tag), "w") as out:
        subprocess.run([sys.executable, os.path.join(root, "tools", "summarize_sq.py")] + sq, check=True, stdout=out)
for kind, name in (("sqenc", "encode"), ("sqadpcm", "adpcm")):
    sq = [os.path.join(src, "%s%s_pmc%d" % (tag, kind, i)) for i in (1, 2, 3)]
    if all(os.path.isdir(d) for d in sq):
        with open(os.path.join(dst, "%s_sq_counters_%s.json" % (tag, name)), "w") as out:
            subprocess.run([sys.executable, os.path.join(root, "tools", "summarize_sq.py")] + sq, check=True, stdout=out)
for size in ("10k", "1250"):
    npy, js = os.path.join(src, "%s_trace_%s.npy" % (tag, size)), os.path.join(src, "%s_trace_%s.json" % (tag, size))
    if os.path.exists(npy) and os.path.exists(js):
        subprocess.run([sys.executable, os.path.join(root, "tools", "summarize_entropy_trace.py"), npy, js,
                        os.path.join(dst, "%s_entropy_trace_%s.json" % (tag, size))], check=True, stdout=subprocess.DEVNULL)
print("\n".join(sorted(os.listdir(dst))))
