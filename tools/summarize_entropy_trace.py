#!/usr/bin/env python3
"""tools/time_kernels.py --trace x.npy -> profiles/<tag>.json: what a launch of the several-lanes-per-frame entropy kernel
that is one generation of waves deep lasts as long as.  Per wave (amvhip_entropy_trace): begin / end on the constant-rate
clock, shader clocks of the first walk / the synchronisation rounds / the strict pass, rounds of its worst frame, share
length, workgroup and wave number.  Summarised by wave number inside the workgroup (wave k of a workgroup sits on SIMD
k mod 4: with ten waves per workgroup SIMDs 0 and 1 hold three, and the issue arbiter serves the oldest first) and by rounds.

    python tools/summarize_entropy_trace.py gpurun_out/r06_trace_10k.npy gpurun_out/r06_trace_10k.json profiles/r06_entropy_trace_10k.json
"""
import json
import sys

import numpy as np

npy, tk_json, out_path = sys.argv[1:4]
tr = np.load(npy)
tr = tr[tr[:, 1] != 0]
t0 = int(tr[:, 0].min())
beg = (tr[:, 0].astype(np.int64) - t0) / 100.0
dur = (tr[:, 1] - tr[:, 0]).astype(np.int64) / 100.0
end = beg + dur
wave = ((tr[:, 7] >> np.uint64(16)) & np.uint64(0xffff)).astype(int)
wg = (tr[:, 7] >> np.uint64(32)).astype(int)
rounds = (tr[:, 6] & np.uint64(0xffffffff)).astype(int)
share = (tr[:, 6] >> np.uint64(32)).astype(int)
ph = tr[:, 2:6].astype(np.int64) / 1000.0
res = json.load(open(tk_json))
res.pop("trace", None)
out = {"source": "tools/time_kernels.py --trace (amvhip_entropy_trace), one launch", "time_kernels": res,
       "waves": int(len(tr)), "workgroups": int(len(np.unique(wg))), "waves_per_workgroup": int(np.bincount(wg).max()),
       "launch_us": float(end.max()), "begin_us": {"p50": float(np.median(beg)), "max": float(beg.max())},
       "duration_us": {"p50": float(np.median(dur)), "p99": float(np.percentile(dur, 99)), "max": float(dur.max())},
       "share_bits": {"min": int(share.min()), "p50": int(np.median(share)), "max": int(share.max())},
       "by_wave_number": [], "by_rounds": [], "last_to_end": []}
for w in range(wave.max() + 1):
    m = wave == w
    out["by_wave_number"].append({"wave": w, "simd": w % 4, "n": int(m.sum()), "duration_us": round(float(dur[m].mean()), 1),
                                  "end_us_max": round(float(end[m].max()), 1),
                                  "kclk": {"first_walk": round(float(ph[m, 0].mean())), "rounds": round(float(ph[m, 1].mean())),
                                           "strict": round(float(ph[m, 2].mean())), "per_round": round(float((ph[m, 1] / np.maximum(rounds[m], 1)).mean()))}})
for r in np.unique(rounds):
    m = rounds == r
    out["by_rounds"].append({"rounds": int(r), "waves": int(m.sum()), "duration_us": round(float(dur[m].mean()), 1),
                             "end_us_max": round(float(end[m].max()), 1)})
for i in np.argsort(-end)[:16]:
    out["last_to_end"].append({"end_us": round(float(end[i]), 1), "begin_us": round(float(beg[i]), 1), "wave": int(wave[i]), "rounds": int(rounds[i]),
                               "share_bits": int(share[i]), "kclk": [int(x) for x in ph[i, :3]]})
# what the launch would last if every wave with more rounds than r had needed only r (its rounds phase scaled down)
out["if_rounds_were_capped"] = {}
for cap in (3, 4, 5):
    d2 = dur - np.where(rounds > cap, ph[:, 1] * (1.0 - cap / np.maximum(rounds, 1)) / 2.4, 0.0)    # kclk at 2.4 GHz -> us
    out["if_rounds_were_capped"][str(cap)] = round(float((beg + d2).max()), 1)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("last_to_end",)}, indent=1)[:3000])
