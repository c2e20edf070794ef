"""Regenerates tests/golden/synth_golden.json: hashes of the seeded synthetic sources, of the chunks
the oracle's encoder makes from them and of the frames the oracle's decoder makes from those.
The oracle itself is pinned against the reference in tests/test_oracle_pin.py; these vectors
freeze it so that later edits cannot drift unnoticed.  Run from the repo root:
    python tests/golden/make_golden.py
AMV1.amv in this directory is the reference's own fixture (C-AMVDecoder/bin/AMV1.amv), copied as data.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

orc = entry.load_oracle()
SEED = 0xA11CE
fnv = lambda a: "%016x" % orc.fnv1a64(orc.FNV_BASIS, a)  # noqa: E731

video = []
for (w, h) in ((160, 120), (320, 240), (128, 96), (176, 144)):
    for t, qbias, flags in ((0, 0, 0), (7, 0, 1), (123, 128, 0)):
        src = orc.synth_frame(SEED, t, w, h)
        chunk = orc.encode_frame(src, w, h, qbias=qbias)
        out, st, ok = orc.decode_frame(chunk, w, h, flags)
        assert st == 0
        video.append({"w": w, "h": h, "frame": t, "qbias": qbias, "flags": flags, "rgb_fnv": fnv(src),
                      "chunk_len": len(chunk), "chunk_fnv": fnv(np.frombuffer(chunk, np.uint8)), "bgr_fnv": fnv(out)})
pcm = orc.synth_audio(SEED, 1000, 1378)
chunk, idx = orc.adpcm_encode_chunk(pcm, 5)
audio = {"first": 1000, "n": 1378, "step_in": 5, "step_out": idx, "pcm_fnv": fnv(pcm),
         "chunk_fnv": fnv(np.frombuffer(chunk, np.uint8))}
json.dump({"seed": SEED, "video": video, "audio": audio}, open(os.path.join(ROOT, "tests", "golden", "synth_golden.json"), "w"), indent=1)
print("wrote", len(video), "video cases")
