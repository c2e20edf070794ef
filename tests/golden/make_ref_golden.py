"""Regenerates tests/golden/ref_adpcm_stereo.json from the REFERENCE's own AdpcmIma.c object (oracle/_ref/libamvref.so, compiled
by oracle/Makefile from /root/reference where it lies): what AdpcmImaDecodeFrame with channel == 2 (AdpcmIma.c:222-237) makes
of seeded inputs -- FNV-1a-64 of the samples, the length it reports and both channels' end states, for two calls in a row per
input (the context's state carries).  The GPU test test_amvlib_adpcm_stereo_decode rebuilds the same inputs from the same seed
and compares with these values, so the compiled reference object does not have to be loaded on the GPU box.  Data only: no text
of the reference is stored.  Run here (needs /root/reference), from the repo root:
    python tests/golden/make_ref_golden.py
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SEED, SIZES = 23, (8, 16, 689, 690, 1000, 1378, 5)
START = ((-1234, 17), (30000, 80))      # predictor, step index of the two channels before the first call


def inputs():
    rng = np.random.default_rng(SEED)
    return [rng.integers(0, 256, size, dtype=np.uint8) for size in SIZES]


def compute():
    orc = entry.load_oracle()
    orc.build()
    R = orc.ref()
    assert R is not None, "oracle/_ref/libamvref.so is not built (needs /root/reference)"
    cases = []
    for buf in inputs():
        size = int(buf.size)
        n8 = (size + 7) & ~7
        ref = orc.RefADPCMContext()
        ref.channel = 2
        for ch in (0, 1):
            ref.status[ch].predictor, ref.status[ch].step_index = START[ch]
        calls = []
        for call in range(2):
            theirs = np.zeros(2 * n8 + 16, np.int16)
            dl = ctypes.c_int(0)
            pad = np.concatenate([buf, np.zeros(16, np.uint8)])        # the reference reads past buf_size: give it zeros
            rc = R.AdpcmImaDecodeFrame(ctypes.byref(ref), theirs.ctypes.data, ctypes.byref(dl), pad.ctypes.data, size)
            calls.append({"rc": int(rc), "declen": int(dl.value),
                          "pcm_fnv": "%016x" % orc.fnv1a64(orc.FNV_BASIS, theirs[: 2 * n8].view(np.uint8)),
                          "end": [[int(ref.status[ch].predictor), int(ref.status[ch].step_index)] for ch in (0, 1)]})
        cases.append({"size": size, "calls": calls})
    return {"what": "reference AdpcmIma.c AdpcmImaDecodeFrame, channel = 2", "seed": SEED, "start": [list(x) for x in START], "cases": cases}


if __name__ == "__main__":
    out = compute()
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "ref_adpcm_stereo.json"), "w"), indent=1)
    print("wrote", len(out["cases"]), "cases")
