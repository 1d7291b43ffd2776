#!/usr/bin/env python3
"""Debug aid: the 64-bit-rows builder forced onto sizes the 32-bit builder takes too; compares the layouts."""
import os
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch  # noqa: F401
import fmsi_amd as fm

for n in [int(float(x)) for x in sys.argv[1:]]:
    t0 = time.time()
    os.environ["FMSI_HIP_FORCE_WIDE_BUILD"] = "1"
    try:
        with fm.builder_chain(0):
            w = fm.DeviceIndex.build_synthetic(n, 99, 31)
    except fm.FmsiError as e:
        print(n, "wide build failed:", e, flush=True)
        continue
    finally:
        os.environ.pop("FMSI_HIP_FORCE_WIDE_BUILD")
    t1 = time.time()
    print(n, "wide build", round(t1 - t0, 2), "s rows64", w.info.rows64, flush=True)
    if n < (1 << 32) - 8192:
        with fm.builder_chain(0):
            a = fm.DeviceIndex.build_synthetic(n, 99, 31)
        bw, mw = w.download_layout()
        ba, ma = a.download_layout()
        print("   layouts equal:", bool(np.array_equal(bw, ba) and np.array_equal(mw, ma)), "scalars", list(w.info.counts) == list(a.info.counts), w.info.dollar_position == a.info.dollar_position, flush=True)
        del a, ba, ma, bw, mw
    del w
