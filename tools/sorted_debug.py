#!/usr/bin/env python3
"""GPU: FMSI_HIP_KERNEL_SORTED against the pair kernel on a golden index, mode by mode; prints where they differ."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import fmsi_amd as fm
from conftest import GOLDEN, index_prefix, mixed_kmers

name, k = "sars_k31", 31
ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_test2_ones.fna.ms.k31") if not l.startswith(">"))
kmers = mixed_kmers(ms, k, int(sys.argv[1]) if len(sys.argv) > 1 else 3000, seed=17)
packed = fm.pack_kmers(kmers, k)
dev = fm.DeviceIndex.from_files(index_prefix(name)).enable_pair_layout()
for label, flags in (("all", fm.MODE_ALL), ("or", fm.MODE_OR), ("all strict", fm.MODE_ALL | fm.STRICT), ("or strict", fm.MODE_OR | fm.STRICT)):
    a = dev.query_kmers(packed, k, flags | fm.KERNEL_PAIR | fm.NO_PREFIX_TABLE)
    b, cn = dev.query_kmers(packed, k, flags | fm.KERNEL_SORTED, counters=True)
    b2 = dev.query_kmers(packed, k, flags | fm.KERNEL_SORTED)
    bad = np.nonzero(a != b)[0]
    print("   sorted twice: same", bool(np.array_equal(b, b2)))
    print(label, "mismatches", bad.size, "of", a.size, "first", bad[:20], "pair", a[bad[:20]], "sorted", b[bad[:20]], "counters", cn)
