"""TEST INFRASTRUCTURE ONLY -- ctypes access to the CPU oracle (oracle/_build/liboracle.so) and, when
it has been built, to the real reference (oracle/_ref/libfmsi_ref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package fmsi_amd never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_LIB = HERE / "_build" / "liboracle.so"
ORACLE_CLI = HERE / "_build" / "fmsi_oracle"
REF_LIB = HERE / "_ref" / "libfmsi_ref.so"
REF_CLI = HERE / "_ref" / "fmsi_ref"
REF_TUNED_LIB = HERE / "_ref" / "libfmsi_ref_tuned.so"   # same sources, -O3 -DNDEBUG -march=x86-64-v3 (Makefile.ref)
REFERENCE_ROOT = Path("/root/reference")

MODE_OR, MODE_ALL = 0, 1


def build_oracle() -> None:
    res = subprocess.run(["make", "-C", str(HERE)], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + res.stdout + res.stderr)


def build_ref() -> bool:
    """Build oracle/_ref from /root/reference when that tree is present (never on the GPU box)."""
    if not (REFERENCE_ROOT / "src" / "fms_index.h").exists():
        return REF_LIB.exists()
    res = subprocess.run(["make", "-C", str(HERE), "-f", "Makefile.ref"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("reference build failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    return True


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class _Base:
    """Common surface of the oracle and the reference harness (same function shapes, prefix differs)."""

    P = ""

    def __init__(self, lib: C.CDLL, handle):
        self.L = lib
        self.h = handle

    def _f(self, name):
        return getattr(self.L, self.P + name)

    def close(self):
        if self.h:
            self._f("index_free")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n(self):
        return int(self._f("index_n")(self.h))

    @property
    def k(self):
        return int(self._f("index_k")(self.h))

    def predictor_reset(self):
        self._f("predictor_reset")(self.h)

    def rank(self, i, c):
        return int(self._f("rank")(self.h, i, c))

    def rank1(self, which, i):
        return int(self._f("rank1")(self.h, which, i))

    def update_range(self, i, j, c):
        a, b = C.c_uint64(i), C.c_uint64(j)
        self._f("update_range")(self.h, C.byref(a), C.byref(b), c)
        return int(a.value), int(b.value)

    def extend_range_with_klcp(self, i, j):
        a, b = C.c_uint64(i), C.c_uint64(j)
        self._f("extend_range_with_klcp")(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def get_range_with_pattern(self, pattern: str):
        a, b = C.c_uint64(), C.c_uint64()
        p = pattern.encode()
        if self.P == "orc_":
            self.L.orc_get_range_with_pattern(self.h, p, len(p), C.byref(a), C.byref(b))
        else:
            self.L.ref_get_range_with_pattern(self.h, p, len(p), C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def mask_at(self, i):
        return int(self._f("mask_at")(self.h, i))

    def mask_rank(self, i):
        return int(self._f("mask_rank")(self.h, i))

    def kmer_order_if_present(self, i, j):
        return int(self._f("kmer_order_if_present")(self.h, i, j))

    def _text(self, fn, *args, cap=1 << 16):
        while True:
            buf = C.create_string_buffer(cap)
            r = fn(*args, buf, cap)
            if r >= 0:
                return buf.raw[:r].decode()
            cap = -r + 16

    def query_kmers(self, seq: str, k: int, has_klcp=False, mode=MODE_OR, orders=False) -> str:
        s = seq.encode()
        name = "query_kmers_buf" if self.P == "orc_" else "query_kmers"
        return self._text(self._f(name), self.h, s, len(s), k, int(has_klcp), mode, int(orders), cap=max(1 << 12, 24 * len(s)))

    def query_kmers_streaming(self, seq: str, k: int, maximized_ones=False, orders=False) -> str:
        s = seq.encode()
        name = "query_kmers_streaming_buf" if self.P == "orc_" else "query_kmers_streaming"
        return self._text(self._f(name), self.h, s, len(s), k, int(maximized_ones), int(orders), cap=max(1 << 12, 24 * len(s)))

    def query_file(self, path: str, k: int, has_klcp=False, mode=MODE_OR, orders=False, cap=1 << 20):
        """Runs the ms_query record loop over a FASTA/FASTQ file; returns (stdout text, query seconds)."""
        sec = C.c_double()
        name = "query_file_buf" if self.P == "orc_" else "query_file"
        fn = self._f(name)
        while True:
            buf = C.create_string_buffer(cap)
            r = fn(self.h, os.fsencode(path), k, int(has_klcp), mode, int(orders), buf, cap, C.byref(sec))
            if r >= 0:
                return buf.raw[:r].decode(), float(sec.value)
            if r == -(2 ** 63):
                raise FileNotFoundError(path)
            cap = -r + 16

    def time_isolated_kmers(self, kmers: bytes, n: int, k: int, mode=MODE_ALL):
        """query_kmers() per isolated k-mer; returns (seconds, uint8 array of 0/1)."""
        out = C.create_string_buffer(n)
        sec = self._f("time_isolated_kmers")(self.h, kmers, n, k, mode, out)
        return float(sec), np.frombuffer(out.raw, dtype=np.uint8) - ord("0")


def _proto(L, p):
    u64, i32, vp = C.c_uint64, C.c_int, C.c_void_p
    pu64 = C.POINTER(C.c_uint64)
    sig = {
        "index_free": (None, [vp]), "index_n": (u64, [vp]), "index_k": (i32, [vp]),
        "predictor_reset": (None, [vp]), "rank": (u64, [vp, u64, i32]), "rank1": (u64, [vp, i32, u64]),
        "update_range": (None, [vp, pu64, pu64, i32]), "extend_range_with_klcp": (None, [vp, pu64, pu64]),
        "mask_at": (i32, [vp, u64]), "mask_rank": (u64, [vp, u64]), "kmer_order_if_present": (C.c_int64, [vp, u64, u64]),
        "time_isolated_kmers": (C.c_double, [vp, C.c_char_p, u64, i32, i32, C.c_char_p]),
        "index_load": (vp, [C.c_char_p, i32]),
        "index_literal": (vp, [vp, u64, vp, u64, vp, u64, vp, u64, vp, u64, vp, u64, i32]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, p + name)
        f.restype = res
        f.argtypes = args


_oracle_lib = None
_ref_lib = None


def oracle_lib() -> C.CDLL:
    global _oracle_lib
    if _oracle_lib is None:
        if not ORACLE_LIB.exists():
            build_oracle()
        L = C.CDLL(str(ORACLE_LIB))
        _proto(L, "orc_")
        u64, i32, vp = C.c_uint64, C.c_int, C.c_void_p
        L.orc_get_range_with_pattern.argtypes = [vp, C.c_char_p, i32, C.POINTER(u64), C.POINTER(u64)]
        L.orc_get_range_with_pattern.restype = None
        L.orc_query_kmers_buf.argtypes = [vp, C.c_char_p, u64, i32, i32, i32, i32, C.c_char_p, u64]
        L.orc_query_kmers_buf.restype = C.c_int64
        L.orc_query_kmers_streaming_buf.argtypes = [vp, C.c_char_p, u64, i32, i32, i32, C.c_char_p, u64]
        L.orc_query_kmers_streaming_buf.restype = C.c_int64
        L.orc_query_file_buf.argtypes = [vp, C.c_char_p, i32, i32, i32, i32, C.c_char_p, u64, C.POINTER(C.c_double)]
        L.orc_query_file_buf.restype = C.c_int64
        L.orc_index_construct.argtypes = [C.c_char_p, u64, i32, i32]
        L.orc_index_construct.restype = vp
        L.orc_index_dump.argtypes = [vp, C.c_char_p]
        L.orc_index_dump.restype = i32
        L.orc_index_has_klcp.argtypes = [vp]
        L.orc_index_words.argtypes = [vp, i32, C.POINTER(u64)]
        L.orc_index_words.restype = C.POINTER(u64)
        L.orc_index_count.argtypes = [vp, i32]
        L.orc_index_count.restype = u64
        L.orc_index_dollar.argtypes = [vp]
        L.orc_index_dollar.restype = u64
        L.orc_counters_reset.argtypes = [vp]
        L.orc_counters_reset.restype = None
        L.orc_counters_get.argtypes = [vp, C.POINTER(u64)]
        L.orc_counters_get.restype = None
        L.orc_kmer_ternaries.argtypes = [vp, C.c_char_p, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.orc_kmer_ternaries.restype = None
        L.orc_kmers_ternaries_bulk.argtypes = [vp, C.c_char_p, u64, i32, i32, vp, vp, vp, vp]
        L.orc_kmers_ternaries_bulk.restype = None
        L.orc_kmer_general.argtypes = [vp, C.c_char_p, i32, C.POINTER(u64), C.POINTER(u64)]
        L.orc_kmer_general.restype = None
        L.orc_apply_function.argtypes = [i32, u64, u64, u64, u64]
        L.orc_apply_function.restype = i32
        _oracle_lib = L
    return _oracle_lib


def ref_available() -> bool:
    return REF_LIB.exists()


_ref_tuned_lib = None


def ref_tuned_lib() -> C.CDLL:
    """The reference built with tuned flags (timing only; the parity checks use the as-shipped build)."""
    global _ref_tuned_lib
    if _ref_tuned_lib is None:
        L = C.CDLL(str(REF_TUNED_LIB))
        _proto(L, "ref_")
        _ref_tuned_lib = L
    return _ref_tuned_lib


def ref_lib() -> C.CDLL:
    global _ref_lib
    if _ref_lib is None:
        L = C.CDLL(str(REF_LIB))
        _proto(L, "ref_")
        u64, i32, vp = C.c_uint64, C.c_int, C.c_void_p
        L.ref_get_range_with_pattern.argtypes = [vp, C.c_char_p, i32, C.POINTER(u64), C.POINTER(u64)]
        L.ref_get_range_with_pattern.restype = None
        L.ref_query_kmers.argtypes = [vp, C.c_char_p, u64, i32, i32, i32, i32, C.c_char_p, u64]
        L.ref_query_kmers.restype = C.c_int64
        L.ref_query_kmers_streaming.argtypes = [vp, C.c_char_p, u64, i32, i32, i32, C.c_char_p, u64]
        L.ref_query_kmers_streaming.restype = C.c_int64
        L.ref_query_file.argtypes = [vp, C.c_char_p, i32, i32, i32, i32, C.c_char_p, u64, C.POINTER(C.c_double)]
        L.ref_query_file.restype = C.c_int64
        _ref_lib = L
    return _ref_lib


class Oracle(_Base):
    P = "orc_"

    @classmethod
    def load(cls, prefix: str, use_klcp=False):
        L = oracle_lib()
        h = L.orc_index_load(os.fsencode(prefix), int(use_klcp))
        if not h:
            raise FileNotFoundError(f"oracle: cannot load index {prefix}")
        return cls(L, h)

    @classmethod
    def construct(cls, ms: str, k: int, use_klcp=True):
        L = oracle_lib()
        b = ms.encode()
        return cls(L, L.orc_index_construct(b, len(b), k, int(use_klcp)))

    @classmethod
    def literal(cls, ac_gt, ac, gt, mask, counts, dollar, klcp=(), k=0):
        L = oracle_lib()
        arrs = [_u8(x) for x in (ac_gt, ac, gt, mask)]
        kl = _u8(list(klcp))
        cnt = np.asarray(counts, dtype=np.uint64)
        h = L.orc_index_literal(arrs[0].ctypes.data, arrs[0].size, arrs[1].ctypes.data, arrs[1].size, arrs[2].ctypes.data,
                                arrs[2].size, arrs[3].ctypes.data, arrs[3].size, cnt.ctypes.data, dollar, kl.ctypes.data, kl.size, k)
        return cls(L, h)

    def dump(self, prefix: str):
        if self.L.orc_index_dump(self.h, os.fsencode(prefix)) != 0:
            raise OSError(f"oracle: cannot write {prefix}.fmsi.*")

    def has_klcp(self):
        return bool(self.L.orc_index_has_klcp(self.h))

    def words(self, which: int):
        """(uint64 words copy, nbits) of ac_gt(0) / ac(1) / gt(2) / klcp(3)."""
        nb = C.c_uint64()
        p = self.L.orc_index_words(self.h, which, C.byref(nb))
        nw = (nb.value + 63) // 64 + 1
        return np.ctypeslib.as_array(p, shape=(nw,)).copy(), int(nb.value)

    def mask_bits(self) -> np.ndarray:
        return np.array([self.mask_at(i) for i in range(self.n)], dtype=np.uint8)

    def counts(self):
        return [int(self.L.orc_index_count(self.h, c)) for c in range(4)]

    def dollar(self):
        return int(self.L.orc_index_dollar(self.h))

    def counters_reset(self):
        self.L.orc_counters_reset(self.h)

    def counters(self):
        out = (C.c_uint64 * 4)()
        self.L.orc_counters_get(self.h, out)
        return dict(lf_steps=int(out[0]), mask_probes=int(out[1]), klcp_bits=int(out[2]), strand_searches=int(out[3]))

    def kmer_ternaries(self, kmer: str, maximized_ones: bool):
        f, r = C.c_int(), C.c_int()
        b = kmer.encode()
        self.L.orc_kmer_ternaries(self.h, b, len(b), int(maximized_ones), C.byref(f), C.byref(r))
        return int(f.value), int(r.value)


    def ternaries_bulk(self, kmers: bytes, n: int, k: int, maximized_ones: bool, orders: bool = False):
        """Both strands of n isolated k-mers: (fwd int8, rc int8[, ord_fwd int64, ord_rc int64])."""
        fwd = np.zeros(n, dtype=np.int8)
        rc = np.zeros(n, dtype=np.int8)
        of = np.zeros(n, dtype=np.int64) if orders else None
        orc = np.zeros(n, dtype=np.int64) if orders else None
        self.L.orc_kmers_ternaries_bulk(self.h, kmers, n, k, int(maximized_ones), fwd.ctypes.data, rc.ctypes.data,
                                        of.ctypes.data if orders else None, orc.ctypes.data if orders else None)
        return (fwd, rc, of, orc) if orders else (fwd, rc)


    def kmer_general(self, kmer: bytes, k: int):
        """(ones, total) of a k-mer over both strands, a self-complementary k-mer counted once (query_mode::general)."""
        o, t = C.c_uint64(), C.c_uint64()
        self.L.orc_kmer_general(self.h, kmer, k, C.byref(o), C.byref(t))
        return int(o.value), int(t.value)

    def apply_function(self, kind: int, r: int, s: int, ones: int, total: int) -> bool:
        return bool(self.L.orc_apply_function(kind, r, s, ones, total))


class Reference(_Base):
    """The REAL reference (oracle/_ref), same call surface as Oracle where the harness offers it."""

    P = "ref_"

    @classmethod
    def load(cls, prefix: str, use_klcp=False):
        L = ref_lib()
        return cls(L, L.ref_index_load(os.fsencode(prefix), int(use_klcp)))

    @classmethod
    def load_tuned(cls, prefix: str, use_klcp=False):
        L = ref_tuned_lib()
        return cls(L, L.ref_index_load(os.fsencode(prefix), int(use_klcp)))

    @classmethod
    def literal(cls, ac_gt, ac, gt, mask, counts, dollar, klcp=(), k=0):
        L = ref_lib()
        arrs = [_u8(x) for x in (ac_gt, ac, gt, mask)]
        kl = _u8(list(klcp))
        cnt = np.asarray(counts, dtype=np.uint64)
        h = L.ref_index_literal(arrs[0].ctypes.data, arrs[0].size, arrs[1].ctypes.data, arrs[1].size, arrs[2].ctypes.data,
                                arrs[2].size, arrs[3].ctypes.data, arrs[3].size, cnt.ctypes.data, dollar, kl.ctypes.data, kl.size, k)
        return cls(L, h)


def run_cli(binary: Path, args, stdin: bytes | None = None):
    res = subprocess.run([str(binary), *args], input=stdin, capture_output=True)
    return res.returncode, res.stdout, res.stderr
