"""fmsi_amd -- thin Python binding (ctypes) over libfmsi_hip.so, the MI355X `fmsi query` hot path.

The product is the C-ABI shared library (include/fmsi_hip.h) and the `fmsi` CLI built from
fmsi_amd/csrc; this module only mirrors the library for test drivers and bench.py.  There is no CPU
fallback anywhere in this package: if the library is missing or no HIP device is usable the calls
raise.  Nothing here imports or calls oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
REPO = _PKG.parent
LIB_PATH = _PKG / "lib" / "libfmsi_hip.so"
CLI_PATH = _PKG / "bin" / "fmsi"

# ---- constants of include/fmsi_hip.h ------------------------------------------------------------
MODE_OR = 0x0
MODE_ALL = 0x1
STRICT = 0x2
ORDERS = 0x4
DEVICE_PTRS = 0x8
KERNEL_LOCKSTEP = 0x10
KERNEL_REFILL = 0x20
KERNEL_PAIR = 0x30
KERNEL_PHASED = 0x40
KERNEL_DICT = 0x50
KERNEL_SORTED = 0x60
LOAD_KLCP, LOAD_ASYNC_TABLE = 0x1, 0x2
DICT_SPARSE = 0x1
DICT_CHAIN = 0x2
DICT_SUPERKMER = 0x4
DICT_SUPERKMER_ONLY = 0x8
NO_PREFIX_TABLE = 0x100
OUT_BITS = 0x200
NO_DICT_CHAIN = 0x400
NO_SUPERKMER = 0x800
FN_AND, FN_XOR, FN_RANGE = 1, 2, 3   # demasking functions of query_runs_fn (`-f and|xor|r-s`)
N_COUNTERS = 8

E_INVAL, E_NODEV, E_HIP, E_NOMEM, E_IO, E_RANGE, E_COMM = -1, -2, -3, -4, -5, -6, -7


class FmsiError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libfmsi_hip error {code}: {message}")
        self.code = code


class IndexView(C.Structure):
    _fields_ = [
        ("ac_gt", C.c_void_p), ("ac_gt_bits", C.c_uint64),
        ("ac", C.c_void_p), ("ac_bits", C.c_uint64),
        ("gt", C.c_void_p), ("gt_bits", C.c_uint64),
        ("mask", C.c_void_p), ("mask_bits", C.c_uint64),
        ("klcp", C.c_void_p), ("klcp_bits", C.c_uint64),
        ("counts", C.c_uint64 * 4),
        ("dollar_position", C.c_uint64),
        ("k", C.c_int32),
        ("prefix_len", C.c_int32),
    ]


class IndexInfo(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("k", C.c_int32), ("has_klcp", C.c_int32), ("device", C.c_int32),
        ("prefix_len", C.c_int32), ("block_bytes", C.c_uint64), ("block_symbols", C.c_uint64),
        ("n_blocks", C.c_uint64), ("device_bytes", C.c_uint64), ("counts", C.c_uint64 * 4), ("dollar_position", C.c_uint64),
        ("candidate_entry_bytes", C.c_uint64), ("dict_bytes", C.c_uint64), ("dict_keys", C.c_uint64 * 4), ("dict_ambiguous", C.c_uint64),
        ("dict_levels", C.c_uint32), ("dict_log2_buckets", C.c_uint32), ("dict_chain_bytes", C.c_uint64), ("rows64", C.c_uint64),
        ("sk_bytes", C.c_uint64), ("sk_records", C.c_uint64), ("sk_heavy_minimizers", C.c_uint64), ("sk_overflowed_buckets", C.c_uint64),
        ("sk_aux_keys", C.c_uint64), ("sk_log2_buckets", C.c_uint32), ("sk_minimizer_len", C.c_uint32),
    ]


class SyntheticSpec(C.Structure):
    _fields_ = [
        ("len", C.c_uint64), ("seed", C.c_uint64), ("k", C.c_int32), ("with_klcp", C.c_int32), ("prefix_len", C.c_int32),
        ("repeat_permille", C.c_uint32), ("families", C.c_uint32), ("divergence_permille", C.c_uint32), ("mask_permille", C.c_uint32),
    ]


def build(verbose: bool = False) -> None:
    """Compile libfmsi_hip.so and the fmsi CLI for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(_PKG / "csrc")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libfmsi_hip.so failed")


_lib = None


def lib() -> C.CDLL:
    """Load libfmsi_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                f"(or make -C fmsi_amd/csrc); there is no fallback path")
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7,
    # the SONAME libfmsi_hip.so needs).  Importing torch first makes the loader reuse that copy for us; the
    # other order would load the system runtime here and torch's copy later, and the second one finds no GPU.
    # (The C++ CLI never loads torch and uses the system runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(str(LIB_PATH))
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32
    L.fmsi_hip_last_error.restype = C.c_char_p
    L.fmsi_hip_abi_version.restype = i32
    L.fmsi_hip_device_count.restype = i32
    L.fmsi_hip_index_create.argtypes = [C.POINTER(IndexView), i32, C.POINTER(vp)]
    L.fmsi_hip_index_destroy.argtypes = [vp]
    L.fmsi_hip_index_destroy.restype = None
    L.fmsi_hip_index_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    L.fmsi_hip_index_download_layout.argtypes = [vp, vp, vp]
    L.fmsi_hip_index_build.argtypes = [C.c_char_p, u64, i32, i32, i32, i32, C.POINTER(vp)]
    L.fmsi_hip_host_alloc.argtypes = [u64]
    L.fmsi_hip_host_alloc.restype = vp
    L.fmsi_hip_host_free.argtypes = [vp]
    L.fmsi_hip_host_free.restype = None
    L.fmsi_hip_index_build_synthetic.argtypes = [u64, u64, i32, i32, i32, i32, C.POINTER(vp)]
    L.fmsi_hip_index_build_synthetic_repeats.argtypes = [u64, u64, i32, i32, i32, i32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.fmsi_hip_index_build_synthetic_spec.argtypes = [C.POINTER(SyntheticSpec), i32, C.POINTER(vp)]
    L.fmsi_hip_synthetic_mask_bit.argtypes = [u64, u64, C.c_uint32]
    L.fmsi_hip_synthetic_mask_bit.restype = C.c_uint32
    L.fmsi_hip_synthetic_text_word_repeats.argtypes = [u64, u64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.fmsi_hip_synthetic_text_word_repeats.restype = u64
    L.fmsi_hip_synthetic_text_word.argtypes = [u64, u64]
    L.fmsi_hip_synthetic_text_word.restype = u64
    L.fmsi_hip_synthetic_queries.argtypes = [vp, i32, u64, u64, u64, vp, vp]
    L.fmsi_hip_index_download_text.argtypes = [vp, vp]
    L.fmsi_hip_index_download_klcp.argtypes = [vp, vp]
    L.fmsi_hip_index_enable_pair_layout.argtypes = [vp]
    L.fmsi_hip_index_enable_verify.argtypes = [vp]
    L.fmsi_hip_index_enable_dict.argtypes = [vp, u32]
    L.fmsi_hip_index_load.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
    L.fmsi_hip_index_wait_background.argtypes = [vp]
    L.fmsi_hip_index_release_scratch.argtypes = [vp]
    L.fmsi_hip_trim_scratch.argtypes = [i32]
    L.fmsi_hip_keep_scratch.argtypes = [i32]
    L.fmsi_hip_keep_scratch.restype = None
    L.fmsi_hip_scratch_bytes.argtypes = [i32]
    L.fmsi_hip_scratch_bytes.restype = u64
    L.fmsi_hip_synthetic_reads.argtypes = [vp, u64, u64, u64, u64, u32, vp, vp]
    L.fmsi_hip_query_kmers.argtypes = [vp, vp, u64, i32, u32, vp, vp, vp]
    L.fmsi_hip_query_windows.argtypes = [vp, vp, u64, vp, u64, i32, u32, vp, vp, vp]
    L.fmsi_hip_query_stream.argtypes = [vp, vp, u64, vp, vp, u64, u64, i32, u32, vp, vp, vp]
    L.fmsi_hip_query_runs.argtypes = [vp, vp, u64, vp, vp, u64, u64, i32, u32, vp, vp, vp]
    L.fmsi_hip_query_runs_fn.restype = i32
    L.fmsi_hip_query_runs_fn.argtypes = [vp, vp, vp, u64, vp, vp, u64, u64, i32, i32, u64, u64, u32, vp, vp, vp]
    L.fmsi_hip_sync.argtypes = [vp, vp]
    L.fmsi_hip_profile_enable.argtypes = [vp, i32]
    L.fmsi_hip_profile_read.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_double), i32]
    L.fmsi_host_index_load.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.fmsi_host_index_free.argtypes = [vp]
    L.fmsi_host_index_free.restype = None
    L.fmsi_host_index_view.argtypes = [vp, C.POINTER(IndexView)]
    L.fmsi_host_layout_blocks.argtypes = [C.POINTER(IndexView)]
    L.fmsi_host_layout_blocks.restype = u64
    L.fmsi_host_layout_build.argtypes = [C.POINTER(IndexView), vp, vp]
    L.fmsi_host_pack_2bit.argtypes = [C.c_char_p, u64, vp]
    L.fmsi_host_pack_2bit.restype = u64
    L.fmsi_cli_query.argtypes = [i32, C.POINTER(C.c_char_p), i32]
    L.fmsi_cli_index.argtypes = [i32, C.POINTER(C.c_char_p)]
    L.fmsi_cli_export.argtypes = [i32, C.POINTER(C.c_char_p)]
    L.fmsi_hip_index_export.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.fmsi_host_index_write.argtypes = [C.c_char_p, C.POINTER(IndexView)]
    L.fmsi_hip_index_write_files.argtypes = [vp, C.c_char_p]
    L.fmsi_hip_comm_unique_id.argtypes = [vp]
    L.fmsi_hip_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.fmsi_hip_comm_destroy.argtypes = [vp]
    L.fmsi_hip_comm_destroy.restype = None
    L.fmsi_hip_gather.argtypes = [vp, vp, u64, vp, vp, i32, vp]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise FmsiError(rc, lib().fmsi_hip_last_error().decode(errors="replace"))


def device_count() -> int:
    return lib().fmsi_hip_device_count()


_CODE = np.full(256, 4, dtype=np.uint8)
for _ch, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3)):
    _CODE[ord(_ch)] = _v
    _CODE[ord(_ch.lower())] = _v


def pack_kmers(kmers, k: int) -> np.ndarray:
    """2-bit pack equal-length k-mers (k <= 32): base t (0 = leftmost) in bits [2t, 2t+1]."""
    if isinstance(kmers, (bytes, bytearray)):
        raw = np.frombuffer(kmers, dtype=np.uint8)
    elif isinstance(kmers, np.ndarray):
        raw = kmers.reshape(-1).view(np.uint8)
    else:
        raw = np.frombuffer("".join(kmers).encode(), dtype=np.uint8)
    assert k <= 32 and raw.size % k == 0
    codes = _CODE[raw].reshape(-1, k).astype(np.uint64)
    if (codes > 3).any():
        raise ValueError("invalid character in k-mer")
    shifts = (2 * np.arange(k, dtype=np.uint64))
    return np.bitwise_or.reduce(codes << shifts, axis=1).astype(np.uint64)


def pack_sequence(seq: bytes) -> np.ndarray:
    """2-bit pack a sequence: base p in bits [2(p%32), 2(p%32)+1] of word p//32 (+2 pad words)."""
    words = np.zeros(len(seq) // 32 + 2, dtype=np.uint64)
    bad = lib().fmsi_host_pack_2bit(seq, len(seq), words.ctypes.data)
    if bad != len(seq):
        raise ValueError(f"invalid character at {bad}")
    return words


def lower_case_bits(seq: bytes) -> np.ndarray:
    """One bit per character of a sequence, set where it is lower case (bit p & 63 of word p >> 6): query_runs_fn's lower_case."""
    a = np.frombuffer(seq, dtype=np.uint8)
    bits = np.zeros((len(seq) // 64 + 1) * 64, dtype=np.uint8)
    bits[:len(seq)] = (a >= ord("a")) & (a <= ord("z"))
    return np.packbits(bits, bitorder="little").view(np.uint64)


class HostIndex:
    """Parsed <prefix>.fmsi.* files (loader half of load_index, reference src/fms_index.h:502-526)."""

    def __init__(self, prefix: str, use_klcp: bool = False):
        self._h = C.c_void_p()
        _check(lib().fmsi_host_index_load(os.fsencode(prefix), int(use_klcp), C.byref(self._h)))
        self.view = IndexView()
        _check(lib().fmsi_host_index_view(self._h, C.byref(self.view)))

    @property
    def n(self) -> int:
        return int(self.view.mask_bits)

    @property
    def k(self) -> int:
        return int(self.view.k)

    def bits(self, which: str) -> np.ndarray:
        """Copy of one bit-vector as a uint8 array of 0/1 (test helper)."""
        ptr = getattr(self.view, which)
        nbits = int(getattr(self.view, which + "_bits"))
        if not ptr or nbits == 0:
            return np.zeros(0, dtype=np.uint8)
        nw = (nbits + 63) // 64
        words = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(nw,)).copy()
        return np.unpackbits(words.view(np.uint8), bitorder="little")[:nbits]

    def write(self, prefix: str) -> None:
        """Write <prefix>.fmsi.* again (dump_index, reference src/fms_index.h:484-500); the mask is re-encoded as RRR-63."""
        _check(lib().fmsi_host_index_write(os.fsencode(prefix), C.byref(self.view)))

    def layout(self):
        """(blocks[n_blocks,16] u32, mrank[n_blocks+1] u32): the device layout, built on the host."""
        nb = int(lib().fmsi_host_layout_blocks(C.byref(self.view)))
        blocks = np.zeros((nb, 16), dtype=np.uint32)
        mrank = np.zeros(nb + 1, dtype=np.uint32)
        _check(lib().fmsi_host_layout_build(C.byref(self.view), blocks.ctypes.data, mrank.ctypes.data))
        return blocks, mrank

    def close(self):
        if self._h:
            lib().fmsi_host_index_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def view_from_arrays(ac_gt, ac, gt, mask, counts, dollar_position, k, klcp=None, prefix_len=0):
    """IndexView over numpy uint64 word arrays given as (words, nbits) pairs.  Returns (view, keepalive)."""
    v = IndexView()
    keep = []
    for name, pair in (("ac_gt", ac_gt), ("ac", ac), ("gt", gt), ("mask", mask), ("klcp", klcp)):
        if pair is None:
            continue
        words, nbits = pair
        words = np.ascontiguousarray(words, dtype=np.uint64)
        keep.append(words)
        setattr(v, name, words.ctypes.data)
        setattr(v, name + "_bits", int(nbits))
    for i in range(4):
        v.counts[i] = int(counts[i])
    v.dollar_position = int(dollar_position)
    v.k = int(k)
    v.prefix_len = int(prefix_len)
    return v, keep


class DeviceIndex:
    """Device-resident index (fmsi_hip_index).  Raises FmsiError(E_NODEV) when no GPU is usable."""

    def __init__(self, view: IndexView | None, device: int = 0, keepalive=None, _handle=None):
        self._h = C.c_void_p()
        self._keep = keepalive
        if _handle is not None:
            self._h = _handle
        else:
            _check(lib().fmsi_hip_index_create(C.byref(view), device, C.byref(self._h)))
        self.info = IndexInfo()
        _check(lib().fmsi_hip_index_get_info(self._h, C.byref(self.info)))

    @classmethod
    def build(cls, ms: bytes, k: int, with_klcp: bool = False, device: int = 0, prefix_len: int = 0):
        """On-device construction from a masked superstring (upper case = ON)."""
        h = C.c_void_p()
        _check(lib().fmsi_hip_index_build(ms, len(ms), k, int(with_klcp), device, prefix_len, C.byref(h)))
        return cls(None, device, _handle=h)

    @classmethod
    def build_synthetic(cls, length: int, seed: int, k: int, with_klcp: bool = False, device: int = 0, prefix_len: int = 0,
                        repeats=(0, 0, 0), mask_permille: int = 1000):
        """Synthetic superstring generated and indexed in HBM (see include/fmsi_hip.h).
        repeats = (repeat_permille, families, divergence_permille): share of 2048-base segments that are diverged copies of
        one of `families` consensus segments; mask_permille < 1000: a mixed mask (fmsi_hip_index_build_synthetic_spec)."""
        h = C.c_void_p()
        rp, fam, div = (int(x) for x in repeats)
        spec = SyntheticSpec(length, seed, k, int(with_klcp), prefix_len, rp, fam, div, int(mask_permille))
        _check(lib().fmsi_hip_index_build_synthetic_spec(C.byref(spec), device, C.byref(h)))
        return cls(None, device, _handle=h)

    def download_layout(self):
        nb = int(self.info.n_blocks)
        blocks = np.zeros((nb, 16), dtype=np.uint32)
        mrank = np.zeros(nb + 1, dtype=np.uint32)
        _check(lib().fmsi_hip_index_download_layout(self._h, blocks.ctypes.data, mrank.ctypes.data))
        return blocks, mrank

    def export(self) -> bytes:
        """The masked superstring of the index (export_ms, reference src/fms_index.h:462-473), walked on the device."""
        n = int(self.info.n) - 1
        buf = C.create_string_buffer(max(1, n))
        got = C.c_uint64()
        _check(lib().fmsi_hip_index_export(self._h, buf, n, C.byref(got)))
        return buf.raw[:int(got.value)]

    def write_files(self, prefix: str) -> None:
        """Write the index as the reference's <prefix>.fmsi.* files (fmsi_hip_index_write_files)."""
        _check(lib().fmsi_hip_index_write_files(self._h, os.fsencode(prefix)))

    has_pair_layout = False

    def enable_pair_layout(self):
        """Add the 128-B two-base-step layout (fmsi_hip_index_enable_pair_layout); AUTO kernels then use it."""
        _check(lib().fmsi_hip_index_enable_pair_layout(self._h))
        _check(lib().fmsi_hip_index_get_info(self._h, C.byref(self.info)))
        self.has_pair_layout = True
        return self

    def enable_verify(self):
        """Add text verification for one-row table intervals (fmsi_hip_index_enable_verify); used by the phased kernels."""
        _check(lib().fmsi_hip_index_enable_verify(self._h))
        _check(lib().fmsi_hip_index_get_info(self._h, C.byref(self.info)))
        return self

    def enable_dict(self, sparse: bool = False, chain: bool = False, superkmer: bool = False, superkmer_only: bool = False):
        """Add the k-mer dictionary (fmsi_hip_index_enable_dict): presence queries with the index's k become one bucket read;
        chain: also the extras of chained lookups for reads (query_runs over reads of one length)."""
        _check(lib().fmsi_hip_index_enable_dict(self._h, (DICT_SPARSE if sparse else 0) | (DICT_CHAIN if chain else 0) | (DICT_SUPERKMER if superkmer else 0) |
                                                (DICT_SUPERKMER_ONLY if superkmer_only else 0)))
        _check(lib().fmsi_hip_index_get_info(self._h, C.byref(self.info)))
        return self

    def download_klcp(self) -> np.ndarray:
        words = np.zeros((int(self.info.n) + 63) // 64 + 1, dtype=np.uint64)
        _check(lib().fmsi_hip_index_download_klcp(self._h, words.ctypes.data))
        return words

    def download_text(self, length: int) -> np.ndarray:
        words = np.zeros(length // 32 + 2, dtype=np.uint64)
        _check(lib().fmsi_hip_index_download_text(self._h, words.ctypes.data))
        return words

    def synthetic_reads_dev(self, read_len: int, seed: int, first_read: int, n_reads: int, packed_ptr: int,
                            subst_per_million: int = 0, stream: int = 0):
        _check(lib().fmsi_hip_synthetic_reads(self._h, read_len, seed, first_read, n_reads, subst_per_million, packed_ptr, stream or None))

    def synthetic_queries_dev(self, k: int, seed: int, first: int, n: int, kmers_ptr: int, stream: int = 0):
        _check(lib().fmsi_hip_synthetic_queries(self._h, k, seed, first, n, kmers_ptr, stream or None))

    @classmethod
    def from_files(cls, prefix: str, use_klcp: bool = False, device: int = 0, prefix_len: int = 0):
        host = HostIndex(prefix, use_klcp)
        host.view.prefix_len = prefix_len
        return cls(host.view, device, keepalive=host)

    @classmethod
    def load(cls, prefix: str, use_klcp: bool = False, device: int = 0, prefix_len: int = 0, async_table: bool = False):
        """load_index() straight into HBM (fmsi_hip_index_load): RRR decode and block layout on the device.  async_table
        (FMSI_HIP_LOAD_ASYNC_TABLE): the handle comes back when the prefix table's intervals stand; the signatures of its short
        entries are written beside the first queries (wait_background() waits for them)."""
        h = C.c_void_p()
        _check(lib().fmsi_hip_index_load(os.fsencode(prefix), (LOAD_KLCP if use_klcp else 0) | (LOAD_ASYNC_TABLE if async_table else 0), device, prefix_len, C.byref(h)))
        return cls(None, device, _handle=h)

    def wait_background(self):
        _check(lib().fmsi_hip_index_wait_background(self._h))
        return self

    def release_scratch(self):
        """Return the handle's staging / scratch buffers to the driver (fmsi_hip_index_release_scratch); info.device_bytes does not count them."""
        _check(lib().fmsi_hip_index_release_scratch(self._h))
        return self

    # -- host-pointer calls (library stages through its own device buffers) --
    def query_kmers(self, kmers: np.ndarray, k: int, flags: int = MODE_OR, counters: bool = False):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        n = kmers.size
        out = self._alloc_out(n, flags)
        cn = np.zeros(N_COUNTERS, dtype=np.uint64) if counters else None
        _check(lib().fmsi_hip_query_kmers(self._h, kmers.ctypes.data, n, k, flags, out.ctypes.data,
                                          cn.ctypes.data if counters else None, None))
        self.sync()
        return (out, cn) if counters else out

    def query_windows(self, packed: np.ndarray, seq_bases: int, k: int, flags: int = MODE_OR,
                      window_start: np.ndarray | None = None, n_windows: int | None = None, counters: bool = False):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        if window_start is not None:
            window_start = np.ascontiguousarray(window_start, dtype=np.uint64)
            n = window_start.size
        else:
            n = n_windows if n_windows is not None else max(0, seq_bases - k + 1)
        out = self._alloc_out(n, flags)
        cn = np.zeros(N_COUNTERS, dtype=np.uint64) if counters else None
        _check(lib().fmsi_hip_query_windows(self._h, packed.ctypes.data, seq_bases,
                                            window_start.ctypes.data if window_start is not None else None, n, k, flags,
                                            out.ctypes.data, cn.ctypes.data if counters else None, None))
        self.sync()
        return (out, cn) if counters else out

    def query_runs(self, packed: np.ndarray, seq_bases: int, run_start, run_windows, k: int, flags: int = MODE_OR,
                   counters: bool = False):
        """Non-streaming queries over runs of consecutive windows (window starts expanded on the device)."""
        return self.query_stream(packed, seq_bases, run_start, run_windows, k, flags, counters, _fn="fmsi_hip_query_runs")

    def query_runs_fn(self, packed: np.ndarray, seq_bases: int, run_start, run_windows, k: int, fn: int, r: int = 0, s: int = 0,
                      flags: int = 0, counters: bool = False, lower_case: np.ndarray | None = None):
        """`-f and|xor|r-s` over runs of consecutive windows: f(ones, total) of both strands per window (FN_AND / FN_XOR / FN_RANGE).
        lower_case: one bit per base (lower_case_bits(seq)), which decides whether a self-complementary k-mer counts once or twice."""
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        if lower_case is not None:
            lower_case = np.ascontiguousarray(lower_case, dtype=np.uint64)
            assert lower_case.size >= seq_bases // 64 + 1
        run_start = np.ascontiguousarray(run_start, dtype=np.uint64)
        run_windows = np.ascontiguousarray(run_windows, dtype=np.uint32)
        n = int(run_windows.sum())
        out = self._alloc_out(n, flags)
        cn = np.zeros(N_COUNTERS, dtype=np.uint64) if counters else None
        _check(lib().fmsi_hip_query_runs_fn(self._h, packed.ctypes.data, lower_case.ctypes.data if lower_case is not None else None, seq_bases, run_start.ctypes.data, run_windows.ctypes.data, run_start.size, n, k,
                                            fn, r, s, flags, out.ctypes.data, cn.ctypes.data if counters else None, None))
        self.sync()
        return (out, cn) if counters else out

    def query_stream(self, packed: np.ndarray, seq_bases: int, run_start, run_windows, k: int, flags: int = MODE_OR,
                     counters: bool = False, _fn: str = "fmsi_hip_query_stream"):
        """kLCP streaming queries (`-S`): runs of consecutive windows; results in run order."""
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        run_start = np.ascontiguousarray(run_start, dtype=np.uint64)
        run_windows = np.ascontiguousarray(run_windows, dtype=np.uint32)
        n = int(run_windows.sum())
        out = self._alloc_out(n, flags)
        cn = np.zeros(N_COUNTERS, dtype=np.uint64) if counters else None
        _check(getattr(lib(), _fn)(self._h, packed.ctypes.data, seq_bases, run_start.ctypes.data, run_windows.ctypes.data,
                                   run_start.size, n, k, flags, out.ctypes.data, cn.ctypes.data if counters else None, None))
        self.sync()
        return (out, cn) if counters else out

    def query_stream_dev(self, packed_ptr: int, seq_bases: int, run_start, run_windows, k: int, flags: int, out_ptr: int,
                         counters_ptr: int = 0, stream: int = 0, _fn: str = "fmsi_hip_query_stream"):
        """Same with the sequence / results resident in HBM; the run arrays are host arrays in every mode."""
        run_start = np.ascontiguousarray(run_start, dtype=np.uint64)
        run_windows = np.ascontiguousarray(run_windows, dtype=np.uint32)
        _check(getattr(lib(), _fn)(self._h, packed_ptr, seq_bases, run_start.ctypes.data, run_windows.ctypes.data, run_start.size,
                                   int(run_windows.sum()), k, flags | DEVICE_PTRS, out_ptr, counters_ptr or None, stream or None))

    def query_runs_dev(self, packed_ptr: int, seq_bases: int, run_start, run_windows, k: int, flags: int, out_ptr: int,
                       counters_ptr: int = 0, stream: int = 0):
        """Runs of windows through the non-streaming kernels, sequence / results resident in HBM."""
        self.query_stream_dev(packed_ptr, seq_bases, run_start, run_windows, k, flags, out_ptr, counters_ptr, stream, _fn="fmsi_hip_query_runs")

    # -- device-pointer calls (buffers already resident in HBM, e.g. torch tensors' data_ptr()) --
    def query_kmers_dev(self, kmers_ptr: int, n: int, k: int, flags: int, out_ptr: int, counters_ptr: int = 0, stream: int = 0):
        _check(lib().fmsi_hip_query_kmers(self._h, kmers_ptr, n, k, flags | DEVICE_PTRS, out_ptr, counters_ptr or None,
                                          stream or None))

    def query_windows_dev(self, packed_ptr: int, seq_bases: int, wstart_ptr: int, n: int, k: int, flags: int, out_ptr: int,
                          counters_ptr: int = 0, stream: int = 0):
        _check(lib().fmsi_hip_query_windows(self._h, packed_ptr, seq_bases, wstart_ptr or None, n, k, flags | DEVICE_PTRS,
                                            out_ptr, counters_ptr or None, stream or None))

    def sync(self, stream: int = 0):
        _check(lib().fmsi_hip_sync(self._h, stream or None))

    def profile(self, on: bool):
        _check(lib().fmsi_hip_profile_enable(self._h, int(on)))

    def profile_read(self, reset: bool = True):
        n, ms = C.c_uint64(), C.c_double()
        _check(lib().fmsi_hip_profile_read(self._h, C.byref(n), C.byref(ms), int(reset)))
        return int(n.value), float(ms.value)

    @staticmethod
    def _alloc_out(n: int, flags: int) -> np.ndarray:
        if flags & ORDERS:
            return np.zeros((n, 2) if flags & STRICT else n, dtype=np.int64)
        if flags & OUT_BITS:   # u64 words: 1 bit per window, STRICT: one nibble
            return np.zeros((n + 15) // 16 if flags & STRICT else (n + 63) // 64, dtype=np.uint64)
        return np.zeros(n, dtype=np.uint8)

    def close(self):
        if self._h:
            lib().fmsi_hip_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_M64 = (1 << 64) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def synthetic_text_words(length: int, seed: int) -> np.ndarray:
    """Host (numpy) restatement of the builder's synthetic superstring: 2-bit words, zero past `length`."""
    nw = length // 32 + 2
    with np.errstate(over="ignore"):
        w = np.arange(nw, dtype=np.uint64)
        words = _splitmix64(np.uint64(seed) ^ (w * np.uint64(0xD1B54A32D192ED03)))
    first = np.arange(nw, dtype=np.int64) * 32
    words[first >= length] = 0
    part = np.nonzero((first < length) & (first + 32 > length))[0]
    for i in part:
        words[i] &= np.uint64((1 << (2 * (length - int(first[i])))) - 1)
    return words


def synthetic_mask(length: int, seed: int, k: int, mask_permille: int = 1000) -> np.ndarray:
    """Host (numpy) restatement of the builder's synthetic mask (fmsi_hip_synthetic_mask_bit): bool per position."""
    p = np.arange(length, dtype=np.uint64)
    on = p + np.uint64(k) <= np.uint64(length)
    if mask_permille < 1000:
        with np.errstate(over="ignore"):
            h = _splitmix64(np.uint64(seed) ^ np.uint64(0x3A5C0FF5E7) ^ (p * np.uint64(0xC2B2AE3D27D4EB4F)))
        on &= (h % np.uint64(1000)) < np.uint64(mask_permille)
    return on


def synthetic_text(length: int, seed: int, k: int, mask_permille: int = 1000) -> bytes:
    """The synthetic masked superstring as text (ON positions upper case; max-ones: positions 0..length-k)."""
    words = synthetic_text_words(length, seed)
    codes = ((words[:, None] >> (2 * np.arange(32, dtype=np.uint64))) & np.uint64(3)).reshape(-1)[:length].astype(np.uint8)
    del words
    on = synthetic_mask(length, seed, k, mask_permille)
    out = np.frombuffer(b"acgt", dtype=np.uint8)[codes]
    out = out - (on.astype(np.uint8) << 5)   # upper case where ON
    return out.tobytes()


def decode_strict(out: np.ndarray):
    """Split STRICT presence bytes into (fwd, rc) int8 arrays with values in {-1, 0, 1}."""
    return (out & 3).astype(np.int8) - 1, ((out >> 2) & 3).astype(np.int8) - 1


def trim_scratch(device: int = 0) -> None:
    """Return the library's cached device blocks (temporaries of the builders) to the driver (fmsi_hip_trim_scratch)."""
    _check(lib().fmsi_hip_trim_scratch(device))


def scratch_bytes(device: int = 0) -> int:
    """Bytes of freed device blocks the library keeps for reuse (fmsi_hip_scratch_bytes)."""
    return int(lib().fmsi_hip_scratch_bytes(device))


class builder_chain:
    """`with builder_chain(device):` -- a chain of builder calls (build / load, enable_pair_layout, enable_dict ...) shares the library's
    cached device blocks (fmsi_hip_keep_scratch) and returns them to the driver once, when the chain ends."""

    def __init__(self, device: int = 0):
        self.device = device

    def __enter__(self):
        lib().fmsi_hip_keep_scratch(1)
        return self

    def __exit__(self, *exc):
        lib().fmsi_hip_keep_scratch(0)
        try:
            trim_scratch(self.device)
        except FmsiError:
            pass
        return False


def packed_words(n: int, flags: int) -> int:
    """u64 words of an OUT_BITS result buffer over n windows."""
    return (n + 15) // 16 if flags & STRICT else (n + 63) // 64


def unpack_bits(words: np.ndarray, n: int, flags: int) -> np.ndarray:
    """OUT_BITS result words -> the one-byte-per-window form of the same call without the flag."""
    raw = np.ascontiguousarray(words, dtype=np.uint64).view(np.uint8)
    if flags & STRICT:
        return np.stack([raw & 0xF, raw >> 4], axis=1).reshape(-1)[:n].astype(np.uint8)
    return np.unpackbits(raw, bitorder="little")[:n]


def run_cli(args, stdin: bytes | None = None):
    """Run the built `fmsi` CLI; returns (returncode, stdout bytes, stderr bytes)."""
    if not CLI_PATH.exists():
        raise FileNotFoundError(f"{CLI_PATH} is missing: build first")
    res = subprocess.run([str(CLI_PATH), *args], input=stdin, capture_output=True)
    return res.returncode, res.stdout, res.stderr


COMM_ID_BYTES = 128


class Comm:
    """RCCL communicator of the result gather (fmsi_hip_comm_*): one per process / GPU."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(lib().fmsi_hip_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int = 0):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be COMM_ID_BYTES long")
        self.world, self.rank, self.device = world, rank, device
        self._id = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _check(lib().fmsi_hip_comm_create(self._id, world, rank, device, C.byref(h)))
        self._h = h

    def gather(self, send_ptr: int, recv_ptr: int, bytes_per_rank, root: int = 0, stream: int = 0) -> None:
        """Device buffers: rank r's bytes_per_rank[r] bytes at send_ptr land at recv_ptr + sum(bytes_per_rank[:r]) on root."""
        sizes = np.ascontiguousarray(bytes_per_rank, dtype=np.uint64)
        if sizes.size != self.world:
            raise ValueError("bytes_per_rank needs one entry per rank")
        _check(lib().fmsi_hip_gather(self._h, send_ptr or None, int(sizes[self.rank]), recv_ptr or None, sizes.ctypes.data, root,
                                     stream or None))

    def close(self) -> None:
        if self._h:
            lib().fmsi_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

