"""The team's reader of ORDINARY gzip files (fmsi_amd/csrc/pgunzip.h; the reference reads its query file through zlib's gzread on one
thread: src/parser.h:14-27, src/kseq.h:75-93): what it decodes is what zlib decodes, byte for byte, whatever the chunk size and the
number of threads -- for streams of every block type, compression level and strategy, with flush points, several members, long
repeats, text that is not text -- and a stream that zlib would refuse (or accept only in part) makes it give up at a byte that zlib
reaches, too.  CPU only: the entry point is a host function of the C-ABI."""
import ctypes as C
import gzip
import random
import struct
import zlib

import numpy as np
import pytest

import fmsi_amd


@pytest.fixture(scope="module")
def lib(product_built):
    L = fmsi_amd.lib()
    L.fmsi_host_gunzip.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.fmsi_host_gunzip.restype = C.c_int
    return L


def gunzip(L, path, threads, chunk, cap):
    buf = np.empty(max(1, cap), dtype=np.uint8)
    n = C.c_uint64(0)
    st = (C.c_uint64 * 7)()
    rc = L.fmsi_host_gunzip(str(path).encode(), threads, chunk, buf.ctypes.data, cap, C.byref(n), st)
    return rc, bytes(buf[:min(n.value, cap)]), n.value, dict(zip(("rounds", "chunks", "passed_over", "no_start", "marker_symbols", "plain_bytes", "members"), st))


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flushes=0, wbits=31, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    out = []
    if flushes:
        step = max(1, len(data) // flushes)
        for i in range(0, len(data), step):
            out.append(c.compress(data[i:i + step]))
            out.append(c.flush(zlib.Z_SYNC_FLUSH if (i // step) % 2 else zlib.Z_FULL_FLUSH))
    else:
        out.append(c.compress(data))
    out.append(c.flush())
    return b"".join(out)


def zlib_reach(gz):
    """What zlib itself gets out of a stream, member after member, fed in small pieces, up to the byte where it stops or fails."""
    out = []
    data = gz
    while data:
        d = zlib.decompressobj(47)
        try:
            for i in range(0, len(data), 4096):
                out.append(d.decompress(data[i:i + 4096]))
                if d.eof:
                    break
        except zlib.error:
            break
        if not d.eof:
            break
        used = min(len(data), (i + 4096)) - len(d.unused_data)
        data = data[used:]
    return b"".join(out)


def member(raw, data, flg=0, extra=b"", name=b"", comment=b""):
    h = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0" + b"\0\x03"
    if flg & 4:
        h += struct.pack("<H", len(extra)) + extra
    if flg & 8:
        h += name + b"\0"
    if flg & 16:
        h += comment + b"\0"
    return h + raw + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


@pytest.fixture(scope="module")
def texts():
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def dna(n):
        return acgt[rng.integers(0, 4, n)].tobytes()

    fasta = b"".join(b">read_%d a comment\n" % i + dna(5000) + b"\n" for i in range(500))
    fastq = b"".join(b"@r%d\n" % i + dna(150) + b"\n+\n" + rng.integers(33, 74, 150).astype(np.uint8).tobytes() + b"\n" for i in range(12000))
    kmers = b"".join(b">%d\n" % i + dna(31) + b"\n" for i in range(60000))
    rep = (b">x\n" + b"ACGTTGCA" * 60000 + b"\n") * 4 + b"A" * 2000000 + dna(70000)
    binary = rng.integers(0, 256, 2000000).astype(np.uint8).tobytes()
    return {"fasta": fasta, "fastq": fastq, "kmers": kmers, "rep": rep, "binary": binary}


SCHEDULES = ((1, 0), (3, 1 << 12), (8, 1 << 15), (5, 100000), (8, 1 << 18), (8, 0))


def test_equals_zlib_on_every_kind_of_stream(lib, texts, tmp_path):
    fa, fq = texts["fasta"], texts["fastq"]
    raw = deflate(fa, 6, wbits=-15)
    cases = {
        "fasta1": (deflate(fa, 1), fa), "fasta6": (deflate(fa), fa), "fasta9": (deflate(fa, 9), fa), "fastq6": (deflate(fq), fq), "fastq1": (deflate(fq, 1), fq),
        "kmers": (deflate(texts["kmers"]), texts["kmers"]), "empty": (deflate(b""), b""), "tiny": (deflate(b">a\nACGT\n"), b">a\nACGT\n"),
        "members": (deflate(fa[:700000]) + deflate(fa[700000:1500000], 9) + deflate(b"") + deflate(fa[1500000:], 1), fa),
        "flushes": (deflate(fa, 6, flushes=37), fa), "stored": (deflate(fa, 0), fa), "fixed": (deflate(fa, 6, zlib.Z_FIXED), fa),
        "huffman_only": (deflate(fa, 6, zlib.Z_HUFFMAN_ONLY), fa), "rle": (deflate(fa, 6, zlib.Z_RLE), fa), "filtered": (deflate(fa, 6, zlib.Z_FILTERED), fa),
        "repetitive": (deflate(texts["rep"]), texts["rep"]), "binary": (deflate(texts["binary"]), texts["binary"]),
        "memlevel1": (deflate(fa, 6, memlevel=1), fa), "memlevel9": (deflate(fa, 9, memlevel=9), fa),
        "header_fields": (member(raw, fa, 4 | 8 | 16, b"AB\x02\x00xy", b"file.fa", b"a comment"), fa),
        "python_gzip": (gzip.compress(fq, 4), fq),
    }
    for name, (gz, want) in cases.items():
        p = tmp_path / f"{name}.gz"
        p.write_bytes(gz)
        assert zlib.decompressobj(47).decompress(gz)[:len(want)] == want[:len(zlib.decompressobj(47).decompress(gz))] or name == "members"
        for th, ch in SCHEDULES:
            rc, out, n, st = gunzip(lib, p, th, ch, len(want) + 16)
            assert rc == 0 and n == len(want) and out == want, (name, th, ch, rc, n, st)
            assert st["members"] == (4 if name == "members" else 1)
            assert st["marker_symbols"] + st["plain_bytes"] == len(want)
    # the decoder of unknown windows did work: small chunks of a stream of dynamic blocks start in its middle
    rc, out, n, st = gunzip(lib, tmp_path / "fasta6.gz", 8, 1 << 15, len(fa))
    assert st["marker_symbols"] > len(fa) // 2 and st["rounds"] >= 2 and st["chunks"] > st["rounds"] * 4
    # bytes beyond the room are decoded, checked and counted
    rc, out, n, st = gunzip(lib, tmp_path / "fasta6.gz", 4, 1 << 16, 1000)
    assert rc == 0 and n == len(fa) and out == fa[:1000]


def test_gives_up_where_zlib_would(lib, texts, tmp_path):
    fa = texts["fasta"]
    raw = deflate(fa, 6, wbits=-15)
    good = member(raw, fa)
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x55
    cases = {
        "crc": good[:-8] + bytes([good[-8] ^ 1]) + good[-7:], "isize": good[:-1] + bytes([good[-1] ^ 1]), "cut": good[:len(good) // 2], "cut_in_trailer": good[:-3],
        "junk_behind": good + b"\0" * 100, "flipped_byte": bytes(flipped), "second_member_broken": good + good[:len(good) // 3],
    }
    for name, gz in cases.items():
        p = tmp_path / f"{name}.gz"
        p.write_bytes(gz)
        reach = zlib_reach(gz)
        for th, ch in SCHEDULES:
            rc, out, n, st = gunzip(lib, p, th, ch, len(fa) * 2 + 16)
            m = min(len(out), len(reach))   # (zlib fed in pieces loses the output of the piece it fails in: either may be the longer one)
            assert rc == 1 and out[:m] == reach[:m], (name, th, ch, rc, n)
            assert name == "flipped_byte" or (fa + fa).startswith(out), (name, th, ch)
    # text that deflates a thousand-fold: a chunk's text has a bound (96 MB: a round of 64 chunks must not hold tens of gigabytes), the round
    # fails and zlib -- fast on such a stream -- is asked to go on from the byte the reader got to
    bomb = tmp_path / "bomb.gz"
    bomb.write_bytes(deflate(b">x\n" + b"ACGT" * (30 << 20) + b"\n", 6))
    rc, out, n, st = gunzip(lib, bomb, 4, 0, 1024)
    assert rc == 1 and n == 0, (rc, n, st)
    # not a gzip file at all / a header that asks for zlib's own checks
    for name, data in (("plain", fa[:10000]), ("header_crc", member(raw, fa, 2)[:10] + b"\0\0" + member(raw, fa)[10:]), ("reserved_flag", member(raw, fa, 0x20)), ("short", b"\x1f\x8b")):
        p = tmp_path / f"{name}.bin"
        p.write_bytes(data)
        assert gunzip(lib, p, 4, 0, 16)[0] == -1


def test_random_streams_and_schedules(lib, tmp_path):
    """Seeded mixtures: texts of random alphabets and repeat structure, random levels / strategies / flush points / member cuts, random chunk sizes."""
    rng = random.Random(11)
    nrng = np.random.default_rng(11)
    for it in range(40):
        parts = []
        for _ in range(rng.randrange(1, 6)):
            kind = rng.randrange(4)
            n = rng.randrange(1, 400000)
            if kind == 0:
                alpha = np.frombuffer(bytes(rng.sample(range(256), rng.randrange(2, 40))), dtype=np.uint8)
                parts.append(alpha[nrng.integers(0, len(alpha), n)].tobytes())
            elif kind == 1:
                unit = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 300)))
                parts.append((unit * (n // len(unit) + 1))[:n])
            elif kind == 2:
                parts.append(b"".join(b">%d\n" % i + np.frombuffer(b"ACGT", dtype=np.uint8)[nrng.integers(0, 4, 80)].tobytes() + b"\n" for i in range(n // 90 + 1)))
            else:
                parts.append(nrng.integers(0, 256, n // 4 + 1).astype(np.uint8).tobytes())
        text = b"".join(parts)
        cuts = sorted(rng.sample(range(len(text) + 1), rng.randrange(0, 3)))
        gz = b""
        for a, b in zip([0] + cuts, cuts + [len(text)]):
            gz += deflate(text[a:b], rng.choice([1, 2, 4, 6, 9]), rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY]),
                          flushes=rng.choice([0, 0, 3, 17]), memlevel=rng.choice([1, 5, 8, 9]))
        p = tmp_path / f"r{it}.gz"
        p.write_bytes(gz)
        for _ in range(3):
            th, ch = rng.randrange(1, 9), rng.choice([0, rng.randrange(64, 5000), rng.randrange(5000, 200000)])
            rc, out, n, st = gunzip(lib, p, th, ch, len(text) + 8)
            assert rc == 0 and out == text, (it, th, ch, rc, n, len(text), st)
