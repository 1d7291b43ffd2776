"""CPU tests of the product's host side: the C-ABI library loads and exports what include/fmsi_hip.h
declares, the .fmsi.* loader + RRR-63 decoder, the interleaved device layout (checked by emulating the
kernel's rank arithmetic in numpy against the oracle), 2-bit packing, the CLI's option handling -- and
that the product fails LOUDLY without a GPU instead of falling back to anything.
"""
import ctypes as C
import re
import random
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_py
from conftest import DATASETS, GOLDEN, INDEX_PARTS, REPO, golden_index_part, index_prefix, synthetic_ms


def declared_functions():
    text = (REPO / "include" / "fmsi_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fmsi_(?:hip|host|cli)_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(product_built):
    lib = product_built.lib()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fmsi_hip.h but not exported"
    assert lib.fmsi_hip_abi_version() == 8


def test_no_oracle_in_product():
    """The product must not link, import or call anything under oracle/."""
    for p in list((REPO / "fmsi_amd").rglob("*.py")) + list((REPO / "fmsi_amd" / "csrc").glob("*")):
        if p.is_file() and p.suffix in {".py", ".cpp", ".h", ".hip", ""}:
            t = p.read_text(errors="ignore")
            assert "oracle_py" not in t and "liboracle" not in t and "fmsi_oracle" not in t and "_ref/" not in t, p
    out = subprocess.run(["ldd", str(REPO / "fmsi_amd" / "lib" / "libfmsi_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "fmsi_ref" not in out


@pytest.mark.parametrize("name", list(DATASETS))
def test_loader_matches_oracle(product_built, oracle_built, name):
    host = product_built.HostIndex(index_prefix(name), use_klcp=True)
    orc = oracle_py.Oracle.load(index_prefix(name), use_klcp=True)
    assert host.n == orc.n and host.k == orc.k == DATASETS[name][0]
    assert list(host.view.counts) == orc.counts() and host.view.dollar_position == orc.dollar()
    for which, key in ((0, "ac_gt"), (1, "ac"), (2, "gt"), (3, "klcp")):
        words, nbits = orc.words(which)
        want = np.unpackbits(words.view(np.uint8), bitorder="little")[:nbits]
        assert np.array_equal(host.bits(key), want), key
    assert np.array_equal(host.bits("mask"), orc.mask_bits())     # RRR-63 decode
    nok = product_built.HostIndex(index_prefix(name), use_klcp=False)
    assert nok.view.klcp_bits == 0


@pytest.mark.parametrize("name", list(DATASETS) + ["repeats_k40"])
def test_writer_reproduces_reference_files(product_built, tmp_path, name):
    """load -> fmsi_host_index_write == the reference's dump_index() bytes (src/fms_index.h:484-500): pins the RRR-63
    encoder (class / offset / sample layout of sdsl rrr_vector<63>) and the int_vector framing."""
    host = product_built.HostIndex(index_prefix(name), use_klcp=True)
    host.write(str(tmp_path / name))
    for part in INDEX_PARTS:
        assert (tmp_path / f"{name}.fmsi.{part}").read_bytes() == golden_index_part(name, part), part
    nok = product_built.HostIndex(index_prefix(name), use_klcp=False)
    nok.write(str(tmp_path / "nok"))
    assert not (tmp_path / "nok.fmsi.klcp").exists()


def test_writer_rrr_edge_lengths(product_built, oracle_built, tmp_path):
    """Mask lengths around the 63-bit block and 32-block superblock boundaries, several densities: written by the
    product, read back by the oracle's independent RRR decoder."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 62, 63, 64, 125, 126, 127, 63 * 32 - 1, 63 * 32, 63 * 32 + 1, 63 * 64, 5000):
        for density in (0.0, 0.03, 0.5, 0.97, 1.0):
            bits = (rng.random(n) < density).astype(np.uint8)
            sym = rng.integers(0, 4, n)
            sym[0] = 0
            planes = {}
            hi, lo = (sym >> 1).astype(np.uint8), (sym & 1).astype(np.uint8)
            planes["ac_gt"] = hi
            planes["ac"] = lo[hi == 0]
            planes["gt"] = lo[hi == 1]
            arrs = {}
            for key, b in list(planes.items()) + [("mask", bits)]:
                w = np.zeros((len(b) + 63) // 64 + 1, dtype=np.uint64)
                packed = np.packbits(b, bitorder="little")
                w.view(np.uint8)[:len(packed)] = packed
                arrs[key] = (w, len(b))
            cnt = [1, 1 + int((sym == 0).sum()), 1 + int((sym <= 1).sum()), 1 + int((sym <= 2).sum())]
            view, keep = product_built.view_from_arrays(arrs["ac_gt"], arrs["ac"], arrs["gt"], arrs["mask"], cnt, 0, 5)
            rc = product_built.lib().fmsi_host_index_write(str(tmp_path / "e").encode(), C.byref(view))
            assert rc == 0
            orc = oracle_py.Oracle.load(str(tmp_path / "e"), use_klcp=False)
            assert orc.n == n and np.array_equal(orc.mask_bits(), bits), (n, density)
            back = product_built.HostIndex(str(tmp_path / "e"))
            assert np.array_equal(back.bits("mask"), bits) and np.array_equal(back.bits("ac_gt"), hi)


def test_loader_errors(product_built, tmp_path):
    with pytest.raises(product_built.FmsiError) as e:
        product_built.HostIndex(str(tmp_path / "missing"))
    assert e.value.code == product_built.E_IO
    # truncated mask
    for part in ("ac_gt", "ac", "gt", "misc"):
        (tmp_path / f"t.fmsi.{part}").write_bytes((GOLDEN / "idx" / f"sars_k13.fmsi.{part}").read_bytes())
    (tmp_path / "t.fmsi.mask").write_bytes((GOLDEN / "idx" / "sars_k13.fmsi.mask").read_bytes()[:100])
    with pytest.raises(product_built.FmsiError):
        product_built.HostIndex(str(tmp_path / "t"))


def emulate_rank(blocks, counts, dollar, i, c, upper):
    """numpy restatement of the kernel's rank_in_block()/lf_step() address and popcount arithmetic."""
    if upper:
        b = (i - 1) >> 7
        p = i - (b << 7)
    else:
        b = i >> 7
        p = i & 127
    blk = blocks[b]
    r = int(blk[c])
    for w in range(4):
        s = min(max(p - 32 * w, 0), 32)
        pm = (1 << s) - 1
        h, l = int(blk[4 + w]), int(blk[8 + w])
        m = (h if c & 2 else ~h) & (l if c & 1 else ~l) & pm & 0xFFFFFFFF
        r += bin(m).count("1")
    if c == 0 and i > dollar:
        r -= 1
    return r


@pytest.mark.parametrize("name", ["integration_a", "sars_k13", "mixed_k11"])
def test_layout_rank_equals_oracle(product_built, oracle_built, name):
    host = product_built.HostIndex(index_prefix(name))
    orc = oracle_py.Oracle.load(index_prefix(name))
    blocks, mrank = host.layout()
    n = host.n
    assert blocks.shape == (((n + 127) >> 7) + 1, 16)
    rng = random.Random(1)
    pos = list(range(min(n + 1, 300))) + [rng.randrange(0, n + 1) for _ in range(1500)] + [n, n - 1, max(0, n - 127)]
    dollar = host.view.dollar_position
    for i in pos:
        for c in range(4):
            want = orc.rank(i, c)
            if i < n:
                assert emulate_rank(blocks, None, dollar, i, c, upper=False) == want
            if i > 0:
                assert emulate_rank(blocks, None, dollar, i, c, upper=True) == want
    mask = orc.mask_bits()
    pref = np.concatenate([[0], np.cumsum(mask)])
    for b in range(blocks.shape[0]):
        assert mrank[b] == pref[min(n, b * 128)]
    got_mask = np.unpackbits(blocks[:, 12:16].copy().view(np.uint8).reshape(-1), bitorder="little")[:n]
    assert np.array_equal(got_mask, mask)


def test_layout_sizes_around_block_boundaries(product_built, oracle_built, tmp_path):
    """n = 127, 128, 129, 255, 256, 257 ... rows: last-block padding, n % 64 == 0, n % 128 == 0."""
    for length in (126, 127, 128, 191, 254, 255, 256, 383, 511, 512):
        k = 5
        ms = synthetic_ms(length, k, seed=length, mask="mixed")
        orc = oracle_py.Oracle.construct(ms, k)
        prefix = str(tmp_path / f"n{length}")
        orc.dump(prefix)
        host = product_built.HostIndex(prefix)
        blocks, _ = host.layout()
        n = host.n
        assert n == length + 1
        for i in range(n + 1):
            for c in range(4):
                want = orc.rank(i, c)
                if i < n:
                    assert emulate_rank(blocks, None, host.view.dollar_position, i, c, False) == want
                if i > 0:
                    assert emulate_rank(blocks, None, host.view.dollar_position, i, c, True) == want


def test_pack(product_built):
    k = 31
    s = "ACGTTGCAAACCGGTTACGTACGTAGCTAGC"
    w = product_built.pack_kmers([s, s.lower()], k)
    assert w[0] == w[1]
    codes = {"A": 0, "C": 1, "G": 2, "T": 3}
    assert int(w[0]) == sum(codes[ch] << (2 * t) for t, ch in enumerate(s))
    seq = ("ACGT" * 40).encode()
    words = product_built.pack_sequence(seq)
    assert int(words[0]) & 0xFF == 0b11100100
    with pytest.raises(ValueError):
        product_built.pack_sequence(b"ACGNACGT")


def test_product_fails_loudly_without_gpu(product_built):
    """No CPU fallback: on a box without a HIP device index creation must fail with ENODEV."""
    if product_built.device_count() > 0:
        pytest.skip("a GPU is present")
    host = product_built.HostIndex(index_prefix("sars_k31"))
    with pytest.raises(product_built.FmsiError) as e:
        product_built.DeviceIndex(host.view)
    assert e.value.code == product_built.E_NODEV
    rc, out, err = product_built.run_cli(["query", "-q", str(GOLDEN / "rand_sars_k31.fa"), index_prefix("sars_k31")])
    assert rc == 1 and out == b"" and b"cannot create the device index" in err


def test_cli_option_handling(product_built):
    """Exit codes and messages of ms_query's validation (reference src/main.cpp:288-320)."""
    run = product_built.run_cli
    rc, out, err = run(["query", "-h"])
    assert rc == 0 and b"Usage:   fmsi query [options] <index-prefix>" in err
    rc, out, err = run(["lookup", "-h"])
    assert rc == 0 and b"Usage:   fmsi lookup" in err
    rc, out, err = run(["query"])
    assert rc == 1 and b"required argument" in err
    rc, out, err = run(["query", "-k", "5", index_prefix("sars_k31")])
    assert rc == 1 and b"Mismatch. Provided k (5) does not match the k of the index (31)" in err
    rc, out, err = run(["query", str(GOLDEN / "idx" / "nope")])
    assert rc == 1 and b"index not correctly loaded" in err
    rc, out, err = run(["query", "-f", "nand", index_prefix("sars_k31")])               # src/main.cpp:255-263
    assert rc == 1 and b"ERROR: Function 'nand' not recognized." in err and b"Usage:   fmsi query" in err
    rc, out, err = run(["lookup", "-f", "xor", index_prefix("sars_k31")])               # src/main.cpp:297-300
    assert rc == 1 and b"not allowed with f-masked superstrings" in err
    rc, out, err = run(["-v"])
    assert rc == 0 and out.strip()
    rc, out, err = run(["merge", "x"])
    assert rc == 1 and b"not part of the MI355X query path" in err
    rc, out, err = run(["export", "-h"])
    assert rc == 0 and b"Usage:   fmsi export <index-prefix>" in err
    rc, out, err = run([])
    assert rc == 1


def test_gather_argument_errors_without_a_gpu(product_built):
    """fmsi_hip_gather / fmsi_hip_comm_create reject bad arguments before RCCL or a device is touched."""
    import ctypes as C

    fm = product_built
    L = fm.lib()
    h = C.c_void_p()
    ident = (C.c_uint8 * fm.COMM_ID_BYTES)()
    assert L.fmsi_hip_comm_create(ident, 0, 0, 0, C.byref(h)) == -1            # world < 1
    assert L.fmsi_hip_comm_create(ident, 2, 2, 0, C.byref(h)) == -1            # rank outside the world
    assert L.fmsi_hip_comm_create(None, 1, 0, 0, C.byref(h)) == -1
    assert L.fmsi_hip_comm_unique_id(None) == -1
    sizes = (C.c_uint64 * 1)(0)
    assert L.fmsi_hip_gather(None, None, 0, None, sizes, 0, None) == -1
    assert b"bad argument" in L.fmsi_hip_last_error()
    L.fmsi_hip_comm_destroy(None)                                              # a no-op



def test_host_code_under_sanitizers(tmp_path):
    """The host-only code of the library -- index file parser + RRR-63 decoder, writer + RRR-63 encoder, the FASTA/FASTQ
    reader in file and in memory mode -- built with AddressSanitizer + UBSan (tests/native/host_sanitize.cpp, g++, no HIP):
    clean runs, the files it writes back are the golden bytes, and a file cut at its '>' lines reads like the whole."""
    from pathlib import Path
    from conftest import messy_fasta

    repo = REPO
    src = repo / "fmsi_amd" / "csrc"
    exe = tmp_path / "host_sanitize"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           f"-I{repo / 'include'}", str(repo / "tests" / "native" / "host_sanitize.cpp"), str(src / "host_index.cpp"), str(src / "index_writer.cpp"),
           str(src / "layout.cpp"), str(src / "fastx.cpp"), "-o", str(exe), "-lz", "-lpthread"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13") if not l.startswith(">"))
    fastas = {}
    for tag, fq in (("fasta", None), ("mixed", 150)):
        fastas[tag] = tmp_path / f"{tag}.fa"
        fastas[tag].write_bytes(messy_fasta(ms, random.Random(5), fq))
    for name, klcp, tag in (("sars_k31", 1, "fasta"), ("mixed_k11", 1, "mixed"), ("repeats_k64", 1, "fasta"), ("integration_a", 0, "fasta")):
        out = tmp_path / f"w_{name}"
        res = subprocess.run([str(exe), str(GOLDEN / "idx" / name), str(klcp), str(out), str(fastas[tag])], capture_output=True, text=True)
        assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
        assert "400 records" in res.stdout and ("fastq=1" in res.stdout) == (tag == "mixed")
        for part in ("ac_gt", "ac", "gt", "mask", "misc") + (("klcp",) if klcp else ()):
            assert (GOLDEN / "idx" / f"{name}.fmsi.{part}").read_bytes() == Path(f"{out}.fmsi.{part}").read_bytes(), (name, part)


def test_view_parser_equals_reader_fuzz(tmp_path):
    """The CLI's block-parallel reader parses without copying (FastxViewParser: names and single-line sequences are views of the
    mapped file): record by record it must yield what the ordinary reader (the reference's kseq rules) yields on the same bytes --
    well-formed and mutated FASTA / FASTQ, CRLF, stray header characters, truncations.  ASan + UBSan build."""
    exe = tmp_path / "fastx_view_fuzz"
    res = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                          str(REPO / "tests" / "native" / "fastx_view_fuzz.cpp"), str(REPO / "fmsi_amd" / "csrc" / "fastx.cpp"), "-lz", "-o", str(exe)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    res = subprocess.run([str(exe), "150000"], capture_output=True, text=True)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    assert "on 150000 inputs" in res.stdout


def test_cli_host_kernels_equal_their_definitions(tmp_path):
    """The CLI's word-at-a-time host loops -- SSE2 validity scan, PEXT / plain 2-bit packer, PDEP / plain expansion of result bits
    into characters, SSE2 sums and output bytes of a -S chunk -- against their character-at-a-time definitions (ASan + UBSan)."""
    import fmsi_amd
    fmsi_amd.build()
    exe = tmp_path / "cli_host_kernels"
    lib = REPO / "fmsi_amd" / "lib"
    src = REPO / "fmsi_amd" / "csrc"
    res = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", f"-I{REPO / 'include'}",
                          str(REPO / "tests" / "native" / "cli_host_kernels.cpp"), str(src / "fastx.cpp"), str(src / "trace.cpp"), "-o", str(exe),
                          f"-L{lib}", "-lfmsi_hip", "-lz", "-lpthread", "-ldl", f"-Wl,-rpath,{lib}"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    res = subprocess.run([str(exe)], capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0"})
    assert res.returncode == 0 and "cli host kernels OK" in res.stdout, (res.stdout + res.stderr)[-3000:]


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_cli_host_pipeline_under_sanitizers(tmp_path, sanitizer):
    """The whole host pipeline of `fmsi query` / `lookup` (readers -- ordinary, block-parallel on a mapped file, block-gzip --, packing
    teams, two workers per handle, in-order hand-over, predictor replay, formatting, sink) with the device stood in for by
    tests/native/cli_pipeline_stub.cpp, which answers every window from a hash of its k-mer: built with ThreadSanitizer and with
    ASan + UBSan, no reports; and since such answers depend on the k-mer alone, the output must not depend on how the input is cut
    into batches, on the team size, on the number of handles or on how the file is packed."""
    from conftest import bgzf_bytes, messy_fasta
    src = REPO / "fmsi_amd" / "csrc"
    exe = tmp_path / "cli_stub"
    res = subprocess.run(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", f"-I{REPO / 'include'}",
                          str(REPO / "tests" / "native" / "cli_pipeline_stub.cpp"), *(str(src / f) for f in ("query_cli.cpp", "fastx.cpp", "pgunzip.cpp", "trace.cpp", "host_index.cpp",
                                                                                                            "layout.cpp", "index_writer.cpp")),
                          "-o", str(exe), "-lz", "-lpthread", "-ldl"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    ms = "".join(l.strip() for l in open(GOLDEN / "ref_data" / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13") if not l.startswith(">"))
    rng = random.Random(3)
    fasta = b">first\nACGTTGCA\n" + messy_fasta(ms, rng) + messy_fasta(ms, rng)
    mixed = messy_fasta(ms, rng, fastq_at=150)
    files = {}
    for tag, data in (("fa", fasta), ("fq", mixed)):
        files[tag] = [tmp_path / f"q_{tag}.txt", tmp_path / f"q_{tag}.txt.gz", tmp_path / f"q_{tag}.plain.gz"]
        files[tag][0].write_bytes(data)
        files[tag][1].write_bytes(bgzf_bytes(data, block=900))
        files[tag][2].write_bytes(gzip.compress(data, 1))   # one ordinary gzip member: stretches of the stream, parsed by the team (and stdin below)
    # (atexit_sleep_ms: the sanitizer sleeps a SECOND at exit while other threads live -- the team's pool threads always do, idle, in a pool that is
    # never destroyed -- and this test starts the program some three hundred times)
    env = {**os.environ, "FMSI_HIP_READER_MIN_BYTES": "1", "TSAN_OPTIONS": "halt_on_error=1 atexit_sleep_ms=20", "ASAN_OPTIONS": "detect_leaks=0"}
    prefix = str(GOLDEN / "idx" / "sars_k13")
    # a last batch whose whole text fits a std::string's inline buffer (the ordinary reader's batches hold views into their own text:
    # moving such a batch once left the views dangling -- found by the differential fuzzer, kept here for the sanitizers)
    tiny = tmp_path / "tiny_tail.fa"
    tiny.write_bytes(b">long\n" + ms[:3000].upper().encode() + b"\n>x\nACGTA\n")
    res = subprocess.run([str(exe), "query", "--batch", "1024", "-q", str(tiny), prefix], capture_output=True, env={**env, "FMSI_HIP_SERIAL_READER": "1"})
    assert res.returncode == 0 and res.stdout.endswith(b"\nx\t\n") and res.stdout.startswith(b"long\t"), res.stderr.decode()[-2000:]
    for mode in (["query"], ["query", "-O"], ["query", "-S", "-O"], ["lookup"], ["lookup", "-S"], ["query", "-f", "xor"]):
        for tag, paths in files.items():
            outs = set()
            for q in paths:
                for opts in ([], ["--batch", "2048", "--threads", "3", "--devices", "0,0"], ["--batch", "7000", "--threads", "5"]):
                    res = subprocess.run([str(exe), *mode, *opts, "-q", str(q), prefix], capture_output=True, env=env)
                    assert res.returncode == 0, (mode, q.name, opts, res.stderr.decode()[-3000:])
                    outs.add(res.stdout)
            res = subprocess.run([str(exe), *mode, "-q", str(paths[0]), prefix], capture_output=True, env={**env, "FMSI_HIP_SERIAL_READER": "1"})
            assert res.returncode == 0
            outs.add(res.stdout)
            for opts in ([], ["--batch", "3000", "--threads", "4"]):   # the same bytes through a pipe
                res = subprocess.run([str(exe), *mode, *opts, "-q", "-", prefix], input=paths[0].read_bytes(), capture_output=True, env=env)
                assert res.returncode == 0, (mode, tag, opts, res.stderr.decode()[-2000:])
                outs.add(res.stdout)
            # the ordinary gzip member again, inflated by the team in MANY chunks (pgunzip.h: block starts found inside the stream, chunks decoded
            # with an unknown window, zlib from the first known one), and with that reader switched off
            for extra in ({"FMSI_HIP_PGZ_CHUNK_BYTES": "1500"}, {"FMSI_HIP_PGZ_CHUNK_BYTES": "6000"}, {"FMSI_HIP_NO_PARALLEL_GUNZIP": "1"}):
                res = subprocess.run([str(exe), *mode, "--batch", "5000", "--threads", "6", "--stats", "-q", str(paths[2]), prefix], capture_output=True, env={**env, **extra})
                assert res.returncode == 0, (mode, tag, extra, res.stderr.decode()[-2000:])
                outs.add(res.stdout)
                stats = json.loads([l for l in res.stderr.decode().splitlines() if l.startswith("{")][-1])
                if "FMSI_HIP_PGZ_CHUNK_BYTES" in extra and tag == "fa":
                    assert stats["parallel_gunzip_bytes"] == len(paths[0].read_bytes()) and stats["parallel_gunzip_rounds"] >= 2 and not stats["parallel_gunzip_gave_up"], stats
                elif "FMSI_HIP_PGZ_CHUNK_BYTES" in extra:   # (the file with FASTQ records in it: the stretches stop at the first of them and the ordinary reader goes on)
                    assert 0 < stats["parallel_gunzip_bytes"] <= len(paths[0].read_bytes()) and not stats["parallel_gunzip_gave_up"], stats
                else:
                    assert stats["parallel_gunzip_bytes"] == 0
            assert len(outs) == 1 and len(res.stdout) > 10000, (mode, tag)
    # ---- stdout is a regular file: the pages of its first gigabytes are allocated beyond its end while the index loads (OutPrealloc) and what
    # was not written is given back -- the bytes are the pipe's, appended where the caller asked for it, and nothing stays allocated behind them
    want = subprocess.run([str(exe), "query", "-O", "-q", str(files["fa"][0]), prefix], capture_output=True, env=env)
    assert want.returncode == 0 and len(want.stdout) > 10000
    for how in ("truncate", "append", "off", "gzip", "bgzf"):
        target = tmp_path / f"out_{how}.txt"
        head = b"what was there\n" if how == "append" else b""
        target.write_bytes(head)
        source = files["fa"][2] if how == "gzip" else files["fa"][1] if how == "bgzf" else files["fa"][0]   # (a compressed file: its text's length estimated from its first megabytes)
        with open(target, "ab" if how == "append" else "wb") as so:
            res = subprocess.run([str(exe), "query", "-O", "-q", str(source), prefix], stdout=so, stderr=subprocess.PIPE,
                                 env={**env, **({"FMSI_HIP_NO_OUT_PREALLOC": "1"} if how == "off" else {})})
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        assert target.read_bytes() == head + want.stdout, how
        st = os.stat(target)
        assert st.st_blocks * 512 <= st.st_size + (1 << 20), (how, st.st_blocks * 512, st.st_size)   # (no gigabyte of unwritten pages left behind the end)
    # ---- gzip files that are not what the team's reader takes -- a wrong CRC, a stream cut short, bytes behind the member, a flipped byte inside --
    # give what zlib alone gives for them (the reader hands over at the last round that stood)
    good = gzip.compress(fasta, 6)
    broken = {"crc": good[:-8] + bytes([good[-8] ^ 1]) + good[-7:], "cut": good[:len(good) * 2 // 3], "tail": good + b"junk" * 10,
              "flip": good[:len(good) // 2] + bytes([good[len(good) // 2] ^ 0x40]) + good[len(good) // 2 + 1:], "two": gzip.compress(fasta[:40000], 9) + gzip.compress(fasta[40000:], 1)}
    for tag, data in broken.items():
        q = tmp_path / f"broken_{tag}.gz"
        q.write_bytes(data)
        outs = {}
        for name, extra in (("zlib", {"FMSI_HIP_NO_PARALLEL_GUNZIP": "1"}), ("one", {}), ("many", {"FMSI_HIP_PGZ_CHUNK_BYTES": "2000"})):
            res = subprocess.run([str(exe), "query", "-O", "--batch", "4000", "--threads", "5", "--stats", "-q", str(q), prefix], capture_output=True, env={**env, **extra})
            outs[name] = (res.returncode, res.stdout)
            stats = json.loads([l for l in res.stderr.decode().splitlines() if l.startswith("{")][-1])
            if name != "zlib":
                assert stats["parallel_gunzip_gave_up"] == (tag != "two"), (tag, name, stats)
        # (a stream that FAILS -- crc, cut, flip -- loses zlib's last buffer of output, which at this size is all of it: the rounds that stood before
        # the failing one were handed out, as the reference's 16-KiB reads hand out what precedes the failing one, src/kseq.h:75-93)
        assert outs["zlib"] == outs["one"] and outs["many"][0] == outs["zlib"][0] == 0, tag
        whole_lines = outs["zlib"][1][:outs["zlib"][1].rfind(b"\n") + 1]
        assert outs["many"][1].startswith(whole_lines), tag
        if tag in ("tail", "two"):
            assert outs["many"] == outs["zlib"] and len(outs["zlib"][1]) > 1000, tag
        else:
            assert len(outs["many"][1]) > 1000, tag
    # ---- records of ONE k-mer each (cli_reader.h: KmerPart): the parser packs the k-mer and writes the record's line, the answer is filled
    # in behind the device.  The output is the general path's (FMSI_HIP_CLI_KMER_BATCHES=0), whatever the batch size and the team, for
    # files of that shape alone, with header comments, without the last newline -- and for files where a record of another shape sends
    # its range back to the ordinary rules (CRLF, N, a short or long or two-line sequence, a blank line, a FASTQ record, a lower-case k-mer
    # is fine): some ranges are k-mer batches then, some are not, and the predictor's state crosses them.
    k = 13
    up = ms.upper()

    def kmer_records(n, odd_every=0, tail_newline=True):
        recs = []
        for i in range(n):
            p0 = rng.randrange(0, len(up) - k)
            s = up[p0:p0 + k] if rng.random() < 0.5 else "".join(rng.choice("ACGT") for _ in range(k))
            if rng.random() < 0.1:
                s = s.lower()
            head = f"{i}" if rng.random() < 0.7 else f"name_{i} a comment >x @y\tz"
            if i == 7:
                head = ""
            rec = f">{head}\n{s}\n"
            if odd_every and i % odd_every == odd_every - 1:
                rec = rng.choice([f">{head}\r\n{s}\r\n", f">{head}\n{s[:5]}N{s[6:]}\n", f">{head}\n{s[:-1]}\n", f">{head}\n{s}A\n", f">{head}\n{s[:6]}\n{s[6:]}\n",
                                  f">{head}\n{s}\n\n", f"@{head}\n{s}\n+\n{'I' * k}\n", f">{head}\n\n", f">{head}\v\n{s}\n"])
            recs.append(rec)
        text = "".join(recs)
        return (text if tail_newline else text.rstrip("\n")).encode()

    kfiles = {"pure": kmer_records(6000), "no_tail": kmer_records(3000, tail_newline=False), "odd": kmer_records(6000, odd_every=997), "odd_often": kmer_records(3000, odd_every=40)}
    for tag, data in kfiles.items():
        plain, gz, zgz = tmp_path / f"k_{tag}.fa", tmp_path / f"k_{tag}.fa.gz", tmp_path / f"k_{tag}.plain.gz"
        plain.write_bytes(data)
        gz.write_bytes(bgzf_bytes(data, block=3000))
        zgz.write_bytes(gzip.compress(data, 1))
        for mode in (["query"], ["query", "-O"], ["query", "-S", "-O"], ["lookup"]):
            general = subprocess.run([str(exe), *mode, "-q", str(plain), prefix], capture_output=True, env={**env, "FMSI_HIP_CLI_KMER_BATCHES": "0"})
            assert general.returncode == 0 and general.stdout.count(b"\n") >= 3000, (tag, mode, general.stderr.decode()[-2000:])
            for q in (plain, gz, zgz, "-"):
                for opts in ([], ["--batch", "4096", "--threads", "3"], ["--batch", "30000", "--threads", "7", "--devices", "0,0"]):
                    res = subprocess.run([str(exe), *mode, *opts, "--stats", "-q", str(q), prefix], capture_output=True, env=env, input=data if q == "-" else None)
                    assert res.returncode == 0, (tag, mode, opts, res.stderr.decode()[-3000:])
                    assert res.stdout == general.stdout, (tag, mode, str(q), opts)
                    stats = json.loads([l for l in res.stderr.decode().splitlines() if l.startswith("{")][-1])
                    if mode[0] != "query":
                        assert stats["kmer_batches"] == 0
                    elif tag in ("pure", "no_tail") or (tag == "odd" and opts and opts[1] == "4096"):
                        assert stats["kmer_batches"] > 0, (tag, mode, opts)   # (the path was taken: at least by the ranges without an odd record)


def test_threaded_rrr_decode_under_tsan_and_corrupt_mask(oracle_built, tmp_path):
    """The threaded part of the RRR-63 decoder (>= 64 superblock pairs: the golden indexes are too small to reach it) under
    ThreadSanitizer -- neighbouring threads must not touch each other's boundary word -- and equal to the oracle's mask; a
    .mask file whose offset pointers / class widths run past the offset vector is an error (FMSI_HIP_EIO), not an
    out-of-bounds read (AddressSanitizer build)."""
    from pathlib import Path

    src = REPO / "fmsi_amd" / "csrc"
    ms = synthetic_ms(1_300_000, 31, 11, mask="mixed")     # (enough words for several threads in every threaded loop)
    orc = oracle_py.Oracle.construct(ms, 31, use_klcp=False)
    prefix = tmp_path / "big"
    orc.dump(str(prefix))
    fasta = tmp_path / "q.fa"
    fasta.write_text(">r\nACGTACGTACGTACGTACGTACGTACGTACGTACG\n")
    srcs = [str(REPO / "tests" / "native" / "host_sanitize.cpp"), str(src / "host_index.cpp"), str(src / "index_writer.cpp"),
            str(src / "layout.cpp"), str(src / "fastx.cpp")]
    exes = {}
    for tag, san in (("tsan", "-fsanitize=thread"), ("asan", "-fsanitize=address,undefined")):
        exes[tag] = tmp_path / f"host_{tag}"
        res = subprocess.run(["g++", "-std=c++17", "-O1", "-g", san, "-fno-omit-frame-pointer", f"-I{REPO / 'include'}", *srcs, "-o", str(exes[tag]),
                              "-lz", "-lpthread"], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-3000:]
    out = tmp_path / "w_big"
    res = subprocess.run([str(exes["tsan"]), str(prefix), "0", str(out), str(fasta)], capture_output=True, text=True)
    assert res.returncode == 0 and "ThreadSanitizer" not in res.stderr, (res.stdout + res.stderr)[-3000:]
    for part in ("ac_gt", "ac", "gt", "mask", "misc"):   # decoded (threads) and re-encoded: the oracle's bytes
        assert Path(f"{prefix}.fmsi.{part}").read_bytes() == Path(f"{out}.fmsi.{part}").read_bytes(), part
    # corrupt images: (a) the first superblock's offset pointer far past the offsets, (b) the offsets' bit-vector cut short
    good = Path(f"{prefix}.fmsi.mask").read_bytes()
    import struct
    n, bt_bits = struct.unpack_from("<QQ", good, 0)
    off_btnr = 8 + 9 + ((bt_bits + 63) // 64) * 8                     # n | int_vector bt | bit_vector btnr ...
    btnr_bits = struct.unpack_from("<Q", good, off_btnr)[0]
    off_btnrp = off_btnr + 8 + ((btnr_bits + 63) // 64) * 8
    cases = {}
    b = bytearray(good)
    b[off_btnrp + 9: off_btnrp + 17] = b"\xff" * 8                     # every pointer in the first word: all ones
    cases["pointer"] = bytes(b)
    short_words = max(1, ((btnr_bits + 63) // 64) // 2)
    b = bytearray(good[:off_btnr]) + struct.pack("<Q", short_words * 64) + good[off_btnr + 8: off_btnr + 8 + short_words * 8] + \
        good[off_btnrp:]
    cases["short"] = bytes(b)
    for tag, image in cases.items():
        bad = tmp_path / f"bad_{tag}"
        for part in ("ac_gt", "ac", "gt", "misc"):
            Path(f"{bad}.fmsi.{part}").write_bytes(Path(f"{prefix}.fmsi.{part}").read_bytes())
        Path(f"{bad}.fmsi.mask").write_bytes(image)
        res = subprocess.run([str(exes["asan"]), str(bad), "0", str(tmp_path / "w_bad"), str(fasta)], capture_output=True, text=True)
        assert res.returncode == 1 and "load failed" in res.stderr and "Sanitizer" not in res.stderr, (tag, (res.stdout + res.stderr)[-2000:])


def test_layout_of_many_chunks_equals_numpy(product_built):
    """build_layout() at a size where its threads take several chunks each (6.3 M rows: the per-range totals, the scan over
    the ranges and the per-range prefix sums all matter), on random bit-vectors, against a plain numpy computation of the
    planes and of every block's cumulative counts."""
    fm = product_built
    rng = np.random.default_rng(77)
    n = 6_300_001
    sym = rng.integers(0, 4, size=n, dtype=np.uint8)            # BWT symbol per row: 0 A, 1 C, 2 G, 3 T
    maskbits = rng.integers(0, 2, size=n, dtype=np.uint8)
    hi = (sym >> 1).astype(np.uint8)                              # ac_gt: 1 for G/T
    lo = (sym & 1).astype(np.uint8)
    ac = lo[hi == 0]
    gt = lo[hi == 1]

    def words(bits):
        pad = (-len(bits)) % 64
        b = np.concatenate([bits, np.zeros(pad + 64, dtype=np.uint8)])
        return np.packbits(b.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy(), len(bits)

    # (C[] as construct() writes it, fms_index.h:451, the `$` slot counted among the A rows: build_layout checks it against |ac|)
    cnt = [1, int((sym == 0).sum()), int((sym <= 1).sum()), int((sym <= 2).sum())]
    view, keep = fm.view_from_arrays(words(hi), words(ac), words(gt), words(maskbits), cnt, 0, 31)
    nb = int(fm.lib().fmsi_host_layout_blocks(C.byref(view)))
    blocks = np.zeros((nb, 16), dtype=np.uint32)
    mrank = np.zeros(nb + 1, dtype=np.uint32)
    assert fm.lib().fmsi_host_layout_build(C.byref(view), blocks.ctypes.data, mrank.ctypes.data) == 0
    assert nb == (n + 127) // 128 + 1
    rows = nb * 128
    pad = lambda a: np.concatenate([a, np.zeros(rows - n, dtype=np.uint8)])
    onehot = np.zeros((rows, 4), dtype=np.int64)
    onehot[np.arange(n), sym] = 1
    cum = np.concatenate([np.zeros((1, 4), dtype=np.int64), np.cumsum(onehot.reshape(nb, 128, 4).sum(axis=1), axis=0)])
    assert np.array_equal(blocks[:, 0:4].astype(np.int64), cum[:-1])
    plane = lambda bits: np.packbits(pad(bits).reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint32).reshape(nb, 4)
    assert np.array_equal(blocks[:, 4:8], plane(hi))
    assert np.array_equal(blocks[:, 8:12], plane(lo))
    assert np.array_equal(blocks[:, 12:16], plane(maskbits))
    mcum = np.concatenate([[0], np.cumsum(pad(maskbits).reshape(nb, 128).sum(axis=1))])
    assert np.array_equal(mrank.astype(np.int64), mcum)


def test_packed_result_helpers_and_synthetic_generators(product_built):
    """The host-side mirrors the GPU tests and bench.py lean on, without a GPU: FMSI_HIP_OUT_BITS unpacking (bit w & 63 of word w >> 6;
    STRICT: nibble w & 15 of word w >> 4), and the numpy restatements of the synthetic superstring and mask against the library's own
    host functions (fmsi_hip_synthetic_text_word / _mask_bit), which is what lets tests/golden/make_scale_md5.py hand the REAL reference
    the very text the device builder indexes."""
    fm = product_built
    w = np.array([0b1011, 1 << 63], dtype=np.uint64)
    got = fm.unpack_bits(w, 70, 0)
    assert got[:4].tolist() == [1, 1, 0, 1] and got[63] == 0 and got.size == 70 and got[4:63].sum() == 0
    assert fm.unpack_bits(np.array([0xA5], dtype=np.uint64), 3, fm.STRICT).tolist() == [5, 10, 0]
    assert (fm.packed_words(64, 0), fm.packed_words(65, 0), fm.packed_words(16, fm.STRICT), fm.packed_words(17, fm.STRICT)) == (1, 2, 1, 2)
    L = fm.lib()
    words = fm.synthetic_text_words(1000, 77)
    for i in (0, 1, 7, 30):
        assert int(words[i]) == L.fmsi_hip_synthetic_text_word(77, i)
    assert int(words[31]) == L.fmsi_hip_synthetic_text_word(77, 31) & ((1 << (2 * (1000 - 31 * 32))) - 1)    # the last word is cut at the length
    mask = fm.synthetic_mask(3000, 5, 31, 600)
    assert all(int(mask[p]) == (L.fmsi_hip_synthetic_mask_bit(5, p, 600) if p + 31 <= 3000 else 0) for p in range(3000))
    text = fm.synthetic_text(3000, 5, 31, 600)
    assert all(chr(text[p]).isupper() == bool(mask[p]) for p in range(3000)) and text[-30:].islower()
    assert fm.synthetic_text(200, 5, 31)[:170].isupper()


def test_bench_defaults():
    """bench.py's contract defaults (driver: `python bench.py` = N = 1, K = 10, W = 2) and the round-2 defaults: the k-mer dictionary,
    bit-packed results with the copy inside the timed region, the LF-walk legs, chained lookups for the stream workload."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", REPO / "bench.py")
    m = importlib.util.module_from_spec(spec)
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        spec.loader.exec_module(m)
        a = m.parse_args()
        assert (a.gpus, a.steps, a.warmup, a.workload, a.accel, a.byte_results, a.no_lf_leg, a.k) == (1, 10, 2, "kmers", "compact", False, False, 31)
        assert int(a.n) == 2_800_000_000 and int(a.batch) == 1 << 27
        sys.argv = ["bench.py", "--workload", "stream"]
        a = m.parse_args()
        assert a.stream_via == "runs" and a.dict_chain
        sys.argv = ["bench.py", "--workload", "stream", "--accel", "tables"]
        assert m.parse_args().stream_via == "klcp"
    finally:
        sys.argv = argv


def test_bench_compact_line_of_the_final_record_keeps_every_leg():
    """Round 5's record has 22 legs (the kLCP kernel at the CLI's table depth and at 16, the whole-process legs incl. the query file of one
    k-mer per record): its compact form is the committed one, below the limit with every leg, and every leg keeps name / value /
    matches_reference and says what it was measured on -- `hbm_gb` of the handle, or `load_s` and `after_load` of the process."""
    import importlib.util
    import json
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod3", REPO / "bench.py")
    m = importlib.util.module_from_spec(spec)
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    full = json.loads((REPO / "profiles" / "r05m_bench_default_full.json").read_text())
    last = (REPO / "profiles" / "r05m_bench_default_compact.json").read_text().strip()
    assert len(last) <= m.COMPACT_LIMIT and m.compact_line(full) == last
    d = json.loads(last)
    names = [c["name"] for c in d["configs"]]
    assert len(names) == len(full["configs"]) == 22
    for want in ("c4_klcp", "c4_klcp_subst1pct", "c4_klcp_lookup", "c4_klcp_lookup_fast", "c4_klcp_t16", "c4_klcp_k48", "c4_superkmer", "c3_superkmer", "e2e_or", "e2e_all_S",
                 "e2e_isolated_or", "e2e_isolated_all"):
        assert want in names, want
    for c in d["configs"]:
        assert c["matches_reference"] is True and c["value"] > 0, c
        if c["name"].startswith("e2e"):
            assert c["load_s"] > 0 and c["after_load"] > c["value"], c
        else:
            assert c["hbm_gb"] > 0, c
            if not c["name"].startswith("c2"):
                assert 0 < c["hbm_frac"] < 1, c
    assert d["roofline"]["kernel"].startswith("sk_kmers_kernel") and len(d["roofline"]["lf_path"]) == 3 and d["cpu_baseline"]["kind"] == "reference"


def test_bench_compact_line_fits_the_drivers_tail():
    """The driver keeps the last 8 KB of bench.py's stdout and parses the LAST line (round 3's 28-KB line was cut: `parsed: null`).
    bench.compact_line() of round 3's full record -- 13 legs -- and of the same record with twice the legs must stay below 4 KB, be valid
    JSON and carry the contract's keys, `roofline` (with the LF-walk legs) and `cpu_baseline`."""
    import importlib.util
    import json
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod2", REPO / "bench.py")
    m = importlib.util.module_from_spec(spec)
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    full = json.loads((REPO / "profiles" / "r03u_bench_default.json").read_text().splitlines()[0])
    for grow in (1, 2):
        rec = dict(full, configs=full["configs"] * grow)
        s = m.compact_line(rec)
        assert len(s) <= m.COMPACT_LIMIT < 4096 and "\n" not in s
        d = json.loads(s)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                    "roofline", "cpu_baseline", "configs"):
            assert key in d, key
        assert d["config"]["workload"] and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
        assert [l["kernel"] for l in r["lf_path"]] == ["query_pair_kernel<false,false,false>", "query_lockstep_kernel<false>"]
        c = d["cpu_baseline"]
        assert c["kind"] == "reference" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "k-mers/s" and c["sample"]
        assert d["configs"][0]["name"] == "c2" and d["configs"][0]["matches_reference"] is True
        if grow == 1:
            assert len(d["configs"]) == len(full["configs"])
    # a record without legs, counters or baseline (N > 1 ranks print this form)
    bare = {k: v for k, v in full.items() if k != "configs"}
    bare["roofline"], bare["cpu_baseline"] = None, None
    d = json.loads(m.compact_line(bare))
    assert d["roofline"] is None and d["cpu_baseline"] is None and "configs" not in d
