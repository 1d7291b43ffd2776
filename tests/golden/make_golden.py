#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by running the REAL reference.

Run in the build container only (needs /root/reference and oracle/_ref built by
`make -C oracle -f Makefile.ref`); the GPU box never runs this -- it only reads the committed files.

What it writes (all data: inputs + expected outputs, no reference source):
  ref_data/                  copies of the reference's own test DATA files (tests/testfiles/*.fa, queries.txt,
                             result_*_complements*.txt goldens, tests/data/*.ms.k13|k31)
  idx/<name>.fmsi.*          index files produced by the reference's construct()+dump_index() for every data set
  idx/md5.json               md5 of every index part (the k31 .ac_gt etc. equal the values recorded in SURVEY.md 8(c))
  mixed.fa                   synthetic masked superstring with all-ON / all-OFF / mixed mask regions (inverted RRR superblocks)
  edge_queries.fa            edge-case queries (invalid characters, short records, lower case, comments, FASTQ, CRLF)
  rand_<name>.fa             seeded random reads: half substrings of the superstring (half of those reverse-complemented), half uniform
  out/<name>.<mode>.txt      stdout of the reference for each (data set, query file, mode); modes:
                             or, all (-O), or_S (-S), all_S (-S -O), lookup, lookup_S
  kat_fms_index.json         known-answer tables transcribed from the reference's tests/fms_index_test.h (data only)
"""
import hashlib
import json
import random
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
REF_CLI = REPO / "oracle" / "_ref" / "fmsi_ref"

MODES = {
    "or": ["query"], "all": ["query", "-O"], "or_S": ["query", "-S"], "all_S": ["query", "-S", "-O"],
    "lookup": ["lookup"], "lookup_S": ["lookup", "-S"],
}


def run(args, **kw):
    return subprocess.run([str(a) for a in args], check=True, capture_output=True, **kw)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def read_ms(path):
    return "".join(l.strip() for l in open(path) if not l.startswith(">"))


def random_reads(ms, n, length, rng, sub_rate=0.0):
    out = []
    up = ms.upper()
    for i in range(n):
        if rng.random() < 0.5 and len(up) > length:
            p = rng.randrange(0, len(up) - length + 1)
            r = up[p:p + length]
            if rng.random() < 0.5:
                r = revcomp(r)
            if sub_rate:
                r = "".join(rng.choice("ACGT") if rng.random() < sub_rate else ch for ch in r)
        else:
            r = "".join(rng.choice("ACGT") for _ in range(length))
        out.append(f">r{i}\n{r}\n")
    return "".join(out)


def main():
    if not REF_CLI.exists():
        sys.exit("build oracle/_ref first: make -C oracle -f Makefile.ref")
    rd = HERE / "ref_data"
    idx = HERE / "idx"
    out = HERE / "out"
    for d in (rd, idx, out):
        d.mkdir(exist_ok=True)

    # 1. the reference's own test data
    for f in ["integration_a.fa", "integration_b.fa", "queries.txt", "result_a_complements.txt", "result_a_complements_hash.txt",
              "result_b_complements.txt", "result_b_complements_xor.txt", "result_merged_complements.txt"]:
        shutil.copy(REF / "tests" / "testfiles" / f, rd / f)
    for f in ["GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13", "GCF_test2_ones.fna.ms.k31"]:
        shutil.copy(REF / "tests" / "data" / f, rd / f)
    for p in rd.iterdir():
        p.chmod(0o644)

    # 2. synthetic mixed-density mask superstring (k = 11): long all-ON, all-OFF and random-mask stretches
    rng = random.Random(20251002)
    k_mixed = 11
    parts = []
    for seg in range(24):
        length = rng.choice([1500, 2500, 4100])
        s = "".join(rng.choice("ACGT") for _ in range(length))
        kind = seg % 4
        if kind == 0:
            s = s.upper()
        elif kind == 1:
            s = s.lower()
        elif kind == 2:
            s = "".join(ch.upper() if rng.random() < 0.5 else ch.lower() for ch in s)
        else:
            s = "".join(ch.upper() if rng.random() < 0.93 else ch.lower() for ch in s)
        parts.append(s)
    ms = "".join(parts)
    ms = ms[:-(k_mixed - 1)] + ms[-(k_mixed - 1):].lower()
    ms = ms[:-k_mixed] + ms[-k_mixed].upper() + ms[-(k_mixed - 1):]
    (HERE / "mixed.fa").write_text(f">mixed k={k_mixed}\n{ms}\n")

    datasets = {
        "integration_a": (rd / "integration_a.fa", 3),
        "integration_b": (rd / "integration_b.fa", 3),
        "sars_k13": (rd / "GCF_009858895.2_ASM985889v3_genomic.fna.ms.k13", 13),
        "sars_k31": (rd / "GCF_test2_ones.fna.ms.k31", 31),
        "mixed_k11": (HERE / "mixed.fa", k_mixed),
    }

    # 3. index every data set with the reference
    md5 = {}
    for name, (src, k) in datasets.items():
        dst = idx / name
        shutil.copy(src, dst)
        run([REF_CLI, "index", dst])
        dst.unlink()
        for part in sorted(idx.glob(name + ".fmsi.*")):
            md5[part.name] = hashlib.md5(part.read_bytes()).hexdigest()
    # 3b. index-only fixtures for `fmsi index` with k > 32 (the reference switches to __uint128_t k-mers for the kLCP,
    # src/main.cpp:229-230): a superstring with planted repeats of 20..90 bases so that the kLCP differs between k.
    rng = random.Random(7)
    segs, text = [], ""
    for _ in range(160):
        if segs and rng.random() < 0.55:
            src_seg = rng.choice(segs)
            a = rng.randrange(0, len(src_seg) - 20)
            piece = src_seg[a:a + rng.randrange(20, 91)]
            if rng.random() < 0.3:
                piece = revcomp(piece)
        else:
            piece = "".join(rng.choice("ACGT") for _ in range(rng.randrange(60, 200)))
            segs.append(piece)
        text += piece
    k_rep = 40
    text = "".join(ch if rng.random() < 0.8 else ch.lower() for ch in text[:-k_rep]) + text[-k_rep] + text[-(k_rep - 1):].lower()
    (HERE / "repeats.fa").write_text(f">repeats k={k_rep}\n{text}\n")
    for name, extra in (("repeats_k40", []), ("repeats_k33", ["-k", 33]), ("repeats_k34", ["-k", 34]), ("repeats_k64", ["-k", 64]),
                        ("repeats_k40x", ["-x"])):
        dst = idx / name
        shutil.copy(HERE / "repeats.fa", dst)
        run([REF_CLI, "index", *extra, dst])
        dst.unlink()
        for part in sorted(idx.glob(name + ".fmsi.*")):
            md5[part.name] = hashlib.md5(part.read_bytes()).hexdigest()
    assert not (idx / "repeats_k40x.fmsi.klcp").exists()
    assert len({md5[f"repeats_k{k}.fmsi.klcp"] for k in (33, 34, 40, 64)}) == 4
    (idx / "md5.json").write_text(json.dumps(md5, indent=1, sort_keys=True) + "\n")

    # 3c. `fmsi export` of every golden index gives the masked superstring back, case included (export_ms,
    # src/fms_index.h:462-473) -- so the superstring files themselves are the goldens of export
    for name, (src, k) in {**datasets, "repeats_k40": (HERE / "repeats.fa", 40)}.items():
        res = run([REF_CLI, "export", idx / name])
        assert res.stdout.decode() == ">exported masked superstring\n" + read_ms(src) + "\n", name

    # 4. query files
    edge = (
        ">e0\nACGN\n>e1\nACNAC\n>e2\nAC\n>e3 lower and invalid\nacgtNNacgta\n>e4\n\n>e5\nNACG\n>e6\nACGTACGTACG\n"
        ">e7 multi\nACGTAC\nGTACGT\nAC\n>e8\nNNNN\n>e9\nACGTNACGTNNACGTACGTAXCGT\n@fq1 comment\nACGTACG\n+\nIIIIIII\n"
        ">crlf\r\nACGTA\r\nCGT\r\n>tail\nTACGTAC"
    )
    (HERE / "edge_queries.fa").write_text(edge)
    qfiles = {"integration_a": ["queries", "edge"], "integration_b": ["queries", "edge"]}
    rng = random.Random(42)
    datasets["repeats_k40"] = (HERE / "repeats.fa", 40)
    datasets["repeats_k64"] = (HERE / "repeats.fa", 64)
    for name, length, n in (("sars_k13", 62, 300), ("sars_k31", 90, 300), ("mixed_k11", 200, 300), ("repeats_k40", 150, 200),
                            ("repeats_k64", 200, 150)):
        src, k = datasets[name]
        text = random_reads(read_ms(src), n, length, rng, sub_rate=0.01 if name == "mixed_k11" else 0.0)
        # a few long reads and some with invalid characters
        up = read_ms(src).upper()
        text += f">long0\n{up[1000:3500]}\n>long1\n{revcomp(up[5000:7700])}\n>longN\n{up[100:400]}N{up[400:450]}NN{up[450:900]}\n"
        if name.startswith("repeats"):   # reads shorter than k, exactly k, and k +- 1
            text += "".join(f">short{d}\n{up[700:700 + k + d]}\n" for d in (-1, 0, 1)) + f">tiny\n{up[10:40]}\n"
        (HERE / f"rand_{name}.fa").write_text(text)
        qfiles[name] = ["rand"]

    # 5. reference outputs
    for name, kinds in qfiles.items():
        _, k = datasets[name]
        for kind in kinds:
            qf = {"queries": rd / "queries.txt", "edge": HERE / "edge_queries.fa", "rand": HERE / f"rand_{name}.fa"}[kind]
            for mode, args in MODES.items():
                res = run([REF_CLI, *args, "-k", k, "-q", qf, idx / name])
                (out / f"{name}.{kind}.{mode}.txt").write_bytes(res.stdout)

    # the reference's committed goldens must equal what the harness produced (pins the harness' record loop)
    assert (out / "integration_a.queries.or.txt").read_bytes() == (rd / "result_a_complements.txt").read_bytes()
    assert (out / "integration_a.queries.lookup.txt").read_bytes() == (rd / "result_a_complements_hash.txt").read_bytes()
    assert (out / "integration_b.queries.or.txt").read_bytes() == (rd / "result_b_complements.txt").read_bytes()
    print("golden fixtures written; index md5:")
    for k_, v in sorted(md5.items()):
        print(" ", k_, v)


if __name__ == "__main__":
    main()
