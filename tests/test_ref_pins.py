"""Two pieces of the product held against the REFERENCE ITSELF rather than against this repo's reading of it: the SV
record / VCF row (csrc/sv_record.h vs /root/reference/sv.cpp) and the command line (csrc/cli_options.h vs
/root/reference/config.cpp + its vendored cxxopts.hpp) -- the two reference files that compile from their own sources
(`make -C oracle ref` -> oracle/_ref/libsvdss_ref.so, built where /root/reference is present; the built file travels,
the sources do not).  Skipped where the built file is missing."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.common import ROOT

REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsvdss_ref.so")
PROD_SO = os.path.join(ROOT, "tests", "_prod_shim.so")
PROD_SRC = [os.path.join(ROOT, "tests", "prod_shim.cpp"), os.path.join(ROOT, "svdss_amd", "csrc", "cli_options.h"),
            os.path.join(ROOT, "svdss_amd", "csrc", "sv_record.h")]


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(REF_SO) and os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref is not built (no /root/reference on this machine)")
    if not os.path.exists(PROD_SO) or any(os.path.getmtime(s) > os.path.getmtime(PROD_SO) for s in PROD_SRC):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", PROD_SO, PROD_SRC[0]], check=True)
    return C.CDLL(REF_SO), C.CDLL(PROD_SO)


def _row(lib, fn, *a):
    out = C.create_string_buffer(1 << 20)
    ty, chrom, s, refall, altall, w, cov, ngaps, score, imprecise, l, cigar, reads = a
    n = getattr(lib, fn)(ty.encode(), chrom.encode(), C.c_uint(s), refall.encode(), altall.encode(), C.c_uint(w), C.c_uint(cov),
                         C.c_int(ngaps), C.c_int(score), C.c_int(imprecise), C.c_uint(l), cigar.encode(), reads.encode(), out, len(out))
    assert n >= 0
    return out.value.decode()


def test_sv_rows_and_order_are_the_references(libs):
    ref, prod = libs
    rng = np.random.default_rng(1)
    bases = np.array(list("ACGTN"))
    for k in range(3000):
        ty = "INS" if k % 2 else "DEL"
        chrom = str(rng.choice(["chr1", "chr10", "chr2", "chrX", "1", "HLA-A*01:01", "c"]))
        s = int(rng.integers(1, 250_000_000))
        refall = "".join(rng.choice(bases, size=1 if ty == "INS" else int(rng.integers(1, 400))))
        altall = "".join(rng.choice(bases, size=int(rng.integers(1, 400)))) if ty == "INS" else refall[:1]
        if k % 50 == 0:
            altall = "<INS>" if ty == "INS" else "<DEL>"
        l = int(rng.integers(0, 100000))
        names = "\n".join(f"m64/{int(rng.integers(0, 10**6))}/ccs" for _ in range(int(rng.integers(0, 6))))
        cigar = str(rng.choice([".", "100=5I200=", "10S50=30D7X"]))
        args = (ty, chrom, s, refall, altall, int(rng.integers(0, 60)), int(rng.integers(0, 90)), int(rng.integers(0, 9)),
                int(rng.integers(-5000, 5000)), int(k % 7 == 0), l, cigar, names)
        a, b = _row(ref, "ref_sv_row", *args), _row(prod, "prod_sv_row", *args)
        assert a == b, (args, a, b)
    chroms = ["chr1", "chr10", "chr2", "chrX", "1", "2", "10", "chrM", ""]
    for _ in range(3000):
        ca, cb = str(rng.choice(chroms)), str(rng.choice(chroms))
        sa, sb = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
        assert ref.ref_sv_less(ca.encode(), C.c_uint(sa), cb.encode(), C.c_uint(sb)) == prod.prod_sv_less(ca.encode(), C.c_uint(sa), cb.encode(), C.c_uint(sb))


def _parse(lib, fn, args):
    argv = (C.c_char_p * (len(args) + 1))(*[a.encode() for a in ["SVDSS"] + args])
    out = C.create_string_buffer(1 << 16)
    n = getattr(lib, fn)(len(args) + 1, argv, out, len(out))
    assert n >= 0
    return out.value.decode()


def test_command_lines_parse_as_the_references_cxxopts_does(libs):
    """Option names, defaults, `--opt value` / `--opt=value`, flags with explicit values, numbers that are not numbers,
    unknown options, missing values, positional arguments, the post-processing of config.cpp:87,106 -- field by field and
    error text by error text.  (Left out: --threads below 1, on which the reference divides by zero, and this program's
    own --gpus / --io-threads, which the reference does not have.)"""
    ref, prod = libs
    fixed = [[], ["--bam", "x.bam", "--threads", "3", "--bsize", "1000"], ["--frobnicate"], ["--threads"], ["--bam"],
             ["--min-sv-length", "10", "--noht", "-l", "0.5"], ["--min-sv-length=300"], ["--threads=7", "--accp=0.9"],
             ["--binary", "--append", "q"], ["positional"], ["search", "--index", "i.fmd", "--fastx", "r.fq"], ["--threads", "abc"],
             ["--threads", "0x10"], ["--bsize", "-5"], ["--bsize", "99999999999"], ["--accp", "x"], ["--accp", "0.5x"], ["-l=0.25"],
             ["--noht=false"], ["--noassemble=true"], ["--noputative=maybe"], ["--clipped", "--verbose"], ["-h"], ["--help"],
             ["--version"], ["-x"], ["-"], ["--"], ["--index=a=b"], ["--bam", "--sfs"], ["--omax", "5"], ["--min-mapq", "0"],
             ["--min-cluster-weight", "7"], ["--poa", "p.sam", "--clusters", "c.txt"], ["--bsize", "10", "--threads", "4"],
             ["--threads", "16", "--bsize", "7"], ["--overlap", "3"], ["--noref"], ["--threads", "2", "--threads", "5"]]
    for args in fixed:
        a, b = _parse(ref, "ref_config_parse", args), _parse(prod, "prod_config_parse", args)
        assert a == b, (args, a, b)
        assert not a.startswith("crash"), args
    rng = np.random.default_rng(2)
    strs = ["--bam", "--sfs", "--poa", "--clusters", "--index", "--fastx", "--reference", "--append"]
    ints = ["--bsize", "--omax", "--min-sv-length", "--min-mapq", "--min-cluster-weight"]
    flts = ["--accp", "-l"]
    flags = ["--clipped", "--noht", "--noassemble", "--noputative", "--binary", "--verbose", "--version", "--help"]
    vals = ["x.bam", "7", "0", "25", "300", "-3", "0.5", "1e-2", "abc", "12abc", "0x1f", "", "true", "a b", "99999999999", "--x"]
    for _ in range(350):
        args = []
        if rng.random() < 0.8:
            args += ["--threads", str(int(rng.integers(1, 40)))]
        for _ in range(int(rng.integers(0, 7))):
            kind = rng.random()
            name = str(rng.choice(strs if kind < 0.3 else ints if kind < 0.55 else flts if kind < 0.7 else flags))
            if name in flags:
                args.append(name if rng.random() < 0.8 else name + "=" + str(rng.choice(["true", "false", "1", "0", "T", "F", "no"])))
            elif rng.random() < 0.5:
                args += [name, str(rng.choice(vals))]
            else:
                args.append(name + "=" + str(rng.choice(vals)))
        if rng.random() < 0.1:
            args.insert(int(rng.integers(0, len(args) + 1)), str(rng.choice(["file.txt", "--nope", "-q", "--bam="])))
        rng.shuffle(args) if rng.random() < 0.2 else None
        a, b = _parse(ref, "ref_config_parse", args), _parse(prod, "prod_config_parse", args)
        if a.startswith("crash"):
            continue                                  # (the reference itself died: a shuffled --threads value of 0, ...)
        assert a == b, (args, a, b)
