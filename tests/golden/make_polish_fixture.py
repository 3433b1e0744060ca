#!/usr/bin/env python3
"""Golden outputs of the reference's `polish` (row N4; run in the BUILD container only): oracle/_ref/polish, compiled in place from
/root/reference/Polish_src by oracle/Makefile, on SAM the reference's `salt` wrote for the lambda fixture.

  expect_polish_se_lv.sam     polish     idx expect_se_default.sam          (Landau-Vishkin re-scoring, the default)
  expect_polish_se_r1_lv.sam  polish     idx expect_se_r1_m500.sam          (more alternative hits per read)
  expect_polish_se_sw.sam     polish -s  idx expect_se_default.sam          (Smith-Waterman re-scoring)
  expect_polish_pe_lv.sam     polish -p  idx <expect_pe_default.sam>        (pairs; without the empty line the reference's `salt -p` prints after
  expect_polish_pe_sw.sam     polish -p -s ...                              every record: `polish` stops at the first empty line, samParser.c:87-90)
Inputs go through polish_input() below (reads with N dropped: the reference's output for them is undefined).
"""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
POLISH = os.path.join(ROOT, "oracle", "_ref", "polish")
L = os.path.join(HERE, "lambda")


def polish_input(sam_path, paired):
    """The SAM `polish` is given: header kept, the empty lines of `salt -p` dropped, and reads with a base other than A C G T
    dropped (pairs as a whole): for those the reference prints "ACGT"[4] (a NUL byte) or indexes past that literal (memory garbage,
    newlines included) -- undefined output nothing can be compared with.  Used by this script and by the tests alike."""
    lines = [l for l in open(sam_path, "rb").read().split(b"\n") if l.strip()]
    hdr = [l for l in lines if l.startswith(b"@")]
    rec = [l for l in lines if not l.startswith(b"@")]
    clean = lambda l: set(l.split(b"\t")[9]) <= set(b"ACGT")
    if paired:
        keep = []
        for i in range(0, len(rec) - 1, 2):
            if clean(rec[i]) and clean(rec[i + 1]):
                keep += [rec[i], rec[i + 1]]
        rec = keep
    else:
        rec = [l for l in rec if clean(l)]
    return b"\n".join(hdr + rec) + b"\n"


CASES = (("expect_polish_se_lv.sam", [], "expect_se_default.sam"), ("expect_polish_se_r1_lv.sam", [], "expect_se_r1_m500.sam"),
         ("expect_polish_se_sw.sam", ["-s"], "expect_se_default.sam"), ("expect_polish_pe_lv.sam", ["-p"], "expect_pe_default.sam"),
         ("expect_polish_pe_sw.sam", ["-p", "-s"], "expect_pe_default.sam"))


EDGE_CASES = (("expect_polish_edge_se.sam", []), ("expect_polish_edge_pe.sam", ["-p"]),
              ("expect_polish_edge_se_sw.sam", ["-s"]), ("expect_polish_edge_pe_sw.sam", ["-p", "-s"]))


def main():
    idx = os.path.join(L, "idx")
    for f in os.listdir(L):
        if f.startswith("expect_polish") or f in ("polish_pe_in.sam", "polish_edge_in.sam"):
            os.unlink(os.path.join(L, f))
    for out, args, src in CASES:
        with tempfile.NamedTemporaryFile("wb", suffix=".sam", delete=False) as t:
            t.write(polish_input(os.path.join(L, src), "-p" in args))
        with open(os.path.join(L, out), "wb") as f:
            p = subprocess.run([POLISH] + args + [idx, t.name], stdout=f, stderr=subprocess.PIPE)
        os.unlink(t.name)
        data = open(os.path.join(L, out), "rb").read()
        print(out, "rc", p.returncode, len(data), "records", data.count(b"\n"), "NUL bytes", data.count(b"\0"), p.stderr.decode()[-200:])
    # hand-made edge cases (tests/polish_edge.py): many XA hits, both contigs, hits clipped at the genome end
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from polish_edge import edge_records
    src = os.path.join(L, "polish_edge_in.sam")
    open(src, "w").write("@HD\tVN:1\n" + "\n".join(edge_records()) + "\n")
    for out, args in EDGE_CASES:
        with open(os.path.join(L, out), "wb") as f:
            p = subprocess.run([POLISH] + args + [idx, src], stdout=f, stderr=subprocess.PIPE)
        print(out, "rc", p.returncode, os.path.getsize(os.path.join(L, out)), p.stderr.decode()[-200:])


if __name__ == "__main__":
    main()
