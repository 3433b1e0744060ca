#!/usr/bin/env python3
"""Adds the contig-boundary fixture to tests/golden/lambda/ (run in the BUILD container only, after make_fixtures.py).

Reads the committed genome.fa / snps.txt, builds the index in /tmp with the real reference (oracle/_ref/salt-idx) and asks
the real reference (oracle/_ref/salt) for the SAM of reads that a simulator never produces but real data holds:
  * reads that straddle the lambdaA | lambdaB_div2pct boundary of the concatenated genome (every 3rd offset, both strands,
    some with one substitution);
  * reads hanging over the start of the genome (junk head + the first bases) and over its end (last bases + junk tail):
    candidate positions wrap below 0 / run past mixRef.l (alnse.c:678-719 range checks, u32 arithmetic);
  * reads lying exactly at the first / last 100 bases of each contig.
Outputs: reads_span.fq, expect_span_default.sam (-d -c), expect_span_r5.sam (-d -c -r 5), @PG line stripped.
"""
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import REF_BIN, OUT, K, read_fa, revcomp, strip_pg      # noqa: E402

L = 100


def contigs(path):
    out, name, seq = [], None, []
    for line in open(path):
        if line.startswith(">"):
            if name is not None:
                out.append((name, "".join(seq)))
            name, seq = line[1:].strip(), []
        else:
            seq.append(line.strip().upper())
    out.append((name, "".join(seq)))
    return out


def main():
    rng = random.Random(20261005)
    cs = contigs(os.path.join(OUT, "genome.fa"))
    cat = "".join(s for _, s in cs)
    la = len(cs[0][1])
    fix = lambda s: "".join(c if c in "ACGT" else rng.choice("ACGT") for c in s)
    junk = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    reads = []
    for o in range(-(L - 1), 0, 3):                           # straddling the contig boundary
        r = fix(cat[la + o:la + o + L])
        if o % 2 == 0:
            p = rng.randrange(L)
            r = r[:p] + rng.choice([c for c in "ACGT" if c != r[p]]) + r[p + 1:]
        reads.append(("span_%d" % (la + o + 1), r))
    for h in (5, 20, 21, 40, 60, 79, 80):                     # over the start / the end of the genome
        reads.append(("head_junk%d" % h, junk(h) + fix(cat[:L - h])))
        reads.append(("tail_junk%d" % h, fix(cat[len(cat) - (L - h):]) + junk(h)))
    off = 0
    for name, s in cs:                                        # exactly at contig ends
        reads.append(("first_%s" % name, fix(cat[off:off + L])))
        reads.append(("last_%s" % name, fix(cat[off + len(s) - L:off + len(s)])))
        off += len(s)
    fq = os.path.join(OUT, "reads_span.fq")
    with open(fq, "w") as f:
        for i, (nm, r) in enumerate(reads):
            if i % 2:
                r = revcomp(r)
            f.write("@s%d_%s_%s\n%s\n+\n%s\n" % (i, nm, "-" if i % 2 else "+", r, "I" * L))
    with tempfile.TemporaryDirectory() as tmp:
        idx = os.path.join(tmp, "idx")
        subprocess.run([os.path.join(REF_BIN, "salt-idx"), "-k", str(K), os.path.join(OUT, "genome.fa"), os.path.join(OUT, "snps.txt"), idx],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for case, args in (("span_default", ["-d", "-c"]), ("span_r5", ["-d", "-c", "-r", "5"])):
            sam = os.path.join(tmp, "o.sam")
            with open(sam, "w") as g:
                subprocess.run([os.path.join(REF_BIN, "salt")] + args + [idx, fq], check=True, stdout=g, stderr=subprocess.DEVNULL)
            strip_pg(sam, os.path.join(OUT, "expect_%s.sam" % case))
    print("%d reads" % len(reads))


def make_wrapped():
    """reads_wrapped.fq: the first 60 reads of reads_se.fq with sequence and quality wrapped at 37 characters and CRLF line
    ends on every third record (kseq.h reads such records; the reference's SAM for them is the fixture)."""
    lines = open(os.path.join(OUT, "reads_se.fq")).read().split("\n")
    fq = os.path.join(OUT, "reads_wrapped.fq")
    with open(fq, "w", newline="") as f:
        for i in range(60):
            name, seq, _, qual = lines[4 * i:4 * i + 4]
            nl = "\r\n" if i % 3 == 2 else "\n"
            wrap = lambda x: nl.join(x[j:j + 37] for j in range(0, len(x), 37))
            f.write(name + nl + wrap(seq) + nl + "+" + nl + wrap(qual) + nl)
    with tempfile.TemporaryDirectory() as tmp:
        idx = os.path.join(tmp, "idx")
        subprocess.run([os.path.join(REF_BIN, "salt-idx"), "-k", str(K), os.path.join(OUT, "genome.fa"), os.path.join(OUT, "snps.txt"), idx],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sam = os.path.join(tmp, "o.sam")
        with open(sam, "w") as g:
            subprocess.run([os.path.join(REF_BIN, "salt"), "-d", "-c", idx, fq], check=True, stdout=g, stderr=subprocess.DEVNULL)
        strip_pg(sam, os.path.join(OUT, "expect_wrapped.sam"))


if __name__ == "__main__":
    if "--wrapped" in sys.argv:
        make_wrapped()
    else:
        main()
