#!/usr/bin/env python3
"""Adds the ragged-length fixture to tests/golden/lambda/ (run in the BUILD container only, after make_fixtures.py).

The main fixture's reads are all 100 bp; here one FASTQ mixes lengths from 19 (= k: a single seed) to 300 bp, and the pairs
mix 75 / 100 / 125 / 150 bp mates, so that the length-dependent parts of the reference are pinned by its own output:
seed slot count, gapped max_diff = L/10 (alnse.c:1090), the LV text window L+4, SSW maskLen = L/2, soft clips.
Same read models as make_fixtures.py (plain / noisy / indel / withN / manyN / junk).
Outputs: reads_ragged.fq, expect_ragged_default.sam (-d -c), expect_ragged_r7_s10.sam (-d -c -r 7 -s 10),
         reads_ragged_pe_[12].fq, expect_ragged_pe.sam (-d -p -c -a 300 -b 700); @PG line stripped.
"""
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import REF_BIN, OUT, K, KINDS, revcomp, sim_read, strip_pg      # noqa: E402
from make_span_fixture import contigs                                             # noqa: E402

LENS = [19, 20, 25, 36, 50, 75, 100, 101, 120, 121, 129, 130, 150, 151, 160, 161, 200, 250, 300]


def main():
    rng = random.Random(20261006)
    genome = contigs(os.path.join(OUT, "genome.fa"))
    snp_map = {}
    names = [n for n, _ in genome]
    for line in open(os.path.join(OUT, "snps.txt")):
        c, p, al, _ = line.split()
        snp_map[(names.index(c), int(p) - 1)] = al.split("/")
    fq = os.path.join(OUT, "reads_ragged.fq")
    with open(fq, "w") as f:
        for i in range(700):
            L = rng.choice(LENS)
            kind = rng.choice(KINDS)
            if L < 40 and kind in ("indel", "indel2", "manyN"):
                kind = "plain"
            name, pos, strand, read = sim_read(rng, genome, snp_map, L, kind)
            f.write("@g%d_%s_%d_%s_%s_L%d\n%s\n+\n%s\n" % (i, name, pos + 1, "-" if strand else "+", kind, L, read, "H" * L))
    p1, p2 = os.path.join(OUT, "reads_ragged_pe_1.fq"), os.path.join(OUT, "reads_ragged_pe_2.fq")
    with open(p1, "w") as f1, open(p2, "w") as f2:
        for i in range(300):
            ci = rng.randrange(len(genome))
            name, s = genome[ci]
            l1, l2 = rng.choice([75, 100, 125, 150]), rng.choice([75, 100, 125, 150])
            isz = max(l1 + l2 + 10, int(rng.gauss(500, 50)))
            pos = rng.randrange(0, len(s) - isz - 8)
            frag = list(s[pos:pos + isz])
            for j in range(len(frag)):
                al = snp_map.get((ci, pos + j))
                if al is not None:
                    frag[j] = rng.choice(al)
                if frag[j] == "N":
                    frag[j] = rng.choice("ACGT")
                if rng.random() < 0.005:
                    frag[j] = rng.choice([c for c in "ACGT" if c != frag[j]])
            kind = rng.choice(["plain"] * 80 + ["damaged"] * 14 + ["junk1"] * 4 + ["junk"] * 2)
            m1, m2f = frag[:l1], frag[isz - l2:]
            if kind == "damaged":                                 # the second mate only rescuable by SW
                for j in range(len(m2f)):
                    if rng.random() < 0.09:
                        m2f[j] = rng.choice([c for c in "ACGT" if c != m2f[j]])
                p = rng.randrange(20, l2 - 20)
                del m2f[p:p + 2]
                m2f.extend(rng.choice("ACGT") for _ in range(2))
            if kind in ("junk", "junk1"):
                m2f = [rng.choice("ACGT") for _ in range(l2)]
            if kind == "junk":
                m1 = [rng.choice("ACGT") for _ in range(l1)]
            r1, r2 = "".join(m1), revcomp("".join(m2f))
            if rng.random() < 0.5:
                r1, r2 = r2, r1
            f1.write("@q%d_%s_%d_%d_%s/1\n%s\n+\n%s\n" % (i, name, pos + 1, isz, kind, r1, "I" * len(r1)))
            f2.write("@q%d_%s_%d_%d_%s/2\n%s\n+\n%s\n" % (i, name, pos + 1, isz, kind, r2, "I" * len(r2)))
    with tempfile.TemporaryDirectory() as tmp:
        idx = os.path.join(tmp, "idx")
        subprocess.run([os.path.join(REF_BIN, "salt-idx"), "-k", str(K), os.path.join(OUT, "genome.fa"), os.path.join(OUT, "snps.txt"), idx],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for case, args, files in (("ragged_default", ["-d", "-c"], [fq]), ("ragged_r7_s10", ["-d", "-c", "-r", "7", "-s", "10"], [fq]),
                                  ("ragged_pe", ["-d", "-p", "-c", "-a", "300", "-b", "700"], [p1, p2])):
            sam = os.path.join(tmp, "o.sam")
            with open(sam, "w") as g:
                subprocess.run([os.path.join(REF_BIN, "salt")] + args + [idx] + files, check=True, stdout=g, stderr=subprocess.DEVNULL)
            strip_pg(sam, os.path.join(OUT, "expect_%s.sam" % case))


if __name__ == "__main__":
    main()
