#!/usr/bin/env python3
"""Build-container check of the index builder against the REAL reference: random small genomes (several contigs, N runs, lower
case, contigs without SNPs, SNP clusters at contig ends so that the window loop picks up the stale array entry of an earlier
group, localPattern.c:239) indexed by oracle/_ref/salt-idx and by salt_amd/bin/salt-idx; every file must be identical except the
documented spots (last .ref nibbles, ONE .R.backward.sa entry).  Usage: tools/idx_stress.py [n_cases] [seed]"""
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "salt-idx")
OURS = os.path.join(ROOT, "salt_amd", "bin", "salt-idx")
FILES = (".R.seedLen", ".C.pac", ".C.ann", ".C.amb", ".C.bwt", ".C.sa", ".C.lkt", ".lp", ".R.backward.bwt", ".R.backward.occ")


def make_case(rng, d):
    n_contigs = rng.randint(1, 6)
    k = rng.choice([12, 15, 19, 21, 25])
    conts, lines = [], []
    for ci in range(n_contigs):
        L = rng.choice([30, 80, 300, 1500, 6000])
        s = [rng.choice("ACGT") for _ in range(L)]
        if rng.random() < 0.4:                                     # N runs (also other IUPAC letters)
            for _ in range(rng.randint(1, 3)):
                a = rng.randrange(L); b = min(L, a + rng.randint(1, 40)); ch = rng.choice("NNNnRY")
                for i in range(a, b):
                    s[i] = ch
        if rng.random() < 0.3:                                     # soft-masked stretch
            a = rng.randrange(L); b = min(L, a + rng.randint(5, 200))
            for i in range(a, b):
                s[i] = s[i].lower()
        seq = "".join(s)
        name = "c%d" % ci
        conts.append((name, seq))
        mode = rng.random()
        if mode < 0.15 and ci > 0:
            continue                                               # contig without SNPs (groups shift to the next contig)
        dens = rng.choice([0.005, 0.02, 0.08, 0.2])
        pos = sorted(set([p for p in range(L) if rng.random() < dens] +
                         ([L - 1 - rng.randint(0, min(L - 1, k))] if rng.random() < 0.7 else []) +
                         ([rng.randint(0, min(L - 1, k))] if rng.random() < 0.5 else [])))
        for p in pos:
            c = seq[p].upper()
            if c not in "ACGT":
                c = "N"
            alts = rng.sample([x for x in "ACGT" if x != c], rng.choice([1, 1, 1, 2, 3]))
            al = sorted(([c] if c != "N" and rng.random() < 0.95 else []) + alts)
            lines.append("%s\t%d\t%s\t%s\n" % (name, p + 1, "/".join(al), c))
    fa, snp = os.path.join(d, "g.fa"), os.path.join(d, "s.txt")
    with open(fa, "w") as f:
        for n, s in conts:
            f.write(">%s\n" % n)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    open(snp, "w").write("".join(lines))
    return fa, snp, k, len(lines)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = skipped = 0
    for case in range(n_cases):
        with tempfile.TemporaryDirectory() as d:
            fa, snp, k, ns = make_case(rng, d)
            if ns == 0:
                skipped += 1
                continue
            r = subprocess.run([REF, "-k", str(k), fa, snp, os.path.join(d, "ref")], capture_output=True)
            o = subprocess.run([OURS, "-k", str(k), fa, snp, os.path.join(d, "our")], capture_output=True)
            if r.returncode != 0:                                  # the reference crashes on some inputs (writes past its arrays)
                skipped += 1
                continue
            if o.returncode != 0:
                print("case %d: ours failed: %s" % (case, o.stderr.decode()[-300:])); bad += 1
                continue
            diffs = []
            for sfx in FILES:
                if open(os.path.join(d, "ref" + sfx), "rb").read() != open(os.path.join(d, "our" + sfx), "rb").read():
                    diffs.append(sfx)
            a, b = np.fromfile(os.path.join(d, "ref.R.backward.sa"), dtype=np.uint32), np.fromfile(os.path.join(d, "our.R.backward.sa"), dtype=np.uint32)
            if len(a) != len(b) or int((a != b).sum()) > 1:
                diffs.append(".R.backward.sa")
            a, b = np.fromfile(os.path.join(d, "ref.ref"), dtype=np.uint32), np.fromfile(os.path.join(d, "our.ref"), dtype=np.uint32)
            l = int(a[0]); live = (1 << (4 * (l % 8))) - 1 if l % 8 else 0xFFFFFFFF
            if len(a) != len(b) or (a[:-1] != b[:-1]).any() or (int(a[-1]) & live) != (int(b[-1]) & live):
                diffs.append(".ref")
            if diffs:
                bad += 1
                keep = "/tmp/idx_stress_case%d" % case
                subprocess.run(["cp", "-r", d, keep])
                print("case %d (k=%d, %d SNPs): DIFFERENT %s  -> %s" % (case, k, ns, diffs, keep))
    print("%d cases, %d skipped (no SNPs / reference crashed), %d different" % (n_cases, skipped, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
