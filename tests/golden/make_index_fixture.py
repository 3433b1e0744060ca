#!/usr/bin/env python3
"""Index-builder fixtures beyond the lambda set (run in the BUILD container only): tiny genomes cut from the committed lambda
FASTA with SNP sets that stress `salt-idx`, indexed by the REAL reference (oracle/_ref/salt-idx -k 19); the files it wrote are
the expected output of salt_amd/bin/salt-idx (tests/test_index_builder.py).  `.C.lkt` (64 MiB) is kept as a sha256.

  two_contigs      SNPs in both contigs
  gap_short        a 15-base contig WITHOUT SNPs between two with: the reference matches SNP groups to contigs by ORDER
                   (mixRef.c:149-152), writes the third contig's SNP past the second contig's end and then wipes it
  all_snp_contigs  three contigs, each with a SNP, incl. multi-allelic and one 3 bases from a contig end
  dense            a 3 kb contig with a SNP every ~7 bases (windows with > 5 SNPs are skipped, localPattern.c:246-250)
"""
import hashlib
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import REF_BIN                       # noqa: E402
OUT = os.path.join(HERE, "index_cases")
KEEP = (".R.seedLen", ".C.pac", ".C.ann", ".C.amb", ".C.bwt", ".C.sa", ".lp", ".R.backward.bwt", ".R.backward.occ", ".R.backward.sa", ".ref")


def main():
    rng = random.Random(11)
    seq = "".join(l.strip() for l in open(os.path.join(HERE, "lambda", "genome.fa")) if not l.startswith(">"))
    A, Bs, C, D = seq[:500], seq[1000:1015], seq[2000:2300], seq[5000:8000]
    other = lambda c: rng.choice([x for x in "ACGT" if x != c])
    snp = lambda name, s, p, n_alt=1: "%s\t%d\t%s\t%s\n" % (name, p + 1, "/".join(sorted([s[p]] + rng.sample([x for x in "ACGT" if x != s[p]], n_alt))), s[p])
    cases = {
        "two_contigs": ([("a", A), ("c", C)], snp("a", A, 9) + snp("c", C, 149)),
        "gap_short": ([("a", A), ("b", Bs), ("c", C)], snp("a", A, 9) + snp("c", C, 149)),
        "all_snp_contigs": ([("a", A), ("b", seq[1000:1100]), ("c", C)], snp("a", A, 9) + snp("a", A, 400, 2) + snp("b", seq[1000:1100], 49, 3) + snp("c", C, 149) + snp("c", C, 296)),
        "dense": ([("d", D)], "".join(snp("d", D, p, rng.choice([1, 1, 1, 2])) for p in range(5, 2990) if rng.random() < 0.14)),
    }
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    for name, (conts, snps) in cases.items():
        d = os.path.join(OUT, name)
        os.makedirs(d)
        with open(os.path.join(d, "genome.fa"), "w") as f:
            for n, s in conts:
                f.write(">%s\n%s\n" % (n, s))
        open(os.path.join(d, "snps.txt"), "w").write(snps)
        subprocess.run([os.path.join(REF_BIN, "salt-idx"), "-k", "19", os.path.join(d, "genome.fa"), os.path.join(d, "snps.txt"), os.path.join(d, "idx")],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sha = hashlib.sha256(open(os.path.join(d, "idx.C.lkt"), "rb").read()).hexdigest()
        open(os.path.join(d, "idx.C.lkt.sha256"), "w").write(sha + "\n")
        for fn in os.listdir(d):
            if fn.startswith("idx.") and not (fn == "idx.C.lkt.sha256" or any(fn == "idx" + k for k in KEEP)):
                os.remove(os.path.join(d, fn))


if __name__ == "__main__":
    main()
