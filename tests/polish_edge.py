"""Hand-made SAM records for `polish` (test data generator, no product or oracle code): hits whose window runs past the end of the genome,
reads with a dozen XA hits on both strands and in both contigs, pairs that are / are not 350..650 apart."""
import os

import numpy as np

LAMBDA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lambda")


def edge_records(n=600, L=100, seed=4):
    seqs = {}
    name = None
    for l in open(os.path.join(LAMBDA, "genome.fa")):
        if l.startswith(">"):
            name = l[1:].strip(); seqs[name] = []
        else:
            seqs[name].append(l.strip().upper())
    contigs = [(k, "".join(v).replace("N", "A")) for k, v in seqs.items()]
    rng = np.random.default_rng(seed)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    def mutate(s, n_sub, indel):
        s = list(s)
        for _ in range(n_sub):
            i = int(rng.integers(0, len(s))); s[i] = "ACGT"[(("ACGT".index(s[i])) + 1 + int(rng.integers(0, 3))) % 4]
        if indel:
            i = int(rng.integers(10, len(s) - 10))
            s = s[:i] + s[i + 2:] + ["A", "C"] if rng.random() < 0.5 else s[:i] + ["G", "T"] + s[i:-2]
        return "".join(s)
    recs = []
    for i in range(n):
        ci = int(rng.integers(0, len(contigs))); cn, cs = contigs[ci]
        # primaries stay clear of the genome end: a winner whose window is clipped there makes the reference abort (its CIGAR routine is then
        # called with k = read length - window length, lv.c:177 asserts) or exit with "push cigar error"
        p = len(cs) - L - int(rng.integers(0, 3)) if (i % 6 == 0 and ci == 0) else int(rng.integers(0, len(cs) - L - 700))
        ref = cs[p:p + L]
        read = mutate(ref, int(rng.integers(0, 6)), rng.random() < 0.3)
        rev = rng.random() < 0.5
        seq = "".join(comp[c] for c in reversed(read)) if rev else read
        xa = []
        for _ in range(int(rng.integers(0, 13))):
            cj = int(rng.integers(0, len(contigs))); q = int(rng.integers(1, len(contigs[cj][1]) - L - 1)); sign = "+-"[int(rng.integers(0, 2))]
            if rng.random() < 0.3:
                q = p + 1 + int(rng.integers(-3, 4)); cj = ci
            if rng.random() < 0.15:
                # a hit hanging over the END OF THE GENOME, always on the reverse strand: it is the last window of the record, so the clipped
                # length and the bytes the reference's buffer keeps behind the clip touch only this (losing) hit's score
                cj = len(contigs) - 1; q = len(contigs[cj][1]) - int(rng.integers(1, 90)); sign = "-"
            xa.append("%s,%s%d,%dM,%d;" % (contigs[cj][0], sign, max(q, 1), L, int(rng.integers(0, 4))))
        tags = ("\tXA:Z:" + "".join(xa) if xa else "") + "\tMD:Z:%d\tNM:i:0" % L
        qual = "".join(chr(33 + int(x)) for x in rng.integers(5, 40, size=L))
        if rev:
            qual = qual[::-1]
        recs.append("q%d\t%d\t%s\t%d\t0\t%dM\t*\t0\t0\t%s\t%s%s" % (i, 16 if rev else 0, cn, p + 1, L, seq, qual, tags))
    return recs
