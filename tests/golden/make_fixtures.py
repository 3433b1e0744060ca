#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container only).

Inputs : the lambda-phage FASTA the reference's own test harness holds
         (/root/reference/Test/Genome/lambda_virus.fa -- a data file, copied as data),
         seeded PRNG for everything else.
Tools  : oracle/_ref/salt-idx and oracle/_ref/salt (the real reference compiled in place by
         oracle/Makefile) produce the index files and the expected SAM.
Outputs: tests/golden/lambda/{genome.fa,snps.txt,reads_se.fq,reads_pe_[12].fq}
         tests/golden/lambda/idx.*            (all files `salt` loads, except the 64 MiB .C.lkt
                                               which tests rebuild from .C.pac)
         tests/golden/lambda/expect_*.sam     (@PG line stripped)
         tests/golden/lv_vectors.txt          (LV / mismatch unit vectors from the reference units)

Nothing here is needed at test time except the files it wrote.
"""
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "lambda")
LAMBDA_FA = "/root/reference/Test/Genome/lambda_virus.fa"
K = 19

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def read_fa(path):
    seq = []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                continue
            seq.append(line.strip().upper())
    return "".join(seq)


def make_genome(rng):
    a = read_fa(LAMBDA_FA)
    # contig B: a diverged copy (2 % substitutions), one 60-N hole, one 300-bp inversion
    b = list(a)
    for i in range(len(b)):
        if rng.random() < 0.02:
            b[i] = rng.choice([c for c in "ACGT" if c != b[i]])
    b[20000:20060] = list("N" * 60)
    b[30000:30300] = list(revcomp("".join(b[30000:30300])))
    return [("lambdaA", a), ("lambdaB_div2pct", "".join(b))]


def make_snps(rng, genome, rate=0.04):
    snps = []  # (contig index, pos0, alleles sorted, ref)
    for ci, (_, s) in enumerate(genome):
        for p, c in enumerate(s):
            if c == "N" or rng.random() >= rate:
                continue
            others = [x for x in "ACGT" if x != c]
            n_alt = 2 if rng.random() < 0.03 else 1
            alts = rng.sample(others, n_alt)
            snps.append((ci, p, sorted([c] + alts), c))
    return snps


def write_fa(path, genome):
    with open(path, "w") as f:
        for name, s in genome:
            f.write(">%s\n" % name)
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")


def sim_read(rng, genome, snp_map, L, kind):
    ci = rng.randrange(len(genome))
    name, s = genome[ci]
    # near-end reads now and then (exercise range checks at contig/genome ends)
    r = rng.random()
    if r < 0.02:
        pos = rng.randrange(0, 30)
    elif r < 0.04:
        pos = len(s) - L - rng.randrange(0, 30)
    else:
        pos = rng.randrange(0, len(s) - L - 8)
    frag = list(s[pos:pos + L + 8])
    for i in range(len(frag)):
        al = snp_map.get((ci, pos + i))
        if al is not None:
            frag[i] = rng.choice(al)
        if frag[i] == "N":
            frag[i] = rng.choice("ACGT")
    err = 0.005
    if kind == "noisy":
        err = 0.05
    for i in range(len(frag)):
        if rng.random() < err:
            frag[i] = rng.choice([c for c in "ACGT" if c != frag[i]])
    if kind == "indel":
        p = rng.randrange(10, L - 10)
        n = rng.choice([1, 1, 2, 3])
        if rng.random() < 0.5:
            del frag[p:p + n]
        else:
            frag[p:p] = [rng.choice("ACGT") for _ in range(n)]
    if kind == "indel2":
        for _ in range(2):
            p = rng.randrange(10, L - 10)
            if rng.random() < 0.5:
                del frag[p:p + 1]
            else:
                frag[p:p] = [rng.choice("ACGT")]
    frag = frag[:L]
    if kind == "withN":
        for _ in range(rng.choice([1, 1, 2, 3])):
            frag[rng.randrange(L)] = "N"
    if kind == "manyN":
        for i in rng.sample(range(L), 8):
            frag[i] = "N"
    if kind == "junk":
        frag = [rng.choice("ACGT") for _ in range(L)]
    read = "".join(frag)
    strand = rng.random() < 0.5
    if strand:
        read = revcomp(read)
    return name, pos, strand, read


KINDS = (["plain"] * 80 + ["noisy"] * 5 + ["indel"] * 6 + ["indel2"] * 2 + ["withN"] * 3 +
         ["manyN"] * 1 + ["junk"] * 3)


def write_reads(rng, genome, snp_map, path, n, L):
    with open(path, "w") as f:
        for i in range(n):
            kind = rng.choice(KINDS)
            name, pos, strand, read = sim_read(rng, genome, snp_map, L, kind)
            qual = "".join(chr(33 + rng.randrange(20, 41)) for _ in range(L))
            f.write("@r%d_%s_%d_%s_%s\n%s\n+\n%s\n" % (i, name, pos + 1, "-" if strand else "+", kind, read, qual))


def write_pairs(rng, genome, snp_map, p1, p2, n, L):
    with open(p1, "w") as f1, open(p2, "w") as f2:
        for i in range(n):
            ci = rng.randrange(len(genome))
            name, s = genome[ci]
            isz = max(2 * L + 10, int(rng.gauss(500, 50)))
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
            kind = rng.choice(["plain"] * 85 + ["indel"] * 8 + ["junk1"] * 4 + ["junk"] * 3)
            m1 = frag[:L]
            m2f = frag[isz - L:]
            if kind == "indel":
                p = rng.randrange(20, L - 20)
                tgt = m1 if rng.random() < 0.5 else m2f
                if rng.random() < 0.5:
                    del tgt[p:p + 2]
                    tgt.extend(rng.choice("ACGT") for _ in range(2))
                else:
                    tgt[p:p] = [rng.choice("ACGT"), rng.choice("ACGT")]
                    del tgt[L:]
            if kind in ("junk", "junk1"):
                m2f = [rng.choice("ACGT") for _ in range(L)]
            if kind == "junk":
                m1 = [rng.choice("ACGT") for _ in range(L)]
            r1 = "".join(m1)
            r2 = revcomp("".join(m2f))
            if rng.random() < 0.5:
                r1, r2 = r2, r1
            q = "I" * L
            f1.write("@p%d_%s_%d_%d_%s/1\n%s\n+\n%s\n" % (i, name, pos + 1, isz, kind, r1, q))
            f2.write("@p%d_%s_%d_%d_%s/2\n%s\n+\n%s\n" % (i, name, pos + 1, isz, kind, r2, q))


def run(cmd, stdout=None):
    print("+", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, stdout=stdout, stderr=subprocess.DEVNULL)


def strip_pg(src, dst):
    with open(src) as f, open(dst, "w") as g:
        for line in f:
            if line.startswith("@PG"):
                continue
            g.write(line)


SE_CASES = {
    "se_default": ["-d", "-c"],
    "se_r1_m500": ["-d", "-c", "-r", "1", "-m", "500", "-n", "20"],
    "se_refonly": ["-d", "-c", "-v"],
    "se_r5_s4_m16": ["-d", "-c", "-r", "5", "-s", "4", "-m", "16"],
    "se_plain_t4": ["-t", "4"],
}
PE_CASES = {
    "pe_default": ["-d", "-p", "-c", "-a", "350", "-b", "650"],
    "pe_r5": ["-d", "-p", "-e", "-c", "-a", "350", "-b", "650", "-r", "5"],
}


def main():
    rng = random.Random(20261004)
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    genome = make_genome(rng)
    write_fa(os.path.join(OUT, "genome.fa"), genome)
    snps = make_snps(rng, genome)
    with open(os.path.join(OUT, "snps.txt"), "w") as f:
        for ci, p, al, ref in snps:
            f.write("%s\t%d\t%s\t%s\n" % (genome[ci][0], p + 1, "/".join(al), ref))
    snp_map = {(ci, p): al for ci, p, al, _ in snps}
    write_reads(rng, genome, snp_map, os.path.join(OUT, "reads_se.fq"), 2000, 100)
    write_pairs(rng, genome, snp_map, os.path.join(OUT, "reads_pe_1.fq"),
                os.path.join(OUT, "reads_pe_2.fq"), 1000, 100)

    idx = os.path.join(OUT, "idx")
    run([os.path.join(REF_BIN, "salt-idx"), "-k", str(K), os.path.join(OUT, "genome.fa"),
         os.path.join(OUT, "snps.txt"), idx])
    tmp = os.path.join(OUT, "tmp.sam")
    for name, args in SE_CASES.items():
        with open(tmp, "w") as g:
            run([os.path.join(REF_BIN, "salt")] + args + [idx, os.path.join(OUT, "reads_se.fq")], stdout=g)
        strip_pg(tmp, os.path.join(OUT, "expect_%s.sam" % name))
    for name, args in PE_CASES.items():
        with open(tmp, "w") as g:
            run([os.path.join(REF_BIN, "salt")] + args + [idx, os.path.join(OUT, "reads_pe_1.fq"),
                                                           os.path.join(OUT, "reads_pe_2.fq")], stdout=g)
        strip_pg(tmp, os.path.join(OUT, "expect_%s.sam" % name))
    os.remove(tmp)
    # keep the reference-built .C.lkt out of the repo (64 MiB; rebuilt from .C.pac by tests), and
    # drop files `salt` never reads at align time except .lp (kept: index-builder parity input)
    keep_lkt_sha = subprocess.run(["sha256sum", idx + ".C.lkt"], capture_output=True, text=True).stdout.split()[0]
    with open(os.path.join(OUT, "idx.C.lkt.sha256"), "w") as f:
        f.write(keep_lkt_sha + "\n")
    os.remove(idx + ".C.lkt")
    with open(os.path.join(OUT, "cases.txt"), "w") as f:
        for name, args in list(SE_CASES.items()) + list(PE_CASES.items()):
            f.write("%s\t%s\n" % (name, " ".join(args)))


if __name__ == "__main__":
    main()
