"""Seeded synthetic workloads for the benchmark configs: real GRCh38 / snp144 /
chr21 cannot be fetched (no network), so each config is restated as a generated genome + SNP set +
read set.  PRNG: numpy PCG64 with the seeds given below; everything is a pure function of them.

  chr21-scale (BASELINE.json configs[1]): 1 contig of 40 000 000 bp, i.i.d. bases with GC 0.41,
      10 % of the length overwritten by copies of a 300-bp repeat family at 5-15 % divergence
      (seed 21); 190 000 SNP sites uniform, 98 % bi-allelic / 2 % tri-allelic, reference allele
      always listed (seed 144); k = 21; 1 000 000 reads of 100 bp, uniform start, 50 % reverse
      complement, every SNP site takes a random listed allele, 0.5 %/base substitutions, 0.02 % of
      reads with one 1-2 bp indel, 0.1 % of reads with >= 1 N (seed 1).
"""
import os

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

CONFIGS = {
    # name: genome_len, n_snps, k, n_reads, read_len
    "chr21": dict(genome_len=40_000_000, n_snps=190_000, k=21, n_reads=1_000_000, read_len=100),
    # a tenth of GRCh38 (configs[2] scaled to what one gpurun call can index): 8 contigs, SNP density of snp144Common
    "grch38_tenth": dict(genome_len=320_000_000, n_snps=1_520_000, k=21, n_reads=1_000_000, read_len=100, contigs=8),
    "mini": dict(genome_len=2_000_000, n_snps=9_500, k=21, n_reads=50_000, read_len=100),
    "tiny": dict(genome_len=200_000, n_snps=1_000, k=19, n_reads=4_000, read_len=100),
}


def make_genome(n, seed=21):
    rng = np.random.Generator(np.random.PCG64(seed))
    p = np.array([0.295, 0.205, 0.205, 0.295])          # A C G T, GC = 0.41
    g = rng.choice(4, size=n, p=p).astype(np.uint8)
    fam = rng.choice(4, size=300, p=p).astype(np.uint8)
    n_copies = n // 10 // 300
    starts = rng.integers(0, n - 300, size=n_copies)
    for s in starts:
        div = rng.uniform(0.05, 0.15)
        c = fam.copy()
        m = rng.random(300) < div
        c[m] = (c[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        g[s:s + 300] = c
    return g


def make_tandem(unit_len=30, copies=40000, divergence=0.01, flank=30000, seed=77):
    """Adversarial genome for the locate caps and the interval-size sort: `copies` diverged copies of one `unit_len`-base unit
    between two random flanks -- every seed of a read from the block has hundreds to thousands of suffix-array rows."""
    rng = np.random.Generator(np.random.PCG64(seed))
    unit = rng.integers(0, 4, size=unit_len).astype(np.uint8)
    block = np.tile(unit, copies)
    m = rng.random(len(block)) < divergence
    block[m] = (block[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
    return np.concatenate([rng.integers(0, 4, size=flank).astype(np.uint8), block, rng.integers(0, 4, size=flank).astype(np.uint8)])


def make_snps(genome, n_snps, seed=144):
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = np.sort(rng.choice(len(genome), size=n_snps, replace=False))
    ref = genome[pos]
    alt1 = (ref + rng.integers(1, 4, size=n_snps)) & 3
    tri = rng.random(n_snps) < 0.02
    alt2 = (ref + rng.integers(1, 4, size=n_snps)) & 3
    tri &= alt2 != alt1
    mask = (1 << ref) | (1 << alt1)
    mask = np.where(tri, mask | (1 << alt2), mask).astype(np.uint8)
    return pos.astype(np.int64), mask


def contig_bounds(n, contigs):
    """Start offsets (contigs + 1 of them) of `contigs` near-equal contigs named synth1 .. synthN over n bases."""
    return [n * i // contigs for i in range(contigs + 1)]


def write_fasta(path, name, genome, contigs=1):
    b = contig_bounds(len(genome), contigs)
    with open(path, "wb") as f:
        for ci in range(contigs):
            s = ACGT[genome[b[ci]:b[ci + 1]]]
            f.write(b">" + (name if contigs == 1 else "synth%d" % (ci + 1)).encode() + b"\n")
            w = 80
            full = len(s) // w * w
            body = np.concatenate([s[:full].reshape(-1, w), np.full((full // w, 1), 10, dtype=np.uint8)], axis=1)
            f.write(body.tobytes())
            if full < len(s):
                f.write(s[full:].tobytes() + b"\n")


def write_snps(path, name, genome, pos, mask, contigs=1):
    lines = []
    b = contig_bounds(len(genome), contigs)
    ci = 0
    for p, m in zip(pos.tolist(), mask.tolist()):
        while p >= b[ci + 1]:
            ci += 1
        nm = (name if contigs == 1 else "synth%d" % (ci + 1)).encode()
        al = b"/".join(bytes([ACGT[x]]) for x in range(4) if (m >> x) & 1)
        lines.append(b"%s\t%d\t%s\t%s\n" % (nm, p - b[ci] + 1, al, bytes([ACGT[genome[p]]])))
    with open(path, "wb") as f:
        f.write(b"".join(lines))


def make_reads(genome, pos, mask, n_reads, L, seed=1):
    """Returns (seqs uint8 codes [n_reads*L], offs uint32, truth start, truth strand)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(genome)
    start = rng.integers(0, n - L - 4, size=n_reads)
    idx = start[:, None] + np.arange(L + 4)[None, :]
    frag = genome[idx]                                   # (n_reads, L+4)
    # random listed allele at every SNP site covered
    site = np.zeros(n, dtype=np.uint8)
    site[pos] = mask
    m = site[idx]
    has = m != 0
    r = rng.integers(0, 4, size=int(has.sum()))
    mm = m[has]
    # pick the (r mod popcount)-th set bit
    pc = np.array([bin(x).count("1") for x in range(16)], dtype=np.uint8)[mm]
    want = (r % pc).astype(np.uint8)
    pick = np.zeros(len(mm), dtype=np.uint8)
    seen = np.zeros(len(mm), dtype=np.uint8)
    for b in range(4):
        isset = ((mm >> b) & 1).astype(bool)
        hit = isset & (seen == want)
        pick[hit] = b
        seen[isset] += 1
    frag[has] = pick
    # substitutions
    e = rng.random(frag.shape) < 0.005
    frag[e] = (frag[e] + rng.integers(1, 4, size=int(e.sum()))) & 3
    reads = frag[:, :L].copy()
    # indels in 0.02 % of the reads
    for i in np.nonzero(rng.random(n_reads) < 0.0002)[0]:
        p = int(rng.integers(10, L - 10))
        k = int(rng.integers(1, 3))
        row = frag[i]
        if rng.random() < 0.5:
            new = np.concatenate([row[:p], row[p + k:]])[:L]
        else:
            new = np.concatenate([row[:p], rng.integers(0, 4, size=k).astype(np.uint8), row[p:]])[:L]
        reads[i] = new
    # N in 0.1 % of the reads
    for i in np.nonzero(rng.random(n_reads) < 0.001)[0]:
        for _ in range(int(rng.integers(1, 4))):
            reads[i, int(rng.integers(0, L))] = 4
    strand = rng.random(n_reads) < 0.5
    rc = reads[strand][:, ::-1]
    rc = np.where(rc < 4, 3 - rc, rc).astype(np.uint8)
    reads[strand] = rc
    offs = (np.arange(n_reads + 1, dtype=np.uint64) * L).astype(np.uint32)
    return reads.reshape(-1), offs, start, strand


def make_pairs(genome, pos, mask, n_pairs, L, seed=2, insert_mean=450, insert_sd=35, damaged=0.08, orphan=0.02):
    """Paired-end reads, mates interleaved (pair i = reads 2i, 2i+1): FR fragments of insert_mean +- insert_sd,
    listed SNP alleles, 0.5 % substitutions; `damaged` of the second mates get 9 % substitutions and a 2-base
    deletion (so seed-and-verify misses them and the Smith-Waterman rescue runs), `orphan` of them are random."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(genome)
    isz = np.clip(np.rint(rng.normal(insert_mean, insert_sd, size=n_pairs)).astype(np.int64), 2 * L + 10, insert_mean + 6 * insert_sd)
    w = int(isz.max()) + 4
    start = rng.integers(0, n - w, size=n_pairs)
    idx = start[:, None] + np.arange(w)[None, :]
    frag = genome[idx]
    site = np.zeros(n, dtype=np.uint8)
    site[pos] = mask
    m = site[idx]
    has = m != 0
    mm = m[has]
    alt = rng.random(len(mm)) < 0.5                      # a listed allele (lowest set bit other than the base, if any)
    cur = frag[has]
    other = mm & ~(1 << cur).astype(np.uint8)
    low = np.zeros(len(mm), dtype=np.uint8)
    for b in (3, 2, 1, 0):
        low[((other >> b) & 1).astype(bool)] = b
    use = alt & (other != 0)
    cur[use] = low[use]
    frag[has] = cur
    e = rng.random(frag.shape) < 0.005
    frag[e] = (frag[e] + rng.integers(1, 4, size=int(e.sum()))) & 3
    m1 = frag[:, :L].copy()
    j = (isz - L)[:, None] + np.arange(L + 4)[None, :]
    tail = np.take_along_axis(frag, j, axis=1)           # (n_pairs, L+4) ending past the fragment end by <= 4
    m2f = tail[:, :L].copy()
    dmg = np.nonzero(rng.random(n_pairs) < damaged)[0]
    for i in dmg:
        row = tail[i].copy()
        ee = rng.random(L + 4) < 0.09
        row[ee] = (row[ee] + rng.integers(1, 4, size=int(ee.sum()))) & 3
        p = int(rng.integers(20, L - 20))
        m2f[i] = np.concatenate([row[:p], row[p + 2:]])[:L]
    orp = np.nonzero(rng.random(n_pairs) < orphan)[0]
    m2f[orp] = rng.integers(0, 4, size=(len(orp), L)).astype(np.uint8)
    m2 = (3 - m2f[:, ::-1]).astype(np.uint8)             # second mate reads the reverse strand
    flip = rng.random(n_pairs) < 0.5                     # which file holds the forward mate
    a = np.where(flip[:, None], m2, m1)
    b = np.where(flip[:, None], m1, m2)
    reads = np.stack([a, b], axis=1).reshape(-1)
    offs = (np.arange(2 * n_pairs + 1, dtype=np.uint64) * L).astype(np.uint32)
    return reads.astype(np.uint8), offs, start, isz


def write_fastq(path, seqs, offs, n=None):
    chars = np.frombuffer(b"ACGTN", dtype=np.uint8)
    n = len(offs) - 1 if n is None else n
    with open(path, "wb") as f:
        for i in range(n):
            s = chars[seqs[offs[i]:offs[i + 1]]].tobytes()
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)))


def prepare(config, cache_dir):
    """Generates (or finds cached) genome FASTA + SNP file + index for `config`; returns paths."""
    from . import api
    import ctypes
    c = dict(CONFIGS[config])
    nc = c.setdefault("contigs", 1)
    d = os.path.join(cache_dir, "salt_%s_g%d_s%d_k%d" % (config, c["genome_len"], c["n_snps"], c["k"]))
    os.makedirs(d, exist_ok=True)
    fa, snp, prefix = os.path.join(d, "genome.fa"), os.path.join(d, "snps.txt"), os.path.join(d, "idx")
    done = os.path.join(d, "DONE")
    genome = make_genome(c["genome_len"])
    pos, mask = make_snps(genome, c["n_snps"])
    if not os.path.exists(done):
        write_fasta(fa, "synth1", genome, nc)
        write_snps(snp, "synth1", genome, pos, mask, nc)
        lib = api.host_lib()
        lib.salt_idx_build.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        lib.salt_idx_last_error.restype = ctypes.c_char_p
        if lib.salt_idx_build(fa.encode(), snp.encode(), prefix.encode(), c["k"]) != 0:
            raise api.SaltError("index build failed: %s" % lib.salt_idx_last_error().decode())
        open(done, "w").write("ok\n")
    return dict(dir=d, fasta=fa, snps=snp, prefix=prefix, genome=genome, snp_pos=pos, snp_mask=mask, **c)
