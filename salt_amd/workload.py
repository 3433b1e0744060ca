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
    # BASELINE.json configs[2] restated: GRCh38-sized genome, SNP density of snp144Common
    "grch38": dict(genome_len=3_100_000_000, n_snps=14_800_000, k=21, n_reads=1_000_000, read_len=100, contigs=24, fast=True),
    # the same generator at a size the CPU tests and quick GPU checks can afford
    "grch38_mini": dict(genome_len=24_000_000, n_snps=115_000, k=21, n_reads=200_000, read_len=100, contigs=24, fast=True),
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


# ---- GRCh38-scale generator: counter-based, the same bits on any device --------------------------------------------------------
# Every random quantity is splitmix64(key(seed, stream) + index) evaluated with torch int64 tensor arithmetic (wrapping multiply,
# logical shifts emulated), so the data are a pure function of (seed, stream, index) whether the tensors live on an MI355X or on
# the host: 3.1e9 bases take a second on the GPU where numpy generators take minutes.  torch is plumbing here (test / bench data),
# never part of the aligner.
_M64 = (1 << 64) - 1


def _s64(v):
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x, k):
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix(x):
    x = x + _s64(0x9E3779B97F4A7C15)
    x = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _s64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def _key(seed, stream):
    z = (seed * 0x100000001B3 + stream * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & _M64
    for _ in range(2):                                   # splitmix64 on python ints
        z = (z + 0x9E3779B97F4A7C15) & _M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        z ^= z >> 31
    return _s64(z)


def hbits(seed, stream, idx):
    """63 random bits (non-negative int64) per element of the int64 index tensor `idx`."""
    return _lsr(_mix(idx + _key(seed, stream)), 1)


def hrange(seed, stream, start, count, device):
    import torch
    return hbits(seed, stream, torch.arange(start, start + count, dtype=torch.int64, device=device))


REPEAT_SLOT = 3000          # one 300-base repeat copy per 3000-base slot: 10 % of the genome, copies never overlap


def make_genome_hash(n, seed=38, device="cpu"):
    """uint8 tensor of n base codes: i.i.d. bases with GC 0.41 (a byte through a 256-entry table: A, T 75/256 each; C, G 53/256 each,
    stream 0), then one copy of a 300-base family (stream 1) per 3000-base slot at a random offset (stream 2), each copy with its own
    divergence in [5 %, 15 %) (stream 3) applied per base (streams 4, 5)."""
    import torch
    lut = torch.zeros(256, dtype=torch.uint8, device=device)
    lut[75:128] = 1; lut[128:181] = 2; lut[181:] = 3
    g = torch.empty(n, dtype=torch.uint8, device=device)
    chunk = 1 << 27
    for lo in range(0, n, chunk):
        c = min(chunk, n - lo)
        g[lo:lo + c] = lut[hrange(seed, 0, lo, c, device) & 255]
    fam = lut[hrange(seed, 1, 0, 300, device) & 255]
    n_slots = n // REPEAT_SLOT
    ar = torch.arange(300, dtype=torch.int64, device=device)
    blk = 1 << 18
    for lo in range(0, n_slots, blk):
        c = min(blk, n_slots - lo)
        slot = torch.arange(lo, lo + c, dtype=torch.int64, device=device)
        start = slot * REPEAT_SLOT + hbits(seed, 2, slot) % (REPEAT_SLOT - 300)
        thr = 50 + hbits(seed, 3, slot) % 100                             # divergence in 1/1000
        cell = slot[:, None] * 300 + ar[None, :]
        mut = (hbits(seed, 4, cell) % 1000) < thr[:, None]
        cp = fam[None, :].expand(c, 300).clone()
        sub = ((cp.to(torch.int64) + 1 + hbits(seed, 5, cell) % 3) & 3).to(torch.uint8)
        cp[mut] = sub[mut]
        g[(start[:, None] + ar[None, :]).reshape(-1)] = cp.reshape(-1)
    return g


def make_snps_hash(genome, n_snps, seed=144):
    """(sorted int64 positions, uint8 allele masks) as tensors on genome.device: distinct uniform positions (an oversampled draw,
    stream 0; the n_snps with the smallest stream-1 hash are kept), 98 % bi-allelic / 2 % tri-allelic, reference allele listed."""
    import torch
    dev, n = genome.device, genome.numel()
    pos = torch.unique(hrange(seed, 0, 0, int(n_snps * 1.03) + 64, dev) % n)
    if pos.numel() < n_snps:
        raise ValueError("SNP density too high for the oversampled draw")
    keep = torch.argsort(hbits(seed, 1, pos))[:n_snps]
    pos = torch.sort(pos[keep]).values
    ref = genome[pos].to(torch.int64)
    alt1 = (ref + 1 + hbits(seed, 2, pos) % 3) & 3
    alt2 = (ref + 1 + hbits(seed, 3, pos) % 3) & 3
    tri = ((hbits(seed, 4, pos) % 100) < 2) & (alt2 != alt1)
    mask = (1 << ref) | (1 << alt1)
    mask = torch.where(tri, mask | (1 << alt2), mask).to(torch.uint8)
    return pos, mask


_PICK = None


def _pick_table(device):
    """[mask][r] -> the (r mod popcount)-th listed allele, r in 0..11 (12 = lcm(1..4): every listed allele equally likely)."""
    global _PICK
    import torch
    if _PICK is None or _PICK.device != torch.device(device):
        t = np.zeros((16, 12), dtype=np.uint8)
        for m in range(1, 16):
            bits = [b for b in range(4) if (m >> b) & 1]
            for r in range(12):
                t[m, r] = bits[r % len(bits)]
        _PICK = torch.from_numpy(t).to(device)
    return _PICK


def make_site_map(n, pos, mask):
    """uint8[n]: the allele mask at SNP positions, 0 elsewhere (what the read generators look alleles up in)."""
    import torch
    site = torch.zeros(n, dtype=torch.uint8, device=pos.device)
    site[pos] = mask
    return site


def make_reads_hash(genome, site, n_reads, L, seed=1, batch=0):
    """One batch of single-end reads as tensors on genome.device: (codes uint8 [n_reads * L], offs int32 [n_reads + 1], start int64,
    reverse bool).  Model of make_reads: uniform start over the concatenated genome, every SNP site under the read takes a random
    listed allele, 0.5 %/base substitutions, 0.02 % of the reads with one 1-2 base indel, 0.1 % with 1-3 N, half reverse-complemented.
    Streams are keyed by (seed, 16 * batch + k), indices by read and base."""
    import torch
    dev, n = genome.device, genome.numel()
    st = lambda k: 16 * batch + k
    rid = torch.arange(n_reads, dtype=torch.int64, device=dev)
    W = L + 4
    start = hbits(seed, st(0), rid) % (n - W)
    ar = torch.arange(W, dtype=torch.int64, device=dev)
    idx = start[:, None] + ar[None, :]
    cell = rid[:, None] * W + ar[None, :]
    frag = genome[idx]
    m = site[idx]
    has = m != 0
    pick = _pick_table(dev)[m.to(torch.int64), hbits(seed, st(1), cell) % 12]
    frag = torch.where(has, pick, frag)
    err = (hbits(seed, st(2), cell) % 1000) < 5
    frag = torch.where(err, ((frag.to(torch.int64) + 1 + hbits(seed, st(3), cell) % 3) & 3).to(torch.uint8), frag)
    # indels: shift the tail of the window left (deletion of k bases at p) or right (insertion of k random bases at p)
    ind = (hbits(seed, st(4), rid) % 10000) < 2
    p = 10 + hbits(seed, st(5), rid) % (L - 20)
    k = 1 + hbits(seed, st(6), rid) % 2
    dele = (hbits(seed, st(7), rid) % 2) == 0
    col = ar[None, :].expand(n_reads, W)
    src_del = torch.where(col >= p[:, None], col + k[:, None], col).clamp(max=W - 1)
    src_ins = torch.where(col >= p[:, None] + k[:, None], col - k[:, None], col)
    src = torch.where((ind & dele)[:, None], src_del, torch.where((ind & ~dele)[:, None], src_ins, col))
    out = torch.gather(frag, 1, src)
    insz = (ind & ~dele)[:, None] & (col >= p[:, None]) & (col < p[:, None] + k[:, None])
    out = torch.where(insz, (hbits(seed, st(8), cell) % 4).to(torch.uint8), out)
    reads = out[:, :L].contiguous()
    # N: 0.1 % of the reads get 1-3 of them
    hasn = (hbits(seed, st(9), rid) % 1000) < 1
    nn = 1 + hbits(seed, st(10), rid) % 3
    colL = col[:, :L]
    for j in range(3):
        q = hbits(seed, st(11 + j), rid) % L
        reads = torch.where((hasn & (nn > j))[:, None] & (colL == q[:, None]), torch.full_like(reads, 4), reads)
    rev = (hbits(seed, st(14), rid) % 2) == 1
    rc = torch.flip(reads, dims=[1])
    rc = torch.where(rc < 4, 3 - rc, rc)
    reads = torch.where(rev[:, None], rc, reads)
    offs = (torch.arange(n_reads + 1, dtype=torch.int64, device=dev) * L).to(torch.int32)
    return reads.reshape(-1), offs, start, rev


def make_pairs_hash(genome, site, n_pairs, L, seed=2, batch=0, insert_mean=400, insert_sd=50, damaged=0.0, orphan=0.0):
    """One batch of read pairs as tensors on genome.device, mates interleaved (2p, 2p + 1) the way salt_gpu_align_pe takes them:
    (codes uint8 [2 * n_pairs * L], offs int32 [2 * n_pairs + 1], start1 int64, start2 int64, flipped bool).  A fragment of
    insert_mean + insert_sd * z bases (z: sum of four uniforms, variance 1) starts uniformly over the concatenated genome; one mate is
    its first L bases as they lie, the other the reverse complement of its last L; `flipped` pairs swap the roles (the fragment came
    from the reverse strand).  Every SNP site takes a random listed allele, 0.5 %/base substitutions; no N.  `damaged` of the
    fragment-end mates get 9 % substitutions and a 2-base deletion (seed-and-verify misses them and the Smith-Waterman rescue runs, as in
    make_pairs), `orphan` of them are random bases; both 0 by default: no indels.
    start1 / start2: the leftmost genome position of mate 2p / 2p + 1."""
    import torch
    dev, n = genome.device, genome.numel()
    st = lambda k: 16 * batch + k
    pid = torch.arange(n_pairs, dtype=torch.int64, device=dev)
    z = sum((hbits(seed, st(1 + j), pid) % 100000).to(torch.float64) / 100000.0 for j in range(4))      # mean 2, variance 1/3
    isz = (insert_mean + insert_sd * (z - 2.0) * (3.0 ** 0.5)).round().to(torch.int64).clamp(min=L + 10, max=insert_mean + 6 * insert_sd)
    f0 = hbits(seed, st(0), pid) % (n - (insert_mean + 6 * insert_sd) - 8)
    ar = torch.arange(L, dtype=torch.int64, device=dev)

    def window(start, stream):
        idx = start[:, None] + ar[None, :]
        cell = pid[:, None] * L + ar[None, :]
        frag = genome[idx]
        m = site[idx]
        pick = _pick_table(dev)[m.to(torch.int64), hbits(seed, st(stream), cell) % 12]
        frag = torch.where(m != 0, pick, frag)
        err = (hbits(seed, st(stream + 1), cell) % 1000) < 5
        return torch.where(err, ((frag.to(torch.int64) + 1 + hbits(seed, st(stream + 2), cell) % 3) & 3).to(torch.uint8), frag)

    left, right = window(f0, 5), window(f0 + isz - L, 8)
    if damaged > 0 or orphan > 0:
        u = hbits(seed, st(12), pid) % 100000
        is_dmg, is_orph = u < int(damaged * 100000), (u >= int(damaged * 100000)) & (u < int((damaged + orphan) * 100000))
        # damaged: the L + 2 bases that end where the fragment ends, two of them (at p, p + 1) deleted, 9 % substitutions on top
        ar2 = torch.arange(L + 2, dtype=torch.int64, device=dev)
        idx = (f0 + isz - L - 2)[:, None] + ar2[None, :]
        wide = genome[idx]
        m = site[idx]
        cell = pid[:, None] * (L + 2) + ar2[None, :]
        wide = torch.where(m != 0, _pick_table(dev)[m.to(torch.int64), hbits(seed, st(13), cell) % 12], wide)
        err = (hbits(seed, st(14), cell) % 100) < 9
        wide = torch.where(err, ((wide.to(torch.int64) + 1 + hbits(seed, st(15), cell) % 3) & 3).to(torch.uint8), wide)
        p = 20 + hbits(seed, st(12), pid + n_pairs) % (L - 40)
        take = ar[None, :] + 2 * (ar[None, :] >= p[:, None]).to(torch.int64)
        dmg = torch.gather(wide, 1, take)
        rnd = (hbits(seed, st(13), pid[:, None] * L + ar[None, :] + 7 * n_pairs * (L + 2)) % 4).to(torch.uint8)
        right = torch.where(is_dmg[:, None], dmg, torch.where(is_orph[:, None], rnd, right))
    rc = lambda x: 3 - torch.flip(x, dims=[1])
    flipped = (hbits(seed, st(11), pid) % 2) == 1
    m1 = torch.where(flipped[:, None], rc(right), left)
    m2 = torch.where(flipped[:, None], left, rc(right))
    reads = torch.stack([m1, m2], dim=1).reshape(-1)
    offs = (torch.arange(2 * n_pairs + 1, dtype=torch.int64, device=dev) * L).to(torch.int32)
    s_left, s_right = f0, f0 + isz - L
    return reads, offs, torch.where(flipped, s_right, s_left), torch.where(flipped, s_left, s_right), flipped


def as_builder_input(genome, pos, mask, contigs=1, name="synth1"):
    """(contigs, SNP groups) as salt_amd.idx_build_mem takes them: base letters per contig, and per contig its SNPs with 0-based
    positions inside the contig, allele masks and reference codes (what write_fasta + write_snps would put into files)."""
    b = contig_bounds(len(genome), contigs)
    cs, gs = [], []
    edges = np.searchsorted(pos, b)
    for ci in range(contigs):
        nm = name if contigs == 1 else "synth%d" % (ci + 1)
        cs.append((nm, ACGT[genome[b[ci]:b[ci + 1]]]))
        p = pos[edges[ci]:edges[ci + 1]]
        if len(p):
            gs.append((nm, (p - b[ci]).astype(np.uint32), mask[edges[ci]:edges[ci + 1]], genome[p]))
    return cs, gs


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


def generate_device(config, device):
    """(genome uint8 codes, sorted int64 SNP positions, uint8 allele masks) of a config as torch tensors on `device`.  The
    `fast` configs are generated there (counter-based hash generator: the same data on any device); the numpy-generated ones
    (chr21, mini, tiny, grch38_tenth) are generated on the host and copied."""
    import torch
    c = CONFIGS[config]
    if c.get("fast"):
        g = make_genome_hash(c["genome_len"], 38, device)
        p, m = make_snps_hash(g, c["n_snps"], 144)
        return g, p, m
    genome = make_genome(c["genome_len"])
    pos, mask = make_snps(genome, c["n_snps"])
    return torch.from_numpy(genome).to(device), torch.from_numpy(pos).to(device), torch.from_numpy(mask).to(device)


def generate(config):
    """The same as numpy arrays (genome codes, SNP positions, SNP allele masks) -- a pure function of the config's seeds."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    g, p, m = generate_device(config, dev)
    return g.cpu().numpy(), p.cpu().numpy(), m.cpu().numpy()


def fastq_bytes(seqs, n, L, first_id=0):
    """FASTQ text of n fixed-length reads (uint8 codes, n * L of them): @r<10-digit id> / bases / + / 'I' qualities, built without a
    per-read Python loop."""
    chars = np.frombuffer(b"ACGTN", dtype=np.uint8)
    w = 1 + 1 + 10 + 1 + L + 1 + 2 + L + 1                     # "@r" + id + nl + seq + nl + "+\n" + qual + nl
    rec = np.empty((n, w), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    ids = np.arange(first_id, first_id + n, dtype=np.int64)
    for d in range(10):
        rec[:, 2 + 9 - d] = 48 + (ids // 10 ** d) % 10
    rec[:, 12] = 10
    rec[:, 13:13 + L] = chars[np.asarray(seqs[:n * L]).reshape(n, L)]
    rec[:, 13 + L] = 10; rec[:, 14 + L] = ord("+"); rec[:, 15 + L] = 10
    rec[:, 16 + L:16 + 2 * L] = ord("I")
    rec[:, 16 + 2 * L] = 10
    return rec.tobytes()


def prepare(config, cache_dir, gpu_device="auto", log=None, arrays=None):
    """Generates the genome + SNP set of `config` and indexes it with the product's own salt-idx (from memory, without FASTA / SNP
    text files; `.lp` is not written) unless the cache holds the index already.  gpu_device: the device the suffix sorter runs on
    (None = host SA-IS, minutes for hundreds of Mbp and hopeless for GRCh38).  arrays: (genome, positions, masks) when the caller has
    generated them already.  Returns paths, the generated arrays and stage times."""
    import time
    from . import api
    if gpu_device == "auto":                                # test / bench tooling: the device sorter wherever a GPU is visible
        import torch
        gpu_device = torch.cuda.current_device() if torch.cuda.is_available() else None
    c = dict(CONFIGS[config])
    nc = c.setdefault("contigs", 1)
    d = os.path.join(cache_dir, "salt_%s_g%d_s%d_k%d" % (config, c["genome_len"], c["n_snps"], c["k"]))
    os.makedirs(d, exist_ok=True)
    prefix = os.path.join(d, "idx")
    done = os.path.join(d, "DONE")
    times = {}
    t0 = time.time()
    if arrays is not None:                                   # already generated (torch tensors on a device, or numpy arrays)
        genome, pos, mask = [a.cpu().numpy() if hasattr(a, "cpu") else a for a in arrays]
        times["to_host"] = round(time.time() - t0, 2)
    else:
        genome, pos, mask = generate(config)
        times["generate_host"] = round(time.time() - t0, 2)
    if not os.path.exists(done):
        t0 = time.time()
        contigs, groups = as_builder_input(genome, pos, mask, nc)
        times["letters"] = round(time.time() - t0, 2)
        t0 = time.time()
        api.idx_build_mem(contigs, groups, prefix, c["k"], gpu_device=gpu_device, flags=api.IDX_NO_LP)
        del contigs, groups
        times["index_build"] = round(time.time() - t0, 2)
        if log:
            log("index built in %.1f s (%s suffix sorter)" % (times["index_build"], "device" if gpu_device is not None else "host"))
        open(done, "w").write("ok\n")
    return dict(dir=d, prefix=prefix, genome=genome, snp_pos=pos, snp_mask=mask, times=times, **c)
