"""GPU parity for the paired-end path (alnpe_core1: per-mate alnse_overlap, pairing2 / pairing_singleton, SSW mate
rescue, alnpe_sam), through the C ABI: against the reference's own SAM output and its ssw.c known answers."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN, LAMBDA, read_cases

pytestmark = pytest.mark.gpu

PE_CASES = [c for c in read_cases() if c.startswith("pe_")]


@pytest.fixture(scope="module")
def lam_pe():
    import salt_amd
    idx = salt_amd.Index.reload(os.path.join(LAMBDA, "idx"))
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=4096)
    reads = salt_amd.interleave_pairs(salt_amd.read_fastq(os.path.join(LAMBDA, "reads_pe_1.fq")),
                                      salt_amd.read_fastq(os.path.join(LAMBDA, "reads_pe_2.fq")))
    yield salt_amd, idx, aln, reads
    aln.close()
    idx.destroy()


def _diff_report(got, want):
    g, w = got.split(b"\n"), want.split(b"\n")
    bad = [i for i in range(min(len(g), len(w))) if g[i] != w[i]]
    msg = "\n".join("line %d\n  got  %r\n  want %r" % (i, g[i][:500], w[i][:500]) for i in bad[:6])
    return "%d differing lines (of %d / %d)\n%s" % (len(bad), len(g), len(w), msg)


@pytest.mark.parametrize("case", PE_CASES)
def test_gpu_pe_sam_matches_reference_golden(case, lam_pe):
    salt_amd, idx, aln, (names, seqs, offs, quals) = lam_pe
    opt, _ = salt_amd.AlnOpt.from_argv(read_cases()[case], idx.l_seed)
    assert opt.paired
    res = aln.alnpe_core1(opt, idx, seqs, offs)
    got = salt_amd.sam_text_pe(idx, opt, names, seqs, offs, quals, res)
    want = open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
    if got != want:
        pytest.fail(_diff_report(got, want))


def test_gpu_pe_small_batches_equal_one_batch(lam_pe):
    """Pairs are independent: a workspace that holds 64 mates at a time gives the same rows as one big batch."""
    salt_amd, idx, aln, (names, seqs, offs, quals) = lam_pe
    opt, _ = salt_amd.AlnOpt.from_argv(read_cases()["pe_default"], idx.l_seed)
    n = 600
    whole = aln.alnpe_core1(opt, idx, seqs[:offs[n]], offs[:n + 1])
    small = salt_amd.GpuAligner(idx, device=0, max_reads=64)
    try:
        parts = small.alnpe_core1(opt, idx, seqs[:offs[n]], offs[:n + 1])
    finally:
        small.close()
    for f in ("pos", "strand", "n_diff", "is_gap", "mapq", "b0", "b1", "seq_start", "seq_end", "n_hits", "n_cigar"):
        assert np.array_equal(whole[f], parts[f]), f
    a = salt_amd.sam_text_pe(idx, opt, names[:n], seqs, offs[:n + 1], quals[:n], whole)     # CIGARs, XA lists
    b = salt_amd.sam_text_pe(idx, opt, names[:n], seqs, offs[:n + 1], quals[:n], parts)
    assert a == b


def test_gpu_ssw_unit_matches_reference_vectors():
    """k_sw (striped word kernel emulated on 8 lanes, second best, reverse pass, banded traceback) against the
    answers the reference's ssw.c printed for 600 windows, both score matrices (tests/golden/ssw_vectors.txt)."""
    import salt_amd
    lib = salt_amd.gpu_lib()
    lib.salt_gpu_diag_ssw.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p] * 8
    aware, refs, reads, want, roff, qoff = [], [], [], [], [0], [0]
    with open(os.path.join(GOLDEN, "ssw_vectors.txt")) as f:
        for line in f:
            t = line.split()
            aware.append(int(t[1]))
            refs.append(np.array([int(c, 16) for c in t[2]], dtype=np.uint8))
            reads.append(np.frombuffer(t[3].encode(), dtype=np.uint8) - 48)
            want.append(([int(x) for x in t[4:10]], t[10]))
            roff.append(roff[-1] + len(refs[-1])); qoff.append(qoff[-1] + len(reads[-1]))
    n = len(aware)
    assert n == 600
    aw = np.array(aware, dtype=np.uint8)
    rs, qs = np.concatenate(refs), np.concatenate(reads).astype(np.uint8)
    ro, qo = np.array(roff, dtype=np.uint32), np.array(qoff, dtype=np.uint32)
    out6 = np.zeros((n, 6), dtype=np.int32)
    cig = np.zeros((n, 64), dtype=np.uint16)
    ncig = np.zeros(n, dtype=np.uint16)
    rc = lib.salt_gpu_diag_ssw(n, aw.ctypes.data, rs.ctypes.data, ro.ctypes.data, qs.ctypes.data, qo.ctypes.data,
                               out6.ctypes.data, cig.ctypes.data, ncig.ctypes.data)
    assert rc == 0, lib.salt_gpu_last_error()
    for i in range(n):
        w6, wc = want[i]
        assert [int(x) for x in out6[i]] == w6, (i, out6[i], w6)
        got = "".join("%d%s" % (int(x) >> 4, "MID"[int(x) & 3]) for x in cig[i, :int(ncig[i])]) or "-"
        assert got == wc, (i, got, wc)


def test_gpu_pe_needs_even_mates_and_pac(lam_pe):
    salt_amd, idx, aln, (names, seqs, offs, quals) = lam_pe
    opt, _ = salt_amd.AlnOpt.from_argv(["-p"], idx.l_seed)
    with pytest.raises(salt_amd.SaltError):
        aln.alnpe_core1(opt, idx, seqs[:offs[3]], offs[:4])
    fresh = salt_amd.GpuAligner(idx, device=0, max_reads=64)
    try:
        lib = salt_amd.gpu_lib()
        co, pe = opt._c(), opt._pe()
        res = np.zeros(2, dtype=salt_amd.RESULT_DTYPE)
        o = offs[:3].astype(np.uint32)
        s = np.ascontiguousarray(seqs[:offs[2]])
        rc = lib.salt_gpu_align_pe(fresh._ws, ctypes.byref(co), ctypes.byref(pe), 1, s.ctypes.data, o.ctypes.data, res.ctypes.data)
        assert rc != 0 and b"pac" in lib.salt_gpu_last_error()
    finally:
        fresh.close()


@pytest.fixture(scope="module")
def tiny_pe(tmp_path_factory):
    """Seeded repeat-bearing genome (salt_amd/workload.py) with FR pairs: 8 % damaged second mates, 2 % orphans."""
    from salt_amd import workload
    w = workload.prepare("tiny", str(tmp_path_factory.mktemp("wlpe")))
    seqs, offs, _, _ = workload.make_pairs(w["genome"], w["snp_pos"], w["snp_mask"], 3000, w["read_len"], seed=11)
    return w, seqs, offs


@pytest.mark.parametrize("optargs", [["-p"], ["-p", "-a", "380", "-b", "520", "-r", "5"], ["-p", "-a", "100", "-b", "460", "-m", "40", "-v"]])
def test_gpu_pe_fields_match_oracle_on_synthetic_pairs(tiny_pe, optargs):
    """Every result field of both mates after pairing / rescue (pos, strand, n_diff, is_gap, mapq, SW scores, soft clips,
    alt hits, CIGAR text) against the CPU oracle on 3000 synthetic pairs; the window options move pairs between the
    'proper pair', 'pick among alternative hits', 'SNP-aware rescue' and 'singleton rescue' branches."""
    import sys
    import salt_amd
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    w, seqs, offs = tiny_pe
    idx = salt_amd.Index.reload(w["prefix"])
    opt, _ = salt_amd.AlnOpt.from_argv(optargs, idx.l_seed)
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=len(offs) - 1)
    res = aln.alnpe_core1(opt, idx, seqs, offs)
    aln.close()
    ora = oracle_py.Oracle(w["prefix"])
    oo = ora.opt(l_overlap=opt.l_overlap, max_seed=opt.max_seed, max_locate=opt.max_locate, seed_only_ref=opt.seed_only_ref)
    want = ora.align_pe(oo, seqs, offs, opt.min_tlen, opt.max_tlen, n_threads=8)
    ora.close()
    idx.destroy()
    bad = oracle_py.compare(res, want, pe=True)
    detail = [(int(i), [(f, res[f][i].tolist(), want[f][i].tolist()) for f in ("pos", "strand", "n_diff", "is_gap", "mapq", "b0", "b1", "seq_start", "seq_end")]) for i in bad[:4]]
    assert len(bad) == 0, (len(bad), detail)
    rescued = int(((want["seq_start"] != 0) | (want["seq_end"] != w["read_len"] - 1)).sum())
    assert (want["pos"] != 0xFFFFFFFF).mean() > 0.9 and rescued > 20, rescued


def test_cli_salt_pe_matches_reference_golden(tmp_path):
    """`salt -p` (C++ CLI) on an index written by salt-idx: the reference's paired-end SAM stream byte for byte."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    salt, salt_idx = os.path.join(root, "salt_amd", "bin", "salt"), os.path.join(root, "salt_amd", "bin", "salt-idx")
    if not (os.path.exists(salt) and os.path.exists(salt_idx)):
        subprocess.run(["make", "-C", os.path.join(root, "salt_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    prefix = str(tmp_path / "idx")
    subprocess.run([salt_idx, "-k", "19", os.path.join(LAMBDA, "genome.fa"), os.path.join(LAMBDA, "snps.txt"), prefix],
                   check=True, stderr=subprocess.DEVNULL)
    for case, extra in (("pe_default", []), ("pe_r5", ["-t", "4"])):
        out = subprocess.run([salt] + read_cases()[case] + extra + [prefix, os.path.join(LAMBDA, "reads_pe_1.fq"), os.path.join(LAMBDA, "reads_pe_2.fq")],
                             check=True, capture_output=True).stdout
        got = b"".join(l for l in out.splitlines(keepends=True) if not l.startswith(b"@PG"))
        want = open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
        assert got == want, (case, _diff_report(got, want))


def test_gpu_pe_locate_cap_is_the_references(tmp_path):
    """alnse_locate (PE) stops at MAX_LOC_POS = 0x40000 loci per strand (alnse.c:42,533), far above the SE cap: on a tandem
    repeat (40 000 diverged copies of a 30-bp unit) with -r 2 -s 100000 (no seed extension) a mate enumerates up to 65 seeds x
    (max_locate + 1) rows = 65 000 distinct loci per strand.  Fields of both mates against the oracle."""
    import sys
    import salt_amd
    from salt_amd import workload
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    genome = workload.make_tandem()
    pos, mask = workload.make_snps(genome, 600, seed=5)
    fa, snp, prefix = str(tmp_path / "g.fa"), str(tmp_path / "s.txt"), str(tmp_path / "idx")
    workload.write_fasta(fa, "tandem", genome)
    workload.write_snps(snp, "tandem", genome, pos, mask)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import subprocess
    subprocess.run([os.path.join(root, "salt_amd", "bin", "salt-idx"), "-k", "21", fa, snp, prefix], check=True, stderr=subprocess.DEVNULL)
    seqs, offs, _, _ = workload.make_pairs(genome, pos, mask, 100, 150, seed=9, insert_mean=400, insert_sd=40)
    L = 150
    idx = salt_amd.Index.reload(prefix)
    opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-r", "2", "-s", "100000"], idx.l_seed)
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=600, max_bases=600 * L)
    res = aln.alnpe_core1(opt, idx, seqs, offs)
    aln.close()
    ora = oracle_py.Oracle(prefix)
    oo = ora.opt(l_overlap=opt.l_overlap, max_seed=opt.max_seed, max_locate=opt.max_locate, seed_only_ref=opt.seed_only_ref)
    want = ora.align_pe(oo, seqs, offs, opt.min_tlen, opt.max_tlen, n_threads=16)
    ora.close()
    idx.destroy()
    bad = oracle_py.compare(res, want, pe=True)
    detail = [(int(i), [(f, res[f][i].tolist(), want[f][i].tolist()) for f in ("pos", "strand", "n_diff", "is_gap", "mapq", "b0", "b1")]) for i in bad[:4]]
    assert len(bad) == 0, (len(bad), detail)
    assert (want["pos"] != 0xFFFFFFFF).mean() > 0.9


@pytest.mark.parametrize("L,window", [(100, ("250", "550")), (200, ("400", "1100")), (300, ("600", "1500")), (500, ("1000", "2400"))])
def test_gpu_pe_long_mates_match_oracle(tiny_pe, L, window):
    """k_sw has one variant per stripe count (13 / 19 / 32 stripes of 8 in registers, rows in LDS beyond 256 bases) and the seed /
    verify / LV kernels have their own length classes: mates of 100, 200, 300 and 500 bases against the oracle, every field."""
    import sys
    import salt_amd
    from salt_amd import workload
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    w, _, _ = tiny_pe
    seqs, offs, _, _ = workload.make_pairs(w["genome"], w["snp_pos"], w["snp_mask"], 1200, L, seed=21 + L,
                                           insert_mean=int(window[0]) + 3 * L // 2, insert_sd=30)
    idx = salt_amd.Index.reload(w["prefix"])
    opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", window[0], "-b", window[1]], idx.l_seed)
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=len(offs) - 1, max_bases=int(offs[-1]) + 64)
    res = aln.alnpe_core1(opt, idx, seqs, offs)
    pc = aln.pe_counts()
    aln.close()
    ora = oracle_py.Oracle(w["prefix"])
    oo = ora.opt(l_overlap=opt.l_overlap, max_seed=opt.max_seed, max_locate=opt.max_locate, seed_only_ref=opt.seed_only_ref)
    want = ora.align_pe(oo, seqs, offs, opt.min_tlen, opt.max_tlen, n_threads=16)
    ora.close()
    idx.destroy()
    bad = oracle_py.compare(res, want, pe=True)
    detail = [(int(i), [(f, res[f][i].tolist(), want[f][i].tolist()) for f in ("pos", "strand", "n_diff", "is_gap", "mapq", "b0", "b1", "seq_start", "seq_end")]) for i in bad[:4]]
    assert len(bad) == 0, (len(bad), detail)
    assert pc[0] > 50 and pc[4] == 0, pc                # rescues ran, none overflowed


def test_gpu_pe_mixed_mate_lengths_match_oracle(tiny_pe):
    """Mates trimmed to 90 ... 150 bases in one batch.  k_swf packs two rescue requests per 8-lane group when both have the launch's read
    length (request 0's) and hands every other pair to k_swf1; the reverse pass and the traceback see every stripe count.  Every field
    against the oracle."""
    import sys
    import salt_amd
    from salt_amd import workload
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py
    w, _, _ = tiny_pe
    L = 150
    seqs, offs, _, _ = workload.make_pairs(w["genome"], w["snp_pos"], w["snp_mask"], 1500, L, seed=77, insert_mean=400, insert_sd=40, damaged=0.25, orphan=0.02)
    rng = np.random.default_rng(5)
    keep = np.where(rng.random(2 * 1500) < 0.5, L, rng.integers(90, L, 2 * 1500)).astype(np.uint32)
    keep[0] = L                                                      # request 0 sets the packed shape: most pairs keep it
    pieces = [seqs[offs[i]:offs[i] + keep[i]] for i in range(2 * 1500)]
    seqs2 = np.concatenate(pieces); offs2 = np.concatenate([[0], np.cumsum(keep)]).astype(np.uint32)
    idx = salt_amd.Index.reload(w["prefix"])
    opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", "250", "-b", "550"], idx.l_seed)
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=len(offs2) - 1, max_bases=int(offs2[-1]) + 64)
    res = aln.alnpe_core1(opt, idx, seqs2, offs2)
    pc = aln.pe_counts()
    aln.close()
    ora = oracle_py.Oracle(w["prefix"])
    oo = ora.opt(l_overlap=opt.l_overlap, max_seed=opt.max_seed, max_locate=opt.max_locate, seed_only_ref=opt.seed_only_ref)
    want = ora.align_pe(oo, seqs2, offs2, opt.min_tlen, opt.max_tlen, n_threads=16)
    ora.close()
    idx.destroy()
    bad = oracle_py.compare(res, want, pe=True)
    detail = [(int(i), [(f, res[f][i].tolist(), want[f][i].tolist()) for f in ("pos", "strand", "n_diff", "is_gap", "mapq", "b0", "b1", "seq_start", "seq_end")]) for i in bad[:4]]
    assert len(bad) == 0, (len(bad), detail)
    assert pc[0] > 300 and pc[4] == 0, pc                # rescues ran, none overflowed


def test_cli_pe_infers_the_insert_window_like_the_oracle(tmp_path):
    """`salt -p -b 0`: the window is inferred from the first batch (N3; the reference prints "not implemented" there).  The product CLI and
    the oracle CLI must announce the same window and print the same SAM."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    salt, salt_idx = os.path.join(root, "salt_amd", "bin", "salt"), os.path.join(root, "salt_amd", "bin", "salt-idx")
    oracle = os.path.join(root, "oracle", "salt_oracle")
    prefix = str(tmp_path / "idx")
    subprocess.run([salt_idx, "-k", "19", os.path.join(LAMBDA, "genome.fa"), os.path.join(LAMBDA, "snps.txt"), prefix], check=True, stderr=subprocess.DEVNULL)
    files = [os.path.join(LAMBDA, "reads_pe_1.fq"), os.path.join(LAMBDA, "reads_pe_2.fq")]
    got = subprocess.run([salt, "-d", "-c", "-p", "-b", "0", prefix] + files, capture_output=True)
    want = subprocess.run([oracle, "-d", "-c", "-p", "-b", "0", "-t", "8", prefix] + files, capture_output=True)
    assert got.returncode == 0 and want.returncode == 0, (got.stderr[-300:], want.stderr[-300:])
    line = lambda err: [l for l in err.decode().splitlines() if "insert size window" in l]
    assert line(got.stderr) and line(got.stderr) == line(want.stderr), (line(got.stderr), line(want.stderr))
    strip = lambda out: b"".join(l for l in out.splitlines(keepends=True) if not l.startswith(b"@PG"))
    assert strip(got.stdout) == strip(want.stdout)
    # too few pairs: a loud error, not a guess
    few = []
    for k, fn in enumerate(files):
        p = tmp_path / ("few_%d.fq" % k)
        p.write_bytes(b"".join(open(fn, "rb").readlines()[:40]))
        few.append(str(p))
    bad = subprocess.run([salt, "-p", "-b", "0", prefix] + few, capture_output=True)
    assert bad.returncode == 1 and b"cannot infer the insert size" in bad.stderr


def test_cli_pe_text_path_and_host_pipeline_give_the_reference_sam(tmp_path):
    """`salt -p` through the text path (both FASTQ files parsed and the pairs' SAM formatted by kernels; chunks of ~13 pairs here, so
    hundreds of them, cut by record count by the two scanner threads) and through the host pipeline: the reference's SAM both ways.
    Files with different numbers of reads end with an error, not with output."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    salt, salt_idx = os.path.join(root, "salt_amd", "bin", "salt"), os.path.join(root, "salt_amd", "bin", "salt-idx")
    prefix = str(tmp_path / "idx")
    subprocess.run([salt_idx, "-k", "19", os.path.join(LAMBDA, "genome.fa"), os.path.join(LAMBDA, "snps.txt"), prefix], check=True, stderr=subprocess.DEVNULL)
    files = [os.path.join(LAMBDA, "reads_pe_1.fq"), os.path.join(LAMBDA, "reads_pe_2.fq")]
    strip = lambda out: b"".join(l for l in out.splitlines(keepends=True) if not l.startswith(b"@PG"))
    for case in ("pe_default", "pe_r5"):
        want = open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
        for env in (dict(os.environ, SALT_CHUNK_BYTES="4000"), dict(os.environ, SALT_CHUNK_BYTES="250000"), dict(os.environ, SALT_HOST_PIPELINE="1")):
            out = subprocess.run([salt] + read_cases()[case] + ["-t", "8", prefix] + files, capture_output=True, env=env)
            assert out.returncode == 0, out.stderr[-400:]
            assert strip(out.stdout) == want, (case, env.get("SALT_CHUNK_BYTES"), _diff_report(strip(out.stdout), want))
            assert (b"text path (paired end)" in out.stderr) == ("SALT_HOST_PIPELINE" not in env)
    short = tmp_path / "short_2.fq"
    short.write_bytes(b"".join(open(files[1], "rb").readlines()[:-8]))
    bad = subprocess.run([salt, "-p", prefix, files[0], str(short)], capture_output=True, env=dict(os.environ, SALT_CHUNK_BYTES="4000"))
    assert bad.returncode == 1 and b"different numbers of reads" in bad.stderr, bad.stderr[-300:]


def test_cli_pe_text_path_reads_crlf_and_last_records_without_newline(tmp_path):
    """CRLF line ends in one file, a missing final newline in the other, a quality line that starts with '@': the paired-end text path
    (chunks cut by the scanner threads every 9 pairs) and the host pipeline print the same SAM."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    salt, salt_idx = os.path.join(root, "salt_amd", "bin", "salt"), os.path.join(root, "salt_amd", "bin", "salt-idx")
    prefix = str(tmp_path / "idx")
    subprocess.run([salt_idx, "-k", "19", os.path.join(LAMBDA, "genome.fa"), os.path.join(LAMBDA, "snps.txt"), prefix], check=True, stderr=subprocess.DEVNULL)
    l1 = open(os.path.join(LAMBDA, "reads_pe_1.fq"), "rb").read().split(b"\n")[:1200]
    l2 = open(os.path.join(LAMBDA, "reads_pe_2.fq"), "rb").read().split(b"\n")[:1200]
    for i in range(3, len(l2), 8):                          # every second quality line of the mates starts with '@'
        l2[i] = b"@" + l2[i][1:]
    f1, f2 = tmp_path / "crlf_1.fq", tmp_path / "nonl_2.fq"
    f1.write_bytes(b"\r\n".join(l1) + b"\r\n")
    f2.write_bytes(b"\n".join(l2))                          # no newline behind the last quality line
    cmd = [salt, "-d", "-c", "-p", "-a", "250", "-b", "550", prefix, str(f1), str(f2)]
    strip = lambda out: b"".join(l for l in out.splitlines(keepends=True) if not l.startswith(b"@PG"))
    a = subprocess.run(cmd, capture_output=True, env=dict(os.environ, SALT_CHUNK_BYTES="2500"))
    b = subprocess.run(cmd, capture_output=True, env=dict(os.environ, SALT_HOST_PIPELINE="1"))
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-300:], b.stderr[-300:])
    assert b"text path (paired end)" in a.stderr
    assert strip(a.stdout) == strip(b.stdout)
    assert strip(a.stdout).count(b"\n\n") == 600
