"""Parity on the BASELINE.json workloads themselves (SURVEY 8d configs 2, 3, 4 restated as seeded synthetic data), through the C ABI,
against the CPU oracle:
  config 2   chr21-scale: 40 Mbp, 190 000 SNPs, the WHOLE batch of 1 000 000 x 100-base SE reads, every result field
  config 4'  the same genome, 2 x 150-base pairs (-p -a 250 -b 550): 60 000 pairs, every field after pairing / rescue
  config 3   GRCh38-scale: 3.1e9 bases in 24 contigs, 14.8 M SNPs, indexed here by the device suffix sorter: 200 000 SE reads
  config 4   2 x 150-base pairs on that index: 20 000 pairs
Each index is built once per session with the product's own salt-idx (device backend).  At BASELINE's FULL sizes (50 M reads, 50 M
pairs on the GRCh38-scale index) the size-independent properties: a simulated read / pair comes back where it was drawn from, a
result does not depend on the batch its read travels in, the same call twice gives the same bytes."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _workload(name, tmp):
    import torch
    import salt_amd
    from salt_amd import workload
    dev = torch.device("cuda", 0)
    g, p, m = workload.generate_device(name, dev)
    w = workload.prepare(name, str(tmp), gpu_device=0, arrays=(g, p, m))
    return dict(w=w, genome=g, pos=p, mask=m, site=workload.make_site_map(g.numel(), p, m))


@pytest.fixture(scope="module")
def chr21(tmp_path_factory):
    return _workload("chr21", tmp_path_factory.mktemp("chr21"))


@pytest.fixture(scope="module")
def grch38(tmp_path_factory):
    d = _workload("grch38", tmp_path_factory.mktemp("grch38"))
    yield d
    for f in os.listdir(d["w"]["dir"]):                    # 6 GB of index files
        os.unlink(os.path.join(d["w"]["dir"], f))


def _se_parity(d, n_reads, seed, threads=64):
    import salt_amd
    import oracle_py
    from salt_amd import workload
    w = d["w"]
    seqs, offs, start, rev = workload.make_reads_hash(d["genome"], d["site"], n_reads, 100, seed=seed, batch=0)
    hs, ho = seqs.cpu().numpy(), offs.cpu().numpy().view(np.uint32)
    idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=n_reads, max_bases=n_reads * 100)
    opt = salt_amd.AlnOpt(l_seed=w["k"])
    res = aln.alnse_core1(opt, hs, ho).copy()
    aln.close()
    idx.destroy()
    ora = oracle_py.Oracle(w["prefix"])
    want = ora.align(ora.opt(), hs, ho, n_threads=min(os.cpu_count() or 1, threads))
    ora.close()
    bad = oracle_py.compare(res, want)
    mapped = res["pos"] != 0xFFFFFFFF
    exact = mapped & (res["pos"].astype(np.int64) == start.cpu().numpy())
    return bad, float(mapped.mean()), float(exact.mean()), res, want


def _pe_parity(d, n_pairs, seed):
    import salt_amd
    import oracle_py
    from salt_amd import workload
    w = d["w"]
    g, p, m = d["genome"].cpu().numpy(), d["pos"].cpu().numpy(), d["mask"].cpu().numpy()
    seqs, offs, _, _ = workload.make_pairs(g, p, m, n_pairs, 150, seed=seed, insert_mean=400, insert_sd=50, damaged=0.03, orphan=0.01)
    idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
    opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", "250", "-b", "550"], idx.l_seed)
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=2 * n_pairs, max_bases=2 * n_pairs * 150)
    res = aln.alnpe_core1(opt, idx, seqs, offs).copy()
    aln.close()
    idx.destroy()
    ora = oracle_py.Oracle(w["prefix"])
    oo = ora.opt(l_overlap=opt.l_overlap, max_seed=opt.max_seed, max_locate=opt.max_locate, seed_only_ref=opt.seed_only_ref)
    want = ora.align_pe(oo, seqs, offs, opt.min_tlen, opt.max_tlen, n_threads=min(os.cpu_count() or 1, 64))
    ora.close()
    bad = oracle_py.compare(res, want, pe=True)
    rescued = int(((want["seq_start"] != 0) | (want["seq_end"] != 149)).sum())
    return bad, float((res["pos"] != 0xFFFFFFFF).mean()), rescued, res, want


def _detail(bad, res, want):
    return [(int(i), [(f, res[f][i].tolist(), want[f][i].tolist()) for f in ("pos", "strand", "n_diff", "is_gap", "mapq", "b0", "b1")]) for i in bad[:4]]


def test_config2_chr21_whole_se_batch_equals_the_oracle(chr21):
    bad, mapped, exact, res, want = _se_parity(chr21, 1_000_000, seed=1)
    assert len(bad) == 0, (len(bad), _detail(bad, res, want))
    assert mapped > 0.995 and exact > 0.99, (mapped, exact)


def test_config4_shape_on_chr21_pairs_equal_the_oracle(chr21):
    bad, mapped, rescued, res, want = _pe_parity(chr21, 60_000, seed=3)
    assert len(bad) == 0, (len(bad), _detail(bad, res, want))
    assert mapped > 0.97 and rescued > 500, (mapped, rescued)


def test_config3_grch38_scale_se_reads_equal_the_oracle(grch38):
    """3.1e9 bases: suffix-array rows and positions beyond 2^31, a 12 GB suffix array and a 64 GiB k-mer table on the device, 24
    contigs; the index was built a moment ago by the device suffix sorter."""
    bad, mapped, exact, res, want = _se_parity(grch38, 200_000, seed=2)
    assert len(bad) == 0, (len(bad), _detail(bad, res, want))
    assert mapped > 0.995 and exact > 0.99, (mapped, exact)
    assert int((res["pos"][res["pos"] != 0xFFFFFFFF] > 2**31).sum()) > 10_000        # positions in the upper half of the u32 range are exercised


def test_config4_grch38_scale_pairs_equal_the_oracle(grch38):
    bad, mapped, rescued, res, want = _pe_parity(grch38, 20_000, seed=4)
    assert len(bad) == 0, (len(bad), _detail(bad, res, want))
    assert mapped > 0.97 and rescued > 150, (mapped, rescued)


def test_config3_cli_fastq_to_sam_equals_the_oracle_cli(grch38, tmp_path):
    """The drop-in command on the GRCh38-scale index, end to end: `salt -d -c idx reads.fq > out.sam` (text path: FASTQ parsed and SAM
    formatted by kernels, workers that pread / call the device / write in turn, chunks of 8 MiB so that several workers and chunk boundaries take part) against
    the oracle CLI's SAM for the same 300 000 reads, byte for byte (the @PG line carries the command line and is dropped)."""
    import subprocess
    from salt_amd import workload
    w = grch38["w"]
    n = 300_000
    seqs, _, _, _ = workload.make_reads_hash(grch38["genome"], grch38["site"], n, 100, seed=9, batch=0)
    fq = str(tmp_path / "reads.fq")
    with open(fq, "wb") as f:
        f.write(workload.fastq_bytes(seqs.cpu().numpy(), n, 100))
    got_fn, want_fn = str(tmp_path / "gpu.sam"), str(tmp_path / "cpu.sam")
    with open(got_fn, "wb") as fo:
        p = subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt"), "-d", "-c", "-t", "32", w["prefix"], fq], stdout=fo, stderr=subprocess.PIPE,
                           env=dict(os.environ, SALT_CHUNK_MB="8"))
    assert p.returncode == 0 and b"text path" in p.stderr, p.stderr[-400:]
    with open(want_fn, "wb") as fo:
        q = subprocess.run([os.path.join(ROOT, "oracle", "salt_oracle"), "-d", "-c", "-t", str(min(os.cpu_count() or 1, 64)), w["prefix"], fq], stdout=fo, stderr=subprocess.PIPE)
    assert q.returncode == 0, q.stderr[-400:]
    strip = lambda fn: [l for l in open(fn, "rb").read().split(b"\n") if not l.startswith(b"@PG")]
    g, c = strip(got_fn), strip(want_fn)
    bad = [i for i in range(min(len(g), len(c))) if g[i] != c[i]]
    assert len(g) == len(c) and not bad, (len(g), len(c), len(bad), [(g[i][:160], c[i][:160]) for i in bad[:2]])


def _fields_on_device(d_res, n, names):
    """columns of the salt_result_t rows a resident call left in HBM, as torch tensors (no copy of the 880-byte rows to the host)"""
    import torch
    import salt_amd
    rows = d_res.view(-1, salt_amd.RESULT_DTYPE.itemsize)[:n]
    out = {}
    for f in names:
        dt, off = salt_amd.RESULT_DTYPE.fields[f][0], salt_amd.RESULT_DTYPE.fields[f][1]
        col = rows[:, off:off + dt.itemsize].contiguous()
        out[f] = col.view({1: torch.uint8, 2: torch.int16, 4: torch.int32}[dt.itemsize]).reshape(-1).to(torch.int64)
    if "pos" in out:
        out["pos"] = out["pos"] & 0xFFFFFFFF
    return out


def _defined_equal(a, b):
    """a, b: (n, 880) uint8 rows in HBM.  True when every DEFINED byte agrees: the header, hits[s][j] for j < n_hits[s], the CIGAR up to
    n_cigar, hit_n_cigar / hit_cigar of the hits listed (what lies behind those counts is whatever the buffer held before)."""
    import torch
    import salt_amd
    from salt_amd import api
    off = {f: salt_amd.RESULT_DTYPE.fields[f][1] for f in salt_amd.RESULT_DTYPE.names}
    ok = (a[:, :off["hits"]] == b[:, :off["hits"]]).all(dim=1)
    nh = a[:, off["n_hits"]:off["n_hits"] + 2].to(torch.int64)
    ar5 = torch.arange(api.MAX_HITS, device=a.device)[None, :]
    for s_ in range(2):
        lo = off["hits"] + s_ * api.MAX_HITS * 8
        eq = (a[:, lo:lo + 8 * api.MAX_HITS] == b[:, lo:lo + 8 * api.MAX_HITS]).view(-1, api.MAX_HITS, 8).all(dim=2)
        ok &= (eq | (ar5 >= nh[:, s_:s_ + 1])).all(dim=1)
    tot = nh.sum(dim=1, keepdim=True).clamp(max=api.MAX_HITS)
    hnc_a = a[:, off["hit_n_cigar"]:off["hit_n_cigar"] + api.MAX_HITS].to(torch.int64)
    hnc_b = b[:, off["hit_n_cigar"]:off["hit_n_cigar"] + api.MAX_HITS].to(torch.int64)
    ok &= ((hnc_a == hnc_b) | (ar5 >= tot)).all(dim=1)
    ar64 = torch.arange(api.MAX_CIGAR_OPS, device=a.device)[None, :]
    nc = a[:, off["n_cigar"]].to(torch.int64)[:, None]
    w = 2 * api.MAX_CIGAR_OPS
    eq = (a[:, off["cigar"]:off["cigar"] + w] == b[:, off["cigar"]:off["cigar"] + w]).view(-1, api.MAX_CIGAR_OPS, 2).all(dim=2)
    ok &= (eq | (ar64 >= nc)).all(dim=1)
    for h in range(api.MAX_HITS):
        lo = off["hit_cigar"] + h * w
        eq = (a[:, lo:lo + w] == b[:, lo:lo + w]).view(-1, api.MAX_CIGAR_OPS, 2).all(dim=2)
        live = (h < tot[:, 0])[:, None] & (ar64 < hnc_a[:, h:h + 1])
        ok &= (eq | ~live).all(dim=1)
    return bool(ok.all()), int((~ok).sum())


def test_config3_full_size_50M_se_reads_by_properties(grch38):
    """BASELINE configs[2] at its full size -- 50 M x 100-base reads on the GRCh38-scale index -- through the properties that do not
    need an oracle run of 6 CPU-minutes per million: (1) a simulated read comes back at the position and strand it was drawn from
    (reads are independent, so the fraction is the one the oracle-checked 200 000 have); (2) a result does not depend on the batch
    its read travels in: two of the 50 batches aligned again as ONE batch of 2 M reads in the other order give the same rows (every
    defined byte); (3) the same call twice gives the same bytes."""
    import torch
    import salt_amd
    from salt_amd import workload
    w, dev = grch38["w"], torch.device("cuda", 0)
    n, L, n_batches = 1_000_000, 100, 50
    idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=2 * n, max_bases=2 * n * L)
    opt = salt_amd.AlnOpt(l_seed=w["k"])
    isz = salt_amd.RESULT_DTYPE.itemsize
    d_res = torch.zeros(2 * n * isz, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    mapped = exact = 0
    keep = {}
    for b in range(n_batches):
        seqs, offs, start, rev = workload.make_reads_hash(grch38["genome"], grch38["site"], n, L, seed=11, batch=b)
        aln.align_resident(opt, n, L, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st)
        torch.cuda.synchronize()
        f = _fields_on_device(d_res, n, ("pos", "strand"))
        ok = f["pos"] != 0xFFFFFFFF
        mapped += int(ok.sum())
        exact += int((ok & (f["pos"] == start) & (f["strand"] == rev.to(torch.int64))).sum())
        if b in (7, 31):
            keep[b] = (seqs.clone(), d_res[:n * isz].clone())
        if b == 7:                                         # (3) the same call again
            aln.align_resident(opt, n, L, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st)
            torch.cuda.synchronize()
            assert torch.equal(d_res[:n * isz], keep[7][1])
    total = n * n_batches
    assert mapped / total > 0.995 and exact / total > 0.99, (mapped / total, exact / total)
    # (2) batches 31 and 7 as one batch of 2 M reads
    seqs2 = torch.cat([keep[31][0], keep[7][0]])
    offs2 = (torch.arange(2 * n + 1, dtype=torch.int64, device=dev) * L).to(torch.int32)
    aln.align_resident(opt, 2 * n, L, seqs2.data_ptr(), offs2.data_ptr(), d_res.data_ptr(), st)
    torch.cuda.synchronize()
    rows = d_res.view(-1, isz)
    assert _defined_equal(rows[:n], keep[31][1].view(-1, isz)) == (True, 0)
    assert _defined_equal(rows[n:2 * n], keep[7][1].view(-1, isz)) == (True, 0)
    aln.close()
    idx.destroy()


def test_config4_full_size_50M_pairs_by_properties(grch38):
    """BASELINE configs[3] at its full size -- 50 M pairs of 2 x 150 bases, -p -a 250 -b 550 -- by properties: both mates of a simulated
    pair come back where the fragment was drawn (position and strand), and a pair's rows do not depend on the batch it travels in
    (two batches again as one, in the other order: the same bytes)."""
    import torch
    import salt_amd
    from salt_amd import workload
    w, dev = grch38["w"], torch.device("cuda", 0)
    n_pairs, L, n_batches = 500_000, 150, 100
    idx = salt_amd.Index.reload(w["prefix"], rebuild_lkt=False)
    opt, _ = salt_amd.AlnOpt.from_argv(["-p", "-a", "250", "-b", "550"], idx.l_seed)
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=4 * n_pairs, max_bases=4 * n_pairs * L)
    isz = salt_amd.RESULT_DTYPE.itemsize
    d_res = torch.zeros(4 * n_pairs * isz, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    both = exact = 0
    keep = {}
    for b in range(n_batches):
        seqs, offs, s1, s2, fl = workload.make_pairs_hash(grch38["genome"], grch38["site"], n_pairs, L, seed=12, batch=b)
        aln.align_pe_resident(opt, idx, n_pairs, L, seqs.data_ptr(), offs.data_ptr(), d_res.data_ptr(), st)
        torch.cuda.synchronize()
        f = _fields_on_device(d_res, 2 * n_pairs, ("pos", "strand"))
        p1, p2, t1, t2 = f["pos"][0::2], f["pos"][1::2], f["strand"][0::2], f["strand"][1::2]
        ok = (p1 != 0xFFFFFFFF) & (p2 != 0xFFFFFFFF)
        both += int(ok.sum())
        flq = fl.to(torch.int64)
        exact += int((ok & (p1 == s1) & (p2 == s2) & (t1 == flq) & (t2 == 1 - flq)).sum())
        if b in (3, 64):
            keep[b] = (seqs.clone(), d_res[:2 * n_pairs * isz].clone())
    total = n_pairs * n_batches
    assert both / total > 0.98 and exact / total > 0.95, (both / total, exact / total)
    seqs2 = torch.cat([keep[64][0], keep[3][0]])
    offs2 = (torch.arange(4 * n_pairs + 1, dtype=torch.int64, device=dev) * L).to(torch.int32)
    aln.align_pe_resident(opt, idx, 2 * n_pairs, L, seqs2.data_ptr(), offs2.data_ptr(), d_res.data_ptr(), st)
    torch.cuda.synchronize()
    rows = d_res.view(-1, isz)
    assert _defined_equal(rows[:2 * n_pairs], keep[64][1].view(-1, isz)) == (True, 0)
    assert _defined_equal(rows[2 * n_pairs:4 * n_pairs], keep[3][1].view(-1, isz)) == (True, 0)
    aln.close()
    idx.destroy()
