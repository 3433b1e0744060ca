"""GPU parity tests proper: everything goes through the C ABI (libsalt_gpu.so) and is compared with
(a) the committed golden SAM the real reference printed and (b) the CPU oracle on seeded inputs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import LAMBDA, read_cases

pytestmark = pytest.mark.gpu

SE_CASES = [c for c in read_cases() if c.startswith("se_")]


@pytest.fixture(scope="module")
def lam():
    import salt_amd
    idx = salt_amd.Index.reload(os.path.join(LAMBDA, "idx"))
    aln = salt_amd.GpuAligner(idx, device=0, max_reads=4096)
    reads = salt_amd.read_fastq(os.path.join(LAMBDA, "reads_se.fq"))
    yield salt_amd, idx, aln, reads
    aln.close()
    idx.destroy()


@pytest.mark.parametrize("case", SE_CASES)
def test_gpu_sam_matches_reference_golden(case, lam):
    salt_amd, idx, aln, (names, seqs, offs, quals) = lam
    opt, _ = salt_amd.AlnOpt.from_argv(read_cases()[case], idx.l_seed)
    res = aln.alnse_core1(opt, seqs, offs)
    got = salt_amd.sam_text(idx, opt, names, seqs, offs, quals, res)
    want = open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        bad = [i for i in range(min(len(g), len(w))) if g[i] != w[i]]
        msg = "\n".join("line %d\n  got  %r\n  want %r" % (i, g[i][:400], w[i][:400]) for i in bad[:5])
        pytest.fail("%d differing lines (of %d)\n%s" % (len(bad), len(w), msg))
