"""N4 -- the product's `polish` (salt_amd/host/polish_main.cc; every edit distance and CIGAR by k_polish on the GPU) against the outputs
(default) or by k_sw with polish's matrix (-s)) against the outputs of the REAL reference `polish` committed under
tests/golden/lambda/expect_polish_*.sam, including hand-made edge cases: windows clipped at the genome end (the reference's shrinking
window + stale buffer), many XA hits."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, LAMBDA, ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, GOLDEN)
from make_polish_fixture import CASES, EDGE_CASES, polish_input          # noqa: E402

POLISH = os.path.join(ROOT, "salt_amd", "bin", "polish")


@pytest.fixture(scope="module")
def lam_index(tmp_path_factory):
    prefix = str(tmp_path_factory.mktemp("polidx") / "idx")
    subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt-idx"), "-k", "19", os.path.join(LAMBDA, "genome.fa"), os.path.join(LAMBDA, "snps.txt"), prefix],
                   check=True, stderr=subprocess.DEVNULL)
    return prefix


@pytest.mark.parametrize("out,args,src", CASES)
def test_polish_equals_the_reference(out, args, src, lam_index, tmp_path):
    sam = tmp_path / "in.sam"
    sam.write_bytes(polish_input(os.path.join(LAMBDA, src), "-p" in args))
    p = subprocess.run([POLISH] + list(args) + [lam_index, str(sam)], capture_output=True)
    assert p.returncode == 0, p.stderr[-300:]
    assert p.stdout == open(os.path.join(LAMBDA, out), "rb").read()


def test_polish_needs_the_gpu_library_entry_points():
    """both scoring modes are device calls: the binary links salt_gpu_polish_lv and salt_gpu_polish_sw and nothing that scores on the host"""
    syms = subprocess.run(["nm", "-D", "--undefined-only", POLISH], capture_output=True, text=True).stdout
    assert "salt_gpu_polish_lv" in syms and "salt_gpu_polish_sw" in syms


@pytest.mark.parametrize("out,args", EDGE_CASES)
def test_polish_edge_cases_equal_the_reference(out, args, lam_index):
    """tests/polish_edge.py: reads with a dozen XA hits on both strands and in both contigs, hits that coincide after the offset sort,
    pairs that are / are not 350..650 apart, and hits whose window runs past the end of the genome (the reference clips the window, keeps
    the shorter length for the rest of the record and leaves the previous window's bases behind the clip).  Expected output: the REAL
    reference's (tests/golden/make_polish_fixture.py)."""
    p = subprocess.run([POLISH] + list(args) + [lam_index, os.path.join(LAMBDA, "polish_edge_in.sam")], capture_output=True)
    assert p.returncode == 0, p.stderr[-300:]
    g, w = p.stdout.split(b"\n"), open(os.path.join(LAMBDA, out), "rb").read().split(b"\n")
    bad = [i for i in range(min(len(g), len(w))) if g[i] != w[i]]
    assert len(g) == len(w) and not bad, (len(g), len(w), len(bad), [(g[i][:120], w[i][:120]) for i in bad[:2]])
