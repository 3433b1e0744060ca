"""N4 -- the oracle's restatement of the reference's SAM post-processor (`polish`, Polish_src/polish.c) against the outputs of the REAL
`polish` (oracle/_ref/polish, compiled in place by oracle/Makefile) committed under tests/golden/lambda/expect_polish_*.sam
(tests/golden/make_polish_fixture.py): Landau-Vishkin and Smith-Waterman re-scoring, single end and pairs, byte for byte."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, LAMBDA, ROOT

sys.path.insert(0, GOLDEN)
from make_polish_fixture import CASES, EDGE_CASES, polish_input          # noqa: E402


@pytest.mark.parametrize("out,args,src", CASES)
def test_polish_oracle_equals_the_reference(out, args, src, oracle_lib, tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "polish_oracle"], check=True, stdout=subprocess.DEVNULL)
    sam = tmp_path / "in.sam"
    sam.write_bytes(polish_input(os.path.join(LAMBDA, src), "-p" in args))
    p = subprocess.run([os.path.join(ROOT, "oracle", "polish_oracle")] + list(args) + [os.path.join(LAMBDA, "idx"), str(sam)], capture_output=True)
    assert p.returncode == 0, p.stderr[-300:]
    assert p.stdout == open(os.path.join(LAMBDA, out), "rb").read()


@pytest.mark.parametrize("out,args", EDGE_CASES)
def test_polish_oracle_equals_the_reference_on_edge_cases(out, args, oracle_lib):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "polish_oracle"], check=True, stdout=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(ROOT, "oracle", "polish_oracle")] + list(args) + [os.path.join(LAMBDA, "idx"), os.path.join(LAMBDA, "polish_edge_in.sam")], capture_output=True)
    assert p.returncode == 0, p.stderr[-300:]
    assert p.stdout == open(os.path.join(LAMBDA, out), "rb").read()
