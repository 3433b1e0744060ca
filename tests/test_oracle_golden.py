"""Pins the CPU oracle (oracle/salt_oracle.c) against the reference's own outputs.

* tests/golden/lambda/expect_se_*.sam were printed by the real reference (`oracle/_ref/salt`,
  compiled in place from /root/reference by oracle/Makefile) -- see tests/golden/make_fixtures.py.
* tests/golden/lv_vectors.txt was printed by oracle/ref_harness.c linked against the reference's
  LandauVishkin.c / editdistance.c.
Bar: byte-identical SAM, exact integers / CIGAR strings.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import EXTRA_CASES, LAMBDA, GOLDEN, read_cases

SE_CASES = [c for c in read_cases() if c.startswith("se_")]


@pytest.mark.parametrize("case", SE_CASES)
def test_oracle_sam_matches_reference(case, oracle_cli, tmp_path):
    args = read_cases()[case]
    out = subprocess.run([oracle_cli] + args + [os.path.join(LAMBDA, "idx"), os.path.join(LAMBDA, "reads_se.fq")],
                         check=True, capture_output=True).stdout
    want = open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
    assert out == want


@pytest.mark.parametrize("case", sorted(EXTRA_CASES))
def test_oracle_sam_matches_reference_on_boundary_and_ragged_reads(case, oracle_cli):
    """The reference's own SAM for reads straddling the contig boundary / hanging over the genome's ends
    (make_span_fixture.py) and for mixed read lengths 19..300 bp, SE and PE (make_ragged_fixture.py)."""
    args, files = EXTRA_CASES[case]
    out = subprocess.run([oracle_cli] + args + [os.path.join(LAMBDA, "idx")] + [os.path.join(LAMBDA, f) for f in files],
                         check=True, capture_output=True).stdout
    assert out == open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()


def test_oracle_threads_do_not_change_output(oracle_cli):
    base = [os.path.join(LAMBDA, "idx"), os.path.join(LAMBDA, "reads_se.fq")]
    a = subprocess.run([oracle_cli, "-d", "-c"] + base, check=True, capture_output=True).stdout
    b = subprocess.run([oracle_cli, "-d", "-c", "-t", "4"] + base, check=True, capture_output=True).stdout
    assert a == b


def test_oracle_units_match_reference_vectors(oracle_lib):
    lib = oracle_lib
    ref = None
    n = 0
    with open(os.path.join(GOLDEN, "lv_vectors.txt")) as f:
        for line in f:
            t = line.split()
            if t[0] == "R":
                l_ref = int(t[1])
                ref = np.array([int(x, 16) for x in t[2:]] + [0] * 8, dtype=np.uint32)
                rp = ref.ctypes.data_as(ctypes.c_void_p)
                continue
            pos, L, kmis, kdiff = map(int, t[1:5])
            seq = np.ascontiguousarray(np.frombuffer(t[5].encode(), dtype=np.uint8) - 48)
            sp = seq.ctypes.data_as(ctypes.c_void_p)
            mis, diff, cret, cig = int(t[6]), int(t[7]), int(t[8]), t[9]
            assert lib.so_ed_mismatch(rp, pos, sp, L, kmis) == mis
            assert lib.so_ed_diff(rp, l_ref, pos, L + 4, sp, L, kdiff) == diff
            if 0 <= diff < 31:
                buf = ctypes.create_string_buffer(160)
                assert lib.so_ed_diff_cigar(rp, pos, L + 4, sp, L, diff, buf, 128) == cret
                assert (buf.value.decode() or "-") == cig
            n += 1
    assert n == 4000


PE_CASES = [c for c in read_cases() if c.startswith("pe_")]


@pytest.mark.parametrize("case", PE_CASES)
def test_oracle_pe_sam_matches_reference(case, oracle_cli):
    """Paired end: pairing2 / pairing_singleton / SSW mate rescue (emulated lane by lane) / alnpe_sam against the
    SAM the real reference printed.  (The reference's PE locate calls rand() when an R interval exceeds max_locate;
    this fixture has none, so its output is deterministic.)"""
    args = read_cases()[case]
    out = subprocess.run([oracle_cli] + args + [os.path.join(LAMBDA, "idx"), os.path.join(LAMBDA, "reads_pe_1.fq"),
                                                os.path.join(LAMBDA, "reads_pe_2.fq")], check=True, capture_output=True).stdout
    want = open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
    assert out == want


def _ssw_vectors():
    out = []
    with open(os.path.join(GOLDEN, "ssw_vectors.txt")) as f:
        for line in f:
            t = line.split()
            out.append((int(t[1]), np.array([int(c, 16) for c in t[2]], dtype=np.uint8),
                        np.frombuffer(t[3].encode(), dtype=np.uint8) - 48, [int(x) for x in t[4:10]], t[10]))
    return out


def test_oracle_ssw_matches_reference_vectors(oracle_lib):
    """SSW 0.1.4 word kernel (forward, reverse, second best, banded traceback) emulated lane by lane, against answers
    printed by the reference's own ssw.c (oracle/ref_harness_ssw.c) for both score matrices."""
    lib = oracle_lib
    n = 0
    for aware, ref, codes, want6, cig in _ssw_vectors():
        out6 = (ctypes.c_int * 6)()
        buf = ctypes.create_string_buffer(512)
        codes = np.ascontiguousarray(codes)
        lib.so_ssw_unit(aware, ref.ctypes.data_as(ctypes.c_void_p), len(ref), codes.ctypes.data_as(ctypes.c_void_p), len(codes), out6, buf, 512)
        assert list(out6) == want6, (n, list(out6), want6)
        assert (buf.value.decode() or "-") == cig, (n, buf.value, cig)
        n += 1
    assert n == 600
