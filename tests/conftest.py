"""Shared fixtures.  `-m "not gpu"` runs here (no GPU); `-m gpu` runs on the MI355X box."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
LAMBDA = os.path.join(GOLDEN, "lambda")
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    # torch must initialise its HIP context before libsalt_gpu.so touches the device, whatever order the test files run in
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def read_cases():
    cases = {}
    with open(os.path.join(LAMBDA, "cases.txt")) as f:
        for line in f:
            name, args = line.rstrip("\n").split("\t")
            cases[name] = args.split()
    return cases


# fixtures beyond cases.txt, each with its own FASTQ (tests/golden/make_span_fixture.py, make_ragged_fixture.py):
# name -> (salt arguments, FASTQ file names under tests/golden/lambda)
EXTRA_CASES = {
    "span_default": (["-d", "-c"], ["reads_span.fq"]),
    "span_r5": (["-d", "-c", "-r", "5"], ["reads_span.fq"]),
    "ragged_default": (["-d", "-c"], ["reads_ragged.fq"]),
    "ragged_r7_s10": (["-d", "-c", "-r", "7", "-s", "10"], ["reads_ragged.fq"]),
    "ragged_pe": (["-d", "-p", "-c", "-a", "300", "-b", "700"], ["reads_ragged_pe_1.fq", "reads_ragged_pe_2.fq"]),
}


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU restatement (checker only).  Built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "libsalt_oracle.so")
    exe = os.path.join(ROOT, "oracle", "salt_oracle")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("salt_oracle.c", "salt_oracle.h", "salt_oracle_main.c")]
    if (not os.path.exists(so) or not os.path.exists(exe)
            or max(os.path.getmtime(s) for s in srcs) > min(os.path.getmtime(so), os.path.getmtime(exe))):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True,
                       stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.so_ed_mismatch.restype = ctypes.c_int
    lib.so_ed_diff.restype = ctypes.c_int
    lib.so_ed_diff_cigar.restype = ctypes.c_int
    lib.so_index_load.restype = ctypes.c_void_p
    lib.so_index_load.argtypes = [ctypes.c_char_p]
    return lib


@pytest.fixture(scope="session")
def oracle_cli(oracle_lib):
    return os.path.join(ROOT, "oracle", "salt_oracle")
