"""CPU-only AddressSanitizer + UBSan builds of the host-side code (index builder; the oracle CLI): the runs must finish without a
sanitizer report and give the golden bytes.  (GPU sanitizers are not available on the pool; device code is covered by the parity
tests.)"""
import os
import subprocess

import pytest

from conftest import GOLDEN, LAMBDA, ROOT

ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1")


@pytest.fixture(scope="module")
def asan_bins():
    subprocess.run(["make", "-C", os.path.join(ROOT, "salt_amd", "host"), "asan"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "salt_amd", "bin", "salt-idx.asan"), os.path.join(ROOT, "oracle", "salt_oracle.asan")


@pytest.mark.parametrize("case", ["lambda"] + sorted(os.listdir(os.path.join(GOLDEN, "index_cases"))))
def test_index_builder_under_asan_ubsan(case, asan_bins, tmp_path):
    d = LAMBDA if case == "lambda" else os.path.join(GOLDEN, "index_cases", case)
    prefix = str(tmp_path / "idx")
    p = subprocess.run([asan_bins[0], "--all-files", "-k", "19", os.path.join(d, "genome.fa"), os.path.join(d, "snps.txt"), prefix], capture_output=True, env=ENV)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"runtime error" not in p.stderr and b"AddressSanitizer" not in p.stderr, p.stderr.decode()[-2000:]
    for sfx in (".C.bwt", ".C.sa", ".R.backward.bwt", ".R.backward.occ"):
        assert open(prefix + sfx, "rb").read() == open(os.path.join(d, "idx" + sfx), "rb").read(), sfx
    import hashlib
    for sfx, sha in (l.split() for l in open(os.path.join(d, "idx.unread.sha256"))):      # --all-files: the R text files and the forward R index
        assert hashlib.sha256(open(prefix + sfx, "rb").read()).hexdigest() == sha, sfx


@pytest.mark.parametrize("case", ["se_default", "pe_default"])
def test_oracle_cli_under_asan_ubsan(case, asan_bins):
    from conftest import read_cases
    args = read_cases()[case]
    files = ["reads_pe_1.fq", "reads_pe_2.fq"] if case.startswith("pe") else ["reads_se.fq"]
    p = subprocess.run([asan_bins[1]] + args + [os.path.join(LAMBDA, "idx")] + [os.path.join(LAMBDA, f) for f in files], capture_output=True, env=ENV)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"runtime error" not in p.stderr and b"AddressSanitizer" not in p.stderr, p.stderr.decode()[-2000:]
    got = b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@PG"))
    assert got == open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()


@pytest.fixture(scope="module")
def salt_sanitized(tmp_path_factory, oracle_lib):
    """`salt` (salt_main.cc + the host library's sources) built with ASan+UBSan and with TSan against tests/stub/salt_gpu_stub.c: the
    driver's own code -- chunking, the paired-end scanners, workers, ordered output -- runs instrumented, the "device" is the oracle."""
    import shutil
    d = tmp_path_factory.mktemp("saltsan")
    os.makedirs(d / "lib")
    host = os.path.join(ROOT, "salt_amd", "host")
    subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-shared", "-o", str(d / "lib" / "libsalt_gpu.so"),
                    os.path.join(ROOT, "tests", "stub", "salt_gpu_stub.c"), os.path.join(ROOT, "oracle", "salt_oracle.c"), "-lm", "-lpthread"], check=True)
    bins = {}
    for tag, san in (("asan", "address,undefined"), ("tsan", "thread")):
        out = str(d / ("salt." + tag))
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-o", out,
                        os.path.join(host, "salt_main.cc"), os.path.join(host, "salt_host.cc"), os.path.join(host, "salt_idx.cc"),
                        "-L" + str(d / "lib"), "-lsalt_gpu", "-lz", "-lpthread", "-ldl"], check=True)
        bins[tag] = out
    prefix = str(d / "idx")
    subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt-idx"), "-k", "19", os.path.join(LAMBDA, "genome.fa"), os.path.join(LAMBDA, "snps.txt"), prefix],
                   check=True, stderr=subprocess.DEVNULL)
    return d, bins, prefix


@pytest.mark.parametrize("tag", ["asan", "tsan"])
@pytest.mark.parametrize("case", ["se_default", "pe_default"])
def test_salt_driver_under_sanitizers(tag, case, salt_sanitized):
    """Text path of `salt` with chunks of a few KB (hundreds of chunk boundaries inside records; for -p the two scanner threads and
    the workers of two devices): no AddressSanitizer / UBSan / ThreadSanitizer report, the reference's SAM."""
    from conftest import read_cases
    d, bins, prefix = salt_sanitized
    files = ["reads_pe_1.fq", "reads_pe_2.fq"] if case.startswith("pe") else ["reads_se.fq"]
    env = dict(ENV, SALT_STUB_PREFIX=prefix, SALT_CHUNK_BYTES="3001", LD_LIBRARY_PATH=str(d / "lib"),
               TSAN_OPTIONS="halt_on_error=0:exitcode=96:report_signal_unsafe=0")
    p = subprocess.run([bins[tag]] + read_cases()[case] + ["-t", "16", "--gpus", "2", prefix] + [os.path.join(LAMBDA, f) for f in files], capture_output=True, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    for word in (b"runtime error", b"AddressSanitizer", b"ThreadSanitizer"):
        assert word not in p.stderr, p.stderr.decode()[-3000:]
    got = b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@PG"))
    assert got == open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
