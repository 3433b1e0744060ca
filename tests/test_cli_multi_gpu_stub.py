"""Host side of `salt --gpus N` without a GPU: the real `salt` binary and libsalt_host.so run against
tests/stub/salt_gpu_stub.c, a stand-in for libsalt_gpu.so whose "devices" are labels and whose per-batch work is the CPU oracle.
What is tested is the driver: option handling, chunks of the FASTQ file dealt to the workers of two devices, SAM blocks put out
in input order (the reference's puts loop, Align_src/alnse.c:1433-1439) -- to a pipe and to a regular file."""
import os
import shutil
import subprocess

import pytest

from conftest import LAMBDA, ROOT, read_cases


@pytest.fixture(scope="module")
def stub_tree(tmp_path_factory, oracle_lib):
    d = tmp_path_factory.mktemp("stubtree")
    os.makedirs(d / "bin"); os.makedirs(d / "lib")
    subprocess.run(["make", "-C", os.path.join(ROOT, "salt_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    shutil.copy(os.path.join(ROOT, "salt_amd", "bin", "salt"), d / "bin" / "salt")
    shutil.copy(os.path.join(ROOT, "salt_amd", "lib", "libsalt_host.so"), d / "lib" / "libsalt_host.so")
    subprocess.run(["gcc", "-O2", "-g", "-fPIC", "-shared", "-Wall", "-o", str(d / "lib" / "libsalt_gpu.so"),
                    os.path.join(ROOT, "tests", "stub", "salt_gpu_stub.c"), os.path.join(ROOT, "oracle", "salt_oracle.c"), "-lm", "-lpthread"], check=True)
    prefix = str(d / "idx")
    subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt-idx"), "-k", "19", os.path.join(LAMBDA, "genome.fa"), os.path.join(LAMBDA, "snps.txt"), prefix],
                   check=True, stderr=subprocess.DEVNULL)
    return d, prefix


@pytest.mark.parametrize("case,gpus", [("se_default", 2), ("se_r5_s4_m16", 2), ("se_refonly", 3), ("se_default", 1)])
def test_salt_gpus_n_deals_chunks_and_keeps_the_input_order(case, gpus, stub_tree, tmp_path):
    d, prefix = stub_tree
    want = open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
    strip = lambda out: b"".join(l for l in out.splitlines(keepends=True) if not l.startswith(b"@PG"))
    log = str(tmp_path / "stub.log")
    env = dict(os.environ, SALT_STUB_PREFIX=prefix, SALT_STUB_LOG=log, SALT_CHUNK_BYTES="9000", LD_LIBRARY_PATH=str(d / "lib"))
    cmd = [str(d / "bin" / "salt")] + read_cases()[case] + ["-t", "16", "--gpus", str(gpus), prefix, os.path.join(LAMBDA, "reads_se.fq")]
    out = subprocess.run(cmd, capture_output=True, env=env)
    assert out.returncode == 0, out.stderr[-500:]
    assert strip(out.stdout) == want
    rows = [l.split() for l in open(log).read().splitlines()]
    assert sum(int(r[1]) for r in rows) == 2000 and len(rows) > 10            # many chunks, every read exactly once
    assert {int(r[0]) for r in rows} == set(range(gpus))                      # every device took chunks
    f = tmp_path / "out.sam"
    with open(f, "wb") as fo:
        out = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, env=env)
    assert out.returncode == 0 and b"blocks written in turn" in out.stderr, out.stderr[-300:]
    assert strip(open(f, "rb").read()) == want


def test_salt_reports_a_device_error_and_exits_nonzero(stub_tree, tmp_path):
    """Paired end goes through the host pipeline, which the stub does not serve: the device error must reach stderr and the exit code."""
    d, prefix = stub_tree
    env = dict(os.environ, SALT_STUB_PREFIX=prefix, SALT_STUB_NO_PE="1", LD_LIBRARY_PATH=str(d / "lib"))
    out = subprocess.run([str(d / "bin" / "salt"), "-p", prefix, os.path.join(LAMBDA, "reads_pe_1.fq"), os.path.join(LAMBDA, "reads_pe_2.fq")],
                         capture_output=True, env=env, timeout=120)
    assert out.returncode == 1 and b"not part of the stub" in out.stderr


@pytest.mark.parametrize("case,gpus", [("pe_default", 2), ("pe_r5", 3)])
def test_salt_pe_text_path_cuts_both_files_by_record_count(case, gpus, stub_tree, tmp_path):
    """The paired-end driver: two scanner threads publish the offsets of every n-th record of both files, workers of several
    "devices" take chunk k of both, blocks come out in input order.  Chunks of a dozen pairs; mates whose records differ in length
    between the two files (names of different lengths), so that the same chunk has different byte ranges in the two files."""
    d, prefix = stub_tree
    want = open(os.path.join(LAMBDA, "expect_%s.sam" % case), "rb").read()
    strip = lambda out: b"".join(l for l in out.splitlines(keepends=True) if not l.startswith(b"@PG"))
    f1 = os.path.join(LAMBDA, "reads_pe_1.fq")
    lines = open(os.path.join(LAMBDA, "reads_pe_2.fq"), "rb").read().split(b"\n")
    for i in range(0, len(lines) - 3, 4):                 # a comment of growing length behind every second-mate name: same SAM, other offsets
        lines[i] = lines[i] + b" " + b"c" * (i % 37)
    f2 = tmp_path / "mates.fq"
    f2.write_bytes(b"\n".join(lines))
    log = str(tmp_path / "stub.log")
    env = dict(os.environ, SALT_STUB_PREFIX=prefix, SALT_STUB_LOG=log, SALT_CHUNK_BYTES="3000", LD_LIBRARY_PATH=str(d / "lib"))
    out = subprocess.run([str(d / "bin" / "salt")] + read_cases()[case] + ["-t", "16", "--gpus", str(gpus), prefix, f1, str(f2)], capture_output=True, env=env)
    assert out.returncode == 0, out.stderr[-500:]
    assert b"text path (paired end)" in out.stderr
    assert strip(out.stdout) == want
    rows = [l.split() for l in open(log).read().splitlines()]
    assert sum(int(r[1]) for r in rows) == 1000 and len(rows) > 20
    assert {int(r[0]) for r in rows} == set(range(gpus))
    # one read less in the second file: an error, no shifted pairs
    f3 = tmp_path / "short.fq"
    f3.write_bytes(b"\n".join(lines[:-5]) + b"\n")
    bad = subprocess.run([str(d / "bin" / "salt")] + read_cases()[case] + [prefix, f1, str(f3)], capture_output=True, env=env)
    assert bad.returncode == 1 and b"different numbers of reads" in bad.stderr, bad.stderr[-300:]
