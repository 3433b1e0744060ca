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
    """A device error must reach stderr and the exit code (the stub refuses paired-end calls here)."""
    d, prefix = stub_tree
    env = dict(os.environ, SALT_STUB_PREFIX=prefix, SALT_STUB_NO_PE="1", LD_LIBRARY_PATH=str(d / "lib"))
    out = subprocess.run([str(d / "bin" / "salt"), "-p", prefix, os.path.join(LAMBDA, "reads_pe_1.fq"), os.path.join(LAMBDA, "reads_pe_2.fq")],
                         capture_output=True, env=env, timeout=120)
    assert out.returncode == 1 and b"not part of the stub" in out.stderr


@pytest.mark.parametrize("case,gpus", [("pe_default", 2), ("pe_r5", 3)])
def test_salt_pe_text_path_cuts_both_files_by_record_count(case, gpus, stub_tree, tmp_path):
    """The paired-end driver: the scanners of the two files publish the offsets of every n-th record, workers of several
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
    # the scanners count lines in segments on several threads each: a few hundred segments per file instead of one
    for seg in ("1000", "4097", "70000"):
        seg_out = subprocess.run([str(d / "bin" / "salt")] + read_cases()[case] + ["-t", "16", "--gpus", str(gpus), prefix, f1, str(f2)], capture_output=True,
                                 env=dict(env, SALT_PE_SCAN_SEG_BYTES=seg, SALT_STUB_LOG=str(tmp_path / "seg.log")))
        assert seg_out.returncode == 0 and strip(seg_out.stdout) == want, (seg, seg_out.stderr[-300:])
    rows = [l.split() for l in open(log).read().splitlines()]
    assert sum(int(r[1]) for r in rows) == 1000 and len(rows) > 20
    assert {int(r[0]) for r in rows} == set(range(gpus))
    # one read less in the second file: an error, no shifted pairs
    f3 = tmp_path / "short.fq"
    f3.write_bytes(b"\n".join(lines[:-5]) + b"\n")
    bad = subprocess.run([str(d / "bin" / "salt")] + read_cases()[case] + [prefix, f1, str(f3)], capture_output=True, env=env)
    assert bad.returncode == 1 and b"different numbers of reads" in bad.stderr, bad.stderr[-300:]


def _strip(out):
    return b"".join(l for l in out.splitlines(keepends=True) if not l.startswith(b"@PG"))


@pytest.mark.parametrize("gpus", [1, 8])
def test_salt_text_path_hands_over_to_the_host_parser_where_the_input_stops_being_four_line(gpus, stub_tree, tmp_path):
    """the text path is chosen from the first 256 KiB; a later multi-line record, a blank line or a blank last line is
    something kseq.h reads (query.c:103-239), so the chunk the device parser refuses and everything behind it goes to the host
    pipeline -- after the blocks before it are out, once, in order.  Eight device labels: every label takes chunks."""
    d, prefix = stub_tree
    want = open(os.path.join(LAMBDA, "expect_se_default.sam"), "rb").read()
    recs = open(os.path.join(LAMBDA, "reads_se.fq"), "rb").read().split(b"\n")
    recs = [recs[i:i + 4] for i in range(0, len(recs) - 3, 4)]
    assert len(recs) == 2000
    out = []
    for i, r in enumerate(recs):
        if i == 1500:                                           # far behind the sniffed head: sequence and quality wrapped over three lines
            out += [r[0], r[1][:40], r[1][40:77], r[1][77:], r[2], r[3][:15], r[3][15:]]
        elif i == 1700:
            out += [b""] + r                                    # a blank line between records
        else:
            out += r
    fq = tmp_path / "late_multiline.fq"
    fq.write_bytes(b"\n".join(out) + b"\n\n\n")                 # and blank lines at the end
    log = str(tmp_path / "stub.log")
    env = dict(os.environ, SALT_STUB_PREFIX=prefix, SALT_STUB_LOG=log, SALT_CHUNK_BYTES="9000", LD_LIBRARY_PATH=str(d / "lib"))
    res = subprocess.run([str(d / "bin" / "salt")] + read_cases()["se_default"] + ["-t", "8", "--gpus", str(gpus), prefix, str(fq)],
                         capture_output=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-600:]
    assert b"the host parser takes over" in res.stderr
    assert _strip(res.stdout) == want
    rows = [l.split() for l in open(log).read().splitlines()]
    assert sum(int(r[1]) for r in rows) > 1000                  # the text path did the part in front of the first odd record (chunks behind it may have run and been dropped)
    if gpus == 8:
        assert {int(r[0]) for r in rows} == set(range(8))
    # only a blank last line: everything through the text path but the last chunk
    fq2 = tmp_path / "blank_tail.fq"
    fq2.write_bytes(open(os.path.join(LAMBDA, "reads_se.fq"), "rb").read() + b"\n")
    res = subprocess.run([str(d / "bin" / "salt")] + read_cases()["se_default"] + ["-t", "8", "--gpus", str(gpus), prefix, str(fq2)],
                         capture_output=True, env=env, timeout=300)
    assert res.returncode == 0 and _strip(res.stdout) == want, res.stderr[-600:]


def test_salt_pe_text_path_neither_hangs_nor_shifts_pairs_on_odd_files(stub_tree, tmp_path):
    """a mate file with a trailing blank line (line count not a multiple of four) or mate files of different lengths used to
    leave workers waiting for a block that never came.  Now the chunk where it shows goes to the host pipeline: a blank tail is read
    as the reference reads it, unequal files end with the error -- and nothing hangs (timeout)."""
    d, prefix = stub_tree
    want = open(os.path.join(LAMBDA, "expect_pe_default.sam"), "rb").read()
    f1, f2 = os.path.join(LAMBDA, "reads_pe_1.fq"), os.path.join(LAMBDA, "reads_pe_2.fq")
    env = dict(os.environ, SALT_STUB_PREFIX=prefix, SALT_CHUNK_BYTES="3000", LD_LIBRARY_PATH=str(d / "lib"))
    args = [str(d / "bin" / "salt")] + read_cases()["pe_default"] + ["-t", "16", "--gpus", "8", prefix]
    tail = tmp_path / "blank_tail.fq"
    tail.write_bytes(open(f2, "rb").read() + b"\n")
    for _ in range(3):                                         # the old hang needed a particular interleaving
        res = subprocess.run(args + [f1, str(tail)], capture_output=True, env=env, timeout=300)
        assert res.returncode == 0, res.stderr[-600:]
        assert _strip(res.stdout) == want
    lines = open(f2, "rb").read().split(b"\n")
    short = tmp_path / "short.fq"
    short.write_bytes(b"\n".join(lines[:-41]) + b"\n")          # ten records less
    for _ in range(3):
        res = subprocess.run(args + [f1, str(short)], capture_output=True, env=env, timeout=300)
        assert res.returncode == 1 and b"different numbers of reads" in res.stderr, res.stderr[-600:]
        got = _strip(res.stdout)
        assert want.startswith(got[:len(got) - len(got) % 1])   # whatever came out is a prefix of the right answer: no shifted pairs


def _bgzf(data, block=65280):
    """Blocked gzip as bgzip writes it: every block a gzip member with the 'BC' extra field (its compressed size - 1), an empty block last."""
    import struct
    import zlib
    out = bytearray()
    for o in list(range(0, len(data), block)) + [None]:
        raw = b"" if o is None else data[o:o + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(raw) + c.flush()
        bsize = 12 + 6 + len(body) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += body + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw))
    return bytes(out)


@pytest.mark.parametrize("gpus,block", [(1, 65280), (3, 700)])
def test_salt_reads_blocked_gzip_through_the_text_path(gpus, block, stub_tree, tmp_path):
    """BGZF input: the text path runs over the uncompressed byte range -- workers inflate the blocks their chunks touch
    (blocks of 700 bytes here: dozens per chunk, chunk ends inside blocks) -- and the SAM is the reference's.  A plain single-member .gz of
    the same reads goes through the host pipeline, as before."""
    import gzip
    d, prefix = stub_tree
    want = open(os.path.join(LAMBDA, "expect_se_default.sam"), "rb").read()
    raw = open(os.path.join(LAMBDA, "reads_se.fq"), "rb").read()
    bg = tmp_path / "reads.bgzf.fq.gz"
    bg.write_bytes(_bgzf(raw, block))
    import gzip as _g
    assert _g.decompress(bg.read_bytes()) == raw                  # a valid multi-member gzip file to everyone else
    log = str(tmp_path / "stub.log")
    env = dict(os.environ, SALT_STUB_PREFIX=prefix, SALT_STUB_LOG=log, SALT_CHUNK_BYTES="9000", LD_LIBRARY_PATH=str(d / "lib"))
    res = subprocess.run([str(d / "bin" / "salt")] + read_cases()["se_default"] + ["-t", "8", "--gpus", str(gpus), prefix, str(bg)],
                         capture_output=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-600:]
    assert b"text path:" in res.stderr and b"the host parser takes over" not in res.stderr
    assert _strip(res.stdout) == want
    rows = [l.split() for l in open(log).read().splitlines()]
    assert sum(int(r[1]) for r in rows) == 2000 and len(rows) > 10
    plain = tmp_path / "reads.plain.fq.gz"
    with gzip.open(plain, "wb") as f:
        f.write(raw)
    res = subprocess.run([str(d / "bin" / "salt")] + read_cases()["se_default"] + ["-t", "8", prefix, str(plain)], capture_output=True, env=env, timeout=300)
    assert res.returncode == 0 and b"host phases" in res.stderr and _strip(res.stdout) == want, res.stderr[-600:]
    # a damaged block is an error, not silence
    dmg = bytearray(bg.read_bytes()); dmg[len(dmg) // 2] ^= 0x55
    bad = tmp_path / "damaged.fq.gz"
    bad.write_bytes(bytes(dmg))
    res = subprocess.run([str(d / "bin" / "salt")] + read_cases()["se_default"] + ["-t", "8", prefix, str(bad)], capture_output=True, env=env, timeout=300)
    assert res.returncode != 0
