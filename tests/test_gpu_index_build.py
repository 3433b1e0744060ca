"""The device index builder (salt_amd/csrc/salt_sufsort.hip: prefix-doubling suffix sorter + the arrays salt-idx derives from a
suffix array) against the host suffix sorter and against the files the REAL reference's salt-idx wrote (tests/golden)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, LAMBDA, ROOT

pytestmark = pytest.mark.gpu


def _texts(bits):
    rng = np.random.default_rng(17 + bits)
    hi = 4 if bits == 2 else 5
    out = [np.zeros(1, np.uint8), np.zeros(2, np.uint8), np.zeros(31, np.uint8), np.zeros(33, np.uint8), np.zeros(100, np.uint8), np.zeros(5000, np.uint8),
           np.array([1, 0, 0, 0], np.uint8), np.array([0, 0, 0, 1], np.uint8), np.array([hi - 1] * 70, np.uint8)]
    for n in (1, 5, 20, 21, 22, 31, 32, 33, 63, 64, 65, 200, 1000, 4097, 100000):
        out.append(rng.integers(0, hi, size=n).astype(np.uint8))
    # runs of A at the end and in the middle (keys equal up to the zero padding), periodic texts (deep doubling), tandem repeats
    out.append(np.concatenate([rng.integers(0, hi, size=500), np.zeros(70)]).astype(np.uint8))
    out.append(np.concatenate([np.zeros(40), rng.integers(0, hi, size=50), np.zeros(40), [1], np.zeros(45)]).astype(np.uint8))
    out.append(np.tile(np.array([0, 1, 2, 3, 1], np.uint8), 3000))
    out.append(np.tile(rng.integers(0, hi, size=37).astype(np.uint8), 2000))
    unit = rng.integers(0, hi, size=300).astype(np.uint8)
    t = np.tile(unit, 400)
    m = rng.random(len(t)) < 0.01
    t[m] = (t[m] + 1) % hi
    out.append(t)
    if bits == 3:                                            # shaped like the local-pattern text: similar segments between '#'
        seg = rng.integers(0, 4, size=41).astype(np.uint8)
        parts = [np.array([4], np.uint8)]
        for _ in range(3000):
            s = seg.copy()
            s[20] = rng.integers(0, 4)
            if rng.random() < 0.2:
                seg = rng.integers(0, 4, size=41).astype(np.uint8)
            parts += [s, np.array([4], np.uint8)]
        out.append(np.concatenate(parts))
    return out


@pytest.mark.parametrize("bits", [2, 3])
def test_device_suffix_sorter_equals_the_host_one(bits):
    import salt_amd
    for t in _texts(bits):
        got = salt_amd.suffix_array(t, bits, gpu_device=0)
        want = salt_amd.suffix_array(t, bits)
        assert np.array_equal(got, want), (bits, len(t), np.nonzero(got != want)[0][:5])


def _same_index_files(prefix, d, with_lp=True):
    for sfx in (".R.seedLen", ".C.pac", ".C.ann", ".C.amb", ".C.bwt", ".C.sa", ".R.backward.bwt", ".R.backward.occ") + ((".lp",) if with_lp else ()):
        assert open(prefix + sfx, "rb").read() == open(os.path.join(d, "idx" + sfx), "rb").read(), sfx
    got, want = np.fromfile(prefix + ".R.backward.sa", dtype=np.uint32), np.fromfile(os.path.join(d, "idx.R.backward.sa"), dtype=np.uint32)
    assert len(got) == len(want) and int((got != want).sum()) <= 1          # the reference's one out-of-bounds entry
    got, want = np.fromfile(prefix + ".ref", dtype=np.uint32), np.fromfile(os.path.join(d, "idx.ref"), dtype=np.uint32)
    assert len(got) == len(want) and (got[:-1] == want[:-1]).all()
    assert hashlib.sha256(open(prefix + ".C.lkt", "rb").read()).hexdigest() == open(os.path.join(d, "idx.C.lkt.sha256")).read().strip()


@pytest.mark.parametrize("case", ["lambda"] + sorted(os.listdir(os.path.join(GOLDEN, "index_cases"))))
def test_device_builder_writes_the_reference_files(case, tmp_path):
    """`salt-idx --gpu` on the fixtures the real reference indexed: byte-identical files."""
    d = LAMBDA if case == "lambda" else os.path.join(GOLDEN, "index_cases", case)
    prefix = str(tmp_path / "idx")
    p = subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt-idx"), "--gpu", "-k", "19", os.path.join(d, "genome.fa"), os.path.join(d, "snps.txt"), prefix],
                       capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    _same_index_files(prefix, d)


@pytest.mark.parametrize("case", ["lambda", "dense"])
def test_device_builder_all_files(case, tmp_path):
    """`salt-idx --gpu --all-files`: the R text files and the forward R index (the device suffix sorter on the reversed text), by the
    digests of the real reference indexer's files."""
    d = LAMBDA if case == "lambda" else os.path.join(GOLDEN, "index_cases", case)
    prefix = str(tmp_path / "idx")
    p = subprocess.run([os.path.join(ROOT, "salt_amd", "bin", "salt-idx"), "--gpu", "--all-files", "-k", "19", os.path.join(d, "genome.fa"),
                        os.path.join(d, "snps.txt"), prefix], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    _same_index_files(prefix, d)
    for sfx, sha in (l.split() for l in open(os.path.join(d, "idx.unread.sha256"))):
        assert hashlib.sha256(open(prefix + sfx, "rb").read()).hexdigest() == sha, sfx


def test_device_and_host_builders_agree_on_a_repeat_rich_genome(tmp_path):
    """2 Mbp with a 300-base repeat family, a tandem block and 9 500 SNPs, from memory (salt_idx_build_mem): every file of the device
    build equals the host build's."""
    import salt_amd
    from salt_amd import workload
    g = workload.make_genome(2_000_000, seed=5)
    t = workload.make_tandem(30, 4000, 0.01, 1000, seed=6)
    g[500_000:500_000 + len(t)] = t
    pos, mask = workload.make_snps(g, 9_500, seed=7)
    contigs, groups = workload.as_builder_input(g, pos, mask, 3)
    a, b = str(tmp_path / "dev"), str(tmp_path / "host")
    salt_amd.idx_build_mem(contigs, groups, a, 21, gpu_device=0)
    salt_amd.idx_build_mem(contigs, groups, b, 21)
    for sfx in (".R.seedLen", ".C.pac", ".C.ann", ".C.amb", ".C.bwt", ".C.sa", ".C.lkt", ".lp", ".R.backward.bwt", ".R.backward.occ", ".R.backward.sa", ".ref"):
        assert open(a + sfx, "rb").read() == open(b + sfx, "rb").read(), sfx
