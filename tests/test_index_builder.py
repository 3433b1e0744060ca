"""The product's index builder (salt_amd/host/salt_idx.cc, row N1) against the files the real reference's
salt-idx wrote for the same FASTA + SNP file (tests/golden/lambda/idx.*).  Bar: byte-identical files.
Known, documented exception: the unused high nibbles of the last .ref word (the reference leaves
uninitialised realloc memory there, Index_src/mixRef.c:131-142)."""
import ctypes
import hashlib
import os

import numpy as np

from conftest import LAMBDA


def test_builder_writes_reference_identical_files(tmp_path):
    import salt_amd
    lib = salt_amd.host_lib()
    lib.salt_idx_build.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    lib.salt_idx_last_error.restype = ctypes.c_char_p
    prefix = str(tmp_path / "idx")
    rc = lib.salt_idx_build(os.path.join(LAMBDA, "genome.fa").encode(), os.path.join(LAMBDA, "snps.txt").encode(),
                            prefix.encode(), 19)
    assert rc == 0, lib.salt_idx_last_error()
    for sfx in (".R.seedLen", ".C.pac", ".C.ann", ".C.amb", ".C.bwt", ".C.sa", ".lp", ".R.backward.bwt",
                ".R.backward.occ", ".R.backward.sa"):
        got = open(prefix + sfx, "rb").read()
        want = open(os.path.join(LAMBDA, "idx" + sfx), "rb").read()
        assert got == want, sfx
    got = np.fromfile(prefix + ".ref", dtype=np.uint32)
    want = np.fromfile(os.path.join(LAMBDA, "idx.ref"), dtype=np.uint32)
    assert len(got) == len(want) and got[0] == want[0]
    l = int(got[0])
    assert (got[1:-1] == want[1:-1]).all()
    live = (1 << (4 * (l % 8))) - 1 if l % 8 else 0xFFFFFFFF
    assert (int(got[-1]) & live) == (int(want[-1]) & live)
    sha = hashlib.sha256(open(prefix + ".C.lkt", "rb").read()).hexdigest()
    assert sha == open(os.path.join(LAMBDA, "idx.C.lkt.sha256")).read().strip()


import pytest                                             # noqa: E402
from conftest import GOLDEN                               # noqa: E402
INDEX_CASES = sorted(os.listdir(os.path.join(GOLDEN, "index_cases")))


@pytest.mark.parametrize("case", INDEX_CASES)
def test_builder_on_stress_cases(case, tmp_path):
    """tests/golden/make_index_fixture.py: contigs with and without SNPs (the reference matches SNP groups to contigs by ORDER),
    multi-allelic and contig-end SNPs, a SNP every ~7 bases.  Every file byte-identical to the real reference's, except the two
    documented spots where the reference writes memory it does not own: the dead high nibbles of the last .ref word and ONE entry of
    .R.backward.sa (`sharp2Ri_array[n]`, rbwt.c:377,461) that it reads one past its malloc."""
    import salt_amd
    lib = salt_amd.host_lib()
    lib.salt_idx_build.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    lib.salt_idx_last_error.restype = ctypes.c_char_p
    d = os.path.join(GOLDEN, "index_cases", case)
    prefix = str(tmp_path / "idx")
    rc = lib.salt_idx_build(os.path.join(d, "genome.fa").encode(), os.path.join(d, "snps.txt").encode(), prefix.encode(), 19)
    assert rc == 0, lib.salt_idx_last_error()
    for sfx in (".R.seedLen", ".C.pac", ".C.ann", ".C.amb", ".C.bwt", ".C.sa", ".lp", ".R.backward.bwt", ".R.backward.occ"):
        assert open(prefix + sfx, "rb").read() == open(os.path.join(d, "idx" + sfx), "rb").read(), sfx
    got, want = np.fromfile(prefix + ".R.backward.sa", dtype=np.uint32), np.fromfile(os.path.join(d, "idx.R.backward.sa"), dtype=np.uint32)
    assert len(got) == len(want) and int((got != want).sum()) <= 1, int((got != want).sum())
    got, want = np.fromfile(prefix + ".ref", dtype=np.uint32), np.fromfile(os.path.join(d, "idx.ref"), dtype=np.uint32)
    assert len(got) == len(want) and got[0] == want[0] and (got[1:-1] == want[1:-1]).all()
    l = int(got[0])
    live = (1 << (4 * (l % 8))) - 1 if l % 8 else 0xFFFFFFFF
    assert (int(got[-1]) & live) == (int(want[-1]) & live)
    assert hashlib.sha256(open(prefix + ".C.lkt", "rb").read()).hexdigest() == open(os.path.join(d, "idx.C.lkt.sha256")).read().strip()


UNREAD_DIRS = [LAMBDA] + [os.path.join(GOLDEN, "index_cases", c) for c in INDEX_CASES]


@pytest.mark.parametrize("d", UNREAD_DIRS, ids=[os.path.basename(x) for x in UNREAD_DIRS])
def test_builder_all_files_equal_the_reference_indexers(d, tmp_path):
    """SALT_IDX_ALL_FILES (salt-idx --all-files): the files the reference indexer writes and `salt` never reads -- the R text as 4-bit
    pac and reversed pac, its .ann / .amb, and the FORWARD R index (BWT, Occ, '#' table of the reversed text) -- byte for byte, by the
    digests of the real reference indexer's files (tests/golden/make_unread_file_digests.py); the files `salt` reads are unchanged by it."""
    import salt_amd
    lib = salt_amd.host_lib()
    lib.salt_idx_build_ex.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.salt_idx_build.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    lib.salt_idx_last_error.restype = ctypes.c_char_p
    prefix, plain = str(tmp_path / "idx"), str(tmp_path / "plain")
    fa, snp = os.path.join(d, "genome.fa").encode(), os.path.join(d, "snps.txt").encode()
    assert lib.salt_idx_build_ex(fa, snp, prefix.encode(), 19, None, 2) == 0, lib.salt_idx_last_error()
    assert lib.salt_idx_build(fa, snp, plain.encode(), 19) == 0, lib.salt_idx_last_error()
    want = dict(l.split() for l in open(os.path.join(d, "idx.unread.sha256")))
    assert len(want) == 7
    for sfx, sha in want.items():
        assert hashlib.sha256(open(prefix + sfx, "rb").read()).hexdigest() == sha, sfx
        assert not os.path.exists(plain + sfx)
    for sfx in (".R.seedLen", ".C.pac", ".C.bwt", ".C.sa", ".lp", ".R.backward.bwt", ".R.backward.occ", ".R.backward.sa", ".ref"):
        assert open(prefix + sfx, "rb").read() == open(plain + sfx, "rb").read(), sfx


def test_host_loader_and_abi_symbols():
    """The C-ABI libraries load without a GPU and export every symbol include/*.h declares."""
    import re
    import salt_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for hdr, lib in (("salt_gpu.h", salt_amd.gpu_lib()), ("salt_host.h", salt_amd.host_lib())):
        text = open(os.path.join(root, "include", hdr)).read()
        names = set(re.findall(r"\b(salt_[a-z0-9_]+)\s*\(", text))
        assert names
        for n in names:
            assert hasattr(lib, n), "%s declared in %s but not exported" % (n, hdr)
    idx = salt_amd.Index.reload(os.path.join(LAMBDA, "idx"))
    assert idx.l_seed == 19
    v = idx.view.contents
    assert v.c_seq_len == 97004 and v.ref_len == 97004 and v.lkt_len == 12
    idx.destroy()


def test_gpu_entry_points_fail_loudly_without_a_device():
    import pytest
    import torch
    import salt_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    idx = salt_amd.Index.reload(os.path.join(LAMBDA, "idx"))
    with pytest.raises(salt_amd.SaltError):
        salt_amd.GpuAligner(idx)
    idx.destroy()
