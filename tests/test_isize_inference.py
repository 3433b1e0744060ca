"""N3 -- the insert-size window `-p -b 0` infers from the first batch.  The reference has no such function (alnpe.c:586-589 prints
"infer isize func haven't been implemented" and stops), so the definition is this build's, stated in oracle/salt_oracle.c and restated
independently in salt_amd/host/salt_host.cc: here both are held against a third, numpy statement of the same rule, on CPU."""
import ctypes
import os
import sys

import numpy as np

from conftest import LAMBDA, ROOT


def _rule(t):
    t = np.sort(np.asarray(t, dtype=np.int64))
    n = len(t)
    if n < 25:
        return None
    q1, q3 = int(t[n // 4]), int(t[3 * n // 4])
    iqr = q3 - q1
    lo, hi = max(q1 - 2 * iqr, 0), q3 + 2 * iqr
    v = t[(t >= lo) & (t <= hi)]
    m = len(v)
    mean = (int(v.sum()) + m // 2) // m
    var = int(((v - mean) ** 2).sum()) // m
    sd = int(np.floor(np.sqrt(var)))
    while (sd + 1) ** 2 <= var:
        sd += 1
    while sd * sd > var:
        sd -= 1
    if sd * sd < var:
        sd += 1
    a, b = (mean - 4 * sd if mean > 4 * sd else 1), mean + 4 * sd
    a = min(a, q1 - 3 * iqr if q1 > 3 * iqr else 1)
    b = max(b, q3 + 3 * iqr)
    return max(a, 1), b


def test_oracle_and_host_estimators_equal_the_rule(oracle_lib):
    import salt_amd
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    host = salt_amd.host_lib()
    host.salt_isize_infer.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32),
                                      ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    oracle_lib.so_isize_estimate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    oracle_lib.so_isize_templates.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    idx = salt_amd.Index.reload(os.path.join(LAMBDA, "idx"), rebuild_lkt=True)
    ora = oracle_py.Oracle(os.path.join(LAMBDA, "idx"))
    rng = np.random.default_rng(3)
    for trial in range(30):
        n_pairs = int(rng.integers(10, 400))
        L = 100
        offs = (np.arange(2 * n_pairs + 1, dtype=np.uint64) * L).astype(np.uint32)
        g = np.zeros(2 * n_pairs, dtype=salt_amd.RESULT_DTYPE)
        o = np.zeros(2 * n_pairs, dtype=oracle_py.RESULT)
        want_t = []
        for i in range(n_pairs):
            fpos = int(rng.integers(0, 40000)) + (48502 if rng.random() < 0.5 else 0)          # either contig of the lambda fixture
            ins = int(np.clip(rng.normal(500 if trial % 2 else 320, 40 + trial), 120, 2000))
            if rng.random() < 0.03:
                ins = int(rng.integers(3000, 90000))                                          # outliers
            rpos = fpos + ins - L
            kind = rng.random()
            a = dict(pos=fpos, strand=0, is_gap=0, nh=(0, 0)); b = dict(pos=rpos, strand=1, is_gap=0, nh=(0, 0))
            ok = True
            if kind < 0.05: b["pos"] = 0xFFFFFFFF; ok = False                                 # unmapped mate
            elif kind < 0.10: a["is_gap"] = 1; ok = False                                     # gapped
            elif kind < 0.15: b["nh"] = (1, 0); ok = False                                    # an alternative hit
            elif kind < 0.20: b["strand"] = 0; ok = False                                     # same strand
            elif kind < 0.25: a["pos"], b["pos"] = rpos + 50, fpos; ok = False                # reverse mate in front
            elif kind < 0.30: a["pos"] = 100; b["pos"] = 48502 + 300; ok = False              # different contigs
            if rng.random() < 0.5:
                a, b = b, a                                                                   # which file holds the forward mate
            for row, src in ((2 * i, a), (2 * i + 1, b)):
                for arr in (g, o):
                    arr["pos"][row] = src["pos"]; arr["strand"][row] = src["strand"]; arr["is_gap"][row] = src["is_gap"]
                    arr["n_hits"][row] = src["nh"]
            rp, fp_ = (a, b) if a["strand"] == 1 else (b, a)
            if ok and rp["pos"] >= fp_["pos"] and rp["pos"] + L - fp_["pos"] <= 100000 and (fp_["pos"] < 48502) == (rp["pos"] < 48502):
                want_t.append(rp["pos"] + L - fp_["pos"])
        want = _rule(want_t)
        mn, mx, used = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        rc = host.salt_isize_infer(idx._h, n_pairs, offs.ctypes.data, g.ctypes.data, ctypes.byref(mn), ctypes.byref(mx), ctypes.byref(used))
        t = np.zeros(n_pairs + 1, dtype=np.uint32)
        n_t = oracle_lib.so_isize_templates(ora.h, n_pairs, offs.ctypes.data, o.ctypes.data, t.ctypes.data)
        omn, omx = ctypes.c_uint32(), ctypes.c_uint32()
        orc = oracle_lib.so_isize_estimate(t.ctypes.data, n_t, ctypes.byref(omn), ctypes.byref(omx))
        assert used.value == n_t == len(want_t), (trial, used.value, n_t, len(want_t))
        if want is None:
            assert rc == -1 and orc == -1
        else:
            assert rc == 0 and orc == 0 and (mn.value, mx.value) == (omn.value, omx.value) == want, (trial, mn.value, mx.value, omn.value, omx.value, want)
    ora.close(); idx.destroy()
