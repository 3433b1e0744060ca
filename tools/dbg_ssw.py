#!/usr/bin/env python3
"""Debug helper: the first N ssw vectors through salt_gpu_diag_ssw; prints what differs.  usage: dbg_ssw.py N [first]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import salt_amd
lib = salt_amd.gpu_lib()
lib.salt_gpu_diag_ssw.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p] * 8
N = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
aware, refs, reads, want, roff, qoff = [], [], [], [], [0], [0]
for li, line in enumerate(open(os.path.join(ROOT, "tests", "golden", "ssw_vectors.txt"))):
    if li < first or li >= first + N:
        continue
    t = line.split()
    aware.append(int(t[1]))
    refs.append(np.array([int(c, 16) for c in t[2]], dtype=np.uint8))
    reads.append(np.frombuffer(t[3].encode(), dtype=np.uint8) - 48)
    want.append(([int(x) for x in t[4:10]], t[10]))
    roff.append(roff[-1] + len(refs[-1])); qoff.append(qoff[-1] + len(reads[-1]))
n = len(aware)
aw = np.array(aware, dtype=np.uint8)
rs, qs = np.concatenate(refs), np.concatenate(reads).astype(np.uint8)
ro, qo = np.array(roff, dtype=np.uint32), np.array(qoff, dtype=np.uint32)
out6 = np.zeros((n, 6), dtype=np.int32); cig = np.zeros((n, 64), dtype=np.uint16); ncig = np.zeros(n, dtype=np.uint16)
print("calling with", n, "cases", flush=True)
rc = lib.salt_gpu_diag_ssw(n, aw.ctypes.data, rs.ctypes.data, ro.ctypes.data, qs.ctypes.data, qo.ctypes.data, out6.ctypes.data, cig.ctypes.data, ncig.ctypes.data)
print("rc", rc, lib.salt_gpu_last_error() if rc else "", flush=True)
bad6 = badc = 0
for i in range(n):
    w6, wc = want[i]
    got = "".join("%d%s" % (int(x) >> 4, "MID"[int(x) & 3]) for x in cig[i, :int(ncig[i])]) or "-"
    if [int(x) for x in out6[i]] != w6:
        bad6 += 1
        if bad6 < 4: print("scores differ", first + i, list(out6[i]), w6)
    if got != wc:
        badc += 1
        if badc < 6: print("cigar differs", first + i, got, wc, w6)
print("cases", n, "score mismatches", bad6, "cigar mismatches", badc, flush=True)
