"""ctypes wrapper around oracle/libsalt_oracle.so -- TEST INFRASTRUCTURE (checker + CPU baseline).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (salt_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

HIT = np.dtype([("pos", "<u4"), ("n_diff", "u1"), ("is_gap", "u1"), ("strand", "<u2")])
RESULT = np.dtype([("pos", "<u4"), ("strand", "<i4"), ("n_diff", "u1"), ("is_gap", "u1"), ("mapq", "u1"), ("pad", "u1"),
                   ("b0", "<i4"), ("b1", "<i4"), ("seq_start", "<u4"), ("seq_end", "<u4"), ("n_hits", "<i4", (2,)),
                   ("hits", HIT, (2, 5)), ("cigar", "S128")])
assert RESULT.itemsize == 244
CTR_FIELDS = ["n_lkt", "n_occC", "n_occR", "n_occR_syms", "n_saC", "n_saR", "n_bwt2nt", "n_verify", "n_verify_words",
              "n_lv", "n_reads", "n_occC_seed", "n_occR_seed", "n_occR_syms_seed", "n_bases", "n_hits_out"]


class _Opt(ctypes.Structure):
    _fields_ = [("l_seed", ctypes.c_int32), ("l_overlap", ctypes.c_int32), ("max_seed", ctypes.c_uint32),
                ("max_locate", ctypes.c_uint32), ("max_hits", ctypes.c_int32), ("seed_only_ref", ctypes.c_int32),
                ("print_xa_cigar", ctypes.c_int32), ("print_nm_md", ctypes.c_int32), ("rg_id", ctypes.c_char_p)]


def build():
    subprocess.run(["make", "-C", _HERE, "port"], check=True, stdout=subprocess.DEVNULL)


def lib():
    so = os.path.join(_HERE, "libsalt_oracle.so")
    if not os.path.exists(so):
        build()
    L = ctypes.CDLL(so)
    L.so_index_load.restype = ctypes.c_void_p
    L.so_index_load.argtypes = [ctypes.c_char_p]
    L.so_index_free.argtypes = [ctypes.c_void_p]
    L.so_opt_default.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Opt)]
    L.so_align_se_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Opt), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.so_align_pe_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Opt), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return L


class Oracle:
    def __init__(self, prefix):
        self.L = lib()
        self.h = self.L.so_index_load(os.fsencode(prefix))
        if not self.h:
            raise RuntimeError("oracle: cannot load index %s" % prefix)

    def opt(self, l_overlap=-1, max_seed=50, max_locate=1000, seed_only_ref=0):
        o = _Opt()
        self.L.so_opt_default(self.h, ctypes.byref(o))
        if l_overlap > 0:
            o.l_overlap = l_overlap
        o.max_seed, o.max_locate, o.seed_only_ref = max_seed, max_locate, seed_only_ref
        return o

    def align(self, opt, seqs, offs, n_threads=1, counters=False):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint32)
        n = len(offs) - 1
        res = np.zeros(n, dtype=RESULT)
        ctr = np.zeros(len(CTR_FIELDS), dtype=np.uint64)
        self.L.so_align_se_batch(self.h, ctypes.byref(opt), n, seqs.ctypes.data, offs.ctypes.data, res.ctypes.data,
                                 n_threads, ctr.ctypes.data if counters else None)
        if counters:
            return res, dict(zip(CTR_FIELDS, [int(x) for x in ctr]))
        return res

    def align_pe(self, opt, seqs, offs, min_tlen=250, max_tlen=550, n_threads=1):
        """Mates interleaved (pair i = reads 2i, 2i+1) -> RESULT[2 * n_pairs] after pairing / rescue."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint32)
        n = len(offs) - 1
        res = np.zeros(n, dtype=RESULT)
        self.L.so_align_pe_batch(self.h, ctypes.byref(opt), min_tlen, max_tlen, n // 2, seqs.ctypes.data, offs.ctypes.data,
                                 res.ctypes.data, n_threads)
        return res

    def close(self):
        if self.h:
            self.L.so_index_free(self.h)
            self.h = None


def cigar_text(ops, n):
    return "".join("%d%s" % (int(x) >> 4, "MID"[int(x) & 3]) for x in ops[:n])


def compare(gpu_res, ora_res, pe=False):
    """Field-by-field comparison of salt_amd.RESULT_DTYPE rows with oracle RESULT rows.
    Returns the indices of the reads that differ.  pe: rows after pairing (soft clips and the CIGAR text of
    rescued mates are compared too)."""
    bad = np.zeros(len(ora_res), dtype=bool)
    if pe:
        mapped = ora_res["pos"] != 0xFFFFFFFF
        for f in ("seq_start", "seq_end"):
            bad |= mapped & (gpu_res[f].astype(np.uint32) != ora_res[f])
    bad |= gpu_res["pos"] != ora_res["pos"]
    bad |= gpu_res["strand"].astype(np.int32) != ora_res["strand"]
    for f in ("n_diff", "is_gap", "mapq", "b0", "b1"):
        bad |= gpu_res[f] != ora_res[f]
    bad |= (gpu_res["n_hits"].astype(np.int32) != ora_res["n_hits"]).any(axis=1)
    for s in range(2):
        for j in range(5):
            live = ora_res["n_hits"][:, s] > j
            for f in ("pos", "n_diff", "is_gap"):
                bad |= live & (gpu_res["hits"][f][:, s, j] != ora_res["hits"][f][:, s, j])
    mapped = np.nonzero((ora_res["pos"] != 0xFFFFFFFF) & ~bad)[0]
    # CIGARs: cheap check first (gap-free = one op), full text for the gapped ones
    for i in mapped:
        if pe:
            if cigar_text(gpu_res["cigar"][i], int(gpu_res["n_cigar"][i])) != ora_res["cigar"][i].decode():
                bad[i] = True
        elif ora_res["is_gap"][i] == 0:
            if gpu_res["n_cigar"][i] != 1:
                bad[i] = True
        elif cigar_text(gpu_res["cigar"][i], int(gpu_res["n_cigar"][i])) != ora_res["cigar"][i].decode():
            bad[i] = True
    return np.nonzero(bad)[0]
