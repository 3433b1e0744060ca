"""ctypes bindings over the product's two C-ABI libraries (include/salt_gpu.h, include/salt_host.h).

The names follow the reference (weiquan/salt, Align_src/):
    Index.reload(prefix)          alnse_index_reload            indexio.c:23-49
    AlnOpt                        aln_opt_t / opt_init defaults aln.h:63-89, aln.c:28-56
    GpuAligner.alnse_core1(...)   alnse_core1 over one batch    alnse.c:1316-1352
    sam_header / samse            aln_samhead / aln_samse       sam.c:56-182

There is no CPU fallback here: every compute entry point goes through libsalt_gpu.so and raises
SaltError when the library or a HIP device is missing.
"""
import ctypes
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

MAX_HITS = 5
MAX_CIGAR_OPS = 64
CTR_NAMES = ["lkt", "occ_c", "occ_r", "sa_c", "sa_r", "verify", "verify_words", "lv", "reads", "bases", "loci",
             "t_load", "t_gather", "t_locate", "t_sort", "t_dedup", "t_verify", "t_scan", "t_gap", "t_tail", "heavy_reads", "x0", "x1", "x2", "x3",
             "lt_seeds", "lt_locate", "lt_sort", "lt_verify", "lt_out", "lt_samples", "max_heavy", "max_gapfin",
             "d_wlkt", "d_cocc_seed", "d_rocc_seed", "d_sa_seed", "d_text_seed", "d_sa_light", "d_verify_light", "d_out_light",
             "d_sa_heavy", "d_verify_heavy", "d_out_heavy", "d_ctx_rows", "d_ctx_rejected"]


class SaltError(RuntimeError):
    pass


HIT_DTYPE = np.dtype([("pos", "<u4"), ("n_diff", "u1"), ("is_gap", "u1"), ("strand", "<u2")])
RESULT_DTYPE = np.dtype([
    ("pos", "<u4"), ("strand", "u1"), ("n_diff", "u1"), ("is_gap", "u1"), ("mapq", "u1"),
    ("b0", "<i4"), ("b1", "<i4"), ("seq_start", "<u2"), ("seq_end", "<u2"),
    ("n_hits", "u1", (2,)), ("n_cigar", "u1"), ("skipped", "u1"),
    ("hits", HIT_DTYPE, (2, MAX_HITS)),
    ("hit_n_cigar", "u1", (MAX_HITS,)), ("pad", "u1", (3,)),
    ("cigar", "<u2", (MAX_CIGAR_OPS,)),
    ("hit_cigar", "<u2", (MAX_HITS, MAX_CIGAR_OPS)),
])
assert RESULT_DTYPE.itemsize == 880


class _HostIndex(ctypes.Structure):
    _fields_ = [
        ("c_primary", ctypes.c_uint32), ("c_L2", ctypes.c_uint32 * 5), ("c_seq_len", ctypes.c_uint32),
        ("c_bwt_size", ctypes.c_uint32), ("c_bwt", ctypes.c_void_p),
        ("c_sa_intv", ctypes.c_uint32), ("c_n_sa", ctypes.c_uint32), ("c_sa", ctypes.c_void_p),
        ("lkt_len", ctypes.c_uint32), ("lkt_n", ctypes.c_uint32), ("lkt", ctypes.c_void_p),
        ("r_text_len", ctypes.c_uint32), ("r_inv_sa0", ctypes.c_uint32), ("r_cum", ctypes.c_uint32 * 6),
        ("r_bwt_words", ctypes.c_uint32), ("r_bwt", ctypes.c_void_p),
        ("r_occ_words", ctypes.c_uint32), ("r_occ", ctypes.c_void_p),
        ("r_major_words", ctypes.c_uint32), ("r_major", ctypes.c_void_p),
        ("r_n_sa", ctypes.c_uint32), ("r_sa", ctypes.c_void_p),
        ("ref_len", ctypes.c_uint32), ("ref", ctypes.c_void_p),
        ("l_seed", ctypes.c_int32),
    ]


class _AlnOpt(ctypes.Structure):
    _fields_ = [("l_seed", ctypes.c_int32), ("l_overlap", ctypes.c_int32), ("max_seed", ctypes.c_uint32),
                ("max_locate", ctypes.c_uint32), ("max_hits", ctypes.c_int32), ("seed_only_ref", ctypes.c_int32),
                ("collect_counters", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class _PeOpt(ctypes.Structure):
    _fields_ = [("min_tlen", ctypes.c_uint32), ("max_tlen", ctypes.c_uint32)]


class _SamOpt(ctypes.Structure):
    _fields_ = [("print_xa_cigar", ctypes.c_int32), ("print_nm_md", ctypes.c_int32), ("rg_id", ctypes.c_char_p)]


_gpu = None
_host = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libsalt_host.so")
        if not os.path.exists(path):
            raise SaltError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        lib = ctypes.CDLL(path)
        lib.salt_index_load.restype = ctypes.c_void_p
        lib.salt_index_load.argtypes = [ctypes.c_char_p, ctypes.c_int]
        lib.salt_index_free.argtypes = [ctypes.c_void_p]
        lib.salt_index_host_view.restype = ctypes.POINTER(_HostIndex)
        lib.salt_index_host_view.argtypes = [ctypes.c_void_p]
        lib.salt_index_seed_len.argtypes = [ctypes.c_void_p]
        lib.salt_index_n_seqs.argtypes = [ctypes.c_void_p]
        lib.salt_host_last_error.restype = ctypes.c_char_p
        lib.salt_sam_header.argtypes = [ctypes.c_void_p, ctypes.POINTER(_SamOpt), ctypes.c_char_p, ctypes.c_size_t]
        lib.salt_index_pac.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        lib.salt_index_pac.restype = ctypes.c_void_p
        lib.salt_sam_pe.argtypes = [ctypes.c_void_p, ctypes.POINTER(_SamOpt), ctypes.POINTER(_PeOpt), ctypes.POINTER(ctypes.c_char_p),
                                    ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_char_p),
                                    ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        lib.salt_sam_se.argtypes = [ctypes.c_void_p, ctypes.POINTER(_SamOpt), ctypes.c_char_p, ctypes.c_void_p,
                                    ctypes.c_int32, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        lib.salt_lkt_build.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        lib.salt_cigar_text.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        _host = lib
    return _host


def gpu_lib():
    """The HIP library.  Missing library = hard error (never a silent fallback)."""
    global _gpu
    if _gpu is None:
        path = os.path.join(LIB_DIR, "libsalt_gpu.so")
        if not os.path.exists(path):
            raise SaltError("%s is missing: the HIP extension was not built" % path)
        lib = ctypes.CDLL(path)
        lib.salt_gpu_last_error.restype = ctypes.c_char_p
        lib.salt_gpu_result_size.restype = ctypes.c_uint32
        lib.salt_gpu_index_attach.argtypes = [ctypes.POINTER(_HostIndex), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.salt_gpu_index_detach.argtypes = [ctypes.c_void_p]
        lib.salt_gpu_index_image.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        lib.salt_gpu_index_attach_image.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.salt_gpu_ws_create.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
        lib.salt_gpu_ws_destroy.argtypes = [ctypes.c_void_p]
        lib.salt_gpu_align_se.argtypes = [ctypes.c_void_p, ctypes.POINTER(_AlnOpt), ctypes.c_uint32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p]
        lib.salt_gpu_align_se_resident.argtypes = [ctypes.c_void_p, ctypes.POINTER(_AlnOpt), ctypes.c_uint32, ctypes.c_uint32,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.salt_gpu_index_set_pac.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        lib.salt_gpu_align_pe.argtypes = [ctypes.c_void_p, ctypes.POINTER(_AlnOpt), ctypes.POINTER(_PeOpt), ctypes.c_uint32,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.salt_gpu_align_pe_resident.argtypes = [ctypes.c_void_p, ctypes.POINTER(_AlnOpt), ctypes.POINTER(_PeOpt), ctypes.c_uint32, ctypes.c_uint32,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.salt_gpu_ws_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        lib.salt_gpu_ws_pe_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        lib.salt_gpu_index_image_compact.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
        lib.salt_gpu_index_attach_compact.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.salt_gpu_index_image_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        lib.salt_gpu_ws_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.salt_gpu_ws_queue_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        lib.salt_gpu_ws_heavy_reads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        lib.salt_gpu_ws_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint32)]
        lib.salt_gpu_diag_verify.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        assert lib.salt_gpu_result_size() == RESULT_DTYPE.itemsize
        _gpu = lib
    return _gpu


def _gpu_check(rc):
    if rc != 0:
        raise SaltError("salt_gpu error %d: %s" % (rc, gpu_lib().salt_gpu_last_error().decode()))


@dataclass
class AlnOpt:
    """aln_opt_t as the SE path reads it; defaults of opt_init()/aln_opt_init() (aln.c:28-56, aln.h:112-145)."""
    l_seed: int = 25
    l_overlap: int = -1
    max_seed: int = 50
    max_locate: int = 1000
    max_hits: int = 5
    seed_only_ref: int = 0
    print_xa_cigar: int = 0
    print_nm_md: int = 0
    rg_id: str = None
    collect_counters: int = 0
    paired: int = 0             # -p
    min_tlen: int = 250         # -a (aln.c:43)
    max_tlen: int = 550         # -b (aln.c:44)

    def _pe(self):
        return _PeOpt(self.min_tlen, self.max_tlen)

    def _c(self):
        ov = self.l_overlap if self.l_overlap > 0 else self.l_seed          # aln.c:223
        return _AlnOpt(self.l_seed, ov, self.max_seed, self.max_locate, self.max_hits, self.seed_only_ref,
                       self.collect_counters, 0)

    def _sam(self):
        return _SamOpt(self.print_xa_cigar, self.print_nm_md, self.rg_id.encode() if self.rg_id else None)

    @classmethod
    def from_argv(cls, argv, l_seed):
        """Parses salt's option letters (optstring aln.c:102); ignored flags stay ignored."""
        import getopt
        opts, rest = getopt.getopt(argv, "t:n:hpa:b:g:em:s:l:cdr:vM:O:E:X:")
        o = cls(l_seed=l_seed)
        for k, v in opts:
            if k == "-s":
                o.max_seed = int(v)
            elif k == "-m":
                o.max_locate = int(v)
            elif k == "-r":
                o.l_overlap = int(v)
            elif k == "-v":
                o.seed_only_ref = 1
            elif k == "-c":
                o.print_xa_cigar = 1
            elif k == "-d":
                o.print_nm_md = 1
            elif k == "-g":
                o.rg_id = v
            elif k == "-p":
                o.paired = 1
            elif k == "-a":
                o.min_tlen = int(v)
            elif k == "-b":
                o.max_tlen = int(v)
        return o, rest


class Index:
    """index_t: the arrays of the index files, on the host."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def reload(cls, prefix, rebuild_lkt=True):
        h = host_lib().salt_index_load(os.fsencode(prefix), 1 if rebuild_lkt else 0)
        if not h:
            raise SaltError(host_lib().salt_host_last_error().decode())
        return cls(h)

    @property
    def view(self):
        return host_lib().salt_index_host_view(self._h)

    @property
    def l_seed(self):
        return host_lib().salt_index_seed_len(self._h)

    def destroy(self):
        if self._h:
            host_lib().salt_index_free(self._h)
            self._h = None

    def sam_header(self, opt):
        buf = ctypes.create_string_buffer(1 << 20)
        so = opt._sam()
        n = host_lib().salt_sam_header(self._h, ctypes.byref(so), buf, len(buf))
        if n < 0:
            raise SaltError("SAM header too large")
        return buf.raw[:n]

    def pac(self):
        n = ctypes.c_uint64()
        p = host_lib().salt_index_pac(self._h, ctypes.byref(n))
        return p, n.value

    def sampe(self, opt, names, seqs, quals, result_rows):
        """Both SAM records of a pair (bytes, with the reference's blank line after each)."""
        buf = ctypes.create_string_buffer(16384 + 16 * (len(seqs[0]) + len(seqs[1])))
        so, pe = opt._sam(), opt._pe()
        sq = [np.ascontiguousarray(x, dtype=np.uint8) for x in seqs]
        rows = np.ascontiguousarray(result_rows)
        nm = (ctypes.c_char_p * 2)(names[0], names[1])
        ql = (ctypes.c_char_p * 2)(quals[0], quals[1])
        sp = (ctypes.c_void_p * 2)(sq[0].ctypes.data, sq[1].ctypes.data)
        ls = (ctypes.c_int32 * 2)(len(sq[0]), len(sq[1]))
        n = host_lib().salt_sam_pe(self._h, ctypes.byref(so), ctypes.byref(pe), nm, sp, ls, ql, rows.ctypes.data, buf, len(buf))
        if n < 0:
            raise SaltError("SAM record too large")
        return buf.raw[:n]

    def samse(self, opt, name, seq, qual, result_row):
        """One SAM record (bytes, no newline) for a row of RESULT_DTYPE."""
        buf = ctypes.create_string_buffer(4096 + 8 * len(seq))
        so = opt._sam()
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        row = np.ascontiguousarray(result_row)
        n = host_lib().salt_sam_se(self._h, ctypes.byref(so), name, seq.ctypes.data, len(seq), qual,
                                   row.ctypes.data, buf, len(buf))
        if n < 0:
            raise SaltError("SAM record too large")
        return buf.raw[:n]


# ---- index builder (salt-idx; include/salt_host.h, include/salt_gpu.h "index construction on the device") ----
_BUILD_C = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32)
_BUILD_R = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                            ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p)
_LAST_ERR = ctypes.CFUNCTYPE(ctypes.c_char_p)


class _IdxBackend(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("build_c", _BUILD_C), ("build_r", _BUILD_R), ("last_error", _LAST_ERR)]


class _IdxContig(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("comment", ctypes.c_char_p), ("seq", ctypes.c_void_p), ("len", ctypes.c_uint64)]


class _IdxSnps(ctypes.Structure):
    _fields_ = [("chr", ctypes.c_char_p), ("pos", ctypes.c_void_p), ("alleles", ctypes.c_void_p), ("ref", ctypes.c_void_p), ("n", ctypes.c_uint32)]


IDX_NO_LP = 1


def _idx_backend(gpu_device):
    """None -> the host suffix sorter; a device number -> the device builder of libsalt_gpu.so."""
    if gpu_device is None:
        return None
    g = gpu_lib()
    return _IdxBackend(int(gpu_device), ctypes.cast(g.salt_gpu_idx_build_c, _BUILD_C), ctypes.cast(g.salt_gpu_idx_build_r, _BUILD_R),
                       ctypes.cast(g.salt_gpu_idx_last_error, _LAST_ERR))


def idx_build(fasta, snps, prefix, l_seed, gpu_device=None, flags=0):
    """salt-idx: FASTA + SNP file -> index files (index_main, Index_src/index1.c:46-185)."""
    lib = host_lib()
    be = _idx_backend(gpu_device)
    lib.salt_idx_build_ex.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.salt_idx_last_error.restype = ctypes.c_char_p
    if lib.salt_idx_build_ex(os.fsencode(fasta), os.fsencode(snps), os.fsencode(prefix), int(l_seed), ctypes.byref(be) if be else None, int(flags)) != 0:
        raise SaltError("index build failed: %s" % lib.salt_idx_last_error().decode())


def idx_build_mem(contigs, snp_groups, prefix, l_seed, gpu_device=None, flags=0):
    """The same from memory.  contigs: [(name, uint8 array of base LETTERS)]; snp_groups: [(chr name, uint32 0-based positions,
    uint8 allele masks, uint8 reference codes)] in contig order (group i belongs to contig i)."""
    lib = host_lib()
    be = _idx_backend(gpu_device)
    keep = []
    cs = (_IdxContig * len(contigs))()
    for i, (name, seq) in enumerate(contigs):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        keep.append(seq)
        cs[i] = _IdxContig(name.encode(), None, seq.ctypes.data, len(seq))
    gs = (_IdxSnps * max(len(snp_groups), 1))()
    for i, (name, pos, al, ref) in enumerate(snp_groups):
        pos, al, ref = np.ascontiguousarray(pos, dtype=np.uint32), np.ascontiguousarray(al, dtype=np.uint8), np.ascontiguousarray(ref, dtype=np.uint8)
        keep += [pos, al, ref]
        gs[i] = _IdxSnps(name.encode(), pos.ctypes.data, al.ctypes.data, ref.ctypes.data, len(pos))
    lib.salt_idx_build_mem.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.salt_idx_last_error.restype = ctypes.c_char_p
    rc = lib.salt_idx_build_mem(cs, len(contigs), gs, len(snp_groups), os.fsencode(prefix), int(l_seed), ctypes.byref(be) if be else None, int(flags))
    if rc != 0:
        raise SaltError("index build failed: %s" % lib.salt_idx_last_error().decode())


def suffix_array(text, bits, gpu_device=None):
    """Full suffix array (empty suffix first) of a uint8 symbol array: host SA-IS, or the device sorter."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    sa = np.zeros(len(text) + 1, dtype=np.uint32)
    if gpu_device is None:
        lib = host_lib()
        lib.salt_idx_suffix_array_cpu.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
        if lib.salt_idx_suffix_array_cpu(text.ctypes.data, len(text), bits, sa.ctypes.data) != 0:
            raise SaltError("host suffix sorter failed")
    else:
        g = gpu_lib()
        g.salt_gpu_suffix_array.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
        g.salt_gpu_idx_last_error.restype = ctypes.c_char_p
        if g.salt_gpu_suffix_array(int(gpu_device), text.ctypes.data, len(text), bits, sa.ctypes.data) != 0:
            raise SaltError("device suffix sorter: %s" % g.salt_gpu_idx_last_error().decode())
    return sa


KERNELS = ("k_pack", "k_seed", "k_light", "k_heavy", "k_gap", "k_gapfin", "k_cigar", "k_pair", "k_sw", "k_pe_final")


class GpuAligner:
    """Device copy of the index + one batch workspace on one GPU."""

    def __init__(self, index=None, device=0, max_reads=100000, max_bases=None, image=None, compact=None):
        """index: a host Index to re-pack and upload; or image=(device_ptr, bytes): an already packed full
        device image that stays owned by the caller; or compact=(device_ptr, bytes): the compact part of an image
        (e.g. received by an RCCL broadcast), from which this device builds its own full image."""
        lib = gpu_lib()
        self._ix = ctypes.c_void_p()
        if compact is not None:
            _gpu_check(lib.salt_gpu_index_attach_compact(compact[0], compact[1], device, ctypes.byref(self._ix)))
        elif image is not None:
            _gpu_check(lib.salt_gpu_index_attach_image(image[0], image[1], device, ctypes.byref(self._ix)))
        else:
            _gpu_check(lib.salt_gpu_index_attach(index.view, device, ctypes.byref(self._ix)))
        self._ws = ctypes.c_void_p()
        self.max_reads = max_reads
        self.max_bases = max_bases if max_bases is not None else max_reads * 160
        try:
            _gpu_check(lib.salt_gpu_ws_create(self._ix, max_reads, self.max_bases, ctypes.byref(self._ws)))
        except SaltError:
            lib.salt_gpu_index_detach(self._ix)
            raise
        self.device = device

    def image(self):
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        _gpu_check(gpu_lib().salt_gpu_index_image(self._ix, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def pe_counts(self):
        """Counters of the last paired batch: [0] rescue requests, [2] CIGAR items, [4] overflowed rescues."""
        out = (ctypes.c_uint32 * 8)()
        _gpu_check(gpu_lib().salt_gpu_ws_pe_counts(self._ws, out))
        return list(out)

    def image_compact(self):
        """(device pointer, bytes) of the part of the image a multi-GPU driver broadcasts (everything but the W-mer table)."""
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        _gpu_check(gpu_lib().salt_gpu_index_image_compact(self._ix, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def image_copy(self, dst_ptr, dst_bytes):
        _gpu_check(gpu_lib().salt_gpu_index_image_copy(self._ix, dst_ptr, dst_bytes))

    def timing(self, enable=True):
        _gpu_check(gpu_lib().salt_gpu_ws_timing(self._ws, 1 if enable else 0))

    def kernel_ms(self):
        """({kernel: ms summed since the last read}, calls)."""
        ms = (ctypes.c_double * len(KERNELS))()
        n = ctypes.c_uint32()
        _gpu_check(gpu_lib().salt_gpu_ws_kernel_ms(self._ws, ms, ctypes.byref(n)))
        return dict(zip(KERNELS, list(ms))), n.value

    def queue_counts(self):
        out = (ctypes.c_uint32 * 8)()
        _gpu_check(gpu_lib().salt_gpu_ws_queue_counts(self._ws, out))
        return list(out)

    def heavy_reads(self):
        """Indices (in the last batch) of the reads k_light queued for k_heavy."""
        n = ctypes.c_uint32()
        ids = np.zeros(self.max_reads, dtype=np.uint32)
        _gpu_check(gpu_lib().salt_gpu_ws_heavy_reads(self._ws, ids.ctypes.data, len(ids), ctypes.byref(n)))
        return ids[:n.value]

    def alnse_core1(self, opt, seqs, offs):
        """seqs: uint8 codes 0..4 concatenated; offs: uint32 n+1 offsets.  Returns RESULT_DTYPE[n]."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint32)
        n = len(offs) - 1
        res = np.zeros(n, dtype=RESULT_DTYPE)
        co = opt._c()
        done = 0
        while done < n:                                   # batches of the workspace size (N_SEQS, aln.h:27)
            m = min(self.max_reads, n - done)
            while m > 1 and int(offs[done + m]) - int(offs[done]) > self.max_bases:
                m //= 2
            o = (offs[done:done + m + 1] - offs[done]).astype(np.uint32)
            s = seqs[int(offs[done]):int(offs[done + m])]
            _gpu_check(gpu_lib().salt_gpu_align_se(self._ws, ctypes.byref(co), m, s.ctypes.data, o.ctypes.data,
                                                   res[done:].ctypes.data))
            done += m
        return res

    def alnpe_core1(self, opt, index, seqs, offs):
        """Paired end: mates interleaved (pair i = reads 2i, 2i+1).  Returns RESULT_DTYPE[2 * n_pairs]."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint32)
        n = len(offs) - 1
        if n % 2:
            raise SaltError("paired end needs an even number of reads")
        lib = gpu_lib()
        if not getattr(self, "_pac_set", False):
            pac, l_pac = index.pac()
            _gpu_check(lib.salt_gpu_index_set_pac(self._ix, pac, l_pac))
            self._pac_set = True
        res = np.zeros(n, dtype=RESULT_DTYPE)
        co, pe = opt._c(), opt._pe()
        done = 0
        cap = self.max_reads & ~1
        while done < n:
            m = min(cap, n - done)
            while m > 2 and int(offs[done + m]) - int(offs[done]) > self.max_bases:
                m = (m // 2) & ~1
            o = (offs[done:done + m + 1] - offs[done]).astype(np.uint32)
            s = seqs[int(offs[done]):int(offs[done + m])]
            _gpu_check(lib.salt_gpu_align_pe(self._ws, ctypes.byref(co), ctypes.byref(pe), m // 2, s.ctypes.data,
                                             o.ctypes.data, res[done:].ctypes.data))
            done += m
        return res

    def set_pac(self, index):
        """Uploads the 2-bit genome the singleton rescue aligns against (once per device index; forks made afterwards share it)."""
        if not getattr(self, "_pac_set", False):
            pac, l_pac = index.pac()
            _gpu_check(gpu_lib().salt_gpu_index_set_pac(self._ix, pac, l_pac))
            self._pac_set = True

    def align_pe_resident(self, opt, index, n_pairs, max_read_len, d_seqs, d_offs, d_results, stream=0):
        lib = gpu_lib()
        self.set_pac(index)
        co, pe = opt._c(), opt._pe()
        _gpu_check(lib.salt_gpu_align_pe_resident(self._ws, ctypes.byref(co), ctypes.byref(pe), n_pairs, max_read_len, d_seqs, d_offs,
                                                  d_results, stream))

    def align_resident(self, opt, n_reads, max_read_len, d_seqs, d_offs, d_results, stream=0):
        co = opt._c()
        _gpu_check(gpu_lib().salt_gpu_align_se_resident(self._ws, ctypes.byref(co), n_reads, max_read_len, d_seqs, d_offs,
                                                        d_results, stream))

    def counters(self):
        out = (ctypes.c_uint64 * len(CTR_NAMES))()
        _gpu_check(gpu_lib().salt_gpu_ws_counters(self._ws, out))
        return dict(zip(CTR_NAMES, [int(x) for x in out]))

    def fork(self, max_reads=None, max_bases=None):
        """Another workspace on the SAME device index (one per host thread / stream, as `salt` runs several per GPU).
        Close the forks before the aligner they came from."""
        other = GpuAligner.__new__(GpuAligner)
        other._ix, other._owns_ix = self._ix, False
        other._ws = ctypes.c_void_p()
        other.max_reads = max_reads or self.max_reads
        other.max_bases = max_bases or self.max_bases
        other.device = self.device
        other._pac_set = getattr(self, "_pac_set", False)      # the 2-bit genome belongs to the device index, not to the workspace
        _gpu_check(gpu_lib().salt_gpu_ws_create(self._ix, other.max_reads, other.max_bases, ctypes.byref(other._ws)))
        return other

    def close(self):
        lib = gpu_lib()
        if self._ws:
            lib.salt_gpu_ws_destroy(self._ws)
            self._ws = None
        if self._ix and getattr(self, "_owns_ix", True):
            lib.salt_gpu_index_detach(self._ix)
        self._ix = None


_NT4 = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _NT4[ord(_c)] = _i
    _NT4[ord(_c.lower())] = _i


def read_fastq(path):
    """4-line FASTQ (optionally .gz) -> names, seqs (codes), offs, quals.  Names end at the first
    blank and lose a trailing /1 or /2 like trim_readno (query.c:139-143)."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    names, quals, chunks, offs = [], [], [], [0]
    with op(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            if not h.startswith(b"@"):
                continue
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            nm = h[1:].split()[0] if len(h) > 1 else b""
            if len(nm) > 2 and nm[-2:-1] == b"/" and nm[-1:].isdigit():
                nm = nm[:-2]
            names.append(nm)
            quals.append(q)
            chunks.append(_NT4[np.frombuffer(s, dtype=np.uint8)])
            offs.append(offs[-1] + len(s))
    seqs = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    return names, seqs, np.array(offs, dtype=np.uint32), quals


def interleave_pairs(r1, r2):
    """Two read_fastq() results -> one with mates interleaved as query_read_multiPairedSeqs (query.c:252-268)."""
    n1, s1, o1, q1 = r1
    n2, s2, o2, q2 = r2
    if len(n1) != len(n2):
        raise SaltError("mate files differ in length")
    names, quals, chunks, offs = [], [], [], [0]
    for i in range(len(n1)):
        for nm, sq, of, ql in ((n1, s1, o1, q1), (n2, s2, o2, q2)):
            names.append(nm[i]); quals.append(ql[i]); chunks.append(sq[of[i]:of[i + 1]])
            offs.append(offs[-1] + int(of[i + 1] - of[i]))
    seqs = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    return names, seqs, np.array(offs, dtype=np.uint32), quals


def sam_text_pe(index, opt, names, seqs, offs, quals, results, header=True):
    """The SAM stream `salt -p` prints (without @PG)."""
    out = [index.sam_header(opt)] if header else []
    for i in range(0, len(names), 2):
        out.append(index.sampe(opt, names[i:i + 2], [seqs[offs[i]:offs[i + 1]], seqs[offs[i + 1]:offs[i + 2]]],
                               quals[i:i + 2], results[i:i + 2]))
    return b"".join(out)


def sam_text(index, opt, names, seqs, offs, quals, results, header=True):
    """The SAM stream `salt` prints (without @PG): header, then one line per read in input order."""
    out = []
    if header:
        out.append(index.sam_header(opt))
    for i in range(len(names)):
        rec = index.samse(opt, names[i], seqs[offs[i]:offs[i + 1]], quals[i], results[i:i + 1])
        out.append(rec + b"\n")
    return b"".join(out)
