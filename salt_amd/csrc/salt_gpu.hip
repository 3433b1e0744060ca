// salt_amd/csrc/salt_gpu.hip -- the C ABI of include/salt_gpu.h: device-index attach (re-packing
// the file-format arrays into the HBM layout of salt_device.h), per-batch workspaces, and the
// align entry points.  gfx950 only; fails loudly when no HIP device is usable.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <cstddef>
#include <cctype>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include "salt_kernels.h"
#include <rccl/rccl.h>

using namespace salt;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(SALT_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct salt_gpu_index {
    int device = 0;
    uint8_t *image = nullptr;      // device
    uint64_t bytes = 0;
    bool owns = true;
    ImageHeader hdr;               // host copy
    IndexView view;
    uint8_t *d_pac = nullptr; uint64_t l_pac = 0;      // 2-bit genome for the PE singleton rescue (not part of the image)
    uint4 *d_rctx = nullptr;                            // context records of the R rows (paired end; not part of the image)
    int64_t *d_c_off = nullptr; uint32_t *d_c_name_off = nullptr; char *d_c_names = nullptr; int32_t n_contigs = 0;     // contig table for the SAM kernels
};

struct salt_gpu_ws {
    salt_gpu_index *ix = nullptr;
    uint32_t max_reads = 0; uint64_t max_bases = 0;
    uint8_t *d_seqs = nullptr; uint32_t *d_offs = nullptr; salt_result_t *d_results = nullptr;
    uint4 *d_sai_c = nullptr, *d_sai_r = nullptr; uint64_t sai_cap = 0;
    uint4 *d_wq = nullptr; uint32_t *d_wq_cnt = nullptr; uint32_t walk_blocks = 2048;      // k_seed's walk queues (sized with the seed arrays), k_seed_walk's grid
    uint32_t *d_pm = nullptr, *d_tb = nullptr; uint64_t pm_cap = 0, tb_cap = 0;     // k_pack's records (words)
    uint8_t *d_heads = nullptr, *h_heads = nullptr;          // first 128 bytes of every result row: dense device copy + pinned host staging
    unsigned long long *d_ctr = nullptr;
    uint32_t *d_queue = nullptr, *d_qctl = nullptr;   // reads k_light hands to k_heavy; {count, head}
    void *d_lvtab = nullptr;                          // one LV traceback table per persistent k_heavy block
    uint8_t *d_gap = nullptr; uint32_t gcap = 0; GapBufs gap{};             // deferred gapped passes
    // paired end (allocated on first use)
    uint8_t *d_pe_scr = nullptr;                       // per persistent block: PE_LOCI_CAP loci + distances
    PePair *d_pairs = nullptr; PeSwReq *d_req = nullptr; PeSwRes *d_swres = nullptr; uint32_t *d_pctl = nullptr;
    uint8_t *d_sw_scr = nullptr; uint64_t sw_scr_bytes = 0; uint32_t sw_blocks = 0; uint32_t pe_pairs_cap = 0;
    uint32_t *d_pcq = nullptr;                         // k_cigar items of the gapped, not rescued mates
    // FASTQ text in / SAM text out (allocated on first use, grown on demand)
    uint8_t *d_raw = nullptr; uint64_t raw_cap = 0; uint32_t *d_tile = nullptr; uint64_t tile_cap = 0; uint32_t *d_lines = nullptr; uint64_t lines_cap = 0;
    FqRec *d_rec = nullptr; uint32_t *d_tctl = nullptr, *d_samoff = nullptr; void *d_scan = nullptr; size_t scan_bytes = 0;
    char *d_sam = nullptr, *h_sam = nullptr; uint64_t sam_cap = 0; char *d_rg = nullptr; std::string rg;
    char *d_samslot = nullptr; SamSeg *d_samseg = nullptr;      // [max_reads]: the records' formatted heads and tails between k_sam_len and k_sam_write
    bool h_sam_owned = true;                                                 // false: the caller's page-locked buffer (salt_gpu_ws_reserve_text)
    uint32_t text_calls = 0;                                                 // SALT_TEXT_TRACE: stage clocks of the first text call
    uint32_t heavy_blocks = 2048, gap_blocks = 2048;
    uint32_t *d_qsub = nullptr;                                               // counters of the queue's segments
    int all_heavy = 0;
    hipStream_t stream = nullptr;
    bool timing = false;
    std::vector<hipEvent_t> ev;        // EV_PER_CALL per call: before k_pack, k_seed, k_light, k_heavy, k_gap, k_gapfin, k_cigar, after; paired end: after k_pair, k_sw, k_pe_final (+ its k_cigar)
    std::vector<uint8_t> ev_pe;        // the call was a paired-end one (its last three events are recorded)
    uint32_t n_timed = 0;
};
static const uint32_t MAX_TIMED = 256, EV_PER_CALL = 12;

static inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

static void make_view(salt_gpu_index *ix)
{
    const ImageHeader &h = ix->hdr;
    IndexView &v = ix->view;
    uint8_t *b = ix->image;
    v.c_occ = reinterpret_cast<const COcc *>(b + h.off_c_occ);
    v.c_sa = reinterpret_cast<const uint32_t *>(b + h.off_c_sa);
    v.lkt = reinterpret_cast<const uint32_t *>(b + h.off_lkt);
    v.r_occ = reinterpret_cast<const ROcc *>(b + h.off_r_occ);
    v.r_pos = reinterpret_cast<const uint32_t *>(b + h.off_r_pos);
    v.wlkt = reinterpret_cast<const uint4 *>(b + h.off_wlkt);
    v.ref = reinterpret_cast<const uint32_t *>(b + h.off_ref);
    v.text = reinterpret_cast<const uint32_t *>(b + h.off_text);
    v.c_ctx = h.off_ctx ? reinterpret_cast<const uint4 *>(b + h.off_ctx) : nullptr; v.ctx_k = h.ctx_k;
    v.r_ctx = ix->d_rctx;
    v.c_primary = h.c_primary; memcpy(v.c_L2, h.c_L2, sizeof v.c_L2); v.c_seq_len = h.c_seq_len;
    v.r_text_len = h.r_text_len; v.r_inv_sa0 = h.r_inv_sa0; memcpy(v.r_cum, h.r_cum, sizeof v.r_cum);
    v.ref_len = h.ref_len; v.lkt_len = h.lkt_len; v.r_lkt_len = h.r_lkt_len;
}

extern "C" const char *salt_gpu_last_error(void) { return g_err.c_str(); }
extern "C" uint32_t salt_gpu_result_size(void) { return (uint32_t)sizeof(salt_result_t); }

extern "C" int salt_gpu_index_attach(const salt_host_index_t *h, int device, salt_gpu_index_t **out)
{
    if (!h || !out) return fail(SALT_E_INVAL, "null argument");
    if (!h->c_bwt || !h->c_sa || !h->lkt || !h->r_bwt || !h->r_sa || !h->ref) return fail(SALT_E_INDEX, "missing index array");
    if (h->c_sa_intv == 0 || h->c_n_sa != (h->c_seq_len + h->c_sa_intv) / h->c_sa_intv) return fail(SALT_E_INDEX, "inconsistent C suffix-array sampling");
    if (h->lkt_len < 1 || h->lkt_len > 14 || h->lkt_n != (1u << (2 * h->lkt_len)) + 1) return fail(SALT_E_INDEX, "unsupported lookup-table length");
    if (h->r_n_sa != h->r_text_len - h->r_cum[4] + 1) return fail(SALT_E_INDEX, "R suffix array size does not match '#' count");
    if ((uint64_t)h->r_bwt_words * 8 < h->r_text_len) return fail(SALT_E_INDEX, "R BWT shorter than its text length");
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) return fail(SALT_E_HIP, "no HIP device visible: the salt GPU path cannot run (there is no CPU fallback)");
    HIPCHK(hipSetDevice(device));

    salt_gpu_index *ix = new salt_gpu_index();
    ix->device = device;
    ImageHeader &hd = ix->hdr;
    memset(&hd, 0, sizeof hd);
    hd.magic = IMAGE_MAGIC;
    hd.c_primary = h->c_primary; memcpy(hd.c_L2, h->c_L2, sizeof hd.c_L2); hd.c_seq_len = h->c_seq_len; hd.c_sa_intv = h->c_sa_intv;
    hd.lkt_len = h->lkt_len; hd.lkt_n = h->lkt_n;
    hd.r_text_len = h->r_text_len; hd.r_inv_sa0 = h->r_inv_sa0; memcpy(hd.r_cum, h->r_cum, sizeof hd.r_cum);
    hd.ref_len = h->ref_len;
    // The context table (c_ctx, salt_device.h): 16 B per suffix-array row (46 GiB at GRCh38 scale), taken whenever the seed length is
    // known; SALT_GPU_NO_CTX=1 leaves it out (A/B runs, small devices).
    const uint64_t ctx_bytes = (((uint64_t)h->c_seq_len + 1) * 16 + 255) / 256 * 256;
    bool want_ctx = h->l_seed > 0 && !(getenv("SALT_GPU_NO_CTX") && atoi(getenv("SALT_GPU_NO_CTX")));
    {   // width of the device k-mer table: 32 B x 4^W (14: 8 GiB, 15: 32 GiB, 16: 128 GiB).  Every extra base saves each seed
        // one C and one R backward-search step (k_seed: -9 % per base) and makes more C intervals one row wide, which the entry then
        // resolves by itself, so the widest table that leaves room for the rest is taken: W = 16 on a 288 GB MI355X.
        uint32_t w = 14;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (want_ctx && (uint64_t)free_b < ctx_bytes + (24ull << 30)) want_ctx = false;          // a device this small keeps the table narrow and the windows random
            const uint64_t avail = (uint64_t)free_b - (want_ctx ? ctx_bytes : 0);
            w = avail >= (200ull << 30) ? 16 : avail >= (72ull << 30) ? 15 : 14;
        }
        if (const char *e = getenv("SALT_GPU_LKT_LEN")) w = (uint32_t)atoi(e);
        if (h->l_seed > 0 && w > (uint32_t)h->l_seed) w = (uint32_t)h->l_seed;
        if (h->l_seed <= 0) w = h->lkt_len;
        if (w < h->lkt_len) w = h->lkt_len;
        if (w > 16) w = 16;
        hd.r_lkt_len = w;
    }
    hd.n_c_blocks = (uint64_t)h->c_seq_len / 64 + 1;
    hd.n_r_blocks = (uint64_t)h->r_text_len / 128 + 1;
    const uint64_t ref_words = ((uint64_t)h->ref_len + 7) / 8;
    uint64_t off = align_up(sizeof(ImageHeader), 256);
    hd.off_c_occ = off; off = align_up(off + hd.n_c_blocks * sizeof(COcc), 256);
    hd.off_c_sa = off;  off = align_up(off + ((uint64_t)h->c_seq_len + 1) * 4, 256);
    hd.off_lkt = off;   off = align_up(off + (uint64_t)h->lkt_n * 4, 256);
    hd.off_r_occ = off; off = align_up(off + hd.n_r_blocks * sizeof(ROcc), 256);
    hd.off_r_pos = off; off = align_up(off + ((uint64_t)h->r_text_len + 1) * 4, 256);
    hd.off_ref = off;   off = align_up(off + (ref_words + 4) * 4, 256);
    hd.off_text = off;  off = align_up(off + ((uint64_t)h->c_seq_len / 16 + 4) * 4, 256);
    // last: everything before it is the COMPACT image, from which the W-mer table can be rebuilt on any device
    hd.off_wlkt = off; off = align_up(off + (1ull << (2 * hd.r_lkt_len)) * 32, 256);
    if (want_ctx) { hd.off_ctx = off; hd.ctx_k = (uint32_t)h->l_seed; off += ctx_bytes; }
    hd.bytes = off;
    ix->bytes = off;

    // ---- upload the file-format arrays and re-pack them on the device ----
    hipError_t e = hipMalloc((void **)&ix->image, ix->bytes);
    if (e != hipSuccess) { delete ix; return fail(SALT_E_NOMEM, std::string("hipMalloc(index image): ") + hipGetErrorString(e)); }
    uint32_t *d_sa_s = nullptr, *d_r_sa = nullptr, *d_raw = nullptr, *d_minor = nullptr, *d_major = nullptr, *d_err = nullptr;
#define CHK2(x) do { hipError_t e2 = (x); if (e2 != hipSuccess) { hipFree(ix->image); hipFree(d_sa_s); hipFree(d_r_sa); hipFree(d_raw); hipFree(d_minor); hipFree(d_major); hipFree(d_err); delete ix; \
    return fail(SALT_E_HIP, std::string(#x) + ": " + hipGetErrorString(e2)); } } while (0)
    CHK2(hipMemset(ix->image, 0, hd.off_wlkt));      // the W-mer table (last) is fully written by its kernel
    CHK2(hipMemcpy(ix->image, &hd, sizeof hd, hipMemcpyHostToDevice));
    CHK2(hipMalloc((void **)&d_err, 4)); CHK2(hipMemset(d_err, 0, 4));
    {   // C: 2-bit BWT with interleaved counts (bwt.h:57-64) -> 32-byte COcc blocks
        CHK2(hipMalloc((void **)&d_raw, (uint64_t)h->c_bwt_size * 4 + 4));
        CHK2(hipMemcpy(d_raw, h->c_bwt, (uint64_t)h->c_bwt_size * 4, hipMemcpyHostToDevice));
        launch_pack_c_occ(d_raw, h->c_bwt_size, h->c_seq_len, hd.n_c_blocks, reinterpret_cast<COcc *>(ix->image + hd.off_c_occ), d_err, nullptr);
        CHK2(hipGetLastError()); CHK2(hipDeviceSynchronize());
        hipFree(d_raw); d_raw = nullptr;
    }
    {   // R: 4-bit BWT + explicit Occ values (rbwt.c:40-80) -> 64-byte ROcc blocks
        CHK2(hipMalloc((void **)&d_raw, (uint64_t)h->r_bwt_words * 4 + 4));
        CHK2(hipMemcpy(d_raw, h->r_bwt, (uint64_t)h->r_bwt_words * 4, hipMemcpyHostToDevice));
        CHK2(hipMalloc((void **)&d_minor, (uint64_t)h->r_occ_words * 4 + 4)); CHK2(hipMemcpy(d_minor, h->r_occ, (uint64_t)h->r_occ_words * 4, hipMemcpyHostToDevice));
        CHK2(hipMalloc((void **)&d_major, (uint64_t)h->r_major_words * 4 + 4)); CHK2(hipMemcpy(d_major, h->r_major, (uint64_t)h->r_major_words * 4, hipMemcpyHostToDevice));
        launch_pack_r_occ(d_raw, h->r_bwt_words, d_minor, h->r_occ_words, d_major, h->r_major_words, h->r_text_len, hd.n_r_blocks,
                          reinterpret_cast<ROcc *>(ix->image + hd.off_r_occ), d_err, nullptr);
        CHK2(hipGetLastError()); CHK2(hipDeviceSynchronize());
        hipFree(d_raw); hipFree(d_minor); hipFree(d_major); d_raw = d_minor = d_major = nullptr;
    }
    {
        uint32_t perr = 0;
        CHK2(hipMemcpy(&perr, d_err, 4, hipMemcpyDeviceToHost));
        hipFree(d_err); d_err = nullptr;
        if (perr) {
            hipFree(ix->image); delete ix;
            return fail(SALT_E_INDEX, perr & 1 ? "C BWT array shorter than seq_len" : perr & 4 ? "R BWT holds a symbol outside {A,C,G,T,#}" : "R BWT / Occ arrays shorter than the text length");
        }
    }
    CHK2(hipMemcpy(ix->image + hd.off_lkt, h->lkt, (uint64_t)h->lkt_n * 4, hipMemcpyHostToDevice));
    CHK2(hipMemcpy(ix->image + hd.off_ref, h->ref, ref_words * 4, hipMemcpyHostToDevice));
    make_view(ix);
    // ---- expand the sampled suffix arrays / tabulate the R 12-mers on the device ----
    CHK2(hipMalloc((void **)&d_sa_s, (uint64_t)h->c_n_sa * 4));
    CHK2(hipMalloc((void **)&d_r_sa, (uint64_t)h->r_n_sa * 4));
    CHK2(hipMemcpy(d_sa_s, h->c_sa, (uint64_t)h->c_n_sa * 4, hipMemcpyHostToDevice));
    CHK2(hipMemcpy(d_r_sa, h->r_sa, (uint64_t)h->r_n_sa * 4, hipMemcpyHostToDevice));
    launch_build_c_sa(ix->view, d_sa_s, h->c_sa_intv, reinterpret_cast<uint32_t *>(ix->image + hd.off_c_sa), nullptr);
    launch_build_r_pos(ix->view, d_r_sa, reinterpret_cast<uint32_t *>(ix->image + hd.off_r_pos), nullptr);
    launch_build_text(ix->view, reinterpret_cast<uint32_t *>(ix->image + hd.off_text), nullptr);       // after c_sa (same stream)
    launch_build_wlkt(ix->view, hd.r_lkt_len, reinterpret_cast<uint4 *>(ix->image + hd.off_wlkt), nullptr);
    if (hd.off_ctx) launch_build_c_ctx(ix->view, hd.ctx_k, reinterpret_cast<uint4 *>(ix->image + hd.off_ctx), nullptr);      // after c_sa and text
    CHK2(hipGetLastError());
    CHK2(hipDeviceSynchronize());
    hipFree(d_sa_s); hipFree(d_r_sa);
#undef CHK2
    *out = ix;
    return SALT_OK;
}

extern "C" void salt_gpu_index_detach(salt_gpu_index_t *ix)
{
    if (!ix) return;
    if (ix->owns && ix->image) { hipSetDevice(ix->device); hipFree(ix->image); }
    if (ix->d_pac) { hipSetDevice(ix->device); hipFree(ix->d_pac); }
    if (ix->d_rctx) { hipSetDevice(ix->device); hipFree(ix->d_rctx); }
    if (ix->d_c_off) { hipSetDevice(ix->device); hipFree(ix->d_c_off); hipFree(ix->d_c_name_off); hipFree(ix->d_c_names); }
    delete ix;
}

extern "C" int salt_gpu_index_image(const salt_gpu_index_t *ix, void **dev_ptr, uint64_t *bytes)
{
    if (!ix || !dev_ptr || !bytes) return fail(SALT_E_INVAL, "null argument");
    *dev_ptr = ix->image; *bytes = ix->bytes;
    return SALT_OK;
}

extern "C" int salt_gpu_index_attach_image(void *dev_ptr, uint64_t bytes, int device, salt_gpu_index_t **out)
{
    if (!dev_ptr || !out || bytes < sizeof(ImageHeader)) return fail(SALT_E_INVAL, "bad image");
    HIPCHK(hipSetDevice(device));
    salt_gpu_index *ix = new salt_gpu_index();
    ix->device = device; ix->image = static_cast<uint8_t *>(dev_ptr); ix->bytes = bytes; ix->owns = false;
    hipError_t e = hipMemcpy(&ix->hdr, dev_ptr, sizeof(ImageHeader), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete ix; return fail(SALT_E_HIP, std::string("hipMemcpy(image header): ") + hipGetErrorString(e)); }
    if (ix->hdr.magic != IMAGE_MAGIC || ix->hdr.bytes != bytes) { delete ix; return fail(SALT_E_INDEX, "not a salt device-index image"); }
    make_view(ix);
    *out = ix;
    return SALT_OK;
}

extern "C" int salt_gpu_index_image_compact(const salt_gpu_index_t *ix, void **dev_ptr, uint64_t *bytes)
{
    if (!ix || !dev_ptr || !bytes) return fail(SALT_E_INVAL, "null argument");
    *dev_ptr = ix->image; *bytes = ix->hdr.off_wlkt;
    return SALT_OK;
}

// the W-mer table and the context table of an image whose compact part is in place (device of ix current)
static int rebuild_wlkt(salt_gpu_index *ix)
{
    launch_build_wlkt(ix->view, ix->hdr.r_lkt_len, reinterpret_cast<uint4 *>(ix->image + ix->hdr.off_wlkt), nullptr);
    if (ix->hdr.off_ctx) launch_build_c_ctx(ix->view, ix->hdr.ctx_k, reinterpret_cast<uint4 *>(ix->image + ix->hdr.off_ctx), nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return SALT_OK;
}

extern "C" int salt_gpu_index_attach_compact(const void *dev_ptr, uint64_t bytes, int device, salt_gpu_index_t **out)
{
    if (!dev_ptr || !out || bytes < sizeof(ImageHeader)) return fail(SALT_E_INVAL, "bad image");
    HIPCHK(hipSetDevice(device));
    ImageHeader hd;
    HIPCHK(hipMemcpy(&hd, dev_ptr, sizeof hd, hipMemcpyDeviceToHost));
    if (hd.magic != IMAGE_MAGIC || hd.off_wlkt != bytes || hd.bytes < bytes) return fail(SALT_E_INDEX, "not a compact salt device-index image");
    salt_gpu_index *ix = new salt_gpu_index();
    ix->device = device; ix->bytes = hd.bytes; ix->owns = true; ix->hdr = hd;
    hipError_t e = hipMalloc((void **)&ix->image, hd.bytes);
    if (e != hipSuccess) { delete ix; return fail(SALT_E_NOMEM, std::string("hipMalloc(index image): ") + hipGetErrorString(e)); }
    e = hipMemcpy(ix->image, dev_ptr, bytes, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { hipFree(ix->image); delete ix; return fail(SALT_E_HIP, std::string("hipMemcpy(compact image): ") + hipGetErrorString(e)); }
    make_view(ix);
    int rc = rebuild_wlkt(ix);
    if (rc) { hipFree(ix->image); delete ix; return rc; }
    *out = ix;
    return SALT_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int salt_gpu_ws_create(salt_gpu_index_t *ix, uint32_t max_reads, uint64_t max_bases, salt_gpu_ws_t **out)
{
    if (!ix || !out || max_reads == 0) return fail(SALT_E_INVAL, "bad workspace size");
    HIPCHK(hipSetDevice(ix->device));
    salt_gpu_ws *ws = new salt_gpu_ws();
    ws->ix = ix; ws->max_reads = max_reads; ws->max_bases = max_bases;
#define CHKW(x) do { hipError_t e2 = (x); if (e2 != hipSuccess) { salt_gpu_ws_destroy(ws); \
    return fail(SALT_E_NOMEM, std::string(#x) + ": " + hipGetErrorString(e2)); } } while (0)
    CHKW(hipMalloc((void **)&ws->d_seqs, max_bases + 64));
    CHKW(hipMalloc((void **)&ws->d_offs, ((uint64_t)max_reads + 1) * 4));
    CHKW(hipMalloc((void **)&ws->d_results, (uint64_t)max_reads * sizeof(salt_result_t)));
    CHKW(hipMemset(ws->d_results, 0, (uint64_t)max_reads * sizeof(salt_result_t)));
    CHKW(hipMalloc((void **)&ws->d_queue, queue_words(max_reads) * 4));         // the reads k_light queues (flat, and the segments they arrive in) + k_heavy's overflow queue
    CHKW(hipMalloc((void **)&ws->d_qsub, (size_t)queue_sub_words() * 4));
    CHKW(hipMalloc((void **)&ws->d_qctl, (size_t)QCTL_WORDS * 4));
    ws->gcap = max_reads < (1u << 20) ? max_reads : (1u << 20);         // slots for reads whose gapped pass is deferred (44 B each + their rows in the pool)
    if (const char *e3 = getenv("SALT_GPU_NO_GAP_DEFER")) if (atoi(e3)) ws->gcap = 0;
    if (const char *e3 = getenv("SALT_GPU_GAP_SLOTS")) { const int v = atoi(e3); if (v > 0 && (uint32_t)v < ws->gcap) ws->gcap = (uint32_t)v; }     // tests: the overflow pass
    if (ws->gcap) {
        size_t gbytes = 0;
        gap_bufs_layout(nullptr, ws->gcap, nullptr, &gbytes);
        CHKW(hipMalloc((void **)&ws->d_gap, gbytes));
    }
    {
        hipDeviceProp_t prop;
        CHKW(hipGetDeviceProperties(&prop, ix->device));
        uint32_t per_cu = heavy_blocks_per_cu();                           // what LDS / VGPRs admit (16)
        // All of them: the persistent kernels are latency bound (alone: 4 -> 1.98, 5 -> 1.64, 6 -> 1.40, 8 -> 1.12, 12 -> 0.82, 16 -> 0.73 ms
        // per 10^6 GRCh38-scale reads; paired end 8 -> 2.65, 12 -> 2.01, 16 -> 1.76).  Until the waves took their reads through ONE counter the kernel stood at
        // 1.165 ms from 8 blocks per CU on (the counter's ~14 ns per pop x 75 700 reads) and eight was the default; with the ranged heads
        // (pop_ranged, salt_align.hip) it follows the waves again (profiles/r03/ab_heavy_ranged_pops.log)
        if (const char *e2 = getenv("SALT_GPU_HEAVY_PER_CU")) { int v = atoi(e2); if (v > 0 && (uint32_t)v <= heavy_blocks_per_cu()) per_cu = (uint32_t)v; }
        ws->heavy_blocks = (uint32_t)prop.multiProcessorCount * per_cu;    // persistent one-wave blocks
        // k_gap's items (64 candidates' Landau-Vishkin distances, ~70 us each) are independent and need no table of their own: its grid is
        // what its 9.3 KB of LDS admit per CU (16), not k_heavy's (13 079 items per 10^6 GRCh38-scale reads: 0.54 ms on 2 048 waves, 0.31 on 4 096)
        uint32_t gap_per_cu = 16;
        if (const char *e2 = getenv("SALT_GPU_GAP_PER_CU")) { int v = atoi(e2); if (v > 0 && v <= 16) gap_per_cu = (uint32_t)v; }
        ws->gap_blocks = (uint32_t)prop.multiProcessorCount * gap_per_cu;
        ws->walk_blocks = (uint32_t)prop.multiProcessorCount * 8u;            // k_seed_walk: 8 waves per SIMD in blocks of four waves
        if (const char *e2 = getenv("SALT_GPU_WALK_PER_CU")) { int v = atoi(e2); if (v > 0 && v <= 16) ws->walk_blocks = (uint32_t)prop.multiProcessorCount * (uint32_t)v; }
        CHKW(hipMalloc((void **)&ws->d_wq_cnt, (size_t)seed_wq_cnt_words() * 4));
        CHKW(hipMalloc(&ws->d_lvtab, (uint64_t)ws->heavy_blocks * lv_table_bytes()));
        const char *e = getenv("SALT_GPU_ALL_HEAVY");
        ws->all_heavy = e && atoi(e) != 0;
    }
    CHKW(hipMalloc((void **)&ws->d_ctr, SALT_CTR_N * sizeof(unsigned long long)));
    CHKW(hipMemset(ws->d_ctr, 0, SALT_CTR_N * sizeof(unsigned long long)));
    CHKW(hipStreamCreate(&ws->stream));
#undef CHKW
    *out = ws;
    return SALT_OK;
}

extern "C" void salt_gpu_ws_destroy(salt_gpu_ws_t *ws)
{
    if (!ws) return;
    hipSetDevice(ws->ix->device);
    hipFree(ws->d_seqs); hipFree(ws->d_offs); hipFree(ws->d_results); hipFree(ws->d_sai_c); hipFree(ws->d_sai_r); hipFree(ws->d_wq); hipFree(ws->d_wq_cnt); hipFree(ws->d_pm); hipFree(ws->d_tb); hipFree(ws->d_heads); if (ws->h_heads) hipHostFree(ws->h_heads); hipFree(ws->d_ctr); hipFree(ws->d_queue); hipFree(ws->d_qsub); hipFree(ws->d_qctl); hipFree(ws->d_lvtab); hipFree(ws->d_gap);
    hipFree(ws->d_raw); hipFree(ws->d_tile); hipFree(ws->d_lines); hipFree(ws->d_rec); hipFree(ws->d_tctl); hipFree(ws->d_samoff); hipFree(ws->d_samslot); hipFree(ws->d_samseg); hipFree(ws->d_scan); hipFree(ws->d_sam); hipFree(ws->d_rg);
    if (ws->h_sam && ws->h_sam_owned) hipHostFree(ws->h_sam);
    hipFree(ws->d_pe_scr); hipFree(ws->d_pairs); hipFree(ws->d_req); hipFree(ws->d_swres); hipFree(ws->d_pctl); hipFree(ws->d_sw_scr); hipFree(ws->d_pcq);
    if (ws->stream) hipStreamDestroy(ws->stream);
    for (auto &e : ws->ev) if (e) hipEventDestroy(e);
    delete ws;
}

static int check_opt(const salt_gpu_index *ix, const salt_aln_opt_t *o, uint32_t max_len, uint32_t *spr_out)
{
    if (o->l_seed < (int32_t)ix->hdr.r_lkt_len) return fail(SALT_E_INVAL, "l_seed shorter than the device k-mer table (set SALT_GPU_LKT_LEN or pass l_seed at attach)");
    if (o->l_overlap <= 0) return fail(SALT_E_INVAL, "l_overlap must be positive (aln.c:223 sets it to l_seed when -r is absent)");
    if (o->max_locate == 0 || o->max_locate > PE_LOCI_CAP) return fail(SALT_E_INVAL, "max_locate (-m) must be in 1..262144");
    if (o->max_hits != SALT_MAX_HITS) return fail(SALT_E_INVAL, "max_hits is fixed at 5 (aln.h:133)");
    if (max_len > SALT_MAX_READ_LEN) return fail(SALT_E_INVAL, "read longer than SALT_MAX_READ_LEN (512)");
    uint32_t spr = 1;
    if (max_len >= (uint32_t)o->l_seed) spr = (max_len - (uint32_t)o->l_seed) / (uint32_t)o->l_overlap + 1;
    if (spr > SALT_MAX_SEED_SLOTS) return fail(SALT_E_INVAL, "more seeds per strand than SALT_MAX_SEED_SLOTS (512): raise -r or shorten the reads");
    *spr_out = spr;
    return SALT_OK;
}

static int align_resident_impl(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, uint32_t n_reads, uint32_t max_read_len,
                               const void *d_seqs, const void *d_offs, void *d_results, void *hip_stream, int pe);

extern "C" int salt_gpu_align_se_resident(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, uint32_t n_reads, uint32_t max_read_len,
                                          const void *d_seqs, const void *d_offs, void *d_results, void *hip_stream)
{
    return align_resident_impl(ws, o, n_reads, max_read_len, d_seqs, d_offs, d_results, hip_stream, 0);
}

static int align_resident_impl(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, uint32_t n_reads, uint32_t max_read_len,
                               const void *d_seqs, const void *d_offs, void *d_results, void *hip_stream, int pe)
{
    if (!ws || !o || !d_seqs || !d_offs || !d_results) return fail(SALT_E_INVAL, "null argument");
    if (n_reads == 0) return SALT_OK;
    uint32_t spr = 0;
    int rc = check_opt(ws->ix, o, max_read_len, &spr);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ws->ix->device));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    uint64_t items = (uint64_t)n_reads * 2u * spr;
    if (items > ws->sai_cap) {                     // grows rarely; not on the steady-state path
        HIPCHK(hipStreamSynchronize(st));
        hipFree(ws->d_sai_c); hipFree(ws->d_sai_r); hipFree(ws->d_wq); ws->d_sai_c = ws->d_sai_r = ws->d_wq = nullptr; ws->sai_cap = 0;
        HIPCHK(hipMalloc((void **)&ws->d_sai_c, items * sizeof(uint4)));
        HIPCHK(hipMalloc((void **)&ws->d_sai_r, items * sizeof(uint4)));
        HIPCHK(hipMalloc((void **)&ws->d_wq, seed_wq_words(items) * 4));
        ws->sai_cap = items;
    }
    const PackGeom pg = PackGeom::make(max_read_len);
    if ((uint64_t)n_reads * pg.pm_stride > ws->pm_cap || (uint64_t)n_reads * pg.tb_stride > ws->tb_cap) {
        HIPCHK(hipStreamSynchronize(st));
        // d_heads / h_heads are sized by max_reads alone and stay (freeing them here left fetch_results with dangling pointers)
        hipFree(ws->d_pm); hipFree(ws->d_tb); ws->d_pm = ws->d_tb = nullptr; ws->pm_cap = ws->tb_cap = 0;
        const uint64_t nr = n_reads > ws->max_reads ? n_reads : ws->max_reads;
        HIPCHK(hipMalloc((void **)&ws->d_pm, nr * pg.pm_stride * 4));
        HIPCHK(hipMalloc((void **)&ws->d_tb, nr * pg.tb_stride * 4));
        ws->pm_cap = nr * pg.pm_stride; ws->tb_cap = nr * pg.tb_stride;
    }
    SeedParams sp; sp.pg = pg; sp.n_reads = n_reads; sp.spr = spr; sp.l_seed = o->l_seed; sp.l_overlap = o->l_overlap;
    sp.max_seed = o->max_seed; sp.seed_only_ref = o->seed_only_ref;
    { static const bool off = getenv("SALT_GPU_NO_UNIQUE") && atoi(getenv("SALT_GPU_NO_UNIQUE")); sp.resolve_unique = !off; }
    if (n_reads > ws->max_reads) return fail(SALT_E_CAPACITY, "more reads than the workspace holds");
    AlignParams ap; ap.pg = pg; ap.n_reads = n_reads; ap.spr = spr; ap.l_seed = o->l_seed; ap.max_locate = o->max_locate; ap.max_hits = o->max_hits;
    ap.all_heavy = ws->all_heavy; ap.pe = pe; ap.dbg_stop = 0; ap.heavy_stop = 0; ap.max_amb = pe ? 5u : 200u;
#ifdef SALT_DIAG
    { const char *e = getenv("SALT_GPU_LIGHT_STOP"); ap.dbg_stop = e ? atoi(e) : 0; } { static const int hs = getenv("SALT_GPU_HEAVY_STOP") ? atoi(getenv("SALT_GPU_HEAVY_STOP")) : 0; ap.heavy_stop = hs; }
#endif
    // located rows beyond the LDS list (SALT_MAX_LOCATE) live in a global list per persistent block: paired end always may need it (0x40000
    // loci per strand, alnse.c:42,533), single end when -m is above the LDS list (the reference grows its vector, alnse.c:678, kvec.h)
    const bool glob_loci = pe || o->max_locate > SALT_MAX_LOCATE;
    if (glob_loci && !ws->d_pe_scr) HIPCHK(hipMalloc((void **)&ws->d_pe_scr, (uint64_t)ws->heavy_blocks * PE_LOCI_CAP * 5));
    unsigned long long *ctr = o->collect_counters ? ws->d_ctr : nullptr;
    const bool timed = ws->timing && ws->n_timed < MAX_TIMED;
    hipEvent_t *ev = timed ? &ws->ev[(size_t)ws->n_timed * EV_PER_CALL] : nullptr;
    HIPCHK(hipMemsetAsync(ws->d_qctl, 0, (size_t)QCTL_WORDS * 4, st));
    if (timed) HIPCHK(hipEventRecord(ev[0], st));
    launch_pack(pg, n_reads, static_cast<const uint8_t *>(d_seqs), static_cast<const uint32_t *>(d_offs), ws->d_pm, ws->d_tb, st);
    if (timed) HIPCHK(hipEventRecord(ev[1], st));
    launch_seed(ws->ix->view, sp, ws->d_tb, ws->d_sai_c, ws->d_sai_r, ws->d_wq, ws->d_wq_cnt, ws->walk_blocks, ctr, st);
    if (timed) HIPCHK(hipEventRecord(ev[2], st));
    if (!ap.all_heavy)
        launch_light(ws->ix->view, ap, ws->d_pm, static_cast<const uint8_t *>(d_seqs), static_cast<const uint32_t *>(d_offs), ws->d_sai_c, ws->d_sai_r,
                     static_cast<salt_result_t *>(d_results), ws->d_queue, ws->d_qctl, ws->d_queue + 2 * (size_t)ws->max_reads, ws->d_qsub, ctr, st);
    else HIPCHK(hipMemsetAsync(ws->d_qsub, 0, (size_t)queue_sub_words() * 4, st));          // (launch_light does it otherwise: k_heavy's queue heads live there)
    if (timed) HIPCHK(hipEventRecord(ev[3], st));
    launch_heavy(ws->ix->view, ap, ws->d_pm, ws->d_sai_c, ws->d_sai_r,
                 static_cast<salt_result_t *>(d_results), ws->d_queue, ctr, ws->heavy_blocks, ws->gap_blocks, ws->d_lvtab,
                 gap_bufs_layout(ws->d_gap, ws->gcap, ws->d_qctl, nullptr), ws->d_queue + ws->max_reads, ws->d_qsub + queue_heads_offset(), glob_loci ? ws->d_pe_scr : nullptr, timed ? ev + 4 : nullptr, st);
    if (timed) { HIPCHK(hipEventRecord(ev[7], st)); ws->ev_pe[ws->n_timed] = 0; ++ws->n_timed; }
    HIPCHK(hipGetLastError());
    return SALT_OK;
}

// Results to host memory without moving 880 bytes per read over PCIe: nearly every row is fully described by its first
// 128 bytes (header, hits, hit_n_cigar, the first 8 CIGAR ops); those travel as one dense copy, the few rows with longer
// or alternative-hit CIGARs are fetched whole.
static const uint32_t HEAD_BYTES = 128;
static_assert(offsetof(salt_result_t, cigar) + 8 * sizeof(uint16_t) == HEAD_BYTES, "result head");
static int fetch_results(salt_gpu_ws_t *ws, uint32_t n_reads, salt_result_t *results, hipStream_t st)
{
    if (!ws->d_heads) {
        HIPCHK(hipMalloc((void **)&ws->d_heads, (uint64_t)ws->max_reads * HEAD_BYTES));
        HIPCHK(hipHostMalloc((void **)&ws->h_heads, (uint64_t)ws->max_reads * HEAD_BYTES, hipHostMallocDefault));
    }
    launch_heads(ws->d_results, n_reads, ws->d_heads, st);
    HIPCHK(hipMemcpyAsync(ws->h_heads, ws->d_heads, (uint64_t)n_reads * HEAD_BYTES, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // heads -> rows: 128 bytes into every 880-byte row (two cache lines of a strided 88 MB per 100 000 reads); big batches are
    // split over a few threads, the caller's thread among them
    std::vector<uint32_t> full;
    const uint32_t n_thr = n_reads >= 262144 ? 4u : 1u;        // a driver with 100 000-read batches (salt) runs several workers already
    std::vector<std::vector<uint32_t>> full_t(n_thr);
    auto scatter = [&](uint32_t t) {
        const uint32_t lo = (uint32_t)((uint64_t)n_reads * t / n_thr), hi = (uint32_t)((uint64_t)n_reads * (t + 1) / n_thr);
        for (uint32_t i = lo; i < hi; ++i) {
            memcpy(&results[i], ws->h_heads + (uint64_t)i * HEAD_BYTES, HEAD_BYTES);
            const salt_result_t &r = results[i];
            bool more = r.n_cigar > 8;
            for (int h = 0; h < SALT_MAX_HITS; ++h) more |= r.hit_n_cigar[h] != 0;
            if (more) full_t[t].push_back(i);
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < n_thr; ++t) th.emplace_back(scatter, t);
        scatter(0);
        for (auto &x : th) x.join();
    }
    for (auto &v : full_t) full.insert(full.end(), v.begin(), v.end());
    if (full.size() > 4096) {                                  // unusual batch: one plain copy is cheaper than thousands of small ones
        HIPCHK(hipMemcpyAsync(results, ws->d_results, (uint64_t)n_reads * sizeof(salt_result_t), hipMemcpyDeviceToHost, st));
    } else {
        for (uint32_t i : full) HIPCHK(hipMemcpyAsync(&results[i], ws->d_results + i, sizeof(salt_result_t), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return SALT_OK;
}

extern "C" int salt_gpu_align_se(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, uint32_t n_reads, const uint8_t *seqs,
                                 const uint32_t *offs, salt_result_t *results)
{
    if (!ws || !o || !seqs || !offs || !results) return fail(SALT_E_INVAL, "null argument");
    if (n_reads == 0) return SALT_OK;
    if (n_reads > ws->max_reads) return fail(SALT_E_CAPACITY, "more reads than the workspace holds");
    if (offs[0] != 0) return fail(SALT_E_INVAL, "offs[0] must be 0");
    uint64_t bases = offs[n_reads];
    if (bases > ws->max_bases) return fail(SALT_E_CAPACITY, "more bases than the workspace holds");
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n_reads; ++i) {
        if (offs[i + 1] < offs[i]) return fail(SALT_E_INVAL, "offs must be non-decreasing");
        uint32_t l = offs[i + 1] - offs[i];
        if (l == 0) return fail(SALT_E_INVAL, "empty read");
        max_len = l > max_len ? l : max_len;
    }
    HIPCHK(hipSetDevice(ws->ix->device));
    HIPCHK(hipMemcpyAsync(ws->d_seqs, seqs, bases, hipMemcpyHostToDevice, ws->stream));
    HIPCHK(hipMemcpyAsync(ws->d_offs, offs, ((uint64_t)n_reads + 1) * 4, hipMemcpyHostToDevice, ws->stream));
    int rc = salt_gpu_align_se_resident(ws, o, n_reads, max_len, ws->d_seqs, ws->d_offs, ws->d_results, ws->stream);
    if (rc) return rc;
    return fetch_results(ws, n_reads, results, ws->stream);
}

// ---------------------------------------------------------------------------------------------
// FASTQ text in, SAM text out
// ---------------------------------------------------------------------------------------------
extern "C" int salt_gpu_index_set_contigs(salt_gpu_index_t *ix, int32_t n, const int64_t *offsets, const char *const *names)
{
    if (!ix || n <= 0 || !offsets || !names) return fail(SALT_E_INVAL, "bad contig table");
    HIPCHK(hipSetDevice(ix->device));
    std::vector<uint32_t> noff((size_t)n + 1, 0); std::string blob;
    for (int i = 0; i < n; ++i) { if (!names[i]) return fail(SALT_E_INVAL, "contig without a name"); blob += names[i]; noff[(size_t)i + 1] = (uint32_t)blob.size(); }
    if (ix->d_c_off) { hipFree(ix->d_c_off); hipFree(ix->d_c_name_off); hipFree(ix->d_c_names); ix->d_c_off = nullptr; ix->d_c_name_off = nullptr; ix->d_c_names = nullptr; ix->n_contigs = 0; }
    HIPCHK(hipMalloc((void **)&ix->d_c_off, (size_t)n * 8));
    HIPCHK(hipMalloc((void **)&ix->d_c_name_off, ((size_t)n + 1) * 4));
    HIPCHK(hipMalloc((void **)&ix->d_c_names, blob.size() + 1));
    HIPCHK(hipMemcpy(ix->d_c_off, offsets, (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_c_name_off, noff.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_c_names, blob.data(), blob.size(), hipMemcpyHostToDevice));
    ix->n_contigs = n;
    HIPCHK(text_warm());                                     // the text kernels' code object is loaded now, not inside the first block's call
    return SALT_OK;
}

extern "C" int salt_gpu_host_alloc(uint64_t bytes, void **ptr)
{
    if (!ptr || !bytes) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return SALT_OK;
}
extern "C" void salt_gpu_host_free(void *ptr) { if (ptr) hipHostFree(ptr); }

// The host NUMA node a device hangs off (its PCI function's numa_node in sysfs); -1 when the platform does not say.  A multi-GPU driver pins
// each device's worker threads -- and allocates their page-locked buffers from them -- on that node: the reads' and the SAM text's DMA then
// stays on the socket the GPU is attached to.
extern "C" int salt_gpu_device_numa_node(int device, int *node)
{
    if (!node) return fail(SALT_E_INVAL, "null argument");
    *node = -1;
    char bus[64] = { 0 };
    HIPCHK(hipDeviceGetPCIBusId(bus, (int)sizeof bus, device));
    for (char *c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (FILE *f = fopen(path.c_str(), "r")) { int v = -1; if (fscanf(f, "%d", &v) == 1) *node = v; fclose(f); }
    return SALT_OK;
}
extern "C" int salt_gpu_device_count(int *n)
{
    if (!n) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipGetDeviceCount(n));
    return SALT_OK;
}

#define REGROW(ptr, cap, need, type) do { if ((need) > (cap)) { HIPCHK(hipStreamSynchronize(st)); hipFree(ptr); (ptr) = nullptr; (cap) = 0; \
    const uint64_t want_ = (need) + (need) / 4; HIPCHK(hipMalloc((void **)&(ptr), want_ * sizeof(type))); (cap) = want_; } } while (0)

// Sizes every buffer a text call of up to max_block_bytes / est_reads reads of max_read_len bases will ask for, so that the first call
// on the workspace finds them (a later, larger block still regrows them).  One-time work a driver does next to attaching the index.
extern "C" int salt_gpu_ws_reserve_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, uint64_t max_block_bytes, uint32_t est_reads, uint32_t max_read_len,
                                        uint64_t est_sam_bytes, void *host_sam, uint64_t host_sam_bytes)
{
    if (!ws || !o || max_block_bytes == 0 || est_reads == 0) return fail(SALT_E_INVAL, "bad reserve arguments");
    if (est_reads > ws->max_reads) est_reads = ws->max_reads;
    uint32_t spr = 0;
    int rc = check_opt(ws->ix, o, max_read_len, &spr);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ws->ix->device));
    hipStream_t st = ws->stream;
    REGROW(ws->d_raw, ws->raw_cap, max_block_bytes + 64, uint8_t);
    if (!ws->d_tctl) HIPCHK(hipMalloc((void **)&ws->d_tctl, 32));      // parse ctl[4] | the SAM block's byte count in 64 bits
    REGROW(ws->d_tile, ws->tile_cap, ws->raw_cap / FQ_TILE + 4, uint32_t);
    {
        const size_t need = text_scan_bytes(std::max<uint64_t>(ws->tile_cap, (uint64_t)ws->max_reads + 2));
        if (need > ws->scan_bytes) { hipFree(ws->d_scan); ws->d_scan = nullptr; ws->scan_bytes = 0; HIPCHK(hipMalloc(&ws->d_scan, need)); ws->scan_bytes = need; }
    }
    REGROW(ws->d_lines, ws->lines_cap, 4ull * est_reads + 8, uint32_t);
    if (!ws->d_rec) HIPCHK(hipMalloc((void **)&ws->d_rec, (uint64_t)ws->max_reads * sizeof(FqRec)));
    if (!ws->d_samoff) HIPCHK(hipMalloc((void **)&ws->d_samoff, ((uint64_t)ws->max_reads + 2) * 4));
    if (!ws->d_samslot) { HIPCHK(hipMalloc((void **)&ws->d_samslot, (uint64_t)ws->max_reads * SAM_SLOT)); HIPCHK(hipMalloc((void **)&ws->d_samseg, (uint64_t)ws->max_reads * sizeof(SamSeg))); }
    const uint64_t items = (uint64_t)est_reads * 2u * spr;
    if (items > ws->sai_cap) {
        hipFree(ws->d_sai_c); hipFree(ws->d_sai_r); hipFree(ws->d_wq); ws->d_sai_c = ws->d_sai_r = ws->d_wq = nullptr; ws->sai_cap = 0;
        HIPCHK(hipMalloc((void **)&ws->d_sai_c, items * sizeof(uint4)));
        HIPCHK(hipMalloc((void **)&ws->d_sai_r, items * sizeof(uint4)));
        HIPCHK(hipMalloc((void **)&ws->d_wq, seed_wq_words(items) * 4));
        ws->sai_cap = items;
    }
    const PackGeom pg = PackGeom::make(max_read_len);
    if ((uint64_t)ws->max_reads * pg.pm_stride > ws->pm_cap || (uint64_t)ws->max_reads * pg.tb_stride > ws->tb_cap) {
        hipFree(ws->d_pm); hipFree(ws->d_tb); ws->d_pm = ws->d_tb = nullptr; ws->pm_cap = ws->tb_cap = 0;
        HIPCHK(hipMalloc((void **)&ws->d_pm, (uint64_t)ws->max_reads * pg.pm_stride * 4));
        HIPCHK(hipMalloc((void **)&ws->d_tb, (uint64_t)ws->max_reads * pg.tb_stride * 4));
        ws->pm_cap = (uint64_t)ws->max_reads * pg.pm_stride; ws->tb_cap = (uint64_t)ws->max_reads * pg.tb_stride;
    }
    if (host_sam && host_sam_bytes > est_sam_bytes + 64) est_sam_bytes = host_sam_bytes - 64;
    if (est_sam_bytes + 64 > ws->sam_cap) {
        hipFree(ws->d_sam); ws->d_sam = nullptr; if (ws->h_sam && ws->h_sam_owned) hipHostFree(ws->h_sam); ws->h_sam = nullptr; ws->sam_cap = 0;
        HIPCHK(hipMalloc((void **)&ws->d_sam, est_sam_bytes + 64));
        if (host_sam && host_sam_bytes >= est_sam_bytes + 64) { ws->h_sam = static_cast<char *>(host_sam); ws->h_sam_owned = false; }
        else { HIPCHK(hipHostMalloc((void **)&ws->h_sam, est_sam_bytes + 64, hipHostMallocDefault)); ws->h_sam_owned = true; }
        ws->sam_cap = est_sam_bytes + 64;
    }
    return SALT_OK;
}

extern "C" int salt_gpu_align_se_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_text_opt_t *to, const char *fastq, uint64_t n_bytes,
                                      const char **sam, uint64_t *sam_bytes, uint32_t *n_reads)
{
    if (!ws || !o || !to || !fastq || !sam || !sam_bytes || !n_reads) return fail(SALT_E_INVAL, "null argument");
    *sam = nullptr; *sam_bytes = 0; *n_reads = 0;
    if (n_bytes == 0) return SALT_OK;
    if (n_bytes >= 0xFFFFFFF0ull) return fail(SALT_E_CAPACITY, "FASTQ block of 4 GiB or more");
    if (fastq[n_bytes - 1] != '\n') return fail(SALT_E_INVAL, "FASTQ block must end with a newline");
    salt_gpu_index *ix = ws->ix;
    if (!ix->d_c_off) return fail(SALT_E_INVAL, "SAM text needs the contig table: call salt_gpu_index_set_contigs first");
    HIPCHK(hipSetDevice(ix->device));
    hipStream_t st = ws->stream;
    const bool trace = ws->text_calls++ == 0 && getenv("SALT_TEXT_TRACE");
    double tm[8]; int n_tm = 0;
    auto mark = [&]() { if (trace && n_tm < 8) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); tm[n_tm++] = (double)ts.tv_sec + ts.tv_nsec * 1e-9; } };
    mark();
    // ---- the raw block and its lines ----
    REGROW(ws->d_raw, ws->raw_cap, n_bytes + 64, uint8_t);
    const uint64_t n_tiles = (n_bytes + FQ_TILE - 1) / FQ_TILE;
    if (!ws->d_tctl) HIPCHK(hipMalloc((void **)&ws->d_tctl, 32));      // parse ctl[4] | the SAM block's byte count in 64 bits
    {   // tile counters + scan scratch follow the raw capacity
        const uint64_t tiles_cap = ws->raw_cap / FQ_TILE + 4;
        REGROW(ws->d_tile, ws->tile_cap, tiles_cap, uint32_t);
        const size_t need = text_scan_bytes(std::max<uint64_t>(ws->tile_cap, (uint64_t)ws->max_reads + 2));
        if (need > ws->scan_bytes) { HIPCHK(hipStreamSynchronize(st)); hipFree(ws->d_scan); ws->d_scan = nullptr; ws->scan_bytes = 0; HIPCHK(hipMalloc(&ws->d_scan, need)); ws->scan_bytes = need; }
    }
    mark();
    HIPCHK(hipMemcpyAsync(ws->d_raw, fastq, n_bytes, hipMemcpyHostToDevice, st));
    // newline count first: the line table is sized by it
    uint32_t n_nl = 0;
    HIPCHK(launch_fq_count(ws->d_raw, n_bytes, ws->d_tile, ws->d_scan, ws->scan_bytes, st));
    HIPCHK(hipMemcpyAsync(&n_nl, ws->d_tile + n_tiles, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_nl % 4 != 0) return fail(SALT_E_INVAL, "FASTQ block does not hold whole 4-line records (" + std::to_string(n_nl) + " lines)");
    const uint32_t n_rec = n_nl / 4;
    if (n_rec > ws->max_reads) return fail(SALT_E_CAPACITY, "more reads in the block (" + std::to_string(n_rec) + ") than the workspace holds");
    if (n_rec == 0) return SALT_OK;
    mark();
    REGROW(ws->d_lines, ws->lines_cap, (uint64_t)n_nl + 8, uint32_t);
    HIPCHK(launch_fq_lines(ws->d_raw, n_bytes, ws->d_tile, ws->d_lines, st));
    // ---- records, offsets, codes ----
    if (!ws->d_rec) HIPCHK(hipMalloc((void **)&ws->d_rec, (uint64_t)ws->max_reads * sizeof(FqRec)));
    HIPCHK(launch_fq_parse(ws->d_raw, ws->d_lines, n_rec, ws->d_rec, ws->d_offs, ws->d_tctl, ws->d_scan, ws->scan_bytes, st));
    uint32_t ctl[4] = { 0, 0, 0, 0 }, bases = 0;
    HIPCHK(hipMemcpyAsync(ctl, ws->d_tctl, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&bases, ws->d_offs + n_rec, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (ctl[0]) {
        const char *what = ctl[0] & 1 ? "a record does not start with '@'" : ctl[0] & 2 ? "the third line of a record does not start with '+'"
                         : ctl[0] & 4 ? "sequence and quality lengths differ" : "empty read";
        return fail(SALT_E_INVAL, std::string("input is not 4-line FASTQ at record ") + std::to_string(ctl[2]) + " of the block: " + what);
    }
    if ((uint64_t)bases > ws->max_bases) {
        HIPCHK(hipStreamSynchronize(st));
        hipFree(ws->d_seqs); ws->d_seqs = nullptr;
        HIPCHK(hipMalloc((void **)&ws->d_seqs, (uint64_t)bases + bases / 4 + 64));
        ws->max_bases = (uint64_t)bases + bases / 4;
    }
    HIPCHK(launch_fq_codes(ws->d_raw, ws->d_rec, ws->d_offs, n_rec, ws->d_seqs, st));
    // ---- align ----
    mark();
    int rc = align_resident_impl(ws, o, n_rec, ctl[1], ws->d_seqs, ws->d_offs, ws->d_results, st, 0);
    if (rc) return rc;
    mark();
    // ---- SAM text ----
    const std::string rg = to->rg_id ? to->rg_id : "";
    if (rg != ws->rg || (!rg.empty() && !ws->d_rg)) {
        HIPCHK(hipStreamSynchronize(st));
        hipFree(ws->d_rg); ws->d_rg = nullptr;
        if (!rg.empty()) { HIPCHK(hipMalloc((void **)&ws->d_rg, rg.size() + 1)); HIPCHK(hipMemcpy(ws->d_rg, rg.data(), rg.size(), hipMemcpyHostToDevice)); }
        ws->rg = rg;
    }
    if (!ws->d_samoff) HIPCHK(hipMalloc((void **)&ws->d_samoff, ((uint64_t)ws->max_reads + 2) * 4));
    if (!ws->d_samslot) { HIPCHK(hipMalloc((void **)&ws->d_samslot, (uint64_t)ws->max_reads * SAM_SLOT)); HIPCHK(hipMalloc((void **)&ws->d_samseg, (uint64_t)ws->max_reads * sizeof(SamSeg))); }
    SamDev d;
    d.raw = ws->d_raw; d.rec = ws->d_rec; d.seqs = ws->d_seqs; d.offs = ws->d_offs; d.res = ws->d_results;
    d.c_off = ix->d_c_off; d.c_name_off = ix->d_c_name_off; d.c_names = ix->d_c_names; d.n_contigs = ix->n_contigs;
    d.text = ix->view.text; d.ref = ix->view.ref; d.xa_cigar = to->print_xa_cigar; d.nm_md = to->print_nm_md;
    d.rg = ws->d_rg; d.rg_len = to->rg_id ? (int32_t)rg.size() : 0;
    d.pe = 0; d.min_tlen = d.max_tlen = 0; d.slot = ws->d_samslot; d.seg = ws->d_samseg; d.tb = ws->d_tb; d.pg = PackGeom::make(ctl[1]);
    if (to->rg_id && rg.empty()) return fail(SALT_E_INVAL, "empty read group id");
    HIPCHK(launch_sam_len(d, n_rec, ws->d_samoff, reinterpret_cast<unsigned long long *>(ws->d_tctl + 4), ws->d_scan, ws->scan_bytes, st));
    uint32_t total = 0; unsigned long long total64 = 0;
    HIPCHK(hipMemcpyAsync(&total, ws->d_samoff + n_rec, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&total64, ws->d_tctl + 4, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (total64 >> 32) return fail(SALT_E_CAPACITY, "the SAM text of this block passes 4 GiB (its offsets are 32-bit): hand over smaller blocks (SALT_CHUNK_MB)");
    mark();
    if ((uint64_t)total + 64 > ws->sam_cap) {
        hipFree(ws->d_sam); ws->d_sam = nullptr; if (ws->h_sam && ws->h_sam_owned) hipHostFree(ws->h_sam); ws->h_sam = nullptr; ws->sam_cap = 0; ws->h_sam_owned = true;
        const uint64_t want = (uint64_t)total + total / 4 + 64;
        HIPCHK(hipMalloc((void **)&ws->d_sam, want));
        HIPCHK(hipHostMalloc((void **)&ws->h_sam, want, hipHostMallocDefault));
        ws->sam_cap = want;
    }
    mark();
    HIPCHK(launch_sam_write(d, n_rec, ws->d_samoff, ws->d_sam, st));
    HIPCHK(hipMemcpyAsync(ws->h_sam, ws->d_sam, total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    mark();
    if (trace && n_tm == 8)
        fprintf(stderr, "[salt_gpu] first text call (ms): raw buffers %.1f, copy in + count %.1f, lines/parse/codes %.1f, align launch (+ its buffers) %.1f, "
                        "kernels + SAM lengths %.1f, SAM buffers %.1f, write + copy out %.1f\n", (tm[1] - tm[0]) * 1e3, (tm[2] - tm[1]) * 1e3, (tm[3] - tm[2]) * 1e3,
                (tm[4] - tm[3]) * 1e3, (tm[5] - tm[4]) * 1e3, (tm[6] - tm[5]) * 1e3, (tm[7] - tm[6]) * 1e3);
    *sam = ws->h_sam; *sam_bytes = total; *n_reads = n_rec;
    return SALT_OK;
}

static int pe_resident_impl(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_pe_opt_t *pe, uint32_t n_pairs, uint32_t max_len,
                            const void *d_seqs, const void *d_offs, void *d_results, hipStream_t st);

// Paired end: two blocks holding the same number of whole 4-line records (mates in file order); the SAM block holds both records of
// every pair, each followed by the reference's empty line (alnpe.c:640-648).
extern "C" int salt_gpu_align_pe_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_pe_opt_t *pe, const salt_text_opt_t *to,
                                      const char *fastq1, uint64_t n1, const char *fastq2, uint64_t n2,
                                      const char **sam, uint64_t *sam_bytes, uint32_t *n_pairs)
{
    if (!ws || !o || !pe || !to || !fastq1 || !fastq2 || !sam || !sam_bytes || !n_pairs) return fail(SALT_E_INVAL, "null argument");
    *sam = nullptr; *sam_bytes = 0; *n_pairs = 0;
    if (n1 == 0 && n2 == 0) return SALT_OK;
    if (n1 == 0 || n2 == 0) return fail(SALT_E_INVAL, "the two FASTQ blocks hold different numbers of reads");
    if (n1 + n2 >= 0xFFFFFFE0ull) return fail(SALT_E_CAPACITY, "FASTQ blocks of 4 GiB or more");
    if (fastq1[n1 - 1] != '\n' || fastq2[n2 - 1] != '\n') return fail(SALT_E_INVAL, "FASTQ block must end with a newline");
    salt_gpu_index *ix = ws->ix;
    if (!ix->d_c_off) return fail(SALT_E_INVAL, "SAM text needs the contig table: call salt_gpu_index_set_contigs first");
    if (!ix->d_pac) return fail(SALT_E_INVAL, "paired end needs the 2-bit genome: call salt_gpu_index_set_pac first");
    HIPCHK(hipSetDevice(ix->device));
    hipStream_t st = ws->stream;
    const uint64_t b2 = (n1 + 3) & ~3ull;                      // block 2 behind block 1, on a word boundary
    REGROW(ws->d_raw, ws->raw_cap, b2 + n2 + 64, uint8_t);
    const uint64_t t1 = (n1 + FQ_TILE - 1) / FQ_TILE, t2 = (n2 + FQ_TILE - 1) / FQ_TILE;
    if (!ws->d_tctl) HIPCHK(hipMalloc((void **)&ws->d_tctl, 32));      // parse ctl[4] | the SAM block's byte count in 64 bits
    {
        REGROW(ws->d_tile, ws->tile_cap, ws->raw_cap / FQ_TILE + 16, uint32_t);
        const size_t need = text_scan_bytes(std::max<uint64_t>(ws->tile_cap, (uint64_t)ws->max_reads + 2));
        if (need > ws->scan_bytes) { HIPCHK(hipStreamSynchronize(st)); hipFree(ws->d_scan); ws->d_scan = nullptr; ws->scan_bytes = 0; HIPCHK(hipMalloc(&ws->d_scan, need)); ws->scan_bytes = need; }
    }
    uint32_t *tile1 = ws->d_tile, *tile2 = ws->d_tile + t1 + 4;
    HIPCHK(hipMemcpyAsync(ws->d_raw, fastq1, n1, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ws->d_raw + b2, fastq2, n2, hipMemcpyHostToDevice, st));
    uint32_t nl[2] = { 0, 0 };
    HIPCHK(launch_fq_count(ws->d_raw, n1, tile1, ws->d_scan, ws->scan_bytes, st));
    HIPCHK(launch_fq_count(ws->d_raw + b2, n2, tile2, ws->d_scan, ws->scan_bytes, st));
    HIPCHK(hipMemcpyAsync(&nl[0], tile1 + t1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&nl[1], tile2 + t2, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (nl[0] % 4 != 0 || nl[1] % 4 != 0) return fail(SALT_E_INVAL, "FASTQ block does not hold whole 4-line records (" + std::to_string(nl[0]) + " / " + std::to_string(nl[1]) + " lines)");
    if (nl[0] != nl[1]) return fail(SALT_E_INVAL, "the two FASTQ blocks hold different numbers of reads (" + std::to_string(nl[0] / 4) + " / " + std::to_string(nl[1] / 4) + ")");
    const uint32_t n = nl[0] / 4, n_rec = 2 * n;
    if (n_rec > ws->max_reads) return fail(SALT_E_CAPACITY, "more reads in the blocks (" + std::to_string(n_rec) + ") than the workspace holds");
    if (n == 0) return SALT_OK;
    REGROW(ws->d_lines, ws->lines_cap, (uint64_t)nl[0] + nl[1] + 24, uint32_t);
    uint32_t *lines1 = ws->d_lines, *lines2 = ws->d_lines + nl[0] + 8;
    HIPCHK(launch_fq_lines(ws->d_raw, n1, tile1, lines1, st));
    HIPCHK(launch_fq_lines(ws->d_raw + b2, n2, tile2, lines2, st));
    if (!ws->d_rec) HIPCHK(hipMalloc((void **)&ws->d_rec, (uint64_t)ws->max_reads * sizeof(FqRec)));
    HIPCHK(launch_fq_ctl_init(ws->d_tctl, st));
    HIPCHK(hipMemsetAsync(ws->d_offs + n_rec, 0, 4, st));
    HIPCHK(launch_fq_parse_mate(ws->d_raw, 0u, lines1, n, 0u, ws->d_rec, ws->d_offs, ws->d_tctl, st));
    HIPCHK(launch_fq_parse_mate(ws->d_raw, (uint32_t)b2, lines2, n, 1u, ws->d_rec, ws->d_offs, ws->d_tctl, st));
    HIPCHK(launch_text_scan(ws->d_offs, n_rec + 1, ws->d_scan, ws->scan_bytes, st));
    uint32_t ctl[4] = { 0, 0, 0, 0 }, bases = 0;
    HIPCHK(hipMemcpyAsync(ctl, ws->d_tctl, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&bases, ws->d_offs + n_rec, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (ctl[0]) {
        const char *what = ctl[0] & 1 ? "a record does not start with '@'" : ctl[0] & 2 ? "the third line of a record does not start with '+'"
                         : ctl[0] & 4 ? "sequence and quality lengths differ" : "empty read";
        return fail(SALT_E_INVAL, std::string("input is not 4-line FASTQ at record ") + std::to_string(ctl[2]) + " of the blocks: " + what);
    }
    if ((uint64_t)bases > ws->max_bases) {
        hipFree(ws->d_seqs); ws->d_seqs = nullptr;
        HIPCHK(hipMalloc((void **)&ws->d_seqs, (uint64_t)bases + bases / 4 + 64));
        ws->max_bases = (uint64_t)bases + bases / 4;
    }
    HIPCHK(launch_fq_codes(ws->d_raw, ws->d_rec, ws->d_offs, n_rec, ws->d_seqs, st));
    int rc = pe_resident_impl(ws, o, pe, n, ctl[1], ws->d_seqs, ws->d_offs, ws->d_results, st);
    if (rc) return rc;
    const std::string rg = to->rg_id ? to->rg_id : "";
    if (to->rg_id && rg.empty()) return fail(SALT_E_INVAL, "empty read group id");
    if (rg != ws->rg || (!rg.empty() && !ws->d_rg)) {
        HIPCHK(hipStreamSynchronize(st));
        hipFree(ws->d_rg); ws->d_rg = nullptr;
        if (!rg.empty()) { HIPCHK(hipMalloc((void **)&ws->d_rg, rg.size() + 1)); HIPCHK(hipMemcpy(ws->d_rg, rg.data(), rg.size(), hipMemcpyHostToDevice)); }
        ws->rg = rg;
    }
    if (!ws->d_samoff) HIPCHK(hipMalloc((void **)&ws->d_samoff, ((uint64_t)ws->max_reads + 2) * 4));
    if (!ws->d_samslot) { HIPCHK(hipMalloc((void **)&ws->d_samslot, (uint64_t)ws->max_reads * SAM_SLOT)); HIPCHK(hipMalloc((void **)&ws->d_samseg, (uint64_t)ws->max_reads * sizeof(SamSeg))); }
    SamDev d;
    d.raw = ws->d_raw; d.rec = ws->d_rec; d.seqs = ws->d_seqs; d.offs = ws->d_offs; d.res = ws->d_results;
    d.c_off = ix->d_c_off; d.c_name_off = ix->d_c_name_off; d.c_names = ix->d_c_names; d.n_contigs = ix->n_contigs;
    d.text = ix->view.text; d.ref = ix->view.ref; d.xa_cigar = to->print_xa_cigar; d.nm_md = to->print_nm_md;
    d.rg = ws->d_rg; d.rg_len = to->rg_id ? (int32_t)rg.size() : 0;
    d.pe = 1; d.min_tlen = pe->min_tlen; d.max_tlen = pe->max_tlen; d.slot = ws->d_samslot; d.seg = ws->d_samseg; d.tb = ws->d_tb; d.pg = PackGeom::make(ctl[1]);
    HIPCHK(launch_sam_len(d, n_rec, ws->d_samoff, reinterpret_cast<unsigned long long *>(ws->d_tctl + 4), ws->d_scan, ws->scan_bytes, st));
    uint32_t total = 0, n_over = 0; unsigned long long total64 = 0;
    HIPCHK(hipMemcpyAsync(&total, ws->d_samoff + n_rec, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&total64, ws->d_tctl + 4, 8, hipMemcpyDeviceToHost, st));
    if (ws->d_pctl) HIPCHK(hipMemcpyAsync(&n_over, ws->d_pctl + 4, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_over) return fail(SALT_E_CAPACITY, std::to_string(n_over) + " mate rescue(s) need a Smith-Waterman band wider than this build holds (SW_BAND_W): "
                                             "the rows of this batch would differ from the reference's");
    if (total64 >> 32) return fail(SALT_E_CAPACITY, "the SAM text of this block passes 4 GiB (its offsets are 32-bit): hand over smaller blocks (SALT_CHUNK_MB)");
    if ((uint64_t)total + 64 > ws->sam_cap) {
        hipFree(ws->d_sam); ws->d_sam = nullptr; if (ws->h_sam && ws->h_sam_owned) hipHostFree(ws->h_sam); ws->h_sam = nullptr; ws->sam_cap = 0; ws->h_sam_owned = true;
        const uint64_t want = (uint64_t)total + total / 4 + 64;
        HIPCHK(hipMalloc((void **)&ws->d_sam, want));
        HIPCHK(hipHostMalloc((void **)&ws->h_sam, want, hipHostMallocDefault));
        ws->sam_cap = want;
    }
    HIPCHK(launch_sam_write(d, n_rec, ws->d_samoff, ws->d_sam, st));
    HIPCHK(hipMemcpyAsync(ws->h_sam, ws->d_sam, total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *sam = ws->h_sam; *sam_bytes = total; *n_pairs = n;
    return SALT_OK;
}
#undef REGROW

// ---------------------------------------------------------------------------------------------
// polish (row N4)
// ---------------------------------------------------------------------------------------------
struct salt_gpu_polish { int device = 0; uint8_t *d_pac = nullptr; uint64_t l_pac = 0; void *d_tabs = nullptr; uint32_t n_blocks = 0; };

extern "C" int salt_gpu_polish_open(int device, const uint8_t *pac, uint64_t l_pac, salt_gpu_polish_t **out)
{
    if (!pac || !out || l_pac == 0) return fail(SALT_E_INVAL, "null argument");
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) return fail(SALT_E_HIP, "no HIP device visible: polish cannot run (there is no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    salt_gpu_polish *p = new salt_gpu_polish();
    p->device = device; p->l_pac = l_pac;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_pac, l_pac / 4 + 8);
    if (e == hipSuccess) e = hipMemcpy(p->d_pac, pac, l_pac / 4 + 1, hipMemcpyHostToDevice);
    p->n_blocks = (uint32_t)prop.multiProcessorCount * 8u;
    if (e == hipSuccess) e = hipMalloc(&p->d_tabs, (uint64_t)p->n_blocks * lv_table_bytes());
    if (e != hipSuccess) { hipFree(p->d_pac); hipFree(p->d_tabs); delete p; return fail(SALT_E_HIP, std::string("polish open: ") + hipGetErrorString(e)); }
    *out = p;
    return SALT_OK;
}
extern "C" void salt_gpu_polish_close(salt_gpu_polish_t *p) { if (!p) return; hipSetDevice(p->device); hipFree(p->d_pac); hipFree(p->d_tabs); delete p; }

extern "C" int salt_gpu_polish_lv(salt_gpu_polish_t *p, const uint8_t *codes, const uint32_t *offs, uint32_t n_reads, const salt_polish_item_t *items,
                                  uint32_t n_items, const uint8_t *pool, uint32_t pool_stride, uint32_t n_pool, int want_cigar,
                                  int32_t *dist, uint16_t *cigars, uint8_t *n_cigar)
{
    if (!p || !codes || !offs || !items || !dist || (want_cigar && (!cigars || !n_cigar)) || (n_pool && !pool)) return fail(SALT_E_INVAL, "null argument");
    if (n_items == 0) return SALT_OK;
    for (uint32_t i = 0; i < n_items; ++i) {                   // shapes the kernel assumes, checked before anything is launched
        const salt_polish_item_t &x = items[i];
        if (x.read >= n_reads) return fail(SALT_E_INVAL, "polish item names a read outside the batch");
        const uint32_t L = offs[x.read + 1] - offs[x.read];
        if (L == 0 || L > SALT_MAX_READ_LEN || x.tlen > L || x.k >= 31) return fail(SALT_E_INVAL, "polish item: read length / window / bound outside the kernel's range");
        if (x.pool == 0xFFFFFFFFu) { if ((uint64_t)x.offset + x.tlen > p->l_pac) return fail(SALT_E_INVAL, "polish item: window beyond the genome"); }
        else if (x.pool >= n_pool || pool_stride < L) return fail(SALT_E_INVAL, "polish item: explicit window outside the pool");
    }
    HIPCHK(hipSetDevice(p->device));
    uint8_t *d_codes = nullptr, *d_pool = nullptr, *d_nc = nullptr; uint32_t *d_offs = nullptr; salt_polish_item_t *d_items = nullptr; int32_t *d_dist = nullptr; uint16_t *d_cig = nullptr;
    const uint64_t bases = offs[n_reads];
    auto done = [&](int rc) { hipFree(d_codes); hipFree(d_pool); hipFree(d_nc); hipFree(d_offs); hipFree(d_items); hipFree(d_dist); hipFree(d_cig); return rc; };
#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return done(fail(SALT_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_))); } while (0)
    PCHK(hipMalloc((void **)&d_codes, bases + 64)); PCHK(hipMemcpy(d_codes, codes, bases, hipMemcpyHostToDevice));
    PCHK(hipMalloc((void **)&d_offs, ((uint64_t)n_reads + 1) * 4)); PCHK(hipMemcpy(d_offs, offs, ((uint64_t)n_reads + 1) * 4, hipMemcpyHostToDevice));
    PCHK(hipMalloc((void **)&d_items, (uint64_t)n_items * sizeof(salt_polish_item_t))); PCHK(hipMemcpy(d_items, items, (uint64_t)n_items * sizeof(salt_polish_item_t), hipMemcpyHostToDevice));
    if (n_pool) { PCHK(hipMalloc((void **)&d_pool, (uint64_t)n_pool * pool_stride + 8)); PCHK(hipMemcpy(d_pool, pool, (uint64_t)n_pool * pool_stride, hipMemcpyHostToDevice)); }
    PCHK(hipMalloc((void **)&d_dist, (uint64_t)n_items * 4));
    if (want_cigar) { PCHK(hipMalloc((void **)&d_cig, (uint64_t)n_items * SALT_MAX_CIGAR_OPS * 2)); PCHK(hipMalloc((void **)&d_nc, n_items)); PCHK(hipMemset(d_nc, 0, n_items)); }
    const uint32_t blocks = n_items < p->n_blocks ? n_items : p->n_blocks;
    launch_polish(p->d_pac, d_codes, d_offs, d_items, n_items, d_pool, pool_stride, want_cigar, d_dist, d_cig, d_nc, p->d_tabs, blocks, nullptr);
    PCHK(hipGetLastError());
    PCHK(hipDeviceSynchronize());
    PCHK(hipMemcpy(dist, d_dist, (uint64_t)n_items * 4, hipMemcpyDeviceToHost));
    if (want_cigar) { PCHK(hipMemcpy(cigars, d_cig, (uint64_t)n_items * SALT_MAX_CIGAR_OPS * 2, hipMemcpyDeviceToHost)); PCHK(hipMemcpy(n_cigar, d_nc, n_items, hipMemcpyDeviceToHost)); }
#undef PCHK
    return done(SALT_OK);
}

extern "C" int salt_gpu_polish_sw(salt_gpu_polish_t *p, const uint8_t *codes, const uint32_t *offs, uint32_t n_reads, const salt_polish_item_t *items,
                                  uint32_t n_items, int want_cigar, int32_t *score, int32_t *read_span, uint16_t *cigars, uint16_t *n_cigar)
{
    if (!p || !codes || !offs || !items || !score || (want_cigar && (!read_span || !cigars || !n_cigar))) return fail(SALT_E_INVAL, "null argument");
    if (n_items == 0) return SALT_OK;
    uint32_t max_len = 1;
    std::vector<PeSwReq> h_req(n_items);
    for (uint32_t i = 0; i < n_items; ++i) {                   // shapes the kernel assumes, checked before anything is launched
        const salt_polish_item_t &x = items[i];
        if (x.read >= n_reads) return fail(SALT_E_INVAL, "polish item names a read outside the batch");
        const uint32_t L = offs[x.read + 1] - offs[x.read];
        if (L == 0 || L > SALT_MAX_READ_LEN || x.tlen == 0 || x.tlen > L) return fail(SALT_E_INVAL, "polish item: read length / window outside the kernel's range");
        if ((uint64_t)x.offset + x.tlen > p->l_pac) return fail(SALT_E_INVAL, "polish item: window beyond the genome");
        if (L > max_len) max_len = L;
        h_req[i] = PeSwReq{ x.offset, x.offset + x.tlen - 1u, x.read, (uint8_t)(x.strand ? 1 : 0), 2, (uint16_t)(want_cigar ? 0 : 1) };
    }
    HIPCHK(hipSetDevice(p->device));
    uint8_t *d_codes = nullptr, *d_scr = nullptr; uint32_t *d_offs = nullptr, *d_ctl = nullptr; PeSwReq *d_req = nullptr; PeSwRes *d_res = nullptr;
    const uint64_t bases = offs[n_reads];
    auto done = [&](int rc) { hipFree(d_codes); hipFree(d_scr); hipFree(d_offs); hipFree(d_ctl); hipFree(d_req); hipFree(d_res); return rc; };
#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return done(fail(SALT_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_))); } while (0)
    PCHK(hipMalloc((void **)&d_codes, bases + 64)); PCHK(hipMemcpy(d_codes, codes, bases, hipMemcpyHostToDevice));
    PCHK(hipMalloc((void **)&d_offs, ((uint64_t)n_reads + 1) * 4)); PCHK(hipMemcpy(d_offs, offs, ((uint64_t)n_reads + 1) * 4, hipMemcpyHostToDevice));
    PCHK(hipMalloc((void **)&d_req, (uint64_t)n_items * sizeof(PeSwReq))); PCHK(hipMemcpy(d_req, h_req.data(), (uint64_t)n_items * sizeof(PeSwReq), hipMemcpyHostToDevice));
    PCHK(hipMalloc((void **)&d_res, (uint64_t)n_items * sizeof(PeSwRes)));
    SwGeom geom = sw_geom(max_len, max_len, p->n_blocks / 8u);
    sw_geom_limit(geom, (n_items + 7u) / 8u);
    PCHK(hipMalloc((void **)&d_scr, sw_scratch_bytes(geom)));
    const uint32_t ctl[9] = { n_items, 0, 0, 0, 0, 0, 0, 0, 0 };   // requests, -, overflow count, -, the four kernels' queue heads + pending count
    PCHK(hipMalloc((void **)&d_ctl, 36)); PCHK(hipMemcpy(d_ctl, ctl, 36, hipMemcpyHostToDevice));
    IndexView v; memset(&v, 0, sizeof v);
    v.ref_len = (uint32_t)p->l_pac;                            // k_sw's range check; mode 2 reads the 2-bit genome only
    launch_sw(v, p->d_pac, d_codes, d_offs, d_req, d_ctl, d_res, d_ctl + 4, d_ctl + 2, d_scr, geom, max_len, nullptr);
    PCHK(hipGetLastError());
    PCHK(hipDeviceSynchronize());
    uint32_t h_ctl[3];
    PCHK(hipMemcpy(h_ctl, d_ctl, 12, hipMemcpyDeviceToHost));
    if (h_ctl[2]) return done(fail(SALT_E_INVAL, "polish -s: an alignment needs a wider band or more CIGAR operations than this build holds"));
    std::vector<PeSwRes> h_res(n_items);
    PCHK(hipMemcpy(h_res.data(), d_res, (uint64_t)n_items * sizeof(PeSwRes), hipMemcpyDeviceToHost));
#undef PCHK
    for (uint32_t i = 0; i < n_items; ++i) {
        const PeSwRes &r = h_res[i];
        score[i] = r.score1;
        if (want_cigar) {
            read_span[2 * (uint64_t)i] = r.read_begin; read_span[2 * (uint64_t)i + 1] = r.read_end;
            n_cigar[i] = r.n_cigar;
            memcpy(cigars + (uint64_t)i * SALT_MAX_CIGAR_OPS, r.cigar, sizeof r.cigar);
        }
    }
    return done(SALT_OK);
}

extern "C" int salt_gpu_index_replicate(salt_gpu_index_t *src, const int *devices, int n, salt_gpu_index_t **out)
{
    if (!src || !devices || !out || n < 1 || devices[0] != src->device) return fail(SALT_E_INVAL, "bad replicate arguments");
    out[0] = src;
    for (int i = 1; i < n; ++i) out[i] = nullptr;
    if (n == 1) return SALT_OK;
    std::vector<void *> buf((size_t)n, nullptr);
    std::vector<ncclComm_t> comm((size_t)n, nullptr);
    std::vector<hipStream_t> st((size_t)n, nullptr);
    bool comm_ok = false;
    // every exit passes here: what a failed step leaves behind (buffers not yet owned by an index, communicators, streams) is released
    auto cleanup = [&](bool failed) {
        for (int i = 0; i < n; ++i) {
            hipSetDevice(devices[i]);
            if (st[i]) hipStreamDestroy(st[i]);
            if (comm_ok && comm[i]) ncclCommDestroy(comm[i]);
            if (failed && i > 0) {
                if (out[i]) { salt_gpu_index_detach(out[i]); out[i] = nullptr; }       // owns buf[i]
                else if (buf[i]) hipFree(buf[i]);
            }
        }
        hipSetDevice(devices[0]);
    };
#define REPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(true); \
    return fail(SALT_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    buf[0] = src->image;
    for (int i = 1; i < n; ++i) { REPCHK(hipSetDevice(devices[i])); REPCHK(hipMalloc(&buf[i], src->bytes)); }
    ncclResult_t rc = ncclCommInitAll(comm.data(), n, devices);
    if (rc != ncclSuccess) { cleanup(true); return fail(SALT_E_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(rc)); }
    comm_ok = true;
    for (int i = 0; i < n; ++i) { REPCHK(hipSetDevice(devices[i])); REPCHK(hipStreamCreate(&st[i])); }
    ncclGroupStart();
    for (int i = 0; i < n; ++i) {
        rc = ncclBroadcast(buf[i], buf[i], src->hdr.off_wlkt, ncclUint8, 0, comm[i], st[i]);      // the compact part only
        if (rc != ncclSuccess) { ncclGroupEnd(); cleanup(true); return fail(SALT_E_HIP, std::string("ncclBroadcast: ") + ncclGetErrorString(rc)); }
    }
    rc = ncclGroupEnd();
    if (rc != ncclSuccess) { cleanup(true); return fail(SALT_E_HIP, std::string("ncclGroupEnd: ") + ncclGetErrorString(rc)); }
    for (int i = 0; i < n; ++i) { REPCHK(hipSetDevice(devices[i])); REPCHK(hipStreamSynchronize(st[i])); }
    for (int i = 1; i < n; ++i) {
        int r2 = salt_gpu_index_attach_image(buf[i], src->bytes, devices[i], &out[i]);
        if (r2) { out[i] = nullptr; const std::string m = g_err; cleanup(true); return fail(r2, m); }
        out[i]->owns = true;                                        // from here on detach releases buf[i]
        REPCHK(hipSetDevice(devices[i]));
        r2 = rebuild_wlkt(out[i]);                                  // each device tabulates its own W-mer table
        if (r2) { const std::string m = g_err; cleanup(true); return fail(r2, m); }
    }
#undef REPCHK
    cleanup(false);
    return SALT_OK;
}

extern "C" int salt_gpu_index_image_copy(const salt_gpu_index_t *ix, void *dst, uint64_t dst_bytes)
{
    if (!ix || !dst) return fail(SALT_E_INVAL, "null argument");
    const uint64_t n = dst_bytes >= ix->bytes ? ix->bytes : ix->hdr.off_wlkt;       // the full image, or its compact part
    if (dst_bytes < n) return fail(SALT_E_INVAL, "destination too small for the (compact) index image");
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipMemcpy(dst, ix->image, n, hipMemcpyDeviceToDevice));
    return SALT_OK;
}

extern "C" int salt_gpu_ws_timing(salt_gpu_ws_t *ws, int enable)
{
    if (!ws) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipSetDevice(ws->ix->device));
    if (enable && ws->ev.empty()) {
        ws->ev.resize((size_t)MAX_TIMED * EV_PER_CALL);
        ws->ev_pe.assign(MAX_TIMED, 0);
        for (auto &e : ws->ev) HIPCHK(hipEventCreate(&e));
    }
    ws->timing = enable != 0; ws->n_timed = 0;
    return SALT_OK;
}

extern "C" int salt_gpu_ws_kernel_ms(salt_gpu_ws_t *ws, double ms[SALT_N_KERNELS], uint32_t *n_calls)
{
    if (!ws || !ms || !n_calls) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipSetDevice(ws->ix->device));
    for (int k = 0; k < SALT_N_KERNELS; ++k) ms[k] = 0;
    *n_calls = ws->n_timed;
    for (uint32_t i = 0; i < ws->n_timed; ++i) {
        hipEvent_t *ev = &ws->ev[(size_t)i * EV_PER_CALL];
        const int n_k = ws->ev_pe[i] ? SALT_N_KERNELS : 7;
        HIPCHK(hipEventSynchronize(ev[n_k]));
        for (int k = 0; k < n_k; ++k) {
            float a = 0;
            HIPCHK(hipEventElapsedTime(&a, ev[k], ev[k + 1]));
            ms[k] += a;
        }
    }
    ws->n_timed = 0;
    return SALT_OK;
}

// Unit access for tests: verify / LV device functions on a caller-supplied mixRef (no index needed).
extern "C" int salt_gpu_diag_lv(const uint32_t *ref_words, uint32_t ref_len, uint32_t n_cases, const uint32_t *pos,
                                const uint32_t *kdiff, const uint8_t *seqs, const uint32_t *offs, int32_t *out4,
                                uint16_t *cigars)
{
    if (!ref_words || !pos || !kdiff || !seqs || !offs || !out4 || !cigars) return fail(SALT_E_INVAL, "null argument");
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) return fail(SALT_E_HIP, "no HIP device visible");
    const uint64_t nw = ((uint64_t)ref_len + 7) / 8 + 4;
    uint32_t *d_ref = nullptr, *d_pos = nullptr, *d_k = nullptr, *d_offs = nullptr; uint8_t *d_seqs = nullptr;
    int32_t *d_out = nullptr; uint16_t *d_cig = nullptr;
    const uint64_t bases = offs[n_cases];
    HIPCHK(hipMalloc((void **)&d_ref, nw * 4)); HIPCHK(hipMemset(d_ref, 0, nw * 4));
    HIPCHK(hipMemcpy(d_ref, ref_words, (nw - 4) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_pos, (uint64_t)n_cases * 4)); HIPCHK(hipMemcpy(d_pos, pos, (uint64_t)n_cases * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_k, (uint64_t)n_cases * 4)); HIPCHK(hipMemcpy(d_k, kdiff, (uint64_t)n_cases * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_offs, ((uint64_t)n_cases + 1) * 4)); HIPCHK(hipMemcpy(d_offs, offs, ((uint64_t)n_cases + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_seqs, bases + 64)); HIPCHK(hipMemcpy(d_seqs, seqs, bases, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_out, (uint64_t)n_cases * 16));
    HIPCHK(hipMalloc((void **)&d_cig, (uint64_t)n_cases * SALT_MAX_CIGAR_OPS * 2)); HIPCHK(hipMemset(d_cig, 0, (uint64_t)n_cases * SALT_MAX_CIGAR_OPS * 2));
    IndexView v; memset(&v, 0, sizeof v);
    v.ref = d_ref; v.ref_len = ref_len;
    void *d_tab = nullptr;
    HIPCHK(hipMalloc(&d_tab, (uint64_t)n_cases * lv_table_bytes()));
    launch_diag_lv(v, n_cases, d_pos, d_k, d_seqs, d_offs, d_out, d_cig, d_tab, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out4, d_out, (uint64_t)n_cases * 16, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cigars, d_cig, (uint64_t)n_cases * SALT_MAX_CIGAR_OPS * 2, hipMemcpyDeviceToHost));
    hipFree(d_ref); hipFree(d_pos); hipFree(d_k); hipFree(d_offs); hipFree(d_seqs); hipFree(d_out); hipFree(d_cig); hipFree(d_tab);
    return SALT_OK;
}

// Unit entry of the Smith-Waterman rescue kernel: case i aligns read codes[read_offs[i]..) against the reference symbols
// ref_syms[ref_offs[i]..) (4-bit allele masks when aware[i], bases 0..3 otherwise), the way snpaln_sw_snpaware / snpaln_sw
// call ssw_init + ssw_align (alnpe.c:260-393).  out6: score1, score2, ref_begin, ref_end, read_begin, read_end.
extern "C" int salt_gpu_diag_ssw(uint32_t n_cases, const uint8_t *aware, const uint8_t *ref_syms, const uint32_t *ref_offs,
                                 const uint8_t *codes, const uint32_t *read_offs, int32_t *out6, uint16_t *cigars, uint16_t *n_cigar)
{
    if (!aware || !ref_syms || !ref_offs || !codes || !read_offs || !out6 || !cigars || !n_cigar) return fail(SALT_E_INVAL, "null argument");
    if (n_cases == 0) return SALT_OK;
    const uint64_t n_sym = ref_offs[n_cases], n_base = read_offs[n_cases];
    std::vector<uint32_t> h_ref(n_sym / 8 + 8, 0u);
    std::vector<uint8_t> h_pac(n_sym / 4 + 8, 0);
    std::vector<PeSwReq> h_req(n_cases);
    uint32_t diag_max_len = 1;
    for (uint32_t i = 0; i < n_cases; ++i) {
        if (ref_offs[i + 1] <= ref_offs[i] || read_offs[i + 1] <= read_offs[i]) return fail(SALT_E_INVAL, "empty case");
        for (uint64_t p = ref_offs[i]; p < ref_offs[i + 1]; ++p) {
            h_ref[p >> 3] |= (uint32_t)(ref_syms[p] & 15u) << (4 * (p & 7u));
            h_pac[p >> 2] |= (uint8_t)((ref_syms[p] & 3u) << ((~p & 3u) << 1));
        }
        h_req[i] = PeSwReq{ ref_offs[i], ref_offs[i + 1] - 1, i, 0, (uint8_t)(aware[i] ? 1 : 0), 0 };
        if (read_offs[i + 1] - read_offs[i] > diag_max_len) diag_max_len = read_offs[i + 1] - read_offs[i];
    }
    uint32_t *d_ref = nullptr, *d_offs = nullptr, *d_ctl = nullptr; uint8_t *d_pac = nullptr, *d_codes = nullptr, *d_scr = nullptr;
    PeSwReq *d_req = nullptr; PeSwRes *d_res = nullptr;
    const uint32_t blocks = n_cases / 8 + 1 < 256 ? n_cases / 8 + 1 : 256;
    HIPCHK(hipMalloc((void **)&d_ref, h_ref.size() * 4)); HIPCHK(hipMemcpy(d_ref, h_ref.data(), h_ref.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_pac, h_pac.size())); HIPCHK(hipMemcpy(d_pac, h_pac.data(), h_pac.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_codes, n_base + 16)); HIPCHK(hipMemcpy(d_codes, codes, n_base, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_offs, ((uint64_t)n_cases + 1) * 4)); HIPCHK(hipMemcpy(d_offs, read_offs, ((uint64_t)n_cases + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_req, (uint64_t)n_cases * sizeof(PeSwReq))); HIPCHK(hipMemcpy(d_req, h_req.data(), (uint64_t)n_cases * sizeof(PeSwReq), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_res, (uint64_t)n_cases * sizeof(PeSwRes)));
    uint64_t diag_max_win = 1;
    for (uint32_t i = 0; i < n_cases; ++i) if (ref_offs[i + 1] - ref_offs[i] > diag_max_win) diag_max_win = ref_offs[i + 1] - ref_offs[i];
    if (diag_max_len > SALT_MAX_READ_LEN) return fail(SALT_E_INVAL, "read longer than SALT_MAX_READ_LEN");
    SwGeom geom = sw_geom(diag_max_len, diag_max_win, 1);
    geom.n_blocks = blocks; geom.tb_blocks = blocks;
    HIPCHK(hipMalloc((void **)&d_scr, sw_scratch_bytes(geom)));
    const uint32_t ctl[9] = { n_cases, 0, 0, 0, 0, 0, 0, 0, 0 };    // requests, -, overflow count, -, the four kernels' queue heads + pending count
    HIPCHK(hipMalloc((void **)&d_ctl, 36)); HIPCHK(hipMemcpy(d_ctl, ctl, 36, hipMemcpyHostToDevice));
    IndexView v; memset(&v, 0, sizeof v);
    v.ref = d_ref; v.ref_len = (uint32_t)n_sym;
    launch_sw(v, d_pac, d_codes, d_offs, d_req, d_ctl, d_res, d_ctl + 4, d_ctl + 2, d_scr, geom, diag_max_len, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    std::vector<PeSwRes> h_res(n_cases);
    HIPCHK(hipMemcpy(h_res.data(), d_res, (uint64_t)n_cases * sizeof(PeSwRes), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_cases; ++i) {
        const PeSwRes &r = h_res[i];
        int32_t *o = out6 + 6 * (uint64_t)i;
        o[0] = r.score1; o[1] = r.score2; o[2] = r.ref_begin; o[3] = r.ref_end; o[4] = r.read_begin; o[5] = r.read_end;
        n_cigar[i] = r.n_cigar;
        memcpy(cigars + (uint64_t)i * SALT_MAX_CIGAR_OPS, r.cigar, sizeof r.cigar);
    }
    hipFree(d_ref); hipFree(d_pac); hipFree(d_codes); hipFree(d_offs); hipFree(d_req); hipFree(d_res); hipFree(d_scr); hipFree(d_ctl);
    return SALT_OK;
}

extern "C" int salt_gpu_buffer_alloc(int device, uint64_t bytes, void **dev_ptr)
{
    if (!dev_ptr || !bytes) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMalloc(dev_ptr, bytes));
    return SALT_OK;
}

extern "C" int salt_gpu_buffer_free(int device, void *dev_ptr)
{
    if (!dev_ptr) return SALT_OK;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(dev_ptr));
    return SALT_OK;
}

__global__ void k_buffer_diff(const uint32_t *a, const uint32_t *b, uint64_t n_words, unsigned long long *n_diff)
{
    unsigned long long d = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) d += a[i] != b[i];
    if (d) atomicAdd(n_diff, d);
}

extern "C" int salt_gpu_buffer_equal(int device, const void *a, const void *b, uint64_t bytes, int *equal)
{
    if (!a || !b || !equal || (bytes & 3u)) return fail(SALT_E_INVAL, "bad buffer compare arguments (bytes must be a multiple of 4)");
    HIPCHK(hipSetDevice(device));
    unsigned long long *d = nullptr, h = 0;
    HIPCHK(hipMalloc((void **)&d, 8));
    HIPCHK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(k_buffer_diff, dim3(4096), dim3(256), 0, nullptr, static_cast<const uint32_t *>(a), static_cast<const uint32_t *>(b), bytes / 4, d);
    hipError_t e = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(SALT_E_HIP, std::string("buffer compare: ") + hipGetErrorString(e));
    *equal = h == 0;
    return SALT_OK;
}

// Unit access to the candidate rule (rule_unsorted / rule_sparse): see include/salt_gpu.h
extern "C" int salt_gpu_diag_rule(uint32_t n_cases, const uint32_t *pos, const uint8_t *val, const uint32_t *offs, const uint32_t *bound_in,
                                  uint32_t L, uint32_t ref_len, int mode, uint32_t *out)
{
    if (!pos || !val || !offs || !bound_in || !out) return fail(SALT_E_INVAL, "null argument");
    if (n_cases == 0) return SALT_OK;
    const uint64_t n = offs[n_cases];
    uint32_t *d_pos = nullptr, *d_offs = nullptr, *d_b = nullptr, *d_out = nullptr; uint8_t *d_val = nullptr;
    const uint64_t ow = (uint64_t)n_cases * diag_rule_words();
    HIPCHK(hipMalloc((void **)&d_pos, (n + 64) * 4)); HIPCHK(hipMemcpy(d_pos, pos, n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_val, n + 64)); HIPCHK(hipMemcpy(d_val, val, n, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_offs, ((uint64_t)n_cases + 1) * 4)); HIPCHK(hipMemcpy(d_offs, offs, ((uint64_t)n_cases + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_b, (uint64_t)n_cases * 4)); HIPCHK(hipMemcpy(d_b, bound_in, (uint64_t)n_cases * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_out, ow * 4));
    launch_diag_rule(n_cases, d_pos, d_val, d_offs, d_b, L, ref_len, mode, d_out, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d_out, ow * 4, hipMemcpyDeviceToHost));
    hipFree(d_pos); hipFree(d_val); hipFree(d_offs); hipFree(d_b); hipFree(d_out);
    return SALT_OK;
}

// Unit entry of the candidate verifiers on a caller-supplied mixRef (see k_diag_verify): fault-free check of the guards that keep a
// wrapped locate from being used as an address.
extern "C" int salt_gpu_diag_verify(const uint32_t *ref_words, uint32_t ref_len, uint32_t n_cases, const uint8_t *seqs, const uint32_t *offs,
                                    const uint32_t *cand, const uint32_t *cand_offs, int mode, uint8_t *out)
{
    if (!ref_words || !seqs || !offs || !cand || !cand_offs || !out) return fail(SALT_E_INVAL, "null argument");
    if (mode < 0 || mode > 4) return fail(SALT_E_INVAL, "mode must be 0..4");
    if (n_cases == 0) return SALT_OK;
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) return fail(SALT_E_HIP, "no HIP device visible");
    for (uint32_t i = 0; i < n_cases; ++i) {
        const uint32_t L = offs[i + 1] - offs[i];
        if (L == 0 || L > SALT_MAX_READ_LEN || ((mode == 1 || mode == 3) && L > 120) || ((mode == 2 || mode == 4) && L > 248)) return fail(SALT_E_INVAL, "read length outside the verifier's range");
        if (cand_offs[i + 1] - cand_offs[i] > 256) return fail(SALT_E_INVAL, "at most 256 candidates per case");
    }
    const uint64_t nw = ((uint64_t)ref_len + 7) / 8 + 36, bases = offs[n_cases], nc = cand_offs[n_cases];
    uint32_t *d_ref = nullptr, *d_offs = nullptr, *d_cand = nullptr, *d_coffs = nullptr; uint8_t *d_seqs = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_ref, nw * 4)); HIPCHK(hipMemset(d_ref, 0, nw * 4));
    HIPCHK(hipMemcpy(d_ref, ref_words, (nw - 36) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_offs, ((uint64_t)n_cases + 1) * 4)); HIPCHK(hipMemcpy(d_offs, offs, ((uint64_t)n_cases + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_coffs, ((uint64_t)n_cases + 1) * 4)); HIPCHK(hipMemcpy(d_coffs, cand_offs, ((uint64_t)n_cases + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_cand, (nc + 1) * 4)); HIPCHK(hipMemcpy(d_cand, cand, nc * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_seqs, bases + 64)); HIPCHK(hipMemcpy(d_seqs, seqs, bases, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&d_out, nc + 1));
    launch_diag_verify(d_ref, ref_len, n_cases, d_seqs, d_offs, d_cand, d_coffs, mode, d_out, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d_out, nc, hipMemcpyDeviceToHost));
    hipFree(d_ref); hipFree(d_offs); hipFree(d_coffs); hipFree(d_cand); hipFree(d_seqs); hipFree(d_out);
    return SALT_OK;
}

extern "C" int salt_gpu_ws_queue_counts(salt_gpu_ws_t *ws, uint32_t out[8])
{
    if (!ws || !out) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipSetDevice(ws->ix->device));
    HIPCHK(hipDeviceSynchronize());
    uint32_t all[QCTL_WORDS];
    HIPCHK(hipMemcpy(all, ws->d_qctl, sizeof all, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < 8; ++k) out[k] = all[QC(k)];
    return SALT_OK;
}

extern "C" int salt_gpu_ws_heavy_reads(salt_gpu_ws_t *ws, uint32_t *ids, uint32_t cap, uint32_t *n)
{
    if (!ws || !n) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipSetDevice(ws->ix->device));
    HIPCHK(hipDeviceSynchronize());
    uint32_t ctl[1];
    HIPCHK(hipMemcpy(ctl, ws->d_qctl + QC(0), 4, hipMemcpyDeviceToHost));
    *n = ctl[0];
    if (ids && cap) HIPCHK(hipMemcpy(ids, ws->d_queue, (uint64_t)(ctl[0] < cap ? ctl[0] : cap) * 4, hipMemcpyDeviceToHost));
    return SALT_OK;
}

extern "C" int salt_gpu_index_set_pac(salt_gpu_index_t *ix, const uint8_t *pac, uint64_t l_pac)
{
    if (!ix || !pac || l_pac == 0) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipSetDevice(ix->device));
    if (ix->d_pac) { hipFree(ix->d_pac); ix->d_pac = nullptr; }
    const uint64_t bytes = l_pac / 4 + 2;
    HIPCHK(hipMalloc((void **)&ix->d_pac, bytes + 16));
    HIPCHK(hipMemset(ix->d_pac, 0, bytes + 16));
    HIPCHK(hipMemcpy(ix->d_pac, pac, bytes, hipMemcpyHostToDevice));
    ix->l_pac = l_pac;
    // Paired end: context records for the R rows as well (16 B per row of the R index, 20 GiB at GRCh38 scale), when the C rows have theirs
    // and the device has the room; SALT_GPU_NO_RCTX=1 leaves them out.  Results do not depend on it (ctx_reject is a lower bound).
    if (ix->view.c_ctx && !ix->d_rctx && !(getenv("SALT_GPU_NO_RCTX") && atoi(getenv("SALT_GPU_NO_RCTX")))) {
        const uint64_t rbytes = ((uint64_t)ix->hdr.r_text_len + 1) * 16;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= rbytes + (32ull << 30) && hipMalloc((void **)&ix->d_rctx, rbytes) == hipSuccess) {
            launch_build_r_ctx(ix->view, ix->hdr.ctx_k, ix->d_rctx, nullptr);
            if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { hipFree(ix->d_rctx); ix->d_rctx = nullptr; return fail(SALT_E_HIP, "building the R context records failed"); }
            ix->view.r_ctx = ix->d_rctx;
        } else ix->d_rctx = nullptr;
    }
    return SALT_OK;
}

extern "C" int salt_gpu_ws_pe_overflow(salt_gpu_ws_t *ws, uint32_t *n)
{
    if (!ws || !n) return fail(SALT_E_INVAL, "null argument");
    *n = 0;
    if (!ws->d_pctl) return SALT_OK;
    HIPCHK(hipSetDevice(ws->ix->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(n, ws->d_pctl + 4, 4, hipMemcpyDeviceToHost));
    return SALT_OK;
}

extern "C" int salt_gpu_ws_pe_counts(salt_gpu_ws_t *ws, uint32_t out[8])
{
    if (!ws || !out) return fail(SALT_E_INVAL, "null argument");
    memset(out, 0, 32);
    if (!ws->d_pctl) return SALT_OK;
    HIPCHK(hipSetDevice(ws->ix->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, ws->d_pctl, 32, hipMemcpyDeviceToHost));
    return SALT_OK;
}

static int pe_prepare(salt_gpu_ws_t *ws, uint32_t n_pairs, hipStream_t st)
{
    if (n_pairs > ws->pe_pairs_cap) {
        HIPCHK(hipStreamSynchronize(st));
        hipFree(ws->d_pairs); hipFree(ws->d_req); hipFree(ws->d_swres); hipFree(ws->d_pcq);
        ws->d_pairs = nullptr; ws->d_req = nullptr; ws->d_swres = nullptr; ws->d_pcq = nullptr; ws->pe_pairs_cap = 0;   // a failed malloc below leaves a consistent (empty) state
        HIPCHK(hipMalloc((void **)&ws->d_pcq, (uint64_t)n_pairs * 2 * 4));
        HIPCHK(hipMalloc((void **)&ws->d_pairs, (uint64_t)n_pairs * sizeof(PePair)));
        HIPCHK(hipMalloc((void **)&ws->d_req, (uint64_t)n_pairs * 2 * sizeof(PeSwReq)));
        HIPCHK(hipMalloc((void **)&ws->d_swres, (uint64_t)n_pairs * 2 * sizeof(PeSwRes)));
        ws->pe_pairs_cap = n_pairs;
    }
    if (!ws->d_pctl) {
        HIPCHK(hipMalloc((void **)&ws->d_pctl, 20 * 4));         // [0..7] see pe_resident_impl; [8..11] k_swtb's phase clocks (diagnostics build); [12..15] the queue heads of k_swf, k_swf1, k_swr, k_swtb, [16] k_swf1's request count
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, ws->ix->device));
        ws->sw_blocks = (uint32_t)prop.multiProcessorCount;             // CUs: k_sw runs up to SW_MAX_BLOCKS_PER_CU blocks on each
    }
    return SALT_OK;
}

// everything of alnpe_core1 on device-resident buffers; only enqueues on st
static int pe_resident_impl(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_pe_opt_t *pe, uint32_t n_pairs, uint32_t max_len,
                            const void *d_seqs, const void *d_offs, void *d_results, hipStream_t st)
{
    int rc = pe_prepare(ws, n_pairs, st);
    if (rc) return rc;
    const uint32_t ti = ws->n_timed;
    rc = align_resident_impl(ws, o, 2 * n_pairs, max_len, d_seqs, d_offs, d_results, st, 1);
    if (rc) return rc;
    hipEvent_t *ev = ws->n_timed == ti + 1 ? &ws->ev[(size_t)ti * EV_PER_CALL] : nullptr;      // the call above was timed: three more events
    HIPCHK(hipMemsetAsync(ws->d_pctl, 0, 80, st));
    launch_pair(n_pairs, pe->min_tlen, pe->max_tlen, (uint32_t)ws->ix->l_pac, static_cast<const uint32_t *>(d_offs), static_cast<salt_result_t *>(d_results),
                ws->d_pairs, ws->d_req, ws->d_pctl, st);
    if (ev) HIPCHK(hipEventRecord(ev[8], st));
    // rescue windows are as long as the insert-size window plus a mate (alnpe.c:213-252, 395-480), whatever -a / -b say
    const uint64_t l_pac = (uint64_t)ws->ix->l_pac;
    uint64_t max_win = (uint64_t)pe->max_tlen + max_len + 2;
    if (max_win > l_pac + 1) max_win = l_pac + 1;
    SwGeom geom = sw_geom(max_len, max_win, ws->sw_blocks);
    // one group (8 lanes) per rescue in flight; rescues are a few per cent of the mates, so a small batch does not need the full grid
    sw_geom_limit(geom, n_pairs / 96u < 256u ? 256u : n_pairs / 96u);
    const uint64_t need = sw_scratch_bytes(geom);
    if (need > ws->sw_scr_bytes) {
        HIPCHK(hipStreamSynchronize(st));
        hipFree(ws->d_sw_scr); ws->d_sw_scr = nullptr; ws->sw_scr_bytes = 0;
        hipError_t e = hipMalloc((void **)&ws->d_sw_scr, need);
        if (e != hipSuccess) return fail(SALT_E_NOMEM, std::string("hipMalloc(rescue scratch): ") + hipGetErrorString(e));
        ws->sw_scr_bytes = need;
    }
    launch_sw(ws->ix->view, ws->ix->d_pac, static_cast<const uint8_t *>(d_seqs), static_cast<const uint32_t *>(d_offs), ws->d_req, ws->d_pctl, ws->d_swres,
              ws->d_pctl + 12, ws->d_pctl + 4, ws->d_sw_scr, geom, max_len, st);   // pctl: requests, -, CIGAR items + head, overflow, (diagnostics); [12..15] the Smith-Waterman kernels' queue heads
    if (ev) HIPCHK(hipEventRecord(ev[9], st));
    launch_pe_final(ws->ix->view, PackGeom::make(max_len), n_pairs, ws->d_pm, static_cast<salt_result_t *>(d_results), ws->d_pairs, ws->d_swres, ws->d_lvtab,
                    ws->d_pcq, ws->d_pctl + 2, ws->heavy_blocks, st);
    if (ev) { HIPCHK(hipEventRecord(ev[10], st)); ws->ev_pe[ti] = 1; }
#ifdef SALT_DIAG
    if (getenv("SALT_GPU_TB_CLOCKS")) {
        uint32_t c[16]; HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipMemcpy(c, ws->d_pctl, 64, hipMemcpyDeviceToHost));
        if (c[11]) fprintf(stderr, "[k_swtb] %u tracebacks: operands %.1f us, band passes %.1f us, walk + write %.1f us each (s_memtime, 10 ns ticks)\n", c[11],
                           c[8] / 100.0 / c[11], c[9] / 100.0 / c[11], c[10] / 100.0 / c[11]);
    }
#endif
    HIPCHK(hipGetLastError());
    return SALT_OK;
}

extern "C" int salt_gpu_align_pe_resident(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_pe_opt_t *pe, uint32_t n_pairs,
                                          uint32_t max_read_len, const void *d_seqs, const void *d_offs, void *d_results, void *hip_stream)
{
    if (!ws || !o || !pe || !d_seqs || !d_offs || !d_results) return fail(SALT_E_INVAL, "null argument");
    if (n_pairs == 0) return SALT_OK;
    if (2ull * n_pairs > ws->max_reads) return fail(SALT_E_CAPACITY, "more mates than the workspace holds");
    if (!ws->ix->d_pac) return fail(SALT_E_INVAL, "paired end needs the 2-bit genome: call salt_gpu_index_set_pac first");
    HIPCHK(hipSetDevice(ws->ix->device));
    return pe_resident_impl(ws, o, pe, n_pairs, max_read_len, d_seqs, d_offs, d_results, static_cast<hipStream_t>(hip_stream));
}

extern "C" int salt_gpu_align_pe(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_pe_opt_t *pe, uint32_t n_pairs,
                                 const uint8_t *seqs, const uint32_t *offs, salt_result_t *results)
{
    if (!ws || !o || !pe || !seqs || !offs || !results) return fail(SALT_E_INVAL, "null argument");
    if (n_pairs == 0) return SALT_OK;
    const uint32_t n_reads = 2 * n_pairs;
    if (n_reads > ws->max_reads) return fail(SALT_E_CAPACITY, "more mates than the workspace holds");
    if (!ws->ix->d_pac) return fail(SALT_E_INVAL, "paired end needs the 2-bit genome: call salt_gpu_index_set_pac first");
    if (offs[0] != 0) return fail(SALT_E_INVAL, "offs[0] must be 0");
    const uint64_t bases = offs[n_reads];
    if (bases > ws->max_bases) return fail(SALT_E_CAPACITY, "more bases than the workspace holds");
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n_reads; ++i) {
        if (offs[i + 1] <= offs[i]) return fail(SALT_E_INVAL, "empty read or decreasing offsets");
        const uint32_t l = offs[i + 1] - offs[i];
        max_len = l > max_len ? l : max_len;
    }
    HIPCHK(hipSetDevice(ws->ix->device));
    hipStream_t st = ws->stream;
    HIPCHK(hipMemcpyAsync(ws->d_seqs, seqs, bases, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ws->d_offs, offs, ((uint64_t)n_reads + 1) * 4, hipMemcpyHostToDevice, st));
    int rc = pe_resident_impl(ws, o, pe, n_pairs, max_len, ws->d_seqs, ws->d_offs, ws->d_results, st);
    if (rc) return rc;
    rc = fetch_results(ws, n_reads, results, st);
    if (rc) return rc;
    uint32_t n_over = 0;
    rc = salt_gpu_ws_pe_overflow(ws, &n_over);
    if (rc) return rc;
    if (n_over) return fail(SALT_E_CAPACITY, std::to_string(n_over) + " mate rescue(s) need a Smith-Waterman band wider than this build holds (SW_BAND_W): "
                                             "the rows of this batch would differ from the reference's");
    return SALT_OK;
}

extern "C" int salt_gpu_ws_counters(salt_gpu_ws_t *ws, uint64_t out[SALT_CTR_N])
{
    if (!ws || !out) return fail(SALT_E_INVAL, "null argument");
    HIPCHK(hipSetDevice(ws->ix->device));
    HIPCHK(hipDeviceSynchronize());
    unsigned long long tmp[SALT_CTR_N];
    HIPCHK(hipMemcpy(tmp, ws->d_ctr, sizeof tmp, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(ws->d_ctr, 0, sizeof tmp));
    for (int i = 0; i < SALT_CTR_N; ++i) out[i] = tmp[i];
    return SALT_OK;
}
