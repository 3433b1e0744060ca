/* tests/stub/salt_gpu_stub.c -- TEST INFRASTRUCTURE: a stand-in for libsalt_gpu.so without a GPU, so that the host side of the `salt`
 * binary (option handling, dealing of chunks to the workers of several "GPUs", ordered output) can be tested in the build container.
 * Device d is a label; the per-batch work is done by the CPU oracle (oracle/salt_oracle.c, linked in) and its own SAM writer.
 * The entry points `salt` calls are real (text path and host pipeline); the others are absent.  Never shipped, never loaded by salt_amd/. */
#include "../../include/salt_gpu.h"
#include "../../oracle/salt_oracle.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct salt_gpu_index { int device; so_index_t *ora; int owner; };
struct salt_gpu_ws { struct salt_gpu_index *ix; char *sam; size_t cap; };
static __thread char g_err[256];
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

const char *salt_gpu_last_error(void) { return g_err; }
static int fail(const char *m) { snprintf(g_err, sizeof g_err, "stub: %s", m); return SALT_E_INVAL; }

int salt_gpu_index_attach(const salt_host_index_t *h, int device, salt_gpu_index_t **out)
{
    (void)h;
    const char *p = getenv("SALT_STUB_PREFIX");
    if (!p) return fail("SALT_STUB_PREFIX is not set");
    struct salt_gpu_index *ix = calloc(1, sizeof *ix);
    ix->device = device; ix->ora = so_index_load(p); ix->owner = 1;
    if (!ix->ora) { free(ix); return fail("cannot load the index for the oracle"); }
    *out = ix;
    return SALT_OK;
}
int salt_gpu_index_replicate(salt_gpu_index_t *src, const int *devices, int n, salt_gpu_index_t **out)
{
    out[0] = src;
    for (int i = 1; i < n; ++i) { struct salt_gpu_index *ix = calloc(1, sizeof *ix); ix->device = devices[i]; ix->ora = src->ora; ix->owner = 0; out[i] = ix; }
    return SALT_OK;
}
void salt_gpu_index_detach(salt_gpu_index_t *ix) { if (!ix) return; if (ix->owner) so_index_free(ix->ora); free(ix); }
int salt_gpu_index_set_pac(salt_gpu_index_t *ix, const uint8_t *pac, uint64_t l_pac) { (void)ix; (void)pac; (void)l_pac; return SALT_OK; }
int salt_gpu_index_set_contigs(salt_gpu_index_t *ix, int32_t n, const int64_t *offsets, const char *const *names) { (void)ix; (void)n; (void)offsets; (void)names; return SALT_OK; }
int salt_gpu_ws_create(salt_gpu_index_t *ix, uint32_t max_reads, uint64_t max_bases, salt_gpu_ws_t **out)
{
    (void)max_reads; (void)max_bases;
    struct salt_gpu_ws *ws = calloc(1, sizeof *ws); ws->ix = ix; *out = ws; return SALT_OK;
}
int salt_gpu_ws_reserve_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, uint64_t b, uint32_t r, uint32_t l, uint64_t sb, void *hs, uint64_t hb)
{ (void)ws; (void)o; (void)b; (void)r; (void)l; (void)sb; (void)hs; (void)hb; return SALT_OK; }
void salt_gpu_ws_destroy(salt_gpu_ws_t *ws) { if (ws) { free(ws->sam); free(ws); } }
int salt_gpu_device_numa_node(int device, int *node) { (void)device; *node = -1; return SALT_OK; }
int salt_gpu_host_alloc(uint64_t bytes, void **ptr) { *ptr = malloc(bytes); return *ptr ? SALT_OK : SALT_E_NOMEM; }
void salt_gpu_host_free(void *ptr) { free(ptr); }
/* so_result_t -> salt_result_t: binary CIGARs; the CIGARs of gapped alternative hits (the device delivers them for sam_add_xa, sam.c:216-225)
 * by the oracle's Landau-Vishkin traceback on the hit's strand */
static int cigar_bin(const char *txt, uint16_t *ops, int cap)
{
    int n = 0;
    while (*txt) {
        int len = 0; while (*txt >= '0' && *txt <= '9') len = len * 10 + (*txt++ - '0');
        const int op = *txt == 'M' ? 0 : *txt == 'I' ? 1 : *txt == 'D' ? 2 : -1;
        if (op < 0 || n >= cap) return -1;
        ops[n++] = (uint16_t)((len << 4) | op); ++txt;
    }
    return n;
}
static void to_row(const so_index_t *ora, const uint8_t *seq, int L, const so_result_t *q, salt_result_t *r)
{
    memset(r, 0, sizeof *r);
    r->pos = q->pos; r->strand = (uint8_t)q->strand; r->n_diff = q->n_diff; r->is_gap = q->is_gap; r->mapq = q->mapq;
    r->b0 = q->b0; r->b1 = q->b1; r->seq_start = (uint16_t)q->seq_start; r->seq_end = (uint16_t)q->seq_end;
    r->n_hits[0] = (uint8_t)q->n_hits[0]; r->n_hits[1] = (uint8_t)q->n_hits[1];
    if (q->pos != 0xFFFFFFFFu) { const int n = cigar_bin(q->cigar, r->cigar, SALT_MAX_CIGAR_OPS); r->n_cigar = (uint8_t)(n > 0 ? n : 0); }
    so_arrays_t a; so_index_arrays(ora, &a);
    int h = 0;
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < q->n_hits[s]; ++j, ++h) {
            r->hits[s][j].pos = q->hits[s][j].pos; r->hits[s][j].n_diff = q->hits[s][j].n_diff; r->hits[s][j].is_gap = q->hits[s][j].is_gap; r->hits[s][j].strand = (uint16_t)s;
            if (!q->hits[s][j].is_gap) continue;
            uint8_t rd[4096]; char txt[SO_CIGAR_MAX];
            for (int i = 0; i < L; ++i) { const uint8_t c = s ? seq[L - 1 - i] : seq[i]; rd[i] = (uint8_t)(s && c < 4 ? 3 - c : c); }
            if (so_ed_diff_cigar(a.ref, q->hits[s][j].pos, (uint32_t)L + 4, rd, (uint32_t)L, q->hits[s][j].n_diff, txt, sizeof txt) >= 0) {
                const int n = cigar_bin(txt, r->hit_cigar[h], SALT_MAX_CIGAR_OPS); r->hit_n_cigar[h] = (uint8_t)(n > 0 ? n : 0);
            }
        }
}
static void opt_from(const salt_gpu_ws_t *ws, const salt_aln_opt_t *o, so_opt_t *so)
{
    so_opt_default(ws->ix->ora, so);
    so->l_overlap = o->l_overlap; so->max_seed = o->max_seed; so->max_locate = o->max_locate; so->seed_only_ref = o->seed_only_ref;
}
/* the host pipeline's per-batch calls (gzip, pipes, multi-line records, and what the text path hands over) */
int salt_gpu_align_se(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, uint32_t n, const uint8_t *s, const uint32_t *f, salt_result_t *r)
{
    if (getenv("SALT_STUB_NO_HOST")) return fail("salt_gpu_align_se is not part of the stub");
    so_opt_t so; opt_from(ws, o, &so);
    for (uint32_t i = 0; i < n; ++i) {
        const int L = (int)(f[i + 1] - f[i]);
        if (L <= 0 || L >= 4096) return fail("read length outside the stub's range");
        so_result_t q; so_align_se1(ws->ix->ora, &so, s + f[i], L, &q, NULL);
        to_row(ws->ix->ora, s + f[i], L, &q, r + i);
        int n_amb = 0; for (int b = 0; b < L; ++b) n_amb += s[f[i] + b] > 3;
        r[i].skipped = n_amb > 200;                          /* alnse.c:1328: the record is never built */
    }
    return SALT_OK;
}
int salt_gpu_align_pe(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_pe_opt_t *pe, uint32_t n, const uint8_t *s, const uint32_t *f, salt_result_t *r)
{
    if (getenv("SALT_STUB_NO_PE") || getenv("SALT_STUB_NO_HOST")) return fail("salt_gpu_align_pe is not part of the stub");
    so_opt_t so; opt_from(ws, o, &so);
    for (uint32_t p = 0; p < n; ++p) {
        const int l0 = (int)(f[2 * p + 1] - f[2 * p]), l1 = (int)(f[2 * p + 2] - f[2 * p + 1]);
        if (l0 <= 0 || l1 <= 0 || l0 >= 4096 || l1 >= 4096) return fail("read length outside the stub's range");
        so_result_t q[2]; so_align_pe1(ws->ix->ora, &so, pe->min_tlen, pe->max_tlen, s + f[2 * p], l0, s + f[2 * p + 1], l1, q);
        to_row(ws->ix->ora, s + f[2 * p], l0, &q[0], r + 2 * p); to_row(ws->ix->ora, s + f[2 * p + 1], l1, &q[1], r + 2 * p + 1);
    }
    return SALT_OK;
}

static uint8_t nt4(int c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }

int salt_gpu_align_se_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_text_opt_t *to, const char *fq, uint64_t n_bytes,
                           const char **sam, uint64_t *sam_bytes, uint32_t *n_reads)
{
    *sam = NULL; *sam_bytes = 0; *n_reads = 0;
    if (n_bytes == 0) return SALT_OK;
    if (fq[n_bytes - 1] != '\n') return fail("block must end with a newline");
    so_opt_t so; so_opt_default(ws->ix->ora, &so);
    so.l_overlap = o->l_overlap; so.max_seed = o->max_seed; so.max_locate = o->max_locate; so.seed_only_ref = o->seed_only_ref;
    so.print_xa_cigar = to->print_xa_cigar; so.print_nm_md = to->print_nm_md; so.rg_id = to->rg_id;
    size_t used = 0; uint32_t n = 0;
    uint64_t p = 0;
    while (p < n_bytes) {
        const char *l[5]; l[0] = fq + p;
        for (int k = 1; k <= 4; ++k) { const char *nl = memchr(l[k - 1], '\n', (size_t)(fq + n_bytes - l[k - 1])); if (!nl) return fail("block does not hold whole 4-line records"); l[k] = nl + 1; }
        if (l[0][0] != '@' || l[2][0] != '+') return fail("not 4-line FASTQ");
        char name[512]; size_t nl_ = 0;
        for (const char *c = l[0] + 1; c < l[1] - 1 && *c != ' ' && !(*c >= 9 && *c <= 13) && nl_ < sizeof name - 1; ++c) name[nl_++] = *c;
        if (nl_ > 2 && name[nl_ - 2] == '/' && name[nl_ - 1] >= '0' && name[nl_ - 1] <= '9') nl_ -= 2;
        name[nl_] = 0;
        size_t L = (size_t)(l[2] - 1 - l[1]); while (L && l[1][L - 1] == '\r') --L;
        uint8_t seq[4096]; char qual[4097];
        if (L == 0 || L >= sizeof seq) return fail("read length outside the stub's range");
        for (size_t i = 0; i < L; ++i) { seq[i] = nt4((unsigned char)l[1][i]); qual[i] = l[3][i]; }
        qual[L] = 0;
        so_result_t res;
        so_align_se1(ws->ix->ora, &so, seq, (int)L, &res, NULL);
        if (used + 8 * L + 8192 > ws->cap) { ws->cap = (used + 8 * L + 8192) * 2; ws->sam = realloc(ws->sam, ws->cap); }
        int w = so_sam_se(ws->ix->ora, &so, name, seq, (int)L, qual, &res, ws->sam + used, ws->cap - used);
        if (w < 0) return fail("SAM record too long");
        used += (size_t)w; ws->sam[used++] = '\n';
        ++n; p = (uint64_t)(l[4] - fq);
    }
    const char *log = getenv("SALT_STUB_LOG");
    if (log) { pthread_mutex_lock(&g_mu); FILE *f = fopen(log, "a"); if (f) { fprintf(f, "%d %u\n", ws->ix->device, n); fclose(f); } pthread_mutex_unlock(&g_mu); }
    *sam = ws->sam; *sam_bytes = used; *n_reads = n;
    return SALT_OK;
}

/* one 4-line record at *p of fq[0..n): name (trimmed like query.c:139-143), codes, qualities; advances *p; 0 = ok */
static int stub_record(const char *fq, uint64_t n, uint64_t *p, char *name, size_t name_cap, uint8_t *seq, char *qual, size_t cap, int *L_out)
{
    const char *l[5]; l[0] = fq + *p;
    for (int k = 1; k <= 4; ++k) { const char *nl = memchr(l[k - 1], '\n', (size_t)(fq + n - l[k - 1])); if (!nl) return -1; l[k] = nl + 1; }
    if (l[0][0] != '@' || l[2][0] != '+') return -2;
    size_t nl_ = 0;
    for (const char *c = l[0] + 1; c < l[1] - 1 && *c != ' ' && !(*c >= 9 && *c <= 13) && nl_ < name_cap - 1; ++c) name[nl_++] = *c;
    if (nl_ > 2 && name[nl_ - 2] == '/' && name[nl_ - 1] >= '0' && name[nl_ - 1] <= '9') nl_ -= 2;
    name[nl_] = 0;
    size_t L = (size_t)(l[2] - 1 - l[1]); while (L && l[1][L - 1] == '\r') --L;
    if (L == 0 || L >= cap) return -3;
    for (size_t i = 0; i < L; ++i) { seq[i] = nt4((unsigned char)l[1][i]); qual[i] = l[3][i]; }
    qual[L] = 0;
    *L_out = (int)L; *p = (uint64_t)(l[4] - fq);
    return 0;
}

int salt_gpu_align_pe_text(salt_gpu_ws_t *ws, const salt_aln_opt_t *o, const salt_pe_opt_t *pe, const salt_text_opt_t *to, const char *f1, uint64_t n1,
                           const char *f2, uint64_t n2, const char **sam, uint64_t *sam_bytes, uint32_t *n_pairs)
{
    *sam = NULL; *sam_bytes = 0; *n_pairs = 0;
    if (getenv("SALT_STUB_NO_PE")) return fail("salt_gpu_align_pe_text is not part of the stub");
    if (n1 == 0 && n2 == 0) return SALT_OK;
    if (n1 == 0 || n2 == 0 || f1[n1 - 1] != '\n' || f2[n2 - 1] != '\n') return fail("blocks must hold the same number of whole records");
    so_opt_t so; so_opt_default(ws->ix->ora, &so);
    so.l_overlap = o->l_overlap; so.max_seed = o->max_seed; so.max_locate = o->max_locate; so.seed_only_ref = o->seed_only_ref;
    so.print_xa_cigar = to->print_xa_cigar; so.print_nm_md = to->print_nm_md; so.rg_id = to->rg_id;
    size_t used = 0; uint32_t n = 0;
    uint64_t p1 = 0, p2 = 0;
    while (p1 < n1 && p2 < n2) {
        char nm0[512], nm1[512], q0[4097], q1[4097]; uint8_t s0[4096], s1[4096]; int l0 = 0, l1 = 0;
        if (stub_record(f1, n1, &p1, nm0, sizeof nm0, s0, q0, sizeof s0, &l0) || stub_record(f2, n2, &p2, nm1, sizeof nm1, s1, q1, sizeof s1, &l1)) return fail("not whole 4-line FASTQ records");
        so_result_t res[2];
        so_align_pe1(ws->ix->ora, &so, pe->min_tlen, pe->max_tlen, s0, l0, s1, l1, res);
        const size_t need = 16 * (size_t)(l0 + l1) + 16384;
        if (used + need > ws->cap) { ws->cap = (used + need) * 2; ws->sam = realloc(ws->sam, ws->cap); }
        const char *nm[2] = { nm0, nm1 }, *ql[2] = { q0, q1 }; const uint8_t *sq[2] = { s0, s1 }; const int ls[2] = { l0, l1 };
        int w = so_sam_pe(ws->ix->ora, &so, pe->min_tlen, pe->max_tlen, nm, sq, ls, ql, res, ws->sam + used, ws->cap - used);
        if (w < 0) return fail("SAM record too long");
        used += (size_t)w;
        ++n;
    }
    if (p1 < n1 || p2 < n2) return fail("the two FASTQ blocks hold different numbers of reads");
    const char *log = getenv("SALT_STUB_LOG");
    if (log) { pthread_mutex_lock(&g_mu); FILE *f = fopen(log, "a"); if (f) { fprintf(f, "%d %u\n", ws->ix->device, n); fclose(f); } pthread_mutex_unlock(&g_mu); }
    *sam = ws->sam; *sam_bytes = used; *n_pairs = n;
    return SALT_OK;
}
