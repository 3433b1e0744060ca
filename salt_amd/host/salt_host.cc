// salt_amd/host/salt_host.cc -- host side of the drop-in: index files in, SAM text out.
// See include/salt_host.h for the reference interfaces each entry point mirrors.
#include "../../include/salt_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

struct Ann { int64_t offset; int32_t len; int32_t n_ambs; std::string name; };
struct Amb { int64_t offset; int32_t len; char amb; };

bool read_file(const std::string &fn, std::vector<uint8_t> &out)
{
    std::ifstream f(fn, std::ios::binary | std::ios::ate);
    if (!f) { g_err = "cannot open " + fn; return false; }
    std::streamsize n = f.tellg();
    f.seekg(0);
    out.resize((size_t)n);
    if (n && !f.read(reinterpret_cast<char *>(out.data()), n)) { g_err = "short read on " + fn; return false; }
    return true;
}

inline uint32_t u32_at(const std::vector<uint8_t> &b, size_t word) { uint32_t v; memcpy(&v, b.data() + 4 * word, 4); return v; }

} // namespace

struct salt_index {
    std::vector<uint32_t> c_bwt, c_sa, lkt, r_bwt, r_occ, r_major, r_sa, ref;
    std::vector<uint8_t> pac;
    std::vector<Ann> anns;
    std::vector<Amb> ambs;
    int64_t l_pac = 0;
    int32_t seed_len = 0;
    salt_host_index_t view;
};

extern "C" const char *salt_host_last_error(void) { return g_err.c_str(); }

extern "C" void salt_lkt_build(const uint8_t *pac, uint32_t l_ref, int len, uint32_t *item)
{
    // counts of every 12-mer start + the `len` A-padded tail suffixes, then prefix sums
    // (Index_src/LookUpTable.c:70-150; the padded tail is why a bucket may hold a too-short suffix)
    const uint32_t n_item = (1u << (2 * len)) + 1, mask = n_item - 2;
    memset(item, 0, (size_t)n_item * 4);
    uint32_t x = 0;
    for (uint32_t i = 0; i < l_ref; ++i) {
        x = ((x << 2) & mask) | ((pac[i >> 2] >> ((~i & 3u) << 1)) & 3u);
        if (i + 1 >= (uint32_t)len) ++item[x + 1];
    }
    for (int i = 0; i < len; ++i) { x = (x << 2) & mask; ++item[x + 1]; }
    for (uint32_t i = 1; i < n_item; ++i) item[i] += item[i - 1];
}

extern "C" salt_index_t *salt_index_load(const char *prefix_c, int rebuild_lkt)
{
    const std::string p(prefix_c);
    std::vector<uint8_t> b;
    salt_index *ix = new salt_index();
    salt_host_index_t &v = ix->view;
    memset(&v, 0, sizeof v);
    auto bail = [&](const std::string &m) -> salt_index_t * { if (!m.empty()) g_err = m; delete ix; return nullptr; };

    if (!read_file(p + ".R.seedLen", b) || b.size() < 4) return bail(b.size() < 4 ? "seed length can't be parsed: " + p + ".R.seedLen" : "");
    memcpy(&ix->seed_len, b.data(), 4);

    if (!read_file(p + ".C.bwt", b)) return bail("");
    if (b.size() < 20 || b.size() % 4) return bail(p + ".C.bwt: malformed");
    v.c_primary = u32_at(b, 0);
    for (int i = 0; i < 4; ++i) v.c_L2[i + 1] = u32_at(b, 1 + i);
    v.c_seq_len = v.c_L2[4];
    ix->c_bwt.assign(reinterpret_cast<const uint32_t *>(b.data()) + 5, reinterpret_cast<const uint32_t *>(b.data() + b.size()));
    v.c_bwt_size = (uint32_t)ix->c_bwt.size();

    if (!read_file(p + ".C.sa", b)) return bail("");
    if (b.size() < 28) return bail(p + ".C.sa: malformed");
    if (u32_at(b, 0) != v.c_primary || u32_at(b, 6) != v.c_seq_len) return bail("SA-BWT inconsistency: " + p + ".C.sa");
    v.c_sa_intv = u32_at(b, 5);
    if (v.c_sa_intv == 0) return bail(p + ".C.sa: zero sampling interval");
    v.c_n_sa = (v.c_seq_len + v.c_sa_intv) / v.c_sa_intv;
    if (b.size() < 28 + 4 * (size_t)(v.c_n_sa - 1)) return bail(p + ".C.sa: truncated");
    ix->c_sa.resize(v.c_n_sa);
    ix->c_sa[0] = 0xFFFFFFFFu;
    memcpy(ix->c_sa.data() + 1, b.data() + 28, 4 * (size_t)(v.c_n_sa - 1));

    if (!read_file(p + ".R.backward.bwt", b)) return bail("");
    if (b.size() < 32) return bail(p + ".R.backward.bwt: malformed");
    v.r_text_len = u32_at(b, 0); v.r_inv_sa0 = u32_at(b, 1);
    for (int i = 0; i < 5; ++i) v.r_cum[i + 1] = u32_at(b, 2 + i);
    v.r_bwt_words = u32_at(b, 7);
    if (b.size() < 32 + 4 * (size_t)v.r_bwt_words) return bail(p + ".R.backward.bwt: truncated");
    ix->r_bwt.assign((size_t)v.r_bwt_words + 64, 0);
    memcpy(ix->r_bwt.data(), b.data() + 32, 4 * (size_t)v.r_bwt_words);

    if (!read_file(p + ".R.backward.occ", b)) return bail("");
    {
        if (b.size() < 8) return bail(p + ".R.backward.occ: malformed");
        uint32_t n = u32_at(b, 0);
        if (b.size() < 8 + 4 * (size_t)n) return bail(p + ".R.backward.occ: truncated");
        ix->r_occ.assign(reinterpret_cast<const uint32_t *>(b.data()) + 1, reinterpret_cast<const uint32_t *>(b.data()) + 1 + n);
        uint32_t m = u32_at(b, 1 + n);
        if (b.size() < 8 + 4 * ((size_t)n + m)) return bail(p + ".R.backward.occ: truncated");
        ix->r_major.assign(reinterpret_cast<const uint32_t *>(b.data()) + 2 + n, reinterpret_cast<const uint32_t *>(b.data()) + 2 + n + m);
        v.r_occ_words = n; v.r_major_words = m;
    }
    if (!read_file(p + ".R.backward.sa", b)) return bail("");
    {
        if (b.size() < 4) return bail(p + ".R.backward.sa: malformed");
        uint32_t n = u32_at(b, 0);
        if (b.size() < 4 + 4 * (size_t)n) return bail(p + ".R.backward.sa: truncated");
        ix->r_sa.assign(reinterpret_cast<const uint32_t *>(b.data()) + 1, reinterpret_cast<const uint32_t *>(b.data()) + 1 + n);
        v.r_n_sa = n;
    }
    if (!read_file(p + ".ref", b)) return bail("");
    {
        if (b.size() < 4) return bail(p + ".ref: malformed");
        v.ref_len = u32_at(b, 0);
        size_t nw = ((size_t)v.ref_len + 7) / 8;
        if (b.size() < 4 + 4 * nw) return bail(p + ".ref: truncated");
        ix->ref.assign(nw + 4, 0);
        memcpy(ix->ref.data(), b.data() + 4, 4 * nw);
    }
    {   // .C.ann / .C.amb (text, BWA 0.5/0.6 layout)
        std::ifstream f(p + ".C.ann");
        if (!f) return bail("cannot open " + p + ".C.ann");
        long long lp; int n_seqs; unsigned seed;
        if (!(f >> lp >> n_seqs >> seed)) return bail(p + ".C.ann: malformed");
        ix->l_pac = lp;
        for (int i = 0; i < n_seqs; ++i) {
            unsigned gi; Ann a; std::string rest; long long off;
            if (!(f >> gi >> a.name)) return bail(p + ".C.ann: malformed");
            std::getline(f, rest);
            if (!(f >> off >> a.len >> a.n_ambs)) return bail(p + ".C.ann: malformed");
            a.offset = off;
            ix->anns.push_back(a);
        }
        std::ifstream g(p + ".C.amb");
        if (!g) return bail("cannot open " + p + ".C.amb");
        long long lp2; int ns2, n_holes;
        if (!(g >> lp2 >> ns2 >> n_holes)) return bail(p + ".C.amb: malformed");
        if (lp2 != lp || ns2 != n_seqs) return bail("inconsistent .ann and .amb files");
        for (int i = 0; i < n_holes; ++i) {
            long long off; Amb a; std::string s;
            if (!(g >> off >> a.len >> s)) return bail(p + ".C.amb: malformed");
            a.offset = off; a.amb = s.empty() ? 'N' : s[0];
            ix->ambs.push_back(a);
        }
    }
    if (!read_file(p + ".C.pac", ix->pac)) return bail("");
    if (ix->pac.size() < (size_t)ix->l_pac / 4 + 1) return bail(p + ".C.pac: truncated");
    ix->pac.resize((size_t)ix->l_pac / 4 + 8, 0);
    {
        std::ifstream f(p + ".C.lkt", std::ios::binary);
        if (f) {
            int32_t len = 0;
            f.read(reinterpret_cast<char *>(&len), 4);
            if (!f || len < 1 || len > 14) return bail(p + ".C.lkt: malformed");
            v.lkt_len = (uint32_t)len; v.lkt_n = (1u << (2 * len)) + 1;
            ix->lkt.resize(v.lkt_n);
            f.read(reinterpret_cast<char *>(ix->lkt.data()), 4 * (std::streamsize)v.lkt_n);
            if (!f) return bail(p + ".C.lkt: truncated");
        } else if (rebuild_lkt) {
            v.lkt_len = 12; v.lkt_n = (1u << 24) + 1;
            ix->lkt.resize(v.lkt_n);
            salt_lkt_build(ix->pac.data(), (uint32_t)ix->l_pac, 12, ix->lkt.data());
        } else return bail("cannot open " + p + ".C.lkt");
    }
    v.c_bwt = ix->c_bwt.data(); v.c_sa = ix->c_sa.data(); v.lkt = ix->lkt.data();
    v.r_bwt = ix->r_bwt.data(); v.r_occ = ix->r_occ.data(); v.r_major = ix->r_major.data(); v.r_sa = ix->r_sa.data();
    v.ref = ix->ref.data();
    v.l_seed = ix->seed_len;
    return ix;
}

extern "C" void salt_index_free(salt_index_t *ix) { delete ix; }
extern "C" const salt_host_index_t *salt_index_host_view(const salt_index_t *ix) { return &ix->view; }
extern "C" int32_t salt_index_seed_len(const salt_index_t *ix) { return ix->seed_len; }
extern "C" const uint8_t *salt_index_pac(const salt_index_t *ix, uint64_t *l_pac) { if (l_pac) *l_pac = (uint64_t)ix->l_pac; return ix->pac.data(); }
extern "C" int32_t salt_index_n_seqs(const salt_index_t *ix) { return (int32_t)ix->anns.size(); }
extern "C" int salt_index_seq(const salt_index_t *ix, int32_t i, int64_t *offset, int32_t *len, const char **name)
{
    if (!ix || i < 0 || (size_t)i >= ix->anns.size()) return -1;
    if (offset) *offset = ix->anns[(size_t)i].offset;
    if (len) *len = ix->anns[(size_t)i].len;
    if (name) *name = ix->anns[(size_t)i].name.c_str();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// SAM text
// ---------------------------------------------------------------------------------------------
namespace {

struct Out {
    char *s; size_t n, cap; bool ovf;
    void put(char c) { if (n + 1 < cap) s[n++] = c; else ovf = true; }
    void puts(const char *t) { while (*t) put(*t++); }
    void putu(uint64_t v) { char tmp[24]; int k = 0; do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k) put(tmp[--k]); }
    int done() { if (ovf || n >= cap) return -1; s[n] = 0; return (int)n; }
};

int seq_id(const salt_index *ix, int64_t coor)          // bns_coor_pac2real's sequence search (bntseq.c:269-289)
{
    int left = 0, mid = 0, right = (int)ix->anns.size();
    while (left < right) {
        mid = (left + right) >> 1;
        if (coor >= ix->anns[mid].offset) {
            if (mid == (int)ix->anns.size() - 1) break;
            if (coor < ix->anns[mid + 1].offset) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

inline uint32_t pac_at(const salt_index *ix, uint32_t l) { return (ix->pac[l >> 2] >> ((~l & 3u) << 1)) & 3u; }
inline uint32_t mask_at(const salt_index *ix, uint32_t l) { return (ix->ref[l >> 3] >> (4 * (l & 7u))) & 15u; }

void put_cigar(Out &o, const uint16_t *ops, int n)
{
    for (int i = 0; i < n; ++i) { o.putu(ops[i] >> 4); o.put("MID?"[ops[i] & 3]); }
}

void put_xa(Out &o, const salt_index *ix, const salt_sam_opt_t *opt, int L, const salt_result_t *q)
{
    // XA (sam.c:186-240)
    bool first = true; int h = 0;
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < q->n_hits[s]; ++i, ++h) {
            const salt_hit_t &hit = q->hits[s][i];
            if (hit.pos == q->pos) continue;
            if (first) { o.puts("\tXA:Z:"); first = false; }
            int r2 = seq_id(ix, hit.pos);
            o.puts(ix->anns[r2].name.c_str()); o.put(','); o.put("+-"[s]);
            o.putu((uint64_t)((int64_t)hit.pos - ix->anns[r2].offset + 1)); o.put(',');
            if (opt->print_xa_cigar) {
                if (hit.is_gap) put_cigar(o, q->hit_cigar[h], q->hit_n_cigar[h]);
                else { o.putu((uint64_t)L); o.put('M'); }
                o.put(',');
            } else o.puts("*,");
            o.putu(hit.n_diff); o.put(';');
        }
}

void put_md_nm(Out &o, const salt_index *ix, const uint8_t *sq, const salt_result_t *q)
{
    static const char NT[] = "ACGTN";
    // MD / NM / XV against the 2-bit pac (sam.c:246-328)
    {
        int nm = 0, n_match = 0, n_rs = 0, rsv[64];
        uint32_t rp = q->pos; int si = q->seq_start;
        o.puts("\tMD:Z:");
        for (int c = 0; c < q->n_cigar; ++c) {
            int n = q->cigar[c] >> 4, op = q->cigar[c] & 15;
            if (op == 0) {
                for (int i = 0; i < n; ++i, ++rp, ++si) {
                    uint32_t bt = pac_at(ix, rp);
                    if (bt == sq[si]) { ++n_match; continue; }
                    if (sq[si] < 5 && (mask_at(ix, rp) & (1u << sq[si])) != 0 && n_rs < 64) rsv[n_rs++] = si - q->seq_start;
                    ++nm;
                    if (n_match) o.putu((uint64_t)n_match);
                    n_match = 0;
                    o.put(NT[bt]);
                }
            } else if (op == 1) { nm += n; si += n; }
            else if (op == 2) {
                if (n_match) o.putu((uint64_t)n_match);
                n_match = 0; nm += n; o.put('^');
                for (int i = 0; i < n; ++i, ++rp) o.put(NT[pac_at(ix, rp)]);
            }
        }
        if (n_match) o.putu((uint64_t)n_match);
        o.puts("\tNM:i:"); o.putu((uint64_t)nm);
        if (n_rs > 0) { o.puts("\tXV:i:"); for (int i = 0; i < n_rs; ++i) { if (i) o.put(','); o.putu((uint64_t)rsv[i]); } }
    }
}

} // namespace

extern "C" int salt_isize_infer(const salt_index_t *ix, uint32_t n_pairs, const uint32_t *offs, const salt_result_t *res,
                                uint32_t *min_tlen, uint32_t *max_tlen, uint32_t *n_used)
{
    std::vector<uint32_t> t;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        const salt_result_t &a = res[2 * (size_t)i], &b = res[2 * (size_t)i + 1];
        if (a.skipped || b.skipped || a.pos == 0xFFFFFFFFu || b.pos == 0xFFFFFFFFu || a.is_gap || b.is_gap) continue;
        if (a.n_hits[0] || a.n_hits[1] || b.n_hits[0] || b.n_hits[1]) continue;
        if (a.strand > 1 || b.strand > 1 || a.strand == b.strand) continue;
        const bool a_fwd = a.strand == 0;
        const salt_result_t &f = a_fwd ? a : b, &r = a_fwd ? b : a;
        const uint32_t len_r = a_fwd ? offs[2 * (size_t)i + 2] - offs[2 * (size_t)i + 1] : offs[2 * (size_t)i + 1] - offs[2 * (size_t)i];
        if (r.pos < f.pos || seq_id(ix, f.pos) != seq_id(ix, r.pos)) continue;
        const uint64_t tl = (uint64_t)r.pos + len_r - f.pos;
        if (tl <= 100000) t.push_back((uint32_t)tl);
    }
    if (n_used) *n_used = (uint32_t)t.size();
    const size_t n = t.size();
    if (n < 25) return -1;
    std::sort(t.begin(), t.end());
    const uint64_t q1 = t[n / 4], q3 = t[3 * n / 4], iqr = q3 - q1;
    const uint64_t lo = q1 > 2 * iqr ? q1 - 2 * iqr : 0, hi = q3 + 2 * iqr;
    uint64_t m = 0, sum = 0;
    for (uint32_t v : t) if (v >= lo && v <= hi) { ++m; sum += v; }
    const uint64_t mean = (sum + m / 2) / m;
    uint64_t var = 0;
    for (uint32_t v : t) if (v >= lo && v <= hi) { const uint64_t d = v > mean ? v - mean : mean - v; var += d * d; }
    var /= m;
    uint64_t sd = 0;
    while ((sd + 1) * (sd + 1) <= var) ++sd;                  // floor(sqrt(var)) ...
    if (sd * sd < var) ++sd;                                  // ... rounded up
    uint64_t wa = mean > 4 * sd ? mean - 4 * sd : 1, wb = mean + 4 * sd;
    wa = std::min<uint64_t>(wa, q1 > 3 * iqr ? q1 - 3 * iqr : 1);
    wb = std::max<uint64_t>(wb, q3 + 3 * iqr);
    *min_tlen = (uint32_t)std::max<uint64_t>(wa, 1); *max_tlen = (uint32_t)wb;
    return 0;
}

extern "C" int salt_cigar_text(const uint16_t *ops, int n_ops, char *buf, size_t cap)
{
    Out o{ buf, 0, cap, false };
    put_cigar(o, ops, n_ops);
    return o.done();
}

extern "C" int salt_sam_header(const salt_index_t *ix, const salt_sam_opt_t *opt, char *buf, size_t cap)
{
    Out o{ buf, 0, cap, false };
    o.puts("@HD\tVN:ec1fec2\tSO:unsorted\n");
    for (const Ann &a : ix->anns) { o.puts("@SQ\tSN:"); o.puts(a.name.c_str()); o.puts("\tLN:"); o.putu((uint64_t)a.len); o.put('\n'); }
    o.puts("@RG\tID:"); o.puts(opt && opt->rg_id ? opt->rg_id : "(null)"); o.put('\n');     // printf("%s", NULL) in sam.c:69
    return o.done();
}

extern "C" int salt_sam_se(const salt_index_t *ix, const salt_sam_opt_t *opt, const char *name, const uint8_t *seq,
                           int32_t L, const char *qual, const salt_result_t *q, char *buf, size_t cap)
{
    Out o{ buf, 0, cap, false };
    if (cap) buf[0] = 0;
    if (q->skipped) return 0;
    static const char NT[] = "ACGTN";
    if (q->pos == 0xFFFFFFFFu) {                                 // sam.c:105-125
        o.puts(name); o.puts("\t4\t*\t0\t0\t*\t*\t0\t0\t");
        for (int i = 0; i < L; ++i) o.put(NT[seq[i] > 4 ? 4 : seq[i]]);
        o.put('\t'); o.puts(qual ? qual : "*");
        return o.done();
    }
    uint8_t rs_stack[1024]; std::vector<uint8_t> rs_heap;       // the reverse complement: no allocation per read for ordinary lengths
    uint8_t *rs = rs_stack;
    if (L > (int)sizeof rs_stack) { rs_heap.resize((size_t)L); rs = rs_heap.data(); }
    if (q->strand) for (int i = 0; i < L; ++i) { uint8_t c = seq[L - 1 - i]; rs[i] = c < 4 ? (uint8_t)(3 - c) : c; }
    const uint8_t *sq = q->strand ? rs : seq;                    // bases as aligned
    int rid = seq_id(ix, q->pos);
    o.puts(name); o.put('\t'); o.putu(q->strand ? 16 : 0); o.put('\t'); o.puts(ix->anns[rid].name.c_str()); o.put('\t');
    o.putu((uint64_t)((int64_t)q->pos - ix->anns[rid].offset + 1)); o.put('\t'); o.putu(q->mapq); o.put('\t');
    put_cigar(o, q->cigar, q->n_cigar);
    o.puts("\t*\t0\t0\t");
    for (int i = 0; i < L; ++i) o.put(NT[sq[i] > 4 ? 4 : sq[i]]);
    o.put('\t');
    if (q->strand) { if (qual) for (int i = L - 1; i >= 0; --i) o.put(qual[i]); else o.put('*'); }
    else o.puts(qual && qual[0] ? qual : "*");
    put_xa(o, ix, opt, L, q);
    if (opt->print_nm_md) put_md_nm(o, ix, sq, q);
    if (opt->rg_id) { o.puts("\tRG:Z:"); o.puts(opt->rg_id); }
    return o.done();
}

// alnpe_sam (sam.c:331-457): both records of pair; each is followed by "\n\n" exactly as the reference's driver
// prints them (the record's own newline plus the printf("%s\n") around it, alnpe.c:640-648)
extern "C" int salt_sam_pe(const salt_index_t *ix, const salt_sam_opt_t *opt, const salt_pe_opt_t *pe, const char *const name[2],
                           const uint8_t *const seq[2], const int32_t l_seq[2], const char *const qual[2],
                           const salt_result_t *q, char *buf, size_t cap)
{
    Out o{ buf, 0, cap, false };
    static const char NT[] = "ACGTN";
    int rid[2] = { -1, -1 }, tlen = 0;
    bool is_map[2] = { false, false };
    uint32_t pos[2] = { 0, 0 };
    for (int i = 0; i < 2; ++i)
        if (q[i].pos != 0xFFFFFFFFu) { is_map[i] = true; rid[i] = seq_id(ix, q[i].pos); pos[i] = q[i].pos - (uint32_t)ix->anns[rid[i]].offset + 1; }
    if (is_map[0] && is_map[1]) {
        if (rid[0] != rid[1]) tlen = 0;
        else if (pos[0] < pos[1]) tlen = (int)(pos[1] + q[1].seq_end - q[1].seq_start + 1 - pos[0]);
        else tlen = (int)(pos[0] + q[0].seq_end - q[1].seq_start + 1 - pos[1]);              // sam.c:355-356: q[1].seq_start in both arms
        if ((uint32_t)tlen > pe->max_tlen || (uint32_t)tlen < pe->min_tlen) tlen = 0;
    }
    for (int i = 0; i < 2; ++i) {
        const int L = l_seq[i];
        uint8_t rs_stack[1024]; std::vector<uint8_t> rs_heap;
        uint8_t *rs = rs_stack;
        if (L > (int)sizeof rs_stack) { rs_heap.resize((size_t)L); rs = rs_heap.data(); }
        for (int k = 0; k < L; ++k) { uint8_t c = seq[i][L - 1 - k]; rs[k] = c < 4 ? (uint8_t)(3 - c) : c; }
        unsigned flag = 0x1;
        if (!is_map[i]) flag |= 0x4;
        if (!is_map[1 - i]) flag |= 0x8;
        if (q[i].strand == 1) flag |= 0x10;
        if (q[1 - i].strand == 1) flag |= 0x20;
        if (tlen != 0) flag |= 0x2;
        flag |= i == 0 ? 0x40 : 0x80;
        o.puts(name[i]); o.put('\t'); o.putu(flag); o.put('\t');
        if (is_map[i]) {
            o.puts(ix->anns[rid[i]].name.c_str()); o.put('\t'); o.putu(pos[i]); o.put('\t'); o.putu(q[i].mapq); o.put('\t');
            if (q[i].seq_start != 0) { o.putu(q[i].seq_start); o.put('S'); }
            put_cigar(o, q[i].cigar, q[i].n_cigar);
            if (q[i].seq_end != (uint32_t)L - 1) { o.putu((uint64_t)(L - (int)q[i].seq_end - 1)); o.put('S'); }
            o.put('\t');
        } else if (is_map[1 - i]) { o.puts(ix->anns[rid[1 - i]].name.c_str()); o.put('\t'); o.putu(pos[1 - i]); o.puts("\t255\t*\t"); }
        else o.puts("*\t0\t255\t*\t");
        if (is_map[1 - i]) {
            if (rid[i] == rid[1 - i] || !is_map[i]) o.puts("=\t"); else { o.puts(ix->anns[rid[1 - i]].name.c_str()); o.put('\t'); }
            o.putu(pos[1 - i]); o.put('\t');
        } else o.puts("*\t0\t");
        if (tlen != 0) { if (q[i].pos >= q[1 - i].pos) o.put('-'); o.putu((uint64_t)tlen); o.put('\t'); }
        else o.puts("0\t");
        const uint8_t *sq = q[i].strand == 1 ? rs : seq[i];
        for (int k = 0; k < L; ++k) o.put(NT[sq[k] > 4 ? 4 : sq[k]]);
        o.put('\t');
        const bool has_q = qual[i] && qual[i][0];
        if (!has_q) o.put('*');
        else if (q[i].strand == 1) for (int k = L - 1; k >= 0; --k) o.put(qual[i][k]);
        else o.puts(qual[i]);
        put_xa(o, ix, opt, L, &q[i]);
        if (opt->print_nm_md && is_map[i]) put_md_nm(o, ix, q[i].strand == 0 ? seq[i] : rs, &q[i]);
        if (opt->rg_id) { o.puts("\tRG:Z:"); o.puts(opt->rg_id); }
        o.puts("\n\n");
    }
    return o.done();
}
