// salt_amd/host/salt_idx.cc -- index builder: writes the files `salt` loads, in salt-idx's on-disk
// formats.  Row N1 of the scope table: needed so that the synthetic
// chr21-/GRCh38-scale genomes of the benchmark can be indexed on the GPU box, where the reference
// binary does not exist.
//
// What it mirrors (paths under Index_src/):
//   fa -> .C.pac/.C.ann/.C.amb         bns_fasta2bntseq                   bntseq.c:164-253
//   .C.lkt                             LKT_build_lookuptable              LookUpTable.c:70-150
//   .C.bwt (+ interleaved Occ), .C.sa  bwt_bwtgen + bwt_bwtupdate_core    bwtmisc.c:121-143
//                                      + bwt_cal_sa(8)                    bwt.c:48-68, index1.c:44
//   .lp (local patterns)               ss_core_alt                        localPattern.c:171-324
//   .R.backward.bwt/.occ               Rbwt_bwt_bwtgen                    4bit_bwt_gen.c:1044-1130,1409-1459
//   .R.backward.sa                     Rbwt_gen_sa(direction=-1)          rbwt.c:424-475
//   .ref                               build_mixRef                       mixRef.c:96-190
//   .R.seedLen                         index1.c:138-141
// The reference builds its BWTs incrementally (BWT-SW); a BWT is canonical, so here both are
// derived from suffix arrays built with SA-IS (written from the published algorithm).
// Not written: .R.forward.*, .R.pac/.rpac/.ann/.amb -- `salt` never reads them (rbwt.c:495-498).
#include "../../include/salt_host.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <algorithm>

namespace {

// ---------------------------------------------------------------------------------------------
// SA-IS (Nong, Zhang, Chan 2009), int32 suffix array, generic symbol type.
// s[n-1] must be the unique smallest symbol (0).
// ---------------------------------------------------------------------------------------------
template <class T>
struct Sais {
    static inline bool tget(const std::vector<uint8_t> &t, int32_t i) { return (t[(size_t)i >> 3] >> (i & 7)) & 1; }
    static inline void tset(std::vector<uint8_t> &t, int32_t i, bool b)
    {
        if (b) t[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7)); else t[(size_t)i >> 3] &= (uint8_t)~(1u << (i & 7));
    }
    static void buckets(const T *s, std::vector<int32_t> &bkt, int32_t n, int32_t K, bool end)
    {
        std::fill(bkt.begin(), bkt.end(), 0);
        for (int32_t i = 0; i < n; ++i) ++bkt[(size_t)s[i]];
        int32_t sum = 0;
        for (int32_t i = 0; i < K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
    }
    static void induce_l(const std::vector<uint8_t> &t, int32_t *SA, const T *s, std::vector<int32_t> &bkt, int32_t n, int32_t K)
    {
        buckets(s, bkt, n, K, false);
        for (int32_t i = 0; i < n; ++i) {
            int32_t j = SA[i] - 1;
            if (j >= 0 && !tget(t, j)) SA[bkt[(size_t)s[j]]++] = j;
        }
    }
    static void induce_s(const std::vector<uint8_t> &t, int32_t *SA, const T *s, std::vector<int32_t> &bkt, int32_t n, int32_t K)
    {
        buckets(s, bkt, n, K, true);
        for (int32_t i = n - 1; i >= 0; --i) {
            int32_t j = SA[i] - 1;
            if (j >= 0 && tget(t, j)) SA[--bkt[(size_t)s[j]]] = j;
        }
    }
    static void run(const T *s, int32_t *SA, int32_t n, int32_t K)
    {
        if (n == 1) { SA[0] = 0; return; }
        std::vector<uint8_t> t((size_t)n / 8 + 1, 0);
        tset(t, n - 2, false); tset(t, n - 1, true);
        for (int32_t i = n - 3; i >= 0; --i) tset(t, i, s[i] < s[i + 1] || (s[i] == s[i + 1] && tget(t, i + 1)));
        auto is_lms = [&](int32_t i) { return i > 0 && tget(t, i) && !tget(t, i - 1); };
        std::vector<int32_t> bkt((size_t)K);
        // stage 1: sort the LMS substrings
        buckets(s, bkt, n, K, true);
        for (int32_t i = 0; i < n; ++i) SA[i] = -1;
        for (int32_t i = 1; i < n; ++i) if (is_lms(i)) SA[--bkt[(size_t)s[i]]] = i;
        induce_l(t, SA, s, bkt, n, K);
        induce_s(t, SA, s, bkt, n, K);
        int32_t n1 = 0;
        for (int32_t i = 0; i < n; ++i) if (is_lms(SA[i])) SA[n1++] = SA[i];
        for (int32_t i = n1; i < n; ++i) SA[i] = -1;
        int32_t name = 0, prev = -1;
        for (int32_t i = 0; i < n1; ++i) {
            int32_t pos = SA[i];
            bool diff = false;
            for (int32_t d = 0; d < n; ++d) {
                if (prev == -1 || s[pos + d] != s[prev + d] || tget(t, pos + d) != tget(t, prev + d)) { diff = true; break; }
                if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
            }
            if (diff) { ++name; prev = pos; }
            SA[n1 + (pos >> 1)] = name - 1;
        }
        for (int32_t i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];
        // stage 2: solve the reduced problem
        int32_t *SA1 = SA, *s1 = SA + n - n1;
        if (name < n1) Sais<int32_t>::run(s1, SA1, n1, name);
        else for (int32_t i = 0; i < n1; ++i) SA1[s1[i]] = i;
        // stage 3: induce the result
        buckets(s, bkt, n, K, true);
        for (int32_t i = 1, j = 0; i < n; ++i) if (is_lms(i)) s1[j++] = i;
        for (int32_t i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
        for (int32_t i = n1; i < n; ++i) SA[i] = -1;
        for (int32_t i = n1 - 1; i >= 0; --i) { int32_t j = SA[i]; SA[i] = -1; SA[--bkt[(size_t)s[j]]] = j; }
        induce_l(t, SA, s, bkt, n, K);
        induce_s(t, SA, s, bkt, n, K);
    }
};

thread_local std::string g_ierr;

bool write_file(const std::string &fn, const void *p, size_t n, const char *mode = "wb")
{
    FILE *f = fopen(fn.c_str(), mode);
    if (!f) { g_ierr = "cannot write " + fn; return false; }
    bool ok = n == 0 || fwrite(p, 1, n, f) == n;
    fclose(f);
    if (!ok) g_ierr = "short write on " + fn;
    return ok;
}

struct Contig { std::string name, comment, seq; };

bool read_fasta(const char *fn, std::vector<Contig> &out)
{
    gzFile fp = gzopen(fn, "r");
    if (!fp) { g_ierr = std::string("cannot open ") + fn; return false; }
    static char line[1 << 16];
    Contig *cur = nullptr;
    while (gzgets(fp, line, sizeof line)) {
        size_t n = strlen(line);
        while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        if (line[0] == '>') {
            out.emplace_back();
            cur = &out.back();
            char *p = line + 1, *q = p;
            while (*q && !isspace((unsigned char)*q)) ++q;
            cur->name.assign(p, q);
            while (*q && isspace((unsigned char)*q)) ++q;
            cur->comment = q;
        } else if (cur) cur->seq.append(line, n);
    }
    gzclose(fp);
    if (out.empty()) { g_ierr = std::string("no sequence in ") + fn; return false; }
    return true;
}

inline int nt4(int c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

// one group of the SNP file = consecutive lines with the same chrID (Index_src/hapmap.c:66-160)
struct SnpFile {
    std::vector<std::string> lines; size_t at = 0;
    std::vector<uint32_t> pos; std::vector<uint8_t> type;   // persistent arrays, stale tails included
    size_t alloc = 0; uint32_t n = 0; std::string chr;
    bool load(const char *fn)
    {
        FILE *f = fopen(fn, "r");
        if (!f) { g_ierr = std::string("cannot open ") + fn; return false; }
        char buf[128];                                       // TMP_SIZE: longer lines split, as there
        while (fgets(buf, sizeof buf, f)) lines.emplace_back(buf);
        fclose(f);
        return true;
    }
    static std::string tok(const std::string &s, int idx)
    {
        size_t b = 0; int k = 0;
        while (b <= s.size()) {
            while (b < s.size() && s[b] == '\t') ++b;      // strtok skips empty fields
            size_t e = s.find('\t', b); if (e == std::string::npos) e = s.size();
            if (k == idx) return s.substr(b, e - b);
            ++k; b = e + 1;
        }
        return "";
    }
    int next_group()                                         // hapmap_readhm
    {
        uint32_t cnt = 0;
        if (at < lines.size()) {
            std::string c0 = tok(lines[at], 0);
            cnt = 1;
            while (at + cnt < lines.size() && tok(lines[at + cnt], 0) == c0) ++cnt;
        }
        if (cnt > n) { pos.resize(cnt); type.resize(cnt); alloc = cnt; }   // realloc to the new count
        n = cnt;
        if (cnt == 0) return -1;
        chr = tok(lines[at], 0).substr(0, 31);
        for (uint32_t i = 0; i < cnt; ++i) {
            const std::string &l = lines[at + i];
            pos[i] = (uint32_t)(atoi(tok(l, 1).c_str()) - 1);
            std::string al = tok(l, 2), rf = tok(l, 3);
            uint8_t t = 0;
            for (size_t j = 0; j < al.size(); j += 2) t |= (uint8_t)(1u << nt4(al[j]));
            t = (uint8_t)(t | (nt4(rf.empty() ? 'N' : rf[0]) << 4));
            type[i] = t;
        }
        at += cnt;
        return 0;
    }
};

const int OCC1[16] = { 0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4 };
inline int n_alleles(uint8_t t) { return OCC1[t & 15]; }
inline int allele_at(uint8_t t, int iter)                   // hapmap_get_snptype: iter-th set bit, 4 if none
{
    int m = t & 15;
    for (int b = 0; b < 4; ++b) if ((m >> b) & 1) { if (iter == 0) return b; --iter; }
    return 4;
}

// interleave running counts into the 2-bit BWT (bwt_bwtupdate_core)
void write_c_bwt_sa(const std::string &prefix, const std::vector<uint8_t> &text, const int32_t *SA, bool &ok)
{
    const uint32_t n = (uint32_t)text.size();
    uint32_t primary = 0, L2[5] = { 0, 0, 0, 0, 0 };
    for (uint32_t i = 0; i < n; ++i) ++L2[text[i] + 1];
    for (int i = 1; i <= 4; ++i) L2[i] += L2[i - 1];
    const uint32_t raw_words = (n + 15) / 16, n_occ = (n + 127) / 128 + 1;
    std::vector<uint32_t> buf((size_t)raw_words + (size_t)n_occ * 4, 0);
    uint32_t c[4] = { 0, 0, 0, 0 };
    size_t k = 0; uint32_t i = 0;                            // i: index in the $-removed BWT
    uint32_t word = 0;
    for (uint32_t r = 0; r <= n; ++r) {
        int32_t sa = SA[r];
        if (sa == 0) { primary = r; continue; }
        uint32_t sym = text[(uint32_t)sa - 1];
        if (i % 128 == 0) { memcpy(&buf[k], c, 16); k += 4; }
        word |= sym << ((~i & 15u) << 1);
        if (i % 16 == 15) { buf[k++] = word; word = 0; }
        ++c[sym]; ++i;
    }
    if (i % 16 != 0) buf[k++] = word;
    memcpy(&buf[k], c, 16); k += 4;
    std::vector<uint32_t> out; out.reserve(buf.size() + 5);
    out.push_back(primary); for (int j = 1; j <= 4; ++j) out.push_back(L2[j]);
    out.insert(out.end(), buf.begin(), buf.begin() + (long)k);
    ok = ok && write_file(prefix + ".C.bwt", out.data(), out.size() * 4);
    const uint32_t intv = 8, n_sa = (n + intv) / intv;
    std::vector<uint32_t> sa; sa.reserve(n_sa + 7);
    sa.push_back(primary); for (int j = 1; j <= 4; ++j) sa.push_back(L2[j]);
    sa.push_back(intv); sa.push_back(n);
    for (uint32_t j = 1; j < n_sa; ++j) sa.push_back((uint32_t)SA[(size_t)j * intv]);
    ok = ok && write_file(prefix + ".C.sa", sa.data(), sa.size() * 4);
}

} // namespace

extern "C" const char *salt_idx_last_error(void) { return g_ierr.c_str(); }

extern "C" int salt_idx_build(const char *fn_fa, const char *fn_snp, const char *prefix_c, int l_seed)
{
    const std::string prefix(prefix_c);
    std::vector<Contig> fa;
    if (!read_fasta(fn_fa, fa)) return -1;
    bool ok = true;
    // ---------------- C part: pac / ann / amb ----------------
    uint64_t l_pac = 0;
    for (auto &c : fa) l_pac += c.seq.size();
    if (l_pac == 0 || l_pac >= 0x7FFFFFF0ull) { g_ierr = "genome empty or too long for the 32-bit index"; return -1; }
    std::vector<uint8_t> text((size_t)l_pac);                 // 2-bit codes, N randomised like bntseq.c:222
    {
        srand48(11);
        std::string ann, amb; char tmp[256];
        struct Hole { uint64_t off; uint32_t len; char c; };
        std::vector<Hole> holes;
        snprintf(tmp, sizeof tmp, "%lld %d %u\n", (long long)l_pac, (int)fa.size(), 11u); ann += tmp;
        uint64_t off = 0, at = 0;
        for (auto &c : fa) {
            int n_ambs = 0, lasts = 0;
            for (size_t i = 0; i < c.seq.size(); ++i) {
                int ch = (unsigned char)c.seq[i], code = nt4(ch);
                if (code >= 4) {
                    if (lasts == ch && !holes.empty()) ++holes.back().len;
                    else { holes.push_back(Hole{ off + i, 1, (char)ch }); ++n_ambs; }
                    code = (int)(lrand48() & 3);
                }
                lasts = ch;
                text[(size_t)at++] = (uint8_t)code;
            }
            ann += "0 " + c.name; ann += " "; ann += c.comment.empty() ? "(null)" : c.comment; ann += "\n";
            snprintf(tmp, sizeof tmp, "%lld %d %d\n", (long long)off, (int)c.seq.size(), n_ambs); ann += tmp;
            off += c.seq.size();
        }
        snprintf(tmp, sizeof tmp, "%lld %d %u\n", (long long)l_pac, (int)fa.size(), (unsigned)holes.size()); amb += tmp;
        for (auto &h : holes) { snprintf(tmp, sizeof tmp, "%lld %d %c\n", (long long)h.off, (int)h.len, h.c); amb += tmp; }
        ok = ok && write_file(prefix + ".C.ann", ann.data(), ann.size()) && write_file(prefix + ".C.amb", amb.data(), amb.size());
        std::vector<uint8_t> pac((size_t)l_pac / 4 + 2, 0);
        for (uint64_t i = 0; i < l_pac; ++i) pac[(size_t)i >> 2] |= (uint8_t)(text[(size_t)i] << ((3 - (i & 3)) << 1));
        size_t nbytes = (size_t)(l_pac >> 2) + ((l_pac & 3) ? 1 : 0);
        if (l_pac % 4 == 0) pac[nbytes++] = 0;
        pac[nbytes++] = (uint8_t)(l_pac % 4);
        ok = ok && write_file(prefix + ".C.pac", pac.data(), nbytes);
        // .C.lkt
        std::vector<uint32_t> lkt((size_t)(1u << 24) + 2);
        lkt[0] = 12;
        salt_lkt_build(pac.data(), (uint32_t)l_pac, 12, lkt.data() + 1);
        ok = ok && write_file(prefix + ".C.lkt", lkt.data(), ((size_t)(1u << 24) + 2) * 4);
    }
    // ---------------- C part: BWT + SA ----------------
    {
        const int32_t n = (int32_t)l_pac + 1;
        std::vector<uint8_t> s((size_t)n);
        for (int32_t i = 0; i < n - 1; ++i) s[(size_t)i] = (uint8_t)(text[(size_t)i] + 1);
        s[(size_t)n - 1] = 0;
        std::vector<int32_t> SA((size_t)n);
        Sais<uint8_t>::run(s.data(), SA.data(), n, 5);
        write_c_bwt_sa(prefix, text, SA.data(), ok);
    }
    // ---------------- .R.seedLen ----------------
    { int32_t k = l_seed; ok = ok && write_file(prefix + ".R.seedLen", &k, 4); }
    // ---------------- local patterns (.lp) and the R text ----------------
    const int D = l_seed - 1;                                 // WIN_SNP_DISTANCE
    std::string lp;                                           // the .lp file
    std::vector<uint8_t> rtext;                               // symbols 0..4
    std::vector<uint32_t> sharp_pos;                          // header value of the record of every '#'
    {
        SnpFile hm;
        if (!hm.load(fn_snp)) return -1;
        srand48(11);                                          // R_bns_fasta2bntseq re-seeds (4bit_bntseq.c:227)
        uint32_t tot_l = 0, snp_tot = 0;
        char tmp[128];
        auto emit = [&](char ch) {
            lp.push_back(ch);
            if (ch == '\n') return;
            int code = ch == '#' ? 4 : nt4(ch);
            if (ch != '#' && code >= 4) code = (int)(lrand48() & 3);
            rtext.push_back((uint8_t)code);
        };
        for (auto &c : fa) {
            const int l = (int)c.seq.size();
            hm.next_group();
            const uint32_t snp_num = hm.n;
            if (snp_num == 0) continue;                      // tot_l is NOT advanced (localPattern.c:218-221)
            if (hm.chr != c.name) continue;                  // localPattern.c:224-227
            std::string seq = c.seq;                         // alleles are written in place
            uint32_t mid = 0;
            while (mid < snp_num) {
                uint32_t ws = mid, we = mid + 1;
                while (we <= snp_num) {                      // reads one past the group (stale entry) like :239
                    if (we >= hm.alloc) break;               // past the allocation: undefined there, "no SNP" here
                    if (hm.pos[we] - hm.pos[mid] > (uint32_t)D) break;
                    ++we;
                }
                int wn = (int)(we - ws);
                if (wn > 5) { ++mid; continue; }             // WIN_MAX_SNP_NUM
                int win_start = hm.pos[ws] > (uint32_t)D ? (int)(hm.pos[ws] - (uint32_t)D) : 0;
                if (ws > 0 && hm.pos[ws] - hm.pos[ws - 1] <= (uint32_t)D) win_start = (int)hm.pos[ws - 1] + 1;
                int win_end = (int)hm.pos[mid] + D < l ? (int)hm.pos[mid] + D : l - 1;
                int nseg = 1;
                for (int i = 0; i < wn; ++i) nseg *= n_alleles(hm.type[ws + (uint32_t)i]);
                const uint32_t hdr = hm.pos[mid] + tot_l + (uint32_t)D;
                snprintf(tmp, sizeof tmp, ">%d_%u\t%u\n", (int)snp_tot, (unsigned)nseg, hdr);
                lp += tmp;
                ++snp_tot;
                if (snp_tot == 1) { emit('#'); sharp_pos.push_back(hdr); }
                for (int i = 0; i < nseg; ++i) {
                    int kk = i, f1 = 1;
                    for (int j = 0; j < wn; ++j) {
                        uint8_t t = hm.type[ws + (uint32_t)j];
                        f1 *= n_alleles(t);
                        int f2 = f1 ? nseg / f1 : 0;
                        int ti = f2 ? kk / f2 : 0;
                        kk -= ti * f2;
                        uint32_t p = hm.pos[ws + (uint32_t)j];
                        if (p < seq.size()) seq[p] = "ACGTN"[allele_at(t, ti)];
                    }
                    for (int j = win_start; j <= win_end; ++j) emit(seq[(size_t)j]);
                    emit('#'); emit('\n');
                    sharp_pos.push_back(hdr);
                }
                ++mid;
            }
            tot_l += (uint32_t)l;
        }
        ok = ok && write_file(prefix + ".lp", lp.data(), lp.size());
    }
    // ---------------- R part: BWT / Occ / SA ----------------
    {
        const uint32_t n = (uint32_t)rtext.size();
        if (n == 0) { g_ierr = "no local pattern was generated (SNP file empty or chromosome names do not match)"; return -1; }
        std::vector<uint8_t> s((size_t)n + 1);
        for (uint32_t i = 0; i < n; ++i) s[i] = (uint8_t)(rtext[i] + 1);
        s[n] = 0;
        std::vector<int32_t> SA((size_t)n + 1);
        Sais<uint8_t>::run(s.data(), SA.data(), (int32_t)n + 1, 6);
        uint32_t cum[6] = { 0, 0, 0, 0, 0, 0 };
        for (uint32_t i = 0; i < n; ++i) ++cum[rtext[i] + 1];
        for (int i = 1; i <= 5; ++i) cum[i] += cum[i - 1];
        const uint32_t words = (n + 255) / 256 * 256 / 8;       // BWTResidentSizeInWord
        std::vector<uint32_t> code((size_t)words, 0);
        uint32_t inv_sa0 = 0, i = 0;
        for (uint32_t r = 0; r <= n; ++r) {
            if (SA[r] == 0) { inv_sa0 = r; continue; }
            uint32_t sym = rtext[(uint32_t)SA[r] - 1];
            code[i >> 3] |= sym << ((7u - (i & 7u)) * 4u);
            ++i;
        }
        std::vector<uint32_t> out;
        out.push_back(n); out.push_back(inv_sa0); for (int j = 1; j <= 5; ++j) out.push_back(cum[j]); out.push_back(words);
        out.insert(out.end(), code.begin(), code.end());
        ok = ok && write_file(prefix + ".R.backward.bwt", out.data(), out.size() * 4);
        // explicit Occ: 16-bit values every 256 symbols relative to 32-bit values every 65536
        const uint32_t n_val = (n + 255) / 256 + 1;
        const uint32_t occ_words = (n_val + 1) / 2 * 5, major_words = (n_val + 255) / 256 * 5;
        std::vector<uint32_t> occ((size_t)occ_words, 0), major((size_t)major_words, 0);
        {
            uint32_t run[5] = { 0, 0, 0, 0, 0 }, base[5] = { 0, 0, 0, 0, 0 };
            const uint64_t stored = (uint64_t)words * 8;
            for (uint32_t e = 0; e < n_val; ++e) {
                if (e % 256 == 0) { memcpy(base, run, sizeof run); for (int c = 0; c < 5; ++c) major[(size_t)(e / 256) * 5 + c] = base[c]; }
                for (int c = 0; c < 5; ++c) {
                    uint32_t v = run[c] - base[c];
                    if (e % 2 == 0) occ[(size_t)(e / 2) * 5 + c] |= v << 16; else occ[(size_t)(e / 2) * 5 + c] |= v & 0xFFFFu;
                    // the last word always gets both halves (4bit_bwt_gen.c:1146-1163): with an odd number of values its low half
                    // repeats the high one
                    if (e + 1 == n_val && e % 2 == 0) occ[(size_t)(e / 2) * 5 + c] |= v & 0xFFFFu;
                }
                for (uint32_t q = 0; q < 256; ++q) {
                    uint64_t p = (uint64_t)e * 256 + q;
                    if (p >= stored) break;
                    uint32_t sym = (code[(size_t)p >> 3] >> ((7u - (p & 7u)) * 4u)) & 15u;
                    if (sym < 5) ++run[sym];
                }
            }
        }
        std::vector<uint32_t> oo;
        oo.push_back(occ_words); oo.insert(oo.end(), occ.begin(), occ.end());
        oo.push_back(major_words); oo.insert(oo.end(), major.begin(), major.end());
        ok = ok && write_file(prefix + ".R.backward.occ", oo.data(), oo.size() * 4);
        // saValueSharp (Rbwt_gen_sa, direction -1): for the '#' that opens segment j the stored value is
        // (header value of the record holding the '#' two places further on) - (len(segment j) + 1);
        // the last segment's entry reads one past sharp2Ri_array there (undefined) -- 0 here.
        const uint32_t n_sharp_rows = n - cum[4] + 1;
        std::vector<uint32_t> rsa((size_t)n_sharp_rows, 0);
        {
            std::vector<uint32_t> isa_of_sharp;               // text offsets of '#', in order
            for (uint32_t p = 0; p < n; ++p) if (rtext[p] == 4) isa_of_sharp.push_back(p);
            std::vector<uint32_t> row_of((size_t)n + 1);
            for (uint32_t r = 0; r <= n; ++r) row_of[(uint32_t)SA[r]] = r;
            const size_t ns = isa_of_sharp.size();
            for (size_t j = 0; j + 1 < ns; ++j) {
                uint32_t row = row_of[isa_of_sharp[j]];
                uint32_t seg_len = isa_of_sharp[j + 1] - isa_of_sharp[j] - 1;
                uint32_t hdr = j + 2 < sharp_pos.size() ? sharp_pos[j + 2] : 0;
                rsa[row - cum[4] - 1] = hdr - (seg_len + 1);
            }
            // the walk's last step reads the '$' row as '#' (Rbwt_bwt2nt) and lands one row past the
            // table's real rows: the spare last slot receives header[1] - 1 (never read by `salt`)
            if (sharp_pos.size() > 1) rsa[n_sharp_rows - 1] = sharp_pos[1] - 1;
        }
        std::vector<uint32_t> so; so.push_back(n_sharp_rows); so.insert(so.end(), rsa.begin(), rsa.end());
        ok = ok && write_file(prefix + ".R.backward.sa", so.data(), so.size() * 4);
    }
    // ---------------- mixRef (.ref) ----------------
    {
        SnpFile hm;
        if (!hm.load(fn_snp)) return -1;
        std::vector<uint32_t> ref((size_t)(l_pac + 7) / 8, 0);
        uint32_t tot_l = 0;
        static const uint8_t M[6] = { 1, 2, 4, 8, 0, 0 };
        for (auto &c : fa) {
            for (size_t i = 0; i < c.seq.size(); ++i) {
                uint32_t p = tot_l + (uint32_t)i;
                // __clear_pac + __set_pac (mixRef.c:143-146): the i-th SNP GROUP is applied to the i-th contig whatever its name
                // (mixRef.c:149-152), so a contig without SNPs shifts the later groups one contig down; what such a group writes
                // past its contig's end is wiped here when the next contig is laid down
                ref[p >> 3] = (ref[p >> 3] & ~(15u << (4 * (p & 7u)))) | ((uint32_t)M[nt4((unsigned char)c.seq[i])] << (4 * (p & 7u)));
            }
            int rc = hm.next_group();
            if (rc == 0)
                for (uint32_t i = 0; i < hm.n; ++i) {
                    uint64_t p = (uint64_t)tot_l + hm.pos[i];
                    if (p < l_pac) ref[(size_t)p >> 3] |= (uint32_t)(hm.type[i] & 15u) << (4 * (p & 7u));
                }
            tot_l += (uint32_t)c.seq.size();
        }
        std::vector<uint32_t> out; out.push_back((uint32_t)l_pac); out.insert(out.end(), ref.begin(), ref.end());
        ok = ok && write_file(prefix + ".ref", out.data(), out.size() * 4);
    }
    return ok ? 0 : -1;
}
