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
// The reference builds its BWTs incrementally (BWT-SW); a BWT is canonical, so here both are derived from suffix
// arrays.  Two suffix-sorting backends produce the same bytes: the device builder of libsalt_gpu.so
// (salt_gpu_idx_build_c / _r: prefix doubling over a radix sort, texts up to 2^32 symbols -- what makes GRCh38-scale
// genomes indexable; handed in through salt_idx_backend_t) and SA-IS on the host (written from the published algorithm;
// 32-bit or 64-bit indices by text length) for machines without a GPU.  Everything around the sort -- packing, local
// patterns, Occ tables, mixRef -- runs on host threads.
// `salt` never reads .R.forward.*, .R.pac/.rpac/.ann/.amb (rbwt.c:495-498); SALT_IDX_ALL_FILES (salt-idx --all-files) writes them too, so
// that the directory equals the reference indexer's file for file:
//   .R.pac/.R.rpac/.R.ann/.R.amb       R_bns_fasta2bntseq                 4bit_bntseq.c:170-295
//   .R.forward.bwt/.occ                Rbwt_bwt_bwtgen on .R.rpac         index1.c:162-167
//   .R.forward.sa                      Rbwt_gen_sa(direction=+1)          rbwt.c:424-475
#include "../../include/salt_host.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <algorithm>
#include <thread>
#include <functional>
#include <chrono>

namespace {

// ---------------------------------------------------------------------------------------------
// SA-IS (Nong, Zhang, Chan 2009), suffix array with index type I, generic symbol type.
// s[n-1] must be the unique smallest symbol (0).
// ---------------------------------------------------------------------------------------------
template <class T, class I>
struct Sais {
    static inline bool tget(const std::vector<uint8_t> &t, I i) { return (t[(size_t)i >> 3] >> (i & 7)) & 1; }
    static inline void tset(std::vector<uint8_t> &t, I i, bool b)
    {
        if (b) t[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7)); else t[(size_t)i >> 3] &= (uint8_t)~(1u << (i & 7));
    }
    static void buckets(const T *s, std::vector<I> &bkt, I n, I K, bool end)
    {
        std::fill(bkt.begin(), bkt.end(), 0);
        for (I i = 0; i < n; ++i) ++bkt[(size_t)s[i]];
        I sum = 0;
        for (I i = 0; i < K; ++i) { sum += bkt[(size_t)i]; bkt[(size_t)i] = end ? sum : sum - bkt[(size_t)i]; }
    }
    static void induce_l(const std::vector<uint8_t> &t, I *SA, const T *s, std::vector<I> &bkt, I n, I K)
    {
        buckets(s, bkt, n, K, false);
        for (I i = 0; i < n; ++i) {
            I j = SA[i] - 1;
            if (j >= 0 && !tget(t, j)) SA[bkt[(size_t)s[j]]++] = j;
        }
    }
    static void induce_s(const std::vector<uint8_t> &t, I *SA, const T *s, std::vector<I> &bkt, I n, I K)
    {
        buckets(s, bkt, n, K, true);
        for (I i = n - 1; i >= 0; --i) {
            I j = SA[i] - 1;
            if (j >= 0 && tget(t, j)) SA[--bkt[(size_t)s[j]]] = j;
        }
    }
    static void run(const T *s, I *SA, I n, I K)
    {
        if (n == 1) { SA[0] = 0; return; }
        std::vector<uint8_t> t((size_t)n / 8 + 1, 0);
        tset(t, n - 2, false); tset(t, n - 1, true);
        for (I i = n - 3; i >= 0; --i) tset(t, i, s[i] < s[i + 1] || (s[i] == s[i + 1] && tget(t, i + 1)));
        auto is_lms = [&](I i) { return i > 0 && tget(t, i) && !tget(t, i - 1); };
        std::vector<I> bkt((size_t)K);
        // stage 1: sort the LMS substrings
        buckets(s, bkt, n, K, true);
        for (I i = 0; i < n; ++i) SA[i] = -1;
        for (I i = 1; i < n; ++i) if (is_lms(i)) SA[--bkt[(size_t)s[i]]] = i;
        induce_l(t, SA, s, bkt, n, K);
        induce_s(t, SA, s, bkt, n, K);
        I n1 = 0;
        for (I i = 0; i < n; ++i) if (is_lms(SA[i])) SA[n1++] = SA[i];
        for (I i = n1; i < n; ++i) SA[i] = -1;
        I name = 0, prev = -1;
        for (I i = 0; i < n1; ++i) {
            I pos = SA[i];
            bool diff = false;
            for (I d = 0; d < n; ++d) {
                if (prev == -1 || s[pos + d] != s[prev + d] || tget(t, pos + d) != tget(t, prev + d)) { diff = true; break; }
                if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
            }
            if (diff) { ++name; prev = pos; }
            SA[n1 + (pos >> 1)] = name - 1;
        }
        for (I i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];
        // stage 2: solve the reduced problem
        I *SA1 = SA, *s1 = SA + n - n1;
        if (name < n1) Sais<I, I>::run(s1, SA1, n1, name);
        else for (I i = 0; i < n1; ++i) SA1[s1[i]] = i;
        // stage 3: induce the result
        buckets(s, bkt, n, K, true);
        for (I i = 1, j = 0; i < n; ++i) if (is_lms(i)) s1[j++] = i;
        for (I i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
        for (I i = n1; i < n; ++i) SA[i] = -1;
        for (I i = n1 - 1; i >= 0; --i) { I j = SA[i]; SA[i] = -1; SA[--bkt[(size_t)s[j]]] = j; }
        induce_l(t, SA, s, bkt, n, K);
        induce_s(t, SA, s, bkt, n, K);
    }
};

// full suffix array (empty suffix first) of codes[0..n), codes < sigma; sa_out[n + 1].  64-bit indices when n + 1 >= 2^31.
void cpu_suffix_array(const uint8_t *codes, uint64_t n, int sigma, uint32_t *sa_out)
{
    std::vector<uint8_t> s((size_t)n + 1);
    for (uint64_t i = 0; i < n; ++i) s[(size_t)i] = (uint8_t)(codes[i] + 1);
    s[(size_t)n] = 0;
    if (n + 1 < 0x7FFFFFF0ull) {
        std::vector<int32_t> SA((size_t)n + 1);
        Sais<uint8_t, int32_t>::run(s.data(), SA.data(), (int32_t)(n + 1), sigma + 1);
        for (uint64_t i = 0; i <= n; ++i) sa_out[i] = (uint32_t)SA[(size_t)i];
    } else {
        std::vector<int64_t> SA((size_t)n + 1);
        Sais<uint8_t, int64_t>::run(s.data(), SA.data(), (int64_t)(n + 1), sigma + 1);
        for (uint64_t i = 0; i <= n; ++i) sa_out[i] = (uint32_t)SA[(size_t)i];
    }
}

thread_local std::string g_ierr;
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool verbose() { static const bool v = getenv("SALT_IDX_VERBOSE") && atoi(getenv("SALT_IDX_VERBOSE")); return v; }

int n_threads()
{
    static const int n = [] {
        if (const char *e = getenv("SALT_IDX_THREADS")) { int v = atoi(e); if (v > 0) return v < 256 ? v : 256; }
        unsigned h = std::thread::hardware_concurrency();
        return (int)(h == 0 ? 1 : h > 32 ? 32 : h);
    }();
    return n;
}
// f(lo, hi, t) over [0, n) split into contiguous ranges, one per thread (ranges are multiples of `grain`)
void parallel_for(uint64_t n, uint64_t grain, const std::function<void(uint64_t, uint64_t, int)> &f)
{
    int T = n_threads();
    if (n < grain * 4 || T == 1) { f(0, n, 0); return; }
    const uint64_t units = (n + grain - 1) / grain;
    if ((uint64_t)T > units) T = (int)units;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) {
        const uint64_t lo = units * (uint64_t)t / (uint64_t)T * grain, hi = std::min(n, units * (uint64_t)(t + 1) / (uint64_t)T * grain);
        th.emplace_back([=, &f] { f(lo, hi, t); });
    }
    for (auto &x : th) x.join();
}

bool write_file(const std::string &fn, const void *p, size_t n, const char *mode = "wb")
{
    FILE *f = fopen(fn.c_str(), mode);
    if (!f) { g_ierr = "cannot write " + fn; return false; }
    bool ok = n == 0 || fwrite(p, 1, n, f) == n;
    fclose(f);
    if (!ok) g_ierr = "short write on " + fn;
    return ok;
}
// header words + a body, without concatenating them in memory
bool write_file2(const std::string &fn, const void *h, size_t hn, const void *p, size_t n, const void *p2 = nullptr, size_t n2 = 0, const void *p3 = nullptr, size_t n3 = 0)
{
    FILE *f = fopen(fn.c_str(), "wb");
    if (!f) { g_ierr = "cannot write " + fn; return false; }
    bool ok = (hn == 0 || fwrite(h, 1, hn, f) == hn) && (n == 0 || fwrite(p, 1, n, f) == n) && (n2 == 0 || fwrite(p2, 1, n2, f) == n2) && (n3 == 0 || fwrite(p3, 1, n3, f) == n3);
    fclose(f);
    if (!ok) g_ierr = "short write on " + fn;
    return ok;
}

struct OwnedContig { std::string name, comment, seq; };

bool read_fasta(const char *fn, std::vector<OwnedContig> &out)
{
    gzFile fp = gzopen(fn, "r");
    if (!fp) { g_ierr = std::string("cannot open ") + fn; return false; }
    gzbuffer(fp, 1 << 20);
    static thread_local char line[1 << 16];
    OwnedContig *cur = nullptr;
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

// ---------------------------------------------------------------------------------------------
// SNP groups.  One group of the SNP file = consecutive lines with the same chrID (Index_src/hapmap.c:66-160).  The reference keeps
// ONE pair of arrays for all groups and re-allocates it to each group's count, so the entry just past a group (read by the window
// loop, localPattern.c:239) is whatever an earlier, larger group left there: `alloc` and the stale entry carry that along.
// ---------------------------------------------------------------------------------------------
struct Group {
    std::string chr;
    std::vector<uint32_t> pos; std::vector<uint8_t> type;      // pos 0-based; type = allele mask | ref code << 4
    uint32_t n = 0, alloc = 0, stale_pos = 0; uint8_t stale_type = 0;
    bool valid = false;                                       // hapmap_readhm returned 0 (a group was there)
    inline uint32_t P(uint32_t i) const { return i < n ? pos[i] : stale_pos; }
    inline uint8_t  T(uint32_t i) const { return i < n ? type[i] : stale_type; }
};

std::string tok(const std::string &s, int idx)
{
    size_t b = 0; int k = 0;
    while (b <= s.size()) {
        while (b < s.size() && s[b] == '\t') ++b;              // strtok skips empty fields
        size_t e = s.find('\t', b); if (e == std::string::npos) e = s.size();
        if (k == idx) return s.substr(b, e - b);
        ++k; b = e + 1;
    }
    return "";
}

bool read_snp_groups(const char *fn, std::vector<Group> &groups)
{
    FILE *f = fopen(fn, "r");
    if (!f) { g_ierr = std::string("cannot open ") + fn; return false; }
    char buf[128];                                             // TMP_SIZE: longer lines split, as there
    std::string cur;
    while (fgets(buf, sizeof buf, f)) {
        const std::string l(buf);
        std::string c0 = tok(l, 0);
        if (groups.empty() || c0 != cur) { groups.emplace_back(); groups.back().chr = c0.substr(0, 31); cur = c0; }
        Group &g = groups.back();
        g.pos.push_back((uint32_t)(atoi(tok(l, 1).c_str()) - 1));
        std::string al = tok(l, 2), rf = tok(l, 3);
        uint8_t t = 0;
        for (size_t j = 0; j < al.size(); j += 2) t |= (uint8_t)(1u << nt4(al[j]));
        t = (uint8_t)(t | (nt4(rf.empty() ? 'N' : rf[0]) << 4));
        g.type.push_back(t);
    }
    fclose(f);
    return true;
}

// replays the reference's one persistent array pair over the groups in order (hapmap_readhm): sets n / alloc / the stale entry
void replay_group_arrays(std::vector<Group> &groups, size_t n_contigs)
{
    std::vector<uint32_t> P; std::vector<uint8_t> T;
    uint32_t prev_n = 0; size_t alloc = 0;
    if (groups.size() < n_contigs) groups.resize(n_contigs);   // contigs beyond the last group see an empty read
    for (auto &g : groups) {
        const uint32_t cnt = (uint32_t)g.pos.size();
        if (cnt > prev_n) { P.resize(cnt); T.resize(cnt); alloc = cnt; }     // realloc to the new count
        prev_n = cnt;
        g.n = cnt; g.valid = cnt != 0;
        if (cnt) { std::copy(g.pos.begin(), g.pos.end(), P.begin()); std::copy(g.type.begin(), g.type.end(), T.begin()); }
        g.alloc = (uint32_t)alloc;
        if (cnt < alloc) { g.stale_pos = P[cnt]; g.stale_type = T[cnt]; }
    }
}

const int OCC1[16] = { 0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4 };
inline int n_alleles(uint8_t t) { return OCC1[t & 15]; }
inline int allele_at(uint8_t t, int iter)                   // hapmap_get_snptype: iter-th set bit, 4 if none
{
    int m = t & 15;
    for (int b = 0; b < 4; ++b) if ((m >> b) & 1) { if (iter == 0) return b; --iter; }
    return 4;
}

struct ContigView { std::string name, comment; const char *seq; uint64_t len; };

// ---------------------------------------------------------------------------------------------
// Local patterns (ss_core_alt, localPattern.c:171-324).  One WINDOW per SNP `mid` of a group: the SNPs from mid up to D = k - 1 bases
// downstream, every allele combination of them written over the bases from D before mid (or just behind the previous SNP when that
// is within D) to D behind mid.  The reference walks the windows in order, writing alleles into the contig IN PLACE and drawing a
// random base (lrand48) for every N it emits.  Here: per-window sizes first, a prefix sum for the output offsets, then the windows
// are filled by threads.  A window's bytes depend on earlier windows only through (a) the stale array entry a window at the end of a
// group may pick up, whose in-place write later windows can see, (b) unsorted positions, (c) the lrand48 stream; such windows (and,
// for (a)/(b), every later window of the contig) are left to one ordered pass that replays the reference's loop literally.
// ---------------------------------------------------------------------------------------------
struct WinDesc { uint32_t ws, we; int64_t win_start, win_end; int nseg; uint32_t hdr; bool skip; };

inline WinDesc window_of(const Group &g, uint32_t mid, int64_t l, int D, uint32_t tot_l)
{
    WinDesc w; w.ws = mid; w.we = mid + 1;
    const uint32_t n = g.n;
    while (w.we <= n) {                                      // reads one past the group (stale entry) like :239
        if (w.we >= g.alloc) break;                          // past the allocation: undefined there, "no SNP" here
        if (g.P(w.we) - g.P(mid) > (uint32_t)D) break;
        ++w.we;
    }
    const int wn = (int)(w.we - w.ws);
    w.skip = wn > 5;                                         // WIN_MAX_SNP_NUM
    w.win_start = g.P(w.ws) > (uint32_t)D ? (int64_t)(g.P(w.ws) - (uint32_t)D) : 0;
    if (w.ws > 0 && g.P(w.ws) - g.P(w.ws - 1) <= (uint32_t)D) w.win_start = (int64_t)g.P(w.ws - 1) + 1;
    w.win_end = (int64_t)(int)g.P(mid) + D < l ? (int64_t)(int)g.P(mid) + D : l - 1;
    w.nseg = 1;
    if (!w.skip) for (int i = 0; i < wn; ++i) w.nseg *= n_alleles(g.T(w.ws + (uint32_t)i));
    w.hdr = g.P(mid) + tot_l + (uint32_t)D;
    return w;
}
inline int64_t seg_len_of(const WinDesc &w) { return w.win_end >= w.win_start ? w.win_end - w.win_start + 1 : 0; }
inline int n_digits(uint64_t v) { int k = 1; while (v >= 10) { v /= 10; ++k; } return k; }

struct LocalPatterns {
    std::vector<uint8_t> rtext;          // symbols 0..4 ('#' = 4)
    std::vector<uint32_t> sharp_off;     // offset of every '#'
    std::vector<uint32_t> sharp_hdr;     // header value of the record of every '#'
    std::vector<char> lp;                // the .lp text (when wanted)
};

bool local_patterns(const std::vector<ContigView> &fa, const std::vector<Group> &groups, int l_seed, bool want_lp, LocalPatterns &out)
{
    const int D = l_seed - 1;                                 // WIN_SNP_DISTANCE
    struct Job { size_t ci; uint32_t tot_l; uint64_t w0; uint32_t seq_from; bool sorted; };   // windows [w0, w0 + n) of contig ci; seq_from: first mid of the literal replay
    std::vector<Job> jobs;
    uint64_t n_win = 0; uint32_t tot_l = 0;
    for (size_t ci = 0; ci < fa.size(); ++ci) {
        const Group &g = groups[ci];
        if (g.n == 0) continue;                               // tot_l is NOT advanced (localPattern.c:218-221)
        if (g.chr != fa[ci].name) continue;                   // localPattern.c:224-227
        jobs.push_back(Job{ ci, tot_l, n_win, g.n, true });
        n_win += g.n;
        tot_l += (uint32_t)fa[ci].len;
    }
    // sizes
    std::vector<uint32_t> w_sym(n_win), w_sharp(n_win); std::vector<uint64_t> w_lp(want_lp ? n_win : 0);
    std::vector<uint8_t> w_flag(n_win);                       // 1: emitted (not skipped), 2: needs the ordered pass
    for (Job &jb : jobs) {
        const Group &g = groups[jb.ci];
        const int64_t l = (int64_t)fa[jb.ci].len;
        const char *seq = fa[jb.ci].seq;
        bool sorted = true;
        for (uint32_t i = 1; i < g.n && sorted; ++i) sorted = g.pos[i] > g.pos[i - 1];
        uint32_t first_stale = g.n;
        std::vector<uint32_t> fs((size_t)n_threads(), g.n);
        parallel_for(g.n, 4096, [&](uint64_t lo, uint64_t hi, int t) {
            for (uint64_t m = lo; m < hi; ++m) {
                const WinDesc w = window_of(g, (uint32_t)m, l, D, jb.tot_l);
                const uint64_t wi = jb.w0 + m;
                if (w.we > g.n && fs[(size_t)t] == g.n) fs[(size_t)t] = (uint32_t)m;
                if (w.skip) { w_sym[wi] = 0; w_sharp[wi] = 0; w_flag[wi] = 0; if (want_lp) w_lp[wi] = 0; continue; }
                const int64_t sl = seg_len_of(w);
                w_sym[wi] = (uint32_t)((uint64_t)w.nseg * (uint64_t)(sl + 1));
                w_sharp[wi] = (uint32_t)w.nseg;
                uint8_t fl = 1;
                for (int64_t j = w.win_start; j <= w.win_end; ++j) if (nt4((unsigned char)seq[j]) >= 4) { fl |= 2; break; }
                w_flag[wi] = fl;
                if (want_lp) w_lp[wi] = (uint64_t)w.nseg * (uint64_t)(sl + 2) + 1 /* > */ + 1 /* _ */ + (uint64_t)n_digits((uint64_t)w.nseg) + 1 /* tab */ + (uint64_t)n_digits(w.hdr) + 1 /* nl */;
            }
        });
        for (uint32_t v : fs) first_stale = std::min(first_stale, v);
        jb.sorted = sorted;
        jb.seq_from = !sorted ? 0u : first_stale >= g.n ? g.n : first_stale > 5 ? first_stale - 5 : 0u;
        for (uint32_t m = jb.seq_from; m < g.n; ++m) if (w_flag[jb.w0 + m] & 1) w_flag[jb.w0 + m] |= 2;
    }
    // offsets: the first emitted window also opens the text with a '#'
    std::vector<uint64_t> off_sym(n_win + 1), off_sharp(n_win + 1), off_lp(want_lp ? n_win + 1 : 0); std::vector<uint32_t> ord(n_win);
    {
        uint64_t a = 0, b = 0, c = 0; uint32_t k = 0; bool first = true;
        for (uint64_t w = 0; w < n_win; ++w) {
            off_sym[w] = a; off_sharp[w] = b; ord[w] = k; if (want_lp) off_lp[w] = c;
            if (w_flag[w] & 1) {
                a += w_sym[w] + (first ? 1 : 0); b += w_sharp[w] + (first ? 1 : 0);
                if (want_lp) c += w_lp[w] + (uint64_t)n_digits(k) + (first ? 1 : 0);
                first = false; ++k;
            }
        }
        off_sym[n_win] = a; off_sharp[n_win] = b; if (want_lp) off_lp[n_win] = c;
        if (a >= 0xFFFFFFF0ull) { g_ierr = "local-pattern text too long for the 32-bit R index"; return false; }
    }
    out.rtext.resize(off_sym[n_win]); out.sharp_off.resize(off_sharp[n_win]); out.sharp_hdr.resize(off_sharp[n_win]);
    if (want_lp) out.lp.resize(off_lp[n_win]);
    uint64_t first_win = n_win;
    for (uint64_t w = 0; w < n_win; ++w) if (w_flag[w] & 1) { first_win = w; break; }

    // one window.  `scratch`: the contig copy the literal replay writes alleles into (nullptr: stateless); draws lrand48 for N when `rnd`
    auto fill = [&](const Job &jb, uint32_t mid, char *scratch, int64_t scratch_base, bool rnd) {
        const Group &g = groups[jb.ci];
        const int64_t l = (int64_t)fa[jb.ci].len;
        const char *seq = fa[jb.ci].seq;
        const uint64_t wi = jb.w0 + mid;
        const WinDesc w = window_of(g, mid, l, D, jb.tot_l);
        if (w.skip) return;
        const int wn = (int)(w.we - w.ws);
        uint64_t o = off_sym[wi], so = off_sharp[wi];
        char *lp = want_lp ? out.lp.data() + off_lp[wi] : nullptr;
        if (lp) lp += snprintf(lp, 64, ">%d_%u\t%u", (int)ord[wi], (unsigned)w.nseg, w.hdr), *lp++ = '\n';   // the byte snprintf's NUL takes is the newline's
        if (wi == first_win) { out.rtext[o] = 4; out.sharp_off[so] = (uint32_t)o; out.sharp_hdr[so] = w.hdr; ++o; ++so; if (lp) *lp++ = '#'; }
        for (int i = 0; i < w.nseg; ++i) {
            int kk = i, f1 = 1; uint32_t spos[5]; char sal[5];
            for (int j = 0; j < wn; ++j) {
                const uint8_t t = g.T(w.ws + (uint32_t)j);
                f1 *= n_alleles(t);
                const int f2 = f1 ? w.nseg / f1 : 0;
                const int ti = f2 ? kk / f2 : 0;
                kk -= ti * f2;
                spos[j] = g.P(w.ws + (uint32_t)j); sal[j] = "ACGTN"[allele_at(t, ti)];
                if (scratch && (int64_t)spos[j] < l) scratch[(int64_t)spos[j] - scratch_base] = sal[j];
            }
            for (int64_t j = w.win_start; j <= w.win_end; ++j) {
                char ch;
                if (scratch) ch = scratch[j - scratch_base];
                else { ch = seq[j]; for (int q = 0; q < wn; ++q) if ((int64_t)spos[q] == j) ch = sal[q]; }
                int code = nt4((unsigned char)ch);
                if (code >= 4) code = rnd ? (int)(lrand48() & 3) : 0;
                out.rtext[o++] = (uint8_t)code;
                if (lp) *lp++ = ch;
            }
            out.rtext[o] = 4; out.sharp_off[so] = (uint32_t)o; out.sharp_hdr[so] = w.hdr; ++o; ++so;
            if (lp) { *lp++ = '#'; *lp++ = '\n'; }
        }
    };
    // threads: the windows that depend on nothing
    for (const Job &jb : jobs) {
        const Group &g = groups[jb.ci];
        parallel_for(g.n, 1024, [&](uint64_t lo, uint64_t hi, int) {
            for (uint64_t m = lo; m < hi; ++m) if (w_flag[jb.w0 + m] == 1) fill(jb, (uint32_t)m, nullptr, 0, false);
        });
    }
    // ordered pass: N-bearing windows (lrand48 stream, R_bns_fasta2bntseq re-seeds: 4bit_bntseq.c:227) and the literal replays
    srand48(11);
    for (const Job &jb : jobs) {
        const Group &g = groups[jb.ci];
        std::vector<char> scratch; int64_t base = 0;
        for (uint32_t m = 0; m < g.n; ++m) {
            if (!(w_flag[jb.w0 + m] & 2)) continue;
            if (m >= jb.seq_from) {
                if (scratch.empty()) {
                    const WinDesc w0 = window_of(g, jb.seq_from, (int64_t)fa[jb.ci].len, D, jb.tot_l);
                    base = !jb.sorted ? 0 : std::max<int64_t>(0, std::min<int64_t>(w0.win_start, (int64_t)g.P(jb.seq_from)));
                    scratch.assign(fa[jb.ci].seq + base, fa[jb.ci].seq + fa[jb.ci].len);
                }
                fill(jb, m, scratch.data(), base, true);
            } else fill(jb, m, nullptr, 0, true);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// C part on the host: BWT words with interleaved running counts (bwt_bwtupdate_core), SA samples, k-mer table
// ---------------------------------------------------------------------------------------------
int cpu_build_c(int, const uint8_t *text, uint64_t n, uint32_t intv, uint32_t *primary_out, uint32_t L2[5], uint32_t *buf, uint32_t *sa_out, uint32_t *lkt, uint32_t lkt_len)
{
    std::vector<uint32_t> SA((size_t)n + 1);
    cpu_suffix_array(text, n, 4, SA.data());
    uint32_t primary = 0;
    L2[0] = L2[1] = L2[2] = L2[3] = L2[4] = 0;
    for (uint64_t i = 0; i < n; ++i) ++L2[text[i] + 1];
    for (int i = 1; i <= 4; ++i) L2[i] += L2[i - 1];
    uint32_t c[4] = { 0, 0, 0, 0 };
    size_t k = 0; uint64_t i = 0;                            // i: index in the $-removed BWT
    uint32_t word = 0;
    for (uint64_t r = 0; r <= n; ++r) {
        const uint32_t sa = SA[(size_t)r];
        if (sa == 0) { primary = (uint32_t)r; continue; }
        const uint32_t sym = text[sa - 1];
        if (i % 128 == 0) { memcpy(&buf[k], c, 16); k += 4; }
        word |= sym << ((~i & 15u) << 1);
        if (i % 16 == 15) { buf[k++] = word; word = 0; }
        ++c[sym]; ++i;
    }
    if (i % 16 != 0) buf[k++] = word;
    memcpy(&buf[k], c, 16); k += 4;
    *primary_out = primary;
    const uint64_t n_sa = (n + intv) / intv;
    sa_out[0] = 0xFFFFFFFFu;
    for (uint64_t j = 1; j < n_sa; ++j) sa_out[j] = SA[(size_t)(j * intv)];
    // counts of every k-mer start + the A-padded tail suffixes, then prefix sums (LookUpTable.c:70-150)
    const uint64_t n_item = (1ull << (2 * lkt_len)) + 1; const uint32_t mask = (uint32_t)(n_item - 2);
    memset(lkt, 0, n_item * 4);
    uint32_t x = 0;
    for (uint64_t p = 0; p < n; ++p) { x = ((x << 2) & mask) | text[p]; if (p + 1 >= lkt_len) ++lkt[x + 1]; }
    for (uint32_t q = 0; q < lkt_len; ++q) { x = (x << 2) & mask; ++lkt[x + 1]; }
    for (uint64_t q = 1; q < n_item; ++q) lkt[q] += lkt[q - 1];
    return 0;
}

int cpu_build_r(int, const uint8_t *rtext, uint64_t n, const uint32_t *sharp_off, const uint32_t *sharp_hdr, uint64_t n_sharp, uint32_t cum4,
                uint32_t *inv_sa0_out, uint32_t *code, uint64_t code_words, uint32_t *rsa)
{
    std::vector<uint32_t> SA((size_t)n + 1);
    cpu_suffix_array(rtext, n, 5, SA.data());
    memset(code, 0, code_words * 4);
    uint32_t inv_sa0 = 0; uint64_t i = 0;
    for (uint64_t r = 0; r <= n; ++r) {
        if (SA[(size_t)r] == 0) { inv_sa0 = (uint32_t)r; continue; }
        const uint32_t sym = rtext[SA[(size_t)r] - 1];
        code[i >> 3] |= sym << ((7u - (i & 7u)) * 4u);
        ++i;
    }
    *inv_sa0_out = inv_sa0;
    std::vector<uint32_t> row_of((size_t)n + 1);
    for (uint64_t r = 0; r <= n; ++r) row_of[SA[(size_t)r]] = (uint32_t)r;
    memset(rsa, 0, (n_sharp + 1) * 4);
    for (uint64_t j = 0; j + 1 < n_sharp; ++j) {
        const uint32_t row = row_of[sharp_off[j]];
        const uint32_t seg_len = sharp_off[j + 1] - sharp_off[j] - 1;
        const uint32_t hdr = j + 2 < n_sharp ? sharp_hdr[j + 2] : 0;
        rsa[row - cum4 - 1] = hdr - (seg_len + 1);
    }
    return 0;
}
const char *cpu_last_error(void) { return "host suffix sorter failed"; }

const salt_idx_backend_t CPU_BACKEND = { 0, cpu_build_c, cpu_build_r, cpu_last_error };

// explicit Occ of a 4-bit BWT code: 16-bit values every 256 symbols relative to 32-bit values every 65536 (4bit_bwt_gen.c:1409-1459).
// Padding nibbles (zero) count as 'A' exactly as in the stored code.
void r_occ_tables(const std::vector<uint32_t> &code, uint64_t n, std::vector<uint32_t> &occ, std::vector<uint32_t> &major)
{
    const uint64_t n_val = (n + 255) / 256 + 1;
    const uint64_t occ_words = (n_val + 1) / 2 * 5, major_words = (n_val + 255) / 256 * 5;
    occ.assign((size_t)occ_words, 0); major.assign((size_t)major_words, 0);
    const uint64_t n_blk = n_val - 1;                   // 256-symbol blocks of the stored code
    std::vector<uint32_t> bc((size_t)(n_blk + 1) * 5, 0);      // running counts in front of every block
    const int T = n_threads();
    std::vector<std::vector<uint32_t>> part((size_t)T, std::vector<uint32_t>(5, 0));
    std::vector<std::pair<uint64_t, uint64_t>> rng((size_t)T, { 0, 0 });
    parallel_for(n_blk, 256, [&](uint64_t lo, uint64_t hi, int t) {
        uint32_t run[5] = { 0, 0, 0, 0, 0 };
        rng[(size_t)t] = { lo, hi };
        for (uint64_t b = lo; b < hi; ++b) {
            memcpy(&bc[(size_t)b * 5], run, sizeof run);        // relative to the thread's start; rebased below
            for (uint32_t q = 0; q < 32; ++q) {
                uint32_t wv = code[(size_t)(b * 32 + q)];
                for (int k = 0; k < 8; ++k) { const uint32_t s = (wv >> (28 - 4 * k)) & 15u; if (s < 5) ++run[s]; }
            }
        }
        memcpy(part[(size_t)t].data(), run, sizeof run);
    });
    uint32_t base[5] = { 0, 0, 0, 0, 0 };
    std::vector<std::vector<uint32_t>> start((size_t)T, std::vector<uint32_t>(5, 0));
    for (int t = 0; t < T; ++t) { if (rng[(size_t)t].second <= rng[(size_t)t].first) continue; memcpy(start[(size_t)t].data(), base, sizeof base); for (int c = 0; c < 5; ++c) base[c] += part[(size_t)t][(size_t)c]; }
    memcpy(&bc[(size_t)n_blk * 5], base, sizeof base);
    parallel_for(n_blk, 256, [&](uint64_t lo, uint64_t hi, int) {
        int t = 0; while (t < T && !(rng[(size_t)t].first <= lo && lo < rng[(size_t)t].second)) ++t;
        for (uint64_t b = lo; b < hi; ++b) {
            while (!(rng[(size_t)t].first <= b && b < rng[(size_t)t].second)) ++t;
            for (int c = 0; c < 5; ++c) bc[(size_t)b * 5 + (size_t)c] += start[(size_t)t][(size_t)c];
        }
    });
    parallel_for(n_val, 512, [&](uint64_t lo, uint64_t hi, int) {
        for (uint64_t e = lo; e < hi; ++e) {
            const uint32_t *run = &bc[(size_t)e * 5], *mb = &bc[(size_t)(e / 256 * 256) * 5];
            if (e % 256 == 0) for (int c = 0; c < 5; ++c) major[(size_t)(e / 256) * 5 + (size_t)c] = run[c];
            for (int c = 0; c < 5; ++c) {
                const uint32_t v = run[c] - mb[c];
                uint32_t &o = occ[(size_t)(e / 2) * 5 + (size_t)c];
                // two values per word, the even one in the high half; a thread owns whole words (ranges are multiples of 512)
                if (e % 2 == 0) o |= v << 16; else o |= v & 0xFFFFu;
                // the last word always gets both halves (4bit_bwt_gen.c:1146-1163): with an odd number of values its low half repeats the high one
                if (e + 1 == n_val && e % 2 == 0) o |= v & 0xFFFFu;
            }
        }
    });
}

// .R.ann / .R.amb from the local-pattern text as R_bns_fasta2bntseq reads it with kseq (4bit_bntseq.c:170-217, R_bns_dump :57-84): one
// "sequence" per record, name up to the first blank, the rest of the header line as comment; holes = runs of the SAME non-ACGT# letter
// inside a record
bool r_ann_amb(const std::vector<char> &lp, uint64_t l_pac, std::string &ann, std::string &amb)
{
    struct Hole { uint64_t off; uint32_t len; char c; };
    std::vector<Hole> holes;
    std::string body; char num[96];
    uint64_t offset = 0; uint32_t n_seqs = 0;
    size_t i = 0; const size_t N = lp.size();
    auto is_base = [](char ch) { return ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'a' || ch == 'c' || ch == 'g' || ch == 't' || ch == '#'; };
    while (i < N) {
        if (lp[i] != '>') { ++i; continue; }
        size_t e = i + 1; while (e < N && lp[e] != '\n') ++e;
        size_t ne = i + 1; while (ne < e && lp[ne] != ' ' && lp[ne] != '\t') ++ne;
        const std::string name(&lp[i + 1], ne - (i + 1));
        const std::string anno = ne < e ? std::string(&lp[ne + 1], e - (ne + 1)) : std::string("(null)");
        i = e < N ? e + 1 : N;
        uint32_t len = 0, n_ambs = 0; char lasts = 0;
        while (i < N && lp[i] != '>') {                         // the record's lines (a '>' only opens a record at a line's start)
            size_t le = i; while (le < N && lp[le] != '\n') ++le;
            for (size_t k = i; k < le; ++k) {
                const char ch = lp[k];
                if (ch == '\r' || ch == ' ' || ch == '\t') continue;      // kseq keeps graphic characters only
                if (!is_base(ch)) {
                    if (lasts == ch && !holes.empty()) ++holes.back().len;
                    else { holes.push_back(Hole{ offset + len, 1, ch }); ++n_ambs; }
                }
                lasts = ch; ++len;
            }
            i = le < N ? le + 1 : N;
        }
        snprintf(num, sizeof num, "%d ", 0); body += num; body += name;
        if (!anno.empty()) { body += ' '; body += anno; }
        body += '\n';
        snprintf(num, sizeof num, "%lld %d %d\n", (long long)offset, (int)len, (int)n_ambs); body += num;
        offset += len; ++n_seqs;
    }
    if (offset != l_pac) { g_ierr = "internal: local-pattern text and R text differ in length"; return false; }
    snprintf(num, sizeof num, "%lld %d %u\n", (long long)l_pac, (int)n_seqs, 11u);
    ann = std::string(num) + body;
    snprintf(num, sizeof num, "%lld %d %u\n", (long long)l_pac, (int)n_seqs, (unsigned)holes.size());
    amb = num;
    for (const Hole &h : holes) { snprintf(num, sizeof num, "%lld %d %c\n", (long long)h.off, (int)h.len, h.c); amb += num; }
    return true;
}

// ---------------------------------------------------------------------------------------------
// the build proper
// ---------------------------------------------------------------------------------------------
int build_core(const std::vector<ContigView> &fa, std::vector<Group> &groups, const std::string &prefix, int l_seed,
               const salt_idx_backend_t *be, int flags)
{
    if (!be) be = &CPU_BACKEND;
    if (l_seed < 2 || l_seed > 255) { g_ierr = "seed length out of range"; return -1; }
    bool ok = true;
    double t0 = now_s();
    uint64_t l_pac = 0;
    for (auto &c : fa) l_pac += c.len;
    if (l_pac == 0 || l_pac >= 0xFFFFFFE0ull) { g_ierr = "genome empty or too long for the 32-bit index"; return -1; }
    for (auto &c : fa) if (c.len >= 0x7FFFFFF0ull) { g_ierr = "contig too long (the reference's contig coordinates are int)"; return -1; }
    replay_group_arrays(groups, fa.size());
    std::vector<uint64_t> c_off(fa.size() + 1, 0);
    for (size_t i = 0; i < fa.size(); ++i) c_off[i + 1] = c_off[i] + fa[i].len;

    // ---------------- C part: text codes, pac / ann / amb ----------------
    std::vector<uint8_t> text((size_t)l_pac);                 // 2-bit codes, N randomised like bntseq.c:222
    struct Hole { uint64_t off; uint32_t len; char c; size_t contig; };
    std::vector<Hole> holes;
    {
        // codes by threads; ambiguous bases are collected per contig chunk and get their lrand48 draw in text order afterwards
        struct Amb { uint64_t off; char c; };
        for (size_t ci = 0; ci < fa.size(); ++ci) {
            const char *seq = fa[ci].seq; const uint64_t base = c_off[ci];
            std::vector<std::vector<Amb>> amb((size_t)n_threads());
            parallel_for(fa[ci].len, 1 << 16, [&](uint64_t lo, uint64_t hi, int t) {
                for (uint64_t i = lo; i < hi; ++i) {
                    const int code = nt4((unsigned char)seq[i]);
                    if (code >= 4) { amb[(size_t)t].push_back(Amb{ i, seq[i] }); text[(size_t)(base + i)] = 0; }
                    else text[(size_t)(base + i)] = (uint8_t)code;
                }
            });
            for (auto &v : amb) for (const Amb &a : v) {      // threads cover ascending ranges: this is text order
                if (!holes.empty() && holes.back().contig == ci && holes.back().c == a.c && holes.back().off + holes.back().len == base + a.off) ++holes.back().len;
                else holes.push_back(Hole{ base + a.off, 1, a.c, ci });
            }
        }
        srand48(11);
        for (const Hole &h : holes) for (uint32_t j = 0; j < h.len; ++j) text[(size_t)(h.off + j)] = (uint8_t)(lrand48() & 3);
        std::string ann, amb; char tmp[256];
        snprintf(tmp, sizeof tmp, "%lld %d %u\n", (long long)l_pac, (int)fa.size(), 11u); ann += tmp;
        size_t hi = 0;
        for (size_t ci = 0; ci < fa.size(); ++ci) {
            int n_ambs = 0;
            while (hi < holes.size() && holes[hi].contig == ci) { ++n_ambs; ++hi; }
            ann += "0 " + fa[ci].name; ann += " "; ann += fa[ci].comment.empty() ? "(null)" : fa[ci].comment; ann += "\n";
            snprintf(tmp, sizeof tmp, "%lld %d %d\n", (long long)c_off[ci], (int)fa[ci].len, n_ambs); ann += tmp;
        }
        snprintf(tmp, sizeof tmp, "%lld %d %u\n", (long long)l_pac, (int)fa.size(), (unsigned)holes.size()); amb += tmp;
        for (auto &h : holes) { snprintf(tmp, sizeof tmp, "%lld %d %c\n", (long long)h.off, (int)h.len, h.c); amb += tmp; }
        ok = ok && write_file(prefix + ".C.ann", ann.data(), ann.size()) && write_file(prefix + ".C.amb", amb.data(), amb.size());
        std::vector<uint8_t> pac((size_t)l_pac / 4 + 2, 0);
        parallel_for((l_pac + 3) / 4, 1 << 16, [&](uint64_t lo, uint64_t hi2, int) {
            for (uint64_t b = lo; b < hi2; ++b) {
                uint8_t v = 0;
                for (uint64_t q = 0; q < 4 && b * 4 + q < l_pac; ++q) v |= (uint8_t)(text[(size_t)(b * 4 + q)] << ((3 - q) << 1));
                pac[(size_t)b] = v;
            }
        });
        size_t nbytes = (size_t)(l_pac >> 2) + ((l_pac & 3) ? 1 : 0);
        if (l_pac % 4 == 0) pac[nbytes++] = 0;
        pac[nbytes++] = (uint8_t)(l_pac % 4);
        ok = ok && write_file(prefix + ".C.pac", pac.data(), nbytes);
    }
    if (verbose()) fprintf(stderr, "[salt-idx] text, pac, ann, amb: %.2f s (%llu bases, %zu contigs, %d threads)\n", now_s() - t0, (unsigned long long)l_pac, fa.size(), n_threads());
    // ---------------- C part: BWT + SA + k-mer table ----------------
    {
        t0 = now_s();
        const uint32_t intv = 8, lkt_len = 12;
        const uint64_t n = l_pac, bwt_words = (n + 15) / 16 + ((n + 127) / 128 + 1) * 4, n_sa = (n + intv) / intv, n_item = (1ull << (2 * lkt_len)) + 1;
        std::vector<uint32_t> bwt((size_t)bwt_words), sa((size_t)n_sa), lkt((size_t)n_item + 1);
        uint32_t primary = 0, L2[5];
        lkt[0] = lkt_len;
        if (be->build_c(be->device, text.data(), n, intv, &primary, L2, bwt.data(), sa.data(), lkt.data() + 1, lkt_len) != 0) {
            g_ierr = std::string("C index: ") + be->last_error(); return -1;
        }
        const uint32_t h5[5] = { primary, L2[1], L2[2], L2[3], L2[4] };
        ok = ok && write_file2(prefix + ".C.bwt", h5, sizeof h5, bwt.data(), bwt.size() * 4);
        const uint32_t h7[7] = { primary, L2[1], L2[2], L2[3], L2[4], intv, (uint32_t)n };
        ok = ok && write_file2(prefix + ".C.sa", h7, sizeof h7, sa.data() + 1, (sa.size() - 1) * 4);
        ok = ok && write_file(prefix + ".C.lkt", lkt.data(), lkt.size() * 4);
        if (verbose()) fprintf(stderr, "[salt-idx] C index (suffix array, BWT, samples, 12-mer table) + files: %.2f s\n", now_s() - t0);
    }
    // ---------------- .R.seedLen ----------------
    { int32_t k = l_seed; ok = ok && write_file(prefix + ".R.seedLen", &k, 4); }
    // ---------------- local patterns (.lp) and the R text ----------------
    t0 = now_s();
    LocalPatterns lpat;
    const bool want_lp = !(flags & SALT_IDX_NO_LP), need_lp = want_lp || (flags & SALT_IDX_ALL_FILES);      // (.R.ann / .R.amb are read off the text)
    if (!local_patterns(fa, groups, l_seed, need_lp, lpat)) return -1;
    if (want_lp) ok = ok && write_file(prefix + ".lp", lpat.lp.data(), lpat.lp.size());
    if (!(flags & SALT_IDX_ALL_FILES)) { std::vector<char>().swap(lpat.lp); }
    if (verbose()) fprintf(stderr, "[salt-idx] local patterns: %.2f s (%zu symbols, %zu segments)\n", now_s() - t0, lpat.rtext.size(), lpat.sharp_off.size());
    // ---------------- R part: BWT / Occ / SA ----------------
    {
        t0 = now_s();
        const uint64_t n = lpat.rtext.size();
        if (n == 0) { g_ierr = "no local pattern was generated (SNP file empty or chromosome names do not match)"; return -1; }
        const uint8_t *rtext = lpat.rtext.data();
        uint32_t cum[6] = { 0, 0, 0, 0, 0, 0 };
        {
            std::vector<std::vector<uint64_t>> cnt((size_t)n_threads(), std::vector<uint64_t>(8, 0));
            parallel_for(n, 1 << 16, [&](uint64_t lo, uint64_t hi, int t) { auto &c = cnt[(size_t)t]; for (uint64_t i = lo; i < hi; ++i) ++c[rtext[i]]; });
            for (auto &c : cnt) for (int s = 0; s < 5; ++s) cum[s + 1] += (uint32_t)c[(size_t)s];
            for (int i = 1; i <= 5; ++i) cum[i] += cum[i - 1];
        }
        const uint64_t words = (n + 255) / 256 * 256 / 8;       // BWTResidentSizeInWord
        const uint64_t n_sharp = lpat.sharp_off.size();
        if (n_sharp != (uint64_t)(cum[5] - cum[4])) { g_ierr = "internal: '#' count mismatch"; return -1; }
        std::vector<uint32_t> code((size_t)words, 0), rsa((size_t)n_sharp + 1, 0);
        uint32_t inv_sa0 = 0;
        if (be->build_r(be->device, rtext, n, lpat.sharp_off.data(), lpat.sharp_hdr.data(), n_sharp, cum[4], &inv_sa0, code.data(), words, rsa.data()) != 0) {
            g_ierr = std::string("R index: ") + be->last_error(); return -1;
        }
        // the walk's last step reads the '$' row as '#' (Rbwt_bwt2nt) and lands one row past the table's real rows: the spare last
        // slot receives header[1] - 1 (never read by `salt`); the last segment's own entry reads one past sharp2Ri_array in the
        // reference (undefined) -- 0 here
        if (n_sharp > 1) rsa[(size_t)n_sharp] = lpat.sharp_hdr[1] - 1;
        const uint32_t h8[8] = { (uint32_t)n, inv_sa0, cum[1], cum[2], cum[3], cum[4], cum[5], (uint32_t)words };
        ok = ok && write_file2(prefix + ".R.backward.bwt", h8, sizeof h8, code.data(), code.size() * 4);
        std::vector<uint32_t> occ, major;
        r_occ_tables(code, n, occ, major);
        const uint32_t ow = (uint32_t)occ.size(), mw = (uint32_t)major.size();
        ok = ok && write_file2(prefix + ".R.backward.occ", &ow, 4, occ.data(), occ.size() * 4, &mw, 4, major.data(), major.size() * 4);
        const uint32_t n_rows = (uint32_t)(n_sharp + 1);
        ok = ok && write_file2(prefix + ".R.backward.sa", &n_rows, 4, rsa.data(), rsa.size() * 4);
        if (verbose()) fprintf(stderr, "[salt-idx] R index (suffix array, BWT, Occ, '#' rows) + files: %.2f s\n", now_s() - t0);
        if (flags & SALT_IDX_ALL_FILES) {
            t0 = now_s();
            // .R.pac / .R.rpac: the R text and its reverse, two 4-bit symbols per byte, the first in the high nibble; the tail bytes of
            // bntseq's pac files (4bit_bntseq.c:262-287)
            std::vector<uint8_t> rev((size_t)n);
            parallel_for(n, 1 << 16, [&](uint64_t lo, uint64_t hi, int) { for (uint64_t i = lo; i < hi; ++i) rev[(size_t)i] = rtext[(size_t)(n - 1 - i)]; });
            for (int which = 0; which < 2; ++which) {
                const uint8_t *t = which ? rev.data() : rtext;
                std::vector<uint8_t> pac((size_t)(n / 2 + (n & 1)) + 2, 0);
                parallel_for((n + 1) / 2, 1 << 15, [&](uint64_t lo, uint64_t hi, int) {
                    for (uint64_t b = lo; b < hi; ++b) pac[(size_t)b] = (uint8_t)((t[(size_t)(2 * b)] << 4) | (2 * b + 1 < n ? t[(size_t)(2 * b + 1)] : 0));
                });
                size_t len = (size_t)(n / 2 + (n & 1));
                if (n % 2 == 0) pac[len++] = 0;
                pac[len++] = (uint8_t)(n % 2);
                ok = ok && write_file(prefix + (which ? ".R.rpac" : ".R.pac"), pac.data(), len);
            }
            std::string ann, amb;
            if (!r_ann_amb(lpat.lp, n, ann, amb)) return -1;
            ok = ok && write_file(prefix + ".R.ann", ann.data(), ann.size()) && write_file(prefix + ".R.amb", amb.data(), amb.size());
            // the forward index = the same construction on the reversed text.  Its '#' table (Rbwt_gen_sa, direction +1) gives the k-th '#'
            // of the R text, in text order, the header value of its own record.  The backend reports rows through the backward rule
            // (row of the j-th '#' <- hdr[j + 2] - (distance to the next '#'), last '#' not written), so it is handed headers that make
            // that value j + 1, and the rows are read back from there.
            std::vector<uint32_t> off2((size_t)n_sharp), hdr2((size_t)n_sharp, 0), fcode((size_t)words, 0), fidx((size_t)n_sharp + 1, 0), fsa((size_t)n_sharp + 1, 0);
            for (uint64_t j = 0; j < n_sharp; ++j) off2[(size_t)j] = (uint32_t)(n - 1 - lpat.sharp_off[(size_t)(n_sharp - 1 - j)]);
            for (uint64_t j = 0; j + 2 < n_sharp; ++j) hdr2[(size_t)j + 2] = (uint32_t)(j + 1) + (off2[(size_t)j + 1] - off2[(size_t)j]);
            uint32_t finv = 0;
            if (be->build_r(be->device, rev.data(), n, off2.data(), hdr2.data(), n_sharp, cum[4], &finv, fcode.data(), words, fidx.data()) != 0) {
                g_ierr = std::string("R forward index: ") + be->last_error(); return -1;
            }
            const uint32_t tail = n_sharp > 1 ? 0u - (off2[(size_t)n_sharp - 1] - off2[(size_t)n_sharp - 2]) : 0u;      // what the rule leaves for j = n_sharp - 2
            bool seen_last = false;
            for (uint64_t r = 0; r < n_sharp; ++r) {
                const uint32_t v = fidx[(size_t)r];
                uint64_t j;
                if (n_sharp > 1 && v == tail) j = n_sharp - 2;
                else if (v == 0) { if (seen_last) { g_ierr = "internal: forward '#' rows"; return -1; } seen_last = true; j = n_sharp - 1; }
                else j = (uint64_t)v - 1;
                if (j >= n_sharp) { g_ierr = "internal: forward '#' rows"; return -1; }
                fsa[(size_t)r] = lpat.sharp_hdr[(size_t)(n_sharp - 1 - j)];
            }
            const uint32_t fh8[8] = { (uint32_t)n, finv, cum[1], cum[2], cum[3], cum[4], cum[5], (uint32_t)words };
            ok = ok && write_file2(prefix + ".R.forward.bwt", fh8, sizeof fh8, fcode.data(), fcode.size() * 4);
            std::vector<uint32_t> focc, fmajor;
            r_occ_tables(fcode, n, focc, fmajor);
            const uint32_t fow = (uint32_t)focc.size(), fmw = (uint32_t)fmajor.size();
            ok = ok && write_file2(prefix + ".R.forward.occ", &fow, 4, focc.data(), focc.size() * 4, &fmw, 4, fmajor.data(), fmajor.size() * 4);
            ok = ok && write_file2(prefix + ".R.forward.sa", &n_rows, 4, fsa.data(), fsa.size() * 4);
            if (verbose()) fprintf(stderr, "[salt-idx] R text files and forward index: %.2f s\n", now_s() - t0);
        }
        { std::vector<char>().swap(lpat.lp); }
    }
    // ---------------- mixRef (.ref) ----------------
    {
        t0 = now_s();
        std::vector<uint32_t> ref((size_t)(l_pac + 7) / 8, 0);
        static const uint8_t M[6] = { 1, 2, 4, 8, 0, 0 };
        // __clear_pac + __set_pac (mixRef.c:143-146) lay a contig down; the i-th SNP GROUP is then ORed over the i-th contig whatever its
        // name (mixRef.c:149-152).  What a group writes past its contig's end is wiped when the next contig is laid down, so only the
        // SNPs inside their own contig (and, for the last contig, inside the genome) survive: lay everything down first, then OR those.
        for (size_t ci = 0; ci < fa.size(); ++ci) {
            const char *seq = fa[ci].seq; const uint64_t base = c_off[ci], len = fa[ci].len;
            // whole words by threads, the (at most two) words shared with the neighbouring contigs afterwards
            const uint64_t w_lo = (base + 7) / 8, w_hi = (base + len) / 8;
            auto put = [&](uint64_t p) { ref[(size_t)(p >> 3)] = (ref[(size_t)(p >> 3)] & ~(15u << (4 * (p & 7u)))) | ((uint32_t)M[nt4((unsigned char)seq[p - base])] << (4 * (p & 7u))); };
            if (w_hi > w_lo) {
                parallel_for(w_hi - w_lo, 1 << 14, [&](uint64_t lo, uint64_t hi, int) {
                    for (uint64_t w = w_lo + lo; w < w_lo + hi; ++w) {
                        uint32_t v = 0;
                        for (uint32_t q = 0; q < 8; ++q) v |= (uint32_t)M[nt4((unsigned char)seq[w * 8 + q - base])] << (4 * q);
                        ref[(size_t)w] = v;
                    }
                });
                for (uint64_t p = base; p < w_lo * 8; ++p) put(p);
                for (uint64_t p = w_hi * 8; p < base + len; ++p) put(p);
            } else for (uint64_t p = base; p < base + len; ++p) put(p);
        }
        for (size_t ci = 0; ci < fa.size(); ++ci) {
            const Group &g = groups[ci];
            if (!g.valid) continue;
            const uint64_t end = ci + 1 == fa.size() ? l_pac : c_off[ci + 1];
            for (uint32_t i = 0; i < g.n; ++i) {
                const uint64_t p = c_off[ci] + g.pos[i];
                if (p < end) ref[(size_t)p >> 3] |= (uint32_t)(g.type[i] & 15u) << (4 * (p & 7u));
            }
        }
        const uint32_t l32 = (uint32_t)l_pac;
        ok = ok && write_file2(prefix + ".ref", &l32, 4, ref.data(), ref.size() * 4);
        if (verbose()) fprintf(stderr, "[salt-idx] mixRef: %.2f s\n", now_s() - t0);
    }
    return ok ? 0 : -1;
}

} // namespace

extern "C" const char *salt_idx_last_error(void) { return g_ierr.c_str(); }

extern "C" int salt_idx_build_ex(const char *fn_fa, const char *fn_snp, const char *prefix_c, int l_seed, const salt_idx_backend_t *be, int flags)
{
    if (!fn_fa || !fn_snp || !prefix_c) { g_ierr = "null argument"; return -1; }
    std::vector<OwnedContig> own;
    if (!read_fasta(fn_fa, own)) return -1;
    std::vector<ContigView> fa;
    for (auto &c : own) fa.push_back(ContigView{ c.name, c.comment, c.seq.data(), c.seq.size() });
    std::vector<Group> groups;
    if (!read_snp_groups(fn_snp, groups)) return -1;
    return build_core(fa, groups, prefix_c, l_seed, be, flags);
}

extern "C" int salt_idx_build(const char *fn_fa, const char *fn_snp, const char *prefix_c, int l_seed)
{
    return salt_idx_build_ex(fn_fa, fn_snp, prefix_c, l_seed, nullptr, 0);
}

extern "C" int salt_idx_build_mem(const salt_idx_contig_t *contigs, int n_contigs, const salt_idx_snps_t *snps, int n_groups, const char *prefix_c,
                                  int l_seed, const salt_idx_backend_t *be, int flags)
{
    if (!contigs || n_contigs <= 0 || (n_groups > 0 && !snps) || !prefix_c) { g_ierr = "null argument"; return -1; }
    std::vector<ContigView> fa;
    for (int i = 0; i < n_contigs; ++i) {
        if (!contigs[i].name || !contigs[i].seq) { g_ierr = "contig without name or bases"; return -1; }
        fa.push_back(ContigView{ contigs[i].name, contigs[i].comment ? contigs[i].comment : "", contigs[i].seq, contigs[i].len });
    }
    std::vector<Group> groups((size_t)(n_groups > 0 ? n_groups : 0));
    for (int i = 0; i < n_groups; ++i) {
        Group &g = groups[(size_t)i];
        g.chr = std::string(snps[i].chr ? snps[i].chr : "").substr(0, 31);
        g.pos.assign(snps[i].pos, snps[i].pos + snps[i].n);
        g.type.resize(snps[i].n);
        for (uint32_t j = 0; j < snps[i].n; ++j) g.type[j] = (uint8_t)((snps[i].alleles[j] & 15u) | ((snps[i].ref ? snps[i].ref[j] & 7u : 4u) << 4));
    }
    return build_core(fa, groups, prefix_c, l_seed, be, flags);
}

// the host suffix sorter by itself (tests compare the device sorter with it)
extern "C" int salt_idx_suffix_array_cpu(const uint8_t *text, uint64_t n, int bits, uint32_t *sa_out)
{
    if (!text || !sa_out || (bits != 2 && bits != 3)) { g_ierr = "bad argument"; return -1; }
    cpu_suffix_array(text, n, 1 << bits, sa_out);
    return 0;
}
