// salt_amd/host/polish_main.cc -- `polish [-p] <index.prefix> <SAM>`: the reference's SAM post-processor (row N4;
// Polish_src/polish.c:448-816) with its re-scoring on the GPU.
//
// What the reference does per record (pair): the primary hit and the XA hits are turned into genome offsets, sorted, made unique and
// re-scored against the 2-bit genome by plain edit distance (stock Landau-Vishkin, k = 13); the best hit becomes the alignment (MAPQ 60
// when it is the only scored hit, else 0), pairs prefer a hit pair 350..650 bases apart, the CIGAR of the winner is generated and a
// bare SAM record printed.  Here the host parses and prints; every edit distance and every CIGAR is computed by k_polish through
// salt_gpu_polish_lv (two device calls per batch of records: all hits, then the winners).  No CPU re-scoring path exists.
// Quirks kept (each cited at its place): strtok-style field splitting, only the first optional field containing "XA" is read, the
// window length that shrinks for good at the genome end, the tab behind QUAL in two of four cases, the second record of a pair carrying
// the first mate's name, an empty line ending the input.  `-s` re-scores by Smith-Waterman instead (+2 / -2, N 0, gaps 3 / 1): the same
// two device calls go to salt_gpu_polish_sw, which runs the mate-rescue kernel k_sw with polish's matrix; soft clips come from the
// alignment's read span (polish.c:209-222).
#include "../../include/salt_host.h"
#include <getopt.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

const int UNMAPPED = -100000, MAX_DISTANCE = 13;               // polish.c:447, 150
const uint32_t MIN_ISIZE = 350, MAX_ISIZE = 650;               // polish.c:148-149

struct Hit { const char *chrom; uint32_t pos, offset; int score; uint32_t item; };
struct Rec {
    std::string line;                                           // the record's text, fields NUL-terminated in place
    const char *name = nullptr, *seq = nullptr, *qual = nullptr;
    int flag = 0, l_seq = 0, strand = -1, primary = -1, b0 = UNMAPPED, b1 = UNMAPPED;
    uint32_t read = 0;                                          // index of its codes in the batch
    std::vector<Hit> h[2];
    std::string cigar;
};

// strtok(s, delim) with its state in `save`: skips leading delimiters, so empty fields vanish (samParser.c:100-125)
char *tok(char *s, char delim, char **save)
{
    if (!s) s = *save;
    if (!s) return nullptr;
    while (*s == delim) ++s;
    if (!*s) { *save = nullptr; return nullptr; }
    char *e = strchr(s, delim);
    if (e) { *e = 0; *save = e + 1; } else *save = nullptr;
    return s;
}

bool parse(Rec &r)                                              // sam_readline (samParser.c:84-190)
{
    char *save = nullptr, *line = &r.line[0];
    r.name = tok(line, '\t', &save);
    char *f = tok(nullptr, '\t', &save); if (!f) return false;
    r.flag = atoi(f);
    char *chrom = tok(nullptr, '\t', &save), *pos = tok(nullptr, '\t', &save);
    if (!chrom || !pos) return false;
    if ((r.flag & 4) == 0 && strcmp(chrom, "*") != 0) r.h[(r.flag & 0x10) ? 1 : 0].push_back(Hit{ chrom, (uint32_t)strtoul(pos, nullptr, 10), 0, 0, 0 });
    for (int k = 0; k < 5; ++k) if (!tok(nullptr, '\t', &save)) return false;          // MAPQ CIGAR MRNM MPOS ISIZE
    r.seq = tok(nullptr, '\t', &save); if (!r.seq) return false;
    r.l_seq = (int)strlen(r.seq);
    r.qual = tok(nullptr, '\t', &save); if (!r.qual) return false;
    for (char *opt = tok(nullptr, '\t', &save); opt; opt = tok(nullptr, '\t', &save)) {
        if (!strstr(opt, "XA")) continue;
        // "XA:Z:chr,+pos,cigar,nd;..." -- the reference's nested strtok leaves nothing behind this field to be looked at (samParser.c:143-186)
        char *sv = nullptr, *multi = tok(opt, ':', &sv);
        multi = tok(nullptr, ':', &sv); multi = tok(nullptr, ':', &sv);
        while (multi && *multi) {
            char *semi = strchr(multi, ';');
            if (semi) *semi = 0;
            if (!*multi) break;
            char *s3 = nullptr, *achrom = tok(multi, ',', &s3), *apos = tok(nullptr, ',', &s3);
            if (!achrom || !apos) break;
            if (apos[0] != '-') r.h[0].push_back(Hit{ achrom, (uint32_t)strtoul(apos, nullptr, 10), 0, 0, 0 });
            else r.h[1].push_back(Hit{ achrom, (uint32_t)strtoul(apos + 1, nullptr, 10), 0, 0, 0 });
            if (!semi) break;
            multi = semi + 1;
        }
        break;
    }
    return true;
}

inline uint8_t code_of(int c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }

struct Ctx {
    salt_index_t *ix = nullptr; salt_gpu_polish_t *gp = nullptr;
    const uint8_t *pac = nullptr; uint64_t l_pac = 0;
    std::unordered_map<std::string, int64_t> contig_off;
};

// one batch: scores of every hit, the winners, their CIGARs, the records
bool run_batch(Ctx &C, std::vector<Rec> &recs, bool paired, bool use_sw)
{
    const uint32_t n = (uint32_t)recs.size();
    std::vector<uint8_t> codes; std::vector<uint32_t> offs(n + 1, 0);
    uint32_t max_l = 1;
    for (uint32_t i = 0; i < n; ++i) { offs[i + 1] = offs[i] + (uint32_t)recs[i].l_seq; max_l = std::max<uint32_t>(max_l, (uint32_t)recs[i].l_seq); }
    codes.resize(offs[n]);
    for (uint32_t i = 0; i < n; ++i) {
        Rec &r = recs[i]; r.read = i;
        uint8_t *d = codes.data() + offs[i];
        // the read as sequenced: SEQ of a reverse-strand record is its reverse complement (samParser.c:131-141)
        if (r.flag & 0x10) for (int j = 0; j < r.l_seq; ++j) { const uint8_t c = code_of((unsigned char)r.seq[r.l_seq - 1 - j]); d[j] = c < 4 ? (uint8_t)(3 - c) : c; }
        else for (int j = 0; j < r.l_seq; ++j) d[j] = code_of((unsigned char)r.seq[j]);
    }
    // ---- items: every unique hit of every record ----
    std::vector<salt_polish_item_t> items; std::vector<uint8_t> pool; const uint32_t stride = (max_l + 15) / 8 * 8;
    for (Rec &r : recs) {
        int l_ref = r.l_seq;                                    // shrinks for good once a window is clipped (polish.c:466: l_refseq0 = __get_refseq(...))
        std::vector<uint8_t> buf((size_t)stride, 0);           // the reference's window buffer (calloc): what a clipped window leaves behind stays
        bool dirty = false; uint32_t prev_full = 0xFFFFFFFFu;   // offset of the last window written in full
        for (int s = 0; s < 2; ++s) {
            std::vector<Hit> &h = r.h[s];
            for (Hit &x : h) {
                auto it = C.contig_off.find(x.chrom);
                if (it == C.contig_off.end()) { fprintf(stderr, "[polish] sequence %s is not in the index\n", x.chrom); return false; }
                x.offset = (uint32_t)(it->second + x.pos - 1);
            }
            std::sort(h.begin(), h.end(), [](const Hit &a, const Hit &b) { return a.offset < b.offset; });
            h.erase(std::unique(h.begin(), h.end(), [](const Hit &a, const Hit &b) { return a.offset == b.offset; }), h.end());     // rm_repeat_hits
            for (Hit &x : h) {
                if ((uint64_t)x.offset > C.l_pac) { fprintf(stderr, "[Error]: Out of reference length!\n"); return false; }
                if ((uint64_t)x.offset + (uint64_t)l_ref > C.l_pac) l_ref = (int)(C.l_pac - x.offset);
                salt_polish_item_t itx; itx.read = r.read; itx.offset = x.offset; itx.pool = 0xFFFFFFFFu; itx.tlen = (uint16_t)l_ref; itx.strand = (uint8_t)s; itx.k = (uint8_t)MAX_DISTANCE;
                if (l_ref < r.l_seq && !use_sw) {               // rare: the window as the reference's buffer holds it -- fresh bases up to the clip, the bases of the
                    if (!dirty && prev_full != 0xFFFFFFFFu)     // last unclipped window behind it (the buffer is written in place, polish.c:84-92) -- handed over explicitly
                        for (int j = 0; j < r.l_seq; ++j) { const uint64_t l = (uint64_t)prev_full + (uint64_t)j; buf[(size_t)j] = (uint8_t)((C.pac[l >> 2] >> ((~l & 3) << 1)) & 3); }
                    for (int j = 0; j < l_ref; ++j) { const uint64_t l = (uint64_t)x.offset + (uint64_t)j; buf[(size_t)j] = (uint8_t)((C.pac[l >> 2] >> ((~l & 3) << 1)) & 3); }
                    itx.pool = (uint32_t)(pool.size() / stride);
                    pool.insert(pool.end(), buf.begin(), buf.end());
                    dirty = true;
                } else prev_full = x.offset;
                x.item = (uint32_t)items.size();
                items.push_back(itx);
            }
        }
    }
    std::vector<int32_t> dist(items.size(), -1);
    if (use_sw) {                                               // ssw_align flag 0: score1 of the forward pass (polish.c:509-512); always a score
        if (!items.empty() && salt_gpu_polish_sw(C.gp, codes.data(), offs.data(), n, items.data(), (uint32_t)items.size(), 0, dist.data(), nullptr, nullptr, nullptr)) {
            fprintf(stderr, "[polish] %s\n", salt_gpu_last_error()); return false;
        }
        for (Rec &r : recs) for (int s = 0; s < 2; ++s) for (Hit &x : r.h[s]) x.score = dist[x.item];
    } else {
    if (!items.empty() && salt_gpu_polish_lv(C.gp, codes.data(), offs.data(), n, items.data(), (uint32_t)items.size(), pool.data(), stride, (uint32_t)(pool.size() / stride), 0,
                                            dist.data(), nullptr, nullptr)) { fprintf(stderr, "[polish] %s\n", salt_gpu_last_error()); return false; }
    for (Rec &r : recs) for (int s = 0; s < 2; ++s) for (Hit &x : r.h[s]) x.score = dist[x.item] == -1 ? UNMAPPED : -dist[x.item];
    }
    // ---- winners ----
    auto pick = [](Rec &r) {                                    // polish.c:718-737
        int best0 = UNMAPPED, best1 = UNMAPPED;
        r.strand = r.primary = -1;
        for (int s = 0; s < 2; ++s)
            for (size_t j = 0; j < r.h[s].size(); ++j) {
                const int sc = r.h[s][j].score;
                if (sc == UNMAPPED) continue;
                if (sc > best1) { best1 = sc; if (best1 > best0) { std::swap(best0, best1); r.strand = s; r.primary = (int)j; } }
            }
        r.b0 = best0; r.b1 = best1;
    };
    auto pairing = [](std::vector<Hit> &fw, std::vector<Hit> &bw) -> unsigned {          // __pairing (polish.c:155-188)
        unsigned k = 0; size_t i = 0, j = 0;
        while (i < fw.size() && j < bw.size()) {
            const uint32_t a = fw[i].offset, b = bw[j].offset, d = a > b ? a - b : b - a;
            if (a > b || d < MIN_ISIZE) ++j;
            else if (d > MAX_ISIZE) ++i;
            else { std::swap(fw[k], fw[i]); std::swap(bw[k], bw[j]); ++i; ++j; ++k; }
        }
        return k;
    };
    std::vector<char> proper(n, 0);
    if (!paired) for (Rec &r : recs) pick(r);
    else
        for (uint32_t i = 0; i + 1 < n; i += 2) {
            Rec &a = recs[i], &b = recs[i + 1];
            const unsigned n0 = pairing(a.h[0], b.h[1]), n1 = pairing(b.h[0], a.h[1]);
            if (n0 + n1 == 0) { pick(a); pick(b); continue; }
            proper[i] = proper[i + 1] = 1;
            int best0 = UNMAPPED, best1 = UNMAPPED;             // polish.c:600-640
            a.strand = b.strand = a.primary = b.primary = -1;
            for (unsigned k = 0; k < n0; ++k) {
                const int sc = a.h[0][k].score + b.h[1][k].score;
                if (sc == UNMAPPED) continue;
                if (sc > best1) { best1 = sc; if (best1 > best0) { std::swap(best0, best1); a.strand = 0; b.strand = 1; a.primary = b.primary = (int)k; } }
            }
            for (unsigned k = 0; k < n1; ++k) {
                const int sc = a.h[1][k].score + b.h[0][k].score;
                if (sc == UNMAPPED) continue;
                if (sc > best1) { best1 = sc; if (best1 > best0) { std::swap(best0, best1); a.strand = 1; b.strand = 0; a.primary = b.primary = (int)k; } }
            }
            a.b0 = b.b0 = best0; a.b1 = b.b1 = best1;
        }
    // ---- CIGARs of the winners (gen_cigar, polish.c:190-249): a fresh window of l_seq bases, k = the winner's distance ----
    std::vector<salt_polish_item_t> citems; std::vector<uint32_t> owner;
    for (Rec &r : recs) {
        if (r.strand == -1 || r.primary == -1) continue;
        const Hit &x = r.h[r.strand][(size_t)r.primary];
        if (!use_sw && x.score == -MAX_DISTANCE) { r.cigar = "*"; continue; }             // polish.c:231-233
        if (!use_sw && x.score == UNMAPPED) { fprintf(stderr, "[polish] %s: the mate of a proper pair has no alignment within %d edits (the reference runs its CIGAR routine with k = 100000 here)\n", r.name, MAX_DISTANCE); return false; }
        salt_polish_item_t itx; itx.read = r.read; itx.offset = x.offset; itx.pool = 0xFFFFFFFFu; itx.strand = (uint8_t)r.strand; itx.k = use_sw ? 0 : (uint8_t)(-x.score);
        int l_ref = r.l_seq;
        if ((uint64_t)x.offset + (uint64_t)l_ref > C.l_pac) l_ref = (int)(C.l_pac - x.offset);
        itx.tlen = (uint16_t)l_ref;
        owner.push_back(r.read); citems.push_back(itx);
    }
    if (!citems.empty() && use_sw) {                            // ssw_align flag 2 with filters = the hit's score (polish.c:209-222)
        std::vector<int32_t> cd(citems.size()), span(2 * citems.size()); std::vector<uint16_t> cg(citems.size() * SALT_MAX_CIGAR_OPS), nc(citems.size());
        if (salt_gpu_polish_sw(C.gp, codes.data(), offs.data(), n, citems.data(), (uint32_t)citems.size(), 1, cd.data(), span.data(), cg.data(), nc.data())) {
            fprintf(stderr, "[polish] %s\n", salt_gpu_last_error()); return false;
        }
        for (size_t k = 0; k < citems.size(); ++k) {
            Rec &r = recs[owner[k]];
            if (cd[k] != r.h[r.strand][(size_t)r.primary].score) { fprintf(stderr, "push cigar error!\n"); return false; }      // polish.c:211-214
            char tmp[16];
            if (span[2 * k] != 0) { snprintf(tmp, sizeof tmp, "%dS", span[2 * k]); r.cigar += tmp; }
            for (int j = 0; j < nc[k]; ++j) { const uint16_t op = cg[k * SALT_MAX_CIGAR_OPS + (size_t)j]; snprintf(tmp, sizeof tmp, "%u%c", op >> 4, "MID"[op & 3]); r.cigar += tmp; }
            if (span[2 * k + 1] + 1 != r.l_seq) { snprintf(tmp, sizeof tmp, "%dS", r.l_seq - span[2 * k + 1] - 1); r.cigar += tmp; }
        }
    } else if (!citems.empty()) {
        std::vector<int32_t> cd(citems.size()); std::vector<uint16_t> cg(citems.size() * SALT_MAX_CIGAR_OPS); std::vector<uint8_t> nc(citems.size());
        if (salt_gpu_polish_lv(C.gp, codes.data(), offs.data(), n, citems.data(), (uint32_t)citems.size(), nullptr, stride, 0, 1, cd.data(), cg.data(), nc.data())) {
            fprintf(stderr, "[polish] %s\n", salt_gpu_last_error()); return false;
        }
        for (size_t k = 0; k < citems.size(); ++k) {
            Rec &r = recs[owner[k]];
            if (cd[k] != (int)citems[k].k) { fprintf(stderr, "push cigar error!\n"); return false; }      // polish.c:238-241
            char tmp[16];
            for (int j = 0; j < nc[k]; ++j) { const uint16_t op = cg[k * SALT_MAX_CIGAR_OPS + (size_t)j]; snprintf(tmp, sizeof tmp, "%u%c", op >> 4, "MID"[op & 3]); r.cigar += tmp; }
        }
    }
    // ---- records (polish_sam_se / polish_sam_pe, polish.c:251-445) ----
    std::string out;
    auto seq_qual = [&](const Rec &r) {
        const uint8_t *d = codes.data() + offs[r.read];
        // the winner's strand (an unmapped read prints its reverse complement: strand == -1 takes the `else` of polish.c:283)
        if (r.strand == 0) for (int j = 0; j < r.l_seq; ++j) out += "ACGTN"[d[j] > 4 ? 4 : d[j]];
        else for (int j = r.l_seq - 1; j >= 0; --j) { const uint8_t c = d[j]; out += "ACGTN"[c < 4 ? 3 - c : 4]; }
        out += '\t';
        const bool rev_in = (r.flag & 0x10) != 0;
        if ((rev_in && r.strand == 0) || (!rev_in && r.strand != 0)) for (int j = r.l_seq - 1; j >= 0; --j) out += r.qual[j];
        else { out += r.qual; out += '\t'; }                    // the tab of printf("%s\t", s) (polish.c:289,292)
        out += '\n';
    };
    char num[64];
    if (!paired)
        for (const Rec &r : recs) {
            const bool mapped = r.strand != -1;
            out += r.name; out += '\t';
            snprintf(num, sizeof num, "%u\t", 0x40u | (r.strand == 1 ? 0x10u : 0u) | (mapped ? 0u : 4u)); out += num;
            if (!mapped) out += "*\t0\t"; else { const Hit &x = r.h[r.strand][(size_t)r.primary]; out += x.chrom; snprintf(num, sizeof num, "\t%u\t", x.pos); out += num; }
            out += (r.b1 == UNMAPPED && r.b0 != UNMAPPED) ? "60\t" : "0\t";
            if (mapped) { out += r.cigar; out += '\t'; } else out += "*\t";
            out += "*\t0\t0\t";
            seq_qual(r);
        }
    else
        for (uint32_t i = 0; i + 1 < n; i += 2)
            for (int k = 0; k < 2; ++k) {
                const Rec &me = recs[i + (uint32_t)k], &mate = recs[i + 1 - (uint32_t)k];
                const bool m0 = me.strand != -1, m1 = mate.strand != -1;
                const Hit *x0 = m0 ? &me.h[me.strand][(size_t)me.primary] : nullptr, *x1 = m1 ? &mate.h[mate.strand][(size_t)mate.primary] : nullptr;
                unsigned flag = 1u | (proper[i] ? 2u : 0u) | (me.strand == 1 ? 0x10u : 0u) | (mate.strand == 1 ? 0x20u : 0u) | (k == 0 ? 0x40u : 0x80u) | (m0 ? 0u : 4u);
                if (!m1) flag |= k == 0 ? 8u : 4u;              // the second record marks itself unmapped when EITHER mate is (polish.c:383-384)
                out += recs[i].name; out += '\t';               // both records print the first mate's name (polish.c:378)
                snprintf(num, sizeof num, "%u\t", flag & 0xFFu); out += num;
                if (!m0) out += "*\t0\t"; else { out += x0->chrom; snprintf(num, sizeof num, "\t%u\t", x0->pos); out += num; }
                out += (me.b1 == UNMAPPED && me.b0 != UNMAPPED) ? "60\t" : "0\t";
                if (m0) { out += me.cigar; out += '\t'; } else out += "*\t";
                if (!m1) out += "*\t0\t";
                else if (!m0 || strcmp(x0->chrom, x1->chrom) != 0) { out += x1->chrom; snprintf(num, sizeof num, "\t%u\t", x1->pos); out += num; }
                else { snprintf(num, sizeof num, "=\t%u\t", x1->pos); out += num; }
                if (m0 && m1) { const int a = (int)(x0->pos < x1->pos ? x1->pos - x0->pos : x0->pos - x1->pos); snprintf(num, sizeof num, "%d\t", me.strand == 0 ? a : -a); out += num; }
                else out += "0\t";
                seq_qual(me);
            }
    fwrite(out.data(), 1, out.size(), stdout);
    return true;
}

int usage()
{
    fprintf(stderr, "\npolish  [OPT]  <index.prefix>  <SAM>\n\nOPT:    -h, --help  print help\n        -p, --pe    paired end mode\n"
                    "        -s, --sw    re-score by Smith-Waterman instead of edit distance\n\n");
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    static const struct option lo[] = { { "sw", 0, 0, 's' }, { "help", 0, 0, 'h' }, { "pe", 0, 0, 'p' }, { 0, 0, 0, 0 } };
    int c, paired = 0, use_sw = 0;
    while ((c = getopt_long(argc, argv, "shp", lo, nullptr)) >= 0) {
        if (c == 'p') paired = 1;
        else if (c == 's') use_sw = 1;
        else if (c == 'h') return usage();
        else { fprintf(stderr, "Unkown argument!\n"); usage(); return 1; }
    }
    if (argc - optind != 2) return usage();
    Ctx C;
    C.ix = salt_index_load(argv[optind], 0);
    if (!C.ix) { fprintf(stderr, "[polish] %s\n", salt_host_last_error()); return 1; }
    C.pac = salt_index_pac(C.ix, &C.l_pac);
    for (int i = 0; i < salt_index_n_seqs(C.ix); ++i) { int64_t off = 0; const char *nm = nullptr; salt_index_seq(C.ix, i, &off, nullptr, &nm); C.contig_off[nm] = off; }
    if (salt_gpu_polish_open(0, C.pac, C.l_pac, &C.gp)) { fprintf(stderr, "[polish] %s\n", salt_gpu_last_error()); return 1; }
    FILE *fp = fopen(argv[optind + 1], "r");
    if (!fp) { fprintf(stderr, "[Error]: Can't open file %s\n", argv[optind + 1]); return 1; }
    const size_t BATCH = 200000;                                // records per pair of device calls (even: pairs stay together)
    std::vector<Rec> recs;
    char *line = nullptr; size_t cap = 0; ssize_t got;
    bool in_header = true, ok = true, stop = false;
    while (!stop && (got = getline(&line, &cap, fp)) >= 0) {
        while (got > 0 && line[got - 1] == '\n') line[--got] = 0;
        if (in_header && line[0] == '@') continue;              // sam_skipHeader (samParser.c:43-55)
        in_header = false;
        if (got == 0) { stop = true; break; }                   // an empty line ends the input (samParser.c:87-90)
        recs.emplace_back();
        recs.back().line.assign(line, (size_t)got);
        if (!parse(recs.back())) { fprintf(stderr, "[polish] malformed SAM record: %.60s\n", line); ok = false; break; }
        if (recs.size() == BATCH) { if (!(ok = run_batch(C, recs, paired != 0, use_sw != 0))) break; recs.clear(); }
    }
    if (ok && paired && (recs.size() & 1)) recs.pop_back();     // a last record without its mate is dropped (polish.c:455-456, 652-653)
    if (ok && !recs.empty()) ok = run_batch(C, recs, paired != 0, use_sw != 0);
    free(line); fclose(fp);
    salt_gpu_polish_close(C.gp);
    salt_index_free(C.ix);
    return ok ? 0 : 1;
}
