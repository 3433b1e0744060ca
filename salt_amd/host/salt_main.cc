// salt_amd/host/salt_main.cc -- `salt [opts] <idx-prefix> <reads.fq[.gz]>`: the reference's command line
// (Align_src/aln.c:102-227), index files and SAM stream, with the per-batch work on the GPU(s).
//
// It is the batch driver of alnse_core (Align_src/alnse.c:1353-1480) re-done for a device:
//   reader thread : FASTQ(.gz) -> batches of N_SEQS = 100000 reads          (query_read_multiSeqs, aln.h:27)
//   one worker per GPU : salt_gpu_align_se on its batch                     (stands where alnse_core1 ran)
//   formatter threads (-t) : SAM text per read                              (aln_samse, sam.c:87-182)
//   writer : records in input order                                         (the puts() loop, alnse.c:1433-1439)
// Extra long options (not in the reference): --gpus N (default 1).
// Flags the reference parses but ignores stay ignored (-n -e -M -O -E -l -X); -p is rejected for now.
#include "../../include/salt_host.h"
#include <getopt.h>
#include <zlib.h>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

const int N_SEQS = 100000;

struct Batch {
    long seq_no = 0;
    std::vector<std::string> name, qual;
    std::vector<uint8_t> seqs;
    std::vector<uint32_t> offs{ 0 };
    std::vector<salt_result_t> res;
    std::string sam;
    int n() const { return (int)name.size(); }
};

inline uint8_t nt4(int c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return 4; }
}

bool gets_trim(gzFile fp, std::string &s, std::vector<char> &buf)
{
    s.clear();
    for (;;) {
        if (!gzgets(fp, buf.data(), (int)buf.size())) return !s.empty();
        size_t n = strlen(buf.data());
        bool eol = n && buf[n - 1] == '\n';
        while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
        s.append(buf.data(), n);
        if (eol) return true;
    }
}

std::unique_ptr<Batch> read_batch(gzFile fp, long seq_no, std::vector<char> &buf)
{
    auto b = std::make_unique<Batch>();
    b->seq_no = seq_no;
    std::string line, seq, plus, qual;
    while (b->n() < N_SEQS && gets_trim(fp, line, buf)) {
        if (line.empty() || line[0] != '@') continue;
        size_t e = 1;
        while (e < line.size() && !isspace((unsigned char)line[e])) ++e;
        std::string nm = line.substr(1, e - 1);
        if (nm.size() > 2 && nm[nm.size() - 2] == '/' && isdigit((unsigned char)nm.back())) nm.resize(nm.size() - 2);   // query.c:139-143
        if (!gets_trim(fp, seq, buf)) break;
        if (!gets_trim(fp, plus, buf)) break;
        if (!gets_trim(fp, qual, buf)) break;
        if (seq.empty()) continue;
        b->name.push_back(nm); b->qual.push_back(qual);
        for (char c : seq) b->seqs.push_back(nt4((unsigned char)c));
        b->offs.push_back((uint32_t)b->seqs.size());
    }
    if (b->n() == 0) return nullptr;
    return b;
}

void format_batch(const salt_index_t *ix, const salt_sam_opt_t *so, Batch &b, int n_threads)
{
    const int n = b.n();
    std::vector<std::string> part((size_t)n_threads);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t]() {
            std::vector<char> buf(1 << 16);
            int lo = (int)((long)n * t / n_threads), hi = (int)((long)n * (t + 1) / n_threads);
            std::string &out = part[(size_t)t];
            out.reserve((size_t)(hi - lo) * 400);
            for (int i = lo; i < hi; ++i) {
                const int L = (int)(b.offs[i + 1] - b.offs[i]);
                if ((size_t)L * 4 + 4096 > buf.size()) buf.resize((size_t)L * 4 + 4096);
                int w = salt_sam_se(ix, so, b.name[i].c_str(), b.seqs.data() + b.offs[i], L, b.qual[i].c_str(), &b.res[i], buf.data(), buf.size());
                if (w < 0) { fprintf(stderr, "[salt] SAM record too long for read %s\n", b.name[i].c_str()); exit(1); }
                out.append(buf.data(), (size_t)w);
                out.push_back('\n');
            }
        });
    for (auto &t : th) t.join();
    b.sam.clear();
    for (auto &p : part) b.sam += p;
}

double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + ts.tv_nsec * 1e-9; }

int usage()
{
    fprintf(stderr,
            "\nUsage:     salt [Options] <Index.prefix> <Read_mate1> [Read_mate2]\n\n"
            "Options:   -h, --help                   help\n"
            "           -t, --threads       <int>    host threads (SAM formatting)\n"
            "           -g, --group         <str>    read group id\n"
            "           -c, --xa_cigar               print cigar in XA fields [False]\n"
            "           -d, --md                     print tag NM and MD [False]\n"
            "           -r, --overlap       <int>    seed stride [seed length]\n"
            "           -v, --ref                    only seed on the primary reference\n"
            "           -s, --max_seed      <int>    max seed occ [50]\n"
            "           -m, --max_locate    <int>    max loci per strand [1000]\n"
            "           -p, --pe                     paired end mode (not available on the GPU path yet)\n"
            "               --gpus          <int>    GPUs to shard batches over [1]\n"
            "           (-n -e -l -a -b -M -O -E -X are accepted and ignored like in the reference)\n\n");
    return 1;
}

} // namespace

int main(int argc, char **argv)
{
    int n_threads = 1, n_gpus = 1, overlap = -1, pe = 0;
    salt_aln_opt_t ao; memset(&ao, 0, sizeof ao);
    ao.max_seed = 50; ao.max_locate = 1000; ao.max_hits = 5;             // aln.c:46-47, aln.h:133
    salt_sam_opt_t so; memset(&so, 0, sizeof so);
    std::string cmd;
    for (int i = 0; i < argc; ++i) { if (i) cmd += " "; cmd += argv[i]; }
    static const struct option lo[] = {
        { "threads", 1, 0, 't' }, { "num", 1, 0, 'n' }, { "help", 0, 0, 'h' }, { "pe", 0, 0, 'p' }, { "min_tlen", 1, 0, 'a' },
        { "max_tlen", 1, 0, 'b' }, { "group", 1, 0, 'g' }, { "sw", 0, 0, 'e' }, { "max_locate", 1, 0, 'm' }, { "max_seed", 1, 0, 's' },
        { "read_length", 1, 0, 'l' }, { "overlap", 1, 0, 'r' }, { "xa_cigar", 0, 0, 'c' }, { "md", 0, 0, 'd' }, { "ref", 0, 0, 'v' },
        { "mismatch", 1, 0, 'M' }, { "gapop", 1, 0, 'O' }, { "gapex", 1, 0, 'E' }, { "extend", 1, 0, 'X' }, { "gpus", 1, 0, 1000 }, { 0, 0, 0, 0 } };
    int c;
    while ((c = getopt_long(argc, argv, "t:n:hpa:b:g:em:s:l:cdr:vM:O:E:X:", lo, nullptr)) >= 0) {
        switch (c) {
        case 't': n_threads = atoi(optarg); break;
        case 'g': so.rg_id = optarg; break;
        case 's': ao.max_seed = (uint32_t)atoi(optarg); break;
        case 'm': ao.max_locate = (uint32_t)atoi(optarg); break;
        case 'c': so.print_xa_cigar = 1; break;
        case 'd': so.print_nm_md = 1; break;
        case 'v': ao.seed_only_ref = 1; break;
        case 'r': overlap = atoi(optarg); break;
        case 'p': pe = 1; break;
        case 1000: n_gpus = atoi(optarg); break;
        case 'h': return usage();
        case '?': fprintf(stderr, "[ERROR]: no arg %c\n", optopt); return 1;
        default: break;
        }
    }
    if (optind + 2 > argc) { fprintf(stderr, "[opt_parse]: index prefix and read file can't be omited!\n"); return 1; }
    if (pe) { fprintf(stderr, "[salt] paired-end mode is not available on the GPU path yet\n"); return 1; }
    if (n_threads < 1) n_threads = 1;
    if (n_gpus < 1) n_gpus = 1;
    const char *prefix = argv[optind], *fn_reads = argv[optind + 1];

    double t0 = now();
    fprintf(stderr, "[alnse_core]:  Reload index...\n");
    salt_index_t *ix = salt_index_load(prefix, 0);
    if (!ix) { fprintf(stderr, "[salt] %s\n", salt_host_last_error()); return 1; }
    ao.l_seed = salt_index_seed_len(ix);
    ao.l_overlap = overlap > 0 ? overlap : ao.l_seed;                      // aln.c:223
    std::vector<int> devs((size_t)n_gpus);
    for (int i = 0; i < n_gpus; ++i) devs[(size_t)i] = i;
    std::vector<salt_gpu_index_t *> gix((size_t)n_gpus, nullptr);
    if (salt_gpu_index_attach(salt_index_host_view(ix), 0, &gix[0])) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    if (salt_gpu_index_replicate(gix[0], devs.data(), n_gpus, gix.data())) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    std::vector<salt_gpu_ws_t *> ws((size_t)n_gpus, nullptr);
    for (int i = 0; i < n_gpus; ++i)
        if (salt_gpu_ws_create(gix[(size_t)i], N_SEQS, (uint64_t)N_SEQS * SALT_MAX_READ_LEN, &ws[(size_t)i])) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    fprintf(stderr, "%lf sec escaped.\n", now() - t0);
    t0 = now();

    gzFile fp = gzopen(fn_reads, "r");
    if (!fp) { fprintf(stderr, "[query_open]: file %s open fail!\n", fn_reads); return 1; }
    gzbuffer(fp, 1 << 20);

    {   // header (aln_samhead, sam.c:56-84)
        std::vector<char> hb(1 << 20);
        int w = salt_sam_header(ix, &so, hb.data(), hb.size());
        if (w < 0) { fprintf(stderr, "[salt] SAM header too large\n"); return 1; }
        fwrite(hb.data(), 1, (size_t)w, stdout);
        time_t tt = time(nullptr); struct tm *tmv = localtime(&tt);
        printf("@PG\tID:snpaln\tPN:snpaln\tCL:\"%s\"\tDS:%d-%d-%d\tVN:0.1beta\n", cmd.c_str(), tmv->tm_year + 1900, tmv->tm_mon + 1, tmv->tm_mday);
    }

    // ---- pipeline: reader -> per-GPU workers -> ordered writer ----
    std::mutex mu; std::condition_variable cv;
    std::deque<std::unique_ptr<Batch>> todo;          // read, not yet aligned
    std::deque<std::unique_ptr<Batch>> done;          // aligned + formatted, any order
    bool eof = false; long n_tot = 0; std::atomic<bool> failed{ false };
    const size_t max_inflight = (size_t)n_gpus * 2 + 1;
    size_t inflight = 0;

    std::thread reader([&]() {
        std::vector<char> buf(1 << 16);
        long seq_no = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return inflight < max_inflight || failed; }); if (failed) break; }
            auto b = read_batch(fp, seq_no, buf);
            std::unique_lock<std::mutex> lk(mu);
            if (!b) { eof = true; cv.notify_all(); break; }
            ++seq_no; ++inflight;
            todo.push_back(std::move(b));
            cv.notify_all();
        }
    });
    std::vector<std::thread> workers;
    const int fmt_threads = n_threads / n_gpus > 0 ? n_threads / n_gpus : 1;
    for (int g = 0; g < n_gpus; ++g)
        workers.emplace_back([&, g]() {
            for (;;) {
                std::unique_ptr<Batch> b;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !todo.empty() || eof || failed; });
                    if (failed || (todo.empty() && eof)) break;
                    b = std::move(todo.front()); todo.pop_front();
                }
                b->res.resize((size_t)b->n());
                if (salt_gpu_align_se(ws[(size_t)g], &ao, (uint32_t)b->n(), b->seqs.data(), b->offs.data(), b->res.data())) {
                    fprintf(stderr, "[salt] %s\n", salt_gpu_last_error());
                    failed = true; cv.notify_all(); break;
                }
                format_batch(ix, &so, *b, fmt_threads);
                std::unique_lock<std::mutex> lk(mu);
                done.push_back(std::move(b));
                cv.notify_all();
            }
        });
    long next = 0;
    for (;;) {
        std::unique_ptr<Batch> b;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] {
                if (failed) return true;
                for (auto &d : done) if (d->seq_no == next) return true;
                return eof && inflight == 0;
            });
            if (failed) break;
            for (auto it = done.begin(); it != done.end(); ++it) if ((*it)->seq_no == next) { b = std::move(*it); done.erase(it); break; }
            if (!b) break;                                   // eof and nothing in flight
        }
        fwrite(b->sam.data(), 1, b->sam.size(), stdout);
        n_tot += b->n(); ++next;
        fprintf(stderr, "%ld reads have been aligned!\n", n_tot);
        { std::unique_lock<std::mutex> lk(mu); --inflight; cv.notify_all(); }
    }
    reader.join();
    for (auto &w : workers) w.join();
    fflush(stdout);
    double dt = now() - t0;
    fprintf(stderr, "[alnse_core]: total %lf sec escaped\n", dt);
    fprintf(stderr, "[salt] %ld reads, %.3f Mreads/s end to end (FASTQ -> SAM, %d GPU(s), %d host thread(s))\n", n_tot, dt > 0 ? n_tot / dt / 1e6 : 0.0, n_gpus, n_threads);
    gzclose(fp);
    for (int i = 0; i < n_gpus; ++i) salt_gpu_ws_destroy(ws[(size_t)i]);
    for (int i = n_gpus - 1; i >= 0; --i) salt_gpu_index_detach(gix[(size_t)i]);
    salt_index_free(ix);
    return failed ? 1 : 0;
}
