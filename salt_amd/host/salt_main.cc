// salt_amd/host/salt_main.cc -- `salt [opts] <idx-prefix> <reads.fq[.gz]>`: the reference's command line
// (Align_src/aln.c:102-227), index files and SAM stream, with the per-batch work on the GPU(s).
//
// It is the batch driver of alnse_core (Align_src/alnse.c:1353-1480) re-done for a device:
//   reader thread : FASTQ(.gz) -> batches of N_SEQS = 100000 reads          (query_read_multiSeqs, aln.h:27)
//   one worker per GPU : salt_gpu_align_se on its batch                     (stands where alnse_core1 ran)
//   formatter threads (-t) : SAM text per read                              (aln_samse, sam.c:87-182)
//   writer : records in input order                                         (the puts() loop, alnse.c:1433-1439)
// Extra long options (not in the reference): --gpus N (default 1).
// Flags the reference parses but ignores stay ignored (-n -e -M -O -E -l -X).  -p <mate1> <mate2>: paired end
// (alnpe_core, Align_src/alnpe.c:530-661) through salt_gpu_align_pe.
#include "../../include/salt_host.h"
#include <fcntl.h>
#include <getopt.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

const int N_SEQS = 100000;

struct Batch {
    long seq_no = 0;
    std::vector<char> raw;                    // the FASTQ text; names and qualities are NUL-terminated in place
    std::vector<uint32_t> rec;                // offset of every record's '@' line in raw (found by the reader)
    std::vector<uint32_t> name, qual;         // offsets into raw
    std::vector<uint8_t> seqs;
    std::vector<uint32_t> offs{ 0 };
    std::vector<salt_result_t> res;
    std::vector<std::string> sam;                    // the batch's SAM text in order, one piece per formatting thread
    std::unique_ptr<Batch> mate;              // -p: the second file's records of the same pairs
    int n() const { return (int)name.size(); }
};

inline uint8_t nt4(int c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return 4; }
}

// True when every complete record in the first 256 KiB of the (possibly gzipped) file is strict 4-line FASTQ:
// '@' line, one sequence line, '+' line, one quality line of the same length.
static bool sniff_four_line(const char *fn)
{
    gzFile f = gzopen(fn, "r");
    if (!f) return true;
    std::vector<char> b(256u << 10);
    int n = gzread(f, b.data(), (unsigned)b.size());
    gzclose(f);
    if (n <= 0) return true;
    size_t p = 0; const size_t e = (size_t)n;
    if (b[0] != '@') return false;
    for (;;) {
        size_t st[5]; st[0] = p; bool whole = true;
        for (int k = 0; k < 4; ++k) {
            const char *nl = (const char *)memchr(b.data() + st[k], '\n', e - st[k]);
            if (!nl) { whole = false; break; }
            st[k + 1] = (size_t)(nl - b.data()) + 1;
        }
        if (!whole) return true;
        auto len = [&](int k) { size_t l = st[k + 1] - st[k] - 1; while (l > 0 && b[st[k] + l - 1] == '\r') --l; return l; };
        if (b[st[0]] != '@' || b[st[2]] != '+' || len(1) != len(3)) return false;
        p = st[4];
        if (p >= e) return true;
    }
}

// Raw text of up to N_SEQS FASTQ records.  The reader thread only finds record boundaries (4 lines per
// record, like the 4-line FASTQ the reference's test data uses); parsing runs on the worker threads.
// The general mode (chosen when the head of the file is not strict 4-line FASTQ, sniff_four_line) reads records the way
// the reference's kseq.h does -- sequence and quality may span lines -- and hands them on re-written as 4 lines.
struct RawReader {
    gzFile fp; std::vector<char> buf; size_t have = 0, pos = 0; bool eof = false; bool general = false;
    explicit RawReader(gzFile f, bool general_ = false) : fp(f), buf(8u << 20), general(general_) {}
    // kseq_read (kseq.h): '@name comment' line; sequence lines up to a line starting with '+'; that line; quality lines
    // until as many characters as the sequence has.  A '>' record (FASTA) has no quality: the reference cannot print it.
    int take_general(std::vector<char> &out, int n_rec, std::vector<uint32_t> &rec)
    {
        int got = 0;
        rec.clear();
        std::string line, seq, qual;
        auto getline = [&](std::string &l) -> bool {
            l.clear();
            int c;
            while ((c = gzgetc(fp)) != -1 && c != '\n') l.push_back((char)c);
            if (c == -1 && l.empty()) return false;
            while (!l.empty() && l.back() == '\r') l.pop_back();
            return true;
        };
        while (got < n_rec) {
            int c;
            while ((c = gzgetc(fp)) != -1 && c != '@' && c != '>') {}          // to the next header, as kseq does
            if (c == -1) { eof = true; break; }
            if (c == '>') { fprintf(stderr, "[salt] FASTA input has no base qualities: the reference cannot print SAM for it either\n"); exit(1); }
            if (!getline(line)) { eof = true; break; }
            const std::string head = line;
            seq.clear(); qual.clear();
            bool plus = false;
            for (;;) {
                c = gzgetc(fp);
                if (c == -1) break;
                if (c == '+') { getline(line); plus = true; break; }
                if (c == '>' || c == '@') { gzungetc(c, fp); break; }
                gzungetc(c, fp);
                if (!getline(line)) break;
                for (char ch : line) if (!isspace((unsigned char)ch)) seq.push_back(ch);
            }
            if (!plus) { fprintf(stderr, "[salt] record '%s' has no quality line\n", head.c_str()); exit(1); }
            while (qual.size() < seq.size() && getline(line)) for (char ch : line) if (!isspace((unsigned char)ch)) qual.push_back(ch);
            if (qual.size() != seq.size()) { fprintf(stderr, "[salt] record '%s': %zu bases but %zu qualities\n", head.c_str(), seq.size(), qual.size()); exit(1); }
            rec.push_back((uint32_t)out.size());
            out.push_back('@'); out.insert(out.end(), head.begin(), head.end()); out.push_back('\n');
            out.insert(out.end(), seq.begin(), seq.end()); out.push_back('\n');
            out.push_back('+'); out.push_back('\n');
            out.insert(out.end(), qual.begin(), qual.end()); out.push_back('\n');
            ++got;
        }
        return got;
    }
    bool fill()
    {
        if (eof) return false;
        if (pos > 0) { memmove(buf.data(), buf.data() + pos, have - pos); have -= pos; pos = 0; }
        if (have == buf.size()) buf.resize(buf.size() * 2);
        int n = gzread(fp, buf.data() + have, (unsigned)(buf.size() - have));
        if (n <= 0) { eof = true; return false; }
        have += (size_t)n;
        return true;
    }
    // appends whole records to out until n_rec records or end of file; returns records appended
    int take(std::vector<char> &out, int n_rec, std::vector<uint32_t> &rec)
    {
        if (general) return take_general(out, n_rec, rec);
        int got = 0;
        rec.clear();
        for (;;) {
            size_t scan = pos; int lines = 0; size_t rec_end = pos, rec_start = pos;
            while (got < n_rec) {
                const char *nl = (const char *)memchr(buf.data() + scan, '\n', have - scan);
                if (!nl) break;
                if (lines == 0) rec_start = scan;
                scan = (size_t)(nl - buf.data()) + 1;
                if (++lines == 4) { lines = 0; ++got; rec_end = scan; rec.push_back((uint32_t)(out.size() + (rec_start - pos))); }
            }
            out.insert(out.end(), buf.begin() + (long)pos, buf.begin() + (long)rec_end);
            pos = rec_end;
            if (got >= n_rec) return got;
            if (!fill()) {                                  // end of file: a last record without trailing newline
                if (have > pos) {
                    int nl = 0; for (size_t i = pos; i < have; ++i) nl += buf[i] == '\n';
                    if (nl >= 3) { rec.push_back((uint32_t)out.size()); out.insert(out.end(), buf.begin() + (long)pos, buf.begin() + (long)have); out.push_back('\n'); ++got; }
                    pos = have;
                }
                return got;
            }
        }
    }
};


// Helper threads of one align worker, created once (a batch needs three parallel loops; spawning threads for each costs more
// than the loops' bodies at 100 000 reads per batch).
class Pool {
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv_go, cv_done;
    std::function<void(int)> fn; int n_tasks = 0, next = 0, pending = 0; long gen = 0; bool stop = false;
    void run() {
        long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            while (next < n_tasks) { int t = next++; lk.unlock(); fn(t); lk.lock(); if (--pending == 0) cv_done.notify_all(); }
        }
    }
public:
    const int n;
    explicit Pool(int n_threads) : n(n_threads < 1 ? 1 : n_threads) { for (int i = 1; i < n; ++i) th.emplace_back([this] { run(); }); }
    ~Pool() { { std::unique_lock<std::mutex> lk(mu); stop = true; } cv_go.notify_all(); for (auto &t : th) t.join(); }
    // fn(t) for t in [0, n): the calling thread takes part
    void parallel(const std::function<void(int)> &f) {
        if (n == 1) { f(0); return; }
        std::unique_lock<std::mutex> lk(mu);
        fn = f; n_tasks = n; next = 0; pending = n; ++gen;
        cv_go.notify_all();
        while (next < n_tasks) { int t = next++; lk.unlock(); fn(t); lk.lock(); --pending; }
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

void parse_batch(std::vector<char> &raw, Batch &b, Pool &pool);

void format_batch(const salt_index_t *ix, const salt_sam_opt_t *so, Batch &b, Pool &pool)
{
    const int n = b.n(), n_threads = pool.n;
    std::vector<std::string> &part = b.sam;
    part.assign((size_t)n_threads, std::string());
    pool.parallel([&](int t) {
            std::vector<char> buf(1 << 16);
            int lo = (int)((long)n * t / n_threads), hi = (int)((long)n * (t + 1) / n_threads);
            std::string &out = part[(size_t)t];
            out.reserve((size_t)(hi - lo) * 400);
            for (int i = lo; i < hi; ++i) {
                const int L = (int)(b.offs[i + 1] - b.offs[i]);
                if ((size_t)L * 4 + 4096 > buf.size()) buf.resize((size_t)L * 4 + 4096);
                int w = salt_sam_se(ix, so, b.raw.data() + b.name[i], b.seqs.data() + b.offs[i], L, b.raw.data() + b.qual[i], &b.res[i], buf.data(), buf.size());
                if (w < 0) { fprintf(stderr, "[salt] SAM record too long for read %s\n", b.raw.data() + b.name[i]); exit(1); }
                out.append(buf.data(), (size_t)w);
                out.push_back('\n');
            }
    });
}

// -p: both SAM records of every pair (alnpe_sam, sam.c:331-457); res holds the mates interleaved
void format_batch_pe(const salt_index_t *ix, const salt_sam_opt_t *so, const salt_pe_opt_t *po, Batch &b, Pool &pool)
{
    const int n = b.n(), n_threads = pool.n;
    const Batch &m = *b.mate;
    std::vector<std::string> &part = b.sam;
    part.assign((size_t)n_threads, std::string());
    pool.parallel([&](int t) {
            std::vector<char> buf(1 << 17);
            int lo = (int)((long)n * t / n_threads), hi = (int)((long)n * (t + 1) / n_threads);
            std::string &out = part[(size_t)t];
            out.reserve((size_t)(hi - lo) * 900);
            for (int i = lo; i < hi; ++i) {
                const char *nm[2] = { b.raw.data() + b.name[i], m.raw.data() + m.name[i] };
                const char *ql[2] = { b.raw.data() + b.qual[i], m.raw.data() + m.qual[i] };
                const uint8_t *sq[2] = { b.seqs.data() + b.offs[i], m.seqs.data() + m.offs[i] };
                const int32_t ls[2] = { (int32_t)(b.offs[i + 1] - b.offs[i]), (int32_t)(m.offs[i + 1] - m.offs[i]) };
                if ((size_t)(ls[0] + ls[1]) * 8 + 16384 > buf.size()) buf.resize((size_t)(ls[0] + ls[1]) * 8 + 16384);
                int w = salt_sam_pe(ix, so, po, nm, sq, ls, ql, &b.res[2 * (size_t)i], buf.data(), buf.size());
                if (w < 0) { fprintf(stderr, "[salt] SAM record too long for pair %s\n", nm[0]); exit(1); }
                out.append(buf.data(), (size_t)w);
            }
    });
}

void parse_batch(std::vector<char> &raw, Batch &b, Pool &pool)
{
    const std::vector<uint32_t> &rec = b.rec;                  // record starts, found by the reader
    const int n = (int)rec.size(), n_threads = pool.n;
    b.name.assign((size_t)n, 0u); b.qual.assign((size_t)n, 0u);
    std::vector<uint32_t> len((size_t)n, 0);
    std::vector<std::pair<size_t, size_t>> seq_span((size_t)n);
    auto line_end = [&](size_t p) { const char *nl = (const char *)memchr(raw.data() + p, '\n', raw.size() - p); size_t e = nl ? (size_t)(nl - raw.data()) : raw.size(); return e; };
    pool.parallel([&](int t) {
        for (int i = (int)((long)n * t / n_threads); i < (int)((long)n * (t + 1) / n_threads); ++i) {
            size_t p = rec[(size_t)i], e = line_end(p);
            size_t ne = p + 1;
            while (ne < e && !isspace((unsigned char)raw[ne])) ++ne;
            size_t nl_ = ne - (p + 1);
            if (nl_ > 2 && raw[ne - 2] == '/' && isdigit((unsigned char)raw[ne - 1])) ne -= 2;   // trim_readno (query.c:139-143)
            b.name[(size_t)i] = (uint32_t)(p + 1);
            const size_t name_end = ne;
            size_t s0 = e + 1, s1 = line_end(s0);
            size_t se = s1; while (se > s0 && raw[se - 1] == '\r') --se;
            seq_span[(size_t)i] = { s0, se }; len[(size_t)i] = (uint32_t)(se - s0);
            size_t p2 = s1 + 1, e2 = line_end(p2);           // '+' line
            size_t q0 = e2 + 1, q1 = line_end(q0);
            while (q1 > q0 && raw[q1 - 1] == '\r') --q1;
            b.qual[(size_t)i] = (uint32_t)q0;
            if (raw[p] != '@' || p2 >= raw.size() || raw[p2] != '+' || q1 - q0 != se - s0) {
                fprintf(stderr, "[salt] input is not 4-line FASTQ at record %d of a batch ('%.60s'): multi-line records are only read when "
                                "the head of the file shows them\n", i, raw.data() + p);
                exit(1);
            }
            raw[name_end] = 0;                               // terminate in place (after every read of these lines)
            if (q1 < raw.size()) raw[q1] = 0;
        }
    });
    b.offs.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) b.offs[(size_t)i + 1] = b.offs[(size_t)i] + len[(size_t)i];
    b.seqs.resize(b.offs[(size_t)n]);
    pool.parallel([&](int t) {
        for (int i = (int)((long)n * t / n_threads); i < (int)((long)n * (t + 1) / n_threads); ++i) {
            uint8_t *d = b.seqs.data() + b.offs[(size_t)i];
            const char *sp = raw.data() + seq_span[(size_t)i].first;
            for (uint32_t j = 0; j < len[(size_t)i]; ++j) d[j] = nt4((unsigned char)sp[j]);
        }
    });
}

// The calling thread (and the threads it creates) onto the host NUMA node its GPU is attached to: the chunk a worker preads, the page-locked
// buffers allocated from it and the SAM text it writes then stay on that socket, and with --gpus N every GPU's workers use their own
// socket's cores and memory instead of contending for one.  Nothing happens when the platform does not name a node (or SALT_NO_PIN=1).
void pin_to_device_node(int device)
{
    static const bool off = getenv("SALT_NO_PIN") && atoi(getenv("SALT_NO_PIN"));
    int node = -1;
    if (off || salt_gpu_device_numa_node(device, &node) || node < 0) return;
    char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return;
    cpu_set_t set; CPU_ZERO(&set);
    int a, b, any = 0; char sep;
    while (fscanf(f, "%d", &a) == 1) {                        // "0-63,128-191"
        b = a;
        if (fscanf(f, "%c", &sep) == 1 && sep == '-') { if (fscanf(f, "%d", &b) != 1) break; if (fscanf(f, "%c", &sep) != 1) sep = 0; }
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET(c, &set); any = 1; }
        if (sep != ',') break;
    }
    fclose(f);
    if (any) sched_setaffinity(0, sizeof set, &set);
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
            "           -m, --max_locate    <int>    max loci per strand [1000] (up to 262144)\n"
            "           -p, --pe                     paired end mode (two read files)\n"
            "           -a, --min_tlen      <int>    min template length [250]\n"
            "           -b, --max_tlen      <int>    max template length [550]\n"
            "               --gpus          <int>    GPUs to shard batches over [1]\n"
            "           (-n -e -l -M -O -E -X are accepted and ignored like in the reference)\n\n");
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Single end, plain 4-line FASTQ in a regular file: the text path.  FASTQ text goes to the GPU as it lies in the file, SAM text comes
// back (salt_gpu_align_se_text: parse, align and format are kernels), and the host only moves bytes:
//   every worker claims the next chunk of the file (pread into page-locked memory), cuts it at record boundaries -- a record starts
//   at a line that begins with '@' and whose next-but-one line begins with '+' (a quality line may begin with '@', but then the
//   line two further on is a sequence) --, calls the device, and writes its SAM block when the blocks before it have been written.
//   One writer at a time also for regular files: buffered writes to one file serialize on the inode anyway (one thread alone writes
//   10.6 GB/s into the page cache of the GPU box, eight threads with pwrite at their own offsets 9.5 GB/s between them), and the
//   blocks behind the one being written are read and aligned meanwhile.  Output order = input order, as the reference's puts loop
//   gives it (alnse.c:1433-1439).
// ---------------------------------------------------------------------------------------------
// ---- BGZF input (blocked gzip: every block is a gzip member of at most 64 KiB with its compressed size in a 'BC' extra field -- what
// bgzip and most sequencing pipelines write).  The reference reads any .gz through gzopen (query.c:103-112), one stream, one thread; a
// blocked file can be cut anywhere: the block table (compressed offset, uncompressed offset) is read from the headers and trailers
// without inflating, and the text path treats the UNCOMPRESSED byte range as its file -- a worker inflates the blocks its chunk touches
// into its page-locked buffer, a few at a time on helper threads.  Plain single-member gzip stays on the host pipeline.
struct Bgzf {
    std::vector<uint64_t> coff, uoff;                        // per block: offset in the file, offset in the uncompressed text; one entry past the end
    bool ok = false;
};
static bool bgzf_index(const char *fn, Bgzf &B)
{
    const int fd = open(fn, O_RDONLY);
    if (fd < 0) return false;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); return false; }
    const uint64_t size = (uint64_t)sb.st_size;
    uint64_t at = 0, u = 0;
    bool good = true;
    while (at < size) {
        unsigned char h[18 + 256];
        const ssize_t got = pread(fd, h, sizeof h, (off_t)at);
        if (got < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { good = false; break; }
        const uint32_t xlen = h[10] | (h[11] << 8);
        uint32_t bsize = 0;
        for (uint32_t p = 12; p + 4 <= 12 + xlen && p + 4 <= (uint32_t)got; ) {       // extra subfields: SI1 SI2 SLEN data
            const uint32_t slen = h[p + 2] | (h[p + 3] << 8);
            if (h[p] == 'B' && h[p + 1] == 'C' && slen == 2 && p + 6 <= (uint32_t)got) { bsize = (h[p + 4] | (h[p + 5] << 8)) + 1u; break; }
            p += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || at + bsize > size) { good = false; break; }
        unsigned char t[4];
        if (pread(fd, t, 4, (off_t)(at + bsize - 4)) != 4) { good = false; break; }
        const uint32_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isize > 65536) { good = false; break; }
        B.coff.push_back(at); B.uoff.push_back(u);
        at += bsize; u += isize;
    }
    close(fd);
    if (!good || B.coff.empty()) { B.coff.clear(); B.uoff.clear(); return false; }
    B.coff.push_back(at); B.uoff.push_back(u);
    B.ok = true;
    return true;
}
// inflates block b (bytes cbuf[0 .. csize) of the file) to exactly usize bytes at dst; false on a damaged block
static bool bgzf_inflate(const unsigned char *cbuf, size_t csize, char *dst, size_t usize)
{
    if (csize < 26) return false;
    const uint32_t xlen = cbuf[10] | (cbuf[11] << 8);
    const size_t hdr = 12 + (size_t)xlen;
    if (hdr + 8 > csize) return false;
    z_stream z; memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) return false;
    z.next_in = const_cast<unsigned char *>(cbuf + hdr); z.avail_in = (uInt)(csize - hdr - 8);
    z.next_out = reinterpret_cast<unsigned char *>(dst); z.avail_out = (uInt)usize;
    const int rc = inflate(&z, Z_FINISH);
    bool ok = (rc == Z_STREAM_END || (usize == 0 && rc == Z_BUF_ERROR)) && z.total_out == usize;
    inflateEnd(&z);
    if (ok) {                                                // the member's CRC-32, as gzread checks it
        const unsigned char *t = cbuf + csize - 8;
        const uint32_t want = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        ok = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(dst), (uInt)usize) == want;
    }
    return ok;
}

struct TextRun {
    const Bgzf *bgzf = nullptr;                              // non-null: `file_size` and every offset below are in the uncompressed text
    int fd = -1; uint64_t file_size = 0, chunk = 0;
    std::atomic<uint64_t> next_chunk{ 0 };
    // output sequencing: block k is written once block k - 1 has been
    std::mutex mu; std::condition_variable cv;
    uint64_t written = 0; long reads_done = 0;
    int ready = 0; bool go = false;                         // workers that finished their set-up; the clock starts when all have
    std::atomic<bool> failed{ false };
    // a chunk the device parser refused (SALT_E_INVAL: a multi-line record, a blank line, an empty read -- things kseq.h reads, query.c:103-239):
    // the blocks before it are written, then the host parser takes over at fb_off (a record start: everything before it was strict 4-line FASTQ)
    std::atomic<bool> fallback{ false }; uint64_t fb_off = 0;
};

static const uint64_t TEXT_SLACK = 1u << 20;             // how far past a chunk's end a worker looks for the next record start (longest record it can cut)

// first record start at or after `from` in buf[0..n): a line start whose line begins with '@' and whose next-but-one line begins with '+'.
// `from` itself counts only when the byte before it is a newline (or it is the start of the file).  Returns n when there is none.
static uint64_t next_record_start(const char *buf, uint64_t n, uint64_t from, bool from_is_line_start)
{
    uint64_t p = from;
    if (!from_is_line_start) { const char *nl = (const char *)memchr(buf + p, '\n', n - p); if (!nl) return n; p = (uint64_t)(nl - buf) + 1; }
    while (p < n) {
        const char *n1 = (const char *)memchr(buf + p, '\n', n - p);
        if (!n1) return n;
        const uint64_t l1 = (uint64_t)(n1 - buf) + 1;
        if (buf[p] == '@' && l1 < n) {
            const char *n2 = (const char *)memchr(buf + l1, '\n', n - l1);
            if (!n2) return n;
            const uint64_t l2 = (uint64_t)(n2 - buf) + 1;
            if (l2 < n && buf[l2] == '+') return p;
            if (l2 >= n) return n;
        }
        p = l1;
    }
    return n;
}

// What the text path needs to know before the index is there: chunk size, workers, and -- from the first 64 KB of the file -- how many
// reads a chunk is expected to hold and how much SAM they turn into.  Workspaces are sized by that expectation (+ 30 %), not by the
// worst case of 32-byte records; a chunk that holds more reads (SALT_E_CAPACITY) has its worker re-create the workspace for the worst
// case and call again.
struct TextPlan {
    uint64_t chunk = 0, in_cap = 0, sam_cap = 0;
    int n_workers = 0, wpg = 0;
    uint32_t worst_reads = 0, max_reads = 0, head_read_len = 0;
    // page-locked buffers, allocated by a thread of their own while the index is loaded and attached
    std::vector<char *> in_buf, sam_buf;
    std::thread alloc; bool alloc_ok = true; double alloc_s = 0;
    bool pe = false; uint64_t pairs_per_chunk = 0;          // paired end: chunks are cut by record count (both files the same), about `chunk` bytes per file
    Bgzf bgzf;                                               // single end, blocked gzip input: its block table
    ~TextPlan();
};

static void text_plan(TextPlan &P, const char *fn_reads, int n_gpus, int n_threads, bool pe)
{
    P.pe = pe;
    { const char *e = getenv("SALT_CHUNK_MB"); int mb = e ? atoi(e) : 32; if (mb < 1) mb = 1; if (mb > 1024) mb = 1024; P.chunk = (uint64_t)mb << 20; }
    if (const char *e = getenv("SALT_CHUNK_BYTES")) { long v = atol(e); if (v >= 256) P.chunk = (uint64_t)v; }      // tests: many chunks on a small file
    // workers per GPU: each preads a chunk, calls the device and writes its block in turn.  The write stream is the narrow part (one
    // writer at a time); three to four workers keep it busy, more only lengthen the start-up (measured 2: 25.8, 3: 28.7, 4: 27.9,
    // 6: 24.2, 8: 23.8 Mreads/s on 16 M reads)
    P.wpg = n_threads / n_gpus >= 4 ? 4 : 2;
    if (const char *e = getenv("SALT_TEXT_WORKERS")) { int v = atoi(e); if (v >= 1 && v <= 32) P.wpg = v; }
    P.n_workers = n_gpus * P.wpg;
    P.worst_reads = (uint32_t)((P.chunk + TEXT_SLACK) / 32);
    P.max_reads = P.worst_reads;
    double sam_per_fq_byte = 1.5;                            // SAM bytes a FASTQ byte turns into, for the first sizing of the SAM buffers
    char head[65536];
    ssize_t got = -1;
    if (P.bgzf.ok) {                                         // the head of the TEXT: the first block, inflated
        const int fd = open(fn_reads, O_RDONLY);
        std::vector<unsigned char> cb((size_t)(P.bgzf.coff[1] - P.bgzf.coff[0]));
        const size_t us = (size_t)(P.bgzf.uoff[1] - P.bgzf.uoff[0]);
        if (fd >= 0 && pread(fd, cb.data(), cb.size(), 0) == (ssize_t)cb.size() && us <= sizeof head && bgzf_inflate(cb.data(), cb.size(), head, us)) got = (ssize_t)us;
        if (fd >= 0) close(fd);
    } else { const int fd = open(fn_reads, O_RDONLY); if (fd >= 0) { got = pread(fd, head, sizeof head, 0); close(fd); } }
    uint64_t lines = 0, last_rec_end = 0, line_start = 0, name_bytes = 0;
    for (ssize_t i = 0; i < got; ++i)
        if (head[i] == '\n') {
            const uint32_t len = (uint32_t)((uint64_t)i - line_start);
            if ((lines & 3) == 0) name_bytes += len;
            if ((lines & 3) == 1 && len > P.head_read_len) P.head_read_len = len;
            line_start = (uint64_t)i + 1;
            if ((++lines & 3) == 0) last_rec_end = (uint64_t)i + 1;
        }
    if (lines >= 4 && last_rec_end) {
        const double n_rec = (double)(lines / 4), rec_bytes = (double)last_rec_end / n_rec;
        const double est = (double)(P.chunk + TEXT_SLACK) / rec_bytes * 1.3 + 1024.0;
        if (est < (double)P.worst_reads) P.max_reads = (uint32_t)est;
        // a record: name, the fixed fields (~45 bytes), SEQ, QUAL, tags (NM MD XV XA: ~40 + alternative hits)
        sam_per_fq_byte = (name_bytes / n_rec + 2.0 * P.head_read_len + 160.0) / rec_bytes;
    }
    if (P.head_read_len > SALT_MAX_READ_LEN) P.head_read_len = 0;      // the call itself reports it
    P.in_cap = P.chunk + 2 * TEXT_SLACK + 64 + (P.bgzf.ok ? (3u << 16) : 0u);      // blocked input: whole 64 KiB blocks at both ends
    P.sam_cap = (uint64_t)((double)(P.chunk + TEXT_SLACK) * sam_per_fq_byte) + 4096;
    if (pe) {                                                // a chunk = pairs_per_chunk records of EACH file; both blocks share the input buffer
        const double rec_bytes = lines >= 4 && last_rec_end ? (double)last_rec_end / (double)(lines / 4) : 250.0;
        P.pairs_per_chunk = (uint64_t)((double)P.chunk / rec_bytes) + 1;
        P.max_reads = (uint32_t)(2 * P.pairs_per_chunk);
        P.worst_reads = P.max_reads;                         // exact: the record count is what defines a chunk
        P.in_cap = 2 * (uint64_t)((double)P.chunk * 1.5) + 2 * TEXT_SLACK + 64;
        P.sam_cap = 2 * (uint64_t)((double)P.chunk * (sam_per_fq_byte + 0.3)) + 4096;
    }
    P.in_buf.assign((size_t)P.n_workers, nullptr); P.sam_buf.assign((size_t)P.n_workers, nullptr);
    P.alloc = std::thread([&P]() {
        const double t = now();
        for (int w = 0; w < P.n_workers && P.alloc_ok; ++w) {
            if (w % P.wpg == 0) pin_to_device_node(w / P.wpg);    // worker w's buffers from its GPU's node
            if (salt_gpu_host_alloc(P.in_cap, (void **)&P.in_buf[(size_t)w]) || salt_gpu_host_alloc(P.sam_cap, (void **)&P.sam_buf[(size_t)w])) P.alloc_ok = false;
        }
        P.alloc_s = now() - t;
    });
}

TextPlan::~TextPlan()
{
    if (alloc.joinable()) alloc.join();
    for (char *b : in_buf) if (b) salt_gpu_host_free(b);
    for (char *b : sam_buf) if (b) salt_gpu_host_free(b);
}

// returns 0 = done, 1 = failed, 2 = *resume_off is where the host pipeline has to take over (everything before it is written)
static int run_se_text(const char *fn_reads, salt_index_t *ix, const std::vector<salt_gpu_index_t *> &gix, int n_gpus, TextPlan &P,
                       const salt_aln_opt_t &ao, const salt_sam_opt_t &so, double t0, uint64_t *resume_off)
{
    TextRun R;
    R.fd = open(fn_reads, O_RDONLY);
    if (R.fd < 0) { fprintf(stderr, "[query_open]: file %s open fail!\n", fn_reads); return 1; }
    struct stat sb;
    if (fstat(R.fd, &sb) != 0) { fprintf(stderr, "[salt] cannot stat %s\n", fn_reads); return 1; }
    R.file_size = (uint64_t)sb.st_size;
    if (P.bgzf.ok) { R.bgzf = &P.bgzf; R.file_size = P.bgzf.uoff.back(); }
    R.chunk = P.chunk;
    fflush(stdout);
    // contig table for RNAME / POS on the device
    {
        const int n = salt_index_n_seqs(ix);
        std::vector<int64_t> off((size_t)n); std::vector<const char *> nm((size_t)n);
        for (int i = 0; i < n; ++i) salt_index_seq(ix, i, &off[(size_t)i], nullptr, &nm[(size_t)i]);
        for (int g = 0; g < n_gpus; ++g)
            if (salt_gpu_index_set_contigs(gix[(size_t)g], n, off.data(), nm.data())) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    }
    if (P.alloc.joinable()) P.alloc.join();
    if (!P.alloc_ok) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    const int n_workers = P.n_workers, n_workers_per_gpu = P.wpg;
    const uint32_t worst_reads = P.worst_reads, max_reads = P.max_reads, head_read_len = P.head_read_len;
    const salt_text_opt_t to = { so.print_xa_cigar, so.print_nm_md, so.rg_id };
    const uint64_t n_chunks = (R.file_size + R.chunk - 1) / R.chunk;
    // blocked gzip input: every worker inflates with a few helper threads (a block inflates at ~0.3 GB/s on one core; a worker's chunk must
    // not take longer to inflate than the device takes for the chunks of the other workers)
    int inflate_helpers = 0;
    if (R.bgzf) { inflate_helpers = 5; if (const char *e = getenv("SALT_INFLATE_HELPERS")) { const int v = atoi(e); if (v >= 0 && v <= 64) inflate_helpers = v; } }
    std::vector<std::thread> workers;
    std::atomic<double> t_read{ 0 }, t_gpu{ 0 }, t_write{ 0 }, t_last{ t0 };     // t_last: when the last SAM byte so far was written
    auto set_failed = [&]() { { std::lock_guard<std::mutex> lk(R.mu); R.failed = true; } R.cv.notify_all(); };
    for (int wk = 0; wk < n_workers; ++wk)
        workers.emplace_back([&, wk]() {
            salt_gpu_ws_t *ws = nullptr; char *buf = P.in_buf[(size_t)wk];
            pin_to_device_node(wk / n_workers_per_gpu);
            std::vector<unsigned char> cbuf;                              // blocked gzip input: the compressed bytes of a chunk's blocks
            const bool trace = getenv("SALT_TEXT_TRACE") != nullptr;      // per-worker timeline on stderr
            const double tw_start = now(); int n_calls = 0; double t_first = 0, t_rest = 0;
            uint32_t ws_reads = max_reads;
            if (salt_gpu_ws_create(gix[(size_t)(wk / n_workers_per_gpu)], ws_reads, (uint64_t)ws_reads * 160, &ws) ||
                (head_read_len && salt_gpu_ws_reserve_text(ws, &ao, R.chunk + TEXT_SLACK, (uint32_t)(ws_reads / 1.3), head_read_len, P.sam_cap - 64, P.sam_buf[(size_t)wk], P.sam_cap))) {
                fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); set_failed(); return;
            }
            const double t_setup = now() - tw_start;
            for (;;) {
                const uint64_t k = R.next_chunk.fetch_add(1);
                if (k >= n_chunks || R.failed || R.fallback) break;
                // bytes [lo - 1, hi + slack) of the file: one byte of context in front (is `lo` a line start?), slack behind (where does the last record end?)
                const uint64_t lo = k * R.chunk, hi = std::min(R.file_size, lo + R.chunk);
                const uint64_t rd_lo = lo ? lo - 1 : 0, rd_hi = std::min(R.file_size, hi + TEXT_SLACK);
                double tr0 = now();
                uint64_t got = 0;
                char *const buf0 = buf;                               // (blocked input inflates whole blocks: the chunk's bytes then start inside the buffer)
                if (R.bgzf) {
                    // the blocks that hold text bytes [rd_lo, rd_hi): read as one piece, inflated side by side by this worker and its helpers
                    const Bgzf &B = *R.bgzf;
                    const size_t b0 = (size_t)(std::upper_bound(B.uoff.begin(), B.uoff.end(), rd_lo) - B.uoff.begin()) - 1;
                    size_t b1 = (size_t)(std::lower_bound(B.uoff.begin(), B.uoff.end(), rd_hi) - B.uoff.begin());
                    if (b1 >= B.uoff.size()) b1 = B.uoff.size() - 1;
                    const uint64_t c0 = B.coff[b0], c1 = B.coff[b1], u0 = B.uoff[b0];
                    bool ok = B.uoff[b1] - u0 <= P.in_cap - 64;
                    if (ok) {
                        cbuf.resize((size_t)(c1 - c0));
                        uint64_t cg = 0;
                        while (cg < c1 - c0) { const ssize_t r = pread(R.fd, cbuf.data() + cg, c1 - c0 - cg, (off_t)(c0 + cg)); if (r <= 0) break; cg += (uint64_t)r; }
                        ok = cg == c1 - c0;
                    }
                    if (ok) {
                        std::atomic<size_t> nb{ b0 }; std::atomic<bool> bad{ false };
                        auto work = [&]() {
                            for (;;) {
                                const size_t b = nb.fetch_add(1);
                                if (b >= b1 || bad) break;
                                if (!bgzf_inflate(cbuf.data() + (B.coff[b] - c0), (size_t)(B.coff[b + 1] - B.coff[b]), buf0 + (B.uoff[b] - u0), (size_t)(B.uoff[b + 1] - B.uoff[b]))) bad = true;
                            }
                        };
                        std::vector<std::thread> helpers;
                        for (int h = 0; h < inflate_helpers; ++h) helpers.emplace_back(work);
                        work();
                        for (auto &h : helpers) h.join();
                        ok = !bad;
                    }
                    if (!ok) { fprintf(stderr, "[salt] %s: damaged or oversized gzip block near text offset %llu\n", fn_reads, (unsigned long long)rd_lo); set_failed(); break; }
                    buf = buf0 + (rd_lo - u0);
                    got = rd_hi - rd_lo;
                } else {
                    while (got < rd_hi - rd_lo) {
                        ssize_t r = pread(R.fd, buf + got, rd_hi - rd_lo - got, (off_t)(rd_lo + got));
                        if (r <= 0) break;
                        got += (uint64_t)r;
                    }
                    if (got != rd_hi - rd_lo) { fprintf(stderr, "[salt] short read on %s\n", fn_reads); set_failed(); break; }
                }
                t_read = t_read + (now() - tr0);
                uint64_t n = got;
                if (rd_hi == R.file_size && n && buf[n - 1] != '\n') buf[n++] = '\n';          // a last record without its newline
                const uint64_t b0 = lo - rd_lo;                                                  // offset of file byte `lo` in buf
                const uint64_t beg = next_record_start(buf, n, b0, lo == 0 || buf[b0 - 1] == '\n');
                uint64_t end = n;
                if (hi < R.file_size) end = next_record_start(buf, n, hi - rd_lo, buf[hi - rd_lo - 1] == '\n');
                if (hi < R.file_size && end == n) { fprintf(stderr, "[salt] a FASTQ record longer than %llu bytes near offset %llu\n", (unsigned long long)TEXT_SLACK, (unsigned long long)hi); set_failed(); break; }
                const uint64_t beg2 = std::min(beg, end);
                const char *sam = nullptr; uint64_t sam_bytes = 0; uint32_t n_reads = 0;
                double tg0 = now();
                int grc = end > beg2 ? salt_gpu_align_se_text(ws, &ao, &to, buf + beg2, end - beg2, &sam, &sam_bytes, &n_reads) : 0;
                if (grc == SALT_E_CAPACITY && ws_reads < worst_reads) {                         // shorter records than the file's head promised
                    if (trace) fprintf(stderr, "[salt] worker %d: chunk %llu holds more than %u reads, workspace re-created for %u\n", wk, (unsigned long long)k, ws_reads, worst_reads);
                    salt_gpu_ws_destroy(ws); ws = nullptr; ws_reads = worst_reads;
                    grc = salt_gpu_ws_create(gix[(size_t)(wk / n_workers_per_gpu)], ws_reads, (uint64_t)ws_reads * 160, &ws);
                    if (!grc) grc = salt_gpu_align_se_text(ws, &ao, &to, buf + beg2, end - beg2, &sam, &sam_bytes, &n_reads);
                }
                if (grc == SALT_E_INVAL) {
                    // not strict 4-line FASTQ in this chunk: when the blocks before it are out, the host parser continues from its first record
                    const std::string why = salt_gpu_last_error();
                    std::unique_lock<std::mutex> lk(R.mu);
                    R.cv.wait(lk, [&] { return R.failed || R.fallback || R.written == k; });
                    if (!R.failed && !R.fallback) {
                        R.fb_off = rd_lo + beg2; R.fallback = true;
                        fprintf(stderr, "[salt] %s: the host parser takes over at byte %llu of %s\n", why.c_str(), (unsigned long long)R.fb_off, fn_reads);
                    }
                    lk.unlock(); R.cv.notify_all();
                    break;
                }
                if (grc) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); set_failed(); break; }
                t_gpu = t_gpu + (now() - tg0);
                if (n_calls++ == 0) t_first = now() - tg0; else t_rest += now() - tg0;
                // block k is written when block k - 1 has been
                {
                    std::unique_lock<std::mutex> lk(R.mu);
                    R.cv.wait(lk, [&] { return R.failed || R.fallback || R.written == k; });
                    if (R.failed || R.written != k) break;
                }
                double tw0 = now();
                bool ok = true;
                for (uint64_t w = 0; w < sam_bytes && ok; ) {
                    ssize_t r = write(1, sam + w, sam_bytes - w);
                    if (r <= 0) ok = false; else w += (uint64_t)r;
                }
                t_write = t_write + (now() - tw0);
                { const double tn = now(); double cur = t_last.load(); while (tn > cur && !t_last.compare_exchange_weak(cur, tn)) {} }
                if (!ok) { fprintf(stderr, "[salt] write error on the SAM stream\n"); set_failed(); break; }
                {
                    std::lock_guard<std::mutex> lk(R.mu);
                    R.written = k + 1; R.reads_done += n_reads;
                    fprintf(stderr, "%ld reads have been aligned!\n", R.reads_done);
                }
                R.cv.notify_all();
                buf = buf0;
            }
            if (trace) fprintf(stderr, "[salt] worker %d: started %.3f s after the clock, setup %.3f s, first device call %.3f s, %d later calls %.4f s each, done at %.3f s\n", wk,
                               tw_start - t0, t_setup, t_first, n_calls - 1, n_calls > 1 ? t_rest / (n_calls - 1) : 0.0, now() - t0);
            salt_gpu_ws_destroy(ws);
        });
    for (auto &w : workers) w.join();
    close(R.fd);
    if (!R.failed && R.fallback) { *resume_off = R.fb_off; return 2; }
    const double dt = t_last.load() - t0;                   // first chunk claimed .. last SAM byte written (releasing the workspaces is not alignment time)
    fprintf(stderr, "[alnse_core]: total %lf sec escaped\n", dt);
    fprintf(stderr, "[salt] text path: %d worker(s), chunk %llu MiB, blocks written in turn; seconds summed over workers: read %.3f device call %.3f write %.3f "
                    "(page-locked buffers: %.3f s, while the index was loading)\n", n_workers, (unsigned long long)(R.chunk >> 20), t_read.load(), t_gpu.load(), t_write.load(), P.alloc_s);
    fprintf(stderr, "[salt] %ld reads, %.3f Mreads/s end to end (FASTQ -> SAM, %d GPU(s))\n", R.reads_done, dt > 0 ? R.reads_done / dt / 1e6 : 0.0, n_gpus);
    return R.failed ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// Paired end, two plain 4-line FASTQ files: the same path with chunks cut by RECORD COUNT.  One scanner thread per file counts newlines
// (8 bytes per step) and publishes the byte offset of every pairs_per_chunk-th record; a worker takes chunk k of both files as soon
// as both offsets are there, calls salt_gpu_align_pe_text and writes its block in turn.
// ---------------------------------------------------------------------------------------------
struct PeScan {
    std::mutex mu; std::condition_variable cv;
    std::vector<uint64_t> off[2];                           // off[f][k] = start of chunk k in file f; the last entry of a finished file = its size
    bool done[2] = { false, false }; uint64_t records[2] = { 0, 0 };
    bool bad = false;
};

static inline uint64_t count_nl8(uint64_t w)               // newlines among the 8 bytes of w
{
    const uint64_t x = w ^ 0x0A0A0A0A0A0A0A0Aull, m = 0x7F7F7F7F7F7F7F7Full;
    const uint64_t y = ~(((x & m) + m) | x | m);            // bit 7 of every byte that is zero in x
    return (uint64_t)__builtin_popcountll(y);
}

// The chunk boundaries of one file of a pair: chunk k starts where line 4 k pairs_per_chunk starts.  Counting lines is a pass over the
// whole file, and one thread counts ~4 GB/s -- 30 M mates/s of 150-base pairs, less than one GPU aligns.  So the file goes in segments of
// 32 MiB to PE_SCAN_THREADS threads: each counts its segment's newlines, learns how many lines lie in front of it once the segments before
// it are counted (a running sum kept under the scan's mutex), finds the chunk starts that fall inside its segment in the bytes it still
// holds, and publishes them after the segment before it has published its own.
static const int PE_SCAN_THREADS = 4;
static void pe_scan_file(const char *fn, int f, uint64_t pairs_per_chunk, PeScan &S, std::atomic<bool> &failed)
{
    const int fd = open(fn, O_RDONLY);
    const uint64_t lines_per_chunk = 4 * pairs_per_chunk;
    if (fd < 0) { { std::lock_guard<std::mutex> lk(S.mu); S.bad = true; S.done[f] = true; } S.cv.notify_all(); return; }
    struct stat st;
    const uint64_t size = fstat(fd, &st) == 0 ? (uint64_t)st.st_size : 0;
    { std::lock_guard<std::mutex> lk(S.mu); S.off[f].push_back(0); }
    S.cv.notify_all();
    uint64_t SEG = 32ull << 20;
    if (const char *e = getenv("SALT_PE_SCAN_SEG_BYTES")) { const long long v = atoll(e); if (v >= 64) SEG = (uint64_t)v; }      // tests: many segments in a small file
    const uint64_t n_seg = (size + SEG - 1) / SEG;
    std::mutex mu; std::condition_variable cv;               // the order among this file's segments
    std::vector<int64_t> cnt((size_t)n_seg, -1);             // newlines of a segment, once counted
    std::vector<uint64_t> before((size_t)n_seg + 1, 0);      // lines in front of a segment, known for segments < known
    uint64_t known = 0, published = 0;                       // segments whose `before` is known / whose boundaries are out
    std::atomic<uint64_t> next_seg{ 0 };
    std::atomic<bool> short_read{ false };
    char last_byte = '\n';
    auto work = [&]() {
        std::vector<char> buf((size_t)SEG);
        for (;;) {
            const uint64_t sg = next_seg.fetch_add(1);
            if (sg >= n_seg || failed) break;
            const uint64_t lo = sg * SEG, want = std::min<uint64_t>(SEG, size - lo);
            uint64_t got = 0;
            while (got < want) { const ssize_t r = pread(fd, buf.data() + got, (size_t)(want - got), (off_t)(lo + got)); if (r <= 0) break; got += (uint64_t)r; }
            if (got != want) short_read = true;               // (the file shrank under us: the workers' own reads will fail on it)
            uint64_t c = 0; size_t j = 0;
            for (; j + 8 <= (size_t)got; j += 8) { uint64_t w; memcpy(&w, buf.data() + j, 8); c += count_nl8(w); }
            for (; j < (size_t)got; ++j) c += buf[j] == '\n';
            uint64_t l0;
            {
                std::unique_lock<std::mutex> lk(mu);
                cnt[(size_t)sg] = (int64_t)c;
                while (known < n_seg && cnt[(size_t)known] >= 0) { before[(size_t)known + 1] = before[(size_t)known] + (uint64_t)cnt[(size_t)known]; ++known; }
                cv.notify_all();
                while (known <= sg && !failed) cv.wait_for(lk, std::chrono::milliseconds(50));      // (a failed run notifies nobody here)
                if (known <= sg) break;
                l0 = before[(size_t)sg];
                if (sg + 1 == n_seg && got) last_byte = buf[(size_t)got - 1];
            }
            // chunk starts inside this segment: the byte behind the newline that completes line m * lines_per_chunk, l0 < m * lpc <= l0 + c
            std::vector<uint64_t> mine;
            uint64_t nextb = (l0 / lines_per_chunk + 1) * lines_per_chunk, lines = l0;
            for (size_t i = 0; i < (size_t)got && nextb <= l0 + c; ) {
                const size_t piece = std::min<size_t>(4096, (size_t)got - i);
                uint64_t pc = 0; size_t q = 0;
                for (; q + 8 <= piece; q += 8) { uint64_t w; memcpy(&w, buf.data() + i + q, 8); pc += count_nl8(w); }
                for (; q < piece; ++q) pc += buf[i + q] == '\n';
                if (lines + pc < nextb) { lines += pc; i += piece; continue; }
                for (q = 0; q < piece; ++q)
                    if (buf[i + q] == '\n' && ++lines == nextb) { mine.push_back(lo + i + q + 1); nextb += lines_per_chunk; }
                i += piece;
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                while (published != sg && !failed) cv.wait_for(lk, std::chrono::milliseconds(50));
                if (published != sg) break;
                if (!mine.empty()) {
                    { std::lock_guard<std::mutex> lk2(S.mu); for (uint64_t o : mine) if (o < size) S.off[f].push_back(o); }      // (a start at the very end is the file's end, added below)
                    S.cv.notify_all();
                }
                published = sg + 1;
                cv.notify_all();
            }
        }
    };
    std::vector<std::thread> th;
    const int nt = (int)std::min<uint64_t>((uint64_t)PE_SCAN_THREADS, std::max<uint64_t>(n_seg, 1));
    for (int t = 0; t < nt; ++t) th.emplace_back(work);
    for (auto &t : th) t.join();
    close(fd);
    uint64_t lines = known == n_seg ? before[(size_t)n_seg] : 0;
    if (size && last_byte != '\n') ++lines;                   // a last line without its newline
    {
        std::lock_guard<std::mutex> lk(S.mu);
        if (S.off[f].back() != size) S.off[f].push_back(size);
        S.records[f] = lines / 4; S.done[f] = true;
        if (lines % 4 || short_read || (failed && known != n_seg)) S.bad = true;
    }
    S.cv.notify_all();
}

// returns 0 = done, 1 = failed, 2 = the host pipeline has to take over at resume_off[0 / 1] of the two files (everything before is written):
// a chunk the device parser refused, a file whose line count is not a multiple of four (a trailing blank line is fine for kseq.h), or files
// with different numbers of chunks -- the host parser reads what the reference reads and reports what it cannot
static int run_pe_text(const char *fn1, const char *fn2, salt_index_t *ix, const std::vector<salt_gpu_index_t *> &gix, int n_gpus, TextPlan &P,
                       const salt_aln_opt_t &ao, const salt_sam_opt_t &so, const salt_pe_opt_t &po, double t0, uint64_t resume_off[2])
{
    const int fd[2] = { open(fn1, O_RDONLY), open(fn2, O_RDONLY) };
    if (fd[0] < 0 || fd[1] < 0) { fprintf(stderr, "[query_open]: file %s open fail!\n", fd[0] < 0 ? fn1 : fn2); return 1; }
    fflush(stdout);
    {
        const int n = salt_index_n_seqs(ix);
        std::vector<int64_t> off((size_t)n); std::vector<const char *> nm((size_t)n);
        for (int i = 0; i < n; ++i) salt_index_seq(ix, i, &off[(size_t)i], nullptr, &nm[(size_t)i]);
        for (int g = 0; g < n_gpus; ++g)
            if (salt_gpu_index_set_contigs(gix[(size_t)g], n, off.data(), nm.data())) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    }
    if (P.alloc.joinable()) P.alloc.join();
    if (!P.alloc_ok) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    const salt_text_opt_t to = { so.print_xa_cigar, so.print_nm_md, so.rg_id };
    PeScan S;
    std::atomic<bool> failed{ false }, fallback{ false };
    std::mutex wmu; std::condition_variable wcv; uint64_t written = 0, fb_chunk = 0; long pairs_done = 0;
    std::atomic<uint64_t> next_chunk{ 0 };
    std::atomic<double> t_read{ 0 }, t_gpu{ 0 }, t_write{ 0 }, t_last{ t0 };
    auto set_failed = [&]() { failed = true; { std::lock_guard<std::mutex> lk(wmu); } wcv.notify_all(); { std::lock_guard<std::mutex> lk(S.mu); } S.cv.notify_all(); };
    // chunk k cannot go through the device parser: once the blocks before it are out, everyone stops and the host pipeline continues there
    auto fall_back = [&](uint64_t k, const std::string &why) {
        {
            std::unique_lock<std::mutex> lk(wmu);
            wcv.wait(lk, [&] { return failed.load() || fallback.load() || written == k; });
            if (!failed && !fallback) { fb_chunk = k; fallback = true; fprintf(stderr, "[salt] %s: the host parser takes over at pair chunk %llu\n", why.c_str(), (unsigned long long)k); }
        }
        wcv.notify_all(); { std::lock_guard<std::mutex> lk(S.mu); } S.cv.notify_all();
    };
    std::thread scan1(pe_scan_file, fn1, 0, P.pairs_per_chunk, std::ref(S), std::ref(failed)), scan2(pe_scan_file, fn2, 1, P.pairs_per_chunk, std::ref(S), std::ref(failed));
    std::vector<std::thread> workers;
    for (int wk = 0; wk < P.n_workers; ++wk)
        workers.emplace_back([&, wk]() {
            salt_gpu_ws_t *ws = nullptr; char *buf = P.in_buf[(size_t)wk];
            pin_to_device_node(wk / P.wpg);
            if (salt_gpu_ws_create(gix[(size_t)(wk / P.wpg)], P.max_reads + 64, (uint64_t)(P.max_reads + 64) * 160, &ws) ||
                (P.head_read_len && salt_gpu_ws_reserve_text(ws, &ao, P.in_cap, P.max_reads, P.head_read_len, P.sam_cap - 64, P.sam_buf[(size_t)wk], P.sam_cap))) {
                fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); set_failed(); return;
            }
            for (;;) {
                const uint64_t k = next_chunk.fetch_add(1);
                uint64_t lo[2], hi[2]; bool end = false, odd = false;
                {   // chunk k of both files: its start and end offsets (or the news that there is no chunk k)
                    std::unique_lock<std::mutex> lk(S.mu);
                    S.cv.wait(lk, [&] { return failed.load() || fallback.load() || ((S.off[0].size() > k + 1 || S.done[0]) && (S.off[1].size() > k + 1 || S.done[1])); });
                    if (failed || fallback) break;
                    if (S.bad) odd = true;
                    else if (S.off[0].size() <= k + 1 || S.off[1].size() <= k + 1) {
                        // one file has no chunk k: fine if neither has (both finished with the same number of chunks), else the files differ
                        if ((S.off[0].size() > k + 1) != (S.off[1].size() > k + 1)) { S.bad = true; odd = true; }
                        else end = true;
                    } else for (int f = 0; f < 2; ++f) { lo[f] = S.off[f][k]; hi[f] = S.off[f][k + 1]; }
                }
                if (end) break;
                if (odd) { fall_back(k, "the read files are not two equally long runs of 4-line records"); break; }
                const uint64_t n1 = hi[0] - lo[0], n2 = hi[1] - lo[1], b2 = (n1 + 64) & ~63ull;
                if (b2 + n2 + 2 > P.in_cap) { fprintf(stderr, "[salt] a chunk of %llu pairs is larger than its buffer (records much longer than the file's first ones)\n", (unsigned long long)P.pairs_per_chunk); set_failed(); break; }
                double tr0 = now();
                bool ok = true;
                for (int f = 0; f < 2 && ok; ++f) {
                    char *dst = buf + (f ? b2 : 0); const uint64_t want = f ? n2 : n1; uint64_t got = 0;
                    while (got < want) { const ssize_t r = pread(fd[f], dst + got, want - got, (off_t)(lo[f] + got)); if (r <= 0) break; got += (uint64_t)r; }
                    ok = got == want;
                }
                if (!ok) { fprintf(stderr, "[salt] short read on the FASTQ files\n"); set_failed(); break; }
                uint64_t m1 = n1, m2 = n2;
                if (m1 && buf[m1 - 1] != '\n') buf[m1++] = '\n';                              // a last record without its newline
                if (m2 && buf[b2 + m2 - 1] != '\n') buf[b2 + m2++] = '\n';
                t_read = t_read + (now() - tr0);
                const char *sam = nullptr; uint64_t sam_bytes = 0; uint32_t n_pairs = 0;
                double tg0 = now();
                const int grc = salt_gpu_align_pe_text(ws, &ao, &po, &to, buf, m1, buf + b2, m2, &sam, &sam_bytes, &n_pairs);
                if (grc == SALT_E_INVAL) { fall_back(k, salt_gpu_last_error()); break; }
                if (grc) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); set_failed(); break; }
                t_gpu = t_gpu + (now() - tg0);
                {
                    std::unique_lock<std::mutex> lk(wmu);
                    wcv.wait(lk, [&] { return failed.load() || fallback.load() || written == k; });
                    if (failed || written != k) break;
                }
                double tw0 = now();
                for (uint64_t w = 0; w < sam_bytes && ok; ) { const ssize_t r = write(1, sam + w, sam_bytes - w); if (r <= 0) ok = false; else w += (uint64_t)r; }
                t_write = t_write + (now() - tw0);
                { const double tn = now(); double cur = t_last.load(); while (tn > cur && !t_last.compare_exchange_weak(cur, tn)) {} }
                if (!ok) { fprintf(stderr, "[salt] write error on the SAM stream\n"); set_failed(); break; }
                { std::lock_guard<std::mutex> lk(wmu); written = k + 1; pairs_done += n_pairs; fprintf(stderr, "%ld reads have been aligned!\n", 2 * pairs_done); }
                wcv.notify_all();
            }
            salt_gpu_ws_destroy(ws);
        });
    for (auto &w : workers) w.join();
    failed = failed.load();
    { std::lock_guard<std::mutex> lk(S.mu); } S.cv.notify_all();
    scan1.join(); scan2.join();
    close(fd[0]); close(fd[1]);
    if (!failed && fallback) {
        for (int f = 0; f < 2; ++f) resume_off[f] = S.off[f].size() > fb_chunk ? S.off[f][fb_chunk] : (S.off[f].empty() ? 0 : S.off[f].back());
        return 2;
    }
    if (!failed && (S.bad || S.records[0] != S.records[1])) {
        fprintf(stderr, "[salt] the two read files hold different numbers of reads (%llu / %llu) or broken records\n", (unsigned long long)S.records[0], (unsigned long long)S.records[1]);
        return 1;
    }
    const double dt = t_last.load() - t0;
    fprintf(stderr, "[alnpe_core]: total %lf sec escaped\n", dt);
    fprintf(stderr, "[salt] text path (paired end): %d worker(s), %llu pairs per chunk, blocks written in turn; seconds summed over workers: read %.3f device call %.3f write %.3f "
                    "(page-locked buffers: %.3f s, while the index was loading)\n", P.n_workers, (unsigned long long)P.pairs_per_chunk, t_read.load(), t_gpu.load(), t_write.load(), P.alloc_s);
    fprintf(stderr, "[salt] %ld pairs, %.3f M mates/s end to end (FASTQ -> SAM, %d GPU(s))\n", pairs_done, dt > 0 ? 2.0 * pairs_done / dt / 1e6 : 0.0, n_gpus);
    return failed ? 1 : 0;
}

} // namespace

int main(int argc, char **argv)
{
    int n_threads = 1, n_gpus = 1, overlap = -1, pe = 0;
    salt_aln_opt_t ao; memset(&ao, 0, sizeof ao);
    ao.max_seed = 50; ao.max_locate = 1000; ao.max_hits = 5;             // aln.c:46-47, aln.h:133
    salt_sam_opt_t so; memset(&so, 0, sizeof so);
    salt_pe_opt_t po = { 250, 550 };                                       // aln.c:43-44
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
        case 'a': po.min_tlen = (uint32_t)atoi(optarg); break;
        case 'b': po.max_tlen = (uint32_t)atoi(optarg); break;
        case 1000: n_gpus = atoi(optarg); break;
        case 'h': return usage();
        case '?': fprintf(stderr, "[ERROR]: no arg %c\n", optopt); return 1;
        default: break;
        }
    }
    if (optind + 2 + pe > argc) { fprintf(stderr, "[opt_parse]: index prefix and read file can't be omited!\n"); return 1; }
    if (n_threads < 1) n_threads = 1;
    if (n_gpus < 1) n_gpus = 1;
    const char *prefix = argv[optind], *fn_reads = argv[optind + 1], *fn_mates = pe ? argv[optind + 2] : nullptr;

    // Single end + a plain (not gzipped) strict 4-line FASTQ in a regular file: the text path -- parse, align and format on the device
    // (run_se_text).  Everything else (paired end, gzip, pipes, multi-line records) goes through the host pipeline below.
    // SALT_HOST_PIPELINE=1 forces the latter.  Decided before the index is loaded: the text path's page-locked buffers are allocated
    // by a thread of their own meanwhile.
    TextPlan plan; bool text_path = false;
    if (!(getenv("SALT_HOST_PIPELINE") && atoi(getenv("SALT_HOST_PIPELINE")))) {
        auto plain4 = [](const char *fn) {
            struct stat sb; unsigned char magic[2] = { 0, 0 };
            bool plain = stat(fn, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0;
            if (plain) { FILE *f = fopen(fn, "rb"); plain = f && fread(magic, 1, 2, f) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b); if (f) fclose(f); }
            return plain && sniff_four_line(fn);
        };
        text_path = plain4(fn_reads) && (!pe || plain4(fn_mates));
        // single end, blocked gzip (BGZF) whose text starts as strict 4-line FASTQ: the text path over the uncompressed byte range
        if (!text_path && !pe && bgzf_index(fn_reads, plan.bgzf) && sniff_four_line(fn_reads)) text_path = true;
        else if (!text_path) plan.bgzf = Bgzf();
        if (text_path) text_plan(plan, fn_reads, n_gpus, n_threads, pe != 0);
    }
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
    if (pe) {                                             // the singleton rescue aligns against the 2-bit genome (alnpe.c:327-393)
        uint64_t l_pac = 0; const uint8_t *pac = salt_index_pac(ix, &l_pac);
        for (int i = 0; i < n_gpus; ++i)
            if (salt_gpu_index_set_pac(gix[(size_t)i], pac, l_pac)) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    }
    auto print_header = [&]() -> bool {                  // aln_samhead (sam.c:56-84)
        std::vector<char> hb(16 << 20);
        int w = salt_sam_header(ix, &so, hb.data(), hb.size());
        if (w < 0) { fprintf(stderr, "[salt] SAM header too large\n"); return false; }
        fwrite(hb.data(), 1, (size_t)w, stdout);
        time_t tt = time(nullptr); struct tm *tmv = localtime(&tt);
        printf("@PG\tID:snpaln\tPN:snpaln\tCL:\"%s\"\tDS:%d-%d-%d\tVN:0.1beta\n", cmd.c_str(), tmv->tm_year + 1900, tmv->tm_mon + 1, tmv->tm_mday);
        return true;
    };
    bool header_out = false; uint64_t resume[2] = { 0, 0 };      // set when the text path hands the rest of the input to the host pipeline
    if (text_path && !pe) {
        fprintf(stderr, "%lf sec escaped.\n", now() - t0);
        if (!print_header()) return 1;
        const int rc = run_se_text(fn_reads, ix, gix, n_gpus, plan, ao, so, now(), &resume[0]);
        if (rc != 2) {
            for (int i = n_gpus - 1; i >= 0; --i) salt_gpu_index_detach(gix[(size_t)i]);
            salt_index_free(ix);
            return rc;
        }
        header_out = true;
    }
    if (pe && po.max_tlen == 0) {
        // N3: -b 0 = infer the insert-size window from the first batch (N_SEQS / 2 pairs), the mates aligned as single-end reads
        // (salt_isize_infer; the reference prints "infer isize func haven't been implemented" here, alnpe.c:586-589)
        gzFile g1 = gzopen(fn_reads, "r"), g2 = gzopen(fn_mates, "r");
        if (!g1 || !g2) { fprintf(stderr, "[query_open]: file %s open fail!\n", g1 ? fn_mates : fn_reads); return 1; }
        RawReader r1(g1, !sniff_four_line(fn_reads)), r2(g2, !sniff_four_line(fn_mates));
        Batch b1, b2; Pool pool(n_threads < 16 ? n_threads : 16);
        const int got = r1.take(b1.raw, N_SEQS / 2, b1.rec);
        if (got == 0 || r2.take(b2.raw, got, b2.rec) != got) { fprintf(stderr, "[salt] the two read files hold different numbers of reads\n"); return 1; }
        parse_batch(b1.raw, b1, pool); parse_batch(b2.raw, b2, pool);
        std::vector<uint8_t> iseq(b1.seqs.size() + b2.seqs.size()); std::vector<uint32_t> ioff(2 * (size_t)got + 1, 0);
        for (int i = 0; i < got; ++i) {
            const uint32_t l0 = b1.offs[(size_t)i + 1] - b1.offs[(size_t)i], l1 = b2.offs[(size_t)i + 1] - b2.offs[(size_t)i];
            memcpy(iseq.data() + ioff[2 * (size_t)i], b1.seqs.data() + b1.offs[(size_t)i], l0); ioff[2 * (size_t)i + 1] = ioff[2 * (size_t)i] + l0;
            memcpy(iseq.data() + ioff[2 * (size_t)i + 1], b2.seqs.data() + b2.offs[(size_t)i], l1); ioff[2 * (size_t)i + 2] = ioff[2 * (size_t)i + 1] + l1;
        }
        gzclose(g1); gzclose(g2);
        salt_gpu_ws_t *w0 = nullptr;
        std::vector<salt_result_t> r((size_t)2 * got);
        if (salt_gpu_ws_create(gix[0], (uint32_t)(2 * got), (uint64_t)iseq.size() + 64, &w0) ||
            salt_gpu_align_se(w0, &ao, (uint32_t)(2 * got), iseq.data(), ioff.data(), r.data())) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
        salt_gpu_ws_destroy(w0);
        uint32_t used = 0;
        if (salt_isize_infer(ix, (uint32_t)got, ioff.data(), r.data(), &po.min_tlen, &po.max_tlen, &used)) {
            fprintf(stderr, "[alnpe_core]: cannot infer the insert size: %u usable pairs in the first batch (25 needed); give -a / -b\n", used);
            return 1;
        }
        fprintf(stderr, "[alnpe_core]: insert size window [%u, %u] inferred from %u pairs\n", po.min_tlen, po.max_tlen, used);
    }
    if (text_path && pe) {
        fprintf(stderr, "%lf sec escaped.\n", now() - t0);
        if (!print_header()) return 1;
        const int rc = run_pe_text(fn_reads, fn_mates, ix, gix, n_gpus, plan, ao, so, po, now(), resume);
        if (rc != 2) {
            for (int i = n_gpus - 1; i >= 0; --i) salt_gpu_index_detach(gix[(size_t)i]);
            salt_index_free(ix);
            return rc;
        }
        header_out = true;
    }
    // workers per GPU: each takes a batch through parse -> device -> format, so several batches overlap on the host
    const int WPG = n_threads / n_gpus >= 32 ? 4 : n_threads / n_gpus >= 12 ? 3 : 2;
    std::vector<salt_gpu_ws_t *> ws((size_t)n_gpus * WPG, nullptr);
    for (int i = 0; i < n_gpus * WPG; ++i)
        if (salt_gpu_ws_create(gix[(size_t)(i / WPG)], N_SEQS, (uint64_t)N_SEQS * SALT_MAX_READ_LEN, &ws[(size_t)i])) { fprintf(stderr, "[salt] %s\n", salt_gpu_last_error()); return 1; }
    fprintf(stderr, "%lf sec escaped.\n", now() - t0);
    t0 = now();

    gzFile fp = gzopen(fn_reads, "r");
    if (!fp) { fprintf(stderr, "[query_open]: file %s open fail!\n", fn_reads); return 1; }
    gzbuffer(fp, 1 << 20);
    gzFile fp2 = nullptr;
    if (pe) {
        fp2 = gzopen(fn_mates, "r");
        if (!fp2) { fprintf(stderr, "[query_open]: file %s open fail!\n", fn_mates); return 1; }
        gzbuffer(fp2, 1 << 20);
    }
    if (header_out) {                                     // the text path wrote everything up to these offsets (plain files: a seek)
        fflush(stdout);
        if (gzseek(fp, (z_off_t)resume[0], SEEK_SET) < 0 || (pe && gzseek(fp2, (z_off_t)resume[1], SEEK_SET) < 0)) { fprintf(stderr, "[salt] cannot seek in the read files\n"); return 1; }
    } else if (!print_header()) return 1;

    // ---- pipeline: reader -> per-GPU workers -> ordered writer ----
    std::mutex mu; std::condition_variable cv;
    std::deque<std::unique_ptr<Batch>> todo;          // read, not yet aligned
    std::deque<std::unique_ptr<Batch>> done;          // aligned + formatted, any order
    bool eof = false; long n_tot = 0; std::atomic<bool> failed{ false };
    const size_t max_inflight = (size_t)n_gpus * WPG * 2 + 1;
    size_t inflight = 0;
    // the flag is stored with `mu` held: a waiter that has evaluated its predicate but not yet blocked cannot miss the notify
    auto set_failed = [&]() { { std::lock_guard<std::mutex> lk(mu); failed = true; } cv.notify_all(); };

    std::atomic<double> t_parse{ 0 }, t_gpu{ 0 }, t_fmt{ 0 };
    double t_write = 0, t_read = 0;
    std::thread reader([&]() {
        // (after a hand-over from the text path the head of the file says nothing about what follows: kseq's general reader)
        RawReader rr(fp, header_out || !sniff_four_line(fn_reads));
        std::unique_ptr<RawReader> rr2(pe ? new RawReader(fp2, header_out || !sniff_four_line(fn_mates)) : nullptr);
        const int per_batch = pe ? N_SEQS / 2 : N_SEQS;    // pairs per batch: N_SEQS mates (query_read_multiPairedSeqs, query.c:252-268)
        long seq_no = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return inflight < max_inflight || failed; }); if (failed) break; }
            auto b = std::make_unique<Batch>();
            b->seq_no = seq_no;
            b->raw.reserve((size_t)N_SEQS * 260);
            double tr0 = now();
            const int got = rr.take(b->raw, per_batch, b->rec);
            if (pe && got) {
                b->mate = std::make_unique<Batch>();
                b->mate->raw.reserve((size_t)per_batch * 260);
                if (rr2->take(b->mate->raw, got, b->mate->rec) != got) { fprintf(stderr, "[salt] the two read files hold different numbers of reads\n"); set_failed(); }
            }
            if (got == 0 || failed) b.reset();
            t_read += now() - tr0;
            std::unique_lock<std::mutex> lk(mu);
            if (!b) { eof = true; cv.notify_all(); break; }
            ++seq_no; ++inflight;
            todo.push_back(std::move(b));
            cv.notify_all();
        }
    });
    std::vector<std::thread> workers;
    const int fmt_threads = n_threads / (n_gpus * WPG) > 0 ? n_threads / (n_gpus * WPG) : 1;
    for (int g = 0; g < n_gpus * WPG; ++g)
        workers.emplace_back([&, g]() {
            pin_to_device_node(g / WPG);                      // before the pool: its threads inherit the node
            Pool pool(fmt_threads);
            for (;;) {
                std::unique_ptr<Batch> b;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !todo.empty() || eof || failed; });
                    if (failed || (todo.empty() && eof)) break;
                    b = std::move(todo.front()); todo.pop_front();
                }
                double tp0 = now();
                parse_batch(b->raw, *b, pool);
                std::vector<uint8_t> iseq; std::vector<uint32_t> ioff;
                if (pe && b->n()) {                              // interleave the mates: pair i = reads 2i, 2i+1
                    Batch &m = *b->mate;
                    parse_batch(m.raw, m, pool);
                    if (m.n() != b->n()) { fprintf(stderr, "[salt] the two read files hold different numbers of reads\n"); set_failed(); break; }
                    const size_t n = (size_t)b->n();
                    ioff.resize(2 * n + 1); ioff[0] = 0;
                    iseq.resize(b->seqs.size() + m.seqs.size());
                    for (size_t i = 0; i < n; ++i) {
                        const uint32_t l0 = b->offs[i + 1] - b->offs[i], l1 = m.offs[i + 1] - m.offs[i];
                        memcpy(iseq.data() + ioff[2 * i], b->seqs.data() + b->offs[i], l0); ioff[2 * i + 1] = ioff[2 * i] + l0;
                        memcpy(iseq.data() + ioff[2 * i + 1], m.seqs.data() + m.offs[i], l1); ioff[2 * i + 2] = ioff[2 * i + 1] + l1;
                    }
                }
                t_parse = t_parse + (now() - tp0);
                if (b->n() == 0) { std::unique_lock<std::mutex> lk(mu); done.push_back(std::move(b)); cv.notify_all(); continue; }
                b->res.resize((size_t)b->n() * (pe ? 2 : 1));
                double tg0 = now();
                int grc = pe ? salt_gpu_align_pe(ws[(size_t)g], &ao, &po, (uint32_t)b->n(), iseq.data(), ioff.data(), b->res.data())
                             : salt_gpu_align_se(ws[(size_t)g], &ao, (uint32_t)b->n(), b->seqs.data(), b->offs.data(), b->res.data());
                t_gpu = t_gpu + (now() - tg0);
                if (grc) {
                    fprintf(stderr, "[salt] %s\n", salt_gpu_last_error());
                    set_failed(); break;
                }
                double tf0 = now();
                if (pe) format_batch_pe(ix, &so, &po, *b, pool); else format_batch(ix, &so, *b, pool);
                t_fmt = t_fmt + (now() - tf0);
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
        double tw0 = now();
        for (const std::string &piece : b->sam) fwrite(piece.data(), 1, piece.size(), stdout);
        t_write += now() - tw0;
        n_tot += b->n() * (pe ? 2 : 1); ++next;
        fprintf(stderr, "%ld reads have been aligned!\n", n_tot);
        { std::unique_lock<std::mutex> lk(mu); --inflight; cv.notify_all(); }
    }
    reader.join();
    for (auto &w : workers) w.join();
    fflush(stdout);
    double dt = now() - t0;
    fprintf(stderr, "[alnse_core]: total %lf sec escaped\n", dt);
    fprintf(stderr, "[salt] host phases (s, summed over workers): read %.3f parse %.3f gpu-call %.3f format %.3f write %.3f\n", t_read, t_parse.load(), t_gpu.load(), t_fmt.load(), t_write);
    fprintf(stderr, "[salt] %ld reads, %.3f Mreads/s end to end (FASTQ -> SAM, %d GPU(s), %d host thread(s))\n", n_tot, dt > 0 ? n_tot / dt / 1e6 : 0.0, n_gpus, n_threads);
    gzclose(fp);
    if (fp2) gzclose(fp2);
    for (int i = 0; i < n_gpus * WPG; ++i) salt_gpu_ws_destroy(ws[(size_t)i]);
    for (int i = n_gpus - 1; i >= 0; --i) salt_gpu_index_detach(gix[(size_t)i]);
    salt_index_free(ix);
    return failed ? 1 : 0;
}
