// salt_amd/host/salt_idx_main.cc -- `salt-idx [-k seedlen] [--gpu [device]] [--no-lp] <ref.fa> <snps.txt> <out.prefix>`
// (Index_src/index1.c:46-185: index_main / index_usage; default seed length 25).
// --gpu sorts the suffixes on an MI355X through libsalt_gpu.so (loaded here, so that the tool still runs on a machine without
// ROCm); without it the host suffix sorter is used.  The files are the same either way.
#include "../../include/salt_host.h"
#include <dlfcn.h>
#include <getopt.h>
#include <unistd.h>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <string>

static int usage()
{
    fprintf(stderr, "Usage: salt-idx [-k seed_length(25)] [--gpu[=device]] [--no-lp] [--all-files] <ref.fa> <snp file> <index prefix>\n");
    return 1;
}

int main(int argc, char **argv)
{
    int k = 25, c, gpu = -1, flags = 0;
    static const struct option lo[] = { { "gpu", 2, 0, 1000 }, { "no-lp", 0, 0, 1001 }, { "all-files", 0, 0, 1002 }, { "help", 0, 0, 'h' }, { 0, 0, 0, 0 } };
    while ((c = getopt_long(argc, argv, "k:h", lo, nullptr)) >= 0) {
        if (c == 'k') k = atoi(optarg);
        else if (c == 1000) gpu = optarg ? atoi(optarg) : 0;
        else if (c == 1001) flags |= SALT_IDX_NO_LP;
        else if (c == 1002) flags |= SALT_IDX_ALL_FILES;          // the files the reference indexer writes and `salt` never reads, too
        else return usage();
    }
    if (argc - optind != 3) return usage();
    salt_idx_backend_t be, *bep = nullptr;
    if (gpu >= 0) {
        char exe[PATH_MAX]; ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
        std::string dir = n > 0 ? std::string(exe, (size_t)n) : std::string(argv[0]);
        dir = dir.substr(0, dir.find_last_of('/'));
        const std::string lib = dir + "/../lib/libsalt_gpu.so";
        void *h = dlopen(lib.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!h) { fprintf(stderr, "[salt-idx] --gpu: %s\n", dlerror()); return 1; }
        be.device = gpu;
        be.build_c = reinterpret_cast<decltype(be.build_c)>(dlsym(h, "salt_gpu_idx_build_c"));
        be.build_r = reinterpret_cast<decltype(be.build_r)>(dlsym(h, "salt_gpu_idx_build_r"));
        be.last_error = reinterpret_cast<decltype(be.last_error)>(dlsym(h, "salt_gpu_idx_last_error"));
        if (!be.build_c || !be.build_r || !be.last_error) { fprintf(stderr, "[salt-idx] --gpu: %s lacks the index-construction entry points\n", lib.c_str()); return 1; }
        bep = &be;
    }
    if (salt_idx_build_ex(argv[optind], argv[optind + 1], argv[optind + 2], k, bep, flags) != 0) {
        fprintf(stderr, "[salt-idx] %s\n", salt_idx_last_error());
        return 1;
    }
    return 0;
}
