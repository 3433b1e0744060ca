// salt_amd/host/salt_idx_main.cc -- `salt-idx [-k seedlen] <ref.fa> <snps.txt> <out.prefix>`
// (Index_src/index1.c:46-185: index_main / index_usage; default seed length 25).
#include "../../include/salt_host.h"
#include <getopt.h>
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    int k = 25, c;
    while ((c = getopt(argc, argv, "k:h")) >= 0) {
        if (c == 'k') k = atoi(optarg);
        else { fprintf(stderr, "Usage: salt-idx [-k seed_length(25)] <ref.fa> <snp file> <index prefix>\n"); return 1; }
    }
    if (argc - optind != 3) { fprintf(stderr, "Usage: salt-idx [-k seed_length(25)] <ref.fa> <snp file> <index prefix>\n"); return 1; }
    if (salt_idx_build(argv[optind], argv[optind + 1], argv[optind + 2], k) != 0) {
        fprintf(stderr, "[salt-idx] %s\n", salt_idx_last_error());
        return 1;
    }
    return 0;
}
