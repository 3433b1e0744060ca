/* oracle/polish_oracle_main.c -- TEST INFRASTRUCTURE: `polish_oracle [-s] [-p] <idx> <SAM>`, the CLI around so_polish (the CPU
 * restatement of the reference's Polish_src/polish.c). */
#include "salt_oracle.h"
#include <getopt.h>
#include <stdio.h>
int main(int argc, char **argv)
{
    int c, sw = 0, pe = 0;
    while ((c = getopt(argc, argv, "shp")) >= 0) { if (c == 's') sw = 1; else if (c == 'p') pe = 1; else { fprintf(stderr, "polish_oracle [-s] [-p] <index.prefix> <SAM>\n"); return 0; } }
    if (argc - optind != 2) { fprintf(stderr, "polish_oracle [-s] [-p] <index.prefix> <SAM>\n"); return 0; }
    so_index_t *ix = so_index_load(argv[optind]);
    if (!ix) return 1;
    const int rc = so_polish(ix, argv[optind + 1], sw, pe, stdout);
    so_index_free(ix);
    return rc;
}
