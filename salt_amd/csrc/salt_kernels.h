// salt_amd/csrc/salt_kernels.h -- host-visible declarations of the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/salt_gpu.h"
#include "salt_device.h"

namespace salt {

struct SeedParams {
    uint32_t n_reads, spr;          // spr: seed slots per (read, strand) = ceil((Lmax-k+1)/overlap)
    int32_t  l_seed, l_overlap;
    uint32_t max_seed;
    int32_t  seed_only_ref;
};

struct AlignParams {
    uint32_t n_reads, spr;
    int32_t  l_seed;
    uint32_t max_locate;
    int32_t  max_hits;
    int32_t  all_heavy;             // debug/A-B: skip k_light, k_heavy walks reads 0..n_reads-1
    int32_t  pe;                    // 1: mates of a paired-end batch -- alnse_overlap semantics (alnse.c:985-1044, 501-629):
                                    //    per-interval locate cap, gapped bound stays 3, > 5 N skips the mate
    uint32_t max_amb;               // reads with more N than this are left untouched (200 SE / 5 PE)
};
static const uint32_t PE_LOCI_CAP = 32768;      // loci per strand a PE mate may enumerate (global scratch)

void launch_seed(const IndexView &ix, const SeedParams &sp, const uint8_t *seqs, const uint32_t *offs, uint4 *sai_c,
                 uint4 *sai_r, unsigned long long *ctr, hipStream_t st);
void launch_light(const IndexView &ix, const AlignParams &ap, const uint8_t *seqs, const uint32_t *offs, const uint4 *sai_c,
                  const uint4 *sai_r, salt_result_t *results, uint32_t *queue, uint32_t *qctl, unsigned long long *ctr, hipStream_t st);
void launch_heavy(const IndexView &ix, const AlignParams &ap, const uint8_t *seqs, const uint32_t *offs, const uint4 *sai_c,
                  const uint4 *sai_r, salt_result_t *results, const uint32_t *queue, uint32_t *qctl, unsigned long long *ctr,
                  uint32_t n_blocks, void *lvtab, uint32_t *gq, uint8_t *ge, uint32_t gcap, uint8_t *pe_scr, hipStream_t st);
size_t gap_e_bytes_per_read();
size_t lv_table_bytes();                // per-block LV traceback table (global memory)

uint32_t heavy_blocks_per_cu();
void launch_diag_lv(const IndexView &ix, uint32_t n, const uint32_t *pos, const uint32_t *kdiff, const uint8_t *seqs,
                    const uint32_t *offs, int32_t *out, uint16_t *cig, void *lvtab, hipStream_t st);

// attach-time expansion kernels (salt_index.hip)
void launch_build_c_sa(const IndexView &ix, const uint32_t *sa_sampled, uint32_t sa_intv, uint32_t *out, hipStream_t st);
void launch_build_r_pos(const IndexView &ix, const uint32_t *r_sa, uint32_t *out, hipStream_t st);
void launch_build_r_lkt(const IndexView &ix, uint32_t len, uint2 *out, hipStream_t st);
void launch_build_c_wlkt(const IndexView &ix, uint32_t len, uint2 *out, hipStream_t st);

} // namespace salt
