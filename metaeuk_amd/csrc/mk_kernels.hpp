// metaeuk_amd/csrc/mk_kernels.hpp -- device-side data structures and launch wrappers shared by the
// HIP kernels (mk_sw.hip, mk_prefilter.hip) and the C-ABI glue (mk_abi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mk {

// One Smith-Waterman DP: rows = query residues q_res[q_start + k*q_step], k < q_len (with their int8
// composition bias), columns = target residues t_res[t_start + c*t_step], c < t_len.
// seg_len = stripe length ceil(q_len / simd lanes) of the reference build being reproduced.
struct SwJob {
    uint64_t t_start;
    uint32_t q_start;
    uint32_t q_len;
    uint32_t t_len;
    int32_t q_step;      // +1 forward, -1 reverse pass
    int32_t t_step;
    uint32_t seg_len;
};
// result: score, first column (in pass order) where the maximum is reached, smallest row with the maximum there
struct SwOut { int32_t score; int32_t end_col; int32_t end_row; int32_t pad; };

struct SwLaunch {
    const uint8_t *q_res; const int8_t *q_bias8;
    const uint8_t *t_res;
    const int8_t *mat;          // 21x21 int8 substitution scores, mat[t*21+q]
    const SwJob *jobs; SwOut *out; uint64_t n_jobs;
    uint2 *boundary;            // multi-tile scratch: per job max_tlen entries (may be null when single tile)
    uint32_t boundary_stride;   // entries per job
    int gap_open, gap_extend;
};
// G lanes per DP (16 or 64), R rows per lane (2,4,8,16)
hipError_t launch_sw(const SwLaunch &L, int G, int R, hipStream_t stream);

struct UngappedJob { uint64_t t_start; uint32_t q_start; uint32_t q_len; uint32_t t_len; uint32_t diagonal; };
struct UngappedLaunch {
    const uint8_t *q_res; const int8_t *q_corr;
    const uint8_t *t_masked;
    const int8_t *mat;          // 21x21 int8: BLOSUM62 x2 (bias -0.2) scores, mat[q*21+t]
    const UngappedJob *jobs; int32_t *out; uint64_t n_jobs;
};
hipError_t launch_ungapped(const UngappedLaunch &L, hipStream_t stream);

}  // namespace mk
