// metaeuk_amd/csrc/mk_sw.hip -- gapped Smith-Waterman and ungapped diagonal scoring kernels for
// gfx950 (wave64).  Replaces SmithWaterman::sw_sse2_byte / sw_sse2_word
// (M/src/alignment/StripedSmithWaterman.cpp:638-940, 942-1214) and
// UngappedAlignment::computeSingelSequenceScores (M/src/prefiltering/UngappedAlignment.cpp:416-431).
//
// SW design (MI355X-first, not a port of the striped-SIMD layout):
//   * one DP per G-lane group (G = 16 = one DPP row, 32, or 64 = a whole wave); every lane owns R
//     consecutive query rows in registers (H, E and a packed best-key per row);
//   * the group sweeps the target as an anti-diagonal wavefront: at step s lane l is on column s-l;
//     what crosses the lane boundary -- H of the lane's last row, the running F, and the target
//     residue itself -- travels lane-to-lane with two DPP row_shr / wave_shr moves per step:
//     no LDS traffic, no bpermute; lane 0 injects residues (and tile borders).  The residues themselves are
//     fetched 16 columns per DPP row, a block of steps ahead, and rotated into lane 0 (row_ror) -- no load in the step;
//   * substitution scores come from an LDS query profile prof[residue][row] (+int8 composition
//     bias), one ds_read of R bytes per lane per column;
//   * the running maximum is a packed key (score << 17 | ~column) per row, so the reference's
//     "first column where the maximum is reached, smallest row in it" costs one v_max_u32 per cell;
//   * queries longer than G*R rows are processed in row tiles; the bottom row of a tile is parked in
//     HBM (4 B per column) and re-read by lane 0 of the next tile;
//   * shared-query mode (pipeline forward pass): the DPs of a wave belong to one query and share ONE LDS profile; the
//     launch is persistent (one-wave workgroups pull waves of jobs from a counter);
//   * swp_kernel: the score-only forward pass in packed int16, two targets per lane group (see below) -- the e-value
//     gate needs nothing but the score, and only the pairs that pass are re-run here for their end cells.
// Recurrence: affine-gap local alignment, H = max(0, diag+s, E, F), E/F opened from H with gap_open and
// extended with gap_extend.  The reference's striped kernels never open E out of a lazy-F-corrected
// cell and restart the in-register F at stripe heads; both only forbid an F-gap directly followed by an
// E-gap, and every such path has an equal-scoring twin with the two gaps swapped (E then F), which is
// allowed.  H is therefore identical cell by cell to plain Gotoh for every SIMD width (DESIGN.md,
// "SW recurrence"); the oracle keeps the literal striped semantics and the tests compare against it.
#include "mk_kernels.hpp"
#include <algorithm>
#include <mutex>

namespace mk {

template <int G>
__device__ __forceinline__ uint32_t shift_up(uint32_t top, uint32_t x, int laneInGroup) {
    // value held by lane-1 of the same group; lane 0 of the group receives `top`
    if constexpr (G == 16) {
        return (uint32_t) __builtin_amdgcn_update_dpp((int) top, (int) x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    } else if constexpr (G == 64) {
        return (uint32_t) __builtin_amdgcn_update_dpp((int) top, (int) x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    } else {
        const uint32_t v = (uint32_t) __builtin_amdgcn_update_dpp((int) top, (int) x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        return laneInGroup == 0 ? top : v;
    }
}

// the same with zero for lane 0 of the group: within a DPP row the hardware's bound control supplies the zero
template <int G>
__device__ __forceinline__ uint32_t shift_up_zero(uint32_t x, int laneInGroup) {
    if constexpr (G == 16) return (uint32_t) __builtin_amdgcn_update_dpp((int) x, (int) x, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    else return shift_up<G>(0u, x, laneInGroup);
}

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

template <int R>
__device__ __forceinline__ void load_scores(const int8_t *p, int (&sc)[R]) {
    if constexpr (R == 2) {
        const uint32_t w = *reinterpret_cast<const uint16_t *>(p);
        sc[0] = (int) (int8_t) (w & 0xFF);
        sc[1] = (int) (int8_t) (w >> 8);
    } else {
        const uint32_t *pw = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
        for (int k = 0; k < R / 4; k++) {
            const uint32_t w = pw[k];
            sc[4 * k + 0] = (int) (int8_t) (w & 0xFF);
            sc[4 * k + 1] = (int) (int8_t) ((w >> 8) & 0xFF);
            sc[4 * k + 2] = (int) (int8_t) ((w >> 16) & 0xFF);
            sc[4 * k + 3] = (int) (int8_t) (w >> 24);
        }
    }
}

// one unit of work: BLOCK/G jobs with their own profiles, or (SHARED) the jobs of wave `unit` on one shared profile
template <int G, int R, int BLOCK, bool SHARED>
__device__ __forceinline__ void sw_unit(const SwLaunch &L, const uint32_t unit, int8_t *smem, const int8_t *sMat) {
    constexpr int GPB = BLOCK / G;
    constexpr int ROWS = G * R;
    static_assert(!SHARED || BLOCK == 64, "shared-query mode runs one wave per workgroup");
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    uint64_t jobId;
    bool have;
    SwJob job;
    uint32_t profQStart = 0; int profQStep = 1;       // query whose profile this lane helps to build
    if constexpr (SHARED) {
        const uint32_t w0 = L.wave_start[unit], w1 = L.wave_start[unit + 1];
        const uint32_t count = min((uint32_t) GPB, w1 - w0);
        have = (uint32_t) grp < count;
        jobId = (uint64_t) w0 + (have ? grp : 0);     // idle groups mirror the first job (same query) with no columns
        job = L.jobs[L.order ? (uint64_t) L.order[jobId] : jobId];
        if (!have) job.t_len = 0;
    } else {
        jobId = (uint64_t) unit * GPB + grp;
        have = jobId < L.n_jobs;
        if (have) job = L.jobs[L.order ? (uint64_t) L.order[jobId] : jobId];
        else { job.t_start = 0; job.q_start = 0; job.q_len = 0; job.t_len = 0; job.q_step = 1; job.t_step = 1; job.slot = 0; }
    }
    profQStart = job.q_start; profQStep = job.q_step;
    int8_t *prof = SHARED ? smem : smem + (size_t) grp * 24 * ROWS;      // 22 profile rows + the staged residues and bias of the tile
    const int go = L.gap_open, ge = L.gap_extend;
    const int qLen = (int) job.q_len, tLen = (int) job.t_len;
    const int nTiles = (qLen + ROWS - 1) / ROWS;
    uint32_t *border = L.boundary ? L.boundary + (jobId - L.boundary_job0) * (uint64_t) L.boundary_stride : nullptr;

    // position / reverse pass: the maximum this DP will reach is known (the score pass found it).  What is asked for is the FIRST column that
    // reaches it and the smallest row there -- final as soon as every lane has passed that column, i.e. G - 1 steps after the first lane saw
    // the score: the DP stops NEED whole blocks of 16 steps after the block in which the score appeared (a reverse job is the whole prefix
    // before the end cell, ~ 200 columns, of which the alignment spans ~ as many as the query has rows)
    constexpr int NEED = (G - 1 + 15) / 16;
    int known = -1, after = 0;
    if constexpr (!SHARED) { if (L.known_score && have && nTiles == 1) known = L.known_score[job.slot]; }
    uint32_t bestKey = 0;
    int bestRow = 0;
    for (int tile = 0; tile < nTiles; tile++) {
        const int row0 = tile * ROWS;
        // ---- LDS query profile for this row tile: prof[t][row] = mat[t][q_row] + bias8[q_row]; row 21 = zeros ----
        // (residues and bias of the tile's rows are staged in LDS first, the matrix sits there too: two global loads per row instead of a
        //  dependent pair per profile entry -- these kernels must not stall when the prefilter of the other stream saturates HBM)
        if (L.q_prof) {
            // profile query (ssw_init with Sequence::getAlignmentProfile, StripedSmithWaterman.cpp:1243-1247): the scores of a row are the
            // query's own column [row][32] (score / 4 per residue, X and "no column" 0), read as dwords: one 32-byte line per row
            for (int i = SHARED ? (int) threadIdx.x : lane; i < ROWS * 6; i += SHARED ? BLOCK : G) {
                const int row = i / 6, w = i - row * 6;
                const int q = row0 + row;
                uint32_t v = 0;
                if (q < qLen) v = *reinterpret_cast<const uint32_t *>(L.q_prof + ((int64_t) profQStart + (int64_t) q * profQStep) * 32 + w * 4);
#pragma unroll
                for (int k = 0; k < 4; k++) if (4 * w + k < 22) prof[(4 * w + k) * ROWS + row] = (int8_t) (v >> (8 * k));
            }
        } else {
            uint8_t *sQ = reinterpret_cast<uint8_t *>(prof) + 22 * ROWS;           // behind the group's / wave's profile
            int8_t *sB = reinterpret_cast<int8_t *>(sQ) + ROWS;
            for (int row = SHARED ? (int) threadIdx.x : lane; row < ROWS; row += SHARED ? BLOCK : G) {
                const int q = row0 + row;
                const bool real = q < qLen;
                const int64_t qi = (int64_t) profQStart + (int64_t) (real ? q : 0) * profQStep;
                sQ[row] = real ? L.q_res[qi] : (uint8_t) 255;
                sB[row] = real ? L.q_bias8[qi] : (int8_t) 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int idx = SHARED ? (int) threadIdx.x : lane; idx < 22 * ROWS; idx += SHARED ? BLOCK : G) {
                const int t = idx / ROWS, row = idx - t * ROWS;
                const uint32_t qc = sQ[row];
                prof[idx] = (t < 21 && qc != 255u) ? (int8_t) ((int) sMat[t * 21 + (int) qc] + (int) sB[row]) : (int8_t) 0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        int H[R], E[R];
        uint32_t key[R];
#pragma unroll
        for (int r = 0; r < R; r++) { H[r] = 0; E[r] = 0; key[r] = 0; }
        uint32_t out0 = 0, out1 = 21u;          // what this lane hands to lane+1: (H_last | F << 16), residue
        int hupPrev = 0;
        const int steps = tLen > 0 ? tLen + G - 1 : 0;
        const bool readTop = (tile > 0);
        const bool writeBottom = (tile + 1 < nTiles);
        // Target residues are fetched 16 columns at a time, one byte per lane of a DPP row, one block of 16 steps ahead
        // of their use; a row rotation per step brings the current column's residue to row lane 0 (= lane 0 of the group).
        const int laneRow = (int) (threadIdx.x & 15u);
        const int64_t tBase = (int64_t) job.t_start;
        const int64_t tStep = (int64_t) job.t_step;
        const int tLast = max(tLen - 1, 0);
        // (fetched three blocks ahead, like the score kernel)
        uint32_t tq0 = L.t_res[tBase + (int64_t) min(laneRow, tLast) * tStep], tq1 = L.t_res[tBase + (int64_t) min(16 + laneRow, tLast) * tStep],
                 tq2 = L.t_res[tBase + (int64_t) min(32 + laneRow, tLast) * tStep];
        for (int s0 = 0; s0 < steps; s0 += 16) {
            uint32_t tcur = tq0;
            tq0 = tq1; tq1 = tq2;
            tq2 = L.t_res[tBase + (int64_t) min(s0 + 48 + laneRow, tLast) * tStep];
            const int sEnd = min(s0 + 16, steps);
            for (int s = s0; s < sEnd; s++) {
                uint32_t top0 = 0;
                const uint32_t top1 = s < tLen ? tcur : 21u;   // residue code 21 = "no column": an all-zero profile row
                tcur = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) tcur, 0x12F /* row_ror:15 = one lane towards lane 0 */, 0xf, 0xf, false);
                if (readTop && lane == 0 && s < tLen) top0 = border[s];
                const uint32_t in0 = shift_up<G>(top0, out0, lane);
                const uint32_t tres = shift_up<G>(top1, out1, lane);
                const int hup = (int) (in0 & 0xFFFFu);
                int F = (int) (in0 >> 16);
                const int c = s - lane;
                const uint32_t cinv = 0x1FFFFu - (uint32_t) max(c, 0);
                int sc[R];
                load_scores<R>(prof + tres * ROWS + lane * R, sc);
                int dsave = hupPrev;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int d = min(dsave + sc[r], 32767);       // int16 saturating add of the reference's word pass
                    dsave = H[r];
                    const int h = max3i(d, E[r], F);               // E >= 0 keeps H non-negative
                    key[r] = max(key[r], ((uint32_t) h << 17) | cinv);
                    const int ho = h - go;
                    E[r] = max3i(E[r] - ge, ho, 0);
                    F = max3i(F - ge, ho, 0);
                    H[r] = h;
                }
                hupPrev = hup;
                out0 = (uint32_t) H[R - 1] | ((uint32_t) F << 16);
                out1 = tres;
                if (writeBottom && lane == G - 1 && c >= 0 && c < tLen) border[c] = out0;
            }
            if constexpr (!SHARED) {
                if (known > 0) {                                   // (the same for every lane of the group)
                    bool seen = false;
#pragma unroll
                    for (int r = 0; r < R; r++) seen |= (key[r] >> 17) == (uint32_t) known;
                    const unsigned long long m = __ballot(seen);
                    const unsigned long long mine = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull) << (((unsigned) threadIdx.x & 63u) / G * G);
                    if ((m & mine) != 0ull && ++after > NEED) break;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (key[r] > bestKey) { bestKey = key[r]; bestRow = row0 + lane * R + r; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- group reduction: largest key, then smallest row ----
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) {
        const uint32_t ok = (uint32_t) __shfl_xor((int) bestKey, m, G);
        const int orow = __shfl_xor(bestRow, m, G);
        if (ok > bestKey || (ok == bestKey && orow < bestRow)) { bestKey = ok; bestRow = orow; }
    }
    if (have && lane == 0) {
        SwOut o;
        o.score = (int32_t) (bestKey >> 17);
        o.end_col = o.score > 0 ? (int32_t) (0x1FFFFu - (bestKey & 0x1FFFFu)) : -1;
        o.end_row = o.score > 0 ? bestRow : -1;
        o.pad = 0;
        L.out[job.slot] = o;
    }
}

// Shared-query mode is a persistent launch: a fixed number of one-wave workgroups per CU pull units from a counter, which
// leaves wave slots, registers and LDS on every CU for the (memory-latency-bound) prefilter kernels of the other stream.
#ifndef MK_HELPER_PRIO
#define MK_HELPER_PRIO 3         // older waves win the CU's issue arbitration: short launches beside persistent workgroups ask for priority
#endif
template <int G, int R, int BLOCK, bool SHARED>
__global__ __launch_bounds__(BLOCK) void sw_kernel(SwLaunch L) {
    if (!SHARED && MK_HELPER_PRIO) __builtin_amdgcn_s_setprio(MK_HELPER_PRIO);       // position / reverse passes: short launches beside the prefilter's persistent waves
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    __shared__ int8_t sMat[448];
    for (int k = (int) threadIdx.x; k < 441; k += BLOCK) sMat[k] = L.mat[k];
    __syncthreads();
    if constexpr (SHARED) {
        for (uint32_t done = 0; L.units_per_block == 0 || done < L.units_per_block; done++) {
            uint32_t u = 0;
            if (threadIdx.x == 0) u = atomicAdd(L.work_counter, 1u);
            u = (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
            if ((uint64_t) u >= L.n_waves) break;
            sw_unit<G, R, BLOCK, true>(L, u, smem, sMat);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        sw_unit<G, R, BLOCK, false>(L, blockIdx.x, smem, sMat);
    }
}

// Position / reverse pass of the pipeline as ONE launch per register class (round 5).  The jobs are ordered by (tile configuration, target
// length class) -- a counting sort on the device -- and `bounds[c]` = first job of tile configuration c stays on the device: nothing of it
// passes through the host, where rounds 1-4 fetched the bounds, synchronised, and launched one kernel per tile configuration (~ 100 short
// launches of ~ 6 000 jobs per config-2 step, each behind a host round trip, beside three streams of persistent workgroups).  Persistent
// one-wave workgroups pull units of 64 / G jobs from a counter, the largest tile configuration of the class first; a unit runs in the lane
// shape of its tile configuration (sw_unit, own profile per job).  Three classes so that the small tiles keep their occupancy: <= 64 rows
// (72 VGPRs, 6 KB of LDS per wave), 96 .. 256 rows and >= 384 rows (128 VGPRs, 24.5 KB).
template <int CLS>
__global__ __launch_bounds__(64) void sw_multi_kernel(SwLaunch L, const uint32_t *bounds, uint32_t *counter, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(MK_HELPER_PRIO);
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    __shared__ int8_t sMat[448];
    for (int k = (int) threadIdx.x; k < 441; k += 64) sMat[k] = L.mat[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int C0 = CLS == 0 ? 0 : (CLS == 1 ? 3 : 7), C1 = CLS == 0 ? 3 : (CLS == 1 ? 7 : SW_NCFG), NC = C1 - C0;
    // units of every tile configuration of the class, the largest configuration first
    uint32_t lo[NC], nj[NC], ub[NC + 1];
    ub[0] = 0;
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int c = C1 - 1 - k;
        lo[k] = bounds[c]; nj[k] = bounds[c + 1] - lo[k];
        const uint32_t gpb = c >= 9 ? 1u : (c >= 7 ? 2u : 4u);         // 64 / G jobs per wave: G = 64 for 768 / 1024 rows, 32 for 384 / 512, 16 below
        ub[k + 1] = ub[k] + (nj[k] + gpb - 1u) / gpb;
    }
    for (;;) {
        uint32_t u = 0;
        if (threadIdx.x == 0) u = atomicAdd(counter, 1u);
        u = (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
        if (u >= ub[NC]) break;
        int k = 0;
#pragma unroll
        for (int j = 1; j < NC; j++) if (u >= ub[j]) k = j;
        SwLaunch Lc = L;
        uint32_t first = 0, base = 0, count = 0;
#pragma unroll
        for (int j = 0; j < NC; j++) if (j == k) { first = lo[j]; base = ub[j]; count = nj[j]; }
        Lc.order = L.order + first; Lc.n_jobs = count;
        const uint32_t unit = u - base;
        const int c = C1 - 1 - k;
        if constexpr (CLS == 0) {
            if (c == 0) sw_unit<16, 2, 64, false>(Lc, unit, smem, sMat);
            else sw_unit<16, 4, 64, false>(Lc, unit, smem, sMat);                 // 48 and 64 rows
        } else if constexpr (CLS == 1) {
            if (c <= 4) sw_unit<16, 8, 64, false>(Lc, unit, smem, sMat);         // 96 and 128 rows
            else sw_unit<16, 16, 64, false>(Lc, unit, smem, sMat);               // 192 and 256 rows
        } else {
            if (c == 7) sw_unit<32, 12, 64, false>(Lc, unit, smem, sMat);
            else if (c == 8) sw_unit<32, 16, 64, false>(Lc, unit, smem, sMat);
            else if (c == 9) sw_unit<64, 12, 64, false>(Lc, unit, smem, sMat);
            else sw_unit<64, 16, 64, false>(Lc, unit, smem, sMat);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// L.order = the jobs' order by (tile configuration, target length class), bounds = device array of SW_NCFG + 1 first-job indices into it,
// counter = a zeroed device word; cls 0 / 1 / 2 = tile configurations of <= 64 / 96 .. 256 / >= 384 rows.  Single-tile jobs only (no border).
hipError_t launch_sw_multi(const SwLaunch &L, const uint32_t *bounds, uint32_t *counter, int cls, uint32_t blocks, hipStream_t stream, bool prio) {
    if (!L.order || !bounds || !counter || L.boundary || blocks == 0) return hipErrorInvalidValue;
    switch (cls) {
        case 0: hipLaunchKernelGGL((sw_multi_kernel<0>), dim3(blocks), dim3(64), (size_t) 64 * 24 * 4, stream, L, bounds, counter, prio ? 1 : 0); break;
        case 1: hipLaunchKernelGGL((sw_multi_kernel<1>), dim3(blocks), dim3(64), (size_t) 64 * 24 * 16, stream, L, bounds, counter, prio ? 1 : 0); break;
        case 2: hipLaunchKernelGGL((sw_multi_kernel<2>), dim3(blocks), dim3(64), (size_t) 64 * 24 * 16, stream, L, bounds, counter, prio ? 1 : 0); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Score-only forward pass, two DPs per lane group in packed int16 (v_pk_add/max/sub_i16): the low halves of every
// register belong to one target, the high halves to another; both run against the SAME query (shared-query mode), so
// all DP state packs perfectly -- the serial F dependence runs down the rows inside each half.  ~5.5 vector ops per cell
// instead of 10.  It yields the maximum score only: the e-value gate needs nothing else, and the ~9 % of the pairs that
// pass are re-run by sw_kernel for their end positions.  For tiles of at most 256 rows the int16 arithmetic cannot
// saturate (256 rows x 127); beyond, diag + score saturates at 32767 exactly like the reference's word pass.
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk_from(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ uint32_t pk_bits(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ pk16 pk_splat(int v) { pk16 r; r.x = (short) v; r.y = (short) v; return r; }
// a - b clamped at 0 for non-negative halves (v_pk_sub_u16 clamp): the reference's simdui16_subs on gap penalties
typedef unsigned short pku16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk_subs0(pk16 a, pk16 b) {
    return __builtin_bit_cast(pk16, __builtin_elementwise_sub_sat(__builtin_bit_cast(pku16, a), __builtin_bit_cast(pku16, b)));
}

// ---------------------------------------------------------------------------------------------
// Round 6: the score pass of PROFILE queries, transposed.  A profile of 300-700 columns meets ORF fragments of 40-100 residues
// (searchslicedtargetprofile.sh: the fragments are the targets): with the profile in the lanes' rows a 512-row tile on 32 lanes spends 31 ramp steps per
// ~ 90 useful ones and a fifth of its rows on padding -- 45-58 % of the lane-steps carry a cell (DESIGN.md 8.3 of round 5).  Here the FRAGMENT lies in the
// rows (16 lanes x R rows, R chosen per wave from its longest fragment: 32 ... 256 rows) and the profile's columns are the steps: qLen + 15 of them, the
// rows filled to ~ 90 %.  Two fragments per lane group in the packed halves, eight per wave, all against one profile -- the jobs of a wave as before.
// The scores of a step come from an LDS image of the profile LETTER-major, prof[t][16 + c] (int8, 16 zero entries in front of column 0 and 32 behind the
// last: the ramps read zeros without a test): a row's address is (its residue's row + the lane's column at the head of the block) + the step within the
// unrolled block as the instruction's immediate offset -- no address arithmetic per cell.  H, E, F and the lane hand-over are swp_kernel's; what
// crosses the lanes is H of a lane's last row and F (two DPP moves), no residues.  The maximum is kept per ROW, so the bound handed to the position pass
// is exact: the first fragment position that reaches the maximum.  Fragments of at most 256 rows: 256 x 127 < 32 767, no saturation.
constexpr uint32_t SWT_MAX_ROWS = 256;                            // the longest fragment the transposed kernels take (16 lanes x 16 rows)
constexpr int SWT_MARGIN = 16, SWT_TAIL = 32, SWT_COL = 24;       // zero columns in front of / behind the profile; bytes per column of the LDS image
__host__ __device__ inline size_t swt_image_bytes(int qLen) { return (size_t) (SWT_MARGIN + qLen + SWT_TAIL) * SWT_COL; }

template <int R>
__device__ __forceinline__ void swt_unit(const SwLaunch &L, const uint32_t w0, const uint32_t count, const int qLen) {
    constexpr int G = 16;
    // steps per trip of the column loop: what it reads ahead (2 R scores per step, a register each) stays within ~ 64 registers
    constexpr int UNR = R <= 2 ? 16 : (R <= 4 ? 8 : (R <= 8 ? 4 : 2));
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];      // the image (the kernel's dynamic LDS): addressed by 32-bit offsets, read with ds_read_i8
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const pk16 go2 = pk_splat(L.gap_open), ge2 = pk_splat(L.gap_extend), zero2 = pk_splat(0);
    const bool haveA = (uint32_t) (2 * grp) < count, haveB = (uint32_t) (2 * grp + 1) < count;
    const SwJob jobA = L.jobs[L.order[(uint64_t) w0 + (haveA ? 2 * grp : 0)]];
    const SwJob jobB = L.jobs[L.order[(uint64_t) w0 + (haveB ? 2 * grp + 1 : 0)]];
    const int tLenA = haveA ? (int) jobA.t_len : 0, tLenB = haveB ? (int) jobB.t_len : 0;
    // this lane's rows: the residue of the fragment position (21 = the zero entry of every column: no fragment position here), as the LDS offset of that
    // entry in this lane's column at the head of the current trip -- column s0 - lane, >= -15 (the front margin)
    uint32_t offA[R], offB[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = lane * R + r;
        const uint32_t a = i < tLenA ? (uint32_t) L.t_res[(int64_t) jobA.t_start + (int64_t) i * jobA.t_step] : 21u;
        const uint32_t b = i < tLenB ? (uint32_t) L.t_res[(int64_t) jobB.t_start + (int64_t) i * jobB.t_step] : 21u;
        offA[r] = min(a, 21u) + (uint32_t) ((SWT_MARGIN - lane) * SWT_COL); offB[r] = min(b, 21u) + (uint32_t) ((SWT_MARGIN - lane) * SWT_COL);
    }
    pk16 H[R], E[R], best[R];
#pragma unroll
    for (int r = 0; r < R; r++) { H[r] = zero2; E[r] = zero2; best[r] = zero2; }
    pk16 hupPrev = zero2;
    uint32_t outH = 0, outF = 0;
    const int steps = (qLen + G - 1 + 15) & ~15;                       // (<= qLen + 30: inside the tail margin)
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += UNR) {
#pragma unroll
        for (int k = 0; k < UNR; k++) {
            const pk16 hup = pk_from(shift_up_zero<G>(outH, lane));
            pk16 F = pk_from(shift_up_zero<G>(outF, lane));
            pk16 dsave = hupPrev;
#pragma unroll
            for (int r = 0; r < R; r++) {
                // ds_read_i8, the step as the immediate offset (k columns on)
                const int sa = (int) smem[offA[r] + (uint32_t) (k * SWT_COL)], sb = (int) smem[offB[r] + (uint32_t) (k * SWT_COL)];
                const pk16 sc = pk_from(__builtin_amdgcn_perm((uint32_t) sb, (uint32_t) sa, 0x05040100u));
                const pk16 d = dsave + sc;
                dsave = H[r];
                const pk16 h = pk_max(pk_max(d, E[r]), F);
                best[r] = pk_max(best[r], h);
                const pk16 ho = pk_subs0(h, go2);
                E[r] = pk_max(pk_subs0(E[r], ge2), ho);
                F = pk_max(pk_subs0(F, ge2), ho);
                H[r] = h;
            }
            hupPrev = hup;
            outH = pk_bits(H[R - 1]);
            outF = pk_bits(F);
        }
#pragma unroll
        for (int r = 0; r < R; r++) { offA[r] += (uint32_t) (UNR * SWT_COL); offB[r] += (uint32_t) (UNR * SWT_COL); }
    }
    // the two maxima of the group, and the first ROW (fragment position) that reaches each
    pk16 bAll = zero2;
#pragma unroll
    for (int r = 0; r < R; r++) bAll = pk_max(bAll, best[r]);
    uint32_t b = pk_bits(bAll);
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) b = pk_bits(pk_max(pk_from(b), pk_from((uint32_t) __shfl_xor((int) b, m, G))));
    uint32_t cA = 0x7FFFFFFFu, cB = 0x7FFFFFFFu;
#pragma unroll
    for (int r = R - 1; r >= 0; r--) {
        if ((pk_bits(best[r]) & 0xFFFFu) == (b & 0xFFFFu)) cA = (uint32_t) (lane * R + r);
        if ((pk_bits(best[r]) >> 16) == (b >> 16)) cB = (uint32_t) (lane * R + r);
    }
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) {
        cA = min(cA, (uint32_t) __shfl_xor((int) cA, m, G));
        cB = min(cB, (uint32_t) __shfl_xor((int) cB, m, G));
    }
    if (lane == 0) {
        SwOut o;
        o.end_row = -1; o.pad = 0;                                  // end_col: the first target column that reaches the maximum (the position pass stops there)
        if (haveA) { o.score = (int32_t) (int16_t) (b & 0xFFFFu); o.end_col = (int32_t) min(cA, (uint32_t) max(tLenA - 1, 0)); L.out[jobA.slot] = o; }
        if (haveB) { o.score = (int32_t) (int16_t) (b >> 16); o.end_col = (int32_t) min(cB, (uint32_t) max(tLenB - 1, 0)); L.out[jobB.slot] = o; }
    }
}

// The POSITION pass of the same jobs, transposed (round 6): the maximum S of a job is known from the score pass; asked for is the first target column
// that reaches it and the smallest query row in that column (StripedSmithWaterman.cpp:820-850) -- here: the smallest fragment ROW with a cell equal to S and the
// first profile COLUMN in that row.  Per row and step five instructions beside the recurrence (no running maximum is kept): x = h ^ S, e = 1 -sat x (1 where
// the half is zero), the inverted mask e + 0xFFFF, the candidate column | mask, an unsigned minimum into the row's first column.  Every profile column is
// walked (the rule's first key is the row: no early exit), the rows are the target cut at the score pass's bound.  A padding row or a column behind the
// profile can only repeat S one row further down / in a later column than a real cell that holds it, so the smallest (row, column) is a real cell.
template <int R>
__device__ __forceinline__ void swtp_unit(const SwLaunch &L, const uint32_t (&jobIdx)[8], const uint32_t count, const int qLen) {
    constexpr int G = 16;
    constexpr int UNR = R <= 2 ? 16 : (R <= 4 ? 8 : (R <= 8 ? 4 : 2));
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const pk16 go2 = pk_splat(L.gap_open), ge2 = pk_splat(L.gap_extend), zero2 = pk_splat(0);
    const bool haveA = (uint32_t) (2 * grp) < count, haveB = (uint32_t) (2 * grp + 1) < count;
    uint32_t ia = jobIdx[0], ib = jobIdx[0];
#pragma unroll
    for (int k = 0; k < 8; k++) { if (k == 2 * grp && haveA) ia = jobIdx[k]; if (k == 2 * grp + 1 && haveB) ib = jobIdx[k]; }
    const SwJob jobA = L.jobs[ia], jobB = L.jobs[ib];
    const int tLenA = haveA ? (int) jobA.t_len : 0, tLenB = haveB ? (int) jobB.t_len : 0;
    const uint32_t SA = haveA ? (uint32_t) L.known_score[jobA.slot] & 0xFFFFu : 0x7FFFu, SB = haveB ? (uint32_t) L.known_score[jobB.slot] & 0xFFFFu : 0x7FFFu;
    const uint32_t S2 = SA | (SB << 16);
    uint32_t offA[R], offB[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int i = lane * R + r;
        const uint32_t a = i < tLenA ? (uint32_t) L.t_res[(int64_t) jobA.t_start + (int64_t) i * jobA.t_step] : 21u;
        const uint32_t b = i < tLenB ? (uint32_t) L.t_res[(int64_t) jobB.t_start + (int64_t) i * jobB.t_step] : 21u;
        offA[r] = min(a, 21u) + (uint32_t) ((SWT_MARGIN - lane) * SWT_COL); offB[r] = min(b, 21u) + (uint32_t) ((SWT_MARGIN - lane) * SWT_COL);
    }
    pk16 H[R], E[R];
    uint32_t first[R];                                              // per half: the first column of this row with H == S (0xFFFF: none yet)
#pragma unroll
    for (int r = 0; r < R; r++) { H[r] = zero2; E[r] = zero2; first[r] = 0xFFFFFFFFu; }
    pk16 hupPrev = zero2;
    uint32_t outH = 0, outF = 0;
    const uint32_t c0 = (uint32_t) (-lane) & 0xFFFFu;
    uint32_t cvec = c0 | (c0 << 16);                                // this lane's column, in both halves (negative columns: 0xFFF1 ..: never a minimum)
    const int steps = (qLen + G - 1 + 15) & ~15;
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += UNR) {
#pragma unroll
        for (int k = 0; k < UNR; k++) {
            const pk16 hup = pk_from(shift_up_zero<G>(outH, lane));
            pk16 F = pk_from(shift_up_zero<G>(outF, lane));
            pk16 dsave = hupPrev;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int sa = (int) smem[offA[r] + (uint32_t) (k * SWT_COL)], sb = (int) smem[offB[r] + (uint32_t) (k * SWT_COL)];
                const pk16 sc = pk_from(__builtin_amdgcn_perm((uint32_t) sb, (uint32_t) sa, 0x05040100u));
                const pk16 d = dsave + sc;
                dsave = H[r];
                const pk16 h = pk_max(pk_max(d, E[r]), F);
                // first column with h == S, per half
                const pku16 x = __builtin_bit_cast(pku16, pk_bits(h) ^ S2);
                const pku16 one = {1, 1}, ffff = {0xFFFF, 0xFFFF};
                const pku16 e = __builtin_elementwise_sub_sat(one, x);                       // 1 where the half is zero
                const uint32_t inv = __builtin_bit_cast(uint32_t, (pku16) (e + ffff));       // 0x0000 there, 0xFFFF elsewhere
                first[r] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pku16, first[r]), __builtin_bit_cast(pku16, cvec | inv)));
                const pk16 ho = pk_subs0(h, go2);
                E[r] = pk_max(pk_subs0(E[r], ge2), ho);
                F = pk_max(pk_subs0(F, ge2), ho);
                H[r] = h;
            }
            hupPrev = hup;
            outH = pk_bits(H[R - 1]);
            outF = pk_bits(F);
            cvec = __builtin_bit_cast(uint32_t, (pku16) (__builtin_bit_cast(pku16, cvec) + (pku16) {1, 1}));
        }
#pragma unroll
        for (int r = 0; r < R; r++) { offA[r] += (uint32_t) (UNR * SWT_COL); offB[r] += (uint32_t) (UNR * SWT_COL); }
    }
    // smallest row with a hit, and its first column: key = row << 16 | column, minimum over the lanes of the group
    uint32_t kA = 0xFFFFFFFFu, kB = 0xFFFFFFFFu;
#pragma unroll
    for (int r = R - 1; r >= 0; r--) {
        if ((first[r] & 0xFFFFu) < (uint32_t) qLen) kA = ((uint32_t) (lane * R + r) << 16) | (first[r] & 0xFFFFu);
        if ((first[r] >> 16) < (uint32_t) qLen) kB = ((uint32_t) (lane * R + r) << 16) | (first[r] >> 16);
    }
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) {
        kA = min(kA, (uint32_t) __shfl_xor((int) kA, m, G));
        kB = min(kB, (uint32_t) __shfl_xor((int) kB, m, G));
    }
    if (lane == 0) {
        SwOut o;
        o.pad = 0;
        if (haveA) {
            const bool ok = kA != 0xFFFFFFFFu && (int) (kA >> 16) < tLenA;
            o.score = ok ? (int32_t) SA : 0; o.end_col = ok ? (int32_t) (kA >> 16) : -1; o.end_row = ok ? (int32_t) (kA & 0xFFFFu) : -1;
            L.out[jobA.slot] = o;
        }
        if (haveB) {
            const bool ok = kB != 0xFFFFFFFFu && (int) (kB >> 16) < tLenB;
            o.score = ok ? (int32_t) SB : 0; o.end_col = ok ? (int32_t) (kB >> 16) : -1; o.end_row = ok ? (int32_t) (kB & 0xFFFFu) : -1;
            L.out[jobB.slot] = o;
        }
    }
}

// the profile image of swt_unit / swtp_unit for the query that starts at column qStart
__device__ __forceinline__ void swt_image(const SwLaunch &L, const uint32_t qStart, const int qStep, const int qLen, int8_t *smem) {
    uint32_t *img32 = reinterpret_cast<uint32_t *>(smem);
    constexpr int CW = SWT_COL / 4;
    for (int i = (int) threadIdx.x; i < SWT_MARGIN * CW; i += 64) img32[i] = 0u;
    for (int i = (int) threadIdx.x; i < SWT_TAIL * CW; i += 64) img32[(SWT_MARGIN + qLen) * CW + i] = 0u;
    for (int i = (int) threadIdx.x; i < qLen * 6; i += 64) {
        const int c = i / 6, w = i - c * 6;
        uint32_t v = *reinterpret_cast<const uint32_t *>(L.q_prof + ((int64_t) qStart + (int64_t) c * qStep) * 32 + w * 4);
        if (w == 5) v &= 0xFFu;                                      // (letters 20 | 21 22 23: entry 21 is the "no row" zero)
        img32[(SWT_MARGIN + c) * CW + w] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// position jobs of profile queries in units of 8 consecutive entries of L.order (sorted so that a query's jobs are neighbours: mk_align.hip gate_emit_kernel);
// inside a unit the jobs are taken query by query (one LDS image each); CLS as in swt_kernel, per run of a query
template <int CLS>
__global__ __launch_bounds__(64) void swtp_kernel(SwLaunch L) {
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    if (MK_HELPER_PRIO) __builtin_amdgcn_s_setprio(MK_HELPER_PRIO);
    constexpr uint32_t LO = CLS == 0 ? 0u : (CLS == 1 ? 64u : 128u), HI = CLS == 0 ? 64u : (CLS == 1 ? 128u : SWT_MAX_ROWS);
    constexpr uint32_t UNIT = 8;                                    // jobs looked at together (32: 41 -> 50 / 62 -> 76 ms, the runs mix target lengths): a query's run inside them is taken 8 jobs at a time
    const uint64_t nUnits = (L.n_jobs + UNIT - 1) / UNIT;
    for (uint64_t u = blockIdx.x; u < nUnits; u += gridDim.x) {
        const uint64_t j0 = u * UNIT;
        const uint32_t cnt = (uint32_t) min((uint64_t) UNIT, L.n_jobs - j0);
        // lanes 0 .. UNIT - 1 look at one job each
        uint32_t myJob = 0, myQ = 0xFFFFFFFFu, myT = 0, myL = 0;
        if (threadIdx.x < cnt) { myJob = L.order[j0 + threadIdx.x]; const SwJob j = L.jobs[myJob]; myQ = j.q_start; myT = j.t_len; myL = j.q_len; }
        uint32_t todo = (uint32_t) (__ballot(threadIdx.x < cnt) & 0xFFFFFFFFull);
        while (todo) {
            const int lead = __ffs((int) todo) - 1;
            const uint32_t q = (uint32_t) __shfl((int) myQ, lead, 64), qLen = (uint32_t) __shfl((int) myL, lead, 64);
            uint32_t run = (uint32_t) (__ballot(threadIdx.x < UNIT && ((todo >> (threadIdx.x & 31u)) & 1u) && myQ == q && myL == qLen) & 0xFFFFFFFFull);
            todo &= ~run;
            // the longest target of the run decides the class (a class's kernel takes whole runs)
            uint32_t tl = (threadIdx.x < UNIT && ((run >> (threadIdx.x & 31u)) & 1u)) ? myT : 0u;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) tl = max(tl, (uint32_t) __shfl_xor((int) tl, m, 64));
            if (tl > HI || (CLS > 0 && tl <= LO)) continue;
            swt_image(L, q, 1, (int) qLen, smem);
            while (run) {
                uint32_t jobIdx[8];
                uint32_t n = 0;
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    if (run) {
                        const int k = __ffs((int) run) - 1;
                        run &= run - 1u;
                        jobIdx[s] = (uint32_t) __shfl((int) myJob, k, 64);
                        n++;
                    } else jobIdx[s] = jobIdx[0];
                }
                if constexpr (CLS == 0) {
                    if (tl <= 32u) swtp_unit<2>(L, jobIdx, n, (int) qLen);
                    else if (tl <= 48u) swtp_unit<3>(L, jobIdx, n, (int) qLen);
                    else swtp_unit<4>(L, jobIdx, n, (int) qLen);
                } else if constexpr (CLS == 1) {
                    if (tl <= 96u) swtp_unit<6>(L, jobIdx, n, (int) qLen);
                    else swtp_unit<8>(L, jobIdx, n, (int) qLen);
                } else {
                    if (tl <= 192u) swtp_unit<12>(L, jobIdx, n, (int) qLen);
                    else swtp_unit<16>(L, jobIdx, n, (int) qLen);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// the longest fragment among the jobs of a wave: up to 256 residues the transposed kernel takes the wave, beyond the classic one
__device__ __forceinline__ uint32_t swt_longest(const SwLaunch &L, const uint32_t w0, const uint32_t count) {
    uint32_t tl = 0;
    if (threadIdx.x < count) tl = L.jobs[L.order[(uint64_t) w0 + threadIdx.x]].t_len;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) tl = max(tl, (uint32_t) __shfl_xor((int) tl, m, 64));
    return tl;
}

// the profile of the wave's query as the LDS image swt_unit reads -- [SWT_MARGIN + column][24 bytes: the 21 scores, entry 21 = 0], zero columns around
// it: the query's own 32-byte lines copied dword by dword --, then the unit in the shape the longest fragment of the wave needs
template <int CLS>
__device__ __forceinline__ void swt_wave(const SwLaunch &L, const uint32_t w0, const uint32_t count, const uint32_t tl, int8_t *smem) {
    const SwJob j0 = L.jobs[L.order[w0]];
    const int qLen = (int) j0.q_len;
    swt_image(L, j0.q_start, j0.q_step, qLen, smem);
    if constexpr (CLS == 0) {
        if (tl <= 32u) swt_unit<2>(L, w0, count, qLen);
        else if (tl <= 48u) swt_unit<3>(L, w0, count, qLen);
        else swt_unit<4>(L, w0, count, qLen);
    } else if constexpr (CLS == 1) {
        if (tl <= 96u) swt_unit<6>(L, w0, count, qLen);
        else swt_unit<8>(L, w0, count, qLen);
    } else {
        if (tl <= 192u) swt_unit<12>(L, w0, count, qLen);
        else swt_unit<16>(L, w0, count, qLen);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// persistent one-wave workgroups over the waves of ONE tile configuration of a profile-query batch (the same wave list swp_kernel walks, a counter of
// its own): the waves whose fragments all fit 256 rows; swp_kernel skips exactly those
// (three register classes, like the position / reverse passes: fragments of up to 64 / 128 / 256 rows -- 2-4 / 6-8 / 12-16 rows per lane)
template <int CLS>
__global__ __launch_bounds__(64) void swt_kernel(SwLaunch L) {
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    constexpr uint32_t LO = CLS == 0 ? 0u : (CLS == 1 ? 64u : 128u), HI = CLS == 0 ? 64u : (CLS == 1 ? 128u : SWT_MAX_ROWS);
    for (uint32_t done = 0; L.units_per_block == 0 || done < L.units_per_block; done++) {
        uint32_t u = 0;
        if (threadIdx.x == 0) u = atomicAdd(L.work_counter_t + 16 * CLS, 1u);
        u = (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
        if ((uint64_t) u >= L.n_waves) break;
        const uint32_t w0 = L.wave_start[u], count = min(8u, L.wave_start[u + 1] - w0);
        const uint32_t tl = swt_longest(L, w0, count);
        if (tl > HI || (CLS > 0 && tl <= LO) || tl > L.t_max_rows) continue;
        swt_wave<CLS>(L, w0, count, tl, smem);
    }
}

// R rows per lane; the LDS profile keeps RP >= R (even) int16 slots per lane so that a lane's scores are whole dwords.
// G = 16 lanes per pair of DPs for tiles of at most 256 rows; G = 32 for 384 / 512 rows, where diag + score can pass 32767
// and the add saturates like the reference's word pass (simdi16_adds, StripedSmithWaterman.cpp:1059).
template <int R, int RP = R, int G = 16>
__global__ __launch_bounds__(64) void swp_kernel(SwLaunch L) {
    constexpr int GPB = 64 / G, ROWS = G * RP;
    constexpr bool SAT = G * R * 127 > 32767;
    static_assert(RP >= R && RP % 2 == 0, "profile slots per lane: even and at least R");
    static_assert(G == 16 || G == 32, "a pair of DPs runs on one or two DPP rows");
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int16_t *prof = reinterpret_cast<int16_t *>(smem);            // prof[t][row], int16; row 21 = zeros
    // behind the profile: the substitution matrix (loaded once per persistent wave) and the query's residues / composition bias of the
    // current unit -- the profile is then built from LDS, two global loads per lane instead of a dependent pair per profile entry
    int8_t *sMat = smem + (size_t) 22 * ROWS * sizeof(int16_t);
    uint8_t *sQ = reinterpret_cast<uint8_t *>(sMat + 448);
    int8_t *sB = reinterpret_cast<int8_t *>(sQ + ROWS);
    for (int k = (int) threadIdx.x; k < 441; k += 64) sMat[k] = L.mat[k];
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const pk16 go2 = pk_splat(L.gap_open), ge2 = pk_splat(L.gap_extend), zero2 = pk_splat(0);
    for (uint32_t done = 0; L.units_per_block == 0 || done < L.units_per_block; done++) {
        uint32_t u = 0;
        if (threadIdx.x == 0) u = atomicAdd(L.work_counter, 1u);
        u = (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
        if ((uint64_t) u >= L.n_waves) break;
        const uint32_t wFirst = L.wave_start[u], wEnd = L.wave_start[u + 1];
        // profile queries (L.narrow: their waves hold up to 8 jobs whatever the tile): the transposed unit takes the wave unless a fragment is long
        if (L.t_max_rows && swt_longest(L, wFirst, min(8u, wEnd - wFirst)) <= L.t_max_rows) continue;       // swt_kernel's
        for (uint32_t w0 = wFirst; w0 < wEnd; w0 += (uint32_t) (2 * GPB)) {          // (a wave of 8 jobs on a 32-lane tile: two rounds of 4)
        const uint32_t w1 = wEnd;
        const uint32_t count = min((uint32_t) (2 * GPB), w1 - w0);
        const bool haveA = (uint32_t) (2 * grp) < count, haveB = (uint32_t) (2 * grp + 1) < count;
        const SwJob jobA = L.jobs[L.order[(uint64_t) w0 + (haveA ? 2 * grp : 0)]];
        const SwJob jobB = L.jobs[L.order[(uint64_t) w0 + (haveB ? 2 * grp + 1 : 0)]];
        const int qLen = (int) jobA.q_len;                          // every job of the wave has this query
        const int tLenA = haveA ? (int) jobA.t_len : 0, tLenB = haveB ? (int) jobB.t_len : 0;
        if (L.q_prof) {
            // profile query: the row's scores are its own column of the alignment profile (see sw_unit)
            for (int i = (int) threadIdx.x; i < ROWS * 6; i += 64) {
                const int slot = i / 6, w = i - slot * 6;
                const int row = RP == R ? slot : (slot / RP) * R + slot % RP;
                const bool real = row < qLen && (RP == R || slot % RP < R);
                uint32_t v = 0;
                if (real) v = *reinterpret_cast<const uint32_t *>(L.q_prof + ((int64_t) jobA.q_start + (int64_t) row * jobA.q_step) * 32 + w * 4);
#pragma unroll
                for (int k = 0; k < 4; k++) if (4 * w + k < 22) prof[(4 * w + k) * ROWS + slot] = (int16_t) (int8_t) (v >> (8 * k));
            }
        } else {
        for (int slot = (int) threadIdx.x; slot < ROWS; slot += 64) {
            const int row = RP == R ? slot : (slot / RP) * R + slot % RP;    // lane-major: lane * R + r
            const bool real = row < qLen && (RP == R || slot % RP < R);
            const int64_t qi = (int64_t) jobA.q_start + (int64_t) (real ? row : 0) * jobA.q_step;
            sQ[slot] = real ? L.q_res[qi] : (uint8_t) 255;
            sB[slot] = real ? L.q_bias8[qi] : (int8_t) 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int idx = (int) threadIdx.x; idx < 22 * ROWS; idx += 64) {
            const int t = idx / ROWS, slot = idx - t * ROWS;
            const uint32_t qc = sQ[slot];
            prof[idx] = (t < 21 && qc != 255u) ? (int16_t) ((int) sMat[t * 21 + (int) qc] + (int) sB[slot]) : (int16_t) 0;
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        pk16 H[R], E[R];
#pragma unroll
        for (int r = 0; r < R; r++) { H[r] = zero2; E[r] = zero2; }
        pk16 best = zero2, hupPrev = zero2;
        int roseA = 0, roseB = 0;                                  // first step of the block of 16 in which this lane's running maximum last rose
        // what travels with a column is not the two residues but the byte offsets of their profile rows (residue * ROWS * 2, code 21 =
        // "no column": an all-zero row), packed in one dword and prepared when the column is fetched -- nothing per step
        constexpr uint32_t ROWB = (uint32_t) ROWS * 2u, NOCOL = 21u * ROWB | (21u * ROWB) << 16;
        uint32_t outH = 0, outF = 0, outRes = NOCOL;              // handed to lane+1: H of the last row, F, the two profile-row offsets
        const int tMax = max(tLenA, tLenB);
        // whole blocks of 16 steps (the steps past the last column see all-zero profile rows, which cannot raise the maximum): the
        // block is unrolled, so the lane-to-lane hand-over needs no register copies and no loop control
        const int steps = tMax > 0 ? (tMax + G - 1 + 15) & ~15 : 0;
        const int laneRow = lane & 15;                             // position in the DPP row (both rows of a 32-lane group fetch the same residues)
        const int lastA = max(tLenA - 1, 0), lastB = max(tLenB - 1, 0);
        const int64_t baseA = (int64_t) jobA.t_start, baseB = (int64_t) jobB.t_start, stepA = jobA.t_step, stepB = jobB.t_step;
        const auto fetch = [&](int col) -> uint32_t {
            const uint32_t ra = col < tLenA ? (uint32_t) L.t_res[baseA + (int64_t) min(col, lastA) * stepA] : 21u;
            const uint32_t rb = col < tLenB ? (uint32_t) L.t_res[baseB + (int64_t) min(col, lastB) * stepB] : 21u;
            return ra * ROWB | (rb * ROWB) << 16;
        };
        const char *profLane = reinterpret_cast<const char *>(prof + lane * RP);
        // the residues of a block of 16 steps are fetched three blocks ahead: the loads have ~3 x 16 steps to land, which also covers the
        // memory latency of a chip whose HBM the prefilter of the other stream keeps saturated
        uint32_t tq0 = fetch(laneRow), tq1 = fetch(16 + laneRow), tq2 = fetch(32 + laneRow);
        for (int s0 = 0; s0 < steps; s0 += 16) {
            uint32_t tcur = tq0;
            tq0 = tq1; tq1 = tq2;
            tq2 = fetch(s0 + 48 + laneRow);
            const uint32_t bestIn = pk_bits(best);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t top = tcur;
                tcur = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) tcur, 0x12F /* row_ror:15 */, 0xf, 0xf, false);
                const pk16 hup = pk_from(shift_up_zero<G>(outH, lane));
                pk16 F = pk_from(shift_up_zero<G>(outF, lane));
                const uint32_t tres = shift_up<G>(top, outRes, lane);
                const int16_t *pa = reinterpret_cast<const int16_t *>(profLane + (tres & 0xFFFFu)), *pb = reinterpret_cast<const int16_t *>(profLane + (tres >> 16));
                uint32_t wa[RP / 2], wb[RP / 2];                   // R int16 scores per target, two per dword
                if constexpr (RP % 4 == 0) {
                    // a lane's RP scores are RP * 2 bytes at a multiple of 8: read as 64-bit words.  As dwords (ds_read2_b32) the lanes of a group sit on
                    // every second bank and two groups share a cycle: 3.0 conflict cycles per active LDS cycle in the 48- / 64-row tiles
                    // (profiles/r06_contention.txt)
                    const uint64_t *qa = reinterpret_cast<const uint64_t *>(__builtin_assume_aligned(pa, 8)), *qb = reinterpret_cast<const uint64_t *>(__builtin_assume_aligned(pb, 8));
#pragma unroll
                    for (int k = 0; k < RP / 4; k++) {
                        const uint64_t xa = qa[k], xb = qb[k];
                        wa[2 * k] = (uint32_t) xa; wa[2 * k + 1] = (uint32_t) (xa >> 32); wb[2 * k] = (uint32_t) xb; wb[2 * k + 1] = (uint32_t) (xb >> 32);
                    }
                } else {
#pragma unroll
                for (int k = 0; k < RP / 2; k++) { wa[k] = reinterpret_cast<const uint32_t *>(pa)[k]; wb[k] = reinterpret_cast<const uint32_t *>(pb)[k]; }
                }
                pk16 dsave = hupPrev;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    // (score of target A, score of target B) for row r: one byte permute
                    const pk16 sc = pk_from(__builtin_amdgcn_perm(wb[r / 2], wa[r / 2], (r & 1) ? 0x07060302u : 0x05040100u));
                    const pk16 d = SAT ? __builtin_elementwise_add_sat(dsave, sc) : dsave + sc;
                    dsave = H[r];
                    const pk16 h = pk_max(pk_max(d, E[r]), F);         // E, F >= 0 (clamped subtractions) keep H non-negative
                    best = pk_max(best, h);
                    const pk16 ho = pk_subs0(h, go2);
                    E[r] = pk_max(pk_subs0(E[r], ge2), ho);
                    F = pk_max(pk_subs0(F, ge2), ho);
                    H[r] = h;
                }
                hupPrev = hup;
                outH = pk_bits(H[R - 1]);
                outF = pk_bits(F);
                outRes = tres;
            }
            const uint32_t rose = pk_bits(best) ^ bestIn;
            if (rose & 0xFFFFu) roseA = s0;
            if (rose >> 16) roseB = s0;
        }
        // group reduction of the two maxima
        uint32_t b = pk_bits(best);
#pragma unroll
        for (int m = G / 2; m >= 1; m >>= 1) b = pk_bits(pk_max(pk_from(b), pk_from((uint32_t) __shfl_xor((int) b, m, G))));
        // ... and a bound for the position pass: a lane whose rows hold the maximum reached it in the block of steps in which its running
        // maximum last rose, i.e. in a column <= that block's last step - lane; the first column that reaches the maximum (what the position
        // pass reports) is not beyond the smallest such bound, so the position pass may stop there (the columns behind it cannot change its answer)
        uint32_t cA = (pk_bits(best) & 0xFFFFu) == (b & 0xFFFFu) ? (uint32_t) max(roseA + 15 - lane, 0) : 0x7FFFFFFFu;
        uint32_t cB = (pk_bits(best) >> 16) == (b >> 16) ? (uint32_t) max(roseB + 15 - lane, 0) : 0x7FFFFFFFu;
#pragma unroll
        for (int m = G / 2; m >= 1; m >>= 1) {
            cA = min(cA, (uint32_t) __shfl_xor((int) cA, m, G));
            cB = min(cB, (uint32_t) __shfl_xor((int) cB, m, G));
        }
        if (lane == 0) {
            SwOut o;
            o.end_row = -1; o.pad = 0;                              // end_col: not the end cell's column but a bound for it (gate_emit_kernel cuts the position job there)
            if (haveA) { o.score = (int32_t) (int16_t) (b & 0xFFFFu); o.end_col = (int32_t) min(cA, (uint32_t) lastA); L.out[jobA.slot] = o; }
            if (haveB) { o.score = (int32_t) (int16_t) (b >> 16); o.end_col = (int32_t) min(cB, (uint32_t) lastB); L.out[jobB.slot] = o; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }   // rounds of the wave
    }
}

// ---------------------------------------------------------------------------------------------
// End cell of a KNOWN maximum, packed int16: the position pass (the ~9 % of the pairs that pass the e-value gate: their score is
// known from the score pass) and the reverse pass (reversed prefixes ending at the forward end cell: same score).  The reference
// reports the first column in which the maximum is reached and the smallest row in it; with the value S known in advance that is
// the first step at which a lane's running maximum equals S -- no per-cell position key, so the DP runs at the score pass's price.
// Eight INDEPENDENT DPs per wave: each 16-lane group runs two jobs in the packed halves, every job with its own query profile
// (the survivors of a query are too few to fill a wave with one query), any direction (q_step / t_step = -1 for the reverse pass).
// For tiles of at most 64 rows (no saturation, no row tiles); larger tiles keep the int32 kernel.
template <int R, int RP>
__global__ __launch_bounds__(64) void swq_kernel(SwLaunch L) {
    constexpr int G = 16, ROWS = G * RP, NDP = 8;
    static_assert(RP >= R && RP % 2 == 0 && G * R * 127 <= 32767, "small tiles only");
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int16_t *prof = reinterpret_cast<int16_t *>(smem);            // prof[dp][t][slot]
    constexpr int PROF1 = 22 * ROWS;                              // int16 entries of one profile
    int8_t *sMat = smem + (size_t) NDP * PROF1 * sizeof(int16_t);
    uint8_t *sQ = reinterpret_cast<uint8_t *>(sMat + 448);         // [dp][slot]
    int8_t *sB = reinterpret_cast<int8_t *>(sQ + NDP * ROWS);
    for (int k = (int) threadIdx.x; k < 441; k += 64) sMat[k] = L.mat[k];
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const pk16 go2 = pk_splat(L.gap_open), ge2 = pk_splat(L.gap_extend), zero2 = pk_splat(0);
    const uint64_t nUnits = (L.n_jobs + NDP - 1) / NDP;
    for (;;) {
        uint32_t u = 0;
        if (threadIdx.x == 0) u = atomicAdd(L.work_counter, 1u);
        u = (uint32_t) __builtin_amdgcn_readfirstlane((int) u);
        if ((uint64_t) u >= nUnits) break;
        const uint64_t j0 = (uint64_t) u * NDP;
        // every lane's two jobs; the lanes 0..7 of the wave also describe job j0 + lane for the profile staging below
        const uint64_t ja = j0 + 2 * grp, jb = ja + 1;
        const bool haveA = ja < L.n_jobs, haveB = jb < L.n_jobs;
        const SwJob jobA = L.jobs[L.order[haveA ? ja : j0]], jobB = L.jobs[L.order[haveB ? jb : j0]];
        const int tLenA = haveA ? (int) jobA.t_len : 0, tLenB = haveB ? (int) jobB.t_len : 0;
        const uint32_t SA = haveA ? (uint32_t) L.known_score[jobA.slot] & 0xFFFFu : 0xFFFFu, SB = haveB ? (uint32_t) L.known_score[jobB.slot] & 0xFFFFu : 0xFFFFu;
        const uint32_t S2 = SA | (SB << 16);
        // ---- the eight query profiles: residues + bias staged in LDS, then prof[dp][t][slot] = mat[t][q] + bias ----
        for (int i = (int) threadIdx.x; i < NDP * ROWS; i += 64) {
            const int dp = i / ROWS, slot = i - dp * ROWS;
            const uint64_t j = j0 + (uint64_t) dp;
            uint8_t qc = 255; int8_t bc = 0;
            if (j < L.n_jobs) {
                const SwJob jj = L.jobs[L.order[j]];
                const int row = RP == R ? slot : (slot / RP) * R + slot % RP;
                if (row < (int) jj.q_len && (RP == R || slot % RP < R)) {
                    const int64_t qi = (int64_t) jj.q_start + (int64_t) row * jj.q_step;
                    qc = L.q_res[qi]; bc = L.q_bias8[qi];
                }
            }
            sQ[i] = qc; sB[i] = bc;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = (int) threadIdx.x; i < NDP * PROF1; i += 64) {
            const int dp = i / PROF1, rem = i - dp * PROF1, t = rem / ROWS, slot = rem - t * ROWS;
            const uint32_t qc = sQ[dp * ROWS + slot];
            prof[i] = (t < 21 && qc != 255u) ? (int16_t) ((int) sMat[t * 21 + (int) qc] + (int) sB[dp * ROWS + slot]) : (int16_t) 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        pk16 H[R], E[R];
#pragma unroll
        for (int r = 0; r < R; r++) { H[r] = zero2; E[r] = zero2; }
        pk16 best = zero2, hupPrev = zero2;
        constexpr uint32_t ROWB = (uint32_t) ROWS * 2u, NOCOL = 21u * ROWB | (21u * ROWB) << 16;
        uint32_t outH = 0, outF = 0, outRes = NOCOL;
        const int tMax = max(tLenA, tLenB);
        const int steps = tMax > 0 ? (tMax + G - 1 + 15) & ~15 : 0;
        const int laneRow = lane & 15;
        const int lastA = max(tLenA - 1, 0), lastB = max(tLenB - 1, 0);
        const int64_t baseA = (int64_t) jobA.t_start, baseB = (int64_t) jobB.t_start, stepA = jobA.t_step, stepB = jobB.t_step;
        const auto fetch = [&](int col) -> uint32_t {
            const uint32_t ra = col < tLenA ? (uint32_t) L.t_res[baseA + (int64_t) min(col, lastA) * stepA] : 21u;
            const uint32_t rb = col < tLenB ? (uint32_t) L.t_res[baseB + (int64_t) min(col, lastB) * stepB] : 21u;
            return ra * ROWB | (rb * ROWB) << 16;
        };
        const char *profA = reinterpret_cast<const char *>(prof + (size_t) (2 * grp) * PROF1 + lane * RP);
        const char *profB = reinterpret_cast<const char *>(prof + (size_t) (2 * grp + 1) * PROF1 + lane * RP);
        uint32_t foundA = 0xFFFFFFFFu, foundB = 0xFFFFFFFFu;     // column << 12 | row of the lane's first cell at the known score
        uint32_t closed2 = 0u;                                     // 0x0001 in the halves that have found theirs (keeps them out of the test below)
        uint32_t tq0 = fetch(laneRow), tq1 = fetch(16 + laneRow), tq2 = fetch(32 + laneRow);
        for (int s0 = 0; s0 < steps; s0 += 16) {
            uint32_t tcur = tq0;
            tq0 = tq1; tq1 = tq2;
            tq2 = fetch(s0 + 48 + laneRow);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t top = tcur;
                tcur = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) tcur, 0x12F /* row_ror:15 */, 0xf, 0xf, false);
                const pk16 hup = pk_from(shift_up_zero<G>(outH, lane));
                pk16 F = pk_from(shift_up_zero<G>(outF, lane));
                const uint32_t tres = shift_up<G>(top, outRes, lane);
                const int16_t *pa = reinterpret_cast<const int16_t *>(profA + (tres & 0xFFFFu)), *pb = reinterpret_cast<const int16_t *>(profB + (tres >> 16));
                uint32_t wa[RP / 2], wb[RP / 2];
#pragma unroll
                for (int kk = 0; kk < RP / 2; kk++) { wa[kk] = reinterpret_cast<const uint32_t *>(pa)[kk]; wb[kk] = reinterpret_cast<const uint32_t *>(pb)[kk]; }
                pk16 dsave = hupPrev;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const pk16 sc = pk_from(__builtin_amdgcn_perm(wb[r / 2], wa[r / 2], (r & 1) ? 0x07060302u : 0x05040100u));
                    const pk16 d = dsave + sc;
                    dsave = H[r];
                    const pk16 h = pk_max(pk_max(d, E[r]), F);
                    best = pk_max(best, h);
                    const pk16 ho = pk_subs0(h, go2);
                    E[r] = pk_max(pk_subs0(E[r], ge2), ho);
                    F = pk_max(pk_subs0(F, ge2), ho);
                    H[r] = h;
                }
                hupPrev = hup;
                outH = pk_bits(H[R - 1]);
                outF = pk_bits(F);
                outRes = tres;
                // a half of (best ^ S2) is zero when that DP's running maximum has reached its known score; zero-half test
                // (x - 0x00010001) & ~x & 0x80008000 -- it can flag the high half falsely when the low half is zero, so the branch re-checks
                const uint32_t x = (pk_bits(best) ^ S2) | closed2;
                const uint32_t hit = (x - 0x00010001u) & ~x & 0x80008000u;
                if (__builtin_amdgcn_ballot_w64(hit != 0u) != 0ull) {                  // rare: once per DP and lane (plus the odd false alarm)
                    const int c = s0 + k - lane;                                       // this lane's column at this step
                    if ((x & 0xFFFFu) == 0u) {
                        int row = 0;
#pragma unroll
                        for (int r = R - 1; r >= 0; r--) if ((pk_bits(H[r]) & 0xFFFFu) == SA) row = r;
                        foundA = ((uint32_t) c << 12) | (uint32_t) (lane * R + row);
                        closed2 |= 0x1u;
                    }
                    if ((x >> 16) == 0u) {
                        int row = 0;
#pragma unroll
                        for (int r = R - 1; r >= 0; r--) if ((pk_bits(H[r]) >> 16) == SB) row = r;
                        foundB = ((uint32_t) c << 12) | (uint32_t) (lane * R + row);
                        closed2 |= 0x10000u;
                    }
                }
            }
        }
        // first column, then smallest row: the minimum of the lanes' findings
#pragma unroll
        for (int m = G / 2; m >= 1; m >>= 1) {
            foundA = min(foundA, (uint32_t) __shfl_xor((int) foundA, m, G));
            foundB = min(foundB, (uint32_t) __shfl_xor((int) foundB, m, G));
        }
        if (lane == 0) {
            SwOut o;
            o.pad = 0;
            if (haveA) { const bool ok = foundA != 0xFFFFFFFFu; o.score = ok ? (int32_t) SA : 0; o.end_col = ok ? (int32_t) (foundA >> 12) : -1; o.end_row = ok ? (int32_t) (foundA & 0xFFFu) : -1; L.out[jobA.slot] = o; }
            if (haveB) { const bool ok = foundB != 0xFFFFFFFFu; o.score = ok ? (int32_t) SB : 0; o.end_col = ok ? (int32_t) (foundB >> 12) : -1; o.end_row = ok ? (int32_t) (foundB & 0xFFFu) : -1; L.out[jobB.slot] = o; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

hipError_t launch_sw_known(const SwLaunch &L, int cfg, hipStream_t stream) {
    if (!L.order || !L.work_counter || !L.known_score || cfg < 0 || cfg >= SW_NCFG || !sw_cfg_known(cfg)) return hipErrorInvalidValue;
    if (L.n_jobs == 0) return hipSuccess;
    const uint64_t nUnits = (L.n_jobs + 7) / 8;
    const uint64_t grid = std::min<uint64_t>(nUnits, L.persistent_blocks ? L.persistent_blocks : nUnits);
    const int rows = sw_cfg_rows(cfg);
    const size_t prows = rows == 48 ? 64 : rows;
    const size_t lds = (size_t) 8 * 22 * prows * sizeof(int16_t) + 448 + 2 * 8 * prows;
    switch (rows) {
        case 32: hipLaunchKernelGGL((swq_kernel<2, 2>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 48: hipLaunchKernelGGL((swq_kernel<3, 4>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 64: hipLaunchKernelGGL((swq_kernel<4, 4>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// score-only forward launch for the sw_cfg_packed configurations; wave_start must cut the job list into waves of at most
// sw_cfg_jobs_per_wave = 2 * 64/G jobs of one query
hipError_t launch_sw_score(const SwLaunch &L, int cfg, hipStream_t stream) {
    if (!L.wave_start || !L.work_counter || !L.order || cfg < 0 || cfg >= SW_NCFG || !sw_cfg_packed(cfg)) return hipErrorInvalidValue;
    if (L.n_waves == 0) return hipSuccess;
    const uint64_t grid = L.units_per_block ? (L.n_waves + L.units_per_block - 1) / L.units_per_block
                                            : std::min<uint64_t>(L.n_waves, L.persistent_blocks ? L.persistent_blocks : L.n_waves);
    const int rows = sw_cfg_rows(cfg);
    const size_t prows = rows == 48 ? 64 : rows;
    const size_t lds = (size_t) 22 * prows * sizeof(int16_t) + 448 + 2 * prows;     // profile + matrix + query residues / bias
    if (L.t_max_rows) {
        // profile queries on the large tiles: the waves whose fragments fit t_max_rows rows go through the transposed kernels (counters of their own), the
        // rest through the classic one below
        if (!L.work_counter_t || !L.q_prof || !L.narrow || L.t_max_rows > SWT_MAX_ROWS) return hipErrorInvalidValue;
        hipLaunchKernelGGL(swt_kernel<0>, dim3((unsigned) grid), dim3(64), swt_image_bytes(rows), stream, L);
        if (L.t_max_rows > 64u) hipLaunchKernelGGL(swt_kernel<1>, dim3((unsigned) grid), dim3(64), swt_image_bytes(rows), stream, L);
        if (L.t_max_rows > 128u) hipLaunchKernelGGL(swt_kernel<2>, dim3((unsigned) grid), dim3(64), swt_image_bytes(rows), stream, L);
    }
    switch (rows) {
        case 32: hipLaunchKernelGGL((swp_kernel<2>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 48: hipLaunchKernelGGL((swp_kernel<3, 4>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 64: hipLaunchKernelGGL((swp_kernel<4>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 96: hipLaunchKernelGGL((swp_kernel<6>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 128: hipLaunchKernelGGL((swp_kernel<8>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 192: hipLaunchKernelGGL((swp_kernel<12>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 256: hipLaunchKernelGGL((swp_kernel<16>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 384: if (L.narrow) hipLaunchKernelGGL((swp_kernel<24, 24, 16>), dim3((unsigned) grid), dim3(64), lds, stream, L);
                  else hipLaunchKernelGGL((swp_kernel<12, 12, 32>), dim3((unsigned) grid), dim3(64), lds, stream, L);
                  break;
        case 512: hipLaunchKernelGGL((swp_kernel<16, 16, 32>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;
        case 768: hipLaunchKernelGGL((swp_kernel<24, 24, 32>), dim3((unsigned) grid), dim3(64), lds, stream, L); break;   // (profile queries: 513 .. 768 columns)
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// transposed position pass (profile queries, forward jobs whose score is known and whose target is at most 256 residues; rows = the tile of the query's
// length: sizes the LDS image): L.order / L.n_jobs = the jobs, a query's jobs next to each other
hipError_t launch_sw_tpos(const SwLaunch &L, int rows, uint32_t blocksPerClass, hipStream_t stream) {
    if (!L.q_prof || !L.order || !L.known_score || rows <= 0) return hipErrorInvalidValue;
    if (L.n_jobs == 0) return hipSuccess;
    const unsigned grid = (unsigned) std::min<uint64_t>((L.n_jobs + 7) / 8, std::max<uint32_t>(1u, blocksPerClass));
    hipLaunchKernelGGL(swtp_kernel<0>, dim3(grid), dim3(64), swt_image_bytes(rows), stream, L);
    hipLaunchKernelGGL(swtp_kernel<1>, dim3(grid), dim3(64), swt_image_bytes(rows), stream, L);
    hipLaunchKernelGGL(swtp_kernel<2>, dim3(grid), dim3(64), swt_image_bytes(rows), stream, L);
    return hipGetLastError();
}

template <int G, int R, int BLOCK>
static hipError_t launch_one(const SwLaunch &L, hipStream_t stream) {
    if (L.wave_start) {                                     // one wave per workgroup, one profile per wave
        const size_t lds = (size_t) 24 * G * R;
        if (L.n_waves == 0) return hipSuccess;
        const uint64_t grid = L.units_per_block ? (L.n_waves + L.units_per_block - 1) / L.units_per_block
                                                : std::min<uint64_t>(L.n_waves, L.persistent_blocks ? L.persistent_blocks : L.n_waves);
        hipLaunchKernelGGL((sw_kernel<G, R, 64, true>), dim3((unsigned) grid), dim3(64), lds, stream, L);
        return hipGetLastError();
    }
    constexpr int GPB = BLOCK / G;
    const size_t lds = (size_t) GPB * 24 * G * R;
    const uint64_t blocks = (L.n_jobs + GPB - 1) / GPB;
    if (blocks == 0) return hipSuccess;
    if (lds > 48 * 1024) {                               // once per kernel (several alignment workers launch from their own threads)
        static std::once_flag once;
        static hipError_t attrErr = hipSuccess;
        std::call_once(once, [&] { attrErr = hipFuncSetAttribute(reinterpret_cast<const void *>(&sw_kernel<G, R, BLOCK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); });
        if (attrErr != hipSuccess) return attrErr;
    }
    hipLaunchKernelGGL((sw_kernel<G, R, BLOCK, false>), dim3((unsigned) blocks), dim3(BLOCK), lds, stream, L);
    return hipGetLastError();
}

hipError_t launch_sw(const SwLaunch &L, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= SW_NCFG) return hipErrorInvalidValue;
    switch (sw_cfg_rows(cfg)) {
#ifndef MK_SW_POS_BLOCK_S
#define MK_SW_POS_BLOCK_S 256
#define MK_SW_POS_BLOCK_L 128
#endif
        case 32: return launch_one<16, 2, MK_SW_POS_BLOCK_S>(L, stream);
        case 48:                                              // tiles the int32 kernel has no lane shape for run in the next one, padded
        case 64: return launch_one<16, 4, MK_SW_POS_BLOCK_S>(L, stream);
        case 96:
        case 128: return launch_one<16, 8, MK_SW_POS_BLOCK_S>(L, stream);
        case 192:
        case 256: return launch_one<16, 16, MK_SW_POS_BLOCK_L>(L, stream);
        // (Measured in round 5 and not kept: half the lanes per DP and twice the rows per lane for profile queries against ~40-column fragments --
        //  <16,24> / <16,32> / <32,24>: 1.3-1.5 x fewer lane-instructions per job on paper, 152-192 VGPRs in fact, and the config-4 pass 1.20-1.22 s against
        //  1.13-1.15 s (position pass 1 620 against 1 200 ms of kernel time): occupancy, not the ramp, is what these kernels live on.)
        case 384: return launch_one<32, 12, MK_SW_POS_BLOCK_L>(L, stream);
        case 512: return launch_one<32, 16, MK_SW_POS_BLOCK_L>(L, stream);
        case 768: return launch_one<64, 12, MK_SW_POS_BLOCK_L>(L, stream);
        case 1024: return launch_one<64, 16, MK_SW_POS_BLOCK_L>(L, stream);
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// Ungapped diagonal scoring: one lane per (query, target, diagonal) candidate; exact (unclamped)
// maximum of the prefix-reset running sum along the diagonal.  The 21x21 matrix sits in LDS; the
// per-position int8 correction is UngappedAlignment::createProfile's aaCorrectionScore.
__global__ __launch_bounds__(256) void ungapped_kernel(UngappedLaunch L) {
    __shared__ int8_t smat[21 * 21 + 3];
    for (int i = threadIdx.x; i < 21 * 21; i += blockDim.x) smat[i] = L.mat[i];
    __syncthreads();
    const uint64_t id = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= L.n_jobs) return;
    const UngappedJob j = L.jobs[id];
    const int best = ungapped_score(smat, L.q_res + j.q_start, L.q_corr + j.q_start, j.q_len, L.t_masked + j.t_start, j.t_len, j.diagonal & 0xFFFFu);
    L.out[id] = best;
}

hipError_t launch_ungapped(const UngappedLaunch &L, hipStream_t stream) {
    if (L.n_jobs == 0) return hipSuccess;
    const uint64_t blocks = (L.n_jobs + 255) / 256;
    hipLaunchKernelGGL(ungapped_kernel, dim3((unsigned) blocks), dim3(256), 0, stream, L);
    return hipGetLastError();
}

}  // namespace mk
