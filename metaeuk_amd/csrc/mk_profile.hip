// metaeuk_amd/csrc/mk_profile.hip -- profile queries on gfx950: the per-column arrays of a batch of profiles and the lists of
// their similar k-mers.  See mk_profile.hpp for the reference functions restated (Sequence::mapProfile, Util::rankedDescSort20,
// KmerGenerator's profile divide strategy).
//
// The k-mer lists are materialised in HBM (4 B per similar k-mer, ~170 per column at -s 4) and then walked by the same probe kernels
// as the sequence path: the reference's list ORDER matters downstream (arrival order of the index hits decides the double-diagonal
// rule), and for six one-column steps that order is simply the lexicographic order of the rank tuples -- a depth-first walk per
// k-mer start reproduces it with no intermediate lists.  One thread per start; the sorted columns of a workgroup's 256 + 9 positions
// are staged in LDS (10.6 KB), so the inner loops read LDS only.
#include "mk_profile.hpp"
#include "mk_enum.hpp"

namespace mk {

namespace {

constexpr int SPACED[6] = {0, 1, 3, 5, 8, 9};     // spaced seed 1101010011 (k = 6)
constexpr int SPAN = 10;
constexpr int SPACED7[7] = {0, 1, 3, 5, 6, 9, 10}; // spaced seed 11010110011 (k = 7, Sequence.h:25)
constexpr int SPAN7 = 11;
constexpr int SPAN_MAX = 11;

__device__ __forceinline__ uint32_t find_profile(const uint64_t *off, uint32_t n, uint64_t p) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= p) lo = mid; else hi = mid; }
    return lo;
}

// Util::rankedDescSort20 (M/src/commons/Util.cpp:88-114)
__device__ __forceinline__ void ranked_desc_sort20(int (&val)[20], int (&idx)[20]) {
#define MK_SWAP(x, y) { if (val[x] < val[y]) { const int t1 = val[x]; val[x] = val[y]; val[y] = t1; const int t2 = idx[x]; idx[x] = idx[y]; idx[y] = t2; } }
    MK_SWAP(0,16) MK_SWAP(1,17) MK_SWAP(2,18) MK_SWAP(3,19) MK_SWAP(4,12) MK_SWAP(5,13) MK_SWAP(6,14) MK_SWAP(7,15)
    MK_SWAP(0,8) MK_SWAP(1,9) MK_SWAP(2,10) MK_SWAP(3,11)
    MK_SWAP(8,16) MK_SWAP(9,17) MK_SWAP(10,18) MK_SWAP(11,19) MK_SWAP(0,4) MK_SWAP(1,5) MK_SWAP(2,6) MK_SWAP(3,7)
    MK_SWAP(8,12) MK_SWAP(9,13) MK_SWAP(10,14) MK_SWAP(11,15) MK_SWAP(4,16) MK_SWAP(5,17) MK_SWAP(6,18) MK_SWAP(7,19) MK_SWAP(0,2) MK_SWAP(1,3)
    MK_SWAP(4,8) MK_SWAP(5,9) MK_SWAP(6,10) MK_SWAP(7,11) MK_SWAP(12,16) MK_SWAP(13,17) MK_SWAP(14,18) MK_SWAP(15,19) MK_SWAP(0,1)
    MK_SWAP(4,6) MK_SWAP(5,7) MK_SWAP(8,10) MK_SWAP(9,11) MK_SWAP(12,14) MK_SWAP(13,15) MK_SWAP(16,18) MK_SWAP(17,19)
    MK_SWAP(2,16) MK_SWAP(3,17) MK_SWAP(6,12) MK_SWAP(7,13) MK_SWAP(18,19)
    MK_SWAP(2,8) MK_SWAP(3,9) MK_SWAP(10,16) MK_SWAP(11,17)
    MK_SWAP(2,4) MK_SWAP(3,5) MK_SWAP(6,8) MK_SWAP(7,9) MK_SWAP(10,12) MK_SWAP(11,13) MK_SWAP(14,16) MK_SWAP(15,17)
    MK_SWAP(2,3) MK_SWAP(4,5) MK_SWAP(6,7) MK_SWAP(8,9) MK_SWAP(10,11) MK_SWAP(12,13) MK_SWAP(14,15) MK_SWAP(16,17)
    MK_SWAP(1,16) MK_SWAP(3,18) MK_SWAP(5,12) MK_SWAP(7,14)
    MK_SWAP(1,8) MK_SWAP(3,10) MK_SWAP(9,16) MK_SWAP(11,18)
    MK_SWAP(1,4) MK_SWAP(3,6) MK_SWAP(5,8) MK_SWAP(7,10) MK_SWAP(9,12) MK_SWAP(11,14) MK_SWAP(13,16) MK_SWAP(15,18)
    MK_SWAP(1,2) MK_SWAP(3,4) MK_SWAP(5,6) MK_SWAP(7,8) MK_SWAP(9,10) MK_SWAP(11,12) MK_SWAP(13,14) MK_SWAP(15,16) MK_SWAP(17,18)
#undef MK_SWAP
}

__global__ __launch_bounds__(256) void profile_derive_kernel(const uint8_t *raw, const uint64_t *off, uint32_t n, uint64_t total, int kmerThr, int kmerSize,
                                                             uint8_t *letters, int8_t *sorted, int8_t *aln, int16_t *kthr) {
    const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const uint8_t *col = raw + p * PROFILE_COL_BYTES;
    int val[20], idx[20];
#pragma unroll
    for (int a = 0; a < 20; a++) { val[a] = (int) (int8_t) col[a]; idx[a] = a; }
    int8_t *al = aln + p * PROFILE_ALN_STRIDE;
#pragma unroll
    for (int a = 0; a < 20; a++) al[a] = (int8_t) (val[a] / 4);            // C division: towards zero (Sequence.cpp:272-276)
#pragma unroll
    for (int a = 20; a < PROFILE_ALN_STRIDE; a++) al[a] = 0;               // X scores 0 (:278-280); 21 = "no column"
    ranked_desc_sort20(val, idx);
    int8_t *so = sorted + p * PROFILE_SORTED_STRIDE;
#pragma unroll
    for (int a = 0; a < 20; a++) { so[a] = (int8_t) val[a]; so[20 + a] = (int8_t) idx[a]; }
    letters[p] = col[20];
    // k-mer start: Sequence::hasNextKmer (i + span <= L), kmerContainsX on the query letters
    const uint32_t q = find_profile(off, n, p);
    int16_t out = -1;
    if (p + (kmerSize == 7 ? SPAN7 : SPAN) <= off[q + 1]) {
        bool hasX = false;
        if (kmerSize == 7) {
#pragma unroll
            for (int k = 0; k < 7; k++) hasX |= raw[(p + SPACED7[k]) * PROFILE_COL_BYTES + 20] == 20;
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) hasX |= raw[(p + SPACED[k]) * PROFILE_COL_BYTES + 20] == 20;
        }
        if (!hasX) out = (int16_t) max(kmerThr, 0);
    }
    kthr[p] = out;
}

// the k-mer thresholds alone, from the query letters: a batch derived for one k-mer size that meets a database of the other
__global__ __launch_bounds__(256) void profile_kthr_kernel(const uint8_t *letters, const uint64_t *off, uint32_t n, uint64_t total, int kmerThr, int kmerSize, int16_t *kthr) {
    const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const uint32_t q = find_profile(off, n, p);
    int16_t out = -1;
    if (p + (kmerSize == 7 ? SPAN7 : SPAN) <= off[q + 1]) {
        bool hasX = false;
        for (int k = 0; k < kmerSize; k++) hasX |= letters[p + (kmerSize == 7 ? SPACED7[k] : SPACED[k])] == 20;
        if (!hasX) out = (int16_t) max(kmerThr, 0);
    }
    kthr[p] = out;
}

// k = 7: seven one-column steps, cells in the reference's numbering (sum of residue * 20^step)
template <bool FILL>
__global__ __launch_bounds__(256) void profile_kmer7_kernel(const int8_t *sorted, const int16_t *kthr, uint64_t posBegin, uint64_t posEnd,
                                                            uint32_t *counts, const uint64_t *listOff, uint32_t *list) {
    __shared__ int8_t sCol[(256 + SPAN_MAX) * PROFILE_SORTED_STRIDE];
    const uint64_t p0 = posBegin + (uint64_t) blockIdx.x * 256;
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(sorted + p0 * PROFILE_SORTED_STRIDE);
        uint32_t *dst = reinterpret_cast<uint32_t *>(sCol);
        for (int k = threadIdx.x; k < (256 + SPAN_MAX) * PROFILE_SORTED_STRIDE / 4; k += 256) dst[k] = src[k];
    }
    __syncthreads();
    const uint64_t p = p0 + threadIdx.x;
    if (p >= posEnd) return;
    const int thr = (int) kthr[p];
    if (thr < 0) { if (!FILL) counts[p - posBegin] = 0; return; }
    const int8_t *c[7];
#pragma unroll
    for (int i = 0; i < 7; i++) c[i] = sCol + (threadIdx.x + SPACED7[i]) * PROFILE_SORTED_STRIDE;
    int rest[7];
    rest[6] = 0;
#pragma unroll
    for (int i = 6; i >= 1; i--) rest[i - 1] = rest[i] + (int) c[i][0];
    uint32_t n = 0;
    uint32_t *out = FILL ? list + listOff[p - posBegin] : nullptr;
    for (int j0 = 0; j0 < 20; j0++) {
        const int s0 = (int) c[0][j0];
        if (s0 < thr - rest[0]) break;
        const uint32_t a0 = (uint32_t) c[0][20 + j0];
        for (int j1 = 0; j1 < 20; j1++) {
            const int v1 = (int) c[1][j1];
            if (v1 < thr - s0 - rest[1]) break;
            const int s1 = s0 + v1;
            const uint32_t a1 = a0 + 20u * (uint32_t) c[1][20 + j1];
            for (int j2 = 0; j2 < 20; j2++) {
                const int v2 = (int) c[2][j2];
                if (v2 < thr - s1 - rest[2]) break;
                const int s2 = s1 + v2;
                const uint32_t a2 = a1 + 400u * (uint32_t) c[2][20 + j2];
                for (int j3 = 0; j3 < 20; j3++) {
                    const int v3 = (int) c[3][j3];
                    if (v3 < thr - s2 - rest[3]) break;
                    const int s3 = s2 + v3;
                    const uint32_t a3 = a2 + 8000u * (uint32_t) c[3][20 + j3];
                    for (int j4 = 0; j4 < 20; j4++) {
                        const int v4 = (int) c[4][j4];
                        if (v4 < thr - s3 - rest[4]) break;
                        const int s4 = s3 + v4;
                        const uint32_t a4 = a3 + 160000u * (uint32_t) c[4][20 + j4];
                        for (int j5 = 0; j5 < 20; j5++) {
                            const int v5 = (int) c[5][j5];
                            if (v5 < thr - s4 - rest[5]) break;
                            const int s5 = s4 + v5;
                            const uint32_t a5 = a4 + 3200000u * (uint32_t) c[5][20 + j5];
                            for (int j6 = 0; j6 < 20; j6++) {
                                if ((int) c[6][j6] < thr - s5) break;
                                if (FILL) out[n] = a5 + 64000000u * (uint32_t) c[6][20 + j6];
                                n++;
                            }
                        }
                    }
                }
            }
        }
    }
    if (!FILL) counts[p - posBegin] = n;
}

// Depth-first walk of one k-mer start.  rest[i] = best score of the columns behind step i (possibleRest, KmerGenerator.cpp:124-126).
// emit(rank tuple's residues) is called in list order; COUNT only counts.
template <bool FILL>
__global__ __launch_bounds__(256) void profile_kmer_kernel(const int8_t *sorted, const int16_t *kthr, const uint16_t *addr3, uint64_t posBegin, uint64_t posEnd,
                                                           uint32_t *counts, const uint64_t *listOff, uint32_t *list) {
    __shared__ int8_t sCol[(256 + SPAN) * PROFILE_SORTED_STRIDE];
    const uint64_t p0 = posBegin + (uint64_t) blockIdx.x * 256;
    {   // the sorted columns of this workgroup's starts (the array is padded by more than SPAN columns behind the last profile)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(sorted + p0 * PROFILE_SORTED_STRIDE);
        uint32_t *dst = reinterpret_cast<uint32_t *>(sCol);
        for (int k = threadIdx.x; k < (256 + SPAN) * PROFILE_SORTED_STRIDE / 4; k += 256) dst[k] = src[k];
    }
    __syncthreads();
    const uint64_t p = p0 + threadIdx.x;
    if (p >= posEnd) return;
    const int thr = (int) kthr[p];
    if (thr < 0) { if (!FILL) counts[p - posBegin] = 0; return; }
    const int8_t *c[6];
#pragma unroll
    for (int i = 0; i < 6; i++) c[i] = sCol + (threadIdx.x + SPACED[i]) * PROFILE_SORTED_STRIDE;
    int rest[6];
    rest[5] = 0;
#pragma unroll
    for (int i = 5; i >= 1; i--) rest[i - 1] = rest[i] + (int) c[i][0];
    uint32_t n = 0;
    uint32_t *out = FILL ? list + listOff[p - posBegin] : nullptr;
    for (int j0 = 0; j0 < 20; j0++) {
        const int s0 = (int) c[0][j0];
        if (s0 < thr - rest[0]) break;
        const uint32_t a0 = (uint32_t) c[0][20 + j0];
        for (int j1 = 0; j1 < 20; j1++) {
            const int v1 = (int) c[1][j1];
            if (v1 < thr - s0 - rest[1]) break;
            const int s1 = s0 + v1;
            const uint32_t a1 = a0 + 20u * (uint32_t) c[1][20 + j1];
            for (int j2 = 0; j2 < 20; j2++) {
                const int v2 = (int) c[2][j2];
                if (v2 < thr - s1 - rest[2]) break;
                const int s2 = s1 + v2;
                const uint32_t first = FILL ? enumk::cell_first((uint32_t) addr3[a1 + 400u * (uint32_t) c[2][20 + j2]]) : 0u;
                for (int j3 = 0; j3 < 20; j3++) {
                    const int v3 = (int) c[3][j3];
                    if (v3 < thr - s2 - rest[3]) break;
                    const int s3 = s2 + v3;
                    const uint32_t a3 = (uint32_t) c[3][20 + j3];
                    for (int j4 = 0; j4 < 20; j4++) {
                        const int v4 = (int) c[4][j4];
                        if (v4 < thr - s3 - rest[4]) break;
                        const int s4 = s3 + v4;
                        const uint32_t a4 = a3 + 20u * (uint32_t) c[4][20 + j4];
                        for (int j5 = 0; j5 < 20; j5++) {
                            if ((int) c[5][j5] < thr - s4) break;
                            if (FILL) out[n] = first + enumk::cell_second((uint32_t) addr3[a4 + 400u * (uint32_t) c[5][20 + j5]]);
                            n++;
                        }
                    }
                }
            }
        }
    }
    if (!FILL) counts[p - posBegin] = n;
}

}  // namespace

hipError_t launch_profile_derive(const uint8_t *dRaw, const uint64_t *dOff, uint32_t nProfiles, uint64_t totalCols, int kmerThr, int kmerSize,
                                 uint8_t *dLetters, int8_t *dSorted, int8_t *dAln, int16_t *dKthr, hipStream_t stream) {
    if (totalCols == 0) return hipSuccess;
    hipLaunchKernelGGL(profile_derive_kernel, dim3((unsigned) ((totalCols + 255) / 256)), dim3(256), 0, stream, dRaw, dOff, nProfiles, totalCols, kmerThr, kmerSize,
                       dLetters, dSorted, dAln, dKthr);
    return hipGetLastError();
}

hipError_t launch_profile_kthr(const uint8_t *dLetters, const uint64_t *dOff, uint32_t nProfiles, uint64_t totalCols, int kmerThr, int kmerSize, int16_t *dKthr, hipStream_t stream) {
    if (totalCols == 0) return hipSuccess;
    hipLaunchKernelGGL(profile_kthr_kernel, dim3((unsigned) ((totalCols + 255) / 256)), dim3(256), 0, stream, dLetters, dOff, nProfiles, totalCols, kmerThr, kmerSize, dKthr);
    return hipGetLastError();
}

hipError_t launch_profile_kmer_count(const int8_t *dSorted, const int16_t *dKthr, uint64_t posBegin, uint64_t posEnd, uint32_t *dCounts, hipStream_t stream, int kmerSize) {
    if (posEnd <= posBegin) return hipSuccess;
    if (kmerSize == 7) {
        hipLaunchKernelGGL(profile_kmer7_kernel<false>, dim3((unsigned) ((posEnd - posBegin + 255) / 256)), dim3(256), 0, stream, dSorted, dKthr, posBegin, posEnd, dCounts,
                           (const uint64_t *) nullptr, (uint32_t *) nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(profile_kmer_kernel<false>, dim3((unsigned) ((posEnd - posBegin + 255) / 256)), dim3(256), 0, stream, dSorted, dKthr,
                       (const uint16_t *) nullptr, posBegin, posEnd, dCounts, (const uint64_t *) nullptr, (uint32_t *) nullptr);
    return hipGetLastError();
}

hipError_t launch_profile_kmer_fill(const int8_t *dSorted, const int16_t *dKthr, const uint16_t *dAddr3, uint64_t posBegin, uint64_t posEnd,
                                    const uint64_t *dListOff, uint32_t *dList, hipStream_t stream, int kmerSize) {
    if (posEnd <= posBegin) return hipSuccess;
    if (kmerSize == 7) {
        hipLaunchKernelGGL(profile_kmer7_kernel<true>, dim3((unsigned) ((posEnd - posBegin + 255) / 256)), dim3(256), 0, stream, dSorted, dKthr, posBegin, posEnd,
                           (uint32_t *) nullptr, dListOff, dList);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(profile_kmer_kernel<true>, dim3((unsigned) ((posEnd - posBegin + 255) / 256)), dim3(256), 0, stream, dSorted, dKthr, dAddr3,
                       posBegin, posEnd, (uint32_t *) nullptr, dListOff, dList);
    return hipGetLastError();
}

}  // namespace mk
