// metaeuk_amd/csrc/mk_kmer7.hip -- see mk_kmer7.hpp.  COUNT: one lane per k-mer start (the innermost row needs no walk: its
// per-row cumulative score histogram gives the number of partners in one lookup).  FILL: one wave per start -- the two 2-mer loops are
// wave-uniform, the lanes share the partners of the 3-mer row, so the list is written in coalesced runs.
#include "mk_kmer7.hpp"

namespace mk {

namespace {

constexpr int WAVE = 64;
constexpr int N2 = 400, N3 = 8000;

struct Start7 { uint32_t idx0, idx1, idx2; };
__device__ __forceinline__ Start7 start_of(const uint8_t *r) {      // window letters at 0,1,3,5,6,9,10
    Start7 s;
    s.idx0 = r[0] + 20u * r[1];
    s.idx1 = r[3] + 20u * r[5];
    s.idx2 = r[6] + 20u * r[9] + 400u * r[10];
    return s;
}
__device__ __forceinline__ uint32_t partners3(const Kmer7Tables &T, uint32_t idx2, int cutoff) {   // entries of row idx2 with score >= cutoff
    const int xb = cutoff - T.hist_lo;
    return xb <= 0 ? (uint32_t) N3 : (xb >= T.hist_range ? 0u : (uint32_t) T.cum3[(size_t) idx2 * T.hist_range + xb]);
}

__global__ __launch_bounds__(256) void kmer7_count_kernel(Kmer7Tables T, const uint8_t *res, const int16_t *kthr, uint64_t posBegin, uint64_t posEnd, uint32_t *counts) {
    const uint64_t p = posBegin + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= posEnd) return;
    const int thr = (int) kthr[p];
    if (thr < 0) { counts[p - posBegin] = 0; return; }
    const Start7 S = start_of(res + p);
    const int16_t *s0 = T.score2 + (size_t) S.idx0 * N2, *s1 = T.score2 + (size_t) S.idx1 * N2;
    const int rest1 = (int) T.score3[(size_t) S.idx2 * N3], rest0 = (int) s1[0] + rest1;
    uint64_t n = 0;
    for (int a = 0; a < N2; a++) {
        const int sa = (int) s0[a];
        if (sa < thr - rest0) break;
        for (int b = 0; b < N2; b++) {
            const int sb = (int) s1[b];
            if (sb < thr - sa - rest1) break;
            n += partners3(T, S.idx2, thr - sa - sb);
        }
    }
    counts[p - posBegin] = (uint32_t) min(n, (uint64_t) 0xFFFFFFFFull);
}

__global__ __launch_bounds__(256) void kmer7_fill_kernel(Kmer7Tables T, const uint8_t *res, const int16_t *kthr, uint64_t posBegin, uint64_t posEnd,
                                                         const uint64_t *listOff, uint32_t *list) {
    const int lane = threadIdx.x & (WAVE - 1);
    const uint64_t p = posBegin + ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    if (p >= posEnd) return;
    const int thr = (int) kthr[p];
    if (thr < 0) return;
    const Start7 S = start_of(res + p);
    const int16_t *s0 = T.score2 + (size_t) S.idx0 * N2, *s1 = T.score2 + (size_t) S.idx1 * N2;
    const uint16_t *i0 = T.index2 + (size_t) S.idx0 * N2, *i1 = T.index2 + (size_t) S.idx1 * N2, *i2 = T.index3 + (size_t) S.idx2 * N3;
    const int rest1 = (int) T.score3[(size_t) S.idx2 * N3], rest0 = (int) s1[0] + rest1;
    uint32_t *out = list + listOff[p - posBegin];
    uint64_t n = 0;
    for (int a = 0; a < N2; a++) {
        const int sa = (int) s0[a];
        if (sa < thr - rest0) break;
        const uint32_t ca = (uint32_t) i0[a];
        for (int b = 0; b < N2; b++) {
            const int sb = (int) s1[b];
            if (sb < thr - sa - rest1) break;
            const uint32_t cab = ca + 400u * (uint32_t) i1[b];
            const uint32_t nc = partners3(T, S.idx2, thr - sa - sb);
            for (uint32_t c = (uint32_t) lane; c < nc; c += WAVE) out[n + c] = cab + 160000u * (uint32_t) T.num3[i2[c]];
            n += nc;
        }
    }
}

}  // namespace

hipError_t launch_kmer7_count(const Kmer7Tables &T, const uint8_t *dRes, const int16_t *dKthr, uint64_t posBegin, uint64_t posEnd, uint32_t *dCounts, hipStream_t stream) {
    if (posEnd <= posBegin) return hipSuccess;
    hipLaunchKernelGGL(kmer7_count_kernel, dim3((unsigned) ((posEnd - posBegin + 255) / 256)), dim3(256), 0, stream, T, dRes, dKthr, posBegin, posEnd, dCounts);
    return hipGetLastError();
}

hipError_t launch_kmer7_fill(const Kmer7Tables &T, const uint8_t *dRes, const int16_t *dKthr, uint64_t posBegin, uint64_t posEnd,
                             const uint64_t *dListOff, uint32_t *dList, hipStream_t stream) {
    if (posEnd <= posBegin) return hipSuccess;
    hipLaunchKernelGGL(kmer7_fill_kernel, dim3((unsigned) ((posEnd - posBegin + 3) / 4)), dim3(256), 0, stream, T, dRes, dKthr, posBegin, posEnd, dListOff, dList);
    return hipGetLastError();
}

}  // namespace mk
