// metaeuk_amd/csrc/mk_derive.hip -- per-residue query-side inputs of the two stages, computed on the GPU
// right after the batch upload (no host pass over the residues):
//   * composition bias  SubstitutionMatrix::calcLocalAaBiasCorrection (M/src/commons/SubstitutionMatrix.cpp:79-109)
//     once with the seed matrix (VTML80 x8) and once with the alignment matrix (BLOSUM62 x2);
//   * k-mer threshold per k-mer start: QueryMatcher::match (M/src/prefiltering/QueryMatcher.cpp:225-244);
//   * int8 diagonal correction: UngappedAlignment::createProfile (M/src/prefiltering/UngappedAlignment.cpp:391-396);
//   * int8 SW composition bias: SmithWaterman::ssw_init (M/src/alignment/StripedSmithWaterman.cpp:1228-1235).
// The float/double expression types of the reference are kept literally; the build uses -ffp-contract=off, the
// f64 division of gfx950 is correctly rounded, so every value is bit-identical to the host expressions
// (tests/test_gpu_parity.py::test_device_derive_matches_host compares all four arrays).
#include "mk_kernels.hpp"
#include "mk_host.hpp"

namespace mk {

namespace {

struct DeriveMats {
    short kmerSub[21][21]; short alnSub[21][21];
    double kmerPback[21]; double alnPback[21];
};

__device__ __forceinline__ float local_bias(const short (*sub)[21], const double *pback, const uint8_t *seq, int L, int i, float scale) {
    const int lo = max(0, i - 20), hi = min(L, i + 20);
    const short *row = sub[seq[i]];
    int sum = 0;
    for (int j = lo; j < hi; j++) sum += row[seq[j]];
    sum -= row[seq[i]];
    float d = (float) sum;
    d = (float) ((double) d / (-1.0 * (double) ((float) (hi - lo))));
    for (int a = 0; a < 21; a++) d = (float) ((double) d + pback[a] * (double) ((float) row[a]));
    return scale * d;
}

__global__ __launch_bounds__(256) void derive_kernel(const uint8_t *res, const uint64_t *off, uint32_t nq, uint64_t total, const DeriveMats *M,
                                                     int compBias, float scale, float *bias1, int8_t *corr, int8_t *sw8) {
    __shared__ DeriveMats sM;
    for (int k = threadIdx.x; k < (int) (sizeof(DeriveMats) / 4); k += blockDim.x) reinterpret_cast<uint32_t *>(&sM)[k] = reinterpret_cast<const uint32_t *>(M)[k];
    __syncthreads();
    const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    uint32_t lo = 0, hi = nq;                       // query of residue p
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= p) lo = mid; else hi = mid; }
    const uint64_t qs = off[lo];
    const int L = (int) (off[lo + 1] - qs), i = (int) (p - qs);
    float b1 = 0.0f, b2 = 0.0f;
    if (compBias) {
        b1 = local_bias(sM.kmerSub, sM.kmerPback, res + qs, L, i, scale);
        b2 = local_bias(sM.alnSub, sM.alnPback, res + qs, L, i, scale);
    }
    bias1[p] = b1;
    float c = b1;
    c = (float) ((c < 0.0) ? (double) (c / 4) - 0.5 : (double) (c / 4) + 0.5);
    corr[p] = (int8_t) (signed char) c;
    sw8[p] = (int8_t) ((b2 < 0.0) ? (double) b2 - 0.5 : (double) b2 + 0.5);
}

template <int K>
__global__ __launch_bounds__(256) void kthr_kernel(const uint8_t *res, const uint64_t *off, uint32_t nq, uint64_t total, const float *bias1,
                                                   int kmerThr, int16_t *kthr) {
    const uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    uint32_t lo = 0, hi = nq;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= p) lo = mid; else hi = mid; }
    const uint64_t qe = off[lo + 1];
    int16_t out = -1;
    constexpr int SPANK = K == 7 ? 11 : 10;         // spaced seeds 1101010011 / 11010110011 (Sequence.h:23,25)
    if (p + SPANK <= qe) {                          // Sequence::hasNextKmer: i + span <= L
        const int sp[7] = {0, 1, 3, 5, K == 7 ? 6 : 8, 9, 10};
        float acc = 0;
        bool hasX = false;
        for (int k = 0; k < K; k++) { acc += bias1[p + sp[k]]; hasX |= (res[p + sp[k]] == 20); }
        if (!hasX) {
            const short r = (short) ((acc < 0.0) ? (double) acc - 0.5 : (double) acc + 0.5);
            out = (int16_t) max(kmerThr - (int) r, 0);
        }
    }
    kthr[p] = out;
}

}  // namespace

hipError_t launch_derive(const uint8_t *dRes, const uint64_t *dOff, uint32_t nq, uint64_t total, const SubMat &kmerMat, const SubMat &alnMat,
                         int kmerThr, bool compBias, float scale, int16_t *dKthr, int8_t *dCorr, int8_t *dSw8, hipStream_t stream, int kmerSize) {
    if (total == 0) return hipSuccess;
    DeriveMats hm;
    for (int i = 0; i < 21; i++) {
        for (int j = 0; j < 21; j++) { hm.kmerSub[i][j] = kmerMat.sub[i][j]; hm.alnSub[i][j] = alnMat.sub[i][j]; }
        hm.kmerPback[i] = kmerMat.pback[i]; hm.alnPback[i] = alnMat.pback[i];
    }
    DeriveMats *dM = (DeriveMats *) dev_scratch("derive_mats", sizeof(DeriveMats));
    float *dBias = (float *) dev_scratch("derive_bias", total * sizeof(float));
    if (!dM || !dBias) return hipErrorOutOfMemory;
    hipError_t e = hipMemcpyAsync(dM, &hm, sizeof(hm), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);               // hm lives on this stack frame
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned) ((total + 255) / 256);
    hipLaunchKernelGGL(derive_kernel, dim3(blocks), dim3(256), 0, stream, dRes, dOff, nq, total, dM, compBias ? 1 : 0, scale, dBias, dCorr, dSw8);
    if (kmerSize == 7) hipLaunchKernelGGL(kthr_kernel<7>, dim3(blocks), dim3(256), 0, stream, dRes, dOff, nq, total, dBias, kmerThr, dKthr);
    else hipLaunchKernelGGL(kthr_kernel<6>, dim3(blocks), dim3(256), 0, stream, dRes, dOff, nq, total, dBias, kmerThr, dKthr);
    return hipGetLastError();
}

}  // namespace mk
