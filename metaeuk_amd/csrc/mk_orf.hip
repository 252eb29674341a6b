// metaeuk_amd/csrc/mk_orf.hip -- six-frame ORF fragments of a batch of contigs, translated, on the GPU
// (the producer of the hot path's queries).  Replaces, for `predictexons`' settings
// (orf-start-mode 1, both strands, all frames, contig start/end mode 2, genetic code 1):
//   Orf::setSequence / findAll / findForward    M/src/commons/Orf.cpp:118-345
//   the extractorfs loop, --translate branch    M/src/util/extractorfs.cpp:64-125
//   TranslateNucl::translate                    M/src/commons/TranslateNucl.h:488-503 (IUPAC-aware table: mk_host.cpp)
// Layout: contigs stay as the caller's ASCII in HBM; nothing is copied or reverse-complemented -- the minus strand is an
// index transform + complement lookup.  Three kernels:
//   orf_scan_kernel<false>   one lane per (contig, strand): the reference's three-frame state machine, counting only
//   orf_scan_kernel<true>    the same walk, writing one record per fragment at its scanned position -- fragments come out
//                            in the order the reference writes them (= the renumbered ORF ids)
//   orf_translate_kernel     one lane per amino acid: codon -> residue (ASCII, case preserved) and aa2num code
// The fragment codes stay in HBM and become the query batch of mk_search without a host round trip.
#include "mk_orf.hpp"
#include "mk_kernels.hpp"
#include <hipcub/hipcub.hpp>
#include <climits>
#include <cstring>

namespace mk {

namespace {

// Orf::iupacReverseComplementTable (Orf.cpp:48-52), '.' = not a nucleotide code
__constant__ char cComplement[256];
__constant__ uint8_t cBaseCode[256];      // TranslateNucl::sm_BaseToIdx
__constant__ uint8_t cAaCode[256];        // aa2num of the residue characters (mk::encode)

struct Strand {
    const char *seq; uint32_t len; bool minus;
    // Orf::setSequence: only 'u' becomes 't' (the second assignment of the reference wins); minus strand = complement of
    // the mirrored position, '.' -> 'N'; CHAR_MAX behind the end
    __device__ __forceinline__ char at(uint32_t i) const {
        if (i >= len) return (char) CHAR_MAX;
        char c = seq[minus ? len - 1 - i : i];
        if (c == 'u') c = 't';
        if (minus) { c = cComplement[(unsigned char) c]; if (c == '.') c = 'N'; }
        return c;
    }
};

__device__ __forceinline__ char upper_or_max(char c) { return c == (char) CHAR_MAX ? c : (char) (c & (unsigned char) ~0x20); }
__device__ __forceinline__ bool not_nucleotide(char c) { return c == 'N' || cComplement[(unsigned char) c] == '.'; }

template <bool WRITE>
__global__ __launch_bounds__(64) void orf_scan_kernel(OrfScanArgs A) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;       // contig * 2 + strand
    if (id >= 2 * A.n_contigs) return;
    const uint32_t contig = id >> 1;
    Strand S;
    S.seq = A.nucl + A.offsets[contig]; S.len = (uint32_t) (A.offsets[contig + 1] - A.offsets[contig]); S.minus = id & 1u;
    uint32_t nFrag = 0;
    uint64_t nAa = 0;
    uint64_t fragAt = 0, aaAt = 0;
    if (WRITE) { fragAt = A.frag_base[id]; aaAt = A.aa_base[id]; }
    if (S.len >= 3) {
        bool inside[3] = {true, true, true}, hasStart[3] = {false, false, false};
        uint32_t count[3] = {0, 0, 0}, from[3] = {0, 1, 2};
        uint64_t gaps[3] = {0, 0, 0};
        // sliding window of the upper-cased strand: c0 c1 c2 = codon at `position`, n0 n1 n2 = the frame's next codon
        for (uint32_t i = 0; i < S.len - 2; i += 3) {
            for (uint32_t position = i; position < i + 3; position++) {
                const char c0 = upper_or_max(S.at(position)), c1 = upper_or_max(S.at(position + 1)), c2 = upper_or_max(S.at(position + 2));
                const uint32_t frame = position % 3;
                const bool thisIncomplete = c0 == (char) CHAR_MAX || c1 == (char) CHAR_MAX || c2 == (char) CHAR_MAX;
                const bool nextIncomplete = S.at(position + 3) == (char) CHAR_MAX || S.at(position + 4) == (char) CHAR_MAX || S.at(position + 5) == (char) CHAR_MAX;
                const bool isLast = !thisIncomplete && nextIncomplete;
                if (!inside[frame]) {                                  // ANY_TO_STOP: a fragment starts right behind every stop
                    inside[frame] = true; hasStart[frame] = true; from[frame] = position; gaps[frame] = 0; count[frame] = 0;
                }
                const bool stop = c0 == 'T' && ((c1 == 'A' && (c2 == 'A' || c2 == 'G')) || (c1 == 'G' && c2 == 'A'));
                if (!stop) count[frame]++;
                if (not_nucleotide(c0) || not_nucleotide(c1) || not_nucleotide(c2)) gaps[frame]++;
                if (stop || isLast) {
                    inside[frame] = false;
                    if (count[frame] == 0 && stop) continue;
                    const uint32_t to = (isLast && !stop) ? position + 2 : position - 1;
                    if (gaps[frame] > A.max_gaps || count[frame] > A.max_length || count[frame] < A.min_length) continue;
                    const uint32_t naa = (to - from[frame] + 1) / 3;
                    if (WRITE) {
                        OrfRecord r;
                        r.contig = contig; r.s_from = from[frame]; r.n_aa = naa;
                        r.flags = (hasStart[frame] ? 0u : 1u) | (stop ? 0u : 2u) | (S.minus ? 4u : 0u);
                        A.records[fragAt + nFrag] = r;
                        A.aa_off[fragAt + nFrag] = aaAt + nAa;
                    }
                    nFrag++;
                    nAa += naa;
                }
            }
        }
    }
    if (!WRITE) { A.frag_count[id] = nFrag; A.aa_count[id] = nAa; }
}

__global__ __launch_bounds__(256) void orf_translate_kernel(OrfScanArgs A, uint64_t nFrag, uint64_t nAa, const char *table /* [4096] */,
                                                            char *aaAscii, uint8_t *aaCode) {
    __shared__ char sTable[4096];
    for (int k = threadIdx.x; k < 4096; k += blockDim.x) sTable[k] = table[k];
    __syncthreads();
    const uint64_t x = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nAa) return;
    uint64_t lo = 0, hi = nFrag;                                       // fragment of residue x: largest k with aa_off[k] <= x
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (A.aa_off[mid] <= x) lo = mid; else hi = mid; }
    const OrfRecord r = A.records[lo];
    Strand S;
    S.seq = A.nucl + A.offsets[r.contig]; S.len = (uint32_t) (A.offsets[r.contig + 1] - A.offsets[r.contig]); S.minus = (r.flags & 4u) != 0;
    const uint32_t p = r.s_from + 3u * (uint32_t) (x - A.aa_off[lo]);
    const char n0 = S.at(p), n1 = S.at(p + 1), n2 = S.at(p + 2);
    char aa = sTable[256 * cBaseCode[(unsigned char) n0] + 16 * cBaseCode[(unsigned char) n1] + cBaseCode[(unsigned char) n2]];
    const bool lower = (n0 >= 'a' && n0 <= 'z') || (n1 >= 'a' && n1 <= 'z') || (n2 >= 'a' && n2 <= 'z');
    if (lower && aa >= 'A' && aa <= 'Z') aa = (char) (aa + 32);
    aaAscii[x] = aa;
    aaCode[x] = cAaCode[(unsigned char) aa];
}

bool g_tablesReady = false;

}  // namespace

#define OCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return MK_ERR_DEVICE; } } while (0)
#define ONULL(p) do { if (!(p)) { err = "device scratch allocation failed (" #p ")"; return MK_ERR_DEVICE; } } while (0)

int run_extract_orfs(const char *dNucl, const uint64_t *dOffsets, uint32_t nContigs, uint32_t minLength, uint32_t maxLength, uint64_t maxGaps,
                     hipStream_t stream, OrfDeviceResult &R, std::string &err) {
    if (!g_tablesReady) {
        char comp[256]; uint8_t base[256], code[256];
        build_orf_tables(comp, base);
        for (int c = 0; c < 256; c++) { const char ch = (char) c; encode(&ch, 1, &code[c]); }
        OCHK(hipMemcpyToSymbol(HIP_SYMBOL(cComplement), comp, 256));
        OCHK(hipMemcpyToSymbol(HIP_SYMBOL(cBaseCode), base, 256));
        OCHK(hipMemcpyToSymbol(HIP_SYMBOL(cAaCode), code, 256));
        g_tablesReady = true;
    }
    R.n_frag = 0; R.n_aa = 0;
    if (nContigs == 0) return MK_OK;
    if (nContigs >= (1u << 30)) { err = "more than 2^30 contigs in one batch"; return MK_ERR_UNSUPPORTED; }
    const uint32_t nUnits = 2 * nContigs;
    uint64_t *dFragCount = (uint64_t *) dev_scratch("orf_fragcount", ((size_t) nUnits + 1) * 8);
    uint64_t *dAaCount = (uint64_t *) dev_scratch("orf_aacount", ((size_t) nUnits + 1) * 8);
    char *dTable = (char *) dev_scratch("orf_table", 4096);
    uint64_t *hTotals = (uint64_t *) pinned_scratch("orf_totals_h", 32);
    ONULL(dFragCount); ONULL(dAaCount); ONULL(dTable); ONULL(hTotals);
    {
        char table[4096];
        build_translation_table(table);
        OCHK(hipMemcpyAsync(dTable, table, 4096, hipMemcpyHostToDevice, stream));
        OCHK(hipStreamSynchronize(stream));                              // `table` lives on this frame
    }
    OrfScanArgs A;
    A.nucl = dNucl; A.offsets = dOffsets; A.n_contigs = nContigs;
    A.min_length = minLength; A.max_length = maxLength; A.max_gaps = maxGaps;
    A.frag_count = dFragCount; A.aa_count = dAaCount; A.frag_base = dFragCount; A.aa_base = dAaCount;
    A.records = nullptr; A.aa_off = nullptr;
    OCHK(hipMemsetAsync(dFragCount + nUnits, 0, 8, stream));
    OCHK(hipMemsetAsync(dAaCount + nUnits, 0, 8, stream));
    hipLaunchKernelGGL(orf_scan_kernel<false>, dim3((nUnits + 63) / 64), dim3(64), 0, stream, A);
    OCHK(hipGetLastError());
    size_t tb = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, tb, dFragCount, dFragCount, (int) nUnits + 1, stream);
    void *temp = dev_scratch("orf_temp", tb);
    ONULL(temp);
    OCHK(hipcub::DeviceScan::ExclusiveSum(temp, tb, dFragCount, dFragCount, (int) nUnits + 1, stream));
    OCHK(hipcub::DeviceScan::ExclusiveSum(temp, tb, dAaCount, dAaCount, (int) nUnits + 1, stream));
    OCHK(hipMemcpyAsync(hTotals, dFragCount + nUnits, 8, hipMemcpyDeviceToHost, stream));
    OCHK(hipMemcpyAsync(hTotals + 1, dAaCount + nUnits, 8, hipMemcpyDeviceToHost, stream));
    OCHK(hipStreamSynchronize(stream));
    const uint64_t nFrag = hTotals[0], nAa = hTotals[1];
    if (nFrag >= 0xFFFFFFFFull || nAa >= 0xFFFFFFFFull) { err = "more than 2^32 ORF fragments or residues in one batch: split the contigs"; return MK_ERR_UNSUPPORTED; }
    R.n_frag = nFrag; R.n_aa = nAa;
    if (nFrag == 0) return MK_OK;
    OCHK(hipMalloc((void **) &R.records, nFrag * sizeof(OrfRecord)));
    OCHK(hipMalloc((void **) &R.aa_off, (nFrag + 1) * 8));
    OCHK(hipMalloc((void **) &R.aa_ascii, nAa));
    OCHK(hipMalloc((void **) &R.aa_code, nAa));
    A.records = R.records; A.aa_off = R.aa_off;
    hipLaunchKernelGGL(orf_scan_kernel<true>, dim3((nUnits + 63) / 64), dim3(64), 0, stream, A);
    OCHK(hipGetLastError());
    OCHK(hipMemcpyAsync(R.aa_off + nFrag, dAaCount + nUnits, 8, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(orf_translate_kernel, dim3((unsigned) ((nAa + 255) / 256)), dim3(256), 0, stream, A, nFrag, nAa, dTable, R.aa_ascii, R.aa_code);
    OCHK(hipGetLastError());
    OCHK(hipStreamSynchronize(stream));
    return MK_OK;
}

void OrfDeviceResult::release() {
    if (records) (void) hipFree(records);
    if (aa_off) (void) hipFree(aa_off);
    if (aa_ascii) (void) hipFree(aa_ascii);
    if (aa_code) (void) hipFree(aa_code);
    records = nullptr; aa_off = nullptr; aa_ascii = nullptr; aa_code = nullptr; n_frag = 0; n_aa = 0;
}

}  // namespace mk
