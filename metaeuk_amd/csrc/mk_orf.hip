// metaeuk_amd/csrc/mk_orf.hip -- six-frame ORF fragments of a batch of contigs, translated, on the GPU
// (the producer of the hot path's queries).  Replaces, for `predictexons`' settings
// (orf-start-mode 1, both strands, all frames, contig start/end mode 2, genetic code 1):
//   Orf::setSequence / findAll / findForward    M/src/commons/Orf.cpp:118-345
//   the extractorfs loop, --translate branch    M/src/util/extractorfs.cpp:64-125
//   TranslateNucl::translate                    M/src/commons/TranslateNucl.h:488-503 (IUPAC-aware table: mk_host.cpp)
// Layout: contigs stay as the caller's ASCII in HBM; nothing is copied or reverse-complemented -- the minus strand is an
// index transform + complement lookup.  Three kernels:
//   orf_mark_kernel          one lane per strand position: does a fragment end here, and how many codons has it
//   orf_scan_*_kernel        two prefix sums in one pass (fragments, residues): the rank of every fragment = the reference's output order
//                            (= the renumbered ORF ids) and its residue offset
//   orf_write_kernel         one record per fragment at its rank
//   orf_translate_kernel     one lane per amino acid: codon -> residue (ASCII, case preserved) and aa2num code
// The fragment codes stay in HBM and become the query batch of mk_search without a host round trip.
#include "mk_orf.hpp"
#include "mk_kernels.hpp"
#include "mk_segsort.hpp"
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

// With orf-start-mode 1 a fragment is a maximal stop-free codon run of one frame: it ENDS at a stop codon or at the last
// complete codon of the frame, and STARTS three behind the previous stop of the frame (or at the frame offset: incomplete
// start).  So every strand position decides on its own whether a fragment ends there (one lane per position, a short
// backward walk to the previous stop), and the reference's output order -- contig, plus strand then minus strand, end
// position ascending across the three frames -- is the order of the position index itself: ranks are a prefix sum.
// (Orf::findForward's per-frame state machine, Orf.cpp:220-345; --max-gaps is the reference's default INT_MAX, i.e. off.)
__device__ __forceinline__ bool stop_at(const Strand &S, uint32_t p) {
    const char c0 = upper_or_max(S.at(p));
    if (c0 != 'T') return false;
    const char c1 = upper_or_max(S.at(p + 1)), c2 = upper_or_max(S.at(p + 2));
    return (c1 == 'A' && (c2 == 'A' || c2 == 'G')) || (c1 == 'G' && c2 == 'A');
}

// fragment ending at strand position p (0 codons = none): start, length, flags
__device__ __forceinline__ uint32_t fragment_at(const Strand &S, uint32_t p, uint32_t minLength, uint32_t maxLength, uint32_t &from, uint32_t &flags) {
    if (S.len < 3 || p + 3 > S.len) return 0;                           // incomplete codon: never ends a fragment
    const bool stop = stop_at(S, p);
    const bool last = p + 6 > S.len;                                    // the frame's next codon is incomplete
    if (!stop && !last) return 0;
    uint32_t q = p;
    bool hasStart = false;
    while (q >= 3) {                                                    // previous stop of this frame
        if (stop_at(S, q - 3)) { hasStart = true; break; }
        q -= 3;
    }
    from = q;                                                           // = p % 3 when no stop precedes
    const uint32_t count = (p - from) / 3 + (stop ? 0u : 1u);
    if (count == 0 || count < minLength || count > maxLength) return 0;
    flags = (hasStart ? 0u : 1u) | (stop ? 0u : 2u) | (S.minus ? 4u : 0u);
    return count;
}

// virtual position index x = 2 * offsets[contig] + strand * len + p
__device__ __forceinline__ bool locate(const OrfScanArgs &A, uint64_t x, uint32_t &contig, Strand &S, uint32_t &p) {
    uint32_t lo = 0, hi = A.n_contigs;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (2 * A.offsets[mid] <= x) lo = mid; else hi = mid; }
    contig = lo;
    const uint64_t base = A.offsets[lo];
    const uint64_t len = A.offsets[lo + 1] - base;
    const uint64_t xl = x - 2 * base;
    if (xl >= 2 * len) return false;                                    // (cannot happen: offsets are ascending)
    S.seq = A.nucl + base; S.len = (uint32_t) len; S.minus = xl >= len;
    p = (uint32_t) (xl - (S.minus ? len : 0));
    return true;
}

__global__ __launch_bounds__(256) void orf_mark_kernel(OrfScanArgs A, uint64_t nPos, uint16_t *naa) {
    const uint64_t x = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nPos) return;
    uint32_t contig, p, from, flags;
    Strand S;
    uint32_t n = 0;
    if (locate(A, x, contig, S, p)) n = fragment_at(S, p, A.min_length, A.max_length, from, flags);
    naa[x] = (uint16_t) n;                                              // <= 32734 codons
}

__global__ __launch_bounds__(256) void orf_write_kernel(OrfScanArgs A, uint64_t nPos, const uint16_t *naa, const uint64_t *rank, const uint64_t *aaBase) {
    const uint64_t x = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nPos || naa[x] == 0) return;
    uint32_t contig, p, from = 0, flags = 0;
    Strand S;
    locate(A, x, contig, S, p);
    const uint32_t n = fragment_at(S, p, A.min_length, A.max_length, from, flags);
    OrfRecord r;
    r.contig = contig; r.s_from = from; r.n_aa = n; r.flags = flags;
    A.records[rank[x]] = r;
    A.aa_off[rank[x]] = aaBase[x];
}

// ---- the two exclusive prefix sums over the strand positions -- fragments ending at or before a position (its rank) and their residues (its
// offset) -- in one pass of two sweeps: 4096 positions per workgroup (16 per thread), the workgroups' sums in 64 bits, the rest in 32
// (a workgroup holds at most 4096 x 65535 residues).  [n] of both outputs = the totals.
constexpr uint32_t ORF_SCAN_TILE = 4096;
__global__ __launch_bounds__(256) void orf_scan_sums_kernel(const uint16_t *naa, uint64_t n, uint64_t *blockSums /* [2 * blocks] */) {
    __shared__ uint32_t sm[2 * 16];
    const uint64_t i0 = (uint64_t) blockIdx.x * ORF_SCAN_TILE + (uint64_t) threadIdx.x * 16u;
    uint32_t v[2] = {0, 0}, excl[2], total[2];
    for (uint32_t k = 0; k < 16; k++) if (i0 + k < n) { const uint32_t x = naa[i0 + k]; v[0] += x ? 1u : 0u; v[1] += x; }
    segsort::block_scan<2>(v, excl, total, sm);
    if (threadIdx.x == 0) { blockSums[2 * (size_t) blockIdx.x] = total[0]; blockSums[2 * (size_t) blockIdx.x + 1] = total[1]; }
}
// one workgroup: exclusive scan of the workgroups' sums in place, the grand totals behind them ([2 * blocks], [2 * blocks + 1])
__global__ __launch_bounds__(1024) void orf_scan_blocks_kernel(uint64_t *blockSums, uint32_t blocks) {
    __shared__ uint64_t sCarry[2];
    __shared__ uint64_t sWave[2][16];
    if (threadIdx.x == 0) { sCarry[0] = 0; sCarry[1] = 0; }
    __syncthreads();
    const int lane = (int) (threadIdx.x & 63u), w = (int) (threadIdx.x >> 6);
    for (uint32_t b0 = 0; b0 < blocks; b0 += 1024) {
        const uint32_t b = b0 + threadIdx.x;
        uint64_t v[2] = {b < blocks ? blockSums[2 * (size_t) b] : 0ull, b < blocks ? blockSums[2 * (size_t) b + 1] : 0ull}, inc[2];
        for (int k = 0; k < 2; k++) {
            uint64_t x = v[k];
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t lo = (uint32_t) __shfl_up((int) (uint32_t) x, d, 64), hi = (uint32_t) __shfl_up((int) (uint32_t) (x >> 32), d, 64);
                if (lane >= d) x += ((uint64_t) hi << 32) | lo;
            }
            inc[k] = x;
            if (lane == 63) sWave[k][w] = x;
        }
        __syncthreads();
        for (int k = 0; k < 2; k++) {
            uint64_t base = sCarry[k];
            for (int i = 0; i < w; i++) base += sWave[k][i];
            if (b < blocks) blockSums[2 * (size_t) b + k] = base + inc[k] - v[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) for (int k = 0; k < 2; k++) { uint64_t t = 0; for (int i = 0; i < 16; i++) t += sWave[k][i]; sCarry[k] += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { blockSums[2 * (size_t) blocks] = sCarry[0]; blockSums[2 * (size_t) blocks + 1] = sCarry[1]; }
}
__global__ __launch_bounds__(256) void orf_scan_apply_kernel(const uint16_t *naa, uint64_t n, const uint64_t *blockSums, uint32_t blocks, uint64_t *rank, uint64_t *aaBase) {
    __shared__ uint32_t sm[2 * 16];
    const uint64_t i0 = (uint64_t) blockIdx.x * ORF_SCAN_TILE + (uint64_t) threadIdx.x * 16u;
    uint32_t x[16], v[2] = {0, 0}, excl[2], total[2];
    for (uint32_t k = 0; k < 16; k++) { x[k] = i0 + k < n ? (uint32_t) naa[i0 + k] : 0u; v[0] += x[k] ? 1u : 0u; v[1] += x[k]; }
    segsort::block_scan<2>(v, excl, total, sm);
    uint64_t r = blockSums[2 * (size_t) blockIdx.x] + excl[0], a = blockSums[2 * (size_t) blockIdx.x + 1] + excl[1];
    for (uint32_t k = 0; k < 16; k++) {
        if (i0 + k < n) { rank[i0 + k] = r; aaBase[i0 + k] = a; }
        r += x[k] ? 1u : 0u; a += x[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { rank[n] = blockSums[2 * (size_t) blocks]; aaBase[n] = blockSums[2 * (size_t) blocks + 1]; }
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
    if (maxGaps < (uint64_t) INT_MAX) { err = "--max-gaps below the reference default (unlimited) is not implemented"; return MK_ERR_UNSUPPORTED; }
    uint64_t *hTotals = (uint64_t *) pinned_scratch("orf_totals_h", 32);
    ONULL(hTotals);
    OCHK(hipMemcpyAsync(hTotals, dOffsets + nContigs, 8, hipMemcpyDeviceToHost, stream));
    OCHK(hipStreamSynchronize(stream));
    const uint64_t nPos = 2 * hTotals[0];                              // both strands of every contig
    if (nPos == 0) return MK_OK;
    if (nPos >= 0x7FFFFFFFull) { err = "more than 2^30 nucleotides in one batch: split the contigs"; return MK_ERR_UNSUPPORTED; }
    uint16_t *dNaa = (uint16_t *) dev_scratch("orf_naa", nPos * 2);
    uint64_t *dRank = (uint64_t *) dev_scratch("orf_rank", (nPos + 1) * 8);
    uint64_t *dAaBase = (uint64_t *) dev_scratch("orf_aabase", (nPos + 1) * 8);
    char *dTable = (char *) dev_scratch("orf_table", 4096);
    ONULL(dNaa); ONULL(dRank); ONULL(dAaBase); ONULL(dTable);
    {
        char table[4096];
        build_translation_table(table);
        OCHK(hipMemcpyAsync(dTable, table, 4096, hipMemcpyHostToDevice, stream));
        OCHK(hipStreamSynchronize(stream));                              // `table` lives on this frame
    }
    OrfScanArgs A;
    A.nucl = dNucl; A.offsets = dOffsets; A.n_contigs = nContigs;
    A.min_length = minLength; A.max_length = maxLength; A.max_gaps = maxGaps;
    A.records = nullptr; A.aa_off = nullptr;
    const unsigned blocks = (unsigned) ((nPos + 255) / 256);
    hipLaunchKernelGGL(orf_mark_kernel, dim3(blocks), dim3(256), 0, stream, A, nPos, dNaa);
    OCHK(hipGetLastError());
    {
        const uint32_t sb = (uint32_t) ((nPos + ORF_SCAN_TILE - 1) / ORF_SCAN_TILE);
        uint64_t *dBlockSums = (uint64_t *) dev_scratch("orf_scansums", ((size_t) sb + 1) * 16);
        ONULL(dBlockSums);
        hipLaunchKernelGGL(orf_scan_sums_kernel, dim3(sb), dim3(256), 0, stream, (const uint16_t *) dNaa, nPos, dBlockSums);
        hipLaunchKernelGGL(orf_scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, dBlockSums, sb);
        hipLaunchKernelGGL(orf_scan_apply_kernel, dim3(sb), dim3(256), 0, stream, (const uint16_t *) dNaa, nPos, (const uint64_t *) dBlockSums, sb, dRank, dAaBase);
        OCHK(hipGetLastError());
    }
    OCHK(hipMemcpyAsync(hTotals, dRank + nPos, 8, hipMemcpyDeviceToHost, stream));
    OCHK(hipMemcpyAsync(hTotals + 1, dAaBase + nPos, 8, hipMemcpyDeviceToHost, stream));
    OCHK(hipStreamSynchronize(stream));
    const uint64_t nFrag = hTotals[0], nAa = hTotals[1];
    if (nFrag >= 0xFFFFFFFFull || nAa >= 0xFFFFFFFFull) { err = "more than 2^32 ORF fragments or residues in one batch: split the contigs"; return MK_ERR_UNSUPPORTED; }
    R.n_frag = nFrag; R.n_aa = nAa;
    if (nFrag == 0) return MK_OK;
    // (pooled blocks: a hipFree at the end of the batch would wait for the search of the next one)
    R.records = (OrfRecord *) dev_block_alloc(nFrag * sizeof(OrfRecord), &R.cap[0]);
    R.aa_off = (uint64_t *) dev_block_alloc((nFrag + 1) * 8, &R.cap[1]);
    R.aa_ascii = (char *) dev_block_alloc(std::max<uint64_t>(nAa, 1), &R.cap[2]);
    R.aa_code = (uint8_t *) dev_block_alloc(std::max<uint64_t>(nAa, 1), &R.cap[3]);
    ONULL(R.records); ONULL(R.aa_off); ONULL(R.aa_ascii); ONULL(R.aa_code);
    A.records = R.records; A.aa_off = R.aa_off;
    hipLaunchKernelGGL(orf_write_kernel, dim3(blocks), dim3(256), 0, stream, A, nPos, dNaa, dRank, dAaBase);
    OCHK(hipGetLastError());
    OCHK(hipMemcpyAsync(R.aa_off + nFrag, dAaBase + nPos, 8, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(orf_translate_kernel, dim3((unsigned) ((nAa + 255) / 256)), dim3(256), 0, stream, A, nFrag, nAa, dTable, R.aa_ascii, R.aa_code);
    OCHK(hipGetLastError());
    OCHK(hipStreamSynchronize(stream));
    return MK_OK;
}

void OrfDeviceResult::release() {
    dev_block_free(records, cap[0]);
    dev_block_free(aa_off, cap[1]);
    dev_block_free(aa_ascii, cap[2]);
    dev_block_free(aa_code, cap[3]);
    records = nullptr; aa_off = nullptr; aa_ascii = nullptr; aa_code = nullptr; n_frag = 0; n_aa = 0;
}

}  // namespace mk
