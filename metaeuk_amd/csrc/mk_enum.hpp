// metaeuk_amd/csrc/mk_enum.hpp -- wave-level enumeration of the similar k-mers of one k-mer start (device code).
// Reference: KmerGenerator::generateKmerList / calculateArrayProduct (M/src/prefiltering/KmerGenerator.cpp:107-216).
// The reference multiplies two score-sorted 3-mer rows: for the first-half candidates a = 0,1,.. (descending score s0[a])
// it takes the second-half candidates b = 0 .. nb(a)-1 with s0[a] + s1[b] >= threshold, and stops at the first a whose
// best partner fails.  The k-mers come out in "product order" (a major, b minor); that order is what the
// double-diagonal rule later sees, so it is kept exactly:  lane j of a batch holds product number base + j.
//
// What is different from a transcription: nothing is searched, and the dependent-load chain of a position is short.
//   * nb(a) comes from per-row cumulative score histograms (cum3): one table lookup;
//   * 256 first-half candidates are taken per step (4 per lane), which covers nearly every position in one step;
//   * the product -> (a, b) map of a 64-product window is a counting trick: owner(x) = #{a : E_a <= x} for the inclusive
//     prefix E of nb, i.e. a histogram of E over the window (LDS atomics) + a wave prefix sum in DPP;
//   * prefix sums run in DPP (row_shr / row_bcast), not through the LDS crossbar.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "mk_prefilter.hpp"

namespace mk {
namespace enumk {

constexpr int WAVE = 64;
constexpr int N3 = 8000;

// inclusive prefix sum over the 64 lanes of a wave (GFX9 DPP: row_shr within the 16-lane rows, then row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    int x = (int) v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return (uint32_t) x;
}
// inclusive prefix maximum over the 64 lanes (values >= 0)
__device__ __forceinline__ uint32_t wave_incl_max_scan(uint32_t v) {
    int x = (int) v;
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false));
    return (uint32_t) x;
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t) __builtin_amdgcn_readlane((int) v, 63); }

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t wave_read_lane(uint32_t v, uint32_t srcLane) { return (uint32_t) __builtin_amdgcn_ds_bpermute((int) (srcLane << 2), (int) v); }

// Index lists longer than one entry: every lane holds `rem` further entries of its list.  Instead of each lane walking its own
// list (a wave then loops as often as its longest list, with one or two lanes busy), the entries are dealt out one per lane:
// item x of the wave belongs to the lane whose [exclusive prefix, +rem) range covers it; the owners mark their range starts in a
// 64-byte LDS window and a prefix maximum spreads the owner over its range.  fn(ownerLane, e) is called by the lane that got
// entry e (>= 1) of ownerLane's list; it fetches what it needs from the owner with wave_read_lane.
template <typename F>
__device__ __forceinline__ void wave_deal_tail(uint32_t rem, int lane, uint8_t *mark /* [64] per wave */, F &&fn) {
    const uint32_t incl = wave_incl_scan(rem), excl = incl - rem, total = wave_last(incl);
    for (uint32_t base = 0; base < total; base += WAVE) {
        mark[lane] = 0;
        wave_sync_lds();
        if (rem && excl < base + WAVE && incl > base) mark[max(excl, base) - base] = (uint8_t) (lane + 1);
        wave_sync_lds();
        const uint32_t owner = wave_incl_max_scan(mark[lane]);      // 1 + owner lane (an item of the window always has one)
        const uint32_t src = owner ? owner - 1 : 0;
        const uint32_t ownerExcl = wave_read_lane(excl, src);
        const bool valid = base + (uint32_t) lane < total;
        fn(src, 1u + (base + (uint32_t) lane - ownerExcl), valid);
        wave_sync_lds();
    }
}

// The same deal for the tails of U lists per lane at once (items in the order list group u, lane, entry -- the order the hits of a batch are numbered
// in): one scan chain and one round of windows for the whole batch instead of one per group, and -- what matters to a kernel that lives on requests in
// flight -- the caller gets the owner of a window's items BEFORE it touches memory, so it can issue the window's entry loads beside the first entries'.
// An owner is named by id = u * 64 + lane.
template <int U>
struct TailDeal {
    static_assert(U * WAVE <= 256, "owner ids are marked in bytes");
    uint32_t excl[U], incl[U], total;
    // rem = size - 1 of the non-empty lists; ex = the exclusive prefix of the sizes over (u, lane), totAll their sum: the prefix of rem is that of size less
    // the number of non-empty lists before this one -- a ballot and a bit count, no second scan
    __device__ __forceinline__ void init(const uint32_t (&size)[U], const uint32_t (&rem)[U], const uint32_t (&ex)[U], uint32_t totAll) {
        uint32_t lists = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned long long nz = __ballot(size[u] != 0u);
            excl[u] = ex[u] - lists - __builtin_amdgcn_mbcnt_hi((uint32_t) (nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) nz, 0u));
            incl[u] = excl[u] + rem[u];
            lists += (uint32_t) __popcll(nz);
        }
        total = totAll - lists;
    }
    // window [base, base + 64): the lane learns whose item it got (id), which entry of that list (e >= 1), and whether there is an item at all
    __device__ __forceinline__ void window(uint32_t base, const uint32_t (&rem)[U], int lane, uint8_t *mark /* [64] per wave */, uint32_t &id, uint32_t &e, bool &valid) const {
        mark[lane] = 0;
        wave_sync_lds();
#pragma unroll
        for (int u = 0; u < U; u++)
            if (rem[u] && excl[u] < base + WAVE && incl[u] > base) mark[max(excl[u], base) - base] = (uint8_t) (u * WAVE + lane);
        wave_sync_lds();
        // the ranges ascend with the id, so the prefix maximum spreads an owner over its range; id 0 needs no mark of its own (an item of the window always
        // has an owner at or before it, and the only owner "no mark" can stand for is the smallest)
        id = wave_incl_max_scan(mark[lane]);
        e = 1u + (base + (uint32_t) lane - pick(excl, id));
        valid = base + (uint32_t) lane < total;
        wave_sync_lds();
    }
    // v[u] of lane l for id = u * 64 + l
    __device__ __forceinline__ static uint32_t pick(const uint32_t (&v)[U], uint32_t id) {
        uint32_t r = wave_read_lane(v[0], id & 63u);
#pragma unroll
        for (int u = 1; u < U; u++) { const uint32_t x = wave_read_lane(v[u], id & 63u); if ((id >> 6) == (uint32_t) u) r = x; }
        return r;
    }
};

constexpr int AJ = 2;                    // first-half candidates per lane and step
constexpr int ABLOCK = AJ * WAVE;        // ... per step (one step covers almost every position: the dependent-load chain
                                         // of a position is rows -> cumulative counts -> second-half indices -> index probes)

// per-wave LDS scratch of the enumerator
template <int U>
struct EnumLds {
    uint32_t start[ABLOCK];      // exclusive prefix of nb over the first-half candidates of the current step
    uint32_t idx0[ABLOCK];       // their share of the k-mer's table cell (cell_first of the 3-mer's address code)
    uint32_t cnt[U * WAVE];      // histogram of the inclusive prefixes over the current product window
    uint32_t sec[WAVE];          // the 64 best second halves (cell_second of their address codes): fetched with the row heads, one dependent
                                 // load less for nearly every product (a first half rarely pairs with more than a few second halves)
};

// owner(x) of the U * 64 products of a window from the histogram of the inclusive prefixes: carry + inclusive prefix sum of cnt over the window.  A step has
// at most ABLOCK candidates, so the counts of two 64-product groups are scanned as the halves of ONE register (a scan is 6 DPP adds).
template <int U>
__device__ __forceinline__ void window_owners(const uint32_t *cnt, int lane, uint32_t carry, uint32_t (&owner)[U]) {
#pragma unroll
    for (int u = 0; u + 1 < U; u += 2) {
        const uint32_t sc = wave_incl_scan(cnt[u * WAVE + lane] | (cnt[(u + 1) * WAVE + lane] << 16));
        const uint32_t tot = wave_last(sc);
        owner[u] = carry + (sc & 0xFFFFu);
        owner[u + 1] = carry + (tot & 0xFFFFu) + (sc >> 16);
        carry += (tot & 0xFFFFu) + (tot >> 16);
    }
    if constexpr (U % 2 == 1) owner[U - 1] = carry + wave_incl_scan(cnt[(U - 1) * WAVE + lane]);
}

// Table cell of a k-mer = cell_first(address code of its first 3-mer) + cell_second(code of its second 3-mer); a code is tile << 6 | w
// (mk_host.cpp: the tiled address order -- k-mers that differ by substitutions inside the residue quads share 128-byte lines)
__device__ __forceinline__ uint32_t cell_first(uint32_t code) { return (code >> 6) * 4096u + (code & 63u); }
__device__ __forceinline__ uint32_t cell_second(uint32_t code) { return (code >> 6) * 512000u + (code & 63u) * 64u; }

// Calls onBatch(kmer[U], has[U]) for consecutive windows of U*64 products of the k-mer start whose residues are r[0..9]
// (spaced seed 1101010011), in product order; onBatch returns false to stop early.  Returns the number of similar k-mers.
template <int U, class F>
__device__ __forceinline__ uint32_t enumerate_position(const PrefilterDeviceView &V, const uint8_t *r, int thr, int lane, EnumLds<U> &S, F &&onBatch) {
    const uint32_t idx0 = r[0] + 20u * r[1] + 400u * r[3];
    const uint32_t idx1 = r[5] + 20u * r[8] + 400u * r[9];
    const int16_t *s0 = V.score3 + (size_t) idx0 * N3;
    const uint16_t *i0 = V.index3 + (size_t) idx0 * N3;
    const uint16_t *i1 = V.index3 + (size_t) idx1 * N3;
    const int R = V.hist_range, lo = V.hist_lo;
    const uint16_t *cum1 = V.cum3 + (size_t) idx1 * R;
    const int cutoff1 = thr - (int) V.score3[(size_t) idx1 * N3];   // first halves below this cannot reach the threshold
    S.sec[lane] = cell_second(i1[lane]);
    uint32_t kmers = 0;
    for (uint32_t a0 = 0; a0 < (uint32_t) N3; a0 += ABLOCK) {
        // candidate a = a0 + j*64 + lane: (j, lane) ascending = a ascending
        uint32_t nb[AJ], incl[AJ], ia[AJ];
        int sa[AJ];
#pragma unroll
        for (int j = 0; j < AJ; j++) {
            const uint32_t a = a0 + (uint32_t) (j * WAVE + lane);
            sa[j] = a < (uint32_t) N3 ? (int) s0[a] : -32768;
            ia[j] = a < (uint32_t) N3 ? (uint32_t) i0[a] : 0u;
        }
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < AJ; j++) {
            nb[j] = 0;
            if (sa[j] >= cutoff1) {
                const int xb = thr - sa[j] - lo;
                nb[j] = xb <= 0 ? (uint32_t) N3 : (xb >= R ? 0u : (uint32_t) cum1[xb]);
            }
        }
#pragma unroll
        for (int j = 0; j < AJ; j++) {
            const uint32_t sc = wave_incl_scan(nb[j]);
            incl[j] = total + sc;
            total += wave_last(sc);
            S.start[j * WAVE + lane] = incl[j] - nb[j];
            S.idx0[j * WAVE + lane] = cell_first(ia[j]);
        }
        const bool more = __builtin_amdgcn_readlane(sa[AJ - 1], WAVE - 1) >= cutoff1;   // the step's last candidate is still valid
        kmers += total;
        for (uint32_t base = 0; base < total; base += U * WAVE) {
            // owner(x) = #{a : incl_a <= x}: histogram of incl over the window, prefix-summed
#pragma unroll
            for (int u = 0; u < U; u++) S.cnt[u * WAVE + lane] = 0;
            wave_sync_lds();
            uint32_t carry = 0;
#pragma unroll
            for (int j = 0; j < AJ; j++) {
                const uint32_t rel = incl[j] - base;               // wraps for incl < base: not in the window
                if (incl[j] >= base && rel < (uint32_t) (U * WAVE)) atomicAdd(&S.cnt[rel], 1u);
                carry += (uint32_t) __popcll(__ballot(incl[j] < base));
            }
            wave_sync_lds();
            uint32_t kmer[U], owner[U];
            bool has[U];
            window_owners<U>(S.cnt, lane, carry, owner);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t x = base + (uint32_t) (u * WAVE + lane);
                has[u] = x < total;
                kmer[u] = 0;
                if (has[u]) {
                    const uint32_t b = x - S.start[owner[u]];
                    kmer[u] = S.idx0[owner[u]] + (b < (uint32_t) WAVE ? S.sec[b] : cell_second(i1[b]));
                }
            }
            if (!onBatch(kmer, has)) return kmers;
            wave_sync_lds();
        }
        wave_sync_lds();
        if (!more) break;
    }
    return kmers;
}

// ---- k = 7 (databases of 3.35e9 residues or more; spaced seed 11010110011) ----
// KmerGenerator::setDivideStrategy cuts a 7-mer into a 2-mer, a 2-mer and a 3-mer row and multiplies them step by step
// (M/src/prefiltering/KmerGenerator.cpp:41-86,107-187): the list is the lexicographic order of the three ranks (a, b, c), pruned at
// every step against the best of the remaining rows (mk_kmer7.hip restates it as list kernels).  In a wave: the first-row candidates a
// are walked one after the other (their scores sit in a register of the wave: v_readlane, no load in the loop), the 64 best second-row
// candidates b are the lanes, the number of third-row partners of a pair is one lookup in the row's cumulative score histogram (staged
// in LDS once per start), and the product -> (b, c) map of a 64 x U window is the counting trick of enumerate_position.
// Table cell of a 7-mer = n2a + 400 n2b + 160000 n3 (the reference's numbering; V.num3 maps an index3 address code to the 3-mer number).
template <int U>
struct Enum7Lds {
    uint32_t start[WAVE];        // exclusive prefix of the partner counts over the second-row candidates of the current step
    uint32_t idx0[WAVE];         // n2a + 400 n2b of the pair
    uint32_t cnt[U * WAVE];      // histogram of the inclusive prefixes over the current product window
    uint32_t sec[WAVE];          // 160000 n3 of the 64 best third-row partners
    uint16_t cum[256];           // the third row's cumulative score histogram (hist_range <= 256, checked by the host)
};

template <int U, class F>
__device__ __forceinline__ uint32_t enumerate7_position(const PrefilterDeviceView &V, const uint8_t *r, int thr, int lane, Enum7Lds<U> &S, F &&onBatch) {
    constexpr int N2 = 400;
    const uint32_t idx0 = r[0] + 20u * r[1];
    const uint32_t idx1 = r[3] + 20u * r[5];
    const uint32_t idx2 = r[6] + 20u * r[9] + 400u * r[10];
    const int16_t *s0 = V.score2 + (size_t) idx0 * N2, *s1 = V.score2 + (size_t) idx1 * N2;
    const uint16_t *i0 = V.index2 + (size_t) idx0 * N2, *i1 = V.index2 + (size_t) idx1 * N2, *i2 = V.index3 + (size_t) idx2 * N3;
    const int R = V.hist_range, lo = V.hist_lo;
    const uint16_t *cum = V.cum3 + (size_t) idx2 * R;
    for (int k = lane; k < 256; k += WAVE) S.cum[k] = k < R ? cum[k] : (uint16_t) 0;
    S.sec[lane] = 160000u * (uint32_t) V.num3[i2[lane]];
    const int rest1 = (int) V.score3[(size_t) idx2 * N3];
    const int saReg = (int) s0[lane];                    // the 64 best first-row candidates (a walk rarely passes them) ...
    const uint32_t iaReg = (uint32_t) i0[lane];
    const int sb0 = (int) s1[lane];                      // ... and second-row candidates
    const uint32_t ib0 = (uint32_t) i1[lane];
    const int rest0 = __builtin_amdgcn_readlane(sb0, 0) + rest1;
    wave_sync_lds();
    uint32_t kmers = 0;
    for (int a = 0; a < N2; a++) {
        int sa;
        uint32_t ca;
        if (a < WAVE) { sa = __builtin_amdgcn_readlane(saReg, a); ca = (uint32_t) __builtin_amdgcn_readlane((int) iaReg, a); }
        else { sa = (int) s0[a]; ca = (uint32_t) i0[a]; }
        if (sa < thr - rest0) break;
        const int cutB = thr - sa - rest1;
        for (int b0 = 0; b0 < N2; b0 += WAVE) {
            int sb = sb0;
            uint32_t cb = ib0;
            if (b0 != 0) { const int b = b0 + lane; sb = b < N2 ? (int) s1[b] : -32768; cb = b < N2 ? (uint32_t) i1[b] : 0u; }
            uint32_t nb = 0;
            if (sb >= cutB) {
                const int xb = thr - sa - sb - lo;
                nb = xb <= 0 ? (uint32_t) N3 : (xb >= R ? 0u : (uint32_t) S.cum[xb]);
            }
            const uint32_t incl = wave_incl_scan(nb);
            const uint32_t total = wave_last(incl);
            S.start[lane] = incl - nb;
            S.idx0[lane] = ca + 400u * cb;
            const bool more = __builtin_amdgcn_readlane(sb, WAVE - 1) >= cutB;      // the step's last candidate is still valid
            kmers += total;
            for (uint32_t base = 0; base < total; base += U * WAVE) {
#pragma unroll
                for (int u = 0; u < U; u++) S.cnt[u * WAVE + lane] = 0;
                wave_sync_lds();
                const uint32_t rel = incl - base;                       // wraps for incl < base: not in the window
                if (incl >= base && rel < (uint32_t) (U * WAVE)) atomicAdd(&S.cnt[rel], 1u);
                uint32_t carry = (uint32_t) __popcll(__ballot(incl < base));
                wave_sync_lds();
                uint32_t kmer[U], owner[U];
                bool has[U];
                window_owners<U>(S.cnt, lane, carry, owner);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t x = base + (uint32_t) (u * WAVE + lane);
                    has[u] = x < total;
                    kmer[u] = 0;
                    if (has[u]) {
                        const uint32_t c = x - S.start[owner[u]];
                        kmer[u] = S.idx0[owner[u]] + (c < (uint32_t) WAVE ? S.sec[c] : 160000u * (uint32_t) V.num3[i2[c]]);
                    }
                }
                if (!onBatch(kmer, has)) return kmers;
                wave_sync_lds();
            }
            wave_sync_lds();
            if (!more) break;
        }
    }
    return kmers;
}

}  // namespace enumk
}  // namespace mk
