// metaeuk_amd/csrc/mk_segsort.hpp -- per-segment sorting and small block-wide scans (device code).
//
// The tails of both stages order things PER QUERY: the hits of a query by (score, target) (QueryMatcher::getResult ->
// std::sort(..., hit_t::compareHitsByScoreAndId), M/src/prefiltering/QueryMatcher.cpp:117-125), its alignment jobs by target
// length.  A query has tens of such items, rarely thousands -- a device-wide radix sort over (query | key) moves every record
// eight times through HBM to establish an order that a wave can produce in registers.  Here a segment is sorted by whoever owns
// it: 64 keys by one wave (rank by counting, no LDS), up to a tile by a workgroup in LDS (bitonic network), longer ones by the
// same workgroup with the network's wide strides in HBM and the narrow ones tile by tile in LDS.
//
// Keys are uint64, distinct within a segment, ~0 is the padding value.  The networks use ascending comparators only (the first
// half-cleaner of a merge level mirrors its partner index), so a segment of n < 2^k keys is sorted as if padded with +inf above n
// and the padding never has to exist.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "mk_enum.hpp"

namespace mk {
namespace segsort {

__device__ __forceinline__ void partner(uint32_t i, uint32_t k, uint32_t j, bool flip, uint32_t &l, uint32_t &r) {
    if (flip) {
        const uint32_t half = k >> 1, blk = i / half, o = i % half;
        l = blk * k + o; r = blk * k + k - 1u - o;
    } else {
        l = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)); r = l + j;
    }
}

__device__ __forceinline__ uint32_t pow2_at_least(uint32_t n) { uint32_t p = 2; while (p < n) p <<= 1; return p; }

// rank of this lane's key among the keys of lanes [0, n) of the wave (n wave-uniform; keys distinct except for the padding ~0)
__device__ __forceinline__ uint32_t wave_rank(uint64_t key, uint32_t n) {
    const int lo = (int) (uint32_t) key, hi = (int) (uint32_t) (key >> 32);
    n = (uint32_t) __builtin_amdgcn_readfirstlane((int) n);
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < n; j++) {
        const uint64_t kj = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane(hi, (int) j) << 32) | (uint64_t) (uint32_t) __builtin_amdgcn_readlane(lo, (int) j);
        cnt += kj < key ? 1u : 0u;
    }
    return cnt;
}

// sK[0, P) ascending, P a power of two >= 2; every thread of the workgroup calls it.  Ends with a barrier.
template <int THREADS>
__device__ __forceinline__ void lds_sort(uint64_t *sK, uint32_t P) {
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const bool flip = j == (k >> 1);
            for (uint32_t i = threadIdx.x; i < (P >> 1); i += THREADS) {
                uint32_t l, r;
                partner(i, k, j, flip, l, r);
                const uint64_t x = sK[l], y = sK[r];
                if (x > y) { sK[l] = y; sK[r] = x; }
            }
            __syncthreads();
        }
    }
}

// a[0, n) ascending in HBM by one workgroup, TILE keys of LDS (TILE a power of two).  Ends with a barrier.
template <int THREADS, uint32_t TILE>
__device__ __forceinline__ void global_sort(uint64_t *a, uint32_t n, uint64_t *sK) {
    if (n < 2) return;
    const uint32_t P = pow2_at_least(n);
    // the tiles, each completely
    for (uint32_t c0 = 0; c0 < n; c0 += TILE) {
        const uint32_t m = min(TILE, n - c0), Pm = min(TILE, pow2_at_least(m));
        for (uint32_t t = threadIdx.x; t < Pm; t += THREADS) sK[t] = t < m ? a[c0 + t] : ~0ull;
        __syncthreads();
        lds_sort<THREADS>(sK, Pm);
        for (uint32_t t = threadIdx.x; t < m; t += THREADS) a[c0 + t] = sK[t];
        __syncthreads();
    }
    // merge levels wider than a tile: strides >= TILE in HBM, the rest per tile in LDS
    for (uint32_t k = TILE << 1; k <= P && k != 0; k <<= 1) {
        for (uint32_t j = k >> 1; j >= TILE; j >>= 1) {
            const bool flip = j == (k >> 1);
            for (uint32_t i = threadIdx.x; i < (P >> 1); i += THREADS) {
                uint32_t l, r;
                partner(i, k, j, flip, l, r);
                if (r >= n) continue;                          // the partner is padding
                const uint64_t x = a[l], y = a[r];
                if (x > y) { a[l] = y; a[r] = x; }
            }
            __syncthreads();
        }
        for (uint32_t c0 = 0; c0 < n; c0 += TILE) {
            const uint32_t m = min(TILE, n - c0);
            for (uint32_t t = threadIdx.x; t < TILE; t += THREADS) sK[t] = t < m ? a[c0 + t] : ~0ull;
            __syncthreads();
            for (uint32_t j = TILE >> 1; j > 0; j >>= 1) {
                for (uint32_t i = threadIdx.x; i < (TILE >> 1); i += THREADS) {
                    const uint32_t l = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), r = l + j;
                    const uint64_t x = sK[l], y = sK[r];
                    if (x > y) { sK[l] = y; sK[r] = x; }
                }
                __syncthreads();
            }
            for (uint32_t t = threadIdx.x; t < m; t += THREADS) a[c0 + t] = sK[t];
            __syncthreads();
        }
    }
}

// exclusive prefix and total of NV values per thread over the workgroup (<= 1024 threads); sm holds NV * 16 words
template <int NV>
__device__ __forceinline__ void block_scan(const uint32_t (&v)[NV], uint32_t (&excl)[NV], uint32_t (&total)[NV], uint32_t *sm) {
    const int lane = (int) (threadIdx.x & 63u), w = (int) (threadIdx.x >> 6), nw = (int) ((blockDim.x + 63u) >> 6);
    uint32_t inc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        inc[k] = enumk::wave_incl_scan(v[k]);
        if (lane == 63) sm[k * 16 + w] = inc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        uint32_t base = 0, tot = 0;
        for (int i = 0; i < nw; i++) { const uint32_t x = sm[k * 16 + i]; if (i < w) base += x; tot += x; }
        excl[k] = base + inc[k] - v[k];
        total[k] = tot;
    }
    __syncthreads();
}

}  // namespace segsort
}  // namespace mk
