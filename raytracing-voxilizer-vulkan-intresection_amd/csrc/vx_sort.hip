// vx_sort.hip -- device sort of the octree's 64-bit Morton items (replaces std::sort(par_unseq), octTree.hpp:363).
//
// Hand-written LSD radix sort for gfx950, keys only (duplicates are identical values, so any stable or unstable order of equal keys
// is the same array; the passes themselves must be stable).  Only the `bits` low bits that can be set are sorted
// (3 * bitsPerAxis: 33 at 2048^3), in ceil(bits / 9) passes of 8-9 bit digits (four passes for 33 bits).
//
// One pass = three launches:
//   k_sort_hist     every workgroup counts the digits of its tile of 4096 keys in LDS and writes the counts DIGIT-MAJOR
//                   (H[digit * tiles + tile]): one exclusive scan of H (the library's single-pass scan) then yields, for every
//                   (digit, tile), where that tile's keys of that digit start in the output;
//   scan            over digits * tiles counters (4.7M for the 75M items of BASELINE configs[4]);
//   k_sort_scatter  re-reads the tile (wave w owns 1024 consecutive keys, 64 per round: 512-byte loads), ranks every key among the
//                   equal digits before it -- per wave and round by a ballot per digit bit (the lanes that share my digit), carried
//                   across rounds in the wave's own LDS counters (no atomics: one wave, one counter row), across waves by a prefix
//                   over the four rows -- orders the tile by digit in LDS and stores it from there: a run of equal digits leaves as
//                   one contiguous piece at (start of my tile's digit in the output).
// Memory order inside a tile is (wave, round, lane) and every rank is taken in that order, so each pass is stable.
// Traffic per pass: keys read twice and written once (24 B per key) + the counters; what bounds it is HBM.
#include "vx_internal.h"

#include <cstring>

namespace vx {

#define VX_KL(kern, grid, block, shmem, stream, ...)                         \
    do {                                                                     \
        ProfScope ps_(#kern, stream);                                        \
        hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);   \
    } while (0)

namespace {

constexpr uint32_t kSortThreads = 256;
constexpr uint32_t kSortWaves = kSortThreads / 64;
constexpr uint32_t kSortRounds = 16;                               // keys per lane and tile
constexpr uint32_t kSortTile = kSortThreads * kSortRounds;         // 4096 keys
constexpr uint32_t kSortMaxDigitBits = 9;
constexpr uint32_t kSortMaxDigits = 1u << kSortMaxDigitBits;

__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint64_t* __restrict__ keys, uint64_t n, uint32_t shift, uint32_t nd /*digits: 2^bits of this pass*/,
                                                           uint32_t ntiles, uint32_t* __restrict__ H)
{
    __shared__ uint32_t cnt[kSortMaxDigits];
    for (uint32_t d = threadIdx.x; d < nd; d += kSortThreads) cnt[d] = 0u;
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * kSortTile;
    const uint32_t mask = nd - 1u;
    // (the histogram does not care about the order inside the tile: two keys per 16-byte load, all eight loads of a thread in flight)
    if (t0 + kSortTile <= n) {
        ulonglong2 v[kSortRounds / 2];
#pragma unroll
        for (uint32_t k = 0; k < kSortRounds / 2; ++k) v[k] = reinterpret_cast<const ulonglong2*>(keys + t0)[k * kSortThreads + threadIdx.x];
#pragma unroll
        for (uint32_t k = 0; k < kSortRounds / 2; ++k) {
            atomicAdd(&cnt[(uint32_t)(v[k].x >> shift) & mask], 1u);
            atomicAdd(&cnt[(uint32_t)(v[k].y >> shift) & mask], 1u);
        }
    } else {
        for (uint32_t k = 0; k < kSortRounds; ++k) {
            const uint64_t i = t0 + (uint64_t)k * kSortThreads + threadIdx.x;
            if (i < n) atomicAdd(&cnt[(uint32_t)(keys[i] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nd; d += kSortThreads) H[(uint64_t)d * ntiles + blockIdx.x] = cnt[d];
}

template <int NBITS>
__device__ __forceinline__ unsigned long long same_digit_lanes(uint32_t d, bool valid)
{
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < NBITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(valid && bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

template <int NBITS>
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint64_t* __restrict__ keys, uint64_t* __restrict__ out, uint64_t n, uint32_t shift,
                                                              uint32_t ntiles, const uint32_t* __restrict__ Hs /*exclusive scan of H*/)
{
    constexpr uint32_t nd = 1u << NBITS;
    __shared__ uint32_t wcnt[kSortWaves][nd];   // per wave: keys of the digit seen so far; afterwards: the earlier waves' total of the digit
    __shared__ uint32_t dbase[nd];              // where this tile's keys of the digit start in the output
    __shared__ uint32_t dstart[nd + 1];         // where they start inside the tile once it is ordered by digit
    __shared__ uint64_t skey[kSortTile];        // the tile ordered by digit: runs of equal digits leave as contiguous stores
    __shared__ uint32_t wtot[kSortWaves];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < kSortWaves * nd; i += kSortThreads) (&wcnt[0][0])[i] = 0u;
    for (uint32_t d = threadIdx.x; d < nd; d += kSortThreads) dbase[d] = Hs[(uint64_t)d * ntiles + blockIdx.x];
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * kSortTile;
    const uint64_t w0 = t0 + (uint64_t)wv * (64u * kSortRounds);
    const uint32_t nvalid = n - t0 < kSortTile ? (uint32_t)(n - t0) : kSortTile;  // keys of this tile (the last one may be short)
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint64_t key[kSortRounds];
    uint32_t rnk[kSortRounds];   // rank among the wave's keys of the same digit (rounds before + lanes before)
#pragma unroll
    for (uint32_t k = 0; k < kSortRounds; ++k) {
        const uint64_t i = w0 + (uint64_t)k * 64u + lane;
        key[k] = i < n ? keys[i] : 0ull;
    }
#pragma unroll
    for (uint32_t k = 0; k < kSortRounds; ++k) {
        const uint64_t i = w0 + (uint64_t)k * 64u + lane;
        const bool valid = i < n;
        const uint32_t d = (uint32_t)(key[k] >> shift) & (nd - 1u);
        const unsigned long long m = same_digit_lanes<NBITS>(d, valid);
        const uint32_t before = (uint32_t)__popcll(m & lt);
        uint32_t c = 0u;
        if (valid && before == 0u) {  // the first lane of every digit group carries the wave's counter of that digit forward
            c = wcnt[wv][d];
            wcnt[wv][d] = c + (uint32_t)__popcll(m);
        }
        c = __shfl(c, valid ? __ffsll((long long)m) - 1 : 0, 64);
        rnk[k] = c + before;
    }
    __syncthreads();
    // wcnt[w][d] = the wave's total of digit d  ->  the total of the waves before it; dstart = exclusive scan of the tile's digit totals
    {
        constexpr uint32_t per = (nd + kSortThreads - 1) / kSortThreads;  // digits per thread (1 or 2), consecutive
        uint32_t tot[per], mine = 0u;
#pragma unroll
        for (uint32_t j = 0; j < per; ++j) {
            const uint32_t d = threadIdx.x * per + j;
            uint32_t run = 0u;
            if (d < nd) {
#pragma unroll
                for (uint32_t w = 0; w < kSortWaves; ++w) {
                    const uint32_t c = wcnt[w][d];
                    wcnt[w][d] = run;
                    run += c;
                }
            }
            tot[j] = run;
            mine += run;
        }
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off, 64);
            if ((int)lane >= off) inc += o;
        }
        if (lane == 63u) wtot[wv] = inc;
        __syncthreads();
        uint32_t pre = inc - mine;
#pragma unroll
        for (uint32_t w = 0; w < kSortWaves; ++w)
            if (w < wv) pre += wtot[w];
#pragma unroll
        for (uint32_t j = 0; j < per; ++j) {
            const uint32_t d = threadIdx.x * per + j;
            if (d < nd) dstart[d] = pre;
            pre += tot[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kSortRounds; ++k) {
        const uint64_t i = w0 + (uint64_t)k * 64u + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[k] >> shift) & (nd - 1u);
            skey[dstart[d] + wcnt[wv][d] + rnk[k]] = key[k];
        }
    }
    __syncthreads();
    // consecutive threads, consecutive keys of the digit-ordered tile: within a run of equal digits consecutive output addresses
#pragma unroll
    for (uint32_t k = 0; k < kSortRounds; ++k) {
        const uint32_t q = k * kSortThreads + threadIdx.x;
        if (q < nvalid) {
            const uint64_t kk = skey[q];
            const uint32_t d = (uint32_t)(kk >> shift) & (nd - 1u);
            out[(uint64_t)dbase[d] + (q - dstart[d])] = kk;
        }
    }
}

inline uint32_t sort_tiles(uint64_t n) { return (uint32_t)((n + kSortTile - 1) / kSortTile); }

}  // namespace

// scratch: the counters H and their exclusive scan (digits * tiles + 1 each) and the scan's own state words
size_t sort_tmp_bytes(uint64_t n)
{
    const uint64_t cells = (uint64_t)kSortMaxDigits * sort_tiles(n) + 16;
    return (size_t)(cells * 4 * 2 + scan_tmp_bytes(cells) + 256);
}

// Sorts n keys whose set bits lie below `bits`.  Buffers ping-pong; returns which one holds the result: 0 = keys_a, 1 = keys_b
// (the other one is scratch afterwards).  n < 2^32 (positions are 32-bit).
int launch_sort_u64(uint64_t* keys_a, uint64_t* keys_b, uint64_t n, int bits, void* tmp, size_t tmp_bytes, hipStream_t s)
{
    (void)tmp_bytes;
    if (!n) return 0;
    if (bits < 1) bits = 1;
    if (bits > 64) bits = 64;
    const int passes = (bits + (int)kSortMaxDigitBits - 1) / (int)kSortMaxDigitBits;
    const uint32_t ntiles = sort_tiles(n);
    const uint64_t cells_max = (uint64_t)kSortMaxDigits * ntiles + 16;
    uint32_t* H = reinterpret_cast<uint32_t*>(tmp);
    uint32_t* Hs = H + cells_max;
    void* scan_tmp = reinterpret_cast<void*>(Hs + cells_max);
    (void)hipMemsetAsync(scan_tmp, 0, scan_tmp_bytes(cells_max), s);  // (the scan leaves its state zero again after every run)
    uint64_t* src = keys_a;
    uint64_t* dst = keys_b;
    int done = 0;
    for (int p = 0; p < passes; ++p) {
        // spread the bits evenly: 33 bits = 9 + 8 + 8 + 8
        const int left = bits - done, nb = (left + (passes - p) - 1) / (passes - p);
        const uint32_t nd = 1u << nb;
        const uint64_t cells = (uint64_t)nd * ntiles;
        VX_KL(k_sort_hist, dim3(ntiles), dim3(kSortThreads), 0, s, (const uint64_t*)src, n, (uint32_t)done, nd, ntiles, H);
        (void)launch_scan_u32(H, Hs, cells, false, scan_tmp, nullptr, s, /*tmp_is_zero=*/true);
        switch (nb) {
#define VX_SORT_CASE(B) case B: VX_KL((k_sort_scatter<B>), dim3(ntiles), dim3(kSortThreads), 0, s, (const uint64_t*)src, dst, n, (uint32_t)done, ntiles, (const uint32_t*)Hs); break;
            VX_SORT_CASE(1) VX_SORT_CASE(2) VX_SORT_CASE(3) VX_SORT_CASE(4) VX_SORT_CASE(5) VX_SORT_CASE(6) VX_SORT_CASE(7) VX_SORT_CASE(8) VX_SORT_CASE(9)
#undef VX_SORT_CASE
        }
        done += nb;
        uint64_t* t = src; src = dst; dst = t;
    }
    return src == keys_a ? 0 : 1;
}

}  // namespace vx
