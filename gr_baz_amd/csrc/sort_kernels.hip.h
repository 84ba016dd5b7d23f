// sort_kernels.hip.h -- grouping the items of a batch by where their nulls are (round 5; VERDICT r4 task 5).  gfx950 only.
//
// The coarse-gated scan (scan_coarse_kernels.hip.h; the scan when port 2 is not wired, lib/baz_music_doa.cc:97-99) evaluates a 16-item x 16-bin
// tile exactly only where one of the 16 rows can still hold a top-n member.  Items of one stream see one scene: their nulls fall into the same
// tiles and 1.7 % of the tiles are evaluated.  In a batch of UNRELATED items (every item its own emitter angles) a row group's 16 rows have their
// nulls in 16 different places, the tiles that fire are the union -- 22 % -- and the scan costs what the full scan costs.  That rate is a property
// of how the rows are grouped, not of any row: here the items are put in an order in which neighbours share their nulls.
//   1. coarse_key_kernel   one thread per item: d ~ sum_e q_e F_e in float32 at up to 256 sample bins (one per few 16-bin tiles), the two deepest
//                          local minima -> a 16-bit key (lo, hi sample index).  A HEURISTIC: it decides nothing but the order.
//   2. key_sweep_kernel<false>   counting sort over the 16,384 keys, first sweep: the count of every key (no global atomics, see below)
//   3. key_sweep_kernel<true>    second sweep: perm[position] = item
// The scan then takes row x's coefficients from item perm[x] and writes that item's candidates; items stay where they are.  ang / lvl are
// bit-identical to the unsorted scan's (every exact value is computed from its item alone; the gate only decides what need not be computed).
// Three small launches cost ~0.03 ms per 262,144 items -- a loss on a coherent batch -- so the context sorts only while the scan reports a high
// share of fired tiles (baz_music_hip.hip: sort_decide()).
#pragma once

#include "scan_coarse_kernels.hip.h"

namespace bazsort {

constexpr uint32_t KEY_SAMPLES_MAX = 128;
constexpr uint32_t KEY_BUCKETS = KEY_SAMPLES_MAX * KEY_SAMPLES_MAX;

// samples of the key table: one bin per `tile_stride` 16-bin tiles, at most KEY_SAMPLES_MAX
__host__ __device__ inline uint32_t key_tile_stride(uint32_t res)
{
    const uint32_t tiles = (res + 15u) / 16u;
    return (tiles + KEY_SAMPLES_MAX - 1u) / KEY_SAMPLES_MAX;
}
__host__ __device__ inline uint32_t key_samples(uint32_t res)
{
    const uint32_t tiles = (res + 15u) / 16u, ts = key_tile_stride(res);
    return (tiles + ts - 1u) / ts;
}
__host__ __device__ inline uint32_t key_sample_bin(uint32_t j, uint32_t res)
{
    const uint32_t b = j * key_tile_stride(res) * 16u + 8u;
    return b < res ? b : res - 1u;
}

template <int M>
__global__ __launch_bounds__(256) void coarse_key_kernel(const double* __restrict__ Qs, const float* __restrict__ KT, uint32_t nsamp,
                                                         uint32_t batch, uint32_t qstride, uint16_t* __restrict__ keys)
{
    constexpr int MM = M * M;
    const uint32_t it = blockIdx.x * 256u + threadIdx.x;
    const uint32_t itc = it < batch ? it : batch - 1u;
    float q[MM];
#pragma unroll
    for (int e = 0; e < MM; ++e) q[e] = (float)Qs[(size_t)e * qstride + itc];
    // one walk over the samples: the two deepest LOCAL minima (a local minimum = below its left neighbour, not above its right one)
    float d1 = __builtin_inff(), d2 = __builtin_inff();              // d1 <= d2: the two deepest so far
    uint32_t j1 = 0, j2 = 0;
    float pp = __builtin_inff(), pv = __builtin_inff();              // d at j - 2 and j - 1
    auto offer = [&](const float v, const uint32_t at) {
        const bool first = v < d1, second = !first && (v < d2);
        d2 = first ? d1 : (second ? v : d2);
        j2 = first ? j1 : (second ? at : j2);
        d1 = first ? v : d1;
        j1 = first ? at : j1;
    };
    for (uint32_t j = 0; j < nsamp; ++j) {
        const float* __restrict__ f = KT + (size_t)j * MM;           // wave-uniform address: scalar loads
        float d = 0.0f;
#pragma unroll
        for (int e = 0; e < MM; ++e) d = __builtin_fmaf(q[e], f[e], d);
        if (j > 0 && pv < pp && pv <= d) offer(pv, j - 1u);          // (all comparisons false for NaN: such an item keeps key 0)
        pp = pv;
        pv = d;
    }
    if (nsamp > 0 && pv < pp) offer(pv, nsamp - 1u);                 // the last sample
    if (!(d2 < __builtin_inff())) j2 = j1;                           // one minimum only
    const uint32_t lo = j1 < j2 ? j1 : j2, hi = j1 < j2 ? j2 : j1;
    const uint32_t key = (lo % KEY_SAMPLES_MAX) * KEY_SAMPLES_MAX + (hi % KEY_SAMPLES_MAX);
    keys[it] = (it < batch) ? (uint16_t)key : (uint16_t)0xFFFFu;     // (the key array is padded to whole 8-key loads: 0xFFFF is no key)
}

// Counting sort WITHOUT global atomics (262,144 of them ran at ~1.3 per clock chip-wide: 95 + 64 us for the two kernels that used them).
// KEY_BLOCKS workgroups, one per range of KEY_RANGE consecutive keys; every workgroup sweeps ALL keys (2 bytes per item: the array stays in
// L2) and handles the ones of its range with LDS atomics.
//   key_count_kernel   counts[key] and totals[workgroup]
//   key_place_kernel   base of the workgroup's range = sum of the totals before it; positions inside the range by an LDS cursor per key
constexpr uint32_t KEY_RANGE = 64;
constexpr uint32_t KEY_BLOCKS = KEY_BUCKETS / KEY_RANGE;             // 256: one workgroup per CU

template <bool PLACE>
__global__ __launch_bounds__(256) void key_sweep_kernel(const uint16_t* __restrict__ keys, uint32_t nkeys8, uint32_t* __restrict__ counts,
                                                        uint32_t* __restrict__ totals, uint32_t* __restrict__ perm)
{
    __shared__ uint32_t slot[KEY_RANGE];                             // PLACE: the next position of every key of the range; else its count
    __shared__ uint32_t part[256];
    const uint32_t t = threadIdx.x, lo = blockIdx.x * KEY_RANGE;
    if constexpr (PLACE) {
        uint32_t s = 0;                                              // items in front of this workgroup's range
        for (uint32_t b = t; b < blockIdx.x; b += 256u) s += totals[b];
        part[t] = s;
        __syncthreads();
        if (t < KEY_RANGE) {
            uint32_t base = 0;
            for (uint32_t i = 0; i < 256u; ++i) base += part[i];
            for (uint32_t k = 0; k < t; ++k) base += counts[lo + k];
            slot[t] = base;
        }
    } else {
        if (t < KEY_RANGE) slot[t] = 0u;
    }
    __syncthreads();
    const uint4* __restrict__ k8 = reinterpret_cast<const uint4*>(keys);
    for (uint32_t i = t; i < nkeys8; i += 256u) {
        const uint4 v = k8[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const uint32_t key = (w[h >> 1] >> (16 * (h & 1))) & 0xFFFFu;
            const uint32_t rel = key - lo;
            if (rel < KEY_RANGE) {
                const uint32_t pos = atomicAdd(&slot[rel], 1u);
                if constexpr (PLACE) perm[pos] = i * 8u + (uint32_t)h;
            }
        }
    }
    if constexpr (!PLACE) {
        __syncthreads();
        if (t < KEY_RANGE) counts[lo + t] = slot[t];
        if (t == 0) {
            uint32_t s = 0;
            for (uint32_t k = 0; k < KEY_RANGE; ++k) s += slot[k];
            totals[blockIdx.x] = s;
        }
    }
}

// the key table: F at the sample bins, float32, [sample][e]
__global__ __launch_bounds__(256) void build_key_table_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, uint32_t nsamp,
                                                              float* __restrict__ KT)
{
    const uint32_t mm = m * m;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= nsamp * mm) return;
    const uint32_t j = idx / mm, e = idx - j * mm;
    const uint32_t bin = key_sample_bin(j, res);
    const float* a = tab + (size_t)bin * m * 2;
    // (tab_F of table_kernels.hip.h, restated: this header sits below it)
    const uint32_t r = e / m, c = e - r * m;
    double v;
    if (r == c) v = (double)a[2 * r] * a[2 * r] + (double)a[2 * r + 1] * a[2 * r + 1];
    else if (r < c) v = (double)a[2 * r] * a[2 * c] + (double)a[2 * r + 1] * a[2 * c + 1];
    else v = (double)a[2 * c] * a[2 * r + 1] - (double)a[2 * c + 1] * a[2 * r];
    KT[idx] = (float)v;
}

}  // namespace bazsort
