/*
 * sort_kernels.h — the order in which a batch of rays is traced: by the cell of the scene box each origin lies in, so
 * that the wavefronts that run side by side walk neighbouring parts of the tree and share its lines in the caches.
 * The rays stay where they are; the kernels produce order[k] = index of the k-th ray to trace, and the traversal
 * kernel fetches its rays through it (moving the rays into order as well costs 2 ms more per 2^26 rays than the
 * scattered fetch does).  A ray's result does not depend on when it is traced and is stored under the ray's own
 * index, so the order only changes speed.
 *
 * One counting-sort pass over a grid of 2^bits cells per axis (bits <= 4, Morton key of 3*bits bits):
 *   k_bin_count    per tile of rays: key of every ray (kept for the scatter pass), tile histogram in LDS, one global
 *                  atomic per non-empty bin of the tile
 *   k_bin_scan     exclusive prefix sum of the 4096 bin counts (one workgroup)
 *   k_bin_scatter  per tile: reserves the tile's slice of every bin with one global atomic, ranks the tile's rays
 *                  inside the slice with LDS atomics, writes each ray's index to its place in the order
 * The order of rays inside a bin depends on the order of atomics and differs from run to run; nothing observable does.
 */
#pragma once
#include "../../include/ptcore.h"

#define PT_SORT_BINS 4096
#define PT_SORT_TILE 32768       /* 16384 and 49152 measure the same, 8192 is slower */
#define PT_SORT_BLOCK 1024

__device__ __forceinline__ uint32_t spread4(uint32_t v)      /* bits 0..3 of v to bits 0,3,6,9 */
{
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
__device__ __forceinline__ uint32_t ray_bin(const float4 o, float lox, float loy, float loz, float sx, float sy, float sz, float top)
{
    /* NaN and out-of-box origins clamp into the grid: any bin is a valid place for a ray */
    int cx = (int)fminf(fmaxf((o.x - lox) * sx, 0.0f), top);
    int cy = (int)fminf(fmaxf((o.y - loy) * sy, 0.0f), top);
    int cz = (int)fminf(fmaxf((o.z - loz) * sz, 0.0f), top);
    return spread4((uint32_t)cx) | (spread4((uint32_t)cy) << 1) | (spread4((uint32_t)cz) << 2);
}

__global__ __launch_bounds__(PT_SORT_BLOCK) void k_bin_count(const PtRay* rays, uint64_t n, uint16_t* keys, uint32_t* binCount, int nbins,
                                                            float lox, float loy, float loz, float sx, float sy, float sz, float top)
{
    __shared__ uint32_t hist[PT_SORT_BINS];
    for (int b = threadIdx.x; b < nbins; b += PT_SORT_BLOCK) hist[b] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * PT_SORT_TILE;
    for (uint32_t k = threadIdx.x; k < PT_SORT_TILE; k += PT_SORT_BLOCK) {
        const uint64_t i = base + k;
        if (i < n) {
            const uint32_t bin = ray_bin(*(const float4*)(rays + i), lox, loy, loz, sx, sy, sz, top);
            keys[i] = (uint16_t)bin;
            atomicAdd(&hist[bin], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += PT_SORT_BLOCK) { uint32_t c = hist[b]; if (c) atomicAdd(binCount + b, c); }
}

/* binStart[b] = number of rays in bins < b; cursor starts as a copy; binCount is cleared for the next batch. */
__global__ __launch_bounds__(PT_SORT_BLOCK) void k_bin_scan(uint32_t* binCount, uint32_t* cursor)
{
    __shared__ uint32_t part[PT_SORT_BLOCK];
    const int t = threadIdx.x, per = PT_SORT_BINS / PT_SORT_BLOCK;
    uint32_t c[per], s = 0;
    for (int k = 0; k < per; k++) { c[k] = binCount[t * per + k]; s += c[k]; }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < PT_SORT_BLOCK; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (int k = 0; k < per; k++) { cursor[t * per + k] = run; run += c[k]; binCount[t * per + k] = 0; }
}

__device__ __forceinline__ void bin_scatter_tile(uint64_t n, const uint16_t* keys, uint32_t* cursor, uint32_t* order)
{
    /* The tile is first ordered inside LDS, then written out in that order: a tile's rays of one bin are one run in the
       output, so lanes next to each other write indices next to each other. */
    __shared__ uint32_t hist[PT_SORT_BINS];       /* tile count per bin, then the next free place of the bin in tile order */
    __shared__ uint32_t gbase[PT_SORT_BINS];      /* place of the tile's slice of the bin in the output, minus the bin's start in tile order */
    __shared__ uint16_t local[PT_SORT_TILE];      /* local[place in tile order] = index in the tile */
    __shared__ uint32_t part[PT_SORT_BLOCK];
    const int t = threadIdx.x, per = PT_SORT_BINS / PT_SORT_BLOCK;
    for (int b = t; b < PT_SORT_BINS; b += PT_SORT_BLOCK) hist[b] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * PT_SORT_TILE;
    if (base >= n) return;                                /* the same for the whole workgroup */
    const uint32_t count = (uint32_t)((n - base < PT_SORT_TILE) ? n - base : PT_SORT_TILE);
    /* The thread's keys are fetched once, all loads in flight together, and kept in registers for both passes over them:
       a loop that feeds each loaded key to an LDS atomic pays the memory latency turn by turn, and with one workgroup per
       CU (the LDS it needs) there is nobody else to hide it (0.54 -> 0.42 ms for 2^26 rays). */
    constexpr int KEYS = PT_SORT_TILE / PT_SORT_BLOCK;
    constexpr uint32_t NO_KEY = 0xFFFFu;                  /* keys are below PT_SORT_BINS */
    uint32_t key[KEYS];
#pragma unroll
    for (int j = 0; j < KEYS; j++) {
        const uint32_t k = (uint32_t)t + (uint32_t)j * PT_SORT_BLOCK;
        key[j] = k < count ? (uint32_t)keys[base + k] : NO_KEY;
    }
#pragma unroll
    for (int j = 0; j < KEYS; j++) if (key[j] != NO_KEY) atomicAdd(&hist[key[j]], 1u);
    __syncthreads();
    /* exclusive scan of the tile histogram: four consecutive bins per thread */
    uint32_t c[per], s = 0;
    for (int k = 0; k < per; k++) { c[k] = hist[t * per + k]; s += c[k]; }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < PT_SORT_BLOCK; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (int k = 0; k < per; k++) {
        const int b = t * per + k;
        if (c[k]) gbase[b] = atomicAdd(cursor + b, c[k]) - run;
        hist[b] = run;
        run += c[k];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KEYS; j++)
        if (key[j] != NO_KEY) local[atomicAdd(&hist[key[j]], 1u)] = (uint16_t)((uint32_t)t + (uint32_t)j * PT_SORT_BLOCK);
    __syncthreads();
#pragma unroll 8
    for (uint32_t pos = t; pos < count; pos += PT_SORT_BLOCK) {
        const uint32_t k = local[pos];
        order[gbase[keys[base + k]] + pos] = (uint32_t)(base + k);
    }
}
__global__ __launch_bounds__(PT_SORT_BLOCK) void k_bin_scatter(uint64_t n, const uint16_t* keys, uint32_t* cursor, int nbins, uint32_t* order)
{
    bin_scatter_tile(n, keys, cursor, order);
    (void)nbins;
}
