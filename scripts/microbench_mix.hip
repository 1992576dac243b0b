/*
 * Microbenchmark (GPU box): do L2-hit and L2-miss record fetches add up or overlap?
 *
 * Every lane fetches one 64 B record per step (four buffer_load_dwordx4, as k_trace_flat does) at an address the
 * previous record named.  A step goes to the HOT set (1 MB: stays in every XCD's L2, far larger than the 32 KB L1)
 * with probability h and otherwise to a record anywhere in the whole table (742 MB: an L2 miss), named by the loaded word.  Which of the
 * two a lane takes is drawn per lane and step, so the lanes of a wavefront mix both kinds in one load instruction, as
 * the traversal kernel's lanes do.  For each h the time is set beside two models built from the pure streams:
 *   sum:  t = n_miss / R_miss + n_hit / R_hit      (the two paths take turns)
 *   max:  t = max(n_miss / R_miss, n_hit / R_hit)  (the two paths overlap completely)
 *
 * build: hipcc -O3 --offload-arch=gfx950 -o microbench_mix scripts/microbench_mix.hip
 */
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k_mix(const uint4* records, uint32_t bytes, uint32_t n, uint32_t steps, uint32_t hotThreshold, uint32_t* out,
                                             unsigned long long* counts)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)records, 0, (int)bytes, 0x00020000);
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = (uint32_t)(((uint64_t)gid * 2654435761u) % n);
    uint32_t acc = 0, nHot = 0;
    for (uint32_t s = 0; s < steps; s++) {
        const uint32_t off = idx * 64u;
        u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
        u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0);
        u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48, 0, 0);
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        const uint32_t rnd = mix32(gid * 977u + s);
        const bool hot = rnd < hotThreshold;                         /* the NEXT record is a hot one */
        /* a cold step lands anywhere in the table: the loaded word (so the step depends on the fetch) scrambled with a
           per-lane, per-step number (or the cold steps that follow a hot record would always land on the same few
           records, which would then sit in the L2 too) */
        const uint32_t cold = __umulhi(mix32(q0.x ^ (rnd * 0x9E3779B9u)), n);
        idx = hot ? q0.y : cold;
        nHot += hot ? 1u : 0u;
        acc += q1.y + q2.z + q3.w;
    }
    out[gid] = idx + acc;
    atomicAdd(counts, (unsigned long long)nHot);
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 11599407u;
    const uint32_t steps = argc > 2 ? (uint32_t)atol(argv[2]) : 200u;
    const int perCU = argc > 3 ? atoi(argv[3]) : 6;
    const uint32_t nHot = 16384;                                       /* 1 MB of records */
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * perCU;
    std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937_64 rng(7); std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<uint32_t> hotSet(perm.begin(), perm.begin() + nHot);   /* a random subset, scattered over the table */
    std::vector<uint32_t> hotNext(n);
    for (uint32_t i = 0; i < n; i++) hotNext[i] = hotSet[rng() % nHot];
    for (uint32_t i = 0; i < nHot; i++) hotNext[hotSet[i]] = hotSet[(i + 1) % nHot];
    std::vector<uint32_t> rec((size_t)n * 16);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t* r = &rec[(size_t)perm[i] * 16];
        r[0] = perm[(i + 1) % n];                                      /* the cycle through everything */
        r[1] = hotNext[perm[i]];                                       /* a record of the hot set */
        for (int k = 2; k < 16; k++) r[k] = perm[i] * 16u + (uint32_t)k;
    }
    uint4* d; uint32_t* out; unsigned long long* counts;
    CHECK(hipMalloc((void**)&d, (size_t)n * 64)); CHECK(hipMalloc((void**)&out, (size_t)blocks * 256 * 4)); CHECK(hipMalloc((void**)&counts, 8));
    CHECK(hipMemcpy(d, rec.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    const double lanes = (double)blocks * 256;
    printf("%u records (%.0f MB), hot set %u records (%.1f MB), %u steps, %d workgroups of 256 (%d per CU)\n", n, n * 64.0 / 1e6, nHot, nHot * 64.0 / 1e6,
           steps, blocks, perCU);
    const double fr[] = { 0.0, 1.0, 0.25, 0.49, 0.66, 0.85 };
    double ms[6], share[6];
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int k = 0; k < 6; k++) {
        const uint32_t thr = fr[k] >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(fr[k] * 4294967296.0);
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(counts, 0, 8));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, d, n * 64u, n, steps, thr, out, counts);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float t; CHECK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
        }
        unsigned long long c; CHECK(hipMemcpy(&c, counts, 8, hipMemcpyDeviceToHost));
        ms[k] = best; share[k] = (double)c / (lanes * steps);
    }
    const double total = lanes * steps;
    const double rMiss = total / (ms[0] * 1e-3), rHit = total / (ms[1] * 1e-3);
    printf("pure miss stream: %.3f ms, %.1f G records/s;  pure hit stream: %.3f ms, %.1f G records/s\n", ms[0], rMiss / 1e9, ms[1], rHit / 1e9);
    printf("%10s %10s %10s %10s %10s\n", "hot share", "measured", "sum model", "max model", "where");
    for (int k = 2; k < 6; k++) {
        const double nh = total * share[k], nm = total - nh;
        const double sum = (nm / rMiss + nh / rHit) * 1e3, mx = std::max(nm / rMiss, nh / rHit) * 1e3;
        printf("%10.3f %9.3f ms %7.3f ms %7.3f ms %9.2f   (0 = max model, 1 = sum model)\n", share[k], ms[k], sum, mx, (ms[k] - mx) / (sum - mx));
    }
    return 0;
}
