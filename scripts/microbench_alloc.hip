/*
 * Microbenchmark (GPU box): does the kind of device allocation change what an L2-missing 64 B record fetch costs?
 * Every lane walks a random cycle through N 64 B records (the first word of a record names the next), four
 * buffer_load_dwordx4 per record as k_trace_flat does, with the table in
 *   0  hipMalloc memory (what the library uses),
 *   1  hipExtMallocWithFlags(hipDeviceMallocUncached),
 *   2  hipExtMallocWithFlags(hipDeviceMallocFinegrained),
 * and each cache-policy combination on the loads (aux: 1 sc0, 2 nt, 16 sc1).  Prints records per second.
 * The question behind it: every L2 miss of the traversal kernel moves a 128 B line for a 64 B record; if some allocation
 * kind made the fabric request 64 B, cold records (leaves, the deepest interior levels) could live there.
 *
 * build: hipcc -O3 --offload-arch=gfx950 -o scripts/bin/microbench_alloc scripts/microbench_alloc.hip
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

template <int AUX>
__global__ __launch_bounds__(256) void k_chase(const uint4* records, uint32_t bytes, uint32_t n, uint32_t steps, uint32_t* out)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)records, 0, (int)bytes, 0x00020000);
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = (uint32_t)(((uint64_t)gid * 2654435761u) % n), acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        const uint32_t off = idx * 64u;
        u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, AUX);
        u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, AUX);
        u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, AUX);
        u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48, 0, AUX);
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        idx = q0.x; acc += q1.y + q2.z + q3.w;
    }
    out[gid] = idx + acc;
}
template <int AUX> double run(const uint4* d, uint32_t bytes, uint32_t n, uint32_t steps, uint32_t* out, int blocks)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_chase<AUX>), dim3(blocks), dim3(256), 0, 0, d, bytes, n, steps, out);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}
int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 11599407u;
    const uint32_t steps = argc > 2 ? (uint32_t)atol(argv[2]) : 100u;
    const int perCU = argc > 3 ? atoi(argv[3]) : 7;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * perCU;
    std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937_64 rng(7); std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<uint32_t> rec((size_t)n * 16);
    for (uint32_t i = 0; i < n; i++) { uint32_t* r = &rec[(size_t)perm[i] * 16]; r[0] = perm[(i + 1) % n]; for (int k = 1; k < 16; k++) r[k] = perm[i] * 16u + (uint32_t)k; }
    uint32_t* out; CHECK(hipMalloc((void**)&out, (size_t)blocks * 256 * 4));
    const double lanes = (double)blocks * 256;
    const char* kinds[] = { "hipMalloc", "hipDeviceMallocUncached", "hipDeviceMallocFinegrained" };
    for (int kind = 0; kind < 3; kind++) {
        uint4* d = nullptr;
        hipError_t e = kind == 0 ? hipMalloc((void**)&d, (size_t)n * 64)
                                 : hipExtMallocWithFlags((void**)&d, (size_t)n * 64, kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("%s: allocation refused (%s)\n", kinds[kind], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CHECK(hipMemcpy(d, rec.data(), (size_t)n * 64, hipMemcpyHostToDevice));
#define POL(A) { double t = run<A>(d, n * 64u, n, steps, out, blocks); printf("%-28s aux %2d: %8.3f ms  %7.1f G records/s\n", kinds[kind], A, t, lanes * steps / (t * 1e-3) / 1e9); fflush(stdout); }
        POL(0) POL(2) POL(16) POL(17) POL(19)
        CHECK(hipFree(d));
    }
    return 0;
}
