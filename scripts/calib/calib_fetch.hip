// Calibration of rocprofv3 FETCH_SIZE for the traversal kernel's access pattern: every lane reads one
// random 64 B record with four 16 B buffer loads.  Three patterns over one 4 GiB buffer, each byte
// read exactly once:  0 = streaming (lane i reads record i), 1 = random single records (64 B),
// 2 = random pairs (lanes 2k, 2k+1 read the two halves of one random 128 B line).
// usage: calib_fetch <pattern> [log2 records]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_gather(const void* buf, unsigned long long bytes, unsigned logn, int pattern, unsigned* out)
{
    const unsigned n = 1u << logn;
    unsigned acc = 0;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        unsigned idx;
        if (pattern == 3) {
            // quad-cooperative: in turn k the four lanes of a quad read the four 16 B pieces of ONE random record
            const unsigned quadBase = i & ~3u, piece = (i & 3u) * 16u;
            for (int k = 0; k < 4; k++) {
                unsigned id2 = ((quadBase + k) * 2654435761u) & (n - 1);
                const char* base2 = (const char*)buf + ((id2 >> 25) ? (1ull << 31) : 0ull);
                __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)base2, 0, 0x7FFFFFFF, 0x00020000);
                u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r2, ((id2 & ((1u << 25) - 1)) << 6) + piece, 0, 0);
                asm volatile("" : "+v"(q));
                acc += q.x ^ q.w;
            }
            continue;
        }
        if (pattern == 0) idx = i;
        else if (pattern == 1) idx = (i * 2654435761u) & (n - 1);                 // odd multiplier: a bijection on 2^logn
        else idx = ((((i >> 1) * 2654435761u) & ((n >> 1) - 1)) << 1) | (i & 1);  // random 128 B line, both halves by a lane pair
        // two descriptors of 2 GiB each keep offsets within 32 bits
        const char* base = (const char*)buf + ((idx >> 25) ? (1ull << 31) : 0ull);
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
        unsigned off = (idx & ((1u << 25) - 1)) << 6;
        u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0);
        u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32, 0, 0);
        u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 48, 0, 0);
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        acc += q0.x ^ q1.y ^ q2.z ^ q3.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv)
{
    int pattern = argc > 1 ? atoi(argv[1]) : 1;
    unsigned logn = argc > 2 ? atoi(argv[2]) : 26;
    unsigned long long bytes = 64ull << logn;
    void* buf; unsigned* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int reps = argc > 3 ? atoi(argv[3]) : 3;
    for (int rep = 0; rep < reps; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather, dim3(256 * 8), dim3(256), 0, 0, buf, bytes, logn, pattern, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("pattern %d: %.3f GB useful in %.3f ms = %.2f TB/s\n", pattern, bytes / 1e9, ms, bytes / ms / 1e9);
    }
    return 0;
}
