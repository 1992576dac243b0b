/*
 * Microbenchmark (GPU box): what does one dependent 64 B record fetch per lane cost, by access pattern?
 *
 * Every lane walks a random cycle through N 64 B records (the first word of a record names the next), S steps, at the
 * occupancy of the traversal kernel.  Variants:
 *   L1..L4   the lane reads 16, 32, 48 or 64 B of its record with 1..4 buffer_load_dwordx4 (L4 = what k_trace_flat does)
 *   Q        the four lanes of a quad fetch the quad's four records together: instruction j reads record j, each lane a
 *            different 16 B of it, so a quad's access is one contiguous 64 B; the lanes then exchange words with DPP
 *   QT       the same with a complete 4x4 transpose, so that every lane ends up with all 64 B of its own record
 * Prints records per second and CU cycles per record for each.
 *
 * build: hipcc -O3 --offload-arch=gfx950 -o microbench_fetch scripts/microbench_fetch.hip
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


template <int CTRL> __device__ __forceinline__ uint32_t quad_perm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
/* quad_perm control: lane i of a quad reads lane sel[i]; encoded sel0 | sel1<<2 | sel2<<4 | sel3<<6 */
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

template <int MODE, int AUX>   /* MODE 1..4 = loads per lane; 5 = quad-cooperative; 6 = quad-cooperative with full transpose;
                                  AUX = cache-policy bits of the loads (1 sc0, 2 nt, 16 sc1) */
__global__ __launch_bounds__(256) void k_chase(const uint4* records, uint32_t bytes, uint32_t n, uint32_t steps, uint32_t* out)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)records, 0, (int)bytes, 0x00020000);
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = (uint32_t)(((uint64_t)gid * 2654435761u) % n);
    uint32_t acc = 0;
    const uint32_t sub = (threadIdx.x & 3u) * 16u;
    for (uint32_t s = 0; s < steps; s++) {
        if (MODE <= 4) {
            const uint32_t off = idx * 64u;
            u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, AUX), q1 = { 0, 0, 0, 0 }, q2 = q1, q3 = q1;
            if (MODE >= 2) q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, AUX);
            if (MODE >= 3) q2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, AUX);
            if (MODE >= 4) q3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48, 0, AUX);
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            idx = q0.x;
            acc += q1.y + q2.z + q3.w;
        } else {
            /* record indices of the quad's four lanes */
            const uint32_t i0 = quad_perm<QP(0, 0, 0, 0)>(idx), i1 = quad_perm<QP(1, 1, 1, 1)>(idx);
            const uint32_t i2 = quad_perm<QP(2, 2, 2, 2)>(idx), i3 = quad_perm<QP(3, 3, 3, 3)>(idx);
            u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i0 * 64u + sub, 0, AUX);
            u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i1 * 64u + sub, 0, AUX);
            u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i2 * 64u + sub, 0, AUX);
            u32x4 v3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i3 * 64u + sub, 0, AUX);
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (MODE == 5) {
                /* lane j wants word 0 of record j: lane 0's v_j.x */
                const uint32_t a0 = quad_perm<QP(0, 0, 0, 0)>(v0.x), a1 = quad_perm<QP(0, 0, 0, 0)>(v1.x);
                const uint32_t a2 = quad_perm<QP(0, 0, 0, 0)>(v2.x), a3 = quad_perm<QP(0, 0, 0, 0)>(v3.x);
                const uint32_t l = threadIdx.x & 3u;
                idx = l == 0 ? a0 : l == 1 ? a1 : l == 2 ? a2 : a3;
                acc += v0.y + v1.z + v2.w + v3.y;
            } else {
                /* 4x4 transpose of 16 B elements inside the quad: afterwards lane j holds w_k = the k-th 16 B of record j.
                   Two butterfly stages; in each a lane keeps two of its registers and receives two from its partner. */
                const bool odd = (threadIdx.x & 1u) != 0, high = (threadIdx.x & 2u) != 0;
                u32x4 w0, w1, w2, w3;
#define XCHG1(KEEP_E, KEEP_O, SEND_E, SEND_O, OUT_K, OUT_R)                                                       \
                for (int c = 0; c < 4; c++) {                                                                     \
                    const uint32_t give = odd ? SEND_O[c] : SEND_E[c];                                            \
                    const uint32_t got = quad_perm<QP(1, 0, 3, 2)>(give);                                         \
                    OUT_K[c] = odd ? KEEP_O[c] : KEEP_E[c];                                                       \
                    OUT_R[c] = got;                                                                               \
                }
                /* stage 1 (partner = lane^1): even lanes keep v0,v2 and send v1,v3; odd lanes keep v1,v3 and send v0,v2 */
                u32x4 a0, a1, b0, b1;
                XCHG1(v0, v1, v1, v0, a0, a1)      /* even: a0 = own v0, a1 = partner's v0;  odd: a0 = own v1, a1 = partner's v1 */
                XCHG1(v2, v3, v3, v2, b0, b1)
                /* order by source lane: for an even lane own part comes first, for an odd lane the partner's */
                u32x4 p0, p1, r0, r1;
                for (int c = 0; c < 4; c++) { p0[c] = odd ? a1[c] : a0[c]; p1[c] = odd ? a0[c] : a1[c]; r0[c] = odd ? b1[c] : b0[c]; r1[c] = odd ? b0[c] : b1[c]; }
                /* now: p = parts 2*(lane>>1)+{0,1} of record (lane&1) [from v0/v1], r = the same of record 2+(lane&1) */
                /* stage 2 (partner = lane^2): low lanes keep p and send r; high lanes keep r and send p */
                u32x4 g0, g1;
                for (int c = 0; c < 4; c++) {
                    g0[c] = quad_perm<QP(2, 3, 0, 1)>(high ? p0[c] : r0[c]);
                    g1[c] = quad_perm<QP(2, 3, 0, 1)>(high ? p1[c] : r1[c]);
                }
                for (int c = 0; c < 4; c++) {
                    w0[c] = high ? g0[c] : p0[c]; w1[c] = high ? g1[c] : p1[c];
                    w2[c] = high ? r0[c] : g0[c]; w3[c] = high ? r1[c] : g1[c];
                }
#undef XCHG1
                idx = w0.x;
                acc += w1.y + w2.z + w3.w;
            }
        }
    }
    out[gid] = idx + acc;
}

template <int MODE, int AUX = 0> double run(const uint4* d, uint32_t bytes, uint32_t n, uint32_t steps, uint32_t* out, int blocks, uint32_t* check)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_chase<MODE, AUX>), dim3(blocks), dim3(256), 0, 0, d, bytes, n, steps, out);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CHECK(hipMemcpy(check, out, 4, hipMemcpyDeviceToHost));
    return best;
}

int main(int argc, char** argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 11599407u;       /* records: 742 MB, the 5.8 M-triangle scene */
    const uint32_t steps = argc > 2 ? (uint32_t)atol(argv[2]) : 200u;
    const int perCU = argc > 3 ? atoi(argv[3]) : 6;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * perCU;
    std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937_64 rng(7); std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<uint32_t> rec((size_t)n * 16);
    for (uint32_t i = 0; i < n; i++) {       /* one cycle through all records */
        uint32_t* r = &rec[(size_t)perm[i] * 16];
        r[0] = perm[(i + 1) % n];
        for (int k = 1; k < 16; k++) r[k] = perm[i] * 16u + (uint32_t)k;
    }
    uint4* d; uint32_t* out;
    CHECK(hipMalloc((void**)&d, (size_t)n * 64)); CHECK(hipMalloc((void**)&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(d, rec.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    const double lanes = (double)blocks * 256, clk = prop.clockRate * 1e3;
    printf("%u records (%.0f MB), %u steps, %d workgroups of 256 (%d per CU), %d CUs at %.2f GHz\n", n, n * 64.0 / 1e6, steps, blocks, perCU,
           prop.multiProcessorCount, clk / 1e9);
    const char* names[] = { "", "L1 16 B per lane", "L2 32 B per lane", "L3 48 B per lane", "L4 64 B per lane", "Q  quad-cooperative, next index only", "QT quad-cooperative, full transpose" };
    uint32_t chk[7] = { 0 };
    double ms[7];
    ms[1] = run<1>(d, n * 64u, n, steps, out, blocks, &chk[1]);
    ms[2] = run<2>(d, n * 64u, n, steps, out, blocks, &chk[2]);
    ms[3] = run<3>(d, n * 64u, n, steps, out, blocks, &chk[3]);
    ms[4] = run<4>(d, n * 64u, n, steps, out, blocks, &chk[4]);
    ms[5] = run<5>(d, n * 64u, n, steps, out, blocks, &chk[5]);
    ms[6] = run<6>(d, n * 64u, n, steps, out, blocks, &chk[6]);
    for (int m = 1; m <= 6; m++) {
        double recs = lanes * steps / (ms[m] * 1e-3);
        printf("%-40s %8.3f ms  %7.1f G records/s  %5.2f CU cycles per record  (lane 0 -> %u)\n", names[m], ms[m], recs / 1e9,
               prop.multiProcessorCount * clk / recs, chk[m]);
    }
    if (argc > 4) {      /* cache-policy bits on the loads */
        uint32_t c;
#define POL(M, A) { double t = run<M, A>(d, n * 64u, n, steps, out, blocks, &c); printf("mode %d aux %2d: %8.3f ms  %7.1f G records/s\n", M, A, t, lanes * steps / (t * 1e-3) / 1e9); }
        POL(4, 0) POL(4, 1) POL(4, 2) POL(4, 3) POL(4, 16) POL(4, 17) POL(4, 18) POL(4, 19)
        POL(1, 0) POL(1, 1) POL(1, 2) POL(1, 3) POL(1, 16) POL(1, 17) POL(1, 18) POL(1, 19)
        POL(6, 0) POL(6, 1) POL(6, 2) POL(6, 3) POL(6, 16) POL(6, 17) POL(6, 18) POL(6, 19)
    }
    return 0;
}
