/*
 * frame_kernels.h — the wavefront stages around traversal: ray generation, shading with
 * next-event estimation, job compaction, path resolve + accumulate.
 *
 * The reference fuses closest-hit, shading, the shadow-ray walk and the spawn of the next ray in
 * one kernel (reference trace.comp:165-233).  Here they are separate passes over dense queues:
 *   k_launch            -> slot 0 of every pixel + one job per pixel          (launch.comp:28-51)
 *   k_trace_*<JOBS>     -> one PtHitRec per job                               (trace_kernels.h)
 *   k_shade             -> slot fields, the next slot, one PtShadowRay per hit (trace.comp:58-233)
 *   k_trace_*<SHADOW>   -> clears wIn of slots whose light sample is occluded (trace.comp:123)
 *   k_compact           -> dense list of the jobs still alive                 (raysort.comp:23-55)
 *   k_resolve           -> radiance per pixel, += accumulation                (composite.frag:30-55)
 * All shading arithmetic is fp16 as in the reference, one rounding per operation.
 */
#pragma once
#include "../../include/ptcore.h"
#include "device_math.h"
#include "trace_kernels.h"

#pragma clang fp contract(off)

#define PT_MAX_TRAIN_DEPTH 64     /* deepest path (slots per pixel) the frame train keeps per-depth ray counts for */

enum {
    CTR_NJOBS = 0,        /* entries of the current job list */
    CTR_FETCH_CLOSEST = 1,
    CTR_COMPACT_OUT = 2,
    CTR_LIVE = 3,         /* live jobs seen by the shade pass = closest-hit rays traced */
    CTR_SHADOW = 4        /* then two words per depth: shadow rays queued by that depth's shade pass, and the shadow
                             pass's fetch counter.  Per depth, so that nothing has to be cleared while a shadow pass
                             of an earlier depth may still be running on the second stream. */
};
/* after the shadow words, two more per depth for the certified walk: closest-hit rays listed for a second walk, and that
   walk's fetch counter */
__host__ __device__ constexpr uint32_t ctr_flag_base(uint32_t slots) { return CTR_SHADOW + 2u * slots; }
__host__ __device__ constexpr uint32_t ctr_words(uint32_t slots) { return CTR_SHADOW + 4u * slots; }

/* Counters of the frame train (several frames, each at its own depth, stepped together): the first three are the words the
   shade pass already uses under the names above. */
enum {
    TC_NJOBS = 0,         /* entries of the step's job list: the survivors of the last step's shade pass + the pixels of a frame that enters */
    TC_FETCH = 1,
    TC_OUT = 2,           /* entries the shade pass has appended to the next step's list */
    TC_BASE = 3,          /* where the launch pass puts the new frame's jobs: behind the survivors */
    TC_NSHADOW = 4,       /* two words: lengths of the two shadow queues (one is traced while the other is filled) */
    TC_FLAG = 6,          /* certified walk: jobs listed for a second walk, that walk's fetch counter */
    TC_WORDS = 16
};

struct FrameArgs {
    PtShaderConstants c;
    uint32_t W, H, slots;
    uint32_t zeroNextSlot, reseed;
    uint32_t seedFromJitter;         /* launch: the accumulation holds no sample yet (launch.comp:49-50) */
    PtRayStackSlot* stack;
    const PtRayStackSlot* rngStack;  /* launch: the stack whose slot 0 holds the RNG state the last frame left (with several
                                        frames in flight that is another frame's stack) */
    uint32_t* jobs;
    uint32_t* jobsOut;
    uint32_t* ctr;
    uint32_t* shadowCtr;             /* = ctr + CTR_SHADOW + 2*depth */
    const PtHitRec* hitRecs;
    PtShadowRay* shadowRays;         /* the shadow queue this depth appends to */
    uint32_t fuseCompact;            /* shade appends surviving jobs to jobsOut instead of rewriting jobs in place */
    uint32_t tileShadow;             /* shade puts the shadow ray of pixel (x, y) at queue place tile * 64 + lane of the 8x8 tile
                                        it lies in (lane = (y % 8) * 8 + x % 8) and marks the places of pixels without one: the
                                        voting shadow walk of the SPECULATIVE build takes one tile per wavefront */
    const PtVertexAux* aux;
    const PtLight* lights;
    const uint16_t* blueNoise;
    float4* accumulation;
    uint2* current;                  /* RGBA16F */
    unsigned long long* rayCounts;   /* [2*depth] closest rays, [2*depth+1] shadow rays */
    /* frame train */
    uint32_t slotBase;               /* launch: index of the frame's first slot in the ring of ray stacks (jobs carry ring-wide slot indices;
                                        `stack` is the ring's base for the shade pass, the frame's own stack for launch and resolve) */
    uint32_t row0, row1;             /* launch, resolve: the rows of the image this part of the train renders ([0, H) elsewhere) */
    uint32_t shadowQueue;            /* step end: the queue this step's traversal launch emptied */
};

__device__ __forceinline__ void mat_mul_vec(const float* m, const float v[4], float out[4])
{
    for (int i = 0; i < 4; i++) out[i] = ((m[0 + i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2]) + m[12 + i] * v[3];
}

/* launch.comp:28-51 */
template <bool TRAIN>
__global__ __launch_bounds__(256) void k_launch_t(FrameArgs f)
{
    const uint32_t gx = blockIdx.x * 32 + (threadIdx.x & 31), gy = (TRAIN ? f.row0 : 0u) + blockIdx.y * 8 + (threadIdx.x >> 5);
    const uint32_t lx = gx + (uint32_t)f.c.viewportBase[0], ly = gy + (uint32_t)f.c.viewportBase[1];
    if (!TRAIN && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        f.ctr[CTR_NJOBS] = f.W * f.H;
        for (uint32_t k = 1; k < ctr_words(f.slots); k++) f.ctr[k] = 0;            /* the frame starts from clean counters */
    }
    /* the train's step list grows by this part's pixels, behind the survivors of the last step's shade pass (nothing reads
       the length before this launch has ended; the other threads read TC_BASE only) */
    if (TRAIN && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) f.ctr[TC_NJOBS] = f.ctr[TC_BASE] + (f.row1 - f.row0) * f.W;
    if ((float)lx >= f.c.viewportSize[0] || (float)ly >= f.c.viewportSize[1]) return;
    if (TRAIN && gy >= f.row1) return;
    uint32_t jitter = (uint32_t)f.blueNoise[(lx & 0xFF) + ((ly & 0xFF) << 8)] * 256u + (uint32_t)f.blueNoise[f.c.frameIndex & 0xFFFF];
    h16 wx, wy; WeylNth(jitter, wx, wy);
    float px = (((float)lx + (float)wx) / f.c.viewportSize[0]) * 2.0f - 1.0f;
    float py = (((float)ly + (float)wy) / f.c.viewportSize[1]) * 2.0f - 1.0f;
    float projPos[4] = { px, py, 1.0f, 1.0f }, viewPos[4], world[4], cam[4];
    mat_mul_vec(f.c.projInvMtx, projPos, viewPos);
    float vw = viewPos[3];
    for (int k = 0; k < 4; k++) viewPos[k] /= vw;
    mat_mul_vec(f.c.viewInvMtx, viewPos, world);
    const float origin[4] = { 0, 0, 0, 1 };
    mat_mul_vec(f.c.viewInvMtx, origin, cam);
    float cw = cam[3];
    for (int k = 0; k < 4; k++) cam[k] /= cw;

    const uint32_t jobIndex = ly * (uint32_t)f.c.viewportSize[0] + lx, stackIndex = jobIndex * f.c.numRays;
    PtRayStackSlot* s = f.stack + stackIndex;
    F3 d = { world[0] - cam[0], world[1] - cam[1], world[2] - cam[2] };
    float inv = 1.0f / __builtin_sqrtf(fdot(d, d));
    uint32_t rs = f.rngStack[stackIndex].randState;
    if (f.reseed || f.seedFromJitter) rs = jitter;
    float4* sp = (float4*)s;
    sp[0] = make_float4(d.x * inv, d.y * inv, d.z * inv, __uint_as_float(rs));
    sp[1] = make_float4(cam[0], cam[1], cam[2], __uint_as_float(0u));
    s->prob = 0;
    /* (TC_BASE was set when the last step ended; a part's jobs are its own rows') */
    if (TRAIN) f.jobs[f.ctr[TC_BASE] + (jobIndex - f.row0 * (uint32_t)f.c.viewportSize[0])] = f.slotBase + stackIndex;
    else f.jobs[jobIndex] = stackIndex;
}
#define k_launch k_launch_t<false>

/* trace.comp:58-233 minus the two walks: one job.  `depth` comes back for live jobs (hits always, misses where TRAIN asks). */
template <bool TRAIN>
__device__ __forceinline__ void shade_job(const FrameArgs& f, uint32_t j, uint32_t slot, bool live, bool& wantShadow, uint32_t& nextJob, PtShadowRay& sr,
                                          uint32_t& depth)
{
    if (live) {
        PtRayStackSlot* s = f.stack + slot;
        const uint4* hp = (const uint4*)(f.hitRecs + j);
        uint4 h0 = hp[0], h1 = hp[1];
        const int32_t i1 = (int32_t)h0.y, i2 = (int32_t)h0.z, i3 = (int32_t)h0.w;
        if (i1 < 0) {                                                          /* trace.comp:192-196; the job ends here */
            s->prob = 0;
            if (TRAIN) depth = s->currentDepth;
            if (!f.fuseCompact) f.jobs[j] = PT_JOB_DEAD;
        } else {
            const float t = __uint_as_float(h0.x);
            const h16 bx = lo16(h1.x), by = hi16(h1.x), bz = lo16(h1.y);
            const uint32_t origBvhId = h1.z;
            const float4* sp = (const float4*)s;
            float4 s0 = sp[0], s1 = sp[1];
            F3 rd = { s0.x, s0.y, s0.z }, ro = { s1.x, s1.y, s1.z };
            Rng rng; rng.rnd = __float_as_uint(s0.w);
            const uint32_t currentDepth = __float_as_uint(s1.w);
            depth = currentDepth;
            F3 hitPos = { t * rd.x + ro.x, t * rd.y + ro.y, t * rd.z + ro.z };    /* trace.comp:198 */

            /* shadeHit, trace.comp:58-163 */
            const uint4* ap = (const uint4*)f.aux;
            uint4 a1 = ap[i1], a2 = ap[i2], a3 = ap[i3];
            const h16 inv255 = (h16)1.0f / (h16)255.0f;
            h16 albedo[4];
            for (int k = 0; k < 4; k++) {
                h16 c1 = (h16)(float)((a1.w >> (8 * k)) & 0xFF) * inv255;
                h16 c2 = (h16)(float)((a2.w >> (8 * k)) & 0xFF) * inv255;
                h16 c3 = (h16)(float)((a3.w >> (8 * k)) & 0xFF) * inv255;
                albedo[k] = (bx * c1 + by * c2) + bz * c3;
            }
            H3 n1 = { (h16)__uint_as_float(a1.x), (h16)__uint_as_float(a1.y), (h16)__uint_as_float(a1.z) };
            H3 n2 = { (h16)__uint_as_float(a2.x), (h16)__uint_as_float(a2.y), (h16)__uint_as_float(a2.z) };
            H3 n3 = { (h16)__uint_as_float(a3.x), (h16)__uint_as_float(a3.y), (h16)__uint_as_float(a3.z) };
            H3 normal = hadd3(hadd3(hscale(bx, n1), hscale(by, n2)), hscale(bz, n3));
            const h16 g = (h16)2.2f;
            for (int k = 0; k < 3; k++) albedo[k] = (h16)powf((float)albedo[k], (float)g);
            H3 rd16 = f2h3(rd);
            if ((float)hdot(normal, rd16) > 0.0f) normal = { -normal.x, -normal.y, -normal.z };

            float coinFlip = rng.getRandF();
            h16 falloff; F3 lightDir; H3 lightRadiance; float lightMaxT;
            if (coinFlip < 0.5f) {                                             /* trace.comp:85-105 */
                const float4* lp = (const float4*)f.lights;
                float4 l0 = lp[0], l1 = lp[1];
                F3 posDiff = fsub({ l0.x, l0.y, l0.z }, hitPos);
                float dist = __builtin_sqrtf(fdot(posDiff, posDiff));
                lightDir = { posDiff.x / dist, posDiff.y / dist, posDiff.z / dist };
                lightMaxT = dist - 0.00005f;
                lightRadiance = { (h16)l1.x, (h16)l1.y, (h16)l1.z };
                falloff = (h16)(1.0f / (dist * dist + 1.0f)) * hmax16((h16)0.0f, hdot(normal, f2h3(lightDir)));
            } else {                                                           /* trace.comp:106-121 */
                h16 gx, gy; WeylNth(rng.getRand(), gx, gy);
                lightDir = h2f3(to_coord_space(normal, cosineHemisphere(gx, gy)));
                lightMaxT = 1000.0f;
                lightRadiance = { (h16)f.c.ambientRadiance[0], (h16)f.c.ambientRadiance[1], (h16)f.c.ambientRadiance[2] };
                falloff = (h16)1.0f;
            }
            sr.o[0] = hitPos.x; sr.o[1] = hitPos.y; sr.o[2] = hitPos.z; sr.max_t = lightMaxT;
            sr.d[0] = lightDir.x; sr.d[1] = lightDir.y; sr.d[2] = lightDir.z; sr.origBvhId = origBvhId;
            sr.slot = slot; sr._pad[0] = sr._pad[1] = sr._pad[2] = 0;
            wantShadow = true;

            /* the contribution if the light sample is visible; the shadow pass clears it otherwise (trace.comp:123-130) */
            h16 wIn[3];
            const h16 lr[3] = { lightRadiance.x, lightRadiance.y, lightRadiance.z };
            if ((float)albedo[3] > 0.5f) {
                for (int k = 0; k < 3; k++) wIn[k] = (((h16)2.0f * falloff) * albedo[k]) * lr[k];
            } else {
                h16 ind = (h16)(fdot(lightDir, rd) > 0.995f ? 1.0f : 0.0f);
                for (int k = 0; k < 3; k++) wIn[k] = ((h16)2.0f * ind) * lr[k];
            }

            h16 gx, gy; WeylNth(rng.getRand(), gx, gy);                        /* trace.comp:133-134 */
            F3 nextDir = h2f3(to_coord_space(normal, cosineHemisphere(gx, gy)));
            h16 prob;
            bool cont = true;
            if ((float)albedo[3] < 0.5f) {                                     /* trace.comp:136-149: ior is always 1.3 */
                const h16 eta = (h16)1.3f;
                H3 I = rd16, N = normal;
                h16 d = hdot(N, I);
                h16 k = (h16)1.0f - (eta * eta) * ((h16)1.0f - d * d);
                H3 refr;
                if ((float)k < 0.0f) refr = { (h16)0.0f, (h16)0.0f, (h16)0.0f };
                else refr = hsub3(hscale(eta, I), hscale(eta * d + hsqrt16(k), N));
                nextDir = h2f3(refr);
                if (nextDir.x == 0.0f && nextDir.y == 0.0f && nextDir.z == 0.0f) nextDir = h2f3(hsub3(I, hscale((h16)2.0f * d, N)));
                for (int q = 0; q < 4; q++) albedo[q] = hsqrt16(albedo[q]);
                prob = (h16)1.0f;
            } else {                                                           /* trace.comp:150-154 */
                prob = (h16)1.0f / (h16)0.7f;
                if (rng.getRandF() > 0.7f) cont = false;
            }

            if (cont && currentDepth != f.c.numRays - 1) {                     /* trace.comp:212-223 */
                float4* np = (float4*)(s + 1);
                np[0] = make_float4(nextDir.x, nextDir.y, nextDir.z, __uint_as_float(rng.rnd));
                np[1] = make_float4(hitPos.x, hitPos.y, hitPos.z, __uint_as_float(currentDepth + 1));
                ((uint2*)(s + 1))[4] = make_uint2(0u, 0u);                     /* hitAlbedo = 0 */
                nextJob = slot + 1;
            } else {                                                           /* trace.comp:224-227 */
                if (f.zeroNextSlot && currentDepth != f.c.numRays - 1) (s + 1)->prob = 0;
            }
            s->randState = rng.rnd;                                            /* trace.comp:229-232 */
            uint4 o;
            o.x = (uint32_t)hbits(albedo[0]) | ((uint32_t)hbits(albedo[1]) << 16);
            o.y = (uint32_t)hbits(albedo[2]) | ((uint32_t)hbits(albedo[3]) << 16);
            o.z = (uint32_t)hbits(wIn[0]) | ((uint32_t)hbits(wIn[1]) << 16);
            o.w = (uint32_t)hbits(wIn[2]) | ((uint32_t)hbits(prob) << 16);
            ((uint4*)s)[2] = o;
        }
    }

}

#define PT_SHADE_BLOCK 256
/* What the shade pass hands on: the job's next entry (in place, the reference's way, trace.comp:222-226, or appended to the
 * next, dense list) and its shadow ray.
 * Counts go to the device counters once per workgroup: one atomic per wavefront on the same word serialises in one
 * L2 channel and was the larger part of this kernel's time.  Four wavefronts per workgroup is the measured optimum
 * for the path tracer: with sixteen the atomics are fewer still, but a workgroup then needs sixteen free wave slots
 * on one CU at once, which the persistent traversal kernels of the frames in flight rarely leave (201 against
 * 220 spp/s at 1080p).  Every thread of the workgroup calls this. */
__device__ __forceinline__ void shade_emit(const FrameArgs& f, uint32_t j, bool live, bool wantShadow, uint32_t nextJob, const PtShadowRay& sr,
                                           uint32_t (*shCount)[PT_SHADE_BLOCK / 64], uint32_t* shBase)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool cont = nextJob != PT_JOB_DEAD;
    const uint64_t cm = __ballot(cont), m = __ballot(wantShadow), lm = __ballot(live);
    if (lane == 0) { shCount[0][wave] = (uint32_t)__popcll(cm); shCount[1][wave] = (uint32_t)__popcll(m); shCount[2][wave] = (uint32_t)__popcll(lm); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot[3] = { 0, 0, 0 };
        for (int w = 0; w < PT_SHADE_BLOCK / 64; w++)
            for (int k = 0; k < 3; k++) { uint32_t c = shCount[k][w]; shCount[k][w] = tot[k]; tot[k] += c; }
        shBase[0] = (f.fuseCompact && tot[0]) ? atomicAdd(f.ctr + CTR_COMPACT_OUT, tot[0]) : 0u;
        shBase[1] = tot[1] ? atomicAdd(f.shadowCtr, tot[1]) : 0u;
        if (tot[2] && !f.fuseCompact) atomicAdd(f.ctr + CTR_LIVE, tot[2]);      /* a compacted list holds live jobs only: its length is the count */
    }
    __syncthreads();

    if (f.fuseCompact) {
        if (cont) f.jobsOut[shBase[0] + shCount[0][wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0))] = nextJob;
    } else if (live && wantShadow) {
        f.jobs[j] = nextJob;
    }
    if (f.tileShadow) {
        /* the job list is still in pixel order here (depth 0, right after the launch pass): job j is pixel j */
        if (j < f.W * f.H) {
            const uint32_t x = j % f.W, y = j / f.W;
            PtShadowRay* dst = f.shadowRays + ((size_t)(y >> 3) * (f.W >> 3) + (x >> 3)) * 64u + ((y & 7u) << 3) + (x & 7u);
            if (wantShadow) { const uint4* src = (const uint4*)&sr; uint4* d4 = (uint4*)dst; d4[0] = src[0]; d4[1] = src[1]; d4[2] = src[2]; }
            else dst->slot = PT_JOB_DEAD;
        }
    } else if (wantShadow) {
        uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        const uint4* src = (const uint4*)&sr;
        uint4* dst = (uint4*)(f.shadowRays + shBase[1] + shCount[1][wave] + rank);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    }
}

__global__ __launch_bounds__(PT_SHADE_BLOCK) void k_shade(FrameArgs f, uint32_t depthIndex)
{
    __shared__ uint32_t shCount[3][PT_SHADE_BLOCK / 64];
    __shared__ uint32_t shBase[2];
    const uint32_t j = blockIdx.x * PT_SHADE_BLOCK + threadIdx.x;
    const uint32_t njobs = f.ctr[CTR_NJOBS];
    const uint32_t slot = (j < njobs) ? f.jobs[j] : PT_JOB_DEAD;
    const bool live = slot != PT_JOB_DEAD;
    bool wantShadow = false;
    uint32_t nextJob = PT_JOB_DEAD, depth = 0;
    PtShadowRay sr;
    shade_job<false>(f, j, slot, live, wantShadow, nextJob, sr, depth);
    shade_emit(f, j, live, wantShadow, nextJob, sr, shCount, shBase);
    (void)depthIndex;
}

/* The shade pass of the frame train: the step's job list holds jobs of several frames, each frame at its own depth, and its
 * length is known on the device only: the launch has a workgroup for every entry the list can have (the frames in flight times
 * the band's pixels) and the workgroups beyond its end leave at once.  (A fixed grid that walked the list in strides kept the
 * loop's invariants - light, ambient colour, constants - in vector registers: 72 of them, one wavefront per SIMD beside the
 * other band's traversal launch; one trip per workgroup takes 54, two wavefronts per SIMD there: 194.3 -> 195.9 spp/s.)  Ray
 * counts are kept per depth (each job knows its own): per workgroup in LDS, one global atomic per depth and workgroup at the end. */
/* Under the certified walk the launch also is the second walk of a short list: its first walkBlocks workgroups take the
 * jobs the first pass listed, four each, one to a wavefront (wave_walk_ray: the reference-order walk, its answer written
 * into the job's hit record with the `listed` mark left on), and shade them then and there, one lane per job; the ordinary
 * workgroups skip the jobs that carry the mark.  A list's walk is all latency - the dependent fetches of its longest ray -
 * and in a launch of its own it stood between the traversal launch and this one; here it runs beside the shading of
 * everything else.  A long list (a.onlyBelow jobs or more) has been walked one job to a lane by a launch before this one,
 * which writes its answers without the mark: the walking workgroups then leave at once and every job is shaded in place. */
template <bool WALK>        /* false: no walking workgroups (walkBlocks = 0), and the registers of the plain shade pass: 54 instead of 58 */
__global__ __launch_bounds__(PT_SHADE_BLOCK) void k_shade_train(FrameArgs f, TraceArgs a, uint32_t walkBlocks)
{
    __shared__ uint32_t shCount[3][PT_SHADE_BLOCK / 64];
    __shared__ uint32_t shBase[2];
    __shared__ uint32_t shDepth[2][PT_MAX_TRAIN_DEPTH];
    const int lane = threadIdx.x & 63;
    /* one wavefront's jobs are of a few depths: one LDS atomic per depth present */
    auto count_depths = [&](bool live, uint32_t depth, bool wantShadow) {
        uint64_t todo = __ballot(live);
        while (todo) {
            const int first = __ffsll((unsigned long long)todo) - 1;
            const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)depth, first);
            const uint64_t same = __ballot(live && depth == d);
            const uint64_t sh = __ballot(live && depth == d && wantShadow);
            if (lane == first && d < PT_MAX_TRAIN_DEPTH) { atomicAdd(&shDepth[0][d], (uint32_t)__popcll(same)); if (sh) atomicAdd(&shDepth[1][d], (uint32_t)__popcll(sh)); }
            todo &= ~same;
        }
    };
    auto flush_depths = [&]() {
        for (int k = threadIdx.x; k < 2 * PT_MAX_TRAIN_DEPTH; k += PT_SHADE_BLOCK) {
            const uint32_t c = (&shDepth[0][0])[k];
            if (c) atomicAdd(f.rayCounts + 2 * (k % PT_MAX_TRAIN_DEPTH) + k / PT_MAX_TRAIN_DEPTH, (unsigned long long)c);
        }
    };
    if (WALK && blockIdx.x < walkBlocks) {
        const uint32_t listed = *a.countPtr;                                /* the same for every thread of the launch */
        if (blockIdx.x * (PT_SHADE_BLOCK / 64) >= listed || (a.onlyBelow && listed >= a.onlyBelow)) return;      /* a workgroup per four listed jobs */
        __builtin_amdgcn_s_setprio(3);                                       /* the launch lasts as long as the longest of these walks */
        for (int k = threadIdx.x; k < 2 * PT_MAX_TRAIN_DEPTH; k += PT_SHADE_BLOCK) (&shDepth[0][0])[k] = 0;
        __syncthreads();
        unsigned long long n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        {   /* one trip per workgroup, as for the ordinary workgroups (a loop would keep the shading's invariants in registers across the walk) */
            const uint32_t k = blockIdx.x * (PT_SHADE_BLOCK / 64) + wave;
            const bool have = k < listed;                                    /* the same for the whole wavefront */
            const uint32_t j = have ? a.order[k] : 0u;
            if (have) wave_walk_ray<MODE_JOBS, false>(a, j, 1u, n0, n1, n2, n3);
            const bool live = have && lane == 0;                            /* every lane wrote the record; one shades the job */
            const uint32_t slot = live ? f.jobs[j] : PT_JOB_DEAD;
            bool wantShadow = false;
            uint32_t nextJob = PT_JOB_DEAD, depth = 0;
            PtShadowRay sr;
            shade_job<true>(f, j, slot, live, wantShadow, nextJob, sr, depth);
            count_depths(live, depth, wantShadow);
            shade_emit(f, j, live, wantShadow, nextJob, sr, shCount, shBase);
            __syncthreads();
        }
        flush_depths();
        return;
    }
    const uint32_t njobs = f.ctr[TC_NJOBS];
    const uint32_t block = blockIdx.x - (WALK ? walkBlocks : 0u);
    if (block * PT_SHADE_BLOCK >= njobs) return;                           /* the same for every thread of a workgroup */
    for (int k = threadIdx.x; k < 2 * PT_MAX_TRAIN_DEPTH; k += PT_SHADE_BLOCK) (&shDepth[0][0])[k] = 0;
    __syncthreads();
    {
        const uint32_t j = block * PT_SHADE_BLOCK + threadIdx.x;
        const uint32_t slot = (j < njobs) ? f.jobs[j] : PT_JOB_DEAD;
        bool live = slot != PT_JOB_DEAD;
        if (WALK && live && f.hitRecs[j].listed) live = false;               /* a walking workgroup's job */
        bool wantShadow = false;
        uint32_t nextJob = PT_JOB_DEAD, depth = 0;
        PtShadowRay sr;
        shade_job<true>(f, j, slot, live, wantShadow, nextJob, sr, depth);
        count_depths(live, depth, wantShadow);
        shade_emit(f, j, live, wantShadow, nextJob, sr, shCount, shBase);
    }
    __syncthreads();
    flush_depths();
}

/* Ends a step of the frame train; one thread.  The list the shade pass appended to becomes the next step's list (a frame
 * that enters in that step puts its jobs behind it, k_launch_t); the shadow queue this step's traversal launch emptied is
 * free again. */
__global__ void k_train_step_end(FrameArgs f)
{
    const uint32_t survivors = f.ctr[TC_OUT];
    f.ctr[TC_BASE] = survivors;
    f.ctr[TC_NJOBS] = survivors;
    f.ctr[TC_OUT] = 0; f.ctr[TC_FETCH] = 0;
    f.ctr[TC_NSHADOW + f.shadowQueue] = 0;
    f.ctr[TC_FLAG] = 0; f.ctr[TC_FLAG + 1] = 0;
}

/* Records the ray counts of one depth; one thread.  With the fused compaction it also hands the new list length
 * over and clears the counters the next depth's closest-hit and shade passes start from. */
__global__ void k_record_counts(FrameArgs f, uint32_t depthIndex)
{
    f.rayCounts[2 * depthIndex] += f.fuseCompact ? f.ctr[CTR_NJOBS] : f.ctr[CTR_LIVE];
    f.rayCounts[2 * depthIndex + 1] += f.shadowCtr[0];
    if (f.fuseCompact) {
        f.ctr[CTR_NJOBS] = f.ctr[CTR_COMPACT_OUT];
        f.ctr[CTR_FETCH_CLOSEST] = 0; f.ctr[CTR_COMPACT_OUT] = 0; f.ctr[CTR_LIVE] = 0;
    }
}

/* Stream compaction of the job list (the role of raysort.comp:23-55, over the whole list instead of
 * per 256-wide row segment; the order of jobs does not influence any result). */
__global__ __launch_bounds__(256) void k_compact(FrameArgs f)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    const uint32_t njobs = f.ctr[CTR_NJOBS];
    uint32_t slot = (j < njobs) ? f.jobs[j] : PT_JOB_DEAD;
    const bool live = slot != PT_JOB_DEAD;
    const int lane = threadIdx.x & 63;
    uint64_t m = __ballot(live);
    if (!m) return;
    int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(f.ctr + CTR_COMPACT_OUT, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (live) {
        uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        f.jobsOut[base + rank] = slot;
    }
}
__global__ void k_compact_finish(FrameArgs f) { f.ctr[CTR_NJOBS] = f.ctr[CTR_COMPACT_OUT]; }

/* composite.frag:30-55 */
__global__ __launch_bounds__(256) void k_resolve(FrameArgs f)
{
    const uint32_t pix = f.row0 * f.W + blockIdx.x * 256 + threadIdx.x;
    if (pix >= f.row1 * f.W) return;
    const PtRayStackSlot* s = f.stack + (size_t)pix * f.c.numRays;
    /* composite.frag walks the slots from the deepest to the first and starts over from zero at every slot whose prob is not
       positive: what comes out depends on the slots before the first such slot only.  Most paths end after two or three
       bounces, so finding that slot first and walking back from it reads a third of the stack's lines */
    int end = (int)f.c.numRays;
    for (int depth = 0; depth < (int)f.c.numRays; depth++)
        if (!((float)hi16(((const uint32_t*)(s + depth))[11]) > 0.0f)) { end = depth; break; }
    float L[3] = { 0, 0, 0 };
    for (int depth = end - 1; depth >= 0; depth--) {
        uint4 q = ((const uint4*)(s + depth))[2];
        float prob = (float)hi16(q.w);
        float w[3] = { (float)lo16(q.z), (float)hi16(q.z), (float)lo16(q.w) };
        float al[3] = { (float)lo16(q.x), (float)hi16(q.x), (float)lo16(q.y) };
        for (int k = 0; k < 3; k++) { L[k] *= prob; L[k] += w[k]; L[k] *= al[k]; }
    }
    uint2 cur;
    cur.x = (uint32_t)hbits((h16)L[0]) | ((uint32_t)hbits((h16)L[1]) << 16);
    cur.y = (uint32_t)hbits((h16)L[2]) | ((uint32_t)hbits((h16)1.0f) << 16);
    f.current[pix] = cur;
    float4 acc = f.accumulation[pix];
    acc.x += L[0]; acc.y += L[1]; acc.z += L[2]; acc.w += 1.0f;
    f.accumulation[pix] = acc;
}

/* Display colour of composite.frag:18-26,57: pow(ACESFilm(acc.rgb / acc.w), 1/2.2), fp32.  The reference
 * writes it to the swapchain; here it goes to a float3 image (the shader's value) and/or an RGBA8 image
 * (UNORM conversion: round to nearest, alpha 255).  clamp is min(max(x,0),1) with minNum/maxNum, so a
 * pixel without samples (0/0) comes out black. */
__device__ __forceinline__ float aces_film(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float v = (x * (a * x + b)) / (x * (c * x + d) + e);
    return fminf(fmaxf(v, 0.0f), 1.0f);
}
__global__ __launch_bounds__(256) void k_tonemap(const float4* accumulation, uint32_t npix, float* rgb32f, uint32_t* rgba8)
{
    const uint32_t pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    float4 acc = accumulation[pix];
    float o[3] = { acc.x / acc.w, acc.y / acc.w, acc.z / acc.w };
    uint32_t packed = 0xFF000000u;
    for (int k = 0; k < 3; k++) {
        o[k] = powf(aces_film(o[k]), 1.0f / 2.2f);
        packed |= (uint32_t)rintf(o[k] * 255.0f) << (8 * k);
    }
    if (rgb32f) { rgb32f[3 * (size_t)pix + 0] = o[0]; rgb32f[3 * (size_t)pix + 1] = o[1]; rgb32f[3 * (size_t)pix + 2] = o[2]; }
    if (rgba8) rgba8[pix] = packed;
}

/* Synthetic incoherent rays for the traversal microbench: five 24-bit uniforms per ray from
 * successive splitmix64 outputs; origin uniform in the box, direction uniform on the sphere. */
__device__ __forceinline__ uint64_t splitmix64(uint64_t& s)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_generate_rays(PtRay* rays, uint64_t seed, uint64_t first, uint64_t n, float ax, float ay, float az,
                                                       float bx, float by, float bz)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed + (first + i) * 0x9E3779B97F4A7C15ull;
    float u[5];
    for (int k = 0; k < 5; k++) u[k] = (float)(splitmix64(s) >> 40) * (1.0f / 16777216.0f);
    float z = 1.0f - 2.0f * u[3], phi = 6.283185307179586f * u[4];
    float rr = __builtin_sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float4 q0 = make_float4(ax + (bx - ax) * u[0], ay + (by - ay) * u[1], az + (bz - az) * u[2], 0.001f);
    float4 q1 = make_float4(rr * cosf(phi), rr * sinf(phi), z, 100000.0f);
    float4* p = (float4*)(rays + i);
    p[0] = q0; p[1] = q1;
}
