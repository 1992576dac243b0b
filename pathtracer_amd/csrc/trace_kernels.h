/*
 * trace_kernels.h — BVH traversal + ray/triangle intersection kernels for gfx950.
 *
 * Two families, both one ray per lane of a 64-wide wavefront, both persistent: a launch is
 * sized to the machine (workgroups per CU x CUs) and every wavefront keeps pulling rays from one
 * global counter until the pool is empty, refilling its idle lanes in batches so that a lane
 * whose ray ended early does not idle for the rest of its wavefront's longest ray.  A refill has no
 * wait of its own: the loads of the new rays are issued before the step of the lanes that still
 * have rays and used after it.
 *
 *  k_trace_ref   walks the reference's threaded 32 B node array exactly as traceRay does
 *                (reference intersections.glsl:57-115): no stack, one dependent 32 B fetch per
 *                visit, two more dependent fetches per leaf.  Kept as the on-device restatement
 *                and for the subgroup-vote ("SPECULATIVE") build of trace.comp.
 *
 *  k_trace_flat  walks the flattened layout: one 64 B record per step - either an interior
 *                record holding both children's pre-digested boxes or a leaf record holding the
 *                triangle inline - so a step is one line fetch followed by either two box tests
 *                or one triangle test.  Pending siblings wait on a short per-lane stack in LDS
 *                (entry e of lane l at [e][l]: conflict-free for any mix of stack depths), with
 *                the static part of their box verdict already decided and the max_t part
 *                re-checked when they are popped, which makes the walk visit exactly the nodes
 *                the reference visits, in the same order.  ORDERED = true descends into the
 *                nearer child first instead.
 */
#pragma once
#include "../../include/ptcore.h"
#include "device_math.h"

#pragma clang fp contract(off)

/* Diagnostic builds of k_trace_flat (extra loads or instructions per step, cycle counters, a refill that waits for its own
 * loads, same-line statistics) hang on seven hooks that are empty in the product: the code behind them lives in
 * trace_diagnostics.h, which only a build with -DPT_DIAGNOSTICS includes (scripts/build_variant.sh <name> -DPT_DIAGNOSTICS
 * -DPT_EXPERIMENT_...). */
#ifdef PT_DIAGNOSTICS
#include "trace_diagnostics.h"
#else
#define PT_DIAG_STATE
#define PT_DIAG_COUNT_ROOT_BOX nBox++;
#define PT_DIAG_AFTER_REFILL
#define PT_DIAG_BEFORE_FETCH
#define PT_DIAG_AFTER_FETCH
#define PT_DIAG_COUNT_BOX_PAIR nBox += 2;
#define PT_DIAG_AT_EXIT
#endif

#ifndef PT_STACK_LDS
#define PT_STACK_LDS 13          /* per-lane stack entries held in LDS; deeper entries spill to global memory.  Every entry that moves
                                    from the spill area into LDS saves traffic (the certified batch kernel: 9 / 10 / 11 entries ->
                                    27.3 / 26.7 / 26.1 ms); 13 is what fits beside everything else: 20 KB per workgroup of the
                                    near-first batch kernel (seven per CU), 26.6 KB of the certified frame kernel (five per CU;
                                    with 14 the other band's passes find no LDS: 172 -> 147 spp/s).  12 / 13 / 14: near-first batch
                                    23.70 / 23.58 / 23.50 ms, near-first frames 195.5 / 196.7 / 197.4 spp/s */
#endif
/* where entry e (counted from the first one beyond the LDS part) of this lane's stack sits in the spill area: entry by entry,
   the lanes of the launch side by side (a lane's entries side by side instead - one line per lane - measured no better on the
   near-first batch, 23.56 against 23.52 ms, and costs two registers) */
#define PT_SPILL_AT(e) ((size_t)(e) * totalLanes + gid)
#ifndef PT_COMPACT_STRIDE
#define PT_COMPACT_STRIDE 48     /* distance between 48 B records in memory (64 = padded, an experiment) */
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define PT_WAVES_PER_BLOCK 4
#define PT_BLOCK (64 * PT_WAVES_PER_BLOCK)

/* Result of a closest-hit walk as the shade stage consumes it, 32 B. */
struct PtHitRec {
    float    t;
    int32_t  i1, i2, i3;         /* -1: miss */
    uint16_t bary[3];            /* f16(gamma, alpha, beta), intersections.glsl:49 */
    uint16_t _pad0;
    uint32_t origBvhId;          /* what traceRay leaves in r.origBvhId (intersections.glsl:113) */
    uint32_t listed;             /* certified walk: 1 = the job is on the list for the second, reference-order walk and this record
                                    is not its answer yet (a second walk that is a launch of its own writes the answer with 0 here;
                                    the frame train's shade pass, which walks a short list itself, leaves the 1 and skips the job
                                    in its ordinary workgroups) */
};

/* A next-event-estimation ray, 48 B (trace.comp:95-100 / 111-116). */
struct PtShadowRay {
    float    o[3]; float max_t;
    float    d[3]; uint32_t origBvhId;
    uint32_t slot;               /* ray-stack slot whose wIn is cleared when the ray is occluded */
    uint32_t _pad[3];
};

enum { MODE_RAYS = 0, MODE_JOBS = 1, MODE_SHADOW = 2,
       MODE_MIXED = 3 };    /* k_trace_flat only: ids below *countPtr are jobs (closest hit), the *countPtr2 ids above them shadow rays
                               (any hit) - the closest-hit pass of one bounce and the shadow pass of the bounce before it, which
                               both wait for the same shade pass and for nothing else, as one launch */

/* One node of the path from the root to the last node of the reference array (always the right child: the nodes whose
 * skip link is 0), digested like every other box. */
struct PtSpineBox {
    float    c[3];
    uint32_t index;          /* the node's index in the reference array */
    uint16_t h[6];
    uint32_t pad;
};
#define PT_SPINE_MAX 64

struct TraceArgs {
    /* reference layout */
    const PtBVHNode* nodes;  uint32_t numNodes;
    const uint32_t*  indices;
    const float4*    positions;
    /* flattened layout: one buffer, interior records first, then leaf records */
    const uint4*     records;
    uint32_t         recordBytes;    /* size of that buffer */
    uint32_t         leafBase;       /* byte offset of the first leaf record */
    const uint4*     tris;           /* = records + leafBase/16 */
    const uint32_t*  leafRef;        /* 48 B records: reference node index of each leaf */
    uint32_t         lastLeaf;       /* 48 B records: triangle index of the leaf that is the reference array's last node */
    float            rootC[3]; uint16_t rootH[6]; uint32_t rootRef;
    float            lastC[3]; uint16_t lastH[6];          /* the same digest of the box of the reference array's last node */
    const PtSpineBox* spine;         /* the nodes from the last node up to (not including) the root, deepest first; see k_trace_cert */
    uint32_t         spineCount;     /* 0: no table (a path of more than PT_SPINE_MAX nodes) */
    /* treelet layout: records (128 B) and leaves (64 B) in one buffer addressed in 64 B units */
    const uint4*     treelets;
    uint32_t         treeletBytes;
    uint2*           overflow;       /* [entry][global lane] spill area of the LDS stacks */
    uint32_t         overflowEntries;
    /* work */
    uint32_t*        fetchCounter;   /* next unclaimed ray */
    const uint32_t*  countPtr;       /* number of rays, read on the device (nullptr: use count) */
    const uint32_t*  countPtr2;      /* MODE_MIXED: number of shadow rays behind the *countPtr jobs */
    uint64_t         count;
    uint32_t         refillMin;      /* refill a wavefront when this many lanes are idle */
    uint32_t         claimChunk;     /* ids a wavefront takes from the counter at a time (0: chosen from the ray count) */
    /* second walk of the certified walk: two launches are queued and the list's length, known on the device only, decides
       which of them works - k_trace_ref_wave leaves at once when the list holds onlyBelow rays or more (0: never),
       k_trace_flat when it holds fewer than onlyFrom */
    uint32_t         onlyBelow, onlyFrom;
    /* MODE_RAYS */
    const PtRay*     rays;  PtHit* hits;
    const uint32_t*  order;          /* the k-th ray to trace is rays[order[k]] (nullptr: rays[k]) */
    /* MODE_JOBS */
    const uint32_t*  jobs;  PtRayStackSlot* stack;  PtHitRec* hitRecs;
    /* MODE_SHADOW */
    const PtShadowRay* shadowRays;
    /* counting build */
    unsigned long long* stats;       /* {box tests, triangle tests, records fetched, rays} */
    /* certified walk */
    uint32_t*        flagList;       /* indices of the rays whose result the walk could not certify */
    uint32_t*        flagCount;
};

/* ---- per-mode ray fetch / result store ------------------------------------------------------- */

/* A ray is fetched in two halves so that the traversal loop can issue the loads of new rays, go on with a step of
 * the rays it already has, and only then turn the loaded words into a ray: the loads then cost no round trip of their
 * own.  issue_ray returns false for a dead job (nothing was loaded). */
struct RayWords { float4 w0, w1; uint32_t tag; };

/* The rays of a batch come in and its hits go out once per launch, gigabytes of them beside the traversal records the L2
 * is there for: their loads and stores are marked non-temporal so that the caches do not keep them (a 2^26-ray batch:
 * 24.4 -> 24.0 ms with both; the store alone gains half of it, the loads alone nothing).  The path tracer's passes keep
 * plain accesses: what they write the next pass reads.  PT_STREAM_NT=0 builds the plain accesses for comparison. */
#ifndef PT_STREAM_NT
#define PT_STREAM_NT 1
#endif
typedef float nt_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 stream_load(const float4* p)
{
    if (!NT) return *p;
    nt_f4 v = __builtin_nontemporal_load((const nt_f4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT> __device__ __forceinline__ void stream_store(float4* p, float4 o)
{
    if (!NT) { *p = o; return; }
    nt_f4 v = { o.x, o.y, o.z, o.w };
    __builtin_nontemporal_store(v, (nt_f4*)p);
}

template <int MODE, bool NT = (PT_STREAM_NT != 0)>
__device__ __forceinline__ bool issue_ray(const TraceArgs& a, uint64_t id, RayWords& w)
{
    if (MODE == MODE_RAYS) {
        const float4* p = (const float4*)(a.rays + id);          /* id: the ray's own index */
        w.w0 = stream_load<NT>(p); w.w1 = stream_load<NT>(p + 1); w.tag = 0;
        return true;
    } else if (MODE == MODE_JOBS) {
        uint32_t slot = a.jobs[id];
        if (slot == PT_JOB_DEAD) return false;
        const float4* p = (const float4*)(a.stack + slot);                  /* trace.comp:185-190 */
        w.w0 = p[0]; w.w1 = p[1]; w.tag = slot;
        return true;
    } else {
        const float4* p = (const float4*)(a.shadowRays + id);
        w.w0 = p[0]; w.w1 = p[1]; w.tag = a.shadowRays[id].slot;
        return true;
    }
}

template <int MODE>
__device__ __forceinline__ void finish_ray(const RayWords& w, RayRegs& r, uint32_t& exclude, uint32_t& tag)
{
    tag = w.tag;
    if (MODE == MODE_RAYS) {
        r.o = { w.w0.x, w.w0.y, w.w0.z }; r.min_t = w.w0.w;
        ray_set_dir(r, { w.w1.x, w.w1.y, w.w1.z }); r.max_t = w.w1.w;
        exclude = 0;
    } else if (MODE == MODE_JOBS) {
        ray_set_dir(r, { w.w0.x, w.w0.y, w.w0.z });
        r.o = { w.w1.x, w.w1.y, w.w1.z };
        r.min_t = 0.001f; r.max_t = 100000.0f;
        exclude = 0;
    } else {
        r.o = { w.w0.x, w.w0.y, w.w0.z }; r.max_t = w.w0.w;
        ray_set_dir(r, { w.w1.x, w.w1.y, w.w1.z });
        r.min_t = 0.001f;
        exclude = __float_as_uint(w.w1.w);
    }
}

template <int MODE>
__device__ __forceinline__ bool load_ray(const TraceArgs& a, uint64_t id, RayRegs& r, uint32_t& exclude, uint32_t& tag)
{
    RayWords w;
    if (!issue_ray<MODE>(a, id, w)) return false;
    finish_ray<MODE>(w, r, exclude, tag);
    return true;
}

template <int MODE, bool NT = (PT_STREAM_NT != 0)>
__device__ __forceinline__ void store_result(const TraceArgs& a, uint64_t id, uint32_t tag, bool hit, float t, int32_t i1, int32_t i2, int32_t i3,
                                             float alpha, float beta, float gamma, uint32_t origBvhId, uint32_t listed = 0u)
{
    if (MODE == MODE_RAYS) {
        float4 o;
        o.x = t;
        o.y = __int_as_float(hit ? i1 : -1); o.z = __int_as_float(hit ? i2 : -1); o.w = __int_as_float(hit ? i3 : -1);
        stream_store<NT>((float4*)(a.hits + id), o);
    } else if (MODE == MODE_JOBS) {
        uint4 o0, o1;
        o0.x = __float_as_uint(t);
        o0.y = (uint32_t)(hit ? i1 : -1); o0.z = (uint32_t)(hit ? i2 : -1); o0.w = (uint32_t)(hit ? i3 : -1);
        o1.x = (uint32_t)hbits((h16)gamma) | ((uint32_t)hbits((h16)alpha) << 16);
        o1.y = (uint32_t)hbits((h16)beta);
        o1.z = origBvhId; o1.w = listed;      /* PtHitRec.listed */
        uint4* p = (uint4*)(a.hitRecs + id);
        p[0] = o0; p[1] = o1;
    } else {
        if (hit) {      /* occluded: the light sample contributes nothing (trace.comp:123) */
            PtRayStackSlot* s = a.stack + tag;
            s->wIn[0] = 0; s->wIn[1] = 0; s->wIn[2] = 0;
        }
    }
}

/* Ray ids for the idle lanes of a wavefront.  The wavefront draws ids from a private reserve and tops the reserve up
 * from the global counter a chunk at a time (one atomic by one lane), so that thousands of wavefronts do not queue
 * on one word for every refill.  All members are wave-uniform. */
struct RayPool {
    uint64_t next = 0, end = 0;      /* the reserve: ids [next, end) belong to this wavefront */
    uint64_t total = 0;
    uint32_t chunk = 64;
    bool     drained = false;        /* the global counter has passed total: no more chunks */

    __device__ __forceinline__ void init(const TraceArgs& a, uint64_t totalRays, uint32_t numWaves)
    {
        total = totalRays;
        /* small launches keep the chunk at one wavefront's worth so that the rays spread over all wavefronts */
        uint64_t per = totalRays / (numWaves ? numWaves : 1u);
        uint32_t c = a.claimChunk ? a.claimChunk : (uint32_t)(per > 128 ? 128 : per);
        chunk = c < 64 ? 64u : (c & ~63u);
    }
    __device__ __forceinline__ bool exhausted() const { return drained && next >= end; }

    /* Wave-uniform call; returns this lane's new id, or ~0 if it gets none. */
    __device__ __forceinline__ uint64_t claim(uint32_t* counter, bool idle, uint64_t idleMask, int nIdle)
    {
        const int lane = threadIdx.x & 63;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0));
        const uint64_t avail = end - next;
        uint64_t id = ~0ull;
        if (avail >= (uint64_t)nIdle || drained) {
            if (idle && rank < avail) id = next + rank;
            next += avail < (uint64_t)nIdle ? avail : (uint64_t)nIdle;
        } else {
            const int leader = __ffsll((unsigned long long)idleMask) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(counter, chunk);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, leader));
            if (idle) id = rank < avail ? next + rank : (uint64_t)base + (rank - avail);
            next = (uint64_t)base + ((uint64_t)nIdle - avail);
            end = (uint64_t)base + chunk;
            if (end >= total) { end = end < total ? end : total; drained = true; if (next > end) next = end; }
        }
        return (id < total) ? id : ~0ull;
    }
};

/* ---- reference-layout walk -------------------------------------------------------------------- */

template <int MODE, bool ANYHIT, bool COUNT>
__global__ __launch_bounds__(PT_BLOCK) void k_trace_ref(TraceArgs a)
{
    const uint64_t total = a.countPtr ? (uint64_t)*a.countPtr : a.count;
    RayPool pool; pool.init(a, total, gridDim.x * (PT_BLOCK / 64));
    bool active = false;
    RayRegs r; uint32_t exclude = 0, tag = 0; uint64_t id = 0;
    uint32_t index = 0, hitIndex = 0; bool hit = false;
    int32_t i1 = -1, i2 = -1, i3 = -1; float alpha = 0, beta = 0, gamma = 0;
    unsigned long long nBox = 0, nTri = 0, nRec = 0, nRay = 0;

    for (;;) {
        uint64_t idleMask = __ballot(!active);
        int nIdle = __popcll(idleMask);
        if (nIdle == 64 && pool.exhausted()) break;
        if (!pool.exhausted() && (nIdle >= (int)a.refillMin || nIdle == 64)) {
            uint64_t nid = pool.claim(a.fetchCounter, !active, idleMask, nIdle);
            if (!active && nid != ~0ull) {
                id = nid;
                if (load_ray<MODE>(a, id, r, exclude, tag)) {
                    active = true; index = 0; hitIndex = 0; hit = false; i1 = i2 = i3 = -1;
                    if (COUNT) nRay++;
                }
            }
        }
        if (!active) continue;

        /* one visit of the loop at intersections.glsl:64-111 */
        bool done = false, ranOff = false;
        const float4* np = (const float4*)(a.nodes + index);
        float4 na = np[0], nb = np[1];
        if (COUNT) { nBox++; nRec++; }
        if (box_raw(r, na, nb)) {
            int32_t right = __float_as_int(nb.w);
            if (right <= 0 && (exclude == 0 || exclude != index)) {
                uint32_t tri = (uint32_t)(-right) / 3u;
                uint32_t v1 = a.indices[3 * tri], v2 = a.indices[3 * tri + 1], v3 = a.indices[3 * tri + 2];
                float4 p1 = a.positions[v1], p2 = a.positions[v2], p3 = a.positions[v3];
                if (COUNT) { nTri++; nRec += 4; }
                float t, al, be, ga;
                F3 P1 = { p1.x, p1.y, p1.z };
                if (tri_test(r, P1, fsub({ p2.x, p2.y, p2.z }, P1), fsub({ p3.x, p3.y, p3.z }, P1), t, al, be, ga)) {
                    r.max_t = t; alpha = al; beta = be; gamma = ga;
                    i1 = (int32_t)v1; i2 = (int32_t)v2; i3 = (int32_t)v3;
                    hit = true; hitIndex = index;
                    if (ANYHIT) done = true;
                }
            }
            index++;
            if (index >= a.numNodes) { done = true; ranOff = true; }
        } else {
            index = (uint32_t)__float_as_int(na.w);
            if (index == 0) done = true;
        }
        if (done) {
            store_result<MODE>(a, id, tag, hit, r.max_t, i1, i2, i3, alpha, beta, gamma, (ranOff && !(ANYHIT && hit)) ? hitIndex : 0u);
            active = false;
        }
    }
    if (COUNT) {
        atomicAdd(a.stats + 0, nBox); atomicAdd(a.stats + 1, nTri); atomicAdd(a.stats + 2, nRec); atomicAdd(a.stats + 3, nRay);
    }
}

/* The SPECULATIVE build of trace.comp at depth 0 (intersections.glsl:73-77): one wavefront per
 * 8x8 pixel tile, lane k at tile position (k % 8, k / 8); every lane follows the wavefront's vote,
 * so all lanes share one node index. */
__global__ __launch_bounds__(64) void k_trace_ref_speculative(TraceArgs a, uint32_t W, uint32_t H)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t x = blockIdx.x * 8 + (lane & 7), y = blockIdx.y * 8 + (lane >> 3);
    const uint64_t id = (uint64_t)y * W + x;
    RayRegs r; uint32_t exclude = 0, tag = 0;
    bool live = (x < W && y < H) && load_ray<MODE_JOBS>(a, id, r, exclude, tag);
    if (!live) return;      /* lanes without a job take no part in the vote (trace.comp:168) */
    uint32_t index = 0, hitIndex = 0; bool hit = false, ranOff = false;
    int32_t i1 = -1, i2 = -1, i3 = -1; float alpha = 0, beta = 0, gamma = 0;
    while (index < a.numNodes) {
        const float4* np = (const float4*)(a.nodes + index);
        float4 na = np[0], nb = np[1];
        bool pass = box_raw(r, na, nb);
        if (__any(pass)) {
            int32_t right = __float_as_int(nb.w);
            if (right <= 0) {
                uint32_t tri = (uint32_t)(-right) / 3u;
                uint32_t v1 = a.indices[3 * tri], v2 = a.indices[3 * tri + 1], v3 = a.indices[3 * tri + 2];
                float4 p1 = a.positions[v1], p2 = a.positions[v2], p3 = a.positions[v3];
                float t, al, be, ga;
                F3 P1 = { p1.x, p1.y, p1.z };
                if (tri_test(r, P1, fsub({ p2.x, p2.y, p2.z }, P1), fsub({ p3.x, p3.y, p3.z }, P1), t, al, be, ga)) {
                    r.max_t = t; alpha = al; beta = be; gamma = ga;
                    i1 = (int32_t)v1; i2 = (int32_t)v2; i3 = (int32_t)v3;
                    hit = true; hitIndex = index;
                }
            }
            index++;
            if (index >= a.numNodes) ranOff = true;
        } else {
            index = (uint32_t)__float_as_int(na.w);
            if (index == 0) break;
        }
    }
    store_result<MODE_JOBS>(a, id, tag, hit, r.max_t, i1, i2, i3, alpha, beta, gamma, ranOff ? hitIndex : 0u);
}

/* The any-hit walk shadeHit makes (trace.comp:123) in the SPECULATIVE build: the same vote, among the lanes of the 8x8
 * tile whose closest-hit walk found something (the others have returned, trace.comp:194-195).  One wavefront per tile,
 * shadow rays at queue place tile * 64 + lane (k_shade with tileShadow), places without a ray marked PT_JOB_DEAD.  A
 * lane that finds an occluder clears its light sample and leaves: it votes no more (stopIfHit, intersections.glsl:97). */
__global__ __launch_bounds__(64) void k_trace_ref_speculative_shadow(TraceArgs a, uint32_t W, uint32_t H)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t id = ((uint64_t)blockIdx.y * (W >> 3) + blockIdx.x) * 64u + lane;
    if (a.shadowRays[id].slot == PT_JOB_DEAD) return;
    RayRegs r; uint32_t exclude = 0, tag = 0;
    load_ray<MODE_SHADOW>(a, id, r, exclude, tag);
    uint32_t index = 0;
    while (index < a.numNodes) {
        const float4* np = (const float4*)(a.nodes + index);
        float4 na = np[0], nb = np[1];
        bool pass = box_raw(r, na, nb);
        if (__any(pass)) {
            int32_t right = __float_as_int(nb.w);
            if (right <= 0 && (exclude == 0 || exclude != index)) {
                uint32_t tri = (uint32_t)(-right) / 3u;
                uint32_t v1 = a.indices[3 * tri], v2 = a.indices[3 * tri + 1], v3 = a.indices[3 * tri + 2];
                float4 p1 = a.positions[v1], p2 = a.positions[v2], p3 = a.positions[v3];
                float t, al, be, ga;
                F3 P1 = { p1.x, p1.y, p1.z };
                if (tri_test(r, P1, fsub({ p2.x, p2.y, p2.z }, P1), fsub({ p3.x, p3.y, p3.z }, P1), t, al, be, ga)) {
                    store_result<MODE_SHADOW>(a, id, tag, true, t, -1, -1, -1, 0, 0, 0, 0);
                    return;
                }
            }
            index++;
        } else {
            index = (uint32_t)__float_as_int(na.w);
            if (index == 0) break;
        }
    }
    (void)H;
}

/* The reference-order walk with one WAVEFRONT per ray, for the few rays the certified walk hands back: a list that short
 * is all latency (the walk of its longest ray, one dependent fetch per node), so the 64 lanes work on one ray instead.
 * The reference's array is in pre-order with skip links, so wherever the walk stands, the nodes it can visit next are the
 * ones right behind it: lane l fetches node index + l (one coalesced 2 KB read) and runs the max_t-independent part of
 * its box test; the walk through those 64 nodes - enter and go to the next node, or fail and follow the skip link - is
 * then a chase through registers (a scalar loop over v_readlane), no memory involved.  Every leaf on that path tests its
 * triangle in its own lane, all at once, acceptance left open; the first of them, in walk order, that the reference
 * would accept (t <= max_t of that moment: nothing before it on the path changed max_t) shrinks max_t, and the chase is
 * redone from the node behind it with the boxes and triangle results already in registers.  When the path leaves the
 * window the next 64 nodes are fetched from where it went.  Node for node this is the loop of intersections.glsl:64-111
 * with the arithmetic of k_trace_ref; closest hit only, rays from a.order (the list) or in batch order. */
/* One ray, walked by the whole wavefront (every lane calls this with the same id and leaves with the same result stored by
 * every lane); `listed` goes into the hit record of a job (PtHitRec.listed). */
template <int MODE, bool COUNT>      /* MODE_RAYS or MODE_JOBS */
__device__ __forceinline__ void wave_walk_ray(const TraceArgs& a, uint64_t id, uint32_t listed,
                                              unsigned long long& nBox, unsigned long long& nTri, unsigned long long& nRec, unsigned long long& nRay)
{
    const int lane = threadIdx.x & 63;
    enum : uint32_t { SUCC_NEXT_WINDOW = 64, SUCC_JUMP = 65, SUCC_END = 66, SUCC_RAN_OFF = 67 };
    RayRegs r; uint32_t exclude = 0, tag = 0;
    if (load_ray<MODE>(a, id, r, exclude, tag)) {                   /* the same for all lanes: one ray per wavefront */
    if (COUNT) nRay++;
    uint32_t index = 0, hitIndex = 0;
    bool hit = false, ranOff = false, ended = false;
    int32_t i1 = -1, i2 = -1, i3 = -1; float alpha = 0, beta = 0, gamma = 0;

    /* every window moves the walk forward by at least one node, every redone chase by at least one lane, every chase
       step by at least one lane: the three loops end by themselves, and are bounded besides so that a malformed
       array cannot hold a wavefront for ever */
    for (uint32_t windows = 0; !ended && windows < a.numNodes; windows++) {
        const uint32_t base = index, n = base + (uint32_t)lane;
        const bool valid = n < a.numNodes;
        float4 na = make_float4(0, 0, 0, 0), nb = na;
        if (valid) { const float4* np = (const float4*)(a.nodes + n); na = np[0]; nb = np[1]; }
        if (COUNT) nRec++;
        float tmin = 0.0f;
        bool s = false;
        if (valid) {
            const float cx = (na.x + nb.x) * 0.5f, cy = (na.y + nb.y) * 0.5f, cz = (na.z + nb.z) * 0.5f;
            s = box_static(r, cx, cy, cz, (h16)(na.x - cx), (h16)(na.y - cy), (h16)(na.z - cz), (h16)(nb.x - cx), (h16)(nb.y - cy), (h16)(nb.z - cz), tmin);
        }
        const int32_t right = __float_as_int(nb.w);
        const uint32_t nextAbs = (uint32_t)__float_as_int(na.w);
        const uint64_t leafBits = __ballot(valid && right <= 0);
        uint64_t triDone = 0;
        float t = 0, al = 0, be = 0, ga = 0; bool ok0 = false; int32_t v1 = -1, v2 = -1, v3 = -1;
        uint32_t p = 0;
        for (int redo = 0; redo <= 64; redo++) {
            /* where the walk goes from each node of the window, with the max_t of now */
            const bool pass = s && r.max_t > tmin;
            uint32_t succ;
            if (pass) succ = (n + 1u >= a.numNodes) ? (uint32_t)SUCC_RAN_OFF : (uint32_t)lane + 1u;
            else if (nextAbs == 0u) succ = SUCC_END;
            else succ = (nextAbs - base < 64u) ? nextAbs - base : (uint32_t)SUCC_JUMP;
            uint64_t visited = 0;
            uint32_t q = p, last = p;
            for (int hop = 0; hop < 64 && q < 64u; hop++) { visited |= 1ull << q; last = q; q = (uint32_t)__builtin_amdgcn_readlane((int)succ, (int)q); }
            if (COUNT) nBox += (unsigned long long)__popcll(visited);
            /* the leaves on that path test their triangles, each in its own lane; results are kept for a redone chase */
            const uint64_t leafVisited = visited & leafBits & __ballot(pass);
            const uint64_t need = leafVisited & ~triDone;
            if ((need >> lane) & 1ull) {
                const uint32_t tri = (uint32_t)(-right) / 3u;
                const uint4* tp = a.tris + (size_t)tri * 4;
                const uint4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                F3 p1 = { __uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z) };
                F3 e1 = { __uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z) };
                F3 e2 = { __uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z) };
                RayRegs open = r; open.max_t = __builtin_huge_valf();           /* acceptance against max_t is decided in walk order below */
                ok0 = tri_test(open, p1, e1, e2, t, al, be, ga);
                v1 = (int32_t)q0.w; v2 = (int32_t)q1.w; v3 = (int32_t)q2.w;
            }
            if (COUNT) nTri += (unsigned long long)__popcll(need);
            triDone |= need;
            const uint64_t accept = __ballot(((leafVisited >> lane) & 1ull) && ok0 && t <= r.max_t);
            if (accept == 0) {                                     /* nothing on the path changes max_t: the walk leaves the window */
                if (q == SUCC_NEXT_WINDOW) index = base + 64u;
                else if (q == SUCC_JUMP) index = (uint32_t)__builtin_amdgcn_readlane((int)nextAbs, (int)last);
                else { ended = true; ranOff = q == SUCC_RAN_OFF; }
                break;
            }
            const int l = __ffsll((unsigned long long)accept) - 1;            /* the first leaf, in walk order, that the reference accepts */
            r.max_t = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(t), l));
            alpha = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(al), l));
            beta = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(be), l));
            gamma = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ga), l));
            i1 = __builtin_amdgcn_readlane(v1, l); i2 = __builtin_amdgcn_readlane(v2, l); i3 = __builtin_amdgcn_readlane(v3, l);
            hit = true; hitIndex = base + (uint32_t)l;
            /* the leaf's box passed, so the walk goes on at the node behind it, now with the smaller max_t */
            if (base + (uint32_t)l + 1u >= a.numNodes) { ended = true; ranOff = true; break; }
            p = (uint32_t)l + 1u;
            if (p >= 64u) { index = base + 64u; break; }
        }
    }
    store_result<MODE>(a, id, tag, hit, r.max_t, i1, i2, i3, alpha, beta, gamma, ranOff ? hitIndex : 0u, listed);
    }
}

template <int MODE, bool COUNT>      /* MODE_RAYS or MODE_JOBS */
__global__ __launch_bounds__(PT_BLOCK) void k_trace_ref_wave(TraceArgs a)      /* any whole number of wavefronts per workgroup up to four */
{
    const int lane = threadIdx.x & 63;
    const uint64_t total = a.countPtr ? (uint64_t)*a.countPtr : a.count;
    if (a.onlyBelow && total >= a.onlyBelow) return;        /* a list this long is the one-ray-per-lane launch's */
    unsigned long long nBox = 0, nTri = 0, nRec = 0, nRay = 0;
    /* Everything a wavefront decides here it decides for all its lanes: no lane may leave a loop or skip a block on its
       own, or the votes and lane reads below would run without it (a version that let lane 0 alone claim the ray and
       store the result had the compiler send the other 63 lanes round the loop without it).  Hence rays are dealt to
       wavefronts by stride, not by an atomic, and every lane stores the (same) result. */
    /* these few wavefronts are what a whole pass waits for, usually beside another launch that fills the CUs: they go first
       wherever the hardware lets a wavefront's priority decide (second walk of a frame-train step 2.2 -> 0.9 ms) */
    __builtin_amdgcn_s_setprio(3);
    const uint32_t wavesPerBlock = blockDim.x >> 6;       /* launched with one to four wavefronts per workgroup */
    const uint32_t numWaves = gridDim.x * wavesPerBlock;
    const uint32_t firstRay = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wavesPerBlock + (threadIdx.x >> 6)));
    for (uint64_t k = firstRay; k < total; k += numWaves) {
        const uint64_t id = a.order ? (uint64_t)a.order[k] : k;
        wave_walk_ray<MODE, COUNT>(a, id, 0u, nBox, nTri, nRec, nRay);
    }
    if (COUNT && lane == 0) {
        atomicAdd(a.stats + 0, nBox); atomicAdd(a.stats + 1, nTri); atomicAdd(a.stats + 2, nRec); atomicAdd(a.stats + 3, nRay);
    }
}

/* ---- flattened-layout walk -------------------------------------------------------------------- */

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h16x2 as_h2(uint32_t u) { return __builtin_bit_cast(h16x2, u); }

/* Both children's boxes of one interior record at once: the fields are stored as (left, right) half pairs, so every
 * fp16 operation of intersections.glsl:11-27 runs as one packed instruction on both boxes.  Element-wise the
 * arithmetic is the scalar box_static's, operation for operation. */
__device__ __forceinline__ void box_pair(const RayRegs& r, h16x2 rx, h16x2 ry, h16x2 rz, f32x2 cx, f32x2 cy, f32x2 cz,
                                         h16x2 ax, h16x2 ay, h16x2 az, h16x2 bx, h16x2 by, h16x2 bz,
                                         bool& sL, bool& sR, float& tminL, float& tminR)
{
    /* -O = f16(c - o): negation commutes with both roundings, so a + (-O) is the reference's a - O bit for bit, and
       the compiler no longer spends an instruction per axis on negating O for its add */
    /* element by element: written as vector minus splat, the compiler keeps three (o,o) register pairs alive across the
       whole loop, 16 more VGPRs per lane, which costs a wavefront per SIMD when several of these kernels share the CUs */
    f32x2 dx = { cx.x - r.o.x, cx.y - r.o.x }, dy = { cy.x - r.o.y, cy.y - r.o.y }, dz = { cz.x - r.o.z, cz.y - r.o.z };
    h16x2 nox = __builtin_convertvector(dx, h16x2), noy = __builtin_convertvector(dy, h16x2), noz = __builtin_convertvector(dz, h16x2);
    h16x2 t0 = (ax + nox) * rx, t1 = (bx + nox) * rx;
    h16x2 tmin = __builtin_elementwise_min(t0, t1), tmax = __builtin_elementwise_max(t0, t1);
    t0 = (ay + noy) * ry; t1 = (by + noy) * ry;
    tmin = __builtin_elementwise_max(tmin, __builtin_elementwise_min(t0, t1));
    tmax = __builtin_elementwise_min(tmax, __builtin_elementwise_max(t0, t1));
    t0 = (az + noz) * rz; t1 = (bz + noz) * rz;
    tmin = __builtin_elementwise_max(tmin, __builtin_elementwise_min(t0, t1));
    tmax = __builtin_elementwise_min(tmax, __builtin_elementwise_max(t0, t1));
    const float fminL = (float)tmin.x, fminR = (float)tmin.y, fmaxL = (float)tmax.x, fmaxR = (float)tmax.y;
    sL = (fmaxL > 0.0f || fminL > 0.0f) && fmaxL >= fminL && r.min_t < fmaxL;
    sR = (fmaxR > 0.0f || fminR > 0.0f) && fmaxR >= fminR && r.min_t < fmaxR;
    tminL = fminL; tminR = fminR;
}
__device__ __forceinline__ f32x2 as_f2(uint32_t lo, uint32_t hi) { f32x2 v = { __uint_as_float(lo), __uint_as_float(hi) }; return v; }

/* f16(b-c) of both children for one axis of a 48 B record: magnitude bits of f16(a-c) plus the stored 5-bit difference
 * (biased by 15), sign 0.  Both halves at once: neither the sum nor the bias removal carries across the halves,
 * because each half's result is a valid non-negative fp16 pattern by construction. */
__device__ __forceinline__ h16x2 rebuild_b(uint32_t aPair, uint32_t delta, int axis)
{
    return as_h2((aPair & 0x7FFF7FFFu) + ((delta >> (5 * axis)) & 0x001F001Fu) + 0xFFF0FFF1u);
}

template <int MODE, bool ANYHIT, bool ORDERED, bool COUNT, bool COMPACT>
__global__ __launch_bounds__(PT_BLOCK) void k_trace_flat(TraceArgs a)
{
    __shared__ uint32_t sRef[PT_WAVES_PER_BLOCK][PT_STACK_LDS][64];
    __shared__ uint16_t sTmin[PT_WAVES_PER_BLOCK][PT_STACK_LDS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gid = (uint64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    const uint64_t totalLanes = (uint64_t)gridDim.x * PT_BLOCK;
    const uint64_t numJobs = a.countPtr ? (uint64_t)*a.countPtr : a.count;       /* MODE_MIXED: ids below this are jobs, the rest shadow rays */
    const uint64_t total = numJobs + (MODE == MODE_MIXED ? (uint64_t)*a.countPtr2 : 0ull);
    if (MODE != MODE_MIXED && total < a.onlyFrom) return;      /* a list this short is the one-ray-per-wavefront launch's */
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.records, 0, (int)a.recordBytes, 0x00020000);

    RayPool pool; pool.init(a, total, gridDim.x * PT_WAVES_PER_BLOCK);
    bool active = false, pend = false;      /* pend: this lane has fetched the index of its next ray but not the ray */
    bool shadowLane = MODE == MODE_SHADOW, freshShadow = false;      /* MODE_MIXED: what kind of ray the lane walks / has just loaded */
    uint32_t pendIdx = 0;
    RayRegs r; uint32_t exclude = 0, tag = 0; uint64_t id = 0;
    h16x2 rx = { (h16)0.0f, (h16)0.0f }, ry = rx, rz = rx;       /* rcpD duplicated into both halves */
    uint32_t cur = 0; int sp = 0;
    bool hit = false, ranOff = false;
    int32_t hitI1 = -1, hitI2 = -1, hitI3 = -1; uint32_t hitLeaf = 0;     /* ids come with the leaf record: no second fetch when the ray ends */
    float alpha = 0, beta = 0, gamma = 0;
    unsigned long long nBox = 0, nTri = 0, nRec = 0, nRay = 0;
    PT_DIAG_STATE

    const h16 rAx = hfrombits(a.rootH[0]), rAy = hfrombits(a.rootH[1]), rAz = hfrombits(a.rootH[2]);
    const h16 rBx = hfrombits(a.rootH[3]), rBy = hfrombits(a.rootH[4]), rBz = hfrombits(a.rootH[5]);

    /* turns the words a lane loaded into its next ray: root box test, traversal state */
    /* the three places where a mixed launch has to ask what kind of ray a lane holds */
    auto issue = [&](uint64_t rid, RayWords& w) -> bool {
        if (MODE != MODE_MIXED) return issue_ray<MODE>(a, rid, w);
        freshShadow = rid >= numJobs;
        return freshShadow ? issue_ray<MODE_SHADOW>(a, rid - numJobs, w) : issue_ray<MODE_JOBS>(a, rid, w);
    };
    auto store = [&](bool h, float t, int32_t s1, int32_t s2, int32_t s3, float al, float be, float ga, uint32_t orig) {
        if (MODE != MODE_MIXED) store_result<MODE>(a, id, tag, h, t, s1, s2, s3, al, be, ga, orig);
        else if (shadowLane) store_result<MODE_SHADOW>(a, id - numJobs, tag, h, t, s1, s2, s3, al, be, ga, orig);
        else store_result<MODE_JOBS>(a, id, tag, h, t, s1, s2, s3, al, be, ga, orig);
    };
    auto start_ray = [&](const RayWords& w) {
        if (MODE != MODE_MIXED) finish_ray<MODE>(w, r, exclude, tag);
        else { shadowLane = freshShadow; if (shadowLane) finish_ray<MODE_SHADOW>(w, r, exclude, tag); else finish_ray<MODE_JOBS>(w, r, exclude, tag); }
        if (COUNT) { nRay++; PT_DIAG_COUNT_ROOT_BOX }
        /* origBvhId is a reference node index (a leaf's); 48 B records are told apart by their reference */
        if (COMPACT && exclude != 0) exclude = PT_LEAF_BIT | ((uint32_t)(-a.nodes[exclude].right) / 3u);
        hit = false; ranOff = false; sp = 0;
        rx = { r.rcpD.x, r.rcpD.x }; ry = { r.rcpD.y, r.rcpD.y }; rz = { r.rcpD.z, r.rcpD.z };
        float ftmin;
        bool s = box_static(r, a.rootC[0], a.rootC[1], a.rootC[2], rAx, rAy, rAz, rBx, rBy, rBz, ftmin);
        if (s && r.max_t > ftmin) { active = true; cur = a.rootRef; }
        else store(false, r.max_t, -1, -1, -1, 0, 0, 0, 0);      /* root box missed */
    };

    for (;;) {
        /* ---- refill: idle lanes take ray ids and issue the loads of their rays; the words are used after the step
                below, so that these loads and the step's record fetch are one wait.  With an order to follow, a lane
                first fetches the index of its ray the same way and the ray a turn later ---- */
        const bool canTake = !active && !pend;
        uint64_t idleMask = __ballot(canTake);
        int nIdle = __popcll(idleMask);
        if (nIdle == 64 && pool.exhausted()) break;
        bool fresh = false;
        RayWords w;
        if (MODE != MODE_SHADOW && pend) {      /* the index fetched a turn ago names this lane's next ray */
            id = pendIdx; pend = false;
            fresh = issue(id, w);
        }
        if (!pool.exhausted() && (nIdle >= (int)a.refillMin || nIdle == 64)) {
            uint64_t nid = pool.claim(a.fetchCounter, canTake, idleMask, nIdle);
            if (canTake && nid != ~0ull) {
                if (MODE != MODE_SHADOW && a.order) { pendIdx = a.order[nid]; pend = true; }   /* used next turn: no wait here */
                else { id = nid; fresh = issue(id, w); }
            }
        }
        PT_DIAG_AFTER_REFILL

        if (active) {
        /* ---- one step: fetch the 64 B record `cur` names with four 16 B buffer loads issued back to back (the
                empty asm keeps the compiler from splitting them into narrower per-branch loads, which cost twice
                the L2 requests), then run two box tests or one triangle test ---- */
        const bool leaf = (cur & PT_LEAF_BIT) != 0;
        const uint32_t off = (cur & ~PT_LEAF_BIT) * (COMPACT ? (uint32_t)PT_COMPACT_STRIDE : 64u) + (leaf ? a.leafBase : 0u);
        PT_DIAG_BEFORE_FETCH
        u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
        u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0);
        u32x4 q3 = { 0u, 0u, 0u, 0u };
        if (COMPACT) {
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
        } else {
            q3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48, 0, 0);
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        }
        PT_DIAG_AFTER_FETCH
        if (COUNT) nRec++;
        bool needPop = true, done = false;

        if (leaf) {
            /* the reference array's last node was entered; which leaf this is: its reference index (64 B record) or
               the reference to it (48 B record; `exclude` was translated to the same form when the ray was loaded) */
            if (COMPACT ? (cur == (PT_LEAF_BIT | a.lastLeaf)) : ((q3.y & 1u) != 0)) ranOff = true;
            const uint32_t leafId = COMPACT ? cur : q3.x;
            if (exclude == 0 || exclude != leafId) {                        /* intersections.glsl:81 */
                if (COUNT) nTri++;
                float t, al, be, ga;
                F3 p1 = { __uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z) };
                F3 e1 = { __uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z) };
                F3 e2 = { __uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z) };
                if (tri_test(r, p1, e1, e2, t, al, be, ga)) {
                    r.max_t = t; alpha = al; beta = be; gamma = ga;
                    hit = true;
                    if (MODE != MODE_SHADOW) { hitI1 = (int32_t)q0.w; hitI2 = (int32_t)q1.w; hitI3 = (int32_t)q2.w; }
                    if (MODE == MODE_JOBS || MODE == MODE_MIXED) hitLeaf = leafId;
                    if (ANYHIT || (MODE == MODE_MIXED && shadowLane)) { done = true; needPop = false; }
                }
            }
        } else {
            if (COUNT) { PT_DIAG_COUNT_BOX_PAIR }
            bool sL, sR; float tminL, tminR;
            uint32_t refL, refR;
            if (COMPACT) {
                box_pair(r, rx, ry, rz, as_f2(q0.x, q0.y), as_f2(q0.z, q0.w), as_f2(q1.x, q1.y), as_h2(q1.z), as_h2(q1.w), as_h2(q2.x),
                         rebuild_b(q1.z, q2.w, 0), rebuild_b(q1.w, q2.w, 1), rebuild_b(q2.x, q2.w, 2), sL, sR, tminL, tminR);
                refL = q2.y; refR = q2.z;
            } else {
                box_pair(r, rx, ry, rz, as_f2(q0.x, q0.y), as_f2(q0.z, q0.w), as_f2(q1.x, q1.y), as_h2(q1.z), as_h2(q1.w), as_h2(q2.x),
                         as_h2(q2.y), as_h2(q2.z), as_h2(q2.w), sL, sR, tminL, tminR);
                refL = q3.x; refR = q3.y;
            }
            const bool pL = sL && r.max_t > tminL, pR = sR && r.max_t > tminR;
            uint32_t first = refL, second = refR; float tminSecond = tminR;
            if (ORDERED && pL && pR && tminR < tminL) { first = refR; second = refL; tminSecond = tminL; }
            if (pL && pR) {
                /* the sibling waits with its tmin; `max_t > tmin` is re-evaluated when it is popped */
                const uint16_t tb = hbits((h16)tminSecond);                 /* exact: tmin came from an fp16 */
                if (sp < PT_STACK_LDS) { sRef[wave][sp][lane] = second; sTmin[wave][sp][lane] = tb; }
                else a.overflow[PT_SPILL_AT(sp - PT_STACK_LDS)] = make_uint2(second, tb);
                sp++;
                cur = first; needPop = false;
            } else if (pL) { cur = refL; needPop = false; }
            else if (pR) { cur = refR; needPop = false; }
        }

        /* pending siblings come off the stack until one still passes `max_t > tmin` */
        while (needPop) {
            if (sp == 0) { done = true; break; }
            sp--;
            /* the LDS read is unconditional (clamped index) so that it stays a plain ds_read: merged with the rare
               overflow load it would become a flat load, which goes through the vector-memory path */
            const int e = sp < PT_STACK_LDS ? sp : PT_STACK_LDS - 1;
            uint32_t ref = sRef[wave][e][lane]; uint16_t tb = sTmin[wave][e][lane];
            asm volatile("" : "+v"(ref));       /* keeps the two sources apart: see above */
            if (sp >= PT_STACK_LDS) { uint2 o = a.overflow[PT_SPILL_AT(sp - PT_STACK_LDS)]; ref = o.x; tb = (uint16_t)o.y; }
            if (r.max_t > (float)hfrombits(tb)) { cur = ref; needPop = false; }
        }

        if (done) {
            uint32_t orig = 0u;
            if ((MODE == MODE_JOBS || (MODE == MODE_MIXED && !shadowLane)) && ranOff && hit) orig = COMPACT ? a.leafRef[hitLeaf & ~PT_LEAF_BIT] : hitLeaf;
            store(hit, r.max_t, hitI1, hitI2, hitI3, alpha, beta, gamma, orig);
            active = false;
        }
        }   /* if (active) */

        /* ---- the rays loaded above start here; they step with the others from the next turn ---- */
        if (fresh) start_ray(w);
    }
    PT_DIAG_AT_EXIT
    if (COUNT) {
        atomicAdd(a.stats + 0, nBox); atomicAdd(a.stats + 1, nTri); atomicAdd(a.stats + 2, nRec); atomicAdd(a.stats + 3, nRay);
    }
}

/* ---- treelet walk ---------------------------------------------------------------------------------
 *
 * One step = one PtTreelet (128 B, eight 16 B loads) or one PtTri (64 B).  The record of a node X holds the boxes of
 * X's children L, R and of their children, so a ray pays one dependent fetch per two levels of the tree where the
 * walk descends, and every byte of the line it fetched is there to be used.  Node for node the walk is k_trace_flat's:
 *
 *   stage A (a lane that arrives at X): L and R are judged with the max_t of now, as the two-box walk judges them when
 *     it visits X; the half entered first is the nearer one (ORDERED) or L; if both pass, the other half waits on the
 *     stack as ONE entry with its tmin - a leaf half as its triangle, an interior half as "X's record, this half"
 *     (PT_HALF_BIT) - and `max_t > tmin` is evaluated again when it is popped, exactly as the two-box walk does with
 *     the sibling it pushed;
 *   stage B (the same lane in the same step if the half it enters is interior; or a lane that popped a waiting half):
 *     the half's two children, whose boxes are in the record already, are judged with the max_t of now - the two-box
 *     walk would fetch the half's record for this - the first is entered, the second waits with its tmin.
 *
 * A popped interior half fetches X's record a second time (the line is often still in a cache) and runs stage B only.
 * So the boxes tested, their order and the max_t each test sees are those of k_trace_flat<ORDERED>: hits are bit-identical
 * to it (and, with ORDERED = false, to the reference's walk), and the stack holds one entry per level, as there. */
#define PT_HALF_BIT   0x40000000u       /* traversal reference: enter the record at one of its halves (stage B only) */
#define PT_HALF_RIGHT 0x20000000u       /* ... the right one */
#define PT_UNIT_MASK  0x03FFFFFFu
template <int MODE, bool ANYHIT, bool ORDERED, bool COUNT>
__global__ __launch_bounds__(PT_BLOCK) void k_trace_tree(TraceArgs a)
{
    __shared__ uint32_t sRef[PT_WAVES_PER_BLOCK][PT_STACK_LDS][64];
    __shared__ uint16_t sTmin[PT_WAVES_PER_BLOCK][PT_STACK_LDS][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gid = (uint64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    const uint64_t totalLanes = (uint64_t)gridDim.x * PT_BLOCK;
    const uint64_t numJobs = a.countPtr ? (uint64_t)*a.countPtr : a.count;       /* MODE_MIXED: ids below this are jobs, the rest shadow rays */
    const uint64_t total = numJobs + (MODE == MODE_MIXED ? (uint64_t)*a.countPtr2 : 0ull);
    if (MODE != MODE_MIXED && total < a.onlyFrom) return;      /* a list this short is the one-ray-per-wavefront launch's */
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.treelets, 0, (int)a.treeletBytes, 0x00020000);

    RayPool pool; pool.init(a, total, gridDim.x * PT_WAVES_PER_BLOCK);
    bool active = false, pend = false;
    bool shadowLane = MODE == MODE_SHADOW, freshShadow = false;      /* MODE_MIXED: what kind of ray the lane walks / has just loaded */
    uint32_t pendIdx = 0;
    RayRegs r; uint32_t exclude = 0, tag = 0; uint64_t id = 0;
    h16x2 rx = { (h16)0.0f, (h16)0.0f }, ry = rx, rz = rx;
    uint32_t cur = 0; int sp = 0;
    bool hit = false, ranOff = false;
    int32_t hitI1 = -1, hitI2 = -1, hitI3 = -1; uint32_t hitLeaf = 0;
    float alpha = 0, beta = 0, gamma = 0;
    unsigned long long nBox = 0, nTri = 0, nRec = 0, nRay = 0;

    const h16 rAx = hfrombits(a.rootH[0]), rAy = hfrombits(a.rootH[1]), rAz = hfrombits(a.rootH[2]);
    const h16 rBx = hfrombits(a.rootH[3]), rBy = hfrombits(a.rootH[4]), rBz = hfrombits(a.rootH[5]);

    /* the three places where a mixed launch has to ask what kind of ray a lane holds (as in k_trace_flat) */
    auto issue = [&](uint64_t rid, RayWords& w) -> bool {
        if (MODE != MODE_MIXED) return issue_ray<MODE>(a, rid, w);
        freshShadow = rid >= numJobs;
        return freshShadow ? issue_ray<MODE_SHADOW>(a, rid - numJobs, w) : issue_ray<MODE_JOBS>(a, rid, w);
    };
    auto store = [&](bool h, float t, int32_t s1, int32_t s2, int32_t s3, float al, float be, float ga, uint32_t orig) {
        if (MODE != MODE_MIXED) store_result<MODE>(a, id, tag, h, t, s1, s2, s3, al, be, ga, orig);
        else if (shadowLane) store_result<MODE_SHADOW>(a, id - numJobs, tag, h, t, s1, s2, s3, al, be, ga, orig);
        else store_result<MODE_JOBS>(a, id, tag, h, t, s1, s2, s3, al, be, ga, orig);
    };
    auto start_ray = [&](const RayWords& w) {
        if (MODE != MODE_MIXED) finish_ray<MODE>(w, r, exclude, tag);
        else { shadowLane = freshShadow; if (shadowLane) finish_ray<MODE_SHADOW>(w, r, exclude, tag); else finish_ray<MODE_JOBS>(w, r, exclude, tag); }
        if (COUNT) { nRay++; nBox++; }
        hit = false; ranOff = false; sp = 0;
        rx = { r.rcpD.x, r.rcpD.x }; ry = { r.rcpD.y, r.rcpD.y }; rz = { r.rcpD.z, r.rcpD.z };
        float ftmin;
        bool s = box_static(r, a.rootC[0], a.rootC[1], a.rootC[2], rAx, rAy, rAz, rBx, rBy, rBz, ftmin);
        if (s && r.max_t > ftmin) { active = true; cur = a.rootRef; }
        else store(false, r.max_t, -1, -1, -1, 0, 0, 0, 0);
    };
    /* the one push site: the sibling waits with its tmin; `max_t > tmin` is re-evaluated when it is popped */
    auto push = [&](uint32_t ref, float tmin) {
        const uint16_t tb = hbits((h16)tmin);                 /* exact: tmin came from an fp16 */
        if (sp < PT_STACK_LDS) { sRef[wave][sp][lane] = ref; sTmin[wave][sp][lane] = tb; }
        else a.overflow[PT_SPILL_AT(sp - PT_STACK_LDS)] = make_uint2(ref, tb);
        sp++;
    };

    for (;;) {
        const bool canTake = !active && !pend;
        uint64_t idleMask = __ballot(canTake);
        int nIdle = __popcll(idleMask);
        if (nIdle == 64 && pool.exhausted()) break;
        bool fresh = false;
        RayWords w;
        if (MODE != MODE_SHADOW && pend) { id = pendIdx; pend = false; fresh = issue(id, w); }
        if (!pool.exhausted() && (nIdle >= (int)a.refillMin || nIdle == 64)) {
            uint64_t nid = pool.claim(a.fetchCounter, canTake, idleMask, nIdle);
            if (canTake && nid != ~0ull) {
                if (MODE != MODE_SHADOW && a.order) { pendIdx = a.order[nid]; pend = true; }
                else { id = nid; fresh = issue(id, w); }
            }
        }

        if (active) {
        const bool leaf = (cur & PT_LEAF_BIT) != 0;
        const uint32_t unit = cur & PT_UNIT_MASK;
        const uint32_t off = unit * 64u;
        u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
        u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0);
        u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48, 0, 0);
        u32x4 q4 = { 0u, 0u, 0u, 0u }, q5 = q4, q6 = q4, q7 = q4;
        if (!leaf) {
            q4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 64, 0, 0);
            q5 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 80, 0, 0);
            q6 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 96, 0, 0);
            q7 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 112, 0, 0);
        }
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        asm volatile("" : "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
        if (COUNT) nRec += leaf ? 1 : 2;
        bool needPop = true, done = false;

        if (leaf) {
            if ((q3.y & 1u) != 0) ranOff = true;                             /* the reference array's last node was entered */
            const uint32_t leafId = q3.x;
            if (exclude == 0 || exclude != leafId) {                         /* intersections.glsl:81 */
                if (COUNT) nTri++;
                float t, al, be, ga;
                F3 p1 = { __uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z) };
                F3 e1 = { __uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z) };
                F3 e2 = { __uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z) };
                if (tri_test(r, p1, e1, e2, t, al, be, ga)) {
                    r.max_t = t; alpha = al; beta = be; gamma = ga;
                    hit = true;
                    if (MODE != MODE_SHADOW) { hitI1 = (int32_t)q0.w; hitI2 = (int32_t)q1.w; hitI3 = (int32_t)q2.w; }
                    if (MODE == MODE_JOBS || MODE == MODE_MIXED) hitLeaf = leafId;
                    if (ANYHIT || (MODE == MODE_MIXED && shadowLane)) { done = true; needPop = false; }
                }
            }
        } else {
            /* words 0..9 pair (L, R), 10..19 pair (LL, LR), 20..29 pair (RL, RR), 30..31 links */
            const uint32_t d0 = q2.y;
            bool goRight = (cur & PT_HALF_RIGHT) != 0;           /* a popped half says which; stage A decides otherwise */
            bool enterHalf = (cur & PT_HALF_BIT) != 0;
            /* starts of the four targets: T0 right behind the record, T1..T3 from the link word */
            const uint32_t w0 = q7.z, w1 = q7.w;
            const uint32_t sh = (w1 >> 31) ? 5u : 0u;
            const uint32_t u0 = unit + 2u;
            const uint32_t u1 = unit + ((w0 & 0x1FFFFFu) << sh);
            const uint32_t u2 = unit + ((((w0 >> 21) | (w1 << 11)) & 0x1FFFFFu) << sh);
            const uint32_t u3 = unit + (((w1 >> 10) & 0x1FFFFFu) << sh);
            if (!enterHalf) {                                    /* stage A: the two-box walk's visit of X */
                if (COUNT) nBox += 2;
                bool sL, sR; float tL, tR;
                box_pair(r, rx, ry, rz, as_f2(q0.x, q0.y), as_f2(q0.z, q0.w), as_f2(q1.x, q1.y), as_h2(q1.z), as_h2(q1.w), as_h2(q2.x),
                         rebuild_b(q1.z, d0, 0), rebuild_b(q1.w, d0, 1), rebuild_b(q2.x, d0, 2), sL, sR, tL, tR);
                const bool pL = sL && r.max_t > tL, pR = sR && r.max_t > tR;
                if (pL || pR) {
                    const bool leafL = (d0 & 0x8000u) != 0, leafR = (d0 & 0x80000000u) != 0;
                    goRight = (ORDERED && pL && pR && tR < tL) || !pL;
                    if (pL && pR) {                              /* the other half waits */
                        const bool farLeaf = goRight ? leafL : leafR;
                        const uint32_t farTri = (goRight ? u0 : u2) | PT_LEAF_BIT;
                        const uint32_t farHalf = unit | PT_HALF_BIT | (goRight ? 0u : PT_HALF_RIGHT);
                        push(farLeaf ? farTri : farHalf, goRight ? tL : tR);
                    }
                    if (goRight ? leafR : leafL) { cur = (goRight ? u2 : u0) | PT_LEAF_BIT; needPop = false; }
                    else enterHalf = true;
                }
            }
            if (enterHalf) {                                     /* stage B: the two-box walk's visit of the half */
                if (COUNT) nBox += 2;
                const uint32_t c0 = goRight ? q5.x : q2.z, c1 = goRight ? q5.y : q2.w, c2 = goRight ? q5.z : q3.x;
                const uint32_t c3 = goRight ? q5.w : q3.y, c4 = goRight ? q6.x : q3.z, c5 = goRight ? q6.y : q3.w;
                const uint32_t h0 = goRight ? q6.z : q4.x, h1 = goRight ? q6.w : q4.y, h2 = goRight ? q7.x : q4.z;
                const uint32_t dd = goRight ? q7.y : q4.w;
                bool sA, sB; float tA, tB;
                box_pair(r, rx, ry, rz, as_f2(c0, c1), as_f2(c2, c3), as_f2(c4, c5), as_h2(h0), as_h2(h1), as_h2(h2),
                         rebuild_b(h0, dd, 0), rebuild_b(h1, dd, 1), rebuild_b(h2, dd, 2), sA, sB, tA, tB);
                const bool pA = sA && r.max_t > tA, pB = sB && r.max_t > tB;
                const uint32_t refA = (goRight ? u2 : u0) | ((dd & 0x8000u) ? PT_LEAF_BIT : 0u);
                const uint32_t refB = (goRight ? u3 : u1) | ((dd & 0x80000000u) ? PT_LEAF_BIT : 0u);
                const bool bFirst = (ORDERED && pA && pB && tB < tA) || !pA;
                if (pA && pB) push(bFirst ? refA : refB, bFirst ? tA : tB);
                if (pA || pB) { cur = bFirst ? refB : refA; needPop = false; }
            }
        }

        /* pending siblings come off the stack until one still passes `max_t > tmin` */
        while (needPop) {
            if (sp == 0) { done = true; break; }
            sp--;
            const int e = sp < PT_STACK_LDS ? sp : PT_STACK_LDS - 1;
            uint32_t ref = sRef[wave][e][lane]; uint16_t tb = sTmin[wave][e][lane];
            asm volatile("" : "+v"(ref));
            if (sp >= PT_STACK_LDS) { uint2 o = a.overflow[PT_SPILL_AT(sp - PT_STACK_LDS)]; ref = o.x; tb = (uint16_t)o.y; }
            if (r.max_t > (float)hfrombits(tb)) { cur = ref; needPop = false; }
        }

        if (done) {
            uint32_t orig = 0u;
            if ((MODE == MODE_JOBS || (MODE == MODE_MIXED && !shadowLane)) && ranOff && hit) orig = hitLeaf;
            store(hit, r.max_t, hitI1, hitI2, hitI3, alpha, beta, gamma, orig);
            active = false;
        }
        }   /* if (active) */

        if (fresh) start_ray(w);
    }
    if (COUNT) {
        atomicAdd(a.stats + 0, nBox); atomicAdd(a.stats + 1, nTri); atomicAdd(a.stats + 2, nRec); atomicAdd(a.stats + 3, nRay);
    }
}

/* ---- treelet walk, six boxes per step (the frame passes) -------------------------------------------
 *
 * The same records walked with all six box tests at once (three packed pair tests per step) and nothing fetched twice:
 *
 *   - L and R are judged now, with the max_t of now, as the two-box walk judges them when it visits the node;
 *   - the half entered first (the nearer one if ORDERED, else L) is entered at once: a leaf goes to its triangle; of an
 *     interior half's children, judged now as well (the two-box walk would visit that half next, with the same max_t),
 *     the first is entered and the second waits on the stack with its tmin, re-checked against max_t when it is popped;
 *   - the other half waits.  The two-box walk would pop it later, check max_t > its tmin with the max_t of THEN, fetch it
 *     and judge its children with that same max_t.  Here its children's boxes are already known, so they wait in its
 *     place: each with its own tmin as gate, the one to be popped first (PT_PAIR_BIT if its sibling lies beneath it)
 *     also with the half's tmin as parent gate.  When it is popped, the parent gate is evaluated with the max_t of that
 *     moment - the moment the two-box walk pops the half; if it fails, the sibling beneath goes with it; if it holds,
 *     the sibling stays with its own gate only (the half was entered, and max_t only shrinks from there on).  The
 *     order among the two children is by tmin (ORDERED) exactly as the two-box walk orders two children that both
 *     pass: if only one of them would pass at that time it is the one with the smaller tmin, which is on top.
 *
 * Leaves tested, their order and the max_t each is tested with are again those of k_trace_flat<ORDERED>: identical bits.
 * A level of this walk can leave three entries behind where two levels of the two-box walk leave two; the spill area is
 * sized for that.  Which of the two treelet walks runs where is a matter of measurement (5.8 M triangles): a 2^26-ray batch
 * takes 28.0 ms with this step and 26.8 with k_trace_tree's (the third pair test and the three push sites are most of its
 * 474 vector instructions, against 294); 1080p x 8 frames run at 186 spp/s with this one and 176 with k_trace_tree's, whose
 * second fetch of a record is one more dependent round trip in launches that wait for their longest rays. */
template <int MODE, bool ANYHIT, bool ORDERED, bool COUNT>
__global__ __launch_bounds__(PT_BLOCK) void k_trace_tree6(TraceArgs a)
{
    __shared__ uint32_t sRef[PT_WAVES_PER_BLOCK][PT_STACK_LDS][64];
    __shared__ uint32_t sGate[PT_WAVES_PER_BLOCK][PT_STACK_LDS][64];     /* own tmin | parent's tmin << 16, fp16 bits */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gid = (uint64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    const uint64_t totalLanes = (uint64_t)gridDim.x * PT_BLOCK;
    const uint64_t numJobs = a.countPtr ? (uint64_t)*a.countPtr : a.count;       /* MODE_MIXED: ids below this are jobs, the rest shadow rays */
    const uint64_t total = numJobs + (MODE == MODE_MIXED ? (uint64_t)*a.countPtr2 : 0ull);
    if (MODE != MODE_MIXED && total < a.onlyFrom) return;      /* a list this short is the one-ray-per-wavefront launch's */
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.treelets, 0, (int)a.treeletBytes, 0x00020000);
    const uint32_t NO_GATE = 0xFC00u;                                   /* -inf: max_t > -inf always holds */

    RayPool pool; pool.init(a, total, gridDim.x * PT_WAVES_PER_BLOCK);
    bool active = false, pend = false;
    bool shadowLane = MODE == MODE_SHADOW, freshShadow = false;      /* MODE_MIXED: what kind of ray the lane walks / has just loaded */
    uint32_t pendIdx = 0;
    RayRegs r; uint32_t exclude = 0, tag = 0; uint64_t id = 0;
    h16x2 rx = { (h16)0.0f, (h16)0.0f }, ry = rx, rz = rx;
    uint32_t cur = 0; int sp = 0;
    bool hit = false, ranOff = false;
    int32_t hitI1 = -1, hitI2 = -1, hitI3 = -1; uint32_t hitLeaf = 0;
    float alpha = 0, beta = 0, gamma = 0;
    unsigned long long nBox = 0, nTri = 0, nRec = 0, nRay = 0;

    const h16 rAx = hfrombits(a.rootH[0]), rAy = hfrombits(a.rootH[1]), rAz = hfrombits(a.rootH[2]);
    const h16 rBx = hfrombits(a.rootH[3]), rBy = hfrombits(a.rootH[4]), rBz = hfrombits(a.rootH[5]);

    /* the three places where a mixed launch has to ask what kind of ray a lane holds (as in k_trace_flat) */
    auto issue = [&](uint64_t rid, RayWords& w) -> bool {
        if (MODE != MODE_MIXED) return issue_ray<MODE>(a, rid, w);
        freshShadow = rid >= numJobs;
        return freshShadow ? issue_ray<MODE_SHADOW>(a, rid - numJobs, w) : issue_ray<MODE_JOBS>(a, rid, w);
    };
    auto store = [&](bool h, float t, int32_t s1, int32_t s2, int32_t s3, float al, float be, float ga, uint32_t orig) {
        if (MODE != MODE_MIXED) store_result<MODE>(a, id, tag, h, t, s1, s2, s3, al, be, ga, orig);
        else if (shadowLane) store_result<MODE_SHADOW>(a, id - numJobs, tag, h, t, s1, s2, s3, al, be, ga, orig);
        else store_result<MODE_JOBS>(a, id, tag, h, t, s1, s2, s3, al, be, ga, orig);
    };
    auto start_ray = [&](const RayWords& w) {
        if (MODE != MODE_MIXED) finish_ray<MODE>(w, r, exclude, tag);
        else { shadowLane = freshShadow; if (shadowLane) finish_ray<MODE_SHADOW>(w, r, exclude, tag); else finish_ray<MODE_JOBS>(w, r, exclude, tag); }
        if (COUNT) { nRay++; nBox++; }
        hit = false; ranOff = false; sp = 0;
        rx = { r.rcpD.x, r.rcpD.x }; ry = { r.rcpD.y, r.rcpD.y }; rz = { r.rcpD.z, r.rcpD.z };
        float ftmin;
        bool s = box_static(r, a.rootC[0], a.rootC[1], a.rootC[2], rAx, rAy, rAz, rBx, rBy, rBz, ftmin);
        if (s && r.max_t > ftmin) { active = true; cur = a.rootRef; }
        else store(false, r.max_t, -1, -1, -1, 0, 0, 0, 0);
    };
    auto push = [&](uint32_t ref, uint32_t gates) {
        if (sp < PT_STACK_LDS) { sRef[wave][sp][lane] = ref; sGate[wave][sp][lane] = gates; }
        else a.overflow[PT_SPILL_AT(sp - PT_STACK_LDS)] = make_uint2(ref, gates);
        sp++;
    };

    for (;;) {
        const bool canTake = !active && !pend;
        uint64_t idleMask = __ballot(canTake);
        int nIdle = __popcll(idleMask);
        if (nIdle == 64 && pool.exhausted()) break;
        bool fresh = false;
        RayWords w;
        if (MODE != MODE_SHADOW && pend) { id = pendIdx; pend = false; fresh = issue(id, w); }
        if (!pool.exhausted() && (nIdle >= (int)a.refillMin || nIdle == 64)) {
            uint64_t nid = pool.claim(a.fetchCounter, canTake, idleMask, nIdle);
            if (canTake && nid != ~0ull) {
                if (MODE != MODE_SHADOW && a.order) { pendIdx = a.order[nid]; pend = true; }
                else { id = nid; fresh = issue(id, w); }
            }
        }

        if (active) {
        const bool leaf = (cur & PT_LEAF_BIT) != 0;
        const uint32_t unit = cur & 0x03FFFFFFu;
        const uint32_t off = unit * 64u;
        u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
        u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0);
        u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48, 0, 0);
        u32x4 q4 = { 0u, 0u, 0u, 0u }, q5 = q4, q6 = q4, q7 = q4;
        if (!leaf) {
            q4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 64, 0, 0);
            q5 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 80, 0, 0);
            q6 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 96, 0, 0);
            q7 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 112, 0, 0);
        }
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        asm volatile("" : "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
        if (COUNT) nRec += leaf ? 1 : 2;
        bool needPop = true, done = false;

        if (leaf) {
            if ((q3.y & 1u) != 0) ranOff = true;                             /* the reference array's last node was entered */
            const uint32_t leafId = q3.x;
            if (exclude == 0 || exclude != leafId) {                         /* intersections.glsl:81 */
                if (COUNT) nTri++;
                float t, al, be, ga;
                F3 p1 = { __uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z) };
                F3 e1 = { __uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z) };
                F3 e2 = { __uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z) };
                if (tri_test(r, p1, e1, e2, t, al, be, ga)) {
                    r.max_t = t; alpha = al; beta = be; gamma = ga;
                    hit = true;
                    if (MODE != MODE_SHADOW) { hitI1 = (int32_t)q0.w; hitI2 = (int32_t)q1.w; hitI3 = (int32_t)q2.w; }
                    if (MODE == MODE_JOBS || MODE == MODE_MIXED) hitLeaf = leafId;
                    if (ANYHIT || (MODE == MODE_MIXED && shadowLane)) { done = true; needPop = false; }
                }
            }
        } else {
            /* words 0..9 pair (L, R), 10..19 pair (LL, LR), 20..29 pair (RL, RR), 30..31 links */
            const uint32_t d0 = q2.y, d1 = q4.w, d2 = q7.y;
            bool sL, sR, sLa, sLb, sRa, sRb; float tL, tR, tLa, tLb, tRa, tRb;
            box_pair(r, rx, ry, rz, as_f2(q0.x, q0.y), as_f2(q0.z, q0.w), as_f2(q1.x, q1.y), as_h2(q1.z), as_h2(q1.w), as_h2(q2.x),
                     rebuild_b(q1.z, d0, 0), rebuild_b(q1.w, d0, 1), rebuild_b(q2.x, d0, 2), sL, sR, tL, tR);
            box_pair(r, rx, ry, rz, as_f2(q2.z, q2.w), as_f2(q3.x, q3.y), as_f2(q3.z, q3.w), as_h2(q4.x), as_h2(q4.y), as_h2(q4.z),
                     rebuild_b(q4.x, d1, 0), rebuild_b(q4.y, d1, 1), rebuild_b(q4.z, d1, 2), sLa, sLb, tLa, tLb);
            box_pair(r, rx, ry, rz, as_f2(q5.x, q5.y), as_f2(q5.z, q5.w), as_f2(q6.x, q6.y), as_h2(q6.z), as_h2(q6.w), as_h2(q7.x),
                     rebuild_b(q6.z, d2, 0), rebuild_b(q6.w, d2, 1), rebuild_b(q7.x, d2, 2), sRa, sRb, tRa, tRb);
            const bool leafL = (d0 & 0x8000u) != 0, leafR = (d0 & 0x80000000u) != 0;
            if (COUNT) nBox += 2 + (leafL ? 0 : 2) + (leafR ? 0 : 2);
            /* starts of the four targets */
            const uint32_t w0 = q7.z, w1 = q7.w;
            const uint32_t sh = (w1 >> 31) ? 5u : 0u;
            const uint32_t u0 = unit + 2u;
            const uint32_t u1 = unit + ((w0 & 0x1FFFFFu) << sh);
            const uint32_t u2 = unit + ((((w0 >> 21) | (w1 << 11)) & 0x1FFFFFu) << sh);
            const uint32_t u3 = unit + (((w1 >> 10) & 0x1FFFFFu) << sh);
            /* references as the walk uses them: unit | leaf bit.  A leaf half has one target, its triangle */
            const uint32_t refLa = u0 | ((leafL || (d1 & 0x8000u)) ? PT_LEAF_BIT : 0u), refLb = u1 | ((d1 & 0x80000000u) ? PT_LEAF_BIT : 0u);
            const uint32_t refRa = u2 | ((leafR || (d2 & 0x8000u)) ? PT_LEAF_BIT : 0u), refRb = u3 | ((d2 & 0x80000000u) ? PT_LEAF_BIT : 0u);

            const bool pL = sL && r.max_t > tL, pR = sR && r.max_t > tR;
            const bool rightFirst = (ORDERED && pL && pR && tR < tL) || !pL;
            /* the half entered first (1) and the half that waits (2) */
            const bool leaf1 = rightFirst ? leafR : leafL, leaf2 = rightFirst ? leafL : leafR;
            const uint32_t ref1a = rightFirst ? refRa : refLa, ref1b = rightFirst ? refRb : refLb;
            const uint32_t ref2a = rightFirst ? refLa : refRa, ref2b = rightFirst ? refLb : refRb;
            const bool s1a = rightFirst ? sRa : sLa, s1b = rightFirst ? sRb : sLb, s2a = rightFirst ? sLa : sRa, s2b = rightFirst ? sLb : sRb;
            const float t1a = rightFirst ? tRa : tLa, t1b = rightFirst ? tRb : tLb, t2a = rightFirst ? tLa : tRa, t2b = rightFirst ? tLb : tRb;
            const float t2 = rightFirst ? tL : tR;

            if (pL && pR) {                                    /* the second half waits: as itself if a leaf, else as its children */
                const uint32_t g2 = (uint32_t)hbits((h16)t2);                                /* exact: every tmin came from an fp16 */
                if (leaf2) push(ref2a, g2 | (NO_GATE << 16));
                else {
                    const bool bFirst = ORDERED && s2a && s2b && t2b < t2a;
                    if (s2a && s2b) {
                        push(bFirst ? ref2a : ref2b, (uint32_t)hbits((h16)(bFirst ? t2a : t2b)) | (NO_GATE << 16));
                        push((bFirst ? ref2b : ref2a) | PT_PAIR_BIT, (uint32_t)hbits((h16)(bFirst ? t2b : t2a)) | (g2 << 16));
                    } else if (s2a || s2b) {
                        push(s2a ? ref2a : ref2b, (uint32_t)hbits((h16)(s2a ? t2a : t2b)) | (g2 << 16));
                    }
                }
            }
            if (pL || pR) {
                if (leaf1) { cur = ref1a; needPop = false; }
                else {
                    const bool pa = s1a && r.max_t > t1a, pb = s1b && r.max_t > t1b;
                    const bool bFirst = (ORDERED && pa && pb && t1b < t1a) || !pa;
                    if (pa && pb) push(bFirst ? ref1a : ref1b, (uint32_t)hbits((h16)(bFirst ? t1a : t1b)) | (NO_GATE << 16));
                    if (pa || pb) { cur = bFirst ? ref1b : ref1a; needPop = false; }
                }
            }
        }

        while (needPop) {
            if (sp == 0) { done = true; break; }
            sp--;
            const int e = sp < PT_STACK_LDS ? sp : PT_STACK_LDS - 1;
            uint32_t ref = sRef[wave][e][lane], gates = sGate[wave][e][lane];
            asm volatile("" : "+v"(ref));
            if (sp >= PT_STACK_LDS) { uint2 o = a.overflow[PT_SPILL_AT(sp - PT_STACK_LDS)]; ref = o.x; gates = o.y; }
            const bool parentOk = r.max_t > (float)hi16(gates);
            if (parentOk && r.max_t > (float)lo16(gates)) { cur = ref & ~PT_PAIR_BIT; needPop = false; }
            else if (!parentOk && (ref & PT_PAIR_BIT)) sp--;                /* its sibling beneath hangs on the same parent */
        }

        if (done) {
            uint32_t orig = 0u;
            if ((MODE == MODE_JOBS || (MODE == MODE_MIXED && !shadowLane)) && ranOff && hit) orig = hitLeaf;
            store(hit, r.max_t, hitI1, hitI2, hitI3, alpha, beta, gamma, orig);
            active = false;
        }
        }   /* if (active) */

        if (fresh) start_ray(w);
    }
    if (COUNT) {
        atomicAdd(a.stats + 0, nBox); atomicAdd(a.stats + 1, nTri); atomicAdd(a.stats + 2, nRec); atomicAdd(a.stats + 3, nRay);
    }
}

/* ---- certified walk: the reference's answer without the reference's visit order ----------------
 *
 * The reference's closest hit depends on its visit order only because its fp16 box test is not conservative: a
 * subtree can be skipped (max_t <= tmin of its box) although it holds a closer triangle, and whether that happens
 * depends on what max_t was when the walk came by.  k_trace_cert walks near child first like k_trace_flat<ORDERED>,
 * with two changes that make its answer the reference's for almost every ray, and tell which rays are left:
 *
 *  1. it skips a subtree only when that is safe in every order: the box's fp16 verdict minus its max_t part (which
 *     does not depend on order: a subtree the reference can never enter is not entered here either) must pass, and
 *     the subtree is skipped on distance only if a lower bound of the distance at which the ray enters the box the
 *     builder made (box_pair_bound: the box test's own near-plane distances less everything that separates them from
 *     the exact ones, less PT_CERT_SLACK of itself) lies beyond max_t.  Every triangle of a subtree lies inside its
 *     box, so nothing skipped this way can be nearer than what the walk already holds - as long as the t the triangle
 *     test computes is where the triangle is, to within that slack.  That is the one assumption.  It holds unless the
 *     test's determinant d.(e1 x e2) is lost to rounding: for a triangle so thin that e1 x e2 itself is noise (the
 *     flattening pass marks the subtrees holding one, sine of the angle between e1 and e2 below 2^-14, and they
 *     are never skipped on distance), or for a ray that lies in a triangle's plane to a few parts in 10^7 (measured:
 *     2 to 6 of 10^7 rays constructed to lie in triangle planes; none among 10^8 others, vertex- and edge-aimed ones
 *     included).  With it, the walk ends with the nearest triangle X among all the reference could ever reach,
 *     whatever the order.
 *  2. it keeps, for the hit X it holds, the largest fp16 tmin G among the boxes on the way down to X, and the smallest
 *     t among all OTHER triangles it saw the ray hit.  The reference enters a box when its max_t of that moment exceeds
 *     the box's tmin, and its max_t is the ray's limit or the t of a triangle it has found.  So if no other triangle
 *     the ray hits has t <= G, and G is below the ray's own limit, the reference enters every box on the way to X
 *     whenever it comes by, tests X with a max_t that is not below t of X, accepts it, and accepts nothing after it
 *     but an exact tie: its answer is X, with the same t (same arithmetic in the triangle test).  To be sure of having
 *     seen every triangle with t <= G the walk skips on distance only beyond max(t of X, G), not beyond t of X.
 *
 * A ray is flagged, and traced again by the reference-order walk, when another hit lies at or before G (about one ray
 * in a thousand: hits on an edge two triangles share, doubled surfaces), when G reaches the ray's limit, when two
 * triangles tied for the final t, or when its direction has a zero or non-finite component or its origin is not finite
 * (the bound's arithmetic assumes neither).
 */
#ifndef PT_CERT_SLACK
#define PT_CERT_SLACK 0x1p-7f
#ifndef PT_CERT_NT_LOAD
#define PT_CERT_NT_LOAD false
#endif
#ifndef PT_CERT_NT_STORE
#define PT_CERT_NT_STORE (PT_STREAM_NT != 0)
#endif
#endif
/* Lower bounds of the distances at which the ray enters the two boxes of one interior record, from the box test's own
 * fp16 values: box_pair with, per axis, the near slab distance min(t0, t1) it computes anyway, less a bound E of everything
 * that separates that fp16 value from the exact (a - o) / d, (b - o) / d of the builder's fp32 box [a, b]:
 *
 *   digest      ha = f16(a - c):  |ha - (a - c)| <= |a - c| (2^-11 + 2^-23), or 2^-25 below the normal range;
 *   origin      nox = f16(c - o): the same for c - o;
 *   sum         s = ha + nox in fp16: one rounding, 2^-11 relative (fp16 sums below the normal range are exact);
 *   direction   rx = f16(1 / d):  (1 / d)(1 + 2^-11 + 2^-24);
 *   product     t = s rx in fp16: 2^-11 relative, or 2^-25 below the normal range.
 *
 * With H = max(|ha|, |hb|) + |nox| that is  |t - T| <= H |rx| 2^-9 (1 + 2^-8) + |rx| 2^-23 (1 + 2^-8) + 2^-25  for both
 * planes of the slab, and  E = (|nox| + W) k,  k = max(|rx| 2^-8, 2^-10) (a per-ray constant, exact in fp16),
 * W >= max(|ha|, |hb|) + 2^-12 (a per-box constant the flattening pass rounds up to fp16: the largest over the three
 * axes, so that one value serves all of them), is at least twice that, which pays for the two roundings of E itself and
 * for the one of near - E (a single fused operation: this is the bound's own arithmetic, not the reference's).  The entry
 * distance is the largest near-plane distance, and the largest of the three lower bounds is a lower bound of it; where an
 * axis yields NaN or -inf (0 x inf, an overflow) the IEEE maximum leaves it out, which keeps what is left a bound.  An
 * fp16 tmin of +inf (a slab beyond the fp16 range) gives a bound of +inf: the reference's own `max_t > tmin` can never
 * hold for that box either.  W = +inf (half extents beyond the fp16 range, or a subtree the flattening pass marks as never
 * to be skipped on distance) gives -inf or NaN on every axis, and neither compares greater than anything: never skipped.
 * Thirteen packed operations for both boxes on top of the box test. */
__device__ __forceinline__ void box_pair_bound(const RayRegs& r, h16x2 rx, h16x2 ry, h16x2 rz, h16x2 kx, h16x2 ky, h16x2 kz,
                                               f32x2 cx, f32x2 cy, f32x2 cz, h16x2 ax, h16x2 ay, h16x2 az, h16x2 bx, h16x2 by, h16x2 bz,
                                               h16x2 wmax, bool& sL, bool& sR, float& tminL, float& tminR, h16x2& bound)
{
    auto habs = [](h16x2 v) { return as_h2(__builtin_bit_cast(uint32_t, v) & 0x7FFF7FFFu); };
    f32x2 dx = { cx.x - r.o.x, cx.y - r.o.x }, dy = { cy.x - r.o.y, cy.y - r.o.y }, dz = { cz.x - r.o.z, cz.y - r.o.z };
    h16x2 nox = __builtin_convertvector(dx, h16x2), noy = __builtin_convertvector(dy, h16x2), noz = __builtin_convertvector(dz, h16x2);
    h16x2 t0 = (ax + nox) * rx, t1 = (bx + nox) * rx;
    h16x2 near = __builtin_elementwise_min(t0, t1);
    h16x2 tmin = near, tmax = __builtin_elementwise_max(t0, t1);
    bound = __builtin_elementwise_fma(-(habs(nox) + wmax), kx, near);
    t0 = (ay + noy) * ry; t1 = (by + noy) * ry;
    near = __builtin_elementwise_min(t0, t1);
    tmin = __builtin_elementwise_max(tmin, near);
    tmax = __builtin_elementwise_min(tmax, __builtin_elementwise_max(t0, t1));
    bound = __builtin_elementwise_max(bound, __builtin_elementwise_fma(-(habs(noy) + wmax), ky, near));
    t0 = (az + noz) * rz; t1 = (bz + noz) * rz;
    near = __builtin_elementwise_min(t0, t1);
    tmin = __builtin_elementwise_max(tmin, near);
    tmax = __builtin_elementwise_min(tmax, __builtin_elementwise_max(t0, t1));
    bound = __builtin_elementwise_max(bound, __builtin_elementwise_fma(-(habs(noz) + wmax), kz, near));
    const float fminL = (float)tmin.x, fminR = (float)tmin.y, fmaxL = (float)tmax.x, fmaxR = (float)tmax.y;
    sL = (fmaxL > 0.0f || fminL > 0.0f) && fmaxL >= fminL && r.min_t < fmaxL;
    sR = (fmaxR > 0.0f || fminR > 0.0f) && fmaxR >= fminR && r.min_t < fmaxR;
    tminL = fminL; tminR = fminR;
}
/* k of box_pair_bound for one component of f16(1 / d), in both halves */
__device__ __forceinline__ h16x2 bound_factor(h16 rcp)
{
    const h16 k = (h16)fmaxf(fabsf((float)rcp) * 0x1p-8f, 0x1p-10f);          /* exact: a power of two times an fp16 value */
    return h16x2{ k, k };
}
/* The slack the walk allows the triangle test (see PT_CERT_SLACK), taken off both bounds in fp16: 9/8 of it, so that the
 * rounding of this operation stays inside what was to be taken off. */
__device__ __forceinline__ h16x2 bound_slack(h16x2 b)
{
    const h16x2 s = { (h16)(PT_CERT_SLACK * 1.125f), (h16)(PT_CERT_SLACK * 1.125f) };
    return __builtin_elementwise_fma(-as_h2(__builtin_bit_cast(uint32_t, b) & 0x7FFF7FFFu), s, b);
}

/* MODE_RAYS, MODE_JOBS, or MODE_MIXED: a job list and, behind it, a shadow queue in one pool (see k_trace_flat).  A lane with
 * a shadow ray walks near child first with the reference's own culling and stops at the first hit: nothing about an any-hit
 * verdict depends on the order, so there is nothing to certify. */
#ifndef PT_CERT_STACK_LDS
#define PT_CERT_STACK_LDS PT_STACK_LDS
#endif
#ifndef PT_CERT_STACK_LDS_RAYS
#define PT_CERT_STACK_LDS_RAYS 11     /* batches: seven workgroups of 22.5 KB fit a CU's 160 KB, seven of 24.5 KB do not */
#endif
#ifndef PT_CERT_LEAF_MIN
#define PT_CERT_LEAF_MIN 8       /* batches: lanes at a leaf wait for this many before the triangle tests run (4 / 8 / 12 / 16: 28.8 / 28.4 / 28.4 / 28.7 ms; 1 = never wait: 29.3) */
#endif
#ifdef PT_CERT_WAVES
#define PT_CERT_ATTR __attribute__((amdgpu_waves_per_eu(PT_CERT_WAVES, PT_CERT_WAVES)))
#else
#define PT_CERT_ATTR
#endif
template <int MODE, bool COUNT>
__global__ __launch_bounds__(PT_BLOCK) PT_CERT_ATTR void k_trace_cert(TraceArgs a)
{
    /* per entry and lane two words, 64 words apart (one two-word LDS access each way): [0] the waiting child's reference,
       [1] its entry bound (low half; fp16) and the largest fp16 tmin from the root down to it (high half) */
    constexpr int STK = MODE == MODE_RAYS ? PT_CERT_STACK_LDS_RAYS : PT_CERT_STACK_LDS;        /* entries of a lane's stack held in LDS */
    __shared__ uint32_t sStk[PT_WAVES_PER_BLOCK][STK][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t gid = (uint64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    const uint64_t totalLanes = (uint64_t)gridDim.x * PT_BLOCK;
    const uint64_t numJobs = a.countPtr ? (uint64_t)*a.countPtr : a.count;
    const uint64_t total = numJobs + (MODE == MODE_MIXED ? (uint64_t)*a.countPtr2 : 0ull);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.records, 0, (int)a.recordBytes, 0x00020000);

    RayPool pool; pool.init(a, total, gridDim.x * PT_WAVES_PER_BLOCK);
    bool active = false, pend = false;
    bool shadowLane = false, freshShadow = false;      /* MODE_MIXED: what kind of ray the lane walks / has just loaded */
    uint32_t pendIdx = 0;
    RayRegs r; uint32_t exclude = 0, tag = 0;
    uint32_t id = 0;                      /* a launch has fewer than 2^32 rays: one register, not two */
    /* per axis one register: f16(1 / d) in the low half, the factor k of the entry bound's error term (box_pair_bound) in the
       high half; the packed operations take either half for both of theirs */
    h16x2 rkx = { (h16)0.0f, (h16)0.0f }, rky = rkx, rkz = rkx;
    uint32_t cur = 0; int sp = 0;
    bool hit = false, tie = false, nearLast = false;
    int32_t hitI1 = -1, hitI2 = -1, hitI3 = -1;
    uint32_t hitLeaf = 0;                 /* jobs: the hit leaf's index in the reference array */
    float alpha = 0, beta = 0, gamma = 0;
    float pathGate = 0, hitGate = 0;      /* largest fp16 tmin on the way to the current record / to the hit held */
    float limitReg = 0, otherT = 0;       /* the ray's own max_t (batches; a job's is the constant of trace.comp:190); smallest t of any other triangle the ray was seen to hit */
#define limit0 (MODE == MODE_RAYS ? limitReg : 100000.0f)
    unsigned long long nBox = 0, nTri = 0, nRec = 0, nRay = 0;
    const float NEG_INF = -__builtin_huge_valf();

    const h16 rAx = hfrombits(a.rootH[0]), rAy = hfrombits(a.rootH[1]), rAz = hfrombits(a.rootH[2]);
    const h16 rBx = hfrombits(a.rootH[3]), rBy = hfrombits(a.rootH[4]), rBz = hfrombits(a.rootH[5]);

    auto flag_ray = [&]() { a.flagList[atomicAdd(a.flagCount, 1u)] = (uint32_t)id; };
    auto issue = [&](uint64_t rid, RayWords& w) -> bool {
        if (MODE != MODE_MIXED) return issue_ray<MODE, PT_CERT_NT_LOAD>(a, rid, w);
        freshShadow = rid >= numJobs;
        return freshShadow ? issue_ray<MODE_SHADOW>(a, rid - numJobs, w) : issue_ray<MODE_JOBS>(a, rid, w);
    };
    auto store = [&](bool h, uint32_t orig, bool listed) {       /* listed: the ray is on the list for the second walk (PtHitRec.listed) */
        if (MODE != MODE_MIXED) store_result<MODE, PT_CERT_NT_STORE>(a, id, tag, h, r.max_t, hitI1, hitI2, hitI3, alpha, beta, gamma, orig, listed ? 1u : 0u);
        else if (shadowLane) store_result<MODE_SHADOW>(a, id - numJobs, tag, h, r.max_t, -1, -1, -1, 0, 0, 0, 0u);
        else store_result<MODE_JOBS>(a, id, tag, h, r.max_t, hitI1, hitI2, hitI3, alpha, beta, gamma, orig, listed ? 1u : 0u);
    };

    auto start_ray = [&](const RayWords& w) {
        if (MODE != MODE_MIXED) finish_ray<MODE>(w, r, exclude, tag);
        else { shadowLane = freshShadow; if (shadowLane) finish_ray<MODE_SHADOW>(w, r, exclude, tag); else finish_ray<MODE_JOBS>(w, r, exclude, tag); }
        if (COUNT) { nRay++; nBox++; }
        hit = false; tie = false; sp = 0; pathGate = NEG_INF; hitGate = NEG_INF;
        limitReg = r.max_t; otherT = __builtin_huge_valf();
        rkx = { r.rcpD.x, bound_factor(r.rcpD.x).x }; rky = { r.rcpD.y, bound_factor(r.rcpD.y).x }; rkz = { r.rcpD.z, bound_factor(r.rcpD.z).x };
        const float invx = 1.0f / r.d.x, invy = 1.0f / r.d.y, invz = 1.0f / r.d.z;
        /* finite and non-zero, all nine: x - x is 0 exactly then, NaN for an infinity or a NaN */
        const float probe = (invx - invx) + (invy - invy) + (invz - invz) + (r.o.x - r.o.x) + (r.o.y - r.o.y) + (r.o.z - r.o.z) +
                            (r.d.x - r.d.x) + (r.d.y - r.d.y) + (r.d.z - r.d.z);
        const bool walkable = shadowLane || (probe == 0.0f && invx != 0.0f && invy != 0.0f && invz != 0.0f);
        float ftmin;
        nearLast = false;
        if (MODE == MODE_JOBS || (MODE == MODE_MIXED && !shadowLane)) {
            /* what the shade pass is told besides the hit - whether the reference's walk went on into the last node of
               its array and out at its end (intersections.glsl:109-113) - depends on that walk's max_t when it got there.
               It cannot have gone in if the order-independent part of that node's box test fails; for the other rays the
               question is settled when the walk is over (ran_off below) */
            nearLast = box_static(r, a.lastC[0], a.lastC[1], a.lastC[2], hfrombits(a.lastH[0]), hfrombits(a.lastH[1]), hfrombits(a.lastH[2]),
                                  hfrombits(a.lastH[3]), hfrombits(a.lastH[4]), hfrombits(a.lastH[5]), ftmin);
        }
        bool s = box_static(r, a.rootC[0], a.rootC[1], a.rootC[2], rAx, rAy, rAz, rBx, rBy, rBz, ftmin);
        if (!walkable) { flag_ray(); store(false, 0u, true); }
        else if (s && r.max_t > ftmin) { active = true; cur = a.rootRef; }          /* the root's test is the reference's own: it comes first in every order */
        else store(false, 0u, false);
    };

    for (;;) {
        const bool canTake = !active && !pend;
        uint64_t idleMask = __ballot(canTake);
        int nIdle = __popcll(idleMask);
        if (nIdle == 64 && pool.exhausted()) break;
        bool fresh = false;
        RayWords w;
        if (MODE == MODE_RAYS && pend) { id = pendIdx; pend = false; fresh = issue(id, w); }      /* only batches come with an order */
        if (!pool.exhausted() && (nIdle >= (int)a.refillMin || nIdle == 64)) {
            uint64_t nid = pool.claim(a.fetchCounter, canTake, idleMask, nIdle);
            if (canTake && nid != ~0ull) {
                if (MODE == MODE_RAYS && a.order) { pendIdx = a.order[nid]; pend = true; }
                else { id = (uint32_t)nid; fresh = issue(id, w); }
            }
        }

        /* Leaves wait (batches): a lane that stands at a leaf sits steps out until PT_CERT_LEAF_MIN lanes of its wavefront stand
           at one, or no lane has anything else to do, so that the triangle test and the bookkeeping behind it are issued for
           several lanes at a time instead of for two or three in nearly every step.  Every lane's own sequence of tests and
           the max_t each sees are unchanged, hence its result.  (This walk's step is the longest of the traversal kernels;
           the near-first walks, measured the same way, gain nothing: their launches wait for memory, not for instructions.) */
        bool stepNow = active;
        if (MODE == MODE_RAYS && PT_CERT_LEAF_MIN > 1) {
            const bool atLeaf = active && (cur & PT_LEAF_BIT) != 0;
            const uint64_t leafMask = __ballot(atLeaf), actMask = __ballot(active);
            if (__popcll(leafMask) < PT_CERT_LEAF_MIN && leafMask != actMask) stepNow = active && !atLeaf;
        }
        if (stepNow) {
        const bool leaf = (cur & PT_LEAF_BIT) != 0;
        const uint32_t off = (cur & ~PT_LEAF_BIT) * 64u + (leaf ? a.leafBase : 0u);
        u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16, 0, 0);
        u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0);
        u32x4 q3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 48, 0, 0);
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        if (COUNT) nRec++;
        bool needPop = true, done = false;

        if (leaf) {
            if (COUNT) nTri++;
            float t, al, be, ga;
            F3 p1 = { __uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z) };
            F3 e1 = { __uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z) };
            F3 e2 = { __uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z) };
            const bool accepted = tri_test(r, p1, e1, e2, t, al, be, ga);
            if (MODE == MODE_MIXED && shadowLane) {
                /* any hit ends the walk, except the triangle the ray starts on (intersections.glsl:81) */
                if (accepted && (exclude == 0 || exclude != q3.x)) { hit = true; r.max_t = t; done = true; needPop = false; }
            } else
            /* intersections.glsl:45 in its positive form, against the ray's own limit first */
            if (t >= r.min_t && t <= limit0 && al >= 0.0f && be >= 0.0f && ga >= 0.0f) {
                if (t <= r.max_t) {                         /* the reference's acceptance: replaces the hit held */
                    if (hit) { if (t == r.max_t) tie = true; otherT = fminf(otherT, r.max_t); }
                    r.max_t = t; hit = true; hitGate = pathGate; alpha = al; beta = be; gamma = ga;
                    hitI1 = (int32_t)q0.w; hitI2 = (int32_t)q1.w; hitI3 = (int32_t)q2.w;
                    if (MODE != MODE_RAYS) hitLeaf = q3.x;
                } else otherT = fminf(otherT, t);
            }
        } else {
            if (COUNT) nBox += 2;
            bool sL, sR; float tminL, tminR;
            const f32x2 cx = as_f2(q0.x, q0.y), cy = as_f2(q0.z, q0.w), cz = as_f2(q1.x, q1.y);
            const h16x2 ax = as_h2(q1.z), ay = as_h2(q1.w), az = as_h2(q2.x), bx = as_h2(q2.y), by = as_h2(q2.z), bz = as_h2(q2.w);
            /* the bounds are not used by a lane with a shadow ray (it culls with the reference's test below), and cost it
               nothing it could skip: they ride in the box test's packed operations */
            h16x2 bound16;
            const h16x2 rx = { rkx.x, rkx.x }, ry = { rky.x, rky.x }, rz = { rkz.x, rkz.x };
            const h16x2 kx = { rkx.y, rkx.y }, ky = { rky.y, rky.y }, kz = { rkz.y, rkz.y };
            box_pair_bound(r, rx, ry, rz, kx, ky, kz, cx, cy, cz, ax, ay, az, bx, by, bz, as_h2(q3.w), sL, sR, tminL, tminR, bound16);
            bound16 = bound_slack(bound16);
            float boundL = (float)bound16.x, boundR = (float)bound16.y;
            /* enter unless the order-independent part of the reference's verdict fails or the box certainly begins beyond
               everything that still matters (a subtree with a triangle too thin for its test to mean anything geometric
               comes with a bound of -inf or NaN: never skipped on distance) */
            const float reach = fmaxf(r.max_t, hitGate);
            bool pL = sL && !(boundL > reach), pR = sR && !(boundR > reach);
            if (MODE == MODE_MIXED && shadowLane) {       /* the reference's own culling; the stack keeps tmin where the bound would be */
                pL = sL && r.max_t > tminL; pR = sR && r.max_t > tminR;
                boundL = tminL; boundR = tminR;
            }
            const uint32_t refL = q3.x, refR = q3.y;
            const bool rightFirst = pL && pR && tminR < tminL;
            if (pL && pR) {
                const uint32_t second = rightFirst ? refL : refR;
                const float boundSecond = rightFirst ? boundL : boundR, tminSecond = rightFirst ? tminL : tminR;
                const uint16_t bb = hbits((h16)boundSecond);        /* exact: an fp16 value (the bound, or a shadow ray's tmin) */
                const uint16_t gb = hbits((h16)fmaxf(pathGate, tminSecond));             /* exact: both came from fp16 values */
                const uint32_t bg = (uint32_t)bb | ((uint32_t)gb << 16);
                if (sp < STK) { sStk[wave][sp][0][lane] = second; sStk[wave][sp][1][lane] = bg; }
                else a.overflow[PT_SPILL_AT(sp - STK)] = make_uint2(second, bg);
                sp++;
            }
            if (pL || pR) {
                const bool goRight = rightFirst || !pL;
                cur = goRight ? refR : refL;
                pathGate = fmaxf(pathGate, goRight ? tminR : tminL);
                needPop = false;
            }
        }

        while (needPop) {
            if (sp == 0) { done = true; break; }
            sp--;
            const int e = sp < STK ? sp : STK - 1;
            uint32_t ref = sStk[wave][e][0][lane], bg = sStk[wave][e][1][lane];
            asm volatile("" : "+v"(ref));
            if (sp >= STK) { uint2 o = a.overflow[PT_SPILL_AT(sp - STK)]; ref = o.x; bg = o.y; }
            const uint16_t bb = (uint16_t)bg, gb = (uint16_t)(bg >> 16);
            const float waiting = (float)hfrombits(bb);
            const bool enter = (MODE == MODE_MIXED && shadowLane) ? r.max_t > waiting : !(waiting > fmaxf(r.max_t, hitGate));
            if (enter) { cur = ref; pathGate = (float)hfrombits(gb); needPop = false; }
        }

        if (done) {
            /* Did the reference's walk run off the end of its array (the hit's leaf is then handed to the shade pass)?  It
               did if every node on the way from the root to the last node passed its box test when the walk came by; a node
               that fails there has skip link 0 and ends the walk.  With the certificate in hand (the ray is not listed
               below) those tests can be redone here: the nodes of that path in front of the hit's leaf X are X's own
               ancestors, which the reference entered; the ones behind it are reached with X found and nothing accepted
               since (X is the nearest, a tie is listed), so max_t is t of X there.  The table holds them deepest first: the
               last node's own box, the smallest, is the likeliest to fail. */
            uint32_t orig = 0u;
            bool unsettled = false;
            if (MODE != MODE_RAYS && !(MODE == MODE_MIXED && shadowLane) && nearLast && hit) {
                if (a.spineCount == 0u) unsettled = true;
                else {
                    bool ran = true;
                    for (uint32_t k = 0; k < a.spineCount && ran; k++) {
                        const PtSpineBox sb = a.spine[k];
                        if (sb.index <= hitLeaf) break;
                        float tm;
                        const bool s = box_static(r, sb.c[0], sb.c[1], sb.c[2], hfrombits(sb.h[0]), hfrombits(sb.h[1]), hfrombits(sb.h[2]),
                                                  hfrombits(sb.h[3]), hfrombits(sb.h[4]), hfrombits(sb.h[5]), tm);
                        ran = s && r.max_t > tm;
                    }
                    if (ran) orig = hitLeaf;
                }
            }
            const bool listed = !(MODE == MODE_MIXED && shadowLane) && hit && (tie || unsettled || !(otherT > hitGate) || !(hitGate < limit0));
            if (listed) flag_ray();
            store(hit, orig, listed);
            active = false;
        }
        }   /* if (active) */

        if (fresh) start_ray(w);
    }
    if (COUNT) {
        atomicAdd(a.stats + 0, nBox); atomicAdd(a.stats + 1, nTri); atomicAdd(a.stats + 2, nRec); atomicAdd(a.stats + 3, nRay);
    }
#undef limit0
}

/* Self-test of box_pair_bound, the lemma the certified walk rests on: for random fp32 boxes [a, b] (sizes and offsets over
 * twenty octaves, including half extents that are fp16 subnormals or beyond the fp16 range) digested exactly as the
 * flattening pass digests them, and random rays (starting inside, outside, far away; directions down to 1e-6 in a
 * component), the bound never exceeds the distance at which the ray enters [a, b], computed here in fp64.  Counts the
 * cases checked (the ray does reach the box) and the violations. */
__device__ __forceinline__ uint64_t selftest_rand(uint64_t& s)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_selftest_entry_bound(uint64_t seed, uint64_t n, unsigned long long* out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed + i * 0xD1B54A32D192ED03ull;
    auto uni = [&]() { return (float)(selftest_rand(s) >> 40) * (1.0f / 16777216.0f); };
    /* two boxes (the two halves of one record), one ray */
    float a[2][3], b[2][3], o[3], d[3];
    const float offScale = __builtin_exp2f(uni() * 20.0f - 8.0f);            /* where the boxes sit: 2^-8 .. 2^12 */
    for (int k = 0; k < 2; k++) {
        const float size = __builtin_exp2f(uni() * 30.0f - 20.0f);           /* half extents 2^-20 .. 2^10, times up to 2^7 below */
        for (int x = 0; x < 3; x++) {
            const float c = (uni() * 2.0f - 1.0f) * offScale, h = size * (uni() < 0.1f ? 128.0f * uni() : uni()) + 6e-5f;
            a[k][x] = c - h; b[k][x] = c + h * (0.5f + uni());
        }
    }
    const int pick = (int)(selftest_rand(s) & 1);
    const float where = uni();
    for (int x = 0; x < 3; x++) {
        const float mid = (a[pick][x] + b[pick][x]) * 0.5f, half = (b[pick][x] - a[pick][x]) * 0.5f;
        const float spread = where < 0.3f ? 0.9f : (where < 0.7f ? 4.0f : 300.0f);     /* inside, near, far */
        o[x] = mid + (uni() * 2.0f - 1.0f) * half * spread + (where >= 0.7f ? (uni() * 2.0f - 1.0f) * offScale : 0.0f);
    }
    {
        float z = 1.0f - 2.0f * uni(), phi = 6.2831853f * uni(), rr = __builtin_sqrtf(fmaxf(0.0f, 1.0f - z * z));
        d[0] = rr * cosf(phi); d[1] = rr * sinf(phi); d[2] = z;
        if (uni() < 0.2f) d[(int)(selftest_rand(s) % 3)] *= 1e-6f * (0.01f + uni());           /* nearly parallel to a slab */
        if (uni() < 0.3f) {        /* aimed at a point of the picked box, so that most cases count */
            float len = 0;
            for (int x = 0; x < 3; x++) { d[x] = a[pick][x] + (b[pick][x] - a[pick][x]) * uni() - o[x]; len += d[x] * d[x]; }
            len = __builtin_sqrtf(len);
            for (int x = 0; x < 3; x++) d[x] = len > 0 ? d[x] / len : (x == 0 ? 1.0f : 0.0f);
        }
    }
    if (d[0] == 0.0f || d[1] == 0.0f || d[2] == 0.0f) return;
    /* digest as scene_prep.cpp does, bound as k_trace_cert does */
    f32x2 c[3], dd[3]; h16x2 ha[3], hb[3];
    for (int x = 0; x < 3; x++) {
        float cc[2];
        for (int k = 0; k < 2; k++) cc[k] = (a[k][x] + b[k][x]) * 0.5f;
        c[x] = { cc[0], cc[1] };
        ha[x] = { (h16)(a[0][x] - cc[0]), (h16)(a[1][x] - cc[1]) };
        hb[x] = { (h16)(b[0][x] - cc[0]), (h16)(b[1][x] - cc[1]) };
        dd[x] = { cc[0] - o[x], cc[1] - o[x] };
    }
    float boundL, boundR;
    bool unreachable[2] = { false, false };      /* fp16 tmin of +inf: the reference's max_t > tmin never holds, nothing is claimed */
    /* the per-box constant as the flattening pass makes it: largest digested half extent + 2^-12, rounded up to fp16 */
    h16x2 wmax;
    {
        uint16_t wb[2];
        for (int k = 0; k < 2; k++) {
            double w = 0.0; bool inf = false;
            for (int x = 0; x < 3; x++) {
                const h16 lo = k == 0 ? ha[x].x : ha[x].y, hi = k == 0 ? hb[x].x : hb[x].y;
                const uint16_t ml = hbits(lo) & 0x7FFFu, mh = hbits(hi) & 0x7FFFu;
                inf = inf || ml >= 0x7C00u || mh >= 0x7C00u;
                w = fmax(w, fmax((double)(float)hfrombits(ml), (double)(float)hfrombits(mh)));
            }
            w += 0x1p-12;
            h16 rr = (h16)w;
            uint16_t u = hbits(rr);
            if ((double)(float)rr < w) u++;
            wb[k] = inf ? (uint16_t)0x7C00u : u;
        }
        wmax = { hfrombits(wb[0]), hfrombits(wb[1]) };
    }
    {
        RayRegs r; r.o = { o[0], o[1], o[2] }; ray_set_dir(r, { d[0], d[1], d[2] }); r.min_t = 0.0f; r.max_t = __builtin_huge_valf();
        const h16x2 rx = { r.rcpD.x, r.rcpD.x }, ry = { r.rcpD.y, r.rcpD.y }, rz = { r.rcpD.z, r.rcpD.z };
        bool sL, sR; float tL, tR; h16x2 b16;
        box_pair_bound(r, rx, ry, rz, bound_factor(r.rcpD.x), bound_factor(r.rcpD.y), bound_factor(r.rcpD.z), c[0], c[1], c[2],
                       ha[0], ha[1], ha[2], hb[0], hb[1], hb[2], wmax, sL, sR, tL, tR, b16);
        boundL = (float)b16.x; boundR = (float)b16.y;
        unreachable[0] = tL == __builtin_huge_valf(); unreachable[1] = tR == __builtin_huge_valf();
        (void)dd;
    }
    for (int k = 0; k < 2; k++) {
        double enter = -1e300, leave = 1e300;
        for (int x = 0; x < 3; x++) {
            const double t0 = ((double)a[k][x] - (double)o[x]) / (double)d[x], t1 = ((double)b[k][x] - (double)o[x]) / (double)d[x];
            enter = fmax(enter, fmin(t0, t1)); leave = fmin(leave, fmax(t0, t1));
        }
        if (!(enter <= leave) || unreachable[k]) continue;                /* the ray does not reach this box: nothing is claimed */
        atomicAdd(out + 0, 1ull);
        const float bound = k == 0 ? boundL : boundR;
        if ((double)bound > enter) atomicAdd(out + 1, 1ull);              /* (a NaN bound - every axis overflowed - skips nothing) */
    }
}
