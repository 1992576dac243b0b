/*
 * pt_oracle.cpp — TEST INFRASTRUCTURE ONLY.  Nothing under pathtracer_amd/ may link, load or call
 * this file; it is the checker for the HIP path, used by tests/, the smoke check and the
 * cpu_baseline leg of bench.py.
 *
 * A scalar CPU restatement of the hot path of JovianGraphics/PathTracer, one function per
 * reference function, each citing the reference file:line it follows.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or dumps, and neither its
 * shaders (no GLSL compiler / Vulkan ICD here) nor its host code (empty submodules: glm,
 * Ganymede, Himalia) can be built, so this restatement cannot be checked against outputs of
 * the reference itself.  It follows the reference sources line by line; where GLSL leaves the
 * arithmetic to the implementation, the choice made here is stated next to the code:
 *   - every fp16 operation rounds to nearest-even after each single operation (no fused
 *     multiply-add, no excess precision), subnormals kept;
 *   - min/max return the other operand when one operand is a NaN;
 *   - dot products and matrix products sum left to right, fp32 without contraction;
 *   - fp16 sqrt / inversesqrt / pow / sin / cos are the fp32 libm result rounded to fp16.
 *
 * Built with ROCm clang++ (gcc 11 has no _Float16) and -ffp-contract=off.
 */
#include "../include/ptcore.h"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#pragma STDC FP_CONTRACT OFF

#define PO_API extern "C" __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * fp16 arithmetic: explicit float round trips so that every operation rounds once, whatever
 * excess-precision mode the compiler is in.
 * ---------------------------------------------------------------------------------------- */
typedef _Float16 h16;
static inline h16   f2h(float x) { return (h16)x; }
static inline float h2f(h16 x) { return (float)x; }
static inline h16 hadd(h16 a, h16 b) { return (h16)((float)a + (float)b); }
static inline h16 hsub(h16 a, h16 b) { return (h16)((float)a - (float)b); }
static inline h16 hmul(h16 a, h16 b) { return (h16)((float)a * (float)b); }
static inline h16 hdiv(h16 a, h16 b) { return (h16)((float)a / (float)b); }
static inline h16 hneg(h16 a) { return (h16)(-(float)a); }
static inline bool hisnan(h16 a) { float f = (float)a; return f != f; }
static inline h16 hmin(h16 a, h16 b) { if (hisnan(a)) return b; if (hisnan(b)) return a; return ((float)b < (float)a) ? b : a; }
static inline h16 hmax(h16 a, h16 b) { if (hisnan(a)) return b; if (hisnan(b)) return a; return ((float)a < (float)b) ? b : a; }
static inline h16 habs(h16 a) { return (h16)fabsf((float)a); }
static inline h16 hsqrt(h16 a) { return (h16)sqrtf((float)a); }
static inline uint16_t hbits(h16 a) { uint16_t u; memcpy(&u, &a, 2); return u; }
static inline h16 hfrombits(uint16_t u) { h16 a; memcpy(&a, &u, 2); return a; }

struct H3 { h16 x, y, z; };
struct F3 { float x, y, z; };

static inline h16 hdot(H3 a, H3 b) { return hadd(hadd(hmul(a.x, b.x), hmul(a.y, b.y)), hmul(a.z, b.z)); }
static inline H3 hcross(H3 a, H3 b)
{
    return { hsub(hmul(a.y, b.z), hmul(b.y, a.z)), hsub(hmul(a.z, b.x), hmul(b.z, a.x)), hsub(hmul(a.x, b.y), hmul(b.x, a.y)) };
}
static inline H3 hnormalize(H3 v)
{
    h16 inv = f2h(1.0f / sqrtf(h2f(hdot(v, v))));
    return { hmul(v.x, inv), hmul(v.y, inv), hmul(v.z, inv) };
}
static inline H3 hscale(h16 s, H3 v) { return { hmul(s, v.x), hmul(s, v.y), hmul(s, v.z) }; }
static inline H3 hadd3(H3 a, H3 b) { return { hadd(a.x, b.x), hadd(a.y, b.y), hadd(a.z, b.z) }; }
static inline H3 hsub3(H3 a, H3 b) { return { hsub(a.x, b.x), hsub(a.y, b.y), hsub(a.z, b.z) }; }
static inline H3 f2h3(F3 v) { return { f2h(v.x), f2h(v.y), f2h(v.z) }; }
static inline F3 h2f3(H3 v) { return { h2f(v.x), h2f(v.y), h2f(v.z) }; }

static inline float fdot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline F3 fcross(F3 a, F3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
static inline F3 fsub(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline F3 fadd(F3 a, F3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }

/* ------------------------------------------------------------------------------------------
 * BuildBVH — BVH.cpp:5-53 (BBox), :65-81 (ComputeBBox), :83-309 (BuildBVH).
 * ---------------------------------------------------------------------------------------- */
namespace {

struct BBox {                       /* BVH.h:9-25 */
    F3 a{0, 0, 0}, b{0, 0, 0};
    bool empty = true;
    void Extend(F3 p)               /* BVH.cpp:21-34; glm::min(x,y) = y<x ? y : x, glm::max(x,y) = x<y ? y : x */
    {
        if (empty) { a = p; b = p; empty = false; return; }
        a = { p.x < a.x ? p.x : a.x, p.y < a.y ? p.y : a.y, p.z < a.z ? p.z : a.z };
        b = { b.x < p.x ? p.x : b.x, b.y < p.y ? p.y : b.y, b.z < p.z ? p.z : b.z };
    }
    void Extend(const BBox& o)      /* BVH.cpp:5-19 */
    {
        if (o.empty) return;
        if (empty) { a = o.a; b = o.b; empty = false; return; }
        a = { o.a.x < a.x ? o.a.x : a.x, o.a.y < a.y ? o.a.y : a.y, o.a.z < a.z ? o.a.z : a.z };
        b = { b.x < o.b.x ? o.b.x : b.x, b.y < o.b.y ? o.b.y : b.y, b.z < o.b.z ? o.b.z : b.z };
    }
    F3 GetSize() const              /* BVH.cpp:50-53 */
    {
        F3 d = fsub(b, a);
        return { 0.0f < d.x ? d.x : 0.0f, 0.0f < d.y ? d.y : 0.0f, 0.0f < d.z ? d.z : 0.0f };
    }
};

struct Task { uint32_t start, end, depth; bool isRight; int32_t writeTo; };   /* BVH.cpp:55-63 (bbox field unused) */

inline F3 vtx(const float* pos4, uint32_t i) { return { pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2] }; }

BBox ComputeBBox(const float* pos4, const uint32_t* idx, uint32_t start, uint32_t end)   /* BVH.cpp:65-81 */
{
    BBox bb;
    for (uint32_t i = start; i < end; i += 3) {
        bb.Extend(vtx(pos4, idx[i])); bb.Extend(vtx(pos4, idx[i + 1])); bb.Extend(vtx(pos4, idx[i + 2]));
    }
    return bb;
}

F3 glm_normalize(F3 v) { float s = 1.0f / sqrtf(fdot(v, v)); return { v.x * s, v.y * s, v.z * s }; }

}  // namespace

PO_API int po_build_bvh(const float* pos4, uint32_t* idx, uint32_t num_indices, PtBVHNode* nodes_out,
                        uint32_t nodes_cap, uint32_t* num_nodes_out, uint32_t* max_depth_out)
{
    std::vector<PtBVHNode> nodes;
    std::vector<Task> tasks;
    nodes.reserve(num_indices / 3 * 2);
    tasks.push_back({ 0, num_indices, 0, false, 0 });                    /* BVH.cpp:91 */
    uint32_t maxDepth = 0;

    while (!tasks.empty()) {                                             /* BVH.cpp:95 */
        Task task = tasks.back();
        tasks.pop_back();
        int32_t index = (int32_t)nodes.size();
        if (task.isRight) nodes[task.writeTo].right = index;             /* BVH.cpp:102-105 */
        nodes.push_back(PtBVHNode{});
        if (task.depth > maxDepth) maxDepth = task.depth;
        uint32_t start = task.start, end = task.end;

        BBox bbox = ComputeBBox(pos4, idx, start, end);                  /* BVH.cpp:115 */
        {
            PtBVHNode& node = nodes.back();
            node.a[0] = bbox.a.x - 0.00006f; node.a[1] = bbox.a.y - 0.00006f; node.a[2] = bbox.a.z - 0.00006f;   /* :117 */
            node.b[0] = bbox.b.x + 0.00006f; node.b[1] = bbox.b.y + 0.00006f; node.b[2] = bbox.b.z + 0.00006f;   /* :118 */
            if (end - start <= 3) {                                      /* BVH.cpp:120-130 */
                if (end - start != 3) return -1;
                node.right = -(int32_t)start;
                continue;
            }
        }

        F3 selectedAxis{0, 0, 0};                                        /* BVH.cpp:132-142 */
        const float splitRatio[9] = { -1e24f, 0.125f, 0.25f, 0.375f, 0.5f, 0.625f, 0.75f, 0.875f, 1e24f };
        F3 bboxSize = bbox.GetSize();
        float totalArea = fdot(bboxSize, bboxSize);
        float maxScore = -(1.0f + 2.0f * (float)(end - start) / 3.0f);
        float selectedRatio = 0.5f;
        F3 ab = fsub(bbox.a, bbox.b);
        float radius = sqrtf(fdot(ab, ab));                              /* glm::distance */
        F3 bboxCentroid = { (bbox.a.x + bbox.b.x) * 0.5f, (bbox.a.y + bbox.b.y) * 0.5f, (bbox.a.z + bbox.b.z) * 0.5f };

        const F3 availableAxes[9] = {                                    /* BVH.cpp:144-154 */
            {1, 0, 0}, {0, 1, 0}, {0, 0, 1},
            glm_normalize({-1, 0, 1}), glm_normalize({1, 0, 1}), glm_normalize({1, -1, 0}),
            glm_normalize({-1, -1, 0}), glm_normalize({0, 1, -1}), glm_normalize({0, -1, -1}),
        };

        for (uint32_t x = 0; x < 9; x++) {                               /* BVH.cpp:156-228 */
            F3 axis = availableAxes[x];
            BBox bboxBins[8];
            uint32_t numPrimBins[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t i = start; i < end; i += 3) {
                F3 p0 = vtx(pos4, idx[i]), p1 = vtx(pos4, idx[i + 1]), p2 = vtx(pos4, idx[i + 2]);
                F3 s = fadd(fadd(p0, p1), p2);
                F3 centroid = { s.x / 3.0f, s.y / 3.0f, s.z / 3.0f };
                for (uint32_t j = 0; j < 8; j++) {
                    /* float + float*double: evaluated in double, stored to float (BVH.cpp:174-175) */
                    float splitPos0 = (float)((double)fdot(bboxCentroid, axis) + (double)radius * ((double)splitRatio[j] - 0.5));
                    float splitPos1 = (float)((double)fdot(bboxCentroid, axis) + (double)radius * ((double)splitRatio[j + 1] - 0.5));
                    float l = fdot(centroid, axis);
                    if ((double)l > (double)splitPos0 - 0.00001 && (double)l < (double)splitPos1 + 0.00001) {   /* :179 */
                        bboxBins[j].Extend(p0); bboxBins[j].Extend(p1); bboxBins[j].Extend(p2);
                        numPrimBins[j]++;
                    }
                }
            }
            for (uint32_t i = 1; i < 8; i++) {                           /* BVH.cpp:189-227 */
                BBox left, right;
                uint32_t leftCount = 0, rightCount = 0;
                for (uint32_t j = 0; j < i; j++) { left.Extend(bboxBins[j]); leftCount += numPrimBins[j]; }
                for (uint32_t j = i; j < 8; j++) { right.Extend(bboxBins[j]); rightCount += numPrimBins[j]; }
                F3 leftSize = left.GetSize(), rightSize = right.GetSize();
                float pA = fdot(leftSize, leftSize) / totalArea;
                float pB = fdot(rightSize, rightSize) / totalArea;
                float score = -((1.0f + pA * (float)leftCount * 2.0f) + pB * (float)rightCount * 2.0f);
                if (score > maxScore) { maxScore = score; selectedRatio = splitRatio[i]; selectedAxis = axis; }
            }
        }

        float splitPos = (float)((double)fdot(bboxCentroid, selectedAxis) + (double)radius * ((double)selectedRatio - 0.5));   /* :230 */
        std::vector<uint32_t> leftPrims, rightPrims;
        for (uint32_t i = start; i < end; i += 3) {                      /* BVH.cpp:235-255 */
            F3 p0 = vtx(pos4, idx[i]), p1 = vtx(pos4, idx[i + 1]), p2 = vtx(pos4, idx[i + 2]);
            F3 s = fadd(fadd(p0, p1), p2);
            F3 centroid = { s.x / 3.0f, s.y / 3.0f, s.z / 3.0f };
            std::vector<uint32_t>& dst = (fdot(centroid, selectedAxis) > splitPos) ? rightPrims : leftPrims;
            dst.push_back(idx[i]); dst.push_back(idx[i + 1]); dst.push_back(idx[i + 2]);
        }
        uint32_t w = start;                                              /* BVH.cpp:257-267 */
        for (uint32_t v : leftPrims) idx[w++] = v;
        for (uint32_t v : rightPrims) idx[w++] = v;

        uint32_t center = start + ((end - start) / 3) / 2 * 3;           /* BVH.cpp:269-280 */
        if (leftPrims.empty() || rightPrims.empty()) {
            tasks.push_back({ center, end, task.depth + 1, true, index });
            tasks.push_back({ start, center, task.depth + 1, false, index });
        } else {
            uint32_t mid = start + (uint32_t)leftPrims.size();
            tasks.push_back({ mid, end, task.depth + 1, true, index });
            tasks.push_back({ start, mid, task.depth + 1, false, index });
        }
    }

    struct CN { int32_t index, next; };                                  /* BVH.cpp:284-304 */
    std::vector<CN> st = { {0, 0} };
    while (!st.empty()) {
        CN c = st.back(); st.pop_back();
        nodes[c.index].next = c.next;
        if (nodes[c.index].right > 0) {
            st.push_back({ c.index + 1, nodes[c.index].right });
            st.push_back({ nodes[c.index].right, c.next });
        }
    }
    if (num_nodes_out) *num_nodes_out = (uint32_t)nodes.size();
    if (max_depth_out) *max_depth_out = maxDepth;
    if (nodes.size() > nodes_cap) return -2;
    memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(PtBVHNode));
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Scene + traversal — structures.glsl:13-33, intersections.glsl:1-115.
 * ---------------------------------------------------------------------------------------- */
struct PoScene {
    std::vector<float> pos4;
    std::vector<PtVertexAux> aux;
    std::vector<uint32_t> idx;
    std::vector<PtBVHNode> nodes;
    std::vector<PtLight> lights;
    std::vector<uint16_t> blueNoise;
    std::vector<uint8_t> oddInterior;   /* analysis only, see PO_FLAG_SKIP_ODD_INTERIOR */
};

struct Ray {                        /* structures.glsl:13-21 */
    F3 o; float min_t; F3 d; float max_t; uint32_t origBvhId; H3 rcpD;
};
struct Intersection { H3 bary; int32_t i1, i2, i3; };   /* structures.glsl:29-33 */

struct TraceCounters { uint64_t boxTests = 0, triTests = 0; };

/* flags shared by the entry points below */
enum {
    PO_FLAG_REJECT_NAN = 1,         /* positive-form acceptance test: a NaN t/alpha/beta rejects (see intersect) */
    PO_FLAG_SKIP_ODD_INTERIOR = 2   /* ANALYSIS ONLY, not the reference's walk: interior nodes at odd depth pass their box
                                       test unconditionally, which is what a four-wide node that stores only its
                                       grandchildren's boxes would do.  Measures how often the reference's own (not
                                       conservative) fp16 test of such a node changes a result. */
};

static inline H3 rcp_dir(F3 d) { return { f2h(1.0f / d.x), f2h(1.0f / d.y), f2h(1.0f / d.z) }; }

/* intersections.glsl:1-30 */
static bool intersectBBox(const Ray& r, const float* _a, const float* _b)
{
    float cx = (_a[0] + _b[0]) * 0.5f, cy = (_a[1] + _b[1]) * 0.5f, cz = (_a[2] + _b[2]) * 0.5f;
    H3 a = { f2h(_a[0] - cx), f2h(_a[1] - cy), f2h(_a[2] - cz) };
    H3 b = { f2h(_b[0] - cx), f2h(_b[1] - cy), f2h(_b[2] - cz) };
    H3 o = { f2h(r.o.x - cx), f2h(r.o.y - cy), f2h(r.o.z - cz) };
    H3 rrd = r.rcpD;

    h16 t0 = hmul(hsub(a.x, o.x), rrd.x), t1 = hmul(hsub(b.x, o.x), rrd.x);
    h16 tmin = hmin(t0, t1), tmax = hmax(t0, t1);
    t0 = hmul(hsub(a.y, o.y), rrd.y); t1 = hmul(hsub(b.y, o.y), rrd.y);
    tmin = hmax(tmin, hmin(t0, t1)); tmax = hmin(tmax, hmax(t0, t1));
    t0 = hmul(hsub(a.z, o.z), rrd.z); t1 = hmul(hsub(b.z, o.z), rrd.z);
    tmin = hmax(tmin, hmin(t0, t1)); tmax = hmin(tmax, hmax(t0, t1));

    float ftmin = h2f(tmin), ftmax = h2f(tmax);
    return (ftmax > 0.0f || ftmin > 0.0f) && ftmax >= ftmin && r.min_t < ftmax && r.max_t > ftmin;
}

/* intersections.glsl:32-55 */
static bool intersect(Ray& r, F3 p1, F3 p2, F3 p3, int32_t i1, int32_t i2, int32_t i3, Intersection& isect, int flags)
{
    F3 e1 = fsub(p2, p1), e2 = fsub(p3, p1);
    F3 s = fsub(r.o, p1), s1 = fcross(r.d, e2), s2 = fcross(s, e1);
    float det = fdot(s1, e1);
    float t = fdot(s2, e2) / det, alpha = fdot(s1, s) / det, beta = fdot(s2, r.d) / det;
    float gamma = 1.0f - alpha - beta;
    if (flags & PO_FLAG_REJECT_NAN) {
        /* same as the reference for ordered values; a NaN (zero-area triangle) rejects instead of being accepted */
        if (!(t >= r.min_t && t <= r.max_t && alpha >= 0.0f && beta >= 0.0f && gamma >= 0.0f)) return false;
    } else {
        if (t < r.min_t || t > r.max_t || alpha < 0.0f || beta < 0.0f || gamma < 0.0f) return false;
    }
    r.max_t = t;
    isect.bary = { f2h(gamma), f2h(alpha), f2h(beta) };
    isect.i1 = i1; isect.i2 = i2; isect.i3 = i3;
    return true;
}

/* intersections.glsl:57-115 for a subgroup of n lanes in lockstep.  n = 1 is the plain build.
 * With speculative = true every lane follows the vote of intersections.glsl:73-77: all lanes
 * start at index 0 and take the same branch, so they share one index.  hit[] receives the
 * per-lane return value. */
static void traceRayGroup(const PoScene& sc, Ray* rays, Intersection* isects, char* hit, int n, bool stopIfHit,
                          bool speculative, int flags, TraceCounters* cnt)
{
    const uint32_t numBVHNodes = (uint32_t)sc.nodes.size();
    std::vector<uint32_t> index(n, 0), hitIndex(n, 0);
    std::vector<char> active(n, 1);
    for (int l = 0; l < n; l++) hit[l] = 0;
    int remaining = n;

    while (remaining > 0) {
        /* loop condition + box test per active lane */
        std::vector<char> pass(n, 0);
        bool any = false;
        for (int l = 0; l < n; l++) {
            if (!active[l]) continue;
            if (!(index[l] < numBVHNodes)) {                 /* ran off the end: intersections.glsl:113-114 */
                rays[l].origBvhId = hitIndex[l];
                active[l] = 0; remaining--;
                continue;
            }
            const PtBVHNode& nd = sc.nodes[index[l]];
            pass[l] = intersectBBox(rays[l], nd.a, nd.b);
            if (cnt) cnt->boxTests++;
            any |= (pass[l] != 0);
        }
        for (int l = 0; l < n; l++) {
            if (!active[l]) continue;
            bool take = speculative ? any : (pass[l] != 0);
            const PtBVHNode& nd = sc.nodes[index[l]];
            if (take) {
                int32_t right = nd.right;
                if (right <= 0 && (rays[l].origBvhId == 0 || rays[l].origBvhId != index[l])) {
                    uint32_t tri = (uint32_t)(-right) / 3;   /* texelFetch(indicies, -right / 3) */
                    uint32_t i1 = sc.idx[3 * tri], i2 = sc.idx[3 * tri + 1], i3 = sc.idx[3 * tri + 2];
                    if (cnt) cnt->triTests++;
                    if (intersect(rays[l], vtx(sc.pos4.data(), i1), vtx(sc.pos4.data(), i2), vtx(sc.pos4.data(), i3),
                                  (int32_t)i1, (int32_t)i2, (int32_t)i3, isects[l], flags)) {
                        hit[l] = 1;
                        if (stopIfHit) { active[l] = 0; remaining--; continue; }
                        hitIndex[l] = index[l];
                    }
                }
                index[l]++;
            } else {
                index[l] = (uint32_t)nd.next;
                if (index[l] == 0) { active[l] = 0; remaining--; }    /* return hit, origBvhId untouched */
            }
        }
    }
}

static bool traceRay(const PoScene& sc, Ray& r, Intersection& isect, bool stopIfHit, int flags, TraceCounters* cnt)
{
    char hit;
    traceRayGroup(sc, &r, &isect, &hit, 1, stopIfHit, false, flags, cnt);
    return hit != 0;
}

/* A faster single-ray walk of the same loop, used for the multi-threaded CPU baseline and the
 * large parity runs.  Identical results to traceRayGroup with n = 1 (checked by tests). */
static bool traceRayFast(const PoScene& sc, Ray& r, Intersection& isect, bool stopIfHit, int flags, TraceCounters* cnt)
{
    const PtBVHNode* nodes = sc.nodes.data();
    const uint32_t numBVHNodes = (uint32_t)sc.nodes.size();
    const float* pos4 = sc.pos4.data();
    const uint32_t* idx = sc.idx.data();
    bool hit = false;
    uint32_t index = 0, hitIndex = 0;
    uint64_t nb = 0, nt = 0;
    const uint8_t* skipOdd = ((flags & PO_FLAG_SKIP_ODD_INTERIOR) && !sc.oddInterior.empty()) ? sc.oddInterior.data() : nullptr;
    while (index < numBVHNodes) {
        const PtBVHNode& nd = nodes[index];
        nb++;
        if ((skipOdd && skipOdd[index]) || intersectBBox(r, nd.a, nd.b)) {
            int32_t right = nd.right;
            if (right <= 0 && (r.origBvhId == 0 || r.origBvhId != index)) {
                uint32_t tri = (uint32_t)(-right) / 3;
                uint32_t i1 = idx[3 * tri], i2 = idx[3 * tri + 1], i3 = idx[3 * tri + 2];
                nt++;
                if (intersect(r, vtx(pos4, i1), vtx(pos4, i2), vtx(pos4, i3), (int32_t)i1, (int32_t)i2, (int32_t)i3, isect, flags)) {
                    hit = true;
                    if (stopIfHit) { if (cnt) { cnt->boxTests += nb; cnt->triTests += nt; } return true; }
                    hitIndex = index;
                }
            }
            index++;
        } else {
            index = (uint32_t)nd.next;
            if (index == 0) { if (cnt) { cnt->boxTests += nb; cnt->triTests += nt; } return hit; }
        }
    }
    r.origBvhId = hitIndex;
    if (cnt) { cnt->boxTests += nb; cnt->triTests += nt; }
    return hit;
}

PO_API PoScene* po_scene_create(const float* pos4, uint32_t V, const PtVertexAux* aux, const uint32_t* idx, uint32_t T,
                                const PtBVHNode* nodes, uint32_t N, const PtLight* lights, uint32_t L, const uint16_t* blueNoise)
{
    PoScene* s = new PoScene;
    s->pos4.assign(pos4, pos4 + 4 * (size_t)V);
    if (aux) s->aux.assign(aux, aux + V);
    s->idx.assign(idx, idx + 3 * (size_t)T);
    s->nodes.assign(nodes, nodes + N);
    if (lights) s->lights.assign(lights, lights + L);
    if (blueNoise) s->blueNoise.assign(blueNoise, blueNoise + 65536);
    /* depth parity of interior nodes (pre-order array: left child = i + 1, right child = nodes[i].right) */
    s->oddInterior.assign(N, 0);
    std::vector<std::pair<uint32_t, uint32_t>> st; st.push_back({ 0u, 0u });
    while (!st.empty()) {
        auto [i, d] = st.back(); st.pop_back();
        if (i >= N || nodes[i].right <= 0) continue;
        s->oddInterior[i] = (uint8_t)(d & 1u);
        st.push_back({ (uint32_t)nodes[i].right, d + 1 }); st.push_back({ i + 1, d + 1 });
    }
    return s;
}
PO_API void po_scene_destroy(PoScene* s) { delete s; }

/* Traces rays[0..n): closest hit (any_hit = 0) or first accepted hit in walk order (any_hit = 1).
 * hits: t + vertex triple (miss: -1,-1,-1 and t = max_t); bary (optional): 3 fp16 bit patterns per ray;
 * ran_off (optional): the value traceRay leaves in origBvhId (intersections.glsl:113);
 * counters (optional): {box tests, triangle tests} summed over all rays. */
PO_API void po_trace_rays(const PoScene* sc, const PtRay* rays, uint64_t n, int any_hit, int flags, PtHit* hits,
                          uint16_t* bary, uint32_t* ran_off, uint64_t* counters, int num_threads)
{
    if (num_threads <= 0) num_threads = (int)std::thread::hardware_concurrency();
    if (num_threads < 1) num_threads = 1;
    if ((uint64_t)num_threads > n) num_threads = n ? (int)n : 1;
    std::vector<TraceCounters> tc(num_threads);
    auto work = [&](int tid) {
        uint64_t lo = n * tid / num_threads, hi = n * (tid + 1) / num_threads;
        for (uint64_t i = lo; i < hi; i++) {
            Ray r;
            r.o = { rays[i].o[0], rays[i].o[1], rays[i].o[2] };
            r.d = { rays[i].d[0], rays[i].d[1], rays[i].d[2] };
            r.min_t = rays[i].min_t; r.max_t = rays[i].max_t;
            r.origBvhId = 0;
            r.rcpD = rcp_dir(r.d);
            Intersection is{};
            bool h = traceRayFast(*sc, r, is, any_hit != 0, flags, &tc[tid]);
            hits[i].t = r.max_t;
            if (h) { hits[i].i1 = is.i1; hits[i].i2 = is.i2; hits[i].i3 = is.i3; }
            else { hits[i].i1 = hits[i].i2 = hits[i].i3 = -1; }
            if (bary) {
                bary[3 * i] = h ? hbits(is.bary.x) : 0; bary[3 * i + 1] = h ? hbits(is.bary.y) : 0; bary[3 * i + 2] = h ? hbits(is.bary.z) : 0;
            }
            if (ran_off) ran_off[i] = r.origBvhId;
        }
    };
    if (num_threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < num_threads; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (counters) {
        counters[0] = counters[1] = 0;
        for (auto& c : tc) { counters[0] += c.boxTests; counters[1] += c.triTests; }
    }
}

/* Same rays through the lockstep group walk (group = consecutive `group` rays), for checking
 * traceRayFast against traceRayGroup and for the speculative build. */
PO_API void po_trace_rays_group(const PoScene* sc, const PtRay* rays, uint64_t n, int any_hit, int flags, int group,
                                int speculative, PtHit* hits)
{
    std::vector<Ray> r(group);
    std::vector<Intersection> is(group);
    std::vector<char> hv(group);
    for (uint64_t base = 0; base < n; base += group) {
        int m = (int)((n - base < (uint64_t)group) ? n - base : group);
        for (int l = 0; l < m; l++) {
            const PtRay& q = rays[base + l];
            r[l].o = { q.o[0], q.o[1], q.o[2] }; r[l].d = { q.d[0], q.d[1], q.d[2] };
            r[l].min_t = q.min_t; r[l].max_t = q.max_t; r[l].origBvhId = 0; r[l].rcpD = rcp_dir(r[l].d);
            is[l] = Intersection{};
        }
        traceRayGroup(*sc, r.data(), is.data(), hv.data(), m, any_hit != 0, speculative != 0, flags, nullptr);
        for (int l = 0; l < m; l++) {
            PtHit& h = hits[base + l];
            h.t = r[l].max_t;
            if (hv[l]) { h.i1 = is[l].i1; h.i2 = is[l].i2; h.i3 = is[l].i3; } else { h.i1 = h.i2 = h.i3 = -1; }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * RNG and sampling — noise.glsl:53-77, 95-109.
 * ---------------------------------------------------------------------------------------- */
struct Rng {
    uint32_t rnd = 0;
    uint32_t getRand() { rnd = rnd * 1103515245u + 12345u; return rnd & 0x3FFFFFFFu; }     /* noise.glsl:55-59 */
    float getRandF() { return (float)getRand() / (float)0x3FFFFFFF; }                      /* noise.glsl:61-64 */
};

static void WeylNth(uint32_t n, h16 out[2])                                                /* noise.glsl:66-69 */
{
    float a = (float)(n * 12664745u) / 16777216.0f, b = (float)(n * 9560333u) / 16777216.0f;
    a = a - floorf(a); b = b - floorf(b);
    out[0] = f2h(a); out[1] = f2h(b);
}

static H3 cosineHemisphere(const h16 i[2])                                                 /* noise.glsl:71-77 */
{
    h16 theta = hmul(hmul(f2h(2.0f), f2h(3.1415926f)), i[1]);
    h16 sqrtPhi = hsqrt(i[0]);
    h16 c = f2h(cosf(h2f(theta))), s = f2h(sinf(h2f(theta)));
    return { hmul(c, sqrtPhi), hmul(s, sqrtPhi), hsqrt(hsub(f2h(1.0f), i[0])) };
}

static H3 to_coord_space(H3 n, H3 r)                                                       /* noise.glsl:95-109 */
{
    H3 h = n;
    float ax = fabsf(h2f(h.x)), ay = fabsf(h2f(h.y)), az = fabsf(h2f(h.z));
    if (ax <= ay && ax <= az) h.x = f2h(1.0f);
    else if (ay <= ax && ay <= az) h.y = f2h(1.0f);
    else h.z = f2h(1.0f);
    H3 y = hnormalize(hcross(h, n));
    H3 x = hnormalize(hcross(n, y));
    return hadd3(hadd3(hscale(r.x, x), hscale(r.y, y)), hscale(r.z, n));
}

/* ------------------------------------------------------------------------------------------
 * Frame state and the four kernels — launch.comp:28-51, trace.comp:58-258, raysort.comp:23-55,
 * composite.frag:30-55.
 * ---------------------------------------------------------------------------------------- */
struct PoFrame {
    uint32_t W = 0, H = 0, slots = 0;
    std::vector<PtRayStackSlot> stack;
    std::vector<uint32_t> jobs;
    std::vector<float> accumulation;     /* RGBA32F */
    std::vector<uint16_t> current;       /* RGBA16F */
    std::atomic<uint64_t> closestRays{0}, shadowRays{0};     /* atomic: po_render_frame_mt walks tile rows on several threads */
    int traceThreads = 1;                                    /* set by po_render_frame_mt for the duration of a frame */
    TraceCounters counters;
};

enum {                                   /* behaviour switches of the frame functions */
    PO_MODE_ZERO_NEXT_SLOT = 1,          /* zero prob of slot d+1 when a path ends at d (default build behaviour);
                                            off = leave deeper slots stale as the reference does */
    PO_MODE_RESEED_EVERY_FRAME = 2,      /* launch seeds randState every frame instead of only after a clear */
    PO_MODE_KILL_ON_MISS = 4,            /* a missed ray's job is killed instead of re-traced each bounce (same image) */
    PO_MODE_PLAIN_SHADOW_WALK = 8,       /* ANALYSIS ONLY, not the reference: in the SPECULATIVE build the shadow walk does not
                                            vote.  Lets a test show that the vote in the shadow walk changes results */
};

PO_API PoFrame* po_frame_create(uint32_t W, uint32_t H, uint32_t slots)
{
    PoFrame* f = new PoFrame;
    f->W = W; f->H = H; f->slots = slots;
    f->stack.assign((size_t)W * H * slots, PtRayStackSlot{});
    f->jobs.assign((size_t)W * H, PT_JOB_DEAD);
    f->accumulation.assign((size_t)W * H * 4, 0.0f);
    f->current.assign((size_t)W * H * 4, 0);
    return f;
}
PO_API void po_frame_destroy(PoFrame* f) { delete f; }
PO_API PtRayStackSlot* po_frame_stack(PoFrame* f) { return f->stack.data(); }
PO_API uint32_t* po_frame_jobs(PoFrame* f) { return f->jobs.data(); }
PO_API float* po_frame_accumulation(PoFrame* f) { return f->accumulation.data(); }
PO_API uint16_t* po_frame_current(PoFrame* f) { return f->current.data(); }
PO_API void po_frame_clear_accumulation(PoFrame* f) { std::fill(f->accumulation.begin(), f->accumulation.end(), 0.0f); }
PO_API void po_frame_ray_counts(PoFrame* f, uint64_t out[4])
{
    out[0] = f->closestRays; out[1] = f->shadowRays; out[2] = f->counters.boxTests; out[3] = f->counters.triTests;
}

static inline void mat_mul_vec(const float* m, const float v[4], float out[4])   /* column-major mat4 * vec4 */
{
    for (int i = 0; i < 4; i++) out[i] = ((m[0 + i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2]) + m[12 + i] * v[3];
}

/* launch.comp:28-51 */
PO_API void po_launch(const PoScene* sc, PoFrame* f, const PtShaderConstants* c, int mode)
{
    const uint32_t W = (uint32_t)c->viewportSize[0], H = (uint32_t)c->viewportSize[1];
    for (uint32_t gy = 0; gy < H; gy++)
        for (uint32_t gx = 0; gx < W; gx++) {
            uint32_t lx = gx + (uint32_t)c->viewportBase[0], ly = gy + (uint32_t)c->viewportBase[1];
            if ((float)lx >= c->viewportSize[0] || (float)ly >= c->viewportSize[1]) continue;
            uint32_t jitter = (uint32_t)sc->blueNoise[(lx & 0xFF) + ((ly & 0xFF) << 8)] * 256u + (uint32_t)sc->blueNoise[c->frameIndex & 0xFFFF];
            h16 w[2]; WeylNth(jitter, w);
            float px = (((float)lx + h2f(w[0])) / c->viewportSize[0]) * 2.0f - 1.0f;
            float py = (((float)ly + h2f(w[1])) / c->viewportSize[1]) * 2.0f - 1.0f;
            float projPos[4] = { px, py, 1.0f, 1.0f }, viewPos[4], world[4], cam[4];
            mat_mul_vec(c->projInvMtx, projPos, viewPos);
            float vw = viewPos[3];
            for (int k = 0; k < 4; k++) viewPos[k] /= vw;
            mat_mul_vec(c->viewInvMtx, viewPos, world);
            const float origin[4] = { 0, 0, 0, 1 };
            mat_mul_vec(c->viewInvMtx, origin, cam);
            float cw = cam[3];
            for (int k = 0; k < 4; k++) cam[k] /= cw;

            uint32_t jobIndex = ly * W + lx, stackIndex = jobIndex * c->numRays;
            PtRayStackSlot& s = f->stack[stackIndex];
            s.prob = 0;
            s.rayOrigin[0] = cam[0]; s.rayOrigin[1] = cam[1]; s.rayOrigin[2] = cam[2];
            F3 d = { world[0] - cam[0], world[1] - cam[1], world[2] - cam[2] };
            float inv = 1.0f / sqrtf(fdot(d, d));                        /* normalize = v * inversesqrt(dot(v,v)) */
            s.rayDirection[0] = d.x * inv; s.rayDirection[1] = d.y * inv; s.rayDirection[2] = d.z * inv;
            s.currentDepth = 0;
            f->jobs[jobIndex] = stackIndex;
            if ((mode & PO_MODE_RESEED_EVERY_FRAME) || f->accumulation[4 * (size_t)jobIndex + 3] < 0.5f) s.randState = jitter;
        }
}

/* What shadeHit holds when it reaches its shadow walk (trace.comp:123). */
struct ShadeState {
    h16 albedo[4]; H3 normal, rd16, lightRadiance; h16 falloff; F3 lightDir, hitPos; Ray rLight;
};

/* trace.comp:58-121: everything before the shadow walk. */
static void shadeHitBefore(const PoScene& sc, const PtShaderConstants& c, Rng& rng, F3 hitPos, const Intersection& isect, const Ray& r, ShadeState& st)
{
    h16* albedo = st.albedo;
    const PtVertexAux& v1 = sc.aux[isect.i1]; const PtVertexAux& v2 = sc.aux[isect.i2]; const PtVertexAux& v3 = sc.aux[isect.i3];
    const h16 inv255 = hdiv(f2h(1.0f), f2h(255.0f));
    h16 bx = isect.bary.x, by = isect.bary.y, bz = isect.bary.z;
    for (int k = 0; k < 4; k++) {
        h16 c1 = hmul(f2h((float)v1.color[k]), inv255), c2 = hmul(f2h((float)v2.color[k]), inv255), c3 = hmul(f2h((float)v3.color[k]), inv255);
        albedo[k] = hadd(hadd(hmul(bx, c1), hmul(by, c2)), hmul(bz, c3));                   /* trace.comp:71 */
    }
    H3 n1 = { f2h(v1.normal[0]), f2h(v1.normal[1]), f2h(v1.normal[2]) };
    H3 n2 = { f2h(v2.normal[0]), f2h(v2.normal[1]), f2h(v2.normal[2]) };
    H3 n3 = { f2h(v3.normal[0]), f2h(v3.normal[1]), f2h(v3.normal[2]) };
    H3 normal = hadd3(hadd3(hscale(bx, n1), hscale(by, n2)), hscale(bz, n3));               /* trace.comp:70 */
    const h16 g = f2h(2.2f);
    for (int k = 0; k < 3; k++) albedo[k] = f2h(powf(h2f(albedo[k]), h2f(g)));              /* trace.comp:72 */

    H3 rd16 = f2h3(r.d);
    if (h2f(hdot(normal, rd16)) > 0.0f) normal = { hneg(normal.x), hneg(normal.y), hneg(normal.z) };   /* trace.comp:74 */

    float coinFlip = rng.getRandF();                                                        /* trace.comp:77 */
    h16 falloff;
    F3 lightDir;
    H3 lightRadiance;
    Ray rLight;
    if (coinFlip < 0.5f) {                                                                  /* trace.comp:85-105 */
        const PtLight& L = sc.lights[0];
        F3 posDiff = fsub({ L.pos[0], L.pos[1], L.pos[2] }, hitPos);
        float dist = sqrtf(fdot(posDiff, posDiff));
        lightDir = { posDiff.x / dist, posDiff.y / dist, posDiff.z / dist };
        rLight.max_t = dist - 0.00005f;
        lightRadiance = { f2h(L.radiance[0]), f2h(L.radiance[1]), f2h(L.radiance[2]) };
        falloff = hmul(f2h(1.0f / (dist * dist + 1.0f)), hmax(f2h(0.0f), hdot(normal, f2h3(lightDir))));
    } else {                                                                                /* trace.comp:106-121 */
        h16 gs[2]; WeylNth(rng.getRand(), gs);
        lightDir = h2f3(to_coord_space(normal, cosineHemisphere(gs)));
        rLight.max_t = 1000.0f;
        lightRadiance = { f2h(c.ambientRadiance[0]), f2h(c.ambientRadiance[1]), f2h(c.ambientRadiance[2]) };
        falloff = f2h(1.0f);
    }
    rLight.o = hitPos; rLight.d = lightDir; rLight.origBvhId = r.origBvhId;
    rLight.rcpD = rcp_dir(lightDir); rLight.min_t = 0.001f;
    st.normal = normal; st.rd16 = rd16; st.lightRadiance = lightRadiance; st.falloff = falloff; st.lightDir = lightDir; st.hitPos = hitPos;
    st.rLight = rLight;
}

/* trace.comp:123-163: from the shadow walk's verdict on.  Returns false when the path ends here. */
static bool shadeHitAfter(const PtShaderConstants& c, Rng& rng, const ShadeState& st, bool occluded, Ray& r, h16 albedo[4], h16 wIn[3], h16& prob)
{
    (void)c;
    wIn[0] = wIn[1] = wIn[2] = f2h(0.0f);
    for (int k = 0; k < 4; k++) albedo[k] = st.albedo[k];
    const H3 normal = st.normal, rd16 = st.rd16, lightRadiance = st.lightRadiance;
    const h16 falloff = st.falloff;
    const F3 lightDir = st.lightDir, hitPos = st.hitPos;
    if (!occluded) {                                                                        /* trace.comp:123-130 */
        h16 lr[3] = { lightRadiance.x, lightRadiance.y, lightRadiance.z };
        if (h2f(albedo[3]) > 0.5f) {
            for (int k = 0; k < 3; k++) wIn[k] = hadd(wIn[k], hmul(hmul(hmul(f2h(2.0f), falloff), albedo[k]), lr[k]));
        } else {
            h16 ind = f2h(fdot(lightDir, r.d) > 0.995f ? 1.0f : 0.0f);
            for (int k = 0; k < 3; k++) wIn[k] = hadd(wIn[k], hmul(hmul(f2h(2.0f), ind), lr[k]));
        }
    }

    h16 gs[2]; WeylNth(rng.getRand(), gs);                                                  /* trace.comp:133-134 */
    F3 nextDir = h2f3(to_coord_space(normal, cosineHemisphere(gs)));

    if (h2f(albedo[3]) < 0.5f) {                                                            /* trace.comp:136-149 */
        /* isLastHitDelta is only ever read for depth 0 and only used for depth > 0
           (trace.comp:139, 202-205), so the index of refraction is always 1.3 */
        h16 eta = f2h(1.3f);
        H3 I = rd16, N = normal;
        h16 d = hdot(N, I);
        h16 k = hsub(f2h(1.0f), hmul(hmul(eta, eta), hsub(f2h(1.0f), hmul(d, d))));         /* GLSL refract */
        H3 refr;
        if (h2f(k) < 0.0f) refr = { f2h(0.0f), f2h(0.0f), f2h(0.0f) };
        else refr = hsub3(hscale(eta, I), hscale(hadd(hmul(eta, d), hsqrt(k)), N));
        nextDir = h2f3(refr);
        if (nextDir.x == 0.0f && nextDir.y == 0.0f && nextDir.z == 0.0f)
            nextDir = h2f3(hsub3(I, hscale(hmul(f2h(2.0f), d), N)));                        /* GLSL reflect */
        for (int q = 0; q < 4; q++) albedo[q] = hsqrt(albedo[q]);
        prob = f2h(1.0f);
    } else {                                                                                /* trace.comp:150-154 */
        prob = hdiv(f2h(1.0f), f2h(0.7f));
        if (rng.getRandF() > 0.7f) return false;
    }
    r.o = hitPos; r.d = nextDir;                                                            /* trace.comp:156-160 */
    return true;
}

struct TraceJobState { uint32_t jobIndex, stackIndex, currentDepth; Rng rng; Ray r; Intersection isect; bool hit; bool live; ShadeState shade; };

/* trace.comp:192-212 up to the shadow walk inside shadeHit.  Returns false for a missed ray (nothing more to do). */
static bool finishJobBefore(const PoScene& sc, PoFrame& f, const PtShaderConstants& c, TraceJobState& j, int mode)
{
    PtRayStackSlot* stack = f.stack.data();
    if (!j.hit) {                                                                           /* trace.comp:192-196 */
        stack[j.stackIndex].prob = 0;
        if (mode & PO_MODE_KILL_ON_MISS) f.jobs[j.jobIndex] = PT_JOB_DEAD;
        return false;
    }
    F3 hitPos = { j.r.max_t * j.r.d.x + j.r.o.x, j.r.max_t * j.r.d.y + j.r.o.y, j.r.max_t * j.r.d.z + j.r.o.z };   /* :198 */
    shadeHitBefore(sc, c, j.rng, hitPos, j.isect, j.r, j.shade);
    f.shadowRays++;
    return true;
}

/* the rest of shadeHit and trace.comp:212-233, given the shadow walk's verdict */
static void finishJobAfter(PoFrame& f, const PtShaderConstants& c, TraceJobState& j, bool occluded, int mode)
{
    PtRayStackSlot* stack = f.stack.data();
    h16 albedo[4], wIn[3], prob = f2h(1.0f);
    bool cont = shadeHitAfter(c, j.rng, j.shade, occluded, j.r, albedo, wIn, prob);
    if (cont && j.currentDepth != c.numRays - 1) {                                          /* trace.comp:212-223 */
        PtRayStackSlot& nx = stack[j.stackIndex + 1];
        nx.rayOrigin[0] = j.r.o.x; nx.rayOrigin[1] = j.r.o.y; nx.rayOrigin[2] = j.r.o.z;
        nx.rayDirection[0] = j.r.d.x; nx.rayDirection[1] = j.r.d.y; nx.rayDirection[2] = j.r.d.z;
        nx.randState = j.rng.rnd;
        nx.currentDepth = j.currentDepth + 1;
        for (int k = 0; k < 4; k++) nx.hitAlbedo[k] = 0;
        f.jobs[j.jobIndex] = j.stackIndex + 1;
    } else {                                                                                /* trace.comp:224-227 */
        f.jobs[j.jobIndex] = PT_JOB_DEAD;
        if ((mode & PO_MODE_ZERO_NEXT_SLOT) && j.currentDepth != c.numRays - 1) stack[j.stackIndex + 1].prob = 0;
    }
    PtRayStackSlot& s = stack[j.stackIndex];                                                /* trace.comp:229-232 */
    s.randState = j.rng.rnd;
    s.prob = hbits(prob);
    for (int k = 0; k < 4; k++) s.hitAlbedo[k] = hbits(albedo[k]);
    for (int k = 0; k < 3; k++) s.wIn[k] = hbits(wIn[k]);
}

static void beginJob(const PoFrame& f, uint32_t jobIndex, TraceJobState& j)                 /* trace.comp:165-190 */
{
    j.jobIndex = jobIndex;
    j.stackIndex = f.jobs[jobIndex];
    j.live = j.stackIndex != PT_JOB_DEAD;
    if (!j.live) return;
    const PtRayStackSlot& s = f.stack[j.stackIndex];
    j.currentDepth = s.currentDepth;
    j.rng.rnd = s.randState;
    j.r.o = { s.rayOrigin[0], s.rayOrigin[1], s.rayOrigin[2] };
    j.r.d = { s.rayDirection[0], s.rayDirection[1], s.rayDirection[2] };
    j.r.origBvhId = 0;
    j.r.rcpD = rcp_dir(j.r.d);
    j.r.min_t = 0.001f; j.r.max_t = 100000.0f;
    j.isect = Intersection{};
    j.hit = false;
}

/* trace.comp:237-258 over the whole job grid.  subgroup = 0: plain build.  subgroup > 0: the
 * SPECULATIVE build with subgroups of that many lanes; lane k of an 8x8 workgroup handles tile
 * position (k % 8, k / 8) and a subgroup is `subgroup` consecutive lanes. */
static void trace_tile_rows(const PoScene* sc, PoFrame* f, const PtShaderConstants* c, int subgroup, int mode, int flags,
                            uint32_t tyBegin, uint32_t tyEnd, TraceCounters* counters)
{
    const uint32_t W = (uint32_t)c->viewportSize[0], H = (uint32_t)c->viewportSize[1];
    std::vector<TraceJobState> js(64);
    for (uint32_t ty = tyBegin; ty < tyEnd; ty++)
        for (uint32_t tx = 0; tx < (W + 7) / 8; tx++) {
            int nj = 0;
            for (uint32_t k = 0; k < 64; k++) {
                uint32_t x = tx * 8 + (k % 8), y = ty * 8 + (k / 8);
                js[k].live = false;
                if (!(y < H) || !(x < W)) continue;       /* the reference checks y only (trace.comp:256): widths here are multiples of 8 */
                beginJob(*f, y * W + x, js[k]);
                nj++;
            }
            (void)nj;
            if (subgroup <= 0) {
                for (uint32_t k = 0; k < 64; k++) {
                    if (!js[k].live) continue;
                    f->closestRays++;
                    js[k].hit = traceRayFast(*sc, js[k].r, js[k].isect, false, flags, counters);
                    if (finishJobBefore(*sc, *f, *c, js[k], mode)) {
                        Intersection isectDL{};
                        Ray rl = js[k].shade.rLight;
                        bool occluded = traceRayFast(*sc, rl, isectDL, true, flags, counters);            /* trace.comp:123 */
                        finishJobAfter(*f, *c, js[k], occluded, mode);
                    }
                }
            } else {
                for (uint32_t base = 0; base < 64; base += subgroup) {
                    std::vector<Ray> rr; std::vector<Intersection> ii; std::vector<int> who;
                    for (uint32_t k = base; k < base + (uint32_t)subgroup && k < 64; k++)
                        if (js[k].live) { rr.push_back(js[k].r); ii.push_back(js[k].isect); who.push_back((int)k); }
                    if (rr.empty()) continue;
                    std::vector<char> hv(rr.size());
                    f->closestRays += rr.size();
                    traceRayGroup(*sc, rr.data(), ii.data(), hv.data(), (int)rr.size(), false, true, flags, counters);
                    for (size_t q = 0; q < rr.size(); q++) { js[who[q]].r = rr[q]; js[who[q]].isect = ii[q]; js[who[q]].hit = hv[q] != 0; }
                    /* The whole of trace.comp is compiled with SPECULATIVE (CMakeLists.txt:33-36), so the vote of
                       intersections.glsl:73-77 also governs the any-hit walk shadeHit makes (trace.comp:123).  Its
                       subgroup is what is left of this one: the lanes whose ray hit (the others returned at
                       trace.comp:194-195); a lane that finds an occluder returns and votes no more. */
                    std::vector<Ray> sr; std::vector<Intersection> si; std::vector<int> sw;
                    for (size_t q = 0; q < rr.size(); q++)
                        if (finishJobBefore(*sc, *f, *c, js[who[q]], mode)) { sr.push_back(js[who[q]].shade.rLight); si.push_back(Intersection{}); sw.push_back(who[q]); }
                    if (!sr.empty()) {
                        std::vector<char> occ(sr.size());
                        traceRayGroup(*sc, sr.data(), si.data(), occ.data(), (int)sr.size(), true, !(mode & PO_MODE_PLAIN_SHADOW_WALK), flags, counters);
                        for (size_t q = 0; q < sr.size(); q++) finishJobAfter(*f, *c, js[sw[q]], occ[q] != 0, mode);
                    }
                }
            }
        }
}

/* A job reads and writes its own pixel's slots and its own job entry only, so tile rows can be walked side by side:
 * with f->traceThreads > 1 (po_render_frame_mt, the timed CPU baseline) they are, without the visit statistics. */
PO_API void po_trace(const PoScene* sc, PoFrame* f, const PtShaderConstants* c, int subgroup, int mode, int flags)
{
    const uint32_t H = (uint32_t)c->viewportSize[1], rows = (H + 7) / 8;
    int nt = f->traceThreads;
    if (nt <= 1 || rows < 2) { trace_tile_rows(sc, f, c, subgroup, mode, flags, 0, rows, &f->counters); return; }
    if ((uint32_t)nt > rows) nt = (int)rows;
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&]() { for (uint32_t r = next.fetch_add(1); r < rows; r = next.fetch_add(1)) trace_tile_rows(sc, f, c, subgroup, mode, flags, r, r + 1, nullptr); });
    for (auto& t : th) t.join();
}

/* raysort.comp:23-55: per (row, 256-wide segment) the live jobs move to the front of the segment.
 * The reference's order inside a segment is atomic-arrival order; this restatement keeps the
 * original order, which is one of the orders the reference can produce. */
PO_API void po_raysort(PoFrame* f, const PtShaderConstants* c)
{
    const uint32_t W = (uint32_t)c->viewportSize[0], H = (uint32_t)c->viewportSize[1];
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x0 = 0; x0 < W; x0 += 256) {
            uint32_t base = y * W + x0, n = (W - x0 < 256) ? W - x0 : 256, cnt = 0;
            uint32_t live[256];
            for (uint32_t k = 0; k < n; k++) if (f->jobs[base + k] != PT_JOB_DEAD) live[cnt++] = f->jobs[base + k];
            for (uint32_t k = 0; k < n; k++) f->jobs[base + k] = (k < cnt) ? live[k] : PT_JOB_DEAD;
        }
}

/* composite.frag:30-55 (path resolve + accumulate; the display value of :57 is po_tonemap) */
PO_API void po_composite(PoFrame* f, const PtShaderConstants* c)
{
    const uint32_t W = (uint32_t)c->viewportSize[0], H = (uint32_t)c->viewportSize[1];
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            size_t pix = (size_t)y * W + x;
            size_t stackIndex = pix * c->numRays;
            float L[3] = { 0, 0, 0 };
            for (int depth = (int)c->numRays - 1; depth >= 0; depth--) {
                const PtRayStackSlot& s = f->stack[stackIndex + depth];
                float prob = h2f(hfrombits(s.prob));
                if (prob > 0.0f) {
                    for (int k = 0; k < 3; k++) { L[k] *= prob; L[k] += h2f(hfrombits(s.wIn[k])); L[k] *= h2f(hfrombits(s.hitAlbedo[k])); }
                } else {
                    L[0] = L[1] = L[2] = 0.0f;
                }
            }
            for (int k = 0; k < 3; k++) { f->current[4 * pix + k] = hbits(f2h(L[k])); f->accumulation[4 * pix + k] += L[k]; }
            f->current[4 * pix + 3] = hbits(f2h(1.0f));
            f->accumulation[4 * pix + 3] += 1.0f;
        }
}

/* Display colour, composite.frag:18-26 (ACESFilm) and :57: pow(ACESFilm(acc.rgb / acc.w), 1/2.2), fp32.
 * clamp(x,0,1) is min(max(x,0),1) with minNum/maxNum, so 0/0 shows as black. */
PO_API void po_tonemap(const float* accumulation, uint64_t npix, float* rgb)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    for (uint64_t p = 0; p < npix; p++)
        for (int k = 0; k < 3; k++) {
            float x = accumulation[4 * p + k] / accumulation[4 * p + 3];
            float v = (x * (a * x + b)) / (x * (c * x + d) + e);
            v = fminf(fmaxf(v, 0.0f), 1.0f);
            rgb[3 * p + k] = powf(v, 1.0f / 2.2f);
        }
}

/* One frame as PathTracer.cpp:776-838 records it: launch, numRays x {trace (speculative at depth 0
 * when subgroup0 > 0), raysort except after the last depth}, composite. */
PO_API void po_render_frame(const PoScene* sc, PoFrame* f, const PtShaderConstants* c, int subgroup0, int ray_sort, int mode, int flags)
{
    po_launch(sc, f, c, mode);
    for (uint32_t d = 0; d < c->numRays; d++) {
        po_trace(sc, f, c, d == 0 ? subgroup0 : 0, mode, flags);
        if (ray_sort && d != c->numRays - 1) po_raysort(f, c);
    }
    po_composite(f, c);
}

/* The same frame with the trace passes on num_threads threads (tile rows dealt dynamically); every buffer comes out as
 * po_render_frame leaves it, only the box/triangle test statistics are not kept.  For the timed CPU baseline. */
PO_API void po_render_frame_mt(const PoScene* sc, PoFrame* f, const PtShaderConstants* c, int subgroup0, int ray_sort, int mode, int flags, int num_threads)
{
    if (num_threads <= 0) num_threads = (int)std::thread::hardware_concurrency();
    f->traceThreads = num_threads < 1 ? 1 : num_threads;
    po_render_frame(sc, f, c, subgroup0, ray_sort, mode, flags);
    f->traceThreads = 1;
}

/* Exposes single operations for unit tests of the fp16 helpers. */
PO_API int po_intersect_bbox(const PtRay* ray, const float a[3], const float b[3])
{
    Ray r; r.o = { ray->o[0], ray->o[1], ray->o[2] }; r.d = { ray->d[0], ray->d[1], ray->d[2] };
    r.min_t = ray->min_t; r.max_t = ray->max_t; r.origBvhId = 0; r.rcpD = rcp_dir(r.d);
    return intersectBBox(r, a, b) ? 1 : 0;
}
PO_API void po_weyl(uint32_t n, uint16_t out[2]) { h16 w[2]; WeylNth(n, w); out[0] = hbits(w[0]); out[1] = hbits(w[1]); }
PO_API void po_cosine_hemisphere(const uint16_t in[2], uint16_t out[3])
{
    h16 i[2] = { hfrombits(in[0]), hfrombits(in[1]) };
    H3 v = cosineHemisphere(i); out[0] = hbits(v.x); out[1] = hbits(v.y); out[2] = hbits(v.z);
}
PO_API void po_to_coord_space(const uint16_t n[3], const uint16_t r[3], uint16_t out[3])
{
    H3 v = to_coord_space({ hfrombits(n[0]), hfrombits(n[1]), hfrombits(n[2]) }, { hfrombits(r[0]), hfrombits(r[1]), hfrombits(r[2]) });
    out[0] = hbits(v.x); out[1] = hbits(v.y); out[2] = hbits(v.z);
}
