/*
 * bvh_build.cpp — host-side BVH construction.
 *
 * Produces the node array that BuildBVH (reference BVH.cpp:83-309) produces, bit for bit, but
 * is organised for many cores instead of one:
 *   - a subtree over n triangles always has 2n-1 nodes in depth-first pre-order, so every
 *     node's index (left = i+1, right = i+2*nLeft) and skip link are known before the subtree
 *     is built; subtrees are therefore built independently, in any order, straight into place;
 *   - centroids and per-triangle bounds are computed once, and triangles are tracked through
 *     a permutation instead of re-gathering index triplets at every level;
 *   - all 9 split axes are binned in one sweep over a node's triangles;
 *   - the nodes near the root are processed one at a time with every thread sweeping a chunk,
 *     the remaining small subtrees are handed out to threads whole.
 * The arithmetic of every comparison and score follows the reference expression by
 * expression (including its float/double promotions); only the order of exact min/max
 * reductions differs, which cannot change a result.
 */
#include "../../include/ptcore.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#pragma STDC FP_CONTRACT OFF

namespace {

struct V3 { float x, y, z; };
inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }      /* glm::dot: (x+y)+z */
inline float fminr(float a, float b) { return b < a ? b : a; }                      /* glm::min(a,b) */
inline float fmaxr(float a, float b) { return a < b ? b : a; }                      /* glm::max(a,b) */

struct Box {
    V3 a{0, 0, 0}, b{0, 0, 0};
    bool empty = true;
    inline void extend(const V3& lo, const V3& hi)
    {
        if (empty) { a = lo; b = hi; empty = false; return; }
        a = { fminr(a.x, lo.x), fminr(a.y, lo.y), fminr(a.z, lo.z) };
        b = { fmaxr(b.x, hi.x), fmaxr(b.y, hi.y), fmaxr(b.z, hi.z) };
    }
    inline void extend(const Box& o) { if (!o.empty) extend(o.a, o.b); }
    inline V3 size() const                                                         /* BBox::GetSize, BVH.cpp:50-53 */
    {
        return { fmaxr(0.0f, b.x - a.x), fmaxr(0.0f, b.y - a.y), fmaxr(0.0f, b.z - a.z) };
    }
};

struct TriPre { V3 centroid, lo, hi; };

struct Bins {
    Box box[9][8];
    uint32_t count[9][8];
    Bins() { memset(count, 0, sizeof(count)); }
    void merge(const Bins& o)
    {
        for (int x = 0; x < 9; x++)
            for (int j = 0; j < 8; j++) { box[x][j].extend(o.box[x][j]); count[x][j] += o.count[x][j]; }
    }
};

struct SplitPlan {                  /* everything about a node that does not depend on the sweep */
    V3 axes[9];
    double lo[9][8], hi[9][8];      /* bin j of axis x holds l with lo < l < hi (BVH.cpp:179) */
};

struct Builder {
    const float* pos4;
    const uint32_t* idxIn;
    uint32_t T;
    std::vector<TriPre> pre;
    std::vector<uint32_t> order;    /* permutation: position -> original triangle */
    std::vector<uint32_t> scratch;
    PtBVHNode* nodes;
    V3 axes[9];
    int nthreads;
    std::atomic<uint32_t> maxDepth{0};

    V3 vtx(uint32_t i) const { return { pos4[4 * (size_t)i], pos4[4 * (size_t)i + 1], pos4[4 * (size_t)i + 2] }; }

    template <class F> void parallel_for(uint32_t n, F f)
    {
        int nt = std::min<uint32_t>(nthreads, std::max<uint32_t>(1, n / 4096));
        if (nt <= 1) { f(0, 0, n); return; }
        std::vector<std::thread> th;
        th.reserve(nt);
        for (int t = 0; t < nt; t++) {
            uint32_t lo = (uint64_t)n * t / nt, hi = (uint64_t)n * (t + 1) / nt;
            /* a thread the system refuses to start: its slice runs here instead (the slices do not allocate) */
            try { th.emplace_back([=, &f] { f(t, lo, hi); }); } catch (const std::system_error&) { f(t, lo, hi); }
        }
        for (auto& t : th) t.join();
    }

    void precompute()
    {
        pre.resize(T);
        order.resize(T);
        parallel_for(T, [&](int, uint32_t lo, uint32_t hi) {
            for (uint32_t t = lo; t < hi; t++) {
                V3 p0 = vtx(idxIn[3 * (size_t)t]), p1 = vtx(idxIn[3 * (size_t)t + 1]), p2 = vtx(idxIn[3 * (size_t)t + 2]);
                TriPre& q = pre[t];
                /* (p0 + p1 + p2) / 3.0f, BVH.cpp:170 */
                q.centroid = { ((p0.x + p1.x) + p2.x) / 3.0f, ((p0.y + p1.y) + p2.y) / 3.0f, ((p0.z + p1.z) + p2.z) / 3.0f };
                q.lo = { fminr(fminr(p0.x, p1.x), p2.x), fminr(fminr(p0.y, p1.y), p2.y), fminr(fminr(p0.z, p1.z), p2.z) };
                q.hi = { fmaxr(fmaxr(p0.x, p1.x), p2.x), fmaxr(fmaxr(p0.y, p1.y), p2.y), fmaxr(fmaxr(p0.z, p1.z), p2.z) };
                order[t] = t;
            }
        });
        auto nrm = [](V3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return V3{ v.x * s, v.y * s, v.z * s }; };   /* glm::normalize */
        const V3 raw[9] = { {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, 0, 1}, {1, 0, 1}, {1, -1, 0}, {-1, -1, 0}, {0, 1, -1}, {0, -1, -1} };
        for (int i = 0; i < 9; i++) axes[i] = i < 3 ? raw[i] : nrm(raw[i]);                                     /* BVH.cpp:144-154 */
    }

    Box bounds(uint32_t s, uint32_t e) const
    {
        Box bb;
        for (uint32_t k = s; k < e; k++) bb.extend(pre[order[k]].lo, pre[order[k]].hi);
        return bb;
    }

    static void plan_splits(const Box& bbox, const V3* axes, float& radius, V3& centre, SplitPlan& sp)
    {
        const float splitRatio[9] = { -1e24f, 0.125f, 0.25f, 0.375f, 0.5f, 0.625f, 0.75f, 0.875f, 1e24f };
        V3 d = { bbox.a.x - bbox.b.x, bbox.a.y - bbox.b.y, bbox.a.z - bbox.b.z };
        radius = sqrtf(dot3(d, d));                                                              /* glm::distance, BVH.cpp:141 */
        centre = { (bbox.a.x + bbox.b.x) * 0.5f, (bbox.a.y + bbox.b.y) * 0.5f, (bbox.a.z + bbox.b.z) * 0.5f };
        for (int x = 0; x < 9; x++) {
            sp.axes[x] = axes[x];
            float pos[9];
            for (int j = 0; j < 9; j++)      /* float + float*double evaluated in double, stored to float (BVH.cpp:174-175) */
                pos[j] = (float)((double)dot3(centre, axes[x]) + (double)radius * ((double)splitRatio[j] - 0.5));
            for (int j = 0; j < 8; j++) { sp.lo[x][j] = (double)pos[j] - 0.00001; sp.hi[x][j] = (double)pos[j + 1] + 0.00001; }
        }
    }

    void sweep(uint32_t s, uint32_t e, const SplitPlan& sp, Bins& bins) const
    {
        for (uint32_t k = s; k < e; k++) {
            const TriPre& q = pre[order[k]];
            for (int x = 0; x < 9; x++) {
                double l = (double)dot3(q.centroid, sp.axes[x]);
                for (int j = 0; j < 8; j++)
                    if (l > sp.lo[x][j] && l < sp.hi[x][j]) { bins.box[x][j].extend(q.lo, q.hi); bins.count[x][j]++; }
            }
        }
    }

    /* BVH.cpp:189-230: pick the best of 9 x 7 candidate planes; returns the split position, axis in selAxis. */
    static float choose(const Bins& bins, const Box& bbox, uint32_t count, float radius, V3 centre, const V3* axes, V3& selAxis)
    {
        const float splitRatio[9] = { -1e24f, 0.125f, 0.25f, 0.375f, 0.5f, 0.625f, 0.75f, 0.875f, 1e24f };
        V3 bs = bbox.size();
        float totalArea = dot3(bs, bs);
        float maxScore = -(1.0f + 2.0f * (float)(count * 3u) / 3.0f);                           /* BVH.cpp:136 */
        float selectedRatio = 0.5f;
        selAxis = { 0, 0, 0 };
        for (int x = 0; x < 9; x++)
            for (int i = 1; i < 8; i++) {
                Box left, right;
                uint32_t lc = 0, rc = 0;
                for (int j = 0; j < i; j++) { left.extend(bins.box[x][j]); lc += bins.count[x][j]; }
                for (int j = i; j < 8; j++) { right.extend(bins.box[x][j]); rc += bins.count[x][j]; }
                V3 ls = left.size(), rs = right.size();
                float pA = dot3(ls, ls) / totalArea, pB = dot3(rs, rs) / totalArea;
                float score = -((1.0f + pA * (float)lc * 2.0f) + pB * (float)rc * 2.0f);         /* BVH.cpp:211-215 */
                if (score > maxScore) { maxScore = score; selectedRatio = splitRatio[i]; selAxis = axes[x]; }
            }
        return (float)((double)dot3(centre, selAxis) + (double)radius * ((double)selectedRatio - 0.5));   /* BVH.cpp:230 */
    }

    struct Sub { uint32_t s, e, node; int32_t next; uint32_t depth; };

    void note_depth(uint32_t d)
    {
        uint32_t cur = maxDepth.load(std::memory_order_relaxed);
        while (d > cur && !maxDepth.compare_exchange_weak(cur, d, std::memory_order_relaxed)) {}
    }

    void write_box(PtBVHNode& nd, const Box& bbox)
    {
        nd.a[0] = bbox.a.x - 0.00006f; nd.a[1] = bbox.a.y - 0.00006f; nd.a[2] = bbox.a.z - 0.00006f;    /* BVH.cpp:117-118 */
        nd.b[0] = bbox.b.x + 0.00006f; nd.b[1] = bbox.b.y + 0.00006f; nd.b[2] = bbox.b.z + 0.00006f;
    }

    /* Emits the node of task t; returns false for a leaf, else fills the two child tasks. */
    template <bool PAR> bool process(const Sub& t, Sub& left, Sub& right, std::vector<uint32_t>& tmp)
    {
        PtBVHNode& nd = nodes[t.node];
        nd.next = t.next;
        note_depth(t.depth);
        const uint32_t n = t.e - t.s;
        Box bbox;
        if (PAR) {
            std::vector<Box> part(nthreads);
            parallel_for(n, [&](int tid, uint32_t lo, uint32_t hi) { part[tid] = bounds(t.s + lo, t.s + hi); });
            for (auto& b : part) bbox.extend(b);
        } else {
            bbox = bounds(t.s, t.e);
        }
        write_box(nd, bbox);
        if (n == 1) { nd.right = -(int32_t)(t.s * 3u); return false; }                          /* BVH.cpp:120-130 */

        float radius; V3 centre; SplitPlan sp;
        plan_splits(bbox, axes, radius, centre, sp);
        Bins bins;
        if (PAR) {
            std::vector<Bins> part(nthreads);
            parallel_for(n, [&](int tid, uint32_t lo, uint32_t hi) { sweep(t.s + lo, t.s + hi, sp, part[tid]); });
            for (auto& b : part) bins.merge(b);
        } else {
            sweep(t.s, t.e, sp, bins);
        }
        V3 selAxis;
        float splitPos = choose(bins, bbox, n, radius, centre, axes, selAxis);

        /* stable partition by centroid (BVH.cpp:235-267): left keeps `dot <= splitPos`, right `dot > splitPos` */
        uint32_t nLeft = 0;
        if (PAR) {
            std::vector<uint32_t> cl(nthreads + 1, 0), lo_(nthreads, 0), hi_(nthreads, 0);
            parallel_for(n, [&](int tid, uint32_t lo, uint32_t hi) {
                uint32_t c = 0;
                for (uint32_t k = lo; k < hi; k++) c += !(dot3(pre[order[t.s + k]].centroid, selAxis) > splitPos);
                cl[tid + 1] = c; lo_[tid] = lo; hi_[tid] = hi;
            });
            std::vector<uint32_t> lbase(nthreads + 1, 0), rbase(nthreads + 1, 0);
            for (int i = 0; i < nthreads; i++) { lbase[i + 1] = lbase[i] + cl[i + 1]; rbase[i + 1] = rbase[i] + ((hi_[i] - lo_[i]) - cl[i + 1]); }
            nLeft = lbase[nthreads];
            if (scratch.size() < T) scratch.resize(T);
            parallel_for(n, [&](int tid, uint32_t lo, uint32_t hi) {
                uint32_t wl = lbase[tid], wr = nLeft + rbase[tid];
                for (uint32_t k = lo; k < hi; k++) {
                    uint32_t id = order[t.s + k];
                    if (dot3(pre[id].centroid, selAxis) > splitPos) scratch[t.s + wr++] = id; else scratch[t.s + wl++] = id;
                }
            });
            parallel_for(n, [&](int, uint32_t lo, uint32_t hi) { memcpy(&order[t.s + lo], &scratch[t.s + lo], (size_t)(hi - lo) * 4); });
        } else {
            tmp.clear();
            uint32_t w = t.s;
            for (uint32_t k = t.s; k < t.e; k++) {
                uint32_t id = order[k];
                if (dot3(pre[id].centroid, selAxis) > splitPos) tmp.push_back(id); else order[w++] = id;
            }
            nLeft = w - t.s;
            if (!tmp.empty()) memcpy(&order[w], tmp.data(), tmp.size() * 4);
        }
        if (nLeft == 0 || nLeft == n) nLeft = n / 2;                                             /* median fallback, BVH.cpp:269-275 */

        uint32_t rightNode = t.node + 2 * nLeft;
        nd.right = (int32_t)rightNode;
        left = { t.s, t.s + nLeft, t.node + 1, (int32_t)rightNode, t.depth + 1 };                /* skip links: BVH.cpp:284-304 */
        right = { t.s + nLeft, t.e, rightNode, t.next, t.depth + 1 };
        return true;
    }

    void run()
    {
        precompute();
        /* nodes at least this large are swept by all threads together; smaller subtrees go to one thread each */
        const uint32_t PAR_MIN = std::max<uint32_t>(4096, T / (8u * (uint32_t)nthreads));
        std::vector<Sub> big = { { 0, T, 0, 0, 0 } }, small;
        std::vector<uint32_t> tmp;
        while (!big.empty()) {
            Sub t = big.back(); big.pop_back();
            if (t.e - t.s < PAR_MIN || nthreads == 1) { small.push_back(t); continue; }
            Sub l, r;
            if (process<true>(t, l, r, tmp)) { big.push_back(r); big.push_back(l); }
        }
        std::sort(small.begin(), small.end(), [](const Sub& a, const Sub& b) { return (a.e - a.s) > (b.e - b.s); });
        std::atomic<size_t> cursor{0};
        std::atomic<bool> failed{false};
        auto worker = [&] {
            try {
                std::vector<Sub> st;
                std::vector<uint32_t> tl;
                for (;;) {
                    size_t i = cursor.fetch_add(1);
                    if (i >= small.size() || failed.load()) break;
                    st.push_back(small[i]);
                    while (!st.empty()) {
                        Sub t = st.back(); st.pop_back();
                        Sub l, r;
                        if (process<false>(t, l, r, tl)) { st.push_back(r); st.push_back(l); }
                    }
                }
            } catch (...) { failed.store(true); }      /* out of memory in a worker: reported by the caller's thread */
        };
        int nt = (int)std::min<size_t>(nthreads, small.size());
        if (nt <= 1) worker();
        else {
            std::vector<std::thread> th;
            th.reserve(nt);
            for (int i = 0; i + 1 < nt; i++) {
                try { th.emplace_back(worker); } catch (const std::system_error&) { break; }      /* fewer threads than asked for */
            }
            worker();                                   /* this thread helps; with no thread started it does it all */
            for (auto& t : th) t.join();
        }
        if (failed.load()) throw std::bad_alloc();
    }
};

void set_err(char* err, size_t n, const char* msg) { if (err && n) { snprintf(err, n, "%s", msg); } }

}  // namespace

extern "C" PT_API int pt_build_bvh(const float* positions, uint32_t num_vertices, uint32_t* indices, uint32_t num_triangles,
                                   PtBVHNode* nodes, uint32_t* max_depth, int num_threads, char* err, size_t err_len)
{
    if (!positions || !indices || !nodes) { set_err(err, err_len, "pt_build_bvh: null argument"); return -1; }
    if (num_triangles == 0) { set_err(err, err_len, "pt_build_bvh: no triangles"); return -1; }
    if (num_triangles > 0x2AAAAAAAu) { set_err(err, err_len, "pt_build_bvh: too many triangles for int32 leaf offsets"); return -1; }
    for (size_t i = 0; i < 3 * (size_t)num_triangles; i++)
        if (indices[i] >= num_vertices) { set_err(err, err_len, "pt_build_bvh: vertex index out of range"); return -2; }

    try {
        Builder b;
        b.pos4 = positions; b.idxIn = indices; b.T = num_triangles; b.nodes = nodes;
        b.nthreads = num_threads > 0 ? num_threads : (int)std::thread::hardware_concurrency();
        if (b.nthreads < 1) b.nthreads = 1;
        if (b.nthreads > 256) b.nthreads = 256;
        b.run();

        /* reorder the index triplets as the reference's in-place partitions leave them (BVH.cpp:257-267) */
        std::vector<uint32_t> out(3 * (size_t)num_triangles);
        for (size_t k = 0; k < num_triangles; k++) {
            size_t t = b.order[k];
            out[3 * k] = indices[3 * t]; out[3 * k + 1] = indices[3 * t + 1]; out[3 * k + 2] = indices[3 * t + 2];
        }
        memcpy(indices, out.data(), out.size() * 4);
        if (max_depth) *max_depth = b.maxDepth.load();
    } catch (...) {      /* nothing may unwind through the C boundary; nodes and indices are then unspecified */
        set_err(err, err_len, "pt_build_bvh: out of memory");
        return -3;
    }
    return 0;
}
