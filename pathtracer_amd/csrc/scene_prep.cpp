/*
 * scene_prep.cpp — host-side passes that run after the BVH build and before upload:
 * the flattening pass (threaded 32 B nodes -> 64 B two-child records + inline triangle records)
 * and the per-frame constants block.
 */
#include "../../include/ptcore.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#pragma STDC FP_CONTRACT OFF

namespace {

inline uint16_t f16_bits(float x)
{
    _Float16 h = (_Float16)x;      /* round-to-nearest-even, subnormals kept: the conversion f16vec3() performs */
    uint16_t u; memcpy(&u, &h, 2); return u;
}

/* The ray-independent part of intersectBBox (reference intersections.glsl:3-5):
 * c = (a + b) * 0.5f, hA = f16(a - c), hB = f16(b - c). */
inline void digest_box(const PtBVHNode& n, float c[3], uint16_t h[6])
{
    for (int k = 0; k < 3; k++) {
        c[k] = (n.a[k] + n.b[k]) * 0.5f;
        h[k] = f16_bits(n.a[k] - c[k]);
        h[3 + k] = f16_bits(n.b[k] - c[k]);
    }
}

/* The certified walk's per-box constant: the largest magnitude among the six digested half extents plus 2^-12, rounded up
 * to fp16 (+inf where that leaves the fp16 range, or where a half extent is not finite). */
inline uint16_t half_extent_bound(const uint16_t h[6])
{
    double w = 0.0;
    for (int k = 0; k < 6; k++) {
        const uint16_t m = (uint16_t)(h[k] & 0x7FFFu);
        _Float16 v; memcpy(&v, &m, 2);
        if (m >= 0x7C00u) return 0x7C00u;
        if ((double)v > w) w = (double)v;
    }
    w += 0x1p-12;
    _Float16 r = (_Float16)w;
    uint16_t u; memcpy(&u, &r, 2);
    if ((double)r < w) u++;          /* positive and finite: the next bit pattern is the next value up (or +inf) */
    return u;
}

void set_err(char* err, size_t n, const char* msg) { if (err && n) snprintf(err, n, "%s", msg); }

/* Is the array the pre-order tree BuildBVH lays down (BVH.cpp:124, 284-304)?  Node i's subtree occupies [i, end): a leaf
 * has end = i + 1, an interior node its left child at i + 1 and its right child inside (i + 1, end); every node's skip
 * link names `end`, or 0 where that is the end of the array; every triangle sits in exactly one leaf.  Anything else -
 * a child index pointing back up, a skip link into the middle of a subtree - would send a walk round in circles on
 * the device, so it is refused here. */
const char* validate_tree(const PtBVHNode* nodes, uint32_t num_nodes, uint32_t num_triangles)
{
    if (num_triangles == 0 || num_nodes != 2 * num_triangles - 1) return "node count is not 2T-1";
    std::vector<uint8_t> seen(num_triangles, 0);
    std::vector<uint32_t> st = { 0u, num_nodes };
    while (!st.empty()) {
        const uint32_t end = st.back(); st.pop_back();
        const uint32_t i = st.back(); st.pop_back();
        const PtBVHNode& n = nodes[i];
        if ((uint32_t)n.next != (end == num_nodes ? 0u : end)) return "a skip link does not name the end of its subtree";
        if (n.right > 0) {
            const uint32_t r = (uint32_t)n.right;
            if (r <= i + 1 || r >= end) return "a right child lies outside its parent's subtree";
            st.push_back(r); st.push_back(end);
            st.push_back(i + 1); st.push_back(r);
        } else {
            if (end != i + 1) return "a leaf is followed by nodes of its own subtree";
            const uint32_t off = (uint32_t)(-(int64_t)n.right);
            if (off % 3 != 0 || off / 3 >= num_triangles) return "leaf offset out of range";
            if (seen[off / 3]) return "two leaves hold the same triangle";
            seen[off / 3] = 1;
        }
    }
    return nullptr;
}

/* Shared by both flattening passes: rank of every reference node among the interior nodes (pre-order) or, for a leaf,
 * its triangle index; returns 0 or an error code with a message. */
int rank_nodes(const PtBVHNode* nodes, uint32_t num_nodes, uint32_t num_triangles, std::vector<uint32_t>& rank, char* err, size_t err_len)
{
    if (num_nodes != 2 * num_triangles - 1) { set_err(err, err_len, "pt_flatten_bvh: node count is not 2T-1"); return -2; }
    rank.resize(num_nodes);
    uint32_t ni = 0;
    for (uint32_t i = 0; i < num_nodes; i++) {
        if (nodes[i].right > 0) {
            if ((uint32_t)nodes[i].right >= num_nodes || i + 1 >= num_nodes) { set_err(err, err_len, "pt_flatten_bvh: child index out of range"); return -3; }
            rank[i] = ni++;
        } else {
            uint32_t off = (uint32_t)(-nodes[i].right);
            if (off % 3 != 0 || off / 3 >= num_triangles) { set_err(err, err_len, "pt_flatten_bvh: leaf offset out of range"); return -3; }
            rank[i] = off / 3;
        }
    }
    if (ni != num_triangles - 1) { set_err(err, err_len, "pt_flatten_bvh: interior count is not T-1"); return -2; }
    return 0;
}

uint32_t tree_depth(const PtBVHNode* nodes)
{
    /* depth of the deepest node = the most entries a depth-first walk can have pending */
    uint32_t deepest = 0;
    std::vector<uint32_t> st = { 0 }, dp = { 0 };
    while (!st.empty()) {
        uint32_t i = st.back(), d = dp.back(); st.pop_back(); dp.pop_back();
        if (d > deepest) deepest = d;
        if (nodes[i].right > 0) { st.push_back((uint32_t)nodes[i].right); dp.push_back(d + 1); st.push_back(i + 1); dp.push_back(d + 1); }
    }
    return deepest;
}

}  // namespace

extern "C" PT_API int pt_flatten_bvh(const PtBVHNode* nodes, uint32_t num_nodes, const float* positions, const uint32_t* indices,
                                     uint32_t num_triangles, PtNode2* nodes2, PtTri* tris, float* root_box, uint32_t* max_stack,
                                     char* err, size_t err_len)
try {
    if (!nodes || !positions || !indices || !tris || (num_triangles > 1 && !nodes2)) { set_err(err, err_len, "pt_flatten_bvh: null argument"); return -1; }
    if (num_nodes != 2 * num_triangles - 1) { set_err(err, err_len, "pt_flatten_bvh: node count is not 2T-1"); return -2; }
    if (const char* why = validate_tree(nodes, num_nodes, num_triangles)) { set_err(err, err_len, (std::string("pt_flatten_bvh: ") + why).c_str()); return -3; }
    std::vector<uint32_t> rank;
    if (int rc = rank_nodes(nodes, num_nodes, num_triangles, rank, err, err_len)) return rc;

    auto child_ref = [&](uint32_t i) { return nodes[i].right > 0 ? rank[i] : (PT_LEAF_BIT | rank[i]); };

    for (uint32_t i = 0; i < num_nodes; i++) {
        const PtBVHNode& n = nodes[i];
        if (n.right > 0) {
            PtNode2& o = nodes2[rank[i]];
            memset(&o, 0, sizeof(o));
            uint16_t hl[6], hr[6]; float cl[3], cr[3];
            digest_box(nodes[i + 1], cl, hl);
            digest_box(nodes[n.right], cr, hr);
            for (int k = 0; k < 3; k++) { o.c[k][0] = cl[k]; o.c[k][1] = cr[k]; }
            for (int k = 0; k < 6; k++) { o.h[k][0] = hl[k]; o.h[k][1] = hr[k]; }
            o.left = child_ref(i + 1);
            o.right = child_ref((uint32_t)n.right);
            o.refIndex = i;
            o.wmax[0] = half_extent_bound(hl); o.wmax[1] = half_extent_bound(hr);
        } else {
            uint32_t t = rank[i];
            PtTri& o = tris[t];
            memset(&o, 0, sizeof(o));
            uint32_t i1 = indices[3 * (size_t)t], i2 = indices[3 * (size_t)t + 1], i3 = indices[3 * (size_t)t + 2];
            const float* p1 = positions + 4 * (size_t)i1; const float* p2 = positions + 4 * (size_t)i2; const float* p3 = positions + 4 * (size_t)i3;
            for (int k = 0; k < 3; k++) { o.p1[k] = p1[k]; o.e1[k] = p2[k] - p1[k]; o.e2[k] = p3[k] - p1[k]; }   /* intersections.glsl:34-35 */
            o.i1 = (int32_t)i1; o.i2 = (int32_t)i2; o.i3 = (int32_t)i3;
            o.refLeaf = i;
            o.flags = (i == num_nodes - 1) ? 1u : 0u;
        }
    }
    if (root_box) { for (int k = 0; k < 3; k++) { root_box[k] = nodes[0].a[k]; root_box[3 + k] = nodes[0].b[k]; } }

    /* which subtrees hold a triangle whose test is ill-conditioned for every ray: |e1 x e2| < 2^-14 |e1| |e2|.  Children
       come after their parent in the array, so one backward sweep settles every node. */
    {
        std::vector<uint8_t> thin(num_nodes, 0);
        for (uint32_t i = num_nodes; i-- > 0;) {
            const PtBVHNode& n = nodes[i];
            if (n.right > 0) thin[i] = thin[i + 1] | thin[n.right];
            else {
                const PtTri& t = tris[rank[i]];
                const double e1[3] = { t.e1[0], t.e1[1], t.e1[2] }, e2[3] = { t.e2[0], t.e2[1], t.e2[2] };
                const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                const double c2 = cx * cx + cy * cy + cz * cz;
                const double l2 = (e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) * (e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
                thin[i] = !(c2 >= l2 * 0x1p-28) ? 1 : 0;          /* NaN counts as thin */
            }
        }
        for (uint32_t i = 0; i < num_nodes; i++)
            if (nodes[i].right > 0) {
                if (thin[i + 1]) nodes2[rank[i]].wmax[0] = 0x7C00u;
                if (thin[nodes[i].right]) nodes2[rank[i]].wmax[1] = 0x7C00u;
            }
    }

    if (max_stack) *max_stack = tree_depth(nodes);
    return 0;
} catch (...) { set_err(err, err_len, "pt_flatten_bvh: out of memory"); return -4; }

extern "C" PT_API int pt_flatten_bvh_compact(const PtBVHNode* nodes, uint32_t num_nodes, const float* positions, const uint32_t* indices,
                                             uint32_t num_triangles, PtNode2c* nodes2, PtTriC* tris, uint32_t* leaf_ref, float* root_box,
                                             uint32_t* max_stack, char* err, size_t err_len)
try {
    if (!nodes || !positions || !indices || !tris || !leaf_ref || (num_triangles > 1 && !nodes2)) { set_err(err, err_len, "pt_flatten_bvh_compact: null argument"); return -1; }
    if (const char* why = validate_tree(nodes, num_nodes, num_triangles)) { set_err(err, err_len, (std::string("pt_flatten_bvh_compact: ") + why).c_str()); return -3; }
    std::vector<uint32_t> rank;
    if (int rc = rank_nodes(nodes, num_nodes, num_triangles, rank, err, err_len)) return rc;
    auto child_ref = [&](uint32_t i) { return nodes[i].right > 0 ? rank[i] : (PT_LEAF_BIT | rank[i]); };

    for (uint32_t i = 0; i < num_nodes; i++) {
        const PtBVHNode& n = nodes[i];
        if (n.right > 0) {
            PtNode2c& o = nodes2[rank[i]];
            memset(&o, 0, sizeof(o));
            const uint32_t child[2] = { i + 1, (uint32_t)n.right };
            for (int s = 0; s < 2; s++) {
                float c[3]; uint16_t h[6];
                digest_box(nodes[child[s]], c, h);
                for (int k = 0; k < 3; k++) {
                    o.c[k][s] = c[k];
                    o.hA[k][s] = h[k];
                    const int d = (int)h[3 + k] - (int)(h[k] & 0x7FFFu);        /* sign of f16(b-c) must be 0: it is part of the value */
                    if ((h[3 + k] & 0x8000u) || d < -15 || d > 15) {
                        set_err(err, err_len, "pt_flatten_bvh_compact: a box does not fit the 48 B encoding");
                        return 1;
                    }
                    o.delta |= (uint32_t)(d + 15) << (5 * k + 16 * s);
                }
            }
            o.left = child_ref(child[0]);
            o.right = child_ref(child[1]);
        } else {
            uint32_t t = rank[i];
            PtTriC& o = tris[t];
            uint32_t i1 = indices[3 * (size_t)t], i2 = indices[3 * (size_t)t + 1], i3 = indices[3 * (size_t)t + 2];
            const float* p1 = positions + 4 * (size_t)i1; const float* p2 = positions + 4 * (size_t)i2; const float* p3 = positions + 4 * (size_t)i3;
            for (int k = 0; k < 3; k++) { o.p1[k] = p1[k]; o.e1[k] = p2[k] - p1[k]; o.e2[k] = p3[k] - p1[k]; }   /* intersections.glsl:34-35 */
            o.i1 = (int32_t)i1; o.i2 = (int32_t)i2; o.i3 = (int32_t)i3;
            leaf_ref[t] = i;
        }
    }
    if (root_box) { for (int k = 0; k < 3; k++) { root_box[k] = nodes[0].a[k]; root_box[3 + k] = nodes[0].b[k]; } }
    if (max_stack) *max_stack = tree_depth(nodes);
    return 0;
} catch (...) { set_err(err, err_len, "pt_flatten_bvh_compact: out of memory"); return -4; }

/* ------------------------------------------------------------------------------------------
 * Constants block — the values PathTracer.cpp:731-750 writes: glm::lookAt (right-handed),
 * glm::perspective(60 deg, W/H, 0.01, 256) with [1][1] negated, and their inverses.
 * ---------------------------------------------------------------------------------------- */
namespace {

struct V3 { float x, y, z; };
inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross3(V3 a, V3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline V3 norm3(V3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return { v.x * s, v.y * s, v.z * s }; }

/* column-major 4x4 inverse by cofactors */
void invert4(const float* m, float* out)
{
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    float r = 1.0f / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * r;
}

}  // namespace

extern "C" PT_API void pt_fill_constants(PtShaderConstants* c, uint32_t width, uint32_t height, const float focus_center[3],
                                         float orbit_height, float orbit_radius, float orbit_angle, uint32_t num_lights,
                                         uint32_t num_triangles, uint32_t frame_index, uint32_t num_rays, uint32_t num_bvh_nodes,
                                         const float ambient[3])
{
    memset(c, 0, sizeof(*c));
    V3 centre = { focus_center[0], focus_center[1], focus_center[2] };
    V3 eye = { cosf(orbit_angle) * orbit_radius + centre.x, orbit_height + centre.y, sinf(orbit_angle) * orbit_radius + centre.z };   /* :734 */
    V3 f = norm3({ centre.x - eye.x, centre.y - eye.y, centre.z - eye.z });
    V3 s = norm3(cross3(f, { 0, 1, 0 }));
    V3 u = cross3(s, f);
    float* v = c->viewMtx;
    v[0] = s.x; v[4] = s.y; v[8] = s.z;
    v[1] = u.x; v[5] = u.y; v[9] = u.z;
    v[2] = -f.x; v[6] = -f.y; v[10] = -f.z;
    v[12] = -dot3(s, eye); v[13] = -dot3(u, eye); v[14] = dot3(f, eye); v[15] = 1.0f;

    const float fovy = 60.0f * 0.01745329251994329576923690768489f, zNear = 0.01f, zFar = 256.0f;                                     /* :735 */
    float aspect = (float)width / (float)height;
    float th = tanf(fovy / 2.0f);
    float* p = c->projMtx;
    p[0] = 1.0f / (aspect * th);
    p[5] = -(1.0f / th);                                                                                                              /* :737 */
    p[10] = -(zFar + zNear) / (zFar - zNear);
    p[11] = -1.0f;
    p[14] = -(2.0f * zFar * zNear) / (zFar - zNear);

    invert4(c->viewMtx, c->viewInvMtx);                                                                                               /* :739-740 */
    invert4(c->projMtx, c->projInvMtx);
    c->viewportSize[0] = (float)width; c->viewportSize[1] = (float)height;
    c->numLights = num_lights; c->numTriangles = num_triangles; c->frameIndex = frame_index;
    c->numRays = num_rays; c->numBVHNodes = num_bvh_nodes;
    c->ambientRadiance[0] = ambient[0]; c->ambientRadiance[1] = ambient[1]; c->ambientRadiance[2] = ambient[2];
}

/* ------------------------------------------------------------------------------------------
 * The flattening pass with two-level treelet records (PtTreelet): records and leaves in one
 * depth-first stream of 64 B units.
 * ---------------------------------------------------------------------------------------- */
namespace {

/* One box of a pair: digest + 5-bit differences.  False if f16(b-c) cannot be rebuilt from f16(a-c). */
bool encode_half(const PtBVHNode& n, PtTreeletPair& o, int s)
{
    float c[3]; uint16_t h[6];
    digest_box(n, c, h);
    for (int k = 0; k < 3; k++) {
        o.c[k][s] = c[k];
        o.hA[k][s] = h[k];
        const int d = (int)h[3 + k] - (int)(h[k] & 0x7FFFu);
        if ((h[3 + k] & 0x8000u) || d < -15 || d > 15) return false;
        o.delta |= (uint32_t)(d + 15) << (5 * k + 16 * s);
    }
    return true;
}

enum : uint64_t { TREELET_OFF_LIMIT = 1ull << 21, TREELET_BIG_STEP = 32 };

}  // namespace

extern "C" PT_API int pt_flatten_bvh_treelets(const PtBVHNode* nodes, uint32_t num_nodes, const float* positions, const uint32_t* indices,
                                              uint32_t num_triangles, void* units, uint64_t capacity_units, uint64_t* num_units,
                                              float* root_box, uint32_t* max_stack, char* err, size_t err_len)
try {
    if (!nodes || !positions || !indices || !num_units) { set_err(err, err_len, "pt_flatten_bvh_treelets: null argument"); return -1; }
    if (const char* why = validate_tree(nodes, num_nodes, num_triangles)) { set_err(err, err_len, (std::string("pt_flatten_bvh_treelets: ") + why).c_str()); return -3; }
    const uint32_t N = num_nodes;
    auto is_leaf = [&](uint32_t i) { return nodes[i].right <= 0; };
    auto subtree_nodes = [&](uint32_t i) { const uint32_t e = nodes[i].next ? (uint32_t)nodes[i].next : N; return (uint64_t)(e - i); };

    /* pass 1: where every leaf and every treelet root goes.  Depth first: a root's record, then the subtrees of its
       targets in order, so that the first target starts right behind the record. */
    std::vector<uint32_t> unitOf(N, 0xFFFFFFFFu);
    struct Frame { uint32_t node, u; int k, nt; uint32_t tgt[4]; bool big; };
    std::vector<Frame> st;
    uint64_t pos = 0;
    size_t deepest = 0;
    auto open = [&](uint32_t i) {
        if (is_leaf(i)) { unitOf[i] = (uint32_t)pos++; return; }
        if (pos & 1) pos++;                                   /* records start on a 128 B line */
        Frame f; f.node = i; f.u = (uint32_t)pos; pos += 2; f.k = 0; f.nt = 0;
        unitOf[i] = f.u;
        const uint32_t L = i + 1, R = (uint32_t)nodes[i].right;
        if (is_leaf(L)) f.tgt[f.nt++] = L; else { f.tgt[f.nt++] = L + 1; f.tgt[f.nt++] = (uint32_t)nodes[L].right; }
        if (is_leaf(R)) f.tgt[f.nt++] = R; else { f.tgt[f.nt++] = R + 1; f.tgt[f.nt++] = (uint32_t)nodes[R].right; }
        /* at most three units per node (a record and its padding, or a leaf), plus the padding of the few big records above */
        f.big = 3 * subtree_nodes(i) + 8192 >= TREELET_OFF_LIMIT;
        st.push_back(f);
        if (st.size() > deepest) deepest = st.size();
    };
    open(0);
    while (!st.empty()) {
        Frame& f = st.back();
        if (f.k == f.nt) { st.pop_back(); continue; }
        const int k = f.k++;
        const uint32_t t = f.tgt[k], u = f.u;
        if (f.big && k > 0) while ((pos - u) % TREELET_BIG_STEP) pos++;
        open(t);                                              /* may grow the stack: f is not used after this */
    }
    if (pos >= (1ull << 26)) { set_err(err, err_len, "pt_flatten_bvh_treelets: more than 2^26 units (4 GiB) of records"); return -2; }
    *num_units = pos;
    if (max_stack) *max_stack = (uint32_t)(3 * deepest);      /* a level of the walk leaves at most three entries behind */
    if (root_box) { for (int k = 0; k < 3; k++) { root_box[k] = nodes[0].a[k]; root_box[3 + k] = nodes[0].b[k]; } }
    if (!units) return 0;
    if (capacity_units < pos) { set_err(err, err_len, "pt_flatten_bvh_treelets: output too small"); return -1; }

    /* pass 2: fill */
    unsigned char* out = (unsigned char*)units;
    memset(out, 0, (size_t)pos * 64);
    for (uint32_t i = 0; i < N; i++) {
        if (unitOf[i] == 0xFFFFFFFFu) continue;               /* an interior node that lives inside its parent's record */
        if (is_leaf(i)) {
            PtTri o; memset(&o, 0, sizeof(o));
            const uint32_t t = (uint32_t)(-nodes[i].right) / 3u;
            const uint32_t i1 = indices[3 * (size_t)t], i2 = indices[3 * (size_t)t + 1], i3 = indices[3 * (size_t)t + 2];
            const float* p1 = positions + 4 * (size_t)i1; const float* p2 = positions + 4 * (size_t)i2; const float* p3 = positions + 4 * (size_t)i3;
            for (int k = 0; k < 3; k++) { o.p1[k] = p1[k]; o.e1[k] = p2[k] - p1[k]; o.e2[k] = p3[k] - p1[k]; }   /* intersections.glsl:34-35 */
            o.i1 = (int32_t)i1; o.i2 = (int32_t)i2; o.i3 = (int32_t)i3;
            o.refLeaf = i;
            o.flags = (i == N - 1) ? 1u : 0u;
            memcpy(out + (size_t)unitOf[i] * 64, &o, 64);
            continue;
        }
        PtTreelet o; memset(&o, 0, sizeof(o));
        const uint32_t u = unitOf[i], L = i + 1, R = (uint32_t)nodes[i].right;
        bool ok = encode_half(nodes[L], o.pair[0], 0) && encode_half(nodes[R], o.pair[0], 1);
        uint32_t tgt[4] = { 0, 0, 0, 0 }; bool have[4] = { false, false, false, false };
        if (is_leaf(L)) { o.pair[0].delta |= 1u << 15; tgt[0] = L; have[0] = true; }
        else {
            const uint32_t a = L + 1, b = (uint32_t)nodes[L].right;
            ok = ok && encode_half(nodes[a], o.pair[1], 0) && encode_half(nodes[b], o.pair[1], 1);
            if (is_leaf(a)) o.pair[1].delta |= 1u << 15;
            if (is_leaf(b)) o.pair[1].delta |= 1u << 31;
            tgt[0] = a; tgt[1] = b; have[0] = have[1] = true;
        }
        if (is_leaf(R)) { o.pair[0].delta |= 1u << 31; tgt[2] = R; have[2] = true; }
        else {
            const uint32_t a = R + 1, b = (uint32_t)nodes[R].right;
            ok = ok && encode_half(nodes[a], o.pair[2], 0) && encode_half(nodes[b], o.pair[2], 1);
            if (is_leaf(a)) o.pair[2].delta |= 1u << 15;
            if (is_leaf(b)) o.pair[2].delta |= 1u << 31;
            tgt[2] = a; tgt[3] = b; have[2] = have[3] = true;
        }
        if (!ok) { set_err(err, err_len, "pt_flatten_bvh_treelets: a box does not fit the treelet encoding"); return 1; }
        if (unitOf[tgt[0]] != u + 2) { set_err(err, err_len, "pt_flatten_bvh_treelets: internal error (first target is not behind its record)"); return -5; }
        uint64_t off[4] = { 0, 0, 0, 0 };
        bool big = false;
        for (int k = 1; k < 4; k++) if (have[k]) { off[k] = (uint64_t)unitOf[tgt[k]] - u; if (off[k] >= TREELET_OFF_LIMIT) big = true; }
        if (big) {
            for (int k = 1; k < 4; k++) {
                if (off[k] % TREELET_BIG_STEP || off[k] / TREELET_BIG_STEP >= TREELET_OFF_LIMIT) {
                    set_err(err, err_len, "pt_flatten_bvh_treelets: internal error (a subtree start does not fit its link)"); return -5;
                }
                off[k] /= TREELET_BIG_STEP;
            }
        }
        const uint64_t link = off[1] | (off[2] << 21) | (off[3] << 42) | (big ? 1ull << 63 : 0ull);
        o.link[0] = (uint32_t)link; o.link[1] = (uint32_t)(link >> 32);
        memcpy(out + (size_t)u * 64, &o, 128);
    }
    return 0;
} catch (...) { set_err(err, err_len, "pt_flatten_bvh_treelets: out of memory"); return -4; }
