/*
 * device_math.h — arithmetic shared by the gfx950 kernels.
 *
 * The reference shaders do the box test and all of shading in GLSL float16_t.  Here every
 * fp16 operation is a native CDNA4 half instruction with round-to-nearest-even, subnormals
 * kept (hipcc's default float mode for f16), and no fused multiply-add: the whole file is
 * compiled with contraction off so that `a*b+c` stays two rounded operations.
 * min/max are the IEEE minNum/maxNum forms (v_min_f16 / v_max_f16): a NaN operand yields
 * the other operand.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

typedef _Float16 h16;

struct H3 { h16 x, y, z; };
struct F3 { float x, y, z; };

__device__ __forceinline__ h16 hmin16(h16 a, h16 b) { return __builtin_fminf16(a, b); }
__device__ __forceinline__ h16 hmax16(h16 a, h16 b) { return __builtin_fmaxf16(a, b); }
__device__ __forceinline__ h16 hsqrt16(h16 a) { return (h16)__builtin_sqrtf((float)a); }
__device__ __forceinline__ uint16_t hbits(h16 a) { return __builtin_bit_cast(uint16_t, a); }
__device__ __forceinline__ h16 hfrombits(uint16_t u) { return __builtin_bit_cast(h16, u); }
__device__ __forceinline__ h16 lo16(uint32_t u) { return hfrombits((uint16_t)(u & 0xFFFFu)); }
__device__ __forceinline__ h16 hi16(uint32_t u) { return hfrombits((uint16_t)(u >> 16)); }

__device__ __forceinline__ h16 hdot(H3 a, H3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ H3 hcross(H3 a, H3 b)
{
    return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y };
}
__device__ __forceinline__ H3 hnormalize(H3 v)
{
    h16 inv = (h16)(1.0f / __builtin_sqrtf((float)hdot(v, v)));
    return { v.x * inv, v.y * inv, v.z * inv };
}
__device__ __forceinline__ H3 hscale(h16 s, H3 v) { return { s * v.x, s * v.y, s * v.z }; }
__device__ __forceinline__ H3 hadd3(H3 a, H3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ H3 hsub3(H3 a, H3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ H3 f2h3(F3 v) { return { (h16)v.x, (h16)v.y, (h16)v.z }; }
__device__ __forceinline__ F3 h2f3(H3 v) { return { (float)v.x, (float)v.y, (float)v.z }; }

__device__ __forceinline__ float fdot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ F3 fcross(F3 a, F3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
__device__ __forceinline__ F3 fsub(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }

/* Per-ray traversal state kept in registers. */
struct RayRegs {
    F3 o, d;
    H3 rcpD;            /* f16vec3(1.0 / d), reference trace.comp:188 */
    float min_t, max_t;
};

__device__ __forceinline__ void ray_set_dir(RayRegs& r, F3 d)
{
    r.d = d;
    r.rcpD = { (h16)(1.0f / d.x), (h16)(1.0f / d.y), (h16)(1.0f / d.z) };
}

/* Box test on a pre-digested box (c, hA = f16(a-c), hB = f16(b-c)); reference
 * intersections.glsl:6-29.  Returns the part of the verdict that does not depend on max_t and
 * leaves tmin (as float) for the `max_t > tmin` part, which the caller evaluates when the node
 * is actually visited. */
__device__ __forceinline__ bool box_static(const RayRegs& r, float cx, float cy, float cz, h16 ax, h16 ay, h16 az, h16 bx, h16 by, h16 bz,
                                           float& ftmin)
{
    h16 ox = (h16)(r.o.x - cx), oy = (h16)(r.o.y - cy), oz = (h16)(r.o.z - cz);
    h16 t0 = (ax - ox) * r.rcpD.x, t1 = (bx - ox) * r.rcpD.x;
    h16 tmin = hmin16(t0, t1), tmax = hmax16(t0, t1);
    t0 = (ay - oy) * r.rcpD.y; t1 = (by - oy) * r.rcpD.y;
    tmin = hmax16(tmin, hmin16(t0, t1)); tmax = hmin16(tmax, hmax16(t0, t1));
    t0 = (az - oz) * r.rcpD.z; t1 = (bz - oz) * r.rcpD.z;
    tmin = hmax16(tmin, hmin16(t0, t1)); tmax = hmin16(tmax, hmax16(t0, t1));
    float ftmax = (float)tmax;
    ftmin = (float)tmin;
    return (ftmax > 0.0f || ftmin > 0.0f) && ftmax >= ftmin && r.min_t < ftmax;
}

/* Box test on a raw fp32 box as the reference node stores it (intersections.glsl:1-30). */
__device__ __forceinline__ bool box_raw(const RayRegs& r, const float4 na, const float4 nb)
{
    float cx = (na.x + nb.x) * 0.5f, cy = (na.y + nb.y) * 0.5f, cz = (na.z + nb.z) * 0.5f;
    float ftmin;
    bool s = box_static(r, cx, cy, cz, (h16)(na.x - cx), (h16)(na.y - cy), (h16)(na.z - cz), (h16)(nb.x - cx), (h16)(nb.y - cy), (h16)(nb.z - cz), ftmin);
    return s && r.max_t > ftmin;
}

/* Moeller-Trumbore as intersections.glsl:32-55 writes it: true divisions, left-to-right sums.
 * The acceptance test is the positive form of the reference's (same verdict for ordered
 * values; a NaN from a zero-area triangle rejects instead of being accepted). */
__device__ __forceinline__ bool tri_test(const RayRegs& r, F3 p1, F3 e1, F3 e2, float& t, float& alpha, float& beta, float& gamma)
{
    F3 s = fsub(r.o, p1), s1 = fcross(r.d, e2), s2 = fcross(s, e1);
    float det = fdot(s1, e1);
    t = fdot(s2, e2) / det;
    alpha = fdot(s1, s) / det;
    beta = fdot(s2, r.d) / det;
    gamma = 1.0f - alpha - beta;
    return t >= r.min_t && t <= r.max_t && alpha >= 0.0f && beta >= 0.0f && gamma >= 0.0f;
}

/* RNG and sampling, reference noise.glsl:53-77 and 95-109. */
struct Rng {
    uint32_t rnd;
    __device__ __forceinline__ uint32_t getRand() { rnd = rnd * 1103515245u + 12345u; return rnd & 0x3FFFFFFFu; }
    __device__ __forceinline__ float getRandF() { return (float)getRand() / (float)0x3FFFFFFF; }
};

__device__ __forceinline__ void WeylNth(uint32_t n, h16& a, h16& b)
{
    float fa = (float)(n * 12664745u) / 16777216.0f, fb = (float)(n * 9560333u) / 16777216.0f;
    fa = fa - __builtin_floorf(fa); fb = fb - __builtin_floorf(fb);
    a = (h16)fa; b = (h16)fb;
}

__device__ __forceinline__ H3 cosineHemisphere(h16 ix, h16 iy)
{
    h16 theta = ((h16)2.0f * (h16)3.1415926f) * iy;
    h16 sqrtPhi = hsqrt16(ix);
    h16 c = (h16)cosf((float)theta), s = (h16)sinf((float)theta);
    return { c * sqrtPhi, s * sqrtPhi, hsqrt16((h16)1.0f - ix) };
}

__device__ __forceinline__ H3 to_coord_space(H3 n, H3 r)
{
    H3 h = n;
    float ax = __builtin_fabsf((float)h.x), ay = __builtin_fabsf((float)h.y), az = __builtin_fabsf((float)h.z);
    if (ax <= ay && ax <= az) h.x = (h16)1.0f;
    else if (ay <= ax && ay <= az) h.y = (h16)1.0f;
    else h.z = (h16)1.0f;
    H3 y = hnormalize(hcross(h, n));
    H3 x = hnormalize(hcross(n, y));
    return hadd3(hadd3(hscale(r.x, x), hscale(r.y, y)), hscale(r.z, n));
}
