/*
 * ptcore.h — C ABI of the MI355X (gfx950) ray-tracing core.
 *
 * The library stands in for the Vulkan compute dispatch of JovianGraphics/PathTracer:
 * it keeps that renderer's buffer layouts (ShaderData.h / structures.glsl) and mirrors
 * its dispatch sites (PathTracer.cpp:776-838) one entry point each.  All signatures are
 * plain C: pointers, sizes and integer status codes; no exceptions cross this boundary.
 *
 * Every entry point cites the reference interface it replaces as file:line relative to
 * the reference checkout.
 *
 * Status codes: 0 = success, negative = failure; pt_last_error() gives the text.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Data contract.  Sizes and offsets are checked by static asserts at the end of this file.
 * ---------------------------------------------------------------------------------------- */

/* ShaderData.h:9-13, structures.glsl:7-11 — per-vertex shading attributes, 16 B. */
typedef struct PtVertexAux {
    float   normal[3];
    uint8_t color[4];            /* RGBA8; alpha < 128 marks the refractive ("delta") material */
} PtVertexAux;

/* ShaderData.h:15-28, structures.glsl:59-72 — the uniform block, 320 B, column-major mat4. */
typedef struct PtShaderConstants {
    float    viewMtx[16];
    float    projMtx[16];
    float    viewInvMtx[16];
    float    projInvMtx[16];
    float    viewportSize[2];
    float    viewportBase[2];    /* never written by the reference host (PathTracer.cpp:742-750): keep 0 */
    uint32_t numLights;
    uint32_t numTriangles;
    uint32_t frameIndex;
    uint32_t numRays;            /* ray-stack slots per pixel = maximum path depth */
    uint32_t numBVHNodes;
    uint32_t _pad0[3];
    float    ambientRadiance[3];
    uint32_t _pad1;
} PtShaderConstants;

/* ShaderData.h:30-34, structures.glsl:1-5 — point light, 32 B. */
typedef struct PtLight {
    float pos[3];      float _pad0;
    float radiance[3]; float _pad1;
} PtLight;

/* structures.glsl:43-52 — one ray-stack slot, 48 B.  (The host twin ShaderData.h:36-42 has a
 * different field order; only its sizeof is used, PathTracer.cpp:448.)  Half fields are raw
 * IEEE binary16 bit patterns. */
typedef struct PtRayStackSlot {
    float    rayDirection[3];
    uint32_t randState;
    float    rayOrigin[3];
    uint32_t currentDepth;
    uint16_t hitAlbedo[4];
    uint16_t wIn[3];
    uint16_t prob;
} PtRayStackSlot;

/* ShaderData.h:44-47, structures.glsl:54-57 — a job is the index of a live slot or PT_JOB_DEAD. */
typedef struct PtRayJob { uint32_t index; } PtRayJob;
#define PT_JOB_DEAD 0xFFFFFFFFu

/* BVH.h:27-33, structures.glsl:35-41 — threaded binary BVH node, 32 B.
 * right > 0: interior, index of right child (left child is own index + 1).
 * right <= 0: leaf, -right is the offset of its triangle in the (reordered) index array.
 * next: node to continue with when this node's box is missed; 0 ends the walk. */
typedef struct PtBVHNode {
    float   a[3]; int32_t next;
    float   b[3]; int32_t right;
} PtBVHNode;

/* structures.glsl:13-18 (the o/min_t/d/max_t prefix of Ray) — 32 B ray of the traversal microbench. */
typedef struct PtRay {
    float o[3]; float min_t;
    float d[3]; float max_t;
} PtRay;

/* Result of one traversal: t and the vertex-index triple of the hit triangle
 * (intersections.glsl:47-52); a miss has i1 = i2 = i3 = -1 and t = the ray's max_t. */
typedef struct PtHit {
    float   t;
    int32_t i1, i2, i3;
} PtHit;

/* Flattened two-wide node, 64 B: the boxes of both children of one interior node of the
 * reference tree, pre-digested exactly as intersections.glsl:3-5 does per visit
 * (c = (a+b)*0.5f in fp32, hA = f16(a-c), hB = f16(b-c)); nothing is re-quantised.
 * Child reference: bit 31 set = leaf, low 31 bits = index into the triangle records;
 * clear = index into the PtNode2 array. */
typedef struct PtNode2 {
    float    c[3][2];            /* box centres x, y, z, each stored as the pair (left child, right child) */
    uint16_t h[6][2];            /* f16(a-c).xyz then f16(b-c).xyz, each as the pair (left child, right child): one
                                    packed instruction then works on both children's boxes */
    uint32_t left, right;
    uint32_t refIndex;           /* index of this interior node in the reference array */
    uint16_t wmax[2];            /* (left child, right child), fp16: an upper bound of the largest magnitude among the
                                    box's six digested half extents plus 2^-12, which the certified walk's entry bound
                                    takes for every axis; +inf where the child's subtree holds a triangle so thin (sine
                                    of the angle between its edges e1, e2 below 2^-14) that the t its triangle test
                                    computes need not lie anywhere near the triangle: the bound is then -inf and the
                                    certified walk never skips such a subtree on distance */
} PtNode2;

/* The same two boxes in 48 B (PT_OPT_RECORD_LAYOUT = 1), three 16 B accesses per lane instead of four.
 * f16(b-c) is not stored: it is non-negative and its magnitude bits differ from those of
 * f16(a-c) by a few units at most (c is the rounded midpoint), so it is rebuilt exactly from hA and a 5-bit
 * difference.  pt_flatten_bvh_compact checks every box and refuses a tree with a box that does not fit. */
typedef struct PtNode2c {
    float    c[3][2];            /* as PtNode2 */
    uint16_t hA[3][2];           /* f16(a-c).xyz as (left child, right child) pairs */
    uint32_t left, right;        /* as PtNode2 */
    uint32_t delta;              /* axis k: left child in bits 5k..5k+4, right child in bits 16+5k..16+5k+4, each
                                    d + 15 with  bits(f16(b-c)) = (bits(f16(a-c)) & 0x7FFF) + d,  -15 <= d <= 15 */
} PtNode2c;

/* Flattened leaf in 48 B: PtTri without its last quarter.  The leaf's reference node index, needed only when a walk
 * runs off the end of the reference array (intersections.glsl:109-113), comes from a separate table. */
typedef struct PtTriC {
    float    p1[3];  int32_t i1;
    float    e1[3];  int32_t i2;
    float    e2[3];  int32_t i3;
} PtTriC;

/* Flattened leaf, 64 B: the triangle the reference reaches through two dependent texel
 * fetches (intersections.glsl:86-94), stored inline.  e1 = p2-p1 and e2 = p3-p1 are the
 * fp32 differences intersections.glsl:34-35 computes per test. */
typedef struct PtTri {
    float    p1[3];  int32_t i1;
    float    e1[3];  int32_t i2;
    float    e2[3];  int32_t i3;
    uint32_t refLeaf;            /* index of the leaf node in the reference array */
    uint32_t flags;              /* bit 0: last node of the reference array */
    uint32_t _pad[2];
} PtTri;
#define PT_LEAF_BIT 0x80000000u

/* Two-level treelet record, 128 B = one HBM line (PT_OPT_RECORD_LAYOUT = 2): the boxes of the two children L, R of
 * one interior node X of the reference tree AND of their children LL, LR, RL, RR, so that a walk pays one dependent
 * fetch per two levels of the tree and uses every byte of the line it fetched.  Every intermediate box is kept and
 * tested exactly as the reference tests it (intersections.glsl:1-30 on the digest c, f16(a-c), f16(b-c), the last
 * rebuilt from the 5-bit differences as in PtNode2c): nothing is re-quantised and no test is dropped.
 * Records and leaves (PtTri) share one buffer addressed in 64 B units, laid out depth first: X's record at unit u
 * (u even), then the subtrees of its up to four targets T0..T3 in that order, where
 *   L interior: T0 = LL, T1 = LR;  L a leaf: T0 = L's PtTri, no T1;   R interior: T2 = RL, T3 = RR;  R a leaf: T2 = R's PtTri.
 * An interior target is the treelet record of that node, a leaf target its PtTri.  T0 always starts at u + 2; the
 * starts of T1, T2, T3 are stored relative to u. */
typedef struct PtTreeletPair {
    float    c[3][2];            /* box centres x, y, z as (first, second) pairs */
    uint16_t hA[3][2];           /* f16(a-c).xyz as pairs */
    uint32_t delta;              /* as PtNode2c.delta (bits 0..14 first box, 16..30 second box); bits 15 and 31 are flags:
                                    pair[0]: L is a leaf / R is a leaf;  pair[1]: LL / LR is a leaf;  pair[2]: RL / RR is a leaf */
} PtTreeletPair;
typedef struct PtTreelet {
    PtTreeletPair pair[3];       /* (L, R), (LL, LR), (RL, RR); a pair whose parent is a leaf is zero */
    uint32_t link[2];            /* 64 bits: start of T1, T2, T3 minus u in bits 0..20, 21..41, 42..62, in 64 B units,
                                    or in steps of 32 units if bit 63 is set (subtrees beyond 2^21 units) */
} PtTreelet;
#define PT_PAIR_BIT 0x40000000u  /* traversal-stack flag of the six-box treelet walk, see k_trace_tree6 */

/* ------------------------------------------------------------------------------------------
 * Host-side scene preparation (no GPU needed).
 * ---------------------------------------------------------------------------------------- */

/* Replaces HimaliaPlyModel::LoadFile + BuildVertices + BuildIndices (PathTracer.cpp:246-262).
 * Reads ASCII and binary_little_endian PLY with x,y,z[,nx,ny,nz][,red,green,blue[,alpha]]
 * vertices and `list uchar uint|int vertex_indices` faces; polygons are fan-triangulated.
 * Call once with all output pointers NULL to get the counts, then again with buffers:
 * positions = 4 floats per vertex (w = 1), aux = PtVertexAux per vertex, indices = 3 per triangle. */
PT_API int pt_ply_load(const char* path, uint32_t* num_vertices, uint32_t* num_triangles,
                       float* positions, PtVertexAux* aux, uint32_t* indices,
                       char* err, size_t err_len);

/* Replaces BuildBVH (BVH.cpp:83-309).  positions: vec4 per vertex; indices: 3 per triangle,
 * reordered in place exactly as the reference does; nodes: room for 2*T-1 entries.
 * The node array is bit-identical to the reference's single-threaded build for any
 * num_threads (0 = all hardware threads).  max_depth may be NULL. */
PT_API int pt_build_bvh(const float* positions, uint32_t num_vertices,
                        uint32_t* indices, uint32_t num_triangles,
                        PtBVHNode* nodes, uint32_t* max_depth, int num_threads,
                        char* err, size_t err_len);

/* The flattening pass added to BVH.cpp: turns the threaded 32 B node array into PtNode2
 * (T-1 entries) and PtTri (T entries) records.  root_box receives the root's a,b (6 floats). */
PT_API int pt_flatten_bvh(const PtBVHNode* nodes, uint32_t num_nodes,
                          const float* positions, const uint32_t* indices, uint32_t num_triangles,
                          PtNode2* nodes2, PtTri* tris, float* root_box, uint32_t* max_stack,
                          char* err, size_t err_len);

/* The flattening pass with 48 B records.  leaf_ref[t] receives the reference node index of the leaf that holds
 * triangle t.  Returns 1 (and says so in err) if some box cannot be encoded; the caller then uses pt_flatten_bvh. */
PT_API int pt_flatten_bvh_compact(const PtBVHNode* nodes, uint32_t num_nodes,
                                  const float* positions, const uint32_t* indices, uint32_t num_triangles,
                                  PtNode2c* nodes2, PtTriC* tris, uint32_t* leaf_ref, float* root_box,
                                  uint32_t* max_stack, char* err, size_t err_len);

/* The flattening pass with treelet records.  units: room for capacity_units x 64 B, or NULL to only compute
 * *num_units (call twice).  max_stack receives the most entries a walk over this layout can have pending.
 * Returns 1 (and says so in err) if some box cannot be encoded; the caller then uses pt_flatten_bvh. */
PT_API int pt_flatten_bvh_treelets(const PtBVHNode* nodes, uint32_t num_nodes,
                                   const float* positions, const uint32_t* indices, uint32_t num_triangles,
                                   void* units, uint64_t capacity_units, uint64_t* num_units, float* root_box,
                                   uint32_t* max_stack, char* err, size_t err_len);

/* Replaces the constants block filled at PathTracer.cpp:731-750 (glm::lookAt, glm::perspective
 * with the y flip, glm::inverse) for the orbit camera of PathTracer.cpp:109-112. */
PT_API void pt_fill_constants(PtShaderConstants* c, uint32_t width, uint32_t height,
                              const float focus_center[3], float orbit_height, float orbit_radius,
                              float orbit_angle, uint32_t num_lights, uint32_t num_triangles,
                              uint32_t frame_index, uint32_t num_rays, uint32_t num_bvh_nodes,
                              const float ambient[3]);

/* ------------------------------------------------------------------------------------------
 * Device context.  One context per GPU per process; to the caller all work of a context is
 * ordered on one HIP stream (pt_stream), which replaces the Vulkan barriers of
 * PathTracer.cpp:782-818.  pt_render_samples runs frames side by side on further streams of
 * its own and joins them to that stream before it returns.  Every call makes the context's
 * device the calling thread's current HIP device.  A context is not thread-safe: calls on one
 * context must come from one thread at a time; contexts of different devices are independent.
 * ---------------------------------------------------------------------------------------- */
typedef struct PtContext PtContext;

PT_API int         pt_ctx_create(int device_id, PtContext** out);
PT_API void        pt_ctx_destroy(PtContext* ctx);
PT_API const char* pt_last_error(PtContext* ctx);      /* ctx may be NULL: last create error */
PT_API int         pt_synchronize(PtContext* ctx);
PT_API void*       pt_stream(PtContext* ctx);          /* the context's hipStream_t */

/* Options (no reference counterpart; they select among documented behaviours). */
enum {
    PT_OPT_TRAVERSAL      = 1,   /* 0 = reference 32 B nodes, stackless walk; 1 = PtNode2, reference visit
                                    order (results identical to 0); 2 = PtNode2, near child first: the fastest walk,
                                    1.15 times the speed of 3, whose closest hit differs from the reference's on about
                                    one ray in 10^5 (the reference's fp16 box test is not conservative, so its answer
                                    depends on its visit order);
                                    3 (default) = certified walk: closest-hit rays (batches of pt_trace_rays and the
                                    closest-hit passes of the path tracer) are walked near child first with skips that
                                    are safe in any order, and the few rays whose answer that does not prove equal to
                                    the reference's are walked again in reference order - results identical to 0 and 1
                                    (hits, and with them every buffer of a frame), on the one assumption that the t a
                                    triangle test computes lies where the triangle is (to 2^-7 relative): it fails only
                                    where that test's determinant is lost to rounding, for a ray lying in a triangle's
                                    plane to a few parts in 10^7.  Needs 64 B records and a tree whose boxes nest (every
                                    pt_build_bvh tree), else 3 is served by the reference-order walk; any-hit rays
                                    get the near-first walk, whose verdict no order changes */
    PT_OPT_STALE_SLOTS    = 2,   /* 0 = zero prob of the next slot when a path ends (default); 1 = leave
                                    deeper slots as they are, as composite.frag:35-49 finds them */
    PT_OPT_RESEED         = 3,   /* 0 = seed the LCG only after a clear (launch.comp:49-50, default);
                                    1 = reseed every frame (needed when frames are split over GPUs) */
    PT_OPT_PERSIST_BLOCKS = 4,   /* workgroups (256 lanes) per CU of the persistent traversal kernels; unset: 5 for a batch
                                    traced as given, 6 for an ordered batch (7 for its near-first walk over 64 B records, whose
                                    66 VGPRs let a SIMD hold seven wavefronts), 3 for the passes of the path tracer */
    PT_OPT_REFILL_MIN     = 5,   /* a wavefront refills its idle lanes once this many of its 64 are idle; unset: 36, 24
                                    (26 and, for any-hit rays, 32 where the launch is seven wide) and 12 for the same three uses */
    PT_OPT_CLAIM_CHUNK    = 6,   /* ray ids a wavefront takes from the global counter at a time: 0 = chosen from the
                                    ray count (64..128), else 64..1024 in steps of 64 */
    PT_OPT_RECORD_LAYOUT  = 8,   /* read by pt_scene_upload: 0 = 64 B records; 1 = 48 B records when every box
                                    can be encoded (pt_flatten_bvh_compact), else 64 B; 2 = 128 B two-level treelet
                                    records (PtTreelet) beside the 64 B records when every box can be encoded, else
                                    64 B only: traversals 1 and 2 and every any-hit walk then run on the treelets, the
                                    certified walk's first pass and traversal 0 on what they always use.  Results do
                                    not depend on the layout, bit for bit.  48 B records save a quarter of the record
                                    memory and measure about 5 % slower (one record in four lies across two lines) */
    PT_OPT_SORT_RAYS      = 9,   /* pt_trace_rays orders a batch by the cell of the scene box each origin lies in before
                                    tracing it (results are stored under the rays' own indices and do not change):
                                    0 = never, 1 = always, 2 (default) = batches of 2^20 rays or more */
    PT_OPT_SORT_BITS      = 10,  /* 1..4 (default 4): the ordering grid has 2^bits cells per axis */
    PT_OPT_SECOND_WALK    = 11,  /* how PT_OPT_TRAVERSAL = 3 walks the rays its first pass lists, in reference order: 0 = one
                                    ray per lane (the walk of traversal 1); 1 = one ray per wavefront, 64 nodes of the
                                    reference array fetched and box-tested at a time; 2 (default) = both launches are queued
                                    and the length of the list, which only the device knows, decides which of them works
                                    (a short list is all latency and goes one ray to a wavefront, a long one is throughput
                                    and goes one to a lane); in the frame train the short list is then walked inside the
                                    shade launch, beside the shading of the jobs that are not on it.  Same hits, bit for bit */
    PT_OPT_SHADOW_STREAM  = 12,  /* inside pt_render_sample[s], where a bounce's shadow pass is a launch of its own: 1 = on a
                                    second stream beside the closest-hit pass of the next bounce; 0 = on the frame's own
                                    stream, after the shade pass (half as many streams: for hosts short of hardware
                                    queues); 2 (default) = 1 where the passes cannot be merged (PT_OPT_MERGE_PASSES), else 0 */
    PT_OPT_MERGE_PASSES   = 13,  /* inside pt_render_sample[s]: 1 (default) = the shadow rays of bounce d are traced by the same
                                    launch as the closest-hit rays of bounce d + 1 (both wait for the shade pass of d only):
                                    half the launches, one stream per frame, and a frame rate that no longer hinges on the
                                    number of hardware queues; needs traversal 1, 2 or 3 on flattened records (the reference layout keeps the passes apart).
                                    Results do not depend on it */
    PT_OPT_FRAME_TRAIN    = 14,  /* pt_render_sample[s]: 0 = never, 1 = wherever it can serve, 2 (default) = where it can serve and
                                    numRays >= 3 or the scene's traversal records exceed 8 MiB (a small scene with one bounce
                                    or none runs better three frames in flight).  The frame train: the
                                    frames of a call advance together, one bounce per step:
                                    a step is one launch pass (the frame that enters), one traversal launch over the
                                    closest-hit rays of every frame in flight and the shadow rays of the step before, one
                                    resolve pass (the frame that leaves) and one shade pass, all on the context's stream;
                                    numRays + 1 ray stacks are kept in a ring (48 B x W x H x slots each) and a call of N
                                    frames takes N + numRays steps.  Serves traversals 1, 2 and 3 on every record layout
                                    (3 walks 48 B records in reference order); traversal 0, the stale-slot option and the
                                    voting build run one frame per pipe (PT_OPT_FRAMES_IN_FLIGHT), as does 0 here.  Every
                                    buffer comes out the same byte for byte */
    PT_OPT_TRAIN_PARTS    = 15,  /* 1..4 (default 2): the frame train renders the image in this many bands of rows, each a train
                                    of its own on its own stream (a pixel's path never leaves its band): while one band's
                                    shade, launch and resolve passes run, another band's traversal launch has the GPU */
    PT_OPT_TRAIN_MAX_MB   = 16,  /* 0 (default) = no limit; else the most device memory, in MiB, the frame train may take for its
                                    ring of ray stacks and its lists (numRays + 1 stacks: 10 GiB at 1080p x 8 slots, 38 GiB
                                    at 4K).  A call whose train would need more - or more than the device has free - renders
                                    frame by frame instead, with the same results */
    PT_OPT_TRAIN_KEEP     = 17,  /* 0 (default): a pt_render_sample[s] call returns with all its frames finished (a call of N
                                    frames takes N + numRays steps of the frame train).  1: the train is kept between calls -
                                    a call of N frames takes N steps and returns with up to numRays frames still in the
                                    train, which the next call's steps carry further: a host that renders one frame per call
                                    (PathTracer.cpp:672, one f_onRender per presented image) runs at the rate of one long
                                    call.  pt_readback_accumulation, pt_readback_current, pt_dump_current, pt_tonemap and
                                    pt_synchronize then see the frames finished so far (.w of the accumulation counts them):
                                    the picture lags numRays frames behind the calls.  pt_render_flush finishes the frames
                                    in flight, and so does every other call first (options, constants other than
                                    frameIndex, clear, resize, scene upload, the dispatch sites, ray-stack reads, batches,
                                    the all-reduce, a render call that asks for ray counts), after which every buffer is
                                    what frame-by-frame rendering leaves, byte for byte.  While frames are in flight, work
                                    the caller queues on pt_stream is ordered behind the calls above only */
    PT_OPT_TIME_TRAVERSAL = 18,  /* measurement: 1 = the frame train brackets every traversal launch (one per band of rows and step,
                                    with the certified walk's second pass inside the bracket) with a HIP event pair on the launch's
                                    own stream; pt_traversal_ms reads and resets the sums.  At most 4096 launches per reading; the
                                    events cost a few microseconds per launch, so timed frame rates are taken with it off */
    PT_OPT_FRAMES_IN_FLIGHT = 7  /* 1..8 (default 3): frames the per-frame pipeline keeps in flight (PT_OPT_FRAME_TRAIN = 0, and what
                                    the frame train does not serve), each on its own ray stack and stream; results do not
                                    depend on it.  With merged passes three frames are three streams and HIP's default of
                                    four hardware queues per process is enough.  Where the passes stay apart (the reference
                                    layout, traversal 3 where it falls back to the reference-order walk) every frame in flight has a second stream; with four queues, six
                                    streams share queues and wait on each other (about a fifth fewer frames per second at
                                    1080p).  A host that renders that way should export GPU_MAX_HW_QUEUES=8 (2 per frame in
                                    flight plus 2; never above 32) before its first HIP call - the library cannot set it
                                    once the runtime is up - or set PT_OPT_SHADOW_STREAM = 0.  The frame train uses two
                                    streams whatever the options */
};
PT_API int pt_set_option(PtContext* ctx, int option, int value);
PT_API int pt_get_option(PtContext* ctx, int option, int* value);

/* Replaces the six CreateBuffer + UploadToBufferEx calls of PathTracer.cpp:272-321 (bindings
 * 1,2,3,4,7,8 of PathTracer.cpp:490-502).  Host pointers are borrowed for the call only.
 * `nodes` is the reference-layout array; the flattening pass runs inside and first checks that the
 * array is a pre-order tree with consistent skip links and one leaf per triangle (a malformed array is
 * refused with -1 and a reason: a walk over it would not end), and that every vertex index is in range.
 * The traversal records of a scene live in one buffer addressed with unsigned 32-bit byte offsets: 64 B x (2T - 1)
 * must stay below 4 GiB, that is T <= 33.5 million triangles (-1 beyond). */
PT_API int pt_scene_upload(PtContext* ctx,
                           const float* positions, uint32_t num_vertices,
                           const PtVertexAux* aux,
                           const uint32_t* indices, uint32_t num_triangles,
                           const PtBVHNode* nodes, uint32_t num_nodes,
                           const PtLight* lights, uint32_t num_lights,
                           const uint16_t* blue_noise /* 65536 entries */);

/* Replaces the per-resolution resources of PathTracer.cpp:378-459 (bindings 5,6,9,10):
 * current image RGBA16F, accumulation RGBA32F, ray stack 48 B x W x H x max_slots, jobs.
 * Constants set earlier for another size (or more slots) are dropped: call pt_set_constants again. */
PT_API int pt_resize(PtContext* ctx, uint32_t width, uint32_t height, uint32_t max_slots);

/* Replaces the transient uniform slice written at PathTracer.cpp:731-752 (binding 0). */
PT_API int pt_set_constants(PtContext* ctx, const PtShaderConstants* constants);

/* Replaces ClearImage(accumulation) at PathTracer.cpp:947-950. */
PT_API int pt_clear_accumulation(PtContext* ctx);

/* Dispatch sites, in the order PathTracer.cpp records them. */
PT_API int pt_launch(PtContext* ctx);                               /* PathTracer.cpp:776-778, launch.comp */
PT_API int pt_trace(PtContext* ctx, uint32_t depth, int speculative);/* PathTracer.cpp:788-793, trace.comp */
PT_API int pt_compact(PtContext* ctx);                              /* PathTracer.cpp:795-811, raysort.comp */
PT_API int pt_resolve_accumulate(PtContext* ctx);                   /* PathTracer.cpp:835-836, composite.frag:30-55 */

/* One whole frame = one sample per pixel: set frameIndex, launch, numRays x {trace, compact},
 * resolve + accumulate (PathTracer.cpp:729-838).  ray_counts, if not NULL, receives 2*numRays
 * uint64: closest-hit rays and shadow rays traced at each depth. */
PT_API int pt_render_sample(PtContext* ctx, uint32_t frame_index, int speculative_depth0,
                            uint64_t* ray_counts);
/* num_frames consecutive frames (frame_index = first, first+1, ...) with the constants last set.  The current
 * image, the accumulation and every ray-stack slot the last frame's paths reached are what num_frames calls of
 * pt_render_sample produce, byte for byte (slots beyond a path's end hold leftovers of an earlier frame, and
 * which one depends on the number of frames in flight): a frame's launch pass takes each pixel's RNG state from the frame before it (launch.comp:49-50)
 * and the accumulation adds run in frame order, but up to PT_OPT_FRAMES_IN_FLIGHT frames overlap on the GPU.
 * With PT_OPT_STALE_SLOTS set the frames run one at a time.  ray_counts receives the sums over the frames. */
PT_API int pt_render_samples(PtContext* ctx, uint32_t first_frame_index, uint32_t num_frames,
                             int speculative_depth0, uint64_t* ray_counts);
/* Finishes the frames a kept frame train still holds (PT_OPT_TRAIN_KEEP = 1): numRays steps at most, queued like a render
 * call.  Nothing to do otherwise.  The reference has no counterpart: its frames are finished when f_onRender returns. */
PT_API int pt_render_flush(PtContext* ctx);

/* Replaces the dump read-back of PathTracer.cpp:848-854 (RGBA16F current image) and adds the
 * linear accumulation (RGBA32F, .w = sample count).  Host destination buffers. */
PT_API int pt_readback_current(PtContext* ctx, uint16_t* rgba16f);
PT_API int pt_readback_accumulation(PtContext* ctx, float* rgba32f);
PT_API int pt_readback_raystack(PtContext* ctx, PtRayStackSlot* slots, PtRayJob* jobs);
PT_API int pt_upload_raystack(PtContext* ctx, const PtRayStackSlot* slots, const PtRayJob* jobs);
PT_API void* pt_accumulation_device_ptr(PtContext* ctx);           /* 4 floats x W x H, for the all-reduce */

/* Frame-dump interchange (PathTracer.cpp:180-202, 840-861): <dir>/RayStackBuffer_<index>.bin.z is one
 * zlib stream of the W x H RGBA16F current image, <same>.txt holds "W,H".  pt_dump_current reads the
 * current image back and writes the pair; pt_frame_dump_write / _read are the host-only halves
 * (read with rgba16f == NULL only returns the size).  Note that the reference fills file <index> from the
 * staging buffer it copied to a swapchain cycle earlier (:849 runs before :850 records the copy), so its
 * file f holds the image of frame f - framesInFlight; here <index> is whatever the caller passes.
 * Host-only calls return -1 bad argument, -4 file I/O, -5 zlib, -6 malformed size file or stream length,
 * -7 destination too small, -8 out of host memory. */
PT_API int pt_dump_current(PtContext* ctx, const char* dir, uint32_t index);
PT_API int pt_frame_dump_write(const char* dir, uint32_t index, uint32_t width, uint32_t height,
                               const uint16_t* rgba16f);
PT_API int pt_frame_dump_read(const char* dir, uint32_t index, uint32_t* width, uint32_t* height,
                              uint16_t* rgba16f, uint64_t capacity_texels);

/* Display colour of composite.frag:18-26,57, pow(ACESFilm(acc.rgb / acc.w), 1/2.2), computed on the GPU
 * from the accumulation and copied to HOST buffers: rgb32f = 3 floats per pixel (the shader's outColor),
 * rgba8 = 4 bytes per pixel (UNORM, round to nearest, alpha 255).  Either may be NULL.
 * pt_write_ppm stores an rgba8 image as binary PPM (host only, codes as above). */
PT_API int pt_tonemap(PtContext* ctx, float* rgb32f, uint8_t* rgba8);
PT_API int pt_write_ppm(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8);

/* Traversal microbench (no reference dispatch site; it calls traceRay, intersections.glsl:57-115,
 * on caller-supplied rays).  d_rays / d_hits are DEVICE pointers; stats, if not NULL, is a host
 * array of 4 uint64 that receives {box tests, triangle tests, node records fetched in 64 B units, ray walks} (a
 * counting build of the same kernel runs instead of the plain one).  All four count every walk: under
 * PT_OPT_TRAVERSAL = 3 a ray listed for the second, reference-order walk is walked twice and counts twice, so
 * stats[3] - num_rays is the number of rays that took the second walk. */
PT_API int pt_trace_rays(PtContext* ctx, const PtRay* d_rays, uint64_t num_rays, int any_hit,
                         PtHit* d_hits, uint64_t* stats);
/* pt_trace_rays for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc).
 * No counterpart in the reference.  Same rays in, same hits out, byte for byte, but
 * the call allocates and frees nothing: the ray-id counters, the ordering pass's bins, keys and order, the certified
 * walk's list and the spill area of the traversal stacks all live in one static device array of 52 MiB that every GPU
 * the library runs on holds from the first pt_ctx_create on it until the process ends (part of the library's code object:
 * a process pays for it on every GPU it creates a context on, whether or not it ever calls this).  Limits: at most 2^21
 * (2 097 152) rays per call (-1 beyond that), and a tree whose stack need beyond the 11 entries every walk keeps in LDS fits the
 * array's 32 MiB spill area with at least one workgroup per CU (-1 otherwise; a deep tree runs with fewer workgroups
 * than pt_trace_rays would use).  A walk over 64 B or 48 B records needs one stack entry per level of the tree, a walk
 * over treelet records three per two levels: on a 256-CU part that is a depth of 75, or of 50 with
 * PT_OPT_RECORD_LAYOUT = 2.  All calls on one GPU share that GPU's array, from whatever context.  Calls from one
 * process take turns on it: every call makes its stream wait for an event recorded behind the previous call's last
 * launch on that GPU (the event is made at pt_ctx_create), so contexts on different streams may call back to back
 * without a host synchronisation.  Calls from different PROCESSES do not share the array (each process has its own
 * copy).  The hand-over uses stream waits and an event record, which a stream under HIP graph capture does not take
 * from outside the capture: the call is not for use under capture.  -2 if the array's address could not be resolved
 * at pt_ctx_create (the reason is in pt_last_error). */
PT_API int pt_trace_rays_noalloc(PtContext* ctx, const PtRay* d_rays, uint64_t num_rays, int any_hit, PtHit* d_hits);
/* Same with HOST pointers: copies in, traces, copies out. */
PT_API int pt_trace_rays_host(PtContext* ctx, const PtRay* rays, uint64_t num_rays, int any_hit,
                              PtHit* hits, uint64_t* stats);
/* Fills d_rays[0..n) with the synthetic incoherent ray set: ray i takes five 24-bit uniforms from
 * successive splitmix64 outputs of state seed + (first+i)*0x9E3779B97F4A7C15; origin uniform in the
 * scene's root box, direction uniform on the sphere, min_t = 0.001, max_t = 1e5. */
PT_API int pt_generate_rays(PtContext* ctx, uint64_t seed, uint64_t first, uint64_t n, PtRay* d_rays);

PT_API void* pt_device_alloc(PtContext* ctx, size_t bytes);
PT_API int   pt_device_free(PtContext* ctx, void* p);
PT_API int   pt_copy_to_host(PtContext* ctx, void* dst, const void* d_src, size_t bytes);
PT_API int   pt_copy_to_device(PtContext* ctx, void* d_dst, const void* src, size_t bytes);


/* Self-test with no counterpart in the reference: checks on the device, over num_cases random boxes and rays, that the
 * entry-distance bound the certified walk (PT_OPT_TRAVERSAL = 3) skips subtrees by never exceeds the true entry distance
 * (fp64) into the box the digest was made from.  *checked = cases in which the ray reaches the box, *violations must be 0. */
PT_API int pt_selftest_entry_bound(PtContext* ctx, uint64_t seed, uint64_t num_cases, uint64_t* checked, uint64_t* violations);
/* Time of the last traversal launch in milliseconds, measured with HIP events on the
 * context's stream (closest-hit launch of pt_trace_rays, or the sum over a pt_render_sample). */
PT_API float pt_last_kernel_ms(PtContext* ctx);
/* With PT_OPT_TIME_TRAVERSAL set: the durations, in milliseconds, of the frame train's traversal launches since the last
 * reading, in the order they were queued (step by step, within a step band by band), as many as fit `capacity`; *launches
 * receives their number, and the record starts over.  Launches of different bands overlap in time.  Finishes the frames in
 * flight and waits for the device first.  No reference counterpart (the reference has no GPU timing at all). */
PT_API int pt_traversal_ms(PtContext* ctx, float* per_launch_ms, uint32_t capacity, uint32_t* launches);

/* Multi-GPU (no reference counterpart): sums the accumulation buffers of all ranks in place.
 * unique_id is the 128-byte ncclUniqueId made by pt_comm_unique_id on rank 0 and handed to
 * every rank by the caller. */
PT_API int pt_comm_unique_id(uint8_t unique_id[128]);
PT_API int pt_comm_init(PtContext* ctx, int rank, int num_ranks, const uint8_t unique_id[128]);
PT_API int pt_allreduce_accumulation(PtContext* ctx);

PT_API const char* pt_version(void);

#ifdef __cplusplus
}
#endif

/* Layout checks (C11 / C++11). */
#ifdef __cplusplus
#define PT_SA(c, m) static_assert(c, m)
#else
#define PT_SA(c, m) _Static_assert(c, m)
#endif
PT_SA(sizeof(PtVertexAux) == 16, "VertexAux is 16 B");
PT_SA(offsetof(PtVertexAux, color) == 12, "VertexAux.color at 12");
PT_SA(sizeof(PtShaderConstants) == 320, "ShaderConstants is 320 B");
PT_SA(offsetof(PtShaderConstants, viewportSize) == 256, "viewportSize at 256");
PT_SA(offsetof(PtShaderConstants, viewportBase) == 264, "viewportBase at 264");
PT_SA(offsetof(PtShaderConstants, numLights) == 272, "numLights at 272");
PT_SA(offsetof(PtShaderConstants, frameIndex) == 280, "frameIndex at 280");
PT_SA(offsetof(PtShaderConstants, numRays) == 284, "numRays at 284");
PT_SA(offsetof(PtShaderConstants, numBVHNodes) == 288, "numBVHNodes at 288");
PT_SA(offsetof(PtShaderConstants, ambientRadiance) == 304, "ambientRadiance at 304");
PT_SA(sizeof(PtLight) == 32 && offsetof(PtLight, radiance) == 16, "Light is 32 B");
PT_SA(sizeof(PtRayStackSlot) == 48, "ray-stack slot is 48 B");
PT_SA(offsetof(PtRayStackSlot, randState) == 12 && offsetof(PtRayStackSlot, rayOrigin) == 16 &&
      offsetof(PtRayStackSlot, currentDepth) == 28 && offsetof(PtRayStackSlot, hitAlbedo) == 32 &&
      offsetof(PtRayStackSlot, wIn) == 40 && offsetof(PtRayStackSlot, prob) == 46, "slot field offsets");
PT_SA(sizeof(PtRayJob) == 4, "job is 4 B");
PT_SA(sizeof(PtBVHNode) == 32 && offsetof(PtBVHNode, next) == 12 && offsetof(PtBVHNode, b) == 16 &&
      offsetof(PtBVHNode, right) == 28, "BVHNode is 32 B");
PT_SA(sizeof(PtRay) == 32 && sizeof(PtHit) == 16, "microbench ray 32 B, hit 16 B");
PT_SA(sizeof(PtNode2) == 64 && sizeof(PtTri) == 64, "flattened records are 64 B");
PT_SA(sizeof(PtNode2c) == 48 && sizeof(PtTriC) == 48, "compact flattened records are 48 B");
PT_SA(sizeof(PtTreeletPair) == 40 && sizeof(PtTreelet) == 128 && offsetof(PtTreelet, link) == 120, "treelet record is 128 B");

#endif /* PTCORE_H */
