/*
 * ptcore.hip — the device context behind include/ptcore.h: owns every buffer the reference binds
 * to its descriptor set (reference PathTracer.cpp:490-502), launches the kernels of
 * trace_kernels.h / frame_kernels.h / sort_kernels.h, and reports failures as status codes.
 * Every call is ordered on the context's stream as far as the caller can tell; inside
 * pt_render_samples several frames run side by side on streams of their own and are joined
 * to that stream before the call returns.  A context is not thread-safe: one thread at a time.
 * There is no CPU path: without a gfx950 device pt_ctx_create fails.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "frame_kernels.h"
#include "sort_kernels.h"
#include "trace_kernels.h"

namespace {

std::string g_createError;

struct NcclId { char internal[128]; };
struct NcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
NcclApi g_nccl;
std::mutex g_ncclMutex;

bool load_nccl(std::string& err)
{
    std::lock_guard<std::mutex> lk(g_ncclMutex);
    if (g_nccl.lib) return true;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
    g_nccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_nccl.CommInitRank = (int (*)(void**, int, NcclId, int))dlsym(h, "ncclCommInitRank");
    g_nccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_nccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_nccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_nccl.GetUniqueId || !g_nccl.CommInitRank || !g_nccl.AllReduce || !g_nccl.CommDestroy) { err = "RCCL library lacks a required symbol"; return false; }
    g_nccl.lib = h;
    return true;
}

}  // namespace

/* ---- scratch of pt_trace_rays_noalloc, the entry point that allocates nothing -------------------------------------
 * Everything pt_trace_rays keeps in buffers of its own (allocated on first use, grown on demand) has a fixed place in
 * one file-scope device array: the ray-id counter, the certified walk's two counters, the ordering pass's bin counts
 * and cursors, its keys and the order it produces, the certified walk's list of rays to walk again, and the spill area
 * of the per-lane traversal stacks.  Every GPU the library runs on holds its own copy of the array, at its own address,
 * for the life of the process.  In host code the array's name is not a device address: the address is asked for once
 * per device (pt_ctx_create, with that device current) and kept in g_scratchPtr; kernels, memsets and copies are given
 * that pointer only. */
enum : size_t {
    NA_MAX_RAYS    = (size_t)1 << 21,                          /* rays per call */
    NA_OFF_CTR     = 0,                                        /* u32[64]: [0] ray-id counter, [1] listed rays, [2] second walk's ray-id counter */
    NA_OFF_BINS    = 256,                                      /* u32[2 * PT_SORT_BINS]: counts, cursors */
    NA_OFF_ORDER   = NA_OFF_BINS + 2 * PT_SORT_BINS * 4,       /* u32[NA_MAX_RAYS] */
    NA_OFF_KEYS    = NA_OFF_ORDER + NA_MAX_RAYS * 4,           /* u16[NA_MAX_RAYS] */
    NA_OFF_FLAGS   = NA_OFF_KEYS + NA_MAX_RAYS * 2,            /* u32[NA_MAX_RAYS] */
    NA_OFF_SPILL   = NA_OFF_FLAGS + NA_MAX_RAYS * 4,
    NA_SPILL_BYTES = (size_t)32 << 20,                         /* uint2[entries beyond the LDS part][lanes of the launch] */
    NA_TOTAL       = NA_OFF_SPILL + NA_SPILL_BYTES             /* 52.03 MiB */
};
static_assert(NA_OFF_BINS % 256 == 0 && NA_OFF_ORDER % 256 == 0 && NA_OFF_KEYS % 256 == 0 && NA_OFF_FLAGS % 256 == 0 && NA_OFF_SPILL % 256 == 0,
              "every part of the scratch array starts on a 256 B boundary");
static __device__ __align__(256) unsigned char scratch[NA_TOTAL];

namespace {
enum { PT_MAX_DEVICES = 64 };
unsigned char* g_scratchPtr[PT_MAX_DEVICES] = { nullptr };     /* device address of `scratch` on each device, once resolved */
/* Recorded behind the last launch of every pt_trace_rays_noalloc call on the device, waited for by the next call's stream
 * before it touches the array: calls from different contexts (different streams) of one process then take turns on the
 * device's one array instead of overwriting each other's counters, bins, order and lists.  Made with the address, at
 * pt_ctx_create, so that the call itself creates nothing. */
hipEvent_t g_scratchEvent[PT_MAX_DEVICES] = { nullptr };
std::mutex g_scratchMutex;

/* Device address of this device's copy of the scratch array; the device must be current.  nullptr + message on failure. */
unsigned char* resolve_scratch(int device, std::string& err)
{
    if (device < 0 || device >= PT_MAX_DEVICES) { err = "device id beyond the scratch table"; return nullptr; }
    std::lock_guard<std::mutex> lk(g_scratchMutex);
    if (g_scratchPtr[device]) return g_scratchPtr[device];
    unsigned char* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) { err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); return nullptr; }
    if (!g_scratchEvent[device] && (e = hipEventCreateWithFlags(&g_scratchEvent[device], hipEventDisableTiming)) != hipSuccess) {
        g_scratchEvent[device] = nullptr;
        err = std::string("hipEventCreate for the scratch array: ") + hipGetErrorString(e);
        return nullptr;
    }
    g_scratchPtr[device] = ptr;
    return ptr;
}
}  // namespace

enum { PT_MAX_PIPES = 8 };

/* Everything one frame in flight owns: its ray stack, job lists, hit records, shadow queues, counters, and the two
 * streams its passes run on.  Pipe 0 serves the dispatch-site API and pt_render_sample; pt_render_samples deals
 * frames round-robin to several pipes so that the short launches of one frame's deep bounces run beside the large
 * launches of another's first bounces. */
struct FramePipe {
    int index = 0;
    bool ready = false;
    hipStream_t stream = nullptr, stream2 = nullptr;    /* stream2: shadow rays of depth d beside the closest-hit pass of depth d+1 */
    hipEvent_t evShade[2] = { nullptr, nullptr }, evShadow[2] = { nullptr, nullptr };
    hipEvent_t evShade0 = nullptr;                        /* slot 0 of the stack holds the frame's final RNG states */
    hipEvent_t evDone = nullptr;                          /* the frame's resolve pass has run */
    bool shadowBusy[2] = { false, false }, shade0Pending = false, donePending = false;
    PtRayStackSlot* dStack = nullptr; uint32_t* dJobs[2] = { nullptr, nullptr }; int curJobs = 0;
    PtHitRec* dHitRecs = nullptr; PtShadowRay* dShadow[2] = { nullptr, nullptr };
    uint32_t* dCtr = nullptr; unsigned long long* dRayCounts = nullptr;
    uint32_t* dFlag = nullptr;                            /* certified walk: jobs listed for the second walk (allocated on first use) */
    int deferredShadow = -1;                              /* bounce whose shadow pass waits to run inside the next bounce's closest-hit launch */
};

/* The frame train: pt_render_samples steps all its frames in flight together, each one bounce further per step, on one
 * stream.  A step is one launch pass (the frame that enters), ONE traversal launch over the closest-hit rays of every frame in
 * flight and the shadow rays the last step's shade pass queued, one resolve pass (the frame that leaves) and ONE shade pass
 * over all jobs.  Frame i enters at step i, is at depth d in step i + d and leaves in step i + numRays, so with numRays + 1
 * frames in the ring a frame is finished every step, and the launches are as large as the chip likes them (ten million rays
 * at 1080p x 8 bounces) instead of eight shrinking ones per frame that need several frames on several streams to fill the
 * GPU.  Jobs carry ring-wide slot indices, so the traversal and shade kernels need not know which frame a job belongs to;
 * the order in which a pixel's path is advanced is the same, so every buffer comes out byte for byte as frame-by-frame
 * rendering leaves it. */
enum { PT_MAX_TRAIN_PARTS = 4 };
/* One part of the train: a band of rows of the image with its own job lists, hit records, shadow queues and counters, on
 * its own stream.  A pixel's path and RNG chain never leave its part, so the parts are independent of each other; while
 * one part's shade, launch and resolve passes run, or its traversal launch thins out towards its end, another part's
 * traversal launch fills the GPU. */
struct TrainPart {
    uint32_t row0 = 0, row1 = 0;
    uint64_t cap = 0;                 /* entries of each job list, of the hit records, of each shadow queue */
    uint32_t* dJobs[2] = { nullptr, nullptr }; PtHitRec* dHitRecs = nullptr; PtShadowRay* dShadow[2] = { nullptr, nullptr };
    uint32_t* dCtr = nullptr; uint32_t* dFlag = nullptr; uint32_t* dLate = nullptr;
    hipStream_t stream = nullptr;     /* part 0: the context's stream */
    hipEvent_t evDone = nullptr, evStagger = nullptr;
};
struct FrameTrain {
    bool ready = false;
    uint32_t F = 0;                   /* ray stacks in the ring */
    int parts = 0;
    PtRayStackSlot* dRing = nullptr;  /* F stacks of W x H x slots slots; every part works on its own rows of each */
    unsigned long long* dRayCounts = nullptr;
    TrainPart part[PT_MAX_TRAIN_PARTS];
    uint32_t head = 0;                /* ring position of the most recent frame */
    hipEvent_t evFork = nullptr;
    /* the train between steps */
    bool live = false;                /* the parts' counters and lists are those of a train under way (frames may be in flight) */
    bool forked = false;              /* the parts' streams hold work the context's stream has not waited for */
    bool stagger = false;             /* the next step starts the parts one launch apart */
    int cur = 0, q = 0;               /* which job list the next step reads, which shadow queue it traces */
    uint64_t stepNo = 0;
    std::deque<std::pair<uint32_t, uint64_t>> flight;     /* ring position and entering step of every frame not yet resolved, oldest first */
};

struct PtContext {
    int device = 0;
    hipStream_t stream = nullptr;                         /* = pipe[0].stream: the stream the API's ordering promises refer to */
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evFork = nullptr;
    std::string err;
    int numCUs = 0;

    /* scene (bindings 1,2,3,4,7,8) */
    float4* dPositions = nullptr; PtVertexAux* dAux = nullptr; uint32_t* dIndices = nullptr; PtBVHNode* dNodes = nullptr;
    PtLight* dLights = nullptr; uint16_t* dBlueNoise = nullptr;
    uint4* dRecords = nullptr; uint2* dOverflow = nullptr; uint32_t leafBase = 0, recordBytes = 0;
    bool compact = false; uint32_t* dLeafRef = nullptr; uint32_t lastLeaf = 0;      /* 48 B records */
    uint4* dTreelets = nullptr; uint32_t treeletBytes = 0, treeletMaxStack = 0;      /* 128 B treelet records + leaves, beside the 64 B records */
    uint32_t V = 0, T = 0, N = 0, L = 0, maxStack = 0, overflowEntries = 0;
    uint64_t overflowLanes = 0; int overflowAreas = 0;
    float rootBox[6] = {0}; float rootC[3] = {0}; uint16_t rootH[6] = {0};
    float lastC[3] = {0}; uint16_t lastH[6] = {0};        /* box digest of the reference array's last node */

    /* per-resolution (bindings 5,6,9,10) */
    uint32_t W = 0, H = 0, slots = 0;
    FramePipe pipe[PT_MAX_PIPES];
    FrameTrain train;
    bool rngInTrain = false, lastInTrain = false;     /* the RNG states / the most recent frame live in the train's ring, at its head */
    int rngPipe = 0;            /* the pipe whose stack holds the RNG states the next launch continues from (launch.comp:49-50) */
    int lastPipe = 0;           /* the pipe of the most recent launch: what pt_readback_raystack returns */
    int resolvePipe = 0;        /* the pipe of the most recent resolve pass: accumulation adds stay in frame order */
    bool accumEmpty = true;     /* no sample in the accumulation since pt_resize / pt_clear_accumulation */
    float4* dAccum = nullptr; uint2* dCurrent = nullptr;
    uint32_t* dFetch = nullptr; unsigned long long* dStats = nullptr;      /* pt_trace_rays: id counter, visit statistics */
    uint32_t* dFlagList = nullptr; uint32_t* dFlagCtr = nullptr; uint64_t flagCapacity = 0;     /* certified walk: rays to trace again, {count, fetch counter} */
    bool boxesNest = false;     /* every child box inside its parent's, every triangle inside its leaf's: what the certified walk rests on */
    uint32_t* dSortOrder = nullptr; uint16_t* dSortKeys = nullptr; uint32_t* dSortBins = nullptr;   /* ray ordering: scratch for sortCapacity rays, [0..4096) counts, [4096..8192) cursors */
    uint64_t sortCapacity = 0;

    PtShaderConstants consts; bool haveConsts = false;
    int optTraversal = 2, optStale = 0, optReseed = 0, optBlocksPerCU = 5, optRefillMin = 36, optClaimChunk = 0, optFramesInFlight = 3, optRecordLayout = 0, optSortRays = 2, optSortBits = 4, optSecondWalk = 1, optShadowStream = 2, optMergePasses = 1, optFrameTrain = 2, optTrainParts = 2, optTrainMaxMB = 0, optTrainKeep = 0;
    bool userBlocks = false, userRefill = false;      /* PT_OPT_PERSIST_BLOCKS / PT_OPT_REFILL_MIN were set by the caller */
    float lastMs = 0.0f; bool evPending = false;
    void* comm = nullptr;
    unsigned char* scratchPtr = nullptr;    /* this device's copy of the scratch array (pt_trace_rays_noalloc) */
    std::string scratchErr;                 /* why scratchPtr is null, if it is */

    int fail(int code, const std::string& m) { err = m; return code; }
};

#define HIPCHK(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return ctx->fail(-10, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

static int settle(PtContext* ctx);         /* finishes the frames a kept train still holds */
static int train_flush(PtContext* ctx);
static int train_join(PtContext* ctx);

namespace {

template <class T> void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

void free_scene(PtContext* c)
{
    dfree(c->dPositions); dfree(c->dAux); dfree(c->dIndices); dfree(c->dNodes); dfree(c->dLights); dfree(c->dBlueNoise);
    dfree(c->dRecords); dfree(c->dOverflow); dfree(c->dLeafRef); dfree(c->dTreelets);
    c->treeletBytes = 0; c->treeletMaxStack = 0;
    c->V = c->T = c->N = c->L = 0;
}
void free_pipe_buffers(FramePipe& p)
{
    dfree(p.dStack); dfree(p.dJobs[0]); dfree(p.dJobs[1]); dfree(p.dHitRecs); dfree(p.dShadow[0]); dfree(p.dShadow[1]);
    dfree(p.dCtr); dfree(p.dRayCounts); dfree(p.dFlag);
    p.ready = false; p.curJobs = 0;
    p.shadowBusy[0] = p.shadowBusy[1] = p.shade0Pending = p.donePending = false;
}
void free_train_buffers(FrameTrain& t)
{
    dfree(t.dRing); dfree(t.dRayCounts);
    for (TrainPart& p : t.part) {
        dfree(p.dJobs[0]); dfree(p.dJobs[1]); dfree(p.dHitRecs); dfree(p.dShadow[0]); dfree(p.dShadow[1]); dfree(p.dCtr); dfree(p.dFlag); dfree(p.dLate);
        p.cap = 0;
    }
    t.ready = false; t.F = 0; t.parts = 0; t.head = 0;
    t.live = t.forked = t.stagger = false; t.flight.clear();
}
void free_train(PtContext* c)        /* streams and events stay */
{
    free_train_buffers(c->train);
    c->rngInTrain = c->lastInTrain = false;
}
void destroy_train(PtContext* c)
{
    free_train(c);
    FrameTrain& t = c->train;
    for (TrainPart& p : t.part) {
        if (p.evDone) (void)hipEventDestroy(p.evDone);
        if (p.evStagger) (void)hipEventDestroy(p.evStagger);
        if (p.stream && p.stream != c->stream) (void)hipStreamDestroy(p.stream);
        p = TrainPart();
    }
    if (t.evFork) (void)hipEventDestroy(t.evFork);
    t = FrameTrain();
}
void free_frame(PtContext* c)
{
    free_train(c);
    for (FramePipe& p : c->pipe) free_pipe_buffers(p);
    dfree(c->dAccum); dfree(c->dCurrent);
    c->W = c->H = c->slots = 0;
    c->rngPipe = c->lastPipe = c->resolvePipe = 0;
}

/* Launch shape of the persistent traversal kernels: workgroups per CU and the number of idle lanes at which a wavefront
 * refills.  Measured optima differ by use: a batch in the order it was given 5 and 36; a batch ordered by origin cell
 * 6 and 24 (its lanes share lines, so more of them in flight cost less); the passes of the path tracer 3 and 12
 * (several of them share the CUs at any time, and their rays are coherent).  PT_OPT_PERSIST_BLOCKS / PT_OPT_REFILL_MIN
 * override all three. */
enum { SHAPE_BATCH = 0, SHAPE_ORDERED_BATCH = 1, SHAPE_FRAME = 2 };
/* A scene whose traversal records an XCD's L2 holds twice over (CBbunny: 3.6 MB): its walks are short, a frame's time goes to
 * the passes between them, and narrow traversal launches that leave those passes wave slots do better (1080p, CBbunny,
 * primary + 1 bounce, three frames in flight: 2 / 3 / 5 workgroups per CU -> 1 416 / 1 349 / 1 124 spp/s; on 33 MB and 742 MB of
 * records three is as good or better). */
bool small_scene(const PtContext* c) { return (uint64_t)c->N * 64u <= (8u << 20); }
int blocks_per_cu(PtContext* c, int shape)
{
    static const int d[3] = { 5, 6, 3 };
    if (c->userBlocks) return c->optBlocksPerCU;
    return shape == SHAPE_FRAME && small_scene(c) ? 2 : d[shape];
}
int refill_min(PtContext* c, int shape) { static const int d[3] = { 36, 24, 12 }; return c->userRefill ? c->optRefillMin : d[shape]; }
int grid_blocks(PtContext* c) { return c->numCUs * (c->optBlocksPerCU > 7 || c->userBlocks ? c->optBlocksPerCU : 7); }   /* sizes the spill areas: the widest default launch is seven workgroups per CU */

/* Spill areas of the LDS stacks: one per traversal kernel that can be running at a time, that is two per pipe
 * (closest-hit pass and shadow pass side by side). */
/* Entries a lane's stack can need beyond its LDS part: the tree's depth for the two-box walks; a level of the treelet
 * walk can leave three entries behind where two levels of a two-box walk leave two. */
uint32_t spill_entries(const PtContext* c)
{
    const uint32_t deepest = c->maxStack > c->treeletMaxStack ? c->maxStack : c->treeletMaxStack;
    const uint32_t inLds = PT_CERT_STACK_LDS < PT_STACK_LDS ? PT_CERT_STACK_LDS : PT_STACK_LDS;      /* the kernel that keeps the fewest entries in LDS spills the most */
    return deepest > inLds ? deepest - inLds : 0;
}

int ensure_overflow(PtContext* ctx)
{
    uint32_t need = spill_entries(ctx);
    uint64_t lanes = (uint64_t)grid_blocks(ctx) * PT_BLOCK;
    int areas = 2;
    for (int k = 1; k < PT_MAX_PIPES; k++) if (ctx->pipe[k].ready) areas = 2 * (k + 1);
    if (areas < PT_MAX_TRAIN_PARTS) areas = PT_MAX_TRAIN_PARTS;      /* the frame train: one per part */
    if (need <= ctx->overflowEntries && lanes <= ctx->overflowLanes && areas <= ctx->overflowAreas && (ctx->dOverflow || need == 0)) return 0;
    for (FramePipe& p : ctx->pipe) { if (p.stream) (void)hipStreamSynchronize(p.stream); if (p.stream2) (void)hipStreamSynchronize(p.stream2); }
    dfree(ctx->dOverflow);
    ctx->overflowEntries = need; ctx->overflowLanes = lanes; ctx->overflowAreas = areas;
    if (need) HIPCHK(hipMalloc((void**)&ctx->dOverflow, (size_t)areas * need * lanes * sizeof(uint2)));
    return 0;
}

void fill_trace_args(PtContext* c, TraceArgs& a)
{
    memset(&a, 0, sizeof(a));
    a.nodes = c->dNodes; a.numNodes = c->N; a.indices = c->dIndices; a.positions = c->dPositions;
    a.records = c->dRecords; a.recordBytes = c->recordBytes; a.leafBase = c->leafBase; a.tris = c->dRecords + c->leafBase / 16;
    a.leafRef = c->dLeafRef; a.lastLeaf = c->lastLeaf;
    a.treelets = c->dTreelets; a.treeletBytes = c->treeletBytes;
    for (int k = 0; k < 3; k++) a.rootC[k] = c->rootC[k];
    for (int k = 0; k < 6; k++) a.rootH[k] = c->rootH[k];
    for (int k = 0; k < 3; k++) a.lastC[k] = c->lastC[k];
    for (int k = 0; k < 6; k++) a.lastH[k] = c->lastH[k];
    a.rootRef = (c->T == 1) ? PT_LEAF_BIT : 0u;
    a.overflow = c->dOverflow; a.overflowEntries = c->overflowEntries;
    a.refillMin = (uint32_t)c->optRefillMin; a.claimChunk = (uint32_t)c->optClaimChunk;
    a.stats = c->dStats;
}

/* Where a batch's scratch lives: buffers the context owns (pt_trace_rays) or parts of the static scratch array
 * (pt_trace_rays_noalloc, which also fixes the spill area and with it the number of workgroups). */
struct BatchScratch {
    uint32_t* fetch = nullptr; uint32_t* sortOrder = nullptr; uint16_t* sortKeys = nullptr; uint32_t* sortBins = nullptr;
    uint32_t* flagList = nullptr; uint32_t* flagCtr = nullptr;
    uint2* spill = nullptr; uint32_t spillEntries = 0; int maxBlocks = 0;     /* maxBlocks != 0: spill and grid are fixed, nothing is allocated */
};

/* area: which spill area the launch may use (2*pipe for the pipe's first stream, 2*pipe+1 for its second) */
template <int MODE, bool ANYHIT>
int launch_trace(PtContext* ctx, TraceArgs& a, bool count, hipStream_t stream, int area, const BatchScratch* fixed = nullptr)
{
    const int shape = MODE != MODE_RAYS ? SHAPE_FRAME : (a.order ? SHAPE_ORDERED_BATCH : SHAPE_BATCH);
    int blocks = ctx->numCUs * blocks_per_cu(ctx, shape);
    if (fixed && fixed->maxBlocks) {
        a.overflow = fixed->spill; a.overflowEntries = fixed->spillEntries;
        if (blocks > fixed->maxBlocks) blocks = fixed->maxBlocks;
    } else {
        if (int rc = ensure_overflow(ctx)) return rc;
        a.overflow = ctx->dOverflow ? ctx->dOverflow + (size_t)area * ctx->overflowEntries * ctx->overflowLanes : nullptr;
        a.overflowEntries = ctx->overflowEntries;
    }
    dim3 grid(blocks), block(PT_BLOCK);
    a.refillMin = (uint32_t)refill_min(ctx, shape);
    /* 3 (certified walk): closest-hit passes launch k_trace_cert themselves (launch_cert) and come here for the second
       walk of the rays it listed, in reference order; any-hit passes get the near-first walk, whose verdict no order
       changes; 48 B records and trees whose boxes do not nest get the reference-order walk throughout */
    int trav = ctx->optTraversal;
    if (trav == 3) trav = ANYHIT ? 2 : 1;
    /* The near-first walk over 64 B records takes 66 VGPRs, so a SIMD holds seven of its wavefronts: an ordered batch runs
       seven workgroups per CU wide and refills a little later (closest hit, workgroups x idle lanes -> ms per 2^26 rays:
       6 x 20 / 24 / 28: 24.89 / 24.98 / 25.29; 7 x 20 / 24 / 26 / 28 / 30: 24.94 / 24.63 / 24.58 / 24.62 / 24.68; any hit:
       6 x 24 19.73, 7 x 24 / 28 / 32 19.93 / 19.64 / 19.52).  The other walks keep more registers and the table's shape. */
    if (MODE == MODE_RAYS && shape == SHAPE_ORDERED_BATCH && trav == 2 && !ctx->dTreelets && !ctx->compact) {
        if (!ctx->userBlocks) {
            int wide = ctx->numCUs * 7;
            if (fixed && fixed->maxBlocks && wide > fixed->maxBlocks) wide = fixed->maxBlocks;      /* a fixed spill area bounds the launch */
            grid = dim3(wide);
        }
        if (!ctx->userRefill) a.refillMin = ANYHIT ? 32u : 26u;
    }
    if (trav == 0) {
        if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_ref<MODE_RAYS, ANYHIT, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_ref<MODE, ANYHIT, false>), grid, block, 0, stream, a);
    } else if (ctx->dTreelets) {
        if (trav == 1) {
            if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_tree<MODE_RAYS, ANYHIT, false, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k_trace_tree<MODE, ANYHIT, false, false>), grid, block, 0, stream, a);
        } else {
            if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_tree<MODE_RAYS, ANYHIT, true, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k_trace_tree<MODE, ANYHIT, true, false>), grid, block, 0, stream, a);
        }
    } else if (ctx->compact) {
        if (trav == 1) {
            if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_flat<MODE_RAYS, ANYHIT, false, true, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k_trace_flat<MODE, ANYHIT, false, false, true>), grid, block, 0, stream, a);
        } else {
            if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_flat<MODE_RAYS, ANYHIT, true, true, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k_trace_flat<MODE, ANYHIT, true, false, true>), grid, block, 0, stream, a);
        }
    } else if (trav == 1) {
        if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_flat<MODE_RAYS, ANYHIT, false, true, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_flat<MODE, ANYHIT, false, false, false>), grid, block, 0, stream, a);
    } else {
        if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_flat<MODE_RAYS, ANYHIT, true, true, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_flat<MODE, ANYHIT, true, false, false>), grid, block, 0, stream, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

/* The closest-hit pass of one bounce and the shadow pass of the bounce before it as one launch (MODE_MIXED of k_trace_flat,
 * 64 B or 48 B records, traversals 1 and 2): both wait for the same shade pass and for nothing else, so one pool of rays keeps
 * more lanes busy than two launches side by side, and a frame needs one stream instead of two. */
int launch_mixed(PtContext* ctx, TraceArgs& a, hipStream_t stream, int area, bool train = false)
{
    if (int rc = ensure_overflow(ctx)) return rc;
    a.overflow = ctx->dOverflow ? ctx->dOverflow + (size_t)area * ctx->overflowEntries * ctx->overflowLanes : nullptr;
    a.overflowEntries = ctx->overflowEntries;
    /* two passes' rays in one pool: one more workgroup per CU than a pass alone gets (measured: 3, 4, 5 -> 158, 160, 161 spp/s).
       The train's launches are batch-sized, but six workgroups per CU leave the other band's shade, launch and resolve passes
       no wave slots beside them: 3, 4, 5, 6 -> 183, 189, 194, 189 spp/s (atrium: 5, 6 -> 209, 197) */
    const int wide = small_scene(ctx) ? (train ? 4 : 2) : (train ? 5 : 4);      /* small scene, train: 3, 4, 5 -> 978, 1 025, 999 spp/s at 8 slots */
    dim3 grid(ctx->numCUs * (ctx->userBlocks ? ctx->optBlocksPerCU : wide)), block(PT_BLOCK);
    a.refillMin = (uint32_t)refill_min(ctx, SHAPE_FRAME);
    const bool ordered = ctx->optTraversal == 2;      /* 3 comes here where the certified walk is not to be had: reference order */
    if (ctx->dTreelets) {
        if (ordered) hipLaunchKernelGGL((k_trace_tree<MODE_MIXED, false, true, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_tree<MODE_MIXED, false, false, false>), grid, block, 0, stream, a);
    } else if (ctx->compact) {
        if (ordered) hipLaunchKernelGGL((k_trace_flat<MODE_MIXED, false, true, false, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_flat<MODE_MIXED, false, false, false, true>), grid, block, 0, stream, a);
    } else {
        if (ordered) hipLaunchKernelGGL((k_trace_flat<MODE_MIXED, false, true, false, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_flat<MODE_MIXED, false, false, false, false>), grid, block, 0, stream, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

/* Every entry point that touches the device starts here: allocations go to the current device, and a host thread that
 * drives several contexts has whichever device it used last. */
int enter(PtContext* ctx) { HIPCHK(hipSetDevice(ctx->device)); return 0; }

/* The certified walk over the batch or job list `a` describes, then the reference-order walk over the rays it listed.
 * flag: room for one index per ray; ctr: two zeroed words (count of listed rays, the second walk's fetch counter). */
template <int MODE>
int launch_cert(PtContext* ctx, TraceArgs& a, bool count, hipStream_t stream, int area, uint32_t* flag, uint32_t* ctr, const BatchScratch* fixed = nullptr)
{
    const int shape = MODE != MODE_RAYS ? SHAPE_FRAME : (a.order ? SHAPE_ORDERED_BATCH : SHAPE_BATCH);
    int blocks = ctx->numCUs * blocks_per_cu(ctx, shape);
    if (fixed && fixed->maxBlocks) {
        a.overflow = fixed->spill; a.overflowEntries = fixed->spillEntries;
        if (blocks > fixed->maxBlocks) blocks = fixed->maxBlocks;
    } else {
        if (int rc = ensure_overflow(ctx)) return rc;
        a.overflow = ctx->dOverflow ? ctx->dOverflow + (size_t)area * ctx->overflowEntries * ctx->overflowLanes : nullptr;
        a.overflowEntries = ctx->overflowEntries;
    }
    a.flagList = flag; a.flagCount = ctr;
    /* a step of this walk is half again as long as the near-first walk's, so idle lanes cost more and are refilled sooner:
       batches at 4 / 8 / 12 / 16 / 24 idle lanes -> 2 094 / 2 170 / 2 165 / 2 138 / 2 067 Mrays/s */
    a.refillMin = (uint32_t)(MODE == MODE_RAYS && !ctx->userRefill ? 10 : refill_min(ctx, shape));
    dim3 grid(blocks), block(PT_BLOCK);
    if (count) hipLaunchKernelGGL((k_trace_cert<MODE, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_trace_cert<MODE, false>), grid, block, 0, stream, a);
    HIPCHK(hipGetLastError());
    TraceArgs b = a;
    b.order = flag; b.countPtr = ctr; b.fetchCounter = ctr + 1; b.flagList = nullptr; b.flagCount = nullptr;
    if (ctx->optSecondWalk == 0)
        return launch_trace<MODE, false>(ctx, b, count, stream, area, fixed);    /* traversal 3 maps to the reference-order walk there */
    /* the listed rays are few and their walks long: one wavefront per ray (k_trace_ref_wave); no stack, no spill area */
    dim3 grid2(ctx->numCUs * 8), block2(64);        /* one wavefront per workgroup */
    if (count) hipLaunchKernelGGL((k_trace_ref_wave<MODE, true>), grid2, block2, 0, stream, b);
    else hipLaunchKernelGGL((k_trace_ref_wave<MODE, false>), grid2, block2, 0, stream, b);
    HIPCHK(hipGetLastError());
    return 0;
}

/* The frame train's launch under traversal 3: the certified walk over the step's jobs, with the shadow rays of the step before
 * riding along, then the reference-order walk over the jobs it listed. */
int launch_cert_mixed(PtContext* ctx, TraceArgs& a, hipStream_t stream, int area, uint32_t* flag, uint32_t* ctr)
{
    if (int rc = ensure_overflow(ctx)) return rc;
    a.overflow = ctx->dOverflow ? ctx->dOverflow + (size_t)area * ctx->overflowEntries * ctx->overflowLanes : nullptr;
    a.overflowEntries = ctx->overflowEntries;
    a.flagList = flag; a.flagCount = ctr;
    a.refillMin = (uint32_t)refill_min(ctx, SHAPE_FRAME);
    dim3 grid(ctx->numCUs * (ctx->userBlocks ? ctx->optBlocksPerCU : small_scene(ctx) ? 3 : 6)), block(PT_BLOCK);      /* small scene: 2 ... 6 -> 402, 415, 402, 383, 382 spp/s */
    hipLaunchKernelGGL((k_trace_cert<MODE_MIXED, false>), grid, block, 0, stream, a);
    HIPCHK(hipGetLastError());
    TraceArgs b = a;
    b.order = flag; b.countPtr = ctr; b.countPtr2 = nullptr; b.fetchCounter = ctr + 1; b.flagList = nullptr; b.flagCount = nullptr;
    if (ctx->optSecondWalk == 0) return launch_trace<MODE_JOBS, false>(ctx, b, false, stream, area);
    /* one wavefront per workgroup: these find room beside a launch that fills the CUs more easily than groups of four */
    hipLaunchKernelGGL((k_trace_ref_wave<MODE_JOBS, false>), dim3(ctx->numCUs * 8), dim3(64), 0, stream, b);
    HIPCHK(hipGetLastError());
    return 0;
}

int check_scene(PtContext* ctx) { return ctx->T ? 0 : ctx->fail(-2, "no scene uploaded (call pt_scene_upload first)"); }
int check_frame(PtContext* ctx)
{
    if (int rc = check_scene(ctx)) return rc;
    if (!ctx->W) return ctx->fail(-2, "no frame resources (call pt_resize first)");
    if (!ctx->haveConsts) return ctx->fail(-2, "no constants (call pt_set_constants first)");
    return 0;
}

void fill_frame_args(PtContext* c, FramePipe& p, FrameArgs& f, uint32_t depth = 0)
{
    memset(&f, 0, sizeof(f));
    f.c = c->consts; f.W = c->W; f.H = c->H; f.slots = c->slots;
    f.zeroNextSlot = c->optStale ? 0u : 1u; f.reseed = c->optReseed ? 1u : 0u; f.seedFromJitter = c->accumEmpty ? 1u : 0u;
    f.stack = p.dStack;
    f.rngStack = c->rngInTrain ? c->train.dRing + (size_t)c->train.head * c->W * c->H * c->slots : c->pipe[c->rngPipe].dStack;
    f.jobs = p.dJobs[p.curJobs]; f.jobsOut = p.dJobs[p.curJobs ^ 1]; f.ctr = p.dCtr; f.shadowCtr = p.dCtr + CTR_SHADOW + 2 * depth;
    f.hitRecs = p.dHitRecs; f.shadowRays = p.dShadow[0]; f.fuseCompact = 0; f.aux = c->dAux; f.lights = c->dLights; f.blueNoise = c->dBlueNoise;
    f.accumulation = c->dAccum; f.current = c->dCurrent; f.rayCounts = p.dRayCounts;
    f.row0 = 0; f.row1 = c->H;
}

/* Streams, events and buffers of pipe k at the current resolution. */
int ensure_pipe(PtContext* ctx, int k)
{
    FramePipe& p = ctx->pipe[k];
    p.index = k;
    if (!p.stream) {
        if (k == 0) p.stream = ctx->stream;
        else HIPCHK(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&p.stream2, hipStreamNonBlocking));
        for (int q = 0; q < 2; q++) {
            HIPCHK(hipEventCreateWithFlags(&p.evShade[q], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&p.evShadow[q], hipEventDisableTiming));
        }
        HIPCHK(hipEventCreateWithFlags(&p.evShade0, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p.evDone, hipEventDisableTiming));
    }
    if (p.ready) return 0;
    free_pipe_buffers(p);                                 /* what an earlier attempt that ran out of memory left behind */
    const size_t px = (size_t)ctx->W * ctx->H;
    HIPCHK(hipMalloc((void**)&p.dStack, px * ctx->slots * sizeof(PtRayStackSlot)));
    HIPCHK(hipMalloc((void**)&p.dJobs[0], px * 4));
    HIPCHK(hipMalloc((void**)&p.dJobs[1], px * 4));
    HIPCHK(hipMalloc((void**)&p.dHitRecs, px * sizeof(PtHitRec)));
    HIPCHK(hipMalloc((void**)&p.dShadow[0], px * sizeof(PtShadowRay)));
    HIPCHK(hipMalloc((void**)&p.dShadow[1], px * sizeof(PtShadowRay)));
    HIPCHK(hipMalloc((void**)&p.dCtr, (size_t)ctr_words(ctx->slots) * 4));
    HIPCHK(hipMalloc((void**)&p.dRayCounts, (size_t)2 * ctx->slots * 8));
    /* the reference leaves these undefined; zero makes the first frame deterministic.  The fills are queued on the
       pipe's own stream, ahead of everything that will ever touch these buffers: a fill of device memory on the null
       stream returns before it has run and is not ordered against the non-blocking streams the frames run on (a
       3 GB ray stack was still being zeroed while the first frame wrote to it) */
    HIPCHK(hipMemsetAsync(p.dStack, 0, px * ctx->slots * sizeof(PtRayStackSlot), p.stream));
    HIPCHK(hipMemsetAsync(p.dJobs[0], 0xFF, px * 4, p.stream));
    HIPCHK(hipMemsetAsync(p.dJobs[1], 0xFF, px * 4, p.stream));
    HIPCHK(hipMemsetAsync(p.dCtr, 0, (size_t)ctr_words(ctx->slots) * 4, p.stream));
    HIPCHK(hipMemsetAsync(p.dRayCounts, 0, (size_t)2 * ctx->slots * 8, p.stream));
    p.ready = true; p.curJobs = 0;
    return 0;
}

void destroy_pipe(PtContext* ctx, FramePipe& p)
{
    free_pipe_buffers(p);
    for (int q = 0; q < 2; q++) { if (p.evShade[q]) (void)hipEventDestroy(p.evShade[q]); if (p.evShadow[q]) (void)hipEventDestroy(p.evShadow[q]); }
    if (p.evShade0) (void)hipEventDestroy(p.evShade0);
    if (p.evDone) (void)hipEventDestroy(p.evDone);
    if (p.stream2) (void)hipStreamDestroy(p.stream2);
    if (p.stream && p.stream != ctx->stream) (void)hipStreamDestroy(p.stream);
    p = FramePipe();
}

int sync_all(PtContext* ctx)
{
    for (FramePipe& p : ctx->pipe) {
        if (p.stream) HIPCHK(hipStreamSynchronize(p.stream));
        if (p.stream2) HIPCHK(hipStreamSynchronize(p.stream2));
    }
    for (TrainPart& p : ctx->train.part) if (p.stream) HIPCHK(hipStreamSynchronize(p.stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // namespace

extern "C" {

PT_API const char* pt_version(void) { return "ptcore 0.1 (gfx950)"; }

PT_API int pt_ctx_create(int device_id, PtContext** out)
{
    if (!out) return -1;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) { g_createError = "no HIP device available (this library has no CPU path)"; return -3; }
    if (device_id < 0 || device_id >= n) { g_createError = "device id out of range"; return -1; }
    if ((e = hipSetDevice(device_id)) != hipSuccess) { g_createError = hipGetErrorString(e); return -10; }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) { g_createError = hipGetErrorString(e); return -10; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { g_createError = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only"; return -3; }
    PtContext* ctx = new PtContext;
    ctx->device = device_id;
    ctx->numCUs = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->evFork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc((void**)&ctx->dFetch, 4)) != hipSuccess || (e = hipMalloc((void**)&ctx->dStats, 4 * 8)) != hipSuccess) {
        g_createError = hipGetErrorString(e);
        pt_ctx_destroy(ctx);
        return -10;
    }
    memset(&ctx->consts, 0, sizeof(ctx->consts));
    /* this device's copy of the static scratch array: its address is asked for here, with the device current.  A host
       that never calls pt_trace_rays_noalloc does not need it: a failure is kept for that call to report (-2) */
    ctx->scratchPtr = resolve_scratch(device_id, ctx->scratchErr);
    *out = ctx;
    return 0;
}

PT_API void pt_ctx_destroy(PtContext* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)sync_all(ctx);
    if (ctx->comm && g_nccl.CommDestroy) g_nccl.CommDestroy(ctx->comm);
    free_scene(ctx); free_frame(ctx); destroy_train(ctx);
    for (FramePipe& p : ctx->pipe) destroy_pipe(ctx, p);
    dfree(ctx->dFetch); dfree(ctx->dStats);
    dfree(ctx->dSortOrder); dfree(ctx->dSortKeys); dfree(ctx->dSortBins);
    dfree(ctx->dFlagList); dfree(ctx->dFlagCtr);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->evFork) (void)hipEventDestroy(ctx->evFork);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

PT_API const char* pt_last_error(PtContext* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

PT_API int pt_synchronize(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = train_join(ctx)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API void* pt_stream(PtContext* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

PT_API int pt_set_option(PtContext* ctx, int option, int value)
{
    if (!ctx) return -1;
    if (int rc = settle(ctx)) return rc;
    switch (option) {
        case PT_OPT_TRAVERSAL: if (value < 0 || value > 3) return ctx->fail(-1, "PT_OPT_TRAVERSAL must be 0, 1, 2 or 3"); ctx->optTraversal = value; return 0;
        case PT_OPT_STALE_SLOTS: ctx->optStale = value != 0; return 0;
        case PT_OPT_RESEED: ctx->optReseed = value != 0; return 0;
        case PT_OPT_PERSIST_BLOCKS: if (value < 1 || value > 8) return ctx->fail(-1, "PT_OPT_PERSIST_BLOCKS must be 1..8"); ctx->optBlocksPerCU = value; ctx->userBlocks = true; return 0;
        case PT_OPT_REFILL_MIN: if (value < 1 || value > 64) return ctx->fail(-1, "PT_OPT_REFILL_MIN must be 1..64"); ctx->optRefillMin = value; ctx->userRefill = true; return 0;
        case PT_OPT_CLAIM_CHUNK: if (value < 0 || value > 1024 || value % 64) return ctx->fail(-1, "PT_OPT_CLAIM_CHUNK must be 0 or a multiple of 64 up to 1024"); ctx->optClaimChunk = value; return 0;
        case PT_OPT_RECORD_LAYOUT: if (value < 0 || value > 2) return ctx->fail(-1, "PT_OPT_RECORD_LAYOUT must be 0, 1 or 2"); ctx->optRecordLayout = value; return 0;
        case PT_OPT_SORT_BITS: if (value < 1 || value > 4) return ctx->fail(-1, "PT_OPT_SORT_BITS must be 1..4"); ctx->optSortBits = value; return 0;
        case PT_OPT_SORT_RAYS: if (value < 0 || value > 2) return ctx->fail(-1, "PT_OPT_SORT_RAYS must be 0, 1 or 2"); ctx->optSortRays = value; return 0;
        case PT_OPT_SHADOW_STREAM: if (value < 0 || value > 2) return ctx->fail(-1, "PT_OPT_SHADOW_STREAM must be 0, 1 or 2"); ctx->optShadowStream = value; return 0;
        case PT_OPT_MERGE_PASSES: ctx->optMergePasses = value != 0; return 0;
        case PT_OPT_FRAME_TRAIN: if (value < 0 || value > 2) return ctx->fail(-1, "PT_OPT_FRAME_TRAIN must be 0, 1 or 2"); ctx->optFrameTrain = value; return 0;
        case PT_OPT_TRAIN_KEEP: ctx->optTrainKeep = value != 0; return 0;
        case PT_OPT_TRAIN_MAX_MB: if (value < 0) return ctx->fail(-1, "PT_OPT_TRAIN_MAX_MB must not be negative"); ctx->optTrainMaxMB = value; return 0;
        case PT_OPT_TRAIN_PARTS: if (value < 1 || value > PT_MAX_TRAIN_PARTS) return ctx->fail(-1, "PT_OPT_TRAIN_PARTS must be 1..4"); ctx->optTrainParts = value; return 0;
        case PT_OPT_SECOND_WALK: if (value < 0 || value > 1) return ctx->fail(-1, "PT_OPT_SECOND_WALK must be 0 or 1"); ctx->optSecondWalk = value; return 0;
        case PT_OPT_FRAMES_IN_FLIGHT: if (value < 1 || value > PT_MAX_PIPES) return ctx->fail(-1, "PT_OPT_FRAMES_IN_FLIGHT must be 1..8"); ctx->optFramesInFlight = value; return 0;
        default: return ctx->fail(-1, "unknown option");
    }
}
PT_API int pt_get_option(PtContext* ctx, int option, int* value)
{
    if (!ctx || !value) return -1;
    switch (option) {
        case PT_OPT_TRAVERSAL: *value = ctx->optTraversal; return 0;
        case PT_OPT_STALE_SLOTS: *value = ctx->optStale; return 0;
        case PT_OPT_RESEED: *value = ctx->optReseed; return 0;
        case PT_OPT_PERSIST_BLOCKS: *value = ctx->optBlocksPerCU; return 0;
        case PT_OPT_REFILL_MIN: *value = ctx->optRefillMin; return 0;
        case PT_OPT_CLAIM_CHUNK: *value = ctx->optClaimChunk; return 0;
        case PT_OPT_FRAMES_IN_FLIGHT: *value = ctx->optFramesInFlight; return 0;
        case PT_OPT_SECOND_WALK: *value = ctx->optSecondWalk; return 0;
        case PT_OPT_SHADOW_STREAM: *value = ctx->optShadowStream; return 0;
        case PT_OPT_MERGE_PASSES: *value = ctx->optMergePasses; return 0;
        case PT_OPT_FRAME_TRAIN: *value = ctx->optFrameTrain; return 0;
        case PT_OPT_TRAIN_MAX_MB: *value = ctx->optTrainMaxMB; return 0;
        case PT_OPT_TRAIN_KEEP: *value = ctx->optTrainKeep; return 0;
        case PT_OPT_TRAIN_PARTS: *value = ctx->optTrainParts; return 0;
        case PT_OPT_SORT_RAYS: *value = ctx->optSortRays; return 0;
        case PT_OPT_SORT_BITS: *value = ctx->optSortBits; return 0;
        case PT_OPT_RECORD_LAYOUT: *value = ctx->optRecordLayout; return 0;
        default: return ctx->fail(-1, "unknown option");
    }
}

PT_API int pt_scene_upload(PtContext* ctx, const float* positions, uint32_t V, const PtVertexAux* aux, const uint32_t* indices, uint32_t T,
                           const PtBVHNode* nodes, uint32_t N, const PtLight* lights, uint32_t L, const uint16_t* blue_noise)
try {
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (!positions || !indices || !nodes || V == 0 || T == 0) return ctx->fail(-1, "pt_scene_upload: null or empty geometry");
    if (N != 2 * T - 1) return ctx->fail(-1, "pt_scene_upload: node count must be 2*T-1");
    for (size_t i = 0; i < 3 * (size_t)T; i++) if (indices[i] >= V) return ctx->fail(-1, "pt_scene_upload: vertex index out of range");
    for (uint32_t i = 0; i < N; i++) {
        if (nodes[i].next < 0 || (uint32_t)nodes[i].next >= N || nodes[i].right >= (int32_t)N) return ctx->fail(-1, "pt_scene_upload: node link out of range");
        if (nodes[i].right <= 0 && ((uint32_t)(-nodes[i].right) % 3 != 0 || (uint32_t)(-nodes[i].right) / 3 >= T)) return ctx->fail(-1, "pt_scene_upload: leaf offset out of range");
    }
    if (int rc = sync_all(ctx)) return rc;
    free_scene(ctx);

    /* traversal records: 64 B each, or 48 B each when the option asks for it and every box fits that encoding; interior
       and leaf records share one buffer so that one buffer descriptor serves every fetch */
    char ebuf[256] = {0};
    uint32_t maxStack = 0;
    std::vector<PtNode2c> c2; std::vector<PtTriC> ct; std::vector<uint32_t> leafRef;
    std::vector<PtNode2> n2; std::vector<PtTri> tr;
    ctx->compact = false;
    if (ctx->optRecordLayout == 1) {
        c2.resize(T > 1 ? T - 1 : 1); ct.resize(T); leafRef.resize(T);
        int rc = pt_flatten_bvh_compact(nodes, N, positions, indices, T, c2.data(), ct.data(), leafRef.data(), ctx->rootBox, &maxStack, ebuf, sizeof(ebuf));
        if (rc < 0) return ctx->fail(-1, ebuf);
        ctx->compact = rc == 0;
    }
    if (!ctx->compact) {
        c2.clear(); ct.clear(); leafRef.clear();
        n2.resize(T > 1 ? T - 1 : 1); tr.resize(T);
        if (pt_flatten_bvh(nodes, N, positions, indices, T, n2.data(), tr.data(), ctx->rootBox, &maxStack, ebuf, sizeof(ebuf)) != 0)
            return ctx->fail(-1, ebuf);
    }
    ctx->maxStack = maxStack;
    std::vector<unsigned char> tl; uint64_t tlUnits = 0; uint32_t tlStack = 0;
    if (ctx->optRecordLayout == 2 && T > 1) {
        float rb[6];
        int rc = pt_flatten_bvh_treelets(nodes, N, positions, indices, T, nullptr, 0, &tlUnits, rb, &tlStack, ebuf, sizeof(ebuf));
        if (rc < 0) return ctx->fail(-1, ebuf);
        if (rc == 0) {
            tl.resize((size_t)tlUnits * 64 + 64);          /* one spare unit: a 128 B fetch may start at the last leaf */
            rc = pt_flatten_bvh_treelets(nodes, N, positions, indices, T, tl.data(), tlUnits, &tlUnits, rb, &tlStack, ebuf, sizeof(ebuf));
            if (rc < 0) return ctx->fail(-1, ebuf);
        }
        if (rc != 0) { tl.clear(); tlUnits = 0; tlStack = 0; }          /* a box that does not fit the encoding: 64 B records only */
    }
    /* the certified walk skips a subtree when the ray enters its box beyond max_t, which is sound only if everything in the
       subtree lies inside that box: true of every tree pt_build_bvh makes, checked here because the array is the caller's */
    ctx->boxesNest = true;
    for (uint32_t i = 0; i < N && ctx->boxesNest; i++) {
        const PtBVHNode& n = nodes[i];
        auto inside = [&](const float* lo, const float* hi) { for (int k = 0; k < 3; k++) if (!(lo[k] >= n.a[k] && hi[k] <= n.b[k])) return false; return true; };
        if (n.right > 0) ctx->boxesNest = inside(nodes[i + 1].a, nodes[i + 1].b) && inside(nodes[n.right].a, nodes[n.right].b);
        else {
            const uint32_t t = (uint32_t)(-n.right) / 3u;
            for (int v = 0; v < 3 && ctx->boxesNest; v++) { const float* p = positions + 4 * (size_t)indices[3 * (size_t)t + v]; ctx->boxesNest = inside(p, p); }
        }
    }
    ctx->lastLeaf = (uint32_t)(-nodes[N - 1].right) / 3u;          /* the last node of a pre-order array is a leaf */
    for (int k = 0; k < 3; k++) {      /* root box digest, as for every other box (intersections.glsl:3-5) */
        ctx->rootC[k] = (ctx->rootBox[k] + ctx->rootBox[3 + k]) * 0.5f;
        _Float16 ha = (_Float16)(ctx->rootBox[k] - ctx->rootC[k]), hb = (_Float16)(ctx->rootBox[3 + k] - ctx->rootC[k]);
        memcpy(&ctx->rootH[k], &ha, 2); memcpy(&ctx->rootH[3 + k], &hb, 2);
        const PtBVHNode& last = nodes[N - 1];
        ctx->lastC[k] = (last.a[k] + last.b[k]) * 0.5f;
        _Float16 la = (_Float16)(last.a[k] - ctx->lastC[k]), lb = (_Float16)(last.b[k] - ctx->lastC[k]);
        memcpy(&ctx->lastH[k], &la, 2); memcpy(&ctx->lastH[3 + k], &lb, 2);
    }

    HIPCHK(hipMalloc((void**)&ctx->dPositions, (size_t)V * 16));
    HIPCHK(hipMalloc((void**)&ctx->dIndices, (size_t)T * 12));
    HIPCHK(hipMalloc((void**)&ctx->dNodes, (size_t)N * sizeof(PtBVHNode)));
    const size_t recSize = ctx->compact ? (size_t)PT_COMPACT_STRIDE : sizeof(PtNode2);
    const size_t interiorBytes = (size_t)(T - 1) * recSize, recBytes = interiorBytes + (size_t)T * recSize;
    if (recBytes >= 0xFFFFFFFFull) return ctx->fail(-1, "pt_scene_upload: more than 2^32 bytes of traversal records");
    HIPCHK(hipMalloc((void**)&ctx->dRecords, recBytes));
    ctx->leafBase = (uint32_t)interiorBytes; ctx->recordBytes = (uint32_t)recBytes;
    HIPCHK(hipMemcpy(ctx->dPositions, positions, (size_t)V * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->dIndices, indices, (size_t)T * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->dNodes, nodes, (size_t)N * sizeof(PtBVHNode), hipMemcpyHostToDevice));
    if (ctx->compact && recSize == sizeof(PtNode2c)) {
        if (T > 1) HIPCHK(hipMemcpy(ctx->dRecords, c2.data(), interiorBytes, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((char*)ctx->dRecords + interiorBytes, ct.data(), (size_t)T * recSize, hipMemcpyHostToDevice));
    } else if (ctx->compact) {      /* records spaced wider than their 48 B: staged with the padding */
        std::vector<unsigned char> staged(recBytes, 0);
        for (size_t i = 0; i + 1 < (size_t)T; i++) memcpy(staged.data() + i * recSize, &c2[i], sizeof(PtNode2c));
        for (size_t i = 0; i < (size_t)T; i++) memcpy(staged.data() + interiorBytes + i * recSize, &ct[i], sizeof(PtTriC));
        HIPCHK(hipMemcpy(ctx->dRecords, staged.data(), recBytes, hipMemcpyHostToDevice));
    } else {
        if (T > 1) HIPCHK(hipMemcpy(ctx->dRecords, n2.data(), interiorBytes, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((char*)ctx->dRecords + interiorBytes, tr.data(), (size_t)T * recSize, hipMemcpyHostToDevice));
    }
    if (!tl.empty()) {
        HIPCHK(hipMalloc((void**)&ctx->dTreelets, tl.size()));
        HIPCHK(hipMemcpy(ctx->dTreelets, tl.data(), tl.size(), hipMemcpyHostToDevice));
        ctx->treeletBytes = (uint32_t)((size_t)tlUnits * 64); ctx->treeletMaxStack = tlStack;
    }
    if (ctx->compact) {
        HIPCHK(hipMalloc((void**)&ctx->dLeafRef, (size_t)T * 4));
        HIPCHK(hipMemcpy(ctx->dLeafRef, leafRef.data(), (size_t)T * 4, hipMemcpyHostToDevice));
    }
    if (aux) {
        HIPCHK(hipMalloc((void**)&ctx->dAux, (size_t)V * 16));
        HIPCHK(hipMemcpy(ctx->dAux, aux, (size_t)V * 16, hipMemcpyHostToDevice));
    }
    if (lights && L) {
        HIPCHK(hipMalloc((void**)&ctx->dLights, (size_t)L * sizeof(PtLight)));
        HIPCHK(hipMemcpy(ctx->dLights, lights, (size_t)L * sizeof(PtLight), hipMemcpyHostToDevice));
    }
    if (blue_noise) {
        HIPCHK(hipMalloc((void**)&ctx->dBlueNoise, 65536 * 2));
        HIPCHK(hipMemcpy(ctx->dBlueNoise, blue_noise, 65536 * 2, hipMemcpyHostToDevice));
    }
    ctx->V = V; ctx->T = T; ctx->N = N; ctx->L = (lights && L) ? L : 0;
    return ensure_overflow(ctx);
} catch (...) { return ctx->fail(-11, "pt_scene_upload: out of host memory"); }

PT_API int pt_resize(PtContext* ctx, uint32_t W, uint32_t H, uint32_t max_slots)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (W == 0 || H == 0 || max_slots == 0) return ctx->fail(-1, "pt_resize: zero dimension");
    if ((uint64_t)W * H * max_slots >= 0xFFFFFFFFull) return ctx->fail(-1, "pt_resize: W*H*slots must fit a 32-bit slot index");
    if (int rc = settle(ctx)) return rc;
    if (int rc = sync_all(ctx)) return rc;
    free_frame(ctx);
    size_t px = (size_t)W * H;
    /* constants set for another size would make the launch pass index pixels this size does not have */
    if (ctx->haveConsts && ((uint32_t)ctx->consts.viewportSize[0] != W || (uint32_t)ctx->consts.viewportSize[1] != H || ctx->consts.numRays > max_slots))
        ctx->haveConsts = false;
    ctx->W = W; ctx->H = H; ctx->slots = max_slots;
    HIPCHK(hipMalloc((void**)&ctx->dAccum, px * 16));
    HIPCHK(hipMalloc((void**)&ctx->dCurrent, px * 8));
    HIPCHK(hipMemsetAsync(ctx->dAccum, 0, px * 16, ctx->stream));        /* on the context's stream: see ensure_pipe */
    HIPCHK(hipMemsetAsync(ctx->dCurrent, 0, px * 8, ctx->stream));
    ctx->accumEmpty = true;
    if (int rc = ensure_pipe(ctx, 0)) { free_frame(ctx); return rc; }      /* further pipes come with the first pt_render_samples */
    return 0;
}

PT_API int pt_set_constants(PtContext* ctx, const PtShaderConstants* c)
{
    if (!ctx || !c) return -1;
    if (ctx->W && ((uint32_t)c->viewportSize[0] != ctx->W || (uint32_t)c->viewportSize[1] != ctx->H)) return ctx->fail(-1, "pt_set_constants: viewportSize differs from pt_resize");
    if (ctx->slots && (c->numRays == 0 || c->numRays > ctx->slots)) return ctx->fail(-1, "pt_set_constants: numRays exceeds the slots allocated by pt_resize");
    if (ctx->N && c->numBVHNodes != ctx->N) return ctx->fail(-1, "pt_set_constants: numBVHNodes differs from the uploaded scene");
    if (c->viewportBase[0] != 0.0f || c->viewportBase[1] != 0.0f) return ctx->fail(-1, "pt_set_constants: viewportBase must be 0");
    if (ctx->train.live) {      /* frames in a kept train are finished under the constants they entered with; a new frameIndex alone changes nothing for them */
        PtShaderConstants was = ctx->consts, now = *c;
        was.frameIndex = now.frameIndex = 0;
        memset(was._pad0, 0, sizeof was._pad0); memset(now._pad0, 0, sizeof now._pad0); was._pad1 = now._pad1 = 0;      /* nothing reads the padding */
        if (memcmp(&was, &now, sizeof was)) if (int rc = settle(ctx)) return rc;
    }
    ctx->consts = *c; ctx->haveConsts = true;
    return 0;
}

PT_API int pt_clear_accumulation(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    if (int rc = settle(ctx)) return rc;
    HIPCHK(hipMemsetAsync(ctx->dAccum, 0, (size_t)ctx->W * ctx->H * 16, ctx->stream));
    ctx->accumEmpty = true;
    return 0;
}

/* launch.comp on pipe p.  The RNG state a pixel continues from lives in slot 0 of the stack the previous frame ran
 * on; when that is another pipe's, this pipe's stream first waits for that frame's depth-0 shade pass, after which
 * nothing writes those states any more. */
static int launch_on(PtContext* ctx, FramePipe& p)
{
    if (!ctx->dBlueNoise) return ctx->fail(-2, "pt_launch: scene was uploaded without a blue-noise table");
    FramePipe& src = ctx->pipe[ctx->rngPipe];
    /* (states left by the frame train need no event: its work is on the context's stream, which every pipe's stream joins
       before its first frame) */
    if (!ctx->rngInTrain && &src != &p && src.shade0Pending && !ctx->optReseed) HIPCHK(hipStreamWaitEvent(p.stream, src.evShade0, 0));
    p.curJobs = 0; p.deferredShadow = -1;
    FrameArgs f; fill_frame_args(ctx, p, f);
    dim3 grid((ctx->W + 31) / 32, (ctx->H + 7) / 8), block(256);
    hipLaunchKernelGGL(k_launch, grid, block, 0, p.stream, f);
    HIPCHK(hipGetLastError());
    ctx->rngPipe = ctx->lastPipe = p.index;
    ctx->rngInTrain = ctx->lastInTrain = false;
    return 0;
}

PT_API int pt_launch(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (int rc = check_frame(ctx)) return rc;
    return launch_on(ctx, ctx->pipe[0]);
}

/* Makes the pipe's first stream wait for shadow passes still running on its second stream. */
static int join_shadow(PtContext* ctx, FramePipe& p)
{
    for (int q = 0; q < 2; q++)
        if (p.shadowBusy[q]) { HIPCHK(hipStreamWaitEvent(p.stream, p.evShadow[q], 0)); p.shadowBusy[q] = false; }
    return 0;
}

/* One depth of the frame on pipe p.  overlapped = false: the three passes run in order on the pipe's first stream and
 * jobs are rewritten in place (what pt_trace promises).  overlapped = true (pt_render_sample[s]): the shade pass also
 * compacts the job list, and the shadow pass runs on the second stream beside the next depth's closest-hit pass; two
 * shadow queues alternate, and a queue is rewritten only after the shadow pass that read it has finished. */
static int trace_depth(PtContext* ctx, FramePipe& p, uint32_t depth, int speculative, bool overlapped)
{
    const int q = overlapped ? (int)(depth & 1u) : 0;
    uint32_t* shadowCtr = p.dCtr + CTR_SHADOW + 2 * depth;
    if (!overlapped) {
        if (int rc = join_shadow(ctx, p)) return rc;
        HIPCHK(hipMemsetAsync(p.dCtr + CTR_FETCH_CLOSEST, 0, 3 * 4, p.stream));                 /* fetch, compact-out, live */
        HIPCHK(hipMemsetAsync(shadowCtr, 0, 2 * 4, p.stream));
    }   /* overlapped: k_launch cleared every counter, k_record_counts clears the first three after each depth */
    FrameArgs f; fill_frame_args(ctx, p, f, depth);
    f.shadowRays = p.dShadow[q]; f.fuseCompact = overlapped ? 1u : 0u; f.tileShadow = speculative ? 1u : 0u;
    TraceArgs a; fill_trace_args(ctx, a);
    a.stack = p.dStack; a.hitRecs = p.dHitRecs;
    a.jobs = f.jobs; a.countPtr = p.dCtr + CTR_NJOBS; a.fetchCounter = p.dCtr + CTR_FETCH_CLOSEST;
    if (speculative) {
        /* needs the job list in pixel order, which holds right after pt_launch */
        dim3 grid(ctx->W / 8, ctx->H / 8), block(64);
        hipLaunchKernelGGL(k_trace_ref_speculative, grid, block, 0, p.stream, a, ctx->W, ctx->H);
        HIPCHK(hipGetLastError());
    } else if (ctx->optTraversal == 3 && !ctx->compact && ctx->boxesNest) {
        if (!p.dFlag) HIPCHK(hipMalloc((void**)&p.dFlag, (size_t)ctx->W * ctx->H * 4));
        uint32_t* flagCtr = p.dCtr + ctr_flag_base(ctx->slots) + 2 * depth;
        if (!overlapped) HIPCHK(hipMemsetAsync(flagCtr, 0, 2 * 4, p.stream));          /* overlapped: k_launch cleared them */
        if (int rc = launch_cert<MODE_JOBS>(ctx, a, false, p.stream, 2 * p.index, p.dFlag, flagCtr)) return rc;
    } else if (p.deferredShadow >= 0) {
        /* the shadow rays the last bounce's shade pass queued ride along */
        const int d0 = p.deferredShadow;
        a.shadowRays = p.dShadow[d0 & 1]; a.countPtr2 = p.dCtr + CTR_SHADOW + 2 * d0;
        if (int rc = launch_mixed(ctx, a, p.stream, 2 * p.index)) return rc;
    } else {
        if (int rc = launch_trace<MODE_JOBS, false>(ctx, a, false, p.stream, 2 * p.index)) return rc;
    }
    p.deferredShadow = -1;
    if (p.shadowBusy[q]) {         /* the shade pass rewrites queue q: the shadow pass of depth-2 must be through with it */
        HIPCHK(hipStreamWaitEvent(p.stream, p.evShadow[q], 0));
        p.shadowBusy[q] = false;
    }
    dim3 sgrid(((size_t)ctx->W * ctx->H + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK), sblock(PT_SHADE_BLOCK);
    hipLaunchKernelGGL(k_shade, sgrid, sblock, 0, p.stream, f, depth);
    hipLaunchKernelGGL(k_record_counts, dim3(1), dim3(1), 0, p.stream, f, depth);
    HIPCHK(hipGetLastError());
    if (overlapped) p.curJobs ^= 1;                                                         /* the shade pass wrote the next list */
    if (depth == 0) { HIPCHK(hipEventRecord(p.evShade0, p.stream)); p.shade0Pending = true; }

    TraceArgs s; fill_trace_args(ctx, s);
    s.stack = p.dStack; s.shadowRays = p.dShadow[q];
    s.countPtr = shadowCtr; s.fetchCounter = shadowCtr + 1;
    /* the SPECULATIVE build votes in its shadow walk too (the whole of trace.comp is compiled with it): one wavefront per
       8x8 tile over the tile-ordered queue the shade pass wrote */
    auto shadow_pass = [&](hipStream_t st, int area) -> int {
        if (speculative) {
            hipLaunchKernelGGL(k_trace_ref_speculative_shadow, dim3(ctx->W / 8, ctx->H / 8), dim3(64), 0, st, s, ctx->W, ctx->H);
            HIPCHK(hipGetLastError());
            return 0;
        }
        return launch_trace<MODE_SHADOW, true>(ctx, s, false, st, area);
    };
    /* merged passes: this bounce's shadow rays wait for the next bounce's closest-hit launch, unless there is none, or that
       launch is one of the kinds that cannot take them (certified walk, treelets, the reference layout) */
    const bool mergeable = ctx->optMergePasses && (ctx->optTraversal == 1 || ctx->optTraversal == 2);
    if (overlapped && mergeable && !speculative && depth + 1 < ctx->consts.numRays) {
        p.deferredShadow = (int)depth;
        return 0;
    }
    /* a side stream pays where the shadow pass is a launch of its own at every bounce; with merged passes only the last
       bounce's is, and one stream per frame leaves more hardware queues to the other frames in flight */
    const bool sideStream = ctx->optShadowStream == 2 ? !mergeable : ctx->optShadowStream != 0;
    if (overlapped && sideStream) {
        hipStream_t side = p.stream2;       /* (one side stream shared by all frames in flight was tried: 94 against 136 spp/s) */
        HIPCHK(hipEventRecord(p.evShade[q], p.stream));
        HIPCHK(hipStreamWaitEvent(side, p.evShade[q], 0));
        if (int rc = shadow_pass(side, 2 * p.index + 1)) return rc;
        HIPCHK(hipEventRecord(p.evShadow[q], side));
        p.shadowBusy[q] = true;
    } else {
        if (int rc = shadow_pass(p.stream, 2 * p.index)) return rc;
    }
    return 0;
}

PT_API int pt_trace(PtContext* ctx, uint32_t depth, int speculative)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (int rc = check_frame(ctx)) return rc;
    if (!ctx->dAux || !ctx->dLights) return ctx->fail(-2, "pt_trace: scene was uploaded without vertex attributes or lights");
    if (depth >= ctx->slots) return ctx->fail(-1, "pt_trace: depth exceeds the allocated slots");
    if (speculative && (ctx->W % 8 || ctx->H % 8)) return ctx->fail(-1, "pt_trace: the speculative build needs W and H to be multiples of 8");
    return trace_depth(ctx, ctx->pipe[0], depth, speculative, false);
}

PT_API int pt_compact(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (int rc = check_frame(ctx)) return rc;
    FramePipe& p = ctx->pipe[0];
    HIPCHK(hipMemsetAsync(p.dCtr + CTR_COMPACT_OUT, 0, 4, p.stream));
    FrameArgs f; fill_frame_args(ctx, p, f);
    dim3 grid(((size_t)ctx->W * ctx->H + 255) / 256), block(256);
    hipLaunchKernelGGL(k_compact, grid, block, 0, p.stream, f);
    hipLaunchKernelGGL(k_compact_finish, dim3(1), dim3(1), 0, p.stream, f);
    HIPCHK(hipGetLastError());
    p.curJobs ^= 1;
    return 0;
}

/* composite.frag on pipe p.  Accumulation adds happen in frame order: the pass waits for the resolve pass of the
 * frame before it when that ran on another pipe. */
static int resolve_on(PtContext* ctx, FramePipe& p)
{
    if (int rc = join_shadow(ctx, p)) return rc;
    FramePipe& prev = ctx->pipe[ctx->resolvePipe];
    if (&prev != &p && prev.donePending) HIPCHK(hipStreamWaitEvent(p.stream, prev.evDone, 0));
    FrameArgs f; fill_frame_args(ctx, p, f);
    dim3 grid(((size_t)ctx->W * ctx->H + 255) / 256), block(256);
    hipLaunchKernelGGL(k_resolve, grid, block, 0, p.stream, f);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(p.evDone, p.stream)); p.donePending = true;
    ctx->resolvePipe = p.index; ctx->accumEmpty = false;
    return 0;
}

PT_API int pt_resolve_accumulate(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (int rc = check_frame(ctx)) return rc;
    return resolve_on(ctx, ctx->pipe[0]);
}

/* One frame (PathTracer.cpp:729-838) queued on pipe p. */
static int frame_on(PtContext* ctx, FramePipe& p, uint32_t frame_index, int speculative_depth0)
{
    ctx->consts.frameIndex = frame_index & 0xFFFF;                                          /* PathTracer.cpp:729 */
    if (int rc = launch_on(ctx, p)) return rc;
    for (uint32_t d = 0; d < ctx->consts.numRays; d++)                                      /* PathTracer.cpp:780-812 */
        if (int rc = trace_depth(ctx, p, d, d == 0 && speculative_depth0, true)) return rc;
    return resolve_on(ctx, p);
}


/* Device memory the frame train needs for F stacks and `parts` bands. */
static uint64_t train_bytes(const PtContext* ctx, uint32_t F, int parts)
{
    const uint64_t px = (uint64_t)ctx->W * ctx->H;
    const uint64_t cap = (uint64_t)(F < ctx->slots ? F : ctx->slots) * (px + (uint64_t)parts * 8 * ctx->W);     /* bands round up to 8 rows */
    return (uint64_t)F * px * ctx->slots * sizeof(PtRayStackSlot) + cap * (2 * 4 + sizeof(PtHitRec) + 2 * sizeof(PtShadowRay));
}

/* Buffers of the frame train for F stacks in the ring and `parts` bands of rows.  Returns 1 if the train is not to be had -
 * the ring's slots do not fit a 32-bit index, the memory it needs exceeds PT_OPT_TRAIN_MAX_MB or the device's free memory -
 * and the caller renders frame by frame; the stack at the ring's head (the RNG states the next frame continues from, what
 * pt_readback_raystack returns) then moves to pipe 0, and a ring that is rebuilt keeps it. */
static int ensure_train(PtContext* ctx, uint32_t F, int parts)
{
    FrameTrain& t = ctx->train;
    const size_t px = (size_t)ctx->W * ctx->H, stride = px * ctx->slots;
    if ((uint32_t)parts * 8u > ctx->H) parts = 1;
    if (t.ready && t.F >= F && t.parts == parts) return 0;
    if (int rc = train_flush(ctx)) return rc;              /* frames a kept train holds are finished in the ring as it is */
    if (t.ready && F < t.F) F = t.F;                       /* a change of parts keeps the ring's size */
    for (TrainPart& p : t.part) if (p.stream) HIPCHK(hipStreamSynchronize(p.stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const bool carry = t.ready && (ctx->rngInTrain || ctx->lastInTrain);
    /* no train: what it holds that later frames need goes to pipe 0 (allocated since pt_resize), its buffers are freed */
    auto give_up = [&](PtRayStackSlot* head) -> int {
        FramePipe& p0 = ctx->pipe[0];
        if (head && p0.dStack) {
            HIPCHK(hipMemcpy(p0.dStack, head, stride * sizeof(PtRayStackSlot), hipMemcpyDeviceToDevice));
            ctx->rngPipe = ctx->lastPipe = 0; p0.shade0Pending = false;
        }
        free_train_buffers(t);
        ctx->rngInTrain = ctx->lastInTrain = false;
        return 1;
    };
    if ((uint64_t)F * stride >= 0xFFFFFFFFull || (ctx->optTrainMaxMB > 0 && train_bytes(ctx, F, parts) > (uint64_t)ctx->optTrainMaxMB << 20))
        return give_up(carry ? t.dRing + (size_t)t.head * stride : nullptr);
    if (!t.evFork) HIPCHK(hipEventCreateWithFlags(&t.evFork, hipEventDisableTiming));
    for (int k = 0; k < parts; k++) {
        TrainPart& p = t.part[k];
        if (p.stream) continue;
        if (k == 0) p.stream = ctx->stream;
        else HIPCHK(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&p.evDone, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p.evStagger, hipEventDisableTiming));
    }
    /* the head's stack survives in a buffer of its own while the ring is rebuilt */
    PtRayStackSlot* keep = nullptr;
    if (carry) {
        if (hipMalloc((void**)&keep, stride * sizeof(PtRayStackSlot)) != hipSuccess) { (void)hipGetLastError(); return give_up(t.dRing + (size_t)t.head * stride); }
        HIPCHK(hipMemcpy(keep, t.dRing + (size_t)t.head * stride, stride * sizeof(PtRayStackSlot), hipMemcpyDeviceToDevice));
    }
    free_train_buffers(t);
    bool short_of_memory = false;
    auto alloc = [&](void** ptr, size_t bytes) { if (!short_of_memory && hipMalloc(ptr, bytes) != hipSuccess) { (void)hipGetLastError(); *ptr = nullptr; short_of_memory = true; } };
    alloc((void**)&t.dRing, (size_t)F * stride * sizeof(PtRayStackSlot));
    alloc((void**)&t.dRayCounts, (size_t)2 * PT_MAX_TRAIN_DEPTH * 8);
    for (int k = 0; k < parts; k++) {
        TrainPart& p = t.part[k];
        /* bands of whole 8-row blocks (the launch pass works in 32 x 8 tiles) */
        const uint32_t blocks = (ctx->H + 7) / 8;
        p.row0 = (uint32_t)((uint64_t)blocks * k / parts) * 8; p.row1 = (uint32_t)((uint64_t)blocks * (k + 1) / parts) * 8;
        if (p.row1 > ctx->H) p.row1 = ctx->H;
        p.cap = (uint64_t)(F < ctx->slots ? F : ctx->slots) * (p.row1 - p.row0) * ctx->W;
        alloc((void**)&p.dJobs[0], p.cap * 4);
        alloc((void**)&p.dJobs[1], p.cap * 4);
        alloc((void**)&p.dHitRecs, p.cap * sizeof(PtHitRec));
        alloc((void**)&p.dShadow[0], p.cap * sizeof(PtShadowRay));
        alloc((void**)&p.dShadow[1], p.cap * sizeof(PtShadowRay));
        alloc((void**)&p.dCtr, (size_t)TC_WORDS * 4);
    }
    if (short_of_memory) {                                  /* the device cannot hold the train: frame by frame, then */
        const int rc = give_up(keep);
        dfree(keep);
        return rc;
    }
    t.F = F; t.parts = parts; t.head = 0;
    /* queued on the context's stream, which every part's stream joins before its first step (see ensure_pipe) */
    HIPCHK(hipMemsetAsync(t.dRing, 0, (size_t)F * stride * sizeof(PtRayStackSlot), ctx->stream));
    HIPCHK(hipMemsetAsync(t.dRayCounts, 0, (size_t)2 * PT_MAX_TRAIN_DEPTH * 8, ctx->stream));
    if (carry) HIPCHK(hipMemcpyAsync(t.dRing, keep, stride * sizeof(PtRayStackSlot), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dfree(keep);
    t.ready = true;
    return 0;
}

/* The parts' streams take over from the context's stream: whatever is queued there comes first. */
static int train_fork(PtContext* ctx)
{
    FrameTrain& t = ctx->train;
    if (t.forked) return 0;
    HIPCHK(hipEventRecord(t.evFork, ctx->stream));
    for (int k = 1; k < t.parts; k++) HIPCHK(hipStreamWaitEvent(t.part[k].stream, t.evFork, 0));
    t.forked = t.stagger = true;
    return 0;
}

/* The context's stream waits for everything the parts have queued: what is queued on it next sees every step so far finished. */
static int train_join(PtContext* ctx)
{
    FrameTrain& t = ctx->train;
    if (!t.forked) return 0;
    for (int k = 1; k < t.parts; k++) {
        HIPCHK(hipEventRecord(t.part[k].evDone, t.part[k].stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, t.part[k].evDone, 0));
    }
    t.forked = false;
    return 0;
}

/* Before the first step of a train that holds no frame: clean counters, lists at their start. */
static int train_begin(PtContext* ctx)
{
    FrameTrain& t = ctx->train;
    if (int rc = ensure_overflow(ctx)) return rc;
    if (ctx->optTraversal == 3 && !ctx->compact && ctx->boxesNest)      /* the list of jobs for the second walk: allocated before anything is queued */
        for (int k = 0; k < t.parts; k++) if (!t.part[k].dFlag) HIPCHK(hipMalloc((void**)&t.part[k].dFlag, t.part[k].cap * 4));
    if (t.live) return train_fork(ctx);
    if (!ctx->rngInTrain) {      /* the most recent frame's RNG states are in a pipe's stack: the first launch pass reads them there */
        FramePipe& src = ctx->pipe[ctx->rngPipe];
        if (src.stream && src.stream != ctx->stream && src.shade0Pending && !ctx->optReseed) HIPCHK(hipStreamWaitEvent(ctx->stream, src.evShade0, 0));
    }
    if (int rc = train_fork(ctx)) return rc;
    for (int k = 0; k < t.parts; k++) HIPCHK(hipMemsetAsync(t.part[k].dCtr, 0, (size_t)TC_WORDS * 4, t.part[k].stream));
    t.cur = t.q = 0; t.stepNo = 0; t.flight.clear();
    t.live = true;
    return 0;
}

/* One step of the train: a frame enters (or none does, when the train is being emptied), every frame in flight goes one
 * bounce further, the frame that entered numRays steps ago leaves. */
static int train_step(PtContext* ctx, bool enter, uint32_t frame_index)
{
    FrameTrain& t = ctx->train;
    const uint32_t nr = ctx->consts.numRays;
    const size_t stride = (size_t)ctx->W * ctx->H * ctx->slots;
    const bool certified = ctx->optTraversal == 3 && !ctx->compact && ctx->boxesNest;
    if (int rc = train_fork(ctx)) return rc;
    uint32_t pos = 0; const PtRayStackSlot* rngStack = nullptr; bool seed = false;
    if (enter) {
        rngStack = ctx->rngInTrain ? t.dRing + (size_t)t.head * stride : ctx->pipe[ctx->rngPipe].dStack;
        pos = (t.head + 1u) % t.F;
        seed = ctx->accumEmpty;                                                             /* launch.comp:49-50: this frame's resolve pass is the first since the clear */
        ctx->consts.frameIndex = frame_index & 0xFFFF;                                      /* PathTracer.cpp:729 */
        t.flight.push_back({ pos, t.stepNo });
    }
    /* the frame that leaves in this step, and how many have an entry in its job list */
    const bool leaves = !t.flight.empty() && t.flight.front().second + nr == t.stepNo;
    const uint32_t leavingPos = leaves ? t.flight.front().first : 0u;
    if (leaves) t.flight.pop_front();
    const uint32_t shaded = (uint32_t)t.flight.size();
    FrameArgs f; fill_frame_args(ctx, ctx->pipe[0], f);
    f.rayCounts = t.dRayCounts; f.fuseCompact = 1;
    const int cur = t.cur, q = t.q;
    for (int k = 0; k < t.parts; k++) {
        TrainPart& p = t.part[k];
        const uint32_t ppx = (p.row1 - p.row0) * ctx->W;
        f.ctr = p.dCtr; f.hitRecs = p.dHitRecs; f.row0 = p.row0; f.row1 = p.row1;
        if (enter) {
            f.c = ctx->consts;
            f.seedFromJitter = seed ? 1u : 0u;
            f.stack = t.dRing + (size_t)pos * stride; f.slotBase = (uint32_t)(pos * stride);
            f.rngStack = rngStack;
            f.jobs = p.dJobs[cur];
            dim3 grid((ctx->W + 31) / 32, (p.row1 - p.row0 + 7) / 8), block(256);
            hipLaunchKernelGGL(k_launch_t<true>, grid, block, 0, p.stream, f);
        }
        {   /* every frame in flight one bounce further, and the shadow rays of the bounce before */
            TraceArgs a; fill_trace_args(ctx, a);
            a.stack = t.dRing; a.hitRecs = p.dHitRecs;
            a.jobs = p.dJobs[cur]; a.countPtr = p.dCtr + TC_NJOBS; a.fetchCounter = p.dCtr + TC_FETCH;
            a.shadowRays = p.dShadow[q]; a.countPtr2 = p.dCtr + TC_NSHADOW + q;
            if (certified) {
                if (int rc = launch_cert_mixed(ctx, a, p.stream, k, p.dFlag, p.dCtr + TC_FLAG)) return rc;
            } else if (int rc = launch_mixed(ctx, a, p.stream, k, true)) return rc;
        }
        /* parts that start from rest start one after the other, so that their traversal launches do not all thin out at the same time */
        if (t.stagger && k + 1 < t.parts) {
            HIPCHK(hipEventRecord(p.evStagger, p.stream));
            HIPCHK(hipStreamWaitEvent(t.part[k + 1].stream, p.evStagger, 0));
        }
        if (leaves) {                                                                     /* its last shadow rays were in that launch */
            f.stack = t.dRing + (size_t)leavingPos * stride;
            hipLaunchKernelGGL(k_resolve, dim3(((size_t)ppx + 255) / 256), dim3(256), 0, p.stream, f);
        }
        if (shaded) {
            f.stack = t.dRing;
            f.jobs = p.dJobs[cur]; f.jobsOut = p.dJobs[cur ^ 1];
            f.shadowRays = p.dShadow[q ^ 1]; f.shadowCtr = p.dCtr + TC_NSHADOW + (q ^ 1);
            const uint64_t most = (uint64_t)shaded * ppx;
            uint64_t blocks = (most + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK;
            const uint64_t wide = (uint64_t)ctx->numCUs * 32;
            if (blocks > wide) blocks = wide;
            hipLaunchKernelGGL(k_shade_train, dim3((uint32_t)blocks), dim3(PT_SHADE_BLOCK), 0, p.stream, f);
        }
        f.shadowQueue = (uint32_t)q;
        hipLaunchKernelGGL(k_train_step_end, dim3(1), dim3(1), 0, p.stream, f);
        HIPCHK(hipGetLastError());
    }
    t.stagger = false;
    t.cur ^= 1; t.q ^= 1; t.stepNo++;
    if (enter) {
        t.head = pos;
        ctx->accumEmpty = false;
        ctx->rngInTrain = ctx->lastInTrain = true;
    }
    return 0;
}

/* Steps without a new frame until every frame in the train has left it; then the context's stream has everything. */
static int train_flush(PtContext* ctx)
{
    FrameTrain& t = ctx->train;
    if (!t.live) return 0;
    while (!t.flight.empty()) if (int rc = train_step(ctx, false, 0)) return rc;
    t.live = false;
    return train_join(ctx);
}

/* What every entry point that reads or changes more than the images calls first: with PT_OPT_TRAIN_KEEP a render call
 * returns with frames still in the train. */
static int settle(PtContext* ctx)
{
    if (!ctx->train.live) return 0;
    if (int rc = enter(ctx)) return rc;
    return train_flush(ctx);
}

/* num_frames frames through the frame train.  keep: the call ends after num_frames steps, with the last numRays frames still
 * in the train (a later call carries on from there; train_flush finishes them). */
static int train_render(PtContext* ctx, uint32_t first_frame_index, uint32_t num_frames, uint64_t* ray_counts, bool keep)
{
    FrameTrain& t = ctx->train;
    const uint32_t nr = ctx->consts.numRays;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    if (ray_counts) HIPCHK(hipMemsetAsync(t.dRayCounts, 0, (size_t)2 * PT_MAX_TRAIN_DEPTH * 8, ctx->stream));
    if (int rc = train_begin(ctx)) return rc;
    for (uint32_t i = 0; i < num_frames; i++) if (int rc = train_step(ctx, true, first_frame_index + i)) return rc;
    if (!keep) { if (int rc = train_flush(ctx)) return rc; }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->evPending = true;
    if (ray_counts) {
        std::vector<unsigned long long> part((size_t)2 * PT_MAX_TRAIN_DEPTH);
        HIPCHK(hipMemcpyAsync(part.data(), t.dRayCounts, part.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (uint32_t j = 0; j < 2 * nr; j++) ray_counts[j] = part[j];
    }
    return 0;
}

PT_API int pt_render_samples(PtContext* ctx, uint32_t first_frame_index, uint32_t num_frames, int speculative_depth0, uint64_t* ray_counts)
try {
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = check_frame(ctx)) return rc;
    if (!ctx->dAux || !ctx->dLights) return ctx->fail(-2, "pt_render_sample: scene was uploaded without vertex attributes or lights");
    if (speculative_depth0 && (ctx->W % 8 || ctx->H % 8)) return ctx->fail(-1, "pt_render_sample: the speculative build needs W and H to be multiples of 8");
    if (num_frames == 0) return 0;
    /* the frame train serves whatever its one traversal launch per step can: near-first and reference-order walks on 64 B
       or 48 B records.  (The stale-slot behaviour needs every frame on the one ray stack; the voting build its own launches.) */
    /* (2 = auto, the default: the train where it pays.  On a small scene with one bounce or none the per-frame pipeline's three
       frames in flight do better: 1 416 against 1 308 spp/s on the bunny at numRays = 2, 1 962 against 1 686 at 1; on 262 k
       triangles the train is ahead at any depth: 372 against 352 and 644 against 561, on 5.8 M 319 against 294 and 507 against 432) */
    const bool wantTrain = ctx->optFrameTrain == 1 || (ctx->optFrameTrain == 2 && (ctx->consts.numRays >= 3 || !small_scene(ctx)));
    if (wantTrain && !ctx->optStale && !speculative_depth0 && ctx->optTraversal != 0 && ctx->consts.numRays <= PT_MAX_TRAIN_DEPTH) {
        /* ray counts are those of this call's frames: nothing of an earlier call in the train, nothing of this one left in it */
        const bool keep = ctx->optTrainKeep && !ray_counts;
        if (!keep) if (int rc = settle(ctx)) return rc;
        const uint32_t want = !keep && num_frames < ctx->consts.numRays + 1 ? num_frames : ctx->consts.numRays + 1;
        const int rc = ensure_train(ctx, want, ctx->optTrainParts);
        if (rc < 0) return rc;
        if (rc == 0) return train_render(ctx, first_frame_index, num_frames, ray_counts, keep);
    }
    if (int rc = settle(ctx)) return rc;
    /* the reference's stale-slot behaviour needs every frame on the one ray stack */
    int K = ctx->optStale ? 1 : ctx->optFramesInFlight;
    if ((uint32_t)K > num_frames) K = (int)num_frames;
    for (int k = 0; k < K; k++) if (int rc = ensure_pipe(ctx, k)) return rc;
    if (int rc = ensure_overflow(ctx)) return rc;
    const uint32_t nr = ctx->consts.numRays;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    if (K > 1) {       /* the other pipes start after whatever is already queued on the context's stream */
        HIPCHK(hipEventRecord(ctx->evFork, ctx->stream));
        for (int k = 1; k < K; k++) HIPCHK(hipStreamWaitEvent(ctx->pipe[k].stream, ctx->evFork, 0));
    }
    if (ray_counts) for (int k = 0; k < K; k++) HIPCHK(hipMemsetAsync(ctx->pipe[k].dRayCounts, 0, (size_t)2 * ctx->slots * 8, ctx->pipe[k].stream));
    for (uint32_t i = 0; i < num_frames; i++)
        if (int rc = frame_on(ctx, ctx->pipe[i % (uint32_t)K], first_frame_index + i, speculative_depth0)) return rc;
    for (int k = 1; k < K; k++) {      /* join: later calls on the context's stream see every frame finished */
        FramePipe& p = ctx->pipe[k];
        if (p.donePending) { HIPCHK(hipStreamWaitEvent(ctx->stream, p.evDone, 0)); p.donePending = false; }
        p.shade0Pending = false;
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->evPending = true;
    if (ray_counts) {
        std::vector<uint64_t> part((size_t)2 * nr);
        memset(ray_counts, 0, (size_t)2 * nr * 8);
        for (int k = 0; k < K; k++) {
            HIPCHK(hipMemcpyAsync(part.data(), ctx->pipe[k].dRayCounts, (size_t)2 * nr * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            for (uint32_t j = 0; j < 2 * nr; j++) ray_counts[j] += part[j];
        }
    }
    return 0;
} catch (...) { return ctx->fail(-11, "pt_render_samples: out of host memory"); }

PT_API int pt_render_sample(PtContext* ctx, uint32_t frame_index, int speculative_depth0, uint64_t* ray_counts)
{
    return pt_render_samples(ctx, frame_index, 1, speculative_depth0, ray_counts);
}

PT_API int pt_render_flush(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    return settle(ctx);
}

PT_API int pt_readback_current(PtContext* ctx, uint16_t* rgba16f)
{
    if (!ctx || !rgba16f) return -1;
    if (int rc = enter(ctx)) return rc;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    if (int rc = train_join(ctx)) return rc;           /* with a kept train: the frames resolved so far */
    HIPCHK(hipMemcpyAsync(rgba16f, ctx->dCurrent, (size_t)ctx->W * ctx->H * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API int pt_readback_accumulation(PtContext* ctx, float* rgba32f)
{
    if (!ctx || !rgba32f) return -1;
    if (int rc = enter(ctx)) return rc;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    if (int rc = train_join(ctx)) return rc;           /* with a kept train: the frames resolved so far */
    HIPCHK(hipMemcpyAsync(rgba32f, ctx->dAccum, (size_t)ctx->W * ctx->H * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API int pt_readback_raystack(PtContext* ctx, PtRayStackSlot* slots, PtRayJob* jobs)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    size_t px = (size_t)ctx->W * ctx->H;
    FramePipe& p = ctx->pipe[ctx->lastPipe];           /* the stack and job list of the most recent frame */
    if (ctx->lastInTrain) {
        /* a frame the train rendered: its stack is the ring's head; its job list ended empty */
        if (slots) HIPCHK(hipMemcpyAsync(slots, ctx->train.dRing + (size_t)ctx->train.head * px * ctx->slots, px * ctx->slots * sizeof(PtRayStackSlot), hipMemcpyDeviceToHost, ctx->stream));
        if (jobs) memset(jobs, 0xFF, px * 4);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return 0;
    }
    if (slots) HIPCHK(hipMemcpyAsync(slots, p.dStack, px * ctx->slots * sizeof(PtRayStackSlot), hipMemcpyDeviceToHost, ctx->stream));
    if (jobs) HIPCHK(hipMemcpyAsync(jobs, p.dJobs[p.curJobs], px * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API int pt_upload_raystack(PtContext* ctx, const PtRayStackSlot* slots, const PtRayJob* jobs)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    size_t px = (size_t)ctx->W * ctx->H;
    FramePipe& p = ctx->pipe[0];                        /* the dispatch-site calls that follow work on pipe 0 */
    if (slots) { HIPCHK(hipMemcpyAsync(p.dStack, slots, px * ctx->slots * sizeof(PtRayStackSlot), hipMemcpyHostToDevice, ctx->stream)); ctx->rngPipe = 0; ctx->rngInTrain = false; }
    if (jobs) {
        /* a job is a slot index the kernels follow: one past the ray stack would be read and written on the device */
        const uint64_t limit = (uint64_t)px * ctx->slots;
        for (size_t i = 0; i < px; i++)
            if (jobs[i].index != PT_JOB_DEAD && jobs[i].index >= limit) return ctx->fail(-1, "pt_upload_raystack: a job names a slot outside the ray stack");
        uint32_t n = (uint32_t)px;
        HIPCHK(hipMemcpyAsync(p.dJobs[p.curJobs], jobs, px * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(p.dCtr + CTR_NJOBS, &n, 4, hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->lastPipe = 0; ctx->lastInTrain = false;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API void* pt_accumulation_device_ptr(PtContext* ctx) { return ctx ? (void*)ctx->dAccum : nullptr; }

PT_API int pt_dump_current(PtContext* ctx, const char* dir, uint32_t index)
try {
    if (!ctx) return -1;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    std::vector<uint16_t> img((size_t)ctx->W * ctx->H * 4);
    if (int rc = pt_readback_current(ctx, img.data())) return rc;
    int rc = pt_frame_dump_write(dir, index, ctx->W, ctx->H, img.data());
    if (rc) return ctx->fail(rc, std::string("pt_dump_current: cannot write the dump pair under ") + (dir ? dir : "."));
    return 0;
} catch (...) { return ctx->fail(-11, "pt_dump_current: out of host memory"); }

PT_API int pt_tonemap(PtContext* ctx, float* rgb32f, uint8_t* rgba8)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    if (!rgb32f && !rgba8) return 0;
    if (int rc = train_join(ctx)) return rc;           /* with a kept train: the frames resolved so far */
    const size_t px = (size_t)ctx->W * ctx->H;
    float* dF = nullptr; uint32_t* dB = nullptr;
    if (rgb32f) HIPCHK(hipMalloc((void**)&dF, px * 12));
    if (rgba8 && hipMalloc((void**)&dB, px * 4) != hipSuccess) { dfree(dF); return ctx->fail(-10, "pt_tonemap: out of device memory"); }
    hipLaunchKernelGGL(k_tonemap, dim3((px + 255) / 256), dim3(256), 0, ctx->stream, ctx->dAccum, (uint32_t)px, dF, dB);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && rgb32f) e = hipMemcpyAsync(rgb32f, dF, px * 12, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rgba8) e = hipMemcpyAsync(rgba8, dB, px * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    dfree(dF); dfree(dB);
    if (e != hipSuccess) return ctx->fail(-10, std::string("pt_tonemap: ") + hipGetErrorString(e));
    return 0;
}

/* One batch through the ordering pass and the traversal kernels.  `sc` names the scratch: the context's own buffers
 * (grown here when they are too small) or, with sc.maxBlocks set, fixed parts of the static scratch array, in which
 * case nothing is allocated on the way. */
static int trace_batch(PtContext* ctx, const PtRay* d_rays, uint64_t num_rays, int any_hit, PtHit* d_hits, uint64_t* stats, BatchScratch& sc)
{
    const bool fixed = sc.maxBlocks != 0;
    const bool sort = ctx->optTraversal != 0 &&        /* the reference-layout kernel takes its rays in batch order */
                      (ctx->optSortRays == 1 || (ctx->optSortRays == 2 && num_rays >= (1u << 20)));
    const bool certified = ctx->optTraversal == 3 && !any_hit && !ctx->compact && ctx->boxesNest;
    if (!fixed) {
        sc.fetch = ctx->dFetch;
        if (sort) {
            if (ctx->sortCapacity < num_rays) {
                HIPCHK(hipStreamSynchronize(ctx->stream));
                dfree(ctx->dSortOrder); dfree(ctx->dSortKeys); ctx->sortCapacity = 0;
                HIPCHK(hipMalloc((void**)&ctx->dSortOrder, num_rays * 4));
                HIPCHK(hipMalloc((void**)&ctx->dSortKeys, num_rays * 2));
                ctx->sortCapacity = num_rays;
            }
            if (!ctx->dSortBins) {
                HIPCHK(hipMalloc((void**)&ctx->dSortBins, 2 * PT_SORT_BINS * 4));
                HIPCHK(hipMemsetAsync(ctx->dSortBins, 0, 2 * PT_SORT_BINS * 4, ctx->stream));
            }
            sc.sortOrder = ctx->dSortOrder; sc.sortKeys = ctx->dSortKeys; sc.sortBins = ctx->dSortBins;
        }
        if (certified) {
            if (ctx->flagCapacity < num_rays) {
                HIPCHK(hipStreamSynchronize(ctx->stream));
                dfree(ctx->dFlagList); ctx->flagCapacity = 0;
                HIPCHK(hipMalloc((void**)&ctx->dFlagList, num_rays * 4));
                ctx->flagCapacity = num_rays;
            }
            if (!ctx->dFlagCtr) HIPCHK(hipMalloc((void**)&ctx->dFlagCtr, 2 * 4));
            sc.flagList = ctx->dFlagList; sc.flagCtr = ctx->dFlagCtr;
        }
    }
    HIPCHK(hipMemsetAsync(sc.fetch, 0, 4, ctx->stream));
    if (certified) HIPCHK(hipMemsetAsync(sc.flagCtr, 0, 2 * 4, ctx->stream));
    if (stats) HIPCHK(hipMemsetAsync(ctx->dStats, 0, 32, ctx->stream));
    TraceArgs a; fill_trace_args(ctx, a);
    a.rays = d_rays; a.hits = d_hits; a.count = num_rays; a.countPtr = nullptr; a.fetchCounter = sc.fetch;
    /* large batches are first ordered by the cell of the scene box their origin lies in (sort_kernels.h) */
    if (sort) {
        const float* b = ctx->rootBox;
        const float cells = (float)(1 << ctx->optSortBits);
        const int nbins = 1 << (3 * ctx->optSortBits);
        auto scale = [cells](float lo, float hi) { float d = hi - lo; return d > 0.0f ? cells / d : 0.0f; };
        const uint32_t tiles = (uint32_t)((num_rays + PT_SORT_TILE - 1) / PT_SORT_TILE);
        hipLaunchKernelGGL(k_bin_count, dim3(tiles), dim3(PT_SORT_BLOCK), 0, ctx->stream, d_rays, num_rays, sc.sortKeys, sc.sortBins, nbins,
                           b[0], b[1], b[2], scale(b[0], b[3]), scale(b[1], b[4]), scale(b[2], b[5]), cells - 1.0f);
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(PT_SORT_BLOCK), 0, ctx->stream, sc.sortBins, sc.sortBins + PT_SORT_BINS);
        hipLaunchKernelGGL(k_bin_scatter, dim3(tiles), dim3(PT_SORT_BLOCK), 0, ctx->stream, num_rays, sc.sortKeys,
                           sc.sortBins + PT_SORT_BINS, nbins, sc.sortOrder);
        HIPCHK(hipGetLastError());
        a.order = sc.sortOrder;
    }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    int rc = 0;
    const BatchScratch* fx = fixed ? &sc : nullptr;
    if (certified) {
        rc = launch_cert<MODE_RAYS>(ctx, a, stats != nullptr, ctx->stream, 0, sc.flagList, sc.flagCtr, fx);
    } else {
        rc = any_hit ? launch_trace<MODE_RAYS, true>(ctx, a, stats != nullptr, ctx->stream, 0, fx) : launch_trace<MODE_RAYS, false>(ctx, a, stats != nullptr, ctx->stream, 0, fx);
    }
    if (rc) return rc;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->evPending = true;
    if (stats) {
        HIPCHK(hipMemcpyAsync(stats, ctx->dStats, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

PT_API int pt_trace_rays(PtContext* ctx, const PtRay* d_rays, uint64_t num_rays, int any_hit, PtHit* d_hits, uint64_t* stats)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (int rc = check_scene(ctx)) return rc;
    if (!d_rays || !d_hits) return ctx->fail(-1, "pt_trace_rays: null buffer");
    /* the id counter is 32 bits wide and every wavefront may overshoot the count by one chunk */
    if (num_rays > 0xFE000000ull) return ctx->fail(-1, "pt_trace_rays: at most 2^32-2^25 rays per call");
    if (num_rays == 0) return 0;
    BatchScratch sc;
    return trace_batch(ctx, d_rays, num_rays, any_hit, d_hits, stats, sc);
}

PT_API int pt_trace_rays_noalloc(PtContext* ctx, const PtRay* d_rays, uint64_t num_rays, int any_hit, PtHit* d_hits)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (int rc = check_scene(ctx)) return rc;
    if (!d_rays || !d_hits) return ctx->fail(-1, "pt_trace_rays_noalloc: null buffer");
    if (num_rays > NA_MAX_RAYS) return ctx->fail(-1, "pt_trace_rays_noalloc: at most 2^21 (2 097 152) rays per call; larger batches need pt_trace_rays");
    if (num_rays == 0) return 0;
    unsigned char* base = ctx->scratchPtr;
    if (!base) return ctx->fail(-2, "pt_trace_rays_noalloc: the scratch array's device address was not resolved at pt_ctx_create (" + ctx->scratchErr + ")");
    BatchScratch sc;
    /* the spill area holds (tree depth - LDS part) entries per lane of the launch: the launch takes as many workgroups
       as fit, up to its usual number; a tree too deep for even one workgroup per CU is refused */
    sc.spillEntries = spill_entries(ctx);
    sc.maxBlocks = ctx->numCUs * 8;
    if (sc.spillEntries) {
        const size_t perBlock = (size_t)sc.spillEntries * PT_BLOCK * sizeof(uint2);
        const size_t fit = NA_SPILL_BYTES / perBlock;
        if (fit < (size_t)ctx->numCUs)
            return ctx->fail(-1, "pt_trace_rays_noalloc: the tree is too deep for the 32 MiB spill area of the static scratch (use pt_trace_rays)");
        if (fit < (size_t)sc.maxBlocks) sc.maxBlocks = (int)fit;
    }
    sc.fetch = (uint32_t*)(base + NA_OFF_CTR);
    sc.flagCtr = sc.fetch + 1;
    sc.sortBins = (uint32_t*)(base + NA_OFF_BINS);
    sc.sortOrder = (uint32_t*)(base + NA_OFF_ORDER);
    sc.sortKeys = (uint16_t*)(base + NA_OFF_KEYS);
    sc.flagList = (uint32_t*)(base + NA_OFF_FLAGS);
    sc.spill = (uint2*)(base + NA_OFF_SPILL);
    /* One array per GPU, any number of contexts: this call's work starts when the last call's work on this GPU has ended,
       whatever stream that ran on, and the host side of the hand-over (wait, queue, record) is one critical section. */
    std::lock_guard<std::mutex> lk(g_scratchMutex);
    hipEvent_t turn = g_scratchEvent[ctx->device];
    HIPCHK(hipStreamWaitEvent(ctx->stream, turn, 0));          /* an event never recorded holds nothing up */
    /* the array keeps what the last call on this GPU left: counters and bin counts start from zero (the rest is
       written before it is read) */
    HIPCHK(hipMemsetAsync(base + NA_OFF_CTR, 0, NA_OFF_ORDER - NA_OFF_CTR, ctx->stream));
    const int rc = trace_batch(ctx, d_rays, num_rays, any_hit, d_hits, nullptr, sc);
    /* recorded whether or not every launch was queued: what was queued must end before the next call starts */
    hipError_t e = hipEventRecord(turn, ctx->stream);
    if (rc) return rc;
    if (e != hipSuccess) return ctx->fail(-10, std::string("hipEventRecord(scratch turn): ") + hipGetErrorString(e));
    return 0;
}

PT_API int pt_trace_rays_host(PtContext* ctx, const PtRay* rays, uint64_t n, int any_hit, PtHit* hits, uint64_t* stats)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (!rays || !hits) return ctx->fail(-1, "pt_trace_rays_host: null buffer");
    if (n == 0) return 0;
    PtRay* dr = nullptr; PtHit* dh = nullptr;
    HIPCHK(hipMalloc((void**)&dr, n * sizeof(PtRay)));
    hipError_t e = hipMalloc((void**)&dh, n * sizeof(PtHit));
    if (e != hipSuccess) { (void)hipFree(dr); return ctx->fail(-10, hipGetErrorString(e)); }
    int rc = 0;
    if ((e = hipMemcpyAsync(dr, rays, n * sizeof(PtRay), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) rc = ctx->fail(-10, hipGetErrorString(e));
    if (!rc) rc = pt_trace_rays(ctx, dr, n, any_hit, dh, stats);
    if (!rc && (e = hipMemcpyAsync(hits, dh, n * sizeof(PtHit), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) rc = ctx->fail(-10, hipGetErrorString(e));
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = ctx->fail(-10, hipGetErrorString(e));
    (void)hipFree(dr); (void)hipFree(dh);
    return rc;
}

PT_API int pt_generate_rays(PtContext* ctx, uint64_t seed, uint64_t first, uint64_t n, PtRay* d_rays)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = check_scene(ctx)) return rc;
    if (!d_rays) return ctx->fail(-1, "pt_generate_rays: null buffer");
    if (n == 0) return 0;
    if (n > 0xFFFFFF00ull) return ctx->fail(-1, "pt_generate_rays: at most 2^32-256 rays per call");
    dim3 grid((uint32_t)((n + 255) / 256)), block(256);
    const float* b = ctx->rootBox;
    hipLaunchKernelGGL(k_generate_rays, grid, block, 0, ctx->stream, d_rays, seed, first, n, b[0], b[1], b[2], b[3], b[4], b[5]);
    HIPCHK(hipGetLastError());
    return 0;
}

PT_API void* pt_device_alloc(PtContext* ctx, size_t bytes)
{
    if (!ctx || enter(ctx)) return nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { ctx->fail(-10, std::string("hipMalloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}
PT_API int pt_device_free(PtContext* ctx, void* p)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(p));
    return 0;
}
PT_API int pt_copy_to_host(PtContext* ctx, void* dst, const void* d_src, size_t bytes)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API int pt_copy_to_device(PtContext* ctx, void* d_dst, const void* src, size_t bytes)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    HIPCHK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

PT_API int pt_selftest_entry_bound(PtContext* ctx, uint64_t seed, uint64_t num_cases, uint64_t* checked, uint64_t* violations)
{
    if (!ctx || !checked || !violations) return -1;
    if (int rc = enter(ctx)) return rc;
    if (num_cases == 0 || num_cases > 0xFFFFFF00ull) return ctx->fail(-1, "pt_selftest_entry_bound: 1 .. 2^32-256 cases");
    HIPCHK(hipMemsetAsync(ctx->dStats, 0, 16, ctx->stream));
    hipLaunchKernelGGL(k_selftest_entry_bound, dim3((uint32_t)((num_cases + 255) / 256)), dim3(256), 0, ctx->stream, seed, num_cases, ctx->dStats);
    HIPCHK(hipGetLastError());
    unsigned long long out[2] = { 0, 0 };
    HIPCHK(hipMemcpyAsync(out, ctx->dStats, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *checked = out[0]; *violations = out[1];
    return 0;
}

PT_API float pt_last_kernel_ms(PtContext* ctx)
{
    if (!ctx) return -1.0f;
    if (ctx->evPending) {
        if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
        ctx->lastMs = ms; ctx->evPending = false;
    }
    return ctx->lastMs;
}

PT_API int pt_comm_unique_id(uint8_t unique_id[128])
{
    std::string err;
    if (!unique_id) return -1;
    if (!load_nccl(err)) { g_createError = err; return -20; }
    NcclId id; memset(&id, 0, sizeof(id));
    int rc = g_nccl.GetUniqueId(&id);
    if (rc != 0) { g_createError = std::string("ncclGetUniqueId: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "error"); return -21; }
    memcpy(unique_id, &id, 128);
    return 0;
}

PT_API int pt_comm_init(PtContext* ctx, int rank, int num_ranks, const uint8_t unique_id[128])
{
    if (!ctx || !unique_id) return -1;
    if (rank < 0 || rank >= num_ranks) return ctx->fail(-1, "pt_comm_init: rank out of range");
    std::string err;
    if (!load_nccl(err)) return ctx->fail(-20, err);
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->comm) { g_nccl.CommDestroy(ctx->comm); ctx->comm = nullptr; }
    NcclId id; memcpy(&id, unique_id, 128);
    int rc = g_nccl.CommInitRank(&ctx->comm, num_ranks, id, rank);
    if (rc != 0) return ctx->fail(-21, std::string("ncclCommInitRank: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "error"));
    return 0;
}

PT_API int pt_allreduce_accumulation(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (!ctx->comm) return ctx->fail(-2, "pt_allreduce_accumulation: call pt_comm_init first");
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    /* one sum over RGBA32F: .w carries the sample count and sums with the radiance */
    int rc = g_nccl.AllReduce(ctx->dAccum, ctx->dAccum, (size_t)ctx->W * ctx->H * 4, /*ncclFloat32*/ 7, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    if (rc != 0) return ctx->fail(-21, std::string("ncclAllReduce: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "error"));
    return 0;
}

}  // extern "C"
