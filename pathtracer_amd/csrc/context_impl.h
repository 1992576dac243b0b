/*
 * context_impl.h - the context behind include/ptcore.h: RCCL entry table, the per-frame and frame-train resources, PtContext,
 * their release, launch shapes of the persistent traversal kernels, spill areas of the LDS stacks, argument blocks.
 * A part of ptcore.hip's one translation unit (included there, in order); not a header for anyone else.
 */
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
    PtSpineBox* dSpine = nullptr; uint32_t spineCount = 0;      /* the path from the last node up to the root's child (certified walk) */

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
    int optTraversal = 3, optStale = 0, optReseed = 0, optBlocksPerCU = 5, optRefillMin = 36, optClaimChunk = 0, optFramesInFlight = 3, optRecordLayout = 0, optSortRays = 2, optSortBits = 4, optSecondWalk = 2, optShadowStream = 2, optMergePasses = 1, optFrameTrain = 2, optTrainParts = 2, optTrainMaxMB = 0, optTrainKeep = 0;
    bool userBlocks = false, userRefill = false;      /* PT_OPT_PERSIST_BLOCKS / PT_OPT_REFILL_MIN were set by the caller */
    float lastMs = 0.0f; bool evPending = false;
    /* PT_OPT_TIME_TRAVERSAL: an event pair around every traversal launch of the frame train, read by pt_traversal_ms */
    int optTimeTraversal = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> travEvents; size_t travUsed = 0;
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

unsigned char* resolve_scratch(int device, std::string& err);       /* batch_impl.h: device address of the no-alloc scratch array */

template <class T> void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

void free_scene(PtContext* c)
{
    dfree(c->dPositions); dfree(c->dAux); dfree(c->dIndices); dfree(c->dNodes); dfree(c->dLights); dfree(c->dBlueNoise);
    dfree(c->dRecords); dfree(c->dOverflow); dfree(c->dLeafRef); dfree(c->dTreelets); dfree(c->dSpine); c->spineCount = 0;
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
    uint32_t inLds = PT_CERT_STACK_LDS < PT_STACK_LDS ? PT_CERT_STACK_LDS : PT_STACK_LDS;      /* the kernel that keeps the fewest entries in LDS spills the most */
    if (PT_CERT_STACK_LDS_RAYS < inLds) inLds = PT_CERT_STACK_LDS_RAYS;
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
    a.spine = c->dSpine; a.spineCount = c->spineCount;
    a.rootRef = (c->T == 1) ? PT_LEAF_BIT : 0u;
    a.overflow = c->dOverflow; a.overflowEntries = c->overflowEntries;
    a.refillMin = (uint32_t)c->optRefillMin; a.claimChunk = (uint32_t)c->optClaimChunk;
    a.stats = c->dStats;
}

/* Every entry point that touches the device starts here: allocations go to the current device, and a host thread that
 * drives several contexts has whichever device it used last. */
int enter(PtContext* ctx) { HIPCHK(hipSetDevice(ctx->device)); return 0; }

int check_scene(PtContext* ctx) { return ctx->T ? 0 : ctx->fail(-2, "no scene uploaded (call pt_scene_upload first)"); }
int check_frame(PtContext* ctx)
{
    if (int rc = check_scene(ctx)) return rc;
    if (!ctx->W) return ctx->fail(-2, "no frame resources (call pt_resize first)");
    if (!ctx->haveConsts) return ctx->fail(-2, "no constants (call pt_set_constants first)");
    return 0;
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
