/*
 * ptcore.hip — the device context behind include/ptcore.h: owns every buffer the reference binds
 * to its descriptor set (reference PathTracer.cpp:490-502), launches the kernels of
 * trace_kernels.h / frame_kernels.h in order on one HIP stream, and reports failures as status
 * codes.  There is no CPU path: without a gfx950 device pt_ctx_create fails.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "frame_kernels.h"
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

struct PtContext {
    int device = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;      /* stream2: shadow rays of depth d beside the closest-hit pass of depth d+1 */
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evShade[2] = { nullptr, nullptr }, evShadow[2] = { nullptr, nullptr };
    bool shadowBusy[2] = { false, false };
    std::string err;
    int numCUs = 0;

    /* scene (bindings 1,2,3,4,7,8) */
    float4* dPositions = nullptr; PtVertexAux* dAux = nullptr; uint32_t* dIndices = nullptr; PtBVHNode* dNodes = nullptr;
    PtLight* dLights = nullptr; uint16_t* dBlueNoise = nullptr;
    uint4* dRecords = nullptr; uint2* dOverflow = nullptr; uint32_t leafBase = 0, recordBytes = 0;
    uint32_t V = 0, T = 0, N = 0, L = 0, maxStack = 0, overflowEntries = 0;
    uint64_t overflowLanes = 0;
    float rootBox[6] = {0}; float rootC[3] = {0}; uint16_t rootH[6] = {0};

    /* per-resolution (bindings 5,6,9,10) */
    uint32_t W = 0, H = 0, slots = 0;
    PtRayStackSlot* dStack = nullptr; uint32_t* dJobs[2] = { nullptr, nullptr }; int curJobs = 0;
    PtHitRec* dHitRecs = nullptr; PtShadowRay* dShadow[2] = { nullptr, nullptr }; float4* dAccum = nullptr; uint2* dCurrent = nullptr;
    uint32_t* dCtr = nullptr; unsigned long long* dStats = nullptr; unsigned long long* dRayCounts = nullptr;

    PtShaderConstants consts; bool haveConsts = false;
    int optTraversal = 2, optStale = 0, optReseed = 0, optBlocksPerCU = 5, optRefillMin = 36;
    float lastMs = 0.0f; bool evPending = false;
    void* comm = nullptr;

    int fail(int code, const std::string& m) { err = m; return code; }
};

#define HIPCHK(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return ctx->fail(-10, std::string(#call) + ": " + hipGetErrorString(e_));      \
    } while (0)

namespace {

template <class T> void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

void free_scene(PtContext* c)
{
    dfree(c->dPositions); dfree(c->dAux); dfree(c->dIndices); dfree(c->dNodes); dfree(c->dLights); dfree(c->dBlueNoise);
    dfree(c->dRecords); dfree(c->dOverflow);
    c->V = c->T = c->N = c->L = 0;
}
void free_frame(PtContext* c)
{
    dfree(c->dStack); dfree(c->dJobs[0]); dfree(c->dJobs[1]); dfree(c->dHitRecs); dfree(c->dShadow[0]); dfree(c->dShadow[1]); dfree(c->dAccum); dfree(c->dCurrent);
    c->W = c->H = c->slots = 0;
}

int grid_blocks(PtContext* c) { return c->numCUs * c->optBlocksPerCU; }

int ensure_overflow(PtContext* ctx)
{
    uint32_t need = ctx->maxStack > PT_STACK_LDS ? ctx->maxStack - PT_STACK_LDS : 0;
    uint64_t lanes = (uint64_t)grid_blocks(ctx) * PT_BLOCK;
    if (need <= ctx->overflowEntries && lanes <= ctx->overflowLanes && (ctx->dOverflow || need == 0)) return 0;
    dfree(ctx->dOverflow);
    ctx->overflowEntries = need; ctx->overflowLanes = lanes;
    /* two areas: kernels on the two streams of pt_render_sample run side by side and must not share spill slots */
    if (need) HIPCHK(hipMalloc((void**)&ctx->dOverflow, 2 * (size_t)need * lanes * sizeof(uint2)));
    return 0;
}

void fill_trace_args(PtContext* c, TraceArgs& a)
{
    memset(&a, 0, sizeof(a));
    a.nodes = c->dNodes; a.numNodes = c->N; a.indices = c->dIndices; a.positions = c->dPositions;
    a.records = c->dRecords; a.recordBytes = c->recordBytes; a.leafBase = c->leafBase; a.tris = c->dRecords + c->leafBase / 16;
    for (int k = 0; k < 3; k++) a.rootC[k] = c->rootC[k];
    for (int k = 0; k < 6; k++) a.rootH[k] = c->rootH[k];
    a.rootRef = (c->T == 1) ? PT_LEAF_BIT : 0u;
    a.overflow = c->dOverflow; a.overflowEntries = c->overflowEntries;
    a.refillMin = (uint32_t)c->optRefillMin;
    a.stack = c->dStack; a.hitRecs = c->dHitRecs; a.shadowRays = c->dShadow[0];
    a.stats = c->dStats;
}

template <int MODE, bool ANYHIT>
int launch_trace(PtContext* ctx, TraceArgs& a, bool count, hipStream_t stream = nullptr)
{
    if (!stream) stream = ctx->stream;
    if (int rc = ensure_overflow(ctx)) return rc;
    a.overflow = ctx->dOverflow; a.overflowEntries = ctx->overflowEntries;
    if (stream == ctx->stream2 && ctx->dOverflow) a.overflow = ctx->dOverflow + (size_t)ctx->overflowEntries * ctx->overflowLanes;
    dim3 grid(grid_blocks(ctx)), block(PT_BLOCK);
    const int trav = ctx->optTraversal;
    if (trav == 0) {
        if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_ref<MODE_RAYS, ANYHIT, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_ref<MODE, ANYHIT, false>), grid, block, 0, stream, a);
    } else if (trav == 1) {
        if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_flat<MODE_RAYS, ANYHIT, false, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_flat<MODE, ANYHIT, false, false>), grid, block, 0, stream, a);
    } else {
        if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_flat<MODE_RAYS, ANYHIT, true, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_flat<MODE, ANYHIT, true, false>), grid, block, 0, stream, a);
    }
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

void fill_frame_args(PtContext* c, FrameArgs& f)
{
    memset(&f, 0, sizeof(f));
    f.c = c->consts; f.W = c->W; f.H = c->H; f.slots = c->slots;
    f.zeroNextSlot = c->optStale ? 0u : 1u; f.reseed = c->optReseed ? 1u : 0u;
    f.stack = c->dStack; f.jobs = c->dJobs[c->curJobs]; f.jobsOut = c->dJobs[c->curJobs ^ 1]; f.ctr = c->dCtr;
    f.hitRecs = c->dHitRecs; f.shadowRays = c->dShadow[0]; f.shadowQueue = 0; f.fuseCompact = 0; f.aux = c->dAux; f.lights = c->dLights; f.blueNoise = c->dBlueNoise;
    f.accumulation = c->dAccum; f.current = c->dCurrent; f.rayCounts = c->dRayCounts;
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
        (e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->evShade[0], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->evShade[1], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->evShadow[0], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->evShadow[1], hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc((void**)&ctx->dCtr, CTR_WORDS * 4)) != hipSuccess || (e = hipMalloc((void**)&ctx->dStats, 4 * 8)) != hipSuccess) {
        g_createError = hipGetErrorString(e);
        delete ctx;
        return -10;
    }
    (void)hipMemset(ctx->dCtr, 0, CTR_WORDS * 4);
    memset(&ctx->consts, 0, sizeof(ctx->consts));
    *out = ctx;
    return 0;
}

PT_API void pt_ctx_destroy(PtContext* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->comm && g_nccl.CommDestroy) g_nccl.CommDestroy(ctx->comm);
    free_scene(ctx); free_frame(ctx);
    dfree(ctx->dCtr); dfree(ctx->dStats); dfree(ctx->dRayCounts);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (int k = 0; k < 2; k++) { if (ctx->evShade[k]) (void)hipEventDestroy(ctx->evShade[k]); if (ctx->evShadow[k]) (void)hipEventDestroy(ctx->evShadow[k]); }
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

PT_API const char* pt_last_error(PtContext* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

PT_API int pt_synchronize(PtContext* ctx)
{
    if (!ctx) return -1;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API void* pt_stream(PtContext* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

PT_API int pt_set_option(PtContext* ctx, int option, int value)
{
    if (!ctx) return -1;
    switch (option) {
        case PT_OPT_TRAVERSAL: if (value < 0 || value > 2) return ctx->fail(-1, "PT_OPT_TRAVERSAL must be 0, 1 or 2"); ctx->optTraversal = value; return 0;
        case PT_OPT_STALE_SLOTS: ctx->optStale = value != 0; return 0;
        case PT_OPT_RESEED: ctx->optReseed = value != 0; return 0;
        case PT_OPT_PERSIST_BLOCKS: if (value < 1 || value > 8) return ctx->fail(-1, "PT_OPT_PERSIST_BLOCKS must be 1..8"); ctx->optBlocksPerCU = value; return 0;
        case PT_OPT_REFILL_MIN: if (value < 1 || value > 64) return ctx->fail(-1, "PT_OPT_REFILL_MIN must be 1..64"); ctx->optRefillMin = value; return 0;
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
        default: return ctx->fail(-1, "unknown option");
    }
}

PT_API int pt_scene_upload(PtContext* ctx, const float* positions, uint32_t V, const PtVertexAux* aux, const uint32_t* indices, uint32_t T,
                           const PtBVHNode* nodes, uint32_t N, const PtLight* lights, uint32_t L, const uint16_t* blue_noise)
{
    if (!ctx) return -1;
    if (!positions || !indices || !nodes || V == 0 || T == 0) return ctx->fail(-1, "pt_scene_upload: null or empty geometry");
    if (N != 2 * T - 1) return ctx->fail(-1, "pt_scene_upload: node count must be 2*T-1");
    for (size_t i = 0; i < 3 * (size_t)T; i++) if (indices[i] >= V) return ctx->fail(-1, "pt_scene_upload: vertex index out of range");
    for (uint32_t i = 0; i < N; i++) {
        if (nodes[i].next < 0 || (uint32_t)nodes[i].next >= N || nodes[i].right >= (int32_t)N) return ctx->fail(-1, "pt_scene_upload: node link out of range");
        if (nodes[i].right <= 0 && ((uint32_t)(-nodes[i].right) % 3 != 0 || (uint32_t)(-nodes[i].right) / 3 >= T)) return ctx->fail(-1, "pt_scene_upload: leaf offset out of range");
    }
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_scene(ctx);

    std::vector<PtNode2> n2(T > 1 ? T - 1 : 1);
    std::vector<PtTri> tr(T);
    char ebuf[256] = {0};
    uint32_t maxStack = 0;
    if (pt_flatten_bvh(nodes, N, positions, indices, T, n2.data(), tr.data(), ctx->rootBox, &maxStack, ebuf, sizeof(ebuf)) != 0)
        return ctx->fail(-1, ebuf);
    ctx->maxStack = maxStack;
    for (int k = 0; k < 3; k++) {      /* root box digest, as for every other box (intersections.glsl:3-5) */
        ctx->rootC[k] = (ctx->rootBox[k] + ctx->rootBox[3 + k]) * 0.5f;
        _Float16 ha = (_Float16)(ctx->rootBox[k] - ctx->rootC[k]), hb = (_Float16)(ctx->rootBox[3 + k] - ctx->rootC[k]);
        memcpy(&ctx->rootH[k], &ha, 2); memcpy(&ctx->rootH[3 + k], &hb, 2);
    }

    HIPCHK(hipMalloc((void**)&ctx->dPositions, (size_t)V * 16));
    HIPCHK(hipMalloc((void**)&ctx->dIndices, (size_t)T * 12));
    HIPCHK(hipMalloc((void**)&ctx->dNodes, (size_t)N * sizeof(PtBVHNode)));
    /* interior records and leaf records share one buffer so that one buffer descriptor serves every fetch */
    const size_t interiorBytes = (size_t)(T - 1) * sizeof(PtNode2), recBytes = interiorBytes + (size_t)T * sizeof(PtTri);
    if (recBytes >= 0xFFFFFFFFull) return ctx->fail(-1, "pt_scene_upload: more than 2^32 bytes of traversal records (about 33 M triangles)");
    HIPCHK(hipMalloc((void**)&ctx->dRecords, recBytes));
    ctx->leafBase = (uint32_t)interiorBytes; ctx->recordBytes = (uint32_t)recBytes;
    HIPCHK(hipMemcpy(ctx->dPositions, positions, (size_t)V * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->dIndices, indices, (size_t)T * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->dNodes, nodes, (size_t)N * sizeof(PtBVHNode), hipMemcpyHostToDevice));
    if (T > 1) HIPCHK(hipMemcpy(ctx->dRecords, n2.data(), interiorBytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy((char*)ctx->dRecords + interiorBytes, tr.data(), (size_t)T * sizeof(PtTri), hipMemcpyHostToDevice));
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
}

PT_API int pt_resize(PtContext* ctx, uint32_t W, uint32_t H, uint32_t max_slots)
{
    if (!ctx) return -1;
    if (W == 0 || H == 0 || max_slots == 0) return ctx->fail(-1, "pt_resize: zero dimension");
    if ((uint64_t)W * H * max_slots >= 0xFFFFFFFFull) return ctx->fail(-1, "pt_resize: W*H*slots must fit a 32-bit slot index");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_frame(ctx);
    dfree(ctx->dRayCounts);
    size_t px = (size_t)W * H;
    HIPCHK(hipMalloc((void**)&ctx->dStack, px * max_slots * sizeof(PtRayStackSlot)));
    HIPCHK(hipMalloc((void**)&ctx->dJobs[0], px * 4));
    HIPCHK(hipMalloc((void**)&ctx->dJobs[1], px * 4));
    HIPCHK(hipMalloc((void**)&ctx->dHitRecs, px * sizeof(PtHitRec)));
    HIPCHK(hipMalloc((void**)&ctx->dShadow[0], px * sizeof(PtShadowRay)));
    HIPCHK(hipMalloc((void**)&ctx->dShadow[1], px * sizeof(PtShadowRay)));
    HIPCHK(hipMalloc((void**)&ctx->dAccum, px * 16));
    HIPCHK(hipMalloc((void**)&ctx->dCurrent, px * 8));
    HIPCHK(hipMalloc((void**)&ctx->dRayCounts, (size_t)2 * max_slots * 8));
    /* the reference leaves these undefined; zero makes the first frame deterministic */
    HIPCHK(hipMemset(ctx->dStack, 0, px * max_slots * sizeof(PtRayStackSlot)));
    HIPCHK(hipMemset(ctx->dJobs[0], 0xFF, px * 4));
    HIPCHK(hipMemset(ctx->dJobs[1], 0xFF, px * 4));
    HIPCHK(hipMemset(ctx->dAccum, 0, px * 16));
    HIPCHK(hipMemset(ctx->dCurrent, 0, px * 8));
    HIPCHK(hipMemset(ctx->dRayCounts, 0, (size_t)2 * max_slots * 8));
    HIPCHK(hipMemset(ctx->dCtr, 0, CTR_WORDS * 4));
    ctx->W = W; ctx->H = H; ctx->slots = max_slots; ctx->curJobs = 0;
    return 0;
}

PT_API int pt_set_constants(PtContext* ctx, const PtShaderConstants* c)
{
    if (!ctx || !c) return -1;
    if (ctx->W && ((uint32_t)c->viewportSize[0] != ctx->W || (uint32_t)c->viewportSize[1] != ctx->H)) return ctx->fail(-1, "pt_set_constants: viewportSize differs from pt_resize");
    if (ctx->slots && (c->numRays == 0 || c->numRays > ctx->slots)) return ctx->fail(-1, "pt_set_constants: numRays exceeds the slots allocated by pt_resize");
    if (ctx->N && c->numBVHNodes != ctx->N) return ctx->fail(-1, "pt_set_constants: numBVHNodes differs from the uploaded scene");
    if (c->viewportBase[0] != 0.0f || c->viewportBase[1] != 0.0f) return ctx->fail(-1, "pt_set_constants: viewportBase must be 0");
    ctx->consts = *c; ctx->haveConsts = true;
    return 0;
}

PT_API int pt_clear_accumulation(PtContext* ctx)
{
    if (!ctx) return -1;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    HIPCHK(hipMemsetAsync(ctx->dAccum, 0, (size_t)ctx->W * ctx->H * 16, ctx->stream));
    return 0;
}

PT_API int pt_launch(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = check_frame(ctx)) return rc;
    if (!ctx->dBlueNoise) return ctx->fail(-2, "pt_launch: scene was uploaded without a blue-noise table");
    ctx->curJobs = 0;
    FrameArgs f; fill_frame_args(ctx, f);
    dim3 grid((ctx->W + 31) / 32, (ctx->H + 7) / 8), block(256);
    hipLaunchKernelGGL(k_launch, grid, block, 0, ctx->stream, f);
    HIPCHK(hipGetLastError());
    return 0;
}

/* One depth of the frame.  overlapped = false: the three passes run in order on the context's stream and jobs are
 * rewritten in place (what pt_trace promises).  overlapped = true (pt_render_sample): the shade pass also compacts
 * the job list, and the shadow pass runs on the second stream beside the next depth's closest-hit pass; two shadow
 * queues alternate, and a queue is reused only after the shadow pass that read it has finished. */
static int trace_depth(PtContext* ctx, uint32_t depth, int speculative, bool overlapped)
{
    const int q = overlapped ? (int)(depth & 1u) : 0;
    if (ctx->shadowBusy[q]) {      /* queue q and its counters were last read by a shadow pass on stream2 */
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->evShadow[q], 0));
        ctx->shadowBusy[q] = false;
    }
    HIPCHK(hipMemsetAsync(ctx->dCtr + CTR_FETCH_CLOSEST, 0, 3 * 4, ctx->stream));              /* fetch, compact-out, live */
    HIPCHK(hipMemsetAsync(ctx->dCtr + CTR_NSHADOW + q, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->dCtr + CTR_FETCH_SHADOW + q, 0, 4, ctx->stream));
    FrameArgs f; fill_frame_args(ctx, f);
    f.shadowRays = ctx->dShadow[q]; f.shadowQueue = (uint32_t)q; f.fuseCompact = overlapped ? 1u : 0u;
    TraceArgs a; fill_trace_args(ctx, a);
    a.jobs = f.jobs; a.countPtr = ctx->dCtr + CTR_NJOBS; a.fetchCounter = ctx->dCtr + CTR_FETCH_CLOSEST;
    if (speculative) {
        /* needs the job list in pixel order, which holds right after pt_launch */
        dim3 grid(ctx->W / 8, ctx->H / 8), block(64);
        hipLaunchKernelGGL(k_trace_ref_speculative, grid, block, 0, ctx->stream, a, ctx->W, ctx->H);
        HIPCHK(hipGetLastError());
    } else {
        if (int rc = launch_trace<MODE_JOBS, false>(ctx, a, false)) return rc;
    }
    dim3 sgrid(((size_t)ctx->W * ctx->H + 255) / 256), sblock(256);
    hipLaunchKernelGGL(k_shade, sgrid, sblock, 0, ctx->stream, f, depth);
    hipLaunchKernelGGL(k_record_counts, dim3(1), dim3(1), 0, ctx->stream, f, depth);
    HIPCHK(hipGetLastError());
    if (overlapped) ctx->curJobs ^= 1;                                                      /* the shade pass wrote the next list */

    TraceArgs s; fill_trace_args(ctx, s);
    s.shadowRays = ctx->dShadow[q];
    s.countPtr = ctx->dCtr + CTR_NSHADOW + q; s.fetchCounter = ctx->dCtr + CTR_FETCH_SHADOW + q;
    if (overlapped) {
        HIPCHK(hipEventRecord(ctx->evShade[q], ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->evShade[q], 0));
        if (int rc = launch_trace<MODE_SHADOW, true>(ctx, s, false, ctx->stream2)) return rc;
        HIPCHK(hipEventRecord(ctx->evShadow[q], ctx->stream2));
        ctx->shadowBusy[q] = true;
    } else {
        if (int rc = launch_trace<MODE_SHADOW, true>(ctx, s, false)) return rc;
    }
    return 0;
}

/* Makes the context's stream wait for shadow passes still running on the second stream. */
static int join_shadow(PtContext* ctx)
{
    for (int q = 0; q < 2; q++)
        if (ctx->shadowBusy[q]) { HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->evShadow[q], 0)); ctx->shadowBusy[q] = false; }
    return 0;
}

PT_API int pt_trace(PtContext* ctx, uint32_t depth, int speculative)
{
    if (!ctx) return -1;
    if (int rc = check_frame(ctx)) return rc;
    if (!ctx->dAux || !ctx->dLights) return ctx->fail(-2, "pt_trace: scene was uploaded without vertex attributes or lights");
    if (depth >= ctx->slots) return ctx->fail(-1, "pt_trace: depth exceeds the allocated slots");
    if (speculative && (ctx->W % 8 || ctx->H % 8)) return ctx->fail(-1, "pt_trace: the speculative build needs W and H to be multiples of 8");
    return trace_depth(ctx, depth, speculative, false);
}

PT_API int pt_compact(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = check_frame(ctx)) return rc;
    HIPCHK(hipMemsetAsync(ctx->dCtr + CTR_COMPACT_OUT, 0, 4, ctx->stream));
    FrameArgs f; fill_frame_args(ctx, f);
    dim3 grid(((size_t)ctx->W * ctx->H + 255) / 256), block(256);
    hipLaunchKernelGGL(k_compact, grid, block, 0, ctx->stream, f);
    hipLaunchKernelGGL(k_compact_finish, dim3(1), dim3(1), 0, ctx->stream, f);
    HIPCHK(hipGetLastError());
    ctx->curJobs ^= 1;
    return 0;
}

PT_API int pt_resolve_accumulate(PtContext* ctx)
{
    if (!ctx) return -1;
    if (int rc = check_frame(ctx)) return rc;
    FrameArgs f; fill_frame_args(ctx, f);
    dim3 grid(((size_t)ctx->W * ctx->H + 255) / 256), block(256);
    hipLaunchKernelGGL(k_resolve, grid, block, 0, ctx->stream, f);
    HIPCHK(hipGetLastError());
    return 0;
}

PT_API int pt_render_sample(PtContext* ctx, uint32_t frame_index, int speculative_depth0, uint64_t* ray_counts)
{
    if (!ctx) return -1;
    if (int rc = check_frame(ctx)) return rc;
    ctx->consts.frameIndex = frame_index & 0xFFFF;                                          /* PathTracer.cpp:729 */
    if (ray_counts) HIPCHK(hipMemsetAsync(ctx->dRayCounts, 0, (size_t)2 * ctx->slots * 8, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    if (int rc = pt_launch(ctx)) return rc;
    if (!ctx->dAux || !ctx->dLights) return ctx->fail(-2, "pt_render_sample: scene was uploaded without vertex attributes or lights");
    if (speculative_depth0 && (ctx->W % 8 || ctx->H % 8)) return ctx->fail(-1, "pt_render_sample: the speculative build needs W and H to be multiples of 8");
    for (uint32_t d = 0; d < ctx->consts.numRays; d++)                                      /* PathTracer.cpp:780-812 */
        if (int rc = trace_depth(ctx, d, d == 0 && speculative_depth0, true)) return rc;
    if (int rc = join_shadow(ctx)) return rc;
    if (int rc = pt_resolve_accumulate(ctx)) return rc;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->evPending = true;
    if (ray_counts) {
        HIPCHK(hipMemcpyAsync(ray_counts, ctx->dRayCounts, (size_t)2 * ctx->consts.numRays * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

PT_API int pt_readback_current(PtContext* ctx, uint16_t* rgba16f)
{
    if (!ctx || !rgba16f) return -1;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    HIPCHK(hipMemcpyAsync(rgba16f, ctx->dCurrent, (size_t)ctx->W * ctx->H * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API int pt_readback_accumulation(PtContext* ctx, float* rgba32f)
{
    if (!ctx || !rgba32f) return -1;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    HIPCHK(hipMemcpyAsync(rgba32f, ctx->dAccum, (size_t)ctx->W * ctx->H * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API int pt_readback_raystack(PtContext* ctx, PtRayStackSlot* slots, PtRayJob* jobs)
{
    if (!ctx) return -1;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    size_t px = (size_t)ctx->W * ctx->H;
    if (slots) HIPCHK(hipMemcpyAsync(slots, ctx->dStack, px * ctx->slots * sizeof(PtRayStackSlot), hipMemcpyDeviceToHost, ctx->stream));
    if (jobs) HIPCHK(hipMemcpyAsync(jobs, ctx->dJobs[ctx->curJobs], px * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API int pt_upload_raystack(PtContext* ctx, const PtRayStackSlot* slots, const PtRayJob* jobs)
{
    if (!ctx) return -1;
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    size_t px = (size_t)ctx->W * ctx->H;
    if (slots) HIPCHK(hipMemcpyAsync(ctx->dStack, slots, px * ctx->slots * sizeof(PtRayStackSlot), hipMemcpyHostToDevice, ctx->stream));
    if (jobs) {
        uint32_t n = (uint32_t)px;
        HIPCHK(hipMemcpyAsync(ctx->dJobs[ctx->curJobs], jobs, px * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->dCtr + CTR_NJOBS, &n, 4, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API void* pt_accumulation_device_ptr(PtContext* ctx) { return ctx ? (void*)ctx->dAccum : nullptr; }

PT_API int pt_trace_rays(PtContext* ctx, const PtRay* d_rays, uint64_t num_rays, int any_hit, PtHit* d_hits, uint64_t* stats)
{
    if (!ctx) return -1;
    if (int rc = check_scene(ctx)) return rc;
    if (!d_rays || !d_hits) return ctx->fail(-1, "pt_trace_rays: null buffer");
    if (num_rays >= 0xFFFFFF00ull) return ctx->fail(-1, "pt_trace_rays: at most 2^32-256 rays per call");
    if (num_rays == 0) return 0;
    HIPCHK(hipMemsetAsync(ctx->dCtr + CTR_FETCH_CLOSEST, 0, 4, ctx->stream));
    if (stats) HIPCHK(hipMemsetAsync(ctx->dStats, 0, 32, ctx->stream));
    TraceArgs a; fill_trace_args(ctx, a);
    a.rays = d_rays; a.hits = d_hits; a.count = num_rays; a.countPtr = nullptr; a.fetchCounter = ctx->dCtr + CTR_FETCH_CLOSEST;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    int rc = any_hit ? launch_trace<MODE_RAYS, true>(ctx, a, stats != nullptr) : launch_trace<MODE_RAYS, false>(ctx, a, stats != nullptr);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->evPending = true;
    if (stats) {
        HIPCHK(hipMemcpyAsync(stats, ctx->dStats, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

PT_API int pt_trace_rays_host(PtContext* ctx, const PtRay* rays, uint64_t n, int any_hit, PtHit* hits, uint64_t* stats)
{
    if (!ctx) return -1;
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
    if (!ctx) return nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { ctx->fail(-10, std::string("hipMalloc: ") + hipGetErrorString(e)); return nullptr; }
    return p;
}
PT_API int pt_device_free(PtContext* ctx, void* p)
{
    if (!ctx) return -1;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(p));
    return 0;
}
PT_API int pt_copy_to_host(PtContext* ctx, void* dst, const void* d_src, size_t bytes)
{
    if (!ctx) return -1;
    HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
PT_API int pt_copy_to_device(PtContext* ctx, void* d_dst, const void* src, size_t bytes)
{
    if (!ctx) return -1;
    HIPCHK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
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
    if (!ctx->comm) return ctx->fail(-2, "pt_allreduce_accumulation: call pt_comm_init first");
    if (!ctx->W) return ctx->fail(-2, "no frame resources");
    /* one sum over RGBA32F: .w carries the sample count and sums with the radiance */
    int rc = g_nccl.AllReduce(ctx->dAccum, ctx->dAccum, (size_t)ctx->W * ctx->H * 4, /*ncclFloat32*/ 7, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    if (rc != 0) return ctx->fail(-21, std::string("ncclAllReduce: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "error"));
    return 0;
}

}  // extern "C"
