/*
 * batch_impl.h - ray batches: pt_trace_rays (ordering pass + traversal), pt_trace_rays_noalloc and its static scratch array,
 * pt_trace_rays_host, pt_generate_rays, the entry-bound self-test, kernel timing.
 * A part of ptcore.hip's one translation unit (included there, in order); not a header for anyone else.
 */
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

extern "C" {

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

PT_API int pt_traversal_ms(PtContext* ctx, float* per_launch_ms, uint32_t capacity, uint32_t* launches)
{
    if (!ctx || !launches || (capacity && !per_launch_ms)) return -1;
    if (int rc = enter(ctx)) return rc;
    if (int rc = settle(ctx)) return rc;
    if (int rc = sync_all(ctx)) return rc;
    for (size_t i = 0; i < ctx->travUsed && i < capacity; i++)
        HIPCHK(hipEventElapsedTime(&per_launch_ms[i], ctx->travEvents[i].first, ctx->travEvents[i].second));
    *launches = (uint32_t)ctx->travUsed;
    ctx->travUsed = 0;
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

}  // extern "C"
