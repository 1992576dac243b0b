/*
 * launch_impl.h - the launches of the traversal kernels: which kernel, how wide, which spill area; the certified walk's
 * two passes; the merged closest-hit + shadow launch.
 * A part of ptcore.hip's one translation unit (included there, in order); not a header for anyone else.
 */
namespace {

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
    /* the batch kernel takes 72 VGPRs and 22.5 KB of LDS per workgroup: seven workgroups per CU, which an ordered batch uses
       (workgroups x idle lanes at refill -> ms per 2^26 rays: 6 x 8 / 10 / 14 / 20: 27.91 / 27.59 / 27.40 / 27.52;
       7 x 10 / 14 / 20 / 24 / 28: 27.55 / 27.00 / 26.67 / 26.83 / 27.26; 8 x 20: 26.65) */
    const bool wide = MODE == MODE_RAYS && shape == SHAPE_ORDERED_BATCH;
    if (wide && !ctx->userBlocks) blocks = ctx->numCUs * 7;
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
        /* batches: the lean step (k_trace_tree); the frame passes: six boxes per step (k_trace_tree6) - each where it measures best */
        if (trav == 1) {
            if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_tree<MODE_RAYS, ANYHIT, false, true>), grid, block, 0, stream, a);
            else if (MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_tree<MODE_RAYS, ANYHIT, false, false>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k_trace_tree6<MODE, ANYHIT, false, false>), grid, block, 0, stream, a);
        } else {
            if (count && MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_tree<MODE_RAYS, ANYHIT, true, true>), grid, block, 0, stream, a);
            else if (MODE == MODE_RAYS) hipLaunchKernelGGL((k_trace_tree<MODE_RAYS, ANYHIT, true, false>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k_trace_tree6<MODE, ANYHIT, true, false>), grid, block, 0, stream, a);
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
        if (ordered) hipLaunchKernelGGL((k_trace_tree6<MODE_MIXED, false, true, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_trace_tree6<MODE_MIXED, false, false, false>), grid, block, 0, stream, a);
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

/* The certified walk over the batch or job list `a` describes, then the reference-order walk over the rays it listed.
 * flag: room for one index per ray; ctr: two zeroed words (count of listed rays, the second walk's fetch counter). */
template <int MODE>
int launch_cert(PtContext* ctx, TraceArgs& a, bool count, hipStream_t stream, int area, uint32_t* flag, uint32_t* ctr, const BatchScratch* fixed = nullptr)
{
    const int shape = MODE != MODE_RAYS ? SHAPE_FRAME : (a.order ? SHAPE_ORDERED_BATCH : SHAPE_BATCH);
    int blocks = ctx->numCUs * blocks_per_cu(ctx, shape);
    /* the batch kernel takes 72 VGPRs and 22.5 KB of LDS per workgroup: seven workgroups per CU, which an ordered batch uses
       (workgroups x idle lanes at refill -> ms per 2^26 rays: 6 x 8 / 10 / 14 / 20: 27.91 / 27.59 / 27.40 / 27.52;
       7 x 10 / 14 / 20 / 24 / 28: 27.55 / 27.00 / 26.67 / 26.83 / 27.26; 8 x 20: 26.65) */
    const bool wide = MODE == MODE_RAYS && shape == SHAPE_ORDERED_BATCH;
    if (wide && !ctx->userBlocks) blocks = ctx->numCUs * 7;
    if (fixed && fixed->maxBlocks) {
        a.overflow = fixed->spill; a.overflowEntries = fixed->spillEntries;
        if (blocks > fixed->maxBlocks) blocks = fixed->maxBlocks;
    } else {
        if (int rc = ensure_overflow(ctx)) return rc;
        a.overflow = ctx->dOverflow ? ctx->dOverflow + (size_t)area * ctx->overflowEntries * ctx->overflowLanes : nullptr;
        a.overflowEntries = ctx->overflowEntries;
    }
    a.flagList = flag; a.flagCount = ctr;
    /* a step of this walk is longer than the near-first walk's, so idle lanes cost more and are refilled sooner (six
       workgroups per CU wide, as an unordered batch is: 4 / 8 / 12 / 16 / 24 idle lanes -> 2 094 / 2 170 / 2 165 / 2 138 /
       2 067 Mrays/s); seven wide, later again (above) */
    a.refillMin = (uint32_t)(MODE == MODE_RAYS && !ctx->userRefill ? (wide ? 20 : 10) : refill_min(ctx, shape));
    dim3 grid(blocks), block(PT_BLOCK);
    if (count) hipLaunchKernelGGL((k_trace_cert<MODE, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_trace_cert<MODE, false>), grid, block, 0, stream, a);
    HIPCHK(hipGetLastError());
    TraceArgs b = a;
    b.order = flag; b.countPtr = ctr; b.fetchCounter = ctr + 1; b.flagList = nullptr; b.flagCount = nullptr;
    /* Few listed rays are all latency - the walk of the longest of them - and go one to a wavefront; many are throughput
       and go one to a lane (the walk of traversal 1).  How many there are only the device knows, so by default both
       launches are queued and each looks at the count: the one whose case it is not leaves at once. */
#ifndef PT_SECOND_WALK_SPLIT
#define PT_SECOND_WALK_SPLIT 16384
#endif
    if (ctx->optSecondWalk == 2) { b.onlyFrom = PT_SECOND_WALK_SPLIT; b.onlyBelow = PT_SECOND_WALK_SPLIT; }
    if (ctx->optSecondWalk != 1)
        if (int rc = launch_trace<MODE, false>(ctx, b, count, stream, area, fixed)) return rc;    /* traversal 3 maps to the reference-order walk there */
    if (ctx->optSecondWalk == 0) return 0;
    /* the listed rays are few and their walks long: one wavefront per ray (k_trace_ref_wave); no stack, no spill area */
#ifndef PT_WAVE_WALK_BLOCKS
#define PT_WAVE_WALK_BLOCKS 32        /* a batch lists some ten thousand rays: one per wavefront where the chip holds that many (8 / 16 / 32 per CU: 29.31 / 29.09 / 29.00 ms per 2^26-ray call) */
#endif
    dim3 grid2(ctx->numCUs * (MODE == MODE_RAYS ? PT_WAVE_WALK_BLOCKS : 8)), block2(64);        /* one wavefront per workgroup */
    if (count) hipLaunchKernelGGL((k_trace_ref_wave<MODE, true>), grid2, block2, 0, stream, b);
    else hipLaunchKernelGGL((k_trace_ref_wave<MODE, false>), grid2, block2, 0, stream, b);
    HIPCHK(hipGetLastError());
    return 0;
}

/* The frame train's launch under traversal 3: the certified walk over the step's jobs, with the shadow rays of the step before
 * riding along, then the reference-order walk over the jobs it listed. */
/* walkInShade: the caller's shade launch walks a short list itself (k_shade_train), so only the launch for a long list is
 * queued here; *second then holds what that shade launch needs to know of the list. */
int launch_cert_mixed(PtContext* ctx, TraceArgs& a, hipStream_t stream, int area, uint32_t* flag, uint32_t* ctr, bool walkInShade = false, TraceArgs* second = nullptr)
{
    if (int rc = ensure_overflow(ctx)) return rc;
    a.overflow = ctx->dOverflow ? ctx->dOverflow + (size_t)area * ctx->overflowEntries * ctx->overflowLanes : nullptr;
    a.overflowEntries = ctx->overflowEntries;
    a.flagList = flag; a.flagCount = ctr;
    a.refillMin = (uint32_t)refill_min(ctx, SHAPE_FRAME);
    /* workgroups per CU, small scene: 2 ... 6 -> 402, 415, 402, 383, 382 spp/s; large: 4, 5, 6 -> 159, 169.5, 151.5 (at 80 VGPRs a
       sixth fits, and leaves the other band's passes no room) */
    dim3 grid(ctx->numCUs * (ctx->userBlocks ? ctx->optBlocksPerCU : small_scene(ctx) ? 3 : 5)), block(PT_BLOCK);
    hipLaunchKernelGGL((k_trace_cert<MODE_MIXED, false>), grid, block, 0, stream, a);
    HIPCHK(hipGetLastError());
    TraceArgs b = a;
    b.order = flag; b.countPtr = ctr; b.countPtr2 = nullptr; b.fetchCounter = ctr + 1; b.flagList = nullptr; b.flagCount = nullptr;
    if (ctx->optSecondWalk == 2) { b.onlyFrom = PT_SECOND_WALK_SPLIT; b.onlyBelow = PT_SECOND_WALK_SPLIT; }      /* see launch_cert */
    if (second) *second = b;
    if (ctx->optSecondWalk != 1)
        if (int rc = launch_trace<MODE_JOBS, false>(ctx, b, false, stream, area)) return rc;
    if (ctx->optSecondWalk == 0 || walkInShade) return 0;
    /* one wavefront per workgroup: these find room beside a launch that fills the CUs more easily than groups of four */
    hipLaunchKernelGGL((k_trace_ref_wave<MODE_JOBS, false>), dim3(ctx->numCUs * 8), dim3(64), 0, stream, b);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace
