/*
 * pipeline_impl.h - the per-frame pipeline: frame pipes (one frame in flight each), the dispatch sites pt_launch, pt_trace,
 * pt_compact, pt_resolve_accumulate (reference PathTracer.cpp:776-838) and one whole frame on a pipe.
 * A part of ptcore.hip's one translation unit (included there, in order); not a header for anyone else.
 */
namespace {

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

}  // namespace

extern "C" {

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
        if (p.deferredShadow >= 0) {
            /* the shadow rays the last bounce's shade pass queued ride along (they cull with the reference's own test) */
            const int d0 = p.deferredShadow;
            a.shadowRays = p.dShadow[d0 & 1]; a.countPtr2 = p.dCtr + CTR_SHADOW + 2 * d0;
            if (int rc = launch_cert_mixed(ctx, a, p.stream, 2 * p.index, p.dFlag, flagCtr)) return rc;
        } else if (int rc = launch_cert<MODE_JOBS>(ctx, a, false, p.stream, 2 * p.index, p.dFlag, flagCtr)) return rc;
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
       launch is one of the kinds that cannot take them (the reference layout; the certified walk where it falls back to the
       reference-order walk) */
    const bool mergeable = ctx->optMergePasses && (ctx->optTraversal == 1 || ctx->optTraversal == 2 ||
                                                   (ctx->optTraversal == 3 && !ctx->compact && ctx->boxesNest));
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
}  // extern "C"
