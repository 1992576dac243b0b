/*
 * train_impl.h - the frame train (all frames of a call stepped together, one bounce per step, in bands of rows) and the
 * render entry points pt_render_samples, pt_render_sample, pt_render_flush.
 * A part of ptcore.hip's one translation unit (included there, in order); not a header for anyone else.
 */
extern "C" {

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
        /* certified walk: a short list of jobs for the second walk is walked by the shade launch itself (k_shade_train), which has a
           workgroup per four jobs of the longest list the device sends that way; where the host has fixed the kind of second walk
           (PT_OPT_SECOND_WALK 0 or 1) a list of any length is walked by a launch of its own */
        const bool walkInShade = certified && ctx->optSecondWalk == 2 && shaded;
        TraceArgs second; memset(&second, 0, sizeof(second));
        {   /* every frame in flight one bounce further, and the shadow rays of the bounce before */
            TraceArgs a; fill_trace_args(ctx, a);
            a.stack = t.dRing; a.hitRecs = p.dHitRecs;
            a.jobs = p.dJobs[cur]; a.countPtr = p.dCtr + TC_NJOBS; a.fetchCounter = p.dCtr + TC_FETCH;
            a.shadowRays = p.dShadow[q]; a.countPtr2 = p.dCtr + TC_NSHADOW + q;
            /* measurement only (PT_OPT_TIME_TRAVERSAL): an event pair around the launch, at most 4096 per reading */
            std::pair<hipEvent_t, hipEvent_t>* timed = nullptr;
            if (ctx->optTimeTraversal && ctx->travUsed < 4096) {
                if (ctx->travUsed == ctx->travEvents.size()) {
                    std::pair<hipEvent_t, hipEvent_t> e{ nullptr, nullptr };
                    HIPCHK(hipEventCreate(&e.first));
                    if (hipEventCreate(&e.second) != hipSuccess) { (void)hipEventDestroy(e.first); return ctx->fail(-10, "hipEventCreate (traversal timing)"); }
                    ctx->travEvents.push_back(e);
                }
                timed = &ctx->travEvents[ctx->travUsed++];
                HIPCHK(hipEventRecord(timed->first, p.stream));
            }
            if (certified) {
                if (int rc = launch_cert_mixed(ctx, a, p.stream, k, p.dFlag, p.dCtr + TC_FLAG, walkInShade, &second)) return rc;
            } else if (int rc = launch_mixed(ctx, a, p.stream, k, true)) return rc;
            if (timed) HIPCHK(hipEventRecord(timed->second, p.stream));
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
            /* a workgroup for every entry the list can have; those beyond its end (known on the device only) leave at once */
            const uint64_t most = (uint64_t)shaded * ppx;
            const uint64_t blocks = (most + PT_SHADE_BLOCK - 1) / PT_SHADE_BLOCK;
            /* in front of them, under the certified walk, the workgroups that walk the listed jobs: one per four jobs of the
               longest list that is walked this way (those beyond the list's end leave at once) */
            const uint32_t walkBlocks = walkInShade ? (uint32_t)(PT_SECOND_WALK_SPLIT / (PT_SHADE_BLOCK / 64)) : 0u;
            if (walkBlocks) hipLaunchKernelGGL(k_shade_train<true>, dim3((uint32_t)blocks + walkBlocks), dim3(PT_SHADE_BLOCK), 0, p.stream, f, second, walkBlocks);
            else hipLaunchKernelGGL(k_shade_train<false>, dim3((uint32_t)blocks), dim3(PT_SHADE_BLOCK), 0, p.stream, f, second, 0u);
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

}  // extern "C"
