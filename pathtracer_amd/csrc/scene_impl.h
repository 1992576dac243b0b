/*
 * scene_impl.h - context life cycle, options, scene upload (flattening passes included), per-resolution resources, constants,
 * read-backs, frame dump, display colour, device memory helpers (reference PathTracer.cpp:272-321, 378-459, 731-752, 848-854, 947-950).
 * A part of ptcore.hip's one translation unit (included there, in order); not a header for anyone else.
 */
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
    for (auto& e : ctx->travEvents) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
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
        case PT_OPT_TIME_TRAVERSAL: ctx->optTimeTraversal = value != 0; ctx->travUsed = 0; return 0;
        case PT_OPT_TRAIN_MAX_MB: if (value < 0) return ctx->fail(-1, "PT_OPT_TRAIN_MAX_MB must not be negative"); ctx->optTrainMaxMB = value; return 0;
        case PT_OPT_TRAIN_PARTS: if (value < 1 || value > PT_MAX_TRAIN_PARTS) return ctx->fail(-1, "PT_OPT_TRAIN_PARTS must be 1..4"); ctx->optTrainParts = value; return 0;
        case PT_OPT_SECOND_WALK: if (value < 0 || value > 2) return ctx->fail(-1, "PT_OPT_SECOND_WALK must be 0, 1 or 2"); ctx->optSecondWalk = value; return 0;
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
        case PT_OPT_TIME_TRAVERSAL: *value = ctx->optTimeTraversal; return 0;
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
    {
        /* the nodes from the root's right child down to the last node (each the right child of the one before), stored
           deepest first; the certified walk redoes their box tests for the few rays that need them */
        std::vector<PtSpineBox> spine;
        for (uint32_t i = 0; nodes[i].right > 0 && spine.size() <= PT_SPINE_MAX;) {
            i = (uint32_t)nodes[i].right;
            PtSpineBox sb; memset(&sb, 0, sizeof(sb));
            sb.index = i;
            for (int k = 0; k < 3; k++) {
                sb.c[k] = (nodes[i].a[k] + nodes[i].b[k]) * 0.5f;
                _Float16 ha = (_Float16)(nodes[i].a[k] - sb.c[k]), hb = (_Float16)(nodes[i].b[k] - sb.c[k]);
                memcpy(&sb.h[k], &ha, 2); memcpy(&sb.h[3 + k], &hb, 2);
            }
            spine.insert(spine.begin(), sb);
        }
        ctx->spineCount = 0;
        if (!spine.empty() && spine.size() <= PT_SPINE_MAX && spine.front().index == N - 1) {
            HIPCHK(hipMalloc((void**)&ctx->dSpine, spine.size() * sizeof(PtSpineBox)));
            HIPCHK(hipMemcpy(ctx->dSpine, spine.data(), spine.size() * sizeof(PtSpineBox), hipMemcpyHostToDevice));
            ctx->spineCount = (uint32_t)spine.size();
        }
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

}  // extern "C"
