/*
 * comm_impl.h - multi-GPU: RCCL communicator per context and the sum all-reduce of the accumulation buffer (no reference counterpart).
 * A part of ptcore.hip's one translation unit (included there, in order); not a header for anyone else.
 */
extern "C" {

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
