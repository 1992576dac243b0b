/*
 * pt_host.cpp - a C++ host above the C ABI of include/ptcore.h, in the shape of the reference's own host:
 * what PathTracer.cpp does between loading a model and presenting a frame, with every Vulkan object and dispatch
 * replaced by the library call that stands in for it (table below).  No Python, no torch: the program links
 * libptcore.so and nothing else.
 *
 *   reference (PathTracer.cpp)                                      here
 *   :246-262  HimaliaPlyModel::LoadFile / BuildVertices / Indices    pt_ply_load
 *   :264      BuildBVH                                               pt_build_bvh
 *   :272-321  CreateBuffer + UploadToBufferEx x 6                    pt_scene_upload
 *   :378-459  per-resolution images and the ray stack                pt_resize
 *   :729-750  frameIndex, camera matrices, constants                 pt_fill_constants + pt_set_constants
 *   :776-778  launch dispatch                                        pt_launch
 *   :780-812  per bounce: trace dispatch, raysort dispatch           pt_trace, pt_compact
 *   :835-836  composite pass (resolve + accumulate)                  pt_resolve_accumulate
 *   :840-861  RayStackBuffer_<n>.bin.z dump                          pt_dump_current
 *   (swapchain image)                                                pt_tonemap + pt_write_ppm
 *
 * usage: pt_host <model.ply> <blue_noise_u16le.bin> <out dir> [width height bounces frames mode]
 *   mode "sites"   the reference's own loop: one frame at a time through the dispatch sites (default)
 *   mode "batch"   the same frames in one pt_render_samples call
 *   mode "kept"    one pt_render_sample call per frame, as f_onRender asks for them, with the frame train kept between
 *                  the calls (PT_OPT_TRAIN_KEEP): every call is one step of the train, the picture read in between lags
 *                  numRays frames behind, pt_render_flush finishes what is in flight
 * All three leave the same accumulation; the program prints its FNV-1a hash so that two runs can be compared.
 */
#include "../include/ptcore.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int fail(PtContext* ctx, const char* what)
{
    fprintf(stderr, "pt_host: %s: %s\n", what, pt_last_error(ctx));
    if (ctx) pt_ctx_destroy(ctx);
    return 1;
}

static uint64_t fnv1a(const void* p, size_t n)
{
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <model.ply> <blue_noise_u16le.bin> <out dir> [width height bounces frames sites|batch|kept]\n", argv[0]);
        return 2;
    }
    const char* modelPath = argv[1];
    const char* noisePath = argv[2];
    const std::string outDir = argv[3];
    const uint32_t W = argc > 4 ? (uint32_t)atoi(argv[4]) : 1920, H = argc > 5 ? (uint32_t)atoi(argv[5]) : 1080;
    const uint32_t numRays = argc > 6 ? (uint32_t)atoi(argv[6]) : 5;            /* PathTracer.cpp:96 */
    const uint32_t frames = argc > 7 ? (uint32_t)atoi(argv[7]) : 16;
    const bool batch = argc > 8 && strcmp(argv[8], "batch") == 0, kept = argc > 8 && strcmp(argv[8], "kept") == 0;
    char err[256] = "";

    /* the model: PathTracer.cpp:246-264 */
    uint32_t V = 0, T = 0;
    if (pt_ply_load(modelPath, &V, &T, nullptr, nullptr, nullptr, err, sizeof err)) { fprintf(stderr, "pt_host: %s\n", err); return 1; }
    std::vector<float> positions(4 * (size_t)V);
    std::vector<PtVertexAux> aux(V);
    std::vector<uint32_t> indices(3 * (size_t)T);
    if (pt_ply_load(modelPath, &V, &T, positions.data(), aux.data(), indices.data(), err, sizeof err)) { fprintf(stderr, "pt_host: %s\n", err); return 1; }
    std::vector<PtBVHNode> nodes(2 * (size_t)T - 1);
    uint32_t depth = 0;
    auto t0 = std::chrono::steady_clock::now();
    if (pt_build_bvh(positions.data(), V, indices.data(), T, nodes.data(), &depth, 0, err, sizeof err)) { fprintf(stderr, "pt_host: %s\n", err); return 1; }
    const double buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("model: %u vertices, %u triangles, BVH %zu nodes, depth %u, built in %.1f ms\n", V, T, nodes.size(), depth, buildMs);

    /* blueNoise.h is compiled into the reference; here the same table comes from a file */
    std::vector<uint16_t> noise(65536);
    FILE* nf = fopen(noisePath, "rb");
    if (!nf || fread(noise.data(), 2, noise.size(), nf) != noise.size()) { fprintf(stderr, "pt_host: cannot read %s\n", noisePath); return 1; }
    fclose(nf);
    PtLight light;                                                               /* PathTracer.cpp:41-43 */
    memset(&light, 0, sizeof light);
    light.pos[1] = 1.4f;
    light.radiance[0] = light.radiance[1] = light.radiance[2] = 1.0f;

    PtContext* ctx = nullptr;
    if (pt_ctx_create(0, &ctx)) return fail(nullptr, "pt_ctx_create");
    printf("%s\n", pt_version());
    if (pt_scene_upload(ctx, positions.data(), V, aux.data(), indices.data(), T, nodes.data(), (uint32_t)nodes.size(), &light, 1, noise.data()))
        return fail(ctx, "pt_scene_upload");
    if (pt_resize(ctx, W, H, numRays)) return fail(ctx, "pt_resize");

    /* the orbit camera of PathTracer.cpp:109-112 looking at the Cornell-box models */
    const float focus[3] = { 0.0f, 0.75f, 0.0f }, ambient[3] = { 0.4f, 0.5f, 0.7f };
    PtShaderConstants c;
    t0 = std::chrono::steady_clock::now();
    if (batch) {
        pt_fill_constants(&c, W, H, focus, 0.75f, 3.0f, -1.5707964f, 1, T, 1, numRays, (uint32_t)nodes.size(), ambient);
        if (pt_set_constants(ctx, &c)) return fail(ctx, "pt_set_constants");
        if (pt_render_samples(ctx, 1, frames, 0, nullptr)) return fail(ctx, "pt_render_samples");
    } else if (kept) {
        if (pt_set_option(ctx, PT_OPT_FRAME_TRAIN, 1) || pt_set_option(ctx, PT_OPT_TRAIN_KEEP, 1)) return fail(ctx, "pt_set_option");
        for (uint32_t frame = 1; frame <= frames; frame++) {                     /* f_onRender: one frame asked for, whatever is finished shown */
            pt_fill_constants(&c, W, H, focus, 0.75f, 3.0f, -1.5707964f, 1, T, frame, numRays, (uint32_t)nodes.size(), ambient);
            if (pt_set_constants(ctx, &c)) return fail(ctx, "pt_set_constants");
            if (pt_render_sample(ctx, frame, 0, nullptr)) return fail(ctx, "pt_render_sample");
        }
        if (pt_render_flush(ctx)) return fail(ctx, "pt_render_flush");
    } else {
        for (uint32_t frame = 1; frame <= frames; frame++) {                     /* f_onRender, PathTracer.cpp:729-838 */
            pt_fill_constants(&c, W, H, focus, 0.75f, 3.0f, -1.5707964f, 1, T, frame, numRays, (uint32_t)nodes.size(), ambient);
            if (pt_set_constants(ctx, &c)) return fail(ctx, "pt_set_constants");
            if (pt_launch(ctx)) return fail(ctx, "pt_launch");
            for (uint32_t d = 0; d < numRays; d++) {
                if (pt_trace(ctx, d, 0)) return fail(ctx, "pt_trace");
                if (d + 1 < numRays && pt_compact(ctx)) return fail(ctx, "pt_compact");
            }
            if (pt_resolve_accumulate(ctx)) return fail(ctx, "pt_resolve_accumulate");
        }
    }
    if (pt_synchronize(ctx)) return fail(ctx, "pt_synchronize");
    const double renderMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("%u frames of %u x %u x %u bounces (%s): %.1f ms, %.1f spp/s\n", frames, W, H, numRays, batch ? "one pt_render_samples call" : kept ? "one call per frame, train kept" : "dispatch sites",
           renderMs, frames / renderMs * 1e3);

    std::vector<float> acc(4 * (size_t)W * H);
    if (pt_readback_accumulation(ctx, acc.data())) return fail(ctx, "pt_readback_accumulation");
    printf("accumulation fnv1a %016llx, samples per pixel %.0f\n", (unsigned long long)fnv1a(acc.data(), acc.size() * 4), acc[3]);
    if (pt_dump_current(ctx, outDir.c_str(), frames)) return fail(ctx, "pt_dump_current");                 /* PathTracer.cpp:840-861 */
    std::vector<uint8_t> rgba(4 * (size_t)W * H);
    if (pt_tonemap(ctx, nullptr, rgba.data())) return fail(ctx, "pt_tonemap");
    if (pt_write_ppm((outDir + "/pt_host.ppm").c_str(), W, H, rgba.data())) { fprintf(stderr, "pt_host: cannot write the picture\n"); pt_ctx_destroy(ctx); return 1; }
    pt_ctx_destroy(ctx);
    return 0;
}
