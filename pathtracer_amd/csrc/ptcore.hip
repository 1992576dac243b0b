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


/* The implementation, by concern (one translation unit: the kernels above are instantiated once). */
#include "context_impl.h"
#include "launch_impl.h"
#include "pipeline_impl.h"
#include "train_impl.h"
#include "batch_impl.h"
#include "scene_impl.h"
#include "comm_impl.h"
