/*
 * trace_diagnostics.h - what the seven PT_DIAG_* hooks of k_trace_flat (trace_kernels.h) expand to in a diagnostic build
 * (-DPT_DIAGNOSTICS plus one or more of the switches below).  Never part of the product library: pathtracer_amd/build.py does
 * not define PT_DIAGNOSTICS; scripts/build_variant.sh builds the variants, scripts/cmp_variants.sh runs them beside the product.
 *
 *   PT_EXPERIMENT_EXTRA_VALU=N  N dependent v_add_f32 more in every step: how much does the step's time depend on its
 *                               vector instruction count?
 *   PT_EXPERIMENT_EXTRA_LOAD    one more 16 B access to the step's line: what does a fifth access cost?
 *   PT_EXPERIMENT_TIMING        per-wavefront cycle sums (fetch wait, whole loop, steps, lane-steps) in the stats words,
 *                               which nothing else reads in that build (the kernel returns before the counting code)
 *   PT_EXPERIMENT_SAMELINE      counting build: the box-test counter counts the steps whose record lies in the 128 B line of
 *                               the previous step's record instead of box tests
 *   PT_SYNC_REFILL              the refill waits for its own loads instead of riding on the step's wait
 */
#pragma once

#if defined(PT_EXPERIMENT_SAMELINE) && defined(PT_EXPERIMENT_TIMING)
#define PT_DIAG_STATE uint32_t prevLine = 0xFFFFFFFFu; unsigned long long tFetch = 0, tIter = 0, nIter = 0, nLaneSteps = 0, t0 = __builtin_amdgcn_s_memtime();
#elif defined(PT_EXPERIMENT_SAMELINE)
#define PT_DIAG_STATE uint32_t prevLine = 0xFFFFFFFFu;
#elif defined(PT_EXPERIMENT_TIMING)
#define PT_DIAG_STATE unsigned long long tFetch = 0, tIter = 0, nIter = 0, nLaneSteps = 0, t0 = __builtin_amdgcn_s_memtime();
#else
#define PT_DIAG_STATE
#endif

#ifdef PT_EXPERIMENT_SAMELINE
#define PT_DIAG_COUNT_ROOT_BOX prevLine = 0xFFFFFFFFu;
#define PT_DIAG_COUNT_BOX_PAIR
#define PT_DIAG_SAMELINE_STEP if (COUNT) { if ((off >> 7) == prevLine) nBox++; prevLine = off >> 7; }
#else
#define PT_DIAG_COUNT_ROOT_BOX nBox++;
#define PT_DIAG_COUNT_BOX_PAIR nBox += 2;
#define PT_DIAG_SAMELINE_STEP
#endif

#ifdef PT_SYNC_REFILL
#define PT_DIAG_AFTER_REFILL if (fresh) { start_ray(w); fresh = false; }
#else
#define PT_DIAG_AFTER_REFILL
#endif

#ifdef PT_EXPERIMENT_TIMING
#define PT_DIAG_BEFORE_FETCH unsigned long long ta = __builtin_amdgcn_s_memtime();
#define PT_DIAG_TIMING_STEP { unsigned long long tb_ = __builtin_amdgcn_s_memtime(); unsigned long long am = __ballot(true); \
                              if (lane == __ffsll(am) - 1) { tFetch += tb_ - ta; nIter++; nLaneSteps += __popcll(am); } }
#define PT_DIAG_AT_EXIT { tIter = __builtin_amdgcn_s_memtime() - t0; \
                          atomicAdd(a.stats + 0, tFetch); atomicAdd(a.stats + 2, nIter); atomicAdd(a.stats + 3, nLaneSteps); \
                          if (lane == 0) atomicAdd(a.stats + 1, tIter); return; }
#else
#define PT_DIAG_BEFORE_FETCH
#define PT_DIAG_TIMING_STEP
#define PT_DIAG_AT_EXIT
#endif

#ifdef PT_EXPERIMENT_EXTRA_LOAD
#define PT_DIAG_EXTRA_LOAD { u32x4 q4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 32, 0, 0); asm volatile("" :: "v"(q4)); }
#else
#define PT_DIAG_EXTRA_LOAD
#endif

#ifdef PT_EXPERIMENT_EXTRA_VALU
#define PT_DIAG_EXTRA_VALU { float dummy = __uint_as_float(q0.x); \
                             _Pragma("unroll") for (int k = 0; k < PT_EXPERIMENT_EXTRA_VALU; k++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dummy)); \
                             asm volatile("" :: "v"(dummy)); }
#else
#define PT_DIAG_EXTRA_VALU
#endif

#define PT_DIAG_AFTER_FETCH PT_DIAG_EXTRA_LOAD PT_DIAG_TIMING_STEP PT_DIAG_EXTRA_VALU PT_DIAG_SAMELINE_STEP
