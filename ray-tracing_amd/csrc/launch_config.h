// Block sizes and occupancy knobs of the kernels, and the wave-level LDS fence.
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

#define TRACE_BLOCK 256
#ifndef TRACE_EARLY_OUT
#define TRACE_EARLY_OUT true
#endif
#ifndef TRACE_STAGE
#define TRACE_STAGE 64 // primitives of a leaf staged in LDS at a time (40 B each, per wave)
#endif
// trace_kernel<SLICES>: lanes per primary ray (they share the scan of a leaf).  Two halve
// the longest chain of dependent work in the kernel (the leaves over the densest part of a
// mesh) and double the number of waves that walk the boxes: a gain where leaves are large,
// a loss where they are small.  rtx_scene_create picks by the mean leaf size.
#ifndef TRACE_SLICE_MIN_LEAF
#define TRACE_SLICE_MIN_LEAF 20u
#endif
#ifndef TRACE_QUEUE
#define TRACE_QUEUE 1 // hand the tiles that can hit something out through queues
#endif
#ifndef TRACE_SLICES_QUEUED
#define TRACE_SLICES_QUEUED 2u // lanes per primary ray in scenes with large leaves when the tiles are queued (4: faster only on small shares of a frame)
#endif
#ifndef TRACE_QUEUE_MAX_COVER
#define TRACE_QUEUE_MAX_COVER 0.4 // queues only if the boxes' rectangles cover less of the frame than this
#endif
#ifndef TRACE_QUEUE_BLOCKS_PER_CU
#define TRACE_QUEUE_BLOCKS_PER_CU 4 // resident blocks of the queued trace_kernel (115 VGPRs: 4 waves per SIMD)
#endif
#ifndef TRACE_BLOCKS_PER_CU
#define TRACE_BLOCKS_PER_CU 8 // resident blocks of trace_kernel per CU (8 waves per SIMD)
#endif
#define SHADE_BLOCK 256
#define SHADE_WAVES (SHADE_BLOCK / 64)
#define MERGE_BLOCK 128
#define MAX_NEST 24 // deepest box nesting the merge handles (checked at scene creation)

// LDS written by some lanes of a wave is about to be read by others of the same wave.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
