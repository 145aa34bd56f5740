// merge_kernel: per-nesting-level dominance fold of the record colours, clamp, truncate.
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

// ---- merge -----------------------------------------------------------------------
// vec3.h:97-101: a colour is "less" only if all three channels are strictly less.
__device__ __forceinline__ bool all_less(float ax, float ay, float az, float bx, float by, float bz) {
    return ax < bx && ay < by && az < bz;
}

// utils.h:3-10; a NaN (which fails both comparisons) is stored as 0.
__device__ __forceinline__ uint8_t clamp_color(float v) {
    if (v > 255.f)
        v = 255.f;
    else if (v < 0.f)
        v = 0.f;
    if (v != v) return 0;
    return (uint8_t)v;
}

// Every trace_ray_boxs call (raytracing.h:74-98) folds the colours of its own
// list of boxes, starting from black, and returns the result to its caller,
// which folds it into its own colour.  The fold "replace if strictly greater
// in all channels" is not associative, so the nesting must be reproduced: lev
// is the nesting level of the list a leaf sits in, acc[lev] the colour of the
// call that is open at that level.  A box without a hit folds to black and
// black never replaces anything, so only the hit records matter.
__global__ __launch_bounds__(MERGE_BLOCK) void merge_kernel(
    DFrame fr, const DNode *__restrict__ nodes, const DUp *__restrict__ up,
    const DRec *__restrict__ recs, const float4 *__restrict__ colors,
    const uint32_t *__restrict__ first_rec, const uint8_t *__restrict__ traced, uint32_t tile_h,
    uint8_t *__restrict__ rgb) {
    extern __shared__ float merge_stack[]; // (deepest nesting + 1) * 3 * MERGE_BLOCK floats
    const uint32_t local = blockIdx.x * MERGE_BLOCK + threadIdx.x;
    if (local >= fr.rows * fr.width) return;
    float rx = 0.f, ry = 0.f, rz = 0.f;
    uint32_t r = RTXD_NONE;
    if (traced) { // only the listed tile parts were traced (tile_list_kernel): the rest hit nothing
        const uint32_t i = local % fr.width, row = local / fr.width, tiles_x = (fr.width + 15) / 16;
        const uint32_t part = ((row / tile_h) * tiles_x + (i >> 4)) * 4 + ((i >> 3) & 1) + 2 * ((row % tile_h) / (tile_h / 2));
        if (traced[part]) r = first_rec[local];
    } else {
        r = first_rec[local];
    }
    if (r != RTXD_NONE && recs[r].next == RTXD_NONE) {
        // One record (the common case): every level from the leaf's up to the top starts
        // black and takes the colour below it iff that is strictly greater than black in
        // all three channels; the condition is the same at every level.
        const float4 c = colors[r];
        if (all_less(0.f, 0.f, 0.f, c.x, c.y, c.z)) {
            rx = c.x;
            ry = c.y;
            rz = c.z;
        }
    } else if (r != RTXD_NONE) {
        // the colours of the open calls, one per nesting level: LDS, [level][channel][thread]
#define AX(l) merge_stack[((l) * 3 + 0) * MERGE_BLOCK + threadIdx.x]
#define AY(l) merge_stack[((l) * 3 + 1) * MERGE_BLOCK + threadIdx.x]
#define AZ(l) merge_stack[((l) * 3 + 2) * MERGE_BLOCK + threadIdx.x]
        AX(0) = 0.f;
        AY(0) = 0.f;
        AZ(0) = 0.f;
        uint32_t lev = 0, prev = RTXD_NONE;
        while (r != RTXD_NONE) {
            const uint32_t n = recs[r].node;
            const uint32_t depth = up[n].depth;
            if (prev != RTXD_NONE) {
                // close the calls that end between the previous leaf and this one
                uint32_t a = up[prev].parent, l = up[prev].depth;
                while (a != RTXD_NONE && nodes[a].link <= n) {
                    if (all_less(AX(l - 1), AY(l - 1), AZ(l - 1), AX(l), AY(l), AZ(l))) {
                        AX(l - 1) = AX(l);
                        AY(l - 1) = AY(l);
                        AZ(l - 1) = AZ(l);
                    }
                    l--;
                    a = up[a].parent;
                }
                lev = l;
            }
            for (uint32_t l = lev + 1; l <= depth; l++) {
                AX(l) = 0.f;
                AY(l) = 0.f;
                AZ(l) = 0.f;
            }
            const float4 c = colors[r];
            if (all_less(AX(depth), AY(depth), AZ(depth), c.x, c.y, c.z)) {
                AX(depth) = c.x;
                AY(depth) = c.y;
                AZ(depth) = c.z;
            }
            lev = depth;
            prev = n;
            r = recs[r].next;
        }
        for (uint32_t l = lev; l >= 1; l--) {
            if (all_less(AX(l - 1), AY(l - 1), AZ(l - 1), AX(l), AY(l), AZ(l))) {
                AX(l - 1) = AX(l);
                AY(l - 1) = AY(l);
                AZ(l - 1) = AZ(l);
            }
        }
        rx = AX(0);
        ry = AY(0);
        rz = AZ(0);
#undef AX
#undef AY
#undef AZ
    }
    uint8_t *out = rgb + (size_t)local * 3;
    out[0] = clamp_color(rx);
    out[1] = clamp_color(ry);
    out[2] = clamp_color(rz);
}
