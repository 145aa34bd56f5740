// trace_kernel: primary rays, box walk, nearest hit per leaf -> hit records.
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

// ---- trace ------------------------------------------------------------------
// One wave = the primary rays of an 8 x 8 pixel tile (with TRACE_SLICES = 2: of 8 x 4
// pixels, two lanes per ray).  The wave walks the pre-order box array ONCE for all its rays: `n` is wave-uniform, lane l takes part in box n when
// its own next box n_lane == n.  After an inner box the walk goes to n + 1 if any ray
// entered it and to its skip link otherwise; a ray that missed an ancestor m waits
// with n_lane = link(m), and because boxes nest, every waiting ray's link is at or
// beyond the link of the box just missed, so the uniform walk can never step past a
// waiting ray and visits exactly the union of the boxes its rays visit, each ray
// seeing its own boxes in the reference's order (raytracing.h:74-98).  Box and
// triangle data are therefore wave-uniform: scalar loads (next one already in flight),
// used as SGPR operands; only the rays live in VGPRs.
template <uint32_t TRACE_SLICES, bool QUEUED>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_kernel(
    DFrame fr, const DNode *__restrict__ nodes, uint32_t n_nodes, const DPrim *__restrict__ prims,
    DRec *__restrict__ recs, uint32_t rec_cap, uint32_t *__restrict__ first_rec,
    const uint32_t *__restrict__ tile_list, uint32_t tile_seg, DCounters *__restrict__ ctr) {
    // A wave takes 8 x (8 / TRACE_SLICES) pixels; lane l carries ray l % RAYS and, in a leaf,
    // tests every TRACE_SLICES-th pair of primitives, starting with pair l / RAYS.
    constexpr uint32_t RAYS = 64 / TRACE_SLICES, TILE_H = 16 / TRACE_SLICES;
    const uint32_t tiles_x = (fr.width + 15) / 16;
    const uint32_t n_tiles = tiles_x * ((fr.rows + TILE_H - 1) / TILE_H);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t ray = lane % RAYS, slice = lane / RAYS;
    __shared__ __attribute__((aligned(16))) float stage[TRACE_BLOCK / 64][(TRACE_STAGE / 2) * 20];
    float *S = stage[wave];
    // the block appends its hit records to one shard of the record array
    const uint32_t shard = blockIdx.x % REC_SHARDS, shard_cap = rec_cap / REC_SHARDS;
    unsigned long long leaf_prims = 0;
    // a block takes the 16 x TILE_H pixel tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    // (QUEUED: every wave takes the next tile part of a queue instead; tile_list_kernel has
    // listed the tile parts in which a ray may enter a top-level box)
    uint32_t next_tile = blockIdx.x, queue = blockIdx.x % TILE_QUEUES, moves = 0;
    for (;;) {
    uint32_t tile, sub = wave;
    if constexpr (QUEUED) {
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(&ctr->tiles[queue].next, 1u);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= ctr->tiles[queue].count) { // this queue is used up: go on with one that is not, if any
            const uint32_t taken = __hip_atomic_load(&ctr->tiles[lane].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool is_open = taken < ctr->tiles[lane].count;
            const unsigned long long open = __ballot(is_open);
            if (open == 0) break;
            const uint32_t pick = (blockIdx.x * (TRACE_BLOCK / 64) + wave + moves++) % (uint32_t)__popcll(open);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(open >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)open, 0u));
            queue = (uint32_t)__ffsll((long long)__ballot(is_open && rank == pick)) - 1u;
            continue;
        }
        const uint32_t part = tile_list[queue * tile_seg + k];
        tile = part >> 2;
        sub = part & 3u;
    } else {
        tile = next_tile;
        next_tile += gridDim.x;
        if (tile >= n_tiles) break;
    }
    const uint32_t bx = tile % tiles_x, by = tile / tiles_x;
    const uint32_t i = bx * 16 + (sub & 1) * 8 + (ray & 7);
    const uint32_t r = by * TILE_H + (sub >> 1) * (TILE_H / 2) + (ray >> 3);
    const bool valid = i < fr.width && r < fr.rows;
    if (!__any(valid) || n_nodes == 0) {
        if (valid && slice == 0) first_rec[r * fr.width + i] = RTXD_NONE;
        continue;
    }
    const uint32_t j = part_row(fr, valid ? r : 0);
    const uint32_t pixel = j * fr.width + i;
    const uint32_t local = r * fr.width + i;

    f3 o, d;
    primary_ray(fr, i, j, o, d);
    const f3 inv = f3_make(1.f / d.x, 1.f / d.y, 1.f / d.z); // intersect.h:72-74

    uint32_t head = RTXD_NONE, prev = RTXD_NONE, ordinal = 0;
    uint32_t n_lane = valid ? 0u : n_nodes; // lanes outside the frame never take part
    uint32_t n = 0;
    DNode cur = nodes[0];
    while (n < n_nodes) {
        const DNode ahead = nodes[(n + 1 < n_nodes) ? n + 1 : n]; // the likely next box
        const bool mine = n_lane == n;
        const bool hit = mine && box_hit(o, inv, cur.lb, cur.rt);
        uint32_t next;
        if (cur.info == RTXD_NONE) { // inner box
            next = __any(hit) ? n + 1 : cur.link;
            if (mine) n_lane = hit ? n + 1 : cur.link;
        } else {
            next = n + 1;
            if (mine) n_lane = n + 1;
            if (__any(hit)) {
                // leaf (intersect.h:95-132): every primitive is tested, the smallest
                // distance wins, the earliest primitive on ties
                const uint32_t first = cur.link, count = cur.info >> 16, leaf = cur.info & 0xFFFFu;
                const DPrim *__restrict__ lp = prims + first;
                float best_t = 0.f;
                uint32_t best = RTXD_NONE;
                // The leaf goes through LDS, TRACE_STAGE primitives at a time, laid out in
                // pairs (component c of primitives 2p and 2p + 1 side by side): the lanes fetch
                // one primitive each and the scan then tests two triangles per step with
                // packed arithmetic.  A heavy leaf costs a wave ~100 instructions per primitive
                // one at a time, and the waves over the densest part of a mesh set the
                // kernel's duration.
                for (uint32_t base = 0; base < count; base += TRACE_STAGE) {
                    const uint32_t m = min(count - base, (uint32_t)TRACE_STAGE);
                    wave_lds_sync(); // the previous batch has been read
                    if (lane < ((m + 1) & ~1u)) {
                        float c[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // type 0: no primitive
                        if (lane < m) {
                            const DPrim *pp = lp + base + lane;
                            const float4 *src = reinterpret_cast<const float4 *>(pp);
                            const float4 q0 = src[0], q1 = src[1]; // a.xyz b.x | b.yz c.xy
                            c[0] = q0.x, c[1] = q0.y, c[2] = q0.z, c[3] = q0.w;
                            c[4] = q1.x, c[5] = q1.y, c[6] = q1.z, c[7] = q1.w;
                            c[8] = pp->c.z;
                            c[9] = __uint_as_float(pp->type_mat & 0xFFu);
                        }
                        float *dst = S + (lane >> 1) * 20 + (lane & 1);
#pragma unroll
                        for (int k = 0; k < 10; k++) dst[k * 2] = c[k];
                    }
                    wave_lds_sync();
                    const uint32_t n_pairs = (m + 1) >> 1;
                    const float4 *S4 = reinterpret_cast<const float4 *>(S);
                    uint32_t pr = min(slice, n_pairs - 1);
                    float4 q0 = S4[pr * 5 + 0], q1 = S4[pr * 5 + 1], q2 = S4[pr * 5 + 2], q3 = S4[pr * 5 + 3],
                           q4 = S4[pr * 5 + 4];
                    for (uint32_t step = 0; step < n_pairs; step += TRACE_SLICES) {
                        const bool mine_pair = step + slice < n_pairs;
                        pr = min(step + slice, n_pairs - 1);
                        const uint32_t pn = min(pr + TRACE_SLICES, n_pairs - 1);
                        const float4 n0 = S4[pn * 5 + 0], n1 = S4[pn * 5 + 1], n2 = S4[pn * 5 + 2],
                                     n3 = S4[pn * 5 + 3], n4 = S4[pn * 5 + 4];
                        uint32_t type_a = __float_as_uint(q4.z), type_b = __float_as_uint(q4.w);
                        if (TRACE_SLICES == 1) { // one pair for the whole wave: branch on scalars
                            type_a = __builtin_amdgcn_readfirstlane(type_a);
                            type_b = __builtin_amdgcn_readfirstlane(type_b);
                        }
                        if (hit && mine_pair) {
                            const uint32_t ka = base + 2 * pr;
                            if (type_a == RTX_PRIM_TRIANGLE && type_b == RTX_PRIM_TRIANGLE) {
                                tri_hit_pk<float, v2f, TRACE_EARLY_OUT>(
                                    o.x, o.y, o.z, d.x, d.y, d.z, (v2f){q0.x, q0.y}, (v2f){q0.z, q0.w}, (v2f){q1.x, q1.y},
                                    (v2f){q1.z, q1.w}, (v2f){q2.x, q2.y}, (v2f){q2.z, q2.w}, (v2f){q3.x, q3.y},
                                    (v2f){q3.z, q3.w}, (v2f){q4.x, q4.y}, fr.wide_dets != 0, true, true,
                                    [&](bool hit_a, bool hit_b, v2f t) {
                                        if (hit_a && (best == RTXD_NONE || t.x < best_t)) {
                                            best_t = t.x;
                                            best = ka;
                                        }
                                        if (hit_b && (best == RTXD_NONE || t.y < best_t)) {
                                            best_t = t.y;
                                            best = ka + 1;
                                        }
                                    });
                            } else { // a sphere, or the odd primitive at the end of the leaf
                                float t;
                                bool ok = false;
                                if (type_a == RTX_PRIM_TRIANGLE)
                                    ok = tri_hit(o, d, f3_make(q0.x, q0.z, q1.x), f3_make(q1.z, q2.x, q2.z),
                                                 f3_make(q3.x, q3.z, q4.x), t);
                                else if (type_a == RTX_PRIM_SPHERE)
                                    ok = sph_hit(o, d, f3_make(q0.x, q0.z, q1.x), q2.x, t);
                                if (ok && (best == RTXD_NONE || t < best_t)) {
                                    best_t = t;
                                    best = ka;
                                }
                                ok = false;
                                if (type_b == RTX_PRIM_TRIANGLE)
                                    ok = tri_hit(o, d, f3_make(q0.y, q0.w, q1.y), f3_make(q1.w, q2.y, q2.w),
                                                 f3_make(q3.y, q3.w, q4.y), t);
                                else if (type_b == RTX_PRIM_SPHERE)
                                    ok = sph_hit(o, d, f3_make(q0.y, q0.w, q1.y), q2.y, t);
                                if (ok && (best == RTXD_NONE || t < best_t)) {
                                    best_t = t;
                                    best = ka + 1;
                                }
                            }
                        }
                        q0 = n0, q1 = n1, q2 = n2, q3 = n3, q4 = n4;
                    }
                }
                // the slices of a ray agree on its nearest hit: smallest distance, then earliest primitive
#pragma unroll
                for (uint32_t off = RAYS; off < 64; off <<= 1) {
                    const float t_o = __shfl_xor(best_t, (int)off);
                    const uint32_t b_o = (uint32_t)__shfl_xor((int)best, (int)off);
                    if (b_o != RTXD_NONE && (best == RTXD_NONE || t_o < best_t || (t_o == best_t && b_o < best))) {
                        best_t = t_o;
                        best = b_o;
                    }
                }
                const bool found = best != RTXD_NONE && slice == 0;
                const unsigned long long fmask = __ballot(found);
                if (fmask) {
                    // one atomic per wave reserves the records of all rays that hit
                    const uint32_t nfound = (uint32_t)__popcll(fmask);
                    uint32_t base = 0;
                    if (lane == (uint32_t)__ffsll((long long)fmask) - 1u)
                        base = atomicAdd(&ctr->shard[shard].count, nfound);
                    leaf_prims += (unsigned long long)count * nfound;
                    base = __shfl(base, __ffsll((long long)fmask) - 1);
                    if (found) {
                        const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(fmask >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)fmask, 0u));
                        const uint32_t idx = shard * shard_cap + slot;
                        const DPrim *p = lp + best;
                        const f3 pos = f3_add(o, f3_scale(d, best_t)); // intersect.h:20, :58
                        const uint32_t tm = p->type_mat;
                        f3 nrm;
                        if ((tm & 0xFFu) == RTX_PRIM_TRIANGLE)
                            nrm = p->n;
                        else
                            nrm = f3_normalize(f3_sub(pos, p->a)); // intersect.h:21
                        if (slot < shard_cap) {
                            DRec rc;
                            rc.pixel = pixel;
                            rc.node = n;
                            rc.prim = p->id;
                            rc.t = best_t;
                            rc.px = pos.x;
                            rc.py = pos.y;
                            rc.pz = pos.z;
                            rc.mat = tm >> 8;
                            rc.nx = nrm.x;
                            rc.ny = nrm.y;
                            rc.nz = nrm.z;
                            rc.next = RTXD_NONE;
                            rc.leaf = leaf;
                            rc.first = first;
                            rc.count = count;
                            rc.ordinal = ordinal;
                            recs[idx] = rc;
                            if (prev == RTXD_NONE)
                                head = idx;
                            else
                                recs[prev].next = idx; // only this lane ever touches its own chain
                            prev = idx;
                        }
                        ordinal++;
                    }
                }
            }
        }
        cur = (next == n + 1) ? ahead : nodes[(next < n_nodes) ? next : n];
        n = next;
    }
    if (valid && slice == 0) first_rec[local] = head;
    } // tiles
    if (lane == 0 && leaf_prims) atomicAdd(&ctr->stat[blockIdx.x & (STAT_SHARDS - 1u)].leaf_prims, leaf_prims);
}

// First pass of a frame when the tiles are queued: one lane per tile part (one wave's
// pixels of trace_kernel: tile * 4 + part).  A part goes to queue tile % TILE_QUEUES if it
// meets one of the frame's pixel rectangles (the projections of the top-level boxes onto the
// image, widened: a ray outside all of them enters no box and hits nothing,
// raytracing.h:74-98); flag[] tells merge_kernel which parts were traced at all.  A queue
// holds at most tile_seg = 4 * ceil(n_tiles / TILE_QUEUES) entries.
template <uint32_t TRACE_SLICES>
__global__ __launch_bounds__(256) void tile_list_kernel(DFrame fr, DTileRects rc, uint8_t *__restrict__ flag,
                                                         uint32_t *__restrict__ tile_list, uint32_t tile_seg,
                                                         DCounters *__restrict__ ctr) {
    constexpr uint32_t TILE_H = 16 / TRACE_SLICES;
    const uint32_t tiles_x = (fr.width + 15) / 16;
    const uint32_t n_tiles = tiles_x * ((fr.rows + TILE_H - 1) / TILE_H);
    const uint32_t part = blockIdx.x * blockDim.x + threadIdx.x;
    if (part >= n_tiles * 4) return;
    const uint32_t tile = part >> 2, sub = part & 3u;
    const uint32_t bx = tile % tiles_x, by = tile / tiles_x;
    const uint32_t x0 = bx * 16 + (sub & 1) * 8, r0 = by * TILE_H + (sub >> 1) * (TILE_H / 2);
    bool any = false;
    if (x0 < fr.width && r0 < fr.rows) {
        const int32_t x1 = (int32_t)min(x0 + 7, fr.width - 1);
        const int32_t g0 = (int32_t)part_row(fr, r0), g1 = (int32_t)part_row(fr, min(r0 + TILE_H / 2 - 1, fr.rows - 1));
        for (uint32_t k = 0; k < rc.n; k++)
            any = any || !(x1 < rc.x0[k] || (int32_t)x0 > rc.x1[k] || g1 < rc.y0[k] || g0 > rc.y1[k]);
    }
    flag[part] = any ? 1 : 0;
    if (any) {
        const uint32_t q = tile % TILE_QUEUES;
        const uint32_t k = atomicAdd(&ctr->tiles[q].count, 1u);
        tile_list[q * tile_seg + k] = part;
    }
}
