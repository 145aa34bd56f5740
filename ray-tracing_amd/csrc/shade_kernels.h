// classify_kernel, shade_kernel (work lists drained by resident waves) and
// shade_few_kernel (several records per wave for at most 32 samples).
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

// ---- work lists ------------------------------------------------------------------
// The record goes to the list of its class.  A wave classifies CLASSIFY_PER_LANE x 64
// records and then reserves their entries with one atomic per class (returning atomics
// on one address are slow: one per 64 records would cost more than the rest of the kernel).
#ifndef CLASSIFY_PER_LANE
#define CLASSIFY_PER_LANE 8
#endif
__global__ __launch_bounds__(256) void classify_kernel(DFrame fr, const DRec *__restrict__ recs,
                                                        uint32_t rec_cap, float4 *__restrict__ colors,
                                                        uint32_t *__restrict__ cand_off,
                                                        DCounters *__restrict__ ctr) {
    const RecMap map = rec_map(ctr, rec_cap); // final: written by the previous kernel
    const uint32_t n_recs = map.total;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t per_wave = 64 * CLASSIFY_PER_LANE;
    for (uint32_t base = wave_id * per_wave; base < n_recs; base += n_waves * per_wave) {
        uint32_t cls[CLASSIFY_PER_LANE], leaf_size[CLASSIFY_PER_LANE];
        uint32_t lane_sum = 0;
#pragma unroll
        for (int g = 0; g < CLASSIFY_PER_LANE; g++) {
            const uint32_t i = base + g * 64 + lane;
            cls[g] = N_CLASSES; // none
            leaf_size[g] = 0;
            if (i < n_recs) {
                const DRec *rc = recs + rec_index(map, i);
                const uint32_t count = rc->count;
                bool closed = false;
                if (fr.sort_samples && count >= SORT_MIN_LEAF && count <= CAND_MAX) {
                    float dlt;
                    closed = acne_chain_delta(rc->px, rc->nx * fr.acne, fr.n_lights, dlt) &&
                             acne_chain_delta(rc->py, rc->ny * fr.acne, fr.n_lights, dlt) &&
                             acne_chain_delta(rc->pz, rc->nz * fr.acne, fr.n_lights, dlt);
                }
                cls[g] = (record_sorts(fr, count, closed) ? CLASS_SORTED : CLASS_PLAIN) + weight_group(count);
                leaf_size[g] = count;
                lane_sum += count;
            }
        }
        // a block of the candidate pool for every record, as many entries as its leaf has
        // primitives: one atomic for the wave's 512 records, a scan for the rest
        {
            uint32_t incl = lane_sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if (lane >= (uint32_t)off) incl += up;
            }
            const uint32_t total = __shfl(incl, 63);
            uint32_t first = 0;
            if (lane == 0 && total) first = atomicAdd(&ctr->cand_total, total);
            uint32_t run = __builtin_amdgcn_readfirstlane(first) + (incl - lane_sum);
#pragma unroll
            for (int g = 0; g < CLASSIFY_PER_LANE; g++) {
                if (cls[g] != N_CLASSES) cand_off[rec_index(map, base + g * 64 + lane)] = run;
                run += leaf_size[g];
            }
        }
#pragma unroll
        for (uint32_t c = 0; c < N_CLASSES; c++) {
            unsigned long long mask[CLASSIFY_PER_LANE];
            uint32_t total = 0;
#pragma unroll
            for (int g = 0; g < CLASSIFY_PER_LANE; g++) {
                mask[g] = __ballot(cls[g] == c);
                total += (uint32_t)__popcll(mask[g]);
            }
            if (total == 0) continue;
            uint32_t first = 0;
            if (lane == 0) first = atomicAdd(&ctr->cls[c].count, total);
            first = __builtin_amdgcn_readfirstlane(first);
            uint32_t *__restrict__ list = work_list(colors, rec_cap, c);
#pragma unroll
            for (int g = 0; g < CLASSIFY_PER_LANE; g++) {
                if (cls[g] == c)
                    list[first + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[g] >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)mask[g], 0u))] =
                        rec_index(map, base + g * 64 + lane);
                first += (uint32_t)__popcll(mask[g]);
            }
        }
    }
}

// ---- the shading kernel ------------------------------------------------------------
// classify_kernel has sorted the hit records into N_CLASSES work lists: by kind (shaded
// with sorted samples, or plain) and by leaf size.  Every wave drains the lists in the
// same order, large leaves first and the most numerous list last, so that the
// long records start early and the tail of the kernel is made of short ones.  A list is
// handed out through QUEUE_HEADS heads (entry e belongs to head e % QUEUE_HEADS): a wave
// starts on the head of its block (blocks b and b + 8 tend to share an XCD, hence an L2)
// and moves on when it is drained; heads only ever advance, so a wave that has seen all
// of them drained is done with the list.
template <bool SORTED, bool BOUNCE, bool IEEE>
__device__ __forceinline__ void drain_list(uint32_t cls, const DFrame &fr, const DPrim *__restrict__ prims,
                                           const DMat *__restrict__ mats, const DRec *__restrict__ recs,
                                           uint32_t rec_cap, float4 *__restrict__ colors, const CandLists &lists,
                                           DCounters *__restrict__ ctr, ShadeLds &LP, SortLds &LS, BounceLds &B,
                                           uint32_t lane, ShadeTotals &tot) {
    const uint32_t n = ctr->cls[cls].count; // final: written by classify_kernel
    if (n == 0) return;
    const uint32_t *__restrict__ list = work_list(colors, rec_cap, cls);
    // short records are popped several at a time, unless the list is too short for that
    const uint32_t batch = (cls == CLASS_LIGHTEST && n >= 8u * gridDim.x * SHADE_WAVES) ? QUEUE_BATCH : 1u;
    uint32_t q = blockIdx.x % QUEUE_HEADS, moves = 0;
    // The pop for the next turn is issued before the current records are shaded, so that the
    // returning atomic (microseconds under contention) is in flight meanwhile.
    uint32_t k_ahead = 0;
    if (lane == 0) k_ahead = atomicAdd(&ctr->head[cls][q].next, batch);
    for (;;) {
        const uint32_t k = __builtin_amdgcn_readfirstlane(k_ahead);
        if (k * QUEUE_HEADS + q >= n) {
            // This head is drained.  Look at all of them (plain loads: a failing atomic
            // per head and wave would add up) and go on with one that is not, if any; the
            // waves spread over the open heads instead of queueing on the same one.
            uint32_t next = 0xFFFFFFFFu;
            if (lane < QUEUE_HEADS)
                next = __hip_atomic_load(&ctr->head[cls][lane].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool is_open = lane < QUEUE_HEADS && next < n && next * QUEUE_HEADS + lane < n;
            const unsigned long long open = __ballot(is_open);
            if (open == 0) break; // heads only advance: nothing will ever be left for this wave
            const uint32_t pick = (blockIdx.x * SHADE_WAVES + (threadIdx.x >> 6) + moves++) % (uint32_t)__popcll(open);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(open >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)open, 0u));
            q = (uint32_t)__ffsll((long long)__ballot(is_open && rank == pick)) - 1u;
            if (lane == 0) k_ahead = atomicAdd(&ctr->head[cls][q].next, batch);
            continue;
        }
        if (lane == 0) k_ahead = atomicAdd(&ctr->head[cls][q].next, batch);
        for (uint32_t kb = k; kb < k + batch; kb++) {
            const uint32_t e = kb * QUEUE_HEADS + q;
            if (e >= n) break;
            const uint32_t ridx = list[e];
            if constexpr (SORTED)
                shade_record_sorted<IEEE>(fr, prims, mats, recs, colors, lists, LS, lane, ridx, tot);
            else
                shade_record<BOUNCE, IEEE>(fr, prims, mats, recs, colors, lists, LP, LS, B, lane, ridx, tot);
        }
    }
}

// WAVES: waves per SIMD the kernel is compiled for (its register budget); the host launches
// exactly that many per SIMD.
template <bool BOUNCE, bool IEEE, int WAVES>
__global__ __launch_bounds__(SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void shade_kernel(
    DFrame fr, const DPrim *__restrict__ prims, const DMat *__restrict__ mats,
    const DRec *__restrict__ recs, uint32_t rec_cap, float4 *__restrict__ colors, CandLists lists,
    DCounters *__restrict__ ctr) {
    __shared__ union {
        ShadeLds plain;
        SortLds sorted;
    } lds[SHADE_WAVES];
    __shared__ BounceLds levels[BOUNCE ? SHADE_WAVES : 1]; // (one unused entry in the build without bounces)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ShadeTotals tot;
    // Long records first, and the most numerous list last: the plain records with a large
    // leaf (every candidate in every pass), the sorted ones with a large leaf, the plain
    // ones with a small leaf, then the bulk, sorted with a small leaf.  The kernel ends
    // when the last record does, and only a list with many more entries than there are
    // waves keeps every wave busy until then.
    for (uint32_t step = 0; step < N_CLASSES; step++) {
        const uint32_t w = step >> 1; // weight group; within it plain before sorted
        const uint32_t cls = (step & 1u) ? CLASS_SORTED + w : CLASS_PLAIN + w;
        if (cls < CLASS_PLAIN && !BOUNCE) {
            drain_list<true, false, IEEE>(cls, fr, prims, mats, recs, rec_cap, colors, lists, ctr, lds[wave].plain,
                                          lds[wave].sorted, levels[0], lane, tot);
        } else { // (the kernel of the mirror bounces takes every list the same way)
            drain_list<false, BOUNCE, IEEE>(cls, fr, prims, mats, recs, rec_cap, colors, lists, ctr, lds[wave].plain,
                                            lds[wave].sorted, levels[BOUNCE ? wave : 0], lane, tot);
        }
    }
    auto &stat = ctr->stat[blockIdx.x & (STAT_SHARDS - 1u)];
    if (lane == 0 && tot.lit) atomicAdd(&stat.lit, (unsigned long long)tot.lit);
    if (lane == 0 && tot.tests) atomicAdd(&stat.tests_run, tot.tests);
#if defined(STEP_STATS) || defined(PHASE_TIMES)
    if (lane == 0) {
        unsigned long long *out = reinterpret_cast<unsigned long long *>(ctr->pad1);
#ifdef PHASE_TIMES
        for (int k = 0; k < 8; k++) atomicAdd(out + k, tot.st[k]);
        if (atomicMax(out + 8, tot.st[8]) < tot.st[8]) out[9] = tot.st[9]; // (the longest record; racy, good enough)
#else
        for (int k = 0; k < 10; k++) atomicAdd(out + k, tot.st[k]);
#endif
    }
#endif
    if (lane == 0 && tot.bounce_hits) {
        atomicAdd(&stat.bounce_hits, (unsigned long long)tot.bounce_hits);
        atomicAdd(&stat.bounce_prims, tot.bounce_prims);
    }
}

// ---- mirror bounces: which records have one ---------------------------------------
// With max_depth > 0 every primary hit reflects its ray (raytracing.h:51-61), but the bounce
// is traced inside the hit's own leaf and only a few in a hundred hit anything there.  One
// lane per record: the end of the acne chain, the reflected ray and the leaf's primitives
// exactly as shade_record<true> computes them, and the records whose bounce hits something
// (or whose Fresnel weight is not finite, so that "weight times black" is not zero) go
// to the work list that the kernel of the deeper levels drains.  The others keep the colour
// of their own samples.
__global__ __launch_bounds__(256) void bounce_find_kernel(DFrame fr, const DPrim *__restrict__ prims,
                                                          const DMat *__restrict__ mats,
                                                          const DRec *__restrict__ recs, uint32_t rec_cap,
                                                          float4 *__restrict__ colors,
                                                          DCounters *__restrict__ ctr) {
    const RecMap map = rec_map(ctr, rec_cap); // final: written by the trace kernel
    const uint32_t n_recs = map.total;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    uint32_t *__restrict__ list = work_list(colors, rec_cap, CLASS_PLAIN);
    for (uint32_t base = wave_id * 64; base < n_recs; base += n_waves * 64) {
        const uint32_t i = base + lane;
        bool want = false;
        uint32_t ridx = 0;
        if (i < n_recs) {
            ridx = rec_index(map, i);
            const DRec *rc = recs + ridx;
            const uint32_t pixel = rc->pixel, count = rc->count;
            const DPrim *__restrict__ leaf_prims = prims + rc->first;
            const f3 nrm = f3_make(rc->nx, rc->ny, rc->nz);
            const f3 p0 = f3_make(rc->px, rc->py, rc->pz);
            f3 o, d;
            primary_ray(fr, pixel % fr.width, pixel / fr.width, o, d);
            const f3 step = f3_scale(nrm, fr.acne);
            // where the bounce starts: the hit position after all the acne steps (closed form
            // only if all three components have one, as in shade_record)
            f3 dlt = f3_make(0.f, 0.f, 0.f), pos = p0;
            const bool cx = acne_chain_delta(p0.x, step.x, fr.n_lights, dlt.x);
            const bool cy = acne_chain_delta(p0.y, step.y, fr.n_lights, dlt.y);
            const bool cz = acne_chain_delta(p0.z, step.z, fr.n_lights, dlt.z);
            if (fr.n_lights && cx && cy && cz) {
                const float sn = (float)fr.n_lights;
                pos = f3_make(p0.x + sn * dlt.x, p0.y + sn * dlt.y, p0.z + sn * dlt.z);
            } else {
                for (uint32_t k = 0; k < fr.n_lights; k++) pos = f3_add(pos, step);
            }
            const f3 n2 = f3_scale(nrm, 2.f); // reflect(), vec3.h:103-106
            const f3 rd = f3_normalize(f3_sub(d, f3_scale(n2, f3_dot(nrm, d))));
            const f3 ro = f3_add(pos, f3_scale(rd, fr.acne));
            const float fres = rdm_fresnel(f3_dot(rd, nrm), mats[rc->mat]);
            want = !(__builtin_fabsf(fres) < __builtin_inff());
            for (uint32_t kk = 0; kk < count && !want; kk++) {
                const DPrim *pp = leaf_prims + kk;
                float t;
                if ((pp->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE)
                    want = tri_hit(ro, rd, pp->a, pp->b, pp->c, t);
                else
                    want = sph_hit(ro, rd, pp->a, pp->b.y, t);
            }
        }
        const unsigned long long mask = __ballot(want);
        if (mask) {
            uint32_t first = 0;
            if (lane == 0) first = atomicAdd(&ctr->cls[CLASS_PLAIN].count, (uint32_t)__popcll(mask));
            first = __builtin_amdgcn_readfirstlane(first);
            if (want)
                list[first + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = ridx;
        }
    }
}

// ---- shadow + shade for few light samples ---------------------------------------
// With n_lights <= G <= 32 (the reference as committed uses 1) a wave per hit record
// would leave most lanes idle, so here a wave takes 64 / G records at once: lane
// l handles sample l % G of record l / G, walks that record's leaf on its own (vector
// loads, next primitive in flight) and the G lanes of a record are summed with the
// same butterfly the one-record kernel uses, so both kernels return identical colours.
// The primary hits' own samples only: in a frame with mirror bounces the deeper levels are
// added afterwards by the bounce build of shade_kernel.
template <uint32_t G, bool IEEE>
__global__ __launch_bounds__(SHADE_BLOCK) void shade_few_kernel(
    DFrame fr, const DPrim *__restrict__ prims, const DMat *__restrict__ mats,
    const DRec *__restrict__ recs, uint32_t rec_cap, float4 *__restrict__ colors,
    DCounters *__restrict__ ctr) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_id = (blockIdx.x * SHADE_BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * SHADE_BLOCK) >> 6;
    const uint32_t per_wave = 64 / G;
    const RecMap map = rec_map(ctr, rec_cap); // final: written by the previous kernel
    const uint32_t n_recs = map.total;
    const float nl = (float)fr.n_lights;
    const uint32_t k = lane % G;
    uint32_t lit_total = 0;
    unsigned long long tests_total = 0;
    for (uint32_t base = wave_id * per_wave; base < n_recs; base += n_waves * per_wave) {
        const uint32_t flat = base + lane / G;
        const uint32_t ridx = rec_index(map, min(flat, n_recs - 1));
        const bool live = flat < n_recs && k < fr.n_lights;
        f3 acc = f3_make(0.f, 0.f, 0.f);
        bool lit = false;
        if (live) {
            const DRec *rc = recs + ridx;
            const uint32_t pixel = rc->pixel, count = rc->count;
            const DPrim *__restrict__ lp = prims + rc->first;
            const f3 nrm = f3_make(rc->nx, rc->ny, rc->nz);
            f3 p = f3_make(rc->px, rc->py, rc->pz);
            f3 o, d;
            primary_ray(fr, pixel % fr.width, pixel / fr.width, o, d);
            const f3 step = f3_scale(nrm, fr.acne);
            for (uint32_t s = 0; s <= k; s++) p = f3_add(p, step); // raytracing.h:40, k + 1 times
            f3 lpos = fr.light_pos;
            light_sample(fr.seed, pixel, rc->leaf, k, lpos.x, lpos.y);
            const f3 sd = f3_normalize(f3_sub(lpos, p)); // raytracing.h:43
            float tmin = 0.f;
            bool any = false;
            f3 a = lp->a, b = lp->b, c = lp->c;
            uint32_t tm = lp->type_mat;
            for (uint32_t q = 0; q < count; q++) {
                const DPrim *pn = lp + ((q + 1 < count) ? q + 1 : q);
                const f3 na = pn->a, nb = pn->b, nc = pn->c;
                const uint32_t ntm = pn->type_mat;
                float t;
                bool ok;
                if ((tm & 0xFFu) == RTX_PRIM_TRIANGLE)
                    ok = tri_hit(p, sd, a, b, c, t);
                else
                    ok = sph_hit(p, sd, a, b.y, t);
                if (ok && (!any || t < tmin)) {
                    tmin = t;
                    any = true;
                }
                a = na;
                b = nb;
                c = nc;
                tm = ntm;
            }
            tests_total += count;
            if (!any || out_of_scene(f3_add(p, f3_scale(sd, tmin)))) {
                const DMat mat = mats[rc->mat];
                acc = f3_add(acc, shade1<IEEE>(nrm, f3_scale(d, -1.f), sd, fr.light_color, mat, nl)); // raytracing.h:48
                lit = true;
            }
        }
        lit_total += (uint32_t)__popcll(__ballot(lit));
#pragma unroll
        for (uint32_t off = G / 2; off >= 1; off >>= 1) {
            acc.x += __shfl_xor(acc.x, (int)off);
            acc.y += __shfl_xor(acc.y, (int)off);
            acc.z += __shfl_xor(acc.z, (int)off);
        }
        if (k == 0 && flat < n_recs) colors[ridx] = make_float4(acc.x, acc.y, acc.z, 0.f);
    }
    // statistics: one atomic per wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tests_total += __shfl_xor(tests_total, off);
    auto &stat = ctr->stat[blockIdx.x & (STAT_SHARDS - 1u)];
    if (lane == 0 && lit_total) atomicAdd(&stat.lit, (unsigned long long)lit_total);
    if (lane == 0 && tests_total) atomicAdd(&stat.tests_run, tests_total);
}
