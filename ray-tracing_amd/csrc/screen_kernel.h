// screen_kernel: the candidate lists of the hit records, ahead of the shading kernel.
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

// ---- candidate lists ------------------------------------------------------------------
// The shading kernel tests a record's shadow rays only against the leaf primitives that
// survive the record's screens (shade_record.h, shade_sorted.h).  Screening is per record,
// not per sample, and needs none of the sampling loop's state, so it runs here, one wave per
// record, lanes over the leaf's primitives: the record's start segment, its screens with the
// grazing guard, and the survivors (with their light-cell masks where the record will sort
// its samples) written to the record's block of the pool.  classify_kernel has given every
// record a block of as many entries as its leaf has primitives.  The shading kernel then
// carries no screening code (fewer registers, next to no scratch), and this kernel can spend
// registers freely (it is bound by vector issue at three waves per SIMD).  Deeper levels of
// mirror bounces start from other points and screen for themselves in the bounce build of
// the shading kernel.

// Whether a record is shaded with sorted samples: classify_kernel, screen_kernel and the
// shading kernel must agree.  closed: all three components of the start points have the
// closed form (acne_chain_delta).
__device__ __forceinline__ bool record_sorts(const DFrame &fr, uint32_t count, bool closed) {
    return fr.sort_samples && count >= SORT_MIN_LEAF && count <= CAND_MAX && closed;
}

// graze_check with few branches, for the one pass over the leaf below: the same bound (rho from
// the same formula, with |.|_1 for the distance sigma, which only enlarges it), the same three
// answers.  first / count are valid for answer 1.
__device__ __forceinline__ uint32_t graze_quick(const GrazeCtx &g, const DPrim *__restrict__ p, float sep,
                                                uint32_t &first, uint32_t &count) {
    const f3 s1 = f3_sub(g.o1, p->a), sn = f3_sub(g.on, p->a);
    const float sig = fmaxf(__builtin_fabsf(s1.x) + __builtin_fabsf(s1.y) + __builtin_fabsf(s1.z),
                            __builtin_fabsf(sn.x) + __builtin_fabsf(sn.y) + __builtin_fabsf(sn.z));
    const float m = sep - g.drift - 2.f * g.m4;
    const float rho =
        GRAZE_EPS * sig * (p->shape * (29.f * (sig + p->ell) * __builtin_amdgcn_rcpf(m) + 19.4f) + 8.f) * 1.05f + g.drift + 1.0e-30f;
    const float h1 = f3_dot(p->n, s1), hn = f3_dot(p->n, sn);
    // bounded at all?  (a NaN anywhere fails a comparison and keeps the triangle)
    const bool bounded = (GRAZE_EPS * 5.f * sig <= g.m4) && (m >= 2.f * g.m4) && (rho < 3.0e38f) && (h1 - hn == h1 - hn);
    const bool clear = fminf(__builtin_fabsf(h1), __builtin_fabsf(hn)) > rho && h1 * hn > 0.f;
    // samples whose start point is within rho of the plane: h1 + (k0 + k_add) * dh, |.| <= rho
    const float dh = (hn - h1) * g.per_k_den;
    const float per_dh = __builtin_amdgcn_rcpf(dh);
    const float x = (-rho - h1) * per_dh - g.k_add, y = (rho - h1) * per_dh - g.k_add;
    const float slack = 0.02f + 1.0e-5f * fmaxf(__builtin_fabsf(x), __builtin_fabsf(y));
    float lo = fmaxf(__builtin_ceilf(fminf(x, y) - slack), 0.f);
    float hi = fminf(__builtin_floorf(fmaxf(x, y) + slack), g.steps);
    if (dh == 0.f) { // every start point at the same distance: all of them, or none
        lo = 0.f;
        hi = (__builtin_fabsf(h1) <= rho) ? g.steps : -1.f;
    }
    const float n = hi - lo + 1.f;
    first = (uint32_t)fmaxf(lo, 0.f);
    count = (uint32_t)fminf(fmaxf(n, 0.f), (float)GRAZE_KMAX);
    if (!bounded) return 2u;
    if (clear || n <= 0.f) return 0u;
    return (n <= (float)GRAZE_KMAX) ? 1u : 2u; // (a NaN count keeps the triangle)
}

#define SCREEN_BLOCK 256
#define SCREEN_GRAZE_LIST 768u // the guard's (triangle, sample) pairs of one record; more: no screens
#ifndef SCREEN_WAVES
#define SCREEN_WAVES 3 // waves per SIMD the kernel is compiled for
#endif
__global__ __launch_bounds__(SCREEN_BLOCK) __attribute__((amdgpu_waves_per_eu(SCREEN_WAVES, SCREEN_WAVES)))
void screen_kernel(DFrame fr, const DPrim *__restrict__ prims,
                                                              const DRec *__restrict__ recs, uint32_t rec_cap,
                                                              const uint32_t *__restrict__ cand_off,
                                                              uint32_t *__restrict__ cand_n,
                                                              uint32_t *__restrict__ pool, uint32_t pool_cap,
                                                              DCounters *__restrict__ ctr) {
    __shared__ uint32_t graze_lists[SCREEN_BLOCK / 64][SCREEN_GRAZE_LIST];
    __shared__ uint32_t graze_counts[SCREEN_BLOCK / 64]; // pairs listed for the wave's current record
    const RecMap map = rec_map(ctr, rec_cap); // final: written by the trace kernel
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const float nl = (float)fr.n_lights;
    uint32_t *__restrict__ graze_list = graze_lists[wave];
    uint32_t *graze_count = &graze_counts[wave];
#ifdef STEP_STATS
    unsigned long long st_pairs = 0, st_recs = 0, st_open = 0, st_prims = 0;
#endif
    for (uint32_t flat = wave_id; flat < map.total; flat += n_waves) {
        const uint32_t ridx = rec_index(map, flat);
        const DRec rc = recs[ridx];
        const uint32_t count = rc.count, off = cand_off[ridx];
        if ((unsigned long long)off + count > pool_cap) {
            // the pool is too small for this frame: the host sees the total, grows the pool and
            // renders the frame again; until then the record has no list
            if (lane == 0) cand_n[ridx] = 0u;
            continue;
        }
        const DPrim *__restrict__ leaf_prims = prims + rc.first;
        const f3 nrm = f3_make(rc.nx, rc.ny, rc.nz), p0 = f3_make(rc.px, rc.py, rc.pz);
        const f3 step = f3_scale(nrm, fr.acne); // raytracing.h:40
        f3 dlt = f3_make(0.f, 0.f, 0.f);
        const bool closed = acne_chain_delta3(lane, p0, step, fr.n_lights, dlt);
        f3 o_first, o_last;
        const float margin = screen_segment(closed, p0, step, dlt, nl, fr.cull_margin, o_first, o_last);
        const bool sorted = record_sorts(fr, count, closed);
        uint32_t *__restrict__ out = pool + off;
        // One pass over the leaf, a primitive per lane: how far outside the rays' pyramid it lies
        // (fan_separation; records with samples in index order use the same planes), for a record
        // that sorts its samples the light cells it can be hit from, the grazing guard for what is
        // dropped or left out of cells, and the survivors appended to the record's block.
#ifdef NO_GRAZE
        const bool screened = false; // (measurement build only: the screens without their guard)
#else
        const bool screened = margin < 1.0e38f;
#endif
        const RayFan fan = make_fan(o_first, o_last, fr.light_pos.z, margin);
        const GrazeCtx G = graze_ctx(closed, o_first, o_last, fr.n_lights, fr.cull_margin, margin);
        uint32_t n_cand = 0;
        if (lane == 0) *graze_count = 0u;
        wave_lds_sync();
        for (uint32_t base = 0; base < count; base += 64) {
            const uint32_t t = base + lane;
            bool keep = false;
            uint32_t cells = 0xFFFFu, code = 0, first = 0, cnt = 0;
            if (t < count) {
                const DPrim *__restrict__ p = leaf_prims + t;
                keep = true;
                if ((p->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) {
                    float sep;
                    if (sorted) {
                        fan_screen(fan, p->a, p->b, p->c, sep, cells);
                    } else {
                        sep = fan_separation(fan, p->a, p->b, p->c);
                    }
                    keep = !(sep > fan.margin);
                    if (!keep) cells = 0xFFFFu; // (dropped altogether: the guard looks at it once)
                    if (screened && (!keep || cells != 0xFFFFu)) {
                        // (left out of a cell: by the margin at least)
                        code = graze_quick(G, p, keep ? fan.margin : sep, first, cnt);
                        if (code == 2u) {
                            keep = true;
                            cells = 0xFFFFu;
                        }
                    }
                }
            }
            if (code == 1u) { // the pairs of this lane's triangle, in any order (few lanes have any)
                const uint32_t pos = atomicAdd(graze_count, cnt);
                for (uint32_t r = 0; r < cnt; r++)
                    if (pos + r < SCREEN_GRAZE_LIST) graze_list[pos + r] = t | ((first + r) << 16);
            }
            const unsigned long long mask = __ballot(keep);
            if (keep)
                out[n_cand + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] =
                    t | (cells << 16);
            n_cand += (uint32_t)__popcll(mask);
        }
        wave_lds_sync();
        const uint32_t n_graze = *graze_count;
#ifdef STEP_STATS
        st_pairs += n_graze, st_recs += n_graze != 0, st_prims += count;
#endif
        if (n_graze) {
            if (graze_resolve(fr, leaf_prims, graze_list, SCREEN_GRAZE_LIST, n_graze, closed, p0, step, dlt, rc.pixel, rc.leaf,
                              0u, lane)) {
                // a pair the screens would drop is accepted by the float test: this record without screens
                for (uint32_t t = lane; t < count; t += 64) out[t] = t | 0xFFFF0000u;
                n_cand = count;
#ifdef STEP_STATS
                st_open++;
#endif
            }
            wave_lds_sync();
        }
        if (lane == 0) cand_n[ridx] = n_cand | ((sorted && n_cand >= SORT_MIN_CAND) ? CAND_N_SORT : 0u);
    }
#ifdef STEP_STATS
    if (lane == 0) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(ctr->pad1);
        atomicAdd(dst + 10, st_pairs), atomicAdd(dst + 11, st_recs), atomicAdd(dst + 12, st_open), atomicAdd(dst + 13, st_prims);
    }
#endif
}
