// screen_kernel: the candidate lists of the hit records, ahead of the shading kernel.
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

// ---- candidate lists ------------------------------------------------------------------
// The shading kernel tests a record's shadow rays only against the leaf primitives that
// survive the record's screens (shade_record.h, shade_sorted.h).  Screening is per record,
// not per sample, and needs none of the sampling loop's state, so it runs here: the record's
// start segment, its screens with the grazing guard, and the survivors (with their light-cell
// masks where the record will sort its samples) written to the record's block of the pool.
// classify_kernel has given every record a block of as many entries as its leaf has
// primitives.  The shading kernel then carries no screening code (fewer registers, next to no
// scratch), and this kernel can spend registers freely (it is bound by vector issue at three
// waves per SIMD).  Deeper levels of mirror bounces start from other points and screen for
// themselves in the bounce build of the shading kernel.
//
// A wave takes SCREEN_BATCH records at a time, in three phases:
//  1. one lane per record: what depends on the record alone (closed form of the start points,
//     start segment, ray fan) goes to LDS - the same for all lanes of a wave if it were computed
//     record by record, two hundred vector instructions each time;
//  2. the leaves of the batch as one stream of primitives, a primitive per lane whatever record
//     it belongs to: screens, guard, survivors to their record's block of the pool, the guard's
//     (triangle, sample) pairs to one list for the whole batch;
//  3. the listed pairs, one per lane whatever record they belong to (a record lists a dozen
//     on average: checked record by record most lanes would idle), then the list lengths.

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
#ifndef SCREEN_BATCH
#define SCREEN_BATCH 16u // records a wave takes at a time (at most 64: one lane per record in phase 1)
#endif
#ifndef SCREEN_GRAZE_LIST
#define SCREEN_GRAZE_LIST 768u // pairs a wave can list before they are checked (a step lists up to 64 * GRAZE_KMAX)
#endif
static_assert(SCREEN_GRAZE_LIST > 64u * GRAZE_KMAX, "a step's pairs must fit behind what is already listed");
#ifndef SCREEN_WAVES
#define SCREEN_WAVES 5 // waves per SIMD the kernel is compiled for
#endif

// What phase 1 leaves for a record (LDS, 176 B: a multiple of 16, and lanes r and r + 8 do not
// meet in a bank).
#define SCREEN_REC_CLOSED 1u   // start points in closed form
#define SCREEN_REC_SORTED 2u   // the record sorts its samples: its list carries cell masks
#define SCREEN_REC_NO_BLOCK 4u // the pool is too small for this frame: no list
#define SCREEN_REC_OPEN 8u     // shaded without screens (set by phase 3)
#define SCREEN_REC_PAIRS 16u   // (measurement build) the record listed pairs
struct alignas(16) ScreenRec {
    RayFan fan;
    float oz_first, oz_last; // (the fan keeps the start points' z times sigma)
    uint32_t flags;
    uint32_t count, first, off, ridx; // leaf size, first primitive, block in the pool, record index
    f3 p0, step, dlt;
    uint32_t pixel, leaf;
    uint32_t pad[6];
};
static_assert(sizeof(ScreenRec) == 176, "ScreenRec: 44 words");

struct ScreenLds {
    ScreenRec rec[SCREEN_BATCH];
    uint32_t pair[SCREEN_GRAZE_LIST]; // primitive offset in the leaf | sample << 16
    uint8_t slot[SCREEN_GRAZE_LIST];  // the pair's record: index in rec[]
    uint32_t end[SCREEN_BATCH];       // primitives of records 0 .. r of the batch
    uint32_t n_pairs;                 // listed and not yet checked
    uint32_t pad[3];
};

// Phase 3: the float test's verdict on pairs [0, n) of the list.  A record one of whose pairs
// is accepted is marked SCREEN_REC_OPEN.
__device__ __forceinline__ void screen_resolve(const DFrame &fr, const DPrim *__restrict__ prims, ScreenLds &L, uint32_t n,
                                               uint32_t lane) {
    const float per_m = 4.f / fr.cull_margin;
    for (uint32_t base = 0; base < n; base += 64) {
        bool open = false;
        f3 o = fr.light_pos, D = fr.light_pos;
        const DPrim *__restrict__ p = prims;
        uint32_t s = 0;
        if (base + lane < n) {
            const uint32_t w = L.pair[base + lane], k0 = w >> 16;
            s = L.slot[base + lane];
            const ScreenRec &c = L.rec[s];
            p = prims + c.first + (w & 0xFFFFu);
            open = graze_pair_open(fr, per_m, p, (c.flags & SCREEN_REC_CLOSED) != 0u, c.p0, c.step, c.dlt, c.pixel, c.leaf, k0,
                                   k0, o, D);
        }
        if (__any(open)) {
            if (open && graze_pair_hit(p, o, D)) atomicOr(&L.rec[s].flags, SCREEN_REC_OPEN);
        }
    }
    wave_lds_sync();
}

__global__ __launch_bounds__(SCREEN_BLOCK) __attribute__((amdgpu_waves_per_eu(SCREEN_WAVES, SCREEN_WAVES)))
void screen_kernel(DFrame fr, const DPrim *__restrict__ prims,
                                                              const DRec *__restrict__ recs, uint32_t rec_cap,
                                                              const uint32_t *__restrict__ cand_off,
                                                              uint32_t *__restrict__ cand_n,
                                                              uint32_t *__restrict__ pool, uint32_t pool_cap,
                                                              float4 *__restrict__ colors,
                                                              DCounters *__restrict__ ctr) {
    __shared__ ScreenLds lds[SCREEN_BLOCK / 64];
    const RecMap map = rec_map(ctr, rec_cap); // final: written by the trace kernel
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const float nl = (float)fr.n_lights;
    ScreenLds &L = lds[threadIdx.x >> 6];
#ifdef STEP_STATS
    unsigned long long st_pairs = 0, st_recs = 0, st_open = 0, st_prims = 0;
#endif
    // Batch b takes records b, b + n_batches, b + 2 n_batches, ...: neighbouring records hit the
    // same leaves, and a batch of 32 records of the densest leaf would last as long as the rest of
    // the kernel (measured: 0.58 ms against 0.46 on the 4K buddha frame).
    const uint32_t n_batches = (map.total + SCREEN_BATCH - 1u) / SCREEN_BATCH;
    for (uint32_t batch = wave_id; batch < n_batches; batch += n_waves) {
        const uint32_t nb = min(SCREEN_BATCH, (map.total - batch + n_batches - 1u) / n_batches);
        // ---- phase 1: a record per lane ----
        uint32_t my_prims = 0; // this lane's record: primitives it sends through phase 2
        if (lane < nb) {
            ScreenRec c;
            c.ridx = rec_index(map, lane * n_batches + batch);
            const DRec rc = recs[c.ridx];
            c.count = rc.count, c.first = rc.first, c.pixel = rc.pixel, c.leaf = rc.leaf;
            c.off = cand_off[c.ridx];
            c.p0 = f3_make(rc.px, rc.py, rc.pz);
            c.step = f3_scale(f3_make(rc.nx, rc.ny, rc.nz), fr.acne); // raytracing.h:40
            c.dlt = f3_make(0.f, 0.f, 0.f);
            const bool cx = acne_chain_delta(c.p0.x, c.step.x, fr.n_lights, c.dlt.x);
            const bool cy = acne_chain_delta(c.p0.y, c.step.y, fr.n_lights, c.dlt.y);
            const bool cz = acne_chain_delta(c.p0.z, c.step.z, fr.n_lights, c.dlt.z);
            const bool closed = cx && cy && cz;
            f3 o_first, o_last;
            const float margin = screen_segment(closed, c.p0, c.step, c.dlt, nl, fr.cull_margin, o_first, o_last);
            c.fan = make_fan(o_first, o_last, fr.light_pos.z, margin);
            c.oz_first = o_first.z, c.oz_last = o_last.z;
            // (a block beyond the pool: the host sees the total, grows the pool and renders the
            // frame again; until then the record has no list)
            c.flags = (closed ? SCREEN_REC_CLOSED : 0u) | (record_sorts(fr, c.count, closed) ? SCREEN_REC_SORTED : 0u) |
                      (((unsigned long long)c.off + c.count > pool_cap) ? SCREEN_REC_NO_BLOCK : 0u);
            L.rec[lane] = c;
            my_prims = (c.flags & SCREEN_REC_NO_BLOCK) ? 0u : c.count;
            // the direction the hit is seen from (raytracing.h:47: minus the primary ray's), left where
            // the shading kernel will put the record's colour: there it is the same in all 64 lanes,
            // an integer division and a normalisation per record
            f3 ro, rd;
            primary_ray(fr, rc.pixel % fr.width, rc.pixel / fr.width, ro, rd);
            colors[c.ridx] = make_float4(-rd.x, -rd.y, -rd.z, 0.f);
        }
        // the leaves of the batch one after the other: end[r] = primitives of records 0 .. r
        // (a record without a block in the pool takes no part)
        {
            uint32_t incl = my_prims;
#pragma unroll
            for (int off = 1; off < (int)SCREEN_BATCH; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off);
                if (lane >= (uint32_t)off) incl += up;
            }
            if (lane < SCREEN_BATCH) L.end[lane] = incl;
        }
        if (lane == 0) L.n_pairs = 0u;
        wave_lds_sync();
        // ---- phase 2: the batch's primitives as one stream, a primitive per lane ----
        // Lane l of a step takes primitive f0 + l of the stream, whatever record it belongs to (a
        // leaf has 20 to 70 primitives on average: record by record a step would fill a third to
        // a half of the lanes).  How far outside the rays' pyramid the primitive lies
        // (fan_separation; records with samples in index order use the same planes), for a record
        // that sorts its samples the light cells it can be hit from, the grazing guard for what is
        // dropped or left out of cells, and the survivors appended to their record's block.
        const uint32_t total_prims = L.end[nb - 1u];
        uint32_t my_n_cand = 0; // lane r: entries of record r's list so far
        uint32_t n_listed = 0;  // pairs listed and not yet checked
        uint32_t r_lo = 0;      // first record with primitives at or beyond f0
        for (uint32_t f0 = 0; f0 < total_prims; f0 += 64) {
            if (n_listed > SCREEN_GRAZE_LIST - 64u * GRAZE_KMAX) { // room for whatever this step lists
#ifdef STEP_STATS
                st_pairs += n_listed;
#endif
                screen_resolve(fr, prims, L, n_listed, lane);
                if (lane == 0) L.n_pairs = 0u;
                n_listed = 0;
                wave_lds_sync();
            }
            while (L.end[r_lo] <= f0) r_lo++;
            r_lo = __builtin_amdgcn_readfirstlane(r_lo); // (the same in every lane)
            const uint32_t f = f0 + lane;
            const bool valid = f < total_prims;
            uint32_t r = r_lo;
            if (valid)
                while (L.end[r] <= f) r++;
            const uint32_t r_hi = __builtin_amdgcn_readlane(r, (int)(min(total_prims - f0, 64u) - 1u));
            bool keep = false;
            uint32_t cells = 0xFFFFu, code = 0, first = 0, cnt = 0, t = 0;
            uint32_t *__restrict__ out = pool;
            if (valid) {
                const ScreenRec &c = L.rec[r];
                t = f - (r ? L.end[r - 1u] : 0u);
                out = pool + c.off;
                const DPrim *__restrict__ p = prims + c.first + t;
                keep = true;
                if ((p->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) {
                    const RayFan fan = c.fan;
                    const bool closed = (c.flags & SCREEN_REC_CLOSED) != 0u, sorted = (c.flags & SCREEN_REC_SORTED) != 0u;
                    float sep;
                    if (fr.sort_samples) { // (records that will not sort their samples ignore the cells)
                        fan_screen(fan, p->a, p->b, p->c, sep, cells);
                        if (!sorted) cells = 0xFFFFu;
                    } else {
                        sep = fan_separation(fan, p->a, p->b, p->c);
                    }
                    keep = !(sep > fan.margin);
                    if (!keep) cells = 0xFFFFu; // (dropped altogether: the guard looks at it once)
#ifdef NO_GRAZE
                    const bool screened = false; // (measurement build only: the screens without their guard)
#else
                    const bool screened = fan.margin < 1.0e38f;
#endif
                    if (screened && (!keep || cells != 0xFFFFu)) {
                        // (left out of a cell: by the margin at least)
                        const GrazeCtx G = graze_ctx(closed, f3_make(fan.ox[0], fan.oy[0], c.oz_first),
                                                     f3_make(fan.ox[1], fan.oy[1], c.oz_last), fr.n_lights, fr.cull_margin,
                                                     fan.margin);
                        code = graze_quick(G, p, keep ? fan.margin : sep, first, cnt);
                        if (code == 2u) {
                            keep = true;
                            cells = 0xFFFFu;
                        }
                    }
                }
            }
            if (code == 1u) { // the pairs of this lane's triangle, in any order (few lanes have any)
                const uint32_t pos = atomicAdd(&L.n_pairs, cnt);
                for (uint32_t q = 0; q < cnt; q++) {
                    L.pair[pos + q] = t | ((first + q) << 16);
                    L.slot[pos + q] = (uint8_t)r;
                }
#ifdef STEP_STATS
                atomicOr(&L.rec[r].flags, SCREEN_REC_PAIRS);
#endif
            }
            // the survivors, record by record of those the step touches, in stream order
            for (uint32_t rr = r_lo; rr <= r_hi; rr++) {
                const unsigned long long mask = __ballot(keep && r == rr);
                const uint32_t n_rr = __builtin_amdgcn_readlane(my_n_cand, (int)rr);
                if (keep && r == rr)
                    out[n_rr + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] =
                        t | (cells << 16);
                if (lane == rr) my_n_cand += (uint32_t)__popcll(mask);
            }
            wave_lds_sync();
            n_listed = L.n_pairs;
        }
#ifdef STEP_STATS
        st_prims += total_prims, st_pairs += n_listed;
#endif
        // ---- phase 3: the listed pairs, then every record's list length ----
        if (n_listed) screen_resolve(fr, prims, L, n_listed, lane);
        uint32_t my_flags = 0, my_count = 0, my_ridx = 0;
        if (lane < nb) my_flags = L.rec[lane].flags, my_count = L.rec[lane].count, my_ridx = L.rec[lane].ridx;
        // a pair the screens would drop is accepted by the float test: that record without screens
        unsigned long long open = __ballot((my_flags & SCREEN_REC_OPEN) != 0u);
#ifdef STEP_STATS
        st_open += (unsigned long long)__popcll(open);
        st_recs += (unsigned long long)__popcll(__ballot((my_flags & SCREEN_REC_PAIRS) != 0u));
#endif
        while (open) {
            const uint32_t r = (uint32_t)__builtin_ctzll(open);
            open &= open - 1ull;
            uint32_t *__restrict__ out = pool + L.rec[r].off;
            const uint32_t count = L.rec[r].count;
            for (uint32_t t = lane; t < count; t += 64) out[t] = t | 0xFFFF0000u;
        }
        if (lane < nb) {
            uint32_t n = (my_flags & SCREEN_REC_OPEN) ? my_count : my_n_cand;
            if (my_flags & SCREEN_REC_NO_BLOCK) n = 0u;
            cand_n[my_ridx] = n | (((my_flags & SCREEN_REC_SORTED) && n >= SORT_MIN_CAND) ? CAND_N_SORT : 0u);
        }
        wave_lds_sync(); // rec[] is rewritten by the next batch
    }
#ifdef STEP_STATS
    if (lane == 0) {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(ctr->pad1);
        atomicAdd(dst + 10, st_pairs), atomicAdd(dst + 11, st_recs), atomicAdd(dst + 12, st_open), atomicAdd(dst + 13, st_prims);
    }
#endif
}
