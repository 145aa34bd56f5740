// Shading of one hit record with its light samples sorted by light cell.
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

// ---- shadow + shade with sorted light samples -------------------------------------
// Second way to shade a hit record, for the records classify_kernel puts on the
// "sorted" work lists (large leaf, several passes of samples, closed-form origins):
// the same result as shade_record, fewer tests.
//
// Take a line x = X in the light plane and a start point o: the plane through both
// splits space into the side of the rays that aim at x > X and the side of those that
// aim at x < X.  g_X(o, v) = s * ((z_l - o.z) * (v.x - o.x) - (X - o.x) * (v.z - o.z)),
// s = sign(z_l - o.z), is positive on the first side, proportional to the distance from
// the plane, and affine in o and in v.  So if all three vertices have g_X < -margin for
// the first and for the last start point, the whole triangle is on the "x < X" side for
// every start point, and no ray that aims at x >= X can hit it; likewise with > and <=,
// and for lines y = Y.  (With X = 250 and 750 these are the pyramid faces of
// shade_kernel's screen.)
//
// The screen first drops the triangles beyond an outer line (250 or 750): they are
// outside the pyramid of all the record's rays.  Five lines per axis (250, 375, ..., 750)
// cut the light into 4 x 4 cells; every remaining triangle then gets the set of cells
// whose rays may hit it, as a 16-bit mask.  The bits are numbered along a Hilbert curve through the cells,
// so that a run of consecutive bits is a compact group of cells.  The light samples of
// the record are sorted by the cell they fall in (a counting sort over the 16 keys,
// stable, so the result does not depend on timing); a pass then takes 128 consecutive
// samples of that order, i.e. a few neighbouring cells, and walks only the candidates
// whose mask meets those cells.  A sample's origin follows from its index alone (closed
// form of the acne chain), so the order of the samples is free.
#ifndef SORT_MAX
#define SORT_MAX 512        // samples sorted at a time
#endif
#ifndef SORT_MIN_CAND
#define SORT_MIN_CAND 10    // with fewer candidates the sort costs more than it saves
#endif

// cell (cx, cy) -> position along the Hilbert curve, one nibble per cell, cell = cy * 4 + cx
#define HILBERT_OF_CELL 0xA965B874CD23FE10ull
// cells of column cx / row cy as bits along the curve
#define HILBERT_X0 0x0039u
#define HILBERT_X1 0x00C6u
#define HILBERT_X2 0x6300u
#define HILBERT_X3 0x9C00u
#define HILBERT_Y0 0xC003u
#define HILBERT_Y1 0x300Cu
#define HILBERT_Y2 0x0990u
#define HILBERT_Y3 0x0660u

struct SortLds {
    uint32_t sample[SORT_MAX];            // by sample: ux | uy << 9 | key << 18 | rank << 22
    uint16_t order[SORT_MAX];             // sorted: sample | key << 12
    uint16_t count[(SORT_MAX / 64) * 16]; // samples per (64-sample block, key), then offsets
    uint16_t key_base[16];
    uint16_t cand[CAND_MAX];  // screened primitives of the leaf (offsets inside the leaf)
    uint16_t cells[CAND_MAX]; // their cell masks
    uint16_t pass[CAND_MAX];  // those of them that meet the cells of the current pass
};

// Wave-uniform description of a record's shadow rays: index 0 is the first start
// point, 1 the last.
struct RayFan {
    float ox[2], oy[2], soz[2]; // start point; z times sigma
    float h[2];                 // |z_light - o.z|
    float gx0[2], gy0[2];       // 250 - o.x, 250 - o.y
    float thr[2];               // margin times (an upper bound of) the norm of a plane's normal
    float per_thr[2];           // margin / thr: turns a plane function's value into a distance (at least)
    float sigma;                // +1 light above both start points, -1 below, 0 neither
    float z_lo, z_hi;           // all rays stay within z_lo <= z (or z <= z_hi), before the margin
    float margin;
};

__device__ __forceinline__ RayFan make_fan(f3 o_first, f3 o_last, float zl, float margin) {
    RayFan f;
    const float o_zmin = fminf(o_first.z, o_last.z), o_zmax = fmaxf(o_first.z, o_last.z);
    f.sigma = (zl > o_zmax) ? 1.f : (zl < o_zmin) ? -1.f : 0.f;
    f.z_lo = (zl > o_zmax) ? o_zmin : -3.0e38f;
    f.z_hi = (zl < o_zmin) ? o_zmax : 3.0e38f;
    f.margin = margin;
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const f3 o = w ? o_last : o_first;
        f.ox[w] = o.x;
        f.oy[w] = o.y;
        f.soz[w] = f.sigma * o.z;
        f.h[w] = f.sigma * (zl - o.z);
        f.gx0[w] = 250.f - o.x;
        f.gy0[w] = 250.f - o.y;
        const float far = fmaxf(fmaxf(fabsf(250.f - o.x), fabsf(750.f - o.x)),
                                fmaxf(fabsf(250.f - o.y), fabsf(750.f - o.y)));
        const float len = 1.0001f * __builtin_amdgcn_sqrtf(f.h[w] * f.h[w] + far * far);
        f.thr[w] = margin * len;
        f.per_thr[w] = (len > 0.f && margin < 1.0e38f) ? 0.9999f / len : 0.f;
    }
    return f;
}

// Distance (a lower bound of it) by which triangle (v0, v0 + e1, v0 + e2) lies outside the
// pyramid of all the record's shadow rays: below line 0 or above line 4 in x or in y, or behind
// the rays in z; the triangle is outside_fan when it exceeds the margin.  Finite coordinates only.
__device__ __forceinline__ float fan_separation(const RayFan &f, f3 v0, f3 e1, f3 e2) {
    const f3 v[3] = {v0, f3_add(v0, e1), f3_add(v0, e2)};
    float sep = fmaxf(f.z_lo - fmaxf(fmaxf(v[0].z, v[1].z), v[2].z), fminf(fminf(v[0].z, v[1].z), v[2].z) - f.z_hi);
    if (f.sigma == 0.f) return sep;
    float xb = 3.0e38f, xa = 3.0e38f, yb = 3.0e38f, ya = 3.0e38f; // smallest distance beyond each face
#pragma unroll
    for (int w = 0; w < 2; w++) {
        float gx_max = -3.0e38f, gy_max = -3.0e38f, gx_min = 3.0e38f, gy_min = 3.0e38f;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float dz = f.sigma * v[j].z - f.soz[w];
            const float t = 500.f * dz;
            const float gx = f.h[w] * (v[j].x - f.ox[w]) - f.gx0[w] * dz;
            const float gy = f.h[w] * (v[j].y - f.oy[w]) - f.gy0[w] * dz;
            gx_max = fmaxf(gx_max, gx);
            gy_max = fmaxf(gy_max, gy);
            gx_min = fminf(gx_min, gx - t); // line 4: X + 500
            gy_min = fminf(gy_min, gy - t);
        }
        xb = fminf(xb, -gx_max * f.per_thr[w]);
        yb = fminf(yb, -gy_max * f.per_thr[w]);
        xa = fminf(xa, gx_min * f.per_thr[w]);
        ya = fminf(ya, gy_min * f.per_thr[w]);
    }
    return fmaxf(sep, fmaxf(fmaxf(xb, xa), fmaxf(yb, ya)));
}

// Cells of the light whose rays may hit a triangle that is not outside the fan.
__device__ __forceinline__ uint32_t cell_mask(const RayFan &f, f3 v0, f3 e1, f3 e2) {
    if (f.sigma == 0.f) return 0xFFFFu;
    const f3 v[3] = {v0, f3_add(v0, e1), f3_add(v0, e2)};
    // xb[i]: the triangle is on the low side of line i (i = 0..3) for every start point, i.e. the
    // largest of its vertices' plane functions is below -thr from both ends of the segment;
    // xa[i]: on the high side of line i + 1 (the smallest above thr)
    bool xb[4] = {true, true, true, true}, xa[4] = {true, true, true, true};
    bool yb[4] = {true, true, true, true}, ya[4] = {true, true, true, true};
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const float lo = -f.thr[w], hi = f.thr[w];
        float gx[3], gy[3], t[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float dz = f.sigma * v[j].z - f.soz[w];
            t[j] = 125.f * dz;
            gx[j] = f.h[w] * (v[j].x - f.ox[w]) - f.gx0[w] * dz;
            gy[j] = f.h[w] * (v[j].y - f.oy[w]) - f.gy0[w] * dz;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            xb[i] = xb[i] && fmaxf(fmaxf(gx[0], gx[1]), gx[2]) < lo;
            yb[i] = yb[i] && fmaxf(fmaxf(gy[0], gy[1]), gy[2]) < lo;
#pragma unroll
            for (int j = 0; j < 3; j++) { // next line: X + 125
                gx[j] = gx[j] - t[j];
                gy[j] = gy[j] - t[j];
            }
            xa[i] = xa[i] && fminf(fminf(gx[0], gx[1]), gx[2]) > hi;
            ya[i] = ya[i] && fminf(fminf(gy[0], gy[1]), gy[2]) > hi;
        }
    }
    // column c is out if the triangle is below one of the lines 0..c or above one of c+1..4
    const bool x0 = !(xb[0] || xa[0] || xa[1] || xa[2] || xa[3]);
    const bool x1 = !(xb[0] || xb[1] || xa[1] || xa[2] || xa[3]);
    const bool x2 = !(xb[0] || xb[1] || xb[2] || xa[2] || xa[3]);
    const bool x3 = !(xb[0] || xb[1] || xb[2] || xb[3] || xa[3]);
    const bool y0 = !(yb[0] || ya[0] || ya[1] || ya[2] || ya[3]);
    const bool y1 = !(yb[0] || yb[1] || ya[1] || ya[2] || ya[3]);
    const bool y2 = !(yb[0] || yb[1] || yb[2] || ya[2] || ya[3]);
    const bool y3 = !(yb[0] || yb[1] || yb[2] || yb[3] || ya[3]);
    const uint32_t mx = (x0 ? HILBERT_X0 : 0u) | (x1 ? HILBERT_X1 : 0u) | (x2 ? HILBERT_X2 : 0u) |
                        (x3 ? HILBERT_X3 : 0u);
    const uint32_t my = (y0 ? HILBERT_Y0 : 0u) | (y1 ? HILBERT_Y1 : 0u) | (y2 ? HILBERT_Y2 : 0u) |
                        (y3 ? HILBERT_Y3 : 0u);
    return mx & my;
}

// fan_separation and cell_mask in one evaluation of the plane functions (the outer faces of the
// pyramid are the light's lines 0 and 4): for records that sort their samples.
__device__ __forceinline__ void fan_screen(const RayFan &f, f3 v0, f3 e1, f3 e2, float &sep_out, uint32_t &cells_out) {
    const f3 v[3] = {v0, f3_add(v0, e1), f3_add(v0, e2)};
    float sep = fmaxf(f.z_lo - fmaxf(fmaxf(v[0].z, v[1].z), v[2].z), fminf(fminf(v[0].z, v[1].z), v[2].z) - f.z_hi);
    sep_out = sep;
    cells_out = 0xFFFFu;
    if (f.sigma == 0.f) return;
    bool xb[4] = {true, true, true, true}, xa[4] = {true, true, true, true};
    bool yb[4] = {true, true, true, true}, ya[4] = {true, true, true, true};
    float sxb = 3.0e38f, sxa = 3.0e38f, syb = 3.0e38f, sya = 3.0e38f; // smallest distance beyond each outer face
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const float lo = -f.thr[w], hi = f.thr[w];
        float gx[3], gy[3], t[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float dz = f.sigma * v[j].z - f.soz[w];
            t[j] = 125.f * dz;
            gx[j] = f.h[w] * (v[j].x - f.ox[w]) - f.gx0[w] * dz;
            gy[j] = f.h[w] * (v[j].y - f.oy[w]) - f.gy0[w] * dz;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float mx = fmaxf(fmaxf(gx[0], gx[1]), gx[2]), my = fmaxf(fmaxf(gy[0], gy[1]), gy[2]);
            xb[i] = xb[i] && mx < lo;
            yb[i] = yb[i] && my < lo;
            if (i == 0) {
                sxb = fminf(sxb, -mx * f.per_thr[w]);
                syb = fminf(syb, -my * f.per_thr[w]);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) { // next line: X + 125
                gx[j] = gx[j] - t[j];
                gy[j] = gy[j] - t[j];
            }
            const float nx = fminf(fminf(gx[0], gx[1]), gx[2]), ny = fminf(fminf(gy[0], gy[1]), gy[2]);
            xa[i] = xa[i] && nx > hi;
            ya[i] = ya[i] && ny > hi;
            if (i == 3) {
                sxa = fminf(sxa, nx * f.per_thr[w]);
                sya = fminf(sya, ny * f.per_thr[w]);
            }
        }
    }
    sep_out = fmaxf(sep, fmaxf(fmaxf(sxb, sxa), fmaxf(syb, sya)));
    const bool x0 = !(xb[0] || xa[0] || xa[1] || xa[2] || xa[3]);
    const bool x1 = !(xb[0] || xb[1] || xa[1] || xa[2] || xa[3]);
    const bool x2 = !(xb[0] || xb[1] || xb[2] || xa[2] || xa[3]);
    const bool x3 = !(xb[0] || xb[1] || xb[2] || xb[3] || xa[3]);
    const bool y0 = !(yb[0] || ya[0] || ya[1] || ya[2] || ya[3]);
    const bool y1 = !(yb[0] || yb[1] || ya[1] || ya[2] || ya[3]);
    const bool y2 = !(yb[0] || yb[1] || yb[2] || ya[2] || ya[3]);
    const bool y3 = !(yb[0] || yb[1] || yb[2] || yb[3] || ya[3]);
    const uint32_t mx = (x0 ? HILBERT_X0 : 0u) | (x1 ? HILBERT_X1 : 0u) | (x2 ? HILBERT_X2 : 0u) |
                        (x3 ? HILBERT_X3 : 0u);
    const uint32_t my = (y0 ? HILBERT_Y0 : 0u) | (y1 ? HILBERT_Y1 : 0u) | (y2 ? HILBERT_Y2 : 0u) |
                        (y3 ? HILBERT_Y3 : 0u);
    cells_out = mx & my;
}

// Screens the n <= CAND_MAX primitives of the leaf: offsets inside the leaf of those that
// are not outside_fan, in order, to L.cand.  Returns how many.  G (may be null: nothing is
// dropped then): the record's grazing guard, see shade_record.h; a triangle it keeps for every
// sample carries FULL_MASK in its list entry.
#define CAND_FULL_MASK 0x8000u
__device__ __forceinline__ uint32_t build_fan_candidates(SortLds &L, const RayFan &fan,
                                                          const DPrim *__restrict__ leaf_prims, uint32_t n,
                                                          uint32_t lane, const GrazeCtx *G,
                                                          uint32_t *__restrict__ graze_list, uint32_t graze_cap,
                                                          uint32_t &n_graze) {
    uint32_t total = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t t = base + lane;
        bool keep = false;
        uint32_t code = 0, first = 0, cnt = 0;
        if (t < n) {
            const DPrim *p = leaf_prims + t;
            keep = true;
            if ((p->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) {
                const float sep = fan_separation(fan, p->a, p->b, p->c);
                keep = !(sep > fan.margin);
                if (!keep && G) {
                    code = graze_check(*G, p, sep, first, cnt);
                    keep = code == 2u;
                }
            }
        }
        if (G) graze_push(graze_list, graze_cap, n_graze, code, t, first, cnt);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            L.cand[pos] = (uint16_t)(t | (code == 2u ? CAND_FULL_MASK : 0u));
        }
        total += (uint32_t)__popcll(mask);
    }
    return total;
}

// Cell masks of the n_cand candidates, to L.cells (and the marks on the list entries cleared).
// A triangle that is left out of some cells goes through the grazing guard like one that is
// dropped altogether.
__device__ __forceinline__ void build_cell_masks(SortLds &L, const RayFan &fan,
                                                 const DPrim *__restrict__ leaf_prims, uint32_t n_cand,
                                                 uint32_t lane, const GrazeCtx *G, uint32_t *__restrict__ graze_list,
                                                 uint32_t graze_cap, uint32_t &n_graze) {
    for (uint32_t base = 0; base < n_cand; base += 64) {
        const uint32_t c = base + lane;
        uint32_t code = 0, first = 0, cnt = 0, t = 0;
        if (c < n_cand) {
            const uint32_t entry = L.cand[c];
            t = entry & (CAND_FULL_MASK - 1u);
            const DPrim *p = leaf_prims + t;
            uint32_t cells = 0xFFFFu;
            if ((p->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE && !(entry & CAND_FULL_MASK)) {
                cells = cell_mask(fan, p->a, p->b, p->c);
                if (cells != 0xFFFFu && G) {
                    // (left out of a cell by the margin at least)
                    code = graze_check(*G, p, fan.margin, first, cnt);
                    if (code == 2u) cells = 0xFFFFu;
                }
            }
            L.cells[c] = (uint16_t)cells;
            L.cand[c] = (uint16_t)t;
        }
        if (G) graze_push(graze_list, graze_cap, n_graze, code, t, first, cnt);
    }
}

// Copies to L.pass the candidates whose cells meet `cells`; returns how many.
__device__ __forceinline__ uint32_t filter_candidates(SortLds &L, uint32_t n_cand, uint32_t cells,
                                                       uint32_t lane) {
    uint32_t total = 0;
    for (uint32_t base = 0; base < n_cand; base += 64) {
        const uint32_t c = base + lane;
        const bool keep = c < n_cand && ((uint32_t)L.cells[c] & cells) != 0;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            L.pass[pos] = L.cand[c];
        }
        total += (uint32_t)__popcll(mask);
    }
    return total;
}

// Sorts the light samples k0 .. k0 + n - 1 (n <= SORT_MAX) of a record by the Hilbert
// key of their light cell: L.sample[i] keeps the two integers of sample k0 + i,
// L.order lists the samples by key, samples of one key in index order.
__device__ __forceinline__ void sort_samples(SortLds &L, uint32_t lane, uint32_t seed, uint32_t pixel,
                                             uint32_t leaf, uint32_t k0, uint32_t n) {
    const uint32_t n_blocks = (n + 63) >> 6;
    for (uint32_t e = lane; e < n_blocks * 16; e += 64) L.count[e] = 0;
    wave_lds_sync();
    for (uint32_t blk = 0; blk < n_blocks; blk++) {
        const uint32_t i = blk * 64 + lane;
        const bool valid = i < n;
        uint32_t ux, uy;
        light_sample_u(seed, pixel, leaf, k0 + i, ux, uy);
        const uint32_t cx = min((ux * 525u) >> 16, 3u), cy = min((uy * 525u) >> 16, 3u); // u / 125
        const uint32_t key = (uint32_t)(HILBERT_OF_CELL >> ((cy * 4 + cx) * 4)) & 15u;
        // lanes with my key: intersect, bit by bit, the ballots of the lanes that agree with me
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const bool bit = (key >> b) & 1u;
            const unsigned long long set = __ballot(bit);
            same &= bit ? set : ~set;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
        if (valid) {
            L.sample[i] = ux | (uy << 9) | (key << 18) | (rank << 22);
            if (rank == 0) L.count[blk * 16 + key] = (uint16_t)__popcll(same);
        }
    }
    wave_lds_sync();
    // lane key < 16: turn the counts of its key into offsets inside the key's run
    uint32_t total = 0;
    if (lane < 16) {
        for (uint32_t blk = 0; blk < n_blocks; blk++) {
            const uint32_t c = L.count[blk * 16 + lane];
            L.count[blk * 16 + lane] = (uint16_t)total;
            total += c;
        }
    }
    uint32_t incl = total;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += up;
    }
    if (lane < 16) L.key_base[lane] = (uint16_t)(incl - total);
    wave_lds_sync();
    for (uint32_t blk = 0; blk < n_blocks; blk++) {
        const uint32_t i = blk * 64 + lane;
        if (i < n) {
            const uint32_t w = L.sample[i];
            const uint32_t key = (w >> 18) & 15u, rank = w >> 22;
            const uint32_t pos = (uint32_t)L.key_base[key] + (uint32_t)L.count[blk * 16 + key] + rank;
            L.order[pos] = (uint16_t)(i | (key << 12));
        }
    }
    wave_lds_sync();
}

// Screens the leaf (count <= CAND_MAX primitives) for a level whose start points have the closed
// form, grazing guard included: candidates to L.cand, and when there are SORT_MIN_CAND or more
// (use_sort) their cell masks to L.cells.  Returns the number of candidates.
__device__ __forceinline__ uint32_t screen_sorted_level(const DFrame &fr, SortLds &L,
                                                         const DPrim *__restrict__ leaf_prims, uint32_t count,
                                                         uint32_t lane, f3 o_first, f3 o_last, float margin, f3 p0,
                                                         f3 dlt, uint32_t pixel, uint32_t leaf, uint32_t k_off,
                                                         bool &use_sort, unsigned long long *stats) {
    RayFan fan = make_fan(o_first, o_last, fr.light_pos.z, margin);
#ifdef NO_GRAZE
    const bool screened = false; // (measurement build only)
#else
    const bool screened = margin < 1.0e38f;
#endif
    const GrazeCtx G = graze_ctx(true, o_first, o_last, fr.n_lights, fr.cull_margin, margin);
    uint32_t *graze_list = L.sample; // (free until the samples are sorted)
    const uint32_t graze_cap = SORT_MAX;
    uint32_t n_graze = 0;
    uint32_t n_cand = build_fan_candidates(L, fan, leaf_prims, count, lane, screened ? &G : nullptr, graze_list, graze_cap,
                                           n_graze);
    use_sort = n_cand >= SORT_MIN_CAND;
    wave_lds_sync();
    // (without the sort every candidate is tested in every pass: the marks just go)
    build_cell_masks(L, fan, leaf_prims, use_sort ? n_cand : 0u, lane, screened ? &G : nullptr, graze_list, graze_cap,
                     n_graze);
    if (!use_sort)
        for (uint32_t c = lane; c < n_cand; c += 64) L.cand[c] &= (uint16_t)(CAND_FULL_MASK - 1u);
    wave_lds_sync();
#ifdef STEP_STATS
    if (stats) {
        stats[10] += n_graze;
        stats[11] += n_graze != 0;
        stats[13] += count;
    }
#endif
    if (n_graze && graze_resolve(fr, leaf_prims, graze_list, graze_cap, n_graze, true, p0, dlt, dlt, pixel, leaf, k_off, lane)) {
#ifdef STEP_STATS
        if (stats) stats[12]++;
#endif
        // a pair the screens would drop is not cleared: this record without screens
        fan = make_fan(o_first, o_last, fr.light_pos.z, 3.0e38f);
        wave_lds_sync();
        n_cand = build_fan_candidates(L, fan, leaf_prims, count, lane, nullptr, graze_list, graze_cap, n_graze);
        use_sort = n_cand >= SORT_MIN_CAND;
        wave_lds_sync();
        build_cell_masks(L, fan, leaf_prims, n_cand, lane, nullptr, graze_list, graze_cap, n_graze);
        wave_lds_sync();
    }
    (void)stats;
    return n_cand;
}

// One level of a hit record with sorted light samples (see above): the hit at p0 with
// normal nrm and material mat, seen along -view, in the leaf leaf_prims[0 .. count); its
// samples have the sampler's indices k_off .. k_off + n_lights - 1.  Needs n_lights >= 1,
// a leaf of at most CAND_MAX primitives and the closed form of all three origin components.
// PRE: the candidates come from the list pre[0 .. pre_n & ~CAND_N_SORT) that screen_kernel wrote
// (cell masks included when pre_n has CAND_N_SORT); else the level screens the leaf itself.
// Returns the sum of the samples' contributions (raytracing.h:36-50), the same value in
// every lane, and the number of lit samples.
template <bool IEEE, bool PRE>
__device__ __forceinline__ float4 shade_level_sorted(const DFrame &fr, const DPrim *__restrict__ leaf_prims,
                                                     uint32_t count, uint32_t pixel, uint32_t leaf, const DMat &mat,
                                                     f3 nrm, f3 p0, f3 view, uint32_t k_off, SortLds &L,
                                                     uint32_t lane, ShadeTotals &tot, uint32_t &lit_out,
                                                     const uint32_t *__restrict__ pre, uint32_t pre_n) {
        const float nl = (float)fr.n_lights;
        unsigned long long tests_total = 0;
#ifdef PHASE_TIMES
        unsigned long long phase_t = __builtin_readcyclecounter();
        const unsigned long long record_t = phase_t;
#endif
        const f3 step = f3_scale(nrm, fr.acne); // raytracing.h:40
        f3 dlt = f3_make(0.f, 0.f, 0.f);        // origin of sample k = p0 + (k + 1) * dlt
        acne_chain_delta3(lane, p0, step, fr.n_lights, dlt);

        uint32_t n_cand;
        bool use_sort;
        if constexpr (PRE) { // the list screen_kernel left for this record (primary level, build without bounces)
            n_cand = pre_n & ~CAND_N_SORT;
            use_sort = (pre_n & CAND_N_SORT) != 0;
            for (uint32_t c = lane; c < n_cand; c += 64) {
                const uint32_t w = pre[c];
                L.cand[c] = (uint16_t)w;
                L.cells[c] = (uint16_t)(w >> 16);
            }
            PHASE_MARK(1);
        } else {
            // the start points are exactly p0 + k * dlt, k = 1 .. n_lights (closed form)
            f3 o_first, o_last;
            const float margin = screen_segment(true, p0, step, dlt, nl, fr.cull_margin, o_first, o_last);
            PHASE_MARK(0);
#if defined(STEP_STATS)
            unsigned long long *stats = tot.st;
#else
            unsigned long long *stats = nullptr;
#endif
            n_cand = screen_sorted_level(fr, L, leaf_prims, count, lane, o_first, o_last, margin, p0, dlt, pixel, leaf, k_off,
                                         use_sort, stats);
            PHASE_MARK(1);
        }
#ifdef STEP_STATS
        tot.st[8] += use_sort;
#endif
        wave_lds_sync();
        PHASE_MARK(2);

        f3 acc = f3_make(0.f, 0.f, 0.f);
        uint32_t lit = 0;
        for (uint32_t chunk = 0; chunk < fr.n_lights; chunk += SORT_MAX) {
            const uint32_t n_chunk = min(fr.n_lights - chunk, (uint32_t)SORT_MAX);
            if (use_sort) sort_samples(L, lane, fr.seed, pixel, leaf, k_off + chunk, n_chunk);
            PHASE_MARK(3);
            // a pass: entries base .. base + 127 of the sorted order, two per lane
            for (uint32_t base = 0; base < n_chunk; base += PASS_SAMPLES) {
                const uint32_t last = n_chunk - 1;
                const uint32_t e1 = base + lane, e2 = e1 + 64;
                const bool live1 = e1 <= last, live2 = e2 <= last;
                uint32_t s1 = min(e1, last), s2 = min(e2, last), w1, w2;
                const uint16_t *list = L.cand;
                uint32_t n_list = n_cand;
                if (use_sort) {
                    s1 = L.order[s1];
                    s2 = L.order[s2];
                    w1 = L.sample[s1 & 0xFFFu];
                    w2 = L.sample[s2 & 0xFFFu];
                    // the keys of the pass are a range: first entry .. last entry
                    const uint32_t key_lo = __builtin_amdgcn_readfirstlane((uint32_t)L.order[base]) >> 12;
                    const uint32_t key_hi =
                        __builtin_amdgcn_readfirstlane((uint32_t)L.order[min(base + PASS_SAMPLES - 1, last)]) >> 12;
                    const uint32_t pass_cells = ((2u << key_hi) - 1u) & ~((1u << key_lo) - 1u);
                    n_list = filter_candidates(L, n_cand, pass_cells, lane);
                    list = L.pass;
                    wave_lds_sync();
                } else { // few candidates: samples in index order, every candidate in every pass
                    uint32_t ux, uy;
                    light_sample_u(fr.seed, pixel, leaf, k_off + chunk + s1, ux, uy);
                    w1 = ux | (uy << 9);
                    light_sample_u(fr.seed, pixel, leaf, k_off + chunk + s2, ux, uy);
                    w2 = ux | (uy << 9);
                }

                const float f1 = (float)(chunk + (s1 & 0xFFFu) + 1u), f2 = (float)(chunk + (s2 & 0xFFFu) + 1u);
                const f3 p1 = f3_make(p0.x + f1 * dlt.x, p0.y + f1 * dlt.y, p0.z + f1 * dlt.z);
                const f3 p2 = f3_make(p0.x + f2 * dlt.x, p0.y + f2 * dlt.y, p0.z + f2 * dlt.z);
                const f3 l1 = f3_make(250.f + (float)(w1 & 511u), 250.f + (float)((w1 >> 9) & 511u),
                                      fr.light_pos.z); // raytracing.h:38-39
                const f3 l2 = f3_make(250.f + (float)(w2 & 511u), 250.f + (float)((w2 >> 9) & 511u),
                                      fr.light_pos.z);
                ray2 rr;
                rr.ox = (v2f){p1.x, p2.x};
                rr.oy = (v2f){p1.y, p2.y};
                rr.oz = (v2f){p1.z, p2.z};
                rr.dx = (v2f){l1.x, l2.x} - rr.ox; // raytracing.h:43: normalize(light - position)
                rr.dy = (v2f){l1.y, l2.y} - rr.oy;
                rr.dz = (v2f){l1.z, l2.z} - rr.oz;
                normalize2(rr.dx, rr.dy, rr.dz);
                const f3 sd1 = f3_make(rr.dx.x, rr.dy.x, rr.dz.x), sd2 = f3_make(rr.dx.y, rr.dy.y, rr.dz.y);
                // nearest hit of each shadow ray inside the leaf (raytracing.h:45)
                // (no hit yet = +infinity: a hit at t = +inf, were there one, puts the point
                // out of the scene, which lights the sample just as no hit does)
                float tmin1 = __builtin_inff(), tmin2 = __builtin_inff();
                PHASE_MARK(4);
                if (n_list) {
                    tests_total += (unsigned long long)n_list * min(n_chunk - base, (uint32_t)PASS_SAMPLES);
                    uint32_t ci = __builtin_amdgcn_readfirstlane((uint32_t)list[0]);
                    DPrim cur = leaf_prims[ci];
                    for (uint32_t c = 0; c < n_list; c++) {
                        const uint32_t cn = __builtin_amdgcn_readfirstlane(
                            (uint32_t)list[(c + 1 < n_list) ? c + 1 : c]);
                        const DPrim nxt = leaf_prims[cn];
                        // the nearest hit so far (raytracing.h:45 takes the smallest distance)
                        const auto nearest = [&](bool hit1, bool hit2, v2f t) {
                            if (hit1 && t.x < tmin1) tmin1 = t.x;
                            if (hit2 && t.y < tmin2) tmin2 = t.y;
                        };
#ifdef STEP_STATS
                        step_stats(tot, rr, cur, live1, live2);
#endif
                        if ((cur.type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) {
                            tri_hit2(rr, cur.a, cur.b, cur.c, fr.wide_dets != 0, live1, live2, nearest);
                        } else {
                            float ta = 0.f, tb = 0.f;
                            const bool hit1 = live1 && sph_hit(p1, sd1, cur.a, cur.b.y, ta);
                            const bool hit2 = live2 && sph_hit(p2, sd2, cur.a, cur.b.y, tb);
                            nearest(hit1, hit2, (v2f){ta, tb});
                        }
                        cur = nxt;
                    }
                }
                PHASE_MARK(5);
                // lit unless something is hit inside [0, 1000]^3 (utils.h:12-17)
                const bool lit1 = live1 && (!(tmin1 < __builtin_inff()) || out_of_scene(f3_add(p1, f3_scale(sd1, tmin1))));
                const bool lit2 = live2 && (!(tmin2 < __builtin_inff()) || out_of_scene(f3_add(p2, f3_scale(sd2, tmin2))));
                if (lit1 || lit2) {
                    f3 c1, c2;
                    shade2<IEEE>(nrm, view, rr.dx, rr.dy, rr.dz, lit1, lit2, fr.light_color, mat, nl, c1, c2);
                    if (lit1) {
                        acc = f3_add(acc, c1); // raytracing.h:48
                        lit++;
                    }
                    if (lit2) {
                        acc = f3_add(acc, c2);
                        lit++;
                    }
                }
                wave_lds_sync();
                PHASE_MARK(6);
            }
        }
        // sum over the lanes (fixed butterfly order, so a frame is reproducible)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acc.x += __shfl_xor(acc.x, off);
            acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off);
            lit += __shfl_xor(lit, off);
        }
        tot.tests += tests_total;
        PHASE_MARK(7);
#ifdef PHASE_TIMES
        if (phase_t - record_t > tot.st[8]) { // the longest record of this wave, and its candidates
            tot.st[8] = phase_t - record_t;
            tot.st[9] = n_cand;
        }
#endif
        lit_out = lit;
        return make_float4(acc.x, acc.y, acc.z, 0.f);
}

// One hit record with sorted light samples: its primary level, to colors[ridx].
template <bool IEEE>
__device__ __forceinline__ void shade_record_sorted(const DFrame &fr, const DPrim *__restrict__ prims,
                                                    const DMat *__restrict__ mats,
                                                    const DRec *__restrict__ recs, float4 *__restrict__ colors,
                                                    const CandLists &lists, SortLds &L, uint32_t lane, uint32_t ridx,
                                                    ShadeTotals &tot) {
        const DRec rc = recs[ridx];
        const uint32_t pixel = rc.pixel;
        const DMat mat = mats[rc.mat];
        // (raytracing.h:47: the hit is seen from minus the primary ray's direction; screen_kernel has
        // left that in the place of the record's colour)
        const float4 vw = colors[ridx];
        uint32_t lit = 0;
        const float4 sum = shade_level_sorted<IEEE, true>(fr, prims + rc.first, rc.count, pixel, rc.leaf, mat,
                                                    f3_make(rc.nx, rc.ny, rc.nz), f3_make(rc.px, rc.py, rc.pz),
                                                    f3_make(wave_uniform(vw.x), wave_uniform(vw.y), wave_uniform(vw.z)), 0u, L, lane,
                                                    tot, lit,
                                                    lists.pool + lists.cand_off[ridx], lists.cand_n[ridx]);
        if (lane == 0) colors[ridx] = sum;
        tot.lit += lit;
}
