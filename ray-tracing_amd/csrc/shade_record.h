// Shading of one hit record by one wave: shadow rays against the record's leaf + BSDF,
// mirror bounces.
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

// ---- shadow + shade ---------------------------------------------------------------
// One wave per hit record.  A pass covers 128 light samples, two per lane, carried
// through the triangle test as packed pairs.  All lanes of a wave walk the same leaf,
// so the primitive data is wave-uniform (scalar loads) and only the rays differ per
// lane.  The shadow-ray origin of sample k is the hit position after k + 1 float
// additions of normal * acne (raytracing.h:40).  Where acne_chain_delta proves that
// chain equal, bit for bit, to p0 + (k + 1) * delta, every lane computes its origins
// directly; otherwise lanes 0..2 (x, y, z) run the additions 128 steps at a time into
// LDS and every lane then picks up its own.
//
// Candidate list.  The reference tests every primitive of the leaf for every
// shadow ray (intersect.h:99-117).  All shadow rays of one record start on the
// segment between the first and the last start point and aim at the light rectangle
// [250, 750]^2 x {z_light} (raytracing.h:38-39), so they stay inside the pyramid
// spanned by the start points and that rectangle (continued beyond it: there is
// no far limit).  The faces are the planes through a start point and one edge of
// the rectangle; g(o, v) = det[v - o, La - o, Lb - o] and the face's normal are both
// affine in o, so "all three vertices outside a face by the margin, seen from the
// first and from the last start point" implies "the whole triangle outside that face
// by the margin for every start point on the segment".  Once per record, lanes run
// that test over the leaf's triangles (one each) and compact the survivors, in order,
// into LDS; the sample loop then walks only those.  Leaves too large for the list are
// screened in segments, once per pass.
//
// What the margin stands for.  In exact arithmetic a triangle outside the pyramid is missed
// by every ray of the record, whatever the margin.  The margin separates that geometry from
// the float arithmetic it stands for:
//  * the start points.  Where acne_chain_delta gives the closed form they are exactly
//    p0 + k * delta, k = 1 .. n_lights, and the screens use exactly that segment.  Otherwise
//    they are the rounded running sum (raytracing.h:40), which leaves the line p0 + k * step
//    by at most half an ulp of the running value per step and component; that bound is
//    added to the record's margin, and a record whose bound exceeds three margins is not
//    screened at all (screen_segment);
//  * the screens' own plane arithmetic and the rounding of the normalised ray direction:
//    each at most 2^-20 times the largest coordinate in play; screen_margin() raises the
//    margin with the scene's size so that this stays below a quarter of it;
//  * the float test itself (intersect.h:28-67), which can accept a ray that misses the triangle
//    in exact arithmetic when ray and start point lie almost in the triangle's plane: the
//    grazing guard below bounds when, and hands those (triangle, sample) pairs to the float
//    test itself, so that no pair is dropped that the reference's test would accept.
#define CAND_MAX 512
// Margin of the candidate screens, in world units (raised for scenes with very large
// coordinates, see screen_margin).
#ifndef CULL_MARGIN
#define CULL_MARGIN 0.1f
#endif

#define PASS_SAMPLES 128 // light samples per pass: two per lane

#define MAX_BOUNCES 8 // deepest mirror recursion (rtx_params::max_depth) the kernel folds

// Register budget of shade_kernel: 5 waves per SIMD (96 VGPRs).  Measured on buddha
// 3840x2160 / 500 samples: 4 waves (the compiler's own choice, 110 VGPRs) 5.12 ms,
// 5 waves 4.87 ms, 6 waves 4.86 ms with three times the scratch traffic.  The build
// with mirror bounces keeps the compiler's 3 waves (167 VGPRs) rather than spill (it sees
// only the few records whose bounce hits something; 4..6 waves: within 4 %).
#ifndef SHADE_WAVES_PER_SIMD
#define SHADE_WAVES_PER_SIMD 5
#endif
#ifndef SHADE_WAVES_PER_SIMD_BOUNCE
#define SHADE_WAVES_PER_SIMD_BOUNCE 3
#endif
// Frames whose records are shaded with sorted samples (SORT_MIN_LIGHTS samples or more): 6 waves
// (80 VGPRs).  Measured with 5 / 4 / 6 waves: buddha 4K/500 3.14 / 3.28 / 3.07 ms, the
// multi-mesh frame 36.5 / 38.4 / 35.7, dino 2.37 / 2.38 / 2.37, bunny (100 samples, unsorted)
// 0.625 / 0.604 / 0.656.
#ifndef SHADE_WAVES_PER_SIMD_SORT
#define SHADE_WAVES_PER_SIMD_SORT 6
#endif

// A wave takes this many records of a queue per atomic.  Returning atomics on one
// address complete at only a few tens of millions per second, which is the same order
// as the record rate of a frame with few light samples.
#ifndef QUEUE_BATCH
#define QUEUE_BATCH 2u
#endif

// Work classes (shade_kernel takes them weight group by weight group, plain before
// sorted).  Records with a leaf of
// SORT_MIN_LEAF..CAND_MAX primitives, more than one pass of light samples and the closed
// form of all three origin components are shaded with sorted samples (below).
#ifndef N_WEIGHTS
#define N_WEIGHTS 2 // weight groups per kind of record (sorted / plain), by leaf size
#endif
#define N_CLASSES (2 * N_WEIGHTS)
static_assert(N_CLASSES <= WORK_LISTS, "DCounters holds one list per class");
#define CLASS_SORTED 0u          // classes 0 .. N_WEIGHTS - 1: sorted samples, heaviest first
#define CLASS_PLAIN N_WEIGHTS    // then the plain records, heaviest first
#define CLASS_LIGHTEST (N_CLASSES - 1)
// Weight group of a leaf of `count` primitives, 0 = heaviest.  Two groups (leaves from 96
// primitives, the rest) measured best: one group leaves long records to the end of the
// kernel, three or four change nothing that can be measured.
__device__ __forceinline__ uint32_t weight_group(uint32_t count) {
    if (N_WEIGHTS == 1) return 0u;
    if (N_WEIGHTS == 2) return (count >= 96u) ? 0u : 1u;
    return (count >= 192u) ? 0u : (count >= 96u) ? 1u : (count >= 48u) ? 2u : 3u;
}
#define SORT_MIN_LIGHTS 129   // fewer samples are a single pass: nothing to gain
#ifndef SORT_MIN_LEAF
#define SORT_MIN_LEAF 12u     // smaller leaves leave too few candidates for the sort to pay
#endif

// The work lists (record numbers) lie behind the rec_cap colours, rec_cap entries each.
__device__ __forceinline__ uint32_t *work_list(float4 *colors, uint32_t rec_cap, uint32_t cls) {
    return reinterpret_cast<uint32_t *>(colors + rec_cap) + (size_t)cls * rec_cap;
}

// (bounce build only, beside the per-wave LDS of the sampling paths)
struct BounceLds {
    float4 level_sum[MAX_BOUNCES + 2];    // per bounce level: sum of its light samples
    float4 level_weight[MAX_BOUNCES + 1]; // specular * Fresnel of the level's hit
};

struct ShadeLds {
    float chain[PASS_SAMPLES * 3];
    float4 plane[8];
    uint16_t cand[CAND_MAX];
};

// Lanes 0..7 each set up one pyramid face: edge (lane & 3) of the light rectangle
// seen from the first (lane < 4) or last start point, as unit normal (pointing
// inwards) and offset.  A degenerate face becomes all zeros, which culls nothing; so do
// all faces when the light plane passes between the two start points.
__device__ __forceinline__ void setup_planes(ShadeLds &L, uint32_t lane, f3 o_first, f3 o_last, float zl) {
    if (lane < 8) {
        const f3 o = (lane < 4) ? o_first : o_last;
        const uint32_t e = lane & 3;
        const float xa = (e == 0 || e == 3) ? 250.f : 750.f, ya = (e < 2) ? 250.f : 750.f;
        const float xb = (e < 2) ? 750.f : 250.f, yb = (e == 1 || e == 2) ? 750.f : 250.f;
        // corners in order (250,250) (750,250) (750,750) (250,750); edge e = corner e -> e + 1
        const f3 da = f3_make(xa - o.x, ya - o.y, zl - o.z);
        const f3 db = f3_make((e == 0 || e == 1) ? 750.f - o.x : 250.f - o.x,
                              (e == 1 || e == 2) ? 750.f - o.y : 250.f - o.y, zl - o.z);
        (void)xb;
        (void)yb;
        f3 m = f3_cross(da, db);
        const f3 dc = f3_make(500.f - o.x, 500.f - o.y, zl - o.z);
        if (f3_dot(m, dc) < 0.f) m = f3_scale(m, -1.f);
        const float len2 = f3_dot(m, m);
        float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
        // Both start points must lie on the same side of the light plane: in between, the
        // pyramid turns inside out and "outside for both" says nothing about the others.
        const bool same_side = (zl > o_first.z && zl > o_last.z) || (zl < o_first.z && zl < o_last.z);
        if (same_side && len2 > 1e-12f && len2 < 1e30f) {
            // (v_rsq_f32, 1 ulp: a screen's own rounding is covered by its margin)
            m = f3_scale(m, __builtin_amdgcn_rsqf(len2));
            pl = make_float4(m.x, m.y, m.z, f3_dot(m, o));
        }
        L.plane[lane] = pl;
    }
}

// Distance by which triangle (v0, v0 + e1, v0 + e2) lies outside the pyramid of the record's
// shadow rays: the largest, over the faces, of the smallest distance of a vertex beyond the
// face as seen from the first and from the last start point (zero or negative: not outside
// that face).  z_lo / z_hi: the lowest start point when the light is above all of them (every
// ray then moves up), the highest when it is below, else -/+ 3e38.  Finite coordinates only.
__device__ __forceinline__ float pyramid_separation(const ShadeLds &L, f3 v0, f3 e1, f3 e2, float z_lo, float z_hi) {
    const f3 v1 = f3_add(v0, e1), v2 = f3_add(v0, e2);
    float sep = fmaxf(z_lo - fmaxf(fmaxf(v0.z, v1.z), v2.z), fminf(fminf(v0.z, v1.z), v2.z) - z_hi);
#pragma unroll 1
    for (int e = 0; e < 4; e++) {
        float worst = -3.0e38f; // the vertex nearest to (or farthest inside) face e, from either end
#pragma unroll
        for (int w = 0; w < 2; w++) {
            const float4 pl = L.plane[e + 4 * w];
            const float d0 = pl.x * v0.x + pl.y * v0.y + pl.z * v0.z - pl.w;
            const float d1 = pl.x * v1.x + pl.y * v1.y + pl.z * v1.z - pl.w;
            const float d2 = pl.x * v2.x + pl.y * v2.y + pl.z * v2.z - pl.w;
            worst = fmaxf(worst, fmaxf(fmaxf(d0, d1), d2));
        }
        sep = fmaxf(sep, -worst);
    }
    return sep;
}

// ---- grazing guard ------------------------------------------------------------------
// A screen drops a (triangle, sample) pair when a plane through the sample's start point o
// has the ray on one side and the whole triangle at least m beyond it on the other: in
// exact arithmetic the ray misses.  The reference's float test (intersect.h:28-67) can
// still accept such a ray, and then the reference counts the sample as occluded.  What
// follows shows when it cannot, and graze_check() / graze_resolve() deal with the rest.
//
// The float test, on its inputs d, e1, e2, s' = fl(o - v0), computes rounded versions of
//   A = e1 . (d x e2),  U = s' . (d x e2),  V = d . (s' x e1),  T = e2 . (s' x e1),
// which satisfy A s' = U e1 + V e2 - T d exactly (Cramer's rule).  With eps = 2^-24,
// a = |e1|, b = |e2|, sigma = |s'| and |d| = 1 up to 3 eps, the roundings of the products,
// differences and left-to-right sums give (first order; the code's constant GRAZE_EPS has
// room for the rest)
//   |dA| <= 6.5 eps a b,  |dU| <= 6.5 eps sigma b,  |dV| <= 6.5 eps sigma a,
//   |dT| <= 6.5 eps sigma a b        (6.5 > 3 + 2 sqrt 3: a three-term dot product of a
//                                     cross product whose components carry 2 eps each).
// Acceptance (say with A~ > 0; the other sign is symmetric) needs U~ >= 0, V~ >= 0,
// U~ + V~ <= A~ (1 + 3 eps) and T~ > 0.  Let w be the unit normal of the separating plane
// and g_i = w . (v_i - o') <= -m for the three vertices.  Dotting the identity with w:
//   T (w . d) = (A - U - V) g_0 + U g_1 + V g_2 .
// With the rounded coefficients the right side is at most -m A~ + 3 eps A~ sigma; replacing
// them by the exact ones changes it by at most |dA| sigma + |dU| a + |dV| b; the left side is
// at least -(|dT| + 3 eps sigma a b) since T > -|dT| and w . d >= -3 eps.  Hence
//   A~ <= 29 eps sigma a b / (m - 3 eps sigma) =: A_max :
// only a ray within about 1e-4 of parallel to the triangle's plane can be wrongly accepted.
// For such a ray U~, V~ <= A~ (1 + 2 eps) as well, and the identity dotted with d bounds
//   |T| <= A_max (sigma + a + b) + 19.4 eps sigma a b ,
// i.e. the start point lies within |T| / |e1 x e2| of the triangle's plane: with
// shape = a b / |e1 x e2| and ell = a + b,
//   rho = eps sigma (shape (29 (sigma + ell) / m + 19.4) + 8)
// (the 8 eps sigma cover the guard's own evaluation of the distance).  A pair whose start
// point is farther than rho from the triangle's plane is rejected by the float test.
//
// The start points of a record lie on a segment, so their distances from a triangle's plane
// form an arithmetic sequence: at most a few consecutive samples can be within rho (rho is
// below a sample's step in the usual case).  graze_check() finds them for every triangle a
// screen is about to drop: none - dropping is safe; more than GRAZE_KMAX, or anything that
// cannot be bounded (a triangle without area, coordinates so large that m is within
// rounding) - the triangle
// is kept for every sample; else the (triangle, sample) pairs go to a list, and
// graze_resolve() runs the float test itself on each of them: if it accepts one, the record
// is shaded without screens.  m is half the frame's screen margin: a quarter of that margin
// is the screens' own rounding (screen_margin()), and the rest covers s' against o - v0.
#define GRAZE_EPS 6.1e-8f // 2^-24 and room for the higher-order terms
#ifndef GRAZE_KMAX
#define GRAZE_KMAX 8u
#endif
#define GRAZE_LIST 256u   // pairs per level screened inside the shading kernel (bounce levels); more: no screens

struct GrazeCtx {
    f3 o1, on;    // ends of the segment the screens were built on (screen_segment)
    float steps;  // n_lights - 1
    // sample k0 starts at o1 + (k0 + k_add) / k_den * (on - o1): the closed form's segment runs
    // from the first to the last start point, the other from the hit position to one step
    // beyond the last
    float k_add, k_den, per_k_den;
    float m4;     // screen margin / 4
    float drift;  // bound on a start point's distance from the segment (0 with the closed form)
};
__device__ __forceinline__ GrazeCtx graze_ctx(bool closed, f3 o_first, f3 o_last, uint32_t n_lights, float base_margin,
                                              float margin) {
    GrazeCtx g;
    g.o1 = o_first;
    g.on = o_last;
    g.steps = (float)(n_lights > 0 ? n_lights - 1u : 0u);
    g.k_add = closed ? 0.f : 1.f;
    g.k_den = closed ? g.steps : (float)n_lights + 1.f;
    g.per_k_den = (g.k_den > 0.f) ? 1.f / g.k_den : 0.f;
    g.m4 = 0.25f * base_margin;
    g.drift = closed ? 0.f : margin - base_margin; // what screen_segment added for the chain's rounding
    return g;
}

// For the triangle p that a screen is about to drop for some or all samples of the record,
// being outside the separating planes by `sep` (at least the record's screen margin):
// 0 = safe, 2 = keep it for every sample, 1 = safe but for samples first .. first + count - 1
// (count <= GRAZE_KMAX; sample k0 starts at the (k0 + 1)-th point of the segment).
__device__ __forceinline__ uint32_t graze_check(const GrazeCtx &g, const DPrim *__restrict__ p, float sep,
                                                uint32_t &first, uint32_t &count) {
    const f3 s1 = f3_sub(g.o1, p->a), sn = f3_sub(g.on, p->a);
    // (the hardware's square root and reciprocals, 1 ulp: the factors 1.001 and 1.05 cover them)
    const float sig = __builtin_amdgcn_sqrtf(fmaxf(f3_dot(s1, s1), f3_dot(sn, sn))) * 1.001f;
    if (!(GRAZE_EPS * 5.f * sig <= g.m4)) return 2u; // (a NaN ends here too)
    // m: the separation less the start points' distance from the segment, the screens' own
    // rounding (a quarter of the margin) and s' against o - v0 (the check above: another quarter)
    const float m = sep - g.drift - 2.f * g.m4;
    if (!(m >= 2.f * g.m4)) return 2u;
    const float rho =
        GRAZE_EPS * sig * (p->shape * (29.f * (sig + p->ell) * __builtin_amdgcn_rcpf(m) + 19.4f) + 8.f) * 1.05f + g.drift + 1.0e-30f;
    const float h1 = f3_dot(p->n, s1), hn = f3_dot(p->n, sn);
    if ((h1 > rho && hn > rho) || (h1 < -rho && hn < -rho)) return 0u;
    // distance of sample k0's start point from the plane: h1 + (k0 + k_add) * dh, k0 = 0 .. steps
    const float dh = (g.k_den > 0.f) ? (hn - h1) * g.per_k_den : 0.f;
    float lo = 0.f, hi = g.steps;
    if (dh != 0.f) {
        // (the interval is widened by 0.02 + 1e-5 of its ends: more than the reciprocal's rounding
        // and that of h1 + k dh can move them)
        const float per_dh = __builtin_amdgcn_rcpf(dh);
        const float x = (-rho - h1) * per_dh - g.k_add, y = (rho - h1) * per_dh - g.k_add;
        const float slack = 0.02f + 1.0e-5f * fmaxf(__builtin_fabsf(x), __builtin_fabsf(y));
        lo = fmaxf(__builtin_ceilf(fminf(x, y) - slack), 0.f);
        hi = fminf(__builtin_floorf(fmaxf(x, y) + slack), g.steps);
    } else if (!(__builtin_fabsf(h1) <= rho)) {
        return (h1 == h1) ? 0u : 2u;
    }
    const float n = hi - lo + 1.f;
    if (n <= 0.f) return 0u;
    if (!(n <= (float)GRAZE_KMAX)) return 2u; // (rho infinite or NaN: a triangle without area)
    first = (uint32_t)lo;
    count = (uint32_t)n;
    return 1u;
}

// Appends the pairs of the lanes with code == 1 to the record's list (entry: primitive offset
// in the leaf | sample << 16), in lane order.  All lanes of the wave take part.
__device__ __forceinline__ void graze_push(uint32_t *__restrict__ list, uint32_t cap, uint32_t &n_list, uint32_t code,
                                           uint32_t prim, uint32_t first, uint32_t count) {
    const uint32_t mine = (code == 1u) ? count : 0u;
    if (!__any(mine != 0u)) return;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if ((threadIdx.x & 63u) >= (uint32_t)off) incl += up;
    }
    const uint32_t pos = n_list + incl - mine;
    for (uint32_t r = 0; r < mine; r++)
        if (pos + r < cap) list[pos + r] = prim | ((first + r) << 16);
    n_list += __shfl(incl, 63);
}

// One listed pair (triangle p, sample k0 of a record): the sample's start point o and its
// un-normalised direction D exactly as the sample loop has them (the closed form, or the running
// sum of raytracing.h:40 step by step; light sample k_idx of the record's sequence).  Nearly all
// pairs are cleared by the first part of the argument above alone: acceptance needs
// |A~| <= A_max = 29 eps sigma a b / m, here with a b <= (ell / 2)^2, m at its smallest (half
// the screen margin, and half of that as in graze_check; per_m = 4 / margin) and 7 eps a b for
// A~ against the value computed here, tested on the direction before it is normalised:
// |e1 . (D x e2)| > A_max |D|.  Returns false for a pair cleared that way; the others must go
// through the float test (graze_pair_hit).
__device__ __forceinline__ bool graze_pair_open(const DFrame &fr, float per_m, const DPrim *__restrict__ p, bool closed,
                                                f3 p0, f3 step, f3 dlt, uint32_t pixel, uint32_t leaf, uint32_t k0,
                                                uint32_t k_idx, f3 &o, f3 &D) {
    o = p0;
    if (closed) {
        const float s = (float)(k0 + 1u);
        o = f3_make(p0.x + s * dlt.x, p0.y + s * dlt.y, p0.z + s * dlt.z);
    } else {
        for (uint32_t k = 0; k <= k0; k++) o = f3_add(o, step);
    }
    f3 l = fr.light_pos;
    light_sample(fr.seed, pixel, leaf, k_idx, l.x, l.y);
    D = f3_sub(l, o);
    const f3 sv = f3_sub(o, p->a);
    const float A = f3_dot(p->b, f3_cross(D, p->c));
    const float sig = __builtin_amdgcn_sqrtf(f3_dot(sv, sv)) * 1.001f, ab = 0.25f * p->ell * p->ell;
    const float a_max = GRAZE_EPS * ab * (29.f * sig * per_m + 7.f) * 1.05f;
    return !(A * A > a_max * a_max * f3_dot(D, D) * 1.001f); // (a NaN leaves the pair open)
}
__device__ __forceinline__ bool graze_pair_hit(const DPrim *__restrict__ p, f3 o, f3 D) {
    const f3 d = f3_normalize(D); // raytracing.h:43
    float t;
    return tri_hit(o, d, p->a, p->b, p->c, t);
}

// The float test itself on the listed pairs of one record, one per lane.  True if any pair is
// accepted, or the list overflowed: the record is then shaded without screens (a few dozen
// records of a 4K frame).
__device__ __forceinline__ bool graze_resolve(const DFrame &fr, const DPrim *__restrict__ leaf_prims,
                                              const uint32_t *__restrict__ list, uint32_t cap, uint32_t n_list,
                                              bool closed, f3 p0, f3 step, f3 dlt, uint32_t pixel, uint32_t leaf,
                                              uint32_t k_off, uint32_t lane) {
    if (n_list > cap) return true;
    const float per_m = 4.f / fr.cull_margin;
    for (uint32_t base = 0; base < n_list; base += 64) {
        bool open = false;
        f3 o = p0, D = p0;
        const DPrim *__restrict__ p = leaf_prims;
        if (base + lane < n_list) {
            const uint32_t w = list[base + lane], k0 = w >> 16;
            p = leaf_prims + (w & 0xFFFFu);
            open = graze_pair_open(fr, per_m, p, closed, p0, step, dlt, pixel, leaf, k0, k_off + k0, o, D);
        }
        if (__any(open)) {
            const bool accepted = open && graze_pair_hit(p, o, D);
            if (__any(accepted)) return true;
        }
    }
    return false;
}

// Screens primitives [seg, seg + n) of the leaf (n <= CAND_MAX) and writes the
// survivors' offsets inside the leaf (seg + index), in order, to L.cand.  Returns how many.  G (may be
// null: nothing is dropped then) is the record's grazing guard: the pairs it cannot clear go
// to graze_list / n_graze.
__device__ __forceinline__ uint32_t build_candidates(ShadeLds &L, const DPrim *__restrict__ leaf_prims,
                                                      uint32_t seg, uint32_t n, uint32_t lane, float z_lo,
                                                      float z_hi, float margin, const GrazeCtx *G,
                                                      uint32_t *__restrict__ graze_list, uint32_t graze_cap,
                                                      uint32_t &n_graze) {
    uint32_t total = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t t = base + lane;
        bool keep = false;
        uint32_t code = 0, first = 0, cnt = 0;
        if (t < n) {
            const DPrim *p = leaf_prims + seg + t;
            keep = true;
            if ((p->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) {
                const float sep = pyramid_separation(L, p->a, p->b, p->c, z_lo, z_hi);
                keep = !(sep > margin);
                if (!keep && G) {
                    code = graze_check(*G, p, sep, first, cnt);
                    keep = code == 2u;
                }
            }
        }
        if (G) graze_push(graze_list, graze_cap, n_graze, code, seg + t, first, cnt);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            L.cand[pos] = (uint16_t)(seg + t);
        }
        total += (uint32_t)__popcll(mask);
    }
    return total;
}

// Start segment and margin of a record's screens.  closed: the start points are exactly
// p0 + k * dlt, k = 1 .. n; else they follow the rounded running sum of `step`, which
// leaves the line p0 + k * step by at most 2^-24 of the running value per step and
// component.  Returns the margin to screen with (3e38: do not screen this record).
__device__ __forceinline__ float screen_segment(bool closed, f3 p0, f3 step, f3 dlt, float n, float margin,
                                                f3 &o_first, f3 &o_last) {
    if (closed) {
        o_first = f3_add(p0, dlt);
        o_last = f3_make(p0.x + n * dlt.x, p0.y + n * dlt.y, p0.z + n * dlt.z);
        return margin;
    }
    o_first = p0;
    o_last = f3_add(p0, f3_scale(step, n + 1.f));
    const float mag = fmaxf(__builtin_fabsf(p0.x), __builtin_fabsf(o_last.x)) +
                      fmaxf(__builtin_fabsf(p0.y), __builtin_fabsf(o_last.y)) +
                      fmaxf(__builtin_fabsf(p0.z), __builtin_fabsf(o_last.z));
    const float drift = (n + 1.f) * 6.0e-8f * 1.01f * mag; // 2^-24 = 5.96e-8
    return (drift <= 3.f * margin) ? margin + drift : 3.0e38f; // (a NaN drift does not screen either)
}

// The candidate lists of the hit records as screen_kernel leaves them in device memory (the
// kernel without mirror bounces reads them; deeper bounce levels screen for themselves).
// Record r owns pool[cand_off[r] .. + leaf size); cand_n[r] = entries used | CAND_N_SORT when
// the entries carry cell masks.  Entry: offset of the primitive inside the leaf | cell mask << 16.
struct CandLists {
    const uint32_t *__restrict__ pool;
    const uint32_t *__restrict__ cand_off;
    const uint32_t *__restrict__ cand_n;
};
#define CAND_N_SORT 0x80000000u

// Screens primitives [seg, seg + seg_n) of the leaf for a level whose samples are taken in index
// order, grazing guard included: the survivors' offsets inside the leaf to L.cand, returns how
// many.  The faces must have been set up (setup_planes) for the same start points.
__device__ __forceinline__ uint32_t screen_plain_segment(const DFrame &fr, ShadeLds &L,
                                                          const DPrim *__restrict__ leaf_prims, uint32_t seg,
                                                          uint32_t seg_n, uint32_t lane, float z_lo, float z_hi,
                                                          float margin, bool closed, f3 o_first, f3 o_last, f3 p0,
                                                          f3 step, f3 dlt, uint32_t pixel, uint32_t leaf,
                                                          uint32_t k_off, uint32_t *__restrict__ graze_list,
                                                          uint32_t graze_cap, unsigned long long *stats) {
    // (a margin no triangle can exceed screens nothing: no guard needed)
#ifdef NO_GRAZE
    const bool screened = false; // (measurement build only: the screens without their guard)
#else
    const bool screened = margin < 1.0e38f;
#endif
    const GrazeCtx G = graze_ctx(closed, o_first, o_last, fr.n_lights, fr.cull_margin, margin);
    uint32_t n_graze = 0;
    uint32_t n_cand = build_candidates(L, leaf_prims, seg, seg_n, lane, z_lo, z_hi, margin, screened ? &G : nullptr,
                                       graze_list, graze_cap, n_graze);
    wave_lds_sync();
#ifdef STEP_STATS
    if (stats) {
        stats[10] += n_graze;      // pairs listed
        stats[11] += n_graze != 0; // records with a list
        stats[13] += seg_n;        // primitives screened
    }
#endif
    if (n_graze && graze_resolve(fr, leaf_prims, graze_list, graze_cap, n_graze, closed, p0, step, dlt, pixel, leaf, k_off, lane)) {
#ifdef STEP_STATS
        if (stats) stats[12]++; // records shaded without screens
#endif
        wave_lds_sync();
        n_cand = build_candidates(L, leaf_prims, seg, seg_n, lane, z_lo, z_hi, 3.0e38f, nullptr, graze_list, graze_cap,
                                  n_graze);
        wave_lds_sync();
    }
    (void)stats;
    return n_cand;
}

// One hit record, light samples in index order: one wave, lanes over the samples.
// BOUNCE = false is the build for max_depth = 0 (the reference as committed): the
// mirror-bounce code and its registers are compiled out.  IEEE selects the BSDF build.
struct ShadeTotals {
    uint32_t lit = 0, bounce_hits = 0;
    unsigned long long tests = 0, bounce_prims = 0;
#if defined(STEP_STATS) || defined(PHASE_TIMES)
    unsigned long long st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
};
// (measurement build only, -DPHASE_TIMES) wave cycles per phase of shade_record_sorted
#ifdef PHASE_TIMES
#define PHASE_MARK(k)                                                   \
    do {                                                                \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        tot.st[k] += now_ - phase_t;                                    \
        phase_t = now_;                                                 \
    } while (0)
#else
#define PHASE_MARK(k)
#endif
#ifdef STEP_STATS
// (measurement build only) why the rays of one candidate step fail
__device__ __forceinline__ void step_stats(ShadeTotals &tot, const ray2 &r, const DPrim &p, bool live1, bool live2) {
    if ((p.type_mat & 0xFFu) != RTX_PRIM_TRIANGLE) return;
    uint32_t why[2];
    for (int k = 0; k < 2; k++) {
        const f3 o = f3_make(r.ox[k], r.oy[k], r.oz[k]), d = f3_make(r.dx[k], r.dy[k], r.dz[k]);
        const f3 h = f3_cross(d, p.c);
        const float a = f3_dot(p.b, h), f = 1.f / a;
        const f3 s = f3_sub(o, p.a);
        const float u = f * f3_dot(s, h);
        const f3 q = f3_cross(s, p.b);
        const float v = f * f3_dot(d, q), t = f * f3_dot(p.c, q);
        why[k] = ((u < 0.f || u > 1.f) ? 1u : 0u) | ((v < 0.f || u + v > 1.f) ? 2u : 0u) | (!(t > 1e-7f) ? 4u : 0u);
        // 8: fails u or v by a clear margin; 16: behind by a clear margin
        why[k] |= ((u < -0.01f || u > 1.01f || v < -0.01f || u + v > 1.01f) ? 8u : 0u) | ((t < -1e-3f) ? 16u : 0u);
    }
    if (!live1) why[0] = 31u;
    if (!live2) why[1] = 31u;
    const bool all_u = __all((why[0] & 1u) && (why[1] & 1u));
    const bool all_uv = __all((why[0] & 3u) && (why[1] & 3u));
    const bool all_t = __all((why[0] & 4u) && (why[1] & 4u));
    const bool none = __all(why[0] && why[1]);
    const bool clear_uv = __all((why[0] & 8u) && (why[1] & 8u));
    const bool clear_t = __all((why[0] & 16u) && (why[1] & 16u));
    const bool clear_any = __all((why[0] & 24u) && (why[1] & 24u));
    tot.st[0]++;
    tot.st[1] += all_u;
    tot.st[2] += all_uv;
    tot.st[3] += all_t;
    tot.st[4] += none;
    tot.st[5] += clear_uv;
    tot.st[6] += clear_t;
    tot.st[7] += clear_any;
}
#endif

struct SortLds;
template <bool IEEE, bool PRE>
__device__ __forceinline__ float4 shade_level_sorted(const DFrame &fr, const DPrim *__restrict__ leaf_prims,
                                                     uint32_t count, uint32_t pixel, uint32_t leaf, const DMat &mat,
                                                     f3 nrm, f3 p0, f3 view, uint32_t k_off, SortLds &L,
                                                     uint32_t lane, ShadeTotals &tot, uint32_t &lit_out,
                                                     const uint32_t *__restrict__ pre, uint32_t pre_n);

// One hit record with its light samples in index order; the BOUNCE build adds the levels
// of the mirror bounces (raytracing.h:51-61) to the stored sum of the primary level.  LS:
// the wave's LDS for a level with sorted samples (it overlaps L), B: where the bounce build
// keeps the levels' sums and weights (neither is touched by the build without bounces).
template <bool BOUNCE, bool IEEE>
__device__ __forceinline__ void shade_record(const DFrame &fr, const DPrim *__restrict__ prims,
                                             const DMat *__restrict__ mats, const DRec *__restrict__ recs,
                                             float4 *__restrict__ colors, const CandLists &lists, ShadeLds &L,
                                             SortLds &LS, BounceLds &B, uint32_t lane, uint32_t ridx,
                                             ShadeTotals &tot) {
        const float nl = (float)fr.n_lights;
        uint32_t lit_total = 0, bounce_hits = 0;
        unsigned long long bounce_prims = 0, tests_total = 0;
#ifdef PHASE_TIMES
        unsigned long long phase_t = __builtin_readcyclecounter();
#endif
        const DRec rc = recs[ridx];
        const uint32_t pixel = rc.pixel, leaf = rc.leaf, count = rc.count;
        const DPrim *__restrict__ leaf_prims = prims + rc.first;
        // (build without bounces) the record's candidates, screened by screen_kernel
        const uint32_t *__restrict__ pre = nullptr;
        uint32_t pre_n = 0;
        if constexpr (!BOUNCE) {
            pre = lists.pool + lists.cand_off[ridx];
            pre_n = lists.cand_n[ridx] & ~CAND_N_SORT;
        }
        // state of the current hit: the primary one first, then one per mirror bounce
        // (raytracing.h:51-61; a bounce is traced inside the same leaf)
        DMat mat = mats[rc.mat];
        f3 nrm = f3_make(rc.nx, rc.ny, rc.nz);
        f3 p0 = f3_make(rc.px, rc.py, rc.pz);
        f3 o, d;
        if constexpr (BOUNCE) {
            primary_ray_uniform(fr, pixel % fr.width, pixel / fr.width, lane, o, d);
        } else { // screen_kernel has left the view direction (-d) in the place of the record's colour
            const float4 vw = colors[ridx];
            d = f3_make(-wave_uniform(vw.x), -wave_uniform(vw.y), -wave_uniform(vw.z));
        }
        uint32_t depth = 0;
        for (;;) {
        const f3 view = f3_scale(d, -1.f);      // raytracing.h:47
        const f3 step = f3_scale(nrm, fr.acne); // raytracing.h:40
        const uint32_t k_off = depth * fr.n_lights; // every bounce draws its own light samples
        // shadow-ray origins: closed form where provably exact, else the serial chain run
        // by lanes 0..2 (running origin component and its increment)
        f3 dlt = f3_make(0.f, 0.f, 0.f);
        const bool closed = acne_chain_delta3(lane, p0, step, fr.n_lights, dlt);
#ifdef STEP_STATS
        tot.st[9] += !closed;
#endif
        // (BOUNCE build: the samples of the primary hit have been added by the kernel without
        // bounces that ran before; this one starts from that sum and adds the deeper levels)
        const bool level_done = BOUNCE && depth == 0;
        // a deeper level whose leaf and start points allow it takes the samples sorted by light
        // cell, as the primary levels of such records do
        bool level_sorted = false;
        if constexpr (BOUNCE)
            level_sorted = depth > 0 && fr.sort_samples && closed && count >= SORT_MIN_LEAF && count <= CAND_MAX;
        float run = lane == 0 ? p0.x : lane == 1 ? p0.y : p0.z;
        const float inc = lane == 0 ? step.x : lane == 1 ? step.y : step.z;

        // pyramid of the record's shadow rays
        f3 o_first, o_last;
        const float margin = screen_segment(closed, p0, step, dlt, nl, fr.cull_margin, o_first, o_last);
        const float zl = fr.light_pos.z;
        const float o_zmin = fminf(o_first.z, o_last.z), o_zmax = fmaxf(o_first.z, o_last.z);
        const float z_lo = (zl > o_zmax) ? o_zmin : -3.0e38f; // all rays move towards the light plane in z
        const float z_hi = (zl < o_zmin) ? o_zmax : 3.0e38f;
#ifdef DUP_SETUP
        { // (measurement build only) the per-record set-up a second time, on other values
            f3 o2, d2;
            primary_ray_uniform(fr, (pixel + 1) % fr.width, pixel / fr.width, lane, o2, d2);
            f3 dlt2 = f3_make(0.f, 0.f, 0.f);
            const f3 q0 = f3_add(p0, d2);
            const bool c2 = acne_chain_delta3(lane, q0, step, fr.n_lights, dlt2);
            f3 of2, ol2;
            const float m2 = screen_segment(c2, q0, step, dlt2, nl, fr.cull_margin, of2, ol2);
            asm volatile("" ::"v"(m2));
            setup_planes(L, lane, of2, ol2, zl);
            wave_lds_sync();
        }
#endif
        if (BOUNCE && !level_done && !level_sorted) setup_planes(L, lane, o_first, o_last, zl);
        wave_lds_sync();
        PHASE_MARK(0);
        // What the passes walk: the leaf, screened here in segments of CAND_MAX (bounce build), or
        // the record's list from screen_kernel, copied to LDS in such segments.  A list that fits is
        // set up once, on the first pass.
        const uint32_t seg_total = BOUNCE ? count : pre_n;
        const bool one_segment = seg_total <= CAND_MAX;
        uint32_t n_cand = 0;

        f3 acc = f3_make(0.f, 0.f, 0.f);
        uint32_t lit = 0;
        if (level_sorted) {
            uint32_t lit_s = 0;
            const float4 sum = shade_level_sorted<IEEE, false>(fr, leaf_prims, count, pixel, leaf, mat, nrm, p0, view, k_off, LS,
                                                        lane, tot, lit_s, nullptr, 0u);
            acc = f3_make(sum.x, sum.y, sum.z);
            lit = lit_s;
            wave_lds_sync();
        }
        if (level_done && !closed && lane < 3) { // only the end of the chain is needed: where the bounce starts
            for (uint32_t sidx = 0; sidx < fr.n_lights; sidx++) run = run + inc;
            if (fr.n_lights) L.chain[((fr.n_lights - 1) % PASS_SAMPLES) * 3 + lane] = run;
        }
        if (level_done && !closed) wave_lds_sync(); // every lane reads the three components at the bounce
        // Each pass takes 128 samples: lane l owns samples base + l and base + 64 + l and
        // carries both rays through the triangle tests as packed pairs.
        for (uint32_t base = 0; !level_done && !level_sorted && base < fr.n_lights; base += PASS_SAMPLES) {
            const bool second = base + 64 < fr.n_lights; // wave-uniform: any sample in the upper half?
            const uint32_t k1 = base + lane, k2 = k1 + 64;
            const bool live1 = k1 < fr.n_lights, live2 = k2 < fr.n_lights;
            f3 p1, p2;
            if (closed) {
                const float s1 = (float)(k1 + 1u), s2 = (float)(k2 + 1u);
                p1 = f3_make(p0.x + s1 * dlt.x, p0.y + s1 * dlt.y, p0.z + s1 * dlt.z);
                p2 = f3_make(p0.x + s2 * dlt.x, p0.y + s2 * dlt.y, p0.z + s2 * dlt.z);
            } else {
                if (lane < 3) {
#pragma unroll
                    for (int s = 0; s < 64; s++) {
                        run = run + inc;
                        L.chain[s * 3 + lane] = run;
                    }
                    if (second) {
#pragma unroll
                        for (int s = 64; s < PASS_SAMPLES; s++) {
                            run = run + inc;
                            L.chain[s * 3 + lane] = run;
                        }
                    }
                }
                wave_lds_sync();
                p1 = f3_make(L.chain[lane * 3 + 0], L.chain[lane * 3 + 1], L.chain[lane * 3 + 2]);
                p2 = p1;
                if (second)
                    p2 = f3_make(L.chain[(64 + lane) * 3 + 0], L.chain[(64 + lane) * 3 + 1],
                                 L.chain[(64 + lane) * 3 + 2]);
            }
            f3 l1 = fr.light_pos;
            light_sample(fr.seed, pixel, leaf, k_off + k1, l1.x, l1.y);
            f3 l2 = l1; // (without an upper half the second ray repeats the first)
            if (second) light_sample(fr.seed, pixel, leaf, k_off + k2, l2.x, l2.y);
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
            for (uint32_t seg = 0; seg < seg_total; seg += CAND_MAX) {
                if (!one_segment || base == 0) {
                    wave_lds_sync();
                    const uint32_t seg_n = min(seg_total - seg, (uint32_t)CAND_MAX);
                    if constexpr (BOUNCE) {
#if defined(STEP_STATS)
                        unsigned long long *stats = tot.st;
#else
                        unsigned long long *stats = nullptr;
#endif
                        // (the pairs' list shares LDS with the chain of a level without the closed form:
                        // such a level lists nothing, and is shaded without screens if it would have to)
                        n_cand = screen_plain_segment(fr, L, leaf_prims, seg, seg_n, lane, z_lo, z_hi, margin, closed, o_first,
                                                      o_last, p0, step, dlt, pixel, leaf, k_off,
                                                      reinterpret_cast<uint32_t *>(L.chain), closed ? GRAZE_LIST : 0u, stats);
                    } else {
                        for (uint32_t c = lane; c < seg_n; c += 64) L.cand[c] = (uint16_t)pre[seg + c];
                        n_cand = seg_n;
                        wave_lds_sync();
                    }
                    PHASE_MARK(1);
                }
                if (n_cand == 0) continue;
                tests_total += (unsigned long long)n_cand * min(fr.n_lights - base, (uint32_t)PASS_SAMPLES);
                // walk the list with the next primitive's scalar loads in flight
                uint32_t ci = __builtin_amdgcn_readfirstlane((uint32_t)L.cand[0]);
                DPrim cur = leaf_prims[ci];
                for (uint32_t c = 0; c < n_cand; c++) {
                    const uint32_t cn = __builtin_amdgcn_readfirstlane(
                        (uint32_t)L.cand[(c + 1 < n_cand) ? c + 1 : c]);
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
            // lit unless something is hit inside [0, 1000]^3 (utils.h:12-17); the two
            // samples of a lane are added in sample order
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
        // sum over the lanes (fixed butterfly order, so a frame is reproducible; a level with
        // sorted samples comes back summed)
        if (!level_sorted) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                acc.x += __shfl_xor(acc.x, off);
                acc.y += __shfl_xor(acc.y, off);
                acc.z += __shfl_xor(acc.z, off);
                lit += __shfl_xor(lit, off);
            }
        }
        lit_total += lit;
        PHASE_MARK(7);
        if constexpr (!BOUNCE) { // one level: its sum is the record's colour
            if (lane == 0) colors[ridx] = make_float4(acc.x, acc.y, acc.z, 0.f);
            break;
        }
        if (lane == 0) B.level_sum[depth] = level_done ? colors[ridx] : make_float4(acc.x, acc.y, acc.z, 0.f);
        if (depth >= fr.max_depth) break;

        // ---- mirror bounce (raytracing.h:53-60) ----
        // the reflected ray starts from the hit position after all the acne steps
        f3 pos = p0;
        if (fr.n_lights && closed) {
            const float sn = (float)fr.n_lights;
            pos = f3_make(p0.x + sn * dlt.x, p0.y + sn * dlt.y, p0.z + sn * dlt.z);
        } else if (fr.n_lights) {
            const uint32_t slot = (fr.n_lights - 1) % PASS_SAMPLES;
            pos = f3_make(L.chain[slot * 3 + 0], L.chain[slot * 3 + 1], L.chain[slot * 3 + 2]);
        }
        const f3 n2 = f3_scale(nrm, 2.f); // reflect(), vec3.h:103-106: I - N * 2.f * dot(N, I)
        const f3 rd = f3_normalize(f3_sub(d, f3_scale(n2, f3_dot(nrm, d))));
        const f3 ro = f3_add(pos, f3_scale(rd, fr.acne));
        const float fres = rdm_fresnel(f3_dot(rd, nrm), mat);
        if (lane == 0) {
            const f3 w = f3_scale(mat.spec, fres);
            B.level_weight[depth] = make_float4(w.x, w.y, w.z, 0.f);
        }
        // nearest hit of the reflected ray in the leaf: lanes over the primitives, then
        // the smallest distance wins, the earliest primitive on ties (intersect.h:120-128)
        float bt = 0.f;
        uint32_t bk = RTXD_NONE;
        for (uint32_t kk = lane; kk < count; kk += 64) {
            const DPrim *pp = leaf_prims + kk;
            float t;
            bool ok;
            if ((pp->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE)
                ok = tri_hit(ro, rd, pp->a, pp->b, pp->c, t);
            else
                ok = sph_hit(ro, rd, pp->a, pp->b.y, t);
            if (ok && (bk == RTXD_NONE || t < bt)) {
                bt = t;
                bk = kk;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ot = __shfl_xor(bt, off);
            const uint32_t ok2 = __shfl_xor(bk, off);
            if (ok2 != RTXD_NONE && (bk == RTXD_NONE || ot < bt || (ot == bt && ok2 < bk))) {
                bt = ot;
                bk = ok2;
            }
        }
        bk = __builtin_amdgcn_readfirstlane(bk); // the same in every lane after the reduction
        bt = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(bt)));
        depth++;
        if (bk == RTXD_NONE) { // the bounce leaves the leaf: it contributes black
            if (lane == 0) B.level_sum[depth] = make_float4(0.f, 0.f, 0.f, 0.f);
            break;
        }
        bounce_hits++;
        bounce_prims += count;
        const DPrim *hp = leaf_prims + bk;
        const uint32_t htm = hp->type_mat;
        p0 = f3_add(ro, f3_scale(rd, bt)); // intersect.h:20, :58
        nrm = ((htm & 0xFFu) == RTX_PRIM_TRIANGLE) ? hp->n : f3_normalize(f3_sub(p0, hp->a));
        mat = mats[htm >> 8];
        d = rd;
        wave_lds_sync();
        } // bounce loop
        // colour of the primary hit = its own samples + weight * colour of the bounce, folded
        // from the deepest level (ret = color + specular * Fresnel * color_reflect, :60)
        if constexpr (BOUNCE) {
            wave_lds_sync();
            if (lane == 0) {
                float4 c = B.level_sum[depth];
                for (uint32_t lv = depth; lv-- > 0;) {
                    const float4 sm = B.level_sum[lv], w = B.level_weight[lv];
                    c = make_float4(sm.x + w.x * c.x, sm.y + w.y * c.y, sm.z + w.z * c.z, 0.f);
                }
                colors[ridx] = c;
            }
            wave_lds_sync();
        }
        tot.lit += lit_total;
        tot.tests += tests_total;
        tot.bounce_hits += bounce_hits;
        tot.bounce_prims += bounce_prims;
}
