// Device-side data layout and arithmetic shared by the kernels of rtx_amd.hip.
//
// Everything here is binary32 and is written so that, compiled with
// -ffp-contract=off (no fused multiply-add) and the default correctly rounded
// division and square root, each function returns bit for bit what the
// reference's C++ returns on x86-64 (vec3.h, intersect.h; g++ without -march:
// SSE scalar arithmetic, no contraction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTXD_NONE 0xFFFFFFFFu

struct f3 {
    float x, y, z;
};

// One box of the pruned pre-order array (32 B).  Boxes whose subtree holds no
// primitive are left out, which cannot change a frame: an empty leaf never
// yields a hit and a hit-free subtree folds to black (raytracing.h:69, :76).
//   inner box: link = index of the next box once this subtree is done or
//              missed; info = RTXD_NONE
//   leaf box:  link = first primitive reference of the leaf (the next box is
//              simply index + 1); info = count << 16 | scene-wide leaf number
struct DNode {
    float lb[3];
    float rt[3];
    uint32_t link;
    uint32_t info;
};

// Parent (index into the same array, RTXD_NONE at top level) and nesting
// depth (0 at top level) of a box; the colour merge needs them.
struct DUp {
    uint32_t parent;
    uint32_t depth;
};

struct DLeaf {
    uint32_t first;
    uint32_t count;
};

// One leaf primitive reference, 64 B, so that a wave that walks a leaf with a
// uniform index fetches it with one 16-dword scalar load.
//   triangle: a = v0, b = edge1 = v1 - v0, c = edge2 = v2 - v0 (intersect.h:33-34),
//             n = normalize(cross(edge1, edge2)) (intersect.h:59)
//   sphere:   a = centre, b.x = radius, b.y = radius * radius (intersect.h:6)
// The derived values are computed once on the host with the reference's
// operation order; they depend on the primitive only, not on the ray.
struct __attribute__((aligned(64))) DPrim {
    f3 a, b, c, n;
    uint32_t type_mat; // type | material << 8
    uint32_t id;
    // triangle: |e1| |e2| / |e1 x e2| and |e1| + |e2|, rounded up (the shading kernel's screens
    // bound the rounding of the intersection test with them); infinite for a triangle without area
    float shape, ell;
};

// Material with the ray-independent subexpressions of shade.h folded in.
struct __attribute__((aligned(16))) DMat {
    float ior;
    float roughness;
    float alpha2;    // roughness * roughness            (shade.h:14)
    float pi_alpha2; // PI * alpha2                      (shade.h:15)
    float ratio2;    // (1 / ior) * (1 / ior)            (shade.h:25-26)
    f3 spec;
    f3 dif_pi; // diffuseColor / PI                 (shade.h:70)
    // for the fast build of the BSDF
    float exp_scale;     // -log2(e) / alpha2
    float inv_pi_alpha2; // 1 / (PI * alpha2)
    float inv_alpha;     // 1 / roughness
    float pad[2];
};

// One (pixel, leaf) hit produced by the trace kernel and consumed by the
// shading and merge kernels (64 B).
struct __attribute__((aligned(16))) DRec {
    uint32_t pixel; // global pixel index j * width + i
    uint32_t node;  // index of the leaf's box in the pruned array
    uint32_t prim;  // global primitive id
    float t;
    float px, py, pz; // hit position (intersect.h:20, :58)
    uint32_t mat;
    float nx, ny, nz; // normal
    uint32_t next;    // next record of the same pixel, visit order
    uint32_t leaf;
    uint32_t first, count; // the leaf's primitive range
    uint32_t ordinal;
};

struct DFrame {
    uint32_t width, height;
    uint32_t rows; // rows rendered by this call
    uint32_t stripe_rows, n_parts, part;
    uint32_t n_lights, seed;
    uint32_t max_depth;
    uint32_t sort_samples; // records with large leaves are shaded with sorted light samples
    uint32_t wide_dets;    // some triangle has edges so long that a determinant may leave rcp2_exact's range
    float acne;
    float cull_margin; // margin of the candidate screens of the shading kernel, world units
    f3 camera;
    f3 light_pos;
    f3 light_color;
    float scale; // 1000 / height
    float half_width;
};

#define WORK_LISTS 8 // work lists of the shading kernel (classes of hit records)
#ifndef REC_SHARDS
#define REC_SHARDS 8 // parts of the record array, each with its own append counter
#endif

// Counters of one frame (device memory, zeroed before the trace kernel).
#define TILE_QUEUES 64u // = lanes of a wave (a wave looks at all of them at once)
#define TILE_RECTS 8    // pixel rectangles (one per top-level box) a frame's tiles are checked against
// Pixel rectangles (inclusive, global rows) outside of which no primary ray enters a box.
struct DTileRects {
    uint32_t n;
    int32_t x0[TILE_RECTS], x1[TILE_RECTS], y0[TILE_RECTS], y1[TILE_RECTS];
};
#ifndef QUEUE_HEADS
#define QUEUE_HEADS 64u // queue heads per work list (a power of two, at most 64: one per lane)
#endif
#define STAT_SHARDS 32u // copies of the frame's statistics (a power of two)
struct DCounters {
    uint32_t cand_total; // entries of the candidate pool handed out to the records (classify_kernel)
    uint32_t pad0[11];
    uint32_t pad1[52]; // (the measurement builds keep their statistics here)
    // Statistics of the frame, added up by every wave when it ends.  Atomics on one address
    // complete one after the other, about 80 million a second on this chip, whether anyone waits
    // for the result or not: two per wave on one line cost a kernel of 8192 short waves 0.14 ms
    // (more than the rest of it).  Hence STAT_SHARDS copies on lines of their own, picked by
    // block number; the host adds them up.
    struct {
        unsigned long long leaf_prims;   // sum of the leaf sizes over all records
        unsigned long long lit;          // light samples that were not occluded
        unsigned long long tests_run;    // shadow-ray primitive tests executed
        unsigned long long bounce_hits;  // mirror rays that hit something in their leaf
        unsigned long long bounce_prims; // sum of the leaf sizes over those
        uint32_t pad[22];
    } stat[STAT_SHARDS];
    // Hit records appended per shard of the record array (may exceed the shard's capacity).
    // The trace kernel appends with one returning atomic per wave and leaf; on one word
    // those would complete at ~50 M/s chip-wide and pace the kernel, so the array is cut
    // into REC_SHARDS equal parts with a counter each (a block uses shard blockIdx % REC_SHARDS).
    struct {
        uint32_t count;
        uint32_t pad[31];
    } shard[REC_SHARDS];
    // Work lists of the shading kernel (one per class of record): length, and the work
    // queue over it, sharded QUEUE_HEADS ways so that the returning atomic on one word (a
    // few tens of millions per second chip-wide) does not pace the kernel, least of all
    // when a list runs dry and every wave finds that out with an atomic: entry e belongs
    // to head e % QUEUE_HEADS; every word that is hit by atomics sits on a 128-byte line
    // of its own.
    struct {
        uint32_t count;
        uint32_t pad[31];
    } cls[WORK_LISTS];
    struct {
        uint32_t next;
        uint32_t pad[31];
    } head[WORK_LISTS][QUEUE_HEADS]; // the host reads the counters back up to here
    // Tile parts of the frame in which a primary ray may enter a top-level box, in TILE_QUEUES
    // lists (tile_list_kernel fills them, trace_kernel takes its work from them).
    struct {
        uint32_t count;
        uint32_t next;
        uint32_t pad[30];
    } tiles[TILE_QUEUES];
};

// ---- the sharded record array --------------------------------------------------
// Shard s holds records [s * cap, s * cap + min(count_s, cap)), cap = rec_cap / REC_SHARDS.
// RecMap enumerates the valid records of all shards as 0 .. total - 1.
struct RecMap {
    uint32_t start[REC_SHARDS]; // flat index of the first record of each shard
    uint32_t total, cap;
};
__device__ __forceinline__ RecMap rec_map(const DCounters *ctr, uint32_t rec_cap) {
    RecMap m;
    m.cap = rec_cap / REC_SHARDS;
    uint32_t run = 0;
#pragma unroll
    for (int s = 0; s < REC_SHARDS; s++) {
        m.start[s] = run;
        run += min(ctr->shard[s].count, m.cap);
    }
    m.total = run;
    return m;
}
__device__ __forceinline__ uint32_t rec_index(const RecMap &m, uint32_t flat) {
    uint32_t s = 0, first = 0;
#pragma unroll
    for (int k = 1; k < REC_SHARDS; k++) {
        const bool ge = flat >= m.start[k];
        s = ge ? (uint32_t)k : s;
        first = ge ? m.start[k] : first;
    }
    return s * m.cap + (flat - first);
}

// ---- vec3.h -----------------------------------------------------------------
__device__ __forceinline__ f3 f3_make(float x, float y, float z) {
    f3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
__device__ __forceinline__ f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 f3_mul(f3 a, f3 b) { return f3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 f3_scale(f3 a, float s) { return f3_make(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 f3_div(f3 a, float s) { return f3_make(a.x / s, a.y / s, a.z / s); }
// vec3.h:10-13: evaluated left to right
__device__ __forceinline__ float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vec3.h:28-31: three divisions by the norm
__device__ __forceinline__ f3 f3_normalize(f3 a) { return f3_div(a, __builtin_sqrtf(f3_dot(a, a))); }
// vec3.h:34-41
__device__ __forceinline__ f3 f3_cross(f3 u, f3 v) {
    return f3_make(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
// std::min / std::max as libstdc++ defines them; they differ from fminf /
// fmaxf when an operand is NaN, which the slab test produces for rays parallel
// to an axis (0 * inf).
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

// ---- intersect.h -------------------------------------------------------------
// intersect.h:28-67 with the ray-independent edges precomputed.
__device__ __forceinline__ bool tri_hit(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float &t) {
    const float eps = 0.0000001f;
    f3 h = f3_cross(d, e2);
    float a = f3_dot(e1, h);
    if (a > -eps && a < eps) return false;
    float f = 1.f / a;
    f3 s = f3_sub(o, v0);
    float u = f * f3_dot(s, h);
    if (u < 0.f || u > 1.f) return false;
    f3 q = f3_cross(s, e1);
    float v = f * f3_dot(d, q);
    if (v < 0.f || u + v > 1.f) return false;
    t = f * f3_dot(e2, q);
    return t > eps;
}

// a pair of binary32 values: arithmetic on it maps to the packed v_pk_*_f32 instructions
typedef float v2f __attribute__((ext_vector_type(2)));

// n / d for pairs of values that share their denominators, bit-identical to the correctly
// rounded IEEE quotient: the reciprocal is refined once per denominator (recip2_make) and
// every numerator then takes one multiply and four fused multiply-adds (div2_exact), the
// same operations in the same order as the compiler's division expansion when no operand
// scaling is needed.  That is the case for DIV2_DEN_MIN <= d <= DIV2_DEN_MAX and
// DIV2_NUM_MIN <= |n| <= 4 d (quotients of vector components by the vector's norm);
// verified over that range by rtx_selftest().  Elsewhere use the plain division.
#define DIV2_DEN_MIN 1.0e-12f
#define DIV2_DEN_MAX 1.0e12f
#define DIV2_NUM_MIN 1.0e-18f
struct recip2 {
    v2f d, r;
};
__device__ __forceinline__ recip2 recip2_make(v2f d) {
    const v2f one = (v2f){1.f, 1.f};
    const v2f r0 = (v2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const v2f e0 = __builtin_elementwise_fma(-d, r0, one);
    recip2 R;
    R.d = d;
    R.r = __builtin_elementwise_fma(e0, r0, r0);
    return R;
}
__device__ __forceinline__ v2f div2_exact(v2f n, const recip2 &R) {
    const v2f q0 = n * R.r;
    const v2f e1 = __builtin_elementwise_fma(-R.d, q0, n);
    const v2f q1 = __builtin_elementwise_fma(e1, R.r, q0);
    const v2f e2 = __builtin_elementwise_fma(-R.d, q1, n);
    return __builtin_elementwise_fma(e2, R.r, q1);
}

// sqrtf of two values at once, bit-identical to the correctly rounded square root for
// SQRT2_MIN <= x <= SQRT2_MAX: g = x * rsq(x) (v_rsq_f32, 1 ulp) is within 2 ulp of the
// root, the fused d = x - g * g is exact, and g + d * (rsq(x) / 2) then has a relative
// error of about 2^-47 before its rounding; no square root of a binary32 value lies that
// close to the midpoint of two neighbours.  rtx_selftest() compares with sqrtf for every
// value of the range.  (The compiler's expansion: v_sqrt_f32, two trial residuals one ulp
// below and above, range scaling and special cases, 16 instructions per value.)
#define SQRT2_MIN 1.0e-30f
#define SQRT2_MAX 1.0e30f
__device__ __forceinline__ v2f sqrt2_exact(v2f x) {
    const v2f y = (v2f){__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    const v2f g = x * y, h = 0.5f * y;
    const v2f d = __builtin_elementwise_fma(-g, g, x);
    return __builtin_elementwise_fma(d, h, g);
}

// f3_normalize of two vectors at once (x, y, z hold the pairs of components), bit for bit.
__device__ __forceinline__ void normalize2(v2f &x, v2f &y, v2f &z) {
    const v2f dd = x * x + y * y + z * z;
    // smallest |component| of each vector (a NaN component makes the squared norm NaN, which
    // fails the range test on it, so fminf dropping NaNs loses nothing)
    const float lo1 = __builtin_fminf(__builtin_fminf(__builtin_fabsf(x.x), __builtin_fabsf(y.x)), __builtin_fabsf(z.x));
    const float lo2 = __builtin_fminf(__builtin_fminf(__builtin_fabsf(x.y), __builtin_fabsf(y.y)), __builtin_fabsf(z.y));
    // squared norms whose root lies inside DIV2_DEN_MIN .. DIV2_DEN_MAX (and inside sqrt2_exact's range)
    const float dd_min = DIV2_DEN_MIN * DIV2_DEN_MIN * 1.001f, dd_max = DIV2_DEN_MAX * DIV2_DEN_MAX * 0.999f;
    const bool plain = dd.x >= dd_min && dd.x <= dd_max && dd.y >= dd_min && dd.y <= dd_max &&
                       lo1 >= DIV2_NUM_MIN && lo2 >= DIV2_NUM_MIN;
    if (__builtin_expect(__all(plain), 1)) {
        const recip2 R = recip2_make(sqrt2_exact(dd));
        x = div2_exact(x, R);
        y = div2_exact(y, R);
        z = div2_exact(z, R);
    } else { // a zero or extreme component somewhere in the wave
        const v2f len = (v2f){__builtin_sqrtf(dd.x), __builtin_sqrtf(dd.y)};
        x = x / len;
        y = y / len;
        z = z / len;
    }
}

// Two rays per lane against one triangle.  Every operation is the same IEEE binary32
// operation as in tri_hit, done on a pair of values at once so that it maps to the
// packed v_pk_mul_f32 / v_pk_add_f32 instructions (one issue slot for two results);
// no operation is fused or reordered.  ok1 / ok2: on entry whether the ray takes
// part, on return whether it hits (then t holds the distance).

struct ray2 {
    v2f ox, oy, oz; // origins
    v2f dx, dy, dz; // unit directions
};

// 1 / a for two values at once, bit-identical to the correctly rounded IEEE quotient
// 1.f / a for RCP2_MIN <= |a| < RCP2_MAX: the v_rcp_f32 seed (1 ulp) and one Newton step
// with fused multiply-adds, issued as packed v_pk_fma_f32.  The step leaves a relative
// error of about 2^-46 before its final rounding, and no reciprocal of a binary32 value
// lies that close to the midpoint of two neighbours: rtx_selftest() compares the result
// with 1.f / a for every value of the range (the compiler's expansion of the division
// goes on with two more correction steps, which never change a bit here).  Outside the
// range use the plain division.
#define RCP2_MIN 5.0e-8f
#define RCP2_MAX 1.0e18f
__device__ __forceinline__ v2f rcp2_exact(v2f a) {
    const v2f one = (v2f){1.f, 1.f};
    const v2f r0 = (v2f){__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
    const v2f e0 = __builtin_elementwise_fma(-a, r0, one);
    return __builtin_elementwise_fma(e0, r0, r0);
}

// tri_hit on a pair of (ray, triangle) problems at once.  R is the type of the ray's
// components and P the type of the triangle's: (v2f, float) tests two rays against one
// triangle, (float, v2f) one ray against two triangles; every operation is the IEEE
// operation tri_hit performs, on both problems in one packed instruction.  EARLY_OUT =
// false runs every problem to the end (straight-line code: worth it where the wave has
// the SIMD to itself and the branches of the early exits cost more than the arithmetic).
// ok1 / ok2: whether the problem takes part.  on_hit(hit1, hit2, t) is called once all
// tests are done, unless an early exit found that neither problem can hit: what the
// caller does with a hit (keep the nearest) then sits behind the exits, in the part of
// the code that few steps reach, and no distance has to exist on the other paths.
template <typename R, typename P, bool EARLY_OUT = true, typename F>
__device__ __forceinline__ void tri_hit_pk(R ox, R oy, R oz, R dx, R dy, R dz, P v0x, P v0y, P v0z, P e1x, P e1y,
                                           P e1z, P e2x, P e2y, P e2z, bool check_wide, bool ok1, bool ok2,
                                           F &&on_hit) {
    const float eps = 0.0000001f;
    const v2f hx = dy * e2z - dz * e2y;
    const v2f hy = dz * e2x - dx * e2z;
    const v2f hz = dx * e2y - dy * e2x;
    const v2f a = e1x * hx + e1y * hy + e1z * hz;
    // intersect.h:39 gives up when -eps < a < eps.  That test is made on f = 1 / a, at the
    // end: for the correctly rounded f, |a| < eps exactly when |f| > 1e7f (1 / eps rounds to
    // 1e7f and the reciprocal of the float below eps to 10000001); a = 0 or a subnormal a
    // make f infinite or NaN, and `|f| <= 1e7f` is false for these; a NaN or infinite a
    // makes t NaN or zero in the reference, which fails `t > eps` there.  A problem that fails it goes on with values nobody uses.
    // (The upper end of rcp2_exact's range needs a check only in a scene with triangle edges
    // so long that |a| <= |e1| |e2| could reach it.)
    v2f f = rcp2_exact(a);
    if (__builtin_expect(check_wide, 0)) {
        if (__any((ok1 && !(__builtin_fabsf(a.x) < RCP2_MAX)) || (ok2 && !(__builtin_fabsf(a.y) < RCP2_MAX))))
            f = 1.f / a;
    }
    const v2f sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
    const v2f u = f * (sx * hx + sy * hy + sz * hz);
    ok1 = ok1 && !(u.x < 0.f || u.x > 1.f);
    ok2 = ok2 && !(u.y < 0.f || u.y > 1.f);
    if (EARLY_OUT && !(ok1 || ok2)) return;
    const v2f qx = sy * e1z - sz * e1y;
    const v2f qy = sz * e1x - sx * e1z;
    const v2f qz = sx * e1y - sy * e1x;
    const v2f v = f * (dx * qx + dy * qy + dz * qz);
    const v2f uv = u + v;
    ok1 = ok1 && !(v.x < 0.f || uv.x > 1.f);
    ok2 = ok2 && !(v.y < 0.f || uv.y > 1.f);
    if (EARLY_OUT && !(ok1 || ok2)) return;
    const v2f t = f * (e2x * qx + e2y * qy + e2z * qz);
    const float f_max = 10000000.f; // 1 / eps
    on_hit(ok1 && __builtin_fabsf(f.x) <= f_max && t.x > eps, ok2 && __builtin_fabsf(f.y) <= f_max && t.y > eps, t);
}

// two rays, one triangle
template <typename F>
__device__ __forceinline__ void tri_hit2(const ray2 &r, f3 v0, f3 e1, f3 e2, bool check_wide, bool ok1, bool ok2,
                                         F &&on_hit) {
    tri_hit_pk<v2f, float>(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, v0.x, v0.y, v0.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z,
                           check_wide, ok1, ok2, on_hit);
}

// intersect.h:1-18 (distance only)
__device__ __forceinline__ bool sph_hit(f3 o, f3 d, f3 c, float r2, float &t) {
    f3 pos = f3_sub(o, c);
    float a = 1.f;
    float b = 2.f * f3_dot(d, pos);
    float cc = f3_dot(pos, pos) - r2;
    float delta = b * b - 4.f * a * cc;
    if (delta <= 0.f) return false;
    float sq = __builtin_sqrtf(delta);
    float t1 = (-b - sq) / (2.f * a);
    float t2 = (-b + sq) / (2.f * a);
    t = t1 > 0.f ? t1 : t2 > 0.f ? t2 : -1.f;
    return t > 0.f;
}

// intersect.h:69-93 with 1 / direction hoisted out of the per-box call (the
// reference recomputes the same three quotients at every box).
__device__ __forceinline__ bool box_hit(f3 o, f3 inv, const float *lb, const float *rt) {
    float t1 = (lb[0] - o.x) * inv.x;
    float t2 = (rt[0] - o.x) * inv.x;
    float t3 = (lb[1] - o.y) * inv.y;
    float t4 = (rt[1] - o.y) * inv.y;
    float t5 = (lb[2] - o.z) * inv.z;
    float t6 = (rt[2] - o.z) * inv.z;
    float tmin = std_max(std_max(std_min(t1, t2), std_min(t3, t4)), std_min(t5, t6));
    float tmax = std_min(std_min(std_max(t1, t2), std_max(t3, t4)), std_max(t5, t6));
    if (tmax < 0.f) return false;
    if (tmin > tmax) return false;
    return true;
}

// utils.h:12-17
__device__ __forceinline__ bool out_of_scene(f3 c) {
    return c.x < 0.f || c.x > 1000.f || c.y < 0.f || c.y > 1000.f || c.z < 0.f || c.z > 1000.f;
}

// ---- running sum of the acne offset (raytracing.h:40) ---------------------------
// The shadow-ray origin of sample k is p_k = fl(p_{k-1} + c) with p_0 the hit position
// and c one component of normal * acne: a serial chain of rounded additions.  While
// the running value stays inside one binade [2^e, 2^(e+1)) it is a multiple of the
// binade's spacing u = 2^(e-23), and adding c rounds to the nearest multiple of u:
// every step adds the same delta = u * rne(c / u), exactly, so p_k = p_0 + k * delta
// with a single exact multiply and add.  That holds when c / u is not halfway between
// integers (the rounding would then depend on the parity of p_k), k * delta is exact,
// p_(n+1) still has p_0's sign and exponent (so no p_k, k <= n, leaves the binade),
// and p_0 is not on the binade's lower edge heading towards zero.  Returns false when it cannot be guaranteed; the caller then runs the chain
// step by step.  rtx_selftest() checks it against the serial chain.
__device__ __forceinline__ bool acne_chain_delta(float p0, float c, uint32_t n, float &delta) {
    const uint32_t bits = __float_as_uint(p0);
    const uint32_t ex = (bits >> 23) & 0xFFu;
    if (ex < 25u || ex > 250u) return false; // zero, subnormal, tiny, huge, inf, NaN
    // exactly on the lower edge of its binade and moving towards zero: the sum falls
    // into the finer grid below the edge even when it would round back to p_0 on u
    if ((bits & 0x7FFFFFu) == 0u && ((__float_as_uint(c) ^ bits) >> 31) && c != 0.f) return false;
    const float u = __uint_as_float((ex - 23u) << 23);
    // c / u as a product: 1 / u = 2^(150 - ex) is a normal power of two for every ex admitted
    // above, and the correctly rounded product c * (1 / u) is the correctly rounded quotient
    // c / u (the same real number)
    const float q = c * __uint_as_float((277u - ex) << 23);
    if (!(__builtin_fabsf(q) * (float)(n + 1u) < 8388608.f)) return false;
    const float r = __builtin_rintf(q);
    if (__builtin_fabsf(q - r) == 0.5f) return false;
    delta = r * u;
    const float pend = p0 + (float)(n + 1u) * delta;
    return (__float_as_uint(pend) >> 23) == (bits >> 23);
}

// acne_chain_delta for the three components of a wave-uniform hit position and step: lanes
// 0, 1 and 2 take one component each (the same arithmetic; done component after component in
// every lane it costs three times the instructions).  All lanes of the wave must be active.
__device__ __forceinline__ bool acne_chain_delta3(uint32_t lane, f3 p0, f3 step, uint32_t n, f3 &dlt) {
    const float pc = lane == 0 ? p0.x : lane == 1 ? p0.y : p0.z;
    const float sc = lane == 0 ? step.x : lane == 1 ? step.y : step.z;
    float dc = 0.f;
    const bool ok = acne_chain_delta(pc, sc, n, dc);
    const int bits = __float_as_int(dc);
    dlt.x = __int_as_float(__builtin_amdgcn_readlane(bits, 0));
    dlt.y = __int_as_float(__builtin_amdgcn_readlane(bits, 1));
    dlt.z = __int_as_float(__builtin_amdgcn_readlane(bits, 2));
    return (__ballot(ok) & 7ull) == 7ull;
}

// ---- light sampler -------------------------------------------------------------
// Philox-2x32-10, key = seed, counter = (pixel, leaf << 16 | sample); each
// output word is mapped to 0..500 by a 32x32 -> high-32 multiply with 501.
// The two integers 0..500 of sample k (the reference's dist(gen), raytracing.h:38-39).
__device__ __forceinline__ void light_sample_u(uint32_t seed, uint32_t pixel, uint32_t leaf,
                                               uint32_t k, uint32_t &ux, uint32_t &uy) {
    uint32_t c0 = pixel, c1 = (leaf << 16) | k, key = seed;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        unsigned long long prod = (unsigned long long)0xD256D193u * c0;
        uint32_t hi = (uint32_t)(prod >> 32), lo = (uint32_t)prod;
        c0 = hi ^ key ^ c1;
        c1 = lo;
        key += 0x9E3779B9u;
    }
    ux = __umulhi(c0, 501u);
    uy = __umulhi(c1, 501u);
}

__device__ __forceinline__ void light_sample(uint32_t seed, uint32_t pixel, uint32_t leaf,
                                             uint32_t k, float &lx, float &ly) {
    uint32_t ux, uy;
    light_sample_u(seed, pixel, leaf, k, ux, uy);
    lx = 250.f + (float)ux; // raytracing.h:38
    ly = 250.f + (float)uy; // raytracing.h:39
}

// ---- primary ray (raytracing.h:119-121) ----------------------------------------
// World units are the pixels of a 1000-row image; at 1000 x 1000 the origin is
// exactly (i, j, 1000).
__device__ __forceinline__ void primary_ray(const DFrame &fr, uint32_t i, uint32_t j, f3 &o, f3 &d) {
    o = f3_make(500.f + ((float)i - fr.half_width) * fr.scale, (float)j * fr.scale, 1000.f);
    d = f3_normalize(f3_sub(o, fr.camera));
}

// primary_ray for a wave-uniform pixel: the three divisions of normalize (vec3.h:28-31) in
// lanes 0, 1 and 2 at once.  All lanes of the wave must be active.
__device__ __forceinline__ void primary_ray_uniform(const DFrame &fr, uint32_t i, uint32_t j, uint32_t lane, f3 &o,
                                                    f3 &d) {
    o = f3_make(500.f + ((float)i - fr.half_width) * fr.scale, (float)j * fr.scale, 1000.f);
    const f3 v = f3_sub(o, fr.camera);
    const float len = __builtin_sqrtf(f3_dot(v, v));
    const float q = (lane == 0 ? v.x : lane == 1 ? v.y : v.z) / len;
    const int bits = __float_as_int(q);
    d.x = __int_as_float(__builtin_amdgcn_readlane(bits, 0));
    d.y = __int_as_float(__builtin_amdgcn_readlane(bits, 1));
    d.z = __int_as_float(__builtin_amdgcn_readlane(bits, 2));
}

// The same value in every lane, moved to a scalar register (a float that lives as long as a record
// is shaded is cheaper there than in a vector register).
__device__ __forceinline__ float wave_uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Global row of local row r of this part (rows are dealt to parts in stripes).
__device__ __forceinline__ uint32_t part_row(const DFrame &fr, uint32_t r) {
    if (fr.n_parts <= 1) return r;
    uint32_t s = r / fr.stripe_rows;
    return (s * fr.n_parts + fr.part) * fr.stripe_rows + (r - s * fr.stripe_rows);
}
