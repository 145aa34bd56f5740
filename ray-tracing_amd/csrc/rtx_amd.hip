// rtx_amd.hip - MI355X (gfx950) render hot path behind the C ABI of include/rtx.h.
//
// One frame = four kernels on one stream, no host round trip in between:
//   trace_kernel     one lane per pixel: primary ray, box walk, nearest hit per
//                    leaf (raytracing.h:74-98, :29-34; intersect.h) -> hit records
//   classify_kernel  one lane per hit record: the record goes to the work list of its
//                    class (how it will be shaded, how heavy its leaf is)
//   shade_kernel     one wave per hit record, lanes over the light samples: shadow
//                    rays against the record's leaf + BSDF (raytracing.h:36-50,
//                    shade.h) -> one colour per record   (shade_few_kernel: several
//                    records per wave when there are at most 32 samples)
//   merge_kernel     one lane per pixel: per-nesting-level dominance fold of the
//                    record colours, clamp, truncate (raytracing.h:85-93, :126-130)
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off (see Makefile).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_device.h"

// ===========================================================================
// Kernels
// ===========================================================================

#define TRACE_BLOCK 256
#ifndef TRACE_EARLY_OUT
#define TRACE_EARLY_OUT true
#endif
#ifndef TRACE_STAGE
#define TRACE_STAGE 64 // primitives of a leaf staged in LDS at a time (40 B each, per wave)
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

// ---- trace ------------------------------------------------------------------
// One wave = the 64 primary rays of an 8 x 8 pixel tile.  The wave walks the pre-order
// box array ONCE for all its rays: `n` is wave-uniform, lane l takes part in box n when
// its own next box n_lane == n.  After an inner box the walk goes to n + 1 if any ray
// entered it and to its skip link otherwise; a ray that missed an ancestor m waits
// with n_lane = link(m), and because boxes nest, every waiting ray's link is at or
// beyond the link of the box just missed, so the uniform walk can never step past a
// waiting ray and visits exactly the union of the boxes its rays visit, each ray
// seeing its own boxes in the reference's order (raytracing.h:74-98).  Box and
// triangle data are therefore wave-uniform: scalar loads (next one already in flight),
// used as SGPR operands; only the rays live in VGPRs.
__global__ __launch_bounds__(TRACE_BLOCK) void trace_kernel(
    DFrame fr, const DNode *__restrict__ nodes, uint32_t n_nodes, const DPrim *__restrict__ prims,
    DRec *__restrict__ recs, uint32_t rec_cap, uint32_t *__restrict__ first_rec,
    DCounters *__restrict__ ctr) {
    const uint32_t tiles_x = (fr.width + 15) / 16;
    const uint32_t n_tiles = tiles_x * ((fr.rows + 15) / 16);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ __attribute__((aligned(16))) float stage[TRACE_BLOCK / 64][(TRACE_STAGE / 2) * 20];
    float *S = stage[wave];
    // the block appends its hit records to one shard of the record array
    const uint32_t shard = blockIdx.x % REC_SHARDS, shard_cap = rec_cap / REC_SHARDS;
    unsigned long long leaf_prims = 0;
    // a block takes the 16 x 16 pixel tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t bx = tile % tiles_x, by = tile / tiles_x;
    const uint32_t i = bx * 16 + (wave & 1) * 8 + (lane & 7);
    const uint32_t r = by * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool valid = i < fr.width && r < fr.rows;
    if (!__any(valid) || n_nodes == 0) {
        if (valid) first_rec[r * fr.width + i] = RTXD_NONE;
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
                    float4 q0 = S4[0], q1 = S4[1], q2 = S4[2], q3 = S4[3], q4 = S4[4];
                    for (uint32_t pr = 0; pr < n_pairs; pr++) {
                        const uint32_t pn = (pr + 1 < n_pairs) ? pr + 1 : pr;
                        const float4 n0 = S4[pn * 5 + 0], n1 = S4[pn * 5 + 1], n2 = S4[pn * 5 + 2],
                                     n3 = S4[pn * 5 + 3], n4 = S4[pn * 5 + 4];
                        const uint32_t type_a = __builtin_amdgcn_readfirstlane(__float_as_uint(q4.z));
                        const uint32_t type_b = __builtin_amdgcn_readfirstlane(__float_as_uint(q4.w));
                        if (hit) {
                            const uint32_t ka = base + 2 * pr;
                            if (type_a == RTX_PRIM_TRIANGLE && type_b == RTX_PRIM_TRIANGLE) {
                                bool ok_a = true, ok_b = true;
                                v2f t = (v2f){0.f, 0.f};
                                tri_hit_pk<float, v2f, TRACE_EARLY_OUT>(o.x, o.y, o.z, d.x, d.y, d.z, (v2f){q0.x, q0.y}, (v2f){q0.z, q0.w},
                                                       (v2f){q1.x, q1.y}, (v2f){q1.z, q1.w}, (v2f){q2.x, q2.y},
                                                       (v2f){q2.z, q2.w}, (v2f){q3.x, q3.y}, (v2f){q3.z, q3.w},
                                                       (v2f){q4.x, q4.y}, fr.wide_dets != 0, ok_a, ok_b, t);
                                if (ok_a && (best == RTXD_NONE || t.x < best_t)) {
                                    best_t = t.x;
                                    best = ka;
                                }
                                if (ok_b && (best == RTXD_NONE || t.y < best_t)) {
                                    best_t = t.y;
                                    best = ka + 1;
                                }
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
                const bool found = best != RTXD_NONE;
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
    if (valid) first_rec[local] = head;
    } // tiles
    if (lane == 0 && leaf_prims) atomicAdd(&ctr->leaf_prims, leaf_prims);
}

// ---- shade.h -------------------------------------------------------------------
// The BSDF is pure colour arithmetic: nothing it computes feeds a hit, an occlusion or a
// fold decision other than through the final colour, which is compared with the CPU to
// 1/255.  It has two builds.  The IEEE build below follows shade.h operation by
// operation with correctly rounded division, square root and expf (what the rest of the
// file uses).  The fast build further down uses the hardware's 1-ulp v_rcp / v_rsq /
// v_sqrt / v_exp, about a tenth of the instructions per operation; everything that feeds
// a cancellation (the half vector, the dot products, 1 - cos^2) is computed the same way
// in both builds, so the two differ by a few ulp of the result, not by an amplified error.

// shade.h:7-18
__device__ __forceinline__ float rdm_beckmann(float NdotH, const DMat &m) {
    if (NdotH > 0.f) {
        float cos2 = NdotH * NdotH;
        float tan2 = (1.f - cos2) / cos2;
        float cos4 = cos2 * cos2;
        return expf(-tan2 / m.alpha2) / (m.pi_alpha2 * cos4);
    }
    return 0.f;
}

// shade.h:22-33 with extIOR = 1 (the only call site, shade.h:61); powf(x, 2)
// is taken as x * x.
__device__ __forceinline__ float rdm_fresnel(float cosI, const DMat &m) {
    float sinT2 = m.ratio2 * (1.f - (cosI * cosI));
    if (sinT2 > 1.f) return 1.f;
    float cosT = __builtin_sqrtf(1.f - sinT2);
    float a = cosI - cosT * m.ior, b = cosI + cosT * m.ior;
    float c = cosT - cosI * m.ior, e = cosT + cosI * m.ior;
    float Rs = (a * a) / (b * b);
    float Rp = (c * c) / (e * e);
    return 0.5f * (Rs + Rp);
}

// shade.h:35-47
__device__ __forceinline__ float rdm_g1(float DdotH, float DdotN, float alpha) {
    float tn = __builtin_sqrtf(1.f - DdotN * DdotN) / DdotN;
    float b = 1.f / (alpha * tn);
    float k = DdotH / DdotN;
    if (k > 0.f) {
        if (b < 1.6f) return (3.535f * b + 2.181f * b * b) / (1.f + 2.276f * b + 2.577f * b * b);
        return 1.f;
    }
    return 0.f;
}

// shade.h:80-90 with RDM_bsdf, _s and _d inlined, times the per-sample weight
// 20 / n_lights of raytracing.h:48.  RDM_bsdf receives LdotH in the place of LdotN
// (shade.h:88); the true LdotN is only the final cosine.
__device__ __forceinline__ f3 shade_ieee(f3 n, f3 v, f3 l, f3 lc, const DMat &m, float nl) {
    f3 h = f3_normalize(f3_add(v, l));
    float LdotH = f3_dot(l, h);
    float VdotH = f3_dot(v, h);
    float NdotH = f3_dot(n, h);
    float LdotN = f3_dot(l, n);
    float VdotN = f3_dot(v, n);
    float dd = rdm_beckmann(NdotH, m);
    float ff = rdm_fresnel(LdotH, m);
    float gg = rdm_g1(LdotH, LdotH, m.roughness) * rdm_g1(VdotH, VdotN, m.roughness);
    f3 s = f3_div(f3_scale(f3_scale(f3_scale(m.spec, dd), ff), gg), 4.f * LdotH * VdotN);
    f3 bsdf = f3_add(m.dif_pi, s);
    return f3_scale(f3_div(f3_scale(f3_mul(lc, bsdf), LdotN), nl), 20.f);
}

// The fast build of the BSDF, written once for one sample (T = float) and for the two
// samples of a lane (T = v2f, packed arithmetic; lx, ly, lz are the unit directions to the
// light): the same operations either way, so both give the same bits.  The half vector
// and the dot products are computed exactly as shade() computes them, the rest is
// regrouped around fewer hardware reciprocals (tan^2 and 1/cos^4 share one, the two
// Fresnel quotients are squared after the division, G1 takes one reciprocal square root,
// and G1 of the view direction depends on the sample only through a sign).  Same special
// cases as shade.h: D = 0 for NdotH <= 0, F = 1 beyond the critical angle, G1 = 0 unless
// DdotH / DdotN > 0, = 1 for b >= 1.6 or NaN.
__device__ __forceinline__ float rcp_t(float a) { return __builtin_amdgcn_rcpf(a); }
__device__ __forceinline__ v2f rcp_t(v2f a) { return (v2f){__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ float sqrt_t(float a) { return __builtin_amdgcn_sqrtf(a); }
__device__ __forceinline__ v2f sqrt_t(v2f a) { return (v2f){__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)}; }
__device__ __forceinline__ float rsq_t(float a) { return __builtin_amdgcn_rsqf(a); }
__device__ __forceinline__ v2f rsq_t(v2f a) { return (v2f){__builtin_amdgcn_rsqf(a.x), __builtin_amdgcn_rsqf(a.y)}; }
__device__ __forceinline__ float exp2_t(float a) { return __builtin_amdgcn_exp2f(a); }
__device__ __forceinline__ v2f exp2_t(v2f a) { return (v2f){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
// c > th ? a : b and c < th ? a : b, per element
__device__ __forceinline__ float if_gt(float c, float th, float a, float b) { return c > th ? a : b; }
__device__ __forceinline__ v2f if_gt(v2f c, float th, v2f a, v2f b) {
    return (v2f){c.x > th ? a.x : b.x, c.y > th ? a.y : b.y};
}
__device__ __forceinline__ float if_lt(float c, float th, float a, float b) { return c < th ? a : b; }
__device__ __forceinline__ v2f if_lt(v2f c, float th, v2f a, v2f b) {
    return (v2f){c.x < th ? a.x : b.x, c.y < th ? a.y : b.y};
}
__device__ __forceinline__ void normalize_t(float &x, float &y, float &z) {
    const f3 h = f3_normalize(f3_make(x, y, z));
    x = h.x;
    y = h.y;
    z = h.z;
}
__device__ __forceinline__ void normalize_t(v2f &x, v2f &y, v2f &z) { normalize2(x, y, z); }

template <typename T>
__device__ __forceinline__ void bsdf_fast(f3 n, f3 v, T lx, T ly, T lz, f3 lc, const DMat &m, float nl, T &rx,
                                          T &ry, T &rz) {
    const T zero = lx - lx, one = zero + 1.f;
    T hx = v.x + lx, hy = v.y + ly, hz = v.z + lz;
    normalize_t(hx, hy, hz);
    const T LdotH = lx * hx + ly * hy + lz * hz;
    const T VdotH = v.x * hx + v.y * hy + v.z * hz;
    const T NdotH = n.x * hx + n.y * hy + n.z * hz;
    const T LdotN = lx * n.x + ly * n.y + lz * n.z;
    const float VdotN = f3_dot(v, n);
    // Beckmann (shade.h:7-18)
    const T cos2 = NdotH * NdotH;
    const T inv_cos2 = rcp_t(cos2);
    T dd = exp2_t(((1.f - cos2) * inv_cos2) * m.exp_scale) * (inv_cos2 * inv_cos2) * m.inv_pi_alpha2;
    dd = if_gt(NdotH, 0.f, dd, zero);
    // Fresnel (shade.h:22-33), cosI = LdotH
    const T sinT2 = m.ratio2 * (1.f - LdotH * LdotH);
    const T cosT = sqrt_t(1.f - sinT2);
    const T ct = cosT * m.ior, ci = LdotH * m.ior;
    const T ra = (LdotH - ct) * rcp_t(LdotH + ct);
    const T rc = (cosT - ci) * rcp_t(cosT + ci);
    const T ff = if_gt(sinT2, 1.f, one, 0.5f * (ra * ra + rc * rc));
    // G1(LdotH, LdotH) (shade.h:35-47): DdotH / DdotN = 1 unless LdotH is 0 or NaN
    const T ll = LdotH * LdotH;
    const T bl = LdotH * rsq_t(1.f - ll) * m.inv_alpha;
    const T bl2 = bl * bl;
    T gl = if_lt(bl, 1.6f, (3.535f * bl + 2.181f * bl2) * rcp_t(1.f + 2.276f * bl + 2.577f * bl2), one);
    gl = if_gt(ll, 0.f, gl, zero);
    // G1(VdotH, VdotN): its value depends on the view direction only, the sample decides the sign test
    const float bv = VdotN * __builtin_amdgcn_rsqf(1.f - VdotN * VdotN) * m.inv_alpha;
    float gv1 = 1.f;
    if (bv < 1.6f)
        gv1 = (3.535f * bv + 2.181f * bv * bv) * __builtin_amdgcn_rcpf(1.f + 2.276f * bv + 2.577f * bv * bv);
    const T gv = if_gt(VdotH * VdotN, 0.f, zero + gv1, zero);
    // shade.h:80-90; raytracing.h:48
    const T sc = (dd * ff) * (gl * gv) * rcp_t((4.f * VdotN) * LdotH);
    const T wl = LdotN * (20.f * __builtin_amdgcn_rcpf(nl));
    rx = (lc.x * (m.dif_pi.x + m.spec.x * sc)) * wl;
    ry = (lc.y * (m.dif_pi.y + m.spec.y * sc)) * wl;
    rz = (lc.z * (m.dif_pi.z + m.spec.z * sc)) * wl;
}

// One sample: the weighted BSDF colour (raytracing.h:46-48), IEEE or fast build.
template <bool IEEE>
__device__ __forceinline__ f3 shade1(f3 n, f3 v, f3 l, f3 lc, const DMat &m, float nl) {
    if constexpr (IEEE) {
        return shade_ieee(n, v, l, lc, m, nl);
    } else {
        float rx, ry, rz;
        bsdf_fast<float>(n, v, l.x, l.y, l.z, lc, m, nl, rx, ry, rz);
        return f3_make(rx, ry, rz);
    }
}

// The two samples of a lane.
template <bool IEEE>
__device__ __forceinline__ void shade2(f3 n, f3 v, v2f lx, v2f ly, v2f lz, bool want1, bool want2, f3 lc,
                                       const DMat &m, float nl, f3 &c1, f3 &c2) {
    if constexpr (IEEE) {
        if (want1) c1 = shade_ieee(n, v, f3_make(lx.x, ly.x, lz.x), lc, m, nl);
        if (want2) c2 = shade_ieee(n, v, f3_make(lx.y, ly.y, lz.y), lc, m, nl);
    } else {
        v2f rx, ry, rz;
        bsdf_fast<v2f>(n, v, lx, ly, lz, lc, m, nl, rx, ry, rz);
        c1 = f3_make(rx.x, ry.x, rz.x);
        c2 = f3_make(rx.y, ry.y, rz.y);
    }
}

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
// shadow ray (intersect.h:99-117).  All shadow rays of one record start within
// n_lights * acne of the hit point and aim at the light rectangle
// [250, 750]^2 x {z_light} (raytracing.h:38-39), so they stay inside the pyramid
// spanned by the start points and that rectangle (continued beyond it: there is
// no far limit).  A triangle that lies outside one face of that pyramid, by a
// margin far above any rounding, fails the exact test for every one of those
// rays, so leaving it out cannot change a result.  The faces are the planes
// through a start point and one edge of the rectangle; g(o, v) = det[v - o,
// La - o, Lb - o] is affine in o and in v, so "all three vertices outside for
// the first and the last start point" implies "the whole triangle outside for
// every start point".  Once per record, lanes run that test over the leaf's
// triangles (one each) and compact the survivors, in order, into LDS; the
// sample loop then walks only those.  Leaves too large for the list are
// screened in segments, once per pass.
#define CAND_MAX 512
#define CULL_MARGIN 0.5f // world units; start points drift by < 0.1 through rounding

#define PASS_SAMPLES 128 // light samples per pass: two per lane

#define MAX_BOUNCES 8 // deepest mirror recursion (rtx_params::max_depth) the kernel folds

// Register budget of shade_kernel: 5 waves per SIMD (96 VGPRs).  Measured on buddha
// 3840x2160 / 500 samples: 4 waves (the compiler's own choice, 110 VGPRs) 5.12 ms,
// 5 waves 4.87 ms, 6 waves 4.86 ms with three times the scratch traffic.  The build
// with mirror bounces keeps the compiler's 3 waves (158 VGPRs) rather than spill.
#ifndef SHADE_WAVES_PER_SIMD
#define SHADE_WAVES_PER_SIMD 5
#endif
#define SHADE_WAVES_PER_SIMD_BOUNCE 3

// A wave takes this many records of a queue per atomic.  Returning atomics on one
// address complete at only a few tens of millions per second, which is the same order
// as the record rate of a frame with few light samples.
#ifndef QUEUE_BATCH
#define QUEUE_BATCH 2u
#endif

// Work classes, in the order the shading kernel takes them.  Records with a leaf of
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
// kernel, four cost more in drained-list checks than the finer order gains.
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

struct ShadeLds {
    float4 level_sum[MAX_BOUNCES + 2];    // per bounce level: sum of its light samples
    float4 level_weight[MAX_BOUNCES + 1]; // specular * Fresnel of the level's hit
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
        const float len = __builtin_sqrtf(f3_dot(m, m));
        float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
        // Both start points must lie on the same side of the light plane: in between, the
        // pyramid turns inside out and "outside for both" says nothing about the others.
        const bool same_side = (zl > o_first.z && zl > o_last.z) || (zl < o_first.z && zl < o_last.z);
        if (same_side && len > 1e-6f && len < 1e30f) {
            m = f3_scale(m, 1.f / len);
            pl = make_float4(m.x, m.y, m.z, f3_dot(m, o));
        }
        L.plane[lane] = pl;
    }
}

// True if triangle (v0, v0 + e1, v0 + e2) may be hit by a shadow ray of the record.
__device__ __forceinline__ bool may_be_hit(const ShadeLds &L, f3 v0, f3 e1, f3 e2, float z_lo, float z_hi) {
    const f3 v1 = f3_add(v0, e1), v2 = f3_add(v0, e2);
    bool out = false;
#pragma unroll 1
    for (int e = 0; e < 4; e++) {
        // outside face e as seen from the first AND from the last start point
        bool both = true;
#pragma unroll
        for (int w = 0; w < 2; w++) {
            const float4 pl = L.plane[e + 4 * w];
            const float d0 = pl.x * v0.x + pl.y * v0.y + pl.z * v0.z - pl.w;
            const float d1 = pl.x * v1.x + pl.y * v1.y + pl.z * v1.z - pl.w;
            const float d2 = pl.x * v2.x + pl.y * v2.y + pl.z * v2.z - pl.w;
            both &= (d0 < -CULL_MARGIN) & (d1 < -CULL_MARGIN) & (d2 < -CULL_MARGIN);
        }
        out |= both;
    }
    // all rays move towards the light plane in z
    out |= (v0.z < z_lo) & (v1.z < z_lo) & (v2.z < z_lo);
    out |= (v0.z > z_hi) & (v1.z > z_hi) & (v2.z > z_hi);
    return !out;
}

// Screens primitives [seg, seg + n) of the leaf (n <= CAND_MAX) and writes the
// survivors' offsets inside the leaf, in order, to L.cand.  Returns how many.
__device__ __forceinline__ uint32_t build_candidates(ShadeLds &L, const DPrim *__restrict__ leaf_prims,
                                                      uint32_t seg, uint32_t n, uint32_t lane, float z_lo,
                                                      float z_hi) {
    uint32_t total = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t t = base + lane;
        bool keep = false;
        if (t < n) {
            const DPrim *p = leaf_prims + seg + t;
            keep = true;
            if ((p->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) keep = may_be_hit(L, p->a, p->b, p->c, z_lo, z_hi);
        }
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            L.cand[pos] = (uint16_t)t;
        }
        total += (uint32_t)__popcll(mask);
    }
    return total;
}

// One hit record, light samples in index order: one wave, lanes over the samples.
// BOUNCE = false is the build for max_depth = 0 (the reference as committed): the
// mirror-bounce code and its registers are compiled out.  IEEE selects the BSDF build.
struct ShadeTotals {
    uint32_t lit = 0, bounce_hits = 0;
    unsigned long long tests = 0, bounce_prims = 0;
};

template <bool BOUNCE, bool IEEE>
__device__ __forceinline__ void shade_record(const DFrame &fr, const DPrim *__restrict__ prims,
                                             const DMat *__restrict__ mats, const DRec *__restrict__ recs,
                                             float4 *__restrict__ colors, ShadeLds &L, uint32_t lane,
                                             uint32_t ridx, ShadeTotals &tot) {
        const float nl = (float)fr.n_lights;
        uint32_t lit_total = 0, bounce_hits = 0;
        unsigned long long bounce_prims = 0, tests_total = 0;
        const DRec rc = recs[ridx];
        const uint32_t pixel = rc.pixel, leaf = rc.leaf, count = rc.count;
        const DPrim *__restrict__ leaf_prims = prims + rc.first;
        // state of the current hit: the primary one first, then one per mirror bounce
        // (raytracing.h:51-61; a bounce is traced inside the same leaf)
        DMat mat = mats[rc.mat];
        f3 nrm = f3_make(rc.nx, rc.ny, rc.nz);
        f3 p0 = f3_make(rc.px, rc.py, rc.pz);
        f3 o, d;
        primary_ray(fr, pixel % fr.width, pixel / fr.width, o, d);
        uint32_t depth = 0;
        for (;;) {
        const f3 view = f3_scale(d, -1.f);      // raytracing.h:47
        const f3 step = f3_scale(nrm, fr.acne); // raytracing.h:40
        const uint32_t k_off = depth * fr.n_lights; // every bounce draws its own light samples
        // shadow-ray origins: closed form where provably exact, else the serial chain run
        // by lanes 0..2 (running origin component and its increment)
        f3 dlt = f3_make(0.f, 0.f, 0.f);
        const bool cx = acne_chain_delta(p0.x, step.x, fr.n_lights, dlt.x);
        const bool cy = acne_chain_delta(p0.y, step.y, fr.n_lights, dlt.y);
        const bool cz = acne_chain_delta(p0.z, step.z, fr.n_lights, dlt.z);
        const bool closed = cx && cy && cz;
        float run = lane == 0 ? p0.x : lane == 1 ? p0.y : p0.z;
        const float inc = lane == 0 ? step.x : lane == 1 ? step.y : step.z;

        // pyramid of the record's shadow rays
        const f3 o_last = f3_add(p0, f3_scale(step, nl + 1.f));
        const float zl = fr.light_pos.z;
        const float o_zmin = fminf(p0.z, o_last.z), o_zmax = fmaxf(p0.z, o_last.z);
        const float z_lo = (zl > o_zmax) ? o_zmin - CULL_MARGIN : -3.0e38f;
        const float z_hi = (zl < o_zmin) ? o_zmax + CULL_MARGIN : 3.0e38f;
        setup_planes(L, lane, p0, o_last, zl);
        wave_lds_sync();
        // a leaf that fits the list is screened once, on the first pass
        const bool one_segment = count <= CAND_MAX;
        uint32_t n_cand = 0;

        f3 acc = f3_make(0.f, 0.f, 0.f);
        uint32_t lit = 0;
        // Each pass takes 128 samples: lane l owns samples base + l and base + 64 + l and
        // carries both rays through the triangle tests as packed pairs.
        for (uint32_t base = 0; base < fr.n_lights; base += PASS_SAMPLES) {
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
            float tmin1 = 0.f, tmin2 = 0.f;
            bool any1 = false, any2 = false;
            for (uint32_t seg = 0; seg < count; seg += CAND_MAX) {
                if (!one_segment || base == 0) {
                    wave_lds_sync();
                    n_cand = build_candidates(L, leaf_prims, seg, min(count - seg, (uint32_t)CAND_MAX), lane,
                                              z_lo, z_hi);
                    wave_lds_sync();
                }
                if (n_cand == 0) continue;
                tests_total += (unsigned long long)n_cand * min(fr.n_lights - base, (uint32_t)PASS_SAMPLES);
                // walk the list with the next primitive's scalar loads in flight
                uint32_t ci = __builtin_amdgcn_readfirstlane((uint32_t)L.cand[0]);
                DPrim cur = leaf_prims[seg + ci];
                for (uint32_t c = 0; c < n_cand; c++) {
                    const uint32_t cn = __builtin_amdgcn_readfirstlane(
                        (uint32_t)L.cand[(c + 1 < n_cand) ? c + 1 : c]);
                    const DPrim nxt = leaf_prims[seg + cn];
                    bool ok1 = live1, ok2 = live2;
                    v2f t = (v2f){0.f, 0.f};
                    if ((cur.type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) {
                        tri_hit2(rr, cur.a, cur.b, cur.c, fr.wide_dets != 0, ok1, ok2, t);
                    } else {
                        float ta = 0.f, tb = 0.f;
                        ok1 = ok1 && sph_hit(p1, sd1, cur.a, cur.b.y, ta);
                        ok2 = ok2 && sph_hit(p2, sd2, cur.a, cur.b.y, tb);
                        t = (v2f){ta, tb};
                    }
                    if (ok1 && (!any1 || t.x < tmin1)) {
                        tmin1 = t.x;
                        any1 = true;
                    }
                    if (ok2 && (!any2 || t.y < tmin2)) {
                        tmin2 = t.y;
                        any2 = true;
                    }
                    cur = nxt;
                }
            }
            // lit unless something is hit inside [0, 1000]^3 (utils.h:12-17); the two
            // samples of a lane are added in sample order
            const bool lit1 = live1 && (!any1 || out_of_scene(f3_add(p1, f3_scale(sd1, tmin1))));
            const bool lit2 = live2 && (!any2 || out_of_scene(f3_add(p2, f3_scale(sd2, tmin2))));
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
        }
        // sum over the lanes (fixed butterfly order, so a frame is reproducible)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acc.x += __shfl_xor(acc.x, off);
            acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off);
            lit += __shfl_xor(lit, off);
        }
        lit_total += lit;
        if (lane == 0) L.level_sum[depth] = make_float4(acc.x, acc.y, acc.z, 0.f);
        if (!BOUNCE || depth >= fr.max_depth) break;

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
            L.level_weight[depth] = make_float4(w.x, w.y, w.z, 0.f);
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
            if (lane == 0) L.level_sum[depth] = make_float4(0.f, 0.f, 0.f, 0.f);
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
        wave_lds_sync();
        if (lane == 0) {
            float4 c = L.level_sum[depth];
            for (uint32_t lv = depth; lv-- > 0;) {
                const float4 sm = L.level_sum[lv], w = L.level_weight[lv];
                c = make_float4(sm.x + w.x * c.x, sm.y + w.y * c.y, sm.z + w.z * c.z, 0.f);
            }
            colors[ridx] = c;
        }
        wave_lds_sync();
        tot.lit += lit_total;
        tot.tests += tests_total;
        tot.bounce_hits += bounce_hits;
        tot.bounce_prims += bounce_prims;
}

// ---- shadow + shade with sorted light samples -------------------------------------
// Second shading kernel, for the hit records shade_kernel leaves to it (large leaf,
// several passes of samples, closed-form origins): the same result, fewer tests.
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
    float sigma;                // +1 light above both start points, -1 below, 0 neither
    float z_lo, z_hi;           // all rays stay within z_lo <= z (or z <= z_hi)
};

__device__ __forceinline__ RayFan make_fan(f3 o_first, f3 o_last, float zl) {
    RayFan f;
    const float o_zmin = fminf(o_first.z, o_last.z), o_zmax = fmaxf(o_first.z, o_last.z);
    f.sigma = (zl > o_zmax) ? 1.f : (zl < o_zmin) ? -1.f : 0.f;
    f.z_lo = (zl > o_zmax) ? o_zmin - CULL_MARGIN : -3.0e38f;
    f.z_hi = (zl < o_zmin) ? o_zmax + CULL_MARGIN : 3.0e38f;
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
        f.thr[w] = CULL_MARGIN * 1.0001f * __builtin_amdgcn_sqrtf(f.h[w] * f.h[w] + far * far);
    }
    return f;
}

// True if triangle (v0, v0 + e1, v0 + e2) lies outside the pyramid of all the record's
// shadow rays: below line 0 or above line 4 in x or in y, or behind the rays in z.
__device__ __forceinline__ bool outside_fan(const RayFan &f, f3 v0, f3 e1, f3 e2) {
    const f3 v[3] = {v0, f3_add(v0, e1), f3_add(v0, e2)};
    bool out = (v[0].z < f.z_lo) & (v[1].z < f.z_lo) & (v[2].z < f.z_lo);
    out |= (v[0].z > f.z_hi) & (v[1].z > f.z_hi) & (v[2].z > f.z_hi);
    if (f.sigma == 0.f) return out;
    bool xb = true, xa = true, yb = true, ya = true;
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const float lo = -f.thr[w], hi = f.thr[w];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float dz = f.sigma * v[j].z - f.soz[w];
            const float t = 500.f * dz;
            const float gx = f.h[w] * (v[j].x - f.ox[w]) - f.gx0[w] * dz;
            const float gy = f.h[w] * (v[j].y - f.oy[w]) - f.gy0[w] * dz;
            xb = xb && gx < lo;
            yb = yb && gy < lo;
            xa = xa && (gx - t) > hi; // line 4: X + 500
            ya = ya && (gy - t) > hi;
        }
    }
    return out || xb || xa || yb || ya;
}

// Cells of the light whose rays may hit a triangle that is not outside_fan.
__device__ __forceinline__ uint32_t cell_mask(const RayFan &f, f3 v0, f3 e1, f3 e2) {
    if (f.sigma == 0.f) return 0xFFFFu;
    const f3 v[3] = {v0, f3_add(v0, e1), f3_add(v0, e2)};
    // xb[i]: the triangle is on the low side of line i (i = 0..3) for every start point;
    // xa[i]: on the high side of line i + 1
    bool xb[4] = {true, true, true, true}, xa[4] = {true, true, true, true};
    bool yb[4] = {true, true, true, true}, ya[4] = {true, true, true, true};
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const float lo = -f.thr[w], hi = f.thr[w];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float dz = f.sigma * v[j].z - f.soz[w];
            const float t = 125.f * dz;
            float gx = f.h[w] * (v[j].x - f.ox[w]) - f.gx0[w] * dz;
            float gy = f.h[w] * (v[j].y - f.oy[w]) - f.gy0[w] * dz;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xb[i] = xb[i] && gx < lo;
                yb[i] = yb[i] && gy < lo;
                gx = gx - t; // next line: X + 125
                gy = gy - t;
                xa[i] = xa[i] && gx > hi;
                ya[i] = ya[i] && gy > hi;
            }
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

// Screens the n <= CAND_MAX primitives of the leaf: offsets inside the leaf of those that
// are not outside_fan, in order, to L.cand.  Returns how many.
__device__ __forceinline__ uint32_t build_fan_candidates(SortLds &L, const RayFan &fan,
                                                          const DPrim *__restrict__ leaf_prims, uint32_t n,
                                                          uint32_t lane) {
    uint32_t total = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t t = base + lane;
        bool keep = false;
        if (t < n) {
            const DPrim *p = leaf_prims + t;
            keep = true;
            if ((p->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) keep = !outside_fan(fan, p->a, p->b, p->c);
        }
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            L.cand[pos] = (uint16_t)t;
        }
        total += (uint32_t)__popcll(mask);
    }
    return total;
}

// Cell masks of the n_cand candidates, to L.cells.
__device__ __forceinline__ void build_cell_masks(SortLds &L, const RayFan &fan,
                                                 const DPrim *__restrict__ leaf_prims, uint32_t n_cand,
                                                 uint32_t lane) {
    for (uint32_t c = lane; c < n_cand; c += 64) {
        const DPrim *p = leaf_prims + L.cand[c];
        uint32_t cells = 0xFFFFu;
        if ((p->type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) cells = cell_mask(fan, p->a, p->b, p->c);
        L.cells[c] = (uint16_t)cells;
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

// One hit record with sorted light samples (see above).  Needs n_lights >= 1, a leaf of at
// most CAND_MAX primitives and the closed form of all three origin components.
template <bool IEEE>
__device__ __forceinline__ void shade_record_sorted(const DFrame &fr, const DPrim *__restrict__ prims,
                                                    const DMat *__restrict__ mats,
                                                    const DRec *__restrict__ recs, float4 *__restrict__ colors,
                                                    SortLds &L, uint32_t lane, uint32_t ridx, ShadeTotals &tot) {
        const float nl = (float)fr.n_lights;
        uint32_t lit_total = 0;
        unsigned long long tests_total = 0;
        const DRec rc = recs[ridx];
        const uint32_t pixel = rc.pixel, leaf = rc.leaf, count = rc.count;
        const DPrim *__restrict__ leaf_prims = prims + rc.first;
        const DMat mat = mats[rc.mat];
        const f3 nrm = f3_make(rc.nx, rc.ny, rc.nz);
        const f3 p0 = f3_make(rc.px, rc.py, rc.pz);
        f3 o, d;
        primary_ray(fr, pixel % fr.width, pixel / fr.width, o, d);
        const f3 view = f3_scale(d, -1.f);      // raytracing.h:47
        const f3 step = f3_scale(nrm, fr.acne); // raytracing.h:40
        f3 dlt = f3_make(0.f, 0.f, 0.f);        // origin of sample k = p0 + (k + 1) * dlt
        acne_chain_delta(p0.x, step.x, fr.n_lights, dlt.x);
        acne_chain_delta(p0.y, step.y, fr.n_lights, dlt.y);
        acne_chain_delta(p0.z, step.z, fr.n_lights, dlt.z);

        uint32_t n_cand;
        bool use_sort;
        {
            const RayFan fan = make_fan(p0, f3_add(p0, f3_scale(step, nl + 1.f)), fr.light_pos.z);
            n_cand = build_fan_candidates(L, fan, leaf_prims, count, lane);
            use_sort = n_cand >= SORT_MIN_CAND;
            wave_lds_sync();
            if (use_sort) build_cell_masks(L, fan, leaf_prims, n_cand, lane);
        }
        wave_lds_sync();

        f3 acc = f3_make(0.f, 0.f, 0.f);
        uint32_t lit = 0;
        for (uint32_t chunk = 0; chunk < fr.n_lights; chunk += SORT_MAX) {
            const uint32_t n_chunk = min(fr.n_lights - chunk, (uint32_t)SORT_MAX);
            if (use_sort) sort_samples(L, lane, fr.seed, pixel, leaf, chunk, n_chunk);
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
                    light_sample_u(fr.seed, pixel, leaf, chunk + s1, ux, uy);
                    w1 = ux | (uy << 9);
                    light_sample_u(fr.seed, pixel, leaf, chunk + s2, ux, uy);
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
                float tmin1 = 0.f, tmin2 = 0.f;
                bool any1 = false, any2 = false;
                if (n_list) {
                    tests_total += (unsigned long long)n_list * min(n_chunk - base, (uint32_t)PASS_SAMPLES);
                    uint32_t ci = __builtin_amdgcn_readfirstlane((uint32_t)list[0]);
                    DPrim cur = leaf_prims[ci];
                    for (uint32_t c = 0; c < n_list; c++) {
                        const uint32_t cn = __builtin_amdgcn_readfirstlane(
                            (uint32_t)list[(c + 1 < n_list) ? c + 1 : c]);
                        const DPrim nxt = leaf_prims[cn];
                        bool ok1 = live1, ok2 = live2;
                        v2f t = (v2f){0.f, 0.f};
                        if ((cur.type_mat & 0xFFu) == RTX_PRIM_TRIANGLE) {
                            tri_hit2(rr, cur.a, cur.b, cur.c, fr.wide_dets != 0, ok1, ok2, t);
                        } else {
                            float ta = 0.f, tb = 0.f;
                            ok1 = ok1 && sph_hit(p1, sd1, cur.a, cur.b.y, ta);
                            ok2 = ok2 && sph_hit(p2, sd2, cur.a, cur.b.y, tb);
                            t = (v2f){ta, tb};
                        }
                        if (ok1 && (!any1 || t.x < tmin1)) {
                            tmin1 = t.x;
                            any1 = true;
                        }
                        if (ok2 && (!any2 || t.y < tmin2)) {
                            tmin2 = t.y;
                            any2 = true;
                        }
                        cur = nxt;
                    }
                }
                // lit unless something is hit inside [0, 1000]^3 (utils.h:12-17)
                const bool lit1 = live1 && (!any1 || out_of_scene(f3_add(p1, f3_scale(sd1, tmin1))));
                const bool lit2 = live2 && (!any2 || out_of_scene(f3_add(p2, f3_scale(sd2, tmin2))));
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
        lit_total += lit;
        if (lane == 0) colors[ridx] = make_float4(acc.x, acc.y, acc.z, 0.f);
        tot.lit += lit_total;
        tot.tests += tests_total;
}

// ---- work lists ------------------------------------------------------------------
// The record goes to the list of its class.  A wave classifies CLASSIFY_PER_LANE x 64
// records and then reserves their entries with one atomic per class (returning atomics
// on one address are slow: one per 64 records would cost more than the rest of the kernel).
#ifndef CLASSIFY_PER_LANE
#define CLASSIFY_PER_LANE 8
#endif
__global__ __launch_bounds__(256) void classify_kernel(DFrame fr, const DRec *__restrict__ recs,
                                                        uint32_t rec_cap, float4 *__restrict__ colors,
                                                        DCounters *__restrict__ ctr) {
    const RecMap map = rec_map(ctr, rec_cap); // final: written by the previous kernel
    const uint32_t n_recs = map.total;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t per_wave = 64 * CLASSIFY_PER_LANE;
    for (uint32_t base = wave_id * per_wave; base < n_recs; base += n_waves * per_wave) {
        uint32_t cls[CLASSIFY_PER_LANE];
#pragma unroll
        for (int g = 0; g < CLASSIFY_PER_LANE; g++) {
            const uint32_t i = base + g * 64 + lane;
            cls[g] = N_CLASSES; // none
            if (i < n_recs) {
                const DRec *rc = recs + rec_index(map, i);
                const uint32_t count = rc->count;
                bool sorted = false;
                if (fr.sort_samples && count >= SORT_MIN_LEAF && count <= CAND_MAX) {
                    float dlt;
                    sorted = acne_chain_delta(rc->px, rc->nx * fr.acne, fr.n_lights, dlt) &&
                             acne_chain_delta(rc->py, rc->ny * fr.acne, fr.n_lights, dlt) &&
                             acne_chain_delta(rc->pz, rc->nz * fr.acne, fr.n_lights, dlt);
                }
                cls[g] = (sorted ? CLASS_SORTED : CLASS_PLAIN) + weight_group(count);
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
// classify_kernel has sorted the hit records into N_CLASSES work lists, heaviest class
// first: records shaded with sorted samples (large leaf, then smaller), then the others
// (large leaf, then small).  Every wave drains the lists in that order, so that the
// long records start early and the tail of the kernel is made of short ones.  A list is
// handed out through eight heads (entry e belongs to head e & 7): a wave starts on the
// head of its block (blocks b and b + 8 tend to share an XCD, hence an L2) and moves on
// when it is drained; heads only ever advance, so a wave that has seen all eight
// drained is done with the list.
template <bool SORTED, bool BOUNCE, bool IEEE, typename Lds>
__device__ __forceinline__ void drain_list(uint32_t cls, const DFrame &fr, const DPrim *__restrict__ prims,
                                           const DMat *__restrict__ mats, const DRec *__restrict__ recs,
                                           uint32_t rec_cap, float4 *__restrict__ colors,
                                           DCounters *__restrict__ ctr, Lds &L, uint32_t lane,
                                           ShadeTotals &tot) {
    const uint32_t n = ctr->cls[cls].count; // final: written by classify_kernel
    if (n == 0) return;
    const uint32_t *__restrict__ list = work_list(colors, rec_cap, cls);
    const uint32_t batch = (cls == CLASS_LIGHTEST) ? QUEUE_BATCH : 1u;
    uint32_t q = blockIdx.x & 7;
    for (;;) {
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(&ctr->head[cls][q].next, batch);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k * 8 + q >= n) {
            // This head is drained.  Look at all eight (plain loads: a failing atomic
            // per head and wave would add up) and go on with one that is not, if any.
            uint32_t next = 0xFFFFFFFFu;
            if (lane < 8)
                next = __hip_atomic_load(&ctr->head[cls][lane].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t open = (uint32_t)__ballot(lane < 8 && next < n && next * 8 + lane < n);
            if (open == 0) break; // heads only advance: nothing will ever be left for this wave
            const uint32_t ahead = open & ~((2u << q) - 1u); // prefer the next one upwards
            q = (uint32_t)__ffs((int)(ahead ? ahead : open)) - 1u;
            continue;
        }
        for (uint32_t kb = k; kb < k + batch; kb++) {
            const uint32_t e = kb * 8 + q;
            if (e >= n) break;
            const uint32_t ridx = list[e];
            if constexpr (SORTED)
                shade_record_sorted<IEEE>(fr, prims, mats, recs, colors, L, lane, ridx, tot);
            else
                shade_record<BOUNCE, IEEE>(fr, prims, mats, recs, colors, L, lane, ridx, tot);
        }
    }
}

template <bool BOUNCE, bool IEEE>
__global__ __launch_bounds__(SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(BOUNCE ? SHADE_WAVES_PER_SIMD_BOUNCE : SHADE_WAVES_PER_SIMD,
                                                   BOUNCE ? SHADE_WAVES_PER_SIMD_BOUNCE : SHADE_WAVES_PER_SIMD)))
void shade_kernel(
    DFrame fr, const DPrim *__restrict__ prims, const DMat *__restrict__ mats,
    const DRec *__restrict__ recs, uint32_t rec_cap, float4 *__restrict__ colors,
    DCounters *__restrict__ ctr) {
    __shared__ union {
        ShadeLds plain;
        SortLds sorted;
    } lds[SHADE_WAVES];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ShadeTotals tot;
    if constexpr (!BOUNCE) {
        for (uint32_t cls = CLASS_SORTED; cls < CLASS_SORTED + N_WEIGHTS; cls++)
            drain_list<true, false, IEEE>(cls, fr, prims, mats, recs, rec_cap, colors, ctr, lds[wave].sorted, lane,
                                          tot);
    }
    for (uint32_t cls = CLASS_PLAIN; cls < CLASS_PLAIN + N_WEIGHTS; cls++)
        drain_list<false, BOUNCE, IEEE>(cls, fr, prims, mats, recs, rec_cap, colors, ctr, lds[wave].plain, lane,
                                        tot);
    if (lane == 0 && tot.lit) atomicAdd(&ctr->lit, (unsigned long long)tot.lit);
    if (lane == 0 && tot.tests) atomicAdd(&ctr->tests_run, tot.tests);
    if (lane == 0 && tot.bounce_hits) {
        atomicAdd(&ctr->bounce_hits, (unsigned long long)tot.bounce_hits);
        atomicAdd(&ctr->bounce_prims, tot.bounce_prims);
    }
}

// ---- shadow + shade for few light samples ---------------------------------------
// With n_lights <= G <= 32 (the reference as committed uses 1) a wave per hit record
// would leave most lanes idle, so here a wave takes 64 / G records at once: lane
// l handles sample l % G of record l / G, walks that record's leaf on its own (vector
// loads, next primitive in flight) and the G lanes of a record are summed with the
// same butterfly the one-record kernel uses, so both kernels return identical colours.
// No mirror bounces (max_depth = 0 only).
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
    if (lane == 0 && lit_total) atomicAdd(&ctr->lit, (unsigned long long)lit_total);
    if (lane == 0 && tests_total) atomicAdd(&ctr->tests_run, tests_total);
}

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
    const uint32_t *__restrict__ first_rec, uint8_t *__restrict__ rgb) {
    extern __shared__ float merge_stack[]; // (deepest nesting + 1) * 3 * MERGE_BLOCK floats
    const uint32_t local = blockIdx.x * MERGE_BLOCK + threadIdx.x;
    if (local >= fr.rows * fr.width) return;
    float rx = 0.f, ry = 0.f, rz = 0.f;
    uint32_t r = first_rec[local];
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

// ---- self test ------------------------------------------------------------------
// rcp2_exact against the plain division for every binary32 value in its range.
__global__ void selftest_rcp_kernel(unsigned long long *bad, unsigned long long *checked) {
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t nb = 0, nc = 0;
    // 2^32 bit patterns, two per iteration
    for (unsigned long long i = (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 2; i < (1ull << 32);
         i += (unsigned long long)stride * 2) {
        const float x = __uint_as_float((uint32_t)i), y = __uint_as_float((uint32_t)i + 1u);
        const v2f r = rcp2_exact((v2f){x, y});
        const bool inx = __builtin_fabsf(x) >= RCP2_MIN && __builtin_fabsf(x) < RCP2_MAX;
        const bool iny = __builtin_fabsf(y) >= RCP2_MIN && __builtin_fabsf(y) < RCP2_MAX;
        if (inx) {
            nc++;
            nb += __float_as_uint(r.x) != __float_as_uint(1.f / x);
        }
        if (iny) {
            nc++;
            nb += __float_as_uint(r.y) != __float_as_uint(1.f / y);
        }
    }
    if (nb) atomicAdd(bad, (unsigned long long)nb);
    atomicAdd(checked, (unsigned long long)nc);
}

// div2_exact / normalize2 against the plain quotient: pseudo-random denominators over
// DIV2_DEN_MIN..DIV2_DEN_MAX (log-uniform) and numerators from DIV2_NUM_MIN up to 4 d,
// of both signs, plus vectors through normalize2 against f3_normalize.
__global__ void selftest_div_kernel(unsigned long long *bad, unsigned long long *checked) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nb = 0, nc = 0;
    for (uint32_t it = 0; it < 4096; it++) {
        uint32_t u[6];
        for (int k = 0; k < 3; k++) light_sample_u(0x51ED270Bu + k, tid, it & 0xFFFFu, it >> 16, u[2 * k], u[2 * k + 1]);
        uint32_t w[4];
        { // four more words: the raw Philox output of another counter
            uint32_t c0 = tid, c1 = it ^ 0xA5A5A5A5u, key = 0x243F6A88u;
            for (int r = 0; r < 10; r++) {
                unsigned long long prod = (unsigned long long)0xD256D193u * c0;
                uint32_t hi = (uint32_t)(prod >> 32), lo = (uint32_t)prod;
                c0 = hi ^ key ^ c1;
                c1 = lo;
                key += 0x9E3779B9u;
            }
            w[0] = c0; w[1] = c1; w[2] = c0 * 0x9E3779B9u + c1; w[3] = c1 * 0x85EBCA6Bu + c0;
        }
        // denominators 2^e * m, e in -39..39; numerators = d * ratio, ratio 2^-f * m', f in 0..20 (or up to 4)
        v2f dn, nm;
        for (int k = 0; k < 2; k++) {
            const int e = (int)(w[k] % 79u) - 39;
            const float m = __uint_as_float(0x3F800000u | (w[k] >> 9));
            const float dv = ldexpf(m, e);
            const int f = (int)(w[k + 2] % 23u) - 2;
            const float m2 = __uint_as_float(0x3F800000u | (w[k + 2] >> 9));
            float nv = ldexpf(m2, -f) * dv * 0.999f;
            if (w[k + 2] & 0x100u) nv = -nv;
            if (k == 0) { dn.x = dv; nm.x = nv; } else { dn.y = dv; nm.y = nv; }
        }
        const bool ok0 = __builtin_fabsf(nm.x) >= DIV2_NUM_MIN && __builtin_fabsf(nm.x) <= 4.f * dn.x;
        const bool ok1 = __builtin_fabsf(nm.y) >= DIV2_NUM_MIN && __builtin_fabsf(nm.y) <= 4.f * dn.y;
        const v2f q = div2_exact(nm, recip2_make(dn));
        if (ok0) { nc++; nb += __float_as_uint(q.x) != __float_as_uint(nm.x / dn.x); }
        if (ok1) { nc++; nb += __float_as_uint(q.y) != __float_as_uint(nm.y / dn.y); }
        // two vectors with components -500..500 (some exactly zero), scaled by 2^-12..2^3
        f3 a = f3_make((float)u[0] - 250.f, (float)u[1] - 250.f, (float)u[2] - 250.f);
        f3 b = f3_make((float)u[3] - 250.f, (float)u[4] - 250.f, (float)u[5] - 250.f);
        a = f3_scale(a, ldexpf(1.f + (float)(w[0] & 1023u) * 0x1p-10f, (int)(w[1] & 15u) - 12));
        b = f3_scale(b, ldexpf(1.f + (float)(w[2] & 1023u) * 0x1p-10f, (int)(w[3] & 15u) - 12));
        v2f x = (v2f){a.x, b.x}, y = (v2f){a.y, b.y}, z = (v2f){a.z, b.z};
        normalize2(x, y, z);
        const f3 na = f3_normalize(a), nb3 = f3_normalize(b);
        nc += 2;
        nb += !(__float_as_uint(x.x) == __float_as_uint(na.x) && __float_as_uint(y.x) == __float_as_uint(na.y) &&
                __float_as_uint(z.x) == __float_as_uint(na.z));
        nb += !(__float_as_uint(x.y) == __float_as_uint(nb3.x) && __float_as_uint(y.y) == __float_as_uint(nb3.y) &&
                __float_as_uint(z.y) == __float_as_uint(nb3.z));
    }
    if (nb) atomicAdd(bad, (unsigned long long)nb);
    atomicAdd(checked, (unsigned long long)nc);
}

// acne_chain_delta against the serial chain on pseudo-random and adversarial inputs
// (values next to powers of two, increments with few significant bits, both signs).
__global__ void selftest_chain_kernel(unsigned long long *bad, unsigned long long *checked,
                                      unsigned long long *refused) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t w0 = id, w1 = 0x9E3779B9u, key = 0xC0FFEEu;
    for (int r = 0; r < 10; r++) {
        const unsigned long long prod = (unsigned long long)0xD256D193u * w0;
        w0 = (uint32_t)(prod >> 32) ^ key ^ w1;
        w1 = (uint32_t)prod;
        key += 0x9E3779B9u;
    }
    // p0: exponent 2^-12 .. 2^13, mantissa random or within a few ulps of a binade edge
    const int e = (int)(w0 % 26u) - 12;
    uint32_t man = (w0 >> 5) & 0x7FFFFFu;
    const uint32_t mode = (w1 >> 3) & 7u;
    if (mode == 0) man = (w0 >> 9) & 0x3Fu;              // just above 2^e
    if (mode == 1) man = 0x7FFFFFu - ((w0 >> 9) & 0x3Fu); // just below 2^(e+1)
    float p0 = __uint_as_float(((uint32_t)(e + 127) << 23) | man);
    if (w1 & 1u) p0 = -p0;
    // c: |c| 2^-24 .. 2^-1 times a random significand, sometimes cut to a few bits
    uint32_t cm = (w1 >> 8) & 0x7FFFFFu;
    if (mode >= 5) cm &= 0x7FFFFFu << (12 + (mode - 5) * 5);
    float c = __uint_as_float(((uint32_t)(127 - 1 - (int)((w0 >> 28) + ((w1 >> 6) & 15u))) << 23) | cm);
    if (w1 & 2u) c = -c;
    const uint32_t n = 1u + (w0 >> 11) % 1000u;
    float delta = 0.f;
    const bool ok = acne_chain_delta(p0, c, n, delta);
    uint32_t nb = 0;
    if (ok) {
        float run = p0;
        for (uint32_t k = 1; k <= n; k++) {
            run = run + c;
            const float cf = p0 + (float)k * delta;
            nb += __float_as_uint(run) != __float_as_uint(cf);
        }
    }
    if (nb) atomicAdd(bad, 1ull);
    atomicAdd(ok ? checked : refused, 1ull);
}

// ===========================================================================
// Host side: scene upload and the C ABI
// ===========================================================================

static thread_local std::string g_err;

static rtx_status fail(rtx_status s, const std::string &msg) {
    g_err = msg;
    return s;
}

#define HIP_TRY(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(RTX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

struct rtx_scene {
    int device = 0;
    int n_cu = 256;
    // scene (immutable after creation)
    DNode *d_nodes = nullptr;
    DUp *d_up = nullptr;
    DPrim *d_prims = nullptr;
    DMat *d_mats = nullptr;
    uint32_t n_nodes = 0, n_prims = 0, n_mats = 0;
    uint32_t nest_levels = 1; // deepest nesting of a box + 1
    uint32_t wide_dets = 0;   // some triangle's |e1| |e2| is not below the bound of rcp2_exact's range
    rtx_light light{};
    // per-frame work buffers, grown on demand
    DRec *d_recs = nullptr;
    uint64_t n_hits = 0; // hit records of the last frame
    float4 *d_colors = nullptr;
    uint32_t rec_cap = 0;
    uint32_t *d_first = nullptr;
    uint64_t pix_cap = 0;
    uint8_t *d_rgb = nullptr; // staging for rtx_render's host output
    uint64_t rgb_cap = 0;
    DCounters *d_ctr = nullptr;
    DCounters *h_ctr = nullptr; // pinned
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

extern "C" uint32_t rtx_abi_version(void) { return RTX_ABI_VERSION; }

extern "C" int rtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *rtx_last_error(void) { return g_err.c_str(); }

extern "C" rtx_status rtx_selftest(int device, uint64_t out[7]) {
    if (!out) return fail(RTX_ERR_INVALID, "null argument");
    if (device < 0 || device >= rtx_device_count()) return fail(RTX_ERR_NO_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 56));
    hipError_t e = hipMemset(d, 0, 56);
    if (e == hipSuccess) {
        selftest_rcp_kernel<<<4096, 256>>>(d + 1, d);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) {
            selftest_chain_kernel<<<16384, 256>>>(d + 3, d + 2, d + 4);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) {
            selftest_div_kernel<<<1024, 256>>>(d + 6, d + 5);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    unsigned long long h[7] = {0, 0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, 56, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RTX_ERR_HIP, std::string("selftest: ") + hipGetErrorString(e));
    for (int k = 0; k < 7; k++) out[k] = h[k];
    return RTX_OK;
}

extern "C" void rtx_params_default(rtx_params *p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->width = 1000;  // raytracing.h:12
    p->height = 1000; // raytracing.h:13
    p->n_lights = 1;  // raytracing.h:19
    p->max_depth = 0; // raytracing.h:18
    p->seed = 1;
    p->acne = 0.01f; // raytracing.h:21
    p->camera[0] = 500.f; // raytracing.h:112
    p->camera[1] = 500.f;
    p->camera[2] = 3000.f;
}

extern "C" uint32_t rtx_part_rows(const rtx_params *p) {
    if (!p) return 0;
    if (p->n_parts <= 1 || p->stripe_rows == 0) return p->height;
    // stripes s = part, part + n_parts, ...; the last stripe may be cut short
    uint64_t rows = 0;
    for (uint64_t s = p->part; s * p->stripe_rows < p->height; s += p->n_parts) {
        uint64_t lo = s * p->stripe_rows, hi = std::min<uint64_t>(lo + p->stripe_rows, p->height);
        rows += hi - lo;
    }
    return (uint32_t)rows;
}

static f3 host_f3(const float *p) {
    f3 r;
    r.x = p[0];
    r.y = p[1];
    r.z = p[2];
    return r;
}

extern "C" void rtx_scene_destroy(rtx_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipFree(s->d_nodes);
    (void)hipFree(s->d_up);
    (void)hipFree(s->d_prims);
    (void)hipFree(s->d_mats);
    (void)hipFree(s->d_recs);
    (void)hipFree(s->d_colors);
    (void)hipFree(s->d_first);
    (void)hipFree(s->d_rgb);
    (void)hipFree(s->d_ctr);
    if (s->h_ctr) (void)hipHostFree(s->h_ctr);
    for (auto &e : s->ev)
        if (e) (void)hipEventDestroy(e);
    delete s;
}

extern "C" rtx_status rtx_scene_create(const rtx_scene_desc *desc, int device, rtx_scene **out) {
    if (!desc || !out) return fail(RTX_ERR_INVALID, "null argument");
    *out = nullptr;
    const uint32_t N = desc->n_nodes;
    if ((N && !desc->nodes) || (desc->n_leaves && !desc->leaves) || (desc->n_prims && !desc->prims) ||
        (desc->n_materials && !desc->materials))
        return fail(RTX_ERR_INVALID, "null array with non-zero count");
    if (desc->n_leaves > RTX_MAX_LEAVES) return fail(RTX_ERR_LIMIT, "too many leaves (limit 65535)");
    if (desc->n_materials > RTX_MAX_MATERIALS) return fail(RTX_ERR_LIMIT, "too many materials");

    // ---- validate the pre-order array, find parents and depths ----
    std::vector<uint32_t> parent(N, RTX_NONE), depth(N, 0);
    {
        std::vector<uint32_t> open; // indices of the inner boxes that enclose the current one
        std::vector<uint8_t> leaf_seen(desc->n_leaves, 0);
        for (uint32_t n = 0; n < N; n++) {
            while (!open.empty() && n >= desc->nodes[open.back()].skip) open.pop_back();
            const rtx_node &nd = desc->nodes[n];
            const uint32_t limit = open.empty() ? N : desc->nodes[open.back()].skip;
            if (nd.skip <= n || nd.skip > limit) return fail(RTX_ERR_INVALID, "node skip index out of order");
            parent[n] = open.empty() ? RTX_NONE : open.back();
            depth[n] = (uint32_t)open.size();
            if (depth[n] >= MAX_NEST) return fail(RTX_ERR_LIMIT, "boxes nested too deep (limit 23)");
            if (nd.leaf == RTX_NONE) {
                open.push_back(n);
            } else {
                if (nd.leaf >= desc->n_leaves || nd.skip != n + 1 || leaf_seen[nd.leaf])
                    return fail(RTX_ERR_INVALID, "bad leaf reference in node");
                leaf_seen[nd.leaf] = 1;
                const rtx_leaf &lf = desc->leaves[nd.leaf];
                if ((uint64_t)lf.first + lf.count > desc->n_prims)
                    return fail(RTX_ERR_INVALID, "leaf range exceeds the primitive array");
                if (lf.count >= 0xFFFFu) return fail(RTX_ERR_LIMIT, "leaf with more than 65534 primitives");
            }
        }
    }
    for (uint32_t k = 0; k < desc->n_prims; k++) {
        const rtx_prim &p = desc->prims[k];
        if (p.type != RTX_PRIM_SPHERE && p.type != RTX_PRIM_TRIANGLE)
            return fail(RTX_ERR_INVALID, "unknown primitive type");
        if (p.material >= desc->n_materials) return fail(RTX_ERR_INVALID, "material index out of range");
    }

    // ---- prune boxes whose subtree holds no primitive ----
    std::vector<uint32_t> weight(N + 1, 0); // primitives in leaves with index < n
    for (uint32_t n = 0; n < N; n++) {
        const rtx_node &nd = desc->nodes[n];
        weight[n + 1] = weight[n] + (nd.leaf == RTX_NONE ? 0u : desc->leaves[nd.leaf].count);
    }
    std::vector<uint32_t> newidx(N + 1, 0);
    std::vector<uint8_t> keep(N, 0);
    for (uint32_t n = 0; n < N; n++) {
        keep[n] = weight[desc->nodes[n].skip] > weight[n];
        newidx[n + 1] = newidx[n] + keep[n];
    }
    const uint32_t NK = newidx[N];
    std::vector<DNode> nodes(NK);
    std::vector<DUp> ups(NK);
    uint32_t nest_levels = 1; // deepest nesting of a kept box + 1 (sizes the merge kernel's stack)
    for (uint32_t n = 0; n < N; n++) {
        if (!keep[n]) continue;
        const rtx_node &nd = desc->nodes[n];
        DNode &o = nodes[newidx[n]];
        memcpy(o.lb, nd.lb, sizeof(o.lb));
        memcpy(o.rt, nd.rt, sizeof(o.rt));
        if (nd.leaf == RTX_NONE) {
            o.link = newidx[nd.skip];
            o.info = RTXD_NONE;
        } else {
            o.link = desc->leaves[nd.leaf].first;
            o.info = (desc->leaves[nd.leaf].count << 16) | nd.leaf;
        }
        ups[newidx[n]].parent = parent[n] == RTX_NONE ? RTXD_NONE : newidx[parent[n]];
        ups[newidx[n]].depth = depth[n];
        nest_levels = std::max(nest_levels, depth[n] + 1);
    }

    // ---- primitives: ray-independent values precomputed in the reference's
    // operation order (host x86-64, no contraction) ----
    // A ray direction has unit length, so |e1 . (d x e2)| <= |e1| |e2| up to rounding: below
    // DET_BOUND no determinant can reach RCP2_MAX and the intersection kernels skip that check.
    const double DET_BOUND = 1.0e17;
    uint32_t wide_dets = 0;
    std::vector<DPrim> prims(desc->n_prims);
    for (uint32_t k = 0; k < desc->n_prims; k++) {
        const rtx_prim &p = desc->prims[k];
        DPrim &o = prims[k];
        memset(&o, 0, sizeof(o));
        o.type_mat = p.type | (p.material << 8);
        o.id = p.id;
        if (p.type == RTX_PRIM_TRIANGLE) {
            const float *v0 = p.g, *v1 = p.g + 3, *v2 = p.g + 6;
            volatile float e1x = v1[0] - v0[0], e1y = v1[1] - v0[1], e1z = v1[2] - v0[2];
            volatile float e2x = v2[0] - v0[0], e2y = v2[1] - v0[1], e2z = v2[2] - v0[2];
            // cross (vec3.h:34-41)
            volatile float ax = e1y * e2z, bx = e1z * e2y;
            volatile float ay = e1z * e2x, by = e1x * e2z;
            volatile float az = e1x * e2y, bz = e1y * e2x;
            volatile float cx = ax - bx, cy = ay - by, cz = az - bz;
            // norm (vec3.h:10-25)
            volatile float xx = cx * cx, yy = cy * cy, zz = cz * cz;
            volatile float s1 = xx + yy;
            volatile float s2 = s1 + zz;
            const float len = sqrtf(s2);
            o.a = host_f3(v0);
            o.b.x = e1x, o.b.y = e1y, o.b.z = e1z;
            o.c.x = e2x, o.c.y = e2y, o.c.z = e2z;
            o.n.x = cx / len, o.n.y = cy / len, o.n.z = cz / len;
            const double l1 = sqrt((double)e1x * e1x + (double)e1y * e1y + (double)e1z * e1z);
            const double l2 = sqrt((double)e2x * e2x + (double)e2y * e2y + (double)e2z * e2z);
            if (!(l1 * l2 < DET_BOUND)) wide_dets = 1;
        } else {
            o.a = host_f3(p.g);
            volatile float r2 = p.g[3] * p.g[3];
            o.b.x = p.g[3];
            o.b.y = r2;
        }
    }
    std::vector<DMat> mats(std::max<uint32_t>(desc->n_materials, 1));
    const float pi = atanf(1.f) * 4.f; // shade.h:3
    for (uint32_t k = 0; k < desc->n_materials; k++) {
        const rtx_material &m = desc->materials[k];
        DMat &o = mats[k];
        memset(&o, 0, sizeof(o));
        o.ior = m.ior;
        o.roughness = m.roughness;
        volatile float a2 = m.roughness * m.roughness;
        volatile float pa2 = pi * a2;
        volatile float ratio = 1.f / m.ior;
        volatile float r2 = ratio * ratio;
        o.alpha2 = a2;
        o.pi_alpha2 = pa2;
        o.ratio2 = r2;
        o.spec = host_f3(m.specular);
        o.dif_pi.x = m.diffuse[0] / pi;
        o.dif_pi.y = m.diffuse[1] / pi;
        o.dif_pi.z = m.diffuse[2] / pi;
        o.exp_scale = -1.44269504088896341f / a2;
        o.inv_pi_alpha2 = 1.f / pa2;
        o.inv_alpha = 1.f / m.roughness;
    }

    // ---- upload ----
    int ndev = rtx_device_count();
    if (ndev <= 0) return fail(RTX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(RTX_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RTX_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950");

    rtx_scene *s = new rtx_scene();
    s->device = device;
    s->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    s->n_nodes = NK;
    s->nest_levels = nest_levels;
    s->wide_dets = wide_dets;
    s->n_prims = desc->n_prims;
    s->n_mats = desc->n_materials;
    s->light = desc->light;
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 64);
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up((void **)&s->d_nodes, nodes.data(), nodes.size() * sizeof(DNode));
    if (e == hipSuccess) e = up((void **)&s->d_up, ups.data(), ups.size() * sizeof(DUp));
    if (e == hipSuccess) e = up((void **)&s->d_prims, prims.data(), prims.size() * sizeof(DPrim));
    if (e == hipSuccess) e = up((void **)&s->d_mats, mats.data(), mats.size() * sizeof(DMat));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_ctr, sizeof(DCounters));
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_ctr, sizeof(DCounters), hipHostMallocDefault);
    for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipEventCreate(&s->ev[k]);
    if (e != hipSuccess) {
        rtx_scene_destroy(s);
        return fail(e == hipErrorOutOfMemory ? RTX_ERR_NOMEM : RTX_ERR_HIP,
                    std::string("scene upload: ") + hipGetErrorString(e));
    }
    *out = s;
    return RTX_OK;
}

static rtx_status check_params(const rtx_scene *s, const rtx_params *p) {
    if (!s || !p) return fail(RTX_ERR_INVALID, "null argument");
    if (p->width == 0 || p->height == 0) return fail(RTX_ERR_INVALID, "empty frame");
    if ((uint64_t)p->width * p->height > 0xFFFFFFFFull) return fail(RTX_ERR_LIMIT, "more than 2^32 pixels");
    if (p->max_depth > MAX_BOUNCES) return fail(RTX_ERR_LIMIT, "max_depth above 8");
    if ((uint64_t)p->n_lights * (p->max_depth + 1ull) > RTX_MAX_LIGHTS)
        return fail(RTX_ERR_LIMIT, "more than 65535 light samples per leaf hit (n_lights * (max_depth + 1))");
    if (p->n_parts > 1 && (p->stripe_rows == 0 || p->part >= p->n_parts))
        return fail(RTX_ERR_INVALID, "bad row tiling");
    return RTX_OK;
}

template <typename T>
static hipError_t grow(T **ptr, uint64_t *cap, uint64_t need) {
    if (need <= *cap) return hipSuccess;
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void **)ptr, need * sizeof(T));
    if (e == hipSuccess) *cap = need;
    return e;
}

static rtx_status ensure_records(rtx_scene *s, uint64_t need) {
    if (need <= s->rec_cap) return RTX_OK;
    if (need > 0xFFFFFFF0ull) return fail(RTX_ERR_LIMIT, "more than 2^32 hit records in one call");
    (void)hipFree(s->d_recs);
    (void)hipFree(s->d_colors);
    s->d_recs = nullptr;
    s->d_colors = nullptr;
    s->rec_cap = 0;
    hipError_t e = hipMalloc((void **)&s->d_recs, need * sizeof(DRec));
    // the colours are followed by the work lists of the shading kernel
    if (e == hipSuccess)
        e = hipMalloc((void **)&s->d_colors, need * (sizeof(float4) + N_CLASSES * sizeof(uint32_t)));
    if (e != hipSuccess) return fail(RTX_ERR_NOMEM, std::string("record buffers: ") + hipGetErrorString(e));
    s->rec_cap = (uint32_t)need;
    return RTX_OK;
}

// Enqueue and complete one frame.  On return s->h_ctr holds the frame's counters.
static rtx_status render_frame(rtx_scene *s, const rtx_params *p, uint8_t *d_rgb, hipStream_t st,
                               rtx_stats *stats) {
    rtx_status rs = check_params(s, p);
    if (rs != RTX_OK) return rs;
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t rows = rtx_part_rows(p);
    if (stats) memset(stats, 0, sizeof(*stats));
    if (rows == 0) return RTX_OK;
    const uint64_t npix = (uint64_t)rows * p->width;

    DFrame fr;
    memset(&fr, 0, sizeof(fr));
    fr.width = p->width;
    fr.height = p->height;
    fr.rows = rows;
    fr.n_parts = (p->n_parts > 1) ? p->n_parts : 1;
    fr.stripe_rows = (p->n_parts > 1) ? p->stripe_rows : p->height;
    fr.part = (p->n_parts > 1) ? p->part : 0;
    fr.n_lights = p->n_lights;
    fr.max_depth = p->max_depth;
    fr.wide_dets = s->wide_dets;
    fr.sort_samples = (p->max_depth == 0 && p->n_lights >= SORT_MIN_LIGHTS && !(p->flags & RTX_FLAG_NO_SORT)) ? 1u : 0u;
    fr.seed = p->seed;
    fr.acne = p->acne;
    fr.camera = host_f3(p->camera);
    fr.light_pos = host_f3(s->light.position);
    fr.light_color = host_f3(s->light.color);
    fr.scale = 1000.f / (float)p->height;
    fr.half_width = (float)p->width * 0.5f;

    if (hipError_t e = grow(&s->d_first, &s->pix_cap, npix); e != hipSuccess)
        return fail(RTX_ERR_NOMEM, std::string("pixel buffer: ") + hipGetErrorString(e));
    if (s->rec_cap == 0) {
        rs = ensure_records(s, std::max<uint64_t>(npix / 2, 1u << 16));
        if (rs != RTX_OK) return rs;
    }
    const bool timing = stats && (p->flags & RTX_FLAG_TIMING);
    const uint32_t tiles = ((p->width + 15) / 16) * ((rows + 15) / 16);
    // the queue-driven shading kernels get exactly the waves that are resident at once
    // (a block is one wave per SIMD); the few-samples kernel strides over the records
    const uint32_t few_blocks = (uint32_t)s->n_cu * 8;
    const uint32_t trace_blocks = (uint32_t)s->n_cu * TRACE_BLOCKS_PER_CU;
    const uint32_t shade_blocks = (uint32_t)s->n_cu * (p->max_depth ? SHADE_WAVES_PER_SIMD_BOUNCE : SHADE_WAVES_PER_SIMD);
    const uint32_t merge_blocks = (uint32_t)((npix + MERGE_BLOCK - 1) / MERGE_BLOCK);

    uint64_t n_hits = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        HIP_TRY(hipMemsetAsync(s->d_ctr, 0, sizeof(DCounters), st));
        if (timing) HIP_TRY(hipEventRecord(s->ev[0], st));
        trace_kernel<<<std::min<uint32_t>(tiles, trace_blocks), TRACE_BLOCK, 0, st>>>(fr, s->d_nodes, s->n_nodes, s->d_prims, s->d_recs,
                                                    s->rec_cap, s->d_first, s->d_ctr);
#define LAUNCH_FEW(G)                                                                                        \
    do {                                                                                                     \
        if (ieee)                                                                                            \
            shade_few_kernel<G, true><<<few_blocks, SHADE_BLOCK, 0, st>>>(                                 \
                fr, s->d_prims, s->d_mats, s->d_recs, s->rec_cap, s->d_colors, s->d_ctr);                    \
        else                                                                                                 \
            shade_few_kernel<G, false><<<few_blocks, SHADE_BLOCK, 0, st>>>(                                \
                fr, s->d_prims, s->d_mats, s->d_recs, s->rec_cap, s->d_colors, s->d_ctr);                    \
    } while (0)
#define LAUNCH_SHADE(B, I)                                                                                   \
    shade_kernel<B, I><<<shade_blocks, SHADE_BLOCK, 0, st>>>(fr, s->d_prims, s->d_mats, s->d_recs,          \
                                                             s->rec_cap, s->d_colors, s->d_ctr)
        const uint32_t nlg = p->n_lights;
        const bool ieee = (p->flags & RTX_FLAG_IEEE_SHADE) != 0;
        const bool few = p->max_depth == 0 && nlg <= 32 && !(p->flags & RTX_FLAG_WAVE_PER_HIT);
        if (!few)
            classify_kernel<<<few_blocks, 256, 0, st>>>(fr, s->d_recs, s->rec_cap, s->d_colors, s->d_ctr);
        if (timing) HIP_TRY(hipEventRecord(s->ev[1], st)); // so that [1] is the shading kernel alone
        if (few) {
            // few samples: several hit records per wave
            if (nlg <= 1) LAUNCH_FEW(1);
            else if (nlg <= 2) LAUNCH_FEW(2);
            else if (nlg <= 4) LAUNCH_FEW(4);
            else if (nlg <= 8) LAUNCH_FEW(8);
            else if (nlg <= 16) LAUNCH_FEW(16);
            else LAUNCH_FEW(32);
        } else if (p->max_depth == 0) {
            if (ieee) LAUNCH_SHADE(false, true);
            else LAUNCH_SHADE(false, false);
        } else {
            if (ieee) LAUNCH_SHADE(true, true);
            else LAUNCH_SHADE(true, false);
        }
        if (timing) HIP_TRY(hipEventRecord(s->ev[2], st));
        merge_kernel<<<merge_blocks, MERGE_BLOCK, (size_t)s->nest_levels * 3 * MERGE_BLOCK * sizeof(float), st>>>(fr, s->d_nodes, s->d_up, s->d_recs,
                                                          s->d_colors, s->d_first, d_rgb);
        if (timing) HIP_TRY(hipEventRecord(s->ev[3], st));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(s->h_ctr, s->d_ctr, sizeof(DCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        uint32_t fullest = 0;
        n_hits = 0;
        for (int k = 0; k < REC_SHARDS; k++) {
            fullest = std::max(fullest, s->h_ctr->shard[k].count);
            n_hits += s->h_ctr->shard[k].count;
        }
        if (fullest <= s->rec_cap / REC_SHARDS) break;
        // a shard got more hits than it holds: grow the record buffer and render again
        if (attempt == 2) return fail(RTX_ERR_HIP, "record buffer overflow persisted");
        rs = ensure_records(s, ((uint64_t)fullest + fullest / 8 + 128) * REC_SHARDS);
        if (rs != RTX_OK) return rs;
    }
    s->n_hits = n_hits;
    if (stats) {
        stats->primary_rays = npix;
        stats->hits = n_hits;
        stats->shadow_rays = (n_hits + s->h_ctr->bounce_hits) * p->n_lights;
        stats->shadow_tests = ((uint64_t)s->h_ctr->leaf_prims + s->h_ctr->bounce_prims) * p->n_lights;
        stats->lit_samples = s->h_ctr->lit;
        stats->shadow_tests_run = s->h_ctr->tests_run;
        if (timing) {
            for (int k = 0; k < 3; k++) HIP_TRY(hipEventElapsedTime(&stats->kernel_ms[k], s->ev[k], s->ev[k + 1]));
            HIP_TRY(hipEventElapsedTime(&stats->kernel_ms[3], s->ev[0], s->ev[3]));
        }
    }
    return RTX_OK;
}

extern "C" rtx_status rtx_render_device(rtx_scene *s, const rtx_params *p, void *d_rgb, void *hip_stream,
                                        rtx_stats *stats) {
    if (!d_rgb) return fail(RTX_ERR_INVALID, "null output");
    return render_frame(s, p, (uint8_t *)d_rgb, (hipStream_t)hip_stream, stats);
}

static rtx_status render_to_staging(rtx_scene *s, const rtx_params *p, rtx_stats *stats) {
    rtx_status rs = check_params(s, p);
    if (rs != RTX_OK) return rs;
    HIP_TRY(hipSetDevice(s->device));
    const uint64_t bytes = (uint64_t)rtx_part_rows(p) * p->width * 3;
    if (hipError_t e = grow(&s->d_rgb, &s->rgb_cap, std::max<uint64_t>(bytes, 64)); e != hipSuccess)
        return fail(RTX_ERR_NOMEM, std::string("output staging: ") + hipGetErrorString(e));
    return render_frame(s, p, s->d_rgb, nullptr, stats);
}

extern "C" rtx_status rtx_render(rtx_scene *s, const rtx_params *p, uint8_t *rgb_host, rtx_stats *stats) {
    if (!rgb_host) return fail(RTX_ERR_INVALID, "null output");
    rtx_status rs = render_to_staging(s, p, stats);
    if (rs != RTX_OK) return rs;
    const uint64_t bytes = (uint64_t)rtx_part_rows(p) * p->width * 3;
    if (bytes) HIP_TRY(hipMemcpy(rgb_host, s->d_rgb, bytes, hipMemcpyDeviceToHost));
    return RTX_OK;
}

extern "C" rtx_status rtx_render_hits(rtx_scene *s, const rtx_params *p, rtx_hit *hits, uint64_t capacity,
                                      uint64_t *n_hits, uint8_t *rgb_host) {
    if (!n_hits) return fail(RTX_ERR_INVALID, "null n_hits");
    rtx_status rs = render_to_staging(s, p, nullptr);
    if (rs != RTX_OK) return rs;
    const uint64_t bytes = (uint64_t)rtx_part_rows(p) * p->width * 3;
    if (rgb_host && bytes) HIP_TRY(hipMemcpy(rgb_host, s->d_rgb, bytes, hipMemcpyDeviceToHost));
    const uint64_t n = rtx_part_rows(p) ? s->n_hits : 0;
    *n_hits = n;
    if (!hits || n == 0) return RTX_OK;
    std::vector<DRec> recs(n);
    std::vector<float4> cols(n);
    const uint64_t shard_cap = s->rec_cap / REC_SHARDS;
    for (uint64_t k = 0, at = 0; k < REC_SHARDS; k++) { // the shards of the record array, one after the other
        const uint64_t m = s->h_ctr->shard[k].count;
        if (m == 0) continue;
        HIP_TRY(hipMemcpy(recs.data() + at, s->d_recs + k * shard_cap, m * sizeof(DRec), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(cols.data() + at, s->d_colors + k * shard_cap, m * sizeof(float4), hipMemcpyDeviceToHost));
        at += m;
    }
    std::vector<uint32_t> order(n);
    for (uint64_t k = 0; k < n; k++) order[k] = (uint32_t)k;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if (recs[a].pixel != recs[b].pixel) return recs[a].pixel < recs[b].pixel;
        return recs[a].ordinal < recs[b].ordinal;
    });
    for (uint64_t k = 0; k < n && k < capacity; k++) {
        const DRec &r = recs[order[k]];
        const float4 &c = cols[order[k]];
        rtx_hit &h = hits[k];
        h.pixel = r.pixel;
        h.ordinal = r.ordinal;
        h.leaf = r.leaf;
        h.prim = r.prim;
        h.t = r.t;
        h.color[0] = c.x;
        h.color[1] = c.y;
        h.color[2] = c.z;
    }
    return RTX_OK;
}
