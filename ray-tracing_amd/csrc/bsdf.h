// The BSDF of shade.h: a build that follows it operation by operation and a fast one.
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

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
