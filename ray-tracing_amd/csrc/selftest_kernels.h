// Device-side checks of the exact arithmetic shortcuts (rtx_selftest).
// Part of the single translation unit rtx_amd.hip (included there, in order).
#pragma once

// ---- self test ------------------------------------------------------------------
// rcp2_exact against the plain division and sqrt2_exact against sqrtf, for every binary32
// value in their ranges; the threshold on the reciprocal that stands for intersect.h:39's
// test of the determinant, for every binary32 value.
// every = 1 checks all 2^32 bit patterns; every = n checks one pair of neighbouring patterns in
// n (evenly spaced, so every exponent and sign is visited): the sampled check that scene
// creation runs.
__global__ void selftest_rcp_kernel(unsigned long long *bad, unsigned long long *checked,
                                    unsigned long long *bad_sqrt, unsigned long long *checked_sqrt, uint32_t every) {
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t nb = 0, nc = 0, sb = 0, sc = 0;
    // 2^32 bit patterns, two per iteration
    const unsigned long long pairs = (1ull << 31) / every;
    for (unsigned long long pr = blockIdx.x * blockDim.x + threadIdx.x; pr < pairs; pr += stride) {
        // (the sampled pairs also move through the low mantissa bits)
        const unsigned long long i = (pr * every + (pr * 0x9E3779B1ull >> 7) % every) * 2;
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
        // the determinant test of the triangle intersection, made on the reciprocal: every
        // finite value, zero and subnormals included (an infinite determinant makes f zero
        // and the distance zero or NaN, not a hit either way)
        if (__builtin_fabsf(x) < __builtin_inff()) {
            nc++;
            nb += (__builtin_fabsf(x) < 0.0000001f) != !(__builtin_fabsf(r.x) <= 10000000.f);
        }
        if (__builtin_fabsf(y) < __builtin_inff()) {
            nc++;
            nb += (__builtin_fabsf(y) < 0.0000001f) != !(__builtin_fabsf(r.y) <= 10000000.f);
        }
        const v2f q = sqrt2_exact((v2f){x, y});
        if (x >= SQRT2_MIN && x <= SQRT2_MAX) {
            sc++;
            sb += __float_as_uint(q.x) != __float_as_uint(__builtin_sqrtf(x));
        }
        if (y >= SQRT2_MIN && y <= SQRT2_MAX) {
            sc++;
            sb += __float_as_uint(q.y) != __float_as_uint(__builtin_sqrtf(y));
        }
    }
    if (nb) atomicAdd(bad, (unsigned long long)nb);
    atomicAdd(checked, (unsigned long long)nc);
    if (sb) atomicAdd(bad_sqrt, (unsigned long long)sb);
    atomicAdd(checked_sqrt, (unsigned long long)sc);
}

// div2_exact / normalize2 against the plain quotient: pseudo-random denominators over
// DIV2_DEN_MIN..DIV2_DEN_MAX (log-uniform) and numerators from DIV2_NUM_MIN up to 4 d,
// of both signs, plus vectors through normalize2 against f3_normalize.
__global__ void selftest_div_kernel(unsigned long long *bad, unsigned long long *checked, uint32_t iterations) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t nb = 0, nc = 0;
    for (uint32_t it = 0; it < iterations; it++) {
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
