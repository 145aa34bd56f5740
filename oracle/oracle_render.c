/*
 * oracle_render.c - test-only CPU restatement of the reference's render path
 * (raytracing.h:29-145, intersect.h:1-132, shade.h:1-90, utils.h:1-17,
 * vec3.h), binary32 in the reference's operation order, with a counter-based
 * light sampler in place of std::random_device.  Build with
 * -ffp-contract=off and without -ffast-math.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "rtx_oracle.h"

#define ORC_MAX_NEST 64 /* deepest box nesting the walk accepts */

typedef struct {
    float x, y, z;
} v3;

/* ---- vec3.h ------------------------------------------------------------- */
static inline v3 v3_add(v3 a, v3 b) { return (v3){a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline v3 v3_sub(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline v3 v3_mul(v3 a, v3 b) { return (v3){a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline v3 v3_scale(v3 a, float s) { return (v3){a.x * s, a.y * s, a.z * s}; }
static inline v3 v3_div(v3 a, float s) { return (v3){a.x / s, a.y / s, a.z / s}; }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_normalize(v3 a) { return v3_div(a, sqrtf(v3_dot(a, a))); }
static inline v3 v3_cross(v3 u, v3 v) {
    return (v3){u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
/* vec3.h:97-101: "less" means all three components strictly less */
static inline int v3_all_less(v3 a, v3 b) { return a.x < b.x && a.y < b.y && a.z < b.z; }
static inline float std_min(float a, float b) { return (b < a) ? b : a; }
static inline float std_max(float a, float b) { return (a < b) ? b : a; }
static inline v3 ld3(const float *p) { return (v3){p[0], p[1], p[2]}; }
static inline void st3(float *p, v3 a) {
    p[0] = a.x;
    p[1] = a.y;
    p[2] = a.z;
}

/* ---- sampler ------------------------------------------------------------- */
void orc_philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t out[2]) {
    for (int r = 0; r < 10; r++) {
        uint64_t prod = (uint64_t)0xD256D193u * c0;
        uint32_t hi = (uint32_t)(prod >> 32), lo = (uint32_t)prod;
        c0 = hi ^ key ^ c1;
        c1 = lo;
        key += 0x9E3779B9u;
    }
    out[0] = c0;
    out[1] = c1;
}

void orc_light_sample(uint32_t seed, uint32_t pixel, uint32_t leaf, uint32_t k, float xy[2]) {
    uint32_t w[2];
    orc_philox2x32_10(pixel, (leaf << 16) | k, seed, w);
    uint32_t ux = (uint32_t)(((uint64_t)w[0] * 501u) >> 32);
    uint32_t uy = (uint32_t)(((uint64_t)w[1] * 501u) >> 32);
    xy[0] = 250.f + (float)ux; /* raytracing.h:38 */
    xy[1] = 250.f + (float)uy; /* raytracing.h:39 */
}

/* ---- intersect.h --------------------------------------------------------- */
typedef struct {
    float t;
    v3 pos, nrm;
} isect;

/* intersect.h:1-26 */
static inline int hit_sphere(v3 o, v3 d, const float *g, isect *out) {
    v3 c = ld3(g);
    float radius = g[3];
    v3 pos = v3_sub(o, c);
    float a = 1.f;
    float b = 2.f * v3_dot(d, pos);
    float cc = v3_dot(pos, pos) - radius * radius;
    float delta = b * b - 4.f * a * cc;
    if (delta <= 0.f) return 0;
    float t1 = (-b - sqrtf(delta)) / (2.f * a);
    float t2 = (-b + sqrtf(delta)) / (2.f * a);
    float t = t1 > 0.f ? t1 : t2 > 0.f ? t2 : -1.f;
    if (t > 0.f) {
        out->t = t;
        out->pos = v3_add(o, v3_scale(d, t));
        out->nrm = v3_normalize(v3_sub(out->pos, c));
        return 1;
    }
    return 0;
}

/* intersect.h:28-67 */
static inline int hit_triangle(v3 o, v3 d, const float *g, isect *out) {
    const float epsilon = 0.0000001f;
    v3 v0 = ld3(g), v1 = ld3(g + 3), v2 = ld3(g + 6);
    v3 edge1 = v3_sub(v1, v0);
    v3 edge2 = v3_sub(v2, v0);
    v3 h = v3_cross(d, edge2);
    float a = v3_dot(edge1, h);
    if (a > -epsilon && a < epsilon) return 0;
    float f = 1.f / a;
    v3 s = v3_sub(o, v0);
    float u = f * v3_dot(s, h);
    if (u < 0.f || u > 1.f) return 0;
    v3 q = v3_cross(s, edge1);
    float v = f * v3_dot(d, q);
    if (v < 0.f || u + v > 1.f) return 0;
    float t = f * v3_dot(edge2, q);
    if (t > epsilon) {
        out->t = t;
        out->pos = v3_add(o, v3_scale(d, t));
        out->nrm = v3_normalize(v3_cross(edge1, edge2));
        return 1;
    }
    return 0;
}

/* intersect.h:69-93 */
static inline int hit_box(v3 o, v3 d, const float *lb, const float *rt) {
    float fx = 1.f / d.x, fy = 1.f / d.y, fz = 1.f / d.z;
    float t1 = (lb[0] - o.x) * fx;
    float t2 = (rt[0] - o.x) * fx;
    float t3 = (lb[1] - o.y) * fy;
    float t4 = (rt[1] - o.y) * fy;
    float t5 = (lb[2] - o.z) * fz;
    float t6 = (rt[2] - o.z) * fz;
    float tmin = std_max(std_max(std_min(t1, t2), std_min(t3, t4)), std_min(t5, t6));
    float tmax = std_min(std_min(std_max(t1, t2), std_max(t3, t4)), std_max(t5, t6));
    if (tmax < 0.f) return 0;
    if (tmin > tmax) return 0;
    return 1;
}

/* intersect.h:95-132: every primitive of the leaf is tested; the smallest
 * distance wins, the earliest primitive on ties.  Returns the index of the
 * winning reference inside the leaf, or -1. */
static inline int nearest_in_leaf(v3 o, v3 d, const rtx_prim *pr, uint32_t n, isect *best,
                                  uint64_t *tests) {
    int win = -1;
    isect cur;
    for (uint32_t k = 0; k < n; k++) {
        int ok = (pr[k].type == RTX_PRIM_TRIANGLE) ? hit_triangle(o, d, pr[k].g, &cur)
                 : (pr[k].type == RTX_PRIM_SPHERE) ? hit_sphere(o, d, pr[k].g, &cur)
                                                   : 0;
        if (ok && (win < 0 || cur.t < best->t)) {
            *best = cur;
            win = (int)k;
        }
    }
    *tests += n;
    return win;
}

/* ---- shade.h -------------------------------------------------------------- */
static float g_pi; /* shade.h:3 */

static inline float rdm_beckmann(float NdotH, float alpha) {
    if (NdotH > 0.f) {
        float cos2 = NdotH * NdotH;
        float tan2 = (1.f - cos2) / cos2;
        float cos4 = cos2 * cos2;
        float alpha2 = alpha * alpha;
        return expf(-tan2 / alpha2) / (g_pi * alpha2 * cos4);
    }
    return 0.f;
}

static inline float rdm_fresnel(float LdotH, float extIOR, float intIOR) {
    float cosI = LdotH;
    float ratioIOR = extIOR / intIOR;
    float sinT2 = ratioIOR * ratioIOR * (1.f - (cosI * cosI));
    if (sinT2 > 1.f) return 1.f;
    float cosT = sqrtf(1.f - sinT2);
    float Rs = powf(cosI * extIOR - cosT * intIOR, 2) / powf(cosI * extIOR + cosT * intIOR, 2);
    float Rp = powf(cosT * extIOR - cosI * intIOR, 2) / powf(cosT * extIOR + cosI * intIOR, 2);
    return 0.5f * (Rs + Rp);
}

static inline float rdm_g1(float DdotH, float DdotN, float alpha) {
    float tn = sqrtf(1.f - DdotN * DdotN) / DdotN;
    float b = 1.f / (alpha * tn);
    float k = DdotH / DdotN;
    if (k > 0.f) {
        if (b < 1.6f)
            return (3.535f * b + 2.181f * b * b) / (1.f + 2.276f * b + 2.577f * b * b);
        return 1.f;
    }
    return 0.f;
}

/* shade.h:80-90, with RDM_bsdf / _s / _d (shade.h:58-78) inlined.  The fourth
 * argument of RDM_bsdf is LdotH, not LdotN (shade.h:88). */
static inline v3 shade(v3 n, v3 v, v3 l, v3 lc, const rtx_material *m) {
    v3 h = v3_normalize(v3_add(v, l));
    float LdotH = v3_dot(l, h);
    float VdotH = v3_dot(v, h);
    float NdotH = v3_dot(n, h);
    float LdotN = v3_dot(l, n);
    float VdotN = v3_dot(v, n);
    float bs_LdotN = LdotH;
    float dd = rdm_beckmann(NdotH, m->roughness);
    float ff = rdm_fresnel(LdotH, 1.f, m->ior);
    float gg = rdm_g1(LdotH, bs_LdotN, m->roughness) * rdm_g1(VdotH, VdotN, m->roughness);
    v3 spec = ld3(m->specular);
    v3 s = v3_div(v3_scale(v3_scale(v3_scale(spec, dd), ff), gg), 4.f * bs_LdotN * VdotN);
    v3 dif = v3_div(ld3(m->diffuse), g_pi);
    v3 bsdf = v3_add(dif, s);
    return v3_scale(v3_mul(lc, bsdf), LdotN);
}

/* ---- utils.h:3-17 ---------------------------------------------------------- */
uint8_t orc_clamp_color(float v) {
    if (v > 255.f)
        v = 255.f;
    else if (v < 0.f)
        v = 0.f;
    if (v != v) return 0; /* (unsigned char)NaN is undefined in C; counted in stats */
    return (uint8_t)v;
}

static inline int out_of_scene(v3 c) {
    return c.x < 0.f || c.x > 1000.f || c.y < 0.f || c.y > 1000.f || c.z < 0.f || c.z > 1000.f;
}

/* ---- frame ------------------------------------------------------------------ */
static void ensure_pi(void) { g_pi = atanf(1.f) * 4.f; }

/* raytracing.h:119-121 for a width x height frame.  World units are the pixels
 * of a 1000-row image: s = 1000 / height; x is centred on 500.  At 1000 x 1000
 * this is exactly (i, j, 1000). */
void orc_primary_ray(const rtx_params *p, uint32_t i, uint32_t j, float o[3], float d[3]) {
    float s = 1000.f / (float)p->height;
    v3 org = {500.f + ((float)i - (float)p->width * 0.5f) * s, (float)j * s, 1000.f};
    v3 cam = ld3(p->camera);
    v3 dir = v3_normalize(v3_sub(org, cam));
    st3(o, org);
    st3(d, dir);
}

static uint32_t part_rows(const rtx_params *p) {
    if (p->n_parts <= 1 || p->stripe_rows == 0) return p->height;
    uint32_t rows = 0;
    for (uint32_t j = 0; j < p->height; j++)
        if ((j / p->stripe_rows) % p->n_parts == p->part) rows++;
    return rows;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

typedef struct {
    rtx_hit *v;
    uint64_t n, cap;
} hitvec;

static void hv_push(hitvec *h, const rtx_hit *r) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 256;
        h->v = (rtx_hit *)realloc(h->v, h->cap * sizeof(rtx_hit));
    }
    h->v[h->n++] = *r;
}

/* raytracing.h:29-72 for one leaf.  `depth` is Ray::depth: 0 for the primary ray, +1
 * per mirror bounce (raytracing.h:51-61; the bounce is traced inside the SAME leaf box).
 * The light sampler is indexed by depth * n_lights + k so that every bounce draws its
 * own samples.  Returns 1 and the colour when the ray hits something in the leaf. */
static int trace_leaf(const rtx_scene_desc *sc, const rtx_params *p, uint32_t pixel,
                      uint32_t leaf_id, v3 o, v3 d, uint32_t depth, v3 *color, rtx_hit *rec,
                      orc_stats *st) {
    const rtx_leaf *lf = &sc->leaves[leaf_id];
    const rtx_prim *pr = sc->prims + lf->first;
    isect hit;
    int win = nearest_in_leaf(o, d, pr, lf->count, &hit,
                              depth ? &st->bounce_prim_tests : &st->primary_prim_tests);
    if (win < 0) return 0;
    const rtx_material *mat = &sc->materials[pr[win].material];
    v3 lc = ld3(sc->light.color);
    v3 lpos = ld3(sc->light.position);
    v3 col = {0.f, 0.f, 0.f};
    v3 pos = hit.pos;
    float nl = (float)p->n_lights;
    /* ORC_FLAG_WAVE_SUM: 64 partial sums, sample k to sum k % 64, combined pairwise at the end */
    const int wave_sum = (p->flags & ORC_FLAG_WAVE_SUM) != 0;
    v3 part[64];
    if (wave_sum) memset(part, 0, sizeof(part));
    for (uint32_t k = 0; k < p->n_lights; k++) {
        float xy[2];
        orc_light_sample(p->seed, pixel, leaf_id, depth * p->n_lights + k, xy);
        lpos.x = xy[0];
        lpos.y = xy[1];
        pos = v3_add(pos, v3_scale(hit.nrm, p->acne)); /* accumulates, raytracing.h:40 */
        v3 sd = v3_normalize(v3_sub(lpos, pos));
        isect sh = {0.f, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        int occ = nearest_in_leaf(pos, sd, pr, lf->count, &sh, &st->shadow_prim_tests);
        st->shadow_rays++;
        if (occ < 0 || out_of_scene(sh.pos)) {
            v3 inv = v3_scale(d, -1.f);
            v3 c = shade(hit.nrm, inv, sd, lc, mat);
            if (wave_sum)
                part[k & 63u] = v3_add(part[k & 63u], v3_scale(v3_div(c, nl), 20.f));
            else
                col = v3_add(col, v3_scale(v3_div(c, nl), 20.f));
            st->shaded_samples++;
        }
    }
    if (wave_sum) {
        for (uint32_t off = 32; off >= 1; off >>= 1) {
            v3 next[64];
            for (uint32_t l = 0; l < 64; l++) next[l] = v3_add(part[l], part[l ^ off]);
            memcpy(part, next, sizeof(part));
        }
        col = part[0];
    }
    v3 ret = col;
    if (depth < p->max_depth) {
        /* raytracing.h:53-60; reflect() is vec3.h:103-106: I - N * 2.f * dot(N, I) */
        v3 n2 = v3_scale(hit.nrm, 2.f);
        v3 rd = v3_normalize(v3_sub(d, v3_scale(n2, v3_dot(hit.nrm, d))));
        v3 ro = v3_add(pos, v3_scale(rd, p->acne)); /* pos: after all the acne steps */
        v3 cr = {0.f, 0.f, 0.f};
        st->bounce_rays++;
        trace_leaf(sc, p, pixel, leaf_id, ro, rd, depth + 1, &cr, NULL, st);
        float fres = rdm_fresnel(v3_dot(rd, hit.nrm), 1.f, mat->ior);
        ret = v3_add(col, v3_mul(v3_scale(ld3(mat->specular), fres), cr));
    }
    *color = ret;
    if (rec) {
        rec->pixel = pixel;
        rec->leaf = leaf_id;
        rec->prim = pr[win].id;
        rec->t = hit.t;
        st3(rec->color, ret);
    }
    return 1;
}

int orc_render(const rtx_scene_desc *sc, const rtx_params *p, uint8_t *rgb, float *fcolor,
               rtx_hit *hits, uint64_t hit_cap, uint64_t *n_hits, orc_stats *stats, int threads) {
    if (!sc || !p || p->width == 0 || p->height == 0) return 1;
    if ((uint64_t)p->n_lights * ((uint64_t)p->max_depth + 1) > RTX_MAX_LIGHTS ||
        sc->n_leaves > RTX_MAX_LEAVES)
        return 1;
    {
        /* nesting depth of the pre-order array */
        uint32_t end[ORC_MAX_NEST];
        int sp = 0;
        end[0] = sc->n_nodes;
        for (uint32_t n = 0; n < sc->n_nodes; n++) {
            while (sp > 0 && n >= end[sp]) sp--;
            if (sc->nodes[n].skip <= n || sc->nodes[n].skip > end[sp]) return 1;
            if (sc->nodes[n].leaf == RTX_NONE) {
                if (++sp >= ORC_MAX_NEST) return 1;
                end[sp] = sc->nodes[n].skip;
            } else if (sc->nodes[n].leaf >= sc->n_leaves || sc->nodes[n].skip != n + 1) {
                return 1;
            }
        }
    }
    ensure_pi();
    uint32_t W = p->width, H = p->height;
    uint32_t rows = part_rows(p);
    uint32_t *row_j = (uint32_t *)malloc((rows ? rows : 1) * sizeof(uint32_t));
    {
        uint32_t r = 0;
        for (uint32_t j = 0; j < H; j++)
            if (p->n_parts <= 1 || p->stripe_rows == 0 ||
                (j / p->stripe_rows) % p->n_parts == p->part)
                row_j[r++] = j;
    }
    hitvec *row_hits = (hitvec *)calloc(rows ? rows : 1, sizeof(hitvec));
    orc_stats total;
    memset(&total, 0, sizeof(total));
    int nthreads = threads > 0 ? threads : orc_max_threads();
    (void)nthreads;
    int want_hits = (hits != NULL) || (n_hits != NULL);

#pragma omp parallel num_threads(nthreads)
    {
        orc_stats st;
        memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 1)
        for (uint32_t r = 0; r < rows; r++) {
            uint32_t j = row_j[r];
            for (uint32_t i = 0; i < W; i++) {
                uint32_t pixel = j * W + i;
                float of[3], df[3];
                orc_primary_ray(p, i, j, of, df);
                v3 o = ld3(of), d = ld3(df);
                st.primary_rays++;
                /* raytracing.h:74-98 on the pre-order array.  Every call of
                 * trace_ray_boxs folds its own list of boxes into its own
                 * colour, starting from black, and hands the result to its
                 * caller, which folds it in turn: the fold is per nesting
                 * level, not one running colour over all leaves ("<" on
                 * colours is a partial order, so the two differ).  acc[sp]
                 * is the colour of the call that is open at nesting level
                 * sp; end[sp] is where its list of boxes ends. */
                v3 acc[ORC_MAX_NEST];
                uint32_t end[ORC_MAX_NEST];
                int sp = 0;
                acc[0] = (v3){0.f, 0.f, 0.f};
                end[0] = sc->n_nodes;
                uint32_t ordinal = 0;
                uint32_t n = 0;
                for (;;) {
                    while (sp > 0 && n >= end[sp]) {
                        if (v3_all_less(acc[sp - 1], acc[sp])) acc[sp - 1] = acc[sp];
                        sp--;
                    }
                    if (n >= sc->n_nodes) break;
                    const rtx_node *nd = &sc->nodes[n];
                    st.slab_tests++;
                    if (!hit_box(o, d, nd->lb, nd->rt)) {
                        n = nd->skip;
                        continue;
                    }
                    if (nd->leaf == RTX_NONE) {
                        sp++;
                        acc[sp] = (v3){0.f, 0.f, 0.f};
                        end[sp] = nd->skip;
                        n = n + 1;
                        continue;
                    }
                    st.leaf_visits++;
                    v3 nc;
                    rtx_hit rec;
                    if (trace_leaf(sc, p, pixel, nd->leaf, o, d, 0, &nc, &rec, &st)) {
                        st.hits++;
                        rec.ordinal = ordinal++;
                        if (want_hits) hv_push(&row_hits[r], &rec);
                        if (v3_all_less(acc[sp], nc)) acc[sp] = nc;
                    }
                    n = nd->skip;
                }
                v3 color = acc[0];
                size_t px = (size_t)r * W + i;
                if (fcolor) st3(fcolor + 3 * px, color);
                if (color.x != color.x) st.nan_channels++;
                if (color.y != color.y) st.nan_channels++;
                if (color.z != color.z) st.nan_channels++;
                if (rgb) {
                    rgb[3 * px + 0] = orc_clamp_color(color.x);
                    rgb[3 * px + 1] = orc_clamp_color(color.y);
                    rgb[3 * px + 2] = orc_clamp_color(color.z);
                }
            }
        }
#pragma omp critical
        {
            total.primary_rays += st.primary_rays;
            total.slab_tests += st.slab_tests;
            total.leaf_visits += st.leaf_visits;
            total.primary_prim_tests += st.primary_prim_tests;
            total.hits += st.hits;
            total.shadow_rays += st.shadow_rays;
            total.shadow_prim_tests += st.shadow_prim_tests;
            total.shaded_samples += st.shaded_samples;
            total.nan_channels += st.nan_channels;
            total.bounce_rays += st.bounce_rays;
            total.bounce_prim_tests += st.bounce_prim_tests;
        }
    }
    uint64_t nh = 0;
    for (uint32_t r = 0; r < rows; r++) {
        for (uint64_t k = 0; k < row_hits[r].n; k++) {
            if (hits && nh < hit_cap) hits[nh] = row_hits[r].v[k];
            nh++;
        }
        free(row_hits[r].v);
    }
    free(row_hits);
    free(row_j);
    if (n_hits) *n_hits = nh;
    if (stats) *stats = total;
    return 0;
}

/* ---- function-level entry points --------------------------------------------- */
int orc_intersect_sphere(const float o[3], const float d[3], const float c[3], float r, float *t,
                         float pos[3], float nrm[3]) {
    float g[9] = {c[0], c[1], c[2], r, 0, 0, 0, 0, 0};
    isect h;
    if (!hit_sphere(ld3(o), ld3(d), g, &h)) return 0;
    *t = h.t;
    st3(pos, h.pos);
    st3(nrm, h.nrm);
    return 1;
}

int orc_intersect_triangle(const float o[3], const float d[3], const float v0[3],
                           const float v1[3], const float v2[3], float *t, float pos[3],
                           float nrm[3]) {
    float g[9] = {v0[0], v0[1], v0[2], v1[0], v1[1], v1[2], v2[0], v2[1], v2[2]};
    isect h;
    if (!hit_triangle(ld3(o), ld3(d), g, &h)) return 0;
    *t = h.t;
    st3(pos, h.pos);
    st3(nrm, h.nrm);
    return 1;
}

int orc_intersect_box(const float o[3], const float d[3], const float lb[3], const float rt[3]) {
    return hit_box(ld3(o), ld3(d), lb, rt);
}

void orc_shade(const float n[3], const float v[3], const float l[3], const float lc[3],
               const rtx_material *m, float out[3]) {
    ensure_pi();
    st3(out, shade(ld3(n), ld3(v), ld3(l), ld3(lc), m));
}

int orc_coord_out_of_scene(const float c[3]) { return out_of_scene(ld3(c)); }
