/*
 * rtx_oracle.h - CPU restatement of the reference's render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing that ships (the HIP library, the host
 * drop-in layer) may include, link or call this; it exists so that tests,
 * the smoke check and the CPU-baseline leg of the benchmark have something
 * to compare the GPU against.
 *
 * It restates, in plain C and with the reference's exact binary32 operation
 * order, raytracing.h:29-145, intersect.h:1-132, shade.h:1-90, bvh.h:5-61,
 * meshs.h:1-125, utils.h:1-45 and vec3.h of romain-roy/ray-tracing, with
 * three additions the reference lacks: run-time width / height / sample
 * count / box depth, primitive ids, and a counter-based light sampler
 * (Philox-2x32-10) in place of std::random_device (raytracing.h:25-27).
 *
 * Parity pinning: the restatement is checked bit for bit against the
 * reference's own functions (compiled from /root/reference by
 * oracle/Makefile into oracle/_ref/) through the fixtures in tests/golden/.
 * The reference has no tests or golden vectors of its own.
 *
 * The scene exchange format is the product's plain-data rtx_scene_desc
 * (include/rtx.h) so that both sides can be fed identical inputs.
 */
#ifndef RTX_ORACLE_H
#define RTX_ORACLE_H

#include <stdint.h>
#include "../include/rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- light sampler -------------------------------------------------------
 * Philox-2x32-10 (Salmon et al., SC'11), key = seed, counter =
 * (pixel, leaf << 16 | sample).  Word 0 -> x, word 1 -> y, each mapped to the
 * integer grid 0..500 with (w * 501) >> 32 (uniform_int_distribution(0, 500),
 * raytracing.h:27), then 250.f + (float)U (raytracing.h:38-39). */
void orc_philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t out[2]);
void orc_light_sample(uint32_t seed, uint32_t pixel, uint32_t leaf, uint32_t k, float xy[2]);

/* ---- scene building (meshs.h, bvh.h, scene.h) ---------------------------- */
typedef struct orc_builder orc_builder;

orc_builder *orc_builder_new(uint32_t box_depth /* DEPTH_BOX, bvh.h:3 */);
void orc_builder_free(orc_builder *b);
void orc_set_light(orc_builder *b, const rtx_light *l);
uint32_t orc_add_material(orc_builder *b, const rtx_material *m);
/* A top-level box that is never split (depth = DEPTH_BOX, scene.h:49-54).
 * prims[k].id is ignored and assigned by the builder. */
int orc_add_leaf_box(orc_builder *b, const float lb[3], const float rt[3],
                     const rtx_prim *prims, uint32_t n);
/* create_mesh (meshs.h:1-56): normalise, place, emit triangles into a new
 * top-level box of depth 0. */
int orc_add_mesh(orc_builder *b, const float *verts, uint32_t nv, const uint32_t *faces,
                 uint32_t nf, const float position[3], float taille, uint32_t material);
/* create_boxs (bvh.h:5-61) + flattening.  The arrays belong to the builder. */
int orc_build(orc_builder *b, rtx_scene_desc *out);

/* parse (meshs.h:60-125).  Returns 0 on success, 1 when the first line is not
 * "OFF", 2 when the file cannot be opened.  Arrays are malloc'ed. */
int orc_parse_off(const char *path, float **verts, uint32_t *nv, uint32_t **faces, uint32_t *nf);
void orc_free(void *p);

/* ---- rendering (raytracing.h, intersect.h, shade.h) ---------------------- */
typedef struct orc_stats {
    uint64_t primary_rays;
    uint64_t slab_tests;         /* intersect_box calls */
    uint64_t leaf_visits;        /* trace_ray calls */
    uint64_t primary_prim_tests; /* intersect_triangle/sphere calls, primary */
    uint64_t hits;               /* successful primary intersect_scene calls */
    uint64_t shadow_rays;
    uint64_t shadow_prim_tests;
    uint64_t shaded_samples;     /* shade() calls */
    uint64_t nan_channels;       /* pixel channels that reached clamp_color as NaN */
    uint64_t bounce_rays;        /* mirror rays (raytracing.h:51-61), max_depth > 0 only */
    uint64_t bounce_prim_tests;  /* primitive tests done for them */
} orc_stats;

/* Oracle-only bit of rtx_params::flags.  The reference adds the light samples of a hit one
 * after the other (raytracing.h:36-50); a GPU wave adds sample k to the running sum of lane
 * k % 64 and combines the 64 sums pairwise at the end (lane l with lane l ^ 32, ^ 16, ... ^ 1).
 * With this bit the oracle adds in that order, so that a comparison with the GPU's colours
 * (unsorted samples, IEEE build of the BSDF) is free of the order of summation. */
#define ORC_FLAG_WAVE_SUM 0x10000u

/* rgb: rtx_part_rows * width * 3 bytes (may be NULL).  hits: up to hit_cap
 * records in (pixel, ordinal) order (may be NULL); *n_hits gets the number
 * found.  fcolor: optional float RGB before clamp_color, same shape as rgb.
 * threads <= 0: all OpenMP threads.  Returns 0 on success. */
int orc_render(const rtx_scene_desc *scene, const rtx_params *p, uint8_t *rgb, float *fcolor,
               rtx_hit *hits, uint64_t hit_cap, uint64_t *n_hits, orc_stats *stats, int threads);
int orc_max_threads(void);

/* Function-level entry points for known-answer tests. */
int orc_intersect_sphere(const float o[3], const float d[3], const float c[3], float r,
                         float *t, float pos[3], float nrm[3]);
int orc_intersect_triangle(const float o[3], const float d[3], const float v0[3],
                           const float v1[3], const float v2[3], float *t, float pos[3],
                           float nrm[3]);
int orc_intersect_box(const float o[3], const float d[3], const float lb[3], const float rt[3]);
void orc_shade(const float n[3], const float v[3], const float l[3], const float lc[3],
               const rtx_material *m, float out[3]);
uint8_t orc_clamp_color(float v);
int orc_coord_out_of_scene(const float c[3]);
void orc_primary_ray(const rtx_params *p, uint32_t i, uint32_t j, float o[3], float d[3]);

#ifdef __cplusplus
}
#endif
#endif
