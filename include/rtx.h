/*
 * rtx.h - C ABI of the MI355X render hot path.
 *
 * The reference (romain-roy/ray-tracing) has no FFI layer: its seam is two free
 * functions in one translation unit, init_scene(Light&, Boxs&) (scene.h:4) and
 * render_image(Light&, Boxs&) (raytracing.h:100).  The entry points below are
 * what a binding for that seam calls: the host side flattens the `Boxs` tree
 * that init_scene / create_mesh / create_boxs produced into plain arrays
 * (rtx_scene_desc), uploads it once (rtx_scene_create) and renders frames
 * (rtx_render*).  Plain pointers and sizes only.
 *
 * Pixel addressing is the reference's (i, j) space (raytracing.h:114-130):
 * pixel (i, j) of a full frame is rgb[(j * width + i) * 3 + c]; j = 0 is the
 * row FreeImage stores at the bottom of out.png.
 */
#ifndef RTX_H
#define RTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 2u /* 2: rtx_stats grew render_attempts, record_capacity, screen_ms */
#define RTX_NONE 0xFFFFFFFFu

typedef enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID = 1,   /* bad argument or malformed scene description */
    RTX_ERR_NO_DEVICE = 2, /* no usable gfx950 device / extension not usable */
    RTX_ERR_HIP = 3,       /* a HIP call failed; see rtx_last_error() */
    RTX_ERR_NOMEM = 4,
    RTX_ERR_LIMIT = 5      /* a documented limit was exceeded */
} rtx_status;

/* Primitive kinds; same values as the reference's `Type` enum (structs.h:1-5). */
enum { RTX_PRIM_SPHERE = 1, RTX_PRIM_TRIANGLE = 2 };

/* Same fields and order as the reference's Material (structs.h:7-13), 32 bytes. */
typedef struct rtx_material {
    float ior;
    float roughness;
    float specular[3];
    float diffuse[3];
} rtx_material;

/* Same fields as the reference's Light (structs.h:39-43). */
typedef struct rtx_light {
    float position[3];
    float color[3];
} rtx_light;

/*
 * One primitive reference of a leaf box (an element of Box::objects,
 * structs.h:62-68).  A triangle that straddles a split plane is referenced by
 * both children (bvh.h:28-32): it appears once per leaf that holds it, always
 * with the same id.  48 bytes.
 */
typedef struct rtx_prim {
    uint32_t type;     /* RTX_PRIM_SPHERE | RTX_PRIM_TRIANGLE */
    float g[9];        /* triangle: v0 v1 v2; sphere: centre xyz, radius, 5 x 0 */
    uint32_t material; /* index into rtx_scene_desc::materials */
    uint32_t id;       /* global primitive id: index of the primitive in the
                          concatenation of the top-level boxes' object lists,
                          in push order */
} rtx_prim;

/*
 * One box of the hierarchy.  Boxes are stored in the depth-first pre-order in
 * which trace_ray_boxs (raytracing.h:74-98) visits them: top-level boxes in
 * push order, children as [low, high] (bvh.h:56-57).  `skip` is the index of
 * the next box in that order once this box's subtree is done (or missed), so
 * the recursive walk becomes `i = hit && inner ? i + 1 : nodes[i].skip`.
 * 32 bytes.
 */
typedef struct rtx_node {
    float lb[3];   /* Box::lb */
    float rt[3];   /* Box::rt */
    uint32_t skip; /* > own index, <= n_nodes */
    uint32_t leaf; /* RTX_NONE for an inner box (Box::depth < DEPTH_BOX), else
                      the leaf's index; leaves are numbered in visit order,
                      empty ones included */
} rtx_node;

/* The primitive references of one leaf: prims[first .. first + count). */
typedef struct rtx_leaf {
    uint32_t first;
    uint32_t count;
} rtx_leaf;

typedef struct rtx_scene_desc {
    uint32_t n_nodes;
    uint32_t n_leaves;
    uint32_t n_prims;
    uint32_t n_materials;
    const rtx_node *nodes;
    const rtx_leaf *leaves;
    const rtx_prim *prims;
    const rtx_material *materials;
    rtx_light light; /* position[2] is the height of the area light; x and y
                        are re-drawn per sample (raytracing.h:38-39) */
} rtx_scene_desc;

/* Limits (checked, RTX_ERR_LIMIT when exceeded). */
#define RTX_MAX_LEAVES 65535u  /* leaf index is 16 bits of the sampler counter */
#define RTX_MAX_LIGHTS 65535u  /* sample index is the other 16 bits:
                                  n_lights * (max_depth + 1) must not exceed it */
#define RTX_MAX_MATERIALS 256u

/*
 * Frame parameters.  The reference fixes all of them at compile time
 * (raytracing.h:12-21, :112); rtx_params_default() fills in those values.
 */
typedef struct rtx_params {
    uint32_t width;     /* WIDTH  (raytracing.h:12) */
    uint32_t height;    /* HEIGHT (raytracing.h:13) */
    uint32_t n_lights;  /* NB_LIGHTS (raytracing.h:19) */
    uint32_t max_depth; /* MAX_DEPTH (raytracing.h:18): mirror bounces, 0..8 */
    uint32_t seed;      /* key of the counter-based light sampler */
    float acne;         /* raytracing.h:21 */
    float camera[3];    /* raytracing.h:112 */
    /*
     * Row tiling for multi-GPU frames.  Rows are grouped in stripes of
     * `stripe_rows`; stripe s belongs to part s % n_parts.  A call renders
     * the stripes of `part` only and packs their rows, in increasing j, into
     * the output.  n_parts = 0 or 1 renders the whole frame.  The light
     * sampler is keyed by the global pixel index, so the assembled frame does
     * not depend on the split.
     */
    uint32_t stripe_rows;
    uint32_t n_parts;
    uint32_t part;
    uint32_t flags; /* RTX_FLAG_* */
} rtx_params;

#define RTX_FLAG_TIMING 1u /* fill rtx_stats::kernel_ms with HIP-event times */
#define RTX_FLAG_WAVE_PER_HIT 2u /* always use the one-wave-per-hit shading kernel, also for
                                    few light samples (for testing; results are identical) */
#define RTX_FLAG_IEEE_SHADE 4u   /* evaluate the BSDF (shade.h) with correctly rounded division,
                                    square root and expf instead of the hardware's 1-ulp
                                    reciprocal / square root / exp2.  Hits, distances and
                                    occlusion decisions are the same either way; colours differ
                                    by a few ulp.  About 15 % slower at 500 light samples. */
#define RTX_FLAG_NO_SORT 8u      /* shade every hit record with the plain kernel (light samples in
                                    index order); by default records with a large leaf and more
                                    than 128 light samples go to a second kernel that sorts the
                                    samples by light cell.  Same hits and occlusion decisions;
                                    colours differ by the order of the per-sample sum. */

#define RTX_FLAG_NO_SCREEN 16u    /* test every primitive of the leaf for every shadow ray, as the
                                    reference does (intersect.h:99-117), instead of first ruling
                                    out, per hit record, the triangles no shadow ray of the record
                                    can reach.  For testing: lit-sample counts and colours must not
                                    depend on it.  Several times slower. */

/*
 * One (pixel, leaf) hit: the nearest primitive a primary ray hits inside one
 * leaf box, i.e. one successful intersect_scene call of trace_ray
 * (raytracing.h:34).  The reference shades every such hit (not only the
 * nearest over the whole scene) and merges the colours in visit order.
 */
typedef struct rtx_hit {
    uint32_t pixel;   /* j * width + i (global, also under row tiling) */
    uint32_t ordinal; /* rank of this hit among the pixel's hits, visit order */
    uint32_t leaf;
    uint32_t prim;    /* rtx_prim::id */
    float t;          /* Intersection::distance */
    float color[3];   /* what trace_ray returned for this leaf */
} rtx_hit;

typedef struct rtx_stats {
    uint64_t primary_rays;  /* pixels rendered by this call */
    uint64_t hits;          /* (pixel, leaf) hit records */
    uint64_t shadow_rays;   /* (hits + mirror-bounce hits) * n_lights */
    uint64_t shadow_tests;  /* primitive tests the reference does for those rays:
                               n_lights * sum over those hits of the leaf's size */
    uint64_t lit_samples;   /* shadow rays that reached the light, i.e. shade()
                               evaluations (raytracing.h:45-48) */
    uint64_t shadow_tests_run; /* of shadow_tests, those the GPU actually executed;
                               the rest were ruled out per record by a conservative
                               geometric screen and cannot affect a result */
    float kernel_ms[4];     /* trace (with the sorting of the hit records into work
                               lists), shadow+shade kernel, merge, total on the stream
                               (all kernels, screen_ms included; only with RTX_FLAG_TIMING) */
    uint32_t render_attempts; /* times the frame's kernels ran: 1, or more when the hit
                               records outgrew their buffer and the frame was rendered
                               again into a larger one (first frames of a scene only) */
    uint32_t record_capacity; /* hit records the buffer held on the last attempt */
    float screen_ms;        /* the kernel that builds the records' candidate lists, between
                               kernel_ms[0] and kernel_ms[1] (only with RTX_FLAG_TIMING) */
    uint32_t reserved;
} rtx_stats;

/* Library / device ----------------------------------------------------------*/

uint32_t rtx_abi_version(void);
/* Number of HIP devices visible to the process (0 when there is none). */
int rtx_device_count(void);
/* Message for the calling thread's last failed call ("" if none). */
const char *rtx_last_error(void);

/* Device self-test with no counterpart in the reference.  The arithmetic shortcuts of
 * the kernels are compared with the plain IEEE evaluation they replace:
 *   out[0], out[1]  values checked / mismatches of the packed reciprocal against the
 *                   correctly rounded quotient 1.f / a (the reference's `f = 1.f / a`,
 *                   intersect.h:41), every binary32 value of its working range; and of
 *                   `|f| > 1e7f` against `-eps < a < eps` (intersect.h:39), every
 *                   binary32 value
 *   out[2], out[3]  cases checked / mismatching cases of the closed form of the acne
 *                   offset chain against the step-by-step float additions
 *                   (raytracing.h:40); out[4] = cases the closed form declined
 *   out[5], out[6]  values checked / mismatches of the parts of the two-vector normalisation
 *                   against `v / sqrt(dot(v, v))` (vec3.h:28-31): the packed square root
 *                   against sqrtf (every binary32 value of its working range), the packed
 *                   division by a shared denominator and the normalisation as a whole
 *                   (pseudo-random operands)
 * Mismatches must be 0. */
rtx_status rtx_selftest(int device, uint64_t out[7]);

void rtx_params_default(rtx_params *p);
/* Rows / pixels a call with these parameters renders (row tiling applied). */
uint32_t rtx_part_rows(const rtx_params *p);

/* Scene -----------------------------------------------------------------------
 * Replaces the in-memory `Boxs` argument of render_image (raytracing.h:100):
 * validates the description, converts it to the device layout and uploads it
 * to HIP device `device`.  The description is not referenced after return.
 * The first scene created on a device in a process runs a sampled version of
 * rtx_selftest there (one value in 64, under 2 ms) and fails with
 * RTX_ERR_NO_DEVICE if the device's arithmetic does not reproduce the IEEE
 * results the kernels' shortcuts rely on; rtx_selftest itself is exhaustive. */
typedef struct rtx_scene rtx_scene;
rtx_status rtx_scene_create(const rtx_scene_desc *desc, int device, rtx_scene **out);
void rtx_scene_destroy(rtx_scene *scene);

/* Render ----------------------------------------------------------------------
 * Replaces the pixel loop of render_image (raytracing.h:114-136).
 * rgb_host receives rtx_part_rows(p) * width * 3 bytes. */
rtx_status rtx_render(rtx_scene *scene, const rtx_params *p, uint8_t *rgb_host,
                      rtx_stats *stats /* may be NULL */);

/* Same, output left in device memory (d_rgb: device pointer on the scene's
 * device) and work enqueued on `hip_stream` (a hipStream_t, NULL = the null
 * stream).  The call is synchronous on that stream: it returns after the
 * frame is complete (the host reads the frame's counters back to detect a
 * full record buffer), with or without stats. */
rtx_status rtx_render_device(rtx_scene *scene, const rtx_params *p, void *d_rgb,
                             void *hip_stream, rtx_stats *stats);

/* rtx_render_device for callers that cannot allocate.  No counterpart in the
 * reference: it exists for callers that may not call hipMalloc / hipFree
 * while they render (tight serving loops, harnesses that forbid allocation,
 * processes whose device memory is fully committed).  Like rtx_render_device
 * it synchronises `hip_stream` before it returns (the frame's counters are
 * read back), so it cannot be recorded into a HIP graph: a capturing stream
 * makes the call fail with RTX_ERR_HIP.  Same arguments, same kernels and the
 * same output bytes as rtx_render_device, but every per-frame work buffer
 * (hit records, their colours, work lists and candidate lists, one record
 * head per pixel, tile queues, counters) is carved from a static device array of
 * RTX_STATIC_SCRATCH_BYTES that each GPU the library runs on holds from the
 * first rtx_scene_create on it until the process ends; the call allocates
 * and frees nothing.  The caller still passes the scene and d_rgb.
 *   Limits: 4 bytes per rendered pixel plus the tile queues come out of the
 * array first; what is left is split per hit record into 104 bytes for the
 * record itself and 24 entries (96 bytes) of the pool that holds the
 * records' candidate lists (a record needs as many entries as its leaf has
 * primitives; frames with at most 32 light samples keep no lists).  A frame
 * that leaves room for fewer than RTX_STATIC_MIN_RECORDS records, that
 * produces more hit records than fit (in any of the 8 equal shards of the
 * record array), or whose lists need more than the pool holds, fails with
 * RTX_ERR_LIMIT and a message; nothing is retried or grown.  As a guide:
 * 3840 x 2160 leaves about 165 000 records, 1920 x 1080 about 290 000.  (The
 * array is part of the library's code object: every process that loads the
 * library pays for its allocation once per device, about 0.35 ms per MiB.)
 *   All calls on one GPU share that GPU's array (whatever the scene or
 * stream): calls that overlap in time on one GPU race and give undefined
 * frames; calls on different GPUs are independent. */
#define RTX_STATIC_SCRATCH_BYTES (64u << 20)
#define RTX_STATIC_MIN_RECORDS 4096u
rtx_status rtx_render_device_static(rtx_scene *scene, const rtx_params *p, void *d_rgb,
                                    void *hip_stream, rtx_stats *stats);

/* Page-locked host memory for rgb_host: rtx_render copies a frame into such a buffer
 * at the full rate of the link; into ordinary memory it runs at roughly half of it.  No
 * counterpart in the reference (its frame lives in a FreeImage bitmap,
 * raytracing.h:105). */
rtx_status rtx_host_alloc(uint64_t bytes, void **out);
void rtx_host_free(void *p);

/* Render and also return the per-(pixel, leaf) hit records, sorted by
 * (pixel, ordinal).  At most `capacity` records are written; *n_hits receives
 * the number found.  rgb_host may be NULL. */
rtx_status rtx_render_hits(rtx_scene *scene, const rtx_params *p, rtx_hit *hits,
                           uint64_t capacity, uint64_t *n_hits, uint8_t *rgb_host);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
