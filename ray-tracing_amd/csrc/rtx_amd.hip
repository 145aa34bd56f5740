// rtx_amd.hip - MI355X (gfx950) render hot path behind the C ABI of include/rtx.h.
//
// One frame = these kernels on one stream, no host round trip in between:
//   tile_list_kernel (when the boxes leave much of the frame empty) lists the tiles of the
//                    frame that a primary ray can hit anything in
//   trace_kernel     one lane per pixel: primary ray, box walk, nearest hit per
//                    leaf (raytracing.h:74-98, :29-34; intersect.h) -> hit records
//   classify_kernel  one lane per hit record: the record goes to the work list of its
//                    class (how it will be shaded, how heavy its leaf is)
//   screen_kernel    sixteen hit records per wave, each with lanes over its leaf: the
//                    primitives the record's shadow rays can reach at all -> one
//                    candidate list per record
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
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_device.h"

// Kernels (device code), in dependency order
#include "launch_config.h"
#include "trace_kernel.h"
#include "bsdf.h"
#include "shade_record.h"
#include "shade_sorted.h"
#include "screen_kernel.h"
#include "shade_kernels.h"
#include "merge_kernel.h"
#include "selftest_kernels.h"

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

// ---- static work area of rtx_render_device_static ---------------------------------
// Per-frame work buffers (hit records, colours and work lists, per-pixel record heads,
// tile queues, counters) for callers that may not allocate.  Every device the library
// runs on holds its own copy of the array until the process ends.  In host code the
// array's name is only a handle for hipGetSymbolAddress: the address the kernels get is
// the one resolved per device in scratch_resolve().
#define STATIC_SCRATCH_BYTES RTX_STATIC_SCRATCH_BYTES
#define STATIC_POOL_PER_RECORD 24u // entries of the candidate pool per hit record in the static work area
#define POOL_PER_RECORD 64u        // first guess for the allocated pool
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

#define SCRATCH_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static unsigned char *g_scratch_ptr[SCRATCH_MAX_DEVICES]; // per device: address of its copy of `scratch`

// Resolves the address of `device`'s copy of the work area, once.  The device must be current.
static rtx_status scratch_resolve(int device) {
    if (device < 0 || device >= SCRATCH_MAX_DEVICES) return fail(RTX_ERR_LIMIT, "device index beyond the library's table of 64 devices");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return RTX_OK;
    unsigned char *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(RTX_ERR_HIP, "hipGetSymbolAddress returned a null address for the static work area");
    g_scratch_ptr[device] = ptr;
    return RTX_OK;
}

// The work buffers of one frame: the scene's own (grown on demand) or a carving of the
// static work area (fixed).
struct FrameBufs {
    DRec *recs = nullptr;
    float4 *colors = nullptr; // rec_cap colours followed by the work lists
    uint32_t rec_cap = 0;
    uint32_t *first = nullptr;
    uint32_t *tiles = nullptr;
    DCounters *ctr = nullptr;
    // candidate lists of the records: block offset and length per record (rec_cap each), the pool
    uint32_t *cand_off = nullptr, *cand_n = nullptr, *pool = nullptr;
    uint32_t pool_cap = 0;
};

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
    uint32_t trace_slices = 1; // lanes per primary ray in trace_kernel (1 or 2)
    float coord_max = 0.f;     // largest |coordinate| of any primitive (sphere: centre +- radius)
    rtx_light light{};
    // per-frame work buffers, grown on demand
    DRec *d_recs = nullptr;
    uint64_t n_hits = 0; // hit records of the last frame
    float4 *d_colors = nullptr;
    uint32_t rec_cap = 0;
    uint32_t *d_cand = nullptr; // rec_cap block offsets, then rec_cap list lengths
    uint32_t *d_pool = nullptr; // the candidate lists
    uint64_t pool_cap = 0;
    uint32_t *d_first = nullptr;
    uint64_t pix_cap = 0;
    uint32_t *d_tiles = nullptr; // tile queues of the trace kernel, then one flag byte per tile part
    uint64_t tile_cap = 0;
    std::vector<float> top_boxes; // lb, rt of every top-level box (6 floats each)
    uint8_t *d_rgb = nullptr; // staging for rtx_render's host output
    uint64_t rgb_cap = 0;
    DCounters *d_ctr = nullptr;
    DCounters *h_ctr = nullptr; // pinned
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

extern "C" uint32_t rtx_abi_version(void) { return RTX_ABI_VERSION; }

extern "C" int rtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *rtx_last_error(void) { return g_err.c_str(); }

// RTX_TRACE_COLD=1 in the environment: where the time of the first calls goes, on stderr.
struct ColdTrace {
    bool on = getenv("RTX_TRACE_COLD") != nullptr;
    double t = now();
    static double now() {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    void mark(const char *where, const char *what) {
        if (!on) return;
        const double n = now();
        fprintf(stderr, "%s: %-52s %8.3f ms\n", where, what, n - t);
        t = n;
    }
};

// sampled = false: the exhaustive comparisons (some 50 ms).  sampled = true: one bit pattern
// pair in 64, 1/64 of the chain cases and 1/32 of the division operands (about 1 ms of kernel
// time); a part whose v_rcp_f32 / v_rsq_f32 differ from gfx950's would fail either.
static rtx_status selftest_run(int device, uint64_t out[7], bool sampled) {
    if (!out) return fail(RTX_ERR_INVALID, "null argument");
    if (device < 0 || device >= rtx_device_count()) return fail(RTX_ERR_NO_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    ColdTrace tr;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 56));
    tr.mark("selftest", "hipMalloc");
    hipError_t e = hipMemsetAsync(d, 0, 56, nullptr);
    tr.mark("selftest", "hipMemsetAsync");
    if (e == hipSuccess) {
        // (one after the other on the null stream; a single synchronisation at the end)
        selftest_rcp_kernel<<<sampled ? 1024 : 4096, 256>>>(d + 1, d, d + 6, d + 5, sampled ? 64u : 1u);
        e = hipGetLastError();
        tr.mark("selftest", "first kernel launch");
        if (e == hipSuccess) {
            selftest_chain_kernel<<<sampled ? 256 : 16384, 256>>>(d + 3, d + 2, d + 4);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            selftest_div_kernel<<<1024, 256>>>(d + 6, d + 5, sampled ? 128u : 4096u);
            e = hipGetLastError();
        }
    }
    unsigned long long h[7] = {0, 0, 0, 0, 0, 0, 0};
    tr.mark("selftest", "other launches");
    if (e == hipSuccess) e = hipMemcpy(h, d, 56, hipMemcpyDeviceToHost); // (synchronises)
    tr.mark("selftest", "kernels run, counters copied back");
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RTX_ERR_HIP, std::string("selftest: ") + hipGetErrorString(e));
    for (int k = 0; k < 7; k++) out[k] = h[k];
    return RTX_OK;
}

extern "C" rtx_status rtx_selftest(int device, uint64_t out[7]) { return selftest_run(device, out, false); }

// The kernels' exact arithmetic shortcuts (reciprocal, square root, division, determinant
// threshold) start from the hardware's v_rcp_f32 / v_rsq_f32 and are exact because of what
// those return on gfx950, as rtx_selftest establishes value by value.  Hit ids that are
// silently off on a part that rounds differently would be worse than no result: the first
// scene created on a device runs a sampled self-test there (under 2 ms; rtx_selftest is the
// exhaustive one) and refuses the device if anything mismatches.
static std::mutex g_check_mutex;
static int g_checked[SCRATCH_MAX_DEVICES]; // 0 not yet, 1 passed, 2 failed
static rtx_status arithmetic_check(int device) {
    if (device < 0 || device >= SCRATCH_MAX_DEVICES) return fail(RTX_ERR_LIMIT, "device index beyond the library's table of 64 devices");
    std::lock_guard<std::mutex> lock(g_check_mutex);
    if (g_checked[device] == 0) {
        uint64_t out[7];
        if (rtx_status rs = selftest_run(device, out, true); rs != RTX_OK) return rs;
        g_checked[device] = (out[1] == 0 && out[3] == 0 && out[6] == 0) ? 1 : 2;
    }
    if (g_checked[device] == 2)
        return fail(RTX_ERR_NO_DEVICE, "the device's arithmetic self-test found mismatches with the IEEE results "
                                       "(see rtx_selftest): refusing to render on it");
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
    (void)hipFree(s->d_cand);
    (void)hipFree(s->d_pool);
    (void)hipFree(s->d_first);
    (void)hipFree(s->d_rgb);
    (void)hipFree(s->d_tiles);
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
    double coord_max = 0.0;
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
            {
                const double nx = (double)e1y * e2z - (double)e1z * e2y, ny = (double)e1z * e2x - (double)e1x * e2z,
                             nz = (double)e1x * e2y - (double)e1y * e2x;
                const double area2 = sqrt(nx * nx + ny * ny + nz * nz);
                const double shape = (area2 > 0.0) ? l1 * l2 / area2 * 1.0001 : INFINITY;
                o.shape = (shape < 3.0e38) ? (float)shape : INFINITY;
                o.ell = (float)std::min((l1 + l2) * 1.0001, 3.0e38);
            }
            for (int c = 0; c < 9; c++) coord_max = std::max(coord_max, (double)fabsf(p.g[c]));
        } else {
            o.a = host_f3(p.g);
            volatile float r2 = p.g[3] * p.g[3];
            o.b.x = p.g[3];
            o.b.y = r2;
            for (int c = 0; c < 3; c++) coord_max = std::max(coord_max, (double)fabsf(p.g[c]) + fabs((double)p.g[3]));
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
    ColdTrace tr;
    auto mark = [&](const char *what) { tr.mark("rtx_scene_create", what); };
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RTX_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950");
    mark("device selected, properties read");
    if (rtx_status rs = scratch_resolve(device); rs != RTX_OK) return rs;
    mark("static work area located (loads the code object)");
    if (rtx_status rs = arithmetic_check(device); rs != RTX_OK) return rs;
    mark("arithmetic self-test (sampled, first scene only)");

    rtx_scene *s = new rtx_scene();
    s->device = device;
    s->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    s->n_nodes = NK;
    s->nest_levels = nest_levels;
    for (uint32_t n = 0; n < NK;) { // top-level boxes: skip every subtree
        s->top_boxes.insert(s->top_boxes.end(), nodes[n].lb, nodes[n].lb + 3);
        s->top_boxes.insert(s->top_boxes.end(), nodes[n].rt, nodes[n].rt + 3);
        n = (nodes[n].info == RTXD_NONE) ? nodes[n].link : n + 1;
    }
    s->wide_dets = wide_dets;
    s->coord_max = (coord_max < 3.0e38) ? (float)coord_max : 3.0e38f;
    s->trace_slices = (desc->n_leaves && desc->n_prims / desc->n_leaves >= TRACE_SLICE_MIN_LEAF) ? 2u : 1u;
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
    for (int k = 0; k < 5 && e == hipSuccess; k++) e = hipEventCreate(&s->ev[k]);
    if (e != hipSuccess) {
        rtx_scene_destroy(s);
        return fail(e == hipErrorOutOfMemory ? RTX_ERR_NOMEM : RTX_ERR_HIP,
                    std::string("scene upload: ") + hipGetErrorString(e));
    }
    mark("scene arrays, counters, events allocated and uploaded");
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

// Margin of the shading kernel's candidate screens for a frame (see shade_record.h): the
// fixed CULL_MARGIN, raised in scenes with large coordinates so that half of it always
// covers the rounding of the screens' own plane arithmetic (at most 2^-20 times the largest
// coordinate in play: vertices, light rectangle, start points after all their acne steps).
// RTX_FLAG_NO_SCREEN: a margin no triangle can exceed, i.e. every leaf primitive is tested.
static float screen_margin(const rtx_scene *s, const rtx_params *p) {
    if (p->flags & RTX_FLAG_NO_SCREEN) return 3.0e38f;
    double r = std::max<double>(s->coord_max, 750.0);
    r = std::max(r, fabs((double)s->light.position[2]));
    r += fabs((double)p->acne) * ((double)p->n_lights + 1.0);
    // (the screens take maxima of products of coordinates: beyond this they could overflow, and a
    // NaN would drop out of a maximum instead of failing a comparison)
    if (!(r < 1.0e18)) return 3.0e38f;
    const double m = std::max<double>(CULL_MARGIN, r * (1.0 / 262144.0)); // 2^-18
    return (m < 3.0e38) ? (float)m : 3.0e38f;
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
    (void)hipFree(s->d_cand);
    s->d_recs = nullptr;
    s->d_colors = nullptr;
    s->d_cand = nullptr;
    s->rec_cap = 0;
    hipError_t e = hipMalloc((void **)&s->d_recs, need * sizeof(DRec));
    // the colours are followed by the work lists of the shading kernel
    if (e == hipSuccess)
        e = hipMalloc((void **)&s->d_colors, need * (sizeof(float4) + N_CLASSES * sizeof(uint32_t)));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_cand, need * 2 * sizeof(uint32_t));
    if (e != hipSuccess) return fail(RTX_ERR_NOMEM, std::string("record buffers: ") + hipGetErrorString(e));
    s->rec_cap = (uint32_t)need;
    return RTX_OK;
}

// Pixel rectangles of the top-level boxes: the eye, a pixel's ray origin on the plane
// z = 1000 and every point of its ray are collinear (primary_ray), so the rays that can
// enter a box are those whose pixel lies in the perspective projection of the box from
// the eye onto that plane, a convex set inside the bounding rectangle of the eight
// projected corners.  Two pixels are added all round (the kernels' float arithmetic differs
// from this one by a tiny fraction of a pixel).  A box that reaches the eye's height, or an
// eye at or below the plane, gives the whole frame.  Returns the fraction of the frame the
// rectangles cover at most (their areas summed).
static double tile_rects(const rtx_scene *s, const rtx_params *p, DTileRects *out) {
    const double W = p->width, H = p->height, sc = 1000.0 / H;
    const double cx = p->camera[0], cy = p->camera[1], cz = p->camera[2];
    const size_t n_boxes = s->top_boxes.size() / 6;
    out->n = 0;
    bool whole = !(cz > 1000.0 + 1e-3) || n_boxes == 0;
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300, area = 0.0; // union, for too many boxes
    for (size_t b = 0; b < n_boxes && !whole; b++) {
        const float *q = &s->top_boxes[b * 6];
        double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
        for (int c = 0; c < 8 && !whole; c++) {
            const double qx = q[(c & 1) ? 3 : 0], qy = q[(c & 2) ? 4 : 1], qz = q[(c & 4) ? 5 : 2];
            const double below = cz - qz; // the corner must lie below the eye
            if (!(below > 1e-6 * (fabs(cz) + fabs(qz) + 1.0))) {
                whole = true;
                break;
            }
            const double lam = (cz - 1000.0) / below;
            const double i = (cx + (qx - cx) * lam - 500.0) / sc + W * 0.5, j = (cy + (qy - cy) * lam) / sc;
            if (!(fabs(i) < 1e12 && fabs(j) < 1e12)) {
                whole = true;
                break;
            }
            bx0 = std::min(bx0, i), bx1 = std::max(bx1, i), by0 = std::min(by0, j), by1 = std::max(by1, j);
        }
        if (whole) break;
        bx0 = std::max(0.0, floor(bx0) - 2.0), by0 = std::max(0.0, floor(by0) - 2.0);
        bx1 = std::min(W - 1.0, ceil(bx1) + 2.0), by1 = std::min(H - 1.0, ceil(by1) + 2.0);
        if (bx0 > bx1 || by0 > by1) continue; // off the frame
        area += (bx1 - bx0 + 1.0) * (by1 - by0 + 1.0);
        x0 = std::min(x0, bx0), x1 = std::max(x1, bx1), y0 = std::min(y0, by0), y1 = std::max(y1, by1);
        if (out->n < TILE_RECTS) {
            out->x0[out->n] = (int32_t)bx0, out->x1[out->n] = (int32_t)bx1;
            out->y0[out->n] = (int32_t)by0, out->y1[out->n] = (int32_t)by1;
        }
        out->n++;
    }
    if (whole) {
        out->n = 1;
        out->x0[0] = 0, out->y0[0] = 0, out->x1[0] = (int32_t)p->width - 1, out->y1[0] = (int32_t)p->height - 1;
        return 1.0;
    }
    if (out->n > TILE_RECTS) { // more boxes than rectangles: one rectangle around all of them
        out->n = 1;
        out->x0[0] = (int32_t)x0, out->x1[0] = (int32_t)x1, out->y0[0] = (int32_t)y0, out->y1[0] = (int32_t)y1;
        area = (x1 - x0 + 1.0) * (y1 - y0 + 1.0);
    }
    return std::min(1.0, area / (W * H));
}

// Tile grid of the trace kernel for a frame of `rows` rows: whether the tiles are handed
// out through queues, lanes per primary ray, tiles, entries per queue, words of the queue
// buffer (queues + one flag byte per tile part).
struct TilePlan {
    bool queued;
    double cover; // share of the frame inside the projected top-level boxes, at most
    uint32_t slices, tile_h, tiles, tile_seg;
    uint64_t words;
    DTileRects rects;
};
static TilePlan plan_tiles(const rtx_scene *s, const rtx_params *p, uint32_t rows) {
    TilePlan t;
    // Tiles through queues when the boxes leave a good part of the frame empty: only the
    // tile parts that can hit something are traced, each by the next free wave.  Where
    // nearly every tile has work the fixed assignment is cheaper.
    const double cover = tile_rects(s, p, &t.rects);
    t.cover = cover;
    t.queued = TRACE_QUEUE && cover < TRACE_QUEUE_MAX_COVER;
    // Lanes per primary ray.  With queues the extra waves walk only the listed tiles.  In a
    // scene with large leaves two lanes share a ray; when the listed tile parts (8 x 4 pixels
    // each, estimated from the rectangles' area) are fewer than three quarters of the resident
    // waves - a small frame, or one GPU's share of a split frame - the kernel lasts as long as
    // its heaviest parts, and four lanes per ray halve those (measured on buddha, bunny and
    // the multi-mesh scene at 1, 1/2, 1/4 and 1/8 of their frames).
    t.slices = s->trace_slices;
    if (t.queued && s->trace_slices == 2) {
        const double listed = cover * (double)rows * p->width / 32.0;
        const double resident = (double)s->n_cu * TRACE_QUEUE_BLOCKS_PER_CU * (TRACE_BLOCK / 64);
        t.slices = (listed < 0.75 * resident) ? 4u : TRACE_SLICES_QUEUED;
    }
    t.tile_h = 16 / t.slices;
    t.tiles = ((p->width + 15) / 16) * ((rows + t.tile_h - 1) / t.tile_h);
    t.tile_seg = 4 * ((t.tiles + TILE_QUEUES - 1) / TILE_QUEUES);
    t.words = t.queued ? (uint64_t)t.tile_seg * TILE_QUEUES + ((uint64_t)t.tiles * 4 + 3) / 4 : 0;
    return t;
}

// Enqueue and complete one frame.  On return s->h_ctr holds the frame's counters.
// fixed = nullptr: the scene's own work buffers, allocated and grown on demand.
// fixed != nullptr: exactly these buffers (already sized for the frame); nothing is
// allocated or freed, and a frame with more hit records than they hold is an error.
static rtx_status render_frame(rtx_scene *s, const rtx_params *p, uint8_t *d_rgb, hipStream_t st,
                               rtx_stats *stats, const FrameBufs *fixed = nullptr) {
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
    fr.sort_samples = (p->n_lights >= SORT_MIN_LIGHTS && !(p->flags & RTX_FLAG_NO_SORT)) ? 1u : 0u;
    fr.seed = p->seed;
    fr.acne = p->acne;
    fr.cull_margin = screen_margin(s, p);
    fr.camera = host_f3(p->camera);
    fr.light_pos = host_f3(s->light.position);
    fr.light_color = host_f3(s->light.color);
    fr.scale = 1000.f / (float)p->height;
    fr.half_width = (float)p->width * 0.5f;

    const TilePlan plan = plan_tiles(s, p, rows);
    const bool queued = plan.queued;
    const uint32_t slices = plan.slices, tile_h = plan.tile_h, tiles = plan.tiles, tile_seg = plan.tile_seg;
    const DTileRects &rects = plan.rects;
    if (!fixed) {
        if (hipError_t e = grow(&s->d_first, &s->pix_cap, npix); e != hipSuccess)
            return fail(RTX_ERR_NOMEM, std::string("pixel buffer: ") + hipGetErrorString(e));
        if (s->rec_cap == 0) {
            // A first guess (96 bytes per record); a frame with more hits grows the buffers to
            // what it needs and is rendered again.  Only pixels inside the projected top-level
            // boxes can have records; of the reference's scenes the committed one has the most
            // per such pixel (1.3; a ray that meets a mesh at all crosses two or three of its
            // leaf columns), so 2.5 per covered pixel holds a first frame of any of them with
            // room for the uneven filling of the shards.
            const uint64_t guess = (uint64_t)(2.5 * plan.cover * (double)npix);
            rs = ensure_records(s, std::max<uint64_t>(std::min<uint64_t>(guess, std::max<uint64_t>(npix / 2, 1u << 20)),
                                                      1u << 16));
            if (rs != RTX_OK) return rs;
        }
        if (queued)
            if (hipError_t e = grow(&s->d_tiles, &s->tile_cap, plan.words); e != hipSuccess)
                return fail(RTX_ERR_NOMEM, std::string("tile queues: ") + hipGetErrorString(e));
        // the candidate pool: one entry per primitive of every record's leaf.  A first guess of
        // POOL_PER_RECORD per record (the reference's meshes average 20 to 70); a frame that needs
        // more is rendered again with what it needs.
        if (p->n_lights > 32 || (p->flags & RTX_FLAG_WAVE_PER_HIT))
            if (hipError_t e = grow(&s->d_pool, &s->pool_cap, (uint64_t)s->rec_cap * POOL_PER_RECORD); e != hipSuccess)
                return fail(RTX_ERR_NOMEM, std::string("candidate pool: ") + hipGetErrorString(e));
    }
    FrameBufs wb;
    if (fixed) {
        wb = *fixed;
    } else {
        wb.recs = s->d_recs, wb.colors = s->d_colors, wb.rec_cap = s->rec_cap;
        wb.first = s->d_first, wb.tiles = s->d_tiles, wb.ctr = s->d_ctr;
        wb.cand_off = s->d_cand, wb.cand_n = s->d_cand + s->rec_cap, wb.pool = s->d_pool;
        wb.pool_cap = (uint32_t)std::min<uint64_t>(s->pool_cap, 0xFFFFFFFFu);
    }
    const bool timing = stats && (p->flags & RTX_FLAG_TIMING);
    uint8_t *d_traced = queued ? reinterpret_cast<uint8_t *>(wb.tiles + (uint64_t)tile_seg * TILE_QUEUES) : nullptr;
    // the queue-driven shading kernels get exactly the waves that are resident at once
    // (a block is one wave per SIMD); the few-samples kernel strides over the records
#ifndef FEW_BLOCKS_PER_CU
#define FEW_BLOCKS_PER_CU 8
#endif
    const uint32_t few_blocks = (uint32_t)s->n_cu * FEW_BLOCKS_PER_CU;
    const uint32_t trace_blocks = (uint32_t)s->n_cu * TRACE_BLOCKS_PER_CU;
    // screen_kernel: a batch of records per wave when the record buffer were full (the hardware's
    // block scheduler balances the batches; waves beyond the frame's records leave at once)
    const uint32_t screen_per_block = SCREEN_BATCH * (SCREEN_BLOCK / 64);
    const uint32_t shade_waves = fr.sort_samples ? SHADE_WAVES_PER_SIMD_SORT : SHADE_WAVES_PER_SIMD;
    uint32_t shade_blocks = (uint32_t)s->n_cu * shade_waves;
    const uint32_t bounce_blocks = (uint32_t)s->n_cu * SHADE_WAVES_PER_SIMD_BOUNCE;
    // A frame with few hit records per resident wave (a small share of a frame split over
    // several GPUs) ends with most SIMDs waiting for the last records; with fewer waves per
    // SIMD every record finishes sooner and the kernel with it (measured on 1/4, 1/8 and 1/16 of
    // the buddha frame: 5, 4-5 and 4 waves instead of 6 take 2, 6 and 17 % less).  The number
    // of records is the previous frame's of this scene handle; the first frame takes the default.
    if (fr.sort_samples && s->n_hits) {
        const uint64_t per_wave = s->n_hits / ((uint64_t)s->n_cu * 4 * shade_waves);
        if (per_wave < 4) shade_blocks = (uint32_t)s->n_cu * 4;
        else if (per_wave < 14) shade_blocks = (uint32_t)s->n_cu * 5;
    }
    const uint32_t merge_blocks = (uint32_t)((npix + MERGE_BLOCK - 1) / MERGE_BLOCK);

    uint64_t n_hits = 0;
    uint32_t attempts = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        attempts++;
        HIP_TRY(hipMemsetAsync(wb.ctr, 0, sizeof(DCounters), st));
        if (timing) HIP_TRY(hipEventRecord(s->ev[0], st));
#define LAUNCH_TRACE(SL)                                                                                     \
    do {                                                                                                     \
        if (queued) {                                                                                        \
            tile_list_kernel<SL><<<(tiles * 4 + 255) / 256, 256, 0, st>>>(fr, rects, d_traced, wb.tiles,   \
                                                                         tile_seg, wb.ctr);                \
            trace_kernel<SL, true><<<(uint32_t)s->n_cu * TRACE_QUEUE_BLOCKS_PER_CU, TRACE_BLOCK, 0, st>>>(   \
                fr, s->d_nodes, s->n_nodes, s->d_prims, wb.recs, wb.rec_cap, wb.first, wb.tiles,       \
                tile_seg, wb.ctr);                                                                         \
        } else {                                                                                             \
            trace_kernel<SL, false><<<std::min<uint32_t>(tiles, trace_blocks), TRACE_BLOCK, 0, st>>>(        \
                fr, s->d_nodes, s->n_nodes, s->d_prims, wb.recs, wb.rec_cap, wb.first, nullptr, 0,       \
                wb.ctr);                                                                                   \
        }                                                                                                    \
    } while (0)
        if (slices == 4)
            LAUNCH_TRACE(4);
        else if (slices == 2)
            LAUNCH_TRACE(2);
        else
            LAUNCH_TRACE(1);
#undef LAUNCH_TRACE
#define LAUNCH_FEW(G)                                                                                        \
    do {                                                                                                     \
        if (ieee)                                                                                            \
            shade_few_kernel<G, true><<<few_blocks, SHADE_BLOCK, 0, st>>>(                                 \
                fr, s->d_prims, s->d_mats, wb.recs, wb.rec_cap, wb.colors, wb.ctr);                    \
        else                                                                                                 \
            shade_few_kernel<G, false><<<few_blocks, SHADE_BLOCK, 0, st>>>(                                \
                fr, s->d_prims, s->d_mats, wb.recs, wb.rec_cap, wb.colors, wb.ctr);                    \
    } while (0)
#define LAUNCH_SHADE(B, I, W)                                                                                \
    shade_kernel<B, I, W><<<(B) ? bounce_blocks : shade_blocks, SHADE_BLOCK, 0, st>>>(                       \
        fr, s->d_prims, s->d_mats, wb.recs, wb.rec_cap, wb.colors, lists, wb.ctr)
        const CandLists lists = {wb.pool, wb.cand_off, wb.cand_n};
        const uint32_t nlg = p->n_lights;
        const bool ieee = (p->flags & RTX_FLAG_IEEE_SHADE) != 0;
        const bool few = nlg <= 32 && !(p->flags & RTX_FLAG_WAVE_PER_HIT);
        if (!few)
            classify_kernel<<<few_blocks, 256, 0, st>>>(fr, wb.recs, wb.rec_cap, wb.colors, wb.cand_off, wb.ctr);
        if (timing) HIP_TRY(hipEventRecord(s->ev[4], st));
#ifdef SCREEN_GRID_RESIDENT
        const uint32_t screen_blocks = (uint32_t)s->n_cu * SCREEN_GRID_RESIDENT;
        (void)screen_per_block;
#else
        const uint32_t screen_blocks = std::max<uint32_t>(
            (uint32_t)s->n_cu * SCREEN_WAVES, std::min<uint32_t>((wb.rec_cap + screen_per_block - 1) / screen_per_block, 1u << 18));
#endif
        if (!few)
            screen_kernel<<<screen_blocks, SCREEN_BLOCK, 0, st>>>(fr, s->d_prims, wb.recs, wb.rec_cap, wb.cand_off, wb.cand_n,
                                                                  wb.pool, wb.pool_cap, wb.colors, wb.ctr);
        if (timing) HIP_TRY(hipEventRecord(s->ev[1], st)); // so that [1] is the shading kernel alone
        if (few) {
            // few samples: several hit records per wave
            if (nlg <= 1) LAUNCH_FEW(1);
            else if (nlg <= 2) LAUNCH_FEW(2);
            else if (nlg <= 4) LAUNCH_FEW(4);
            else if (nlg <= 8) LAUNCH_FEW(8);
            else if (nlg <= 16) LAUNCH_FEW(16);
            else LAUNCH_FEW(32);
        } else if (fr.sort_samples) {
            if (ieee) LAUNCH_SHADE(false, true, SHADE_WAVES_PER_SIMD_SORT);
            else LAUNCH_SHADE(false, false, SHADE_WAVES_PER_SIMD_SORT);
        } else {
            if (ieee) LAUNCH_SHADE(false, true, SHADE_WAVES_PER_SIMD);
            else LAUNCH_SHADE(false, false, SHADE_WAVES_PER_SIMD);
        }
        if (p->max_depth) {
            // Mirror bounces (raytracing.h:51-61): the kernel above has added every primary hit's
            // own light samples.  bounce_find_kernel lists the records whose reflected ray hits
            // something in its leaf (a few in a hundred), and the bounce build of the shading
            // kernel adds the deeper levels of those.  The work lists start again from empty.
            HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(wb.ctr) + offsetof(DCounters, cls), 0,
                                   offsetof(DCounters, tiles) - offsetof(DCounters, cls), st));
            bounce_find_kernel<<<few_blocks, 256, 0, st>>>(fr, s->d_prims, s->d_mats, wb.recs, wb.rec_cap, wb.colors,
                                                           wb.ctr);
            if (ieee) LAUNCH_SHADE(true, true, SHADE_WAVES_PER_SIMD_BOUNCE);
            else LAUNCH_SHADE(true, false, SHADE_WAVES_PER_SIMD_BOUNCE);
        }
        if (timing) HIP_TRY(hipEventRecord(s->ev[2], st));
        merge_kernel<<<merge_blocks, MERGE_BLOCK, (size_t)s->nest_levels * 3 * MERGE_BLOCK * sizeof(float), st>>>(fr, s->d_nodes, s->d_up, wb.recs,
                                                          wb.colors, wb.first, d_traced, tile_h, d_rgb);
        if (timing) HIP_TRY(hipEventRecord(s->ev[3], st));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(s->h_ctr, wb.ctr, offsetof(DCounters, head), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        uint32_t fullest = 0;
        n_hits = 0;
        for (int k = 0; k < REC_SHARDS; k++) {
            fullest = std::max(fullest, s->h_ctr->shard[k].count);
            n_hits += s->h_ctr->shard[k].count;
        }
        const uint64_t pool_need = few ? 0 : s->h_ctr->cand_total;
        const bool recs_fit = fullest <= wb.rec_cap / REC_SHARDS, pool_fits = pool_need <= wb.pool_cap;
        if (recs_fit && pool_fits) break;
        if (fixed && recs_fit)
            return fail(RTX_ERR_LIMIT, "the frame's candidate lists (" + std::to_string(pool_need) +
                                           " entries) exceed what the static work area holds for them (" +
                                           std::to_string(wb.pool_cap) + "); use rtx_render_device");
        if (recs_fit) { // only the candidate pool was too small: now its size is known
            if (attempt == 2) return fail(RTX_ERR_HIP, "candidate pool overflow persisted");
            if (pool_need > 0xFFFFFFF0ull) return fail(RTX_ERR_LIMIT, "more than 2^32 candidate list entries in one call");
            if (hipError_t e = grow(&s->d_pool, &s->pool_cap, pool_need + pool_need / 8 + 1024); e != hipSuccess)
                return fail(RTX_ERR_NOMEM, std::string("candidate pool: ") + hipGetErrorString(e));
            wb.pool = s->d_pool, wb.pool_cap = (uint32_t)std::min<uint64_t>(s->pool_cap, 0xFFFFFFFFu);
            continue;
        }
        // a shard got more hits than it holds
        if (fixed)
            return fail(RTX_ERR_LIMIT, "the frame has more hit records (" + std::to_string(n_hits) +
                                           ") than the static work area holds for it (" +
                                           std::to_string(wb.rec_cap) + " in " + std::to_string(REC_SHARDS) +
                                           " equal shards); use rtx_render_device");
        // grow the record buffer and render again
        if (attempt == 2) return fail(RTX_ERR_HIP, "record buffer overflow persisted");
        rs = ensure_records(s, ((uint64_t)fullest + fullest / 8 + 128) * REC_SHARDS);
        if (rs != RTX_OK) return rs;
        wb.recs = s->d_recs, wb.colors = s->d_colors, wb.rec_cap = s->rec_cap;
        wb.cand_off = s->d_cand, wb.cand_n = s->d_cand + s->rec_cap;
        if (!few) {
            if (hipError_t e = grow(&s->d_pool, &s->pool_cap, (uint64_t)s->rec_cap * POOL_PER_RECORD); e != hipSuccess)
                return fail(RTX_ERR_NOMEM, std::string("candidate pool: ") + hipGetErrorString(e));
            wb.pool = s->d_pool, wb.pool_cap = (uint32_t)std::min<uint64_t>(s->pool_cap, 0xFFFFFFFFu);
        }
    }
#ifdef PHASE_TIMES
    {
        const unsigned long long *c = reinterpret_cast<const unsigned long long *>(s->h_ctr->pad1);
        double sum = 0;
        for (int k = 0; k < 8; k++) sum += (double)c[k];
        fprintf(stderr, "longest sorted record: %llu cycles (%llu candidates)\n", c[8], c[9]);
        fprintf(stderr, "hit records, wave cycles by phase (%%): set-up %.1f  screen %.1f  cell masks %.1f  sort %.1f  "
                        "pass set-up %.1f  candidate loop %.1f  shade %.1f  reduce %.1f   (total %.3g cycles)\n",
                100 * c[0] / sum, 100 * c[1] / sum, 100 * c[2] / sum, 100 * c[3] / sum, 100 * c[4] / sum, 100 * c[5] / sum,
                100 * c[6] / sum, 100 * c[7] / sum, sum);
    }
#endif
#ifdef STEP_STATS
    {
        const unsigned long long *c = reinterpret_cast<const unsigned long long *>(s->h_ctr->pad1);
        fprintf(stderr, "step stats: steps %llu  all fail u %llu  all fail u|v %llu  all fail t %llu  no hit %llu  "
                        "clearly outside uv %llu  clearly behind %llu  clearly either %llu\n",
                c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
        fprintf(stderr, "classes (sorted heavy, sorted light, plain heavy, plain light): %u %u %u %u; sorted records that sort %llu; plain records not closed %llu\n",
                s->h_ctr->cls[0].count, s->h_ctr->cls[1].count, s->h_ctr->cls[2].count, s->h_ctr->cls[3].count, c[8], c[9]);
        fprintf(stderr, "grazing guard: pairs listed %llu in %llu records, records shaded without screens %llu, primitives screened %llu\n",
                c[10], c[11], c[12], c[13]);
    }
#endif
    s->n_hits = n_hits;
    if (stats) {
        stats->primary_rays = npix;
        stats->hits = n_hits;
        uint64_t leaf_prims = 0, lit = 0, tests_run = 0, bounce_hits = 0, bounce_prims = 0;
        for (uint32_t k = 0; k < STAT_SHARDS; k++) {
            leaf_prims += s->h_ctr->stat[k].leaf_prims, lit += s->h_ctr->stat[k].lit;
            tests_run += s->h_ctr->stat[k].tests_run, bounce_hits += s->h_ctr->stat[k].bounce_hits;
            bounce_prims += s->h_ctr->stat[k].bounce_prims;
        }
        stats->shadow_rays = (n_hits + bounce_hits) * p->n_lights;
        stats->shadow_tests = (leaf_prims + bounce_prims) * p->n_lights;
        stats->lit_samples = lit;
        stats->shadow_tests_run = tests_run;
        stats->render_attempts = attempts;
        stats->record_capacity = wb.rec_cap;
        if (timing) {
            // [0] tile list, trace, classify; screen_ms the candidate screens; [1] the shading kernel(s); [2] merge
            HIP_TRY(hipEventElapsedTime(&stats->kernel_ms[0], s->ev[0], s->ev[4]));
            HIP_TRY(hipEventElapsedTime(&stats->screen_ms, s->ev[4], s->ev[1]));
            for (int k = 1; k < 3; k++) HIP_TRY(hipEventElapsedTime(&stats->kernel_ms[k], s->ev[k], s->ev[k + 1]));
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

// Carves the device's static work area into the buffers of one frame: counters, one record
// head per pixel, the tile queues, and records + colours + work lists from what is left.
extern "C" rtx_status rtx_render_device_static(rtx_scene *s, const rtx_params *p, void *d_rgb, void *hip_stream,
                                               rtx_stats *stats) {
    if (!d_rgb) return fail(RTX_ERR_INVALID, "null output");
    rtx_status rs = check_params(s, p);
    if (rs != RTX_OK) return rs;
    unsigned char *base = nullptr;
    if (s->device >= 0 && s->device < SCRATCH_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        base = g_scratch_ptr[s->device];
    }
    if (!base) return fail(RTX_ERR_NO_DEVICE, "no static work area was resolved for the scene's device");
    const uint32_t rows = rtx_part_rows(p);
    const uint64_t npix = (uint64_t)rows * p->width;
    const TilePlan plan = plan_tiles(s, p, rows);
    auto round256 = [](uint64_t b) { return (b + 255u) & ~(uint64_t)255u; };
    const uint64_t ctr_bytes = round256(sizeof(DCounters)), first_bytes = round256(npix * sizeof(uint32_t)),
                   tile_bytes = round256(plan.words * sizeof(uint32_t));
    // per record: the record, its colour, its entries in the work lists, offset and length of its
    // candidate list, and STATIC_POOL_PER_RECORD entries of the candidate pool
    const uint64_t per_rec = sizeof(DRec) + sizeof(float4) + N_CLASSES * sizeof(uint32_t) + 2 * sizeof(uint32_t) +
                             STATIC_POOL_PER_RECORD * sizeof(uint32_t);
    const uint64_t fixed_bytes = ctr_bytes + first_bytes + tile_bytes;
    const uint64_t rec_cap = fixed_bytes < STATIC_SCRATCH_BYTES
                                 ? ((STATIC_SCRATCH_BYTES - fixed_bytes) / per_rec) & ~(uint64_t)(4 * REC_SHARDS - 1)
                                 : 0;
    if (rec_cap < RTX_STATIC_MIN_RECORDS)
        return fail(RTX_ERR_LIMIT, "frame too large for the static work area: " + std::to_string(npix) +
                                       " pixels leave room for " + std::to_string(rec_cap) + " hit records of " +
                                       std::to_string((uint64_t)STATIC_SCRATCH_BYTES) +
                                       " bytes (fewer than RTX_STATIC_MIN_RECORDS); use rtx_render_device");
    FrameBufs B;
    B.ctr = reinterpret_cast<DCounters *>(base);
    B.first = reinterpret_cast<uint32_t *>(base + ctr_bytes);
    B.tiles = reinterpret_cast<uint32_t *>(base + ctr_bytes + first_bytes);
    B.recs = reinterpret_cast<DRec *>(base + fixed_bytes);
    B.rec_cap = (uint32_t)std::min<uint64_t>(rec_cap, 0xFFFFFFF0u & ~(uint32_t)(4 * REC_SHARDS - 1));
    B.colors = reinterpret_cast<float4 *>(base + fixed_bytes + (uint64_t)B.rec_cap * sizeof(DRec));
    B.cand_off = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(B.colors) +
                                              (uint64_t)B.rec_cap * (sizeof(float4) + N_CLASSES * sizeof(uint32_t)));
    B.cand_n = B.cand_off + B.rec_cap;
    B.pool = B.cand_n + B.rec_cap;
    B.pool_cap = (uint32_t)std::min<uint64_t>((uint64_t)B.rec_cap * STATIC_POOL_PER_RECORD, 0xFFFFFFFFu);
    return render_frame(s, p, (uint8_t *)d_rgb, (hipStream_t)hip_stream, stats, &B);
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

extern "C" rtx_status rtx_host_alloc(uint64_t bytes, void **out) {
    if (!out) return fail(RTX_ERR_INVALID, "null argument");
    *out = nullptr;
    if (rtx_device_count() <= 0) return fail(RTX_ERR_NO_DEVICE, "no HIP device visible");
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 64, hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(RTX_ERR_NOMEM, std::string("page-locked host memory: ") + hipGetErrorString(e));
    }
    return RTX_OK;
}

extern "C" void rtx_host_free(void *p) {
    if (p) (void)hipHostFree(p);
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
