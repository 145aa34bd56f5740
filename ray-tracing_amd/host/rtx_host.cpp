// rtx_host.cpp - host side of the drop-in layer (include/rtx_dropin.hpp): OFF reader,
// mesh placement, box hierarchy, the committed scene, flattening into the plain arrays
// of the C ABI, and render_image on top of librtx_amd.so.
//
// The float arithmetic that decides geometry (create_mesh, create_boxs) follows the
// reference's operation order (meshs.h:1-56, bvh.h:5-61) so that the triangles and
// boxes come out bit-identical; build without -ffast-math and with -ffp-contract=off.
#include "../../include/rtx_dropin.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <thread>

// ---------------------------------------------------------------------------------------
// small float helpers (vec3.h semantics: dot left to right, norm = sqrt(dot))
// ---------------------------------------------------------------------------------------
namespace {

inline float dot3(const Vec3F &a, const Vec3F &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float norm3(const Vec3F &a) { return std::sqrt(dot3(a, a)); }
inline Vec3F add3(const Vec3F &a, const Vec3F &b) { return Vec3F{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3F div3(const Vec3F &a, float s) { return Vec3F{a.x / s, a.y / s, a.z / s}; }
inline float lesser(float a, float b) { return (b < a) ? b : a; }  // std::min
inline float greater(float a, float b) { return (a < b) ? b : a; } // std::max

void copy3(float *dst, const Vec3F &v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

} // namespace

// ---------------------------------------------------------------------------------------
// parse (meshs.h:60-125)
// ---------------------------------------------------------------------------------------
// The reference cuts each line at single blanks and converts the pieces with atoi /
// atof.  Kept behaviours: the first line must be exactly "OFF"; each coordinate is
// (float)atof(field); indices are stored as floats; the face count is read from at most
// (position of the first blank + 1) characters after that blank (meshs.h:88-91), so a
// face count with more digits than the vertex count is cut short, as in the reference.
// A file that cannot be opened reads as "not OFF" there too (the stream is never
// checked, meshs.h:74-83).
bool parse(std::string filename, Mesh &mesh) {
    std::ifstream in(filename.c_str());
    std::string line;
    std::getline(in, line);
    if (line != "OFF") {
        std::cout << "The file to read is not in OFF format." << std::endl;
        return false;
    }
    std::getline(in, line);
    int nv = 0, nf = 0;
    {
        const size_t sp = line.find(' ');
        nv = std::atoi(line.substr(0, sp == std::string::npos ? 0 : sp + 1).c_str());
        if (sp != std::string::npos) nf = std::atoi(line.substr(sp, sp + 1).c_str());
    }
    auto field_after = [](const std::string &s, size_t &pos) -> float {
        // value of the text starting at pos; then advance pos to the next blank
        float v = (float)std::atof(s.c_str() + (pos == std::string::npos ? s.size() : pos));
        if (pos != std::string::npos) pos = s.find(' ', pos + 1);
        return v;
    };
    mesh.vertices.reserve(mesh.vertices.size() + (nv > 0 ? nv : 0));
    for (int n = 0; n < nv; n++) {
        std::getline(in, line);
        Vec3F v;
        v.x = (float)std::atof(line.c_str());
        size_t pos = line.find(' ');
        v.y = field_after(line, pos);
        v.z = field_after(line, pos);
        mesh.vertices.push_back(v);
    }
    mesh.facades.reserve(mesh.facades.size() + (nf > 0 ? nf : 0));
    for (int n = 0; n < nf; n++) {
        std::getline(in, line);
        Vec3F f;
        size_t pos = line.find(' '); // skip the leading vertex count
        f.x = field_after(line, pos);
        f.y = field_after(line, pos);
        f.z = field_after(line, pos);
        mesh.facades.push_back(f);
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// create_mesh (meshs.h:1-56)
// ---------------------------------------------------------------------------------------
// scale = (largest vertex norm, measured from the file origin) / taille; the running
// float sum of the vertices / count is the centre; offset = position - centre / scale;
// the root box is min / scale + offset .. max / scale + offset of the unscaled extremes.
void create_mesh(Mesh &mesh, Boxs &boxs) {
    std::vector<Vec3F> &vs = mesh.vertices;
    Vec3F lo = vs.at(0), hi = vs.at(0), sum = vs.at(0);
    float longest = norm3(vs[0]);
    for (size_t i = 1; i < vs.size(); i++) {
        sum = add3(sum, vs[i]);
        if (norm3(vs[i]) > longest) longest = norm3(vs[i]);
        lo = Vec3F{lesser(lo.x, vs[i].x), lesser(lo.y, vs[i].y), lesser(lo.z, vs[i].z)};
        hi = Vec3F{greater(hi.x, vs[i].x), greater(hi.y, vs[i].y), greater(hi.z, vs[i].z)};
    }
    const float scale = longest / mesh.taille;
    const Vec3F centre = div3(sum, (float)vs.size());
    const Vec3F shrunk = div3(centre, scale);
    const Vec3F offset{mesh.position.x - shrunk.x, mesh.position.y - shrunk.y, mesh.position.z - shrunk.z};
    for (Vec3F &v : vs) v = div3(v, scale);

    Box root;
    root.lb = add3(div3(lo, scale), offset);
    root.rt = add3(div3(hi, scale), offset);
    root.depth = 0;
    root.objects.reserve(mesh.facades.size());
    for (const Vec3F &f : mesh.facades) {
        Object tri;
        tri.type = TRIANGLE;
        tri.material = mesh.material;
        tri.triangle.v0 = add3(vs.at((int)f.x), offset);
        tri.triangle.v1 = add3(vs.at((int)f.y), offset);
        tri.triangle.v2 = add3(vs.at((int)f.z), offset);
        root.objects.push_back(tri);
    }
    boxs.push_back(root);
}

// ---------------------------------------------------------------------------------------
// create_boxs (bvh.h:5-61, utils.h:19-45)
// ---------------------------------------------------------------------------------------
namespace {

// Split one box at the midpoint of the longer of its x / y extents (x only if strictly
// longer; z is never split).  A triangle goes low if all three vertices are below the
// plane, high if all three are above, otherwise into both children.  Children are
// stored [low, high]; recursion runs to `stop` even for empty boxes.
void split_box(Box &box, int stop) {
    if (box.depth >= stop) return;
    const float len_x = box.rt.x - box.lb.x;
    const float len_y = box.rt.y - box.lb.y;
    const bool cut_x = len_x > len_y;
    const float plane = cut_x ? box.lb.x + len_x / 2 : box.lb.y + len_y / 2;
    Box low, high;
    low.lb = box.lb;
    low.rt = box.rt;
    high.lb = box.lb;
    high.rt = box.rt;
    (cut_x ? low.rt.x : low.rt.y) = plane;
    (cut_x ? high.lb.x : high.lb.y) = plane;
    low.depth = high.depth = box.depth + 1;
    for (const Object &o : box.objects) {
        // the reference reads the triangle arm whatever the type (utils.h:19-45)
        const float a = cut_x ? o.triangle.v0.x : o.triangle.v0.y;
        const float b = cut_x ? o.triangle.v1.x : o.triangle.v1.y;
        const float c = cut_x ? o.triangle.v2.x : o.triangle.v2.y;
        const bool below = a < plane && b < plane && c < plane;
        const bool above = a > plane && b > plane && c > plane;
        if (below)
            low.objects.push_back(o);
        else if (above)
            high.objects.push_back(o);
        else {
            low.objects.push_back(o);
            high.objects.push_back(o);
        }
    }
    box.boxs.push_back(std::move(low));
    box.boxs.push_back(std::move(high));
    for (Box &child : box.boxs) split_box(child, stop);
}

} // namespace

void create_boxs(Boxs &boxs) {
    const int stop = rtx_host::settings().box_depth;
    for (Box &b : boxs) {
        if (b.depth < stop && b.boxs.empty()) split_box(b, stop);
    }
}

// ---------------------------------------------------------------------------------------
// init_scene (scene.h:4-108): white sphere + red bunny + blue dino, light at height 800
// ---------------------------------------------------------------------------------------
namespace {

Material material(float ior, float rough, Vec3F spec, Vec3F dif) {
    Material m;
    m.IOR = ior;
    m.roughness = rough;
    m.specularColor = spec;
    m.diffuseColor = dif;
    return m;
}

} // namespace

bool init_scene(Light &light, Boxs &boxs) {
    // scene.h:10-28 (mat_green and mat_nickel are defined there but unused)
    const Material red = material(1.0771f, 0.0589f, {1.f, 0.852f, 1.172f}, {0.26f, 0.036f, 0.014f});
    const Material blue = material(1.1153f, 0.068f, {1.f, 0.965f, 1.07f}, {0.012f, 0.036f, 0.106f});
    const Material white = material(1.1022f, 0.0579f, {1.f, 0.766f, 0.762f}, {0.2f, 0.2f, 0.2f});

    // scene.h:44-54: the white sphere in a box of its own that is never split
    Object ball;
    std::memset(&ball, 0, sizeof(ball));
    ball.type = SPHERE;
    ball.sphere.position = {250.f, 750.f, 500.f};
    ball.sphere.radius = 175.f;
    ball.material = white;
    Box ball_box;
    const Vec3F c = ball.sphere.position;
    const float r = ball.sphere.radius;
    ball_box.rt = {c.x + r, c.y + r, c.z + r};
    ball_box.lb = {c.x - r, c.y - r, c.z - r};
    ball_box.depth = DEPTH_BOX;
    ball_box.objects.push_back(ball);
    boxs.push_back(ball_box);

    light.position = {250.f, 250.f, 800.f}; // scene.h:80
    light.color = {255.f, 255.f, 255.f};    // scene.h:81

    // scene.h:85-101
    Mesh bunny, dino;
    bunny.position = {725.f, 725.f, 500.f};
    bunny.taille = 500.f;
    bunny.material = red;
    dino.position = {575.f, 250.f, 500.f};
    dino.taille = 500.f;
    dino.material = blue;
    if (!parse("meshs/bunny.off", bunny)) return false;
    create_mesh(bunny, boxs);
    if (!parse("meshs/dino.off", dino)) return false;
    create_mesh(dino, boxs);

    create_boxs(boxs); // scene.h:105
    return true;
}

// ---------------------------------------------------------------------------------------
// settings, flattening
// ---------------------------------------------------------------------------------------
namespace rtx_host {

Settings::Settings() { rtx_params_default(&frame); }

Settings &settings() {
    static Settings s;
    return s;
}

rtx_scene_desc FlatScene::desc() const {
    rtx_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_nodes = (uint32_t)nodes.size();
    d.n_leaves = (uint32_t)leaves.size();
    d.n_prims = (uint32_t)prims.size();
    d.n_materials = (uint32_t)materials.size();
    d.nodes = nodes.data();
    d.leaves = leaves.data();
    d.prims = prims.data();
    d.materials = materials.data();
    d.light = light;
    return d;
}

namespace {

std::string geometry_key(const Object &o) {
    std::string k((const char *)&o.type, sizeof(o.type));
    k.append((const char *)&o.triangle, sizeof(o.triangle));
    return k;
}

struct Flattener {
    FlatScene &out;
    int box_depth;
    // An Object carries no identity.  A primitive's id is its position in the
    // concatenated top-level object lists; copies further down are matched to it by
    // geometry (the n-th copy with a given geometry inside a leaf is the n-th
    // top-level primitive with that geometry: splitting keeps relative order).
    std::map<std::string, std::vector<uint32_t>> ids;

    uint32_t material_index(const Material &m) {
        for (size_t k = 0; k < out.materials.size(); k++)
            if (!std::memcmp(&out.materials[k], &m, sizeof(Material))) return (uint32_t)k;
        rtx_material r;
        static_assert(sizeof(rtx_material) == sizeof(Material), "material layouts must match");
        std::memcpy(&r, &m, sizeof(r));
        out.materials.push_back(r);
        return (uint32_t)out.materials.size() - 1;
    }

    void walk(const Boxs &list) {
        for (const Box &bx : list) {
            const size_t me = out.nodes.size();
            rtx_node nd;
            copy3(nd.lb, bx.lb);
            copy3(nd.rt, bx.rt);
            nd.skip = 0;
            nd.leaf = RTX_NONE;
            out.nodes.push_back(nd);
            if (bx.depth < box_depth) {
                walk(bx.boxs);
            } else {
                rtx_leaf lf;
                lf.first = (uint32_t)out.prims.size();
                lf.count = (uint32_t)bx.objects.size();
                std::map<std::string, uint32_t> seen;
                for (const Object &o : bx.objects) {
                    rtx_prim p;
                    std::memset(&p, 0, sizeof(p));
                    p.type = (uint32_t)o.type;
                    std::memcpy(p.g, &o.triangle, o.type == TRIANGLE ? 36 : 16);
                    p.material = material_index(o.material);
                    const std::string key = geometry_key(o);
                    const uint32_t nth = seen[key]++;
                    const std::vector<uint32_t> &v = ids[key];
                    p.id = nth < v.size() ? v[nth] : (v.empty() ? RTX_NONE : v.back());
                    out.prims.push_back(p);
                }
                out.nodes[me].leaf = (uint32_t)out.leaves.size();
                out.leaves.push_back(lf);
            }
            out.nodes[me].skip = (uint32_t)out.nodes.size();
        }
    }
};

} // namespace

bool flatten(const Light &light, const Boxs &boxs, int box_depth, FlatScene &out, std::string *error) {
    out.nodes.clear();
    out.leaves.clear();
    out.prims.clear();
    Flattener f{out, box_depth, {}};
    uint32_t next = 0;
    for (const Box &b : boxs)
        for (const Object &o : b.objects) f.ids[geometry_key(o)].push_back(next++);
    f.walk(boxs);
    copy3(out.light.position, light.position);
    copy3(out.light.color, light.color);
    if (out.leaves.size() > RTX_MAX_LEAVES) {
        if (error) *error = "more than 65535 leaf boxes";
        return false;
    }
    for (const rtx_prim &p : out.prims) {
        if (p.type != RTX_PRIM_SPHERE && p.type != RTX_PRIM_TRIANGLE) {
            if (error) *error = "object with an unknown type";
            return false;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// rendering
// ---------------------------------------------------------------------------------------
// Scene replicas kept between calls (see rtx_dropin.hpp).  The caller is single-threaded,
// as the reference's main() is; the worker threads below each touch only their own slot.
namespace {

struct Replica {
    rtx_scene *scene = nullptr;
    int device = -1;
    uint8_t *staging = nullptr; // page-locked, for the packed rows of this replica's part
    size_t staging_bytes = 0;
};

struct ReplicaCache {
    uint64_t key = 0;
    FlatScene content; // what the replicas were uploaded from (a hash match is confirmed on it)
    bool valid = false;
    std::vector<Replica> slots;
    uint64_t created = 0, reused = 0;
} g_replicas;

uint64_t fnv1a(uint64_t h, const void *data, size_t n) {
    const uint8_t *p = static_cast<const uint8_t *>(data);
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}

uint64_t content_key(const FlatScene &f) {
    uint64_t h = 0xCBF29CE484222325ull;
    h = fnv1a(h, f.nodes.data(), f.nodes.size() * sizeof(rtx_node));
    h = fnv1a(h, f.leaves.data(), f.leaves.size() * sizeof(rtx_leaf));
    h = fnv1a(h, f.prims.data(), f.prims.size() * sizeof(rtx_prim));
    h = fnv1a(h, f.materials.data(), f.materials.size() * sizeof(rtx_material));
    return fnv1a(h, &f.light, sizeof(f.light));
}

// Element-wise equality of two flattened scenes (the structs are made of 4-byte fields
// without padding, so comparing their bytes compares their fields).
template <typename T>
bool same_array(const std::vector<T> &a, const std::vector<T> &b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
}
bool same_content(const FlatScene &a, const FlatScene &b) {
    static_assert(sizeof(rtx_node) == 32 && sizeof(rtx_leaf) == 8 && sizeof(rtx_prim) == 48 &&
                      sizeof(rtx_material) == 32 && sizeof(rtx_light) == 24,
                  "no padding in the flattened scene's structs");
    return same_array(a.nodes, b.nodes) && same_array(a.leaves, b.leaves) && same_array(a.prims, b.prims) &&
           same_array(a.materials, b.materials) && std::memcmp(&a.light, &b.light, sizeof(rtx_light)) == 0;
}

void free_slot(Replica &r) {
    if (r.scene) rtx_scene_destroy(r.scene);
    if (r.staging) rtx_host_free(r.staging);
    r = Replica();
}

// The replica of slot k on `device` for the scene `d`; uploads it if the slot is empty or
// holds another device's.  Returns "" or an error message.
std::string get_replica(size_t k, int device, const rtx_scene_desc &d, Replica **out) {
    if (g_replicas.slots.size() <= k) g_replicas.slots.resize(k + 1);
    Replica &r = g_replicas.slots[k];
    if (r.scene && r.device != device) free_slot(r);
    if (!r.scene) {
        if (rtx_scene_create(&d, device, &r.scene) != RTX_OK) {
            r.scene = nullptr;
            return rtx_last_error();
        }
        r.device = device;
        g_replicas.created++;
    } else {
        g_replicas.reused++;
    }
    *out = &r;
    return "";
}

} // namespace

void replica_stats(uint64_t *created, uint64_t *reused) {
    if (created) *created = g_replicas.created;
    if (reused) *reused = g_replicas.reused;
}

void release_replicas() {
    for (Replica &r : g_replicas.slots) free_slot(r);
    g_replicas.slots.clear();
    g_replicas.content = FlatScene();
    g_replicas.valid = false;
}

bool render_to_memory(const Light &light, const Boxs &boxs, std::vector<uint8_t> &rgb, rtx_stats *stats,
                      std::string *error) {
    Settings &cfg = settings();
    FlatScene flat;
    if (!flatten(light, boxs, cfg.box_depth, flat, error)) return false;
    const rtx_scene_desc d = flat.desc();
    // another scene than last time: the replicas are of no use any more (the hash only
    // saves the comparison when the scenes differ; a match is confirmed array by array)
    const uint64_t key = content_key(flat);
    if (!g_replicas.valid || g_replicas.key != key || !same_content(flat, g_replicas.content)) {
        release_replicas();
        g_replicas.key = key;
        g_replicas.content = flat;
        g_replicas.valid = true;
    }
    // (keep_replicas = false: this call's replicas are released before it returns)
    struct ReleaseAtExit {
        bool on;
        ~ReleaseAtExit() {
            if (on) release_replicas();
        }
    } release_at_exit{!cfg.keep_replicas};
    rtx_params p = cfg.frame;
    p.stripe_rows = p.n_parts = p.part = 0;
    rgb.assign((size_t)p.width * p.height * 3, 0);
    if (cfg.devices.size() <= 1) {
        Replica *r = nullptr;
        std::string err = get_replica(0, cfg.devices.empty() ? cfg.device : cfg.devices[0], d, &r);
        if (err.empty() && rtx_render(r->scene, &p, rgb.data(), stats) != RTX_OK) err = rtx_last_error();
        if (!err.empty() && error) *error = err;
        return err.empty();
    }
    // Row stripes dealt round-robin to the devices: stripe s goes to part s % n_parts.
    // Every part renders its rows, packed, into page-locked memory of its own and its host
    // thread then copies them to their place in the frame (the parts' rows are disjoint).
    // The sampler is keyed by the global pixel index, so the frame does not depend on the
    // split.
    const uint32_t parts = (uint32_t)cfg.devices.size();
    const uint32_t stripe = cfg.stripe_rows ? cfg.stripe_rows : 8;
    std::vector<Replica *> replica(parts, nullptr);
    if (g_replicas.slots.size() < parts) g_replicas.slots.resize(parts); // (before any pointer into it is taken)
    for (uint32_t k = 0; k < parts; k++) { // (uploads happen here, on the caller's thread)
        const std::string err = get_replica(k, cfg.devices[k], d, &replica[k]);
        if (!err.empty()) {
            if (error) *error = err;
            return false;
        }
    }
    std::vector<std::string> errors(parts);
    std::vector<rtx_stats> part_stats(parts);
    std::vector<std::thread> workers;
    for (uint32_t k = 0; k < parts; k++) {
        workers.emplace_back([&, k]() {
            rtx_params q = p;
            q.stripe_rows = stripe;
            q.n_parts = parts;
            q.part = k;
            Replica &r = *replica[k];
            const size_t row_bytes = (size_t)q.width * 3, need = (size_t)rtx_part_rows(&q) * row_bytes;
            if (need == 0) return;
            if (r.staging_bytes < need) {
                if (r.staging) rtx_host_free(r.staging);
                r.staging = nullptr;
                r.staging_bytes = 0;
                void *mem = nullptr;
                if (rtx_host_alloc(need, &mem) != RTX_OK) {
                    errors[k] = rtx_last_error();
                    return;
                }
                r.staging = static_cast<uint8_t *>(mem);
                r.staging_bytes = need;
            }
            if (rtx_render(r.scene, &q, r.staging, &part_stats[k]) != RTX_OK) {
                errors[k] = rtx_last_error();
                return;
            }
            size_t src = 0;
            for (uint32_t j0 = k * stripe; j0 < q.height; j0 += parts * stripe) { // this part's stripes
                const size_t rows = std::min<uint32_t>(stripe, q.height - j0);
                std::memcpy(rgb.data() + (size_t)j0 * row_bytes, r.staging + src, rows * row_bytes);
                src += rows * row_bytes;
            }
        });
    }
    for (std::thread &t : workers) t.join();
    if (stats) std::memset(stats, 0, sizeof(*stats));
    for (uint32_t k = 0; k < parts; k++) {
        if (!errors[k].empty()) {
            if (error) *error = errors[k];
            return false;
        }
        if (stats) {
            stats->primary_rays += part_stats[k].primary_rays;
            stats->hits += part_stats[k].hits;
            stats->shadow_rays += part_stats[k].shadow_rays;
            stats->shadow_tests += part_stats[k].shadow_tests;
            stats->lit_samples += part_stats[k].lit_samples;
            stats->shadow_tests_run += part_stats[k].shadow_tests_run;
            stats->render_attempts = std::max(stats->render_attempts, part_stats[k].render_attempts);
            stats->record_capacity += part_stats[k].record_capacity;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// image files
// ---------------------------------------------------------------------------------------
namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}

void put32(std::vector<uint8_t> &v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}

void chunk(std::vector<uint8_t> &file, const char tag[4], const std::vector<uint8_t> &body) {
    put32(file, (uint32_t)body.size());
    const size_t start = file.size();
    file.insert(file.end(), tag, tag + 4);
    file.insert(file.end(), body.begin(), body.end());
    put32(file, ~crc32_update(0xFFFFFFFFu, file.data() + start, file.size() - start));
}

} // namespace

bool write_png(const std::string &path, const uint8_t *rgb, uint32_t width, uint32_t height) {
    // raw scanlines, top image row first = our last row; filter byte 0 each
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + (size_t)width * 3));
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);
        const uint8_t *row = rgb + (size_t)(height - 1 - y) * width * 3;
        raw.insert(raw.end(), row, row + (size_t)width * 3);
    }
    // zlib stream of stored blocks
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    size_t pos = 0;
    uint32_t a = 1, b = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF));
        z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        pos += n;
    } while (pos < raw.size());
    put32(z, (b << 16) | a);

    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr;
    put32(ihdr, width);
    put32(ihdr, height);
    ihdr.push_back(8); // bit depth
    ihdr.push_back(2); // colour type RGB
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    chunk(file, "IHDR", ihdr);
    chunk(file, "IDAT", z);
    chunk(file, "IEND", {});
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
    return (std::fclose(f) == 0) && ok;
}

bool write_ppm(const std::string &path, const uint8_t *rgb, uint32_t width, uint32_t height) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    bool ok = true;
    for (uint32_t y = 0; y < height && ok; y++)
        ok = std::fwrite(rgb + (size_t)(height - 1 - y) * width * 3, 1, (size_t)width * 3, f) == (size_t)width * 3;
    return (std::fclose(f) == 0) && ok;
}

} // namespace rtx_host

// ---------------------------------------------------------------------------------------
// render_image (raytracing.h:100-145)
// ---------------------------------------------------------------------------------------
// Renders the frame on the GPU and writes it to settings().output (out.png by
// default).  Returns true on success.  Unlike the reference it does not leave the last
// random sample in light.position, and a failure to allocate returns false (the
// reference returns `1`, i.e. true, there: raytracing.h:107-108).
bool render_image(Light &light, Boxs &boxs) {
    rtx_host::Settings &cfg = rtx_host::settings();
    std::vector<uint8_t> rgb;
    std::string error;
    if (!rtx_host::render_to_memory(light, boxs, rgb, nullptr, &error)) {
        std::fprintf(stderr, "render_image: %s\n", error.c_str());
        return false;
    }
    // raytracing.h:132-135 prints "Rendering image... N %" after every tenth row, N = rows
    // done / 10, i.e. the percentage of its 1000-row frame.  Here the rows arrive all at once:
    // the same lines, one per tenth row, N the percentage of this frame's rows.
    if (!cfg.quiet)
        for (uint32_t j = 0; j < cfg.frame.height; j++)
            if ((j + 1) % 10 == 0)
                std::printf("Rendering image... %u %%\r", (unsigned)(((uint64_t)(j + 1) * 100) / cfg.frame.height));
    const std::string &path = cfg.output;
    const bool ppm = path.size() > 4 && path.compare(path.size() - 4, 4, ".ppm") == 0;
    const bool saved = ppm ? rtx_host::write_ppm(path, rgb.data(), cfg.frame.width, cfg.frame.height)
                           : rtx_host::write_png(path, rgb.data(), cfg.frame.width, cfg.frame.height);
    if (saved && !cfg.quiet) std::printf("\nImage successfully saved!\n"); // raytracing.h:141
    if (!saved) std::fprintf(stderr, "render_image: cannot write %s\n", path.c_str());
    return saved;
}
