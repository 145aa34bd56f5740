// rtx_dropin.hpp - the reference's C++ scene / mesh / render entry points, backed by the
// MI355X library (include/rtx.h).
//
// A program written against the reference includes "raytracing.h" and calls
//     bool init_scene(Light &, Boxs &);      (scene.h:4)
//     bool render_image(Light &, Boxs &);    (raytracing.h:100)
// and, when it assembles its own scene,
//     bool parse(std::string, Mesh &);       (meshs.h:60)
//     void create_mesh(Mesh &, Boxs &);      (meshs.h:1)
//     void create_boxs(Boxs &);              (bvh.h:5)
// on the plain structs of structs.h / vec3.h.  Including this header instead and
// linking librtx_host.so + librtx_amd.so keeps those names, argument meanings and the
// bool (true = success) error convention; the pixel loop runs on the GPU.
//
// The data types below have the reference's member names and layout (they are the API);
// the functions are implemented from scratch in ray-tracing_amd/host/.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "rtx.h"

// ---- vec3.h ---------------------------------------------------------------------------
template <typename T>
struct Vec3 {
    T x, y, z;
};
using Vec3F = Vec3<float>;

// The operators and free functions of vec3.h:9-112, for callers that assemble their own
// scene with them.  Evaluation order is part of the contract (the products of dot are added
// left to right, normalize is three divisions by the norm, reflect is I - N * 2.f * dot(N, I)
// with the scalings applied one after the other): compiled like the reference (no
// -ffast-math, no contraction) they return the same bits.  `<` and `>` hold when ALL three
// components compare that way (a partial order; vec3.h:91-101).
template <typename T> inline float dot(const Vec3<T> &a, const Vec3<T> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> inline float dot2(const Vec3<T> &v) { return dot(v, v); }
template <typename T> inline float norm(const Vec3<T> &v) { return std::sqrt(dot2(v)); }
template <typename T> inline bool operator==(const Vec3<T> &a, const Vec3<T> &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
template <typename T> inline bool operator!=(const Vec3<T> &a, const Vec3<T> &b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
template <typename T> inline Vec3<T> operator+(const Vec3<T> &a, const Vec3<T> &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> inline Vec3<T> operator-(const Vec3<T> &a, const Vec3<T> &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> inline Vec3<T> operator*(const Vec3<T> &a, const Vec3<T> &b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename T> inline Vec3<T> operator/(const Vec3<T> &a, const Vec3<T> &b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
template <typename T> inline Vec3<T> operator*(const Vec3<T> &a, const float &s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> inline Vec3<T> operator/(const Vec3<T> &a, const float &s) { return {a.x / s, a.y / s, a.z / s}; }
template <typename T> inline bool operator>(const Vec3<T> &a, const Vec3<T> &b) { return a.x > b.x && a.y > b.y && a.z > b.z; }
template <typename T> inline bool operator<(const Vec3<T> &a, const Vec3<T> &b) { return a.x < b.x && a.y < b.y && a.z < b.z; }
template <typename T> inline Vec3<T> normalize(const Vec3<T> &v) { return v / norm(v); }
template <typename T> inline Vec3<T> cross(const Vec3<T> &u, const Vec3<T> &v) {
    return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
inline Vec3F reflect(const Vec3F &I, const Vec3F &N) { return I - N * 2.f * dot(N, I); }
template <typename T> inline void print(const Vec3<T> &v) { std::printf("%.1f %.1f %.1f\n", v.x, v.y, v.z); }

// ---- structs.h --------------------------------------------------------------------------
enum Type { SPHERE = 1, TRIANGLE };

struct Material {
    float IOR;
    float roughness;
    Vec3F specularColor;
    Vec3F diffuseColor;
};

struct Object {
    Type type;
    union {
        struct {
            Vec3F position;
            float radius;
        } sphere;
        struct {
            Vec3F v0, v1, v2;
        } triangle;
    };
    Material material;
};

struct Ray {
    Vec3F origin;
    Vec3F direction;
    int depth;
};

struct Light {
    Vec3F position;
    Vec3F color;
};

// What the reference's intersect_* functions fill in (structs.h:45-51).  The GPU path keeps
// its hits in its own records; the type is here for callers that name it.
struct Intersection {
    Object object;
    Vec3F position;
    float distance;
    Vec3F normale;
};

struct Mesh {
    Vec3F position;
    float taille;
    Material material;
    std::vector<Vec3F> vertices;
    std::vector<Vec3F> facades; // vertex indices kept as floats, as the reference does
};

struct Box {
    Vec3F lb, rt;
    std::vector<Box> boxs;
    int depth;
    std::vector<Object> objects;
};

typedef std::vector<Box> Boxs;
typedef std::vector<Object> Objects;
typedef std::vector<Intersection> Intersections;

// ---- settings the reference fixes with #define ---------------------------------------------
#ifndef DEPTH_BOX
#define DEPTH_BOX 10 // bvh.h:3
#endif

namespace rtx_host {

// Run-time counterparts of WIDTH, HEIGHT, NB_LIGHTS (raytracing.h:12-19), the camera
// (raytracing.h:112), acne (raytracing.h:21) and DEPTH_BOX; the defaults are the
// reference's values.  render_image() and create_boxs() read them.
struct Settings {
    rtx_params frame;               // width, height, n_lights, seed, acne, camera
    int box_depth = DEPTH_BOX;      // depth at which create_boxs stops splitting
    int device = 0;                 // HIP device used by render_image
    std::vector<int> devices;       // if not empty: split the frame over these devices
                                    // (row stripes, one host thread and one scene replica
                                    // per entry; an id may repeat)
    uint32_t stripe_rows = 8;       // rows per stripe of that split
    std::string output = "out.png"; // raytracing.h:140
    bool quiet = false;             // suppress the progress / "saved" messages
    bool keep_replicas = true;      // keep the uploaded scene between render calls (see
                                    // replica_stats); false = free it before the call returns,
                                    // as the reference holds nothing after render_image
    Settings();
};
Settings &settings();

// Flattened form of a `Boxs` tree (what rtx_scene_create takes).  The arrays live in
// the object; desc() points into them.
struct FlatScene {
    std::vector<rtx_node> nodes;
    std::vector<rtx_leaf> leaves;
    std::vector<rtx_prim> prims;
    std::vector<rtx_material> materials;
    rtx_light light;
    rtx_scene_desc desc() const;
};

// Walks `boxs` in the order trace_ray_boxs does (raytracing.h:74-98).  Primitive ids
// are positions in the concatenated top-level object lists; `materials` may be
// pre-seeded to fix the material numbering.  Returns false (and a message in *error)
// if the tree cannot be expressed (more than 65535 leaves, ...).
bool flatten(const Light &light, const Boxs &boxs, int box_depth, FlatScene &out, std::string *error);

// Render into memory instead of a file: rgb receives width * height * 3 bytes, pixel
// (i, j) at (j * width + i) * 3.  Returns false on failure (message in *error).
bool render_to_memory(const Light &light, const Boxs &boxs, std::vector<uint8_t> &rgb,
                      rtx_stats *stats, std::string *error);

// render_image / render_to_memory keep what they uploaded: one scene replica per entry of
// settings().devices (or one on settings().device), reused by later calls as long as the
// flattened scene has the same content (compared array by array with a copy kept on the
// host), so that a caller rendering the same `Boxs` again (other frame size, sample count,
// camera ...) pays neither the conversion to the device layout nor the upload again.  What
// stays allocated until release_replicas() or the end of the process: per device the scene
// (64 B per leaf primitive reference), the work buffers of its largest frame (96 B per hit
// record, 4 B per pixel) and the page-locked rows of a split frame; on the host one copy of
// the flattened scene (48 B per leaf primitive reference).  settings().keep_replicas = false
// turns the keeping off.  created / reused count replicas since the process started;
// release_replicas() frees them (call it before the process ends if the HIP runtime must
// see every free).
void replica_stats(uint64_t *created, uint64_t *reused);
void release_replicas();

// Minimal PNG (8-bit RGB, stored deflate blocks) and PPM writers.  Row j = 0 of `rgb`
// is written as the BOTTOM scanline, as FreeImage_SetPixelColor places it
// (raytracing.h:130).
bool write_png(const std::string &path, const uint8_t *rgb, uint32_t width, uint32_t height);
bool write_ppm(const std::string &path, const uint8_t *rgb, uint32_t width, uint32_t height);

} // namespace rtx_host

// ---- the reference's entry points ---------------------------------------------------------
bool parse(std::string filename, Mesh &mesh);
void create_mesh(Mesh &mesh, Boxs &boxs);
void create_boxs(Boxs &boxs);
bool init_scene(Light &light, Boxs &boxs);
bool render_image(Light &light, Boxs &boxs);
