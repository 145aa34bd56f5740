// rtx_host_capi.cpp - C entry points over the C++ drop-in layer so that scripts (tests,
// bench.py) can drive exactly the code a C++ caller uses: parse, create_mesh,
// create_boxs, init_scene, flatten and render_image.
#include "../../include/rtx_dropin.hpp"

#include <cstring>
#include <string>
#include <unistd.h>

namespace {

struct HostBuilder {
    Light light;
    Boxs boxs;
    std::vector<rtx_material> materials; // registered up front to fix the numbering
    int box_depth;
    bool split = false;
    rtx_host::FlatScene flat;
    std::string error;
};

Vec3F v3(const float *p) { return Vec3F{p[0], p[1], p[2]}; }

Material to_material(const rtx_material &m) {
    Material r;
    r.IOR = m.ior;
    r.roughness = m.roughness;
    r.specularColor = v3(m.specular);
    r.diffuseColor = v3(m.diffuse);
    return r;
}

} // namespace

extern "C" {

void *rtxh_builder_new(uint32_t box_depth) {
    HostBuilder *b = new HostBuilder();
    b->box_depth = (int)box_depth;
    b->light.position = {250.f, 250.f, 800.f}; // scene.h:80-81
    b->light.color = {255.f, 255.f, 255.f};
    return b;
}

void rtxh_builder_free(void *h) { delete (HostBuilder *)h; }

const char *rtxh_error(void *h) { return ((HostBuilder *)h)->error.c_str(); }

void rtxh_set_light(void *h, const rtx_light *l) {
    HostBuilder *b = (HostBuilder *)h;
    b->light.position = v3(l->position);
    b->light.color = v3(l->color);
}

uint32_t rtxh_add_material(void *h, const rtx_material *m) {
    HostBuilder *b = (HostBuilder *)h;
    b->materials.push_back(*m);
    return (uint32_t)b->materials.size() - 1;
}

// A top-level box that create_boxs leaves alone (depth = the split limit, scene.h:52).
int rtxh_add_leaf_box(void *h, const float lb[3], const float rt[3], const rtx_prim *prims, uint32_t n) {
    HostBuilder *b = (HostBuilder *)h;
    Box bx;
    bx.lb = v3(lb);
    bx.rt = v3(rt);
    bx.depth = b->box_depth;
    for (uint32_t k = 0; k < n; k++) {
        if (prims[k].material >= b->materials.size()) return 1;
        Object o;
        std::memset(&o, 0, sizeof(o));
        o.type = (Type)prims[k].type;
        std::memcpy(&o.triangle, prims[k].g, 36);
        o.material = to_material(b->materials[prims[k].material]);
        bx.objects.push_back(o);
    }
    b->boxs.push_back(bx);
    return 0;
}

int rtxh_add_mesh(void *h, const float *verts, uint32_t nv, const uint32_t *faces, uint32_t nf,
                  const float position[3], float taille, uint32_t material) {
    HostBuilder *b = (HostBuilder *)h;
    if (material >= b->materials.size() || nv == 0) return 1;
    Mesh mesh;
    mesh.position = v3(position);
    mesh.taille = taille;
    mesh.material = to_material(b->materials[material]);
    mesh.vertices.reserve(nv);
    for (uint32_t k = 0; k < nv; k++) mesh.vertices.push_back(v3(verts + 3 * k));
    for (uint32_t k = 0; k < nf; k++) {
        for (int c = 0; c < 3; c++)
            if (faces[3 * k + c] >= nv) return 2;
        mesh.facades.push_back(Vec3F{(float)faces[3 * k], (float)faces[3 * k + 1], (float)faces[3 * k + 2]});
    }
    create_mesh(mesh, b->boxs);
    return 0;
}

int rtxh_parse_off(const char *path, float **verts, uint32_t *nv, uint32_t **faces, uint32_t *nf) {
    Mesh mesh;
    if (!parse(path, mesh)) return 1;
    *nv = (uint32_t)mesh.vertices.size();
    *nf = (uint32_t)mesh.facades.size();
    *verts = (float *)malloc((*nv ? *nv : 1) * 12);
    *faces = (uint32_t *)malloc((*nf ? *nf : 1) * 12);
    for (uint32_t k = 0; k < *nv; k++) {
        (*verts)[3 * k] = mesh.vertices[k].x;
        (*verts)[3 * k + 1] = mesh.vertices[k].y;
        (*verts)[3 * k + 2] = mesh.vertices[k].z;
    }
    for (uint32_t k = 0; k < *nf; k++) {
        (*faces)[3 * k] = (uint32_t)(int)mesh.facades[k].x;
        (*faces)[3 * k + 1] = (uint32_t)(int)mesh.facades[k].y;
        (*faces)[3 * k + 2] = (uint32_t)(int)mesh.facades[k].z;
    }
    return 0;
}
void rtxh_free(void *p) { free(p); }

// init_scene() with `dir` as the working directory (it reads meshs/*.off from there).
int rtxh_init_scene(void *h, const char *dir) {
    HostBuilder *b = (HostBuilder *)h;
    char old[4096];
    if (!getcwd(old, sizeof(old)) || chdir(dir)) return 2;
    b->boxs.clear();
    rtx_host::settings().box_depth = b->box_depth;
    const bool ok = init_scene(b->light, b->boxs);
    if (chdir(old)) return 2;
    b->split = true;
    return ok ? 0 : 1;
}

// create_boxs + flatten.  The arrays stay owned by the builder.
int rtxh_build(void *h, rtx_scene_desc *out) {
    HostBuilder *b = (HostBuilder *)h;
    if (!b->split) {
        rtx_host::settings().box_depth = b->box_depth;
        create_boxs(b->boxs);
        b->split = true;
    }
    b->flat.materials = b->materials;
    if (!rtx_host::flatten(b->light, b->boxs, b->box_depth, b->flat, &b->error)) return 1;
    *out = b->flat.desc();
    return 0;
}

// render_image() with run-time settings; path "" keeps the default out.png.  When
// n_devices > 1 the frame is split over devices[0 .. n_devices) (ids may repeat).
int rtxh_render_image(void *h, const rtx_params *p, const int *devices, uint32_t n_devices, const char *path) {
    HostBuilder *b = (HostBuilder *)h;
    rtx_host::Settings &cfg = rtx_host::settings();
    cfg.frame = *p;
    cfg.box_depth = b->box_depth;
    cfg.device = n_devices ? devices[0] : 0;
    cfg.devices.assign(devices, devices + n_devices);
    cfg.quiet = true;
    if (path && *path) cfg.output = path;
    if (!b->split) {
        create_boxs(b->boxs);
        b->split = true;
    }
    return render_image(b->light, b->boxs) ? 0 : 1;
}

// created / reused scene replicas of render_image since the process started; release != 0
// also frees the replicas it keeps.
void rtxh_replica_stats(uint64_t out[2], int release) {
    rtx_host::replica_stats(&out[0], &out[1]);
    if (release) rtx_host::release_replicas();
}

int rtxh_write_png(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h) {
    return rtx_host::write_png(path, rgb, w, h) ? 0 : 1;
}

} // extern "C"
