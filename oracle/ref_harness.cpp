/*
 * ref_harness.cpp - test-only shim that compiles the REFERENCE's own headers,
 * included from where they lie under /root/reference (never copied), and
 * exposes their functions through a C ABI so that tests can (1) pin the CPU
 * oracle against the reference function by function, (2) flatten the `Boxs`
 * tree that the reference's create_mesh / create_boxs build and compare it
 * with the oracle's, and (3) render frames by composing the reference's
 * intersect_box / intersect_scene / shade / coord_out_of_scene /
 * clamp_color / Vec3 operator< on the reference's own `Boxs`.
 *
 * raytracing.h itself cannot be compiled here (it includes the FreeImage
 * header, which the image lacks), so the ~60 lines of control flow of
 * trace_ray / trace_ray_boxs / render_image (raytracing.h:29-136) are
 * re-expressed below around the reference's functions, with the
 * counter-based light sampler in place of std::random_device.
 *
 * Output: oracle/_ref/libref.so (git-ignored).  Used by tests and by
 * tests/golden/make_golden.py only.
 */
#include <ctime>
#include <fstream>
#include <iostream>
#include <random>
#include <math.h>
#include <cstring>
#include <map>
#include <string>
#include <unistd.h>
#include <vector>

/* the reference's include order (main.cpp:1-5, raytracing.h:1-7) */
#include "vec3.h"
#include "shade.h"
#include "scene.h"
#include "intersect.h"

#include "rtx.h"

namespace {

/* local copy of the sampler so that libref.so stands alone */
void light_sample(uint32_t seed, uint32_t pixel, uint32_t leaf, uint32_t k, float xy[2]) {
    uint32_t c0 = pixel, c1 = (leaf << 16) | k, key = seed;
    for (int r = 0; r < 10; r++) {
        uint64_t prod = (uint64_t)0xD256D193u * c0;
        uint32_t hi = (uint32_t)(prod >> 32), lo = (uint32_t)prod;
        c0 = hi ^ key ^ c1;
        c1 = lo;
        key += 0x9E3779B9u;
    }
    xy[0] = 250.f + (float)(uint32_t)(((uint64_t)c0 * 501u) >> 32);
    xy[1] = 250.f + (float)(uint32_t)(((uint64_t)c1 * 501u) >> 32);
}

Vec3F V(const float *p) { return Vec3F{p[0], p[1], p[2]}; }
void S(float *p, const Vec3F &v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
Material M(const rtx_material *m) {
    Material r;
    r.IOR = m->ior;
    r.roughness = m->roughness;
    r.specularColor = V(m->specular);
    r.diffuseColor = V(m->diffuse);
    return r;
}

std::string geom_key(const Object &o) {
    /* type + the 36 bytes of the union */
    std::string k((const char *)&o.type, sizeof(o.type));
    k.append((const char *)&o.triangle, sizeof(o.triangle));
    return k;
}

struct Builder {
    Light light;
    Boxs boxs;
    std::vector<rtx_material> mats;
    bool built = false;
    /* flattened */
    std::vector<rtx_node> nodes;
    std::vector<rtx_leaf> leaves;
    std::vector<rtx_prim> prims;
    /* geometry -> ids of the top-level primitives that have it, in order */
    std::map<std::string, std::vector<uint32_t>> ids;
    /* leaf Box* -> leaf index, for the composed render */
    std::map<const Box *, uint32_t> leaf_index;
};

uint32_t material_index(Builder *b, const Material &m) {
    for (size_t k = 0; k < b->mats.size(); k++)
        if (!memcmp(&b->mats[k], &m, sizeof(Material))) return (uint32_t)k;
    rtx_material r;
    memcpy(&r, &m, sizeof(r));
    b->mats.push_back(r);
    return (uint32_t)b->mats.size() - 1;
}

void flatten(Builder *b, Boxs &list) {
    for (size_t k = 0; k < list.size(); k++) {
        Box &bx = list[k];
        uint32_t me = (uint32_t)b->nodes.size();
        rtx_node nd;
        S(nd.lb, bx.lb);
        S(nd.rt, bx.rt);
        nd.skip = 0;
        nd.leaf = RTX_NONE;
        b->nodes.push_back(nd);
        if (bx.depth < DEPTH_BOX) {
            flatten(b, bx.boxs);
        } else {
            rtx_leaf lf;
            lf.first = (uint32_t)b->prims.size();
            lf.count = (uint32_t)bx.objects.size();
            std::map<std::string, uint32_t> seen;
            for (size_t o = 0; o < bx.objects.size(); o++) {
                const Object &ob = bx.objects[o];
                rtx_prim p;
                memset(&p, 0, sizeof(p));
                p.type = (uint32_t)ob.type;
                if (ob.type == TRIANGLE) {
                    memcpy(p.g, &ob.triangle, 36);
                } else {
                    memcpy(p.g, &ob.sphere, 16);
                }
                p.material = material_index(b, ob.material);
                std::string key = geom_key(ob);
                uint32_t occ = seen[key]++;
                const std::vector<uint32_t> &v = b->ids[key];
                p.id = occ < v.size() ? v[occ] : (v.empty() ? RTX_NONE : v.back());
                b->prims.push_back(p);
            }
            b->nodes[me].leaf = (uint32_t)b->leaves.size();
            b->leaf_index[&bx] = (uint32_t)b->leaves.size();
            b->leaves.push_back(lf);
        }
        b->nodes[me].skip = (uint32_t)b->nodes.size();
    }
}

struct Ctx {
    Builder *b;
    const rtx_params *p;
    uint32_t pixel;
    uint32_t ordinal;
    std::vector<rtx_hit> *hits;
    uint64_t slab, leaf_visits, nhits, shadow, shaded;
};

/* raytracing.h:29-72 around the reference's intersect_scene / shade /
 * coord_out_of_scene (MAX_DEPTH = 0). */
bool leaf_trace(Ctx &c, Light &light, Ray &ray, Box &box, Vec3F &out, bool record = true) {
    Intersection hit;
    if (!intersect_scene(ray, box, hit)) return false;
    uint32_t leaf = c.b->leaf_index[&box];
    /* which reference of the leaf won: first object with that geometry */
    uint32_t prim = RTX_NONE;
    {
        const rtx_leaf &lf = c.b->leaves[leaf];
        std::string key = geom_key(hit.object);
        for (uint32_t k = 0; k < lf.count; k++) {
            if (geom_key(box.objects[k]) == key) {
                prim = c.b->prims[lf.first + k].id;
                break;
            }
        }
    }
    Vec3F sum = {0.f, 0.f, 0.f};
    float dist = hit.distance;
    for (uint32_t k = 0; k < c.p->n_lights; k++) {
        float xy[2];
        light_sample(c.p->seed, c.pixel, leaf, (uint32_t)ray.depth * c.p->n_lights + k, xy);
        light.position.x = xy[0];
        light.position.y = xy[1];
        hit.position = hit.position + (hit.normale * c.p->acne);
        Ray sray;
        sray.origin = hit.position;
        sray.direction = normalize(light.position - hit.position);
        sray.depth = 0;
        Intersection block;
        c.shadow++;
        if (!intersect_scene(sray, box, block) || coord_out_of_scene(block.position)) {
            Vec3F inv = ray.direction * -1.f;
            sum = sum + (shade(hit.normale, inv, sray.direction, light.color, hit.object.material) /
                         (float)c.p->n_lights * 20.f);
            c.shaded++;
        }
    }
    /* raytracing.h:51-61 with the reference's reflect() and RDM_Fresnel() */
    if ((uint32_t)ray.depth < c.p->max_depth) {
        Ray bounce;
        Vec3F dir = reflect(ray.direction, hit.normale);
        dir = normalize(dir);
        bounce.origin = hit.position + dir * c.p->acne;
        bounce.direction = dir;
        bounce.depth = ray.depth + 1;
        Vec3F far = {0.f, 0.f, 0.f};
        if (!leaf_trace(c, light, bounce, box, far, false)) far = Vec3F{0.f, 0.f, 0.f};
        sum = sum + hit.object.material.specularColor *
                        RDM_Fresnel(dot(bounce.direction, hit.normale), 1.f, hit.object.material.IOR) * far;
    }
    out = sum;
    if (record) {
        rtx_hit h;
        h.pixel = c.pixel;
        h.ordinal = c.ordinal++;
        h.leaf = leaf;
        h.prim = prim;
        h.t = dist;
        S(h.color, sum);
        if (c.hits) c.hits->push_back(h);
        c.nhits++;
    }
    return true;
}

/* raytracing.h:74-98 */
Vec3F walk(Ctx &c, Light &light, Ray &ray, Boxs &list) {
    Vec3F best = {0.f, 0.f, 0.f};
    for (size_t k = 0; k < list.size(); k++) {
        Intersection tmp;
        c.slab++;
        if (!intersect_box(ray, list[k], tmp)) continue;
        Vec3F got = {0.f, 0.f, 0.f};
        if (list[k].depth < DEPTH_BOX) {
            got = walk(c, light, ray, list[k].boxs);
        } else {
            c.leaf_visits++;
            if (!leaf_trace(c, light, ray, list[k], got)) got = Vec3F{0.f, 0.f, 0.f};
        }
        if (best < got) best = got;
    }
    return best;
}

} // namespace

extern "C" {

int ref_depth_box(void) { return DEPTH_BOX; }
float ref_pi(void) { return PI; }

/* ---- function-level known answers ------------------------------------------ */
int ref_intersect_sphere(const float o[3], const float d[3], const float c[3], float r, float *t,
                         float pos[3], float nrm[3]) {
    Ray ray;
    ray.origin = V(o);
    ray.direction = V(d);
    ray.depth = 0;
    Object ob;
    memset(&ob, 0, sizeof(ob));
    ob.type = SPHERE;
    ob.sphere.position = V(c);
    ob.sphere.radius = r;
    Intersection in;
    memset(&in, 0, sizeof(in));
    if (!intersect_sphere(ray, ob, in)) return 0;
    *t = in.distance;
    S(pos, in.position);
    S(nrm, in.normale);
    return 1;
}

int ref_intersect_triangle(const float o[3], const float d[3], const float v0[3],
                           const float v1[3], const float v2[3], float *t, float pos[3],
                           float nrm[3]) {
    Ray ray;
    ray.origin = V(o);
    ray.direction = V(d);
    ray.depth = 0;
    Object ob;
    memset(&ob, 0, sizeof(ob));
    ob.type = TRIANGLE;
    ob.triangle.v0 = V(v0);
    ob.triangle.v1 = V(v1);
    ob.triangle.v2 = V(v2);
    Intersection in;
    memset(&in, 0, sizeof(in));
    if (!intersect_triangle(ray, ob, in)) return 0;
    *t = in.distance;
    S(pos, in.position);
    S(nrm, in.normale);
    return 1;
}

int ref_intersect_box(const float o[3], const float d[3], const float lb[3], const float rt[3]) {
    Ray ray;
    ray.origin = V(o);
    ray.direction = V(d);
    ray.depth = 0;
    Box bx;
    bx.lb = V(lb);
    bx.rt = V(rt);
    bx.depth = 0;
    Intersection in;
    return intersect_box(ray, bx, in) ? 1 : 0;
}

void ref_shade(const float n[3], const float v[3], const float l[3], const float lc[3],
               const rtx_material *m, float out[3]) {
    Material mat = M(m);
    S(out, shade(V(n), V(v), V(l), V(lc), mat));
}

uint8_t ref_clamp_color(float v) { return clamp_color(v); }
int ref_coord_out_of_scene(const float c[3]) { return coord_out_of_scene(V(c)) ? 1 : 0; }
void ref_normalize(const float a[3], float out[3]) { S(out, normalize(V(a))); }
void ref_reflect(const float i[3], const float n[3], float out[3]) { S(out, reflect(V(i), V(n))); }
float ref_fresnel(float ldoth, float ext, float in) { return RDM_Fresnel(ldoth, ext, in); }

/* ---- scene ----------------------------------------------------------------- */
void *ref_builder_new(void) {
    Builder *b = new Builder();
    b->light.position = Vec3F{250.f, 250.f, 800.f};
    b->light.color = Vec3F{255.f, 255.f, 255.f};
    return b;
}
void ref_builder_free(void *h) { delete (Builder *)h; }

uint32_t ref_add_material(void *h, const rtx_material *m) {
    Builder *b = (Builder *)h;
    b->mats.push_back(*m);
    return (uint32_t)b->mats.size() - 1;
}

void ref_set_light(void *h, const rtx_light *l) {
    Builder *b = (Builder *)h;
    b->light.position = V(l->position);
    b->light.color = V(l->color);
}

/* scene.h:49-54: a top-level box that create_boxs leaves alone */
int ref_add_leaf_box(void *h, const float lb[3], const float rt[3], const rtx_prim *prims,
                     uint32_t n) {
    Builder *b = (Builder *)h;
    Box bx;
    bx.lb = V(lb);
    bx.rt = V(rt);
    bx.depth = DEPTH_BOX;
    for (uint32_t k = 0; k < n; k++) {
        Object ob;
        memset(&ob, 0, sizeof(ob));
        ob.type = (Type)prims[k].type;
        memcpy(&ob.triangle, prims[k].g, 36);
        ob.material = M(&b->mats.at(prims[k].material));
        bx.objects.push_back(ob);
    }
    b->boxs.push_back(bx);
    return 0;
}

/* meshs.h:1-56 through the reference's create_mesh */
int ref_add_mesh(void *h, const float *verts, uint32_t nv, const uint32_t *faces, uint32_t nf,
                 const float position[3], float taille, uint32_t material) {
    Builder *b = (Builder *)h;
    Mesh mesh;
    mesh.position = V(position);
    mesh.taille = taille;
    mesh.material = M(&b->mats.at(material));
    for (uint32_t k = 0; k < nv; k++) mesh.vertices.push_back(V(verts + 3 * k));
    for (uint32_t k = 0; k < nf; k++)
        mesh.facades.push_back(
            Vec3F{(float)faces[3 * k], (float)faces[3 * k + 1], (float)faces[3 * k + 2]});
    create_mesh(mesh, b->boxs);
    return 0;
}

/* meshs.h:60-125 through the reference's parse */
int ref_parse_off(const char *path, float **verts, uint32_t *nv, uint32_t **faces, uint32_t *nf) {
    Mesh mesh;
    if (!parse(path, mesh)) return 1;
    *nv = (uint32_t)mesh.vertices.size();
    *nf = (uint32_t)mesh.facades.size();
    *verts = (float *)malloc((*nv ? *nv : 1) * 12);
    *faces = (uint32_t *)malloc((*nf ? *nf : 1) * 12);
    for (uint32_t k = 0; k < *nv; k++) S(*verts + 3 * k, mesh.vertices[k]);
    for (uint32_t k = 0; k < *nf; k++) {
        (*faces)[3 * k + 0] = (uint32_t)(int)mesh.facades[k].x;
        (*faces)[3 * k + 1] = (uint32_t)(int)mesh.facades[k].y;
        (*faces)[3 * k + 2] = (uint32_t)(int)mesh.facades[k].z;
    }
    return 0;
}
void ref_free(void *p) { free(p); }

/* The scene exactly as scene.h:4-108 assembles it; `dir` must hold meshs/. */
int ref_init_scene(void *h, const char *dir) {
    Builder *b = (Builder *)h;
    char old[4096];
    if (!getcwd(old, sizeof(old))) return 2;
    if (chdir(dir)) return 2;
    b->boxs.clear();
    bool ok = init_scene(b->light, b->boxs);
    if (chdir(old)) return 2;
    b->built = true; /* init_scene already ran create_boxs */
    return ok ? 0 : 1;
}

/* create_boxs (bvh.h:5-61) + flattening of the reference's tree */
int ref_build(void *h, rtx_scene_desc *out) {
    Builder *b = (Builder *)h;
    /* ids from the top-level object lists (create_boxs keeps them intact) */
    b->ids.clear();
    uint32_t next = 0;
    for (size_t k = 0; k < b->boxs.size(); k++)
        for (size_t o = 0; o < b->boxs[k].objects.size(); o++)
            b->ids[geom_key(b->boxs[k].objects[o])].push_back(next++);
    if (!b->built) {
        create_boxs(b->boxs);
        b->built = true;
    }
    b->nodes.clear();
    b->leaves.clear();
    b->prims.clear();
    b->leaf_index.clear();
    flatten(b, b->boxs);
    memset(out, 0, sizeof(*out));
    out->n_nodes = (uint32_t)b->nodes.size();
    out->n_leaves = (uint32_t)b->leaves.size();
    out->n_prims = (uint32_t)b->prims.size();
    out->n_materials = (uint32_t)b->mats.size();
    out->nodes = b->nodes.data();
    out->leaves = b->leaves.data();
    out->prims = b->prims.data();
    out->materials = b->mats.data();
    S(out->light.position, b->light.position);
    S(out->light.color, b->light.color);
    return 0;
}

/* ---- frame -------------------------------------------------------------------
 * raytracing.h:112-130 around the reference's functions.  Single-threaded.
 * counts: slab tests, leaf visits, hits, shadow rays, shaded samples. */
int ref_render(void *h, const rtx_params *p, uint8_t *rgb, float *fcolor, rtx_hit *hits,
               uint64_t hit_cap, uint64_t *n_hits, uint64_t counts[5]) {
    Builder *b = (Builder *)h;
    if (!b->built || b->nodes.empty()) return 1;
    std::vector<rtx_hit> hv;
    Ctx c;
    memset(&c, 0, sizeof(c));
    c.b = b;
    c.p = p;
    c.hits = (hits || n_hits) ? &hv : NULL;
    Light light = b->light;
    Vec3F camera = V(p->camera);
    float s = 1000.f / (float)p->height;
    size_t row = 0;
    for (uint32_t j = 0; j < p->height; j++) {
        if (p->n_parts > 1 && p->stripe_rows && (j / p->stripe_rows) % p->n_parts != p->part)
            continue;
        for (uint32_t i = 0; i < p->width; i++) {
            Ray ray;
            ray.origin = Vec3F{500.f + ((float)i - (float)p->width * 0.5f) * s, (float)j * s, 1000.f};
            ray.direction = normalize(ray.origin - camera);
            ray.depth = 0;
            c.pixel = j * p->width + i;
            c.ordinal = 0;
            Vec3F color = walk(c, light, ray, b->boxs);
            size_t px = row * p->width + i;
            if (fcolor) S(fcolor + 3 * px, color);
            if (rgb) {
                rgb[3 * px + 0] = color.x != color.x ? 0 : clamp_color(color.x);
                rgb[3 * px + 1] = color.y != color.y ? 0 : clamp_color(color.y);
                rgb[3 * px + 2] = color.z != color.z ? 0 : clamp_color(color.z);
            }
        }
        row++;
    }
    if (hits)
        for (size_t k = 0; k < hv.size() && k < hit_cap; k++) hits[k] = hv[k];
    if (n_hits) *n_hits = hv.size();
    if (counts) {
        counts[0] = c.slab;
        counts[1] = c.leaf_visits;
        counts[2] = c.nhits;
        counts[3] = c.shadow;
        counts[4] = c.shaded;
    }
    return 0;
}

} /* extern "C" */
