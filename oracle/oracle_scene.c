/*
 * oracle_scene.c - test-only CPU restatement of the reference's scene
 * assembly: OFF reader (meshs.h:60-125), create_mesh (meshs.h:1-56) and
 * create_boxs (bvh.h:5-61, utils.h:19-45), flattened into rtx_scene_desc.
 * All arithmetic is binary32 in the reference's operation order.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtx_oracle.h"

typedef struct {
    float x, y, z;
} v3;

/* vec3.h:10-31.  dot() is evaluated left to right in float. */
static float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static float v3_norm(v3 a) { return sqrtf(v3_dot(a, a)); }
/* std::min(a, b) = (b < a) ? b : a ; std::max(a, b) = (a < b) ? b : a */
static float std_min(float a, float b) { return (b < a) ? b : a; }
static float std_max(float a, float b) { return (a < b) ? b : a; }

typedef struct {
    float lb[3], rt[3];
    uint32_t depth;
    uint32_t first, count; /* range in builder prims */
} top_box;

struct orc_builder {
    uint32_t box_depth;
    rtx_light light;
    rtx_material *mats;
    uint32_t n_mats;
    rtx_prim *prims; /* primitives of all top-level boxes, push order */
    uint32_t n_prims, cap_prims;
    top_box *tops;
    uint32_t n_tops;
    /* flattened output */
    rtx_node *nodes;
    uint32_t n_nodes, cap_nodes;
    rtx_leaf *leaves;
    uint32_t n_leaves, cap_leaves;
    rtx_prim *refs;
    uint32_t n_refs, cap_refs;
};

orc_builder *orc_builder_new(uint32_t box_depth) {
    orc_builder *b = (orc_builder *)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->box_depth = box_depth;
    /* scene.h:80-81 */
    b->light.position[0] = 250.f;
    b->light.position[1] = 250.f;
    b->light.position[2] = 800.f;
    b->light.color[0] = b->light.color[1] = b->light.color[2] = 255.f;
    return b;
}

void orc_builder_free(orc_builder *b) {
    if (!b) return;
    free(b->mats);
    free(b->prims);
    free(b->tops);
    free(b->nodes);
    free(b->leaves);
    free(b->refs);
    free(b);
}

void orc_free(void *p) { free(p); }

void orc_set_light(orc_builder *b, const rtx_light *l) { b->light = *l; }

uint32_t orc_add_material(orc_builder *b, const rtx_material *m) {
    b->mats = (rtx_material *)realloc(b->mats, (b->n_mats + 1) * sizeof(*b->mats));
    b->mats[b->n_mats] = *m;
    return b->n_mats++;
}

static rtx_prim *push_prim(orc_builder *b) {
    if (b->n_prims == b->cap_prims) {
        b->cap_prims = b->cap_prims ? b->cap_prims * 2 : 1024;
        b->prims = (rtx_prim *)realloc(b->prims, b->cap_prims * sizeof(*b->prims));
    }
    rtx_prim *p = &b->prims[b->n_prims];
    memset(p, 0, sizeof(*p));
    p->id = b->n_prims++;
    return p;
}

static top_box *push_top(orc_builder *b) {
    b->tops = (top_box *)realloc(b->tops, (b->n_tops + 1) * sizeof(*b->tops));
    return &b->tops[b->n_tops++];
}

int orc_add_leaf_box(orc_builder *b, const float lb[3], const float rt[3], const rtx_prim *prims,
                     uint32_t n) {
    top_box t;
    memcpy(t.lb, lb, sizeof(t.lb));
    memcpy(t.rt, rt, sizeof(t.rt));
    t.depth = b->box_depth;
    t.first = b->n_prims;
    t.count = n;
    for (uint32_t k = 0; k < n; k++) {
        rtx_prim *p = push_prim(b);
        uint32_t id = p->id;
        *p = prims[k];
        p->id = id;
    }
    *push_top(b) = t;
    return 0;
}

/* meshs.h:1-56 */
int orc_add_mesh(orc_builder *b, const float *verts, uint32_t nv, const uint32_t *faces,
                 uint32_t nf, const float position[3], float taille, uint32_t material) {
    if (nv == 0) return 1;
    v3 *v = (v3 *)malloc(nv * sizeof(v3));
    memcpy(v, verts, nv * sizeof(v3));
    v3 mn = v[0], mx = v[0], sum = v[0];
    float norm_max = v3_norm(v[0]);
    for (uint32_t i = 1; i < nv; i++) {
        sum.x = sum.x + v[i].x;
        sum.y = sum.y + v[i].y;
        sum.z = sum.z + v[i].z;
        if (v3_norm(v[i]) > norm_max) norm_max = v3_norm(v[i]);
        mn.x = std_min(mn.x, v[i].x);
        mn.y = std_min(mn.y, v[i].y);
        mn.z = std_min(mn.z, v[i].z);
        mx.x = std_max(mx.x, v[i].x);
        mx.y = std_max(mx.y, v[i].y);
        mx.z = std_max(mx.z, v[i].z);
    }
    norm_max /= taille;
    float cnt = (float)nv;
    v3 cg = {sum.x / cnt, sum.y / cnt, sum.z / cnt};
    v3 off = {position[0] - cg.x / norm_max, position[1] - cg.y / norm_max,
              position[2] - cg.z / norm_max};
    for (uint32_t i = 0; i < nv; i++) {
        v[i].x = v[i].x / norm_max;
        v[i].y = v[i].y / norm_max;
        v[i].z = v[i].z / norm_max;
    }
    top_box t;
    t.lb[0] = mn.x / norm_max + off.x;
    t.lb[1] = mn.y / norm_max + off.y;
    t.lb[2] = mn.z / norm_max + off.z;
    t.rt[0] = mx.x / norm_max + off.x;
    t.rt[1] = mx.y / norm_max + off.y;
    t.rt[2] = mx.z / norm_max + off.z;
    t.depth = 0;
    t.first = b->n_prims;
    t.count = nf;
    for (uint32_t f = 0; f < nf; f++) {
        rtx_prim *p = push_prim(b);
        p->type = RTX_PRIM_TRIANGLE;
        p->material = material;
        for (int c = 0; c < 3; c++) {
            uint32_t idx = faces[3 * f + c];
            if (idx >= nv) {
                free(v);
                return 2;
            }
            p->g[3 * c + 0] = v[idx].x + off.x;
            p->g[3 * c + 1] = v[idx].y + off.y;
            p->g[3 * c + 2] = v[idx].z + off.z;
        }
    }
    *push_top(b) = t;
    free(v);
    return 0;
}

static uint32_t emit_node(orc_builder *b, const float lb[3], const float rt[3]) {
    if (b->n_nodes == b->cap_nodes) {
        b->cap_nodes = b->cap_nodes ? b->cap_nodes * 2 : 4096;
        b->nodes = (rtx_node *)realloc(b->nodes, b->cap_nodes * sizeof(*b->nodes));
    }
    rtx_node *n = &b->nodes[b->n_nodes];
    memcpy(n->lb, lb, sizeof(n->lb));
    memcpy(n->rt, rt, sizeof(n->rt));
    n->skip = 0;
    n->leaf = RTX_NONE;
    return b->n_nodes++;
}

static void emit_leaf(orc_builder *b, uint32_t node, const uint32_t *idx, uint32_t n) {
    if (b->n_leaves == b->cap_leaves) {
        b->cap_leaves = b->cap_leaves ? b->cap_leaves * 2 : 1024;
        b->leaves = (rtx_leaf *)realloc(b->leaves, b->cap_leaves * sizeof(*b->leaves));
    }
    while (b->n_refs + n > b->cap_refs) {
        b->cap_refs = b->cap_refs ? b->cap_refs * 2 : 4096;
        b->refs = (rtx_prim *)realloc(b->refs, b->cap_refs * sizeof(*b->refs));
    }
    b->leaves[b->n_leaves].first = b->n_refs;
    b->leaves[b->n_leaves].count = n;
    for (uint32_t k = 0; k < n; k++) b->refs[b->n_refs++] = b->prims[idx[k]];
    b->nodes[node].leaf = b->n_leaves++;
}

/* bvh.h:5-61 for one box: split the longer of the x / y extents at its
 * midpoint, distribute the triangles (utils.h:19-45), recurse low then high. */
static int split_box(orc_builder *b, const float lb[3], const float rt[3], uint32_t depth,
                     const uint32_t *idx, uint32_t n) {
    uint32_t node = emit_node(b, lb, rt);
    if (depth >= b->box_depth) {
        emit_leaf(b, node, idx, n);
        b->nodes[node].skip = b->n_nodes;
        return 0;
    }
    float longueur = rt[0] - lb[0];
    float hauteur = rt[1] - lb[1];
    int axis = (longueur > hauteur) ? 0 : 1;
    float cut = lb[axis] + ((axis == 0) ? longueur : hauteur) / 2;
    float lo_rt[3] = {rt[0], rt[1], rt[2]};
    float hi_lb[3] = {lb[0], lb[1], lb[2]};
    lo_rt[axis] = cut;
    hi_lb[axis] = cut;
    uint32_t *lo = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *hi = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t nlo = 0, nhi = 0;
    for (uint32_t k = 0; k < n; k++) {
        const rtx_prim *p = &b->prims[idx[k]];
        if (p->type != RTX_PRIM_TRIANGLE) {
            /* the reference would read the triangle arm of the union for a
             * sphere here; only boxes of triangles may be split */
            free(lo);
            free(hi);
            return 3;
        }
        float a0 = p->g[axis], a1 = p->g[3 + axis], a2 = p->g[6 + axis];
        if (a0 < cut && a1 < cut && a2 < cut)
            lo[nlo++] = idx[k];
        else if (a0 > cut && a1 > cut && a2 > cut)
            hi[nhi++] = idx[k];
        else {
            lo[nlo++] = idx[k];
            hi[nhi++] = idx[k];
        }
    }
    int rc = split_box(b, lb, lo_rt, depth + 1, lo, nlo);
    if (!rc) rc = split_box(b, hi_lb, rt, depth + 1, hi, nhi);
    free(lo);
    free(hi);
    b->nodes[node].skip = b->n_nodes;
    return rc;
}

int orc_build(orc_builder *b, rtx_scene_desc *out) {
    b->n_nodes = b->n_leaves = b->n_refs = 0;
    for (uint32_t t = 0; t < b->n_tops; t++) {
        top_box *tb = &b->tops[t];
        uint32_t *idx = (uint32_t *)malloc((tb->count ? tb->count : 1) * sizeof(uint32_t));
        for (uint32_t k = 0; k < tb->count; k++) idx[k] = tb->first + k;
        int rc = split_box(b, tb->lb, tb->rt, tb->depth, idx, tb->count);
        free(idx);
        if (rc) return rc;
    }
    memset(out, 0, sizeof(*out));
    out->n_nodes = b->n_nodes;
    out->n_leaves = b->n_leaves;
    out->n_prims = b->n_refs;
    out->n_materials = b->n_mats;
    out->nodes = b->nodes;
    out->leaves = b->leaves;
    out->prims = b->refs;
    out->materials = b->mats;
    out->light = b->light;
    return 0;
}

/* meshs.h:60-125.  The reference cuts each line at the blanks and converts the
 * pieces with atof / atoi (stopping at the first non-numeric character), and
 * keeps face indices as floats; (float)atof(s) is what strtod + cast gives. */
static char *read_line(FILE *f, char **buf, size_t *cap) {
    size_t len = 0;
    int c;
    while ((c = fgetc(f)) != EOF && c != '\n') {
        if (len + 2 > *cap) {
            *cap = *cap ? *cap * 2 : 256;
            *buf = (char *)realloc(*buf, *cap);
        }
        (*buf)[len++] = (char)c;
    }
    if (c == EOF && len == 0) return NULL;
    if (len + 1 > *cap) {
        *cap = *cap ? *cap * 2 : 256;
        *buf = (char *)realloc(*buf, *cap);
    }
    (*buf)[len] = 0;
    return *buf;
}

int orc_parse_off(const char *path, float **verts, uint32_t *nv, uint32_t **faces, uint32_t *nf) {
    FILE *f = fopen(path, "rb");
    if (!f) return 2;
    char *buf = NULL;
    size_t cap = 0;
    *verts = NULL;
    *faces = NULL;
    *nv = *nf = 0;
    if (!read_line(f, &buf, &cap) || strcmp(buf, "OFF") != 0) {
        fclose(f);
        free(buf);
        return 1;
    }
    char *s = read_line(f, &buf, &cap);
    long lnv = 0, lnf = 0;
    if (s) {
        /* meshs.h:88-91: nv = atoi(line[0 .. pos1]); the second find starts AT
         * pos1, so nf = atoi of the pos1 + 1 characters from pos1 on: a face
         * count with more digits than the vertex count is cut short. */
        char *sp = strchr(s, ' ');
        lnv = atoi(s);
        if (sp) {
            size_t pos1 = (size_t)(sp - s);
            char tmp[64];
            size_t n = pos1 + 1;
            if (n > sizeof(tmp) - 1) n = sizeof(tmp) - 1;
            strncpy(tmp, sp, n);
            tmp[n] = 0;
            lnf = atoi(tmp);
        }
    }
    if (lnv < 0) lnv = 0;
    if (lnf < 0) lnf = 0;
    float *v = (float *)calloc((size_t)(lnv ? lnv : 1) * 3, sizeof(float));
    uint32_t *fc = (uint32_t *)calloc((size_t)(lnf ? lnf : 1) * 3, sizeof(uint32_t));
    for (long n = 0; n < lnv; n++) {
        s = read_line(f, &buf, &cap);
        if (!s) break;
        /* three blank-separated fields, the first starting at column 0 */
        char *p = s;
        for (int c = 0; c < 3; c++) {
            v[3 * n + c] = (float)strtod(p, NULL);
            p = strchr(c == 0 ? p : p + 1, ' ');
            if (!p) break;
        }
    }
    for (long n = 0; n < lnf; n++) {
        s = read_line(f, &buf, &cap);
        if (!s) break;
        /* the leading vertex count is skipped, then three indices */
        char *p = strchr(s, ' ');
        for (int c = 0; c < 3 && p; c++) {
            float fi = (float)strtod(p, NULL);
            fc[3 * n + c] = (uint32_t)(int)fi;
            p = strchr(p + 1, ' ');
        }
    }
    fclose(f);
    free(buf);
    *verts = v;
    *nv = (uint32_t)lnv;
    *faces = fc;
    *nf = (uint32_t)lnf;
    return 0;
}
