/*
 * pt_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; pinned against the reference's compiled shaders, see pt_oracle.h).
 *
 * Restates, in plain C with fully specified float32 arithmetic:
 *   shaders/common.glsl:13-37        pcg, pcg2d, rand
 *   shaders/raygen.rgen:14-39        createCoordinateSystem, sampleHemisphere, sampleDirection
 *   shaders/raygen.rgen:41-91        sample loop, primary ray, bounce loop, running mean
 *   shaders/closesthit.rchit:24-65   vertex/face fetch, barycentric position, geometric normal
 *   shaders/miss.rmiss:8-12          environment term
 *   main.cpp:497-538 + raygen.rgen:63-75   closest-hit query semantics (opaque, no culling,
 *                                          tMin < t < tMax), here Woop/Benthin/Wald 2013.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -shared -fPIC (see Makefile).
 * -ffp-contract=off is REQUIRED: the canonical arithmetic forbids FMA contraction.
 */
#define _GNU_SOURCE
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MISS 0xFFFFFFFFu
#define ORC_LEAF 0x80000000u

/* ===== RNG: shaders/common.glsl ===================================================== */

uint32_t orc_pcg(uint32_t *state) /* common.glsl:13-19 */
{
    uint32_t prev = *state * 747796405u + 2891336453u;
    uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
    *state = prev;
    return (word >> 22u) ^ word;
}

void orc_pcg2d(uint32_t vx, uint32_t vy, uint32_t out[2]) /* common.glsl:21-31 */
{
    vx = vx * 1664525u + 1013904223u;
    vy = vy * 1664525u + 1013904223u;
    vx += vy * 1664525u;
    vy += vx * 1664525u;
    vx ^= vx >> 16u;
    vy ^= vy >> 16u;
    vx += vy * 1664525u;
    vy += vx * 1664525u;
    vx ^= vx >> 16u;
    vy ^= vy >> 16u;
    out[0] = vx;
    out[1] = vy;
}

float orc_rand(uint32_t *seed) /* common.glsl:33-37; 1.0/float(0xffffffffu) folds to 2^-32 */
{
    uint32_t val = orc_pcg(seed);
    return (float)val * 2.3283064365386963e-10f; /* u32->f32 is RNE; result in [0,1] incl. */
}

uint32_t orc_seed(uint32_t px, uint32_t py, uint32_t sample, int32_t frame, uint32_t spp)
{
    /* raygen.rgen:47-48: m = sampleNum + uint(maxSamples*frame) + 1 (i32 multiply, wrap) */
    uint32_t m = sample + (uint32_t)((int32_t)spp * frame) + 1u;
    uint32_t s[2];
    orc_pcg2d(px * m, py * m, s);
    return s[0] + s[1];
}

/* ===== canonical sin/cos ============================================================= */

void orc_sincos(float a, float *s, float *c)
{
    /* quadrant index j = nearest multiple of pi/2, a in [0, 2*pi] -> j in 0..4 */
    int j = (int)(a * 0.636619772f + 0.5f);
    float fj = (float)j;
    /* Cody-Waite: pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8 */
    float r = a - fj * 1.5703125f;
    r = r - fj * 4.837512969970703125e-4f;
    r = r - fj * 7.54978995489188e-8f;
    float z = r * r;
    /* cephes sinf / cosf minimax polynomials on |r| <= pi/4, Horner, no FMA */
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    ps = ps * z;
    ps = ps * r + r;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    switch (j & 3) {
    case 0: *s = ps;  *c = pc;  break;
    case 1: *s = pc;  *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
    }
}

/* ===== params ======================================================================== */

void orc_params_default(orc_params *p)
{
    memset(p, 0, sizeof(*p));
    p->frame = 0;
    p->width = 1024;  /* main.cpp:16 */
    p->height = 1024; /* main.cpp:17 */
    p->spp_per_frame = 32;
    p->max_depth = 8;
    p->tmin = 0.001f;
    p->tmax = 10000.0f;
    p->cam_origin[0] = 0.0f; p->cam_origin[1] = -1.0f; p->cam_origin[2] = 5.0f;
    p->cam_target[0] = 0.0f; p->cam_target[1] = -1.0f; p->cam_target[2] = 2.0f;
    p->env[0] = 0.7f; p->env[1] = 0.6f; p->env[2] = 0.5f;
    p->libm_sincos = 0;
}

/* ===== scene + LBVH ================================================================== */

typedef struct orc_node {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    uint32_t left, right, pad0, pad1;
} orc_node;

/* one LBVH over n boxes (triangles of the BLAS, or instances of the TLAS) */
typedef struct orc_bvh {
    uint32_t n;
    uint64_t *keys;      /* sorted */
    uint32_t *prim_of;   /* sorted position -> box id */
    orc_node *nodes;     /* n-1 internal nodes (>=1) */
    uint32_t n_nodes, height;
    float bmin[3], bmax[3];
} orc_bvh;

typedef struct orc_instance {
    float m[12];     /* object -> world, 3x4 row major (VkTransformMatrixKHR layout)       */
    float inv[12];   /* world -> object, 3x4 row major: rows of A^-1 with -A^-1 t in column 3 */
} orc_instance;

struct orc_scene {
    uint32_t n_tris;
    float *tri;    /* 9 floats per triangle, original (prim id) order: v0 v1 v2          */
    float *face;   /* 6 floats per triangle: Kd, Ke                                      */
    orc_bvh blas;  /* LBVH over the triangles */
    /* two-level scenes (config C4; the reference has one identity instance, main.cpp:515-538) */
    uint32_t n_inst;
    orc_instance *inst;
    orc_bvh tlas;  /* LBVH over the instances' world boxes */
    /* emitters (Ke != 0) in primitive order, for the NEE estimator: 16 floats each
     * {A.xyz, B.xyz, C.xyz, N.xyz, Ke.rgb, cdf} and the sum of their areas */
    uint32_t n_lights;
    float *lights;
    float light_area;
};

static inline uint64_t expand21(uint32_t v)
{
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

static inline uint32_t quant21(float c, float lo, float ext)
{
    float n = ext > 0.0f ? (c - lo) / ext : 0.0f;
    float q = n * 2097152.0f;
    if (!(q >= 0.0f)) q = 0.0f;
    if (q > 2097151.0f) q = 2097151.0f;
    return (uint32_t)q;
}

static inline int delta(const uint64_t *keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
    return __builtin_clzll(a ^ b);
}

typedef struct { uint64_t key; uint32_t prim; } keyprim;
static int cmp_keyprim(const void *a, const void *b)
{
    const keyprim *x = a, *y = b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->prim < y->prim ? -1 : (x->prim > y->prim);
}

static void tri_bounds(const float *t, float mn[3], float mx[3])
{
    for (int k = 0; k < 3; k++) {
        mn[k] = fminf(fminf(t[k], t[3 + k]), t[6 + k]);
        mx[k] = fmaxf(fmaxf(t[k], t[3 + k]), t[6 + k]);
    }
}

/* box of child reference (leaf or internal) written into mn/mx; returns subtree height */
static uint32_t build_boxes(orc_bvh *b, const float *blo, const float *bhi, uint32_t ref, float pad,
                            float mn[3], float mx[3])
{
    if (ref & ORC_LEAF) {
        uint32_t id = b->prim_of[ref & ~ORC_LEAF];
        for (int k = 0; k < 3; k++) { mn[k] = blo[3 * (size_t)id + k] - pad; mx[k] = bhi[3 * (size_t)id + k] + pad; }
        return 0;
    }
    orc_node *nd = &b->nodes[ref];
    uint32_t hl = build_boxes(b, blo, bhi, nd->left, pad, nd->lmin, nd->lmax);
    uint32_t hr = build_boxes(b, blo, bhi, nd->right, pad, nd->rmin, nd->rmax);
    for (int k = 0; k < 3; k++) {
        mn[k] = fminf(nd->lmin[k], nd->rmin[k]);
        mx[k] = fmaxf(nd->lmax[k], nd->rmax[k]);
    }
    return 1 + (hl > hr ? hl : hr);
}

static void bvh_free(orc_bvh *b)
{
    free(b->keys); free(b->prim_of); free(b->nodes);
    memset(b, 0, sizeof(*b));
}

/* LBVH over n boxes blo/bhi (3 floats each): Morton keys of the box centres, stable sort,
 * Karras 2012 hierarchy, leaf boxes padded by 2^-18 of the scene scale. */
static void build_lbvh(orc_bvh *b, const float *blo, const float *bhi, uint32_t n_boxes)
{
    const int n = (int)n_boxes;
    b->n = n_boxes;
    for (int k = 0; k < 3; k++) { b->bmin[k] = INFINITY; b->bmax[k] = -INFINITY; }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            b->bmin[k] = fminf(b->bmin[k], blo[3 * (size_t)i + k]);
            b->bmax[k] = fmaxf(b->bmax[k], bhi[3 * (size_t)i + k]);
        }
    float ext[3];
    for (int k = 0; k < 3; k++) ext[k] = b->bmax[k] - b->bmin[k];
    /* Morton keys of box centres, 21 bits per axis, x most significant */
    keyprim *kp = malloc(sizeof(keyprim) * (size_t)n);
    for (int i = 0; i < n; i++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++)
            q[k] = quant21((blo[3 * (size_t)i + k] + bhi[3 * (size_t)i + k]) * 0.5f, b->bmin[k], ext[k]);
        kp[i].key = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
        kp[i].prim = (uint32_t)i;
    }
    qsort(kp, (size_t)n, sizeof(keyprim), cmp_keyprim); /* (key, id) order == stable sort */
    b->keys = malloc(sizeof(uint64_t) * (size_t)n);
    b->prim_of = malloc(sizeof(uint32_t) * (size_t)n);
    for (int i = 0; i < n; i++) { b->keys[i] = kp[i].key; b->prim_of[i] = kp[i].prim; }
    free(kp);

    b->n_nodes = n > 1 ? (uint32_t)(n - 1) : 1u;
    b->nodes = calloc(b->n_nodes, sizeof(orc_node));
    if (n == 1) {
        b->nodes[0].left = ORC_LEAF | 0u;
        b->nodes[0].right = ORC_LEAF | 0u;
    }
    /* Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees" */
    for (int i = 0; i < n - 1; i++) {
        const uint64_t *K = b->keys;
        int d = (delta(K, n, i, i + 1) - delta(K, n, i, i - 1)) < 0 ? -1 : 1;
        int dmin = delta(K, n, i, i - d);
        int lmax = 2;
        while (delta(K, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(K, n, i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d;
        int dnode = delta(K, n, i, j);
        int sp = 0;
        int t = l;
        do {
            t = (t + 1) >> 1;
            if (delta(K, n, i, i + (sp + t) * d) > dnode) sp += t;
        } while (t > 1);
        int gamma = i + sp * d + (d < 0 ? -1 : 0);
        int lo = i < j ? i : j, hi = i < j ? j : i;
        b->nodes[i].left = (lo == gamma) ? (ORC_LEAF | (uint32_t)gamma) : (uint32_t)gamma;
        b->nodes[i].right = (hi == gamma + 1) ? (ORC_LEAF | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    }
    /* leaf boxes are padded by 2^-18 * scene scale so the slab test is conservative w.r.t.
     * the (rounded) watertight triangle test */
    float scale = 0.0f;
    for (int k = 0; k < 3; k++) scale = fmaxf(scale, fmaxf(fabsf(b->bmin[k]), fabsf(b->bmax[k])));
    float pad = scale * 3.814697265625e-06f;
    float mn[3], mx[3];
    b->height = build_boxes(b, blo, bhi, 0u, pad, mn, mx);
}

static void build_blas(orc_scene *s)
{
    const size_t n = s->n_tris;
    float *blo = malloc(sizeof(float) * 3 * n), *bhi = malloc(sizeof(float) * 3 * n);
    for (size_t i = 0; i < n; i++) tri_bounds(s->tri + 9 * i, blo + 3 * i, bhi + 3 * i);
    build_lbvh(&s->blas, blo, bhi, s->n_tris);
    free(blo); free(bhi);
}

/* ---- instances (beyond the reference: it has exactly one identity instance) --------------- */
/* world -> object matrix: adjugate / determinant in binary64, rounded once to float */
static void invert_3x4(const float m[12], float inv[12])
{
    const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9],
                 a22 = m[10];
    const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const double det = (a00 * c00 + a01 * c01) + a02 * c02;
    const double i00 = c00 / det, i01 = (a02 * a21 - a01 * a22) / det, i02 = (a01 * a12 - a02 * a11) / det;
    const double i10 = c01 / det, i11 = (a00 * a22 - a02 * a20) / det, i12 = (a02 * a10 - a00 * a12) / det;
    const double i20 = c02 / det, i21 = (a01 * a20 - a00 * a21) / det, i22 = (a00 * a11 - a01 * a10) / det;
    const double tx = m[3], ty = m[7], tz = m[11];
    inv[0] = (float)i00; inv[1] = (float)i01; inv[2] = (float)i02;  inv[3] = (float)(-((i00 * tx + i01 * ty) + i02 * tz));
    inv[4] = (float)i10; inv[5] = (float)i11; inv[6] = (float)i12;  inv[7] = (float)(-((i10 * tx + i11 * ty) + i12 * tz));
    inv[8] = (float)i20; inv[9] = (float)i21; inv[10] = (float)i22; inv[11] = (float)(-((i20 * tx + i21 * ty) + i22 * tz));
}

static inline void xform_point(const float m[12], const float p[3], float out[3])
{
    for (int k = 0; k < 3; k++) out[k] = ((m[4 * k] * p[0] + m[4 * k + 1] * p[1]) + m[4 * k + 2] * p[2]) + m[4 * k + 3];
}
static inline void xform_vector(const float m[12], const float v[3], float out[3])
{
    for (int k = 0; k < 3; k++) out[k] = (m[4 * k] * v[0] + m[4 * k + 1] * v[1]) + m[4 * k + 2] * v[2];
}

/* Emitters for the NEE estimator (pt_api: PT_PIPELINE_WAVEFRONT_NEE): every triangle with Ke != 0, in primitive order --
 * for an instanced scene once per instance, in gl_InstanceID order, with the vertices taken to world space by the
 * instance's matrix (xform_point); normal = -cross / |cross| as closesthit.rchit:43-48 (of the world-space triangle: only
 * |cos| of it is used); area = |cross| / 2; cdf = running float sum of the areas. */
static void build_lights(orc_scene *s)
{
    free(s->lights); s->lights = NULL; s->n_lights = 0; s->light_area = 0.0f;
    uint32_t per = 0;
    for (uint32_t t = 0; t < s->n_tris; t++) {
        const float *f = s->face + 6 * (size_t)t;
        if (f[3] != 0.0f || f[4] != 0.0f || f[5] != 0.0f) per++;
    }
    const uint32_t copies = s->n_inst ? s->n_inst : 1u;
    if (!per) return;
    s->n_lights = per * copies;
    s->lights = malloc(sizeof(float) * 16 * (size_t)s->n_lights);
    uint32_t k = 0;
    float run = 0.0f;
    for (uint32_t i = 0; i < copies; i++)
        for (uint32_t t = 0; t < s->n_tris; t++) {
            const float *f = s->face + 6 * (size_t)t;
            if (!(f[3] != 0.0f || f[4] != 0.0f || f[5] != 0.0f)) continue;
            const float *tv = s->tri + 9 * (size_t)t;
            float *L = s->lights + 16 * (size_t)k++;
            float e1[3], e2[3];
            if (s->n_inst) for (int c = 0; c < 3; c++) xform_point(s->inst[i].m, tv + 3 * c, L + 3 * c);
            else for (int c = 0; c < 9; c++) L[c] = tv[c];
            for (int c = 0; c < 3; c++) { e1[c] = L[3 + c] - L[c]; e2[c] = L[6 + c] - L[c]; }
            float cx = e1[1] * e2[2] - e1[2] * e2[1];
            float cy = e1[2] * e2[0] - e1[0] * e2[2];
            float cz = e1[0] * e2[1] - e1[1] * e2[0];
            float len = sqrtf((cx * cx + cy * cy) + cz * cz);
            L[9] = -(cx / len); L[10] = -(cy / len); L[11] = -(cz / len);
            L[12] = f[3]; L[13] = f[4]; L[14] = f[5];
            run = run + 0.5f * len;
            L[15] = run;
        }
    s->light_area = run;
}

int orc_scene_set_instances(orc_scene *s, const float *xforms3x4, uint32_t n)
{
    if (!s || (n && !xforms3x4)) return 1;
    free(s->inst); s->inst = NULL; s->n_inst = 0;
    bvh_free(&s->tlas);
    if (n == 0) { build_lights(s); return 0; }
    s->inst = malloc(sizeof(orc_instance) * (size_t)n);
    s->n_inst = n;
    float *blo = malloc(sizeof(float) * 3 * (size_t)n), *bhi = malloc(sizeof(float) * 3 * (size_t)n);
    /* object box = the BLAS root box (children of node 0, already padded) */
    float omin[3], omax[3];
    for (int k = 0; k < 3; k++) {
        omin[k] = fminf(s->blas.nodes[0].lmin[k], s->blas.nodes[0].rmin[k]);
        omax[k] = fmaxf(s->blas.nodes[0].lmax[k], s->blas.nodes[0].rmax[k]);
    }
    for (uint32_t i = 0; i < n; i++) {
        memcpy(s->inst[i].m, xforms3x4 + 12 * (size_t)i, sizeof(float) * 12);
        invert_3x4(s->inst[i].m, s->inst[i].inv);
        for (int k = 0; k < 3; k++) { blo[3 * (size_t)i + k] = INFINITY; bhi[3 * (size_t)i + k] = -INFINITY; }
        for (int c = 0; c < 8; c++) { /* world box of the 8 transformed corners */
            float p[3] = { (c & 1) ? omax[0] : omin[0], (c & 2) ? omax[1] : omin[1], (c & 4) ? omax[2] : omin[2] }, w[3];
            xform_point(s->inst[i].m, p, w);
            for (int k = 0; k < 3; k++) {
                blo[3 * (size_t)i + k] = fminf(blo[3 * (size_t)i + k], w[k]);
                bhi[3 * (size_t)i + k] = fmaxf(bhi[3 * (size_t)i + k], w[k]);
            }
        }
    }
    build_lbvh(&s->tlas, blo, bhi, n);
    free(blo); free(bhi);
    build_lights(s);
    return 0;
}

orc_scene *orc_scene_create(const float *vertices, uint32_t n_verts, const uint32_t *indices,
                            uint32_t n_tris, const float *faces)
{
    if (!vertices || !indices || !faces || n_tris == 0) return NULL;
    for (uint32_t i = 0; i < 3 * n_tris; i++)
        if (indices[i] >= n_verts) return NULL;
    orc_scene *s = calloc(1, sizeof(*s));
    s->n_tris = n_tris;
    s->tri = malloc(sizeof(float) * 9 * (size_t)n_tris);
    s->face = malloc(sizeof(float) * 6 * (size_t)n_tris);
    for (uint32_t t = 0; t < n_tris; t++)
        for (int c = 0; c < 3; c++) /* closesthit.rchit:52-54: vertices[3*indices[3*prim+c] + k] */
            for (int k = 0; k < 3; k++)
                s->tri[9 * (size_t)t + 3 * c + k] = vertices[3 * (size_t)indices[3 * (size_t)t + c] + k];
    memcpy(s->face, faces, sizeof(float) * 6 * (size_t)n_tris);
    build_blas(s);
    build_lights(s);
    return s;
}

void orc_scene_destroy(orc_scene *s)
{
    if (!s) return;
    free(s->tri); free(s->face); free(s->inst); free(s->lights); bvh_free(&s->blas); bvh_free(&s->tlas); free(s);
}

void orc_scene_bvh_info(const orc_scene *s, orc_bvh_info *info)
{
    info->n_tris = s->n_tris; info->n_nodes = s->blas.n_nodes; info->height = s->blas.height;
    for (int k = 0; k < 3; k++) { info->bbox_min[k] = s->blas.bmin[k]; info->bbox_max[k] = s->blas.bmax[k]; }
}
void orc_scene_bvh_keys(const orc_scene *s, uint64_t *keys, uint32_t *prim_of_pos)
{
    memcpy(keys, s->blas.keys, sizeof(uint64_t) * s->n_tris);
    memcpy(prim_of_pos, s->blas.prim_of, sizeof(uint32_t) * s->n_tris);
}
void orc_scene_bvh_nodes(const orc_scene *s, uint32_t *nodes16)
{
    memcpy(nodes16, s->blas.nodes, sizeof(orc_node) * s->blas.n_nodes);
}

/* ===== closest hit =================================================================== */

typedef struct ray_pre {
    float org[3];
    int kx, ky, kz;
    float Sx, Sy, Sz;
    float tmin;
} ray_pre;

static void ray_setup(ray_pre *r, const float org[3], const float dir[3], float tmin)
{
    /* Woop et al. 2013, section 3: kz = dimension of largest |dir| (first wins on ties).
     * The kx/ky swap that preserves winding is omitted: without culling it only negates
     * U,V,W,det together, which changes no result bit. */
    int kz = 0;
    if (fabsf(dir[1]) > fabsf(dir[0])) kz = 1;
    if (fabsf(dir[2]) > fabsf(dir[kz])) kz = 2;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    r->kx = kx; r->ky = ky; r->kz = kz;
    r->Sx = dir[kx] / dir[kz];
    r->Sy = dir[ky] / dir[kz];
    r->Sz = 1.0f / dir[kz];
    r->org[0] = org[0]; r->org[1] = org[1]; r->org[2] = org[2];
    r->tmin = tmin;
}

/* returns 1 and fills t,u,v if tmin < t < tmax */
static inline int tri_test(const ray_pre *r, const float *tv, float tmax, float *t_out,
                           float *u_out, float *v_out)
{
    float A[3], B[3], C[3];
    for (int k = 0; k < 3; k++) {
        A[k] = tv[k] - r->org[k];
        B[k] = tv[3 + k] - r->org[k];
        C[k] = tv[6 + k] - r->org[k];
    }
    const float Ax = A[r->kx] - r->Sx * A[r->kz];
    const float Ay = A[r->ky] - r->Sy * A[r->kz];
    const float Bx = B[r->kx] - r->Sx * B[r->kz];
    const float By = B[r->ky] - r->Sy * B[r->kz];
    const float Cx = C[r->kx] - r->Sx * C[r->kz];
    const float Cy = C[r->ky] - r->Sy * C[r->kz];
    const float U = Cx * By - Cy * Bx;
    const float V = Ax * Cy - Ay * Cx;
    const float W = Bx * Ay - By * Ax;
    /* zero edge functions count as inside (conservative, keeps shared edges watertight) */
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    const float det = (U + V) + W;
    if (det == 0.0f) return 0;
    const float Az = r->Sz * A[r->kz];
    const float Bz = r->Sz * B[r->kz];
    const float Cz = r->Sz * C[r->kz];
    const float T = (U * Az + V * Bz) + W * Cz;
    const float t = T / det;
    if (!(t > r->tmin && t < tmax)) return 0; /* raygen.rgen:71,73: tMin < t < tMax; NaN -> miss */
    *t_out = t;
    *u_out = V / det; /* weight of v1 == attribs.x, closesthit.rchit:56 */
    *v_out = W / det; /* weight of v2 == attribs.y */
    return 1;
}

static inline void hit_consider(orc_hit *h, uint32_t inst, uint32_t prim, float t, float u, float v)
{
    /* closest t; equal t -> lowest (instance id, primitive id): deterministic for the OBJ's
     * duplicated quads and for coincident instances */
    if (t < h->t || (t == h->t && (inst < h->inst || (inst == h->inst && prim < h->prim)))) {
        h->t = t; h->u = u; h->v = v; h->prim = prim; h->inst = inst;
    }
}

static inline float safe_inv(float d)
{
    if (fabsf(d) < 1e-20f) d = copysignf(1e-20f, d);
    return 1.0f / d;
}

static inline int box_test(const float mn[3], const float mx[3], const float org[3],
                           const float inv[3], float tmin, float tbest, float *tnear)
{
    float tn = tmin, tf = tbest;
    for (int k = 0; k < 3; k++) {
        float t0 = (mn[k] - org[k]) * inv[k];
        float t1 = (mx[k] - org[k]) * inv[k];
        tn = fmaxf(tn, fminf(t0, t1));
        tf = fminf(tf, fmaxf(t0, t1));
    }
    *tnear = tn;
    return tn <= tf * 1.0000004f;
}

/* closest hit against the triangles of one BLAS copy; inst = id recorded with the hit */
static void trace_blas(const orc_scene *s, int mode, uint32_t inst, const float org[3], const float dir[3],
                       float tmin, float tmax, orc_hit *h, uint64_t *nodes, uint64_t *tris)
{
    ray_pre r;
    ray_setup(&r, org, dir, tmin);
    float t, u, v;
    if (mode == 0) {
        for (uint32_t p = 0; p < s->n_tris; p++) {
            (*tris)++;
            if (tri_test(&r, s->tri + 9 * (size_t)p, tmax, &t, &u, &v)) hit_consider(h, inst, p, t, u, v);
        }
        return;
    }
    const orc_bvh *b = &s->blas;
    float inv[3] = { safe_inv(dir[0]), safe_inv(dir[1]), safe_inv(dir[2]) };
    uint32_t stack[128];
    int sp = 0;
    uint32_t ref = 0; /* root internal node */
    for (;;) {
        if (ref & ORC_LEAF) {
            uint32_t prim = b->prim_of[ref & ~ORC_LEAF];
            (*tris)++;
            if (tri_test(&r, s->tri + 9 * (size_t)prim, tmax, &t, &u, &v)) hit_consider(h, inst, prim, t, u, v);
        } else {
            const orc_node *nd = &b->nodes[ref];
            float tl, tr;
            *nodes += 2;
            int hl = box_test(nd->lmin, nd->lmax, org, inv, tmin, h->t, &tl);
            int hr = box_test(nd->rmin, nd->rmax, org, inv, tmin, h->t, &tr);
            if (hl && hr) {
                uint32_t nearc = nd->left, farc = nd->right;
                if (tr < tl) { nearc = nd->right; farc = nd->left; }
                stack[sp++] = farc;
                ref = nearc;
                continue;
            } else if (hl) { ref = nd->left; continue; }
            else if (hr) { ref = nd->right; continue; }
        }
        if (sp == 0) break;
        ref = stack[--sp];
    }
}

/* one instance: ray into object space (direction NOT renormalised, so t is the same parameter
 * in both spaces, as in Vulkan), then the BLAS */
static void trace_instance(const orc_scene *s, int mode, uint32_t inst, const float org[3], const float dir[3],
                           float tmin, float tmax, orc_hit *h, uint64_t *nodes, uint64_t *tris)
{
    float oo[3], od[3];
    xform_point(s->inst[inst].inv, org, oo);
    xform_vector(s->inst[inst].inv, dir, od);
    trace_blas(s, mode, inst, oo, od, tmin, tmax, h, nodes, tris);
}

void orc_trace(const orc_scene *s, int mode, const float org[3], const float dir[3], float tmin,
               float tmax, orc_hit *hit, orc_counters *cnt)
{
    orc_hit h; h.prim = ORC_MISS; h.inst = ORC_MISS; h.t = tmax; h.u = 0.0f; h.v = 0.0f;
    /* h.t starts at tmax: candidates need t < tmax strictly; the tie-break can not fire for
     * t == tmax because tri_test already rejected t >= tmax. */
    uint64_t nodes = 0, tris = 0;
    if (s->n_inst == 0) {
        trace_blas(s, mode, 0u, org, dir, tmin, tmax, &h, &nodes, &tris);
    } else if (mode == 0) {
        for (uint32_t i = 0; i < s->n_inst; i++) trace_instance(s, 0, i, org, dir, tmin, tmax, &h, &nodes, &tris);
    } else {
        const orc_bvh *b = &s->tlas;
        float inv[3] = { safe_inv(dir[0]), safe_inv(dir[1]), safe_inv(dir[2]) };
        uint32_t stack[128];
        int sp = 0;
        uint32_t ref = 0;
        for (;;) {
            if (ref & ORC_LEAF) {
                trace_instance(s, 1, b->prim_of[ref & ~ORC_LEAF], org, dir, tmin, tmax, &h, &nodes, &tris);
            } else {
                const orc_node *nd = &b->nodes[ref];
                float tl, tr;
                nodes += 2;
                int hl = box_test(nd->lmin, nd->lmax, org, inv, tmin, h.t, &tl);
                int hr = box_test(nd->rmin, nd->rmax, org, inv, tmin, h.t, &tr);
                if (hl && hr) {
                    uint32_t nearc = nd->left, farc = nd->right;
                    if (tr < tl) { nearc = nd->right; farc = nd->left; }
                    stack[sp++] = farc;
                    ref = nearc;
                    continue;
                } else if (hl) { ref = nd->left; continue; }
                else if (hr) { ref = nd->right; continue; }
            }
            if (sp == 0) break;
            ref = stack[--sp];
        }
    }
    if (h.prim == ORC_MISS) { h.t = 0.0f; h.inst = ORC_MISS; }
    else if (s->n_inst == 0) h.inst = 0u;
    *hit = h;
    if (cnt) { cnt->rays += 1; cnt->nodes_visited += nodes; cnt->tris_tested += tris; }
}

void orc_trace_batch(const orc_scene *s, int mode, uint32_t n, const float *rays6, float tmin,
                     float tmax, orc_hit *hits, orc_counters *cnt)
{
    for (uint32_t i = 0; i < n; i++)
        orc_trace(s, mode, rays6 + 6 * (size_t)i, rays6 + 6 * (size_t)i + 3, tmin, tmax, &hits[i], cnt);
}

/* ===== shading pieces ================================================================ */

void orc_primary_ray(const orc_params *p, uint32_t px, uint32_t py, uint32_t *seed, float org[3],
                     float dir[3])
{
    /* raygen.rgen:51-57; jitter x first, then y (SPIR-V order) */
    float jx = orc_rand(seed);
    float jy = orc_rand(seed);
    float sx = (float)px + jx;
    float sy = (float)py + jy;
    float ux = sx / (float)p->width;  /* no aspect-ratio correction: kept */
    float uy = sy / (float)p->height;
    float dx = ux * 2.0f - 1.0f;
    float dy = uy * 2.0f - 1.0f;
    float tx = dx + p->cam_target[0];
    float ty = dy + p->cam_target[1];
    float tz = p->cam_target[2];
    float vx = tx - p->cam_origin[0];
    float vy = ty - p->cam_origin[1];
    float vz = tz - p->cam_origin[2];
    float len = sqrtf((vx * vx + vy * vy) + vz * vz);
    org[0] = p->cam_origin[0]; org[1] = p->cam_origin[1]; org[2] = p->cam_origin[2];
    dir[0] = vx / len; dir[1] = vy / len; dir[2] = vz / len;
}

void orc_shade_hit(const orc_scene *s, const orc_hit *h, float position[3], float normal[3],
                   float brdf[3], float emission[3])
{
    /* closesthit.rchit:52-62 */
    const float *tv = s->tri + 9 * (size_t)h->prim;
    const float b0 = (1.0f - h->u) - h->v;
    for (int k = 0; k < 3; k++)
        position[k] = (tv[k] * b0 + tv[3 + k] * h->u) + tv[6 + k] * h->v;
    float e1[3], e2[3];
    for (int k = 0; k < 3; k++) { e1[k] = tv[3 + k] - tv[k]; e2[k] = tv[6 + k] - tv[k]; }
    float cx = e1[1] * e2[2] - e1[2] * e2[1];
    float cy = e1[2] * e2[0] - e1[0] * e2[2];
    float cz = e1[0] * e2[1] - e1[1] * e2[0];
    float len = sqrtf((cx * cx + cy * cy) + cz * cz);
    normal[0] = -(cx / len); normal[1] = -(cy / len); normal[2] = -(cz / len); /* never flipped */
    const float *f = s->face + 6 * (size_t)h->prim;
    for (int k = 0; k < 3; k++) {
        brdf[k] = f[k] / 3.1415927410125732f; /* true divide by float(pi), closesthit.rchit:60 */
        emission[k] = f[3 + k];
    }
    if (s->n_inst) {
        /* instanced scenes (not in the reference, whose closesthit uses object-space vertices
         * with one identity instance): position by the object->world matrix, normal by the
         * inverse transpose, renormalised */
        const orc_instance *in = &s->inst[h->inst];
        float pw[3], nw[3];
        xform_point(in->m, position, pw);
        for (int k = 0; k < 3; k++)
            nw[k] = (in->inv[k] * normal[0] + in->inv[4 + k] * normal[1]) + in->inv[8 + k] * normal[2];
        float l = sqrtf((nw[0] * nw[0] + nw[1] * nw[1]) + nw[2] * nw[2]);
        for (int k = 0; k < 3; k++) { position[k] = pw[k]; normal[k] = nw[k] / l; }
    }
}

void orc_sample_direction(float r1, float r2, const float n[3], int libm, float out[3])
{
    /* raygen.rgen:14-21 createCoordinateSystem */
    float T[3], B[3];
    if (fabsf(n[0]) > fabsf(n[1])) {
        float l = sqrtf(n[0] * n[0] + n[2] * n[2]);
        T[0] = n[2] / l; T[1] = 0.0f / l; T[2] = -n[0] / l;
    } else {
        float l = sqrtf(n[1] * n[1] + n[2] * n[2]);
        T[0] = 0.0f / l; T[1] = -n[2] / l; T[2] = n[1] / l;
    }
    B[0] = n[1] * T[2] - n[2] * T[1];
    B[1] = n[2] * T[0] - n[0] * T[2];
    B[2] = n[0] * T[1] - n[1] * T[0];
    /* raygen.rgen:23-30 sampleHemisphere (uniform, NOT cosine weighted) */
    float sq = sqrtf(1.0f - r1 * r1);
    float phi = 6.2831854820251465f * r2;
    float sn, cs;
    if (libm) { sn = sinf(phi); cs = cosf(phi); } else orc_sincos(phi, &sn, &cs);
    float dx = cs * sq, dy = sn * sq, dz = r1;
    /* raygen.rgen:38: dir.x*T + dir.y*B + dir.z*N, left to right */
    for (int k = 0; k < 3; k++) out[k] = (T[k] * dx + B[k] * dy) + n[k] * dz;
}

/* ===== the frame ===================================================================== */

typedef struct job {
    const orc_scene *s; const orc_params *p; int mode, tid, nthreads;
    float *frame_color; orc_hit *first_hits; orc_counters cnt;
    uint32_t x0, y0, rw, rh; /* rectangle to render; output is rw x rh, row-major */
    atomic_uint *next_tile;  /* shared by the workers of one call: the next 16x16 tile to take */
    uint32_t tile_stride, tile_offset; /* only the 16x16 tiles t with t % tile_stride == tile_offset (1, 0: all of them) */
    uint32_t *ray_map;       /* optional: traceRayEXT calls of every pixel of the rectangle (per-rank counts of a tile-sharded job) */
} job;

static void render_pixel(job *jb, uint32_t px, uint32_t py)
{
    const orc_scene *s = jb->s; const orc_params *p = jb->p;
    float color[3] = { 0.0f, 0.0f, 0.0f };
    const uint64_t rays_before = jb->cnt.rays;
    for (uint32_t sample = 0; sample < p->spp_per_frame; sample++) { /* raygen.rgen:45 */
        uint32_t seed = orc_seed(px, py, sample, p->frame, p->spp_per_frame);
        float org[3], dir[3];
        orc_primary_ray(p, px, py, &seed, org, dir);
        float weight[3] = { 1.0f, 1.0f, 1.0f };
        for (uint32_t depth = 0; depth < p->max_depth; depth++) { /* raygen.rgen:62 */
            orc_hit h;
            orc_trace(s, jb->mode, org, dir, p->tmin, p->tmax, &h, &jb->cnt);
            if (jb->first_hits && sample == 0 && depth == 0)
                jb->first_hits[(size_t)(py - jb->y0) * jb->rw + (px - jb->x0)] = h;
            if (h.prim == ORC_MISS) { /* miss.rmiss:10-11 then raygen.rgen:76, 81-83 */
                for (int k = 0; k < 3; k++) color[k] = color[k] + weight[k] * p->env[k];
                break;
            }
            float pos[3], n[3], brdf[3], emi[3];
            orc_shade_hit(s, &h, pos, n, brdf, emi);
            if (!p->nee || depth == 0)
                for (int k = 0; k < 3; k++) color[k] = color[k] + weight[k] * emi[k]; /* :76 */
            /* (not at the path's last hit: its direct light stands for the emission the NEXT ray would find, and there is no
             * next ray -- the reference sums emission over the hits of rays 0 .. max_depth-1, raygen.rgen:62-83) */
            if (p->nee && s->n_lights && depth + 1u < p->max_depth) {
                /* next-event estimation (not in the reference): one point on one emitter, chosen by area; the
                 * contribution weight * brdf * Ke * cos_s |cos_l| / d^2 * total_area if the shadow ray is free */
                const float rl = orc_rand(&seed), ru = orc_rand(&seed), rv = orc_rand(&seed);
                const float pick = rl * s->light_area;
                /* first emitter whose running area exceeds pick (the last one if none does): binary search of the cdf */
                uint32_t li = 0, hi_ = s->n_lights - 1u;
                while (li < hi_) {
                    const uint32_t mid = (li + hi_) >> 1;
                    if (s->lights[16 * (size_t)mid + 15] > pick) hi_ = mid; else li = mid + 1u;
                }
                const float *L = s->lights + 16 * (size_t)li;
                const float su = sqrtf(ru);
                const float b0 = 1.0f - su, b1 = su * (1.0f - rv), b2 = su * rv;
                float d[3];
                for (int k = 0; k < 3; k++) d[k] = ((L[k] * b0 + L[3 + k] * b1) + L[6 + k] * b2) - pos[k];
                const float d2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
                if (d2 > 0.0f) {
                    const float dist = sqrtf(d2);
                    float wi[3] = { d[0] / dist, d[1] / dist, d[2] / dist };
                    const float cs = (wi[0] * n[0] + wi[1] * n[1]) + wi[2] * n[2];
                    const float cl = fabsf((wi[0] * L[9] + wi[1] * L[10]) + wi[2] * L[11]);
                    if (cs > 0.0f && cl > 0.0f) {
                        const float fgeo = ((cs * cl) / d2) * s->light_area;
                        orc_hit sh;
                        orc_trace(s, jb->mode, pos, wi, p->tmin, dist * 0.999f, &sh, &jb->cnt);
                        if (sh.prim == ORC_MISS)
                            for (int k = 0; k < 3; k++) color[k] = color[k] + ((weight[k] * brdf[k]) * L[12 + k]) * fgeo;
                    }
                }
            }
            for (int k = 0; k < 3; k++) org[k] = pos[k];                            /* :77 */
            float r1 = orc_rand(&seed); /* cos(theta) first, azimuth second (appendix B.3) */
            float r2 = orc_rand(&seed);
            orc_sample_direction(r1, r2, n, (int)p->libm_sincos, dir);              /* :78 */
            float dt = (dir[0] * n[0] + dir[1] * n[1]) + dir[2] * n[2];
            for (int k = 0; k < 3; k++)                                             /* :79-80 */
                weight[k] = weight[k] * ((brdf[k] * dt) / 0.15915493667125702f);
        }
    }
    float *out = jb->frame_color + 3 * ((size_t)(py - jb->y0) * jb->rw + (px - jb->x0));
    for (int k = 0; k < 3; k++) out[k] = color[k] / (float)p->spp_per_frame; /* :86 */
    if (jb->ray_map) jb->ray_map[(size_t)(py - jb->y0) * jb->rw + (px - jb->x0)] = (uint32_t)(jb->cnt.rays - rays_before);
}

/* Work distribution: the workers pull 16x16-pixel tiles from one shared counter (dynamic: the border of the image --
 * 44 % of the Cornell frame -- ends after one ray, so static row interleaving left most threads idle at the end), and
 * each keeps its job record, ray counters included, in a private copy on its own stack (the records of neighbouring
 * threads used to share cache lines that every traced ray wrote to).  Pixels are independent (raygen.rgen:47-48, 88-90),
 * so the order changes no output bit. */
#define ORC_TILE 16u
static void *worker(void *arg)
{
    job *shared = arg;
    job jb = *shared;
    const uint32_t tx_n = (jb.rw + ORC_TILE - 1u) / ORC_TILE, ty_n = (jb.rh + ORC_TILE - 1u) / ORC_TILE;
    const uint32_t n_tiles = tx_n * ty_n;
    for (;;) {
        const uint32_t t = atomic_fetch_add_explicit(jb.next_tile, 1u, memory_order_relaxed) * jb.tile_stride + jb.tile_offset;
        if (t >= n_tiles) break;
        const uint32_t ty = t / tx_n, tx = t - ty * tx_n;
        const uint32_t y1 = jb.y0 + (ty + 1u) * ORC_TILE < jb.y0 + jb.rh ? jb.y0 + (ty + 1u) * ORC_TILE : jb.y0 + jb.rh;
        const uint32_t x1 = jb.x0 + (tx + 1u) * ORC_TILE < jb.x0 + jb.rw ? jb.x0 + (tx + 1u) * ORC_TILE : jb.x0 + jb.rw;
        for (uint32_t y = jb.y0 + ty * ORC_TILE; y < y1; y++)
            for (uint32_t x = jb.x0 + tx * ORC_TILE; x < x1; x++) render_pixel(&jb, x, y);
    }
    shared->cnt = jb.cnt;
    return NULL;
}

uint64_t orc_render_frame(const orc_scene *s, const orc_params *p, int mode, int nthreads,
                          float *frame_color, orc_hit *first_hits, orc_counters *cnt)
{
    return orc_render_rect(s, p, mode, nthreads, 0, 0, p->width, p->height, frame_color, first_hits, cnt);
}

uint64_t orc_render_rect(const orc_scene *s, const orc_params *p, int mode, int nthreads, uint32_t x0, uint32_t y0,
                         uint32_t rw, uint32_t rh, float *frame_color, orc_hit *first_hits, orc_counters *cnt)
{
    return orc_render_rect_map(s, p, mode, nthreads, x0, y0, rw, rh, frame_color, first_hits, cnt, NULL);
}

static uint64_t render_tiles(const orc_scene *s, const orc_params *p, int mode, int nthreads, uint32_t x0, uint32_t y0,
                             uint32_t rw, uint32_t rh, float *frame_color, orc_hit *first_hits, orc_counters *cnt,
                             uint32_t *ray_map, uint32_t tile_stride, uint32_t tile_offset);

uint64_t orc_render_tile_subset(const orc_scene *s, const orc_params *p, int mode, int nthreads, uint32_t tile_stride,
                                uint32_t tile_offset, float *frame_color, orc_counters *cnt)
{
    if (tile_stride == 0u || tile_offset >= tile_stride) return 0;
    return render_tiles(s, p, mode, nthreads, 0, 0, p->width, p->height, frame_color, NULL, cnt, NULL, tile_stride, tile_offset);
}

uint64_t orc_render_rect_map(const orc_scene *s, const orc_params *p, int mode, int nthreads, uint32_t x0, uint32_t y0,
                             uint32_t rw, uint32_t rh, float *frame_color, orc_hit *first_hits, orc_counters *cnt,
                             uint32_t *ray_map)
{
    return render_tiles(s, p, mode, nthreads, x0, y0, rw, rh, frame_color, first_hits, cnt, ray_map, 1u, 0u);
}

static uint64_t render_tiles(const orc_scene *s, const orc_params *p, int mode, int nthreads, uint32_t x0, uint32_t y0,
                             uint32_t rw, uint32_t rh, float *frame_color, orc_hit *first_hits, orc_counters *cnt,
                             uint32_t *ray_map, uint32_t tile_stride, uint32_t tile_offset)
{
    if (x0 + rw > p->width || y0 + rh > p->height) return 0;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job jobs[256];
    pthread_t th[256];
    atomic_uint next_tile;
    atomic_init(&next_tile, 0u);
    for (int t = 0; t < nthreads; t++) {
        jobs[t].next_tile = &next_tile;
        jobs[t].s = s; jobs[t].p = p; jobs[t].mode = mode; jobs[t].tid = t; jobs[t].nthreads = nthreads;
        jobs[t].frame_color = frame_color; jobs[t].first_hits = first_hits;
        jobs[t].x0 = x0; jobs[t].y0 = y0; jobs[t].rw = rw; jobs[t].rh = rh;
        jobs[t].ray_map = ray_map;
        jobs[t].tile_stride = tile_stride; jobs[t].tile_offset = tile_offset;
        memset(&jobs[t].cnt, 0, sizeof(orc_counters));
    }
    if (nthreads == 1) worker(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    orc_counters tot = { 0, 0, 0 };
    for (int t = 0; t < nthreads; t++) {
        tot.rays += jobs[t].cnt.rays;
        tot.nodes_visited += jobs[t].cnt.nodes_visited;
        tot.tris_tested += jobs[t].cnt.tris_tested;
    }
    if (cnt) *cnt = tot;
    return tot.rays;
}

void orc_accumulate_f32(float *film, const float *frame_color, int32_t frame, uint64_t n_pixels)
{
    /* raygen.rgen:88-90, alpha dropped: new = (color + old*frame) / (frame+1) */
    const float f = (float)frame, f1 = (float)(frame + 1);
    for (uint64_t i = 0; i < 3 * n_pixels; i++) {
        /* frame 0 multiplies the undefined initial image by 0 in the reference; the float
         * film defines old*0 = 0 even for NaN/Inf garbage by never reading it at frame 0 */
        float old = frame == 0 ? 0.0f : film[i];
        film[i] = (frame_color[i] + old * f) / f1;
    }
}

static inline uint8_t to_unorm8(float c)
{
    if (!(c > 0.0f)) return 0; /* NaN and negatives -> 0 */
    if (c > 1.0f) c = 1.0f;
    return (uint8_t)(c * 255.0f + 0.5f);
}

void orc_accumulate_bgra8(uint8_t *bgra, const float *frame_color, int32_t frame, uint64_t n_pixels)
{
    /* rgba8 storage image backed by B8G8R8A8Unorm memory (raygen.rgen:7, main.cpp:483):
     * component names are preserved, bytes in memory are B,G,R,A */
    const float f = (float)frame, f1 = (float)(frame + 1);
    for (uint64_t i = 0; i < n_pixels; i++) {
        float oldc[4];
        oldc[0] = (float)bgra[4 * i + 2] / 255.0f; /* R */
        oldc[1] = (float)bgra[4 * i + 1] / 255.0f; /* G */
        oldc[2] = (float)bgra[4 * i + 0] / 255.0f; /* B */
        oldc[3] = (float)bgra[4 * i + 3] / 255.0f; /* A */
        float newc[4];
        for (int k = 0; k < 3; k++) {
            float old = frame == 0 ? 0.0f : oldc[k];
            newc[k] = (frame_color[3 * i + k] + old * f) / f1;
        }
        float olda = frame == 0 ? 0.0f : oldc[3];
        newc[3] = (1.0f + olda * f) / f1;
        bgra[4 * i + 2] = to_unorm8(newc[0]);
        bgra[4 * i + 1] = to_unorm8(newc[1]);
        bgra[4 * i + 0] = to_unorm8(newc[2]);
        bgra[4 * i + 3] = to_unorm8(newc[3]);
    }
}
