/*
 * pt_oracle.c - CPU restatement of the reference's radiance path.  TEST INFRASTRUCTURE.
 *
 * This file is the checker the HIP kernels are compared against.  Nothing in the
 * product (the package, the C-ABI library, the CLI) may call, link or import it; only
 * tests/, the smoke check and the benchmark's cpu_baseline leg do.
 *
 * It restates, in plain C and in the reference's own evaluation order, the algorithm of
 * rweyrauch/gpupathtracer's host-compilable hit / material / integrator code.  Every
 * function names the reference file:line it follows.  Arithmetic is written out one
 * operation at a time (no FMA contraction: build with -ffp-contract=off, no fast-math)
 * so that float results are those of the reference built by g++ for x86-64.
 *
 * Pinning status: the reference ships no tests, fixtures or golden vectors, and it cannot
 * be compiled in this image without stand-ins for CUDA headers it includes (<cuda.h>,
 * <math_constants.h>, <math_functions.h>), so there is no reference build beside this
 * file.  The restatement is pinned by the known-answer values that the structural survey
 * of the reference recorded from the reference's own code (RNG streams, the first
 * random_scene sphere centre, object counts, and FNV-1a hashes of whole rendered images);
 * tests/test_oracle_pins.py checks them.  Results no recorded value covers are
 * "parity unpinned" and are marked so in the tests.
 *
 * Input is the same flat PtSceneDesc (include/ptcore.h) that the product consumes, so
 * both paths see identical inputs.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ptcore.h"

/* include/ptRNG.h and friends use CUDART_PI_F from CUDA's math_constants.h */
#define PI_F 3.141592654f

/* ---- work model -------------------------------------------------------------
 * Built with -DPT_ORACLE_COUNT (liboracle_count.so) every function adds the arithmetic it
 * performs to per-thread counters, so that one render yields the algorithmic work of the
 * reference's own evaluation: FL = fp32 add/sub/mul/compare/min/max/div/sqrt (one each, no
 * FMA credit), TR = libm calls (sin, cos, log, pow, atan2, asin), IO = integer operations of
 * the generators and the noise lattice, plus event counts.  The plain build compiles them out. */
#ifdef PT_ORACLE_COUNT
typedef struct {
    uint64_t flop, libm, iop, draws, paths, world_hits, rect_tests, sphere_tests, triangle_tests,
             aabb_tests, diffuse_scatters, accepted_hits, tree_nodes, tree_leaves;
} work_t;
#define WORK_FIELDS 14
static __thread work_t W;
static work_t W_total;
#define FL(n) (W.flop += (n))
#define TR(n) (W.libm += (n))
#define IO(n) (W.iop += (n))
#define EV(field) (W.field += 1)
#else
#define FL(n) ((void)0)
#define TR(n) ((void)0)
#define IO(n) ((void)0)
#define EV(field) ((void)0)
#endif

typedef struct { float x, y, z; } v3;
typedef struct { v3 o, d; float time; } ray_t;
typedef struct { uint64_t state, inc; } pcg_t;

/* HitRecord (include/ptHitable.h:23-30) plus the id of the winning leaf. */
typedef struct {
    float t; v3 p; v3 n; int mat; float u, v; int prim;
} hitrec;

/* ---- Vector3 (include/ptVector3.h) --------------------------------------- */
static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { FL(3); return V(a.x + b.x, a.y + b.y, a.z + b.z); }          /* :136 */
static inline v3 vsub(v3 a, v3 b) { FL(3); return V(a.x - b.x, a.y - b.y, a.z - b.z); }          /* :142 */
static inline v3 vmul(v3 a, v3 b) { FL(3); return V(a.x * b.x, a.y * b.y, a.z * b.z); }          /* :148 */
static inline v3 smul(float s, v3 a) { FL(3); return V(s * a.x, s * a.y, s * a.z); }             /* :160 */
static inline v3 vmuls(v3 a, float s) { FL(3); return V(a.x * s, a.y * s, a.z * s); }            /* :172 */
static inline v3 vdivs(v3 a, float s) { FL(3); return V(a.x / s, a.y / s, a.z / s); }            /* :166 true divides */
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }                               /* :43 */
static inline float vdot(v3 a, v3 b) { FL(5); return a.x * b.x + a.y * b.y + a.z * b.z; }        /* :178 */
static inline float vsqlen(v3 a) { FL(5); return a.x * a.x + a.y * a.y + a.z * a.z; }            /* :103 */
static inline float vlen(v3 a) { FL(1); return sqrtf(vsqlen(a)); }                               /* :98 */
static inline v3 vcross(v3 a, v3 b)                                                       /* :184 */
{
    FL(9);
    return V(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
static inline v3 vunit(v3 a) { return vdivs(a, vlen(a)); }                                /* :192 */
static inline v3 vdiv_eq(v3 a, float s)                                                   /* :89 operator/= */
{
    const float inv = 1 / s;
    FL(4);
    return V(a.x * inv, a.y * inv, a.z * inv);
}
static inline v3 vmake_unit(v3 a)                                                         /* :108 */
{
    float k = 1 / vlen(a);
    FL(4);
    return V(a.x * k, a.y * k, a.z * k);
}
static inline v3 vreflect(v3 v, v3 n) { FL(1); return vsub(v, smul(2 * vdot(v, n), n)); }        /* :198 */
static int vrefract(v3 v, v3 n, float ni_over_nt, v3* refracted)                          /* :204 */
{
    v3 uv = vunit(v);
    float dt = vdot(uv, n);
    float disc = 1 - ni_over_nt * ni_over_nt * (1 - dt * dt);
    FL(6);
    if (disc > 0) {
        FL(1);
        *refracted = vsub(smul(ni_over_nt, vsub(uv, vmuls(n, dt))), vmuls(n, sqrtf(disc)));
        return 1;
    }
    return 0;
}
static inline v3 ray_at(const ray_t* r, float t) { return vadd(r->o, smul(t, r->d)); }    /* ptRay.h:36 */

/* ---- RNG (include/ptRNG.h) ------------------------------------------------ */
static uint32_t pcg_u32(pcg_t* g)                                                         /* :122-131 */
{
    IO(11);
    uint64_t old = g->state;
    g->state = old * 0x5851f42d4c957f2dULL + g->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
static void pcg_seed(pcg_t* g, uint64_t seed)                                             /* :101-108 */
{
    g->state = 0u;
    g->inc = (seed << 1u) | 1u;
    pcg_u32(g);
    g->state += 0x853c49e6748fea9bULL;
    pcg_u32(g);
}
static float pcg_rand(pcg_t* g)                                                           /* :110-113 */
{
    FL(2); EV(draws);
    return fminf(0.99999994f, (float)pcg_u32(g) * 2.3283064365386963e-10f);
}
static float simple_rng_rand(unsigned* s0, unsigned* s1)                                  /* :38-54 */
{
    *s0 = 36969 * ((*s0) & 65535) + ((*s0) >> 16);
    *s1 = 18000 * ((*s1) & 65535) + ((*s1) >> 16);
    unsigned ires = ((*s0) << 16) + (*s1);
    union { float f; unsigned ui; } res;
    res.ui = (ires & 0x007fffff) | 0x40000000;
    return (res.f - 2.0f) / 2.0f;
}

static v3 random_in_unit_sphere(pcg_t* g)                                                 /* :137-143 */
{
    const float phi = pcg_rand(g) * 2 * PI_F;
    const float z = 1 - 2 * pcg_rand(g);
    const float r = sqrtf(fmaxf(0, 1 - z * z));
    FL(10); TR(2);
    return V(r * cosf(phi), r * sinf(phi), z);
}
static v3 random_in_unit_disk(pcg_t* g)                                                   /* :145-150 */
{
    const float r = sqrtf(pcg_rand(g));
    const float theta = pcg_rand(g) * 2 * PI_F;
    FL(5); TR(2);
    return V(r * cosf(theta), r * sinf(theta), 0);
}
static v3 random_cosine_direction(pcg_t* g)                                               /* :152-161 */
{
    float r1 = pcg_rand(g);
    float r2 = pcg_rand(g);
    float z = sqrtf(1 - r2);
    float phi = 2 * PI_F * r1;
    float x = cosf(phi) * 2 * sqrtf(r2);   /* the factor 2 is the reference's */
    float y = sinf(phi) * 2 * sqrtf(r2);
    FL(10); TR(2);
    return V(x, y, z);
}
static v3 random_to_unit_sphere(float radius, float dist_sqrd, pcg_t* g)                  /* :163-172 */
{
    float r1 = pcg_rand(g);
    float r2 = pcg_rand(g);
    float z = 1 + r2 * (sqrtf(1 - radius * radius / dist_sqrd) - 1);
    float phi = 2 * PI_F * r1;
    float x = cosf(phi) * sqrtf(1 - z * z);
    float y = sinf(phi) * sqrtf(1 - z * z);
    FL(17); TR(2);
    return V(x, y, z);
}

/* ---- ONB (include/ptONB.h:27-39) ------------------------------------------ */
typedef struct { v3 u, v, w; } onb_t;
static void onb_from_w(onb_t* b, v3 n)
{
    b->w = vunit(n);
    v3 a;
    FL(1);
    if (fabsf(b->w.x) > 0.9f) a = V(0, 1, 0); else a = V(1, 0, 0);
    b->v = vunit(vcross(b->w, a));
    b->u = vcross(b->w, b->v);
}
static v3 onb_local(const onb_t* b, v3 a)
{
    return vadd(vadd(smul(a.x, b->u), smul(a.y, b->v)), smul(a.z, b->w));
}

/* ---- Perlin noise (src/ptNoise.cu) ---------------------------------------- */
/* Ken Perlin's published 2002 reference permutation, doubled (src/ptNoise.cu:47-112). */
static const unsigned char PERM[256] = {
    151,160,137,91,90,15,131,13,201,95,96,53,194,233,7,225,140,36,103,30,69,142,8,99,37,240,21,10,23,
    190,6,148,247,120,234,75,0,26,197,62,94,252,219,203,117,35,11,32,57,177,33,88,237,149,56,87,174,
    20,125,136,171,168,68,175,74,165,71,134,139,48,27,166,77,146,158,231,83,111,229,122,60,211,133,
    230,220,105,92,41,55,46,245,40,244,102,143,54,65,25,63,161,1,216,80,73,209,76,132,187,208,89,18,
    169,200,196,135,130,116,188,159,86,164,100,109,198,173,186,3,64,52,217,226,250,124,123,5,202,38,
    147,118,126,255,82,85,212,207,206,59,227,47,16,58,17,182,189,28,42,223,183,170,213,119,248,152,
    2,44,154,163,70,221,153,101,155,167,43,172,9,129,22,39,253,19,98,108,110,79,113,224,232,178,185,
    112,104,218,246,97,228,251,34,242,193,238,210,144,12,191,179,162,241,81,51,145,235,249,14,239,
    107,49,192,214,31,181,199,106,157,184,84,204,176,115,121,50,45,127,4,150,254,138,236,205,93,222,
    114,67,29,24,72,243,141,128,195,78,66,215,61,156,180
};
static inline int perm(int i) { return PERM[i & 255]; }   /* table is periodic with period 256 */

static inline float fade(float t) { FL(7); return t * t * t * (t * (t * 6 - 15) + 10); }         /* :19-23 */
static inline float lerp(float t, float a, float b) { FL(3); return a + t * (b - a); }           /* :25-29 */
static inline float grad(int hash, float x, float y, float z)                             /* :31-38 */
{
    FL(1); IO(8);
    int h = hash & 15;
    float u = ((h < 8) || (h == 12) || (h == 13)) ? x : y;
    float v = ((h < 4) || (h == 12) || (h == 13)) ? y : z;
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}
static float noise3(v3 p)                                                                 /* :114-140 */
{
    FL(18); IO(30);          /* 3 floors, 3 fractions, 12 lattice offsets */
    int xi = (int)floorf(p.x) & 255;
    int yi = (int)floorf(p.y) & 255;
    int zi = (int)floorf(p.z) & 255;
    float x = p.x - floorf(p.x);
    float y = p.y - floorf(p.y);
    float z = p.z - floorf(p.z);
    float u = fade(x), v = fade(y), w = fade(z);
    int A = perm(xi) + yi, AA = perm(A) + zi, AB = perm(A + 1) + zi;
    int B = perm(xi + 1) + yi, BA = perm(B) + zi, BB = perm(B + 1) + zi;
    return lerp(w, lerp(v, lerp(u, grad(perm(AA), x, y, z),
                                   grad(perm(BA), x - 1, y, z)),
                           lerp(u, grad(perm(AB), x, y - 1, z),
                                   grad(perm(BB), x - 1, y - 1, z))),
                   lerp(v, lerp(u, grad(perm(AA + 1), x, y, z - 1),
                                   grad(perm(BA + 1), x - 1, y, z - 1)),
                           lerp(u, grad(perm(AB + 1), x, y - 1, z - 1),
                                   grad(perm(BB + 1), x - 1, y - 1, z - 1))));
}
static float turbulence(v3 p, int depth)                                                  /* :142-154 */
{
    float accum = 0;
    v3 tp = p;
    float weight = 1;
    for (int i = 0; i < depth; i++) {
        FL(3);
        accum += weight * noise3(tp);
        weight *= 0.5f;
        tp = vmuls(tp, 2.0f);
    }
    return fabsf(accum);
}

/* ---- textures (include/ptTexture.h) --------------------------------------- */
static v3 texture_value(const PtSceneDesc* sc, int tex, float u, float v, v3 p)
{
    const PtTexture* t = &sc->textures[tex];
    switch (t->type) {
    case PT_TEX_CONSTANT:                                                                 /* :47-50 */
        return V(t->color[0], t->color[1], t->color[2]);
    case PT_TEX_CHECKER: {                                                                /* :90-97 */
        FL(6); TR(3);
        float sines = sinf(t->scale * p.x) * sinf(t->scale * p.y) * sinf(t->scale * p.z);
        if (sines < 0) return texture_value(sc, t->odd, u, v, p);
        return texture_value(sc, t->even, u, v, p);
    }
    case PT_TEX_NOISE: {                                                                  /* :142-146 */
        FL(5); TR(1);
        float n = 0.5f * (1 + sinf(t->scale * p.z + 10 * turbulence(p, 7)));
        return V(n, n, n);
    }
    case PT_TEX_IMAGE: {                                                                  /* :187-199 */
        FL(7); IO(10);
        int nx = t->nx, ny = t->ny;
        int i = (int)(u * nx);
        int j = (int)((1 - v) * ny - 0.001f);
        if (i < 0) i = 0;
        if (j < 0) j = 0;
        if (i > nx - 1) i = nx - 1;
        if (j > ny - 1) j = ny - 1;
        const uint8_t* d = sc->texels + t->texel_offset;
        float r = (int)d[3 * i + 3 * nx * j + 0] / 255.0f;
        float g = (int)d[3 * i + 3 * nx * j + 1] / 255.0f;
        float b = (int)d[3 * i + 3 * nx * j + 2] / 255.0f;
        return V(r, g, b);
    }
    }
    return V(0, 0, 0);
}

/* ---- leaf geometry -------------------------------------------------------- */

static void get_uv(v3 p, float* u, float* v)                                              /* ptSphere.h:22-28 */
{
    FL(5); TR(2);
    float phi = atan2f(p.z, p.x);
    float theta = asinf(p.y);
    *u = 1 - (phi + PI_F) / (2 * PI_F);
    *v = (theta + PI_F / 2) / PI_F;
}

static int sphere_hit(const float* q, int mat, const ray_t* r, float tmin, float tmax, hitrec* rec)
{                                                                                         /* ptSphere.cu:11-40 */
    v3 center = V(q[0], q[1], q[2]);
    float radius = q[3];
    v3 oc = vsub(r->o, center);
    float a = vdot(r->d, r->d);
    float b = vdot(oc, r->d);
    float c = vdot(oc, oc) - radius * radius;
    float disc = b * b - a * c;
    FL(6); EV(sphere_tests);
    if (disc > 0) {
        float temp = (-b - sqrtf(disc)) / a;
        FL(5);
        if (temp < tmax && temp > tmin) {
            rec->t = temp;
            rec->p = ray_at(r, rec->t);
            rec->n = vdivs(vsub(rec->p, center), radius);
            rec->mat = mat;
            return 1;                     /* rec->u, rec->v are NOT written (stale), as in the reference */
        }
        temp = (-b + sqrtf(disc)) / a;
        FL(5);
        if (temp < tmax && temp > tmin) {
            rec->t = temp;
            rec->p = ray_at(r, rec->t);
            rec->n = vdivs(vsub(rec->p, center), radius);
            rec->mat = mat;
            return 1;
        }
    }
    return 0;
}

static v3 msphere_center(const float* q, float time)                                      /* ptSphere.h:106-109 */
{
    v3 c0 = V(q[0], q[1], q[2]), c1 = V(q[3], q[4], q[5]);
    float t0 = q[6], t1 = q[7];
    FL(3);
    return vadd(c0, smul((time - t0) / (t1 - t0), vsub(c1, c0)));
}
static int msphere_hit(const float* q, int mat, const ray_t* r, float tmin, float tmax, hitrec* rec)
{                                                                                         /* ptSphere.cu:66-97 */
    float radius = q[8];
    v3 oc = vsub(r->o, msphere_center(q, r->time));
    float a = vdot(r->d, r->d);
    float b = vdot(oc, r->d);
    float c = vdot(oc, oc) - radius * radius;
    float disc = b * b - a * c;
    FL(6); EV(sphere_tests);
    if (disc > 0) {
        float temp = (-b - sqrtf(disc)) / a;
        FL(5);
        if (temp < tmax && temp > tmin) {
            rec->t = temp;
            rec->p = ray_at(r, rec->t);
            rec->n = vdivs(vsub(rec->p, msphere_center(q, r->time)), radius);
            rec->mat = mat;
            get_uv(rec->p, &rec->u, &rec->v);
            return 1;
        }
        temp = (-b + sqrtf(disc)) / a;
        FL(5);
        if (temp < tmax && temp > tmin) {
            rec->t = temp;
            rec->p = ray_at(r, rec->t);
            rec->n = vdivs(vsub(rec->p, msphere_center(q, r->time)), radius);
            rec->mat = mat;
            get_uv(rec->p, &rec->u, &rec->v);
            return 1;
        }
    }
    return 0;
}

/* Axis-aligned rectangles (src/ptRectangle.cu:12-28,66-82,120-136).
 * axis k: 2 = XY (plane z=k), 1 = XZ (plane y=k), 0 = YZ (plane x=k). */
static int rect_hit(int kaxis, float a0, float a1, float b0, float b1, float k, int mat,
                    const ray_t* r, float t0, float t1, hitrec* rec)
{
    float ok, dk, oa, da, ob, db;
    if (kaxis == 2)      { ok = r->o.z; dk = r->d.z; oa = r->o.x; da = r->d.x; ob = r->o.y; db = r->d.y; }
    else if (kaxis == 1) { ok = r->o.y; dk = r->d.y; oa = r->o.x; da = r->d.x; ob = r->o.z; db = r->d.z; }
    else                 { ok = r->o.x; dk = r->d.x; oa = r->o.y; da = r->d.y; ob = r->o.z; db = r->d.z; }
    float t = (k - ok) / dk;
    FL(4); EV(rect_tests);
    if (t < t0 || t > t1) return 0;
    float a = oa + t * da;
    float b = ob + t * db;
    FL(8);
    if (a < a0 || a > a1 || b < b0 || b > b1) return 0;
    FL(6);
    rec->u = (a - a0) / (a1 - a0);
    rec->v = (b - b0) / (b1 - b0);
    rec->t = t;
    rec->mat = mat;
    rec->p = ray_at(r, t);
    rec->n = (kaxis == 2) ? V(0, 0, 1) : (kaxis == 1) ? V(0, 1, 0) : V(1, 0, 0);
    return 1;
}

/* Box = HitableList of six rectangles, three of them under FlipNormals
 * (include/ptRectangle.h:199-217), searched as src/ptHitableList.cu:12-27 does. */
static int box_hit(const float* q, int mat, const ray_t* r, float t0, float t1, hitrec* rec)
{
    const float x0 = q[0], y0 = q[1], z0 = q[2], x1 = q[3], y1 = q[4], z1 = q[5];
    hitrec temp = *rec;            /* stands for the list's own temp_rec */
    int hit_anything = 0;
    float closest = t1;
    for (int f = 0; f < 6; f++) {
        int h;
        switch (f) {
        case 0: h = rect_hit(2, x0, x1, y0, y1, z1, mat, r, t0, closest, &temp); break;
        case 1: h = rect_hit(2, x0, x1, y0, y1, z0, mat, r, t0, closest, &temp); break;
        case 2: h = rect_hit(1, x0, x1, z0, z1, y1, mat, r, t0, closest, &temp); break;
        case 3: h = rect_hit(1, x0, x1, z0, z1, y0, mat, r, t0, closest, &temp); break;
        case 4: h = rect_hit(0, y0, y1, z0, z1, x1, mat, r, t0, closest, &temp); break;
        default: h = rect_hit(0, y0, y1, z0, z1, x0, mat, r, t0, closest, &temp); break;
        }
        if (h) {
            if (f & 1) temp.n = vneg(temp.n);          /* FlipNormals::hit, ptRectangle.h:148-156 */
            hit_anything = 1;
            closest = temp.t;
            *rec = temp;
        }
    }
    return hit_anything;
}

static int triangle_hit(const PtSceneDesc* sc, const PtShape* s, const ray_t* r,
                        float t_min, float t_max, hitrec* rec)
{                                                                                         /* ptTriangle.cu:28-82 */
    const float* q = s->p;
    v3 v0 = V(q[0], q[1], q[2]), v1 = V(q[3], q[4], q[5]), v2 = V(q[6], q[7], q[8]);
    v3 edge1 = vsub(v1, v0);
    v3 edge2 = vsub(v2, v0);
    v3 pvec = vcross(r->d, edge2);
    float det = vdot(edge1, pvec);
    FL(1); EV(triangle_tests);
    if (det < 0.0001f) return 0;
    v3 tvec = vsub(r->o, v0);
    float u = vdot(tvec, pvec);
    FL(2);
    if (u < 0 || u > det) return 0;
    v3 qvec = vcross(tvec, edge1);
    float v = vdot(r->d, qvec);
    FL(3);
    if (v < 0 || u + v > det) return 0;
    float t = vdot(edge2, qvec);             /* NOT divided by det, as in the reference */
    FL(2);
    if (t < t_min || t > t_max) return 0;
    FL(17);
    const float inv_det = 1 / det;
    u *= inv_det;
    v *= inv_det;
    rec->t = t;
    rec->p = vadd(vadd(smul(1 - u - v, v0), smul(u, v1)), smul(v, v2));
    rec->n = vmake_unit(vcross(edge1, edge2));
    rec->mat = s->material;
    float b0 = (float)(1.0 - (double)u - (double)v);       /* Vector3f bary(1.0 - u - v, u, v) */
    if (s->tri_uv >= 0) {
        const PtTriUV* tu = &sc->tri_uvs[s->tri_uv];       /* calcTexCoord, ptTriangle.cu:98-101 */
        rec->u = tu->uv0[0] * b0 + tu->uv1[0] * u + tu->uv2[0] * v;
        rec->v = tu->uv0[1] * b0 + tu->uv1[1] * u + tu->uv2[1] * v;
    } else {
        rec->u = 0.0f * b0 + 0.0f * u + 0.0f * v;
        rec->v = 0.0f * b0 + 0.0f * u + 0.0f * v;
    }
    return 1;
}

static int bare_shape_hit(const PtSceneDesc* sc, const PtShape* s, const ray_t* r,
                          float tmin, float tmax, hitrec* rec)
{
    switch (s->type) {
    case PT_SHAPE_SPHERE:        return sphere_hit(s->p, s->material, r, tmin, tmax, rec);
    case PT_SHAPE_MOVING_SPHERE: return msphere_hit(s->p, s->material, r, tmin, tmax, rec);
    case PT_SHAPE_RECT_XY: return rect_hit(2, s->p[0], s->p[1], s->p[2], s->p[3], s->p[4], s->material, r, tmin, tmax, rec);
    case PT_SHAPE_RECT_XZ: return rect_hit(1, s->p[0], s->p[1], s->p[2], s->p[3], s->p[4], s->material, r, tmin, tmax, rec);
    case PT_SHAPE_RECT_YZ: return rect_hit(0, s->p[0], s->p[1], s->p[2], s->p[3], s->p[4], s->material, r, tmin, tmax, rec);
    case PT_SHAPE_BOX:           return box_hit(s->p, s->material, r, tmin, tmax, rec);
    case PT_SHAPE_TRIANGLE:      return triangle_hit(sc, s, r, tmin, tmax, rec);
    }
    return 0;
}

/* Translate::hit / RotateY::hit (include/ptRectangle.h:272-281,368-393), outermost first. */
static int xformed_hit(const PtSceneDesc* sc, const PtXform* xf, int nxf, int k,
                       const PtShape* s, int bvh, const ray_t* r, float tmin, float tmax,
                       hitrec* rec, pcg_t* rng);

static int bvh_node_hit(const PtSceneDesc* sc, int node, const ray_t* r, float tmin, float tmax,
                        hitrec* rec, pcg_t* rng);

static int shape_hit(const PtSceneDesc* sc, int shape, const ray_t* r, float tmin, float tmax,
                     hitrec* rec, pcg_t* rng)
{
    const PtShape* s = &sc->shapes[shape];
    int h = xformed_hit(sc, sc->xforms + s->xform_first, s->xform_count, 0, s, -1, r, tmin, tmax, rec, rng);
    if (h) {
        if (s->flags & PT_FLIP_NORMAL) rec->n = vneg(rec->n);   /* ptRectangle.h:148-156 */
        rec->prim = shape;
    }
    return h;
}

static int xformed_hit(const PtSceneDesc* sc, const PtXform* xf, int nxf, int k,
                       const PtShape* s, int bvh, const ray_t* r, float tmin, float tmax,
                       hitrec* rec, pcg_t* rng)
{
    if (k == nxf) {
        if (s) return bare_shape_hit(sc, s, r, tmin, tmax, rec);
        return bvh_node_hit(sc, bvh, r, tmin, tmax, rec, rng);
    }
    const PtXform* x = &xf[k];
    if (x->type == PT_XFORM_TRANSLATE) {
        v3 off = V(x->v[0], x->v[1], x->v[2]);
        ray_t moved = *r;
        moved.o = vsub(r->o, off);
        if (xformed_hit(sc, xf, nxf, k + 1, s, bvh, &moved, tmin, tmax, rec, rng)) {
            rec->p = vadd(rec->p, off);
            return 1;
        }
        return 0;
    } else {
        const float sinT = x->v[0], cosT = x->v[1];
        ray_t rot = *r;
        FL(12);
        rot.o.x = cosT * r->o.x - sinT * r->o.z;
        rot.o.z = sinT * r->o.x + cosT * r->o.z;
        rot.d.x = cosT * r->d.x - sinT * r->d.z;
        rot.d.z = sinT * r->d.x + cosT * r->d.z;
        if (xformed_hit(sc, xf, nxf, k + 1, s, bvh, &rot, tmin, tmax, rec, rng)) {
            v3 p = rec->p, n = rec->n;
            FL(12);
            p.x = cosT * rec->p.x + sinT * rec->p.z;
            p.z = -sinT * rec->p.x + cosT * rec->p.z;
            n.x = cosT * rec->n.x + sinT * rec->n.z;
            n.z = -sinT * rec->n.x + cosT * rec->n.z;
            rec->p = p;
            rec->n = n;
            return 1;
        }
        return 0;
    }
}

/* AABB::hit (include/ptAABB.h:30-48) */
static int aabb_hit(const float* bmin, const float* bmax, const ray_t* r, float tmin, float tmax)
{
    const float o[3] = { r->o.x, r->o.y, r->o.z }, d[3] = { r->d.x, r->d.y, r->d.z };
    EV(aabb_tests);
    for (int a = 0; a < 3; a++) {
        FL(9);
        const float invD = 1 / d[a];
        float t0 = (bmin[a] - o[a]) * invD;
        float t1 = (bmax[a] - o[a]) * invD;
        if (invD < 0) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return 0;
    }
    return 1;
}

/* BVH::hit (src/ptBVH.cu:45-74): both children, same range, right wins unless left is nearer. */
static int bvh_child_hit(const PtSceneDesc* sc, int child, const ray_t* r, float tmin, float tmax,
                         hitrec* rec, pcg_t* rng)
{
    if (child >= 0) return bvh_node_hit(sc, child, r, tmin, tmax, rec, rng);
    return shape_hit(sc, ~child, r, tmin, tmax, rec, rng);
}
static int bvh_node_hit(const PtSceneDesc* sc, int node, const ray_t* r, float tmin, float tmax,
                        hitrec* rec, pcg_t* rng)
{
    const PtBvhNode* n = &sc->bvh_nodes[node];
    if (aabb_hit(n->bmin, n->bmax, r, tmin, tmax)) {
        hitrec lrec = *rec, rrec = *rec;
        int hl = bvh_child_hit(sc, n->left, r, tmin, tmax, &lrec, rng);
        int hr = bvh_child_hit(sc, n->right, r, tmin, tmax, &rrec, rng);
        if (hl && hr) {
            if (lrec.t < rrec.t) *rec = lrec; else *rec = rrec;
            return 1;
        } else if (hl) { *rec = lrec; return 1; }
        else if (hr) { *rec = rrec; return 1; }
        return 0;
    }
    return 0;
}

/* ConstantMedium::hit (include/ptMedium.h:31-55) */
static int medium_hit(const PtSceneDesc* sc, int m, const ray_t* r, float t0, float t1,
                      hitrec* rec, pcg_t* rng)
{
    const PtMedium* med = &sc->media[m];
    hitrec rec1, rec2;
    memset(&rec1, 0, sizeof rec1);
    memset(&rec2, 0, sizeof rec2);
    if (shape_hit(sc, med->boundary, r, -FLT_MAX, FLT_MAX, &rec1, rng)) {
        FL(1);
        if (shape_hit(sc, med->boundary, r, (float)((double)rec1.t + 0.0001), FLT_MAX, &rec2, rng)) {
            FL(4);
            if (rec1.t < t0) rec1.t = t0;
            if (rec2.t > t1) rec2.t = t1;
            if (rec1.t > rec2.t) return 0;
            if (rec1.t < 0) rec1.t = 0;
            float dist_inside = (rec2.t - rec1.t) * vlen(r->d);
            float hit_dist = -(1 / med->density) * logf(pcg_rand(rng));
            FL(5); TR(1);
            if (hit_dist < dist_inside) {
                FL(2);
                rec->t = rec1.t + hit_dist / vlen(r->d);
                rec->p = ray_at(r, rec->t);
                rec->n = V(1, 0, 0);
                rec->mat = med->material;
                rec->prim = PT_ID_MEDIUM_BASE + m;
                return 1;
            }
        }
    }
    return 0;
}

#ifdef PT_ORACLE_COUNT
/* ---- work of a search through the product's tree (counting build only) ---------------------------
 * The product does not walk long lists entry by entry: it searches a binary tree it built over the
 * list's leading run of shapes (64-byte nodes holding both children's boxes) and walks only the
 * rest of the list in order.  To price THAT search, the tree is handed over here (copied back from
 * the device by the caller) and every world_hit counts the work of a replay of it -- node visits,
 * leaf tests with their arithmetic -- instead of the work of the flat walk.  The hit itself still
 * comes from the flat walk unless the caller asks for the replay's own answer (a 2 M-entry list
 * would take the flat walk seconds per ray); the counters, not the hit, are what this is for. */
typedef struct { float lmin[3], lmax[3], rmin[3], rmax[3]; int32_t left, right, pad0, pad1; } tree_node;
static const tree_node* T_nodes = NULL;
static int T_root = 0, T_cull = 0, T_prefix = 0, T_trust = 0, T_wide = 0;
/* wide != 0: price the search as the product's 4-wide walk of the same tree -- every second level folded
 * away, a node holds the boxes of its up to four grandchildren (or of a child that is a leaf) and one
 * 64-byte fetch decides four descents; tree_nodes then counts the wide nodes visited. */
void oracle_count_tree_form(int wide) { T_wide = wide; }
void oracle_count_through_tree(const void* nodes64, int root, int cull, int prefix, int trust_replay)
{
    T_nodes = (const tree_node*)nodes64; T_root = root; T_cull = cull; T_prefix = prefix; T_trust = trust_replay;
}
static int tree_slab(const float* bmin, const float* bmax, v3 o, v3 inv, float tmin, float tmax, float* tnear)
{
    float tx0 = (bmin[0] - o.x) * inv.x, tx1 = (bmax[0] - o.x) * inv.x;
    float ty0 = (bmin[1] - o.y) * inv.y, ty1 = (bmax[1] - o.y) * inv.y;
    float tz0 = (bmin[2] - o.z) * inv.z, tz1 = (bmax[2] - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tmax));
    FL(27);                      /* 6 sub, 6 mul, 12 min/max, 2 mul, 1 compare */
    *tnear = tn;
    return tn * 0.9999990f <= tf * 1.0000010f;
}
static int shape_hit(const PtSceneDesc* sc, int shape, const ray_t* r, float tmin, float tmax, hitrec* rec, pcg_t* rng);
static int inclusive_kind(int type) { return type != PT_SHAPE_SPHERE && type != PT_SHAPE_MOVING_SPHERE; }
/* near child first, pruned by the best distance unless triangles are in the tree (their reported t is
 * not a distance); a later list entry wins an exact tie iff it is of an inclusive kind */
static int tree_replay(const PtSceneDesc* sc, const ray_t* r, float tmin, float tmax, hitrec* rec, pcg_t* rng)
{
    int stack[128], sp = 0, cur = T_root, any = 0;
    float best = tmax;
    v3 inv = V(1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z);
    FL(3);
    for (;;) {
        if (cur >= 0 && T_wide) {
            const tree_node* n = &T_nodes[cur];
            const float lim = T_cull ? best : tmax, from = T_cull ? tmin : 0.0f;
            int kid[4], nk = 0, next = 0, have = 0;
            const float* kmin[4]; const float* kmax[4];
            float tnext = 0.0f;
            EV(tree_nodes);
            for (int side = 0; side < 2; side++) {
                const int c = side ? n->right : n->left;
                if (c >= 0) {
                    const tree_node* m = &T_nodes[c];
                    kid[nk] = m->left; kmin[nk] = m->lmin; kmax[nk] = m->lmax; nk++;
                    kid[nk] = m->right; kmin[nk] = m->rmin; kmax[nk] = m->rmax; nk++;
                } else {
                    kid[nk] = c; kmin[nk] = side ? n->rmin : n->lmin; kmax[nk] = side ? n->rmax : n->lmax; nk++;
                }
            }
            for (int k = 0; k < nk; k++) {
                float t;
                if (!tree_slab(kmin[k], kmax[k], r->o, inv, from, lim, &t)) continue;
                if (!have) { have = 1; next = kid[k]; tnext = t; continue; }
                FL(1);
                if (t < tnext) { if (sp < 128) stack[sp++] = next; next = kid[k]; tnext = t; }
                else if (sp < 128) stack[sp++] = kid[k];
            }
            if (have) { cur = next; continue; }
        } else if (cur >= 0) {
            const tree_node* n = &T_nodes[cur];
            float tl, tr;
            const float lim = T_cull ? best : tmax, from = T_cull ? tmin : 0.0f;
            EV(tree_nodes);
            int hl = tree_slab(n->lmin, n->lmax, r->o, inv, from, lim, &tl);
            int hr = tree_slab(n->rmin, n->rmax, r->o, inv, from, lim, &tr);
            int cl = n->left, cr = n->right;
            if (hl && hr) {
                FL(1);
                if (tr < tl) { int tmp = cl; cl = cr; cr = tmp; }
                if (sp < 128) stack[sp++] = cr;
                cur = cl;
                continue;
            }
            if (hl) { cur = cl; continue; }
            if (hr) { cur = cr; continue; }
        } else {
            hitrec temp;
            memset(&temp, 0, sizeof temp);
            EV(tree_leaves);
            if (shape_hit(sc, ~cur, r, tmin, tmax, &temp, rng)) {
                int better = !any || temp.t < best;
                if (any && temp.t == best) better = (~cur > rec->prim) ? inclusive_kind(sc->shapes[~cur].type) : !inclusive_kind(sc->shapes[rec->prim].type);
                FL(2);
                if (better) { any = 1; best = temp.t; *rec = temp; }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return any;
}
#endif

/* The top-level HitableList::hit (src/ptHitableList.cu:12-27). */
static int world_hit_list(const PtSceneDesc* sc, const ray_t* r, float tmin, float tmax,
                          hitrec* rec, pcg_t* rng, int first, int hit_anything, float closest);
static int world_hit(const PtSceneDesc* sc, const ray_t* r, float tmin, float tmax,
                     hitrec* rec, pcg_t* rng)
{
#ifdef PT_ORACLE_COUNT
    if (T_nodes) {
        EV(world_hits);
        /* the work: a replay of the tree over the list's leading run, then the rest of the list in order */
        pcg_t count_rng = *rng;
        hitrec replay;
        memset(&replay, 0, sizeof replay);
        int any = tree_replay(sc, r, tmin, tmax, &replay, &count_rng);
        any = world_hit_list(sc, r, tmin, tmax, &replay, &count_rng, T_prefix, any, any ? replay.t : tmax);
        if (T_trust) { *rng = count_rng; if (any) *rec = replay; return any; }
        /* the answer: the flat walk, its own work left out of the count */
        const work_t keep = W;
        int h = world_hit_list(sc, r, tmin, tmax, rec, rng, 0, 0, tmax);
        W = keep;
        return h;
    }
#endif
    EV(world_hits);
    return world_hit_list(sc, r, tmin, tmax, rec, rng, 0, 0, tmax);
}
static int world_hit_list(const PtSceneDesc* sc, const ray_t* r, float tmin, float tmax,
                          hitrec* rec, pcg_t* rng, int first, int hit_anything, float closest)
{
    hitrec temp;
    memset(&temp, 0, sizeof temp);     /* the reference leaves temp_rec uninitialised */
    (void)tmax;
    for (int i = first; i < sc->n_world; i++) {
        const PtObject* ob = &sc->world[i];
        int h = 0;
        if (ob->type == PT_OBJ_SHAPE) h = shape_hit(sc, ob->index, r, tmin, closest, &temp, rng);
        else if (ob->type == PT_OBJ_MEDIUM) h = medium_hit(sc, ob->index, r, tmin, closest, &temp, rng);
        else if (ob->type == PT_OBJ_BVH) {
            const PtBvh* b = &sc->bvhs[ob->index];
            h = xformed_hit(sc, sc->xforms + b->xform_first, b->xform_count, 0, NULL, b->root,
                            r, tmin, closest, &temp, rng);
        }
        if (h) {
            EV(accepted_hits);
            hit_anything = 1;
            closest = temp.t;
            *rec = temp;
        }
    }
    return hit_anything;
}

/* ---- light shapes: Hitable::pdfValue / random ------------------------------ */

static float light_shape_pdf(const PtSceneDesc* sc, int shape, v3 o, v3 v, pcg_t* rng)
{
    const PtShape* s = &sc->shapes[shape];
    ray_t r; r.o = o; r.d = v; r.time = 0;
    hitrec rec;
    memset(&rec, 0, sizeof rec);
    if (s->flags & PT_FLIP_NORMAL) return 0;                                              /* FlipNormals: ptHitable.h:58 */
    if (s->xform_count == 0 && s->type == PT_SHAPE_RECT_XZ) {                             /* ptRectangle.h:77-89 */
        if (rect_hit(1, s->p[0], s->p[1], s->p[2], s->p[3], s->p[4], s->material, &r, 0.001f, FLT_MAX, &rec)) {
            FL(8);
            float area = (s->p[1] - s->p[0]) * (s->p[3] - s->p[2]);
            float dist_sqrd = rec.t * rec.t * vsqlen(v);
            float cosine = fabsf(vdot(v, rec.n) / vlen(v));
            return dist_sqrd / (cosine * area);
        }
        return 0;
    }
    if (s->xform_count == 0 && s->type == PT_SHAPE_SPHERE) {                              /* ptSphere.h:48-58 */
        if (sphere_hit(s->p, s->material, &r, 0.001f, FLT_MAX, &rec)) {
            v3 center = V(s->p[0], s->p[1], s->p[2]);
            float radius = s->p[3];
            FL(8);
            float cos_theta_max = sqrtf(1 - radius * radius / vsqlen(vsub(center, o)));
            float solid_angle = 2 * PI_F * (1 - cos_theta_max);
            return 1 / solid_angle;
        }
        return 0;
    }
    (void)rng;
    return 0;                                                                             /* ptHitable.h:58 */
}
static v3 light_shape_random(const PtSceneDesc* sc, int shape, v3 o, pcg_t* rng)
{
    const PtShape* s = &sc->shapes[shape];
    if (s->flags & PT_FLIP_NORMAL) return V(1, 0, 0);                                     /* FlipNormals: ptHitable.h:59 */
    if (s->xform_count == 0 && s->type == PT_SHAPE_RECT_XZ) {                             /* ptRectangle.h:91-95 */
        /* g++ evaluates the constructor arguments right to left: z is drawn before x */
        float rz = pcg_rand(rng);
        float rx = pcg_rand(rng);
        FL(6);
        v3 rp = V(s->p[0] + rx * (s->p[1] - s->p[0]), s->p[4], s->p[2] + rz * (s->p[3] - s->p[2]));
        return vsub(rp, o);
    }
    if (s->xform_count == 0 && s->type == PT_SHAPE_SPHERE) {                              /* ptSphere.h:60-67 */
        v3 center = V(s->p[0], s->p[1], s->p[2]);
        v3 dir = vsub(center, o);
        float dist_sqrd = vsqlen(dir);
        onb_t uvw;
        onb_from_w(&uvw, dir);
        return onb_local(&uvw, random_to_unit_sphere(s->p[3], dist_sqrd, rng));
    }
    return V(1, 0, 0);                                                                    /* ptHitable.h:59 */
}
/* BVH::pdfValue / BVH::random (src/ptBVH.cu:82-95) */
static float light_tree_pdf(const PtSceneDesc* sc, int child, v3 o, v3 v, pcg_t* rng)
{
    if (child < 0) return light_shape_pdf(sc, ~child, o, v, rng);
    const PtBvhNode* n = &sc->bvh_nodes[child];
    float weight = 0.5f;
    FL(3);
    float sum = weight * light_tree_pdf(sc, n->left, o, v, rng) + weight * light_tree_pdf(sc, n->right, o, v, rng);
    return sum;
}
static v3 light_tree_random(const PtSceneDesc* sc, int child, v3 o, pcg_t* rng)
{
    if (child < 0) return light_shape_random(sc, ~child, o, rng);
    const PtBvhNode* n = &sc->bvh_nodes[child];
    FL(1);
    if (pcg_rand(rng) < 0.5f) return light_tree_random(sc, n->left, o, rng);
    return light_tree_random(sc, n->right, o, rng);
}

static float lights_pdf(const PtSceneDesc* sc, v3 o, v3 v, pcg_t* rng)
{
    if (sc->lights_is_list == 2) {
        const PtBvh* b = &sc->bvhs[sc->lights[0]];
        if (b->xform_count > 0) return 0;                                                 /* ptHitable.h:58 */
        return light_tree_pdf(sc, b->root, o, v, rng);
    }
    if (!sc->lights_is_list) return light_shape_pdf(sc, sc->lights[0], o, v, rng);
    FL(1 + 2 * sc->n_lights);
    float weight = 1 / (float)sc->n_lights;                                               /* ptHitableList.h:29-38 */
    float sum = 0;
    for (int i = 0; i < sc->n_lights; i++) sum += weight * light_shape_pdf(sc, sc->lights[i], o, v, rng);
    return sum;
}
static v3 lights_random(const PtSceneDesc* sc, v3 o, pcg_t* rng)
{
    if (sc->lights_is_list == 2) {
        const PtBvh* b = &sc->bvhs[sc->lights[0]];
        if (b->xform_count > 0) return V(1, 0, 0);                                        /* ptHitable.h:59 */
        return light_tree_random(sc, b->root, o, rng);
    }
    if (!sc->lights_is_list) return light_shape_random(sc, sc->lights[0], o, rng);
    FL(1);
    int index = (int)(pcg_rand(rng) * sc->n_lights);                                      /* ptHitableList.h:40-44 */
    return light_shape_random(sc, sc->lights[index], o, rng);
}

/* ---- materials (include/ptMaterial.h) -------------------------------------- */

static float schlick(float cs, float ri)                                                  /* :31-37 */
{
    FL(8); TR(1);
    float r0 = (1 - ri) / (1 + ri);
    r0 = r0 * r0;
    return r0 + (1 - r0) * powf(1 - cs, 5);
}

typedef struct { ray_t specular_ray; int is_specular; v3 attenuation; int cosine_pdf; } scatter_t;

static v3 material_emitted(const PtSceneDesc* sc, int mat, const ray_t* r_in, const hitrec* rec)
{
    const PtMaterial* m = &sc->materials[mat];
    if (m->type == PT_MAT_DIFFUSE_LIGHT) {                                                /* :266-272 */
        FL(1);
        if (vdot(rec->n, r_in->d) < 0) return texture_value(sc, m->texture, rec->u, rec->v, rec->p);
        return V(0, 0, 0);
    }
    return V(0, 0, 0);                                                                    /* :55 */
}

static int material_scatter(const PtSceneDesc* sc, int mat, const ray_t* r_in, const hitrec* rec,
                            scatter_t* srec, pcg_t* rng)
{
    const PtMaterial* m = &sc->materials[mat];
    switch (m->type) {
    case PT_MAT_LAMBERTIAN:                                                               /* :71-78 */
        srec->is_specular = 0;
        srec->attenuation = texture_value(sc, m->texture, rec->u, rec->v, rec->p);
        srec->cosine_pdf = 1;
        return 1;
    case PT_MAT_METAL: {                                                                  /* :134-143 */
        v3 reflected = vreflect(vunit(r_in->d), rec->n);
        srec->specular_ray.o = rec->p;
        srec->specular_ray.d = vadd(reflected, smul(m->param, random_in_unit_sphere(rng)));
        srec->specular_ray.time = 0;
        srec->attenuation = V(m->albedo[0], m->albedo[1], m->albedo[2]);
        srec->is_specular = 1;
        srec->cosine_pdf = 0;
        return 1;
    }
    case PT_MAT_DIELECTRIC: {                                                             /* :184-224 */
        const float ref_idx = m->param;
        FL(5);
        srec->is_specular = 1;
        srec->cosine_pdf = 0;
        srec->attenuation = V(1, 1, 1);
        v3 outward_normal;
        v3 reflected = vreflect(r_in->d, rec->n);
        v3 refracted = V(0, 0, 0);
        float reflect_prob, cosine, ni_over_nt;
        if (vdot(r_in->d, rec->n) > 0.0f) {
            outward_normal = vneg(rec->n);
            ni_over_nt = ref_idx;
            cosine = ref_idx * vdot(r_in->d, rec->n) / vlen(r_in->d);
        } else {
            outward_normal = rec->n;
            ni_over_nt = 1.0f / ref_idx;
            cosine = -vdot(r_in->d, rec->n) / vlen(r_in->d);
        }
        if (vrefract(r_in->d, outward_normal, ni_over_nt, &refracted)) reflect_prob = schlick(cosine, ref_idx);
        else reflect_prob = 1.0f;
        srec->specular_ray.o = rec->p;
        srec->specular_ray.time = 0;
        if (pcg_rand(rng) < reflect_prob) srec->specular_ray.d = reflected;
        else srec->specular_ray.d = refracted;
        return 1;
    }
    case PT_MAT_DIFFUSE_LIGHT:                                                            /* :262-265 */
        return 0;
    case PT_MAT_ISOTROPIC:                                                                /* :317-325 */
        srec->is_specular = 0;
        srec->attenuation = texture_value(sc, m->texture, rec->u, rec->v, rec->p);
        srec->cosine_pdf = 0;
        return 1;
    }
    return 0;
}

static float material_scattering_pdf(const PtSceneDesc* sc, int mat, const hitrec* rec, const ray_t* scattered)
{
    const PtMaterial* m = &sc->materials[mat];
    if (m->type == PT_MAT_LAMBERTIAN) {                                                   /* :80-85 */
        float cosine = vdot(rec->n, vunit(scattered->d));
        FL(2);
        if (cosine < 0) return 0;
        return cosine / PI_F;
    }
    if (m->type == PT_MAT_ISOTROPIC) return 1.0f / (4.0f * PI_F);                         /* :327-330 */
    return 0;                                                                             /* :54 */
}

/* ---- pdfs (include/ptPDF.h) ------------------------------------------------- */
static float cosine_pdf_value(const onb_t* uvw, v3 dir)                                   /* :46-53 */
{
    float cosine = vdot(vunit(dir), uvw->w);
    FL(2);
    if (cosine > 0) return cosine / PI_F;
    return 0;
}

/* ---- ambient (include/ptAmbientLight.h) ------------------------------------- */
static v3 ambient_emitted(const PtSceneDesc* sc, const ray_t* r)
{
    if (sc->ambient_type == PT_AMBIENT_SKY) {                                             /* :86-91 */
        v3 unit_dir = vunit(r->d);
        FL(3);
        float t = 0.5f * (unit_dir.y + 1);
        return vadd(smul(1 - t, V(1, 1, 1)), smul(t, V(0.5f, 0.7f, 1.0f)));
    }
    return V(sc->ambient_color[0], sc->ambient_color[1], sc->ambient_color[2]);           /* :48-51 */
}

/* ---- camera (include/ptCamera.h:28-34) --------------------------------------- */
static ray_t camera_get_ray(const PtSceneDesc* sc, float s, float t, pcg_t* rng)
{
    const float* c = sc->camera;
    v3 origin = V(c[0], c[1], c[2]), llc = V(c[3], c[4], c[5]), horiz = V(c[6], c[7], c[8]);
    v3 vert = V(c[9], c[10], c[11]), cu = V(c[12], c[13], c[14]), cv = V(c[15], c[16], c[17]);
    float time0 = c[21], time1 = c[22], lens_radius = c[23];
    v3 rd = smul(lens_radius, random_in_unit_disk(rng));
    v3 offset = vadd(vmuls(cu, rd.x), vmuls(cv, rd.y));
    FL(3);
    float time = time0 + pcg_rand(rng) * (time1 - time0);
    ray_t r;
    r.o = vadd(origin, offset);
    r.d = vsub(vsub(vadd(vadd(llc, smul(s, horiz)), smul(t, vert)), origin), offset);
    r.time = time;
    return r;
}

/* ---- integrator (src/ptMain.cu:46-169) ---------------------------------------- */

static v3 de_nan(v3 c)                                                                    /* :46-53 */
{
    FL(3);
    if (!(c.x == c.x)) c.x = 0;
    if (!(c.y == c.y)) c.y = 0;
    if (!(c.z == c.z)) c.z = 0;
    return c;
}

static v3 color(const PtSceneDesc* sc, ray_t current, pcg_t* rng, int max_depth)          /* :95-151 */
{
    v3 accum = V(1, 1, 1);
    for (int depth = 0; depth < max_depth; depth++) {
        hitrec rec;
        memset(&rec, 0, sizeof rec);
        if (world_hit(sc, &current, 0.001f, FLT_MAX, &rec, rng)) {
            scatter_t srec;
            v3 emitted = material_emitted(sc, rec.mat, &current, &rec);
            if (material_scatter(sc, rec.mat, &current, &rec, &srec, rng)) {
                if (srec.is_specular) {
                    accum = vmul(accum, srec.attenuation);
                    current = srec.specular_ray;
                } else {
                    onb_t uvw;
                    EV(diffuse_scatters);
                    onb_from_w(&uvw, rec.n);                                              /* CosinePdf pdf(rec.normal) */
                    ray_t scattered;
                    float pdf_value;
                    if (sc->n_lights > 0) {
                        v3 dir;                                                           /* MixturePdf::generate, ptPDF.h:94-100 */
                        FL(4);
                        if (pcg_rand(rng) < 0.5f) dir = lights_random(sc, rec.p, rng);
                        else dir = onb_local(&uvw, random_cosine_direction(rng));
                        scattered.o = rec.p; scattered.d = dir; scattered.time = current.time;
                        pdf_value = 0.5f * lights_pdf(sc, rec.p, dir, rng) + 0.5f * cosine_pdf_value(&uvw, dir);
                    } else {
                        v3 dir = srec.cosine_pdf ? onb_local(&uvw, random_cosine_direction(rng))
                                                 : random_to_unit_sphere(1, 1, rng);      /* ConstPdf, ptPDF.h:35-38 */
                        scattered.o = rec.p; scattered.d = dir; scattered.time = current.time;
                        pdf_value = srec.cosine_pdf ? cosine_pdf_value(&uvw, dir) : 1;
                    }
                    float spdf = material_scattering_pdf(sc, rec.mat, &rec, &scattered);
                    accum = vmul(accum, vadd(emitted, vdivs(vmuls(srec.attenuation, spdf), pdf_value)));
                    current = scattered;
                }
            } else {
                accum = vmul(accum, emitted);
                break;
            }
        } else {
            if (sc->ambient_type != PT_AMBIENT_NONE) accum = vmul(accum, ambient_emitted(sc, &current));
            break;
        }
    }
    return accum;
}

static v3 render_pixel(const PtSceneDesc* sc, int x, int y, int nx, int ny, int ns, pcg_t* rng, int max_depth)
{                                                                                         /* :153-169 */
    v3 accum = V(0, 0, 0);
    for (int s = 0; s < ns; s++) {
        FL(4); EV(paths);
        float u = (x + pcg_rand(rng)) / (float)nx;
        float v = (y + pcg_rand(rng)) / (float)ny;
        ray_t r = camera_get_ray(sc, u, v, rng);
        accum = vadd(accum, de_nan(color(sc, r, rng, max_depth)));
    }
    accum = vdiv_eq(accum, (float)ns);
    FL(6);
    accum.x = sqrtf(fmaxf(0.0f, accum.x));
    accum.y = sqrtf(fmaxf(0.0f, accum.y));
    accum.z = sqrtf(fmaxf(0.0f, accum.z));
    return accum;
}

/* ---- exported checker entry points ------------------------------------------- */

static int owns_pixel(const PtRenderParams* p, int x, int ytop)
{
    int tw = p->tile_w > 0 ? p->tile_w : 8, th = p->tile_h > 0 ? p->tile_h : 8;
    int sc = p->shard_count > 0 ? p->shard_count : 1;
    int tiles_x = (p->nx + tw - 1) / tw;
    int tile = (ytop / th) * tiles_x + (x / tw);
    return (tile % sc) == p->shard_index;
}

/* render_kernel's body (src/ptMain.cu:171-190) driven by a host loop over rows. */
int oracle_render(const PtSceneDesc* sc, const PtRenderParams* p, float* out_rgb, int n_threads)
{
    const int nx = p->nx, ny = p->ny;
    (void)n_threads;
#ifdef PT_ORACLE_COUNT
    memset(&W_total, 0, sizeof W_total);
#endif
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#endif
    {
#ifdef PT_ORACLE_COUNT
        memset(&W, 0, sizeof W);
#endif
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int y = 0; y < ny; y++) {
            for (int x = 0; x < nx; x++) {
                unsigned i = (unsigned)((ny - y - 1) * nx + x);
                if (!owns_pixel(p, x, ny - y - 1)) {
                    out_rgb[3 * i] = out_rgb[3 * i + 1] = out_rgb[3 * i + 2] = 0.0f;
                    continue;
                }
                pcg_t rng;
                pcg_seed(&rng, i);
                v3 c = render_pixel(sc, x, y, nx, ny, p->ns, &rng, p->max_depth);
                out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
            }
        }
#ifdef PT_ORACLE_COUNT
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            const uint64_t* src = (const uint64_t*)&W;
            uint64_t* dst = (uint64_t*)&W_total;
            for (int k = 0; k < WORK_FIELDS; k++) dst[k] += src[k];
        }
#endif
    }
    return 0;
}

#ifdef PT_ORACLE_COUNT
/* Counters of the most recent oracle_render, in the field order of work_t. */
int oracle_work_counters(uint64_t* out, int n)
{
    const uint64_t* src = (const uint64_t*)&W_total;
    for (int k = 0; k < n && k < WORK_FIELDS; k++) out[k] = src[k];
    return WORK_FIELDS;
}
#endif

int oracle_trace_closest(const PtSceneDesc* sc, const PtRay* rays, size_t n, int32_t* prim_id, float* t)
{
    /* rays are independent (each seeds its own generator): threads only shorten the wait on large worlds */
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; i++) {
        ray_t r;
        r.o = V(rays[i].o[0], rays[i].o[1], rays[i].o[2]);
        r.d = V(rays[i].d[0], rays[i].d[1], rays[i].d[2]);
        r.time = rays[i].time;
        pcg_t rng;
        pcg_seed(&rng, rays[i].seed);
        hitrec rec;
        memset(&rec, 0, sizeof rec);
        if (world_hit(sc, &r, 0.001f, FLT_MAX, &rec, &rng)) { prim_id[i] = rec.prim; t[i] = rec.t; }
        else { prim_id[i] = PT_ID_MISS; t[i] = FLT_MAX; }
    }
    return 0;
}

/* Full hit record for unit checks: out = t, p xyz, n xyz, u, v (9 floats per ray). */
int oracle_trace_records(const PtSceneDesc* sc, const PtRay* rays, size_t n, int32_t* prim_id, float* out9)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; i++) {
        ray_t r;
        r.o = V(rays[i].o[0], rays[i].o[1], rays[i].o[2]);
        r.d = V(rays[i].d[0], rays[i].d[1], rays[i].d[2]);
        r.time = rays[i].time;
        pcg_t rng;
        pcg_seed(&rng, rays[i].seed);
        hitrec rec;
        memset(&rec, 0, sizeof rec);
        float* o = out9 + 9 * i;
        if (world_hit(sc, &r, 0.001f, FLT_MAX, &rec, &rng)) {
            prim_id[i] = rec.prim;
            o[0] = rec.t; o[1] = rec.p.x; o[2] = rec.p.y; o[3] = rec.p.z;
            o[4] = rec.n.x; o[5] = rec.n.y; o[6] = rec.n.z; o[7] = rec.u; o[8] = rec.v;
        } else {
            prim_id[i] = PT_ID_MISS;
            for (int k = 0; k < 9; k++) o[k] = 0;
        }
    }
    return 0;
}

/* Camera rays of the first `ns` samples of every pixel (or of pixel `only_pixel` alone when it is
 * >= 0) plus the chain of secondary rays the integrator follows from them, the pixel's random
 * stream running on from sample to sample as in render_pixel: the ray set used for closest-hit
 * parity checks.  Returns the number of rays written (<= capacity). */
size_t oracle_record_rays_ns(const PtSceneDesc* sc, int nx, int ny, int ns, int max_depth, int only_pixel, PtRay* out, size_t capacity)
{
    size_t n = 0;
    for (int y = 0; y < ny && n < capacity; y++) {
        for (int x = 0; x < nx && n < capacity; x++) {
            unsigned i = (unsigned)((ny - y - 1) * nx + x);
            if (only_pixel >= 0 && i != (unsigned)only_pixel) continue;
            pcg_t rng;
            pcg_seed(&rng, i);
            for (int s = 0; s < ns && n < capacity; s++) {
                float u = (x + pcg_rand(&rng)) / (float)nx;
                float v = (y + pcg_rand(&rng)) / (float)ny;
                ray_t cur = camera_get_ray(sc, u, v, &rng);
                for (int depth = 0; depth < max_depth && n < capacity; depth++) {
                    PtRay* o = &out[n];
                    o->o[0] = cur.o.x; o->o[1] = cur.o.y; o->o[2] = cur.o.z;
                    o->d[0] = cur.d.x; o->d[1] = cur.d.y; o->d[2] = cur.d.z;
                    o->time = cur.time; o->seed = (uint32_t)n;
                    n++;
                    /* advance exactly as color() would, to get realistic secondary rays */
                    hitrec rec;
                    memset(&rec, 0, sizeof rec);
                    if (!world_hit(sc, &cur, 0.001f, FLT_MAX, &rec, &rng)) break;
                    scatter_t srec;
                    if (!material_scatter(sc, rec.mat, &cur, &rec, &srec, &rng)) break;
                    if (srec.is_specular) { cur = srec.specular_ray; continue; }
                    onb_t uvw;
                    onb_from_w(&uvw, rec.n);
                    v3 dir;
                    if (sc->n_lights > 0) {
                        if (pcg_rand(&rng) < 0.5f) dir = lights_random(sc, rec.p, &rng);
                        else dir = onb_local(&uvw, random_cosine_direction(&rng));
                    } else {
                        dir = srec.cosine_pdf ? onb_local(&uvw, random_cosine_direction(&rng))
                                              : random_to_unit_sphere(1, 1, &rng);
                    }
                    cur.o = rec.p; cur.d = dir;
                }
            }
        }
    }
    return n;
}
/* The first camera ray of each listed pixel (pixel = render_kernel's index i, src/ptMain.cu:178), as
 * render_pixel generates it (src/ptMain.cu:158-160); no world is searched, so this is cheap on
 * worlds of any size. */
void oracle_camera_rays(const PtSceneDesc* sc, int nx, int ny, const uint32_t* pixels, size_t n, PtRay* out)
{
    for (size_t k = 0; k < n; k++) {
        const unsigned i = pixels[k];
        const int ytop = (int)(i / (unsigned)nx), x = (int)(i % (unsigned)nx), y = ny - 1 - ytop;
        pcg_t rng;
        pcg_seed(&rng, i);
        float u = (x + pcg_rand(&rng)) / (float)nx;
        float v = (y + pcg_rand(&rng)) / (float)ny;
        ray_t r = camera_get_ray(sc, u, v, &rng);
        out[k].o[0] = r.o.x; out[k].o[1] = r.o.y; out[k].o[2] = r.o.z;
        out[k].d[0] = r.d.x; out[k].d[1] = r.d.y; out[k].d[2] = r.d.z;
        out[k].time = r.time; out[k].seed = (uint32_t)k;
    }
}
size_t oracle_record_rays(const PtSceneDesc* sc, int nx, int ny, int max_depth, PtRay* out, size_t capacity)
{
    return oracle_record_rays_ns(sc, nx, ny, 1, max_depth, -1, out, capacity);
}

/* Known-answer helpers. */
void oracle_pcg_stream(uint64_t seed, int n, float* out)
{
    pcg_t g;
    pcg_seed(&g, seed);
    for (int i = 0; i < n; i++) out[i] = pcg_rand(&g);
}
void oracle_simple_rng_stream(unsigned s0, unsigned s1, int n, float* out)
{
    for (int i = 0; i < n; i++) out[i] = simple_rng_rand(&s0, &s1);
}
void oracle_noise(const float* pts, int n, float* out_noise, float* out_turb)
{
    for (int i = 0; i < n; i++) {
        v3 p = V(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        out_noise[i] = noise3(p);
        out_turb[i] = turbulence(p, 7);
    }
}
void oracle_texture_values(const PtSceneDesc* sc, int tex, const float* uvp, int n, float* out_rgb)
{
    for (int i = 0; i < n; i++) {
        v3 c = texture_value(sc, tex, uvp[5 * i], uvp[5 * i + 1], V(uvp[5 * i + 2], uvp[5 * i + 3], uvp[5 * i + 4]));
        out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
    }
}
/* AABB::hit on n (box, ray) pairs: boxes = 6 floats each. */
void oracle_aabb_hits(const float* boxes, const PtRay* rays, int n, float tmin, float tmax, int32_t* out)
{
    for (int i = 0; i < n; i++) {
        ray_t r;
        r.o = V(rays[i].o[0], rays[i].o[1], rays[i].o[2]);
        r.d = V(rays[i].d[0], rays[i].d[1], rays[i].d[2]);
        r.time = rays[i].time;
        out[i] = aabb_hit(boxes + 6 * i, boxes + 6 * i + 3, &r, tmin, tmax);
    }
}
/* 64-bit FNV-1a over the bytes of a float image. */
/* medianFilter3x3 / medianFilter2x2 (src/ptMain.cu:700-788): the neighbourhood is sorted with the C
 * library's qsort under the reference's comparator, exactly as the reference does it. */
static int median_compare(const void* a, const void* b)                                   /* :700-717 */
{
    const v3 v0 = *(const v3*)a, v1 = *(const v3*)b;
    const float k0 = v0.x * v0.x + v0.y * v0.y + v0.z * v0.z;
    const float k1 = v1.x * v1.x + v1.y * v1.y + v1.z * v1.z;
    if (k0 < k1) return -1;
    if (k0 > k1) return 1;
    return 0;
}
void oracle_median_filter(const float* in, float* out, int nx, int ny, int size)
{
    const v3* src = (const v3*)in;
    v3* dst = (v3*)out;
    for (int l = 0; l < ny; l++) {
        const int rows[3] = { l > 0 ? l - 1 : 0, l, l < ny - 1 ? l + 1 : ny - 1 };
        for (int s = 0; s < nx; s++) {
            const int cols[3] = { s > 0 ? s - 1 : 0, s, s < nx - 1 ? s + 1 : nx - 1 };
            v3 hood[9];
            int n = 0;
            for (int c = 0; c < size; c++)                  /* hood[0..] = (s0,l0), (s0,l1), (s0,l2), (s1,l0), ...  :737-747 */
                for (int q = 0; q < size; q++) hood[n++] = src[rows[q] * nx + cols[c]];
            qsort(hood, (size_t)n, sizeof(v3), median_compare);
            dst[l * nx + s] = hood[(n - 1) / 2];                                          /* :719-723 */
        }
    }
}

uint64_t oracle_fnv1a(const void* data, size_t n_bytes)
{
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n_bytes; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

/* The host libm the reference's Sin/Cos/Log/Pow wrappers resolve to (include/ptMath.h:34-45):
 * fn 0 sinf, 1 cosf, 2 logf, 3 powf(x, 5); and get_uv's calls (include/ptSphere.h:22-28):
 * fn 6 asinf, fn 7 atan2f on n (y, x) pairs (x then holds 2 * n floats). */
void oracle_libm(int fn, const float* x, size_t n, float* out)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = sinf(x[i]); break;
        case 1: out[i] = cosf(x[i]); break;
        case 2: out[i] = logf(x[i]); break;
        case 6: out[i] = asinf(x[i]); break;
        case 7: out[i] = atan2f(x[2 * i], x[2 * i + 1]); break;
        default: out[i] = powf(x[i], 5); break;
        }
    }
}
