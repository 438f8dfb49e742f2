// Device-side radiance stages for gfx950: camera, closest hit over tagged SoA geometry,
// hit completion, materials, light/cosine mixture sampling, textures, noise, media.
//
// The reference runs these as virtual calls on device-`new`ed objects
// (include/ptHitable.h:51-65, include/ptMaterial.h:48-61); here they are switch-free where
// the scene allows it: every stage is a template over a feature mask, and a scene only pays
// for the primitive / material / texture kinds it contains.
//
// Arithmetic contract: every float expression is evaluated in the reference's order with
// one rounding per operation (build with -ffp-contract=off; HIP's default correctly-rounded
// fp32 divide and sqrt).  Where the reference's C++ leaves an order unspecified the order
// g++ picks is used (z before x in XZRectangle::random).  sin, cos, log and pow(.,5) are the
// host libm's results bit for bit (pt_libm.h), and so are atan2f/asinf (MovingSphere uv, read only
// by image textures).
#ifndef PT_DEVICE_H
#define PT_DEVICE_H

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "../../include/ptcore.h"
#include "pt_libm.h"

namespace ptd {

#define PT_FN_EARLY __host__ __device__ __forceinline__
// ---- feature mask ---------------------------------------------------------------------
enum : uint32_t {
    F_SPHERE     = 1u << 0,
    F_MSPHERE    = 1u << 1,
    F_RECT       = 1u << 2,
    F_BOX        = 1u << 3,
    F_TRIANGLE   = 1u << 4,
    F_XFORM      = 1u << 5,
    F_MEDIUM     = 1u << 6,
    F_REFBVH     = 1u << 7,
    F_LBVH       = 1u << 8,
    F_LIGHTS     = 1u << 9,    // a light shape is present (mixture sampling)
    F_LIGHT_MISC = 1u << 10,   // light shapes other than one bare XZ rectangle
    F_METAL      = 1u << 11,
    F_DIELECTRIC = 1u << 12,
    F_ISOTROPIC  = 1u << 13,
    F_CHECKER    = 1u << 14,
    F_NOISE      = 1u << 15,
    F_IMAGE      = 1u << 16,
    F_SKY        = 1u << 17,
    F_ALL        = (1u << 18) - 1,
    F_LBVH4      = 1u << 18    // with F_LBVH: the walk reads the tree's 4-wide form (not part of F_ALL: a way to walk, not a scene feature)
};

constexpr float PI_F = 3.141592654f;   // CUDART_PI_F

// Section timers of a kernel experiment (-DPT_PROFILE on one translation unit, tools/section_profile.py): every wave
// adds up, per section of its loop, the clock ticks spent, the lanes that had work there and the times it ran, and
// adds its sums to a device array when it ends.  Without the flag the type is empty and the calls vanish.
#ifdef PT_PROFILE
constexpr int PROF_SECTIONS = 12;
struct Prof {
    unsigned long long t[PROF_SECTIONS], l[PROF_SECTIONS], n[PROF_SECTIONS], last;
    __device__ __forceinline__ void begin() { for (int k = 0; k < PROF_SECTIONS; k++) { t[k] = 0; l[k] = 0; n[k] = 0; } last = clock64(); }
    // the time since the last mark belongs to section k; `lanes`: ballot of the lanes that had work in it
    __device__ __forceinline__ void mark(int k, unsigned long long lanes)
    {
        const unsigned long long now = clock64();
        if (lanes != 0ull) { t[k] += now - last; l[k] += (unsigned long long)__popcll(lanes); n[k] += 1; }
        else t[PROF_SECTIONS - 1] += now - last;      // nobody needed it: the loop's own overhead
        last = clock64();
    }
};
#else
struct Prof {
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ void mark(int, unsigned long long) {}
};
#endif

// 64-byte LBVH node: both children's boxes plus two links (>= 0 inner node, < 0 ~shape).
struct LbvhNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad0, pad1;
};

// The same tree with every second level folded away: up to four children per node in 64 bytes.  The
// children's boxes are held on a grid of 256 steps per axis laid over the box that holds them all: its
// corner, a power-of-two step per axis, and one byte per plane (low planes rounded down, high planes up, so
// plane q of an axis -- corner + q * step, in exact arithmetic -- never lies inside the fitted box: the walk
// may visit a child it need not have, never the reverse).  One 64-byte fetch decides four descents: half the
// bytes and half the dependent fetches of two binary nodes.  Dword layout: 0-2 corner, 3 the three biased
// step exponents (byte a = exponent of axis a), 4-6 low planes of x, y, z (byte c = child c), 7-9 high
// planes, 10-13 links, 14-15 unused.
constexpr int32_t LBVH4_EMPTY = 0x7fffffff;
struct LbvhNode4 {
    float   corner[3];
    uint8_t step_exp[4];   // [axis]: the step is 2^(step_exp - 127); [3] unused
    uint8_t lo[3][4];      // [axis][child]
    uint8_t hi[3][4];
    int32_t link[4];       // >= 0 node, < 0 ~shape, LBVH4_EMPTY: no child
    int32_t pad[2];
};
static_assert(sizeof(LbvhNode4) == 64, "one 64-byte fetch per node");
// plane q of the grid along one axis, decoded (the walk of the rays that barely move along some axis)
PT_FN_EARLY float lbvh4_plane(float corner, float step, unsigned q) { return corner + (float)q * step; }

// The scene's arrays are read-only for as long as a kernel runs, and the kernels read them through the constant
// address space, which says exactly that.  It matters: a load whose address is the same in every lane goes through the
// scalar unit only if the compiler can show that nothing in the kernel may have written the memory, and an atomic load
// or store, s_sleep, s_setprio, a clock read or inline assembly anywhere in the kernel makes it give up -- the
// persistent kernels, which have all of those, then fetched every list entry and every table row through the vector
// unit, once per lane (a list walk a third slower, 290 more vector instructions per bounce of the rectangle scene's
// chunk kernel).  Loads through a constant-address-space pointer need no such proof.
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_SCENE __attribute__((address_space(4)))
#else
#define PT_SCENE                                           // the host pass only parses the device functions
#endif
template <class T> struct ScenePtr {
    const T* p;                                            // a device pointer; the host fills it
    __device__ __forceinline__ const PT_SCENE T* get() const { return (const PT_SCENE T*)p; }
    __device__ __forceinline__ const PT_SCENE T& operator[](size_t i) const { return ((const PT_SCENE T*)p)[i]; }
    __host__ __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
    __host__ __device__ __forceinline__ bool operator==(std::nullptr_t) const { return p == nullptr; }
    __host__ __device__ __forceinline__ bool operator!=(std::nullptr_t) const { return p != nullptr; }
    template <class I> __device__ __forceinline__ const PT_SCENE T* operator+(I i) const { return get() + i; }
    __host__ ScenePtr& operator=(const T* q) { p = q; return *this; }
};
// The same interface over an ordinary pointer, for the signature walk's tables: there the compiler's own choice is
// the better one (scalar loads in the tile kernels, vector loads in the chunk kernels, which are short of scalar
// registers; rectangle scene 1024^2 x 4096 in chunks: 205.6 ms so, 214.2 with the tables in the constant address
// space, half of the frame 128.0 against 135.2).
template <class T> struct PlainPtr {
    const T* p;
    __device__ __forceinline__ const T* get() const { return p; }
    __device__ __forceinline__ const T& operator[](size_t i) const { return p[i]; }
    __host__ __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
    __host__ __device__ __forceinline__ bool operator==(std::nullptr_t) const { return p == nullptr; }
    __host__ __device__ __forceinline__ bool operator!=(std::nullptr_t) const { return p != nullptr; }
    template <class I> __device__ __forceinline__ const T* operator+(I i) const { return p + i; }
    __host__ PlainPtr& operator=(const T* q) { p = q; return *this; }
};
// A whole record out of a scene array (C++ will not copy a struct out of another address space by itself).
template <class T> __device__ __forceinline__ T scene_load(const PT_SCENE T* p)
{
    static_assert(sizeof(T) % 4 == 0, "records are whole dwords");
    union { T v; uint32_t w[sizeof(T) / 4]; } u;
    const PT_SCENE uint32_t* s = (const PT_SCENE uint32_t*)p;
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; k++) u.w[k] = s[k];
    return u.v;
}

struct SceneView {
    ScenePtr<PtShape>    shapes;
    ScenePtr<PtTriUV>    tri_uvs;
    ScenePtr<PtXform>    xforms;
    ScenePtr<PtMaterial> materials;
    ScenePtr<PtTexture>  textures;
    ScenePtr<uint8_t>    texels;
    ScenePtr<PtMedium>   media;
    ScenePtr<PtBvhNode>  bvh_nodes;
    ScenePtr<PtBvh>      bvhs;
    ScenePtr<PtObject>   world;
    ScenePtr<int32_t>    lights;
    ScenePtr<LbvhNode>   lbvh;
    ScenePtr<LbvhNode4>  lbvh4;    // the 4-wide form of the same tree (kernels compiled with F_LBVH4 walk this one), or nullptr
    PlainPtr<float>      sig;      // signature walk: SIG_SLOT floats per world entry (see world_entry)
    PlainPtr<float>      sig_faces;    // signature walk: SIG_FACE floats per (entry, face): the finished surface (see shade_sig); nullptr = none
    float             sig_light[5];  // signature walk: x0, x1, z0, z1, y of the one XZ rectangle that is the light (shade_sig)
    ScenePtr<int32_t>    lbvh_order;   // per shape: its place in the sequence of tests the in-order walk makes (nullptr: the shape index)
    ScenePtr<int32_t>    lbvh_group;   // per shape: the reference BVH it is a leaf of, -1 for a list entry (with lbvh_order)
    ScenePtr<float>      ref_leaf_gate;    // per shape: box of the reference BVH node that names it as a leaf (6 floats; with lbvh_order)
    int32_t lbvh_prefix;        // world entries the LBVH stands for; the list walk continues from there
    int32_t reads_uv;           // the scene has an image texture (the only reader of a hit's uv)
    ScenePtr<float>      tree_planes;  // plane coordinates of the un-wrapped rectangles and box faces in the LBVH, x planes, then y, then z
    int32_t tree_plane_end[3];  // end of each axis' run in tree_planes
    ScenePtr<int32_t>    tree_plane_first;    // per plane (+1): start of its run in tree_plane_members
    ScenePtr<int32_t>    tree_plane_members;  // the shapes that have a rectangle in that plane
    const SceneView*  self;                // this view in device memory, for code kept out of line
    int32_t tree_wrapped;       // a wrapped rectangle or box is among the LBVH's leaves
    int32_t n_world, n_lights, lights_is_list, ambient_type;
    int32_t lbvh_root;          // >= 0 inner node, < 0 ~shape, when an LBVH is in use
    int32_t lbvh_cull;          // 1: prune by distance (no triangles in the tree)
    int32_t leaf_vote;          // lanes of a wave that gather at leaves before the wave tests leaves (lbvh_voted_step)
    float   ambient[3];
    float   camera[24];
};

// Vector helpers compile for the host as well: the signature tables below are filled on the host with the
// very functions the kernels would run (IEEE add/mul/div/sqrt, no contraction: the same bits on both sides).
#define PT_FN __host__ __device__ __forceinline__
// ---- small vector type ------------------------------------------------------------------
struct V3 { float x, y, z; };
PT_FN V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_FN V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_FN V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_FN V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_FN V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
PT_FN V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
PT_FN V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
PT_FN V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
PT_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_FN float sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PT_FN float len(V3 a) { return sqrtf(sqlen(a)); }
PT_FN V3 cross(V3 a, V3 b)
{
    return mk(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
PT_FN V3 unit(V3 a) { return a / len(a); }

struct Ray { V3 o, d; float time; };
__device__ __forceinline__ V3 at(const Ray& r, float t) { return r.o + t * r.d; }

// ---- PCG-XSH-RR 64/32 (include/ptRNG.h:96-134) ----------------------------------------------
struct Pcg {
    uint64_t state, inc;
    __device__ __forceinline__ uint32_t next_u32()
    {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    __device__ __forceinline__ void seed(uint64_t s)
    {
        state = 0u;
        inc = (s << 1u) | 1u;
        next_u32();
        state += 0x853c49e6748fea9bULL;
        next_u32();
    }
    __device__ __forceinline__ float rand()
    {
        return fminf(0.99999994f, (float)next_u32() * 2.3283064365386963e-10f);
    }
};

// ---- sampling helpers (include/ptRNG.h:137-172) ---------------------------------------------
__device__ __forceinline__ V3 random_in_unit_sphere(Pcg& g)
{
    const float phi = g.rand() * 2 * PI_F;
    const float z = 1 - 2 * g.rand();
    const float r = sqrtf(fmaxf(0.0f, 1 - z * z));
    float sn, cs;
    ptm::sincosf_glibc<true>(phi, &sn, &cs);        // phi = 2 pi rand() < 6.3
    return mk(r * cs, r * sn, z);
}
__device__ __forceinline__ V3 random_in_unit_disk(Pcg& g)
{
    const float r = sqrtf(g.rand());
    const float theta = g.rand() * 2 * PI_F;
    float sn, cs;
    ptm::sincosf_glibc<true>(theta, &sn, &cs);      // theta = 2 pi rand() < 6.3
    return mk(r * cs, r * sn, 0.0f);
}
// r1, r2: results of Pcg::rand(), i.e. multiples of 2^-32 in [0, 1): both square roots take an
// argument that is 0 or at least 2^-32 (sqrt_tame; the functions are declared further down).
__device__ __forceinline__ float sqrt_tame(float x);
__device__ __forceinline__ V3 random_cosine_direction(float r1, float r2)
{
    float z = sqrt_tame(1 - r2);
    float phi = 2 * PI_F * r1;
    float sn, cs;
    ptm::sincosf_glibc<true>(phi, &sn, &cs);        // phi = 2 pi rand() < 6.3
    const float root = sqrt_tame(r2);
    float x = cs * 2 * root;
    float y = sn * 2 * root;
    return mk(x, y, z);
}
__device__ __forceinline__ V3 random_cosine_direction(Pcg& g)
{
    float r1 = g.rand();
    float r2 = g.rand();
    return random_cosine_direction(r1, r2);
}
__device__ __forceinline__ V3 random_to_unit_sphere(float radius, float dist_sqrd, Pcg& g)
{
    float r1 = g.rand();
    float r2 = g.rand();
    float z = 1 + r2 * (sqrtf(1 - radius * radius / dist_sqrd) - 1);
    float phi = 2 * PI_F * r1;
    float sn, cs;
    ptm::sincosf_glibc<true>(phi, &sn, &cs);        // phi = 2 pi rand() < 6.3
    float x = cs * sqrtf(1 - z * z);
    float y = sn * sqrtf(1 - z * z);
    return mk(x, y, z);
}

// ---- orthonormal basis (include/ptONB.h:27-39) ------------------------------------------------
struct Onb { V3 u, v, w; };
PT_FN Onb onb_from_w(V3 n)
{
    Onb b;
    b.w = unit(n);
    V3 a = (fabsf(b.w.x) > 0.9f) ? mk(0, 1, 0) : mk(1, 0, 0);
    b.v = unit(cross(b.w, a));
    b.u = cross(b.w, b.v);
    return b;
}
PT_FN V3 onb_local(const Onb& b, V3 a) { return a.x * b.u + a.y * b.v + a.z * b.w; }

// ---- camera (include/ptCamera.h:28-34) ---------------------------------------------------------
__device__ __forceinline__ Ray camera_ray(const SceneView& sv, float s, float t, Pcg& g)
{
    const float* c = sv.camera;
    const V3 origin = mk(c[0], c[1], c[2]), llc = mk(c[3], c[4], c[5]);
    const V3 horiz = mk(c[6], c[7], c[8]), vert = mk(c[9], c[10], c[11]);
    const V3 cu = mk(c[12], c[13], c[14]), cv = mk(c[15], c[16], c[17]);
    Ray r;
    if (c[23] == 0.0f && c[0] != 0.0f && c[1] != 0.0f && c[2] != 0.0f) {
        // Pinhole: the lens offset is 0 * (r cos, r sin) = (+-0, +-0).  Adding a signed zero to a
        // non-zero origin, or subtracting it from a direction that is not -0 (a difference of
        // finite terms never is), changes no bit, so the disk sample's sin/cos are not needed.
        // Its two random numbers are still drawn.
        (void)g.rand();
        (void)g.rand();
        float time = c[21] + g.rand() * (c[22] - c[21]);
        r.o = origin;
        r.d = llc + s * horiz + t * vert - origin;
        r.time = time;
        return r;
    }
    V3 rd = c[23] * random_in_unit_disk(g);
    V3 offset = cu * rd.x + cv * rd.y;
    float time = c[21] + g.rand() * (c[22] - c[21]);
    r.o = origin + offset;
    r.d = llc + s * horiz + t * vert - origin - offset;
    r.time = time;
    return r;
}

// ---- leaf intersection: parametric distance only -------------------------------------------------
// A candidate is (t, face).  face: BOX -> which of its six rectangles won.

// Sphere::hit (src/ptSphere.cu:11-40): both roots, strict range on both ends.
__device__ __forceinline__ bool sphere_t(V3 center, float radius, const Ray& r, float tmin, float tmax, float& t)
{
    V3 oc = r.o - center;
    float a = dot(r.d, r.d);
    float b = dot(oc, r.d);
    float c = dot(oc, oc) - radius * radius;
    float disc = b * b - a * c;
    if (disc > 0) {
        float sq = sqrtf(disc);
        float temp = (-b - sq) / a;
        if (temp < tmax && temp > tmin) { t = temp; return true; }
        temp = (-b + sq) / a;
        if (temp < tmax && temp > tmin) { t = temp; return true; }
    }
    return false;
}
// MovingSphere::center (include/ptSphere.h:106-109)
template <class P>
__device__ __forceinline__ V3 msphere_center(P q, float time)
{
    V3 c0 = mk(q[0], q[1], q[2]), c1 = mk(q[3], q[4], q[5]);
    return c0 + ((time - q[6]) / (q[7] - q[6])) * (c1 - c0);
}
// ---- division by a denominator in a stated range ------------------------------------------------------
// `n / d` in single precision compiles to the IEEE sequence
//     ds, ns = v_div_scale(d), v_div_scale(n);  r0 = v_rcp(ds);  e = fma(-ds, r0, 1);  r = fma(e, r0, r0);
//     q = ns * r;  e = fma(-ds, q, ns);  q = fma(e, r, q);  e = fma(-ds, q, ns);  q = v_div_fmas(e, r, q);
//     v_div_fixup(q, d, n)
// whose first and last three instructions only act outside a range of operands: with
// 2^-30 <= |d| <= 2^60 and 2^-103 <= |n| <= 2^60 both v_div_scale return their operand unchanged and
// clear VCC, v_div_fmas is then a plain fma and v_div_fixup passes q through.  Inside that range the
// eight instructions in between are therefore the whole division, bit for bit, and the two that
// depend on d alone can be shared by every numerator over the same denominator.  A numerator below
// 2^-103 in magnitude (zero included) gives a finite |q| < 2^-70 here and |n / d| < 2^-70 there:
// not the same bits, but callers that discard every quotient below a bound larger than that see
// no difference (the rectangle test below discards t < tmin).
struct Rcp { float nd, r; };                               // -d and the refined reciprocal of d
__device__ __forceinline__ Rcp rcp_refined(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    Rcp c; c.nd = -d; c.r = __builtin_fmaf(e, r0, r0);
    return c;
}
__device__ __forceinline__ float quot(float n, const Rcp& c)
{
    float q = n * c.r;
    float e = __builtin_fmaf(c.nd, q, n);
    q = __builtin_fmaf(e, c.r, q);
    e = __builtin_fmaf(c.nd, q, n);
    return __builtin_fmaf(e, c.r, q);
}
__device__ __forceinline__ float quot(float n, float d) { return n / d; }
// sqrtf for an argument that is +0 or lies in [2^-96, FLT_MAX].  The correctly rounded sequence the
// compiler emits is: scale the argument up if it is below 2^-96, v_sqrt, step the root one ulp down and
// one ulp up wherever the residual asks for it, scale back, and hand zeros and infinity through.  In
// the stated range the scale is 1 and nothing is handed through (+0 leaves the two steps as +0: the
// stepped-down root is a NaN and the stepped-up one gives a residual of 0), so the middle is the whole.
__device__ __forceinline__ float sqrt_tame(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float down = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float r_down = __builtin_fmaf(-down, s, x), r_up = __builtin_fmaf(-up, s, x);
    const float r = (0.0f >= r_down) ? down : s;
    return (0.0f < r_up) ? up : r;
}
constexpr float SQRT_TAME_MIN = 0x1p-96f;
// The range test for `count` denominators at once and for the origin the numerators are formed from.
constexpr float RCP_D_MIN = 0x1p-30f, RCP_BIG = 0x1p58f, RCP_TMIN_MIN = 0x1p-20f, SHADE_DIR_MAX = 0x1p20f;

// X?Rectangle::hit (src/ptRectangle.cu:12-28,66-82,120-136): inclusive range.
// (ok,dk) plane axis, (oa,da) and (ob,db) the two in-plane axes; dk is the direction component
// itself or its refined reciprocal (only where t0 >= RCP_TMIN_MIN, see above).
template <class D>
__device__ __forceinline__ bool rect_t(float ok, const D& dk, float oa, float da, float ob, float db,
                                       float a0, float a1, float b0, float b1, float k,
                                       float t0, float t1, float& t)
{
    float tt = quot(k - ok, dk);
    if (tt < t0 || tt > t1) return false;
    float a = oa + tt * da;
    float b = ob + tt * db;
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    t = tt;
    return true;
}
// The same test without control flow, for the straight-line signature walk: all of it is
// evaluated and the verdict is one boolean (the comparisons are rect_t's, so NaN behaves alike).
// With no branches between the entries of a list the compiler schedules the whole walk as one
// block: the entries' constants are loaded ahead of their use and the tests overlap.
template <class D>
__device__ __forceinline__ bool rect_ok(float ok, const D& dk, float oa, float da, float ob, float db,
                                        float a0, float a1, float b0, float b1, float k,
                                        float t0, float t1, float& tt)
{
    tt = quot(k - ok, dk);
    const float a = oa + tt * da;
    const float b = ob + tt * db;
    const bool out_t = (tt < t0) | (tt > t1);
    const bool out_ab = (a < a0) | (a > a1) | (b < b0) | (b > b1);
    return !(out_t | out_ab);
}
template <class D, class P>
__device__ __forceinline__ void box_sel(P q, const Ray& r, const D& dx, const D& dy, const D& dz,
                                        float t0, float& closest, int& code, int base)
{
    const float x0 = q[0], y0 = q[1], z0 = q[2], x1 = q[3], y1 = q[4], z1 = q[5];
    float tt;
    if (rect_ok(r.o.z, dz, r.o.x, r.d.x, r.o.y, r.d.y, x0, x1, y0, y1, z1, t0, closest, tt)) { closest = tt; code = base + 0; }
    if (rect_ok(r.o.z, dz, r.o.x, r.d.x, r.o.y, r.d.y, x0, x1, y0, y1, z0, t0, closest, tt)) { closest = tt; code = base + 1; }
    if (rect_ok(r.o.y, dy, r.o.x, r.d.x, r.o.z, r.d.z, x0, x1, z0, z1, y1, t0, closest, tt)) { closest = tt; code = base + 2; }
    if (rect_ok(r.o.y, dy, r.o.x, r.d.x, r.o.z, r.d.z, x0, x1, z0, z1, y0, t0, closest, tt)) { closest = tt; code = base + 3; }
    if (rect_ok(r.o.x, dx, r.o.y, r.d.y, r.o.z, r.d.z, y0, y1, z0, z1, x1, t0, closest, tt)) { closest = tt; code = base + 4; }
    if (rect_ok(r.o.x, dx, r.o.y, r.d.y, r.o.z, r.d.z, y0, y1, z0, z1, x0, t0, closest, tt)) { closest = tt; code = base + 5; }
}
// Box = list of six rectangles searched with a shrinking upper bound
// (include/ptRectangle.h:199-217, src/ptHitableList.cu:12-27): a later face wins ties.
// (dx, dy, dz): the ray's direction components as rect_t takes them.
template <class D, class P>
__device__ __forceinline__ bool box_t(P q, const Ray& r, const D& dx, const D& dy, const D& dz,
                                      float t0, float t1, float& t, int& face)
{
    const float x0 = q[0], y0 = q[1], z0 = q[2], x1 = q[3], y1 = q[4], z1 = q[5];
    bool any = false;
    float closest = t1, tt;
    if (rect_t(r.o.z, dz, r.o.x, r.d.x, r.o.y, r.d.y, x0, x1, y0, y1, z1, t0, closest, tt)) { any = true; closest = tt; face = 0; }
    if (rect_t(r.o.z, dz, r.o.x, r.d.x, r.o.y, r.d.y, x0, x1, y0, y1, z0, t0, closest, tt)) { any = true; closest = tt; face = 1; }
    if (rect_t(r.o.y, dy, r.o.x, r.d.x, r.o.z, r.d.z, x0, x1, z0, z1, y1, t0, closest, tt)) { any = true; closest = tt; face = 2; }
    if (rect_t(r.o.y, dy, r.o.x, r.d.x, r.o.z, r.d.z, x0, x1, z0, z1, y0, t0, closest, tt)) { any = true; closest = tt; face = 3; }
    if (rect_t(r.o.x, dx, r.o.y, r.d.y, r.o.z, r.d.z, y0, y1, z0, z1, x1, t0, closest, tt)) { any = true; closest = tt; face = 4; }
    if (rect_t(r.o.x, dx, r.o.y, r.d.y, r.o.z, r.d.z, y0, y1, z0, z1, x0, t0, closest, tt)) { any = true; closest = tt; face = 5; }
    t = closest;
    return any;
}
// A ray whose origin and direction components, and a box whose corners, lie in the range where the shortened
// division is the whole division (rcp_refined) divides that way, with one refined reciprocal per axis for the
// two faces on it: the same quotients bit for bit (numerators too small for that give |t| < 2^-70 either way,
// below any t0 >= 2^-20, and are turned away by both), 36 + 60 issue slots instead of six IEEE divides' 204.
template <class P>
__device__ __forceinline__ bool box_t(P q, const Ray& r, float t0, float t1, float& t, int& face)
{
    const float ax = fabsf(r.d.x), ay = fabsf(r.d.y), az = fabsf(r.d.z);
    const float big = fmaxf(fmaxf(fmaxf(fabsf(q[0]), fabsf(q[1])), fmaxf(fabsf(q[2]), fabsf(q[3]))), fmaxf(fabsf(q[4]), fabsf(q[5])));
    const bool tame = t0 >= RCP_TMIN_MIN && fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z) <= RCP_BIG && big <= RCP_BIG &&
                      fminf(fminf(ax, ay), az) >= RCP_D_MIN && fmaxf(fmaxf(ax, ay), az) <= RCP_BIG;      // false for NaN anywhere
    if (tame) return box_t(q, r, rcp_refined(r.d.x), rcp_refined(r.d.y), rcp_refined(r.d.z), t0, t1, t, face);
    return box_t(q, r, r.d.x, r.d.y, r.d.z, t0, t1, t, face);
}
// Triangle::hit (src/ptTriangle.cu:28-72): Moller-Trumbore with back-face cull; the value
// compared against the range and reported is dot(edge2,qvec) WITHOUT the 1/det factor.
template <class P>
__device__ __forceinline__ bool triangle_t(P q, const Ray& r, float t_min, float t_max, float& t)
{
    V3 v0 = mk(q[0], q[1], q[2]), v1 = mk(q[3], q[4], q[5]), v2 = mk(q[6], q[7], q[8]);
    V3 edge1 = v1 - v0;
    V3 edge2 = v2 - v0;
    V3 pvec = cross(r.d, edge2);
    float det = dot(edge1, pvec);
    if (det < 0.0001f) return false;
    V3 tvec = r.o - v0;
    float u = dot(tvec, pvec);
    if (u < 0 || u > det) return false;
    V3 qvec = cross(tvec, edge1);
    float v = dot(r.d, qvec);
    if (v < 0 || u + v > det) return false;
    float tt = dot(edge2, qvec);
    if (tt < t_min || tt > t_max) return false;
    t = tt;
    return true;
}

// Translate / RotateY applied to the ray, outermost first (include/ptRectangle.h:272-281,368-379).
__device__ __forceinline__ Ray to_local(const SceneView& sv, int first, int count, Ray r)
{
    for (int k = 0; k < count; k++) {
        const PtXform x = sv.xforms[first + k];
        if (x.type == PT_XFORM_TRANSLATE) {
            r.o = r.o - mk(x.v[0], x.v[1], x.v[2]);
        } else {
            const float sinT = x.v[0], cosT = x.v[1];
            float ox = cosT * r.o.x - sinT * r.o.z;
            float oz = sinT * r.o.x + cosT * r.o.z;
            float dx = cosT * r.d.x - sinT * r.d.z;
            float dz = sinT * r.d.x + cosT * r.d.z;
            r.o.x = ox; r.o.z = oz; r.d.x = dx; r.d.z = dz;
        }
    }
    return r;
}
// Undo the wrappers on a hit point and normal, innermost first (ptRectangle.h:277,382-389).
__device__ __forceinline__ void to_world(const SceneView& sv, int first, int count, V3& p, V3& n)
{
    for (int k = count - 1; k >= 0; k--) {
        const PtXform x = sv.xforms[first + k];
        if (x.type == PT_XFORM_TRANSLATE) {
            p = p + mk(x.v[0], x.v[1], x.v[2]);
        } else {
            const float sinT = x.v[0], cosT = x.v[1];
            float px = cosT * p.x + sinT * p.z;
            float pz = -sinT * p.x + cosT * p.z;
            float nx = cosT * n.x + sinT * n.z;
            float nz = -sinT * n.x + cosT * n.z;
            p.x = px; p.z = pz; n.x = nx; n.z = nz;
        }
    }
}

// One leaf shape against a ray: true when shape.hit(ray, tmin, tmax) would be.
template <uint32_t F>
__device__ __forceinline__ bool shape_t(const SceneView& sv, int index, const Ray& ray, float tmin, float tmax,
                                        float& t, int& face)
{
    const PtShape* s = &sv.shapes[index];
    const int type = s->type;
    Ray r = ray;
    if (F & F_XFORM) {
        const int nx = s->xform_count;
        if (nx > 0) r = to_local(sv, s->xform_first, nx, ray);
    }
    face = 0;
    if ((F & F_SPHERE) && type == PT_SHAPE_SPHERE)
        return sphere_t(mk(s->p[0], s->p[1], s->p[2]), s->p[3], r, tmin, tmax, t);
    if ((F & F_MSPHERE) && type == PT_SHAPE_MOVING_SPHERE)
        return sphere_t(msphere_center(s->p, r.time), s->p[8], r, tmin, tmax, t);
    if (F & F_RECT) {
        if (type == PT_SHAPE_RECT_XY)
            return rect_t(r.o.z, r.d.z, r.o.x, r.d.x, r.o.y, r.d.y, s->p[0], s->p[1], s->p[2], s->p[3], s->p[4], tmin, tmax, t);
        if (type == PT_SHAPE_RECT_XZ)
            return rect_t(r.o.y, r.d.y, r.o.x, r.d.x, r.o.z, r.d.z, s->p[0], s->p[1], s->p[2], s->p[3], s->p[4], tmin, tmax, t);
        if (type == PT_SHAPE_RECT_YZ)
            return rect_t(r.o.x, r.d.x, r.o.y, r.d.y, r.o.z, r.d.z, s->p[0], s->p[1], s->p[2], s->p[3], s->p[4], tmin, tmax, t);
    }
    if ((F & F_BOX) && type == PT_SHAPE_BOX) return box_t(s->p, r, tmin, tmax, t, face);
    if ((F & F_TRIANGLE) && type == PT_SHAPE_TRIANGLE) return triangle_t(s->p, r, tmin, tmax, t);
    return false;
}

// shape_t for a leaf of the library's tree, where the index differs from lane to lane: the whole
// 64-byte record is fetched at once and the test dispatches on the copy.  shape_t reads the type,
// waits, and reads the parameters of that type: two memory round trips per leaf.
template <uint32_t F>
__device__ __forceinline__ bool leaf_t(const SceneView& sv, int index, const Ray& ray, float tmin, float tmax,
                                       float& t, int& face)
{
    const uint4* w = reinterpret_cast<const uint4*>(&sv.shapes[index]);      // records are 64 bytes, 64-byte aligned
    const uint4 a = w[0], b = w[1], c = w[2], d = w[3];
    asm volatile("" :: "v"(a.x), "v"(b.x), "v"(c.x), "v"(d.x));              // all four loads land here
    const int type = (int)a.x;
    const float p[10] = { __uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z),
                          __uint_as_float(c.w), __uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w) };
    Ray r = ray;
    if (F & F_XFORM) {
        const int nx = (int)b.x;
        if (nx > 0) r = to_local(sv, (int)a.w, nx, ray);
    }
    face = 0;
    if constexpr ((F & F_SPHERE) != 0 && (F & F_MSPHERE) != 0) {
        // Both sphere kinds through one copy of the test (a wave whose lanes hold both kinds would run
        // two): the moving sphere's centre is computed for every lane and selected, the values each
        // kind tests with are its own.
        if (type == PT_SHAPE_SPHERE || type == PT_SHAPE_MOVING_SPHERE) {
            const bool moving = type == PT_SHAPE_MOVING_SPHERE;
            const V3 cm = msphere_center(p, r.time);
            const V3 c = mk(moving ? cm.x : p[0], moving ? cm.y : p[1], moving ? cm.z : p[2]);
            return sphere_t(c, moving ? p[8] : p[3], r, tmin, tmax, t);
        }
    }
    if ((F & F_SPHERE) && type == PT_SHAPE_SPHERE)
        return sphere_t(mk(p[0], p[1], p[2]), p[3], r, tmin, tmax, t);
    if ((F & F_MSPHERE) && type == PT_SHAPE_MOVING_SPHERE)
        return sphere_t(msphere_center(p, r.time), p[8], r, tmin, tmax, t);
    if (F & F_RECT) {
        if (type == PT_SHAPE_RECT_XY)
            return rect_t(r.o.z, r.d.z, r.o.x, r.d.x, r.o.y, r.d.y, p[0], p[1], p[2], p[3], p[4], tmin, tmax, t);
        if (type == PT_SHAPE_RECT_XZ)
            return rect_t(r.o.y, r.d.y, r.o.x, r.d.x, r.o.z, r.d.z, p[0], p[1], p[2], p[3], p[4], tmin, tmax, t);
        if (type == PT_SHAPE_RECT_YZ)
            return rect_t(r.o.x, r.d.x, r.o.y, r.d.y, r.o.z, r.d.z, p[0], p[1], p[2], p[3], p[4], tmin, tmax, t);
    }
    if ((F & F_BOX) && type == PT_SHAPE_BOX) return box_t(p, r, tmin, tmax, t, face);
    if ((F & F_TRIANGLE) && type == PT_SHAPE_TRIANGLE) return triangle_t(p, r, tmin, tmax, t);
    return false;
}

// ---- closest hit over the world ---------------------------------------------------------------------
struct Hit {
    float t;
    int32_t id;      // shape index, PT_ID_MEDIUM_BASE + medium, or PT_ID_MISS
    int32_t face;
    // The reference's Sphere::hit leaves rec.uv untouched, so a plain sphere reports the uv of
    // the last accepted uv-writing hit earlier in the same list walk (src/ptHitableList.cu:14-24).
    // Only image textures read uv; uv_id/uv_face/uv_t remember that earlier hit.
    int32_t uv_id, uv_face;
    float uv_t;
    int32_t sig_code;  // signature walk only: 8 * world entry + face of the winner (row of SceneView::sig_faces)
};

// AABB::hit (include/ptAABB.h:30-48)
template <class P>
__device__ __forceinline__ bool aabb_hit(P bmin, P bmax, const Ray& r, float tmin, float tmax)
{
    {
        const float invD = 1 / r.d.x;
        float t0 = (bmin[0] - r.o.x) * invD, t1 = (bmax[0] - r.o.x) * invD;
        if (invD < 0) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    {
        const float invD = 1 / r.d.y;
        float t0 = (bmin[1] - r.o.y) * invD, t1 = (bmax[1] - r.o.y) * invD;
        if (invD < 0) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    {
        const float invD = 1 / r.d.z;
        float t0 = (bmin[2] - r.o.z) * invD, t1 = (bmax[2] - r.o.z) * invD;
        if (invD < 0) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    return true;
}

__device__ __forceinline__ bool writes_uv(int shape_type)
{
    return shape_type != PT_SHAPE_SPHERE;
}

// AABB::hit with the three reciprocals taken once per ray: 1 / d is the same value at every node.
template <class P>
__device__ __forceinline__ bool aabb_hit_inv(P bmin, P bmax, V3 o, V3 inv, float tmin, float tmax)
{
    {
        float t0 = (bmin[0] - o.x) * inv.x, t1 = (bmax[0] - o.x) * inv.x;
        if (inv.x < 0) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    {
        float t0 = (bmin[1] - o.y) * inv.y, t1 = (bmax[1] - o.y) * inv.y;
        if (inv.y < 0) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    {
        float t0 = (bmin[2] - o.z) * inv.z, t1 = (bmax[2] - o.z) * inv.z;
        if (inv.z < 0) { float tmp = t0; t0 = t1; t1 = tmp; }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    return true;
}

// BVH::hit (src/ptBVH.cu:45-74) evaluated node by node, for rays on which a leaf reports t = NaN
// (a rectangle or box face accepts the 0/0 of a ray lying in its plane).  The reference keeps the
// left child's record only when `left.t < right.t`; with NaN among the values that rule is not an
// ordering (NaN loses on the left and wins on the right), so the winner depends on how the tree
// pairs the leaves and a left-to-right scan no longer finds it.  This walk holds, per open node,
// the finished left result while the right subtree is evaluated, then applies the rule itself.
// It is the slow, exact path: refbvh_t falls back to it when it meets a NaN.
// Masks whose shape tests can accept a NaN t (see ray_may_enter_tree).
template <uint32_t F> struct accepts_nan { static constexpr bool value = (F & (F_RECT | F_BOX | F_TRIANGLE)) != 0; };
constexpr int REFBVH_DEPTH = 48;
template <uint32_t F>
__device__ bool refbvh_by_node(const SceneView& sv, const Ray& r, const V3& inv, int root, float tmin, float tmax,
                               float& t_out, int& id_out, int& face_out)
{
    int   link[REFBVH_DEPTH];                                      // the open nodes' right children
    float lt[REFBVH_DEPTH];                                        // their finished left results
    int   lid[REFBVH_DEPTH], lface[REFBVH_DEPTH];
    uint64_t right_open = 0, left_hit = 0;                         // bit k: node k is in its right subtree / its left subtree hit
    int sp = 0, cur = root;
    while (true) {
        bool hit = false; float t = 0.0f; int id = -1, face = 0;   // the result of the subtree at cur
        bool finished = true;
        if (cur < 0) {
            hit = shape_t<F>(sv, ~cur, r, tmin, tmax, t, face);
            id = ~cur;
        } else {
            const PtBvhNode* n = &sv.bvh_nodes[cur];
            if (sp < REFBVH_DEPTH && aabb_hit_inv(n->bmin, n->bmax, r.o, inv, tmin, tmax)) {
                link[sp] = n->right;
                right_open &= ~(1ull << sp);
                sp++;
                cur = n->left;
                finished = false;
            }
        }
        while (finished) {
            if (sp == 0) { t_out = t; id_out = id; face_out = face; return hit; }
            const uint64_t bit = 1ull << (sp - 1);
            if (!(right_open & bit)) {                             // that was the node's left subtree: keep it, go right
                lt[sp - 1] = t; lid[sp - 1] = id; lface[sp - 1] = face;
                left_hit = hit ? (left_hit | bit) : (left_hit & ~bit);
                right_open |= bit;
                cur = link[sp - 1];
                finished = false;
            } else {                                               // both children known: src/ptBVH.cu:56-70
                const bool hl = (left_hit & bit) != 0;
                if (hl && (!hit || lt[sp - 1] < t)) { t = lt[sp - 1]; id = lid[sp - 1]; face = lface[sp - 1]; }
                hit = hit || hl;
                sp--;
            }
        }
    }
}

// BVH::hit (src/ptBVH.cu:45-74) without recursion.  The reference visits both children of every
// node whose box the ray meets within the caller's [tmin,tmax] (the range never shrinks inside
// the tree) and keeps the left result only when it is strictly nearer; the winner over a subtree
// is therefore the right-most leaf among those with the smallest t.  A left-to-right walk that
// lets a later leaf replace the best one unless the best is strictly nearer gives the same leaf.
// The walk alternates two phases so that the lanes of a wave test leaves together: descend
// through inner nodes until this lane holds a leaf (cheap box tests), then test it (a Box leaf is
// six rectangle tests).  Pending right children wait on a per-lane stack whose first entries are
// in LDS (`stack`, stride 64) and the rest in private memory.
constexpr int REFBVH_STACK = 48;
constexpr int TREE_LDS_STACK = 24;
template <uint32_t F>
__device__ __forceinline__ bool refbvh_t(const SceneView& sv, int bvh_index, const Ray& ray, float tmin, float tmax,
                                         float& t_out, int& id_out, int& face_out, int* stack)
{
    const PtBvh b = sv.bvhs[bvh_index];
    Ray r = ray;
    if (F & F_XFORM) { if (b.xform_count > 0) r = to_local(sv, b.xform_first, b.xform_count, ray); }
    const V3 inv = mk(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);            // const float invD = 1 / r.direction()[a], include/ptAABB.h:34
    int spill[REFBVH_STACK - TREE_LDS_STACK];
    int sp = 0;
    int cur = b.root;
    bool any = false, done = false, nan_seen = false;
    float best = 0.0f;
    while (true) {
        while (!done && cur >= 0) {                                // inner nodes
            const PtBvhNode* n = &sv.bvh_nodes[cur];
            if (aabb_hit_inv(n->bmin, n->bmax, r.o, inv, tmin, tmax)) {
                const int right = n->right;
                if (sp < TREE_LDS_STACK) stack[sp * 64] = right;
                else if (sp < REFBVH_STACK) spill[sp - TREE_LDS_STACK] = right;
                sp++;
                cur = n->left;
            } else if (sp == 0) {
                done = true;
            } else {
                --sp;
                cur = (sp < TREE_LDS_STACK) ? stack[sp * 64] : spill[sp - TREE_LDS_STACK];
            }
        }
        if (done) break;
        float t; int face;                                         // a leaf
        if (shape_t<F>(sv, ~cur, r, tmin, tmax, t, face)) {
            if (!any || !(best < t)) { any = true; best = t; id_out = ~cur; face_out = face; }
            if (accepts_nan<F>::value && t != t) nan_seen = true;
        }
        if (sp == 0) break;
        --sp;
        cur = (sp < TREE_LDS_STACK) ? stack[sp * 64] : spill[sp - TREE_LDS_STACK];
    }
    if (accepts_nan<F>::value && nan_seen) return refbvh_by_node<F>(sv, r, inv, b.root, tmin, tmax, t_out, id_out, face_out);
    t_out = best;
    return any;
}

// ConstantMedium::hit (include/ptMedium.h:31-55).  Draws one random number whenever the ray
// crosses the boundary twice and the clipped interval is not inverted.
template <uint32_t F>
__device__ __forceinline__ bool medium_t(const SceneView& sv, int m, const Ray& r, float t0, float t1, float& t, Pcg& g)
{
    const PtMedium med = sv.media[m];
    float ta, tb; int face;
    if constexpr ((F & F_SPHERE) != 0) {
        // A plain sphere as the boundary (both volumes of the final scene): the two boundary searches
        // (include/ptMedium.h:35-36) evaluate the same quadratic of the same ray and differ in the range they
        // accept a root in, so the roots are computed once.  Sphere::hit (src/ptSphere.cu:11-40) takes the
        // near root when it lies in the range, else the far one; computing a root it would not have looked at
        // changes no value.
        const PtShape* s = &sv.shapes[med.boundary];
        if (s->type == PT_SHAPE_SPHERE && (!(F & F_XFORM) || s->xform_count == 0)) {
            const V3 oc = r.o - mk(s->p[0], s->p[1], s->p[2]);
            const float radius = s->p[3];
            const float a = dot(r.d, r.d);
            const float b = dot(oc, r.d);
            const float c = dot(oc, oc) - radius * radius;
            const float disc = b * b - a * c;
            if (!(disc > 0)) return false;
            const float sq = sqrtf(disc);
            const float near_root = (-b - sq) / a, far_root = (-b + sq) / a;
            if (near_root < FLT_MAX && near_root > -FLT_MAX) ta = near_root;
            else if (far_root < FLT_MAX && far_root > -FLT_MAX) ta = far_root;
            else return false;
            const float from = (float)((double)ta + 0.0001);
            if (near_root < FLT_MAX && near_root > from) tb = near_root;
            else if (far_root < FLT_MAX && far_root > from) tb = far_root;
            else return false;
            if (ta < t0) ta = t0;
            if (tb > t1) tb = t1;
            if (ta > tb) return false;
            if (ta < 0) ta = 0;
            const float dist_inside = (tb - ta) * len(r.d);
            const float hit_dist = -(1 / med.density) * ptm::logf_glibc(g.rand());
            if (hit_dist < dist_inside) {
                t = ta + hit_dist / len(r.d);
                return true;
            }
            return false;
        }
    }
    if (shape_t<F>(sv, med.boundary, r, -FLT_MAX, FLT_MAX, ta, face)) {
        if (shape_t<F>(sv, med.boundary, r, (float)((double)ta + 0.0001), FLT_MAX, tb, face)) {
            if (ta < t0) ta = t0;
            if (tb > t1) tb = t1;
            if (ta > tb) return false;
            if (ta < 0) ta = 0;
            float dist_inside = (tb - ta) * len(r.d);
            float hit_dist = -(1 / med.density) * ptm::logf_glibc(g.rand());
            if (hit_dist < dist_inside) {
                t = ta + hit_dist / len(r.d);
                return true;
            }
        }
    }
    return false;
}

// Tie rule of an in-order list walk, usable in any visiting order.  Spheres accept
// tmin < t < closest (an earlier entry keeps an exact tie), rectangles, boxes and triangles accept
// t <= closest (a later entry takes it): src/ptSphere.cu:21,30, src/ptRectangle.cu:15,
// src/ptTriangle.cu:66.  Between two candidates with equal t the later one wins iff it is of the
// inclusive kind.
__device__ __forceinline__ bool inclusive_kind(int shape_type)
{
    return shape_type != PT_SHAPE_SPHERE && shape_type != PT_SHAPE_MOVING_SPHERE;
}
__device__ __forceinline__ bool list_order_better(const SceneView& sv, float t, int id, float best_t, int best_id)
{
    if (t < best_t) return true;
    if (t > best_t) return false;
    int oa = id, ob = best_id;
    if (sv.lbvh_order) {
        // Leaves of one reference BVH: the tree keeps the right-most of its nearest leaves whatever
        // their kind (src/ptBVH.cu:54-57).  Such a tree is only taken into the LBVH when all its
        // leaves are of one kind, which keeps this rule and the list's rule below consistent when
        // three candidates tie.
        const int ga = sv.lbvh_group[id];
        oa = sv.lbvh_order[id]; ob = sv.lbvh_order[best_id];
        if (ga >= 0 && ga == sv.lbvh_group[best_id]) return oa > ob;
    }
    if (oa > ob) return inclusive_kind(sv.shapes[id].type);
    return !inclusive_kind(sv.shapes[best_id].type);
}

// Library-built LBVH over the world list: any-order stack walk with the tie rule above.
// The first LBVH_LDS_STACK pending nodes of a lane live in LDS (one 4-byte column per lane, stride
// 64 so a wave's pushes hit 64 different banks); deeper entries spill to private memory.
constexpr int LBVH_LDS_STACK = TREE_LDS_STACK;
constexpr int LBVH_STACK = 64;
template <class P>
__device__ __forceinline__ bool slab(P bmin, P bmax, V3 o, V3 inv, float tmin, float tmax, float& tnear)
{
    float tx0 = (bmin[0] - o.x) * inv.x, tx1 = (bmax[0] - o.x) * inv.x;
    float ty0 = (bmin[1] - o.y) * inv.y, ty1 = (bmax[1] - o.y) * inv.y;
    float tz0 = (bmin[2] - o.z) * inv.z, tz1 = (bmax[2] - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tmax));
    tnear = tn;
    // widened by a few ulps so a leaf the in-order walk would accept is never pruned
    return tn * 0.9999990f <= tf * 1.0000010f;
}
// The walk as a resumable state: begin, then one node or one leaf per step until step() says the
// walk has ended.  A scheduler that shares one step at a time between the lanes of a wave can
// stop stepping when too few lanes are still walking and come back later.
struct LbvhWalk {
    V3 inv;
    bool flat;           // a direction component is zero, denormal or out of [2^-30, 2^30]: the wide walk decodes its planes for this ray
    int cur, sp;
    float best;
    int best_id, best_face;
    bool any;
    bool odd;            // a leaf accepted a hit whose t is NaN: only the in-order walk can say what that leads to
};
// Rectangles and boxes reject with comparisons that are all false for NaN, so they *accept* a ray
// whose t comes out NaN: a ray lying in a rectangle's plane (direction component 0 and origin on
// the plane: 0 / 0) "hits" that rectangle wherever it is, and so does any ray carrying NaN
// (triangles likewise, for NaN input).  From then on the in-order walk's bound is NaN: every
// later inclusive test passes, every sphere fails.  Neither a box test (the rectangle may be nowhere
// near the ray) nor a comparison of candidates two at a time can reproduce that, so such rays do
// not enter the tree; the caller walks the list in order for them.  The test is on the ray: a
// non-finite component or a magnitude beyond 2^40 (with the tree's shapes bounded by 2^40 as well
// -- checked when the tree is built -- no product in the triangle test can then overflow into
// inf - inf), or a direction component that is exactly zero while the origin lies exactly in the
// plane of one of the tree's rectangles or box faces on that axis (their coordinates are listed per
// axis when the tree is built; exact zeros are not rare: a few primary rays per frame on the
// horizon row).  A NaN accepted inside the tree all the same is noted (LbvhWalk::odd) and sends
// the ray the same way.  Spheres never accept NaN.
constexpr float TREE_RAY_LIMIT = 0x1p40f;
__device__ __forceinline__ bool on_a_tree_plane(const SceneView& sv, int axis, float o)
{
    if (sv.tree_wrapped) return true;                          // planes of wrapped shapes are not listed: assume the worst
    for (int k = axis == 0 ? 0 : sv.tree_plane_end[axis - 1]; k < sv.tree_plane_end[axis]; k++)
        if (sv.tree_planes[k] == o) return true;
    return false;
}
__device__ __forceinline__ bool ray_may_enter_tree(const SceneView& sv, const Ray& r)
{
    const float s = fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z) + fabsf(r.d.x) + fabsf(r.d.y) + fabsf(r.d.z);
    if (!(s <= TREE_RAY_LIMIT)) return false;                  // also NaN and infinity
    if (r.d.x != 0.0f && r.d.y != 0.0f && r.d.z != 0.0f) return true;
    if (r.d.x == 0.0f && on_a_tree_plane(sv, 0, r.o.x)) return false;
    if (r.d.y == 0.0f && on_a_tree_plane(sv, 1, r.o.y)) return false;
    if (r.d.z == 0.0f && on_a_tree_plane(sv, 2, r.o.z)) return false;
    return true;
}
__device__ __forceinline__ void lbvh_begin(const SceneView& sv, const Ray& r, float tmax, LbvhWalk& w)
{
    w.inv = mk(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    {
        const float ax = fabsf(r.d.x), ay = fabsf(r.d.y), az = fabsf(r.d.z);
        w.flat = !(fminf(fminf(ax, ay), az) >= 0x1p-30f && fmaxf(fmaxf(ax, ay), az) <= 0x1p30f);     // true for NaN as well
    }
    w.cur = sv.lbvh_root;
    w.sp = 0;
    w.any = false;
    w.odd = false;
    w.best = tmax;
    w.best_id = -1; w.best_face = 0;
}
// one node or one leaf per step: sphere and triangle leaves cost about what a node's two box
// tests cost, and grouping them by phase (as the reference-tree walk does) measured slower here
// One box of a 4-wide node against the ray: slab()'s arithmetic on bounds held in registers.
__device__ __forceinline__ bool slab6(float x0, float y0, float z0, float x1, float y1, float z1, V3 o, V3 inv, float tmin, float tmax, float& tnear)
{
    float tx0 = (x0 - o.x) * inv.x, tx1 = (x1 - o.x) * inv.x;
    float ty0 = (y0 - o.y) * inv.y, ty1 = (y1 - o.y) * inv.y;
    float tz0 = (z0 - o.z) * inv.z, tz1 = (z1 - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tmax));
    tnear = tn;
    return tn * 0.9999990f <= tf * 1.0000010f;
}
__device__ __forceinline__ void lbvh_push(LbvhWalk& w, int link, int* stack, int* spill)
{
    if (w.sp < LBVH_LDS_STACK) stack[w.sp * 64] = link;
    else if (w.sp < LBVH_STACK) spill[w.sp - LBVH_LDS_STACK] = link;
    w.sp++;
}
// The walk's two kinds of step.  Both return true when the walk has ended (nothing left on the stack).
__device__ __forceinline__ bool lbvh_pop(LbvhWalk& w, const int* stack, const int* spill)
{
    if (w.sp == 0) return true;
    --w.sp;
    w.cur = (w.sp < LBVH_LDS_STACK) ? stack[w.sp * 64] : spill[w.sp - LBVH_LDS_STACK];
    return false;
}
template <uint32_t F>
__device__ __forceinline__ bool lbvh_leaf_step(const SceneView& sv, const Ray& r, float tmin, float tmax, LbvhWalk& w, int* stack, int* spill)
{
    const int id = ~w.cur;
    float t; int face;
    if (leaf_t<F>(sv, id, r, tmin, tmax, t, face)) {
        if (accepts_nan<F>::value && t != t) w.odd = true;
        if (!w.any || list_order_better(sv, t, id, w.best, w.best_id)) { w.any = true; w.best = t; w.best_id = id; w.best_face = face; }
    }
    return lbvh_pop(w, stack, spill);
}
template <uint32_t F>
__device__ __forceinline__ bool lbvh_node_step(const SceneView& sv, const Ray& r, float tmin, float tmax, LbvhWalk& w, int* stack, int* spill)
{
    // With triangles in the tree neither end of [tmin, best] bounds the distance: the reference
    // compares the un-normalised dot(edge2, qvec) = distance * det (src/ptTriangle.cu:65-66), so a
    // hit a hair in front of the origin (distance << tmin) passes tmin when det > 1 -- a ray
    // leaving a triangle can hit that same triangle again, and it has left the leaf's box before
    // tmin.  det > 0 there, so the distance is at least positive.
    const float lim = sv.lbvh_cull != 0 ? w.best : tmax;
    const float from = sv.lbvh_cull != 0 ? tmin : 0.0f;
    if constexpr ((F & F_LBVH4) != 0) {
        // One 64-byte node: four child boxes on the node's byte grid and four links.  The nearest child that
        // the ray meets is walked next, the others wait on the stack (in no particular order: the winner of a
        // search does not depend on the visiting order, see list_order_better; only pruning does, a little).
        // The planes are not decoded: along an axis the ray meets plane q at q * a + b with a = step * inv and
        // b = (corner - o) * inv, one multiply-add per plane, and which of a child's two planes comes first is
        // the sign of inv.  Rounding leaves that off by at most 2^-22 (|b| + 255 |a|); the near planes are
        // therefore taken 2^-21 (|b| + 256 |a|) early and the far planes as much late, per axis.  A zero
        // direction component would make a and b infinite or NaN and the axis useless (NaN, which the min/max
        // chains ignore), so such rays take the decoded form below.
        const int4* q = reinterpret_cast<const int4*>(&sv.lbvh4[w.cur]);
        const int4 qa = q[0], qb = q[1], qc = q[2], qd = q[3];
        const int4 lk = make_int4(qc.z, qc.w, qd.x, qd.y);
        float t0, t1, t2, t3;
        bool h0, h1, h2, h3;
#define PT_BYTE(w, c) (((unsigned)(w) >> (8 * (c))) & 0xffu)
        if (!w.flat) {
            const float ax = __int_as_float((qa.w & 0xff) << 23) * w.inv.x, bx = (__int_as_float(qa.x) - r.o.x) * w.inv.x;
            const float ay = __int_as_float(((qa.w >> 8) & 0xff) << 23) * w.inv.y, by = (__int_as_float(qa.y) - r.o.y) * w.inv.y;
            const float az = __int_as_float(((qa.w >> 16) & 0xff) << 23) * w.inv.z, bz = (__int_as_float(qa.z) - r.o.z) * w.inv.z;
            const float ex = fmaf(0x1p-13f, fabsf(ax), 0x1p-21f * fabsf(bx));
            const float ey = fmaf(0x1p-13f, fabsf(ay), 0x1p-21f * fabsf(by));
            const float ez = fmaf(0x1p-13f, fabsf(az), 0x1p-21f * fabsf(bz));
            const float bnx = bx - ex, bfx = bx + ex, bny = by - ey, bfy = by + ey, bnz = bz - ez, bfz = bz + ez;
            const bool mx = w.inv.x < 0.0f, my = w.inv.y < 0.0f, mz = w.inv.z < 0.0f;       // the ray runs down the axis: high planes first
            const unsigned nx = (unsigned)(mx ? qb.w : qb.x), fx = (unsigned)(mx ? qb.x : qb.w);
            const unsigned ny = (unsigned)(my ? qc.x : qb.y), fy = (unsigned)(my ? qb.y : qc.x);
            const unsigned nz = (unsigned)(mz ? qc.y : qb.z), fz = (unsigned)(mz ? qb.z : qc.y);
#define PT_CHILD(c, tn_out, hit_out) \
            { const float tn_ = fmaxf(fmaxf(fmaf((float)PT_BYTE(nx, c), ax, bnx), fmaf((float)PT_BYTE(ny, c), ay, bny)), fmaxf(fmaf((float)PT_BYTE(nz, c), az, bnz), from)); \
              const float tf_ = fminf(fminf(fmaf((float)PT_BYTE(fx, c), ax, bfx), fmaf((float)PT_BYTE(fy, c), ay, bfy)), fminf(fmaf((float)PT_BYTE(fz, c), az, bfz), lim)); \
              tn_out = tn_; hit_out = tn_ <= tf_; }
            PT_CHILD(0, t0, h0)                                               // link 0 is never empty
            PT_CHILD(1, t1, h1)                                               // nor is link 1
            PT_CHILD(2, t2, h2)
            PT_CHILD(3, t3, h3)
#undef PT_CHILD
        } else {
            // a direction component that is zero, denormal or beyond 2^30 either way: the planes themselves,
            // and the test the binary nodes get -- (plane - o) * inv is a signed infinity on an axis the ray
            // does not move along, which is what decides whether the ray runs inside the child's slab
            const float cx = __int_as_float(qa.x), cy = __int_as_float(qa.y), cz = __int_as_float(qa.z);
            const float sx = __int_as_float((qa.w & 0xff) << 23), sy = __int_as_float(((qa.w >> 8) & 0xff) << 23),
                        sz = __int_as_float(((qa.w >> 16) & 0xff) << 23);
#define PT_CHILD_BOX(c) lbvh4_plane(cx, sx, PT_BYTE(qb.x, c)), lbvh4_plane(cy, sy, PT_BYTE(qb.y, c)), lbvh4_plane(cz, sz, PT_BYTE(qb.z, c)), \
                        lbvh4_plane(cx, sx, PT_BYTE(qb.w, c)), lbvh4_plane(cy, sy, PT_BYTE(qc.x, c)), lbvh4_plane(cz, sz, PT_BYTE(qc.y, c))
            h0 = slab6(PT_CHILD_BOX(0), r.o, w.inv, from, lim, t0);
            h1 = slab6(PT_CHILD_BOX(1), r.o, w.inv, from, lim, t1);
            h2 = slab6(PT_CHILD_BOX(2), r.o, w.inv, from, lim, t2);
            h3 = slab6(PT_CHILD_BOX(3), r.o, w.inv, from, lim, t3);
#undef PT_CHILD_BOX
        }
#undef PT_BYTE
        h2 = h2 && lk.z != LBVH4_EMPTY;
        h3 = h3 && lk.w != LBVH4_EMPTY;
        int next = LBVH4_EMPTY;
        float tn = 0.0f;
        if (h0) { next = lk.x; tn = t0; }
        if (h1) {
            if (next == LBVH4_EMPTY) { next = lk.y; tn = t1; }
            else { const bool nearer = t1 < tn; lbvh_push(w, nearer ? next : lk.y, stack, spill); if (nearer) { next = lk.y; tn = t1; } }
        }
        if (h2) {
            if (next == LBVH4_EMPTY) { next = lk.z; tn = t2; }
            else { const bool nearer = t2 < tn; lbvh_push(w, nearer ? next : lk.z, stack, spill); if (nearer) { next = lk.z; tn = t2; } }
        }
        if (h3) {
            if (next == LBVH4_EMPTY) { next = lk.w; tn = t3; }
            else { const bool nearer = t3 < tn; lbvh_push(w, nearer ? next : lk.w, stack, spill); if (nearer) { next = lk.w; tn = t3; } }
        }
        if (next != LBVH4_EMPTY) { w.cur = next; return false; }
        return lbvh_pop(w, stack, spill);
    } else {
        const LbvhNode* n = &sv.lbvh[w.cur];
        float tl, tr;
        bool hl = slab(n->lmin, n->lmax, r.o, w.inv, from, lim, tl);
        bool hr = slab(n->rmin, n->rmax, r.o, w.inv, from, lim, tr);
        int cl = n->left, cr = n->right;
        if (hl && hr) {
            if (tr < tl) { int tmp = cl; cl = cr; cr = tmp; }
            lbvh_push(w, cr, stack, spill);
            w.cur = cl;
            return false;
        } else if (hl) { w.cur = cl; return false; }
        else if (hr) { w.cur = cr; return false; }
        return lbvh_pop(w, stack, spill);
    }
}
// One node or one leaf, whichever the lane is at.
template <uint32_t F>
__device__ __forceinline__ bool lbvh_step(const SceneView& sv, const Ray& r, float tmin, float tmax, LbvhWalk& w, int* stack, int* spill)
{
    if (w.cur >= 0) return lbvh_node_step<F>(sv, r, tmin, tmax, w, stack, spill);
    return lbvh_leaf_step<F>(sv, r, tmin, tmax, w, stack, spill);
}
// One round of a wave's walk.  A sphere or triangle test costs about what a node's box tests cost, and
// with every lane taking whichever step it is at, the wave pays for both kinds in nearly every round
// while few of its lanes are at a leaf.  Lanes at a node therefore always step; lanes at a leaf wait
// until SceneView::leaf_vote of them have gathered (or nobody is at a node), then test their leaves
// together.  Measured on MI355X against a vote of 1 (kernel ms): random spheres 1920x1080x64 under the
// shared walk 49.1 -> 43.4 at 12; RTNW final, tiles, 66.0 -> 61.0 at 16; 2 M triangles 3840x2160x16
// 58.2 -> 54.6 at 8 (a 1000-triangle mesh loses: 83.7 -> 98.5; such meshes keep 1).
// `walking`: this lane still has steps to take; returns true when its walk has ended.
constexpr int PROF_NODE = 0, PROF_LEAF = 1;
template <uint32_t F>
__device__ __forceinline__ bool lbvh_voted_step(const SceneView& sv, const Ray& r, float tmin, float tmax, LbvhWalk& w, int* stack, int* spill,
                                                bool walking, Prof& pf)
{
    const bool at_leaf = walking && w.cur < 0, at_node = walking && w.cur >= 0;
    const unsigned long long leaf_lanes = __ballot(at_leaf), node_lanes = __ballot(at_node);
    const int leaves = __popcll(leaf_lanes);
    const bool nodes_left = node_lanes != 0ull;
    bool ended = false;
    if (at_node) ended = lbvh_node_step<F>(sv, r, tmin, tmax, w, stack, spill);
    pf.mark(PROF_NODE, node_lanes);
    if (leaves >= sv.leaf_vote || !nodes_left) {
        if (at_leaf) ended = lbvh_leaf_step<F>(sv, r, tmin, tmax, w, stack, spill);
        pf.mark(PROF_LEAF, leaf_lanes);
    }
    return ended;
}
template <uint32_t F>
__device__ __forceinline__ bool lbvh_voted_step(const SceneView& sv, const Ray& r, float tmin, float tmax, LbvhWalk& w, int* stack, int* spill,
                                                bool walking)
{
    Prof none;
    return lbvh_voted_step<F>(sv, r, tmin, tmax, w, stack, spill, walking, none);
}
template <uint32_t F>
__device__ __forceinline__ bool lbvh_t(const SceneView& sv, const Ray& r, float tmin, float tmax,
                                       float& t_out, int& id_out, int& face_out, bool& odd_out, int* stack)
{
    int spill[LBVH_STACK - LBVH_LDS_STACK];
    LbvhWalk w;
    lbvh_begin(sv, r, tmax, w);
    bool walking = true;
    while (__ballot(walking) != 0ull) {
        if (lbvh_voted_step<F>(sv, r, tmin, tmax, w, stack, spill, walking)) walking = false;
    }
    t_out = w.best; id_out = w.best_id; face_out = w.best_face; odd_out = w.odd;
    return w.any;
}

// The in-order walk's answer for a finite ray with one or two zero direction components that lies
// in a plane of the tree's rectangles or box faces, without walking the list (a 2 M-triangle list
// takes a lane seconds, and a Cornell-like room sends about one such ray per 10^8 paths: a wall
// point at exactly the lamp's height sampling the lamp).  Which entries the in-order walk accepts
// depends on the running bound, but an entry that can be accepted under *some* bound is either
// one of the shapes with a rectangle in that plane (t = 0/0 there: accepted wherever they are;
// listed per plane when the tree is built) or a shape the ray really crosses, and those are the
// leaves an un-pruned tree walk reaches and whose test passes under the most permissive bound
// (NaN for the kinds that reject with `t > bound`, +inf for spheres, which reject a NaN bound).
// Those few candidates are then tested in list order with the walk's real, running bound: every
// other entry would have been rejected whatever the bound, so this is the walk's result.
// Returns 1 (hit) / 0 (miss), or -1 when the ray or the tree is not of that kind (wrapped
// rectangles or boxes in the tree, a zero direction, non-finite values) or the candidates
// do not fit: then the caller walks the list.
constexpr int GATHER_MAX = 24;
__device__ __forceinline__ int list_place(const SceneView& sv, int id) { return sv.lbvh_order ? sv.lbvh_order[id] : id; }
// Kept out of line (arguments and result by value, the scene through its copy in device memory):
// inlined, this rare path cost the kernels around it registers and spills.
struct InPlaneHit { int found; float t; int id, face; };
template <uint32_t F>
__device__ __attribute__((noinline)) InPlaneHit lbvh_in_plane_t(const SceneView* view, float ox, float oy, float oz, float dx, float dy, float dz,
                                                                float time, float tmin, float tmax)
{
    const SceneView& sv = *view;
    Ray r; r.o = mk(ox, oy, oz); r.d = mk(dx, dy, dz); r.time = time;
    InPlaneHit out; out.found = -1; out.t = tmax; out.id = -1; out.face = 0;
    int id_out = -1, face_out = 0;
    if (sv.tree_wrapped || sv.tree_plane_first == nullptr) return out;
    const float s = fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z) + fabsf(r.d.x) + fabsf(r.d.y) + fabsf(r.d.z);
    if (!(s <= TREE_RAY_LIMIT)) return out;
    const int zeros = (r.d.x == 0.0f) + (r.d.y == 0.0f) + (r.d.z == 0.0f);
    if (zeros < 1 || zeros > 2) return out;                        // a zero direction makes every test NaN or infinite
    int cand[GATHER_MAX];
    int n = 0;
    for (int axis = 0; axis < 3; axis++) {
        const float dk = axis == 0 ? r.d.x : (axis == 1 ? r.d.y : r.d.z);
        const float o = axis == 0 ? r.o.x : (axis == 1 ? r.o.y : r.o.z);
        if (dk != 0.0f) continue;
        for (int k = axis == 0 ? 0 : sv.tree_plane_end[axis - 1]; k < sv.tree_plane_end[axis]; k++) {
            if (!(sv.tree_planes[k] == o)) continue;
            for (int m = sv.tree_plane_first[k]; m < sv.tree_plane_first[k + 1]; m++) {
                const int id = sv.tree_plane_members[m];
                bool known = false;                                // a box can have faces in the planes of both axes
                for (int c = 0; c < n; c++) known = known || cand[c] == id;
                if (known) continue;
                if (n == GATHER_MAX) return out;
                cand[n++] = id;
            }
        }
    }
    const int n_in_plane = n;
    const V3 inv = mk(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    int pending[LBVH_STACK];
    int sp = 0, cur = sv.lbvh_root;
    while (true) {
        if (cur >= 0) {
            const LbvhNode* nd = &sv.lbvh[cur];
            float tl, tr;
            const float from = sv.lbvh_cull != 0 ? tmin : 0.0f;    // see lbvh_step
            const bool hl = slab(nd->lmin, nd->lmax, r.o, inv, from, INFINITY, tl);
            const bool hr = slab(nd->rmin, nd->rmax, r.o, inv, from, INFINITY, tr);
            if (hl && hr) {
                if (sp == LBVH_STACK) return out;
                pending[sp++] = nd->right;
                cur = nd->left;
                continue;
            }
            if (hl) { cur = nd->left; continue; }
            if (hr) { cur = nd->right; continue; }
        } else {
            const int id = ~cur;
            float t; int face;
            const float loosest = inclusive_kind(sv.shapes[id].type) ? __builtin_nanf("") : INFINITY;
            if (leaf_t<F>(sv, id, r, tmin, loosest, t, face)) {
                bool known = false;
                for (int c = 0; c < n_in_plane; c++) known = known || cand[c] == id;
                if (!known) {
                    if (n == GATHER_MAX) return out;
                    cand[n++] = id;
                }
            }
        }
        if (sp == 0) break;
        cur = pending[--sp];
    }
    for (int a = 1; a < n; a++) {                                  // into list order
        const int id = cand[a], place = list_place(sv, id);
        int b = a;
        while (b > 0 && list_place(sv, cand[b - 1]) > place) { cand[b] = cand[b - 1]; b--; }
        cand[b] = id;
    }
    bool any = false;
    float closest = tmax;
    for (int c = 0; c < n; c++) {
        float t; int face;
        if (leaf_t<F>(sv, cand[c], r, tmin, closest, t, face)) { any = true; closest = t; id_out = cand[c]; face_out = face; }
    }
    out.found = any ? 1 : 0; out.t = closest; out.id = id_out; out.face = face_out;
    return out;
}

// ---- world signatures -------------------------------------------------------------------------------
// A world list whose sequence of entry kinds is known at compile time is walked by straight-line
// code: entry k of the list is addressed with a constant, so every scene value the walk reads is
// loop-invariant and the compiler lifts the loads out of the bounce loop (they stay in scalar
// registers for the life of the kernel), and no type dispatch is left.  On the Cornell box this is
// 14 % faster at full occupancy and 30 % faster when a GPU holds an eighth of the frame, where the
// generic walk's dependent scene loads are no longer hidden by other waves.  The arithmetic of every
// entry is the generic walk's, in the same order, so the hits are the same.
enum : int { WK_RECT_XY = 1, WK_RECT_XZ = 2, WK_RECT_YZ = 3, WK_BOX_TR = 4, WK_SPHERE = 5 };
struct AnyWorld {};                                        // any list: the generic walk
template <int... K> struct WorldSig { static constexpr int count = (int)sizeof...(K); };
// The same signature walked without control flow between the tests (rect_ok): a little more
// vector work per test, but one scheduling block, which pays when few waves share a SIMD.
template <class Sig> struct Branchless : Sig {};
template <class W> struct is_branchless { static constexpr bool value = false; };
template <class Sig> struct is_branchless<Branchless<Sig>> { static constexpr bool value = true; };
// cornell_box (src/ptMain.cu:286-317): five walls and the lamp, then the two Translate(RotateY(Box))
using CornellWorld = WorldSig<WK_RECT_YZ, WK_RECT_YZ, WK_RECT_XZ, WK_RECT_XZ, WK_RECT_XZ, WK_RECT_XY, WK_BOX_TR, WK_BOX_TR>;

// The entries' constants are read from a flat table built when the signature is matched: one
// SIG_SLOT-float slot per world entry at a constant offset from one base, so every load of the
// walk is independent of every other (the scene tables would be reached through world[k].index ->
// shape -> wrapper list, three dependent scalar loads per entry that no other work of the wave
// can cover).  Slot layout: rectangle a0, a1, b0, b1, k; sphere centre, radius; wrapped box
// pmin, pmax, translation, sin, cos; last word the shape index.
constexpr int SIG_SLOT = 16;
template <uint32_t F, int KIND, bool TAME, bool FLAT, class D>
__device__ __forceinline__ void world_entry(const SceneView& sv, int k, const Ray& r, const D& dx, const D& dy, const D& dz,
                                            float tmin, float& closest, int& code)
{
    // `code` = 8 * entry + face of the nearest accepted test so far (-1: none).  A box's faces are
    // tested against the list's own shrinking bound: the box's inner list starts from that bound and
    // the list takes the box's answer whenever there is one, so the flattened walk accepts the
    // same tests in the same order (include/ptRectangle.h:199-217, src/ptHitableList.cu:12-27).
    const float* q = sv.sig + SIG_SLOT * k;
    float t;
    if constexpr (!FLAT) {
        int face = 0;
        bool hit = false;
        if (KIND == WK_RECT_XY) hit = rect_t(r.o.z, dz, r.o.x, r.d.x, r.o.y, r.d.y, q[0], q[1], q[2], q[3], q[4], tmin, closest, t);
        if (KIND == WK_RECT_XZ) hit = rect_t(r.o.y, dy, r.o.x, r.d.x, r.o.z, r.d.z, q[0], q[1], q[2], q[3], q[4], tmin, closest, t);
        if (KIND == WK_RECT_YZ) hit = rect_t(r.o.x, dx, r.o.y, r.d.y, r.o.z, r.d.z, q[0], q[1], q[2], q[3], q[4], tmin, closest, t);
        if (KIND == WK_SPHERE) hit = sphere_t(mk(q[0], q[1], q[2]), q[3], r, tmin, closest, t);
        if (KIND == WK_BOX_TR) {
            Ray lr = r;
            lr.o = lr.o - mk(q[6], q[7], q[8]);
            const float sinT = q[9], cosT = q[10];
            const float ox = cosT * lr.o.x - sinT * lr.o.z;
            const float oz = sinT * lr.o.x + cosT * lr.o.z;
            const float ldx = cosT * lr.d.x - sinT * lr.d.z;
            const float ldz = sinT * lr.d.x + cosT * lr.d.z;
            lr.o.x = ox; lr.o.z = oz; lr.d.x = ldx; lr.d.z = ldz;
            if constexpr (TAME) {
                const bool tame = fabsf(lr.o.x) + fabsf(lr.o.y) + fabsf(lr.o.z) <= RCP_BIG && fabsf(ldx) >= RCP_D_MIN && fabsf(ldz) >= RCP_D_MIN;
                if (tame) hit = box_t(q, lr, rcp_refined(ldx), dy, rcp_refined(ldz), tmin, closest, t, face);
                else hit = box_t(q, lr, tmin, closest, t, face);
            } else {
                hit = box_t(q, lr, tmin, closest, t, face);
            }
        }
        if (hit) { closest = t; code = 8 * k + face; }
        return;
    }
    if (KIND == WK_RECT_XY) { if (rect_ok(r.o.z, dz, r.o.x, r.d.x, r.o.y, r.d.y, q[0], q[1], q[2], q[3], q[4], tmin, closest, t)) { closest = t; code = 8 * k; } }
    if (KIND == WK_RECT_XZ) { if (rect_ok(r.o.y, dy, r.o.x, r.d.x, r.o.z, r.d.z, q[0], q[1], q[2], q[3], q[4], tmin, closest, t)) { closest = t; code = 8 * k; } }
    if (KIND == WK_RECT_YZ) { if (rect_ok(r.o.x, dx, r.o.y, r.d.y, r.o.z, r.d.z, q[0], q[1], q[2], q[3], q[4], tmin, closest, t)) { closest = t; code = 8 * k; } }
    if (KIND == WK_SPHERE) { if (sphere_t(mk(q[0], q[1], q[2]), q[3], r, tmin, closest, t)) { closest = t; code = 8 * k; } }
    if (KIND == WK_BOX_TR) {                               // Translate(RotateY(Box)): to_local's arithmetic, written out
        Ray lr = r;
        lr.o = lr.o - mk(q[6], q[7], q[8]);
        const float sinT = q[9], cosT = q[10];
        const float ox = cosT * lr.o.x - sinT * lr.o.z;
        const float oz = sinT * lr.o.x + cosT * lr.o.z;
        const float ldx = cosT * lr.d.x - sinT * lr.d.z;
        const float ldz = sinT * lr.d.x + cosT * lr.d.z;
        lr.o.x = ox; lr.o.z = oz; lr.d.x = ldx; lr.d.z = ldz;
        if constexpr (TAME) {
            // the rotation keeps d.y (shared reciprocal) and at most doubles magnitudes; what is left
            // to test is that the turned components did not cancel and that the local origin is tame
            const bool tame = fabsf(lr.o.x) + fabsf(lr.o.y) + fabsf(lr.o.z) <= RCP_BIG && fabsf(ldx) >= RCP_D_MIN && fabsf(ldz) >= RCP_D_MIN;
            if (tame) box_sel(q, lr, rcp_refined(ldx), dy, rcp_refined(ldz), tmin, closest, code, 8 * k);
            else box_sel(q, lr, lr.d.x, lr.d.y, lr.d.z, tmin, closest, code, 8 * k);
        } else {
            box_sel(q, lr, lr.d.x, lr.d.y, lr.d.z, tmin, closest, code, 8 * k);
        }
    }
}
// Rays whose origin and direction components lie in the range where the shortened division is
// the whole division (see rcp_refined) take it, with one reciprocal per axis; any other ray
// (an axis-parallel direction, an overflowing origin, NaN) takes the compiler's full sequence.
// The scene's own constants are range-checked when the signature is matched on the host.
template <uint32_t F, bool FLAT, int... K>
__device__ __forceinline__ bool world_hit_sig(const SceneView& sv, const Ray& r, float tmin, float tmax, Hit& h, WorldSig<K...>)
{
    float closest = tmax;
    int code = -1;
    int k = 0;
    const float ax = fabsf(r.d.x), ay = fabsf(r.d.y), az = fabsf(r.d.z);
    const bool tame = tmin >= RCP_TMIN_MIN && fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z) <= RCP_BIG && ax + ay + az <= RCP_BIG &&
                      ax >= RCP_D_MIN && ay >= RCP_D_MIN && az >= RCP_D_MIN;
    if (tame) {
        const Rcp cx = rcp_refined(r.d.x), cy = rcp_refined(r.d.y), cz = rcp_refined(r.d.z);
        ((world_entry<F, K, true, FLAT>(sv, k++, r, cx, cy, cz, tmin, closest, code)), ...);
    } else {
        ((world_entry<F, K, false, FLAT>(sv, k++, r, r.d.x, r.d.y, r.d.z, tmin, closest, code)), ...);
    }
    if (code < 0) return false;
    h.t = closest;
    h.id = __float_as_int(sv.sig[SIG_SLOT * (code >> 3) + SIG_SLOT - 1]);
    h.face = code & 7;
    h.sig_code = code;
    return true;
}
template <class W> struct is_any_world { static constexpr bool value = false; };
template <> struct is_any_world<AnyWorld> { static constexpr bool value = true; };

// True when the winner `id` writes no uv of its own (a plain sphere, a medium) and its material's
// texture can read uv: then the uv of an earlier accepted hit shows through (see Hit).
template <uint32_t F>
__device__ __forceinline__ bool hit_reads_stale_uv(const SceneView& sv, int id)
{
    int m;
    if ((F & F_MEDIUM) && id >= PT_ID_MEDIUM_BASE) m = sv.media[id - PT_ID_MEDIUM_BASE].material;
    else if (id >= 0 && !writes_uv(sv.shapes[id].type)) m = sv.shapes[id].material;
    else return false;
    const int tex = m >= 0 ? sv.materials[m].texture : -1;
    const int kind = tex >= 0 ? sv.textures[tex].type : PT_TEX_CONSTANT;
    return kind == PT_TEX_IMAGE || kind == PT_TEX_CHECKER;
}

// world->hit(ray, tmin, tmax) (src/ptHitableList.cu:12-27).
// lbvh_stack: this lane's column of the LDS traversal stack (stride 64 ints), or nullptr.
template <uint32_t F, class W = AnyWorld>
__device__ __forceinline__ bool world_hit(const SceneView& sv, const Ray& r, float tmin, float tmax, Hit& h, Pcg& g,
                                          int* lbvh_stack)
{
    h.id = PT_ID_MISS; h.face = 0; h.t = tmax; h.uv_id = PT_ID_MISS; h.uv_face = 0; h.uv_t = 0.0f; h.sig_code = -1;
    if constexpr (!is_any_world<W>::value) {
        return world_hit_sig<F, is_branchless<W>::value>(sv, r, tmin, tmax, h, W{});
    }
    bool any = false;
    float closest = tmax;
    int first = 0;
    bool uv_unknown = false;
    const Pcg g_before = g;                                    // media draw inside the list walk
    if (F & F_LBVH) {
        // The tree stands for the first lbvh_prefix entries of the list (all of it when the list
        // holds shapes only); media and what follows them are walked in list order from there.
        float t = 0; int id = 0, face = 0;
        bool odd = false;
        const bool enter = !accepts_nan<F>::value || ray_may_enter_tree(sv, r);
        if (enter) {
            if (lbvh_t<F>(sv, r, tmin, tmax, t, id, face, odd, lbvh_stack)) { any = true; closest = t; h.t = t; h.id = id; h.face = face; }
        }
        const bool in_order = accepts_nan<F>::value && (!enter || odd);      // see accepts_nan
        if (!accepts_nan<F>::value) { if (!(F & (F_MEDIUM | F_REFBVH | F_IMAGE))) return any; }
        else if (!(F & (F_MEDIUM | F_REFBVH | F_IMAGE))) { if (!in_order) return any; }
        first = sv.lbvh_prefix;
        if (in_order) {
            any = false; closest = tmax; first = 0;
            h.id = PT_ID_MISS; h.face = 0; h.t = tmax;
            if (!(F & (F_REFBVH | F_IMAGE))) {                   // the tree's part of the list from a few candidates, if the ray is of that kind
                const InPlaneHit got = lbvh_in_plane_t<F>(sv.self, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.time, tmin, tmax);
                if (got.found >= 0) {
                    first = sv.lbvh_prefix;
                    if (got.found > 0) { any = true; closest = got.t; h.t = got.t; h.id = got.id; h.face = got.face; }
                }
            }
        }
        if ((F & F_IMAGE) && sv.reads_uv && any) {
            // Stale uv (see Hit): a plain sphere reports the uv of the last hit the in-order walk
            // accepted from a shape that writes uv.  Hits are accepted in decreasing t, so when the
            // nearest shape of the tree's part of the list writes uv, it is that hit.  When it is a
            // plain sphere the last uv-writing hit before it is not known here; that only matters if
            // no later entry writes uv and the search ends on a plain sphere whose texture can read
            // uv, and then the ray is searched again the in-order way (below).
            if (writes_uv(sv.shapes[id].type)) { h.uv_id = id; h.uv_face = face; h.uv_t = t; }
            else uv_unknown = true;
        }
        if ((F & F_REFBVH) && any && sv.lbvh_group) {
            // The nearest leaf came out of the reference BVH the list starts with.  The reference only
            // finds it if the box of every node above it lets the ray in (AABB::hit,
            // include/ptAABB.h:30-48), and that test can turn away a ray that grazes a box edge although
            // the leaf's own test accepts it.  Boxes grow towards the root and the test is monotone in
            // the box (its subtractions, multiplications and comparisons all are), so the box of the
            // node that names the leaf decides for all of them; the bound is the caller's, because
            // nothing precedes that tree in the list.  If the box says no -- about once in 10^7 searches
            // on the RTNW final scene -- the ray is searched again the reference's way, in list order.
            if (sv.lbvh_group[id] >= 0) {
                const PT_SCENE float* gate = sv.ref_leaf_gate + 6 * (size_t)id;
                const V3 inv = mk(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);
                if (!aabb_hit_inv(gate, gate + 3, r.o, inv, tmin, tmax)) {
                    any = false; closest = tmax; first = 0;
                    h.id = PT_ID_MISS; h.face = 0; h.t = tmax;
                }
            }
        }
    }
    const int n = sv.n_world;
    while (true) {
        for (int i = first; i < n; i++) {
            // Without a tree every lane walks the whole list, entry by entry: the index is the same in all lanes, and
            // saying so keeps the entries' loads on the scalar unit whatever the control flow around the search looks like.
            const int entry = (F & F_LBVH) ? i : __builtin_amdgcn_readfirstlane(i);
            const PtObject ob = sv.world[entry];
            float t; int id = ob.index, face = 0;
            bool hit = false;
            if (!(F & (F_MEDIUM | F_REFBVH)) || ob.type == PT_OBJ_SHAPE) {
                hit = shape_t<F>(sv, ob.index, r, tmin, closest, t, face);
            } else if ((F & F_MEDIUM) && ob.type == PT_OBJ_MEDIUM) {
                hit = medium_t<F>(sv, ob.index, r, tmin, closest, t, g);
                id = PT_ID_MEDIUM_BASE + ob.index;
            } else if ((F & F_REFBVH) && ob.type == PT_OBJ_BVH) {
                hit = refbvh_t<F>(sv, ob.index, r, tmin, closest, t, id, face, lbvh_stack);
            }
            if (hit) {
                any = true; closest = t; h.t = t; h.id = id; h.face = face;
                if (F & F_IMAGE) {
                    if (id < PT_ID_MEDIUM_BASE && writes_uv(sv.shapes[id].type)) { h.uv_id = id; h.uv_face = face; h.uv_t = t; uv_unknown = false; }
                }
            }
        }
        if (!((F & F_LBVH) && (F & F_IMAGE))) break;
        // the one case the tree could not settle: the search ends on a plain sphere that may read a uv the tree did not see
        bool again = false;
        if (uv_unknown && any) again = hit_reads_stale_uv<F>(sv, h.id);
        if (!again) break;
        uv_unknown = false; g = g_before; first = 0;
        any = false; closest = tmax;
        h.id = PT_ID_MISS; h.face = 0; h.t = tmax; h.uv_id = PT_ID_MISS; h.uv_face = 0; h.uv_t = 0.0f;
    }
    return any;
}

// ---- hit completion: the HitRecord fields the shaders read --------------------------------------------
struct Surface { V3 p, n; float u, v; int32_t mat; };

__device__ __forceinline__ void get_uv(V3 p, float& u, float& v)     // include/ptSphere.h:22-28
{
    float phi = ptm::atan2f_glibc(p.z, p.x);
    float theta = ptm::asinf_glibc(p.y);
    u = 1 - (phi + PI_F) / (2 * PI_F);
    v = (theta + PI_F / 2) / PI_F;
}

// uv a uv-writing leaf records for a hit at parameter t (object-space ray r).
template <uint32_t F>
__device__ __forceinline__ void leaf_uv(const SceneView& sv, const PtShape* s, const Ray& r, float t, int face, float& u, float& v)
{
    const int type = s->type;
    u = 0.0f; v = 0.0f;
    int rtype = type;
    float a0 = s->p[0], a1 = s->p[1], b0 = s->p[2], b1 = s->p[3];
    if ((F & F_BOX) && type == PT_SHAPE_BOX) {
        const float* q = s->p;
        if (face < 2) { rtype = PT_SHAPE_RECT_XY; a0 = q[0]; a1 = q[3]; b0 = q[1]; b1 = q[4]; }
        else if (face < 4) { rtype = PT_SHAPE_RECT_XZ; a0 = q[0]; a1 = q[3]; b0 = q[2]; b1 = q[5]; }
        else { rtype = PT_SHAPE_RECT_YZ; a0 = q[1]; a1 = q[4]; b0 = q[2]; b1 = q[5]; }
    }
    if ((F & (F_RECT | F_BOX)) && (rtype == PT_SHAPE_RECT_XY || rtype == PT_SHAPE_RECT_XZ || rtype == PT_SHAPE_RECT_YZ)) {
        float oa, da, ob, db;
        if (rtype == PT_SHAPE_RECT_XY) { oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; }
        else if (rtype == PT_SHAPE_RECT_XZ) { oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; }
        else { oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; }
        float a = oa + t * da, b = ob + t * db;
        u = (a - a0) / (a1 - a0);
        v = (b - b0) / (b1 - b0);
        return;
    }
    if ((F & F_MSPHERE) && type == PT_SHAPE_MOVING_SPHERE) { get_uv(at(r, t), u, v); return; }
    if ((F & F_TRIANGLE) && type == PT_SHAPE_TRIANGLE) {
        const float* q = s->p;
        V3 v0 = mk(q[0], q[1], q[2]), v1 = mk(q[3], q[4], q[5]), v2 = mk(q[6], q[7], q[8]);
        V3 edge1 = v1 - v0, edge2 = v2 - v0;
        V3 pvec = cross(r.d, edge2);
        float det = dot(edge1, pvec);
        V3 tvec = r.o - v0;
        float bu = dot(tvec, pvec);
        V3 qvec = cross(tvec, edge1);
        float bv = dot(r.d, qvec);
        const float inv_det = 1 / det;
        bu *= inv_det; bv *= inv_det;
        float b0f = (float)(1.0 - (double)bu - (double)bv);
        if (s->tri_uv >= 0) {
            const PtTriUV tu = sv.tri_uvs[s->tri_uv];
            u = tu.uv0[0] * b0f + tu.uv1[0] * bu + tu.uv2[0] * bv;
            v = tu.uv0[1] * b0f + tu.uv1[1] * bu + tu.uv2[1] * bv;
        } else {
            u = 0.0f * b0f + 0.0f * bu + 0.0f * bv;
            v = u;
        }
    }
}

// uv left in the list's record by the last uv-writing hit the walk accepted before the winner
// (see Hit): what a plain sphere or a medium, which write no uv of their own, hand to the shader.
template <uint32_t F>
__device__ __forceinline__ void stale_uv(const SceneView& sv, const Ray& ray, const Hit& h, float& u, float& v)
{
    const PtShape* us = &sv.shapes[h.uv_id];
    Ray ur = ray;
    if (F & F_XFORM) { if (us->xform_count > 0) ur = to_local(sv, us->xform_first, us->xform_count, ray); }
    leaf_uv<F>(sv, us, ur, h.uv_t, h.uv_face, u, v);
}

// Normal of a rectangle or of face `face` of a box in the shape's own frame (src/ptRectangle.cu:25,79,133,
// include/ptRectangle.h:199-212: the odd faces of a box sit under FlipNormals).
PT_FN V3 flat_normal_local(int type, int face)
{
    if (type == PT_SHAPE_RECT_XY) return mk(0, 0, 1);
    if (type == PT_SHAPE_RECT_XZ) return mk(0, 1, 0);
    if (type == PT_SHAPE_RECT_YZ) return mk(1, 0, 0);
    V3 n = (face < 2) ? mk(0, 0, 1) : (face < 4) ? mk(0, 1, 0) : mk(1, 0, 0);
    if (face & 1) n = -n;
    return n;
}
// RotateY::hit's rotation of the normal back to the world (include/ptRectangle.h:385-389); Translate leaves it alone.
PT_FN V3 rotate_y_to_world(V3 n, float sinT, float cosT)
{
    return mk(cosT * n.x + sinT * n.z, n.y, -sinT * n.x + cosT * n.z);
}

// ---- finished surfaces of a signature world ------------------------------------------------------------
// A signature world (Cornell box) has a handful of faces, each with one normal, one orthonormal
// basis and, when every texture is a constant, one colour.  They are worked out once on the host, by
// the functions above (the kernels' own arithmetic: IEEE operations in the same order, so the same
// bits), one SIG_FACE-float row per (entry, face): normal, basis u, v, w, colour (albedo or emission),
// material kind.  shade_sig reads the winner's row instead of chasing shape -> material -> texture
// and rebuilding the basis (a square root and six divisions) at every bounce.
constexpr int SIG_FACE = 16;
struct SigFace { V3 n; Onb uvw; V3 colour; int32_t material_type; };
PT_FN SigFace sig_face_make(int shape_type, int face, bool flip, bool rotated, float sinT, float cosT)
{
    SigFace f;
    f.n = flat_normal_local(shape_type, face);
    if (rotated) f.n = rotate_y_to_world(f.n, sinT, cosT);
    if (flip) f.n = -f.n;
    f.uvw = onb_from_w(f.n);
    f.colour = mk(0, 0, 0);
    f.material_type = 0;
    return f;
}

template <uint32_t F>
__device__ __forceinline__ Surface complete_hit(const SceneView& sv, const Ray& ray, const Hit& h)
{
    Surface sf;
    sf.u = 0.0f; sf.v = 0.0f;
    if ((F & F_MEDIUM) && h.id >= PT_ID_MEDIUM_BASE) {            // include/ptMedium.h:46-49: t, p, normal, material; uv stays
        sf.p = at(ray, h.t);
        sf.n = mk(1, 0, 0);
        sf.mat = sv.media[h.id - PT_ID_MEDIUM_BASE].material;
        if (F & F_IMAGE) { if (h.uv_id >= 0) stale_uv<F>(sv, ray, h, sf.u, sf.v); }
        return sf;
    }
    const PtShape* s = &sv.shapes[h.id];
    const int type = s->type;
    Ray r = ray;
    const int nxf = (F & F_XFORM) ? s->xform_count : 0;
    if (F & F_XFORM) { if (nxf > 0) r = to_local(sv, s->xform_first, nxf, ray); }
    sf.mat = s->material;
    sf.p = at(r, h.t);
    sf.n = mk(0, 0, 0);
    if ((F & F_SPHERE) && type == PT_SHAPE_SPHERE) {
        sf.n = (sf.p - mk(s->p[0], s->p[1], s->p[2])) / s->p[3];
    } else if ((F & F_MSPHERE) && type == PT_SHAPE_MOVING_SPHERE) {
        sf.n = (sf.p - msphere_center(s->p, r.time)) / s->p[8];
    } else if ((F & F_RECT) && (type == PT_SHAPE_RECT_XY || type == PT_SHAPE_RECT_XZ || type == PT_SHAPE_RECT_YZ)) {
        sf.n = flat_normal_local(type, 0);
    } else if ((F & F_BOX) && type == PT_SHAPE_BOX) {
        sf.n = flat_normal_local(type, h.face);
    } else if ((F & F_TRIANGLE) && type == PT_SHAPE_TRIANGLE) {    // src/ptTriangle.cu:68-76
        const float* q = s->p;
        V3 v0 = mk(q[0], q[1], q[2]), v1 = mk(q[3], q[4], q[5]), v2 = mk(q[6], q[7], q[8]);
        V3 edge1 = v1 - v0, edge2 = v2 - v0;
        V3 pvec = cross(r.d, edge2);
        float det = dot(edge1, pvec);
        V3 tvec = r.o - v0;
        float bu = dot(tvec, pvec);
        V3 qvec = cross(tvec, edge1);
        float bv = dot(r.d, qvec);
        const float inv_det = 1 / det;
        bu *= inv_det; bv *= inv_det;
        sf.p = (1 - bu - bv) * v0 + bu * v1 + bv * v2;
        V3 nn = cross(edge1, edge2);
        float k = 1 / len(nn);
        sf.n = mk(nn.x * k, nn.y * k, nn.z * k);
    }
    if (F & F_IMAGE) {
        // uv is read by image textures only
        if (writes_uv(type)) leaf_uv<F>(sv, s, r, h.t, h.face, sf.u, sf.v);
        else if (h.uv_id >= 0) stale_uv<F>(sv, ray, h, sf.u, sf.v);
    }
    if (F & F_XFORM) { if (nxf > 0) to_world(sv, s->xform_first, nxf, sf.p, sf.n); }
    if (s->flags & PT_FLIP_NORMAL) sf.n = -sf.n;                   // include/ptRectangle.h:148-156
    return sf;
}

// ---- Perlin noise (src/ptNoise.cu) --------------------------------------------------------------------
// Ken Perlin's 2002 reference permutation (src/ptNoise.cu:47-112 holds it twice over).
__device__ __constant__ unsigned char g_perm[256] = {
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
__device__ __forceinline__ int perm(int i) { return g_perm[i & 255]; }
__device__ __forceinline__ float fade(float t) { return t * t * t * (t * (t * 6 - 15) + 10); }
__device__ __forceinline__ float lerpf(float t, float a, float b) { return a + t * (b - a); }
__device__ __forceinline__ float grad(int hash, float x, float y, float z)
{
    int h = hash & 15;
    float u = ((h < 8) || (h == 12) || (h == 13)) ? x : y;
    float v = ((h < 4) || (h == 12) || (h == 13)) ? y : z;
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}
__device__ inline float noise3(V3 p)                                // src/ptNoise.cu:114-140
{
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    int xi = (int)fx & 255, yi = (int)fy & 255, zi = (int)fz & 255;
    float x = p.x - fx, y = p.y - fy, z = p.z - fz;
    float u = fade(x), v = fade(y), w = fade(z);
    int A = perm(xi) + yi, AA = perm(A) + zi, AB = perm(A + 1) + zi;
    int B = perm(xi + 1) + yi, BA = perm(B) + zi, BB = perm(B + 1) + zi;
    return lerpf(w, lerpf(v, lerpf(u, grad(perm(AA), x, y, z), grad(perm(BA), x - 1, y, z)),
                             lerpf(u, grad(perm(AB), x, y - 1, z), grad(perm(BB), x - 1, y - 1, z))),
                    lerpf(v, lerpf(u, grad(perm(AA + 1), x, y, z - 1), grad(perm(BA + 1), x - 1, y, z - 1)),
                             lerpf(u, grad(perm(AB + 1), x, y - 1, z - 1), grad(perm(BB + 1), x - 1, y - 1, z - 1))));
}
__device__ inline float turbulence7(V3 p)                           // src/ptNoise.cu:142-154, depth 7
{
    float accum = 0, weight = 1;
    V3 tp = p;
    for (int i = 0; i < 7; i++) {
        accum += weight * noise3(tp);
        weight *= 0.5f;
        tp = tp * 2.0f;
    }
    return fabsf(accum);
}

// ---- textures (include/ptTexture.h) --------------------------------------------------------------------
// float -> int as the reference's x86-64 host build converts (cvttss2si): NaN and every value
// outside [-2^31, 2^31) give INT_MIN, where the GPU's conversion would give 0 or saturate.  A moving
// sphere's uv is NaN whenever |p.y| > 1 (asinf of the un-normalised point, include/ptSphere.h:22-28).
__device__ __forceinline__ int f2i_host(float x)
{
    return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000;
}
template <uint32_t F>
__device__ __forceinline__ V3 texture_value(const SceneView& sv, int tex, float u, float v, V3 p)
{
    PtTexture t = sv.textures[tex];
    if (F & F_CHECKER) {
        // nested checkers resolve iteratively (:90-97)
        for (int guard = 0; guard < 8 && t.type == PT_TEX_CHECKER; guard++) {
            float sines = ptm::sinf_glibc(t.scale * p.x) * ptm::sinf_glibc(t.scale * p.y) * ptm::sinf_glibc(t.scale * p.z);
            t = sv.textures[(sines < 0) ? t.odd : t.even];
        }
    }
    if ((F & F_NOISE) && t.type == PT_TEX_NOISE) {                  // :142-146
        float n = 0.5f * (1 + ptm::sinf_glibc(t.scale * p.z + 10 * turbulence7(p)));
        return mk(n, n, n);
    }
    if ((F & F_IMAGE) && t.type == PT_TEX_IMAGE) {                  // :187-199
        int nx = t.nx, ny = t.ny;
        int i = f2i_host(u * nx);
        int j = f2i_host((1 - v) * ny - 0.001f);
        i = max(0, i); j = max(0, j);
        i = min(i, nx - 1); j = min(j, ny - 1);
        const PT_SCENE uint8_t* d = sv.texels + t.texel_offset + 3 * i + 3 * nx * j;
        return mk((int)d[0] / 255.0f, (int)d[1] / 255.0f, (int)d[2] / 255.0f);
    }
    return mk(t.color[0], t.color[1], t.color[2]);                  // :47-50
}

// ---- light shapes: pdfValue / random ---------------------------------------------------------------------
template <uint32_t F>
__device__ __forceinline__ float light_shape_pdf(const SceneView& sv, int shape, V3 o, V3 v)
{
    const PtShape* s = &sv.shapes[shape];
    Ray r; r.o = o; r.d = v; r.time = 0;
    if ((F & F_LIGHT_MISC) && (s->flags & PT_FLIP_NORMAL)) return 0;                       // FlipNormals has no pdfValue of its own
    if (!(F & F_LIGHT_MISC) || (s->type == PT_SHAPE_RECT_XZ && s->xform_count == 0)) {   // include/ptRectangle.h:77-89
        float t;
        if (rect_t(r.o.y, r.d.y, r.o.x, r.d.x, r.o.z, r.d.z, s->p[0], s->p[1], s->p[2], s->p[3], s->p[4], 0.001f, FLT_MAX, t)) {
            float area = (s->p[1] - s->p[0]) * (s->p[3] - s->p[2]);
            float dist_sqrd = t * t * sqlen(v);
            float cosine = fabsf(dot(v, mk(0, 1, 0)) / len(v));
            return dist_sqrd / (cosine * area);
        }
        return 0;
    }
    if (s->type == PT_SHAPE_SPHERE && s->xform_count == 0) {                               // include/ptSphere.h:48-58
        float t;
        V3 center = mk(s->p[0], s->p[1], s->p[2]);
        if (sphere_t(center, s->p[3], r, 0.001f, FLT_MAX, t)) {
            float cos_theta_max = sqrtf(1 - s->p[3] * s->p[3] / sqlen(center - o));
            float solid_angle = 2 * PI_F * (1 - cos_theta_max);
            return 1 / solid_angle;
        }
        return 0;
    }
    return 0;                                                                              // include/ptHitable.h:58
}
template <uint32_t F>
__device__ __forceinline__ V3 light_shape_random(const SceneView& sv, int shape, V3 o, Pcg& g)
{
    const PtShape* s = &sv.shapes[shape];
    if ((F & F_LIGHT_MISC) && (s->flags & PT_FLIP_NORMAL)) return mk(1, 0, 0);            // FlipNormals has no random() of its own
    if (!(F & F_LIGHT_MISC) || (s->type == PT_SHAPE_RECT_XZ && s->xform_count == 0)) {   // include/ptRectangle.h:91-95
        float rz = g.rand();       // g++ evaluates the constructor arguments right to left
        float rx = g.rand();
        V3 rp = mk(s->p[0] + rx * (s->p[1] - s->p[0]), s->p[4], s->p[2] + rz * (s->p[3] - s->p[2]));
        return rp - o;
    }
    if (s->type == PT_SHAPE_SPHERE && s->xform_count == 0) {                               // include/ptSphere.h:60-67
        V3 dir = mk(s->p[0], s->p[1], s->p[2]) - o;
        float dist_sqrd = sqlen(dir);
        Onb uvw = onb_from_w(dir);
        return onb_local(uvw, random_to_unit_sphere(s->p[3], dist_sqrd, g));
    }
    return mk(1, 0, 0);                                                                    // include/ptHitable.h:59
}
// BVH::pdfValue (src/ptBVH.cu:82-87): 0.5f * left + 0.5f * right at every node, evaluated in the
// recursion's order with an explicit stack (phase 0: left pending, 1: right pending, 2: combine).
template <uint32_t F>
__device__ inline float light_tree_pdf(const SceneView& sv, int bvh_index, V3 o, V3 v)
{
    const PtBvh b = sv.bvhs[bvh_index];
    if (b.xform_count > 0) return 0;                      // Translate / RotateY do not forward pdfValue
    int node[REFBVH_STACK], phase[REFBVH_STACK];
    float left[REFBVH_STACK];
    int sp = 1;
    node[0] = b.root; phase[0] = 0; left[0] = 0;
    float ret = 0;
    while (sp > 0) {
        const int e = sp - 1;
        const PtBvhNode* n = &sv.bvh_nodes[node[e]];
        if (phase[e] == 0) {
            phase[e] = 1;
            if (n->left >= 0) { if (sp < REFBVH_STACK) { node[sp] = n->left; phase[sp] = 0; left[sp] = 0; sp++; continue; } ret = 0; }
            else ret = light_shape_pdf<F>(sv, ~n->left, o, v);
        }
        if (phase[e] == 1) {
            left[e] = ret; phase[e] = 2;
            if (n->right >= 0) { if (sp < REFBVH_STACK) { node[sp] = n->right; phase[sp] = 0; left[sp] = 0; sp++; continue; } ret = 0; }
            else ret = light_shape_pdf<F>(sv, ~n->right, o, v);
        }
        const float weight = 0.5f;
        ret = weight * left[e] + weight * ret;
        sp--;
    }
    return ret;
}
// BVH::random (src/ptBVH.cu:89-95)
template <uint32_t F>
__device__ inline V3 light_tree_random(const SceneView& sv, int bvh_index, V3 o, Pcg& g)
{
    const PtBvh b = sv.bvhs[bvh_index];
    if (b.xform_count > 0) return mk(1, 0, 0);
    int cur = b.root;
    while (cur >= 0) {
        const PtBvhNode* n = &sv.bvh_nodes[cur];
        cur = (g.rand() < 0.5f) ? n->left : n->right;
    }
    return light_shape_random<F>(sv, ~cur, o, g);
}

template <uint32_t F>
__device__ __forceinline__ float lights_pdf(const SceneView& sv, V3 o, V3 v)
{
    if ((F & F_LIGHT_MISC) && (F & F_REFBVH) && sv.lights_is_list == 2) return light_tree_pdf<F>(sv, sv.lights[0], o, v);
    if (!(F & F_LIGHT_MISC) || !sv.lights_is_list) return light_shape_pdf<F>(sv, sv.lights[0], o, v);
    float weight = 1 / (float)sv.n_lights;                                                 // include/ptHitableList.h:29-38
    float sum = 0;
    for (int i = 0; i < sv.n_lights; i++) sum += weight * light_shape_pdf<F>(sv, sv.lights[i], o, v);
    return sum;
}
template <uint32_t F>
__device__ __forceinline__ V3 lights_random(const SceneView& sv, V3 o, Pcg& g)
{
    if ((F & F_LIGHT_MISC) && (F & F_REFBVH) && sv.lights_is_list == 2) return light_tree_random<F>(sv, sv.lights[0], o, g);
    if (!(F & F_LIGHT_MISC) || !sv.lights_is_list) return light_shape_random<F>(sv, sv.lights[0], o, g);
    int index = (int)(g.rand() * sv.n_lights);                                             // include/ptHitableList.h:40-44
    return light_shape_random<F>(sv, sv.lights[index], o, g);
}

__device__ __forceinline__ float cosine_pdf_value(const Onb& uvw, V3 dir)                 // include/ptPDF.h:46-53
{
    float cosine = dot(unit(dir), uvw.w);
    return (cosine > 0) ? cosine / PI_F : 0.0f;
}

__device__ __forceinline__ float schlick(float cs, float ri)                              // include/ptMaterial.h:31-37
{
    float r0 = (1 - ri) / (1 + ri);
    r0 = r0 * r0;
    return r0 + (1 - r0) * ptm::pow5f_glibc(1 - cs);
}

// ---- one bounce of color() (src/ptMain.cu:101-148) --------------------------------------------------------
// Returns true when the path continues with `ray` replaced; false when it ended (thr holds its
// final value).  thr is the running product accumCol.
template <uint32_t F>
__device__ __forceinline__ bool shade(const SceneView& sv, Ray& ray, const Hit& h, V3& thr, Pcg& g)
{
    const Surface sf = complete_hit<F>(sv, ray, h);
    const PtMaterial m = sv.materials[sf.mat];
    if (m.type == PT_MAT_DIFFUSE_LIGHT) {                          // include/ptMaterial.h:262-272
        V3 emitted = mk(0, 0, 0);
        if (dot(sf.n, ray.d) < 0) emitted = texture_value<F>(sv, m.texture, sf.u, sf.v, sf.p);
        thr = thr * emitted;
        return false;
    }
    if ((F & F_METAL) && m.type == PT_MAT_METAL) {                 // include/ptMaterial.h:134-143
        V3 ud = unit(ray.d);
        V3 reflected = ud - (2 * dot(ud, sf.n)) * sf.n;
        V3 fz = m.param * random_in_unit_sphere(g);
        thr = thr * mk(m.albedo[0], m.albedo[1], m.albedo[2]);
        ray.o = sf.p; ray.d = reflected + fz; ray.time = 0;
        return true;
    }
    if ((F & F_DIELECTRIC) && m.type == PT_MAT_DIELECTRIC) {       // include/ptMaterial.h:184-224
        const float ref_idx = m.param;
        const V3 d = ray.d;
        V3 reflected = d - (2 * dot(d, sf.n)) * sf.n;
        V3 outward; float ni_over_nt, cosine;
        const float ddn = dot(d, sf.n);
        if (ddn > 0.0f) { outward = -sf.n; ni_over_nt = ref_idx; cosine = ref_idx * ddn / len(d); }
        else { outward = sf.n; ni_over_nt = 1.0f / ref_idx; cosine = -ddn / len(d); }
        V3 refracted = mk(0, 0, 0);
        float reflect_prob = 1.0f;
        {                                                          // refract(), include/ptVector3.h:204-215
            V3 uv = unit(d);
            float dt = dot(uv, outward);
            float disc = 1 - ni_over_nt * ni_over_nt * (1 - dt * dt);
            if (disc > 0) {
                refracted = ni_over_nt * (uv - outward * dt) - outward * sqrtf(disc);
                reflect_prob = schlick(cosine, ref_idx);
            }
        }
        thr = thr * mk(1, 1, 1);
        ray.o = sf.p; ray.time = 0;
        ray.d = (g.rand() < reflect_prob) ? reflected : refracted;
        return true;
    }
    // Lambertian / Isotropic: non-specular (include/ptMaterial.h:71-85,317-330)
    const bool cosine_pdf = !(F & F_ISOTROPIC) || m.type == PT_MAT_LAMBERTIAN;
    V3 atten = texture_value<F>(sv, m.texture, sf.u, sf.v, sf.p);
    Onb uvw = onb_from_w(sf.n);
    V3 dir; float pdf_value;
    if ((F & F_LIGHTS) && sv.n_lights > 0) {
        if (!(F & F_LIGHT_MISC)) {
            // One bare XZ rectangle as the light: either side of the mixture draws two numbers, so
            // they are drawn before the choice is acted on (one copy of the generator's code for the
            // wave instead of one per side); z then x for the rectangle, r1 then r2 for the cosine lobe.
            const bool pick_light = g.rand() < 0.5f;                // MixturePdf::generate, include/ptPDF.h:94-100
            const float d1 = g.rand();
            const float d2 = g.rand();
            if (pick_light) {                                       // include/ptRectangle.h:91-95
                const PtShape* ls = &sv.shapes[sv.lights[0]];
                V3 rp = mk(ls->p[0] + d2 * (ls->p[1] - ls->p[0]), ls->p[4], ls->p[2] + d1 * (ls->p[3] - ls->p[2]));
                dir = rp - sf.p;
            } else {
                dir = onb_local(uvw, random_cosine_direction(d1, d2));
            }
        } else if (g.rand() < 0.5f) dir = lights_random<F>(sv, sf.p, g);
        else dir = onb_local(uvw, random_cosine_direction(g));
        pdf_value = 0.5f * lights_pdf<F>(sv, sf.p, dir) + 0.5f * cosine_pdf_value(uvw, dir);
    } else {
        if (cosine_pdf) { dir = onb_local(uvw, random_cosine_direction(g)); pdf_value = cosine_pdf_value(uvw, dir); }
        else { dir = random_to_unit_sphere(1, 1, g); pdf_value = 1; }   // ConstPdf, include/ptPDF.h:28-39
    }
    float spdf;
    if (cosine_pdf) {
        float cosine = dot(sf.n, unit(dir));
        spdf = (cosine < 0) ? 0.0f : cosine / PI_F;
    } else {
        spdf = 1.0f / (4.0f * PI_F);
    }
    const V3 emitted = mk(0, 0, 0);
    thr = thr * (emitted + (atten * spdf) / pdf_value);
    ray.o = sf.p; ray.d = dir;           // time is kept (src/ptMain.cu:123,130)
    return true;
}

// shade() for a signature world whose surfaces are tabulated (SceneView::sig_faces): Lambertian and
// DiffuseLight materials over constant textures, one bare XZ rectangle as the light shape.  The
// same arithmetic as shade() from the point where normal, basis and colour are known.
template <uint32_t F, class W>
__device__ __forceinline__ bool shade_sig(const SceneView& sv, const float4* faces, Ray& ray, const Hit& h, V3& thr, Pcg& g)
{
    const float4* row = faces + (SIG_FACE / 4) * h.sig_code;       // the workgroup's copy of SceneView::sig_faces
    const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
    const V3 n = mk(r0.x, r0.y, r0.z);
    Onb uvw;
    uvw.u = mk(r0.w, r1.x, r1.y); uvw.v = mk(r1.z, r1.w, r2.x); uvw.w = mk(r2.y, r2.z, r2.w);
    const V3 colour = mk(r3.x, r3.y, r3.z);
    const int material_type = __float_as_int(r3.w);
    // the hit point: ray.pointAt(t) in the shape's own frame, carried back out (complete_hit)
    V3 p;
    const float* q = sv.sig + SIG_SLOT * (h.sig_code >> 3);
    if (__float_as_int(q[SIG_SLOT - 2]) != 0) {                   // Translate(RotateY(Box)): to_local, pointAt, to_world
        const float sinT = q[9], cosT = q[10];
        V3 o = ray.o - mk(q[6], q[7], q[8]);
        const float ox = cosT * o.x - sinT * o.z, oz = sinT * o.x + cosT * o.z;
        const float dx = cosT * ray.d.x - sinT * ray.d.z, dz = sinT * ray.d.x + cosT * ray.d.z;
        const V3 pl = mk(ox, o.y, oz) + h.t * mk(dx, ray.d.y, dz);
        p = mk(cosT * pl.x + sinT * pl.z, pl.y, -sinT * pl.x + cosT * pl.z) + mk(q[6], q[7], q[8]);
    } else {
        p = at(ray, h.t);
    }
    if (material_type == PT_MAT_DIFFUSE_LIGHT) {                    // include/ptMaterial.h:262-272
        V3 emitted = mk(0, 0, 0);
        if (dot(n, ray.d) < 0) emitted = colour;
        thr = thr * emitted;
        return false;
    }
    V3 dir;
    const bool pick_light = g.rand() < 0.5f;                        // MixturePdf::generate, include/ptPDF.h:94-100 (see shade)
    const float d1 = g.rand();
    const float d2 = g.rand();
    const float* lp = sv.sig_light;                                 // the light shape's constants, held in the view itself
    if (pick_light) {                                               // include/ptRectangle.h:91-95
        V3 rp = mk(lp[0] + d2 * (lp[1] - lp[0]), lp[4], lp[2] + d1 * (lp[3] - lp[2]));
        dir = rp - p;
    } else {
        dir = onb_local(uvw, random_cosine_direction(d1, d2));
    }
    // The rest is shade()'s arithmetic: mixture pdf value, scattering pdf, attenuation.  A direction
    // whose components all lie in [2^-30, 2^20] in magnitude (anything but an axis-parallel or a
    // runaway one) and a hit point within 2^58 put every denominator below into the range where the
    // shortened division is the whole division (rcp_refined): the length, the direction's own y, pi.
    // The numerators are the components again, and cosines that are either +0 (a product sum that
    // cancels) or at least 2^-95 in magnitude, because the table's normals have components that are
    // 0 or at least 2^-20 (checked when the table is built).  The length is then computed once for the
    // light's pdf and for both cosines, its argument inside sqrt_tame's range.
    const float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    const bool tame = fminf(fminf(ax, ay), az) >= RCP_D_MIN && fmaxf(fmaxf(ax, ay), az) <= SHADE_DIR_MAX && fabsf(p.y) <= RCP_BIG;
    float pdf_value, spdf;
    if (tame) {
        const Rcp over_pi = rcp_refined(PI_F);
        const float len2 = sqlen(dir);
        const Rcp over_len = rcp_refined(sqrt_tame(len2));
        const V3 u = mk(quot(dir.x, over_len), quot(dir.y, over_len), quot(dir.z, over_len));
        float t, light_pdf = 0;
        if (rect_ok(p.y, rcp_refined(dir.y), p.x, dir.x, p.z, dir.z, lp[0], lp[1], lp[2], lp[3], lp[4], 0.001f, FLT_MAX, t)) {
            const float area = (lp[1] - lp[0]) * (lp[3] - lp[2]);     // include/ptRectangle.h:77-89
            const float dist_sqrd = t * t * len2;
            const float cosine = fabsf(u.y);                        // dot(v, (0,1,0)) is v.y itself for a finite v with v.y != 0
            light_pdf = dist_sqrd / (cosine * area);
        }
        const float cos_w = dot(u, uvw.w), cos_n = dot(n, u);
        const float cosine_pdf = (cos_w > 0) ? quot(cos_w, over_pi) : 0.0f;       // include/ptPDF.h:46-53
        pdf_value = 0.5f * light_pdf + 0.5f * cosine_pdf;
        spdf = (cos_n < 0) ? 0.0f : quot(cos_n, over_pi);
    } else {
        pdf_value = 0.5f * lights_pdf<F>(sv, p, dir) + 0.5f * cosine_pdf_value(uvw, dir);
        const float cos_n = dot(n, unit(dir));
        spdf = (cos_n < 0) ? 0.0f : cos_n / PI_F;
    }
    const V3 emitted = mk(0, 0, 0);
    const V3 num = colour * spdf;
    // The last division: the pdf value must be in the denominator's range, and a non-zero numerator
    // at least 2^-103 -- colours are 0 or at least 2^-20 (checked with the table), so spdf >= 2^-80 will do.
    V3 ratio;
    if (tame && pdf_value >= RCP_D_MIN && pdf_value <= RCP_BIG && (spdf == 0.0f || spdf >= 0x1p-80f)) {
        const Rcp over_pdf = rcp_refined(pdf_value);
        ratio = mk(quot(num.x, over_pdf), quot(num.y, over_pdf), quot(num.z, over_pdf));
    } else {
        ratio = num / pdf_value;
    }
    thr = thr * (emitted + ratio);
    ray.o = p; ray.d = dir;           // time is kept (src/ptMain.cu:123,130)
    return true;
}

// Miss shader (src/ptMain.cu:143-148, include/ptAmbientLight.h:48-51,86-91)
template <uint32_t F>
__device__ __forceinline__ void shade_miss(const SceneView& sv, const Ray& ray, V3& thr)
{
    if ((F & F_SKY) && sv.ambient_type == PT_AMBIENT_SKY) {
        V3 ud = unit(ray.d);
        float t = 0.5f * (ud.y + 1);
        thr = thr * ((1 - t) * mk(1, 1, 1) + t * mk(0.5f, 0.7f, 1.0f));
    } else if (sv.ambient_type != PT_AMBIENT_NONE) {
        thr = thr * mk(sv.ambient[0], sv.ambient[1], sv.ambient[2]);
    }
}

}  // namespace ptd
#endif
