// Scene description factories: the reference's hard-coded scene builders
// (src/ptMain.cu:192-421) restated as producers of flat PtSceneDesc arrays.
//
// The reference builds a pointer-linked Hitable graph with `new`; here every builder
// appends to tagged arrays.  Random draws that the reference makes inside one
// argument list are unsequenced in C++; g++ (the reference's host compiler) evaluates
// such arguments right to left, and the draws are written here as separate statements
// in that order so every compiler produces the same scene.
//
// Host code only; must be compiled without FMA contraction (-ffp-contract=off).
#include <cfloat>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "pt_host.h"

namespace {

constexpr float PI_F = 3.141592654f;   // CUDART_PI_F

struct Vec { float x, y, z; };
inline Vec operator+(Vec a, Vec b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Vec operator-(Vec a, Vec b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec operator*(float s, Vec a) { return { s * a.x, s * a.y, s * a.z }; }
inline float dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(Vec a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline Vec unit(Vec a) { float l = length(a); return { a.x / l, a.y / l, a.z / l }; }
inline Vec cross(Vec a, Vec b)
{
    return { a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x };
}

// SimpleRng (include/ptRNG.h:30-63): Marsaglia multiply-with-carry pair.
struct SimpleRng {
    unsigned s0, s1;
    SimpleRng(unsigned a, unsigned b) : s0(a), s1(b) {}
    float rand()
    {
        s0 = 36969 * (s0 & 65535) + (s0 >> 16);
        s1 = 18000 * (s1 & 65535) + (s1 >> 16);
        unsigned ires = (s0 << 16) + s1;
        union { float f; unsigned ui; } res;
        res.ui = (ires & 0x007fffff) | 0x40000000;
        return (res.f - 2.0f) / 2.0f;
    }
};

struct Builder {
    pt::DescStorage* st;
    explicit Builder(pt::DescStorage* s) : st(s) {}

    int constant_texture(float r, float g, float b)
    {
        PtTexture t{};
        t.type = PT_TEX_CONSTANT; t.odd = t.even = -1;
        t.color[0] = r; t.color[1] = g; t.color[2] = b;
        st->textures.push_back(t);
        return (int)st->textures.size() - 1;
    }
    // CheckerTexture(t0, t1): odd = t1, even = t0, scaler = 10 (include/ptTexture.h:85-88,128)
    int checker_texture(int t0, int t1)
    {
        PtTexture t{};
        t.type = PT_TEX_CHECKER; t.odd = t1; t.even = t0; t.scale = 10.0f;
        st->textures.push_back(t);
        return (int)st->textures.size() - 1;
    }
    int noise_texture(float scale)
    {
        PtTexture t{};
        t.type = PT_TEX_NOISE; t.odd = t.even = -1; t.scale = scale;
        st->textures.push_back(t);
        return (int)st->textures.size() - 1;
    }
    int image_texture(const uint8_t* rgb, int nx, int ny)
    {
        PtTexture t{};
        t.type = PT_TEX_IMAGE; t.odd = t.even = -1; t.nx = nx; t.ny = ny;
        t.texel_offset = (int)st->texels.size();
        st->texels.insert(st->texels.end(), rgb, rgb + (size_t)3 * nx * ny);
        st->textures.push_back(t);
        return (int)st->textures.size() - 1;
    }
    int material(int type, int tex, float r, float g, float b, float param)
    {
        PtMaterial m{};
        m.type = type; m.texture = tex; m.albedo[0] = r; m.albedo[1] = g; m.albedo[2] = b; m.param = param;
        st->materials.push_back(m);
        return (int)st->materials.size() - 1;
    }
    int lambertian(int tex) { return material(PT_MAT_LAMBERTIAN, tex, 0, 0, 0, 0); }
    // Metal(a, f): fuzz = f < 1 ? f : 1 (include/ptMaterial.h:127-132)
    int metal(float r, float g, float b, float f) { return material(PT_MAT_METAL, -1, r, g, b, (f < 1) ? f : 1.0f); }
    int dielectric(float ri) { return material(PT_MAT_DIELECTRIC, -1, 0, 0, 0, ri); }
    int diffuse_light(int tex) { return material(PT_MAT_DIFFUSE_LIGHT, tex, 0, 0, 0, 0); }
    int isotropic(int tex) { return material(PT_MAT_ISOTROPIC, tex, 0, 0, 0, 0); }

    int push_shape(PtShape s)
    {
        st->shapes.push_back(s);
        return (int)st->shapes.size() - 1;
    }
    static PtShape blank(int type, int mat)
    {
        PtShape s{};
        s.type = type; s.material = mat; s.xform_first = 0; s.xform_count = 0; s.tri_uv = -1;
        return s;
    }
    int sphere(Vec c, float r, int mat)
    {
        PtShape s = blank(PT_SHAPE_SPHERE, mat);
        s.p[0] = c.x; s.p[1] = c.y; s.p[2] = c.z; s.p[3] = r;
        return push_shape(s);
    }
    int moving_sphere(Vec c0, Vec c1, float t0, float t1, float r, int mat)
    {
        PtShape s = blank(PT_SHAPE_MOVING_SPHERE, mat);
        s.p[0] = c0.x; s.p[1] = c0.y; s.p[2] = c0.z; s.p[3] = c1.x; s.p[4] = c1.y; s.p[5] = c1.z;
        s.p[6] = t0; s.p[7] = t1; s.p[8] = r;
        return push_shape(s);
    }
    int rect(int type, float a0, float a1, float b0, float b1, float k, int mat, bool flip = false)
    {
        PtShape s = blank(type, mat);
        s.p[0] = a0; s.p[1] = a1; s.p[2] = b0; s.p[3] = b1; s.p[4] = k;
        if (flip) s.flags |= PT_FLIP_NORMAL;
        return push_shape(s);
    }
    int box(Vec p0, Vec p1, int mat)
    {
        PtShape s = blank(PT_SHAPE_BOX, mat);
        s.p[0] = p0.x; s.p[1] = p0.y; s.p[2] = p0.z; s.p[3] = p1.x; s.p[4] = p1.y; s.p[5] = p1.z;
        return push_shape(s);
    }
    int triangle(Vec v0, Vec v1, Vec v2, int mat)
    {
        PtShape s = blank(PT_SHAPE_TRIANGLE, mat);
        s.p[0] = v0.x; s.p[1] = v0.y; s.p[2] = v0.z; s.p[3] = v1.x; s.p[4] = v1.y; s.p[5] = v1.z;
        s.p[6] = v2.x; s.p[7] = v2.y; s.p[8] = v2.z;
        return push_shape(s);
    }
    // Translate(RotateY(shape, angle), offset): wrappers are stored outermost first.
    void translate_rotate_y(int shape, float angle_deg, Vec offset)
    {
        PtXform t{};
        t.type = PT_XFORM_TRANSLATE; t.v[0] = offset.x; t.v[1] = offset.y; t.v[2] = offset.z;
        PtXform r{};
        float radians = (PI_F / 180) * angle_deg;          // include/ptRectangle.h:337-339
        r.type = PT_XFORM_ROTATE_Y; r.v[0] = sinf(radians); r.v[1] = cosf(radians); r.v[2] = 0;
        st->shapes[shape].xform_first = (int)st->xforms.size();
        st->shapes[shape].xform_count = 2;
        st->xforms.push_back(t);
        st->xforms.push_back(r);
    }
    void world_shape(int shape) { st->world.push_back(PtObject{ PT_OBJ_SHAPE, shape }); }
    void world_medium(int boundary, float density, int tex)
    {
        PtMedium m{};
        m.boundary = boundary; m.density = density; m.material = isotropic(tex);   // include/ptMedium.h:24-29
        st->media.push_back(m);
        st->world.push_back(PtObject{ PT_OBJ_MEDIUM, (int)st->media.size() - 1 });
    }

    // Camera::Camera (src/ptCamera.cu:27-45)
    void camera(Vec from, Vec to, Vec vup, float vfov, float aspect, float aperture, float focal_dist,
                float t0 = 0, float t1 = 1)
    {
        float lens_radius = aperture / 2;
        float theta = vfov * PI_F / 180;
        float halfHeight = tanf(theta / 2);
        float halfWidth = aspect * halfHeight;
        Vec w = unit(from - to);
        Vec u = unit(cross(vup, w));
        Vec v = cross(w, u);
        Vec origin = from;
        Vec llc = origin - (halfWidth * focal_dist) * u - (halfHeight * focal_dist) * v - focal_dist * w;
        Vec horizontal = (2 * halfWidth * focal_dist) * u;
        Vec vertical = (2 * halfHeight * focal_dist) * v;
        float* c = st->desc.camera;
        const Vec all[7] = { origin, llc, horizontal, vertical, u, v, w };
        for (int i = 0; i < 7; i++) { c[3 * i] = all[i].x; c[3 * i + 1] = all[i].y; c[3 * i + 2] = all[i].z; }
        c[21] = t0; c[22] = t1; c[23] = lens_radius;
    }
    void ambient(int type, float r = 0, float g = 0, float b = 0)
    {
        st->desc.ambient_type = type;
        st->desc.ambient_color[0] = r; st->desc.ambient_color[1] = g; st->desc.ambient_color[2] = b;
    }
};

// ---- reference BVH (src/ptBVH.cu:15-43, src/ptQuickSort.cu:19-81) -------------------------

// Lomuto partition on bounds(0,0).min()[axis], pivot = last element, strict '<'.
int partition(const pt::DescStorage* st, std::vector<int>& list, int l, int h, int axis)
{
    int x = list[h];
    int i = l - 1;
    float bmin[3], bmax[3], pmin[3], pmax[3];
    pt::shape_bounds(st->shapes.data(), st->xforms.data(), x, pmin, pmax);
    for (int j = l; j <= h - 1; j++) {
        pt::shape_bounds(st->shapes.data(), st->xforms.data(), list[j], bmin, bmax);
        if (bmin[axis] < pmin[axis]) {
            i++;
            std::swap(list[i], list[j]);
        }
    }
    std::swap(list[i + 1], list[h]);
    return i + 1;
}
void quick_sort(const pt::DescStorage* st, std::vector<int>& list, int l, int h, int axis)
{
    std::vector<int> stack((size_t)(h - l + 1) + 2);
    int top = -1;
    stack[++top] = l;
    stack[++top] = h;
    while (top >= 0) {
        h = stack[top--];
        l = stack[top--];
        int p = partition(st, list, l, h, axis);
        if (p - 1 > l) { stack[++top] = l; stack[++top] = p - 1; }
        if (p + 1 < h) { stack[++top] = p + 1; stack[++top] = h; }
    }
}
void child_bounds(const pt::DescStorage* st, int child, float* bmin, float* bmax)
{
    if (child >= 0) {
        std::memcpy(bmin, st->bvh_nodes[child].bmin, 12);
        std::memcpy(bmax, st->bvh_nodes[child].bmax, 12);
    } else {
        pt::shape_bounds(st->shapes.data(), st->xforms.data(), ~child, bmin, bmax);
    }
}
// Returns the index of the node built over list[begin, begin+n).
int build_ref_bvh(pt::DescStorage* st, std::vector<int>& list, int begin, int n, SimpleRng& rng)
{
    int axis = int(3 * rng.rand());
    if (n > 1) {
        // quickSort(list, 0, length-1, axis) on the sub-array
        std::vector<int> sub(list.begin() + begin, list.begin() + begin + n);
        quick_sort(st, sub, 0, n - 1, axis);
        std::copy(sub.begin(), sub.end(), list.begin() + begin);
    }
    int left, right;
    if (n == 1) {
        left = right = ~list[begin];
    } else if (n == 2) {
        left = ~list[begin];
        right = ~list[begin + 1];
    } else {
        left = build_ref_bvh(st, list, begin, n / 2, rng);
        right = build_ref_bvh(st, list, begin + n / 2, n - n / 2, rng);
    }
    float lmin[3], lmax[3], rmin[3], rmax[3];
    child_bounds(st, left, lmin, lmax);
    child_bounds(st, right, rmin, rmax);
    PtBvhNode node{};
    for (int a = 0; a < 3; a++) {                       // join(), include/ptAABB.h:77-87
        node.bmin[a] = fminf(lmin[a], rmin[a]);
        node.bmax[a] = fmaxf(lmax[a], rmax[a]);
    }
    node.left = left;
    node.right = right;
    st->bvh_nodes.push_back(node);
    return (int)st->bvh_nodes.size() - 1;
}

// ---- the six reference scenes + the synthetic triangle mesh -------------------------------

void simple_spheres(Builder& b, float aspect)           // src/ptMain.cu:192-207
{
    b.world_shape(b.sphere({ 0.0f, 0.0f, -1.0f }, 0.5f, b.lambertian(b.constant_texture((float)0.1, (float)0.2, (float)0.5))));
    b.world_shape(b.sphere({ 0.0f, -100.5f, -1.0f }, 100.0f, b.lambertian(b.constant_texture((float)0.8, (float)0.8, (float)0.0))));
    b.world_shape(b.sphere({ 1, 0, -1 }, (float)0.5, b.metal((float)0.8, (float)0.6, (float)0.2, (float)0.3)));
    b.world_shape(b.sphere({ -1, 0, -1 }, (float)0.5, b.dielectric((float)1.5)));
    b.camera({ -2, 2, 1 }, { 0, 0, -1 }, { 0, 1, 0 }, 90, aspect, 0.0f, 10.0f);
    b.ambient(PT_AMBIENT_SKY);
}

void simple_light(Builder& b, float aspect)             // src/ptMain.cu:209-234
{
    b.camera({ 13, 2, 3 }, { 0, 0, 0 }, { 0, 1, 0 }, 40, aspect, (float)0.0, (float)10.0);
    int noise = b.noise_texture(1.0f);
    b.world_shape(b.sphere({ 0, -1000, 0 }, 1000, b.lambertian(noise)));
    b.world_shape(b.sphere({ 0, 2, 0 }, 2, b.lambertian(noise)));
    b.world_shape(b.sphere({ 0, 7, 0 }, 2, b.diffuse_light(b.constant_texture(4, 4, 4))));
    b.world_shape(b.rect(PT_SHAPE_RECT_XY, 3, 5, 1, 3, -2, b.diffuse_light(b.constant_texture(4, 4, 4))));
    b.st->lights.push_back(b.sphere({ 0, 7, 0 }, 2, -1));
    b.st->lights.push_back(b.rect(PT_SHAPE_RECT_XY, 3, 5, 1, 3, -2, -1));
    b.st->desc.lights_is_list = 1;
    b.ambient(PT_AMBIENT_CONSTANT, 0, 0, 0);
}

void random_scene(Builder& b, float aspect)             // src/ptMain.cu:236-284
{
    b.camera({ 13, 2, 3 }, { 0, 0, 0 }, { 0, 1, 0 }, 20, aspect, (float)0.0, (float)10.0, 0.0f, 1.0f);
    SimpleRng rng(42, 13);
    int checker = b.checker_texture(b.constant_texture((float)0.2, (float)0.3, (float)0.1),
                                    b.constant_texture((float)0.9, (float)0.9, (float)0.9));
    b.world_shape(b.sphere({ 0, -1000, 0 }, 1000, b.lambertian(checker)));
    for (int a = -11; a < 11; a++) {
        for (int bb = -11; bb < 11; bb++) {
            double choose_mat = rng.rand();
            // Vector3f center(a+0.9*rng.rand(), 0.2, b+0.9*rng.rand()): z is drawn first
            double rz = rng.rand();
            double rx = rng.rand();
            Vec center = { (float)(a + 0.9 * rx), (float)0.2, (float)(bb + 0.9 * rz) };
            if ((double)length(center - Vec{ 4, (float)0.2, 0 }) > 0.9) {
                if (choose_mat < 0.8) {
                    // g++ order for the MovingSphere constructor arguments: the second centre's
                    // draw first, then the colour products z, y, x
                    Vec c1 = center + Vec{ 0, (float)(0.5 * rng.rand()), 0 };
                    float cz = rng.rand() * rng.rand();
                    float cy = rng.rand() * rng.rand();
                    float cx = rng.rand() * rng.rand();
                    int mat = b.lambertian(b.constant_texture(cx, cy, cz));
                    b.world_shape(b.moving_sphere(center, c1, 0, 1, (float)0.2, mat));
                } else if (choose_mat < 0.95) {
                    float mz = (float)(0.5 * rng.rand());
                    float my = (float)(0.5 * (1 + rng.rand()));
                    float mx = (float)(0.5 * (1 + rng.rand()));
                    b.world_shape(b.sphere(center, (float)0.2, b.metal(mx, my, mz, (float)0.3)));
                } else {
                    b.world_shape(b.sphere(center, (float)0.2, b.dielectric((float)1.5)));
                }
            }
        }
    }
    b.world_shape(b.sphere({ 0, 1, 0 }, (float)1.0, b.dielectric((float)1.5)));
    b.world_shape(b.sphere({ -4, 1, 0 }, (float)1.0, b.lambertian(b.constant_texture((float)0.4, (float)0.2, (float)0.1))));
    b.world_shape(b.sphere({ 4, 1, 0 }, (float)1.0, b.metal((float)0.7, (float)0.6, (float)0.5, (float)0.0)));
    b.ambient(PT_AMBIENT_SKY);
}

void cornell_walls(Builder& b, int red, int white, int green, int light)
{
    b.world_shape(b.rect(PT_SHAPE_RECT_YZ, 0, 555, 0, 555, 555, green, true));
    b.world_shape(b.rect(PT_SHAPE_RECT_YZ, 0, 555, 0, 555, 0, red));
    b.world_shape(b.rect(PT_SHAPE_RECT_XZ, 213, 343, 227, 332, 554, light, true));
    b.world_shape(b.rect(PT_SHAPE_RECT_XZ, 0, 555, 0, 555, 555, white, true));
    b.world_shape(b.rect(PT_SHAPE_RECT_XZ, 0, 555, 0, 555, 0, white));
    b.world_shape(b.rect(PT_SHAPE_RECT_XY, 0, 555, 0, 555, 555, white, true));
}

void cornell_box(Builder& b, float aspect)              // src/ptMain.cu:286-317
{
    int red = b.lambertian(b.constant_texture(0.65f, 0.05f, 0.05f));
    int white = b.lambertian(b.constant_texture((float)0.73, (float)0.73, (float)0.73));
    int green = b.lambertian(b.constant_texture((float)0.12, (float)0.45, (float)0.15));
    int light = b.diffuse_light(b.constant_texture(15, 15, 15));
    cornell_walls(b, red, white, green, light);
    int b1 = b.box({ 0, 0, 0 }, { 165, 165, 165 }, white);
    b.translate_rotate_y(b1, -18, { 130, 0, 65 });
    b.world_shape(b1);
    int b2 = b.box({ 0, 0, 0 }, { 165, 330, 165 }, white);
    b.translate_rotate_y(b2, 15, { 265, 0, 295 });
    b.world_shape(b2);
    b.camera({ 278, 278, -800 }, { 278, 278, 0 }, { 0, 1, 0 }, 40, aspect, (float)0.0, (float)10.0);
    b.ambient(PT_AMBIENT_CONSTANT, 0, 0, 0);
    b.st->lights.push_back(b.rect(PT_SHAPE_RECT_XZ, 213, 343, 227, 332, 554, -1));
}

void cornell_box_spheres(Builder& b, float aspect)      // src/ptMain.cu:319-348
{
    int red = b.lambertian(b.constant_texture(0.65f, 0.05f, 0.05f));
    int white = b.lambertian(b.constant_texture((float)0.73, (float)0.73, (float)0.73));
    int green = b.lambertian(b.constant_texture((float)0.12, (float)0.45, (float)0.15));
    b.world_shape(b.sphere({ 1e5f + 1.0f, 40.8f, 81.6f }, 1e5f, red));
    b.world_shape(b.sphere({ -1e5f + 99.0f, 40.8f, 81.6f }, 1e5f, red));
    b.world_shape(b.sphere({ 50.0f, 40.8f, 1e5f }, 1e5f, green));
    b.world_shape(b.sphere({ 50.0f, 1e5f, 81.6f }, 1e5f, white));
    b.world_shape(b.sphere({ 50.0f, -1e5f + 81.6f, 81.6f }, 1e5f, white));
    b.world_shape(b.sphere({ 27.0f, 16.5f, 47.0f }, 16.5f, white));
    b.world_shape(b.sphere({ 73.0f, 16.5f, 78.0f }, 16.5f, white));
    b.camera({ 278, 278, -800 }, { 278, 278, 0 }, { 0, 1, 0 }, 40, aspect, (float)0.0, (float)10.0);
    b.ambient(PT_AMBIENT_CONSTANT, 0, 0, 0);
}

void final_scene(Builder& b, float aspect, const uint8_t* earth_rgb, int earth_nx, int earth_ny)
{                                                       // src/ptMain.cu:350-421
    b.camera({ 478, 278, -600 }, { 278, 278, 0 }, { 0, 1, 0 }, 40, aspect, 0.0f, 10.0f);
    const int nb = 20;
    int ground = b.lambertian(b.constant_texture((float)0.48, (float)0.83, (float)0.53));
    std::vector<int> boxes;
    SimpleRng rng(42, 13);
    for (int i = 0; i < nb; i++) {
        for (int j = 0; j < nb; j++) {
            float w = 100;
            float x0 = -1000 + i * w;
            float z0 = -1000 + j * w;
            float y0 = 0;
            float x1 = x0 + w;
            float y1 = 100 * (rng.rand() + 0.01f);
            float z1 = z0 + w;
            boxes.push_back(b.box({ x0, y0, z0 }, { x1, y1, z1 }, ground));
        }
    }
    PtBvh bvh{};
    bvh.root = build_ref_bvh(b.st, boxes, 0, (int)boxes.size(), rng);
    b.st->bvhs.push_back(bvh);
    b.st->world.push_back(PtObject{ PT_OBJ_BVH, 0 });
    int light = b.diffuse_light(b.constant_texture(6, 6, 6));
    b.world_shape(b.rect(PT_SHAPE_RECT_XZ, 123, 423, 147, 412, 554, light, true));
    Vec center = { 400, 400, 200 };
    b.world_shape(b.moving_sphere(center, center + Vec{ 30, 0, 0 }, 0, 1, 50,
                                  b.lambertian(b.constant_texture((float)0.7, (float)0.3, (float)0.1))));
    b.world_shape(b.sphere({ 260, 150, 45 }, 50, b.dielectric((float)1.5)));
    b.world_shape(b.sphere({ 0, 150, 145 }, 50, b.metal((float)0.8, (float)0.8, (float)0.9, 10)));
    int boundary = b.sphere({ 360, 150, 145 }, 70, b.dielectric((float)1.5));
    b.world_shape(boundary);
    b.world_medium(boundary, (float)0.02, b.constant_texture((float)0.2, (float)0.4, (float)0.9));
    boundary = b.sphere({ 0, 0, 0 }, 5000, b.dielectric((float)1.5));
    b.world_medium(boundary, (float)0.0001, b.constant_texture((float)1.0, (float)1.0, (float)1.0));
    if (earth_rgb != nullptr && earth_nx > 0 && earth_ny > 0) {
        int emat = b.lambertian(b.image_texture(earth_rgb, earth_nx, earth_ny));
        b.world_shape(b.sphere({ 400, 200, 400 }, 100, emat));
    }
    b.world_shape(b.sphere({ 220, 280, 300 }, 80, b.lambertian(b.noise_texture((float)0.1))));
    // The 1000-sphere cluster of the reference is built but never added to the world
    // (src/ptMain.cu:405-412), so it is not restated.
    b.st->lights.push_back(b.rect(PT_SHAPE_RECT_XZ, 123, 423, 147, 412, 554, -1));
    b.ambient(PT_AMBIENT_CONSTANT, 0, 0, 0);
}

// PCG-XSH-RR step used as an integer hash for the height field (no libm: exact floats).
inline uint32_t hash_u32(uint64_t v)
{
    uint64_t s = v * 0x5851f42d4c957f2dULL + 0x14057b7ef767814fULL;
    s = s * 0x5851f42d4c957f2dULL + 2025ULL * 2 + 1;
    uint32_t xs = (uint32_t)(((s >> 18u) ^ s) >> 27u);
    uint32_t rot = (uint32_t)(s >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}

// Synthetic tessellated height field inside the Cornell enclosure: (n+1)^2 vertices over
// x,z in [50,505], y = 150 + 80*h with h in [0,1) from an integer hash smoothed over a
// coarse lattice, 2*n*n triangles wound to face +y (Triangle::hit culls back faces,
// src/ptTriangle.cu:45).  No reference scene uses triangles; this is the C4 workload.
void trimesh_scene(Builder& b, float aspect, int n)
{
    int red = b.lambertian(b.constant_texture(0.65f, 0.05f, 0.05f));
    int white = b.lambertian(b.constant_texture((float)0.73, (float)0.73, (float)0.73));
    int green = b.lambertian(b.constant_texture((float)0.12, (float)0.45, (float)0.15));
    int light = b.diffuse_light(b.constant_texture(15, 15, 15));
    cornell_walls(b, red, white, green, light);
    int mesh_mat = b.lambertian(b.constant_texture(0.6f, 0.6f, 0.7f));
    const int nv = n + 1;
    std::vector<float> h((size_t)nv * nv);
    const int coarse = 16;       // bilinear blend of hashed lattice values: gentle hills + fine grain
    for (int iz = 0; iz < nv; iz++) {
        for (int ix = 0; ix < nv; ix++) {
            int cx = ix * coarse / nv, cz = iz * coarse / nv;
            float fx = (float)(ix * coarse % nv) / (float)nv, fz = (float)(iz * coarse % nv) / (float)nv;
            auto lat = [](int x, int z) { return (float)(hash_u32((uint64_t)x * 7919u + (uint64_t)z * 104729u) >> 8) * (1.0f / 16777216.0f); };
            float h00 = lat(cx, cz), h10 = lat(cx + 1, cz), h01 = lat(cx, cz + 1), h11 = lat(cx + 1, cz + 1);
            float hx0 = h00 + fx * (h10 - h00), hx1 = h01 + fx * (h11 - h01);
            float base = hx0 + fz * (hx1 - hx0);
            float grain = (float)(hash_u32(((uint64_t)ix << 32) | (uint64_t)iz) >> 8) * (1.0f / 16777216.0f);
            h[(size_t)iz * nv + ix] = 0.9f * base + 0.1f * grain * (16.0f / (float)n < 1.0f ? 16.0f / (float)n : 1.0f);
        }
    }
    auto vert = [&](int ix, int iz) {
        float x = 50.0f + 455.0f * ((float)ix / (float)n);
        float z = 50.0f + 455.0f * ((float)iz / (float)n);
        float y = 150.0f + 80.0f * h[(size_t)iz * nv + ix];
        return Vec{ x, y, z };
    };
    b.st->shapes.reserve(b.st->shapes.size() + (size_t)2 * n * n + 8);
    b.st->world.reserve(b.st->world.size() + (size_t)2 * n * n);
    for (int iz = 0; iz < n; iz++) {
        for (int ix = 0; ix < n; ix++) {
            Vec v00 = vert(ix, iz), v10 = vert(ix + 1, iz), v01 = vert(ix, iz + 1), v11 = vert(ix + 1, iz + 1);
            b.world_shape(b.triangle(v00, v01, v10, mesh_mat));
            b.world_shape(b.triangle(v10, v01, v11, mesh_mat));
        }
    }
    b.camera({ 278, 278, -800 }, { 278, 278, 0 }, { 0, 1, 0 }, 40, aspect, (float)0.0, (float)10.0);
    b.ambient(PT_AMBIENT_CONSTANT, 0, 0, 0);
    b.st->lights.push_back(b.rect(PT_SHAPE_RECT_XZ, 213, 343, 227, 332, 554, -1));
    b.st->desc.accel = PT_ACCEL_LBVH;
}

}  // namespace

namespace pt {

// Bounding box of a shape including its wrappers; restates the bounds() members
// (include/ptSphere.h:42-46,97-103, include/ptRectangle.h:39-43,71-75,123-127,219-223,
// 283-291,334-366, src/ptTriangle.cu:103-112).
void shape_bounds(const PtShape* shapes, const PtXform* xforms, int index, float* bmin, float* bmax)
{
    const PtShape& s = shapes[index];
    const float* p = s.p;
    const float tol = 0.0001f;   // RECT_TOLERANCE
    switch (s.type) {
    case PT_SHAPE_SPHERE:
        for (int a = 0; a < 3; a++) { bmin[a] = p[a] - p[3]; bmax[a] = p[a] + p[3]; }
        break;
    case PT_SHAPE_MOVING_SPHERE:
        for (int a = 0; a < 3; a++) {
            bmin[a] = fminf(p[a] - p[8], p[3 + a] - p[8]);
            bmax[a] = fmaxf(p[a] + p[8], p[3 + a] + p[8]);
        }
        break;
    case PT_SHAPE_RECT_XY:
        bmin[0] = p[0]; bmin[1] = p[2]; bmin[2] = p[4] - tol; bmax[0] = p[1]; bmax[1] = p[3]; bmax[2] = p[4] + tol;
        break;
    case PT_SHAPE_RECT_XZ:
        bmin[0] = p[0]; bmin[1] = p[4] - tol; bmin[2] = p[2]; bmax[0] = p[1]; bmax[1] = p[4] + tol; bmax[2] = p[3];
        break;
    case PT_SHAPE_RECT_YZ:
        bmin[0] = p[4] - tol; bmin[1] = p[0]; bmin[2] = p[2]; bmax[0] = p[4] + tol; bmax[1] = p[1]; bmax[2] = p[3];
        break;
    case PT_SHAPE_BOX:
        for (int a = 0; a < 3; a++) { bmin[a] = p[a]; bmax[a] = p[3 + a]; }
        break;
    case PT_SHAPE_TRIANGLE:
        for (int a = 0; a < 3; a++) {
            bmin[a] = fminf(p[a] - 0.0001f, fminf(p[3 + a] - 0.0001f, p[6 + a] - 0.0001f));
            bmax[a] = fmaxf(p[a] + 0.0001f, fmaxf(p[3 + a] + 0.0001f, p[6 + a] + 0.0001f));
        }
        break;
    default:
        for (int a = 0; a < 3; a++) { bmin[a] = 0; bmax[a] = 0; }
    }
    // wrappers are stored outermost first; bounds compose innermost first
    for (int k = s.xform_count - 1; k >= 0; k--) {
        const PtXform& x = xforms[s.xform_first + k];
        if (x.type == PT_XFORM_TRANSLATE) {
            for (int a = 0; a < 3; a++) { bmin[a] = bmin[a] + x.v[a]; bmax[a] = bmax[a] + x.v[a]; }
        } else {
            const float sinT = x.v[0], cosT = x.v[1];
            float nmin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, nmax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    for (int k2 = 0; k2 < 2; k2++) {
                        float xx = i * bmax[0] + (1 - i) * bmin[0];
                        float yy = j * bmax[1] + (1 - j) * bmin[1];
                        float zz = k2 * bmax[2] + (1 - k2) * bmin[2];
                        float newx = cosT * xx + sinT * zz;
                        float newz = -sinT * xx + cosT * zz;
                        const float tester[3] = { newx, yy, newz };
                        for (int c = 0; c < 3; c++) {
                            if (tester[c] > nmax[c]) nmax[c] = tester[c];
                            if (tester[c] < nmin[c]) nmin[c] = tester[c];
                        }
                    }
            for (int a = 0; a < 3; a++) { bmin[a] = nmin[a]; bmax[a] = nmax[a]; }
        }
    }
}

void DescStorage::finalize()
{
    desc.n_shapes = (int)shapes.size();       desc.shapes = shapes.data();
    desc.n_tri_uvs = (int)tri_uvs.size();     desc.tri_uvs = tri_uvs.data();
    desc.n_xforms = (int)xforms.size();       desc.xforms = xforms.data();
    desc.n_materials = (int)materials.size(); desc.materials = materials.data();
    desc.n_textures = (int)textures.size();   desc.textures = textures.data();
    desc.n_texels = (int)texels.size();       desc.texels = texels.data();
    desc.n_media = (int)media.size();         desc.media = media.data();
    desc.n_bvh_nodes = (int)bvh_nodes.size(); desc.bvh_nodes = bvh_nodes.data();
    desc.n_bvhs = (int)bvhs.size();           desc.bvhs = bvhs.data();
    desc.n_world = (int)world.size();         desc.world = world.data();
    desc.n_lights = (int)lights.size();       desc.lights = lights.data();
}

}  // namespace pt

extern "C" int pt_scene_desc_trimesh(int32_t grid_n, float aspect, PtSceneDesc** out_desc)
{
    if (!out_desc || grid_n < 1 || grid_n > 4096) { pt::set_error("pt_scene_desc_trimesh: bad arguments"); return PT_ERR_INVALID; }
    auto* st = new pt::DescStorage();
    Builder b(st);
    trimesh_scene(b, aspect, grid_n);
    st->finalize();
    *out_desc = &st->desc;
    return PT_OK;
}

extern "C" int pt_scene_desc_builtin(const char* name, float aspect, const uint8_t* earth_rgb,
                                     int32_t earth_nx, int32_t earth_ny, PtSceneDesc** out_desc)
{
    if (!name || !out_desc) { pt::set_error("pt_scene_desc_builtin: null argument"); return PT_ERR_INVALID; }
    const std::string n(name);
    if (n == "trimesh") return pt_scene_desc_trimesh(1000, aspect, out_desc);
    auto* st = new pt::DescStorage();
    Builder b(st);
    if (n == "simple_spheres") simple_spheres(b, aspect);
    else if (n == "simple_light") simple_light(b, aspect);
    else if (n == "random") random_scene(b, aspect);
    else if (n == "cornell") cornell_box(b, aspect);
    else if (n == "cornell_spheres") cornell_box_spheres(b, aspect);
    else if (n == "final") final_scene(b, aspect, earth_rgb, earth_nx, earth_ny);
    else {
        delete st;
        pt::set_error(("pt_scene_desc_builtin: unknown scene '" + n + "'").c_str());
        return PT_ERR_INVALID;
    }
    st->finalize();
    *out_desc = &st->desc;
    return PT_OK;
}

extern "C" void pt_scene_desc_free(PtSceneDesc* desc)
{
    if (!desc) return;
    // desc is the first member of its DescStorage
    delete reinterpret_cast<pt::DescStorage*>(desc);
}
