// Which order does the host compiler evaluate the random draws in?
//
// The reference's scene builders and XZRectangle::random put several rng.rand() calls into one
// constructor-argument list (src/ptMain.cu:257,262,266,410; include/ptRectangle.h:93).  C++ leaves
// the order of evaluation of function arguments unspecified, so the scene the reference builds
// depends on its host compiler.  This program has the SHAPE of those expressions -- the same nesting
// of constructor calls, the same parameter kinds (const reference, by value, pointer), a virtual
// rand() -- over a generator that just counts, and prints which draw lands in which slot.
// It is this build's own program, not reference code; tests/test_oracle_pins.py compiles it with
// g++ (the reference's host compiler under nvcc) and checks that the scene builders and the
// light sampler of this build consume their draws in the order it reports.
#include <cstdio>

struct Rng { virtual float rand() = 0; virtual ~Rng() {} };
struct Counter : Rng { int n = 0; float rand() override { return (float)++n; } };

struct Vec {
    float e[3];
    Vec() {}
    Vec(float a, float b, float c) { e[0] = a; e[1] = b; e[2] = c; }
};
inline Vec operator+(const Vec& a, const Vec& b) { return Vec(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }

struct Texture { virtual ~Texture() {} };
struct ConstantTexture : Texture { Vec c; explicit ConstantTexture(const Vec& v) : c(v) {} };
struct Material { virtual ~Material() {} };
struct Lambertian : Material { Texture* t; explicit Lambertian(Texture* a) : t(a) {} };
struct Metal : Material { Vec a; float f; Metal(const Vec& al, float fz) : a(al), f(fz) {} };
struct Hitable { virtual ~Hitable() {} };
struct Sphere : Hitable { Vec c; float r; Material* m; Sphere(const Vec& cen, float rad, Material* mat) : c(cen), r(rad), m(mat) {} };
struct MovingSphere : Hitable {
    Vec c0, c1; float t0, t1, r; Material* m;
    MovingSphere(const Vec& cen0, const Vec& cen1, float a, float b, float rad, Material* mtl) : c0(cen0), c1(cen1), t0(a), t1(b), r(rad), m(mtl) {}
};

int main()
{
    {   // Vector3f center(a+0.9*rng.rand(), 0.2, b+0.9*rng.rand());                      src/ptMain.cu:257
        Counter g; Rng& rng = g;
        int a = 0, b = 0;
        Vec center(a + 0.9 * rng.rand() / 0.9, 0.2, b + 0.9 * rng.rand() / 0.9);
        std::printf("center x=%g z=%g\n", center.e[0], center.e[2]);
    }
    {   // new MovingSphere(center, center+Vector3f(0, 0.5*rng.rand(),0), 0, 1, 0.2,
        //     new Lambertian(new ConstantTexture(Vector3f(rng.rand()*rng.rand(), rng.rand()*rng.rand(), rng.rand()*rng.rand()))))   :262
        Counter g; Rng& rng = g;
        Vec center(0, 0, 0);
        // products are replaced by 10*first+second so that both operands stay visible
        MovingSphere* s = new MovingSphere(center, center + Vec(0, 0.5 * rng.rand() / 0.5, 0), 0, 1, 0.2,
                                           new Lambertian(new ConstantTexture(Vec(10 * rng.rand() + rng.rand(), 10 * rng.rand() + rng.rand(), 10 * rng.rand() + rng.rand()))));
        const Vec& c = static_cast<ConstantTexture*>(static_cast<Lambertian*>(s->m)->t)->c;
        std::printf("moving_sphere center1.y=%g colour=%g %g %g\n", s->c1.e[1], c.e[0], c.e[1], c.e[2]);
    }
    {   // new Sphere(center, 0.2, new Metal(Vector3f(0.5*(1+rng.rand()), 0.5*(1+rng.rand()), 0.5*rng.rand()), 0.3))     :266
        Counter g; Rng& rng = g;
        Vec center(0, 0, 0);
        Sphere* s = new Sphere(center, 0.2, new Metal(Vec(0.5 * (1 + rng.rand()) * 2 - 1, 0.5 * (1 + rng.rand()) * 2 - 1, 0.5 * rng.rand() * 2), 0.3));
        const Vec& a = static_cast<Metal*>(s->m)->a;
        std::printf("metal albedo=%g %g %g\n", a.e[0], a.e[1], a.e[2]);
    }
    {   // new Sphere(Vector3f(165*rng.rand(), 165*rng.rand(), 165*rng.rand()), 10, white)      :410
        Counter g; Rng& rng = g;
        Sphere* s = new Sphere(Vec(165 * rng.rand() / 165, 165 * rng.rand() / 165, 165 * rng.rand() / 165), 10, nullptr);
        std::printf("cluster_sphere centre=%g %g %g\n", s->c.e[0], s->c.e[1], s->c.e[2]);
    }
    {   // Vector3f randPoint = Vector3f(x0 + rng.rand() * (x1-x0), k, z0 + rng.rand() * (z1-z0));   include/ptRectangle.h:93
        Counter g; Rng& rng = g;
        float x0 = 0, x1 = 1, z0 = 0, z1 = 1, k = 5;
        Vec p = Vec(x0 + rng.rand() * (x1 - x0), k, z0 + rng.rand() * (z1 - z0));
        std::printf("xz_rectangle_random x=%g z=%g\n", p.e[0], p.e[2]);
    }
    return 0;
}
