// The reference's scene wire format <-> PtSceneDesc.
//
// The reference hands its scene to the GPU as one byte stream: main() calls
// world->serialize, lightShapes->serialize (or writeNull), camera->serialize and
// ambientLight->serialize (or writeNull) into a Stream (src/ptMain.cu:560-572), and
// allocate_world_kernel rebuilds the object graph on the device with Hitable::Create,
// Camera::Create and AmbientLight::Create (src/ptMain.cu:423-430).  This file reads that
// stream into the flat arrays the library renders from, and writes a description back out
// in the same format, so a host that already serialises its scene needs no other change.
//
// Format (src/ptStream.cu:56-90): no header, native little-endian, every polymorphic record
// begins with an int32 type id, -1 stands for a null pointer.  Hitable ids are small integers
// (include/ptHitable.h:32-49), material / texture / ambient ids are FourCCs
// (include/ptMaterial.h:39-46, include/ptTexture.h:19-25, include/ptAmbientLight.h:18-22).
// Every leaf carries its own copy of its material; identical records are merged here.
//
// Host code only.
#include <cfloat>
#include <climits>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "pt_host.h"

namespace {

enum : int32_t {
    ID_NULL = -1, ID_LIST = 1, ID_SPHERE, ID_MOVING_SPHERE, ID_XY, ID_XZ, ID_YZ, ID_FLIP, ID_BOX, ID_TRANSLATE,
    ID_ROTATE_Y, ID_MEDIUM, ID_BVH, ID_TRIANGLE, ID_TRIMESH
};
constexpr int32_t fourcc(char a, char b, char c, char d)
{
    return (int32_t)((uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24));
}
constexpr int32_t ID_LAMB = fourcc('L', 'A', 'M', 'B'), ID_METL = fourcc('M', 'E', 'T', 'L'), ID_DIEL = fourcc('D', 'I', 'E', 'L'),
                  ID_DIFF = fourcc('D', 'I', 'F', 'F'), ID_ISOT = fourcc('I', 'S', 'O', 'T');
constexpr int32_t ID_CNST = fourcc('C', 'N', 'S', 'T'), ID_CHEK = fourcc('C', 'H', 'E', 'K'), ID_NOIS = fourcc('N', 'O', 'I', 'S'),
                  ID_IMAG = fourcc('I', 'M', 'A', 'G'), ID_SKYA = fourcc('S', 'K', 'Y', 'A');

struct Fail { std::string msg; };

// ---- reader -------------------------------------------------------------------------------

struct Reader {
    const uint8_t* p;
    size_t n, off = 0;
    pt::DescStorage* st;
    uint32_t flags;
    int depth = 0;
    std::map<std::string, int> texture_index, material_index;   // serialised bytes -> merged record

    void read(void* dst, size_t size)
    {
        if (size > n - off) throw Fail{ "stream ends inside a record (offset " + std::to_string(off) + ")" };
        std::memcpy(dst, p + off, size);
        off += size;
    }
    int32_t i32() { int32_t v; read(&v, 4); return v; }
    float f32() { float v; read(&v, 4); return v; }
    void vec3(float* v) { read(v, 12); }

    // Texture::Create (src/ptTexture.cu) + the deserialize members (include/ptTexture.h:64-72,113-123,160-168,215-228)
    int tex_depth = 0;
    struct Nest {                                             // bounds the recursion a chain of CHEK records drives
        int& d;
        explicit Nest(int& depth) : d(depth) { if (++d > 64) throw Fail{ "checker textures nested deeper than 64 levels" }; }
        ~Nest() { --d; }
    };
    int texture()
    {
        const Nest nest(tex_depth);
        const size_t begin = off;
        const int32_t id = i32();
        if (id == ID_NULL) return -1;
        PtTexture t{};
        t.odd = t.even = -1;
        if (id == ID_CNST) { t.type = PT_TEX_CONSTANT; vec3(t.color); }
        else if (id == ID_CHEK) {
            t.type = PT_TEX_CHECKER;
            t.scale = f32();
            t.odd = texture();
            t.even = texture();
            if (t.odd < 0 || t.even < 0) throw Fail{ "checker texture with a null child" };
        } else if (id == ID_NOIS) { t.type = PT_TEX_NOISE; t.scale = f32(); }
        else if (id == ID_IMAG) {
            // ImageTexture::serialize writes nx*ny bytes of its 3*nx*ny texels (include/ptTexture.h:201-213);
            // the texels the stream does not carry are zero here (the reference reads past its copy).
            t.type = PT_TEX_IMAGE;
            t.nx = i32(); t.ny = i32();
            if (t.nx <= 0 || t.ny <= 0 || (int64_t)t.nx * t.ny > ((int64_t)1 << 28)) throw Fail{ "image texture with a bad size" };
            const size_t carried = (size_t)t.nx * t.ny, full = 3 * carried;
            if (carried > n - off) throw Fail{ "stream ends inside an image texture (offset " + std::to_string(off) + ")" };
            if (st->texels.size() + full > (size_t)INT32_MAX) throw Fail{ "image textures larger than 2 GiB in total" };
            t.texel_offset = (int32_t)st->texels.size();
            st->texels.resize(st->texels.size() + full, 0);
            read(st->texels.data() + t.texel_offset, carried);
            st->textures.push_back(t);                       // images are not merged
            return (int)st->textures.size() - 1;
        } else throw Fail{ "unknown texture id " + std::to_string(id) + " at offset " + std::to_string(begin) };
        std::string key((const char*)p + begin, off - begin);
        if (t.type == PT_TEX_CHECKER) key = "C" + std::to_string(t.odd) + "," + std::to_string(t.even) + "," + std::string((const char*)&t.scale, 4);
        auto it = texture_index.find(key);
        if (it != texture_index.end()) return it->second;
        st->textures.push_back(t);
        return texture_index[key] = (int)st->textures.size() - 1;
    }

    // Material::Create (src/ptMaterial.cu:13-52) + deserialize members (include/ptMaterial.h)
    int material()
    {
        const size_t begin = off;
        const int32_t id = i32();
        if (id == ID_NULL) return -1;
        PtMaterial m{};
        m.texture = -1;
        if (id == ID_LAMB || id == ID_DIFF || id == ID_ISOT) {
            m.type = (id == ID_LAMB) ? PT_MAT_LAMBERTIAN : (id == ID_DIFF) ? PT_MAT_DIFFUSE_LIGHT : PT_MAT_ISOTROPIC;
            m.texture = texture();
            if (m.texture < 0) throw Fail{ "material without a texture (the reference dereferences it unchecked)" };
        } else if (id == ID_METL) { m.type = PT_MAT_METAL; vec3(m.albedo); m.param = f32(); }
        else if (id == ID_DIEL) { m.type = PT_MAT_DIELECTRIC; m.param = f32(); }
        else throw Fail{ "unknown material id " + std::to_string(id) + " at offset " + std::to_string(begin) };
        std::string key = std::to_string(m.type) + ":" + std::to_string(m.texture) + ":" + std::string((const char*)m.albedo, 12) +
                          std::string((const char*)&m.param, 4);
        auto it = material_index.find(key);
        if (it != material_index.end()) return it->second;
        st->materials.push_back(m);
        return material_index[key] = (int)st->materials.size() - 1;
    }

    int push_leaf(PtShape s, const std::vector<PtXform>& chain, bool flip)
    {
        s.flags = flip ? PT_FLIP_NORMAL : 0;
        s.xform_first = (int32_t)st->xforms.size();
        s.xform_count = (int32_t)chain.size();
        st->xforms.insert(st->xforms.end(), chain.begin(), chain.end());
        st->shapes.push_back(s);
        return (int)st->shapes.size() - 1;
    }
    static PtShape blank(int type)
    {
        PtShape s{};
        s.type = type; s.material = -1; s.tri_uv = -1;
        return s;
    }

    // One rectangle record after its id has been consumed (src/ptRectangle.cu:50-63 and siblings).
    PtShape rect_body(int32_t id)
    {
        PtShape s = blank(id == ID_XY ? PT_SHAPE_RECT_XY : id == ID_XZ ? PT_SHAPE_RECT_XZ : PT_SHAPE_RECT_YZ);
        s.material = material();
        read(s.p, 20);
        return s;
    }

    // Is the child list of a Box record the six rectangles Box's constructor builds
    // (include/ptRectangle.h:199-212)?  Parsed from a scratch position; leaves the offset alone on failure.
    bool canonical_box(const float* lo, const float* hi, int* mat_out)
    {
        const size_t save = off;
        const size_t tex_n = st->textures.size(), mat_n = st->materials.size(), texel_n = st->texels.size();
        auto tex_idx = texture_index;
        auto mat_idx = material_index;
        bool ok = true;
        int mat = -2;
        try {
            if (i32() != ID_LIST || i32() != 6) ok = false;
            const int types[6] = { ID_XY, ID_XY, ID_XZ, ID_XZ, ID_YZ, ID_YZ };
            for (int f = 0; ok && f < 6; f++) {
                int32_t id = i32();
                if (f & 1) { if (id != ID_FLIP) { ok = false; break; } id = i32(); }
                if (id != types[f]) { ok = false; break; }
                const PtShape r = rect_body(id);
                const int ax = f >> 1;                         // 0: XY (x,y; k=z), 1: XZ (x,z; k=y), 2: YZ (y,z; k=x)
                const int a = (ax == 2) ? 1 : 0, b = (ax == 0) ? 1 : 2, k = (ax == 0) ? 2 : (ax == 1) ? 1 : 0;
                const float want[5] = { lo[a], hi[a], lo[b], hi[b], (f & 1) ? lo[k] : hi[k] };
                if (std::memcmp(r.p, want, 20) != 0) ok = false;
                if (mat == -2) mat = r.material; else if (mat != r.material) ok = false;
            }
        } catch (const Fail&) { ok = false; }
        if (!ok) {
            off = save;
            st->textures.resize(tex_n); st->materials.resize(mat_n); st->texels.resize(texel_n);
            texture_index = tex_idx; material_index = mat_idx;
            return false;
        }
        *mat_out = mat;
        return true;
    }

    // Hitable::Create (src/ptHitable.cu:17-85).  Leaves reached through `chain` (wrappers, outermost
    // first) with `flip` FlipNormals around them are appended to `out` in list order; nested lists
    // flatten, which keeps HitableList::hit's answer (src/ptHitableList.cu:12-27: the range shrinks
    // across the children of both levels alike and a later child wins an exact tie in both).
    void hitable(std::vector<PtXform>& chain, bool flip, std::vector<PtObject>& out)
    {
        if (++depth > 256) throw Fail{ "hitable graph nested deeper than 256 levels" };
        const size_t begin = off;
        const int32_t id = i32();
        switch (id) {
        case ID_NULL: break;                                  // a null child never hits
        case ID_LIST: {                                       // src/ptHitableList.cu:67-83
            const int32_t count = i32();
            if (count < 0) throw Fail{ "list with a negative count" };
            for (int32_t i = 0; i < count; i++) hitable(chain, flip, out);
            break;
        }
        case ID_SPHERE: {                                     // src/ptSphere.cu:55-64
            PtShape s = blank(PT_SHAPE_SPHERE);
            read(s.p, 16);
            s.material = material();
            out.push_back(PtObject{ PT_OBJ_SHAPE, push_leaf(s, chain, flip) });
            break;
        }
        case ID_MOVING_SPHERE: {                              // src/ptSphere.cu:116-129
            PtShape s = blank(PT_SHAPE_MOVING_SPHERE);
            read(s.p, 36);                                    // center0, center1, time0, time1, radius
            s.material = material();
            out.push_back(PtObject{ PT_OBJ_SHAPE, push_leaf(s, chain, flip) });
            break;
        }
        case ID_XY: case ID_XZ: case ID_YZ:
            out.push_back(PtObject{ PT_OBJ_SHAPE, push_leaf(rect_body(id), chain, flip) });
            break;
        case ID_FLIP:                                         // include/ptRectangle.h:178-186
            hitable(chain, !flip, out);
            break;
        case ID_BOX: {                                        // include/ptRectangle.h:243-253
            float lo[3], hi[3];
            vec3(lo); vec3(hi);
            int mat;
            if (canonical_box(lo, hi, &mat)) {
                PtShape s = blank(PT_SHAPE_BOX);
                std::memcpy(s.p, lo, 12); std::memcpy(s.p + 3, hi, 12);
                s.material = mat;
                out.push_back(PtObject{ PT_OBJ_SHAPE, push_leaf(s, chain, flip) });
            } else {
                hitable(chain, flip, out);                    // any other child: Box::hit only forwards to it
            }
            break;
        }
        case ID_TRANSLATE: {                                  // include/ptRectangle.h:311-320: child first, then offset
            // the offset follows the child in the stream but the wrapper applies before it: parse the
            // child into a scratch list, then patch the placeholder
            chain.push_back(PtXform{ PT_XFORM_TRANSLATE, { 0, 0, 0 } });
            const size_t slot = chain.size() - 1, first_shape = st->shapes.size();
            hitable(chain, flip, out);
            float v[3];
            vec3(v);
            for (size_t s = first_shape; s < st->shapes.size(); s++) {
                PtXform& x = st->xforms[st->shapes[s].xform_first + slot];
                x.v[0] = v[0]; x.v[1] = v[1]; x.v[2] = v[2];
            }
            chain.pop_back();
            break;
        }
        case ID_ROTATE_Y: {                                   // include/ptRectangle.h:422-436
            chain.push_back(PtXform{ PT_XFORM_ROTATE_Y, { 0, 0, 0 } });
            const size_t slot = chain.size() - 1, first_shape = st->shapes.size();
            hitable(chain, flip, out);
            const float sinT = f32(), cosT = f32();
            (void)i32();                                      // hasBox
            float box[6];
            read(box, 24);                                    // bbox: only bounds() reads it
            for (size_t s = first_shape; s < st->shapes.size(); s++) {
                PtXform& x = st->xforms[st->shapes[s].xform_first + slot];
                x.v[0] = sinT; x.v[1] = cosT;
            }
            chain.pop_back();
            break;
        }
        case ID_MEDIUM: {                                     // include/ptMedium.h:80-90
            if (!chain.empty() || flip)
                throw Fail{ "ConstantMedium under Translate/RotateY/FlipNormals is not supported (offset " + std::to_string(begin) + ")" };
            std::vector<PtObject> boundary;
            std::vector<PtXform> none;
            hitable(none, false, boundary);
            PtMedium m{};
            m.density = f32();
            m.material = material();
            if (boundary.size() != 1 || boundary[0].type != PT_OBJ_SHAPE) throw Fail{ "ConstantMedium boundary must be one shape" };
            if (m.material < 0) throw Fail{ "ConstantMedium without a phase function" };
            m.boundary = boundary[0].index;
            st->media.push_back(m);
            out.push_back(PtObject{ PT_OBJ_MEDIUM, (int)st->media.size() - 1 });
            break;
        }
        case ID_TRIANGLE: {                                   // src/ptTriangle.cu:133-148
            PtShape s = blank(PT_SHAPE_TRIANGLE);
            read(s.p, 36);
            PtTriUV uv;
            read(&uv, 24);
            s.material = material();
            float box[6];
            read(box, 24);                                    // recomputed by calcBounds' rule when needed
            s.tri_uv = (int32_t)st->tri_uvs.size();
            st->tri_uvs.push_back(uv);
            out.push_back(PtObject{ PT_OBJ_SHAPE, push_leaf(s, chain, flip) });
            break;
        }
        case ID_TRIMESH: {                                    // src/ptTriangle.cu:247-271; TriangleMesh::hit never hits (:151-154)
            const int32_t count = i32();
            if (count < 0) throw Fail{ "triangle mesh with a negative count" };
            for (int32_t i = 0; i < count; i++) { uint8_t rec[44]; read(rec, sizeof rec); }
            break;
        }
        case ID_BVH:
            throw Fail{ "BVH record: BVH::serialize writes none (src/ptBVH.cu:97-116), the stream is not the reference's" };
        default:
            throw Fail{ "unknown hitable id " + std::to_string(id) + " at offset " + std::to_string(begin) };
        }
        depth--;
    }
};

// ---- writer -------------------------------------------------------------------------------

struct Writer {
    const PtSceneDesc* d;
    uint8_t* buf;
    size_t cap, off = 0;

    void write(const void* src, size_t size)
    {
        if (buf && size <= cap && off <= cap - size) std::memcpy(buf + off, src, size);
        off += size;                                          // keeps counting past the end: the caller learns the size
    }
    void i32(int32_t v) { write(&v, 4); }
    void f32(float v) { write(&v, 4); }

    // The description comes from the caller unvalidated: every index is checked before it is followed.
    void texture(int t)
    {
        if (t < 0) { i32(ID_NULL); return; }
        if (t >= d->n_textures || !d->textures) throw Fail{ "texture index out of range" };
        const PtTexture& x = d->textures[t];
        switch (x.type) {
        case PT_TEX_CONSTANT: i32(ID_CNST); write(x.color, 12); break;
        case PT_TEX_CHECKER:                                  // children precede their checker (as pt_scene_create demands): no cycles
            if (x.odd < 0 || x.even < 0 || x.odd >= t || x.even >= t) throw Fail{ "checker children must precede the checker" };
            i32(ID_CHEK); f32(x.scale); texture(x.odd); texture(x.even); break;
        case PT_TEX_NOISE: i32(ID_NOIS); f32(x.scale); break;
        case PT_TEX_IMAGE:                                    // nx*ny of the 3*nx*ny bytes, as ImageTexture::serialize does
            if (x.nx <= 0 || x.ny <= 0 || x.texel_offset < 0 || !d->texels ||
                (int64_t)x.texel_offset + (int64_t)3 * x.nx * x.ny > (int64_t)d->n_texels) throw Fail{ "image texels out of range" };
            i32(ID_IMAG); i32(x.nx); i32(x.ny); write(d->texels + x.texel_offset, (size_t)x.nx * x.ny); break;
        default: throw Fail{ "unknown texture type" };
        }
    }
    void material(int m)
    {
        if (m < 0) { i32(ID_NULL); return; }
        if (m >= d->n_materials || !d->materials) throw Fail{ "material index out of range" };
        const PtMaterial& x = d->materials[m];
        switch (x.type) {
        case PT_MAT_LAMBERTIAN: i32(ID_LAMB); texture(x.texture); break;
        case PT_MAT_METAL: i32(ID_METL); write(x.albedo, 12); f32(x.param); break;
        case PT_MAT_DIELECTRIC: i32(ID_DIEL); f32(x.param); break;
        case PT_MAT_DIFFUSE_LIGHT: i32(ID_DIFF); texture(x.texture); break;
        case PT_MAT_ISOTROPIC: i32(ID_ISOT); texture(x.texture); break;
        default: throw Fail{ "unknown material type" };
        }
    }
    void rect(int32_t id, int mat, float a0, float a1, float b0, float b1, float k, bool flip)
    {
        if (flip) i32(ID_FLIP);
        i32(id);
        material(mat);
        const float v[5] = { a0, a1, b0, b1, k };
        write(v, 20);
    }
    void bare(const PtShape& s)
    {
        const float* p = s.p;
        switch (s.type) {
        case PT_SHAPE_SPHERE: i32(ID_SPHERE); write(p, 16); material(s.material); break;
        case PT_SHAPE_MOVING_SPHERE: i32(ID_MOVING_SPHERE); write(p, 36); material(s.material); break;
        case PT_SHAPE_RECT_XY: rect(ID_XY, s.material, p[0], p[1], p[2], p[3], p[4], false); break;
        case PT_SHAPE_RECT_XZ: rect(ID_XZ, s.material, p[0], p[1], p[2], p[3], p[4], false); break;
        case PT_SHAPE_RECT_YZ: rect(ID_YZ, s.material, p[0], p[1], p[2], p[3], p[4], false); break;
        case PT_SHAPE_BOX: {                                  // include/ptRectangle.h:199-212,225-241
            i32(ID_BOX); write(p, 24);
            i32(ID_LIST); i32(6);
            rect(ID_XY, s.material, p[0], p[3], p[1], p[4], p[5], false);
            rect(ID_XY, s.material, p[0], p[3], p[1], p[4], p[2], true);
            rect(ID_XZ, s.material, p[0], p[3], p[2], p[5], p[4], false);
            rect(ID_XZ, s.material, p[0], p[3], p[2], p[5], p[1], true);
            rect(ID_YZ, s.material, p[1], p[4], p[2], p[5], p[3], false);
            rect(ID_YZ, s.material, p[1], p[4], p[2], p[5], p[0], true);
            break;
        }
        case PT_SHAPE_TRIANGLE: {                             // src/ptTriangle.cu:114-131
            i32(ID_TRIANGLE); write(p, 36);
            PtTriUV uv{};
            if (s.tri_uv >= d->n_tri_uvs || (s.tri_uv >= 0 && !d->tri_uvs)) throw Fail{ "tri_uv index out of range" };
            if (s.tri_uv >= 0) uv = d->tri_uvs[s.tri_uv];
            write(&uv, 24);
            material(s.material);
            PtShape t = s;
            t.xform_count = 0;
            float lo[3], hi[3];
            pt::shape_bounds(&t, d->xforms, 0, lo, hi);
            write(lo, 12); write(hi, 12);
            break;
        }
        default: throw Fail{ "unknown shape type" };
        }
    }
    // Wrapper k of the shape's chain around the rest (Translate / RotateY::serialize, include/ptRectangle.h:293-309,401-420)
    void wrapped(const PtShape& s, int k)
    {
        if (k == s.xform_count) {
            if (s.flags & PT_FLIP_NORMAL) i32(ID_FLIP);
            bare(s);
            return;
        }
        const PtXform& x = d->xforms[s.xform_first + k];
        if (x.type == PT_XFORM_TRANSLATE) {
            i32(ID_TRANSLATE);
            wrapped(s, k + 1);
            write(x.v, 12);
        } else {
            i32(ID_ROTATE_Y);
            wrapped(s, k + 1);
            f32(x.v[0]); f32(x.v[1]);
            i32(1);                                           // hasBox: every shape here has bounds
            PtShape t = s;
            t.xform_first = s.xform_first + k;
            t.xform_count = s.xform_count - k;
            float lo[3], hi[3];
            pt::shape_bounds(&t, d->xforms, 0, lo, hi);       // the rotated box RotateY's constructor stores (:340-365)
            write(lo, 12); write(hi, 12);
        }
    }
    void shape(int index)
    {
        if (index < 0 || index >= d->n_shapes || !d->shapes) throw Fail{ "shape index out of range" };
        const PtShape& s = d->shapes[index];
        if (s.xform_count < 0 || s.xform_first < 0 || (int64_t)s.xform_first + s.xform_count > (int64_t)d->n_xforms ||
            (s.xform_count > 0 && !d->xforms)) throw Fail{ "shape wrapper range out of bounds" };
        if (s.xform_count > 64) throw Fail{ "more than 64 wrappers around a shape" };
        wrapped(s, 0);
    }
};

int fail(const std::string& msg, int code)
{
    pt::set_error(msg.c_str());
    return code;
}

}  // namespace

extern "C" int pt_scene_desc_from_stream(const void* data, size_t size, uint32_t flags, PtSceneDesc** out_desc)
{
    if (!data || !out_desc) return fail("pt_scene_desc_from_stream: null argument", PT_ERR_INVALID);
    auto* st = new pt::DescStorage();
    Reader r{ static_cast<const uint8_t*>(data), size, 0, st, flags };
    try {
        std::vector<PtXform> chain;
        r.hitable(chain, false, st->world);                   // *world = Hitable::Create(&stream)
        std::vector<PtObject> lights;                         // *lightShapes = Hitable::Create(&stream)
        const size_t at = r.off;
        int32_t head;
        r.read(&head, 4);
        r.off = at;
        r.hitable(chain, false, lights);
        for (const PtObject& o : lights) {
            if (o.type != PT_OBJ_SHAPE) throw Fail{ "light shapes must be plain shapes" };
            st->lights.push_back(o.index);
        }
        st->desc.lights_is_list = (head == ID_LIST) ? 1 : 0;
        r.read(st->desc.camera, 96);                          // g_cam = Camera::Create(&stream), src/ptCamera.cu:66-84
        const int32_t amb = r.i32();                          // g_ambientLight = AmbientLight::Create(&stream)
        if (amb == ID_NULL) st->desc.ambient_type = PT_AMBIENT_NONE;
        else if (amb == ID_SKYA) st->desc.ambient_type = PT_AMBIENT_SKY;
        else if (amb == ID_CNST) {
            // ConstantAmbient::deserialize calls color.serialize (include/ptAmbientLight.h:65-73): the
            // device-side object keeps its default colour, black, whatever the stream holds.
            st->desc.ambient_type = PT_AMBIENT_CONSTANT;
            float c[3] = { 0, 0, 0 };
            if (r.n - r.off >= 12) r.read(c, 12);
            if (flags & PT_STREAM_KEEP_AMBIENT_COLOR) std::memcpy(st->desc.ambient_color, c, 12);
        } else throw Fail{ "unknown ambient light id " + std::to_string(amb) };
        st->desc.accel = PT_ACCEL_LIST;
        st->finalize();
    } catch (const Fail& f) {
        delete st;
        return fail("pt_scene_desc_from_stream: " + f.msg, (f.msg.find("not supported") != std::string::npos) ? PT_ERR_UNSUPPORTED : PT_ERR_INVALID);
    } catch (const std::bad_alloc&) {                         // nothing may unwind across the C ABI
        delete st;
        return fail("pt_scene_desc_from_stream: out of memory", PT_ERR_OOM);
    } catch (const std::exception& e) {
        delete st;
        return fail(std::string("pt_scene_desc_from_stream: ") + e.what(), PT_ERR_INVALID);
    }
    *out_desc = &st->desc;
    return PT_OK;
}

extern "C" int pt_scene_desc_to_stream(const PtSceneDesc* d, void* buffer, size_t capacity, size_t* out_size, size_t section_sizes[4])
{
    if (!d || !out_size) return fail("pt_scene_desc_to_stream: null argument", PT_ERR_INVALID);
    Writer w{ d, static_cast<uint8_t*>(buffer), buffer ? capacity : 0 };
    size_t mark[5] = { 0, 0, 0, 0, 0 };
    try {
        if (d->n_world < 0 || (d->n_world > 0 && !d->world) || d->n_lights < 0 || (d->n_lights > 0 && !d->lights))
            throw Fail{ "world or light list missing" };
        w.i32(ID_LIST);                                       // world->serialize, src/ptHitableList.cu:52-65
        w.i32(d->n_world);
        for (int i = 0; i < d->n_world; i++) {
            const PtObject& o = d->world[i];
            if (o.type == PT_OBJ_SHAPE) w.shape(o.index);
            else if (o.type == PT_OBJ_MEDIUM) {               // include/ptMedium.h:62-78
                if (o.index < 0 || o.index >= d->n_media || !d->media) throw Fail{ "medium index out of range" };
                const PtMedium& m = d->media[o.index];
                w.i32(ID_MEDIUM);
                w.shape(m.boundary);
                w.f32(m.density);
                w.material(m.material);
            } else {
                // BVH::serialize returns false without writing (src/ptBVH.cu:97-100): the reference
                // itself cannot hand a scene with a BVH to its GPU path
                throw Fail{ "a world with a BVH is not supported by the reference's stream format" };
            }
        }
        mark[1] = w.off;
        if (d->n_lights > 0 && d->lights_is_list == 2) throw Fail{ "light shapes in a BVH are not supported by the reference's stream format" };
        if (d->n_lights <= 0) w.i32(ID_NULL);
        else if (!d->lights_is_list) w.shape(d->lights[0]);
        else {
            w.i32(ID_LIST); w.i32(d->n_lights);
            for (int i = 0; i < d->n_lights; i++) w.shape(d->lights[i]);
        }
        mark[2] = w.off;
        w.write(d->camera, 96);                               // src/ptCamera.cu:47-64
        mark[3] = w.off;
        if (d->ambient_type == PT_AMBIENT_NONE) w.i32(ID_NULL);
        else if (d->ambient_type == PT_AMBIENT_SKY) w.i32(ID_SKYA);
        else { w.i32(ID_CNST); w.write(d->ambient_color, 12); }
        mark[4] = w.off;
    } catch (const Fail& f) {
        return fail("pt_scene_desc_to_stream: " + f.msg, (f.msg.find("not supported") != std::string::npos) ? PT_ERR_UNSUPPORTED : PT_ERR_INVALID);
    } catch (const std::exception& e) {                       // nothing may unwind across the C ABI
        return fail(std::string("pt_scene_desc_to_stream: ") + e.what(), PT_ERR_INVALID);
    }
    *out_size = w.off;
    if (section_sizes) for (int i = 0; i < 4; i++) section_sizes[i] = mark[i + 1] - mark[i];
    if (buffer && w.off > capacity) return fail("pt_scene_desc_to_stream: buffer too small", PT_ERR_INVALID);
    return PT_OK;
}
