// Radiance kernels for gfx950 and the C-ABI entry points that launch them (include/ptcore.h).
//
// The reference's device side is one megakernel, one thread per pixel, all samples and bounces
// in nested loops behind virtual calls (render_kernel, src/ptMain.cu:171-190).  Here:
//   * k_render_tiles / k_render_refill: one lane per pixel, one loop iteration per BOUNCE; a lane whose path ended
//     accumulates it and regenerates the next camera sample inside the same loop, so the 64 lanes
//     of a wave stay busy although their paths have different lengths.  The pixel's PCG stream is
//     carried in registers across samples and bounces exactly as render_pixel does
//     (src/ptMain.cu:153-169), which is what makes images comparable sample for sample.
//   * k_trace_closest: world->hit for a batch of rays (closest-hit parity entry point).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <algorithm>

#include "pt_kernels.h"
#include "pt_host.h"
#include "pt_lbvh.h"

using namespace ptd;
using namespace ptk;

// the specialisations are compiled in pt_k_*.hip
PT_KERNEL_VARIANTS(extern template struct)

// ---- feature-mask specialisations ---------------------------------------------------------------------

namespace {

// Specialisations in the order they are tried: a scene runs on the first whose mask covers it.
const struct { int variant; uint32_t mask; } k_variants[] = {
    { V_CORNELL, M_CORNELL }, { V_SPHERES, M_SPHERES }, { V_SPHERES_LBVH, M_SPHERES_LBVH }, { V_TRIMESH, M_TRIMESH },
    { V_FINAL, M_FINAL }, { V_FINAL_LBVH, M_FINAL_LBVH }, { V_ALL, M_ALL }, { V_ALL_LBVH, M_ALL_LBVH },
    { V_SPHERES_LBVH4, M_SPHERES_LBVH4 }, { V_TRIMESH4, M_TRIMESH4 }, { V_FINAL_LBVH4, M_FINAL_LBVH4 }, { V_ALL_LBVH4, M_ALL_LBVH4 } };

int kernel_family(int variant)
{
    switch (variant) {
    case V_CORNELL: case V_CORNELL_WORLD: case V_CORNELL_WORLD_FLAT: return PT_KERNELS_RECTANGLES;
    case V_SPHERES: case V_SPHERES_LBVH: case V_SPHERES_LBVH4: return PT_KERNELS_SPHERES;
    case V_TRIMESH: case V_TRIMESH4: return PT_KERNELS_TRIANGLES;
    case V_FINAL: case V_FINAL_LBVH: case V_FINAL_LBVH4: return PT_KERNELS_FINAL;
    default: return PT_KERNELS_ALL;
    }
}

int pick_variant(uint32_t scene_mask)
{
    for (const auto& v : k_variants)
        if ((scene_mask & ~v.mask) == 0 && ((v.mask ^ scene_mask) & (F_LBVH | F_LBVH4)) == 0) return v.variant;
    return -1;
}

// One switch for every launcher: LAUNCH(F, W) is expanded for the variant's feature mask and world walk.
#define PT_FOR_VARIANT(variant, LAUNCH)                                \
    switch (variant) {                                                 \
    case V_CORNELL: LAUNCH(M_CORNELL, AnyWorld); break;                \
    case V_SPHERES: LAUNCH(M_SPHERES, AnyWorld); break;                \
    case V_SPHERES_LBVH: LAUNCH(M_SPHERES_LBVH, AnyWorld); break;      \
    case V_TRIMESH: LAUNCH(M_TRIMESH, AnyWorld); break;                \
    case V_ALL: LAUNCH(M_ALL, AnyWorld); break;                        \
    case V_ALL_LBVH: LAUNCH(M_ALL_LBVH, AnyWorld); break;              \
    case V_CORNELL_WORLD: LAUNCH(M_CORNELL, CornellWorld); break;      \
    case V_CORNELL_WORLD_FLAT: LAUNCH(M_CORNELL, Branchless<CornellWorld>); break; \
    case V_FINAL: LAUNCH(M_FINAL, AnyWorld); break;                    \
    case V_FINAL_LBVH: LAUNCH(M_FINAL_LBVH, AnyWorld); break;          \
    case V_SPHERES_LBVH4: LAUNCH(M_SPHERES_LBVH4, AnyWorld); break;    \
    case V_TRIMESH4: LAUNCH(M_TRIMESH4, AnyWorld); break;              \
    case V_FINAL_LBVH4: LAUNCH(M_FINAL_LBVH4, AnyWorld); break;        \
    default: LAUNCH(M_ALL_LBVH4, AnyWorld); break;                     \
    }

void launch_render(int variant, const SceneView& sv, const RenderArgs& ra, float* out, unsigned* next_slot, unsigned long long* chunk_rec, int* fault,
                   hipStream_t st)
{
#define PT_LAUNCH(F, W) FusedLaunch<F, W>::render(sv, ra, out, next_slot, chunk_rec, fault, st)
    PT_FOR_VARIANT(variant, PT_LAUNCH)
#undef PT_LAUNCH
}

void launch_chunks(int variant, const SceneView& sv, const RenderArgs& ra, float* out, unsigned* words, unsigned long long* tile_states, int* fault,
                   hipStream_t st)
{
#define PT_LAUNCH(F, W) FusedLaunch<F, W>::chunks(sv, ra, out, words, tile_states, fault, st)
    PT_FOR_VARIANT(variant, PT_LAUNCH)
#undef PT_LAUNCH
}

void launch_pass(int variant, const SceneView& sv, const RenderArgs& ra, PtAccumPixel* acc, int first, float* out, int* bad_resume,
                 hipStream_t st)
{
#define PT_LAUNCH(F, W) FusedLaunch<F, W>::pass(sv, ra, acc, first, out, bad_resume, st)
    PT_FOR_VARIANT(variant, PT_LAUNCH)
#undef PT_LAUNCH
}

void launch_trace(int variant, const SceneView& sv, const PtRay* rays, size_t n, int32_t* ids, float* ts, hipStream_t st)
{
#define PT_LAUNCH(F, W) FusedLaunch<F, W>::trace(sv, rays, n, ids, ts, st)
    PT_FOR_VARIANT(variant, PT_LAUNCH)
#undef PT_LAUNCH
}

// The entry kinds of a world list as the signature templates name them; 0 = none of them.
// A shape whose constants are not finite or exceed 2^58 in magnitude has no kind: the signature
// walk's shortened division (rcp_refined) relies on that bound for its numerators.
int world_kind(const PtSceneDesc* d, int i)
{
    const PtObject& o = d->world[i];
    if (o.type != PT_OBJ_SHAPE) return 0;
    const PtShape& s = d->shapes[o.index];
    auto tame = [](const float* v, int n) { for (int j = 0; j < n; j++) if (!(std::fabs(v[j]) <= ptd::RCP_BIG)) return false; return true; };
    if (!tame(s.p, (int)(sizeof(s.p) / sizeof(s.p[0])))) return 0;
    for (int x = 0; x < s.xform_count; x++)
        if (!tame(d->xforms[s.xform_first + x].v, (int)(sizeof(d->xforms[0].v) / sizeof(float)))) return 0;
    if (s.xform_count == 0) {
        switch (s.type) {
        case PT_SHAPE_RECT_XY: return WK_RECT_XY;
        case PT_SHAPE_RECT_XZ: return WK_RECT_XZ;
        case PT_SHAPE_RECT_YZ: return WK_RECT_YZ;
        case PT_SHAPE_SPHERE: return WK_SPHERE;
        default: return 0;
        }
    }
    if (s.type == PT_SHAPE_BOX && s.xform_count == 2 && d->xforms[s.xform_first].type == PT_XFORM_TRANSLATE &&
        d->xforms[s.xform_first + 1].type == PT_XFORM_ROTATE_Y) return WK_BOX_TR;
    return 0;
}
template <int... K>
bool world_matches(const PtSceneDesc* d, WorldSig<K...>)
{
    const int kinds[] = { K... };
    if (d->n_world != (int)sizeof...(K)) return false;
    for (int i = 0; i < d->n_world; i++)
        if (world_kind(d, i) != kinds[i]) return false;
    return true;
}

// Which part of a world list with reference BVHs and media the library's LBVH can stand for, and
// the order its leaves are tested in by the in-order walk.  The leading run of shapes and
// un-wrapped BVHs is taken; a BVH contributes its leaves left to right (the reference's recursion
// visits them in that order and keeps the right-most of equally near ones, src/ptBVH.cu:45-74);
// only a BVH at the head of the list is taken, because the walk checks the reference's node boxes
// against the search bound the tree would be entered with, and that is only known there.
// A BVH whose leaves mix the two tie kinds (spheres accept t < tmax, everything else t <= tmax)
// is not taken apart: with mixed kinds a three-way tie between two of its leaves and an earlier
// entry cannot be settled by comparing candidates two at a time.  Neither is a list that names a
// shape twice, whose place in the order would be ambiguous, nor a tree whose boxes do not nest.
struct LbvhPlan {
    std::vector<PtObject> leaves;
    std::vector<int32_t> order, group;      // per shape
    std::vector<float> leaf_gate;           // per shape: box of the reference BVH node naming it (6 floats)
    int prefix = 0;
};
bool lbvh_plan(const PtSceneDesc* d, LbvhPlan* plan)
{
    plan->order.assign((size_t)d->n_shapes, -1);
    plan->group.assign((size_t)d->n_shapes, -1);
    plan->leaf_gate.assign((size_t)d->n_shapes * 6, 0.0f);
    auto add = [&](int shape, int group) {
        if (!plan->leaves.empty() && plan->leaves.back().index == shape && plan->group[shape] == group) return true;   // a one-leaf node names its leaf twice
        if (plan->order[shape] >= 0) return false;
        plan->order[shape] = (int32_t)plan->leaves.size();
        plan->group[shape] = group;
        PtObject o; o.type = PT_OBJ_SHAPE; o.index = shape;
        plan->leaves.push_back(o);
        return true;
    };
    int i = 0;
    for (; i < d->n_world; i++) {
        const PtObject& ob = d->world[i];
        if (ob.type == PT_OBJ_SHAPE) { if (!add(ob.index, -1)) return false; continue; }
        // only a BVH the list starts with: its search bound is then the caller's (see world_hit)
        if (ob.type != PT_OBJ_BVH || i != 0 || d->bvhs[ob.index].xform_count != 0) break;
        std::vector<int> stack;                           // links; in-order = left subtree, then right
        stack.push_back(d->bvhs[ob.index].root);
        int kind = -1;
        size_t guard = 0;
        while (!stack.empty()) {
            const int link = stack.back(); stack.pop_back();
            if (++guard > (size_t)4 * (size_t)(d->n_bvh_nodes + 1)) return false;      // not a tree
            if (link < 0) {
                const int shape = ~link;
                const int k = (d->shapes[shape].type == PT_SHAPE_SPHERE || d->shapes[shape].type == PT_SHAPE_MOVING_SPHERE) ? 0 : 1;
                if (kind >= 0 && k != kind) return false;
                kind = k;
                if (!add(shape, ob.index)) return false;
            } else {
                const PtBvhNode& nd = d->bvh_nodes[link];
                for (int child : { nd.left, nd.right }) {
                    if (child >= 0) {
                        // the walk lets one box speak for all the boxes above it (world_hit): they must nest
                        const PtBvhNode& ch = d->bvh_nodes[child];
                        for (int k = 0; k < 3; k++)
                            if (!(nd.bmin[k] <= ch.bmin[k] && ch.bmax[k] <= nd.bmax[k])) return false;
                        continue;
                    }
                    for (int k = 0; k < 3; k++) { plan->leaf_gate[(size_t)~child * 6 + k] = nd.bmin[k]; plan->leaf_gate[(size_t)~child * 6 + 3 + k] = nd.bmax[k]; }
                }
                stack.push_back(nd.right);
                stack.push_back(nd.left);
            }
        }
    }
    plan->prefix = i;
    return plan->leaves.size() >= 2;
}

// The signature walk's constants, one slot per world entry (layout: world_entry in pt_device.h).
std::vector<float> signature_table(const PtSceneDesc* d)
{
    std::vector<float> t((size_t)d->n_world * ptd::SIG_SLOT, 0.0f);
    for (int i = 0; i < d->n_world; i++) {
        const int id = d->world[i].index;
        const PtShape& s = d->shapes[id];
        float* q = &t[(size_t)i * ptd::SIG_SLOT];
        for (int j = 0; j < 6; j++) q[j] = s.p[j];
        if (world_kind(d, i) == WK_BOX_TR) {
            const PtXform& tr = d->xforms[s.xform_first];
            const PtXform& ro = d->xforms[s.xform_first + 1];
            q[6] = tr.v[0]; q[7] = tr.v[1]; q[8] = tr.v[2];
            q[9] = ro.v[0]; q[10] = ro.v[1];
        }
        const int32_t wrapped = world_kind(d, i) == WK_BOX_TR ? 1 : 0;
        std::memcpy(&q[ptd::SIG_SLOT - 2], &wrapped, sizeof wrapped);
        std::memcpy(&q[ptd::SIG_SLOT - 1], &id, sizeof id);
    }
    return t;
}

// The finished surfaces of a signature world, one row per (entry, face) (SigFace in pt_device.h): filled
// with the kernels' own host-compilable arithmetic.  Only for worlds whose materials are Lambertian or
// DiffuseLight over constant textures (the rectangle-scene mask admits nothing else).
//
// shade_sig divides in the shortened form (rcp_refined), which wants its numerators either zero or
// not too small: a table is only made when every normal component is 0 or at least 2^-20 in
// magnitude (a box turned by less than a millionth of a radian fails), every colour component 0
// or within [2^-20, 2^20], and the light's rectangle within the bound of the world's own constants.
bool signature_faces(const PtSceneDesc* d, std::vector<float>* out)
{
    auto zero_or_sizable = [](float v, float hi) { const float a = std::fabs(v); return v == 0.0f || (a >= 0x1p-20f && a <= hi); };
    if (d->n_lights != 1 || d->lights[0] < 0 || d->lights[0] >= d->n_shapes) return false;
    for (float v : d->shapes[d->lights[0]].p) if (!(std::fabs(v) <= ptd::RCP_BIG)) return false;
    out->assign((size_t)d->n_world * 8 * ptd::SIG_FACE, 0.0f);
    for (int i = 0; i < d->n_world; i++) {
        const PtShape& s = d->shapes[d->world[i].index];
        if (s.material < 0) return false;
        const PtMaterial& m = d->materials[s.material];
        if (m.type != PT_MAT_LAMBERTIAN && m.type != PT_MAT_DIFFUSE_LIGHT) return false;
        if (m.texture < 0 || d->textures[m.texture].type != PT_TEX_CONSTANT) return false;
        const PtTexture& tx = d->textures[m.texture];
        const bool wrapped = world_kind(d, i) == WK_BOX_TR;
        const float sinT = wrapped ? d->xforms[s.xform_first + 1].v[0] : 0.0f, cosT = wrapped ? d->xforms[s.xform_first + 1].v[1] : 1.0f;
        const int faces = s.type == PT_SHAPE_BOX ? 6 : 1;
        for (int f = 0; f < faces; f++) {
            ptd::SigFace sf = ptd::sig_face_make(s.type, f, (s.flags & PT_FLIP_NORMAL) != 0, wrapped, sinT, cosT);
            float* q = &(*out)[((size_t)i * 8 + f) * ptd::SIG_FACE];
            q[0] = sf.n.x; q[1] = sf.n.y; q[2] = sf.n.z;
            q[3] = sf.uvw.u.x; q[4] = sf.uvw.u.y; q[5] = sf.uvw.u.z;
            q[6] = sf.uvw.v.x; q[7] = sf.uvw.v.y; q[8] = sf.uvw.v.z;
            q[9] = sf.uvw.w.x; q[10] = sf.uvw.w.y; q[11] = sf.uvw.w.z;
            q[12] = tx.color[0]; q[13] = tx.color[1]; q[14] = tx.color[2];
            const int32_t mt = m.type;
            std::memcpy(&q[15], &mt, sizeof mt);
            for (int k = 0; k < 3; k++)
                if (!zero_or_sizable(q[k], 2.0f) || !zero_or_sizable(q[9 + k], 2.0f) || !(tx.color[k] >= 0.0f) || !zero_or_sizable(tx.color[k], 0x1p20f)) return false;
        }
    }
    return true;
}

}  // namespace

// ---- host side of the C ABI --------------------------------------------------------------------------

// Scratch of pt_render_device_noalloc: the pixel-slot counters of the persistent-lane schedules (eight words, one per XCD) in
// device memory that exists from the moment the library's code object is loaded, so the call needs
// no hipMalloc.  One copy per GPU; every such call on a GPU uses that GPU's copy.  In host code the
// array's name is only a handle for hipGetSymbolAddress: g_scratch_ptr[] holds the device addresses.
constexpr size_t PT_SCRATCH_BYTES = 256;
static __device__ __align__(256) unsigned char scratch[PT_SCRATCH_BYTES];
constexpr int PT_MAX_DEVICES = 64;
static unsigned* g_scratch_ptr[PT_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

namespace {

bool hip_ok(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    std::string msg = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    pt::set_error(msg.c_str());
    (void)hipGetLastError();      // the runtime keeps the last error until it is read: do not let it surface in a later call
    return false;
}

// Device address of this GPU's copy of `scratch`; looked up once per device, with that device current.
unsigned* scratch_on_device(int device)
{
    if (device < 0 || device >= PT_MAX_DEVICES) { pt::set_error("device index beyond the library's table of scratch addresses"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        int current = -1;
        if (!hip_ok(hipGetDevice(&current), "hipGetDevice")) return nullptr;
        if (current != device) { pt::set_error("the scene's device must be the current device"); return nullptr; }
        void* ptr = nullptr;
        if (!hip_ok(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)")) return nullptr;
        g_scratch_ptr[device] = static_cast<unsigned*>(ptr);
    }
    return g_scratch_ptr[device];
}

template <class T>
bool upload(PtScene* sc, const T* src, size_t n, const T** dst, const char* what)
{
    *dst = nullptr;
    if (n == 0) return true;
    void* d = nullptr;
    if (!hip_ok(hipMalloc(&d, n * sizeof(T)), what)) return false;
    sc->allocs.push_back(d);
    if (!hip_ok(hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice), what)) return false;
    *dst = static_cast<const T*>(d);
    return true;
}

bool fail(const char* msg) { pt::set_error(msg); return false; }

// Structural validation: every index in range, trees acyclic and shallow enough for the
// fixed traversal stacks.  A kernel must never see an index it could fault on.
bool validate(const PtSceneDesc* d, uint32_t* mask_out)
{
    uint32_t mask = 0;
    if (d->n_shapes < 0 || d->n_world <= 0 || d->n_materials < 0 || d->n_textures < 0 || d->n_tri_uvs < 0 || d->n_xforms < 0 ||
        d->n_texels < 0 || d->n_media < 0 || d->n_bvh_nodes < 0 || d->n_bvhs < 0 || d->n_lights < 0)
        return fail("scene: negative or empty counts");
    if ((d->n_shapes && !d->shapes) || !d->world || (d->n_materials && !d->materials) || (d->n_textures && !d->textures))
        return fail("scene: null array with non-zero count");
    if ((d->n_xforms && !d->xforms) || (d->n_media && !d->media) || (d->n_bvh_nodes && !d->bvh_nodes) || (d->n_bvhs && !d->bvhs) ||
        (d->n_lights && !d->lights) || (d->n_texels && !d->texels) || (d->n_tri_uvs && !d->tri_uvs))
        return fail("scene: null array with non-zero count");
    for (int i = 0; i < d->n_textures; i++) {
        const PtTexture& t = d->textures[i];
        switch (t.type) {
        case PT_TEX_CONSTANT: break;
        case PT_TEX_CHECKER:
            if (t.odd < 0 || t.odd >= d->n_textures || t.even < 0 || t.even >= d->n_textures) return fail("texture: checker child out of range");
            if (t.odd >= i || t.even >= i) return fail("texture: checker children must precede the checker");
            mask |= F_CHECKER; break;
        case PT_TEX_NOISE: mask |= F_NOISE; break;
        case PT_TEX_IMAGE:
            if (t.nx <= 0 || t.ny <= 0 || t.texel_offset < 0 || (int64_t)t.texel_offset + (int64_t)3 * t.nx * t.ny > d->n_texels)
                return fail("texture: image texels out of range");
            mask |= F_IMAGE; break;
        default: return fail("texture: unknown type");
        }
    }
    for (int i = 0; i < d->n_materials; i++) {
        const PtMaterial& m = d->materials[i];
        const bool needs_tex = m.type == PT_MAT_LAMBERTIAN || m.type == PT_MAT_DIFFUSE_LIGHT || m.type == PT_MAT_ISOTROPIC;
        if (needs_tex && (m.texture < 0 || m.texture >= d->n_textures)) return fail("material: texture out of range");
        switch (m.type) {
        case PT_MAT_LAMBERTIAN: case PT_MAT_DIFFUSE_LIGHT: break;
        case PT_MAT_METAL: mask |= F_METAL; break;
        case PT_MAT_DIELECTRIC: mask |= F_DIELECTRIC; break;
        case PT_MAT_ISOTROPIC: mask |= F_ISOTROPIC; break;
        default: return fail("material: unknown type");
        }
    }
    for (int i = 0; i < d->n_xforms; i++)
        if (d->xforms[i].type != PT_XFORM_TRANSLATE && d->xforms[i].type != PT_XFORM_ROTATE_Y) return fail("xform: unknown type");
    for (int i = 0; i < d->n_shapes; i++) {
        const PtShape& s = d->shapes[i];
        if (s.material < -1 || s.material >= d->n_materials) return fail("shape: material out of range");
        if (s.xform_count < 0 || s.xform_first < 0 || (int64_t)s.xform_first + s.xform_count > d->n_xforms) return fail("shape: xform range out of bounds");
        if (s.xform_count > 0) mask |= F_XFORM;
        switch (s.type) {
        case PT_SHAPE_SPHERE: mask |= F_SPHERE; break;
        case PT_SHAPE_MOVING_SPHERE: mask |= F_MSPHERE; break;
        case PT_SHAPE_RECT_XY: case PT_SHAPE_RECT_XZ: case PT_SHAPE_RECT_YZ: mask |= F_RECT; break;
        case PT_SHAPE_BOX: mask |= F_BOX; break;
        case PT_SHAPE_TRIANGLE:
            if (s.tri_uv < -1 || s.tri_uv >= d->n_tri_uvs) return fail("shape: tri_uv out of range");
            mask |= F_TRIANGLE; break;
        default: return fail("shape: unknown type");
        }
    }
    auto world_shape_ok = [&](int idx) { return idx >= 0 && idx < d->n_shapes && d->shapes[idx].material >= 0; };
    for (int i = 0; i < d->n_media; i++) {
        const PtMedium& m = d->media[i];
        if (m.boundary < 0 || m.boundary >= d->n_shapes) return fail("medium: boundary out of range");
        if (m.material < 0 || m.material >= d->n_materials) return fail("medium: material out of range");
    }
    // explicit trees: children must have smaller indices than their parent (the builder's
    // post-order), which rules out cycles; depth is bounded for the traversal stack
    std::vector<int> depth((size_t)d->n_bvh_nodes, 0);
    for (int i = 0; i < d->n_bvh_nodes; i++) {
        const PtBvhNode& n = d->bvh_nodes[i];
        int dl = 0, dr = 0;
        for (int side = 0; side < 2; side++) {
            const int c = side ? n.right : n.left;
            if (c >= 0) {
                if (c >= i) return fail("bvh: child index must be smaller than its parent's");
                (side ? dr : dl) = depth[c];
            } else if (!world_shape_ok(~c)) return fail("bvh: leaf shape out of range or without material");
        }
        depth[i] = 1 + (dl > dr ? dl : dr);
        if (depth[i] >= REFBVH_STACK) return fail("bvh: tree deeper than the traversal stack");
    }
    for (int i = 0; i < d->n_bvhs; i++) {
        const PtBvh& b = d->bvhs[i];
        if (b.root < 0 || b.root >= d->n_bvh_nodes) return fail("bvh: root out of range");
        if (b.xform_count < 0 || b.xform_first < 0 || (int64_t)b.xform_first + b.xform_count > d->n_xforms) return fail("bvh: xform range out of bounds");
        if (b.xform_count > 0) mask |= F_XFORM;
    }
    for (int i = 0; i < d->n_world; i++) {
        const PtObject& o = d->world[i];
        if (o.type == PT_OBJ_SHAPE) { if (!world_shape_ok(o.index)) return fail("world: shape out of range or without material"); }
        else if (o.type == PT_OBJ_MEDIUM) { if (o.index < 0 || o.index >= d->n_media) return fail("world: medium out of range"); mask |= F_MEDIUM; }
        else if (o.type == PT_OBJ_BVH) { if (o.index < 0 || o.index >= d->n_bvhs) return fail("world: bvh out of range"); mask |= F_REFBVH; }
        else return fail("world: unknown object type");
    }
    if (d->n_lights < 0) return fail("lights: negative count");
    if (d->lights_is_list < 0 || d->lights_is_list > 2) return fail("lights: lights_is_list must be 0, 1 or 2");
    if (d->n_lights > 0 && d->lights_is_list == 2) {     // a BVH of light shapes
        if (d->n_lights != 1 || d->lights[0] < 0 || d->lights[0] >= d->n_bvhs) return fail("lights: bvh out of range");
        mask |= F_LIGHTS | F_LIGHT_MISC | F_REFBVH;
    } else if (d->n_lights > 0) {
        mask |= F_LIGHTS;
        for (int i = 0; i < d->n_lights; i++)
            if (d->lights[i] < 0 || d->lights[i] >= d->n_shapes) return fail("lights: shape out of range");
        const PtShape& l0 = d->shapes[d->lights[0]];
        if (d->lights_is_list || d->n_lights > 1 || l0.type != PT_SHAPE_RECT_XZ || l0.xform_count != 0 || (l0.flags & PT_FLIP_NORMAL)) mask |= F_LIGHT_MISC;
        if (!d->lights_is_list && d->n_lights != 1) return fail("lights: more than one light shape needs lights_is_list");
    }
    if (d->ambient_type == PT_AMBIENT_SKY) mask |= F_SKY;
    else if (d->ambient_type != PT_AMBIENT_NONE && d->ambient_type != PT_AMBIENT_CONSTANT) return fail("ambient: unknown type");
    *mask_out = mask;
    return true;
}

}  // namespace

extern "C" int pt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int pt_scene_create(const PtSceneDesc* desc, PtScene** out_scene)
{
    if (!desc || !out_scene) { pt::set_error("pt_scene_create: null argument"); return PT_ERR_INVALID; }
    *out_scene = nullptr;
    uint32_t mask = 0;
    if (!validate(desc, &mask)) return PT_ERR_INVALID;
    if (pt_device_count() <= 0) { pt::set_error("pt_scene_create: no HIP device is visible"); return PT_ERR_NO_DEVICE; }

    PtScene* sc = new PtScene();
    if (!hip_ok(hipGetDevice(&sc->device), "hipGetDevice")) { delete sc; return PT_ERR_HIP; }
    SceneView& v = sc->view;
    bool ok = upload(sc, desc->shapes, (size_t)desc->n_shapes, &v.shapes.p, "upload shapes")
           && upload(sc, desc->tri_uvs, (size_t)desc->n_tri_uvs, &v.tri_uvs.p, "upload tri_uvs")
           && upload(sc, desc->xforms, (size_t)desc->n_xforms, &v.xforms.p, "upload xforms")
           && upload(sc, desc->materials, (size_t)desc->n_materials, &v.materials.p, "upload materials")
           && upload(sc, desc->textures, (size_t)desc->n_textures, &v.textures.p, "upload textures")
           && upload(sc, desc->texels, (size_t)desc->n_texels, &v.texels.p, "upload texels")
           && upload(sc, desc->media, (size_t)desc->n_media, &v.media.p, "upload media")
           && upload(sc, desc->bvh_nodes, (size_t)desc->n_bvh_nodes, &v.bvh_nodes.p, "upload bvh nodes")
           && upload(sc, desc->bvhs, (size_t)desc->n_bvhs, &v.bvhs.p, "upload bvhs")
           && upload(sc, desc->world, (size_t)desc->n_world, &v.world.p, "upload world")
           && upload(sc, desc->lights, (size_t)desc->n_lights, &v.lights.p, "upload lights");
    if (!ok) { pt_scene_destroy(sc); return PT_ERR_HIP; }
    v.n_world = desc->n_world;
    v.n_lights = desc->n_lights;
    v.lights_is_list = desc->lights_is_list;
    v.ambient_type = desc->ambient_type;
    std::memcpy(v.ambient, desc->ambient_color, sizeof v.ambient);
    std::memcpy(v.camera, desc->camera, sizeof v.camera);
    v.lbvh = nullptr; v.lbvh_root = 0; v.lbvh_cull = 0; v.leaf_vote = 1;
    v.sig = nullptr;
    sc->n_shapes = desc->n_shapes;
    sc->n_media = desc->n_media;

    // Library-built LBVH over the world list (in place of BVH::BVH + quickSort).  A list of shapes
    // goes into the tree whole.  A list with reference BVHs and media goes in up to its first
    // medium (or wrapped BVH): the shapes and the BVHs' leaves of that leading run become the
    // tree's leaves and the walk continues in list order behind it (lbvh_plan).
    v.lbvh_order = nullptr; v.lbvh_group = nullptr; v.ref_leaf_gate = nullptr;
    v.tree_planes = nullptr; v.tree_plane_first = nullptr; v.tree_plane_members = nullptr; v.tree_plane_end[0] = v.tree_plane_end[1] = v.tree_plane_end[2] = 0; v.tree_wrapped = 0;
    v.lbvh_prefix = desc->n_world;
    v.reads_uv = (mask & F_IMAGE) ? 1 : 0;
    bool tree_sized = true;                                    // see ray_may_enter_tree: shapes within 2^40
    for (int i = 0; i < desc->n_shapes && tree_sized; i++)
        for (float q : desc->shapes[i].p) if (!(std::fabs(q) <= ptd::TREE_RAY_LIMIT)) tree_sized = false;
    for (int i = 0; i < desc->n_xforms && tree_sized; i++)
        for (float q : desc->xforms[i].v) if (!(std::fabs(q) <= ptd::TREE_RAY_LIMIT)) tree_sized = false;
    v.lbvh4 = nullptr;
    const bool want_tree = desc->accel == PT_ACCEL_LBVH || desc->accel == PT_ACCEL_LBVH_BINARY || desc->accel == PT_ACCEL_LBVH_WIDE;
    if (want_tree && desc->n_world >= 2 && tree_sized) {
        void* d_nodes = nullptr;
        void* d_wide = nullptr;
        // Which form the walk reads when the caller leaves it open: the 4-wide one (64-byte nodes, child boxes on a
        // byte grid, LbvhNode4).  Measured on MI355X, binary / 4-wide under the same schedule (kernel ms): random
        // spheres 1920x1080x64 38.0 / 35.2, 2 M triangles 3840x2160x64 185.5 / 158.1, a 20 000-triangle mesh
        // 1920x1080x64 48.2 / 41.6, RTNW final 1920x1080x64 62.0 / 60.1.  (With 128-byte nodes holding the four boxes
        // as floats the same comparison read 47.7 / 62.0, 59.7 / 63.7 per 16 spp, 62.2 / 63.2: four full box tests
        // per fetch cost the vector unit more than the saved fetches gave back.)  Both forms are built; the binary
        // one serves the walks that are not the search proper (rays lying in a plane of the tree, the node export).
        void** want_wide = (desc->accel == PT_ACCEL_LBVH_WIDE || desc->accel == PT_ACCEL_LBVH) ? &d_wide : nullptr;
        int n_nodes = 0, root = 0, rc = PT_OK, depth = 0;
        bool built = false;
        // Tree shape by surface area for worlds of plain shapes (spheres 1920x1080x64 41.9 -> 37.9 ms, a 20 000-
        // triangle mesh 46.1 -> 42.8, a room of seven spheres 50.5 -> 39.4); the final scene's grid of boxes
        // walks 19 % more nodes that way than in Morton order (61.5 -> 62.1 ms) and keeps the Morton tree.
        const bool area_sweep = !(mask & (F_MEDIUM | F_REFBVH));
        LbvhPlan plan;
        bool in_table_order = !(mask & (F_MEDIUM | F_REFBVH));     // shapes only, listed in the order of the shape table:
        for (int i = 1; i < desc->n_world && in_table_order; i++)     // then the shape index is the tie order
            in_table_order = desc->world[i].index > desc->world[i - 1].index;
        if (in_table_order) {
            rc = pt::lbvh_build(desc, v.shapes.p, v.xforms.p, &d_nodes, &n_nodes, &root, want_wide, &depth, area_sweep);
            built = true;
        } else {
            if (lbvh_plan(desc, &plan)) {
                rc = pt::lbvh_build_leaves(plan.leaves.data(), (int)plan.leaves.size(), desc->camera[21], desc->camera[22],
                                           v.shapes.p, v.xforms.p, &d_nodes, &n_nodes, &root, want_wide, &depth, area_sweep);
                if (rc == PT_OK && !(upload(sc, plan.order.data(), plan.order.size(), &v.lbvh_order.p, "upload leaf order") &&
                                     upload(sc, plan.group.data(), plan.group.size(), &v.lbvh_group.p, "upload leaf groups") &&
                                     upload(sc, plan.leaf_gate.data(), plan.leaf_gate.size(), &v.ref_leaf_gate.p, "upload leaf gates"))) {
                    (void)hipFree(d_nodes); if (d_wide) (void)hipFree(d_wide); pt_scene_destroy(sc); return PT_ERR_HIP;
                }
                v.lbvh_prefix = plan.prefix;
                built = true;
            }
        }
        // The walks keep pending nodes on a per-lane stack of LBVH_STACK entries: one per level for the
        // binary tree, up to three per level of the 4-wide one.  A tree too deep for that (many leaves
        // with one centroid) is not used: the wide form first, then the tree altogether (list walk).
        if (built && rc == PT_OK && d_wide && 3 * ((depth + 1) / 2) > ptd::LBVH_STACK) { (void)hipFree(d_wide); d_wide = nullptr; }
        if (built && rc == PT_OK && depth > ptd::LBVH_STACK) {
            (void)hipFree(d_nodes); d_nodes = nullptr;
            if (d_wide) { (void)hipFree(d_wide); d_wide = nullptr; }
            v.lbvh_order = nullptr; v.lbvh_group = nullptr; v.ref_leaf_gate = nullptr; v.lbvh_prefix = desc->n_world;
            built = false;
        }
        if (built && rc == PT_OK) {
            // plane coordinates of the tree's un-wrapped rectangles and box faces, per axis (ray_may_enter_tree)
            std::vector<std::pair<float, int32_t>> on_axis[3];      // (plane, shape)
            bool wrapped = false;
            const PtObject* leaves = in_table_order ? desc->world : plan.leaves.data();
            const size_t n_leaves = in_table_order ? (size_t)desc->n_world : plan.leaves.size();
            for (size_t i = 0; i < n_leaves; i++) {
                const int32_t id = leaves[i].index;
                const PtShape& sh = desc->shapes[id];
                const bool flat = sh.type == PT_SHAPE_RECT_XY || sh.type == PT_SHAPE_RECT_XZ || sh.type == PT_SHAPE_RECT_YZ;
                if (!flat && sh.type != PT_SHAPE_BOX) continue;
                if (sh.xform_count > 0) { wrapped = true; continue; }
                auto put = [&](int axis, float k) { on_axis[axis].push_back({ k == 0.0f ? 0.0f : k, id }); };    // -0 and +0 are one plane
                if (sh.type == PT_SHAPE_RECT_XY) put(2, sh.p[4]);
                else if (sh.type == PT_SHAPE_RECT_XZ) put(1, sh.p[4]);
                else if (sh.type == PT_SHAPE_RECT_YZ) put(0, sh.p[4]);
                else for (int k = 0; k < 3; k++) { put(k, sh.p[k]); put(k, sh.p[3 + k]); }
            }
            std::vector<float> all;
            std::vector<int32_t> first, members;
            for (int k = 0; k < 3; k++) {
                std::sort(on_axis[k].begin(), on_axis[k].end());
                on_axis[k].erase(std::unique(on_axis[k].begin(), on_axis[k].end()), on_axis[k].end());
                for (size_t i = 0; i < on_axis[k].size(); i++) {
                    if (i == 0 || on_axis[k][i].first != on_axis[k][i - 1].first) { all.push_back(on_axis[k][i].first); first.push_back((int32_t)members.size()); }
                    members.push_back(on_axis[k][i].second);
                }
                v.tree_plane_end[k] = (int32_t)all.size();
            }
            first.push_back((int32_t)members.size());
            v.tree_wrapped = wrapped ? 1 : 0;
            if (all.empty()) all.push_back(0.0f);             // never read: every run is empty
            if (members.empty()) members.push_back(0);
            if (!(upload(sc, all.data(), all.size(), &v.tree_planes.p, "upload tree planes") &&
                  upload(sc, first.data(), first.size(), &v.tree_plane_first.p, "upload tree plane runs") &&
                  upload(sc, members.data(), members.size(), &v.tree_plane_members.p, "upload tree plane members"))) {
                (void)hipFree(d_nodes); if (d_wide) (void)hipFree(d_wide); pt_scene_destroy(sc); return PT_ERR_HIP;
            }
        }
        if (built) {
            if (rc != PT_OK) { pt_scene_destroy(sc); return rc; }
            sc->allocs.push_back(d_nodes);
            v.lbvh = static_cast<const LbvhNode*>(d_nodes);
            v.lbvh_root = root;
            v.lbvh_cull = (mask & F_TRIANGLE) ? 0 : 1;
            v.leaf_vote = (mask & F_TRIANGLE) ? (desc->n_shapes >= 100000 ? 8 : 1) : (mask & (F_MEDIUM | F_REFBVH)) ? 16 : 12;   // see lbvh_voted_step
            sc->lbvh_nodes = n_nodes;
            mask |= F_LBVH;
            if (d_wide) {
                sc->allocs.push_back(d_wide);
                v.lbvh4 = static_cast<const ptd::LbvhNode4*>(d_wide);
                mask |= F_LBVH4;
            }
        }
    }
    sc->mask = mask;
    sc->variant = pick_variant(mask);
    v.sig_faces = nullptr;
    for (float& f : v.sig_light) f = 0.0f;
    std::vector<float> faces;
    // the signature kernels shade from the face table and sample the one light shape: both must exist
    if (sc->variant == V_CORNELL && world_matches(desc, CornellWorld{}) && desc->n_lights == 1 && signature_faces(desc, &faces)) {
        const std::vector<float> table = signature_table(desc);
        if (!upload(sc, table.data(), table.size(), &v.sig.p, "upload signature table") ||
            !upload(sc, faces.data(), faces.size(), &v.sig_faces.p, "upload signature faces")) { pt_scene_destroy(sc); return PT_ERR_HIP; }
        for (int k = 0; k < 5; k++) v.sig_light[k] = desc->shapes[desc->lights[0]].p[k];
        sc->variant = V_CORNELL_WORLD;
    }
    if (sc->variant < 0) { pt::set_error("pt_scene_create: no kernel specialisation covers this scene"); pt_scene_destroy(sc); return PT_ERR_UNSUPPORTED; }
    {
        void* copy = nullptr;                                  // the view itself, for device code kept out of line
        if (!hip_ok(hipMalloc(&copy, sizeof(SceneView)), "hipMalloc(scene view)")) { pt_scene_destroy(sc); return PT_ERR_OOM; }
        sc->allocs.push_back(copy);
        v.self = static_cast<const SceneView*>(copy);
        if (!hip_ok(hipMemcpy(copy, &v, sizeof(SceneView), hipMemcpyHostToDevice), "hipMemcpy(scene view)")) { pt_scene_destroy(sc); return PT_ERR_HIP; }
    }
    {
        void* ctr = nullptr;
        if (!hip_ok(hipMalloc(&ctr, 256), "hipMalloc(slot counter)")) { pt_scene_destroy(sc); return PT_ERR_OOM; }
        sc->allocs.push_back(ctr);
        sc->next_slot = static_cast<unsigned*>(ctr);
    }
    {
        // a word of pinned host memory the pass kernel can write: "this accumulator is not where first_sample says"
        void* flag = nullptr;
        if (!hip_ok(hipHostMalloc(&flag, 64, hipHostMallocMapped), "hipHostMalloc(resume flag)")) { pt_scene_destroy(sc); return PT_ERR_OOM; }
        sc->bad_resume_host = static_cast<int*>(flag);
        *sc->bad_resume_host = 0;
        void* dev = nullptr;
        if (!hip_ok(hipHostGetDevicePointer(&dev, flag, 0), "hipHostGetDevicePointer")) { pt_scene_destroy(sc); return PT_ERR_HIP; }
        sc->bad_resume_dev = static_cast<int*>(dev);
    }
    if (!scratch_on_device(sc->device)) { pt_scene_destroy(sc); return PT_ERR_HIP; }     // set-up for pt_render_device_noalloc
    ok = hip_ok(hipEventCreate(&sc->ev_begin), "hipEventCreate") && hip_ok(hipEventCreate(&sc->ev_k0), "hipEventCreate")
      && hip_ok(hipEventCreate(&sc->ev_k1), "hipEventCreate") && hip_ok(hipEventCreate(&sc->ev_end), "hipEventCreate");
    if (!ok) { pt_scene_destroy(sc); return PT_ERR_HIP; }
    *out_scene = sc;
    return PT_OK;
}

extern "C" void pt_scene_destroy(PtScene* sc)
{
    if (!sc) return;
    for (void* p : sc->allocs) (void)hipFree(p);
    if (sc->bad_resume_host) (void)hipHostFree(sc->bad_resume_host);
    if (sc->wf_workspace) (void)hipFree(sc->wf_workspace);
    if (sc->chunk_workspace) (void)hipFree(sc->chunk_workspace);
    if (sc->staging) (void)hipFree(sc->staging);
    if (sc->ev_begin) (void)hipEventDestroy(sc->ev_begin);
    if (sc->ev_k0) (void)hipEventDestroy(sc->ev_k0);
    if (sc->ev_k1) (void)hipEventDestroy(sc->ev_k1);
    if (sc->ev_end) (void)hipEventDestroy(sc->ev_end);
    delete sc;
}

namespace {

bool make_args(const PtRenderParams* p, RenderArgs* ra)
{
    if (p->nx <= 0 || p->ny <= 0 || p->ns <= 0 || p->max_depth < 0) return fail("render: nx, ny, ns must be positive and max_depth non-negative");
    if ((int64_t)p->nx * p->ny > (int64_t)1 << 30) return fail("render: image too large");
    const int sc = p->shard_count > 0 ? p->shard_count : 1;
    if (p->shard_index < 0 || p->shard_index >= sc) return fail("render: shard_index out of range");
    const int tw = p->tile_w > 0 ? p->tile_w : 8, th = p->tile_h > 0 ? p->tile_h : 8;
    ra->nx = p->nx; ra->ny = p->ny; ra->ns = p->ns; ra->max_depth = p->max_depth;
    ra->shard_index = p->shard_index; ra->shard_count = sc;
    ra->tile_w = tw; ra->tile_h = th;
    ra->tiles_x = (p->nx + tw - 1) / tw;
    const int tiles_y = (p->ny + th - 1) / th;
    ra->n_tiles = ra->tiles_x * tiles_y;
    ra->sub_x = (tw + 7) / 8; ra->sub_y = (th + 7) / 8;
    const int owned = (ra->n_tiles - p->shard_index + sc - 1) / sc;     // tiles t with t % sc == shard_index
    const int64_t wave_tiles = (int64_t)owned * ra->sub_x * ra->sub_y;
    // pixel slots are counted in 32 bits (64 per wave tile): tiles much smaller than 8x8 on a huge frame would wrap them
    if (wave_tiles > ((int64_t)1 << 25)) return fail("render: too many tiles (more than 2^25 8x8 wave tiles in this shard); use larger tiles");
    ra->n_wave_tiles = (int32_t)wave_tiles;
    ra->refill = 0;
    if (p->batch_count < 0 || p->batch_index < 0 || (p->batch_count > 0 && p->batch_index >= p->batch_count) ||
        (p->batch_count == 0 && p->batch_index != 0)) return fail("render: batch_index out of range");
    ra->seed_offset = (uint64_t)p->batch_index * (uint64_t)p->nx * (uint64_t)p->ny;
    ra->raw_sum = p->raw_sum ? 1 : 0;
    ra->stage_stats = p->stage_stats ? 1 : 0;
    ra->wf_chunk_samples = p->chunk_samples > 0 ? p->chunk_samples : 0;
    return true;
}

}  // namespace

static uint64_t owned_pixels(const PtRenderParams* p, const RenderArgs& ra)
{
    uint64_t owned_px = 0;
    for (int t = ra.shard_index; t < ra.n_tiles; t += ra.shard_count) {
        const int ty = t / ra.tiles_x, tx = t % ra.tiles_x;
        const int w = std::min(ra.tile_w, p->nx - tx * ra.tile_w), h = std::min(ra.tile_h, p->ny - ty * ra.tile_h);
        owned_px += (uint64_t)w * (uint64_t)h;
    }
    return owned_px;
}

namespace {
// The signature walk comes with and without control flow between its tests.  Without, the walk is
// one scheduling block whose loads and divisions overlap: faster when a SIMD holds few waves to
// switch to, slightly slower (more selects) when it holds many.  Measured on an MI355X (1024 SIMDs),
// Cornell 1024^2 at 512 spp, with / without control flow: whole frame 47.7 / 48.6 ms, half 28.6 / 28.4,
// a quarter 19.7 / 19.2, an eighth 15.1 / 14.2: the two meet at eight wave tiles per SIMD.
constexpr int FLAT_UP_TO_WAVES = 8 * 1024;
constexpr int CHUNK_WAVE_SLOTS = 256 * 4 * 6;      // wave slots of the GPU at the rectangle-scene kernels' six waves per SIMD
constexpr int CHUNK_MIN_SAMPLES = 64;             // AUTO cuts no finer: a chunk's start and end cost a few microseconds
int launch_variant(const PtScene* sc, const RenderArgs& ra)
{
    if (sc->variant != V_CORNELL_WORLD) return sc->variant;
    // in chunks the GPU stays full of waves as long as the shard has more wave tiles than wave slots: half the headline
    // frame x4096 in chunks 133.7 ms without control flow, 126.4 with; a quarter 83.2 and 90.7
    if (ra.n_chunks > 0) return ra.n_wave_tiles <= CHUNK_WAVE_SLOTS ? V_CORNELL_WORLD_FLAT : V_CORNELL_WORLD;
    return ra.n_wave_tiles <= FLAT_UP_TO_WAVES ? V_CORNELL_WORLD_FLAT : V_CORNELL_WORLD;
}
}  // namespace

// slot_counter: the device word the persistent-lane schedules count pixel slots in -- the scene's own
// allocation, or this GPU's static scratch for the call that may not allocate.
static int render_device(PtScene* sc, const PtRenderParams* p, void* d_out_rgb, void* hip_stream, unsigned* slot_counter, bool no_alloc)
{
    RenderArgs ra;
    if (!make_args(p, &ra)) return PT_ERR_INVALID;
    if (p->pipeline != PT_PIPELINE_DEFAULT && p->pipeline != PT_PIPELINE_FUSED && p->pipeline != PT_PIPELINE_WAVEFRONT) {
        pt::set_error("render: unknown pipeline"); return PT_ERR_INVALID;
    }
    if (p->schedule != PT_SCHEDULE_AUTO && p->schedule != PT_SCHEDULE_TILES && p->schedule != PT_SCHEDULE_REFILL &&
        p->schedule != PT_SCHEDULE_WALK && p->schedule != PT_SCHEDULE_CHUNKS) {
        pt::set_error("render: unknown schedule"); return PT_ERR_INVALID;
    }
    if (p->chunk_samples < 0) { pt::set_error("render: chunk_samples is negative"); return PT_ERR_INVALID; }
    // The chunked tiles keep state in a workspace of the scene (states per wave tile, words per group).  The call that may
    // not allocate takes them only when an earlier pt_render_device call has left the scene a workspace that is large enough.
    const size_t tile_chunk_bytes = (size_t)ra.n_wave_tiles * 1280 + (4 + 2 * (size_t)ra.n_wave_tiles) * sizeof(unsigned);
    const bool chunk_workspace_ready = sc->chunk_workspace != nullptr && sc->chunk_workspace_bytes >= tile_chunk_bytes;
    if (p->schedule == PT_SCHEDULE_CHUNKS && no_alloc && !chunk_workspace_ready) {
        pt::set_error("pt_render_device_noalloc: PT_SCHEDULE_CHUNKS keeps per-tile state in memory the scene allocates, and no earlier "
                      "pt_render_device call of this shard size has left it there; make one, or use another schedule");
        return PT_ERR_UNSUPPORTED;
    }
    if (p->schedule == PT_SCHEDULE_WALK && !((sc->mask & ptd::F_LBVH) && sc->view.lbvh_prefix == sc->view.n_world && p->max_depth >= 1)) {
        pt::set_error("render: PT_SCHEDULE_WALK needs the whole world in the library's LBVH and a bounce limit of at least 1");
        return PT_ERR_UNSUPPORTED;
    }
    // Refilling keeps waves full but puts their lanes out of phase; that pays when a bounce is
    // mostly the (wave-uniform) closest-hit search, not when shading dominates it.
    ra.refill = (p->schedule == PT_SCHEDULE_REFILL) ||
                (p->schedule == PT_SCHEDULE_AUTO && !(sc->mask & ptd::F_LBVH) && ((sc->mask & ptd::F_REFBVH) || sc->view.n_world > 16));
    // Measured (MI355X, kernel ms; TILES / REFILL / WALK): spheres in the LBVH, 1920x1080x64: 48.0 / 49.7 / 43.4;
    // RTNW final through the LBVH, 1920x1080x64: 61.0 / 65.7 / -; 2 M triangles in the LBVH, 3840x2160x16:
    // 133 / - / 55.0 with 48 waiting lanes (40: 55.9, 56: 65.9; spheres 24: 49.0, 32: 45.5, 40: 44.5, 56: 44.6).
    const bool whole_world_in_tree = (sc->mask & ptd::F_LBVH) && sc->view.lbvh_prefix == sc->view.n_world && p->max_depth >= 1;
    if (p->schedule == PT_SCHEDULE_WALK ||
        (p->schedule == PT_SCHEDULE_AUTO && whole_world_in_tree && ((sc->mask & ptd::F_TRIANGLE) || ((sc->variant == V_SPHERES_LBVH || sc->variant == V_SPHERES_LBVH4) && sc->view.n_world >= 64)))) ra.refill = 2;   // a 7-sphere room: TILES 47.9, WALK 50.8
    // TILES in chunks of samples (k_render_chunks): asked for, or under AUTO for the kernels it was measured to pay on,
    // when the tiles are few for the GPU's 6 144 wave slots and the samples many: the frame is then cut into about 32
    // rounds of work items.  Measured (MI355X, kernel ms, TILES / CHUNKS): rectangle scene 1024^2 x4096 241.1 / 211.6
    // (12 chunks of 342 samples; chunks of 64 / 128 / 256 / 512 / 1024: 217.7 / 212.0 / 211.8 / 213.3 / 222.7), x512
    // 31.7 / 29.3; half of that frame (8 192 tiles, 1.3 rounds) 18.3 / 19.7 and an eighth 9.10 / 11.0, so launches of
    // less than two rounds stay with TILES; RTNW final through the 4-wide tree 1920x1080 x256 205.3 / 192.1; seven
    // spheres in the room (list walk) 1024^2 x1024 302.9 / 325.4: its chunk kernel compiles 10 % slower, so it stays too.
    ra.chunk_samples = ra.n_chunks = 0;
    // With the later chunks of expensive tiles at a raised wave priority (k_render_chunks) the chunks also pay on shards
    // of two thirds of a round and more once a pixel has a thousand samples -- what two to four GPUs hold of the
    // headline frame (kernel ms, TILES / CHUNKS without / with the priority): x4096 whole frame 241 / 212.3 / 208.4, half
    // 138.9 / 134.1 / 134.3, a quarter 93.9 / 102.2 / 83.2, an eighth 69.1 / 85.2 / 78.1 (stays with TILES); x1024 a third
    // 28.1 / - / 25.6, a quarter 24.3 / - / 22.4, a sixth 20.6 / - / 21.0; x512 half 18.2 / - / 19.4, a quarter 12.5 / - / 12.5.
    const bool chunks_pay = (sc->variant == V_CORNELL_WORLD || sc->variant == V_FINAL_LBVH4 || sc->variant == V_FINAL_LBVH) &&
                            (ra.n_wave_tiles >= 2 * CHUNK_WAVE_SLOTS || (3 * ra.n_wave_tiles >= 2 * CHUNK_WAVE_SLOTS && p->ns >= 1024));
    if (p->pipeline != PT_PIPELINE_WAVEFRONT && ra.refill == 0 && ra.n_wave_tiles > 0 && (!no_alloc || chunk_workspace_ready) &&
        (p->schedule == PT_SCHEDULE_CHUNKS || (p->schedule == PT_SCHEDULE_AUTO && chunks_pay))) {
        int chunk = p->chunk_samples;
        if (chunk == 0) {
            const long long want_items = 32LL * CHUNK_WAVE_SLOTS;
            const int pieces = (int)std::min<long long>((want_items + ra.n_wave_tiles - 1) / ra.n_wave_tiles, std::max(1, p->ns / CHUNK_MIN_SAMPLES));
            chunk = (p->ns + pieces - 1) / std::max(1, pieces);
        }
        chunk = std::max(1, std::min(chunk, std::max(1, p->ns)));
        const int pieces = std::max(1, (p->ns + chunk - 1) / chunk);
        // work items (chunk, group of wave tiles) are numbered, and drawn from a counter, in 32 bits
        if ((long long)ra.n_wave_tiles * pieces > 0x7fffffffLL) {
            pt::set_error("render: PT_SCHEDULE_CHUNKS with this chunk_samples cuts the shard into more than 2^31 work items; use larger chunks");
            return PT_ERR_INVALID;
        }
        if (p->schedule == PT_SCHEDULE_CHUNKS || pieces >= 2) { ra.chunk_samples = chunk; ra.n_chunks = pieces; }
    }
    // The persistent-lane schedules (REFILL, WALK) in chunks of a pixel's samples: slots are (chunk, pixel), see
    // chunk_record_store.  With more pixels than lanes an uncut launch ends pixel by pixel (the lanes that drew the last
    // slots trace whole pixels while the rest of the GPU has run dry); in chunks every pixel advances sweep by sweep and
    // the launch ends within a chunk.  What a cut costs is the wave, not the lane: the lane's new slot (an atomic's round
    // trip), its pixel's record (another) and the code around them run for one lane while 63 wait.  Measured (MI355X,
    // kernel ms, uncut / 4 / 8 / 16 / 32 chunks): random spheres in the tree under WALK 1920x1080 x256 139.8 / 136.4 /
    // 152.1 / 203.7 / 272.0 (the same paths as a 3840x2160 frame at 64 samples, where the end does not matter, take 124.5:
    // the cuts' price is what keeps the rest of that gain); 2 M triangles under WALK 3840x2160 x64 157.1 / 192.7 / 225.5 (a
    // wave waiting on memory all along pays most for one more wait); RTNW final through the reference's tree under REFILL
    // x64 163.0 / 158.7 / 158.0, x256 639.4 / - / - / 626.6; random spheres over the flat list under REFILL x16 90.4 / 77.4
    // (a list walk is uniform over the wave and long: the cut is cheap beside it).  Those were measured with one slot counter
    // for the whole GPU, and much of a cut's price was the wait for that word: with a counter per XCD (slot_band_draw) the
    // same sweeps read, uncut / 4 / 8 / 16 / 32 chunks: random spheres under WALK x256 139.2 / 128.6 / 132.3 / 138.0 / 143.0,
    // x1024 550.9 / 501.3 / 496.5 / 506.9 / 521.8; 2 M triangles x64 uncut / 2 / 4 151.9 / 149.0 / 147.8; final through the
    // reference's tree under REFILL x64 149.8 / 149.2 / 149.8 / 148.4 / 143.4; random spheres over the flat list x64 246.6 /
    // 209.3 / 201.8 / 199.4 / 195.1, and in chunks of one sample 133.3 and 194.7 (final x256: 8 chunks 578.9, chunks of 2
    // samples 571.3, of one 530.8): a lane that draws a slot per sample traces neighbouring pixels beside its wave's other
    // lanes all along, not only at the start of the launch, and the walk of the reference's tree diverges less.  (Only where
    // a sample is long: the rectangle scene under REFILL takes 108 ms in chunks of 8 and 124 in chunks of one against 28 as
    // chunked tiles.)  So: REFILL in chunks of one sample; WALK 4 chunks of at least 32 samples
    // (8 of at least 64 from 512 samples on), with triangles 4 of at least 16; only for shards of more than one and a half
    // sweeps of the GPU's lanes.  chunk_samples: 0 the library's choice, >= ns no chunks.
    if (p->pipeline != PT_PIPELINE_WAVEFRONT && ra.refill != 0 && ra.n_wave_tiles > 0 && !no_alloc && p->ns >= 2) {
        const long long lanes = 256LL * 4 * 64 * 6, slots = (long long)ra.n_wave_tiles * 64;
        int pieces = 1;
        if (p->chunk_samples > 0) pieces = (p->ns + p->chunk_samples - 1) / p->chunk_samples;
        else if (2 * slots > 3 * lanes) {
            if (ra.refill == 1) pieces = p->ns;
            else if (sc->mask & ptd::F_TRIANGLE) pieces = std::min(4, p->ns / 16);
            else pieces = p->ns >= 512 ? std::min(8, p->ns / 64) : std::min(4, p->ns / 32);
        }
        pieces = (int)std::min<long long>(pieces, 0xffffffffLL / slots);      // slots are numbered in 32 bits
        if (pieces >= 2) {
            ra.chunk_samples = (p->ns + pieces - 1) / pieces;
            ra.n_chunks = (p->ns + ra.chunk_samples - 1) / ra.chunk_samples;
            if (ra.n_chunks < 2) ra.chunk_samples = ra.n_chunks = 0;
        }
    }
    ra.chunk_priority = 1;
    // Pixel slots of the persistent-lane schedules from one counter per XCD (slot_band_draw).  One counter against eight with
    // blocks of 64 wave tiles (kernel ms): 2 M triangles 3840x2160 x16 49.1 -> 40.4, x64 154.3 -> 153.6 (blocks of 8 / 32 / 256 /
    // 1024: x16 40.7 / 40.0 / 41.2 / 41.0); random spheres in the tree 1920x1080 x16 12.2 -> 9.7, x256 132.5 -> 129.5, 3840x2160 x64
    // 139.2 -> 127.2.
    ra.slot_bands = p->slot_block_tiles < 0 ? 1 : 8;
    ra.band_block_tiles = p->slot_block_tiles > 0 ? p->slot_block_tiles : 64;
    // Lanes with a finished search that make a wave leave the shared walk's search loop.  32 / 40 / 48 / 56 / 62 of 64 lanes, first sweep:
    // sphere scene 38.2 / 36.1 / 35.6 / 35.6 / 36.4 ms, triangle field 35.1 / 31.2 / 28.8 / 29.1 / 42.7.  Swept again with the slot
    // counters per XCD (the waves no longer wait for their slots, the balance between walking and shading lanes moved): 2 M
    // triangles 3840x2160 x64 at 16 / 24 / 32 / 36 / 40 / 44 / 48 / 56 = 155.4 / 141.1 / 136.0 / 133.8 / 135.6 / 138.9 / 147.8 / 197.9 ms
    // (leaf vote 8; 12 and 6 within 2 %), x16 42.5 / 39.7 / 37.5 / 37.5 / - / 38.6 / 40.4; sphere scene x256 at 40 / 48 / 56 = 133.7 /
    // 129.1 / 130.6; meshes of 20 000 and 1 800 triangles at 32 / 40 / 48 = 39.0 / 38.5 / 36.5 and 30.1 / 28.5 / 28.1.
    ra.walk_wait = ((sc->mask & ptd::F_TRIANGLE) && sc->n_shapes >= 100000) ? 36 : 48;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    float* out = static_cast<float*>(d_out_rgb);
    // A stream under graph capture records the work, it does not run it: the scene's timing events stay out of the
    // graph (an event recorded there belongs to the capture and can no longer be waited for or timed from outside
    // it), so pt_render_stats keeps describing the last render that ran outside a graph; replays are ordered by the
    // stream they are launched on.
    bool capturing = false;
    if (st) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (!hip_ok(hipStreamIsCapturing(st, &cs), "hipStreamIsCapturing")) return PT_ERR_HIP;
        capturing = cs != hipStreamCaptureStatusNone;
    }
    if (capturing && ra.n_chunks > 0 && (p->schedule == PT_SCHEDULE_AUTO || ra.refill != 0)) ra.n_chunks = ra.chunk_samples = 0;     // one work item per pixel or tile instead
    if (capturing && (p->pipeline == PT_PIPELINE_WAVEFRONT || ra.n_chunks > 0)) {
        pt::set_error("render: a stream under graph capture takes the fused pipeline's TILES, REFILL and WALK schedules only "
                      "(the staged pipeline synchronises with the host, the CHUNKS schedule may allocate)");
        return PT_ERR_UNSUPPORTED;
    }
    // One render of a scene at a time: the pixel-slot counter and the timing events belong to the
    // scene, so a call on another stream first waits for the scene's previous render to finish.
    if (!capturing && sc->have_timing && !hip_ok(hipStreamWaitEvent(st, sc->ev_end, 0), "hipStreamWaitEvent")) return PT_ERR_HIP;
    if (!capturing && !hip_ok(hipEventRecord(sc->ev_begin, st), "hipEventRecord")) return PT_ERR_HIP;
    // pixels this shard does not own must read as zero so that shards sum to the frame
    if (!hip_ok(hipMemsetAsync(out, 0, (size_t)3 * p->nx * p->ny * sizeof(float), st), "hipMemsetAsync")) return PT_ERR_HIP;
    if (!capturing && !hip_ok(hipEventRecord(sc->ev_k0, st), "hipEventRecord")) return PT_ERR_HIP;
    const PtRenderStats before_capture = sc->last;
    sc->last = PtRenderStats{};
    sc->last.kernel_family = kernel_family(sc->variant);
    if (p->pipeline == PT_PIPELINE_WAVEFRONT && (ra.seed_offset != 0 || ra.raw_sum)) {
        pt::set_error("render: sample batches run on the fused pipeline"); return PT_ERR_UNSUPPORTED;
    }
    if (p->pipeline == PT_PIPELINE_WAVEFRONT && no_alloc) {
        pt::set_error("pt_render_device_noalloc: the staged pipeline allocates its queues; use the fused pipeline"); return PT_ERR_UNSUPPORTED;
    }
    if (p->pipeline == PT_PIPELINE_WAVEFRONT) {
        const int rc = pt::render_wavefront(sc, ra, out, st);       // one launch; blocks until it has finished
        if (rc != PT_OK) return rc;
        sc->last.pipeline = PT_PIPELINE_WAVEFRONT;
    } else {
        // the counter keeps the last call's final value: it starts every call at zero
        // (the whole 256-byte block: the shared walk counts in up to eight words of it, one per band of slots)
        if (!hip_ok(hipMemsetAsync(slot_counter, 0, PT_SCRATCH_BYTES, st), "hipMemsetAsync(slot counters)")) return PT_ERR_HIP;
        if (ra.n_chunks > 0 && ra.refill != 0) {
            const size_t rec_bytes = (size_t)p->nx * (size_t)p->ny * 5 * sizeof(unsigned long long);     // CHUNK_REC_WORDS per pixel
            if (sc->chunk_workspace_bytes < rec_bytes) {
                if (sc->chunk_workspace) { (void)hipFree(sc->chunk_workspace); sc->chunk_workspace = nullptr; sc->chunk_workspace_bytes = 0; }
                if (hipMalloc(&sc->chunk_workspace, rec_bytes) != hipSuccess) {
                    (void)hipGetLastError();
                    sc->chunk_workspace = nullptr;
                    if (p->chunk_samples > 0) { pt::set_error("render: out of device memory for the chunk records of the persistent-lane schedule"); return PT_ERR_OOM; }
                    ra.chunk_samples = ra.n_chunks = 0;             // the library's own choice: trace the pixels whole instead
                } else sc->chunk_workspace_bytes = rec_bytes;
            }
            if (ra.n_chunks > 0 && !hip_ok(hipMemsetAsync(sc->chunk_workspace, 0, rec_bytes, st), "hipMemsetAsync(chunk records)")) return PT_ERR_HIP;
            launch_render(launch_variant(sc, ra), sc->view, ra, out, slot_counter, static_cast<unsigned long long*>(sc->chunk_workspace), sc->bad_resume_dev, st);
        } else if (ra.n_chunks > 0) {
            const size_t groups = (size_t)ra.n_wave_tiles;          // progress words: one per work item group, at most one per wave tile
            const size_t states_bytes = (size_t)ra.n_wave_tiles * 1280;         // CHUNK_TILE_STATE_WORDS 64-bit words per wave tile
            const size_t words_bytes = (4 + 2 * groups) * sizeof(unsigned);      // ticket, cost sum and count; a progress and a cost word per group
            if (sc->chunk_workspace_bytes < states_bytes + words_bytes) {
                if (no_alloc) { pt::set_error("render: the chunk workspace is too small for a call that may not allocate"); return PT_ERR_UNSUPPORTED; }     // (chunk_workspace_ready said otherwise)
                if (sc->chunk_workspace) { (void)hipFree(sc->chunk_workspace); sc->chunk_workspace = nullptr; sc->chunk_workspace_bytes = 0; }
                if (hipMalloc(&sc->chunk_workspace, states_bytes + words_bytes) != hipSuccess) {
                    (void)hipGetLastError();
                    sc->chunk_workspace = nullptr;
                    pt::set_error("render: out of device memory for the chunked schedule's per-pixel state"); return PT_ERR_OOM;
                }
                sc->chunk_workspace_bytes = states_bytes + words_bytes;
            }
            unsigned* words = reinterpret_cast<unsigned*>(static_cast<char*>(sc->chunk_workspace) + states_bytes);
            if (!hip_ok(hipMemsetAsync(words, 0, words_bytes, st), "hipMemsetAsync(chunk words)")) return PT_ERR_HIP;
            launch_chunks(launch_variant(sc, ra), sc->view, ra, out, words, static_cast<unsigned long long*>(sc->chunk_workspace), sc->bad_resume_dev, st);
        } else if (ra.n_wave_tiles > 0) launch_render(launch_variant(sc, ra), sc->view, ra, out, slot_counter, nullptr, sc->bad_resume_dev, st);
        if (!hip_ok(hipGetLastError(), "render kernel launch")) return PT_ERR_HIP;
        sc->last.n_launches = ra.n_wave_tiles > 0 ? 1 : 0;
        sc->last.pipeline = PT_PIPELINE_FUSED;
        sc->last.schedule = ra.refill == 2 ? PT_SCHEDULE_WALK : ra.refill ? PT_SCHEDULE_REFILL : ra.n_chunks > 0 ? PT_SCHEDULE_CHUNKS : PT_SCHEDULE_TILES;
        sc->last.signature_walk = sc->variant == V_CORNELL_WORLD ? 1 : 0;
        sc->last.n_chunks = ra.n_chunks;
    }
    if (capturing) { sc->last = before_capture; return PT_OK; }
    if (!hip_ok(hipEventRecord(sc->ev_k1, st), "hipEventRecord")) return PT_ERR_HIP;
    if (!hip_ok(hipEventRecord(sc->ev_end, st), "hipEventRecord")) return PT_ERR_HIP;
    sc->have_timing = true;
    sc->last.paths = owned_pixels(p, ra) * (uint64_t)p->ns;
    return PT_OK;
}

extern "C" int pt_render_device(PtScene* sc, const PtRenderParams* p, void* d_out_rgb, void* hip_stream)
{
    if (!sc || !p || !d_out_rgb) { pt::set_error("pt_render_device: null argument"); return PT_ERR_INVALID; }
    return render_device(sc, p, d_out_rgb, hip_stream, sc->next_slot, false);
}

// pt_render_device for callers that cannot allocate device memory (see include/ptcore.h).  Calls no
// hipMalloc / hipFree: the slot counter is this GPU's static scratch, whose address was looked up
// when the scene was created.
extern "C" int pt_render_device_noalloc(PtScene* sc, const PtRenderParams* p, void* d_out_rgb, void* hip_stream)
{
    if (!sc || !p || !d_out_rgb) { pt::set_error("pt_render_device_noalloc: null argument"); return PT_ERR_INVALID; }
    // the scratch holds one 32-bit slot counter: 64 slots per 8x8 wave tile, so at most 2^25 wave tiles per call
    const int64_t tw = p->tile_w > 0 ? p->tile_w : 8, th = p->tile_h > 0 ? p->tile_h : 8;
    if (p->nx > 0 && p->ny > 0) {
        const int64_t tiles = ((p->nx + tw - 1) / tw) * ((p->ny + th - 1) / th);
        const int64_t shards = p->shard_count > 0 ? p->shard_count : 1;
        const int64_t wave_tiles = ((tiles + shards - 1) / shards) * ((tw + 7) / 8) * ((th + 7) / 8);
        if (wave_tiles > PT_NOALLOC_MAX_WAVE_TILES) {
            pt::set_error("pt_render_device_noalloc: the shard has more than PT_NOALLOC_MAX_WAVE_TILES (2^25) 8x8 wave tiles; "
                          "the static slot counter cannot number its pixel slots");
            return PT_ERR_INVALID;
        }
    }
    int current = -1;
    if (!hip_ok(hipGetDevice(&current), "hipGetDevice")) return PT_ERR_HIP;
    if (current != sc->device) { pt::set_error("pt_render_device_noalloc: the scene's device must be the current device"); return PT_ERR_INVALID; }
    unsigned* counter = scratch_on_device(sc->device);       // stored at pt_scene_create; no lookup happens here
    if (!counter) return PT_ERR_HIP;
    return render_device(sc, p, d_out_rgb, hip_stream, counter, true);
}

__global__ __launch_bounds__(256) void k_frame_finish(float* __restrict__ frame, size_t n, float inv_ns)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) frame[i] = sqrtf(fmaxf(0.0f, frame[i] * inv_ns));
}

extern "C" int pt_frame_finish_device(void* d_frame, size_t n_floats, int32_t total_samples, void* hip_stream)
{
    if (!d_frame || total_samples <= 0) { pt::set_error("pt_frame_finish_device: bad arguments"); return PT_ERR_INVALID; }
    if (n_floats == 0) return PT_OK;
    const float inv_ns = 1 / (float)total_samples;
    hipLaunchKernelGGL(k_frame_finish, dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                       static_cast<float*>(d_frame), n_floats, inv_ns);
    if (!hip_ok(hipGetLastError(), "frame finish launch")) return PT_ERR_HIP;
    return PT_OK;
}

// A pass that resumed from an accumulator whose pixels do not hold `first_sample` samples (stale,
// zero-filled, or of another render) has raised the scene's flag; read once its launch has finished.
static bool bad_resume(PtScene* sc)
{
    if (!sc->bad_resume_host || !*(volatile int*)sc->bad_resume_host) return false;
    const int what = *sc->bad_resume_host;
    *sc->bad_resume_host = 0;
    if (what == 2) {
        pt::set_error("render: PT_SCHEDULE_CHUNKS gave up waiting for a tile's previous chunk; the frame is not valid");
        return true;
    }
    pt::set_error("pt_render_pass: the accumulator does not hold the state of first_sample samples for every pixel of this shard "
                  "(resume from the array the earlier passes of this render wrote)");
    return true;
}

extern "C" size_t pt_accum_bytes(int32_t nx, int32_t ny)
{
    if (nx <= 0 || ny <= 0) return 0;
    return (size_t)nx * (size_t)ny * sizeof(PtAccumPixel);
}

extern "C" int pt_render_pass(PtScene* sc, const PtRenderParams* p, void* d_accum, int32_t first_sample, int32_t n_samples,
                              void* d_out_rgb, void* hip_stream)
{
    if (!sc || !p || !d_accum || !d_out_rgb) { pt::set_error("pt_render_pass: null argument"); return PT_ERR_INVALID; }
    RenderArgs ra;
    if (!make_args(p, &ra)) return PT_ERR_INVALID;
    if (first_sample < 0 || n_samples <= 0 || (int64_t)first_sample + n_samples > (int64_t)p->ns) {
        pt::set_error("pt_render_pass: the pass must lie inside [0, ns)"); return PT_ERR_INVALID;
    }
    if (p->pipeline == PT_PIPELINE_WAVEFRONT) { pt::set_error("pt_render_pass: passes run on the fused pipeline"); return PT_ERR_UNSUPPORTED; }
    ra.ns = first_sample + n_samples;                     // the launch stops, and normalises, at the samples done so far
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    float* out = static_cast<float*>(d_out_rgb);
    if (sc->have_timing && !hip_ok(hipStreamWaitEvent(st, sc->ev_end, 0), "hipStreamWaitEvent")) return PT_ERR_HIP;
    if (!hip_ok(hipEventRecord(sc->ev_begin, st), "hipEventRecord")) return PT_ERR_HIP;
    if (!hip_ok(hipMemsetAsync(out, 0, (size_t)3 * p->nx * p->ny * sizeof(float), st), "hipMemsetAsync")) return PT_ERR_HIP;
    if (first_sample == 0 &&
        !hip_ok(hipMemsetAsync(d_accum, 0, pt_accum_bytes(p->nx, p->ny), st), "hipMemsetAsync(accumulators)")) return PT_ERR_HIP;
    if (!hip_ok(hipEventRecord(sc->ev_k0, st), "hipEventRecord")) return PT_ERR_HIP;
    sc->last = PtRenderStats{};
    sc->last.kernel_family = kernel_family(sc->variant);
    if (ra.n_wave_tiles > 0)
        launch_pass(launch_variant(sc, ra), sc->view, ra, static_cast<PtAccumPixel*>(d_accum), first_sample, out, sc->bad_resume_dev, st);
    if (!hip_ok(hipGetLastError(), "render pass launch")) return PT_ERR_HIP;
    sc->last.n_launches = ra.n_wave_tiles > 0 ? 1 : 0;
    sc->last.pipeline = PT_PIPELINE_FUSED;
    sc->last.schedule = PT_SCHEDULE_TILES;
    sc->last.signature_walk = sc->variant == V_CORNELL_WORLD ? 1 : 0;
    if (!hip_ok(hipEventRecord(sc->ev_k1, st), "hipEventRecord")) return PT_ERR_HIP;
    if (!hip_ok(hipEventRecord(sc->ev_end, st), "hipEventRecord")) return PT_ERR_HIP;
    sc->have_timing = true;
    sc->last.paths = owned_pixels(p, ra) * (uint64_t)n_samples;
    return PT_OK;
}

// Device memory behind the entry points that take host buffers: one block per scene, grown when a
// call needs more and kept until the scene is destroyed, so that a host rendering frame after frame
// through pt_render does not pay an allocation per call.
static void* scene_staging(PtScene* sc, size_t bytes)
{
    if (sc->staging_bytes >= bytes) return sc->staging;
    if (sc->staging) { (void)hipFree(sc->staging); sc->staging = nullptr; sc->staging_bytes = 0; }
    void* d = nullptr;
    if (!hip_ok(hipMalloc(&d, bytes), "hipMalloc(staging)")) return nullptr;
    sc->staging = d;
    sc->staging_bytes = bytes;
    return d;
}
static size_t aligned256(size_t n) { return (n + 255) & ~(size_t)255; }

extern "C" int pt_render_progressive(PtScene* sc, const PtRenderParams* p, float* out_rgb, int32_t pass_samples,
                                     PtProgressFn progress, void* user)
{
    if (!sc || !p || !out_rgb) { pt::set_error("pt_render_progressive: null argument"); return PT_ERR_INVALID; }
    if (p->nx <= 0 || p->ny <= 0 || p->ns <= 0) { pt::set_error("render: nx, ny, ns must be positive"); return PT_ERR_INVALID; }
    if (pass_samples <= 0) { pt::set_error("pt_render_progressive: pass_samples must be positive"); return PT_ERR_INVALID; }
    const size_t bytes = (size_t)3 * p->nx * p->ny * sizeof(float);
    char* block = static_cast<char*>(scene_staging(sc, aligned256(bytes) + pt_accum_bytes(p->nx, p->ny)));
    if (!block) return PT_ERR_OOM;
    void *d_out = block, *d_acc = block + aligned256(bytes);
    int rc = PT_OK;
    bool cancelled = false;
    for (int32_t first = 0; first < p->ns && rc == PT_OK && !cancelled; first += pass_samples) {
        const int32_t n = std::min(pass_samples, p->ns - first);
        rc = pt_render_pass(sc, p, d_acc, first, n, d_out, nullptr);
        if (rc == PT_OK && !hip_ok(hipStreamSynchronize(nullptr), "render pass (synchronize)")) rc = PT_ERR_HIP;
        if (rc == PT_OK && bad_resume(sc)) rc = PT_ERR_INVALID;
        if (rc == PT_OK && progress && progress(first + n, p->ns, user) != 0) cancelled = true;
    }
    // a cancelled render still hands back the estimate of the samples it finished
    if (rc == PT_OK && !hip_ok(hipMemcpy(out_rgb, d_out, bytes, hipMemcpyDeviceToHost), "hipMemcpy(frame)")) rc = PT_ERR_HIP;
    if (rc == PT_OK && cancelled) { pt::set_error("pt_render_progressive: cancelled by the progress callback"); return PT_ERR_CANCELLED; }
    return rc;
}

extern "C" int pt_render(PtScene* sc, const PtRenderParams* p, float* out_rgb)
{
    if (!sc || !p || !out_rgb) { pt::set_error("pt_render: null argument"); return PT_ERR_INVALID; }
    if (p->nx <= 0 || p->ny <= 0) { pt::set_error("render: nx, ny must be positive"); return PT_ERR_INVALID; }
    const size_t bytes = (size_t)3 * p->nx * p->ny * sizeof(float);
    void* d_out = scene_staging(sc, bytes);
    if (!d_out) return PT_ERR_OOM;
    int rc = pt_render_device(sc, p, d_out, nullptr);
    if (rc == PT_OK && !hip_ok(hipStreamSynchronize(nullptr), "render (synchronize)")) rc = PT_ERR_HIP;
    if (rc == PT_OK && bad_resume(sc)) rc = PT_ERR_HIP;
    if (rc == PT_OK && !hip_ok(hipMemcpy(out_rgb, d_out, bytes, hipMemcpyDeviceToHost), "hipMemcpy(frame)")) rc = PT_ERR_HIP;
    return rc;
}

extern "C" int pt_render_on_devices(const PtSceneDesc* desc, const PtRenderParams* p, const int32_t* device_ids, int32_t n_devices,
                                    float* out_rgb)
{
    if (!desc || !p || !out_rgb || !device_ids || n_devices <= 0) { pt::set_error("pt_render_on_devices: bad arguments"); return PT_ERR_INVALID; }
    if (p->nx <= 0 || p->ny <= 0) { pt::set_error("render: nx, ny must be positive"); return PT_ERR_INVALID; }
    if (p->shard_count > 1 || p->batch_count > 1) { pt::set_error("pt_render_on_devices: the call shards by itself"); return PT_ERR_INVALID; }
    int previous = 0;
    (void)hipGetDevice(&previous);
    const size_t n_floats = (size_t)3 * p->nx * p->ny;
    std::vector<PtScene*> scenes((size_t)n_devices, nullptr);
    std::vector<void*> frames((size_t)n_devices, nullptr);
    int rc = PT_OK;
    for (int g = 0; g < n_devices && rc == PT_OK; g++) {          // scene replicated on every device
        if (!hip_ok(hipSetDevice(device_ids[g]), "hipSetDevice")) { rc = PT_ERR_HIP; break; }
        rc = pt_scene_create(desc, &scenes[(size_t)g]);
        if (rc == PT_OK && !hip_ok(hipMalloc(&frames[(size_t)g], n_floats * sizeof(float)), "hipMalloc(frame)")) rc = PT_ERR_OOM;
    }
    PtRenderParams q = *p;
    q.shard_count = n_devices;
    if (q.tile_w <= 0) q.tile_w = 8;
    if (q.tile_h <= 0) q.tile_h = 8;
    for (int g = 0; g < n_devices && rc == PT_OK; g++) {          // launches return at once: the devices trace concurrently
        if (!hip_ok(hipSetDevice(device_ids[g]), "hipSetDevice")) { rc = PT_ERR_HIP; break; }
        q.shard_index = g;
        rc = pt_render_device(scenes[(size_t)g], &q, frames[(size_t)g], nullptr);
    }
    std::vector<float> part(rc == PT_OK ? n_floats : 0);
    if (rc == PT_OK) std::fill(out_rgb, out_rgb + n_floats, 0.0f);
    for (int g = 0; g < n_devices && rc == PT_OK; g++) {
        if (!hip_ok(hipSetDevice(device_ids[g]), "hipSetDevice") || !hip_ok(hipDeviceSynchronize(), "render (synchronize)") ||
            !hip_ok(hipMemcpy(part.data(), frames[(size_t)g], n_floats * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy(frame)")) { rc = PT_ERR_HIP; break; }
        if (bad_resume(scenes[(size_t)g])) { rc = PT_ERR_HIP; break; }     // a chunked launch that gave up waiting: the shard is not valid
        for (size_t i = 0; i < n_floats; i++) out_rgb[i] += part[i];      // shards are disjoint and zero elsewhere: x + 0 = x
    }
    for (int g = 0; g < n_devices; g++) {
        if (!scenes[(size_t)g] && !frames[(size_t)g]) continue;
        (void)hipSetDevice(device_ids[g]);
        if (frames[(size_t)g]) (void)hipFree(frames[(size_t)g]);
        pt_scene_destroy(scenes[(size_t)g]);
    }
    (void)hipSetDevice(previous);
    return rc;
}

extern "C" int pt_render_stats(PtScene* sc, PtRenderStats* stats)
{
    if (!sc || !stats) { pt::set_error("pt_render_stats: null argument"); return PT_ERR_INVALID; }
    if (!sc->have_timing) { pt::set_error("pt_render_stats: no render has run on this scene"); return PT_ERR_INVALID; }
    if (!hip_ok(hipEventSynchronize(sc->ev_end), "hipEventSynchronize")) return PT_ERR_HIP;
    if (bad_resume(sc)) return PT_ERR_INVALID;
    float k = 0, tot = 0;
    if (!hip_ok(hipEventElapsedTime(&k, sc->ev_k0, sc->ev_k1), "hipEventElapsedTime")) return PT_ERR_HIP;
    if (!hip_ok(hipEventElapsedTime(&tot, sc->ev_begin, sc->ev_end), "hipEventElapsedTime")) return PT_ERR_HIP;
    sc->last.kernel_ms = k;
    sc->last.total_ms = tot;
    *stats = sc->last;
    return PT_OK;
}

extern "C" int pt_render_stage_stats(PtScene* sc, PtStageStats* stats)
{
    if (!sc || !stats) { pt::set_error("pt_render_stage_stats: null argument"); return PT_ERR_INVALID; }
    if (!sc->have_stage_stats) { pt::set_error("pt_render_stage_stats: no render on the staged pipeline has run on this scene"); return PT_ERR_INVALID; }
    *stats = sc->stage_stats;
    return PT_OK;
}

extern "C" int pt_trace_closest(PtScene* sc, const PtRay* rays, size_t n, int32_t* prim_id, float* t)
{
    if (!sc || (n && (!rays || !prim_id || !t))) { pt::set_error("pt_trace_closest: null argument"); return PT_ERR_INVALID; }
    if (n == 0) return PT_OK;
    const size_t rays_bytes = aligned256(n * sizeof(PtRay)), ids_bytes = aligned256(n * sizeof(int32_t));
    char* block = static_cast<char*>(scene_staging(sc, rays_bytes + ids_bytes + n * sizeof(float)));
    if (!block) return PT_ERR_OOM;
    PtRay* d_rays = reinterpret_cast<PtRay*>(block);
    int32_t* d_ids = reinterpret_cast<int32_t*>(block + rays_bytes);
    float* d_t = reinterpret_cast<float*>(block + rays_bytes + ids_bytes);
    int rc = PT_OK;
    if (rc == PT_OK && !hip_ok(hipMemcpy(d_rays, rays, n * sizeof(PtRay), hipMemcpyHostToDevice), "hipMemcpy(rays)")) rc = PT_ERR_HIP;
    if (rc == PT_OK) {
        int variant = sc->variant;                      // the same rule as launch_variant, in waves of rays
        if (variant == V_CORNELL_WORLD && (n + 63) / 64 <= (size_t)FLAT_UP_TO_WAVES) variant = V_CORNELL_WORLD_FLAT;
        launch_trace(variant, sc->view, d_rays, n, d_ids, d_t, nullptr);
        if (!hip_ok(hipGetLastError(), "trace kernel launch") || !hip_ok(hipStreamSynchronize(nullptr), "trace (synchronize)")) rc = PT_ERR_HIP;
    }
    if (rc == PT_OK && (!hip_ok(hipMemcpy(prim_id, d_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost), "hipMemcpy(ids)") ||
                        !hip_ok(hipMemcpy(t, d_t, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy(t)"))) rc = PT_ERR_HIP;
    return rc;
}

extern "C" int pt_scene_lbvh_form(PtScene* sc)
{
    if (!sc) { pt::set_error("pt_scene_lbvh_form: null scene"); return PT_ERR_INVALID; }
    return !(sc->mask & ptd::F_LBVH) ? 0 : (sc->mask & ptd::F_LBVH4) ? 4 : 2;
}

extern "C" int pt_scene_lbvh_wide_nodes(PtScene* sc, void* nodes, size_t capacity_bytes)
{
    if (!sc) { pt::set_error("pt_scene_lbvh_wide_nodes: null scene"); return PT_ERR_INVALID; }
    if (!sc->view.lbvh4) return 0;
    size_t bytes = 0;                                         // the array was allocated for exactly its nodes
    if (!hip_ok(hipMemPtrGetInfo(const_cast<ptd::LbvhNode4*>(sc->view.lbvh4.p), &bytes), "hipMemPtrGetInfo(wide nodes)")) return PT_ERR_HIP;
    const size_t n = bytes / sizeof(ptd::LbvhNode4);
    if (nodes) {
        if (capacity_bytes < n * sizeof(ptd::LbvhNode4)) { pt::set_error("pt_scene_lbvh_wide_nodes: buffer too small"); return PT_ERR_INVALID; }
        if (!hip_ok(hipMemcpy(nodes, sc->view.lbvh4.p, n * sizeof(ptd::LbvhNode4), hipMemcpyDeviceToHost), "hipMemcpy(wide nodes)")) return PT_ERR_HIP;
    }
    return (int)n;
}

extern "C" int pt_scene_lbvh_nodes(PtScene* sc, void* nodes, size_t capacity_bytes)
{
    if (!sc) { pt::set_error("pt_scene_lbvh_nodes: null scene"); return PT_ERR_INVALID; }
    if (!sc->view.lbvh) return 0;
    const size_t bytes = (size_t)sc->lbvh_nodes * sizeof(LbvhNode);
    if (nodes) {
        if (capacity_bytes < bytes) { pt::set_error("pt_scene_lbvh_nodes: buffer too small"); return PT_ERR_INVALID; }
        if (!hip_ok(hipMemcpy(nodes, sc->view.lbvh.p, bytes, hipMemcpyDeviceToHost), "hipMemcpy(lbvh)")) return PT_ERR_HIP;
    }
    return sc->lbvh_nodes;
}
