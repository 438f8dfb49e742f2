// Staged ("wavefront") scheduler for the radiance stages.
//
// The reference traces a pixel's samples and bounces in nested loops inside one thread
// (render_kernel / render_pixel / color, src/ptMain.cu:95-190).  Here the same per-pixel state
// machine is cut at its divergent points and every stage runs over a compacted queue of pixel lanes:
//
//   extend   : world->hit for every lane with a live ray; misses and light hits end the path in
//              place (accumulate, start the pixel's next sample), other hits are binned by
//              material kind
//   shade    : one pass per material kind over its queue (lanes of a wave share the material),
//              light / cosine mixture sampling, throughput update, next ray
//
// Queues hold lane numbers; a stage appends with one atomic per wave: the wave's ballot gives the
// count, the popcount of lower lanes gives each lane its slot.  Each lane still owns one pixel and
// carries that pixel's PCG state from sample to sample and bounce to bounce, so the image is the
// same, bit for bit, as the fused scheduler's and the reference's.
//
// Lane state is held field by field (structure of arrays, 16-byte elements where a stage reads a
// group together), so a stage moves only what it reads or changes: the search reads the ray (32
// bytes) and writes the hit (16), the shading reads ray, hit, running product and random state and
// writes ray, product and state.  One launch is one round of both stages, skewed by a round: the
// search of round r fills the shade queues that round r + 1 empties, and both feed the extend queue
// of round r + 1, so the two stages of a launch work on different lanes and nothing in a launch
// waits for anything else in it.  Counters rotate through three sets; the set nobody reads or
// appends to in a round is cleared by that round's first workgroup, so a round is exactly one
// launch.  The host enqueues rounds in chunks and reads the finished-lane count of a chunk while the
// next one is already queued.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "pt_host.h"
#include "pt_internal.h"

using namespace ptd;

namespace {

// Queue kinds for the shade stage (material-sorted).
enum : int { Q_LAMBERT = 0, Q_METAL = 1, Q_DIELECTRIC = 2, Q_ISOTROPIC = 3, N_SHADE_Q = 4 };
// counts[]: three sets of the extend queue's length, three sets of the shade queues' lengths, then the
// finished lanes and the allocated lanes.  Every counter has a 128-byte line to itself (C_STRIDE words
// apart): an atomic on a counter is served where the line lives, one at a time -- about 22 ns each on
// MI355X, whichever word of the line it names.
enum : int { C_EXT = 0, C_SHADE = 3, C_DONE = C_SHADE + 3 * N_SHADE_Q, C_ALLOC = C_DONE + 1, N_COUNTS = C_ALLOC + 1, C_STRIDE = 32 };

struct WfState {
    float4*   ray_o;      // origin, time
    float4*   ray_d;      // direction (w unused)
    int4*     hit;        // bits of t, id, face (signature worlds: 8 * entry + face), unused
    int4*     uv;         // scenes with image textures: the earlier uv-writing hit (id, face, bits of t)
    float4*   thr;        // running product of the path (accumCol of color()), bounces done (integer bits)
    float4*   acc;        // pixel sum, samples finished (integer bits)
    uint64_t* rng;        // PCG state (the increment is (pixel << 1) | 1)
    uint32_t* pix;        // pixel index i = ytop * nx + x
    uint32_t* q_ext[3];
    uint32_t* q_shade[2][N_SHADE_Q];
    unsigned* counts;
};

// Which sets a round reads, appends to and clears.
struct RoundSets {
    int e_in, e_out, e_zero;        // extend queue: consumed, produced (by both stages), cleared
    int s_in, s_out, s_zero;        // shade queue lengths: consumed by the shade stage, produced by the search, cleared
    int s_in_arr, s_out_arr;        // shade queue arrays (two sets)
};

// Queue appends are staged per workgroup: a workgroup takes WF_CHUNK (or fewer) consecutive entries of its
// input queue, collects what its lanes append in LDS -- a wave's ballot gives the count, one LDS atomic by
// its first lane the wave's place, the popcount of lower lanes each lane's slot -- and reserves room in
// each output queue with ONE global atomic per chunk before copying the staged entries out together.  With
// one global atomic per wave and queue, a round over a million lanes spent 450 us on 20 000 atomics to
// the same few lines and 40 us on everything else.
constexpr int WF_CHUNK = 1024;              // most entries a workgroup stages between two flushes
constexpr unsigned WF_TARGET_CHUNKS = 1024; // a queue is cut into about this many chunks (of 256 .. WF_CHUNK entries)
template <int ROWS>
struct Staging {
    uint32_t buf[ROWS][WF_CHUNK];
    unsigned n[ROWS], base[ROWS];
    unsigned done;
};
__device__ __forceinline__ unsigned chunk_size(unsigned count)
{
    const unsigned per = (count + WF_TARGET_CHUNKS - 1) / WF_TARGET_CHUNKS;
    return std::min<unsigned>(WF_CHUNK, std::max(256u, (per + 255u) & ~255u));
}
template <int ROWS>
__device__ __forceinline__ void stage_reset(Staging<ROWS>& s)
{
    if (threadIdx.x < ROWS) s.n[threadIdx.x] = 0;
    if (threadIdx.x == 0) s.done = 0;
    __syncthreads();
}
template <int ROWS>
__device__ __forceinline__ void stage_push(Staging<ROWS>& s, int row, bool pred, uint32_t value)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(&s.n[row], (unsigned)__popcll(mask));
    base = __shfl(base, leader);
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (pred) s.buf[row][base + (unsigned)__popcll(mask & lower)] = value;
}
// Row 0 goes to extend queue e_out, row r >= 1 to material queue r - 1 of the shade set (s_out, s_out_arr);
// the workgroup's tally of lanes that left the render goes to counter `tally`.
template <int ROWS>
__device__ __forceinline__ void stage_flush(Staging<ROWS>& s, const WfState& w, int rows_used, int e_out, int s_out, int s_out_arr, int tally)
{
    __syncthreads();
    const int t = (int)threadIdx.x;
    if (t < rows_used && s.n[t] > 0) {
        unsigned* c = (t == 0) ? &w.counts[C_STRIDE * (C_EXT + e_out)] : &w.counts[C_STRIDE * (C_SHADE + N_SHADE_Q * s_out + t - 1)];
        s.base[t] = atomicAdd(c, s.n[t]);
    }
    if (t == 64 && s.done > 0) atomicAdd(&w.counts[C_STRIDE * tally], s.done);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        if (r < rows_used) {
            uint32_t* q = (r == 0) ? w.q_ext[e_out] : w.q_shade[s_out_arr][r - 1];
            const unsigned n = s.n[r], base = s.base[r];
            for (unsigned k = threadIdx.x; k < n; k += blockDim.x) q[base + k] = s.buf[r][k];
        }
    }
    __syncthreads();
}
template <uint32_t F> struct StageRows {     // the next extend queue, then the material queues the mask can fill
    static constexpr int value = (F & F_ISOTROPIC) ? 2 + Q_ISOTROPIC : (F & F_DIELECTRIC) ? 2 + Q_DIELECTRIC : (F & F_METAL) ? 2 + Q_METAL : 2;
};

// Signature worlds shade from the face table (shade_sig); every workgroup keeps a copy in LDS.
template <uint32_t F, class W> struct WfFaces { static constexpr bool used = false; static constexpr int float4s = 1; };
template <uint32_t F, int... K> struct WfFaces<F, WorldSig<K...>> {
    static constexpr bool used = (F & ~ptk::M_CORNELL) == 0;
    static constexpr int float4s = used ? (int)sizeof...(K) * 8 * (SIG_FACE / 4) : 1;
};

// The path of lane l ended with running product thr: accumulate it (deNan, src/ptMain.cu:46-53,161)
// and start the pixel's next sample (src/ptMain.cu:156-160), or write the pixel when it was the last.
// Returns true when a fresh camera ray was stored and the lane must go to the next extend stage.
__device__ __forceinline__ bool end_path(const SceneView& sv, const RenderArgs& ra, const WfState& w, uint32_t l, uint32_t i,
                                         V3 acc, int s, V3 thr, Pcg& rng, float* __restrict__ out)
{
    const int x = (int)(i % (uint32_t)ra.nx), ytop = (int)(i / (uint32_t)ra.nx);
    const int y = ra.ny - 1 - ytop;
    while (true) {
        if (!(thr.x == thr.x)) thr.x = 0;
        if (!(thr.y == thr.y)) thr.y = 0;
        if (!(thr.z == thr.z)) thr.z = 0;
        acc = acc + thr;
        s++;
        if (s >= ra.ns) {
            const float inv_ns = 1 / (float)ra.ns;
            out[3 * (size_t)i + 0] = sqrtf(fmaxf(0.0f, acc.x * inv_ns));
            out[3 * (size_t)i + 1] = sqrtf(fmaxf(0.0f, acc.y * inv_ns));
            out[3 * (size_t)i + 2] = sqrtf(fmaxf(0.0f, acc.z * inv_ns));
            return false;
        }
        float u = ((float)x + rng.rand()) / (float)ra.nx;
        float v = ((float)y + rng.rand()) / (float)ra.ny;
        const Ray r = camera_ray(sv, u, v, rng);
        thr = mk(1, 1, 1);
        if (ra.max_depth > 0) {
            w.ray_o[l] = make_float4(r.o.x, r.o.y, r.o.z, r.time);
            w.ray_d[l] = make_float4(r.d.x, r.d.y, r.d.z, 0.0f);
            w.thr[l] = make_float4(1, 1, 1, __int_as_float(0));
            w.acc[l] = make_float4(acc.x, acc.y, acc.z, __int_as_float(s));
            return true;
        }
        // bounce limit 0: color() returns (1,1,1) without tracing; loop to the next sample
    }
}

__global__ __launch_bounds__(256) void k_wf_init(SceneView sv, RenderArgs ra, WfState w, float* __restrict__ out)
{
    __shared__ Staging<1> stage;
    stage_reset(stage);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    int x = 0, ytop = 0;
    const bool live = wave < ra.n_wave_tiles && wave_pixel(ra, wave, lane, x, ytop);
    // lane number = place in the launch; dead lanes (image edge) are never queued.  The workgroup's
    // tally here is of the lanes that own a pixel and did not finish it on the spot.
    const uint32_t l = (uint32_t)wave * 64u + (uint32_t)lane;
    bool go = false;
    if (live) {
        const unsigned i = (unsigned)(ytop * ra.nx + x);
        w.pix[l] = i;
        Pcg rng;
        rng.seed(i);
        // "end" a virtual empty path whose contribution is exactly zero: the sum stays 0 + 0, and the
        // sample count goes from -1 to 0
        go = end_path(sv, ra, w, l, i, mk(0, 0, 0), -1, mk(0, 0, 0), rng, out);
        w.rng[l] = rng.state;
        if (go) atomicAdd(&stage.done, 1u);
    }
    stage_push(stage, 0, live && go, l);
    stage_flush(stage, w, 1, 0, 0, 0, C_ALLOC);
}

// The search: world->hit for the lanes of the extend queue.
template <uint32_t F, class W>
__device__ __forceinline__ void extend_stage(const SceneView& sv, const RenderArgs& ra, const WfState& w, const RoundSets& rs,
                                             float* __restrict__ out, int* my_stack, const float4* faces, Staging<StageRows<F>::value>& stage,
                                             unsigned block, unsigned n_blocks)
{
    const unsigned count = w.counts[C_STRIDE * (C_EXT + rs.e_in)];
    const unsigned chunk = chunk_size(count);
    const unsigned n_chunks = (count + chunk - 1) / chunk;
    // every loop bound is the same for all lanes of the workgroup: the staging calls hold barriers
    for (unsigned c = block; c < n_chunks; c += n_blocks) {
      stage_reset(stage);
      for (unsigned it = 0; it < chunk; it += blockDim.x) {
        const unsigned idx = c * chunk + it + threadIdx.x;
        const bool active = idx < count;
        uint32_t l = 0;
        int kind = -1;
        bool again = false;
        if (active) {
            l = w.q_ext[rs.e_in][idx];
            const float4 o4 = w.ray_o[l], d4 = w.ray_d[l];
            Ray ray;
            ray.o = mk(o4.x, o4.y, o4.z); ray.d = mk(d4.x, d4.y, d4.z); ray.time = o4.w;
            // only a medium draws a random number inside the search; other worlds fetch the lane's
            // random state when its path ends here
            Pcg rng;
            rng.state = 0; rng.inc = 1;
            uint32_t i = 0;
            if (F & F_MEDIUM) { i = w.pix[l]; rng.state = w.rng[l]; rng.inc = ((uint64_t)i << 1u) | 1u; }
            Hit h;
            bool ended = false, lamp = false;
            if (world_hit<F, W>(sv, ray, 0.001f, FLT_MAX, h, rng, my_stack)) {
                int mtype;
                if constexpr (WfFaces<F, W>::used) {
                    mtype = __float_as_int(faces[(SIG_FACE / 4) * h.sig_code + 3].w);
                } else {
                    int mat;
                    if ((F & F_MEDIUM) && h.id >= PT_ID_MEDIUM_BASE) mat = sv.media[h.id - PT_ID_MEDIUM_BASE].material;
                    else mat = sv.shapes[h.id].material;
                    mtype = sv.materials[mat].type;
                }
                if (mtype == PT_MAT_DIFFUSE_LIGHT) {
                    ended = true; lamp = true;                        // emits and ends the path
                } else {
                    kind = (mtype == PT_MAT_METAL) ? Q_METAL : (mtype == PT_MAT_DIELECTRIC) ? Q_DIELECTRIC
                         : (mtype == PT_MAT_ISOTROPIC) ? Q_ISOTROPIC : Q_LAMBERT;
                    w.hit[l] = make_int4(__float_as_int(h.t), h.id, WfFaces<F, W>::used ? h.sig_code : h.face, 0);
                    if (F & F_IMAGE) w.uv[l] = make_int4(h.uv_id, h.uv_face, __float_as_int(h.uv_t), 0);
                }
            } else {
                ended = true;
            }
            if (ended) {
                if (!(F & F_MEDIUM)) { i = w.pix[l]; rng.state = w.rng[l]; rng.inc = ((uint64_t)i << 1u) | 1u; }
                const float4 t4 = w.thr[l], a4 = w.acc[l];
                V3 thr = mk(t4.x, t4.y, t4.z);
                if (lamp) {
                    if constexpr (WfFaces<F, W>::used) (void)shade_sig<F, W>(sv, faces, ray, h, thr, rng);
                    else (void)shade<F>(sv, ray, h, thr, rng);
                } else {
                    shade_miss<F>(sv, ray, thr);
                }
                again = end_path(sv, ra, w, l, i, mk(a4.x, a4.y, a4.z), __float_as_int(a4.w), thr, rng, out);
                if (!again) atomicAdd(&stage.done, 1u);
                w.rng[l] = rng.state;
            } else if (F & F_MEDIUM) {
                w.rng[l] = rng.state;
            }
        }
        stage_push(stage, 0, active && again, l);
        stage_push(stage, 1 + Q_LAMBERT, kind == Q_LAMBERT, l);
        if (F & F_METAL) stage_push(stage, 1 + Q_METAL, kind == Q_METAL, l);
        if (F & F_DIELECTRIC) stage_push(stage, 1 + Q_DIELECTRIC, kind == Q_DIELECTRIC, l);
        if (F & F_ISOTROPIC) stage_push(stage, 1 + Q_ISOTROPIC, kind == Q_ISOTROPIC, l);
      }
      stage_flush(stage, w, StageRows<F>::value, rs.e_out, rs.s_out, rs.s_out_arr, C_DONE);
    }
}

// Shading of one material queue: every wave shades one material kind.
template <uint32_t F, class W>
__device__ __forceinline__ void shade_stage(const SceneView& sv, const RenderArgs& ra, const WfState& w, const RoundSets& rs, int q,
                                            float* __restrict__ out, const float4* faces, Staging<StageRows<F>::value>& stage,
                                            unsigned block, unsigned n_blocks)
{
    const unsigned count = w.counts[C_STRIDE * (C_SHADE + N_SHADE_Q * rs.s_in + q)];
    const unsigned chunk = chunk_size(count);
    const unsigned n_chunks = (count + chunk - 1) / chunk;
    const uint32_t* queue = w.q_shade[rs.s_in_arr][q];
    for (unsigned c = block; c < n_chunks; c += n_blocks) {
      stage_reset(stage);
      for (unsigned it = 0; it < chunk; it += blockDim.x) {
        const unsigned idx = c * chunk + it + threadIdx.x;
        const bool active = idx < count;
        uint32_t l = 0;
        bool again = false;
        if (active) {
            l = queue[idx];
            const float4 o4 = w.ray_o[l], d4 = w.ray_d[l], t4 = w.thr[l];
            const int4 h4 = w.hit[l];
            const uint32_t i = w.pix[l];
            Ray ray;
            ray.o = mk(o4.x, o4.y, o4.z); ray.d = mk(d4.x, d4.y, d4.z); ray.time = o4.w;
            Pcg rng;
            rng.state = w.rng[l];
            rng.inc = ((uint64_t)i << 1u) | 1u;
            Hit h;
            h.t = __int_as_float(h4.x); h.id = h4.y; h.face = WfFaces<F, W>::used ? (h4.z & 7) : h4.z; h.sig_code = h4.z;
            h.uv_id = PT_ID_MISS; h.uv_face = 0; h.uv_t = 0.0f;
            if (F & F_IMAGE) { const int4 u4 = w.uv[l]; h.uv_id = u4.x; h.uv_face = u4.y; h.uv_t = __int_as_float(u4.z); }
            V3 thr = mk(t4.x, t4.y, t4.z);
            bool cont;
            if constexpr (WfFaces<F, W>::used) cont = shade_sig<F, W>(sv, faces, ray, h, thr, rng);
            else cont = shade<F>(sv, ray, h, thr, rng);
            const int depth = __float_as_int(t4.w) + 1;
            if (cont && depth < ra.max_depth) {
                w.ray_o[l] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.time);
                w.ray_d[l] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f);
                w.thr[l] = make_float4(thr.x, thr.y, thr.z, __int_as_float(depth));
                again = true;
            } else {
                // bounce limit reached (color() returns the running product) or the path ended
                const float4 a4 = w.acc[l];
                again = end_path(sv, ra, w, l, i, mk(a4.x, a4.y, a4.z), __float_as_int(a4.w), thr, rng, out);
                if (!again) atomicAdd(&stage.done, 1u);
            }
            w.rng[l] = rng.state;
        }
        stage_push(stage, 0, active && again, l);
      }
      stage_flush(stage, w, 1, rs.e_out, 0, 0, C_DONE);
    }
}

// One round.  STAGES = 3: both stages in one launch, the workgroups with blockIdx.y == 0 search and
// those with blockIdx.y == 1 + q shade material queue q; 1: the search alone; 2: the shading alone
// (blockIdx.y = q), for worlds whose two stages want different register budgets.
template <uint32_t F, class W, int STAGES>
__global__ __launch_bounds__(256) void k_wf_round(SceneView sv, RenderArgs ra, WfState w, RoundSets rs, float* __restrict__ out)
{
    __shared__ int lds_stack[((STAGES & 1) && (F & (F_LBVH | F_REFBVH))) ? 4 * 64 * LBVH_LDS_STACK : 1];
    __shared__ float4 faces[WfFaces<F, W>::float4s];
    __shared__ Staging<StageRows<F>::value> stage;
    if constexpr (WfFaces<F, W>::used) {
        for (int k = threadIdx.x; k < WfFaces<F, W>::float4s; k += blockDim.x) faces[k] = reinterpret_cast<const float4*>(sv.sig_faces)[k];
        __syncthreads();
    }
    auto search = [&]() {
        if (blockIdx.x == 0 && threadIdx.x <= N_SHADE_Q) {           // the sets no stage of this round touches
            if (threadIdx.x == 0) w.counts[C_STRIDE * (C_EXT + rs.e_zero)] = 0;
            else w.counts[C_STRIDE * (C_SHADE + N_SHADE_Q * rs.s_zero + threadIdx.x - 1)] = 0;
        }
        int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[(threadIdx.x >> 6) * 64 * LBVH_LDS_STACK + (threadIdx.x & 63)] : nullptr;
        extend_stage<F, W>(sv, ra, w, rs, out, my_stack, faces, stage, blockIdx.x, gridDim.x);
    };
    if constexpr (STAGES == 1) {
        search();
    } else if constexpr (STAGES == 2) {
        shade_stage<F, W>(sv, ra, w, rs, (int)blockIdx.y, out, faces, stage, blockIdx.x, gridDim.x);
    } else {
        if (blockIdx.y == 0) search();
        else shade_stage<F, W>(sv, ra, w, rs, (int)blockIdx.y - 1, out, faces, stage, blockIdx.x, gridDim.x);
    }
}

bool hip_ok2(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    std::string msg = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    pt::set_error(msg.c_str());
    (void)hipGetLastError();      // the runtime keeps the last error until it is read: do not let it surface in a later call
    return false;
}

template <uint32_t F, class W, bool SPLIT>
void launch_round(const SceneView& sv, const RenderArgs& ra, const WfState& w, const RoundSets& rs, int n_shade_q, unsigned blocks,
                  float* out, hipStream_t st)
{
    if (SPLIT) {
        hipLaunchKernelGGL((k_wf_round<F, W, 1>), dim3(blocks), dim3(256), 0, st, sv, ra, w, rs, out);
        hipLaunchKernelGGL((k_wf_round<F, W, 2>), dim3(blocks, n_shade_q), dim3(256), 0, st, sv, ra, w, rs, out);
    } else {
        hipLaunchKernelGGL((k_wf_round<F, W, 3>), dim3(blocks, 1 + n_shade_q), dim3(256), 0, st, sv, ra, w, rs, out);
    }
}

// The feature mask of the kernels a scene variant runs on (what launch_round_variant instantiates): it,
// not the scene's own mask, decides which lane fields the kernels touch.
uint32_t variant_mask(int variant)
{
    using namespace ptk;
    switch (variant) {
    case V_CORNELL_WORLD: case V_CORNELL_WORLD_FLAT: case V_CORNELL: return M_CORNELL;
    case V_SPHERES: return M_SPHERES;
    case V_SPHERES_LBVH: case V_SPHERES_LBVH4: return M_SPHERES_LBVH;
    case V_TRIMESH: case V_TRIMESH4: return M_TRIMESH;
    case V_ALL: case V_FINAL: return M_ALL;
    default: return M_ALL_LBVH;
    }
}

void launch_round_variant(int variant, const SceneView& sv, const RenderArgs& ra, const WfState& w, const RoundSets& rs, int nq,
                          unsigned blocks, float* out, hipStream_t st)
{
    using namespace ptk;
    switch (variant) {
    case V_CORNELL_WORLD: case V_CORNELL_WORLD_FLAT: launch_round<M_CORNELL, CornellWorld, false>(sv, ra, w, rs, nq, blocks, out, st); break;
    case V_CORNELL: launch_round<M_CORNELL, AnyWorld, false>(sv, ra, w, rs, nq, blocks, out, st); break;
    case V_SPHERES: launch_round<M_SPHERES, AnyWorld, false>(sv, ra, w, rs, nq, blocks, out, st); break;
    case V_SPHERES_LBVH: case V_SPHERES_LBVH4: launch_round<M_SPHERES_LBVH, AnyWorld, false>(sv, ra, w, rs, nq, blocks, out, st); break;
    case V_TRIMESH: case V_TRIMESH4: launch_round<M_TRIMESH, AnyWorld, false>(sv, ra, w, rs, nq, blocks, out, st); break;
    case V_ALL: case V_FINAL: launch_round<M_ALL, AnyWorld, false>(sv, ra, w, rs, nq, blocks, out, st); break;
    default: launch_round<M_ALL_LBVH, AnyWorld, false>(sv, ra, w, rs, nq, blocks, out, st); break;
    }
}

}  // namespace

namespace pt {

int render_wavefront(PtScene* sc, const RenderArgs& ra, float* d_out, hipStream_t st)
{
    constexpr int HOST_SLOT = 64;          // words per pinned copy of the finished / allocated counters

    const uint32_t n_slots = (uint32_t)ra.n_wave_tiles * 64u;
    if (n_slots == 0) return PT_OK;
    // one allocation: the lane fields, then the queues and counters.  The scene keeps it between calls
    // (a host that renders frame after frame pays for it once); it grows when a larger shard comes along.
    const uint32_t kernel_mask = variant_mask(sc->variant);
    const bool image = (kernel_mask & F_IMAGE) != 0;      // the kernels compiled with image textures keep the uv field
    const size_t per_lane = 16 * (image ? 6 : 5) + 8 + 4 + 4 * (3 + 2 * N_SHADE_Q);
    const size_t counts_bytes = (size_t)N_COUNTS * C_STRIDE * 4;
    const size_t bytes = (size_t)n_slots * per_lane + counts_bytes;
    if (sc->wf_workspace_bytes < bytes) {
        if (sc->wf_workspace) { (void)hipFree(sc->wf_workspace); sc->wf_workspace = nullptr; sc->wf_workspace_bytes = 0; }
        if (!hip_ok2(hipMalloc(&sc->wf_workspace, bytes), "hipMalloc(wavefront state)")) return PT_ERR_OOM;
        sc->wf_workspace_bytes = bytes;
    }
    if (!sc->wf_host) {
        void* h = nullptr;
        if (!hip_ok2(hipHostMalloc(&h, 2 * HOST_SLOT * sizeof(unsigned), hipHostMallocDefault), "hipHostMalloc(wavefront counters)")) return PT_ERR_OOM;
        sc->wf_host = static_cast<unsigned*>(h);
        for (hipEvent_t& e : sc->wf_ev)
            if (!hip_ok2(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return PT_ERR_HIP;
    }
    char* p = static_cast<char*>(sc->wf_workspace);
    WfState w{};
    auto take = [&](size_t nbytes) { char* r = p; p += nbytes; return r; };
    w.ray_o = reinterpret_cast<float4*>(take((size_t)n_slots * 16));
    w.ray_d = reinterpret_cast<float4*>(take((size_t)n_slots * 16));
    w.hit = reinterpret_cast<int4*>(take((size_t)n_slots * 16));
    w.thr = reinterpret_cast<float4*>(take((size_t)n_slots * 16));
    w.acc = reinterpret_cast<float4*>(take((size_t)n_slots * 16));
    w.uv = image ? reinterpret_cast<int4*>(take((size_t)n_slots * 16)) : nullptr;
    w.rng = reinterpret_cast<uint64_t*>(take((size_t)n_slots * 8));
    w.pix = reinterpret_cast<uint32_t*>(take((size_t)n_slots * 4));
    for (int k = 0; k < 3; k++) w.q_ext[k] = reinterpret_cast<uint32_t*>(take((size_t)n_slots * 4));
    for (int a = 0; a < 2; a++)
        for (int q = 0; q < N_SHADE_Q; q++) w.q_shade[a][q] = reinterpret_cast<uint32_t*>(take((size_t)n_slots * 4));
    w.counts = reinterpret_cast<unsigned*>(take(counts_bytes));
    if (!hip_ok2(hipMemsetAsync(w.counts, 0, counts_bytes, st), "hipMemsetAsync(counts)")) return PT_ERR_HIP;

    hipLaunchKernelGGL(k_wf_init, dim3((ra.n_wave_tiles + 3) / 4), dim3(256), 0, st, sc->view, ra, w, d_out);
    if (!hip_ok2(hipGetLastError(), "wavefront init launch")) return PT_ERR_HIP;
    int launches = 1;

    // which shade queues can be non-empty for this scene
    int n_shade_q = 1;
    if (sc->mask & kernel_mask & F_METAL) n_shade_q = Q_METAL + 1;
    if (sc->mask & kernel_mask & F_DIELECTRIC) n_shade_q = Q_DIELECTRIC + 1;
    if (sc->mask & kernel_mask & F_ISOTROPIC) n_shade_q = Q_ISOTROPIC + 1;

    // Rounds are enqueued a chunk at a time; the finished-lane count of chunk c is copied out behind it
    // and read once chunk c + 1 is queued, so the GPU never waits for the host's decision.  The count
    // also bounds the lanes still alive, which sizes the next chunk's launches.
    const int rounds_per_chunk = 32;
    // a lane is searched in one round and shaded in the next, and a pixel is at most ns paths of
    // max_depth bounces plus the round that ends each: more rounds than that mean a lane was lost
    const long long round_limit = 2ll * ra.ns * ((long long)ra.max_depth + 2) + 4 * rounds_per_chunk;
    uint32_t live = n_slots;
    int round = 0;
    bool finished = false;
    for (int chunk = 0; !finished; chunk++) {
        // enough workgroups for every chunk the kernels will cut `live` queue entries into (chunk_size)
        const unsigned blocks = std::max(1u, std::min<unsigned>((live + 255) / 256, std::max<unsigned>(WF_TARGET_CHUNKS, (live + WF_CHUNK - 1) / WF_CHUNK)));
        for (int k = 0; k < rounds_per_chunk; k++, round++) {
            RoundSets rs;
            rs.e_in = round % 3; rs.e_out = (round + 1) % 3; rs.e_zero = (round + 2) % 3;
            rs.s_out = round % 3; rs.s_in = (round + 2) % 3; rs.s_zero = (round + 1) % 3;
            rs.s_out_arr = round & 1; rs.s_in_arr = (round + 1) & 1;
            launch_round_variant(sc->variant, sc->view, ra, w, rs, n_shade_q, blocks, d_out, st);
            launches += 1;
        }
        if (!hip_ok2(hipGetLastError(), "wavefront round launch")) return PT_ERR_HIP;
        unsigned* slot = sc->wf_host + HOST_SLOT * (chunk & 1);     // the two counters, C_STRIDE words apart
        if (!hip_ok2(hipMemcpyAsync(slot, w.counts + C_STRIDE * C_DONE, (C_STRIDE + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(counts)")) return PT_ERR_HIP;
        if (!hip_ok2(hipEventRecord(sc->wf_ev[chunk & 1], st), "hipEventRecord")) return PT_ERR_HIP;
        if (chunk >= 1) {
            const int prev = (chunk - 1) & 1;
            if (!hip_ok2(hipEventSynchronize(sc->wf_ev[prev]), "wavefront rounds")) return PT_ERR_HIP;
            const unsigned done = sc->wf_host[HOST_SLOT * prev], alloc = sc->wf_host[HOST_SLOT * prev + C_STRIDE];
            if (done >= alloc) finished = true;                  // every allocated lane finished its pixel
            else live = alloc - done;
        }
        if (!finished && round > round_limit) {
            (void)hipStreamSynchronize(st);
            pt::set_error("wavefront: the frame did not finish within the rounds its sample and bounce counts allow");
            return PT_ERR_HIP;
        }
    }
    if (!hip_ok2(hipStreamSynchronize(st), "wavefront rounds")) return PT_ERR_HIP;
    sc->last.n_launches = launches;
    return PT_OK;
}

}  // namespace pt
