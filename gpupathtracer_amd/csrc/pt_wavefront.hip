// Staged ("wavefront") scheduler for the radiance stages.
//
// The reference traces a pixel's samples and bounces in nested loops inside one thread
// (render_kernel / render_pixel / color, src/ptMain.cu:95-190).  Here the same per-pixel state
// machine is cut at its divergent points and every stage runs as its own kernel over a compacted
// queue of pixel lanes:
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
#include <hip/hip_runtime.h>

#include <string>

#include "pt_host.h"
#include "pt_internal.h"

using namespace ptd;

namespace {

// Queue kinds for the shade stage (material-sorted).
enum : int { Q_LAMBERT = 0, Q_METAL = 1, Q_DIELECTRIC = 2, Q_ISOTROPIC = 3, N_SHADE_Q = 4 };
// counts[]: 0,1 = extend queue of each parity, 2.. = shade queues, then [C_DONE] finished lanes
enum : int { C_EXT0 = 0, C_EXT1 = 1, C_SHADE = 2, C_DONE = C_SHADE + N_SHADE_Q, N_COUNTS = C_DONE + 1 };

// One 128-byte record per pixel lane: a lane's whole state sits in one cache line, so a stage that
// visits lanes in queue order (scattered after compaction) still fetches whole lines.
struct alignas(128) LaneRec {
    uint32_t pix;         // pixel index i = ytop * nx + x
    int32_t  sample;      // samples finished
    int32_t  depth;       // bounces done on the current path
    int32_t  hit_face;
    uint64_t rng;         // PCG state (the increment is (i << 1) | 1)
    int32_t  hit_id;
    float    hit_t;
    float    ray[7];      // origin, direction, time
    int32_t  uv_id;       // stale-uv bookkeeping, read only by image textures
    float    thr[3];      // running product (accumCol of color())
    int32_t  uv_face;
    float    acc[3];      // pixel sum
    float    uv_t;
    float    pad[8];
};
static_assert(sizeof(LaneRec) == 128, "one cache line per lane");

struct WfState {
    LaneRec*  rec;
    uint32_t* q_ext[2];
    uint32_t* q_shade[N_SHADE_Q];
    unsigned* counts;
    uint32_t  n;          // lanes
};

__device__ __forceinline__ void queue_push(uint32_t* q, unsigned* count, bool pred, uint32_t value)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(mask));
    base = __shfl(base, leader);
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (pred) q[base + (unsigned)__popcll(mask & lower)] = value;
}

__device__ __forceinline__ Ray load_ray(const LaneRec& c)
{
    Ray r;
    r.o = mk(c.ray[0], c.ray[1], c.ray[2]);
    r.d = mk(c.ray[3], c.ray[4], c.ray[5]);
    r.time = c.ray[6];
    return r;
}
__device__ __forceinline__ void store_ray(LaneRec& c, const Ray& r)
{
    c.ray[0] = r.o.x; c.ray[1] = r.o.y; c.ray[2] = r.o.z;
    c.ray[3] = r.d.x; c.ray[4] = r.d.y; c.ray[5] = r.d.z;
    c.ray[6] = r.time;
}
__device__ __forceinline__ V3 load3(const float* a) { return mk(a[0], a[1], a[2]); }
__device__ __forceinline__ void store3(float* a, V3 v) { a[0] = v.x; a[1] = v.y; a[2] = v.z; }

// The path of lane l ended with running product thr: accumulate it (deNan, src/ptMain.cu:46-53,161)
// and start the pixel's next sample (src/ptMain.cu:156-160), or write the pixel when it was the last.
// Returns true when a fresh camera ray was stored and the lane must go to the next extend stage.
__device__ __forceinline__ bool end_path(const SceneView& sv, const RenderArgs& ra, LaneRec& c,
                                         V3 thr, Pcg& rng, float* __restrict__ out)
{
    V3 acc = load3(c.acc);
    int s = c.sample;
    const uint32_t i = c.pix;
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
            c.sample = s;
            return false;
        }
        float u = ((float)x + rng.rand()) / (float)ra.nx;
        float v = ((float)y + rng.rand()) / (float)ra.ny;
        const Ray r = camera_ray(sv, u, v, rng);
        thr = mk(1, 1, 1);
        if (ra.max_depth > 0) {
            store_ray(c, r);
            store3(c.thr, thr);
            store3(c.acc, acc);
            c.sample = s;
            c.depth = 0;
            return true;
        }
        // bounce limit 0: color() returns (1,1,1) without tracing; loop to the next sample
    }
}

__global__ __launch_bounds__(256) void k_wf_init(SceneView sv, RenderArgs ra, WfState w, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    int x = 0, ytop = 0;
    const bool live = wave < ra.n_wave_tiles && wave_pixel(ra, wave, lane, x, ytop);
    // lanes are numbered densely in launch order; dead lanes (image edge) get no slot
    const unsigned long long mask = __ballot(live);
    unsigned base = 0;
    if (mask != 0) {
        const int leader = __ffsll((long long)mask) - 1;
        if (lane == leader) base = atomicAdd(&w.counts[C_DONE + 1], (unsigned)__popcll(mask));   // slot allocator
        base = __shfl(base, leader);
    }
    bool go = false;
    uint32_t l = 0;
    if (live) {
        const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        l = base + (unsigned)__popcll(mask & lower);
        const unsigned i = (unsigned)(ytop * ra.nx + x);
        LaneRec& c = w.rec[l];
        c.pix = i;
        Pcg rng;
        rng.seed(i);
        c.sample = -1;                                   // end_path's first increment makes it 0
        store3(c.acc, mk(0, 0, 0));
        // "end" a virtual empty path whose contribution is exactly zero: acc stays 0 + 0
        go = end_path(sv, ra, c, mk(0, 0, 0), rng, out);
        c.rng = rng.state;
        if (!go) atomicAdd(&w.counts[C_DONE], 1u);
    }
    queue_push(w.q_ext[0], &w.counts[C_EXT0], live && go, l);
}

template <uint32_t F>
__global__ __launch_bounds__(256) void k_wf_extend(SceneView sv, RenderArgs ra, WfState w, int parity, float* __restrict__ out)
{
    __shared__ int lds_stack[(F & (F_LBVH | F_REFBVH)) ? 4 * 64 * LBVH_LDS_STACK : 1];
    int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[(threadIdx.x >> 6) * 64 * LBVH_LDS_STACK + (threadIdx.x & 63)] : nullptr;
    const unsigned count = w.counts[C_EXT0 + parity];
    const unsigned stride = gridDim.x * blockDim.x;
    // whole waves iterate together so the ballots in queue_push see every lane of the wave
    for (unsigned base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < count; base += stride) {
        const unsigned idx = base + (threadIdx.x & 63);
        const bool active = idx < count;
        uint32_t l = 0;
        int kind = -1;
        bool again = false;
        if (active) {
            l = w.q_ext[parity][idx];
            LaneRec& c = w.rec[l];
            Ray ray = load_ray(c);
            Pcg rng;
            rng.state = c.rng;
            rng.inc = ((uint64_t)c.pix << 1u) | 1u;
            Hit h;
            V3 thr = load3(c.thr);
            bool ended = false;
            if (world_hit<F>(sv, ray, 0.001f, FLT_MAX, h, rng, my_stack)) {
                int mat;
                if ((F & F_MEDIUM) && h.id >= PT_ID_MEDIUM_BASE) mat = sv.media[h.id - PT_ID_MEDIUM_BASE].material;
                else mat = sv.shapes[h.id].material;
                const int mtype = sv.materials[mat].type;
                if (mtype == PT_MAT_DIFFUSE_LIGHT) {
                    ended = !shade<F>(sv, ray, h, thr, rng);          // emits and ends the path
                } else {
                    kind = (mtype == PT_MAT_METAL) ? Q_METAL : (mtype == PT_MAT_DIELECTRIC) ? Q_DIELECTRIC
                         : (mtype == PT_MAT_ISOTROPIC) ? Q_ISOTROPIC : Q_LAMBERT;
                    c.hit_t = h.t; c.hit_id = h.id; c.hit_face = h.face;
                    if (F & F_IMAGE) { c.uv_id = h.uv_id; c.uv_face = h.uv_face; c.uv_t = h.uv_t; }
                }
            } else {
                shade_miss<F>(sv, ray, thr);
                ended = true;
            }
            if (ended) {
                again = end_path(sv, ra, c, thr, rng, out);
                if (!again) atomicAdd(&w.counts[C_DONE], 1u);
            }
            c.rng = rng.state;
        }
        queue_push(w.q_ext[parity ^ 1], &w.counts[C_EXT0 + (parity ^ 1)], active && again, l);
        queue_push(w.q_shade[Q_LAMBERT], &w.counts[C_SHADE + Q_LAMBERT], kind == Q_LAMBERT, l);
        if (F & F_METAL) queue_push(w.q_shade[Q_METAL], &w.counts[C_SHADE + Q_METAL], kind == Q_METAL, l);
        if (F & F_DIELECTRIC) queue_push(w.q_shade[Q_DIELECTRIC], &w.counts[C_SHADE + Q_DIELECTRIC], kind == Q_DIELECTRIC, l);
        if (F & F_ISOTROPIC) queue_push(w.q_shade[Q_ISOTROPIC], &w.counts[C_SHADE + Q_ISOTROPIC], kind == Q_ISOTROPIC, l);
    }
}

// blockIdx.y selects the material queue: every wave shades one material kind.
template <uint32_t F>
__global__ __launch_bounds__(256) void k_wf_shade(SceneView sv, RenderArgs ra, WfState w, int parity, float* __restrict__ out)
{
    const int q = blockIdx.y;
    const unsigned count = w.counts[C_SHADE + q];
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < count; base += stride) {
        const unsigned idx = base + (threadIdx.x & 63);
        const bool active = idx < count;
        uint32_t l = 0;
        bool again = false;
        if (active) {
            l = w.q_shade[q][idx];
            LaneRec& c = w.rec[l];
            Ray ray = load_ray(c);
            Pcg rng;
            rng.state = c.rng;
            rng.inc = ((uint64_t)c.pix << 1u) | 1u;
            Hit h;
            h.t = c.hit_t; h.id = c.hit_id; h.face = c.hit_face;
            h.uv_id = PT_ID_MISS; h.uv_face = 0; h.uv_t = 0.0f;
            if (F & F_IMAGE) { h.uv_id = c.uv_id; h.uv_face = c.uv_face; h.uv_t = c.uv_t; }
            V3 thr = load3(c.thr);
            const bool cont = shade<F>(sv, ray, h, thr, rng);
            int depth = c.depth + 1;
            if (cont && depth < ra.max_depth) {
                store_ray(c, ray);
                store3(c.thr, thr);
                c.depth = depth;
                again = true;
            } else {
                // bounce limit reached (color() returns the running product) or the path ended
                again = end_path(sv, ra, c, thr, rng, out);
                if (!again) atomicAdd(&w.counts[C_DONE], 1u);
            }
            c.rng = rng.state;
        }
        queue_push(w.q_ext[parity ^ 1], &w.counts[C_EXT0 + (parity ^ 1)], active && again, l);
    }
}

// After a round: the consumed extend queue and all shade queues are empty again.
__global__ void k_wf_reset(unsigned* counts, int parity)
{
    if (threadIdx.x == 0) counts[C_EXT0 + parity] = 0;
    if (threadIdx.x >= 1 && threadIdx.x <= N_SHADE_Q) counts[C_SHADE + threadIdx.x - 1] = 0;
}

bool hip_ok2(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    std::string msg = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    pt::set_error(msg.c_str());
    (void)hipGetLastError();      // the runtime keeps the last error until it is read: do not let it surface in a later call
    return false;
}

template <uint32_t F>
void launch_round(const SceneView& sv, const RenderArgs& ra, const WfState& w, int parity, int n_shade_q, unsigned blocks,
                  float* out, hipStream_t st)
{
    hipLaunchKernelGGL(k_wf_extend<F>, dim3(blocks), dim3(256), 0, st, sv, ra, w, parity, out);
    hipLaunchKernelGGL(k_wf_shade<F>, dim3(blocks, n_shade_q), dim3(256), 0, st, sv, ra, w, parity, out);
    hipLaunchKernelGGL(k_wf_reset, dim3(1), dim3(64), 0, st, w.counts, parity);
}

void launch_round_variant(int variant, const SceneView& sv, const RenderArgs& ra, const WfState& w, int parity, int nq,
                          unsigned blocks, float* out, hipStream_t st)
{
    using namespace ptk;
    switch (variant) {
    case V_CORNELL: case V_CORNELL_WORLD: case V_CORNELL_WORLD_FLAT: launch_round<M_CORNELL>(sv, ra, w, parity, nq, blocks, out, st); break;
    case V_SPHERES: launch_round<M_SPHERES>(sv, ra, w, parity, nq, blocks, out, st); break;
    case V_SPHERES_LBVH: case V_SPHERES_LBVH4: launch_round<M_SPHERES_LBVH>(sv, ra, w, parity, nq, blocks, out, st); break;
    case V_TRIMESH: case V_TRIMESH4: launch_round<M_TRIMESH>(sv, ra, w, parity, nq, blocks, out, st); break;
    case V_ALL: case V_FINAL: launch_round<M_ALL>(sv, ra, w, parity, nq, blocks, out, st); break;
    default: launch_round<M_ALL_LBVH>(sv, ra, w, parity, nq, blocks, out, st); break;
    }
}

}  // namespace

namespace pt {

int render_wavefront(PtScene* sc, const RenderArgs& ra, float* d_out, hipStream_t st)
{
    const uint32_t n_slots = (uint32_t)ra.n_wave_tiles * 64u;
    if (n_slots == 0) return PT_OK;
    // one allocation: the lane records, then the queues and counters.  The scene keeps it between calls
    // (a host that renders frame after frame pays for it once); it grows when a larger shard comes along.
    const size_t bytes = (size_t)n_slots * sizeof(LaneRec) + (size_t)n_slots * (2 + N_SHADE_Q) * 4 + 64 * 4;
    if (sc->wf_workspace_bytes < bytes) {
        if (sc->wf_workspace) { (void)hipFree(sc->wf_workspace); sc->wf_workspace = nullptr; sc->wf_workspace_bytes = 0; }
        if (!hip_ok2(hipMalloc(&sc->wf_workspace, bytes), "hipMalloc(wavefront state)")) return PT_ERR_OOM;
        sc->wf_workspace_bytes = bytes;
    }
    char* p = static_cast<char*>(sc->wf_workspace);
    WfState w{};
    w.n = n_slots;
    auto take = [&](size_t nbytes) { char* r = p; p += nbytes; return r; };
    w.rec = reinterpret_cast<LaneRec*>(take((size_t)n_slots * sizeof(LaneRec)));
    w.q_ext[0] = reinterpret_cast<uint32_t*>(take((size_t)n_slots * 4));
    w.q_ext[1] = reinterpret_cast<uint32_t*>(take((size_t)n_slots * 4));
    for (int q = 0; q < N_SHADE_Q; q++) w.q_shade[q] = reinterpret_cast<uint32_t*>(take((size_t)n_slots * 4));
    w.counts = reinterpret_cast<unsigned*>(take(64 * 4));
    if (!hip_ok2(hipMemsetAsync(w.counts, 0, 64 * 4, st), "hipMemsetAsync(counts)")) return PT_ERR_HIP;

    hipLaunchKernelGGL(k_wf_init, dim3((ra.n_wave_tiles + 3) / 4), dim3(256), 0, st, sc->view, ra, w, d_out);
    if (!hip_ok2(hipGetLastError(), "wavefront init launch")) return PT_ERR_HIP;
    int launches = 1;

    // which shade queues can be non-empty for this scene
    int n_shade_q = 1;
    if (sc->mask & F_METAL) n_shade_q = Q_METAL + 1;
    if (sc->mask & F_DIELECTRIC) n_shade_q = Q_DIELECTRIC + 1;
    if (sc->mask & F_ISOTROPIC) n_shade_q = Q_ISOTROPIC + 1;
    const unsigned blocks = std::min<unsigned>((n_slots + 255) / 256, 4096u);

    unsigned host_counts[N_COUNTS + 1];
    int parity = 0;
    const int rounds_per_check = 32;
    while (true) {
        for (int r = 0; r < rounds_per_check; r++) {
            launch_round_variant(sc->variant, sc->view, ra, w, parity, n_shade_q, blocks, d_out, st);
            parity ^= 1;
            launches += 3;
        }
        if (!hip_ok2(hipGetLastError(), "wavefront round launch")) return PT_ERR_HIP;
        if (!hip_ok2(hipMemcpyAsync(host_counts, w.counts, sizeof host_counts, hipMemcpyDeviceToHost, st), "hipMemcpyAsync(counts)")) return PT_ERR_HIP;
        if (!hip_ok2(hipStreamSynchronize(st), "wavefront rounds")) return PT_ERR_HIP;
        if (host_counts[C_DONE] >= host_counts[C_DONE + 1]) break;     // every allocated lane finished its pixel
    }
    sc->last.n_launches = launches;
    return PT_OK;
}

}  // namespace pt
