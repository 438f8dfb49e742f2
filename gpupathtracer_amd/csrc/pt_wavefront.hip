// Staged ("wavefront") scheduler for the radiance stages.
//
// The reference traces a pixel's samples and bounces in nested loops inside one thread
// (render_kernel / render_pixel / color, src/ptMain.cu:95-190).  Here the same per-pixel state
// machine is cut at its divergent points and every stage runs over a compacted queue of pixel lanes:
//
//   generate : a pixel's next camera sample (at the start, and wherever a path ends)
//   extend   : world->hit for every lane with a live ray; misses and light hits end the path in
//              place (accumulate, generate), other hits are binned by material kind
//   shade    : one pass per material kind over its queue (lanes of a wave share the material),
//              light / cosine mixture sampling, throughput update, next ray
//
// Queues hold lane numbers; a stage appends with a wave ballot: the ballot gives the count, one LDS
// atomic by the wave's first lane the wave's place in the queue, the popcount of lower lanes each
// lane's slot.  Each lane still owns one pixel and carries that pixel's PCG state from sample to
// sample and bounce to bounce, so the image is the same, bit for bit, as the fused scheduler's and
// the reference's.
//
// Lanes are dealt to workgroups in groups of 256 or 512 consecutive lanes (4 or 8 of the 8x8 pixel
// tiles), and a group's queues live in its workgroup's LDS: the workgroup loops over rounds -- search
// everything in the extend queue, barrier, shade every material queue, barrier -- until its extend
// queue stays empty, so a frame is one launch and no queue counter is ever touched by two
// workgroups.  (Frame-wide queues in global memory, one launch per round, were built first: a
// device-scope atomic on a queue counter is served one at a time at about 22 ns on this part, so
// with one atomic per wave and queue a round over a million lanes spent 450 of its 490 us on 20 000
// atomics; staged per workgroup, one atomic per 1024 entries, the round took 120 us, and three
// quarters of that was lanes of a wave lying anywhere in the frame's state arrays after a few rounds
// of compaction.)  Lane state is held field by field in global memory (16-byte elements where a
// stage reads a group together), so a stage moves only what it reads or changes, and a group's
// lanes are neighbours in every array.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "pt_host.h"
#include "pt_internal.h"

using namespace ptd;

namespace {

// Queue kinds for the shade stage (material-sorted).
enum : int { Q_LAMBERT = 0, Q_METAL = 1, Q_DIELECTRIC = 2, Q_ISOTROPIC = 3, N_SHADE_Q = 4 };

constexpr int WF_GROUP_MAX = 512;          // most lanes of one group (its queues hold 16-bit lane numbers within the group)
constexpr int WF_GROUP_LDS = 256;          // lanes of a group whose state lives in LDS (LdsLanes)

struct WfState {
    float4*   ray_o;      // origin, time
    float4*   ray_d;      // direction (w unused)
    int4*     hit;        // bits of t, id, face (signature worlds: 8 * entry + face), unused
    int2*     hit2;       // kernels with their lanes in LDS: bits of t; 8 * entry + face (signature worlds) or id << 3 | face
    int4*     uv;         // kernels with image textures: the earlier uv-writing hit (id, face, bits of t)
    float4*   thr;        // running product of the path (accumCol of color()), bounces done (integer bits)
    float4*   acc;        // pixel sum, samples finished (integer bits)
    uint64_t* rng;        // PCG state (the increment is (pixel << 1) | 1)
    uint32_t* pix;        // pixel index i = ytop * nx + x
    int*      lost;       // raised by a workgroup that ran out of rounds with lanes still queued
    unsigned long long* stats;    // PtStageStats as 16 counters, summed over the launch's workgroups (see k_wf_groups)
    int32_t   group_lanes;    // lanes per group: a multiple of 256, at most WF_GROUP_MAX
    int32_t   round_limit;    // rounds after which a group gives up (a lane was lost: cannot happen, must not hang)
    int32_t   ns_end;         // the sample count at which a lane's work ends: the pixel's last sample, or the end of a chunk (k_wf_chunks)
    int32_t   chunk_samples, n_chunks;    // k_wf_chunks: samples per work item, items per group
    unsigned* words;          // k_wf_chunks: [0] the ticket, [1] [2] cost sum and count, [4 ..) a progress word per group, then a cost word per group
    unsigned long long* group_states;     // k_wf_chunks: a group's random-stream states and running sums between its chunks (WF_GROUP_STATE_WORDS each)
};

// The material queues a feature mask can fill.
template <uint32_t F> struct ShadeQueues {
    static constexpr int value = (F & F_ISOTROPIC) ? 1 + Q_ISOTROPIC : (F & F_DIELECTRIC) ? 1 + Q_DIELECTRIC : (F & F_METAL) ? 1 + Q_METAL : 1;
};

// A group's queues, in its workgroup's LDS.
template <int NQ, int CAP>
struct GroupQueues {
    uint16_t ext[2][CAP];               // the extend queue being read and the one being filled
    uint16_t shade[NQ][CAP];
    unsigned n_ext[2], n_shade[NQ];
};
// most lanes of a group of kernel F: the kernels that keep their lanes in LDS run groups of WF_GROUP_LDS lanes only
template <uint32_t F> struct GroupCap { static constexpr int value = ((F & (F_LBVH | F_REFBVH | F_IMAGE)) == 0) ? WF_GROUP_LDS : WF_GROUP_MAX; };

// Lane state of a group in its workgroup's LDS, for kernels small enough to afford it (23 KB a group):
// the stages then read and write a lane's fields where its queues live, and nothing of a path but the
// finished pixel ever goes to memory.  (With the fields in global memory the rectangle scene's kernel
// spent 63 % of its wave-cycles waiting on them and wrote 64 GB per 1024^2 x256 frame.)
template <bool ON> struct LdsLanes {};
template <> struct LdsLanes<true> {
    float4   ray_o[WF_GROUP_LDS], ray_d[WF_GROUP_LDS], thr[WF_GROUP_LDS], acc[WF_GROUP_LDS];
    int2     hit[WF_GROUP_LDS];
    uint64_t rng[WF_GROUP_LDS];
    uint32_t pix[WF_GROUP_LDS];
};
// Kernels without a tree (no traversal stack in LDS) and without image textures keep their lanes in LDS.
template <uint32_t F> struct lanes_in_lds { static constexpr bool value = (F & (F_LBVH | F_REFBVH | F_IMAGE)) == 0; };

// Every lane of the wave calls this together.
__device__ __forceinline__ void queue_push(uint16_t* q, unsigned* count, bool pred, uint32_t value)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(mask));
    base = __shfl(base, leader);
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (pred) q[base + (unsigned)__popcll(mask & lower)] = (uint16_t)value;
}

// Signature worlds shade from the face table (shade_sig); every workgroup keeps a copy in LDS.
template <uint32_t F, class W> struct WfFaces { static constexpr bool used = false; static constexpr int float4s = 1; };
template <uint32_t F, int... K> struct WfFaces<F, WorldSig<K...>> {
    static constexpr bool used = (F & ~ptk::M_CORNELL) == 0;
    static constexpr int float4s = used ? (int)sizeof...(K) * 8 * (SIG_FACE / 4) : 1;
};

// The path of lane l ended with running product thr: accumulate it (deNan, src/ptMain.cu:46-53,161)
// and start the pixel's next sample (src/ptMain.cu:156-160), or write the pixel when it was the last.
// Returns true when a fresh camera ray was stored and the lane must go to the next extend stage.
// `begin`: no path has ended, the lane starts its work item at sample s (nothing is accumulated first).  A lane whose
// item ends before the pixel's last sample (w.ns_end < ra.ns: a chunk) leaves its sum where the item's end collects it.
__device__ __forceinline__ bool end_path(const SceneView& sv, const RenderArgs& ra, const WfState& w, uint32_t l, uint32_t i,
                                         V3 acc, int s, V3 thr, Pcg& rng, float* __restrict__ out, bool begin = false)
{
    const int x = (int)(i % (uint32_t)ra.nx), ytop = (int)(i / (uint32_t)ra.nx);
    const int y = ra.ny - 1 - ytop;
    while (true) {
        if (!begin) {
            if (!(thr.x == thr.x)) thr.x = 0;
            if (!(thr.y == thr.y)) thr.y = 0;
            if (!(thr.z == thr.z)) thr.z = 0;
            acc = acc + thr;
            s++;
        }
        begin = false;
        if (s >= w.ns_end) {
            if (w.ns_end < ra.ns) {
                w.acc[l] = make_float4(acc.x, acc.y, acc.z, __int_as_float(s));
                return false;
            }
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

// The search: world->hit for the lanes of the group's extend queue `cur`.
template <uint32_t F, class W, bool STATS>
__device__ __forceinline__ void extend_stage(const SceneView& sv, const RenderArgs& ra, const WfState& w, float* __restrict__ out,
                                             int* my_stack, const float4* faces, GroupQueues<ShadeQueues<F>::value, GroupCap<F>::value>& gq, int cur, uint32_t first,
                                             unsigned* group_stats)
{
    const unsigned count = gq.n_ext[cur];
    if constexpr (STATS) { if (threadIdx.x == 0) { atomicAdd(&group_stats[1], 1u); atomicAdd(&group_stats[2], count); atomicAdd(&group_stats[3], (count + 63u) >> 6); } }
    // whole waves iterate together so the ballots in queue_push see every lane of the wave
    for (unsigned it = 0; it < count; it += blockDim.x) {
        const unsigned idx = it + threadIdx.x;
        const bool active = idx < count;
        uint32_t j = 0;
        int kind = -1;
        bool again = false;
        if (active) {
            j = gq.ext[cur][idx];
            const uint32_t l = first + j;
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
                    if constexpr (lanes_in_lds<F>::value) w.hit2[l] = make_int2(__float_as_int(h.t), WfFaces<F, W>::used ? h.sig_code : ((h.id << 3) | h.face));
                    else w.hit[l] = make_int4(__float_as_int(h.t), h.id, WfFaces<F, W>::used ? h.sig_code : h.face, 0);
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
                w.rng[l] = rng.state;
            } else if (F & F_MEDIUM) {
                w.rng[l] = rng.state;
            }
        }
        queue_push(gq.ext[cur ^ 1], &gq.n_ext[cur ^ 1], active && again, j);
        queue_push(gq.shade[Q_LAMBERT], &gq.n_shade[Q_LAMBERT], kind == Q_LAMBERT, j);
        if (F & F_METAL) queue_push(gq.shade[Q_METAL], &gq.n_shade[Q_METAL], kind == Q_METAL, j);
        if (F & F_DIELECTRIC) queue_push(gq.shade[Q_DIELECTRIC], &gq.n_shade[Q_DIELECTRIC], kind == Q_DIELECTRIC, j);
        if (F & F_ISOTROPIC) queue_push(gq.shade[Q_ISOTROPIC], &gq.n_shade[Q_ISOTROPIC], kind == Q_ISOTROPIC, j);
    }
}

// Shading of material queue q: every wave shades one material kind.
template <uint32_t F, class W, bool STATS>
__device__ __forceinline__ void shade_stage(const SceneView& sv, const RenderArgs& ra, const WfState& w, int q, float* __restrict__ out,
                                            const float4* faces, GroupQueues<ShadeQueues<F>::value, GroupCap<F>::value>& gq, int cur, uint32_t first, unsigned* group_stats)
{
    const unsigned count = gq.n_shade[q];
    if constexpr (STATS) { if (threadIdx.x == 0 && count != 0) { atomicAdd(&group_stats[4 + q], count); atomicAdd(&group_stats[8 + q], (count + 63u) >> 6); } }
    for (unsigned it = 0; it < count; it += blockDim.x) {
        const unsigned idx = it + threadIdx.x;
        const bool active = idx < count;
        uint32_t j = 0;
        bool again = false;
        if (active) {
            j = gq.shade[q][idx];
            const uint32_t l = first + j;
            const float4 o4 = w.ray_o[l], d4 = w.ray_d[l], t4 = w.thr[l];
            int4 h4;
            if constexpr (lanes_in_lds<F>::value) {
                const int2 h2 = w.hit2[l];
                h4 = WfFaces<F, W>::used ? make_int4(h2.x, 0, h2.y, 0) : make_int4(h2.x, h2.y >> 3, h2.y & 7, 0);
            } else {
                h4 = w.hit[l];
            }
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
            }
            w.rng[l] = rng.state;
        }
        queue_push(gq.ext[cur ^ 1], &gq.n_ext[cur ^ 1], active && again, j);
    }
}

// Workgroups per CU the kernel is compiled for (a register cap).  Measured on MI355X with groups of 256 /
// 512 / 1024 lanes (Mpaths/s): rectangle scene 1024^2 x256 at 4 workgroups 7 756 / 5 702 / 3 511, at 5
// 8 461 / 5 273 / 3 505, at 6 7 122 / 4 082 / 3 087, at 8 4 349 / 2 859 / 2 370; random spheres in the tree
// 1920x1080 x32 at 3 1 680 / 1 906 / -, at 4 2 036 / 2 162 / 2 052, at 5 1 241 / 1 428 / 1 550; RTNW final
// through the tree x16 at 3 1 041 / 1 233 / -, at 4 1 097 / 1 225 / 1 004, at 5 473 / 571 / 719; 2 M
// triangles x8 at 4 542 / 459 / 326 (all through the binary tree, before the 4-wide one).
__host__ __device__ constexpr int wf_min_blocks(uint32_t F)
{
#ifdef PT_WF_BLOCKS             // kernel experiments
    return PT_WF_BLOCKS;
#endif
    // 2 M triangles through the 4-wide tree x16: 4 / 5 / 6 = 188 / 147 / 177 ms.  Re-measured once the log tables' rows had left the
    // registers (pt_libm.h), 4 / 5 / 6: random spheres over the flat list 1920x1080 x16 104.2 / 85.2 / 97.8, final over its list with
    // the reference's BVH x64 183.1 / 175.3 / 202.2; through the trees no change (spheres x64 55.7 / 56.2 / 55.7, final 67.5 / 68.6 / 72.3).
    if (F == ptk::M_SPHERES || F == ptk::M_FINAL) return 5;
    return ((F & ~ptk::M_CORNELL) == 0 || (F & ~F_LBVH4) == ptk::M_TRIMESH) ? 5 : 4;
}
// One workgroup renders one group of lanes, round by round.  Every loop bound below is read from
// LDS after a barrier, so all lanes of the workgroup take the same trips and meet at every barrier.
// STATS: the kernel also counts what its stages did (PtRenderParams.stage_stats); a kernel of its own because the
// counting, light as it is, costs the rectangle scene's kernel 6 % of its rate.
template <uint32_t F, class W, bool STATS>
__global__ __launch_bounds__(256, wf_min_blocks(F)) void k_wf_groups(SceneView sv, RenderArgs ra, WfState w, float* __restrict__ out)
{
    constexpr int NQ = ShadeQueues<F>::value;
    __shared__ int lds_stack[(F & (F_LBVH | F_REFBVH)) ? 4 * 64 * LBVH_LDS_STACK : 1];
    __shared__ float4 faces[WfFaces<F, W>::float4s];
    __shared__ GroupQueues<NQ, GroupCap<F>::value> gq;
    __shared__ LdsLanes<lanes_in_lds<F>::value> lanes;
    if constexpr (lanes_in_lds<F>::value) {                   // the fields are this workgroup's own: lane numbers start at 0
        w.ray_o = lanes.ray_o; w.ray_d = lanes.ray_d; w.thr = lanes.thr; w.acc = lanes.acc; w.hit2 = lanes.hit;
        w.rng = lanes.rng; w.pix = lanes.pix;
    }
    if constexpr (WfFaces<F, W>::used) {
        for (int k = threadIdx.x; k < WfFaces<F, W>::float4s; k += blockDim.x) { const float* f = sv.sig_faces + 4 * k; faces[k] = make_float4(f[0], f[1], f[2], f[3]); }
    }
    if (threadIdx.x < 2) gq.n_ext[threadIdx.x] = 0;
    if (threadIdx.x < NQ) gq.n_shade[threadIdx.x] = 0;
    __syncthreads();
    int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[(threadIdx.x >> 6) * 64 * LBVH_LDS_STACK + (threadIdx.x & 63)] : nullptr;
    const uint32_t group_first = blockIdx.x * (uint32_t)w.group_lanes;      // the group's first lane of the launch
    const uint32_t first = lanes_in_lds<F>::value ? 0u : group_first;          // ... and in the state arrays

    // generate: the group's lanes are the pixels of consecutive 8x8 tiles; lanes off the image edge are never queued
    for (uint32_t j = threadIdx.x; j < (uint32_t)w.group_lanes; j += blockDim.x) {
        const uint32_t l = first + j;
        const int wave = (int)((group_first + j) >> 6), lane = (int)(j & 63);
        int x = 0, ytop = 0;
        const bool live = wave < ra.n_wave_tiles && wave_pixel(ra, wave, lane, x, ytop);
        bool go = false;
        if (live) {
            const unsigned i = (unsigned)(ytop * ra.nx + x);
            w.pix[l] = i;
            Pcg rng;
            rng.seed(i);
            go = end_path(sv, ra, w, l, i, mk(0, 0, 0), 0, mk(0, 0, 0), rng, out, true);
            w.rng[l] = rng.state;
        }
        queue_push(gq.ext[0], &gq.n_ext[0], live && go, j);
    }
    __syncthreads();

    // What the stages did, for pt_render_stage_stats (STATS kernels).  The group's sums live in LDS (the layout of
    // PtStageStats; 32 bits do: a group's rounds times its lanes stay far below 2^32) and are added to with LDS atomics
    // whose result nobody waits for, by the stages themselves where they read their queue's length; thread 0 adds them
    // to the launch's at the end.  The clock is read by one workgroup in 64 only: each read waits for the counter to
    // come back, and with every workgroup reading it three times a round the rectangle scene's kernel lost a quarter
    // of its rate.
    __shared__ unsigned group_stats[STATS ? 12 : 1];
    __shared__ unsigned long long group_ticks[STATS ? 3 : 1];   // extend, shade, the last reading
    if constexpr (STATS) {
        if (threadIdx.x < 12) group_stats[threadIdx.x] = 0;
        if (threadIdx.x < 3) group_ticks[threadIdx.x] = 0;
    }
    const bool timed = STATS && (blockIdx.x & 63u) == 0u;
    int cur = 0;
    for (int round = 0; ; round++) {
        if (gq.n_ext[cur] == 0) break;                          // every pixel of the group is written
        if (round >= w.round_limit) { if (threadIdx.x == 0) *w.lost = 1; break; }
        if (timed && threadIdx.x == 0) group_ticks[2] = clock64();
        extend_stage<F, W, STATS>(sv, ra, w, out, my_stack, faces, gq, cur, first, group_stats);
        __syncthreads();                                        // the material queues are complete
        if (timed && threadIdx.x == 0) { const unsigned long long now = clock64(); group_ticks[0] += now - group_ticks[2]; group_ticks[2] = now; }
        for (int q = 0; q < NQ; q++) shade_stage<F, W, STATS>(sv, ra, w, q, out, faces, gq, cur, first, group_stats);
        __syncthreads();                                        // the next extend queue is complete; everyone has read this round's counts
        if (threadIdx.x == 0) {
            if (timed) group_ticks[1] += clock64() - group_ticks[2];
            gq.n_ext[cur] = 0;
        }
        if (threadIdx.x < NQ) gq.n_shade[threadIdx.x] = 0;
        cur ^= 1;
        __syncthreads();
    }
    if constexpr (STATS) {
        if (threadIdx.x == 0) {
            atomicAdd(&w.stats[0], 1ull);
            for (int k = 1; k < 12; k++) if (group_stats[k]) atomicAdd(&w.stats[k], (unsigned long long)group_stats[k]);
            if (timed) { atomicAdd(&w.stats[12], group_ticks[0]); atomicAdd(&w.stats[13], group_ticks[1]); atomicAdd(&w.stats[14], 1ull); }
        }
    }
}


// k_wf_groups with every group's samples cut into chunks, for frames of few groups and many samples (the headline:
// 4 096 groups of 4 096 samples for 1 280 resident workgroups, 3.2 rounds: the launch ends with most of the GPU idle
// behind its last groups, and the same paths traced as a 4096^2 frame at 256 samples take 287 ms against 348).  The
// work item is (chunk, group), numbered chunk-major; persistent workgroups draw item numbers from a counter, so items
// start in order.  A pixel's samples stay one sequence (src/ptMain.cu:156-162): between chunks the lanes' random-stream
// states and running sums wait in `group_states` (WF_GROUP_STATE_WORDS 64-bit words a group: 256 states, then the sums'
// three components as rows of 256 floats), and chunk c of a group starts once the group's progress word says chunk
// c - 1 has been stored.  The wait cannot deadlock (the item waited for has a smaller number, so a running workgroup
// holds it, and chains of waits end at a chunk 0) and is bounded by progress all the same: a word that stands still
// for 2^23 polls raises *w.lost = 2.  Kernels whose lanes live in global memory (the tree families) keep the states
// where they are: a group's lanes have their own places in the state arrays whichever workgroup runs the chunk.
// The loop keeps the shape of k_render_chunks' (one errand of thread 0 per round, between two barriers): see there.
constexpr int WF_GROUP_STATE_WORDS = WF_GROUP_LDS + 3 * (WF_GROUP_LDS / 2);      // 5 120 bytes
template <uint32_t F, class W>
__global__ __launch_bounds__(256, wf_min_blocks(F)) void k_wf_chunks(SceneView sv, RenderArgs ra, WfState w, float* __restrict__ out)
{
    constexpr int NQ = ShadeQueues<F>::value;
    constexpr bool IN_LDS = lanes_in_lds<F>::value;
    __shared__ int lds_stack[(F & (F_LBVH | F_REFBVH)) ? 4 * 64 * LBVH_LDS_STACK : 1];
    __shared__ float4 faces[WfFaces<F, W>::float4s];
    __shared__ GroupQueues<NQ, GroupCap<F>::value> gq;
    __shared__ LdsLanes<IN_LDS> lanes;
    __shared__ unsigned s_item, s_prio;
    unsigned* const words = w.words;
    unsigned long long* const group_states = w.group_states;
    const int n_chunks = w.n_chunks, chunk_samples = w.chunk_samples;
    if constexpr (IN_LDS) {
        w.ray_o = lanes.ray_o; w.ray_d = lanes.ray_d; w.thr = lanes.thr; w.acc = lanes.acc; w.hit2 = lanes.hit;
        w.rng = lanes.rng; w.pix = lanes.pix;
    }
    int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[(threadIdx.x >> 6) * 64 * LBVH_LDS_STACK + (threadIdx.x & 63)] : nullptr;
    if constexpr (WfFaces<F, W>::used) {
        for (int k = threadIdx.x; k < WfFaces<F, W>::float4s; k += blockDim.x) { const float* f = sv.sig_faces + 4 * k; faces[k] = make_float4(f[0], f[1], f[2], f[3]); }
    }
    const unsigned groups = ((unsigned)ra.n_wave_tiles * 64u + (unsigned)w.group_lanes - 1u) / (unsigned)w.group_lanes;
    const unsigned n_items = groups * (unsigned)n_chunks;
    // words: [0] the ticket, [1] [2] sum and count of the groups' first-chunk costs, [4 ..) a progress word per group, then a
    // cost word per group: a group's first chunk is timed and its later chunks run at a wave priority that grows with that
    // time over the running mean, so that the expensive groups' chains of chunks are served first (as in k_render_chunks;
    // which wave issues first changes no value)
    unsigned* const progress = words + 4;
    unsigned* const cost = words + 4 + groups;
    unsigned long long item_began = 0;
    unsigned finished_cost = 0;
    bool finished_one = false;
    unsigned finished_grp = 0, finished_c = 0;
    while (true) {
        if (threadIdx.x == 0) {
            if (finished_one) {
                if (finished_c == 0) {
                    __hip_atomic_store(&cost[finished_grp], finished_cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicAdd(&words[1], finished_cost); atomicAdd(&words[2], 1u);
                }
                __hip_atomic_store(&progress[finished_grp], finished_c + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_prio = 0;
            s_item = atomicAdd(&words[0], 1u);
        }
        if (threadIdx.x < 2) gq.n_ext[threadIdx.x] = 0;
        if (threadIdx.x < NQ) gq.n_shade[threadIdx.x] = 0;
        __syncthreads();
        const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)s_item);
        if (item >= n_items) break;
        const int c = (int)(item / groups);
        const unsigned grp = item - (unsigned)c * groups;
        if (c > 0) {
            if (threadIdx.x == 0) {
                unsigned spins = 0, seen = 0;
                while (true) {                                    // relaxed polls, one acquire fence below (as k_render_chunks)
                    const unsigned now = __hip_atomic_load(&progress[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (now >= (unsigned)c) break;
                    if (now != seen) { seen = now; spins = 0; }
                    __builtin_amdgcn_s_sleep(64);
                    if (++spins > (1u << 23)) { *w.lost = 2; break; }
                }
                const unsigned long long mine = __hip_atomic_load(&cost[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long sum = __hip_atomic_load(&words[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long cnt = __hip_atomic_load(&words[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long tenths = sum ? 10ull * mine * cnt / sum : 10ull;
                s_prio = tenths >= 15 ? 3u : tenths >= 12 ? 2u : tenths >= 9 ? 1u : 0u;
            }
            __syncthreads();
            __threadfence();                                      // every wave: the loads below come after the word
        }
        {
            const unsigned prio = (unsigned)__builtin_amdgcn_readfirstlane((int)s_prio);
            if (prio == 3u) __builtin_amdgcn_s_setprio(3);
            else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
            else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        item_began = clock64();
        w.ns_end = min(ra.ns, (c + 1) * chunk_samples);
        const uint32_t group_first = grp * (uint32_t)w.group_lanes;     // the group's first lane of the launch
        const uint32_t first = IN_LDS ? 0u : group_first;                 // ... and in the state arrays
        unsigned long long* const st_rng = IN_LDS ? group_states + (size_t)grp * WF_GROUP_STATE_WORDS : nullptr;
        float* const st_sum = IN_LDS ? reinterpret_cast<float*>(st_rng + WF_GROUP_LDS) : nullptr;
        // generate.  A kernel whose lanes live in global memory finds the states its group's last chunk left where it left them.
        for (uint32_t j = threadIdx.x; j < (uint32_t)w.group_lanes; j += blockDim.x) {
            const uint32_t l = first + j;
            const int wave = (int)((group_first + j) >> 6), lane = (int)(j & 63);
            int x = 0, ytop = 0;
            const bool live = wave < ra.n_wave_tiles && wave_pixel(ra, wave, lane, x, ytop);
            bool go = false;
            if (live) {
                const unsigned i = (unsigned)(ytop * ra.nx + x);
                w.pix[l] = i;
                Pcg rng;
                rng.seed(i);
                V3 acc = mk(0, 0, 0);
                if (c > 0) {
                    if constexpr (IN_LDS) {
                        rng.state = st_rng[j];
                        acc = mk(st_sum[j], st_sum[WF_GROUP_LDS + j], st_sum[2 * WF_GROUP_LDS + j]);
                    } else {
                        rng.state = w.rng[l];
                        const float4 a4 = w.acc[l];
                        acc = mk(a4.x, a4.y, a4.z);
                    }
                }
                go = end_path(sv, ra, w, l, i, acc, c * chunk_samples, mk(0, 0, 0), rng, out, true);
                w.rng[l] = rng.state;
            }
            queue_push(gq.ext[0], &gq.n_ext[0], live && go, j);
        }
        __syncthreads();
        int cur = 0;
        for (int round = 0; ; round++) {
            if (gq.n_ext[cur] == 0) break;                          // every lane of the group has reached the item's last sample
            if (round >= w.round_limit) { if (threadIdx.x == 0) *w.lost = 1; break; }
            extend_stage<F, W, false>(sv, ra, w, out, my_stack, faces, gq, cur, first, nullptr);
            __syncthreads();
            for (int q = 0; q < NQ; q++) shade_stage<F, W, false>(sv, ra, w, q, out, faces, gq, cur, first, nullptr);
            __syncthreads();
            if (threadIdx.x == 0) gq.n_ext[cur] = 0;
            if (threadIdx.x < NQ) gq.n_shade[threadIdx.x] = 0;
            cur ^= 1;
            __syncthreads();
        }
        if (IN_LDS && c < n_chunks - 1) {                           // the lanes' states wait for the group's next chunk
            const uint32_t j = threadIdx.x;
            const int wave = (int)((group_first + j) >> 6), lane = (int)(j & 63);
            int x = 0, ytop = 0;
            if (wave < ra.n_wave_tiles && wave_pixel(ra, wave, lane, x, ytop)) {
                const float4 a4 = w.acc[j];
                st_rng[j] = w.rng[j];
                st_sum[j] = a4.x; st_sum[WF_GROUP_LDS + j] = a4.y; st_sum[2 * WF_GROUP_LDS + j] = a4.z;
            }
        }
        __threadfence();                                          // every lane's stores before the word
        __syncthreads();
        finished_one = true; finished_grp = grp; finished_c = (unsigned)c;
        finished_cost = (unsigned)min((unsigned long long)0xffffffffu, (clock64() - item_began) >> 6);
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

template <uint32_t F, class W>
void launch_groups(const SceneView& sv, const RenderArgs& ra, const WfState& w, unsigned groups, float* out, hipStream_t st)
{
    if (ra.stage_stats) hipLaunchKernelGGL((k_wf_groups<F, W, true>), dim3(groups), dim3(256), 0, st, sv, ra, w, out);
    else hipLaunchKernelGGL((k_wf_groups<F, W, false>), dim3(groups), dim3(256), 0, st, sv, ra, w, out);
}

template <uint32_t F, class W>
void launch_chunks(const SceneView& sv, const RenderArgs& ra, const WfState& w, unsigned workgroups, float* out, hipStream_t st)
{
    hipLaunchKernelGGL((k_wf_chunks<F, W>), dim3(workgroups), dim3(256), 0, st, sv, ra, w, out);
}
void launch_chunks_variant(int variant, const SceneView& sv, const RenderArgs& ra, const WfState& w, unsigned workgroups, float* out, hipStream_t st)
{
    using namespace ptk;
    switch (variant) {
    case V_CORNELL_WORLD: case V_CORNELL_WORLD_FLAT: launch_chunks<M_CORNELL, CornellWorld>(sv, ra, w, workgroups, out, st); break;
    case V_CORNELL: launch_chunks<M_CORNELL, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_SPHERES: launch_chunks<M_SPHERES, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_SPHERES_LBVH: launch_chunks<M_SPHERES_LBVH, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_SPHERES_LBVH4: launch_chunks<M_SPHERES_LBVH4, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_TRIMESH: launch_chunks<M_TRIMESH, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_TRIMESH4: launch_chunks<M_TRIMESH4, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_FINAL: launch_chunks<M_FINAL, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_FINAL_LBVH: launch_chunks<M_FINAL_LBVH, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_FINAL_LBVH4: launch_chunks<M_FINAL_LBVH4, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_ALL: launch_chunks<M_ALL, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    case V_ALL_LBVH4: launch_chunks<M_ALL_LBVH4, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    default: launch_chunks<M_ALL_LBVH, AnyWorld>(sv, ra, w, workgroups, out, st); break;
    }
}

// The feature mask of the kernel a scene variant runs on (what launch_variant instantiates): it,
// not the scene's own mask, decides which lane fields the kernel touches.
uint32_t variant_mask(int variant)
{
    using namespace ptk;
    switch (variant) {
    case V_CORNELL_WORLD: case V_CORNELL_WORLD_FLAT: case V_CORNELL: return M_CORNELL;
    case V_SPHERES: return M_SPHERES;
    case V_SPHERES_LBVH: return M_SPHERES_LBVH;
    case V_SPHERES_LBVH4: return M_SPHERES_LBVH4;
    case V_TRIMESH: return M_TRIMESH;
    case V_TRIMESH4: return M_TRIMESH4;
    case V_FINAL: return M_FINAL;
    case V_FINAL_LBVH: return M_FINAL_LBVH;
    case V_FINAL_LBVH4: return M_FINAL_LBVH4;
    case V_ALL: return M_ALL;
    case V_ALL_LBVH4: return M_ALL_LBVH4;
    default: return M_ALL_LBVH;
    }
}

void launch_variant(int variant, const SceneView& sv, const RenderArgs& ra, const WfState& w, unsigned groups, float* out, hipStream_t st)
{
    using namespace ptk;
    switch (variant) {
    case V_CORNELL_WORLD: case V_CORNELL_WORLD_FLAT: launch_groups<M_CORNELL, CornellWorld>(sv, ra, w, groups, out, st); break;
    case V_CORNELL: launch_groups<M_CORNELL, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_SPHERES: launch_groups<M_SPHERES, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_SPHERES_LBVH: launch_groups<M_SPHERES_LBVH, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_SPHERES_LBVH4: launch_groups<M_SPHERES_LBVH4, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_TRIMESH: launch_groups<M_TRIMESH, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_TRIMESH4: launch_groups<M_TRIMESH4, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_FINAL: launch_groups<M_FINAL, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_FINAL_LBVH: launch_groups<M_FINAL_LBVH, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_FINAL_LBVH4: launch_groups<M_FINAL_LBVH4, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_ALL: launch_groups<M_ALL, AnyWorld>(sv, ra, w, groups, out, st); break;
    case V_ALL_LBVH4: launch_groups<M_ALL_LBVH4, AnyWorld>(sv, ra, w, groups, out, st); break;
    default: launch_groups<M_ALL_LBVH, AnyWorld>(sv, ra, w, groups, out, st); break;
    }
}

}  // namespace

namespace pt {

int render_wavefront(PtScene* sc, const RenderArgs& ra, float* d_out, hipStream_t st)
{
    const uint32_t n_slots = (uint32_t)ra.n_wave_tiles * 64u;
    if (n_slots == 0) return PT_OK;
    // Group size.  Small groups win (table at wf_min_blocks): a group is one workgroup's worth of
    // parallelism however many lanes it holds, a frame of larger groups is fewer workgroups than the GPU
    // has room for, and the launch ends with its slowest group.  Worlds searched through a tree of boxes
    // and spheres gain a little from 512 (their rounds are long, the barriers between stages few).
    const uint32_t kernel_mask = variant_mask(sc->variant);
    int group_lanes = ((kernel_mask & (F_LBVH | F_REFBVH)) && !(sc->mask & F_TRIANGLE) && n_slots >= 512u * 1024u) ? 512 : 256;
#ifdef PT_WF_BLOCKS
    if (const char* e = getenv("PT_WF_GROUP")) group_lanes = atoi(e);
#endif
    const bool in_lds = (kernel_mask & (F_LBVH | F_REFBVH | F_IMAGE)) == 0;      // lanes_in_lds: the kernel keeps the lane fields itself
    if (in_lds) group_lanes = WF_GROUP_LDS;
    if (group_lanes < 256 || group_lanes > WF_GROUP_MAX || group_lanes % 256 != 0) {      // the queues are sized for WF_GROUP_MAX lanes
        pt::set_error("wavefront: group size must be a multiple of 256 up to WF_GROUP_MAX"); return PT_ERR_INVALID;
    }
    const unsigned groups = (n_slots + (uint32_t)group_lanes - 1) / (uint32_t)group_lanes;
    const uint32_t n_lanes = groups * (uint32_t)group_lanes;
    // one allocation: the lane fields and the flag.  The scene keeps it between calls (a host that renders
    // frame after frame pays for it once); it grows when a larger shard comes along.
    const bool image = (kernel_mask & F_IMAGE) != 0;      // the kernels compiled with image textures keep the uv field
    const size_t per_lane = 16 * (image ? 6 : 5) + 8 + 4;
    // Chunks of samples (k_wf_chunks): asked for (chunk_samples), or by itself when the groups are few for the GPU -- between
    // two and sixteen rounds of its resident workgroups: under two the launch ends with its slowest group either way, with
    // many rounds its end is a small part of it -- and the samples many; about 32 rounds of work items then.  Rectangle
    // scene 1024^2 x4096: 348 ms whole groups, the same paths as a 4096^2 frame at 256 samples (groups to spare) 287; random
    // spheres in the tree 1920x1080 x256 215 against 189 as 3840x2160 x64, final in the tree 260 against 210; 2 M triangles
    // 3840x2160 (25 rounds) 147 against 158.
    int pieces = 1;
    const long long resident = 256LL * wf_min_blocks(kernel_mask);
    if (!ra.stage_stats && ra.ns >= 2) {
        if (ra.wf_chunk_samples > 0) pieces = (ra.ns + ra.wf_chunk_samples - 1) / ra.wf_chunk_samples;
        else if ((long long)groups >= 2 * resident && (long long)groups <= 16 * resident && ra.ns >= 128)
            pieces = (int)std::min<long long>((32 * resident + groups - 1) / groups, ra.ns / 32);
        pieces = (int)std::min<long long>(std::max(pieces, 1), 0x7fffffffLL / groups);      // work items are numbered in 32 bits
    }
    const int chunk_samples = pieces > 1 ? (ra.ns + pieces - 1) / pieces : ra.ns;
    if (pieces > 1) pieces = (ra.ns + chunk_samples - 1) / chunk_samples;
    const size_t states_bytes = (pieces > 1 && in_lds) ? (size_t)groups * WF_GROUP_STATE_WORDS * sizeof(unsigned long long) : 0;     // lanes in global memory keep their states in place
    const size_t words_bytes = pieces > 1 ? (((size_t)4 + 2 * (size_t)groups) * sizeof(unsigned) + 255) / 256 * 256 : 0;      // ticket, cost sum and count; a progress and a cost word per group
    const size_t bytes = (in_lds ? 0 : (size_t)n_lanes * per_lane) + 256 + states_bytes + words_bytes;
    if (sc->wf_workspace_bytes < bytes) {
        if (sc->wf_workspace) { (void)hipFree(sc->wf_workspace); sc->wf_workspace = nullptr; sc->wf_workspace_bytes = 0; }
        if (!hip_ok2(hipMalloc(&sc->wf_workspace, bytes), "hipMalloc(wavefront state)")) return PT_ERR_OOM;
        sc->wf_workspace_bytes = bytes;
    }
    char* p = static_cast<char*>(sc->wf_workspace);
    WfState w{};
    auto take = [&](size_t nbytes) { char* r = p; p += nbytes; return r; };
    if (!in_lds) {
        w.ray_o = reinterpret_cast<float4*>(take((size_t)n_lanes * 16));
        w.ray_d = reinterpret_cast<float4*>(take((size_t)n_lanes * 16));
        w.hit = reinterpret_cast<int4*>(take((size_t)n_lanes * 16));
        w.thr = reinterpret_cast<float4*>(take((size_t)n_lanes * 16));
        w.acc = reinterpret_cast<float4*>(take((size_t)n_lanes * 16));
        w.uv = image ? reinterpret_cast<int4*>(take((size_t)n_lanes * 16)) : nullptr;
        w.rng = reinterpret_cast<uint64_t*>(take((size_t)n_lanes * 8));
        w.pix = reinterpret_cast<uint32_t*>(take((size_t)n_lanes * 4));
    }
    w.lost = reinterpret_cast<int*>(take(256));
    w.ns_end = ra.ns;
    w.chunk_samples = chunk_samples; w.n_chunks = pieces;
    w.words = nullptr; w.group_states = nullptr;
    if (pieces > 1) {
        if (states_bytes) w.group_states = reinterpret_cast<unsigned long long*>(take(states_bytes));
        w.words = reinterpret_cast<unsigned*>(take(words_bytes));
        if (!hip_ok2(hipMemsetAsync(w.words, 0, words_bytes, st), "hipMemsetAsync(chunk words)")) return PT_ERR_HIP;
    }
    w.stats = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w.lost) + 64);
    static_assert(sizeof(PtStageStats) == 16 * sizeof(unsigned long long) && 64 + sizeof(PtStageStats) <= 256, "the flag block holds the stage counters");
    w.group_lanes = group_lanes;
    // a round is one bounce of every live lane of the group, and a pixel is at most ns paths of
    // max_depth bounces plus the round that ends each: more rounds than that mean a lane was lost
    const long long limit = (long long)ra.ns * ((long long)ra.max_depth + 2) + 8;
    w.round_limit = (int32_t)std::min<long long>(limit, 0x7fffffff);
    if (!hip_ok2(hipMemsetAsync(w.lost, 0, 256, st), "hipMemsetAsync(flag and stage counters)")) return PT_ERR_HIP;

    if (pieces > 1) {
        const unsigned workgroups = (unsigned)std::min<long long>((long long)groups * pieces, resident);     // persistent: they draw work items from the ticket
        launch_chunks_variant(sc->variant, sc->view, ra, w, workgroups, d_out, st);
    } else {
        launch_variant(sc->variant, sc->view, ra, w, groups, d_out, st);
    }
    if (!hip_ok2(hipGetLastError(), "wavefront launch")) return PT_ERR_HIP;
    int lost = 0;
    if (!hip_ok2(hipMemcpyAsync(&lost, w.lost, sizeof(int), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(flag)")) return PT_ERR_HIP;
    if (ra.stage_stats && !hip_ok2(hipMemcpyAsync(&sc->stage_stats, w.stats, sizeof(PtStageStats), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(stage counters)")) return PT_ERR_HIP;
    if (!hip_ok2(hipStreamSynchronize(st), "wavefront render")) return PT_ERR_HIP;
    if (lost) {
        pt::set_error(lost == 2 ? "wavefront: a group's chunk gave up waiting for the chunk before it; the frame is not valid"
                                : "wavefront: a group did not finish within the rounds its sample and bounce counts allow");
        return PT_ERR_HIP;
    }
    sc->last.n_launches = 1;
    sc->last.n_chunks = pieces > 1 ? pieces : 0;
    if (ra.stage_stats) sc->have_stage_stats = true;
    return PT_OK;
}

}  // namespace pt
