// The fused radiance kernels (one lane per pixel, stages fused, path regeneration in-lane) and the
// closest-hit kernel, as templates over a feature mask and a world walk, with the bodies of their
// launchers (declared in pt_kernels.h).  Included only by the translation units that compile a
// specialisation (pt_k_*.hip).
#ifndef PT_KERNELS_IMPL_H
#define PT_KERNELS_IMPL_H

#include <hip/hip_runtime.h>

#include <algorithm>

#include "pt_kernels.h"

using namespace ptd;
using namespace ptk;

// ---- kernels ----------------------------------------------------------------------------------------

#ifdef PT_PROFILE
// Section sums of every wave of the profiled translation unit's kernels (see Prof): ticks, lanes, runs per section.
__device__ unsigned long long g_prof[3 * PROF_SECTIONS];
__device__ __forceinline__ void prof_flush(const Prof& pf)
{
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < PROF_SECTIONS; k++) {
            atomicAdd(&g_prof[3 * k + 0], pf.t[k]); atomicAdd(&g_prof[3 * k + 1], pf.l[k]); atomicAdd(&g_prof[3 * k + 2], pf.n[k]);
        }
}
extern "C" int pt_debug_profile(unsigned long long* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long zero[3 * PROF_SECTIONS] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof(zero)) != hipSuccess) return -1; }
    return PROF_SECTIONS;
}
#else
__device__ __forceinline__ void prof_flush(const Prof&) {}
#endif
#ifdef PT_PROFILE
#define PROF_LANES(name, cond) const unsigned long long name = __ballot(cond)
#else
#define PROF_LANES(name, cond) constexpr unsigned long long name = 0ull
#endif
constexpr int PROF_SHADE = 2, PROF_PIXEL = 3, PROF_RECORD = 4, PROF_CAMERA = 5, PROF_WALK_HIT = 6, PROF_END = 7;

// Workgroups per CU the fused kernels are compiled for (a register cap; one workgroup is one wave per
// SIMD).  The compiler's free choice leaves the rectangle-scene kernel at 5 waves per SIMD and the
// full-feature kernels at 2.  Measured on MI355X (kernel ms): rectangle scene 1024^2 x512 at 5/6/7/8 =
// 37.7/37.6/37.8/38.5; RTNW final 800^2 x64 through the tree at 2/3/4/5 = 35.7/31.9/36.4/37.1 and x16
// through the reference's own BVH 22.9/19.6/21.1/21.1; random spheres 1200x800 x64 through the tree at
// 3/4/5/6 = 31.2/27.0/25.9/25.3, x16 over the list at 1/4/5 = 49.0/45.4/43.5; 2 M triangles 3840x2160
// x16 at 4/6 = 64.5/59.7 and, later, x64 at 5/6 = 191.5/185.0 (4 against 24 spilled registers: 0.2 against 1 GB
// of scratch written per 4 spp, which HBM has to spare there; a 20 000-triangle mesh 42.2/41.5).  The
// tree kernels keep 24 KB of stack in LDS per workgroup, so no more than 6 fit a CU whatever is asked.
// With the 4-wide tree the final scene's kernel wants the waves more than the registers (table in the function).
__host__ __device__ constexpr int fused_min_blocks(uint32_t F)
{
#ifdef PT_EXP_BLOCKS            // kernel experiments: -DPT_EXP_BLOCKS=n overrides the table below for the file being compiled
    return PT_EXP_BLOCKS;
#endif
    const uint32_t G = F & ~F_LBVH4;                       // the form of the tree does not change the table
    if (F == M_FINAL_LBVH4) return 6;                       // the final scene through the 4-wide tree: 3 / 4 / 5 / 6 = 59.7 / 57.0 / 56.4 / 53.6 ms (1920x1080 x64)
    // The final scene's kernels wanted registers (3 workgroups) while the rows of the log tables sat in them (pt_libm.h); with
    // those gone the kernel that walks the reference's BVH needs 134 and prefers waves: 3 / 4 / 5 / 6 workgroups = 132.8 / 108.7 /
    // 98.7 / 105.8 ms (1920x1080 x64, REFILL in chunks of one sample; 0 / 0 / 43 / 63 spilled registers), through the binary tree
    // 49.7 / 50.6 / 47.2 / 47.1.  The all-features kernel over the 4-wide tree (no cap before: 1): the rectangle scene asked into
    // the tree 1024^2 x64 at 1 / 4 / 5 / 6 = 29.3 / 24.4 / 24.0 / 27.2, simple_light 10.4 / 9.2 / 9.9 / 12.6.
    // The sphere scene over its flat list, 4 / 5 / 6 / 8: 221.2 / 194.6 / 184.2 / 209.3 ms (1920x1080 x64), simple_spheres 8.21 / 8.12 / 7.90 / 10.70.
    if (G == M_FINAL || G == M_FINAL_LBVH) return 5;
    if (F == M_ALL_LBVH4) return 4;
    // The all-features kernels over lists and the binary tree, 3 / 4 / 5: simple_light 1024x512 x256 28.3 / 24.9 / 26.3 ms, the rectangle
    // scene asked into the binary tree 1024^2 x64 26.0 / 21.7 / 22.3.
    return (G & F_REFBVH) ? 4 : (G == M_CORNELL) ? 6 : (G == M_SPHERES_LBVH) ? 6 : (G == M_SPHERES) ? 6 : (G == M_TRIMESH) ? 6 : 1;
}

// Signature worlds shade from the face table (shade_sig): every workgroup keeps a copy in LDS, 64 bytes
// per (entry, face), read with a lane's own index once per bounce -- a global load there is the
// longest wait of the bounce when few waves share a SIMD.
template <uint32_t F, class W> struct FaceTable { static constexpr int float4s = 1; static constexpr bool used = false; };
template <uint32_t F, int... K> struct FaceTable<F, WorldSig<K...>> { static constexpr bool used = (F & ~M_CORNELL) == 0; static constexpr int float4s = used ? (int)sizeof...(K) * 8 * (SIG_FACE / 4) : 1; };
template <uint32_t F, class Sig> struct FaceTable<F, Branchless<Sig>> : FaceTable<F, Sig> {};
template <uint32_t F, class W>
__device__ __forceinline__ void face_table_to_lds(const SceneView& sv, float4* lds)
{
    if constexpr (FaceTable<F, W>::used) {
        for (int k = threadIdx.x; k < FaceTable<F, W>::float4s; k += blockDim.x) { const float* f = sv.sig_faces + 4 * k; lds[k] = make_float4(f[0], f[1], f[2], f[3]); }
        __syncthreads();
    }
}

// One lane's path tracer state: the pixel's random stream, its running sum and the path in flight.
struct LaneState {
    Pcg rng;
    V3 accum, thr;
    Ray ray;
    int x, y;             // render_kernel's pixel coordinates (y counts up from the bottom row)
    int s, depth;
    bool fresh;           // the next step starts a new camera sample
};

__device__ __forceinline__ void lane_begin_pixel(LaneState& st, const RenderArgs& ra, int x, int ytop, unsigned& i)
{
    st.x = x;
    st.y = ra.ny - 1 - ytop;                              // row 0 of the image is the top
    i = (unsigned)(ytop * ra.nx + x);                     // = (ny - y - 1) * nx + x   (src/ptMain.cu:178)
    st.rng.seed((uint64_t)i + ra.seed_offset);            // PcgRng rng(i)              (src/ptMain.cu:183)
    st.accum = mk(0, 0, 0);
    st.s = 0;
    st.fresh = true;
}

// One bounce of the pixel's sample loop.  Returns true when the pixel's last sample has ended.
// `ns_end`: the sample count at which the lane's work item ends (the pixel's last sample, or the end of a chunk).
template <uint32_t F, class W>
__device__ __forceinline__ bool lane_step(const SceneView& sv, const RenderArgs& ra, LaneState& st, const Rcp& fnx, const Rcp& fny,
                                          int* my_stack, const float4* faces, int ns_end)
{
    if (st.fresh) {                                       // render_pixel, src/ptMain.cu:158-160
        // (x + rand) / nx: the numerator is 0 or at least 2^-32 and the frame size at most 2^31, inside
        // the range where the shortened division is the whole division (rcp_refined, pt_device.h)
        float u = quot((float)st.x + st.rng.rand(), fnx);
        float v = quot((float)st.y + st.rng.rand(), fny);
        st.ray = camera_ray(sv, u, v, st.rng);
        st.thr = mk(1, 1, 1);
        st.depth = 0;
        st.fresh = false;
    }
    bool ended;
    if (st.depth >= ra.max_depth) {
        ended = true;                                     // bounce limit: color() returns the running product
    } else {
        Hit h;
        if (world_hit<F, W>(sv, st.ray, 0.001f, FLT_MAX, h, st.rng, my_stack)) {
            if constexpr (!is_any_world<W>::value && (F & ~M_CORNELL) == 0) ended = !shade_sig<F, W>(sv, faces, st.ray, h, st.thr, st.rng);
            else ended = !shade<F>(sv, st.ray, h, st.thr, st.rng);
        } else {
            shade_miss<F>(sv, st.ray, st.thr);
            ended = true;
        }
        st.depth++;
    }
    if (ended) {                                          // accumCol += deNan(color(...)), src/ptMain.cu:46-53,161
        if (!(st.thr.x == st.thr.x)) st.thr.x = 0;
        if (!(st.thr.y == st.thr.y)) st.thr.y = 0;
        if (!(st.thr.z == st.thr.z)) st.thr.z = 0;
        st.accum = st.accum + st.thr;
        st.s++;
        st.fresh = true;
        return st.s >= ns_end;
    }
    return false;
}

__device__ __forceinline__ void lane_write_pixel(const LaneState& st, const RenderArgs& ra, unsigned i, float* __restrict__ out)
{
    if (ra.raw_sum) {                                     // a sample batch: sums are added across batches, finished later
        out[3 * (size_t)i + 0] = st.accum.x; out[3 * (size_t)i + 1] = st.accum.y; out[3 * (size_t)i + 2] = st.accum.z;
        return;
    }
    const float inv_ns = 1 / (float)ra.ns;                // accumCol /= float(ns): reciprocal, then multiplies
    out[3 * (size_t)i + 0] = sqrtf(fmaxf(0.0f, st.accum.x * inv_ns));
    out[3 * (size_t)i + 1] = sqrtf(fmaxf(0.0f, st.accum.y * inv_ns));
    out[3 * (size_t)i + 2] = sqrtf(fmaxf(0.0f, st.accum.z * inv_ns));
}

__device__ __forceinline__ void lane_init(LaneState& st)
{
    st.rng.state = 0; st.rng.inc = 1;
    st.accum = mk(0, 0, 0); st.thr = mk(1, 1, 1);
    st.ray.o = mk(0, 0, 0); st.ray.d = mk(0, 0, 1); st.ray.time = 0;
    st.x = 0; st.y = 0; st.s = 0; st.depth = 0; st.fresh = true;
}

// Fused scheduler, one 8x8 tile per wave and one pixel per lane.  The lanes of a wave start in
// phase, which keeps the shading branches less divergent when shading dominates a bounce.
template <uint32_t F, class W>
__global__ __launch_bounds__(256, fused_min_blocks(F)) void k_render_tiles(SceneView sv, RenderArgs ra, float* __restrict__ out)
{
    __shared__ int lds_stack[(F & (F_LBVH | F_REFBVH)) ? 4 * 64 * LBVH_LDS_STACK : 1];
    __shared__ float4 faces[FaceTable<F, W>::float4s];
    face_table_to_lds<F, W>(sv, faces);            // before any lane leaves: the copy ends in a barrier
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wave_in_block;
    if (wave >= ra.n_wave_tiles) return;
    int x, ytop;
    if (!wave_pixel(ra, wave, lane, x, ytop)) return;
    int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[wave_in_block * 64 * LBVH_LDS_STACK + lane] : nullptr;
    const Rcp fnx = rcp_refined((float)ra.nx), fny = rcp_refined((float)ra.ny);     // loop-invariant
    LaneState st;
    lane_init(st);
    unsigned i;
    lane_begin_pixel(st, ra, x, ytop, i);
    while (!lane_step<F, W>(sv, ra, st, fnx, fny, my_stack, faces, ra.ns)) {}
    lane_write_pixel(st, ra, i, out);
}

// One pass of a progressive render: samples [first, first + count) of every owned pixel.  The
// pixel's random stream and running sum wait in `acc` between passes (passes end on sample
// boundaries, so no path is ever in flight), which makes any split into passes produce the
// image of a single launch, and lets a host checkpoint `acc` and resume.  Each pass also
// writes the estimate so far, sqrt(sum / samples done), the reference's normalisation
// (src/ptMain.cu:163-166) applied to the samples that exist.
template <uint32_t F, class W>
__global__ __launch_bounds__(256, fused_min_blocks(F)) void k_render_pass(SceneView sv, RenderArgs ra, PtAccumPixel* __restrict__ acc,
                                                     int first, float* __restrict__ out, int* __restrict__ bad_resume)
{
    __shared__ int lds_stack[(F & (F_LBVH | F_REFBVH)) ? 4 * 64 * LBVH_LDS_STACK : 1];
    __shared__ float4 faces[FaceTable<F, W>::float4s];
    face_table_to_lds<F, W>(sv, faces);            // before any lane leaves: the copy ends in a barrier
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wave_in_block;
    if (wave >= ra.n_wave_tiles) return;
    int x, ytop;
    if (!wave_pixel(ra, wave, lane, x, ytop)) return;
    int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[wave_in_block * 64 * LBVH_LDS_STACK + lane] : nullptr;
    const Rcp fnx = rcp_refined((float)ra.nx), fny = rcp_refined((float)ra.ny);     // loop-invariant
    LaneState st;
    lane_init(st);
    unsigned i;
    lane_begin_pixel(st, ra, x, ytop, i);
    if (first > 0) {
        const PtAccumPixel a = acc[i];
        if (a.samples != (uint32_t)first) *bad_resume = 1;    // not the state the earlier passes left: the host reports it
        st.rng.state = ((uint64_t)a.rng_hi << 32) | a.rng_lo;
        st.accum = mk(a.sum[0], a.sum[1], a.sum[2]);
        st.s = first;
    }
    while (!lane_step<F, W>(sv, ra, st, fnx, fny, my_stack, faces, ra.ns)) {}      // ra.ns = first + count for this launch
    PtAccumPixel a;
    a.rng_lo = (uint32_t)st.rng.state; a.rng_hi = (uint32_t)(st.rng.state >> 32);
    a.sum[0] = st.accum.x; a.sum[1] = st.accum.y; a.sum[2] = st.accum.z;
    a.samples = (uint32_t)ra.ns;
    acc[i] = a;
    lane_write_pixel(st, ra, i, out);
}

// k_render_tiles with every pixel's samples cut into chunks, for frames of few tiles and many samples: there
// a wave tile is one long work item (1024^2 pixels at 4096 samples: 16 384 items of about 90 ms for 6 144 wave
// slots), the launch ends with most of the GPU idle behind its last items, and the same paths traced as a
// 4096^2 frame at 256 samples take 16 % less time.  Here the work item is (chunk, group of four wave tiles),
// numbered chunk-major; persistent workgroups draw item numbers from a counter, so items start in order.
// A pixel's samples stay one sequence (src/ptMain.cu:156-162): between chunks its random stream's state and its
// running sum wait in `tile_states`, 1280 bytes per wave tile (64 states of 8 bytes, then the sum's three components as
// three rows of 64 floats), so that a wave stores and loads whole 128-byte lines whatever the tile's place in the frame;
// chunk c of a group starts once the group's progress word says chunk c - 1 has been stored.  That wait cannot deadlock: the item waited for has a
// smaller number, so a running workgroup holds it, and the chain of waits ends at a chunk 0, which waits for
// nothing.  The wait is bounded all the same; running out raises *fault (the frame is then wrong, and the
// host says so).
#ifndef PT_CHUNK_WAVES
#define PT_CHUNK_WAVES 4
#endif
// wave tiles per work item = waves per workgroup.  4 / 2 / 1 (-DPT_CHUNK_WAVES): rectangle scene 1024^2 x4096 213.2 / 216.5 / 211.8 ms,
// RTNW final through the 4-wide tree 1920x1080 x256 192.0 / 190.9 / 187.0: the waves of a workgroup waiting for each other is not what is left
constexpr int CHUNK_WAVES = PT_CHUNK_WAVES;
constexpr int CHUNK_TILE_STATE_WORDS = 64 + 3 * 32;      // 64-bit words of a wave tile's state between chunks: 1280 bytes
template <uint32_t F, class W>
__global__ __launch_bounds__(64 * CHUNK_WAVES, fused_min_blocks(F)) void k_render_chunks(SceneView sv, RenderArgs ra, float* __restrict__ out,
                                                                             unsigned* __restrict__ words, unsigned long long* __restrict__ tile_states,
                                                                             int* __restrict__ fault)
{
    __shared__ int lds_stack[(F & (F_LBVH | F_REFBVH)) ? CHUNK_WAVES * 64 * LBVH_LDS_STACK : 1];
    __shared__ float4 faces[FaceTable<F, W>::float4s];
    __shared__ unsigned s_item, s_prio;
    face_table_to_lds<F, W>(sv, faces);
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[wave_in_block * 64 * LBVH_LDS_STACK + lane] : nullptr;
    const Rcp fnx = rcp_refined((float)ra.nx), fny = rcp_refined((float)ra.ny);     // loop-invariant
    const unsigned groups = (unsigned)(ra.n_wave_tiles + CHUNK_WAVES - 1) / (unsigned)CHUNK_WAVES;
    const unsigned n_items = groups * (unsigned)ra.n_chunks;
    // words: [0] the ticket, [1] [2] sum and count of the groups' first-chunk costs, [4 ..) a progress word per group,
    // then a cost word per group.  A launch cannot end before its most expensive tile has run through all its chunks,
    // one after the other, and among six waves on a SIMD that chain advances at a sixth of the pace it has alone
    // (tiles differ by a factor of five in bounces per pixel).  So a group's first chunk is timed, and its later chunks
    // run at a wave priority that grows with that time over the mean's: the expensive chains are served first by
    // their SIMDs and the cheap tiles fill what is left.  Which wave issues first changes no value.
    unsigned* const progress = words + 4;
    unsigned* const cost = words + 4 + groups;
    unsigned long long item_began = 0;
    unsigned finished_cost = 0;
    // Thread 0 has one errand per round, between the barrier that ends an item and the one that starts the next:
    // publish the finished item, then draw the next number.  (Written as two -- one at the end of the loop body,
    // one at its start -- the compiler joins them across the loop's back edge into a loop of thread 0's own and
    // leaves the other lanes of its wave circling through the barriers without it.)
    bool finished_one = false;
    unsigned finished_grp = 0, finished_c = 0;
    while (true) {
        if (threadIdx.x == 0) {
            if (finished_one) {
                if (finished_c == 0 && ra.chunk_priority) {
                    __hip_atomic_store(&cost[finished_grp], finished_cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicAdd(&words[1], finished_cost); atomicAdd(&words[2], 1u);
                }
                __hip_atomic_store(&progress[finished_grp], finished_c + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_prio = 0;
            s_item = atomicAdd(&words[0], 1u);
        }
        __syncthreads();
        // through a scalar register: the compiler must see that the whole workgroup leaves (or stays in) the loop
        // together, or it arranges the loop for lanes that leave at different times and the barriers no longer pair up
        const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)s_item);
        if (item >= n_items) break;
        const int c = (int)(item / groups);
        const unsigned grp = item - (unsigned)c * groups;
        if (c > 0) {
            if (threadIdx.x == 0) {
                // The bound is on time without progress, not on time: the group's word counts the chunks stored so far, and
                // a long chain of waits (few groups, long chunks) moves it while this workgroup waits.  Only a word that
                // stands still for 2^23 polls (about 14 s) is given up on.
                unsigned spins = 0, seen = 0;
                // relaxed polls: an acquiring load invalidates the cache behind it every time, and a thousand waiting
                // workgroups doing that made the working ones ten times slower; the fence below acquires once
                while (true) {
                    const unsigned now = __hip_atomic_load(&progress[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (now >= (unsigned)c) break;
                    if (now != seen) { seen = now; spins = 0; }
                    __builtin_amdgcn_s_sleep(64);
                    if (++spins > (1u << 23)) { *fault = 2; break; }
                }
                if (ra.chunk_priority) {                      // this group's first chunk against the mean of those finished so far
                    const unsigned long long mine = __hip_atomic_load(&cost[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long sum = __hip_atomic_load(&words[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long cnt = __hip_atomic_load(&words[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long tenths = sum ? 10ull * mine * cnt / sum : 10ull;
                    s_prio = tenths >= 15 ? 3u : tenths >= 12 ? 2u : tenths >= 9 ? 1u : 0u;
                }
            }
            __syncthreads();
            __threadfence();                                  // every wave: the loads below come after the word
        }
        {
            const unsigned prio = (unsigned)__builtin_amdgcn_readfirstlane((int)s_prio);
            if (prio == 3u) __builtin_amdgcn_s_setprio(3);
            else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
            else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        item_began = clock64();
        const int wave = (int)grp * CHUNK_WAVES + wave_in_block;
        int x = 0, ytop = 0;
        if (wave < ra.n_wave_tiles && wave_pixel(ra, wave, lane, x, ytop)) {
            LaneState st;
            lane_init(st);
            unsigned i;
            lane_begin_pixel(st, ra, x, ytop, i);
            unsigned long long* const tile_rng = tile_states + (size_t)wave * CHUNK_TILE_STATE_WORDS;
            float* const tile_sum = reinterpret_cast<float*>(tile_rng + 64);
            if (c > 0) {
                st.rng.state = tile_rng[lane];
                st.accum = mk(tile_sum[lane], tile_sum[64 + lane], tile_sum[128 + lane]);
                st.s = c * ra.chunk_samples;
            }
            const int ns_end = min(ra.ns, (c + 1) * ra.chunk_samples);     // this item's samples end where the next chunk begins
            while (!lane_step<F, W>(sv, ra, st, fnx, fny, my_stack, faces, ns_end)) {}
            if (c == ra.n_chunks - 1) {
                lane_write_pixel(st, ra, i, out);
            } else {
                tile_rng[lane] = st.rng.state;
                tile_sum[lane] = st.accum.x; tile_sum[64 + lane] = st.accum.y; tile_sum[128 + lane] = st.accum.z;
            }
        }
        __threadfence();                                      // every lane's stores before the word
        __syncthreads();                                      // (also: everyone has read s_item)
        finished_one = true; finished_grp = grp; finished_c = (unsigned)c;
        finished_cost = (unsigned)min((unsigned long long)0xffffffffu, (clock64() - item_began) >> 6);
    }
}

// ---- a pixel's samples in chunks under the persistent-lane schedules ----------------------------------------
// With one slot per pixel a launch of few pixels and many samples ends pixel by pixel: the lanes that drew the last
// slots trace all of their samples while the rest of the GPU has run dry (random spheres 1920x1080 at 1024 samples:
// five pixels per lane, a tenth of the launch spent that way).  So a slot is (chunk, pixel), numbered chunk-major, and
// a lane traces one chunk of a pixel's samples.  The samples stay one sequence (src/ptMain.cu:156-162): between
// chunks the pixel's random-stream state and running sum wait in a record of five 64-bit words, each word a
// payload in its low half and the number of chunks finished in its high half, written and read with relaxed
// atomics.  A reader takes the record only when all five words carry the count it needs, so no fence -- which on
// this part writes back or invalidates a whole L2 -- orders the words: a torn read is simply read again.  The lane
// that drew (c, pixel) usually finds chunk c - 1 long finished (it was drawn a whole sweep of the frame earlier);
// when not, the lane keeps the slot and asks again on its next round while the other lanes of its wave carry on
// (the lane that holds chunk c - 1 may be one of them), so nothing spins with a wave's lanes masked off.  The wait
// ends: chunk c - 1 was drawn earlier, so a running lane holds it, and the chain of waits ends at a chunk 0.  It is
// bounded all the same (*fault = 2, the host reports the frame as invalid).  Records are zeroed by the host before
// every launch.
constexpr int CHUNK_REC_WORDS = 5;
// A lane waits at most this long for one record (ticks of the 100 MHz wall clock; the low 32 bits wrap after 42.9 s).
// Far beyond any chunk the library cuts itself; a caller who asks for chunks that take a lane longer has asked for too much.
#ifndef PT_CHUNK_WAIT_TICKS
#define PT_CHUNK_WAIT_TICKS 3000000000u
#endif
constexpr unsigned CHUNK_WAIT_TICKS = PT_CHUNK_WAIT_TICKS;
__device__ __forceinline__ unsigned clock_ticks() { return (unsigned)wall_clock64(); }
__device__ __forceinline__ void chunk_record_store(unsigned long long* __restrict__ rec, unsigned i, const LaneState& st, unsigned done)
{
    unsigned long long* r = rec + (size_t)CHUNK_REC_WORDS * i;
    const unsigned long long tag = (unsigned long long)done << 32;
    __hip_atomic_store(&r[0], tag | (uint32_t)st.rng.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&r[1], tag | (uint32_t)(st.rng.state >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&r[2], tag | (uint32_t)__float_as_uint(st.accum.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&r[3], tag | (uint32_t)__float_as_uint(st.accum.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&r[4], tag | (uint32_t)__float_as_uint(st.accum.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true when the record holds the state after `done` chunks; st then continues from there
__device__ __forceinline__ bool chunk_record_load(unsigned long long* __restrict__ rec, unsigned i, LaneState& st, unsigned done)
{
    unsigned long long* r = rec + (size_t)CHUNK_REC_WORDS * i;
    const unsigned long long w0 = __hip_atomic_load(&r[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w1 = __hip_atomic_load(&r[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w2 = __hip_atomic_load(&r[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w3 = __hip_atomic_load(&r[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w4 = __hip_atomic_load(&r[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool whole = (unsigned)(w0 >> 32) == done && (unsigned)(w1 >> 32) == done && (unsigned)(w2 >> 32) == done &&
                       (unsigned)(w3 >> 32) == done && (unsigned)(w4 >> 32) == done;
    if (whole) {
        st.rng.state = (w0 & 0xffffffffull) | (w1 << 32);
        st.accum = mk(__uint_as_float((unsigned)w2), __uint_as_float((unsigned)w3), __uint_as_float((unsigned)w4));
    }
    return whole;
}
// ---- pixel slots from one counter per XCD ---------------------------------------------------------------
// The persistent-lane schedules hand out pixel slots (64 per 8x8 wave tile; with chunks, (chunk, pixel) pairs) from a
// device counter, one atomic per wave for all its lanes that ran dry.  One word serves one atomic at a time, about 22 ns
// each with the eight XCDs asking, and a launch of few samples a pixel asks often: random spheres 1920x1080 at 16
// samples spent a fifth of its time there.  So the shard's wave tiles are dealt to eight counters in blocks (block k
// to counter k mod 8) and a wave draws from the counter of the XCD it runs on (its id is in a hardware register); a
// wave whose counter has run dry moves on to the next, so the shard still ends with every lane busy.  The eight
// sweeps advance side by side along one front.  (Eight contiguous bands of the frame instead -- each XCD's L2 then
// sees its own eighth of a scene too large for one -- were 12 % slower on 2 M triangles at 64 samples: eight fronts
// far apart share little in the Infinity Cache, and that outweighs what the L2s gain.)  Which lane traces a pixel
// changes no value.  The counters sit 32 bytes apart in the 256-byte counter block.
constexpr int SLOT_BANDS = 8, SLOT_BAND_STRIDE = 8;
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7; }      // hwreg(HW_REG_XCC_ID, 0, 4)
// The j-th draw from `band`: false when the band has no slot j (it is dry); else the chunk and pixel slot, chunk-major
// within the band, so that a pixel's chunk c - 1 is always drawn -- from this same counter -- before its chunk c.
__device__ __forceinline__ bool slot_band_draw(const RenderArgs& ra, int band, unsigned j, unsigned& c, unsigned& pixel_slot)
{
    const unsigned tiles = (unsigned)ra.n_wave_tiles;
    const unsigned chunks = (unsigned)max(1, ra.n_chunks);
    if (ra.slot_bands <= 1) {                                 // one sweep of the shard
        const unsigned pixels = tiles * 64u;
        if (j >= pixels * chunks) return false;               // the host keeps the shard's slots times chunks below 2^32
        c = ra.n_chunks > 1 ? j / pixels : 0u;
        pixel_slot = j - c * pixels;
        return true;
    }
    // blocks of B wave tiles, block k in band k mod 8: the bands' sweeps advance side by side along one front
    const unsigned B = (unsigned)ra.band_block_tiles;
    const unsigned n_blocks = (tiles + B - 1) / B;
    const unsigned b = (unsigned)band;
    const unsigned my_blocks = n_blocks > b ? (n_blocks - b + (unsigned)SLOT_BANDS - 1) / (unsigned)SLOT_BANDS : 0u;
    unsigned band_tiles = my_blocks * B;
    if (my_blocks > 0 && (n_blocks - 1) % (unsigned)SLOT_BANDS == b) band_tiles -= n_blocks * B - tiles;      // the shard's last block is this band's last, and short
    const unsigned band_pixels = band_tiles * 64u;
    if (j >= band_pixels * chunks) return false;
    c = ra.n_chunks > 1 ? j / band_pixels : 0u;
    const unsigned in_chunk = j - c * band_pixels;
    const unsigned tile_in_band = in_chunk >> 6;
    const unsigned block = tile_in_band / B;
    pixel_slot = ((block * (unsigned)SLOT_BANDS + b) * B + (tile_in_band - block * B)) * 64u + (in_chunk & 63u);
    return true;
}

// Fused scheduler with persistent lanes: a lane that has finished its pixel takes the next pixel
// slot of the shard (slots are numbered tile by tile, 64 per 8x8 wave tile) from a device counter,
// so a wave stays full until the shard runs out.  Pays when the closest-hit search, whose loop is
// uniform over the wave, dominates a bounce.  Which lane traces a pixel does not matter: the
// pixel's random stream is seeded by its index (src/ptMain.cu:178-184).
template <uint32_t F, class W>
__global__ __launch_bounds__(256, fused_min_blocks(F)) void k_render_refill(SceneView sv, RenderArgs ra, float* __restrict__ out,
                                                       unsigned* __restrict__ next_slot, unsigned long long* __restrict__ chunk_rec,
                                                       int* __restrict__ fault)
{
    __shared__ int lds_stack[(F & (F_LBVH | F_REFBVH)) ? 4 * 64 * LBVH_LDS_STACK : 1];
    __shared__ float4 faces[FaceTable<F, W>::float4s];
    face_table_to_lds<F, W>(sv, faces);            // before any lane leaves: the copy ends in a barrier
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[wave_in_block * 64 * LBVH_LDS_STACK + lane] : nullptr;
    const Rcp fnx = rcp_refined((float)ra.nx), fny = rcp_refined((float)ra.ny);     // loop-invariant
    const int home_band = ra.slot_bands > 1 ? xcc_id() : 0;   // see slot_band_draw
    int bands_dry = 0;                                        // bands this wave has found empty (the same number in all its lanes)
    bool band_dry = false;
    LaneState st;
    lane_init(st);
    unsigned i = 0;
    int ns_end = ra.ns;
    unsigned need_chunks = 0, wait_from = 0;          // need_chunks > 0: the slot is held, the record of the pixel's earlier chunks is not there yet
    bool have = false;
    while (true) {
        // the lanes still in the loop are all here: the wave moves on to the next band together, and a lane that found the last one dry is done
        if (__ballot(band_dry) != 0ull) bands_dry++;
        if (band_dry) { band_dry = false; if (bands_dry >= ra.slot_bands) break; }     // the shard is exhausted
        const bool wave_busy = __ballot(have) != 0ull;
        if (!have && need_chunks == 0) {
            const int band = ra.slot_bands > 1 ? ((home_band + bands_dry) & (SLOT_BANDS - 1)) : 0;
            const unsigned long long need = __ballot(1);  // one atomic per wave for all lanes that ran dry
            const int leader = __ffsll((long long)need) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(next_slot + band * SLOT_BAND_STRIDE, (unsigned)__popcll(need));
            base = __shfl(base, leader);
            const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            unsigned c = 0, ps = 0;
            if (!slot_band_draw(ra, band, base + (unsigned)__popcll(need & lower), c, ps)) { band_dry = true; continue; }     // ask the next band, or stop (above)
            int x, ytop;
            if (!wave_pixel(ra, (int)(ps >> 6), (int)(ps & 63), x, ytop)) continue;   // slot beyond the image edge
            lane_begin_pixel(st, ra, x, ytop, i);
            ns_end = ra.n_chunks > 1 ? min(ra.ns, (int)(c + 1) * ra.chunk_samples) : ra.ns;
            if (c > 0) { need_chunks = c; wait_from = clock_ticks(); st.s = (int)c * ra.chunk_samples; }
            else have = true;
        }
        if (need_chunks > 0) {
            if (chunk_record_load(chunk_rec, i, st, need_chunks)) { need_chunks = 0; have = true; }
            else if (clock_ticks() - wait_from > CHUNK_WAIT_TICKS) { *fault = 2; break; }
            else { if (!wave_busy) __builtin_amdgcn_s_sleep(32); continue; }      // nobody in this wave has work: ease off the memory system
        }
        if (lane_step<F, W>(sv, ra, st, fnx, fny, my_stack, faces, ns_end)) {
            if (ns_end >= ra.ns) lane_write_pixel(st, ra, i, out);
            else chunk_record_store(chunk_rec, i, st, (unsigned)(ns_end / ra.chunk_samples));
            have = false;
        }
    }
}

// The in-order list walk a ray takes when the tree cannot answer for it (a non-finite ray, a wrapped
// rectangle's plane: see ray_may_enter_tree), out of line for worlds of plain shapes: it runs for a
// handful of rays per frame, and inlined into the shared-walk kernel its shape dispatch costs that
// kernel registers on every bounce.  No medium, no reference BVH: it draws no random number and needs
// no stack.  Arguments and result by value, the scene through its copy in device memory.
struct ListWalkHit { int any; float t; int id, face; };
template <uint32_t F>
__device__ __attribute__((noinline)) ListWalkHit list_walk_out_of_line(const SceneView* view, float ox, float oy, float oz,
                                                                     float dx, float dy, float dz, float time)
{
    static_assert((F & (F_MEDIUM | F_REFBVH | F_LBVH | F_IMAGE)) == 0, "plain shapes, walked in order");
    Ray r; r.o = mk(ox, oy, oz); r.d = mk(dx, dy, dz); r.time = time;
    Hit h;
    Pcg unused; unused.state = 0; unused.inc = 1;
    ListWalkHit out;
    out.any = world_hit<F, AnyWorld>(*view, r, 0.001f, FLT_MAX, h, unused, nullptr) ? 1 : 0;
    out.t = h.t; out.id = h.id; out.face = h.face;
    return out;
}

// Fused scheduler for worlds held in the library's LBVH: persistent lanes as in k_render_refill, and
// the closest-hit search is shared by the wave one tree step at a time.  Lanes whose search has
// ended wait; when WALK_WAIT_LANES of them wait (or nobody is left walking) the wave leaves the
// search loop, those lanes shade, scatter or start their next sample (or pixel) and begin their
// next search, and the wave steps again -- lanes that were still walking simply carry on.  With one
// search per lane_step call the wave instead runs every search to the end of its longest walk: on
// the sphere scene 28 % of the VALU lanes did work, on the 2 M-triangle field 8 %.  What each lane
// computes, and the order of its random draws, is unchanged (the search draws none).
template <uint32_t F>
__global__ __launch_bounds__(256, fused_min_blocks(F)) void k_render_walk(SceneView sv, RenderArgs ra, float* __restrict__ out,
                                                                           unsigned* __restrict__ next_slot, unsigned long long* __restrict__ chunk_rec,
                                                                           int* __restrict__ fault)
{
    static_assert((F & F_LBVH) != 0, "the shared walk is the LBVH walk");
    __shared__ int lds_stack[4 * 64 * LBVH_LDS_STACK];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    int* my_stack = &lds_stack[wave_in_block * 64 * LBVH_LDS_STACK + lane];
    const Rcp fnx = rcp_refined((float)ra.nx), fny = rcp_refined((float)ra.ny);     // loop-invariant
    const int home_band = ra.slot_bands > 1 ? xcc_id() : 0;   // see slot_band_draw
    int bands_dry = 0;                                        // bands this wave has found empty (the same number in all its lanes)
    enum { NEED_PIXEL, NEED_RECORD, NEXT_RAY, WALKING, WALKED, DONE };
    int ns_end = ra.ns;                                       // where this lane's work item ends: the pixel's last sample or a chunk's
    unsigned wait_from = 0;
    LaneState st;
    lane_init(st);
    LbvhWalk w;
    lbvh_begin(sv, st.ray, FLT_MAX, w);
    int spill[LBVH_STACK - LBVH_LDS_STACK];
    unsigned i = 0;
    int mode = NEED_PIXEL;
    Prof pf;
    pf.begin();
    while (true) {
        PROF_LANES(walked_lanes, mode == WALKED);
        if (mode == WALKED) {                                 // the rest of the bounce, as in lane_step
            Hit h;
            h.id = PT_ID_MISS; h.face = 0; h.t = FLT_MAX; h.uv_id = PT_ID_MISS; h.uv_face = 0; h.uv_t = 0.0f;
            bool any = w.any;
            bool searched_in_order = false;                      // the whole list was walked the reference's way: nothing is left to second-guess
            if (any) { h.t = w.best; h.id = w.best_id; h.face = w.best_face; }
            if (accepts_nan<F>::value && (w.odd || !ray_may_enter_tree(sv, st.ray))) {     // see accepts_nan: the list, in order
                // the few-candidates shortcut is for worlds of plain list entries: leaves of a reference BVH follow that tree's
                // own tie rule, and an image texture needs the in-order walk's record of the last uv written (as in world_hit)
                InPlaneHit got; got.found = -1; got.t = FLT_MAX; got.id = -1; got.face = 0;
                if constexpr ((F & (F_REFBVH | F_IMAGE)) == 0)
                    got = lbvh_in_plane_t<F>(sv.self, st.ray.o.x, st.ray.o.y, st.ray.o.z, st.ray.d.x, st.ray.d.y, st.ray.d.z, st.ray.time, 0.001f, FLT_MAX);
                if (got.found < 0) {
                    if constexpr ((F & (F_MEDIUM | F_REFBVH | F_IMAGE)) == 0) {
                        const ListWalkHit lw = list_walk_out_of_line<(F & ~(F_LBVH | F_LBVH4))>(sv.self, st.ray.o.x, st.ray.o.y, st.ray.o.z,
                                                                                              st.ray.d.x, st.ray.d.y, st.ray.d.z, st.ray.time);
                        any = lw.any != 0;
                        h.id = PT_ID_MISS; h.face = 0; h.t = FLT_MAX;
                        if (any) { h.t = lw.t; h.id = lw.id; h.face = lw.face; }
                    } else {
                        any = world_hit<(F & ~F_LBVH), AnyWorld>(sv, st.ray, 0.001f, FLT_MAX, h, st.rng, my_stack);
                        searched_in_order = true;
                    }
                } else {
                    any = got.found > 0;
                    h.id = PT_ID_MISS; h.face = 0; h.t = FLT_MAX;
                    if (any) { h.t = got.t; h.id = got.id; h.face = got.face; }
                }
            }
            if constexpr ((F & F_REFBVH) != 0) {
                // The nearest leaf came out of a reference BVH whose leaves are in the library's tree: the reference only
                // finds it if the boxes above it let the ray in (see world_hit: the box of the node that names the leaf
                // decides, and can turn away a ray that grazes an edge although the leaf's own test accepts it).  Such a ray
                // is searched again the reference's way, in list order.
                if (!searched_in_order && any && sv.lbvh_group && h.id >= 0 && sv.lbvh_group[h.id] >= 0) {
                    const PT_SCENE float* gate = sv.ref_leaf_gate + 6 * (size_t)h.id;
                    const V3 inv = mk(1 / st.ray.d.x, 1 / st.ray.d.y, 1 / st.ray.d.z);
                    if (!aabb_hit_inv(gate, gate + 3, st.ray.o, inv, 0.001f, FLT_MAX)) {
                        any = world_hit<(F & ~F_LBVH), AnyWorld>(sv, st.ray, 0.001f, FLT_MAX, h, st.rng, my_stack);
                        searched_in_order = true;
                    }
                }
            }
            if (F & F_IMAGE) {
                // a plain sphere whose texture reads uv shows the uv of the hit the in-order walk accepted
                // before it, which the tree does not know: that ray is searched again in list order
                if (!searched_in_order && sv.reads_uv && any && hit_reads_stale_uv<F>(sv, h.id))
                    any = world_hit<(F & ~F_LBVH), AnyWorld>(sv, st.ray, 0.001f, FLT_MAX, h, st.rng, my_stack);
            }
            bool ended;
            if (any) {
                ended = !shade<F>(sv, st.ray, h, st.thr, st.rng);
            } else {
                shade_miss<F>(sv, st.ray, st.thr);
                ended = true;
            }
            st.depth++;
            if (!ended && st.depth >= ra.max_depth) ended = true;      // bounce limit: the running product is the sample
            mode = NEXT_RAY;
            if (ended) {
                if (!(st.thr.x == st.thr.x)) st.thr.x = 0;
                if (!(st.thr.y == st.thr.y)) st.thr.y = 0;
                if (!(st.thr.z == st.thr.z)) st.thr.z = 0;
                st.accum = st.accum + st.thr;
                st.s++;
                st.fresh = true;
                if (st.s >= ns_end) {
                    if (ns_end >= ra.ns) lane_write_pixel(st, ra, i, out);
                    else chunk_record_store(chunk_rec, i, st, (unsigned)(ns_end / ra.chunk_samples));
                    mode = NEED_PIXEL;
                }
            }
        }
        pf.mark(PROF_SHADE, walked_lanes);
        PROF_LANES(pixel_lanes, mode == NEED_PIXEL);
        bool band_dry = false;
        if (mode == NEED_PIXEL) {
            const int band = ra.slot_bands > 1 ? ((home_band + bands_dry) & (SLOT_BANDS - 1)) : 0;
            const unsigned long long need = __ballot(1);      // one atomic per wave for all lanes that ran dry
            const int leader = __ffsll((long long)need) - 1;
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(next_slot + band * SLOT_BAND_STRIDE, (unsigned)__popcll(need));
            base = __shfl(base, leader);
            const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            unsigned c = 0, ps = 0;
            int x, ytop;
            if (!slot_band_draw(ra, band, base + (unsigned)__popcll(need & lower), c, ps)) band_dry = true;     // ask the next band, or stop (below)
            else if (wave_pixel(ra, (int)(ps >> 6), (int)(ps & 63), x, ytop)) {
                lane_begin_pixel(st, ra, x, ytop, i);
                ns_end = ra.n_chunks > 1 ? min(ra.ns, (int)(c + 1) * ra.chunk_samples) : ra.ns;
                mode = NEXT_RAY;
                if (c > 0) { st.s = (int)c * ra.chunk_samples; st.depth = (int)c; wait_from = clock_ticks(); mode = NEED_RECORD; }    // depth is free until the first ray
            }
            // else: a slot beyond the image edge; ask again next round
        }
        // every lane of the wave is here: the wave moves on to the next band together, and a lane that found the last one dry is done
        if (__ballot(band_dry) != 0ull) bands_dry++;
        if (band_dry && bands_dry >= ra.slot_bands) mode = DONE;      // the shard is exhausted
        pf.mark(PROF_PIXEL, pixel_lanes);
        PROF_LANES(record_lanes, mode == NEED_RECORD);
        if (mode == NEED_RECORD) {                            // the state the pixel's earlier chunks left; not there yet: ask again next round
            if (chunk_record_load(chunk_rec, i, st, (unsigned)st.depth)) mode = NEXT_RAY;
            else if (clock_ticks() - wait_from > CHUNK_WAIT_TICKS) { *fault = 2; mode = DONE; }
        }
        pf.mark(PROF_RECORD, record_lanes);
        PROF_LANES(ray_lanes, mode == NEXT_RAY);
        if (mode == NEXT_RAY) {
            if (st.fresh) {                                   // render_pixel, src/ptMain.cu:158-160
                float u = quot((float)st.x + st.rng.rand(), fnx);
                float v = quot((float)st.y + st.rng.rand(), fny);
                st.ray = camera_ray(sv, u, v, st.rng);
                st.thr = mk(1, 1, 1);
                st.depth = 0;
                st.fresh = false;
            }
            lbvh_begin(sv, st.ray, FLT_MAX, w);
            mode = WALKING;
        }
        pf.mark(PROF_CAMERA, ray_lanes);
        if (__ballot(mode != DONE) == 0ull) break;
        // lanes that have something to do once the wave leaves the search loop; a lane waiting for a record has not
        // (counted, a wave of mostly waiting lanes would leave the loop at once, every round, and never step the rest)
        const int live = __popcll(__ballot(mode != DONE && mode != NEED_RECORD));
        while (true) {
            const int walking = __popcll(__ballot(mode == WALKING));
            if (walking == 0 || live - walking >= ra.walk_wait) break;
            if (lbvh_voted_step<F>(sv, st.ray, 0.001f, FLT_MAX, w, my_stack, spill, mode == WALKING, pf)) mode = WALKED;
        }
    }
    prof_flush(pf);
}

template <uint32_t F, class W>
__global__ __launch_bounds__(256) void k_trace_closest(SceneView sv, const PtRay* __restrict__ rays, size_t n,
                                                       int32_t* __restrict__ prim_id, float* __restrict__ t_out)
{
    __shared__ int lds_stack[(F & (F_LBVH | F_REFBVH)) ? 4 * 64 * LBVH_LDS_STACK : 1];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int* my_stack = (F & (F_LBVH | F_REFBVH)) ? &lds_stack[(threadIdx.x >> 6) * 64 * LBVH_LDS_STACK + (threadIdx.x & 63)] : nullptr;
    const PtRay pr = rays[i];
    Ray r;
    r.o = mk(pr.o[0], pr.o[1], pr.o[2]);
    r.d = mk(pr.d[0], pr.d[1], pr.d[2]);
    r.time = pr.time;
    Pcg rng;
    rng.seed(pr.seed);
    Hit h;
    if (world_hit<F, W>(sv, r, 0.001f, FLT_MAX, h, rng, my_stack)) { prim_id[i] = h.id; t_out[i] = h.t; }
    else { prim_id[i] = PT_ID_MISS; t_out[i] = FLT_MAX; }
}


// Launchers of one specialisation.
template <uint32_t F, class W>
void FusedLaunch<F, W>::render(const SceneView& sv, const RenderArgs& ra, float* out, unsigned* next_slot, unsigned long long* chunk_rec, int* fault,
                               hipStream_t st)
{
    // REFILL: persistent grid of at most 32 waves per CU, the rest of the shard comes from the counter.
    // A world in the LBVH is searched by the wave-shared walk (which ends a path at the bounce
    // limit after a shade, so a limit of zero bounces stays with the plain refill kernel).
    const bool refill = ra.refill != 0;
    const int tiles4 = (ra.n_wave_tiles + 3) / 4;
    const dim3 block(256), grid(refill ? std::min(tiles4, 2048) : tiles4);
    if (ra.refill == 2) {
        if constexpr ((F & F_LBVH) != 0) hipLaunchKernelGGL((k_render_walk<F>), grid, block, 0, st, sv, ra, out, next_slot, chunk_rec, fault);
    } else if (refill) hipLaunchKernelGGL((k_render_refill<F, W>), grid, block, 0, st, sv, ra, out, next_slot, chunk_rec, fault);
    else hipLaunchKernelGGL((k_render_tiles<F, W>), grid, block, 0, st, sv, ra, out);
}
template <uint32_t F, class W>
void FusedLaunch<F, W>::chunks(const SceneView& sv, const RenderArgs& ra, float* out, unsigned* words, unsigned long long* tile_states, int* fault, hipStream_t st)
{
    const long long items = (long long)((ra.n_wave_tiles + CHUNK_WAVES - 1) / CHUNK_WAVES) * ra.n_chunks;
    const dim3 block(64 * CHUNK_WAVES), grid((unsigned)std::min<long long>(items, 8192 / CHUNK_WAVES));     // persistent: 256 CUs hold at most 24 waves each
    hipLaunchKernelGGL((k_render_chunks<F, W>), grid, block, 0, st, sv, ra, out, words, tile_states, fault);
}
template <uint32_t F, class W>
void FusedLaunch<F, W>::pass(const SceneView& sv, const RenderArgs& ra, PtAccumPixel* acc, int first, float* out, int* bad_resume, hipStream_t st)
{
    const dim3 block(256), grid((ra.n_wave_tiles + 3) / 4);
    hipLaunchKernelGGL((k_render_pass<F, W>), grid, block, 0, st, sv, ra, acc, first, out, bad_resume);
}
template <uint32_t F, class W>
void FusedLaunch<F, W>::trace(const SceneView& sv, const PtRay* rays, size_t n, int32_t* ids, float* ts, hipStream_t st)
{
    const dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL((k_trace_closest<F, W>), grid, block, 0, st, sv, rays, n, ids, ts);
}

#endif
