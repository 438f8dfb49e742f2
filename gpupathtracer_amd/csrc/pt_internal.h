// Declarations shared by the translation units that implement the C ABI.
#ifndef PT_INTERNAL_H
#define PT_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "pt_device.h"

struct RenderArgs {
    int32_t nx, ny, ns, max_depth;
    int32_t shard_index, shard_count;
    int32_t tile_w, tile_h;        // shard tile size in pixels
    int32_t tiles_x;               // tiles per image row
    int32_t n_tiles;               // all tiles of the image
    int32_t sub_x, sub_y;          // 8x8 wave tiles per shard tile
    int32_t n_wave_tiles;          // wave tiles owned by this shard
    int32_t refill;                // fused scheduler: 1 persistent lanes that pull pixel slots from a counter, 2 the same with the shared tree walk
    int32_t walk_wait;             // shared walk: lanes with a finished search that make the wave leave the search loop
    int32_t slot_bands;            // persistent-lane schedules: 1, or 8 = pixel slots drawn from eight counters, a wave starting with its XCD's (slot_band_draw)
    int32_t band_block_tiles;      // ... in blocks of this many wave tiles, block k in band k mod 8
    int32_t raw_sum;               // write the plain sample sums instead of sqrt(sum / ns)
    uint64_t seed_offset;          // added to the pixel index before seeding (sample batches; 0 = the reference's stream)
    int32_t chunk_samples, n_chunks;   // chunked tiles: samples per work item and items per tile (0: not chunked)
    int32_t stage_stats;           // staged scheduler: count what the stages did
    int32_t wf_chunk_samples;      // staged scheduler: PtRenderParams.chunk_samples as the caller gave it (0: the library's choice)
    int32_t chunk_priority;        // chunked tiles: later chunks of expensive tiles run at a raised wave priority
};

// Wave w of the launch -> its 8x8 pixel block.  Shard tiles owned by this shard are
// tile = shard_index + k * shard_count; each is cut into sub_x * sub_y wave tiles.
static __device__ __forceinline__ bool wave_pixel(const RenderArgs& ra, int wave, int lane, int& x, int& ytop)
{
    const int per_tile = ra.sub_x * ra.sub_y;
    const int k = wave / per_tile, sub = wave - k * per_tile;
    const int tile = ra.shard_index + k * ra.shard_count;
    const int ty = tile / ra.tiles_x, tx = tile - ty * ra.tiles_x;
    const int sy = sub / ra.sub_x, sx = sub - sy * ra.sub_x;
    const int lx = sx * 8 + (lane & 7), ly = sy * 8 + (lane >> 3);
    x = tx * ra.tile_w + lx;
    ytop = ty * ra.tile_h + ly;
    return lx < ra.tile_w && ly < ra.tile_h && x < ra.nx && ytop < ra.ny;
}


// Kernel specialisations by feature mask; a scene runs on the first one that covers it.
namespace ptk {
using namespace ptd;
constexpr uint32_t M_CORNELL = F_RECT | F_BOX | F_XFORM | F_LIGHTS;
constexpr uint32_t M_SPHERES = F_SPHERE | F_MSPHERE | F_METAL | F_DIELECTRIC | F_CHECKER | F_SKY;
constexpr uint32_t M_SPHERES_LBVH = M_SPHERES | F_LBVH;
constexpr uint32_t M_TRIMESH = F_RECT | F_TRIANGLE | F_LIGHTS | F_LBVH;
constexpr uint32_t M_ALL = F_ALL & ~F_LBVH;
constexpr uint32_t M_ALL_LBVH = F_ALL;
// The RTNW final scene's kinds (src/ptMain.cu:350-421): spheres, boxes in a reference BVH, a rectangle lamp,
// two media, metal / glass / smoke, marble noise and the image texture -- no wrappers, triangles, checkers, sky
// or light shapes beyond the one rectangle, whose code costs the all-features kernels registers and scratch.
constexpr uint32_t M_FINAL = F_SPHERE | F_MSPHERE | F_RECT | F_BOX | F_MEDIUM | F_REFBVH | F_LIGHTS | F_METAL | F_DIELECTRIC |
                             F_ISOTROPIC | F_NOISE | F_IMAGE;
constexpr uint32_t M_FINAL_LBVH = M_FINAL | F_LBVH;
// the tree scenes again, walked through the tree's 4-wide form
constexpr uint32_t M_SPHERES_LBVH4 = M_SPHERES_LBVH | F_LBVH4;
constexpr uint32_t M_TRIMESH4 = M_TRIMESH | F_LBVH4;
constexpr uint32_t M_FINAL_LBVH4 = M_FINAL_LBVH | F_LBVH4;
constexpr uint32_t M_ALL_LBVH4 = M_ALL_LBVH | F_LBVH4;

enum : int {
    V_CORNELL = 0, V_SPHERES = 1, V_SPHERES_LBVH = 2, V_TRIMESH = 3, V_ALL = 4, V_ALL_LBVH = 5,
    V_CORNELL_WORLD = 6,        // M_CORNELL with the CornellWorld signature walk (the fused kernels and the closest-hit kernel)
    V_CORNELL_WORLD_FLAT = 7,   // the same walk without control flow (launches that leave few waves per SIMD)
    V_FINAL = 8, V_FINAL_LBVH = 9,
    V_SPHERES_LBVH4 = 10, V_TRIMESH4 = 11, V_FINAL_LBVH4 = 12, V_ALL_LBVH4 = 13, N_VARIANTS = 14
};
}  // namespace ptk

struct PtScene {
    int device = 0;
    ptd::SceneView view{};
    std::vector<void*> allocs;          // every device allocation of this scene
    uint32_t mask = 0;
    int variant = -1;
    int n_shapes = 0, n_media = 0;
    int lbvh_nodes = 0;
    unsigned* next_slot = nullptr;      // device counter of the fused scheduler's pixel-slot pool
    int* bad_resume_host = nullptr;     // pinned host word a progressive pass raises when its accumulator is not at first_sample
    int* bad_resume_dev = nullptr;      // the same word as the device sees it
    void* chunk_workspace = nullptr;    // chunked tiles: random-stream states (8 bytes per pixel), ticket and per-group progress words, kept between calls
    size_t chunk_workspace_bytes = 0;
    void* wf_workspace = nullptr;       // staged scheduler: lane state and queues, kept between calls
    void* staging = nullptr;            // device side of the host-buffer entry points (pt_render, pt_trace_closest, ...), kept between calls
    size_t staging_bytes = 0;
    size_t wf_workspace_bytes = 0;
    // timing of the most recent render
    hipEvent_t ev_begin = nullptr, ev_k0 = nullptr, ev_k1 = nullptr, ev_end = nullptr;
    bool have_timing = false;
    PtRenderStats last{};
    PtStageStats stage_stats{};         // of the most recent render on the staged pipeline
    bool have_stage_stats = false;
};


// Every (feature mask, world walk) the fused kernels are compiled for: X is `template struct` in the
// translation unit that compiles one of them, `extern template struct` where they are only launched.
#define PT_KERNEL_VARIANTS(X)                                        \
    X FusedLaunch<ptk::M_CORNELL, ptd::AnyWorld>;                    \
    X FusedLaunch<ptk::M_SPHERES, ptd::AnyWorld>;                    \
    X FusedLaunch<ptk::M_SPHERES_LBVH, ptd::AnyWorld>;               \
    X FusedLaunch<ptk::M_TRIMESH, ptd::AnyWorld>;                    \
    X FusedLaunch<ptk::M_ALL, ptd::AnyWorld>;                        \
    X FusedLaunch<ptk::M_ALL_LBVH, ptd::AnyWorld>;                   \
    X FusedLaunch<ptk::M_CORNELL, ptd::CornellWorld>;                \
    X FusedLaunch<ptk::M_CORNELL, ptd::Branchless<ptd::CornellWorld>>; \
    X FusedLaunch<ptk::M_FINAL, ptd::AnyWorld>;                      \
    X FusedLaunch<ptk::M_FINAL_LBVH, ptd::AnyWorld>;                 \
    X FusedLaunch<ptk::M_SPHERES_LBVH4, ptd::AnyWorld>;              \
    X FusedLaunch<ptk::M_TRIMESH4, ptd::AnyWorld>;                   \
    X FusedLaunch<ptk::M_FINAL_LBVH4, ptd::AnyWorld>;                \
    X FusedLaunch<ptk::M_ALL_LBVH4, ptd::AnyWorld>;

namespace pt {
// Staged (wavefront) scheduler, pt_wavefront.hip: one launch; blocks until the frame is complete.
int render_wavefront(PtScene* sc, const RenderArgs& ra, float* d_out, hipStream_t st);
}

#endif
