/*
 * ptcore.h - C ABI of the MI355X path-tracing core.
 *
 * The reference (rweyrauch/gpupathtracer) has no plugin interface: its one
 * host<->device seam is the `if (!cpu) { ... }` block of main()
 * (src/ptMain.cu:579-653), which serialises a pointer-linked scene, has a
 * <<<1,1>>> kernel rebuild it with device-side `new`
 * (allocate_world_kernel, src/ptMain.cu:423-430), launches render_kernel
 * (src/ptMain.cu:171-190) and copies nx*ny float3 back.  This header is the
 * C ABI a maintainer binds in place of that block.  The scene crosses the
 * boundary as flat arrays with type tags instead of a byte stream of virtual
 * objects; results (closest hit, image) are defined to be those of the
 * reference's host-compiled hit/material/integrator code.
 *
 * All structs are plain old data, 4-byte fields only, no padding surprises.
 * All pointers in PtSceneDesc are host pointers; the library copies what it
 * needs at pt_scene_create() time and the caller may free them afterwards.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- type tags ---------------------------------------------------------- */

/* Leaf geometry; replaces the Hitable subclasses of include/ptHitable.h:32-49. */
enum PtShapeType {
    PT_SHAPE_SPHERE        = 1, /* p = cx cy cz r                       (include/ptSphere.h:30-79)    */
    PT_SHAPE_MOVING_SPHERE = 2, /* p = c0xyz c1xyz t0 t1 r              (include/ptSphere.h:81-122)   */
    PT_SHAPE_RECT_XY       = 3, /* p = x0 x1 y0 y1 k                    (include/ptRectangle.h:24-54) */
    PT_SHAPE_RECT_XZ       = 4, /* p = x0 x1 z0 z1 k                    (include/ptRectangle.h:56-106)*/
    PT_SHAPE_RECT_YZ       = 5, /* p = y0 y1 z0 z1 k                    (include/ptRectangle.h:108-138)*/
    PT_SHAPE_BOX           = 6, /* p = pmin xyz, pmax xyz; six rects    (include/ptRectangle.h:194-260)*/
    PT_SHAPE_TRIANGLE      = 7  /* p = v0 xyz, v1 xyz, v2 xyz           (include/ptTriangle.h:19-54)  */
};

/* PtShape.flags */
enum { PT_FLIP_NORMAL = 1 };  /* odd number of FlipNormals wrappers (include/ptRectangle.h:140-192) */

/* Instance wrappers, applied outermost first (include/ptRectangle.h:262-445). */
enum PtXformType {
    PT_XFORM_TRANSLATE = 1,   /* v = offset xyz            */
    PT_XFORM_ROTATE_Y  = 2    /* v = sinTheta cosTheta 0   */
};

/* include/ptMaterial.h:39-46 */
enum PtMaterialType {
    PT_MAT_LAMBERTIAN    = 1, /* texture = albedo                       */
    PT_MAT_METAL         = 2, /* albedo rgb, param = fuzz (already <=1) */
    PT_MAT_DIELECTRIC    = 3, /* param = refractive index               */
    PT_MAT_DIFFUSE_LIGHT = 4, /* texture = emission                     */
    PT_MAT_ISOTROPIC     = 5  /* texture = albedo                       */
};

/* include/ptTexture.h:19-25 */
enum PtTextureType {
    PT_TEX_CONSTANT = 1,      /* color                                   */
    PT_TEX_CHECKER  = 2,      /* scale (reference: 10), odd, even        */
    PT_TEX_NOISE    = 3,      /* scale                                   */
    PT_TEX_IMAGE    = 4       /* nx, ny, texel_offset into texels (RGB8) */
};

/* Entries of the ordered top-level world list (HitableList, src/ptHitableList.cu:12-27). */
enum PtObjectType {
    PT_OBJ_SHAPE  = 1,        /* index into shapes  */
    PT_OBJ_MEDIUM = 2,        /* index into media   */
    PT_OBJ_BVH    = 3         /* index into bvhs    */
};

/* include/ptAmbientLight.h:18-22; NONE = null pointer (src/ptMain.cu:145-147). */
enum PtAmbientType { PT_AMBIENT_NONE = 0, PT_AMBIENT_CONSTANT = 1, PT_AMBIENT_SKY = 2 };

/* How the library may search the world list. */
enum PtAccel {
    PT_ACCEL_LIST = 0,        /* walk the list in order, exactly as HitableList::hit does           */
    PT_ACCEL_LBVH = 1,        /* library-built Morton LBVH; same closest hit and the same winner on  */
                              /* exact ties as the in-order walk.  A list of PT_OBJ_SHAPE entries    */
                              /* goes into the tree whole.  A list that starts with an un-wrapped    */
                              /* reference BVH (leaves of one tie kind) and/or shapes goes in up to  */
                              /* its first medium or further BVH -- the BVH's leaves become leaves   */
                              /* of the tree, with the reference tree's own answer kept where its   */
                              /* node boxes turn a ray away -- and is walked in order from there.    */
                              /* Lists the tree cannot stand for fall back to LIST.  Rays for which  */
                              /* the reference's rectangle, box and triangle tests accept a NaN t    */
                              /* (src/ptRectangle.cu:15: a ray lying in a rectangle's plane, a       */
                              /* non-finite ray) are answered the in-order way, never by the tree.   */
                              /* The walk reads the 4-wide form where the tree's depth allows it (it  */
                              /* measured faster on every kind of scene); the two values below pick  */
                              /* a form explicitly.                                                  */
    PT_ACCEL_LBVH_BINARY = 2, /* PT_ACCEL_LBVH walked as built: 64-byte nodes, two child boxes each  */
    PT_ACCEL_LBVH_WIDE = 3    /* PT_ACCEL_LBVH with every second level folded away: 64-byte nodes,   */
                              /* four child boxes each on a byte grid (conservative: a box is never  */
                              /* held smaller than it is), half the dependent node fetches per search */
};

/* Which scheduler runs the radiance pipeline stages. */
enum PtPipeline {
    PT_PIPELINE_DEFAULT   = 0,
    PT_PIPELINE_FUSED     = 1, /* one lane per pixel, stages fused, path regeneration in-lane   */
    PT_PIPELINE_WAVEFRONT = 2  /* staged: generate / extend / shade (one queue per material kind) /  */
                               /* accumulate over compacted lane queues, round by round, one launch  */
};

/* How the fused scheduler hands pixels to lanes.  The image is the same either way. */
enum PtSchedule {
    PT_SCHEDULE_AUTO   = 0, /* the measured best: WALK for triangles and for 64 or more spheres in the LBVH, TILES
                             * for other worlds in the LBVH and for short lists (CHUNKS instead when the frame has
                             * few tiles for the GPU and many samples), REFILL for reference BVHs walked as
                             * they are and for lists of more than 16 objects                                   */
    PT_SCHEDULE_TILES  = 1, /* one 8x8 tile per wave, one pixel per lane                                 */
    PT_SCHEDULE_REFILL = 2, /* persistent lanes pull the next pixel from a device counter (one per XCD).  When the shard holds
                             * more pixels than the GPU has lanes, REFILL and WALK hand out (chunk of samples, pixel)
                             * pairs instead of pixels, chunk by chunk, so that the launch does not end pixel by
                             * pixel: the pixel's random stream and running sum wait in device memory between its
                             * chunks (40 bytes per pixel, kept with the scene), the samples stay the one sequence
                             * of src/ptMain.cu:156-162.  PtRenderParams.chunk_samples: 0 = the library cuts,
                             * >= ns = never; pt_render_device_noalloc and captured streams never cut.           */
    PT_SCHEDULE_WALK   = 3, /* REFILL, and the wave shares the closest-hit search one tree step at a time:
                             * lanes whose search has ended shade and start their next search while the
                             * others keep walking.  Worlds in the library's LBVH (PT_ACCEL_LBVH) with
                             * max_depth >= 1 only; PT_ERR_UNSUPPORTED otherwise.                          */
    PT_SCHEDULE_CHUNKS = 4  /* TILES with every pixel's samples cut into chunks: (tile, chunk) pairs are handed to
                             * workgroups in order from a device counter, a pixel's random stream and running sum
                             * wait in device memory between its chunks (src/ptMain.cu:156-162 runs them as one
                             * loop), so a frame of few tiles and many samples keeps the GPU full to its end.
                             * PtRenderParams.chunk_samples sets the chunk.  Keeps 20 bytes per pixel of device
                             * memory with the scene, allocated by the first pt_render_device call that takes this
                             * schedule for a shard of that size; pt_render_device_noalloc takes the schedule only
                             * when that memory is already there (PT_ERR_UNSUPPORTED when asked for without it). */
};

/* Ids reported by pt_trace_closest. */
#define PT_ID_MISS        (-1)
#define PT_ID_MEDIUM_BASE 0x40000000   /* + index into media */

/* ---- scene description -------------------------------------------------- */

typedef struct PtShape {
    int32_t type;         /* PtShapeType                                     */
    int32_t flags;        /* PT_FLIP_NORMAL                                  */
    int32_t material;     /* index into materials, -1 = none (light shapes)  */
    int32_t xform_first;  /* first wrapper in xforms                         */
    int32_t xform_count;  /* number of wrappers, outermost first             */
    int32_t tri_uv;       /* TRIANGLE: index into tri_uvs, -1 = all zero     */
    float   p[10];
} PtShape;                /* 64 bytes */

typedef struct PtTriUV { float uv0[2], uv1[2], uv2[2]; } PtTriUV;

typedef struct PtXform { int32_t type; float v[3]; } PtXform;

typedef struct PtMaterial {
    int32_t type;
    int32_t texture;      /* -1 = none */
    float   albedo[3];
    float   param;
} PtMaterial;

typedef struct PtTexture {
    int32_t type;
    int32_t odd, even;    /* CHECKER children (indices into textures) */
    int32_t nx, ny;       /* IMAGE */
    int32_t texel_offset; /* IMAGE: byte offset into texels */
    float   color[3];
    float   scale;
} PtTexture;

typedef struct PtObject { int32_t type; int32_t index; } PtObject;

/* ConstantMedium (include/ptMedium.h:19-98): boundary shape, density, Isotropic material. */
typedef struct PtMedium { int32_t boundary; float density; int32_t material; int32_t reserved; } PtMedium;

/* Explicit binary tree, hit exactly as BVH::hit (src/ptBVH.cu:45-74): both children
 * are always visited with the caller's [tmin,tmax]; the right child wins ties.
 * child >= 0: node index; child < 0: ~child is a shape index. */
typedef struct PtBvhNode { float bmin[3]; float bmax[3]; int32_t left; int32_t right; } PtBvhNode;
typedef struct PtBvh { int32_t root; int32_t xform_first; int32_t xform_count; int32_t reserved; } PtBvh;

typedef struct PtSceneDesc {
    int32_t n_shapes;     const PtShape*    shapes;
    int32_t n_tri_uvs;    const PtTriUV*    tri_uvs;
    int32_t n_xforms;     const PtXform*    xforms;
    int32_t n_materials;  const PtMaterial* materials;
    int32_t n_textures;   const PtTexture*  textures;
    int32_t n_texels;     const uint8_t*    texels;      /* RGB8 blobs of all IMAGE textures */
    int32_t n_media;      const PtMedium*   media;
    int32_t n_bvh_nodes;  const PtBvhNode*  bvh_nodes;
    int32_t n_bvhs;       const PtBvh*      bvhs;
    int32_t n_world;      const PtObject*   world;       /* ordered top-level list */
    /* lightShapes argument of color() (src/ptMain.cu:95,119-127).
     * lights_is_list == 0: lights[0] is the one light shape (shape index).
     * lights_is_list == 1: a HitableList of light shapes (uniform mixture, one extra RNG draw per
     *                      sample, include/ptHitableList.h:29-44); lights[] are shape indices.
     * lights_is_list == 2: a BVH of light shapes: lights[0] is an index into bvhs; pdfValue is
     *                      0.5 * left + 0.5 * right at every node, random() descends on
     *                      rand() < 0.5 (src/ptBVH.cu:82-95).
     * Only an un-wrapped XZ rectangle or sphere has a pdf of its own; any other light shape,
     * and any shape under FlipNormals / Translate / RotateY, answers like the Hitable base class:
     * pdfValue 0, random (1,0,0) (include/ptHitable.h:58-59). */
    int32_t n_lights;     const int32_t*    lights;
    int32_t lights_is_list;
    int32_t ambient_type; /* PtAmbientType */
    float   ambient_color[3];
    /* Camera members in declaration order (include/ptCamera.h:43-49):
     * origin, lowerLeftCorner, horizontal, vertical, u, v, w, time0, time1, lens_radius */
    float   camera[24];
    int32_t accel;        /* PtAccel */
} PtSceneDesc;

typedef struct PtRenderParams {
    int32_t nx, ny;       /* image size; out_rgb holds nx*ny*3 floats, row 0 = top (src/ptMain.cu:178) */
    int32_t ns;           /* samples per pixel                                                        */
    int32_t max_depth;    /* bounce limit (reference default 25, src/ptMain.cu:524)                   */
    /* Sharding: the image is cut into tile_w x tile_h tiles numbered row-major from the
     * top-left; this call renders the tiles with (tile % shard_count) == shard_index and
     * writes 0 to every other pixel, so shards sum to the whole frame.  0,0 -> 8x8 tiles. */
    int32_t shard_index, shard_count;
    int32_t tile_w, tile_h;
    int32_t pipeline;     /* PtPipeline */
    int32_t schedule;     /* PtSchedule (fused pipeline only) */
    /* Sample batches: NOT the reference's random sequence.  The reference runs one PCG stream per
     * pixel through all ns samples (src/ptMain.cu:156-162,183), so only whole pixels can be dealt
     * to different GPUs without changing the image.  With batch_count > 1 this call instead traces
     * ns samples of batch batch_index, seeding pixel i with i + batch_index * nx * ny (batch 0 is
     * the reference's stream), so that several GPUs can each take part of a pixel's samples;
     * the batches' images differ from the reference's by sampling noise only.  With raw_sum != 0
     * out_rgb receives the plain sums of the samples (no averaging, no sqrt) so that batches can
     * be added and finished once with pt_frame_finish_device.  Fused pipeline only. */
    int32_t batch_index, batch_count;
    int32_t raw_sum;
    int32_t chunk_samples; /* PT_SCHEDULE_CHUNKS, _REFILL, _WALK (and what AUTO picks): samples per chunk; 0 = the library's choice.
                            * PT_PIPELINE_WAVEFRONT: the same for its groups of pixels (>= ns: whole groups) */
    int32_t stage_stats;   /* PT_PIPELINE_WAVEFRONT: != 0 counts what the stages did, for pt_render_stage_stats (a kernel of
                            * its own: the counting costs the rectangle scene's staged render about 6 % of its rate)      */
    int32_t slot_block_tiles; /* PT_SCHEDULE_REFILL, _WALK (and AUTO where it picks them): how lanes draw pixel slots.  0 = the library's choice;
                               * > 0: from eight counters, one per XCD of the GPU, wave tiles dealt to them in blocks of this
                               * many (block k to counter k mod 8; a wave whose counter has run dry moves on to the next);
                               * < 0: from one counter, one sweep of the shard.  Changes no pixel.                            */
} PtRenderParams;

/* One ray for pt_trace_closest.  The hit range is the integrator's [0.001, FLT_MAX]
 * (src/ptMain.cu:104).  seed feeds the per-ray PCG stream that ConstantMedium::hit
 * draws from (include/ptMedium.h:43); it is unused in scenes without media. */
typedef struct PtRay { float o[3]; float d[3]; float time; uint32_t seed; } PtRay;   /* 32 bytes */

/* Timings of the most recent pt_render*/
/* The radiance kernels are compiled once per family of scenes; a scene runs on the first family that
 * covers every kind of shape, material, texture and light it holds. */
typedef enum PtKernelFamily {
    PT_KERNELS_RECTANGLES = 1, /* rectangles, boxes, wrappers, light shapes: the Cornell box's kinds (src/ptMain.cu:300-348) */
    PT_KERNELS_SPHERES    = 2, /* static and moving spheres, metal, glass, checker, sky: the RTIOW scene's (src/ptMain.cu:192-243) */
    PT_KERNELS_TRIANGLES  = 3, /* triangles and rectangles in the library's tree, light shapes                */
    PT_KERNELS_FINAL      = 4, /* the RTNW final scene's kinds (src/ptMain.cu:350-421)                      */
    PT_KERNELS_ALL        = 5  /* every feature                                                             */
} PtKernelFamily;

typedef struct PtRenderStats {
    float    kernel_ms;      /* HIP-event time of the radiance kernel(s) on their stream */
    float    total_ms;       /* HIP-event time of the whole call on that stream          */
    int32_t  n_launches;     /* kernel launches in the call                               */
    int32_t  pipeline;       /* scheduler that ran                                        */
    uint64_t paths;          /* camera samples traced by this call (this shard only)      */
    int32_t  schedule;       /* PtSchedule the fused pipeline ran (TILES or REFILL), else 0 */
    int32_t  signature_walk; /* 1 when the world list matched a compiled-in signature (straight-line walk) */
    int32_t  kernel_family;  /* PtKernelFamily: which compiled specialisation of the radiance kernels ran */
    int32_t  n_chunks;       /* chunks every pixel's samples were cut into (CHUNKS, or REFILL / WALK on a large shard); 0 = not cut */
    uint64_t reserved[2];
} PtRenderStats;

/* What the stages of the staged pipeline (PT_PIPELINE_WAVEFRONT) did in the most recent render: the reference has
 * one thread per pixel and nothing to report (its only counter is finished pixels, src/ptMain.cu:186-188).  A
 * group is the 256 or 512 pixel lanes one workgroup owns; a round is one pass of its extend stage (closest hit for
 * every queued ray) and of its shade stage (one queue per material kind).  Queue occupancy of a stage is
 * lanes / (64 * waves): the share of lanes that had an entry in the waves the stage ran.  Time share of a stage is
 * its ticks over the sum of both, times PtRenderStats.kernel_ms. */
typedef struct PtStageStats {
    uint64_t groups;          /* workgroups of the launch                                                     */
    uint64_t rounds;          /* rounds, summed over groups                                                   */
    uint64_t extend_lanes;    /* rays searched                                                                */
    uint64_t extend_waves;    /* waves the extend stage ran: 64 queue entries each, a round's last one partly filled */
    uint64_t shade_lanes[4];  /* hits shaded from the Lambertian, Metal, Dielectric and Isotropic queue       */
    uint64_t shade_waves[4];
    uint64_t extend_ticks;    /* shader-clock ticks between a round's start and the barrier after its searches, summed over the timed groups */
    uint64_t shade_ticks;     /* ... and from there to the barrier after its shading                          */
    uint64_t timed_groups;    /* groups that read the clock: one in 64 (reading it every round costs the others too much) */
    uint64_t reserved;
} PtStageStats;

typedef struct PtScene PtScene;

/* ---- entry points -------------------------------------------------------- */

/* Every function returns 0 on success or a negative PtError; the text of the last
 * failure on the calling thread is available from pt_last_error().  The reference
 * prints to std::cerr and returns EXIT_FAILURE instead (src/ptMain.cu:606-610,642-646). */
enum PtError {
    PT_OK = 0, PT_ERR_INVALID = -1, PT_ERR_NO_DEVICE = -2, PT_ERR_HIP = -3,
    PT_ERR_UNSUPPORTED = -4, PT_ERR_OOM = -5, PT_ERR_CANCELLED = -6
};
const char* pt_last_error(void);

/* Replaces scene serialisation + allocate_world_kernel (src/ptMain.cu:560-572,603,423-430):
 * validates the description, uploads it to the current HIP device and, for
 * PT_ACCEL_LBVH, builds the Morton/radix LBVH on the device (in place of
 * BVH::BVH + quickSort, src/ptBVH.cu:15-43, src/ptQuickSort.cu:40-81). */
int  pt_scene_create(const PtSceneDesc* desc, PtScene** out_scene);
void pt_scene_destroy(PtScene* scene);

/* Replaces render_kernel launch + cudaMemcpy (src/ptMain.cu:612-649).  out_rgb is a host
 * buffer of nx*ny*3 floats: averaged, sqrt-gamma'd colour exactly as render_pixel returns
 * it (src/ptMain.cu:153-169), pixel index (ny-y-1)*nx+x.  Blocking.  The device frame behind
 * this call (and behind pt_render_progressive and pt_trace_closest) belongs to the scene: it is
 * allocated on first use, grown when a call needs more, and freed by pt_scene_destroy. */
int  pt_render(PtScene* scene, const PtRenderParams* params, float* out_rgb);

/* Same, but d_out_rgb is device memory on the scene's device and the work is queued on
 * hip_stream (a hipStream_t, NULL = default stream) without a host sync.  A scene renders one
 * frame at a time: its pixel-slot counter and timing events are per scene, so a call first makes
 * hip_stream wait for the scene's previous render (whatever stream that ran on).  To overlap
 * renders, create one scene per stream.  Two schedulers keep per-pixel state with the scene and
 * allocate it in the first call of a frame size, which blocks while it does: PT_PIPELINE_WAVEFRONT
 * for worlds searched through a tree, and PT_SCHEDULE_CHUNKS (which PT_SCHEDULE_AUTO may pick). */
int  pt_render_device(PtScene* scene, const PtRenderParams* params, void* d_out_rgb, void* hip_stream);

/* pt_render_device for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.
 * Same arguments, same frame bit for bit; the call itself performs no hipMalloc / hipFree.  The
 * one piece of device memory the fused pipeline needs beyond the scene and the frame -- the 32-bit
 * pixel-slot counters of the persistent-lane schedules (PT_SCHEDULE_REFILL / _WALK; eight words, one
 * per XCD, see PtRenderParams.slot_block_tiles) -- lives in a 256-byte array that is part of the
 * library's device code, one copy per GPU, instead of in memory the scene allocated.  Limits: the fused pipeline only (PT_PIPELINE_WAVEFRONT allocates its lane state:
 * PT_ERR_UNSUPPORTED); PT_SCHEDULE_CHUNKS, which keeps per-tile state in memory the scene allocates, only when an
 * earlier pt_render_device call has left the scene that memory for a shard at least as large (then PT_SCHEDULE_AUTO
 * picks it here as it does there; without it AUTO runs the tiles whole and asking for CHUNKS is PT_ERR_UNSUPPORTED);
 * REFILL and WALK uncut; at most PT_NOALLOC_MAX_WAVE_TILES 8x8 wave tiles in the shard, so that the
 * counters can number its pixel slots (PT_ERR_INVALID beyond); the scene's device must be current.
 * The counters are cleared at the start of every call.  All calls on one GPU share that GPU's
 * array, whatever the scene or stream: calls that overlap on one GPU race on it and trace pixels
 * twice or not at all -- run them one after another (one stream, or events between streams). */
#define PT_NOALLOC_MAX_WAVE_TILES (1 << 25)
int  pt_render_device_noalloc(PtScene* scene, const PtRenderParams* params, void* d_out_rgb, void* hip_stream);

/* Closest hit of world->hit(ray, 0.001f, FLT_MAX) (src/ptMain.cu:104) for n rays:
 * prim_id = winning shape index (or PT_ID_MEDIUM_BASE+m, or PT_ID_MISS), t = rec.t.
 * Host pointers; blocking. */
int  pt_trace_closest(PtScene* scene, const PtRay* rays, size_t n, int32_t* prim_id, float* t);

/* Fills *stats with the timings of the scene's most recent render call; waits for it. */
int  pt_render_stats(PtScene* scene, PtRenderStats* stats);

/* Counters of the most recent render on this scene that ran on the staged pipeline with PtRenderParams.stage_stats
 * set (PT_ERR_INVALID when none has). */
int  pt_render_stage_stats(PtScene* scene, PtStageStats* stats);

/* One call, several GPUs of this process: the scene is created on every device of device_ids,
 * each traces the 8x8-tile shard (shard_index = its position in device_ids) concurrently, and the
 * shards are added on the host; the image is the one-device image bit for bit.  The call a
 * single-process host like the reference's main() (src/ptMain.cu:579-653, one device) makes
 * instead of running one process per GPU.  params->shard_* and batch_* must be unset. */
int  pt_render_on_devices(const PtSceneDesc* desc, const PtRenderParams* params, const int32_t* device_ids,
                          int32_t n_devices, float* out_rgb);

/* frame[i] = sqrt(max(0, frame[i] * (1 / total_samples))) over n_floats device floats: the
 * averaging and gamma of render_pixel (src/ptMain.cu:163-166) for a frame of raw sums. */
int  pt_frame_finish_device(void* d_frame, size_t n_floats, int32_t total_samples, void* hip_stream);

/* ---- progressive accumulation ----------------------------------------------------------------
 * The reference renders all ns samples of a pixel in one kernel and only counts finished pixels
 * for a progress display (int* progress, src/ptMain.cu:171,186-188; the poller at :616-640 is
 * commented out).  Here a render may be cut into passes of whole samples: between passes each
 * pixel's PCG state and running sum wait in a PtAccumPixel array in device memory, so the
 * passes add up to exactly the image of one pt_render_device call, a host can show the estimate
 * after every pass (out holds sqrt(sum / samples so far), src/ptMain.cu:163-166), and the array
 * can be copied out and back (pt_memcpy_to_host / pt_memcpy_to_device) to checkpoint and resume
 * a long render.  Passes run on the fused pipeline with one pixel per lane. */
typedef struct PtAccumPixel {
    uint32_t rng_lo, rng_hi;   /* PcgRng state (include/ptRNG.h:96-134); its increment follows from the pixel index */
    float    sum[3];           /* accumCol so far (src/ptMain.cu:156-161) */
    uint32_t samples;          /* samples in sum */
} PtAccumPixel;                /* 24 bytes */

/* Bytes of the accumulator array for an nx*ny frame (device memory, caller-allocated). */
size_t pt_accum_bytes(int32_t nx, int32_t ny);
/* Samples [first_sample, first_sample + n_samples) of every pixel this shard owns; params->ns is
 * the total the render is heading for.  first_sample == 0 clears d_accum first; otherwise d_accum
 * must hold the state left by the passes up to first_sample.  d_out_rgb receives the estimate.
 * The launch is queued; a pass that finds an owned pixel whose PtAccumPixel.samples differs from
 * first_sample (a stale, zero-filled or foreign array) raises a flag that the next pt_render_stats
 * on the scene -- which waits for the launch -- reports as PT_ERR_INVALID, and pt_render_progressive
 * checks after every pass. */
int  pt_render_pass(PtScene* scene, const PtRenderParams* params, void* d_accum,
                    int32_t first_sample, int32_t n_samples, void* d_out_rgb, void* hip_stream);
/* Whole render in passes of pass_samples with a callback after each (samples done, total, user);
 * a non-zero return stops the render: out_rgb then holds the estimate so far and the call
 * returns PT_ERR_CANCELLED. */
typedef int (*PtProgressFn)(int32_t samples_done, int32_t samples_total, void* user);
int  pt_render_progressive(PtScene* scene, const PtRenderParams* params, float* out_rgb,
                           int32_t pass_samples, PtProgressFn progress, void* user);

/* Copies the library-built LBVH (64-byte nodes: two child boxes and two child links) to the host:
 * returns the node count, or a negative error; nodes may be NULL to query the count. */
int  pt_scene_lbvh_nodes(PtScene* scene, void* nodes, size_t capacity_bytes);
/* The form of the tree the scene's searches walk: 0 none (the list, or the reference's own BVHs, as they
 * are), 2 the binary nodes pt_scene_lbvh_nodes returns, 4 the same tree with every second level folded
 * away (PT_ACCEL_LBVH_WIDE).  Diagnostic; no counterpart in the reference. */
int  pt_scene_lbvh_form(PtScene* scene);
/* Copies the 4-wide form of the tree to the host, as the searches read it: 64-byte nodes of 16 words --
 * 0-2 the corner of the children's common box (floats), 3 the three step exponents (byte a: the step of axis
 * a is 2^(byte - 127)), 4-6 the children's low planes on x, y, z (byte c = child c, in steps from the
 * corner), 7-9 their high planes, 10-13 the links (>= 0 wide node, < 0 ~shape, 0x7fffffff none), 14-15
 * unused.  Wide node k stands for the k-th binary node of even depth.  Returns the node count (0 when the
 * scene walks no such form), or a negative error; nodes may be NULL to query the count.  Diagnostic. */
int  pt_scene_lbvh_wide_nodes(PtScene* scene, void* nodes, size_t capacity_bytes);

/* The reference's hard-coded scene builders (src/ptMain.cu:192-421) restated as
 * description factories.  name is one of "simple_spheres", "simple_light", "random",
 * "cornell", "cornell_spheres", "final", "trimesh".  The returned description and all its
 * arrays are owned by the library until pt_scene_desc_free().  earth_rgb/earth_nx/earth_ny
 * optionally supply the decoded earthmap.jpg texels for "final" (may be NULL/0: the earth
 * sphere is then left out, as in the reference's device branch, src/ptMain.cu:396-402). */
int  pt_scene_desc_builtin(const char* name, float aspect,
                           const uint8_t* earth_rgb, int32_t earth_nx, int32_t earth_ny,
                           PtSceneDesc** out_desc);
/* "trimesh" with an explicit grid resolution: (n+1)^2 vertices, 2*n*n triangles. */
int  pt_scene_desc_trimesh(int32_t grid_n, float aspect, PtSceneDesc** out_desc);
void pt_scene_desc_free(PtSceneDesc* desc);

/* The reference's scene byte stream (src/ptStream.cu:56-90) -> description.  Replaces what
 * allocate_world_kernel does with it on the device: Hitable::Create for the world and for the
 * light shapes, Camera::Create, AmbientLight::Create, in that order (src/ptMain.cu:423-430).
 * data/size is the buffer main() filled with world->serialize ... ambientLight->serialize
 * (src/ptMain.cu:560-572).  Nested lists flatten into the world list, FlipNormals / Translate /
 * RotateY become the leaf's flag and wrapper chain, a Box record whose child is the constructor's
 * six rectangles becomes one BOX shape, identical material and texture records are merged.
 * A ConstantAmbient arrives black, as it does on the reference's device (its deserialize never
 * reads the colour, include/ptAmbientLight.h:65-73), unless PT_STREAM_KEEP_AMBIENT_COLOR is set.
 * Errors: PT_ERR_INVALID for a malformed stream, PT_ERR_UNSUPPORTED for a ConstantMedium under
 * a wrapper.  Free the result with pt_scene_desc_free(). */
#define PT_STREAM_KEEP_AMBIENT_COLOR 1u
int  pt_scene_desc_from_stream(const void* data, size_t size, uint32_t flags, PtSceneDesc** out_desc);

/* Description -> the same byte stream, as main()'s serialize sequence writes it
 * (src/ptMain.cu:560-572): world list, light shapes or null, camera, ambient light or null.
 * buffer may be NULL to learn the size; *out_size is always the full size;
 * section_sizes (may be NULL) receives the byte counts of those four sections.
 * PT_ERR_UNSUPPORTED for a world that holds a BVH: the reference's BVH::serialize writes
 * nothing (src/ptBVH.cu:97-100). */
int  pt_scene_desc_to_stream(const PtSceneDesc* desc, void* buffer, size_t capacity, size_t* out_size,
                             size_t section_sizes[4]);

/* medianFilter3x3 (size 3) / medianFilter2x2 (size 2) of the reference (src/ptMain.cu:726-788),
 * which its -m flag runs over outImage before writeImage (:684-693): every pixel becomes the
 * neighbour of rank (count-1)/2 by squared length, edges clamped; equal lengths keep the
 * neighbourhood's own order, as glibc's stable qsort leaves them.  in and out are nx*ny*3
 * floats and must not overlap.  The _device form takes device pointers and a hipStream_t. */
int  pt_median_filter(const float* in_rgb, float* out_rgb, int32_t nx, int32_t ny, int32_t size);
int  pt_median_filter_device(const void* d_in_rgb, void* d_out_rgb, int32_t nx, int32_t ny, int32_t size,
                             void* hip_stream);

/* stbi_load("earthmap.jpg", &nx, &ny, &nz, 0) of the reference's final() (src/ptMain.cu:396-402; vendored
 * src/stb_image.h v2.15): a sequential (baseline) JPEG file in memory to nx*ny RGB8 texels, row 0 on top, the
 * bytes that decoder produces (same inverse DCT, chroma filters and colour conversion; a one-component file
 * comes out grey in three channels).  Call with rgb == NULL for the size, then with nx*ny*3 bytes of room.
 * Progressive and arithmetic-coded files are PT_ERR_INVALID.  Host only; the texels go into
 * PtSceneDesc.texels (pt_scene_desc_builtin's earth_rgb). */
int  pt_image_decode_jpeg(const void* data, size_t bytes, int32_t* nx, int32_t* ny, uint8_t* rgb, size_t capacity);

/* Number of visible HIP devices (0 when there is none); never fails. */
int  pt_device_count(void);

/* ---- device plumbing for hosts without a HIP binding of their own ------------------------ */
/* The reference does the same with cudaMalloc / cudaMemcpy / cudaDeviceSynchronize / cudaFree
 * inline in main() (src/ptMain.cu:593-600,637,649-652). */
int  pt_set_device(int device);
int  pt_device_synchronize(void);
int  pt_device_malloc(size_t bytes, void** out_device_ptr);
void pt_device_free(void* device_ptr);
int  pt_memcpy_to_host(void* host_dst, const void* device_src, size_t bytes);
int  pt_memcpy_to_device(void* device_dst, const void* host_src, size_t bytes);

/* ---- one-node multi-GPU frame exchange over RCCL ------------------------------------------ */
/* No counterpart in the reference (single device, no communication library).  One process per
 * GPU: every rank renders its shard with pt_render_device (shard_index = rank, shard_count =
 * world) into a zero-filled full frame, then pt_comm_reduce_frame sums the frames onto `root`
 * over xGMI.  Each pixel is non-zero on exactly one rank, so the sum reproduces the
 * single-GPU frame bit for bit.  id128 is RCCL's 128-byte unique id: rank 0 creates it and the
 * host distributes it to the other ranks by any means before pt_comm_init. */
typedef struct PtComm PtComm;
int  pt_comm_unique_id(void* id128);
int  pt_comm_init(const void* id128, int rank, int world, PtComm** out_comm);
int  pt_comm_reduce_frame(PtComm* comm, void* d_frame, size_t n_floats, int root, void* hip_stream);
/* Host-side synchronisation of the ranks through the same communicator, so that a multi-process host
 * needs no second communication library: a barrier (returns once every rank has called it and the
 * work queued on hip_stream before it has finished) and the maximum of one double over the ranks
 * (a benchmark's max-over-ranks clock).  Both block. */
int  pt_comm_barrier(PtComm* comm, void* hip_stream);
int  pt_comm_allreduce_max_f64(PtComm* comm, double* inout, void* hip_stream);
void pt_comm_destroy(PtComm* comm);

/* ---- diagnostics ---------------------------------------------------------------------------- */
/* Evaluates one of the library's libm replacements on n floats: fn 0 sin, 1 cos, 2 log, 3 x^5
 * (the reference's Sin/Cos/Log/Pow wrappers, include/ptMath.h:34-45); fn 4 and 5 are the sine and
 * the cosine from the combined evaluation the direction samplers use, equal to fn 0 and 1; fn 6 asin; fn 7 atan2,
 * where x holds n (y, x) pairs, 2 * n floats (MovingSphere's get_uv, include/ptSphere.h:22-28).  on_device != 0 runs the
 * device code on the current HIP device, otherwise the same source compiled for the host.
 * fn 8 to 11 expose the kernels' shortened arithmetic next to the compiler's full sequences, for tests
 * of the ranges the kernels use them in: 8 the square root for arguments that are +0 or at least
 * 2^-96, 9 sqrtf; 10 the quotient n / d for 2^-30 <= |d| <= 2^60 and 2^-103 <= |n| <= 2^60, 11 n / d,
 * both on n (numerator, denominator) pairs.  On the host all four are the plain operation.
 * fn 12 and 13: sine and cosine from the combined evaluation in the form the direction samplers use
 * for their angles (arguments in [0, 120) only: the large-argument reduction is left out). */
int  pt_libm_eval(int fn, const float* x, size_t n, float* out, int on_device);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
