"""ctypes mirror of include/ptcore.h (struct layouts and enum values)."""
import ctypes as C

# enum PtShapeType
SHAPE_SPHERE, SHAPE_MOVING_SPHERE, SHAPE_RECT_XY, SHAPE_RECT_XZ, SHAPE_RECT_YZ, SHAPE_BOX, SHAPE_TRIANGLE = range(1, 8)
FLIP_NORMAL = 1
XFORM_TRANSLATE, XFORM_ROTATE_Y = 1, 2
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = range(1, 6)
TEX_CONSTANT, TEX_CHECKER, TEX_NOISE, TEX_IMAGE = range(1, 5)
OBJ_SHAPE, OBJ_MEDIUM, OBJ_BVH = 1, 2, 3
AMBIENT_NONE, AMBIENT_CONSTANT, AMBIENT_SKY = 0, 1, 2
ACCEL_LIST, ACCEL_LBVH, ACCEL_LBVH_BINARY, ACCEL_LBVH_WIDE = 0, 1, 2, 3
PIPELINE_DEFAULT, PIPELINE_FUSED, PIPELINE_WAVEFRONT = 0, 1, 2
SCHEDULE_AUTO, SCHEDULE_TILES, SCHEDULE_REFILL, SCHEDULE_WALK, SCHEDULE_CHUNKS = 0, 1, 2, 3, 4
KERNELS_RECTANGLES, KERNELS_SPHERES, KERNELS_TRIANGLES, KERNELS_FINAL, KERNELS_ALL = 1, 2, 3, 4, 5
STREAM_KEEP_AMBIENT_COLOR = 1
ID_MISS = -1
ID_MEDIUM_BASE = 0x40000000

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_OOM, ERR_CANCELLED = 0, -1, -2, -3, -4, -5, -6


class PtShape(C.Structure):
    _fields_ = [("type", C.c_int32), ("flags", C.c_int32), ("material", C.c_int32),
                ("xform_first", C.c_int32), ("xform_count", C.c_int32), ("tri_uv", C.c_int32),
                ("p", C.c_float * 10)]


class PtTriUV(C.Structure):
    _fields_ = [("uv0", C.c_float * 2), ("uv1", C.c_float * 2), ("uv2", C.c_float * 2)]


class PtXform(C.Structure):
    _fields_ = [("type", C.c_int32), ("v", C.c_float * 3)]


class PtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("albedo", C.c_float * 3), ("param", C.c_float)]


class PtTexture(C.Structure):
    _fields_ = [("type", C.c_int32), ("odd", C.c_int32), ("even", C.c_int32), ("nx", C.c_int32),
                ("ny", C.c_int32), ("texel_offset", C.c_int32), ("color", C.c_float * 3), ("scale", C.c_float)]


class PtObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("index", C.c_int32)]


class PtMedium(C.Structure):
    _fields_ = [("boundary", C.c_int32), ("density", C.c_float), ("material", C.c_int32), ("reserved", C.c_int32)]


class PtBvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("left", C.c_int32), ("right", C.c_int32)]


class PtBvh(C.Structure):
    _fields_ = [("root", C.c_int32), ("xform_first", C.c_int32), ("xform_count", C.c_int32), ("reserved", C.c_int32)]


class PtSceneDesc(C.Structure):
    _fields_ = [
        ("n_shapes", C.c_int32), ("shapes", C.POINTER(PtShape)),
        ("n_tri_uvs", C.c_int32), ("tri_uvs", C.POINTER(PtTriUV)),
        ("n_xforms", C.c_int32), ("xforms", C.POINTER(PtXform)),
        ("n_materials", C.c_int32), ("materials", C.POINTER(PtMaterial)),
        ("n_textures", C.c_int32), ("textures", C.POINTER(PtTexture)),
        ("n_texels", C.c_int32), ("texels", C.POINTER(C.c_uint8)),
        ("n_media", C.c_int32), ("media", C.POINTER(PtMedium)),
        ("n_bvh_nodes", C.c_int32), ("bvh_nodes", C.POINTER(PtBvhNode)),
        ("n_bvhs", C.c_int32), ("bvhs", C.POINTER(PtBvh)),
        ("n_world", C.c_int32), ("world", C.POINTER(PtObject)),
        ("n_lights", C.c_int32), ("lights", C.POINTER(C.c_int32)),
        ("lights_is_list", C.c_int32),
        ("ambient_type", C.c_int32),
        ("ambient_color", C.c_float * 3),
        ("camera", C.c_float * 24),
        ("accel", C.c_int32),
    ]


class PtRenderParams(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("ns", C.c_int32), ("max_depth", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32),
                ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("pipeline", C.c_int32), ("schedule", C.c_int32), ("batch_index", C.c_int32), ("batch_count", C.c_int32),
                ("raw_sum", C.c_int32), ("chunk_samples", C.c_int32), ("stage_stats", C.c_int32), ("slot_block_tiles", C.c_int32)]


class PtRay(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("time", C.c_float), ("seed", C.c_uint32)]


class PtStageStats(C.Structure):
    _fields_ = [("groups", C.c_uint64), ("rounds", C.c_uint64), ("extend_lanes", C.c_uint64), ("extend_waves", C.c_uint64),
                ("shade_lanes", C.c_uint64 * 4), ("shade_waves", C.c_uint64 * 4), ("extend_ticks", C.c_uint64), ("shade_ticks", C.c_uint64),
                ("timed_groups", C.c_uint64), ("reserved", C.c_uint64)]


class PtRenderStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("total_ms", C.c_float), ("n_launches", C.c_int32),
                ("pipeline", C.c_int32), ("paths", C.c_uint64), ("schedule", C.c_int32), ("signature_walk", C.c_int32),
                ("kernel_family", C.c_int32), ("n_chunks", C.c_int32), ("reserved", C.c_uint64 * 2)]


assert C.sizeof(PtShape) == 64
assert C.sizeof(PtRay) == 32
assert C.sizeof(PtBvhNode) == 32
assert C.sizeof(PtRenderParams) == 64

# Every symbol include/ptcore.h declares.
EXPORTS = [
    "pt_last_error", "pt_scene_create", "pt_scene_destroy", "pt_render", "pt_render_device", "pt_render_device_noalloc",
    "pt_trace_closest", "pt_render_stats", "pt_render_stage_stats", "pt_render_on_devices", "pt_frame_finish_device", "pt_accum_bytes", "pt_render_pass", "pt_render_progressive",
    "pt_memcpy_to_device", "pt_scene_lbvh_nodes", "pt_scene_lbvh_form", "pt_scene_lbvh_wide_nodes", "pt_scene_desc_builtin",
    "pt_scene_desc_trimesh", "pt_scene_desc_free", "pt_scene_desc_from_stream", "pt_scene_desc_to_stream",
    "pt_median_filter", "pt_median_filter_device", "pt_image_decode_jpeg", "pt_device_count",
    "pt_set_device", "pt_device_synchronize", "pt_device_malloc", "pt_device_free", "pt_memcpy_to_host",
    "pt_comm_unique_id", "pt_comm_init", "pt_comm_reduce_frame", "pt_comm_barrier", "pt_comm_allreduce_max_f64", "pt_comm_destroy", "pt_libm_eval",
]
