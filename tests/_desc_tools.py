"""Scene descriptions assembled in Python (ctypes arrays owned by the object): lets tests build
inputs the built-in factories do not produce.  Test infrastructure only."""
import ctypes as C

from gpupathtracer_amd import _abi as A

_ARRAYS = (("shapes", "n_shapes", A.PtShape), ("tri_uvs", "n_tri_uvs", A.PtTriUV), ("xforms", "n_xforms", A.PtXform),
           ("materials", "n_materials", A.PtMaterial), ("textures", "n_textures", A.PtTexture), ("texels", "n_texels", C.c_uint8),
           ("media", "n_media", A.PtMedium), ("bvh_nodes", "n_bvh_nodes", A.PtBvhNode), ("bvhs", "n_bvhs", A.PtBvh),
           ("world", "n_world", A.PtObject), ("lights", "n_lights", C.c_int32))


class OwnedDesc:
    """A deep, editable copy of a description; .ptr is what the library and the oracle take."""

    def __init__(self, src):
        s = src.c if hasattr(src, "c") else src
        self.lists = {}
        for name, cnt, typ in _ARRAYS:
            n = getattr(s, cnt)
            arr = getattr(s, name)
            items = []
            for i in range(n):
                if typ in (C.c_uint8, C.c_int32):
                    items.append(arr[i])
                else:
                    item = typ()
                    C.memmove(C.byref(item), C.byref(arr[i]), C.sizeof(typ))
                    items.append(item)
            self.lists[name] = items
        self.scalars = {k: getattr(s, k) for k in ("lights_is_list", "ambient_type", "accel")}
        self.ambient_color = list(s.ambient_color)
        self.camera = list(s.camera)
        self._keep = None
        self.desc = None

    def finalize(self):
        d = A.PtSceneDesc()
        keep = []
        for name, cnt, typ in _ARRAYS:
            items = self.lists[name]
            arr = (typ * max(1, len(items)))(*items)
            keep.append(arr)
            setattr(d, cnt, len(items))
            setattr(d, name, C.cast(arr, C.POINTER(typ)))
        for k, v in self.scalars.items():
            setattr(d, k, v)
        d.ambient_color = (C.c_float * 3)(*self.ambient_color)
        d.camera = (C.c_float * 24)(*self.camera)
        self._keep = keep
        self.desc = d
        return self

    @property
    def ptr(self):
        return C.pointer(self.desc)

    @property
    def c(self):
        return self.desc


def look_at_camera(frm, to, vup, vfov, aspect, aperture, focus, t0=0.0, t1=1.0):
    """The 24 camera floats in the layout of include/ptCamera.h:43-49, derived the way the reference's
    constructor derives them (src/ptCamera.cu:27-45) but in numpy float32: an input for both sides of
    a parity test, not a claim about the constructor's bits."""
    import numpy as np
    f = np.float32
    frm, to, vup = (np.asarray(v, f) for v in (frm, to, vup))
    half_h = f(np.tan(f(vfov) * f(np.pi) / f(180) / f(2)))
    half_w = f(aspect) * half_h
    w = frm - to
    w = w / f(np.linalg.norm(w))
    u = np.cross(vup, w).astype(f)
    u = u / f(np.linalg.norm(u))
    v = np.cross(w, u).astype(f)
    fd = f(focus)
    llc = frm - half_w * fd * u - half_h * fd * v - fd * w
    cam = list(frm) + list(llc) + list(f(2) * half_w * fd * u) + list(f(2) * half_h * fd * v) + list(u) + list(v) + list(w)
    cam += [t0, t1, aperture / 2.0]
    return [float(f(x)) for x in cam]
