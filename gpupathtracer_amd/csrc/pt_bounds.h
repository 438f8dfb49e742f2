// Conservative bounding boxes of leaf shapes, shared by the LBVH builder (device) and the flattened
// world list (host).  "Conservative" means: padded so that no hit the reference's leaf tests can
// report lies outside, whatever their rounding.
#ifndef PT_BOUNDS_H
#define PT_BOUNDS_H

#include <cfloat>
#include <cmath>

#include "../../include/ptcore.h"
#include "pt_libm.h"   // PT_HD

namespace ptb {

struct Box { float mn[3], mx[3]; };

PT_HD void grow(Box& b, float x, float y, float z)
{
    b.mn[0] = fminf(b.mn[0], x); b.mn[1] = fminf(b.mn[1], y); b.mn[2] = fminf(b.mn[2], z);
    b.mx[0] = fmaxf(b.mx[0], x); b.mx[1] = fmaxf(b.mx[1], y); b.mx[2] = fmaxf(b.mx[2], z);
}

// Conservative world-space box of one shape.  times[] are the ray times that can occur
// (shutter open, shutter close, and 0 for specular bounces, include/ptMaterial.h:137,217,221).
PT_HD Box leaf_box(const PtShape* shapes, const PtXform* xforms, int index, const float* times)
{
    const PtShape s = shapes[index];
    const float* p = s.p;
    Box b;
    for (int a = 0; a < 3; a++) { b.mn[a] = FLT_MAX; b.mx[a] = -FLT_MAX; }
    float pad = 0.0f;
    switch (s.type) {
    case PT_SHAPE_SPHERE:
        grow(b, p[0] - p[3], p[1] - p[3], p[2] - p[3]);
        grow(b, p[0] + p[3], p[1] + p[3], p[2] + p[3]);
        pad = 1e-3f * fabsf(p[3]);
        break;
    case PT_SHAPE_MOVING_SPHERE:
        for (int k = 0; k < 3; k++) {
            const float f = (times[k] - p[6]) / (p[7] - p[6]);
            const float cx = p[0] + f * (p[3] - p[0]), cy = p[1] + f * (p[4] - p[1]), cz = p[2] + f * (p[5] - p[2]);
            grow(b, cx - p[8], cy - p[8], cz - p[8]);
            grow(b, cx + p[8], cy + p[8], cz + p[8]);
        }
        pad = 1e-3f * fabsf(p[8]);
        break;
    case PT_SHAPE_RECT_XY: grow(b, p[0], p[2], p[4] - 1e-4f); grow(b, p[1], p[3], p[4] + 1e-4f); break;
    case PT_SHAPE_RECT_XZ: grow(b, p[0], p[4] - 1e-4f, p[2]); grow(b, p[1], p[4] + 1e-4f, p[3]); break;
    case PT_SHAPE_RECT_YZ: grow(b, p[4] - 1e-4f, p[0], p[2]); grow(b, p[4] + 1e-4f, p[1], p[3]); break;
    case PT_SHAPE_BOX: grow(b, p[0], p[1], p[2]); grow(b, p[3], p[4], p[5]); break;
    case PT_SHAPE_TRIANGLE:
        grow(b, p[0], p[1], p[2]); grow(b, p[3], p[4], p[5]); grow(b, p[6], p[7], p[8]);
        pad = 1e-4f;
        break;
    default: grow(b, 0, 0, 0);
    }
    // wrappers are stored outermost first; undo them innermost first on the 8 corners
    for (int k = s.xform_count - 1; k >= 0; k--) {
        const PtXform x = xforms[s.xform_first + k];
        Box nb;
        for (int a = 0; a < 3; a++) { nb.mn[a] = FLT_MAX; nb.mx[a] = -FLT_MAX; }
        for (int c = 0; c < 8; c++) {
            float cx = (c & 1) ? b.mx[0] : b.mn[0], cy = (c & 2) ? b.mx[1] : b.mn[1], cz = (c & 4) ? b.mx[2] : b.mn[2];
            if (x.type == PT_XFORM_TRANSLATE) { cx += x.v[0]; cy += x.v[1]; cz += x.v[2]; }
            else { const float nx = x.v[1] * cx + x.v[0] * cz, nz = -x.v[0] * cx + x.v[1] * cz; cx = nx; cz = nz; }
            grow(nb, cx, cy, cz);
        }
        b = nb;
    }
    // absolute + relative slack so rounding in the leaf tests can never leave the box
    for (int a = 0; a < 3; a++) {
        const float m = fmaxf(fabsf(b.mn[a]), fabsf(b.mx[a]));
        const float e = pad + 1e-5f * m + 1e-6f;
        b.mn[a] -= e; b.mx[a] += e;
    }
    return b;
}


}  // namespace ptb
#endif
