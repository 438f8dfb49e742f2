// Morton-code LBVH builder (device side); replaces the reference's BVH::BVH + quickSort
// (src/ptBVH.cu:15-43, src/ptQuickSort.cu:19-81) for library-accelerated world lists.
#ifndef PT_LBVH_H
#define PT_LBVH_H

#include "../../include/ptcore.h"

namespace pt {

// Builds the tree over the shapes named by desc->world (all entries must be PT_OBJ_SHAPE with
// strictly increasing shape indices).  d_shapes / d_xforms are the already-uploaded device
// copies.  On success *d_nodes is a hipMalloc'ed array of *n_nodes 64-byte nodes and *root is
// the root link (>= 0 node index, < 0 ~shape).  Returns a PtError.
// d_wide (may be null): receives the same tree folded to four children per node (64-byte LbvhNode4,
// the binary nodes of even depth stored without gaps, in the binary nodes' order).  depth (may be null):
// levels of internal nodes of the binary tree.  area_sweep: worlds of up to 32 768 leaves take the
// shape of their tree from a surface-area sweep on the host instead of the Morton order.
int lbvh_build(const PtSceneDesc* desc, const PtShape* d_shapes, const PtXform* d_xforms,
               void** d_nodes, int* n_nodes, int* root, void** d_wide, int* depth, bool area_sweep);

// The same tree over an explicit list of n >= 2 leaves (PT_OBJ_SHAPE entries in any order; the
// caller supplies the tie order the walk needs).  time0 / time1: the shutter interval moving
// spheres are bounded over (PtSceneDesc::camera[21], [22]).
int lbvh_build_leaves(const PtObject* leaves, int n, float time0, float time1, const PtShape* d_shapes, const PtXform* d_xforms,
                      void** d_nodes, int* n_nodes, int* root, void** d_wide, int* depth, bool area_sweep);

}  // namespace pt

#endif
