// Host-side internals shared by the C-ABI translation units.
#ifndef PT_HOST_H
#define PT_HOST_H

#include <cstdint>
#include <vector>

#include "../../include/ptcore.h"

namespace pt {

// Owns the arrays a PtSceneDesc points at.  `desc` must stay the first member:
// pt_scene_desc_free() recovers the storage from the PtSceneDesc pointer.
struct DescStorage {
    PtSceneDesc desc{};
    std::vector<PtShape> shapes;
    std::vector<PtTriUV> tri_uvs;
    std::vector<PtXform> xforms;
    std::vector<PtMaterial> materials;
    std::vector<PtTexture> textures;
    std::vector<uint8_t> texels;
    std::vector<PtMedium> media;
    std::vector<PtBvhNode> bvh_nodes;
    std::vector<PtBvh> bvhs;
    std::vector<PtObject> world;
    std::vector<int32_t> lights;
    void finalize();   // points desc at the vectors
};

void shape_bounds(const PtShape* shapes, const PtXform* xforms, int index, float* bmin, float* bmax);

void set_error(const char* msg);

}  // namespace pt

#endif
