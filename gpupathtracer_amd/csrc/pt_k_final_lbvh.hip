// Compiles the fused radiance kernels and the closest-hit kernel of these specialisations (pt_kernels_impl.h).
#include "pt_kernels_impl.h"

template struct FusedLaunch<ptk::M_FINAL_LBVH, ptd::AnyWorld>;
