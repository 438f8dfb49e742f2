// Compiles the fused radiance kernels and the closest-hit kernel of these specialisations (pt_kernels.h).
#include "pt_kernels.h"

template struct FusedLaunch<ptk::M_FINAL_LBVH4, ptd::AnyWorld>;
