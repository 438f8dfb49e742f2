// Launchers of the fused radiance kernels (one lane per pixel, stages fused, path regeneration in-lane)
// and of the closest-hit kernel, one set per (feature mask, world walk).  Declarations only: the
// kernels and the launchers' bodies are in pt_kernels_impl.h, which only the translation units that
// compile a specialisation include (pt_k_*.hip, one or a few specialisations each, so that the
// library builds in parallel).  Code that merely launches them must not see the bodies: it would
// instantiate -- and then run -- its own copy of every kernel.
#ifndef PT_KERNELS_H
#define PT_KERNELS_H

#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "pt_internal.h"

template <uint32_t F, class W>
struct FusedLaunch {
    // chunk_rec: five 64-bit words per pixel, all zero at launch, when the persistent-lane schedules trace a pixel's
    // samples in chunks (ra.n_chunks > 1 with ra.refill != 0); fault: raised when a lane gives up waiting for a record
    static void render(const ptd::SceneView& sv, const RenderArgs& ra, float* out, unsigned* next_slot, unsigned long long* chunk_rec, int* fault,
                       hipStream_t st);
    // words: [0] the ticket counter, [1 ..] one progress word per group of four wave tiles, all zero at launch
    static void chunks(const ptd::SceneView& sv, const RenderArgs& ra, float* out, unsigned* words, unsigned long long* tile_states, int* fault, hipStream_t st);
    static void pass(const ptd::SceneView& sv, const RenderArgs& ra, PtAccumPixel* acc, int first, float* out, int* bad_resume, hipStream_t st);
    static void trace(const ptd::SceneView& sv, const PtRay* rays, size_t n, int32_t* ids, float* ts, hipStream_t st);
};

#endif
