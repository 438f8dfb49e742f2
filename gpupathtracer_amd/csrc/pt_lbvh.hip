// LBVH construction on the device: leaf boxes -> 30-bit Morton codes -> stable LSD radix sort
// (wave64 ballot ranking) -> Karras radix tree -> bottom-up box fitting -> 64-byte nodes that
// hold both children's boxes, so one node fetch decides both descents.  Worlds of up to 32 768
// leaves take the shape of their tree from a surface-area sweep on the host instead of the
// Morton order (sah_build); fitting and node emission are the same kernels.
//
// The reference builds its BVH on the host with a random-axis median split around a Lomuto
// quicksort (src/ptBVH.cu:15-43, src/ptQuickSort.cu:19-81), quadratic on sorted or tied keys.
// Only the tree changes here; what a ray hits is still decided by the reference's leaf tests
// and tie rules, so any conservative tree gives the same answer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

#include "pt_bounds.h"
#include "pt_device.h"
#include "pt_host.h"
#include "pt_lbvh.h"

using ptd::LbvhNode;

namespace {

using ptb::Box;
using ptb::leaf_box;

// float <-> order-preserving unsigned, for atomicMin / atomicMax on floats
__device__ __forceinline__ unsigned f2o(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float o2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void k_leaf_boxes(const PtShape* shapes, const PtXform* xforms, const PtObject* world, int n,
                             float t0, float t1, Box* boxes, unsigned* cbounds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // the bounds of the centroids: the wave's lanes fold theirs together first, one lane adds the wave's to the scene's
    // (an atomic per lane and word was two thirds of this kernel's 2.1 ms on two million triangles)
    unsigned lo[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, hi[3] = { 0u, 0u, 0u };
    if (i < n) {
        const float times[3] = { t0, t1, 0.0f };
        const Box b = leaf_box(shapes, xforms, world[i].index, times);
        boxes[i] = b;
        for (int a = 0; a < 3; a++) { const unsigned c = f2o(0.5f * (b.mn[a] + b.mx[a])); lo[a] = c; hi[a] = c; }
    }
    for (int off = 32; off >= 1; off >>= 1)
        for (int a = 0; a < 3; a++) {
            lo[a] = min(lo[a], (unsigned)__shfl_xor((int)lo[a], off));
            hi[a] = max(hi[a], (unsigned)__shfl_xor((int)hi[a], off));
        }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) { atomicMin(&cbounds[a], lo[a]); atomicMax(&cbounds[3 + a], hi[a]); }
}

__device__ __forceinline__ unsigned spread3(unsigned v)     // 10 bits -> every third bit
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// key = morton30 << 32 | leaf number: unique, so the radix tree needs no tie handling
__global__ void k_morton(const Box* boxes, int n, const unsigned* cbounds, uint64_t* keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned q[3];
    for (int a = 0; a < 3; a++) {
        const float lo = o2f(cbounds[a]), hi = o2f(cbounds[3 + a]);
        const float c = 0.5f * (boxes[i].mn[a] + boxes[i].mx[a]);
        const float ext = hi - lo;
        float f = ext > 0.0f ? (c - lo) / ext : 0.0f;
        f = fminf(fmaxf(f * 1024.0f, 0.0f), 1023.0f);
        q[a] = (unsigned)f;
    }
    const unsigned m = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    keys[i] = ((uint64_t)m << 32) | (unsigned)i;
}

// ---- stable LSD radix sort, 8 bits per pass, one wave per 1024-key tile ------------------------
constexpr int TILE = 1024;

__global__ __launch_bounds__(64) void k_radix_hist(const uint64_t* keys, int n, int shift, unsigned* hist, int n_tiles)
{
    __shared__ unsigned h[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) h[d] = 0;
    __syncthreads();
    const int base = blockIdx.x * TILE;
    for (int j = lane; j < TILE; j += 64) {
        const int i = base + j;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = lane; d < 256; d += 64) hist[(size_t)d * n_tiles + blockIdx.x] = h[d];
}

// ---- exclusive scan of an array of counts, in place, over the whole GPU ---------------------------------
// Three launches: every workgroup sums its tile of SCAN_TILE counts (16 bytes a lane, coalesced); one workgroup scans the
// tiles' sums; every workgroup scans its tile in LDS and adds what precedes it.  (One workgroup walking the whole array --
// the builder's first form -- took 0.84 ms per radix pass over the 500 000 counts of a 2 M-triangle field and 4.3 ms over
// its two million node flags, a third of the build's kernel time.)
constexpr int SCAN_TILE = 4096;                               // 1024 lanes x 4 counts
__device__ __forceinline__ uint4 scan_load4(const unsigned* data, int n, int i)
{
    if (i + 3 < n) return *reinterpret_cast<const uint4*>(data + i);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (i < n) v.x = data[i];
    if (i + 1 < n) v.y = data[i + 1];
    if (i + 2 < n) v.z = data[i + 2];
    return v;
}
// inclusive scan of one value per lane over the workgroup's 1024 lanes; every lane calls it
__device__ __forceinline__ unsigned block_scan_1024(unsigned v, unsigned* part)
{
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned add = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    return part[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_scan_tile_sums(const unsigned* data, int n, unsigned* sums)
{
    __shared__ unsigned part[1024];
    const uint4 v = scan_load4(data, n, blockIdx.x * SCAN_TILE + 4 * (int)threadIdx.x);
    const unsigned incl = block_scan_1024(v.x + v.y + v.z + v.w, part);
    if (threadIdx.x == 1023) sums[blockIdx.x] = incl;
}
// exclusive scan of the tiles' sums by one workgroup (any count: 1024 at a time with a carry); the grand total to *total
__global__ __launch_bounds__(1024) void k_scan_sums(unsigned* sums, int count, unsigned* total)
{
    __shared__ unsigned part[1024];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < count; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned v = (i < count) ? sums[i] : 0u;
        const unsigned incl = block_scan_1024(v, part);
        if (i < count) sums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total != nullptr) *total = carry;
}
__global__ __launch_bounds__(1024) void k_scan_tiles(unsigned* data, int n, const unsigned* sums)
{
    __shared__ unsigned part[1024];
    const int i = blockIdx.x * SCAN_TILE + 4 * (int)threadIdx.x;
    const uint4 v = scan_load4(data, n, i);
    const unsigned mine = v.x + v.y + v.z + v.w;
    const unsigned before = sums[blockIdx.x] + block_scan_1024(mine, part) - mine;
    if (i < n) data[i] = before;
    if (i + 1 < n) data[i + 1] = before + v.x;
    if (i + 2 < n) data[i + 2] = before + v.x + v.y;
    if (i + 3 < n) data[i + 3] = before + v.x + v.y + v.z;
}
// data[0 .. n) becomes its exclusive scan; *total (a device word, may be null) the sum of all.  `sums`: one word per tile.
void exclusive_scan_in_place(unsigned* data, int n, unsigned* sums, unsigned* total)
{
    const int tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(1024), 0, nullptr, data, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, nullptr, sums, tiles, total);
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(1024), 0, nullptr, data, n, sums);
}

// Each 64-key chunk is ranked with ballots: lanes holding the same digit find each other by
// AND-ing the eight per-bit ballots, their rank is the popcount of lower peers.
__global__ __launch_bounds__(64) void k_radix_scatter(const uint64_t* in, uint64_t* out, int n, int shift,
                                                      const unsigned* hist, int n_tiles)
{
    __shared__ unsigned offs[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) offs[d] = hist[(size_t)d * n_tiles + blockIdx.x];
    __syncthreads();
    const int base = blockIdx.x * TILE;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int j = 0; j < TILE; j += 64) {
        const int i = base + j + lane;
        const bool live = i < n;
        const uint64_t key = live ? in[i] : 0;
        const unsigned digit = (unsigned)(key >> shift) & 255u;
        unsigned long long peers = __ballot(live);
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot(live && ((digit >> b) & 1u));
            peers &= ((digit >> b) & 1u) ? m : ~m;
        }
        const unsigned rank = (unsigned)__popcll(peers & lt);
        const unsigned count = (unsigned)__popcll(peers);
        unsigned dst = 0;
        if (live) dst = offs[digit] + rank;
        __syncthreads();
        if (live && rank == count - 1) offs[digit] += count;
        __syncthreads();
        if (live) out[dst] = key;
    }
}

// ---- Karras radix tree ------------------------------------------------------------------------------
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll(keys[i] ^ keys[j]);
}

// internal node i of n-1; leaves are referred to as ~leaf_number in left/right
__global__ void k_karras(const uint64_t* keys, int n, int* left, int* right, int* parent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int lc = (lo == gamma) ? ~gamma : gamma;
    const int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    left[i] = lc; right[i] = rc;
    if (lc >= 0) parent[lc] = i;
    if (rc >= 0) parent[rc] = i;
    if (i == 0) parent[0] = -1;
}

// One sweep of bottom-up fitting: an internal node whose children are both fitted gets its box.
// Sweeps are separate launches, so no in-kernel hand-off between workgroups is needed.
__global__ void k_fit_sweep(const int* left, const int* right, const uint64_t* keys, const Box* leaf_boxes,
                            Box* node_boxes, int* ready, int n_internal, int* n_changed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal || ready[i]) return;
    const int lc = left[i], rc = right[i];
    if ((lc >= 0 && ready[lc] != 1) || (rc >= 0 && ready[rc] != 1)) return;   // 1 = fitted by an earlier sweep
    const Box a = (lc >= 0) ? node_boxes[lc] : leaf_boxes[(unsigned)keys[~lc]];
    const Box b = (rc >= 0) ? node_boxes[rc] : leaf_boxes[(unsigned)keys[~rc]];
    Box u;
    for (int k = 0; k < 3; k++) { u.mn[k] = fminf(a.mn[k], b.mn[k]); u.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    node_boxes[i] = u;
    ready[i] = 2;            // fitted in this sweep; k_fit_commit turns it into 1 for the next one
    atomicAdd(n_changed, 1);
}
__global__ void k_fit_commit(int* ready, int n_internal)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_internal && ready[i] == 2) ready[i] = 1;
}

// final 64-byte nodes: child boxes inline, leaf links rewritten to ~shape_index
__global__ void k_emit(const int* left, const int* right, const uint64_t* keys, const Box* leaf_boxes, const Box* node_boxes,
                       const PtObject* world, int n_internal, LbvhNode* nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal) return;
    const int lc = left[i], rc = right[i];
    const Box a = (lc >= 0) ? node_boxes[lc] : leaf_boxes[(unsigned)keys[~lc]];
    const Box b = (rc >= 0) ? node_boxes[rc] : leaf_boxes[(unsigned)keys[~rc]];
    LbvhNode nd;
    for (int k = 0; k < 3; k++) { nd.lmin[k] = a.mn[k]; nd.lmax[k] = a.mx[k]; nd.rmin[k] = b.mn[k]; nd.rmax[k] = b.mx[k]; }
    nd.left = (lc >= 0) ? lc : ~world[(unsigned)keys[~lc]].index;
    nd.right = (rc >= 0) ? rc : ~world[(unsigned)keys[~rc]].index;
    nd.pad0 = 0; nd.pad1 = 0;
    nodes[i] = nd;
}

// ---- 4-wide collapse ------------------------------------------------------------------------------------
// Every second level of the binary tree is folded into the level above it: a node at even depth
// keeps its slot and takes its grandchildren (or, where a child is a leaf, that leaf) as its up to
// four children; nodes at odd depth disappear.  One 64-byte fetch then decides four descents,
// which halves the chain of dependent node fetches of a walk, and the node is 64 bytes (LbvhNode4: boxes on a byte grid).  Depth = number of parent links to
// the root, found by following them (trees here are a few dozen levels deep).
__global__ void k_depth(const int* parent, int n_internal, int* depth, int* max_depth)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal) return;
    int d = 0;
    for (int p = parent[i]; p >= 0; p = parent[p]) d++;
    depth[i] = d;
    atomicMax(max_depth, d);
}

// The byte grid of one axis of a wide node: corner = the children's common low bound, step = the smallest
// power of two that reaches their common high bound in 255 steps, and per child the last plane at or below
// its low bound and the first at or above its high one -- all checked in double precision, where corner +
// q * step is exact for every box a scene can hold (or off by 2^-53 of itself, far inside the walk's margins).
__device__ void wide_axis(const Box* kid, int n, int a, float& corner, uint8_t& step_exp, uint8_t* lo, uint8_t* hi)
{
    float mn = kid[0].mn[a], mx = kid[0].mx[a];
    for (int c = 1; c < n; c++) { mn = fminf(mn, kid[c].mn[a]); mx = fmaxf(mx, kid[c].mx[a]); }
    corner = mn;
    int e = 1;                                               // biased exponent: 2^-126, the smallest normal step
    {
        int ex = 0;
        const float m = frexpf((mx - mn) / 255.0f, &ex);     // (mx - mn) / 255 = m * 2^ex, 0.5 <= m < 1
        if (m > 0.0f) e = ex + 127;                           // 2^ex >= (mx - mn) / 255
        if (e < 1) e = 1;
        if (e > 254) e = 254;
    }
    while (e < 254 && !((double)mn + 255.0 * (double)__int_as_float(e << 23) >= (double)mx)) e++;
    const double step = (double)__int_as_float(e << 23), base = (double)mn;
    step_exp = (uint8_t)e;
    for (int c = 0; c < 4; c++) { lo[c] = 255; hi[c] = 0; }  // no child: an empty box
    for (int c = 0; c < n; c++) {
        int ql = (int)floor(((double)kid[c].mn[a] - base) / step);
        ql = ql < 0 ? 0 : ql > 255 ? 255 : ql;
        while (ql > 0 && base + ql * step > (double)kid[c].mn[a]) ql--;                       // plane 0 is the corner itself
        int qh = (int)ceil(((double)kid[c].mx[a] - base) / step);
        qh = qh < 0 ? 0 : qh > 255 ? 255 : qh;
        while (qh < 255 && base + qh * step < (double)kid[c].mx[a]) qh++;                     // plane 255 reaches the common high bound
        lo[c] = (uint8_t)ql; hi[c] = (uint8_t)qh;
    }
}

// The wide nodes are the binary nodes of even depth; they are stored without gaps, in the binary nodes' order
// (slot[i] = how many even-depth nodes precede node i), so that a 128-byte line holds two nodes a walk may
// want and not one of them and a hole.  A flag per node here, then exclusive_scan_in_place.
__global__ void k_wide_flags(const int* depth, int n, unsigned* flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (depth[i] & 1) ? 0u : 1u;
}

__global__ void k_emit_wide(const int* left, const int* right, const int* depth, const int* slot, const uint64_t* keys, const Box* leaf_boxes,
                            const Box* node_boxes, const PtObject* world, int n_internal, ptd::LbvhNode4* nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal) return;
    if (depth[i] & 1) return;                                 // folded into its parent
    ptd::LbvhNode4 nd;
    for (int a = 0; a < 3; a++) { nd.corner[a] = 0.0f; nd.step_exp[a] = 127; for (int c = 0; c < 4; c++) { nd.lo[a][c] = 255; nd.hi[a][c] = 0; } }
    nd.step_exp[3] = 0; nd.pad[0] = 0; nd.pad[1] = 0;
    for (int c = 0; c < 4; c++) nd.link[c] = ptd::LBVH4_EMPTY;
    int kids[4], n = 0;
    const int pair[2] = { left[i], right[i] };
    for (int s = 0; s < 2; s++) {
        const int c = pair[s];
        if (c >= 0) { kids[n++] = left[c]; kids[n++] = right[c]; }
        else kids[n++] = c;
    }
    Box kid[4];
    for (int c = 0; c < n; c++) {
        const int k = kids[c];
        kid[c] = (k >= 0) ? node_boxes[k] : leaf_boxes[(unsigned)keys[~k]];
        nd.link[c] = (k >= 0) ? slot[k] : ~world[(unsigned)keys[~k]].index;
    }
    for (int a = 0; a < 3; a++) wide_axis(kid, n, a, nd.corner[a], nd.step_exp[a], nd.lo[a], nd.hi[a]);
    nodes[slot[i]] = nd;
}

// ---- surface-area-heuristic tree for small worlds (host) -----------------------------------------------
// The Morton tree splits space in the middle whatever lies in it; for a few thousand leaves a full
// sweep over the three axes is affordable on the host and gives a tree with fewer box tests per ray
// (random_scene: 14 node visits per search instead of 16).  Top-down: at every node the leaves are
// ordered by box centre along each axis in turn and cut where area(left) * count(left) +
// area(right) * count(right) is smallest.  Internal nodes are numbered in pre-order, so node 0 is the
// root, as the walk expects; leaves are referred to as ~(position in the caller's leaf list).  Only
// the tree's shape comes from here: boxes are fitted, and nodes emitted, by the kernels above.
constexpr int SAH_MAX_LEAVES = 32768;
constexpr int SAH_MAX_DEPTH = 56;                      // the walks' stacks hold 64 entries

struct HostTree { std::vector<int> left, right, parent; int depth = 0; };

double half_area(const Box& b)
{
    const double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
    const double a = dx * dy + dy * dz + dz * dx;
    return std::isfinite(a) ? a : 1e300;
}
Box join(const Box& a, const Box& b)
{
    Box u;
    for (int k = 0; k < 3; k++) { u.mn[k] = std::fmin(a.mn[k], b.mn[k]); u.mx[k] = std::fmax(a.mx[k], b.mx[k]); }
    return u;
}

// Builds the subtree over ids[begin, end) (reordered in place) and returns its link.
int sah_subtree(const std::vector<Box>& boxes, std::vector<int>& ids, int begin, int end, int parent, int level, HostTree& t,
                std::vector<int>& order, std::vector<double>& from_right)
{
    const int n = end - begin;
    if (n == 1) return ~ids[(size_t)begin];
    const int node = (int)t.left.size();
    t.left.push_back(0); t.right.push_back(0); t.parent.push_back(parent);
    t.depth = std::max(t.depth, level + 1);
    int best_axis = -1, best_cut = n / 2;
    double best_cost = 0;
    if (n > 2 && level < SAH_MAX_DEPTH - 16) {                // deeper than that (degenerate input): halve, 15 more levels at most
        for (int axis = 0; axis < 3; axis++) {
            order.assign(ids.begin() + begin, ids.begin() + end);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                return (double)boxes[(size_t)a].mn[axis] + boxes[(size_t)a].mx[axis] < (double)boxes[(size_t)b].mn[axis] + boxes[(size_t)b].mx[axis];
            });
            Box acc = boxes[(size_t)order[(size_t)n - 1]];
            from_right[(size_t)n - 1] = half_area(acc);
            for (int i = n - 2; i >= 1; i--) { acc = join(acc, boxes[(size_t)order[(size_t)i]]); from_right[(size_t)i] = half_area(acc); }
            acc = boxes[(size_t)order[0]];
            for (int cut = 1; cut < n; cut++) {                // `cut` leaves go left
                const double cost = half_area(acc) * cut + from_right[(size_t)cut] * (n - cut);
                const bool nearer_the_middle = std::abs(2 * cut - n) < std::abs(2 * best_cut - n);
                if (best_axis < 0 || cost < best_cost || (cost == best_cost && nearer_the_middle)) { best_cost = cost; best_axis = axis; best_cut = cut; }
                acc = join(acc, boxes[(size_t)order[(size_t)cut]]);
            }
        }
    } else {
        best_axis = level % 3; best_cut = n / 2;
    }
    std::stable_sort(ids.begin() + begin, ids.begin() + end, [&](int a, int b) {
        return (double)boxes[(size_t)a].mn[best_axis] + boxes[(size_t)a].mx[best_axis] < (double)boxes[(size_t)b].mn[best_axis] + boxes[(size_t)b].mx[best_axis];
    });
    const int l = sah_subtree(boxes, ids, begin, begin + best_cut, node, level + 1, t, order, from_right);
    const int r = sah_subtree(boxes, ids, begin + best_cut, end, node, level + 1, t, order, from_right);
    t.left[(size_t)node] = l; t.right[(size_t)node] = r;
    return node;
}

bool sah_build(const std::vector<Box>& boxes, HostTree& t)
{
    const int n = (int)boxes.size();
    for (const Box& b : boxes)
        for (int k = 0; k < 3; k++) if (!(b.mn[k] <= b.mx[k])) return false;      // NaN or inverted: leave it to the Morton tree
    std::vector<int> ids((size_t)n), order;
    std::iota(ids.begin(), ids.end(), 0);
    std::vector<double> from_right((size_t)n);
    t.left.reserve((size_t)n); t.right.reserve((size_t)n); t.parent.reserve((size_t)n);
    sah_subtree(boxes, ids, 0, n, -1, 0, t, order, from_right);
    return (int)t.left.size() == n - 1 && t.depth <= SAH_MAX_DEPTH;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4) == hipSuccess; }
    template <class T> T* as() { return static_cast<T*>(p); }
};

int hip_fail(hipError_t e, const char* what)
{
    std::string msg = std::string("lbvh_build: ") + what + ": " + hipGetErrorName(e);
    pt::set_error(msg.c_str());
    (void)hipGetLastError();      // the runtime keeps the last error until it is read: do not let it surface in a later call
    return PT_ERR_HIP;
}

}  // namespace

namespace pt {

int lbvh_build(const PtSceneDesc* desc, const PtShape* d_shapes, const PtXform* d_xforms,
               void** d_nodes_out, int* n_nodes_out, int* root_out, void** d_wide_out, int* depth_out, bool area_sweep)
{
    const int n = desc->n_world;
    *d_nodes_out = nullptr; *n_nodes_out = 0; *root_out = 0;
    for (int i = 0; i < n; i++) {
        if (desc->world[i].type != PT_OBJ_SHAPE) { set_error("lbvh_build: world list holds a non-shape entry"); return PT_ERR_UNSUPPORTED; }
        if (i > 0 && desc->world[i].index <= desc->world[i - 1].index) {
            set_error("lbvh_build: world shape indices must be strictly increasing (list order decides ties)");
            return PT_ERR_UNSUPPORTED;
        }
    }
    return lbvh_build_leaves(desc->world, n, desc->camera[21], desc->camera[22], d_shapes, d_xforms, d_nodes_out, n_nodes_out, root_out,
                             d_wide_out, depth_out, area_sweep);
}

int lbvh_build_leaves(const PtObject* leaves, int n, float time0, float time1, const PtShape* d_shapes, const PtXform* d_xforms,
                      void** d_nodes_out, int* n_nodes_out, int* root_out, void** d_wide_out, int* depth_out, bool area_sweep)
{
    *d_nodes_out = nullptr; *n_nodes_out = 0; *root_out = 0;
    if (d_wide_out) *d_wide_out = nullptr;
    if (depth_out) *depth_out = 0;
    if (n < 2) { set_error("lbvh_build: fewer than two shapes"); return PT_ERR_INVALID; }
    for (int i = 0; i < n; i++)
        if (leaves[i].type != PT_OBJ_SHAPE) { set_error("lbvh_build: a leaf that is not a shape"); return PT_ERR_UNSUPPORTED; }

    const int n_int = n - 1;
    const int n_tiles = (n + TILE - 1) / TILE;
    DevBuf world, boxes, cb, keys_a, keys_b, hist, left, right, parent, nboxes, ready, changed, scan_sums;
    if (!world.alloc((size_t)n * sizeof(PtObject)) || !boxes.alloc((size_t)n * sizeof(Box)) || !cb.alloc(6 * sizeof(unsigned)) ||
        !keys_a.alloc((size_t)n * 8) || !keys_b.alloc((size_t)n * 8) || !hist.alloc((size_t)256 * n_tiles * 4) ||
        !left.alloc((size_t)n_int * 4) || !right.alloc((size_t)n_int * 4) || !parent.alloc((size_t)n_int * 4) ||
        !nboxes.alloc((size_t)n_int * sizeof(Box)) || !ready.alloc((size_t)n_int * 4) || !changed.alloc(4) ||
        !scan_sums.alloc(((size_t)std::max(256 * n_tiles, n_int) / SCAN_TILE + 1) * 4)) {
        set_error("lbvh_build: out of device memory");
        return PT_ERR_OOM;
    }
    hipError_t e;
    if ((e = hipMemcpy(world.p, leaves, (size_t)n * sizeof(PtObject), hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "copy world");
    const unsigned init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    if ((e = hipMemcpy(cb.p, init, sizeof init, hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "init bounds");

    const int B = 256, G = (n + B - 1) / B, GI = (n_int + B - 1) / B;
    hipLaunchKernelGGL(k_leaf_boxes, dim3(G), dim3(B), 0, nullptr, d_shapes, d_xforms, world.as<PtObject>(), n,
                       time0, time1, boxes.as<Box>(), cb.as<unsigned>());
    uint64_t* src = keys_a.as<uint64_t>();
    bool by_area = false;
#ifndef PT_LBVH_MORTON_ONLY
    if (area_sweep && n <= SAH_MAX_LEAVES) {
        // small world: the tree's shape from the host's surface-area sweep over the boxes just computed
        std::vector<Box> h_boxes((size_t)n);
        if ((e = hipMemcpy(h_boxes.data(), boxes.p, (size_t)n * sizeof(Box), hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "read leaf boxes");
        HostTree t;
        if (sah_build(h_boxes, t)) {
            std::vector<uint64_t> identity((size_t)n);                 // leaf j of the tree is leaf j of the caller's list
            for (int j = 0; j < n; j++) identity[(size_t)j] = (uint64_t)j;
            if ((e = hipMemcpy(src, identity.data(), (size_t)n * 8, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(left.p, t.left.data(), (size_t)n_int * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(right.p, t.right.data(), (size_t)n_int * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(parent.p, t.parent.data(), (size_t)n_int * 4, hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "copy tree");
            by_area = true;
        }
    }
#endif
    if (!by_area) {
        hipLaunchKernelGGL(k_morton, dim3(G), dim3(B), 0, nullptr, boxes.as<Box>(), n, cb.as<unsigned>(), keys_a.as<uint64_t>());
        // the low 32 key bits (leaf number) are already ascending: four stable passes over the Morton bits
        uint64_t* dst = keys_b.as<uint64_t>();
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 32 + 8 * pass;
            hipLaunchKernelGGL(k_radix_hist, dim3(n_tiles), dim3(64), 0, nullptr, src, n, shift, hist.as<unsigned>(), n_tiles);
            exclusive_scan_in_place(hist.as<unsigned>(), 256 * n_tiles, scan_sums.as<unsigned>(), nullptr);
            hipLaunchKernelGGL(k_radix_scatter, dim3(n_tiles), dim3(64), 0, nullptr, src, dst, n, shift, hist.as<unsigned>(), n_tiles);
            uint64_t* tmp = src; src = dst; dst = tmp;
        }
        hipLaunchKernelGGL(k_karras, dim3(GI), dim3(B), 0, nullptr, src, n, left.as<int>(), right.as<int>(), parent.as<int>());
    }
    if ((e = hipMemset(ready.p, 0, (size_t)n_int * 4)) != hipSuccess) return hip_fail(e, "memset ready");
    bool fitted = false;
    for (int sweep = 0; sweep < 80 && !fitted; sweep++) {
        if ((e = hipMemset(changed.p, 0, 4)) != hipSuccess) return hip_fail(e, "memset changed");
        hipLaunchKernelGGL(k_fit_sweep, dim3(GI), dim3(B), 0, nullptr, left.as<int>(), right.as<int>(), src, boxes.as<Box>(),
                           nboxes.as<Box>(), ready.as<int>(), n_int, changed.as<int>());
        hipLaunchKernelGGL(k_fit_commit, dim3(GI), dim3(B), 0, nullptr, ready.as<int>(), n_int);
        int h_changed = 0;
        if ((e = hipMemcpy(&h_changed, changed.p, 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "read sweep count");
        if (h_changed == 0) fitted = true;
    }
    int root_ready = 0;
    if ((e = hipMemcpy(&root_ready, ready.p, 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "read root flag");
    if (!root_ready) { set_error("lbvh_build: tree deeper than the fitting sweeps allow"); return PT_ERR_UNSUPPORTED; }

    void* nodes = nullptr;
    if ((e = hipMalloc(&nodes, (size_t)n_int * sizeof(LbvhNode))) != hipSuccess) { set_error("lbvh_build: out of device memory"); return PT_ERR_OOM; }
    hipLaunchKernelGGL(k_emit, dim3(GI), dim3(B), 0, nullptr, left.as<int>(), right.as<int>(), src, boxes.as<Box>(), nboxes.as<Box>(),
                       world.as<PtObject>(), n_int, static_cast<LbvhNode*>(nodes));
    // depth of the binary tree (the walks' stacks are sized against it) and, on request, the 4-wide form
    DevBuf depth, max_depth;
    if (!depth.alloc((size_t)n_int * 4) || !max_depth.alloc(4)) { (void)hipFree(nodes); set_error("lbvh_build: out of device memory"); return PT_ERR_OOM; }
    if ((e = hipMemset(max_depth.p, 0, 4)) != hipSuccess) { (void)hipFree(nodes); return hip_fail(e, "memset depth"); }
    hipLaunchKernelGGL(k_depth, dim3(GI), dim3(B), 0, nullptr, parent.as<int>(), n_int, depth.as<int>(), max_depth.as<int>());
    int h_depth = 0;
    if ((e = hipMemcpy(&h_depth, max_depth.p, 4, hipMemcpyDeviceToHost)) != hipSuccess) { (void)hipFree(nodes); return hip_fail(e, "read depth"); }
    if (depth_out) *depth_out = h_depth + 1;                  // levels of internal nodes
    DevBuf slot, n_used;
    if (d_wide_out) {
        if (!slot.alloc((size_t)n_int * 4) || !n_used.alloc(4)) { (void)hipFree(nodes); set_error("lbvh_build: out of device memory"); return PT_ERR_OOM; }
        hipLaunchKernelGGL(k_wide_flags, dim3(GI), dim3(B), 0, nullptr, depth.as<int>(), n_int, slot.as<unsigned>());
        exclusive_scan_in_place(slot.as<unsigned>(), n_int, scan_sums.as<unsigned>(), n_used.as<unsigned>());
        int h_used = 0;
        if ((e = hipMemcpy(&h_used, n_used.p, 4, hipMemcpyDeviceToHost)) != hipSuccess) { (void)hipFree(nodes); return hip_fail(e, "read wide node count"); }
        void* wide = nullptr;
        if (h_used < 1 || h_used > n_int || (e = hipMalloc(&wide, (size_t)h_used * sizeof(ptd::LbvhNode4))) != hipSuccess) {
            (void)hipFree(nodes); set_error("lbvh_build: out of device memory"); return PT_ERR_OOM;
        }
        hipLaunchKernelGGL(k_emit_wide, dim3(GI), dim3(B), 0, nullptr, left.as<int>(), right.as<int>(), depth.as<int>(), slot.as<int>(), src,
                           boxes.as<Box>(), nboxes.as<Box>(), world.as<PtObject>(), n_int, static_cast<ptd::LbvhNode4*>(wide));
        *d_wide_out = wide;
    }
    if ((e = hipDeviceSynchronize()) != hipSuccess) {
        (void)hipFree(nodes);
        if (d_wide_out && *d_wide_out) { (void)hipFree(*d_wide_out); *d_wide_out = nullptr; }
        return hip_fail(e, "kernels");
    }
    *d_nodes_out = nodes;
    *n_nodes_out = n_int;
    *root_out = 0;             // Karras' root is internal node 0
    return PT_OK;
}

}  // namespace pt
