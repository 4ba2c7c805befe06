// Squared distance from every point to its nearest other point.
//
// Replaces distCUDA2 of the reference (scene/gaussian_model.py:729-746): torch.cdist of the cloud with
// itself, diagonal set to inf, row minimum, squared -- O(N^2) time AND memory (an N x N matrix: out of
// memory at 100 k points).  Used once, to initialise the scales from an SfM cloud (:212-217).
//
// Here: uniform grid + the library's own radix sort.
//   bbox       block min/max -> 6 ordered-uint atomics                          (read back: 24 bytes, the one sync)
//   cell keys  cell = floor((p - bbox_min) / cell_size), key = linear cell index; cell size chosen on the host so
//              that the grid has about N / 2 cells (each axis 1..1024 cells)
//   sort       gs_sort_pairs on the key bits; gs_identify_tile_ranges gives every cell its [begin, end)
//   gather     points into cell order (float4: xyz + original index)
//   query      lane per point: the 27 cells around it, then shells 2..KNN_MAX_RING while the best distance found
//              exceeds the distance to the faces of the searched block of cells (exact termination test; faces
//              on the grid boundary impose nothing).  A point still undecided after the last shell is "isolated":
//   brute      one workgroup per isolated point scans all N points.
//   dense      a uniform grid is quadratic inside cells that hold thousands of points (SfM clouds are that uneven): a
//              point whose 27 cells hold more than KNN_DENSE_LIMIT points is set aside, and if there are any, all points
//              are sorted by 63-bit Morton code (cubic cells of extent / 2^21) and every such point is answered
//              adaptively: upper bound from its 16 neighbours in Morton order, then the 27 cells of the coarsest level
//              whose cell edge still covers that bound (each cell = one contiguous code range, found by binary search).
// Exact nearest neighbour; distances are computed from coordinate differences in fp32 (the reference's
// cdist goes through |a|^2 + |b|^2 - 2 a.b for N > 25 and carries that form's cancellation error).
#include "gsplat_common.h"
#include "../../include/gsplat.h"

#include <math.h>
#include <string.h>

namespace {

constexpr int KNN_THREADS = 256;
constexpr int KNN_MAX_RING = 3;
constexpr int KNN_MAX_DIM = 1024;
constexpr uint32_t KNN_DENSE_LIMIT = 4096;   // points in the 27 cells around a point beyond which it takes the Morton path
constexpr int KNN_MORTON_BITS = 21;          // per axis
constexpr uint32_t KNN_SCAN_LIMIT = 16384;   // ... or that would scan more than this many points over its shells

__device__ __forceinline__ uint32_t f2ord(float f) {      // order-preserving float -> uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float ord2f(uint32_t o) {                          // host side of the same mapping
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

__global__ __launch_bounds__(KNN_THREADS) void knn_bbox_kernel(long long N, const float* __restrict__ pts,
                                                                uint32_t* __restrict__ box) {
    __shared__ uint32_t s[6];
    if (threadIdx.x < 6) s[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (long long i = (long long)blockIdx.x * KNN_THREADS + threadIdx.x; i < N; i += (long long)gridDim.x * KNN_THREADS)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t o = f2ord(pts[3 * i + a]);
            lo[a] = o < lo[a] ? o : lo[a];
            hi[a] = o > hi[a] ? o : hi[a];
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        atomicMin(&s[a], lo[a]);
        atomicMax(&s[3 + a], hi[a]);
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&box[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&box[threadIdx.x], s[threadIdx.x]);
}

struct KnnGrid {
    float bmin[3];
    float cell, inv_cell;
    int dim[3];
};

__device__ __forceinline__ void cell_of(const KnnGrid& g, float x, float y, float z, int c[3]) {
    const float p[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float t = fminf(fmaxf((p[a] - g.bmin[a]) * g.inv_cell, 0.0f), (float)(g.dim[a] - 1));
        c[a] = (int)t;
    }
}

__global__ __launch_bounds__(KNN_THREADS) void knn_keys_kernel(long long N, KnnGrid g, const float* __restrict__ pts,
                                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    const long long i = (long long)blockIdx.x * KNN_THREADS + threadIdx.x;
    if (i >= N) return;
    int c[3];
    cell_of(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], c);
    keys[i] = (uint32_t)((c[2] * g.dim[1] + c[1]) * g.dim[0] + c[0]);
    ids[i] = (uint32_t)i;
}

__global__ __launch_bounds__(KNN_THREADS) void knn_gather_kernel(long long N, const float* __restrict__ pts,
                                                                  const uint32_t* __restrict__ order,
                                                                  float4* __restrict__ sorted) {
    const long long r = (long long)blockIdx.x * KNN_THREADS + threadIdx.x;
    if (r >= N) return;
    const uint32_t i = order[r];
    sorted[r] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __uint_as_float(i));
}

__device__ __forceinline__ float scan_cell(const float4* __restrict__ sorted, const uint2 range, long long self,
                                           float x, float y, float z, float best) {
    for (uint32_t j = range.x; j < range.y; ++j) {
        const float4 q = sorted[j];
        const float dx = q.x - x, dy = q.y - y, dz = q.z - z;
        const float d = dx * dx + dy * dy + dz * dz;
        best = ((long long)j != self && d < best) ? d : best;
    }
    return best;
}

__global__ __launch_bounds__(KNN_THREADS) void knn_query_kernel(long long N, KnnGrid g, const float4* __restrict__ sorted,
                                                                 const uint2* __restrict__ ranges, float* __restrict__ out,
                                                                 uint32_t* __restrict__ hard_count,
                                                                 uint32_t* __restrict__ hard_list,
                                                                 uint32_t* __restrict__ dense_count,
                                                                 uint32_t* __restrict__ dense_list) {
    const long long r = (long long)blockIdx.x * KNN_THREADS + threadIdx.x;
    if (r >= N) return;
    const float4 p = sorted[r];
    int c[3];
    cell_of(g, p.x, p.y, p.z, c);
    const float pos[3] = {p.x, p.y, p.z};
    {   // crowded neighbourhood: leave it to the adaptive (Morton) pass instead of scanning thousands of points here
        uint32_t crowd = 0;
        for (int cz = max(c[2] - 1, 0); cz <= min(c[2] + 1, g.dim[2] - 1); ++cz)
            for (int cy = max(c[1] - 1, 0); cy <= min(c[1] + 1, g.dim[1] - 1); ++cy)
                for (int cx = max(c[0] - 1, 0); cx <= min(c[0] + 1, g.dim[0] - 1); ++cx) {
                    const uint2 rg = ranges[(cz * g.dim[1] + cy) * g.dim[0] + cx];
                    crowd += rg.y - rg.x;
                }
        // one atomic per wave, not per point: in a cluster every lane of the wave lands here
        const bool crowded = crowd > KNN_DENSE_LIMIT;
        const uint64_t cm = __ballot(crowded);
        if (cm) {
            const int lane = threadIdx.x & 63, leader = __ffsll((long long)cm) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(dense_count, (uint32_t)__popcll(cm));
            base = __shfl(base, leader);
            if (crowded) {
                dense_list[base + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull))] = (uint32_t)r;
                return;
            }
        }
    }
    float best = INFINITY;
    bool done = false;
    uint32_t scanned = 0;
    for (int ring = 1; ring <= KNN_MAX_RING && !done; ++ring) {
        const int lo[3] = {max(c[0] - ring, 0), max(c[1] - ring, 0), max(c[2] - ring, 0)};
        const int hi[3] = {min(c[0] + ring, g.dim[0] - 1), min(c[1] + ring, g.dim[1] - 1), min(c[2] + ring, g.dim[2] - 1)};
        for (int cz = lo[2]; cz <= hi[2]; ++cz)
            for (int cy = lo[1]; cy <= hi[1]; ++cy) {
                const bool inner_row = ring > 1 && abs(cz - c[2]) < ring && abs(cy - c[1]) < ring;
                for (int cx = lo[0]; cx <= hi[0]; ++cx) {
                    // shells beyond the first visit only the cells the smaller block did not contain
                    if (inner_row && abs(cx - c[0]) < ring) {
                        cx = min(c[0] + ring - 1, hi[0]);   // jump to the far face of the inner block
                        continue;
                    }
                    const uint2 rg = ranges[(cz * g.dim[1] + cy) * g.dim[0] + cx];
                    scanned += rg.y - rg.x;
                    if (scanned > KNN_SCAN_LIMIT) {      // a crowded cell further out: same treatment
                        dense_list[atomicAdd(dense_count, 1u)] = (uint32_t)r;
                        return;
                    }
                    best = scan_cell(sorted, rg, r, p.x, p.y, p.z, best);
                }
            }
        // every point outside the searched block is farther than the nearest face that is not a grid boundary
        float bound = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (c[a] - ring > 0) bound = fminf(bound, pos[a] - (g.bmin[a] + (float)(c[a] - ring) * g.cell));
            if (c[a] + ring < g.dim[a] - 1) bound = fminf(bound, (g.bmin[a] + (float)(c[a] + ring + 1) * g.cell) - pos[a]);
        }
        bound = fmaxf(bound - 1e-3f * g.cell, 0.0f);     // slack for the rounding of the cell assignment (<= 3e-4 cell)
        done = !(bound < INFINITY) || best <= bound * bound;
    }
    const uint32_t id = __float_as_uint(p.w);
    if (done) {
        out[id] = best;
    } else {
        out[id] = best;                                   // upper bound; the brute-force pass settles it
        hard_list[atomicAdd(hard_count, 1u)] = (uint32_t)r;
    }
}

__global__ __launch_bounds__(KNN_THREADS) void knn_brute_kernel(long long N, const float4* __restrict__ sorted,
                                                                 const uint32_t* __restrict__ hard_count,
                                                                 const uint32_t* __restrict__ hard_list,
                                                                 float* __restrict__ out) {
    __shared__ float red[KNN_THREADS / 64];
    const uint32_t nh = *hard_count;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {       // workgroup-uniform loop
        const long long r = hard_list[h];
        const float4 p = sorted[r];
        float best = INFINITY;
        for (long long j = threadIdx.x; j < N; j += KNN_THREADS) {
            const float4 q = sorted[j];
            const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
            const float d = dx * dx + dy * dy + dz * dz;
            best = (j != r && d < best) ? d : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = red[0];
            for (int w = 1; w < KNN_THREADS / 64; ++w) b = fminf(b, red[w]);
            out[__float_as_uint(p.w)] = b;
        }
        __syncthreads();
    }
}

// ---- adaptive pass over Morton-sorted points (63-bit codes: 21 bits per axis, cubic cells of extent / 2^21 -- fine
//      enough for float32 coordinates, so that even the tightest cluster spreads over many finest-level cells)
__device__ __forceinline__ uint64_t spread3(uint64_t v) {       // 21 bits -> every third bit
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__device__ __forceinline__ uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}

struct KnnMorton {
    float bmin[3];
    float s0, inv_s0;      // finest cell edge = largest extent / 2^21
};

__device__ __forceinline__ void quantize(const KnnMorton& M, float x, float y, float z, uint32_t q[3]) {
    const float p[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a)
        q[a] = (uint32_t)fminf(fmaxf((p[a] - M.bmin[a]) * M.inv_s0, 0.0f), (float)((1 << KNN_MORTON_BITS) - 1));
}

__device__ __forceinline__ uint64_t code_of(const KnnMorton& M, const float* __restrict__ pts, uint32_t i) {
    uint32_t q[3];
    quantize(M, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], q);
    return morton3(q[0], q[1], q[2]);
}

// the 63-bit code is sorted as two 32-bit words with the library's (key32, value32) sort: low word first, then stably
// on the high word.  phase 0: key = low word of point i, value = i.  phase 1: key = high word of point order[i].
__global__ __launch_bounds__(KNN_THREADS) void knn_morton_keys_kernel(long long N, KnnMorton M, int phase,
                                                                       const float* __restrict__ pts,
                                                                       const uint32_t* order,   // may alias ids (same index)
                                                                       uint32_t* __restrict__ keys, uint32_t* ids) {
    const long long i = (long long)blockIdx.x * KNN_THREADS + threadIdx.x;
    if (i >= N) return;
    const uint32_t id = phase == 0 ? (uint32_t)i : order[i];
    const uint64_t c = code_of(M, pts, id);
    keys[i] = phase == 0 ? (uint32_t)c : (uint32_t)(c >> 32);
    ids[i] = id;
}

// points and codes into Morton order + the inverse permutation (rank of every original index)
__global__ __launch_bounds__(KNN_THREADS) void knn_morton_gather_kernel(long long N, KnnMorton M, const float* __restrict__ pts,
                                                                         const uint32_t* __restrict__ order,
                                                                         float4* __restrict__ sorted_m,
                                                                         uint64_t* __restrict__ codes,
                                                                         uint32_t* __restrict__ rank) {
    const long long r = (long long)blockIdx.x * KNN_THREADS + threadIdx.x;
    if (r >= N) return;
    const uint32_t i = order[r];
    sorted_m[r] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __uint_as_float(i));
    codes[r] = code_of(M, pts, i);
    rank[i] = (uint32_t)r;
}

__device__ __forceinline__ uint32_t compact3(uint64_t v) {      // inverse of spread3
    v &= 0x1249249249249249ull;
    v = (v | (v >> 2)) & 0x10C30C30C30C30C3ull;
    v = (v | (v >> 4)) & 0x100F00F00F00F00Full;
    v = (v | (v >> 8)) & 0x1F0000FF0000FFull;
    v = (v | (v >> 16)) & 0x1F00000000FFFFull;
    v = (v | (v >> 32)) & 0x1FFFFFull;
    return (uint32_t)v;
}

__device__ __forceinline__ uint32_t lower_bound_in(const uint64_t* __restrict__ a, uint32_t lo, uint32_t hi, uint64_t key) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int KNN_STACK = 48;       // cells waiting to be opened (16 B each, per lane)
constexpr uint32_t KNN_LEAF = 32;   // cells with at most this many points are scanned, larger ones split into 8

// squared distance from p to the level-l cell with Morton prefix m (slack for the rounding of the quantisation; the last
// cell of an axis also holds the points clamped into it, so its far face imposes nothing)
__device__ __forceinline__ float cell_gap2(const KnnMorton& M, const float pos[3], uint64_t m, int l) {
    const uint32_t cc[3] = {compact3(m), compact3(m >> 1), compact3(m >> 2)};
    const float edge = M.s0 * (float)(1u << l);
    const uint32_t last = (1u << (KNN_MORTON_BITS - l)) - 1u;
    float gap2 = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = M.bmin[a] + (float)cc[a] * edge, hi = lo + edge;
        float gdist = 0.0f;
        if (pos[a] < lo) gdist = lo - pos[a];
        else if (pos[a] > hi && cc[a] != last) gdist = pos[a] - hi;
        gdist = fmaxf(gdist - 1e-3f * edge - 4.0f * M.s0, 0.0f);
        gap2 += gdist * gdist;
    }
    return gap2;
}

// A crowded point: upper bound from its neighbours in Morton order, then a depth-first walk over the implicit octree
// (a cell = one contiguous range of the sorted codes) from the 27 cells whose edge covers that bound, opening only
// cells closer than the best distance so far -- a cluster of 10^5 points next to the query costs a few dozen cells.
__global__ __launch_bounds__(KNN_THREADS) void knn_morton_query_kernel(long long N, KnnMorton M,
                                                                        const float4* __restrict__ sorted_grid,
                                                                        const uint32_t* __restrict__ dense_count,
                                                                        const uint32_t* __restrict__ dense_list,
                                                                        const uint64_t* __restrict__ codes,
                                                                        const float4* __restrict__ sorted_m,
                                                                        const uint32_t* __restrict__ rank,
                                                                        float* __restrict__ out) {
    const uint32_t h = blockIdx.x * KNN_THREADS + threadIdx.x;
    if (h >= *dense_count) return;
    const float4 p = sorted_grid[dense_list[h]];
    const uint32_t id = __float_as_uint(p.w);
    const long long me = rank[id];
    float best = INFINITY;
    for (long long j = max(me - 8, 0ll); j <= min(me + 8, N - 1); ++j) {
        const float4 q = sorted_m[j];
        const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
        const float d = dx * dx + dy * dy + dz * dz;
        best = (j != me && d < best) ? d : best;
    }
    if (best > 0.0f) {
        // start level: cell edge s0 * 2^l >= the bound, so the true neighbour is in the 27 cells around p's
        const float need = sqrtf(best) * 1.001f;
        int l0 = 0;
        float edge = M.s0;
        while (l0 < KNN_MORTON_BITS && edge < need) { ++l0; edge *= 2.0f; }
        uint32_t q[3];
        quantize(M, p.x, p.y, p.z, q);
        const int side = 1 << (KNN_MORTON_BITS - l0);
        const int c[3] = {(int)(q[0] >> l0), (int)(q[1] >> l0), (int)(q[2] >> l0)};
        const float pos[3] = {p.x, p.y, p.z};
        uint64_t st_m[KNN_STACK];      // pending cells: Morton prefix,
        uint2 st_r[KNN_STACK];         // range of the sorted codes,
        int st_l[KNN_STACK];           // level
        int sp = 0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int cc[3] = {c[0] + dx, c[1] + dy, c[2] + dz};
                    if (cc[0] < 0 || cc[1] < 0 || cc[2] < 0 || cc[0] >= side || cc[1] >= side || cc[2] >= side) continue;
                    const uint64_t m = morton3((uint32_t)cc[0], (uint32_t)cc[1], (uint32_t)cc[2]);
                    if (cell_gap2(M, pos, m, l0) >= best) continue;
                    const uint32_t lo = lower_bound_in(codes, 0, (uint32_t)N, m << (3 * l0));
                    const uint32_t hi = l0 == KNN_MORTON_BITS ? (uint32_t)N : lower_bound_in(codes, lo, (uint32_t)N, (m + 1) << (3 * l0));
                    // open this cell now (depth first), its pending children live on the stack
                    st_m[0] = m; st_r[0] = make_uint2(lo, hi); st_l[0] = l0; sp = 1;
                    while (sp > 0) {
                        --sp;
                        const uint64_t cm = st_m[sp];
                        const uint2 cr = st_r[sp];
                        const int cl = st_l[sp];
                        if (cr.x >= cr.y || cell_gap2(M, pos, cm, cl) >= best) continue;
                        if (cr.y - cr.x <= KNN_LEAF || cl == 0 || sp + 8 > KNN_STACK) {
                            for (uint32_t j = cr.x; j < cr.y; ++j) {
                                const float4 t = sorted_m[j];
                                const float ex = t.x - p.x, ey = t.y - p.y, ez = t.z - p.z;
                                const float d = ex * ex + ey * ey + ez * ez;
                                best = ((long long)j != me && d < best) ? d : best;
                            }
                            continue;
                        }
                        // split: the 8 children are consecutive sub-ranges of [lo, hi)
                        uint32_t b = cr.x;
#pragma unroll 1
                        for (uint64_t k = 0; k < 8; ++k) {
                            const uint64_t km = (cm << 3) | k;
                            const uint32_t e = k == 7 ? cr.y : lower_bound_in(codes, b, cr.y, (km + 1) << (3 * (cl - 1)));
                            if (e > b) { st_m[sp] = km; st_r[sp] = make_uint2(b, e); st_l[sp] = cl - 1; ++sp; }
                            b = e;
                        }
                    }
                }
    }
    out[id] = best;
}

__global__ void knn_fill_kernel(long long N, float v, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = v;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct KnnWs {
    size_t box, hard_count, dense_count, keys_a, keys_b, ids_a, ids_b, sorted, ranges, hard_list, dense_list, sorted_m, codes, sort_ws, total;
};

KnnWs knn_layout(long long N) {
    KnnWs w;
    const size_t n = (size_t)(N > 0 ? N : 1);
    size_t o = 0;
    w.box = o; o += 256;
    w.hard_count = o; o += 256;
    w.dense_count = o; o += 256;
    w.keys_a = o; o += align256(4 * n);
    w.keys_b = o; o += align256(4 * n);
    w.ids_a = o; o += align256(4 * n);
    w.ids_b = o; o += align256(4 * n);
    w.sorted = o; o += align256(16 * n);
    w.ranges = o; o += align256(8 * (2 * n + 64));        // the grid never has more than 2 N + 64 cells (reused: Morton ranks)
    w.hard_list = o; o += align256(4 * n);
    w.dense_list = o; o += align256(4 * n);
    w.sorted_m = o; o += align256(16 * n);
    w.codes = o; o += align256(8 * n);
    w.sort_ws = o; o += align256(gs_sort_workspace_bytes(N));
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t gs_nearest_dist2_workspace_bytes(long long N) { return knn_layout(N).total; }

extern "C" int gs_nearest_dist2(long long N, const float* points, float* dist2, void* workspace, size_t workspace_bytes,
                                int* stats, void* stream_) {
    GS_REQUIRE(N >= 0 && N < (1ll << 31), "N outside 0..2^31");
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = stats[4] = 0;
    if (N == 0) return GS_OK;
    GS_REQUIRE(points && dist2 && workspace, "null argument");
    GS_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    const KnnWs w = knn_layout(N);
    GS_REQUIRE(workspace_bytes >= w.total, "workspace too small: %zu < %zu", workspace_bytes, w.total);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    uint32_t* box = (uint32_t*)(ws + w.box);
    uint32_t* hard_count = (uint32_t*)(ws + w.hard_count);
    // ---- bounding box (the one host round trip: the grid is chosen from it)
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    GS_CHECK_HIP(hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, stream));
    uint32_t* dense_count = (uint32_t*)(ws + w.dense_count);
    GS_CHECK_HIP(hipMemsetAsync(hard_count, 0, 4, stream));
    GS_CHECK_HIP(hipMemsetAsync(dense_count, 0, 4, stream));
    const int bbox_blocks = (int)(gs_div_up(N, KNN_THREADS) < 1024 ? gs_div_up(N, KNN_THREADS) : 1024);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(bbox_blocks), dim3(KNN_THREADS), 0, stream, N, points, box);
    GS_CHECK_LAUNCH("knn_bbox");
    uint32_t hbox[6];
    GS_CHECK_HIP(hipMemcpyAsync(hbox, box, sizeof(hbox), hipMemcpyDeviceToHost, stream));
    GS_CHECK_HIP(hipStreamSynchronize(stream));
    KnnGrid g;
    double ext[3];
    for (int a = 0; a < 3; ++a) {
        g.bmin[a] = ord2f(hbox[a]);
        ext[a] = (double)ord2f(hbox[3 + a]) - (double)g.bmin[a];
        GS_REQUIRE(isfinite(ext[a]), "points must be finite");
    }
    // cell size: about N / 2 cells in total with every axis between 1 and KNN_MAX_DIM cells (bisection on the size)
    const double emax = fmax(ext[0], fmax(ext[1], ext[2]));
    if (emax == 0.0) {      // every point at the same place: distance 0 (a single point: +inf, as in the reference)
        hipLaunchKernelGGL(knn_fill_kernel, dim3(gs_div_up(N, 256)), dim3(256), 0, stream, N, N > 1 ? 0.0f : INFINITY, dist2);
        GS_CHECK_LAUNCH("knn_fill");
        return GS_OK;
    }
    double cell = 1.0;
    if (emax > 0.0) {
        const double target = fmax(1.0, (double)N / 2.0);
        double lo = emax / KNN_MAX_DIM, hi = emax * 2.0;
        auto cells = [&](double s) {
            double c = 1.0;
            for (int a = 0; a < 3; ++a) c *= fmin((double)KNN_MAX_DIM, fmax(1.0, ceil(ext[a] / s)));
            return c;
        };
        for (int it = 0; it < 60; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (cells(mid) > target) lo = mid; else hi = mid;
        }
        cell = hi * (1.0 + 1e-6);
    }
    long long ncells = 1;
    for (int a = 0; a < 3; ++a) {
        int d = (int)ceil(ext[a] / cell);
        d = d < 1 ? 1 : (d > KNN_MAX_DIM ? KNN_MAX_DIM : d);
        if ((double)d * cell <= ext[a]) ++d;               // the last cell must contain the maximum
        d = d > KNN_MAX_DIM + 1 ? KNN_MAX_DIM + 1 : d;
        g.dim[a] = d;
        ncells *= d;
    }
    // (the bisection keeps the product near N / 2; guard the table all the same)
    while (ncells > 2 * N + 64) {
        cell *= 1.26;
        ncells = 1;
        for (int a = 0; a < 3; ++a) {
            int d = (int)ceil(ext[a] / cell);
            d = d < 1 ? 1 : d;
            if ((double)d * cell <= ext[a]) ++d;
            g.dim[a] = d;
            ncells *= d;
        }
    }
    g.cell = (float)cell;
    g.inv_cell = (float)(1.0 / cell);
    uint32_t *keys_a = (uint32_t*)(ws + w.keys_a), *keys_b = (uint32_t*)(ws + w.keys_b);
    uint32_t *ids_a = (uint32_t*)(ws + w.ids_a), *ids_b = (uint32_t*)(ws + w.ids_b);
    const dim3 grid(gs_div_up(N, KNN_THREADS)), block(KNN_THREADS);
    hipLaunchKernelGGL(knn_keys_kernel, grid, block, 0, stream, N, g, points, keys_a, ids_a);
    GS_CHECK_LAUNCH("knn_keys");
    int bits = 1;
    while ((1ll << bits) < ncells) ++bits;
    int in_b = 0;
    int rc = gs_sort_pairs(N, nullptr, 0, bits, keys_a, ids_a, keys_b, ids_b, ws + w.sort_ws, gs_sort_workspace_bytes(N),
                           &in_b, stream);
    if (rc) return rc;
    const uint32_t* keys = in_b ? keys_b : keys_a;
    const uint32_t* order = in_b ? ids_b : ids_a;
    uint32_t* ranges = (uint32_t*)(ws + w.ranges);
    rc = gs_identify_tile_ranges(N, nullptr, (int)ncells, keys, ranges, stream);
    if (rc) return rc;
    float4* sorted = (float4*)(ws + w.sorted);
    hipLaunchKernelGGL(knn_gather_kernel, grid, block, 0, stream, N, points, order, sorted);
    GS_CHECK_LAUNCH("knn_gather");
    uint32_t* hard_list = (uint32_t*)(ws + w.hard_list);
    uint32_t* dense_list = (uint32_t*)(ws + w.dense_list);
    hipLaunchKernelGGL(knn_query_kernel, grid, block, 0, stream, N, g, sorted, (const uint2*)ranges, dist2, hard_count,
                       hard_list, dense_count, dense_list);
    GS_CHECK_LAUNCH("knn_query");
    hipLaunchKernelGGL(knn_brute_kernel, dim3(1024), dim3(KNN_THREADS), 0, stream, N, sorted, hard_count, hard_list, dist2);
    GS_CHECK_LAUNCH("knn_brute");
    // ---- crowded neighbourhoods (second host round trip: how many, if any)
    uint32_t counts[2] = {0, 0};
    GS_CHECK_HIP(hipMemcpyAsync(&counts[0], hard_count, 4, hipMemcpyDeviceToHost, stream));
    GS_CHECK_HIP(hipMemcpyAsync(&counts[1], dense_count, 4, hipMemcpyDeviceToHost, stream));
    GS_CHECK_HIP(hipStreamSynchronize(stream));
    if (counts[1] > 0) {
        KnnMorton M;
        for (int a = 0; a < 3; ++a) M.bmin[a] = g.bmin[a];
        M.s0 = (float)(emax / (double)(1 << KNN_MORTON_BITS) * (1.0 + 1e-6));
        M.inv_s0 = 1.0f / M.s0;
        hipLaunchKernelGGL(knn_morton_keys_kernel, grid, block, 0, stream, N, M, 0, points, (const uint32_t*)nullptr, keys_a,
                           ids_a);
        GS_CHECK_LAUNCH("knn_morton_keys");
        rc = gs_sort_pairs(N, nullptr, 0, 32, keys_a, ids_a, keys_b, ids_b, ws + w.sort_ws, gs_sort_workspace_bytes(N), &in_b,
                           stream);
        if (rc) return rc;
        // high words in the order of the low-word sort, back into the (a) buffers the sort reads from
        hipLaunchKernelGGL(knn_morton_keys_kernel, grid, block, 0, stream, N, M, 1, points, in_b ? ids_b : ids_a, keys_a, ids_a);
        GS_CHECK_LAUNCH("knn_morton_keys");
        rc = gs_sort_pairs(N, nullptr, 0, 3 * KNN_MORTON_BITS - 32, keys_a, ids_a, keys_b, ids_b, ws + w.sort_ws,
                           gs_sort_workspace_bytes(N), &in_b, stream);
        if (rc) return rc;
        const uint32_t* morder = in_b ? ids_b : ids_a;
        float4* sorted_m = (float4*)(ws + w.sorted_m);
        uint64_t* codes = (uint64_t*)(ws + w.codes);
        uint32_t* rank = ranges;                       // the grid's cell table is no longer needed
        hipLaunchKernelGGL(knn_morton_gather_kernel, grid, block, 0, stream, N, M, points, morder, sorted_m, codes, rank);
        GS_CHECK_LAUNCH("knn_morton_gather");
        hipLaunchKernelGGL(knn_morton_query_kernel, dim3(gs_div_up(counts[1], KNN_THREADS)), block, 0, stream, N, M, sorted,
                           dense_count, dense_list, codes, sorted_m, rank, dist2);
        GS_CHECK_LAUNCH("knn_morton_query");
    }
    if (stats) {
        stats[0] = g.dim[0]; stats[1] = g.dim[1]; stats[2] = g.dim[2]; stats[3] = (int)counts[0]; stats[4] = (int)counts[1];
    }
    return GS_OK;
}
