// rtx_build.hip — the SAH hierarchy built ON THE DEVICE (gfx950): SURVEY.md 8(f)-4, second half.
//
// What is built has no counterpart in the reference (raytracer/octree.rs:15-43 builds the octree, which stays host-built and
// bit-identical); it replaces the host builder of rtx_bvh.cpp, kernel for kernel the same algorithm, so that both give the SAME
// tree and the same two emitted arrays (tests/test_gpu_build.py compares them word for word):
//
//   1. binary tree, top-down, level-synchronous: every range of more than SWEEP_MAX prims is split by a binned SAH pass
//      (16 bins per axis over the centroid bounds, f64, the host's expressions in the host's order) and a STABLE partition
//      (flags -> exclusive scan -> scatter), one set of launches per tree level for all ranges of the level at once;
//   2. ranges of 2 .. SWEEP_MAX prims: one lane builds the whole subtree with the exact SAH sweep (insertion sort by centroid per
//      axis), and its collapse costs;
//   3. collapse DP (Ylitie et al. 2017, the cost model of rtx_bvh.cpp: expected node visits) for the binned nodes, bottom-up,
//      one launch per level;
//   4. breadth-first emission of the 16-wide box lists and of the 4-wide nodes, one wide level per round (count -> scan -> next
//      level list), then one write kernel per structure (f64 -> f32 with directed rounding = outward).
//
// min / max of f64 bounds go through order-preserving u64 keys and integer atomics (exact, order-independent); leaves are not
// nodes: a child reference < 0 is ~(position in `order`) of a single prim.  All arithmetic that decides a split or a collapse
// repeats the host's statement by statement under -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "rtx_build.h"

namespace {

#define BDEV __device__ __forceinline__
constexpr int BT = 256;          // threads per workgroup
constexpr int BINS = 16;         // rtx_bvh.cpp: BINS
constexpr int SWEEP_MAX = 12;    // rtx_bvh.cpp: Builder::SWEEP_MAX
constexpr int NB = 3 * BINS;     // bins per range
constexpr uint64_t KEY_PINF = 0xFFF0000000000000ull;   // okey(+inf)
constexpr uint64_t KEY_NINF = 0x000FFFFFFFFFFFFFull;   // okey(-inf)

struct PBox {            // = RtxPrimBox
    double mn[3], mx[3];
    uint32_t ref, pad;
};
static_assert(sizeof(PBox) == sizeof(RtxPrimBox), "PBox mirrors RtxPrimBox");
struct BSeg {            // a range of `order` that is split by a binning pass
    int begin, end, node, axis;      // axis < 0: no split plane (all centroids coincide): the range is halved
    int split, mid, child0, child1;  // child*: index of the child's BSeg within the next level, -1: not a binned range
    double c0, scale;
};
struct BNodeD {          // an internal node of the binary tree; child >= 0: node index, < 0: ~position in `order` of a prim
    double mn[3], mx[3];
    int left, right, pad0, pad1;
};
struct SmallSeg { int begin, end, node; };
struct BCtr {
    unsigned n_nodes, n_next, n_small, scan_total;
    unsigned long long kids_sum[2];
};

BDEV uint64_t okey(double x) {
    uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
BDEV double okey_inv(uint64_t k) {
    uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
BDEV double area(const double *mn, const double *mx) {   // rtx_bvh.cpp: area()
    double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
BDEV void grow(double *mn, double *mx, const double *bmn, const double *bmx) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (bmn[a] < mn[a]) mn[a] = bmn[a];
        if (bmx[a] > mx[a]) mx[a] = bmx[a];
    }
}
BDEV void reset(double *mn, double *mx) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
}
BDEV int bin_of(double pmn, double pmx, double cmn, double scale) {   // Builder::bin_of
    int b = (int)((0.5 * (pmn + pmx) - cmn) * scale);
    return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
}

// ---------------------------------------------------------------------------------------- level-synchronous binned splits
__global__ void __launch_bounds__(BT) k_init(int n, int *order, int *pos_seg, BSeg *segs, SmallSeg *small, BCtr *ctr) {
    int i = blockIdx.x * BT + threadIdx.x;
    if (i == 0) {
        BCtr c = {};
        c.n_nodes = 1u;   // the root
        if (n > SWEEP_MAX) {
            BSeg s = {};
            s.begin = 0; s.end = n; s.node = 0;
            segs[0] = s;
        } else {
            small[0] = SmallSeg{0, n, 0};
            c.n_small = 1u;
        }
        *ctr = c;
    }
    if (i >= n) return;
    order[i] = i;
    pos_seg[i] = n > SWEEP_MAX ? 0 : -1;
}

// bounds keys and bins of the level's S ranges; also clears the counter of ranges the level hands to the next one
__global__ void __launch_bounds__(BT) k_level_init(int S, uint64_t *bkey, uint64_t *binkey, unsigned *bincnt, BCtr *ctr) {
    int t = blockIdx.x * BT + threadIdx.x;
    if (t == 0) ctr->n_next = 0u;
    if (t >= S * NB) return;
    bincnt[t] = 0u;
#pragma unroll
    for (int c = 0; c < 3; ++c) { binkey[(size_t)t * 6 + c] = KEY_PINF; binkey[(size_t)t * 6 + 3 + c] = KEY_NINF; }
    int s = t / NB, b = t % NB;
    if (b < 12) bkey[(size_t)s * 12 + b] = (b % 6) < 3 ? KEY_PINF : KEY_NINF;
}

BDEV double wave_min(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { double o = __shfl_xor(v, m, 64); v = o < v ? o : v; }
    return v;
}
BDEV double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { double o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return v;
}

// Builder::bounds_of for every range of the level at once: bkey[s] = box of the prims (0..5) and of their centroids (6..11)
__global__ void __launch_bounds__(BT) k_bounds(int n, const int *order, const int *pos_seg, const PBox *pbox, uint64_t *bkey) {
    int i = blockIdx.x * BT + threadIdx.x;
    int s = i < n ? pos_seg[i] : -1;
    const bool valid = s >= 0;
    unsigned long long vb = __ballot(valid);
    if (vb == 0ull) return;
    double v[12];
#pragma unroll
    for (int a = 0; a < 3; ++a) { v[a] = INFINITY; v[3 + a] = -INFINITY; v[6 + a] = INFINITY; v[9 + a] = -INFINITY; }
    if (valid) {
        const PBox &p = pbox[order[i]];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            v[a] = p.mn[a]; v[3 + a] = p.mx[a];
            double c = 0.5 * (p.mn[a] + p.mx[a]);
            v[6 + a] = c; v[9 + a] = c;
        }
    }
    const int s0 = __shfl(s, (int)__ffsll((long long)vb) - 1, 64);
    const bool uni = __ballot(valid && s != s0) == 0ull;
    if (uni) {   // the whole wave lies in one range (the upper levels): one atomic per value and wave
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = (k % 6) < 3 ? wave_min(v[k]) : wave_max(v[k]);
        if ((threadIdx.x & 63) == 0) {
            uint64_t *bk = bkey + (size_t)s0 * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                if ((k % 6) < 3) atomicMin((unsigned long long *)&bk[k], (unsigned long long)okey(v[k]));
                else atomicMax((unsigned long long *)&bk[k], (unsigned long long)okey(v[k]));
            }
        }
    } else if (valid) {
        uint64_t *bk = bkey + (size_t)s * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            if ((k % 6) < 3) atomicMin((unsigned long long *)&bk[k], (unsigned long long)okey(v[k]));
            else atomicMax((unsigned long long *)&bk[k], (unsigned long long)okey(v[k]));
        }
    }
}

// Builder::bin_range for every range of the level: per axis with a centroid extent, the prim's bin gets its count and its box.
// A workgroup whose 256 positions lie in one range bins into LDS and flushes the non-empty bins (the upper levels: one range,
// 48 bins, every prim of the scene).
__global__ void __launch_bounds__(BT) k_bin(int n, const int *order, const int *pos_seg, const PBox *pbox, const uint64_t *bkey,
                                             uint64_t *binkey, unsigned *bincnt) {
    __shared__ unsigned long long s_key[NB * 6];
    __shared__ unsigned s_cnt[NB];
    const int i0 = blockIdx.x * BT, i = i0 + threadIdx.x;
    const int i1 = min(n - 1, i0 + BT - 1);
    const int sa = pos_seg[i0], sb = pos_seg[i1];
    const bool uni = sa >= 0 && sa == sb;   // ranges are contiguous: equal ends => the whole workgroup lies in range sa
    if (uni) {
        for (int t = threadIdx.x; t < NB * 6; t += BT) s_key[t] = ((t % 6) < 3) ? KEY_PINF : KEY_NINF;
        if (threadIdx.x < NB) s_cnt[threadIdx.x] = 0u;
        __syncthreads();
    }
    const int s = i < n ? pos_seg[i] : -1;
    if (s >= 0) {
        const PBox &p = pbox[order[i]];
        const uint64_t *bk = bkey + (size_t)s * 12;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double cmn = okey_inv(bk[6 + a]), cmx = okey_inv(bk[9 + a]);
            const double ext = cmx - cmn;
            if (!(ext > 0.0)) continue;
            const double scale = (double)BINS / ext;
            const int b = a * BINS + bin_of(p.mn[a], p.mx[a], cmn, scale);
            if (uni) {
                atomicAdd(&s_cnt[b], 1u);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    atomicMin(&s_key[b * 6 + c], (unsigned long long)okey(p.mn[c]));
                    atomicMax(&s_key[b * 6 + 3 + c], (unsigned long long)okey(p.mx[c]));
                }
            } else {
                const size_t g = (size_t)s * NB + b;
                atomicAdd(&bincnt[g], 1u);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    atomicMin((unsigned long long *)&binkey[g * 6 + c], (unsigned long long)okey(p.mn[c]));
                    atomicMax((unsigned long long *)&binkey[g * 6 + 3 + c], (unsigned long long)okey(p.mx[c]));
                }
            }
        }
    }
    if (uni) {
        __syncthreads();
        if (threadIdx.x < NB && s_cnt[threadIdx.x]) {
            const size_t g = (size_t)sa * NB + threadIdx.x;
            atomicAdd(&bincnt[g], s_cnt[threadIdx.x]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicMin((unsigned long long *)&binkey[g * 6 + c], s_key[threadIdx.x * 6 + c]);
                atomicMax((unsigned long long *)&binkey[g * 6 + 3 + c], s_key[threadIdx.x * 6 + 3 + c]);
            }
        }
    }
}

// Builder::split's evaluation of the binned SAH, one lane per range; writes the range's node and registers its two children:
// a single prim is a reference, 2 .. SWEEP_MAX prims a "small" range (k_small), more a range of the next level
__global__ void __launch_bounds__(BT) k_split(int S, BSeg *segs, BSeg *next, const uint64_t *bkey, const uint64_t *binkey,
                                               const unsigned *bincnt, BNodeD *nodes, SmallSeg *small, BCtr *ctr) {
    const int s = blockIdx.x * BT + threadIdx.x;
    if (s >= S) return;
    BSeg sg = segs[s];
    const int n = sg.end - sg.begin;
    const uint64_t *bk = bkey + (size_t)s * 12;
    BNodeD nd;
    double cmn[3], cmx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        nd.mn[a] = okey_inv(bk[a]); nd.mx[a] = okey_inv(bk[3 + a]);
        cmn[a] = okey_inv(bk[6 + a]); cmx[a] = okey_inv(bk[9 + a]);
    }
    int best_axis = -1, best_split = -1, best_left = 0;
    double best_cost = INFINITY;
    for (int a = 0; a < 3; ++a) {
        if (!(cmx[a] - cmn[a] > 0.0)) continue;
        const size_t g0 = (size_t)s * NB + (size_t)a * BINS;
        double rarea[BINS];
        int rcnt[BINS];
        double amn[3], amx[3];
        reset(amn, amx);
        int c = 0;
        for (int b = BINS - 1; b > 0; --b) {
            const int cb = (int)bincnt[g0 + b];
            if (cb) {
                double bmn[3], bmx[3];
                for (int k = 0; k < 3; ++k) { bmn[k] = okey_inv(binkey[(g0 + b) * 6 + k]); bmx[k] = okey_inv(binkey[(g0 + b) * 6 + 3 + k]); }
                grow(amn, amx, bmn, bmx);
            }
            c += cb;
            rarea[b] = c ? area(amn, amx) : 0.0;
            rcnt[b] = c;
        }
        reset(amn, amx);
        c = 0;
        for (int b = 0; b < BINS - 1; ++b) {
            const int cb = (int)bincnt[g0 + b];
            if (cb) {
                double bmn[3], bmx[3];
                for (int k = 0; k < 3; ++k) { bmn[k] = okey_inv(binkey[(g0 + b) * 6 + k]); bmx[k] = okey_inv(binkey[(g0 + b) * 6 + 3 + k]); }
                grow(amn, amx, bmn, bmx);
            }
            c += cb;
            if (c == 0 || rcnt[b + 1] == 0) continue;
            const double cost = area(amn, amx) * c + rarea[b + 1] * rcnt[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; best_left = c; }
        }
    }
    int mid;
    if (best_axis < 0) {
        mid = sg.begin + n / 2;
    } else {
        mid = sg.begin + best_left;
        if (mid == sg.begin || mid == sg.end) { best_axis = -1; mid = sg.begin + n / 2; }
    }
    sg.axis = best_axis;
    sg.split = best_split;
    sg.mid = mid;
    sg.c0 = best_axis >= 0 ? cmn[best_axis] : 0.0;
    sg.scale = best_axis >= 0 ? (double)BINS / (cmx[best_axis] - cmn[best_axis]) : 0.0;
    int child_ref[2], child_seg[2];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int b = side ? mid : sg.begin, e = side ? sg.end : mid;
        child_seg[side] = -1;
        if (e - b == 1) {
            child_ref[side] = ~b;
        } else {
            const int node = (int)atomicAdd(&ctr->n_nodes, 1u);
            child_ref[side] = node;
            if (e - b <= SWEEP_MAX) {
                small[atomicAdd(&ctr->n_small, 1u)] = SmallSeg{b, e, node};
            } else {
                const int k = (int)atomicAdd(&ctr->n_next, 1u);
                BSeg ns = {};
                ns.begin = b; ns.end = e; ns.node = node;
                next[k] = ns;
                child_seg[side] = k;
            }
        }
    }
    sg.child0 = child_seg[0];
    sg.child1 = child_seg[1];
    segs[s] = sg;
    nd.left = child_ref[0];
    nd.right = child_ref[1];
    nd.pad0 = nd.pad1 = 0;
    nodes[sg.node] = nd;
}

// is_left of Builder::split, per position (1 / 0), for the scan; positions outside the level's ranges: 0
__global__ void __launch_bounds__(BT) k_flag(int n, const int *order, const int *pos_seg, const PBox *pbox, const BSeg *segs, uint8_t *flag,
                                              unsigned *L) {
    const int i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    const int s = pos_seg[i];
    unsigned f = 0u;
    if (s >= 0) {
        const BSeg &sg = segs[s];
        if (sg.axis < 0) f = i < sg.mid ? 1u : 0u;
        else {
            const PBox &p = pbox[order[i]];
            f = bin_of(p.mn[sg.axis], p.mx[sg.axis], sg.c0, sg.scale) <= sg.split ? 1u : 0u;
        }
    }
    flag[i] = (uint8_t)f;
    L[i] = f;
}

// the stable partition (std::stable_partition / the parallel count-and-scatter of Builder::split) of every range of the level:
// L = exclusive scan of the flags over all positions, so a position's rank among its range's lefts is L[i] - L[begin]
__global__ void __launch_bounds__(BT) k_scatter(int n, const int *order, const int *pos_seg, const BSeg *segs, const uint8_t *flag,
                                                 const unsigned *L, int *order_out, int *pos_seg_out) {
    const int i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    const int s = pos_seg[i];
    if (s < 0) {
        order_out[i] = order[i];
        pos_seg_out[i] = -1;
        return;
    }
    const BSeg &sg = segs[s];
    const int lr = (int)(L[i] - L[sg.begin]);
    const bool left = flag[i] != 0;
    const int to = left ? sg.begin + lr : sg.mid + (i - sg.begin - lr);
    order_out[to] = order[i];
    pos_seg_out[to] = left ? sg.child0 : sg.child1;
}

// ---------------------------------------------------------------------------------------- exclusive scan of u32 (in place)
constexpr int SCAN_ITEMS = 4, SCAN_TILE = BT * SCAN_ITEMS;
BDEV unsigned block_exclusive(unsigned v, unsigned *s_wave, unsigned *total) {   // BT threads; returns the exclusive prefix of v
    unsigned inc = v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    unsigned base = 0u, tot = 0u;
#pragma unroll
    for (int k = 0; k < BT / 64; ++k) {
        if (k < w) base += s_wave[k];
        tot += s_wave[k];
    }
    *total = tot;
    __syncthreads();
    return base + inc - v;
}
__global__ void __launch_bounds__(BT) k_scan_tile(unsigned *data, int n, unsigned *sums, unsigned *total_out) {
    __shared__ unsigned s_wave[BT / 64];
    const int i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    unsigned v[SCAN_ITEMS], t = 0u;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = i0 + k < n ? data[i0 + k] : 0u; t += v[k]; }
    unsigned tot;
    unsigned ex = block_exclusive(t, s_wave, &tot);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (i0 + k < n) data[i0 + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) {
        sums[blockIdx.x] = tot;
        if (gridDim.x == 1 && total_out) *total_out = tot;
    }
}
__global__ void __launch_bounds__(BT) k_scan_sums(unsigned *sums, int nb, unsigned *total_out) {   // one workgroup
    __shared__ unsigned s_wave[BT / 64];
    unsigned carry = 0u;
    for (int c0 = 0; c0 < nb; c0 += BT) {
        const int i = c0 + threadIdx.x;
        unsigned v = i < nb ? sums[i] : 0u, tot;
        unsigned ex = block_exclusive(v, s_wave, &tot);
        if (i < nb) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}
__global__ void __launch_bounds__(BT) k_scan_add(unsigned *data, int n, const unsigned *sums) {
    const int i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    const unsigned add = sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (i0 + k < n) data[i0 + k] += add;
}
void scan_exclusive(hipStream_t st, unsigned *data, int n, unsigned *sums, unsigned *total_out) {
    if (n <= 0) return;
    const int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tile, dim3(nb), dim3(BT), 0, st, data, n, sums, total_out);
    if (nb > 1) {
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BT), 0, st, sums, nb, total_out);
        hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(BT), 0, st, data, n, (const unsigned *)sums);
    }
}

// ---------------------------------------------------------------------------------------- collapse DP
// Builder::collapse_node: cost[i-1] = least expected number of wide-node visits below the node when its subtree is handed i child
// slots; how[i-1] = 0: as with i - 1 slots, k > 0: the left child gets k slots; how[0] holds `own` (the host keeps it apart: the
// split of the W slots between the two children when the node is a wide node itself).  A prim child costs nothing (all zeros).
template <int W>
BDEV void dp_node(const BNodeD &nd, float *cost, uint8_t *how, int node) {
    float lc[W], rc[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        lc[k] = nd.left >= 0 ? cost[(size_t)nd.left * W + k] : 0.f;
        rc[k] = nd.right >= 0 ? cost[(size_t)nd.right * W + k] : 0.f;
    }
    float d[W + 1];
    uint8_t dk[W + 1];
#pragma unroll
    for (int j = 2; j <= W; ++j) {
        float best = INFINITY;
        int bk = 1;
#pragma unroll
        for (int k = 1; k < j; ++k) {
            float cc = lc[k - 1] + rc[j - k - 1];
            if (cc < best) { best = cc; bk = k; }
        }
        d[j] = best;
        dk[j] = (uint8_t)bk;
    }
    float c[W];
    uint8_t h[W];
    c[0] = (float)area(nd.mn, nd.mx) + d[W];
    h[0] = dk[W];   // own
#pragma unroll
    for (int i = 2; i <= W; ++i) {
        if (d[i] < c[i - 2]) { c[i - 1] = d[i]; h[i - 1] = dk[i]; }
        else { c[i - 1] = c[i - 2]; h[i - 1] = 0; }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) { cost[(size_t)node * W + k] = c[k]; how[(size_t)node * W + k] = h[k]; }
}

__global__ void __launch_bounds__(BT) k_dp(int S, const BSeg *segs, const BNodeD *nodes, float *cost16, uint8_t *how16, float *cost4,
                                            uint8_t *how4) {
    const int s = blockIdx.x * BT + threadIdx.x;
    if (s >= S) return;
    const int node = segs[s].node;
    const BNodeD nd = nodes[node];
    dp_node<RTX_WIDE>(nd, cost16, how16, node);
    dp_node<RTX_BVH4>(nd, cost4, how4, node);
}

// ---------------------------------------------------------------------------------------- small ranges: one lane per subtree
// Builder::split_sweep + build_local for a range of 2 .. SWEEP_MAX prims: exact SAH sweep over the prims sorted by centroid, per
// axis (stable insertion sort), the subtree's nodes, and their collapse costs (children before parents).
__global__ void __launch_bounds__(BT) k_small(int n_small, const SmallSeg *small, int *order, const PBox *pbox, BNodeD *nodes, BCtr *ctr,
                                               float *cost16, uint8_t *how16, float *cost4, uint8_t *how4) {
    const int t = blockIdx.x * BT + threadIdx.x;
    if (t >= n_small) return;
    const SmallSeg sm = small[t];
    const int n_all = sm.end - sm.begin;
    const int base = n_all > 2 ? (int)atomicAdd(&ctr->n_nodes, (unsigned)(n_all - 2)) : 0;
    int next = base;
    struct Task { int node, begin, end; };
    Task stack[SWEEP_MAX];
    int top = 0;
    stack[top++] = Task{sm.node, sm.begin, sm.end};
    while (top > 0) {
        const Task tk = stack[--top];
        const int n = tk.end - tk.begin;
        int idx[3][SWEEP_MAX];
        double cen[3][SWEEP_MAX];
        BNodeD nd;
        reset(nd.mn, nd.mx);
        for (int i = 0; i < n; ++i) {
            const int pi = order[tk.begin + i];
            const PBox &p = pbox[pi];
            grow(nd.mn, nd.mx, p.mn, p.mx);
            for (int a = 0; a < 3; ++a) {
                const double c = 0.5 * (p.mn[a] + p.mx[a]);
                int j = i;
                while (j > 0 && cen[a][j - 1] > c) { cen[a][j] = cen[a][j - 1]; idx[a][j] = idx[a][j - 1]; --j; }
                cen[a][j] = c;
                idx[a][j] = pi;
            }
        }
        int best_axis = -1, best_k = 0;
        double best_cost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            if (!(cen[a][n - 1] > cen[a][0])) continue;
            double rarea[SWEEP_MAX];
            double amn[3], amx[3];
            reset(amn, amx);
            for (int i = n - 1; i > 0; --i) {
                const PBox &p = pbox[idx[a][i]];
                grow(amn, amx, p.mn, p.mx);
                rarea[i] = area(amn, amx);
            }
            reset(amn, amx);
            for (int k = 1; k < n; ++k) {
                const PBox &p = pbox[idx[a][k - 1]];
                grow(amn, amx, p.mn, p.mx);
                const double cost = area(amn, amx) * k + rarea[k] * (n - k);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_k = k; }
            }
        }
        int mid;
        if (best_axis < 0) mid = tk.begin + n / 2;
        else {
            for (int i = 0; i < n; ++i) order[tk.begin + i] = idx[best_axis][i];
            mid = tk.begin + best_k;
        }
        // children (the host pushes left then right and pops right first; the numbering of the nodes does not matter, only that
        // children get larger indices than their parent)
        if (mid - tk.begin == 1) nd.left = ~tk.begin;
        else { nd.left = next++; stack[top++] = Task{nd.left, tk.begin, mid}; }
        if (tk.end - mid == 1) nd.right = ~mid;
        else { nd.right = next++; stack[top++] = Task{nd.right, mid, tk.end}; }
        nd.pad0 = nd.pad1 = 0;
        nodes[tk.node] = nd;
    }
    // collapse costs: the subtree's own nodes backwards, then its root
    for (int k = next - 1; k >= base && n_all > 2; --k) {
        const BNodeD nd = nodes[k];
        dp_node<RTX_WIDE>(nd, cost16, how16, k);
        dp_node<RTX_BVH4>(nd, cost4, how4, k);
    }
    const BNodeD nd = nodes[sm.node];
    dp_node<RTX_WIDE>(nd, cost16, how16, sm.node);
    dp_node<RTX_BVH4>(nd, cost4, how4, sm.node);
}

// ---------------------------------------------------------------------------------------- emission (emit_levels of rtx_bvh.cpp)
// One wide level: lane i lists the children of wide node level_first + i (the cut the collapse DP chose below binary node
// wnode_bin[..]) in slot order, and counts those that are nodes.
template <int W>
__global__ void __launch_bounds__(BT) k_emit_count(int level_first, int n, const int *wnode_bin, const BNodeD *nodes, const uint8_t *how,
                                                    int *kids, uint8_t *n_kids, unsigned *nn_out, BCtr *ctr, int which) {
    const int i = blockIdx.x * BT + threadIdx.x;
    int nk = 0, nn = 0;
    if (i < n) {
        const int wn = level_first + i;
        const int bnode = wnode_bin[wn];
        struct Open { int node, slots; };
        Open open[2 * W];
        int top = 0;
        {
            const int own = (int)how[(size_t)bnode * W];
            const int l = nodes[bnode].left, r = nodes[bnode].right;
            open[top++] = Open{r, W - own};
            open[top++] = Open{l, own};
        }
        int *k_out = kids + (size_t)wn * W;
        while (top > 0) {
            Open o = open[--top];
            if (o.node < 0) {   // a primitive
                k_out[nk++] = o.node;
                continue;
            }
            const uint8_t *h = how + (size_t)o.node * W;
            while (o.slots > 1 && h[o.slots - 1] == 0) --o.slots;   // no better than with one slot fewer
            if (o.slots == 1) {                                       // a wide node of its own
                k_out[nk++] = o.node;
                ++nn;
                continue;
            }
            const int kl = (int)h[o.slots - 1];
            const int l = nodes[o.node].left, r = nodes[o.node].right;
            open[top++] = Open{r, o.slots - kl};
            open[top++] = Open{l, kl};
        }
        n_kids[wn] = (uint8_t)nk;
        nn_out[wn] = (unsigned)nn;
    }
    int sum = nk;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&ctr->kids_sum[which], (unsigned long long)sum);
}
// first[] holds the exclusive scan of the level's node-child counts: the node children of wide node level_first + i become the
// wide nodes next_first + first[i] ... in slot order
template <int W>
__global__ void __launch_bounds__(BT) k_emit_next(int level_first, int n, int next_first, int *wnode_bin, const int *kids, const uint8_t *n_kids,
                                                   unsigned *first) {
    const int i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    const int wn = level_first + i;
    const unsigned f = (unsigned)next_first + first[wn];
    unsigned r = f;
    const int *k_in = kids + (size_t)wn * W;
    for (int j = 0; j < (int)n_kids[wn]; ++j)
        if (k_in[j] >= 0) wnode_bin[r++] = k_in[j];
    first[wn] = f;
}

BDEV void kid_box(int kid, const BNodeD *nodes, const int *order, const PBox *pbox, double *mn, double *mx, uint32_t *pref) {
    if (kid >= 0) {
        const BNodeD &k = nodes[kid];
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = k.mn[a]; mx[a] = k.mx[a]; }
        *pref = 0u;
    } else {
        const PBox &p = pbox[order[~kid]];
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = p.mn[a]; mx[a] = p.mx[a]; }
        *pref = p.ref;
    }
}
// DWideChild records: one lane per (wide node, slot)
__global__ void __launch_bounds__(BT) k_write16(int n_nodes, const int *kids, const uint8_t *n_kids, const unsigned *first, const BNodeD *nodes,
                                                 const int *order, const PBox *pbox, double cx, double cy, double cz, DWideChild *out) {
    const int t = blockIdx.x * BT + threadIdx.x;
    const int wn = t / RTX_WIDE, j = t % RTX_WIDE;
    if (wn >= n_nodes) return;
    const int nk = (int)n_kids[wn];
    const int *k_in = kids + (size_t)wn * RTX_WIDE;
    int below = 0, nn = 0;
    for (int q = 0; q < nk; ++q) {
        const int isn = k_in[q] >= 0 ? 1 : 0;
        nn += isn;
        if (q < j) below += isn;
    }
    DWideChild w;
    if (j < nk) {
        double mn[3], mx[3];
        uint32_t pref;
        kid_box(k_in[j], nodes, order, pbox, mn, mx, &pref);
        w.lo[0] = __double2float_rd(mn[0] - cx); w.lo[1] = __double2float_rd(mn[1] - cy); w.lo[2] = __double2float_rd(mn[2] - cz);
        w.hi[0] = __double2float_ru(mx[0] - cx); w.hi[1] = __double2float_ru(mx[1] - cy); w.hi[2] = __double2float_ru(mx[2] - cz);
        w.ref = k_in[j] >= 0 ? (RTX_WREF_NODE | (first[wn] + (unsigned)below)) : pref;
    } else {
        w.lo[0] = w.lo[1] = w.lo[2] = INFINITY;
        w.hi[0] = w.hi[1] = w.hi[2] = -INFINITY;
        w.ref = RTX_WREF_EMPTY;
    }
    w.first_node = nn > 0 ? first[wn] : 0u;
    out[t] = w;
}
// DNode4 records: one lane per node
__global__ void __launch_bounds__(BT) k_write4(int n_nodes, const int *kids, const uint8_t *n_kids, const unsigned *first, const BNodeD *nodes,
                                                const int *order, const PBox *pbox, double cx, double cy, double cz, DNode4 *out) {
    const int wn = blockIdx.x * BT + threadIdx.x;
    if (wn >= n_nodes) return;
    const int nk = (int)n_kids[wn];
    const int *k_in = kids + (size_t)wn * RTX_BVH4;
    DNode4 rec;
    unsigned nn = first[wn];
#pragma unroll
    for (int j = 0; j < RTX_BVH4; ++j) {
        if (j < nk) {
            double mn[3], mx[3];
            uint32_t pref;
            kid_box(k_in[j], nodes, order, pbox, mn, mx, &pref);
            rec.lo[0][j] = __double2float_rd(mn[0] - cx); rec.lo[1][j] = __double2float_rd(mn[1] - cy); rec.lo[2][j] = __double2float_rd(mn[2] - cz);
            rec.hi[0][j] = __double2float_ru(mx[0] - cx); rec.hi[1][j] = __double2float_ru(mx[1] - cy); rec.hi[2][j] = __double2float_ru(mx[2] - cz);
            rec.ref[j] = k_in[j] >= 0 ? (RTX_WREF_NODE | nn++) : pref;
        } else {
            rec.lo[0][j] = rec.lo[1][j] = rec.lo[2][j] = INFINITY;
            rec.hi[0][j] = rec.hi[1][j] = rec.hi[2][j] = -INFINITY;
            rec.ref[j] = RTX_WREF_EMPTY;
        }
    }
    out[wn] = rec;
}

inline int grid(int64_t n) { return (int)((n + BT - 1) / BT); }

struct Arena {
    char *base = nullptr;
    size_t used = 0, cap = 0;
    template <class T>
    T *take(size_t n) {
        used = (used + 255) & ~(size_t)255;
        T *p = (T *)(base + used);
        used += n * sizeof(T);
        return p;
    }
};

}   // namespace

#define BCHECK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            if (err) *err = std::string("device hierarchy build: ") + #expr + ": " + hipGetErrorString(e_); \
            rc = (int)e_;                                                                    \
            goto done;                                                                       \
        }                                                                                    \
    } while (0)

int rtxk_build_hierarchies_device(const RtxPrimBox *prims, int64_t n64, const double centre[3], RtxDeviceHierarchy *out, bool timing,
                                  std::string *err) {
    int rc = 0;
    const int n = (int)n64;
    hipStream_t st = nullptr;
    Arena ar;
    DWideChild *d_wide = nullptr;
    DNode4 *d_bvh4 = nullptr;
    std::vector<int> level_base, level_count;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](double *slot) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        auto now = std::chrono::steady_clock::now();
        *slot += std::chrono::duration<double, std::milli>(now - t_prev).count();
        t_prev = now;
    };
    if (n64 < 2 || n64 >= (1ll << 30)) {
        if (err) *err = "device hierarchy build: prim count out of range";
        return (int)hipErrorInvalidValue;
    }
    {
        // ---- one arena for everything transient
        const size_t N = (size_t)n;
        const size_t max_segs_level = N / (SWEEP_MAX + 1) + 2;
        size_t bytes = 0;
        auto need = [&](size_t b) { bytes = ((bytes + 255) & ~(size_t)255) + b; };
        need(N * sizeof(PBox));                 // pbox
        need(N * 4); need(N * 4);               // order x2
        need(N * 4); need(N * 4);               // pos_seg x2
        need(N * 4); need(N);                   // L, flag
        need((N / SCAN_TILE + 2) * 4);          // scan sums
        need(N * sizeof(BSeg));                 // segs of all levels
        need(max_segs_level * 12 * 8);          // bkey
        need(max_segs_level * NB * 6 * 8);      // binkey
        need(max_segs_level * NB * 4);          // bincnt
        need(N * sizeof(BNodeD));               // nodes (N - 1 internal)
        need(N * sizeof(SmallSeg));             // small ranges
        need(N * RTX_WIDE * 4); need(N * RTX_WIDE);   // cost16, how16
        need(N * RTX_BVH4 * 4); need(N * RTX_BVH4);   // cost4, how4
        need(N * RTX_WIDE * 4);                 // kids (16-wide emission, reused by the 4-wide one)
        need(N * 4); need(N); need(N * 4);      // wnode_bin, n_kids, first
        need(sizeof(BCtr));
        bytes += 4096;
        BCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        BCHECK(hipMalloc((void **)&ar.base, bytes));
        ar.cap = bytes;
        PBox *pbox = ar.take<PBox>(N);
        int *order[2] = {ar.take<int>(N), ar.take<int>(N)};
        int *pos_seg[2] = {ar.take<int>(N), ar.take<int>(N)};
        unsigned *L = ar.take<unsigned>(N);
        uint8_t *flag = ar.take<uint8_t>(N);
        unsigned *sums = ar.take<unsigned>(N / SCAN_TILE + 2);
        BSeg *segs = ar.take<BSeg>(N);
        uint64_t *bkey = ar.take<uint64_t>(max_segs_level * 12);
        uint64_t *binkey = ar.take<uint64_t>(max_segs_level * NB * 6);
        unsigned *bincnt = ar.take<unsigned>(max_segs_level * NB);
        BNodeD *nodes = ar.take<BNodeD>(N);
        SmallSeg *small = ar.take<SmallSeg>(N);
        float *cost16 = ar.take<float>(N * RTX_WIDE);
        uint8_t *how16 = ar.take<uint8_t>(N * RTX_WIDE);
        float *cost4 = ar.take<float>(N * RTX_BVH4);
        uint8_t *how4 = ar.take<uint8_t>(N * RTX_BVH4);
        int *kids = ar.take<int>(N * RTX_WIDE);
        int *wnode_bin = ar.take<int>(N);
        uint8_t *n_kids = ar.take<uint8_t>(N);
        unsigned *first = ar.take<unsigned>(N);
        BCtr *ctr = ar.take<BCtr>(1);

        BCHECK(hipMemcpyAsync(pbox, prims, N * sizeof(PBox), hipMemcpyHostToDevice, st));
        lap(&out->ms_upload);
        hipLaunchKernelGGL(k_init, dim3(grid(n)), dim3(BT), 0, st, n, order[0], pos_seg[0], segs, small, ctr);

        // ---- 1. the binned levels
        BCtr h_ctr;
        int cur = 0, S = n > SWEEP_MAX ? 1 : 0, base = 0;
        while (S > 0) {
            if ((size_t)S > max_segs_level || (size_t)base + (size_t)S > N) {
                if (err) *err = "device hierarchy build: more ranges in a level than prims allow (internal error)";
                rc = (int)hipErrorAssert;
                goto done;
            }
            level_base.push_back(base);
            level_count.push_back(S);
            BSeg *lv = segs + base, *nx = segs + base + S;
            hipLaunchKernelGGL(k_level_init, dim3(grid((int64_t)S * NB)), dim3(BT), 0, st, S, bkey, binkey, bincnt, ctr);
            hipLaunchKernelGGL(k_bounds, dim3(grid(n)), dim3(BT), 0, st, n, (const int *)order[cur], (const int *)pos_seg[cur], (const PBox *)pbox, bkey);
            hipLaunchKernelGGL(k_bin, dim3(grid(n)), dim3(BT), 0, st, n, (const int *)order[cur], (const int *)pos_seg[cur], (const PBox *)pbox,
                               (const uint64_t *)bkey, binkey, bincnt);
            hipLaunchKernelGGL(k_split, dim3(grid(S)), dim3(BT), 0, st, S, lv, nx, (const uint64_t *)bkey, (const uint64_t *)binkey,
                               (const unsigned *)bincnt, nodes, small, ctr);
            hipLaunchKernelGGL(k_flag, dim3(grid(n)), dim3(BT), 0, st, n, (const int *)order[cur], (const int *)pos_seg[cur], (const PBox *)pbox,
                               (const BSeg *)lv, flag, L);
            scan_exclusive(st, L, n, sums, nullptr);
            hipLaunchKernelGGL(k_scatter, dim3(grid(n)), dim3(BT), 0, st, n, (const int *)order[cur], (const int *)pos_seg[cur], (const BSeg *)lv,
                               (const uint8_t *)flag, (const unsigned *)L, order[cur ^ 1], pos_seg[cur ^ 1]);
            BCHECK(hipMemcpyAsync(&h_ctr, ctr, sizeof(BCtr), hipMemcpyDeviceToHost, st));
            BCHECK(hipStreamSynchronize(st));
            cur ^= 1;
            base += S;
            S = (int)h_ctr.n_next;
        }
        out->levels = (int)level_base.size();
        BCHECK(hipMemcpyAsync(&h_ctr, ctr, sizeof(BCtr), hipMemcpyDeviceToHost, st));
        BCHECK(hipStreamSynchronize(st));
        lap(&out->ms_tree);
        // ---- 2. small ranges
        if (h_ctr.n_small)
            hipLaunchKernelGGL(k_small, dim3(grid(h_ctr.n_small)), dim3(BT), 0, st, (int)h_ctr.n_small, (const SmallSeg *)small, order[cur],
                               (const PBox *)pbox, nodes, ctr, cost16, how16, cost4, how4);
        lap(&out->ms_small);
        // ---- 3. collapse costs of the binned nodes, deepest level first
        for (size_t lv = level_base.size(); lv-- > 0;)
            hipLaunchKernelGGL(k_dp, dim3(grid(level_count[lv])), dim3(BT), 0, st, level_count[lv], (const BSeg *)(segs + level_base[lv]),
                               (const BNodeD *)nodes, cost16, how16, cost4, how4);
        lap(&out->ms_dp);
        // ---- 4. emission, twice
        for (int which = 0; which < 2; ++which) {
            const int W = which == 0 ? RTX_WIDE : RTX_BVH4;
            const uint8_t *how = which == 0 ? how16 : how4;
            BCHECK(hipMemsetAsync(wnode_bin, 0, sizeof(int), st));   // wide node 0 = the cut below binary node 0
            int level_first = 0, cnt = 1, depth = 0;
            while (cnt > 0) {
                ++depth;
                if ((size_t)level_first + (size_t)cnt > N) {
                    if (err) *err = "device hierarchy build: more wide nodes than prims (internal error)";
                    rc = (int)hipErrorAssert;
                    goto done;
                }
                if (which == 0)
                    hipLaunchKernelGGL(k_emit_count<RTX_WIDE>, dim3(grid(cnt)), dim3(BT), 0, st, level_first, cnt, (const int *)wnode_bin,
                                       (const BNodeD *)nodes, how, kids, n_kids, first, ctr, which);
                else
                    hipLaunchKernelGGL(k_emit_count<RTX_BVH4>, dim3(grid(cnt)), dim3(BT), 0, st, level_first, cnt, (const int *)wnode_bin,
                                       (const BNodeD *)nodes, how, kids, n_kids, first, ctr, which);
                scan_exclusive(st, first + level_first, cnt, sums, &ctr->scan_total);
                if (which == 0)
                    hipLaunchKernelGGL(k_emit_next<RTX_WIDE>, dim3(grid(cnt)), dim3(BT), 0, st, level_first, cnt, level_first + cnt, wnode_bin,
                                       (const int *)kids, (const uint8_t *)n_kids, first);
                else
                    hipLaunchKernelGGL(k_emit_next<RTX_BVH4>, dim3(grid(cnt)), dim3(BT), 0, st, level_first, cnt, level_first + cnt, wnode_bin,
                                       (const int *)kids, (const uint8_t *)n_kids, first);
                BCHECK(hipMemcpyAsync(&h_ctr, ctr, sizeof(BCtr), hipMemcpyDeviceToHost, st));
                BCHECK(hipStreamSynchronize(st));
                level_first += cnt;
                cnt = (int)h_ctr.scan_total;
            }
            const int n_wide = level_first;
            if (which == 0) {
                BCHECK(hipMalloc((void **)&d_wide, (size_t)n_wide * RTX_WIDE * sizeof(DWideChild)));
                hipLaunchKernelGGL(k_write16, dim3(grid((int64_t)n_wide * RTX_WIDE)), dim3(BT), 0, st, n_wide, (const int *)kids, (const uint8_t *)n_kids,
                                   (const unsigned *)first, (const BNodeD *)nodes, (const int *)order[cur], (const PBox *)pbox, centre[0], centre[1],
                                   centre[2], d_wide);
                out->stats.nodes = n_wide;
                out->stats.depth = depth;
                out->stats.prim_children = (int64_t)h_ctr.kids_sum[0] - ((int64_t)n_wide - 1);
            } else {
                BCHECK(hipMalloc((void **)&d_bvh4, (size_t)n_wide * sizeof(DNode4)));
                hipLaunchKernelGGL(k_write4, dim3(grid(n_wide)), dim3(BT), 0, st, n_wide, (const int *)kids, (const uint8_t *)n_kids,
                                   (const unsigned *)first, (const BNodeD *)nodes, (const int *)order[cur], (const PBox *)pbox, centre[0], centre[1],
                                   centre[2], d_bvh4);
                out->stats.nodes4 = n_wide;
                out->stats.depth4 = depth;
            }
            (void)W;
        }
        BCHECK(hipStreamSynchronize(st));
        BCHECK(hipGetLastError());
        lap(&out->ms_emit);
        out->wide = d_wide;
        out->bvh4 = d_bvh4;
        d_wide = nullptr;
        d_bvh4 = nullptr;
    }
done:
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (ar.base) (void)hipFree(ar.base);
    if (d_wide) (void)hipFree(d_wide);
    if (d_bvh4) (void)hipFree(d_bvh4);
    return rc;
}
