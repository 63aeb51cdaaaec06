// Order of a step's shadow groups by the cell of their hit point (DShadowQ.cell -> DShadowQ.order): a stable LSD radix sort of
// (24-bit cell, group index) pairs, three 8-bit passes of rocPRIM's device radix sort (the vendor's primitive; its workspace comes from
// the caller, it allocates nothing and does not synchronise).  Why: the rays of the second generation and later leave their hit
// points in the order of the ray tree, neighbours in the queue lie anywhere in the scene, and a wave of the per-lane shadow walks then
// runs as long as its longest feeler with a third of its lanes at work; feelers of one cell towards one light walk the same nodes
// (tools/exp/sort_coherence.py: the shadow product of 3.45 M feelers in heptoroid 2.65 ms in tree order, 1.29 ms sorted).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "rtx_device.h"

#include <cstdlib>
#define RTX_CELL_BITS 24u
// the sort looks at the top `bits` of the 24-bit cell (RTX_SORT_BITS, A/B: coarser cells, fewer passes)
static unsigned sort_begin_bit() {
    static const unsigned b = [] { const char *e = std::getenv("RTX_SORT_BITS"); unsigned v = e ? (unsigned)std::atoi(e) : RTX_CELL_BITS; return RTX_CELL_BITS - (v < 1u ? 1u : (v > RTX_CELL_BITS ? RTX_CELL_BITS : v)); }();
    return b;
}

__global__ void k_order_count(const uint32_t *count, uint32_t *order) { order[0] = *count; }
// what a sort that could not be launched leaves behind: the order the launches would have had without it
__global__ void k_order_identity(uint32_t *order, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = i;
}

size_t rtxk_sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    uint32_t *k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, rocprim::counting_iterator<uint32_t>(0u), k, n, sort_begin_bit(), RTX_CELL_BITS, (hipStream_t) nullptr);
    return bytes;
}

void rtxk_sort_groups(rtx_stream_t stream, const DShadowQ &sq, const RtxSortBuf &sb, uint32_t n) {
    if (n == 0 || !sq.cell || !sq.order || !sb.cell_sorted) return;
    size_t bytes = sb.tmp_bytes;
    // (values: the group indices 0, 1, 2, .. from a counting iterator; the sort is stable, so the groups of one cell keep their ray order
    // and the stale slots past the queue's count, whose cell is ~0, stay behind every group of the step)
    if (rocprim::radix_sort_pairs(sb.tmp, bytes, sq.cell, sb.cell_sorted, rocprim::counting_iterator<uint32_t>(0u), sq.order + 1, n, sort_begin_bit(),
                                  RTX_CELL_BITS, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_order_identity, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, sq.order + 1, n);
    }
    hipLaunchKernelGGL(k_order_count, dim3(1), dim3(1), 0, (hipStream_t)stream, sq.count, sq.order);
}

// The same for the rays of a step (DPool.cell -> RtxSortBuf.ray_order): the closest-hit launch of a later generation in scenes with
// transmissive materials takes its rays by direction octant, then by the cell of their origin (tools/exp/sort_coherence.py: 3.8 M rays
// of the fourth generation in heptoroid, k_nearest<8> 1.75 ms in tree order, 1.31 ms in this one).
void rtxk_sort_rays(rtx_stream_t stream, const DPool &pool, const RtxSortBuf &sb, uint32_t begin, uint32_t n) {
    if (n == 0 || !pool.cell || !sb.ray_order || !sb.cell_sorted) return;
    size_t bytes = sb.tmp_bytes;
    if (rocprim::radix_sort_pairs(sb.tmp, bytes, pool.cell + begin, sb.cell_sorted, rocprim::counting_iterator<uint32_t>(0u), sb.ray_order, n, 0u,
                                  RTX_CELL_BITS, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_order_identity, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, sb.ray_order, n);
    }
}
