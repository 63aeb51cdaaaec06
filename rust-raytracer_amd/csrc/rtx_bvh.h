// rtx_bvh.h — host builder of the 16-wide box-list hierarchy (see rtx_bvh.cpp).
#pragma once
#include <stdint.h>

#include <vector>

#include "rtx_device.h"

struct RtxPrimBox {
    double mn[3], mx[3];
    uint32_t ref;        // RTX_WREF_TRI | triangle index  or  RTX_WREF_SPHERE | sphere index
};
struct RtxWideStats {
    int64_t nodes = 0, prim_children = 0;
    int depth = 0;
};
// Boxes are stored relative to `centre`, rounded outward to f32.
void rtx_build_wide_bvh(const std::vector<RtxPrimBox> &prims, const double centre[3], std::vector<DWideChild> &out,
                        RtxWideStats *stats);
