// rtx_build.h — device-side build of the SAH hierarchy (rtx_build.hip): the same binned-SAH binary tree, collapse DP and
// breadth-first emissions as the host builder (rtx_bvh.cpp), as gfx950 kernels.
#pragma once
#include <stdint.h>

#include <string>

#include "rtx_bvh.h"

// scenes with fewer finite prims than this keep the host builder by default (RTX_BUILD_HIERARCHY overrides)
#define RTX_DEVICE_BUILD_MIN 4096

struct RtxDeviceHierarchy {
    DWideChild *wide = nullptr;   // device: nodes x RTX_WIDE children (hipMalloc: the caller owns it)
    DNode4 *bvh4 = nullptr;       // device: nodes4 records (hipMalloc: the caller owns it)
    RtxWideStats stats;
    int levels = 0;               // levels of the binary tree split with binning passes (ranges of more than 12 prims)
    double ms_upload = 0.0, ms_tree = 0.0, ms_small = 0.0, ms_dp = 0.0, ms_emit = 0.0;   // filled when `timing`
};

// Builds both cuts of the SAH tree over `prims` (n >= 2 finite prims, host array) on the CURRENT device.  The tree is the one
// rtx_build_hierarchies() builds (same splits, same collapse, same emission order), so the two arrays equal the host builder's
// (tests/test_gpu_build.py compares them).  Returns 0, or a hipError_t code with *err set.  Synchronous: when it returns the arrays
// are complete.
int rtxk_build_hierarchies_device(const RtxPrimBox *prims, int64_t n, const double centre[3], RtxDeviceHierarchy *out, bool timing,
                                  std::string *err);
