// pf_warm.h -- L2 warm-up helper shared by the kernel files (device code only).
#pragma once
#include "pf_device.h"

// Each XCD has its own 4 MiB L2 and a denoising step streams more bytes than that through it, so a launch finds its
// weights in the Infinity Cache, not in L2, and a CU draws only ~13 B/clk from there -- a third of what one wave of
// the row-group kernels consumes.  The launch BEFORE therefore carries PF_WARM_BLOCKS extra waves on otherwise idle
// CUs that touch every 128-byte line of the next launch's quad streams.  The region is cut in nsl slices; the caller
// deals the slices so that every XCD (workgroups are dealt round-robin over the 8 XCDs) receives each slice once.
__device__ __forceinline__ void l2_warm(pf_gcf base, const int bytes, const int slice, const int nsl, const int lane) {
    const int per = ((bytes + nsl - 1) / nsl + 8191) & ~8191;           // whole 64-line sweeps
    const int beg = slice * per, end = min(beg + per, bytes);
    float acc = 0.f;
#pragma unroll 4
    for (int off = beg + lane * 128; off < end; off += 8192) acc += base[off >> 2];
    asm volatile("" ::"v"(acc));                                        // the loads are the point: keep them
}
