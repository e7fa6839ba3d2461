// pf_rg_common.h -- device helpers shared by the row-group kernel files (pf_rg.hip: one or two waves per 4-row group;
// pf_rgk.hip: four waves per group with the K dimension of the scalar Linears split over them).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include "pf_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrtf_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rsqf_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * rcpf_(1.0f + __expf(-x)); }

// all 16 blocks read their A rows from block ABID (cbsz = 4) / each group of 4 blocks from its block ABID (cbsz = 2)
template <int ABID>
__device__ __forceinline__ f32x4 mfma_b4(const float a, const float b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}
template <int ABID>
__device__ __forceinline__ f32x4 mfma_b2(const float a, const float b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, ABID, 0);
}

// x[l] + x[l ^ 32], x[l] + x[l ^ 16]: the swap instructions exchange register halves / odd-even rows of two
// registers (the compiler's builtin mis-assigns the second result in ROCm 7.2, hence the asm; the s_nop covers the
// VALU-write -> permlane-swap-read hazard the assembler cannot see)
__device__ __forceinline__ float xsum32(const float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xsum16(const float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// sum over the four lane groups g (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float gsum(const float v) { return xsum16(xsum32(v)); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
#ifndef RG_QUAD_MAX
#define RG_QUAD_MAX 4096                       // launches of up to this many item slots use four-wave workgroups
#endif
#define RG_CPASS 16                            // compact work lists: up to 64 * RG_CPASS regions per launch
// integer DPP moves for wave scans: lanes without a source (or rows outside ROW_MASK) read 0
template <int CTRL>
__device__ __forceinline__ int dpp_i(const int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_ir(const int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
// sum over q (lanes l, l^4, l^8, l^12 of a 16-lane row): row_ror 4 and 8
__device__ __forceinline__ float qsum(float v) {
    v += dpp_f<0x124>(v);
    v += dpp_f<0x128>(v);
    return v;
}
// sum over all 16 blocks a (every lane with the same row i)
__device__ __forceinline__ float asum(const float v) { return gsum(qsum(v)); }

// wave-private LDS scratch of one row group
#define RG_T1_STRIDE 132                      // SD -> SA transposition: [4 rows][128 (+4 pad)]
#define RG_TV_STRIDE 52                       // VD -> VA transposition: [4 rows][3 coordinates][16] + 3 (channel 16) + pad
// (t1 double-buffered and tv per wave for the two-wave form, where the SiLU output is exchanged between the waves)
struct __attribute__((aligned(16))) RgLds {
    float t1[2][4 * RG_T1_STRIDE];
    float tv[2][4 * RG_TV_STRIDE];
};
// position of a wave in its workgroup's chain: SPLIT kernels run a 4-row group on TWO waves (two SIMDs), wave h owning
// outputs 64h..64h+63 of every 128-output scalar Linear (its own quad stream: half the main / rbf / sh quads) while the
// cheap vector channel and the gates are computed by both; the halves meet in the T1 buffer (one workgroup barrier per
// GVP).  A launch with fewer groups than SIMDs is bound by the latency of the chain, and a wave cannot stream its
// weights faster than ~16 B/clk.
struct RgWave {
    int half;      // 0 / 1 (0 when not split)
    int par;       // T1 buffer of the next exchange (alternates)
};
__device__ __forceinline__ int pperm(const int u) { return (u & 3) * 4 + (u >> 2); }

// register prefetch ring over the quad stream
// (D quads deep; every block of the stream is a multiple of RG_PAD quads and D divides RG_PAD, so quad qi of a block
// always sits in slot qi % D)
template <int D>
struct RgRing {
    f32x4 q[D];
    const f32x4 PF_AS1* p;                    // quad 0 of the current block, + lane
};
template <int D>
__device__ __forceinline__ void ring_start(RgRing<D>& r, pf_gcf stream, const int lane) {
    r.p = reinterpret_cast<const f32x4 PF_AS1*>(stream) + lane;
    static_for<0, D>([&](auto I) { r.q[decltype(I)::value] = r.p[decltype(I)::value * 64]; });
}
#ifndef RG_SB
#define RG_SB 1                               // quads between scheduling barriers (keeps the ring loads where they are issued)
#endif
#ifndef RG_D1
#define RG_D1 24                              // ring depth at 4 rows per wave (one quad per 4 MFMAs)
#endif
#ifndef RG_D2
#define RG_D2 12                              // ... at 8 rows per wave (one quad per 8 MFMAs)
#endif
template <int RG> struct RgDepth {
    static constexpr int D = RG == 1 ? RG_D1 : RG_D2;
    static_assert(RG_PAD % D == 0, "ring depth must divide the block padding");
};

template <int VI_, int NEXTRA_, int NH_, bool SIG_>
struct RgSpec {
    static constexpr int VI = VI_, NEXTRA = NEXTRA_, NH = NH_;
    static constexpr bool SIG = SIG_, H17 = VI_ == 17;
};
typedef RgSpec<17, PF_R, 2, true> SpecMsg0;         // first message GVP: [h_src, rbf] / [xhat, v_src]
typedef RgSpec<16, 0, 2, true> SpecGen;             // 128 + 16 -> 128 + 16
typedef RgSpec<16, 0, 1, false> SpecHeadLast;       // last noise-head GVP: 64 scalars, 1 vector, identity gate

// what a GVP leaves pending for the next block: its Vu (VD layout) and gate bias -- the gates themselves are computed
// from its SiLU output inside the next block (or the flush block)
template <int RG>
struct RgCarry {
    f32x4 vu[RG];
    float bg;
};

// GVPLayerNorm (gvp.py:159-166) on the SA / VA layouts: lane 4a+i holds 8 of the 128 scalars of row i, the row's
// statistics are a sum over the 16 blocks; vector norms need the three coordinates of a channel (lane groups g)
template <int RG>
__device__ __forceinline__ void rg_layernorm(pf_gcf lw, pf_gcf lb, float (&X)[RG][8], float (&Va)[RG][4], const int lane) {
    const int a = lane >> 2, g = lane >> 4;
    const f32x4 w0 = reinterpret_cast<const f32x4 PF_AS1*>(lw)[2 * a], w1 = reinterpret_cast<const f32x4 PF_AS1*>(lw)[2 * a + 1];
    const f32x4 b0 = reinterpret_cast<const f32x4 PF_AS1*>(lb)[2 * a], b1 = reinterpret_cast<const f32x4 PF_AS1*>(lb)[2 * a + 1];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) sum += X[r][m];
        const float mean = asum(sum) * (1.0f / 128.0f);
        float var = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) { const float c = X[r][m] - mean; var = fmaf(c, c, var); }
        const float rstd = rsqf_(asum(var) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            X[r][m] = (X[r][m] - mean) * rstd * w0[m] + b0[m];
            X[r][4 + m] = (X[r][4 + m] - mean) * rstd * w1[m] + b1[m];
        }
        float vn = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) vn += fmaxf(gsum(Va[r][t] * Va[r][t]), 1e-8f);
        vn = qsum(vn);
        const float rden = rcpf_(sqrtf_(vn * (1.0f / 16.0f) + 1e-5f) + 1e-5f);
#pragma unroll
        for (int t = 0; t < 4; ++t) Va[r][t] = g < 3 ? Va[r][t] * rden : 0.f;
    }
}

// Scalar encoders on the fly (dynamics_gvp.py:107-117, 143-151): h = LayerNorm(SiLU(W [feat, t] + b)) of the rows' nodes,
// straight into the SA layout (lane 4a+i: features 8a..8a+7 of row i).  With layer 0 on the row-group kernels only the
// rows that layer actually reads are ever encoded -- a few thousand of the 8 k nodes of a pruned config-2 step -- and the
// [N][128] encoder output is neither written nor gathered.  nt (0 prot, 1 pharm) is wave-uniform.
template <int RG>
__device__ __forceinline__ void rg_encode(const EncodeParams& ep, const int nt, const int (&node)[RG], float (&X)[RG][8], const int lane) {
    const int a = lane >> 2;
    const int nf = nt ? ep.pharm_nf : ep.rec_nf;
    pf_gcf Wt = (pf_gcf)ep.w[nt] + 8 * a;                            // [nf + 1][128], input-major
    const f32x4 b0 = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)ep.b[nt])[2 * a], b1 = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)ep.b[nt])[2 * a + 1];
    pf_gcf in[RG];
    float tt[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        in[r] = nt ? (pf_gcf)ep.pharm_h + (size_t)(node[r] - ep.Np) * nf : (pf_gcf)ep.prot_h0 + (size_t)node[r] * nf;
        tt[r] = ep.t ? ((pf_gcf)ep.t)[((const int PF_AS1*)ep.gid)[node[r]]] : ep.t_scalar;
#pragma unroll
        for (int m = 0; m < 4; ++m) { X[r][m] = b0[m]; X[r][4 + m] = b1[m]; }
    }
    for (int k = 0; k <= nf; ++k) {
        const f32x4 w0 = reinterpret_cast<const f32x4 PF_AS1*>(Wt + k * PF_S)[0], w1 = reinterpret_cast<const f32x4 PF_AS1*>(Wt + k * PF_S)[1];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const float x = k < nf ? in[r][k] : tt[r];
#pragma unroll
            for (int m = 0; m < 4; ++m) { X[r][m] = fmaf(w0[m], x, X[r][m]); X[r][4 + m] = fmaf(w1[m], x, X[r][4 + m]); }
        }
    }
    const f32x4 lw0 = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)ep.ln_w[nt])[2 * a], lw1 = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)ep.ln_w[nt])[2 * a + 1];
    const f32x4 lb0 = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)ep.ln_b[nt])[2 * a], lb1 = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)ep.ln_b[nt])[2 * a + 1];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) { X[r][m] = siluf_(X[r][m]); sum += X[r][m]; }
        const float mean = asum(sum) * (1.0f / 128.0f);
        float var = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) { const float c = X[r][m] - mean; var = fmaf(c, c, var); }
        const float rstd = rsqf_(asum(var) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            X[r][m] = (X[r][m] - mean) * rstd * lw0[m] + lb0[m];
            X[r][4 + m] = (X[r][4 + m] - mean) * rstd * lw1[m] + lb1[m];
        }
    }
}

// quad_perm broadcast of lane II of every group of four lanes
template <int II>
__device__ __forceinline__ float quad_bcast(const float v) { return dpp_f<II * 0x55>(v); }


}  // namespace
