// pf_rg.hip -- "row-group" kernels of the denoising path: the inference form of the conv layers and the noise head
// (gfx950 only).
//
// Same mathematics as pf_kernels.hip (GVP.forward gvp.py:89-116, GVPMultiEdgeConv gvp.py:459-551, GVPLayerNorm
// gvp.py:159-166, NoisePredictionBlock dynamics_gvp.py:37-42), different mapping onto the matrix cores: at the
// headline batch (32 graphs) a launch has a few hundred to a few thousand rows, and a 32-row MFMA tile per CU leaves
// most of the chip idle while every tile pays the latency of a whole GVP chain.  Here a wave owns RG groups of FOUR
// rows and all 128 output features, on v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, exact fp32, 64 FLOP/clk/SIMD like
// the big shapes, 8 cycles per instruction):
//
//   D_b[i][j] += A_sel[i] * B_b[j]        b = 0..15 blocks, i = row, lane 4b+j = output feature
//
//   * weights are the B operand, one VGPR image per k-step and half of 64 outputs: lane f holds W[f][k];
//   * activations are the A operand; the CBSZ/ABID broadcast controls pick which 4-lane block of the A register feeds
//     all (cbsz=4) or each group of four (cbsz=2) blocks, so ONE register holds 16 k-steps:
//       "SA layout": lane 4a+i holds features 8a..8a+7 of row i in 8 registers   (k-step (m, a) <-> feature 8a+m)
//       "VA layout": lane 16g+4q+i holds V[row i][channel 4t+q][coordinate g], t = 0..3 (4 registers; g = 3 unused)
//   * results come out with the feature on the lane ("SD" / "VD": register i = row i, lane f / lane 16g+u), and go
//     back to the A layouts through a few hundred bytes of wave-private LDS (no barriers: one wave, in-order DS);
//   * the scalar->vector gates and to_scalar_output split K over the four lane groups (cbsz=2) and are summed with
//     v_permlane32_swap / v_permlane16_swap.
//
// A wave therefore needs no partner: no workgroup barriers, 4-row granularity (ragged regions waste at most 3 rows),
// and a GVP level costs ~350 MFMAs x 8 cycles per group.  All weights of a chain are one contiguous stream of 1-KiB
// "quads" in consumption order (pf_host.cpp: pack_gvp_rg), prefetched 12-24 quads ahead into registers across GVP
// boundaries; the chain is software-pipelined (rg_gvp).  Per row the weight traffic is 4-8x that of a 32-row tile, yet
// these kernels won at every batch size measured (32-1024 graphs): 4-row granularity, no barriers, no padding of the
// ragged regions.  The 32-row tile kernels of pf_kernels.hip remain for training.  Launch policy: pf_host.cpp rg_mode;
// measurements and the cycle accounting: DESIGN.md section 4.1, profiles/README.md.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include "pf_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

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

// In-kernel cycle stamps (diagnostic builds only: -DPF_STAMPS; no stamp executes in the product build): lane 0 of the
// first 64 waves writes s_memtime at the phase boundaries of the chain (tools/stamps_rg.py)
#ifdef PF_STAMPS
__device__ unsigned long long* g_rg_stamps = nullptr;
__device__ int g_rg_stamp_off = 0;                    // first recorded item (pfk_rg_set_stamp_offset)
struct RgStamp {
    int k = 0, id = 0;                                // id: item of the wave (64 items from g_rg_stamp_off are recorded)
    __device__ __forceinline__ void operator()(const int lane) {
        const int rid = id - g_rg_stamp_off;
        if (lane == 0 && g_rg_stamps && rid >= 0 && rid < 64 && k < 64) g_rg_stamps[rid * 64 + k] = __builtin_amdgcn_s_memtime();
        ++k;
    }
};
static int g_rg_stamp_which = -1, g_rg_stamp_seen = 0;      // host: which rg launch since the selector was set stamps
static unsigned long long* g_rg_stamp_host_buf = nullptr;
static void rg_stamp_arm(hipStream_t s) {
    unsigned long long* on = (g_rg_stamp_seen++ == g_rg_stamp_which) ? g_rg_stamp_host_buf : nullptr;
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rg_stamps), &on, sizeof(on), 0, hipMemcpyHostToDevice, s);
}
#else
struct RgStamp {
    int id = 0;
    __device__ __forceinline__ void operator()(const int) {}
};
#endif

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

// ---------------------------------------------------------------------------------------------
// One block of the pipelined chain = one GVP on RG groups of four rows (+ the pending gates of the previous one).
//   X  [RG][8]  in: scalar input, SA layout          out: SiLU output, SA layout (next GVP's input)
//   Va [RG][4]  in (PREV == 0): vector input, VA layout; with a pending gate it is produced here
//   R, XH       first message GVP only: rbf image (lane 4a+i: rbf_a(d_i)) and unit x_diff (lane 16g+i: xhat_i[g])
//   slo, shi    SiLU output, SD layout (register i: row i; lane f: feature f / 64+f)
//   carry       in: pending Vu / gate bias of the previous GVP (PREV != 0); out: this GVP's
//   PREV        0: first GVP of a chain; 1: previous GVP gates with a sigmoid; 2: identity
//   VZERO       the 16 node-vector channels are identically zero (conv layer 0, first GVP): only xhat feeds Vh
// ---------------------------------------------------------------------------------------------
template <class S, int RG, int D, int PREV, bool VZERO, bool SPLIT = false>
__device__ __forceinline__ void rg_gvp(RgRing<D>& ring, float (&X)[RG][8], float (&Va)[RG][4], const float (&R)[RG],
                                       const float (&XH)[RG], f32x4 (&slo)[RG], f32x4 (&shi)[RG], RgCarry<RG>& carry,
                                       RgLds* lds, const int lane, RgStamp& stamp, RgWave& wv) {
    constexpr int NH = SPLIT ? 1 : S::NH;             // halves of 64 outputs computed by this wave
    const int hcol = (SPLIT && S::NH == 2) ? 64 * wv.half : 0;
    const int tvw = SPLIT ? wv.half : 0;
    static_assert(!(VZERO && PREV != 0), "VZERO is a property of a chain's first GVP");
    stamp(lane);                                      // 0: block start
    const int a = lane >> 2, i = lane & 3, g = lane >> 4, q = a & 3, u = lane & 15;
    const int gg = g < 3 ? g : 2;
    // with one row group per wave every accumulator is split in two (even / odd image of a quad): back-to-back MFMAs on
    // one accumulator issue every ~13 cycles instead of 8
    constexpr int NA = RG == 1 ? 2 : 1;
    f32x4 lo[RG * NA], hi[RG * NA], vh[RG * NA], vu[RG * NA], gd[RG * NA];
#pragma unroll
    for (int r = 0; r < RG * NA; ++r) {
        lo[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        hi[r] = lo[r]; vh[r] = lo[r]; vu[r] = lo[r]; gd[r] = lo[r];
    }
    // index of the accumulator of row group r for image j of a quad; fold() adds the halves together
    auto acc = [](const int r, const int j) { return RG == 1 ? (j & 1) : r; };
    auto fold = [&](f32x4 (&x)[RG * NA]) {
        if constexpr (RG == 1) { x[0] += x[1]; x[1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    };
    f32x4 cq = {0.f, 0.f, 0.f, 0.f}, xhq = cq, w16 = cq;
    f32x4 VhA[RG];
    float SH[RG], SH16[RG], Vh16[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) { VhA[r] = cq; SH[r] = 0.f; SH16[r] = 0.f; Vh16[r] = 0.f; }

    constexpr RgSched QQ = rg_sched(S::VI, S::NEXTRA, NH, PREV != 0);
    static_for<0, QQ.nq>([&](auto QI) {
        constexpr int qi = decltype(QI)::value;
        constexpr RgSched Q = rg_sched(S::VI, S::NEXTRA, NH, PREV != 0);
        const f32x4 w = ring.q[qi % D];
        ring.q[qi % D] = ring.p[(qi + D) * 64];
        // main k-step group of this quad, or -1
        constexpr int kmain = (qi >= Q.q_a && qi < Q.q_a + 4 * NH) ? qi - Q.q_a
                            : (qi >= Q.q_b && qi < Q.q_b + 8 * NH) ? 4 * NH + qi - Q.q_b
                            : (qi >= Q.q_cc && qi < Q.q_cc + 12 * NH) ? 12 * NH + qi - Q.q_cc
                            : (qi >= Q.q_d && qi < Q.q_d + 8 * NH) ? 24 * NH + qi - Q.q_d : -1;
        if constexpr (qi == Q.q_c) {
            cq = w;                                   // [bias lo, bias hi, gate bias, Wh[0][16] on the xhat lanes]
        } else if constexpr (S::H17 && qi == Q.q_xh) {
            xhq = w;                                  // [Wh[0][:] image, Wu[16][:] image, sh16 column lo, hi]
#pragma unroll
            for (int r = 0; r < RG; ++r) vh[r] = mfma_b2<0>(XH[r], w[0], vh[r]);
        } else if constexpr (S::H17 && qi == Q.q_xh + 1) {
            w16 = w;                                  // Wh[1 + 4t + q][16], t = 0..3
        } else if constexpr (kmain >= 0) {
            constexpr int half = kmain % NH, mq = kmain / NH, m = mq / 4, aq = mq % 4;
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    if constexpr (half == 0) lo[acc(r, j)] = mfma_b4<4 * aq + j>(X[r][m], w[j], lo[acc(r, j)]);
                    else hi[acc(r, j)] = mfma_b4<4 * aq + j>(X[r][m], w[j], hi[acc(r, j)]);
                }
            });
        } else if constexpr (PREV != 0 && qi >= Q.q_gate && qi < Q.q_gate + 8) {
            constexpr int m = qi - Q.q_gate;          // pending gates of the previous GVP, K split over the lane groups
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
#pragma unroll
                for (int r = 0; r < RG; ++r) gd[acc(r, j)] = mfma_b2<j>(X[r][m], w[j], gd[acc(r, j)]);
            });
            if constexpr (m == 7) {                   // sum the K quarters, activation, gate the pending Vu, publish
                stamp(lane);                          // 1: first main k-steps + pending gates issued
                fold(gd);
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    float* tv = lds[r].tv[tvw];
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        float gv = gsum(gd[r][ii]) + carry.bg;
                        if constexpr (PREV == 1) gv = sigmoidf_(gv);
                        const float vd = gv * carry.vu[r][ii];
                        if (lane < 48) tv[ii * RG_TV_STRIDE + g * 16 + pperm(u)] = vd;
                    }
                }
            }
        } else if constexpr (qi >= Q.q_vh && qi < Q.q_vh + 4) {
            constexpr int t = qi - Q.q_vh;
            if constexpr (t == 0 && PREV != 0) {      // the gated vectors of the previous GVP are back: VA layout
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(&lds[r].tv[tvw][i * RG_TV_STRIDE + gg * 16 + 4 * q]);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) Va[r][tt] = g < 3 ? v4[tt] : 0.f;
                }
            }
            if constexpr (!VZERO) {
                static_for<0, 4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
#pragma unroll
                    for (int r = 0; r < RG; ++r) vh[acc(r, j)] = mfma_b2<j>(Va[r][t], w[j], vh[acc(r, j)]);
                });
            }
            if constexpr (t == 3) {                   // Vh complete: hidden channel 16 on the VALU, publish Vh
                stamp(lane);                          // 2: Vh issued
                fold(vh);
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    if constexpr (S::H17) {
                        float p = XH[r] * cq[3];
                        if constexpr (!VZERO) {
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) p = fmaf(Va[r][tt], w16[tt], p);
                        }
                        Vh16[r] = qsum(p);
                    }
                    float* tv = lds[r].tv[tvw];
                    if (lane < 48) {
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) tv[ii * RG_TV_STRIDE + g * 16 + pperm(u)] = vh[r][ii];
                    }
                    if constexpr (S::H17) {
                        if (q == 0 && g < 3) tv[i * RG_TV_STRIDE + 48 + g] = Vh16[r];
                    }
                }
            }
        } else if constexpr (qi >= Q.q_vu && qi < Q.q_vu + 4) {
            constexpr int t = qi - Q.q_vu;
            if constexpr (t == 0) {                   // Vh is back from LDS: A images of the Vu product, sh = |Vh|
                stamp(lane);                          // 3: three quarters of the main k-steps issued
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const float* tv = lds[r].tv[tvw];
                    VhA[r] = *reinterpret_cast<const f32x4*>(&tv[i * RG_TV_STRIDE + gg * 16 + 4 * q]);
                    const float x = tv[i * RG_TV_STRIDE + pperm(a)], y = tv[i * RG_TV_STRIDE + 16 + pperm(a)],
                                z = tv[i * RG_TV_STRIDE + 32 + pperm(a)];
                    SH[r] = sqrtf_(fmaxf(x * x + y * y + z * z, 1e-8f));
                    if constexpr (S::H17) {
                        const float x6 = tv[i * RG_TV_STRIDE + 48], y6 = tv[i * RG_TV_STRIDE + 49], z6 = tv[i * RG_TV_STRIDE + 50];
                        SH16[r] = sqrtf_(fmaxf(x6 * x6 + y6 * y6 + z6 * z6, 1e-8f));
                    }
                }
            }
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
#pragma unroll
                for (int r = 0; r < RG; ++r) vu[acc(r, j)] = mfma_b2<j>(VhA[r][t], w[j], vu[acc(r, j)]);
            });
            if constexpr (t == 3) {
                fold(vu);
                if constexpr (S::H17) {
#pragma unroll
                    for (int r = 0; r < RG; ++r) vu[r] = mfma_b2<0>(Vh16[r], xhq[1], vu[r]);
                }
            }
        } else if constexpr (S::NEXTRA > 0 && qi >= Q.q_rbf && qi < Q.q_rbf + 4 * NH) {
            constexpr int k = qi - Q.q_rbf, half = k % NH, aq = k / NH;
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    if constexpr (half == 0) lo[acc(r, j)] = mfma_b4<4 * aq + j>(R[r], w[j], lo[acc(r, j)]);
                    else hi[acc(r, j)] = mfma_b4<4 * aq + j>(R[r], w[j], hi[acc(r, j)]);
                }
            });
        } else if constexpr (qi >= Q.q_sh && qi < Q.q_sh + 4 * NH) {
            constexpr int k = qi - Q.q_sh, half = k % NH, aq = k / NH;
            if constexpr (k == 0) stamp(lane);        // 4: all main k-steps issued
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    if constexpr (half == 0) lo[acc(r, j)] = mfma_b4<4 * aq + j>(SH[r], w[j], lo[acc(r, j)]);
                    else hi[acc(r, j)] = mfma_b4<4 * aq + j>(SH[r], w[j], hi[acc(r, j)]);
                }
            });
            if constexpr (k == 4 * NH - 1) {          // scalar Linear complete: bias, SiLU, SD -> SA through LDS
                stamp(lane);                          // 5: sh k-steps issued
                fold(lo);
                fold(hi);
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    if constexpr (S::H17) {
                        lo[r] = mfma_b4<0>(SH16[r], xhq[2], lo[r]);
                        if constexpr (NH == 2) hi[r] = mfma_b4<0>(SH16[r], xhq[3], hi[r]);
                    }
                    float* t1 = lds[r].t1[wv.par];
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        slo[r][ii] = siluf_(lo[r][ii] + cq[0]);
                        t1[ii * RG_T1_STRIDE + hcol + lane] = slo[r][ii];
                        if constexpr (NH == 2) {
                            shi[r][ii] = siluf_(hi[r][ii] + cq[1]);
                            t1[ii * RG_T1_STRIDE + 64 + lane] = shi[r][ii];
                        } else shi[r][ii] = 0.f;
                    }
                }
                if constexpr (SPLIT) __syncthreads();  // both halves of the SiLU output are in T1
                else __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const float* t1 = lds[r].t1[wv.par];
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(&t1[i * RG_T1_STRIDE + 8 * a]);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(&t1[i * RG_T1_STRIDE + 8 * a + 4]);
                    const bool on = S::NH == 2 || a < 8;   // 64 outputs: features live in blocks 0..7 only
#pragma unroll
                    for (int m = 0; m < 4; ++m) { X[r][m] = on ? x0[m] : 0.f; X[r][4 + m] = on ? x1[m] : 0.f; }
                }
                stamp(lane);                          // 6: SiLU output back in the SA layout
            }
        }
        if constexpr (qi % RG_SB == RG_SB - 1) __builtin_amdgcn_sched_barrier(0);
    });
    ring.p += QQ.nq * 64;
    if constexpr (SPLIT) wv.par ^= 1;                 // the partner may still be reading this block's T1
#pragma unroll
    for (int r = 0; r < RG; ++r) carry.vu[r] = vu[r];
    carry.bg = cq[2];
}

// end of a chain: the pending gates of its last GVP.  Vd: gated vector output, VD layout (lane 16g+u: channel u,
// coordinate g); Va: the same in the VA layout (when NEEDVA)
template <int RG, int D, bool SIG, bool NEEDVA>
__device__ __forceinline__ void rg_flush(RgRing<D>& ring, const float (&X)[RG][8], float (&Va)[RG][4], f32x4 (&Vd)[RG],
                                         const RgCarry<RG>& carry, RgLds* lds, const int lane, RgStamp& stamp, const int tvw = 0) {
    const int i = lane & 3, g = lane >> 4, q = (lane >> 2) & 3, u = lane & 15;
    const int gg = g < 3 ? g : 2;
    constexpr int NA = RG == 1 ? 2 : 1;
    f32x4 gd[RG * NA];
#pragma unroll
    for (int r = 0; r < RG * NA; ++r) gd[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto acc = [](const int r, const int j) { return RG == 1 ? (j & 1) : r; };
    static_for<0, RG_NQ_FLUSH>([&](auto QI) {
        constexpr int qi = decltype(QI)::value;
        const f32x4 w = ring.q[qi % D];
        ring.q[qi % D] = ring.p[(qi + D) * 64];
        if constexpr (qi < 8) {
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
#pragma unroll
                for (int r = 0; r < RG; ++r) gd[acc(r, j)] = mfma_b2<j>(X[r][qi], w[j], gd[acc(r, j)]);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    ring.p += RG_NQ_FLUSH * 64;
    if constexpr (RG == 1) gd[0] += gd[1];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            float gv = gsum(gd[r][ii]) + carry.bg;
            if constexpr (SIG) gv = sigmoidf_(gv);
            Vd[r][ii] = gv * carry.vu[r][ii];
        }
        if constexpr (NEEDVA) {
            float* tv = lds[r].tv[tvw];
            if (lane < 48) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) tv[ii * RG_TV_STRIDE + g * 16 + pperm(u)] = Vd[r][ii];
            }
        }
    }
    if constexpr (NEEDVA) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(&lds[r].tv[tvw][i * RG_TV_STRIDE + gg * 16 + 4 * q]);
#pragma unroll
            for (int t = 0; t < 4; ++t) Va[r][t] = g < 3 ? v4[t] : 0.f;
        }
    }
    stamp(lane);                                      // chain flushed
}

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

// ---------------------------------------------------------------------------------------------
// Edge messages (gvp.py:472-485, 540-551): a wave = 4*RG consecutive edge slots of one tile of the list.  Slots are
// sorted by destination, so the rows of one destination are consecutive: the wave adds them up in slot order (the
// destination ids are wave-uniform, the test is scalar) and stores one partial row per (wave, destination) run at the
// run's last slot -- what the node kernels read (NodeParams::grp = 4*RG).
// ---------------------------------------------------------------------------------------------
// Workgroup shape: 256 threads = four waves; items are dealt in super-blocks of 1024: wave q of workgroup b takes item
// 1024 (b / 256) + 256 q + b % 256 (two-wave form: 128 threads = one item).  Single-wave workgroups leave the wave -> SIMD placement to the dispatcher's round robin, whose
// state survives kernel boundaries: after a launch of two-wave workgroups the busy waves of the next single-wave launch
// land on half of the SIMDs (layer-0 edge launch of config 2: 21 -> 30 us at unchanged wave cycles).  A four-wave
// workgroup always covers the four SIMDs of its CU, and this assignment puts the first 256 items of a launch on 256
// different workgroups (CUs), the next 256 on their second waves (SIMDs), ... whatever the grid size, which only bounds
// the number of busy items from above -- a launch with fewer busy items than SIMDs gets one busy wave per SIMD, each with as much of a CU's vector memory path as possible (four consecutive
// items per workgroup measured 10 % slower on the 48-item node + head launch).
// Launches with many more items than SIMDs (WAVES = 1) keep single-wave workgroups: a four-wave workgroup holds its
// slots until its slowest wave retires (batch 512: -14 %), and placement history no longer matters there.
// quad_perm broadcast of lane II of every group of four lanes
template <int II>
__device__ __forceinline__ float quad_bcast(const float v) { return dpp_f<II * 0x55>(v); }

// One item of an edge launch: the G = 4 RG slots [e0, e0 + nv) of etype et.
// PRE (conv layer 0, static pp edges; EdgeParams::zs): the first message GVP is not computed -- its SiLU input is
// zs[static slot] + ptab[type of the source] (two row gathers in the SA layout), its Vu is xhat (x) weff -- and the
// chain starts at the second block of the quad stream, with the first GVP's gates pending as usual.
template <bool L0, int RG, int WAVES, bool PRE>
__device__ __forceinline__ void rg_edge_item(const EdgeParams& p, const EncodeParams& ep, RgLds* lds, const int item,
                                             const int wq, const int e0, const int nv, const int et, const int lane) {
    constexpr bool SPLIT = WAVES == 2;
    constexpr int D = RgDepth<RG>::D;
    constexpr int G = 4 * RG;
    static_assert(!(PRE && (SPLIT || !L0)), "the static hoist is a conv-layer-0 path of the one-wave forms");
    RgWave wv;
    wv.half = SPLIT ? wq : 0;
    wv.par = 0;
    RgStamp stamp;
    stamp.id = SPLIT ? 2 * item + wv.half : item;
    stamp(lane);                                       // kernel start
    RgRing<D> ring;
    constexpr int NQ0 = rg_sched(SpecMsg0::VI, SpecMsg0::NEXTRA, 2, false).nq;          // quads of the chain's first block
    ring_start(ring, SPLIT ? p.rgs[et] + (size_t)wv.half * p.rgs_stride : p.rg[et] + (PRE ? (size_t)NQ0 * 256 : 0), lane);   // in flight under the gather
    const int a = lane >> 2, i = lane & 3, g = lane >> 4, q = a & 3, u = lane & 15;
    float X[RG][8], Va[RG][4], R[RG], XH[RG];
    int dstv[RG], srcv[RG];
    f32x4 slo[RG], shi[RG], Vd[RG];
    RgCarry<RG> carry;
    if constexpr (PRE) {
        const float weff = g < 3 ? p.l0c[u] : 0.f;
        carry.bg = p.l0c[16 + u];
        int eo[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int e = e0 + min(4 * r + i, nv - 1);
            srcv[r] = p.esrc[e]; dstv[r] = p.edst[e]; eo[r] = p.eorig[e];
        }
        f32x4 z0[RG], z1[RG];
        int ty[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const float4 xs = p.xn[srcv[r]], xd = p.xn[dstv[r]];
            ty[r] = p.ptype[srcv[r]];
            if (p.ptab_gstride) ty[r] = ty[r] * PF_S + p.l0_gid[srcv[r]] * p.ptab_gstride; else ty[r] *= PF_S;
            const f32x4 PF_AS1* zp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.zs + (size_t)eo[r] * PF_S) + 2 * a;
            z0[r] = zp[0]; z1[r] = zp[1];
            const float dx = xs.x - xd.x, dy = xs.y - xd.y, dz = xs.z - xd.z;
            const float d = sqrtf_(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
            XH[r] = (g == 0 ? dx : (g == 1 ? dy : (g == 2 ? dz : 0.f))) * rcpf_(d);
            R[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const f32x4 PF_AS1* pp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.ptab + ty[r]) + 2 * a;
            const f32x4 p0 = pp[0], p1 = pp[1];
#pragma unroll
            for (int m = 0; m < 4; ++m) { X[r][m] = siluf_(z0[r][m] + p0[m]); X[r][4 + m] = siluf_(z1[r][m] + p1[m]); }
            carry.vu[r] = (f32x4){quad_bcast<0>(XH[r]) * weff, quad_bcast<1>(XH[r]) * weff, quad_bcast<2>(XH[r]) * weff, quad_bcast<3>(XH[r]) * weff};
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) Va[r][tt] = 0.f;
        }
        for (int gi = 1; gi < p.n_gvps; ++gi) rg_gvp<SpecGen, RG, D, 1, false, false>(ring, X, Va, R, XH, slo, shi, carry, lds, lane, stamp, wv);
    } else {
        const float mu_step = (p.rbf_mu[PF_R - 1] - p.rbf_mu[0]) * (1.0f / (float)(PF_R - 1));
        const float mu_a = fmaf((float)a, mu_step, p.rbf_mu[0]);
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int e = e0 + min(4 * r + i, nv - 1);
            const int src = p.esrc[e], dst = p.edst[e];
            dstv[r] = dst; srcv[r] = src;
            const float4 xs = p.xn[src], xd = p.xn[dst];
            const float dx = xs.x - xd.x, dy = xs.y - xd.y, dz = xs.z - xd.z;
            const float d = sqrtf_(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
            const float ze = (d - mu_a) * p.rbf_inv_sigma;
            R[r] = __expf(-(ze * ze));
            XH[r] = (g == 0 ? dx : (g == 1 ? dy : (g == 2 ? dz : 0.f))) * rcpf_(d);
            if (!(L0 && ep.w[0])) {
                const f32x4 PF_AS1* hp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.h + (size_t)src * PF_S) + 2 * a;
                const f32x4 x0 = hp[0], x1 = hp[1];
#pragma unroll
                for (int m = 0; m < 4; ++m) { X[r][m] = x0[m]; X[r][4 + m] = x1[m]; }
            }
            if constexpr (!L0) {
                pf_gcf vp = (pf_gcf)p.v + (size_t)src * 48 + (g < 3 ? g : 0) + 3 * q;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) { const float x = vp[12 * tt]; Va[r][tt] = g < 3 ? x : 0.f; }
            } else {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) Va[r][tt] = 0.f;
            }
        }
        if (L0 && ep.w[0]) rg_encode<RG>(ep, (et == ET_FF || et == ET_FP) ? 1 : 0, srcv, X, lane);   // sources: pharm for ff / fp
        rg_gvp<SpecMsg0, RG, D, 0, L0, SPLIT>(ring, X, Va, R, XH, slo, shi, carry, lds, lane, stamp, wv);
        for (int gi = 1; gi < p.n_gvps; ++gi) rg_gvp<SpecGen, RG, D, 1, false, SPLIT>(ring, X, Va, R, XH, slo, shi, carry, lds, lane, stamp, wv);
    }
    rg_flush<RG, D, true, false>(ring, X, Va, Vd, carry, lds, lane, stamp, wv.half);
    // in-wave segmented sum in slot order; one partial row per (wave, destination) run
    float al = 0.f, ah = 0.f, av = 0.f;
    int prev = -1;
    auto put = [&](const int slot) {                   // two-wave form: wave h holds (and stores) features 64h..64h+63
        p.msg_s[(size_t)slot * PF_S + 64 * wv.half + lane] = al;
        if constexpr (!SPLIT) p.msg_s[(size_t)slot * PF_S + 64 + lane] = ah;
        if (lane < 48 && wv.half == 0) p.msg_v[(size_t)slot * 48 + 3 * u + g] = av;
    };
    static_for<0, G>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if (k < nv) {
            const int dk = __builtin_amdgcn_readlane(dstv[k / 4], k % 4);
            if (k > 0 && dk == prev) {
                al += slo[k / 4][k % 4]; ah += shi[k / 4][k % 4]; av += Vd[k / 4][k % 4];
            } else {
                if (k > 0) put(e0 + k - 1);
                al = slo[k / 4][k % 4]; ah = shi[k / 4][k % 4]; av = Vd[k / 4][k % 4];
            }
            prev = dk;
        }
    });
    put(e0 + nv - 1);
    stamp(lane);                                       // stores issued
}

// RGP: rows-per-wave factor of the static-hoist items (0: that path is not compiled; conv layer 0 only).  With the
// compact work list the "pa" regions (kind 3) are cut into groups of 4 RGP slots and every other region into groups of
// 4 RG; tile lists need RGP == RG.
template <bool L0, int RG, int WAVES, int RGP>
__global__ __launch_bounds__(64 * WAVES) void k_rg_edge(const EdgeParams p, const EncodeParams ep) {
    constexpr int WPB = WAVES == 4 ? 4 : 1;            // waves with their own item per workgroup
    constexpr int RGM = RGP > RG ? RGP : RG;
    __shared__ RgLds lds_all[WPB][RGM];
    constexpr int G = 4 * RG, PER = 32 / G;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    RgLds* lds = lds_all[WPB == 4 ? wq : 0];
    const int item = WPB == 4 ? (int)((blockIdx.x >> 8) * 1024 + wq * 256 + (blockIdx.x & 255)) : (int)blockIdx.x;
    const bool hoist = RGP > 0 && p.zs != nullptr;     // wave-uniform (kernel argument)
    int e0, nv, et;
    bool pre = false;
    if (p.nreg > 0) {
        // Compact work list (launches whose edges all live in per-(etype, graph) regions of run-time length): wave w
        // takes the w-th group of G slots, counting only the groups that hold edges -- the busy waves are the first
        // ones of the grid and spread evenly over the chip, instead of sitting wherever a region's tiles fall.  The
        // region of group w is found by a wave scan over the region lengths (64 regions per pass).
        const int w = item;
        int first = 0, rsel = -1, cnt = 0, start = 0;
        // every pass's lengths and starts are fetched before the first scan: one global round trip whatever the number
        // of regions (up to 64 * RG_CPASS); the scans themselves run on registers
        int cs[RG_CPASS], rs[RG_CPASS];
#pragma unroll
        for (int k = 0; k < RG_CPASS; ++k) {
            const int r = 64 * k + lane;
            cs[k] = r < p.nreg ? p.dyn_cnt[r] : 0;
            rs[k] = r < p.nreg ? p.reg[r] : 0;
        }
        constexpr int SH = RG == 2 ? 3 : 2, SHP = RGP == 2 ? 3 : 2;
        const int pa0 = (hoist && RGP != RG) ? 3 * p.regB : 0x7fffffff;      // first region cut into groups of 4 RGP
        const int abs0 = p.pa_abs ? 3 * p.regB : 0x7fffffff;                 // first region whose groups sit on absolute boundaries
#pragma unroll
        for (int k = 0; k < RG_CPASS; ++k) {
            if (64 * k < p.nreg && rsel < 0) {           // wave-uniform
                const int c = cs[k];
                const int gsh = (64 * k + lane >= pa0) ? SHP : SH;
                const int ng = (64 * k + lane >= abs0) ? (c > 0 ? ((rs[k] + c - 1) >> gsh) - (rs[k] >> gsh) + 1 : 0)
                                                       : (c + (1 << gsh) - 1) >> gsh;
                int incl = ng;
                incl += dpp_i<0x111>(incl); incl += dpp_i<0x112>(incl); incl += dpp_i<0x114>(incl); incl += dpp_i<0x118>(incl);
                incl += dpp_ir<0x142, 0xa>(incl); incl += dpp_ir<0x143, 0xc>(incl);
                incl += first;
                const unsigned long long m = __ballot(incl > w);
                if (m) {
                    const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
                    rsel = 64 * k + l;
                    first = __builtin_amdgcn_readlane(incl - ng, l);
                    cnt = __builtin_amdgcn_readlane(c, l);
                    start = __builtin_amdgcn_readlane(rs[k], l);
                } else first = __builtin_amdgcn_readlane(incl, 63);
            }
        }
        if (rsel < 0) return;                          // wave-uniform: beyond the last group
        const int kind = rsel / p.regB;
        pre = hoist && kind == 3;
        const int Gs = (pre && RGP != RG) ? 4 * RGP : G;
        if (p.pa_abs && kind == 3) {                   // static range: group (w - first) of the absolute grid of Gs slots
            const int lo = ((start / Gs) + (w - first)) * Gs;
            e0 = max(start, lo);
            nv = __builtin_amdgcn_readfirstlane(min(start + cnt, lo + Gs) - e0);
        } else {
            const int loc = (w - first) * Gs;
            e0 = start + loc;
            nv = __builtin_amdgcn_readfirstlane(min(Gs, cnt - loc));
        }
        et = kind == 3 ? (int)ET_PP : kind;            // fourth region kind: pp edges into the active atoms
    } else {
        const int bid = item;
        if (bid >= p.ntiles * PER) return;             // wave-uniform (the grid is rounded up to four items per workgroup)
        const EdgeTile t = p.tiles[bid / PER];
        int nvalid = t.n;
        if (t.cnt_idx >= 0) nvalid = min(nvalid, max(p.dyn_cnt[t.cnt_idx] - t.rel, 0));
        const int base = (bid % PER) * G;
        nv = __builtin_amdgcn_readfirstlane(min(G, nvalid - base));
        if (nv <= 0) return;                           // wave-uniform
        et = __builtin_amdgcn_readfirstlane(t.et);
        e0 = t.e0 + base;
        pre = hoist && RGP == RG && et == ET_PP;
    }
    if constexpr (RGP > 0) {
        if (pre && p.need) {
            // pocket sharing: this group of a representative's static pp edges is computed only if some copy of the
            // pocket reads one of its destinations in this step (its build stamped the atom)
            const int e = e0 + min(lane, nv - 1);
            const bool wanted = p.need[p.edst[e]] == p.need_stamp;
            if (!__any(wanted)) return;                // wave-uniform
        }
        if (pre) { rg_edge_item<L0, RGP, WAVES, true>(p, ep, lds, item, wq, e0, nv, et, lane); return; }
    }
#ifdef PF_TWICE                                       // diagnostic: every item twice through the SAME code (warm instruction cache the second time)
#pragma nounroll
    for (int it = 0; it < 2; ++it) rg_edge_item<L0, RG, WAVES, false>(p, ep, lds, item + 100000 * it, wq, e0, nv, et, lane);
#else
    rg_edge_item<L0, RG, WAVES, false>(p, ep, lds, item, wq, e0, nv, et, lane);
#endif
}

// ---------------------------------------------------------------------------------------------
// Node update (gvp.py:488-536): aggregate the message partial rows (mean per etype or sum, then the norm), residual,
// GVPLayerNorm, update chain, residual, GVPLayerNorm.  A wave = 4*RG nodes of one tile.  HEAD: last conv layer of
// the inference path (pharm tiles): the noise head (dynamics_gvp.py:37-42) runs on the registers right away.
// ---------------------------------------------------------------------------------------------
template <bool L0, int RG, bool HEAD, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_rg_node(const NodeParams p, const HeadParams hp, const EncodeParams ep) {
    constexpr bool SPLIT = WAVES == 2;
    constexpr int D = RgDepth<RG>::D;
    constexpr int WPB = WAVES == 4 ? 4 : 1;            // workgroup shape: see k_rg_edge
    __shared__ RgLds lds_all[WPB][RG];
    constexpr int G = 4 * RG, PER = 32 / G;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    RgLds* lds = lds_all[WPB == 4 ? wq : 0];
    const int item = WPB == 4 ? (int)((blockIdx.x >> 8) * 1024 + wq * 256 + (blockIdx.x & 255)) : (int)blockIdx.x;
    RgWave wv;
    wv.half = SPLIT ? wq : 0;
    wv.par = 0;
    const int bid = item;
    if (bid >= p.ntiles * PER) return;                 // wave-uniform
    const NodeTile t = p.tiles[bid / PER];
    int tn = t.n;
    if (t.cnt_idx >= 0) tn = min(tn, max(p.dyn_cnt[t.cnt_idx] - t.rel, 0));
    const int base = (bid % PER) * G;
    const int nv = __builtin_amdgcn_readfirstlane(min(G, tn - base));
    if (nv <= 0) return;                               // wave-uniform
    const int nt = __builtin_amdgcn_readfirstlane(t.ntype);
    RgStamp stamp;
    stamp.id = SPLIT ? 2 * item + wv.half : item;
    stamp(lane);                                       // kernel start
    RgRing<D> ring;
    ring_start(ring, SPLIT ? p.rgs_upd[nt] + (size_t)wv.half * p.rgs_stride[nt] : p.rg_upd[nt], lane);
    const int a = lane >> 2, i = lane & 3, g = lane >> 4, q = a & 3, u = lane & 15;
    const int gc = g < 3 ? g : 0;
    const NodeW nw = p.w[nt];
    float X[RG][8], Va[RG][4];
    int nid[RG];
    const int gm0 = p.grp - 1, gm1 = (nt == 0 ? p.grp_pa : p.grp) - 1;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const int row = base + min(4 * r + i, nv - 1);
        const int n = t.ids ? p.row_ids[t.n0 + row] : t.n0 + row;
        nid[r] = n;
        float as[8], avv[4];
#pragma unroll
        for (int m = 0; m < 8; ++m) as[m] = 0.f;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) avv[tt] = 0.f;
        // The partial rows of a segment [st, end) are the last slots of the aligned groups of grp slots it touches.  Both
        // segments' descriptors are fetched together, then the first three partial rows of each (absent: the all-zero
        // row) in one batch of loads; longer segments (in-degree beyond ~3 groups) finish in a loop.
        int st[2], cn[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int slot = sl == 0 ? 0 : (nt == 0 ? p.pp_slot : 1);
            st[sl] = p.in_start[slot * p.N + n];
            cn[sl] = p.in_cnt[slot * p.N + n];
        }
        f32x4 x0[2][3], x1[2][3];
        float vv[2][3][4];
        int nxt[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int end = st[sl] + cn[sl];
            const int gm = sl == 0 ? gm0 : gm1;
            int e = st[sl];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool has = e < end;
                const int rw = has ? min(e | gm, end - 1) : p.zero_row;
                const f32x4 PF_AS1* mp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.msg_s + (size_t)rw * PF_S) + 2 * a;
                x0[sl][k] = mp[0]; x1[sl][k] = mp[1];
                pf_gcf vp = (pf_gcf)p.msg_v + (size_t)rw * 48 + gc + 3 * q;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) vv[sl][k][tt] = vp[12 * tt];
                e = has ? rw + 1 : e;
            }
            nxt[sl] = e;
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int end = st[sl] + cn[sl];
            const int gm = sl == 0 ? gm0 : gm1;
            float ps[8], pv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ps[m] = (x0[sl][0][m] + x0[sl][1][m]) + x0[sl][2][m];
                ps[4 + m] = (x1[sl][0][m] + x1[sl][1][m]) + x1[sl][2][m];
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) pv[tt] = (vv[sl][0][tt] + vv[sl][1][tt]) + vv[sl][2][tt];
            for (int e = nxt[sl]; e < end;) {
                const int rw = min(e | gm, end - 1);
                const f32x4 PF_AS1* mp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.msg_s + (size_t)rw * PF_S) + 2 * a;
                const f32x4 y0 = mp[0], y1 = mp[1];
                pf_gcf vp = (pf_gcf)p.msg_v + (size_t)rw * 48 + gc + 3 * q;
                const float v0 = vp[0], v1 = vp[12], v2 = vp[24], v3 = vp[36];
#pragma unroll
                for (int m = 0; m < 4; ++m) { ps[m] += y0[m]; ps[4 + m] += y1[m]; }
                pv[0] += v0; pv[1] += v1; pv[2] += v2; pv[3] += v3;
                e = rw + 1;
            }
            const float sc = (p.norm_mode == 0 && cn[sl] > 0) ? 1.0f / (float)cn[sl] : 1.0f;
#pragma unroll
            for (int m = 0; m < 8; ++m) as[m] = fmaf(ps[m], sc, as[m]);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) avv[tt] = fmaf(pv[tt], sc, avv[tt]);
        }
        float inv_norm = 1.0f;
        if (p.norm_mode == 1) inv_norm = 1.0f / p.norm_value;
        else if (p.norm_mode == 2) inv_norm = 1.0f / p.gnorm[nt * p.B + p.gid[n]];
        if (L0 && ep.w[0]) {                           // residual input encoded below; keep the scaled aggregate
#pragma unroll
            for (int m = 0; m < 8; ++m) X[r][m] = as[m] * inv_norm;
        } else {
            const f32x4 PF_AS1* hpn = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.h_in + (size_t)n * PF_S) + 2 * a;
            const f32x4 h0 = hpn[0], h1 = hpn[1];
#pragma unroll
            for (int m = 0; m < 4; ++m) { X[r][m] = fmaf(as[m], inv_norm, h0[m]); X[r][4 + m] = fmaf(as[4 + m], inv_norm, h1[m]); }
        }
        if constexpr (!L0) {
            pf_gcf vp = (pf_gcf)p.v_in + (size_t)n * 48 + gc + 3 * q;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) Va[r][tt] = fmaf(avv[tt], inv_norm, vp[12 * tt]);
        } else {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) Va[r][tt] = avv[tt] * inv_norm;
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) Va[r][tt] = g < 3 ? Va[r][tt] : 0.f;
    }
    if (L0 && ep.w[0]) {
        float H[RG][8];
        rg_encode<RG>(ep, nt, nid, H, lane);
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int m = 0; m < 8; ++m) X[r][m] += H[r][m];
    }
    stamp(lane);                                       // aggregation done
    rg_layernorm<RG>(nw.ln1_w, nw.ln1_b, X, Va, lane);
    float Xr[RG][8], Vr[RG][4];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
#pragma unroll
        for (int m = 0; m < 8; ++m) Xr[r][m] = X[r][m];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) Vr[r][tt] = Va[r][tt];
    }
    f32x4 slo[RG], shi[RG], Vd[RG];
    const float zero[RG] = {};
    RgCarry<RG> carry;
    rg_gvp<SpecGen, RG, D, 0, false, SPLIT>(ring, X, Va, zero, zero, slo, shi, carry, lds, lane, stamp, wv);
    for (int gi = 1; gi < p.n_upd; ++gi) rg_gvp<SpecGen, RG, D, 1, false, SPLIT>(ring, X, Va, zero, zero, slo, shi, carry, lds, lane, stamp, wv);
    rg_flush<RG, D, true, true>(ring, X, Va, Vd, carry, lds, lane, stamp, wv.half);
#pragma unroll
    for (int r = 0; r < RG; ++r) {
#pragma unroll
        for (int m = 0; m < 8; ++m) X[r][m] += Xr[r][m];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) Va[r][tt] += Vr[r][tt];
    }
    rg_layernorm<RG>(nw.ln2_w, nw.ln2_b, X, Va, lane);
    stamp(lane);                                       // second LayerNorm done
    if constexpr (!HEAD) {
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            if (4 * r + i < nv && wv.half == 0) {      // (both waves of the two-wave form hold the same result)
                f32x4* op = reinterpret_cast<f32x4*>(p.h_out + (size_t)nid[r] * PF_S) + 2 * a;
                op[0] = (f32x4){X[r][0], X[r][1], X[r][2], X[r][3]};
                op[1] = (f32x4){X[r][4], X[r][5], X[r][6], X[r][7]};
                if (g < 3) {
                    float* vp = p.v_out + (size_t)nid[r] * 48 + g + 3 * q;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) vp[12 * tt] = Va[r][tt];
                }
            }
        }
    } else {
        // noise head: its chain and to_scalar_output follow the update chain in the quad stream
        if (hp.n_gvps == 1) rg_gvp<SpecHeadLast, RG, D, 0, false, SPLIT>(ring, X, Va, zero, zero, slo, shi, carry, lds, lane, stamp, wv);
        else {
            rg_gvp<SpecGen, RG, D, 0, false, SPLIT>(ring, X, Va, zero, zero, slo, shi, carry, lds, lane, stamp, wv);
            for (int gi = 1; gi + 1 < hp.n_gvps; ++gi) rg_gvp<SpecGen, RG, D, 1, false, SPLIT>(ring, X, Va, zero, zero, slo, shi, carry, lds, lane, stamp, wv);
            rg_gvp<SpecHeadLast, RG, D, 1, false, SPLIT>(ring, X, Va, zero, zero, slo, shi, carry, lds, lane, stamp, wv);
        }
        rg_flush<RG, D, false, false>(ring, X, Va, Vd, carry, lds, lane, stamp, wv.half);
        // to_scalar_output: Linear(64 -> pharm_nf), K split over the lane groups like the gates
        f32x4 od[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) od[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 oc = {0.f, 0.f, 0.f, 0.f};
        static_for<0, RG_NQ_OUT>([&](auto QI) {
            constexpr int qi = decltype(QI)::value;
            const f32x4 w = ring.q[qi % D];
            ring.q[qi % D] = ring.p[(qi + D) * 64];
            if constexpr (qi == 0) oc = w;
            else if constexpr (qi <= 8) {
                static_for<0, 4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
#pragma unroll
                    for (int r = 0; r < RG; ++r) od[r] = mfma_b2<j>(X[r][qi - 1], w[j], od[r]);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const float o = gsum(od[r][ii]) + oc[0];
                if (4 * r + ii < nv && wv.half == 0) {
                    const int f = __builtin_amdgcn_readlane(nid[r], ii) - hp.node_base;
                    if (g == 0 && u < hp.pharm_nf) hp.eps_h[(size_t)f * hp.pharm_nf + u] = o;
                    if (u == 0 && g < 3) hp.eps_x[(size_t)f * 3 + g] = Vd[r][ii];     // output channel 0, coordinate g
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Static hoist of conv layer 0 (EdgeParams::zs): everything the first message GVP of a pp edge computes from the
// pocket's rigid geometry is constant over a trajectory, and its h_src block sees one of rec_nf encoder outputs per t.
//   k_l0_consts  weff[o] = sum_c Wu[c][o] Wh[0][c]  (Vu = xhat (x) weff when the node vectors are zero), gate bias
//   k_l0_zs      zs[e][f] = b[f] + sum_k W[f][128 + k] rbf_k(d_e) + sum_c W[f][144 + c] sh_c,  sh_c = |Wh[0][c] xhat_e|
//   k_l0_ptab    ptab[row][f] = sum_k W[f][k] LayerNorm(SiLU(W_enc [onehot(type), t] + b_enc))[k]
//   k_l0_types   element type of every protein atom; flag != 0 if a feature row is not a one-hot
// (gvp.py:89-116 with the message inputs of gvp.py:545-547; encoder dynamics_gvp.py:107-117)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_l0_consts(const L0HoistParams p) {
    const int u = threadIdx.x;
    if (u < 16) {
        float w = 0.f;
        for (int c = 0; c < 17; ++c) w = fmaf(p.src[L0H_WU + c * 16 + u], p.src[L0H_WH0 + c], w);
        p.l0c[u] = w;
        p.l0c[16 + u] = p.src[L0H_BG + u];
    }
}
__global__ __launch_bounds__(128) void k_l0_zs(const L0HoistParams p) {
    const int f = threadIdx.x;
    const float b = p.src[L0H_B + f];
    float wr[16], ws[17], wh0[17];
#pragma unroll
    for (int k = 0; k < 16; ++k) wr[k] = p.src[L0H_WR + k * PF_S + f];
#pragma unroll
    for (int c = 0; c < 17; ++c) { ws[c] = p.src[L0H_WSH + c * PF_S + f]; wh0[c] = p.src[L0H_WH0 + c]; }
    for (int e = blockIdx.x; e < p.Epp; e += gridDim.x) {
        const float4 xs = p.xn[p.esrc[e]], xd = p.xn[p.edst[e]];
        const float dx = xs.x - xd.x, dy = xs.y - xd.y, dz = xs.z - xd.z;
        const float d = sqrtf_(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
        const float rd = rcpf_(d), hx = dx * rd, hy = dy * rd, hz = dz * rd;
        float z = b;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float ze = (d - fmaf((float)k, p.rbf_mu_step, p.rbf_mu0)) * p.rbf_inv_sigma;
            z = fmaf(wr[k], __expf(-(ze * ze)), z);
        }
#pragma unroll
        for (int c = 0; c < 17; ++c) {
            const float vx = wh0[c] * hx, vy = wh0[c] * hy, vz = wh0[c] * hz;
            z = fmaf(ws[c], sqrtf_(fmaxf(vx * vx + vy * vy + vz * vz, 1e-8f)), z);
        }
        p.zs[(size_t)e * PF_S + f] = z;
    }
}
__global__ __launch_bounds__(128) void k_l0_ptab(const L0HoistParams p) {
    __shared__ float hs[PF_S];
    __shared__ float red[4];
    const int f = threadIdx.x, row = blockIdx.x;
    const int it = row / p.rec_nf, ty = row % p.rec_nf;
    const float t = p.t_dev ? p.t_dev[it] : p.t_host[it];
    float x = p.enc_b[f];
    x = fmaf(p.enc_w[ty * PF_S + f], 1.0f, x);
    x = fmaf(p.enc_w[p.rec_nf * PF_S + f], t, x);
    x = siluf_(x);
    auto bsum = [&](float v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((f & 63) == 0) red[f >> 6] = v;
        __syncthreads();
        return red[0] + red[1];
    };
    const float mean = bsum(x) * (1.0f / 128.0f);
    const float c = x - mean;
    const float rstd = rsqf_(bsum(c * c) * (1.0f / 128.0f) + 1e-5f);
    hs[f] = c * rstd * p.enc_lw[f] + p.enc_lb[f];
    __syncthreads();
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < PF_S; ++k) acc = fmaf(p.src[L0H_WHT + k * PF_S + f], hs[k], acc);
    p.ptab[(size_t)row * PF_S + f] = acc;
}
__global__ __launch_bounds__(256) void k_l0_types(const L0HoistParams p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    int* eorig = reinterpret_cast<int*>(p.zs);        // [Epp here = Ecap]: identity (the edge build overwrites the "pa" regions)
    for (int e = n; e < p.Epp; e += gridDim.x * 256) eorig[e] = e;
    if (n >= p.Np) return;
    int ty = -1, bad = 0;
    for (int k = 0; k < p.rec_nf; ++k) {
        const float x = p.prot_h0[(size_t)n * p.rec_nf + k];
        if (x == 1.0f) { if (ty >= 0) bad = 1; ty = k; }
        else if (x != 0.0f) bad = 1;
    }
    if (ty < 0) { bad = 1; ty = 0; }
    p.ptype[n] = ty;
    if (bad) atomicOr(p.flag, 1);
}

}  // namespace

extern "C" {
#ifdef PF_STAMPS
int pfk_rg_set_stamp_buffer(unsigned long long* dev) { g_rg_stamp_host_buf = dev; unsigned long long* z = nullptr; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rg_stamps), &z, sizeof(z)); }
int pfk_rg_set_stamp_offset(int off) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rg_stamp_off), &off, sizeof(off)); }
void pfk_rg_set_stamp_which(int which) { g_rg_stamp_which = which; g_rg_stamp_seen = 0; }
#endif
void pfk_l0_hoist(const L0HoistParams* p, int what, hipStream_t s) {       // what: 0 consts + zs, 1 ptab, 2 types
    if (what == 0) {
        hipLaunchKernelGGL(k_l0_consts, dim3(1), dim3(64), 0, s, *p);
        if (p->Epp > 0) hipLaunchKernelGGL(k_l0_zs, dim3(std::min(p->Epp, 8192)), dim3(128), 0, s, *p);
    } else if (what == 1) {
        if (p->nt > 0) hipLaunchKernelGGL(k_l0_ptab, dim3(p->nt * p->rec_nf), dim3(128), 0, s, *p);
    } else hipLaunchKernelGGL(k_l0_types, dim3(std::max(1, (p->Np + 255) / 256)), dim3(256), 0, s, *p);
}
// rows per wave: 8 (RG = 2) once there are enough groups to fill the chip, else 4; rgp: rows-per-wave factor of the
// static-hoist items of conv layer 0 (0: off; tile lists need rgp == rg)
void pfk_rg_edge(const EdgeParams* p, const EncodeParams* enc, int layer0, int rg, int split, int rgp, hipStream_t s) {
    if (p->ntiles == 0) return;
#ifdef PF_STAMPS
    rg_stamp_arm(s);
#endif
    const int per = 32 / (4 * rg);
    const int grid = p->nreg > 0 ? (p->ngroups_sel > 0 ? p->ngroups_sel : (rg == 1 ? p->ngroups4 : p->ngroups8)) : p->ntiles * per;
    if (grid == 0) return;
    const EncodeParams noenc{};
    const EncodeParams& ep = (layer0 && enc) ? *enc : noenc;      // layer 0: encode the gathered rows on the fly
    const int grid4 = grid <= 256 ? grid : (grid + 1023) / 1024 * 256;      // super-blocks of 256 workgroups x 4 waves
    const bool quad = grid <= RG_QUAD_MAX;                                  // latency regime: deterministic SIMD placement
    if (!layer0 || split || !p->zs) rgp = 0;
#define PF_RG_EDGE(L0_, RG_, W_, P_, GRID_) hipLaunchKernelGGL((k_rg_edge<L0_, RG_, W_, P_>), dim3(GRID_), dim3(64 * W_), 0, s, *p, ep)
    if (rg == 1 && split) { if (layer0) PF_RG_EDGE(true, 1, 2, 0, grid); else PF_RG_EDGE(false, 1, 2, 0, grid); }
    else if (!layer0) {
        if (rg == 1 && quad) PF_RG_EDGE(false, 1, 4, 0, grid4);
        else if (rg == 1) PF_RG_EDGE(false, 1, 1, 0, grid);
        else if (quad) PF_RG_EDGE(false, 2, 4, 0, grid4);
        else PF_RG_EDGE(false, 2, 1, 0, grid);
    } else if (rg == 1 && rgp == 2) { if (quad) PF_RG_EDGE(true, 1, 4, 2, grid4); else PF_RG_EDGE(true, 1, 1, 2, grid); }
    else if (rg == 1) { if (quad) PF_RG_EDGE(true, 1, 4, 1, grid4); else PF_RG_EDGE(true, 1, 1, 1, grid); }      // rgp 0 / 1: p->zs decides
    else { if (quad) PF_RG_EDGE(true, 2, 4, 2, grid4); else PF_RG_EDGE(true, 2, 1, 2, grid); }
#undef PF_RG_EDGE
}
void pfk_rg_node(const NodeParams* p, const HeadParams* hp, const EncodeParams* enc, int layer0, int rg, int split, hipStream_t s) {
    if (p->ntiles == 0) return;
#ifdef PF_STAMPS
    rg_stamp_arm(s);
#endif
    const int per = 32 / (4 * rg);
    const HeadParams none{};
    const bool head = hp != nullptr;
    const HeadParams& h = head ? *hp : none;
    const int grid = p->ntiles * per;
    const EncodeParams noenc{};
    const EncodeParams& ep = (layer0 && enc) ? *enc : noenc;
    const int grid4 = grid <= 256 ? grid : (grid + 1023) / 1024 * 256;
    const bool quad = grid <= RG_QUAD_MAX;
#define PF_RG_NODE(L0_, RG_, HEAD_, W_, GRID_) hipLaunchKernelGGL((k_rg_node<L0_, RG_, HEAD_, W_>), dim3(GRID_), dim3(64 * W_), 0, s, *p, h, ep)
#define PF_RG_NODE2(RG_, W_, GRID_)                                                                      \
    do {                                                                                                 \
        if (head) { if (layer0) PF_RG_NODE(true, RG_, true, W_, GRID_); else PF_RG_NODE(false, RG_, true, W_, GRID_); }   \
        else { if (layer0) PF_RG_NODE(true, RG_, false, W_, GRID_); else PF_RG_NODE(false, RG_, false, W_, GRID_); }      \
    } while (0)
    if (rg == 1 && split) PF_RG_NODE2(1, 2, grid);
    else if (rg == 1 && quad) PF_RG_NODE2(1, 4, grid4);
    else if (rg == 1) PF_RG_NODE2(1, 1, grid);
    else if (quad) PF_RG_NODE2(2, 4, grid4);
    else PF_RG_NODE2(2, 1, grid);
#undef PF_RG_NODE2
#undef PF_RG_NODE
}
}
