// pf_n16_core.h -- the device code of the n16 kernels that more than one translation unit runs: the GVP block on 16-row items
// (n16_block), the weight ring, the center encoder, an edge item with its chain and segmented sums (n16_edge_item).  pf_n16.hip has
// the kernels; pf_rg.hip runs conv layer 0's "pa" items of the NEXT call as workgroups of a step's merged last launch (BuildParams::
// pa_same).  The file comment of pf_n16.hip describes the mapping onto the matrix cores.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include "pf_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace pfn16 {

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

__device__ __forceinline__ f32x4 mfma16(const float a, const float b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// x[l] + x[l ^ 32], x[l] + x[l ^ 16] (see pf_rg.hip: the compiler's permlane-swap builtins mis-assign their second result)
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
// sum over the four lane groups g (lanes l, l ^ 16, l ^ 32, l ^ 48): every lane of a row ends with the row's sum
__device__ __forceinline__ float gsum(const float v) { return xsum16(xsum32(v)); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(const int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_ir(const int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
// value of lane L in every lane (the element is copied out first: __builtin_bit_cast applied directly to an element of an
// ext_vector_type read element 0 with this compiler)
__device__ __forceinline__ float lane_bcast(const float v, const int L) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), L));
}
// orders LDS only: __syncthreads() would also drain the weight ring's outstanding global loads
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- -DN16_SPLIT=1 (pf_device.h): the scalar activations of a block as three bf16 planes, this lane's B operands of the four K chunks
#if N16_SPLIT
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct N16P { u32x4 b[3][4]; };                      // [plane][chunk]: 8 bf16, element e <-> XS[8 chunk + e]
__device__ __forceinline__ unsigned bf16_rne(const float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// x = p0 + p1 + p2 to 24 bits: each plane the round-to-nearest-even bf16 of what the planes before it left over (exact subtractions)
__device__ __forceinline__ void n16_split8(const float (&x)[8], u32x4 (&o)[3]) {
    unsigned q[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        q[0][e] = bf16_rne(x[e]);
        const float r1 = x[e] - __uint_as_float(q[0][e] << 16);
        q[1][e] = bf16_rne(r1);
        q[2][e] = bf16_rne(r1 - __uint_as_float(q[1][e] << 16));
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int d = 0; d < 4; ++d) o[p][d] = q[p][2 * d] | (q[p][2 * d + 1] << 16);
}
__device__ __forceinline__ void n16_split_rows(const float (&XS)[32], N16P& P) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x[8];
        u32x4 o[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = XS[8 * c + e];
        n16_split8(x, o);
#pragma unroll
        for (int p = 0; p < 3; ++p) P.b[p][c] = o[p];
    }
}
#else
struct N16P {};
__device__ __forceinline__ void n16_split_rows(const float (&)[32], N16P&) {}
#endif

// register prefetch ring over a wave's quad stream (the stream is read through a buffer descriptor: scalar base and
// stream position, the lane's 16 bytes in one vector register -- no per-quad vector address arithmetic)
struct N16Ring {
    f32x4 q[N16_D];
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned loff;                            // 16 * lane
    unsigned blk;                             // byte offset of quad 0 of the current block (wave-uniform)
    __device__ __forceinline__ f32x4 load(const int qi) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, loff, blk + (unsigned)qi * 1024u, 0));
    }
    __device__ __forceinline__ void advance(const int nq) { blk += (unsigned)nq * 1024u; }
};
__device__ __forceinline__ void ring_start(N16Ring& r, pf_gcf stream, const int lane) {
    const unsigned long long a = (unsigned long long)stream;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    r.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
    r.loff = 16u * (unsigned)lane;
    r.blk = 0;
    static_for<0, N16_D>([&](auto I) { r.q[decltype(I)::value] = r.load(decltype(I)::value); });
}

// In-kernel cycle stamps (diagnostic builds only, -DN16_STAMPS: tools/probes/n16_chain_bench.hip, tools/n16_stamps.py): lane 0 of
// every wave of 64 workgroups writes s_memtime at the phase boundaries of n16_block.  The stamp counter `sk` carries the
// kernel's id in bits 8.. (k_n16_edge<true> 0, <false> 1, k_n16_fused 2, k_n16_tail 3, k_n16_unit 4): g_n16_stamp_kid >= 0
// records that kernel only (a step runs three of them over the same buffer)
#if defined(N16_STAMPS) && defined(N16_STAMP_GLOBALS)     // (pf_n16.hip declares the stamp buffer's symbols in front of this header; pf_rg.hip does not stamp)
#define N16_STAMP(sk, lane, wq)                                                                                        \
    do {                                                                                                               \
        const int rb_ = (int)blockIdx.x - g_n16_stamp_off;                                                             \
        const int k_ = (sk) & 255;                                                                                     \
        if ((lane) == 0 && g_n16_stamps && rb_ >= 0 && rb_ < 64 && k_ < 64 && (g_n16_stamp_kid < 0 || g_n16_stamp_kid == ((sk) >> 8))) \
            g_n16_stamps[((size_t)rb_ * 4 + (wq)) * 64 + k_] = __builtin_amdgcn_s_memtime();                           \
        ++(sk);                                                                                                        \
    } while (0)
#else
#define N16_STAMP(sk, lane, wq) do { } while (0)
#endif
// the five stamps inside every n16_block (-DN16_STAMPS_SPARSE: off -- each stamp waits for the wave's outstanding LDS
// operations, and five per block stretch a chain by ~50 %; the sparse form keeps the phase boundaries outside the blocks)
#if defined(N16_STAMPS) && defined(N16_STAMP_GLOBALS) && !defined(N16_STAMPS_SPARSE)
#define N16_STAMP_B(sk, lane, wq) N16_STAMP(sk, lane, wq)
#else
#define N16_STAMP_B(sk, lane, wq) do { } while (0)
#endif

// diagnostic builds (-DFUSED_CUT=k / -DTAIL_CUT=k): the workgroup stops at cut point k (timing only).  The value that
// reached the cut point is kept alive by a store that never executes.
#define N16_CUT_AT(which, k, val, ptr)                                                                                 \
    do { if ((which) == (k)) { if ((val) == 1.2345e-33f) *(ptr) = (val); __builtin_amdgcn_endpgm(); } } while (0)
#ifndef FUSED_CUT
#define FUSED_CUT 0
#endif
#ifndef EDGE_CUT
#define EDGE_CUT 0
#endif
#ifndef TAIL_CUT
#define TAIL_CUT 0
#endif

// -DN16_TRACE (diagnostic builds): wave 0 of every workgroup of k_n16_edge records [start, end, HW_ID, XCC_ID | item kind]
#ifdef N16_TRACE
__device__ unsigned long long* g_n16_trace = nullptr;
#endif

// LDS of one item (one workgroup): every buffer has a barrier between a read and the next write (see n16_block)
struct __attribute__((aligned(16))) N16Lds {
    float s[8 * 64 * 4];                      // SiLU outputs: [tile T = 2 w + t][lane][r] = feature 16 T + 4 g + r of row j
    float v[3 * 64 * 4];                      // Vh per coordinate: [c][lane][r] = hidden channel 4 g + r of row j
    float g[4 * 64 * 4];                      // K-split partial sums of the gate Linear per wave: [w][lane][r] = channel 4 g + r
    float v16[3 * 16];                        // first message GVP (17 hidden channels): Vh[16] per coordinate and row
    float vn[3 * 64 * 4];                     // GVPLayerNorm: squared vector components per coordinate (n16_layernorm)
    float vx[4 * 16 * 48];                    // node updates: the first four vector partial rows of the 16 nodes, one per wave (n16_rows_sum)
    float ln[4 * 128];                        // node updates: the two LayerNorms' weight / bias rows, requested when the item starts (n16_ln_stage)
    float eln[2 * 128];                       // n16_encode_pharm: the encoder LayerNorm's weight / bias rows (requested with the encoder's inputs)
#if N16_SPLIT
    unsigned pb[3 * 4 * 64 * 4];              // SiLU outputs as bf16 planes: [plane][chunk = producing wave][lane][4 dwords = 8 bf16]
#endif
};

// what the first message GVP of an edge needs besides the source row
struct N16In {
    float rb[4];                              // rbf image: lane 16 g + j holds rbf_{4 g + r}(d_j)
    float xh[3];                              // unit x_diff of row j (every lane of the row)
};

// ---------------------------------------------------------------------------------------------
// One GVP (gvp.py:89-116) on the 16 rows of an item, on the four waves of the workgroup (wq = wave, 0..3).
//   XS [32]  in : scalar input as B operands (k-step ks: feature 16 (ks >> 2) + 4 g + (ks & 3) of row j); unused by M0H
//            out: the SiLU output in the same form (!LAST)
//   VB [4]   in : vector input of coordinate wq as B operands (register r: channel 4 g + r of row j; 0 on wave 3);
//                 unused by the kinds whose node vectors are zero
//            out: the gated output vectors of coordinate wq (waves 0..2)
//   S  [2]   in : M0H only: the pre-activation so far (type-table row of the source, bias folded in), D layout
//            out: the SiLU output of this wave's 32 features (tile t, register r: feature 32 wq + 16 t + 4 g + r)
//   OFF      ring phase: quad qi of the block sits in ring slot (OFF + qi) % N16_D
//   LAST     last GVP of the chain: the SiLU output is not exchanged (XS is not written)
// Barriers: A (Vh of the three coordinates -> sh; only the kinds with a vector input) and B (SiLU outputs, gate sums).
// ---------------------------------------------------------------------------------------------
//   SIG      the vector gate's activation: sigmoid (every GVP but the noise head's last one: identity, dynamics_gvp.py:20)
//   PEND     the previous block of the chain left its vector gate PENDING: VB holds its ungated Vu (waves 0..2), its gate sums wait
//            in lds->g (written in front of its barrier B).  This block requests the sums under its first main k-steps and forms
//            sigmoid(sum) x Vu in front of its vh quad, N16_VH_AT main quads into the block -- not in front of the first one.
//            !PEND: VB is the vector input as it stands (first block of a chain).
//   A block that is not LAST leaves its own gate pending; the chain's caller completes the last one (n16_gate_flush) or runs a
//   LAST block, which completes its gate itself.
//   gs       optional: receives the gate pre-activations (the K-split sums + bias; vector waves) -- the tail kernel packs
//            to_scalar_output into the unused gate rows of the head's last GVP and reads eps_h from here
__device__ __forceinline__ void n16_gate_request(f32x4 (&G)[4], const N16Lds* lds, const int lane) {
#pragma unroll
    for (int w = 0; w < 4; ++w) G[w] = *reinterpret_cast<const f32x4*>(&lds->g[(w * 64 + lane) * 4]);
}
template <bool SIG = true>
__device__ __forceinline__ void n16_gate_apply(const f32x4 (&G)[4], float (&VB)[4], const bool vecw, f32x4* gs = nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float gpre = (G[0][r] + G[1][r]) + (G[2][r] + G[3][r]);
        if (gs) (*gs)[r] = gpre;
        VB[r] = vecw ? (SIG ? sigmoidf_(gpre) : gpre) * VB[r] : 0.f;
    }
}
// completes the gate a chain's last non-LAST block left pending (all four waves; wave 3 ends with VB = 0)
__device__ __forceinline__ void n16_gate_flush(float (&VB)[4], const N16Lds* lds, const int lane, const int wq) {
    f32x4 G[4];
    n16_gate_request(G, lds, lane);
    n16_gate_apply<true>(G, VB, wq < 3);
}

//   OUTF32   (-DN16_SPLIT builds; !LAST) the SiLU outputs leave as fp32 rows in XS -- what a caller that is not a block reads (residual,
//            LayerNorm); false: as bf16 planes in XP, what the next block's main k-steps read.  Without N16_SPLIT: always XS.
template <int KIND, int OFF, bool LAST, bool SIG = true, bool PEND = false, bool OUTF32 = true>
__device__ __forceinline__ void n16_block(N16Ring& ring, float (&XS)[32], N16P& XP, float (&VB)[4], const N16In& in, f32x4 (&S)[2],
                                          N16Lds* lds, const int lane, const int wq, int& sk, f32x4* gs = nullptr) {
    constexpr bool PLANES_OUT = N16_SPLIT && !OUTF32 && !LAST;
    constexpr N16Sched Q = n16_sched(KIND);
    static_assert(!PEND || KIND == N16_GEN, "only GEN blocks follow another block of a chain");
    N16_STAMP_B(sk, lane, wq);                                          // block start
    constexpr bool M0 = KIND != N16_GEN;                              // 17 hidden vector channels, rbf inputs
    constexpr bool VZ = KIND == N16_M0Z || KIND == N16_M0H;           // the node vectors are zero: Vh = Wh[0] (x) xhat
    constexpr bool HOIST = KIND == N16_M0H;
    // where the pending gate's sums are requested (two quads in front of the vh quad) and where barrier A and the reads of Vh sit
    // (main quad 11: every wave wrote its Vh long before, sh is formed under the last main k-steps instead of behind them)
    constexpr int Q_GREQ = PEND ? (Q.q_vh >= 2 ? Q.q_vh - 2 : 0) : -1;
    constexpr int M_A = N16_SPLIT ? 16 : 11, M_SH = N16_SPLIT ? 21 : 14;
    const int g = lane >> 4;
    const bool vecw = wq < 3;                                         // wave-uniform
    const bool g0 = g == 0;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, vh = acc0, vu = acc0, gp = acc0;
    f32x4 x1 = acc0, b0 = acc0, gbias = acc0, sh = acc0;
#if N16_SPLIT
    f32x4 sacc[2][3];
#pragma unroll
    for (int t_ = 0; t_ < 2; ++t_)
#pragma unroll
        for (int p_ = 0; p_ < 3; ++p_) sacc[t_][p_] = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
    f32x4 G[4], va = acc0, vb = acc0, vc = acc0;
    G[0] = acc0; G[1] = acc0; G[2] = acc0; G[3] = acc0;
    if constexpr (HOIST) { acc0 = S[0]; acc1 = S[1]; }
    float vh16 = 0.f, sh16 = 0.f, x6 = 0.f, y6 = 0.f, z6 = 0.f;
    const float xhc = M0 ? (wq == 0 ? in.xh[0] : (wq == 1 ? in.xh[1] : (wq == 2 ? in.xh[2] : 0.f))) : 0.f;
    static_for<0, Q.nq>([&](auto QI) {
        constexpr int qi = decltype(QI)::value;
        constexpr int mq = Q.main_of(qi);                               // main quad number, or -1
        const f32x4 w = ring.q[(OFF + qi) % N16_D];
#ifndef N16_PROBE_NOLOAD                              // (diagnostic builds: the ring is never refilled -- what a block costs without its weight stream)
        ring.q[(OFF + qi) % N16_D] = ring.load(qi + N16_D);
#endif
        if constexpr (PEND && qi == Q_GREQ) n16_gate_request(G, lds, lane);
        if constexpr (qi == Q.q_x1) {
            x1 = w;                                   // [sh16 column tile 0, tile 1, Wu[16][i], Wh[0][i] (lane 16: Wh[0][16])]
        } else if constexpr (qi == Q.q_vh) {
            if constexpr (PEND) n16_gate_apply<true>(G, VB, vecw);
            if constexpr (VZ) {
                // Vh[h][c] = Wh[0][h] xhat_c: this wave's coordinate for Vu, all three for sh (no exchange)
                const float w016 = lane_bcast(x1[3], 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vh[r] = w[r] * xhc;
                    const float a = w[r] * in.xh[0], b = w[r] * in.xh[1], c = w[r] * in.xh[2];
                    sh[r] = sqrtf_(fmaxf(a * a + b * b + c * c, 1e-8f));
                }
                vh16 = w016 * xhc;
                const float a = w016 * in.xh[0], b = w016 * in.xh[1], c = w016 * in.xh[2];
                sh16 = sqrtf_(fmaxf(a * a + b * b + c * c, 1e-8f));
            } else if (vecw) {
#pragma unroll
                for (int r = 0; r < 4; ++r) vh = mfma16(w[r], VB[r], vh);
                if constexpr (M0) vh = mfma16(g0 ? x1[3] : 0.f, g0 ? xhc : 0.f, vh);      // the unit x_diff channel
                if constexpr (!M0) *reinterpret_cast<f32x4*>(&lds->v[(wq * 64 + lane) * 4]) = vh;
            }
        } else if constexpr (qi == Q.q_w16) {         // hidden channel 16 of Vh on the vector ALU: [Wh[v0 + 4 g + r][16]]
            if (vecw) {
                const float w016 = lane_bcast(x1[3], 16);
                float pv = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) pv = fmaf(w[r], VB[r], pv);
                vh16 = fmaf(w016, xhc, gsum(pv));
                *reinterpret_cast<f32x4*>(&lds->v[(wq * 64 + lane) * 4]) = vh;
                if (g0) lds->v16[wq * 16 + (lane & 15)] = vh16;
            }
        } else if constexpr (mq >= 0) {
#if N16_SPLIT
            // quad mq = 8 bf16 of plane p of the weights (output tile t, K chunk c): it meets planes 0 .. 2 - p of the activations
            // (one accumulator per activation plane and tile: a 4-pass instruction that waits for the one issued just before it stalls
            // the wave -- the six of a (chunk, tile) on one accumulator ran no faster than the 8-pass fp32 instructions they replace)
            constexpr int c = mq / 6, t = (mq / 3) % 2, p = mq % 3;
            const bf16x8 a = __builtin_bit_cast(bf16x8, w);
#pragma unroll
            for (int pb = 0; pb + p < 3; ++pb)
                sacc[t][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, XP.b[pb][c]), sacc[t][pb], 0, 0, 0);
#else
            constexpr int ks = 2 * mq;
            acc0 = mfma16(w[0], XS[ks], acc0);
            acc1 = mfma16(w[1], XS[ks], acc1);
            acc0 = mfma16(w[2], XS[ks + 1], acc0);
            acc1 = mfma16(w[3], XS[ks + 1], acc1);
#endif
            if constexpr (!VZ && mq == M_A) {         // barrier A: the three coordinates of Vh are in LDS; their reads travel under the next k-steps
                N16_STAMP_B(sk, lane, wq);              // main k-steps (mostly) issued
                lds_barrier();
                N16_STAMP_B(sk, lane, wq);              // barrier A passed
                va = *reinterpret_cast<const f32x4*>(&lds->v[(0 * 64 + lane) * 4]);
                vb = *reinterpret_cast<const f32x4*>(&lds->v[(1 * 64 + lane) * 4]);
                vc = *reinterpret_cast<const f32x4*>(&lds->v[(2 * 64 + lane) * 4]);
                if constexpr (M0) { x6 = lds->v16[lane & 15]; y6 = lds->v16[16 + (lane & 15)]; z6 = lds->v16[32 + (lane & 15)]; }
            }
            if constexpr (!VZ && mq == M_SH) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sh[r] = sqrtf_(fmaxf(va[r] * va[r] + vb[r] * vb[r] + vc[r] * vc[r], 1e-8f));
                if constexpr (M0) sh16 = sqrtf_(fmaxf(x6 * x6 + y6 * y6 + z6 * z6, 1e-8f));
            }
        } else if constexpr (M0 && qi >= Q.q_rbf && qi < Q.q_rbf + 2) {
            constexpr int r0 = 2 * (qi - Q.q_rbf);
            acc0 = mfma16(w[0], in.rb[r0], acc0);
            acc1 = mfma16(w[1], in.rb[r0], acc1);
            acc0 = mfma16(w[2], in.rb[r0 + 1], acc0);
            acc1 = mfma16(w[3], in.rb[r0 + 1], acc1);
        } else if constexpr (qi >= Q.q_sh && qi < Q.q_sh + 2) {
            constexpr int r0 = 2 * (qi - Q.q_sh);
            acc0 = mfma16(w[0], sh[r0], acc0);
            acc1 = mfma16(w[1], sh[r0], acc1);
            acc0 = mfma16(w[2], sh[r0 + 1], acc0);
            acc1 = mfma16(w[3], sh[r0 + 1], acc1);
            if constexpr (M0 && r0 == 2) {            // hidden channel 16: one more k-step, k = 0 only
                acc0 = mfma16(g0 ? x1[0] : 0.f, g0 ? sh16 : 0.f, acc0);
                acc1 = mfma16(g0 ? x1[1] : 0.f, g0 ? sh16 : 0.f, acc1);
            }
        } else if constexpr (qi == Q.q_vu) {
            if (vecw) {
#pragma unroll
                for (int r = 0; r < 4; ++r) vu = mfma16(w[r], vh[r], vu);
                if constexpr (M0) vu = mfma16(g0 ? x1[2] : 0.f, g0 ? vh16 : 0.f, vu);
            } else gbias = w;                          // wave 3's stream carries the gate bias here
            if constexpr (HOIST) { S[0] = acc0; S[1] = acc1; }
        } else if constexpr (Q.q_b >= 0 && qi == Q.q_b) {
            b0 = w;
        } else if constexpr (Q.q_b >= 0 && qi == Q.q_b + 1) {
#if N16_SPLIT
            S[0] = ((sacc[0][0] + sacc[0][1]) + (sacc[0][2] + acc0)) + b0;
            S[1] = ((sacc[1][0] + sacc[1][1]) + (sacc[1][2] + acc1)) + w;
#else
            S[0] = acc0 + b0;
            S[1] = acc1 + w;
#endif
        } else if constexpr (qi >= Q.q_gate && qi < Q.q_gate + 2) {
            constexpr int t = qi - Q.q_gate;
            if constexpr (t == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { S[0][r] = siluf_(S[0][r]); S[1][r] = siluf_(S[1][r]); }
                if constexpr (PLANES_OUT) {           // the producer splits its own eight outputs: chunk wq of the next block's K
#if N16_SPLIT
                    float x8[8];
                    u32x4 o3[3];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { x8[r] = S[0][r]; x8[4 + r] = S[1][r]; }
                    n16_split8(x8, o3);
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<u32x4*>(&lds->pb[((pp * 4 + wq) * 64 + lane) * 4]) = o3[pp];
#endif
                } else if constexpr (!LAST) {         // the SiLU outputs leave for LDS under the gate k-steps
                    *reinterpret_cast<f32x4*>(&lds->s[((2 * wq) * 64 + lane) * 4]) = S[0];
                    *reinterpret_cast<f32x4*>(&lds->s[((2 * wq + 1) * 64 + lane) * 4]) = S[1];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gp = mfma16(w[r], S[t][r], gp);
            if constexpr (t == 1) {                   // publish; barrier B; collect
                if (!vecw) gp += gbias;
                *reinterpret_cast<f32x4*>(&lds->g[(wq * 64 + lane) * 4]) = gp;
                N16_STAMP_B(sk, lane, wq);              // sh / Vu / SiLU / gate k-steps issued
                lds_barrier();
                N16_STAMP_B(sk, lane, wq);              // barrier B passed
                if constexpr (!LAST) {
                    if constexpr (PLANES_OUT) {
#if N16_SPLIT
#pragma unroll
                        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) XP.b[pp][cc] = *reinterpret_cast<const u32x4*>(&lds->pb[((pp * 4 + cc) * 64 + lane) * 4]);
#endif
                    } else {
#pragma unroll
                        for (int T = 0; T < 8; ++T) {
                            const f32x4 x = *reinterpret_cast<const f32x4*>(&lds->s[(T * 64 + lane) * 4]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) XS[4 * T + r] = x[r];
                        }
                    }
                    // the gate stays pending: VB = the ungated Vu, the sums wait in lds->g (next block, or n16_gate_flush)
#pragma unroll
                    for (int r = 0; r < 4; ++r) VB[r] = vecw ? vu[r] : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) VB[r] = vu[r];
                    f32x4 GL[4];
                    n16_gate_request(GL, lds, lane);
                    n16_gate_apply<SIG>(GL, VB, vecw, gs);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    ring.advance(Q.nq);
}

// n GEN blocks in a row, none of them LAST, the first one taking VB as it stands: leaves the last one's gate pending (n > 0)
// OUTF32: the run's last block leaves fp32 rows in XS (its reader is not a block); false: bf16 planes in XP (N16_SPLIT builds).
// The run takes its input from XS and splits it itself.
template <int OFF, bool OUTF32 = true>
__device__ __forceinline__ void n16_gen_run(const int n, N16Ring& ring, float (&XS)[32], N16P& XP, float (&VB)[4], const N16In& in,
                                            f32x4 (&S)[2], N16Lds* lds, const int lane, const int wq, int& sk) {
    if (n <= 0) return;
    n16_split_rows(XS, XP);
    if (n == 1) { n16_block<N16_GEN, OFF, false, true, false, OUTF32>(ring, XS, XP, VB, in, S, lds, lane, wq, sk); return; }
    n16_block<N16_GEN, OFF, false, true, false, false>(ring, XS, XP, VB, in, S, lds, lane, wq, sk);
    for (int gi = 1; gi + 1 < n; ++gi) n16_block<N16_GEN, OFF, false, true, true, false>(ring, XS, XP, VB, in, S, lds, lane, wq, sk);
    n16_block<N16_GEN, OFF, false, true, true, OUTF32>(ring, XS, XP, VB, in, S, lds, lane, wq, sk);
}

// segmented inclusive scan over the 16 rows of an item (one 16-lane DPP row per lane group): rows are sorted by key
struct SegMask16 { float m1, m2, m4, m8; };
__device__ __forceinline__ SegMask16 seg_masks16(const int key, const int j) {
    const int k1 = dpp_i<0x111>(key), k2 = dpp_i<0x112>(key), k4 = dpp_i<0x114>(key), k8 = dpp_i<0x118>(key);
    SegMask16 m;
    m.m1 = ((j >= 1) & (k1 == key)) ? 1.f : 0.f;
    m.m2 = ((j >= 2) & (k2 == key)) ? 1.f : 0.f;
    m.m4 = ((j >= 4) & (k4 == key)) ? 1.f : 0.f;
    m.m8 = ((j >= 8) & (k8 == key)) ? 1.f : 0.f;
    return m;
}
__device__ __forceinline__ float seg_scan16(float v, const SegMask16& m) {
    v = fmaf(dpp_f<0x111>(v), m.m1, v);
    v = fmaf(dpp_f<0x112>(v), m.m2, v);
    v = fmaf(dpp_f<0x114>(v), m.m4, v);
    v = fmaf(dpp_f<0x118>(v), m.m8, v);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Edge messages (gvp.py:472-485, 540-551): one item = the 16 edge slots [e0, e0 + nv) of etype et.  Slots are sorted by
// destination: the rows of one destination are consecutive lanes of a 16-lane row, a segmented scan leaves each
// (item, destination) run's sum in its last lane, and only those lanes store a partial row (at their own slot) -- what
// the node kernels read with NodeParams::grp = 16.
// ---------------------------------------------------------------------------------------------
// Scalar encoder of the pharmacophore centers on the fly (dynamics_gvp.py:107-117, 143-151: h = LayerNorm(SiLU(W [h_t, t] + b)))
// for the 16 source rows of a conv-layer-0 item: wave w encodes features [32 w, 32 w + 32) (SiLU is the expensive part),
// the slices meet in LDS, and every wave normalises the features it holds as B operands (two-pass statistics over the
// row: its 32 registers and the four lane groups).  in: the features of the row's center; tt: its graph's timestep.
__device__ __forceinline__ void n16_encode_pharm(const EncodeParams& ep, pf_gcf in, const float tt, float (&XS)[32], N16Lds* lds,
                                                 const int lane, const int wq) {
    const int g = lane >> 4;
    const int nf = ep.pharm_nf;
    pf_gcf Wt = (pf_gcf)ep.w[1] + 32 * wq + 4 * g;                    // [nf + 1][128], input-major
    f32x4 z0 = *reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)ep.b[1] + 32 * wq + 4 * g);
    f32x4 z1 = *reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)ep.b[1] + 32 * wq + 16 + 4 * g);
    // the LayerNorm's parameters leave with the encoder's inputs (one float per thread) and wait in LDS: loaded behind the
    // barrier below, where they are used, they were a cold round trip of their own in every item that encodes centers
    const int tl = wq * 64 + lane;
    const float lnq = (tl < 128 ? (pf_gcf)ep.ln_w[1] : (pf_gcf)ep.ln_b[1])[tl & 127];
    // the encoder's inputs and weight rows are requested eight at a time with clamped indices (one round trip per batch: a
    // loop over the run-time input count would wait for every row in turn)
    for (int k0 = 0; k0 <= nf; k0 += 8) {
        float x[8];
        f32x4 w0[8], w1[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = min(k0 + kk, nf);
            x[kk] = in[min(k, nf - 1)];
            w0[kk] = *reinterpret_cast<const f32x4 PF_AS1*>(Wt + k * PF_S);
            w1[kk] = *reinterpret_cast<const f32x4 PF_AS1*>(Wt + k * PF_S + 16);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = k0 + kk;
            const float xv = k < nf ? x[kk] : (k == nf ? tt : 0.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) { z0[r] = fmaf(w0[kk][r], xv, z0[r]); z1[r] = fmaf(w1[kk][r], xv, z1[r]); }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { z0[r] = siluf_(z0[r]); z1[r] = siluf_(z1[r]); }
    *reinterpret_cast<f32x4*>(&lds->s[((2 * wq) * 64 + lane) * 4]) = z0;
    *reinterpret_cast<f32x4*>(&lds->s[((2 * wq + 1) * 64 + lane) * 4]) = z1;
    lds->eln[tl] = lnq;
    lds_barrier();
    float sum = 0.f;
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&lds->s[(T * 64 + lane) * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) { XS[4 * T + r] = x[r]; sum += x[r]; }
    }
    const float mean = gsum(sum) * (1.0f / 128.0f);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) { const float c = XS[k] - mean; var = fmaf(c, c, var); }
    const float rstd = rsqf_(gsum(var) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        const f32x4 lw = *reinterpret_cast<const f32x4*>(&lds->eln[16 * T + 4 * g]);
        const f32x4 lb = *reinterpret_cast<const f32x4*>(&lds->eln[128 + 16 * T + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) XS[4 * T + r] = (XS[4 * T + r] - mean) * rstd * lw[r] + lb[r];
    }
    lds_barrier();                                       // the slices are read: the chain's first block may write lds->s
}

// what an item's prologue hands to its chain: this lane's row (edge slot, destination; rows beyond nv repeat the last one)
// and the coordinates of its end points
struct N16Rows {
    int e, dst;
    float4 xs, xd;
};

// chain + per-destination sums of one item whose first-block inputs are in registers (XS / VB / S as n16_block takes them)
template <int KIND0>
__device__ __forceinline__ void n16_edge_chain(const EdgeParams& p, N16Ring& ring, const N16Rows& rw, float (&XS)[32], float (&VB)[4],
                                               f32x4 (&S)[2], N16Lds* lds, const int nv, const int lane, const int wq, int& sk) {
    constexpr int OFF1 = n16_sched(KIND0).nq % N16_D;
    const int g = lane >> 4, j = lane & 15;
    N16In in;
    {
        const float dx = rw.xs.x - rw.xd.x, dy = rw.xs.y - rw.xd.y, dz = rw.xs.z - rw.xd.z;
        const float d = sqrtf_(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
        const float rd = rcpf_(d);
        in.xh[0] = dx * rd; in.xh[1] = dy * rd; in.xh[2] = dz * rd;
        const float mu_step = (p.rbf_mu[PF_R - 1] - p.rbf_mu[0]) * (1.0f / (float)(PF_R - 1));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ze = (d - fmaf((float)(4 * g + r), mu_step, p.rbf_mu[0])) * p.rbf_inv_sigma;
            in.rb[r] = __expf(-(ze * ze));
        }
    }
    N16_STAMP(sk, lane, wq);                              // source rows gathered / encoded (as far as the compiler keeps the order)
    N16P XP;
    if constexpr (KIND0 != N16_M0H) n16_split_rows(XS, XP);
    n16_block<KIND0, 0, false, true, false, false>(ring, XS, XP, VB, in, S, lds, lane, wq, sk);
    for (int gi = 1; gi + 1 < p.n_gvps; ++gi) n16_block<N16_GEN, OFF1, false, true, true, false>(ring, XS, XP, VB, in, S, lds, lane, wq, sk);
    n16_block<N16_GEN, OFF1, true, true, true>(ring, XS, XP, VB, in, S, lds, lane, wq, sk);
    N16_STAMP(sk, lane, wq);                              // chain done
    // per-destination sums in slot order, one partial row per (item, destination) run
    const SegMask16 sm = seg_masks16(rw.dst, j);
    const int dnext = dpp_i<0x101>(rw.dst);              // row_shl 1: the next row's destination
    const bool tail = (j == nv - 1) | ((j < nv - 1) & (dnext != rw.dst));
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[t][r] = seg_scan16(S[t][r], sm);
#pragma unroll
    for (int r = 0; r < 4; ++r) VB[r] = seg_scan16(VB[r], sm);
    if (tail) {
        float* ms = p.msg_s + (size_t)rw.e * PF_S + 32 * wq + 4 * g;
        *reinterpret_cast<f32x4*>(ms) = S[0];
        *reinterpret_cast<f32x4*>(ms + 16) = S[1];
        if (wq < 3) {
            float* mv = p.msg_v + (size_t)rw.e * 48 + 12 * g + wq;
#pragma unroll
            for (int r = 0; r < 4; ++r) mv[3 * r] = VB[r];
        }
    }
}

// an item whose rows are gathered from global memory: slots -> (source, destination) -> coordinates / source rows -> type-table row
// (e: the row's edge slot, src / dst its endpoints -- requested by the caller, with or ahead of the item's count)
template <int KIND0>
__device__ __forceinline__ void n16_edge_item(const EdgeParams& p, const EncodeParams& ep, N16Lds* lds, const int e, const int src,
                                              const int dst, const int nv, const int et, const int lane, const int wq, int& sk) {
    N16_STAMP(sk, lane, wq);                              // item known (work list scanned)
    N16Ring ring;
    ring_start(ring, p.n16[et] + (size_t)wq * p.n16_stride[et], lane);      // in flight under the gathers
    const int g = lane >> 4;
    N16Rows rw;
    rw.e = e;
    rw.dst = dst;
    if (et == ET_PP && p.x0_static) {                     // (wave-uniform) a rigidly moving pocket: the edge's geometry from the original coordinates
        const float* a = p.x0_static + (size_t)src * 3;
        const float* b = p.x0_static + (size_t)rw.dst * 3;
        rw.xs = make_float4(a[0], a[1], a[2], 0.f); rw.xd = make_float4(b[0], b[1], b[2], 0.f);
    } else { rw.xs = p.xn[src]; rw.xd = p.xn[rw.dst]; }
    float XS[32], VB[4];
    f32x4 S[2];
    S[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; S[1] = S[0];
    if constexpr (KIND0 == N16_M0H) {
        // the h_src block of the first message Linear (+ its bias) is a row of a table: of the etype's type table when the source is a
        // protein atom (pf, pp), of the center hoist's P_et when it is a center (ff, fp; wave-uniform choice)
        pf_gcf tp;
        if (et == ET_PP || et == ET_PF) {
            int ty = p.ptype[src] * PF_S + p.ptab16_off[et];
            if (p.ptab_gstride) ty += p.l0_gid[src] * p.ptab_gstride;
            tp = (pf_gcf)p.ptab + ty;
        } else tp = (pf_gcf)p.pcen + ((size_t)(et == ET_FP ? p.pcen_nf : 0) + (size_t)(src - ep.Np)) * PF_S;
        tp += 32 * wq + 4 * g;
        S[0] = *reinterpret_cast<const f32x4 PF_AS1*>(tp);
        S[1] = *reinterpret_cast<const f32x4 PF_AS1*>(tp + 16);
    }
    if constexpr (KIND0 == N16_M0Z) {
        const float tt = ep.t ? ((pf_gcf)ep.t)[((const int PF_AS1*)ep.gid)[src]] : ep.t_scalar;
        n16_encode_pharm(ep, (pf_gcf)ep.pharm_h + (size_t)(src - ep.Np) * ep.pharm_nf, tt, XS, lds, lane, wq);
    }
    if constexpr (KIND0 == N16_M0F) {
        const f32x4 PF_AS1* hp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.h + (size_t)src * PF_S) + g;
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const f32x4 x = hp[4 * T];
#pragma unroll
            for (int r = 0; r < 4; ++r) XS[4 * T + r] = x[r];
        }
        pf_gcf vp = (pf_gcf)p.v + (size_t)src * 48 + 12 * g + (wq < 3 ? wq : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float x = vp[3 * r]; VB[r] = wq < 3 ? x : 0.f; }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) VB[r] = 0.f;
    }
    N16_CUT_AT(EDGE_CUT, 2, XS[0] + VB[0] + S[0][0] + rw.xs.x + rw.xd.x, p.msg_s);
    n16_edge_chain<KIND0>(p, ring, rw, XS, VB, S, lds, nv, lane, wq, sk);
}

}  // namespace pfn16
