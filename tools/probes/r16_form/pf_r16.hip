// pf_r16.hip -- conv layer 0, statically hoisted protein->protein edge items on 16-row groups (gfx950 only).
//
// Same mathematics as the hoisted items of pf_rg.hip (k_rg_edge<L0, ..., RGP>; DESIGN.md 4.1a: GVP.forward gvp.py:89-116
// for message GVPs 1.., the first one replaced by zs[slot] + ptab[type]), different mapping onto the matrix cores: a wave
// owns SIXTEEN consecutive edge slots and runs v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand
//
//   D[m][n] += A[m][k] B[k][n]      m = output feature of a 16-feature tile, n = row (edge slot), k = 4 inputs
//   lane l:  A: m = l & 15, k = l >> 4;   B: k = l >> 4, n = l & 15;   D register i: m = 4 (l >> 4) + i, n = l & 15
//
// so a 1-KiB quad of weights feeds 4 MFMAs of 32 cycles (the 4x4x1 form: 8 of 8 cycles at 8 rows per wave) and the
// streamed main Linear runs at 87-91 % of the matrix pipe (tools/probes/linear16_stream.hip).  The chain is LANE-LOCAL:
// register i of a D fragment holds feature 16 t + 4 g + i of row n on lane (g = l >> 4, n), which is exactly a B
// operand for "k-step 4 t + i, k = g" -- the next Linear consumes the fragment as it is, its weights packed in that
// order of K (pf_host.cpp: pack_gvp_r16).  Vector channels likewise: D register i of Vh / Vu / the gates = channel
// 4 g + i, so gating, |Vh| and the next Vh product need no data movement either.  No LDS, no cross-lane traffic until
// the per-destination sums of the finished rows (DPP segmented scan inside each 16-lane row).
//
// Block of the quad stream (one GVP + the pending gates of the one before it), in consumption order:
//   [gate bias][8 gate quads][Wh][8 main bias quads][32 main quads: k-steps 0..15][32 main quads: 16..31][Wu]
//   [8 main quads: sh k-steps] [pad to R16_PAD]          quad = [64 lanes][4 images], image = one A operand
// Flush block (end of chain): [gate bias][8 gate quads][pad].
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
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ f32x4 mfma16(const float a, const float b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f0(const float v) {      // lanes without a source read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i0(const int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

// scheduling barriers after every quad keep the ring loads where they are issued (without them the compiler sinks
// every load next to its use).  The loads walk the stream strictly sequentially through a SCALAR pointer that is made
// opaque after every step: with ordinary pointer arithmetic the compiler computes the 64-bit vector address of each of
// the ~100 loads of a block ahead of the first barrier (372-394 registers: one wave per SIMD)
#ifndef R16_SBM
#define R16_SBM 0
#endif
// R16_GROUPS: the issue order is pinned by a pipeline description instead (sched_group_barrier: one VMEM read, then the
// MFMAs of the quad, in source order), which leaves the ALU instructions free: 182 registers, two waves per SIMD
#ifdef R16_GROUPS
#define R16_SG_L() __builtin_amdgcn_sched_group_barrier(0x020, 1, 0)
#define R16_SG_M(N) __builtin_amdgcn_sched_group_barrier(0x008, N, 0)
#define R16_SB() do { } while (0)
#else
#define R16_SG_L() do { } while (0)
#define R16_SG_M(N) do { } while (0)
#define R16_SB() __builtin_amdgcn_sched_barrier(R16_SBM)
#endif
// in-kernel cycle stamps (diagnostic builds only: -DPF_STAMPS; tools/stamps_r16.py)
#ifdef PF_STAMPS
__device__ unsigned long long* g_r16_stamps = nullptr;
#define R16_STAMP(K) do { if (lane == 0 && g_r16_stamps && item < 64) g_r16_stamps[item * 16 + (K)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define R16_STAMP(K) do { } while (0)
#endif
#ifndef R16_D
#define R16_D 24                               // quads in the prefetch ring (a block: 17 k cycles at depth 8, 15 k at 24)
#endif
template <int D>
struct Ring16 {
    f32x4 q[D];
    pf_gcf pl;                                 // next quad to load (wave-uniform: scalar base + lane offset addressing)
};
static_assert(R16_PAD % R16_D == 0, "ring depth must divide the block padding");

// one quad: take it from the ring, refill the slot
__device__ __forceinline__ f32x4 r16_next(Ring16<R16_D>& ring, const int lane) {
    const f32x4 v = reinterpret_cast<const f32x4 PF_AS1*>(ring.pl)[lane];
    ring.pl += 256;
    asm volatile("" : "+s"(ring.pl));
    return v;
}
// quad QI of the current block (every block is a multiple of R16_D quads, so it sits in slot QI % R16_D); the slot is
// refilled with the next quad of the stream
#define R16_TAKE(W, QI) const f32x4 W = ring.q[(QI) % R16_D]; ring.q[(QI) % R16_D] = r16_next(ring, lane)

// pending gates of the previous GVP -> its gated vectors: Vin[c][i] = act(gate[i]) * vu[c][i]   (channel 4 g + i)
template <bool SIG, int Q0>
__device__ __forceinline__ void r16_gates(Ring16<R16_D>& ring, const float (&S)[32], const f32x4 (&vu)[3], f32x4 (&vin)[3], const int lane) {
    // four accumulators (one per image of a quad): back-to-back MFMAs on one accumulator wait for each other
    f32x4 ga[4];
    { R16_TAKE(w, Q0); ga[0] = w; R16_SG_L(); }                   // gate bias image = accumulator init
#pragma unroll
    for (int j = 1; j < 4; ++j) ga[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<0, 8>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        R16_TAKE(w, Q0 + 1 + q);
        static_for<0, 4>([&](auto J) { constexpr int j = decltype(J)::value; ga[j] = mfma16(w[j], S[4 * q + j], ga[j]); });
        R16_SG_L(); R16_SG_M(4); R16_SB();
    });
    const f32x4 gd = (ga[0] + ga[1]) + (ga[2] + ga[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float gv = SIG ? sigmoidf_(gd[i]) : gd[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) vin[c][i] = gv * vu[c][i];
    }
}

// one GVP (vi = vo = h = 16, 128 + 16 -> 128 scalars) with the pending gates of the previous one.
//   S   in: SiLU output of the previous GVP (register 4 t + i: feature 16 t + 4 g + i);  out: this GVP's
//   vu  in: pending Vu of the previous GVP (D layout: register i = channel 4 g + i);     out: this GVP's
__device__ __forceinline__ void r16_gvp(Ring16<R16_D>& ring, float (&S)[32], f32x4 (&vu)[3], const int lane) {
    f32x4 vin[3];
    r16_gates<true, 0>(ring, S, vu, vin, lane);                    // quads 0..8
    f32x4 wh;
    { R16_TAKE(w, 9); wh = w; R16_SG_L(); }
    f32x4 acc[8];
    static_for<0, 8>([&](auto T) { constexpr int t = decltype(T)::value; R16_TAKE(w, 10 + t); acc[t] = w; R16_SG_L(); });
    R16_SB();
    // main k-steps 0..15 (the sigmoids above retire under them)
    static_for<0, 32>([&](auto Q) {
        constexpr int q = decltype(Q)::value, ks = q / 2, half = q % 2;
        R16_TAKE(w, 18 + q);
        static_for<0, 4>([&](auto J) { constexpr int j = decltype(J)::value; acc[4 * half + j] = mfma16(w[j], S[ks], acc[4 * half + j]); });
        R16_SG_L(); R16_SG_M(4); R16_SB();
    });
    // Vh[c] = Wh^T Vin[c]
    f32x4 vh[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) vh[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<0, 4>([&](auto K) {
        constexpr int k = decltype(K)::value;
#pragma unroll
        for (int c = 0; c < 3; ++c) vh[c] = mfma16(wh[k], vin[c][k], vh[c]);
    });
    R16_SG_M(12); R16_SB();
    // main k-steps 16..31
    static_for<0, 32>([&](auto Q) {
        constexpr int q = decltype(Q)::value, ks = 16 + q / 2, half = q % 2;
        R16_TAKE(w, 50 + q);
        static_for<0, 4>([&](auto J) { constexpr int j = decltype(J)::value; acc[4 * half + j] = mfma16(w[j], S[ks], acc[4 * half + j]); });
        R16_SG_L(); R16_SG_M(4); R16_SB();
    });
    // Vu[c] = Wu^T Vh[c];  sh = |Vh|
    f32x4 wu;
    { R16_TAKE(w, 82); wu = w; R16_SG_L(); }
#pragma unroll
    for (int c = 0; c < 3; ++c) vu[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<0, 4>([&](auto K) {
        constexpr int k = decltype(K)::value;
#pragma unroll
        for (int c = 0; c < 3; ++c) vu[c] = mfma16(wu[k], vh[c][k], vu[c]);
    });
    float sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sh[i] = sqrtf_(fmaxf(vh[0][i] * vh[0][i] + vh[1][i] * vh[1][i] + vh[2][i] * vh[2][i], 1e-8f));
    R16_SG_M(12); R16_SB();
    static_for<0, 8>([&](auto Q) {
        constexpr int q = decltype(Q)::value, ks = q / 2, half = q % 2;
        R16_TAKE(w, 83 + q);
        static_for<0, 4>([&](auto J) { constexpr int j = decltype(J)::value; acc[4 * half + j] = mfma16(w[j], sh[ks], acc[4 * half + j]); });
        R16_SG_L(); R16_SG_M(4); R16_SB();
    });
    static_for<91, R16_NQ_GVP>([&](auto Q) { constexpr int q = decltype(Q)::value; R16_TAKE(w, q); (void)w; R16_SG_L(); });   // padding quads keep the slots aligned
    static_for<0, 8>([&](auto T) {
        constexpr int t = decltype(T)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) S[4 * t + i] = siluf_(acc[t][i]);
    });
}

// Items: 16 consecutive slots of a "pa" region (compact work list over regions [rbase, rbase + nreg)) or half a 32-slot
// tile of the list.
__global__ __launch_bounds__(64) void k_r16_pp(const EdgeParams p, const int rbase, const int tile0) {
    const int lane = threadIdx.x;
    const int item = blockIdx.x;
    int e0, nv;
    if (p.nreg > 0) {
        // compact work list, as in k_rg_edge: wave w takes the w-th non-empty group of 16 slots
        int first = 0, rsel = -1, cnt = 0, start = 0;
        int cs[RG_CPASS_R16], rs[RG_CPASS_R16];
#pragma unroll
        for (int k = 0; k < RG_CPASS_R16; ++k) {
            const int r = 64 * k + lane;
            cs[k] = r < p.nreg ? p.dyn_cnt[rbase + r] : 0;
            rs[k] = r < p.nreg ? p.reg[rbase + r] : 0;
        }
#pragma unroll
        for (int k = 0; k < RG_CPASS_R16; ++k) {
            if (64 * k < p.nreg && rsel < 0) {           // wave-uniform
                const int c = cs[k];
                const int ng = (c + 15) >> 4;
                int incl = ng;
                incl += dpp_i0<0x111>(incl); incl += dpp_i0<0x112>(incl); incl += dpp_i0<0x114>(incl); incl += dpp_i0<0x118>(incl);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
                incl += first;
                const unsigned long long m = __ballot(incl > item);
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
        const int loc = (item - first) * 16;
        e0 = start + loc;
        nv = __builtin_amdgcn_readfirstlane(min(16, cnt - loc));
    } else {
        if (item >= 2 * p.ntiles) return;
        const EdgeTile t = p.tiles[tile0 + item / 2];
        int nvalid = t.n;
        if (t.cnt_idx >= 0) nvalid = min(nvalid, max(p.dyn_cnt[t.cnt_idx] - t.rel, 0));
        const int base = (item % 2) * 16;
        nv = __builtin_amdgcn_readfirstlane(min(16, nvalid - base));
        if (nv <= 0) return;                           // wave-uniform
        e0 = t.e0 + base;
    }
    R16_STAMP(0);                                                  // work item known
    Ring16<R16_D> ring;
    ring.pl = p.r16;
    static_for<0, R16_D>([&](auto I) { ring.q[decltype(I)::value] = r16_next(ring, lane); });
    const int row = lane & 15, g = lane >> 4;
    const int e = e0 + min(row, nv - 1);
    const int src = p.esrc[e], dst = p.edst[e], eo = p.eorig[e];
    const float4 xs = p.xn[src], xd = p.xn[dst];
    int tyo = p.ptype[src] * PF_S;
    if (p.ptab_gstride) tyo += p.l0_gid[src] * p.ptab_gstride;
    float S[32];
    {
        const f32x4 PF_AS1* zp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.zs + (size_t)eo * PF_S) + g;
        const f32x4 PF_AS1* pp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.ptab + tyo) + g;
        f32x4 z[8], t8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { z[t] = zp[4 * t]; t8[t] = pp[4 * t]; }          // features 16 t + 4 g .. + 3
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) S[4 * t + i] = siluf_(z[t][i] + t8[t][i]);
    }
    const float dx = xs.x - xd.x, dy = xs.y - xd.y, dz = xs.z - xd.z;
    const float rd = rcpf_(sqrtf_(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f);
    const float xh[3] = {dx * rd, dy * rd, dz * rd};
    f32x4 vu[3];
    {
        const f32x4 weff = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.l0c)[g];       // channels 4 g .. 4 g + 3
#pragma unroll
        for (int c = 0; c < 3; ++c) vu[c] = weff * xh[c];
    }
    R16_STAMP(1);                                                  // rows gathered, S / Vu of the hoisted GVP ready
    for (int gi = 1; gi < p.n_gvps; ++gi) { r16_gvp(ring, S, vu, lane); R16_STAMP(1 + gi); }
    f32x4 vd[3];
    r16_gates<true, 0>(ring, S, vu, vd, lane);                     // flush: the last GVP's gates
    R16_STAMP(8);
    // per-destination sums of consecutive rows (slots are sorted by destination): segmented inclusive scan inside each
    // 16-lane row of the wave, one partial row per (item, destination) run, stored at the run's last slot
    const int did = row < nv ? dst : -1 - row;                     // rows beyond the item: their own segments, never stored
    const bool s1 = dpp_i0<0x111>(did + 1) == did + 1 && row >= 1;
    const bool s2 = dpp_i0<0x112>(did + 1) == did + 1 && row >= 2;
    const bool s4 = dpp_i0<0x114>(did + 1) == did + 1 && row >= 4;
    const bool s8 = dpp_i0<0x118>(did + 1) == did + 1 && row >= 8;
    auto seg = [&](float v) {
        float x = dpp_f0<0x111>(v); v += s1 ? x : 0.f;
        x = dpp_f0<0x112>(v); v += s2 ? x : 0.f;
        x = dpp_f0<0x114>(v); v += s4 ? x : 0.f;
        x = dpp_f0<0x118>(v); v += s8 ? x : 0.f;
        return v;
    };
    const int nxt = __builtin_amdgcn_update_dpp(-1, did, 0x101, 0xf, 0xf, false);       // row_shl:1 : the next row's id
    const bool last = row < nv && (row == nv - 1 || row == 15 || nxt != did);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = seg(S[4 * t + i]);
        if (last) *reinterpret_cast<f32x4*>(p.msg_s + (size_t)e * PF_S + 16 * t + 4 * g) = o;
    }
    {
        float o[12];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 * i + c] = seg(vd[c][i]);            // [channel 4 g + i][coordinate c]
        if (last) {
            f32x4* vp = reinterpret_cast<f32x4*>(p.msg_v + (size_t)e * 48 + 12 * g);
            vp[0] = (f32x4){o[0], o[1], o[2], o[3]};
            vp[1] = (f32x4){o[4], o[5], o[6], o[7]};
            vp[2] = (f32x4){o[8], o[9], o[10], o[11]};
        }
    }
    R16_STAMP(9);                                                  // stores issued
}

}  // namespace

extern "C" {
#ifdef PF_STAMPS
int pfk_r16_set_stamp_buffer(unsigned long long* dev) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_r16_stamps), &dev, sizeof(dev)); }
#endif
// the hoisted pp items of conv layer 0: compact regions [rbase, rbase + p->nreg) (grid: p->ngroups_sel groups of 16), or the
// ntiles tiles from tile0 of p->tiles
void pfk_r16_pp(const EdgeParams* p, int rbase, int tile0, hipStream_t s) {
    const int grid = p->nreg > 0 ? p->ngroups_sel : 2 * p->ntiles;
    if (grid <= 0) return;
    hipLaunchKernelGGL(k_r16_pp, dim3(grid), dim3(64), 0, s, *p, rbase, tile0);
}
}
