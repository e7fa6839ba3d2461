// pf_rgk.hip -- "KS" form of the row-group kernels (gfx950 only): one 4-row group on the FOUR waves of a workgroup,
// i.e. on the four SIMDs of a CU, with the K dimension of every scalar Linear split over the waves.
//
// Why: at the headline batch (32 graphs) every launch of a denoising step has fewer row groups than the chip has SIMDs,
// so a launch lasts as long as ONE wave needs for its chain of GVPs (~6,000 cycles per GVP block for a
// lone wave: 352 MFMAs + 96 one-KiB weight loads that one wave can neither issue nor stream faster).  Here wave w of a
// group owns the k-steps (m, a) with m in {2w, 2w+1} of the 128-deep main product and a quarter of the rbf / sh
// k-steps: 72 instead of 322 scalar MFMAs and 16-20 instead of 72 main weight quads per wave and GVP.  The partial
// pre-activations of the four waves meet in LDS (SD layout: lane = output feature) behind ONE workgroup barrier per
// GVP, and every wave reads back the complete SiLU output in the SA layout (lane 4a+i: features 8a..8a+7 of row i) --
// the reduction and the SD -> SA transposition of pf_rg.hip are the same LDS round trip.  The vector channel
// (Vh, Vu: 32 MFMAs) and the scalar -> vector gates (32 MFMAs) are computed by every wave: they are short, and their
// operands are exactly what the exchange leaves in every wave's registers.  The barrier orders LDS traffic only
// (s_waitcnt lgkmcnt(0); s_barrier): the weight prefetch ring stays in flight across it.
//
// Same mathematics as pf_rg.hip / pf_kernels.hip (GVP.forward gvp.py:89-116, GVPMultiEdgeConv gvp.py:459-551,
// GVPLayerNorm gvp.py:159-166, NoisePredictionBlock dynamics_gvp.py:37-42); the k-steps of a dot product are summed in
// another order (four partial sums, added in wave order), which the fp32 parity tolerance covers.  Results are
// bitwise reproducible: nothing depends on which wave arrives first.
#include "pf_rg_common.h"

namespace {

#ifndef KS_D
#define KS_D KS_PAD                            // ring depth: every block of a wave's stream is a multiple of it
#endif
#define KS_XS 132                              // floats per row of the exchange buffer (128 + pad, 16-byte aligned)
struct __attribute__((aligned(16))) KsLds {
    float xt[2][4][4 * KS_XS];                 // [parity][wave][row][feature]: partial pre-activations of a scalar Linear
    float tv[4][4 * RG_TV_STRIDE];             // per wave: VD <-> VA transposition of the vector channel
};
// In-kernel cycle stamps (diagnostic builds only: -DPF_KS_STAMPS; no stamp executes in the product build): lane 0 of
// every wave of the first 16 workgroups writes s_memtime at the phase boundaries (tools/stamps_ks.py)
#ifdef PF_KS_STAMPS
__device__ unsigned long long* g_ks_stamps = nullptr;
struct KsStamp {
    int k = 0, slot = -1;
    __device__ __forceinline__ void operator()(const int lane) {
        if (lane == 0 && g_ks_stamps && slot >= 0 && k < 96) g_ks_stamps[slot * 96 + k] = __builtin_amdgcn_s_memtime();
        ++k;
    }
};
#define KS_STAMP(st, lane) (st)(lane)
#else
struct KsStamp { int slot = -1; };
#define KS_STAMP(st, lane) ((void)0)
#endif

// workgroup barrier that orders LDS traffic only (the __syncthreads() of the compiler also drains vmcnt, i.e. waits for
// the whole weight prefetch ring)
__device__ __forceinline__ void ks_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// One block of a chain = one GVP on a group of four rows, seen from wave wq of the group's four.
//   X   [8]  in: scalar input, SA layout (complete, in every wave)        out: SiLU output, SA layout (complete)
//   Va  [4]  in (PREV == 0): vector input, VA layout; with a pending gate it is produced here
//   R4       first message GVP only: lane 4a+i holds rbf_{4 wq + (a & 3)}(d_i) -- this wave's four rbf k-steps sit in
//            A blocks 0..3
//   XH       first message GVP only: unit x_diff (lane 16g+i: xhat_i[g])
//   carry    in: pending Vu / gate bias of the previous GVP (PREV != 0); out: this GVP's
//   par      parity of the exchange buffer (alternates per block: a fast wave may write the next block's partials while
//            a slow one still reads this block's)
// ---------------------------------------------------------------------------------------------
template <class S, int PREV, bool VZERO>
__device__ __forceinline__ void ks_gvp(RgRing<KS_D>& ring, float (&X)[8], float (&Va)[4], const float R4, const float XH,
                                       RgCarry<1>& carry, KsLds* L, const int lane, const int wq, int& par, KsStamp& stamp) {
    constexpr int NH = S::NH;
    KS_STAMP(stamp, lane);                             // block start
    static_assert(!(VZERO && PREV != 0), "VZERO is a property of a chain's first GVP");
    const int a = lane >> 2, i = lane & 3, g = lane >> 4, q = a & 3, u = lane & 15;
    const int gg = g < 3 ? g : 2;
    float* tv = L->tv[wq];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 lo[2] = {z4, z4}, hi[2] = {z4, z4}, vh[2] = {z4, z4}, vu[2] = {z4, z4}, gd[2] = {z4, z4};
    f32x4 cq = z4, xhq = z4, w16 = z4, VhA = z4;
    float SHk = 0.f, SH16 = 0.f, Vh16 = 0.f;
    // this wave's two k-registers of the main product: m = 2 wq, 2 wq + 1 (wave-uniform selects)
    const float XK0 = wq == 0 ? X[0] : (wq == 1 ? X[2] : (wq == 2 ? X[4] : X[6]));
    const float XK1 = wq == 0 ? X[1] : (wq == 1 ? X[3] : (wq == 2 ? X[5] : X[7]));

    constexpr KsSched QQ = ks_sched(S::VI, S::NEXTRA, NH, PREV != 0);
    static_for<0, QQ.nq>([&](auto QI) {
        constexpr int qi = decltype(QI)::value;
        constexpr KsSched Q = ks_sched(S::VI, S::NEXTRA, NH, PREV != 0);
        const f32x4 w = ring.q[qi % KS_D];
        ring.q[qi % KS_D] = ring.p[(qi + KS_D) * 64];
        constexpr int kmain = (qi >= Q.q_m0 && qi < Q.q_m0 + Q.nm0) ? qi - Q.q_m0
                            : (qi >= Q.q_m1 && qi < Q.q_m1 + Q.nm1) ? Q.nm0 + qi - Q.q_m1
                            : (qi >= Q.q_m2 && qi < Q.q_m2 + Q.nm2) ? Q.nm0 + Q.nm1 + qi - Q.q_m2 : -1;
        if constexpr (qi == Q.q_c) {
            cq = w;                                   // [bias lo, bias hi, gate bias, Wh[0][16] on the xhat lanes]
            const float b0 = wq == 0 ? w[0] : 0.f, b1 = (wq == 0 && NH == 2) ? w[1] : 0.f;     // wave 0's partial carries the bias
            lo[0] = (f32x4){b0, b0, b0, b0};
            hi[0] = (f32x4){b1, b1, b1, b1};
        } else if constexpr (S::H17 && qi == Q.q_xh) {
            xhq = w;                                  // [Wh[0][:] image, Wu[16][:] image, sh16 column lo, hi]
            vh[0] = mfma_b2<0>(XH, w[0], vh[0]);
        } else if constexpr (S::H17 && qi == Q.q_xh + 1) {
            w16 = w;                                  // Wh[1 + 4t + q][16], t = 0..3
        } else if constexpr (kmain >= 0) {
            constexpr int half = kmain % NH, mq = kmain / NH, mm = mq / 4, aq = mq % 4;
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (half == 0) lo[j & 1] = mfma_b4<4 * aq + j>(mm ? XK1 : XK0, w[j], lo[j & 1]);
                else hi[j & 1] = mfma_b4<4 * aq + j>(mm ? XK1 : XK0, w[j], hi[j & 1]);
            });
        } else if constexpr (PREV != 0 && qi >= Q.q_gate && qi < Q.q_gate + 8) {
            constexpr int m = qi - Q.q_gate;          // pending gates of the previous GVP, K split over the lane groups
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                gd[j & 1] = mfma_b2<j>(X[m], w[j], gd[j & 1]);
            });
            if constexpr (m == 7) {                   // sum the K quarters, activation, gate the pending Vu, publish
                KS_STAMP(stamp, lane);                // gate MFMAs issued
                gd[0] += gd[1];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    float gv = gsum(gd[0][ii]) + carry.bg;
                    if constexpr (PREV == 1) gv = sigmoidf_(gv);
                    const float vd = gv * carry.vu[0][ii];
                    if (lane < 48) tv[ii * RG_TV_STRIDE + g * 16 + pperm(u)] = vd;
                }
            }
        } else if constexpr (qi >= Q.q_vh && qi < Q.q_vh + 4) {
            constexpr int t = qi - Q.q_vh;
            if constexpr (t == 0 && PREV != 0) {      // the gated vectors of the previous GVP are back: VA layout
                __builtin_amdgcn_wave_barrier();
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(&tv[i * RG_TV_STRIDE + gg * 16 + 4 * q]);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) Va[tt] = g < 3 ? v4[tt] : 0.f;
            }
            if constexpr (!VZERO) {
                static_for<0, 4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    vh[j & 1] = mfma_b2<j>(Va[t], w[j], vh[j & 1]);
                });
            }
            if constexpr (t == 3) {                   // Vh complete: hidden channel 16 on the VALU, publish Vh
                KS_STAMP(stamp, lane);                // Vh MFMAs issued
                vh[0] += vh[1];
                if constexpr (S::H17) {
                    float pz = XH * cq[3];
                    if constexpr (!VZERO) {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) pz = fmaf(Va[tt], w16[tt], pz);
                    }
                    Vh16 = qsum(pz);
                }
                if (lane < 48) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) tv[ii * RG_TV_STRIDE + g * 16 + pperm(u)] = vh[0][ii];
                }
                if constexpr (S::H17) {
                    if (q == 0 && g < 3) tv[i * RG_TV_STRIDE + 48 + g] = Vh16;
                }
            }
        } else if constexpr (qi >= Q.q_vu && qi < Q.q_vu + 4) {
            constexpr int t = qi - Q.q_vu;
            if constexpr (t == 0) {                   // Vh is back from LDS: A images of the Vu product, this wave's sh
                __builtin_amdgcn_wave_barrier();
                VhA = *reinterpret_cast<const f32x4*>(&tv[i * RG_TV_STRIDE + gg * 16 + 4 * q]);
                // hidden channel c = 4 wq + q sits at position pperm(c) = 4 q + wq of a coordinate's 16 slots
                const float x = tv[i * RG_TV_STRIDE + 4 * q + wq], y = tv[i * RG_TV_STRIDE + 16 + 4 * q + wq],
                            z = tv[i * RG_TV_STRIDE + 32 + 4 * q + wq];
                SHk = sqrtf_(fmaxf(x * x + y * y + z * z, 1e-8f));
                if constexpr (S::H17) {
                    const float x6 = tv[i * RG_TV_STRIDE + 48], y6 = tv[i * RG_TV_STRIDE + 49], z6 = tv[i * RG_TV_STRIDE + 50];
                    SH16 = sqrtf_(fmaxf(x6 * x6 + y6 * y6 + z6 * z6, 1e-8f));
                }
            }
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                vu[j & 1] = mfma_b2<j>(VhA[t], w[j], vu[j & 1]);
            });
            if constexpr (t == 3) {
                KS_STAMP(stamp, lane);                // Vu MFMAs issued
                vu[0] += vu[1];
                if constexpr (S::H17) vu[0] = mfma_b2<0>(Vh16, xhq[1], vu[0]);
            }
        } else if constexpr (S::NEXTRA > 0 && qi >= Q.q_rbf && qi < Q.q_rbf + NH) {
            constexpr int half = qi - Q.q_rbf;        // this wave's rbf k-steps 4 wq .. 4 wq + 3 (A blocks 0..3 of R4)
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (half == 0) lo[j & 1] = mfma_b4<j>(R4, w[j], lo[j & 1]);
                else hi[j & 1] = mfma_b4<j>(R4, w[j], hi[j & 1]);
            });
        } else if constexpr (qi >= Q.q_sh && qi < Q.q_sh + NH) {
            constexpr int half = qi - Q.q_sh;         // this wave's sh k-steps: hidden channels 4 wq .. 4 wq + 3
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (half == 0) lo[j & 1] = mfma_b4<j>(SHk, w[j], lo[j & 1]);
                else hi[j & 1] = mfma_b4<j>(SHk, w[j], hi[j & 1]);
            });
            if constexpr (half == NH - 1) {           // this wave's part of the scalar Linear is complete
                lo[0] += lo[1];
                hi[0] += hi[1];
                if constexpr (S::H17) {
                    if (wq == 0) {                    // hidden channel 16 (wave-uniform branch)
                        lo[0] = mfma_b4<0>(SH16, xhq[2], lo[0]);
                        if constexpr (NH == 2) hi[0] = mfma_b4<0>(SH16, xhq[3], hi[0]);
                    }
                }
                float* xw = L->xt[par][wq];
                KS_STAMP(stamp, lane);                // scalar k-steps issued
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    xw[ii * KS_XS + lane] = lo[0][ii];
                    if constexpr (NH == 2) xw[ii * KS_XS + 64 + lane] = hi[0][ii];
                }
                KS_STAMP(stamp, lane);                // partials written
                ks_barrier();                         // the four partial sums are in LDS
                KS_STAMP(stamp, lane);                // barrier passed
                f32x4 s0 = z4, s1 = z4;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {      // fixed order: wave 0 (with the bias) first
                    const float* xr = &L->xt[par][ww][i * KS_XS + 8 * a];
                    s0 += *reinterpret_cast<const f32x4*>(xr);
                    s1 += *reinterpret_cast<const f32x4*>(xr + 4);
                }
                const bool on = NH == 2 || a < 8;     // 64 outputs: features live in blocks 0..7 only
#pragma unroll
                for (int m = 0; m < 4; ++m) { X[m] = on ? siluf_(s0[m]) : 0.f; X[4 + m] = on ? siluf_(s1[m]) : 0.f; }
                KS_STAMP(stamp, lane);                // SiLU output in the SA layout
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    ring.p += QQ.nq * 64;
    par ^= 1;
    carry.vu[0] = vu[0];
    carry.bg = cq[2];
}

// end of a chain: the pending gates of its last GVP (every wave).  Vd: gated vector output, VD layout (lane 16g+u:
// channel u, coordinate g); Va: the same in the VA layout (when NEEDVA)
template <bool SIG, bool NEEDVA>
__device__ __forceinline__ void ks_flush(RgRing<KS_D>& ring, const float (&X)[8], float (&Va)[4], f32x4& Vd, const RgCarry<1>& carry,
                                         KsLds* L, const int lane, const int wq) {
    const int i = lane & 3, g = lane >> 4, q = (lane >> 2) & 3, u = lane & 15;
    const int gg = g < 3 ? g : 2;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 gd[2] = {z4, z4};
    static_for<0, KS_NQ_FLUSH>([&](auto QI) {
        constexpr int qi = decltype(QI)::value;
        const f32x4 w = ring.q[qi % KS_D];
        ring.q[qi % KS_D] = ring.p[(qi + KS_D) * 64];
        if constexpr (qi < 8) {
            static_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                gd[j & 1] = mfma_b2<j>(X[qi], w[j], gd[j & 1]);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    ring.p += KS_NQ_FLUSH * 64;
    gd[0] += gd[1];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        float gv = gsum(gd[0][ii]) + carry.bg;
        if constexpr (SIG) gv = sigmoidf_(gv);
        Vd[ii] = gv * carry.vu[0][ii];
    }
    if constexpr (NEEDVA) {
        float* tv = L->tv[wq];
        if (lane < 48) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) tv[ii * RG_TV_STRIDE + g * 16 + pperm(u)] = Vd[ii];
        }
        __builtin_amdgcn_wave_barrier();
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(&tv[i * RG_TV_STRIDE + gg * 16 + 4 * q]);
#pragma unroll
        for (int t = 0; t < 4; ++t) Va[t] = g < 3 ? v4[t] : 0.f;
    }
}

// L2 warm-up workgroup wb of nw: XCDs have private L2s that start every launch cold, and a lone CU pulls cold lines at
// ~11-18 B/clk, so a row group's weight ring (one reader per CU) is bound by that fill rate; here the 32 warm-up
// workgroups that share an XCD (workgroup ids congruent mod 8) read 1/32 of the stream each, i.e. the XCD fetches the
// stream with all its CUs at once and the rings then hit L2.  Pure speed: nothing depends on the placement.
__device__ __forceinline__ float ks_warm(pf_gcf base, const long nfloats, const int wb, const int nw) {
    const int per_xcd = max(nw >> 3, 1), slice = (wb >> 3) % per_xcd;
    const long nvec = nfloats >> 2, chunk = (nvec + per_xcd - 1) / per_xcd;
    const long lo = (long)slice * chunk, hi = min(lo + chunk, nvec);
    const f32x4 PF_AS1* v = reinterpret_cast<const f32x4 PF_AS1*>(base);
    float acc = 0.f;
    for (long i = lo + threadIdx.x; i < hi; i += 256) { const f32x4 x = v[i]; acc += x[0]; }
    return acc;
}

// quad_perm [0, 0, 1, 2]: lane i of every group of four reads lane i - 1 (lane 0 its own value)
__device__ __forceinline__ float quad_prev(const float v) { return dpp_f<0x90>(v); }
__device__ __forceinline__ int quad_prev_i(const int v) { return __builtin_amdgcn_update_dpp(0, v, 0x90, 0xf, 0xf, false); }
// quad_perm [1, 2, 3, 3]: lane i reads lane i + 1
__device__ __forceinline__ int quad_next_i(const int v) { return __builtin_amdgcn_update_dpp(0, v, 0xf9, 0xf, 0xf, false); }

// ---------------------------------------------------------------------------------------------
// Edge messages (gvp.py:472-485, 540-551): a workgroup = the four slots [e0, e0 + nv) of etype et.
// PRE (conv layer 0, static pp edges; EdgeParams::zs): the chain starts at its second block.
// ---------------------------------------------------------------------------------------------
template <bool L0, bool PRE>
__device__ __forceinline__ void ks_edge_item(const EdgeParams& p, const EncodeParams& ep, KsLds* L, const int wq,
                                             const int e0, const int nv, const int et, const int lane) {
    static_assert(!(PRE && !L0), "the static hoist is a conv-layer-0 path");
    int par = 0;
    KsStamp stamp;
    stamp.slot = ((int)blockIdx.x - p.warm_wgs) < 16 && (int)blockIdx.x >= p.warm_wgs ? ((int)blockIdx.x - p.warm_wgs) * 4 + wq : -1;
    KS_STAMP(stamp, lane);                             // kernel start
    RgRing<KS_D> ring;
    constexpr int NQ0 = ks_sched(SpecMsg0::VI, SpecMsg0::NEXTRA, 2, false).nq;          // quads of the chain's first block
    ring_start(ring, p.rgk[et] + (size_t)wq * p.rgk_stride + (PRE ? (size_t)NQ0 * 256 : 0), lane);   // in flight under the gather
    const int a = lane >> 2, i = lane & 3, g = lane >> 4, q = a & 3, u = lane & 15;
    float X1[1][8], Va[4], R4 = 0.f, XH;
    f32x4 Vd;
    RgCarry<1> carry;
    const int e = e0 + min(i, nv - 1);
    const int src = p.esrc[e], dst = p.edst[e];
    const float4 xs = p.xn[src], xd = p.xn[dst];
    const float dx = xs.x - xd.x, dy = xs.y - xd.y, dz = xs.z - xd.z;
    const float d = sqrtf_(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
    XH = (g == 0 ? dx : (g == 1 ? dy : (g == 2 ? dz : 0.f))) * rcpf_(d);
    float (&X)[8] = X1[0];
    if constexpr (PRE) {
        const float weff = g < 3 ? p.l0c[u] : 0.f;
        carry.bg = p.l0c[16 + u];
        const int eo = p.eorig[e];
        int ty = p.ptype[src];
        if (p.ptab_gstride) ty = ty * PF_S + p.l0_gid[src] * p.ptab_gstride; else ty *= PF_S;
        const f32x4 PF_AS1* zp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.zs + (size_t)eo * PF_S) + 2 * a;
        const f32x4 PF_AS1* pp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.ptab + ty) + 2 * a;
        const f32x4 z0 = zp[0], z1 = zp[1], p0 = pp[0], p1 = pp[1];
#pragma unroll
        for (int m = 0; m < 4; ++m) { X[m] = siluf_(z0[m] + p0[m]); X[4 + m] = siluf_(z1[m] + p1[m]); }
        carry.vu[0] = (f32x4){quad_bcast<0>(XH) * weff, quad_bcast<1>(XH) * weff, quad_bcast<2>(XH) * weff, quad_bcast<3>(XH) * weff};
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) Va[tt] = 0.f;
        for (int gi = 1; gi < p.n_gvps; ++gi) ks_gvp<SpecGen, 1, false>(ring, X, Va, 0.f, XH, carry, L, lane, wq, par, stamp);
    } else {
        const float mu_step = (p.rbf_mu[PF_R - 1] - p.rbf_mu[0]) * (1.0f / (float)(PF_R - 1));
        const float mu_k = fmaf((float)(4 * wq + q), mu_step, p.rbf_mu[0]);        // rbf k-step 4 wq + (a & 3) in A block a & 3
        const float ze = (d - mu_k) * p.rbf_inv_sigma;
        R4 = __expf(-(ze * ze));
        if (!(L0 && ep.w[0])) {
            const f32x4 PF_AS1* hp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.h + (size_t)src * PF_S) + 2 * a;
            const f32x4 x0 = hp[0], x1 = hp[1];
#pragma unroll
            for (int m = 0; m < 4; ++m) { X[m] = x0[m]; X[4 + m] = x1[m]; }
        }
        if constexpr (!L0) {
            pf_gcf vp = (pf_gcf)p.v + (size_t)src * 48 + (g < 3 ? g : 0) + 3 * q;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { const float x = vp[12 * tt]; Va[tt] = g < 3 ? x : 0.f; }
        } else {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) Va[tt] = 0.f;
        }
        if (L0 && ep.w[0]) {
            const int srcv[1] = {src};
            rg_encode<1>(ep, (et == ET_FF || et == ET_FP) ? 1 : 0, srcv, X1, lane);       // sources: pharm for ff / fp
        }
        ks_gvp<SpecMsg0, 0, L0>(ring, X, Va, R4, XH, carry, L, lane, wq, par, stamp);
        for (int gi = 1; gi < p.n_gvps; ++gi) ks_gvp<SpecGen, 1, false>(ring, X, Va, 0.f, XH, carry, L, lane, wq, par, stamp);
    }
    KS_STAMP(stamp, lane);                             // chain done
    ks_flush<true, false>(ring, X, Va, Vd, carry, L, lane, wq);
    KS_STAMP(stamp, lane);                             // flushed
    // ---- per-destination sums in slot order; one partial row per (group, destination) run, at the run's last slot
    // scalars (SA layout: row i on lane 4a+i): wave wq sums and stores features 8a + 2 wq, 8a + 2 wq + 1
    {
        const int dprev = quad_prev_i(dst), dnext = quad_next_i(dst);
        const bool same = i > 0 && i < nv && dprev == dst;
        const bool ends = i < nv && (i == nv - 1 || dnext != dst);
        float s0 = wq == 0 ? X[0] : (wq == 1 ? X[2] : (wq == 2 ? X[4] : X[6]));
        float s1 = wq == 0 ? X[1] : (wq == 1 ? X[3] : (wq == 2 ? X[5] : X[7]));
#pragma unroll
        for (int r = 1; r < 4; ++r) {                   // s_i = same_i ? s_{i-1} + x_i : x_i, rows in slot order
            const float t0 = quad_prev(s0), t1 = quad_prev(s1);
            if (i == r && same) { s0 = t0 + s0; s1 = t1 + s1; }
        }
        if (ends) *reinterpret_cast<float2*>(p.msg_s + (size_t)(e0 + i) * PF_S + 8 * a + 2 * wq) = make_float2(s0, s1);
    }
    // vectors (VD layout: row ii in register ii, identical in every wave): wave 0
    if (wq == 0) {
        float av = 0.f;
        int prev = -1;
        static_for<0, 4>([&](auto K) {
            constexpr int k = decltype(K)::value;
            if (k < nv) {
                const int dk = __builtin_amdgcn_readlane(dst, k);
                if (k > 0 && dk == prev) av += Vd[k];
                else {
                    if (k > 0 && lane < 48) p.msg_v[(size_t)(e0 + k - 1) * 48 + 3 * u + g] = av;
                    av = Vd[k];
                }
                prev = dk;
            }
        });
        if (lane < 48) p.msg_v[(size_t)(e0 + nv - 1) * 48 + 3 * u + g] = av;
    }
}

template <bool L0, bool HOIST>
__global__ __launch_bounds__(256) void k_rgk_edge(const EdgeParams p, const EncodeParams ep) {
    __shared__ KsLds lds;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((int)blockIdx.x < p.warm_wgs) {               // L2 warm-up workgroup (workgroup-uniform)
        const float acc = ks_warm(p.warm_base, p.warm_floats, (int)blockIdx.x, p.warm_wgs);
        if (acc == 1.2345e-30f) p.msg_s[0] = acc;     // never true: keeps the loads
        return;
    }
    const int item = (int)blockIdx.x - p.warm_wgs;
    const bool hoist = HOIST && p.zs != nullptr;       // wave-uniform (kernel argument)
    int e0, nv, et;
    bool pre = false;
    if (p.nreg > 0) {
        // compact work list (see k_rg_edge): group w of 4 slots, counting only the groups that hold edges
        const int w = item;
        int first = 0, rsel = -1, cnt = 0, start = 0;
        int cs[RG_CPASS], rs[RG_CPASS];
#pragma unroll
        for (int k = 0; k < RG_CPASS; ++k) {
            const int r = 64 * k + lane;
            cs[k] = r < p.nreg ? p.dyn_cnt[r] : 0;
            rs[k] = r < p.nreg ? p.reg[r] : 0;
        }
#pragma unroll
        for (int k = 0; k < RG_CPASS; ++k) {
            if (64 * k < p.nreg && rsel < 0) {           // wave-uniform
                const int c = cs[k];
                const int ng = (c + 3) >> 2;
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
        if (rsel < 0) return;                          // workgroup-uniform: beyond the last group
        const int kind = rsel / p.regB;
        pre = hoist && kind == 3;
        const int loc = (w - first) * 4;
        e0 = start + loc;
        nv = __builtin_amdgcn_readfirstlane(min(4, cnt - loc));
        et = kind == 3 ? (int)ET_PP : kind;            // fourth region kind: pp edges into the active atoms
    } else {
        if (item >= p.ntiles * 8) return;
        const EdgeTile t = p.tiles[item / 8];
        int nvalid = t.n;
        if (t.cnt_idx >= 0) nvalid = min(nvalid, max(p.dyn_cnt[t.cnt_idx] - t.rel, 0));
        const int base = (item % 8) * 4;
        nv = __builtin_amdgcn_readfirstlane(min(4, nvalid - base));
        if (nv <= 0) return;                           // workgroup-uniform
        et = __builtin_amdgcn_readfirstlane(t.et);
        e0 = t.e0 + base;
        pre = hoist && et == ET_PP;
    }
    if constexpr (HOIST) {
        if (pre) { ks_edge_item<L0, true>(p, ep, &lds, wq, e0, nv, et, lane); return; }
    }
    ks_edge_item<L0, false>(p, ep, &lds, wq, e0, nv, et, lane);
}

// ---------------------------------------------------------------------------------------------
// Node update (gvp.py:488-536) of four nodes per workgroup; HEAD: the noise head (dynamics_gvp.py:37-42) follows on
// the registers.  Every wave gathers and normalises the rows (the loads hit the CU's L1 after the first wave); only
// the scalar Linears are split.  Wave 0 stores.
// ---------------------------------------------------------------------------------------------
template <bool L0, bool HEAD>
__global__ __launch_bounds__(256) void k_rgk_node(const NodeParams p, const HeadParams hp, const EncodeParams ep) {
    __shared__ KsLds lds;
    KsLds* L = &lds;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((int)blockIdx.x < p.warm_wgs) {               // L2 warm-up workgroup (workgroup-uniform)
        const float acc = ks_warm(p.warm_base, p.warm_floats, (int)blockIdx.x, p.warm_wgs);
        if (acc == 1.2345e-30f) p.h_out[0] = acc;     // never true: keeps the loads
        return;
    }
    const int bid = (int)blockIdx.x - p.warm_wgs;
    if (bid >= p.ntiles * 8) return;
    const NodeTile t = p.tiles[bid / 8];
    int tn = t.n;
    if (t.cnt_idx >= 0) tn = min(tn, max(p.dyn_cnt[t.cnt_idx] - t.rel, 0));
    const int base = (bid % 8) * 4;
    const int nv = __builtin_amdgcn_readfirstlane(min(4, tn - base));
    if (nv <= 0) return;                               // workgroup-uniform
    const int nt = __builtin_amdgcn_readfirstlane(t.ntype);
    int par = 0;
    KsStamp stamp;
    stamp.slot = bid < 16 ? bid * 4 + wq : -1;
    KS_STAMP(stamp, lane);                             // kernel start
    RgRing<KS_D> ring;
    ring_start(ring, p.rgk_upd[nt] + (size_t)wq * p.rgk_stride[nt], lane);
    const int a = lane >> 2, i = lane & 3, g = lane >> 4, q = a & 3, u = lane & 15;
    const int gc = g < 3 ? g : 0;
    const NodeW nw = p.w[nt];
    float X1[1][8], Va1[1][4];
    float (&X)[8] = X1[0];
    float (&Va)[4] = Va1[0];
    int nid[1];
    const int gm0 = p.grp - 1, gm1 = (nt == 0 ? p.grp_pa : p.grp) - 1;
    {
        const int row = base + min(i, nv - 1);
        const int n = t.ids ? p.row_ids[t.n0 + row] : t.n0 + row;
        nid[0] = n;
        float as[8], avv[4];
#pragma unroll
        for (int m = 0; m < 8; ++m) as[m] = 0.f;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) avv[tt] = 0.f;
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
            for (int m = 0; m < 8; ++m) X[m] = as[m] * inv_norm;
        } else {
            const f32x4 PF_AS1* hpn = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)p.h_in + (size_t)n * PF_S) + 2 * a;
            const f32x4 h0 = hpn[0], h1 = hpn[1];
#pragma unroll
            for (int m = 0; m < 4; ++m) { X[m] = fmaf(as[m], inv_norm, h0[m]); X[4 + m] = fmaf(as[4 + m], inv_norm, h1[m]); }
        }
        if constexpr (!L0) {
            pf_gcf vp = (pf_gcf)p.v_in + (size_t)n * 48 + gc + 3 * q;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) Va[tt] = fmaf(avv[tt], inv_norm, vp[12 * tt]);
        } else {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) Va[tt] = avv[tt] * inv_norm;
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) Va[tt] = g < 3 ? Va[tt] : 0.f;
    }
    if (L0 && ep.w[0]) {
        float H[1][8];
        rg_encode<1>(ep, nt, nid, H, lane);
#pragma unroll
        for (int m = 0; m < 8; ++m) X[m] += H[0][m];
    }
    KS_STAMP(stamp, lane);                             // rows gathered (+ encoded)
    rg_layernorm<1>(nw.ln1_w, nw.ln1_b, X1, Va1, lane);
    KS_STAMP(stamp, lane);                             // first LayerNorm
    float Xr[8], Vr[4];
#pragma unroll
    for (int m = 0; m < 8; ++m) Xr[m] = X[m];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) Vr[tt] = Va[tt];
    f32x4 Vd;
    RgCarry<1> carry;
    ks_gvp<SpecGen, 0, false>(ring, X, Va, 0.f, 0.f, carry, L, lane, wq, par, stamp);
    for (int gi = 1; gi < p.n_upd; ++gi) ks_gvp<SpecGen, 1, false>(ring, X, Va, 0.f, 0.f, carry, L, lane, wq, par, stamp);
    ks_flush<true, true>(ring, X, Va, Vd, carry, L, lane, wq);
#pragma unroll
    for (int m = 0; m < 8; ++m) X[m] += Xr[m];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) Va[tt] += Vr[tt];
    rg_layernorm<1>(nw.ln2_w, nw.ln2_b, X1, Va1, lane);
    KS_STAMP(stamp, lane);                             // update chain + second LayerNorm
    if constexpr (!HEAD) {
        if (i < nv && wq == 0) {                       // (every wave holds the same result)
            f32x4* op = reinterpret_cast<f32x4*>(p.h_out + (size_t)nid[0] * PF_S) + 2 * a;
            op[0] = (f32x4){X[0], X[1], X[2], X[3]};
            op[1] = (f32x4){X[4], X[5], X[6], X[7]};
            if (g < 3) {
                float* vp = p.v_out + (size_t)nid[0] * 48 + g + 3 * q;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) vp[12 * tt] = Va[tt];
            }
        }
    } else {
        // noise head: its chain and to_scalar_output follow the update chain in the quad stream
        if (hp.n_gvps == 1) ks_gvp<SpecHeadLast, 0, false>(ring, X, Va, 0.f, 0.f, carry, L, lane, wq, par, stamp);
        else {
            ks_gvp<SpecGen, 0, false>(ring, X, Va, 0.f, 0.f, carry, L, lane, wq, par, stamp);
            for (int gi = 1; gi + 1 < hp.n_gvps; ++gi) ks_gvp<SpecGen, 1, false>(ring, X, Va, 0.f, 0.f, carry, L, lane, wq, par, stamp);
            ks_gvp<SpecHeadLast, 1, false>(ring, X, Va, 0.f, 0.f, carry, L, lane, wq, par, stamp);
        }
        KS_STAMP(stamp, lane);                         // head chain done
        ks_flush<false, false>(ring, X, Va, Vd, carry, L, lane, wq);
        // to_scalar_output: Linear(64 -> pharm_nf), K split over the lane groups like the gates (every wave)
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 od = z4, oc = z4;
        static_for<0, KS_NQ_OUT>([&](auto QI) {
            constexpr int qi = decltype(QI)::value;
            const f32x4 w = ring.q[qi % KS_D];
            ring.q[qi % KS_D] = ring.p[(qi + KS_D) * 64];
            if constexpr (qi == 0) oc = w;
            else if constexpr (qi <= 8) {
                static_for<0, 4>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    od = mfma_b2<j>(X[qi - 1], w[j], od);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const float o = gsum(od[ii]) + oc[0];
            if (ii < nv && wq == 0) {
                const int f = __builtin_amdgcn_readlane(nid[0], ii) - hp.node_base;
                if (g == 0 && u < hp.pharm_nf) hp.eps_h[(size_t)f * hp.pharm_nf + u] = o;
                if (u == 0 && g < 3) hp.eps_x[(size_t)f * 3 + g] = Vd[ii];     // output channel 0, coordinate g
            }
        }
    }
}

}  // namespace

extern "C" {
#ifdef PF_KS_STAMPS
int pfk_ks_set_stamp_buffer(unsigned long long* dev) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ks_stamps), &dev, sizeof(dev)); }
#endif
void pfk_rgk_edge(const EdgeParams* p, const EncodeParams* enc, int layer0, int hoist, hipStream_t s) {
    if (p->ntiles == 0) return;
    const int grid = (p->nreg > 0 ? p->ngroups4 : p->ntiles * 8) + p->warm_wgs;
    if (grid == p->warm_wgs) return;
    const EncodeParams noenc{};
    const EncodeParams& ep = (layer0 && enc) ? *enc : noenc;      // layer 0: encode the gathered rows on the fly
    if (!layer0) hipLaunchKernelGGL((k_rgk_edge<false, false>), dim3(grid), dim3(256), 0, s, *p, ep);
    else if (hoist && p->zs) hipLaunchKernelGGL((k_rgk_edge<true, true>), dim3(grid), dim3(256), 0, s, *p, ep);
    else hipLaunchKernelGGL((k_rgk_edge<true, false>), dim3(grid), dim3(256), 0, s, *p, ep);
}
void pfk_rgk_node(const NodeParams* p, const HeadParams* hp, const EncodeParams* enc, int layer0, hipStream_t s) {
    if (p->ntiles == 0) return;
    const HeadParams none{};
    const bool head = hp != nullptr;
    const HeadParams& h = head ? *hp : none;
    const int grid = p->ntiles * 8 + p->warm_wgs;
    const EncodeParams noenc{};
    const EncodeParams& ep = (layer0 && enc) ? *enc : noenc;
    if (head) { if (layer0) hipLaunchKernelGGL((k_rgk_node<true, true>), dim3(grid), dim3(256), 0, s, *p, h, ep);
                else hipLaunchKernelGGL((k_rgk_node<false, true>), dim3(grid), dim3(256), 0, s, *p, h, ep); }
    else { if (layer0) hipLaunchKernelGGL((k_rgk_node<true, false>), dim3(grid), dim3(256), 0, s, *p, h, ep);
           else hipLaunchKernelGGL((k_rgk_node<false, false>), dim3(grid), dim3(256), 0, s, *p, h, ep); }
}
}
