// pf_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the PharmacoForge denoising path.
//
// Reference being restated (paths relative to the reference root):
//   GVP.forward                      pharmacoforge/models/gvp.py:89-116
//   GVPLayerNorm.forward             pharmacoforge/models/gvp.py:159-166
//   GVPMultiEdgeConv.forward/message pharmacoforge/models/gvp.py:459-551
//   PharmRecDynamicsGVP.forward      pharmacoforge/models/dynamics_gvp.py:131-185
//   add_pharm_edges                  pharmacoforge/models/dynamics_gvp.py:187-227
//   NoisePredictionBlock.forward     pharmacoforge/models/dynamics_gvp.py:37-42
//   sample_p_zs_given_zt/com_removal pharmacoforge/models/pharmacodiff.py:380-431, 88-108
//
// Every dense layer runs on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32) with the
// row (edge / node) on the lane, see pf_device.h "F-layout".  No LDS is needed for activations;
// weights stream from L2 in pre-packed fragment order.
#include <hip/hip_runtime.h>
#include "pf_device.h"
#include "pf_train.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef PF_CH
#define PF_CH 4
#endif
#ifndef PF_WPS_EDGE
#define PF_WPS_EDGE 2
#endif
#ifndef PF_WPS_NODE
#define PF_WPS_NODE 1
#endif
#ifndef PF_WPS_HEAD
#define PF_WPS_HEAD 1
#endif
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// bf16 training leg (BASELINE config 5; no reference counterpart: the reference trains in fp32): the two dense Linears of a
// GVP -- to_feats_out and scalar_to_vector_gates -- on v_mfma_f32_32x32x16_bf16, operands rounded to bf16 (round to nearest
// even, v_cvt_pk_bf16_f32), fp32 accumulation.  One instruction covers eight f32 k-steps: lane half hl supplies the eight
// k-values it would have fed to k-steps 8 kk .. 8 kk + 7 (A and B use the same assignment of K to (half, element), which is all
// a dot product needs); the C/D layout is that of the f32 instruction, so the F-layout chain is unchanged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ bf16x8 bf_pack8(const float (&x)[8]) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)x[i];
    return r;
}

// 1-ulp hardware reciprocal / sqrt / rsqrt (v_rcp_f32, v_sqrt_f32, v_rsq_f32): the IEEE-exact
// expansions cost 10-15 VALU instructions each and parity is judged at 2e-4, not at 1 ulp.
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrtf_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float rsqf_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * rcpf_(1.0f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------
// One GVP (gvp.py:89-116) on a 32-row tile, row-on-lane, scalar AND vector channel on the matrix
// cores.
//   Scalars: F-layout (pf_device.h): 64 registers per lane, reg 16*mt+r <-> feature 32*mt+rho(r,hl).
//   Vectors: "R-layout": channel u(t,hl) = (t&3) + 8*(t>>2) + 4*hl lives in register t (t = 0..7) of
//   lane half hl, one register array per coordinate c -- i.e. rows 0..15 of a 32x32 C/D fragment.
//   The two lanes of a row hold DISJOINT halves of both channels; all cross-lane mixing is done by
//   the MFMAs themselves (no shuffles, no redundant VALU work):
//       Vh^T[hh][row,c] = Wh^T[hh][v] V^T[v][row,c]     8 (+1) k-steps x 3 coordinates
//       Vu^T[u][row,c]  = Wu^T[u][hh] Vh^T[hh][row,c]   B operand = the Vh accumulator registers
//       sh[hh] = |Vh[hh]| enters the scalar Linear as extra k-steps, gates come out of a 16-row tile
//   VI     input vector channels (17 for the first message GVP: channel 0 = unit x_diff; else 16)
//   NEXTRA extra scalar inputs after the 128 features (16 rbf values for the first message GVP)
//   VO     output vector channels (16, or 1 for the last noise-head GVP);  NMO output tiles of 32
//   SIG    sigmoid vector activation (identity for the last noise-head GVP)
//   VROW0  only the extra channel (x_diff) is non-zero (conv layer 0: node vectors are zero,
//          dynamics_gvp.py:162-173)
// ---------------------------------------------------------------------------------------------
//   PRE    the 128-feature block of the scalar Linear (+ bias) was already applied per SOURCE NODE
//          (k_encode_build: P = W[:, :128] h + b); the accumulators start from the gathered row of P and
//          only the rbf / sh k-steps remain  (E/N ~ 7x fewer MFMAs for that block on pp edges)
__device__ __forceinline__ void store_vec_r(float* row, const int hl, const float (&V)[3][8]);
//   SAVE   training forward: the pre-activation scalars Z, the gate pre-activations and the gated output vectors of
//          this row are also written to sv_z [128] / sv_g [16] / sv_v [48] (what the level-by-level backward reads)
//   BF16   to_feats_out and the gate Linear on bf16 matrix instructions (training only: pf_train_set_precision)
template <int VI, int NEXTRA, int VO, int NMO, bool SIG, bool VROW0, bool PRE = false, bool SAVE = false, bool BF16 = false>
__device__ __forceinline__ void gvp_apply(const GvpW w, const float (&s_in)[64], const float* ext,
                                          const float (&Vr)[3][8], const float* xhat,
                                          float (&s_out)[NMO * 16], float (&V_out)[3][8], const int lane,
                                          pf_gcf pre_row = nullptr, float* sv_z = nullptr, float* sv_g = nullptr,
                                          float* sv_v = nullptr) {
    constexpr bool X = (VI == 17);                   // extra (17th) channel present
    constexpr int NVK = 8 + (X ? 1 : 0);             // k-steps of the Vh / Vu products and of the sh block
    constexpr int NKS = 64 + NEXTRA / 2 + NVK;       // k-steps of to_feats_out
    constexpr int KS0 = PRE ? 64 : 0;                // first k-step still to do
    constexpr int CH = PF_CH;                        // k-steps per software-pipeline chunk
    constexpr int NCH = (NKS - KS0 + CH - 1) / CH;
    const int hl = lane >> 5;
    // ---- vector channel on the matrix cores                                     (gvp.py:96-99)
    f32x16 vh[3], vu[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { vh[c][r] = 0.f; vu[c][r] = 0.f; }
    {
        pf_gcf ap = w.a_wh + lane;
        float a[NVK];
#pragma unroll
        for (int t = 0; t < NVK; ++t) a[t] = ap[t * 64];
#pragma unroll
        for (int t = VROW0 ? 8 : 0; t < NVK; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) vh[c] = MFMA(a[t], t < 8 ? Vr[c][t < 8 ? t : 0] : xhat[c], vh[c]);
    }
    {
        pf_gcf ap = w.a_wu + lane;
        float a[NVK];
#pragma unroll
        for (int t = 0; t < NVK; ++t) a[t] = ap[t * 64];
#pragma unroll
        for (int t = 0; t < NVK; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) vu[c] = MFMA(a[t], vh[c][t], vu[c]);
    }
    // sh[u(t,hl)] = |Vh| for this lane's channels (k-step t of the sh block)
    float shsel[NVK];
#pragma unroll
    for (int t = 0; t < NVK; ++t)
        shsel[t] = sqrtf_(fmaxf(vh[0][t] * vh[0][t] + vh[1][t] * vh[1][t] + vh[2][t] * vh[2][t], 1e-8f));
    float Vu[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 8; ++t) Vu[c][t] = vu[c][t];
    // ---- scalar channel: feats_out = SiLU(W [s, sh] + b) on the matrix cores  (gvp.py:101-103)
    f32x16 acc[NMO];
    {
        // bias in F-layout -- or, with PRE, the gathered row of P (row-major, so the F-layout pattern applies)
        const f32x4 PF_AS1* bp = PRE ? reinterpret_cast<const f32x4 PF_AS1*>(pre_row + 4 * hl)
                                     : reinterpret_cast<const f32x4 PF_AS1*>(w.b_main + hl * (NMO * 16));
#pragma unroll
        for (int mo = 0; mo < NMO; ++mo)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = PRE ? bp[mo * 8 + q * 2] : bp[mo * 4 + q];
                acc[mo][4 * q + 0] = b4[0]; acc[mo][4 * q + 1] = b4[1];
                acc[mo][4 * q + 2] = b4[2]; acc[mo][4 * q + 3] = b4[3];
            }
    }
    typedef float fragA __attribute__((ext_vector_type(NMO)));
    const fragA PF_AS1* ap = reinterpret_cast<const fragA PF_AS1*>(w.a_main) + KS0 * 64 + lane;
    if constexpr (BF16) {
        // eight f32 k-steps per instruction; the f32 fragments are rounded as they arrive (the packed table is shared with the
        // f32 kernels: no second copy of the weights to keep current after every optimiser step)
        constexpr int NKI = (NKS - KS0 + 7) / 8;
#pragma unroll
        for (int kk = 0; kk < NKI; ++kk) {
            fragA af[8];
            float b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ks = KS0 + 8 * kk + i;
                if (ks < NKS) af[i] = ap[(8 * kk + i) * 64];
                else {
#pragma unroll
                    for (int mo = 0; mo < NMO; ++mo) af[i][mo] = 0.f;
                }
                if (ks < 64) b[i] = s_in[ks < 64 ? ks : 0];
                else if (ks < 64 + NEXTRA / 2) b[i] = ext[(ks - 64) < (NEXTRA / 2 > 0 ? NEXTRA / 2 : 1) ? ks - 64 : 0];
                else if (ks < NKS) b[i] = shsel[(ks - 64 - NEXTRA / 2) < NVK ? (ks - 64 - NEXTRA / 2) : 0];
                else b[i] = 0.f;
            }
            const bf16x8 bb = bf_pack8(b);
#pragma unroll
            for (int mo = 0; mo < NMO; ++mo) {
                float a8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) a8[i] = af[i][mo];
                acc[mo] = MFMA_BF(bf_pack8(a8), bb, acc[mo]);
            }
        }
    }
    fragA abuf[2][CH];
#pragma unroll
    for (int i = 0; i < CH; ++i)
        if (!BF16 && KS0 + i < NKS) abuf[0][i] = ap[i * 64];
#pragma unroll
    for (int c = 0; c < (BF16 ? 0 : NCH); ++c) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (KS0 + (c + 1) * CH + i < NKS) abuf[(c + 1) & 1][i] = ap[((c + 1) * CH + i) * 64];
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int ks = KS0 + c * CH + i;
            if (ks < NKS) {
                float b;
                if (ks < 64) b = s_in[ks < 64 ? ks : 0];
                else if (ks < 64 + NEXTRA / 2) b = ext[ks - 64];
                else b = shsel[(ks - 64 - NEXTRA / 2) < NVK ? (ks - 64 - NEXTRA / 2) : 0];
                const fragA a = abuf[c & 1][i];
#pragma unroll
                for (int mo = 0; mo < NMO; ++mo) acc[mo] = MFMA(a[mo], b, acc[mo]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SAVE) {
        static_assert(NMO == 4, "SAVE stores full 128-feature rows");
        f32x4* zp = reinterpret_cast<f32x4*>(sv_z + 4 * hl);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 x;
                x[0] = acc[mt][4 * q + 0]; x[1] = acc[mt][4 * q + 1]; x[2] = acc[mt][4 * q + 2]; x[3] = acc[mt][4 * q + 3];
                zp[mt * 8 + q * 2] = x;
            }
    }
    // ---- SiLU feeds the gate MFMAs just in time:  gate = Wg feats_out + bg      (gvp.py:105-111)
    constexpr int NG = NMO * 16;
    constexpr int LOOK = 4;
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = 0.f;
    if constexpr (BF16) {
        pf_gcf gp = w.a_gate + lane;
#pragma unroll
        for (int kk = 0; kk < NG / 8; ++kk) {
            float a8[8], b8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a8[i] = gp[(8 * kk + i) * 64];
                b8[i] = siluf_(acc[(8 * kk + i) / 16][(8 * kk + i) % 16]);
                s_out[8 * kk + i] = b8[i];
            }
            g = MFMA_BF(bf_pack8(a8), bf_pack8(b8), g);
        }
    } else {
#pragma unroll
    for (int q = 0; q < LOOK; ++q) s_out[q] = siluf_(acc[q / 16][q % 16]);
    }
    if constexpr (!BF16) {
        pf_gcf gp = w.a_gate + lane;
        float gbuf[2][CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) gbuf[0][i] = gp[i * 64];
#pragma unroll
        for (int c = 0; c < NG / CH; ++c) {
            if (c + 1 < NG / CH) {
#pragma unroll
                for (int i = 0; i < CH; ++i) gbuf[(c + 1) & 1][i] = gp[((c + 1) * CH + i) * 64];
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int ks = c * CH + i;
                g = MFMA(gbuf[c & 1][i], s_out[ks], g);
                if (ks + LOOK < NG) s_out[ks + LOOK] = siluf_(acc[(ks + LOOK) / 16][(ks + LOOK) % 16]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // V_out = act(gate) * Vu on this lane's channels (gate rows 0..15 of the tile = registers 0..7)
    {
        const f32x4 PF_AS1* bg = reinterpret_cast<const f32x4 PF_AS1*>(w.b_gate + hl * 8);
        const f32x4 b0 = bg[0], b1 = bg[1];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (VO == 1 && t > 0) {
                V_out[0][t] = 0.f; V_out[1][t] = 0.f; V_out[2][t] = 0.f;
                continue;
            }
            float gv = g[t] + (t < 4 ? b0[t & 3] : b1[t & 3]);
            if constexpr (SAVE) sv_g[(t & 3) + 8 * (t >> 2) + 4 * hl] = gv;
            if constexpr (SIG) gv = sigmoidf_(gv);
            V_out[0][t] = gv * Vu[0][t];
            V_out[1][t] = gv * Vu[1][t];
            V_out[2][t] = gv * Vu[2][t];
        }
    }
    if constexpr (SAVE) store_vec_r(sv_v, hl, V_out);
}

// load / store a 128-float row in F-layout (this lane's 64 features)
template <typename P>
__device__ __forceinline__ void load_row_f(P row, const int hl, float (&s)[64]) {
    auto p = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)row + 4 * hl);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 x = p[mt * 8 + q * 2];
            s[mt * 16 + 4 * q + 0] = x[0]; s[mt * 16 + 4 * q + 1] = x[1];
            s[mt * 16 + 4 * q + 2] = x[2]; s[mt * 16 + 4 * q + 3] = x[3];
        }
}
__device__ __forceinline__ void store_row_f(float* row, const int hl, const float (&s)[64]) {
    f32x4* p = reinterpret_cast<f32x4*>(row + 4 * hl);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 x;
            x[0] = s[mt * 16 + 4 * q + 0]; x[1] = s[mt * 16 + 4 * q + 1];
            x[2] = s[mt * 16 + 4 * q + 2]; x[3] = s[mt * 16 + 4 * q + 3];
            p[mt * 8 + q * 2] = x;
        }
}
// load / store this lane's 8 channels (R-layout) of a [16][3] vector row: channels 8q+4hl .. +3 are
// 12 contiguous floats at offset 24q + 12hl
template <typename P>
__device__ __forceinline__ void load_vec_r(P row, const int hl, float (&V)[3][8]) {
    auto p = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)row + 12 * hl);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 a = p[6 * q], b = p[6 * q + 1], d = p[6 * q + 2];
        const float f[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], d[0], d[1], d[2], d[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) V[c][4 * q + i] = f[3 * i + c];
    }
}
__device__ __forceinline__ void store_vec_r(float* row, const int hl, const float (&V)[3][8]) {
    f32x4* p = reinterpret_cast<f32x4*>(row + 12 * hl);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float f[12];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) f[3 * i + c] = V[c][4 * q + i];
        f32x4 a, b, d;
        a[0] = f[0]; a[1] = f[1]; a[2] = f[2]; a[3] = f[3];
        b[0] = f[4]; b[1] = f[5]; b[2] = f[6]; b[3] = f[7];
        d[0] = f[8]; d[1] = f[9]; d[2] = f[10]; d[3] = f[11];
        p[6 * q] = a; p[6 * q + 1] = b; p[6 * q + 2] = d;
    }
}

// ---------------------------------------------------------------------------------------------
// In-tile segmented reduction of the per-edge messages (replaces scatter-add): the edge slots of a
// tile are sorted by destination, so the rows of one destination are consecutive lanes.  A Hillis-
// Steele segmented scan over the 32 rows of each half-wave leaves, in the LAST lane of every
// (tile, destination) segment, the sum of that segment in edge order; only those tail lanes store a
// message row (at their own edge slot).  A destination whose in-edges straddle tile boundaries gets
// one partial row per tile; the node kernels add them in slot order (deterministic).  This cuts the
// message traffic ~5x (mean in-degree 6.8 on pp edges).
// ---------------------------------------------------------------------------------------------
// DPP form: four in-row steps (row_shr 1,2,4,8 inside each 16-lane row) plus one carry from lane 15 of the
// previous row (row_bcast15) for the odd rows -- one v_fmac with a DPP operand per step instead of a
// ds_bpermute + select + add.  The 0/1 masks are shared by all 88 values of a row.
struct SegMask { float m1, m2, m4, m8, mc; };
template <int CTRL>
__device__ __forceinline__ float dpp_f(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(const int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ SegMask seg_masks(const int key, const int j) {
    const int jr = j & 15;
    // every DPP read runs with all lanes enabled: a cross-lane read from a lane that a branch has disabled
    // returns 0, so the comparisons below must not be short-circuited into divergent code
    const int k1 = dpp_i<0x111>(key), k2 = dpp_i<0x112>(key), k4 = dpp_i<0x114>(key), k8 = dpp_i<0x118>(key);
    const int kc = dpp_i<0x142>(key);                 // row_bcast15: lane 15 of the previous row
    SegMask m;
    m.m1 = ((jr >= 1) & (k1 == key)) ? 1.f : 0.f;
    m.m2 = ((jr >= 2) & (k2 == key)) ? 1.f : 0.f;
    m.m4 = ((jr >= 4) & (k4 == key)) ? 1.f : 0.f;
    m.m8 = ((jr >= 8) & (k8 == key)) ? 1.f : 0.f;
    m.mc = (((j & 16) != 0) & (kc == key)) ? 1.f : 0.f;
    return m;
}
__device__ __forceinline__ float seg_scan32(float v, const SegMask& m) {
    v = fmaf(dpp_f<0x111>(v), m.m1, v);
    v = fmaf(dpp_f<0x112>(v), m.m2, v);
    v = fmaf(dpp_f<0x114>(v), m.m4, v);
    v = fmaf(dpp_f<0x118>(v), m.m8, v);
    v = fmaf(dpp_f<0x142>(v), m.mc, v);
    return v;
}
// tail of the segment that contains edge slot e, for a destination whose slots are [st, st+c): tiles are
// aligned to multiples of 32 slots
__device__ __forceinline__ int seg_tail(const int e, const int end) { return min(e | 31, end - 1); }

// ---------------------------------------------------------------------------------------------
// Edge messages: gather source rows -> x_diff / distance / RBF -> message GVP chain -> per-edge
// message rows (gvp.py:472-485, 540-551).  One wave per tile of 32 edge slots, all etypes in one
// launch.  L0: conv layer 0, node vectors are identically zero.
// ---------------------------------------------------------------------------------------------
template <bool L0, bool SAVE = false, bool BF16 = false>
__global__ __launch_bounds__(256, PF_WPS_EDGE) void k_edge_msg(const EdgeParams p) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    if (wid >= p.ntiles) return;
    const EdgeTile t = p.tiles[wid];
    int nvalid = t.n;
    if (t.cnt_idx >= 0) {
        const int c = p.dyn_cnt[t.cnt_idx] - t.rel;
        nvalid = min(nvalid, max(c, 0));
    }
    nvalid = __builtin_amdgcn_readfirstlane(nvalid);
    if (nvalid <= 0) return;
    const int et = __builtin_amdgcn_readfirstlane(t.et);
    const int j = lane & 31, hl = lane >> 5;
    const int e = t.e0 + min(j, nvalid - 1);          // idle lanes shadow the last valid edge
    const int src = p.esrc[e], dst = p.edst[e];
    const float4 xs = p.xn[src], xd = p.xn[dst];
    // x_diff = x_src - x_dst ; d = sqrt(max(|x_diff|^2, 1e-8)) + 1e-8 ; unit vector ; rbf
    const float dx = xs.x - xd.x, dy = xs.y - xd.y, dz = xs.z - xd.z;
    const float d = sqrtf_(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
    const float rd = rcpf_(d);
    const float xhat[3] = {dx * rd, dy * rd, dz * rd};
    // this lane feeds rbf[2t + hl] into k-step t of the rbf block
    float rb[PF_R / 2];
#pragma unroll
    for (int k = 0; k < PF_R / 2; ++k) {
        const float ze = (d - p.rbf_mu[2 * k]) * p.rbf_inv_sigma;
        const float zo = (d - p.rbf_mu[2 * k + 1]) * p.rbf_inv_sigma;
        const float re = __expf(-(ze * ze)), ro = __expf(-(zo * zo));
        rb[k] = hl ? ro : re;
    }
    float V[3][8];
    if constexpr (!L0) load_vec_r(p.v + (size_t)src * 48, hl, V);
    else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) V[c][q] = 0.f;
    }
    const GvpW PF_AS1* wt = (const GvpW PF_AS1*)p.w + et * p.n_gvps;
    float s1[64], V1[3][8];
    // SAVE: per-level rows of this edge slot (level-major: [level][slot])
    float* svz = SAVE ? p.sv_z + (size_t)e * PF_S : nullptr;
    float* svg = SAVE ? p.sv_g + (size_t)e * 16 : nullptr;
    float* svv = SAVE ? p.sv_v + (size_t)e * 48 : nullptr;
    if (et == ET_PP && p.pre != nullptr) {
        // pp edges: W[:, :128] h_src + b was applied once per source node; gather that row into the accumulators
        float s[64];
#pragma unroll
        for (int q = 0; q < 64; ++q) s[q] = 0.f;
        gvp_apply<17, PF_R, 16, 4, true, L0, true, SAVE, BF16>(wt[0], s, rb, V, xhat, s1, V1, lane, (pf_gcf)p.pre + (size_t)src * PF_S, svz, svg, svv);
    } else {
        float s[64];
        load_row_f(p.h + (size_t)src * PF_S, hl, s);
        gvp_apply<17, PF_R, 16, 4, true, L0, false, SAVE, BF16>(wt[0], s, rb, V, xhat, s1, V1, lane, nullptr, svz, svg, svv);
    }
    for (int gi = 1; gi < p.n_gvps; ++gi) {
        float s2[64], V2[3][8];
        if constexpr (SAVE) { svz += p.sv_stride * PF_S; svg += p.sv_stride * 16; svv += p.sv_stride * 48; }
        gvp_apply<16, 0, 16, 4, true, false, false, SAVE, BF16>(wt[gi], s1, nullptr, V1, nullptr, s2, V2, lane, nullptr, svz, svg, svv);
#pragma unroll
        for (int q = 0; q < 64; ++q) s1[q] = s2[q];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) V1[c][q] = V2[c][q];
    }
    // segmented reduction over the rows of the tile; idle lanes (shadow copies) carry zeros and private keys
    const bool valid = j < nvalid;
    const int key = valid ? dst : -1 - j;
    const SegMask sm = seg_masks(key, j);
#pragma unroll
    for (int q = 0; q < 64; ++q) s1[q] = seg_scan32(valid ? s1[q] : 0.f, sm);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) V1[c][q] = seg_scan32(valid ? V1[c][q] : 0.f, sm);
    const int knext = __shfl_down(key, 1, 32);
    if (valid && (j == 31 || knext != key)) {          // tail lane of a (tile, destination) segment
        store_row_f(p.msg_s + (size_t)e * PF_S, hl, s1);
        store_vec_r(p.msg_v + (size_t)e * 48, hl, V1);
    }
}

// ---------------------------------------------------------------------------------------------
// Node update: deterministic segmented reduction of the in-edge messages (mean / sum per etype,
// summed over etypes), residual, GVPLayerNorm, update GVP chain, residual, GVPLayerNorm
// (gvp.py:488-536).  One wave per tile of 32 nodes of one type.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void gvp_layernorm(pf_gcf lw, pf_gcf lb, const int hl, float (&s)[64],
                                              float (&V)[3][8]) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) sum += s[q];
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / 128.0f);
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) { const float c = s[q] - mean; var = fmaf(c, c, var); }
    var += __shfl_xor(var, 32);
    const float rstd = rsqf_(var * (1.0f / 128.0f) + 1e-5f);
    float w[64], b[64];
    load_row_f(lw, hl, w);
    load_row_f(lb, hl, b);
#pragma unroll
    for (int q = 0; q < 64; ++q) s[q] = (s[q] - mean) * rstd * w[q] + b[q];
    float vn = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) vn += fmaxf(V[0][t] * V[0][t] + V[1][t] * V[1][t] + V[2][t] * V[2][t], 1e-8f);
    vn += __shfl_xor(vn, 32);
    const float rden = rcpf_(sqrtf_(vn * (1.0f / 16.0f) + 1e-5f) + 1e-5f);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 8; ++t) V[c][t] = V[c][t] * rden;
}

// GVPDropout (gvp.py:118-149) of one node's scalar row (F-layout) and vector channels (R-layout): the keep mask is a
// counter-based hash of (step seed, stream, node, element) that the backward pass regenerates (pf_train.h)
__device__ __forceinline__ void node_dropout(const NodeParams& p, const uint32_t stream, const int n, const int hl,
                                             float (&s)[64], float (&V)[3][8]) {
    const uint32_t base = (uint32_t)n * 144u;
    const float* ov = p.mask_override ? p.mask_override + (size_t)stream * p.N * 144u : nullptr;
#pragma unroll
    for (int q = 0; q < 64; ++q) {
        const uint32_t f = 32u * (q >> 4) + ((q & 3) + 8 * ((q & 15) >> 2) + 4 * hl);
        s[q] *= ov ? ov[base + f] : (pf_drop_hash(p.seed, stream, base + f) < p.drop_thr ? 0.0f : p.drop_scale);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const uint32_t ch = (t & 3) + 8 * (t >> 2) + 4 * hl;
        const float m = ov ? ov[base + 128u + ch] : (pf_drop_hash(p.seed, stream, base + 128u + ch) < p.drop_thr ? 0.0f : p.drop_scale);
        V[0][t] *= m; V[1][t] *= m; V[2][t] *= m;
    }
}

template <bool L0>
__global__ __launch_bounds__(64, PF_WPS_NODE) void k_node_update(const NodeParams p) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x;                       // one wave per block: the few tiles spread over all CUs
    if (wid >= p.ntiles) return;
    const NodeTile t = p.tiles[wid];
    const int nt = __builtin_amdgcn_readfirstlane(t.ntype);
    const int j = lane & 31, hl = lane >> 5;
    int tn = t.n;                                     // rows of a list tile (active atoms): what the build kernel counted
    if (t.cnt_idx >= 0) tn = min(tn, max(p.dyn_cnt[t.cnt_idx] - t.rel, 0));
    tn = __builtin_amdgcn_readfirstlane(tn);
    if (tn <= 0) return;
    const bool live = j < tn;
    const int n = t.ids ? p.row_ids[t.n0 + min(j, tn - 1)] : t.n0 + min(j, tn - 1);
    float ms[64], mv[3][8];
#pragma unroll
    for (int q = 0; q < 64; ++q) ms[q] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) mv[c][q] = 0.f;
    for (int si = 0; si < 2; ++si) {
        const int slot = si == 0 ? 0 : (nt == 0 ? p.pp_slot : 1);
        const int st = p.in_start[slot * p.N + n];
        const int c = live ? p.in_cnt[slot * p.N + n] : 0;
        int cmax = c;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
        cmax = __builtin_amdgcn_readfirstlane(cmax);
        // fn.mean: scale by 1/in-degree (zero in-degree -> 0); fn.sum: scale 1.  Rows = per-tile partial sums.
        const float sc = (p.norm_mode == 0 && c > 0) ? 1.0f / (float)c : 1.0f;
        const int npart = c > 0 ? ((st + c - 1) >> 5) - (st >> 5) + 1 : 0;
        int pmax = npart;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) pmax = max(pmax, __shfl_xor(pmax, o));
        pmax = __builtin_amdgcn_readfirstlane(pmax);
        (void)cmax;
        int ecur = st;
        for (int i = 0; i < pmax; ++i) {
            if (i < npart) {
                const int row = seg_tail(ecur, st + c);
                ecur = row + 1;
                float r[64], rv[3][8];
                load_row_f(p.msg_s + (size_t)row * PF_S, hl, r);
                load_vec_r(p.msg_v + (size_t)row * 48, hl, rv);
#pragma unroll
                for (int q = 0; q < 64; ++q) ms[q] = fmaf(r[q], sc, ms[q]);
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
#pragma unroll
                    for (int q = 0; q < 8; ++q) mv[cc][q] = fmaf(rv[cc][q], sc, mv[cc][q]);
            }
        }
    }
    float inv_norm = 1.0f;
    if (p.norm_mode == 1) inv_norm = 1.0f / p.norm_value;
    else if (p.norm_mode == 2) inv_norm = 1.0f / p.gnorm[nt * p.B + p.gid[n]];
    float s[64], V[3][8];
    load_row_f(p.h_in + (size_t)n * PF_S, hl, s);
    if constexpr (!L0) load_vec_r(p.v_in + (size_t)n * 48, hl, V);
    else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) V[c][q] = 0.f;
    }
    if (p.drop_thr != 0u || p.mask_override != nullptr) node_dropout(p, (uint32_t)p.layer * 2u, n, hl, ms, mv);      // gvp.py:518 (training forward)
#pragma unroll
    for (int q = 0; q < 64; ++q) s[q] = fmaf(ms[q], inv_norm, s[q]);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) V[c][q] = fmaf(mv[c][q], inv_norm, V[c][q]);
    const NodeW nw = p.w[nt];
    gvp_layernorm(nw.ln1_w, nw.ln1_b, hl, s, V);
    float s1[64], V1[3][8];
#pragma unroll
    for (int q = 0; q < 64; ++q) s1[q] = s[q];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) V1[c][q] = V[c][q];
    for (int gi = 0; gi < p.n_upd; ++gi) {
        float s2[64], V2[3][8];
        gvp_apply<16, 0, 16, 4, true, false>(nw.upd[gi], s1, nullptr, V1, nullptr, s2, V2, lane);
#pragma unroll
        for (int q = 0; q < 64; ++q) s1[q] = s2[q];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) V1[c][q] = V2[c][q];
    }
    if (p.drop_thr != 0u || p.mask_override != nullptr) node_dropout(p, (uint32_t)p.layer * 2u + 1u, n, hl, s1, V1);      // gvp.py:529
#pragma unroll
    for (int q = 0; q < 64; ++q) s[q] += s1[q];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) V[c][q] += V1[c][q];
    gvp_layernorm(nw.ln2_w, nw.ln2_b, hl, s, V);
    if (live) {
        store_row_f(p.h_out + (size_t)n * PF_S, hl, s);
        store_vec_r(p.v_out + (size_t)n * 48, hl, V);
    }
}

// ---------------------------------------------------------------------------------------------
// Noise head on the pharmacophore nodes (dynamics_gvp.py:37-42).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64, PF_WPS_HEAD) void k_noise_head(const HeadParams p) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x;
    if (wid >= p.ntiles) return;
    const NodeTile t = p.tiles[wid];
    const int j = lane & 31, hl = lane >> 5;
    const bool live = j < t.n;
    const int n = t.n0 + min(j, t.n - 1);
    float s1[64], V1[3][8];
    load_row_f(p.h + (size_t)n * PF_S, hl, s1);
    load_vec_r(p.v + (size_t)n * 48, hl, V1);
    for (int gi = 0; gi + 1 < p.n_gvps; ++gi) {
        float s2[64], V2[3][8];
        gvp_apply<16, 0, 16, 4, true, false>(p.gvps[gi], s1, nullptr, V1, nullptr, s2, V2, lane);
#pragma unroll
        for (int q = 0; q < 64; ++q) s1[q] = s2[q];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) V1[c][q] = V2[c][q];
    }
    float so[32], Vo[3][8];
    gvp_apply<16, 0, 1, 2, false, false>(p.gvps[p.n_gvps - 1], s1, nullptr, V1, nullptr, so, Vo, lane);
    // to_scalar_output: Linear(64 -> pharm_nf), rows 0..5 of a 32-row tile
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) o = MFMA(p.a_out[ks * 64 + lane], so[ks], o);
    if (live) {
        const int f = n - p.node_base;
        // lane half 0 holds output rows 0-3 (regs 0-3) and 8-11 (regs 4-7); half 1 rows 4-7, 12-15
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int u = (r & 3) + 8 * (r >> 2) + 4 * hl;
            if (u < p.pharm_nf) p.eps_h[(size_t)f * p.pharm_nf + u] = o[r] + p.b_out[u];
        }
        if (hl == 0) {                                  // output vector channel 0 = register 0 of lane half 0
            p.eps_x[(size_t)f * 3 + 0] = Vo[0][0];
            p.eps_x[(size_t)f * 3 + 1] = Vo[1][0];
            p.eps_x[(size_t)f * 3 + 2] = Vo[2][0];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Cooperative (4-wave) form of the same GVP for launches with few tiles (last-layer edges, node
// update, noise head): one 256-thread block per 32-row tile.  The 128 outputs of the scalar Linear
// are split over the waves (wave w owns output tile w), the three coordinates of the vector channel
// over waves 0..2, the K dimension of the gate Linear over the waves; partial results meet in LDS
// (two barriers per GVP).  Per wave a generic GVP is 104 MFMAs instead of 400, so the serial latency
// of a chain -- what these launches are bound by -- drops ~3.5x, and each wave streams only a quarter
// of the weights.  All waves keep a full copy of the scalar state (F-layout); wave c keeps
// coordinate c of the vector state (R-layout, 8 registers).
// ---------------------------------------------------------------------------------------------
// In-kernel cycle stamps (diagnostic builds only: -DPF_STAMPS; no stamp executes in the product build).
#ifdef PF_STAMPS
__device__ unsigned long long* g_pf_stamps = nullptr;
#define PF_STAMP(L) pf_stamp(L, lane, wv)
#else
#define PF_STAMP(L)
#endif
struct __attribute__((aligned(16))) CoopLds {
    float vh[3][9][64];     // Vh of coordinate c, k-step t          (for sh = |Vh|)
    float so[4][16][64];    // SiLU output tile of wave w             (next layer's input)
    float pg[4][8][64];     // partial gates of wave w
    float vx[3][8][64];     // vector output of coordinate c          (stores / norms)
    float agg[32][177];     // aggregated messages per node (stride 177: conflict-free column reads)
#ifdef PF_STAMPS
    int scnt[4];
#endif
};
#ifdef PF_STAMPS
__device__ __forceinline__ void pf_stamp(CoopLds& L, const int lane, const int wv) {
    if (lane == 0 && g_pf_stamps) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        const int k = L.scnt[wv]++;
        if (k < 64) g_pf_stamps[((size_t)blockIdx.x * 4 + wv) * 64 + k] = t;
    }
}
#endif

// One wave's share of a GVP's weights, held in registers: 9+9 vector fragments, <=81 fragments of its output
// tile, 16 gate fragments and the biases (~130 VGPRs).  Loading is separated from computing so that a chain
// can fetch GVP g+1 (or overlap the first fetch with the message aggregation) while GVP g runs: these launches
// are bound by the serial latency of the chain, and the fetch comes from the Infinity Cache / HBM.
template <int VI, int NEXTRA>
struct CoopW {
    static constexpr int NVK = 8 + (VI == 17 ? 1 : 0);
    static constexpr int NKS = 64 + NEXTRA / 2 + NVK;
    static constexpr int NKS4 = (NKS + 3) / 4;
    // fragments are packed four k-steps per lane ([k/4][lane][4]): one coalesced 16-byte load per lane and group
    f32x4 awh[3], awu[3], am[NKS4], ag[4];
    f32x4 bm[4], bg[2];
};
template <int VI, int NEXTRA, int NMO>
__device__ __forceinline__ void gvp_coop_load(const GvpW w, const int lane, const int wv, CoopW<VI, NEXTRA>& W) {
    constexpr int NKS4 = CoopW<VI, NEXTRA>::NKS4;
    const int hl = lane >> 5;
    const int wm = wv < NMO ? wv : 0;
    const f32x4 PF_AS1* pwh = reinterpret_cast<const f32x4 PF_AS1*>(w.a_wh_c) + lane;
    const f32x4 PF_AS1* pwu = reinterpret_cast<const f32x4 PF_AS1*>(w.a_wu_c) + lane;
#pragma unroll
    for (int q = 0; q < 3; ++q) { W.awh[q] = pwh[q * 64]; W.awu[q] = pwu[q * 64]; }
    const f32x4 PF_AS1* ap = reinterpret_cast<const f32x4 PF_AS1*>(w.a_main_c) + (size_t)wm * NKS4 * 64 + lane;
#pragma unroll
    for (int q = 0; q < NKS4; ++q) W.am[q] = ap[q * 64];
    const f32x4 PF_AS1* gp = reinterpret_cast<const f32x4 PF_AS1*>(w.a_gate_c) + (size_t)wm * 4 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) W.ag[q] = gp[q * 64];
    const f32x4 PF_AS1* bp = reinterpret_cast<const f32x4 PF_AS1*>(w.b_main + hl * (NMO * 16) + wm * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) W.bm[q] = bp[q];
    const f32x4 PF_AS1* bg = reinterpret_cast<const f32x4 PF_AS1*>(w.b_gate + hl * 8);
    W.bg[0] = bg[0]; W.bg[1] = bg[1];
}

template <int VI, int NEXTRA, int VO, int NMO, bool SIG, bool VROW0>
__device__ __forceinline__ void gvp_coop_compute(const CoopW<VI, NEXTRA>& W, const float (&s_in)[64], const float* ext,
                                                 const float (&Vc)[8], const float xhat_c, float (&s_out)[NMO * 16],
                                                 float (&Vc_out)[8], const int lane, const int wv, CoopLds& L) {
    constexpr int NVK = CoopW<VI, NEXTRA>::NVK;
    constexpr int KS_A = 64 + NEXTRA / 2;            // k-steps that do not need sh
    PF_STAMP(L);      // 0: GVP start
    // (1) vector products of coordinate wv on the matrix cores
    float Vu[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) Vu[t] = 0.f;
    if (wv < 3) {
        f32x16 vh, vu;
#pragma unroll
        for (int r = 0; r < 16; ++r) { vh[r] = 0.f; vu[r] = 0.f; }
#pragma unroll
        for (int t = VROW0 ? 8 : 0; t < NVK; ++t) vh = MFMA(W.awh[t >> 2][t & 3], t < 8 ? Vc[t < 8 ? t : 0] : xhat_c, vh);
#pragma unroll
        for (int t = 0; t < NVK; ++t) L.vh[wv][t][lane] = vh[t];
#pragma unroll
        for (int t = 0; t < NVK; ++t) vu = MFMA(W.awu[t >> 2][t & 3], vh[t], vu);
#pragma unroll
        for (int t = 0; t < 8; ++t) Vu[t] = vu[t];
    }
    PF_STAMP(L);      // 1: vector products issued
    // (2) this wave's output tile of the scalar Linear: the k-steps that do not depend on sh
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[4 * q + 0] = W.bm[q][0]; acc[4 * q + 1] = W.bm[q][1]; acc[4 * q + 2] = W.bm[q][2]; acc[4 * q + 3] = W.bm[q][3];
    }
    if (wv < NMO) {
#pragma unroll
        for (int ks = 0; ks < KS_A; ++ks)
            acc = MFMA(W.am[ks >> 2][ks & 3], ks < 64 ? s_in[ks < 64 ? ks : 0] : ext[ks >= 64 ? ks - 64 : 0], acc);
    }
    PF_STAMP(L);      // 2: main k-steps issued
    __syncthreads();                                  // B1: every Vh is in LDS
    PF_STAMP(L);      // 3: past B1
    // (3) sh = |Vh| for this lane's channels, (4) the sh k-steps, SiLU, partial gates
    if (wv < NMO) {
#pragma unroll
        for (int t = 0; t < NVK; ++t) {
            const float x = L.vh[0][t][lane], y = L.vh[1][t][lane], z = L.vh[2][t][lane];
            acc = MFMA(W.am[(KS_A + t) >> 2][(KS_A + t) & 3], sqrtf_(fmaxf(x * x + y * y + z * z, 1e-8f)), acc);
        }
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
        // all 16 activations first (VALU + one LDS store each), then the 16 gate MFMAs back to back
        float so[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { so[r] = siluf_(acc[r]); L.so[wv][r][lane] = so[r]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) g = MFMA(W.ag[r >> 2][r & 3], so[r], g);
#pragma unroll
        for (int t = 0; t < 8; ++t) L.pg[wv][t][lane] = g[t];
    }
    PF_STAMP(L);      // 4: sh + SiLU + gates issued
    __syncthreads();                                  // B2: output tiles and partial gates are in LDS
    PF_STAMP(L);      // 5: past B2
    // (5) assemble the full scalar output; gate this wave's coordinate
#pragma unroll
    for (int mt = 0; mt < NMO; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_out[mt * 16 + r] = L.so[mt][r][lane];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (VO == 1 && t > 0) { Vc_out[t] = 0.f; continue; }
        float gv = (t < 4 ? W.bg[0][t & 3] : W.bg[1][t & 3]);
#pragma unroll
        for (int ww = 0; ww < NMO; ++ww) gv += L.pg[ww][t][lane];
        if constexpr (SIG) gv = sigmoidf_(gv);
        Vc_out[t] = gv * Vu[t];
    }
}

// a chain of n generic GVPs (16 -> 16 channels, 128 -> 128 scalars) with the next GVP's weights in flight;
// W0 must already hold (or be loading) the weights of gvps[0]
__device__ __forceinline__ void gvp_coop_chain(const GvpW PF_AS1* gvps, const int n, CoopW<16, 0>& W0, float (&s1)[64],
                                               float (&V1)[8], const int lane, const int wv, CoopLds& L) {
    CoopW<16, 0> W1;
    for (int gi = 0; gi < n; gi += 2) {
        if (gi + 1 < n) gvp_coop_load<16, 0, 4>(gvps[gi + 1], lane, wv, W1);
        float s2[64], V2[8];
        gvp_coop_compute<16, 0, 16, 4, true, false>(W0, s1, nullptr, V1, 0.f, s2, V2, lane, wv, L);
        if (gi + 1 < n) {
            if (gi + 2 < n) gvp_coop_load<16, 0, 4>(gvps[gi + 2], lane, wv, W0);
            gvp_coop_compute<16, 0, 16, 4, true, false>(W1, s2, nullptr, V2, 0.f, s1, V1, lane, wv, L);
        } else {
#pragma unroll
            for (int q = 0; q < 64; ++q) s1[q] = s2[q];
#pragma unroll
            for (int q = 0; q < 8; ++q) V1[q] = V2[q];
        }
    }
}

// the same chain with ONE weight buffer (no prefetch): half the registers, for kernels that must fit two
// workgroups per CU
__device__ __forceinline__ void gvp_coop_chain1(const GvpW PF_AS1* gvps, const int n, CoopW<16, 0>& W0, float (&s1)[64],
                                                float (&V1)[8], const int lane, const int wv, CoopLds& L) {
    for (int gi = 0; gi < n; ++gi) {
        if (gi > 0) gvp_coop_load<16, 0, 4>(gvps[gi], lane, wv, W0);
        float s2[64], V2[8];
        gvp_coop_compute<16, 0, 16, 4, true, false>(W0, s1, nullptr, V1, 0.f, s2, V2, lane, wv, L);
#pragma unroll
        for (int q = 0; q < 64; ++q) s1[q] = s2[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) V1[q] = V2[q];
    }
}

// this wave's coordinate (wv = 0..2) of the lane's 8 channels of a [16][3] vector row
template <typename P>
__device__ __forceinline__ void load_vec_rc(P row, const int hl, const int wv, float (&Vc)[8]) {
    float V[3][8];
    load_vec_r(row, hl, V);
#pragma unroll
    for (int t = 0; t < 8; ++t) Vc[t] = wv == 0 ? V[0][t] : (wv == 1 ? V[1][t] : V[2][t]);
}

// LOWREG: one weight buffer instead of two (no prefetch of the next GVP while the current one computes) so that the
// kernel fits 256 registers and two workgroups share a CU: for launches with several times more tiles than CUs the
// second workgroup hides the first one's barriers and weight loads.
template <bool L0, bool LOWREG = false>
__device__ __forceinline__ void edge_tile_coop(const EdgeParams& p, const EdgeTile t, CoopLds& L, const int lane, const int wv) {
    int nvalid = t.n;
    if (t.cnt_idx >= 0) {
        const int c = p.dyn_cnt[t.cnt_idx] - t.rel;
        nvalid = min(nvalid, max(c, 0));
    }
    nvalid = __builtin_amdgcn_readfirstlane(nvalid);
    if (nvalid <= 0) return;                           // block-uniform
    const int et = __builtin_amdgcn_readfirstlane(t.et);
    const GvpW PF_AS1* wt = (const GvpW PF_AS1*)p.w + et * p.n_gvps;
    CoopW<17, PF_R> Wfirst;
    gvp_coop_load<17, PF_R, 4>(wt[0], lane, wv, Wfirst);      // in flight under the gather below
    const int j = lane & 31, hl = lane >> 5;
    const int e = t.e0 + min(j, nvalid - 1);
    const int src = p.esrc[e], dst = p.edst[e];
    const float4 xs = p.xn[src], xd = p.xn[dst];
    const float dx = xs.x - xd.x, dy = xs.y - xd.y, dz = xs.z - xd.z;
    const float d = sqrtf_(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
    const float rd = rcpf_(d);
    const float xhat_c = (wv == 0 ? dx : (wv == 1 ? dy : dz)) * rd;
    float rb[PF_R / 2];
#pragma unroll
    for (int k = 0; k < PF_R / 2; ++k) {
        const float ze = (d - p.rbf_mu[2 * k]) * p.rbf_inv_sigma;
        const float zo = (d - p.rbf_mu[2 * k + 1]) * p.rbf_inv_sigma;
        const float re = __expf(-(ze * ze)), ro = __expf(-(zo * zo));
        rb[k] = hl ? ro : re;
    }
    float s[64], Vc[8];
    load_row_f(p.h + (size_t)src * PF_S, hl, s);
    if constexpr (!L0) load_vec_rc(p.v + (size_t)src * 48, hl, wv, Vc);
    else {
#pragma unroll
        for (int q = 0; q < 8; ++q) Vc[q] = 0.f;
    }
    float s1[64], V1[8];
    if constexpr (LOWREG) {
        gvp_coop_compute<17, PF_R, 16, 4, true, L0>(Wfirst, s, rb, Vc, xhat_c, s1, V1, lane, wv, L);
        if (p.n_gvps > 1) {
            CoopW<16, 0> Wn;
            gvp_coop_load<16, 0, 4>(wt[1], lane, wv, Wn);
            gvp_coop_chain1(wt + 1, p.n_gvps - 1, Wn, s1, V1, lane, wv, L);
        }
    } else {
        CoopW<16, 0> Wn;
        if (p.n_gvps > 1) gvp_coop_load<16, 0, 4>(wt[1], lane, wv, Wn);
        gvp_coop_compute<17, PF_R, 16, 4, true, L0>(Wfirst, s, rb, Vc, xhat_c, s1, V1, lane, wv, L);
        if (p.n_gvps > 1) gvp_coop_chain(wt + 1, p.n_gvps - 1, Wn, s1, V1, lane, wv, L);
    }
    const bool valid = j < nvalid;
    const int key = valid ? dst : -1 - j;
    const SegMask sm = seg_masks(key, j);
    if (wv < 3) {
#pragma unroll
        for (int q = 0; q < 8; ++q) L.vx[wv][q][lane] = seg_scan32(valid ? V1[q] : 0.f, sm);
    } else {
#pragma unroll
        for (int q = 0; q < 64; ++q) s1[q] = seg_scan32(valid ? s1[q] : 0.f, sm);
    }
    __syncthreads();
    const int knext = __shfl_down(key, 1, 32);
    if (valid && (j == 31 || knext != key)) {
        if (wv == 3) store_row_f(p.msg_s + (size_t)e * PF_S, hl, s1);
        if (wv == 0) {
            float V[3][8];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int q = 0; q < 8; ++q) V[c][q] = L.vx[c][q][lane];
            store_vec_r(p.msg_v + (size_t)e * 48, hl, V);
        }
    }
}

template <bool L0>
__global__ __launch_bounds__(256, 1) void k_edge_msg_coop(const EdgeParams p) {
    __shared__ CoopLds L;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    edge_tile_coop<L0>(p, p.tiles[blockIdx.x], L, lane, wv);
}
// two workgroups per CU (<= 256 registers): launches with many more tiles than CUs
template <bool L0>
__global__ __launch_bounds__(256, 2) void k_edge_msg_coop2(const EdgeParams p) {
    __shared__ CoopLds L;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    edge_tile_coop<L0, true>(p, p.tiles[blockIdx.x], L, lane, wv);
}

// GVPLayerNorm with the vector state spread over waves 0..2 (coordinate c in wave c)
__device__ __forceinline__ void gvp_layernorm_coop(pf_gcf lw, pf_gcf lb, const int hl, const int lane, const int wv,
                                                   float (&s)[64], float (&Vc)[8], CoopLds& L) {
    if (wv < 3) {
#pragma unroll
        for (int t = 0; t < 8; ++t) L.vx[wv][t][lane] = Vc[t] * Vc[t];
    }
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) sum += s[q];
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / 128.0f);
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) { const float c = s[q] - mean; var = fmaf(c, c, var); }
    var += __shfl_xor(var, 32);
    const float rstd = rsqf_(var * (1.0f / 128.0f) + 1e-5f);
    {   // affine parameters in F-layout, one 32-feature block at a time (keeps the live set small)
        auto pw = reinterpret_cast<const f32x4 PF_AS1*>(lw + 4 * hl);
        auto pb = reinterpret_cast<const f32x4 PF_AS1*>(lb + 4 * hl);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = pw[mt * 8 + q * 2], b4 = pb[mt * 8 + q * 2];
#pragma unroll
                for (int i = 0; i < 4; ++i) s[mt * 16 + 4 * q + i] = (s[mt * 16 + 4 * q + i] - mean) * rstd * w4[i] + b4[i];
            }
    }
    __syncthreads();
    float vn = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) vn += fmaxf(L.vx[0][t][lane] + L.vx[1][t][lane] + L.vx[2][t][lane], 1e-8f);
    vn += __shfl_xor(vn, 32);
    const float rden = rcpf_(sqrtf_(vn * (1.0f / 16.0f) + 1e-5f) + 1e-5f);
#pragma unroll
    for (int t = 0; t < 8; ++t) Vc[t] = Vc[t] * rden;
    __syncthreads();                                   // vx is reused by the caller
}

__device__ __forceinline__ void head_chain_coop(const HeadParams& p, CoopW<16, 0>& Wh0, float (&s1)[64], float (&V1)[8],
                                                const int n, const bool live, CoopLds& L, const int lane, const int wv);
// HEAD: last conv layer of the inference path -- the tile's output never leaves the registers: the noise head runs on it
// right away (dynamics_gvp.py:91 reads the last layer only on the pharm nodes), saving a launch and the head's start-up
template <bool L0, bool HEAD = false>
__device__ __forceinline__ void node_tile_coop(const NodeParams& p, const NodeTile t, CoopLds& L, const int lane, const int wv,
                                               const HeadParams* hp = nullptr) {
    const int nt = __builtin_amdgcn_readfirstlane(t.ntype);
    const int j = lane & 31, hl = lane >> 5;
    int tn = t.n;
    if (t.cnt_idx >= 0) tn = min(tn, max(p.dyn_cnt[t.cnt_idx] - t.rel, 0));
    tn = __builtin_amdgcn_readfirstlane(tn);
    if (tn <= 0) return;                               // block-uniform
    const bool live = j < tn;
    // rows are node ids, or (pruned layer) positions in the list of active protein atoms
    auto node_of = [&](const int row) { return t.ids ? p.row_ids[t.n0 + row] : t.n0 + row; };
    const int n = node_of(min(j, tn - 1));
    const NodeW nw = p.w[nt];
    // Deterministic segmented reduction of the in-edge messages, row-parallel: wave w owns nodes 8w..8w+7 of the
    // tile; for each node the 64 lanes stream its contiguous message rows (128 scalars as float2 + 48 vector
    // floats per row, coalesced, eight rows in flight) and sum them in index order; fn.mean scales by 1/in-degree
    // per etype (zero in-degree -> 0).  Totals meet in LDS and are re-read in the F / R register layouts.
    {
        // the 16 (node, etype) segment descriptors of this wave in one vector load: lane k -> node 8w + (k&7), slot k>>3
        int my_st = 0, my_c = 0;
        if (lane < 16) {
            const int jj = 8 * wv + (lane & 7);
            if (jj < tn) {
                const int slot = (lane >> 3) == 0 ? 0 : (nt == 0 ? p.pp_slot : 1);
                my_st = p.in_start[slot * p.N + node_of(jj)];
                my_c = p.in_cnt[slot * p.N + node_of(jj)];
            }
        }
        // software pipeline over the segments: the partial rows of segment g+1 are in flight while segment g is
        // summed.  A segment's rows are the tails of the (tile, destination) runs inside [st, st+c): slot
        // min(e|31, st+c-1), then the next tile ... ; two rows cover an in-degree of up to 33.
        f32x2 rsA[2], rsB[2];
        float rvA[2], rvB[2];
        auto issue = [&](const int seg, f32x2 (&rs)[2], float (&rv)[2]) {
            const int st = __builtin_amdgcn_readlane(my_st, (seg & 1) * 8 + (seg >> 1));
            const int c = __builtin_amdgcn_readlane(my_c, (seg & 1) * 8 + (seg >> 1));
            const int r0 = c > 0 ? seg_tail(st, st + c) : p.zero_row;
            const int r1 = (c > 0 && r0 + 1 < st + c) ? seg_tail(r0 + 1, st + c) : p.zero_row;
            rs[0] = reinterpret_cast<const f32x2 PF_AS1*>((pf_gcf)p.msg_s + (size_t)r0 * PF_S)[lane];
            rv[0] = lane < 48 ? ((pf_gcf)p.msg_v)[(size_t)r0 * 48 + lane] : 0.f;
            rs[1] = reinterpret_cast<const f32x2 PF_AS1*>((pf_gcf)p.msg_s + (size_t)r1 * PF_S)[lane];
            rv[1] = lane < 48 ? ((pf_gcf)p.msg_v)[(size_t)r1 * 48 + lane] : 0.f;
        };
        float a0 = 0.f, a1 = 0.f, av = 0.f;
        auto reduce = [&](const int seg, const f32x2 (&rs)[2], const float (&rv)[2]) {
            const int st = __builtin_amdgcn_readlane(my_st, (seg & 1) * 8 + (seg >> 1));
            const int c = __builtin_amdgcn_readlane(my_c, (seg & 1) * 8 + (seg >> 1));
            float p0 = rs[0][0] + rs[1][0], p1 = rs[0][1] + rs[1][1], pv = rv[0] + rv[1];   // zero row when absent
            if (c > 0) {
                int e = seg_tail(st, st + c) + 1;
                if (e < st + c) e = seg_tail(e, st + c) + 1;
                while (e < st + c) {                      // rare: more than two partial rows (in-degree > 33)
                    const int row = seg_tail(e, st + c);
                    const f32x2 r = reinterpret_cast<const f32x2 PF_AS1*>((pf_gcf)p.msg_s + (size_t)row * PF_S)[lane];
                    const float v = lane < 48 ? ((pf_gcf)p.msg_v)[(size_t)row * 48 + lane] : 0.f;
                    p0 += r[0]; p1 += r[1]; pv += v;
                    e = row + 1;
                }
            }
            const float sc = (p.norm_mode == 0 && c > 0) ? 1.0f / (float)c : 1.0f;
            a0 = fmaf(p0, sc, a0); a1 = fmaf(p1, sc, a1); av = fmaf(pv, sc, av);
            if (seg & 1) {                                // both etypes of node seg>>1 are in: publish, reset
                const int jj = 8 * wv + (seg >> 1);
                L.agg[jj][2 * lane] = a0;
                L.agg[jj][2 * lane + 1] = a1;
                if (lane < 48) L.agg[jj][128 + lane] = av;
                a0 = 0.f; a1 = 0.f; av = 0.f;
            }
        };
        issue(0, rsA, rvA);
#pragma unroll 1
        for (int seg = 0; seg < 16; seg += 2) {
            issue(seg + 1, rsB, rvB);
            reduce(seg, rsA, rvA);
            if (seg + 2 < 16) issue(seg + 2, rsA, rvA);
            reduce(seg + 1, rsB, rvB);
        }
    }
    float inv_norm = 1.0f;
    if (p.norm_mode == 1) inv_norm = 1.0f / p.norm_value;
    else if (p.norm_mode == 2) inv_norm = 1.0f / p.gnorm[nt * p.B + p.gid[n]];
    float s[64], Vc[8];
    load_row_f(p.h_in + (size_t)n * PF_S, hl, s);
    if constexpr (!L0) {
        if (wv < 3) load_vec_rc(p.v_in + (size_t)n * 48, hl, wv, Vc);
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) Vc[q] = 0.f;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) Vc[q] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            s[mt * 16 + r] = fmaf(L.agg[j][32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hl], inv_norm, s[mt * 16 + r]);
    if (wv < 3) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            Vc[q] = fmaf(L.agg[j][128 + 3 * ((q & 3) + 8 * (q >> 2) + 4 * hl) + wv], inv_norm, Vc[q]);
    }
    gvp_layernorm_coop(nw.ln1_w, nw.ln1_b, hl, lane, wv, s, Vc, L);
    CoopW<16, 0> Wupd;
    gvp_coop_load<16, 0, 4>(nw.upd[0], lane, wv, Wupd);
    // the residual copy of the scalar state waits in LDS (the aggregation buffer is free now) so that the chain
    // fits 256 registers: two workgroups per CU, i.e. every tile of a 262-tile launch is resident at once
    float (*res)[64] = reinterpret_cast<float (*)[64]>(&L.agg[0][0]);
    __syncthreads();                                   // all reads of agg are done
    if (wv == 0) {
#pragma unroll
        for (int q = 0; q < 64; ++q) res[q][lane] = s[q];
    }
    float V1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) V1[q] = Vc[q];
    gvp_coop_chain1(nw.upd, p.n_upd, Wupd, s, V1, lane, wv, L);     // its barriers also publish res
#pragma unroll
    for (int q = 0; q < 64; ++q) s[q] += res[q][lane];
#pragma unroll
    for (int q = 0; q < 8; ++q) Vc[q] += V1[q];
    if constexpr (HEAD) {
        CoopW<16, 0> Wh0;                              // head GVP 0 weights travel while the LayerNorm runs
        if (hp->n_gvps > 1) gvp_coop_load<16, 0, 4>(((const GvpW PF_AS1*)hp->gvps)[0], lane, wv, Wh0);
        gvp_layernorm_coop(nw.ln2_w, nw.ln2_b, hl, lane, wv, s, Vc, L);
        if (wv == 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) Vc[q] = 0.f;
        }
        head_chain_coop(*hp, Wh0, s, Vc, n, live, L, lane, wv);
        return;
    }
    gvp_layernorm_coop(nw.ln2_w, nw.ln2_b, hl, lane, wv, s, Vc, L);
    if (wv < 3) {
#pragma unroll
        for (int q = 0; q < 8; ++q) L.vx[wv][q][lane] = Vc[q];
    }
    __syncthreads();
    if (live) {
        if (wv == 3) store_row_f(p.h_out + (size_t)n * PF_S, hl, s);
        if (wv == 0) {
            float V[3][8];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int q = 0; q < 8; ++q) V[c][q] = L.vx[c][q][lane];
            store_vec_r(p.v_out + (size_t)n * 48, hl, V);
        }
    }
}

template <bool L0>
__global__ __launch_bounds__(256, 2) void k_node_update_coop(const NodeParams p) {
    __shared__ CoopLds L;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    node_tile_coop<L0>(p, p.tiles[blockIdx.x], L, lane, wv);
}
// last conv layer + noise head in one launch (pharm tiles only)
template <bool L0>
__global__ __launch_bounds__(256, 1) void k_node_head_coop(const NodeParams p, const HeadParams hp) {
    __shared__ CoopLds L;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    node_tile_coop<L0, true>(p, p.tiles[blockIdx.x], L, lane, wv, &hp);
}

// noise head on a tile whose state is already in registers (s1: the full scalar row of this lane's node, replicated in
// the four waves; V1: coordinate wv of its vector channels, zeros in wave 3).  Wh0 must hold the weights of head GVP 0
// when n_gvps > 1.
__device__ __forceinline__ void head_chain_coop(const HeadParams& p, CoopW<16, 0>& Wh0, float (&s1)[64], float (&V1)[8],
                                                const int n, const bool live, CoopLds& L, const int lane, const int wv) {
    const int hl = lane >> 5;
    const GvpW PF_AS1* gv = (const GvpW PF_AS1*)p.gvps;
    CoopW<16, 0> Wlast;
    gvp_coop_load<16, 0, 2>(gv[p.n_gvps - 1], lane, wv, Wlast);
    if (p.n_gvps > 1) gvp_coop_chain(gv, p.n_gvps - 1, Wh0, s1, V1, lane, wv, L);
    float so[32], Vo[8];
    gvp_coop_compute<16, 0, 1, 2, false, false>(Wlast, s1, nullptr, V1, 0.f, so, Vo, lane, wv, L);
    PF_STAMP(L);      // chain done
    const int f = n - p.node_base;
    if (wv == 3) {
        // to_scalar_output: Linear(64 -> pharm_nf), rows 0..5 of a 32-row tile
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) o = MFMA(p.a_out[ks * 64 + lane], so[ks], o);
        if (live) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int u = (r & 3) + 8 * (r >> 2) + 4 * hl;
                if (u < p.pharm_nf) p.eps_h[(size_t)f * p.pharm_nf + u] = o[r] + p.b_out[u];
            }
        }
    } else if (live && hl == 0) {
        p.eps_x[(size_t)f * 3 + wv] = Vo[0];           // output vector channel 0, coordinate wv
    }
}

__device__ __forceinline__ void head_tile_coop(const HeadParams& p, const NodeTile t, CoopLds& L, const int lane, const int wv) {
    const int j = lane & 31, hl = lane >> 5;
    const bool live = j < t.n;
    const int n = t.n0 + min(j, t.n - 1);
#ifdef PF_STAMPS
    if (lane == 0) L.scnt[wv] = 0;
#endif
    PF_STAMP(L);      // kernel start
    const GvpW PF_AS1* gv = (const GvpW PF_AS1*)p.gvps;
    CoopW<16, 0> Wh0;
    if (p.n_gvps > 1) gvp_coop_load<16, 0, 4>(gv[0], lane, wv, Wh0);
    float s1[64], V1[8];
    load_row_f(p.h + (size_t)n * PF_S, hl, s1);
    if (wv < 3) load_vec_rc(p.v + (size_t)n * 48, hl, wv, V1);
    else {
#pragma unroll
        for (int q = 0; q < 8; ++q) V1[q] = 0.f;
    }
    head_chain_coop(p, Wh0, s1, V1, n, live, L, lane, wv);
}

__global__ __launch_bounds__(256, 1) void k_noise_head_coop(const HeadParams p) {
    __shared__ CoopLds L;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    head_tile_coop(p, p.tiles[blockIdx.x], L, lane, wv);
}

// ---------------------------------------------------------------------------------------------
// Scalar encoders: h = LayerNorm(SiLU(W [feat, t] + b))   (dynamics_gvp.py:107-117,143-151)
// ---------------------------------------------------------------------------------------------
// Eight nodes of one type per wave: the (transposed, coalesced) weight column of input k is loaded once and
// applied to the 8 nodes; two output features per lane; LayerNorm statistics by wave reduction.
__device__ __forceinline__ void encode_group(const EncodeParams& p, const int nt, const int first, const int cnt, const int lane);
__device__ __forceinline__ void encode_body(const EncodeParams& p, const int blk) {
    const int lane = threadIdx.x & 63;
    const int grp = __builtin_amdgcn_readfirstlane((int)((blk * 256 + threadIdx.x) >> 6));
    const int gprot = (p.Np + 7) >> 3;
    const int nt = grp >= gprot ? 1 : 0;
    const int first = nt ? (grp - gprot) * 8 : grp * 8;            // type-local index of the first node
    const int ntot = nt ? p.Nf : p.Np;
    if (first >= ntot) return;
    encode_group(p, nt, first, min(8, ntot - first), lane);
}
// up to 8 nodes [first, first+cnt) of node type nt
__device__ __forceinline__ void encode_group(const EncodeParams& p, const int nt, const int first, const int cnt, const int lane) {
    const int nf = nt ? p.pharm_nf : p.rec_nf;
    const float* in = (nt ? p.pharm_h : p.prot_h0) + (size_t)first * nf;
    const int nbase = (nt ? p.Np : 0) + first;                     // global node id
    const float* Wt = p.w[nt];                                     // [nf+1][128]
    float a0[8], a1[8];
    const float b0 = p.b[nt][lane], b1 = p.b[nt][lane + 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a0[i] = b0; a1[i] = b1; }
    for (int k = 0; k < nf; ++k) {
        const float w0 = Wt[k * PF_S + lane], w1 = Wt[k * PF_S + lane + 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = in[(size_t)min(i, cnt - 1) * nf + k];
            a0[i] = fmaf(w0, x, a0[i]);
            a1[i] = fmaf(w1, x, a1[i]);
        }
    }
    {
        const float w0 = Wt[nf * PF_S + lane], w1 = Wt[nf * PF_S + lane + 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float tt = p.t ? p.t[p.gid[nbase + min(i, cnt - 1)]] : p.t_scalar;
            a0[i] = fmaf(w0, tt, a0[i]);
            a1[i] = fmaf(w1, tt, a1[i]);
        }
    }
    const float lw0 = p.ln_w[nt][lane], lw1 = p.ln_w[nt][lane + 64], lb0 = p.ln_b[nt][lane], lb1 = p.ln_b[nt][lane + 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float s0 = siluf_(a0[i]), s1 = siluf_(a1[i]);
        float sum = s0 + s1;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum * (1.0f / 128.0f);
        const float c0 = s0 - mean, c1 = s1 - mean;
        float var = c0 * c0 + c1 * c1;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) var += __shfl_xor(var, o);
        const float rstd = rsqf_(var * (1.0f / 128.0f) + 1e-5f);
        if (i < cnt) {
            p.h_out[(size_t)(nbase + i) * PF_S + lane] = c0 * rstd * lw0 + lb0;
            p.h_out[(size_t)(nbase + i) * PF_S + lane + 64] = c1 * rstd * lw1 + lb1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dynamic edges (add_pharm_edges, dynamics_gvp.py:187-215) -- one workgroup per graph, emitted
// destination-major into the graph's fixed-capacity regions; torch_cluster semantics as fixed in
// oracle/pf_oracle.py (strict d^2 < r^2; kNN ordered by (d^2, index)).
// d^2 is evaluated as (dx*dx + dy*dy) + dz*dz with one rounding per operation.
// ---------------------------------------------------------------------------------------------
#ifdef PF_STAMPS
__device__ unsigned long long* g_build_stamps = nullptr;      // [B][32]
#define BSTAMP(k) do { if ((k) < 32 && threadIdx.x == 0 && g_build_stamps) g_build_stamps[blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BSTAMP(k)
#endif
// (sqdist_rn, dkey, wave_min_u64, wave_incl_scan_u32 and the update + build body of the fast path live in pf_stepbuild.h:
// k_n16_tail, pf_n16.hip, runs the same body behind its noise head)
#define SB_STAMP(k) BSTAMP(k)
#include "pf_stepbuild.h"
using pfsb::sqdist_rn; using pfsb::dkey; using pfsb::wave_min_u64; using pfsb::wave_incl_scan_u32;
// exclusive scan of one value per thread over a 256-thread block; returns this thread's offset, *total receives the
// block total.  The value packs three counters (bits 0-15, 16-27, 28-63: the fp edges, the active-atom list and its pp
// in-edges), scanned as two 32-bit halves (low: two 16/12-bit counters that cannot carry into each other at these
// sizes; high part: bits 28-63 shifted down) by wave-level DPP scans and one exchange of the four wave totals in LDS
// (scratch: >= 8 x 8 B).
__device__ __forceinline__ unsigned long long block_excl_scan(const unsigned long long val, unsigned long long* scratch,
                                                              unsigned long long* total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned int lo = (unsigned int)(val & 0xfffffffull), hi = (unsigned int)(val >> 28);
    const unsigned int slo = wave_incl_scan_u32(lo), shi = wave_incl_scan_u32(hi);
    if (lane == 63) scratch[wave] = (unsigned long long)slo | ((unsigned long long)shi << 28);
    __syncthreads();
    unsigned long long before = 0ull, all = 0ull;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned long long t = scratch[w];
        if (w < wave) before += t;
        all += t;
    }
    *total = all;
    __syncthreads();
    return before + ((unsigned long long)slo | ((unsigned long long)shi << 28)) - val;
}

// Protein side of the dynamic edges of one graph, destination-major: the fp edges (sources = the pharm nodes
// that reference atom c, ascending) and -- receptive-field pruning, see DESIGN.md -- the list of ACTIVE atoms
// (atoms that are the source of a pf edge, the only protein rows the last conv layer reads) together with a
// compact copy of their static pp in-edges.  `refs(c, visit)` calls visit(fl) for every referencing pharm node.
// static pp in-edges of this thread's atom of the first 256-atom chunk, fetched BEFORE the neighbour search (the
// loads depend only on the static graph): emit_prot_side then only stores.  Atoms with more than 16 in-edges and
// later chunks take the load-and-store path.
struct PpPrefetch {
    int st, deg;
    int src[16];
};
__device__ __forceinline__ void pp_prefetch(const BuildParams& p, const int p0, const int Np, PpPrefetch& f) {
    f.st = 0; f.deg = 0;
    const int c = threadIdx.x;
    const bool want = p.act_ids && !p.pa_static && c < Np;       // (pocket sharing: no pa copy, nothing to prefetch)
    if (want) {
        f.st = p.in_start[p.N + p0 + c];
        f.deg = p.in_cnt[p.N + p0 + c];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) f.src[k] = want ? p.esrc[f.st + min(k, max(f.deg - 1, 0))] : 0;
}

template <typename Refs>
__device__ __forceinline__ void emit_prot_side(const BuildParams& p, const int g, const int p0, const int Np, const int GF,
                                               unsigned long long* scratch, const PpPrefetch& pre, Refs refs) {
    const int tid = threadIdx.x;
    int* in_start0 = p.in_start;             int* in_cnt0 = p.in_cnt;
    const int* in_start1 = p.in_start + p.N; const int* in_cnt1 = p.in_cnt + p.N;
    int* in_start2 = p.in_start + 2 * p.N;   int* in_cnt2 = p.in_cnt + 2 * p.N;
    const int reg_fp = p.reg[2 * p.B + g], reg_pa = p.reg[3 * p.B + g];
    const int reg_act = p.act_ids ? p.reg_act[g] : 0;
    unsigned long long base = 0ull;
    for (int c0 = 0; c0 < Np; c0 += 256) {
        const int c = c0 + tid;
        int my = 0;
        if (c < Np) refs(c, [&](int) { ++my; });
        const int act = (my > 0 && p.act_ids) ? 1 : 0;
        const int deg = act ? (c0 == 0 ? pre.deg : in_cnt1[p0 + c]) : 0;
        unsigned long long tot;
        const unsigned long long o = base + block_excl_scan((unsigned long long)my | ((unsigned long long)act << 16) |
                                                            ((unsigned long long)deg << 28), scratch, &tot);
        if (c < Np) {
            int e = reg_fp + (int)(o & 0xffffu);
            in_start0[p0 + c] = e;
            in_cnt0[p0 + c] = my;
            if (my) refs(c, [&](int fl) { p.esrc[e] = GF + fl; p.edst[e] = p0 + c; ++e; });
            if (act) p.act_ids[reg_act + (int)((o >> 16) & 0xfffu)] = p0 + c;
            if (act && p.pa_static) p.need[p.rep_base[g] + c] = p.need_stamp;
            if (act && !p.pa_static) {
                const int d0 = reg_pa + (int)(o >> 28), s0 = c0 == 0 ? pre.st : in_start1[p0 + c];
                in_start2[p0 + c] = d0;
                in_cnt2[p0 + c] = deg;
                int i0 = 0;
                if (c0 == 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < deg) { p.esrc[d0 + k] = pre.src[k]; p.edst[d0 + k] = p0 + c; if (p.eorig) p.eorig[d0 + k] = s0 + k; }
                    i0 = 16;
                }
                // eight loads in flight, then eight stores (source and destination alias the same array, so a
                // plain copy loop would serialise on memory latency)
                for (int i = i0; i < deg; i += 8) {
                    int tmp[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) tmp[k] = p.esrc[s0 + min(i + k, deg - 1)];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (i + k < deg) { p.esrc[d0 + i + k] = tmp[k]; p.edst[d0 + i + k] = p0 + c; if (p.eorig) p.eorig[d0 + i + k] = s0 + i + k; }
                }
            }
        }
        base += tot;
    }
    if (tid == 0 && p.act_ids) {
        p.dyn_cnt[3 * p.B + g] = p.pa_static ? p.pa_static[g] : (int)(base >> 28);
        p.dyn_cnt[4 * p.B + g] = (int)((base >> 16) & 0xfffu);
    }
}

__device__ __forceinline__ void build_body(const BuildParams& p, const int g) {
    __shared__ float4 fx[PF_MAXF];
    __shared__ int cnt[PF_MAXF];
    __shared__ int off[PF_MAXF + 1];
    __shared__ int knn_idx[PF_MAXF * PF_MAXK];
    __shared__ unsigned long long scratch[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = p.prot_ptr[g], p1 = p.prot_ptr[g + 1];
    const int f0 = p.pharm_ptr[g], f1 = p.pharm_ptr[g + 1];
    const int Np = p1 - p0, Nf = f1 - f0;
    const int GF = p.Np_tot + f0;                     // global id of this graph's pharm node 0
    int* in_start0 = p.in_start;          int* in_cnt0 = p.in_cnt;            // slot 0: ff (pharm) / fp (prot)
    int* in_start1 = p.in_start + p.N;    int* in_cnt1 = p.in_cnt + p.N;      // slot 1: pf (pharm) / pp (prot)
    BSTAMP(8);                                        // build start
    if (tid < Nf) fx[tid] = p.xn[GF + tid];
    PpPrefetch pre;
    pp_prefetch(p, p0, Np, pre);                      // in flight under the neighbour searches
    __syncthreads();
    // ------------------------------------------------------------------ ff (pharm -> pharm)
    const int reg_ff = p.reg[0 * p.B + g];
    const int kff = p.ff_k > 0 ? min(p.ff_k, Nf - 1) : 0;
    if (tid < Nf) {
        int c = 0;
        if (p.ff_k > 0) c = max(kff, 0);
        else
            for (int jn = 0; jn < Nf; ++jn)
                if (jn != tid && sqdist_rn(fx[jn], fx[tid]) < p.r2_ff) ++c;
        cnt[tid] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < Nf; ++i) { off[i] = run; run += cnt[i]; }
        off[Nf] = run;
        p.dyn_cnt[0 * p.B + g] = run;
    }
    __syncthreads();
    if (tid < Nf) {
        int e = reg_ff + off[tid];
        in_start0[GF + tid] = e;
        in_cnt0[GF + tid] = cnt[tid];
        if (p.ff_k > 0) {
            unsigned long long prev = 0ull;
            bool first = true;
            for (int q = 0; q < kff; ++q) {
                unsigned long long best = ~0ull;
                for (int jn = 0; jn < Nf; ++jn) {
                    if (jn == tid) continue;
                    const unsigned long long k = dkey(sqdist_rn(fx[jn], fx[tid]), jn);
                    if ((first || k > prev) && k < best) best = k;
                }
                prev = best; first = false;
                p.esrc[e] = GF + (int)(best & 0xffffffffu);
                p.edst[e] = GF + tid;
                ++e;
            }
        } else {
            for (int jn = 0; jn < Nf; ++jn)
                if (jn != tid && sqdist_rn(fx[jn], fx[tid]) < p.r2_ff) {
                    p.esrc[e] = GF + jn; p.edst[e] = GF + tid; ++e;
                }
        }
    }
    __syncthreads();   // cnt/off are reused below
    BSTAMP(9);                                        // ff done
    // ------------------------------------------------------------------ pf (prot -> pharm)
    const int reg_pf = p.reg[1 * p.B + g];
    if (p.pf_k > 0) {
        const int kk = min(p.pf_k, Np);
        for (int fl = wave; fl < Nf; fl += 4) {
            const float4 q = fx[fl];
            // up to 8 candidates per lane (Np <= 512) stay in registers as (d^2, index) keys; larger pockets
            // re-read the coordinates every round
            unsigned long long kc[8];
            const bool cached = Np <= 512;
            if (cached) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = lane + 64 * i;
                    kc[i] = c < Np ? dkey(sqdist_rn(p.xn[p0 + min(c, Np - 1)], q), c) : ~0ull;
                }
            }
            unsigned long long prev = 0ull;
            for (int r = 0; r < kk; ++r) {
                unsigned long long best = ~0ull;
                if (cached) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if ((r == 0 || kc[i] > prev) && kc[i] < best) best = kc[i];
                } else {
                    for (int c = lane; c < Np; c += 64) {
                        const unsigned long long k = dkey(sqdist_rn(p.xn[p0 + c], q), c);
                        if ((r == 0 || k > prev) && k < best) best = k;
                    }
                }
                best = wave_min_u64(best);
                prev = best;
                if (lane == 0) {
                    const int pc = (int)(best & 0xffffffffu);
                    knn_idx[fl * PF_MAXK + r] = pc;
                    p.esrc[reg_pf + fl * kk + r] = p0 + pc;
                    p.edst[reg_pf + fl * kk + r] = GF + fl;
                }
            }
            if (lane == 0) { in_start1[GF + fl] = reg_pf + fl * kk; in_cnt1[GF + fl] = kk; }
        }
        if (tid == 0) { p.dyn_cnt[1 * p.B + g] = Nf * kk; p.dyn_cnt[2 * p.B + g] = Nf * kk; }
        __syncthreads();
        BSTAMP(10);                                   // kNN done
        // fp = pf reversed, destination-major over the protein atoms of this graph (+ active atoms, their pp edges)
        emit_prot_side(p, g, p0, Np, GF, scratch, pre, [&](const int c, auto visit) {
            for (int fl = 0; fl < Nf; ++fl)
                for (int r = 0; r < kk; ++r)
                    if (knn_idx[fl * PF_MAXK + r] == c) visit(fl);
        });
    } else {
        // radius(x=pharm, y=prot, r): pf = {prot -> pharm}, fp = reverse
        for (int fl = wave; fl < Nf; fl += 4) {
            const float4 q = fx[fl];
            int c = 0;
            for (int c0 = 0; c0 < Np; c0 += 64) {
                const int pc = c0 + lane;
                const bool in = pc < Np && sqdist_rn(q, p.xn[p0 + pc]) < p.r2_pf;
                c += __popcll(__ballot(in));
            }
            if (lane == 0) cnt[fl] = c;
        }
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int i = 0; i < Nf; ++i) { off[i] = run; run += cnt[i]; }
            off[Nf] = run;
            p.dyn_cnt[1 * p.B + g] = run;
            p.dyn_cnt[2 * p.B + g] = run;
        }
        __syncthreads();
        for (int fl = wave; fl < Nf; fl += 4) {
            const float4 q = fx[fl];
            int e = reg_pf + off[fl];
            if (lane == 0) { in_start1[GF + fl] = e; in_cnt1[GF + fl] = cnt[fl]; }
            for (int c0 = 0; c0 < Np; c0 += 64) {
                const int pc = c0 + lane;
                const bool in = pc < Np && sqdist_rn(q, p.xn[p0 + pc]) < p.r2_pf;
                const unsigned long long m = __ballot(in);
                if (in) {
                    const int pos = e + __popcll(m & ((1ull << lane) - 1ull));
                    p.esrc[pos] = p0 + pc; p.edst[pos] = GF + fl;
                }
                e += __popcll(m);
            }
        }
        emit_prot_side(p, g, p0, Np, GF, scratch, pre, [&](const int c, auto visit) {
            const float4 xc = p.xn[p0 + c];
            for (int fl = 0; fl < Nf; ++fl)
                if (sqdist_rn(fx[fl], xc) < p.r2_pf) visit(fl);
        });
    }
    // per-graph normalisers for message_norm == 0 (gvp.py:504-507)
    __syncthreads();
    BSTAMP(11);                                       // protein side emitted
    if (tid == 0 && p.norm_mode == 2) {
        const int cff = p.dyn_cnt[0 * p.B + g];
        const int cpf = p.pfq_cnt ? p.pfq_cnt[g] : p.dyn_cnt[1 * p.B + g], cfp = p.pfq_cnt ? p.pfq_cnt[g] : p.dyn_cnt[2 * p.B + g];
        p.gnorm[1 * p.B + g] = (float)(cff + cpf) / (float)Nf + 1.0f;
        p.gnorm[0 * p.B + g] = (float)(cfp + p.pp_cnt[g]) / (float)Np + 1.0f;
    }
}

// Protein encoder + per-source-node precompute for the pp messages of conv layer 0, one 256-thread block per 32
// atoms (4-wave cooperative form): h0 = LN(SiLU(W_enc [feat, t] + b)) on the matrix cores (wave w owns output
// tile w), LayerNorm statistics on the full row after an LDS exchange, then P = W_msg0[:, :128] h0 + b_msg0
// (64 MFMAs per wave).  Both rows are stored; the edge kernel gathers P straight into its accumulators.
__device__ __forceinline__ void encode_pre_tile(const PreParams& p, const int tile, CoopLds& L) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, hl = lane >> 5;
    const int n0 = tile * 32, cnt = min(32, p.Np - n0);
    const bool live = j < cnt;
    const int n = n0 + min(j, cnt - 1);
    const float tt = p.t ? p.t[p.gid[n]] : p.t_scalar;
    pf_gcf in = (pf_gcf)p.prot_h0 + (size_t)n * p.rec_nf;
    f32x16 acc;
    {
        const f32x4 PF_AS1* bp = reinterpret_cast<const f32x4 PF_AS1*>(p.b_enc + hl * 64 + wv * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 b4 = bp[q]; acc[4 * q] = b4[0]; acc[4 * q + 1] = b4[1]; acc[4 * q + 2] = b4[2]; acc[4 * q + 3] = b4[3]; }
    }
    for (int t = 0; t < p.nke; ++t) {                 // k-step t: this half feeds input 2t + hl ([features, t])
        const int k = 2 * t + hl;
        const float b = k < p.rec_nf ? in[k] : (k == p.rec_nf ? tt : 0.f);
        acc = MFMA(p.a_enc[((size_t)wv * p.nke + t) * 64 + lane], b, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) L.so[wv][r][lane] = siluf_(acc[r]);
    __syncthreads();
    float s[64];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[mt * 16 + r] = L.so[mt][r][lane];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) sum += s[q];
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / 128.0f);
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) { const float c = s[q] - mean; var = fmaf(c, c, var); }
    var += __shfl_xor(var, 32);
    const float rstd = rsqf_(var * (1.0f / 128.0f) + 1e-5f);
    {
        auto pw = reinterpret_cast<const f32x4 PF_AS1*>(p.ln_w + 4 * hl);
        auto pb = reinterpret_cast<const f32x4 PF_AS1*>(p.ln_b + 4 * hl);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = pw[mt * 8 + q * 2], b4 = pb[mt * 8 + q * 2];
#pragma unroll
                for (int i = 0; i < 4; ++i) s[mt * 16 + 4 * q + i] = (s[mt * 16 + 4 * q + i] - mean) * rstd * w4[i] + b4[i];
            }
    }
    // P tile wv = b + sum_k W[32wv.., k] h0[k]  over the 64 feature k-steps of the first pp message GVP
    f32x16 pa;
    {
        const f32x4 PF_AS1* bp = reinterpret_cast<const f32x4 PF_AS1*>(p.pre_w.b_main + hl * 64 + wv * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 b4 = bp[q]; pa[4 * q] = b4[0]; pa[4 * q + 1] = b4[1]; pa[4 * q + 2] = b4[2]; pa[4 * q + 3] = b4[3]; }
    }
    const f32x4 PF_AS1* ap = reinterpret_cast<const f32x4 PF_AS1*>(p.pre_w.a_main_c) + (size_t)wv * ((p.pre_nks + 3) / 4) * 64 + lane;
    f32x4 a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = ap[q * 64];
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) pa = MFMA(a[ks >> 2][ks & 3], s[ks], pa);
    if (live) {
        // wave w stores tile w (features 32w .. 32w+31) of both rows
        f32x4* ph = reinterpret_cast<f32x4*>(p.h_out + (size_t)n * PF_S + 32 * wv + 4 * hl);
        f32x4* pp = reinterpret_cast<f32x4*>(p.pre_out + (size_t)n * PF_S + 32 * wv + 4 * hl);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 x, y;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = s[wv * 16 + 4 * q + i]; y[i] = pa[4 * q + i]; }
            ph[q * 2] = x; pp[q * 2] = y;
        }
    }
}

// encoders and dynamic edge build are independent (one reads h/t, the other coordinates): one launch,
// the first B blocks build edges, the rest encode
__global__ __launch_bounds__(256) void k_encode_build(const EncodeParams ep, const BuildParams bp) {
    if ((int)blockIdx.x < bp.B) build_body(bp, blockIdx.x);
    else encode_body(ep, blockIdx.x - bp.B);
}
// the same launch with the protein side done by encode_pre_tile: blocks [0,B) build edges, [B, B+gp) encode the
// pharmacophore nodes (8 per wave), the rest encode 32 protein atoms each and precompute P
__global__ __launch_bounds__(256) void k_encode_build_pre(const EncodeParams ep, const BuildParams bp, const PreParams pp) {
    __shared__ CoopLds L;
    const int gp = ((ep.Nf + 7) / 8 + 3) / 4;
    const int b = blockIdx.x;
    if (b < bp.B) build_body(bp, b);
    else if (b < bp.B + gp) {
        const int lane = threadIdx.x & 63;
        const int grp = __builtin_amdgcn_readfirstlane((int)(((b - bp.B) * 256 + threadIdx.x) >> 6));
        const int first = grp * 8;
        if (first < ep.Nf) encode_group(ep, 1, first, min(8, ep.Nf - first), lane);
    } else encode_pre_tile(pp, b - bp.B - gp, L);
}
__global__ __launch_bounds__(256) void k_build_edges(const BuildParams p) { build_body(p, blockIdx.x); }
__global__ __launch_bounds__(256) void k_encode(const EncodeParams p) { encode_body(p, blockIdx.x); }

// ---------------------------------------------------------------------------------------------
// small state kernels
// ---------------------------------------------------------------------------------------------
// copy [n,3] coordinates into the float4 node array (optionally subtracting a per-graph shift)
__global__ void k_load_coords(const float* src, float4* xn, const int n, const int* gid, const float* shift,
                              const float sign) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
    if (shift) {
        const int g = gid[i];
        x += sign * shift[3 * g]; y += sign * shift[3 * g + 1]; z += sign * shift[3 * g + 2];
    }
    xn[i] = make_float4(x, y, z, 0.f);
}
// split [n, 3+nf] noise rows into coordinates (float4 array) and features
__global__ void k_load_noise0(const float* nz, float4* xn, float* hf, const int n, const int nf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = nz + (size_t)i * (3 + nf);
    xn[i] = make_float4(r[0], r[1], r[2], 0.f);
    for (int k = 0; k < nf; ++k) hf[(size_t)i * nf + k] = r[3 + k];
}
__global__ void k_copy(const float* src, float* dst, const size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// clear up to eight regions with one launch (a bind clears four small-to-medium regions, the training workspace five: one
// launch of ~5 us each as separate memsets).  Regions are dword-aligned; 16-byte aligned ones are cleared with b128 stores.
__global__ __launch_bounds__(256) void k_zero_multi(const ZeroList z) {
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < z.cnt; ++r) {
        char* const p = reinterpret_cast<char*>(z.p[r]);
        const size_t nb = z.nbytes[r];
        if (((reinterpret_cast<uintptr_t>(p) | nb) & 15) == 0) {
            float4* q = reinterpret_cast<float4*>(p);
            for (size_t i = t0; i < (nb >> 4); i += nt) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            float* q = reinterpret_cast<float*>(p);
            for (size_t i = t0; i < (nb >> 2); i += nt) q[i] = 0.f;
        }
    }
}
// two copies in one launch (the bind's coordinate and feature rows)
__global__ void k_copy2(const float* a, float* da, const size_t na, const float* b, float* db, const size_t nb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) da[i] = a[i];
    else if (i - na < nb) db[i - na] = b[i - na];
}
// pf_set_pocket_groups with DEVICE rows (pf_set_pocket_batch): the claim "graph g is a copy of its representative" is checked where
// the rows are -- atom i of graph g against the same atom of the representative (coordinates and features, bit patterns);
// any difference sets *flag, which travels to the host with the one-hot verdict and fails the first call that would share
__global__ void k_verify_copies(const float* __restrict__ x0, const float* __restrict__ h0, const int* __restrict__ gid,
                                const int* __restrict__ prot_ptr, const int* __restrict__ rep_base, const int Np, const int rec_nf,
                                int* __restrict__ flag) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= Np) return;
    const int g = gid[i];
    const int j = rep_base[g] + (i - prot_ptr[g]);
    if (j == i) return;
    bool same = true;
    for (int c = 0; c < 3; ++c) same = same && __float_as_uint(x0[(size_t)i * 3 + c]) == __float_as_uint(x0[(size_t)j * 3 + c]);
    for (int k = 0; k < rec_nf; ++k) same = same && __float_as_uint(h0[(size_t)i * rec_nf + k]) == __float_as_uint(h0[(size_t)j * rec_nf + k]);
    if (!same) atomicOr(flag, 1);
}
// per-graph mean of coordinates (dgl.readout_nodes op='mean'); one wave per graph, fixed order
__global__ __launch_bounds__(64) void k_segment_mean(const float4* xn, const int* ptr, const int base, float* out) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int a = ptr[g], b = ptr[g + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = a + lane; i < b; i += 64) { const float4 x = xn[base + i]; sx += x.x; sy += x.y; sz += x.z; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
    if (lane == 0) {
        const float n = (float)max(b - a, 1);
        out[3 * g] = sx / n; out[3 * g + 1] = sy / n; out[3 * g + 2] = sz / n;
    }
}

// z_s = mu + sigma * noise ; remove the pharmacophore COM from pharm and prot coordinates
// (pharmacodiff.py:413-429).  One workgroup per graph.
__device__ __forceinline__ void step_update_body(const StepParams& p, const int g) {
    __shared__ float com[3];
    __shared__ float red[4][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f0 = p.pharm_ptr[g], f1 = p.pharm_ptr[g + 1];
    const int p0 = p.prot_ptr[g], p1 = p.prot_ptr[g + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int f = f0 + tid; f < f1; f += 256) {
        const float4 x = p.xn[p.Np_tot + f];
        const float* nz = p.noise + (size_t)f * (3 + p.nf);
        float m[3];
        const float xi[3] = {x.x, x.y, x.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = p.eps_x[(size_t)f * 3 + c];
            m[c] = pf_feat_update(xi[c], e, nz[c], p.a_ts, p.var, p.sigma, p.ep_zt, p.ep_pred, p.ep_coord);      // (pf_device.h: the one place the update is written)
        }
        p.xn[p.Np_tot + f] = make_float4(m[0], m[1], m[2], 0.f);
        sx += m[0]; sy += m[1]; sz += m[2];
        for (int k = 0; k < p.nf; ++k) {
            const float hv = p.pharm_h[(size_t)f * p.nf + k];
            const float e = p.eps_h[(size_t)f * p.nf + k];
            p.pharm_h[(size_t)f * p.nf + k] = pf_feat_update(hv, e, nz[3 + k], p.a_ts, p.var, p.sigma, p.ep_zt, p.ep_pred, p.ep_feat);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
    if (lane == 0) { red[wave][0] = sx; red[wave][1] = sy; red[wave][2] = sz; }
    __syncthreads();
    if (tid == 0) {
        const float n = (float)max(f1 - f0, 1);
        for (int c = 0; c < 3; ++c) com[c] = (f1 > f0) ? (((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) / n) : 0.f;
    }
    __syncthreads();
    const float cx = com[0], cy = com[1], cz = com[2];
    for (int f = f0 + tid; f < f1; f += 256) {
        float4 x = p.xn[p.Np_tot + f];
        x.x -= cx; x.y -= cy; x.z -= cz;
        p.xn[p.Np_tot + f] = x;
    }
    for (int i = p0 + tid; i < p1; i += 256) {
        float4 x = p.xn[i];
        x.x -= cx; x.y -= cy; x.z -= cz;
        p.xn[i] = x;
    }
}
__global__ __launch_bounds__(256) void k_step_update(const StepParams p) { step_update_body(p, blockIdx.x); }
// p(z_s | z_t) update of graph g and, on the new coordinates, the dynamic edges of the NEXT dynamics call: both are
// one-workgroup-per-graph kernels, and with the encoders computed on the fly by the row-group kernels the edge build is
// all that a step's first launch would do -- one launch less per denoising step.
// ---------------------------------------------------------------------------------------------
// k_step_build_fast: the same update + edge build for the common shape (kNN pf edges, pockets of at most 512 atoms, one
// atom per thread), organised around what a launch actually pays for here: every launch starts with cold caches (each
// XCD's L2 is invalidated at kernel boundaries), a dependent global round trip costs ~2,000 cycles, and __syncthreads
// drains every outstanding load.  The generic bodies above make ~10 dependent trips per graph (44 k cycles); this
// kernel makes three: (A) the graph's pointers and regions, (B) every input row -- pharm state, eps, noise, protein
// coordinates, the static in-edge descriptors -- (C) the static pp sources of the thread's atom.  The updated
// coordinates stay in LDS for the neighbour searches (8 waves: one pharmacophore center each), and nothing is read back
// from global memory.  Results are identical to k_step_build (same arithmetic, same orderings).
// ---------------------------------------------------------------------------------------------
// (the body: pf_stepbuild.h.  The leading scalar arguments repeat the fields round trip (A) needs: preloaded into scalar
// registers with the wave, -mllvm -amdgpu-kernarg-preload-count in the Makefile, so (A) does not wait for the kernel-argument segment)
__global__ __launch_bounds__(512) void k_step_build_fast(const int* __restrict__ a_prot_ptr, const int* __restrict__ a_pharm_ptr,
                                                         const int* __restrict__ a_reg, const int a_B, const int a_Np_tot,
                                                         const StepParams sp, const BuildParams p) {
    __shared__ pfsb::StepBuildLds L;
    const int g = blockIdx.x;
    const pfsb::EpsGlobal eps{sp.eps_x, sp.eps_h, a_pharm_ptr[g], sp.nf};
    pfsb::step_build_fast_body<512>(g, a_prot_ptr, a_pharm_ptr, a_reg, a_B, a_Np_tot, sp, p, eps, L);
}

__global__ __launch_bounds__(256) void k_step_build(const StepParams sp, const BuildParams bp) {
    BSTAMP(0);                                         // kernel start
    step_update_body(sp, blockIdx.x);
    __syncthreads();                                   // this workgroup's coordinate writes are visible to all its waves
    build_body(bp, blockIdx.x);
}

// out[i] = xn[base+i] + (add[g] - sub[g]) ; used for the final frame of reference
__global__ void k_export_coords(const float4* xn, const int base, const int n, const int* gid, const float* add,
                                const float* sub, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = gid[base + i];
    const float4 x = xn[base + i];
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (add) { dx += add[3 * g]; dy += add[3 * g + 1]; dz += add[3 * g + 2]; }
    out[3 * i] = (sub ? x.x - sub[3 * g] : x.x) + dx;
    out[3 * i + 1] = (sub ? x.y - sub[3 * g + 1] : x.y) + dy;
    out[3 * i + 2] = (sub ? x.z - sub[3 * g + 2] : x.z) + dz;
}
__global__ void k_scale_copy(const float* src, float* dst, const size_t n, const float s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * s;
}

// static pp radius graph (dataset/protein_pharm_dataset.py:234-236): one workgroup per graph;
// pass 0 counts per-target neighbours, pass 1 writes (source, target) target-major.
__global__ __launch_bounds__(256) void k_pp_radius(const float4* xn, const int* prot_ptr, const float r2, const int maxn,
                                                   int* deg, const int* row_off, int* src, int* dst, const int pass) {
    const int g = blockIdx.x;
    const int a = prot_ptr[g], b = prot_ptr[g + 1];
    for (int i = a + threadIdx.x; i < b; i += 256) {
        const float4 xi = xn[i];
        int c = 0;
        int e = pass ? row_off[i] : 0;
        for (int jn = a; jn < b && c < maxn; ++jn) {
            if (jn == i) continue;
            if (sqdist_rn(xn[jn], xi) < r2) {
                if (pass) { src[e] = jn; dst[e] = i; ++e; }
                ++c;
            }
        }
        if (!pass) deg[i] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// launch helpers (called from pf_host.cpp)
// ---------------------------------------------------------------------------------------------
extern "C" {
#ifdef PF_STAMPS
int pfk_build_set_stamp_buffer(unsigned long long* dev) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_build_stamps), &dev, sizeof(dev)); }
int pfk_set_stamp_buffer(unsigned long long* dev) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pf_stamps), &dev, sizeof(dev)); }
#endif
void pfk_edge_msg_coop2(const EdgeParams* p, int layer0, hipStream_t s) {
    if (p->ntiles == 0) return;
    if (layer0) hipLaunchKernelGGL(k_edge_msg_coop2<true>, dim3(p->ntiles), dim3(256), 0, s, *p);
    else hipLaunchKernelGGL(k_edge_msg_coop2<false>, dim3(p->ntiles), dim3(256), 0, s, *p);
}
void pfk_edge_msg_coop(const EdgeParams* p, int layer0, hipStream_t s) {
    if (p->ntiles == 0) return;
    if (layer0) hipLaunchKernelGGL(k_edge_msg_coop<true>, dim3(p->ntiles), dim3(256), 0, s, *p);
    else hipLaunchKernelGGL(k_edge_msg_coop<false>, dim3(p->ntiles), dim3(256), 0, s, *p);
}
void pfk_node_head_coop(const NodeParams* p, const HeadParams* hp, int layer0, hipStream_t s) {
    if (p->ntiles == 0) return;
    if (layer0) hipLaunchKernelGGL(k_node_head_coop<true>, dim3(p->ntiles), dim3(256), 0, s, *p, *hp);
    else hipLaunchKernelGGL(k_node_head_coop<false>, dim3(p->ntiles), dim3(256), 0, s, *p, *hp);
}
void pfk_node_update_coop(const NodeParams* p, int layer0, hipStream_t s) {
    if (p->ntiles == 0) return;
    if (layer0) hipLaunchKernelGGL(k_node_update_coop<true>, dim3(p->ntiles), dim3(256), 0, s, *p);
    else hipLaunchKernelGGL(k_node_update_coop<false>, dim3(p->ntiles), dim3(256), 0, s, *p);
}
void pfk_noise_head_coop(const HeadParams* p, hipStream_t s) {
    if (p->ntiles == 0) return;
    hipLaunchKernelGGL(k_noise_head_coop, dim3(p->ntiles), dim3(256), 0, s, *p);
}
void pfk_edge_msg(const EdgeParams* p, int layer0, hipStream_t s) {
    const int blocks = (p->ntiles + 3) / 4;
    if (blocks == 0) return;
    if (p->sv_z != nullptr && p->bf16) {      // training forward of the bf16 leg
        if (layer0) hipLaunchKernelGGL((k_edge_msg<true, true, true>), dim3(blocks), dim3(256), 0, s, *p);
        else hipLaunchKernelGGL((k_edge_msg<false, true, true>), dim3(blocks), dim3(256), 0, s, *p);
        return;
    }
    if (p->sv_z != nullptr) {      // training forward: also write the per-level rows the backward pass reads
        if (layer0) hipLaunchKernelGGL((k_edge_msg<true, true>), dim3(blocks), dim3(256), 0, s, *p);
        else hipLaunchKernelGGL((k_edge_msg<false, true>), dim3(blocks), dim3(256), 0, s, *p);
        return;
    }
    if (layer0) hipLaunchKernelGGL(k_edge_msg<true>, dim3(blocks), dim3(256), 0, s, *p);
    else hipLaunchKernelGGL(k_edge_msg<false>, dim3(blocks), dim3(256), 0, s, *p);
}
void pfk_node_update(const NodeParams* p, int layer0, hipStream_t s) {
    const int blocks = p->ntiles;
    if (blocks == 0) return;
    if (layer0) hipLaunchKernelGGL(k_node_update<true>, dim3(blocks), dim3(64), 0, s, *p);
    else hipLaunchKernelGGL(k_node_update<false>, dim3(blocks), dim3(64), 0, s, *p);
}
void pfk_noise_head(const HeadParams* p, hipStream_t s) {
    const int blocks = p->ntiles;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_noise_head, dim3(blocks), dim3(64), 0, s, *p);
}
void pfk_encode_build(const EncodeParams* e, const BuildParams* b, hipStream_t s) {
    const int groups = (e->Np + 7) / 8 + (e->Nf + 7) / 8;
    hipLaunchKernelGGL(k_encode_build, dim3(b->B + (groups + 3) / 4), dim3(256), 0, s, *e, *b);
}
void pfk_encode_build_pre(const EncodeParams* e, const BuildParams* b, const PreParams* pp, hipStream_t s) {
    const int gp = ((e->Nf + 7) / 8 + 3) / 4;
    hipLaunchKernelGGL(k_encode_build_pre, dim3(b->B + gp + (pp->Np + 31) / 32), dim3(256), 0, s, *e, *b, *pp);
}
void pfk_encode(const EncodeParams* p, hipStream_t s) {
    const int groups = (p->Np + 7) / 8 + (p->Nf + 7) / 8;
    if (groups == 0) return;
    hipLaunchKernelGGL(k_encode, dim3((groups + 3) / 4), dim3(256), 0, s, *p);
}
void pfk_build_edges(const BuildParams* p, hipStream_t s) {
    if (p->B == 0) return;
    hipLaunchKernelGGL(k_build_edges, dim3(p->B), dim3(256), 0, s, *p);
}
void pfk_load_coords(const float* src, float4* xn, int n, const int* gid, const float* shift, float sign, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_load_coords, dim3((n + 255) / 256), dim3(256), 0, s, src, xn, n, gid, shift, sign);
}
void pfk_load_noise0(const float* nz, float4* xn, float* hf, int n, int nf, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_load_noise0, dim3((n + 255) / 256), dim3(256), 0, s, nz, xn, hf, n, nf);
}
void pfk_copy(const float* src, float* dst, size_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
}
void pfk_verify_copies(const float* x0, const float* h0, const int* gid, const int* prot_ptr, const int* rep_base, int Np, int rec_nf,
                       int* flag, hipStream_t s) {
    if (Np <= 0) return;
    hipLaunchKernelGGL(k_verify_copies, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, s, x0, h0, gid, prot_ptr, rep_base, Np, rec_nf, flag);
}
void pfk_copy2(const float* a, float* da, size_t na, const float* b, float* db, size_t nb, hipStream_t s) {
    if (na + nb == 0) return;
    hipLaunchKernelGGL(k_copy2, dim3((unsigned)((na + nb + 255) / 256)), dim3(256), 0, s, a, da, na, b, db, nb);
}
void pfk_zero_multi(const ZeroList* z, hipStream_t s) {
    if (z->cnt == 0) return;
    size_t big = 0;
    for (int r = 0; r < z->cnt; ++r) big = std::max(big, (size_t)z->nbytes[r]);
    const unsigned blocks = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (big / 16 + 255) / 256));
    hipLaunchKernelGGL(k_zero_multi, dim3(blocks), dim3(256), 0, s, *z);
}
void pfk_scale_copy(const float* src, float* dst, size_t n, float sc, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n, sc);
}
void pfk_segment_mean(const float4* xn, const int* ptr, int base, int B, float* out, hipStream_t s) {
    if (B == 0) return;
    hipLaunchKernelGGL(k_segment_mean, dim3(B), dim3(64), 0, s, xn, ptr, base, out);
}
void pfk_step_build(const StepParams* sp, const BuildParams* bp, int fast, hipStream_t s) {
    if (sp->B == 0) return;
    if (fast) hipLaunchKernelGGL(k_step_build_fast, dim3(sp->B), dim3(512), 0, s, bp->prot_ptr, bp->pharm_ptr, bp->reg, bp->B, bp->Np_tot, *sp, *bp);
    else hipLaunchKernelGGL(k_step_build, dim3(sp->B), dim3(256), 0, s, *sp, *bp);
}
void pfk_step_update(const StepParams* p, hipStream_t s) {
    if (p->B == 0) return;
    hipLaunchKernelGGL(k_step_update, dim3(p->B), dim3(256), 0, s, *p);
}
void pfk_export_coords(const float4* xn, int base, int n, const int* gid, const float* add, const float* sub,
                       float* out, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_export_coords, dim3((n + 255) / 256), dim3(256), 0, s, xn, base, n, gid, add, sub, out);
}
void pfk_pp_radius(const float4* xn, const int* prot_ptr, int B, float r2, int maxn, int* deg, const int* row_off,
                   int* src, int* dst, int pass, hipStream_t s) {
    if (B == 0) return;
    hipLaunchKernelGGL(k_pp_radius, dim3(B), dim3(256), 0, s, xn, prot_ptr, r2, maxn, deg, row_off, src, dst, pass);
}
}
