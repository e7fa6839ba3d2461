// pf_cenhoist.h -- the center hoist's producer: one workgroup of 256 threads per four pharmacophore centers, a third kind of
// workgroup in the merged last launch of a denoising step (pf_rg.hip: k_rg_node_hs_build; CenHoistParams in pf_device.h has the why).
//
// For its centers it (1) takes eps_h of THIS step from a second copy of the head's exchange words, (2) applies the step's feature
// update (pharmacodiff.py:414-426: pf_feat_update, the expression the update + build runs) to the features as they were before the
// step -- read from a snapshot the previous step's update left, so that it never races with this launch's update --, (3) encodes them
// for the NEXT dynamics call's timestep (dynamics_gvp.py:107-117, 143-151: h_c = LayerNorm(SiLU(W [h, t] + b))), (4) multiplies h_c
// with the h_src block of the ff and fp etypes' first message Linear (gvp.py:545-549: P_et = W_et[:, :128] h_c + b_et).  Plain vector
// arithmetic: 4 x (128 x 7 + 2 x 128 x 128) multiply-adds on 256 threads are ~2 k cycles, the k-major weights (128 KB) stream
// coalesced from L2, and the whole item ends ~5 us after eps appears -- the update + build of the same launch ends ~7 us after it.
// The poll is bounded like the update + build's: a time-out is counted in the same word (the run is reported invalid) and the
// workgroup finishes on zeros.
#pragma once
#include <hip/hip_runtime.h>
#include "pf_device.h"

namespace pfch {

constexpr int CH_ROWS = 4;                      // centers per workgroup

struct __attribute__((aligned(16))) CenHoistLds {
    float hin[CH_ROWS][16];                     // updated features of the rows (zero beyond pharm_nf)
    float hc[CH_ROWS][128];                     // SiLU outputs, then the encoder outputs
    float part[CH_ROWS][4][2];                  // per wave: partial sums of a row's statistics
};

__device__ __forceinline__ float ch_silu(const float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ch_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// every thread of the (256-thread) workgroup calls it; item = which four centers
__device__ __forceinline__ void cen_hoist_item(const int item, const CenHoistParams& p, int* xstat, const int poll_sleep, const int poll_max,
                                               CenHoistLds& L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = CH_ROWS * item;
    const int nrows = min(CH_ROWS, p.Nf - c0);
    if (nrows <= 0) return;                                          // workgroup-uniform
    const int nf = p.nf;
    // ---- what does not depend on eps: the rows' old features and draws (threads 0..63: row tid >> 4, feature tid & 15), the encoder's
    // column of this thread's output feature, the LayerNorm parameters
    const int r = (tid >> 4) & 3, u = tid & 15;
    const bool mine = tid < 64 && r < nrows && u < nf;
    const int fr = c0 + min(r, nrows - 1), uc = min(u, nf - 1);
    float hv = 0.f, nz = 0.f;
    if (tid < 64) {
        hv = p.pharm_h[(size_t)fr * nf + uc];
        nz = p.noise[(size_t)fr * (3 + nf) + 3 + uc];
    }
    const int ft = tid & 127, half = tid >> 7;                       // output feature; rows 2 half, 2 half + 1 (encode) / etype (products)
    float we[16];                                                    // (pharm_nf <= 16; rows beyond it are read at a clamped index and not used)
#pragma unroll
    for (int k = 0; k < 16; ++k) we[k] = p.enc_w[(size_t)min(k, nf - 1) * 128 + ft];
    const float wtime = p.enc_w[(size_t)nf * 128 + ft];              // the timestep's input row
    const float be = p.enc_b[ft], lw = p.enc_lw[ft], lb = p.enc_lb[ft];
    // the thread's column of its etype's h_src block (etype = half): 128 weights in registers, requested NOW -- the workgroup has
    // ~14 us to wait for eps, and behind the wait only LDS reads and multiply-adds remain
    const float* wt = p.blk + (half ? L0C_WHT_FP : L0C_WHT_FF) + ft;
    const float bias = p.blk[(half ? L0C_B_FP : L0C_B_FF) + ft];
    float wk[128];
#pragma unroll
    for (int k = 0; k < 128; ++k) wk[k] = wt[(size_t)k * 128];
    // ---- eps_h of this step (a word is its own flag; bounded poll)
    float e = 0.f;
    if (mine) {
        unsigned int* w = p.xchg2 + (size_t)fr * PF_XCHG_STRIDE + u;
        unsigned int bits = PF_XCHG_EMPTY;
        for (int it = 0; it < poll_max; ++it) {
            bits = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bits != PF_XCHG_EMPTY) break;
            for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(2);
        }
        if (bits == PF_XCHG_EMPTY) atomicAdd(xstat, 1);
        else {
            e = __uint_as_float(bits);
            __hip_atomic_store(w, PF_XCHG_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < 64) L.hin[r][u] = mine ? pf_feat_update(hv, e, nz, p.a_ts, p.var, p.sigma, p.ep_zt, p.ep_pred, p.ep_feat) : 0.f;
    __syncthreads();
    // ---- encoder: feature ft of rows 2 half and 2 half + 1; LayerNorm statistics over the 128 features of a row = the two waves of a
    // half (mean first, then the centred squares: torch.nn.LayerNorm's biased variance, eps 1e-5)
    float s[2], cs[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = 2 * half + a;
        float z = be;
#pragma unroll
        for (int k = 0; k < 16; ++k) z = fmaf(we[k], k < nf ? L.hin[row][k] : 0.f, z);
        s[a] = ch_silu(fmaf(wtime, p.t_next, z));
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float ws = ch_wave_sum(s[a]);
        if (lane == 0) L.part[2 * half + a][wave][0] = ws;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = 2 * half + a;
        const float mean = (L.part[row][2 * half][0] + L.part[row][2 * half + 1][0]) * (1.0f / 128.0f);
        cs[a] = s[a] - mean;
        const float wq = ch_wave_sum(cs[a] * cs[a]);
        if (lane == 0) L.part[row][wave][1] = wq;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = 2 * half + a;
        const float var = (L.part[row][2 * half][1] + L.part[row][2 * half + 1][1]) * (1.0f / 128.0f);
        const float hc = cs[a] * __builtin_amdgcn_rsqf(var + 1e-5f) * lw + lb;
        L.hc[row][ft] = hc;
        if (row < nrows) p.cen_h[(size_t)(c0 + row) * 128 + ft] = hc;
    }
    __syncthreads();
    // ---- P_et[row][ft] = b_et[ft] + sum_k W_et[ft][k] h_c[row][k]: etype = half, k-major weights (coalesced), h_c from LDS (broadcast)
    float acc[CH_ROWS];
#pragma unroll
    for (int row = 0; row < CH_ROWS; ++row) acc[row] = bias;
#pragma unroll
    for (int k0 = 0; k0 < 128; k0 += 4) {
#pragma unroll
        for (int row = 0; row < CH_ROWS; ++row) {
            const float4 h0 = *reinterpret_cast<const float4*>(&L.hc[row][k0]);
            acc[row] = fmaf(wk[k0], h0.x, acc[row]); acc[row] = fmaf(wk[k0 + 1], h0.y, acc[row]);
            acc[row] = fmaf(wk[k0 + 2], h0.z, acc[row]); acc[row] = fmaf(wk[k0 + 3], h0.w, acc[row]);
        }
    }
#pragma unroll
    for (int row = 0; row < CH_ROWS; ++row)
        if (row < nrows) p.cen_p[((size_t)half * p.Nf + c0 + row) * 128 + ft] = acc[row];
}

}  // namespace pfch
