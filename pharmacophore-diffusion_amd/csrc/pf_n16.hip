// pf_n16.hip -- "n16" kernels of the denoising path (gfx950 only): an item is SIXTEEN rows (edge slots or nodes) on the
// FOUR waves of a workgroup, on v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles, the same 64 FLOP/clk/SIMD as every f32
// matrix instruction).
//
// Same mathematics as pf_rg.hip / pf_kernels.hip (GVP.forward gvp.py:89-116, GVPMultiEdgeConv gvp.py:459-551), third
// mapping onto the matrix cores.  Why: a 4-row item of pf_rg.hip pulls ALL weights of a GVP (96 KiB) through one wave
// for four rows -- ~115 cycles per 1-KiB quad where the matrix pipe needs 64, and 390 MB of L2 traffic per layer-0
// launch of config 2.  Here
//   * the OUTPUT features of every 128-output Linear are dealt over the four waves: wave w owns features [32 w, 32 w + 32)
//     = two 16 x 16 tiles and streams a QUARTER of the weights, for 16 rows: 1/16 of the weight bytes per row and wave,
//     and the four SIMDs of a CU work on one item;
//   * weights are the A operand (lane 16 g + i: output row i of the tile, k = 4 ks + g), activations the B operand
//     (lane 16 g + j: row j, k = 4 ks + g).  A D fragment (lane 16 g + j, register r: output 4 g + r of row j) is a valid B
//     operand of the next Linear as it stands when that Linear's weights are packed in the matching order of K, so inside
//     a wave nothing moves between lanes;
//   * what a wave lacks of the next GVP's input -- the other waves' 96 SiLU outputs per row, the three coordinates of Vh
//     for sh = |Vh|, the K-split partial sums of the 128 -> 16 gate Linear -- goes through 15 KiB of LDS behind two
//     LDS-only barriers per GVP, placed after long stretches of matrix instructions (arrival times are even);
//   * the vector channel is dealt by COORDINATE: wave c < 3 computes Vh, Vu and the gated output of coordinate c (wave 3
//     contributes the gate bias), so a coordinate's chain is lane-local from GVP to GVP;
//   * a wave's quarter of a chain is one contiguous quad stream in consumption order (pf_host.cpp: pack_n16),
//     prefetched N16_D quads ahead in registers across GVP boundaries, read through a buffer descriptor.
// Schedules: pf_device.h (n16_sched).  Launch policy: pf_host.cpp.  Measurements: DESIGN.md section 4.1.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include "pf_device.h"
#ifdef N16_STAMPS
#define N16_STAMP_GLOBALS 1
namespace {
__device__ unsigned long long* g_n16_stamps = nullptr;
__device__ int g_n16_stamp_off = 0;                   // first recorded workgroup
__device__ int g_n16_stamp_kid = -1;                  // kernel filter (see N16_STAMP)
}
// the update + build body of the tail launch stamps slots 40 + k of its wave (k: the build's own phase numbers, <= 15)
#define SB_STAMP(k)                                                                                                    \
    do {                                                                                                               \
        const int rb_ = (int)blockIdx.x - g_n16_stamp_off;                                                             \
        if ((k) < 16 && (threadIdx.x & 63) == 0 && g_n16_stamps && rb_ >= 0 && rb_ < 64 && (g_n16_stamp_kid < 0 || g_n16_stamp_kid == 3)) \
            g_n16_stamps[((size_t)rb_ * 4 + (threadIdx.x >> 6)) * 64 + 40 + (k)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#endif
#include "pf_stepbuild.h"

#include "pf_n16_core.h"

using namespace pfn16;

namespace {

// grid: one workgroup per 16-slot group (compact work list: the w-th non-empty group; tile lists: two groups per tile)
// The leading scalar arguments repeat the fields of p that the work-list scan needs: they are preloaded into scalar
// registers with the wave (-mllvm -amdgpu-kernarg-preload-count, Makefile), so the scan's loads leave without waiting for
// the kernel-argument segment (one dependent round trip of ~2,000 cycles less in front of every item).
template <bool L0>
__global__ __launch_bounds__(256) void k_n16_edge(const int* __restrict__ a_dyn_cnt, const int* __restrict__ a_reg, const int a_nreg,
                                                  const int a_regB, const int a_pa_abs, const EdgeParams p, const EncodeParams ep) {
    __shared__ N16Lds lds;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = (int)blockIdx.x;
    int sk = (L0 ? 0 : 1) << 8;
    N16_STAMP(sk, lane, wq);                              // kernel entry
    int e0, nv, et;
    if (a_nreg > 0) {
        // compact work list (see k_rg_edge): every wave of the workgroup finds the item's region by the same wave scan
        constexpr int NP = 16;                           // up to 64 * NP regions
        const int w = item;
        int first = 0, rsel = -1, cnt = 0, start = 0;
        int cs[NP], rs[NP];
        // (every pass's lengths and starts are requested before the first scan: one global round trip; the tests on the
        // pass are scalar, the lane's index is clamped instead of branched on)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            cs[k] = 0; rs[k] = 0;
            if (64 * k < a_nreg) {
                const int r = min(64 * k + lane, a_nreg - 1);
                int c = a_dyn_cnt[r];
                if (L0 && p.pa_skip && r >= 3 * a_regB && p.pa_skip[r - 3 * a_regB] >= ((c + 15) >> 4)) c = 0;      // (see k_n16_edge_u; a partly valid region is computed whole here)
                rs[k] = a_reg[r];
                cs[k] = 64 * k + lane < a_nreg ? c : 0;
            }
        }
        const int abs0 = a_pa_abs ? 3 * a_regB : 0x7fffffff;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (64 * k < a_nreg && rsel < 0) {
                const int c = cs[k];
                const int ng = (64 * k + lane >= abs0) ? (c > 0 ? ((rs[k] + c - 1) >> 4) - (rs[k] >> 4) + 1 : 0) : (c + 15) >> 4;
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
        if (rsel < 0) return;                            // workgroup-uniform: beyond the last group
        const int kind = rsel / a_regB;
        if (a_pa_abs && kind == 3) {
            const int lo = ((start >> 4) + (w - first)) << 4;
            e0 = max(start, lo);
            nv = __builtin_amdgcn_readfirstlane(min(start + cnt, lo + 16) - e0);
        } else {
            const int loc = (w - first) << 4;
            e0 = start + loc;
            nv = __builtin_amdgcn_readfirstlane(min(16, cnt - loc));
        }
        et = kind == 3 ? (int)ET_PP : kind;
    } else {
        if (item >= p.ntiles * 2) return;
        const EdgeTile t = p.tiles[item >> 1];
        int nvalid = t.n;
        if (t.cnt_idx >= 0) nvalid = min(nvalid, max(p.dyn_cnt[t.cnt_idx] - t.rel, 0));
        const int base = (item & 1) * 16;
        nv = __builtin_amdgcn_readfirstlane(min(16, nvalid - base));
        if (nv <= 0) return;                             // workgroup-uniform
        et = __builtin_amdgcn_readfirstlane(t.et);
        e0 = t.e0 + base;
    }
#ifdef N16_TRACE
    if (g_n16_trace && threadIdx.x == 0) {
        g_n16_trace[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        g_n16_trace[(size_t)blockIdx.x * 4 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        g_n16_trace[(size_t)blockIdx.x * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) | ((unsigned long long)et << 32) | ((unsigned long long)nv << 40);
    }
#endif
    N16_CUT_AT(EDGE_CUT, 1, (float)(e0 + nv), p.msg_s);
    if constexpr (L0) {
        // conv layer 0: every node vector is zero; protein sources (pf, pp) read the type tables of the static hoist
        // (h_src is one of rec_nf encoder outputs per t), centers (ff, fp) are encoded on the fly
        if (et == ET_PP && p.need) {
            // pocket sharing: this group of a representative's static pp edges runs only if some copy of the pocket reads
            // one of its destinations in this step
            const int e = e0 + min(lane & 15, nv - 1);
            if (!__any(p.need[p.edst[e]] == p.need_stamp)) return;      // workgroup-uniform (every wave tests the same rows)
        }
        const int e = e0 + min(lane & 15, nv - 1);
        const int src = p.esrc[e], dst = p.edst[e];
        if (et == ET_PP || et == ET_PF || p.pcen) n16_edge_item<N16_M0H>(p, ep, &lds, e, src, dst, nv, et, lane, wq, sk);
        else n16_edge_item<N16_M0Z>(p, ep, &lds, e, src, dst, nv, et, lane, wq, sk);
    } else {
        const int e = e0 + min(lane & 15, nv - 1);
        n16_edge_item<N16_M0F>(p, ep, &lds, e, p.esrc[e], p.edst[e], nv, et, lane, wq, sk);
    }
#ifdef N16_TRACE
    if (g_n16_trace && threadIdx.x == 0) g_n16_trace[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
#endif
}

// Conv layer 0 of a batch whose graphs all have ff / pf / fp regions of one capacity (the same number of centers everywhere; no pocket
// sharing): the items of those regions -- the long ones: full first GVP on encoded centers, or the pf type-table form -- are mapped
// by arithmetic on preloaded scalars, and with the edge arrays' addresses preloaded too their slots are requested the moment the
// wave starts (groups beyond a region's count leave at once); the "pa" items behind them keep the work-list scan, over B regions
// instead of 4 B.  Same item order as k_n16_edge's compact list: ff, pf, fp, pa.
//   a_s01 = stride_ff | stride_pf << 16; a_s2g = stride_fp | groups_ff << 16 | groups_pf << 19 | groups_fp << 22 | (graphs - 1) << 25
//   (the first eight arguments are preloaded: everything the static items' first loads need; a_reg, for the "pa" scan, is not)
__global__ __launch_bounds__(256) void k_n16_edge_u(const int* __restrict__ a_dyn_cnt, const int* __restrict__ a_esrc,
                                                    const int* __restrict__ a_edst, const int a_b0, const int a_b1, const int a_b2,
                                                    const int a_s01, const int a_s2g, const int* __restrict__ a_reg,
                                                    const EdgeParams p, const EncodeParams ep) {
    __shared__ N16Lds lds;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 15;
    int sk = 0;
    N16_STAMP(sk, lane, wq);                              // kernel entry
#ifdef N16_STAMPS                                         // (kernel filter 100: start / end of every workgroup on the device-wide 100 MHz clock instead)
    struct WgTimes { __device__ ~WgTimes() { if (on) g_n16_stamps[(size_t)blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); } bool on; } wgt_;
    wgt_.on = g_n16_stamp_kid == 100 && g_n16_stamps && threadIdx.x == 0 && blockIdx.x < 8192;
    if (wgt_.on) g_n16_stamps[(size_t)blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
    const int gff = (a_s2g >> 16) & 7, gpf = (a_s2g >> 19) & 7, gfp = (a_s2g >> 22) & 7, a_B = (int)((unsigned)a_s2g >> 25) + 1;
    const int nff = a_B * gff, npf = a_B * gpf, nfp = a_B * gfp;
    int w = (int)blockIdx.x;
#ifdef EDGE_U_SWAP                                        // (diagnostic: ff and fp items trade places in the grid; needs nff == nfp)
    if (w < nff) w += nff + npf; else if (w >= nff + npf && w < nff + npf + nfp) w -= nff + npf;
#endif
    if (w < nff + npf + nfp) {
        int et, base, stride, gk;
        if (w < nff) { et = ET_FF; base = a_b0; stride = a_s01 & 0xffff; gk = gff; }
        else if (w < nff + npf) { w -= nff; et = ET_PF; base = a_b1; stride = (int)((unsigned)a_s01 >> 16); gk = gpf; }
        else { w -= nff + npf; et = ET_FP; base = a_b2; stride = a_s2g & 0xffff; gk = gfp; }
        const int g = w / gk, k = w - g * gk;
        const int e0 = base + g * stride + 16 * k;
        // one level: the region's count and the group's 16 slots (inside the region's 32-aligned capacity; slots beyond the count
        // hold stale ids, which are never dereferenced: rows beyond nv shadow row nv - 1)
        const int cnt = a_dyn_cnt[et * a_B + g];
        const int src_raw = a_esrc[e0 + j], dst_raw = a_edst[e0 + j];
        const int nv = __builtin_amdgcn_readfirstlane(min(16, cnt - 16 * k));
        if (nv <= 0) return;                             // workgroup-uniform
        const int jj = min(j, nv - 1);
        const int from = 4 * ((lane & 48) | jj);
        const int src = __builtin_amdgcn_ds_bpermute(from, src_raw), dst = __builtin_amdgcn_ds_bpermute(from, dst_raw);
        if (et == ET_PF || p.pcen) n16_edge_item<N16_M0H>(p, ep, &lds, e0 + jj, src, dst, nv, et, lane, wq, sk);
        else n16_edge_item<N16_M0Z>(p, ep, &lds, e0 + jj, src, dst, nv, et, lane, wq, sk);
        return;
    }
    // ---- "pa" items: the (w - static items)-th non-empty 16-slot group of the B regions of kind 3 (one scan pass: B <= 64)
    w -= nff + npf + nfp;
    const int r = 3 * a_B + min(lane, a_B - 1);
    const int c = lane < a_B ? a_dyn_cnt[r] : 0;
    // the leading groups of this graph's region whose rows were computed ahead by the previous step's last launch and still apply
    // (BuildParams::pa_same: all of them when the graph's active set did not change, those in front of the first change otherwise)
    const int keep = p.pa_skip ? min(p.pa_skip[min(lane, a_B - 1)], (c + 15) >> 4) : 0;
    const int rs = a_reg[r];
    const int ng = ((c + 15) >> 4) - keep;
    int incl = ng;
    incl += dpp_i<0x111>(incl); incl += dpp_i<0x112>(incl); incl += dpp_i<0x114>(incl); incl += dpp_i<0x118>(incl);
    incl += dpp_ir<0x142, 0xa>(incl); incl += dpp_ir<0x143, 0xc>(incl);
    const unsigned long long m = __ballot(incl > w);
    if (!m) return;                                      // workgroup-uniform: beyond the last group
    const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
    const int first = __builtin_amdgcn_readlane(incl - ng, l);
    const int cnt = __builtin_amdgcn_readlane(c, l), start = __builtin_amdgcn_readlane(rs, l);
    const int loc = (w - first + __builtin_amdgcn_readlane(keep, l)) << 4;
    const int e0 = start + loc;
    const int nv = __builtin_amdgcn_readfirstlane(min(16, cnt - loc));
    const int e = e0 + min(j, nv - 1);
    n16_edge_item<N16_M0H>(p, ep, &lds, e, a_esrc[e], a_edst[e], nv, (int)ET_PP, lane, wq, sk);
}

// ---------------------------------------------------------------------------------------------
// GVPLayerNorm (gvp.py:159-166) on the n16 layouts.  Every wave holds all 128 scalars of its rows (32 registers x the four
// lane groups): the row statistics are a register sum and one cross-group sum.  The vector norm needs the three
// coordinates of a channel, which live on three waves: the squares meet in LDS (lds->vn; one barrier).  Called by all
// four waves (wave 3: VB = 0).
// ---------------------------------------------------------------------------------------------
// The LayerNorm parameters of a node update (2 x (weight, bias) x 128) are requested by the item's 256 threads the moment it starts,
// two floats each, and parked in LDS in front of the first LayerNorm's barrier: loaded where they are used -- behind the chain's
// scheduling barriers the compiler cannot move them up -- each LayerNorm opened with a cold round trip (~1 us).
struct N16LnPre { float a, b; };
__device__ __forceinline__ N16LnPre n16_ln_request(pf_gcf w1, pf_gcf b1, pf_gcf w2, pf_gcf b2, const int lane, const int wq) {
    const int t = wq * 64 + lane;                         // 0..255: arrays 0 / 1 (first LayerNorm), then 2 / 3
    N16LnPre q;
    q.a = (t < 128 ? w1 : b1)[t & 127];
    q.b = (t < 128 ? w2 : b2)[t & 127];
    return q;
}
__device__ __forceinline__ void n16_ln_stage(const N16LnPre& q, N16Lds* lds, const int lane, const int wq) {      // (a barrier follows at the caller)
    const int t = wq * 64 + lane;
    lds->ln[t] = q.a;
    lds->ln[256 + t] = q.b;
}
template <bool FROM_LDS = false>
__device__ __forceinline__ void n16_layernorm(pf_gcf lw, pf_gcf lb, float (&XS)[32], float (&VB)[4], N16Lds* lds, const int lane, const int wq,
                                              const int which = 0) {
    const int g = lane >> 4;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) sum += XS[k];
    const float mean = gsum(sum) * (1.0f / 128.0f);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) { const float c = XS[k] - mean; var = fmaf(c, c, var); }
    const float rstd = rsqf_(gsum(var) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        f32x4 w, b;
        if constexpr (FROM_LDS) {
            w = *reinterpret_cast<const f32x4*>(&lds->ln[256 * which + 16 * T + 4 * g]);
            b = *reinterpret_cast<const f32x4*>(&lds->ln[256 * which + 128 + 16 * T + 4 * g]);
        } else {
            w = *reinterpret_cast<const f32x4 PF_AS1*>(lw + 16 * T + 4 * g);
            b = *reinterpret_cast<const f32x4 PF_AS1*>(lb + 16 * T + 4 * g);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) XS[4 * T + r] = (XS[4 * T + r] - mean) * rstd * w[r] + b[r];
    }
    if (wq < 3) *reinterpret_cast<f32x4*>(&lds->vn[(wq * 64 + lane) * 4]) = (f32x4){VB[0] * VB[0], VB[1] * VB[1], VB[2] * VB[2], VB[3] * VB[3]};
    lds_barrier();
    const f32x4 a = *reinterpret_cast<const f32x4*>(&lds->vn[(0 * 64 + lane) * 4]);
    const f32x4 b = *reinterpret_cast<const f32x4*>(&lds->vn[(1 * 64 + lane) * 4]);
    const f32x4 c = *reinterpret_cast<const f32x4*>(&lds->vn[(2 * 64 + lane) * 4]);
    float vn = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) vn += fmaxf(a[r] + b[r] + c[r], 1e-8f);
    const float rden = rcpf_(sqrtf_(gsum(vn) * (1.0f / 16.0f) + 1e-5f) + 1e-5f);
#pragma unroll
    for (int r = 0; r < 4; ++r) VB[r] *= rden;
}

// ---------------------------------------------------------------------------------------------
// Conv layer 0's node update (gvp.py:488-536) of the 16 rows `node` (one node per lane row; type nt workgroup-uniform):
// sum of the node's message partial rows (mean per etype or the configured norm), residual on the encoder output (the
// node vectors of conv layer 0 are zero), GVPLayerNorm, the update chain (ring: the next blocks of the stream), residual,
// GVPLayerNorm.  Leaves h' in XS (every wave: all 128 features) and v' in VB (wave c: coordinate c) -- exactly what the first
// message GVP of the NEXT layer's edges takes, so nothing is written (k_n16_fused).
// ---------------------------------------------------------------------------------------------
// where a node finds the partial rows of its in-edges: two segments [st, st + cn) of edge slots; the partial rows of a
// segment are the last slots of the aligned groups of (gm + 1) slots it touches
struct NodeDesc { int st[2], cn[2], gm[2]; };
// Sum of a node's partial rows, each segment scaled by 1 / count under message_norm = 'mean' (norm_mode 0).  A wave gathers
// only ITS quarter of the scalars -- tiles T = 2 wq + t, features 32 wq + 16 t + 4 g + r of row j, the D layout of its own
// outputs -- and coordinate wq of the vectors: all four waves fetching whole rows (what the B operands need) moves 4 x the
// bytes through one compute unit's memory path, and at ~35-70 GB/s per CU that, not latency, is what a gather costs here
// (measured: 4.4 us of the fused launch).  The quarters meet in LDS afterwards (n16_quarters_to_rows).
// The first two partial rows of each segment leave in ONE batch of loads (absent: the all-zero row) -- at these in-degrees
// that is all of them -- and longer segments finish in a loop.
// The VECTOR partial rows (48 floats) are fetched whole, ONE of the four first rows per wave -- wave w: segment w >> 1, row w & 1;
// lane (g, j): floats [12 g, 12 g + 12) of node j's row, three 16-byte loads -- and meet in LDS (n16_rows_sum): every wave
// picking its coordinate out of every row with dword loads at a 12-byte stride cost ~900 cache-line requests per wave and item
// against ~130 for the scalars, and a compute unit's vector memory path takes about one request per cycle.
struct RowQ {
    f32x4 x[2][2][2];                           // [segment][row k][tile t]
    f32x4 vq[3];                                // this wave's vector partial row, floats [12 g, 12 g + 12)
    int nxt[2];
};
__device__ __forceinline__ void n16_rows_load(const float* msg_s, const float* msg_v, const int zero_row, const NodeDesc& nd, RowQ& q,
                                              const int lane, const int wq) {
    const int g = lane >> 4;
    int rwv = zero_row;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int end = nd.st[sl] + nd.cn[sl], gm = nd.gm[sl];
        int e = nd.st[sl];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool has = e < end;
            const int rw = has ? min(e | gm, end - 1) : zero_row;
            const f32x4 PF_AS1* mp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)msg_s + (size_t)rw * PF_S + 32 * wq) + g;
            q.x[sl][k][0] = mp[0]; q.x[sl][k][1] = mp[4];
            rwv = (2 * sl + k == wq) ? rw : rwv;         // (wave-uniform choice of the row this wave fetches the vectors of)
            e = has ? rw + 1 : e;
        }
        q.nxt[sl] = e;
    }
    const f32x4 PF_AS1* vp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)msg_v + (size_t)rwv * 48 + 12 * g);
    q.vq[0] = vp[0]; q.vq[1] = vp[1]; q.vq[2] = vp[2];
}
// Q: this wave's two tiles of the sum; VB: coordinate wq of the vector sum
__device__ __forceinline__ void n16_rows_sum(const float* msg_s, const float* msg_v, const int zero_row, const int norm_mode,
                                             const NodeDesc& nd, const RowQ& q, f32x4 (&Q)[2], float (&VB)[4], N16Lds* lds, const int lane,
                                             const int wq) {
    const int g = lane >> 4, j = lane & 15;
    const int cw = wq < 3 ? wq : 0;
    // the four waves' vector rows meet: vx[wave][node j][48]; one barrier (the caller keeps barriers between this read and the
    // next write: the node update's LayerNorms and blocks)
    {
        float* vw = &lds->vx[(wq * 16 + j) * 48 + 12 * g];
        *reinterpret_cast<f32x4*>(vw) = q.vq[0];
        *reinterpret_cast<f32x4*>(vw + 4) = q.vq[1];
        *reinterpret_cast<f32x4*>(vw + 8) = q.vq[2];
    }
    lds_barrier();
    float vv[2][2][4];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[sl][k][r] = lds->vx[((2 * sl + k) * 16 + j) * 48 + 12 * g + 3 * r + cw];
    Q[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; Q[1] = Q[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) VB[r] = 0.f;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int end = nd.st[sl] + nd.cn[sl], gm = nd.gm[sl];
        f32x4 ps0 = q.x[sl][0][0] + q.x[sl][1][0], ps1 = q.x[sl][0][1] + q.x[sl][1][1];
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = vv[sl][0][r] + vv[sl][1][r];
        int e = q.nxt[sl];
        while (__any(e < end)) {                          // (rows that are done add the all-zero row)
            const bool has = e < end;
            const int rw = has ? min(e | gm, end - 1) : zero_row;
            const f32x4 PF_AS1* mp = reinterpret_cast<const f32x4 PF_AS1*>((pf_gcf)msg_s + (size_t)rw * PF_S + 32 * wq) + g;
            const f32x4 y0 = mp[0], y1 = mp[4];
            pf_gcf vp = (pf_gcf)msg_v + (size_t)rw * 48 + 12 * g + cw;
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = vp[3 * r];
            ps0 += y0; ps1 += y1;
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] += yv[r];
            e = has ? rw + 1 : e;
        }
        const float sc = (norm_mode == 0 && nd.cn[sl] > 0) ? 1.0f / (float)nd.cn[sl] : 1.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { Q[0][r] = fmaf(ps0[r], sc, Q[0][r]); Q[1][r] = fmaf(ps1[r], sc, Q[1][r]); VB[r] = fmaf(pv[r], sc, VB[r]); }
    }
}
// the four waves' quarters of 16 rows -> every wave holds the whole rows as B operands (XS[4 T + r]: feature 16 T + 4 g + r).
// One barrier; the caller keeps another barrier between this function's reads and the next write of lds->s.
__device__ __forceinline__ void n16_quarters_to_rows(const f32x4 (&Q)[2], float (&XS)[32], N16Lds* lds, const int lane, const int wq) {
    *reinterpret_cast<f32x4*>(&lds->s[((2 * wq) * 64 + lane) * 4]) = Q[0];
    *reinterpret_cast<f32x4*>(&lds->s[((2 * wq + 1) * 64 + lane) * 4]) = Q[1];
    lds_barrier();
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(&lds->s[(T * 64 + lane) * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) XS[4 * T + r] = x[r];
    }
}

// descriptors of a node's in-edges in conv layer 0 (slot 0: ff | fp; second segment: pf of a center, the pp / "pa" edges of an atom)
__device__ __forceinline__ void n16_node_desc_l0(const FusedParams& f, const int node, const int nt, NodeDesc& nd) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int slot = sl == 0 ? 0 : (nt == 0 ? f.pp_slot : 1);
        nd.st[sl] = f.in_start[slot * f.N + node];
        nd.cn[sl] = f.in_cnt[slot * f.N + node];
        nd.gm[sl] = (sl == 0 ? f.grp : (nt == 0 ? f.grp_pa : f.grp)) - 1;
    }
}
// (nd: n16_node_desc_l0 of the node; pty: f.ptype[node] of an atom -- both requested by the caller together with whatever
// else it loads at that level, so that a workgroup's prologue is four dependent round trips: work list, edge slots, descriptors
// + coordinates + types, rows)
__device__ __forceinline__ void n16_node_update_l0(const FusedParams& f, const EncodeParams& ep, N16Ring& ring, const int node, const int nt,
                                                   const NodeDesc& nd, const int pty, float (&XS)[32], float (&VB)[4], N16Lds* lds,
                                                   const int lane, const int wq, int& sk) {
    const int g = lane >> 4;
    // one batch of loads: the partial rows (this wave's quarter) and, for an atom, its quarter of the residual input -- the
    // encoder output of its element type, a row of the timestep's type table
    const N16LnPre lnq = n16_ln_request(f.ln1_w[nt], f.ln1_b[nt], f.ln2_w[nt], f.ln2_b[nt], lane, wq);
    RowQ rq;
    n16_rows_load(f.msg_s, f.msg_v, f.zero_row, nd, rq, lane, wq);
    f32x4 Hq[2];
    Hq[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; Hq[1] = Hq[0];
    if (nt == 0) {
        pf_gcf hp = (pf_gcf)f.htab + (f.htab_gstride ? (size_t)f.gid[node] * f.htab_gstride : 0) + (size_t)pty * PF_S + 32 * wq + 4 * g;
        Hq[0] = *reinterpret_cast<const f32x4 PF_AS1*>(hp);
        Hq[1] = *reinterpret_cast<const f32x4 PF_AS1*>(hp + 16);
    }
    float inv_norm = 1.0f;
    if (f.norm_mode == 1) inv_norm = 1.0f / f.norm_value;
    else if (f.norm_mode == 2) inv_norm = 1.0f / f.gnorm[nt * f.B + f.gid[node]];
    // a center's residual input: a row of the center hoist's table (this wave's quarter, with the row loads), or encoded on the fly
    // (its own exchange through lds->s, closed by a barrier) under the row loads
    float H[32];
    const bool enc_here = nt != 0 && f.hcen == nullptr;                 // workgroup-uniform
    if (nt != 0 && f.hcen != nullptr) {
        pf_gcf hp = (pf_gcf)f.hcen + (size_t)(node - ep.Np) * PF_S + 32 * wq + 4 * g;
        Hq[0] = *reinterpret_cast<const f32x4 PF_AS1*>(hp);
        Hq[1] = *reinterpret_cast<const f32x4 PF_AS1*>(hp + 16);
    }
    if (enc_here) {
        const float tt = ep.t ? ((pf_gcf)ep.t)[f.gid[node]] : ep.t_scalar;
        n16_encode_pharm(ep, (pf_gcf)ep.pharm_h + (size_t)(node - ep.Np) * ep.pharm_nf, tt, H, lds, lane, wq);
    }
    f32x4 Q[2];
    n16_rows_sum(f.msg_s, f.msg_v, f.zero_row, f.norm_mode, nd, rq, Q, VB, lds, lane, wq);
    N16_STAMP(sk, lane, wq);                              // partial rows summed
    N16_CUT_AT(FUSED_CUT, 2, Q[0][0] + VB[0] + Hq[0][0], f.h_out);
#pragma unroll
    for (int r = 0; r < 4; ++r) { Q[0][r] = fmaf(Q[0][r], inv_norm, Hq[0][r]); Q[1][r] = fmaf(Q[1][r], inv_norm, Hq[1][r]); }
    n16_ln_stage(lnq, lds, lane, wq);                    // (in front of the exchange's barrier)
    n16_quarters_to_rows(Q, XS, lds, lane, wq);          // (the first LayerNorm's barrier closes the reads)
    if (enc_here) {
#pragma unroll
        for (int k = 0; k < 32; ++k) XS[k] += H[k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) VB[r] = wq < 3 ? VB[r] * inv_norm : 0.f;
    n16_layernorm<true>(nullptr, nullptr, XS, VB, lds, lane, wq, 0);
    N16_STAMP(sk, lane, wq);                              // residual input + first LayerNorm
    float Xr[32], Vr[4];
#pragma unroll
    for (int k = 0; k < 32; ++k) Xr[k] = XS[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) Vr[r] = VB[r];
    N16In none{};
    f32x4 S[2];
    S[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; S[1] = S[0];
    N16P XP;
    n16_gen_run<0>(f.n_upd, ring, XS, XP, VB, none, S, lds, lane, wq, sk);
    n16_gate_flush(VB, lds, lane, wq);
#pragma unroll
    for (int k = 0; k < 32; ++k) XS[k] += Xr[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) VB[r] += Vr[r];
    n16_layernorm<true>(nullptr, nullptr, XS, VB, lds, lane, wq, 1);
}

// ---- store item of the fused launch: the centers [16 part, 16 part + 16) of graph gq
__device__ __forceinline__ void n16_fused_store_item(const FusedParams& f, const EncodeParams& ep, N16Lds* lds, const int gq, const int part,
                                                     float (&XS)[32], float (&VB)[4], const int lane, const int wq, int& sk) {
    const int g = lane >> 4, j = lane & 15;
    const int f0 = f.pharm_ptr[gq], nf = f.pharm_ptr[gq + 1] - f0;
    const int nv = __builtin_amdgcn_readfirstlane(min(16, nf - 16 * part));
    if (nv <= 0) return;                                 // workgroup-uniform
    const int node = f.Np + f0 + 16 * part + min(j, nv - 1);
    N16Ring ring;
    ring_start(ring, f.upd_pharm + (size_t)wq * f.upd_pharm_stride, lane);
    NodeDesc nd;
    n16_node_desc_l0(f, node, 1, nd);
    n16_node_update_l0(f, ep, ring, node, 1, nd, 0, XS, VB, lds, lane, wq, sk);
    if (j < nv && wq == 0) {
        float* hp = f.h_out + (size_t)node * PF_S + 4 * g;
#pragma unroll
        for (int T = 0; T < 8; ++T) *reinterpret_cast<f32x4*>(hp + 16 * T) = (f32x4){XS[4 * T], XS[4 * T + 1], XS[4 * T + 2], XS[4 * T + 3]};
    }
    if (j < nv && wq < 3) {
        float* vp = f.v_out + (size_t)node * 48 + 12 * g + wq;
#pragma unroll
        for (int r = 0; r < 4; ++r) vp[3 * r] = VB[r];
    }
}
// ---- edge item of the fused launch: row j = edge slot e (source src, destination dst; rows beyond nv shadow row nv - 1)
// rec: the row's edge record (FusedParams::rec: three 16-byte words) or NULL
__device__ __forceinline__ void n16_fused_edge_item(const EdgeParams& p, const FusedParams& f, const EncodeParams& ep, N16Lds* lds, const int e,
                                                    const int src, const int dst, const int nv, const int et, float (&XS)[32],
                                                    float (&VB)[4], const int lane, const int wq, int& sk, const int4* rec = nullptr) {
    N16Ring ring;
    ring_start(ring, f.chain[et] + (size_t)wq * f.chain_stride[et], lane);
    N16Rows rw;
    rw.e = e;
    rw.dst = dst;
    const int nt = et == ET_FF ? 1 : 0;
    NodeDesc nd;
    int pty;
    if (rec) {                                           // (workgroup-uniform) the slot's edge record: everything of this level, already here
        rw.xs = make_float4(__builtin_bit_cast(float, rec[0].x), __builtin_bit_cast(float, rec[0].y), __builtin_bit_cast(float, rec[0].z), 0.f);
        rw.xd = make_float4(__builtin_bit_cast(float, rec[1].x), __builtin_bit_cast(float, rec[1].y), __builtin_bit_cast(float, rec[1].z), 0.f);
        nd.st[0] = rec[2].x; nd.cn[0] = rec[2].y; nd.st[1] = rec[2].z; nd.cn[1] = rec[2].w;
        nd.gm[0] = f.grp - 1; nd.gm[1] = (nt == 0 ? f.grp_pa : f.grp) - 1;
        pty = (int)((unsigned)rec[0].w >> 24);
    } else {
        // one level of loads: coordinates, the source's in-edge descriptors, its element type
        rw.xs = p.xn[src]; rw.xd = p.xn[dst];
        n16_node_desc_l0(f, src, nt, nd);
        pty = nt == 0 ? f.ptype[src] : 0;
    }
    n16_node_update_l0(f, ep, ring, src, nt, nd, pty, XS, VB, lds, lane, wq, sk);
    N16_CUT_AT(FUSED_CUT, 3, XS[0] + VB[0] + rw.xs.x + rw.xd.x, f.h_out);
    f32x4 S[2];
    S[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; S[1] = S[0];
    n16_edge_chain<N16_M0F>(p, ring, rw, XS, VB, S, lds, nv, lane, wq, sk);
}

// The fused launch (FusedParams): items [0, n_edge_items) are the last conv layer's 16-slot edge groups (compact work list:
// ff and pf regions); the remaining workgroups store conv layer 0's update of the centers, 16 per item, for the node + head
// launch.  An edge item first updates its own source rows (conv layer 0's node update, weights: the first blocks of its
// stream), then runs its message chain on them.
__global__ __launch_bounds__(256) void k_n16_fused(const int* __restrict__ a_dyn_cnt, const int* __restrict__ a_reg, const int a_nreg,
                                                   const int a_regB, const EdgeParams p, const FusedParams f, const EncodeParams ep) {
    __shared__ N16Lds lds;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 15;
    int sk = 2 << 8;
    N16_STAMP(sk, lane, wq);                              // kernel entry
    float XS[32], VB[4];
    auto store_item = [&](const int gq, const int part) { n16_fused_store_item(f, ep, &lds, gq, part, XS, VB, lane, wq, sk); };
    int e0, nv, et;
    if (f.xcd_split) {
        // XCD-aware assignment (workgroup b runs on XCD b % 8): ff edge items and the store items -- the centers' side: they stream
        // the same update chain -- on XCDs 0..3, pf edge items on XCDs 4..7, so that an XCD's L2 fetches half of the launch's
        // weights.  One scan pass over the 2 B regions (the host sets xcd_split only when they fit one pass and every graph has
        // at most 16 centers).
        const int b = (int)blockIdx.x, x = b & 7, kq = b >> 3;
        const int r = min(lane, a_nreg - 1);
        const int c = lane < a_nreg ? a_dyn_cnt[r] : 0;
        const int rs0 = a_reg[r];
        const int ng = (c + 15) >> 4;
        int incl = ng;
        incl += dpp_i<0x111>(incl); incl += dpp_i<0x112>(incl); incl += dpp_i<0x114>(incl); incl += dpp_i<0x118>(incl);
        incl += dpp_ir<0x142, 0xa>(incl); incl += dpp_ir<0x143, 0xc>(incl);
        const int ff_total = __builtin_amdgcn_readlane(incl, a_regB - 1), all_total = __builtin_amdgcn_readlane(incl, 63);
        int w;
        if (x < 4) {
            const int idx = 4 * kq + x;
            if (idx >= ff_total) {
                if (idx - ff_total < f.B) store_item(idx - ff_total, 0);
                return;
            }
            w = idx;
        } else {
            w = ff_total + 4 * kq + (x - 4);
            if (w >= all_total) return;
        }
        const unsigned long long m = __ballot(incl > w);
        const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
        const int first = __builtin_amdgcn_readlane(incl - ng, l);
        const int cnt = __builtin_amdgcn_readlane(c, l), start = __builtin_amdgcn_readlane(rs0, l);
        et = l / a_regB;
        const int loc = (w - first) << 4;
        e0 = start + loc;
        nv = __builtin_amdgcn_readfirstlane(min(16, cnt - loc));
    } else {
    if ((int)blockIdx.x >= f.n_edge_items) {
        const int it = (int)blockIdx.x - f.n_edge_items;
        store_item(it / (PF_MAXF / 16), it % (PF_MAXF / 16));
        return;
    }
    // ---- edge item: the w-th non-empty 16-slot group of the launch's regions (see k_n16_edge)
    {
        constexpr int NP = 16;
        const int w = (int)blockIdx.x;
        int first = 0, rsel = -1, cnt = 0, start = 0;
        int cs[NP], rs[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            cs[k] = 0; rs[k] = 0;
            if (64 * k < a_nreg) {
                const int r = min(64 * k + lane, a_nreg - 1);
                const int c = a_dyn_cnt[r];
                rs[k] = a_reg[r];
                cs[k] = 64 * k + lane < a_nreg ? c : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (64 * k < a_nreg && rsel < 0) {
                const int c = cs[k];
                const int ng = (c + 15) >> 4;
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
        if (rsel < 0) return;                            // workgroup-uniform: beyond the last group
        et = rsel / a_regB;                              // the last layer's regions: ff, pf
        const int loc = (w - first) << 4;
        e0 = start + loc;
        nv = __builtin_amdgcn_readfirstlane(min(16, cnt - loc));
    }
    }
    N16_STAMP(sk, lane, wq);                              // item known
    N16_CUT_AT(FUSED_CUT, 1, (float)(e0 + nv), f.h_out);
    const int e = e0 + min(j, nv - 1);
    n16_fused_edge_item(p, f, ep, &lds, e, p.esrc[e], p.edst[e], nv, et, XS, VB, lane, wq, sk);
}

// The same launch when every graph of the batch has regions of one capacity (all graphs with the same number of centers: the
// headline configuration, batches of copies of one size): the regions of an etype then sit at a fixed stride (pf_host.cpp lays
// them out in order, 32-slot aligned), so the item -> (etype, graph, 16-slot group) map is arithmetic on PRELOADED scalars, and with
// the edge arrays' addresses preloaded too the item's slots are requested the moment the wave starts -- neither the work-list
// round trip nor the kernel-argument segment's stands in front of them (18.37 -> 18.05 us at config 2).
// Groups beyond a region's count leave at once (capacity tiling).  XCD assignment as in k_n16_fused's xcd_split form.
//   a_strides = stride_ff | stride_pf << 16 (slots between consecutive graphs' regions);
//   a_groups  = groups_ff | groups_pf << 8 | B << 16
__global__ __launch_bounds__(256) void k_n16_fused_u(const int* __restrict__ a_dyn_cnt, const int* __restrict__ a_esrc,
                                                     const int* __restrict__ a_edst, const int a_ff_base, const int a_pf_base,
                                                     const int a_strides, const int a_groups, const EdgeParams p,
                                                     const FusedParams f, const EncodeParams ep) {
    __shared__ N16Lds lds;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 15;
    int sk = 2 << 8;
    N16_STAMP(sk, lane, wq);                              // kernel entry
#ifdef N16_STAMPS                                         // (kernel filter 101: start / end of every workgroup on the device-wide 100 MHz clock)
    struct WgTimes { __device__ ~WgTimes() { if (on) g_n16_stamps[(size_t)blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); } bool on; } wgt_;
    wgt_.on = g_n16_stamp_kid == 101 && g_n16_stamps && threadIdx.x == 0 && blockIdx.x < 8192;
    if (wgt_.on) g_n16_stamps[(size_t)blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
    float XS[32], VB[4];
    const int nffg = a_groups & 255, npfg = (a_groups >> 8) & 255, a_B = (int)((unsigned)a_groups >> 16);
    const int b = (int)blockIdx.x, x = b & 7, kq = b >> 3;
    int et, e0, cidx, k;
    if (x < 4) {
        const int idx = 4 * kq + x;
        if (idx >= a_B * nffg) {
            if (idx - a_B * nffg < a_B) n16_fused_store_item(f, ep, &lds, idx - a_B * nffg, 0, XS, VB, lane, wq, sk);
            return;
        }
        const int g = idx / nffg;
        k = idx - g * nffg;
        et = ET_FF; cidx = g; e0 = a_ff_base + g * (a_strides & 0xffff) + 16 * k;
    } else {
        const int w = 4 * kq + (x - 4);
        if (w >= a_B * npfg) return;
        const int g = w / npfg;
        k = w - g * npfg;
        et = ET_PF; cidx = a_B + g; e0 = a_pf_base + g * (int)((unsigned)a_strides >> 16) + 16 * k;
    }
    // one level: the region's count and the 16 slots of the group (inside the region's 32-aligned capacity whatever the count;
    // slots beyond the count hold stale ids, which are never dereferenced: the rows beyond nv shadow row nv - 1)
    const int cnt = a_dyn_cnt[cidx];
    if (f.rec) {                                         // (kernel-uniform) the slots' edge records instead of their end points: one level less
        const int4* rp = f.rec + (size_t)3 * (e0 + j);
        int4 r[3] = {rp[0], rp[1], rp[2]};
        const int nv = __builtin_amdgcn_readfirstlane(min(16, cnt - 16 * k));
        if (nv <= 0) return;                             // workgroup-uniform
        N16_STAMP(sk, lane, wq);                          // item known
        const int jj = min(j, nv - 1);
        const int from = 4 * ((lane & 48) | jj);         // (rows beyond nv shadow row nv - 1: its record, through the LDS crossbar)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            r[u].x = __builtin_amdgcn_ds_bpermute(from, r[u].x); r[u].y = __builtin_amdgcn_ds_bpermute(from, r[u].y);
            r[u].z = __builtin_amdgcn_ds_bpermute(from, r[u].z); r[u].w = __builtin_amdgcn_ds_bpermute(from, r[u].w);
        }
        n16_fused_edge_item(p, f, ep, &lds, e0 + jj, r[0].w & 0xffffff, r[1].w, nv, et, XS, VB, lane, wq, sk, r);
        return;
    }
    const int src_raw = a_esrc[e0 + j], dst_raw = a_edst[e0 + j];
    const int nv = __builtin_amdgcn_readfirstlane(min(16, cnt - 16 * k));
    if (nv <= 0) return;                                 // workgroup-uniform
    N16_STAMP(sk, lane, wq);                              // item known
    const int jj = min(j, nv - 1);
    const int from = 4 * ((lane & 48) | jj);
    const int src = __builtin_amdgcn_ds_bpermute(from, src_raw), dst = __builtin_amdgcn_ds_bpermute(from, dst_raw);
    n16_fused_edge_item(p, f, ep, &lds, e0 + jj, src, dst, nv, et, XS, VB, lane, wq, sk);
}

// ---------------------------------------------------------------------------------------------
// The tail launch (TailParams; pf_denoise_step): workgroup g = graph g.  Its centers, 16 per pass, go through the last conv
// layer's node update (gvp.py:488-536: partial-row sums, residual on the layer's input state, GVPLayerNorm, update chain,
// residual, GVPLayerNorm) and the noise head (dynamics_gvp.py:37-42); the head's last GVP (64 scalars, 1 vector, identity
// gate) runs as a zero-padded GEN block whose unused gate rows 1 .. pharm_nf carry to_scalar_output, so eps_x is the gated
// channel 0 and eps_h the gate pre-activations of channels 1 .. pharm_nf.  eps stays in LDS, and the same workgroup
// finishes with the sampler update + edge build of ITS graph (pf_stepbuild.h).
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) TailLds {
    N16Lds n;
    float ex[PF_MAXF * 4];
    float eh[PF_MAXF * 16];
    pfsb::StepBuildLds sb;
};

// the descriptors of the two in-edge segments of a center in the last conv layer (ff: slot 0, pf: slot 1)
__device__ __forceinline__ void n16_node_desc_last(const TailParams& t, const int node, NodeDesc& nd) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        nd.st[sl] = t.in_start[sl * t.N + node];
        nd.cn[sl] = t.in_cnt[sl * t.N + node];
        nd.gm[sl] = t.grp - 1;
    }
}
// hook(0): called once the gathers have been requested, hook(1): behind the first LayerNorm -- where the tail kernel issues
// the loads of its update + build
template <class Hook>
__device__ __forceinline__ void n16_node_update_last(const TailParams& t, const NodeDesc& nd, N16Ring& ring, const int node, float (&XS)[32],
                                                     float (&VB)[4], N16Lds* lds, const int lane, const int wq, int& sk, Hook&& hook) {
    const int g = lane >> 4;
    const int cw = wq < 3 ? wq : 0;
    // one batch of loads: the partial rows and the residual rows -- the layer's input state of the center --, this wave's quarter
    RowQ rq;
    n16_rows_load(t.msg_s, t.msg_v, t.zero_row, nd, rq, lane, wq);
    f32x4 Hq[2];
    float Vr0[4];
    {
        pf_gcf hp = (pf_gcf)t.h_in + (size_t)node * PF_S + 32 * wq + 4 * g;
        Hq[0] = *reinterpret_cast<const f32x4 PF_AS1*>(hp);
        Hq[1] = *reinterpret_cast<const f32x4 PF_AS1*>(hp + 16);
        pf_gcf vp = (pf_gcf)t.v_in + (size_t)node * 48 + 12 * g + cw;
#pragma unroll
        for (int r = 0; r < 4; ++r) Vr0[r] = vp[3 * r];
    }
    float inv_norm = 1.0f;
    if (t.norm_mode == 1) inv_norm = 1.0f / t.norm_value;
    else if (t.norm_mode == 2) inv_norm = 1.0f / t.gnorm[1 * t.B + t.gid[node]];
    hook(0);
    f32x4 Q[2];
    n16_rows_sum(t.msg_s, t.msg_v, t.zero_row, t.norm_mode, nd, rq, Q, VB, lds, lane, wq);
    N16_STAMP(sk, lane, wq);                              // partial rows summed
    N16_CUT_AT(TAIL_CUT, 1, Q[0][0] + VB[0] + Hq[0][0] + Vr0[0], t.eps_h);
#pragma unroll
    for (int r = 0; r < 4; ++r) { Q[0][r] = fmaf(Q[0][r], inv_norm, Hq[0][r]); Q[1][r] = fmaf(Q[1][r], inv_norm, Hq[1][r]); }
    n16_quarters_to_rows(Q, XS, lds, lane, wq);          // (the first LayerNorm's barrier closes the reads)
#pragma unroll
    for (int r = 0; r < 4; ++r) VB[r] = wq < 3 ? fmaf(VB[r], inv_norm, Vr0[r]) : 0.f;
    n16_layernorm(t.ln1_w, t.ln1_b, XS, VB, lds, lane, wq);
    hook(1);
    float Xr[32], Vr[4];
#pragma unroll
    for (int k = 0; k < 32; ++k) Xr[k] = XS[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) Vr[r] = VB[r];
    N16In none{};
    f32x4 S[2];
    S[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; S[1] = S[0];
    N16P XP;
    n16_gen_run<0>(t.n_upd, ring, XS, XP, VB, none, S, lds, lane, wq, sk);
    n16_gate_flush(VB, lds, lane, wq);
#pragma unroll
    for (int k = 0; k < 32; ++k) XS[k] += Xr[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) VB[r] += Vr[r];
    n16_layernorm(t.ln2_w, t.ln2_b, XS, VB, lds, lane, wq);
}

// (the leading scalar arguments: what round trip (A) of the update + build and the first loads of the node update need,
// preloaded into scalar registers with the wave)
__global__ __launch_bounds__(256) void k_n16_tail(const int* __restrict__ a_prot_ptr, const int* __restrict__ a_pharm_ptr,
                                                  const int* __restrict__ a_reg, const int a_B, const int a_Np_tot,
                                                  const TailParams t, const StepParams sp, const BuildParams bp) {
    __shared__ TailLds L;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = lane >> 4, j = lane & 15;
    const int gq = (int)blockIdx.x;
    int sk = 3 << 8;
    N16_STAMP(sk, lane, wq);                              // kernel entry
    // what the update + build reads besides eps: trips (A) and (B) leave now, (C) behind the node update's descriptors -- all
    // of it has landed long before the head is done
    pfsb::SbPre<256> pre;
    pfsb::sb_load_a<256>(pre, gq, a_prot_ptr, a_pharm_ptr, a_reg, a_B, a_Np_tot, bp);
    const int f0 = pre.f0, Nf = pre.Nf;
    bool c_loaded = false;
    if (Nf == 0) pfsb::sb_load_b<256>(pre, a_Np_tot, sp, bp);       // (a graph without centers: nothing to put the loads under)
    for (int base = 0; base < Nf; base += 16) {           // workgroup-uniform
        const int nv = __builtin_amdgcn_readfirstlane(min(16, Nf - base));
        const int fl = base + min(j, nv - 1);
        const int node = a_Np_tot + f0 + fl;
        N16Ring ring;
        ring_start(ring, t.chain + (size_t)wq * t.chain_stride, lane);
        float XS[32], VB[4];
        NodeDesc nd;
        n16_node_desc_last(t, node, nd);
        // the node update's own gathers leave first (the memory counter is in-order: loads issued in front of them would delay
        // them); trip (B) of the update + build follows underneath, trip (C) behind the first LayerNorm
        n16_node_update_last(t, nd, ring, node, XS, VB, &L.n, lane, wq, sk, [&](const int phase) {
            if (c_loaded) return;
            if (phase == 0) pfsb::sb_load_b<256>(pre, a_Np_tot, sp, bp);
            else { pfsb::sb_load_c<256>(pre, bp); c_loaded = true; }
        });
        N16_CUT_AT(TAIL_CUT, 2, XS[0] + VB[0], t.eps_h);
        N16In none{};
        f32x4 S[2], GS = {0.f, 0.f, 0.f, 0.f};
        S[0] = GS; S[1] = GS;
        N16P XP;
        n16_gen_run<0, false>(t.n_head - 1, ring, XS, XP, VB, none, S, &L.n, lane, wq, sk);
        if (t.n_head > 1) n16_block<N16_GEN, 0, true, false, true>(ring, XS, XP, VB, none, S, &L.n, lane, wq, sk, &GS);
        else { n16_split_rows(XS, XP); n16_block<N16_GEN, 0, true, false, false>(ring, XS, XP, VB, none, S, &L.n, lane, wq, sk, &GS); }
        N16_STAMP(sk, lane, wq);                          // head done
        if (j < nv) {
            if (wq < 3 && g == 0) {                       // gated channel 0 of coordinate wq
                L.ex[fl * 4 + wq] = VB[0];
                t.eps_x[(size_t)(f0 + fl) * 3 + wq] = VB[0];
            }
            if (wq == 0) {                                // gate rows 1 .. pharm_nf = to_scalar_output
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = 4 * g + r;
                    if (ch >= 1 && ch <= t.pharm_nf) {
                        L.eh[fl * 16 + ch - 1] = GS[r];
                        t.eps_h[(size_t)(f0 + fl) * t.pharm_nf + ch - 1] = GS[r];
                    }
                }
            }
        }
        lds_barrier();                                    // eps of this pass is in LDS; the chain's buffers are free again
    }
    if (!c_loaded) pfsb::sb_load_c<256>(pre, bp);         // a graph without centers
#if defined(TAIL_CUT) && TAIL_CUT == 3                    // diagnostic builds: the launch without its update + build (timing only)
    return;
#endif
    const pfsb::EpsLds eps{L.ex, L.eh};
    float ex[3] = {0.f, 0.f, 0.f}, eh[pfsb::SB_MAXNF];
    const int fme = min((int)threadIdx.x, PF_MAXF - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) ex[c] = eps.x(fme, c);
#pragma unroll
    for (int k = 0; k < pfsb::SB_MAXNF; ++k) eh[k] = eps.h(fme, k);
    pfsb::sb_finish<256>(pre, ex, eh, gq, sp, bp, L.sb);
}

// ---------------------------------------------------------------------------------------------
// pf_debug_chain kinds 16 / 17: the message / update chain in the n16 form on caller-supplied rows (layouts of kinds 0 / 1)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_n16_unit(const UnitParams p) {
    __shared__ N16Lds lds;
    const int lane = threadIdx.x & 63;
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int e0 = 16 * (int)blockIdx.x;
    const int nv = __builtin_amdgcn_readfirstlane(min(16, p.n - e0));
    if (nv <= 0) return;
    const int g = lane >> 4, j = lane & 15;
    const int row = e0 + min(j, nv - 1);
    const bool msg = p.kind == 16;
    int sk = 4 << 8;
    N16_STAMP(sk, lane, wq);                              // item start
    const int sw = msg ? 144 : 128, vw = msg ? 51 : 48, v0 = msg ? 3 : 0;
    N16Ring ring;
    ring_start(ring, (pf_gcf)p.stream + (size_t)wq * p.n16_stride, lane);
    float XS[32], VB[4];
    f32x4 S[2];
    S[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; S[1] = S[0];
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) XS[4 * T + r] = p.s_in[(size_t)row * sw + 16 * T + 4 * g + r];
#pragma unroll
    for (int r = 0; r < 4; ++r) VB[r] = wq < 3 ? p.v_in[(size_t)row * vw + v0 + (4 * g + r) * 3 + wq] : 0.f;
    N16In in;
#pragma unroll
    for (int r = 0; r < 4; ++r) in.rb[r] = msg ? p.s_in[(size_t)row * sw + 128 + 4 * g + r] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) in.xh[c] = msg ? p.v_in[(size_t)row * vw + c] : 0.f;
    N16P XP;
    if (msg) {
        constexpr int OFF1 = n16_sched(N16_M0F).nq % N16_D;
        n16_split_rows(XS, XP);
        n16_block<N16_M0F, 0, false, true, false, false>(ring, XS, XP, VB, in, S, &lds, lane, wq, sk);
        for (int gi = 1; gi + 1 < p.n_gvps; ++gi) n16_block<N16_GEN, OFF1, false, true, true, false>(ring, XS, XP, VB, in, S, &lds, lane, wq, sk);
        n16_block<N16_GEN, OFF1, true, true, true>(ring, XS, XP, VB, in, S, &lds, lane, wq, sk);
    } else {
        n16_gen_run<0, false>(p.n_gvps - 1, ring, XS, XP, VB, in, S, &lds, lane, wq, sk);
        if (p.n_gvps > 1) n16_block<N16_GEN, 0, true, true, true>(ring, XS, XP, VB, in, S, &lds, lane, wq, sk);
        else { n16_split_rows(XS, XP); n16_block<N16_GEN, 0, true, true, false>(ring, XS, XP, VB, in, S, &lds, lane, wq, sk); }
    }
    N16_STAMP(sk, lane, wq);                              // chain done
    if (j < nv) {
        float* so = p.s_out + (size_t)row * 128 + 32 * wq + 4 * g;
        *reinterpret_cast<f32x4*>(so) = S[0];
        *reinterpret_cast<f32x4*>(so + 16) = S[1];
        if (wq < 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p.v_out[(size_t)row * 48 + (4 * g + r) * 3 + wq] = VB[r];
        }
    }
}

}  // namespace

extern "C" {
// the grid is the launch's capacity in 16-slot groups (p->ngroups_sel) or two per tile; layer0: the type tables of the static
// hoist (p->ptab, p->ptab16_off) and the encoder of the centers (enc) are required
void pfk_n16_edge(const EdgeParams* p, const EncodeParams* enc, int layer0, hipStream_t s) {
    if (p->ntiles == 0) return;
    const int grid = p->nreg > 0 ? p->ngroups_sel : p->ntiles * 2;
    if (grid <= 0) return;
    const EncodeParams noenc{};
    if (layer0 && p->uni_s2g != 0) {                     // ff / pf / fp regions at fixed strides: arithmetic tiling (k_n16_edge_u)
        const int B = p->regB;
        const int nstat = B * (((p->uni_s2g >> 16) & 7) + ((p->uni_s2g >> 19) & 7) + ((p->uni_s2g >> 22) & 7));
        hipLaunchKernelGGL(k_n16_edge_u, dim3(nstat + p->uni_pa_groups), dim3(256), 0, s, p->dyn_cnt, p->esrc, p->edst, p->uni_base[0],
                           p->uni_base[1], p->uni_base[2], p->uni_s01, p->uni_s2g, p->reg, *p, *enc);
        return;
    }
    if (layer0) hipLaunchKernelGGL((k_n16_edge<true>), dim3(grid), dim3(256), 0, s, p->dyn_cnt, p->reg, p->nreg, p->regB, p->pa_abs, *p, *enc);
    else hipLaunchKernelGGL((k_n16_edge<false>), dim3(grid), dim3(256), 0, s, p->dyn_cnt, p->reg, p->nreg, p->regB, p->pa_abs, *p, noenc);
}
#ifdef N16_STAMPS
int pfk_n16_set_stamp_buffer(unsigned long long* dev, int off) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_n16_stamp_off), &off, sizeof(off));
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_n16_stamps), &dev, sizeof(dev));
}
int pfk_n16_set_stamp_kernel(int kid) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_n16_stamp_kid), &kid, sizeof(kid)); }
#endif
#ifdef N16_TRACE
int pfk_n16_set_trace_buffer(unsigned long long* dev) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_n16_trace), &dev, sizeof(dev)); }
#endif
// grid: the edge launch's capacity in 16-slot groups (f->n_edge_items) + PF_MAXF / 16 store items per graph
void pfk_n16_fused(const EdgeParams* p, const FusedParams* f, const EncodeParams* enc, hipStream_t s) {
    if (f->uni_groups != 0) {                            // regions at a fixed stride: arithmetic tiling (k_n16_fused_u)
        const int nffg = f->uni_groups & 255, npfg = (f->uni_groups >> 8) & 255;
        const int grid_u = 8 * std::max((f->B * nffg + f->B + 3) / 4, (f->B * npfg + 3) / 4);
        if (grid_u > 0)
            hipLaunchKernelGGL(k_n16_fused_u, dim3(grid_u), dim3(256), 0, s, p->dyn_cnt, p->esrc, p->edst, f->uni_ff_base, f->uni_pf_base,
                               f->uni_strides, f->uni_groups | (f->B << 16), *p, *f, *enc);
        return;
    }
    int grid = f->n_edge_items + f->B * (PF_MAXF / 16);
    if (f->xcd_split) grid = 8 * std::max((f->nff_cap + f->B + 3) / 4, (f->npf_cap + 3) / 4);
    if (grid <= 0) return;
    hipLaunchKernelGGL(k_n16_fused, dim3(grid), dim3(256), 0, s, p->dyn_cnt, p->reg, p->nreg, p->regB, *p, *f, *enc);
}
// grid: one workgroup per graph
void pfk_n16_tail(const TailParams* t, const StepParams* sp, const BuildParams* bp, hipStream_t s) {
    if (sp->B <= 0) return;
    hipLaunchKernelGGL(k_n16_tail, dim3(sp->B), dim3(256), 0, s, bp->prot_ptr, bp->pharm_ptr, bp->reg, bp->B, bp->Np_tot, *t, *sp, *bp);
}
void pfk_n16_unit(const UnitParams* p, hipStream_t s) {
    if (p->n <= 0) return;
    hipLaunchKernelGGL(k_n16_unit, dim3((p->n + 15) / 16), dim3(256), 0, s, *p);
}
}
