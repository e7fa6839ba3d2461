// pf_stepbuild.h -- the p(z_s | z_t) update of one graph and, on the new coordinates, the dynamic edges of the NEXT dynamics
// call, as one workgroup-wide device function (gfx950 only).  Shared by k_step_build_fast (pf_kernels.hip: its own launch,
// 512 threads, eps read from global memory) and k_n16_tail (pf_n16.hip: the tail of the node + head launch, 256 threads, eps
// handed over in LDS) -- the same code for both, so the two forms emit identical edge lists by construction.
//
// Reference: sample_p_zs_given_zt pharmacodiff.py:397-429 (mu, sigma * noise, COM removal of center AND protein coordinates),
// add_pharm_edges dynamics_gvp.py:187-227 (ff radius / kNN, pf kNN, fp = pf reversed), torch_cluster semantics as fixed in
// oracle/pf_oracle.py (strict d^2 < r^2; kNN ordered by (d^2, index)).  d^2 is (dx*dx + dy*dy) + dz*dz, one rounding per
// operation.
//
// Organised around what a launch pays for here: every launch starts with cold caches, a dependent global round trip costs
// ~2,000 cycles, and __syncthreads drains every outstanding load.  Three dependent trips: (A) the graph's pointers and
// regions, (B) every input row -- pharm state, eps, noise, protein coordinates, the static in-edge descriptors -- (C) the
// static pp sources of the thread's atoms.  The updated coordinates stay in LDS for the neighbour searches (one center per
// wave), nothing is read back from global memory.  Shape limits: kNN pf edges, pockets of at most 512 atoms (SB_MAXA / NT
// atoms per thread), at most PF_MAXF centers.
#pragma once
#include <hip/hip_runtime.h>
#include "pf_device.h"

#ifndef SB_STAMP
#define SB_STAMP(k)
#endif

namespace pfsb {

constexpr int SB_MAXA = 512;                    // atoms per pocket this path handles

__device__ __forceinline__ float sqdist_rn(const float4 a, const float4 b) {
    const float dx = __fsub_rn(a.x, b.x), dy = __fsub_rn(a.y, b.y), dz = __fsub_rn(a.z, b.z);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
__device__ __forceinline__ unsigned long long dkey(const float d2, const int idx) {
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned int)idx;
}
// wave-wide minimum of 64-bit keys on the DPP network (no LDS round trips): inclusive min-scan inside each row of 16
// lanes (row_shr 1, 2, 4, 8), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15) and lane 31 into the upper
// half (row_bcast31); lane 63 holds the result.  Lanes without a source keep `old` = all ones, the identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_min_u64(const unsigned long long k) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)(unsigned int)k, CTRL, ROW_MASK, 0xf, false);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)(unsigned int)(k >> 32), CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o < k ? o : k;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k) {
    k = dpp_min_u64<0x111, 0xf>(k);
    k = dpp_min_u64<0x112, 0xf>(k);
    k = dpp_min_u64<0x114, 0xf>(k);
    k = dpp_min_u64<0x118, 0xf>(k);
    k = dpp_min_u64<0x142, 0xa>(k);
    k = dpp_min_u64<0x143, 0xc>(k);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)k, 63);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(k >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
// inclusive wave scan (sum) of a 32-bit value on the DPP network, same pattern
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int dpp_add_u32(const unsigned int v) {
    return v + (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned int wave_incl_scan_u32(unsigned int v) {
    v = dpp_add_u32<0x111, 0xf>(v);
    v = dpp_add_u32<0x112, 0xf>(v);
    v = dpp_add_u32<0x114, 0xf>(v);
    v = dpp_add_u32<0x118, 0xf>(v);
    v = dpp_add_u32<0x142, 0xa>(v);
    v = dpp_add_u32<0x143, 0xc>(v);
    return v;
}
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits for every global store
// issued so far to be acknowledged (~2,000 cycles here), and nothing in this function reads its global stores back
__device__ __forceinline__ void sb_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS of one graph's update + build
struct __attribute__((aligned(16))) StepBuildLds {
    float4 fx[PF_MAXF];                         // updated pharm coordinates (COM removed)
    float4 px[SB_MAXA];                         // updated protein coordinates
    float red[8][3];
    unsigned long long scratch[8];
    unsigned int refm[SB_MAXA][2];              // per atom: bit fl set <=> center fl has the atom among its k neighbours
    // active atoms in list order: first slot of their pp in-edges in the "pa" region, node id, static in-edge start;
    // a_src: the first 16 static sources of EVERY atom (prefetched, indexed by atom) -- the copy into the region is
    // then done by ALL threads, one output slot each, with coalesced stores
    int a_d0[SB_MAXA], a_node[SB_MAXA], a_pst[SB_MAXA];
    __attribute__((aligned(16))) int a_src[SB_MAXA][16];
};

// where the noise prediction of the graph's centers comes from (fl = center index inside the graph)
struct EpsGlobal {
    const float* eps_x; const float* eps_h; int f0, nf;
    __device__ __forceinline__ float x(const int fl, const int c) const { return eps_x[(size_t)(f0 + fl) * 3 + c]; }
    __device__ __forceinline__ float h(const int fl, const int k) const { return eps_h[(size_t)(f0 + fl) * nf + k]; }
};
struct EpsLds {                                  // [PF_MAXF][4] / [PF_MAXF][16] in LDS (written by the head of the same workgroup)
    const float* ex; const float* eh;
    __device__ __forceinline__ float x(const int fl, const int c) const { return ex[fl * 4 + c]; }
    __device__ __forceinline__ float h(const int fl, const int k) const { return eh[fl * 16 + k]; }
};

// NT threads (a multiple of 64, >= 256; SB_MAXA / NT atoms per thread), every thread of the workgroup calls it.  The caller
// has made the eps source readable by all threads (a barrier behind the LDS writes of EpsLds).
template <int NT, class Eps>
__device__ __forceinline__ void step_build_fast_body(const int g, const int* __restrict__ a_prot_ptr, const int* __restrict__ a_pharm_ptr,
                                                     const int* __restrict__ a_reg, const int a_B, const int a_Np_tot,
                                                     const StepParams& sp, const BuildParams& p, const Eps& eps, StepBuildLds& L) {
    constexpr int NW = NT / 64, APT = SB_MAXA / NT;
    static_assert(NT % 64 == 0 && NW >= 4 && NW <= 8 && APT * NT == SB_MAXA, "256 or 512 threads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SB_STAMP(0);
    // ---- (A) pointers and regions
    const int p0 = a_prot_ptr[g], p1 = a_prot_ptr[g + 1];
    const int f0 = a_pharm_ptr[g], f1 = a_pharm_ptr[g + 1];
    const int Np = p1 - p0, Nf = f1 - f0;
    const int GF = a_Np_tot + f0;
    const int reg_ff = a_reg[0 * a_B + g], reg_pf = a_reg[1 * a_B + g], reg_fp = a_reg[2 * a_B + g], reg_pa = a_reg[3 * a_B + g];
    const int reg_act = p.act_ids ? p.reg_act[g] : 0;
    int* in_start0 = p.in_start;          int* in_cnt0 = p.in_cnt;
    int* in_start1 = p.in_start + p.N;    int* in_cnt1 = p.in_cnt + p.N;
    int* in_start2 = p.in_start + 2 * p.N; int* in_cnt2 = p.in_cnt + 2 * p.N;
    // ---- (B) every input row of this thread
    const bool isf = tid < Nf;
    bool isp[APT];
    float4 xf = make_float4(0.f, 0.f, 0.f, 0.f), xp[APT];
    float ex[3] = {0.f, 0.f, 0.f}, nzx[3] = {0.f, 0.f, 0.f};
    if (isf) {
        xf = sp.xn[GF + tid];
#pragma unroll
        for (int c = 0; c < 3; ++c) { ex[c] = eps.x(tid, c); nzx[c] = sp.noise[(size_t)(f0 + tid) * (3 + sp.nf) + c]; }
    }
    int pst[APT], pdeg[APT];
#pragma unroll
    for (int a = 0; a < APT; ++a) {
        const int c = APT * tid + a;
        isp[a] = c < Np;
        xp[a] = make_float4(0.f, 0.f, 0.f, 0.f);
        pst[a] = 0; pdeg[a] = 0;
        if (isp[a]) {
            xp[a] = sp.xn[p0 + c];
            if (p.act_ids && !p.pa_static) { pst[a] = in_start1[p0 + c]; pdeg[a] = in_cnt1[p0 + c]; }
        }
    }
    // feature update of the pharm nodes (independent of everything else): load, update, store
    if (isf) {
        for (int k = 0; k < sp.nf; ++k) {
            const size_t o = (size_t)(f0 + tid) * sp.nf + k;
            const float hv = sp.pharm_h[o], e = eps.h(tid, k);
            const float mu = sp.ep_feat ? (sp.ep_zt * hv + sp.ep_pred * e) : (hv / sp.a_ts - sp.var * e);
            sp.pharm_h[o] = mu + sp.sigma * sp.noise[(size_t)(f0 + tid) * (3 + sp.nf) + 3 + k];
        }
    }
    // ---- (C) static pp sources of this thread's atoms (used only if an atom turns out to be active)
    int psrc[APT][16];
#pragma unroll
    for (int a = 0; a < APT; ++a)
#pragma unroll
        for (int k = 0; k < 16; ++k) psrc[a][k] = (isp[a] && p.act_ids && !p.pa_static) ? p.esrc[pst[a] + min(k, max(pdeg[a] - 1, 0))] : 0;
    // ---- coordinate update (pharmacodiff.py:397-426) and COM removal of pharm AND prot coordinates (:429)
    float m[3] = {0.f, 0.f, 0.f};
    if (isf) {
        const float xi[3] = {xf.x, xf.y, xf.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float mu = sp.ep_coord ? (sp.ep_zt * xi[c] + sp.ep_pred * ex[c]) : (xi[c] / sp.a_ts - sp.var * ex[c]);
            m[c] = mu + sp.sigma * nzx[c];
        }
    }
    {   // per-graph mean in the summation order of step_update_body (thread-strided partial sums, xor butterfly, the first four waves)
        float sx = m[0], sy = m[1], sz = m[2];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
        if (lane == 0) { L.red[wave][0] = sx; L.red[wave][1] = sy; L.red[wave][2] = sz; }
    }
    sb_lds_barrier();
    SB_STAMP(8);
    float com[3];
    {
        const float n = (float)max(Nf, 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) com[c] = Nf > 0 ? (((L.red[0][c] + L.red[1][c]) + (L.red[2][c] + L.red[3][c])) / n) : 0.f;
    }
    if (isf) {
        const float4 v = make_float4(m[0] - com[0], m[1] - com[1], m[2] - com[2], 0.f);
        L.fx[tid] = v;
        sp.xn[GF + tid] = v;
    }
#pragma unroll
    for (int a = 0; a < APT; ++a) {
        const int c = APT * tid + a;
        if (isp[a]) { xp[a].x -= com[0]; xp[a].y -= com[1]; xp[a].z -= com[2]; sp.xn[p0 + c] = xp[a]; }
        L.px[c] = xp[a];
        L.refm[c][0] = 0u; L.refm[c][1] = 0u;
    }
    sb_lds_barrier();
    // ---- ff (pharm -> pharm) on the last wave: counts, wave scan for the offsets, emission
    const int kff = p.ff_k > 0 ? max(min(p.ff_k, Nf - 1), 0) : 0;
    int ff_total = 0;                                 // valid in the last wave
    if (wave == NW - 1) {
        int c = 0;
        if (lane < Nf) {
            if (p.ff_k > 0) c = kff;
            else
                for (int jn = 0; jn < Nf; ++jn)
                    if (jn != lane && sqdist_rn(L.fx[jn], L.fx[lane]) < p.r2_ff) ++c;
        }
        const int incl = (int)wave_incl_scan_u32((unsigned int)c);
        ff_total = __builtin_amdgcn_readlane(incl, 63);
        if (lane == 0) p.dyn_cnt[0 * p.B + g] = ff_total;
        if (lane < Nf) {
            int e = reg_ff + incl - c;
            in_start0[GF + lane] = e;
            in_cnt0[GF + lane] = c;
            if (p.ff_k > 0) {
                unsigned long long prev = 0ull;
                bool first = true;
                for (int q = 0; q < kff; ++q) {
                    unsigned long long best = ~0ull;
                    for (int jn = 0; jn < Nf; ++jn) {
                        if (jn == lane) continue;
                        const unsigned long long k = dkey(sqdist_rn(L.fx[jn], L.fx[lane]), jn);
                        if ((first || k > prev) && k < best) best = k;
                    }
                    prev = best; first = false;
                    p.esrc[e] = GF + (int)(best & 0xffffffffu);
                    p.edst[e] = GF + lane;
                    ++e;
                }
            } else {
                for (int jn = 0; jn < Nf; ++jn)
                    if (jn != lane && sqdist_rn(L.fx[jn], L.fx[lane]) < p.r2_ff) { p.esrc[e] = GF + jn; p.edst[e] = GF + lane; ++e; }
            }
        }
    }
    SB_STAMP(9);
    // ---- pf (prot -> pharm): kNN, one center per wave, candidates (d^2, index) from LDS
    const int kk = min(p.pf_k, Np);
    for (int fl = wave; fl < Nf; fl += NW) {
        const float4 q = L.fx[fl];
        unsigned long long kc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            kc[i] = ~0ull;
            if (64 * i < Np) kc[i] = c < Np ? dkey(sqdist_rn(L.px[c], q), c) : ~0ull;    // wave-uniform bound
        }
        unsigned long long prev = 0ull;
        for (int r = 0; r < kk; ++r) {
            unsigned long long best = ~0ull;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (64 * i < Np && (r == 0 || kc[i] > prev) && kc[i] < best) best = kc[i];
            best = wave_min_u64(best);
            prev = best;
            if (lane == 0) {
                const int pc = (int)(best & 0xffffffffu);
                atomicOr(&L.refm[pc][fl >> 5], 1u << (fl & 31));
                p.esrc[reg_pf + fl * kk + r] = p0 + pc;
                p.edst[reg_pf + fl * kk + r] = GF + fl;
            }
        }
        if (lane == 0) { in_start1[GF + fl] = reg_pf + fl * kk; in_cnt1[GF + fl] = kk; }
    }
    if (tid == 0) { p.dyn_cnt[1 * p.B + g] = Nf * kk; p.dyn_cnt[2 * p.B + g] = Nf * kk; }
    {   // the prefetched pp sources go to LDS here, unconditionally: left to their only use (active atoms, below) the
        // compiler sinks the loads into that branch and the round trip (C) is paid there, late and exposed
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            int4* st = reinterpret_cast<int4*>(&L.a_src[APT * tid + a][0]);
            st[0] = make_int4(psrc[a][0], psrc[a][1], psrc[a][2], psrc[a][3]);
            st[1] = make_int4(psrc[a][4], psrc[a][5], psrc[a][6], psrc[a][7]);
            st[2] = make_int4(psrc[a][8], psrc[a][9], psrc[a][10], psrc[a][11]);
            st[3] = make_int4(psrc[a][12], psrc[a][13], psrc[a][14], psrc[a][15]);
        }
    }
    sb_lds_barrier();
    SB_STAMP(10);
    // ---- fp = pf reversed, destination-major over the atoms; active atoms and the compact copy of their pp in-edges
    {
        // the centers that reference an atom, ascending: the bits of its mask (set by the kNN waves above)
        unsigned int m0[APT], m1[APT];
        int my[APT], act[APT], deg[APT];
        unsigned long long val[APT], tval = 0ull;
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            const int c = APT * tid + a;
            m0[a] = isp[a] ? L.refm[c][0] : 0u; m1[a] = isp[a] ? L.refm[c][1] : 0u;
            my[a] = __popc(m0[a]) + __popc(m1[a]);
            act[a] = (my[a] > 0 && p.act_ids) ? 1 : 0;
            deg[a] = act[a] ? pdeg[a] : 0;
            val[a] = (unsigned long long)my[a] | ((unsigned long long)act[a] << 16) | ((unsigned long long)deg[a] << 28);
            tval += val[a];
        }
        SB_STAMP(12);                                 // references counted
        // block scan over the waves: the value packs three counters (bits 0-15 fp edges, 16-27 active atoms, 28-63 their pp
        // in-edges), scanned as two 32-bit halves (the low two cannot carry into each other at these sizes)
        const unsigned int lo = (unsigned int)(tval & 0xfffffffull), hi = (unsigned int)(tval >> 28);
        const unsigned int slo = wave_incl_scan_u32(lo), shi = wave_incl_scan_u32(hi);
        if (lane == 63) L.scratch[wave] = (unsigned long long)slo | ((unsigned long long)shi << 28);
        sb_lds_barrier();
        unsigned long long before = 0ull, all = 0ull;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned long long t = L.scratch[w];
            if (w < wave) before += t;
            all += t;
        }
        unsigned long long o = before + ((unsigned long long)slo | ((unsigned long long)shi << 28)) - tval;
        SB_STAMP(13);                                 // offsets known
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            const int c = APT * tid + a;
            if (isp[a]) {
                int e = reg_fp + (int)(o & 0xffffu);
                in_start0[p0 + c] = e;
                in_cnt0[p0 + c] = my[a];
                {
                    unsigned int mm = m0[a];
                    while (mm) { const int fl = __ffs(mm) - 1; mm &= mm - 1; p.esrc[e] = GF + fl; p.edst[e] = p0 + c; ++e; }
                    mm = m1[a];
                    while (mm) { const int fl = 32 + __ffs(mm) - 1; mm &= mm - 1; p.esrc[e] = GF + fl; p.edst[e] = p0 + c; ++e; }
                }
                if (act[a]) {
                    const int j = (int)((o >> 16) & 0xfffu);
                    p.act_ids[reg_act + j] = p0 + c;
                    if (p.pa_static) p.need[p.rep_base[g] + c] = p.need_stamp;
                    else {
                        in_start2[p0 + c] = reg_pa + (int)(o >> 28);
                        in_cnt2[p0 + c] = deg[a];
                        L.a_d0[j] = (int)(o >> 28); L.a_node[j] = p0 + c; L.a_pst[j] = pst[a];
                    }
                }
            }
            o += val[a];
        }
        SB_STAMP(14);                                 // fp edges / descriptors stored, active atoms staged
        sb_lds_barrier();
        SB_STAMP(15);
        {   // the "pa" region: slot t belongs to the last active atom whose first slot is <= t (binary search in LDS)
            const int n_pa = p.pa_static ? 0 : (int)(all >> 28), n_act = (int)((all >> 16) & 0xfffu);
            for (int t = tid; t < n_pa; t += NT) {
                int lo2 = 0, hi2 = n_act - 1;
                while (lo2 < hi2) {
                    const int mid = (lo2 + hi2 + 1) >> 1;
                    if (L.a_d0[mid] <= t) lo2 = mid; else hi2 = mid - 1;
                }
                const int k = t - L.a_d0[lo2];
                const int src = k < 16 ? L.a_src[L.a_node[lo2] - p0][k] : p.esrc[L.a_pst[lo2] + k];
                p.esrc[reg_pa + t] = src;
                p.edst[reg_pa + t] = L.a_node[lo2];
                if (p.eorig) p.eorig[reg_pa + t] = L.a_pst[lo2] + k;
            }
        }
        if (tid == 0 && p.act_ids) {
            p.dyn_cnt[3 * p.B + g] = p.pa_static ? p.pa_static[g] : (int)(all >> 28);
            p.dyn_cnt[4 * p.B + g] = (int)((all >> 16) & 0xfffu);
        }
        if (tid == NT - 64 && p.norm_mode == 2) {     // per-graph normalisers for message_norm == 0 (gvp.py:504-507)
            const int cpf = p.pfq_cnt ? p.pfq_cnt[g] : Nf * kk;
            p.gnorm[1 * p.B + g] = (float)(ff_total + cpf) / (float)Nf + 1.0f;
            p.gnorm[0 * p.B + g] = (float)(cpf + p.pp_cnt[g]) / (float)Np + 1.0f;
        }
    }
    SB_STAMP(11);
}

}  // namespace pfsb
