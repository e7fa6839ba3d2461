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
// half wave), nothing is read back from global memory.  Shape limits: kNN pf edges, pockets of at most 512 atoms (SB_MAXA / NT
// atoms per thread), at most PF_MAXF centers.
#pragma once
#include <hip/hip_runtime.h>
#include "pf_device.h"

#ifndef SB_STAMP
#define SB_STAMP(k)
#endif
// diagnostic builds (-DSB_CUT=k): the body returns at phase k (timing only: the step's results are then incomplete)
#ifdef SB_CUT
#define SB_PHASE(k) do { SB_STAMP(k); if ((k) == SB_CUT) __builtin_amdgcn_endpgm(); } while (0)
#else
#define SB_PHASE(k) SB_STAMP(k)
#endif

namespace pfsb {

constexpr int SB_MAXA = 512;                    // atoms per pocket this path handles
constexpr int SB_MAXNF = 16;                    // pharmacophore feature count this path handles (pharm_nf)

// (dx*dx + dy*dy) + dz*dz with one rounding per operation, like oracle/pf_oracle.py:_d2 -- the pragma matters: hipcc's
// __fmul_rn / __fadd_rn are plain * and +, which the device compiler contracts into v_fma_f32 by default (one rounding less:
// a squared distance one ulp off the oracle's can flip a radius test or a kNN tie)
__device__ __forceinline__ float sqdist_rn(const float4 a, const float4 b) {
#pragma clang fp contract(off)
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
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
// kNN of one center per HALF wave (lanes 0..31 / 32..63): NC candidates per lane -- atom c = l32 + 32 i, its squared distance
// as the bit pattern of a non-negative float (monotone as an unsigned integer; 0xffffffff: no atom) -- and k rounds of
//   lane minimum (ascending i: the smaller atom index wins a tie), half-wave minimum of (distance, atom) in lexicographic
//   order on the DPP network (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3:
//   lanes 31 / 63 hold the two halves' minima), removal of the winner from its lane's list.
// The order of the k neighbours is (d^2, index) ascending -- what torch_cluster.knn returns (oracle/pf_oracle.py:knn).  All
// 32-bit operations: the 64-bit keys of wave_min_u64 cost 6.8 us per graph here (branches around every candidate, 64-bit
// compares), this form ~1 us.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_lexmin(unsigned int& bd, int& bc) {
    const unsigned int od = (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)bd, CTRL, ROW_MASK, 0xf, false);
    const int oc = __builtin_amdgcn_update_dpp(0x7fffffff, bc, CTRL, ROW_MASK, 0xf, false);
    const bool take = od < bd || (od == bd && oc < bc);
    bd = take ? od : bd;
    bc = take ? oc : bc;
}
// returns, in lane l32 = r < kk of each half, the r-th nearest atom of that half's center (kk <= PF_MAXK <= 32)
template <int NC>
__device__ __forceinline__ int knn_halfwave(const float4* px, const float4 q, const int Np, const int kk, const int lane) {
    const int hw = lane >> 5, l32 = lane & 31;
    int mine = 0;
    unsigned int d[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = l32 + 32 * i;                   // (px holds SB_MAXA rows: the read is in range whatever Np)
        const float d2 = sqdist_rn(px[c], q);
        d[i] = c < Np ? __float_as_uint(d2) : 0xffffffffu;
    }
    for (int r = 0; r < kk; ++r) {
        unsigned int bd = 0xffffffffu;
        int bc = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const bool take = d[i] < bd;
            bd = take ? d[i] : bd;
            bc = take ? l32 + 32 * i : bc;
        }
        dpp_lexmin<0x111, 0xf>(bd, bc);
        dpp_lexmin<0x112, 0xf>(bd, bc);
        dpp_lexmin<0x114, 0xf>(bd, bc);
        dpp_lexmin<0x118, 0xf>(bd, bc);
        dpp_lexmin<0x142, 0xa>(bd, bc);
        const int w0 = __builtin_amdgcn_readlane(bc, 31), w1 = __builtin_amdgcn_readlane(bc, 63);
        const int wc = hw ? w1 : w0;
#pragma unroll
        for (int i = 0; i < NC; ++i) d[i] = (wc == l32 + 32 * i) ? 0xffffffffu : d[i];
        mine = l32 == r ? wc : mine;
    }
    return mine;
}

// The same result with a third of the instructions (round 5; the search is ~1/3 of what a step waits for behind eps).  Keys are the
// distance bits with the candidate's slot i in their low log2(NC) bits: a lane sorts its NC keys once (bitonic network of v_min /
// v_max), a round is then the half-wave minimum of the lanes' heads (v_min_u32 on the DPP network), two v_readlane, the winner lane
// by s_ff1 of a ballot, and a pop of the winner's list (v_cndmask under a scalar mask).  Truncating the distance is exact unless two
// of the k + 1 smallest keys share a bucket (distances equal in all but their low bits: ~1e-6 of the searches, always for
// duplicated atoms) or a key is the "no atom" pattern: then `exact` comes back false (wave-uniform) and the caller runs
// knn_halfwave.  Selected keys with pairwise different buckets are ordered as their distances are, and everything not selected
// is no smaller than the (k + 1)-th key.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int dpp_umin(const unsigned int v) {
    const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
// lane l: bit l of the scalar mask set ? b : a -- as ONE v_cndmask (written as `win ? key[j + 1] : key[j]` over an array, the compiler
// turns the pop into indexed reads of the array, i.e. a compare chain of NC selects per element)
__device__ __forceinline__ unsigned int sel_by_mask(const unsigned int a, const unsigned int b, const unsigned long long m) {
    unsigned int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}
template <int NC>
__device__ __forceinline__ void knn_sort_keys(unsigned int (&k)[NC]) {
#pragma unroll
    for (int sz = 2; sz <= NC; sz <<= 1)
#pragma unroll
        for (int j = sz >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned int a = k[i], b = k[l];
                    const unsigned int lo = a < b ? a : b, hi = a < b ? b : a;
                    const bool up = (i & sz) == 0;
                    k[i] = up ? lo : hi;
                    k[l] = up ? hi : lo;
                }
            }
}
template <int NC>
__device__ __forceinline__ int knn_halfwave_keys(const float4* px, const float4 q, const int Np, const int kk, const int lane, bool& exact) {
    static_assert(NC == 8 || NC == 16, "slots per lane");
    constexpr unsigned int IM = NC - 1;
    constexpr int LOG = NC == 8 ? 3 : 4;
    const int hw = lane >> 5, l32 = lane & 31;
    unsigned int key[NC];
    float4 cand[NC];                                   // every candidate requested before the first is used (left to itself the
#pragma unroll                                         // compiler reads, waits and computes one at a time: NC LDS round trips)
    for (int i = 0; i < NC; ++i) cand[i] = px[l32 + 32 * i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = l32 + 32 * i;
        const float d2 = sqdist_rn(cand[i], q);
        key[i] = c < Np ? ((__float_as_uint(d2) & ~IM) | (unsigned int)i) : 0xffffffffu;
    }
    knn_sort_keys<NC>(key);
    int mine = 0;
    unsigned int kms = 0xffffffffu;                    // lane r of a half: the r-th smallest key of the half (r <= kk)
    for (int r = 0; r <= kk; ++r) {
        unsigned int m = key[0];
        m = dpp_umin<0x111, 0xf>(m);
        m = dpp_umin<0x112, 0xf>(m);
        m = dpp_umin<0x114, 0xf>(m);
        m = dpp_umin<0x118, 0xf>(m);
        m = dpp_umin<0x142, 0xa>(m);
        const unsigned int k0 = (unsigned int)__builtin_amdgcn_readlane((int)m, 31), k1 = (unsigned int)__builtin_amdgcn_readlane((int)m, 63);
        const unsigned int kmin = hw ? k1 : k0;
        kms = l32 == r ? kmin : kms;
        if (r == kk) break;
        const unsigned long long eq = __builtin_amdgcn_ballot_w64(key[0] == kmin);
        const unsigned int e0 = (unsigned int)eq, e1 = (unsigned int)(eq >> 32);      // (neither is 0: the minimum came from a lane)
        const int wl0 = __builtin_ctz(e0 | 0x80000000u), wl1 = __builtin_ctz(e1 | 0x80000000u);
        const unsigned long long wm = (1ull << wl0) | (1ull << (32 + wl1));
        const int wc0 = wl0 + 32 * (int)(k0 & IM), wc1 = wl1 + 32 * (int)(k1 & IM);
#pragma unroll
        for (int j = 0; j + 1 < NC; ++j) key[j] = sel_by_mask(key[j], key[j + 1], wm);
        key[NC - 1] = sel_by_mask(key[NC - 1], 0xffffffffu, wm);
        mine = l32 == r ? (hw ? wc1 : wc0) : mine;
    }
    // exact unless two consecutive ones of the kk + 1 smallest keys share a bucket or a selected key is the "no atom" pattern: lane r
    // of a half compares its key's bucket with lane r + 1's (row_shl:1 inside the rows of 16, the row boundary 15 | 16 through lane 16's
    // value read by v_readlane -- kk <= 16 < 32, so only that one boundary exists)
    const unsigned int bk = kms >> LOG;
    unsigned int nb = (unsigned int)__builtin_amdgcn_update_dpp(-1, (int)bk, 0x101, 0xf, 0xf, false);     // row_shl:1 = lane + 1's value (lane 15 of a row: -1)
    const unsigned int b16lo = (unsigned int)__builtin_amdgcn_readlane((int)bk, 16), b16hi = (unsigned int)__builtin_amdgcn_readlane((int)bk, 48);
    if (l32 == 15) nb = hw ? b16hi : b16lo;
    const bool bad = (l32 < kk && (bk == nb || bk == (0xffffffffu >> LOG)));
    const bool amb = __builtin_amdgcn_ballot_w64(bad) != 0ull;
    exact = !amb;
    return mine;
}

// x[l] + x[l ^ 32] + ... in the pairing order of the xor butterfly `for (o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o)` (what
// step_update_body sums with: same bits), on the register network instead of six LDS-crossbar round trips: lane-half and
// row swaps (v_permlane32_swap / v_permlane16_swap; asm: the compiler's builtins mis-assign their second result), row_ror:8,
// one ds_swizzle for the xor-4 step, two quad_perm steps
__device__ __forceinline__ float wave_xor_sum(float v) {
    {
        unsigned a = __builtin_bit_cast(unsigned, v), b = a;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
    {
        unsigned a = __builtin_bit_cast(unsigned, v), b = a;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));      // row_ror:8 = lane ^ 8
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x101f));                          // lane ^ 4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));       // quad_perm [2,3,0,1] = lane ^ 2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));       // quad_perm [1,0,3,2] = lane ^ 1
    return v;
}
// three sums at once: the same operations per value in the same order (same bits), the three chains' latencies overlapped
__device__ __forceinline__ void wave_xor_sum3(float (&v)[3]) {
    unsigned a[3], b[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] = __builtin_bit_cast(unsigned, v[c]); b[c] = a[c]; }
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5"
                 : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]));
#pragma unroll
    for (int c = 0; c < 3; ++c) { v[c] = __builtin_bit_cast(float, a[c]) + __builtin_bit_cast(float, b[c]); a[c] = __builtin_bit_cast(unsigned, v[c]); b[c] = a[c]; }
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5"
                 : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]));
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = __builtin_bit_cast(float, a[c]) + __builtin_bit_cast(float, b[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[c]), 0x128, 0xf, 0xf, false));
    float w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) w[c] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v[c]), 0x101f));
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] += w[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[c]), 0x4e, 0xf, 0xf, false));
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[c]), 0xb1, 0xf, 0xf, false));
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
    float com[4];
    unsigned long long scratch[2][8];             // [pass][wave]
    unsigned int refm[SB_MAXA][2];              // per atom: bit fl set <=> center fl has the atom among its k neighbours
    // fp edges (destination-major) are stored one per (center, neighbour) PAIR: the kNN lanes leave their pair here, an atom's
    // thread the first slot of the atom's fp in-edges, and a pair's slot is that plus the referencing centers below its own
    int pairs[PF_MAXF * PF_MAXK];               // [center * k + rank] = atom (local index) | center << 16
    int a_e0[SB_MAXA];                          // per atom: first slot of its fp in-edges
    // the compact copy of the active atoms' pp in-edges ("pa" region) is written by 16 threads per active atom from a_src -- the
    // first 16 static sources of EVERY atom (prefetched, indexed by atom); thread 15 of an atom also copies what lies beyond 16
    __attribute__((aligned(16))) int4 act4[SB_MAXA];   // [j-th active atom] = (atom, first slot in the region, in-degree, static in-edge start)
    int2 a_pa[SB_MAXA];                         // BuildParams::rec: per ACTIVE atom, its "pa" descriptor (first slot absolute, in-degree)
    int2 ffd[PF_MAXF];                          // BuildParams::rec: per center, its ff in-edge descriptor (first slot absolute, count)
    unsigned char a_ty[SB_MAXA];                // BuildParams::rec: per atom, its element type
    int chgmin;                                 // BuildParams::pa_same: the first slot of the "pa" region (relative) at which anything changed
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

// What the update + build of graph g reads that does NOT depend on the noise prediction: trips (A), (B) and (C).  A kernel that
// computes eps itself (k_n16_tail) issues these loads in front of / underneath its chain, level by level with its own gathers.
// atom a of thread tid: pass a covers atoms [NT a, NT a + NT) -- a pocket of at most NT atoms gives every thread ONE atom and pass 1
// nothing (dealt as 2 tid + a, a 256-atom pocket kept half of a 256-thread workgroup idle in every per-atom phase and the other half
// working through two atoms one after the other)
template <int NT>
__device__ __forceinline__ int sb_atom(const int tid, const int a) { return tid + NT * a; }
template <int NT>
struct SbPre {
    static constexpr int APT = SB_MAXA / NT;
    int p0, Np, f0, Nf, GF, reg_ff, reg_pf, reg_fp, reg_pa, reg_act;
    bool isf, isp[APT];
    float4 xf, xp[APT];
    float nzx[3];
    float hv[SB_MAXNF], nzh[SB_MAXNF];              // the center's features and their noise (threads tid < Nf; nf <= SB_MAXNF)
    int pst[APT], pdeg[APT], psrc[APT][16];
    int pty[APT];                                   // BuildParams::rec: the atom's element type
    int ost[APT], ocn[APT], ostamp[APT];            // BuildParams::pa_same: the atom's slot-2 range and stamp as the previous step left them
};
// (A) the graph's pointers and regions (scalar loads)
template <int NT>
__device__ __forceinline__ void sb_load_a(SbPre<NT>& q, const int g, const int* __restrict__ a_prot_ptr, const int* __restrict__ a_pharm_ptr,
                                          const int* __restrict__ a_reg, const int a_B, const int a_Np_tot, const BuildParams& p) {
    q.p0 = a_prot_ptr[g]; q.Np = a_prot_ptr[g + 1] - q.p0;
    q.f0 = a_pharm_ptr[g]; q.Nf = a_pharm_ptr[g + 1] - q.f0;
    q.GF = a_Np_tot + q.f0;
    q.reg_ff = a_reg[0 * a_B + g]; q.reg_pf = a_reg[1 * a_B + g]; q.reg_fp = a_reg[2 * a_B + g]; q.reg_pa = a_reg[3 * a_B + g];
    q.reg_act = p.act_ids ? p.reg_act[g] : 0;
}
// (B) every input row of this thread.  Branch-free: rows beyond the graph's counts are read at clamped indices and discarded, so
// the number of loads in flight is a compile-time constant -- a caller that issues these loads underneath its own (k_n16_tail)
// keeps counted vmcnt waits instead of vmcnt(0).
// (Every load of the body is issued here and in sb_load_c, none behind its first store: gfx9 counts loads and stores in ONE
// in-order counter, so a load that follows stores of a run-time count can only be waited for with vmcnt(0) -- which also
// waits for every store before it, ~2,000 cycles each time.)
// FEAT = false: the caller updates the features itself (step_build_wait_body: on other lanes) -- the center threads' feature rows
// are not loaded.
template <int NT, bool FEAT = true>
__device__ __forceinline__ void sb_load_b(SbPre<NT>& q, const int a_Np_tot, const StepParams& sp, const BuildParams& p) {
    constexpr int APT = SbPre<NT>::APT;
    const int tid = threadIdx.x;
    const int* in_start1 = p.in_start + p.N; const int* in_cnt1 = p.in_cnt + p.N;
    const int nf_tot = p.N - a_Np_tot;                                // centers of the batch (>= 1 wherever a kernel runs this)
    q.isf = tid < q.Nf;
    const int frow = max(min(q.f0 + tid, nf_tot - 1), 0);            // clamped center row
    {
        const float4 x = sp.xn[a_Np_tot + frow];
        q.xf = q.isf ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        pf_gcf nz = (pf_gcf)sp.noise + (size_t)frow * (3 + sp.nf);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float v = nz[c]; q.nzx[c] = q.isf ? v : 0.f; }
#pragma unroll
        for (int k = 0; k < SB_MAXNF; ++k) {
            q.hv[k] = 0.f; q.nzh[k] = 0.f;
            if constexpr (FEAT) {
                const int kc = min(k, sp.nf - 1);
                const float hvv = ((pf_gcf)sp.pharm_h)[(size_t)frow * sp.nf + kc], nzv = nz[3 + kc];
                const bool on = q.isf && k < sp.nf;
                q.hv[k] = on ? hvv : 0.f;
                q.nzh[k] = on ? nzv : 0.f;
            }
        }
    }
    const bool stat = p.act_ids && !p.pa_static;                      // kernel-uniform
#pragma unroll
    for (int a = 0; a < APT; ++a) {
        const int c = sb_atom<NT>(tid, a);
        q.isp[a] = c < q.Np;
        const int arow = max(min(q.p0 + c, a_Np_tot - 1), 0);
        const float4 x = sp.xn[arow];
        q.xp[a] = q.isp[a] ? x : make_float4(0.f, 0.f, 0.f, 0.f);
        q.pst[a] = 0; q.pdeg[a] = 0;
        if (stat) {
            const int s0 = in_start1[arow], d0 = in_cnt1[arow];
            q.pst[a] = q.isp[a] ? s0 : 0;
            q.pdeg[a] = q.isp[a] ? d0 : 0;
        }
        q.pty[a] = 0;
        if (p.rec) q.pty[a] = p.ptype[arow];                          // kernel-uniform
        q.ost[a] = 0; q.ocn[a] = 0; q.ostamp[a] = 0;
        if (p.pa_stamp) {                                             // kernel-uniform
            q.ost[a] = p.in_start[2 * p.N + arow]; q.ocn[a] = p.in_cnt[2 * p.N + arow]; q.ostamp[a] = p.pa_stamp[arow];
        }
    }
}
// (C) static pp sources of this thread's atoms (used only if an atom turns out to be active); branch-free like (B)
template <int NT>
__device__ __forceinline__ void sb_load_c(SbPre<NT>& q, const BuildParams& p) {
    const bool stat = p.act_ids && !p.pa_static;                      // kernel-uniform
#pragma unroll
    for (int a = 0; a < SbPre<NT>::APT; ++a)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int v = 0;
            if (stat) v = ((const int PF_AS1*)p.esrc)[q.pst[a] + min(k, max(q.pdeg[a] - 1, 0))];     // (an atom that is not there: slot 0)
            q.psrc[a][k] = q.isp[a] ? v : 0;
        }
}

// the prefetched pp sources go to LDS first, in front of every store (see sb_load_b); their reader is the copy of the active
// atoms' in-edges at the very end.  (A caller that has to wait for eps anyway -- k_rg_node_hs_build -- does this in front of its
// wait: it is the first use of trip (C)'s loads, which the compiler otherwise sinks behind the wait.)
template <int NT>
__device__ __forceinline__ void sb_stage_sources(const SbPre<NT>& q, const BuildParams& p, StepBuildLds& L) {
    constexpr int APT = SB_MAXA / NT;
    const int tid = threadIdx.x;
    if (p.act_ids && !p.pa_static) {                   // kernel-uniform
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            int4* st = reinterpret_cast<int4*>(&L.a_src[sb_atom<NT>(tid, a)][0]);
            st[0] = make_int4(q.psrc[a][0], q.psrc[a][1], q.psrc[a][2], q.psrc[a][3]);
            st[1] = make_int4(q.psrc[a][4], q.psrc[a][5], q.psrc[a][6], q.psrc[a][7]);
            st[2] = make_int4(q.psrc[a][8], q.psrc[a][9], q.psrc[a][10], q.psrc[a][11]);
            st[3] = make_int4(q.psrc[a][12], q.psrc[a][13], q.psrc[a][14], q.psrc[a][15]);
        }
    }
}

// NT threads (a multiple of 64, >= 256; SB_MAXA / NT atoms per thread), every thread of the workgroup calls it.
// ex / eh: eps_x [3] and eps_h [nf] of center tid (threads tid < Nf), read by the caller from wherever the head left them.
// Five LDS-only barriers: COM known | shifted coordinates in LDS | neighbour masks complete | scan totals | first slots and active list.
// FEAT = false: the caller has updated the features (sb_load_b<NT, false>).
template <int NT, bool STAGED = false, bool FEAT = true>
__device__ __forceinline__ void sb_finish(const SbPre<NT>& q, const float (&ex)[3], const float (&eh)[SB_MAXNF], const int g,
                                          const StepParams& sp, const BuildParams& p, StepBuildLds& L) {
    constexpr int NW = NT / 64, APT = SB_MAXA / NT;
    static_assert(NT % 64 == 0 && NW >= 4 && NW <= 8 && APT * NT == SB_MAXA, "256 or 512 threads");
    static_assert(PF_MAXF <= 64, "the centers of a graph are the first lanes of wave 0");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = q.p0, Np = q.Np, f0 = q.f0, Nf = q.Nf, GF = q.GF;
    const int reg_ff = q.reg_ff, reg_pf = q.reg_pf, reg_fp = q.reg_fp, reg_pa = q.reg_pa, reg_act = q.reg_act;
    int* in_start0 = p.in_start;          int* in_cnt0 = p.in_cnt;
    int* in_start1 = p.in_start + p.N;    int* in_cnt1 = p.in_cnt + p.N;
    int* in_start2 = p.in_start + 2 * p.N; int* in_cnt2 = p.in_cnt + 2 * p.N;
    const bool isf = q.isf;
    bool isp[APT];
    float4 xp[APT];
    int pst[APT], pdeg[APT];
#pragma unroll
    for (int a = 0; a < APT; ++a) { isp[a] = q.isp[a]; xp[a] = q.xp[a]; pst[a] = q.pst[a]; pdeg[a] = q.pdeg[a]; }
    if constexpr (!STAGED) sb_stage_sources<NT>(q, p, L);
    if (tid == 0) L.chgmin = 0x7fffffff;              // (written behind three barriers, read behind a fourth)
    // ---- feature update of the pharm nodes (pharmacodiff.py:414-420; independent of everything else, inputs in registers)
    if constexpr (FEAT) {
        if (isf) {
#pragma unroll
            for (int k = 0; k < SB_MAXNF; ++k) {
                if (k < sp.nf) {
                    const float hn = pf_feat_update(q.hv[k], eh[k], q.nzh[k], sp.a_ts, sp.var, sp.sigma, sp.ep_zt, sp.ep_pred, sp.ep_feat);
                    sp.pharm_h[(size_t)(f0 + tid) * sp.nf + k] = hn;
                    if (sp.h_snap_out) sp.h_snap_out[(size_t)(f0 + tid) * sp.nf + k] = hn;
                }
            }
        }
    }
    // ---- coordinate update (pharmacodiff.py:397-426) and COM removal of pharm AND prot coordinates (:429).  The centers are
    // threads 0 .. Nf - 1 of wave 0: the per-graph mean is that wave's butterfly sum -- the summation order of step_update_body
    // (its other three wave sums are sums of zeros; they are added all the same: -0 + 0 = +0)
    float m[3] = {0.f, 0.f, 0.f};
    if (isf) {
        const float xi[3] = {q.xf.x, q.xf.y, q.xf.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            m[c] = pf_feat_update(xi[c], ex[c], q.nzx[c], sp.a_ts, sp.var, sp.sigma, sp.ep_zt, sp.ep_pred, sp.ep_coord);
        }
    }
    if (wave == 0) {
        const float n = (float)max(Nf, 1);
        float r0[3] = {m[0], m[1], m[2]};
        wave_xor_sum3(r0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0));      // (an opaque zero: the additions stay)
            if (lane == 0) L.com[c] = Nf > 0 ? (((r0[c] + z) + (z + z)) / n) : 0.f;
        }
    }
    sb_lds_barrier();
    SB_PHASE(8);
    const float com[3] = {L.com[0], L.com[1], L.com[2]};
    if (isf) {
        const float4 v = make_float4(m[0] - com[0], m[1] - com[1], m[2] - com[2], 0.f);
        L.fx[tid] = v;
        sp.xn[GF + tid] = v;
    }
    const bool two = Np > NT;                         // workgroup-uniform: pass 1 has atoms
#pragma unroll
    for (int a = 0; a < APT; ++a) {
        const int c = sb_atom<NT>(tid, a);
        if (isp[a]) { xp[a].x -= com[0]; xp[a].y -= com[1]; xp[a].z -= com[2]; sp.xn[p0 + c] = xp[a]; }
        if (a == 0 || two) {                          // (the neighbour search reads atoms beyond NT only when there are any)
            if (p.rec) L.a_ty[c] = (unsigned char)q.pty[a];
            L.px[c] = xp[a];
            L.refm[c][0] = 0u; L.refm[c][1] = 0u;
        }
    }
    sb_lds_barrier();
    // ---- ff (pharm -> pharm) on the last wave: counts, wave scan for the offsets, emission
    const int kff = p.ff_k > 0 ? max(min(p.ff_k, Nf - 1), 0) : 0;
    int ff_total = 0;                                 // valid in the last wave
    if (wave == NW - 1) {
        int c = 0;
        unsigned long long inr = 0ull;                // radius mode: bit jn set <=> center jn lies within the radius of center `lane`
        if (p.ff_k > 0) c = lane < Nf ? kff : 0;
        else {
            const float4 me = L.fx[min(lane, max(Nf - 1, 0))];
            for (int j0 = 0; j0 < Nf; j0 += 4) {      // four LDS reads in flight (the sources are broadcast reads)
                float4 o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = L.fx[min(j0 + u, Nf - 1)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jn = j0 + u;
                    const bool in = jn < Nf && jn != lane && sqdist_rn(o[u], me) < p.r2_ff;
                    inr |= in ? (1ull << jn) : 0ull;
                }
            }
            if (lane >= Nf) inr = 0ull;
            c = __popcll(inr);
        }
        const int incl = (int)wave_incl_scan_u32((unsigned int)c);
        ff_total = __builtin_amdgcn_readlane(incl, 63);
        if (lane == 0) p.dyn_cnt[0 * p.B + g] = ff_total;
        const int my_e0 = reg_ff + incl - c;
        if (lane < Nf) {
            in_start0[GF + lane] = my_e0;
            in_cnt0[GF + lane] = c;
        }
        if (p.ff_k > 0) {                                 // (kernel-uniform; no edge records in this mode: the host does not ask for them)
            if (lane < Nf) {
                int e = my_e0;
                unsigned long long prev = 0ull;
                bool first = true;
                for (int qq = 0; qq < kff; ++qq) {
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
            }
        } else {
            // every lane walks its in-range centers in ascending order; the loop is wave-uniform (as many rounds as the longest list) so
            // that, for BuildParams::rec, an edge's record can take the SOURCE center's descriptors and coordinates out of lane jn's
            // registers through the LDS crossbar (ds_bpermute reads active lanes only: five requests in flight, no memory)
            const float4 my_x = L.fx[min(lane, max(Nf - 1, 0))];
            const int kk_pf = min(p.pf_k, Np);
            int e = my_e0;
            while (__builtin_amdgcn_ballot_w64(inr != 0ull) != 0ull) {
                const bool has = inr != 0ull;
                const int jn = has ? __ffsll((long long)inr) - 1 : 0;
                inr &= inr - 1ull;                        // (0 stays 0)
                int se = 0, sc = 0, sx = 0, sy = 0, sz = 0;
                if (p.rec) {                              // kernel-uniform
                    const int a4 = 4 * jn;
                    se = __builtin_amdgcn_ds_bpermute(a4, my_e0); sc = __builtin_amdgcn_ds_bpermute(a4, c);
                    sx = __builtin_amdgcn_ds_bpermute(a4, __builtin_bit_cast(int, my_x.x));
                    sy = __builtin_amdgcn_ds_bpermute(a4, __builtin_bit_cast(int, my_x.y));
                    sz = __builtin_amdgcn_ds_bpermute(a4, __builtin_bit_cast(int, my_x.z));
                }
                if (has) {
                    p.esrc[e] = GF + jn; p.edst[e] = GF + lane;
                    if (p.rec) {
                        int4* r = p.rec + (size_t)3 * e;
                        r[0] = make_int4(sx, sy, sz, GF + jn);
                        r[1] = make_int4(__builtin_bit_cast(int, my_x.x), __builtin_bit_cast(int, my_x.y), __builtin_bit_cast(int, my_x.z), GF + lane);
                        r[2] = make_int4(se, sc, reg_pf + jn * kk_pf, kk_pf);
                    }
                    ++e;
                }
            }
        }
        SB_STAMP(21);
    }
    SB_PHASE(9);
    // ---- pf (prot -> pharm): kNN, one center per HALF wave (knn_halfwave), candidates from LDS; lane r of a half ends with
    // the r-th neighbour of its center: one coalesced store per half
    const int kk = min(p.pf_k, Np);
    const int hw = lane >> 5, l32 = lane & 31;
    for (int fb = 2 * wave; fb < Nf; fb += 2 * NW) {
        const int fl = fb + hw;                       // this half's center (fb + 1 may lie beyond the graph: that half searches
        const float4 qc = L.fx[min(fl, Nf - 1)];      // the last center again and writes nothing)
        bool exact;
        int pc = Np <= 256 ? knn_halfwave_keys<8>(L.px, qc, Np, kk, lane, exact) : knn_halfwave_keys<16>(L.px, qc, Np, kk, lane, exact);   // wave-uniform
        if (!exact) pc = Np <= 256 ? knn_halfwave<8>(L.px, qc, Np, kk, lane) : knn_halfwave<16>(L.px, qc, Np, kk, lane);          // (wave-uniform, rare)
        if (l32 < kk && fl < Nf) {
            atomicOr(&L.refm[pc][fl >> 5], 1u << (fl & 31));
            L.pairs[fl * kk + l32] = pc | (fl << 16);
            p.esrc[reg_pf + fl * kk + l32] = p0 + pc;
            p.edst[reg_pf + fl * kk + l32] = GF + fl;
        }
        if (l32 == 0 && fl < Nf) { in_start1[GF + fl] = reg_pf + fl * kk; in_cnt1[GF + fl] = kk; }
    }
    if (tid == 0) { p.dyn_cnt[1 * p.B + g] = Nf * kk; p.dyn_cnt[2 * p.B + g] = Nf * kk; }
    SB_STAMP(20);
    sb_lds_barrier();
    SB_PHASE(10);
    // ---- fp = pf reversed, destination-major over the atoms; active atoms and the compact copy of their pp in-edges
    {
        // the centers that reference an atom, ascending: the bits of its mask (set by the kNN waves above)
        unsigned int m0[APT], m1[APT];
        int my[APT], act[APT], deg[APT];
        unsigned long long val[APT], incl[APT];
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            const int c = sb_atom<NT>(tid, a);
            m0[a] = 0u; m1[a] = 0u;
            if (a == 0 || two) { m0[a] = isp[a] ? L.refm[c][0] : 0u; m1[a] = isp[a] ? L.refm[c][1] : 0u; }
            my[a] = __popc(m0[a]) + __popc(m1[a]);
            act[a] = (my[a] > 0 && p.act_ids) ? 1 : 0;
            deg[a] = act[a] ? pdeg[a] : 0;
            val[a] = (unsigned long long)my[a] | ((unsigned long long)act[a] << 16) | ((unsigned long long)deg[a] << 28);
        }
        SB_PHASE(12);                                 // references counted
        // block scan in atom order = (pass, thread): the value packs three counters (bits 0-15 fp edges, 16-27 active atoms, 28-63 their
        // pp in-edges), scanned as two 32-bit halves (the low two cannot carry into each other at these sizes); one wave scan per pass
        // that has atoms, the waves' totals through LDS
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            incl[a] = 0ull;
            if (a == 0 || two) {
                const unsigned int lo = (unsigned int)(val[a] & 0xfffffffull), hi = (unsigned int)(val[a] >> 28);
                const unsigned int slo = wave_incl_scan_u32(lo), shi = wave_incl_scan_u32(hi);
                incl[a] = (unsigned long long)slo | ((unsigned long long)shi << 28);
                if (lane == 63) L.scratch[a][wave] = incl[a];
            }
        }
        sb_lds_barrier();
        unsigned long long obase[APT], all = 0ull;
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            obase[a] = all;                           // everything of the passes below
            if (a == 0 || two) {
                unsigned long long before = 0ull;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const unsigned long long t = L.scratch[a][w];
                    if (w < wave) before += t;
                    all += t;
                }
                obase[a] += before + incl[a] - val[a];
            }
        }
        SB_PHASE(13);                                 // offsets known
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            if (a > 0 && !two) break;                 // workgroup-uniform
            const int c = sb_atom<NT>(tid, a);
            const unsigned long long o = obase[a];
            if (isp[a]) {
                const int e = reg_fp + (int)(o & 0xffffu);
                in_start0[p0 + c] = e;
                in_cnt0[p0 + c] = my[a];
                L.a_e0[c] = e;
            }
            // an active atom: its place in the list and (unless the pocket's pp messages are shared) the slots of the "pa" region
            // that receive the compact copy of its static pp in-edges
            if (p.pa_stamp && isp[a]) {
                // unchanged = active in the previous step too, at the same slots (its stamp is that step's, its slot-2 range the same).
                // The region is valid up to the first slot anything changed at: where a new or moved atom now starts, where a moved
                // or departed atom used to start
                const bool was = q.ostamp[a] == p.step_id - 1;
                const int nd0 = (int)(o >> 28), od0 = q.ost[a] - reg_pa;
                int first = 0x7fffffff;
                if (act[a]) {
                    const bool same = was && !p.pa_static && od0 == nd0 && q.ocn[a] == deg[a];
                    if (!same) first = was ? min(od0, nd0) : nd0;
                } else if (was) first = od0;
                if (first != 0x7fffffff) atomicMin(&L.chgmin, max(first, 0));
                if (act[a]) p.pa_stamp[p0 + c] = p.step_id;
            }
            if (isp[a] && act[a]) {
                const int j = (int)((o >> 16) & 0xfffu);
                p.act_ids[reg_act + j] = p0 + c;
                if (p.pa_static) p.need[p.rep_base[g] + c] = p.need_stamp;
                else {
                    const int d0 = (int)(o >> 28);
                    in_start2[p0 + c] = reg_pa + d0;
                    in_cnt2[p0 + c] = deg[a];
                    L.act4[j] = make_int4(c, d0, deg[a], pst[a]);
                    if (p.rec) L.a_pa[c] = make_int2(reg_pa + d0, deg[a]);
                }
            }
        }
        SB_PHASE(14);                                 // descriptors stored, first slots and active atoms staged
        sb_lds_barrier();
        SB_PHASE(15);
        // fp edges: one per (center, neighbour) pair, at the atom's first slot + the referencing centers below this one.  The "pa"
        // region: 16 threads per active atom, runs of coalesced stores from the prefetched static sources.  Both are two LDS round
        // trips in front of their stores: the first pass of each (what a 256-atom pocket with <= 32 active atoms needs) shares them.
        const int npairs = Nf * kk;
        const bool pa_on = p.act_ids && !p.pa_static;                                   // kernel-uniform
        const int n_pa16 = pa_on ? 16 * (int)((all >> 16) & 0xfffu) : 0;
        // t: the pair's index = its pf slot behind reg_pf.  BuildParams::rec: the pf edge's record -- the source atom's coordinates, type and
        // in-edge descriptors (its fp in-edges: first slot and count, the popcount of its reference mask; its "pa" region)
        auto fp_store = [&](const int t, const int pr, const unsigned int r0, const unsigned int r1, const int e0) {
            const int pc = pr & 0xffff, fl = pr >> 16;
            const int below = fl < 32 ? __popc(r0 & ((1u << fl) - 1u)) : __popc(r0) + __popc(r1 & ((1u << (fl - 32)) - 1u));
            p.esrc[e0 + below] = GF + fl;
            p.edst[e0 + below] = p0 + pc;
            if (p.rec) {                                  // kernel-uniform
                const float4 xs = L.px[pc], xd = L.fx[fl];
                const int2 pa = L.a_pa[pc];
                const int ty = (int)L.a_ty[pc];
                int4* r = p.rec + (size_t)3 * (reg_pf + t);
                r[0] = make_int4(__builtin_bit_cast(int, xs.x), __builtin_bit_cast(int, xs.y), __builtin_bit_cast(int, xs.z), (p0 + pc) | (ty << 24));
                r[1] = make_int4(__builtin_bit_cast(int, xd.x), __builtin_bit_cast(int, xd.y), __builtin_bit_cast(int, xd.z), GF + fl);
                r[2] = make_int4(e0, __popc(r0) + __popc(r1), pa.x, pa.y);
            }
        };
        auto pa_store = [&](const int4 a4, const int k, const int src) {
            const int c = a4.x, e0 = reg_pa + a4.y, dg = a4.z, ps = a4.w;
            if (k < dg) {
                p.esrc[e0 + k] = src;
                p.edst[e0 + k] = p0 + c;
                if (p.eorig) p.eorig[e0 + k] = ps + k;
            }
            if (k == 15)
                for (int k2 = 16; k2 < dg; ++k2) {        // (in-degrees beyond 16: their sources were not prefetched)
                    p.esrc[e0 + k2] = p.esrc[ps + k2];
                    p.edst[e0 + k2] = p0 + c;
                    if (p.eorig) p.eorig[e0 + k2] = ps + k2;
                }
        };
        {
            const bool hp = tid < npairs, h0 = tid < n_pa16, h1 = tid + NT < n_pa16;
            const int pr = L.pairs[hp ? tid : 0];
            const int4 a40 = L.act4[h0 ? tid >> 4 : 0], a41 = L.act4[h1 ? (tid + NT) >> 4 : 0];
            const int pc = hp ? (pr & 0xffff) : 0, k = tid & 15;
            const unsigned int r0 = L.refm[pc][0], r1 = L.refm[pc][1];
            const int e0 = L.a_e0[pc];
            const int s0 = L.a_src[h0 ? a40.x : 0][k], s1 = L.a_src[h1 ? a41.x : 0][k];
            if (hp) fp_store(tid, pr, r0, r1, e0);
            if (h0) pa_store(a40, k, s0);
            if (h1) pa_store(a41, k, s1);
        }
        for (int t = tid + NT; t < npairs; t += NT) {
            const int pr = L.pairs[t], pc = pr & 0xffff;
            fp_store(t, pr, L.refm[pc][0], L.refm[pc][1], L.a_e0[pc]);
        }
        for (int t = tid + 2 * NT; t < n_pa16; t += NT) {
            const int4 a4 = L.act4[t >> 4];
            pa_store(a4, t & 15, L.a_src[a4.x][t & 15]);
        }
        if (tid == 0 && p.act_ids) {
            p.dyn_cnt[3 * p.B + g] = p.pa_static ? p.pa_static[g] : (int)(all >> 28);
            p.dyn_cnt[4 * p.B + g] = (int)((all >> 16) & 0xfffu);
            // the number of leading 16-slot groups of the region whose rows, computed ahead, still apply (everything: 0x7fffffff)
            if (p.pa_same) p.pa_same[g] = p.pa_static ? 0 : (L.chgmin == 0x7fffffff ? 0x7fffffff : (L.chgmin >> 4));
        }
        if (tid == NT - 64 && p.norm_mode == 2) {     // per-graph normalisers for message_norm == 0 (gvp.py:504-507)
            const int cpf = p.pfq_cnt ? p.pfq_cnt[g] : Nf * kk;
            p.gnorm[1 * p.B + g] = (float)(ff_total + cpf) / (float)Nf + 1.0f;
            p.gnorm[0 * p.B + g] = (float)(cpf + p.pp_cnt[g]) / (float)Np + 1.0f;
        }
    }
    SB_PHASE(11);
}

// the three trips and the rest in one piece (k_step_build_fast: eps comes from memory, nothing to overlap the loads with)
template <int NT, class Eps>
__device__ __forceinline__ void step_build_fast_body(const int g, const int* __restrict__ a_prot_ptr, const int* __restrict__ a_pharm_ptr,
                                                     const int* __restrict__ a_reg, const int a_B, const int a_Np_tot,
                                                     const StepParams& sp, const BuildParams& p, const Eps& eps, StepBuildLds& L) {
    SB_STAMP(0);
    SbPre<NT> q;
    sb_load_a<NT>(q, g, a_prot_ptr, a_pharm_ptr, a_reg, a_B, a_Np_tot, p);
    sb_load_b<NT>(q, a_Np_tot, sp, p);
    float ex[3] = {0.f, 0.f, 0.f}, eh[SB_MAXNF];
#pragma unroll
    for (int c = 0; c < 3; ++c) ex[c] = q.isf ? eps.x((int)threadIdx.x, c) : 0.f;               // with trip (B)
#pragma unroll
    for (int k = 0; k < SB_MAXNF; ++k) eh[k] = (q.isf && k < sp.nf) ? eps.h((int)threadIdx.x, k) : 0.f;
    sb_load_c<NT>(q, p);
    sb_finish<NT>(q, ex, eh, g, sp, p, L);
}

// the same with eps arriving through the exchange words of a node + head item of the SAME launch (k_rg_node_hs_build): all three
// trips are issued first, then the threads of the graph's centers poll their row's words (each word is its own flag: no ordering
// between words is needed), take them and re-arm them for the next step.  Bounded: after poll_max rounds (SB_XCHG_POLLS unless
// pf_debug_xchg_fault shortened it) the thread counts a time-out in xstat and goes on with zeros.  The run is then invalid, and the
// host says so on the same run: xstat is cumulative, its copy travels behind pf_sample_end and pf_sample_status compares it with
// what was acknowledged.  A timed-out row is NOT re-armed here -- its producer may still store into it, and a re-armed word
// filled late would read as "arrived" in every later step without ever being counted; pf_sample_begin re-arms every word.
constexpr int SB_XCHG_POLLS = 1 << 16;
// one word of a row: polled until it holds a payload (true) or poll_max rounds have passed (false: counted in xstat)
__device__ __forceinline__ bool sb_poll_word(unsigned int* w, unsigned int& bits, int* xstat, const int poll_sleep, const int poll_max) {
    bits = PF_XCHG_EMPTY;
    for (int it = 0; it < poll_max; ++it) {
        bits = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bits != PF_XCHG_EMPTY) return true;
        for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(2);      // ~50 ns each: the words come from memory (agent scope), not from a cache
    }
    atomicAdd(xstat, 1);
    return false;
}
// Who waits for what (round 5): a center's thread (wave 0) polls the THREE coordinate words of its row, all three loads in flight at
// once; the feature words are polled one per lane by the threads behind wave 0 (lane j of them: feature j % nf of center j / nf),
// which update the features there and then -- off the path COM -> kNN that the step waits for.  (The first form polled a row's 3 + nf
// words from one thread with a branch per word: nine dependent round trips to memory per poll round, ~2 us of every step.)
template <int NT>
__device__ __forceinline__ void step_build_wait_body(const int g, const int* __restrict__ a_prot_ptr, const int* __restrict__ a_pharm_ptr,
                                                     const int* __restrict__ a_reg, const int a_B, const int a_Np_tot,
                                                     const StepParams& sp, const BuildParams& p, unsigned int* xchg, int* xstat,
                                                     StepBuildLds& L, const int poll_sleep, const int poll_max) {
    const int tid = threadIdx.x;
    SbPre<NT> q;
    sb_load_a<NT>(q, g, a_prot_ptr, a_pharm_ptr, a_reg, a_B, a_Np_tot, p);
    sb_load_b<NT, false>(q, a_Np_tot, sp, p);
    sb_load_c<NT>(q, p);
    // the feature lanes' old value and draw (the first round of their loop; a graph with more than NT - 64 feature words loads the rest on demand)
    const int nfeat = q.Nf * sp.nf, fj = tid - 64;
    const int nf_tot = p.N - a_Np_tot;
    float f_hv = 0.f, f_nz = 0.f;
    {
        const int jc = max(min(fj, nfeat - 1), 0), fc = jc / sp.nf, kc = jc - fc * sp.nf;
        const int frow = max(min(q.f0 + fc, nf_tot - 1), 0);
        f_hv = ((pf_gcf)sp.pharm_h)[(size_t)frow * sp.nf + kc];
        f_nz = ((pf_gcf)sp.noise)[(size_t)frow * (3 + sp.nf) + 3 + kc];
    }
    float ex[3] = {0.f, 0.f, 0.f}, eh[SB_MAXNF];
#pragma unroll
    for (int k = 0; k < SB_MAXNF; ++k) eh[k] = 0.f;
    // everything the update + build reads is requested -- and has ARRIVED -- before the wait: the static sources go to LDS (their
    // first use), the other rows are pinned in registers (an empty asm the compiler cannot move the loads across)
    sb_stage_sources<NT>(q, p, L);
    {
        float4 pin = q.xf;
#pragma unroll
        for (int a = 0; a < SbPre<NT>::APT; ++a) { pin.x += q.xp[a].x; pin.y += q.xp[a].y; pin.z += q.xp[a].z; pin.w += (float)(q.pst[a] + q.pdeg[a] + q.ost[a] + q.ocn[a] + q.ostamp[a]); }
#pragma unroll
        for (int c = 0; c < 3; ++c) pin.x += q.nzx[c];
        pin.y += f_hv + f_nz;
        asm volatile("" :: "v"(pin.x), "v"(pin.y), "v"(pin.z), "v"(pin.w) : "memory");
    }
    SB_STAMP(100);
    if (q.isf) {                                       // (threads of wave 0: PF_MAXF <= 64)
        unsigned int* row = xchg + (size_t)(q.f0 + tid) * PF_XCHG_STRIDE;
        unsigned int w[3];
        bool ok = false;
        for (int it = 0; it < poll_max; ++it) {
#pragma unroll
            for (int c = 0; c < 3; ++c) w[c] = __hip_atomic_load(row + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = w[0] != PF_XCHG_EMPTY && w[1] != PF_XCHG_EMPTY && w[2] != PF_XCHG_EMPTY;
            if (ok) break;
            for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) atomicAdd(xstat, 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) ex[c] = ok ? __builtin_bit_cast(float, w[c]) : 0.f;
        if (ok) {
#pragma unroll
            for (int c = 0; c < 3; ++c) __hip_atomic_store(row + c, PF_XCHG_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (fj >= 0) {
        for (int j = fj; j < nfeat; j += NT - 64) {
            const int fc = j / sp.nf, k = j - fc * sp.nf;
            const size_t fr = (size_t)(q.f0 + fc);
            float hv = f_hv, nz = f_nz;
            if (j != fj) { hv = ((pf_gcf)sp.pharm_h)[fr * sp.nf + k]; nz = ((pf_gcf)sp.noise)[fr * (3 + sp.nf) + 3 + k]; }
            unsigned int* w = xchg + fr * PF_XCHG_STRIDE + 3 + k;
            unsigned int bits;
            const bool ok = sb_poll_word(w, bits, xstat, poll_sleep, poll_max);
            const float hn = pf_feat_update(hv, ok ? __builtin_bit_cast(float, bits) : 0.f, nz, sp.a_ts, sp.var, sp.sigma, sp.ep_zt, sp.ep_pred, sp.ep_feat);
            sp.pharm_h[fr * sp.nf + k] = hn;
            if (sp.h_snap_out) sp.h_snap_out[fr * sp.nf + k] = hn;
            if (ok) __hip_atomic_store(w, PF_XCHG_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    SB_STAMP(101);
    sb_finish<NT, true, false>(q, ex, eh, g, sp, p, L);
}

}  // namespace pfsb
