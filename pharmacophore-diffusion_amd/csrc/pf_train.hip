// pf_train.hip -- gradient kernels of libpfdyn (gfx950): backward of the noise head, of a conv layer's node update
// and edge messages, and of the encoders.  See pf_train.h for the data model.
//
// Reference semantics: autograd through NoisePredictionBlock.forward (dynamics_gvp.py:37-42), GVPMultiEdgeConv
// (gvp.py:459-551), GVP.forward (gvp.py:89-116), GVPLayerNorm (gvp.py:152-166), GVPDropout (gvp.py:118-149) and the
// encoders (dynamics_gvp.py:107-117).
//
// A backward tile is 16 rows (edges or nodes).  Its activations sit in LDS as [row][feature] with odd row strides;
// every GEMM-shaped step -- forward recompute, input gradients and weight gradients -- goes through mm16, a block
// cooperative v_mfma_f32_16x16x4_f32 loop whose operands are fetched by accessor lambdas (weights from the flat
// parameter vector, activations from LDS), with the 16 rows (or 16 rows x 3 coordinates) as the N dimension of the
// data products and as the K dimension of the weight-gradient products.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "pf_train.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TR PFT_ROWS
#define NT 512            // threads per block (8 waves: two per SIMD hide the operand-fetch latency of mm16)
#define FPP (128 / (NT / 16))   // features per thread in the per-row passes (thread = (row = tid & 15, part = tid >> 4))
#define SWS 165           // LDS row stride of scalar rows [si + h] (<= 161)
#define VWS 53            // ... of vector rows [vi * 3] (<= 51)
#define ZS 129            // ... of pre-activation / activation rows [so]
#define GTS 17            // ... of gate rows [vo]
#define LVL_FLOATS (TR * (SWS + VWS + ZS + GTS))
#define WORK_FLOATS (TR * (SWS * 2 + VWS * 5 + GTS))
#define CHAIN_FLOATS(nlv) ((nlv) * LVL_FLOATS + TR * ZS + WORK_FLOATS)
#define PFT_WST_FLOATS 3200   // >= vi h + h vo + vo so of any GVP (h = max(vi, vo) <= 17: 17 x 17 + 17 x 16 + 16 x 128 = 2609)

// ---- bf16 leg (TrainCommon::bf16): the products of to_feats_out and of the gate Linear run on v_mfma_f32_16x16x32_bf16 /
// v_mfma_f32_16x16x16_bf16 with operands rounded to nearest even (v_cvt_pk_bf16_f32) and fp32 accumulation.  Lane (li, kq)
// holds 8 (or 4) K-values of row / column li; A and B use the same assignment of K to (kq, element), which is all a dot
// product needs, and the C/D layout is that of v_mfma_f32_16x16x4_f32 -- so the f32 code's fragment tables, LDS images and
// accumulators are shared, only the operand fetch differs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 bf_pack8(const float (&x)[8]) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)x[i];
    return r;
}
__device__ __forceinline__ bf16x8 bf_pack8(const f32x4 lo, const f32x4 hi) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = (__bf16)lo[i]; r[4 + i] = (__bf16)hi[i]; }
    return r;
}
__device__ __forceinline__ s16x4 bf_pack4(const float (&x)[4]) {
    bf16x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (__bf16)x[i];
    return __builtin_bit_cast(s16x4, r);
}
__device__ __forceinline__ f32x4 mfma_bf32(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16(const s16x4 a, const s16x4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float t_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float t_silu(float x) { return x * t_sigmoid(x); }
__device__ __forceinline__ float t_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ float drop_mul(const TrainCommon& c, const uint32_t stream, const uint32_t elem) {
    if (c.mask_override != nullptr) return c.mask_override[(size_t)stream * c.mask_N * 144u + elem];
    if (c.drop_thr == 0u) return 1.0f;
    return pf_drop_hash(c.seed, stream, elem) < c.drop_thr ? 0.0f : c.drop_scale;
}

// C[M x N] = A[M x K] B[K x N] on v_mfma_f32_16x16x4_f32 (lane l: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// C[i = 4 (l>>4) + r][j = l&15]).  Output tiles are dealt round-robin to the waves; c(i, j, value) consumes them.
// UNR k-steps are fetched before their MFMAs are issued, so the operand loads of a tile overlap.
template <int UNR, typename FA, typename FB, typename FC>
__device__ __forceinline__ void mm16(const int M, const int N, const int K, FA a, FB b, FC c, const int lane, const int wv) {
    const int mts = (M + 15) >> 4, nts = (N + 15) >> 4;
    const int li = lane & 15, kq = lane >> 4;
    for (int t = wv; t < mts * nts; t += NT / 64) {
        const int mt = t / nts, nt = t - mt * nts;
        const int ai = mt * 16 + li, bj = nt * 16 + li;
        const bool aok = ai < M, bok = bj < N;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 4 * UNR) {
            float av[UNR], bv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                // operands are fetched unconditionally at clamped indices and masked afterwards: a guarded fetch compiles to an
                // exec-masked branch per element, which is what these loops used to spend their time on
                const int k = k0 + 4 * u + kq, kc = min(k, K - 1);
                const float xa = a(min(ai, M - 1), kc), xb = b(kc, min(bj, N - 1));
                av[u] = (aok && k < K) ? xa : 0.f;
                bv[u] = (bok && k < K) ? xb : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = mt * 16 + kq * 4 + r;
            if (ci < M && bok) c(ci, bj, acc[r]);
        }
    }
}
// G[M x N] += A[M x K] B[K x N] into this block's gradient copy (row stride ld): two output tiles per pass, their old
// values fetched before the MFMAs so that the read-modify-write latency overlaps the products
template <int UNR, typename FA, typename FB>
__device__ __forceinline__ void mm16_acc(const int M, const int N, const int K, FA a, FB b, float* G, const int ld,
                                         const int lane, const int wv, const bool fresh = false) {
    const int mts = (M + 15) >> 4, nts = (N + 15) >> 4;
    const int li = lane & 15, kq = lane >> 4;
    const int ntile = mts * nts;
    for (int t0 = wv; t0 < ntile; t0 += 2 * (NT / 64)) {
        float old[2][4];
        f32x4 acc[2];
        int mtv[2], bjv[2];
        bool live[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int t = t0 + x * (NT / 64);
            live[x] = t < ntile;
            const int mt = live[x] ? t / nts : 0, nt = live[x] ? t - mt * nts : 0;
            mtv[x] = mt; bjv[x] = nt * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = mt * 16 + kq * 4 + r;
                old[x][r] = 0.f;                                   // fresh: known zeros (zero_class)
                if (!fresh) {                                      // block-uniform
                    const float o = G[min(ci, M - 1) * ld + min(bjv[x], N - 1)];
                    old[x][r] = (live[x] && ci < M && bjv[x] < N) ? o : 0.f;
                }
            }
            acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int ai = mtv[x] * 16 + li;
            const bool aok = live[x] && ai < M, bok = live[x] && bjv[x] < N;
            for (int k0 = 0; k0 < K; k0 += 4 * UNR) {
                float av[UNR], bv[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int k = k0 + 4 * u + kq, kc = min(k, K - 1);
                    const float xa = a(min(ai, M - 1), kc), xb = b(kc, min(bjv[x], N - 1));
                    av[u] = (aok && k < K) ? xa : 0.f;
                    bv[u] = (bok && k < K) ? xb : 0.f;
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc[x], 0, 0, 0);
            }
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = mtv[x] * 16 + kq * 4 + r;
                if (live[x] && ci < M && bjv[x] < N) G[ci * ld + bjv[x]] = old[x][r] + acc[x][r];
            }
    }
}

// clear the tensors of class `cls` in this block's gradient copy (start of a kernel that accumulates per tile); the
// class's ranges are collected by one pass of the block over the tensor table
__device__ __forceinline__ void zero_class(const TrainCommon& c, float* gp, const int cls, const int tid) {
    __shared__ int s_rng[2 * 96], s_nr;
    if (tid == 0) s_nr = 0;
    __syncthreads();
    for (int t = tid; t < c.ntens; t += NT) {
        const TensorSeg s = c.tseg[t];
        if (s.cls == cls && s.end > s.begin) {
            const int k = atomicAdd(&s_nr, 1);
            if (k < 96) { s_rng[2 * k] = s.begin; s_rng[2 * k + 1] = s.end; }
        }
    }
    __syncthreads();
    const int nr = min(s_nr, 96);
    for (int k = 0; k < nr; ++k)
        for (int i = s_rng[2 * k] + tid; i < s_rng[2 * k + 1]; i += NT) gp[i] = 0.f;
    __syncthreads();
}
// blocks [b0, b0 + nbk) of an edge-level launch of NB blocks serve etype `et`: in proportion to the non-empty tiles, at
// least one where there are tiles, never more than tiles
__device__ __forceinline__ void et_blocks(const int* ccnt, const int n_et, const int NB, const int et, int& b0, int& nbk) {
    int cnt[4], tot = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { cnt[e] = e < n_et ? ccnt[e] : 0; tot += cnt[e]; }
    b0 = 0; nbk = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = cnt[e] > 0 ? min(cnt[e], max(1, (int)((long long)(NB - 3) * cnt[e] / tot))) : 0;
        if (e < et) b0 += n;
        if (e == et) nbk = n;
    }
}

// C[i][j] (i < 16 mts, j < 16 rows) = sum_k A[i][k] B[k][j] with A from a packed fragment table ([m tile][NSB k blocks][lane] x 4:
// lane (li, kq) holds A[16 mt + li][16 sb + 4 kq + t], t = 0..3; k_pack_gvp) -- one 1-KiB load per four MFMAs instead of
// four strided dword loads -- and B[k][j] = b(k, j) from LDS.  Wave wv takes m tiles wv, wv + 8, ...
template <int NSB, typename FB, typename FC>
__device__ __forceinline__ void mm16_packed(const f32x4* P, const int mts, const int nsb, FB b, FC c, const int lane, const int wv) {
    const int li = lane & 15, kq = lane >> 4;
    for (int mt = wv; mt < mts; mt += NT / 64) {
        const f32x4* pp = P + (size_t)mt * (NSB * 64) + lane;
        f32x4 aq[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) aq[sb] = pp[sb * 64];       // all NSB blocks exist in the table (zero padded): no branches between the loads
        __builtin_amdgcn_sched_barrier(0);                            // (every fragment requested before the first product: sunk to their uses they are NSB round trips)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb)
            if (sb < nsb) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[sb][t], b(16 * sb + 4 * kq + t, li), acc, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) c(mt * 16 + kq * 4 + r, li, acc[r]);
    }
}

// mm16_packed on bf16 matrix instructions (NSB even): instruction s covers k blocks 2 s and 2 s + 1 of the f32 table -- lane
// (li, kq) holds A[.][16 (2 s) + 4 kq + t] and A[.][16 (2 s + 1) + 4 kq + t], t = 0..3, and fetches the same eight k of B
template <int NSB, typename FB, typename FC>
__device__ __forceinline__ void mm16_packed_bf(const f32x4* P, const int mts, const int nsb, FB b, FC c, const int lane, const int wv) {
    static_assert(NSB % 2 == 0, "pairs of k blocks");
    const int li = lane & 15, kq = lane >> 4;
    for (int mt = wv; mt < mts; mt += NT / 64) {
        const f32x4* pp = P + (size_t)mt * (NSB * 64) + lane;
        f32x4 aq[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) aq[sb] = pp[sb * 64];
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp = 0; sp < NSB / 2; ++sp)
            if (2 * sp < nsb) {
                const bool hi = 2 * sp + 1 < nsb;           // (block-uniform; beyond nsb the fragments are zero and B is not read)
                float bv[8];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    bv[t] = b(32 * sp + 4 * kq + t, li);
                    const float x = b(hi ? 32 * sp + 16 + 4 * kq + t : 32 * sp + 4 * kq + t, li);
                    bv[4 + t] = hi ? x : 0.f;
                }
                acc = mfma_bf32(bf_pack8(aq[2 * sp], aq[2 * sp + 1]), bf_pack8(bv), acc);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) c(mt * 16 + kq * 4 + r, li, acc[r]);
    }
}

// G[M x N] += A^T-style weight gradient with the wave owning m tile wv (M <= 128): A[i][k] = a(i, k) is fetched once per wave for
// the K = 16 rows of the unit and serves every n tile; fresh: the block's copy is known to be zero there (store, no read)
template <bool BF16 = false, typename FA, typename FB>
__device__ __forceinline__ void mm16_acc_rows(const int M, const int N, FA a, FB b, float* G, const int ld, const int lane, const int wv,
                                              const bool fresh) {
    const int li = lane & 15, kq = lane >> 4;
    if (wv * 16 >= M) return;
    const int ai = wv * 16 + li;
    // (BF16: one v_mfma_f32_16x16x16_bf16 over the 16 rows, lane (li, kq) holding rows 4 kq + u; f32: four k-steps, rows 4 u + kq)
    float av[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const float x = a(min(ai, M - 1), BF16 ? 4 * kq + u : 4 * u + kq); av[u] = ai < M ? x : 0.f; }
    const int nts = (N + 15) >> 4;
    // Whole tiles (block-uniform): the product is taken TRANSPOSED -- the operands swap roles, a lane
    // then holds four consecutive columns of one row -- and leaves as one 16-byte store per lane and tile instead of four 4-byte
    // stores over four rows (36 -> 9 store instructions per wave and GVP level; the phase was bound by them)
    // (the rows of the flat gradient are 4-byte aligned only: dwordx4 accesses at such addresses are what the hardware's unaligned
    // access mode, the default under HSA, is for)
    typedef f32x4 f32x4_u __attribute__((aligned(4)));
    const bool quad = (M & 15) == 0 && (N & 15) == 0;
    if (quad) {
        for (int nt0 = 0; nt0 < nts; nt0 += 4) {
            float bv[4][4];
            f32x4 old[4], acc[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const bool on = nt0 + x < nts;
                const int bj = (on ? nt0 + x : nt0) * 16 + li;
#pragma unroll
                for (int u = 0; u < 4; ++u) bv[x][u] = b(BF16 ? 4 * kq + u : 4 * u + kq, bj);
                old[x] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!fresh && on) old[x] = *reinterpret_cast<const f32x4_u*>(G + (size_t)ai * ld + (nt0 + x) * 16 + 4 * kq);
                acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (BF16) {
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x] = mfma_bf16(bf_pack4(bv[x]), bf_pack4(av), acc[x]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int x = 0; x < 4; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[x][u], av[u], acc[x], 0, 0, 0);
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
                if (nt0 + x < nts) *reinterpret_cast<f32x4_u*>(G + (size_t)ai * ld + (nt0 + x) * 16 + 4 * kq) = old[x] + acc[x];
        }
        return;
    }
    // n tiles in groups of four: the group's operands (and, when the copy is not fresh, its old values) are all requested before
    // the first product, the products of the group are independent of each other, the stores follow -- one n tile at a time was a
    // chain of LDS read -> four dependent matrix instructions -> store per tile (9 k cycles per GVP level for eleven tiles)
    for (int nt0 = 0; nt0 < nts; nt0 += 4) {
        float bv[4][4], old[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int bj = (nt0 + x) * 16 + li;
            const bool on = nt0 + x < nts && bj < N;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float v = b(BF16 ? 4 * kq + u : 4 * u + kq, min(bj, N - 1)); bv[x][u] = on ? v : 0.f; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = wv * 16 + kq * 4 + r;
                old[x][r] = 0.f;
                if (!fresh) {                                      // block-uniform
                    const float o = G[min(ci, M - 1) * ld + min(bj, N - 1)];
                    old[x][r] = (ci < M && on) ? o : 0.f;
                }
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[x] = mfma_bf16(bf_pack4(av), bf_pack4(bv[x]), acc[x]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[x][u], acc[x], 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int bj = (nt0 + x) * 16 + li;
            if (nt0 + x < nts && bj < N)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = wv * 16 + kq * 4 + r;
                    if (ci < M) G[ci * ld + bj] = old[x][r] + acc[x][r];
                }
        }
    }
}

// diagnostic builds (-DPFT_STAMPS): cycle stamps of block 0's first sub-tile at phase boundaries
#ifdef PFT_STAMPS
#ifndef PFT_STAMP_BLOCK
#define PFT_STAMP_BLOCK 0
#endif
__device__ unsigned long long g_pft_stamps[128];
__device__ int g_pft_nstamp;
#ifndef PFT_STAMP_MIN
#define PFT_STAMP_MIN 0
#endif
#ifndef PFT_STAMP_SKIP_LO
#define PFT_STAMP_SKIP_LO 1000
#define PFT_STAMP_SKIP_HI 1000
#endif
#define PFT_STAMP(id) do { if ((id) >= PFT_STAMP_MIN && !((id) >= PFT_STAMP_SKIP_LO && (id) <= PFT_STAMP_SKIP_HI) && blockIdx.x == PFT_STAMP_BLOCK && threadIdx.x == 0 && g_pft_nstamp < 126) { g_pft_stamps[g_pft_nstamp] = ((unsigned long long)(id) << 48) | (__builtin_readcyclecounter() & 0xffffffffffffull); g_pft_nstamp++; } } while (0)
#else
#define PFT_STAMP(id) do { } while (0)
#endif

struct ChainLds {
    float* base; int nlv;
    float *actl, *gX, *gY, *gVX, *gVY, *Vh, *Vu, *gVh, *ggate;
    __device__ __forceinline__ void init(float* b, const int n) {
        base = b; nlv = n;
        actl = b + n * LVL_FLOATS;
        gX = actl + TR * ZS; gY = gX + TR * SWS;
        gVX = gY + TR * SWS; gVY = gVX + TR * VWS; Vh = gVY + TR * VWS; Vu = Vh + TR * VWS; gVh = Vu + TR * VWS;
        ggate = gVh + TR * VWS;
    }
    __device__ __forceinline__ float* Sin(const int l) const { return base + l * LVL_FLOATS; }
    __device__ __forceinline__ float* Vin(const int l) const { return Sin(l) + TR * SWS; }
    __device__ __forceinline__ float* Z(const int l) const { return Vin(l) + TR * VWS; }
    __device__ __forceinline__ float* gate(const int l) const { return Z(l) + TR * ZS; }
};

// Vh = Wh^T V and Vu = Wu^T Vh of one GVP for the 16 rows (gvp.py:97-101); sh goes to Sin[:, si:] when want_sh
// (FX 1: the GVP is the update / head shape -- 16 vector channels in, hidden and out, 128 scalars in and out; FX 2: the noise head's last
// GVP -- one vector channel out, 64 scalars out: the dimensions are
// compile-time constants in every product below, whose tile loops, index clamps and bounds predicates then fold away.  These
// kernels are bound by the vector ALU instructions AROUND their matrix instructions, not by the products.)
template <int FX = 0>
__device__ __forceinline__ void gvp_vec(const GvpT& g, const float* Wh, const float* Wu, float* Sin, const float* Vin, float* Vh, float* Vu,
                                        const bool want_sh, const int tid, const int lane, const int wv) {
    const int KH = FX ? 16 : g.h, VI = FX ? 16 : g.vi, VO = FX == 1 ? 16 : (FX == 2 ? 1 : g.vo), SIv = FX ? 128 : g.si;
    mm16<5>(KH, 3 * TR, VI,
         [&](int i, int k) { return Wh[k * KH + i]; },
         [&](int k, int j) { return Vin[(j & 15) * VWS + k * 3 + (j >> 4)]; },
         [&](int i, int j, float x) { Vh[(j & 15) * VWS + i * 3 + (j >> 4)] = x; }, lane, wv);
    __syncthreads();
    if (want_sh)
        for (int idx = tid; idx < TR * KH; idx += NT) {
            const int row = idx & 15, hh = idx >> 4;
            const float* q = Vh + row * VWS + hh * 3;
            Sin[row * SWS + SIv + hh] = t_sqrt(fmaxf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2], 1e-8f));
        }
    mm16<5>(VO, 3 * TR, KH,
         [&](int i, int k) { return Wu[k * VO + i]; },
         [&](int k, int j) { return Vh[(j & 15) * VWS + k * 3 + (j >> 4)]; },
         [&](int i, int j, float x) { Vu[(j & 15) * VWS + i * 3 + (j >> 4)] = x; }, lane, wv);
    __syncthreads();
}

// forward of one GVP on the tile: reads Sin[:, :si], Vin; fills Sin[:, si:], Z, gate, act (= SiLU(Z)) and, when
// Vout != nullptr, the gated vectors (gvp.py:89-116)
struct PackPtr { const float* f; const float* b; };     // k_pack_gvp tables of every GVP: forward and input-gradient fragments of to_feats_out
__device__ __forceinline__ void gvp_fwd(const GvpT& g, const float* W, const PackPtr pk, float* Sin, const float* Vin, float* Z, float* gate,
                                        float* act, const int act_stride, float* Vout, float* Vh, float* Vu,
                                        const int tid, const int lane, const int wv) {
    const int VO = g.vo, SO = g.so, KM = g.si + g.h;
    PFT_STAMP(1);
    gvp_vec(g, W + g.o_Wh, W + g.o_Wu, Sin, Vin, Vh, Vu, true, tid, lane, wv);
    PFT_STAMP(2);
    mm16_packed<11>(reinterpret_cast<const f32x4*>(pk.f) + (size_t)g.pk * (8 * 11 * 64), (SO + 15) >> 4, (KM + 15) >> 4,
         [&](int k, int j) { return Sin[j * SWS + min(k, KM - 1)]; },         // (the fragments are zero beyond KM)
         [&](int i, int j, float x) {
             if (i < SO) {
                 const float z = x + W[g.o_bm + i];
                 Z[j * ZS + i] = z;
                 act[j * act_stride + i] = t_silu(z);
             }
         }, lane, wv);
    __syncthreads();
    PFT_STAMP(3);
    mm16<16>(VO, TR, SO,
         [&](int i, int k) { return W[g.o_Wg + i * SO + k]; },
         [&](int k, int j) { return act[j * act_stride + k]; },
         [&](int i, int j, float x) { gate[j * GTS + i] = x + W[g.o_bg + i]; }, lane, wv);
    __syncthreads();
    PFT_STAMP(4);
    if (Vout != nullptr) {
        for (int idx = tid; idx < TR * VO; idx += NT) {
            const int row = idx & 15, u = idx >> 4;
            const float gt = gate[row * GTS + u];
            const float f = g.sig ? t_sigmoid(gt) : gt;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) Vout[row * VWS + u * 3 + cc] = f * Vu[row * VWS + u * 3 + cc];
        }
        __syncthreads();
    }
}

// backward of one GVP on the tile.  In: gA = dL/d act [row][so] (stride SWS), gVo = dL/d Vout [row][vo*3].
// Out: gS = dL/d Sin[:, :si+h] (the first si entries are the input-scalar gradient), gVi = dL/d Vin.  gA and gVo are
// overwritten (they become dL/dZ and dL/dVu).  Weight gradients are accumulated into gp (this block's copy).
template <bool BF16, int FX = 0>
__device__ __forceinline__ void gvp_bwd(const GvpT& g, const float* W, const PackPtr pk, const bool fresh, float* gp, const float* Sin, const float* Vin,
                                        const float* Z, const float* gate, const float* act, const int act_stride,
                                        float* gA, float* gS, float* gVo, float* gVi, float* Vh, float* Vu, float* gVh,
                                        float* ggate, const int tid_, const int lane_, const int wv, const bool fill_sh = false,
                                        float* wst = nullptr) {
    // (an opaque copy of the thread id per call: with compile-time shapes every phase's per-lane addresses are loop invariants of the
    // callers' level and unit loops, and hoisted there they do not fit the register file -- as in k_bwd_edge_level, E2_PHASE)
    int zz_ = 0;
    asm volatile("" : "+v"(zz_));
    const int tid = tid_ + zz_, lane = tid & 63;
    (void)lane_;
    const int KH = FX ? 16 : g.h, VI = FX ? 16 : g.vi, VO = FX == 1 ? 16 : (FX == 2 ? 1 : g.vo), SI = FX ? 128 : g.si;
    const int SO = FX == 1 ? 128 : (FX == 2 ? 64 : g.so), KM = SI + KH;
    PFT_STAMP(10);
    // the bias gradients' old values (a copy that is not fresh) are requested here and added where the sums are known: read
    // there, the round trip sat between the sum and the barrier every wave of the block waits at
    const float old_bg = (!fresh && tid < VO) ? gp[g.o_bg + tid] : 0.f;
    const float old_bm = (!fresh && tid < SO) ? gp[g.o_bm + tid] : 0.f;
    // wst (PFT_WST_FLOATS of LDS, optional): the GVP's small matrices -- Wh, Wu and the gate weights, 12 KB -- are copied there
    // once per level; five of this function's products read them element by element, and from global memory each such product
    // starts with a dependent L2 round trip on strided addresses
    const float *Wh = W + g.o_Wh, *Wu = W + g.o_Wu, *Wg = W + g.o_Wg;
    if (wst != nullptr) {
        const int nh = VI * KH, nu = KH * VO, ng = VO * SO;
        for (int i = tid; i < nh + nu + ng; i += NT)
            wst[i] = W[i < nh ? g.o_Wh + i : (i < nh + nu ? g.o_Wu + (i - nh) : g.o_Wg + (i - nh - nu))];
        __syncthreads();
        Wh = wst; Wu = wst + nh; Wg = wst + nh + nu;
    }
    // (fill_sh: the level's rows came from the forward's saved pre-activations, the sh columns of Sin are still to be filled)
    gvp_vec<FX>(g, Wh, Wu, const_cast<float*>(Sin), Vin, Vh, Vu, fill_sh, tid, lane, wv);
    PFT_STAMP(11);
    // gate: V' = f(gate) Vu
    for (int idx = tid; idx < TR * VO; idx += NT) {
        const int row = idx & 15, u = idx >> 4;
        const float gt = gate[row * GTS + u];
        float* go = gVo + row * VWS + u * 3;
        const float* vu = Vu + row * VWS + u * 3;
        const float dot = go[0] * vu[0] + go[1] * vu[1] + go[2] * vu[2];
        float f, df;
        if (g.sig) { f = t_sigmoid(gt); df = f * (1.0f - f); } else { f = gt; df = 1.0f; }
        ggate[row * GTS + u] = dot * df;
        go[0] *= f; go[1] *= f; go[2] *= f;
    }
    __syncthreads();
    PFT_STAMP(12);
    mm16<4>(SO, TR, VO,
         [&](int i, int k) { return Wg[k * SO + i]; },
         [&](int k, int j) { return ggate[j * GTS + k]; },
         [&](int i, int j, float x) { gA[j * SWS + i] += x; }, lane, wv);
    mm16_acc<4>(VO, SO, TR,
         [&](int i, int k) { return ggate[k * GTS + i]; },
         [&](int k, int j) { return act[k * act_stride + j]; }, gp + g.o_Wg, SO, lane, wv, fresh);
    if (tid < VO) {
        float s = 0.f;
        for (int r = 0; r < TR; ++r) s += ggate[r * GTS + tid];
        gp[g.o_bg + tid] = old_bg + s;
    }
    __syncthreads();
    PFT_STAMP(13);
    for (int idx = tid; idx < TR * SO; idx += NT) {
        const int row = idx & 15, o = idx >> 4;
        const float z = Z[row * ZS + o];
        const float s = t_sigmoid(z);
        gA[row * SWS + o] *= s * (1.0f + z * (1.0f - s));
    }
    __syncthreads();
    PFT_STAMP(14);
    if constexpr (BF16)
        mm16_packed_bf<8>(reinterpret_cast<const f32x4*>(pk.b) + (size_t)g.pk * (11 * 8 * 64), (KM + 15) >> 4, (SO + 15) >> 4,
             [&](int k, int j) { return gA[j * SWS + k]; },
             [&](int i, int j, float x) { if (i < KM) gS[j * SWS + i] = x; }, lane, wv);
    else
        mm16_packed<8>(reinterpret_cast<const f32x4*>(pk.b) + (size_t)g.pk * (11 * 8 * 64), (KM + 15) >> 4, (SO + 15) >> 4,
             [&](int k, int j) { return gA[j * SWS + k]; },
             [&](int i, int j, float x) { if (i < KM) gS[j * SWS + i] = x; }, lane, wv);
    PFT_STAMP(15);
    mm16_acc_rows<BF16>(SO, KM,
         [&](int i, int k) { return gA[k * SWS + i]; },
         [&](int k, int j) { return Sin[k * SWS + j]; }, gp + g.o_Wm, KM, lane, wv, fresh);
    if (tid < SO) {
        float s = 0.f;
        for (int r = 0; r < TR; ++r) s += gA[r * SWS + tid];
        gp[g.o_bm + tid] = old_bm + s;
    }
    __syncthreads();
    PFT_STAMP(16);
    mm16<4>(KH, 3 * TR, VO,
         [&](int i, int k) { return Wu[i * VO + k]; },
         [&](int k, int j) { return gVo[(j & 15) * VWS + k * 3 + (j >> 4)]; },
         [&](int i, int j, float x) {
             const int row = j & 15, cc = j >> 4;
             const float* q = Vh + row * VWS + i * 3;
             const float ss = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
             const float extra = ss > 1e-8f ? gS[row * SWS + SI + i] * q[cc] / Sin[row * SWS + SI + i] : 0.f;
             gVh[row * VWS + i * 3 + cc] = x + extra;
         }, lane, wv);
    mm16_acc<4>(KH, VO, 3 * TR,
         [&](int i, int k) { return Vh[(k & 15) * VWS + i * 3 + (k >> 4)]; },
         [&](int k, int j) { return gVo[(k & 15) * VWS + j * 3 + (k >> 4)]; }, gp + g.o_Wu, VO, lane, wv, fresh);
    __syncthreads();
    PFT_STAMP(17);
    mm16<5>(VI, 3 * TR, KH,
         [&](int i, int k) { return Wh[i * KH + k]; },
         [&](int k, int j) { return gVh[(j & 15) * VWS + k * 3 + (j >> 4)]; },
         [&](int i, int j, float x) { gVi[(j & 15) * VWS + i * 3 + (j >> 4)] = x; }, lane, wv);
    mm16_acc<4>(VI, KH, 3 * TR,
         [&](int i, int k) { return Vin[(k & 15) * VWS + i * 3 + (k >> 4)]; },
         [&](int k, int j) { return gVh[(k & 15) * VWS + j * 3 + (k >> 4)]; }, gp + g.o_Wh, KH, lane, wv, fresh);
    __syncthreads();
}

// forward recompute of a chain whose first-level inputs (Sin(0)[:, :si], Vin(0)) are in place.  The last level's gated
// vectors go to vout_last when it is not null.
__device__ __forceinline__ void chain_fwd(const ChainLds& L, const GvpT* g, const float* W, const PackPtr pk, float* vout_last,
                                          const int tid, const int lane, const int wv) {
    for (int l = 0; l < L.nlv; ++l) {
        const bool last = l == L.nlv - 1;
        // (a COPY of the level's table entry: through the reference into global memory every use of a field inside the products'
        // store callbacks -- W[g.o_bg + i] -- was a load of the field, a wait, the load of the weight, a wait, per element)
        const GvpT gl = g[l];
        gvp_fwd(gl, W, pk, L.Sin(l), L.Vin(l), L.Z(l), L.gate(l), last ? L.actl : L.Sin(l + 1), last ? ZS : SWS,
                last ? vout_last : L.Vin(l + 1), L.Vh, L.Vu, tid, lane, wv);
    }
}
// instead of chain_fwd: the forward left every level's pre-activations (rows i0 .. i0 + nv - 1 of [level][stride][128 / 16 / 48]):
// fill Z, gate and the next level's inputs (act = SiLU(Z), gated vectors) with independent loads -- one round trip instead of the
// ~30 k cycles per GVP of the recomputation.  The sh columns of every Sin are filled by gvp_bwd (fill_sh).
// (two halves, so that a caller can put its own loads between the request and the first use)
template <int MAXL>
struct ChainRows { float4 zq[MAXL], gq[MAXL], vq[MAXL]; };
template <int MAXL>
__device__ __forceinline__ void chain_request(const ChainLds& L, const float* sv_z, const float* sv_g, const float* sv_v, const size_t stride,
                                              const size_t i0, const int nv, const int tid, const int* rows, ChainRows<MAXL>& R) {
    // every level's rows are requested before any is consumed: one thread = one 16-byte piece of a Z row (16 rows x 32
    // pieces = the block), the first 64 threads a piece of a gate row, the first 192 a piece of a vector row
    const int zr = tid >> 5, zk = (tid & 31) * 4;
    const int gr = (tid >> 2) & 15, gk = (tid & 3) * 4;
    const int vr = min(tid / 12, 15), vk = (tid - (tid / 12) * 12) * 4;
#pragma unroll
    for (int l = 0; l < MAXL; ++l) {
        R.zq[l] = R.gq[l] = R.vq[l] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < L.nlv) {
            const size_t r0 = (size_t)l * stride + i0;
            const size_t iz = rows ? (size_t)rows[min(zr, nv - 1)] : (size_t)min(zr, nv - 1);
            const size_t ig = rows ? (size_t)rows[min(gr, nv - 1)] : (size_t)min(gr, nv - 1);
            const size_t iv = rows ? (size_t)rows[min(vr, nv - 1)] : (size_t)min(vr, nv - 1);
            R.zq[l] = *reinterpret_cast<const float4*>(sv_z + (r0 + iz) * PF_S + zk);
            if (tid < 64) R.gq[l] = *reinterpret_cast<const float4*>(sv_g + (r0 + ig) * 16 + gk);
            if (tid < 192) R.vq[l] = *reinterpret_cast<const float4*>(sv_v + (r0 + iv) * 48 + vk);
        }
    }
}
template <int MAXL>
__device__ __forceinline__ void chain_commit(const ChainLds& L, const GvpT* g, const ChainRows<MAXL>& R, float* vout_last, const int tid) {
    const int zr = tid >> 5, zk = (tid & 31) * 4;
    const int gr = (tid >> 2) & 15, gk = (tid & 3) * 4;
    const int vr = min(tid / 12, 15), vk = (tid - (tid / 12) * 12) * 4;
#pragma unroll
    for (int l = 0; l < MAXL; ++l)
        if (l < L.nlv) {
            const bool last = l == L.nlv - 1;
            const int so = g[l].so, vo = g[l].vo;
            float* Zl = L.Z(l); float* act = last ? L.actl : L.Sin(l + 1);
            float* gt = L.gate(l);
            float* vout = last ? vout_last : L.Vin(l + 1);
            const int astr = last ? ZS : SWS;
            const float zv[4] = {R.zq[l].x, R.zq[l].y, R.zq[l].z, R.zq[l].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (zk + q < so) { Zl[zr * ZS + zk + q] = zv[q]; act[zr * astr + zk + q] = t_silu(zv[q]); }
            if (tid < 64) {
                const float gv[4] = {R.gq[l].x, R.gq[l].y, R.gq[l].z, R.gq[l].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) if (gk + q < vo) gt[gr * GTS + gk + q] = gv[q];
            }
            if (vout != nullptr && tid < 192) {
                const float vv[4] = {R.vq[l].x, R.vq[l].y, R.vq[l].z, R.vq[l].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) if (vk + q < vo * 3) vout[vr * VWS + vk + q] = vv[q];
            }
        }
}
__device__ __forceinline__ void chain_load(const ChainLds& L, const GvpT* g, const float* sv_z, const float* sv_g, const float* sv_v,
                                           const size_t stride, const size_t i0, const int nv, float* vout_last, const int tid,
                                           const int* rows = nullptr) {     // rows (LDS, 16 entries): row indices instead of i0 + row
    ChainRows<PFT_MAX_CHAIN> R;
    chain_request<PFT_MAX_CHAIN>(L, sv_z, sv_g, sv_v, stride, i0, nv, tid, rows, R);
    chain_commit<PFT_MAX_CHAIN>(L, g, R, vout_last, tid);
    __syncthreads();
}
// backward through the chain: upstream gradients in L.gX ([row][so_last], stride SWS) and L.gVX; returns through
// gs_out / gv_out the buffers that hold dL/d Sin(0) and dL/d Vin(0)
template <bool BF16, bool HEAD = false>
__device__ __forceinline__ void chain_bwd(const ChainLds& L, const GvpT* g, const float* W, const PackPtr pk, const bool fresh, float* gp,
                                          float*& gs_out, float*& gv_out, const int tid, const int lane, const int wv, const bool fill_sh = false,
                                          float* wst = nullptr) {
    float *ga = L.gX, *gs = L.gY, *gvo = L.gVX, *gvi = L.gVY;
    for (int l = L.nlv - 1; l >= 0; --l) {
        const bool last = l == L.nlv - 1;
        const GvpT gl = g[l];                           // (a copy: see chain_fwd)
        const bool fx = gl.vi == 16 && gl.h == 16 && gl.vo == 16 && gl.si == 128 && gl.so == 128 && gl.sig;      // block-uniform
        bool fx2 = false;
        if constexpr (HEAD) fx2 = gl.vi == 16 && gl.h == 16 && gl.vo == 1 && gl.si == 128 && gl.so == 64;      // the noise head's last GVP
        if (fx)
            gvp_bwd<BF16, 1>(gl, W, pk, fresh, gp, L.Sin(l), L.Vin(l), L.Z(l), L.gate(l), last ? L.actl : L.Sin(l + 1), last ? ZS : SWS,
                    ga, gs, gvo, gvi, L.Vh, L.Vu, L.gVh, L.ggate, tid, lane, wv, fill_sh, wst);
        else if (HEAD && fx2)
            gvp_bwd<BF16, HEAD ? 2 : 0>(gl, W, pk, fresh, gp, L.Sin(l), L.Vin(l), L.Z(l), L.gate(l), last ? L.actl : L.Sin(l + 1), last ? ZS : SWS,
                    ga, gs, gvo, gvi, L.Vh, L.Vu, L.gVh, L.ggate, tid, lane, wv, fill_sh, wst);
        else
            gvp_bwd<BF16, 0>(gl, W, pk, fresh, gp, L.Sin(l), L.Vin(l), L.Z(l), L.gate(l), last ? L.actl : L.Sin(l + 1), last ? ZS : SWS,
                    ga, gs, gvo, gvi, L.Vh, L.Vu, L.gVh, L.ggate, tid, lane, wv, fill_sh, wst);
        float* t0 = ga; ga = gs; gs = t0;
        float* t1 = gvo; gvo = gvi; gvi = t1;
    }
    gs_out = ga; gv_out = gvo;
}

// per-row mean over 128 features of f(row, feature); red: [16 parts][16 rows] LDS scratch.  All threads get the result
// of their row (row = tid & 15).
template <typename F>
__device__ __forceinline__ float row_mean128(F f, float* red, const int tid) {
    const int row = tid & 15, part = tid >> 4;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < FPP; ++q) s += f(row, part * FPP + q);
    __syncthreads();
    red[part * 16 + row] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < NT / 16; ++q) tot += red[q * 16 + row];
    return tot * (1.0f / 128.0f);
}

// ---------------------------------------------------------------------------------------------
// noise head backward (dynamics_gvp.py:37-42)
// ---------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(NT, 1) void k_bwd_head(const BwdHeadParams p) {
    __shared__ float lds[CHAIN_FLOATS(PFT_MAX_CHAIN)];
    __shared__ float s_ge[TR * 8];
    __shared__ float s_wst[PFT_WST_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    ChainLds L; L.init(lds, p.n_gvps);
    const float* W = p.c.W;
    float* gp = p.c.gpart + (size_t)blockIdx.x * p.c.gstride;
    zero_class(p.c, gp, PFT_CLS_HEAD, threadIdx.x);
    const PackPtr pk = {p.c.wpack_f, p.c.wpack_b};
    const int NF = p.pharm_nf;
    const int SOL = p.g[p.n_gvps - 1].so;            // 64
    // work unit = one 16-row half of a 32-row tile, dealt over the blocks
    for (int unit = blockIdx.x; unit < 2 * p.ntiles; unit += gridDim.x) {
        const NodeTile t = p.tiles[unit >> 1];
        for (int sub = unit & 1; sub == (unit & 1) && sub * TR < t.n; sub += 2) {
            const int nv = min(TR, t.n - sub * TR);
            const int n0 = t.n0 + sub * TR;
            PFT_STAMP(40);
            const bool saved = p.sv_z != nullptr;
            ChainRows<PFT_MAX_CHAIN> CR;         // the saved levels leave with the unit's input rows (one round trip for both)
            if (saved) chain_request<PFT_MAX_CHAIN>(L, p.sv_z, p.sv_g, p.sv_v, p.sv_stride, (size_t)(n0 - p.node_base), nv, tid, nullptr, CR);
            float* S0 = L.Sin(0); float* V0 = L.Vin(0);
            for (int idx = tid; idx < TR * 32; idx += NT) {
                const int row = idx >> 5, q = idx & 31;
                const int n = n0 + min(row, nv - 1);
                const float4 x = reinterpret_cast<const float4*>(p.h + (size_t)n * PF_S)[q];
                float* d = S0 + row * SWS + 4 * q;
                d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
            }
            for (int idx = tid; idx < TR * 12; idx += NT) {
                const int row = idx / 12, q = idx - row * 12;
                const int n = n0 + min(row, nv - 1);
                const float4 x = reinterpret_cast<const float4*>(p.v + (size_t)n * 48)[q];
                float* d = V0 + row * VWS + 4 * q;
                d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
            }
            for (int idx = tid; idx < TR * NF; idx += NT) {
                const int row = idx / NF, o = idx - row * NF;
                s_ge[row * 8 + o] = row < nv ? p.g_eps_h[(size_t)(n0 - p.node_base + row) * NF + o] : 0.f;
            }
            if (saved) chain_commit<PFT_MAX_CHAIN>(L, p.g, CR, nullptr, tid);
            __syncthreads();
            PFT_STAMP(41);
            if (!saved) chain_fwd(L, p.g, W, pk, nullptr, tid, lane, wv);
            PFT_STAMP(42);
            // to_scalar_output: eps_h = Wout act + b ; eps_x = the single output vector channel
            const bool fresh_u = unit == (int)blockIdx.x;           // this block's first unit: its copy holds zeros (zero_class)
            for (int idx = tid; idx < TR * SOL; idx += NT) {
                const int row = idx & 15, k = idx >> 4;
                float wv8[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) wv8[o] = W[p.o_Wout + min(o, NF - 1) * SOL + k];     // (pharm_nf <= 8: one round trip)
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < 8; ++o) if (o < NF) s += wv8[o] * s_ge[row * 8 + o];
                L.gX[row * SWS + k] = s;
            }
            for (int idx = tid; idx < NF * SOL; idx += NT) {
                const int o = idx / SOL, k = idx - o * SOL;
                const float old = fresh_u ? 0.f : gp[p.o_Wout + idx];
                float s = 0.f;
                for (int r = 0; r < TR; ++r) s += s_ge[r * 8 + o] * L.actl[r * ZS + k];
                gp[p.o_Wout + idx] = old + s;
            }
            if (tid < NF) {
                const float old = fresh_u ? 0.f : gp[p.o_bout + tid];
                float s = 0.f;
                for (int r = 0; r < TR; ++r) s += s_ge[r * 8 + tid];
                gp[p.o_bout + tid] = old + s;
            }
            for (int idx = tid; idx < TR * 3; idx += NT) {
                const int row = idx / 3, cc = idx - row * 3;
                L.gVX[row * VWS + cc] = row < nv ? p.g_eps_x[(size_t)(n0 - p.node_base + row) * 3 + cc] : 0.f;
            }
            __syncthreads();
            PFT_STAMP(43);
            float *gs, *gv;
            chain_bwd<BF16, true>(L, p.g, W, pk, unit == (int)blockIdx.x, gp, gs, gv, tid, lane, wv, saved, s_wst);
            PFT_STAMP(44);
            for (int idx = tid; idx < TR * 128; idx += NT) {
                const int row = idx >> 7, f = idx & 127;
                if (row < nv) p.G_h[(size_t)(n0 + row) * PF_S + f] = gs[row * SWS + f];
            }
            for (int idx = tid; idx < TR * 48; idx += NT) {
                const int row = idx / 48, q = idx - row * 48;
                if (row < nv) p.G_v[(size_t)(n0 + row) * 48 + q] = gv[row * VWS + q];
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// node update backward (gvp.py:499-536): recompute aggregate -> dropout -> residual -> LN -> update chain -> dropout
// -> residual -> LN, then walk it backwards
// ---------------------------------------------------------------------------------------------
#define NODE_LVLS 3
struct VecLn { float den, sq; };
// GVPLayerNorm vector statistics of one row: den = sqrt(mean_ch max(|v_ch|^2, 1e-8) + 1e-5) + 1e-5 (gvp.py:163-165)
__device__ __forceinline__ VecLn vec_ln_stats(const float* v) {
    float m = 0.f;
    for (int ch = 0; ch < PF_V; ++ch) m += fmaxf(v[ch * 3] * v[ch * 3] + v[ch * 3 + 1] * v[ch * 3 + 1] + v[ch * 3 + 2] * v[ch * 3 + 2], 1e-8f);
    VecLn r;
    r.sq = t_sqrt(m * (1.0f / PF_V) + 1e-5f);
    r.den = r.sq + 1e-5f;
    return r;
}
// gradient of out = v / den w.r.t. v for one row, in place on g (48 floats)
__device__ __forceinline__ void vec_ln_bwd_row(const float* v, const VecLn st, float* g) {
    float D = 0.f;
    for (int q = 0; q < 48; ++q) D += g[q] * v[q];
    const float coef = D / (st.den * st.den) / (PF_V * st.sq);
    const float rden = 1.0f / st.den;
    for (int ch = 0; ch < PF_V; ++ch) {
        const float* q = v + ch * 3;
        const float ind = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) > 1e-8f ? 1.0f : 0.0f;
        for (int cc = 0; cc < 3; ++cc) g[ch * 3 + cc] = g[ch * 3 + cc] * rden - coef * ind * q[cc];
    }
}

// sum over the 128 features of row tid >> 5 of f(row, feature), inside a half wave (lane l of the half takes features l, l + 32,
// l + 64, l + 96): no LDS, no barrier -- a row's scalar-LayerNorm statistics stay within the 32 lanes that also normalise it
template <typename F>
__device__ __forceinline__ float row_sum128_hw(F f, const int tid) {
    const int row = tid >> 5, l = tid & 31;
    float s = (f(row, l) + f(row, l + 32)) + (f(row, l + 64) + f(row, l + 96));
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}
// the vector statistics of all 16 rows at once: thread (row = tid >> 4, channel = tid & 15) of the first 256
__device__ __forceinline__ void vec_ln_stats_par(const float* v, VecLn* out, const int tid) {
    if (tid < 256) {
        const int row = tid >> 4, ch = tid & 15;
        const float* q = v + row * VWS + ch * 3;
        float m = fmaxf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2], 1e-8f);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m += __shfl_xor(m, o);
        if (ch == 0) {
            VecLn r;
            r.sq = t_sqrt(m * (1.0f / PF_V) + 1e-5f);
            r.den = r.sq + 1e-5f;
            out[row] = r;
        }
    }
}
// vec_ln_bwd_row for all 16 rows: thread (row, channel) owns its three entries of g
__device__ __forceinline__ void vec_ln_bwd_par(const float* v, const VecLn* st, float* g, const int tid) {
    if (tid < 256) {
        const int row = tid >> 4, ch = tid & 15;
        const float* q = v + row * VWS + ch * 3;
        float* gq = g + row * VWS + ch * 3;
        float D = gq[0] * q[0] + gq[1] * q[1] + gq[2] * q[2];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) D += __shfl_xor(D, o);
        const VecLn s = st[row];
        const float coef = D / (s.den * s.den) / (PF_V * s.sq);
        const float rden = 1.0f / s.den;
        const float ind = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) > 1e-8f ? 1.0f : 0.0f;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) gq[cc] = gq[cc] * rden - coef * ind * q[cc];
    }
}

template <bool BF16>
__global__ __launch_bounds__(NT, 1) void k_bwd_node(const BwdNodeParams p) {
    __shared__ float lds[CHAIN_FLOATS(NODE_LVLS)];
    __shared__ float xh1[TR * ZS], xh2[TR * ZS], gu[TR * ZS];
    __shared__ float vy[TR * VWS], vz[TR * VWS], gvu[TR * VWS], rvl[TR * VWS];
    __shared__ float s_rstd1[TR], s_rstd2[TR], s_inv[TR];
    __shared__ VecLn s_vl1[TR], s_vl2[TR];
    __shared__ int s_n[TR], s_sv[TR];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    ChainLds L; L.init(lds, p.n_upd);
    const float* W = p.c.W;
    float* gp = p.c.gpart + (size_t)blockIdx.x * p.c.gstride;
    zero_class(p.c, gp, PFT_CLS_NODE + p.layer, threadIdx.x);
    const PackPtr pk = {p.c.wpack_f, p.c.wpack_b};
    bool seen[2] = {false, false};          // weight-gradient tiles of a node type: the first unit stores onto the cleared copy without reading it
    float a_ln[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // threads < 128: LayerNorm weight / bias gradients of feature tid, per node type, over all units
    const uint32_t st_msg = (uint32_t)p.layer * 2u, st_res = st_msg + 1u;
    // the non-empty 16-row units of the launch, dealt round-robin (k_compact_units): dealt by table index, the units of the
    // active-atom tiles (capacity 256 atoms per graph, ~60 in use) left every other block with half as much again to do
    const int rows1 = p.ucnt[1], rows0 = p.ucnt[2];
    const int U1 = (rows1 + TR - 1) / TR, U0 = (rows0 + TR - 1) / TR;
    const int2* const list1 = reinterpret_cast<const int2*>(p.ulist);
    const int2* const list0 = list1 + p.ucap;
    for (int ui = blockIdx.x; ui < U1 + U0; ui += gridDim.x) {
        const int nt = ui < U1 ? 1 : 0;
        const int u0 = (nt ? ui : ui - U1) * TR;
        const int2* const lst = nt ? list1 : list0;
        const GvpT* g = p.upd + nt * p.n_upd;
        const int o_l1w = p.o_ln[nt][0], o_l1b = p.o_ln[nt][1], o_l2w = p.o_ln[nt][2];
        {
            const int nv = min(TR, (nt ? rows1 : rows0) - u0);
            PFT_STAMP(20);
            if (tid < TR) {
                const int2 ent = lst[u0 + min(tid, nv - 1)];
                const int n = ent.x;
                s_n[tid] = n;
                s_sv[tid] = ent.y;
                float inv = 1.0f;
                if (p.norm_mode == 1) inv = 1.0f / p.norm_value;
                else if (p.norm_mode == 2) inv = 1.0f / p.gnorm[nt * p.B + p.gid[n]];
                s_inv[tid] = inv;
            }
            __syncthreads();
            // ---- aggregate the message partial rows (as k_node_update), dropout, residual.  One dependent round trip for what the
            // unit needs of its nodes: the in-edge runs of both slots, the layer-input rows and the upstream gradients leave together;
            // then the first two partial rows of BOTH runs (a run of ~7 slots spans one or two groups) -- taken one after the other,
            // run by run and row by row, these were six round trips.  The sums keep their order (slot 0's rows, then the other slot's).
            {
                const int rowq = tid >> 5, q = tid & 31;
                const float4 gq = reinterpret_cast<const float4*>(p.G_h_out + (size_t)s_n[rowq] * PF_S)[q];
                float4 gvq = make_float4(0.f, 0.f, 0.f, 0.f);
                const int rowv = min(tid / 12, TR - 1), qv = tid - (tid / 12) * 12;
                if (tid < TR * 12) gvq = reinterpret_cast<const float4*>(p.G_v_out + (size_t)s_n[rowv] * 48)[qv];
                if (tid < 256) {
                    const int row = tid >> 4, part = tid & 15;
                    const int n = s_n[row];
                    const int slot1 = nt == 0 ? p.pp_slot : 1;
                    const int stv[2] = {p.in_start[n], p.in_start[slot1 * p.N + n]};
                    const int cv[2] = {p.in_cnt[n], p.in_cnt[slot1 * p.N + n]};
                    const float4 hq0 = reinterpret_cast<const float4*>(p.h_in + (size_t)n * PF_S)[part * 2];
                    const float4 hq1 = reinterpret_cast<const float4*>(p.h_in + (size_t)n * PF_S)[part * 2 + 1];
                    float v0[3] = {0.f, 0.f, 0.f};
                    if (!p.l0) {
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) v0[cc] = p.v_in[(size_t)n * 48 + part * 3 + cc];
                    }
                    int rr[2][2]; bool on[2][2];
                    float4 ra[2][2], rb[2][2]; float rv[2][2][3];
#pragma unroll
                    for (int si = 0; si < 2; ++si) {
                        const int end = stv[si] + cv[si];
                        rr[si][0] = min(stv[si] | (p.grp - 1), end - 1); on[si][0] = cv[si] > 0;
                        rr[si][1] = min((rr[si][0] + 1) | (p.grp - 1), end - 1); on[si][1] = on[si][0] && rr[si][0] + 1 < end;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const size_t r = (size_t)(on[si][k] ? rr[si][k] : p.zero_row);
                            ra[si][k] = reinterpret_cast<const float4*>(p.msg_s + r * PF_S)[part * 2];
                            rb[si][k] = reinterpret_cast<const float4*>(p.msg_s + r * PF_S)[part * 2 + 1];
#pragma unroll
                            for (int cc = 0; cc < 3; ++cc) rv[si][k][cc] = p.msg_v[r * 48 + part * 3 + cc];
                        }
                    }
                    float ms[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mv[3] = {0, 0, 0};
#pragma unroll
                    for (int si = 0; si < 2; ++si) {
                        const int st = stv[si], c = cv[si];
                        const float sc = (p.norm_mode == 0 && c > 0) ? 1.0f / (float)c : 1.0f;
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            if (on[si][k]) {
                                const float4 a = ra[si][k], b = rb[si][k];
                                ms[0] = fmaf(a.x, sc, ms[0]); ms[1] = fmaf(a.y, sc, ms[1]); ms[2] = fmaf(a.z, sc, ms[2]); ms[3] = fmaf(a.w, sc, ms[3]);
                                ms[4] = fmaf(b.x, sc, ms[4]); ms[5] = fmaf(b.y, sc, ms[5]); ms[6] = fmaf(b.z, sc, ms[6]); ms[7] = fmaf(b.w, sc, ms[7]);
                                mv[0] = fmaf(rv[si][k][0], sc, mv[0]); mv[1] = fmaf(rv[si][k][1], sc, mv[1]); mv[2] = fmaf(rv[si][k][2], sc, mv[2]);
                            }
                        int e = on[si][1] ? rr[si][1] + 1 : st + c;             // (a run beyond two groups: the rest one at a time)
                        while (e < st + c) {
                            const int r = min(e | (p.grp - 1), st + c - 1);
                            const float4 a = reinterpret_cast<const float4*>(p.msg_s + (size_t)r * PF_S)[part * 2];
                            const float4 b = reinterpret_cast<const float4*>(p.msg_s + (size_t)r * PF_S)[part * 2 + 1];
                            ms[0] = fmaf(a.x, sc, ms[0]); ms[1] = fmaf(a.y, sc, ms[1]); ms[2] = fmaf(a.z, sc, ms[2]); ms[3] = fmaf(a.w, sc, ms[3]);
                            ms[4] = fmaf(b.x, sc, ms[4]); ms[5] = fmaf(b.y, sc, ms[5]); ms[6] = fmaf(b.z, sc, ms[6]); ms[7] = fmaf(b.w, sc, ms[7]);
                            const float* vr = p.msg_v + (size_t)r * 48 + part * 3;
                            mv[0] = fmaf(vr[0], sc, mv[0]); mv[1] = fmaf(vr[1], sc, mv[1]); mv[2] = fmaf(vr[2], sc, mv[2]);
                            e = r + 1;
                        }
                    }
                    const float inv = s_inv[row];
                    const float hin[8] = {hq0.x, hq0.y, hq0.z, hq0.w, hq1.x, hq1.y, hq1.z, hq1.w};
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) {
                        const float dm = drop_mul(p.c, st_msg, (uint32_t)n * 144u + (uint32_t)(part * 8 + q8));
                        xh1[row * ZS + part * 8 + q8] = hin[q8] + ms[q8] * inv * dm;
                    }
                    const float dmv = drop_mul(p.c, st_msg, (uint32_t)n * 144u + 128u + (uint32_t)part);
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) vy[row * VWS + part * 3 + cc] = v0[cc] + mv[cc] * inv * dmv;
                }
                // the upstream gradients of the unit's rows wait in LDS for the LayerNorm's backward (rows beyond nv: zeros)
                {
                    const bool ok = rowq < nv;
                    float* d = gu + rowq * ZS + 4 * q;
                    d[0] = ok ? gq.x : 0.f; d[1] = ok ? gq.y : 0.f; d[2] = ok ? gq.z : 0.f; d[3] = ok ? gq.w : 0.f;
                    if (tid < TR * 12) {
                        const bool okv = rowv < nv;
                        float* dv = gvu + rowv * VWS + 4 * qv;
                        dv[0] = okv ? gvq.x : 0.f; dv[1] = okv ? gvq.y : 0.f; dv[2] = okv ? gvq.z : 0.f; dv[3] = okv ? gvq.w : 0.f;
                    }
                }
            }
            __syncthreads();
            PFT_STAMP(21);
            // ---- LN1: u -> Sin(0), vu -> Vin(0)
            {
                const int row = tid >> 5, l32 = tid & 31;
                const float mean = row_sum128_hw([&](int r, int f) { return xh1[r * ZS + f]; }, tid) * (1.0f / 128.0f);
                const float var = row_sum128_hw([&](int r, int f) { const float d = xh1[r * ZS + f] - mean; return d * d; }, tid) * (1.0f / 128.0f);
                const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
                if (l32 == 0) s_rstd1[row] = rstd;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = l32 + 32 * q;
                    const float xh = (xh1[row * ZS + f] - mean) * rstd;
                    xh1[row * ZS + f] = xh;
                    L.Sin(0)[row * SWS + f] = xh * W[o_l1w + f] + W[o_l1b + f];
                }
                vec_ln_stats_par(vy, s_vl1, tid);
            }
            __syncthreads();
            for (int idx = tid; idx < TR * 48; idx += NT) {
                const int row = idx & 15, q = idx >> 4;
                L.Vin(0)[row * VWS + q] = vy[row * VWS + q] / s_vl1[row].den;
            }
            __syncthreads();
            PFT_STAMP(22);
            // (requesting the saved levels with the first phase's loads, to save this round trip, measured no gain)
            const bool saved = p.sv_z != nullptr;
            if (!saved) chain_fwd(L, g, W, pk, rvl, tid, lane, wv);
            else chain_load(L, g, p.sv_z, p.sv_g, p.sv_v, p.sv_stride, 0, nv, rvl, tid, s_sv);
            PFT_STAMP(23);
            // ---- residual dropout, residual, LN2 statistics
            for (int idx = tid; idx < TR * 128; idx += NT) {
                const int row = idx & 15, f = idx >> 4;
                const float dm = drop_mul(p.c, st_res, (uint32_t)s_n[row] * 144u + (uint32_t)f);
                xh2[row * ZS + f] = L.Sin(0)[row * SWS + f] + L.actl[row * ZS + f] * dm;
            }
            for (int idx = tid; idx < TR * 48; idx += NT) {
                const int row = idx & 15, q = idx >> 4;
                const float dm = drop_mul(p.c, st_res, (uint32_t)s_n[row] * 144u + 128u + (uint32_t)(q / 3));
                vz[row * VWS + q] = L.Vin(0)[row * VWS + q] + rvl[row * VWS + q] * dm;
            }
            __syncthreads();
            {
                const int row = tid >> 5, l32 = tid & 31;
                const float mean = row_sum128_hw([&](int r, int f) { return xh2[r * ZS + f]; }, tid) * (1.0f / 128.0f);
                const float var = row_sum128_hw([&](int r, int f) { const float d = xh2[r * ZS + f] - mean; return d * d; }, tid) * (1.0f / 128.0f);
                const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
                if (l32 == 0) s_rstd2[row] = rstd;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = l32 + 32 * q;
                    xh2[row * ZS + f] = (xh2[row * ZS + f] - mean) * rstd;
                }
                vec_ln_stats_par(vz, s_vl2, tid);
            }
            __syncthreads();
            PFT_STAMP(24);
            // ---- backward: LN2 (gu / gvu: the upstream gradients, in LDS since the unit's first phase)
            if (tid < 128) {
                float sw = 0.f, sb = 0.f;
                for (int r = 0; r < TR; ++r) { const float go = gu[r * ZS + tid]; sw += go * xh2[r * ZS + tid]; sb += go; }
                if (nt == 0) { a_ln[0][2] += sw; a_ln[0][3] += sb; } else { a_ln[1][2] += sw; a_ln[1][3] += sb; }
            }
            {
                const int row = tid >> 5, l32 = tid & 31;
                const float m1 = row_sum128_hw([&](int r, int f) { return gu[r * ZS + f] * W[o_l2w + f]; }, tid) * (1.0f / 128.0f);
                const float m2 = row_sum128_hw([&](int r, int f) { return gu[r * ZS + f] * W[o_l2w + f] * xh2[r * ZS + f]; }, tid) * (1.0f / 128.0f);
                const float rstd = s_rstd2[row];
                __syncthreads();                                                   // the column sums above have read gu
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = l32 + 32 * q;
                    const float gz = rstd * (gu[row * ZS + f] * W[o_l2w + f] - m1 - xh2[row * ZS + f] * m2);
                    gu[row * ZS + f] = gz;                                     // dL/du (residual path)
                    L.gX[row * SWS + f] = gz * drop_mul(p.c, st_res, (uint32_t)s_n[row] * 144u + (uint32_t)f);
                }
                vec_ln_bwd_par(vz, s_vl2, gvu, tid);
            }
            __syncthreads();
            for (int idx = tid; idx < TR * 48; idx += NT) {
                const int row = idx & 15, q = idx >> 4;
                L.gVX[row * VWS + q] = gvu[row * VWS + q] * drop_mul(p.c, st_res, (uint32_t)s_n[row] * 144u + 128u + (uint32_t)(q / 3));
            }
            __syncthreads();
            PFT_STAMP(25);
            float *gs, *gv;
            chain_bwd<BF16>(L, g, W, pk, !seen[nt], gp, gs, gv, tid, lane, wv, saved);
            PFT_STAMP(26);
            seen[nt] = true;
            // ---- LN1 backward
            for (int idx = tid; idx < TR * 128; idx += NT) {
                const int row = idx & 15, f = idx >> 4;
                gu[row * ZS + f] += gs[row * SWS + f];
            }
            for (int idx = tid; idx < TR * 48; idx += NT) {
                const int row = idx & 15, q = idx >> 4;
                gvu[row * VWS + q] = (gvu[row * VWS + q] + gv[row * VWS + q]) ;
            }
            __syncthreads();
            if (tid < 128) {
                float sw = 0.f, sb = 0.f;
                for (int r = 0; r < TR; ++r) { const float go = gu[r * ZS + tid]; sw += go * xh1[r * ZS + tid]; sb += go; }
                if (nt == 0) { a_ln[0][0] += sw; a_ln[0][1] += sb; } else { a_ln[1][0] += sw; a_ln[1][1] += sb; }
            }
            {
                const int row = tid >> 5, l32 = tid & 31;
                const float m1 = row_sum128_hw([&](int r, int f) { return gu[r * ZS + f] * W[o_l1w + f]; }, tid) * (1.0f / 128.0f);
                const float m2 = row_sum128_hw([&](int r, int f) { return gu[r * ZS + f] * W[o_l1w + f] * xh1[r * ZS + f]; }, tid) * (1.0f / 128.0f);
                const float rstd = s_rstd1[row];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = l32 + 32 * q;
                    gu[row * ZS + f] = rstd * (gu[row * ZS + f] * W[o_l1w + f] - m1 - xh1[row * ZS + f] * m2);   // dL/dy
                }
                // the LN divides vy by den: gvu currently holds dL/d vu
                vec_ln_bwd_par(vy, s_vl1, gvu, tid);
            }
            __syncthreads();
            PFT_STAMP(27);
            for (int idx = tid; idx < TR * 128; idx += NT) {
                const int row = idx >> 7, f = idx & 127;
                if (row < nv) {
                    const int n = s_n[row];
                    const float gy = gu[row * ZS + f];
                    p.G_h_in[(size_t)n * PF_S + f] = gy;
                    p.gagg_s[(size_t)n * PF_S + f] = gy * s_inv[row] * drop_mul(p.c, st_msg, (uint32_t)n * 144u + (uint32_t)f);
                }
            }
            for (int idx = tid; idx < TR * 48; idx += NT) {
                const int row = idx / 48, q = idx - row * 48;
                if (row < nv) {
                    const int n = s_n[row];
                    const float gy = gvu[row * VWS + q];
                    if (!p.l0) p.G_v_in[(size_t)n * 48 + q] = gy;
                    p.gagg_v[(size_t)n * 48 + q] = gy * s_inv[row] * drop_mul(p.c, st_msg, (uint32_t)n * 144u + 128u + (uint32_t)(q / 3));
                }
            }
            __syncthreads();
        }
    }
    // the LayerNorm parameters' gradients of this block (its copy was cleared by zero_class: stored, not added)
    if (tid < 128)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            if (seen[nt])
#pragma unroll
                for (int k = 0; k < 4; ++k) gp[p.o_ln[nt][k] + tid] = a_ln[nt][k];
}

// ---------------------------------------------------------------------------------------------
// edge message backward, one GVP level per launch (see BwdEdgeLevelParams).  A pass is one tile of 32 edge slots.
//
// What a pass costs is decided by operand fetch, not by the matrix pipe (its products are ~1,900
// v_mfma_f32_16x16x4_f32 = 15 k cycles over four SIMDs), so the pass is organised around that:
//   * the rows of the NEXT tile (saved Z / gate / V rows, level inputs, upstream gradients: 62 KB) are fetched into
//     registers while this tile's big products run, their indices one tile earlier still, the tile descriptors once per
//     round of 64 tiles: no dependent global round trip is left inside a pass;
//   * the input-gradient product gS = gZ Wm reads to_feats_out as 1-KiB fragments of a packed copy (k_pack_gvp: lane
//     (i, kq) holds four consecutive k of input i -- one global_load_dwordx4 per four MFMAs) and keeps gZ, its other
//     operand, in registers for all of a wave's output tiles (16 ds_read_b128 per wave and pass);
//   * Wh, Wu and the gate weights sit in LDS for the whole launch; the small vector products are dealt over the waves by
//     COLUMN tile (row half x coordinate: six waves) while the two remaining waves take the Wu / Wh weight gradients;
//   * the gate contribution and SiLU' are applied on the accumulator fragments (one LDS round trip less).
// Weight gradients accumulate in registers over all tiles of the block and are flushed once.
// ---------------------------------------------------------------------------------------------
#define ER 32             // rows per pass
#define E2_ZS 144         // LDS row stride of Z rows [128] (16-byte aligned rows: b128 accesses)
#define E2_SS 176         // ... of scalar rows [si + h] (<= 161): whole 16-column tiles
#define E2_WHS 36         // ... of the staged Wh [vi][h], zero padded to [32][36]
#define E2_TILES 64       // passes per round (their edge slots are staged in LDS together)
// Every phase of a pass derives its lane coordinates from an OPAQUE copy of the thread id: otherwise the compiler hoists the
// per-thread addresses of all ten phases out of the pass loop (~100 registers live across it, spilled in the prologue and
// re-read from scratch at every use, and no room left for the rows fetched ahead)
#define E2_PHASE() int zz_ = 0; asm volatile("" : "+v"(zz_)); const int tid = tid_ + zz_; const int lane = tid & 63; \
    const int li = lane & 15, kq = lane >> 4; (void)li; (void)kq


// (a dynamic region's tiles cover its capacity and are partly empty: the backward kernels work on dense lists of what is valid)
// the valid rows of a node tile table, densely and in table order, per node type: list[s * cap + i] = (node id, row in the saved
// update-chain levels), s = 0 pharm, 1 prot; ucnt[1 + s] = rows.  A backward unit is 16 consecutive entries of one type: the
// per-graph active-atom tiles of the pruned layer hold 20-40 of 256 slots, and a unit costs the same whatever it holds.
__global__ __launch_bounds__(1024) void k_compact_node_rows(const NodeTile* tiles, const int ntiles, const int* dyn_cnt, const int* row_ids,
                                                            const int N, int2* list, const int cap, int* ucnt) {
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int type = blockIdx.x == 0 ? 1 : 0;
    int2* out = list + (size_t)blockIdx.x * cap;
    int base = 0;
    for (int c = 0; c < ntiles; c += 1024) {
        const int ti = c + tid;
        int n = 0;
        NodeTile t{};
        if (ti < ntiles) {
            t = tiles[ti];
            n = t.n;
            if (t.cnt_idx >= 0) n = min(n, max(dyn_cnt[t.cnt_idx] - t.rel, 0));
            if (t.ntype != type) n = 0;
        }
        int incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(incl, o); if (lane >= o) incl += x; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int x = s_w[w]; if (w < wv) off += x; tot += x; }
        off += incl - n;
        for (int r = 0; r < n; ++r) {
            const int pos = t.n0 + r;
            out[off + r] = make_int2(t.ids ? row_ids[pos] : pos, (t.ids ? N : 0) + pos);
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) ucnt[1 + blockIdx.x] = base;
}
struct CompactParams { int et_tile0[5]; };
// the valid edge slots of each etype's segment of a tile table, densely and in table order: rlist[32 (et_tile0[et] - et_tile0[0]) + i],
// i < rows; ccnt[et] = passes of 32 rows, ccnt[8 + et] = rows.  A backward pass then takes 32 consecutive entries whatever
// regions they come from: the per-(graph, etype) regions of the dynamic etypes fill their 32-slot tiles to ~65 % (20-56 edges
// per region), and a pass costs the same whatever it holds.
__global__ __launch_bounds__(1024) void k_compact_rows(const EdgeTile* tiles, const CompactParams cp, const int* dyn_cnt, int* rlist, int* ccnt) {
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int t0 = cp.et_tile0[blockIdx.x], t1 = cp.et_tile0[blockIdx.x + 1];     // one block per etype
    int* out = rlist + (size_t)(t0 - cp.et_tile0[0]) * 32;
    int base = 0;
    for (int c = t0; c < t1; c += 1024) {
        const int ti = c + tid;
        int n = 0, e0 = 0;
        if (ti < t1) {
            const EdgeTile t = tiles[ti];
            n = t.n; e0 = t.e0;
            if (t.cnt_idx >= 0) n = min(n, max(dyn_cnt[t.cnt_idx] - t.rel, 0));
        }
        int incl = n;                                   // inclusive scan over the wave, then over the 16 waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(incl, o); if (lane >= o) incl += x; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int x = s_w[w]; if (w < wv) off += x; tot += x; }
        off += incl - n;
        for (int r = 0; r < n; ++r) out[off + r] = e0 + r;
        base += tot;
        __syncthreads();
    }
    if (tid == 0) { ccnt[blockIdx.x] = (base + 31) >> 5; ccnt[8 + blockIdx.x] = base; }
}

// packed to_feats_out of every GVP (message, update and head GVPs in the order of the GvpT table; GvpT::pk):
//   input-gradient product (blockIdx.y == 0): [gvp][m tile (11)][k block (8)][lane] x 4 -- lane (li, kq) of m tile mt, block sb
//     holds W[k = 16 sb + 4 kq + t][i = 16 mt + li], t = 0..3 (zero for i >= si + h, k >= so)
//   forward product (blockIdx.y == 1): [gvp][m tile (8)][k block (11)][lane] x 4 -- W[o = 16 mt + li][k = 16 sb + 4 kq + t]
// (blocks [npack, ...) of row 0: the loss's unit gradients times their upstream scalars -- k_scale_loss's work, which used to be a
// launch of its own right in front of this one)
__global__ __launch_bounds__(64) void k_pack_gvp(const float* W, const GvpT* g, float* out_b, float* out_f, const int npack, const ScaleArgs sa) {
    if ((int)blockIdx.x >= npack) {
        if (blockIdx.y != 0) return;
        const int i = ((int)blockIdx.x - npack) * 64 + (int)threadIdx.x;
        if (i < sa.nx) sa.gx[i] *= sa.a[0] + (sa.a2 ? sa.a2[0] : 0.f);
        else if (i - sa.nx < sa.nh) sa.gh[i - sa.nx] *= sa.b[0] + (sa.b2 ? sa.b2[0] : 0.f);
        return;
    }
    const int gi = blockIdx.x / 88, rem = blockIdx.x - gi * 88;
    const int lane = threadIdx.x, li = lane & 15, kq = lane >> 4;
    const GvpT t = g[gi];
    const int KM = t.si + t.h;
    f32x4 v;
    if (blockIdx.y == 0) {
        const int mt = rem >> 3, sb = rem & 7, i = 16 * mt + li;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int k = 16 * sb + 4 * kq + tt;
            v[tt] = (i < KM && k < t.so) ? W[t.o_Wm + k * KM + i] : 0.f;
        }
        reinterpret_cast<f32x4*>(out_b)[(size_t)blockIdx.x * 64 + lane] = v;
    } else {
        const int mt = rem / 11, sb = rem - mt * 11, o = 16 * mt + li;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int k = 16 * sb + 4 * kq + tt;
            v[tt] = (o < t.so && k < KM) ? W[t.o_Wm + o * KM + k] : 0.f;
        }
        reinterpret_cast<f32x4*>(out_f)[(size_t)blockIdx.x * 64 + lane] = v;
    }
}

// one column tile (16 columns, column = lane & 15) of C[M x 16] = A[M x K] B[K x 16], K <= 4 KS: b(k) is this lane's
// column of B, a(i, k) an element of A (zero padded), c(i, value) consumes row i of the lane's column
template <int KS, typename FA, typename FB, typename FC>
__device__ __forceinline__ void mmcol(const int mts, FA a, FB b, FC c, const int lane) {
    const int li = lane & 15, kq = lane >> 4;
    float bv[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) bv[u] = b(4 * u + kq);
    for (int mt = 0; mt < mts; ++mt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KS; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a(mt * 16 + li, 4 * u + kq), bv[u], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) c(mt * 16 + kq * 4 + r, acc[r]);
    }
}

// FX: the level's GVP shape as compile-time constants (1: message GVPs above the first -- 16 vector channels in / hidden / out, 128
// scalars in; 2: the first message GVP -- 17 channels in / hidden, 144 scalars in; 0: read from the table), BwdEdgeLevelParams::fx
template <bool BF16, int FX>
__global__ __launch_bounds__(NT, 1) void k_bwd_edge_level(const BwdEdgeLevelParams p) {
    __shared__ __attribute__((aligned(16))) float Zb[ER * E2_ZS];
    __shared__ __attribute__((aligned(16))) float Sin[ER * E2_SS];
    __shared__ __attribute__((aligned(16))) float gA[ER * E2_SS];
    __shared__ __attribute__((aligned(16))) float gS[ER * E2_SS];
    __shared__ float gate[ER * GTS], ggate[ER * GTS];
    __shared__ float Vin[ER * VWS], Vh[ER * VWS], Vu[ER * VWS], gVo[ER * VWS], gVh[ER * VWS], gVi[ER * VWS];
    __shared__ float sWh[32 * E2_WHS], sWu[32 * 16], sWg[16 * 128];
    __shared__ int s_src[ER], s_e[ER], s_tnv[E2_TILES];
    __shared__ int s_slot[E2_TILES * ER];            // edge slots of this round's passes (dense row list, k_compact_rows)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int tid_ = tid;
    // blocks per etype in proportion to its passes (k_compact_rows), at least one where there are rows: every
    // block derives the same partition from the four counts
    int et = -1, nb = 0, my = 0, cnt_et = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int b0, nbk;
        et_blocks(p.ccnt, p.n_et, (int)gridDim.x, e, b0, nbk);
        if ((int)blockIdx.x >= b0 && (int)blockIdx.x < b0 + nbk) { et = e; nb = nbk; my = blockIdx.x - b0; cnt_et = p.ccnt[e]; }
    }
    if (et < 0) return;                              // block-uniform: more gradient copies than work
    const int* rlist = p.clist + (size_t)(p.et_tile0[et] - p.et_tile0[0]) * 32;
    const int nrows = p.ccnt[8 + et];
    const GvpT g = p.g[et * p.n_gvps + p.level];
    const float* W = p.c.W;
    const f32x4* Wp = reinterpret_cast<const f32x4*>(p.wpack) + (size_t)(et * p.n_gvps + p.level) * (11 * 8 * 64);
    float* gp = p.c.gpart + (size_t)blockIdx.x * p.c.gstride;
    const int KH = FX == 1 ? 16 : (FX == 2 ? 17 : g.h), VI = FX == 1 ? 16 : (FX == 2 ? 17 : g.vi), VO = FX ? 16 : g.vo;
    const int SI = FX == 1 ? 128 : (FX == 2 ? 144 : g.si), SO = FX ? 128 : g.so, KM = SI + KH;   // SO == 128, VO == 16; KH, VI <= 17 (message GVPs)
    const int nts = (KM + 15) >> 4;                  // <= 11
    const int mth = (KH + 15) >> 4, mti = (VI + 15) >> 4;
    const bool lastl = p.level == p.n_gvps - 1, firstl = p.level == 0;
    const int slot = (et == ET_FF || et == ET_FP) ? 0 : (et == ET_PP ? p.pp_slot : 1);
    const float fix_scale = firstl ? p.fix[0] : 1.0f;
    const bool want_sc = lastl && p.norm_mode == 0;
    // weight-gradient accumulators, kept in registers over all the tiles of this block:
    //   to_feats_out: wave wv owns output features 16 wv .. +15, tile x = inputs 16 x .. +15
    //   gates: wave wv owns features 16 wv .. +15 of all 16 gates
    //   waves 6, 7: Wu tile (hidden channels 16 (wv - 6) ..) and Wh tiles (vi block wv - 6, hidden blocks 0 and 1)
    f32x4 accWm[11], accWg = {0.f, 0.f, 0.f, 0.f}, accU = {0.f, 0.f, 0.f, 0.f}, accH[2];
#pragma unroll
    for (int x = 0; x < 11; ++x) accWm[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    accH[0] = accH[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float acc_bm = 0.f, acc_bg = 0.f;                // bias gradients of feature tid (< SO) / gate tid (< VO)
    const float* zl = p.sv_z + (size_t)p.level * p.sv_stride * PF_S;
    const float* gl = p.sv_g + (size_t)p.level * p.sv_stride * 16;
    const float* zprev = firstl ? nullptr : p.sv_z + (size_t)(p.level - 1) * p.sv_stride * PF_S;
    const float* vprev = firstl ? nullptr : p.sv_v + (size_t)(p.level - 1) * p.sv_stride * 48;
    // ---- small weights of the level, zero padded: sWh[a][b] = Wh[vi a][hidden b], sWu[a][b] = Wu[hidden a][out b], sWg[gate][feature]
    for (int idx = tid; idx < 32 * E2_WHS; idx += NT) {
        const int a = idx / E2_WHS, b = idx - a * E2_WHS;
        sWh[idx] = (a < VI && b < KH) ? W[g.o_Wh + a * KH + b] : 0.f;
    }
    for (int idx = tid; idx < 32 * 16; idx += NT) sWu[idx] = (idx >> 4) < KH ? W[g.o_Wu + idx] : 0.f;
    for (int idx = tid; idx < 16 * 128; idx += NT) sWg[idx] = W[g.o_Wg + idx];

    const int rA = tid >> 5, qA = tid & 31;          // scalar rows: float4 qA of rows rA and rA + 16
    const bool hasV = tid < ER * 12;
    const int rV = hasV ? tid / 12 : 0, qV = hasV ? tid - rV * 12 : 0;      // vector rows: float4 qV of row rV
    const int rG = tid >> 4, uG = tid & 15;          // gate rows
    const int cw = wv >> 1, rb = 16 * (wv & 1);      // waves 0..5: column tile (coordinate cw, rows rb .. rb + 15)

    // fetched ahead, in plain registers (aggregates returned from a lambda end up in scratch memory, and every load is then
    // waited for on the spot): indices of the tile after next (i*), rows of the next tile (r*)
    int iA0 = 0, iA1 = 0, iV = 0, iG = 0, isA0 = 0, isA1 = 0, isV = 0, isT = 0, idA0 = 0, idA1 = 0, idV = 0, idT = 0;
    float4 rz0 = {}, rz1 = {}, rx0 = {}, rx1 = {}, ru0 = {}, ru1 = {}, rvx = {}, rvu = {}, rxs = {}, rxd = {};
    float rgt = 0.f;
    int rcA0 = 1, rcA1 = 1, rcV = 1, rsT = 0;
    // every load is unconditional (addresses are selected, not branched on) so that the fetches of a tile are issued back to
    // back; lanes without a vector row (tid >= 384) fetch row 0 again
    auto load_idx = [&](const int j, const int nv) {
        const int m = max(nv, 1) - 1;
        const int* sl = s_slot + j * ER;
        iA0 = sl[min(rA, m)]; iA1 = sl[min(rA + 16, m)]; iV = sl[min(rV, m)]; iG = sl[min(rG, m)];
        const int eT = sl[min(tid & 31, m)];
        isA0 = p.esrc[iA0]; isA1 = p.esrc[iA1]; isV = p.esrc[iV]; isT = p.esrc[eT];
        idA0 = p.edst[iA0]; idA1 = p.edst[iA1]; idV = p.edst[iV]; idT = p.edst[eT];
    };
    auto load_rows = [&]() {
        const float* xb = firstl ? p.h : zprev;
        const float* ub = lastl ? p.gagg_s : p.gs_buf;
        rz0 = reinterpret_cast<const float4*>(zl + (size_t)iA0 * PF_S)[qA];
        rz1 = reinterpret_cast<const float4*>(zl + (size_t)iA1 * PF_S)[qA];
        rx0 = reinterpret_cast<const float4*>(xb + (size_t)(firstl ? isA0 : iA0) * PF_S)[qA];
        rx1 = reinterpret_cast<const float4*>(xb + (size_t)(firstl ? isA1 : iA1) * PF_S)[qA];
        ru0 = reinterpret_cast<const float4*>(ub + (size_t)(lastl ? idA0 : iA0) * PF_S)[qA];
        ru1 = reinterpret_cast<const float4*>(ub + (size_t)(lastl ? idA1 : iA1) * PF_S)[qA];
        rcA0 = p.in_cnt[want_sc ? slot * p.N + idA0 : 0];
        rcA1 = p.in_cnt[want_sc ? slot * p.N + idA1 : 0];
        const float* vb = firstl ? p.v : vprev;            // (conv layer 0 at level 0: the rows are fetched and not used)
        const float* gb = lastl ? p.gagg_v : p.gv_buf;
        rvx = reinterpret_cast<const float4*>(vb + (size_t)(firstl ? isV : iV) * 48)[qV];
        rvu = reinterpret_cast<const float4*>(gb + (size_t)(lastl ? idV : iV) * 48)[qV];
        rcV = p.in_cnt[want_sc ? slot * p.N + idV : 0];
        rgt = gl[(size_t)iG * 16 + uG];
        rsT = isT;
        rxs = p.xn[isT]; rxd = p.xn[idT];
    };

    const int npass = my < cnt_et ? (cnt_et - my + nb - 1) / nb : 0;
    for (int base = 0; base < npass; base += E2_TILES) {
        __syncthreads();
        // this round's passes: pass base + j of this block is pass P = my + (base + j) nb of the etype = rows 32 P .. of its dense
        // row list; their slots go to LDS once per round (one coalesced fetch instead of a dependent one per pass)
        if (tid < E2_TILES) {
            const int P = my + (base + tid) * nb;
            s_tnv[tid] = (base + tid < npass) ? max(min(ER, nrows - ER * P), 0) : 0;
        }
        for (int idx = tid; idx < E2_TILES * ER; idx += NT) {
            const int j = idx >> 5, r = idx & 31;
            const int P = my + (base + j) * nb;
            const int q = ER * P + r;
            s_slot[idx] = (base + j < npass && q < nrows) ? rlist[q] : 0;
        }
        __syncthreads();
        const int cn = min(E2_TILES, npass - base);
        load_idx(0, s_tnv[0]);
        load_rows();
        if (cn > 1) load_idx(1, s_tnv[1]);
        for (int j = 0; j < cn; ++j) {
            const int nv = __builtin_amdgcn_readfirstlane(s_tnv[j]);
            if (nv > 0) {
                PFT_STAMP(30);
                // ---- this tile's rows: registers -> LDS
                {
                E2_PHASE();
                const int rA = tid >> 5, qA = tid & 31;
                const bool hasV = tid < ER * 12;
                const int rV = hasV ? tid / 12 : 0, qV = hasV ? tid - rV * 12 : 0;
                const int rG = tid >> 4, uG = tid & 15;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int row = rA + 16 * c;
                    *reinterpret_cast<float4*>(Zb + row * E2_ZS + 4 * qA) = c ? rz1 : rz0;
                    float4 x = c ? rx1 : rx0;
                    if (!firstl) { x.x = t_silu(x.x); x.y = t_silu(x.y); x.z = t_silu(x.z); x.w = t_silu(x.w); }
                    *reinterpret_cast<float4*>(Sin + row * E2_SS + 4 * qA) = x;
                    const float sc = row < nv ? (want_sc ? 1.0f / (float)(c ? rcA1 : rcA0) : 1.0f) : 0.f;
                    float4 u = c ? ru1 : ru0;
                    u.x *= sc; u.y *= sc; u.z *= sc; u.w *= sc;
                    *reinterpret_cast<float4*>(gA + row * E2_SS + 4 * qA) = u;
                }
                if (hasV) {
                    float* d = Vin + rV * VWS + (firstl ? 3 : 0) + 4 * qV;
                    const float vk = (firstl && p.l0) ? 0.f : 1.0f;            // conv layer 0 has no vector input
                    d[0] = rvx.x * vk; d[1] = rvx.y * vk; d[2] = rvx.z * vk; d[3] = rvx.w * vk;
                    const float sc = rV < nv ? (want_sc ? 1.0f / (float)rcV : 1.0f) : 0.f;
                    float* go = gVo + rV * VWS + 4 * qV;
                    go[0] = rvu.x * sc; go[1] = rvu.y * sc; go[2] = rvu.z * sc; go[3] = rvu.w * sc;
                }
                gate[rG * GTS + uG] = rgt;
                if (tid < ER) {
                    s_e[tid] = s_slot[j * ER + min(tid, nv - 1)];
                    s_src[tid] = rsT;
                    if (firstl) {
                        const float dx = rxs.x - rxd.x, dy = rxs.y - rxd.y, dz = rxs.z - rxd.z;
                        const float d = t_sqrt(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
                        const float rd = __builtin_amdgcn_rcpf(d);
                        Vin[tid * VWS + 0] = dx * rd; Vin[tid * VWS + 1] = dy * rd; Vin[tid * VWS + 2] = dz * rd;
                        for (int k = 0; k < PF_R; ++k) {
                            const float z = (d - p.rbf_mu[k]) * p.rbf_inv_sigma;
                            Sin[tid * E2_SS + PF_S + k] = __expf(-(z * z));
                        }
                    }
                }
                }
                __syncthreads();
                PFT_STAMP(31);
                // ---- Vh = Wh^T V, then Vu = Wu^T Vh on the SAME column tile (coordinate cw of rows rb .. rb + 15): the D
                // fragment of the first product (lane (col, kq): rows 16 mt + 4 kq + r) is the B operand of the second for
                // the k-steps "k = 16 mt + 4 kq + r" -- no LDS round trip, no barrier in between
                if (wv < 6) {
                    E2_PHASE();
                    float bv[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) { const int k = min(4 * u + kq, VI - 1); bv[u] = Vin[(rb + li) * VWS + k * 3 + cw]; }   // (sWh is zero beyond VI)
                    f32x4 au = {0.f, 0.f, 0.f, 0.f};
                    for (int mt = 0; mt < mth; ++mt) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < 5; ++u)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sWh[(4 * u + kq) * E2_WHS + mt * 16 + li], bv[u], acc, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = mt * 16 + kq * 4 + r;                 // hidden channel of acc[r]; rows >= KH are zero (sWh padding)
                            if (i < KH) Vh[(rb + li) * VWS + i * 3 + cw] = acc[r];
                            au = __builtin_amdgcn_mfma_f32_16x16x4f32(sWu[i * 16 + li], acc[r], au, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) Vu[(rb + li) * VWS + (kq * 4 + r) * 3 + cw] = au[r];
                }
                __syncthreads();
                PFT_STAMP(32);
                // ---- gate: V' = sigmoid(gate) Vu
                {
                E2_PHASE();
                for (int idx = tid; idx < ER * KH; idx += NT) {                     // sh = |Vh|
                    const int row = idx & 31, hh = idx >> 5;
                    const float* q = Vh + row * VWS + hh * 3;
                    Sin[row * E2_SS + SI + hh] = t_sqrt(fmaxf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2], 1e-8f));
                }
                for (int idx = tid; idx < ER * VO; idx += NT) {
                    const int row = idx & 31, u = idx >> 5;
                    const float gt = gate[row * GTS + u];
                    float* go = gVo + row * VWS + u * 3;
                    const float* vu = Vu + row * VWS + u * 3;
                    const float dot = go[0] * vu[0] + go[1] * vu[1] + go[2] * vu[2];
                    const float f = t_sigmoid(gt);
                    ggate[row * GTS + u] = dot * f * (1.0f - f);
                    go[0] *= f; go[1] *= f; go[2] *= f;
                }
                }
                __syncthreads();
                PFT_STAMP(33);
                {   // ---- gZ = (gA + ggate Wg) SiLU'(Z) for features 16 wv .. +15 of all rows, on the accumulator fragments
                    E2_PHASE();
                    // (BF16: the 16 gates are one v_mfma_f32_16x16x16_bf16, lane (li, kq) holding gates 4 kq + s)
                    float aw[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) aw[s] = sWg[(BF16 ? 4 * kq + s : 4 * s + kq) * 128 + wv * 16 + li];
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (BF16) {
                            float gq[4];
#pragma unroll
                            for (int s = 0; s < 4; ++s) gq[s] = ggate[(16 * n + li) * GTS + 4 * kq + s];
                            acc = mfma_bf16(bf_pack4(aw), bf_pack4(gq), acc);
                        } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s], ggate[(16 * n + li) * GTS + 4 * s + kq], acc, 0, 0, 0);
                        }
                        const int row = 16 * n + li, f0 = wv * 16 + kq * 4;
                        float4* gap = reinterpret_cast<float4*>(gA + row * E2_SS + f0);
                        const float4 ga = *gap, z = *reinterpret_cast<const float4*>(Zb + row * E2_ZS + f0);
                        const float zz[4] = {z.x, z.y, z.z, z.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w};
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sg = t_sigmoid(zz[r]);
                            o[r] = (gg[r] + acc[r]) * sg * (1.0f + zz[r] * (1.0f - sg));
                        }
                        *gap = float4{o[0], o[1], o[2], o[3]};
                    }
                    // dWg tile (16 gates x features 16 wv .. +15) += ggate^T SiLU(Z)
                    if constexpr (BF16) {                     // the 32 rows are one v_mfma_f32_16x16x32_bf16, lane (li, kq) holding rows 8 kq + u
                        float a8[8], b8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int k = 8 * kq + u;
                            a8[u] = ggate[k * GTS + li];
                            b8[u] = t_silu(Zb[k * E2_ZS + wv * 16 + li]);
                        }
                        accWg = mfma_bf32(bf_pack8(a8), bf_pack8(b8), accWg);
                    } else {
#pragma unroll
                    for (int u = 0; u < ER / 4; ++u) {
                        const int k = 4 * u + kq;
                        accWg = __builtin_amdgcn_mfma_f32_16x16x4f32(ggate[k * GTS + li], t_silu(Zb[k * E2_ZS + wv * 16 + li]), accWg, 0, 0, 0);
                    }
                    }
                    if (tid < VO) { float sm = 0.f; for (int r = 0; r < ER; ++r) sm += ggate[r * GTS + tid]; acc_bg += sm; }
                }
                __syncthreads();
                PFT_STAMP(35);
                {   // ---- gS = gZ Wm: a wave owns m tiles wv and wv + 8; their packed weight fragments (16 x 1 KiB) are all
                    // requested before the first product, the gZ fragments come from LDS as they are used
                    E2_PHASE();
                    // (FX == 1: nine m tiles on eight waves -- the ninth is dealt by row half to waves 0 and 1, which sit on different
                    // SIMDs: whole on wave 0 it made SIMD 0 issue 192 matrix instructions of this phase against 128 on the others)
                    constexpr bool SPLIT9 = FX == 1;
                    const bool two = SPLIT9 ? wv < 2 : wv + NT / 64 < nts;               // wave-uniform
                    const f32x4* wp0 = Wp + (size_t)wv * (8 * 64) + lane;
                    const f32x4* wp1 = Wp + (size_t)(SPLIT9 ? (two ? 8 : wv) : (two ? wv + NT / 64 : wv)) * (8 * 64) + lane;
                    f32x4 aq0[8], aq1[8];
#pragma unroll
                    for (int sb = 0; sb < 8; ++sb) aq0[sb] = wp0[sb * 64];
                    if (two) {
#pragma unroll
                        for (int sb = 0; sb < 8; ++sb) aq1[sb] = wp1[sb * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float* b0p = gA + li * E2_SS + 4 * kq;
                    const float* b1p = gA + (16 + li) * E2_SS + 4 * kq;
                    if constexpr (BF16) {
                        // k blocks 2 s and 2 s + 1 of the f32 fragment table per instruction (lane (li, kq): features 32 s + 4 kq + t and
                        // 32 s + 16 + 4 kq + t of both operands); the gZ pieces are shared by the wave's two m tiles
                        bf16x8 bz0[4], bz1[4];
#pragma unroll
                        for (int sp = 0; sp < 4; ++sp) {
                            bz0[sp] = bf_pack8(*reinterpret_cast<const f32x4*>(b0p + 32 * sp), *reinterpret_cast<const f32x4*>(b0p + 32 * sp + 16));
                            bz1[sp] = bf_pack8(*reinterpret_cast<const f32x4*>(b1p + 32 * sp), *reinterpret_cast<const f32x4*>(b1p + 32 * sp + 16));
                        }
                        if (wv < nts) {
                            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int sp = 0; sp < 4; ++sp) {
                                const bf16x8 a = bf_pack8(aq0[2 * sp], aq0[2 * sp + 1]);
                                acc0 = mfma_bf32(a, bz0[sp], acc0);
                                acc1 = mfma_bf32(a, bz1[sp], acc1);
                            }
                            *reinterpret_cast<f32x4*>(gS + li * E2_SS + 16 * wv + 4 * kq) = acc0;
                            *reinterpret_cast<f32x4*>(gS + (16 + li) * E2_SS + 16 * wv + 4 * kq) = acc1;
                        }
                        if constexpr (SPLIT9) {
                            if (two) {
                                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int sp = 0; sp < 4; ++sp) acc = mfma_bf32(bf_pack8(aq1[2 * sp], aq1[2 * sp + 1]), wv == 0 ? bz0[sp] : bz1[sp], acc);
                                *reinterpret_cast<f32x4*>(gS + (16 * wv + li) * E2_SS + 16 * 8 + 4 * kq) = acc;
                            }
                        } else if (two) {
                            const int mt = wv + NT / 64;
                            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int sp = 0; sp < 4; ++sp) {
                                const bf16x8 a = bf_pack8(aq1[2 * sp], aq1[2 * sp + 1]);
                                acc0 = mfma_bf32(a, bz0[sp], acc0);
                                acc1 = mfma_bf32(a, bz1[sp], acc1);
                            }
                            *reinterpret_cast<f32x4*>(gS + li * E2_SS + 16 * mt + 4 * kq) = acc0;
                            *reinterpret_cast<f32x4*>(gS + (16 + li) * E2_SS + 16 * mt + 4 * kq) = acc1;
                        }
                    } else {
                    if (wv < nts) {
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int sb = 0; sb < 8; ++sb) {
                            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b0p + 16 * sb), b1 = *reinterpret_cast<const f32x4*>(b1p + 16 * sb);
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq0[sb][t], b0[t], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq0[sb][t], b1[t], acc1, 0, 0, 0);
                            }
                        }
                        *reinterpret_cast<f32x4*>(gS + li * E2_SS + 16 * wv + 4 * kq) = acc0;
                        *reinterpret_cast<f32x4*>(gS + (16 + li) * E2_SS + 16 * wv + 4 * kq) = acc1;
                    }
                    if constexpr (SPLIT9) {
                        if (two) {
                            const float* bp = wv == 0 ? b0p : b1p;
                            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int sb = 0; sb < 8; ++sb) {
                                const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 16 * sb);
#pragma unroll
                                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq1[sb][t], b[t], acc, 0, 0, 0);
                            }
                            *reinterpret_cast<f32x4*>(gS + (16 * wv + li) * E2_SS + 16 * 8 + 4 * kq) = acc;
                        }
                    } else if (two) {
                        const int mt = wv + NT / 64;
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int sb = 0; sb < 8; ++sb) {
                            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b0p + 16 * sb), b1 = *reinterpret_cast<const f32x4*>(b1p + 16 * sb);
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq1[sb][t], b0[t], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq1[sb][t], b1[t], acc1, 0, 0, 0);
                            }
                        }
                        *reinterpret_cast<f32x4*>(gS + li * E2_SS + 16 * mt + 4 * kq) = acc0;
                        *reinterpret_cast<f32x4*>(gS + (16 + li) * E2_SS + 16 * mt + 4 * kq) = acc1;
                    }
                    }
                }
                PFT_STAMP(39);
                {   // dWm tiles (features 16 wv .., inputs 16 x ..) += gZ^T [s, sh]
                    E2_PHASE();
                    // (BF16: the 32 rows are one v_mfma_f32_16x16x32_bf16 per input tile, lane (li, kq) holding rows 8 kq + u)
                    float av[ER / 4];
#pragma unroll
                    for (int u = 0; u < ER / 4; ++u) av[u] = gA[(BF16 ? 8 * kq + u : 4 * u + kq) * E2_SS + wv * 16 + li];
                    bf16x8 avb;
                    if constexpr (BF16) avb = bf_pack8(av);
#pragma unroll
                    for (int x = 0; x < 11; ++x)
                        if (x < nts) {
                            float bv[ER / 4];
#pragma unroll
                            for (int u = 0; u < ER / 4; ++u) bv[u] = Sin[(BF16 ? 8 * kq + u : 4 * u + kq) * E2_SS + x * 16 + li];
                            if constexpr (BF16) accWm[x] = mfma_bf32(avb, bf_pack8(bv), accWm[x]);
                            else {
#pragma unroll
                            for (int u = 0; u < ER / 4; ++u) accWm[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], accWm[x], 0, 0, 0);
                            }
                        }
                    if (tid < SO) { float sm = 0.f; for (int r = 0; r < ER; ++r) sm += gA[r * E2_SS + tid]; acc_bm += sm; }
                }
            }
            // ---- fetch ahead: the rows of the next tile (their indices arrived during the previous pass), the indices of the
            // tile after it.  Nothing below this point waits for a global load.
            __builtin_amdgcn_sched_barrier(0);       // the fetches stay behind the products above (hoisted into them they spill)
            if (j + 1 < cn) load_rows();
            if (j + 2 < cn) load_idx(j + 2, s_tnv[j + 2]);
            if (nv > 0) {
                __syncthreads();
                PFT_STAMP(36);
                // ---- gVh = Wu gVu + (gradient through sh), then gVi = Wh gVh on the same column tile through the registers
                // (as Vh -> Vu above); dWu on waves 6, 7
                if (wv < 6) {
                    E2_PHASE();
                    const bool want_vi = !(firstl && p.l0);          // conv layer 0 has no vector input: nobody reads gVi there
                    float bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) bv[u] = gVo[(rb + li) * VWS + (4 * u + kq) * 3 + cw];
                    f32x4 ai[2];
                    ai[0] = ai[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int row = rb + li;
                    for (int mt = 0; mt < mth; ++mt) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sWu[(mt * 16 + li) * 16 + 4 * u + kq], bv[u], acc, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = mt * 16 + kq * 4 + r;                 // hidden channel of acc[r]
                            float gv = 0.f;
                            if (i < KH) {
                                const float* q = Vh + row * VWS + i * 3;
                                const float ss = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
                                const float extra = ss > 1e-8f ? gS[row * E2_SS + SI + i] * q[cw] / Sin[row * E2_SS + SI + i] : 0.f;
                                gv = acc[r] + extra;
                                gVh[row * VWS + i * 3 + cw] = gv;
                            }
                            if (want_vi) {
#pragma unroll
                                for (int m2 = 0; m2 < 2; ++m2)
                                    if (m2 < mti) ai[m2] = __builtin_amdgcn_mfma_f32_16x16x4f32(sWh[(m2 * 16 + li) * E2_WHS + i], gv, ai[m2], 0, 0, 0);
                            }
                        }
                    }
                    if (want_vi)
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int vi = m2 * 16 + kq * 4 + r;
                                if (vi < VI) gVi[row * VWS + vi * 3 + cw] = ai[m2][r];
                            }
                } else if (16 * (wv - 6) < KH) {     // dWu tile (hidden channels 16 (wv - 6) .., 16 outputs) += sum over rows and coordinates of Vh gVu
                    E2_PHASE();
                    const int hh = min((wv - 6) * 16 + li, KH - 1);
#pragma unroll
                    for (int u = 0; u < 3 * ER / 4; ++u) {
                        const int k = 4 * u + kq;
                        accU = __builtin_amdgcn_mfma_f32_16x16x4f32(Vh[(k & 31) * VWS + hh * 3 + (k >> 5)],
                                                                    gVo[(k & 31) * VWS + li * 3 + (k >> 5)], accU, 0, 0, 0);
                    }
                }
                __syncthreads();
                PFT_STAMP(38);
                // ---- dWh on waves 6, 7 (input channels 16 (wv - 6) .., hidden channels 16 tb ..) += V gVh, ahead of their share of the write-out
                if (wv >= 6 && 16 * (wv - 6) < VI) {
                    E2_PHASE();
                    const int vi = min((wv - 6) * 16 + li, VI - 1);
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)
                        if (16 * tb < KH) {
                            const int hh = min(tb * 16 + li, KH - 1);
#pragma unroll
                            for (int u = 0; u < 3 * ER / 4; ++u) {
                                const int k = 4 * u + kq;
                                accH[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vin[(k & 31) * VWS + vi * 3 + (k >> 5)],
                                                                                gVh[(k & 31) * VWS + hh * 3 + (k >> 5)], accH[tb], 0, 0, 0);
                            }
                        }
                }
                // ---- hand the input gradients down: to the level below, or (level 0) to the source nodes
                {
                E2_PHASE();
                if (!firstl) {
                    for (int idx = tid; idx < ER * 32; idx += NT) {
                        const int row = idx >> 5, q = idx & 31;
                        if (row < nv) reinterpret_cast<float4*>(p.gs_buf + (size_t)s_e[row] * PF_S)[q] = *reinterpret_cast<const float4*>(gS + row * E2_SS + 4 * q);
                    }
                    for (int idx = tid; idx < ER * 48; idx += NT) {
                        const int row = idx / 48, q = idx - row * 48;
                        if (row < nv) p.gv_buf[(size_t)s_e[row] * 48 + q] = gVi[row * VWS + q];
                    }
                } else {
                    for (int idx = tid; idx < ER * 128; idx += NT) {
                        const int row = idx >> 7, f = idx & 127;
                        if (row < nv) atomicAdd(reinterpret_cast<unsigned long long*>(p.A_h + (size_t)s_src[row] * PF_S + f),
                                                (unsigned long long)__float2ll_rn(gS[row * E2_SS + f] * fix_scale));
                    }
                    if (!p.l0)
                        for (int idx = tid; idx < ER * 48; idx += NT) {
                            const int row = idx / 48, q = idx - row * 48;
                            if (row < nv) atomicAdd(reinterpret_cast<unsigned long long*>(p.A_v + (size_t)s_src[row] * 48 + q),
                                                    (unsigned long long)__float2ll_rn(gVi[row * VWS + 3 + q] * fix_scale));
                        }
                }
                }
                __syncthreads();
            }
        }
    }
    // ---- the register accumulators ARE this block's share of the GVP's gradient: stored, not added (every block of the
    // etype's range stores every element of the GVP; waves 6 / 7 hold zeros in the Wu / Wh tiles they never touched)
#pragma unroll
    for (int x = 0; x < 11; ++x)
        if (x < nts) {
            const int cj = x * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = wv * 16 + kq * 4 + r;
                if (cj < KM) gp[g.o_Wm + ci * KM + cj] = accWm[x][r];
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) gp[g.o_Wg + (kq * 4 + r) * SO + wv * 16 + li] = accWg[r];
    if (wv >= 6) {
        const int tb6 = wv - 6;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hh = tb6 * 16 + kq * 4 + r;
            if (hh < KH) gp[g.o_Wu + hh * VO + li] = accU[r];
        }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int vi = tb6 * 16 + kq * 4 + r, hh = tb * 16 + li;
                if (vi < VI && hh < KH) gp[g.o_Wh + vi * KH + hh] = accH[tb][r];
            }
    }
    if (tid < SO) gp[g.o_bm + tid] = acc_bm;
    if (tid < VO) gp[g.o_bg + tid] = acc_bg;
}

// ---------------------------------------------------------------------------------------------
// encoders backward: LayerNorm(SiLU(Linear([h, t])))  (dynamics_gvp.py:107-117,143-151)
// ---------------------------------------------------------------------------------------------
// Gg[graph][element][f] = sum over the graph's atoms of that element of G_h[atom][f] (fixed order: thread (phase, f) walks every
// eighth atom, the eight phases are added at the end in phase order)
__global__ __launch_bounds__(1024) void k_enc_group(const float* G_h, const int* prot_ptr, const int* ptype, const int rec_nf, float* Gg) {
    __shared__ float acc[8][16][PF_S];
    const int g = blockIdx.x, tid = threadIdx.x, ph = tid >> 7, f = tid & 127;
    for (int e = 0; e < rec_nf; ++e) acc[ph][e][f] = 0.f;
    const int p1 = prot_ptr[g + 1];
    for (int n = prot_ptr[g] + ph; n < p1; n += 16) {          // two atoms per trip: both loads are in flight before the first add
        const int n2 = n + 8;
        const float x0 = G_h[(size_t)n * PF_S + f];
        const float x1 = n2 < p1 ? G_h[(size_t)n2 * PF_S + f] : 0.f;
        const int e0 = min(max(ptype[n], 0), rec_nf - 1), e1 = n2 < p1 ? min(max(ptype[n2], 0), rec_nf - 1) : 0;
        acc[ph][e0][f] += x0;
        acc[ph][e1][f] += x1;
    }
    __syncthreads();
    for (int idx = tid; idx < rec_nf * PF_S; idx += 1024) {
        const int e = idx >> 7, ff = idx & 127;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += acc[q][e][ff];
        Gg[((size_t)g * rec_nf + e) * PF_S + ff] = t;
    }
}

// k_fix_apply of conv layer 0's scalar accumulator and k_enc_group in one pass over the rows: G_h += A_h / scale, A_h = 0, and the
// per-(graph, element) sums of the updated protein rows (same per-element arithmetic and the same summation order as the two
// kernels one after the other; the 34 MB of G_h are not read a second time).  Blocks [0, B): one graph's protein atoms; the blocks
// behind them: the pharm rows [Np, Np + Nf), element by element
// (onehot_flag[0] == 0: the encoders' backward reads the protein rows' gradient from Gg only -- their updated G_h rows, which nothing
// else reads after conv layer 0, are then not written back: 34 MB of stores less at 256 pockets)
__global__ __launch_bounds__(1024) void k_fix_enc_group(long long* A_h, float* G_h, const float* fix, const int* prot_ptr, const int* ptype,
                                                        const int B, const int rec_nf, float* Gg, const int Np, const int Nf,
                                                        const int* onehot_flag) {
    __shared__ float acc[8][16][PF_S];
    const int tid = threadIdx.x;
    const double inv = (double)fix[1];
    if ((int)blockIdx.x >= B) {
        const size_t n = (size_t)Nf * PF_S, i = (size_t)(blockIdx.x - B) * 1024 + tid;
        if (i < n) {
            const size_t o = (size_t)Np * PF_S + i;
            const long long a = A_h[o];
            if (a != 0) { G_h[o] += (float)((double)a * inv); A_h[o] = 0; }
        }
        return;
    }
    const int g = blockIdx.x, ph = tid >> 7, f = tid & 127;
    const bool keep = onehot_flag[0] != 0;                      // block-uniform: the ungrouped encoders' backward reads G_h itself
    for (int e = 0; e < rec_nf; ++e) acc[ph][e][f] = 0.f;
    const int p1 = prot_ptr[g + 1];
    for (int n = prot_ptr[g] + ph; n < p1; n += 32) {          // four atoms per trip, all their loads in flight before the first add
        long long a[4]; float x[4]; int e[4]; bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int nn = n + 8 * u;
            on[u] = nn < p1;
            const size_t o = (size_t)(on[u] ? nn : n) * PF_S + f;
            a[u] = A_h[o]; x[u] = G_h[o];
            e[u] = min(max(ptype[on[u] ? nn : n], 0), rec_nf - 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (on[u]) {
                const size_t o = (size_t)(n + 8 * u) * PF_S + f;
                float v = x[u];
                if (a[u] != 0) { v += (float)((double)a[u] * inv); if (keep) G_h[o] = v; A_h[o] = 0; }
                acc[ph][e[u]][f] += v;
            }
    }
    __syncthreads();
    for (int idx = tid; idx < rec_nf * PF_S; idx += 1024) {
        const int e = idx >> 7, ff = idx & 127;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += acc[q][e][ff];
        Gg[((size_t)g * rec_nf + e) * PF_S + ff] = t;
    }
}

__global__ __launch_bounds__(NT) void k_bwd_encode(const BwdEncodeParams p) {
    __shared__ float sin_[TR * 17], z[TR * ZS], xh[TR * ZS], gq[TR * ZS];
    __shared__ float red[NT];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* W = p.c.W;
    float* gp = p.c.gpart_enc + (size_t)blockIdx.x * p.c.enc_n - p.c.enc_begin;       // gp[flat offset] as in the other kernels
    for (int i = threadIdx.x; i < p.c.enc_n; i += NT) gp[p.c.enc_begin + i] = 0.f;
    __syncthreads();
    const int N = p.Np + p.Nf;
    // Protein features that are element one-hots (what the dataset and the CLI produce; the bind's device-side verdict is read here,
    // nobody waits for it on the host): the encoder output of an atom is a function of (graph, element) only, and everything
    // behind it is linear in the upstream gradient, so the atoms of a graph are differentiated as rec_nf VIRTUAL rows whose
    // upstream gradient is the sum over the atoms of that element (k_enc_group) -- 2,816 rows instead of 65,536 at config 5.
    const bool grouped = p.Gg != nullptr && p.onehot_flag[0] == 0;          // block-uniform
    const int n_prot_rows = grouped ? p.B * p.rec_nf : p.Np;
    const int tiles_prot = (n_prot_rows + TR - 1) / TR, tiles_pharm = (p.Nf + TR - 1) / TR;
    for (int ti = blockIdx.x; ti < tiles_prot + tiles_pharm; ti += gridDim.x) {
        const int nt = ti < tiles_prot ? 0 : 1;
        const int first = nt ? p.Np + (ti - tiles_prot) * TR : ti * TR;
        const int end = nt ? N : n_prot_rows;
        const bool virt = grouped && nt == 0;
        auto Gat = [&](const int r, const int f) {     // upstream gradient of row first + r
            return virt ? p.Gg[(size_t)(first + r) * PF_S + f] : p.G_h[(size_t)(first + r) * PF_S + f];
        };
        const int nv = min(TR, end - first);
        const int nf = nt ? p.pharm_nf : p.rec_nf;
        const int K = nf + 1;
        const int ow = p.o_w[nt], ob = p.o_b[nt], olw = p.o_lw[nt], olb = p.o_lb[nt];
        {   // rows outside the receptive field of the output (most protein atoms of a pruned layer) have an exactly zero
            // incoming gradient: their tile contributes nothing to any parameter gradient and is skipped
            int nz = 0;
            for (int idx = tid; idx < TR * 128; idx += NT) {
                const int r = idx >> 7, f = idx & 127;
                if (r < nv && Gat(r, f) != 0.f) nz = 1;
            }
            if (!__syncthreads_or(nz)) continue;       // block-uniform
        }
        for (int idx = tid; idx < TR * K; idx += NT) {
            const int row = idx / K, k = idx - row * K;
            const int n = first + min(row, nv - 1);
            float x;
            if (virt) x = k < nf ? (k == n % nf ? 1.0f : 0.f) : p.t[n / nf];          // virtual row n = graph * rec_nf + element
            else if (k < nf) x = nt ? p.pharm_h[(size_t)(n - p.Np) * nf + k] : p.prot_h0[(size_t)n * nf + k];
            else x = p.t[p.gid[n]];
            sin_[row * 17 + k] = x;
        }
        __syncthreads();
        for (int idx = tid; idx < TR * 128; idx += NT) {
            const int row = idx & 15, o = idx >> 4;
            float s = W[ob + o];
            for (int k = 0; k < K; ++k) s = fmaf(W[ow + o * K + k], sin_[row * 17 + k], s);
            z[row * ZS + o] = s;
            xh[row * ZS + o] = t_silu(s);
        }
        __syncthreads();
        {
            const float mean = row_mean128([&](int r, int f) { return xh[r * ZS + f]; }, red, tid);
            const float var = row_mean128([&](int r, int f) { const float d = xh[r * ZS + f] - mean; return d * d; }, red, tid);
            const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
            const int row = tid & 15, part = tid >> 4;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < FPP; ++q) {
                const int f = part * FPP + q;
                xh[row * ZS + f] = (xh[row * ZS + f] - mean) * rstd;
            }
            for (int idx = tid; idx < TR * 128; idx += NT) {
                const int r = idx >> 7, f = idx & 127;
                gq[r * ZS + f] = r < nv ? Gat(r, f) : 0.f;
            }
            __syncthreads();
            if (tid < 128) {
                float sw = 0.f, sb = 0.f;
                for (int r = 0; r < TR; ++r) { const float go = gq[r * ZS + tid]; sw += go * xh[r * ZS + tid]; sb += go; }
                gp[olw + tid] += sw;
                gp[olb + tid] += sb;
            }
            const float m1 = row_mean128([&](int r, int f) { return gq[r * ZS + f] * W[olw + f]; }, red, tid);
            const float m2 = row_mean128([&](int r, int f) { return gq[r * ZS + f] * W[olw + f] * xh[r * ZS + f]; }, red, tid);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < FPP; ++q) {
                const int f = part * FPP + q;
                const float ga = rstd * (gq[row * ZS + f] * W[olw + f] - m1 - xh[row * ZS + f] * m2);
                const float zz = z[row * ZS + f];
                const float s = t_sigmoid(zz);
                gq[row * ZS + f] = ga * s * (1.0f + zz * (1.0f - s));            // dL/dz
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 128 * K; idx += NT) {
            const int o = idx / K, k = idx - o * K;
            float s = 0.f;
            for (int r = 0; r < TR; ++r) s += gq[r * ZS + o] * sin_[r * 17 + k];
            gp[ow + idx] += s;
        }
        if (tid < 128) {
            float s = 0.f;
            for (int r = 0; r < TR; ++r) s += gq[r * ZS + tid];
            gp[ob + tid] += s;
        }
        __syncthreads();
    }
    (void)lane; (void)wv;
}

// grad[i] = sum over the per-block copies, in block order
// G += A / scale, A = 0: the fixed-point scatter sums of level 0 join the float gradient; the accumulators are left
// clear for the next layer / step.  Element order is fixed, so the result is deterministic.
__global__ void k_fix_apply(long long* A, float* G, const size_t n, const float* fix) {
    const double inv = (double)fix[1];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long long a = A[i];
        if (a != 0) {
            G[i] += (float)((double)a * inv);
            A[i] = 0;
        }
    }
}
// the same for the LAST conv layer, whose edges (ff, pf) have pharm nodes and active protein atoms as their only sources: one
// block per node tile of the pruned layout (pharm tiles, then the per-graph active-atom tiles) instead of a pass over all
// N rows of both accumulators (69 + 26 MB read at 256 pockets for ~7 k touched rows)
__global__ __launch_bounds__(192) void k_fix_apply_rows(const NodeTile* tiles, const int* dyn_cnt, const int* row_ids,
                                                        long long* A_h, float* G_h, long long* A_v, float* G_v, const float* fix) {
    // one block per row slot (tile, row): 176 elements, one per thread -- two dependent loads deep, nothing sequential
    const NodeTile t = tiles[blockIdx.x >> 5];
    const int r = blockIdx.x & 31;
    int n = t.n;
    if (t.cnt_idx >= 0) n = min(n, max(dyn_cnt[t.cnt_idx] - t.rel, 0));
    const int q = threadIdx.x;
    if (r >= n || q >= 176) return;
    const int pos = t.n0 + r;
    const size_t node = (size_t)(t.ids ? row_ids[pos] : pos);
    long long* Ap = q < PF_S ? A_h + node * PF_S + q : A_v + node * 48 + (q - PF_S);
    const long long a = *Ap;
    if (a != 0) {
        float* Gp = q < PF_S ? G_h + node * PF_S + q : G_v + node * 48 + (q - PF_S);
        *Gp += (float)((double)a * (double)fix[1]);
        *Ap = 0;
    }
}
// scale of the fixed-point scatter for one backward call: 2^(PFT_FIX_BITS - ceil(log2(max |upstream gradient|))) -- every
// gradient of the call is linear in the upstream ones, so this keeps ~40 bits below and 23 bits above their largest entry
__global__ __launch_bounds__(256) void k_fix_scale(const float* g_h, const int n_h, const float* g_x, const int n_x, float* fix) {
    __shared__ float red[256];
    float m = 0.f;
    for (int i = threadIdx.x; i < n_h; i += 256) m = fmaxf(m, fabsf(g_h[i]));
    for (int i = threadIdx.x; i < n_x; i += 256) m = fmaxf(m, fabsf(g_x[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) {
        int e = 0;
        const float mx = red[0];
        if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);         // mx = f * 2^e, 0.5 <= f < 1
        const int k = max(-100, min(100, PFT_FIX_BITS - e));
        fix[0] = ldexpf(1.0f, k);
        fix[1] = ldexpf(1.0f, -k);
    }
}
// (the encoders' gradient -- up to PFT_ENC_BLOCKS narrow copies -- is summed by the blocks behind the main ones of k_train_reduce)
template <int UNR>
__global__ __launch_bounds__(256) void k_train_reduce(const ReduceParams p, const int nb_main, const int with_enc) {
    if ((int)blockIdx.x >= nb_main) {
        // the encoders' narrow copies as the blocks behind the main ones (256 threads: 32 parameters x 8
        // slices of every eighth copy, the slices added in order) -- a launch of its own ran its ~100 latency-bound blocks in
        // front of this bandwidth-bound grid instead of under it
        __shared__ float part[8][33];
        if (!with_enc) return;
        const int pi = threadIdx.x & 31, sl = threadIdx.x >> 5;
        const int i = (blockIdx.x - nb_main) * 32 + pi;
        float s = 0.f;
        if (i < p.enc_n) {
            int b = sl;
            for (; b + 8 * 7 < p.enc_grid; b += 8 * 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = p.gpart_enc[(size_t)(b + 8 * u) * p.enc_n + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += x[u];
            }
            for (; b < p.enc_grid; b += 8) s += p.gpart_enc[(size_t)b * p.enc_n + i];
        }
        part[sl][pi] = s;
        __syncthreads();
        if (sl == 0 && i < p.enc_n) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += part[q][pi];
            p.grad[p.enc_begin + i] = t;
        }
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.nparams) return;
    int lo = 0, hi = p.ntens - 1;                    // last tensor that begins at or before i (empty tensors precede their successor)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (p.tseg[mid].begin <= i) lo = mid; else hi = mid - 1;
    }
    const int cls = p.tseg[lo].cls;
    int b0 = 0, b1 = 0;
    if (cls == PFT_CLS_ENC) return;                  // the blocks behind nb_main
    if (cls < 0 || !((p.cls_mask >> cls) & 1u)) return;      // another launch's share (or an empty tensor)
    if (cls == PFT_CLS_HEAD) b1 = p.head_grid;
    else if (cls >= PFT_CLS_MSG) {
        const int l = (cls - PFT_CLS_MSG) >> 2, et = (cls - PFT_CLS_MSG) & 3;
        int nbk;
        et_blocks(p.ccnt + 16 * l, p.n_et[l], p.NB, et, b0, nbk);
        b1 = b0 + nbk;
    } else if (cls >= PFT_CLS_NODE) b1 = p.node_grid[cls - PFT_CLS_NODE];
    float s = 0.f;
    int b = b0;
    for (; b + UNR <= b1; b += UNR) {                // UNR copies in flight, summed in block order
        float x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) x[u] = __builtin_nontemporal_load(p.gpart + (size_t)(b + u) * p.gstride + i);
#pragma unroll
        for (int u = 0; u < UNR; ++u) s += x[u];
    }
    for (; b < b1; ++b) s += __builtin_nontemporal_load(p.gpart + (size_t)b * p.gstride + i);
    p.grad[i] = s;
}

// packed[i] = flat[map[i]] (map[i] < 0: zero padding): re-pack the MFMA-fragment weights after the parameters changed
__global__ void k_gather_weights(const float* flat, const int* map, const size_t n, float* packed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int m = map[i]; packed[i] = m >= 0 ? flat[m] : 0.f; }
}
// -DN16_SPLIT builds: the words of the n16 streams that hold two bf16 of one plane of the split weights (pf_host.cpp: pack_n16_raw):
// tab[i] = (word position in the packed buffer, flat index a, flat index b, plane); -1: a padded weight (zero)
__global__ void k_n16_split_words(const float* flat, const int4* tab, const size_t n, unsigned int* packed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 t = tab[i];
    unsigned int word = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int src = k ? t.z : t.y;
        float x = src >= 0 ? flat[src] : 0.f;
        unsigned int bits = 0;
        for (int p = 0; p <= t.w; ++p) {
            const unsigned int u = __float_as_uint(x);
            bits = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
            x -= __uint_as_float(bits << 16);
        }
        word |= (bits & 0xffffu) << (16 * k);
    }
    packed[t.x] = word;
}
// four consecutive entries per thread (map and packed 16-byte aligned): one 16-byte map load, four gathers in flight, one 16-byte store
__global__ void k_gather_weights4(const float* flat, const int4* map, const size_t n4, float4* packed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int4 m = map[i];
    const float a = flat[max(m.x, 0)], b = flat[max(m.y, 0)], c = flat[max(m.z, 0)], d = flat[max(m.w, 0)];
    packed[i] = make_float4(m.x >= 0 ? a : 0.f, m.y >= 0 ? b : 0.f, m.z >= 0 ? c : 0.f, m.w >= 0 ? d : 0.f);
}

// one Adam step (torch.optim.Adam semantics, pharmacodiff.py:253: L2 weight decay added to the gradient, bias-corrected
// moments, no amsgrad) on flat vectors
// (mirror: the handle's own copy of the flat parameter vector receives the new value in the same pass -- pf_adam_step used to copy
// the 3 MB vector behind this kernel)
__global__ void k_adam(float* p, const float* g, float* m, float* v, const size_t n, const float lr, const float b1,
                       const float b2, const float eps, const float wd, const float bc1, const float bc2_sqrt, float* mirror) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = m[i] + (1.0f - b1) * (gi - m[i]);           // lerp, as torch
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pn = pi - (lr / bc1) * (mi / denom);
    p[i] = pn;
    if (mirror) mirror[i] = pn;
}

// ---------------------------------------------------------------------------------------------
// The loss around the dynamics (PharmacophoreDiff.forward, pharmacodiff.py:162-243, noise parameterisation) as two small
// kernels instead of ~60 framework launches per step:
//   k_loss_prepare (one block per graph): COM of the clean centers taken off the centers and the pocket (:176-183), z_t =
//     alpha_t x0 + sigma_t eps with alpha / sigma looked up at t_int (:186-197), second COM removal (:199-205); writes the
//     dynamics' input state (xn, pharm_h, t) in place
//   k_loss_eval (one block, fixed reduction order): the two losses (:208-232), the four metrics (:234-241) and the unit
//     upstream gradients d(pos loss)/d(eps_x), d(feat loss)/d(eps_h) for pf_train_loss_backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loss_prepare(const LossParams p) {
    __shared__ float s_sh[8];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int f0 = p.pharm_ptr[g], nfg = p.pharm_ptr[g + 1] - f0;          // <= 64 centers per graph
    const int ti = min(max(p.t_int[g], 0), p.T);        // the tables hold T + 1 entries (callers validate; never read outside them)
    const float a = p.alpha_tab[ti], sg = p.sigma_tab[ti];
    if (tid < 64) {
        auto wsum = [](float v) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            return v;
        };
        const bool on = tid < nfg;
        const int f = f0 + (on ? tid : 0);
        float x = on ? p.x0[3 * f] : 0.f, y = on ? p.x0[3 * f + 1] : 0.f, z = on ? p.x0[3 * f + 2] : 0.f;
        const float cnt = (float)max(nfg, 1);
        const float cx = wsum(x) / cnt, cy = wsum(y) / cnt, cz = wsum(z) / cnt;
        x -= cx; y -= cy; z -= cz;
        float xt = on ? a * x + sg * p.eps_x[3 * f] : 0.f, yt = on ? a * y + sg * p.eps_x[3 * f + 1] : 0.f,
              zt = on ? a * z + sg * p.eps_x[3 * f + 2] : 0.f;
        float mx = 0.f, my = 0.f, mz = 0.f;
        if (p.remove_com) { mx = wsum(xt) / cnt; my = wsum(yt) / cnt; mz = wsum(zt) / cnt; }
        if (on) {
            p.x0c[3 * f] = x; p.x0c[3 * f + 1] = y; p.x0c[3 * f + 2] = z;
            p.xn[p.Np + f] = make_float4(xt - mx, yt - my, zt - mz, 0.f);
            for (int k = 0; k < p.nf; ++k)
                p.pharm_h[(size_t)f * p.nf + k] = a * (p.h0[(size_t)f * p.nf + k] / p.feat_norm) + sg * p.eps_h[(size_t)f * p.nf + k];
        }
        if (tid == 0) {
            s_sh[0] = cx; s_sh[1] = cy; s_sh[2] = cz; s_sh[3] = mx; s_sh[4] = my; s_sh[5] = mz;
            p.t[g] = (float)ti / (float)p.T;
            p.alpha_g[g] = a; p.sigma_g[g] = sg;
        }
    }
    __syncthreads();
    const float cx = s_sh[0], cy = s_sh[1], cz = s_sh[2], mx = s_sh[3], my = s_sh[4], mz = s_sh[5];
    for (int n = p.prot_ptr[g] + tid; n < p.prot_ptr[g + 1]; n += 256)
        p.xn[n] = make_float4(p.prot_x0[3 * n] - cx - mx, p.prot_x0[3 * n + 1] - cy - my, p.prot_x0[3 * n + 2] - cz - mz, 0.f);
}
// one center per thread, 64 centers per block (a single block spent its time issuing ~30 strided loads per center from one compute
// unit: 18 us at 1,536 centers); every block leaves its six partial sums, the block that draws the last ticket adds them in block
// order -- the result does not depend on which block that is -- and re-arms the ticket
__global__ __launch_bounds__(64) void k_loss_eval(const LossParams p) {
    const int lane = threadIdx.x;
    const int f = blockIdx.x * 64 + lane;
    const bool on = f < p.Nf;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float inv_x = 1.0f / (float)(p.Nf * 3), inv_h = 1.0f / (float)(p.Nf * p.nf);
    if (on) {
        // every load of the center is requested before the first is used (pharm_nf <= 8)
        const int g = p.gid[p.Np + f];
        float ex[3], dx[3], xc[3], eh[8], dh[8], ph[8], h0v[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) { ex[c] = p.eps_x[3 * f + c]; dx[c] = p.dyn_x[3 * f + c]; xc[c] = p.x0c[3 * f + c]; }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < p.nf) {
                const size_t o = (size_t)f * p.nf + k;
                eh[k] = p.eps_h[o]; dh[k] = p.dyn_h[o]; ph[k] = p.pharm_h[o]; h0v[k] = p.h0[o];
            } else eh[k] = dh[k] = ph[k] = h0v[k] = 0.f;
        const float4 xt = p.xn[p.Np + f];
        const float a = p.alpha_g[g], sg = p.sigma_g[g], wm = 1.0f - p.t[g], wl = p.weighted ? wm : 1.0f;
        const float xtv[3] = {xt.x, xt.y, xt.z};
        float xl = 0.f, err = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = ex[c] - dx[c];
            xl += d * d;
            p.g_x[3 * f + c] = -2.0f * wl * d * inv_x;
            const float e = (xtv[c] - sg * dx[c]) / a - xc[c];
            err += e * e;
        }
        float hl = 0.f, bp = 0.f, bt = 0.f;
        int ip = 0, it = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < p.nf) {
                const float d = eh[k] - dh[k];
                hl += d * d;
                p.g_h[(size_t)f * p.nf + k] = -2.0f * wl * d * inv_h;
                const float hp = (ph[k] - sg * dh[k]) / a, ht = h0v[k];
                if (k == 0 || hp > bp) { bp = hp; ip = k; }                  // first maximum, like argmax
                if (k == 0 || ht > bt) { bt = ht; it = k; }
            }
        const float hit = ip == it ? 1.0f : 0.f;
        acc[0] = xl * wl; acc[1] = hl * wl; acc[2] = err; acc[3] = wm * err; acc[4] = hit; acc[5] = wm * hit;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float v = acc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        acc[q] = v;
    }
    int last = 0;
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) p.part[(size_t)blockIdx.x * 8 + q] = acc[q];
        __threadfence();
        last = atomicAdd(p.ticket, 1) == (int)gridDim.x - 1;
    }
    last = __shfl(last, 0);
    if (!last) return;
    __threadfence();
    if (lane < 6) {
        float tsum = 0.f;
        for (int b = 0; b < (int)gridDim.x; ++b) tsum += __builtin_nontemporal_load(p.part + (size_t)b * 8 + lane);
        const float den = lane == 0 ? (float)(p.Nf * 3) : (lane == 1 ? (float)(p.Nf * p.nf) : (float)p.Nf);
        acc[0] = tsum / den;
    }
    const float v0 = __shfl(acc[0], 0), v1 = __shfl(acc[0], 1), v2 = __shfl(acc[0], 2), v3 = __shfl(acc[0], 3), v4 = __shfl(acc[0], 4),
                v5 = __shfl(acc[0], 5);
    if (lane < 6) p.out[lane] = acc[0];
    if (lane == 0) {
        // what training_step / validation_step derive from the six (pharmacodiff.py:274-277): total loss, total error,
        // weighted total error -- here, so that the step does not spend framework launches on three additions
        p.out[6] = v0 + v1;
        p.out[7] = v2 + 1.0f - v4;
        p.out[8] = v3 + 1.0f - v5;
        *p.ticket = 0;
    }
}
// the unit gradients of the two losses times their upstream scalars (a + a2 for the coordinates, b + b2 for the features;
// a2 / b2 may be null): one launch for both arrays
__global__ void k_scale_loss(float* gx, const int nx, const float* a, const float* a2, float* gh, const int nh, const float* b, const float* b2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nx) gx[i] *= a[0] + (a2 ? a2[0] : 0.f);
    else if (i - nx < nh) gh[i - nx] *= b[0] + (b2 ? b2[0] : 0.f);
}

// dropout masks as the forward applies them, for tests: out[(node * 144 + elem)] in {0, 1/(1-p)}
__global__ void k_drop_masks(const TrainCommon c, const uint32_t stream, const int n_elems, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_elems) out[i] = drop_mul(c, stream, (uint32_t)i);
}

extern "C" {
#ifdef PFT_STAMPS
int pft_read_stamps(unsigned long long* host, int reset) {
    int n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_pft_nstamp), sizeof(int));
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pft_stamps), sizeof(unsigned long long) * 128);
    if (reset) { int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pft_nstamp), &z, sizeof(int)); }
    return n;
}
#endif
void pfk_bwd_head(const BwdHeadParams* p, int nblocks, hipStream_t s) {
    if (p->ntiles == 0) return;
    if (p->c.bf16) hipLaunchKernelGGL(k_bwd_head<true>, dim3(nblocks), dim3(NT), 0, s, *p);
    else hipLaunchKernelGGL(k_bwd_head<false>, dim3(nblocks), dim3(NT), 0, s, *p);
}
void pfk_bwd_node(const BwdNodeParams* p, int nblocks, hipStream_t s) {
    if (p->ntiles == 0) return;
    if (p->c.bf16) hipLaunchKernelGGL(k_bwd_node<true>, dim3(nblocks), dim3(NT), 0, s, *p);
    else hipLaunchKernelGGL(k_bwd_node<false>, dim3(nblocks), dim3(NT), 0, s, *p);
}
void pfk_bwd_edge_level(const BwdEdgeLevelParams* p, int nblocks, hipStream_t s) {
    if (nblocks == 0 || p->et_tile0[p->n_et] == p->et_tile0[0]) return;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), 0, s, *p); };
    if (p->c.bf16) { if (p->fx == 1) go(k_bwd_edge_level<true, 1>); else if (p->fx == 2) go(k_bwd_edge_level<true, 2>); else go(k_bwd_edge_level<true, 0>); }
    else { if (p->fx == 1) go(k_bwd_edge_level<false, 1>); else if (p->fx == 2) go(k_bwd_edge_level<false, 2>); else go(k_bwd_edge_level<false, 0>); }
}
void pfk_compact_node_rows(const NodeTile* tiles, int ntiles, const int* dyn_cnt, const int* row_ids, int N, int* list, int cap, int* ucnt,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_compact_node_rows, dim3(2), dim3(1024), 0, s, tiles, ntiles, dyn_cnt, row_ids, N, reinterpret_cast<int2*>(list), cap, ucnt);
}
void pfk_compact_rows(const EdgeTile* tiles, const int* et_tile0, int n_et, const int* dyn_cnt, int* rlist, int* ccnt, hipStream_t s) {
    if (n_et <= 0) return;
    CompactParams cp;
    for (int et = 0; et <= 4; ++et) cp.et_tile0[et] = et_tile0[et];
    hipLaunchKernelGGL(k_compact_rows, dim3(n_et), dim3(1024), 0, s, tiles, cp, dyn_cnt, rlist, ccnt);
}
void pfk_loss_prepare(const LossParams* p, hipStream_t s) { hipLaunchKernelGGL(k_loss_prepare, dim3(p->B), dim3(256), 0, s, *p); }
void pfk_loss_eval(const LossParams* p, hipStream_t s) { hipLaunchKernelGGL(k_loss_eval, dim3((p->Nf + 63) / 64), dim3(64), 0, s, *p); }
void pfk_scale_loss(float* gx, int nx, const float* a, const float* a2, float* gh, int nh, const float* b, const float* b2, hipStream_t s) {
    if (nx + nh > 0) hipLaunchKernelGGL(k_scale_loss, dim3((nx + nh + 255) / 256), dim3(256), 0, s, gx, nx, a, a2, gh, nh, b, b2);
}
void pfk_pack_gvp(const float* W, const GvpT* g, int n_gvps, float* out_b, float* out_f, const ScaleArgs* sa, hipStream_t s) {
    ScaleArgs none{};
    const ScaleArgs& a = sa ? *sa : none;
    const int extra = (a.nx + a.nh + 63) / 64;
    if (n_gvps == 0 && extra == 0) return;
    hipLaunchKernelGGL(k_pack_gvp, dim3(n_gvps * 88 + extra, 2), dim3(64), 0, s, W, g, out_b, out_f, n_gvps * 88, a);
}
void pfk_fix_apply(long long* A, float* G, size_t n, const float* fix, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_fix_apply, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, s, A, G, n, fix);
}
void pfk_fix_apply_rows(const NodeTile* tiles, int ntiles, const int* dyn_cnt, const int* row_ids, long long* A_h, float* G_h,
                        long long* A_v, float* G_v, const float* fix, hipStream_t s) {
    if (ntiles > 0) hipLaunchKernelGGL(k_fix_apply_rows, dim3(ntiles * 32), dim3(192), 0, s, tiles, dyn_cnt, row_ids, A_h, G_h, A_v, G_v, fix);
}
void pfk_fix_scale(const float* g_h, int n_h, const float* g_x, int n_x, float* fix, hipStream_t s) {
    hipLaunchKernelGGL(k_fix_scale, dim3(1), dim3(256), 0, s, g_h, n_h, g_x, n_x, fix);
}
void pfk_enc_group(const float* G_h, const int* prot_ptr, const int* ptype, int B, int rec_nf, float* Gg, hipStream_t s) {
    if (B > 0) hipLaunchKernelGGL(k_enc_group, dim3(B), dim3(1024), 0, s, G_h, prot_ptr, ptype, rec_nf, Gg);
}
void pfk_fix_enc_group(long long* A_h, float* G_h, const float* fix, const int* prot_ptr, const int* ptype, int B, int rec_nf, float* Gg,
                       int Np, int Nf, const int* onehot_flag, hipStream_t s) {
    const int extra = (int)(((size_t)Nf * PF_S + 1023) / 1024);
    if (B + extra > 0)
        hipLaunchKernelGGL(k_fix_enc_group, dim3(B + extra), dim3(1024), 0, s, A_h, G_h, fix, prot_ptr, ptype, B, rec_nf, Gg, Np, Nf, onehot_flag);
}
void pfk_bwd_encode(const BwdEncodeParams* p, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_bwd_encode, dim3(nblocks), dim3(NT), 0, s, *p);
}
void pfk_train_reduce(const ReduceParams* p, hipStream_t s) {
    const bool enc = p->enc_n > 0 && ((p->cls_mask >> PFT_CLS_ENC) & 1u);
    const bool main_part = (p->cls_mask & ~(1u << PFT_CLS_ENC)) != 0;
    if (!enc && !main_part) return;
    // 16 copies in flight per thread: 43 us per launch against 63 at 8 and 64 at 32 (16-byte loads, four parameters per thread: 97)
    const int nb_main = main_part ? (p->nparams + 255) / 256 : 0;
    const int nb_enc = enc ? (p->enc_n + 31) / 32 : 0;
    hipLaunchKernelGGL(k_train_reduce<16>, dim3(nb_main + nb_enc), dim3(256), 0, s, *p, nb_main, enc ? 1 : 0);
}
void pfk_n16_split_words(const float* flat, const int4* tab, size_t n, float* packed, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_n16_split_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, flat, tab, n, reinterpret_cast<unsigned int*>(packed));
}
void pfk_gather_weights(const float* flat, const int* map, size_t n, float* packed, hipStream_t s) {
    if (n == 0) return;
    if ((((uintptr_t)map | (uintptr_t)packed) & 15) == 0 && n >= 4) {
        const size_t n4 = n / 4;
        hipLaunchKernelGGL(k_gather_weights4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, flat, reinterpret_cast<const int4*>(map), n4,
                           reinterpret_cast<float4*>(packed));
        const size_t rest = n - 4 * n4;
        if (rest) hipLaunchKernelGGL(k_gather_weights, dim3(1), dim3(256), 0, s, flat, map + 4 * n4, rest, packed + 4 * n4);
        return;
    }
    hipLaunchKernelGGL(k_gather_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, flat, map, n, packed);
}
void pfk_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
              float bc1, float bc2_sqrt, float* mirror, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, mirror);
}
void pfk_drop_masks(const TrainCommon* c, uint32_t stream, int n_elems, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_drop_masks, dim3((n_elems + 255) / 256), dim3(256), 0, s, *c, stream, n_elems, out);
}
}
