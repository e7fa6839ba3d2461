// split_bf16_chain.hip -- probe: the scalar Linear of an n16 GVP block -- 16 rows on the four waves of a workgroup,
// wave w owning output features [32 w, 32 w + 32), activations exchanged through LDS once per layer, weights streamed from global
// memory through a register ring -- as a chain of D layers X <- SiLU(W X + b), in three arithmetics:
//   f32    v_mfma_f32_16x16x4_f32 (what ships: exact fp32 products), 64 matrix instructions per layer and wave
//   bf16x6 v_mfma_f32_16x16x32_bf16 on operands split into three bf16 planes each (x = x0 + x1 + x2, 8 mantissa bits per plane),
//          the six products with i + j <= 2, fp32 accumulation: 48 instructions of half the issue time, weights 1.5 x the bytes
//   bf16x3 the three products with i + j <= 1 (two planes): 24 instructions, weights 1.0 x the bytes
// Reports the time per layer (alone on a CU, 1 / 2 / 4 items per CU) and the error of the chain's output against an fp64 host
// evaluation of the same chain.  The K-order trick of pf_n16.hip carries over: a lane's D fragment (features 16 T + 4 g + r of row j)
// is what the SAME lane holds of the next layer's B operand when chunk c of K is {tile 2 c, tile 2 c + 1} and element e <-> (tile 2 c +
// e / 4, register e % 4) -- the weights are packed to that order on the host.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probes/split_bf16_chain.hip -o split_bf16_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float siluf_(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// round-to-nearest-even bf16 of x as a float (the low 16 bits zero), and its 16-bit pattern
__device__ __forceinline__ unsigned bf16_bits(const float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
struct Split3 { unsigned p0, p1, p2; };               // bit patterns of the three planes
__device__ __forceinline__ Split3 split3(const float x) {
    Split3 s;
    s.p0 = bf16_bits(x);
    const float r1 = x - __uint_as_float(s.p0 << 16);
    s.p1 = bf16_bits(r1);
    const float r2 = r1 - __uint_as_float(s.p1 << 16);
    s.p2 = bf16_bits(r2);
    return s;
}

#ifndef RING_DEPTH
#define RING_DEPTH 8
#endif
constexpr int RING = RING_DEPTH;

// MODE 0: f32, 1: bf16x6, 2: bf16x3
template <int MODE>
__global__ __launch_bounds__(256) void k_chain(const float* __restrict__ x_in, const void* __restrict__ wstream, const size_t wave_stride_bytes,
                                               const float* __restrict__ bias, const int depth, float* __restrict__ x_out) {
    constexpr int NPL = MODE == 0 ? 1 : (MODE == 1 ? 3 : 2);          // bf16 planes
    constexpr int LOADS = MODE == 0 ? 16 : 8 * NPL;                    // 16-byte loads per layer and lane
    __shared__ __attribute__((aligned(16))) float lds_s[8 * 64 * 4];                       // f32 exchange: [tile][lane][r]
    __shared__ __attribute__((aligned(16))) unsigned lds_b[3 * 4 * 64 * 4];                // bf16 exchange: [plane][chunk][lane][4 dwords = 8 bf16]
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const int row = blockIdx.x * 16 + j;
    const u32x4* ws = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(wstream) + (size_t)wq * wave_stride_bytes) + lane;
    // activations: XS[4 T + r] = feature 16 T + 4 g + r of row j (every wave holds all 128 of its lanes' share)
    float XS[32];
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) XS[4 * T + r] = x_in[(size_t)row * 128 + 16 * T + 4 * g + r];
    u32x4 XB[NPL][4];                                                   // bf16 modes: [plane][chunk] = 8 bf16 of this lane's B operand
    if constexpr (MODE != 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Split3 s[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = split3(XS[8 * c + e]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                XB[0][c][d] = s[2 * d].p0 | (s[2 * d + 1].p0 << 16);
                XB[1][c][d] = s[2 * d].p1 | (s[2 * d + 1].p1 << 16);
                if constexpr (NPL == 3) XB[2][c][d] = s[2 * d].p2 | (s[2 * d + 1].p2 << 16);
            }
        }
    }
    u32x4 ring[RING];
    size_t pos = 0;
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = ws[(pos + i) * 64];
    f32x4 S0 = {0.f, 0.f, 0.f, 0.f}, S1 = S0;
    for (int l = 0; l < depth; ++l) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        static_assert(LOADS % RING == 0, "ring phase");
#pragma unroll
        for (int q = 0; q < LOADS; ++q) {
            const u32x4 w = ring[q % RING];
#ifndef PROBE_NOLOAD                                                    // (-DPROBE_NOLOAD: the ring is never refilled -- matrix instructions only)
            ring[q % RING] = ws[(pos + q + RING) * 64];
#endif
#ifdef PROBE_NOMFMA                                                     // (-DPROBE_NOMFMA: the loads only; their values are folded into one register)
            acc0[0] += __uint_as_float(w[0] & 1u); (void)XS;
            __builtin_amdgcn_sched_barrier(0);
            continue;
#endif
            if constexpr (MODE == 0) {
                const int ks = 2 * q;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[0]), XS[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[1]), XS[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[2]), XS[ks + 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[3]), XS[ks + 1], acc1, 0, 0, 0);
            } else {
                // stream order: for chunk c, for tile t, for plane p: one 16-byte load = 8 bf16 of the A operand
                const int c = q / (2 * NPL), t = (q / NPL) % 2, p = q % NPL;
                const bf16x8 a = __builtin_bit_cast(bf16x8, w);
                f32x4& acc = t ? acc1 : acc0;
                // plane p of the weights meets planes 0 .. (NPL - 1 - p) of the activations: i + j <= NPL - 1
#pragma unroll
                for (int pb = 0; pb + p < NPL; ++pb)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, XB[pb][c]), acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pos += LOADS;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + (size_t)l * 128 + 32 * wq + 4 * g);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + (size_t)l * 128 + 32 * wq + 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) { S0[r] = siluf_(acc0[r] + b0[r]); S1[r] = siluf_(acc1[r] + b1[r]); }
        if (l + 1 == depth) break;
        if constexpr (MODE == 0) {
            *reinterpret_cast<f32x4*>(&lds_s[((2 * wq) * 64 + lane) * 4]) = S0;
            *reinterpret_cast<f32x4*>(&lds_s[((2 * wq + 1) * 64 + lane) * 4]) = S1;
            lds_barrier();
#pragma unroll
            for (int T = 0; T < 8; ++T) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(&lds_s[(T * 64 + lane) * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) XS[4 * T + r] = x[r];
            }
            lds_barrier();
        } else {
            // the producer splits its own eight outputs (chunk wq of the next layer's K) and publishes the planes
            Split3 s[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r] = split3(S0[r]); s[4 + r] = split3(S1[r]); }
            u32x4 o[3];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                o[0][d] = s[2 * d].p0 | (s[2 * d + 1].p0 << 16);
                o[1][d] = s[2 * d].p1 | (s[2 * d + 1].p1 << 16);
                o[2][d] = s[2 * d].p2 | (s[2 * d + 1].p2 << 16);
            }
#pragma unroll
            for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(&lds_b[((p * 4 + wq) * 64 + lane) * 4]) = o[p];
            lds_barrier();
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c) XB[p][c] = *reinterpret_cast<const u32x4*>(&lds_b[((p * 4 + c) * 64 + lane) * 4]);
            lds_barrier();
        }
    }
    float* o = x_out + (size_t)row * 128 + 32 * wq + 4 * g;
    *reinterpret_cast<f32x4*>(o) = S0;
    *reinterpret_cast<f32x4*>(o + 16) = S1;
}

static unsigned short h_bf16(float x) {
    unsigned u; memcpy(&u, &x, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float h_bf16f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int depth_hi = argc > 1 ? atoi(argv[1]) : 10, depth_lo = 2;
    const int max_wg = 1024;
    srand(3);
    auto rnd = [] { return (rand() % 20001 - 10000) * 1e-4f; };
    std::vector<float> W((size_t)depth_hi * 128 * 128), Bv((size_t)depth_hi * 128), X((size_t)max_wg * 16 * 128);
    for (auto& w : W) w = rnd() * 0.18f;                                 // ~ U(-0.18, 0.18): activations stay O(1) through the chain
    for (auto& b : Bv) b = rnd() * 0.3f;
    for (auto& x : X) x = rnd() * 1.5f;
    // ---- streams.  f32: per layer 16 quads per wave; quad q, lane 16 g + i, element e: k-step ks = 2 q + e / 2, tile e % 2:
    //      W[32 w + 16 (e % 2) + i][16 (ks / 4) + 4 g + ks % 4]  (pf_host.cpp: pack_n16_raw)
    const size_t f32_stride = ((size_t)depth_hi * 16 + 2 * RING) * 1024;
    std::vector<float> sf(4 * f32_stride / 4, 0.f);
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < depth_hi; ++l)
            for (int q = 0; q < 16; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int g = lane >> 4, i = lane & 15, ks = 2 * q + e / 2, f = 16 * (ks / 4) + 4 * g + ks % 4;
                        sf[(w * f32_stride) / 4 + ((size_t)(l * 16 + q) * 64 + lane) * 4 + e] = W[((size_t)l * 128 + 32 * w + 16 * (e % 2) + i) * 128 + f];
                    }
    // bf16: per layer, chunk c, tile t, plane p: lane 16 g + i holds 8 bf16, element e <-> input feature 16 (2 c + e / 4) + 4 g + e % 4
    auto build_bf = [&](int npl, std::vector<unsigned short>& out, size_t& stride) {
        stride = ((size_t)depth_hi * 8 * npl + 2 * RING) * 1024;
        out.assign(4 * stride / 2, 0);
        for (int w = 0; w < 4; ++w)
            for (int l = 0; l < depth_hi; ++l)
                for (int c = 0; c < 4; ++c)
                    for (int t = 0; t < 2; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int g = lane >> 4, i = lane & 15, f = 16 * (2 * c + e / 4) + 4 * g + e % 4;
                                float x = W[((size_t)l * 128 + 32 * w + 16 * t + i) * 128 + f];
                                for (int p = 0; p < npl; ++p) {
                                    const unsigned short b = h_bf16(x);
                                    const size_t q = (size_t)l * 8 * npl + (size_t)(c * 2 + t) * npl + p;
                                    out[(w * stride) / 2 + (q * 64 + lane) * 8 + e] = b;
                                    x -= h_bf16f(b);
                                }
                            }
    };
    std::vector<unsigned short> s6, s3;
    size_t st6 = 0, st3 = 0;
    build_bf(3, s6, st6); build_bf(2, s3, st3);
    float *dX, *dB, *dO, *dflush;
    void *dSf, *dS6, *dS3;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dB, Bv.size() * 4)); CK(hipMalloc(&dO, X.size() * 4)); CK(hipMalloc(&dflush, 1u << 28));
    CK(hipMalloc(&dSf, sf.size() * 4)); CK(hipMalloc(&dS6, s6.size() * 2)); CK(hipMalloc(&dS3, s3.size() * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bv.data(), Bv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dSf, sf.data(), sf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dS6, s6.data(), s6.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dS3, s3.data(), s3.size() * 2, hipMemcpyHostToDevice));
    // ---- fp64 reference of the depth_hi chain on the first 64 rows
    const int nref = 64;
    std::vector<double> ref((size_t)nref * 128), cur((size_t)nref * 128);
    for (int r = 0; r < nref; ++r) for (int f = 0; f < 128; ++f) cur[(size_t)r * 128 + f] = X[(size_t)r * 128 + f];
    for (int l = 0; l < depth_hi; ++l) {
        for (int r = 0; r < nref; ++r)
            for (int o = 0; o < 128; ++o) {
                double a = Bv[(size_t)l * 128 + o];
                for (int f = 0; f < 128; ++f) a += (double)W[((size_t)l * 128 + o) * 128 + f] * cur[(size_t)r * 128 + f];
                ref[(size_t)r * 128 + o] = a / (1.0 + exp(-a));
            }
        cur = ref;
    }
    double refmax = 0;
    for (double v : ref) refmax = std::max(refmax, std::fabs(v));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"f32   ", "bf16x6", "bf16x3"};
    std::vector<float> out((size_t)nref * 128);
    std::vector<float> keep[3];
    for (int mode = 0; mode < 3; ++mode) {
        const void* st = mode == 0 ? dSf : (mode == 1 ? dS6 : dS3);
        const size_t stride = mode == 0 ? f32_stride : (mode == 1 ? st6 : st3);
        auto launch = [&](int nwg, int depth) {
            if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(nwg), dim3(256), 0, 0, dX, st, stride, dB, depth, dO);
            else if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(nwg), dim3(256), 0, 0, dX, st, stride, dB, depth, dO);
            else hipLaunchKernelGGL(k_chain<2>, dim3(nwg), dim3(256), 0, 0, dX, st, stride, dB, depth, dO);
        };
        launch(nref / 16, depth_hi);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
        keep[mode] = out;
        double emax = 0, erel = 0;
        for (size_t i = 0; i < out.size(); ++i) {
            const double d = std::fabs((double)out[i] - ref[i]);
            emax = std::max(emax, d);
            erel = std::max(erel, d / std::max(std::fabs(ref[i]), 1e-3 * refmax));
        }
        double e_vs_f32 = 0;
        if (mode) for (size_t i = 0; i < out.size(); ++i) e_vs_f32 = std::max(e_vs_f32, std::fabs((double)out[i] - keep[0][i]));
        printf("%s: %d-layer chain vs fp64: max |err| %.3e (/ max|ref| %.3e = %.2e), max rel err %.2e", names[mode], depth_hi, emax, refmax, emax / refmax, erel);
        if (mode) printf("; vs the f32 form max |diff| %.3e (%.2e of max|ref|)", e_vs_f32, e_vs_f32 / refmax);
        printf("\n");
        for (int nwg : {1, 256, 512, 1024}) {
            double t[2];
            for (int k = 0; k < 2; ++k) {
                const int depth = k ? depth_hi : depth_lo;
                launch(nwg, depth);                                         // warm
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int rep = 0; rep < 20; ++rep) launch(nwg, depth);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t[k] = ms / 20 * 1e3;
            }
            printf("   %s wgs %4d: %2d layers %.2f us, %2d layers %.2f us -> %.3f us per layer\n", names[mode], nwg, depth_lo, t[0], depth_hi, t[1],
                   (t[1] - t[0]) / (depth_hi - depth_lo));
        }
    }
    return 0;
}
