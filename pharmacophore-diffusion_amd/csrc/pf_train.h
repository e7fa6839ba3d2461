// pf_train.h -- host/device structures of the gradient path of libpfdyn (gfx950 only).
//
// The training step keeps, per conv layer, the node state that entered the layer (h [N][128], v [N][48]) and the
// per-(tile, destination) message partial rows the forward edge kernel wrote; every other activation is
// recomputed inside the backward kernels (reference: PharmacophoreDiff.forward, pharmacodiff.py:162-243; autograd
// through PharmRecDynamicsGVP.forward, dynamics_gvp.py:131-185, and GVPMultiEdgeConv, gvp.py:459-551).
//
// Parameters and gradients live in ONE flat fp32 vector in the reference's state-dict order
// (pf_host.cpp: expected_tensors); a GvpT addresses one GVP's six tensors by offset into that vector, so the
// gradient of a tensor sits at the same offset of the gradient vector.  Every thread block of a backward kernel
// accumulates into its own private copy of the gradient vector (gpart[block][nparams], plain read-modify-write by
// the owning lane, no atomics) and pfk_train_reduce sums the copies in a fixed order.  The scatter of dL/d(h_src, v_src)
// from the edges of level 0 to their source nodes (a node's out-edges live in many tiles) uses atomics on 64-bit
// fixed-point accumulators, whose result is independent of the arrival order: repeated backward passes agree bit for bit.
#pragma once
#include <stdint.h>
#include "pf_device.h"

#define PFT_MAX_CHAIN 4      // GVPs per chain the backward tiles hold in LDS
#define PFT_ROWS 16          // rows (edges / nodes) per backward sub-tile = N of v_mfma_f32_16x16x4_f32
#define PFT_WPACK_FLOATS (11 * 8 * 64 * 4)   // packed to_feats_out of one message GVP (k_pack_gvp: one table per product direction)
#define PFT_FIX_BITS 40      // fixed-point scale of the level-0 scatter RELATIVE to the largest upstream gradient of the call:
                             // resolution 2^-40 of it, head room 2^23 times it (pfk_fix_scale picks the power of two)

struct GvpT {
    int o_Wh, o_Wu, o_Wm, o_bm, o_Wg, o_bg;   // offsets of Wh [vi][h], Wu [h][vo], to_feats_out.0.{weight [so][si+h], bias},
                                              // scalar_to_vector_gates.{weight [vo][so], bias}
    int vi, h, vo, si, so, sig;               // sig: vector activation is a sigmoid (0: identity, last head GVP)
    int pk;                                   // index of this GVP in the packed fragment tables (k_pack_gvp)
};

// Which blocks' gradient copies hold a tensor's gradient is a function of the kernel that differentiates it (its "class"):
// head, encoders and node kernels use the blocks [0, grid) of their launch and clear their tensors in their own copy when they
// start; an edge-level launch deals its blocks over the etypes (k_bwd_edge_level) and every block STORES the whole GVP it
// served.  Nothing is cleared globally and pfk_train_reduce adds up exactly the copies that were written.
#define PFT_ENC_BLOCKS 1024
#define PFT_CLS_HEAD 0
#define PFT_CLS_ENC 1
#define PFT_CLS_NODE 2       // + layer
#define PFT_CLS_MSG 8        // + layer * 4 + etype
#define PFT_CLS_NONE (-1)    // empty tensors
struct TensorSeg { int begin, end, cls, pad; };
struct ReduceParams {
    const float* gpart; int nparams; float* grad;
    int gstride;             // floats between consecutive gradient copies (nparams rounded up to a multiple of 64: 16-byte aligned rows)
    const float* gpart_enc; int enc_begin, enc_n;
    const TensorSeg* tseg; int ntens;
    int NB;                  // gradient copies = grid of the edge-level launches
    int head_grid, enc_grid, node_grid[4];
    int n_et[4];             // etypes that took part in layer l's edge-level launches
    const int* ccnt;         // [layer][16]: passes per etype (k_compact_rows), rows per etype at + 8
    unsigned cls_mask;       // k_train_reduce: the classes this launch sums (bit = class number); the encoders' bit runs k_reduce_enc
};

struct TrainCommon {
    const float* W;          // flat parameters
    float* gpart;            // [gridDim.x][gstride]
    int nparams;
    int gstride;             // floats between consecutive copies (nparams rounded up to a multiple of 64)
    const TensorSeg* tseg; int ntens;
    float* gpart_enc; int enc_begin, enc_n;  // the encoders' parameters [enc_begin, enc_begin + enc_n) have their own, narrow gradient copies
                                             // [PFT_ENC_BLOCKS][enc_n]: k_bwd_encode is latency-bound per tile and runs several blocks per CU
    const float* wpack_b; const float* wpack_f;   // k_pack_gvp tables (PFT_WPACK_FLOATS per GVP each)
    uint32_t drop_thr;       // an element is dropped iff pf_drop_hash(...) < drop_thr  (= p * 2^32; 0: no dropout)
    float drop_scale;        // 1 / (1 - p)
    uint32_t seed;           // dropout stream of this step
    const float* mask_override;   // tests: multipliers [n_convs * 2 streams][N * 144] used instead of the hash, or NULL
    int mask_N;
    int bf16;                // bf16 leg (pf_train_set_precision): to_feats_out / gate products of the gradient kernels on bf16 matrix
                             // instructions (operands rounded to nearest even, fp32 accumulation); everything else stays fp32
};

struct BwdHeadParams {
    TrainCommon c;
    const NodeTile* tiles; int ntiles;       // pharm tiles (32 rows each)
    int node_base;
    const float* h; const float* v;          // output of the last conv layer
    const GvpT* g; int n_gvps;               // device table [n_gvps]
    int o_Wout, o_bout, pharm_nf;
    const float* g_eps_h; const float* g_eps_x;   // upstream gradients [Nf][pharm_nf], [Nf][3]
    float* G_h; float* G_v;                  // gradient w.r.t. the last layer's output (rows of the pharm nodes are stored)
    // what the training forward of the head left per level and pharm row (k_rg_unit<SAVE>): [level][Nf][128 / 16 / 48]; NULL: the
    // kernel recomputes the chain
    const float* sv_z; const float* sv_g; const float* sv_v; size_t sv_stride;
};

struct BwdNodeParams {
    TrainCommon c;
    const NodeTile* tiles; int ntiles;
    const int* ulist; const int* ucnt;       // k_compact_node_rows: per node type (pharm at 0, prot at ucap entries) the valid rows of the
    int ucap;                                // tile table, densely, as (node id, row in the saved levels); ucnt[1], ucnt[2] = their numbers
    const int* in_start; const int* in_cnt; int N;
    int pp_slot;                             // 1: all pp in-edges; 2: compact copy for the active atoms (pruned layer)
    const int* row_ids; const int* dyn_cnt;  // active-atom lists (tiles with ids != 0) and their lengths
    const float* msg_s; const float* msg_v; int zero_row;
    const float* h_in; const float* v_in;    // layer input (v_in unused when l0)
    const float* G_h_out; const float* G_v_out;   // gradient w.r.t. the layer output
    float* G_h_in; float* G_v_in;            // gradient w.r.t. the layer input: the residual path is STORED here (edge kernel adds)
    float* gagg_s; float* gagg_v;            // gradient w.r.t. the aggregated message before normalisation [N][128], [N][48]
    const int* gid; const float* gnorm; int B;
    int norm_mode; float norm_value;
    const GvpT* upd; int n_upd;              // device table [2 ntypes][n_upd]
    int o_ln[2][4];                          // ln1_w ln1_b ln2_w ln2_b per node type
    int layer, l0;
    int grp;                                 // slots per partial-row group of the forward's edge kernel (32: tile kernel; 4 / 8: row groups)
    // update-chain levels left by the training forward (k_rg_node<., SAVE>; NodeParams::sv_*): [level][2 N][128 / 16 / 48]; NULL: recompute
    const float* sv_z; const float* sv_g; const float* sv_v; size_t sv_stride;
};

struct BwdEdgeParams {
    TrainCommon c;
    const EdgeTile* tiles; int ntiles;
    const int* dyn_cnt;
    const int* esrc; const int* edst;
    const float4* xn;
    const float* h; const float* v;          // layer input
    const float* gagg_s; const float* gagg_v;
    const int* in_cnt; int N;
    int norm_mode;
    float* G_h_in; float* G_v_in;            // atomically accumulated
    const GvpT* g; int n_gvps;               // device table [4 etypes][n_gvps]
    float rbf_mu[PF_R]; float rbf_inv_sigma;
    int l0;
};

// level-by-level backward of the message chains (k_bwd_edge_level): one launch per GVP level, last level first.
// The training forward (k_edge_msg<.., SAVE>) left per (level, edge slot) the pre-activation scalars Z, the gate
// pre-activations and the gated output vectors, so nothing of the chain is recomputed but the small vector products.
// A block serves ONE etype (the blocks are dealt to the etypes in proportion to their non-empty tiles, which the block
// walks with a stride): its to_feats_out weight sits in LDS for the whole launch and the weight gradients of
// to_feats_out and of the gates accumulate in registers across all its tiles.
struct BwdEdgeLevelParams {
    TrainCommon c;
    const EdgeTile* tiles;
    int et_tile0[5]; int n_et;               // etype segments of the tile table; etypes [0, n_et) take part
    const int* clist; const int* ccnt;       // k_compact_rows: per etype segment (offset 32 x its first tile) the slots of the valid
                                             // rows, densely, in table order; ccnt[et] = passes of 32 rows, ccnt[8 + et] = rows
    const int* dyn_cnt;
    const int* esrc; const int* edst;
    const float4* xn;
    const float* h; const float* v;          // layer input (level 0 gathers the source rows)
    const float* gagg_s; const float* gagg_v;   // upstream of the last level
    const int* in_cnt; int N;
    int pp_slot;                             // in_cnt slot of the pp tiles (2 in the pruned layer)
    int norm_mode;
    float* G_h_in; float* G_v_in;            // gradient w.r.t. the layer input (the node kernel stored the residual path)
    // level 0 scatters dL/d(h_src, v_src) of every edge to its source node.  A node's out-edges live in many tiles, so
    // the sums are atomic -- on 64-bit FIXED-POINT accumulators (value * 2^k, rounded to nearest; k from the call's upstream gradients): integer
    // addition is associative, so the result does not depend on the order in which the blocks arrive and the gradients
    // are bitwise reproducible.  pfk_fix_apply adds the sums to G_h_in / G_v_in and clears the accumulators.
    long long* A_h; long long* A_v;          // [N][128], [N][48]
    const float* fix;                        // device: [scale = 2^k, 1 / scale] of this backward call (pfk_fix_scale)
    const float* sv_z; const float* sv_g; const float* sv_v; size_t sv_stride;
    float* gs_buf; float* gv_buf;            // dL/d(input scalars / vectors of the level above), per edge slot
    const GvpT* g; int n_gvps; int level;
    int fx;                                  // shape class of this level's GVPs (k_bwd_edge_level's FX), 0: generic
    const float* wpack;                      // k_pack_gvp input-gradient fragments of this layer's message GVPs [et][level][PFT_WPACK_FLOATS]
    float rbf_mu[PF_R]; float rbf_inv_sigma;
    int l0;
};

struct BwdEncodeParams {
    TrainCommon c;
    int Np, Nf;
    const float* prot_h0; const float* pharm_h; const float* t; const int* gid;
    int rec_nf, pharm_nf;
    int o_w[2], o_b[2], o_lw[2], o_lb[2];    // 0 prot, 1 pharm
    const float* G_h;                        // gradient w.r.t. the encoder output [N][128]
    int B;
    const float* Gg;                         // k_enc_group: G_h summed per (graph, element) [B][rec_nf][128], or NULL (rec_nf > 16)
    const int* onehot_flag;                  // device: 0 iff every protein feature row is an element one-hot (k_l0_types at bind time)
};

// the loss around the dynamics (k_loss_prepare / k_loss_eval, pf_train_loss_forward)
// the loss's unit gradients (gx [nx], gh [nh]) times their upstream scalars a (+ a2), b (+ b2): device scalars, a2 / b2 may be null
struct ScaleArgs { float* gx; int nx; const float* a; const float* a2; float* gh; int nh; const float* b; const float* b2; };

struct LossParams {
    int B, Np, Nf, nf, T, remove_com, weighted;
    float feat_norm;
    const int* prot_ptr; const int* pharm_ptr; const int* gid;
    const float* prot_x0;                    // [Np][3] as bound
    const float* x0; const float* h0;        // clean centers [Nf][3], raw feature one-hots [Nf][nf]
    const int* t_int; const float* eps_x; const float* eps_h;
    const float* alpha_tab; const float* sigma_tab;
    float4* xn; float* pharm_h; float* t;    // the dynamics' input state
    float* x0c; float* alpha_g; float* sigma_g;
    const float* dyn_h; const float* dyn_x;  // the dynamics' outputs
    float* g_x; float* g_h; float* out;      // unit upstream gradients, [6] losses and metrics
    float* part; int* ticket;                // k_loss_eval: [blocks][8] partial sums; arrival counter (zero between launches)
};

// dropout keep-mask of (step seed, stream, element): identical in the forward node kernel and the backward pass.
// stream = layer * 2 + (0: message dropout, 1: residual dropout); element = node * 144 + (feature | 128 + channel).
#if defined(__HIPCC__)
__device__ __forceinline__
#else
static inline
#endif
uint32_t pf_drop_hash(uint32_t seed, uint32_t stream, uint32_t elem) {
    uint32_t x = seed ^ (stream * 0x9E3779B1u) ^ (elem * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    x += elem; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x;
}
