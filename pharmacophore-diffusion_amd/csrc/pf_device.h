// pf_device.h -- shared host/device structures of libpfdyn (gfx950 only).
//
// Data layout in HBM (all fp32 / int32, sized by pf_set_pocket_batch):
//   node ids are GLOBAL: protein atoms [0, Np) first, pharmacophore centers [Np, Np+Nf).
//   xn    float4[N]        current coordinates (w unused)
//   h[2]  float[N][128]    node scalars, ping-pong between conv layers
//   v[2]  float[N][48]     node vectors (16 x 3), ping-pong
//   edge slots are GLOBAL: [0,Epp) static pp edges sorted by destination, then one fixed-capacity
//   region per (dynamic etype, graph); each destination's in-edges are contiguous (dst-major),
//   so a node finds its messages as two ranges (in_start[slot][n], in_cnt[slot][n]).
//   msg_s float[Ecap][128], msg_v float[Ecap][48]   per-edge messages of the current layer.
//
// Register layout used by every MFMA stage ("F-layout", row-on-lane):
//   a tile is 32 rows (edges or nodes); row j lives on lanes j and j+32 (half hl = lane>>5).
//   A 128-feature vector is split over the two lanes as 64 registers:
//       reg q = 16*mt + r   <->   feature 32*mt + rho(r,hl),  rho(r,hl) = (r&3) + 8*(r>>2) + 4*hl
//   which is exactly the C/D layout of v_mfma_f32_32x32x2_f32 with the row on the lane
//   (Out^T[feature][row] = W[feature][k] * In^T[k][row]); an output tile is therefore directly
//   the B operand of the next layer's k-steps and no activation ever goes through LDS.
//   A 16-channel vector feature [16][3] is split the same way ("R-layout"): channel
//   u(t,hl) = (t&3) + 8*(t>>2) + 4*hl sits in register t (0..7) of lane half hl, one register
//   array per coordinate -- rows 0..15 of a C/D fragment -- so the vector channel (Wh, Wu) runs
//   on the matrix cores too and the two lanes of a row never exchange data explicitly.
//   Weights are pre-packed on the host in A-operand fragment order (pf_host.cpp: pack_linear).
#pragma once
#include <stdint.h>

#define PF_S 128          // n_hidden_scalars (compile-time specialisation)
#define PF_V 16           // vector_size
#define PF_R 16           // rbf_dim
#define PF_MAXF 64        // max pharmacophore centers per graph
#define PF_MAXK 16        // max k for kNN edges
#define PF_MAX_GVPS 8

enum { ET_FF = 0, ET_PF = 1, ET_FP = 2, ET_PP = 3 };

// Pointers that the device code reads out of memory (not straight out of the kernel-argument
// segment) carry the global address space explicitly: otherwise hipcc treats them as generic
// ("flat") pointers, which rules out scalar loads (s_load) for wave-uniform weights.
#if defined(__HIP_DEVICE_COMPILE__)
#define PF_AS1 __attribute__((address_space(1)))
#else
#define PF_AS1
#endif
typedef const float PF_AS1* pf_gcf;

struct GvpW {              // packed weights of one GVP (device pointers)
    pf_gcf a_wh;           // [8(+1)][64 lanes]      A fragments of Wh^T (vector channel, R-layout k order)
    pf_gcf a_wu;           // [8(+1)][64 lanes]      A fragments of Wu^T
    pf_gcf a_main;         // [NKS][64 lanes][NMO]   A fragments of to_feats_out
    pf_gcf a_main_c;       // [NMO][NKS/4][64 lanes][4]  the same fragments per output tile, four k-steps per lane (4-wave kernels)
    pf_gcf a_gate_c;       // [NMO][4][64 lanes][4]      gate fragments of the wave that owns output tile mo
    pf_gcf a_wh_c;         // [3][64 lanes][4]           a_wh / a_wu, four k-steps per lane (zero padded to 12)
    pf_gcf a_wu_c;
    pf_gcf b_main;         // [2 halves][NMO*16]     bias in F-layout
    pf_gcf a_gate;         // [NMO*16][64 lanes]     A fragments of scalar_to_vector_gates (rows 0..VO-1)
    pf_gcf b_gate;         // [2 halves][8]          gate bias in R-layout
};

// ---- row-group ("rg") kernels: 4 rows per v_mfma_f32_4x4x1_16b_f32, weights streamed as "quads" (pf_rg.hip) ----
// A chain's weights are one contiguous stream of quads ([quad][64 lanes][4 floats] = 1 KiB, one b128 load per lane) in
// the exact order the kernel consumes them, so the register prefetch ring (12-24 quads deep) runs across GVP
// boundaries.  The chain is software-pipelined: the scalar->vector gates of GVP n-1 (which need only its SiLU output)
// and the vector products of GVP n are threaded through the 128 main k-steps of GVP n, whose MFMAs hide the LDS
// round trips in between.  Block of GVP n (vi, nextra, nh = output halves of 64, prev = a gate is pending):
//   [const] [xhat, w16 when vi == 17] [main m=0] [gate of GVP n-1, 8 quads, when prev] [main m=1,2] [Vh x4]
//   [main m=3..5] [Vu x4] [main m=6,7] [rbf when nextra] [sh] [pad]
// and a chain ends with a flush block [gate of the last GVP x8] [pad].  Every block is padded to a multiple of RG_PAD
// quads, a multiple of every ring depth used, so a quad's ring slot is a compile-time constant.
#define RG_PAD 24
#define RG_TAIL_PAD 48      // quads of read-ahead padding behind the last stream (>= the deepest prefetch ring)
#define RG_NQ_FLUSH 24      // [8 gate quads] [pad]
#define RG_NQ_OUT 24        // to_scalar_output: [const] [8 gate-like quads] [pad]
struct RgSched {
    int q_c, q_xh, q_a, q_gate, q_b, q_vh, q_cc, q_vu, q_d, q_rbf, q_sh, nq_raw, nq;
};
constexpr RgSched rg_sched(const int vi, const int nextra, const int nh, const bool prev) {
    RgSched s{};
    s.q_c = 0;
    s.q_xh = 1;
    s.q_a = 1 + (vi == 17 ? 2 : 0);
    s.q_gate = s.q_a + 4 * nh;
    s.q_b = s.q_gate + (prev ? 8 : 0);
    s.q_vh = s.q_b + 8 * nh;
    s.q_cc = s.q_vh + 4;
    s.q_vu = s.q_cc + 12 * nh;
    s.q_d = s.q_vu + 4;
    s.q_rbf = s.q_d + 8 * nh;
    s.q_sh = s.q_rbf + (nextra ? 4 * nh : 0);
    s.nq_raw = s.q_sh + 4 * nh;
    s.nq = (s.nq_raw + RG_PAD - 1) / RG_PAD * RG_PAD;
    return s;
}
// quad of main k-step group k (k = (m * 4 + aq) * nh + half: image j of it is k-step (m, a = 4 aq + j))
constexpr int rg_main_quad(const RgSched& s, const int nh, const int k) {
    return k < 4 * nh ? s.q_a + k : (k < 12 * nh ? s.q_b + k - 4 * nh : (k < 24 * nh ? s.q_cc + k - 12 * nh : s.q_d + k - 24 * nh));
}

// ---- "n16" kernels (pf_n16.hip): an item is 16 rows on the four waves of a workgroup, v_mfma_f32_16x16x4_f32 ----
// Wave w owns output features [32 w, 32 w + 32) of every 128-output scalar Linear (two 16 x 16 tiles) and streams only
// that quarter of its weights; waves 0..2 own one coordinate of the vector channel each.  The activations meet in LDS
// once per GVP.  Every wave has its own quad stream per chain (wave w's stream n16_stride floats after wave w - 1's);
// a block is one GVP, quads in consumption order (a quad = [64 lanes][4 images], image = one lane's share of an A
// operand: lane 16 g + i <-> output row i, k = 4 ks + g of the k-step):
//   GEN  [main x VH_AT] [vh] [main x (16 - VH_AT)] [sh x2] [vu] [bias x2] [gate x2]     = 24   (128 + 16 -> 128 + 16)
//   M0F  [x1] [vh] [w16] [main x16] [rbf x2] [sh x2] [vu] [bias x2] [gate x2]           = 28   first message GVP
//   M0Z  [x1] [xh] [main x16] [rbf x2] [sh x2] [vu] [bias x2] [gate x2]                 = 27   ... when v_src == 0 (conv layer 0)
//   M0H  [x1] [xh] [rbf x2] [sh x2] [vu] [gate x2]                                      = 9    ... and h_src is a row of a type table
// (GEN: the vector product Vh = Wh^T V sits BEHIND the first main k-steps.  Its operand is the previous GVP's gated output --
// sigmoid(gate sums from LDS) x Vu -- and a wave issues in order: in front of the main k-steps it made every block of a chain
// wait for an LDS round trip, four sigmoids and four dependent matrix instructions before the first of its 64 main ones.)
// Scalar k-step ks (0..31), lane group g <-> input feature 16 (ks >> 2) + 4 g + (ks & 3): the D fragment of tile T = 2 w + t
// (lane 16 g + j, register r: feature 16 T + 4 g + r of row j) is the B operand of k-step 4 T + r without any data movement.
// -DN16_SPLIT=1 (the variant library libpfdyn_split.so, Makefile; never the default build): the 128-input scalar Linear of every
// block -- its "main" k-steps, 64 of a GEN block's 88 matrix instructions -- runs on v_mfma_f32_16x16x32_bf16 with both operands
// split into three bf16 planes (x = x0 + x1 + x2, 8 mantissa bits each: 24 together) and the six products with i + j <= 2
// accumulated in fp32: 48 instructions of half the issue time, the rounding of an fp32 product within a factor ~1 (measured on a
// ten-layer chain against fp64: 4.0e-7 of the largest output, the exact-fp32 form 5.0e-7: tools/probes/split_bf16_chain.hip,
// profiles/r05/split_bf16_chain.txt).  A main quad is then one 16-byte load = 8 bf16 of ONE plane: quad m <-> chunk m / 6 of K
// (32 inputs = tiles 2 c, 2 c + 1), output tile (m / 3) % 2, plane m % 3; element e of lane 16 g + i <-> input 16 (2 c + e / 4) +
// 4 g + e % 4 -- the order in which the lane's own D fragments of the previous block line up as the B operand.  Weight bytes of the
// main run x 1.5.  Everything else of a block (vector channel, rbf / sh k-steps, gates) stays on the exact-fp32 instructions.
#ifndef N16_SPLIT
#define N16_SPLIT 0
#endif
#define N16_NQM (N16_SPLIT ? 24 : 16)       // quads of a block's main run
#ifndef N16_D
#define N16_D (N16_SPLIT ? 16 : 12)         // depth of the register prefetch ring in quads (GEN blocks are a multiple of it)
#endif
#ifndef N16_VH_AT
#define N16_VH_AT (N16_SPLIT ? 6 : 4)       // GEN blocks: main quads in front of the vh quad (0: the vh quad leads the block)
#endif
#define N16_TAIL_PAD 24     // quads of read-ahead padding behind every wave's stream
enum { N16_GEN = 0, N16_M0F = 1, N16_M0Z = 2, N16_M0H = 3 };
struct N16Sched {
    int q_x1, q_vh, q_w16, q_main, q_rbf, q_sh, q_vu, q_b, q_gate, nq;
    int vh_at;             // main quads [0, vh_at) sit in front of q_vh, the others behind it (GEN; 16 elsewhere: q_vh is outside the main run)
    // stream position of main quad m (0 .. N16_NQM - 1), and the inverse (-1: quad qi is not a main quad)
    constexpr int main_pos(const int m) const { return q_main + m + ((q_vh >= q_main && m >= vh_at) ? 1 : 0); }
    constexpr int main_of(const int qi) const {
        if (q_main < 0 || qi < q_main || qi == q_vh) return -1;
        const int m = qi - q_main - ((q_vh >= q_main && qi > q_vh) ? 1 : 0);
        return m < N16_NQM ? m : -1;
    }
};
constexpr N16Sched n16_sched(const int kind) {
    N16Sched s{};
    int q = 0;
    const bool m0 = kind != N16_GEN;
    s.vh_at = N16_NQM;
    s.q_x1 = m0 ? q++ : -1;
    if (kind == N16_GEN) { s.q_main = 0; s.vh_at = N16_VH_AT; s.q_vh = N16_VH_AT; s.q_w16 = -1; q = N16_NQM + 1; }
    else {
        s.q_vh = q++;
        s.q_w16 = kind == N16_M0F ? q++ : -1;
        s.q_main = kind != N16_M0H ? q : -1;
        if (kind != N16_M0H) q += N16_NQM;
    }
    s.q_rbf = m0 ? q : -1;
    if (m0) q += 2;
    s.q_sh = q; q += 2;
    s.q_vu = q++;
    s.q_b = kind != N16_M0H ? q : -1;
    if (kind != N16_M0H) q += 2;
    s.q_gate = q; q += 2;
    s.nq = q;
    return s;
}
static_assert(n16_sched(N16_GEN).nq % N16_D == 0, "GEN blocks must keep the ring phase");

struct EdgeTile {          // one wave = 32 edge slots
    int e0;                // first edge slot
    int n;                 // slots in this tile (<= 32)
    int et;                // etype
    int cnt_idx;           // index into dyn_cnt (et*B+g) or -1 for static tiles
    int rel;               // e0 - region start (dynamic tiles)
};

struct NodeTile {          // 32 nodes of one type
    int n0;                // first global node id -- or first position in the active-atom list (ids != 0)
    int n;                 // rows in this tile (<= 32)
    int ntype;             // 0 prot, 1 pharm
    int cnt_idx;           // index into dyn_cnt of the list length, or -1 for static tiles
    int rel;               // position of the tile inside its list
    int ids;               // rows are positions in row_ids (active-atom list) instead of node ids
};

struct EdgeParams {
    const EdgeTile* tiles;
    int ntiles;
    const int* dyn_cnt;
    const int* esrc;
    const int* edst;
    const float4* xn;
    const float* h;        // [N][128]
    const float* v;        // [N][48]
    float* msg_s;
    float* msg_v;
    const GvpW* w;         // [4 etypes][n_gvps]
    const float* pre;      // [Np][128] P = W_msg0[:, :128] h + b for the pp etype of this layer, or NULL
    int n_gvps;
    float rbf_mu[PF_R];
    float rbf_inv_sigma;
    // training forward (one-wave kernel only): per message-GVP level l and edge slot e, Z (pre-activation scalars),
    // the gate pre-activations and the gated output vectors go to sv_*[(l * sv_stride + e)]; NULL: inference
    float* sv_z; float* sv_g; float* sv_v; size_t sv_stride;
    int bf16;              // training forward of the bf16 leg (k_edge_msg<., SAVE, BF16>): dense Linears on bf16 matrix instructions
    pf_gcf rg[4];          // row-group kernels: quad stream of each etype's message chain (this layer)
    // row-group kernels, compact work list (nreg > 0): the launch covers regions [0, nreg) of reg / dyn_cnt (region
    // r = kind * regB + graph; kinds ff, pf, fp, pa); ngroups4 / ngroups8 = capacity in groups of 4 / 8 slots (the grid)
    const int* reg; int nreg, regB, ngroups4, ngroups8;
    pf_gcf rgs[4]; int rgs_stride;   // two-wave form: wave 0's stream of each chain; wave 1's follows rgs_stride floats later
    // conv layer 0, static pp edges (row-group kernels, "static hoist"): the first message GVP of a pp edge
    // reads h_src = encoder(element one-hot, t) and v_src = 0, and the protein moves rigidly, so its scalar
    // pre-activation is zs[eo] (bias + rbf + sh terms: constant over the trajectory) + ptab[type(src)] (the h_src block
    // of the Linear applied to the encoder output of each element type at this t), and its Vu is xhat (x) weff.
    // zs == NULL: off.  eorig[e]: static pp slot of edge slot e (identity below Epp; set by the edge build for "pa").
    const float* zs;       // [Epp][128]
    const float* ptab;     // [graph or 1][ntypes][128]
    int ptab_gstride;      // floats between the tables of consecutive graphs (0: every graph is at the same t)
    const int* ptype;      // [Np] element type of each protein atom
    const int* eorig;      // [Ecap]
    const int* l0_gid;     // graph of each node (ptab_gstride != 0)
    const float* l0c;      // [16 weff][16 gate bias]
    int ngroups_sel;       // compact work list: grid in groups when the regions' group sizes differ by kind (0: ngroups4/8)
    const int* need; int need_stamp;   // pocket sharing: a kind-3 item runs only if one of its destinations carries the stamp
    int pa_abs;            // pocket sharing: the kind-3 regions are ranges of STATIC slots with arbitrary starts; their
                           // groups are cut on absolute multiples of the group size (what the node kernel's e | (grp - 1) expects)
    // n16 kernels (pf_n16.hip): wave 0's quad stream of each etype's message chain; wave w's n16_stride[et] floats further
    pf_gcf n16[4]; int n16_stride[4];
    // conv layer 0, every graph's ff / pf / fp region of one capacity (k_n16_edge_u): first region start per etype, strides
    // (ff | pf << 16), stride_fp | groups_ff << 16 | groups_pf << 19 | groups_fp << 22 | (B - 1) << 25, capacity of the "pa" regions in 16-slot groups;
    // uni_s2g == 0: off
    int uni_base[3], uni_s01, uni_s2g, uni_pa_groups;
    int ptab16_off[4];     // conv layer 0: float offset of the etype's type table (bias folded in) inside ptab's slot, or -1
    // conv layer 0, center hoist (CenHoistParams): P rows of the ff ([0]) and fp ([1]) etypes, [2][pcen_nf][128] -- W_et[:, :128] h_c + b_et
    // for every center c, left by the previous denoising step -- or NULL: the items of those etypes encode their source center
    // themselves (n16 kind M0Z)
    const float* pcen; int pcen_nf;
    // conv layer 0 inside a sampling run: the protein moves rigidly, so a pp edge's geometry comes from the batch's ORIGINAL coordinates
    // ([Np][3], never written) instead of the per-step shifted copy -- the same bits in every step, and no race with the update + build
    // that shifts xn while speculative pp items of the next call read it.  NULL: xn (pf_dynamics_forward with other coordinates).
    const float* x0_static;
    // [B] or NULL: per graph, the number of leading 16-slot groups of the "pa" region this launch skips -- their partial rows were
    // computed ahead (the previous step's last launch) and that step's build found the region unchanged up to there (BuildParams::pa_same)
    const int* pa_skip;
};

// static-hoist source block in the packed weights (pure copies of the first pp message GVP of conv layer 0 and of the
// protein encoder's consumers: the gather map of pf_set_flat_params covers it); offsets in floats
#define L0H_WR 0           // [16 rbf][128]     to_feats_out columns 128..143, k-major
#define L0H_WSH 2048       // [17][128]         to_feats_out columns 144..160 (sh), k-major
#define L0H_B 4224         // [128]             to_feats_out bias
#define L0H_WH0 4352       // [17] (+pad)       Wh row 0 (the unit x_diff channel)
#define L0H_WU 4384        // [17][16]          Wu
#define L0H_BG 4672        // [16]              gate bias
#define L0H_WHT 4736       // [128 k][128 f]    to_feats_out columns 0..127 (h_src), k-major
#define L0H_WHT_PF 21120   // [128 k][128 f]    the pf etype's first message GVP: to_feats_out columns 0..127, k-major
#define L0H_B_PF 37504     // [128]             ... its bias
#define L0H_SIZE (37504 + 128)
// center hoist (pf_cenhoist.h): the first message GVPs of the etypes whose SOURCE is a center (ff, fp) -- to_feats_out columns 0..127
// (h_src), k-major, and bias -- pure copies like the block above; offsets in floats
#define L0C_WHT_FF 0       // [128 k][128 f]
#define L0C_B_FF 16384     // [128]
#define L0C_WHT_FP 16512   // [128 k][128 f]
#define L0C_B_FP 32896     // [128]
#define L0C_SIZE (32896 + 128)
// type tables of one timestep (k_l0_ptab): L0_NTAB tables of [rec_nf][128]: 0 pp without bias (row-group kernels: the bias
// sits in zs), 1 pp with bias, 2 pf with bias (n16 kernels), 3 the protein encoder's output itself (the residual input of
// conv layer 0's node update: n16 fused launch)
#define L0_NTAB 4
struct L0HoistParams {
    const float* src;      // the block above
    float* l0c;            // out: [16 weff][16 gate bias]
    // zs
    const int* esrc; const int* edst; const float4* xn; int Epp;
    float rbf_mu0, rbf_mu_step, rbf_inv_sigma;
    float* zs;
    // ptab
    const float* enc_w; const float* enc_b; const float* enc_lw; const float* enc_lb;   // protein encoder (w: [rec_nf+1][128])
    int rec_nf, nt;        // rows = nt * rec_nf
    const float* t_dev;    // [nt] or NULL: t_host
    float t_host[64];
    float* ptab;
    // types
    const float* prot_h0; int Np; int* ptype; int* flag;
};

struct NodeW {             // per node type
    pf_gcf ln1_w; pf_gcf ln1_b;               // message_layer_norms
    pf_gcf ln2_w; pf_gcf ln2_b;               // update_layer_norms
    const GvpW PF_AS1* upd;                   // [n_upd]
};

// pf_debug_chain: one chain of the row-group kernels on caller-supplied rows (unit tests against the reference's own
// module outputs).  kind 0: message chain (s_in [n][144] = [h_src, rbf], v_in [n][17][3] = [xhat, v_src]); 1: update chain
// (s_in [n][128], v_in [n][16][3]); 2: GVPLayerNorm; 3: noise head (s_out [n][pharm_nf], v_out [n][3])
struct UnitParams {
    const float* s_in; const float* v_in; float* s_out; float* v_out;
    int n, kind, n_gvps, pharm_nf;
    const float* stream;                      // quad stream of the chain (kinds 0, 1, 3)
    int skip_gvps;                            // kind 3: update-chain blocks (+ their flush) in front of the head in that stream
    const float* ln_w; const float* ln_b;     // kind 2
    // kind 3 as the training forward of the noise head: per GVP level and row the pre-activation scalars [128], gate
    // pre-activations [16] and gated vectors [48] k_bwd_head reads instead of recomputing the chain (NULL: not saved)
    float* sv_z; float* sv_g; float* sv_v; size_t sv_stride;
    int n16_stride;                           // kinds 16 / 17: the message / update chain in the n16 form (stream = wave 0's)
};

struct NodeParams {
    const NodeTile* tiles;
    int ntiles;
    const int* in_start;   // [4][N]
    const int* in_cnt;     // [4][N]
    int N;
    int pp_slot;           // which slot holds the prot nodes' pp in-edges: 1 all, 2 compact copy for active atoms
    const int* row_ids;    // active-atom lists (tiles with ids != 0)
    const int* dyn_cnt;
    const float* msg_s;
    const float* msg_v;
    int zero_row;          // index of an all-zero message row (lanes without in-edges)
    const float* h_in; const float* v_in;
    float* h_out; float* v_out;
    const int* gid;        // graph of each node
    const float* gnorm;    // [2 ntypes][B] (PF_NORM_GRAPH)
    int B;
    int norm_mode; float norm_value;
    NodeW w[2];
    int n_upd;
    // training forward only (one-wave kernel): GVPDropout of the aggregated message and of the update residual
    // (gvp.py:518,529); drop_thr == 0: inference
    uint32_t drop_thr; float drop_scale; uint32_t seed; int layer;
    const float* mask_override;   // tests: externally supplied multipliers [n_convs * 2][N * 144] instead of the hash
    int xcd_n;             // two-wave node launches: confine the items to the first xcd_n XCDs (0: off; k_rg_node)
    int st_n0, st_n;       // the launch covers the static tiling of the centers [st_n0, st_n0 + st_n) (fused node + head: k_rg_node_hs); st_n = 0: tile list
    int grp;               // edge slots per message partial row group: 32 (tile kernels) or 4*RG (row-group edge kernel)
    int grp_pa;            // ... of the pp / "pa" segment of the protein nodes (differs from grp under the static hoist)
    pf_gcf rg_upd[2];      // row-group kernels: quad stream of each node type's update chain (pharm of the last layer:
                           // followed by the noise head's chain and to_scalar_output)
    pf_gcf rgs_upd[2]; int rgs_stride[2];   // two-wave form of the same (wave 1's stream rgs_stride floats after wave 0's)
    // training forward (k_rg_node<., SAVE>): per update-GVP level and row the pre-activation scalars / gate pre-activations / gated
    // vectors, indexed by the row's position in the tile lists (+ N for the active-atom lists): k_bwd_node reads them instead of
    // recomputing the chain.  NULL: not saved
    float* sv_z; float* sv_g; float* sv_v; size_t sv_stride;
};

// n16 fused launch (pf_n16.hip: k_n16_fused): the LAST conv layer's edge messages with conv layer 0's node update of every
// item's SOURCE rows computed in front of the item's message chain -- the node launch of conv layer 0 disappears (n_convs = 2,
// receptive-field pruning, kNN pf edges: the sources of the last layer's edges are exactly the rows that launch updates).
// regions cleared by one k_zero_multi launch (dword-aligned pointers, byte counts that are multiples of 4)
struct ZeroList {
    void* p[8];
    unsigned long long nbytes[8];
    int cnt;
};
struct FusedParams {
    // conv layer 0's messages and where a node finds them (NodeParams of that layer)
    const int* in_start; const int* in_cnt; int N, pp_slot;
    const float* msg_s; const float* msg_v; int zero_row, grp, grp_pa;
    const int* gid; const float* gnorm; int B, norm_mode; float norm_value;
    pf_gcf ln1_w[2], ln1_b[2], ln2_w[2], ln2_b[2];   // [node type]: message / update layer norms of conv layer 0
    int n_upd;
    pf_gcf chain[4]; int chain_stride[4];            // per etype (ff, pf): [update chain of the source type][message chain], wave 0's stream
    pf_gcf upd_pharm; int upd_pharm_stride;          // the centers' update chain alone (store items)
    const float* htab; int htab_gstride;             // protein encoder output per element type (table 3 of the timestep's slot)
    const int* ptype;
    float* h_out; float* v_out;                      // conv layer 1's input state: written for the centers (the node + head launch reads it)
    const int* pharm_ptr; int Np, n_edge_items;      // store items: graph g's centers; items [0, n_edge_items) are edge items
    int xcd_split, nff_cap, npf_cap;                 // XCD-aware item assignment (k_n16_fused): capacities of the ff / pf regions in 16-slot groups
    // every graph's ff / pf region has the same capacity (k_n16_fused_u): first region's start, stride between graphs' regions
    // (ff | pf << 16), 16-slot groups per region (ff | pf << 8); uni_groups == 0: off
    int uni_ff_base, uni_pf_base, uni_strides, uni_groups;
    // edge records of the ff / pf slots (BuildParams::rec, written by the previous step's update + build), or NULL: what an edge item's
    // prologue otherwise collects in two dependent round trips -- slot -> (source, destination) -> the source's in-edge descriptors,
    // element type and both coordinates -- in one
    const int4* rec;
    const float* hcen;                               // center hoist: the centers' encoder outputs [Nf][128] (residual input of their node update), or NULL
};

// n16 tail launch (pf_n16.hip: k_n16_tail; pf_denoise_step only): ONE workgroup per graph runs the last conv layer's node
// update of the graph's centers, the noise head, the p(z_s | z_t) update and the edge build of the next dynamics call
// (pf_stepbuild.h) -- the centers of a graph are all the update + build of that graph waits for, so the step's last launch
// boundary disappears and eps never leaves the compute unit.
struct TailParams {
    // the last conv layer's messages and where a center finds them (NodeParams of that layer: slot 0 ff, slot 1 pf)
    const int* in_start; const int* in_cnt; int N;
    const float* msg_s; const float* msg_v; int zero_row, grp;
    const float* h_in; const float* v_in;            // the layer's input state of the centers
    const int* gid; const float* gnorm; int B, norm_mode; float norm_value;
    pf_gcf ln1_w, ln1_b, ln2_w, ln2_b;               // message / update layer norms of the centers
    int n_upd, n_head;
    // wave 0's quad stream: [update chain][head GVPs 0 .. n_head - 2][the head's last GVP, zero-padded to 128 + 16 outputs, with
    // to_scalar_output in gate rows 1 .. pharm_nf (pf_host.cpp: pack_n16_head_last)]; wave w's chain_stride floats further
    pf_gcf chain; int chain_stride;
    int pharm_nf;
    float* eps_h; float* eps_x;                      // [Nf][pharm_nf], [Nf][3]: also written to memory (debug / profiling readers)
};

struct HeadParams {
    const NodeTile* tiles; // pharm tiles
    int ntiles;
    int node_base;         // global id of pharm node 0 (= Np)
    const float* h; const float* v;
    const GvpW* gvps;      // [n_noise]
    int n_gvps;
    const float* a_out;    // [32 ksteps][64 lanes] packed to_scalar_output
    const float* b_out;    // [pharm_nf]
    int pharm_nf;
    float* eps_h;          // [Nf][pharm_nf]
    float* eps_x;          // [Nf][3]
    // the merged node + head / update + build launch (k_rg_node_hs_build): eps of center f also goes, word by word with agent-scope
    // stores, to xchg[f * PF_XCHG_STRIDE + (0..2: eps_x, 3..: eps_h)] -- every word is its own flag (PF_XCHG_EMPTY until written),
    // which the graph's update + build workgroup of the SAME launch polls and re-arms.  NULL everywhere else.
    unsigned int* xchg;
    int xchg_fault;        // diagnostic (pf_debug_xchg_fault): 1 = the producers never store word 0 of center 0 -- its consumer times out
    unsigned int* xchg2;   // center hoist: a second copy of eps_h's words (xchg2[f * PF_XCHG_STRIDE + u]) for the hoist workgroups, or NULL
};
#define PF_XCHG_STRIDE 20
// A NaN pattern the hardware does not generate (its own NaNs are 0x7fc00000 / 0xffc00000) but does PROPAGATE from an input: the
// producers store pf_xchg_word(bits), which replaces an inherited all-ones NaN by the canonical quiet NaN, so a payload can never
// read as "not yet written".
#define PF_XCHG_EMPTY 0xffffffffu
#ifdef __HIPCC__
__device__ __forceinline__ unsigned int pf_xchg_word(const float v) {
    const unsigned int b = __float_as_uint(v);
    return b == PF_XCHG_EMPTY ? 0x7fc00000u : b;
}
#endif

// Center hoist (round 5; k_rg_node_hs_build's third kind of workgroup, pf_cenhoist.h).  Conv layer 0's ff / fp items read their source
// center through the encoder h_c = LayerNorm(SiLU(W_enc [h_t, t] + b)) (dynamics_gvp.py:107-117, 143-151) and the h_src block of the
// first message Linear, P_et = W_et[:, :128] h_c + b_et (gvp.py:545-549) -- per EDGE, 3.5 instead of 2.3 blocks per 16-row item, on the
// conv-layer-0 launch's busiest compute units.  Both are functions of the center alone, and a denoising step knows the NEXT call's
// timestep (pf_prepare_timesteps): four centers per hoist workgroup take the head's eps_h of THIS step from a second copy of the
// exchange words, apply the step's feature update (the very expression of the update + build), encode, and leave h_c and P_ff / P_fp in
// tables -- under the update + build of the same launch, which outlasts them.  The next call's ff / fp items then start from a table
// row like its pf / pp items (n16 kind M0H), and the fused launch reads the centers' residual input from h_c.
struct CenHoistParams {
    int on, Nf, nf;                // nf = pharm_nf
    float t_next;                  // timestep of the next dynamics call
    const float* pharm_h;          // [Nf][nf] features BEFORE this step's update (the hoist reads them before the build overwrites them: see pf_cenhoist.h)
    const float* noise;            // [Nf][3 + nf] this step's draws
    float a_ts, var, sigma, ep_zt, ep_pred; int ep_feat;
    const float* enc_w; const float* enc_b; const float* enc_lw; const float* enc_lb;     // pharm encoder ([nf + 1][128] input-major)
    const float* blk;              // L0C_* block
    float* cen_h; float* cen_p;    // out: [Nf][128], [2][Nf][128]
    unsigned int* xchg2;
};

struct EncodeParams {
    int Np, Nf;
    const float* prot_h0;  // [Np][rec_nf]
    const float* pharm_h;  // [Nf][pharm_nf]
    const float* t;        // [B], or NULL: every graph is at t_scalar
    float t_scalar;
    const int* gid;
    int rec_nf, pharm_nf;
    const float* w[2]; const float* b[2]; const float* ln_w[2]; const float* ln_b[2];  // 0 prot, 1 pharm
    float* h_out;          // [N][128]
};

struct BuildParams {
    int B, Np_tot;
    const int* prot_ptr; const int* pharm_ptr;   // device [B+1]
    const float4* xn;
    const int* reg;        // [4][B] region start (absolute edge slot) of ff, pf, fp, pa (pp edges into active atoms)
    int* dyn_cnt;          // [5][B] edges of ff, pf, fp, pa; number of active atoms
    int* act_ids;          // active-atom lists (global node ids), or NULL: no receptive-field pruning
    const int* reg_act;    // [B] start of each graph's list in act_ids
    int* esrc; int* edst;
    int* eorig;            // [Ecap] static pp slot of each "pa" slot (EdgeParams::eorig), or NULL
    int* in_start; int* in_cnt; int N;   // [4][N]: slot 0 ff|fp, slot 1 pf|pp(all), slot 2 pp into active atoms, slot 3 static (pocket sharing)
    int ff_k, pf_k;
    float r2_ff, r2_pf;
    float* gnorm;          // [2][B]
    const int* pp_cnt;     // [B] static pp edges per graph
    // pocket sharing (conv layer 0 under the static hoist): copies of one pocket at the same t have identical
    // pp messages, so the "pa" region of a pocket's REPRESENTATIVE graph is re-pointed at that graph's static pp edges
    // (EdgeParams::reg) and the copies publish an empty one.  pa_static[g] = count to publish (static pp edges of a
    // representative, 0 for a copy); the per-step pa copy and the slot-2 descriptors are then not written (the node
    // kernel reads slot 3: the representative's static in-edge ranges, uploaded once per batch).  NULL: off.
    const int* pa_static;
    // ... and every graph marks its active atoms on its representative (need[rep_base[g] + atom] = need_stamp, a value
    // that is new for every build: nothing to clear), so that the shared launch computes only the static edge groups
    // that some copy of the pocket reads in this step
    const int* rep_base;   // [B] node id of the first atom of graph g's representative
    int* need; int need_stamp;
    const int* pfq_cnt;    // [B] or NULL: the pf / fp edge counts the REFERENCE books per graph when pf edges are kNN
                           // (dynamics_gvp.py:220 looks center indices up in the protein batch vector); used instead of
                           // the true counts by the per-graph normalisers
    int norm_mode;
    // speculative "pa" messages (round 5; EdgeParams::pa_skip, k_n16_pa_spec): conv layer 0's messages along the pp edges into the active
    // atoms depend on the timestep, the element types and the STATIC pocket geometry only -- not on the centers -- so the NEXT call's
    // can be computed while this step's latency-bound last launch runs, for the active atoms of THIS call; they are the next call's if its
    // "pa" region comes out the same.  The build says so per graph: pa_stamp[atom] = step_id for every active atom; an atom counts as
    // unchanged if it carried step_id - 1 and its slot-2 in-edge range is the one it had; pa_same[g] = the number of leading 16-slot groups
    // of graph g's region in front of the first slot at which anything changed (an atom joined, left or moved): 0x7fffffff when nothing
    // did.  A group's partial rows depend on its own 16 slots only, so the groups in front of the first change are the rows computed
    // ahead, bit for bit.  NULL: off.
    int* pa_stamp; int step_id; int* pa_same;
    // edge records for the NEXT call's fused launch (FusedParams::rec; pf_stepbuild.h only), three 16-byte words per ff / pf slot e:
    //   [3 e]     source coordinates (shifted), source node | element type << 24 (atoms)
    //   [3 e + 1] destination coordinates, destination node
    //   [3 e + 2] the SOURCE's conv-layer-0 in-edge descriptors: slot 0 (ff | fp) start, count; second segment (a center's pf, an atom's
    //             "pa" region) start, count -- the values in_start / in_cnt hold for it
    // ptype: element type per atom (the static hoist's), needed for the records.  NULL: no records.
    int4* rec; const int* ptype;
};

struct StepParams {
    int B, Np_tot;
    const int* prot_ptr; const int* pharm_ptr;
    float4* xn; float* pharm_h;
    const float* eps_h; const float* eps_x;
    const float* noise;        // [Nf][3+nf]
    int nf;
    float a_ts, var, sigma, ep_zt, ep_pred;
    int ep_coord, ep_feat;
    // center hoist: the updated features ALSO go here (two alternating snapshots: the hoist workgroups of the NEXT step's launch read the
    // features as they were before that step's update, while its update + build overwrites pharm_h), or NULL
    float* h_snap_out;
};
#ifdef __HIPCC__
// The p(z_s | z_t) update of sample_p_zs_given_zt (pharmacodiff.py:397-426) for ONE value -- coordinate or feature -- and the only place
// it is written: the generic update, the latency-optimised one (center threads / feature lanes) and the center hoist's copy all call
// this, with FP contraction OFF.  Left to the compiler, whether `hv / a_ts - var * e` and `mu + sigma * nz` become fused multiply-adds
// depends on the code around the expression: the forms of a step agreed bit for bit only as long as the compiler happened to decide
// alike in every one of them.  One rounding per operation is also what the reference's tensor expressions do.
__device__ __forceinline__ float pf_feat_update(const float hv, const float e, const float nz, const float a_ts, const float var,
                                                const float sigma, const float ep_zt, const float ep_pred, const int ep_feat) {
#pragma clang fp contract(off)
    float mu;
    if (ep_feat) { const float a = ep_zt * hv, b = ep_pred * e; mu = a + b; }
    else { const float a = hv / a_ts, b = var * e; mu = a - b; }
    const float n = sigma * nz;
    return mu + n;
}
#endif


struct PreParams {         // protein encoder + pp precompute (encode_pre_tile)
    int Np, rec_nf, nke;   // nke = k-steps of the encoder Linear = (rec_nf + 2) / 2
    const float* prot_h0;
    const float* t; float t_scalar; const int* gid;
    pf_gcf a_enc;          // [4 tiles][nke][64]   A fragments of the protein encoder Linear
    pf_gcf b_enc;          // [2 halves][64]       its bias in F-layout
    pf_gcf ln_w, ln_b;     // [128]
    GvpW pre_w;            // first pp message GVP of conv layer 0
    int pre_nks;           // its k-step count (81)
    float* h_out;          // [N][128]
    float* pre_out;        // [Np][128]
};
