// pf_host.cpp -- C ABI of libpfdyn.so (include/pfdyn.h): handle, weight packing, workspace,
// launch sequencing.  No compute happens on the host; without a HIP device every compute entry
// point fails (there is no CPU fallback).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <unordered_map>
#include <climits>
#include <vector>

#include "../../include/pfdyn.h"
#include "pf_device.h"
#include "pf_train.h"

#define L0_PTAB_SLOTS 2048        // timesteps whose layer-0 type tables stay resident (pf_prepare_timesteps)

extern "C" {
void pfk_edge_msg(const EdgeParams* p, int layer0, hipStream_t s);
void pfk_edge_msg_coop(const EdgeParams* p, int layer0, hipStream_t s);
void pfk_edge_msg_coop2(const EdgeParams* p, int layer0, hipStream_t s);
void pfk_node_update_coop(const NodeParams* p, int layer0, hipStream_t s);
void pfk_node_head_coop(const NodeParams* p, const HeadParams* hp, int layer0, hipStream_t s);
void pfk_noise_head_coop(const HeadParams* p, hipStream_t s);
void pfk_node_update(const NodeParams* p, int layer0, hipStream_t s);
void pfk_noise_head(const HeadParams* p, hipStream_t s);
void pfk_rg_edge(const EdgeParams* p, const EncodeParams* enc, int layer0, int rg, int split, int rgp, hipStream_t s);
void pfk_l0_hoist(const L0HoistParams* p, int what, hipStream_t s);
void pfk_rg_node(const NodeParams* p, const HeadParams* hp, const EncodeParams* enc, int layer0, int rg, int split, hipStream_t s);
void pfk_rg_unit(const UnitParams* p, hipStream_t s);
void pfk_rg_node_hs_build(const NodeParams* p, const HeadParams* hp, const StepParams* sp, const BuildParams* bp, int* xstat, int poll_sleep,
                          int avoid, int poll_max, const CenHoistParams* cp, const EdgeParams* es, const EncodeParams* ees, int spec_groups,
                          hipStream_t s);
void pfk_n16_edge(const EdgeParams* p, const EncodeParams* enc, int layer0, hipStream_t s);
void pfk_n16_unit(const UnitParams* p, hipStream_t s);
void pfk_n16_fused(const EdgeParams* p, const FusedParams* f, const EncodeParams* enc, hipStream_t s);
void pfk_n16_tail(const TailParams* t, const StepParams* sp, const BuildParams* bp, hipStream_t s);
void pfk_rg_tail(const NodeParams* p, const HeadParams* hp, const StepParams* sp, const BuildParams* bp, hipStream_t s);
void pfk_encode(const EncodeParams* p, hipStream_t s);
void pfk_encode_build(const EncodeParams* e, const BuildParams* b, hipStream_t s);
void pfk_encode_build_pre(const EncodeParams* e, const BuildParams* b, const PreParams* pp, hipStream_t s);
void pfk_build_edges(const BuildParams* p, hipStream_t s);
void pfk_load_coords(const float* src, float4* xn, int n, const int* gid, const float* shift, float sign, hipStream_t s);
void pfk_load_noise0(const float* nz, float4* xn, float* hf, int n, int nf, hipStream_t s);
void pfk_copy(const float* src, float* dst, size_t n, hipStream_t s);
void pfk_zero_multi(const ZeroList* z, hipStream_t s);
void pfk_copy2(const float* a, float* da, size_t na, const float* b, float* db, size_t nb, hipStream_t s);
void pfk_verify_copies(const float* x0, const float* h0, const int* gid, const int* prot_ptr, const int* rep_base, int Np, int rec_nf,
                       int* flag, hipStream_t s);
void pfk_scale_copy(const float* src, float* dst, size_t n, float sc, hipStream_t s);
void pfk_segment_mean(const float4* xn, const int* ptr, int base, int B, float* out, hipStream_t s);
void pfk_step_update(const StepParams* p, hipStream_t s);
void pfk_step_build(const StepParams* sp, const BuildParams* bp, int fast, hipStream_t s);
void pfk_export_coords(const float4* xn, int base, int n, const int* gid, const float* add, const float* sub,
                       float* out, hipStream_t s);
void pfk_bwd_head(const BwdHeadParams* p, int nblocks, hipStream_t s);
void pfk_bwd_node(const BwdNodeParams* p, int nblocks, hipStream_t s);
void pfk_bwd_edge_level(const BwdEdgeLevelParams* p, int nblocks, hipStream_t s);
void pfk_fix_apply(long long* A, float* G, size_t n, const float* fix, hipStream_t s);
void pfk_fix_apply_rows(const NodeTile* tiles, int ntiles, const int* dyn_cnt, const int* row_ids, long long* A_h, float* G_h,
                        long long* A_v, float* G_v, const float* fix, hipStream_t s);
void pfk_fix_scale(const float* g_h, int n_h, const float* g_x, int n_x, float* fix, hipStream_t s);
void pfk_bwd_encode(const BwdEncodeParams* p, int nblocks, hipStream_t s);
void pfk_enc_group(const float* G_h, const int* prot_ptr, const int* ptype, int B, int rec_nf, float* Gg, hipStream_t s);
void pfk_fix_enc_group(long long* A_h, float* G_h, const float* fix, const int* prot_ptr, const int* ptype, int B, int rec_nf, float* Gg,
                       int Np, int Nf, const int* onehot_flag, hipStream_t s);
void pfk_train_reduce(const ReduceParams* p, hipStream_t s);
void pfk_gather_weights(const float* flat, const int* map, size_t n, float* packed, hipStream_t s);
void pfk_n16_split_words(const float* flat, const int4* tab, size_t n, float* packed, hipStream_t s);
void pfk_pack_gvp(const float* W, const GvpT* g, int n_gvps, float* out_b, float* out_f, const ScaleArgs* sa, hipStream_t s);
void pfk_loss_prepare(const LossParams* p, hipStream_t s);
void pfk_loss_eval(const LossParams* p, hipStream_t s);
void pfk_scale_loss(float* gx, int nx, const float* a, const float* a2, float* gh, int nh, const float* b, const float* b2, hipStream_t s);
void pfk_compact_node_rows(const NodeTile* tiles, int ntiles, const int* dyn_cnt, const int* row_ids, int N, int* list, int cap, int* ucnt,
                           hipStream_t s);
void pfk_compact_rows(const EdgeTile* tiles, const int* et_tile0, int n_et, const int* dyn_cnt, int* rlist, int* ccnt, hipStream_t s);
void pfk_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
              float bc1, float bc2_sqrt, float* mirror, hipStream_t s);
void pfk_drop_masks(const TrainCommon* c, uint32_t stream, int n_elems, float* out, hipStream_t s);
void pfk_pp_radius(const float4* xn, const int* prot_ptr, int B, float r2, int maxn, int* deg, const int* row_off,
                   int* src, int* dst, int pass, hipStream_t s);
}

namespace {

static std::string g_create_error;

struct RawTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

static const char* kEtKey[4] = {"pharm_ff_pharm", "prot_pf_pharm", "pharm_fp_prot", "prot_pp_prot"};
static const char* kNtKey[2] = {"prot", "pharm"};

inline int rho(int r, int hl) { return (r & 3) + 8 * (r >> 2) + 4 * hl; }

// torch.linspace(start, end, steps) in fp32 (symmetric evaluation like ATen)
static void linspace_f32(float start, float end, int steps, float* out) {
    const float step = (end - start) / (float)(steps - 1);
    const int half = steps / 2;
    for (int i = 0; i < steps; ++i) out[i] = i < half ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Launch policy of the inference path: which kernel form a launch takes.  One place, one table.
//
// The chip: 256 compute units (CUs) = 1,024 SIMDs.  A launch is in the LATENCY regime while it has about as many items
// as the chip has places to put them (its duration is one item's chain of dependent GVPs), in the THROUGHPUT regime beyond.
// Sizes below are EDGE SLOTS (or node rows) of the launch's work list = 32 x its tiles: capacities, known on the host
// (the dynamic edge counts are not; at 256-atom pockets with kNN pf edges about a third of the slots hold an edge).
//
//   launch                        | size (slots / rows)        | form
//   ------------------------------+----------------------------+---------------------------------------------------------
//   conv layer 0's node update    | batch <= n16_fuse_rows_max | none: computed by the last layer's edge items for their own source rows
//     (n_convs = 2, kNN pf edges) |   (20,000: <= 32 graphs)   |   (pf_n16.hip: k_n16_fused; one launch less per step)
//   edge messages, any conv layer | batch <= n16_rows_max      | n16: 16-row items on the four waves of a workgroup
//                                 |   (24,000 ~ 2 busy items   |   (pf_n16.hip; conv layer 0 needs the static hoist's type
//                                 |   per CU)                  |   tables, else the row-group form)
//   edge messages                 | < rg2_rows_min (12,000 ~   | row-group, 4 rows per wave (pf_rg.hip); four-wave workgroups
//                                 |   3 four-row items / SIMD) |   up to RG_QUAD_MAX item slots (fixed SIMD placement)
//                                 | >= rg2_rows_min            | row-group, 8 rows per wave
//     conv layer 0 under the      | >= rg2p_rows_min (11,000)  |   hoisted (two-block) items 8 rows, full-chain items 4
//     static hoist (compact list) | >= rg2_rows_min_hoist      |   all items 8 rows
//                                 |   (30,000)                 |
//   node update                   | < rg2_rows_min_node        | row-group, 4 rows per wave; on TWO waves per item while the
//                                 |                            |   launch has <= rg_split_max_node (256) items: fewer items than CUs
//                                 | >= rg2_rows_min_node       | row-group, 8 rows per wave
//   last layer's nodes + head     | <= rg_split_max_head (512) | row-group, 4 rows on two waves (a 6-7 block chain: the longest of a step)
//   everything, row-group off     | rows > rg_rows_max         | 32-row tile kernels (pf_kernels.hip): four waves per tile up to
//   (training forward of dense    |                            |   coop_edge_max / coop_node_max tiles, two workgroups per CU up to
//   layers; PFDYN_RG_ROWS_MAX=0)  |                            |   coop2_*_max, one wave per tile beyond
//
// Every threshold can be overridden from the environment (tests force every form onto the goldens; sweeps: tools/):
//   PFDYN_N16 (bit 0: conv layers >= 1, bit 1: conv layer 0, bit 2: conv layer 0's node update fused into the last layer's edge
//   launch when n_convs = 2, bit 3: the tail launch of a denoising step -- last node update + noise head + sampler update + edge
//   build, one workgroup per graph: off unless asked for; default 7), PFDYN_TAIL_GRAPHS_MAX, PFDYN_TAIL_FORM (rg | n16), PFDYN_N16_ROWS_MAX (sets both n16 thresholds), PFDYN_N16_FUSE_ROWS_MAX, PFDYN_RG_ROWS_MAX,
//   PFDYN_RG2_ROWS_MIN (sets all four 8-row thresholds) / _NODE / _HOIST, PFDYN_RG2P_ROWS_MIN, PFDYN_L0_RGA / PFDYN_L0_RGP (rows-per-
//   wave factor of the full-chain / hoisted items of a compact layer-0 launch), PFDYN_RG_SPLIT_MAX (all three) / _NODE / _HEAD,
//   PFDYN_COOP_EDGE_MAX, PFDYN_COOP2_EDGE_MAX, PFDYN_COOP_NODE_MAX.  Forcing a row-group form switches the n16 form off
//   unless PFDYN_N16 is given.  Feature switches (not thresholds) are read in pf_handle::init_tuning.
// ------------------------------------------------------------------------------------------------------------------
struct LaunchPolicy {
    static constexpr int kCUs = 256, kSIMDs = 4 * kCUs;
    // bit 3: the tail launch (node update of the last layer + noise head + sampler update + edge build in one launch, one workgroup per
    // graph).  OFF by default: measured on config 2 it LOSES to the separate launches in both forms -- 30.7 us (row-group form: two
    // two-wave items per compute unit stream 192 KB of weights per block through one CU's memory path) and 28.0 us (n16 form: 2.0 us
    // per block on a 16-row tile that holds six centers) against 14.7 + 10.3 us (profiles/r04/tail_forms.txt)
    int n16_mask = 7;
    int tail_graphs_max = 256;              // ... up to this many graphs (one four-wave workgroup per graph; it does not share a CU)
    // k_n16_fused: ff / store items on XCDs 0..3, pf items on XCDs 4..7 -- an XCD's L2 fetches half of the launch's weights: 19.3 -> 18.5 us
    // at config 2 (PFDYN_XCD_SPLIT=0: off).  The same idea on the conv-layer-0 launch (pa / pf items on five XCDs, ff / fp items on three)
    // LOST 1.6 us: that launch is throughput-bound with two items per compute unit, and the split unbalances it (profiles/r04)
    int xcd_split = 1;
    int edge_rec = 1;                          // PFDYN_EDGE_REC: edge records for the fused launch (BuildParams::rec)
    int node_static = 1;                    // ... with its tiles computed, not loaded (k_rg_node_hs; PFDYN_NODE_STATIC=0: the tile-list kernel)
    int node_xcds = 2;                      // the fused node + head launch of a small batch runs on this many XCDs (PFDYN_NODE_XCDS; 0: all eight)
    int fused_uni = 1;                      // the fused launch's arithmetic tiling when every graph's regions have one capacity (PFDYN_FUSED_UNI)
    int xchg_sleep = 1, hsb_avoid = 0;      // its poll interval in units of ~0.2 us (PFDYN_XCHG_SLEEP); update + build workgroups kept off the first n XCDs (PFDYN_HSB_AVOID)
    int hs_build = 1;                       // the merged last launch of a step (k_rg_node_hs_build: node + head items and the update + build of every
                                            // graph as workgroups of one grid; PFDYN_HS_BUILD=0: two launches)
    int tail_form = 4;                      // 4: the row-group form (k_rg_tail: two two-wave items of four centers), 16: the n16 form (k_n16_tail)
    long n16_fuse_rows_max = 20000;         // the fused launch (bit 2): +2-3 % up to 32 graphs of 256 atoms, -4 % at 40 (its items carry five blocks: throughput-bound earlier)
    long n16_rows_max = 24000;              // measured at 256-atom pockets (575 slots per graph): +5 % at 16 graphs, +10 % at 32, -3..-5 % at 64, -15 % at 256
    int rg_rows_max = 1 << 30;
    int rg2_rows_min = 12000;               // ~3 four-row items per SIMD (3 x 4 x kSIMDs = 12,288)
    int rg2_rows_min_node = 12000;
    int rg2p_rows_min = 11000;              // mixed 4 / 8 rows: +5 % at 24 graphs, +2 % at 32, +7 % at 40, +11 % at 48 over all-4-rows (round 2 sweep)
    int rg2_rows_min_hoist = 30000;         // all-8-rows: +10 % at 56-64 graphs
    int l0_rga = 0, l0_rgp = 0;             // 0: by the thresholds above
    int rg_split_max = 128;                 // edge launches: two waves per 4-row item up to this many items (kCUs / 2)
    int rg_split_max_node = 256;            // = kCUs
    int rg_split_max_head = 512;            // = 2 kCUs: +2-3 % at 144-384 items
    int coop_edge_max = 256, coop_node_max = 1024;       // tile kernels: one tile per CU / per SIMD
    int coop2_edge_max = 12000, coop2_dense_max = 1024;
    // 0: tile kernels; 1 / 2: row-group kernels with 4 / 8 rows per wave
    int rg_mode(int ntiles) const {
        const long rows = (long)ntiles * 32;
        if (rows > rg_rows_max) return 0;
        return rows >= rg2_rows_min ? 2 : 1;
    }
    void from_env() {
        auto geti = [](const char* v, int& x) { if (const char* e = getenv(v)) x = atoi(e); };
        geti("PFDYN_COOP_EDGE_MAX", coop_edge_max); geti("PFDYN_COOP_NODE_MAX", coop_node_max);
        if (const char* e = getenv("PFDYN_COOP2_EDGE_MAX")) coop2_edge_max = coop2_dense_max = atoi(e);
        if (const char* e = getenv("PFDYN_RG_SPLIT_MAX")) rg_split_max = rg_split_max_node = rg_split_max_head = atoi(e);
        geti("PFDYN_RG_SPLIT_MAX_NODE", rg_split_max_node); geti("PFDYN_RG_SPLIT_MAX_HEAD", rg_split_max_head);
        geti("PFDYN_RG_ROWS_MAX", rg_rows_max);
        if (const char* e = getenv("PFDYN_RG2_ROWS_MIN")) rg2_rows_min = rg2_rows_min_hoist = rg2p_rows_min = rg2_rows_min_node = atoi(e);
        geti("PFDYN_RG2P_ROWS_MIN", rg2p_rows_min); geti("PFDYN_RG2_ROWS_MIN_NODE", rg2_rows_min_node); geti("PFDYN_RG2_ROWS_MIN_HOIST", rg2_rows_min_hoist);
        geti("PFDYN_L0_RGP", l0_rgp); geti("PFDYN_L0_RGA", l0_rga);
        for (const char* v : {"PFDYN_RG2_ROWS_MIN", "PFDYN_RG2P_ROWS_MIN", "PFDYN_RG2_ROWS_MIN_HOIST", "PFDYN_RG_SPLIT_MAX", "PFDYN_L0_RGA",
                              "PFDYN_L0_RGP", "PFDYN_RG_ROWS_MAX"})
            if (getenv(v)) n16_mask = 0;
        geti("PFDYN_N16", n16_mask);
        geti("PFDYN_TAIL_GRAPHS_MAX", tail_graphs_max);
        geti("PFDYN_XCD_SPLIT", xcd_split);
        geti("PFDYN_EDGE_REC", edge_rec);
        geti("PFDYN_NODE_XCDS", node_xcds); geti("PFDYN_NODE_STATIC", node_static); geti("PFDYN_HS_BUILD", hs_build); geti("PFDYN_FUSED_UNI", fused_uni); geti("PFDYN_XCHG_SLEEP", xchg_sleep); geti("PFDYN_HSB_AVOID", hsb_avoid);
        if (const char* e = getenv("PFDYN_TAIL_FORM")) tail_form = (e[0] == 'n' || atoi(e) == 16) ? 16 : 4;
        if (const char* e = getenv("PFDYN_N16_ROWS_MAX")) n16_rows_max = n16_fuse_rows_max = atol(e);
        if (const char* e = getenv("PFDYN_N16_FUSE_ROWS_MAX")) n16_fuse_rows_max = atol(e);
    }
};

struct pf_handle {
    LaunchPolicy pol;
    pf_config cfg{};
    std::string err;
    std::map<std::string, RawTensor> raw;
    bool committed = false;

    // ---- packed weights (one device allocation)
    float* d_w = nullptr;
    std::vector<float> h_w;                 // staging
    GvpW* d_gvp = nullptr;                  // table of all GvpW
    std::vector<GvpW> h_gvp;
    // indices into the GvpW table
    int msg_base(int layer, int et) const { return ((layer * 4 + et) * cfg.n_message_gvps); }
    int upd_base(int layer, int nt) const { return n_msg_tot + (layer * 2 + nt) * cfg.n_update_gvps; }
    int head_base() const { return n_msg_tot + n_upd_tot; }
    int n_msg_tot = 0, n_upd_tot = 0;
    // raw (unpacked) device weights: offsets into d_w
    size_t enc_w[2]{}, enc_b[2]{}, enc_lw[2]{}, enc_lb[2]{};
    std::vector<size_t> ln_off;             // [layer][nt][4]: ln1_w ln1_b ln2_w ln2_b
    size_t out_a = 0, out_b = 0;
    size_t enc_a = 0, enc_bf = 0;           // protein encoder as A fragments / F-layout bias (encode_pre_tile)
    bool use_pre = true;

    // ---- batch / workspace
    bool have_batch = false;
    int B = 0, Np = 0, Nf = 0, N = 0;
    int64_t Epp = 0, Ecap = 0;
    std::vector<int> h_prot_ptr, h_pharm_ptr;
    std::vector<int> h_reg;                 // [3][B]
    std::vector<int> h_cap;                 // [3][B]
    int n_edge_tiles = 0, n_node_tiles = 0, n_head_tiles = 0;
    int zero_row = 0;
    int n_edge_tiles_last = 0, n_node_tiles_last = 0;
    // receptive-field pruning of the second-to-last conv layer: only the protein atoms that are the source of a
    // pf edge (the only protein rows the last layer reads) are updated, and only the edges into them are computed
    bool prune = true;
    int n_edge_tiles_act = 0, n_node_tiles_act = 0;
    EdgeTile* d_edge_tiles_act = nullptr;
    NodeTile* d_node_tiles_act = nullptr;
    int *d_act_ids = nullptr, *d_reg_act = nullptr;   // last conv layer: only what feeds the pharm nodes
    void* d_ws = nullptr;                   // one allocation, carved below
    size_t ws_capacity = 0;                 // bytes behind d_ws: kept across pocket batches while it is large enough
    // pf_set_pocket_batch stages every host-built table in pinned memory, in the layout of the workspace's table section,
    // and uploads it with ONE asynchronous copy on the caller's stream (two staging buffers alternate; a buffer is reused
    // only after the copy that read it has completed)
    void* stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int stage_next = 0;
    // The table section lives in two device buffers of its own that alternate between binds, and its upload runs on a
    // copy stream: a training loop binds a new batch every step, and on the caller's stream the ~8 MB copy would sit
    // between two steps (0.14 ms of a 2 ms step) instead of under the previous step's kernels.  tab_guard[w]: everything
    // that read buffer w has finished (recorded on the caller's stream at the start of the bind after the one that
    // used w); tab_up[w]: the upload into w is complete (the caller's stream waits for it).
    void* d_tab[2] = {nullptr, nullptr};
    size_t tab_cap[2] = {0, 0};
    hipEvent_t tab_guard[2] = {nullptr, nullptr}, tab_up[2] = {nullptr, nullptr};
    bool tab_guard_set[2] = {false, false};
    int tab_next = 0;
    hipStream_t s_copy = nullptr;
    hipStream_t s_side = nullptr;           // side stream of the backward pass (work lists, early gradient sums, encoders): NOT the copy
                                            // stream -- the next bind's upload must not queue behind the end of this backward
    // one-hot check of the protein features (static hoist): 0 unknown (device flag pending), 1 one-hot, 2 not
    int l0_state = 0;
    int* l0flag_host = nullptr;             // pinned; written by an async copy of d_l0flag
    hipEvent_t l0flag_ev = nullptr;
    size_t tws_capacity = 0;                // bytes behind d_tws (kept across batches like d_ws)
    int *d_prot_ptr = nullptr, *d_pharm_ptr = nullptr, *d_gid = nullptr, *d_reg = nullptr, *d_dyn_cnt = nullptr,
        *d_esrc = nullptr, *d_edst = nullptr, *d_in_start = nullptr, *d_in_cnt = nullptr, *d_pp_cnt = nullptr;
    EdgeTile* d_edge_tiles = nullptr;
    NodeTile* d_node_tiles = nullptr;
    NodeTile* d_head_tiles = nullptr;
    float4* d_xn = nullptr;
    float *d_prot_x0 = nullptr, *d_prot_h0 = nullptr, *d_pharm_h = nullptr, *d_t = nullptr, *d_h[2] = {nullptr, nullptr},
          *d_v[2] = {nullptr, nullptr}, *d_msg_s = nullptr, *d_msg_v = nullptr, *d_eps_h = nullptr, *d_eps_x = nullptr,
          *d_com_init = nullptr, *d_com_tmp = nullptr, *d_gnorm = nullptr, *d_pre = nullptr;
    int* d_pfq_cnt = nullptr;      // [B] reference-booked pf edge counts (message_norm 0 with kNN pf edges), else NULL
    // pocket sharing: pf_set_pocket_groups names, per graph, the representative graph of its pocket; the
    // next bind verifies the claim and, when it pays, prepares the tables of the sharing mode
    std::vector<int> pending_rep;           // consumed by the next pf_set_pocket_batch*
    bool share_ok = false;                  // tables of the sharing mode exist for this batch
    int share_check = 1;                    // the claim's rows: 1 verified (host rows, or nothing claimed), 0 compared on the device and not
                                            // read back yet (l0flag_host[1], behind l0flag_ev), 2 found false
    int* d_reg_share = nullptr;             // [4][B]: d_reg with the kind-3 entries of representatives at their static pp edges
    int* d_pa_static = nullptr;             // [B]: static pp edge count of a representative, 0 for a copy
    int* d_rep_base = nullptr;              // [B]: node id of the first atom of each graph's representative
    int* d_need = nullptr;                  // [Np]: stamp of the last build in which some copy had the atom active
    int need_stamp = 0, edges_stamp = 0;    // stamp of the next build / of the build the current edges came from
    long share_rows = 0;                    // edge slots of a shared layer-0 launch (capacities of ff, pf, fp + the static ranges)
    std::vector<int> h_share_start, h_share_cnt;   // host copies of d_reg_share's kind-3 entries / d_pa_static (grid sizing)
    bool share_disable = false;             // PFDYN_NO_POCKET_SHARE=1
    bool train_rg_node = true;              // PFDYN_TRAIN_TILE_NODE=1: the training forward keeps the 32-row tile node kernel
    bool train_rg_edge = true;              // PFDYN_TRAIN_TILE_EDGE=1: ... and the 32-slot tile edge kernel
    bool train_rg_head = true;              // PFDYN_TRAIN_TILE_HEAD=1: the training forward's noise head on the tile kernel (the backward recomputes it)
    float *t_hsv_z = nullptr, *t_hsv_g = nullptr, *t_hsv_v = nullptr;   // head levels saved by the training forward [n_noise_gvps][Nf][128 / 16 / 48]
    bool t_head_saved = false;              // ... by the last pf_train_forward
    unsigned int* d_xchg = nullptr;         // exchange words of the merged launch [Nf centers][PF_XCHG_STRIDE]: part of the workspace (carved per
                                            // bind), set to PF_XCHG_EMPTY by every pf_sample_begin on the caller's stream, re-armed word by word by the consumers
    int* d_xstat = nullptr;                 // [1] time-outs of the exchange since the handle was created (cumulative, never cleared: no race with a run in flight)
    int* xstat_host = nullptr;              // pinned; an async copy of d_xstat follows every sampling run (pf_sample_end)
    int xstat_ack = 0;                      // the count already reported (pf_sample_status / pf_sample_begin / pf_debug_xchg_timeouts)
    int xchg_fault = 0, xchg_poll_max = 0;  // diagnostics (pf_debug_xchg_fault)
    // center hoist (CenHoistParams; pf_cenhoist.h): M0H streams of conv layer 0's chains for EVERY etype, the L0C block, the per-batch
    // tables and exchange copy, two alternating snapshots of the centers' features, the announced timestep plan (pf_prepare_timesteps)
    // that tells a denoising step the NEXT call's t, and what the tables currently hold
    size_t n16_l0h[4] = {0, 0, 0, 0}, n16_l0h_stride[4] = {0, 0, 0, 0};
    size_t l0c_off = 0;
    bool cen_hoist = true;                  // PFDYN_NO_CENTER_HOIST=1: off
    float *d_cen_h = nullptr, *d_cen_p = nullptr, *d_snap[2] = {nullptr, nullptr};
    unsigned int* d_xchg2 = nullptr;
    int snap_cur = -1;                      // which snapshot holds the features as they are now (-1: none)
    std::vector<float> t_plan; size_t plan_pos = 0;
    bool cen_valid = false; float cen_t = 0.f; uint64_t cen_wver = 0;
    bool last_cen = false;                  // the last dynamics call started its ff / fp items from the tables (pf_debug_kernel_family(n_convs + 1))
    // speculative "pa" messages (BuildParams::pa_same, k_n16_pa_spec): a side stream, the events that order it against the caller's
    // stream, a per-handle step counter (never reset: stamps of an earlier trajectory must not look recent), the conv-layer-0 launch's
    // parameters of the current call, and what the rows computed ahead are for
    bool pa_spec = true;                    // PFDYN_NO_PA_SPEC=1: off
    bool spec_valid = false; float spec_t = 0.f; uint64_t spec_wver = 0;
    int step_id = 2;
    int *d_pa_stamp = nullptr, *d_pa_same = nullptr;
    bool e0_saved = false; EdgeParams e0{}; EncodeParams ep0{}; int e0_groups = 0;
    int last_spec = 0;                      // the last dynamics call skipped "pa" regions computed ahead (pf_debug_kernel_family(n_convs + 2): 1)
                                            // when the next one begins: a time-out there is reported, late but never silently
    bool no_fixed_shapes = false;           // PFDYN_NO_FIXED_SHAPES: k_bwd_edge_level reads every level's GVP shape from the table (the A/B of its FX forms)
    ScaleArgs pend_scale{}; bool has_pend_scale = false;    // loss_backward -> pf_train_backward: the unit gradients' scaling, not yet launched
    bool no_fix_fuse = false;               // PFDYN_NO_FIX_FUSE: k_fix_apply and k_enc_group as two launches (the A/B of k_fix_enc_group)
    bool train_bf16 = false;                // pf_train_set_precision: the bf16 leg (dense Linears of the message chains' forward and of every
                                            // gradient kernel on bf16 matrix instructions; PFDYN_TRAIN_BF16=1 sets it at creation)
    bool train_node_save = true;            // PFDYN_TRAIN_NODE_RECOMPUTE=1: k_bwd_node recomputes the update chains instead of reading saved levels
    std::vector<float*> t_nsv_z, t_nsv_g, t_nsv_v;  // per conv layer: update-chain levels saved by the training forward [n_update_gvps][2 N][128 / 16 / 48]
    std::vector<char> t_node_saved;         // ... by the last pf_train_forward
    std::vector<int> t_grp;                 // per conv layer: slots per message partial-row group of the last training forward
    bool sampling = false;
    int max_np = 0, max_nf = 0;             // largest pocket of the batch, most centers in a graph
    bool edges_built = false;               // the dynamic edges of the current coordinates exist (built by k_step_build)
    bool edges_share = false;               // ... in the pocket-sharing form (no pa copies)
    // row-group kernels (pf_rg.hip): quad streams of the message chains [layer][etype] and update chains [layer][ntype]
    // (offsets into d_w); which form a launch takes: LaunchPolicy above
    std::vector<size_t> rg_msg, rg_upd;
    std::vector<size_t> rgs_msg, rgs_upd, rgs_upd_stride;   // two-wave form: wave 0's stream; wave 1's follows *_stride floats later
    size_t rgs_msg_stride = 0;
    // n16 kernels (pf_n16.hip): per chain the four waves' quad streams, wave w's *_stride floats after wave w - 1's.
    // n16_msg: every message chain with a full first GVP (M0F: what conv layers >= 1 run, and what pf_debug_chain tests)
    std::vector<size_t> n16_msg, n16_upd;
    size_t n16_msg_stride = 0, n16_upd_stride = 0;
    // packed elements [n16_begin, n_packed) are the n16 streams: no training kernel reads them, so pf_set_flat_params (called
    // after every optimiser step) refreshes only what precedes them and marks them stale; the first inference call afterwards
    // (run_dynamics without train, pf_debug_chain) gathers them (n16_refresh)
    size_t n16_begin = 0;
    bool n16_stale = false;
    // -DN16_SPLIT builds: the main quads of the n16 streams hold bf16 planes, two weights per 32-bit word -- not a gather.  While the
    // index-valued packing pass runs (split_record) pack_n16_raw notes (word position, flat index a, flat index b, plane) per word;
    // n16_refresh re-derives those words from the flat vector behind the gather (k_n16_split_words)
    bool split_record = false;
    std::vector<int4> split_pending, split_tab;     // positions relative to the stream being packed / to h_w
    int4* d_split_tab = nullptr; size_t n_split_tab = 0;
    // conv layer 0's message chains in their own forms: protein sources (pf, pp) start from a type-table row (M0H),
    // centers (ff, fp) have zero node vectors (M0Z)
    size_t n16_l0[4] = {0, 0, 0, 0}, n16_l0_stride[4] = {0, 0, 0, 0};
    // fused launch (n_convs = 2): per etype of the last layer (ff, pf) [update chain of conv layer 0 for the source type][message chain]
    size_t n16_fused[2] = {0, 0}, n16_fused_stride[2] = {0, 0};
    // tail launch (pf_n16.hip: k_n16_tail): [update chain of the centers in the last conv layer][noise head; its last GVP padded, with to_scalar_output]
    size_t n16_tail = 0, n16_tail_stride = 0;
    bool tail_done = false;                 // the last run_dynamics call of a denoising step also did the step's update + build
    int last_tail = 0;                      // pf_debug_kernel_family(layer = n_convs): 0 / 4 (k_rg_tail) / 16 (k_n16_tail)
    float *d_msg_s2 = nullptr, *d_msg_v2 = nullptr;   // the last conv layer's message rows when conv layer 0's are still being read (fused launch)
    std::vector<int> last_family;           // per conv layer: pf_debug_kernel_family
    int last_hoist = 0;                     // pf_debug_l0_hoist
    // ---- static hoist of conv layer 0's pp messages (pf_rg.hip, EdgeParams::zs).  Everything derived from the weights
    // carries the version of the weights it was computed from (commit / pf_set_flat_params bump w_version).
    bool l0_hoist = true;                   // PFDYN_NO_L0_HOIST=1: off
    size_t l0h_off = 0;                     // L0H_* block in d_w
    uint64_t w_version = 1, zs_version = 0, ptab_version = 0;
    bool l0_onehot = false;                 // every protein feature row of the batch is an element one-hot
    bool coords_custom = false;             // protein coordinates came from the caller of this call (not the batch's own)
    bool zs_batch_coords = false;           // d_zs was computed from (a rigid translate of) the batch's coordinates
    float* d_l0c = nullptr;                 // [32]
    float* d_ptab = nullptr;                // [L0_PTAB_SLOTS][L0_NTAB][rec_nf][128] tables of the timesteps seen (scalar-t calls)
    std::unordered_map<uint32_t, int> ptab_slot;
    int *d_eorig = nullptr, *d_ptype = nullptr, *d_l0flag = nullptr;
    // edge records of the ff / pf slots (BuildParams::rec / FusedParams::rec): written by the merged launch's update + build for the next
    // call's fused launch; rec_valid: the edges in place are the ones that build emitted, with their records
    int4* d_rec = nullptr; bool rec_valid = false;
    float *d_zs = nullptr, *d_ptab_pg = nullptr;
    bool enc_on_the_fly = true;             // PFDYN_NO_ENC_FLY=1: always launch the encoders
    bool step_build_fast = true;            // PFDYN_NO_FAST_BUILD=1: the generic update + build bodies
    bool rg_compact = true;                 // PFDYN_NO_COMPACT=1: row-group edge launches walk the tile lists
    bool fuse_head = true;                  // last conv layer's node update + noise head in one launch (PFDYN_NO_FUSE_HEAD=1: separate)
    void init_tuning() {
        if (const char* e = getenv("PFDYN_NO_PRE")) use_pre = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_FUSE_HEAD")) fuse_head = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_PRUNE")) prune = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_ENC_FLY")) enc_on_the_fly = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_COMPACT")) rg_compact = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_FAST_BUILD")) step_build_fast = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_L0_HOIST")) l0_hoist = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_CENTER_HOIST")) cen_hoist = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_PA_SPEC")) pa_spec = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_NO_POCKET_SHARE")) share_disable = atoi(e) != 0;
        if (const char* e = getenv("PFDYN_TRAIN_TILE_NODE")) train_rg_node = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_TRAIN_TILE_EDGE")) train_rg_edge = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_TRAIN_TILE_HEAD")) train_rg_head = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_TRAIN_NODE_RECOMPUTE")) train_node_save = atoi(e) == 0;
        if (const char* e = getenv("PFDYN_TRAIN_BF16")) train_bf16 = atoi(e) != 0;
        if (const char* e = getenv("PFDYN_NO_FIX_FUSE")) no_fix_fuse = atoi(e) != 0;
        if (const char* e = getenv("PFDYN_NO_FIXED_SHAPES")) no_fixed_shapes = atoi(e) != 0;
        pol.from_env();
    }

    // ---- gradient path (pf_train_*): flat parameter vector in state-dict order, GvpT tables, per-layer activations
    float* d_flat = nullptr;
    size_t nparams = 0;
    std::vector<std::pair<std::string, std::pair<size_t, size_t>>> flat_layout;   // name -> (offset, numel), state-dict order
    GvpT* d_gvpt = nullptr;                 // same indexing as the GvpW table
    std::vector<int> h_map;                 // packed element -> flat parameter index (-1: zero), see pf_commit_weights
    int* d_map = nullptr;
    size_t n_packed = 0;
    void* d_tws = nullptr;                  // training workspace of the current batch (allocated on first use)
    std::vector<float*> t_H, t_V, t_msg_s, t_msg_v;
    std::vector<float*> t_sv_z, t_sv_g, t_sv_v;     // per layer: [n_message_gvps][Ecap] rows saved by the forward
    float *t_gs_buf = nullptr, *t_gv_buf = nullptr;
    int et_tile0[5] = {0, 0, 0, 0, 0};      // tile ranges of ff, pf, fp, pp in d_edge_tiles
    int et_tile0_act[5] = {0, 0, 0, 0, 0};  // ... of ff, pf, fp, pa in d_edge_tiles_act
    float *t_G_h[2] = {nullptr, nullptr}, *t_G_v[2] = {nullptr, nullptr}, *t_gagg_s = nullptr, *t_gagg_v = nullptr,
          *t_gpart = nullptr, *t_geps_h = nullptr, *t_geps_x = nullptr;
    long long *t_A_h = nullptr, *t_A_v = nullptr;      // fixed-point accumulators of the level-0 scatter (kept clear between uses)
    float* d_lpart = nullptr;               // k_loss_eval's arrival counter (first 64 bytes) + partial sums: in the workspace's zero section
    void* d_tA = nullptr;                   // their own allocation: it outlives the batches, so "kept clear" holds across them
    size_t tA_capacity = 0;                 // bytes
    bool tA_dirty = false;                  // a backward pass stopped between the scatter and pfk_fix_apply
    int enc_begin = 0, enc_n = 0;           // flat range of the encoders' parameters (contiguous: the first tensors of the state dict)
    float* t_gpart_enc = nullptr;
    TensorSeg* d_tseg = nullptr; int n_tseg = 0;   // class of every parameter tensor (pf_train.h: which gradient copies hold it)
    int n_gvpt = 0;                         // entries of d_gvpt (message, update, head GVPs)
    float* d_wpack = nullptr;               // k_pack_gvp tables of every GVP (input-gradient fragments, then forward fragments); valid for w_version == wpack_version
    uint64_t wpack_version = ~0ull;
    float *t_lx0c = nullptr, *t_lag = nullptr, *t_lsg = nullptr, *t_lgx = nullptr, *t_lgh = nullptr, *t_lout = nullptr;   // pf_train_loss_forward
    bool t_have_loss = false;
    float* t_Gg = nullptr;                  // encoder backward: upstream gradient summed per (graph, element)
    int* t_ulist = nullptr;                 // dense per-type row list of the layer being differentiated (k_compact_node_rows; counts: t_ccnt[97], [98])
    int t_ucap = 0;
    size_t t_clist_cap = 0, t_ulist_cap = 0;   // ints per conv layer in t_clist / t_ulist (one list per layer: they are built ahead, on the side stream)
    hipEvent_t cmp_ev[3] = {nullptr, nullptr, nullptr};
    int *t_clist = nullptr, *t_ccnt = nullptr;   // dense list of the valid edge slots of the layer being differentiated (k_compact_rows), counts
    float* t_fix = nullptr;                 // [2] scale / inverse scale of the current backward call
    int t_nblk = 0;
    const float* t_mask_override = nullptr; // pf_debug_set_dropout_masks
    bool t_have_fwd = false;
    bool t_ws_ready = false;                // d_tws is carved (and its zero rows set) for the current batch
    TrainCommon t_common{};
    // (looked up ~26 times per backward pass: an index over the 244 names, rebuilt when the layout is -- a linear search with string
    // compares was 50-100 us of a training step's host time, and that step is host-bound with a new batch every step)
    mutable std::unordered_map<std::string, size_t> flat_index;
    mutable bool flat_index_ok = false;
    size_t flat_offset(const std::string& name) const {
        if (!flat_index_ok) {
            flat_index.clear();
            for (const auto& kv : flat_layout) flat_index.emplace(kv.first, kv.second.first);
            flat_index_ok = true;
        }
        const auto it = flat_index.find(name);
        return it == flat_index.end() ? (size_t)-1 : it->second;
    }
    std::vector<int> edge_fx;               // [conv layer][message GVP level]: k_bwd_edge_level's shape class (msg_spec is string-building host work)

    // ---- optional per-kernel timing with HIP events on the caller's stream (pf_profile_*)
    // classes 0..8: pf_profile_read (inference path); 9..12: pf_profile_read_train (gradient kernels)
    enum { K_ENCODE = 0, K_BUILD, K_EDGE, K_NODE, K_HEAD, K_STEP, K_EDGE_COOP, K_NODE_COOP, K_EDGE_LAST,
           K_BWD_HEAD, K_BWD_NODE, K_BWD_EDGE_LEVEL, K_BWD_REST, K_NUM };
    unsigned prof_mask = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[K_NUM];
    size_t prof_used[K_NUM] = {};
};

namespace {

#define PF_FAIL(h, code, ...)                                   \
    do {                                                        \
        char _b[512];                                           \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                  \
        (h)->err = _b;                                          \
        return (code);                                          \
    } while (0)

#define PF_HIP(h, call)                                                                              \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess) PF_FAIL(h, PF_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// expected state-dict layout (mirrors the reference's module tree)
// ------------------------------------------------------------------------------------------------
struct GvpSpec { std::string prefix; int vi, vo, si, so; };

static void gvp_names(const GvpSpec& g, std::vector<std::pair<std::string, std::vector<int64_t>>>& out) {
    const int h = std::max(g.vi, g.vo);
    out.push_back({g.prefix + "Wh", {g.vi, h}});
    out.push_back({g.prefix + "Wu", {h, g.vo}});
    out.push_back({g.prefix + "to_feats_out.0.weight", {g.so, h + g.si}});
    out.push_back({g.prefix + "to_feats_out.0.bias", {g.so}});
    out.push_back({g.prefix + "scalar_to_vector_gates.weight", {g.vo, g.so}});
    out.push_back({g.prefix + "scalar_to_vector_gates.bias", {g.vo}});
}

static std::string conv_prefix(int layer) {
    return "dynamics.noise_predictor.conv_layers." + std::to_string(layer) + ".";
}
static GvpSpec msg_spec(const pf_config& c, int layer, int et, int j) {
    GvpSpec g;
    g.prefix = conv_prefix(layer) + "edge_message_fns." + kEtKey[et] + "." + std::to_string(j) + ".";
    g.vi = c.vector_size + (j == 0 ? 1 : 0);
    g.vo = c.vector_size;
    g.si = c.n_hidden_scalars + (j == 0 ? c.rbf_dim : 0);
    g.so = c.n_hidden_scalars;
    return g;
}
static GvpSpec upd_spec(const pf_config& c, int layer, int nt, int j) {
    GvpSpec g;
    g.prefix = conv_prefix(layer) + "node_update_fns." + kNtKey[nt] + "." + std::to_string(j) + ".";
    g.vi = g.vo = c.vector_size;
    g.si = g.so = c.n_hidden_scalars;
    return g;
}
static GvpSpec head_spec(const pf_config& c, int k) {
    GvpSpec g;
    g.prefix = "dynamics.noise_predictor.noise_predictor.gvps." + std::to_string(k) + ".";
    g.vi = c.vector_size;
    g.si = c.n_hidden_scalars;
    const bool last = k == c.n_noise_gvps - 1;
    g.vo = last ? 1 : c.vector_size;
    g.so = last ? 64 : c.n_hidden_scalars;
    return g;
}

static std::vector<std::pair<std::string, std::vector<int64_t>>> expected_tensors(const pf_config& c) {
    std::vector<std::pair<std::string, std::vector<int64_t>>> v;
    const int S = c.n_hidden_scalars;
    for (int nt = 0; nt < 2; ++nt) {
        const std::string p = std::string("dynamics.") + kNtKey[nt] + "_encoder.";
        const int nf = nt ? c.pharm_nf : c.rec_nf;
        v.push_back({p + "0.weight", {S, nf + 1}});
        v.push_back({p + "0.bias", {S}});
        v.push_back({p + "2.weight", {S}});
        v.push_back({p + "2.bias", {S}});
    }
    for (int l = 0; l < c.n_convs; ++l) {
        for (int et = 0; et < 4; ++et)
            for (int j = 0; j < c.n_message_gvps; ++j) gvp_names(msg_spec(c, l, et, j), v);
        for (int nt = 0; nt < 2; ++nt) {
            for (int j = 0; j < c.n_update_gvps; ++j) gvp_names(upd_spec(c, l, nt, j), v);
            for (const char* which : {"message_layer_norms", "update_layer_norms"}) {
                const std::string p = conv_prefix(l) + which + "." + kNtKey[nt] + ".feat_norm.";
                v.push_back({p + "weight", {S}});
                v.push_back({p + "bias", {S}});
            }
        }
    }
    for (int k = 0; k < c.n_noise_gvps; ++k) gvp_names(head_spec(c, k), v);
    v.push_back({"dynamics.noise_predictor.noise_predictor.to_scalar_output.weight", {c.pharm_nf, 64}});
    v.push_back({"dynamics.noise_predictor.noise_predictor.to_scalar_output.bias", {c.pharm_nf}});
    return v;
}

// ------------------------------------------------------------------------------------------------
// packing into MFMA A-operand fragment order (see pf_device.h "F-layout")
// ------------------------------------------------------------------------------------------------
static size_t push(std::vector<float>& w, const std::vector<float>& v) {
    while (w.size() % 64) w.push_back(0.f);     // 256-byte alignment of every block
    const size_t off = w.size();
    w.insert(w.end(), v.begin(), v.end());
    return off;
}

struct GvpOff { size_t wh, wu, wh_c, wu_c, a_main, a_main_c, b_main, a_gate, a_gate_c, b_gate; };

static GvpOff pack_gvp(pf_handle* h, const GvpSpec& g) {
    const int H = std::max(g.vi, g.vo);
    const int nextra = g.si - h->cfg.n_hidden_scalars;     // 16 (rbf) for the first message GVP
    const int NMO = g.so / 32;
    const int NSH = 8 + (g.vi == 17 ? 1 : 0);
    const int NKS = 64 + nextra / 2 + NSH;
    const int Kin = H + g.si;
    const RawTensor& W = h->raw[g.prefix + "to_feats_out.0.weight"];   // [so][si + H]
    const RawTensor& Bv = h->raw[g.prefix + "to_feats_out.0.bias"];
    const RawTensor& G = h->raw[g.prefix + "scalar_to_vector_gates.weight"];   // [vo][so]
    GvpOff o;
    {   // vector channel as A fragments.  k-step t < 8: lane half hl carries input channel u(t,hl) = rho(t,hl)
        // (for a 17-channel input that is Wh row 1 + u: row 0 is the unit x_diff, fed at k-step 8 by half 0).
        const std::vector<float>& wh = h->raw[g.prefix + "Wh"].data;      // [vi][H]
        const std::vector<float>& wu = h->raw[g.prefix + "Wu"].data;      // [H][vo]
        const bool X = g.vi == 17;
        const int NVK = 8 + (X ? 1 : 0);
        std::vector<float> awh((size_t)NVK * 64, 0.f), awu((size_t)NVK * 64, 0.f);
        for (int t = 0; t < NVK; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, hl = lane >> 5;
                int vin, hin;                      // input channel of Wh / of Wu at this k-step for this half
                if (t < 8) { vin = (X ? 1 : 0) + rho(t, hl); hin = rho(t, hl); }
                else { vin = hl == 0 ? 0 : -1; hin = hl == 0 ? 16 : -1; }
                awh[(size_t)t * 64 + lane] = (i < H && vin >= 0) ? wh[(size_t)vin * H + i] : 0.f;
                awu[(size_t)t * 64 + lane] = (i < g.vo && hin >= 0 && hin < H) ? wu[(size_t)hin * g.vo + i] : 0.f;
            }
        o.wh = push(h->h_w, awh);
        o.wu = push(h->h_w, awu);
        // four k-steps per lane for the 4-wave kernels: [t/4][lane][t%4]
        std::vector<float> c1((size_t)3 * 64 * 4, 0.f), c2((size_t)3 * 64 * 4, 0.f);
        for (int t = 0; t < NVK; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                c1[((size_t)(t / 4) * 64 + lane) * 4 + t % 4] = awh[(size_t)t * 64 + lane];
                c2[((size_t)(t / 4) * 64 + lane) * 4 + t % 4] = awu[(size_t)t * 64 + lane];
            }
        o.wh_c = push(h->h_w, c1);
        o.wu_c = push(h->h_w, c2);
    }
    std::vector<float> a((size_t)NKS * 64 * NMO, 0.f);
    for (int ks = 0; ks < NKS; ++ks)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, hl = lane >> 5;
            int col;
            if (ks < 64) col = 32 * (ks / 16) + rho(ks % 16, hl);
            else if (ks < 64 + nextra / 2) col = 128 + 2 * (ks - 64) + hl;
            else {                                   // sh block: k-step t carries sh[u(t,hl)]; t == 8: sh[16] on half 0
                const int t = ks - 64 - nextra / 2;
                const int idx = t < 8 ? rho(t, hl) : (hl == 0 ? 16 : -1);
                col = (idx >= 0 && idx < H) ? 128 + nextra + idx : -1;
            }
            for (int mo = 0; mo < NMO; ++mo) {
                const int row = 32 * mo + i;
                a[((size_t)ks * 64 + lane) * NMO + mo] = col >= 0 ? W.data[(size_t)row * Kin + col] : 0.f;
            }
        }
    o.a_main = push(h->h_w, a);
    {   // the same fragments per output tile, four k-steps per lane: [mo][ks/4][lane][ks%4]
        const int NKS4 = (NKS + 3) / 4;
        std::vector<float> ac((size_t)NMO * NKS4 * 64 * 4, 0.f);
        for (int ks = 0; ks < NKS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int mo = 0; mo < NMO; ++mo)
                    ac[(((size_t)mo * NKS4 + ks / 4) * 64 + lane) * 4 + ks % 4] = a[((size_t)ks * 64 + lane) * NMO + mo];
        o.a_main_c = push(h->h_w, ac);
    }
    std::vector<float> b((size_t)2 * NMO * 16);
    for (int hl = 0; hl < 2; ++hl)
        for (int mo = 0; mo < NMO; ++mo)
            for (int r = 0; r < 16; ++r) b[(size_t)hl * NMO * 16 + mo * 16 + r] = Bv.data[32 * mo + rho(r, hl)];
    o.b_main = push(h->h_w, b);
    std::vector<float> ag((size_t)NMO * 16 * 64, 0.f);
    for (int ks = 0; ks < NMO * 16; ++ks)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, hl = lane >> 5;
            const int k = 32 * (ks / 16) + rho(ks % 16, hl);
            ag[(size_t)ks * 64 + lane] = i < g.vo ? G.data[(size_t)i * g.so + k] : 0.f;      // rows 0..vo-1 = gates
        }
    o.a_gate = push(h->h_w, ag);
    {   // gate fragments of the wave owning output tile mo: k-steps 16*mo .. 16*mo+15 as [mo][r/4][lane][r%4]
        std::vector<float> agc((size_t)NMO * 4 * 64 * 4, 0.f);
        for (int mo = 0; mo < NMO; ++mo)
            for (int r = 0; r < 16; ++r)
                for (int lane = 0; lane < 64; ++lane)
                    agc[(((size_t)mo * 4 + r / 4) * 64 + lane) * 4 + r % 4] = ag[(size_t)(mo * 16 + r) * 64 + lane];
        o.a_gate_c = push(h->h_w, agc);
    }
    {   // gate bias in R-layout: half hl, register t <-> gate rho(t,hl)
        const std::vector<float>& bgv = h->raw[g.prefix + "scalar_to_vector_gates.bias"].data;
        std::vector<float> bg(16, 0.f);
        for (int hl = 0; hl < 2; ++hl)
            for (int t = 0; t < 8; ++t) bg[(size_t)hl * 8 + t] = rho(t, hl) < g.vo ? bgv[rho(t, hl)] : 0.f;
        o.b_gate = push(h->h_w, bg);
    }
    return o;
}

// ------------------------------------------------------------------------------------------------
// Quad stream of one GVP for the row-group kernels (pf_rg.hip; schedule: rg_sched in pf_device.h).  A quad is
// [64 lanes][4 images]; an image is what one lane holds of a B operand: lane f <-> output feature f (scalar Linear),
// lane 16g + u <-> output channel u of coordinate group g (vector products, gates; g = 3 unused -> 0).
// ------------------------------------------------------------------------------------------------
// the 8 gate quads of a GVP (or to_scalar_output): K split over the lane groups, quad m, image j <-> features
// 8 (4g + j) + m of lane group g (the A block (g, j) of the SA layout)
static void pack_gate_quads_rg(const std::vector<float>& Wg, int vo, int so, std::vector<float>& out, size_t base, int q0) {
    for (int lane = 0; lane < 64; ++lane) {
        const int gq = lane >> 4, u = lane & 15;
        for (int m = 0; m < 8; ++m)
            for (int j = 0; j < 4; ++j) {
                const int feat = 8 * (4 * gq + j) + m;
                out[base + ((size_t)(q0 + m) * 64 + lane) * 4 + j] = (u < vo && feat < so) ? Wg[(size_t)u * so + feat] : 0.f;
            }
    }
}
// one block of a chain: GVP g, plus the gate quads of the GVP before it (prev) when there is one
// half >= 0: the block of wave `half` of the two-wave form (outputs 64 half .. 64 half + 63 of a 128-output scalar
// Linear; a 64-output GVP is the same block for both waves)
static void pack_gvp_rg(pf_handle* h, const GvpSpec& g, const GvpSpec* prev, std::vector<float>& out, int half = -1) {
    const int S = h->cfg.n_hidden_scalars;
    const int H = std::max(g.vi, g.vo);
    const int nextra = g.si - S;
    const bool split = half >= 0 && g.so == 128;
    const int NH = split ? 1 : g.so / 64;             // halves of 64 outputs in this block
    const int f0 = split ? 64 * half : 0;             // first output feature of the block
    const int so_end = split ? f0 + 64 : g.so;
    const bool X17 = g.vi == 17;
    const RgSched q = rg_sched(g.vi, nextra, NH, prev != nullptr);
    const int Kin = H + g.si;
    const std::vector<float>& W = h->raw[g.prefix + "to_feats_out.0.weight"].data;            // [so][si + H]
    const std::vector<float>& Bv = h->raw[g.prefix + "to_feats_out.0.bias"].data;
    const std::vector<float>& bg = h->raw[g.prefix + "scalar_to_vector_gates.bias"].data;
    const std::vector<float>& wh = h->raw[g.prefix + "Wh"].data;                              // [vi][H]
    const std::vector<float>& wu = h->raw[g.prefix + "Wu"].data;                              // [H][vo]
    const size_t base = out.size();
    out.resize(base + (size_t)q.nq * 256, 0.f);
    auto at = [&](int quad, int lane, int j) -> float& { return out[base + ((size_t)quad * 64 + lane) * 4 + j]; };
    const int v0 = X17 ? 1 : 0;                      // Wh row of node-vector channel 0 (row 0 is the unit x_diff)
    for (int lane = 0; lane < 64; ++lane) {
        const int gq = lane >> 4, u = lane & 15, qq = (lane >> 2) & 3;
        // constants: scalar bias (two halves), gate bias, Wh[0][16] on the lanes that carry xhat
        at(q.q_c, lane, 0) = f0 + lane < so_end ? Bv[f0 + lane] : 0.f;
        at(q.q_c, lane, 1) = f0 + 64 + lane < so_end ? Bv[f0 + 64 + lane] : 0.f;
        at(q.q_c, lane, 2) = u < g.vo ? bg[u] : 0.f;
        at(q.q_c, lane, 3) = (X17 && qq == 0 && gq < 3) ? wh[(size_t)0 * H + 16] : 0.f;
        if (X17) {
            at(q.q_xh, lane, 0) = gq < 3 ? wh[(size_t)0 * H + u] : 0.f;                              // xhat k-step of Vh
            at(q.q_xh, lane, 1) = (gq < 3 && u < g.vo) ? wu[(size_t)16 * g.vo + u] : 0.f;           // Vh[16] k-step of Vu
            at(q.q_xh, lane, 2) = f0 + lane < so_end ? W[(size_t)(f0 + lane) * Kin + g.si + 16] : 0.f;           // sh[16] column
            at(q.q_xh, lane, 3) = f0 + 64 + lane < so_end ? W[(size_t)(f0 + 64 + lane) * Kin + g.si + 16] : 0.f;
            for (int t = 0; t < 4; ++t) at(q.q_xh + 1, lane, t) = gq < 3 ? wh[(size_t)(1 + 4 * t + qq) * H + 16] : 0.f;
        }
        for (int t = 0; t < 4; ++t)
            for (int j = 0; j < 4; ++j) {
                at(q.q_vh + t, lane, j) = gq < 3 ? wh[(size_t)(v0 + 4 * t + j) * H + u] : 0.f;
                at(q.q_vu + t, lane, j) = (gq < 3 && u < g.vo) ? wu[(size_t)(4 * t + j) * g.vo + u] : 0.f;
            }
        for (int hh = 0; hh < NH; ++hh) {
            const int f = f0 + hh * 64 + lane;
            for (int m = 0; m < 8; ++m)
                for (int aq = 0; aq < 4; ++aq)
                    for (int j = 0; j < 4; ++j)
                        at(rg_main_quad(q, NH, (m * 4 + aq) * NH + hh), lane, j) = W[(size_t)f * Kin + 8 * (4 * aq + j) + m];
            for (int aq = 0; aq < 4; ++aq)
                for (int j = 0; j < 4; ++j) {
                    if (nextra) at(q.q_rbf + aq * NH + hh, lane, j) = W[(size_t)f * Kin + S + 4 * aq + j];
                    at(q.q_sh + aq * NH + hh, lane, j) = W[(size_t)f * Kin + g.si + 4 * aq + j];
                }
        }
    }
    if (prev) pack_gate_quads_rg(h->raw[prev->prefix + "scalar_to_vector_gates.weight"].data, prev->vo, prev->so, out, base, q.q_gate);
}
// end of a chain: the gate quads of its last GVP
static void pack_flush_rg(pf_handle* h, const GvpSpec& g, std::vector<float>& out) {
    const size_t base = out.size();
    out.resize(base + (size_t)RG_NQ_FLUSH * 256, 0.f);
    pack_gate_quads_rg(h->raw[g.prefix + "scalar_to_vector_gates.weight"].data, g.vo, g.so, out, base, 0);
}
// to_scalar_output (Linear 64 -> pharm_nf) as a gate-like product: [const] [8 quads] [pad]
static void pack_out_rg(pf_handle* h, std::vector<float>& out) {
    const RawTensor& W = h->raw["dynamics.noise_predictor.noise_predictor.to_scalar_output.weight"];   // [pharm_nf][64]
    const RawTensor& Bv = h->raw["dynamics.noise_predictor.noise_predictor.to_scalar_output.bias"];
    const int nf = h->cfg.pharm_nf;
    const size_t base = out.size();
    out.resize(base + (size_t)RG_NQ_OUT * 256, 0.f);
    auto at = [&](int quad, int lane, int j) -> float& { return out[base + ((size_t)quad * 64 + lane) * 4 + j]; };
    for (int lane = 0; lane < 64; ++lane) {
        const int gq = lane >> 4, u = lane & 15;
        at(0, lane, 0) = u < nf ? Bv.data[u] : 0.f;
        for (int m = 0; m < 8; ++m)
            for (int j = 0; j < 4; ++j) {
                const int feat = 8 * (4 * gq + j) + m;
                at(1 + m, lane, j) = (u < nf && feat < 64) ? W.data[(size_t)u * 64 + feat] : 0.f;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// n16 kernels (pf_n16.hip; schedules: n16_sched in pf_device.h): wave w's block of GVP g.  An image is one lane's share
// of an A operand of v_mfma_f32_16x16x4_f32: lane 16 gq + i <-> output row i of the tile, k = gq of the k-step.
// Scalar k-step ks <-> input feature 16 (ks >> 2) + 4 gq + (ks & 3); vector / rbf / sh k-step r <-> channel 4 gq + r.
// ------------------------------------------------------------------------------------------------
struct N16Raw {                                  // the six tensors of a GVP with 128 scalar and 16 vector outputs (g: its dimensions)
    const std::vector<float>&W, &Bv, &Wg, &bg, &wh, &wu;
    GvpSpec g;
};
// plane p (0..2) of x = p0 + p1 + p2 as a bf16 bit pattern: round-to-nearest-even of what the earlier planes left (the device's
// n16_split8 / k_n16_split_words do the same arithmetic)
[[maybe_unused]] static uint32_t n16_bf16_plane(float x, int p) {
    uint32_t bits = 0;
    for (int k = 0; k <= p; ++k) {
        uint32_t u; memcpy(&u, &x, 4);
        bits = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
        const uint32_t hi = bits << 16;
        float back; memcpy(&back, &hi, 4);
        x -= back;
    }
    return bits & 0xffffu;
}
static void pack_n16_raw(const N16Raw& rw, int kind, int w, std::vector<float>& out, std::vector<int4>* split_rec = nullptr) {
    const GvpSpec& g = rw.g;
    const N16Sched q = n16_sched(kind);
    const int H = std::max(g.vi, g.vo), Kin = H + g.si;
    const bool m0 = kind != N16_GEN, vz = kind == N16_M0Z || kind == N16_M0H;
    const int v0 = g.vi == 17 ? 1 : 0;
    const std::vector<float>& W = rw.W;             // [so][si + H]
    const std::vector<float>& Bv = rw.Bv;
    const std::vector<float>& Wg = rw.Wg;           // [vo][so]
    const std::vector<float>& bg = rw.bg;
    const std::vector<float>& wh = rw.wh;           // [vi][H]
    const std::vector<float>& wu = rw.wu;           // [H][vo]
    const size_t base = out.size();
    out.resize(base + (size_t)q.nq * 256, 0.f);
    auto at = [&](int quad, int lane, int j) -> float& { return out[base + ((size_t)quad * 64 + lane) * 4 + j]; };
    for (int lane = 0; lane < 64; ++lane) {
        const int gq = lane >> 4, i = lane & 15;
        const int f0 = 32 * w + i, f1 = 32 * w + 16 + i;             // this lane's output rows of tile 0 / tile 1
        if (m0) {
            at(q.q_x1, lane, 0) = gq == 0 ? W[(size_t)f0 * Kin + g.si + 16] : 0.f;
            at(q.q_x1, lane, 1) = gq == 0 ? W[(size_t)f1 * Kin + g.si + 16] : 0.f;
            at(q.q_x1, lane, 2) = (gq == 0 && w < 3) ? wu[(size_t)16 * g.vo + i] : 0.f;
            at(q.q_x1, lane, 3) = gq == 0 ? wh[(size_t)0 * H + i] : (lane == 16 ? wh[(size_t)0 * H + 16] : 0.f);
        }
        for (int r = 0; r < 4; ++r) {
            if (vz) at(q.q_vh, lane, r) = wh[(size_t)0 * H + 4 * gq + r];                     // Vh = Wh[0] (x) xhat
            else if (w < 3) at(q.q_vh, lane, r) = wh[(size_t)(v0 + 4 * gq + r) * H + i];
            if (q.q_w16 >= 0 && w < 3) at(q.q_w16, lane, r) = wh[(size_t)(v0 + 4 * gq + r) * H + 16];
            at(q.q_vu, lane, r) = w < 3 ? wu[(size_t)(4 * gq + r) * g.vo + i] : bg[4 * gq + r];
        }
#if N16_SPLIT
        // main quad m = plane m % 3 of the weights of output tile (m / 3) % 2 over K chunk m / 6; a 32-bit word = elements 2 d, 2 d + 1
        if (q.q_main >= 0)
            for (int qm = 0; qm < N16_NQM; ++qm) {
                const int cch = qm / 6, tt = (qm / 3) % 2, pl = qm % 3;
                const int frow = tt ? f1 : f0;
                for (int d = 0; d < 4; ++d) {
                    float v[2];
                    for (int k = 0; k < 2; ++k) {
                        const int e = 2 * d + k;
                        v[k] = W[(size_t)frow * Kin + 16 * (2 * cch + e / 4) + 4 * gq + e % 4];
                    }
                    float& word = at(q.main_pos(qm), lane, d);
                    if (split_rec) {                 // index-valued pass: v = flat index + 1 (0: a padded row)
                        word = 0.f;
                        split_rec->push_back(make_int4((int)(&word - out.data()), (int)v[0] - 1, (int)v[1] - 1, pl));
                    } else {
                        const uint32_t bits = n16_bf16_plane(v[0], pl) | (n16_bf16_plane(v[1], pl) << 16);
                        memcpy(&word, &bits, 4);
                    }
                }
            }
#else
        if (q.q_main >= 0)
            for (int qm = 0; qm < 16; ++qm)
                for (int half = 0; half < 2; ++half) {
                    const int ks = 2 * qm + half, f = 16 * (ks >> 2) + 4 * gq + (ks & 3);
                    at(q.main_pos(qm), lane, 2 * half) = W[(size_t)f0 * Kin + f];
                    at(q.main_pos(qm), lane, 2 * half + 1) = W[(size_t)f1 * Kin + f];
                }
#endif
        for (int qq = 0; qq < 2; ++qq)
            for (int half = 0; half < 2; ++half) {
                const int r = 2 * qq + half;
                if (m0) {
                    at(q.q_rbf + qq, lane, 2 * half) = W[(size_t)f0 * Kin + PF_S + 4 * gq + r];
                    at(q.q_rbf + qq, lane, 2 * half + 1) = W[(size_t)f1 * Kin + PF_S + 4 * gq + r];
                }
                at(q.q_sh + qq, lane, 2 * half) = W[(size_t)f0 * Kin + g.si + 4 * gq + r];
                at(q.q_sh + qq, lane, 2 * half + 1) = W[(size_t)f1 * Kin + g.si + 4 * gq + r];
            }
        if (q.q_b >= 0)
            for (int r = 0; r < 4; ++r) {
                at(q.q_b, lane, r) = Bv[32 * w + 4 * gq + r];
                at(q.q_b + 1, lane, r) = Bv[32 * w + 16 + 4 * gq + r];
            }
        for (int t = 0; t < 2; ++t)
            for (int r = 0; r < 4; ++r) at(q.q_gate + t, lane, r) = Wg[(size_t)i * g.so + 32 * w + 16 * t + 4 * gq + r];
    }
}
static void pack_n16(pf_handle* h, const GvpSpec& g, int kind, int w, std::vector<float>& out) {
    const N16Raw rw{h->raw[g.prefix + "to_feats_out.0.weight"].data, h->raw[g.prefix + "to_feats_out.0.bias"].data,
                    h->raw[g.prefix + "scalar_to_vector_gates.weight"].data, h->raw[g.prefix + "scalar_to_vector_gates.bias"].data,
                    h->raw[g.prefix + "Wh"].data, h->raw[g.prefix + "Wu"].data, g};
    pack_n16_raw(rw, kind, w, out, h->split_record ? &h->split_pending : nullptr);
}
// The noise head's last GVP (dynamics_gvp.py:17-20: 16 vectors -> 1, 128 scalars -> 64, identity vector gate) followed by
// to_scalar_output (Linear 64 -> pharm_nf, :35,39) as ONE GEN block of the tail kernel: the GVP zero-padded to 128 scalar
// and 16 vector outputs (SiLU(0) = 0: the padded scalars feed nothing), and to_scalar_output -- a Linear on the same SiLU
// output as the gate Linear -- in the unused gate rows 1 .. pharm_nf (bias in the gate bias).  Pure data movement, like
// every packing here (the gather map of pf_set_flat_params covers it).  Needs pharm_nf <= 15.
static void pack_n16_head_last(pf_handle* h, const GvpSpec& g, int w, std::vector<float>& out) {
    const int nf = h->cfg.pharm_nf, H = std::max(g.vi, g.vo), Kin = H + g.si;
    const std::vector<float>& W = h->raw[g.prefix + "to_feats_out.0.weight"].data;            // [64][128 + 16]
    const std::vector<float>& Bv = h->raw[g.prefix + "to_feats_out.0.bias"].data;
    const std::vector<float>& Wg = h->raw[g.prefix + "scalar_to_vector_gates.weight"].data;   // [1][64]
    const std::vector<float>& bg = h->raw[g.prefix + "scalar_to_vector_gates.bias"].data;
    const std::vector<float>& wu = h->raw[g.prefix + "Wu"].data;                              // [16][1]
    const std::vector<float>& Wo = h->raw["dynamics.noise_predictor.noise_predictor.to_scalar_output.weight"].data;   // [nf][64]
    const std::vector<float>& bo = h->raw["dynamics.noise_predictor.noise_predictor.to_scalar_output.bias"].data;
    std::vector<float> W2((size_t)PF_S * Kin, 0.f), B2(PF_S, 0.f), G2((size_t)16 * PF_S, 0.f), bg2(16, 0.f), U2((size_t)H * 16, 0.f);
    for (int f = 0; f < g.so; ++f) {
        for (int k = 0; k < Kin; ++k) W2[(size_t)f * Kin + k] = W[(size_t)f * Kin + k];
        B2[f] = Bv[f];
        G2[f] = Wg[f];                                                                         // gate row 0
        for (int k = 0; k < nf; ++k) G2[(size_t)(1 + k) * PF_S + f] = Wo[(size_t)k * g.so + f];
    }
    bg2[0] = bg[0];
    for (int k = 0; k < nf; ++k) bg2[1 + k] = bo[k];
    for (int c = 0; c < H; ++c) U2[(size_t)c * 16] = wu[(size_t)c * g.vo];
    GvpSpec g2 = g;
    g2.vo = 16; g2.so = PF_S;
    const N16Raw rw{W2, B2, G2, bg2, h->raw[g.prefix + "Wh"].data, U2, g2};
    pack_n16_raw(rw, N16_GEN, w, out, h->split_record ? &h->split_pending : nullptr);
}

// keep_ws: the inference workspace stays allocated (pf_set_pocket_batch re-carves it when the next batch fits: a
// hipMalloc / hipFree pair of a few hundred MB per batch costs milliseconds)
static void free_ws(pf_handle* h, bool keep_ws = false) {
    if (h->d_ws && !keep_ws) { (void)hipFree(h->d_ws); h->d_ws = nullptr; h->ws_capacity = 0; h->d_xchg = nullptr; h->d_lpart = nullptr; h->d_xchg2 = nullptr; h->d_cen_h = h->d_cen_p = nullptr; h->d_snap[0] = h->d_snap[1] = nullptr; h->d_pa_stamp = h->d_pa_same = nullptr; }
    if (h->d_tws && !keep_ws) { (void)hipFree(h->d_tws); h->d_tws = nullptr; h->tws_capacity = 0; }
    if (h->d_tA && !keep_ws) { (void)hipFree(h->d_tA); h->d_tA = nullptr; h->tA_capacity = 0; }
    h->t_ws_ready = false;
    h->t_have_fwd = false;
    h->t_mask_override = nullptr;
    h->have_batch = false;
}

template <typename T>
static T* carve(char*& cur, size_t count) {
    T* p = reinterpret_cast<T*>(cur);
    size_t bytes = count * sizeof(T);
    bytes = (bytes + 255) & ~size_t(255);
    cur += bytes;
    return p;
}

struct ProfScope {
    pf_handle* h; int k; hipStream_t s; bool on;
    ProfScope(pf_handle* h_, int k_, hipStream_t s_) : h(h_), k(k_), s(s_), on((h_->prof_mask >> k_) & 1u) {
        if (!on) return;
        if (h->prof_used[k] == h->prof_ev[k].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            h->prof_ev[k].push_back({a, b});
        }
        (void)hipEventRecord(h->prof_ev[k][h->prof_used[k]].first, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(h->prof_ev[k][h->prof_used[k]].second, s);
        h->prof_used[k]++;
    }
};

static bool l0_hoist_ok(pf_handle* h);
// pocket sharing applies to inference calls at one common t whose conv layer 0 is the pruned layer under the static hoist
static bool share_now(pf_handle* h) {
    const pf_config& c = h->cfg;
    if (h->share_ok && h->share_check == 0) {        // device rows: the comparison's verdict arrived with the one-hot flag
        if (h->l0flag_ev) (void)hipEventSynchronize(h->l0flag_ev);
        h->share_check = (h->l0flag_host && h->l0flag_host[1] == 0) ? 1 : 2;
    }
    if (h->share_check == 2) return false;           // (run_dynamics fails the call: a false claim is the caller's bug, not a mode)
    return h->share_ok && !h->share_disable && h->prune && c.n_convs == 2 && h->rg_compact && l0_hoist_ok(h);
}
// a build with these parameters has been enqueued: its stamp is what the next shared edge launch looks for
static void build_done(pf_handle* h, bool share, bool with_records = false) {
    h->rec_valid = with_records;
    h->edges_share = share;
    if (share) h->edges_stamp = ++h->need_stamp;
}
static BuildParams build_params(pf_handle* h, bool share = false) {
    const pf_config& c = h->cfg;
    BuildParams bp{};
    bp.B = h->B; bp.Np_tot = h->Np;
    bp.prot_ptr = h->d_prot_ptr; bp.pharm_ptr = h->d_pharm_ptr; bp.xn = h->d_xn;
    bp.reg = h->d_reg; bp.dyn_cnt = h->d_dyn_cnt; bp.esrc = h->d_esrc; bp.edst = h->d_edst;
    bp.in_start = h->d_in_start; bp.in_cnt = h->d_in_cnt; bp.N = h->N;
    bp.ff_k = c.ff_k; bp.pf_k = c.pf_k;
    bp.r2_ff = c.cutoff_ff * c.cutoff_ff; bp.r2_pf = c.cutoff_pf * c.cutoff_pf;
    bp.gnorm = h->d_gnorm; bp.pp_cnt = h->d_pp_cnt; bp.pfq_cnt = h->d_pfq_cnt; bp.norm_mode = c.message_norm_mode;
    if (h->pa_spec && h->sampling && h->d_pa_stamp && !share) { bp.pa_stamp = h->d_pa_stamp; bp.step_id = h->step_id; bp.pa_same = h->d_pa_same; }
    const int prune_layer = (h->prune && c.n_convs >= 2) ? c.n_convs - 2 : -1;
    bp.act_ids = prune_layer >= 0 ? h->d_act_ids : nullptr; bp.reg_act = h->d_reg_act;
    bp.eorig = h->d_eorig;
    bp.pa_static = share ? h->d_pa_static : nullptr;
    if (share) { bp.rep_base = h->d_rep_base; bp.need = h->d_need; bp.need_stamp = h->need_stamp + 1; }   // committed by build_done()
    return bp;
}
// conv layer 0 of an inference call runs on the row-group kernels: they encode the rows they read on the fly, so the
// call's first launch is the edge build alone -- which the previous denoising step's update launch can do as well
static bool encoders_on_the_fly(const pf_handle* h) {
    const pf_config& c = h->cfg;
    const int prune_layer = (h->prune && c.n_convs >= 2) ? c.n_convs - 2 : -1;
    const int nt0 = c.n_convs == 1 ? h->n_edge_tiles_last : (prune_layer == 0 ? h->n_edge_tiles_act : h->n_edge_tiles);
    return h->enc_on_the_fly && h->pol.rg_mode(nt0) != 0;
}

// ---- static hoist of conv layer 0 (pf_rg.hip: EdgeParams::zs) ----------------------------------------------------
// usable for this handle / batch at all (inference, row-group kernels with encoders on the fly)
// The device-side one-hot check of pf_set_pocket_batch is read back lazily: only an inference call that could use the
// static hoist waits for it (callers that pass the element types themselves never wait).
static void l0_resolve_onehot(pf_handle* h) {
    if (h->l0_state != 0) return;
    if (h->l0flag_ev) (void)hipEventSynchronize(h->l0flag_ev);
    h->l0_state = (h->Np > 0 && h->l0flag_host && *h->l0flag_host == 0) ? 1 : 2;
    h->l0_onehot = h->l0_state == 1;
}
static bool l0_hoist_ok(pf_handle* h) {
    if (h->l0_hoist && h->l0_state == 0) l0_resolve_onehot(h);
    const pf_config& c = h->cfg;
    return h->l0_hoist && h->l0_onehot && h->Epp > 0 && c.n_message_gvps >= 2 && c.rbf_dim == PF_R && c.rec_nf < 128 &&
           encoders_on_the_fly(h);
}
static L0HoistParams l0_params(pf_handle* h) {
    const pf_config& c = h->cfg;
    L0HoistParams lp{};
    lp.src = h->d_w + h->l0h_off; lp.l0c = h->d_l0c;
    lp.esrc = h->d_esrc; lp.edst = h->d_edst; lp.xn = h->d_xn; lp.Epp = (int)h->Epp; lp.zs = h->d_zs;
    float mu[PF_R];
    linspace_f32(0.f, c.rbf_dmax, c.rbf_dim, mu);
    lp.rbf_mu0 = mu[0]; lp.rbf_mu_step = (mu[PF_R - 1] - mu[0]) * (1.0f / (float)(PF_R - 1));
    lp.rbf_inv_sigma = 1.0f / ((c.rbf_dmax - 0.f) / (float)c.rbf_dim);
    lp.enc_w = h->d_w + h->enc_w[0]; lp.enc_b = h->d_w + h->enc_b[0];
    lp.enc_lw = h->d_w + h->enc_lw[0]; lp.enc_lb = h->d_w + h->enc_lb[0];
    lp.rec_nf = c.rec_nf;
    return lp;
}
// the trajectory constants (zs, weff) of the current protein coordinates and weights
static void l0_ensure_static(pf_handle* h, hipStream_t s) {
    if (!h->coords_custom && h->zs_batch_coords && h->zs_version == h->w_version) return;
    const L0HoistParams lp = l0_params(h);
    pfk_l0_hoist(&lp, 0, s);
    h->zs_version = h->w_version;
    h->zs_batch_coords = !h->coords_custom;
}
// type tables of n timesteps (scalar-t calls): slots of the resident cache, computed on a miss
static void l0_prepare_t(pf_handle* h, const float* tv, int n, hipStream_t s) {
    if (h->ptab_version != h->w_version) { h->ptab_slot.clear(); h->ptab_version = h->w_version; }
    const size_t slot_floats = (size_t)L0_NTAB * h->cfg.rec_nf * PF_S;
    int i = 0;
    while (i < n) {
        L0HoistParams lp = l0_params(h);
        if (h->ptab_slot.size() + 64 > L0_PTAB_SLOTS) h->ptab_slot.clear();     // stream order keeps earlier launches valid
        const int slot0 = (int)h->ptab_slot.size();
        int m = 0;
        for (; i < n && m < 64; ++i) {
            uint32_t bits; memcpy(&bits, &tv[i], 4);
            if (h->ptab_slot.count(bits)) continue;
            h->ptab_slot[bits] = slot0 + m;
            lp.t_host[m++] = tv[i];
        }
        if (m == 0) continue;
        lp.nt = m; lp.t_dev = nullptr; lp.ptab = h->d_ptab + (size_t)slot0 * slot_floats;
        pfk_l0_hoist(&lp, 1, s);
    }
}

// sequence one dynamics call on the handle's state (xn, pharm_h, d_t).  train: keep every layer's input and message
// rows (h->t_*), compute every tile (gradients need the full graph only where they are non-zero, but the first
// version of the backward pass walks the dense tile lists) and apply dropout in the node update.
// the n16 streams after pf_set_flat_params left them behind (see pf_handle::n16_begin)
static void n16_refresh(pf_handle* h, hipStream_t s) {
    if (!h->n16_stale) return;
    pfk_gather_weights(h->d_flat, h->d_map + h->n16_begin, h->n_packed - h->n16_begin, h->d_w + h->n16_begin, s);
    if (h->n_split_tab) pfk_n16_split_words(h->d_flat, h->d_split_tab, h->n_split_tab, h->d_w, s);     // (-DN16_SPLIT: the bf16-plane words)
    h->n16_stale = false;
}

// step: this call is the dynamics call of a denoising step (pf_denoise_step) -- when the tail launch applies, the step's
// sampler update and edge build run behind the noise head in the same launch and h->tail_done tells the caller
static int run_dynamics(pf_handle* h, float* eps_h, float* eps_x, hipStream_t s, const float* t_scalar = nullptr,
                        bool train = false, const StepParams* step = nullptr) {
    const pf_config& c = h->cfg;
    h->tail_done = false; h->last_tail = 0;
    // center hoist: the previous denoising step left h_c and P_ff / P_fp of every center for THIS call's timestep
    const bool cen_have = !train && h->cen_valid && h->edges_built && t_scalar != nullptr && *t_scalar == h->cen_t && h->cen_wver == h->w_version;
    h->cen_valid = false; h->last_cen = false;
    // rows computed ahead for this call's "pa" regions (the speculative items of the previous step's merged launch)
    const bool spec_have = !train && h->spec_valid && h->edges_built && t_scalar != nullptr && *t_scalar == h->spec_t && h->spec_wver == h->w_version;
    h->spec_valid = false; h->last_spec = 0; h->e0_saved = false;
    if (!train) n16_refresh(h, s);
    EncodeParams ep{};
    ep.Np = h->Np; ep.Nf = h->Nf;
    ep.prot_h0 = h->d_prot_h0; ep.pharm_h = h->d_pharm_h; ep.t = t_scalar ? nullptr : h->d_t; ep.gid = h->d_gid;
    ep.t_scalar = t_scalar ? *t_scalar : 0.f;
    ep.rec_nf = c.rec_nf; ep.pharm_nf = c.pharm_nf;
    for (int nt = 0; nt < 2; ++nt) {
        ep.w[nt] = h->d_w + h->enc_w[nt]; ep.b[nt] = h->d_w + h->enc_b[nt];
        ep.ln_w[nt] = h->d_w + h->enc_lw[nt]; ep.ln_b[nt] = h->d_w + h->enc_lb[nt];
    }
    ep.h_out = train ? h->t_H[0] : h->d_h[0];

    const int prune_layer = (h->prune && c.n_convs >= 2) ? c.n_convs - 2 : -1;    // the layer restricted to active atoms
    // conv layer 0 on the row-group kernels: they encode the rows they read on the fly, only the edge build is launched
    // -- unless the previous denoising step's update launch has built the edges of these coordinates already
    const bool enc_fly = !train && encoders_on_the_fly(h);
    // pocket sharing: copies of a pocket read one set of layer-0 pp messages (calls at one common t only)
    if (h->share_ok && h->share_check == 0) (void)share_now(h);                    // a claim about device rows: its verdict, whatever this call shares
    const bool share = enc_fly && t_scalar != nullptr && prune_layer == 0 && share_now(h);
    if (h->share_check == 2)
        PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups: the claim made for this batch is false -- a graph differs from its representative in "
                               "coordinates or features (compared on the device); bind the batch again without the claim");
    if (h->edges_built && h->edges_share != share) h->edges_built = false;       // built for the other mode: rebuild
    BuildParams bp = build_params(h, share);
    bool pre_ready = false;
    if (enc_fly) {
        if (!h->edges_built) { ProfScope ps(h, pf_handle::K_BUILD, s); pfk_build_edges(&bp, s); build_done(h, share); }
    }
    else if (h->prof_mask & 3u) {     // timing the two halves separately needs separate launches
        { ProfScope ps(h, pf_handle::K_ENCODE, s); pfk_encode(&ep, s); }
        { ProfScope ps(h, pf_handle::K_BUILD, s); pfk_build_edges(&bp, s); }
    } else if (h->use_pre && h->Np > 0 && prune_layer != 0) {      // layer 0 is dense: precompute P for its pp messages
        PreParams pp{};
        pp.Np = h->Np; pp.rec_nf = c.rec_nf; pp.nke = (c.rec_nf + 2) / 2;
        pp.prot_h0 = h->d_prot_h0; pp.t = ep.t; pp.t_scalar = ep.t_scalar; pp.gid = h->d_gid;
        pp.a_enc = h->d_w + h->enc_a; pp.b_enc = h->d_w + h->enc_bf;
        pp.ln_w = h->d_w + h->enc_lw[0]; pp.ln_b = h->d_w + h->enc_lb[0];
        pp.pre_w = h->h_gvp[h->msg_base(0, ET_PP)]; pp.pre_nks = 64 + c.rbf_dim / 2 + 9;
        pp.h_out = ep.h_out; pp.pre_out = h->d_pre;
        pfk_encode_build_pre(&ep, &bp, &pp, s);
        pre_ready = true;
    } else pfk_encode_build(&ep, &bp, s);

    // conv layer 0: static hoist of the pp messages (trajectory constants + per-timestep type table)
    const bool hoist = !train && enc_fly && l0_hoist_ok(h);
    const float* l0_ptab = nullptr;
    int l0_gstride = 0;
    // the n16 form serves the latency regime: batches whose pruned conv-layer launch has few items per compute unit
    const bool n16_batch = (long)(prune_layer >= 0 ? h->n_edge_tiles_act : h->n_edge_tiles) * 32 <= h->pol.n16_rows_max && !h->n16_msg.empty();
    const bool n16_l0 = hoist && (h->pol.n16_mask & 2) && n16_batch;      // layer 0 on the n16 kernels: no zs
    if (hoist) {
        if (!n16_l0) l0_ensure_static(h, s);
        if (t_scalar) {
            l0_prepare_t(h, t_scalar, 1, s);
            uint32_t bits; memcpy(&bits, t_scalar, 4);
            l0_ptab = h->d_ptab + (size_t)h->ptab_slot[bits] * L0_NTAB * c.rec_nf * PF_S;
        } else {
            L0HoistParams lp = l0_params(h);
            lp.nt = h->B; lp.t_dev = h->d_t; lp.ptab = h->d_ptab_pg;
            pfk_l0_hoist(&lp, 1, s);
            l0_ptab = h->d_ptab_pg; l0_gstride = L0_NTAB * c.rec_nf * PF_S;
        }
    }
    h->last_hoist = 0;
    int cur = 0;
    bool head_done = false;
    // fused launch (pf_n16.hip: k_n16_fused): with two conv layers, receptive-field pruning and kNN pf edges the rows conv
    // layer 0's node update produces are exactly the sources of the last layer's edges (+ the centers): every edge item of
    // the last layer updates its own source rows first, and the node launch of conv layer 0 disappears
    const bool fuse_l0node = !train && n16_batch && (long)h->n_edge_tiles_act * 32 <= h->pol.n16_fuse_rows_max && (h->pol.n16_mask & 4) && (h->pol.n16_mask & 1) && hoist && c.n_convs == 2 && prune_layer == 0 && c.pf_k > 0 &&
                             c.n_update_gvps >= 1 && h->rg_compact && 2 * h->B <= 1024 && h->d_msg_s2 != nullptr && h->n16_fused[0] != 0;
    FusedParams fz{};
    for (int l = 0; l < c.n_convs; ++l) {
        EdgeParams e{};
        const bool last = (l == c.n_convs - 1), pruned = (l == prune_layer);
        e.tiles = pruned ? h->d_edge_tiles_act : h->d_edge_tiles;
        e.ntiles = last ? h->n_edge_tiles_last : (pruned ? h->n_edge_tiles_act : h->n_edge_tiles); e.dyn_cnt = h->d_dyn_cnt;
        e.esrc = h->d_esrc; e.edst = h->d_edst; e.xn = h->d_xn;
        e.h = train ? h->t_H[l] : h->d_h[cur]; e.v = train ? h->t_V[l] : h->d_v[cur];
        e.msg_s = train ? h->t_msg_s[l] : h->d_msg_s; e.msg_v = train ? h->t_msg_v[l] : h->d_msg_v;
        if (fuse_l0node && l == 1) { e.msg_s = h->d_msg_s2; e.msg_v = h->d_msg_v2; }
        e.w = h->d_gvp + h->msg_base(l, 0); e.n_gvps = c.n_message_gvps;
        e.pre = (l == 0 && pre_ready) ? h->d_pre : nullptr;
        linspace_f32(0.f, c.rbf_dmax, c.rbf_dim, e.rbf_mu);
        e.rbf_inv_sigma = 1.0f / ((c.rbf_dmax - 0.f) / (float)c.rbf_dim);
        // few tiles (last layer): 4 waves per tile to cut the serial latency; otherwise one wave per tile
        if (train) {
            e.sv_z = h->t_sv_z[l]; e.sv_g = h->t_sv_g[l]; e.sv_v = h->t_sv_v[l];
            e.sv_stride = (size_t)std::max<int64_t>(h->Ecap, 1);
        }
        for (int et = 0; et < 4; ++et) e.rg[et] = h->d_w + h->rg_msg[(size_t)l * 4 + et];
        // every edge of this launch lives in a dynamic region (each wave scans the region lengths: up to 1024 regions = 64 * RG_CPASS)
        const bool shared = share && pruned && l == 0;       // pocket sharing: kind-3 regions = static ranges of the representatives
        // capacity of region r in groups of gs slots (a shared kind-3 region is cut on absolute multiples of gs)
        auto region_groups = [&](int r, int gs) {
            if (shared && r >= 3 * h->B) {
                const int st = h->h_share_start[r - 3 * h->B], cn = h->h_share_cnt[r - 3 * h->B];
                return cn > 0 ? (st + cn - 1) / gs - st / gs + 1 : 0;
            }
            return (h->h_cap[r] + gs - 1) / gs;
        };
        if ((last || pruned) && h->rg_compact && (last ? 2 : 4) * h->B <= 1024) {
            e.reg = shared ? h->d_reg_share : h->d_reg; e.regB = h->B; e.nreg = (last ? 2 : 4) * h->B;
            e.pa_abs = shared ? 1 : 0;
            if (shared) { e.need = h->d_need; e.need_stamp = h->edges_stamp; }
            for (int r = 0; r < e.nreg; ++r) { e.ngroups4 += region_groups(r, 4); e.ngroups8 += region_groups(r, 8); }
        }
        // (bf16 leg: the message chains run on the 32-slot tile kernel, whose to_feats_out / gate products have a bf16 form)
        if (train && h->train_bf16) e.bf16 = 1;
        int rg = train ? ((h->train_rg_edge && h->train_rg_node && !h->train_bf16) ? h->pol.rg_mode(e.ntiles) : 0)
                       : h->pol.rg_mode(shared ? (int)((h->share_rows + 31) / 32) : e.ntiles);           // the node launch of this layer follows (partial-row grouping)
        // static hoist: the hoisted ("pa") items of a compact layer-0 launch run a two-block chain and may take 8 rows
        // per wave while the full-chain items (ff, pf, fp) take 4
        int rgp = 0;
        if (hoist && l == 0 && rg) {
            e.zs = h->d_zs; e.ptab = l0_ptab; e.ptab_gstride = l0_gstride; e.ptype = h->d_ptype; e.eorig = h->d_eorig;
            e.l0_gid = h->d_gid; e.l0c = h->d_l0c;
            rgp = rg;
            if (e.nreg > 0 && pruned) {
                // (a shared launch: most of the representatives' groups return at once, what runs scales with the batch
                // like the dynamic regions do -- the general threshold applies; measured at 4-5 pockets x 30 copies:
                // 1.21 M sample-steps/s end to end at 4 rows per wave, 1.26 M at 8)
                // (one pocket x 128 copies: few shared rows, but 128 graphs' worth of ff / pf / fp items -- 8 rows per wave is 8 % ahead)
                rg = shared ? ((h->share_rows >= h->pol.rg2_rows_min || (long)e.ntiles * 32 >= h->pol.rg2_rows_min_hoist) ? 2 : 1)
                            : ((long)e.ntiles * 32 >= h->pol.rg2_rows_min_hoist ? 2 : 1);
                const int rgp_pol = shared ? rg : ((long)e.ntiles * 32 >= h->pol.rg2p_rows_min ? 2 : 1);
                if (h->pol.l0_rga) rg = h->pol.l0_rga;
                rgp = h->pol.l0_rgp ? h->pol.l0_rgp : std::max(rg, rgp_pol);
                if (rg == 2) rgp = 2;
                for (int r = 0; r < e.nreg; ++r) e.ngroups_sel += region_groups(r, 4 * (r >= 3 * h->B ? rgp : rg));
            }
            h->last_hoist = 4 * rgp;
        }
        // n16 form (pf_n16.hip): 16-row items on four waves.  Conv layers >= 1 read h / v of the sources from memory; conv
        // layer 0 needs the static hoist's type tables (protein sources) and encodes the centers on the fly
        const bool n16e = !train && rg && n16_batch && (l > 0 ? (h->pol.n16_mask & 1) != 0 : n16_l0);
        if (n16e) {
            for (int et = 0; et < 4; ++et) {
                e.n16[et] = h->d_w + (l > 0 ? h->n16_msg[(size_t)l * 4 + et] : h->n16_l0[et]);
                e.n16_stride[et] = (int)(l > 0 ? h->n16_msg_stride : h->n16_l0_stride[et]);
                e.ptab16_off[et] = et == ET_PP ? c.rec_nf * PF_S : (et == ET_PF ? 2 * c.rec_nf * PF_S : -1);
            }
            if (l == 0) { e.zs = nullptr; rgp = 4; h->last_hoist = 16; }      // (ptab / ptype / l0_gid were set above)
            if (l == 0 && step != nullptr && h->sampling && !h->coords_custom) {
                // a sampling run: pp geometry from the original coordinates (the same bits in every step, with or without the rows
                // computed ahead; no race with the build that shifts xn under the speculative items)
                e.x0_static = h->d_prot_x0;
                // "pa" regions whose rows were computed ahead (k_n16_pa_spec) and still apply are skipped
                if (spec_have && h->pa_spec && !shared && h->B <= 64 && !e.need) { e.pa_skip = h->d_pa_same; h->last_spec = 1; }
            }
            if (l == 0 && cen_have && h->n16_l0h[ET_FF] != 0) {              // ff / fp items start from the center hoist's tables (kind M0H)
                for (int et : {(int)ET_FF, (int)ET_FP}) { e.n16[et] = h->d_w + h->n16_l0h[et]; e.n16_stride[et] = (int)h->n16_l0h_stride[et]; }
                e.pcen = h->d_cen_p; e.pcen_nf = h->Nf;
                h->last_cen = true;
            }
            e.ngroups_sel = 0;
            for (int r = 0; r < e.nreg; ++r) e.ngroups_sel += region_groups(r, 16);
            // conv layer 0, every graph with ff / pf / fp regions of one capacity: their items are mapped by arithmetic (k_n16_edge_u)
            if (l == 0 && h->pol.fused_uni && !shared && e.reg == h->d_reg && e.nreg == 4 * h->B && h->B <= 64 && !e.pa_abs && !e.need) {
                bool uni = true;
                int stride[3] = {0, 0, 0}, grp[3] = {0, 0, 0};
                for (int et = 0; et < 3 && uni; ++et) {
                    const size_t o = (size_t)et * h->B;
                    if (h->B > 1) stride[et] = h->h_reg[o + 1] - h->h_reg[o];
                    for (int g = 0; g < h->B && uni; ++g)
                        uni = h->h_cap[o + g] == h->h_cap[o] && h->h_reg[o + g] == h->h_reg[o] + g * stride[et];
                    grp[et] = (h->h_cap[o] + 15) / 16;
                    uni = uni && stride[et] >= 0 && stride[et] < 65536 && grp[et] > 0 && grp[et] < 8;
                }
                if (uni) {
                    for (int et = 0; et < 3; ++et) e.uni_base[et] = h->h_reg[(size_t)et * h->B];
                    e.uni_s01 = stride[0] | (stride[1] << 16);
                    e.uni_s2g = stride[2] | (grp[0] << 16) | (grp[1] << 19) | (grp[2] << 22) | ((h->B - 1) << 25);
                    e.uni_pa_groups = 0;
                    for (int g = 0; g < h->B; ++g) e.uni_pa_groups += region_groups(3 * h->B + g, 16);
                }
            }
            rg = 4;                                  // 16 slots per partial-row group
            if (l == 0 && e.x0_static && h->pa_spec && !shared && h->B <= 64 && !e.need) {      // what k_n16_pa_spec needs of this launch (the regions, streams and tables of conv layer 0)
                h->e0 = e; h->ep0 = ep; h->e0_saved = true;
                h->e0_groups = 0;
                for (int g = 0; g < h->B; ++g) h->e0_groups += region_groups(3 * h->B + g, 16);
            }
        }
        h->last_family.resize(c.n_convs);
        h->last_family[l] = rg ? 4 * rg : ((!train && e.ntiles <= ((last || pruned) ? std::max(h->pol.coop_edge_max, h->pol.coop2_edge_max) : std::max(h->pol.coop_edge_max, h->pol.coop2_dense_max))) ? 128 : 32);
        for (int et = 0; et < 4; ++et) e.rgs[et] = h->d_w + h->rgs_msg[(size_t)l * 4 + et];
        e.rgs_stride = (int)h->rgs_msg_stride;
        const int esplit = (rg == 1 && e.ntiles * 8 <= h->pol.rg_split_max && !e.zs) ? 1 : 0;    // fewer groups than SIMDs: latency-bound
        if (n16e && fuse_l0node && l == 1) {
            fz.chain[ET_FF] = h->d_w + h->n16_fused[0]; fz.chain_stride[ET_FF] = (int)h->n16_fused_stride[0];
            fz.chain[ET_PF] = h->d_w + h->n16_fused[1]; fz.chain_stride[ET_PF] = (int)h->n16_fused_stride[1];
            fz.upd_pharm = h->d_w + h->n16_upd[(size_t)0 * 2 + 1]; fz.upd_pharm_stride = (int)h->n16_upd_stride;
            fz.htab = l0_ptab + (size_t)3 * c.rec_nf * PF_S; fz.htab_gstride = l0_gstride; fz.ptype = h->d_ptype;
            fz.hcen = h->last_cen ? h->d_cen_h : nullptr;
            fz.h_out = h->d_h[cur]; fz.v_out = h->d_v[cur];          // (cur was flipped behind conv layer 0: its output side)
            fz.pharm_ptr = h->d_pharm_ptr; fz.Np = h->Np; fz.n_edge_items = e.ngroups_sel;
            for (int r = 0; r < e.nreg; ++r) (r < h->B ? fz.nff_cap : fz.npf_cap) += region_groups(r, 16);
            fz.xcd_split = ((h->pol.xcd_split & 1) && 2 * h->B <= 64 && h->max_nf <= 16) ? 1 : 0;
            // every graph with regions of one capacity (the same number of centers everywhere): the regions of an etype sit at a fixed
            // stride and the item map is arithmetic (k_n16_fused_u); PFDYN_FUSED_UNI=0: the work-list form
            if (fz.xcd_split && h->pol.fused_uni && e.reg == h->d_reg && e.nreg == 2 * h->B) {
                bool uni = true;
                int stride[2] = {0, 0};
                for (int et = 0; et < 2 && uni; ++et) {
                    const size_t o = (size_t)et * h->B;
                    if (h->B > 1) stride[et] = h->h_reg[o + 1] - h->h_reg[o];
                    for (int g = 0; g < h->B && uni; ++g)
                        uni = h->h_cap[o + g] == h->h_cap[o] && h->h_reg[o + g] == h->h_reg[o] + g * stride[et];
                    uni = uni && stride[et] >= 0 && stride[et] < 65536 && (h->h_cap[o] + 15) / 16 < 256 && h->h_cap[o] > 0;
                }
                if (uni) {
                    fz.uni_ff_base = h->h_reg[0]; fz.uni_pf_base = h->h_reg[(size_t)h->B];
                    fz.uni_strides = stride[0] | (stride[1] << 16);
                    fz.uni_groups = ((h->h_cap[0] + 15) / 16) | (((h->h_cap[(size_t)h->B] + 15) / 16) << 8);
                }
            }
            h->last_family[l] = 17;                      // pf_debug_kernel_family: 16-row items with conv layer 0's node update in front
            { ProfScope ps(h, pf_handle::K_EDGE_LAST, s); pfk_n16_fused(&e, &fz, &ep, s); }
        }
        else if (n16e) { ProfScope ps(h, (last && c.n_convs > 1) ? pf_handle::K_EDGE_LAST : pf_handle::K_EDGE_COOP, s); pfk_n16_edge(&e, &ep, l == 0, s); }
        else if (rg) { ProfScope ps(h, (last && c.n_convs > 1) ? pf_handle::K_EDGE_LAST : pf_handle::K_EDGE_COOP, s); pfk_rg_edge(&e, enc_fly ? &ep : nullptr, l == 0, rg, esplit, rgp, s); }
        else if (e.ntiles <= h->pol.coop_edge_max && !train) { ProfScope ps(h, (last && c.n_convs > 1) ? pf_handle::K_EDGE_LAST : pf_handle::K_EDGE_COOP, s); pfk_edge_msg_coop(&e, l == 0, s); }
        else if (e.ntiles <= ((last || pruned) ? h->pol.coop2_edge_max : h->pol.coop2_dense_max) && !train) { ProfScope ps(h, (last && c.n_convs > 1) ? pf_handle::K_EDGE_LAST : pf_handle::K_EDGE_COOP, s); pfk_edge_msg_coop2(&e, l == 0, s); }
        else { ProfScope ps(h, pf_handle::K_EDGE, s); pfk_edge_msg(&e, l == 0, s); }

        NodeParams n{};
        n.tiles = pruned ? h->d_node_tiles_act : h->d_node_tiles;
        n.ntiles = last ? h->n_node_tiles_last : (pruned ? h->n_node_tiles_act : h->n_node_tiles);
        n.in_start = h->d_in_start; n.in_cnt = h->d_in_cnt; n.N = h->N;
        n.pp_slot = pruned ? (shared ? 3 : 2) : 1; n.row_ids = h->d_act_ids; n.dyn_cnt = h->d_dyn_cnt;
        n.msg_s = e.msg_s; n.msg_v = e.msg_v; n.zero_row = h->zero_row;
        n.h_in = e.h; n.v_in = e.v;
        n.h_out = train ? h->t_H[l + 1] : h->d_h[cur ^ 1]; n.v_out = train ? h->t_V[l + 1] : h->d_v[cur ^ 1];
        if (train) { n.drop_thr = h->t_common.drop_thr; n.drop_scale = h->t_common.drop_scale; n.seed = h->t_common.seed; n.layer = l; n.mask_override = h->t_common.mask_override; }
        n.gid = h->d_gid; n.gnorm = h->d_gnorm; n.B = h->B;
        n.norm_mode = c.message_norm_mode; n.norm_value = c.message_norm_value;
        for (int nt = 0; nt < 2; ++nt) {
            const size_t* lo = &h->ln_off[(size_t)(l * 2 + nt) * 4];
            n.w[nt].ln1_w = h->d_w + lo[0]; n.w[nt].ln1_b = h->d_w + lo[1];
            n.w[nt].ln2_w = h->d_w + lo[2]; n.w[nt].ln2_b = h->d_w + lo[3];
            n.w[nt].upd = h->d_gvp + h->upd_base(l, nt);
        }
        n.n_upd = c.n_update_gvps;
        n.grp = rg ? 4 * rg : 32;
        n.grp_pa = rgp ? 4 * rgp : n.grp;
        if (train) {
            h->t_grp.resize(c.n_convs); h->t_grp[l] = n.grp;
            h->t_node_saved.resize(c.n_convs); h->t_node_saved[l] = 0;
        }
        for (int nt = 0; nt < 2; ++nt) n.rg_upd[nt] = h->d_w + h->rg_upd[(size_t)l * 2 + nt];
        for (int nt = 0; nt < 2; ++nt) {
            n.rgs_upd[nt] = h->d_w + h->rgs_upd[(size_t)l * 2 + nt];
            n.rgs_stride[nt] = (int)h->rgs_upd_stride[(size_t)l * 2 + nt];
        }
        if (fuse_l0node && l == 0) {       // no node launch: the last layer's edge items (and its store items) compute these rows
            fz.in_start = n.in_start; fz.in_cnt = n.in_cnt; fz.N = n.N; fz.pp_slot = n.pp_slot;
            // (records describe the sources' in-edges as the update + build left them: the "pa" region as an atom's second segment)
            fz.rec = (h->rec_valid && h->d_rec && n.pp_slot == 2 && !shared) ? h->d_rec : nullptr;
            fz.msg_s = n.msg_s; fz.msg_v = n.msg_v; fz.zero_row = n.zero_row; fz.grp = n.grp; fz.grp_pa = n.grp_pa;
            fz.gid = n.gid; fz.gnorm = n.gnorm; fz.B = n.B; fz.norm_mode = n.norm_mode; fz.norm_value = n.norm_value;
            for (int nt = 0; nt < 2; ++nt) {
                fz.ln1_w[nt] = n.w[nt].ln1_w; fz.ln1_b[nt] = n.w[nt].ln1_b; fz.ln2_w[nt] = n.w[nt].ln2_w; fz.ln2_b[nt] = n.w[nt].ln2_b;
            }
            fz.n_upd = n.n_upd;
            cur ^= 1;
            continue;
        }
        if (rg) {
            const int rgn = (long)n.ntiles * 32 >= h->pol.rg2_rows_min_node ? 2 : 1;
            const bool fuse = last && !train && h->fuse_head && h->n_head_tiles == n.ntiles;
            const int nsplit = (!train && rgn == 1 && n.ntiles * 8 <= (fuse ? h->pol.rg_split_max_head : h->pol.rg_split_max_node)) ? 1 : 0;
            // the tail launch: one workgroup per graph does the centers' node update, the head, the sampler update and the
            // edge build (the fast build's shape: kNN pf edges, pockets of at most 512 atoms)
            const bool tail = fuse && step != nullptr && l > 0 && (h->pol.n16_mask & 8) && h->B <= h->pol.tail_graphs_max &&
                              enc_fly && c.pf_k > 0 && h->max_np <= 512 && c.pharm_nf <= 16 && h->step_build_fast;
            const bool tail16 = tail && h->pol.tail_form == 16 && h->n16_tail != 0;
            if (tail && !tail16) {                      // row-group form: the fused node + head item, two per workgroup
                HeadParams hp{};
                hp.tiles = h->d_head_tiles; hp.ntiles = h->n_head_tiles; hp.node_base = h->Np;
                hp.gvps = h->d_gvp + h->head_base(); hp.n_gvps = c.n_noise_gvps;
                hp.a_out = h->d_w + h->out_a; hp.b_out = h->d_w + h->out_b; hp.pharm_nf = c.pharm_nf;
                hp.eps_h = eps_h; hp.eps_x = eps_x;
                const bool share_next = (h->prune && c.n_convs == 2) && share_now(h);     // what the next denoising step's dynamics call will ask for
                const BuildParams bpn = build_params(h, share_next);
                { ProfScope ps(h, pf_handle::K_HEAD, s); pfk_rg_tail(&n, &hp, step, &bpn, s); }
                build_done(h, share_next);
                h->tail_done = true; h->last_tail = 4;
                head_done = true;
            } else if (tail16) {
                TailParams tp{};
                tp.in_start = n.in_start; tp.in_cnt = n.in_cnt; tp.N = n.N;
                tp.msg_s = n.msg_s; tp.msg_v = n.msg_v; tp.zero_row = n.zero_row; tp.grp = n.grp;
                tp.h_in = n.h_in; tp.v_in = n.v_in;
                tp.gid = n.gid; tp.gnorm = n.gnorm; tp.B = n.B; tp.norm_mode = n.norm_mode; tp.norm_value = n.norm_value;
                tp.ln1_w = n.w[1].ln1_w; tp.ln1_b = n.w[1].ln1_b; tp.ln2_w = n.w[1].ln2_w; tp.ln2_b = n.w[1].ln2_b;
                tp.n_upd = n.n_upd; tp.n_head = c.n_noise_gvps;
                tp.chain = h->d_w + h->n16_tail; tp.chain_stride = (int)h->n16_tail_stride;
                tp.pharm_nf = c.pharm_nf; tp.eps_h = eps_h; tp.eps_x = eps_x;
                const bool share_next = (h->prune && c.n_convs == 2) && share_now(h);     // what the next denoising step's dynamics call will ask for
                const BuildParams bpn = build_params(h, share_next);
                { ProfScope ps(h, pf_handle::K_HEAD, s); pfk_n16_tail(&tp, step, &bpn, s); }
                build_done(h, share_next);
                h->tail_done = true; h->last_tail = 16;
                head_done = true;
            } else if (fuse) {
                // few two-wave items: confined to node_xcds XCDs when they fit one per compute unit there (32 CUs per XCD)
                if (nsplit && h->pol.node_xcds > 0 && n.ntiles * 8 <= 32 * h->pol.node_xcds) n.xcd_n = h->pol.node_xcds;
                if (nsplit && l > 0 && h->pol.node_static) { n.st_n0 = h->Np; n.st_n = h->Nf; }      // (the last layer's node tiles ARE the static tiling of the centers)
                HeadParams hp{};
                hp.tiles = h->d_head_tiles; hp.ntiles = h->n_head_tiles; hp.node_base = h->Np;
                hp.gvps = h->d_gvp + h->head_base(); hp.n_gvps = c.n_noise_gvps;
                hp.a_out = h->d_w + h->out_a; hp.b_out = h->d_w + h->out_b; hp.pharm_nf = c.pharm_nf;
                hp.eps_h = eps_h; hp.eps_x = eps_x;
                // a denoising step of a small batch: the step's update + build joins this launch as workgroups of its own (the fast
                // build's shape: kNN pf edges, pockets of at most 512 atoms); timing the two separately needs the separate launches
                const bool hsb = n.st_n > 0 && step != nullptr && h->pol.hs_build && enc_fly && c.pf_k > 0 && h->max_np <= 512 && c.pharm_nf <= 16 &&
                                 h->step_build_fast && h->B <= 256 && !(h->prof_mask & (1u << pf_handle::K_STEP)) && h->d_xchg && h->d_xstat;
                if (hsb) {
                    hp.xchg = h->d_xchg; hp.xchg_fault = h->xchg_fault;
                    // center hoist for the NEXT call: its timestep from the announced plan (the entry behind this step's t); the tables
                    // serve a call that runs conv layer 0 on the n16 kernels with the type tables (as this one did); this step's
                    // features before the update must be in a snapshot (pf_sample_begin / the previous step left it)
                    CenHoistParams cp{};
                    float t_next = NAN;
                    if (h->cen_hoist && t_scalar && !h->t_plan.empty() && h->last_hoist == 16 && h->l0c_off != 0 && h->n16_l0h[ET_FF] != 0 &&
                        h->snap_cur >= 0 && step->h_snap_out != nullptr && h->d_xchg2 && fuse_l0node) {
                        const size_t n = h->t_plan.size();
                        for (size_t k = 0; k < n; ++k) {
                            const size_t i = (h->plan_pos + k) % n;
                            if (h->t_plan[i] == *t_scalar) { h->plan_pos = i; if (i + 1 < n) t_next = h->t_plan[i + 1]; break; }
                        }
                    }
                    // ---- the NEXT call's "pa" messages, ahead of time, as workgroups of this launch (BuildParams::pa_same): conv layer 0's rows
                    // of this call have been consumed by the fused launch, the next timestep's type tables exist (or are made now)
                    EdgeParams es{};
                    EncodeParams ees{};
                    int spec_groups = 0;
                    if (h->e0_saved && h->pa_spec && !h->t_plan.empty() && t_scalar && h->d_pa_stamp && !(h->prune && c.n_convs == 2 && share_now(h))) {
                        float tn = NAN;
                        const size_t n = h->t_plan.size();
                        for (size_t k = 0; k < n; ++k) {
                            const size_t i = (h->plan_pos + k) % n;
                            if (h->t_plan[i] == *t_scalar) { if (i + 1 < n) tn = h->t_plan[i + 1]; break; }
                        }
                        if (tn == tn) {
                            l0_prepare_t(h, &tn, 1, s);                  // (a no-op when the plan was announced)
                            uint32_t bits; memcpy(&bits, &tn, 4);
                            es = h->e0;
                            es.ptab = h->d_ptab + (size_t)h->ptab_slot[bits] * L0_NTAB * c.rec_nf * PF_S;
                            es.pa_skip = nullptr; es.pcen = nullptr;
                            for (int et = 0; et < 4; ++et) { es.n16[et] = h->d_w + h->n16_l0[et]; es.n16_stride[et] = (int)h->n16_l0_stride[et]; }
                            ees = h->ep0; ees.t_scalar = tn;
                            spec_groups = h->e0_groups;
                            h->spec_valid = true; h->spec_t = tn; h->spec_wver = h->w_version;
                        }
                    }
                    if (t_next == t_next) {
                        cp.on = 1; cp.Nf = h->Nf; cp.nf = c.pharm_nf; cp.t_next = t_next;
                        cp.pharm_h = h->d_snap[h->snap_cur]; cp.noise = step->noise;
                        cp.a_ts = step->a_ts; cp.var = step->var; cp.sigma = step->sigma; cp.ep_zt = step->ep_zt; cp.ep_pred = step->ep_pred;
                        cp.ep_feat = step->ep_feat;
                        cp.enc_w = h->d_w + h->enc_w[1]; cp.enc_b = h->d_w + h->enc_b[1];
                        cp.enc_lw = h->d_w + h->enc_lw[1]; cp.enc_lb = h->d_w + h->enc_lb[1];
                        cp.blk = h->d_w + h->l0c_off; cp.cen_h = h->d_cen_h; cp.cen_p = h->d_cen_p; cp.xchg2 = h->d_xchg2;
                        hp.xchg2 = h->d_xchg2;
                        h->cen_valid = true; h->cen_t = t_next; h->cen_wver = h->w_version;
                    }
                    const bool share_next = (h->prune && c.n_convs == 2) && share_now(h);     // what the next denoising step's dynamics call will ask for
                    BuildParams bpn = build_params(h, share_next);
                    // edge records for the next call's fused launch: radius ff edges, compact "pa" regions, the static hoist's element types
                    const bool with_rec = h->d_rec && !share_next && c.ff_k == 0 && bpn.act_ids && !bpn.pa_static && h->last_hoist == 16 && fuse_l0node;
                    if (with_rec) { bpn.rec = h->d_rec; bpn.ptype = h->d_ptype; }
                    { ProfScope ps(h, pf_handle::K_HEAD, s); pfk_rg_node_hs_build(&n, &hp, step, &bpn, h->d_xstat, h->pol.xchg_sleep, h->pol.hsb_avoid, h->xchg_poll_max, &cp, &es, &ees, spec_groups, s); }
                    build_done(h, share_next, with_rec);
                    h->tail_done = true; h->last_tail = 2;
                } else { ProfScope ps(h, pf_handle::K_HEAD, s); pfk_rg_node(&n, &hp, enc_fly ? &ep : nullptr, l == 0, rgn, nsplit, s); }
                head_done = true;
            } else {
                if (train && h->train_node_save) {
                    n.sv_z = h->t_nsv_z[l]; n.sv_g = h->t_nsv_g[l]; n.sv_v = h->t_nsv_v[l]; n.sv_stride = (size_t)2 * h->N;
                    h->t_node_saved[l] = 1;
                }
                ProfScope ps(h, pf_handle::K_NODE_COOP, s); pfk_rg_node(&n, nullptr, enc_fly ? &ep : nullptr, l == 0, rgn, nsplit, s);
            }
        }
        else if (last && !train && h->fuse_head && n.ntiles <= h->pol.coop_node_max && h->n_head_tiles == n.ntiles) {
            // last layer (pharm tiles only) + noise head in one launch: the layer output stays in registers
            HeadParams hp{};
            hp.tiles = h->d_head_tiles; hp.ntiles = h->n_head_tiles; hp.node_base = h->Np;
            hp.gvps = h->d_gvp + h->head_base(); hp.n_gvps = c.n_noise_gvps;
            hp.a_out = h->d_w + h->out_a; hp.b_out = h->d_w + h->out_b; hp.pharm_nf = c.pharm_nf;
            hp.eps_h = eps_h; hp.eps_x = eps_x;
            { ProfScope ps(h, pf_handle::K_HEAD, s); pfk_node_head_coop(&n, &hp, l == 0, s); }
            head_done = true;
        }
        else if (train && h->train_rg_node && h->pol.rg_mode(n.ntiles) > 0) {
            // training forward: the row-group node kernel (with the two GVPDropout sites) on the tile edge kernels' partial rows
            // (one per 32-slot tile and destination: grp = 32); the layer input comes from memory, as the backward kernels read it
            n.grp = 32; n.grp_pa = 32;
            if (h->train_node_save) {
                n.sv_z = h->t_nsv_z[l]; n.sv_g = h->t_nsv_g[l]; n.sv_v = h->t_nsv_v[l]; n.sv_stride = (size_t)2 * h->N;
                h->t_node_saved[l] = 1;
            }
            ProfScope ps(h, pf_handle::K_NODE_COOP, s);
            pfk_rg_node(&n, nullptr, nullptr, l == 0, h->pol.rg_mode(n.ntiles), 0, s);
        }
        else if (n.ntiles <= h->pol.coop_node_max && !train) { ProfScope ps(h, pf_handle::K_NODE_COOP, s); pfk_node_update_coop(&n, l == 0, s); }
        else { ProfScope ps(h, pf_handle::K_NODE, s); pfk_node_update(&n, l == 0, s); }
        cur ^= 1;
    }
    HeadParams hp{};
    hp.tiles = h->d_head_tiles; hp.ntiles = h->n_head_tiles; hp.node_base = h->Np;
    hp.h = train ? h->t_H[c.n_convs] : h->d_h[cur]; hp.v = train ? h->t_V[c.n_convs] : h->d_v[cur];
    hp.gvps = h->d_gvp + h->head_base(); hp.n_gvps = c.n_noise_gvps;
    hp.a_out = h->d_w + h->out_a; hp.b_out = h->d_w + h->out_b; hp.pharm_nf = c.pharm_nf;
    hp.eps_h = eps_h; hp.eps_x = eps_x;
    if (train) h->t_head_saved = false;
    if (!head_done && train && h->train_rg_head && !h->rg_msg.empty() && h->Nf > 0) {
        // training forward: the head chain on the row-group code (4 rows per wave), which also leaves every level's pre-activations,
        // gate pre-activations and gated vectors for k_bwd_head (pharm rows are the contiguous rows [Np, Np + Nf))
        UnitParams up{};
        up.s_in = hp.h + (size_t)h->Np * PF_S; up.v_in = hp.v + (size_t)h->Np * 48; up.s_out = eps_h; up.v_out = eps_x;
        up.n = h->Nf; up.kind = 3; up.n_gvps = c.n_noise_gvps; up.pharm_nf = c.pharm_nf;
        up.stream = h->d_w + h->rg_upd[(size_t)(c.n_convs - 1) * 2 + 1]; up.skip_gvps = c.n_update_gvps;
        up.sv_z = h->t_hsv_z; up.sv_g = h->t_hsv_g; up.sv_v = h->t_hsv_v; up.sv_stride = (size_t)h->Nf;
        { ProfScope ps(h, pf_handle::K_HEAD, s); pfk_rg_unit(&up, s); }
        h->t_head_saved = true;
        head_done = true;
    }
    if (!head_done) { ProfScope ps(h, pf_handle::K_HEAD, s); if (hp.ntiles <= h->pol.coop_node_max) pfk_noise_head_coop(&hp, s); else pfk_noise_head(&hp, s); }
    h->edges_built = h->tail_done;          // whoever moves the coordinates next decides (pf_denoise_step rebuilds; the tail launch has)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PF_FAIL(h, PF_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return PF_OK;
}

// collects regions to clear and clears them with one launch (flush), eight at a time
struct ZeroBatch {
    ZeroList z{};
    hipStream_t s;
    explicit ZeroBatch(hipStream_t st) : s(st) {}
    void add(void* p, size_t nbytes) {
        if (!p || nbytes == 0) return;
        if (z.cnt == 8) flush();
        z.p[z.cnt] = p; z.nbytes[z.cnt] = nbytes; ++z.cnt;
    }
    void flush() { pfk_zero_multi(&z, s); z.cnt = 0; }
};

static int check_ready(pf_handle* h, bool need_batch) {
    if (!h) return PF_ERR_ARG;
    if (!h->committed) PF_FAIL(h, PF_ERR_STATE, "weights not committed (pf_commit_weights)");
    if (need_batch && !h->have_batch) PF_FAIL(h, PF_ERR_STATE, "no pocket batch set (pf_set_pocket_batch)");
    return PF_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* pf_version(void) { return "libpfdyn 0.1 (gfx950, fp32 MFMA)"; }

const char* pf_last_error(const pf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pf_create(const pf_config* cfg, pf_handle** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return PF_ERR_ARG; }
    *out = nullptr;
    auto bad = [&](const char* m) { g_create_error = m; return PF_ERR_ARG; };
    if (cfg->abi_version != PF_ABI_VERSION) return bad("abi_version mismatch");
    if (cfg->vector_size != PF_V) return bad("vector_size must be 16 (kernels are specialised)");
    if (cfg->n_hidden_scalars != PF_S) return bad("n_hidden_scalars must be 128 (kernels are specialised)");
    if (cfg->rbf_dim != PF_R) return bad("rbf_dim must be 16");
    if (cfg->pharm_nf < 1 || cfg->pharm_nf > 16 || cfg->rec_nf < 1) return bad("pharm_nf must be in 1..16, rec_nf >= 1");
    if (cfg->n_convs < 1 || cfg->n_message_gvps < 1 || cfg->n_message_gvps > PF_MAX_GVPS || cfg->n_update_gvps < 1 ||
        cfg->n_update_gvps > PF_MAX_GVPS || cfg->n_noise_gvps < 1 || cfg->n_noise_gvps > PF_MAX_GVPS)
        return bad("layer counts out of range");
    if (cfg->ff_k < 0 || cfg->ff_k > PF_MAXK || cfg->pf_k < 0 || cfg->pf_k > PF_MAXK) return bad("ff_k / pf_k must be in 0..16");
    if (cfg->message_norm_mode < 0 || cfg->message_norm_mode > 2) return bad("bad message_norm_mode");
    if (cfg->message_norm_mode == PF_NORM_VALUE && !(cfg->message_norm_value > 0)) return bad("message_norm_value must be > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_create_error = "no HIP device available (libpfdyn has no CPU fallback)";
        return PF_ERR_HIP;
    }
    pf_handle* h = new pf_handle();
    h->cfg = *cfg;
    h->init_tuning();
    *out = h;
    return PF_OK;
}

void pf_destroy(pf_handle* h) {
    if (!h) return;
    free_ws(h);
    if (h->d_w) (void)hipFree(h->d_w);
    if (h->d_gvp) (void)hipFree(h->d_gvp);
    if (h->d_flat) (void)hipFree(h->d_flat);
    if (h->d_wpack) (void)hipFree(h->d_wpack);
    if (h->d_xstat) (void)hipFree(h->d_xstat);
    if (h->xstat_host) (void)hipHostFree(h->xstat_host);
    if (h->d_tseg) (void)hipFree(h->d_tseg);
    if (h->d_gvpt) (void)hipFree(h->d_gvpt);
    if (h->d_map) (void)hipFree(h->d_map);
    if (h->d_split_tab) (void)hipFree(h->d_split_tab);
    if (h->d_l0c) (void)hipFree(h->d_l0c);
    if (h->d_ptab) (void)hipFree(h->d_ptab);
    for (int k = 0; k < 2; ++k) { if (h->stage[k]) (void)hipHostFree(h->stage[k]); if (h->stage_ev[k]) (void)hipEventDestroy(h->stage_ev[k]); }
    for (int k = 0; k < 2; ++k) {
        if (h->d_tab[k]) (void)hipFree(h->d_tab[k]);
        if (h->tab_guard[k]) (void)hipEventDestroy(h->tab_guard[k]);
        if (h->tab_up[k]) (void)hipEventDestroy(h->tab_up[k]);
    }
    for (int k = 0; k < 3; ++k) if (h->cmp_ev[k]) (void)hipEventDestroy(h->cmp_ev[k]);
    if (h->s_copy) (void)hipStreamDestroy(h->s_copy);
    if (h->s_side) (void)hipStreamDestroy(h->s_side);
    if (h->l0flag_host) (void)hipHostFree(h->l0flag_host);
    if (h->l0flag_ev) (void)hipEventDestroy(h->l0flag_ev);
    for (int k = 0; k < pf_handle::K_NUM; ++k)
        for (auto& ev : h->prof_ev[k]) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    delete h;
}

int pf_set_weight(pf_handle* h, const char* name, const float* host_data, int32_t ndim, const int64_t* shape) {
    if (!h || !name) return PF_ERR_ARG;
    int64_t n = 1;
    RawTensor t;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    if (n == 0 || std::string(name) == "gamma.gamma") return PF_OK;     // dropout dummy_param / schedule table
    if (!host_data) PF_FAIL(h, PF_ERR_ARG, "null data for %s", name);
    t.data.assign(host_data, host_data + n);
    h->raw[name] = std::move(t);
    h->committed = false;
    return PF_OK;
}

int pf_commit_weights(pf_handle* h) {
    if (!h) return PF_ERR_ARG;
    const pf_config& c = h->cfg;
    const auto exp = expected_tensors(c);
    for (const auto& kv : exp) {
        auto it = h->raw.find(kv.first);
        if (it == h->raw.end()) PF_FAIL(h, PF_ERR_WEIGHT, "missing weight tensor %s", kv.first.c_str());
        if (it->second.shape != kv.second) PF_FAIL(h, PF_ERR_WEIGHT, "wrong shape for %s", kv.first.c_str());
    }
    if (h->raw.size() != exp.size()) {
        for (const auto& kv : h->raw) {
            bool found = false;
            for (const auto& e : exp) if (e.first == kv.first) { found = true; break; }
            if (!found) PF_FAIL(h, PF_ERR_WEIGHT, "unexpected weight tensor %s", kv.first.c_str());
        }
    }
    h->h_gvp.clear();
    std::vector<GvpOff> offs;
    // the packing is pure data movement (copies and zero padding), so running it a second time on tensors whose VALUES
    // are their own flat index + 1 yields, per packed element, where it comes from: the gather map that lets
    // pf_set_flat_params refresh the packed weights on the device after an optimiser step
    auto pack_all = [&]() {
        h->h_w.clear();
        offs.clear();
        for (int l = 0; l < c.n_convs; ++l)
            for (int et = 0; et < 4; ++et)
                for (int j = 0; j < c.n_message_gvps; ++j) offs.push_back(pack_gvp(h, msg_spec(c, l, et, j)));
        h->n_msg_tot = (int)offs.size();
        for (int l = 0; l < c.n_convs; ++l)
            for (int nt = 0; nt < 2; ++nt)
                for (int j = 0; j < c.n_update_gvps; ++j) offs.push_back(pack_gvp(h, upd_spec(c, l, nt, j)));
        h->n_upd_tot = (int)offs.size() - h->n_msg_tot;
        for (int k = 0; k < c.n_noise_gvps; ++k) offs.push_back(pack_gvp(h, head_spec(c, k)));
        for (int nt = 0; nt < 2; ++nt) {
            const std::string p = std::string("dynamics.") + kNtKey[nt] + "_encoder.";
            {   // encoder weight transposed to [nf+1][128]: coalesced loads of one input's column
                const RawTensor& W = h->raw[p + "0.weight"];
                const int K = (int)W.shape[1], S = (int)W.shape[0];
                std::vector<float> wt((size_t)K * S);
                for (int f = 0; f < S; ++f) for (int k = 0; k < K; ++k) wt[(size_t)k * S + f] = W.data[(size_t)f * K + k];
                h->enc_w[nt] = push(h->h_w, wt);
            }
            h->enc_b[nt] = push(h->h_w, h->raw[p + "0.bias"].data);
            h->enc_lw[nt] = push(h->h_w, h->raw[p + "2.weight"].data);
            h->enc_lb[nt] = push(h->h_w, h->raw[p + "2.bias"].data);
        }
        {   // protein encoder Linear [128][rec_nf+1] as A fragments [tile][k-step][lane]; k-step t, half hl <-> input 2t+hl
            const RawTensor& W = h->raw["dynamics.prot_encoder.0.weight"];
            const RawTensor& Bv = h->raw["dynamics.prot_encoder.0.bias"];
            const int K = c.rec_nf + 1, nke = (K + 1) / 2;
            std::vector<float> a((size_t)4 * nke * 64, 0.f), bf(128);
            for (int mo = 0; mo < 4; ++mo)
                for (int t = 0; t < nke; ++t)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 31, hl = lane >> 5, k = 2 * t + hl;
                        a[((size_t)mo * nke + t) * 64 + lane] = k < K ? W.data[(size_t)(32 * mo + i) * K + k] : 0.f;
                    }
            for (int hl = 0; hl < 2; ++hl)
                for (int mo = 0; mo < 4; ++mo)
                    for (int r = 0; r < 16; ++r) bf[(size_t)hl * 64 + mo * 16 + r] = Bv.data[32 * mo + rho(r, hl)];
            h->enc_a = push(h->h_w, a);
            h->enc_bf = push(h->h_w, bf);
        }
        h->ln_off.assign((size_t)c.n_convs * 2 * 4, 0);
        for (int l = 0; l < c.n_convs; ++l)
            for (int nt = 0; nt < 2; ++nt) {
                size_t* lo = &h->ln_off[(size_t)(l * 2 + nt) * 4];
                const std::string p1 = conv_prefix(l) + "message_layer_norms." + kNtKey[nt] + ".feat_norm.";
                const std::string p2 = conv_prefix(l) + "update_layer_norms." + kNtKey[nt] + ".feat_norm.";
                lo[0] = push(h->h_w, h->raw[p1 + "weight"].data);
                lo[1] = push(h->h_w, h->raw[p1 + "bias"].data);
                lo[2] = push(h->h_w, h->raw[p2 + "weight"].data);
                lo[3] = push(h->h_w, h->raw[p2 + "bias"].data);
            }
        {   // to_scalar_output as A fragments: K = 64 (32 k-steps), rows = outputs
            const RawTensor& W = h->raw["dynamics.noise_predictor.noise_predictor.to_scalar_output.weight"];
            std::vector<float> a((size_t)32 * 64, 0.f);
            for (int ks = 0; ks < 32; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, hl = lane >> 5;
                    const int k = 32 * (ks / 16) + rho(ks % 16, hl);
                    a[(size_t)ks * 64 + lane] = i < c.pharm_nf ? W.data[(size_t)i * 64 + k] : 0.f;
                }
            h->out_a = push(h->h_w, a);
            h->out_b = push(h->h_w, h->raw["dynamics.noise_predictor.noise_predictor.to_scalar_output.bias"].data);
        }
        {   // static hoist of conv layer 0 (pf_device.h L0H_*): pure copies of the first pp message GVP's pieces
            const GvpSpec g = msg_spec(c, 0, ET_PP, 0);
            const std::vector<float>& W = h->raw[g.prefix + "to_feats_out.0.weight"].data;        // [128][144 + 17]
            const std::vector<float>& wh = h->raw[g.prefix + "Wh"].data;                          // [17][17]
            const std::vector<float>& wu = h->raw[g.prefix + "Wu"].data;                          // [17][16]
            std::vector<float> blk(L0H_SIZE, 0.f);
            if (g.vi == 17 && g.so == PF_S && g.si == PF_S + PF_R && g.vo == 16) {
                const int Kin = g.si + 17;
                for (int f = 0; f < PF_S; ++f) {
                    for (int k = 0; k < PF_R; ++k) blk[L0H_WR + (size_t)k * PF_S + f] = W[(size_t)f * Kin + PF_S + k];
                    for (int k = 0; k < 17; ++k) blk[L0H_WSH + (size_t)k * PF_S + f] = W[(size_t)f * Kin + g.si + k];
                    for (int k = 0; k < PF_S; ++k) blk[L0H_WHT + (size_t)k * PF_S + f] = W[(size_t)f * Kin + k];
                    blk[L0H_B + f] = h->raw[g.prefix + "to_feats_out.0.bias"].data[f];
                }
                for (int k = 0; k < 17; ++k) blk[L0H_WH0 + k] = wh[(size_t)0 * 17 + k];
                for (int k = 0; k < 17 * 16; ++k) blk[L0H_WU + k] = wu[k];
                for (int k = 0; k < 16; ++k) blk[L0H_BG + k] = h->raw[g.prefix + "scalar_to_vector_gates.bias"].data[k];
                const GvpSpec gp = msg_spec(c, 0, ET_PF, 0);                  // the pf etype's type table (n16 kernels)
                const std::vector<float>& Wp = h->raw[gp.prefix + "to_feats_out.0.weight"].data;
                for (int f = 0; f < PF_S; ++f) {
                    for (int k = 0; k < PF_S; ++k) blk[L0H_WHT_PF + (size_t)k * PF_S + f] = Wp[(size_t)f * Kin + k];
                    blk[L0H_B_PF + f] = h->raw[gp.prefix + "to_feats_out.0.bias"].data[f];
                }
            }
            h->l0h_off = push(h->h_w, blk);
            // center hoist (pf_cenhoist.h): the h_src blocks and biases of the ff / fp etypes' first message GVP, k-major
            h->l0c_off = 0;
            if (g.vi == 17 && g.so == PF_S && g.si == PF_S + PF_R && g.vo == 16) {
                std::vector<float> cb(L0C_SIZE, 0.f);
                const int Kin = g.si + 17;
                for (int k2 = 0; k2 < 2; ++k2) {
                    const GvpSpec gc = msg_spec(c, 0, k2 == 0 ? ET_FF : ET_FP, 0);
                    const std::vector<float>& Wc = h->raw[gc.prefix + "to_feats_out.0.weight"].data;
                    const std::vector<float>& bc = h->raw[gc.prefix + "to_feats_out.0.bias"].data;
                    const size_t wo = k2 == 0 ? L0C_WHT_FF : L0C_WHT_FP, bo = k2 == 0 ? L0C_B_FF : L0C_B_FP;
                    for (int f = 0; f < PF_S; ++f) {
                        for (int k = 0; k < PF_S; ++k) cb[wo + (size_t)k * PF_S + f] = Wc[(size_t)f * Kin + k];
                        cb[bo + f] = bc[f];
                    }
                }
                h->l0c_off = push(h->h_w, cb);
            }
        }
        {   // row-group quad streams, one contiguous stream per chain.  The pharm update chain of the last conv layer
            // comes last and is followed by the noise head's chain and to_scalar_output: the fused node + head kernel
            // streams straight through.  RG_TAIL_PAD quads of padding: the prefetch ring reads ahead of the last quad used.
            h->rg_msg.assign((size_t)c.n_convs * 4, 0);
            h->rg_upd.assign((size_t)c.n_convs * 2, 0);
            std::vector<float> st;
            auto flush = [&]() { const size_t off = push(h->h_w, st); st.clear(); return off; };
            auto chain = [&](auto spec_of, int n, int half = -1) {   // blocks of a chain: GVP j carries the gates of GVP j - 1
                GvpSpec prev;
                for (int j = 0; j < n; ++j) {
                    const GvpSpec g = spec_of(j);
                    pack_gvp_rg(h, g, j ? &prev : nullptr, st, half);
                    prev = g;
                }
                pack_flush_rg(h, prev, st);
            };
            for (int l = 0; l < c.n_convs; ++l)
                for (int et = 0; et < 4; ++et) {
                    chain([&](int j) { return msg_spec(c, l, et, j); }, c.n_message_gvps);
                    h->rg_msg[(size_t)l * 4 + et] = flush();
                }
            for (int l = 0; l < c.n_convs; ++l)
                for (int nt = 0; nt < 2; ++nt) {
                    if (l == c.n_convs - 1 && nt == 1) continue;
                    chain([&](int j) { return upd_spec(c, l, nt, j); }, c.n_update_gvps);
                    h->rg_upd[(size_t)l * 2 + nt] = flush();
                }
            chain([&](int j) { return upd_spec(c, c.n_convs - 1, 1, j); }, c.n_update_gvps);
            chain([&](int k) { return head_spec(c, k); }, c.n_noise_gvps);
            pack_out_rg(h, st);
            st.resize(st.size() + (size_t)RG_TAIL_PAD * 256, 0.f);
            h->rg_upd[(size_t)(c.n_convs - 1) * 2 + 1] = flush();
            // the same chains for the two-wave form: per chain wave 0's stream, then wave 1's
            h->rgs_msg.assign((size_t)c.n_convs * 4, 0);
            h->rgs_upd.assign((size_t)c.n_convs * 2, 0);
            h->rgs_upd_stride.assign((size_t)c.n_convs * 2, 0);
            for (int l = 0; l < c.n_convs; ++l)
                for (int et = 0; et < 4; ++et) {
                    for (int half = 0; half < 2; ++half) {
                        chain([&](int j) { return msg_spec(c, l, et, j); }, c.n_message_gvps, half);
                        if (half == 0) h->rgs_msg_stride = st.size();
                    }
                    h->rgs_msg[(size_t)l * 4 + et] = flush();
                }
            for (int l = 0; l < c.n_convs; ++l)
                for (int nt = 0; nt < 2; ++nt) {
                    const bool tail = l == c.n_convs - 1 && nt == 1;
                    for (int half = 0; half < 2; ++half) {
                        chain([&](int j) { return upd_spec(c, l, nt, j); }, c.n_update_gvps, half);
                        if (tail) {
                            chain([&](int k) { return head_spec(c, k); }, c.n_noise_gvps, half);
                            pack_out_rg(h, st);
                        }
                        if (half == 0) h->rgs_upd_stride[(size_t)l * 2 + nt] = st.size();
                    }
                    if (!tail) h->rgs_upd[(size_t)l * 2 + nt] = flush();
                }
            st.resize(st.size() + (size_t)RG_TAIL_PAD * 256, 0.f);
            h->rgs_upd[(size_t)(c.n_convs - 1) * 2 + 1] = flush();
        }
        h->n16_begin = h->h_w.size();            // everything packed from here on serves the n16 (inference-only) kernels
        if (c.n_message_gvps >= 2 && c.n_update_gvps >= 1) {   // n16 quad streams: per chain wave 0's stream, then waves 1..3
            h->n16_msg.assign((size_t)c.n_convs * 4, 0);
            h->n16_upd.assign((size_t)c.n_convs * 2, 0);
            std::vector<float> st;
            // (m0_at: index of the block that is a first message GVP in the full form, M0F; -1: block 0 has kind0)
            auto chain16 = [&](auto spec_of, int n, int kind0, size_t& stride, int m0_at = -1) {
                for (int w = 0; w < 4; ++w) {
                    const size_t b0 = st.size();
                    for (int j = 0; j < n; ++j) pack_n16(h, spec_of(j), j == m0_at ? N16_M0F : (j == 0 ? kind0 : N16_GEN), w, st);
                    st.resize(st.size() + (size_t)N16_TAIL_PAD * 256, 0.f);
                    stride = st.size() - b0;
                }
                const size_t off = push(h->h_w, st);
                for (int4 r : h->split_pending) { r.x += (int)off; h->split_tab.push_back(r); }
                h->split_pending.clear();
                st.clear();
                return off;
            };
            for (int l = 0; l < c.n_convs; ++l)
                for (int et = 0; et < 4; ++et)
                    h->n16_msg[(size_t)l * 4 + et] = chain16([&](int j) { return msg_spec(c, l, et, j); }, c.n_message_gvps, N16_M0F, h->n16_msg_stride);
            for (int et = 0; et < 4; ++et)
                h->n16_l0[et] = chain16([&](int j) { return msg_spec(c, 0, et, j); }, c.n_message_gvps,
                                        (et == ET_PP || et == ET_PF) ? N16_M0H : N16_M0Z, h->n16_l0_stride[et]);
            for (int et = 0; et < 4; ++et)      // center hoist: every etype's chain with a hoisted first block (ff / fp start from P_et rows)
                h->n16_l0h[et] = chain16([&](int j) { return msg_spec(c, 0, et, j); }, c.n_message_gvps, N16_M0H, h->n16_l0h_stride[et]);
            if (c.n_convs == 2)          // fused launch: conv layer 0's update chain of the source type, then the last layer's message chain
                for (int k = 0; k < 2; ++k) {
                    const int et = k == 0 ? ET_FF : ET_PF, nt = k == 0 ? 1 : 0;
                    h->n16_fused[k] = chain16([&](int j) { return j < c.n_update_gvps ? upd_spec(c, 0, nt, j) : msg_spec(c, 1, et, j - c.n_update_gvps); },
                                              c.n_update_gvps + c.n_message_gvps, N16_GEN, h->n16_fused_stride[k], c.n_update_gvps);
                }
            for (int l = 0; l < c.n_convs; ++l)
                for (int nt = 0; nt < 2; ++nt)
                    h->n16_upd[(size_t)l * 2 + nt] = chain16([&](int j) { return upd_spec(c, l, nt, j); }, c.n_update_gvps, N16_GEN, h->n16_upd_stride);
            {   // tail launch: the centers' update chain of the last conv layer, then the noise head (its last GVP padded, with to_scalar_output)
                const GvpSpec hl = head_spec(c, c.n_noise_gvps - 1);
                h->n16_tail = 0;
                if (c.pharm_nf <= 15 && c.n_noise_gvps >= 1 && hl.vi == 16 && hl.vo == 1 && hl.si == PF_S && hl.so == 64) {
                    for (int w = 0; w < 4; ++w) {
                        const size_t b0 = st.size();
                        for (int j = 0; j < c.n_update_gvps; ++j) pack_n16(h, upd_spec(c, c.n_convs - 1, 1, j), N16_GEN, w, st);
                        for (int k = 0; k + 1 < c.n_noise_gvps; ++k) pack_n16(h, head_spec(c, k), N16_GEN, w, st);
                        pack_n16_head_last(h, hl, w, st);
                        st.resize(st.size() + (size_t)N16_TAIL_PAD * 256, 0.f);
                        h->n16_tail_stride = st.size() - b0;
                    }
                    h->n16_tail = push(h->h_w, st);
                    for (int4 r : h->split_pending) { r.x += (int)h->n16_tail; h->split_tab.push_back(r); }
                    h->split_pending.clear();
                    st.clear();
                }
            }
        } else { h->n16_msg.clear(); h->n16_upd.clear(); h->n16_tail = 0; for (int et = 0; et < 4; ++et) h->n16_l0h[et] = 0; }
        while (h->h_w.size() % 64) h->h_w.push_back(0.f);
    };
    {
        std::map<std::string, RawTensor> keep;
        keep.swap(h->raw);
        size_t off = 0;
        for (const auto& kv : exp) {
            RawTensor t;
            t.shape = keep[kv.first].shape;
            t.data.resize(keep[kv.first].data.size());
            for (size_t i = 0; i < t.data.size(); ++i) t.data[i] = (float)(off + i + 1);
            off += t.data.size();
            h->raw[kv.first] = std::move(t);
        }
        h->h_map.clear();
        h->split_tab.clear(); h->split_pending.clear();
        if (off < (size_t(1) << 24)) {           // indices are exact in fp32
            h->split_record = N16_SPLIT != 0;
            pack_all();
            h->split_record = false;
            h->h_map.resize(h->h_w.size());
            for (size_t i = 0; i < h->h_w.size(); ++i) h->h_map[i] = (int)h->h_w[i] - 1;      // -1: zero padding
        }
        h->raw.swap(keep);
    }
    pack_all();
    if (h->d_w) { (void)hipFree(h->d_w); h->d_w = nullptr; }
    if (h->d_gvp) { (void)hipFree(h->d_gvp); h->d_gvp = nullptr; }
    PF_HIP(h, hipMalloc((void**)&h->d_w, h->h_w.size() * sizeof(float)));
    PF_HIP(h, hipMemcpy(h->d_w, h->h_w.data(), h->h_w.size() * sizeof(float), hipMemcpyHostToDevice));
    h->n_packed = h->h_w.size();
    if (h->d_map) { (void)hipFree(h->d_map); h->d_map = nullptr; }
    if (!h->h_map.empty()) {
        if (h->h_map.size() != h->h_w.size()) PF_FAIL(h, PF_ERR_STATE, "internal: gather map does not match the packed weights");
        PF_HIP(h, hipMalloc((void**)&h->d_map, h->h_map.size() * sizeof(int)));
        PF_HIP(h, hipMemcpy(h->d_map, h->h_map.data(), h->h_map.size() * sizeof(int), hipMemcpyHostToDevice));
        h->h_map.clear();
        h->h_map.shrink_to_fit();
    }
    if (h->d_split_tab) { (void)hipFree(h->d_split_tab); h->d_split_tab = nullptr; }
    h->n_split_tab = h->split_tab.size();
    if (h->n_split_tab) {
        PF_HIP(h, hipMalloc((void**)&h->d_split_tab, h->n_split_tab * sizeof(int4)));
        PF_HIP(h, hipMemcpy(h->d_split_tab, h->split_tab.data(), h->n_split_tab * sizeof(int4), hipMemcpyHostToDevice));
        h->split_tab.clear(); h->split_tab.shrink_to_fit();
    }
    for (const GvpOff& o : offs) {
        GvpW g;
        g.a_wh = h->d_w + o.wh; g.a_wu = h->d_w + o.wu; g.a_main = h->d_w + o.a_main; g.a_main_c = h->d_w + o.a_main_c; g.b_main = h->d_w + o.b_main;
        g.a_gate_c = h->d_w + o.a_gate_c; g.a_wh_c = h->d_w + o.wh_c; g.a_wu_c = h->d_w + o.wu_c;
        g.a_gate = h->d_w + o.a_gate; g.b_gate = h->d_w + o.b_gate;
        h->h_gvp.push_back(g);
    }
    PF_HIP(h, hipMalloc((void**)&h->d_gvp, h->h_gvp.size() * sizeof(GvpW)));
    PF_HIP(h, hipMemcpy(h->d_gvp, h->h_gvp.data(), h->h_gvp.size() * sizeof(GvpW), hipMemcpyHostToDevice));
    h->h_w.clear();
    h->h_w.shrink_to_fit();
    {   // gradient path: the parameters once more as one flat vector in state-dict order, and where each GVP's tensors sit
        std::vector<float> flat;
        h->flat_layout.clear();
        h->flat_index.clear(); h->flat_index_ok = false; h->edge_fx.clear();
        for (const auto& kv : exp) {
            const RawTensor& t = h->raw[kv.first];
            h->flat_layout.push_back({kv.first, {flat.size(), t.data.size()}});
            flat.insert(flat.end(), t.data.begin(), t.data.end());
        }
        h->nparams = flat.size();
        {   // class of every tensor: the kernel that differentiates it
            std::vector<TensorSeg> segs;
            for (const auto& kv : h->flat_layout) {
                const std::string& k = kv.first;
                TensorSeg sg{(int)kv.second.first, (int)(kv.second.first + kv.second.second), PFT_CLS_NONE, 0};
                if (kv.second.second == 0) sg.cls = PFT_CLS_NONE;
                else if (k.find("_encoder.") != std::string::npos) sg.cls = PFT_CLS_ENC;
                else if (k.find("noise_predictor.noise_predictor.") != std::string::npos) sg.cls = PFT_CLS_HEAD;
                else {
                    int layer = -1;
                    for (int l = 0; l < c.n_convs; ++l)
                        if (k.compare(0, conv_prefix(l).size(), conv_prefix(l)) == 0) layer = l;
                    if (layer < 0) PF_FAIL(h, PF_ERR_STATE, "internal: parameter %s has no gradient class", k.c_str());
                    if (layer >= 4) { h->n_tseg = -1; break; }           // the gradient path numbers its classes for n_convs <= 4
                    sg.cls = PFT_CLS_NODE + layer;
                    for (int et = 0; et < 4; ++et)
                        if (k.find(std::string("edge_message_fns.") + kEtKey[et] + ".") != std::string::npos) sg.cls = PFT_CLS_MSG + layer * 4 + et;
                }
                segs.push_back(sg);
            }
            {
                int lo = 0x7fffffff, hi = 0, tot = 0;
                for (const TensorSeg& sg : segs)
                    if (sg.cls == PFT_CLS_ENC) { lo = std::min(lo, sg.begin); hi = std::max(hi, sg.end); tot += sg.end - sg.begin; }
                if (tot == 0) { lo = hi = 0; }
                if (hi - lo != tot) PF_FAIL(h, PF_ERR_STATE, "internal: the encoders' parameters are not contiguous in the flat layout");
                h->enc_begin = lo; h->enc_n = hi - lo;
            }
            if (h->d_tseg) { (void)hipFree(h->d_tseg); h->d_tseg = nullptr; }
            h->n_tseg = c.n_convs <= 4 ? (int)segs.size() : -1;
            PF_HIP(h, hipMalloc((void**)&h->d_tseg, std::max<size_t>(segs.size(), 1) * sizeof(TensorSeg)));
            PF_HIP(h, hipMemcpy(h->d_tseg, segs.data(), segs.size() * sizeof(TensorSeg), hipMemcpyHostToDevice));
        }
        std::vector<GvpT> tab;
        auto mk = [&](const GvpSpec& g, bool sig) {
            GvpT t;
            t.o_Wh = (int)h->flat_offset(g.prefix + "Wh"); t.o_Wu = (int)h->flat_offset(g.prefix + "Wu");
            t.o_Wm = (int)h->flat_offset(g.prefix + "to_feats_out.0.weight"); t.o_bm = (int)h->flat_offset(g.prefix + "to_feats_out.0.bias");
            t.o_Wg = (int)h->flat_offset(g.prefix + "scalar_to_vector_gates.weight"); t.o_bg = (int)h->flat_offset(g.prefix + "scalar_to_vector_gates.bias");
            t.vi = g.vi; t.vo = g.vo; t.h = std::max(g.vi, g.vo); t.si = g.si; t.so = g.so; t.sig = sig ? 1 : 0;
            t.pk = (int)tab.size();
            return t;
        };
        for (int l = 0; l < c.n_convs; ++l)
            for (int et = 0; et < 4; ++et)
                for (int j = 0; j < c.n_message_gvps; ++j) tab.push_back(mk(msg_spec(c, l, et, j), true));
        for (int l = 0; l < c.n_convs; ++l)
            for (int nt = 0; nt < 2; ++nt)
                for (int j = 0; j < c.n_update_gvps; ++j) tab.push_back(mk(upd_spec(c, l, nt, j), true));
        for (int k = 0; k < c.n_noise_gvps; ++k) tab.push_back(mk(head_spec(c, k), k != c.n_noise_gvps - 1));
        if (h->d_flat) { (void)hipFree(h->d_flat); h->d_flat = nullptr; }
        if (h->d_gvpt) { (void)hipFree(h->d_gvpt); h->d_gvpt = nullptr; }
        if (h->d_wpack) { (void)hipFree(h->d_wpack); h->d_wpack = nullptr; }
        h->wpack_version = ~0ull;
        PF_HIP(h, hipMalloc((void**)&h->d_flat, std::max<size_t>(flat.size(), 1) * sizeof(float)));
        PF_HIP(h, hipMemcpy(h->d_flat, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice));
        h->n_gvpt = (int)tab.size();
        PF_HIP(h, hipMalloc((void**)&h->d_gvpt, tab.size() * sizeof(GvpT)));
        PF_HIP(h, hipMemcpy(h->d_gvpt, tab.data(), tab.size() * sizeof(GvpT), hipMemcpyHostToDevice));
    }
    h->t_have_fwd = false;
    h->committed = true;
    ++h->w_version;
    if (!h->d_l0c) PF_HIP(h, hipMalloc((void**)&h->d_l0c, 32 * sizeof(float)));
    if (h->d_ptab) { (void)hipFree(h->d_ptab); h->d_ptab = nullptr; }
    PF_HIP(h, hipMalloc((void**)&h->d_ptab, (size_t)L0_PTAB_SLOTS * L0_NTAB * c.rec_nf * PF_S * sizeof(float)));
    return PF_OK;
}

// prot_x / prot_h come either as device pointers (copied on the stream) or as host pointers (staged with the tables)
// pf_set_pocket_batch's pass over the pp edges for the common case -- destination-sorted, every edge inside its graph -- with
// 8 edges per instruction (a training loop binds a new batch every step: 0.65 M edges at 256 pockets, and the bind is on the
// step's host-side critical path).  Returns false when anything is unusual (unsorted, out of range, an edge across graphs):
// the scalar pass then runs and reports.  On success start[d] (d = 0 .. Np) = index of the first edge whose destination
// is >= d, i.e. the in-edge ranges of a destination-sorted list.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static bool pp_edges_fast_avx2(const int* src, const int* dst, int64_t n, const int* prot_ptr, int B, int Np,
                                                                int* start) {
    if (n <= 0 || dst[0] < 0 || dst[n - 1] >= Np) return false;
    // sortedness + boundaries in one sweep: a boundary after edge e (dst[e] < dst[e + 1]) starts the ranges of nodes dst[e] + 1 .. dst[e + 1]
    for (int d = 0; d <= dst[0]; ++d) start[d] = 0;
    int64_t e = 0;
    for (; e + 8 < n; e += 8) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(dst + e));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(dst + e + 1));
        if (_mm256_movemask_epi8(_mm256_cmpgt_epi32(a, b))) return false;                       // descending somewhere
        unsigned m = (unsigned)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(b, a)));
        while (m) {
            const int k = __builtin_ctz(m);
            m &= m - 1;
            const int lo = dst[e + k], hi = dst[e + k + 1];
            for (int d = lo + 1; d <= hi; ++d) start[d] = (int)(e + k + 1);
        }
    }
    for (; e + 1 < n; ++e) {
        if (dst[e] > dst[e + 1]) return false;
        for (int d = dst[e] + 1; d <= dst[e + 1]; ++d) start[d] = (int)(e + 1);
    }
    for (int d = dst[n - 1] + 1; d <= Np; ++d) start[d] = (int)n;
    // every source inside the atom range of its destination's graph (the destinations of graph g are the edges start[p0] .. start[p1])
    for (int g = 0; g < B; ++g) {
        const int lo = prot_ptr[g], hi = prot_ptr[g + 1];
        const int64_t a = start[lo], b = start[hi];
        const __m256i vlo = _mm256_set1_epi32(lo), vhi = _mm256_set1_epi32(hi);
        __m256i bad = _mm256_setzero_si256();
        int64_t i = a;
        for (; i + 8 <= b; i += 8) {
            const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
            bad = _mm256_or_si256(bad, _mm256_or_si256(_mm256_cmpgt_epi32(vlo, x), _mm256_cmpgt_epi32(x, _mm256_sub_epi32(vhi, _mm256_set1_epi32(1)))));
        }
        if (_mm256_movemask_epi8(bad)) return false;
        for (; i < b; ++i) if (src[i] < lo || src[i] >= hi) return false;
    }
    return true;
}
static bool pp_edges_fast(const int* src, const int* dst, int64_t n, const int* prot_ptr, int B, int Np, int* start) {
    static const bool ok = __builtin_cpu_supports("avx2") && getenv("PFDYN_NO_AVX2") == nullptr;
    return ok && pp_edges_fast_avx2(src, dst, n, prot_ptr, B, Np, start);
}
#else
static bool pp_edges_fast(const int*, const int*, int64_t, const int*, int, int, int*) { return false; }
#endif

static int set_pocket_batch_impl(pf_handle* h, int32_t B, const int32_t* prot_ptr, const int32_t* pharm_ptr,
                                 const float* dev_prot_x, const float* dev_prot_h, const float* host_prot_x, const float* host_prot_h,
                                 int64_t n_pp, const int32_t* pp_src, const int32_t* pp_dst, pf_stream stream) {
    // the pocket-group claim (pf_set_pocket_groups) belongs to THIS bind: it is taken off the handle before anything can
    // fail, so that a rejected bind never leaves it behind for the next, unrelated batch
    std::vector<int> rep_claim;
    if (h) rep_claim.swap(h->pending_rep);
    int rc = check_ready(h, false);
    if (rc) return rc;
    const bool from_host = host_prot_x != nullptr;
    if (B < 1 || !prot_ptr || !pharm_ptr || (from_host ? !host_prot_h : (!dev_prot_x || !dev_prot_h)) || n_pp < 0 ||
        (n_pp && (!pp_src || !pp_dst)))
        PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_batch: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const pf_config& c = h->cfg;
    if (prot_ptr[0] != 0 || pharm_ptr[0] != 0) PF_FAIL(h, PF_ERR_ARG, "ptr arrays must start at 0");
    for (int g = 0; g < B; ++g) {
        if (prot_ptr[g + 1] < prot_ptr[g] || pharm_ptr[g + 1] < pharm_ptr[g]) PF_FAIL(h, PF_ERR_ARG, "ptr arrays must be non-decreasing");
        if (pharm_ptr[g + 1] - pharm_ptr[g] > PF_MAXF)
            PF_FAIL(h, PF_ERR_ARG, "graph %d has %d pharmacophore centers (limit %d)", g, pharm_ptr[g + 1] - pharm_ptr[g], PF_MAXF);
    }
    static const bool timing = getenv("PFDYN_TIMING") != nullptr;
    double tm[8] = {0}; int tmi = 0;
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    auto mark = [&] { if (timing && tmi < 8) tm[tmi++] = now(); };
    mark();
    const int Np = prot_ptr[B], Nf = pharm_ptr[B], N = Np + Nf;
    // ---- host-side tables
    std::vector<int> gid(N);
    int max_np = 0, max_nf = 0;
    for (int g = 0; g < B; ++g) {
        max_np = std::max(max_np, prot_ptr[g + 1] - prot_ptr[g]);
        max_nf = std::max(max_nf, pharm_ptr[g + 1] - pharm_ptr[g]);
        for (int i = prot_ptr[g]; i < prot_ptr[g + 1]; ++i) gid[i] = g;
        for (int i = pharm_ptr[g]; i < pharm_ptr[g + 1]; ++i) gid[Np + i] = g;
    }
    // every argument check runs before the previous batch's state is touched: a rejected bind leaves the handle as it was.
    // One pass over the pp edges checks them and counts the in-degrees (a training loop binds a new batch every step and
    // this function is most of that step's host time): the graph of an edge is looked up only when the destination leaves
    // the atom range of the previous edge's graph -- edge lists come grouped by destination.
    std::vector<int> deg(Np + 1, 0);
    bool dst_sorted = true;                    // radius_graph and pf_build_pp_edges emit the edges grouped by destination, ascending:
    const bool deg_is_prefix = pp_edges_fast(pp_src, pp_dst, n_pp, prot_ptr, B, Np, deg.data());     // (the common case, 8 edges at a time)
    if (!deg_is_prefix) {                      // the stable sort below is then the identity and is skipped
        std::fill(deg.begin(), deg.end(), 0);
        // (per-edge increments, no branch on a change of destination: counting per run of equal destinations costs a
        // mispredicted branch per atom and measured 0.3 ms slower at 256 pockets)
        int prev_dst = -1, lo = 0, hi = 0;
        for (int64_t e = 0; e < n_pp; ++e) {
            const int sn = pp_src[e], dn = pp_dst[e];
            if ((unsigned)sn >= (unsigned)Np || (unsigned)dn >= (unsigned)Np) PF_FAIL(h, PF_ERR_ARG, "pp edge %lld out of range", (long long)e);
            if (dn < lo || dn >= hi) { const int g = gid[dn]; lo = prot_ptr[g]; hi = prot_ptr[g + 1]; }
            if (sn < lo || sn >= hi) PF_FAIL(h, PF_ERR_ARG, "pp edge %lld crosses graphs", (long long)e);
            dst_sorted &= dn >= prev_dst;
            prev_dst = dn;
            deg[dn + 1]++;
        }
    }
    if (c.message_norm_mode == PF_NORM_GRAPH && c.pf_k > 0)
        for (int g = 0; g < B; ++g)
            if (std::min(c.pf_k, prot_ptr[g + 1] - prot_ptr[g]) > 0 && pharm_ptr[g + 1] > pharm_ptr[g] && pharm_ptr[g + 1] - 1 >= Np)
                PF_FAIL(h, PF_ERR_ARG, "message_norm 0 with kNN pf edges: center index %d >= %d protein atoms "
                        "(the reference indexes the protein batch vector with it, dynamics_gvp.py:220)", pharm_ptr[g + 1] - 1, Np);
    if (!rep_claim.empty() && (int)rep_claim.size() != B)
        PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups named %d graphs, this batch has %d", (int)rep_claim.size(), B);
    // from here on the handle describes the new batch (a failure below -- a false pocket-group claim, a failing HIP call --
    // leaves it without a batch: loud, never the previous one's tables under the new sizes)
    h->have_batch = false;
    free_ws(h, true);
    h->B = B; h->Np = Np; h->Nf = Nf; h->N = N; h->Epp = n_pp;
    h->h_prot_ptr.assign(prot_ptr, prot_ptr + B + 1);
    h->h_pharm_ptr.assign(pharm_ptr, pharm_ptr + B + 1);
    h->max_np = max_np; h->max_nf = max_nf;
    if (!deg_is_prefix)
        for (int i = 0; i < Np; ++i) deg[i + 1] += deg[i];
    std::vector<int> pp_cnt(B, 0);
    // message_norm == 0 with kNN pf edges: the reference derives the per-graph pf / fp edge counts by looking the
    // pharmacophore-CENTER index of every edge up in the PROTEIN batch vector (dynamics_gvp.py:220), i.e. the min(k, Np_g)
    // edges of center j are booked on the graph that owns protein atom j.  gvp.py:506 normalises with those counts, so
    // they are reproduced (a pure function of the ptr arrays); the reference raises an IndexError when j >= Np_tot.
    std::vector<int> pfq;
    if (c.message_norm_mode == PF_NORM_GRAPH && c.pf_k > 0) {
        pfq.assign(B, 0);
        for (int g = 0; g < B; ++g) {
            const int kg = std::min(c.pf_k, prot_ptr[g + 1] - prot_ptr[g]);
            for (int j = pharm_ptr[g]; j < pharm_ptr[g + 1] && kg > 0; ++j) {
                if (j >= Np) PF_FAIL(h, PF_ERR_ARG, "message_norm 0 with kNN pf edges: center index %d >= %d protein atoms "
                                     "(the reference indexes the protein batch vector with it, dynamics_gvp.py:220)", j, Np);
                pfq[gid[j]] += kg;
            }
        }
    }
    // capacity of dynamic regions: ff, pf, fp and "pa" = compact copy of the pp edges into the active atoms
    h->h_reg.assign((size_t)4 * B, 0);
    h->h_cap.assign((size_t)4 * B, 0);
    std::vector<int> reg_act(B, 0), cap_act(B, 0), maxdeg(B, 0), epp_g(B, 0);
    for (int i = 0; i < Np; ++i) {
        maxdeg[gid[i]] = std::max(maxdeg[gid[i]], deg[i + 1] - deg[i]);
        epp_g[gid[i]] += deg[i + 1] - deg[i];
    }
    int64_t cursor = n_pp;
    int act_total = 0;
    for (int et = 0; et < 4; ++et)
        for (int g = 0; g < B; ++g) {
            const int np = prot_ptr[g + 1] - prot_ptr[g], nf = pharm_ptr[g + 1] - pharm_ptr[g];
            const int nact = c.pf_k > 0 ? std::min(np, nf * std::min(c.pf_k, np)) : (nf > 0 ? np : 0);
            int cap;
            if (et == ET_FF) cap = c.ff_k > 0 ? nf * std::min(c.ff_k, std::max(nf - 1, 0)) : nf * std::max(nf - 1, 0);
            else if (et < 3) cap = c.pf_k > 0 ? nf * std::min(c.pf_k, np) : nf * np;
            else {
                cap = (int)std::min<int64_t>(epp_g[g], (int64_t)nact * maxdeg[g]);
                reg_act[g] = act_total; cap_act[g] = nact; act_total += nact;
            }
            cursor = (cursor + 31) & ~int64_t(31);     // tiles are aligned to multiples of 32 slots (seg_tail in the kernels)
            h->h_reg[(size_t)et * B + g] = (int)cursor;
            h->h_cap[(size_t)et * B + g] = cap;
            cursor += cap;
        }
    if (cursor > 0x7fffffffLL / 128) PF_FAIL(h, PF_ERR_ARG, "edge capacity too large");
    h->Ecap = cursor;
    const int64_t Ecap = std::max<int64_t>(cursor, 1);
    // tiles: dynamic etypes first (they feed the short pharm-side chain), then pp
    std::vector<EdgeTile> et_tiles;
    et_tiles.reserve((size_t)n_pp / 32 + (size_t)(Ecap - n_pp) / 8 + 64);
    for (int et = 0; et < 3; ++et) {
        h->et_tile0[et] = (int)et_tiles.size();
        for (int g = 0; g < B; ++g) {
            const int cap = h->h_cap[(size_t)et * B + g], reg = h->h_reg[(size_t)et * B + g];
            for (int o = 0; o < cap; o += 32) et_tiles.push_back({reg + o, std::min(32, cap - o), et, et * B + g, o});
        }
        // The output of the last conv layer is consumed only on the pharm nodes (dynamics_gvp.py:91), so in
        // that layer only the etypes with a pharm destination (ff, pf: the first tiles) and only the pharm node
        // tiles are computed; the reference computes and discards the protein side.
        if (et == ET_PF) h->n_edge_tiles_last = (int)et_tiles.size();
    }
    h->et_tile0[3] = (int)et_tiles.size();
    for (int64_t o = 0; o < n_pp; o += 32) et_tiles.push_back({(int)o, (int)std::min<int64_t>(32, n_pp - o), ET_PP, -1, 0});
    h->et_tile0[4] = (int)et_tiles.size();
    std::vector<NodeTile> n_tiles, h_tiles;
    n_tiles.reserve((size_t)N / 32 + 8);
    for (int o = 0; o < Nf; o += 32) {
        n_tiles.push_back({Np + o, std::min(32, Nf - o), 1, -1, 0, 0});
        h_tiles.push_back({Np + o, std::min(32, Nf - o), 1, -1, 0, 0});
    }
    for (int o = 0; o < Np; o += 32) n_tiles.push_back({o, std::min(32, Np - o), 0, -1, 0, 0});
    h->n_edge_tiles = (int)et_tiles.size();
    h->n_node_tiles = (int)n_tiles.size();
    h->n_head_tiles = (int)h_tiles.size();
    h->n_node_tiles_last = (int)h_tiles.size();          // pharm tiles come first in n_tiles
    // pruned layer: ff, pf, fp tiles + pa tiles; pharm node tiles + tiles over the active-atom lists
    std::vector<EdgeTile> et_act;
    for (int et = 0; et < 4; ++et) {
        h->et_tile0_act[et] = (int)et_act.size();
        for (int g = 0; g < B; ++g) {
            const int cap = h->h_cap[(size_t)et * B + g], reg = h->h_reg[(size_t)et * B + g];
            for (int o = 0; o < cap; o += 32) et_act.push_back({reg + o, std::min(32, cap - o), et == 3 ? (int)ET_PP : et, et * B + g, o});
        }
    }
    h->et_tile0_act[4] = (int)et_act.size();
    std::vector<NodeTile> n_act(h_tiles);
    for (int g = 0; g < B; ++g)
        for (int o = 0; o < cap_act[g]; o += 32) n_act.push_back({reg_act[g] + o, std::min(32, cap_act[g] - o), 0, 4 * B + g, o, 1});
    h->n_edge_tiles_act = (int)et_act.size();
    h->n_node_tiles_act = (int)n_act.size();
    mark();      // 1: host tables built
    // ---- workspace layout: [table section: host-built, uploaded with one copy][zero section][scratch]
    const size_t n_eta = et_act.size() + 16, n_nta = n_act.size() + 16, n_et = et_tiles.size() + 16, n_nt = n_tiles.size() + 16,
                 n_ht = h_tiles.size() + 16;
    auto rnd = [](size_t b) { return (b + 255) & ~size_t(255); };
    size_t off = 0;
    auto place = [&](size_t b) { const size_t o = off; off += rnd(b); return o; };
    // table section (a buffer of its own: offsets relative to d_tab[w])
    const size_t o_pptr = place((B + 1) * 4), o_fptr = place((B + 1) * 4), o_gid = place((size_t)N * 4), o_reg = place((size_t)4 * B * 4),
                 o_regact = place((size_t)B * 4), o_eta = place(n_eta * sizeof(EdgeTile)), o_nta = place(n_nta * sizeof(NodeTile)),
                 o_esrc = place(Ecap * 4), o_edst = place(Ecap * 4), o_ins = place((size_t)4 * N * 4), o_inc = place((size_t)4 * N * 4),
                 o_ppc = place((size_t)B * 4), o_et = place(n_et * sizeof(EdgeTile)), o_nt = place(n_nt * sizeof(NodeTile)),
                 o_ht = place(n_ht * sizeof(NodeTile)), o_pfq = place((size_t)B * 4),
                 o_regs = place((size_t)4 * B * 4), o_pas = place((size_t)B * 4), o_repb = place((size_t)B * 4);
    const size_t index_bytes = off;
    const size_t o_px0 = place((size_t)Np * 3 * 4 + 16), o_ph0 = place((size_t)Np * c.rec_nf * 4 + 16);
    const size_t table_bytes = from_host ? off : index_bytes;      // what the single upload covers
    const size_t table_total = off;
    off = 0;                                                       // the workspace proper starts with the zero section
    // zero section (cleared with one launch per bind)
    const size_t o_dyn = place((size_t)5 * B * 4), o_act = place((size_t)(act_total + 1) * 4), o_flag = place(256), o_gnorm = place((size_t)2 * B * 4),
                 o_need = place((size_t)std::max(Np, 1) * 4),
                 o_lpart = place(64 + (size_t)((Nf + 63) / 64) * 8 * sizeof(float)),       // k_loss_eval's ticket (re-armed by its last block) + partial sums
                 o_pastamp = place((size_t)std::max(Np, 1) * 4), o_pasame = place((size_t)B * 4);      // speculative "pa" messages: per-atom step stamps, per-graph verdicts
    const size_t zero_bytes = off;
    // scratch
    // (a second set of message rows for the last conv layer: the fused launch of small n_convs = 2 batches writes them while conv
    // layer 0's are still being read)
    const bool msg2 = c.n_convs == 2 && (long)h->n_edge_tiles_act * 32 <= h->pol.n16_rows_max;
    const int64_t rec_slots = B <= 64 ? Ecap : 0;      // edge records: small batches only (the n16 fused launch)
    const size_t o_xn = place((size_t)N * 16),
                 o_fh = place((size_t)Nf * c.pharm_nf * 4 + 16), o_t = place((size_t)B * 4),
                 o_h0 = place((size_t)N * PF_S * 4), o_h1 = place((size_t)N * PF_S * 4), o_v0 = place((size_t)N * 48 * 4), o_v1 = place((size_t)N * 48 * 4),
                 o_ms = place((size_t)(Ecap + 1) * PF_S * 4), o_mv = place((size_t)(Ecap + 1) * 48 * 4),
                 o_ms2 = place(msg2 ? (size_t)(Ecap + 1) * PF_S * 4 : 16), o_mv2 = place(msg2 ? (size_t)(Ecap + 1) * 48 * 4 : 16),
                 o_eh = place((size_t)Nf * c.pharm_nf * 4 + 16), o_ex = place((size_t)Nf * 3 * 4 + 16), o_c0 = place((size_t)B * 3 * 4), o_c1 = place((size_t)B * 3 * 4),
                 o_pre = place((size_t)std::max(Np, 1) * PF_S * 4), o_eorig = place(Ecap * 4), o_ptype = place((size_t)std::max(Np, 1) * 4),
                 o_rec = place((size_t)(h->pol.edge_rec ? 3 * rec_slots : 0) * 16 + 16),
                 o_zs = place((size_t)std::max<int64_t>(n_pp, 1) * PF_S * 4), o_ptpg = place((size_t)B * L0_NTAB * c.rec_nf * PF_S * 4),
                 o_xchg = place((size_t)2 * std::max(Nf, 1) * PF_XCHG_STRIDE * sizeof(unsigned int)),      // (+ the center hoist's copy)
                 o_cenh = place((size_t)std::max(Nf, 1) * PF_S * 4), o_cenp = place((size_t)2 * std::max(Nf, 1) * PF_S * 4),
                 o_snap = place((size_t)2 * (std::max(Nf, 1) * c.pharm_nf + 4) * 4);
    const size_t bytes = off;
    bool fresh = false;
    if (h->ws_capacity < bytes + 4096) {
        // launches of the previous batch may still read the old workspace
        PF_HIP(h, hipDeviceSynchronize());
        if (h->d_ws) { (void)hipFree(h->d_ws); h->d_ws = nullptr; h->ws_capacity = 0; }
        const size_t want = bytes + bytes / 8 + 4096;           // head room: the next batch of similar size fits without a realloc
        PF_HIP(h, hipMalloc(&h->d_ws, want));
        h->ws_capacity = want;
        fresh = true;
    }
    char* const base = reinterpret_cast<char*>(h->d_ws);
    auto at = [&](size_t o) { return base + o; };
    if (!h->d_xstat) {                                           // once per handle: the exchange's time-out counter and its pinned mirror
        PF_HIP(h, hipMalloc((void**)&h->d_xstat, 64));
        PF_HIP(h, hipMemset(h->d_xstat, 0, 64));
        PF_HIP(h, hipHostMalloc((void**)&h->xstat_host, 64, hipHostMallocDefault));
        *h->xstat_host = 0; h->xstat_ack = 0;
    }
    // the table buffer of this bind
    const int tw = h->tab_next;
    h->tab_next ^= 1;
    if (!h->s_copy) PF_HIP(h, hipStreamCreateWithFlags(&h->s_copy, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        if (!h->tab_guard[k]) PF_HIP(h, hipEventCreateWithFlags(&h->tab_guard[k], hipEventDisableTiming));
        if (!h->tab_up[k]) PF_HIP(h, hipEventCreateWithFlags(&h->tab_up[k], hipEventDisableTiming));
    }
    if (h->tab_cap[tw] < table_total + 4096) {
        PF_HIP(h, hipDeviceSynchronize());                       // launches of two binds ago may still read the old buffer
        if (h->d_tab[tw]) { (void)hipFree(h->d_tab[tw]); h->d_tab[tw] = nullptr; h->tab_cap[tw] = 0; }
        const size_t want = table_total + table_total / 8 + 4096;
        PF_HIP(h, hipMalloc(&h->d_tab[tw], want));
        h->tab_cap[tw] = want;
        h->tab_guard_set[tw] = false;
    }
    char* const tbase = reinterpret_cast<char*>(h->d_tab[tw]);
    auto tat = [&](size_t o) { return tbase + o; };
    h->d_prot_ptr = (int*)tat(o_pptr); h->d_pharm_ptr = (int*)tat(o_fptr); h->d_gid = (int*)tat(o_gid); h->d_reg = (int*)tat(o_reg);
    h->d_reg_act = (int*)tat(o_regact); h->d_edge_tiles_act = (EdgeTile*)tat(o_eta); h->d_node_tiles_act = (NodeTile*)tat(o_nta);
    h->d_esrc = (int*)tat(o_esrc); h->d_edst = (int*)tat(o_edst); h->d_in_start = (int*)tat(o_ins); h->d_in_cnt = (int*)tat(o_inc);
    h->d_pp_cnt = (int*)tat(o_ppc); h->d_edge_tiles = (EdgeTile*)tat(o_et); h->d_node_tiles = (NodeTile*)tat(o_nt);
    h->d_head_tiles = (NodeTile*)tat(o_ht); h->d_pfq_cnt = pfq.empty() ? nullptr : (int*)tat(o_pfq);
    h->d_reg_share = (int*)tat(o_regs); h->d_pa_static = (int*)tat(o_pas); h->d_rep_base = (int*)tat(o_repb); h->d_need = (int*)at(o_need);
    h->need_stamp = 0; h->edges_stamp = 0;
    h->d_dyn_cnt = (int*)at(o_dyn); h->d_act_ids = (int*)at(o_act); h->d_l0flag = (int*)at(o_flag); h->d_gnorm = (float*)at(o_gnorm);
    h->d_xn = (float4*)at(o_xn); h->d_prot_x0 = (float*)tat(o_px0); h->d_prot_h0 = (float*)tat(o_ph0); h->d_pharm_h = (float*)at(o_fh);
    h->d_t = (float*)at(o_t); h->d_h[0] = (float*)at(o_h0); h->d_h[1] = (float*)at(o_h1); h->d_v[0] = (float*)at(o_v0); h->d_v[1] = (float*)at(o_v1);
    h->d_msg_s2 = msg2 ? (float*)at(o_ms2) : nullptr; h->d_msg_v2 = msg2 ? (float*)at(o_mv2) : nullptr;
    h->d_msg_s = (float*)at(o_ms); h->d_msg_v = (float*)at(o_mv); h->d_eps_h = (float*)at(o_eh); h->d_eps_x = (float*)at(o_ex);
    h->d_com_init = (float*)at(o_c0); h->d_com_tmp = (float*)at(o_c1); h->d_pre = (float*)at(o_pre); h->d_eorig = (int*)at(o_eorig);
    h->d_ptype = (int*)at(o_ptype); h->d_zs = (float*)at(o_zs); h->d_ptab_pg = (float*)at(o_ptpg);
    h->d_rec = (h->pol.edge_rec && rec_slots > 0) ? (int4*)at(o_rec) : nullptr; h->rec_valid = false;
    h->d_xchg = (unsigned int*)at(o_xchg); h->d_lpart = (float*)at(o_lpart);
    h->d_xchg2 = h->d_xchg + (size_t)std::max(Nf, 1) * PF_XCHG_STRIDE;
    h->d_cen_h = (float*)at(o_cenh); h->d_cen_p = (float*)at(o_cenp);
    h->d_snap[0] = (float*)at(o_snap); h->d_snap[1] = h->d_snap[0] + (size_t)std::max(Nf, 1) * c.pharm_nf + 4;
    h->cen_valid = false; h->snap_cur = -1;
    h->d_pa_stamp = (int*)at(o_pastamp); h->d_pa_same = (int*)at(o_pasame); h->spec_valid = false; h->e0_saved = false;
    mark();      // 2: workspace ready
    // ---- stage the tables in pinned memory and upload them with one asynchronous copy
    const int sb = h->stage_next;
    h->stage_next ^= 1;
    if (!h->stage_ev[sb]) PF_HIP(h, hipEventCreateWithFlags(&h->stage_ev[sb], hipEventDisableTiming));
    else PF_HIP(h, hipEventSynchronize(h->stage_ev[sb]));       // the copy that last read this buffer (two binds ago) is done
    if (h->stage_cap[sb] < table_bytes) {
        if (h->stage[sb]) (void)hipHostFree(h->stage[sb]);
        h->stage[sb] = nullptr; h->stage_cap[sb] = 0;
        PF_HIP(h, hipHostMalloc(&h->stage[sb], table_bytes + table_bytes / 4 + 4096, hipHostMallocDefault));
        h->stage_cap[sb] = table_bytes + table_bytes / 4 + 4096;
    }
    mark();      // 3: staging buffer ready
    char* const st = reinterpret_cast<char*>(h->stage[sb]);
    // pp edges sorted by destination (stable counting sort): CSR-by-dst.  The big index arrays are built in the staging
    // buffer itself (5 MB of edges and 2 MB of in-edge ranges at 256 pockets: no intermediate copies)
    int* const esrc = reinterpret_cast<int*>(st + o_esrc);
    int* const edst = reinterpret_cast<int*>(st + o_edst);
    int* const in_start = reinterpret_cast<int*>(st + o_ins);      // [4 slots][N]: BuildParams::in_start
    int* const in_cnt = reinterpret_cast<int*>(st + o_inc);
    memset(esrc + n_pp, 0, (size_t)(Ecap - n_pp) * 4);
    memset(edst + n_pp, 0, (size_t)(Ecap - n_pp) * 4);
    memset(in_start, 0, (size_t)4 * N * 4);
    memset(in_cnt, 0, (size_t)4 * N * 4);
    {
        if (dst_sorted) {
            if (n_pp > 0) { memcpy(esrc, pp_src, (size_t)n_pp * 4); memcpy(edst, pp_dst, (size_t)n_pp * 4); }
            for (int g = 0; g < B; ++g) pp_cnt[g] = deg[prot_ptr[g + 1]] - deg[prot_ptr[g]];
        } else {
            std::vector<int> fill(deg.begin(), deg.end() - 1);
            for (int64_t e = 0; e < n_pp; ++e) {
                const int pos = fill[pp_dst[e]]++;
                esrc[pos] = pp_src[e];
                edst[pos] = pp_dst[e];
                pp_cnt[gid[pp_dst[e]]]++;
            }
        }
        for (int i = 0; i < Np; ++i) { in_start[(size_t)N + i] = deg[i]; in_cnt[(size_t)N + i] = deg[i + 1] - deg[i]; }
    }
    // ---- pocket sharing (pf_set_pocket_groups): verify the caller's claim and prepare the tables of the sharing mode
    bool share = false;
    std::vector<int> rep_base(B, 0);
    h->share_ok = false; h->share_rows = 0;
    h->h_share_start.assign(B, 0); h->h_share_cnt.assign(B, 0);
    if (!rep_claim.empty()) {
        std::vector<int> rep;
        rep.swap(rep_claim);
        if ((int)rep.size() != B) PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups named %d graphs, this batch has %d", (int)rep.size(), B);
        long dense = 0, percopy = 0;
        int nrep = 0;
        for (int g = 0; g < B; ++g) {
            const int r = rep[g];
            if (r < 0 || r >= B || rep[r] != r) PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups: graph %d names %d, which is not a representative", g, r);
            const int np = prot_ptr[g + 1] - prot_ptr[g];
            if (np != prot_ptr[r + 1] - prot_ptr[r] || epp_g[g] != epp_g[r])
                PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups: graph %d is not a copy of graph %d (%d vs %d atoms, %d vs %d pp edges)",
                        g, r, np, prot_ptr[r + 1] - prot_ptr[r], epp_g[g], epp_g[r]);
            if (r != g) {
                // same static graph: in-degrees and (destination-sorted) sources, pocket-local
                const int p0g = prot_ptr[g], p0r = prot_ptr[r];
                for (int i = 0; i < np; ++i)
                    if (deg[p0g + i + 1] - deg[p0g + i] != deg[p0r + i + 1] - deg[p0r + i])
                        PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups: graph %d is not a copy of graph %d (pp in-degree of atom %d)", g, r, i);
                const int eg = deg[p0g], er = deg[p0r];
                for (int k = 0; k < epp_g[g]; ++k)
                    if (esrc[eg + k] - p0g != esrc[er + k] - p0r)
                        PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups: graph %d is not a copy of graph %d (pp edge %d)", g, r, k);
                if (from_host && (memcmp(host_prot_x + (size_t)p0g * 3, host_prot_x + (size_t)p0r * 3, (size_t)np * 12) ||
                                  memcmp(host_prot_h + (size_t)p0g * c.rec_nf, host_prot_h + (size_t)p0r * c.rec_nf, (size_t)np * c.rec_nf * 4)))
                    PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups: graph %d is not a copy of graph %d (coordinates / features differ)", g, r);
            } else { dense += epp_g[g]; ++nrep; }
            {   // what the per-copy form computes for this graph: the pp in-edges of its active atoms -- at most nf k
                // of them, about 60 % of that once the centers' neighbour sets overlap -- at the pocket's mean in-degree
                const int nf = pharm_ptr[g + 1] - pharm_ptr[g];
                const int nact = c.pf_k > 0 ? std::min(np, nf * std::min(c.pf_k, np)) : (nf > 0 ? np : 0);
                percopy += np > 0 ? (long)(0.6 * nact * (double)epp_g[g] / np) : 0;
            }
        }
        // worth it when the representatives' static edges are clearly fewer than the per-copy edges they replace (about
        // half of them at 30 copies of a 256-atom pocket); the compact work list must cover 4 B regions
        // (with the per-step need stamps only the union of the copies' active atoms is computed, so what a
        // representative costs beyond that is launching the idle groups of its static range)
        share = nrep < B && 4 * B <= 1024 && (dense * 4 <= percopy * 3 || nrep * 4 <= B);
        if (share) {
            for (int g = 0; g < B; ++g) {
                const int r = rep[g], p0g = prot_ptr[g], p0r = prot_ptr[r], np = prot_ptr[g + 1] - p0g;
                for (int i = 0; i < np; ++i) {         // slot 3: the representative's static in-edge range of the same atom
                    in_start[(size_t)3 * N + p0g + i] = deg[p0r + i];
                    in_cnt[(size_t)3 * N + p0g + i] = deg[p0r + i + 1] - deg[p0r + i];
                }
                rep_base[g] = p0r;
                if (r == g) { h->h_share_start[g] = deg[p0g]; h->h_share_cnt[g] = epp_g[g]; }
            }
            h->share_rows = dense;
            for (int et = 0; et < 3; ++et) for (int g = 0; g < B; ++g) h->share_rows += h->h_cap[(size_t)et * B + g];
            h->share_ok = true;
        }
    }
    memcpy(st + o_pptr, prot_ptr, (size_t)(B + 1) * 4);
    memcpy(st + o_fptr, pharm_ptr, (size_t)(B + 1) * 4);
    memcpy(st + o_gid, gid.data(), (size_t)N * 4);
    memcpy(st + o_reg, h->h_reg.data(), (size_t)4 * B * 4);
    memcpy(st + o_regact, reg_act.data(), (size_t)B * 4);
    if (!et_act.empty()) memcpy(st + o_eta, et_act.data(), et_act.size() * sizeof(EdgeTile));
    if (!n_act.empty()) memcpy(st + o_nta, n_act.data(), n_act.size() * sizeof(NodeTile));
    memcpy(st + o_ppc, pp_cnt.data(), (size_t)B * 4);
    if (!et_tiles.empty()) memcpy(st + o_et, et_tiles.data(), et_tiles.size() * sizeof(EdgeTile));
    if (!n_tiles.empty()) memcpy(st + o_nt, n_tiles.data(), n_tiles.size() * sizeof(NodeTile));
    if (!h_tiles.empty()) memcpy(st + o_ht, h_tiles.data(), h_tiles.size() * sizeof(NodeTile));
    if (!pfq.empty()) memcpy(st + o_pfq, pfq.data(), (size_t)B * 4);
    {
        std::vector<int> regs(h->h_reg);
        for (int g = 0; g < B && share; ++g) regs[(size_t)3 * B + g] = h->h_share_start[g];
        memcpy(st + o_regs, regs.data(), (size_t)4 * B * 4);
        if (share) memcpy(st + o_pas, h->h_share_cnt.data(), (size_t)B * 4); else memset(st + o_pas, 0, (size_t)B * 4);
        memcpy(st + o_repb, rep_base.data(), (size_t)B * 4);
    }
    int host_onehot = -1;
    if (from_host) {
        memcpy(st + o_px0, host_prot_x, (size_t)Np * 3 * 4);
        memcpy(st + o_ph0, host_prot_h, (size_t)Np * c.rec_nf * 4);
        // the one-hot verdict (static hoist) on the host copy: nobody will wait for the device-side check
        host_onehot = Np > 0 ? 1 : 0;
        for (int i = 0; i < Np && host_onehot; ++i) {
            int ones = 0;
            for (int k = 0; k < c.rec_nf; ++k) {
                const float x = host_prot_h[(size_t)i * c.rec_nf + k];
                if (x == 1.0f) ++ones; else if (x != 0.0f) host_onehot = 0;
            }
            if (ones != 1) host_onehot = 0;
        }
    }
    mark();      // 4: staged
    // the upload: on the copy stream, once everything that read this table buffer (the bind before the previous one and its
    // steps) has finished; the caller's stream continues when it has arrived.  The guard of the OTHER buffer is recorded now:
    // what is enqueued on the caller's stream at this point is everything that reads it.
    if (h->tab_guard_set[tw]) PF_HIP(h, hipStreamWaitEvent(h->s_copy, h->tab_guard[tw], 0));
    PF_HIP(h, hipEventRecord(h->tab_guard[tw ^ 1], s));
    h->tab_guard_set[tw ^ 1] = true;
    PF_HIP(h, hipMemcpyAsync(tbase, st, table_bytes, hipMemcpyHostToDevice, h->s_copy));
    PF_HIP(h, hipEventRecord(h->stage_ev[sb], h->s_copy));
    PF_HIP(h, hipEventRecord(h->tab_up[tw], h->s_copy));
    PF_HIP(h, hipStreamWaitEvent(s, h->tab_up[tw], 0));
    mark();      // 5: upload enqueued
    // Message rows: the node kernels read only rows the edge kernels of the same layer wrote (the last slot of every
    // aligned group a destination's segment touches) and the all-zero row Ecap, so a reused workspace needs only that row
    // cleared; a fresh allocation is cleared once in full
    {
        ZeroBatch zb(s);
        zb.add(base, zero_bytes);
        zb.add(h->d_v[0], (size_t)N * 48 * 4);
        if (fresh) {
            zb.add(h->d_msg_s, (size_t)(Ecap + 1) * PF_S * 4);
            zb.add(h->d_msg_v, (size_t)(Ecap + 1) * 48 * 4);
        } else {
            zb.add(h->d_msg_s + (size_t)Ecap * PF_S, PF_S * 4);
            zb.add(h->d_msg_v + (size_t)Ecap * 48, 48 * 4);
        }
        if (h->d_msg_s2) {
            if (fresh) {
                zb.add(h->d_msg_s2, (size_t)(Ecap + 1) * PF_S * 4);
                zb.add(h->d_msg_v2, (size_t)(Ecap + 1) * 48 * 4);
            } else {
                zb.add(h->d_msg_s2 + (size_t)Ecap * PF_S, PF_S * 4);
                zb.add(h->d_msg_v2 + (size_t)Ecap * 48, 48 * 4);
            }
        }
        zb.flush();
    }
    h->zero_row = (int)Ecap;
    if (!from_host) {
        pfk_copy2(dev_prot_x, h->d_prot_x0, (size_t)Np * 3, dev_prot_h, h->d_prot_h0, (size_t)Np * c.rec_nf, s);
    }
    h->share_check = 1;
    if (!from_host && h->share_ok) {             // the claim's rows are on the device only: compared there, verdict read back lazily
        pfk_verify_copies(h->d_prot_x0, h->d_prot_h0, h->d_gid, h->d_prot_ptr, h->d_rep_base, Np, c.rec_nf, h->d_l0flag + 1, s);
        h->share_check = 0;
    }
    pfk_load_coords(h->d_prot_x0, h->d_xn, Np, h->d_gid, nullptr, 0.f, s);
    {
        L0HoistParams lp{};
        lp.prot_h0 = h->d_prot_h0; lp.Np = Np; lp.rec_nf = c.rec_nf; lp.ptype = h->d_ptype; lp.flag = h->d_l0flag;
        lp.zs = reinterpret_cast<float*>(h->d_eorig); lp.Epp = (int)Ecap;      // static slot of every edge slot: the identity
        pfk_l0_hoist(&lp, 2, s);
    }
    // the one-hot verdict of k_l0_types comes back through pinned memory; nobody waits for it here (l0_resolve_onehot)
    if (!h->l0flag_host) PF_HIP(h, hipHostMalloc((void**)&h->l0flag_host, 64, hipHostMallocDefault));
    if (!h->l0flag_ev) PF_HIP(h, hipEventCreateWithFlags(&h->l0flag_ev, hipEventDisableTiming));
    else PF_HIP(h, hipEventSynchronize(h->l0flag_ev));           // the previous bind's read-back (long done) before its target is reused
    h->l0flag_host[0] = 1; h->l0flag_host[1] = 1;
    PF_HIP(h, hipMemcpyAsync(h->l0flag_host, h->d_l0flag, 8, hipMemcpyDeviceToHost, s));
    PF_HIP(h, hipEventRecord(h->l0flag_ev, s));
    mark();      // 6: everything enqueued
    if (timing) fprintf(stderr, "[pf_set_pocket_batch] B=%d checks+tables %.2f ws %.2f stage-wait %.2f index arrays (in staging) %.2f upload %.2f launches+waits %.2f ms (fresh %d, %zu MB)\n",
                        B, tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], tm[5] - tm[4], tm[6] - tm[5], (int)fresh, bytes >> 20);
    h->l0_state = 0; h->l0_onehot = false;
    if (host_onehot >= 0) { h->l0_state = host_onehot ? 1 : 2; h->l0_onehot = host_onehot == 1; }
    h->zs_version = 0; h->zs_batch_coords = false; h->coords_custom = false;
    h->have_batch = true;
    h->sampling = false;
    h->edges_built = false; h->rec_valid = false;
    return PF_OK;
}

int pf_set_pocket_batch(pf_handle* h, int32_t B, const int32_t* prot_ptr, const int32_t* pharm_ptr,
                        const float* dev_prot_x, const float* dev_prot_h, int64_t n_pp, const int32_t* pp_src,
                        const int32_t* pp_dst, pf_stream stream) {
    if (h && (!dev_prot_x || !dev_prot_h)) {
        h->pending_rep.clear();                  // a pocket-group claim never outlives the bind it was made for, rejected or not
        PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_batch: bad argument");
    }
    return set_pocket_batch_impl(h, B, prot_ptr, pharm_ptr, dev_prot_x, dev_prot_h, nullptr, nullptr, n_pp, pp_src, pp_dst, stream);
}

int pf_set_pocket_batch_host(pf_handle* h, int32_t B, const int32_t* prot_ptr, const int32_t* pharm_ptr,
                             const float* host_prot_x, const float* host_prot_h, int64_t n_pp, const int32_t* pp_src,
                             const int32_t* pp_dst, pf_stream stream) {
    if (h && (!host_prot_x || !host_prot_h)) {
        h->pending_rep.clear();
        PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_batch_host: bad argument");
    }
    return set_pocket_batch_impl(h, B, prot_ptr, pharm_ptr, nullptr, nullptr, host_prot_x, host_prot_h, n_pp, pp_src, pp_dst, stream);
}

int pf_set_pocket_groups(pf_handle* h, int32_t B, const int32_t* host_rep) {
    if (!h) return PF_ERR_ARG;
    if (B < 0 || (B > 0 && !host_rep)) PF_FAIL(h, PF_ERR_ARG, "pf_set_pocket_groups: bad argument");
    h->pending_rep.assign(host_rep, host_rep + B);
    return PF_OK;
}

int pf_declare_onehot_features(pf_handle* h, int32_t is_onehot) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    h->l0_state = is_onehot ? 1 : 2;
    h->l0_onehot = is_onehot != 0 && h->Np > 0;
    return PF_OK;
}

int64_t pf_build_pp_edges(pf_handle* h, int32_t B, const int32_t* prot_ptr, const float* dev_prot_x, int32_t max_nb,
                          int32_t* host_src, int32_t* host_dst, int64_t capacity, pf_stream stream) {
    if (!h || B < 1 || !prot_ptr || !dev_prot_x) return PF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int Np = prot_ptr[B];
    float4* xn = nullptr; int *dptr = nullptr, *ddeg = nullptr, *doff = nullptr, *dsrc = nullptr, *ddst = nullptr;
    std::vector<int> deg(std::max(Np, 1)), off(std::max(Np, 1) + 1, 0);
    int64_t total = 0;
    int rc = PF_OK;
    auto cleanup = [&]() {
        if (xn) (void)hipFree(xn); if (dptr) (void)hipFree(dptr); if (ddeg) (void)hipFree(ddeg);
        if (doff) (void)hipFree(doff); if (dsrc) (void)hipFree(dsrc); if (ddst) (void)hipFree(ddst);
    };
#define PF_HIP2(call) do { hipError_t _e = (call); if (_e != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(_e); cleanup(); return PF_ERR_HIP; } } while (0)
    PF_HIP2(hipMalloc((void**)&xn, (size_t)std::max(Np, 1) * 16));
    PF_HIP2(hipMalloc((void**)&dptr, (size_t)(B + 1) * 4));
    PF_HIP2(hipMalloc((void**)&ddeg, (size_t)std::max(Np, 1) * 4));
    PF_HIP2(hipMalloc((void**)&doff, (size_t)std::max(Np, 1) * 4));
    PF_HIP2(hipMemcpy(dptr, prot_ptr, (size_t)(B + 1) * 4, hipMemcpyHostToDevice));
    pfk_load_coords(dev_prot_x, xn, Np, nullptr, nullptr, 0.f, s);
    const float r2 = h->cfg.cutoff_pp * h->cfg.cutoff_pp;
    pfk_pp_radius(xn, dptr, B, r2, max_nb, ddeg, nullptr, nullptr, nullptr, 0, s);
    PF_HIP2(hipStreamSynchronize(s));
    PF_HIP2(hipMemcpy(deg.data(), ddeg, (size_t)Np * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < Np; ++i) { off[i] = (int)total; total += deg[i]; }
    if (host_src && host_dst) {
        if (capacity < total) { cleanup(); h->err = "pf_build_pp_edges: capacity too small"; return PF_ERR_ARG; }
        if (total > 0) {
            PF_HIP2(hipMalloc((void**)&dsrc, (size_t)total * 4));
            PF_HIP2(hipMalloc((void**)&ddst, (size_t)total * 4));
            PF_HIP2(hipMemcpy(doff, off.data(), (size_t)Np * 4, hipMemcpyHostToDevice));
            pfk_pp_radius(xn, dptr, B, r2, max_nb, ddeg, doff, dsrc, ddst, 1, s);
            PF_HIP2(hipStreamSynchronize(s));
            PF_HIP2(hipMemcpy(host_src, dsrc, (size_t)total * 4, hipMemcpyDeviceToHost));
            PF_HIP2(hipMemcpy(host_dst, ddst, (size_t)total * 4, hipMemcpyDeviceToHost));
        }
    }
    (void)rc;
    cleanup();
    return total;
}

static int load_state(pf_handle* h, const float* dev_prot_x, const float* dev_pharm_x, const float* dev_pharm_h, hipStream_t s) {
    h->edges_built = false; h->rec_valid = false;
    if (dev_prot_x) { pfk_load_coords(dev_prot_x, h->d_xn, h->Np, h->d_gid, nullptr, 0.f, s); h->coords_custom = true; }
    if (dev_pharm_x) pfk_load_coords(dev_pharm_x, h->d_xn + h->Np, h->Nf, h->d_gid, nullptr, 0.f, s);
    if (dev_pharm_h) pfk_copy(dev_pharm_h, h->d_pharm_h, (size_t)h->Nf * h->cfg.pharm_nf, s);
    return PF_OK;
}

int pf_dynamics_forward(pf_handle* h, const float* dev_prot_x, const float* dev_pharm_x, const float* dev_pharm_h,
                        const float* dev_t, float* dev_eps_h, float* dev_eps_x, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!dev_pharm_x || !dev_pharm_h || !dev_t || !dev_eps_h || !dev_eps_x) PF_FAIL(h, PF_ERR_ARG, "pf_dynamics_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    load_state(h, dev_prot_x, dev_pharm_x, dev_pharm_h, s);
    pfk_copy(dev_t, h->d_t, (size_t)h->B, s);
    return run_dynamics(h, dev_eps_h, dev_eps_x, s);
}

int pf_sample_begin(pf_handle* h, const float* dev_init_pharm_com, const float* dev_noise0, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!dev_noise0) PF_FAIL(h, PF_ERR_ARG, "pf_sample_begin: null noise");
    {                                       // a caller that never asked pf_sample_status still learns of an invalid run here
        int32_t n = 0;
        rc = pf_sample_status(h, &n);
        if (rc) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    // every run starts from armed exchange words, whatever the previous run on the handle left in them (a timed-out row is not
    // re-armed by its consumer, pf_stepbuild.h); stream-ordered like everything else of the run
    if (h->d_xchg) PF_HIP(h, hipMemsetAsync(h->d_xchg, 0xff, (size_t)2 * std::max(h->Nf, 1) * PF_XCHG_STRIDE * sizeof(unsigned int), s));
    // init_prot_com = mean of the ORIGINAL protein coordinates (pharmacodiff.py:442)
    pfk_load_coords(h->d_prot_x0, h->d_xn, h->Np, h->d_gid, nullptr, 0.f, s);
    pfk_segment_mean(h->d_xn, h->d_prot_ptr, 0, h->B, h->d_com_init, s);
    const float* shift = dev_init_pharm_com ? dev_init_pharm_com : h->d_com_init;      // :448-452
    pfk_load_coords(h->d_prot_x0, h->d_xn, h->Np, h->d_gid, shift, -1.f, s);
    pfk_load_noise0(dev_noise0, h->d_xn + h->Np, h->d_pharm_h, h->Nf, h->cfg.pharm_nf, s);  // :455-456
    h->cen_valid = false; h->snap_cur = -1;
    h->spec_valid = false; h->step_id += 2;      // (a gap: no stamp of the run before reads as "the previous step")
    if (h->cen_hoist && h->d_snap[0] && h->l0c_off != 0) {      // center hoist: the features as they are, for the first step's hoist workgroups
        pfk_copy(h->d_pharm_h, h->d_snap[0], (size_t)h->Nf * h->cfg.pharm_nf, s);
        h->snap_cur = 0;
    }
    h->coords_custom = false;               // a rigid translate of the batch's own coordinates from here on
    h->sampling = true;
    h->edges_built = false; h->rec_valid = false;
    return PF_OK;
}

int pf_denoise_step(pf_handle* h, const pf_step_coef* coef, const float* dev_noise, int32_t ep_coord, int32_t ep_feat,
                    pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!coef || !dev_noise) PF_FAIL(h, PF_ERR_ARG, "pf_denoise_step: null argument");
    if (!h->sampling) PF_FAIL(h, PF_ERR_STATE, "pf_denoise_step before pf_sample_begin");
    hipStream_t s = (hipStream_t)stream;
    StepParams sp{};
    sp.B = h->B; sp.Np_tot = h->Np; sp.prot_ptr = h->d_prot_ptr; sp.pharm_ptr = h->d_pharm_ptr;
    sp.xn = h->d_xn; sp.pharm_h = h->d_pharm_h; sp.eps_h = h->d_eps_h; sp.eps_x = h->d_eps_x; sp.noise = dev_noise;
    sp.nf = h->cfg.pharm_nf;
    sp.a_ts = coef->alpha_t_given_s; sp.var = coef->var_terms; sp.sigma = coef->sigma;
    sp.ep_zt = coef->ep_zt; sp.ep_pred = coef->ep_pred; sp.ep_coord = ep_coord; sp.ep_feat = ep_feat;
    // center hoist: the updated features also go to the snapshot the NEXT step's hoist workgroups read (they must not race with that
    // step's update of pharm_h); only the paths through pf_stepbuild.h write it
    ++h->step_id;
    const int snap_next = h->snap_cur < 0 ? 0 : (h->snap_cur ^ 1);
    sp.h_snap_out = (h->cen_hoist && h->d_snap[0] && h->l0c_off != 0) ? h->d_snap[snap_next] : nullptr;
    rc = run_dynamics(h, h->d_eps_h, h->d_eps_x, s, &coef->t, false, &sp);      // every graph of the batch is at the same t
    if (rc) return rc;
    h->snap_cur = (sp.h_snap_out && h->tail_done && h->last_tail == 2) ? snap_next : -1;
    if (h->snap_cur < 0) h->cen_valid = false;
    if (h->tail_done) return PF_OK;         // the tail launch did the update and built the next call's edges
    if (encoders_on_the_fly(h)) {           // update + the edges of the next dynamics call in one launch
        const pf_config& cc = h->cfg;
        const bool share = (h->prune && cc.n_convs == 2) && share_now(h);     // what the next denoising step's dynamics call will ask for
        const BuildParams bp = build_params(h, share);
        // kNN pf edges and pockets of at most 512 atoms: the latency-optimised kernel (one atom per thread)
        const int fast = (h->cfg.pf_k > 0 && h->max_np <= 512 && h->cfg.pharm_nf <= 16 && h->step_build_fast) ? 1 : 0;
        { ProfScope ps(h, pf_handle::K_STEP, s); pfk_step_build(&sp, &bp, fast, s); }
        build_done(h, share);
        h->edges_built = true;
    } else { ProfScope ps(h, pf_handle::K_STEP, s); pfk_step_update(&sp, s); }
    return PF_OK;
}

int pf_prepare_timesteps(pf_handle* h, const float* host_t, int32_t n, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (n < 0 || (n && !host_t)) PF_FAIL(h, PF_ERR_ARG, "pf_prepare_timesteps: bad argument");
    h->t_plan.assign(host_t, host_t + n); h->plan_pos = 0;   // (a denoising step finds the NEXT call's timestep here: center hoist)
    if (n > L0_PTAB_SLOTS - 64) n = L0_PTAB_SLOTS - 64;      // the rest are computed when their steps arrive
    if (l0_hoist_ok(h)) l0_prepare_t(h, host_t, n, (hipStream_t)stream);
    return PF_OK;
}

int pf_sample_frame(pf_handle* h, float feat_norm_constant, float* dev_x, float* dev_h, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!h->sampling) PF_FAIL(h, PF_ERR_STATE, "no sampling run in progress");
    hipStream_t s = (hipStream_t)stream;
    pfk_segment_mean(h->d_xn, h->d_prot_ptr, 0, h->B, h->d_com_tmp, s);
    if (dev_x) pfk_export_coords(h->d_xn, h->Np, h->Nf, h->d_gid, h->d_com_init, h->d_com_tmp, dev_x, s);
    if (dev_h) pfk_scale_copy(h->d_pharm_h, dev_h, (size_t)h->Nf * h->cfg.pharm_nf, feat_norm_constant, s);
    return PF_OK;
}

int pf_sample_end(pf_handle* h, float feat_norm_constant, float* dev_x0, float* dev_h0, pf_stream stream) {
    // x_0 = x_t - protein COM + initial protein COM ; h_0 = h_t * norm constant  (pharmacodiff.py:480-488)
    int rc = pf_sample_frame(h, feat_norm_constant, dev_x0, dev_h0, stream);
    // the exchange's cumulative time-out count travels with the results: once the caller has waited for x_0 / h_0 it is on
    // the host too, and pf_sample_status judges THIS run without touching the device
    if (rc == PF_OK && h->d_xstat)
        PF_HIP(h, hipMemcpyAsync(h->xstat_host, h->d_xstat, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return rc;
}

int pf_sample_status(pf_handle* h, int32_t* n_timeouts) {
    if (!h) return PF_ERR_ARG;
    if (n_timeouts) *n_timeouts = 0;
    if (!h->xstat_host) return PF_OK;
    const int seen = *(volatile int*)h->xstat_host;
    if (seen == h->xstat_ack) return PF_OK;
    const int n = seen - h->xstat_ack;
    h->xstat_ack = seen;
    if (n_timeouts) *n_timeouts = n;
    h->pol.hs_build = 0;                    // this handle goes on with the separate launches
    PF_FAIL(h, PF_ERR_EXCHANGE, "%d time-out(s) in the exchange of the merged last launch (k_rg_node_hs_build): the sampling run(s) that "
                                "ended since the last status call are invalid and must be repeated; the handle now uses the separate "
                                "node + head and update + build launches", n);
}

int pf_debug_xchg_fault(pf_handle* h, int32_t drop_word, int32_t poll_max) {
    if (!h) return PF_ERR_ARG;
    h->xchg_fault = drop_word ? 1 : 0;
    h->xchg_poll_max = poll_max > 0 ? poll_max : 0;
    return PF_OK;
}

int pf_sample(pf_handle* h, int32_t n_steps, const pf_step_coef* host_coef, const float* dev_noise,
              const float* dev_init_pharm_com, int32_t ep_coord, int32_t ep_feat, float feat_norm_constant,
              float* dev_x0, float* dev_h0, float* dev_traj_x, float* dev_traj_h, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (n_steps < 0 || (n_steps && !host_coef) || !dev_noise) PF_FAIL(h, PF_ERR_ARG, "pf_sample: bad argument");
    const size_t row = (size_t)h->Nf * (3 + h->cfg.pharm_nf);
    const size_t fx = (size_t)h->Nf * 3, fh = (size_t)h->Nf * h->cfg.pharm_nf;
    rc = pf_sample_begin(h, dev_init_pharm_com, dev_noise, stream);
    if (rc) return rc;
    if (dev_traj_x || dev_traj_h) {
        rc = pf_sample_frame(h, feat_norm_constant, dev_traj_x, dev_traj_h, stream);
        if (rc) return rc;
    }
    {
        std::vector<float> tv(n_steps);
        for (int i = 0; i < n_steps; ++i) tv[i] = host_coef[i].t;
        rc = pf_prepare_timesteps(h, tv.data(), n_steps, stream);
        if (rc) return rc;
    }
    for (int i = 0; i < n_steps; ++i) {
        rc = pf_denoise_step(h, host_coef + i, dev_noise + (size_t)(i + 1) * row, ep_coord, ep_feat, stream);
        if (rc) return rc;
        if (dev_traj_x || dev_traj_h) {
            rc = pf_sample_frame(h, feat_norm_constant, dev_traj_x ? dev_traj_x + (size_t)(i + 1) * fx : nullptr,
                                 dev_traj_h ? dev_traj_h + (size_t)(i + 1) * fh : nullptr, stream);
            if (rc) return rc;
        }
    }
    return pf_sample_end(h, feat_norm_constant, dev_x0, dev_h0, stream);
}

int64_t pf_debug_get_edges(pf_handle* h, int32_t etype, int32_t* host_src, int32_t* host_dst, int64_t capacity,
                           pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (etype < 0 || etype > 3) PF_FAIL(h, PF_ERR_ARG, "bad etype");
    hipStream_t s = (hipStream_t)stream;
    PF_HIP(h, hipStreamSynchronize(s));
    const int B = h->B, Np = h->Np;
    std::vector<int> cnt((size_t)3 * B);
    PF_HIP(h, hipMemcpy(cnt.data(), h->d_dyn_cnt, (size_t)3 * B * 4, hipMemcpyDeviceToHost));
    int64_t total = 0;
    if (etype == ET_PP) total = h->Epp;
    else for (int g = 0; g < B; ++g) total += cnt[(size_t)etype * B + g];
    if (!host_src || !host_dst) return total;
    if (capacity < total) PF_FAIL(h, PF_ERR_ARG, "capacity too small");
    std::vector<int> es(std::max<int64_t>(h->Ecap, 1)), ed(std::max<int64_t>(h->Ecap, 1));
    PF_HIP(h, hipMemcpy(es.data(), h->d_esrc, (size_t)h->Ecap * 4, hipMemcpyDeviceToHost));
    PF_HIP(h, hipMemcpy(ed.data(), h->d_edst, (size_t)h->Ecap * 4, hipMemcpyDeviceToHost));
    const bool src_pharm = (etype == ET_FF || etype == ET_FP), dst_pharm = (etype == ET_FF || etype == ET_PF);
    int64_t o = 0;
    auto emit = [&](int64_t a, int64_t n) {
        for (int64_t e = a; e < a + n; ++e, ++o) {
            host_src[o] = es[e] - (src_pharm ? Np : 0);
            host_dst[o] = ed[e] - (dst_pharm ? Np : 0);
        }
    };
    if (etype == ET_PP) emit(0, h->Epp);
    else for (int g = 0; g < B; ++g) emit(h->h_reg[(size_t)etype * B + g], cnt[(size_t)etype * B + g]);
    return total;
}

int pf_debug_conv_layer(pf_handle* h, int32_t layer, const float* dev_prot_x, const float* dev_pharm_x,
                        const float* hp_, const float* vp_, const float* hf_, const float* vf_,
                        float* ohp, float* ovp, float* ohf, float* ovf, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    const pf_config& c = h->cfg;
    if (layer < 0 || layer >= c.n_convs) PF_FAIL(h, PF_ERR_ARG, "bad layer");
    hipStream_t s = (hipStream_t)stream;
    load_state(h, dev_prot_x, dev_pharm_x, nullptr, s);
    const size_t Np = h->Np, Nf = h->Nf;
    pfk_copy(hp_, h->d_h[0], Np * PF_S, s);
    pfk_copy(hf_, h->d_h[0] + Np * PF_S, Nf * PF_S, s);
    pfk_copy(vp_, h->d_v[0], Np * 48, s);
    pfk_copy(vf_, h->d_v[0] + Np * 48, Nf * 48, s);
    BuildParams bp{};
    bp.B = h->B; bp.Np_tot = h->Np; bp.prot_ptr = h->d_prot_ptr; bp.pharm_ptr = h->d_pharm_ptr; bp.xn = h->d_xn;
    bp.reg = h->d_reg; bp.dyn_cnt = h->d_dyn_cnt; bp.esrc = h->d_esrc; bp.edst = h->d_edst;
    bp.in_start = h->d_in_start; bp.in_cnt = h->d_in_cnt; bp.N = h->N; bp.ff_k = c.ff_k; bp.pf_k = c.pf_k;
    bp.r2_ff = c.cutoff_ff * c.cutoff_ff; bp.r2_pf = c.cutoff_pf * c.cutoff_pf;
    bp.gnorm = h->d_gnorm; bp.pp_cnt = h->d_pp_cnt; bp.pfq_cnt = h->d_pfq_cnt; bp.norm_mode = c.message_norm_mode;
    bp.act_ids = nullptr; bp.reg_act = h->d_reg_act;
    pfk_build_edges(&bp, s);
    EdgeParams e{};
    e.tiles = h->d_edge_tiles; e.ntiles = h->n_edge_tiles; e.dyn_cnt = h->d_dyn_cnt; e.esrc = h->d_esrc; e.edst = h->d_edst;
    e.xn = h->d_xn; e.h = h->d_h[0]; e.v = h->d_v[0]; e.msg_s = h->d_msg_s; e.msg_v = h->d_msg_v;
    e.w = h->d_gvp + h->msg_base(layer, 0); e.n_gvps = c.n_message_gvps;
    linspace_f32(0.f, c.rbf_dmax, c.rbf_dim, e.rbf_mu);
    e.rbf_inv_sigma = 1.0f / (c.rbf_dmax / (float)c.rbf_dim);
    for (int et = 0; et < 4; ++et) e.rg[et] = h->d_w + h->rg_msg[(size_t)layer * 4 + et];
    const int rg = h->pol.rg_mode(e.ntiles);                  // same choice as run_dynamics
    if (rg) pfk_rg_edge(&e, nullptr, 0, rg, 0, 0, s);
    else if (e.ntiles <= h->pol.coop_edge_max) pfk_edge_msg_coop(&e, 0, s); else pfk_edge_msg(&e, 0, s);
    NodeParams n{};
    n.tiles = h->d_node_tiles; n.ntiles = h->n_node_tiles; n.in_start = h->d_in_start; n.in_cnt = h->d_in_cnt; n.N = h->N;
    n.pp_slot = 1; n.row_ids = h->d_act_ids; n.dyn_cnt = h->d_dyn_cnt;
    n.msg_s = h->d_msg_s; n.msg_v = h->d_msg_v; n.zero_row = h->zero_row; n.h_in = h->d_h[0]; n.v_in = h->d_v[0]; n.h_out = h->d_h[1]; n.v_out = h->d_v[1];
    n.gid = h->d_gid; n.gnorm = h->d_gnorm; n.B = h->B; n.norm_mode = c.message_norm_mode; n.norm_value = c.message_norm_value;
    for (int nt = 0; nt < 2; ++nt) {
        const size_t* lo = &h->ln_off[(size_t)(layer * 2 + nt) * 4];
        n.w[nt].ln1_w = h->d_w + lo[0]; n.w[nt].ln1_b = h->d_w + lo[1]; n.w[nt].ln2_w = h->d_w + lo[2]; n.w[nt].ln2_b = h->d_w + lo[3];
        n.w[nt].upd = h->d_gvp + h->upd_base(layer, nt);
    }
    n.n_upd = c.n_update_gvps;
    n.grp = rg ? 4 * rg : 32;
    n.grp_pa = n.grp;
    for (int nt = 0; nt < 2; ++nt) n.rg_upd[nt] = h->d_w + h->rg_upd[(size_t)layer * 2 + nt];
    if (rg) pfk_rg_node(&n, nullptr, nullptr, 0, std::max(1, h->pol.rg_mode(n.ntiles)), 0, s);
    else if (n.ntiles <= h->pol.coop_node_max) pfk_node_update_coop(&n, 0, s); else pfk_node_update(&n, 0, s);
    pfk_copy(h->d_h[1], ohp, Np * PF_S, s);
    pfk_copy(h->d_h[1] + Np * PF_S, ohf, Nf * PF_S, s);
    pfk_copy(h->d_v[1], ovp, Np * 48, s);
    pfk_copy(h->d_v[1] + Np * 48, ovf, Nf * 48, s);
    // the zero-vector invariant of buffer 0 (layer-0 kernels never read it, later layers overwrite it)
    PF_HIP(h, hipMemsetAsync(h->d_v[0], 0, (size_t)h->N * 48 * 4, s));
    hipError_t er = hipGetLastError();
    if (er != hipSuccess) PF_FAIL(h, PF_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(er));
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------
// gradient path (training step): forward that keeps the per-layer state, backward, parameter layout
// ------------------------------------------------------------------------------------------------
static int ensure_train_ws(pf_handle* h, hipStream_t s) {
    if (h->d_tws && h->t_ws_ready) return PF_OK;
    const pf_config& c = h->cfg;
    if (c.n_message_gvps > PFT_MAX_CHAIN || c.n_noise_gvps > PFT_MAX_CHAIN || c.n_update_gvps > 3)
        PF_FAIL(h, PF_ERR_ARG, "training supports at most %d message / noise GVPs and 3 update GVPs per chain", PFT_MAX_CHAIN);
    if (c.pharm_nf > 8 || c.rec_nf + 1 > 17 || c.pharm_nf + 1 > 17)
        PF_FAIL(h, PF_ERR_ARG, "training supports pharm_nf <= 8 and rec_nf <= 16");
    const int L = c.n_convs, N = h->N;
    const size_t E1 = (size_t)h->Ecap + 1;
    h->t_nblk = std::max(8, std::min(256, h->n_edge_tiles));
    size_t bytes = 0;
    auto need = [&](size_t n_floats) { bytes += (n_floats * 4 + 255) & ~size_t(255); };
    for (int l = 0; l <= L; ++l) { need((size_t)N * PF_S); need((size_t)N * 48); }
    for (int l = 0; l < L; ++l) { need(E1 * PF_S); need(E1 * 48); }
    for (int a = 0; a < 2; ++a) { need((size_t)N * PF_S); need((size_t)N * 48); }
    need((size_t)N * PF_S); need((size_t)N * 48);
    need(64);
    need(64); need((size_t)std::max(h->n_edge_tiles, h->n_edge_tiles_act) + 64);      // compact tile list and its counts
    need((size_t)PFT_ENC_BLOCKS * std::max(h->enc_n, 1));
    need((size_t)2 * std::max(h->n_node_tiles, h->n_node_tiles_act) + 64);
    need((size_t)h->B * c.rec_nf * PF_S);
    need((size_t)h->Nf * 3); need((size_t)h->B); need((size_t)h->B); need((size_t)h->Nf * 3); need((size_t)h->Nf * c.pharm_nf); need(64);   // loss buffers
    for (int l = 0; l < L; ++l) { need((size_t)c.n_update_gvps * 2 * N * PF_S); need((size_t)c.n_update_gvps * 2 * N * 16); need((size_t)c.n_update_gvps * 2 * N * 48); }
    need((size_t)c.n_noise_gvps * std::max(h->Nf, 1) * PF_S); need((size_t)c.n_noise_gvps * std::max(h->Nf, 1) * 16);
    need((size_t)c.n_noise_gvps * std::max(h->Nf, 1) * 48);
    need((size_t)h->t_nblk * ((h->nparams + 63) / 64 * 64));
    const size_t Es = (size_t)std::max<int64_t>(h->Ecap, 1), ng = (size_t)c.n_message_gvps;
    for (int l = 0; l < L; ++l) { need(ng * Es * PF_S); need(ng * Es * 16); need(ng * Es * 48); }
    need(Es * PF_S); need(Es * 48);
    // like d_ws the allocation outlives the batch: a training loop binds a new batch every step, and a hipFree / hipMalloc
    // pair of a few GB (plus clearing it) per step cost two orders of magnitude more than the step itself
    bool fresh = false;
    if (h->tws_capacity < bytes + 4096) {
        PF_HIP(h, hipDeviceSynchronize());
        if (h->d_tws) { (void)hipFree(h->d_tws); h->d_tws = nullptr; h->tws_capacity = 0; }
        const size_t want = bytes + bytes / 8 + 4096;
        PF_HIP(h, hipMalloc(&h->d_tws, want));
        h->tws_capacity = want;
        fresh = true;
    }
    char* cur = reinterpret_cast<char*>(h->d_tws);
    h->t_H.assign(L + 1, nullptr); h->t_V.assign(L + 1, nullptr); h->t_msg_s.assign(L, nullptr); h->t_msg_v.assign(L, nullptr);
    for (int l = 0; l <= L; ++l) { h->t_H[l] = carve<float>(cur, (size_t)N * PF_S); h->t_V[l] = carve<float>(cur, (size_t)N * 48); }
    for (int l = 0; l < L; ++l) { h->t_msg_s[l] = carve<float>(cur, E1 * PF_S); h->t_msg_v[l] = carve<float>(cur, E1 * 48); }
    for (int a = 0; a < 2; ++a) { h->t_G_h[a] = carve<float>(cur, (size_t)N * PF_S); h->t_G_v[a] = carve<float>(cur, (size_t)N * 48); }
    h->t_gagg_s = carve<float>(cur, (size_t)N * PF_S); h->t_gagg_v = carve<float>(cur, (size_t)N * 48);
    {
        // int64 accumulators [N][128] and [N][48]: cleared when allocated, pfk_fix_apply leaves every element it read at zero
        const size_t a_bytes = ((size_t)N * PF_S * 8 + 255) / 256 * 256, need_a = a_bytes + (size_t)N * 48 * 8;
        if (h->tA_capacity < need_a) {
            PF_HIP(h, hipDeviceSynchronize());
            if (h->d_tA) { (void)hipFree(h->d_tA); h->d_tA = nullptr; h->tA_capacity = 0; }
            const size_t want = need_a + need_a / 8;
            PF_HIP(h, hipMalloc(&h->d_tA, want));
            h->tA_capacity = want;
            h->tA_dirty = true;
        }
        if (h->tA_dirty) { PF_HIP(h, hipMemsetAsync(h->d_tA, 0, h->tA_capacity, s)); h->tA_dirty = false; }
        h->t_A_h = reinterpret_cast<long long*>(h->d_tA);
        h->t_A_v = reinterpret_cast<long long*>(reinterpret_cast<char*>(h->d_tA) + a_bytes);
    }
    h->t_fix = carve<float>(cur, 64);
    h->t_ccnt = carve<int>(cur, 128);            // [layer][16]: passes per etype, rows per etype at + 8; node units at [96]
    h->t_clist_cap = (size_t)std::max(h->n_edge_tiles, h->n_edge_tiles_act) * 32 + 64;
    h->t_clist = carve<int>(cur, h->t_clist_cap * L);                          // dense row lists, one per conv layer
    h->t_gpart_enc = carve<float>(cur, (size_t)PFT_ENC_BLOCKS * std::max(h->enc_n, 1));
    h->t_ucap = 32 * std::max(h->n_node_tiles, h->n_node_tiles_act) + 16;      // rows per node type in the dense unit list
    h->t_ulist_cap = (size_t)2 * 2 * h->t_ucap + 64;
    h->t_ulist = carve<int>(cur, h->t_ulist_cap * L);                          // per conv layer: [2 types][t_ucap] x (node id, saved-level row)
    h->t_Gg = carve<float>(cur, (size_t)h->B * c.rec_nf * PF_S);
    h->t_lx0c = carve<float>(cur, (size_t)h->Nf * 3); h->t_lag = carve<float>(cur, (size_t)h->B); h->t_lsg = carve<float>(cur, (size_t)h->B);
    h->t_lgx = carve<float>(cur, (size_t)h->Nf * 3); h->t_lgh = carve<float>(cur, (size_t)h->Nf * c.pharm_nf); h->t_lout = carve<float>(cur, 64);
    h->t_nsv_z.assign(L, nullptr); h->t_nsv_g.assign(L, nullptr); h->t_nsv_v.assign(L, nullptr);
    for (int l = 0; l < L; ++l) {
        h->t_nsv_z[l] = carve<float>(cur, (size_t)c.n_update_gvps * 2 * N * PF_S);
        h->t_nsv_g[l] = carve<float>(cur, (size_t)c.n_update_gvps * 2 * N * 16);
        h->t_nsv_v[l] = carve<float>(cur, (size_t)c.n_update_gvps * 2 * N * 48);
    }
    h->t_hsv_z = carve<float>(cur, (size_t)c.n_noise_gvps * std::max(h->Nf, 1) * PF_S);
    h->t_hsv_g = carve<float>(cur, (size_t)c.n_noise_gvps * std::max(h->Nf, 1) * 16);
    h->t_hsv_v = carve<float>(cur, (size_t)c.n_noise_gvps * std::max(h->Nf, 1) * 48);
    h->t_gpart = carve<float>(cur, (size_t)h->t_nblk * ((h->nparams + 63) / 64 * 64));
    h->t_sv_z.assign(L, nullptr); h->t_sv_g.assign(L, nullptr); h->t_sv_v.assign(L, nullptr);
    for (int l = 0; l < L; ++l) {
        h->t_sv_z[l] = carve<float>(cur, ng * Es * PF_S); h->t_sv_g[l] = carve<float>(cur, ng * Es * 16);
        h->t_sv_v[l] = carve<float>(cur, ng * Es * 48);
    }
    h->t_gs_buf = carve<float>(cur, Es * PF_S); h->t_gv_buf = carve<float>(cur, Es * 48);
    // message buffers: the zero row (index Ecap) must read as zeros; V[0] is the all-zero initial vector state
    // (only rows written by the same forward and the zero row are ever read: a reused allocation needs just that row)
    {
        ZeroBatch zb(s);
        for (int l = 0; l < L; ++l) {
            if (fresh) {
                zb.add(h->t_msg_s[l], E1 * PF_S * 4);
                zb.add(h->t_msg_v[l], E1 * 48 * 4);
            } else {
                zb.add(h->t_msg_s[l] + (E1 - 1) * PF_S, PF_S * 4);
                zb.add(h->t_msg_v[l] + (E1 - 1) * 48, 48 * 4);
            }
        }
        zb.add(h->t_V[0], (size_t)N * 48 * 4);
        zb.flush();
    }
    h->t_ws_ready = true;
    return PF_OK;
}

int pf_param_count(pf_handle* h, int64_t* n_params, int32_t* n_tensors) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (n_params) *n_params = (int64_t)h->nparams;
    if (n_tensors) *n_tensors = (int32_t)h->flat_layout.size();
    return PF_OK;
}

int pf_param_layout(pf_handle* h, int32_t index, const char** name, int64_t* offset, int64_t* numel) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (index < 0 || index >= (int32_t)h->flat_layout.size()) PF_FAIL(h, PF_ERR_ARG, "pf_param_layout: index out of range");
    const auto& kv = h->flat_layout[index];
    if (name) *name = kv.first.c_str();
    if (offset) *offset = (int64_t)kv.second.first;
    if (numel) *numel = (int64_t)kv.second.second;
    return PF_OK;
}

// copied: h->d_flat already holds the values (pf_adam_step's kernel wrote them)
static int set_flat_params_impl(pf_handle* h, const float* dev_flat, pf_stream stream, bool copied) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!dev_flat) PF_FAIL(h, PF_ERR_ARG, "pf_set_flat_params: null argument");
    if (!h->d_map) PF_FAIL(h, PF_ERR_STATE, "pf_set_flat_params: no gather map (more than 2^24 parameters)");
    hipStream_t s = (hipStream_t)stream;
    if (!copied) PF_HIP(h, hipMemcpyAsync(h->d_flat, dev_flat, h->nparams * sizeof(float), hipMemcpyDeviceToDevice, s));
    const size_t n_now = (h->n16_begin > 0 && h->n16_begin < h->n_packed) ? h->n16_begin : h->n_packed;
    pfk_gather_weights(h->d_flat, h->d_map, n_now, h->d_w, s);
    h->n16_stale = n_now < h->n_packed;
    ++h->w_version;
    h->t_have_fwd = false;
    return PF_OK;
}

int pf_set_flat_params(pf_handle* h, const float* dev_flat, pf_stream stream) { return set_flat_params_impl(h, dev_flat, stream, false); }

int pf_adam_step(pf_handle* h, float* dev_params, const float* dev_grad, float* dev_exp_avg, float* dev_exp_avg_sq,
                 int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay, pf_stream stream) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!dev_params || !dev_grad || !dev_exp_avg || !dev_exp_avg_sq || step < 1) PF_FAIL(h, PF_ERR_ARG, "pf_adam_step: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const bool mirror = h->d_map != nullptr && dev_params != h->d_flat;
    pfk_adam(dev_params, dev_grad, dev_exp_avg, dev_exp_avg_sq, h->nparams, lr, beta1, beta2, eps, weight_decay, (float)bc1,
             (float)std::sqrt(bc2), mirror ? h->d_flat : nullptr, s);
    return set_flat_params_impl(h, dev_params, stream, mirror);
}

int pf_get_flat_params(pf_handle* h, float* dev_flat, pf_stream stream) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!dev_flat) PF_FAIL(h, PF_ERR_ARG, "pf_get_flat_params: null argument");
    PF_HIP(h, hipMemcpyAsync(dev_flat, h->d_flat, h->nparams * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PF_OK;
}

int pf_train_forward(pf_handle* h, const float* dev_prot_x, const float* dev_pharm_x, const float* dev_pharm_h,
                     const float* dev_t, float dropout_p, uint32_t seed, float* dev_eps_h, float* dev_eps_x, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!dev_pharm_x || !dev_pharm_h || !dev_t || !dev_eps_h || !dev_eps_x) PF_FAIL(h, PF_ERR_ARG, "pf_train_forward: null argument");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) PF_FAIL(h, PF_ERR_ARG, "pf_train_forward: dropout must be in [0, 1)");
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_train_ws(h, s);
    if (rc) return rc;
    h->t_common = TrainCommon{};
    h->t_common.W = h->d_flat; h->t_common.gpart = h->t_gpart; h->t_common.nparams = (int)h->nparams;
    h->t_common.gstride = (int)((h->nparams + 63) / 64 * 64);
    h->t_common.tseg = h->d_tseg; h->t_common.ntens = h->n_tseg;
    h->t_common.gpart_enc = h->t_gpart_enc; h->t_common.enc_begin = h->enc_begin; h->t_common.enc_n = h->enc_n;
    h->t_common.drop_thr = dropout_p > 0.f ? (uint32_t)std::min(4294967295.0, (double)dropout_p * 4294967296.0) : 0u;
    h->t_common.drop_scale = 1.0f / (1.0f - dropout_p);
    h->t_common.seed = seed;
    h->t_common.mask_override = h->t_mask_override; h->t_common.mask_N = h->N;
    h->t_common.bf16 = h->train_bf16 ? 1 : 0;
    h->t_have_loss = false;
    load_state(h, dev_prot_x, dev_pharm_x, dev_pharm_h, s);
    pfk_copy(dev_t, h->d_t, (size_t)h->B, s);
    rc = run_dynamics(h, dev_eps_h, dev_eps_x, s, nullptr, true);
    h->t_have_fwd = rc == PF_OK;
    return rc;
}

int pf_train_loss_forward(pf_handle* h, const float* dev_pharm_x0, const float* dev_pharm_h0, const int32_t* dev_t_int,
                          const float* dev_eps_x, const float* dev_eps_h, const float* dev_alpha, const float* dev_sigma,
                          int32_t n_timesteps, float feat_norm, int32_t remove_com, int32_t weighted_loss, float dropout_p,
                          uint32_t seed, float* dev_out, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!dev_pharm_x0 || !dev_pharm_h0 || !dev_t_int || !dev_eps_x || !dev_eps_h || !dev_alpha || !dev_sigma || !dev_out)
        PF_FAIL(h, PF_ERR_ARG, "pf_train_loss_forward: null argument");
    if (n_timesteps < 1 || !(feat_norm > 0.f)) PF_FAIL(h, PF_ERR_ARG, "pf_train_loss_forward: bad n_timesteps / feat_norm");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) PF_FAIL(h, PF_ERR_ARG, "pf_train_loss_forward: dropout must be in [0, 1)");
    if (h->Nf == 0) PF_FAIL(h, PF_ERR_ARG, "pf_train_loss_forward: the batch has no pharmacophore centers (the losses are means over them)");
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_train_ws(h, s);
    if (rc) return rc;
    h->t_have_loss = false;
    h->t_common = TrainCommon{};
    h->t_common.W = h->d_flat; h->t_common.gpart = h->t_gpart; h->t_common.nparams = (int)h->nparams;
    h->t_common.gstride = (int)((h->nparams + 63) / 64 * 64);
    h->t_common.tseg = h->d_tseg; h->t_common.ntens = h->n_tseg;
    h->t_common.gpart_enc = h->t_gpart_enc; h->t_common.enc_begin = h->enc_begin; h->t_common.enc_n = h->enc_n;
    h->t_common.drop_thr = dropout_p > 0.f ? (uint32_t)std::min(4294967295.0, (double)dropout_p * 4294967296.0) : 0u;
    h->t_common.drop_scale = 1.0f / (1.0f - dropout_p);
    h->t_common.seed = seed;
    h->t_common.mask_override = h->t_mask_override; h->t_common.mask_N = h->N;
    h->t_common.bf16 = h->train_bf16 ? 1 : 0;
    LossParams lp{};
    lp.part = h->d_lpart + 16; lp.ticket = reinterpret_cast<int*>(h->d_lpart);
    lp.B = h->B; lp.Np = h->Np; lp.Nf = h->Nf; lp.nf = h->cfg.pharm_nf; lp.T = n_timesteps; lp.remove_com = remove_com; lp.weighted = weighted_loss;
    lp.feat_norm = feat_norm;
    lp.prot_ptr = h->d_prot_ptr; lp.pharm_ptr = h->d_pharm_ptr; lp.gid = h->d_gid; lp.prot_x0 = h->d_prot_x0;
    lp.x0 = dev_pharm_x0; lp.h0 = dev_pharm_h0; lp.t_int = dev_t_int; lp.eps_x = dev_eps_x; lp.eps_h = dev_eps_h;
    lp.alpha_tab = dev_alpha; lp.sigma_tab = dev_sigma;
    lp.xn = h->d_xn; lp.pharm_h = h->d_pharm_h; lp.t = h->d_t;
    lp.x0c = h->t_lx0c; lp.alpha_g = h->t_lag; lp.sigma_g = h->t_lsg;
    lp.dyn_h = h->d_eps_h; lp.dyn_x = h->d_eps_x; lp.g_x = h->t_lgx; lp.g_h = h->t_lgh; lp.out = dev_out;
    h->edges_built = false; h->rec_valid = false;
    h->coords_custom = true;                     // the pocket moved with the centers' COM: trajectory constants of the bound coordinates do not apply
    pfk_loss_prepare(&lp, s);
    rc = run_dynamics(h, h->d_eps_h, h->d_eps_x, s, nullptr, true);
    h->t_have_fwd = rc == PF_OK;
    if (rc) return rc;
    pfk_loss_eval(&lp, s);
    h->t_have_loss = true;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PF_FAIL(h, PF_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return PF_OK;
}

static int loss_backward(pf_handle* h, const float* g_pos, const float* g_pos2, const float* g_feat, const float* g_feat2,
                         float* dev_grad, pf_stream stream, const char* who) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!h->t_have_fwd || !h->t_have_loss) PF_FAIL(h, PF_ERR_STATE, "%s: no pf_train_loss_forward on this batch", who);
    if (!g_pos || !g_feat || !dev_grad) PF_FAIL(h, PF_ERR_ARG, "%s: null argument", who);
    // the unit gradients times their upstream scalars: as extra blocks of the backward's first launch (the fragment re-pack) when it
    // has one, else as a launch of its own (pf_train_backward)
    h->pend_scale = ScaleArgs{h->t_lgx, h->Nf * 3, g_pos, g_pos2, h->t_lgh, h->Nf * h->cfg.pharm_nf, g_feat, g_feat2};
    h->has_pend_scale = true;
    h->t_have_loss = false;                      // the unit gradients are consumed
    rc = pf_train_backward(h, h->t_lgh, h->t_lgx, dev_grad, stream);
    h->has_pend_scale = false;
    return rc;
}

int pf_train_loss_backward(pf_handle* h, const float* dev_g_pos, const float* dev_g_feat, float* dev_grad, pf_stream stream) {
    return loss_backward(h, dev_g_pos, nullptr, dev_g_feat, nullptr, dev_grad, stream, "pf_train_loss_backward");
}

int pf_train_loss_backward_out(pf_handle* h, const float* dev_g_out, float* dev_grad, pf_stream stream) {
    if (h && !dev_g_out) PF_FAIL(h, PF_ERR_ARG, "pf_train_loss_backward_out: null argument");
    // upstream gradients of the nine outputs: [0] and [1] of the two losses, [6] of their sum; the metrics carry none
    return loss_backward(h, dev_g_out, dev_g_out ? dev_g_out + 6 : nullptr, dev_g_out ? dev_g_out + 1 : nullptr,
                         dev_g_out ? dev_g_out + 6 : nullptr, dev_grad, stream, "pf_train_loss_backward_out");
}

int pf_train_backward(pf_handle* h, const float* dev_g_eps_h, const float* dev_g_eps_x, float* dev_grad, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!h->t_have_fwd) PF_FAIL(h, PF_ERR_STATE, "pf_train_backward: no pf_train_forward on this batch");
    if (!dev_g_eps_h || !dev_g_eps_x || !dev_grad) PF_FAIL(h, PF_ERR_ARG, "pf_train_backward: null argument");
    if (h->n_tseg < 0) PF_FAIL(h, PF_ERR_ARG, "pf_train_backward: the gradient path supports n_convs <= 4");
    hipStream_t s = (hipStream_t)stream;
    const pf_config& c = h->cfg;
    const int L = c.n_convs, N = h->N, nb = h->t_nblk;
    const bool scale_now = h->has_pend_scale;
    h->has_pend_scale = false;
    if (h->wpack_version != h->w_version) {         // packed to_feats_out fragments of the message GVPs for k_bwd_edge_level
        const int ng = h->n_gvpt;
        if (!h->d_wpack) PF_HIP(h, hipMalloc((void**)&h->d_wpack, (size_t)2 * std::max(ng, 1) * PFT_WPACK_FLOATS * sizeof(float)));
        pfk_pack_gvp(h->d_flat, h->d_gvpt, ng, h->d_wpack, h->d_wpack + (size_t)ng * PFT_WPACK_FLOATS, scale_now ? &h->pend_scale : nullptr, s);
        h->wpack_version = h->w_version;
    } else if (scale_now) {
        const ScaleArgs& a = h->pend_scale;
        pfk_scale_loss(a.gx, a.nx, a.a, a.a2, a.gh, a.nh, a.b, a.b2, s);
    }
    h->t_common.wpack_b = h->d_wpack; h->t_common.wpack_f = h->d_wpack + (size_t)h->n_gvpt * PFT_WPACK_FLOATS;
    const TrainCommon tc = h->t_common;
    ReduceParams rp{};
    rp.gpart = h->t_gpart; rp.nparams = (int)h->nparams; rp.gstride = (int)((h->nparams + 63) / 64 * 64); rp.grad = dev_grad; rp.tseg = h->d_tseg; rp.ntens = h->n_tseg;
    rp.NB = nb; rp.ccnt = h->t_ccnt;
    rp.gpart_enc = h->t_gpart_enc; rp.enc_begin = h->enc_begin; rp.enc_n = h->enc_n;
    if (h->tA_dirty) PF_HIP(h, hipMemsetAsync(h->d_tA, 0, h->tA_capacity, s));      // an earlier pass stopped half way
    h->tA_dirty = true;
    // The dense work lists of every conv layer (k_compact_rows, k_compact_node_rows: one or two workgroups each, 10-20 us) depend
    // on the forward's edge counts only: they are built on the side stream while the head's backward runs here.
    bool first_clear_done = false;
    auto layer_tables = [&](int l, const NodeTile*& nt_tiles, int& nt_n, const EdgeTile*& e_tiles, const int*& et0, int& n_et) {
        const bool last = l == L - 1, pruned = h->prune && L >= 2 && l == L - 2;
        nt_tiles = pruned ? h->d_node_tiles_act : h->d_node_tiles;
        nt_n = last ? h->n_node_tiles_last : (pruned ? h->n_node_tiles_act : h->n_node_tiles);
        e_tiles = pruned ? h->d_edge_tiles_act : h->d_edge_tiles;
        et0 = pruned ? h->et_tile0_act : h->et_tile0;
        n_et = last ? 2 : 4;                         // the last layer's fp / pp messages reach no output
    };
    {
        for (int k = 0; k < 3; ++k)
            if (!h->cmp_ev[k]) PF_HIP(h, hipEventCreateWithFlags(&h->cmp_ev[k], hipEventDisableTiming));
        if (!h->s_side) PF_HIP(h, hipStreamCreateWithFlags(&h->s_side, hipStreamNonBlocking));
        hipStream_t side = h->s_side;
        if (side != s) { PF_HIP(h, hipEventRecord(h->cmp_ev[0], s)); PF_HIP(h, hipStreamWaitEvent(side, h->cmp_ev[0], 0)); }
        // two groups: what the first layer of the loop below needs at once (its work lists, its cleared input gradients: cmp_ev[1]),
        // then the rest (the fixed-point scale, first read by that layer's edge kernels; the other layers' lists: cmp_ev[2], waited
        // for behind that layer's node kernel) -- as one group the side stream outlasted the head's backward by 28 us once that took 59
        auto lists = [&](int l) {
            const NodeTile* ntt; const EdgeTile* ett; const int* et0; int ntn, n_et;
            layer_tables(l, ntt, ntn, ett, et0, n_et);
            pfk_compact_node_rows(ntt, ntn, h->d_dyn_cnt, h->d_act_ids, N, h->t_ulist + h->t_ulist_cap * l, h->t_ucap, h->t_ccnt + 96 + 4 * l, side);
            pfk_compact_rows(ett, et0, n_et, h->d_dyn_cnt, h->t_clist + h->t_clist_cap * l, h->t_ccnt + 16 * l, side);
        };
        lists(L - 1);
        if (side != s) {
            // the first layer of the loop below receives its input gradients in G[1]: cleared here, under the head's backward
            ZeroBatch zb(side);
            zb.add(h->t_G_h[1], (size_t)N * PF_S * 4);
            if (L - 1 != 0) zb.add(h->t_G_v[1], (size_t)N * 48 * 4);
            zb.flush();
            first_clear_done = true;
            PF_HIP(h, hipEventRecord(h->cmp_ev[1], side));
        }
        pfk_fix_scale(dev_g_eps_h, h->Nf * c.pharm_nf, dev_g_eps_x, h->Nf * 3, h->t_fix, side);      // (first read by the last layer's edge kernels)
        for (int l = L - 2; l >= 0; --l) lists(l);
        if (side != s) PF_HIP(h, hipEventRecord(h->cmp_ev[2], side));
    }
    // (the head kernel stores dL/d(last layer output) for every pharm row, and the last layer's node kernel reads those rows
    // only: no clearing of t_G_*[0] here)
    {
        BwdHeadParams p{};
        p.c = tc; p.tiles = h->d_head_tiles; p.ntiles = h->n_head_tiles; p.node_base = h->Np;
        p.h = h->t_H[L]; p.v = h->t_V[L];
        p.g = h->d_gvpt + h->head_base(); p.n_gvps = c.n_noise_gvps;
        p.o_Wout = (int)h->flat_offset("dynamics.noise_predictor.noise_predictor.to_scalar_output.weight");
        p.o_bout = (int)h->flat_offset("dynamics.noise_predictor.noise_predictor.to_scalar_output.bias");
        p.pharm_nf = c.pharm_nf; p.g_eps_h = dev_g_eps_h; p.g_eps_x = dev_g_eps_x;
        p.G_h = h->t_G_h[0]; p.G_v = h->t_G_v[0];
        if (h->t_head_saved) { p.sv_z = h->t_hsv_z; p.sv_g = h->t_hsv_g; p.sv_v = h->t_hsv_v; p.sv_stride = (size_t)h->Nf; }
        rp.head_grid = p.ntiles > 0 ? std::max(1, std::min(nb, 2 * p.ntiles)) : 0;
        { ProfScope ps(h, pf_handle::K_BWD_HEAD, s); pfk_bwd_head(&p, rp.head_grid, s); }
    }
    if (h->s_side != s) PF_HIP(h, hipStreamWaitEvent(s, h->cmp_ev[1], 0));
    int a = 0;
    unsigned early_mask = 0;
    // (the encoders' backward differentiates the protein rows grouped by (graph, element) when the features can be one-hots)
    const bool enc_grouped = c.rec_nf <= 16 && h->Np > 0;
    bool enc_grouped_done = false;
    for (int l = L - 1; l >= 0; --l) {
        // The last conv layer's output is read on the pharm nodes only (dynamics_gvp.py:91): its protein rows have a
        // zero gradient, so -- as in the forward -- only the pharm node tiles and the ff / pf edge tiles do any work
        // there.  The node kernel stores dL/d(layer input) for the rows it walks, the edge kernels add to it: clear first.
        const bool last = l == L - 1;
        // The layer before it is needed only where the last layer reads it: the pharm nodes and the protein atoms that
        // are the source of a pf edge (the active atoms); every other row of its output has a zero gradient.  Same
        // tile lists as the pruned forward.
        const bool pruned = h->prune && L >= 2 && l == L - 2;
        if (!(last && first_clear_done)) {
            ZeroBatch zb(s);
            zb.add(h->t_G_h[a ^ 1], (size_t)N * PF_S * 4);
            if (l != 0) zb.add(h->t_G_v[a ^ 1], (size_t)N * 48 * 4);      // (conv layer 0 has no vector input: nobody writes or reads that gradient)
            zb.flush();
        }
        BwdNodeParams n{};
        n.c = tc; n.tiles = pruned ? h->d_node_tiles_act : h->d_node_tiles;
        n.ntiles = last ? h->n_node_tiles_last : (pruned ? h->n_node_tiles_act : h->n_node_tiles);
        n.pp_slot = pruned ? 2 : 1; n.row_ids = h->d_act_ids; n.dyn_cnt = h->d_dyn_cnt;
        n.in_start = h->d_in_start; n.in_cnt = h->d_in_cnt; n.N = N;
        n.msg_s = h->t_msg_s[l]; n.msg_v = h->t_msg_v[l]; n.zero_row = h->zero_row;
        n.h_in = h->t_H[l]; n.v_in = h->t_V[l];
        n.G_h_out = h->t_G_h[a]; n.G_v_out = h->t_G_v[a]; n.G_h_in = h->t_G_h[a ^ 1]; n.G_v_in = h->t_G_v[a ^ 1];
        n.gagg_s = h->t_gagg_s; n.gagg_v = h->t_gagg_v;
        n.gid = h->d_gid; n.gnorm = h->d_gnorm; n.B = h->B;
        n.norm_mode = c.message_norm_mode; n.norm_value = c.message_norm_value;
        n.upd = h->d_gvpt + h->upd_base(l, 0); n.n_upd = c.n_update_gvps;
        for (int nt = 0; nt < 2; ++nt) {
            const std::string p1 = conv_prefix(l) + "message_layer_norms." + kNtKey[nt] + ".feat_norm.";
            const std::string p2 = conv_prefix(l) + "update_layer_norms." + kNtKey[nt] + ".feat_norm.";
            n.o_ln[nt][0] = (int)h->flat_offset(p1 + "weight"); n.o_ln[nt][1] = (int)h->flat_offset(p1 + "bias");
            n.o_ln[nt][2] = (int)h->flat_offset(p2 + "weight"); n.o_ln[nt][3] = (int)h->flat_offset(p2 + "bias");
        }
        n.layer = l; n.l0 = l == 0;
        n.grp = (int)h->t_grp.size() > l ? h->t_grp[l] : 32;
        if ((int)h->t_node_saved.size() > l && h->t_node_saved[l]) {
            n.sv_z = h->t_nsv_z[l]; n.sv_g = h->t_nsv_g[l]; n.sv_v = h->t_nsv_v[l]; n.sv_stride = (size_t)2 * h->N;
        }
        n.ulist = h->t_ulist + h->t_ulist_cap * l; n.ucnt = h->t_ccnt + 96 + 4 * l; n.ucap = h->t_ucap;
        rp.node_grid[l] = n.ntiles > 0 ? std::max(1, std::min(nb, 2 * n.ntiles)) : 0;
        { ProfScope ps(h, pf_handle::K_BWD_NODE, s); pfk_bwd_node(&n, rp.node_grid[l], s); }
        if (last && h->s_side != s) PF_HIP(h, hipStreamWaitEvent(s, h->cmp_ev[2], 0));      // the side stream's second group (above)
        BwdEdgeLevelParams e{};
        e.c = tc; e.tiles = pruned ? h->d_edge_tiles_act : h->d_edge_tiles; e.dyn_cnt = h->d_dyn_cnt;
        e.pp_slot = pruned ? 2 : 1;
        const int* et0 = pruned ? h->et_tile0_act : h->et_tile0;
        for (int et = 0; et <= 4; ++et) e.et_tile0[et] = et0[et];
        e.n_et = last ? 2 : 4;                       // the last layer's fp / pp messages reach no output
        e.clist = h->t_clist + h->t_clist_cap * l; e.ccnt = h->t_ccnt + 16 * l;
        rp.n_et[l] = e.n_et;
        e.esrc = h->d_esrc; e.edst = h->d_edst; e.xn = h->d_xn;
        e.h = h->t_H[l]; e.v = h->t_V[l];
        e.gagg_s = h->t_gagg_s; e.gagg_v = h->t_gagg_v; e.in_cnt = h->d_in_cnt; e.N = N;
        e.norm_mode = c.message_norm_mode;
        e.G_h_in = h->t_G_h[a ^ 1]; e.G_v_in = h->t_G_v[a ^ 1];
        e.sv_z = h->t_sv_z[l]; e.sv_g = h->t_sv_g[l]; e.sv_v = h->t_sv_v[l];
        e.sv_stride = (size_t)std::max<int64_t>(h->Ecap, 1);
        e.gs_buf = h->t_gs_buf; e.gv_buf = h->t_gv_buf;
        e.g = h->d_gvpt + h->msg_base(l, 0); e.n_gvps = c.n_message_gvps;
        linspace_f32(0.f, c.rbf_dmax, c.rbf_dim, e.rbf_mu);
        e.rbf_inv_sigma = 1.0f / ((c.rbf_dmax - 0.f) / (float)c.rbf_dim);
        e.l0 = l == 0;
        e.A_h = h->t_A_h; e.A_v = h->t_A_v; e.fix = h->t_fix;
        e.wpack = h->d_wpack + (size_t)h->msg_base(l, 0) * PFT_WPACK_FLOATS;
        for (int lv = c.n_message_gvps - 1; lv >= 0; --lv) {
            e.level = lv;
            e.fx = 0;
            if (!h->no_fixed_shapes) {
                if (h->edge_fx.empty()) {
                    h->edge_fx.assign((size_t)L * c.n_message_gvps, 0);
                    for (int ll = 0; ll < L; ++ll)
                        for (int j = 0; j < c.n_message_gvps; ++j) {
                            bool f1 = true, f2 = true;
                            for (int et = 0; et < 4; ++et) {
                                const GvpSpec gs = msg_spec(c, ll, et, j);
                                f1 = f1 && gs.vi == 16 && gs.vo == 16 && gs.si == 128 && gs.so == 128;
                                f2 = f2 && gs.vi == 17 && gs.vo == 16 && gs.si == 144 && gs.so == 128;
                            }
                            h->edge_fx[(size_t)ll * c.n_message_gvps + j] = f1 ? 1 : (f2 ? 2 : 0);
                        }
                }
                e.fx = h->edge_fx[(size_t)l * c.n_message_gvps + lv];
            }
            ProfScope ps(h, pf_handle::K_BWD_EDGE_LEVEL, s);
            pfk_bwd_edge_level(&e, nb, s);
        }
        // the last layer's ff / pf edges scatter into pharm rows and active protein atoms only (the pruned layout's node tiles)
        if (last && l != 0 && h->prune && L >= 2 && h->d_act_ids != nullptr && h->n_node_tiles_act > 0)
            pfk_fix_apply_rows(h->d_node_tiles_act, h->n_node_tiles_act, h->d_dyn_cnt, h->d_act_ids, h->t_A_h, e.G_h_in, h->t_A_v, e.G_v_in,
                               h->t_fix, s);
        else if (l == 0 && enc_grouped && !h->no_fix_fuse) {
            // conv layer 0: the scalar sums join G_h in the same pass that groups the protein rows by (graph, element) for the encoders
            pfk_fix_enc_group(h->t_A_h, e.G_h_in, h->t_fix, h->d_prot_ptr, h->d_ptype, h->B, c.rec_nf, h->t_Gg, h->Np, h->Nf, h->d_l0flag, s);
            enc_grouped_done = true;
        } else {
            pfk_fix_apply(h->t_A_h, e.G_h_in, (size_t)N * PF_S, h->t_fix, s);
            if (l != 0) pfk_fix_apply(h->t_A_v, e.G_v_in, (size_t)N * 48, h->t_fix, s);       // conv layer 0 has no vector input
        }
        a ^= 1;
        // the head's, this layer's node and message classes are complete: with more layers to go, their gradient copies are
        // summed on the side stream now (bandwidth-bound, ~140 MB) under the next layer's kernels
        if (last && L >= 2 && h->s_side != s) {
            ReduceParams early = rp;
            early.cls_mask = (1u << PFT_CLS_HEAD) | (1u << (PFT_CLS_NODE + l));
            for (int et = 0; et < 4; ++et) early.cls_mask |= 1u << (PFT_CLS_MSG + l * 4 + et);
            PF_HIP(h, hipEventRecord(h->cmp_ev[0], s));
            PF_HIP(h, hipStreamWaitEvent(h->s_side, h->cmp_ev[0], 0));
            pfk_train_reduce(&early, h->s_side);
            PF_HIP(h, hipEventRecord(h->cmp_ev[1], h->s_side));
            early_mask = early.cls_mask;
        }
    }
    {
        BwdEncodeParams p{};
        p.c = tc; p.Np = h->Np; p.Nf = h->Nf;
        p.prot_h0 = h->d_prot_h0; p.pharm_h = h->d_pharm_h; p.t = h->d_t; p.gid = h->d_gid;
        p.rec_nf = c.rec_nf; p.pharm_nf = c.pharm_nf;
        for (int nt = 0; nt < 2; ++nt) {
            const std::string pre = std::string("dynamics.") + kNtKey[nt] + "_encoder.";
            p.o_w[nt] = (int)h->flat_offset(pre + "0.weight"); p.o_b[nt] = (int)h->flat_offset(pre + "0.bias");
            p.o_lw[nt] = (int)h->flat_offset(pre + "2.weight"); p.o_lb[nt] = (int)h->flat_offset(pre + "2.bias");
        }
        p.G_h = h->t_G_h[a];
        p.B = h->B; p.onehot_flag = h->d_l0flag;
        p.Gg = (c.rec_nf <= 16 && h->Np > 0) ? h->t_Gg : nullptr;
        if (p.Gg && !enc_grouped_done) pfk_enc_group(p.G_h, h->d_prot_ptr, h->d_ptype, h->B, c.rec_nf, h->t_Gg, s);
        const int tiles = (h->Np + PFT_ROWS - 1) / PFT_ROWS + (h->Nf + PFT_ROWS - 1) / PFT_ROWS;
        rp.enc_grid = std::max(1, std::min(PFT_ENC_BLOCKS, tiles));
        pfk_bwd_encode(&p, rp.enc_grid, s);
    }
    rp.cls_mask = ~early_mask;
    pfk_train_reduce(&rp, s);
    if (early_mask) PF_HIP(h, hipStreamWaitEvent(s, h->cmp_ev[1], 0));       // the gradient is complete on the caller's stream
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PF_FAIL(h, PF_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    h->tA_dirty = false;
    return PF_OK;
}

int pf_train_set_precision(pf_handle* h, int32_t precision) {
    if (!h) return PF_ERR_ARG;
    if (precision != PF_TRAIN_F32 && precision != PF_TRAIN_BF16) PF_FAIL(h, PF_ERR_ARG, "pf_train_set_precision: precision must be PF_TRAIN_F32 or PF_TRAIN_BF16");
    h->train_bf16 = precision == PF_TRAIN_BF16;
    h->t_have_fwd = false;                       // a forward of the other precision is not this backward's
    h->t_have_loss = false;
    return PF_OK;
}

int pf_train_get_precision(pf_handle* h, int32_t* precision) {
    if (!h || !precision) return PF_ERR_ARG;
    *precision = h->train_bf16 ? PF_TRAIN_BF16 : PF_TRAIN_F32;
    return PF_OK;
}

int pf_debug_set_dropout_masks(pf_handle* h, const float* dev_masks) {
    if (!h) return PF_ERR_ARG;
    h->t_mask_override = dev_masks;
    return PF_OK;
}

int pf_debug_dropout_mask(pf_handle* h, int32_t layer, int32_t which, float dropout_p, uint32_t seed, float* dev_out, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!dev_out || layer < 0 || layer >= h->cfg.n_convs || which < 0 || which > 1) PF_FAIL(h, PF_ERR_ARG, "pf_debug_dropout_mask: bad argument");
    TrainCommon tc{};
    tc.drop_thr = dropout_p > 0.f ? (uint32_t)std::min(4294967295.0, (double)dropout_p * 4294967296.0) : 0u;
    tc.drop_scale = 1.0f / (1.0f - dropout_p);
    tc.seed = seed;
    pfk_drop_masks(&tc, (uint32_t)layer * 2u + (uint32_t)which, h->N * 144, dev_out, (hipStream_t)stream);
    return PF_OK;
}

int pf_debug_ahead(pf_handle* h, int64_t* out, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!out) PF_FAIL(h, PF_ERR_ARG, "pf_debug_ahead: null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    PF_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    const int B = h->B;
    if (h->spec_valid && h->d_pa_same && h->d_dyn_cnt) {
        std::vector<int> cnt((size_t)5 * B), same(B);
        PF_HIP(h, hipMemcpy(cnt.data(), h->d_dyn_cnt, (size_t)5 * B * 4, hipMemcpyDeviceToHost));
        PF_HIP(h, hipMemcpy(same.data(), h->d_pa_same, (size_t)B * 4, hipMemcpyDeviceToHost));
        for (int g = 0; g < B; ++g) {                      // same[g]: leading 16-slot groups of the region that still apply (BuildParams::pa_same)
            const int64_t c = cnt[(size_t)3 * B + g];
            out[1] += c;
            out[0] += std::min<int64_t>(c, (int64_t)std::min(same[g], 1 << 26) * 16);
        }
    }
    if (h->cen_valid) { out[2] = 1; out[3] = h->Nf; }
    return PF_OK;
}

int pf_debug_kernel_family(pf_handle* h, int32_t layer, int32_t* rows_per_wave) {
    if (!h || !rows_per_wave) return PF_ERR_ARG;
    if (layer == (int)h->last_family.size() + 2 && layer > 2) {  // three past: 1 when the last call skipped "pa" regions whose rows had been computed ahead
        *rows_per_wave = h->last_spec;
        return PF_OK;
    }
    if (layer == (int)h->last_family.size() + 1 && layer > 1) {  // two past: 1 when the last call started conv layer 0's ff / fp items from the center-hoist tables
        *rows_per_wave = h->last_cen ? 1 : 0;
        return PF_OK;
    }
    if (layer == (int)h->last_family.size() && layer > 0) {      // one past the last conv layer: 16 when the last call's head ran in the tail launch
        *rows_per_wave = h->last_tail;
        return PF_OK;
    }
    if (layer < 0 || layer >= (int)h->last_family.size()) PF_FAIL(h, PF_ERR_STATE, "pf_debug_kernel_family: no dynamics call yet, or bad layer");
    *rows_per_wave = h->last_family[layer];
    return PF_OK;
}

int pf_debug_last_eps(pf_handle* h, float* dev_eps_h, float* dev_eps_x, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (dev_eps_h) PF_HIP(h, hipMemcpyAsync(dev_eps_h, h->d_eps_h, (size_t)h->Nf * h->cfg.pharm_nf * 4, hipMemcpyDeviceToDevice, s));
    if (dev_eps_x) PF_HIP(h, hipMemcpyAsync(dev_eps_x, h->d_eps_x, (size_t)h->Nf * 3 * 4, hipMemcpyDeviceToDevice, s));
    return PF_OK;
}

int pf_debug_xchg_timeouts(pf_handle* h, int32_t* n) {
    if (!h || !n) return PF_ERR_ARG;
    *n = 0;
    if (!h->d_xstat) return PF_OK;
    PF_HIP(h, hipDeviceSynchronize());
    PF_HIP(h, hipMemcpy(n, h->d_xstat, sizeof(int32_t), hipMemcpyDeviceToHost));
    return PF_OK;
}

int pf_debug_chain(pf_handle* h, int32_t kind, int32_t layer, int32_t sub, int32_t n_rows, const float* dev_s_in, const float* dev_v_in,
                   float* dev_s_out, float* dev_v_out, pf_stream stream) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    const pf_config& c = h->cfg;
    if (kind == 16 || kind == 17) {           // the message / update chain in the n16 form (pf_n16.hip); rows as for kinds 0 / 1
        n16_refresh(h, (hipStream_t)stream);
        if (n_rows < 0 || !dev_s_in || !dev_v_in || !dev_s_out || !dev_v_out) PF_FAIL(h, PF_ERR_ARG, "pf_debug_chain: bad argument");
        if (layer < 0 || layer >= c.n_convs || sub < 0 || sub > (kind == 16 ? 3 : 1)) PF_FAIL(h, PF_ERR_ARG, "pf_debug_chain: bad layer / sub index");
        if (h->n16_msg.empty()) PF_FAIL(h, PF_ERR_STATE, "pf_debug_chain: this architecture has no n16 streams (needs n_message_gvps >= 2)");
        UnitParams p{};
        p.s_in = dev_s_in; p.v_in = dev_v_in; p.s_out = dev_s_out; p.v_out = dev_v_out;
        p.n = n_rows; p.kind = kind;
        if (kind == 16) { p.stream = h->d_w + h->n16_msg[(size_t)layer * 4 + sub]; p.n_gvps = c.n_message_gvps; p.n16_stride = (int)h->n16_msg_stride; }
        else { p.stream = h->d_w + h->n16_upd[(size_t)layer * 2 + sub]; p.n_gvps = c.n_update_gvps; p.n16_stride = (int)h->n16_upd_stride; }
        pfk_n16_unit(&p, (hipStream_t)stream);
        hipError_t e16 = hipGetLastError();
        if (e16 != hipSuccess) PF_FAIL(h, PF_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e16));
        return PF_OK;
    }
    if (kind < 0 || kind > 3 || n_rows < 0 || !dev_s_in || !dev_v_in || !dev_s_out || !dev_v_out) PF_FAIL(h, PF_ERR_ARG, "pf_debug_chain: bad argument");
    if (kind != 3 && (layer < 0 || layer >= c.n_convs)) PF_FAIL(h, PF_ERR_ARG, "pf_debug_chain: bad layer");
    if ((kind == 0 && (sub < 0 || sub > 3)) || ((kind == 1 || kind == 2) && (sub < 0 || sub > 3))) PF_FAIL(h, PF_ERR_ARG, "pf_debug_chain: bad sub index");
    if (h->rg_msg.empty()) PF_FAIL(h, PF_ERR_STATE, "pf_debug_chain: no row-group streams (weights not committed)");
    UnitParams p{};
    p.s_in = dev_s_in; p.v_in = dev_v_in; p.s_out = dev_s_out; p.v_out = dev_v_out;
    p.n = n_rows; p.kind = kind; p.pharm_nf = c.pharm_nf;
    if (kind == 0) { p.stream = h->d_w + h->rg_msg[(size_t)layer * 4 + sub]; p.n_gvps = c.n_message_gvps; }
    else if (kind == 1) {
        if (sub > 1) PF_FAIL(h, PF_ERR_ARG, "pf_debug_chain: node type 0 (prot) or 1 (pharm)");
        p.stream = h->d_w + h->rg_upd[(size_t)layer * 2 + sub]; p.n_gvps = c.n_update_gvps;
    } else if (kind == 2) {           // sub: 2 * node type + (0: message_layer_norms, 1: update_layer_norms)
        const size_t* lo = &h->ln_off[(size_t)(layer * 2 + (sub >> 1)) * 4];
        p.ln_w = h->d_w + lo[(sub & 1) * 2]; p.ln_b = h->d_w + lo[(sub & 1) * 2 + 1];
    } else {
        p.stream = h->d_w + h->rg_upd[(size_t)(c.n_convs - 1) * 2 + 1]; p.n_gvps = c.n_noise_gvps; p.skip_gvps = c.n_update_gvps;
    }
    pfk_rg_unit(&p, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) PF_FAIL(h, PF_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return PF_OK;
}

int pf_debug_l0_hoist(pf_handle* h, int32_t* rows_per_wave) {
    if (!h || !rows_per_wave) return PF_ERR_ARG;
    *rows_per_wave = h->last_hoist;
    return PF_OK;
}

int pf_profile_enable(pf_handle* h, uint32_t kernel_mask) {
    if (!h) return PF_ERR_ARG;
    h->prof_mask = kernel_mask;             // recorded events accumulate until pf_profile_read
    return PF_OK;
}

static int profile_read_range(pf_handle* h, int k0, int k1, double* total_ms, int64_t* launches, pf_stream stream) {
    if (!h || !total_ms || !launches) return PF_ERR_ARG;
    PF_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    for (int k = k0; k < k1; ++k) {
        double tot = 0.0;
        for (size_t i = 0; i < h->prof_used[k]; ++i) {
            float ms = 0.f;
            PF_HIP(h, hipEventElapsedTime(&ms, h->prof_ev[k][i].first, h->prof_ev[k][i].second));
            tot += ms;
        }
        total_ms[k - k0] = tot;
        launches[k - k0] = (int64_t)h->prof_used[k];
        h->prof_used[k] = 0;
    }
    return PF_OK;
}

int pf_profile_read(pf_handle* h, double* total_ms, int64_t* launches, pf_stream stream) {
    return profile_read_range(h, 0, pf_handle::K_BWD_HEAD, total_ms, launches, stream);
}
int pf_profile_read_train(pf_handle* h, double* total_ms, int64_t* launches, pf_stream stream) {
    return profile_read_range(h, pf_handle::K_BWD_HEAD, pf_handle::K_NUM, total_ms, launches, stream);
}

int pf_debug_counts(pf_handle* h, int64_t* out, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!out) PF_FAIL(h, PF_ERR_ARG, "pf_debug_counts: null argument");
    PF_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    std::vector<int> cnt((size_t)5 * h->B);
    PF_HIP(h, hipMemcpy(cnt.data(), h->d_dyn_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    for (int g = 0; g < h->B; ++g) {
        for (int et = 0; et < 3; ++et) out[et] += cnt[(size_t)et * h->B + g];
        out[4] += cnt[(size_t)3 * h->B + g];
        out[5] += cnt[(size_t)4 * h->B + g];
    }
    out[3] = h->Epp; out[6] = h->Nf; out[7] = h->Np;
    return PF_OK;
}

int pf_debug_work(pf_handle* h, double* flops, double* bytes, int64_t* n_edges, double* executed_flops,
                  int64_t* executed_edges, pf_stream stream) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    PF_HIP(h, hipStreamSynchronize(s));
    const pf_config& c = h->cfg;
    std::vector<int> cnt((size_t)5 * h->B);
    PF_HIP(h, hipMemcpy(cnt.data(), h->d_dyn_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
    int64_t ne[4] = {0, 0, 0, h->Epp}, n_pa = 0, n_act = 0;
    for (int g = 0; g < h->B; ++g) {
        for (int et = 0; et < 3; ++et) ne[et] += cnt[(size_t)et * h->B + g];
        n_pa += cnt[(size_t)3 * h->B + g];
        n_act += cnt[(size_t)4 * h->B + g];
    }
    const double E = (double)(ne[0] + ne[1] + ne[2] + ne[3]);
    // FLOPs at 2/MAC
    auto gvp_flops = [](int vi, int vo, int si, int so) {
        const int hd = std::max(vi, vo);
        return 2.0 * (vi * hd * 3 + hd * vo * 3 + (double)(hd + si) * so + (double)so * vo);
    };
    const double g0 = gvp_flops(17, 16, 144, 128), gg = gvp_flops(16, 16, 128, 128), gl = gvp_flops(16, 1, 128, 64);
    const double per_edge = g0 + (c.n_message_gvps - 1) * gg;
    const double per_node = c.n_update_gvps * gg;
    const double head = (c.n_noise_gvps - 1) * gg + gl + 2.0 * 64 * c.pharm_nf;
    const double enc = 2.0 * 128 * ((double)h->Np * (c.rec_nf + 1) + (double)h->Nf * (c.pharm_nf + 1));
    if (flops) *flops = c.n_convs * (E * per_edge + (double)h->N * per_node) + (double)h->Nf * head + enc;
    if (bytes) *bytes = c.n_convs * (E * 736.0 + (double)h->N * 1420.0);
    if (n_edges) for (int i = 0; i < 4; ++i) n_edges[i] = ne[i];
    // what the kernels actually compute: the last layer only feeds pharm nodes; the layer before it (when pruning
    // is on) only the active atoms
    const int prune_layer = (h->prune && c.n_convs >= 2) ? c.n_convs - 2 : -1;
    double ex = (double)h->Nf * head + enc;
    for (int l = 0; l < c.n_convs; ++l) {
        double el, nl;
        if (l == c.n_convs - 1) { el = (double)(ne[0] + ne[1]); nl = (double)h->Nf; }
        else if (l == prune_layer) { el = (double)(ne[0] + ne[1] + ne[2] + n_pa); nl = (double)(h->Nf + n_act); }
        else { el = E; nl = (double)h->N; }
        ex += el * per_edge + nl * per_node;
        // static hoist (last call): the pp edges of layer 0 skip their first message GVP but for its gates
        // (n16 form, last_hoist == 16: the pp AND pf edges skip the h_src block of the first scalar Linear and the Vh matrix product
        // -- a type-table row and 17 x 3 multiplications instead)
        if (l == 0 && h->last_hoist == 16) ex -= (double)((l == prune_layer ? n_pa : (l == c.n_convs - 1 ? 0 : ne[3])) + ne[1] + (h->last_cen ? ne[0] + ne[2] : 0)) * (2.0 * 128 * 128 + 2.0 * 16 * 17 * 3);
        else if (l == 0 && h->last_hoist) ex -= (double)(l == prune_layer ? n_pa : (l == c.n_convs - 1 ? 0 : ne[3])) * (g0 - 2.0 * 128 * 16);
        if (executed_edges) executed_edges[l] = (int64_t)el;
    }
    if (executed_flops) *executed_flops = ex;
    return PF_OK;
}

}  // extern "C"
