// Host side of the C ABI for the i-DQN gradient step: layout, workspace, kernel orchestration.
//
// Replaces (reference file:line):
//   iDQN.learn_on_batch          slimdqn/networks/idqn.py:96-109   (jit o vmap of value_and_grad + optax.adam)
//   iDQN.update_target_params    slimdqn/networks/idqn.py:74-94    (copy + shift / sync, as device copies)
//   DQNNet.apply                 slimdqn/networks/architectures/dqn.py:38-70
// One launch sequence per step on the caller's stream; nothing here synchronises with the host.
// Scratch buffers are addressed compactly with the ACTIVE number of 32-sample batch blocks of a call
// (slot = (net * nb + bb) * block); zero borders sit at fixed offsets inside every block-sized slot
// and only interiors are ever written, so they stay zero for any nb.
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>
#include <queue>
#include <functional>

#include <map>
#include <tuple>

#include "act_kernels.h"
#include "cnn_kernels.h"
#include "fc_kernels.h"
#include "fc_par_kernels.h"
#include "iqn_kernels.h"
#include "gcnn_kernels.h"
#include "dp_internal.h"

namespace {

struct ConvL {
    int K, S, PLh, PLw, CI, CO, IH, IW, OH, OW;
    long w_off, b_off;
};

// CUs the conv launches of the step are planned for (one 512-thread workgroup per CU, all co-resident).  IDQN_CUS < 256
// leaves the rest of the chip to a concurrent stream (the overlapped Dense_0 update, tools/probes/overlap_cumask.py).
// the few run-time switches of the shipped library (INTEGRATION.md lists them)
bool plan_print() { static const bool on = getenv("IDQN_PLAN_PRINT") != nullptr; return on; }  // launch plans to stderr
bool act_generic() { static const bool on = getenv("IDQN_ACT_GENERIC") != nullptr; return on; }  // acting through the batched forward

int cu_budget() { return 256; }  // CUs a launch plan may fill

void same_pad(int i, int k, int s, int* out, int* lo, int* hi) {
    *out = (i + s - 1) / s;
    int p = (*out - 1) * s + k - i;
    if (p < 0) p = 0;
    *lo = p / 2;
    *hi = p - *lo;
}

ActGeom make_geom(int H, int W, int C, int lo_h, int hi_h, int lo_w, int hi_w) {
    ActGeom g;
    g.H = H; g.W = W; g.C = C; g.lo_h = lo_h; g.lo_w = lo_w;
    g.Hp = H + lo_h + hi_h; g.Wp = W + lo_w + hi_w;
    g.block = (long)g.Hp * g.Wp * C * 32;
    return g;
}

// zero border a dout buffer needs so that the data-gradient loop never leaves it
void dgrad_pad(int I, int O, int K, int S, int PL, int* lo, int* hi) {
    int mn = 0, mx = O - 1;
    for (int i = 0; i < I; ++i)
        for (int k = 0; k < K; ++k) {
            int t = i + PL - k;
            if (((t % S) + S) % S) continue;
            int o = t >= 0 ? t / S : -((-t) / S);
            if (o < mn) mn = o;
            if (o > mx) mx = o;
        }
    *lo = -mn;
    *hi = mx - (O - 1);
}

struct Layout {
    int n_leaves = 0;
    idqn_leaf_t leaves[IDQN_MAX_LEAVES];
    long head_stride = 0;
};

void add_leaf(Layout& L, const char* name, int ndim, const long* shape, long* off) {
    idqn_leaf_t& lf = L.leaves[L.n_leaves++];
    memset(&lf, 0, sizeof(lf));
    snprintf(lf.name, sizeof(lf.name), "%s", name);
    lf.ndim = ndim;
    long n = 1;
    for (int i = 0; i < ndim; ++i) { lf.shape[i] = shape[i]; n *= shape[i]; }
    lf.offset = *off;
    *off += (n + 63) / 64 * 64;  // every leaf starts on a 256-byte boundary
}

const int KS[3][2] = {{8, 4}, {4, 2}, {3, 1}};  // (kernel, stride) of Conv_0..2, architectures/dqn.py:43-51

// Shapes the MFMA plane kernels (and the f32-MFMA conv kernels) are built for; every other cnn shape the reference's
// DQNNet accepts runs on the general-shape kernels (gcnn_kernels.h).  IDQN_CNN_GENERAL=1 forces those (tests).
bool cnn_fast_shape(const idqn_config_t& c) {
    static const bool force_general = getenv("IDQN_CNN_GENERAL") && atoi(getenv("IDQN_CNN_GENERAL")) != 0;
    if (c.n_quantiles > 0) return true;  // the i-IQN heads exist on the MFMA path only (its checks reject other shapes)
    if (force_general) return false;
    if (c.n_features != 4 || c.obs_c != 4 || c.obs_h < 8 || c.obs_w < 8 || c.n_actions > 32) return false;
    for (int i = 0; i < 3; ++i)
        if (c.features[i] != 32 && c.features[i] != 64) return false;
    return c.features[3] % 128 == 0 && c.features[3] >= 128 && c.features[3] <= 512;
}

int build_layout(const idqn_config_t& c, Layout& L) {
    IDQN_REQUIRE(c.n_heads >= 1 && c.n_actions >= 1 && c.n_actions <= 4096, "n_heads >= 1 and 1 <= n_actions <= 4096 required");
    IDQN_REQUIRE(c.n_features >= 1 && c.n_features <= IDQN_MAX_FEATURES, "n_features out of range");
    IDQN_REQUIRE(c.max_batch >= 1, "max_batch must be positive");
    IDQN_REQUIRE(c.n_quantiles >= 0 && c.n_quantiles <= 64, "n_quantiles must be in [0, 64]");
    IDQN_REQUIRE(c.n_quantiles == 0 || (c.arch == IDQN_ARCH_CNN && c.max_batch <= 32),
                 "i-IQN heads are built for the cnn arch and minibatches of at most 32 samples");
    long off = 0;
    char nm[32];
    if (c.arch == IDQN_ARCH_CNN && !cnn_fast_shape(c)) {
        // general shape: three convs for features[0..2], Dense + ReLU for every features[3:], Dense(n_actions)
        // (architectures/dqn.py:39-53,65-70)
        IDQN_REQUIRE(c.n_quantiles == 0, "i-IQN heads need a shape the MFMA kernels are built for");
        IDQN_REQUIRE(c.n_features >= 3, "cnn: at least 3 features (the three convs), got %d", c.n_features);
        IDQN_REQUIRE(c.obs_h >= 1 && c.obs_w >= 1 && c.obs_c >= 1, "cnn: empty observation");
        int h = c.obs_h, w = c.obs_w, ch = c.obs_c;
        for (int i = 0; i < 3; ++i) {
            IDQN_REQUIRE(c.features[i] >= 1, "cnn: conv width %d", c.features[i]);
            long ks[4] = {KS[i][0], KS[i][0], ch, c.features[i]};
            snprintf(nm, sizeof nm, "Conv_%d/kernel", i);
            add_leaf(L, nm, 4, ks, &off);
            long bs[1] = {c.features[i]};
            snprintf(nm, sizeof nm, "Conv_%d/bias", i);
            add_leaf(L, nm, 1, bs, &off);
            int lo, hi;
            same_pad(h, KS[i][0], KS[i][1], &h, &lo, &hi);
            same_pad(w, KS[i][0], KS[i][1], &w, &lo, &hi);
            ch = c.features[i];
        }
        long fan = (long)h * w * ch;
        for (int i = 3; i <= c.n_features; ++i) {
            long f = i < c.n_features ? c.features[i] : c.n_actions;
            IDQN_REQUIRE(f >= 1, "cnn: dense width %ld", f);
            long ws[2] = {fan, f};
            snprintf(nm, sizeof nm, "Dense_%d/kernel", i - 3);
            add_leaf(L, nm, 2, ws, &off);
            long bs[1] = {f};
            snprintf(nm, sizeof nm, "Dense_%d/bias", i - 3);
            add_leaf(L, nm, 1, bs, &off);
            fan = f;
        }
    } else if (c.arch == IDQN_ARCH_CNN) {
        IDQN_REQUIRE(c.n_actions <= 32, "cnn (MFMA path): n_actions <= 32");
        IDQN_REQUIRE(c.n_features == 4, "cnn: exactly 4 features (three convs + one hidden dense) are built; got %d", c.n_features);
        IDQN_REQUIRE(c.obs_c == 4, "cnn: obs_c must be 4 (Conv_0 packs (kw, c) into 32 rows), got %d", c.obs_c);
        IDQN_REQUIRE(c.obs_h >= 8 && c.obs_w >= 8, "cnn: observation smaller than the first kernel");
        for (int i = 0; i < 3; ++i)
            IDQN_REQUIRE(c.features[i] == 32 || c.features[i] == 64, "cnn: conv features must be 32 or 64, got %d", c.features[i]);
        IDQN_REQUIRE(c.features[3] % 128 == 0 && c.features[3] >= 128 && c.features[3] <= 512,
                     "cnn: dense width must be a multiple of 128 in [128, 512], got %d", c.features[3]);
        int h = c.obs_h, w = c.obs_w, ch = c.obs_c;
        for (int i = 0; i < 3; ++i) {
            long ks[4] = {KS[i][0], KS[i][0], ch, c.features[i]};
            snprintf(nm, sizeof nm, "Conv_%d/kernel", i);
            add_leaf(L, nm, 4, ks, &off);
            long bs[1] = {c.features[i]};
            snprintf(nm, sizeof nm, "Conv_%d/bias", i);
            add_leaf(L, nm, 1, bs, &off);
            int lo, hi;
            same_pad(h, KS[i][0], KS[i][1], &h, &lo, &hi);
            same_pad(w, KS[i][0], KS[i][1], &w, &lo, &hi);
            ch = c.features[i];
        }
        long d0[2] = {(long)h * w * ch, c.features[3]};
        add_leaf(L, "Dense_0/kernel", 2, d0, &off);
        long b0[1] = {c.features[3]};
        add_leaf(L, "Dense_0/bias", 1, b0, &off);
        long d1[2] = {c.features[3], c.n_actions};
        add_leaf(L, "Dense_1/kernel", 2, d1, &off);
        long b1[1] = {c.n_actions};
        add_leaf(L, "Dense_1/bias", 1, b1, &off);
        if (c.n_quantiles > 0) {  // i-IQN heads (extension): the quantile embedding, phi(tau) = relu(Embed_0(cos(pi i tau)))
            long es[2] = {IQN_EMBED, d0[0]};
            add_leaf(L, "Embed_0/kernel", 2, es, &off);
            long eb[1] = {d0[0]};
            add_leaf(L, "Embed_0/bias", 1, eb, &off);
        }
    } else if (c.arch == IDQN_ARCH_FC) {

        long fan = (long)c.obs_h * c.obs_w * c.obs_c;
        for (int i = 0; i <= c.n_features; ++i) {
            long f = i < c.n_features ? c.features[i] : c.n_actions;
            IDQN_REQUIRE(f >= 1 && fan >= 1, "fc: layer widths must be positive");  // (wider than FC_MAX_WIDTH: the generic kernels)
            long ws[2] = {fan, f};
            snprintf(nm, sizeof nm, "Dense_%d/kernel", i);
            add_leaf(L, nm, 2, ws, &off);
            long bs[1] = {f};
            snprintf(nm, sizeof nm, "Dense_%d/bias", i);
            add_leaf(L, nm, 1, bs, &off);
            fan = f;
        }
    } else {
        IDQN_REQUIRE(false, "unknown arch %d (impala is out of scope of the HIP path)", c.arch);
    }
    L.head_stride = off;
    return IDQN_OK;
}

// One set of nets that run forward together: the 2K training nets, or the single inference net.
struct NetSet {
    int n_nets = 0, nb_cap = 0, n_in_sets = 0;
    int NS = 0;  // split-K of this set's Dense_0 forward
    long part_slabs = 0;  // (block, split) slabs the partial buffer holds per net
    const float** wbase = nullptr;  // dev [n_nets]
    int* in_set = nullptr;          // dev [n_nets] input set read by Conv_0
    int* ident = nullptr;           // dev [n_nets] 0..n_nets-1 (later layers read their own activations)
    float *x = nullptr, *a1 = nullptr, *a2 = nullptr, *a3 = nullptr, *part = nullptr;  // x, a1, a2: f32 conv path only
    unsigned short *x1 = nullptr, *a1p = nullptr, *a2p = nullptr, *wq = nullptr;       // plane conv path (convp.h)
};

// host-side plans of the plane conv launches: the work items of a launch depend only on geometry, net count and batch
// blocks, so they are built once and kept on the device
struct FwdPlan { int n_items = 0, NT = 0, ring = 2, items_per_slot = 0, r_begin[4] = {0, 0, 0, 0}, r_cnt[4] = {1, 1, 1, 1}; size_t stage = 0, lds = 0;
                 int row_parts = 0, n_wg = 0; CItem* items_dev = nullptr; };  // row_parts > 0: a plan of the persistent kernel (convp_pp.hip), n_wg workgroups
struct WgradPlan { int n_items = 0, n_chunks = 0, chunk_major = 0, MT = 0, PG = 0; size_t lds = 0; };

// workspace of the i-IQN heads (iqn_kernels.h): V = 3K virtual nets x N fraction blocks
struct IqnWs {
    int N = 0, V = 0, NS = 2;
    const float** wbase_v = nullptr;  // dev [V]: online k | target k | target k
    unsigned short* cosa = nullptr;  // A-fragment planes of the cos blocks [V * N][12][512]
    unsigned short *cosp = nullptr, *wep = nullptr;  // bf16 fragment planes of the cos blocks [V * N][12][512] and of We [2K][F / 32][12][512]
    float *xq = nullptr, *part = nullptr, *hbuf = nullptr, *qpart = nullptr, *dq = nullptr, *dh = nullptr,
          *dx = nullptr, *dpsi = nullptr, *dbg = nullptr, *z = nullptr;
    int QG = 1;              // fraction groups of the embedding backward (partials dpsi / gpart)
    float* gpart = nullptr;  // [QG][K][65][F]
    int HG = 1;              // fraction groups of k_iqn_dh (partials hpart)
    float* hpart = nullptr;  // [HG][K][J * A + J + A]
    float* clk = nullptr;  // IDQN_IQN_CLOCK=1: clock stamps of the forward GEMM's workgroups [<= 1024][4] int64 (debug buffer "iqn_clk")
    int32_t* bwd_items = nullptr;  // dispatch order of the merged Dense_0 gradient launch with Adam in its epilogue (plan_iqn_bwd_order)
    int bwd_blocks = 0;
    unsigned* gate = nullptr;  // [2 * K * ceil(F / 256) + 64]: IqnD0Gate (arrived, passed, err) of the fused Dense_0 update
    bool d0_adam_done = false;  // this step's merged gradient launch updated Dense_0/kernel itself
    float* g1 = nullptr;  // second partial of the Dense_0 weight gradient [K][F * J] (iqn_gemm.h), N a multiple of 16 only
    long off_we = 0, off_be = 0;
};

// workspace of the general-shape cnn path (gcnn_kernels.h): NHWC f32 activations of the 2K nets, their gradients for the K
// online nets, and the dense head as an FcNet over the flattened conv features
struct GcnnWs {
    bool on = false;
    int Bmax = 0;
    float *act[3] = {nullptr, nullptr, nullptr};   // [2K][Bmax][OH][OW][CO] of Conv_0..2 (act[2] flattened = the head's input)
    float *dact[3] = {nullptr, nullptr, nullptr};  // [K][Bmax][...]: gradient w.r.t. the pre-activation of Conv_0..2
    float *inf_act[3] = {nullptr, nullptr, nullptr};  // one net, 32 states (acting)
    float* inf_ws = nullptr;
};

}  // namespace

struct idqn_handle_s {
    idqn_config_t cfg;
    Layout L;
    float *online, *target, *mu, *nu, *grad, *losses;
    int32_t* count;
    double* cum;
    AdamConsts ad;
    float gamma_n;
    int nb_max;
    // cnn
    ConvL conv[3];
    ActGeom gx, ga1, ga2, ga3, gda3, gda2, gda1;
    int F = 0, J = 0, NS = 0;
    long off_w0 = 0, off_b0 = 0, off_w1 = 0, off_b1 = 0;
    NetSet train, infer;
    IqnWs iqn;
    GcnnWs gc;
    float* dpart = nullptr;  // partial Dense_0 data gradients of the fused weight-gradient kernel [n_jt][K * nb][F][32]
    float *da3 = nullptr, *da2 = nullptr, *da1 = nullptr, *qdbg = nullptr, *slab = nullptr;
    float *hbuf = nullptr, *qpart = nullptr, *bcinv = nullptr;
    float *act_a[3] = {nullptr, nullptr, nullptr}, *act_part = nullptr;  // single-state acting path (act_kernels.h)
    hipStream_t act_stream = nullptr;  // capture stream of the acting graphs
    uint8_t* act_state = nullptr;     // the state of idqn_act_host on the device
    int32_t* act_action = nullptr;
    int32_t* act_mail = nullptr;      // host mailbox {action, sequence} (hipHostMalloc, mapped + coherent) of idqn_act_host
    int32_t* act_mail_dev = nullptr;  // its device address
    unsigned* act_seq = nullptr;      // device-side sequence counter
    unsigned act_expected = 0;        // sequence number the next idqn_act_host call waits for
    bool act_use_mail = false;        // set while idqn_act_host issues / captures its launches
    int act_pending = 0;              // idqn_act_host_begin launched, idqn_act_host_end has not collected yet (1 mailbox, 2 copy)
    // IDQN_STEP_GRAPH=1: the plain cnn step replayed as a hipGraph per (batch buffers, size); value = calls seen, graph
    std::map<std::tuple<const void*, const void*, const void*, const void*, const void*, int, int, const void*, const void*>,
             std::pair<int, hipGraphExec_t>> step_graphs;
    std::map<std::tuple<int, const void*, void*, void*>, hipGraphExec_t> act_graphs;  // (net, host state, q out, host action)
    const float* infer_pbase = nullptr;  // parameter base of the net the last idqn_q_values call evaluated
    float *infer_hbuf = nullptr, *infer_qpart = nullptr;  // k_hidden outputs of the single inference net
    float* wt[3] = {nullptr, nullptr, nullptr};  // transformed weights of the Conv_1 / Conv_2 data gradients
    long wt_stride[3] = {0, 0, 0};
    // plane conv path (convp.h; the default): packed weight planes per net, plane dout buffers, per-position dy sums
    bool planes = true;
    unsigned short *da3p = nullptr, *da2p = nullptr, *da1p = nullptr;
    int dh_nb = 0;  // batch blocks of the step that last wrote dL/dh (debug buffer "dh")
    unsigned short* fact_planes = nullptr;  // bf16 planes of the gathered Dense_0 factors (factored data-parallel step)
    long fact_planes_cap = 0;               // in sample blocks
    float* pbuf[3] = {nullptr, nullptr, nullptr};  // [K * nb][OH * OW][CO] of conv layer i
    long wq_stride = 0, wq_fwd[3] = {0, 0, 0}, wq_dg[3] = {0, 0, 0};  // bytes
    std::map<std::tuple<int, int, int, int>, FwdPlan> fwd_plans;  // (role, n_nets, nb, target workgroups)
    std::map<std::tuple<int, int, int>, WgradPlan> wgrad_plans;   // (layer, nb, position chunks)
    int npc_used[3] = {0, 0, 0};  // position chunks (= slabs per head) the weight-gradient launches of THIS step wrote
    int n_cus = 256;  // CUs of the device
    float* td_bpart = nullptr;   // [K * J / 32][nb][TD_BPART] per-block gradient partials of k_td_dh (max_batch > 32)
    unsigned* td_bctr = nullptr;
    float* cprof = nullptr;  // debug (IDQN_CONV_PROF=role): phase stamps of one plane conv launch
    int cprof_role = -1;
    int npc[3], pos_per_chunk[3];
    long slab_stride[3], slab_off[3];
    SlabSeg segs[3];  // slab descriptors of the last backward (consumed by the fused Adam launch)
    // fc
    FcNet fc;
    float* fc_ws = nullptr;
    FcPlan fc_plan_;  // LDS plan of k_fc_step_lds (BS = 0: the net does not fit and the generic kernel runs)
    FcMfmaPlan fcm_plan_;  // LDS plan of k_fc_step_mfma (floats = 0: neither the staged-weights nor the global-weights layout fits)
    bool fcm_global_ = false;  // the plan is fc_mfma_plan_g: weight operands from global memory
    FcParPlan fcp_plan_;  // LDS plan of k_fc_step_par (floats = 0: does not fit); batches of <= 32 samples run it
    // timeline of a whole step (IDQN_F_PROFILE_ALL): one event after every launch; idqn_profile_table averages per name
    std::vector<hipEvent_t> tl_ev;
    std::vector<const char*> tl_name;
    int tl_used = 0;
    bool tl_on = false;
    // profiling of the dominant kernel
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
    const char* dominant = "";
    // Gradient arena layout: [K][gP] small leaves (every leaf but the cnn's Dense_0/kernel, same order, per-head
    // stride gP = P - w0n), 64 floats reserved for the caller (losses), then [K][w0n] Dense_0/kernel gradients.
    // Two contiguous regions = two collectives in the data-parallel step.  fc: w0n = 0, gP = P.
    long gP = 0, g_w0_begin = 0, g_w0_end = 0, g_w0_base = 0;
    // replay-sourced step (idqn_learn_on_replay): set for the duration of the call, the staging launch gathers from the ring
    struct ReplaySrc { const uint8_t* frames; const int32_t* rows; long n_frames, frame_bytes; StageSlots slots; };
    const ReplaySrc* rp = nullptr;
    int32_t* rp_action = nullptr;   // [max_batch] scalars of the sampled rows, written by the staging launch
    float* rp_reward = nullptr;
    uint8_t* rp_terminal = nullptr;
    const float* is_weight = nullptr;  // prioritized-replay extension (idqn_set_per_buffers)
    float* td_abs = nullptr;
    bool d0_rows = false;  // the last fused Dense_0 launch (pairs of column tiles) finished dL/da3 itself
    bool wt_ready = false;  // the data-gradient kernels of this step are built (k_td_dh_wt)
    bool pend_profile = false;
    int pend_stage = 0;  // 1: stopped before the Dense_0 weight gradient, 2: stopped after it
    int pend_B = 0;  // batch of a backward stopped after Dense_0 (idqn_backward_rest resumes it); 0 = none
    std::vector<void*> owned;
    std::vector<std::pair<std::string, std::pair<void*, long>>> dbg;
};

namespace {

int alloc_zero(float** p, long n_floats, idqn_handle_s* h, const char* name) {
    IDQN_HIP_CHECK(hipMalloc((void**)p, (size_t)n_floats * 4));
    IDQN_HIP_CHECK(hipMemset(*p, 0, (size_t)n_floats * 4));
    h->owned.push_back((void*)*p);
    h->dbg.push_back({name, {(void*)*p, n_floats * 4}});
    return IDQN_OK;
}

int alloc_zero16(unsigned short** p, long n, idqn_handle_s* h, const char* name) {
    IDQN_HIP_CHECK(hipMalloc((void**)p, (size_t)n * 2));
    IDQN_HIP_CHECK(hipMemset(*p, 0, (size_t)n * 2));
    h->owned.push_back((void*)*p);
    h->dbg.push_back({name, {(void*)*p, n * 2}});
    return IDQN_OK;
}

// dL/dh [K][nb][J][32] of a batch of nb blocks: ends where the training set's a3 begins (netset_alloc)
float* dh_of(idqn_handle_s* h, int nb) {
    h->dh_nb = nb;
    return h->train.a3 - (long)h->cfg.n_heads * nb * h->J * 32;
}

// step timeline: an event on the stream after the launch just made (name = nullptr: start of a step)
void tl_mark(idqn_handle_s* h, hipStream_t q, const char* name) {
    if (!h->tl_on || h->tl_used >= (int)h->tl_ev.size()) return;
    if (hipEventRecord(h->tl_ev[h->tl_used], q) != hipSuccess) return;
    h->tl_name[h->tl_used++] = name;
}

// The online nets' Dense_0 kernels are streamed three times per step (forward; fused update: read and written).  While they fit the
// 256 MB memory-side cache beside the step's other traffic -- K * F * J * 4 <= 84 MB: K <= 5 at the Nature width -- the forward reads
// them and the update stores them with the default policy, so that part of them is found on chip again; everything read or written
// once per step (target nets, m, v) stays non-temporal.  More heads: every stream non-temporal, as in rounds 3-4 (the dirty lines
// of a set that cannot stay only get in the way: K = 6 the same, K = 8 +13 us; keeping only the 4-5 heads that would fit: no better
// than none -- profiles/r5_d0_keep_online_ab.txt).
int d0_keep_heads(const idqn_handle_s* h) {
    return (long)h->cfg.n_heads * h->F * h->J * 4 <= 84L << 20 ? h->cfg.n_heads : 0;
}
bool d0_keep_online(const idqn_handle_s* h) { return d0_keep_heads(h) >= h->cfg.n_heads; }  // every head's kernel fits

// k-splits of the Dense_0 forward of `n_nets` nets x `nb` sample blocks: as many 4-wave workgroups as CUs, never more (a 257th
// would stream alone after the others), with balanced splits of the F / 32 row units (cnn_setup has the reasoning)
int d0_splits(const idqn_handle_s* h, int n_nets, int nb) {
    const int ns = 256 * 4 / std::max(1, n_nets * nb * (h->J / 128));
    return std::max(1, std::min(std::min(ns, 64), h->F / 32));
}

int netset_alloc(idqn_handle_s* h, NetSet& s, int n_nets, int nb, int n_in_sets, const char* tag, int units_per_split = 0) {
    s.n_nets = n_nets; s.nb_cap = nb; s.n_in_sets = n_in_sets;
    s.NS = h->NS;
    if (units_per_split > 0) {  // a single acting net: more, shorter splits (each wave's MFMA chain is the latency)
        const int units = h->F / 32;
        s.NS = (units + units_per_split - 1) / units_per_split;
    }
    IDQN_HIP_CHECK(hipMalloc((void**)&s.wbase, sizeof(float*) * n_nets));
    IDQN_HIP_CHECK(hipMalloc((void**)&s.in_set, sizeof(int) * n_nets));
    IDQN_HIP_CHECK(hipMalloc((void**)&s.ident, sizeof(int) * n_nets));
    h->owned.push_back((void*)s.wbase); h->owned.push_back((void*)s.in_set); h->owned.push_back((void*)s.ident);
    std::vector<int> id(n_nets);
    for (int i = 0; i < n_nets; ++i) id[i] = i;
    IDQN_HIP_CHECK(hipMemcpy(s.ident, id.data(), sizeof(int) * n_nets, hipMemcpyHostToDevice));
    std::string t(tag);
    int rc;
    if (h->planes) {  // bf16 planes: one for the pixels, three per activation (block = rows x 32 samples)
        if ((rc = alloc_zero16(&s.x1, (long)n_in_sets * nb * h->gx.block, h, (t + "x1").c_str()))) return rc;
        if ((rc = alloc_zero16(&s.a1p, (long)n_nets * nb * h->ga1.block * 3, h, (t + "a1p").c_str()))) return rc;
        if ((rc = alloc_zero16(&s.a2p, (long)n_nets * nb * h->ga2.block * 3, h, (t + "a2p").c_str()))) return rc;
        if ((rc = alloc_zero16(&s.wq, (long)n_nets * h->wq_stride / 2, h, (t + "wq").c_str()))) return rc;
    } else {
        if ((rc = alloc_zero(&s.x, (long)n_in_sets * nb * h->gx.block, h, (t + "x").c_str()))) return rc;
        if ((rc = alloc_zero(&s.a1, (long)n_nets * nb * h->ga1.block, h, (t + "a1").c_str()))) return rc;
        if ((rc = alloc_zero(&s.a2, (long)n_nets * nb * h->ga2.block, h, (t + "a2").c_str()))) return rc;
    }
    {
        // The training set's a3 (online nets first) sits directly behind the dL/dh area: for a batch of nb blocks dL/dh is
        // placed so that it ENDS where a3 begins, and [dL/dh | a3 of the K online nets] is one contiguous run -- the two
        // factors of the Dense_0 gradient a data-parallel rank sends (idqn_dense0_factors) without copying them anywhere.
        const long front = (&s == &h->train) ? (long)h->cfg.n_heads * nb * h->J * 32 : 0;
        const long n_a3 = (long)n_nets * nb * h->ga3.block;
        float* base = nullptr;
        if ((rc = alloc_zero(&base, front + n_a3, h, (t + "a3_block").c_str()))) return rc;
        s.a3 = base + front;
        h->dbg.push_back({t + "a3", {(void*)s.a3, n_a3 * 4}});
    }
    long slabs = (long)nb * s.NS;  // the training set picks its splits per call (fewer, longer ones for more blocks): room for the largest
    if (&s == &h->train)
        for (int b = 1; b <= nb; ++b) slabs = std::max(slabs, (long)b * d0_splits(h, n_nets, b));
    s.part_slabs = slabs;
    if ((rc = alloc_zero(&s.part, (long)n_nets * slabs * h->J * 32, h, (t + "part").c_str()))) return rc;
    return IDQN_OK;
}

int plan_iqn_bwd_order(idqn_handle_s* h, int forced_e);
int cnn_setup(idqn_handle_s* h) {
    const idqn_config_t& c = h->cfg;
    int ih = c.obs_h, iw = c.obs_w, ci = c.obs_c;
    int lo_h[3], hi_h[3], lo_w[3], hi_w[3];
    for (int i = 0; i < 3; ++i) {
        ConvL& l = h->conv[i];
        l.K = KS[i][0]; l.S = KS[i][1]; l.CI = ci; l.CO = c.features[i]; l.IH = ih; l.IW = iw;
        same_pad(ih, l.K, l.S, &l.OH, &lo_h[i], &hi_h[i]);
        same_pad(iw, l.K, l.S, &l.OW, &lo_w[i], &hi_w[i]);
        l.PLh = lo_h[i]; l.PLw = lo_w[i];
        l.w_off = h->L.leaves[2 * i].offset;
        l.b_off = h->L.leaves[2 * i + 1].offset;
        ih = l.OH; iw = l.OW; ci = l.CO;
    }
    h->off_w0 = h->L.leaves[6].offset; h->off_b0 = h->L.leaves[7].offset;
    h->off_w1 = h->L.leaves[8].offset; h->off_b1 = h->L.leaves[9].offset;
    const ConvL *c0 = &h->conv[0], *c1 = &h->conv[1], *c2 = &h->conv[2];
    h->gx = make_geom(c0->IH, c0->IW, c0->CI, lo_h[0], hi_h[0], lo_w[0], hi_w[0]);
    h->ga1 = make_geom(c1->IH, c1->IW, c1->CI, lo_h[1], hi_h[1], lo_w[1], hi_w[1]);
    h->ga2 = make_geom(c2->IH, c2->IW, c2->CI, lo_h[2], hi_h[2], lo_w[2], hi_w[2]);
    h->ga3 = make_geom(c2->OH, c2->OW, c2->CO, 0, 0, 0, 0);
    int l, hh, l2, h2;
    dgrad_pad(c2->IH, c2->OH, c2->K, c2->S, c2->PLh, &l, &hh);
    dgrad_pad(c2->IW, c2->OW, c2->K, c2->S, c2->PLw, &l2, &h2);
    h->gda3 = make_geom(c2->OH, c2->OW, c2->CO, l, hh, l2, h2);
    dgrad_pad(c1->IH, c1->OH, c1->K, c1->S, c1->PLh, &l, &hh);
    dgrad_pad(c1->IW, c1->OW, c1->K, c1->S, c1->PLw, &l2, &h2);
    h->gda2 = make_geom(c1->OH, c1->OW, c1->CO, l, hh, l2, h2);
    h->gda1 = make_geom(c0->OH, c0->OW, c0->CO, 0, 0, 0, 0);
    h->F = c2->OH * c2->OW * c2->CO;
    h->J = c.features[3];
    // split-K of the Dense_0 forward: a weight-streaming kernel is bound by ~10 B/clk per CU (MI355X_MICROARCH.md), so
    // what matters is that EVERY CU streams and that they all stream the same amount: as many 4-wave workgroups as CUs,
    // never more (a 257th would stream alone after the others), with balanced splits of the F / 32 row units
    {
        const int units = h->F / 32;
        int ns = 256 * 4 / std::max(1, 2 * c.n_heads * (c.features[3] / 128));
        h->NS = std::max(1, std::min(std::min(ns, 64), units));
    }
    const int K = c.n_heads, nb = h->nb_max;
    int rc;
    if (h->planes) {  // packed weights of one net: forward kernels of the three layers, then the data-gradient kernels
        long off = 0;
        for (int i = 0; i < 3; ++i) {
            h->wq_fwd[i] = off;
            off += (long)h->conv[i].K * h->conv[i].K * h->conv[i].CI * h->conv[i].CO * 6;  // 3 planes x 2 bytes
        }
        for (int i = 1; i < 3; ++i) {
            h->wq_dg[i] = off;
            off += (long)h->conv[i].K * h->conv[i].K * h->conv[i].CI * h->conv[i].CO * 6;
        }
        h->wq_stride = (off + 1023) / 1024 * 1024;
    }
    if ((rc = netset_alloc(h, h->train, 2 * K, nb, 2, ""))) return rc;
    if ((rc = netset_alloc(h, h->infer, 1, 1, 1, "infer_", 4))) return rc;
    if (h->planes && debug_env("IDQN_CONV_PROF")) {
        h->cprof_role = atoi(debug_env("IDQN_CONV_PROF"));
        // role 10 = the chained forward launch: one [2][4096][8] block of stamps per layer
        if ((rc = alloc_zero(&h->cprof, (h->cprof_role == 10 ? 3 : 1) * 2L * 2 * 8 * 4096, h, "cprof"))) return rc;
    }
    if (h->planes) {
        if (nb > 1) {
            float* c = nullptr;
            if ((rc = alloc_zero(&h->td_bpart, (long)K * (h->J / 32) * nb * TD_BPART, h, "td_bpart"))) return rc;
            if ((rc = alloc_zero(&c, (long)K * (h->J / 32) + 16, h, "td_bctr"))) return rc;
            h->td_bctr = reinterpret_cast<unsigned*>(c);
        }
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) h->n_cus = cus;
        if ((rc = alloc_zero16(&h->da3p, (long)K * nb * h->gda3.block * 3, h, "da3p"))) return rc;
        if ((rc = alloc_zero16(&h->da2p, (long)K * nb * h->gda2.block * 3, h, "da2p"))) return rc;
        if ((rc = alloc_zero16(&h->da1p, (long)K * nb * h->gda1.block * 3, h, "da1p"))) return rc;
        for (int i = 0; i < 3; ++i) {
            char nm[8];
            snprintf(nm, sizeof nm, "pb%d", i);
            if ((rc = alloc_zero(&h->pbuf[i], (long)K * nb * h->conv[i].OH * h->conv[i].OW * h->conv[i].CO, h, nm))) return rc;
        }
    }
    std::vector<const float*> wb(2 * K);
    std::vector<int> is(2 * K);
    for (int k = 0; k < K; ++k) {
        wb[k] = h->online + (long)k * h->L.head_stride; is[k] = 0;
        wb[K + k] = h->target + (long)k * h->L.head_stride; is[K + k] = 1;
    }
    IDQN_HIP_CHECK(hipMemcpy(h->train.wbase, wb.data(), sizeof(float*) * 2 * K, hipMemcpyHostToDevice));
    IDQN_HIP_CHECK(hipMemcpy(h->train.in_set, is.data(), sizeof(int) * 2 * K, hipMemcpyHostToDevice));
    IDQN_HIP_CHECK(hipMemset(h->infer.in_set, 0, sizeof(int)));
    for (int i = 0; i < 3; ++i) {
        char nm[12];
        snprintf(nm, sizeof nm, "act_a%d", i + 1);
        if ((rc = alloc_zero(&h->act_a[i], (long)h->conv[i].OH * h->conv[i].OW * h->conv[i].CO, h, nm))) return rc;
    }
    if ((rc = alloc_zero(&h->act_part, 256L * h->J, h, "act_part"))) return rc;
    {
        float* p = nullptr;
        if ((rc = alloc_zero(&p, ((long)c.obs_h * c.obs_w * c.obs_c + 3) / 4 + 16, h, "act_state"))) return rc;
        h->act_state = (uint8_t*)p;
        if ((rc = alloc_zero(&p, 16, h, "act_action"))) return rc;
        h->act_action = (int32_t*)p;
    }
    if (h->J % 256 == 0 && (rc = alloc_zero(&h->dpart, (long)(h->J / 256) * K * nb * h->F * 32, h, "dpart"))) return rc;
    if (!h->planes) {
        if ((rc = alloc_zero(&h->da3, (long)K * nb * h->gda3.block, h, "da3"))) return rc;
        if ((rc = alloc_zero(&h->da2, (long)K * nb * h->gda2.block, h, "da2"))) return rc;
        if ((rc = alloc_zero(&h->da1, (long)K * nb * h->gda1.block, h, "da1"))) return rc;
    }
    if ((rc = alloc_zero(&h->qdbg, (long)2 * K * nb * 32 * 32, h, "q"))) return rc;
    if ((rc = alloc_zero(&h->hbuf, (long)2 * K * nb * h->J * 32, h, "h"))) return rc;
    if ((rc = alloc_zero(&h->qpart, (long)2 * K * nb * (h->J / 32) * 32 * 32, h, "qpart"))) return rc;
    if ((rc = alloc_zero(&h->infer_hbuf, (long)h->J * 32, h, "infer_h"))) return rc;
    if ((rc = alloc_zero(&h->infer_qpart, (long)(h->J / 32) * 32 * 32, h, "infer_qpart"))) return rc;
    for (int i = 1; i < 3; ++i) {
        const ConvL& cl = h->conv[i];
        IDQN_REQUIRE(cl.K % cl.S == 0, "conv %d: kernel %d not a multiple of stride %d", i, cl.K, cl.S);
        h->wt_stride[i] = ((long)cl.K * cl.K * cl.CI * cl.CO + 63) / 64 * 64;
        if (!h->planes && (rc = alloc_zero(&h->wt[i], (long)K * h->wt_stride[i], h, i == 1 ? "wt1" : "wt2"))) return rc;
    }
    // weight-gradient slabs: one per workgroup chunk of 16 output positions; one region per conv layer
    long slab_total = 0;
    for (int i = 0; i < 3; ++i) {
        const ConvL& cl = h->conv[i];
        int npos = cl.OH * cl.OW;
        // Chunk size = positions per workgroup.  The launch has K * taps * npc equal workgroups, all resident at once,
        // so its time is (workgroups on the busiest CU) x (positions per workgroup + a fixed cost of ~2 positions for
        // the prologue / slab epilogue); a lone workgroup on a CU has nobody to hide its latencies behind (x 1.1).
        // Measured on the Atari shapes (tools/gpu_ppc.sh): Conv_0 19 -> 38 positions 31.4 -> 28.5 us, Conv_1 16 -> 41
        // positions 30.3 -> 26.0 us, Conv_2 stays at 11 (21.4 us); fewer chunks also mean fewer slabs to sum.
        {
            const int taps = (i == 0) ? cl.K : cl.K * cl.K;
            double best = 1e30;
            for (int npc = 1; npc <= npos; ++npc) {
                const int ppc = (npos + npc - 1) / npc;  // balanced chunks
                if (ppc < 4 && npc > 1) break;
                const long wgs = (long)K * taps * ((npos + ppc - 1) / ppc);
                const double cost = (double)((wgs + 255) / 256) * (ppc + 2.0) * (wgs <= 256 ? 1.1 : 1.0);
                if (cost < best) { best = cost; h->pos_per_chunk[i] = ppc; }
            }
        }
        {
            char nm[16];
            snprintf(nm, sizeof nm, "IDQN_PPC%d", i);  // experiment knob: positions per weight-gradient chunk
            if (const char* e = debug_env(nm)) { const int v = atoi(e); if (v >= 1 && v <= npos) h->pos_per_chunk[i] = v; }
        }
        h->npc[i] = (npos + h->pos_per_chunk[i] - 1) / h->pos_per_chunk[i];
        if (h->planes) {  // plane path: (head, kernel row, chunk) workgroups, about one per CU (Conv_0: (head, chunk))
            const int per_chunk = K * (i == 0 ? 1 : cl.K);
            // never more workgroups than CUs (a 257th would run alone after the others), never more than 64 chunks: k_adam adds a leaf's
            // slabs one after the other (K = 1: 256 chunks of Conv_0 cost it 5 us more than they save the weight gradient)
            int nch = std::min(cu_budget() / per_chunk, 64);
            h->npc[i] = std::max(1, std::min(nch, npos));
        }
        h->slab_stride[i] = ((long)cl.K * cl.K * cl.CI * cl.CO + cl.CO + 63) / 64 * 64;
        h->slab_off[i] = slab_total;
        slab_total += (long)h->npc[i] * K * h->slab_stride[i];
    }
    if ((rc = alloc_zero(&h->slab, slab_total, h, "slab"))) return rc;
    IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_dense0_dgrad<4>, hipFuncAttributeMaxDynamicSharedMemorySize, h->J * 32 * 4));
    IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_dense0_dgrad<3>, hipFuncAttributeMaxDynamicSharedMemorySize, h->J * 32 * 4));
    if (c.n_quantiles > 0) {  // i-IQN heads: V = 3K virtual nets x N fraction blocks (iqn_kernels.h)
        IDQN_REQUIRE(h->planes, "i-IQN heads run on the plane conv path (IDQN_CONV=bf16x3)");
        IDQN_REQUIRE(h->J % 256 == 0, "i-IQN heads: dense width %d must be a multiple of 256", h->J);
        IqnWs& w = h->iqn;
        w.N = c.n_quantiles; w.V = 3 * K;
        w.off_we = h->L.leaves[10].offset; w.off_be = h->L.leaves[11].offset;
        const long VN = (long)w.V * w.N, KN = (long)K * w.N;
        IDQN_HIP_CHECK(hipMalloc((void**)&w.wbase_v, sizeof(float*) * w.V));
        h->owned.push_back((void*)w.wbase_v);
        std::vector<const float*> wv(w.V);
        for (int k = 0; k < K; ++k) {
            wv[k] = h->online + (long)k * h->L.head_stride;
            wv[K + k] = wv[2 * K + k] = h->target + (long)k * h->L.head_stride;
        }
        IDQN_HIP_CHECK(hipMemcpy(w.wbase_v, wv.data(), sizeof(float*) * w.V, hipMemcpyHostToDevice));
        {
            float* tmp = nullptr;  // (alloc_zero counts floats: 12 x 512 bf16 = 3072 floats per block)
            if ((rc = alloc_zero(&tmp, VN * 3072, h, "iqn_cosp"))) return rc;
            w.cosp = (unsigned short*)tmp;
            if ((rc = alloc_zero(&tmp, VN * 3072, h, "iqn_cosa"))) return rc;
            w.cosa = (unsigned short*)tmp;
            if ((rc = alloc_zero(&tmp, 2L * K * (h->F / 32) * 3072, h, "iqn_wep"))) return rc;
            w.wep = (unsigned short*)tmp;
        }
        if ((rc = alloc_zero(&w.xq, VN * h->F * 32, h, "iqn_x"))) return rc;
        if ((rc = alloc_zero(&w.part, VN * w.NS * h->J * 32, h, "iqn_part"))) return rc;
        if ((rc = alloc_zero(&w.hbuf, VN * h->J * 32, h, "iqn_h"))) return rc;
        if ((rc = alloc_zero(&w.qpart, VN * (h->J / 32) * 32 * 32, h, "iqn_qpart"))) return rc;
        if ((rc = alloc_zero(&w.z, VN * c.n_actions * 32, h, "iqn_z"))) return rc;
        if ((rc = alloc_zero(&w.dq, KN * c.n_actions * 32, h, "iqn_dq"))) return rc;
        if ((rc = alloc_zero(&w.dh, KN * h->J * 32, h, "iqn_dh"))) return rc;
        if ((rc = alloc_zero(&w.dx, KN * h->F * 32, h, "iqn_dx"))) return rc;
        // the embedding backward deals the fractions of a feature tile to QG workgroups (a divisor of N)
        w.QG = 4;
        while (w.QG > 1 && w.N % w.QG != 0) --w.QG;
        if ((rc = alloc_zero(&w.dpsi, (long)w.QG * K * h->F * 32, h, "iqn_dpsi"))) return rc;
        if ((rc = alloc_zero(&w.gpart, (long)w.QG * K * 65 * h->F, h, "iqn_gpart"))) return rc;
        w.HG = 8;
        while (w.HG > 1 && w.N % w.HG != 0) --w.HG;
        if ((rc = alloc_zero(&w.hpart, (long)w.HG * K * ((long)h->J * c.n_actions + h->J + c.n_actions), h, "iqn_hpart"))) return rc;
        if (debug_env("IDQN_IQN_CLOCK") && (rc = alloc_zero(&w.clk, (1024 + 256 * 8 * 2 + 512) * 2, h, "iqn_clk"))) return rc;
        if (w.N % 16 == 0 && (rc = alloc_zero(&w.g1, (long)K * h->F * h->J, h, "iqn_g1"))) return rc;
        {
            float* gw = nullptr;
            if ((rc = alloc_zero(&gw, 2L * K * cdiv(h->F, 256) + 64, h, "iqn_gate"))) return rc;
            w.gate = reinterpret_cast<unsigned*>(gw);
        }
        if (w.N % 16 == 0 && h->J % 256 == 0 && (rc = plan_iqn_bwd_order(h, debug_int("IDQN_IQN_BWD_EARLY", -1)))) return rc;
        if ((rc = alloc_zero(&w.dbg, (long)K * (2 * w.N + 33) * 32, h, "iqn_dbg"))) return rc;
    }
    h->dominant = "k_dense0_wgrad";
    return IDQN_OK;
}

// Dispatch order of k_iqn_d0_bwd_adam (iqn_gemm.h).  A group = (head, 256 rows of Dense_0/kernel) has nbg data-gradient items, which read
// those rows, and n_jh weight-gradient items, which overwrite them in their epilogue and are ~2 x as long (N = 32).  Rules: a group's
// items sit on ONE XCD (workgroup b runs on XCD b % 8; its slot there is b / 8) with the data-gradient items at EARLIER slots than
// the weight-gradient items -- they are dispatched first and wait for nothing, so the gate of the epilogue cannot block progress.
// Within that, the order decides the launch time: the last item of every valid order is a long one, and with each group's items
// back to back the long items start as late as 0.7 of the launch (513 us measured where 408 us of work per CU were queued:
// profiles/r6_iiqn_adam_fuse.txt).  Candidates `e`: the first e groups of the XCD back to back (their long items start in the
// first round), then every other data-gradient item, then the remaining long items together as the last round.  The candidate
// with the smallest list-scheduled makespan on the XCD's 32 CUs, averaged over +- 8 % of the long items' length, is taken.
static double iqn_bwd_makespan(const std::vector<char>& seq, double dw, int ncu) {
    std::priority_queue<double, std::vector<double>, std::greater<double>> cus;
    for (int i = 0; i < ncu; ++i) cus.push(0.0);
    double end = 0;
    for (char it : seq) {
        const double e = cus.top() + (it ? dw : 1.0);
        cus.pop();
        cus.push(e);
        end = std::max(end, e);
    }
    return end;
}
static std::vector<char> iqn_bwd_seq(int n, int e, int nbg, int n_jh) {  // 0 = data-gradient item, 1 = weight-gradient item
    std::vector<char> s;
    for (int g = 0; g < e; ++g) { s.insert(s.end(), nbg, 0); s.insert(s.end(), n_jh, 1); }
    s.insert(s.end(), (size_t)nbg * (n - e), 0);
    s.insert(s.end(), (size_t)n_jh * (n - e), 1);
    return s;
}
int plan_iqn_bwd_order(idqn_handle_s* h, int forced_e) {
    IqnWs& w = h->iqn;
    const int K = h->cfg.n_heads, nbg = w.N / 8, n_jh = h->J / 256, G = K * cdiv(h->F, 256), per = nbg + n_jh;
    // lengths in units of a data-gradient item: 8 blocks x 256 rows x J against 256 x 256 x (32 N) products (measured 141 : 84 us at N = 32, J = 512) + epilogue
    const double dw = 1.68 * (w.N / 32.0) * (512.0 / h->J) + 0.33;
    const int S = cdiv(G, 8) * per;
    std::vector<int32_t> items((size_t)S * 8, -1);
    for (int x = 0; x < 8; ++x) {
        const int n = (G - x + 7) / 8;  // groups x, x + 8, ...
        int best_e = n;
        double best = 0;
        for (int e = 0; e <= n; ++e) {
            const std::vector<char> seq = iqn_bwd_seq(n, e, nbg, n_jh);
            double sum = 0;
            for (double sc : {0.92, 0.96, 1.0, 1.04, 1.08}) sum += iqn_bwd_makespan(seq, dw * sc, std::max(1, h->n_cus / 8));
            if (e == 0 || sum < best) { best = sum; best_e = e; }
        }
        if (forced_e >= 0) best_e = std::min(forced_e, n);
        if (debug_on("IDQN_PLAN_PRINT") && x == 0) fprintf(stderr, "[plan] iqn dense0 gradients: %d groups on XCD 0, %d of them back to back\n", n, best_e);
        int s = 0;
        auto put = [&](int g, int within) { items[(size_t)(s++) * 8 + x] = ((x + 8 * g) << 8) | within; };
        for (int g = 0; g < best_e; ++g)
            for (int i = 0; i < per; ++i) put(g, i);
        for (int g = best_e; g < n; ++g)
            for (int i = 0; i < nbg; ++i) put(g, i);
        for (int g = best_e; g < n; ++g)
            for (int i = nbg; i < per; ++i) put(g, i);
    }
    if (w.bwd_items) (void)hipFree(w.bwd_items);
    IDQN_HIP_CHECK(hipMalloc((void**)&w.bwd_items, items.size() * 4));
    IDQN_HIP_CHECK(hipMemcpy(w.bwd_items, items.data(), items.size() * 4, hipMemcpyHostToDevice));
    w.bwd_blocks = (int)items.size();
    return IDQN_OK;
}

// ---- general-shape cnn path (gcnn_kernels.h) -------------------------------------------------------------------------
int gcnn_setup(idqn_handle_s* h) {
    const idqn_config_t& c = h->cfg;
    GcnnWs& g = h->gc;
    g.on = true;
    g.Bmax = c.max_batch;
    int ih = c.obs_h, iw = c.obs_w, ci = c.obs_c, lo, hi;
    for (int i = 0; i < 3; ++i) {
        ConvL& l = h->conv[i];
        l.K = KS[i][0]; l.S = KS[i][1]; l.CI = ci; l.CO = c.features[i]; l.IH = ih; l.IW = iw;
        same_pad(ih, l.K, l.S, &l.OH, &l.PLh, &hi);
        same_pad(iw, l.K, l.S, &l.OW, &l.PLw, &hi);
        (void)lo;
        l.w_off = h->L.leaves[2 * i].offset; l.b_off = h->L.leaves[2 * i + 1].offset;
        ih = l.OH; iw = l.OW; ci = l.CO;
    }
    const int K = c.n_heads;
    h->F = ih * iw * ci;
    FcNet& n = h->fc;  // the dense head over the flattened features
    n.L = c.n_features - 3 + 1;
    IDQN_REQUIRE(n.L <= FC_MAX_LAYERS, "cnn: too many dense layers (%d)", n.L);
    n.d[0] = h->F;
    n.dmax = n.d[0];
    for (int l = 0; l < n.L; ++l) {
        n.d[l + 1] = 3 + l < c.n_features ? c.features[3 + l] : c.n_actions;
        n.dmax = std::max(n.dmax, n.d[l + 1]);
        n.w_off[l] = h->L.leaves[6 + 2 * l].offset;
        n.b_off[l] = h->L.leaves[7 + 2 * l].offset;
    }
    int rc;
    const long B = g.Bmax;
    for (int i = 0; i < 3; ++i) {
        const ConvL& l = h->conv[i];
        const long per = (long)l.OH * l.OW * l.CO;
        char nm[16];
        snprintf(nm, sizeof nm, "g_act%d", i);
        if ((rc = alloc_zero(&g.act[i], 2L * K * B * per, h, nm))) return rc;
        snprintf(nm, sizeof nm, "g_dact%d", i);
        if ((rc = alloc_zero(&g.dact[i], (long)K * B * per, h, nm))) return rc;
        snprintf(nm, sizeof nm, "g_iact%d", i);
        if ((rc = alloc_zero(&g.inf_act[i], 32L * per, h, nm))) return rc;
    }
    if ((rc = alloc_zero(&h->fc_ws, K * ((long)(n.L + 3) * B * n.dmax + 2 * B), h, "fc_ws"))) return rc;
    if ((rc = alloc_zero(&g.inf_ws, 2L * 32 * n.dmax, h, "g_infws"))) return rc;
    if ((rc = alloc_zero(&h->qdbg, 2L * K * B * c.n_actions, h, "q"))) return rc;
    // parameter pointers of the 2K training nets (online first), as the conv kernels index them
    IDQN_HIP_CHECK(hipMalloc((void**)&h->train.wbase, sizeof(float*) * 2 * K));
    h->owned.push_back((void*)h->train.wbase);
    std::vector<const float*> wb(2 * K);
    for (int k = 0; k < K; ++k) {
        wb[k] = h->online + (long)k * h->L.head_stride;
        wb[K + k] = h->target + (long)k * h->L.head_stride;
    }
    IDQN_HIP_CHECK(hipMemcpy(h->train.wbase, wb.data(), sizeof(float*) * 2 * K, hipMemcpyHostToDevice));
    h->dominant = "k_fc_step";
    return IDQN_OK;
}

GConvArgs gconv_args(idqn_handle_s* h, int layer, const float* const* wbase, int n_nets, int n_split, int B, const uint8_t* s0,
                     const uint8_t* s1, const float* in, float* out) {
    const ConvL& l = h->conv[layer];
    GConvArgs a;
    a.in_u8[0] = s0; a.in_u8[1] = s1; a.in = in; a.out = out; a.wbase = wbase; a.w_off = l.w_off; a.b_off = l.b_off;
    a.n_nets = n_nets; a.n_split = n_split; a.B = B; a.IH = l.IH; a.IW = l.IW; a.CI = l.CI; a.OH = l.OH; a.OW = l.OW; a.CO = l.CO;
    a.KS = l.K; a.S = l.S; a.PLh = l.PLh; a.PLw = l.PLw;
    return a;
}

unsigned ggrid(long n) { return (unsigned)std::max(1L, std::min((n + 255) / 256, 65536L)); }

// conv trunk of `n_nets` nets (the first n_split read s0, the rest s1) into act[0..2]
int gcnn_trunk(idqn_handle_s* h, const float* const* wbase, int n_nets, int n_split, int B, const uint8_t* s0, const uint8_t* s1,
               float* const act[3], hipStream_t q) {
    for (int i = 0; i < 3; ++i) {
        const GConvArgs a = gconv_args(h, i, wbase, n_nets, n_split, B, s0, s1, i == 0 ? nullptr : act[i - 1], act[i]);
        hipLaunchKernelGGL(k_gconv_fwd, dim3(ggrid((long)n_nets * B * a.OH * a.OW * a.CO)), dim3(256), 0, q, a);
    }
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

int gcnn_learn(idqn_handle_s* h, const uint8_t* st, const uint8_t* st2, const int32_t* action, const float* reward,
               const uint8_t* terminal, int B, int Bdiv, bool grads_only, bool profile, hipStream_t q) {
    GcnnWs& g = h->gc;
    const int K = h->cfg.n_heads;
    IDQN_REQUIRE(B <= g.Bmax, "batch %d exceeds the workspace (%d)", B, g.Bmax);
    int rc = gcnn_trunk(h, h->train.wbase, 2 * K, K, B, st, st2, g.act, q);
    if (rc) return rc;
    FcArgs a;
    a.net = h->fc; a.online = h->online; a.target = h->target; a.grad = h->grad; a.P = h->L.head_stride;
    a.s = g.act[2]; a.s2 = g.act[2] + (long)K * B * h->F; a.s_stride = (long)B * h->F; a.din = g.dact[2];
    a.action = action; a.reward = reward; a.terminal = terminal; a.gamma_n = h->gamma_n; a.B = B; a.Bdiv = Bdiv; a.K = K;
    a.ws = h->fc_ws; a.losses = h->losses; a.q_dbg = h->qdbg;
    a.count = h->count; a.bcinv = h->bcinv; a.adam_b1 = h->ad.b1; a.adam_b2 = h->ad.b2;
    a.cum = h->cum; a.finish_step = grads_only ? 0 : 1;
    a.is_weight = h->is_weight; a.td_abs = h->td_abs;
    a.gm = GradMap{h->gP, h->g_w0_begin, h->g_w0_end, h->g_w0_base};
    if (profile && h->ev_used + 2 <= (int)h->ev.size()) IDQN_HIP_CHECK(hipEventRecord(h->ev[h->ev_used], q));
    hipLaunchKernelGGL(k_fc_step, dim3(K), dim3(256), 0, q, a);
    if (profile && h->ev_used + 2 <= (int)h->ev.size()) {
        IDQN_HIP_CHECK(hipEventRecord(h->ev[h->ev_used + 1], q));
        h->ev_used += 2;
    }
    // conv backward, top down: weight gradient of layer i from dact[i], then dact[i - 1] = (W^T dact[i]) * [act[i - 1] > 0]
    for (int i = 2; i >= 0; --i) {
        GConvBwdArgs b;
        b.f = gconv_args(h, i, h->train.wbase, K, K, B, st, st, i == 0 ? nullptr : g.act[i - 1], nullptr);
        b.dy = g.dact[i]; b.din = i > 0 ? g.dact[i - 1] : nullptr; b.grad = h->grad;
        b.gm = GradMap{h->gP, h->g_w0_begin, h->g_w0_end, h->g_w0_base};
        const long nw = (long)b.f.KS * b.f.KS * b.f.CI * b.f.CO + b.f.CO;
        hipLaunchKernelGGL(k_gconv_wgrad, dim3(ggrid(nw), K), dim3(256), 0, q, b);
        if (i > 0) hipLaunchKernelGGL(k_gconv_dgrad, dim3(ggrid((long)K * B * b.f.IH * b.f.IW * b.f.CI)), dim3(256), 0, q, b);
    }
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

int fc_setup(idqn_handle_s* h) {
    const idqn_config_t& c = h->cfg;
    FcNet& n = h->fc;
    n.L = c.n_features + 1;
    n.d[0] = c.obs_h * c.obs_w * c.obs_c;
    n.dmax = n.d[0];
    for (int l = 0; l < n.L; ++l) {
        n.d[l + 1] = l < c.n_features ? c.features[l] : c.n_actions;
        if (n.d[l + 1] > n.dmax) n.dmax = n.d[l + 1];
        n.w_off[l] = h->L.leaves[2 * l].offset;
        n.b_off[l] = h->L.leaves[2 * l + 1].offset;
    }
    const long B = c.max_batch, K = c.n_heads;
    int rc;
    if ((rc = alloc_zero(&h->fc_ws, K * ((long)(n.L + 3) * B * n.dmax + 2 * B) + 2 * 32 * n.dmax, h, "fc_ws"))) return rc;
    if ((rc = alloc_zero(&h->qdbg, 2 * K * B * c.n_actions, h, "q"))) return rc;
    h->fc_plan_ = fc_plan(n);
    if (n.dmax > FC_MAX_WIDTH) h->fc_plan_.BS = 0;  // wider layers: the generic kernel (any width)
    if (h->fc_plan_.BS) {
        const int bytes = (int)(h->fc_plan_.floats * 4);
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_fc_step_lds<32>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_fc_step_lds<16>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_fc_step_lds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    h->fcp_plan_ = fc_par_plan(n, h->L.head_stride);
    // IDQN_FC_PAR=0: the two-launch path (k_fc_step_mfma / k_fc_step_lds + k_adam) for every batch size
    if ((getenv("IDQN_FC_PAR") && atoi(getenv("IDQN_FC_PAR")) == 0) || n.dmax > FC_MAX_WIDTH) h->fcp_plan_.floats = 0;
    if (h->fcp_plan_.floats)
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_fc_step_par, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(h->fcp_plan_.floats * 4)));
    h->fcm_plan_ = fc_mfma_plan(n);
    if (n.dmax > FC_MAX_WIDTH) h->fcm_plan_.floats = 0;
    h->fcm_global_ = false;
    if (!h->fcm_plan_.floats && n.dmax <= FC_MAX_WIDTH) {
        // the matrix does not fit LDS beside the activations: the same kernel with the weight operand read from global memory
        h->fcm_plan_ = fc_mfma_plan_g(n);
        h->fcm_global_ = h->fcm_plan_.floats != 0;
    }
    if (h->fcm_plan_.floats) {
        if (h->fcm_global_)
            IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_fc_step_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(h->fcm_plan_.floats * 4)));
        else
            IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_fc_step_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(h->fcm_plan_.floats * 4)));
    }
    h->dominant = "k_fc_step";
    return IDQN_OK;
}

// Conv_1 / Conv_2 kernels re-indexed for the data gradient-as-convolution (k_wt_build), f32
int wt_build_args(idqn_handle_s* h, WtBuildArgs& wb) {  // returns the x extent of the launch
    wb.wbase = h->train.wbase; wb.K = h->cfg.n_heads;
    long maxe = 0;
    for (int i = 1; i <= 2; ++i) {
        const ConvL& l = h->conv[i];
        WtLayer& w = wb.layer[i - 1];
        w.wt = h->wt[i]; w.w_off = l.w_off; w.wt_stride = h->wt_stride[i];
        w.KH = l.K; w.KW = l.K; w.CI = l.CI; w.CO = l.CO; w.S = l.S; w.PLh = l.PLh; w.PLw = l.PLw;
        w.n_var = l.S * l.S; w.KHs = l.K / l.S; w.KWs = l.K / l.S;
        maxe = std::max(maxe, (long)l.K * l.K * l.CI * l.CO);
    }
    return cdiv(maxe, 256);
}

int build_dgrad_weights(idqn_handle_s* h, hipStream_t q) {
    if (h->wt_ready) return IDQN_OK;  // already built this step, inside the TD / loss launch
    WtBuildArgs wb;
    const int nx = wt_build_args(h, wb);
    hipLaunchKernelGGL(k_wt_build, dim3(nx, wb.K, 2), dim3(256), 0, q, wb);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

// ---- plane conv path: launch plans -----------------------------------------------------------------------------
// role 0..2: forward of Conv_0..2; 3: data gradient of Conv_2; 4: data gradient of Conv_1
struct RoleGeom {
    int NPA, CT, NQ, NCC, S, SX, KH, n_var;
    CVar var[4];
};

int role_geom(idqn_handle_s* h, int role, RoleGeom& g) {
    memset(&g, 0, sizeof(g));
    if (role <= 2) {
        const ConvL& l = h->conv[role];
        g.NPA = role == 0 ? 1 : 3; g.CT = l.CO / 32; g.S = l.S; g.KH = l.K; g.n_var = 1;
        if (role == 0) {  // (kw, c) rows of a kernel row in 16-row chunks of 4 pixels x 4 channels
            IDQN_REQUIRE(l.CI == 4 && l.K == 8 && l.S == 4, "plane conv: Conv_0 is built for 8x8 stride 4 over 4 channels");
            g.NQ = l.K * l.CI / 16; g.NCC = 1; g.SX = 1;
        } else {
            g.NQ = l.K; g.NCC = l.CI / 16; g.SX = l.S;
        }
        CVar& v = g.var[0];
        v.w_off = h->wq_fwd[role]; v.in_off_h = 0; v.in_off_w = 0; v.OH = l.OH; v.OW = l.OW;
        v.out_mul = 1; v.out_add_h = 0; v.out_add_w = 0;
    } else {
        const int i = role == 3 ? 2 : 1;
        const ConvL& l = h->conv[i];
        const ActGeom& gd = i == 2 ? h->gda3 : h->gda2;
        const int KHs = l.K / l.S;
        g.NPA = 3; g.CT = l.CI / 32; g.NQ = KHs; g.NCC = l.CO / 16; g.S = 1; g.SX = 1; g.KH = KHs; g.n_var = l.S * l.S;
        IDQN_REQUIRE(g.n_var <= 4, "plane conv: stride %d data gradient has more than 4 parities", l.S);
        for (int vi = 0; vi < g.n_var; ++vi) {
            const int rh = vi / l.S, rw = vi % l.S;
            const int ph = (rh + l.PLh) % l.S, pw = (rw + l.PLw) % l.S;
            CVar& v = g.var[vi];
            v.w_off = h->wq_dg[i] + (long)vi * KHs * KHs * l.CO * l.CI * 6;
            v.in_off_h = (rh + l.PLh - ph) / l.S + gd.lo_h - KHs + 1;
            v.in_off_w = (rw + l.PLw - pw) / l.S + gd.lo_w - KHs + 1;
            v.OH = (l.IH - rh + l.S - 1) / l.S; v.OW = (l.IW - rw + l.S - 1) / l.S;
            v.out_mul = l.S; v.out_add_h = rh; v.out_add_w = rw;
            IDQN_REQUIRE(v.in_off_h >= 0 && v.in_off_w >= 0 && v.OH > 0 && v.OW > 0, "conv %d dgrad: bad variant geometry", i);
        }
    }
    return IDQN_OK;
}

// LDS bytes of one stage for positions [p0, p0 + np) of a variant with OW columns
long fwd_stage_bytes(const RoleGeom& g, int OW, int p0, int np, int* n_rows) {
    long bytes = (long)g.NQ * g.CT * 3 * 1024;
    const int oh0 = p0 / OW, oh1 = (p0 + np - 1) / OW;
    for (int row = oh0; row <= oh1; ++row) {
        const int lo = std::max(p0, row * OW), hi = std::min(p0 + np, (row + 1) * OW);
        bytes += (long)((hi - lo - 1) * g.SX + g.NQ) * g.NPA * 1024;
    }
    *n_rows = oh1 - oh0 + 1;
    return bytes;
}

// one candidate plan: about `target` workgroups per launch, `budget` bytes of LDS each (160 KB: one per CU, 80 KB: two)
int plan_fwd_target(int role, int n_nets, int nb, const RoleGeom& g, int target, size_t budget, FwdPlan& pl) {
    const int nt_max = convp_fwd_max_nt(g.CT);
    long npos_total = 0;
    for (int v = 0; v < g.n_var; ++v) npos_total += (long)g.var[v].OH * g.var[v].OW;
    const int per_slot = std::max(1, target / (n_nets * nb));
    std::vector<CItem> items;
    int np_all = 0;
    long stage_max = 0;
    // ranges per variant: proportional to its positions, rounded DOWN (one workgroup more than CUs would run alone after
    // all the others), leftovers to the variants with the most positions per range
    int Rv[4] = {1, 1, 1, 1}, used = 0;
    for (int v = 0; v < g.n_var; ++v) {
        const long npos = (long)g.var[v].OH * g.var[v].OW;
        Rv[v] = (int)std::max(1L, std::min(npos, per_slot * npos / npos_total));
        used += Rv[v];
    }
    for (; used < per_slot; ++used) {
        int best = -1;
        double worst = 0;
        for (int v = 0; v < g.n_var; ++v) {
            const double load = (double)g.var[v].OH * g.var[v].OW / Rv[v];
            if (Rv[v] < g.var[v].OH * g.var[v].OW && load > worst) { worst = load; best = v; }
        }
        if (best < 0) break;
        ++Rv[best];
    }
    for (int v = 0; v < g.n_var; ++v) {
        const int OW = g.var[v].OW, npos = g.var[v].OH * OW;
        int R = Rv[v];
        // limits: tiles per wave, rows spanned, LDS (two stages in 160 KB)
        for (;; ++R) {
            const int np = (npos + R - 1) / R;
            bool ok = (np * g.CT + 3) / 4 <= nt_max;
            const long masks = role >= 3 ? 4L * ((np * g.CT + 3) / 4) * 2048 : 0;  // the data gradient's mask tiles
            for (int r = 0, p0 = 0; ok && r < R; ++r) {
                const int n = npos / R + (r < npos % R ? 1 : 0);
                int rows;
                if (n > 0 && (2 * fwd_stage_bytes(g, OW, p0, n, &rows) + masks > (long)budget - 8 * 1024 || rows > CP_MAX_STRIPS)) ok = false;
                p0 += n;
            }
            if (ok || R >= npos) break;
        }
        Rv[v] = R;
        for (int n = 0; n < n_nets; ++n)
            for (int bb = 0; bb < nb; ++bb)
                for (int r = 0, p0 = 0; r < R; ++r) {
                    const int cnt = npos / R + (r < npos % R ? 1 : 0);
                    if (cnt == 0) continue;
                    CItem it;
                    memset(&it, 0, sizeof(it));
                    it.net = n; it.bb = bb; it.var = v; it.p0 = p0; it.np = cnt;
                    items.push_back(it);
                    int rows;
                    stage_max = std::max(stage_max, fwd_stage_bytes(g, OW, p0, cnt, &rows));
                    IDQN_REQUIRE(rows <= CP_MAX_STRIPS, "plane conv: %d positions span %d rows", cnt, rows);
                    np_all = std::max(np_all, cnt);
                    p0 += cnt;
                }
    }
    // The kernel derives item b arithmetically: slot = b / items_per_slot (net-major: consecutive workgroups, which the
    // XCD-contiguous remap keeps on one XCD, share a net's weights), then variant and balanced range inside the slot.
    pl = FwdPlan();
    pl.n_items = (int)items.size();
    for (int v = 0; v < g.n_var; ++v) {
        pl.r_begin[v] = pl.items_per_slot;
        pl.r_cnt[v] = Rv[v];
        pl.items_per_slot += Rv[v];
    }
    IDQN_REQUIRE(pl.n_items == pl.items_per_slot * n_nets * nb, "plane conv: role %d item count %d does not match its ranges", role, pl.n_items);
    pl.NT = (np_all * g.CT + 3) / 4;
    IDQN_REQUIRE(pl.NT >= 1 && pl.NT <= nt_max, "plane conv: role %d needs %d tiles per wave", role, pl.NT);
    pl.stage = (size_t)stage_max;
    pl.ring = convp_fwd_ring(pl.stage, pl.NT, role <= 2 ? 0 : 1, budget);
    pl.lds = convp_fwd_lds(pl.stage, pl.NT, role <= 2 ? 0 : 1, pl.ring, role != 2, role == 2);
    IDQN_REQUIRE(pl.lds <= budget, "plane conv: role %d needs %zu bytes of LDS", role, pl.lds);
    return IDQN_OK;
}

// The launch plan of a role: one workgroup per CU, all co-resident, equal work.  IDQN_CONV_WGS=512 plans two smaller
// workgroups per CU instead (80 KB of LDS each), so that one's prologue / first fill / epilogue overlaps the other's main
// loop -- measured: Conv_0 forward 18.8 -> 18.5 us, Conv_1 data gradient 16.9 -> 21.7 us (more weight re-staging per
// MFMA, a two-deep ring); not the default.
int plan_fwd(idqn_handle_s* h, int role, int n_nets, int nb, const RoleGeom& g, FwdPlan** out, int target) {
    auto key = std::make_tuple(role, n_nets, nb, target);
    auto itp = h->fwd_plans.find(key);
    if (itp != h->fwd_plans.end()) { *out = &itp->second; return IDQN_OK; }
    FwdPlan pl;
    int rc = plan_fwd_target(role & 7, n_nets, nb, g, target, 160 * 1024, pl);
    if (rc) return rc;
    // A whole-chip forward launch whose workgroups would carry the same number of tiles per wave with ~18 % fewer of them
    // takes the smaller grid: the matrix loop is as long, the staging traffic and the fill burst are smaller (Conv_1 / Conv_2
    // forward at K = 5: 250 -> 210 workgroups, 24.5 -> 23.7 and 24.1 -> 23.2 us, profiles/r3_conv_cu_budget_sweep.txt;
    // Conv_0 would need 6 tiles instead of 5 and keeps 250).
    if (target == cu_budget() && (role & 7) <= 2) {
        FwdPlan p2;
        if (plan_fwd_target(role & 7, n_nets, nb, g, target * 13 / 16, 160 * 1024, p2) == IDQN_OK && p2.NT == pl.NT && p2.n_items < pl.n_items)
            pl = p2;
    }
    if (plan_print())
        fprintf(stderr, "[plan] fwd role %d nets %d nb %d target %d: %d workgroups, NT %d, ring %d, stage %zu B, lds %zu B, supersteps %d, "
                "ranges/slot %d (NPA %d CT %d NQ %d)\n", role, n_nets, nb, target, pl.n_items, pl.NT, pl.ring, pl.stage, pl.lds,
                g.KH * g.NCC, pl.items_per_slot, g.NPA, g.CT, g.NQ);
    *out = &(h->fwd_plans[key] = pl);
    return IDQN_OK;
}


// The persistent form of a forward / data-gradient role (convp_pp.hip) for launches with several items per CU: every
// output row of every variant is cut into `parts` column ranges (items never cross rows: one strip per stage), as few as
// the tile count per wave (<= convp_pp_max_nt) and the LDS (ring x stage + the epilogue slots <= 160 KB) allow.  Candidates
// are ranked by a per-item cost model -- supersteps x max(matrix cycles + ramp, staged bytes / fill rate), the rates read
// off profiles/r6_cprof_b256.txt -- summed over the items and divided over the CUs; IDQN_PP_PARTS<role> (debug build)
// overrides the choice for sweeps.  Returns IDQN_E_INVALID when no candidate fits (the caller keeps the one-item launch).
int plan_fwd_pp(idqn_handle_s* h, int role, int n_nets, int nb, const RoleGeom& g, FwdPlan** out) {
    auto key = std::make_tuple(role + 16, n_nets, nb, 0);
    (void)0;
    auto itp = h->fwd_plans.find(key);
    if (itp != h->fwd_plans.end()) { *out = &itp->second; return itp->second.n_items > 0 ? IDQN_OK : IDQN_E_INVALID; }
    char nm[32];
    role &= 7;  // (bit 3 marks the acting set's plans)
    snprintf(nm, sizeof nm, "IDQN_PP_PARTS%d", role);
    const int forced = debug_int(nm, 0), forced_ring = debug_int("IDQN_PP_RING", 0);
    const int NSS = g.KH * g.NCC, nm_prod = g.NPA == 3 ? 6 : 3, cus = cu_budget();
    const bool epi1 = role >= 3, planes_out = role != 2, f32_out = role == 2;
    int ow_max = 0;
    for (int v = 0; v < g.n_var; ++v) ow_max = std::max(ow_max, g.var[v].OW);
    FwdPlan best;
    double best_cost = 0;
    for (int parts = 1; parts <= ow_max; ++parts) {
        if (forced && parts != forced) continue;
        FwdPlan pl;
        int np_max = 0;
        double work = 0;  // cycles of all items of one slot
        bool ok = true;
        pl.row_parts = parts;
        for (int v = 0; v < g.n_var && ok; ++v) {
            const int OW = g.var[v].OW, OH = g.var[v].OH;
            if (parts > OW) { ok = false; break; }
            np_max = std::max(np_max, (OW + parts - 1) / parts);
            pl.r_begin[v] = pl.items_per_slot;
            pl.r_cnt[v] = OH * parts;
            pl.items_per_slot += OH * parts;
        }
        if (!ok) continue;
        pl.NT = (np_max * g.CT + 3) / 4;
        if (pl.NT < 2) pl.NT = 2;  // (the smallest kernel built)
        if (!convp_pp_built(g.NPA, g.CT, g.NQ, pl.NT) || NSS - 1 < pl.NT) continue;
        pl.stage = (size_t)(g.NQ * g.CT * 3 + ((np_max - 1) * g.SX + g.NQ) * g.NPA) * 1024;
        const size_t epi = convp_pp_epi_bytes(pl.NT, epi1 ? 1 : 0, planes_out, f32_out);
        pl.ring = (forced_ring != 2 && 3 * pl.stage + epi <= 160 * 1024) ? 3 : 2;
        pl.lds = pl.ring * pl.stage + epi;
        if (pl.lds > 160 * 1024) continue;
        for (int v = 0; v < g.n_var; ++v) {
            const int OW = g.var[v].OW, OH = g.var[v].OH;
            for (int part = 0; part < parts; ++part) {
                const int np = OW / parts + (part < OW % parts ? 1 : 0);
                const double mfma = (double)((np * g.CT + 3) / 4) * g.NQ * nm_prod * 32 + 450;
                const double fill = (double)(g.NQ * g.CT * 3 + ((np - 1) * g.SX + g.NQ) * g.NPA) * 1024 / (pl.ring == 3 ? 30.0 : 26.0);
                work += OH * (NSS * std::max(mfma, fill) + 600);
            }
        }
        pl.n_items = pl.items_per_slot * n_nets * nb;
        pl.n_wg = std::min(pl.n_items, cus);
        const int rounds = (pl.n_items + pl.n_wg - 1) / pl.n_wg;
        // whole launch: the slots' work over the CUs, stretched by the last round's idle CUs
        const double cost = work * n_nets * nb / pl.n_wg * ((double)rounds * pl.n_wg / pl.n_items);
        if (plan_print())
            fprintf(stderr, "[plan] pp role %d parts %d: %d items, NT %d, ring %d, stage %zu B, lds %zu B, model %.0f k cycles\n", role, parts,
                    pl.n_items, pl.NT, pl.ring, pl.stage, pl.lds, cost / 1e3);
        if (best.n_items == 0 || cost < best_cost) { best = pl; best_cost = cost; }
    }
    if (best.n_items > 0) {
        // the item table, in launch order: index b -> (slot, range) as the one-item kernels derive it (net-major, or range-major
        // where the nets share their input), range -> (variant, output row, column part)
        std::vector<CItem> tab((size_t)best.n_items);
        const int n_slots = n_nets * nb;
        for (int b = 0; b < best.n_items; ++b) {
            const bool range_major = role == 0;
            const int rr = range_major ? b / n_slots : b % best.items_per_slot, slot = range_major ? b % n_slots : b / best.items_per_slot;
            int vi = 0;
            for (int i = 1; i < g.n_var; ++i) if (rr >= best.r_begin[i]) vi = i;
            const int r = rr - best.r_begin[vi], OW = g.var[vi].OW;
            const int row = r / best.row_parts, part = r % best.row_parts, base = OW / best.row_parts, rem = OW % best.row_parts;
            CItem& it = tab[b];
            memset(&it, 0, sizeof(it));
            it.net = slot / nb; it.bb = slot % nb; it.var = vi;
            it.pad0 = row; it.pad1 = part * base + std::min(part, rem);
            it.p0 = row * OW + it.pad1; it.np = base + (part < rem ? 1 : 0);
        }
        IDQN_HIP_CHECK(hipMalloc((void**)&best.items_dev, tab.size() * sizeof(CItem)));
        h->owned.push_back((void*)best.items_dev);
        IDQN_HIP_CHECK(hipMemcpy(best.items_dev, tab.data(), tab.size() * sizeof(CItem), hipMemcpyHostToDevice));
    }
    *out = &(h->fwd_plans[key] = best);
    if (best.n_items == 0) return IDQN_E_INVALID;
    if (plan_print())
        fprintf(stderr, "[plan] pp role %d nets %d nb %d: parts %d -> %d items on %d workgroups, NT %d, ring %d, stage %zu B, lds %zu B\n", role, n_nets,
                nb, best.row_parts, best.n_items, best.n_wg, best.NT, best.ring, best.stage, best.lds);
    return IDQN_OK;
}

int plan_wgrad(idqn_handle_s* h, int layer, int nb, WgradPlan** out, int n_chunks = 0) {
    if (n_chunks <= 0) n_chunks = h->npc[layer];
    auto key = std::make_tuple(layer, nb, n_chunks);
    auto itp = h->wgrad_plans.find(key);
    if (itp != h->wgrad_plans.end()) { *out = &itp->second; return IDQN_OK; }
    const ConvL& l = h->conv[layer];
    const int K = h->cfg.n_heads, npos = l.OH * l.OW, nch = std::min(n_chunks, std::min(h->npc[layer], npos));  // (the slab region holds npc chunks)
    WgradPlan pl;
    pl.n_chunks = nch;
    pl.MT = layer == 0 ? l.K : l.K * l.CI / 32;
    const long dy_pix = 3L * l.CO * 64, x_pix = (layer == 0 ? 1L : 3L) * l.CI * 64;
    for (pl.PG = layer == 0 ? 4 : 2;; --pl.PG) {  // positions per LDS stage: as many as two stages leave room for
        const long XB = layer == 0 ? (long)l.K * (pl.PG + 1) * 1024 : ((pl.PG - 1) * l.S + l.K) * x_pix;
        pl.lds = (size_t)(2 * (XB + pl.PG * dy_pix));
        if (pl.lds + 2048 <= 160 * 1024 || pl.PG == 1) break;
        if (layer == 0 && pl.PG > 2) --pl.PG;  // Conv_0: 4, 2 or 1 positions (never 0)
    }
    IDQN_REQUIRE(pl.lds + 2048 <= 160 * 1024, "plane wgrad: layer %d needs %zu bytes of LDS", layer, pl.lds);
    // item order = workgroup order (the XCD-contiguous remap gives an XCD consecutive items): head-major, so that the kernel
    // rows of one chunk share its x / dy strips through L2; Conv_0 chunk-major, because its K heads read the SAME staged
    // minibatch and a pixel strip then crosses the fabric once per XCD instead of once per head.  The kernel derives its
    // item (head, chunk, kernel row, balanced position range) from the workgroup index.
    pl.chunk_major = layer == 0 ? 1 : 0;
    pl.n_chunks = nch;
    pl.n_items = K * nch * (layer == 0 ? 1 : l.K);
    (void)nb;
    if (plan_print())
        fprintf(stderr, "[plan] wgrad layer %d: %d workgroups, %d chunks of ~%d positions, PG %d, MT %d, lds %zu B\n", layer,
                pl.n_items, nch, npos / nch, pl.PG, pl.MT, pl.lds);
    *out = &(h->wgrad_plans[key] = pl);
    return IDQN_OK;
}

// staging launch: pixels -> bf16 plane, conv kernels -> packed planes (forward kernels of every net of the set, and
// for the training set the data-gradient kernels of the K online nets)
int planes_stage(idqn_handle_s* h, NetSet& s, const uint8_t* st, const uint8_t* st2, int B, int nb, hipStream_t q) {
    StageArgs a;
    memset(&a, 0, sizeof(a));
    a.src[0] = st; a.src[1] = st2 ? st2 : st;
    a.x1 = s.x1; a.E = (long)h->gx.H * h->gx.W * h->gx.C; a.B = B; a.nb = nb; a.n_sets = s.n_in_sets;
    a.H = h->gx.H; a.W = h->gx.W; a.C = h->gx.C; a.lo_h = h->gx.lo_h; a.lo_w = h->gx.lo_w; a.Hp = h->gx.Hp; a.Wp = h->gx.Wp;
    a.n_prep_blocks = cdiv(a.E, 64) * nb * s.n_in_sets;
    a.wbase = s.wbase; a.wq = s.wq; a.wq_stride = h->wq_stride;
    const bool train = &s == &h->train;
    long blocks = 0;
    int nj = 0;
    for (int i = 0; i < 3; ++i) {
        const ConvL& l = h->conv[i];
        PackJob& j = a.job[nj++];
        j.src_off = l.w_off; j.dst_off = h->wq_fwd[i]; j.n_nets = s.n_nets;
        j.KHv = l.K; j.CT = l.CO / 32; j.mode = 0;
        if (i == 0) { j.NQ = l.K * l.CI / 16; j.NCC = 1; j.div255 = 1; } else { j.NQ = l.K; j.NCC = l.CI / 16; }
        j.KW = l.K; j.CI = l.CI; j.CO = l.CO; j.S = l.S;
        j.blocks_per_net = cdiv((long)j.KHv * j.NCC * j.NQ * j.CT * 64, 256);
        j.first_block = blocks;
        blocks += (long)j.blocks_per_net * j.n_nets;
    }
    for (int i = 2; i >= 1 && train; --i) {
        const ConvL& l = h->conv[i];
        const int KHs = l.K / l.S;
        for (int vi = 0; vi < l.S * l.S; ++vi) {
            IDQN_REQUIRE(nj < 8, "plane conv: more than 8 kernel blocks to pack");
            PackJob& j = a.job[nj++];
            j.src_off = l.w_off; j.dst_off = h->wq_dg[i] + (long)vi * KHs * KHs * l.CO * l.CI * 6; j.n_nets = h->cfg.n_heads;
            j.KHv = KHs; j.NQ = KHs; j.NCC = l.CO / 16; j.CT = l.CI / 32; j.mode = 1;
            j.KW = l.K; j.CI = l.CI; j.CO = l.CO; j.S = l.S; j.PLh = l.PLh; j.PLw = l.PLw; j.rh = vi / l.S; j.rw = vi % l.S; j.KHs = KHs;
            j.blocks_per_net = cdiv((long)j.KHv * j.NCC * j.NQ * j.CT * 64, 256);
            j.first_block = blocks;
            blocks += (long)j.blocks_per_net * j.n_nets;
        }
    }
    a.n_jobs = nj;
    if (train) { a.K = h->cfg.n_heads; a.count = h->count; a.bcinv = h->bcinv; a.b1 = h->ad.b1; a.b2 = h->ad.b2; }
    if (train && h->rp) {
        a.frames = h->rp->frames; a.rows = h->rp->rows; a.n_frames = h->rp->n_frames; a.frame_bytes = h->rp->frame_bytes;
        a.act_out = h->rp_action; a.rew_out = h->rp_reward; a.term_out = h->rp_terminal;
        return convp_launch_stage(a, a.n_prep_blocks + (int)blocks, q, &h->rp->slots);
    }
    if (debug_on("IDQN_STAGE_SKIP_PIXELS")) a.n_prep_blocks = 0;  // timing probe: the launch without its pixel half (wrong results)
    return convp_launch_stage(a, a.n_prep_blocks + (int)blocks, q);
}

// one plane conv launch: role 0..2 forward of Conv_0..2 for net set s, 3 / 4 data gradient of Conv_2 / Conv_1
// launch arguments + plan of a forward / data-gradient role, planned for about `target` workgroups
int conv_args(idqn_handle_s* h, NetSet& s, int role, int nb, int target, CFwdArgs& a, RoleGeom& g, FwdPlan*& pl) {
    int rc = role_geom(h, role, g);
    if (rc) return rc;
    const bool fwd = role <= 2;
    const int n_nets = fwd ? s.n_nets : h->cfg.n_heads;
    if ((rc = plan_fwd(h, role + (fwd && &s == &h->infer ? 8 : 0), n_nets, nb, g, &pl, target))) return rc;
    memset(&a, 0, sizeof(a));
    a.wq = s.wq;
    a.items_per_slot = pl->items_per_slot;
    for (int v = 0; v < 4; ++v) { a.r_begin[v] = pl->r_begin[v]; a.r_cnt[v] = pl->r_cnt[v]; }
    if (&s == &h->infer) { a.pbase[0] = a.pbase[1] = h->infer_pbase; a.n_first = 1; }
    else { a.pbase[0] = h->online; a.pbase[1] = h->target; a.n_first = h->cfg.n_heads; }
    a.pstride = h->L.head_stride; a.wq_stride = h->wq_stride; a.nb = nb; a.n_var = g.n_var;
    a.KH = g.KH; a.NCC = g.NCC; a.S = g.S; a.SX = g.SX;
    for (int v = 0; v < g.n_var; ++v) a.var[v] = g.var[v];
    const ActGeom* gin;   // input planes
    const ActGeom* gout;  // output planes
    if (fwd) {
        const ConvL& l = h->conv[role];
        const unsigned short* ins[3] = {s.x1, s.a1p, s.a2p};
        unsigned short* outs[3] = {s.a1p, s.a2p, nullptr};
        const ActGeom* gi[3] = {&h->gx, &h->ga1, &h->ga2};
        const ActGeom* go[3] = {&h->ga1, &h->ga2, &h->ga3};
        gin = gi[role]; gout = go[role];
        a.in = ins[role]; a.out3 = outs[role]; a.epilogue = 0; a.b_off = l.b_off; a.CO = l.CO;
        a.in_split = role == 0 ? (s.n_in_sets > 1 ? s.n_nets / 2 : s.n_nets + 1) : 0;
        a.range_major = role == 0;
        if (role == 2) { a.out_f32 = s.a3; a.f32_slot = h->ga3.block; a.f32_W = l.OW; }
        a.pix_bytes = g.NPA * l.CI * 64; a.plane_bytes = l.CI * 64;
        a.xstep = role == 0 ? 1024 : a.pix_bytes;
    } else {
        const int i = role == 3 ? 2 : 1;
        const ConvL& l = h->conv[i];
        gin = i == 2 ? &h->gda3 : &h->gda2;
        gout = i == 2 ? &h->gda2 : &h->gda1;
        const ActGeom* gm = i == 2 ? &h->ga2 : &h->ga1;
        a.in = i == 2 ? h->da3p : h->da2p;
        a.out3 = i == 2 ? h->da2p : h->da1p;
        a.mask3 = i == 2 ? s.a2p : s.a1p;
        a.pb = h->pbuf[i - 1];
        a.epilogue = 1; a.CO = l.CI; a.in_split = 0;
        a.mask_slot = gm->block * 6; a.mask_Wp = gm->Wp; a.mask_lo_h = gm->lo_h; a.mask_lo_w = gm->lo_w; a.mask_C = l.CI;
        a.pix_bytes = 3 * l.CO * 64; a.plane_bytes = l.CO * 64; a.xstep = a.pix_bytes;
    }
    a.in_slot = gin->block * (role == 0 ? 2 : 6);
    a.row_bytes = gin->Wp * a.pix_bytes;
    a.out_slot = gout->block * 6; a.out_Wp = gout->Wp; a.out_lo_h = gout->lo_h; a.out_lo_w = gout->lo_w;
    a.out_W = gout->W; a.out_H = gout->H;
    return IDQN_OK;
}

long long* conv_prof(idqn_handle_s* h, NetSet& s, int role, const FwdPlan* pl) {
    return (h->cprof && h->cprof_role == role && &s == &h->train && pl->n_items <= 4096) ? (long long*)h->cprof : nullptr;
}

int planes_conv(idqn_handle_s* h, NetSet& s, int role, int nb, hipStream_t q) {
    CFwdArgs a;
    RoleGeom g;
    FwdPlan* pl;
    int rc = conv_args(h, s, role, nb, cu_budget(), a, g, pl);
    if (rc) return rc;
    // several items per CU (many sample blocks or heads): the persistent form, where the first fill, the table arithmetic
    // and the epilogue of an item run beside another item's matrix loop (convp_pp.hip).  IDQN_CONV_PP=0: one item per workgroup.
    static const int pp_roles = getenv("IDQN_CONV_PP") ? atoi(getenv("IDQN_CONV_PP")) : 15;  // bit r: role r (role 4, the Conv_1 data gradient, measured flat: profiles/r6_pp_ab.txt)
    if (((pp_roles >> role) & 1) && pl->n_items > cu_budget()) {
        FwdPlan* pp;
        const bool fwd = role <= 2;
        if (plan_fwd_pp(h, role + (fwd && &s == &h->infer ? 8 : 0), fwd ? s.n_nets : h->cfg.n_heads, nb, g, &pp) == IDQN_OK) {
            a.items_per_slot = pp->items_per_slot;
            for (int v = 0; v < 4; ++v) { a.r_begin[v] = pp->r_begin[v]; a.r_cnt[v] = pp->r_cnt[v]; }
            a.row_parts = pp->row_parts;
            return convp_launch_fwd_pp(a, g.NPA, g.CT, g.NQ, pp->NT, pp->n_items, pp->n_wg, pp->stage, pp->ring, pp->lds, q, pp->items_dev,
                                       conv_prof(h, s, role, pl));
        }
    }
    return convp_launch_fwd(a, g.NPA, g.CT, g.NQ, pl->NT, pl->n_items, pl->stage, pl->ring, pl->lds, q, conv_prof(h, s, role, pl));
}

// launch arguments + plan of a weight-gradient layer cut into n_chunks position chunks (0: the layer's default)
int wgrad_args(idqn_handle_s* h, int layer, int nb, int n_chunks, CWgradArgs& a, WgradPlan*& pl) {
    NetSet& s = h->train;
    const ConvL& l = h->conv[layer];
    int rc = plan_wgrad(h, layer, nb, &pl, n_chunks);
    if (rc) return rc;
    const unsigned short* xs[3] = {s.x1, s.a1p, s.a2p};
    const unsigned short* dys[3] = {h->da1p, h->da2p, h->da3p};
    const ActGeom* gx[3] = {&h->gx, &h->ga1, &h->ga2};
    const ActGeom* gd[3] = {&h->gda1, &h->gda2, &h->gda3};
    memset(&a, 0, sizeof(a));
    a.x = xs[layer]; a.dy = dys[layer]; a.pb = h->pbuf[layer]; a.slab = h->slab + h->slab_off[layer];
    a.n_chunks = pl->n_chunks; a.chunk_major = pl->chunk_major; a.kh_per_item = layer == 0 ? 1 : l.K;
    const int npx = layer == 0 ? 1 : 3;
    a.x_slot = gx[layer]->block * 2 * npx; a.dy_slot = gd[layer]->block * 6; a.slab_stride = h->slab_stride[layer];
    a.x_shared = layer == 0 ? 1 : 0;
    a.K = h->cfg.n_heads; a.nb = nb; a.KH = l.K; a.KW = l.K; a.S = l.S; a.CI = l.CI; a.CO = l.CO; a.OH = l.OH; a.OW = l.OW;
    a.x_pix = npx * l.CI * 64; a.x_plane = l.CI * 64; a.x_row = gx[layer]->Wp * a.x_pix;
    a.dy_pix = 3 * l.CO * 64; a.dy_Wp = gd[layer]->Wp; a.dy_lo_h = gd[layer]->lo_h; a.dy_lo_w = gd[layer]->lo_w;
    a.PG = pl->PG; a.out_div = layer == 0 ? 255.0f : 1.0f;
    return IDQN_OK;
}

int planes_wgrad(idqn_handle_s* h, int layer, int nb, hipStream_t q) {
    CWgradArgs a;
    WgradPlan* pl;
    int rc = wgrad_args(h, layer, nb, 0, a, pl);
    if (rc) return rc;
    const int NPX = layer == 0 ? 1 : 3, CT = h->conv[layer].CO / 32;
    h->npc_used[layer] = pl->n_chunks;
    return convp_launch_wgrad(a, NPX, pl->MT, CT, pl->n_items, pl->lds, q);
}

// Data gradient of conv `layer` and weight gradient of the same layer in ONE launch (convp_pair.hip) when that pair of
// kernels is built for the plans; *done = false: nothing was launched, the caller runs them one after the other.
int planes_pair(idqn_handle_s* h, int layer, int nb, hipStream_t q, bool* done) {
    *done = false;
    const int cus = cu_budget();
    if (layer < 1 || layer > 2) return IDQN_OK;
    // experiment knobs: IDQN_PAIR_D<layer> = workgroups planned for the data gradient, IDQN_PAIR_C<layer> = position chunks
    // of the weight gradient (default: what the data gradient leaves of the 256 CUs)
    auto knob = [&](const char* stem, int dflt) {
        char nm[32];
        snprintf(nm, sizeof nm, "%s%d", stem, layer);
        const char* e = debug_env(nm);
        return e ? atoi(e) : dflt;
    };
    const int d_target = knob("IDQN_PAIR_D", cus / 2);
    NetSet& s = h->train;
    const int role = layer == 2 ? 3 : 4, K = h->cfg.n_heads;
    const ConvL& l = h->conv[layer];
    CFwdArgs f;
    RoleGeom g;
    FwdPlan* pf;
    int rc = conv_args(h, s, role, nb, d_target, f, g, pf);
    if (rc) return rc;
    const int n_chunks = std::min(knob("IDQN_PAIR_C", 1 << 20), (cus - pf->n_items) / (K * l.K));  // what is left of the chip, in whole position chunks
    if (n_chunks < 1) return IDQN_OK;
    CWgradArgs w;
    WgradPlan* pw;
    if ((rc = wgrad_args(h, layer, nb, n_chunks, w, pw))) return rc;
    const int WCT = l.CO / 32, ntw = (pw->MT * WCT + 3) / 4;
    if (pf->n_items + pw->n_items > cus || !convp_pair_built(g.NPA, g.CT, g.NQ, pf->NT, 3, WCT, ntw, pw->PG)) return IDQN_OK;
    h->npc_used[layer] = pw->n_chunks;
    *done = true;
    return convp_launch_pair(f, g.NPA, g.CT, g.NQ, pf->NT, pf->n_items, pf->stage, pf->ring, pf->lds, w, 3, pw->MT, WCT, pw->n_items,
                             pw->lds, q, conv_prof(h, s, role, pf));
}

// ---- forward of a net set: staging, 3 convs, Dense_0 partials -----------------------------------
int cnn_forward(idqn_handle_s* h, NetSet& s, const uint8_t* st, const uint8_t* st2, int B, hipStream_t q, bool with_dense0 = true) {
    const int nb = cdiv(B, 32);
    IDQN_REQUIRE(nb <= s.nb_cap, "batch %d exceeds the workspace (%d blocks of 32)", B, s.nb_cap);
    if (h->planes) {
        static const char* nm[3] = {"conv0 fwd", "conv1 fwd", "conv2 fwd"};
        int rc = planes_stage(h, s, st, st2, B, nb, q);
        tl_mark(h, q, "stage (pixels + kernel packing)");
        for (int i = 0; i < 3 && !rc; ++i) { rc = planes_conv(h, s, i, nb, q); tl_mark(h, q, nm[i]); }
        if (rc) return rc;
    } else {
        PrepArgs pa;
        pa.src[0] = st; pa.src[1] = st2 ? st2 : st;
        pa.x = s.x;
        pa.E = (long)h->gx.H * h->gx.W * h->gx.C; pa.B = B; pa.nb = nb; pa.n_sets = s.n_in_sets; pa.g = h->gx;
        hipLaunchKernelGGL(k_prep_u8, dim3(cdiv(pa.E, 64), nb, s.n_in_sets), dim3(256), 0, q, pa);
        const float* ins[3] = {s.x, s.a1, s.a2};
        float* outs[3] = {s.a1, s.a2, s.a3};
        const ActGeom* gin[3] = {&h->gx, &h->ga1, &h->ga2};
        const ActGeom* gout[3] = {&h->ga1, &h->ga2, &h->ga3};
        for (int i = 0; i < 3; ++i) {
            const ConvL& l = h->conv[i];
            ConvFwdArgs a;
            memset(&a, 0, sizeof(a));
            a.in = ins[i]; a.out = outs[i]; a.wbase = s.wbase; a.wt_base = nullptr; a.in_set = (i == 0) ? s.in_set : s.ident;
            a.b_off = l.b_off; a.in_block = gin[i]->block; a.out_block = gout[i]->block;
            a.n_nets = s.n_nets; a.nb = nb; a.n_var = 1; a.epilogue = 0;
            a.KH = l.K; a.KWCI = l.K * l.CI; a.S = l.S; a.CI = l.CI; a.CO = l.CO; a.IWp = gin[i]->Wp;
            a.out_Wp = gout[i]->Wp; a.out_lo_h = gout[i]->lo_h; a.out_lo_w = gout[i]->lo_w;
            IDQN_REQUIRE(a.KWCI % 32 == 0, "conv %d: a kernel row of %d (kw, ci) rows is not a multiple of the 32-row chunk", i, a.KWCI);
            const int npw = (l.CO == 32) ? 4 : 2;  // positions per workgroup
            ConvVariant& v = a.var[0];
            v.w_off = l.w_off; v.in_off_h = 0; v.in_off_w = 0; v.OH = l.OH; v.OW = l.OW;
            v.out_mul = 1; v.out_add_h = 0; v.out_add_w = 0; v.pg_begin = 0;
            a.npg = cdiv(l.OH * l.OW, npw);
            a.n_items = (long)s.n_nets * nb * a.npg;
            if (l.CO == 32)
                hipLaunchKernelGGL((k_conv_fwd<1, 1>), dim3((unsigned)a.n_items), dim3(256), 0, q, a);
            else
                hipLaunchKernelGGL((k_conv_fwd<2, 1>), dim3((unsigned)a.n_items), dim3(256), 0, q, a);
        }
    }
    if (!with_dense0) {  // (the i-IQN heads take the trunk features from here)
        IDQN_HIP_CHECK(hipGetLastError());
        return IDQN_OK;
    }
    if (&s == &h->train && nb > 1) s.NS = d0_splits(h, s.n_nets, nb);  // (nb == 1: the value cnn_setup chose)
    else if (&s == &h->train) s.NS = h->NS;
    DenseFwdArgs d;
    d.in = s.a3; d.part = s.part; d.wbase = s.wbase; d.w_off = h->off_w0;
    d.n_nets = s.n_nets; d.nb = nb; d.NS = s.NS; d.n_jt = h->J / 128; d.F = h->F; d.J = h->J;
    d.bb_inner = (h->planes && nb > 1) ? 1 : 0;
    d.n_items = (long)s.n_nets * nb * d.NS * d.n_jt;
    d.net_rot = s.n_in_sets > 1 ? s.n_nets / 2 : 0;
    // the online nets' Dense_0 kernels (re-read by the fused update of the same step) with default-policy loads: k_dense0_fwd3
    // (K <= 3: the frozen target nets' kernels fit beside the online ones -- 2 K F J 4 <= 100 MB -- and are found on chip step after step:
    // K = 2 -3 us, K = 3 -2 us, K = 4 the same)
    const bool keep_target = d0_keep_online(h) && 2L * h->cfg.n_heads * h->F * h->J * 4 <= 100L << 20;
    // (the online nets' loads are default-policy for ANY number of heads: where nothing can stay on chip the fused update still runs 3-6 % faster behind
    // them -- K = 6 / 7 / 16 / 32: step -3.5 / -5 / -5 / -6 us, K = 8 / 64 the same; what depends on the size is only the store policy of theta_new)
    d.nt_from = s.n_in_sets > 1 ? (keep_target ? s.n_nets : s.n_nets / 2) : 0;
    // >= 8 sample blocks per net (B = 256): the tiled bf16x3 GEMM of the i-IQN heads (iqn_gemm.h: 256 x 256 tiles, operands split in
    // registers and parked in LDS as MFMA fragments) -- the same interleaved split-K and product order, so the same partials
    // layout for k_hidden; splits chosen to fill the chip (debug build, IDQN_D0_FWD_GEMM=0: the block-inner streaming kernel k_dense0_fwd3b).
    static const bool fwd_gemm = debug_int("IDQN_D0_FWD_GEMM", 1) != 0;
    if (fwd_gemm && h->planes && &s == &h->train && nb % 8 == 0 && h->J % 256 == 0 && h->F % 16 == 0) {
        const int per_split = s.n_nets * (nb / 8) * (h->J / 256);
        const int nsg = std::max(1, std::min(std::min(256 / std::max(1, per_split), 64), h->F / 16));
        if ((long)nb * nsg <= s.part_slabs) {
            s.NS = nsg;
            IqnD0FwdArgs g;
            g.x = s.a3; g.wbase = s.wbase; g.part = s.part; g.w_off = h->off_w0;
            g.V = s.n_nets; g.nb = nb; g.NS = nsg; g.F = h->F; g.J = h->J; g.clk = nullptr;
            const size_t lds = 2 * (size_t)IG_STAGE;
            static LdsAttrMark attr;
            if (attr.needs(lds)) IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_fwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_iqn_d0_fwd<2>, dim3((unsigned)(per_split * nsg)), dim3(512), lds, q, g);
            tl_mark(h, q, "dense0 fwd (tiled GEMM)");
            IDQN_HIP_CHECK(hipGetLastError());
            return IDQN_OK;
        }
    }
    if (h->planes) {
        if (d.bb_inner) hipLaunchKernelGGL(k_dense0_fwd3b, dim3(cdiv(d.n_items, 4)), dim3(256), 0, q, d);
        else {
            // fewer than 256 four-wave workgroups (plain DQN: 2 nets x 64 splits x 4 column tiles): two waves each, so that every
            // CU streams (a CU is the unit of streaming bandwidth)
            const int wpw = d.n_items <= 512 ? 2 : 4;
            hipLaunchKernelGGL(k_dense0_fwd3, dim3(cdiv(d.n_items, wpw)), dim3(64 * wpw), 0, q, d);
        }
    } else hipLaunchKernelGGL(k_dense0_fwd, dim3(cdiv(d.n_items, 4)), dim3(256), 0, q, d);
    tl_mark(h, q, "dense0 fwd");
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

void fill_adam_args(idqn_handle_s* h, long begin, long end, long skip_b, long skip_e, bool from_slabs, bool epilogue, AdamArgs& a);
int launch_adam(idqn_handle_s* h, long begin, long end, long skip_b, long skip_e, bool from_slabs, hipStream_t q, bool epilogue = false) {
    AdamArgs a;
    IDQN_REQUIRE(skip_b == skip_e || (skip_b >= begin && skip_e <= end && skip_b % 4 == 0 && skip_e % 4 == 0),
                 "launch_adam: bad skip range");
    fill_adam_args(h, begin, end, skip_b, skip_e, from_slabs, epilogue, a);
    hipLaunchKernelGGL(k_adam, dim3(cdiv(end - begin - (skip_e - skip_b), 256), h->cfg.n_heads), dim3(256), 0, q, a);  // 4 lanes per float4
    tl_mark(h, q, "adam (small leaves + slab sums)");
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

// arguments of the weight-gradient launch of conv layer i (shared by the plain and the mixed launch)
ConvWgradArgs make_wgrad_args(idqn_handle_s* h, int i, int nb) {
    NetSet& s = h->train;
    const ConvL& l = h->conv[i];
    const float* acts_in[3] = {s.x, s.a1, s.a2};
    const float* douts[3] = {h->da1, h->da2, h->da3};
    const ActGeom* gact[3] = {&h->gx, &h->ga1, &h->ga2};
    const ActGeom* gdo[3] = {&h->gda1, &h->gda2, &h->gda3};
    ConvWgradArgs a;
    a.in = acts_in[i]; a.dout = douts[i]; a.slab = h->slab + h->slab_off[i];
    a.in_net_stride = (i == 0) ? 0 : (long)nb * gact[i]->block;
    a.slab_stride = h->slab_stride[i];
    a.K = h->cfg.n_heads; a.nb = nb; a.npc = h->npc[i]; a.KH = l.K; a.S = l.S; a.CO = l.CO;
    a.OH = l.OH; a.OW = l.OW; a.pos_per_chunk = h->pos_per_chunk[i];
    a.gin = *gact[i]; a.gd = *gdo[i]; a.in_C = l.CI;
    if (i == 0) { a.KWe = 1; a.CIe = l.K * l.CI; } else { a.KWe = l.K; a.CIe = l.CI; }
    a.n_items = (long)a.K * a.KH * a.KWe * a.npc;  // workgroups
    return a;
}

// arguments of the data-gradient launch of conv layer i (1 or 2): a stride-1 forward convolution over dout with the
// transformed kernels of k_wt_build, one variant per output parity
int make_dgrad_args(idqn_handle_s* h, int i, int nb, ConvFwdArgs& a) {
    NetSet& s = h->train;
    const int K = h->cfg.n_heads;
    const ConvL& l = h->conv[i];
    const float* douts[3] = {h->da1, h->da2, h->da3};
    const ActGeom* gdo[3] = {&h->gda1, &h->gda2, &h->gda3};
    const float* acts_in[3] = {s.x, s.a1, s.a2};
    const ActGeom* gact[3] = {&h->gx, &h->ga1, &h->ga2};
    float* dins[3] = {nullptr, h->da1, h->da2};
    const ActGeom* gdi[3] = {nullptr, &h->gda1, &h->gda2};
    const int KHs = l.K / l.S, nvar = l.S * l.S;
    memset(&a, 0, sizeof(a));
    a.in = douts[i]; a.out = dins[i]; a.wbase = s.wbase; a.wt_base = h->wt[i]; a.wt_stride = h->wt_stride[i];
    a.in_set = s.ident; a.mask = acts_in[i];
    a.in_block = gdo[i]->block; a.out_block = gdi[i]->block; a.mask_block = gact[i]->block;
    a.n_nets = K; a.nb = nb; a.n_var = nvar; a.epilogue = 1;
    a.KH = KHs; a.KWCI = KHs * l.CO; a.S = 1; a.CI = l.CO; a.CO = l.CI; a.IWp = gdo[i]->Wp;
    a.out_Wp = gdi[i]->Wp; a.out_lo_h = gdi[i]->lo_h; a.out_lo_w = gdi[i]->lo_w;
    a.mask_Wp = gact[i]->Wp; a.mask_lo_h = gact[i]->lo_h; a.mask_lo_w = gact[i]->lo_w;
    IDQN_REQUIRE(a.KWCI % 32 == 0, "conv %d dgrad: %d rows per kernel row is not a multiple of 32", i, a.KWCI);
    const int npw = (l.CI == 32) ? 4 : 2;
    int pg = 0;
    for (int vi = 0; vi < nvar; ++vi) {
        const int rh = vi / l.S, rw = vi % l.S;
        const int ph = (rh + l.PLh) % l.S, pw = (rw + l.PLw) % l.S;
        ConvVariant& v = a.var[vi];
        v.w_off = (long)vi * KHs * KHs * l.CO * l.CI;
        v.in_off_h = (rh + l.PLh - ph) / l.S + gdo[i]->lo_h - KHs + 1;
        v.in_off_w = (rw + l.PLw - pw) / l.S + gdo[i]->lo_w - KHs + 1;
        v.OH = (l.IH - rh + l.S - 1) / l.S; v.OW = (l.IW - rw + l.S - 1) / l.S;
        v.out_mul = l.S; v.out_add_h = rh; v.out_add_w = rw; v.pg_begin = pg;
        IDQN_REQUIRE(v.in_off_h >= 0 && v.in_off_w >= 0 && v.OH > 0 && v.OW > 0, "conv %d dgrad: bad variant geometry", i);
        pg += cdiv(v.OH * v.OW, npw);
    }
    a.npg = pg;
    a.n_items = (long)K * nb * a.npg;
    return IDQN_OK;
}

int cnn_backward_rest(idqn_handle_s* h, int B, bool fuse_adam, hipStream_t q);

// Dense_0 weight gradient (+ fused Adam) over nb_total sample blocks addressed through (outer, head, inner) strides
int launch_dense0_wgrad(idqn_handle_s* h, const float* a3, const float* dh, int nb_total, int nb_inner, long a3_outer,
                        long a3_head, long a3_inner, long dh_outer, long dh_head, long dh_inner, bool fuse_adam,
                        bool profile, hipStream_t q, bool fuse_dg = false) {
    const int K = h->cfg.n_heads;
    DenseWgradArgs dw;
    dw.dpart = h->dpart;
    dw.a3p = dw.dhp = nullptr;
    // The fused update over a GLOBAL batch (factored data-parallel step, >= 2 sample blocks per head): the factors are
    // split into bf16 planes once and the contraction runs at the bf16 MFMA rate.
    const bool bf3 = h->planes && fuse_adam && !fuse_dg && nb_total >= 2 && h->J % 256 == 0;
    if (bf3) {
        if (h->fact_planes_cap < nb_total) {  // (first step of a job, or a larger world: outside any timed region)
            if (h->fact_planes) { IDQN_HIP_CHECK(hipStreamSynchronize(q)); IDQN_HIP_CHECK(hipFree(h->fact_planes)); }
            IDQN_HIP_CHECK(hipMalloc((void**)&h->fact_planes, (size_t)3 * nb_total * K * (h->F + h->J) * 32 * 2));
            h->fact_planes_cap = nb_total;
        }
        SplitFactorsArgs sa;
        sa.a3 = a3; sa.dh = dh; sa.a3_outer = a3_outer; sa.a3_head = a3_head; sa.a3_inner = a3_inner;
        sa.dh_outer = dh_outer; sa.dh_head = dh_head; sa.dh_inner = dh_inner;
        sa.K = K; sa.nb = nb_total; sa.nb_inner = nb_inner; sa.F = h->F; sa.J = h->J;
        sa.a3p = h->fact_planes; sa.dhp = h->fact_planes + (long)3 * nb_total * K * h->F * 32;
        // The default update kernel (k_dense0_wgrad_alds<1>) splits its a3 tiles itself, straight from the gathered f32 factors:
        // only dL/dh (6 % of the factor bytes) goes through the plane copy.
        sa.a3p = nullptr;
        hipLaunchKernelGGL(k_split_factors, dim3((unsigned)cdiv((long)h->J * 4, 256), (unsigned)(nb_total * K)), dim3(256), 0, q, sa);
        tl_mark(h, q, "factor planes");
        dw.a3p = sa.a3p; dw.dhp = sa.dhp;
    }
    dw.a3 = a3; dw.dh = dh; dw.grad = h->grad; dw.theta = h->online; dw.mu = h->mu; dw.nu = h->nu;
    dw.bcinv = h->bcinv; dw.ad = h->ad; dw.g_w0_base = h->g_w0_base; dw.g_w0_stride = h->g_w0_end - h->g_w0_begin;
    dw.w_off = h->off_w0; dw.P = h->L.head_stride;
    dw.a3_outer = a3_outer; dw.a3_head = a3_head; dw.a3_inner = a3_inner;
    dw.dh_outer = dh_outer; dw.dh_head = dh_head; dw.dh_inner = dh_inner;
    h->d0_rows = false;
    const int nq = (h->J % 256 == 0) ? 2 : 1;  // 256- or 128-wide column tiles
    dw.K = K; dw.nb = nb_total; dw.nb_inner = nb_inner; dw.n_ft = h->F / 32; dw.n_jt = h->J / (128 * nq);
    dw.F = h->F; dw.J = h->J; dw.item0 = 0; dw.keep_heads = d0_keep_heads(h);
    dw.da3p = nullptr; dw.da3f = nullptr; dw.pb = nullptr; dw.C = 0; memset(&dw.g, 0, sizeof(dw.g));

    dw.n_items = (long)K * dw.n_ft * dw.n_jt;  // workgroups
    const dim3 wgrid((unsigned)dw.n_items);
    // profiling: the start / stop events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL), so the
    // elapsed time is the kernel's, without the gaps that separate marker packets from their neighbours
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profile && h->ev_used + 2 <= (int)h->ev.size()) {
        e0 = h->ev[h->ev_used]; e1 = h->ev[h->ev_used + 1];
        h->ev_used += 2;
    }
    // (the extended launch only when there is something to time: it is not a capturable node of a step graph)
#define D0W_LAUNCH(...)                                                                                   \
    do {                                                                                                  \
        if (e0) hipExtLaunchKernelGGL((k_dense0_wgrad<__VA_ARGS__>), wgrid, dim3(256), 0, q, e0, e1, 0, dw); \
        else hipLaunchKernelGGL((k_dense0_wgrad<__VA_ARGS__>), wgrid, dim3(256), 0, q, dw);               \
    } while (0)
    // Default where it applies (one sample block, J = 512, nothing deferred): one workgroup per PAIR of column tiles
    // (dense0_pair_body) -- the second tile's requests go out under the first tile's last phase, and the workgroup finishes
    // dL/da3 itself (it holds both partials): no partials in HBM, no k_da3_finalize launch, no hand-off.  Bit-identical.
    // Measured on three boxes against two workgroups + finalize (profiles/r4_d0_pair_ab.txt): step -0.7 / -3 / -5 us; the kernel
    // itself takes 2 - 5 us more than the tile kernel (a row's two 1 KB halves are streamed 12 us apart instead of side by side by
    // sibling workgroups), the finalize launch it replaces took 5.3 us + a boundary.
    const bool pair = fuse_adam && nq == 2 && fuse_dg && nb_total == 1 && h->J == 512;
    if (pair) {
        // one workgroup per PAIR of column tiles; the launch finishes dL/da3 itself
        dw.da3p = h->da3p; dw.da3f = h->da3; dw.pb = h->pbuf[2]; dw.g = h->gda3; dw.C = h->conv[2].CO;
        h->d0_rows = true;
        const dim3 pgrid((unsigned)(K * dw.n_ft));
        // one or two heads at the Nature width: theta, m, v and the target nets (4 x 16 MB per head) fit the memory-side cache beside the step's
        // other traffic -- every stream of the update default-policy (K = 1 -1.9 us, K = 2 -4 us; K = 3 +5, K = 5 +16: profiles/r5_d0_keep_online_ab.txt)
        const bool keep_all = d0_keep_online(h) && 4L * K * h->F * h->J * 4 <= 128L << 20;
        // heads [0, keep) keep theta_new on chip, the others store it non-temporally: one launch per policy (a kernel with both bodies spills)
        const int keep = keep_all ? K : std::min(dw.keep_heads, K);
        for (int part = 0; part < 2; ++part) {
            const int k0 = part == 0 ? 0 : keep, k1 = part == 0 ? keep : K;
            if (k1 <= k0) continue;
            DenseWgradArgs dp = dw;
            dp.item0 = k0 * dw.n_ft;
            const dim3 grid((unsigned)((k1 - k0) * dw.n_ft));
            hipEvent_t s0 = (e0 && k0 == 0) ? e0 : nullptr, s1 = (e0 && k1 == K) ? e1 : nullptr;
            const void* fn = keep_all ? (const void*)k_dense0_wgrad_pair<false, false, true>
                             : part == 0 ? (const void*)k_dense0_wgrad_pair<false, false, false> : (const void*)k_dense0_wgrad_pair<false, true, false>;
            void* args[] = {&dp};
            if (s0 || s1) IDQN_HIP_CHECK(hipExtLaunchKernel(fn, grid, dim3(256), args, 0, q, s0, s1, 0));
            else IDQN_HIP_CHECK(hipLaunchKernel(fn, grid, dim3(256), args, 0, q));
        }
    } else if (fuse_adam && nq == 2 && fuse_dg) D0W_LAUNCH(true, 2, true);
    else if (bf3) {  // the factored data-parallel update over >= 2 sample blocks: a3 fragments through LDS
        const int keep = std::min(dw.keep_heads, K);
        for (int part = 0; part < 2; ++part) {  // (one launch per cache policy of theta_new, as for the pair kernel)
            const int k0 = part == 0 ? 0 : keep, k1 = part == 0 ? keep : K;
            if (k1 <= k0) continue;
            DenseWgradArgs dp = dw;
            dp.item0 = k0 * dw.n_ft * dw.n_jt;
            const dim3 grid((unsigned)((k1 - k0) * dw.n_ft * dw.n_jt));
            hipEvent_t s0 = (e0 && k0 == 0) ? e0 : nullptr, s1 = (e0 && k1 == K) ? e1 : nullptr;
            const void* fn = part == 0 ? (const void*)k_dense0_wgrad_alds<1, false> : (const void*)k_dense0_wgrad_alds<1, true>;
            void* args[] = {&dp};
            if (s0 || s1) IDQN_HIP_CHECK(hipExtLaunchKernel(fn, grid, dim3(256), args, 0, q, s0, s1, 0));
            else IDQN_HIP_CHECK(hipLaunchKernel(fn, grid, dim3(256), args, 0, q));
        }
    }
    else if (fuse_adam && nq == 2) D0W_LAUNCH(true, 2);
    else if (fuse_adam) D0W_LAUNCH(true, 1);
    else if (nq == 2) D0W_LAUNCH(false, 2);
    else D0W_LAUNCH(false, 1);
#undef D0W_LAUNCH
    tl_mark(h, q, fuse_dg ? "dense0 wgrad + dgrad + adam" : fuse_adam ? "dense0 wgrad + adam" : "dense0 wgrad");
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

int cnn_backward(idqn_handle_s* h, const int32_t* action, const float* reward, const uint8_t* terminal, int B,
                 int Bdiv, bool fuse_adam, bool profile, bool stop_after_dense0, bool stop_before_dense0_wgrad,
                 hipStream_t q) {
    const int K = h->cfg.n_heads, nb = cdiv(B, 32);
    NetSet& s = h->train;
    const ConvL *c0 = &h->conv[0], *c1 = &h->conv[1], *c2 = &h->conv[2];
    // head: h + Dense_1 partials for all 2K nets, then TD / loss / dL/dq / dL/dh / Dense_1 + Dense_0-bias gradients
    HiddenArgs hi;
    hi.part = s.part; hi.wbase = s.wbase; hi.b0_off = h->off_b0; hi.w1_off = h->off_w1; hi.nb = nb; hi.NS = s.NS;
    hi.J = h->J; hi.A = h->cfg.n_actions; hi.hbuf = h->hbuf; hi.qpart = h->qpart;
    hipLaunchKernelGGL(k_hidden, dim3(h->J / 32, 2 * K * nb), dim3(256), 0, q, hi);
    tl_mark(h, q, "hidden");
    TdArgs ta;
    ta.hbuf = h->hbuf; ta.qpart = h->qpart; ta.wbase = s.wbase; ta.b0_off = h->off_b0; ta.w1_off = h->off_w1;
    ta.b1_off = h->off_b1; ta.P = h->L.head_stride; ta.K = K;
    {
        const long w0n = h->g_w0_end - h->g_w0_begin;
        ta.gP = h->gP; ta.g_b0_off = h->off_b0 - w0n; ta.g_w1_off = h->off_w1 - w0n; ta.g_b1_off = h->off_b1 - w0n;
    } ta.nb = nb; ta.J = h->J; ta.A = h->cfg.n_actions;
    ta.B = B; ta.Bdiv = Bdiv; ta.action = action; ta.reward = reward; ta.terminal = terminal; ta.gamma_n = h->gamma_n;
    ta.prof = (h->cprof && h->cprof_role == 9) ? (long long*)h->cprof : nullptr;
    ta.dh = dh_of(h, nb); ta.q_dbg = h->qdbg; ta.grad = h->grad; ta.losses = h->losses;
    ta.count = h->count; ta.bcinv = h->bcinv; ta.b1 = h->ad.b1; ta.b2 = h->ad.b2;
    ta.cum = h->cum; ta.finish_step = fuse_adam ? 1 : 0;
    ta.is_weight = h->is_weight; ta.td_abs = h->td_abs; ta.bpart = nullptr; ta.bctr = nullptr;
    ta.bcinv_done = h->planes ? 1 : 0;  // (the plane path's staging launch writes the bias corrections)
    h->wt_ready = false;
    if (!h->planes) {  // (the plane path packs the data-gradient kernels in its staging launch)
        WtBuildArgs wb;
        const int nx = wt_build_args(h, wb);
        hipLaunchKernelGGL(k_td_dh_wt, dim3((h->J / 32) * K + nx * K * 2), dim3(256), 0, q, ta, wb, nx);
        h->wt_ready = true;
    } else {
        // several sample blocks: one workgroup per block, the last to arrive adds the blocks' gradient partials in block order
        const bool per_block = nb > 1 && h->td_bpart && debug_int("IDQN_TD_PER_BLOCK", 1);
        if (per_block) { ta.bpart = h->td_bpart; ta.bctr = h->td_bctr; }
        hipLaunchKernelGGL(k_td_dh, dim3(h->J / 32, K, per_block ? nb : 1), dim3(256), 0, q, ta);
    }
    tl_mark(h, q, "td + loss + dh");
    // Dense_0 data gradient -> da3 (zero-bordered for the Conv_2 data gradient).  On the fused single-device path it is
    // computed INSIDE the weight-gradient + Adam kernel (theta streams once) and finished by k_da3_finalize; the two-call
    // paths of the data-parallel step need it before the weight gradient and keep the separate kernel.
    // Several sample blocks on ONE device (B > 32): the schedule of the factored data-parallel step without its collectives -- the
    // data gradient as its own launch, the factors split once into bf16 planes, the update contracting them at the bf16 MFMA rate
    // (k_dense0_wgrad_alds) -- instead of the fused kernel's f32 MFMAs over every block (B = 256: profiles/r5_b256_ab.txt)
    const bool many = nb >= 3 && h->planes && h->J % 256 == 0;  // (two blocks: the same either way, 0.5065 against 0.5085 ms)
    const bool fuse_dg = fuse_adam && !stop_after_dense0 && !stop_before_dense0_wgrad && h->dpart && !many;
    // eight (or a multiple of eight) local sample blocks: the data gradient as the tiled bf16x3 GEMM the i-IQN heads use for their
    // fraction blocks (csrc/iqn_gemm.h: W read once per group of 8 blocks, products at the bf16 rate) + the finalize launch for the
    // ReLU mask / planes / per-position sums, instead of one f32-MFMA pass over W per block
    if (!fuse_dg && many && nb % 8 == 0 && h->dpart && h->J % 16 == 0) {
        IqnD0DgradArgs g;
        g.dh = dh_of(h, nb); g.wbase = s.wbase; g.dx = h->dpart; g.w_off = h->off_w0; g.K = K; g.nb = nb; g.F = h->F; g.J = h->J;
        const size_t lds = 2 * (size_t)IG_STAGE;
        // rows of W per workgroup (64 x row tiles per wave): what fills the chip -- rounds of items x length of an item (K = 5, 8 blocks:
        // 205 items of 192 rows in one round instead of 155 of 256)
        int rt = 4;
        {
            long best = 0;
            for (int c = 4; c >= 2; --c) {
                const long cost = (long)cdiv((long)K * (nb / 8) * cdiv(h->F, 64 * c), cu_budget()) * c;
                if (c == 4 || cost < best) { best = cost; rt = c; }
            }
            if (debug_int("IDQN_DGRAD_RT", 0) >= 2 && debug_int("IDQN_DGRAD_RT", 0) <= 4) rt = debug_int("IDQN_DGRAD_RT", 0);
        }
        const unsigned grid = (unsigned)(K * (nb / 8) * cdiv(h->F, 64 * rt));
        // (mask, planes and per-position sums in the GEMM's epilogue instead of the finalize launch: bit-identical, 110.3 us against
        // 84.3 + 25.6 -- the epilogue's plane stores are 8-byte pieces a row apart)
        {
            static LdsAttrMark attr;
            if (attr.needs(lds)) {
                IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_dgrad<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_dgrad<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_dgrad<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (rt == 4) hipLaunchKernelGGL((k_iqn_d0_dgrad<2, 4>), dim3(grid), dim3(512), lds, q, g);
            else if (rt == 3) hipLaunchKernelGGL((k_iqn_d0_dgrad<2, 3>), dim3(grid), dim3(512), lds, q, g);
            else hipLaunchKernelGGL((k_iqn_d0_dgrad<2, 2>), dim3(grid), dim3(512), lds, q, g);
            tl_mark(h, q, "dense0 dgrad (tiled GEMM)");
            Da3FinalizeArgs fa;
            fa.dpart = h->dpart; fa.a3 = s.a3; fa.da3 = h->da3; fa.da3p = h->da3p; fa.pb = h->pbuf[2];
            fa.n_rows = (long)K * nb * h->F; fa.n_jt = 1; fa.F = h->F; fa.C = c2->CO; fa.K = K; fa.nb = nb; fa.g = h->gda3;
            hipLaunchKernelGGL(k_da3_finalize, dim3(cdiv(fa.n_rows * 8, 256)), dim3(256), 0, q, fa);
            tl_mark(h, q, "da3 finalize (sum, mask, planes)");
        }
    } else
    if (!fuse_dg) {
        DenseDgradArgs dd;
        dd.dh = dh_of(h, nb); dd.a3 = s.a3; dd.da3 = h->da3; dd.da3p = h->da3p; dd.pb = h->pbuf[2]; dd.wbase = s.wbase; dd.w_off = h->off_w0;
        dd.K = K; dd.nb = nb; dd.n_ft = h->F / 32; dd.F = h->F; dd.J = h->J; dd.C = c2->CO; dd.g = h->gda3; dd.raw = nullptr;
        // 4 or 3 f tiles per workgroup, whichever leaves the busiest CU fewer tiles (two workgroups fit a CU's LDS)
        const long wg4 = (long)K * nb * cdiv(dd.n_ft, 4), wg3 = (long)K * nb * cdiv(dd.n_ft, 3);
        const long busy4 = cdiv(wg4, 256) * 4, busy3 = wg3 <= 512 ? cdiv(wg3, 256) * 3 : 1 << 30;
        if (busy3 < busy4) {
            dd.n_items = wg3;
            hipLaunchKernelGGL((k_dense0_dgrad<3>), dim3((unsigned)dd.n_items), dim3(192), h->J * 32 * 4, q, dd);
        } else {
            dd.n_items = wg4;
            hipLaunchKernelGGL((k_dense0_dgrad<4>), dim3((unsigned)dd.n_items), dim3(256), h->J * 32 * 4, q, dd);
        }
        tl_mark(h, q, "dense0 dgrad");
    }
    if (stop_before_dense0_wgrad) {
        h->pend_B = B; h->pend_stage = 1; h->pend_profile = profile;
        IDQN_HIP_CHECK(hipGetLastError());
        return IDQN_OK;
    }
    // Dense_0 weight gradient (+ Adam): the dominant, HBM-bound kernel
    int rcw = launch_dense0_wgrad(h, s.a3, dh_of(h, nb), nb, nb, 0, (long)nb * h->F * 32, (long)h->F * 32, 0,
                                  (long)nb * h->J * 32, (long)h->J * 32, fuse_adam, profile, q, fuse_dg);
    if (rcw) return rcw;
    if (fuse_dg && !h->d0_rows) {
        Da3FinalizeArgs fa;
        fa.dpart = h->dpart; fa.a3 = s.a3; fa.da3 = h->da3; fa.da3p = h->da3p; fa.pb = h->pbuf[2];
        fa.n_rows = (long)K * nb * h->F; fa.n_jt = h->J / 256; fa.F = h->F; fa.C = c2->CO; fa.K = K; fa.nb = nb; fa.g = h->gda3;
        hipLaunchKernelGGL(k_da3_finalize, dim3(cdiv(fa.n_rows * 8, 256)), dim3(256), 0, q, fa);
        tl_mark(h, q, "da3 finalize (sum, mask, planes)");
    }
    if (stop_after_dense0) {
        h->pend_B = B; h->pend_stage = 2;
        IDQN_HIP_CHECK(hipGetLastError());
        return IDQN_OK;
    }
    return cnn_backward_rest(h, B, fuse_adam, q);
}

// the slab descriptor of conv layer i (position chunks = what its weight-gradient launch of THIS step wrote)
void fill_seg(idqn_handle_s* h, int i, SlabSeg& g, long first_block) {
    const ConvL& l = h->conv[i];
    g.slab = h->slab + h->slab_off[i]; g.slab_stride = h->slab_stride[i]; g.w_off = l.w_off; g.b_off = l.b_off;
    g.wsize = (long)l.K * l.K * l.CI * l.CO; g.npc = h->planes ? h->npc_used[i] : h->npc[i]; g.bsize = l.CO;
    g.first_block = first_block;
}

void fill_adam_args(idqn_handle_s* h, long begin, long end, long skip_b, long skip_e, bool from_slabs, bool epilogue, AdamArgs& a) {
    a.ep_count = epilogue ? h->count : nullptr; a.ep_losses = h->losses; a.ep_cum = h->cum;
    a.theta = h->online; a.mu = h->mu; a.nu = h->nu; a.grad = h->grad; a.bcinv = h->bcinv; a.ad = h->ad;
    a.P = h->L.head_stride; a.begin = begin; a.end = end; a.skip_begin = skip_b; a.skip_end = skip_e;
    a.K = h->cfg.n_heads; a.n_seg = from_slabs ? 3 : 0;
    a.gP = h->gP; a.w0_begin = h->g_w0_begin; a.w0_end = h->g_w0_end; a.g_w0_base = h->g_w0_base;
    for (int i = 0; i < 3; ++i) a.seg[i] = h->segs[i];
    if (skip_b == skip_e) a.skip_begin = a.skip_end = end;  // nothing skipped
}

int cnn_backward_rest(idqn_handle_s* h, int B, bool fuse_adam, hipStream_t q) {
    const int K = h->cfg.n_heads, nb = cdiv(B, 32);
    NetSet& s = h->train;
    const ConvL* cl[3] = {&h->conv[0], &h->conv[1], &h->conv[2]};
    SlabReduceArgs r;
    r.grad = h->grad; r.gP = h->gP; r.K = K; r.n_seg = 3;
    long nblk = 0;
    if (h->planes) {
        // data gradients (stride-1 convolutions over the zero-bordered dout planes), then the weight gradients
        // per layer, top down: its data gradient and its weight gradient read the same dy and are independent of each other --
        // one launch for both where that pair of kernels is built (planes_pair), else one after the other
        static const char* nd[3] = {"", "conv1 dgrad", "conv2 dgrad"};
        static const char* nw[3] = {"conv0 wgrad", "conv1 wgrad", "conv2 wgrad"};
        static const char* np[3] = {"", "conv1 dgrad + wgrad", "conv2 dgrad + wgrad"};
        int rc = IDQN_OK;
        for (int i = 2; i >= 0 && !rc; --i) {
            bool paired = false;
            if (i >= 1) rc = planes_pair(h, i, nb, q, &paired);
            if (paired) { tl_mark(h, q, np[i]); continue; }
            if (i >= 1 && !rc) { rc = planes_conv(h, s, i == 2 ? 3 : 4, nb, q); tl_mark(h, q, nd[i]); }
            if (!rc) {
                rc = planes_wgrad(h, i, nb, q);
                tl_mark(h, q, nw[i]);
            }
        }
        if (rc) return rc;
    } else {
        // data gradients as forward convolutions over the zero-bordered dout buffers with transformed weights
        int rcb = build_dgrad_weights(h, q);
        if (rcb) return rcb;
        for (int i = 2; i >= 1; --i) {
            const ConvL& l = *cl[i];
            ConvFwdArgs a;
            int rcd = make_dgrad_args(h, i, nb, a);
            if (rcd) return rcd;
            if (l.CI == 32)
                hipLaunchKernelGGL((k_conv_fwd<1, 1>), dim3((unsigned)a.n_items), dim3(256), 0, q, a);
            else
                hipLaunchKernelGGL((k_conv_fwd<2, 1>), dim3((unsigned)a.n_items), dim3(256), 0, q, a);
        }
    }
    // conv weight gradients: slabs (one region per layer), then ONE reduce launch into the gradient arena
    for (int i = 2; i >= 0; --i) {
        const ConvL& l = *cl[i];
        if (!h->planes) {
            ConvWgradArgs a = make_wgrad_args(h, i, nb);
            const int nit = a.CIe / 32, not_ = a.CO / 32;
            dim3 grid((unsigned)a.n_items);
            if (nit == 1 && not_ == 1) hipLaunchKernelGGL((k_conv_wgrad<1, 1>), grid, dim3(256), 0, q, a);
            else if (nit == 1 && not_ == 2) hipLaunchKernelGGL((k_conv_wgrad<1, 2>), grid, dim3(256), 0, q, a);
            else if (nit == 2 && not_ == 1) hipLaunchKernelGGL((k_conv_wgrad<2, 1>), grid, dim3(256), 0, q, a);
            else hipLaunchKernelGGL((k_conv_wgrad<2, 2>), grid, dim3(256), 0, q, a);
        }
        SlabSeg& g = r.seg[2 - i];
        fill_seg(h, i, g, nblk);
        (void)l;
        nblk += cdiv(g.wsize + g.bsize, 256);
    }
    for (int i = 0; i < 3; ++i) h->segs[i] = r.seg[i];
    if (!fuse_adam)  // two-phase step: the gradient arena must be complete (it gets all-reduced); the fused
        hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)nblk, K), dim3(256), 0, q, r);  // path sums in k_adam
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

int step_epilogue(idqn_handle_s* h, bool bump, hipStream_t q) {
    hipLaunchKernelGGL(k_step_epilogue, dim3(cdiv(h->cfg.n_heads, 64)), dim3(64), 0, q, h->count, h->losses, h->cum,
                       h->cfg.n_heads, bump ? 1 : 0);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int idqn_layout(const idqn_config_t* cfg, int32_t* n_leaves, idqn_leaf_t* leaves, int64_t* head_stride) {
    IDQN_REQUIRE(cfg && n_leaves && leaves && head_stride, "idqn_layout: null pointer");
    Layout L;
    int rc = build_layout(*cfg, L);
    if (rc) return rc;
    *n_leaves = L.n_leaves;
    memcpy(leaves, L.leaves, sizeof(idqn_leaf_t) * L.n_leaves);
    *head_stride = L.head_stride;
    return IDQN_OK;
}

extern "C" int idqn_create(const idqn_config_t* cfg, float* online_dev, float* target_dev, float* mu_dev, float* nu_dev,
                           float* grad_dev, int32_t* count_dev, float* losses_dev, double* cum_losses_dev,
                           idqn_handle_t* out) {
    IDQN_REQUIRE(cfg && out, "idqn_create: null pointer");
    IDQN_REQUIRE(online_dev && target_dev && mu_dev && nu_dev && grad_dev && count_dev && losses_dev && cum_losses_dev,
                 "idqn_create: every arena pointer is required");
    IDQN_REQUIRE(online_dev != target_dev, "idqn_create: online and target must be distinct buffers (the reference's "
                                           "aliasing at idqn.py:56 is only safe for immutable arrays)");
    int ndev = 0;
    IDQN_HIP_CHECK(hipGetDeviceCount(&ndev));
    IDQN_REQUIRE(ndev > 0, "idqn_create: no HIP device");
    idqn_handle_s* h = new idqn_handle_s();
    h->cfg = *cfg;
    int rc = build_layout(*cfg, h->L);
    if (rc) { delete h; return rc; }
    h->online = online_dev; h->target = target_dev; h->mu = mu_dev; h->nu = nu_dev; h->grad = grad_dev;
    h->count = count_dev; h->losses = losses_dev; h->cum = cum_losses_dev;
    h->ad.lr_neg = (float)(-cfg->learning_rate);
    h->ad.b1 = (float)cfg->adam_b1; h->ad.b2 = (float)cfg->adam_b2;
    h->ad.omb1 = (float)(1.0 - cfg->adam_b1); h->ad.omb2 = (float)(1.0 - cfg->adam_b2);
    h->ad.eps = (float)cfg->adam_eps;
    h->gamma_n = (float)cfg->gamma_n;
    h->nb_max = cdiv(cfg->max_batch, 32);
    h->gP = h->L.head_stride;
    if (cfg->arch == IDQN_ARCH_CNN) {
        h->g_w0_begin = h->L.leaves[6].offset; h->g_w0_end = h->L.leaves[7].offset;  // Dense_0/kernel .. Dense_0/bias
        h->gP = h->L.head_stride - (h->g_w0_end - h->g_w0_begin);
    }
    h->g_w0_base = (long)cfg->n_heads * h->gP + 64;
    {  // conv arithmetic: f32-accurate products on the bf16 matrix cores (convp.h, the default), or the f32 MFMA kernels
        const char* mode = getenv("IDQN_CONV");
        h->planes = cfg->arch == IDQN_ARCH_CNN && !(mode && strcmp(mode, "f32") == 0);
        IDQN_REQUIRE(!mode || !strcmp(mode, "f32") || !strcmp(mode, "bf16x3"), "IDQN_CONV must be f32 or bf16x3, got '%s'", mode);
    }
    rc = alloc_zero(&h->bcinv, 2L * cfg->n_heads + 64, h, "bcinv");
    const bool general = cfg->arch == IDQN_ARCH_CNN && !cnn_fast_shape(*cfg);
    if (general) h->planes = false;
    if (!rc) rc = general ? gcnn_setup(h) : (cfg->arch == IDQN_ARCH_CNN ? cnn_setup(h) : fc_setup(h));
    if (rc) { idqn_destroy(h); return rc; }
    h->ev.resize(2 * 2048);
    for (auto& e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) { idqn_set_error("hipEventCreate failed"); idqn_destroy(h); return IDQN_E_HIP; }
    IDQN_HIP_CHECK(hipDeviceSynchronize());
    *out = h;
    return IDQN_OK;
}

extern "C" int idqn_destroy(idqn_handle_t h) {
    if (!h) return IDQN_OK;
    (void)hipDeviceSynchronize();
    for (void* p : h->owned) (void)hipFree(p);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : h->tl_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& g : h->act_graphs) (void)hipGraphExecDestroy(g.second);
    for (auto& g : h->step_graphs)
        if (g.second.second) (void)hipGraphExecDestroy(g.second.second);
    if (h->act_stream) (void)hipStreamDestroy(h->act_stream);
    if (h->act_mail) (void)hipHostFree(h->act_mail);
    if (h->fact_planes) (void)hipFree(h->fact_planes);
    delete h;
    return IDQN_OK;
}

extern "C" int idqn_learn_on_batch(idqn_handle_t h, const void* state_dev, const void* next_state_dev,
                                   const int32_t* action_dev, const float* reward_dev, const uint8_t* terminal_dev,
                                   int32_t batch, int32_t batch_mean_divisor, uint32_t flags, void* stream) {
    IDQN_REQUIRE(h && state_dev && next_state_dev && action_dev && reward_dev && terminal_dev, "idqn_learn_on_batch: null pointer");
    IDQN_REQUIRE(batch >= 1 && batch <= h->cfg.max_batch, "idqn_learn_on_batch: batch %d not in [1, %d]", batch, h->cfg.max_batch);
    IDQN_REQUIRE(batch_mean_divisor >= batch, "idqn_learn_on_batch: mean divisor %d < batch %d", batch_mean_divisor, batch);
    hipStream_t q = (hipStream_t)stream;
    const bool stop0 = flags & IDQN_F_STOP_AFTER_DENSE0, stopb = flags & IDQN_F_STOP_BEFORE_DENSE0_WGRAD;
    const bool grads_only = (flags & IDQN_F_GRADS_ONLY) || stop0 || stopb, profile = flags & IDQN_F_PROFILE;
    IDQN_REQUIRE(!(stop0 || stopb) || h->cfg.arch == IDQN_ARCH_CNN, "the IDQN_F_STOP_* flags belong to the cnn path");
    h->pend_B = 0; h->pend_stage = 0;
    h->tl_on = (flags & IDQN_F_PROFILE_ALL) != 0;
    if (h->tl_on && h->tl_ev.empty()) {
        h->tl_ev.resize(2048);
        h->tl_name.resize(2048);
        for (auto& e : h->tl_ev) IDQN_HIP_CHECK(hipEventCreate(&e));
    }
    tl_mark(h, q, nullptr);
    int rc;
    if (h->gc.on) {
        IDQN_REQUIRE(!(stop0 || stopb), "the IDQN_F_STOP_* flags belong to the MFMA cnn path");
        if ((rc = gcnn_learn(h, (const uint8_t*)state_dev, (const uint8_t*)next_state_dev, action_dev, reward_dev, terminal_dev, batch,
                             batch_mean_divisor, grads_only, profile, q)))
            return rc;
        if (!grads_only && (rc = launch_adam(h, 0, h->L.head_stride, 0, 0, false, q))) return rc;
        return IDQN_OK;
    }
    if (h->cfg.arch == IDQN_ARCH_CNN) {
        auto issue = [&](hipStream_t qs) -> int {
            int r;
            if ((r = cnn_forward(h, h->train, (const uint8_t*)state_dev, (const uint8_t*)next_state_dev, batch, qs))) return r;
            if ((r = cnn_backward(h, action_dev, reward_dev, terminal_dev, batch, batch_mean_divisor, !grads_only, profile, stop0, stopb, qs)))
                return r;
            if (!grads_only) {
                // every leaf except Dense_0/kernel (already updated by the fused weight-gradient kernel)
                const long w0_b = h->off_w0, w0_e = h->off_b0;
                if ((r = launch_adam(h, 0, h->L.head_stride, w0_b, w0_e, true, qs))) return r;
            }
            return IDQN_OK;
        };
        // A whole plain step is 13 launches whose arguments are pointers and sizes only: for a given set of batch buffers it
        // can be replayed as ONE hipGraph (opt-in).  The first call of a key runs eagerly (it builds the launch plans, which
        // allocate), the second is captured on the handle's private stream, every call from then on is one graph launch.
        static const bool step_graph = getenv("IDQN_STEP_GRAPH") && atoi(getenv("IDQN_STEP_GRAPH")) != 0;
        if (step_graph && flags == 0 && !h->rp) {  // (a replay-sourced step carries its slots as kernel arguments: not replayable)
            auto key = std::make_tuple(state_dev, next_state_dev, (const void*)action_dev, (const void*)reward_dev,
                                       (const void*)terminal_dev, (int)batch, (int)batch_mean_divisor,
                                       (const void*)h->is_weight, (const void*)h->td_abs);
            // bounded: a caller that hands over allocator-recycled buffers would otherwise instantiate a graph per pointer
            // tuple that ever recurs; past 32 keys everything is dropped and the cache starts again
            if (h->step_graphs.size() >= 32 && !h->step_graphs.count(key)) {
                // the last replay may still be running on the caller's stream: an executable graph is only destroyed idle
                IDQN_HIP_CHECK(hipStreamSynchronize(q));
                for (auto& g : h->step_graphs)
                    if (g.second.second) (void)hipGraphExecDestroy(g.second.second);
                h->step_graphs.clear();
            }
            auto& ent = h->step_graphs[key];
            if (ent.first++ == 0) return issue(q);
            if (!ent.second) {
                hipGraph_t graph = nullptr;
                if (!h->act_stream) IDQN_HIP_CHECK(hipStreamCreateWithFlags(&h->act_stream, hipStreamNonBlocking));
                IDQN_HIP_CHECK(hipStreamBeginCapture(h->act_stream, hipStreamCaptureModeRelaxed));
                rc = issue(h->act_stream);
                const hipError_t e = hipStreamEndCapture(h->act_stream, &graph);
                if (rc || e != hipSuccess) {  // nothing of a failed capture is kept
                    if (graph) (void)hipGraphDestroy(graph);
                    h->step_graphs.erase(key);
                    if (rc) return rc;
                    IDQN_HIP_CHECK(e);
                }
                const hipError_t ei = hipGraphInstantiate(&ent.second, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ei != hipSuccess) { h->step_graphs.erase(key); IDQN_HIP_CHECK(ei); }
            }
            IDQN_HIP_CHECK(hipGraphLaunch(ent.second, q));
            return IDQN_OK;
        }
        if ((rc = issue(q))) return rc;
    } else {
        FcArgs a;
        a.net = h->fc; a.online = h->online; a.target = h->target; a.grad = h->grad; a.P = h->L.head_stride;
        a.s = (const float*)state_dev; a.s2 = (const float*)next_state_dev; a.action = action_dev; a.reward = reward_dev;
        a.terminal = terminal_dev; a.gamma_n = h->gamma_n; a.B = batch; a.Bdiv = batch_mean_divisor; a.K = h->cfg.n_heads;
        a.ws = h->fc_ws; a.losses = h->losses; a.q_dbg = h->qdbg;
        a.count = h->count; a.bcinv = h->bcinv; a.adam_b1 = h->ad.b1; a.adam_b2 = h->ad.b2;
        a.cum = h->cum; a.finish_step = grads_only ? 0 : 1;
        a.is_weight = h->is_weight; a.td_abs = h->td_abs;
        a.s_stride = 0; a.din = nullptr; a.gm = GradMap{h->gP, h->g_w0_begin, h->g_w0_end, h->g_w0_base};
        if (profile && h->ev_used + 2 <= (int)h->ev.size()) IDQN_HIP_CHECK(hipEventRecord(h->ev[h->ev_used], q));
        const FcPlan& fp = h->fc_plan_;
        const size_t lds = (size_t)fp.floats * 4;
        const FcMfmaPlan& fm = h->fcm_plan_;
        const bool par = h->fcp_plan_.floats && batch <= 32;  // one launch: forwards side by side, Adam in the gradient epilogues
        if (par) hipLaunchKernelGGL(k_fc_step_par, dim3(h->cfg.n_heads), dim3(FCM_T), (size_t)h->fcp_plan_.floats * 4, q, a, h->fcp_plan_, h->ad,
                                    h->online, h->mu, h->nu, grads_only ? 0 : 1, debug_on("IDQN_FC_PROF") ? 1 : 0);
        else if (fm.floats && h->fcm_global_) hipLaunchKernelGGL(k_fc_step_mfma<true>, dim3(h->cfg.n_heads), dim3(FCM_T), (size_t)fm.floats * 4, q, a, fm.ldw, fm.drows, fm.w_floats);
        else if (fm.floats) hipLaunchKernelGGL(k_fc_step_mfma<false>, dim3(h->cfg.n_heads), dim3(FCM_T), (size_t)fm.floats * 4, q, a, fm.ldw, fm.drows, fm.w_floats);
        else if (fp.BS == 32) hipLaunchKernelGGL(k_fc_step_lds<32>, dim3(h->cfg.n_heads), dim3(FC_T), lds, q, a, fp.tw, fp.wfl);
        else if (fp.BS == 16) hipLaunchKernelGGL(k_fc_step_lds<16>, dim3(h->cfg.n_heads), dim3(FC_T), lds, q, a, fp.tw, fp.wfl);
        else if (fp.BS == 8) hipLaunchKernelGGL(k_fc_step_lds<8>, dim3(h->cfg.n_heads), dim3(FC_T), lds, q, a, fp.tw, fp.wfl);
        else hipLaunchKernelGGL(k_fc_step, dim3(h->cfg.n_heads), dim3(256), 0, q, a);
        if (profile && h->ev_used + 2 <= (int)h->ev.size()) {
            IDQN_HIP_CHECK(hipEventRecord(h->ev[h->ev_used + 1], q));
            h->ev_used += 2;
        }
        IDQN_HIP_CHECK(hipGetLastError());
        if (!grads_only && !par && (rc = launch_adam(h, 0, h->L.head_stride, 0, 0, false, q))) return rc;
    }
    return IDQN_OK;  // count += 1 and cum_losses += losses already happened in k_td_dh / k_fc_step (fused path)
}

// ---- i-IQN heads (extension; iqn_kernels.h, oracle/iqn_ref.py) --------------------------------------------------------
namespace {
// fraction blocks -> Dense_0 -> hidden + Dense_1 partials, for `V` virtual nets whose trunk features are psi [.][F * 32]
int iqn_heads_forward(idqn_handle_s* h, const float* const* wbase_v, int V, int K_for_index, const float* psi, const float* tau,
                      int B, hipStream_t q) {
    IqnWs& w = h->iqn;
    IqnCosArgs ca;
    // the embedding on the bf16 matrix cores from operands split once per step
    ca.tau = tau; ca.cosp = w.cosp; ca.cosa = w.cosa; ca.K = K_for_index; ca.N = w.N; ca.B = B;
    hipLaunchKernelGGL(k_iqn_cos, dim3((unsigned)(V * w.N)), dim3(256), 0, q, ca);
    tl_mark(h, q, "iqn cos features");
    {   // fractions per wave: 8 when that still leaves >= 8 waves per SIMD to overlap, else fewer 
        int per = 8;
        while (per > 1 && (w.N % per != 0)) --per;
        const dim3 grid((unsigned)cdiv(h->F / 32, 4), (unsigned)V, (unsigned)(w.N / per));
        {
            const int n_packed = std::min(V, 2 * K_for_index);  // virtual nets 2K .. 3K - 1 are the target nets again
            IqnWePackArgs pa;
            pa.wbase = wbase_v; pa.wep = w.wep; pa.we_off = w.off_we; pa.F = h->F;
            hipLaunchKernelGGL(k_iqn_we_pack, dim3((unsigned)(h->F / 32), (unsigned)n_packed), dim3(256), 0, q, pa);
            tl_mark(h, q, "iqn embedding kernel planes");
            IqnEmbed3Args e3;
            e3.cosp = w.cosp; e3.wep = w.wep; e3.wbase = wbase_v; e3.psi = psi; e3.x = w.xq; e3.be_off = w.off_be;
            e3.K = K_for_index; e3.N = w.N; e3.F = h->F; e3.n_packed = n_packed;
            // cos fragments through LDS, once per workgroup
            hipLaunchKernelGGL(k_iqn_embed3l, grid, dim3(256), 2 * 12288, q, e3);
        }
    }
    tl_mark(h, q, "iqn embedding x features");
    DenseFwdArgs d;
    d.in = w.xq; d.part = w.part; d.wbase = wbase_v; d.w_off = h->off_w0;
    d.n_nets = V; d.nb = w.N; d.NS = w.NS; d.n_jt = h->J / 128; d.F = h->F; d.J = h->J;
    d.n_items = (long)V * w.N * d.NS * d.n_jt;
    d.net_rot = 0; d.bb_inner = 0; d.nt_from = 0;
    // >= 8 fraction blocks per net: the tiled GEMM (iqn_gemm.h; fewer, or a count that is no multiple of 8: the per-block streaming kernel of the plain step)
    if (w.N % 8 == 0 && h->J % 256 == 0 && h->F % 16 == 0) {
        IqnD0FwdArgs g;
        g.x = w.xq; g.wbase = wbase_v; g.part = w.part; g.w_off = h->off_w0;
        g.V = V; g.nb = w.N; g.NS = w.NS; g.F = h->F; g.J = h->J; g.clk = (long long*)w.clk;
        const size_t lds = 2 * (size_t)IG_STAGE;
        const dim3 grid((unsigned)(V * (w.N / 8) * w.NS * (h->J / 256)));
        static LdsAttrMark attr;
        if (attr.needs(lds)) IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_fwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_iqn_d0_fwd<2>, grid, dim3(512), lds, q, g);
    } else {
        hipLaunchKernelGGL(k_dense0_fwd3, dim3(cdiv(d.n_items, 4)), dim3(256), 0, q, d);
    }
    tl_mark(h, q, "iqn dense0 fwd");
    HiddenArgs hi;
    hi.part = w.part; hi.wbase = wbase_v; hi.b0_off = h->off_b0; hi.w1_off = h->off_w1; hi.nb = w.N; hi.NS = w.NS;
    hi.J = h->J; hi.A = h->cfg.n_actions; hi.hbuf = w.hbuf; hi.qpart = w.qpart;
    hipLaunchKernelGGL(k_hidden, dim3(h->J / 32, (unsigned)(V * w.N)), dim3(256), 0, q, hi);
    tl_mark(h, q, "iqn hidden");
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}
}  // namespace

// iDQN.update_online_params (idqn.py:65-72) on the HBM frame ring: replay_buffer.py:215-230's sample() fused into the step
extern "C" int idqn_learn_on_replay(idqn_handle_t h, const uint8_t* frame_ring_dev, int64_t n_frames, int64_t frame_bytes,
                                    const int32_t* rows_dev, const int32_t* slots_host, int32_t batch, int32_t stack,
                                    int32_t batch_mean_divisor, uint32_t flags, void* stream) {
    IDQN_REQUIRE(h && frame_ring_dev && rows_dev && slots_host, "idqn_learn_on_replay: null pointer");
    IDQN_REQUIRE(h->cfg.arch == IDQN_ARCH_CNN && h->planes && !h->gc.on, "idqn_learn_on_replay: needs the cnn arch on the plane conv path");
    IDQN_REQUIRE(batch >= 1 && batch <= 256 && batch <= h->cfg.max_batch, "idqn_learn_on_replay: batch %d not in [1, min(256, %d)]", batch,
                 h->cfg.max_batch);
    IDQN_REQUIRE(stack == 4 && h->cfg.obs_c == 4 && frame_bytes == (int64_t)h->cfg.obs_h * h->cfg.obs_w && frame_bytes % 16 == 0 &&
                     n_frames >= 1 && ((uintptr_t)frame_ring_dev & 15) == 0,
                 "idqn_learn_on_replay: built for uint8 frames of obs_h x obs_w bytes (a multiple of 16), stack 4 == obs_c (got stack %d, "
                 "frame_bytes %ld, obs %d x %d x %d)", stack, (long)frame_bytes, h->cfg.obs_h, h->cfg.obs_w, h->cfg.obs_c);
    IDQN_REQUIRE(!(flags & (IDQN_F_STOP_AFTER_DENSE0 | IDQN_F_STOP_BEFORE_DENSE0_WGRAD)), "idqn_learn_on_replay: the IDQN_F_STOP_* flags are not supported");
    if (!h->rp_action) {
        const int mb = h->cfg.max_batch;
        float* f = nullptr;
        int rc = alloc_zero(&f, 3L * mb + 64, h, "replay scalars");
        if (rc) return rc;
        IDQN_HIP_CHECK(hipStreamSynchronize(nullptr));  // (the zero-fill runs on the null stream)
        h->rp_action = (int32_t*)f; h->rp_reward = f + mb; h->rp_terminal = (uint8_t*)(f + 2L * mb);
    }
    idqn_handle_s::ReplaySrc src;
    src.frames = frame_ring_dev; src.rows = rows_dev; src.n_frames = n_frames; src.frame_bytes = frame_bytes;
    memset(&src.slots, 0, sizeof(src.slots));
    memcpy(src.slots.slot, slots_host, (size_t)batch * 4);
    h->rp = &src;
    // (the state pointers only select the staging path; the replay source replaces them)
    const int rc = idqn_learn_on_batch(h, frame_ring_dev, frame_ring_dev, h->rp_action, h->rp_reward, h->rp_terminal, batch, batch_mean_divisor,
                                       flags, stream);
    h->rp = nullptr;
    return rc;
}

extern "C" int idqn_iqn_learn_on_batch(idqn_handle_t h, const void* state_dev, const void* next_state_dev,
                                       const int32_t* action_dev, const float* reward_dev, const uint8_t* terminal_dev,
                                       const float* tau_dev, int32_t batch, uint32_t flags, void* stream) {
    IDQN_REQUIRE(h && state_dev && next_state_dev && action_dev && reward_dev && terminal_dev && tau_dev,
                 "idqn_iqn_learn_on_batch: null pointer");
    IDQN_REQUIRE(h->iqn.N > 0, "idqn_iqn_learn_on_batch: the handle was created without quantile heads (cfg.n_quantiles)");
    IDQN_REQUIRE(batch >= 1 && batch <= 32 && batch <= h->cfg.max_batch, "idqn_iqn_learn_on_batch: batch %d not in [1, 32]", batch);
    IDQN_REQUIRE(!(flags & ~(IDQN_F_PROFILE | IDQN_F_PROFILE_ALL)), "idqn_iqn_learn_on_batch: only the profile flags are supported");
    // the quantile loss has no importance weights and writes no |TD|: refuse the combination instead of leaving stale priorities
    IDQN_REQUIRE(!h->is_weight && !h->td_abs,
                 "idqn_iqn_learn_on_batch: prioritized-replay buffers are set (idqn_set_per_buffers) but the quantile heads do not use them");
    hipStream_t q = (hipStream_t)stream;
    IqnWs& w = h->iqn;
    const int K = h->cfg.n_heads, A = h->cfg.n_actions;
    h->pend_B = 0; h->pend_stage = 0;
    h->tl_on = (flags & IDQN_F_PROFILE_ALL) != 0;
    if (h->tl_on && h->tl_ev.empty()) {
        h->tl_ev.resize(2048);
        h->tl_name.resize(2048);
        for (auto& e : h->tl_ev) IDQN_HIP_CHECK(hipEventCreate(&e));
    }
    tl_mark(h, q, nullptr);
    int rc;
    // trunk of the 2K nets (online on s, target on s'), then the fraction blocks of the 3K virtual nets
    if ((rc = cnn_forward(h, h->train, (const uint8_t*)state_dev, (const uint8_t*)next_state_dev, batch, q, false))) return rc;
    if ((rc = iqn_heads_forward(h, w.wbase_v, w.V, K, h->train.a3, tau_dev, batch, q))) return rc;
    IqnZArgs za;
    za.qpart = w.qpart; za.wbase = w.wbase_v; za.z = w.z; za.b1_off = h->off_b1; za.N = w.N; za.NJC = h->J / 32; za.A = A;
    hipLaunchKernelGGL(k_iqn_z, dim3((unsigned)(w.V * w.N)), dim3(256), 0, q, za);
    tl_mark(h, q, "iqn quantile values");
    IqnLossArgs la;
    la.z = w.z; la.K = K; la.N = w.N; la.A = A;
    la.B = batch; la.Bdiv = batch; la.action = action_dev; la.reward = reward_dev; la.terminal = terminal_dev; la.tau = tau_dev;
    la.gamma_n = h->gamma_n; la.dq = w.dq; la.losses = h->losses; la.count = h->count; la.cum = h->cum; la.finish_step = 1;
    la.dbg = w.dbg; la.gate_err = w.gate + 2L * K * cdiv(h->F, 256);
    hipLaunchKernelGGL(k_iqn_loss, dim3(K), dim3(256), (size_t)(2 * w.N * 32 + 32 * 32 + 8 * 32) * 4, q, la);
    tl_mark(h, q, "iqn quantile huber loss");
    const long w0n = h->g_w0_end - h->g_w0_begin;
    IqnDhArgs da;
    da.hbuf = w.hbuf; da.dq = w.dq; da.wbase = w.wbase_v; da.w1_off = h->off_w1;
    da.K = K; da.N = w.N; da.J = h->J; da.A = A; da.dh = w.dh; da.hpart = w.hpart;
    hipLaunchKernelGGL(k_iqn_dh, dim3(h->J / 32, K, w.HG), dim3(256), 0, q, da);
    tl_mark(h, q, "iqn dh + dense1 grads");
    {
        IqnHeadGradSumArgs hs;
        hs.hpart = w.hpart; hs.grad = h->grad; hs.gP = h->gP;
        hs.g_b0_off = h->off_b0 - w0n; hs.g_w1_off = h->off_w1 - w0n; hs.g_b1_off = h->off_b1 - w0n;
        hs.K = K; hs.J = h->J; hs.A = A; hs.QG = w.HG;
        hipLaunchKernelGGL(k_iqn_head_grad_sum, dim3((unsigned)cdiv((long)h->J * A + h->J + A, 256), K), dim3(256), 0, q, hs);
        tl_mark(h, q, "iqn head grads (group sums)");
    }
    bool wgrad_done = false;
    {   // W0 . dh for every fraction block (plain rows)
        DenseDgradArgs dd;
        memset(&dd, 0, sizeof(dd));
        dd.dh = w.dh; dd.raw = w.dx; dd.wbase = h->train.wbase; dd.w_off = h->off_w0;
        dd.K = K; dd.nb = w.N; dd.n_ft = h->F / 32; dd.F = h->F; dd.J = h->J; dd.C = h->conv[2].CO; dd.g = h->gda3;
        dd.n_items = (long)K * w.N * cdiv(dd.n_ft, 4);
        if (w.N % 8 == 0 && h->J % 16 == 0) {
            IqnD0DgradArgs g;
            g.dh = w.dh; g.wbase = h->train.wbase; g.dx = w.dx; g.w_off = h->off_w0; g.K = K; g.nb = w.N; g.F = h->F; g.J = h->J;
            const size_t lds = 2 * (size_t)IG_STAGE;
            const int n_d = K * (w.N / 8) * cdiv(h->F, 256);
            // the weight-gradient GEMM rides in the same launch
            const int n_groups = K * cdiv(h->F, 256), n_w1 = n_groups * (h->J / 256);
            w.d0_adam_done = false;
            if (w.g1 && w.bwd_items && n_d + n_w1 > cu_budget() && debug_int("IDQN_IQN_ADAM_FUSE", 1)) {
                // more than one round of items even with unsplit weight-gradient tiles: one split, Adam in its epilogue (k_iqn_d0_bwd_adam)
                IqnD0WgradArgs gw;
                gw.x = w.xq; gw.dh = w.dh; gw.g[0] = gw.g[1] = nullptr; gw.K = K; gw.nb = w.N; gw.F = h->F; gw.J = h->J; gw.KS = 1;
                gw.theta = debug_on("IDQN_IQN_ADAM_SKIP") ? nullptr : h->online; gw.mu = h->mu; gw.nu = h->nu; gw.bcinv = h->bcinv; gw.ad = h->ad; gw.P = h->L.head_stride; gw.w_off = h->off_w0; gw.dump = reinterpret_cast<const char*>(w.g1);  // (g1: K * F * J floats, unused on this path)
                IqnD0Gate gate;
                gate.arrived = w.gate; gate.passed = w.gate + n_groups; gate.err = w.gate + 2L * n_groups;
                gate.prof = (h->cprof && h->cprof_role == 11) ? (long long*)h->cprof : nullptr;
                static LdsAttrMark attr;
                if (attr.needs(lds)) IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_bwd_adam<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_iqn_d0_bwd_adam<2>, dim3((unsigned)w.bwd_blocks), dim3(512), lds, q, g, gw, gate, w.bwd_items);
                wgrad_done = true;
                w.d0_adam_done = true;
            } else if (w.g1 && w.N % 16 == 0) {
                IqnD0WgradArgs gw;
                memset(&gw, 0, sizeof(gw));
                gw.x = w.xq; gw.dh = w.dh; gw.g[0] = h->grad + h->g_w0_base; gw.g[1] = w.g1; gw.K = K; gw.nb = w.N; gw.F = h->F; gw.J = h->J; gw.KS = 2;
                static LdsAttrMark attr;
                if (attr.needs(lds)) IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_bwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_iqn_d0_bwd<2>, dim3((unsigned)(n_d + K * cdiv(h->F, 256) * gw.KS * (h->J / 256))), dim3(512), lds, q, g, n_d, gw);
                wgrad_done = true;
            } else {
                static LdsAttrMark attr;
                if (attr.needs(lds)) IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_dgrad<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_iqn_d0_dgrad<2>, dim3((unsigned)n_d), dim3(512), lds, q, g);
            }
        } else {
            hipLaunchKernelGGL((k_dense0_dgrad<4>), dim3((unsigned)dd.n_items), dim3(256), h->J * 32 * 4, q, dd);
        }
        tl_mark(h, q, w.d0_adam_done ? "iqn dense0 dgrad + wgrad + adam" : wgrad_done ? "iqn dense0 dgrad + wgrad" : "iqn dense0 dgrad");
    }
    const int QG = w.QG;
    {  // (the forward of this step packed the embedding kernels and wrote the cos planes)
        IqnEmbedBwd3Args e3;
        e3.cosp = w.cosp; e3.cosa = w.cosa; e3.wep = w.wep; e3.wbase = w.wbase_v; e3.psi = h->train.a3; e3.dx = w.dx;
        e3.dpsi = w.dpsi; e3.gpart = w.gpart; e3.be_off = w.off_be; e3.K = K; e3.N = w.N; e3.F = h->F;
        const size_t lds = 2 * 2 * 12 * 512 * 2;
        static LdsAttrMark attr;
        if (attr.needs(lds)) IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_embed_bwd3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_iqn_embed_bwd3, dim3((unsigned)cdiv(h->F / 32, 4), K, QG), dim3(256), lds, q, e3);
    }
    tl_mark(h, q, "iqn embedding backward");
    {
        IqnEmbedGradSumArgs gs;
        gs.gpart = w.gpart; gs.grad = h->grad; gs.gP = h->gP; gs.g_we_off = w.off_we - w0n; gs.g_be_off = w.off_be - w0n;
        gs.K = K; gs.F = h->F; gs.QG = QG;
        hipLaunchKernelGGL(k_iqn_embed_grad_sum, dim3((unsigned)cdiv(65L * h->F / 4, 256), K), dim3(256), 0, q, gs);
        tl_mark(h, q, "iqn embedding grads (group sums)");
    }
    {   // dL/dpsi (the fraction groups' partials, in group order) -> ReLU mask, bf16 planes, per-position sums: what the conv
        // backward of the plain step reads
        Da3FinalizeArgs fa;
        fa.dpart = w.dpsi; fa.a3 = h->train.a3; fa.da3 = h->da3; fa.da3p = h->da3p; fa.pb = h->pbuf[2];
        fa.n_rows = (long)K * h->F; fa.n_jt = QG; fa.F = h->F; fa.C = h->conv[2].CO; fa.K = K; fa.nb = 1; fa.g = h->gda3;
        hipLaunchKernelGGL(k_da3_finalize, dim3(cdiv(fa.n_rows * 8, 256)), dim3(256), 0, q, fa);
        tl_mark(h, q, "da3 finalize (sum, mask, planes)");
    }
    // Dense_0 weight gradient over the N fraction blocks of every head + Adam: inside the merged launch above, or as a GEMM with
    // two block splits and one streaming Adam pass (iqn_gemm.h), or (N not a multiple of 16) the plain step's fused kernel
    if (w.d0_adam_done) {
    } else if (w.g1 && w.N % 16 == 0) {
        const long n = (long)h->F * h->J;
        IqnD0WgradArgs g;
        memset(&g, 0, sizeof(g));
        g.x = w.xq; g.dh = w.dh; g.g[0] = h->grad + h->g_w0_base; g.g[1] = w.g1; g.K = K; g.nb = w.N; g.F = h->F; g.J = h->J; g.KS = 2;
        const size_t lds = 2 * (size_t)IG_STAGE;
        static LdsAttrMark attr;
        if (attr.needs(lds)) IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_iqn_d0_wgrad<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (!wgrad_done) {
            hipLaunchKernelGGL(k_iqn_d0_wgrad<2>, dim3((unsigned)(K * cdiv(h->F, 256) * g.KS * (h->J / 256))), dim3(512), lds, q, g);
            tl_mark(h, q, "iqn dense0 wgrad");
        }
        IqnD0AdamArgs aa;
        aa.g[0] = g.g[0]; aa.g[1] = g.g[1]; aa.theta = h->online; aa.mu = h->mu; aa.nu = h->nu; aa.bcinv = h->bcinv; aa.ad = h->ad;
        aa.P = h->L.head_stride; aa.w_off = h->off_w0; aa.n = n; aa.KS = g.KS;
        hipLaunchKernelGGL(k_iqn_d0_adam, dim3((unsigned)cdiv(n / 4, 256), (unsigned)K), dim3(256), 0, q, aa);
        tl_mark(h, q, "iqn dense0 adam");
        IDQN_HIP_CHECK(hipGetLastError());
    } else if ((rc = launch_dense0_wgrad(h, w.xq, w.dh, w.N, w.N, 0, (long)w.N * h->F * 32, (long)h->F * 32, 0, (long)w.N * h->J * 32,
                                  (long)h->J * 32, true, (flags & IDQN_F_PROFILE) != 0, q, false)))
        return rc;
    if ((rc = cnn_backward_rest(h, batch, true, q))) return rc;
    return launch_adam(h, 0, h->L.head_stride, h->off_w0, h->off_b0, true, q);
}

extern "C" int idqn_iqn_q_values(idqn_handle_t h, int32_t which, int32_t head, const void* states_dev, int32_t n,
                                 const float* tau_dev, float* q_out_dev, int32_t* action_out_dev, void* stream) {
    IDQN_REQUIRE(h && states_dev && tau_dev && q_out_dev, "idqn_iqn_q_values: null pointer");
    IDQN_REQUIRE(h->iqn.N > 0, "idqn_iqn_q_values: the handle was created without quantile heads (cfg.n_quantiles)");
    IDQN_REQUIRE(head >= 0 && head < h->cfg.n_heads && (which == 0 || which == 1), "idqn_iqn_q_values: bad head / which");
    IDQN_REQUIRE(n >= 1 && n <= 32, "idqn_iqn_q_values: n = %d, must be in [1, 32]", n);
    hipStream_t q = (hipStream_t)stream;
    IqnWs& w = h->iqn;
    const float* params = (which ? h->target : h->online) + (long)head * h->L.head_stride;
    h->infer.wbase = h->train.wbase + (which * h->cfg.n_heads + head);  // entry of the training table: no pointer upload
    h->infer_pbase = params;
    int rc;
    if ((rc = cnn_forward(h, h->infer, (const uint8_t*)states_dev, nullptr, n, q, false))) return rc;
    if ((rc = iqn_heads_forward(h, h->infer.wbase, 1, 1, h->infer.a3, tau_dev, n, q))) return rc;
    IqnQOutArgs qo;
    qo.qpart = w.qpart; qo.params = params; qo.b1_off = h->off_b1; qo.N = w.N; qo.NJC = h->J / 32; qo.A = h->cfg.n_actions;
    qo.n = n; qo.q_out = q_out_dev; qo.action = action_out_dev;
    hipLaunchKernelGGL(k_iqn_q_out, dim3(1), dim3(256), 0, q, qo);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int idqn_backward_rest(idqn_handle_t h, void* stream) {
    IDQN_REQUIRE(h && h->pend_B > 0, "idqn_backward_rest: no backward pass is waiting (IDQN_F_STOP_* first)");
    const int B = h->pend_B;
    if (h->pend_stage == 2) h->pend_B = 0;  // stage 1 keeps B for idqn_export / idqn_finish_step_factored
    return cnn_backward_rest(h, B, false, (hipStream_t)stream);
}

extern "C" int idqn_export_dense0_factors(idqn_handle_t h, float* a3_out_dev, float* dh_out_dev, void* stream) {
    IDQN_REQUIRE(h && a3_out_dev && dh_out_dev, "idqn_export_dense0_factors: null pointer");
    IDQN_REQUIRE(h->pend_stage == 1 && h->pend_B > 0, "idqn_export_dense0_factors: needs IDQN_F_STOP_BEFORE_DENSE0_WGRAD first");
    const long nb = cdiv(h->pend_B, 32), K = h->cfg.n_heads;
    // the online nets are the first K * nb slots of the activation buffer
    IDQN_HIP_CHECK(hipMemcpyAsync(a3_out_dev, h->train.a3, (size_t)K * nb * h->F * 32 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    IDQN_HIP_CHECK(hipMemcpyAsync(dh_out_dev, dh_of(h, (int)nb), (size_t)K * nb * h->J * 32 * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return IDQN_OK;
}

extern "C" int idqn_dense0_factors(idqn_handle_t h, float** factors_dev, int64_t* n_dh, int64_t* n_a3) {
    IDQN_REQUIRE(h && factors_dev && n_dh && n_a3, "idqn_dense0_factors: null pointer");
    IDQN_REQUIRE(h->pend_stage == 1 && h->pend_B > 0, "idqn_dense0_factors: needs IDQN_F_STOP_BEFORE_DENSE0_WGRAD first");
    const int nb = cdiv(h->pend_B, 32);
    *factors_dev = dh_of(h, nb);
    *n_dh = (int64_t)h->cfg.n_heads * nb * h->J * 32;
    *n_a3 = (int64_t)h->cfg.n_heads * nb * h->F * 32;
    return IDQN_OK;
}

extern "C" int idqn_finish_step_factored(idqn_handle_t h, const float* a3_all_dev, const float* dh_all_dev,
                                         int32_t nb_total, int32_t nb_inner, int64_t a3_outer, int64_t a3_head,
                                         int64_t a3_inner, int64_t dh_outer, int64_t dh_head, int64_t dh_inner,
                                         uint32_t phases, void* stream) {
    IDQN_REQUIRE(h && (phases & 3u) && !(phases & ~3u), "idqn_finish_step_factored: phases must be 1, 2 or 3");
    hipStream_t q = (hipStream_t)stream;
    int rc;
    if (phases & IDQN_FACTORED_DENSE0) {
        IDQN_REQUIRE(a3_all_dev && dh_all_dev, "idqn_finish_step_factored: null pointer");
        IDQN_REQUIRE(h->pend_stage == 1, "idqn_finish_step_factored: needs IDQN_F_STOP_BEFORE_DENSE0_WGRAD first");
        IDQN_REQUIRE(nb_total >= 1 && nb_inner >= 1 && nb_total % nb_inner == 0, "idqn_finish_step_factored: bad block counts");
        h->pend_stage = 3; h->pend_B = 0;
        if ((rc = launch_dense0_wgrad(h, a3_all_dev, dh_all_dev, nb_total, nb_inner, a3_outer, a3_head, a3_inner, dh_outer,
                                      dh_head, dh_inner, true, h->pend_profile, q)))
            return rc;
    }
    if (phases & IDQN_FACTORED_REST) {
        IDQN_REQUIRE(h->pend_stage == 3, "idqn_finish_step_factored: the Dense_0 phase has to come first");
        h->pend_stage = 0;
        // every other leaf, from grad_dev; count += 1 and cum_losses += losses ride in the same launch
        return launch_adam(h, 0, h->L.head_stride, h->off_w0, h->off_b0, false, q, h->cfg.arch == IDQN_ARCH_CNN);
    }
    return IDQN_OK;
}

int idqn_internal_dp_view(idqn_handle_t h, IdqnDpView* v) {  // (csrc/dp.hip)
    IDQN_REQUIRE(h && v, "null handle");
    IDQN_REQUIRE(h->cfg.arch == IDQN_ARCH_CNN && !h->gc.on && h->cfg.n_quantiles == 0,
                 "the data-parallel step is built for the MFMA cnn path of the i-DQN heads");
    v->grad = h->grad; v->losses = h->losses; v->n_small = (long)h->cfg.n_heads * h->gP + 64;
    v->K = h->cfg.n_heads; v->F = h->F; v->J = h->J;
    return IDQN_OK;
}

extern "C" int idqn_set_per_buffers(idqn_handle_t h, const float* weights_dev, float* td_abs_out_dev) {
    IDQN_REQUIRE(h, "idqn_set_per_buffers: null handle");
    h->is_weight = weights_dev;
    h->td_abs = td_abs_out_dev;
    return IDQN_OK;
}

extern "C" int idqn_apply_adam(idqn_handle_t h, void* stream) {
    IDQN_REQUIRE(h, "idqn_apply_adam: null handle");
    int rc = launch_adam(h, 0, h->L.head_stride, 0, 0, false, (hipStream_t)stream);
    if (rc) return rc;
    return step_epilogue(h, true, (hipStream_t)stream);
}

extern "C" int idqn_target_update(idqn_handle_t h, void* stream) {
    IDQN_REQUIRE(h, "idqn_target_update: null handle");
    hipStream_t q = (hipStream_t)stream;
    const long P = h->L.head_stride;
    const int K = h->cfg.n_heads;
    // idqn.py:78  target_params = params.copy()  -- a real copy, BEFORE the shift (idqn.py:80)
    IDQN_HIP_CHECK(hipMemcpyAsync(h->target, h->online, (size_t)K * P * 4, hipMemcpyDeviceToDevice, q));
    // idqn.py:13-17  params[k] <- params[k+1]; ascending k so every source is read before it is overwritten
    for (int k = 0; k + 1 < K; ++k)
        IDQN_HIP_CHECK(hipMemcpyAsync(h->online + (long)k * P, h->online + (long)(k + 1) * P, (size_t)P * 4,
                                      hipMemcpyDeviceToDevice, q));
    return IDQN_OK;
}

extern "C" int idqn_target_sync(idqn_handle_t h, void* stream) {
    IDQN_REQUIRE(h, "idqn_target_sync: null handle");
    const long P = h->L.head_stride;
    const int K = h->cfg.n_heads;
    // idqn.py:20-24  target[k] <- params[k-1] for k >= 1; target[0] stays frozen
    if (K > 1)
        IDQN_HIP_CHECK(hipMemcpyAsync(h->target + P, h->online, (size_t)(K - 1) * P * 4, hipMemcpyDeviceToDevice,
                                      (hipStream_t)stream));
    return IDQN_OK;
}

// argmax over the actions of each row, first maximum on ties (jnp.argmax, idqn.py:131)
__global__ void k_argmax_rows(const float* __restrict__ q, int n, int A, int32_t* __restrict__ out, volatile int32_t* mail,
                              unsigned* seq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int best = 0;
    float bv = q[(long)i * A];
    for (int ac = 1; ac < A; ++ac) {
        const float v = q[(long)i * A + ac];
        if (v > bv) { bv = v; best = ac; }
    }
    out[i] = best;
    if (mail && i == 0) {  // host mailbox of idqn_act_host (see k_act_head): the action, then the number that announces it
        const unsigned nn = seq[0] + 1u;
        seq[0] = nn;
        mail[0] = best;
        __threadfence_system();
        mail[1] = (int32_t)nn;
    }
}

static int q_values_impl(idqn_handle_t h, int32_t which, int32_t head, const void* states_dev, int32_t n,
                         float* q_out_dev, int32_t* action_out_dev, void* stream) {
    IDQN_REQUIRE(h && states_dev && q_out_dev, "idqn_q_values: null pointer");
    IDQN_REQUIRE(head >= 0 && head < h->cfg.n_heads && (which == 0 || which == 1), "idqn_q_values: bad head / which");
    IDQN_REQUIRE(n >= 1 && n <= 32, "idqn_q_values: n = %d, must be in [1, 32]", n);
    hipStream_t q = (hipStream_t)stream;
    const float* params = (which ? h->target : h->online) + (long)head * h->L.head_stride;
    if (h->gc.on) {  // general-shape cnn: trunk of the one net on the n states, then the generic dense forward
        const float* const* wb = h->train.wbase + (which * h->cfg.n_heads + head);
        int rc = gcnn_trunk(h, wb, 1, 1, n, (const uint8_t*)states_dev, (const uint8_t*)states_dev, h->gc.inf_act, q);
        if (rc) return rc;
        FcQArgs a;
        a.net = h->fc; a.params = params; a.s = h->gc.inf_act[2]; a.n = n; a.q_out = q_out_dev; a.ws = h->gc.inf_ws;
        hipLaunchKernelGGL(k_fc_q, dim3(1), dim3(256), 0, q, a);
        if (action_out_dev)
            hipLaunchKernelGGL(k_argmax_rows, dim3(1), dim3(64), 0, q, q_out_dev, n, h->cfg.n_actions, action_out_dev,
                               (volatile int32_t*)(h->act_use_mail ? h->act_mail_dev : nullptr), h->act_seq);
        IDQN_HIP_CHECK(hipGetLastError());
        return IDQN_OK;
    }
    if (h->cfg.arch == IDQN_ARCH_CNN && n == 1 && !act_generic() && h->J <= 512 && h->cfg.n_actions <= 32) {
        // one state: the latency path (act_kernels.h) -- pixels and parameter leaves as they are, five small launches
        const float* in = nullptr;
        int ih = h->cfg.obs_h, iw = h->cfg.obs_w;
        for (int i = 0; i < 3; ++i) {
            const ConvL& l = h->conv[i];
            ActConvArgs a;
            a.in_u8 = i == 0 ? (const uint8_t*)states_dev : nullptr; a.in = in; a.params = params; a.out = h->act_a[i];
            a.w_off = l.w_off; a.b_off = l.b_off; a.IH = ih; a.IW = iw; a.CI = l.CI; a.OH = l.OH; a.OW = l.OW; a.CO = l.CO;
            a.K = l.K; a.S = l.S; a.PLh = l.PLh; a.PLw = l.PLw;
            a.KS = i == 0 ? 8 : 16;  // lane slices: layer 0 has 64 units of 4 channels, the others K*K*CI/32 units of 32
            const int per = 256 / a.KS;
            const dim3 grid(cdiv((long)l.OH * l.OW * l.CO, per));
            const int units = l.K * l.K * (l.CI % 32 == 0 ? l.CI / 32 : l.CI / 4), upt = cdiv(units, a.KS);
            IDQN_REQUIRE(l.CI % 32 == 0 || (l.CI == 4 && upt <= 8), "acting path: Conv_%d has %d input channels", i, l.CI);
            if (l.CI % 32 != 0) hipLaunchKernelGGL((k_act_conv<4, 8>), grid, dim3(256), 0, q, a);
            else if (upt <= 1) hipLaunchKernelGGL((k_act_conv<32, 1>), grid, dim3(256), 0, q, a);
            else if (upt <= 2) hipLaunchKernelGGL((k_act_conv<32, 2>), grid, dim3(256), 0, q, a);
            else { IDQN_REQUIRE(upt <= 4, "acting path: %d units per lane slice", upt); hipLaunchKernelGGL((k_act_conv<32, 4>), grid, dim3(256), 0, q, a); }
            in = h->act_a[i]; ih = l.OH; iw = l.OW;
        }
        ActDenseArgs d;
        d.a3 = h->act_a[2]; d.params = params; d.part = h->act_part; d.w_off = h->off_w0; d.F = h->F; d.J = h->J;
        d.NRG = std::max(1, 256 / (h->J / 128));  // one workgroup per CU
        hipLaunchKernelGGL(k_act_dense0, dim3(d.NRG * (h->J / 128)), dim3(256), 0, q, d);
        ActHeadArgs ha;
        ha.part = h->act_part; ha.params = params; ha.b0_off = h->off_b0; ha.w1_off = h->off_w1; ha.b1_off = h->off_b1;
        ha.NP = d.NRG; ha.J = h->J; ha.A = h->cfg.n_actions; ha.q_out = q_out_dev; ha.action = action_out_dev;
        ha.mail = h->act_use_mail ? h->act_mail_dev : nullptr; ha.seq = h->act_seq;
        hipLaunchKernelGGL(k_act_head, dim3(1), dim3(1024), 0, q, ha);
        IDQN_HIP_CHECK(hipGetLastError());
        return IDQN_OK;
    }
    if (h->cfg.arch == IDQN_ARCH_CNN) {
        // the acting net's parameter pointer is entry (which * K + head) of the training table: no pointer upload
        h->infer.wbase = h->train.wbase + (which * h->cfg.n_heads + head);
        h->infer_pbase = params;
        (void)params;
        int rc = cnn_forward(h, h->infer, (const uint8_t*)states_dev, nullptr, n, q);
        if (rc) return rc;
        HiddenArgs hi;
        hi.part = h->infer.part; hi.wbase = h->infer.wbase; hi.b0_off = h->off_b0; hi.w1_off = h->off_w1; hi.nb = 1;
        hi.NS = h->infer.NS; hi.J = h->J; hi.A = h->cfg.n_actions; hi.hbuf = h->infer_hbuf; hi.qpart = h->infer_qpart;
        hipLaunchKernelGGL(k_hidden, dim3(h->J / 32, 1), dim3(256), 0, q, hi);
        QOutArgs qo;
        qo.qpart = h->infer_qpart; qo.wbase = h->infer.wbase; qo.b1_off = h->off_b1; qo.NJC = h->J / 32;
        qo.A = h->cfg.n_actions; qo.n = n; qo.q_out = q_out_dev; qo.action = action_out_dev;
        hipLaunchKernelGGL(k_q_out, dim3(1), dim3(256), 0, q, qo);
    } else {
        FcQArgs a;
        a.net = h->fc; a.params = params; a.s = (const float*)states_dev; a.n = n; a.q_out = q_out_dev;
        a.ws = h->fc_ws + (long)h->cfg.n_heads * ((long)(h->fc.L + 3) * h->cfg.max_batch * h->fc.dmax + 2 * h->cfg.max_batch);
        if (n == 1 && h->fc.dmax <= FC_MAX_WIDTH) hipLaunchKernelGGL(k_fc_q1, dim3(1), dim3(512), 0, q, a);  // acting: one state
        else hipLaunchKernelGGL(k_fc_q, dim3(1), dim3(256), 0, q, a);
        if (action_out_dev)
            hipLaunchKernelGGL(k_argmax_rows, dim3(1), dim3(64), 0, q, q_out_dev, n, h->cfg.n_actions, action_out_dev,
                               (volatile int32_t*)(h->act_use_mail ? h->act_mail_dev : nullptr), h->act_seq);
    }
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

extern "C" int idqn_q_values(idqn_handle_t h, int32_t which, int32_t head, const void* states_dev, int32_t n,
                             float* q_out_dev, void* stream) {
    return q_values_impl(h, which, head, states_dev, n, q_out_dev, nullptr, stream);
}

extern "C" int idqn_best_action(idqn_handle_t h, int32_t which, int32_t head, const void* states_dev, int32_t n,
                                float* q_out_dev, int32_t* action_out_dev, void* stream) {
    IDQN_REQUIRE(action_out_dev, "idqn_best_action: null pointer");
    return q_values_impl(h, which, head, states_dev, n, q_out_dev, action_out_dev, stream);
}

// select_action's greedy branch for ONE state that lives on the host (slimdqn/sample_collection/utils.py:8-21: the
// reference uploads the state, runs best_action and blocks on `.item()`): upload from pinned memory, the five launches of
// the single-state path, the action back into pinned memory, one stream synchronisation -- replayed as ONE hipGraph per
// (net, buffers) after the first call (seven eager API calls cost more host time than the ~35 us of GPU work).
static int act_host_wait(idqn_handle_t h, int32_t* action_host_pinned, hipStream_t q);
// wait = false: the launch only (idqn_act_host_begin); the result is collected by act_host_wait (idqn_act_host_end)
static int act_host_impl(idqn_handle_t h, int32_t which, int32_t head, const void* state_host_pinned, float* q_out_dev,
                         int32_t* action_host_pinned, void* stream, bool wait) {
    IDQN_REQUIRE(h && state_host_pinned && q_out_dev && action_host_pinned, "idqn_act_host: null pointer");
    IDQN_REQUIRE(head >= 0 && head < h->cfg.n_heads && (which == 0 || which == 1), "idqn_act_host: bad head / which");
    hipStream_t q = (hipStream_t)stream;
    const bool cnn = h->cfg.arch == IDQN_ARCH_CNN;
    // bytes of one state: uint8 pixels (cnn) or float32 features (fc)
    const size_t E = cnn ? (size_t)h->cfg.obs_h * h->cfg.obs_w * h->cfg.obs_c : (size_t)h->fc.d[0] * 4;
    if (!h->act_state) {  // (fc handles: a few floats)
        IDQN_HIP_CHECK(hipMalloc((void**)&h->act_state, E + 64));
        h->owned.push_back((void*)h->act_state);
        IDQN_HIP_CHECK(hipMalloc((void**)&h->act_action, 64));
        h->owned.push_back((void*)h->act_action);
    }
    // The single-state path ends in a kernel that can write the action straight into mapped host memory, followed by a
    // sequence number the host polls (IDQN_ACT_POLL=0: a device-to-host copy and a stream synchronisation instead).
    static const bool no_poll = getenv("IDQN_ACT_POLL") && atoi(getenv("IDQN_ACT_POLL")) == 0;
    const bool poll = !no_poll && ((cnn && !h->gc.on) ? (!act_generic() && h->J <= 512 && h->cfg.n_actions <= 32) : true);
    if (poll && !h->act_mail) {
        IDQN_HIP_CHECK(hipHostMalloc((void**)&h->act_mail, 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(h->act_mail, 0, 64);
        IDQN_HIP_CHECK(hipHostGetDevicePointer((void**)&h->act_mail_dev, h->act_mail, 0));
        IDQN_HIP_CHECK(hipMalloc((void**)&h->act_seq, 4));
        IDQN_HIP_CHECK(hipMemset(h->act_seq, 0, 4));
        IDQN_HIP_CHECK(hipStreamSynchronize(nullptr));  // (the memset runs on the null stream, which does not order against qs)
        h->owned.push_back((void*)h->act_seq);
    }
    // The MLP's state is a few floats: the kernel reads them from the caller's pinned buffer itself (one copy node fewer in front of a
    // 3 us kernel); pixels are read many times over by the first conv layer and are copied to HBM first.
    const void* direct = nullptr;
    if (!cnn && E <= 256) {
        void* dptr = nullptr;
        if (hipHostGetDevicePointer(&dptr, const_cast<void*>(state_host_pinned), 0) == hipSuccess) direct = dptr;
        else (void)hipGetLastError();
    }
    auto issue = [&](hipStream_t qs) -> int {
        if (!direct) IDQN_HIP_CHECK(hipMemcpyAsync(h->act_state, state_host_pinned, E, hipMemcpyHostToDevice, qs));
        h->act_use_mail = poll;
        int rc = q_values_impl(h, which, head, direct ? direct : (const void*)h->act_state, 1, q_out_dev, h->act_action, (void*)qs);
        h->act_use_mail = false;
        if (rc) return rc;
        if (!poll) IDQN_HIP_CHECK(hipMemcpyAsync(action_host_pinned, h->act_action, 4, hipMemcpyDeviceToHost, qs));
        return IDQN_OK;
    };
    static const bool use_graph = !(getenv("IDQN_ACT_GRAPH") && atoi(getenv("IDQN_ACT_GRAPH")) == 0);
    int rc = IDQN_OK;
    if (use_graph) {
        auto key = std::make_tuple(which * h->cfg.n_heads + head, state_host_pinned, (void*)q_out_dev, (void*)action_host_pinned);
        auto it = h->act_graphs.find(key);
        if (it == h->act_graphs.end()) {
            // (captured on a stream of the handle's own: the caller's may be the legacy default stream, which cannot
            // capture; the instantiated graph is then launched on the caller's stream like any other work)
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (!h->act_stream) IDQN_HIP_CHECK(hipStreamCreateWithFlags(&h->act_stream, hipStreamNonBlocking));
            IDQN_HIP_CHECK(hipStreamBeginCapture(h->act_stream, hipStreamCaptureModeRelaxed));
            rc = issue(h->act_stream);
            const hipError_t e = hipStreamEndCapture(h->act_stream, &graph);
            if (rc) return rc;
            IDQN_HIP_CHECK(e);
            IDQN_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            IDQN_HIP_CHECK(hipGraphDestroy(graph));
            it = h->act_graphs.emplace(key, exec).first;
        }
        IDQN_HIP_CHECK(hipGraphLaunch(it->second, q));
    } else {
        rc = issue(q);
        if (rc) return rc;
    }
    h->act_pending = poll ? 1 : 2;  // 1: the mailbox announces the action; 2: a device-to-host copy is queued, synchronise
    if (!wait) return IDQN_OK;
    return act_host_wait(h, action_host_pinned, q);
}

static int act_host_wait(idqn_handle_t h, int32_t* action_host_pinned, hipStream_t q) {
    IDQN_REQUIRE(h->act_pending != 0, "idqn_act_host_end: no acting launch is pending");
    const bool poll = h->act_pending == 1;
    h->act_pending = 0;
    if (poll) {
        const unsigned want = ++h->act_expected;
        volatile int32_t* mail = h->act_mail;
        bool seen = false;
        for (long spin = 0; spin < (1L << 34); ++spin) {  // far longer than any step queued in front of the launch
            if ((unsigned)mail[1] == want) { seen = true; break; }
            __builtin_ia32_pause();
            if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(q) != hipErrorNotReady) {  // the stream ran dry (or failed) without the number
                seen = (unsigned)mail[1] == want;
                break;
            }
        }
        if (!seen) {  // resynchronise the two counters, then report
            IDQN_HIP_CHECK(hipStreamSynchronize(q));
            IDQN_HIP_CHECK(hipMemcpy(&h->act_expected, h->act_seq, 4, hipMemcpyDeviceToHost));
            IDQN_REQUIRE(false, "idqn_act_host: the acting launch finished without delivering its action");
        }
        *action_host_pinned = mail[0];
        return IDQN_OK;
    }
    IDQN_HIP_CHECK(hipStreamSynchronize(q));
    return IDQN_OK;
}

extern "C" int idqn_act_host(idqn_handle_t h, int32_t which, int32_t head, const void* state_host_pinned, float* q_out_dev,
                             int32_t* action_host_pinned, void* stream) {
    IDQN_REQUIRE(h && h->act_pending == 0, "idqn_act_host: null handle, or an idqn_act_host_begin is still waiting for its _end");
    return act_host_impl(h, which, head, state_host_pinned, q_out_dev, action_host_pinned, stream, true);
}
extern "C" int idqn_act_host_begin(idqn_handle_t h, int32_t which, int32_t head, const void* state_host_pinned, float* q_out_dev,
                                   int32_t* action_host_pinned, void* stream) {
    IDQN_REQUIRE(h && h->act_pending == 0, "idqn_act_host_begin: null handle, or an acting launch is already pending");
    return act_host_impl(h, which, head, state_host_pinned, q_out_dev, action_host_pinned, stream, false);
}
extern "C" int idqn_act_host_end(idqn_handle_t h, int32_t* action_host_pinned, void* stream) {
    IDQN_REQUIRE(h && action_host_pinned, "idqn_act_host_end: null pointer");
    return act_host_wait(h, action_host_pinned, (hipStream_t)stream);
}

extern "C" int idqn_debug_buffer(idqn_handle_t h, const char* name, void** ptr_dev, int64_t* nbytes) {
    IDQN_REQUIRE(h && name && ptr_dev && nbytes, "idqn_debug_buffer: null pointer");
    if (!strcmp(name, "dh") && h->cfg.arch == IDQN_ARCH_CNN && h->dh_nb > 0) {  // placed per batch size (dh_of)
        const int nb = h->dh_nb;
        *ptr_dev = (void*)dh_of(h, nb);
        *nbytes = (int64_t)h->cfg.n_heads * nb * h->J * 32 * 4;
        return IDQN_OK;
    }
    for (auto& e : h->dbg)
        if (e.first == name) {
            *ptr_dev = e.second.first;
            *nbytes = e.second.second;
            return IDQN_OK;
        }
    IDQN_REQUIRE(false, "idqn_debug_buffer: no buffer named '%s'", name);
}

extern "C" int idqn_profile_table(idqn_handle_t h, char* out, int32_t out_bytes) {
    IDQN_REQUIRE(h && out && out_bytes > 0, "idqn_profile_table: null pointer");
    std::vector<std::pair<std::string, std::pair<double, int>>> rows;  // first-seen order
    for (int i = 1; i < h->tl_used; ++i) {
        if (!h->tl_name[i]) continue;  // a step's start marker: nothing was launched before it
        float ms = 0;
        IDQN_HIP_CHECK(hipEventSynchronize(h->tl_ev[i]));
        IDQN_HIP_CHECK(hipEventElapsedTime(&ms, h->tl_ev[i - 1], h->tl_ev[i]));
        size_t r = 0;
        for (; r < rows.size(); ++r)
            if (rows[r].first == h->tl_name[i]) break;
        if (r == rows.size()) rows.push_back({h->tl_name[i], {0.0, 0}});
        rows[r].second.first += ms;
        rows[r].second.second += 1;
    }
    std::string txt;
    char line[160];
    for (auto& r : rows) {
        snprintf(line, sizeof line, "%s\t%.3f\t%d\n", r.first.c_str(), 1e3 * r.second.first / r.second.second, r.second.second);
        txt += line;
    }
    snprintf(out, (size_t)out_bytes, "%s", txt.c_str());
    h->tl_used = 0;
    return IDQN_OK;
}

extern "C" int idqn_profile_read(idqn_handle_t h, double* mean_ms, int32_t* n_launches, char* kernel_name) {
    IDQN_REQUIRE(h && mean_ms && n_launches, "idqn_profile_read: null pointer");
    double total = 0;
    int n = h->ev_used / 2;
    for (int i = 0; i < n; ++i) {
        float ms = 0;
        IDQN_HIP_CHECK(hipEventSynchronize(h->ev[2 * i + 1]));
        IDQN_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
        total += ms;
    }
    *mean_ms = n ? total / n : 0.0;
    *n_launches = n;
    if (kernel_name) snprintf(kernel_name, 64, "%s", h->dominant);
    h->ev_used = 0;
    return IDQN_OK;
}
