// Two conv backward roles in ONE launch: the data gradient of a layer (k_cfwd's body) on the first n_f workgroups, the
// weight gradient of a layer (k_cwgrad's body) on the rest -- both read the same dy, neither reads the other's result.
// Why: these launches are bound by what happens OUTSIDE their MFMA loops (argument loads, the first cold LDS fill, the
// epilogue's write burst: 8-9 of 15-20 us, the same whether a workgroup owns 2 positions or 5).  Planned for half the
// chip each they take only 5-25 % longer (dgrad Conv_2 19.7 -> 20 us on 125 workgroups, wgrad Conv_2 16.6 -> 19.0 on 120),
// so side by side the pair costs about what the slower one costs alone.  One workgroup per CU, all co-resident.
#include <algorithm>
#include <cstdlib>

#include "convp_fwd_body.h"
#include "convp_wgrad_body.h"
#include "dense0_update.h"

namespace {

template <int NPA, int CT, int NQ, int NT, int WNPX, int WCT, int WNTW, int WPG>
__global__ __launch_bounds__(512) void k_cpair(CFwdArgs f, unsigned f_stage, int ring, unsigned mask_off, int n_f, long long* prof,
                                               CWgradArgs w, unsigned w_stage, int MT, int split_xcd) {
    constexpr int KA = (int)(sizeof(CFwdArgs) + sizeof(CWgradArgs)) + 64;
    warm_kernargs<(KA < 1024 ? KA : 1024)>();
    // Both roles on EVERY XCD (blocks go to XCDs round-robin: XCD x = blockIdx % 8 holds slots blockIdx / 8): the first
    // f(x) slots of XCD x run the data gradient, the rest the weight gradient, each role numbered XCD-contiguously
    // (neighbouring items share inputs through that XCD's L2).  With the roles on separate XCDs the longer one had only
    // half of the chip's L2 / fabric ports for its staging traffic.
    const int n = (int)gridDim.x, b = (int)blockIdx.x, x = b & 7, slot = b >> 3;
    const int qf = n_f >> 3, rf = n_f & 7, fx = qf + (x < rf ? 1 : 0);
    if (!split_xcd) {
        const int v = xcd_contiguous_id();
        if (v < n_f) cfwd_body<NPA, CT, NQ, NT>(f, f_stage, ring, mask_off, prof, v, n_f);
        else cwgrad_body<WNPX, WCT, WNTW, WPG>(w, w_stage, MT, v - n_f);
    } else if (slot < fx) {
        cfwd_body<NPA, CT, NQ, NT>(f, f_stage, ring, mask_off, prof, x * qf + min(x, rf) + slot, n_f);
    } else {
        const int n_w = n - n_f, qn = n >> 3, rn = n & 7;
        // weight-gradient blocks of the XCDs before x: their block counts minus their data-gradient blocks
        const int before = (x * qn + min(x, rn)) - (x * qf + min(x, rf));
        (void)n_w;
        cwgrad_body<WNPX, WCT, WNTW, WPG>(w, w_stage, MT, before + (slot - fx));
    }
}


template <int NPA, int CT, int NQ, int NT, int WNPX, int WCT, int WNTW, int WPG>
int launch_pair(const CFwdArgs& f, int n_f, size_t f_stage, int ring, size_t f_lds, const CWgradArgs& w, int MT, int n_w, size_t w_lds,
                hipStream_t q, long long* prof) {
    const unsigned mask_off = (unsigned)convp_fwd_mask_off(f_stage, NT, ring, f.out3 != nullptr, f.out_f32 != nullptr);
    const size_t lds = std::max(f_lds, w_lds + 2048);
    // every XCD needs at least as many blocks as data-gradient blocks: true whenever the weight gradient has >= 8 items
    static const bool role_xcds = false;  // A/B switch: roles on separate XCDs
    const int split_xcd = (!role_xcds && n_w >= 8) ? 1 : 0;
    static LdsAttrMark attr;  // per instantiation
    if (attr.needs(lds)) {
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_cpair<NPA, CT, NQ, NT, WNPX, WCT, WNTW, WPG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL((k_cpair<NPA, CT, NQ, NT, WNPX, WCT, WNTW, WPG>), dim3((unsigned)(n_f + n_w)), dim3(512), lds, q, f,
                       (unsigned)f_stage, ring, mask_off, n_f, prof, w, (unsigned)(w_lds / 2), MT, split_xcd);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

}  // namespace

// The pairs that are built: the Nature-CNN widths [32, 64, 64] at the tile counts one-batch-block plans give (K = 5: NT 3 / NT 4;
// fewer heads: fewer tiles per wave, down to NT 1 for plain DQN)
//   Conv_2 data gradient <3, 2, 3, NT 1..3>  beside  Conv_2 weight gradient <3, 2, NTW 3, PG 2>
//   Conv_1 data gradient <3, 1, 2, NT 1..4>  beside  Conv_1 weight gradient <3, 2, NTW 2, PG 2>
// anything else runs as two launches.
bool convp_pair_built(int NPA, int CT, int NQ, int NT, int WNPX, int WCT, int WNTW, int WPG) {
    if (NPA == 3 && CT == 2 && NQ == 3 && NT >= 1 && NT <= 3 && WNPX == 3 && WCT == 2 && WNTW == 3 && WPG == 2) return true;
    if (NPA == 3 && CT == 1 && NQ == 2 && NT >= 1 && NT <= 4 && WNPX == 3 && WCT == 2 && WNTW == 2 && WPG == 2) return true;
    return false;
}

int convp_launch_pair(const CFwdArgs& f, int NPA, int CT, int NQ, int NT, int n_f, size_t f_stage, int ring, size_t f_lds,
                      const CWgradArgs& w, int WNPX, int MT, int WCT, int n_w, size_t w_lds, hipStream_t q, long long* prof) {
    IDQN_REQUIRE(n_f + n_w <= 256, "conv pair: %d + %d workgroups do not fit one per CU", n_f, n_w);
    IDQN_REQUIRE(f_lds <= 160 * 1024 && w_lds + 2048 <= 160 * 1024, "conv pair: %zu / %zu bytes of LDS per workgroup", f_lds, w_lds + 2048);
    const int ntw = (MT * WCT + 3) / 4;
    if (NPA == 3 && CT == 2 && NQ == 3 && WNPX == 3 && WCT == 2 && ntw == 3 && w.PG == 2) {
        if (NT == 3) return launch_pair<3, 2, 3, 3, 3, 2, 3, 2>(f, n_f, f_stage, ring, f_lds, w, MT, n_w, w_lds, q, prof);
        if (NT == 2) return launch_pair<3, 2, 3, 2, 3, 2, 3, 2>(f, n_f, f_stage, ring, f_lds, w, MT, n_w, w_lds, q, prof);
        if (NT == 1) return launch_pair<3, 2, 3, 1, 3, 2, 3, 2>(f, n_f, f_stage, ring, f_lds, w, MT, n_w, w_lds, q, prof);
    }
    if (NPA == 3 && CT == 1 && NQ == 2 && WNPX == 3 && WCT == 2 && ntw == 2 && w.PG == 2) {
        if (NT == 4) return launch_pair<3, 1, 2, 4, 3, 2, 2, 2>(f, n_f, f_stage, ring, f_lds, w, MT, n_w, w_lds, q, prof);
        if (NT == 3) return launch_pair<3, 1, 2, 3, 3, 2, 2, 2>(f, n_f, f_stage, ring, f_lds, w, MT, n_w, w_lds, q, prof);
        if (NT == 2) return launch_pair<3, 1, 2, 2, 3, 2, 2, 2>(f, n_f, f_stage, ring, f_lds, w, MT, n_w, w_lds, q, prof);
        if (NT == 1) return launch_pair<3, 1, 2, 1, 3, 2, 2, 2>(f, n_f, f_stage, ring, f_lds, w, MT, n_w, w_lds, q, prof);
    }
    IDQN_REQUIRE(false, "conv pair: this combination is not built (convp_pair_built says which are)");
}
