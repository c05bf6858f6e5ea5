// Plane-layout convolution, forward and data gradient: kernels and launcher (the workgroup body is convp_fwd_body.h).
#include <algorithm>
#include <cstdlib>

#include "convp_fwd_body.h"

namespace {

template <int NPA, int CT, int NQ, int NT>
__global__ __launch_bounds__(512) void k_cfwd(CFwdArgs a, unsigned stage_bytes, int ring, unsigned mask_off, long long* prof) {
    warm_kernargs<sizeof(CFwdArgs)>();
    cfwd_body<NPA, CT, NQ, NT>(a, stage_bytes, ring, mask_off, prof, xcd_contiguous_id(), (int)gridDim.x);
}

template <int NPA, int CT, int NQ, int NT>
int launch_one(const CFwdArgs& a, int n_items, size_t stage_bytes, int ring, size_t lds_bytes, hipStream_t q, long long* prof) {
    const unsigned mask_off = (unsigned)convp_fwd_mask_off(stage_bytes, NT, ring, a.out3 != nullptr, a.out_f32 != nullptr);
    static LdsAttrMark attr;  // per instantiation
    if (attr.needs(lds_bytes)) {
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_cfwd<NPA, CT, NQ, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    hipLaunchKernelGGL((k_cfwd<NPA, CT, NQ, NT>), dim3((unsigned)n_items), dim3(512), lds_bytes, q, a, (unsigned)stage_bytes, ring, mask_off, prof);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

template <int NPA, int CT, int NQ>
int launch_nt(const CFwdArgs& a, int NT, int n_items, size_t stage_bytes, int ring, size_t lds_bytes, hipStream_t q, long long* prof) {
    switch (NT) {
        case 1: return launch_one<NPA, CT, NQ, 1>(a, n_items, stage_bytes, ring, lds_bytes, q, prof);
        case 2: return launch_one<NPA, CT, NQ, 2>(a, n_items, stage_bytes, ring, lds_bytes, q, prof);
        case 3: return launch_one<NPA, CT, NQ, 3>(a, n_items, stage_bytes, ring, lds_bytes, q, prof);
        default: break;
    }
    if (CT == 1) {
        switch (NT) {
            case 4: return launch_one<NPA, 1, NQ, 4>(a, n_items, stage_bytes, ring, lds_bytes, q, prof);
            case 5: return launch_one<NPA, 1, NQ, 5>(a, n_items, stage_bytes, ring, lds_bytes, q, prof);
            case 6: return launch_one<NPA, 1, NQ, 6>(a, n_items, stage_bytes, ring, lds_bytes, q, prof);
            default: break;
        }
    }
    IDQN_REQUIRE(false, "plane conv: %d tiles per wave with %d channel tiles is not built", NT, CT);
}

}  // namespace

int convp_fwd_max_nt(int CT) { return CT == 1 ? 6 : 3; }

// two stage buffers, behind them the data gradient's mask tiles (2 KB per tile and wave); the epilogue turns every tile
// around in 10 KB per wave of the (then free) stage buffers
int convp_fwd_ring(size_t stage_bytes, int NT, int epilogue, size_t budget) {  // three stage buffers when they fit
    static const int forced = (0);
    if (forced == 2 || forced == 3) return forced;
    return 3 * stage_bytes + (epilogue == 1 ? (size_t)4 * NT * 2048 : 0) <= budget ? 3 : 2;
}
// LDS map: [ring stage buffers | ...] reused by the epilogue as [8 turn-around tiles | hand-off tiles]; the data gradient's
// mask tiles sit behind whichever of the two is larger
size_t convp_fwd_mask_off(size_t stage_bytes, int NT, int ring, bool planes_out, bool f32_out) {
    const size_t rsz = (planes_out ? 6144 : 0) + (f32_out ? 4096 : 0), nh = NT - (NT + 1) / 2;
    return std::max(ring * stage_bytes, 8 * rsz + 4 * nh * 4096);
}
size_t convp_fwd_lds(size_t stage_bytes, int NT, int epilogue, int ring, bool planes_out, bool f32_out) {
    return convp_fwd_mask_off(stage_bytes, NT, ring, planes_out, f32_out) + (epilogue == 1 ? (size_t)4 * NT * 2048 : 0);
}

int convp_launch_fwd(const CFwdArgs& a, int NPA, int CT, int NQ, int NT, int n_items, size_t stage_bytes, int ring,
                     size_t lds_bytes, hipStream_t q, long long* prof) {
    IDQN_REQUIRE(lds_bytes <= 160 * 1024, "plane conv: %zu bytes of LDS per workgroup", lds_bytes);
    if (NPA == 1 && NQ == 2) return CT == 1 ? launch_nt<1, 1, 2>(a, NT, n_items, stage_bytes, ring, lds_bytes, q, prof) : launch_nt<1, 2, 2>(a, NT, n_items, stage_bytes, ring, lds_bytes, q, prof);
    if (NPA == 3 && NQ == 2) return CT == 1 ? launch_nt<3, 1, 2>(a, NT, n_items, stage_bytes, ring, lds_bytes, q, prof) : launch_nt<3, 2, 2>(a, NT, n_items, stage_bytes, ring, lds_bytes, q, prof);
    if (NPA == 3 && NQ == 3) return CT == 1 ? launch_nt<3, 1, 3>(a, NT, n_items, stage_bytes, ring, lds_bytes, q, prof) : launch_nt<3, 2, 3>(a, NT, n_items, stage_bytes, ring, lds_bytes, q, prof);
    if (NPA == 3 && NQ == 4) return CT == 1 ? launch_nt<3, 1, 4>(a, NT, n_items, stage_bytes, ring, lds_bytes, q, prof) : launch_nt<3, 2, 4>(a, NT, n_items, stage_bytes, ring, lds_bytes, q, prof);
    IDQN_REQUIRE(false, "plane conv: no kernel for %d planes, %d taps per superstep", NPA, NQ);
}
