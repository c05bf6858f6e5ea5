// Plane-layout conv weight gradient: kernels and launcher (the workgroup body is convp_wgrad_body.h).
#include <algorithm>

#include "convp_wgrad_body.h"
#include "dense0_update.h"

namespace {

template <int NPX, int CT, int NTW, int PG>
__global__ __launch_bounds__(512) void k_cwgrad(CWgradArgs a, unsigned stage_bytes, int MT) {
    warm_kernargs<sizeof(CWgradArgs)>();
    cwgrad_body<NPX, CT, NTW, PG>(a, stage_bytes, MT, xcd_contiguous_id());
}


template <int NPX, int CT, int NTW, int PG>
int launch_one(const CWgradArgs& a, int MT, int n_items, size_t lds_bytes, hipStream_t q) {
    static LdsAttrMark attr;  // per instantiation
    if (attr.needs(lds_bytes)) {
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_cwgrad<NPX, CT, NTW, PG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes + 2048));
    }
    hipLaunchKernelGGL((k_cwgrad<NPX, CT, NTW, PG>), dim3((unsigned)n_items), dim3(512), lds_bytes + 2048, q, a, (unsigned)(lds_bytes / 2), MT);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

}  // namespace

// MT = 32-row tiles on the M side of one workgroup (KW * CI / 32, Conv_0: its 8 kernel rows)
int convp_launch_wgrad(const CWgradArgs& a, int NPX, int MT, int CT, int n_items, size_t lds_bytes, hipStream_t q) {
    IDQN_REQUIRE(lds_bytes + 2048 <= 160 * 1024, "plane wgrad: %zu bytes of LDS per workgroup", lds_bytes + 2048);
    const int ntw = (MT * CT + 3) / 4;
    if (NPX == 1) {
        if (CT == 1 && ntw == 2 && a.PG == 4) return launch_one<1, 1, 2, 4>(a, MT, n_items, lds_bytes, q);
        if (CT == 2 && ntw == 4 && a.PG == 2) return launch_one<1, 2, 4, 2>(a, MT, n_items, lds_bytes, q);
    } else if (a.PG == 2) {
        if (CT == 1) switch (ntw) {
            case 1: return launch_one<3, 1, 1, 2>(a, MT, n_items, lds_bytes, q);
            case 2: return launch_one<3, 1, 2, 2>(a, MT, n_items, lds_bytes, q);
            default: break;
        }
        else switch (ntw) {
            case 2: return launch_one<3, 2, 2, 2>(a, MT, n_items, lds_bytes, q);
            case 3: return launch_one<3, 2, 3, 2>(a, MT, n_items, lds_bytes, q);
            case 4: return launch_one<3, 2, 4, 2>(a, MT, n_items, lds_bytes, q);
            default: break;
        }
    } else if (a.PG == 1) {  // wide input pixels (64 channels, stride 2): one position per stage fits the LDS
        if (CT == 1) switch (ntw) {
            case 1: return launch_one<3, 1, 1, 1>(a, MT, n_items, lds_bytes, q);
            case 2: return launch_one<3, 1, 2, 1>(a, MT, n_items, lds_bytes, q);
            default: break;
        }
        else switch (ntw) {
            case 2: return launch_one<3, 2, 2, 1>(a, MT, n_items, lds_bytes, q);
            case 3: return launch_one<3, 2, 3, 1>(a, MT, n_items, lds_bytes, q);
            case 4: return launch_one<3, 2, 4, 1>(a, MT, n_items, lds_bytes, q);
            default: break;
        }
    }
    IDQN_REQUIRE(false, "plane wgrad: no kernel for %d planes, %d x %d tiles", NPX, MT, CT);
}
