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

#ifdef IDQN_VARIANTS  // stream role (IDQN_OVERLAP=1) and Adam role (IDQN_ADAM_ROLE=1): measured neutral / slower, DESIGN.md section 3
// The same launch with a STREAM ROLE behind the conv workgroups (dense0_update.h): blocks [0, n_conv) run the weight gradient,
// blocks [n_conv, gridDim.x) a share of the fused Dense_0 update on the CUs this launch leaves free.
template <int NPX, int CT, int NTW, int PG>
__global__ __launch_bounds__(512) void k_cwgrad_s(CWgradArgs a, unsigned stage_bytes, int MT, int n_conv, D0Stream ds) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_s[];
    warm_kernargs<(sizeof(CWgradArgs) + sizeof(D0Stream) + 32 < 1024 ? sizeof(CWgradArgs) + sizeof(D0Stream) + 32 : 1024)>();
    if ((int)blockIdx.x >= n_conv) {
        d0_stream_role(ds, (int)blockIdx.x - n_conv, reinterpret_cast<float*>(lds_s));
        return;
    }
    cwgrad_body<NPX, CT, NTW, PG>(a, stage_bytes, MT, xcd_contiguous_id_n(n_conv));
}

// The same launch with an ADAM ROLE behind the conv workgroups: blocks [n_conv, gridDim.x) run the small-leaf Adam update
// (dense0_update.h, adam_thread) over the leaves whose gradients do not depend on this launch -- every leaf but Conv_0's --
// on the CUs the weight gradient leaves free.  Independent roles, no hand-off: the launch that follows only has Conv_0's
// 8 k parameters per head left.
template <int NPX, int CT, int NTW, int PG>
__global__ __launch_bounds__(512) void k_cwgrad_a(CWgradArgs a, unsigned stage_bytes, int MT, int n_conv, AdamArgs ad, long n_threads) {
    warm_kernargs<(sizeof(CWgradArgs) + sizeof(AdamArgs) + 48 < 1024 ? sizeof(CWgradArgs) + sizeof(AdamArgs) + 48 : 1024)>();
    if ((int)blockIdx.x >= n_conv) {
        const long step = (long)((int)gridDim.x - n_conv) * 512;
        for (int k = 0; k < ad.K; ++k)
            for (long gid = (long)((int)blockIdx.x - n_conv) * 512 + threadIdx.x; gid < n_threads; gid += step) adam_thread(ad, k, gid);
        return;
    }
    cwgrad_body<NPX, CT, NTW, PG>(a, stage_bytes, MT, xcd_contiguous_id_n(n_conv));
}

template <int NPX, int CT, int NTW, int PG>
int launch_one_a(const CWgradArgs& a, int MT, int n_items, size_t lds_bytes, hipStream_t q, const AdamArgs& ad, long n_threads, int n_role) {
    static LdsAttrMark attr;  // per instantiation
    if (attr.needs(lds_bytes + 2048)) {
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_cwgrad_a<NPX, CT, NTW, PG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes + 2048));
    }
    hipLaunchKernelGGL((k_cwgrad_a<NPX, CT, NTW, PG>), dim3((unsigned)(n_items + n_role)), dim3(512), lds_bytes + 2048, q, a,
                       (unsigned)(lds_bytes / 2), MT, n_items, ad, n_threads);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

template <int NPX, int CT, int NTW, int PG>
int launch_one_s(const CWgradArgs& a, int MT, int n_items, size_t lds_bytes, hipStream_t q, const D0Stream& ds) {
    const size_t lds = std::max(lds_bytes + 2048, (size_t)65536);
    static LdsAttrMark attr;  // per instantiation
    if (attr.needs(lds)) {
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_cwgrad_s<NPX, CT, NTW, PG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL((k_cwgrad_s<NPX, CT, NTW, PG>), dim3((unsigned)(n_items + ds.n_sb)), dim3(512), lds, q, a,
                       (unsigned)(lds_bytes / 2), MT, n_items, ds);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

#endif  // IDQN_VARIANTS

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
#ifdef IDQN_VARIANTS
bool convp_wgrad_stream_built(int NPX, int MT, int CT, int PG) {  // which weight-gradient kernels carry the stream role
    const int ntw = (MT * CT + 3) / 4;
    return NPX == 1 && ((CT == 1 && ntw == 2 && PG == 4) || (CT == 2 && ntw == 4 && PG == 2));
}

bool convp_wgrad_adam_built(int NPX, int MT, int CT, int PG) { return NPX == 1 && CT == 1 && (MT * CT + 3) / 4 == 2 && PG == 4; }
#else
bool convp_wgrad_stream_built(int, int, int, int) { return false; }
bool convp_wgrad_adam_built(int, int, int, int) { return false; }
#endif

int convp_launch_wgrad_adam(const CWgradArgs& a, int NPX, int MT, int CT, int n_items, size_t lds_bytes, hipStream_t q, const AdamArgs& ad,
                            long n_threads, int n_role) {
    IDQN_REQUIRE(lds_bytes + 2048 <= 160 * 1024, "plane wgrad: %zu bytes of LDS per workgroup", lds_bytes + 2048);
    IDQN_REQUIRE(n_role >= 1 && n_items + n_role <= 256, "plane wgrad: %d + %d workgroups do not fit one per CU", n_items, n_role);
    IDQN_REQUIRE(convp_wgrad_adam_built(NPX, MT, CT, a.PG), "plane wgrad: no Adam-role kernel for this shape");
#ifdef IDQN_VARIANTS
    return launch_one_a<1, 1, 2, 4>(a, MT, n_items, lds_bytes, q, ad, n_threads, n_role);
#else
    (void)q; (void)ad; (void)n_threads;
    return IDQN_E_INVALID;
#endif
}

int convp_launch_wgrad(const CWgradArgs& a, int NPX, int MT, int CT, int n_items, size_t lds_bytes, hipStream_t q, const D0Stream* ds) {
    IDQN_REQUIRE(lds_bytes + 2048 <= 160 * 1024, "plane wgrad: %zu bytes of LDS per workgroup", lds_bytes + 2048);
    const int ntw = (MT * CT + 3) / 4;
#ifdef IDQN_VARIANTS
    if (ds && ds->n_sb > 0 && ds->rounds > 0) {
        IDQN_REQUIRE(n_items + ds->n_sb <= 256, "plane wgrad: %d + %d workgroups do not fit one per CU", n_items, ds->n_sb);
        if (NPX == 1 && CT == 1 && ntw == 2 && a.PG == 4) return launch_one_s<1, 1, 2, 4>(a, MT, n_items, lds_bytes, q, *ds);
        if (NPX == 1 && CT == 2 && ntw == 4 && a.PG == 2) return launch_one_s<1, 2, 4, 2>(a, MT, n_items, lds_bytes, q, *ds);
        IDQN_REQUIRE(false, "plane wgrad: no stream-role kernel for this shape (convp_wgrad_stream_built says which exist)");
    }
#else
    IDQN_REQUIRE(!ds, "plane wgrad: stream roles exist in the IDQN_VARIANTS build only");
#endif
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
