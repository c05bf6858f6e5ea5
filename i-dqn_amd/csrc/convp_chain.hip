// The three forward convolutions of a net set (architectures/dqn.py:43-53, inside the single jitted step idqn.py:96-109)
// as ONE launch: every workgroup runs its Conv_0 item, then its Conv_1 item, then its Conv_2 item; an item of layer
// L + 1 starts once the producer items of layer L whose output rows it reads have raised their flags (convp.h,
// ChainHand) -- flags only, no data atomics, every sum stays in its fixed order, results are bit-identical to the three
// separate launches (k_cfwd: the default; this launch is opt-in, IDQN_CONV_CHAIN=1, see the measurement note below).
//
// What the chain removes against three launches: two kernel boundaries and two prologues, the chip-wide wait for the
// slowest workgroup of a layer, and the part of a layer's first LDS fill that does not depend on its producers (the packed
// kernels are requested before the flags are polled).  What it adds: write-through output stores, one flag store per item,
// one poll + one agent-scope acquire per consumer item.
//
// Measured (tools/probes/chain_prof.py, profiles/r4_conv_chain_timeline.txt): the chain takes 60.6 us against 59.1 us of
// kernel time for the three launches (step 0.2972 against 0.2979 ms): a consumer item waits for 3-9 producer items, and
// since every item of a layer takes the same time they all finish together -- the layers stay in lockstep, nothing of
// layer L + 1 can start before the end of layer L, and the write-through + drain + flag + acquire of a hand-off (+1.7 us of
// producer epilogue, 1.5 us of poll + acquire) costs what the kernel boundary and the prologue did.  Hence opt-in.
//
// Residency: grid <= number of CUs with one 512-thread workgroup per CU (the LDS request alone keeps a second one out), so
// every workgroup is resident from the start and an item only ever waits on items of an EARLIER layer of resident
// workgroups.  Spins are bounded (CHAIN_SPIN_LIMIT): a launch that cannot make progress (a CU-masked stream, a partitioned
// device) gives up, sets err[0] and the step's losses come out NaN instead of the GPU hanging.
#include <algorithm>

#include "convp_fwd_body.h"

#ifdef IDQN_VARIANTS

namespace {

template <int NT0, int NT1, int NT2>
__global__ __launch_bounds__(512) void k_cchain_fwd(CChainArgs c) {
    warm_kernargs<(sizeof(CChainArgs) < 2048 ? sizeof(CChainArgs) : 2048)>();
    const int v = xcd_contiguous_id();
    const unsigned epoch = __hip_atomic_load(c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // written by the staging launch
    if (v < c.n_items[0])
        cfwd_body<1, 1, 2, NT0, true>(c.a[0], c.stage_bytes[0], c.ring[0], c.mask_off[0], c.prof[0], v, c.n_items[0], &c.hand[0], epoch);
    if (v < c.n_items[1])
        cfwd_body<3, 2, 4, NT1, true>(c.a[1], c.stage_bytes[1], c.ring[1], c.mask_off[1], c.prof[1], v, c.n_items[1], &c.hand[1], epoch);
    if (v < c.n_items[2])
        cfwd_body<3, 2, 3, NT2, true>(c.a[2], c.stage_bytes[2], c.ring[2], c.mask_off[2], c.prof[2], v, c.n_items[2], &c.hand[2], epoch);
}

template <int NT0, int NT1, int NT2>
int launch_chain(const CChainArgs& c, int n_wg, size_t lds_bytes, hipStream_t q) {
    static LdsAttrMark attr;  // per instantiation
    if (attr.needs(lds_bytes)) {
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_cchain_fwd<NT0, NT1, NT2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    hipLaunchKernelGGL((k_cchain_fwd<NT0, NT1, NT2>), dim3((unsigned)n_wg), dim3(512), lds_bytes, q, c);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

}  // namespace

// Built: the Nature-CNN geometry (8x8/4 over 4 frames -> 32, 4x4/2 -> 64, 3x3/1 -> 64) at the tiles per wave the one-batch-
// block plans of K = 1 ... 8 heads give.  Anything else runs as three launches.
bool convp_chain_fwd_built(const int NT[3]) {
    return (NT[0] == 5 && NT[1] == 3 && NT[2] == 3) || (NT[0] == 4 && NT[1] == 3 && NT[2] == 3) || (NT[0] == 6 && NT[1] == 3 && NT[2] == 3);
}

int convp_launch_chain_fwd(const CChainArgs& c, const int NT[3], int n_wg, size_t lds_bytes, hipStream_t q) {
    IDQN_REQUIRE(lds_bytes <= 160 * 1024, "conv chain: %zu bytes of LDS per workgroup", lds_bytes);
    IDQN_REQUIRE(n_wg <= 256, "conv chain: %d workgroups do not fit one per CU", n_wg);
    if (NT[0] == 5 && NT[1] == 3 && NT[2] == 3) return launch_chain<5, 3, 3>(c, n_wg, lds_bytes, q);
    if (NT[0] == 4 && NT[1] == 3 && NT[2] == 3) return launch_chain<4, 3, 3>(c, n_wg, lds_bytes, q);
    if (NT[0] == 6 && NT[1] == 3 && NT[2] == 3) return launch_chain<6, 3, 3>(c, n_wg, lds_bytes, q);
    IDQN_REQUIRE(false, "conv chain: tiles per wave %d / %d / %d are not built (convp_chain_fwd_built says which are)", NT[0], NT[1], NT[2]);
}
#else  // default build: the chained launch is not compiled (measured neutral); three launches always
bool convp_chain_fwd_built(const int*) { return false; }
int convp_launch_chain_fwd(const CChainArgs&, const int*, int, size_t, hipStream_t) {
    IDQN_REQUIRE(false, "conv chain: built with -DIDQN_VARIANTS only");
}
#endif
