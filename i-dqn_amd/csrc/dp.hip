// Data-parallel i-DQN step with the collectives INSIDE the library (RCCL over xGMI), one process per GPU.
//
// The reference is single-device (slimdqn/networks/idqn.py:96-109); its loss is a plain mean over the minibatch
// (idqn.py:111-112), so the gradient of a global batch of B * world samples is the SUM of the shard gradients when every
// shard divides by the global batch size.  What crosses xGMI per step:
//   * ONE all-gather of each rank's Dense_0 gradient FACTORS [dL/dh | a3] (K * (J + F) * 32 floats per 32-sample block, 5.3 MB
//     at K = 5) instead of an all-reduce of the 79 MB product -- the library keeps the two factors as one contiguous run
//     (idqn_dense0_factors), so nothing is packed or copied;
//   * ONE all-reduce (sum) of the small-leaf region of the gradient arena (every leaf but Dense_0/kernel, plus the K losses).
// Every rank then runs the identical fused update over the gathered global batch: replicas stay bit-identical.
// idqn_dp_step enqueues the whole schedule from C -- forward, head, Dense_0 data gradient | all-gather | conv backward |
// all-reduce | fused Dense_0 update | Adam on the other leaves -- with the collectives on the caller's stream or on a side
// stream of the library's own (hipEvents, no host synchronisation).  slimdqn/networks/parallel.py keeps the same schedule in
// Python over torch.distributed: it is the oracle of the tests (gloo on CPU, two ranks on one card) for this function.
//
// RCCL is resolved at run time (dlopen of librccl.so.1: in a process that imported torch this is the RCCL torch itself
// uses, otherwise the one next to the HIP runtime), so that the single-GPU library has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"
#include "dp_internal.h"

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;
char g_rccl_err[256] = "";

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", nullptr};
    for (int i = 0; names[i] && !g_rccl.lib; ++i) g_rccl.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.lib) {
        snprintf(g_rccl_err, sizeof g_rccl_err, "dlopen(librccl.so.1) failed: %s", dlerror());
        return;
    }
#define RCCL_SYM(field, name)                                                                  \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name));          \
    if (!g_rccl.field && !g_rccl_err[0]) snprintf(g_rccl_err, sizeof g_rccl_err, "librccl has no symbol %s", name)
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(CommCount, "ncclCommCount");
    RCCL_SYM(CommUserRank, "ncclCommUserRank");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
}

int need_rccl() {
    std::call_once(g_rccl_once, load_rccl);
    IDQN_REQUIRE(g_rccl.lib && !g_rccl_err[0], "RCCL is not available: %s", g_rccl_err);
    return IDQN_OK;
}

#define IDQN_NCCL_CHECK(expr)                                                                                          \
    do {                                                                                                               \
        ncclResult_t _r = (expr);                                                                                      \
        if (_r != ncclSuccess) {                                                                                       \
            idqn_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?", __FILE__, __LINE__); \
            return IDQN_E_HIP;                                                                                         \
        }                                                                                                              \
    } while (0)

}  // namespace

struct idqn_dp_s {
    idqn_handle_t h = nullptr;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int rank = 0, world = 1;
    uint32_t flags = 0;
    hipStream_t side = nullptr;  // the collectives' stream (IDQN_DP_SIDE_STREAM)
    hipEvent_t ev_fwd = nullptr, ev_gather = nullptr, ev_bwd = nullptr, ev_small = nullptr;
    float* gathered = nullptr;  // [world][dL/dh | a3] of the last step
    long gathered_cap = 0;      // floats
};

extern "C" int idqn_dp_unique_id(void* id_out) {
    IDQN_REQUIRE(id_out, "idqn_dp_unique_id: null pointer");
    int rc = need_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == IDQN_DP_UNIQUE_ID_BYTES, "ncclUniqueId size");
    IDQN_NCCL_CHECK(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
    return IDQN_OK;
}

static int dp_finish_create(idqn_dp_s* dp, idqn_dp_t* out) {
    if (dp->flags & IDQN_DP_SIDE_STREAM) {
        IDQN_HIP_CHECK(hipStreamCreateWithFlags(&dp->side, hipStreamNonBlocking));
        for (hipEvent_t* e : {&dp->ev_fwd, &dp->ev_gather, &dp->ev_bwd, &dp->ev_small})
            IDQN_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    *out = dp;
    return IDQN_OK;
}

extern "C" int idqn_dp_create(idqn_handle_t h, const void* unique_id, int32_t rank, int32_t world, uint32_t flags, idqn_dp_t* out) {
    IDQN_REQUIRE(h && unique_id && out, "idqn_dp_create: null pointer");
    IDQN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "idqn_dp_create: rank %d of %d", rank, world);
    IDQN_REQUIRE(!(flags & ~IDQN_DP_SIDE_STREAM), "idqn_dp_create: unknown flags %u", flags);
    int rc = need_rccl();
    if (rc) return rc;
    IdqnDpView v;
    if ((rc = idqn_internal_dp_view(h, &v))) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    IDQN_NCCL_CHECK(g_rccl.CommInitRank(&comm, world, id, rank));  // (collective: every rank of the job calls it)
    idqn_dp_s* dp = new idqn_dp_s();
    dp->h = h; dp->comm = comm; dp->own_comm = true; dp->rank = rank; dp->world = world; dp->flags = flags;
    rc = dp_finish_create(dp, out);
    if (rc) idqn_dp_destroy(dp);
    return rc;
}

extern "C" int idqn_dp_create_from_comm(idqn_handle_t h, void* nccl_comm, uint32_t flags, idqn_dp_t* out) {
    IDQN_REQUIRE(h && nccl_comm && out, "idqn_dp_create_from_comm: null pointer");
    IDQN_REQUIRE(!(flags & ~IDQN_DP_SIDE_STREAM), "idqn_dp_create_from_comm: unknown flags %u", flags);
    int rc = need_rccl();
    if (rc) return rc;
    IdqnDpView v;
    if ((rc = idqn_internal_dp_view(h, &v))) return rc;
    idqn_dp_s* dp = new idqn_dp_s();
    dp->h = h; dp->comm = (ncclComm_t)nccl_comm; dp->own_comm = false; dp->flags = flags;
    ncclResult_t r1 = g_rccl.CommCount(dp->comm, &dp->world), r2 = g_rccl.CommUserRank(dp->comm, &dp->rank);
    if (r1 != ncclSuccess || r2 != ncclSuccess) {
        delete dp;
        IDQN_REQUIRE(false, "idqn_dp_create_from_comm: the communicator does not answer ncclCommCount / ncclCommUserRank");
    }
    rc = dp_finish_create(dp, out);
    if (rc) idqn_dp_destroy(dp);
    return rc;
}

extern "C" int idqn_dp_destroy(idqn_dp_t dp) {
    if (!dp) return IDQN_OK;
    (void)hipDeviceSynchronize();
    if (dp->own_comm && dp->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(dp->comm);
    for (hipEvent_t e : {dp->ev_fwd, dp->ev_gather, dp->ev_bwd, dp->ev_small})
        if (e) (void)hipEventDestroy(e);
    if (dp->side) (void)hipStreamDestroy(dp->side);
    if (dp->gathered) (void)hipFree(dp->gathered);
    delete dp;
    return IDQN_OK;
}

extern "C" int idqn_dp_info(idqn_dp_t dp, int32_t* rank, int32_t* world, int64_t* gather_bytes_per_rank, int64_t* allreduce_bytes) {
    IDQN_REQUIRE(dp, "idqn_dp_info: null handle");
    IdqnDpView v;
    int rc = idqn_internal_dp_view(dp->h, &v);
    if (rc) return rc;
    if (rank) *rank = dp->rank;
    if (world) *world = dp->world;
    if (gather_bytes_per_rank) *gather_bytes_per_rank = (int64_t)v.K * (v.F + v.J) * 32 * 4;  // per 32-sample block of a rank's shard
    if (allreduce_bytes) *allreduce_bytes = (int64_t)v.n_small * 4;
    return IDQN_OK;
}

extern "C" int idqn_dp_step(idqn_dp_t dp, const void* state_dev, const void* next_state_dev, const int32_t* action_dev,
                            const float* reward_dev, const uint8_t* terminal_dev, int32_t batch, int32_t global_batch,
                            uint32_t flags, void* stream) {
    IDQN_REQUIRE(dp, "idqn_dp_step: null handle");
    IDQN_REQUIRE(!(flags & ~(IDQN_F_PROFILE | IDQN_F_PROFILE_ALL)), "idqn_dp_step: only the profile flags are supported");
    IDQN_REQUIRE(global_batch == batch * dp->world, "idqn_dp_step: global batch %d is not %d ranks x %d samples (equal shards)",
                 global_batch, dp->world, batch);
    idqn_handle_t h = dp->h;
    hipStream_t q = (hipStream_t)stream;
    const bool side = dp->side != nullptr;
    hipStream_t qc = side ? dp->side : q;  // the collectives' stream
    int rc;
    // forward of the 2K nets, head, TD / loss, Dense_0 data gradient (the conv backward needs it first)
    if ((rc = idqn_learn_on_batch(h, state_dev, next_state_dev, action_dev, reward_dev, terminal_dev, batch, global_batch,
                                  IDQN_F_STOP_BEFORE_DENSE0_WGRAD | flags, stream)))
        return rc;
    float* factors = nullptr;
    int64_t n_dh = 0, n_a3 = 0;
    if ((rc = idqn_dense0_factors(h, &factors, &n_dh, &n_a3))) return rc;
    const long n = (long)(n_dh + n_a3);
    if (dp->gathered_cap < n * dp->world) {  // (first step of a job: outside any timed region)
        if (dp->gathered) { IDQN_HIP_CHECK(hipDeviceSynchronize()); IDQN_HIP_CHECK(hipFree(dp->gathered)); dp->gathered = nullptr; }
        IDQN_HIP_CHECK(hipMalloc((void**)&dp->gathered, (size_t)n * dp->world * 4));
        dp->gathered_cap = n * dp->world;
    }
    IdqnDpView v;
    if ((rc = idqn_internal_dp_view(h, &v))) return rc;
    // all-gather of [dL/dh | a3]: ONE collective for both factors, under the conv backward when it has a stream of its own
    if (side) {
        IDQN_HIP_CHECK(hipEventRecord(dp->ev_fwd, q));
        IDQN_HIP_CHECK(hipStreamWaitEvent(qc, dp->ev_fwd, 0));
    }
    IDQN_NCCL_CHECK(g_rccl.AllGather(factors, dp->gathered, (size_t)n, ncclFloat, dp->comm, qc));
    if (side) IDQN_HIP_CHECK(hipEventRecord(dp->ev_gather, qc));
    if ((rc = idqn_backward_rest(h, stream))) return rc;  // conv backward: small-leaf gradients complete in grad_dev
    // all-reduce of the small-leaf region (+ the K losses when the caller keeps them in its reserved floats)
    if (side) {
        IDQN_HIP_CHECK(hipEventRecord(dp->ev_bwd, q));
        IDQN_HIP_CHECK(hipStreamWaitEvent(qc, dp->ev_bwd, 0));
    }
    IDQN_NCCL_CHECK(g_rccl.AllReduce(v.grad, v.grad, (size_t)v.n_small, ncclFloat, ncclSum, dp->comm, qc));
    const bool losses_inside = v.losses >= v.grad && v.losses + v.K <= v.grad + v.n_small;
    if (!losses_inside) IDQN_NCCL_CHECK(g_rccl.AllReduce(v.losses, v.losses, (size_t)v.K, ncclFloat, ncclSum, dp->comm, qc));
    if (side) {
        IDQN_HIP_CHECK(hipEventRecord(dp->ev_small, qc));
        IDQN_HIP_CHECK(hipStreamWaitEvent(q, dp->ev_gather, 0));
    }
    // fused Dense_0 update over the gathered global batch (needs only the gather), then Adam on every other leaf
    const int nb = (batch + 31) / 32;
    const long X = (long)v.F * 32, Y = (long)v.J * 32;
    const float* dh_all = dp->gathered;
    const float* a3_all = dp->gathered + n_dh;
    if ((rc = idqn_finish_step_factored(h, a3_all, dh_all, dp->world * nb, nb, n, nb * X, X, n, nb * Y, Y, IDQN_FACTORED_DENSE0, stream)))
        return rc;
    if (side) IDQN_HIP_CHECK(hipStreamWaitEvent(q, dp->ev_small, 0));
    return idqn_finish_step_factored(h, a3_all, dh_all, dp->world * nb, nb, n, nb * X, X, n, nb * Y, Y, IDQN_FACTORED_REST, stream);
}
