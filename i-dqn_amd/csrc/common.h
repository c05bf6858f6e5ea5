// Shared helpers for the gfx950 kernels of libidqn_hip.so.  CDNA4 only: 64-wide wavefronts,
// v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fmaf chain) for every contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/idqn_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// D = A * B + C on one wave.  Operand maps (cdna_hip_programming.md section 3):
//   a: lane l holds A[i = l & 31][k = l >> 5]      b: lane l holds B[k = l >> 5][j = l & 31]
//   c/d: register r of lane l holds D[i = mfma_row(r, l >> 5)][j = l & 31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// LDS-DMA: 16 bytes per lane, global (per-lane address) -> LDS (wave-uniform base + 16 * lane).
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware work mapping (speed only, never correctness): consecutive blockIdx values are dealt round-robin
// over the 8 XCDs, each with its own 4 MB L2.  This returns a logical work index such that every XCD walks a
// CONTIGUOUS slice of [0, gridDim.x): neighbouring work items (the taps of one position chunk, adjacent output
// positions of one net, the f tiles of one head) then re-read each other's operands from the same L2 instead of
// every L2 fetching everything.  Bijective for any grid size (cdna_hip_programming.md, T1).
__device__ __forceinline__ int xcd_contiguous_id_n(const int n) {  // the same over the FIRST n blocks of a larger grid
    const int b = (int)blockIdx.x;
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
__device__ __forceinline__ int xcd_contiguous_id() {
    const int n = (int)gridDim.x, b = (int)blockIdx.x;
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// Workgroup barrier that orders LDS traffic only: wait for this wave's LDS operations, then s_barrier.
// Unlike __syncthreads() it does NOT drain vmcnt, so global loads prefetched for later rounds stay in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// zero-bordered activation / gradient buffer of one conv layer (rows of 32 samples)
struct ActGeom {
    int H, W, C;        // logical extent
    int lo_h, lo_w;     // zero border before the first row / column
    int Hp, Wp;         // padded extent
    long block;         // floats per (net, batch block) = Hp * Wp * C * 32
};

// Gradient arena addressing (include/idqn_hip.h): the cnn's Dense_0/kernel lives in a second region, every other leaf at
// its parameter offset (minus that leaf's size when it comes later); fc: w0_begin == w0_end, gP = head stride.
struct GradMap {
    long gP, w0_begin, w0_end, g_w0_base;
    __device__ __forceinline__ long at(int k, long e) const {
        const long w0n = w0_end - w0_begin;
        return e < w0_begin ? (long)k * gP + e : (e < w0_end ? g_w0_base + (long)k * w0n + (e - w0_begin) : (long)k * gP + e - w0n);
    }
};

void idqn_set_error(const char* fmt, ...);

#define IDQN_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            idqn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return IDQN_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define IDQN_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            idqn_set_error(__VA_ARGS__); \
            return IDQN_E_INVALID;       \
        }                                \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Run-time switches.  The shipped library (libidqn_hip.so) reads the few documented in INTEGRATION.md, each with a plain getenv().
// Debug knobs -- the phase stamps of the conv / MLP kernels (IDQN_CONV_PROF, IDQN_FC_PROF, IDQN_IQN_CLOCK) and the plan overrides
// of the measurement scripts (IDQN_PP_PARTS<role>, IDQN_PP_RING, IDQN_PAIR_*) -- resolve only in a -DIDQN_DEBUG_KNOBS build
// (__graft_entry__.build_debug() -> libidqn_hip_debug.so, tools/ only); in the shipped build they are constants and fold away.
#include <stdlib.h>
#ifdef IDQN_DEBUG_KNOBS
static inline const char* debug_env(const char* name) { return getenv(name); }
#else
static inline constexpr const char* debug_env(const char*) { return nullptr; }
#endif
static inline int debug_int(const char* name, int dflt) {
    const char* e = debug_env(name);
    return e ? atoi(e) : dflt;
}
static inline bool debug_on(const char* name) { return debug_int(name, 0) != 0; }
