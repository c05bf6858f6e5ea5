// What do the cache-policy bits (sc0 / sc1 / nt) of gfx950 loads and stores do to the Infinity Cache (256 MiB, memory side)?
// Every line is one timed kernel over one buffer; a scenario is a sequence of them.  Questions:
//   (1) does a buffer read with the default policy stay resident for a re-read, and at what rate is it then served?
//   (2) does a stream of nt / sc1 reads or writes over ANOTHER buffer leave that resident buffer in place (LRU defeated)?
//   (3) do nt / sc1 stores leave dirty lines behind (a later cold read then pays for their write-back: 2.9 vs 6.4 TB/s)?
// hipcc --offload-arch=gfx950 -O3 mall_policy_probe.hip -o mall_policy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// aux bits of the raw buffer builtins on gfx950: 1 = sc0, 2 = nt, 16 = sc1
template <int AUX>
__global__ __launch_bounds__(256) void k_rd(const void* p, unsigned bytes, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
    const unsigned stride = gridDim.x * 256u * 16u;
    unsigned off = (blockIdx.x * 256u + threadIdx.x) * 16u;
    unsigned acc = 0;
    for (; off + 3u * stride < bytes; off += 4u * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, off + u * stride, 0, AUX);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345u) out[0] = 1.f;
}
template <int AUX>
__global__ __launch_bounds__(256) void k_wr(void* p, unsigned bytes, unsigned val) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
    const unsigned stride = gridDim.x * 256u * 16u;
    unsigned off = (blockIdx.x * 256u + threadIdx.x) * 16u;
    const u32x4 v = {val, val + 1, val + 2, val + 3};
    for (; off < bytes; off += stride) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
}
// read-modify-write of every line (the Adam pattern): RA = policy of the read, WA = policy of the write
template <int RA, int WA>
__global__ __launch_bounds__(256) void k_rmw(void* p, unsigned bytes) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, bytes, 0x00020000);
    const unsigned stride = gridDim.x * 256u * 16u;
    unsigned off = (blockIdx.x * 256u + threadIdx.x) * 16u;
    for (; off + 3u * stride < bytes; off += 4u * stride) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, off + u * stride, 0, RA);
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u].x += 1; __builtin_amdgcn_raw_buffer_store_b128(v[u], r, off + u * stride, 0, WA); }
    }
}

struct Buf { void* p; unsigned bytes; const char* name; };
hipEvent_t e0, e1;
float* outp;
int WGS = 512;

template <typename F>
float timed(F launch) {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1e3f;
}
const char* aux_name(int a) {
    switch (a) { case 0: return "default"; case 2: return "nt"; case 16: return "sc1"; case 17: return "sc0sc1"; case 18: return "sc1nt"; case 19: return "sc0sc1nt"; case 1: return "sc0"; case 3: return "sc0nt"; }
    return "?";
}
float rd(const Buf& b, int aux) {
    switch (aux) {
        case 0: return timed([&] { hipLaunchKernelGGL(k_rd<0>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, outp); });
        case 2: return timed([&] { hipLaunchKernelGGL(k_rd<2>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, outp); });
        case 16: return timed([&] { hipLaunchKernelGGL(k_rd<16>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, outp); });
        case 17: return timed([&] { hipLaunchKernelGGL(k_rd<17>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, outp); });
        case 18: return timed([&] { hipLaunchKernelGGL(k_rd<18>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, outp); });
        case 19: return timed([&] { hipLaunchKernelGGL(k_rd<19>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, outp); });
    }
    return -1;
}
float wr(const Buf& b, int aux) {
    switch (aux) {
        case 0: return timed([&] { hipLaunchKernelGGL(k_wr<0>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, 7u); });
        case 2: return timed([&] { hipLaunchKernelGGL(k_wr<2>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, 7u); });
        case 16: return timed([&] { hipLaunchKernelGGL(k_wr<16>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, 7u); });
        case 17: return timed([&] { hipLaunchKernelGGL(k_wr<17>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, 7u); });
        case 18: return timed([&] { hipLaunchKernelGGL(k_wr<18>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, 7u); });
        case 19: return timed([&] { hipLaunchKernelGGL(k_wr<19>, dim3(WGS), dim3(256), 0, 0, b.p, b.bytes, 7u); });
    }
    return -1;
}
float rmw(const Buf& b, int ra, int wa) {
    if (ra == 0 && wa == 0) return timed([&] { hipLaunchKernelGGL((k_rmw<0, 0>), dim3(WGS), dim3(256), 0, 0, b.p, b.bytes); });
    if (ra == 2 && wa == 2) return timed([&] { hipLaunchKernelGGL((k_rmw<2, 2>), dim3(WGS), dim3(256), 0, 0, b.p, b.bytes); });
    if (ra == 0 && wa == 2) return timed([&] { hipLaunchKernelGGL((k_rmw<0, 2>), dim3(WGS), dim3(256), 0, 0, b.p, b.bytes); });
    if (ra == 2 && wa == 0) return timed([&] { hipLaunchKernelGGL((k_rmw<2, 0>), dim3(WGS), dim3(256), 0, 0, b.p, b.bytes); });
    if (ra == 16 && wa == 16) return timed([&] { hipLaunchKernelGGL((k_rmw<16, 16>), dim3(WGS), dim3(256), 0, 0, b.p, b.bytes); });
    return -1;
}
void rate(const char* what, const Buf& b, float us, double mult = 1.0) {
    printf("    %-46s %-6s %7.1f us  %6.2f TB/s\n", what, b.name, us, mult * b.bytes / us / 1e6);
}

int main(int argc, char** argv) {
    if (argc > 1) WGS = atoi(argv[1]);
    const unsigned MB = 1u << 20;
    Buf A = {nullptr, 160 * MB, "A160"}, B = {nullptr, 160 * MB, "B160"}, P = {nullptr, 640 * MB, "P640"}, S = {nullptr, 80 * MB, "S80"},
        Q = {nullptr, 320 * MB, "Q320"};
    for (Buf* b : {&A, &B, &P, &S, &Q}) { CK(hipMalloc(&b->p, b->bytes)); CK(hipMemset(b->p, 1, b->bytes)); }
    CK(hipMalloc(&outp, 4));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    printf("mall_policy_probe: %d workgroups of 256\n", WGS);
    const int pol[] = {0, 2, 16, 17, 19};
    for (int rep = 0; rep < 2; ++rep) {
        printf("== pass %d\n", rep);
        printf("-- (1) residency after a default read: re-read rates\n");
        for (const Buf* b : {&S, &A, &Q}) {
            rd(P, 0);                      // clean, cold
            rate("cold read (after 640 MB of default reads)", *b, rd(*b, 0));
            rate("re-read default", *b, rd(*b, 0));
            rate("re-read default again", *b, rd(*b, 0));
            rate("re-read nt", *b, rd(*b, 2));
            rate("re-read default after the nt read", *b, rd(*b, 0));
        }
        printf("-- (2) A resident, then a 640 MB stream over P with policy X, then re-read A (default)\n");
        for (int x : pol) {
            char nm[96];
            rd(P, 0); rd(A, 0); rd(A, 0);
            snprintf(nm, 96, "  stream: read P %s", aux_name(x));
            rate(nm, P, rd(P, x));
            snprintf(nm, 96, "A after reads of P (%s)", aux_name(x));
            rate(nm, A, rd(A, 0));
        }
        for (int x : pol) {
            char nm[96];
            rd(P, 0); rd(A, 0); rd(A, 0);
            snprintf(nm, 96, "  stream: write P %s", aux_name(x));
            rate(nm, P, wr(P, x));
            snprintf(nm, 96, "A after writes of P (%s)", aux_name(x));
            rate(nm, A, rd(A, 0));
            rd(P, 0);  // flush whatever is dirty before the next scenario
        }
        printf("-- (3) cold read of B right after 640 MB of writes with policy X (dirty lines to evict?)\n");
        for (int x : pol) {
            char nm[96];
            rd(P, 0);
            wr(P, x);
            snprintf(nm, 96, "B default, after writes of P (%s)", aux_name(x));
            rate(nm, B, rd(B, 0));
            rd(P, 0); wr(P, x);
            snprintf(nm, 96, "B nt, after writes of P (%s)", aux_name(x));
            rate(nm, B, rd(B, 2));
        }
        printf("-- (4) read-modify-write of Q (320 MB) by policy; then a cold default read of B behind it\n");
        const int rw[][2] = {{0, 0}, {2, 2}, {0, 2}, {2, 0}, {16, 16}};
        for (auto& c : rw) {
            char nm[96];
            rd(P, 0);
            snprintf(nm, 96, "rmw read %s / write %s (cold)", aux_name(c[0]), aux_name(c[1]));
            rate(nm, Q, rmw(Q, c[0], c[1]), 2.0);
            snprintf(nm, 96, "  B default behind it");
            rate(nm, B, rd(B, 0));
        }
        printf("-- (5) rmw of a RESIDENT buffer (S read first, default), by policy\n");
        for (auto& c : rw) {
            char nm[96];
            rd(P, 0); rd(S, 0); rd(S, 0);
            snprintf(nm, 96, "rmw read %s / write %s (resident)", aux_name(c[0]), aux_name(c[1]));
            rate(nm, S, rmw(S, c[0], c[1]), 2.0);
            rate("  S re-read default after it", S, rd(S, 0));
        }
        printf("-- (6) the step's pattern: S80 resident (theta), Q320 streamed rmw (m, v) with policy X, then S80 re-read\n");
        for (auto& c : rw) {
            char nm[96];
            rd(P, 0); rd(S, 0); rd(S, 0);
            snprintf(nm, 96, "  rmw Q read %s / write %s", aux_name(c[0]), aux_name(c[1]));
            rate(nm, Q, rmw(Q, c[0], c[1]), 2.0);
            rate("S after it", S, rd(S, 0));
        }
    }
    return 0;
}
