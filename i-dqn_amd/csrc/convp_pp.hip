// Plane-layout convolution, forward and data gradient, PERSISTENT form for launches with several items per CU
// (B >= 64 sample blocks or many heads: BASELINE configs 4 / 5).  Same arithmetic, operands, layouts and work items as
// k_cfwd (convp_fwd_body.h; architectures/dqn.py:42-52, idqn.py:105) -- what changes is who does what, when.
//
// Why (profiles/r6_cprof_b256.txt, B = 256): a one-item workgroup spends 1.5 k cycles deriving its tables, 5.5-7.5 k waiting
// for its first two stages (the latency of a cold fill, not its bytes), 14-28 k in the superstep loop and 3-8 k in the
// epilogue; with 160 KB of LDS per workgroup nothing of the next workgroup can start before the last store of this one,
// so 30-45 % of a CU's time the matrix pipe has no work.  PMC (profiles/r6_pmc_b256.json): MFMA busy 0.25-0.43, waves
// parked 0.43-0.50 of their cycles.
//
// Here one workgroup per CU walks its items, and its THREE groups of four waves (one of each per SIMD) rotate through
// three roles from item to item:
//     item n    group n % 3          computes: the superstep loop of k_cfwd, accumulators stay in ITS registers
//               group (n + 1) % 3    loads: the LDS-DMA ring, `ring - 1` supersteps ahead, running on INTO item n + 1's
//                                    first supersteps -- the item it computes next
//               group (n + 2) % 3    does the epilogue of item n - 1, which it computed, in slices of half a tile per
//                                    superstep; it loads item n + 1 next
// so the first fill, the table look-up and the epilogue of every item but the first / last run beside another item's
// MFMAs.  No accumulator ever crosses LDS, the turn-around tiles are per wave, and the only state the groups share is the
// ring (one s_barrier per superstep, as before).  (Two groups -- the loader also doing the epilogue -- measured loader-bound:
// a wave blocked in LDS-DMA issue at the ~28 B/clk a CU's fill sustains cannot also run 7 k cycles of epilogue per item,
// profiles/r6_pp_phases.txt.)  Items come from a host-made table (CItem, 32 B each, read one item ahead with scalar loads).
// vmcnt counts a wave's LDS-DMA copies, DMA'd masks / biases and epilogue stores in issue order: every wave keeps a
// running count of what it issued (vm_seq) and waits for "everything up to x" with s_waitcnt vmcnt(vm_seq - x).
#include <algorithm>
#include <cstdlib>

#include "convp_fwd_body.h"

namespace {

__device__ __forceinline__ void wait_vmcnt63(int n) {  // at most n (wave-uniform, any value) vector-memory operations outstanding
#define CP_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        CP_W(0) CP_W(1) CP_W(2) CP_W(3) CP_W(4) CP_W(5) CP_W(6) CP_W(7) CP_W(8) CP_W(9) CP_W(10) CP_W(11) CP_W(12) CP_W(13) CP_W(14) CP_W(15)
        CP_W(16) CP_W(17) CP_W(18) CP_W(19) CP_W(20) CP_W(21) CP_W(22) CP_W(23) CP_W(24) CP_W(25) CP_W(26) CP_W(27) CP_W(28) CP_W(29)
        CP_W(30) CP_W(31) CP_W(32) CP_W(33) CP_W(34) CP_W(35) CP_W(36) CP_W(37) CP_W(38) CP_W(39) CP_W(40) CP_W(41) CP_W(42) CP_W(43)
        CP_W(44) CP_W(45) CP_W(46) CP_W(47) CP_W(48) CP_W(49) CP_W(50) CP_W(51) CP_W(52) CP_W(53) CP_W(54) CP_W(55) CP_W(56) CP_W(57)
        CP_W(58) CP_W(59) CP_W(60) CP_W(61) CP_W(62)
        default: if (n < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef CP_W
}

struct PPStrips {  // what a loader needs of an item (wave-uniform): its pixel strip and packed kernels
    int nx;
    unsigned long sb, wb0;
};
template <int NT>
struct PPTiles {  // what a compute wave needs of an item: where its tiles' fragments sit in a stage, and where they go
    unsigned abase[NT];
    int tpos[NT];
    int net, var, out_slot;
};

template <int NPA, int CT, int NQ, int NT>
__device__ __forceinline__ void cfwd_pp_body(const CFwdArgs& a, const unsigned stage_bytes, const int ring, const unsigned epi_off,
                                             const int n_items, const CItem* __restrict__ items, long long* prof) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int WB = NQ * CT * 3 * 1024, BLKA = NPA * 1024, NWP = NQ * CT * 3;
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, cl = lane & 31;
    const int wave12 = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave12 >> 2, wave = wave12 & 3;  // waves w, w + 4, w + 8 sit on one SIMD: one of them computes at any time
    const int ct = wave % CT, co = ct * 32 + cl;
    const unsigned lane16 = lane * 16;
    const unsigned lo_tr = (8 * h + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&lds[0];
    const int NSS = a.KH * a.NCC, ahead = ring - 1;
    // LDS behind the ring, one slot per wave TRIPLE (w, w + 4, w + 8): only the one in its epilogue role uses it
    //   4 x [turn-around tile rsz (the bias row lands in its first KB before the first tile)][ReLU masks NT x 2 KB (data gradient)]
    const unsigned r_f32 = a.out3 ? 6144 : 0, rsz = r_f32 + (a.out_f32 ? 4096 : 0);
    const unsigned slot_bytes = rsz + (a.epilogue == 1 ? NT * 2048 : 0);
    const unsigned R_off = epi_off + wave * slot_bytes, mask_off = R_off + rsz;

    // ---- this workgroup's items: XCD x (= blockIdx % 8) walks a contiguous slice of the items, its workgroups take
    // consecutive items of the slice round by round -- the order in which the one-item launch dispatches them
    const int G = (int)gridDim.x, bx = (int)blockIdx.x & 7, bl = (int)blockIdx.x >> 3;
    const int q8 = n_items >> 3, r8 = n_items & 7;
    const int s_begin = bx * q8 + min(bx, r8), s_len = q8 + (bx < r8 ? 1 : 0);
    const int Lx = (G >> 3) + (bx < (G & 7) ? 1 : 0);
    const int n_my = bl < s_len ? (s_len - bl + Lx - 1) / Lx : 0;
    if (n_my == 0) return;  // (the whole workgroup)
    // phase stamps (debugging, tools/probes/conv_prof.py): cycles of wave 0 / wave 4 by what they were doing
    const long long pw0 = prof ? wall_clock64() : 0;
    long long p_cwait = 0, p_comp = 0, p_need = 0, p_bar = 0, p_issue = 0, p_epi = 0, p_dec = 0, p_tail = 0;

    // an item's tables from its row of the host-made table (items are row-aligned: ONE strip of pixels per stage)
    auto decode = [&](const CItem& it, PPStrips& S, PPTiles<NT>& T) {
        const CVar& v = a.var[it.var];
        const int in_slot = (a.in_split > 0 ? (it.net >= a.in_split ? 1 : 0) : it.net) * a.nb + it.bb;
        const unsigned long in_base = (unsigned long)a.in + (unsigned long)in_slot * a.in_slot;
        const int row = it.pad0, c0 = it.pad1;  // output row and first column of the item
        S.nx = (it.np - 1) * a.SX + NQ;
        S.sb = in_base + (unsigned long)(row * a.S + v.in_off_h) * (unsigned long)a.row_bytes +
               (unsigned long)(c0 * a.S + v.in_off_w) * (unsigned long)a.pix_bytes;
        S.wb0 = (unsigned long)a.wq + (unsigned long)it.net * a.wq_stride + (unsigned long)v.w_off;
        T.net = it.net; T.var = it.var; T.out_slot = it.net * a.nb + it.bb;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int j = (wave + 4 * i) / CT;
            T.tpos[i] = j < it.np ? it.p0 + j : -1;
            T.abase[i] = WB + (unsigned)(min(j, it.np - 1) * a.SX) * BLKA;
        }
    };

    // ---- vector-memory bookkeeping of this wave ---------------------------------------------------------------------
    int vm_seq = 0;            // operations issued so far
    int rec0 = -1, rec1 = -1;  // vm_seq behind this wave's copies for the superstep of the NEXT barrier / the one after (-1: none)
    auto need = [&](const int x) {
        if (x >= 0) wait_vmcnt63(vm_seq - x);
    };
    auto stage = [&](const PPStrips& S, const int ss, const unsigned buf) {  // this wave's quarter of superstep ss of an item
        const unsigned long wsrc = S.wb0 + (unsigned long)ss * WB;
        for (int i = wave; i < NWP; i += 4) {
            dma16(lane16, wsrc + (unsigned long)i * 1024, buf + i * 1024);
            ++vm_seq;
        }
        const int kh = ss / a.NCC, cc = ss - kh * a.NCC;
        const unsigned long src = S.sb + (unsigned long)kh * (unsigned long)a.row_bytes + (unsigned long)cc * 1024;
        const unsigned dst = buf + WB;
        for (int x = wave; x < S.nx; x += 4) {
#pragma unroll
            for (int pl = 0; pl < NPA; ++pl) {
                dma16(lane16, src + (unsigned long)x * (unsigned long)a.xstep + (unsigned long)pl * (unsigned long)a.plane_bytes,
                      dst + (x * NPA + pl) * 1024);
                ++vm_seq;
            }
        }
    };

    // ---- the epilogue of one tile in two slices: A = bias / mask, 3-way split, into the wave's turn-around tile;
    // B = whole 1 KiB runs out of it to HBM (convp_fwd_body.h has the why of the turn-around and its swizzles)
    const unsigned wsw = (cl >> 1) & 3;
    const unsigned rsw = (lane * 16) ^ ((((unsigned)lane >> 3) & 3) << 4);
    const unsigned fsw = (lane * 16) ^ ((((unsigned)lane >> 3) & 7) << 4);
    unsigned char* const Rt = lds + R_off;
    auto tile_yx = [&](const int p, const int var, int& yh, int& yw) {
        const CVar& v = a.var[var];
        const int oh = p / v.OW, ow = p - oh * v.OW;
        yh = oh * v.out_mul + v.out_add_h;
        yw = ow * v.out_mul + v.out_add_w;
    };
    float bias = 0.f;  // of the item whose epilogue is pending (read out of the turn-around tile before the first slice A)
    auto epiA = [&](const f32x16& av, const int i, const int p, const int var, const int out_slot) {
        float val[16];
        if (a.epilogue == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) val[r] = fmaxf(av[r] + bias, 0.f);
        } else {
            const unsigned char* M = lds + mask_off + i * 2048 + cl * 64 + 8 * h;
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x2 mk = *LDS_PTR(const u32x2, M + 16 * g);
                const float m0 = __uint_as_float(mk.x << 16), m1 = __uint_as_float(mk.x & 0xffff0000u);
                const float m2 = __uint_as_float(mk.y << 16), m3 = __uint_as_float(mk.y & 0xffff0000u);
                val[4 * g + 0] = m0 > 0.f ? av[4 * g + 0] : 0.f;
                val[4 * g + 1] = m1 > 0.f ? av[4 * g + 1] : 0.f;
                val[4 * g + 2] = m2 > 0.f ? av[4 * g + 2] : 0.f;
                val[4 * g + 3] = m3 > 0.f ? av[4 * g + 3] : 0.f;
            }
            if (a.pb) {  // sum over the 32 samples, fixed order: registers, then the two half-waves
                int yh, yw;
                tile_yx(p, var, yh, yw);
#pragma unroll
                for (int r = 0; r < 16; ++r) s += val[r];
                s += __shfl_xor(s, 32);
                if (h == 0) a.pb[((long)out_slot * (a.out_H * a.out_W) + (long)yh * a.out_W + yw) * a.CO + co] = s;
                ++vm_seq;
            }
        }
        if (a.out3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned q0a, q1a, q2a, q0b, q1b, q2b;
                split3_pk(val[4 * g + 0], val[4 * g + 1], q0a, q1a, q2a);
                split3_pk(val[4 * g + 2], val[4 * g + 3], q0b, q1b, q2b);
                unsigned char* wp = Rt + cl * 64 + ((g ^ wsw) * 16) + 8 * h;
                *LDS_PTR(u32x2, wp) = (u32x2){q0a, q0b};
                *LDS_PTR(u32x2, wp + 2048) = (u32x2){q1a, q1b};
                *LDS_PTR(u32x2, wp + 4096) = (u32x2){q2a, q2b};
            }
        }
        if (a.out_f32) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *LDS_PTR(f32x4, Rt + r_f32 + cl * 128 + (((2 * g + h) ^ (cl & 7)) * 16)) =
                    (f32x4){val[4 * g], val[4 * g + 1], val[4 * g + 2], val[4 * g + 3]};
        }
    };
    auto epiB = [&](const int p, const int var, const int out_slot) {
        int yh, yw;
        tile_yx(p, var, yh, yw);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave, LDS is in order: slice A's writes are visible
        if (a.out3) {
            unsigned char* O = (unsigned char*)a.out3 + (unsigned long)out_slot * a.out_slot +
                               ((unsigned long)((yh + a.out_lo_h) * a.out_Wp + (yw + a.out_lo_w)) * (3UL * a.CO) + ct * 32) * 64 + lane16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 x = *LDS_PTR(const u32x4, Rt + pl * 2048 + j * 1024 + rsw);
                    *reinterpret_cast<u32x4*>(O + (unsigned long)pl * a.CO * 64 + j * 1024) = x;
                }
            vm_seq += 6;
        }
        if (a.out_f32) {
            float* F = a.out_f32 + (long)out_slot * a.f32_slot + (((long)yh * a.f32_W + yw) * a.CO + ct * 32) * 32 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 x = *LDS_PTR(const f32x4, Rt + r_f32 + j * 1024 + fsw);
                *reinterpret_cast<f32x4*>(F + j * 256) = x;
            }
            vm_seq += 4;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tile has left LDS before the next one overwrites it
    };
    // bias row (forward) or ReLU masks (data gradient) of the item just computed, by LDS-DMA into this wave pair's slot
    auto epi_operands_landed = [&]() {  // behind the wait for them: the bias moves to a register, the tile is free for slice A
        if (a.epilogue == 0) {
            bias = *LDS_PTR(const float, lds + R_off + co * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    auto epi_operands = [&](const PPTiles<NT>& E) {
        if (a.epilogue == 0) {
            const float* pb = (E.net < a.n_first ? a.pbase[0] + (long)E.net * a.pstride : a.pbase[1] + (long)(E.net - a.n_first) * a.pstride) + a.b_off;
            dma16(min(lane16, (unsigned)a.CO * 4 - 16), (unsigned long)pb, lds0 + R_off);  // lanes past the row re-read its tail
            ++vm_seq;
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (E.tpos[i] < 0) continue;
                int yh, yw;
                tile_yx(E.tpos[i], E.var, yh, yw);
                const unsigned long msrc = (unsigned long)a.mask3 + (unsigned long)E.out_slot * a.mask_slot +
                    ((unsigned long)((yh + a.mask_lo_h) * a.mask_Wp + (yw + a.mask_lo_w)) * (3UL * a.mask_C) + ct * 32) * 64;
                dma16(lane16, msrc, lds0 + mask_off + i * 2048);
                dma16(lane16, msrc + 1024, lds0 + mask_off + i * 2048 + 1024);
                vm_seq += 2;
            }
        }
    };

    // ---- state -----------------------------------------------------------------------------------------------------
    PPStrips Sl, Sn;   // strips of the item this group loads now / of the item behind it (cross-item prefetch)
    PPTiles<NT> Tc;    // tiles of the item this group computes next
    PPTiles<NT> E;     // tiles of the item this group computed last: its epilogue is still to do while have_epi
    bool have_epi = false;
    f32x16 acc[NT];
    int cb = 0;        // ring slot of the superstep of the next barrier (the same in every wave)
    auto item_of = [&](const int n) { return s_begin + bl + n * Lx; };
    auto load_item = [&](const int n) {  // row n of this workgroup's part of the table, as wave-uniform values
        const CItem* p = items + item_of(n);
        CItem it;
        it.net = __builtin_amdgcn_readfirstlane(p->net); it.bb = __builtin_amdgcn_readfirstlane(p->bb);
        it.var = __builtin_amdgcn_readfirstlane(p->var); it.p0 = __builtin_amdgcn_readfirstlane(p->p0);
        it.np = __builtin_amdgcn_readfirstlane(p->np); it.pad0 = __builtin_amdgcn_readfirstlane(p->pad0);
        it.pad1 = __builtin_amdgcn_readfirstlane(p->pad1); it.pad2 = 0;
        return it;
    };

    if (grp <= 1) {  // group 0 computes item 0, group 1 loads it
        const CItem it0 = load_item(0);
        decode(it0, Sl, Tc);
    }
    if (grp == 1) {
        stage(Sl, 0, lds0);
        rec0 = vm_seq;
        if (ahead == 2 && NSS > 1) {
            stage(Sl, 1, lds0 + stage_bytes);
            rec1 = vm_seq;
        }
    }

    int role = grp == 0 ? 0 : grp == 1 ? 1 : 2;  // 0 compute, 1 load, 2 epilogue (of the item computed last, if any); then 2 -> 1 -> 0 -> 2
    for (int n = 0; n < n_my; ++n) {
        const bool has_next = n + 1 < n_my;
        if (role == 0) {
            // ======== compute item n (tables in Tc) ===================================================================
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const long long pc0 = prof ? clock64() : 0;
            for (int ss = 0; ss < NSS; ++ss) {
                const long long pb0 = prof ? clock64() : 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                need(rec0);  // (only just behind the loader role: the prefetched first supersteps of this item are this wave's copies)
                __builtin_amdgcn_s_barrier();
                if (prof) p_cwait += clock64() - pb0;
                rec0 = rec1; rec1 = -1;
                const unsigned char* cur = lds + cb * stage_bytes;
                cb = cb + 1 == ring ? 0 : cb + 1;
                // (fragment schedule of k_cfwd: activation halves of tile-step u + 2 and the weight planes of tap q + 1 are
                // requested in the gaps between the MFMAs of tile-step u.  Moving the barrier in front of the last two tile-steps
                // and requesting the next stage's first fragments under them -- no drain / refill of the matrix pipe at the
                // barrier -- was built and measured: same duration, 10-40 more registers, profiles/r6_pp_phases.txt)
                constexpr int U = NQ * NT, WSTEP = NT >= 2 ? NT - 2 : 0;
                bf16x8 wf[2][3];
                s16x4 ah[3][NPA][2];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[0][pl] = frag_lin(cur + (ct * 3 + pl) * 1024 + lane16);
#pragma unroll
                for (int u0 = 0; u0 < 2 && u0 < U; ++u0)
#pragma unroll
                    for (int pl = 0; pl < NPA; ++pl) {
                        const unsigned char* ap = cur + Tc.abase[u0 % NT] + (u0 / NT) * BLKA + pl * 1024 + lo_tr;
                        ah[u0][pl][0] = tr_half(ap);
                        ah[u0][pl][1] = tr_half(ap + 256);
                    }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const int u = q * NT + i;
                        const bool next_w = (i == WSTEP && q + 1 < NQ), next_a = (u + 2 < U);
                        const int qn = (u + 2) / NT, in_ = (u + 2) % NT;
                        const unsigned char* an = cur + Tc.abase[next_a ? in_ : 0] + qn * BLKA + lo_tr;
                        const unsigned char* wn = cur + (((q + 1) * CT + ct) * 3) * 1024 + lane16;
                        bf16x8 A[NPA];
#pragma unroll
                        for (int pl = 0; pl < NPA; ++pl) A[pl] = join8(ah[u % 3][pl][0], ah[u % 3][pl][1]);
                        const bf16x8* W = wf[q & 1];
                        __builtin_amdgcn_sched_barrier(0);
#define CF_GAP(m)                                                                                              \
    {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (next_a && (m) < 2 * NPA) ah[(u + 2) % 3][((m) / 2) % NPA][(m) % 2] = tr_half(an + ((m) / 2) * 1024 + ((m) % 2) * 256); \
        if (next_w && (m) < 3) wf[(q + 1) & 1][(m) < 3 ? (m) : 0] = frag_lin(wn + (m) * 1024);                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
                        if (NPA == 3) {  // smallest terms first
                            acc[i] = mfma_bf16(A[2], W[0], acc[i]);
                            CF_GAP(0)
                            acc[i] = mfma_bf16(A[0], W[2], acc[i]);
                            CF_GAP(1)
                            acc[i] = mfma_bf16(A[1], W[1], acc[i]);
                            CF_GAP(2)
                            acc[i] = mfma_bf16(A[1], W[0], acc[i]);
                            CF_GAP(3)
                            acc[i] = mfma_bf16(A[0], W[1], acc[i]);
                            CF_GAP(4)
                            acc[i] = mfma_bf16(A[0], W[0], acc[i]);
                            CF_GAP(5)
                        } else {
                            acc[i] = mfma_bf16(A[0], W[2], acc[i]);
                            CF_GAP(0)
                            acc[i] = mfma_bf16(A[0], W[1], acc[i]);
                            CF_GAP(1)
                            acc[i] = mfma_bf16(A[0], W[0], acc[i]);
                            CF_GAP(2)
                        }
#undef CF_GAP
                    }
                }
            }
            if (prof) p_comp += clock64() - pc0;
            E = Tc;
            have_epi = true;
            role = 2;
        } else if (role == 1) {
            // ======== load item n (strips in Sl), then the first supersteps of item n + 1, which this group computes ====
            CItem itn;
            if (has_next) itn = load_item(n + 1);
            for (int ss = 0; ss < NSS; ++ss) {
                const long long s0 = prof ? clock64() : 0;
                need(rec0);
                const long long s1 = prof ? clock64() : 0;
                __builtin_amdgcn_s_barrier();
                const long long s2 = prof ? clock64() : 0;
                rec0 = rec1; rec1 = -1;
                if (ss == 1 && has_next) decode(itn, Sn, Tc);
                const int tgt = ss + ahead;
                int tb = cb + ahead;
                tb = tb >= ring ? tb - ring : tb;
                const bool mine = tgt < NSS;
                if (mine || has_next) {
                    stage(mine ? Sl : Sn, mine ? tgt : tgt - NSS, lds0 + tb * stage_bytes);
                    if (ahead == 2) rec1 = vm_seq; else rec0 = vm_seq;
                }
                cb = cb + 1 == ring ? 0 : cb + 1;
                if (prof) { p_need += s1 - s0; p_bar += s2 - s1; p_issue += clock64() - s2; }
            }
            role = 0;
        } else {
            // ======== epilogue of item n - 1 (which this group computed) beside item n; this group loads item n + 1 ========
            CItem itn;
            if (has_next) itn = load_item(n + 1);
            int seq_aux = -1;
            int ss = 0;
            auto step = [&]() {
                const long long s1 = prof ? clock64() : 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (prof) p_bar += clock64() - s1;
                cb = cb + 1 == ring ? 0 : cb + 1;
            };
            step();  // superstep 0: behind it the wave that had the slot before is done with it
            ++ss;
            if (have_epi) {
                epi_operands(E);
                seq_aux = vm_seq;
            }
            if (has_next) {
                const long long d0 = prof ? clock64() : 0;
                decode(itn, Sl, Tc);  // (Tc unused: this group's next compute item is decoded in its loader role)
                if (prof) p_dec += clock64() - d0;
            }
            // tile e owns supersteps [1 + e (NSS - 1) / NT, 1 + (e + 1) (NSS - 1) / NT): slice A in its first, slice B in its
            // second (or the same, when it has one)
#pragma unroll
            for (int e = 0; e < NT; ++e) {
                const int end = 1 + ((e + 1) * (NSS - 1)) / NT;
                for (int k = 0; ss < end; ++ss, ++k) {
                    step();
                    const long long e0 = prof ? clock64() : 0;
                    if (have_epi && E.tpos[e] >= 0) {
                        const bool last = ss + 1 == end;
                        if (k == 0) {
                            if (seq_aux >= 0) { need(seq_aux); seq_aux = -1; epi_operands_landed(); }
                            epiA(acc[e], e, E.tpos[e], E.var, E.out_slot);
                        }
                        if (k == 1 || (k == 0 && last)) epiB(E.tpos[e], E.var, E.out_slot);
                    }
                    if (prof) p_epi += clock64() - e0;
                }
            }
            for (; ss < NSS; ++ss) step();
            have_epi = false;
            role = 1;
        }
    }
    // ---- the last item's epilogue: every other wave is done with the shared slots behind this barrier -------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long long pt0 = prof ? clock64() : 0;
    if (have_epi) {
        epi_operands(E);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        epi_operands_landed();
#pragma unroll
        for (int e = 0; e < NT; ++e) {
            if (E.tpos[e] < 0) continue;
            epiA(acc[e], e, E.tpos[e], E.var, E.out_slot);
            epiB(E.tpos[e], E.var, E.out_slot);
        }
    }
    if (prof && (t == 0 || t == 256)) {  // (the two planes of the stamp buffer: wave 0 and wave 4)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p_tail = clock64() - pt0;
        long long* pr = prof + (t == 0 ? 0L : 8L * 4096) + (long)blockIdx.x * 8;
        pr[0] = pw0; pr[1] = p_comp; pr[2] = p_cwait; pr[3] = p_need; pr[4] = p_bar; pr[5] = p_issue; pr[6] = wall_clock64(); pr[7] = n_my;
        pr += 8L * 1024;  // (at most 256 workgroups: rows 1024.. of the plane hold the second record)
        pr[0] = p_epi; pr[1] = p_dec; pr[2] = p_tail;
    }
}

template <int NPA, int CT, int NQ, int NT>
__global__ __launch_bounds__(768) void k_cfwd_pp(CFwdArgs a, unsigned stage_bytes, int ring, unsigned epi_off, int n_items, const CItem* items,
                                                  long long* prof) {
    warm_kernargs<sizeof(CFwdArgs)>();
    cfwd_pp_body<NPA, CT, NQ, NT>(a, stage_bytes, ring, epi_off, n_items, items, prof);
}

template <int NPA, int CT, int NQ, int NT>
int launch_pp(const CFwdArgs& a, int n_items, int n_wg, size_t stage_bytes, int ring, size_t lds_bytes, hipStream_t q, const CItem* items, long long* prof) {
    static LdsAttrMark attr;  // per instantiation
    if (attr.needs(lds_bytes)) {
        IDQN_HIP_CHECK(hipFuncSetAttribute((const void*)k_cfwd_pp<NPA, CT, NQ, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    hipLaunchKernelGGL((k_cfwd_pp<NPA, CT, NQ, NT>), dim3((unsigned)n_wg), dim3(768), lds_bytes, q, a, (unsigned)stage_bytes, ring,
                       (unsigned)(ring * stage_bytes), n_items, items, prof);
    IDQN_HIP_CHECK(hipGetLastError());
    return IDQN_OK;
}

template <int NPA, int CT, int NQ>
int launch_pp_nt(const CFwdArgs& a, int NT, int n_items, int n_wg, size_t stage_bytes, int ring, size_t lds_bytes, hipStream_t q, const CItem* items, long long* prof) {
    switch (NT) {
        case 2: return launch_pp<NPA, CT, NQ, 2>(a, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
        case 3: return launch_pp<NPA, CT, NQ, 3>(a, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
        case 4: return launch_pp<NPA, CT, NQ, 4>(a, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
        default: break;
    }
    if constexpr (NPA == 1) {
        if (NT == 5) return launch_pp<1, CT, NQ, 5>(a, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
        if (NT == 6) return launch_pp<1, CT, NQ, 6>(a, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
    }
    IDQN_REQUIRE(false, "persistent plane conv: %d tiles per wave with %d channel tiles is not built", NT, CT);
}

}  // namespace

// the combinations the Nature-CNN widths [32, 64, 64] need (roles 0..4 of qnet.hip's plans)
bool convp_pp_built(int NPA, int CT, int NQ, int NT) {
    const bool geom = (NPA == 1 && CT == 1 && NQ == 2) || (NPA == 3 && CT == 2 && NQ == 4) || (NPA == 3 && CT == 2 && NQ == 3) ||
                      (NPA == 3 && CT == 1 && NQ == 2);
    return geom && NT >= 2 && NT <= (NPA == 1 ? 6 : 4);  // (three waves per SIMD: 168 registers each)
}
// LDS behind the ring: one slot per wave triple (turn-around tile, the data gradient's masks)
size_t convp_pp_epi_bytes(int NT, int epilogue, bool planes_out, bool f32_out) {
    return 4 * ((planes_out ? 6144 : 0) + (f32_out ? 4096 : 0) + (epilogue == 1 ? (size_t)NT * 2048 : 0));
}

int convp_launch_fwd_pp(const CFwdArgs& a, int NPA, int CT, int NQ, int NT, int n_items, int n_wg, size_t stage_bytes, int ring,
                        size_t lds_bytes, hipStream_t q, const CItem* items, long long* prof) {
    IDQN_REQUIRE(items != nullptr && a.row_parts > 0, "persistent plane conv: no item table");
    IDQN_REQUIRE(lds_bytes <= 160 * 1024 && (ring == 2 || ring == 3), "persistent plane conv: %zu bytes of LDS, ring %d", lds_bytes, ring);
    IDQN_REQUIRE(a.KH * a.NCC - 1 >= NT && n_wg >= 1 && n_wg <= n_items, "persistent plane conv: %d supersteps for %d tiles per wave, %d workgroups for %d items",
                 a.KH * a.NCC, NT, n_wg, n_items);
    IDQN_REQUIRE(convp_pp_built(NPA, CT, NQ, NT), "persistent plane conv: <%d, %d, %d, %d> is not built", NPA, CT, NQ, NT);
    if (NPA == 1) return launch_pp_nt<1, 1, 2>(a, NT, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
    if (CT == 2 && NQ == 4) return launch_pp_nt<3, 2, 4>(a, NT, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
    if (CT == 2 && NQ == 3) return launch_pp_nt<3, 2, 3>(a, NT, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
    return launch_pp_nt<3, 1, 2>(a, NT, n_items, n_wg, stage_bytes, ring, lds_bytes, q, items, prof);
}
