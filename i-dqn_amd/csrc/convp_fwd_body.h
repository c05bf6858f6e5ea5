// Plane-layout convolution, forward and data gradient: the workgroup body (convp.h has the layouts and the arithmetic).
// Kernels: convp_fwd.hip (one role per launch), convp_pair.hip (a data gradient beside a weight gradient).
//   forward        architectures/dqn.py:42-52  flax nn.Conv (NHWC x HWIO, cross-correlation) + bias + ReLU
//   data gradient  jax.value_and_grad, idqn.py:105: a stride-1 convolution over the zero-bordered dout planes with the
//                  re-indexed kernel (packed by k_stage), one variant per output parity; ReLU mask of the layer below
//
// Work decomposition: a workgroup (512 threads, ONE per CU: waves 0-3 compute, one per SIMD; waves 4-7 only issue the
// LDS-DMA copies) owns `np` consecutive output positions of one (net, batch block) for all CO channels: np * CT tiles of
// 32 samples x 32 channels, dealt round-robin to the compute waves (<= NT each).  The host sizes the items so that one
// launch is at most one workgroup per CU with equal work.
// K loop: a superstep = one kernel row kh and one 16-channel chunk.  Per superstep the loader waves stage
//   * the NQ taps x CT tiles x 3 planes of packed weights (one contiguous run), and
//   * for every input row its positions touch, the STRIP of pixel chunks they read -- neighbouring output positions
//     share pixels (3x3 stride 1: each staged chunk serves 3 taps), which keeps the L2 -> LDS traffic per MFMA low;
// into a ring of 2-3 stage buffers, two supersteps ahead (counted s_waitcnt vmcnt + one s_barrier per superstep), while
// the compute waves run NQ x NT tile-steps of 6 (Conv_0: 3) MFMAs from LDS fragments.
// Orientation: A = activations (rows = samples), B = weights (columns = channels), so a lane ends up with 4 x 4
// consecutive samples of ONE channel: plane rows are written as 8-byte pieces, the bias / mask is one value per lane.
#pragma once
#include "convp.h"

namespace {


__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* p) {
    auto q = (__attribute__((address_space(3))) s16x4*)p;
    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q);
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q + 32);  // 4 rows (256 B) on
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, v0), __builtin_bit_cast(bf16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ s16x4 tr_half(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 join8(s16x4 v0, s16x4 v1) {
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, v0), __builtin_bit_cast(bf16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 frag_lin(const unsigned char* p) {
    return *(const __attribute__((address_space(3))) bf16x8*)p;
}

// b = this workgroup's item index (already remapped XCD-contiguously), nblk = items of the launch (or of this role).
template <int NPA, int CT, int NQ, int NT>
__device__ __forceinline__ void cfwd_body(const CFwdArgs& a, unsigned stage_bytes, int ring, unsigned mask_off, long long* prof,
                                          const int b, const int nblk) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int WB = NQ * CT * 3 * 1024, BLKA = NPA * 1024, NWP = NQ * CT * 3;
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, cl = lane & 31;
    // waves 0-3 compute (one per SIMD), waves 4-7 only issue the LDS-DMA copies: a wave that did both kept its matrix pipe
    // idle for the ~1000 cycles per superstep its ~11 copies take to issue
    const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    // (stamps are unconditional up to the loop: a branch here splits the block and the kernel-argument loads, which the
    // scheduler otherwise batches at the top, become a chain of dependent round trips)
    const long long pt0 = clock64(), pw0 = wall_clock64();
    CItem it;  // derived from the workgroup index (an XCD walks consecutive items of one net): no dependent load
    int vi = 0, item_in_slot = 0;
    {
        // net-major by default: an XCD walks consecutive ranges of one net and shares its packed kernels through L2.
        // Range-major where the nets read the SAME input (Conv_0: K nets per staged minibatch): an XCD then holds a few
        // position ranges of every net, so a pixel strip crosses the fabric once per XCD instead of once per net.
        int slot, rr;
        if (a.range_major) {
            const int n_slots = nblk / a.items_per_slot;
            rr = b / n_slots;
            slot = b - rr * n_slots;
        } else {
            slot = b / a.items_per_slot;
            rr = b - slot * a.items_per_slot;
        }
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < a.n_var && rr >= a.r_begin[i]) vi = i;
        it.net = slot / a.nb;
        it.bb = slot - it.net * a.nb;
        it.var = vi;
        item_in_slot = rr;
        const int r = rr - a.r_begin[vi], npos_v = a.var[vi].OH * a.var[vi].OW, R = a.r_cnt[vi];
        const int base = npos_v / R, rem = npos_v - base * R;
        it.p0 = r * base + min(r, rem);
        it.np = base + (r < rem ? 1 : 0);
    }
    const CVar& v = a.var[vi];
    const int OW = v.OW, p0 = it.p0, np = it.np;
    const int in_slot = (a.in_split > 0 ? (it.net >= a.in_split ? 1 : 0) : it.net) * a.nb + it.bb;
    const int out_slot = it.net * a.nb + it.bb;
    const unsigned long in_base = (unsigned long)a.in + (unsigned long)in_slot * a.in_slot;
    const int oh0 = p0 / OW;
    // strips: input row r of this workgroup serves output columns [c0, c0 + ncol) of output row oh0 + r
    int nx[CP_MAX_STRIPS], c0[CP_MAX_STRIPS];
    unsigned soff[CP_MAX_STRIPS];
    unsigned long sb[CP_MAX_STRIPS];
    {
        unsigned accb = 0;
#pragma unroll
        for (int r = 0; r < CP_MAX_STRIPS; ++r) {
            const int row = oh0 + r;
            const int lo = max(p0, row * OW), hi = min(p0 + np, (row + 1) * OW);
            const int ncol = hi - lo;
            c0[r] = lo - row * OW;
            nx[r] = ncol > 0 ? (ncol - 1) * a.SX + NQ : 0;
            soff[r] = accb;
            accb += (unsigned)nx[r] * BLKA;
            sb[r] = in_base + (unsigned long)(row * a.S + v.in_off_h) * (unsigned long)a.row_bytes +
                    (unsigned long)(c0[r] * a.S + v.in_off_w) * (unsigned long)a.pix_bytes;
        }
    }
    // tiles of this wave: tile index wave + 4 i -> position j = idx / CT, channel tile ct = idx % CT (= wave % CT)
    const int ct = wave % CT;
    unsigned abase[NT];
    int tpos[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int j = (wave + 4 * i) / CT;
        tpos[i] = j < np ? p0 + j : -1;
        const int p = p0 + min(j, np - 1);
        const int oh = p / OW, ow = p - oh * OW, r = oh - oh0;
        int c0r = c0[0];
        unsigned so = soff[0];
#pragma unroll
        for (int s = 1; s < CP_MAX_STRIPS; ++s) {
            c0r = (r == s) ? c0[s] : c0r;
            so = (r == s) ? soff[s] : so;
        }
        abase[i] = WB + so + (unsigned)((ow - c0r) * a.SX) * BLKA;
    }
    const unsigned lo_tr = (8 * h + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&lds[0];
    const unsigned long wb0 = (unsigned long)a.wq + (unsigned long)it.net * a.wq_stride + (unsigned long)v.w_off;
    const int NSS = a.KH * a.NCC;

    auto stage_w = [&](int ss, unsigned buf, int first, int step) {  // the packed kernels of superstep ss
        const unsigned long wsrc = wb0 + (unsigned long)ss * WB;
        for (int i = first; i < NWP; i += step) dma16(lane16, wsrc + (unsigned long)i * 1024, buf + i * 1024);
    };
    auto stage_x = [&](int ss, unsigned buf, int first, int step) {  // its pixel strips
        const int kh = ss / a.NCC, cc = ss - kh * a.NCC;
        const unsigned long so_ = (unsigned long)kh * (unsigned long)a.row_bytes + (unsigned long)cc * 1024;
#pragma unroll
        for (int r = 0; r < CP_MAX_STRIPS; ++r) {
            const unsigned long src = sb[r] + so_;
            const unsigned dst = buf + WB + soff[r];
            for (int x = first; x < nx[r]; x += step) {
#pragma unroll
                for (int pl = 0; pl < NPA; ++pl)
                    dma16(lane16, src + (unsigned long)x * (unsigned long)a.xstep + (unsigned long)pl * (unsigned long)a.plane_bytes,
                          dst + (x * NPA + pl) * 1024);
            }
        }
    };
    auto stage = [&](int ss, unsigned buf, int first, int step) {
        stage_w(ss, buf, first, step);
        stage_x(ss, buf, first, step);
    };

    // epilogue operands that only depend on the item: requested now, used after the loop
    const int co = ct * 32 + cl;
    float bias = 0.f;
    if (a.epilogue == 0)
        bias = (it.net < a.n_first ? a.pbase[0] + (long)it.net * a.pstride : a.pbase[1] + (long)(it.net - a.n_first) * a.pstride)[a.b_off + co];
    // data gradient: the ReLU mask = plane 0 of the forward activation at the output pixel, 2 KB per tile, copied into
    // LDS behind the stage buffers while the last superstep computes
    const unsigned mask_lds = mask_off + wave * (NT * 2048);
    auto tile_out = [&](int p, int& yh, int& yw) {
        const int oh = p / OW, ow = p - oh * OW;
        yh = oh * v.out_mul + v.out_add_h;
        yw = ow * v.out_mul + v.out_add_w;
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const long long pt1 = clock64();
    long long pwait = 0, pt2 = 0;
    if (loader) {
        // ---- loader waves: copy `ring - 1` supersteps ahead while the compute waves work ----------------------------
        // Superstep ss + ring - 1 goes into the buffer superstep ss - 1 was read from, right after barrier ss.  Before
        // barrier ss a loader only needs ITS copies of superstep ss to have landed: vmcnt counts in issue order, so it
        // waits until at most the copies of the younger supersteps (cnt per superstep, the same every time) are left.
        int cnt = (NWP - wave + 3) / 4, cnt_x = 0;
#pragma unroll
        for (int r = 0; r < CP_MAX_STRIPS; ++r) cnt_x += ((nx[r] - wave + 3) / 4) * NPA;
        cnt += cnt_x;
        const int ahead = ring - 1;
        for (int s0 = 0; s0 < ahead && s0 < NSS; ++s0) stage(s0, lds0 + s0 * stage_bytes, wave, 4);
        int nbuf = ahead == 2 ? 2 : 1;  // buffer of superstep ss + ahead
        long long l_wait = 0, l_bar = 0, l_issue = 0;
        for (int ss = 0; ss < NSS; ++ss) {
            const long long c0 = prof ? clock64() : 0;
            wait_vmcnt((ahead == 2 && ss + 1 < NSS) ? cnt : 0);
            const long long c1 = prof ? clock64() : 0;
            __builtin_amdgcn_s_barrier();  // everybody's copies of ss have landed; nobody still reads the buffer re-filled next
            const long long c2 = prof ? clock64() : 0;
            if (ss + ahead < NSS) stage(ss + ahead, lds0 + nbuf * stage_bytes, wave, 4);
            if (prof) { l_wait += c1 - c0; l_bar += c2 - c1; l_issue += clock64() - c2; }
            nbuf = nbuf + 1 == ring ? 0 : nbuf + 1;
            if (ss + 1 == NSS && a.epilogue == 1) {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (tpos[i] < 0) continue;
                    int yh, yw;
                    tile_out(tpos[i], yh, yw);
                    const unsigned long msrc = (unsigned long)a.mask3 + (unsigned long)out_slot * a.mask_slot +
                        ((unsigned long)((yh + a.mask_lo_h) * a.mask_Wp + (yw + a.mask_lo_w)) * (3UL * a.mask_C) + ct * 32) * 64;
                    dma16(lane16, msrc, lds0 + mask_lds + i * 2048);
                    dma16(lane16, msrc + 1024, lds0 + mask_lds + i * 2048 + 1024);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the mask copies
        if (prof && t == 256) {
            long long* pr = prof + 8L * 4096 + (long)b * 8;
            pr[0] = l_wait; pr[1] = l_bar; pr[2] = l_issue; pr[3] = cnt;
        }
    }
    // ---- compute waves -------------------------------------------------------------------------------------------
    // Fragment reads of tile-step u + 1 are issued one per gap BETWEEN the MFMAs of tile-step u (the wave issues an MFMA,
    // is free for the ~32 cycles it runs, and blocks at the next, dependent one): sched_barrier pins that order.
    int cbuf = 0;
    for (int ss = 0; ss < NSS && !loader; ++ss) {
        const unsigned cur_off = cbuf * stage_bytes;
        cbuf = cbuf + 1 == ring ? 0 : cbuf + 1;
        const long long pw = prof ? clock64() : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (prof) { const long long now = clock64(); pwait += now - pw; if (ss == 0) pt2 = now; }
        const unsigned char* cur = lds + cur_off;
        // Fragments: the activation halves of tile-step u + 2 are requested in the gaps between the MFMAs of tile-step u
        // (a ring of three register sets), the weight planes of tap q + 1 during the last-but-one tile-step of tap q:
        // every ds_read has at least one whole tile-step (~190 cycles) to return before an MFMA waits for it.
        constexpr int U = NQ * NT, NM = NPA == 3 ? 6 : 3, WSTEP = NT >= 2 ? NT - 2 : 0;
        bf16x8 wf[2][3];
        s16x4 ah[3][NPA][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[0][pl] = frag_lin(cur + (ct * 3 + pl) * 1024 + lane16);
#pragma unroll
        for (int u0 = 0; u0 < 2 && u0 < U; ++u0)
#pragma unroll
            for (int pl = 0; pl < NPA; ++pl) {
                const unsigned char* ap = cur + abase[u0 % NT] + (u0 / NT) * BLKA + pl * 1024 + lo_tr;
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
                const unsigned char* an = cur + abase[next_a ? in_ : 0] + qn * BLKA + lo_tr;
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
                (void)NM;
            }
        }
    }

    const long long pt3 = prof ? clock64() : 0;
    // ---- epilogue: lane = channel co, registers = samples (r & 3) + 8 (r >> 2) + 4 h -------------------------------
    // A lane's values of a tile are 8-byte pieces of 64-byte rows; written straight to HBM every store instruction would
    // touch 32 rows (measured: ~330 cycles each).  Each wave therefore turns a tile around in LDS (the stage buffers
    // are free now) and stores whole 1 KiB runs: 16 rows of a plane, or 8 rows of the f32 copy, per instruction.
    // The work (bias / mask, the 3-way bf16 split: ~120 VALU instructions per tile) is shared with the loader waves, idle
    // by now: compute wave w keeps its first KEEP tiles and hands the others, as raw accumulators through LDS, to loader
    // wave w + 4, which has the same tile table.
    constexpr int KEEP = (NT + 1) / 2, NH = NT - KEEP;
    const unsigned r_f32 = a.out3 ? 6144 : 0, rsz = r_f32 + (a.out_f32 ? 4096 : 0);
    unsigned char* R = lds + wave8 * rsz;                       // this wave's turn-around tile: [3 planes x 2 KB][f32 4 KB]
    unsigned char* H = lds + 8 * rsz + wave * (NH * 4096);      // hand-off tiles of compute wave `wave`
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the loaders' mask copies have landed; every wave is done with the stage buffers
    if (NH > 0) {
        if (!loader) {
#pragma unroll
            for (int i = KEEP; i < NT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *LDS_PTR(f32x4, H + (i - KEEP) * 4096 + g * 1024 + lane16) =
                        (f32x4){acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned wsw = (cl >> 1) & 3;                                      // plane rows: 16-byte slot ^ (row >> 1) & 3
    const unsigned rsw = (lane * 16) ^ ((((unsigned)lane >> 3) & 3) << 4);   // = row (lane >> 2), slot (lane & 3) ^ swizzle
    const unsigned fsw = (lane * 16) ^ ((((unsigned)lane >> 3) & 7) << 4);   // f32 rows: slot (lane & 7) ^ (row & 7)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (tpos[i] < 0 || loader != (i >= KEEP)) continue;  // wave-uniform
        f32x16 av;
        if (i >= KEEP) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 x = *LDS_PTR(const f32x4, H + (i >= KEEP ? i - KEEP : 0) * 4096 + g * 1024 + lane16);
                av[4 * g] = x[0]; av[4 * g + 1] = x[1]; av[4 * g + 2] = x[2]; av[4 * g + 3] = x[3];
            }
        } else {
            av = acc[i];
        }
        int yh, yw;
        tile_out(tpos[i], yh, yw);
        float val[16];
        if (a.epilogue == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) val[r] = fmaxf(av[r] + bias, 0.f);
        } else {
            const unsigned char* M = lds + mask_lds + i * 2048 + cl * 64 + 8 * h;
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
#pragma unroll
                for (int r = 0; r < 16; ++r) s += val[r];
                s += __shfl_xor(s, 32);
                if (h == 0) a.pb[((long)out_slot * (a.out_H * a.out_W) + (long)yh * a.out_W + yw) * a.CO + co] = s;
            }
        }
        if (a.out3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned q0a, q1a, q2a, q0b, q1b, q2b;
                split3_pk(val[4 * g + 0], val[4 * g + 1], q0a, q1a, q2a);
                split3_pk(val[4 * g + 2], val[4 * g + 3], q0b, q1b, q2b);
                unsigned char* wp = R + cl * 64 + ((g ^ wsw) * 16) + 8 * h;
                *LDS_PTR(u32x2, wp) = (u32x2){q0a, q0b};
                *LDS_PTR(u32x2, wp + 2048) = (u32x2){q1a, q1b};
                *LDS_PTR(u32x2, wp + 4096) = (u32x2){q2a, q2b};
            }
        }
        if (a.out_f32) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *LDS_PTR(f32x4, R + r_f32 + cl * 128 + (((2 * g + h) ^ (cl & 7)) * 16)) =
                    (f32x4){val[4 * g], val[4 * g + 1], val[4 * g + 2], val[4 * g + 3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave, LDS is in order: its writes are visible to its reads
        if (a.out3) {
            unsigned char* O = (unsigned char*)a.out3 + (unsigned long)out_slot * a.out_slot +
                               ((unsigned long)((yh + a.out_lo_h) * a.out_Wp + (yw + a.out_lo_w)) * (3UL * a.CO) + ct * 32) * 64 + lane16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 x = *LDS_PTR(const u32x4, R + pl * 2048 + j * 1024 + rsw);
                    *reinterpret_cast<u32x4*>(O + (unsigned long)pl * a.CO * 64 + j * 1024) = x;
                }
        }
        if (a.out_f32) {
            float* F = a.out_f32 + (long)out_slot * a.f32_slot + (((long)yh * a.f32_W + yw) * a.CO + ct * 32) * 32 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 x = *LDS_PTR(const f32x4, R + r_f32 + j * 1024 + fsw);
                *reinterpret_cast<f32x4*>(F + j * 256) = x;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tile has left LDS before the next one overwrites it
    }
    if (prof && t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* pr = prof + (long)b * 8;
        pr[0] = pw0; pr[1] = pt1 - pt0; pr[2] = pt2 - pt1; pr[3] = pt3 - pt2; pr[4] = clock64() - pt3; pr[5] = pwait; pr[6] = wall_clock64();
        pr[7] = it.np;
    }
}


}  // namespace
