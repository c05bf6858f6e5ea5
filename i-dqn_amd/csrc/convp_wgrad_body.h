// Plane-layout conv weight gradient: the workgroup body (convp.h has the layouts and the arithmetic).
// Kernels: convp_wgrad.hip (one layer per launch), convp_pair.hip (beside a data gradient).
//   gW[kh][kw][ci][co] = sum over (oh, ow, sample b) of  x[oh*S + kh][ow*S + kw][ci][b] * dy[oh][ow][co][b]
//   (jax.value_and_grad of the flax conv, idqn.py:105); the bias gradient is the sum of dy, taken from the per-position
//   sums `pb` its producer already wrote.
// The contraction index is (position, sample): k = the 32 samples of a row, so an MFMA fragment is 16 contiguous bytes of
// ONE row -- rows of x ([kw, ci] = M side) and of dy ([co] = N side) are copied into LDS by LDS-DMA with the four 16-byte
// slots of every 64-byte row XOR-swizzled by (row >> 2) & 3 (on the SOURCE address: the DMA writes LDS linearly), which
// makes the ds_read_b128 fragment reads conflict-free.
// Workgroup = (head, kernel row kh, chunk of output positions) [Conv_0: all 8 kernel rows]; its (KW * CI / 32) x (CO / 32)
// tiles of 32 x 32 are dealt to the 4 waves; a stage = up to PG consecutive positions of one output row: the strip of
// input pixels they read (shared between neighbours) + their dy pixels, double-buffered, one barrier per stage.
// Every workgroup writes its partial sums to its own slab; k_adam adds the slabs in chunk order (no atomics: replicas of
// a data-parallel run must stay bit-identical).
#pragma once
#include "convp.h"

namespace {


__device__ __forceinline__ bf16x8 frag128(const unsigned char* p) {
    return *(const __attribute__((address_space(3))) bf16x8*)p;
}

// b = this workgroup's item index (already remapped XCD-contiguously)
template <int NPX, int CT, int NTW, int PG>
__device__ __forceinline__ void cwgrad_body(const CWgradArgs& a, unsigned stage_bytes, int MT, const int b) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, cl = lane & 31;
    // waves 0-3 compute (one per SIMD), waves 4-7 only issue the LDS-DMA copies (as in convp_fwd.hip)
    const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    CWItem it;  // derived from the workgroup index (the kernel rows of one chunk are neighbours: they share an XCD)
    {
        const int kh = b % a.kh_per_item, r = b / a.kh_per_item;
        const int o = a.chunk_major ? r / a.K : r / a.n_chunks, i = a.chunk_major ? r - o * a.K : r - o * a.n_chunks;
        it.net = a.chunk_major ? i : o;
        it.chunk = a.chunk_major ? o : i;
        it.kh = kh;
        const int npos = a.OH * a.OW, base = npos / a.n_chunks, rem = npos - base * a.n_chunks;
        it.p0 = it.chunk * base + min(it.chunk, rem);
        it.np = base + (it.chunk < rem ? 1 : 0);
    }
    const int OW = a.OW, p0 = it.p0, p_end = it.p0 + it.np, k = it.net;
    const int dy_pix = 3 * a.CO * 64;
    // LDS stage: [x region][PG dy pixels]
    const unsigned strip_bytes = (PG + 1) * 1024;  // Conv_0: one strip per kernel row, (4 PG + 4) pixels of 256 B
    const unsigned XB = NPX == 3 ? (unsigned)(((PG - 1) * a.S + a.KW) * a.x_pix) : (unsigned)a.KH * strip_bytes;
    const unsigned pos_stride = (unsigned)(a.S * a.x_pix);
    const int ct = wave % CT;
    const int NH = a.CI / 32;
    unsigned tbase[NTW];
    int tm[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int m = (wave + 4 * i) / CT;
        tm[i] = m < MT ? m : -1;
        const int mm = min(m, MT - 1);
        tbase[i] = NPX == 3 ? (unsigned)((mm / NH) * a.x_pix + (mm % NH) * 2048) : (unsigned)mm * strip_bytes;
    }
    const unsigned swz = (cl >> 2) & 3;
    const unsigned rd0 = cl * 64 + ((0u + h) ^ swz) * 16, rd1 = cl * 64 + ((2u + h) ^ swz) * 16;  // k-step 0 / 1
    const unsigned voff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);          // swizzled DMA source
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&lds[0];
    const unsigned long xb = (unsigned long)a.x, dyb = (unsigned long)a.dy;

    auto cnt_of = [&](int pos) { return min(PG, min((pos / OW + 1) * OW, p_end) - pos); };
    // NPX == 3: the x region of the two stages is ONE ring of 2 strips of pixels.  Consecutive stages of an output row slide
    // along the same input row, so a stage copies only the pixels its predecessor did not (PG * S of (PG - 1) * S + KW) and
    // appends them behind the window; a stage that starts a row (or a batch block) appends its whole strip.  The window of
    // the stage being read and the pixels appended meanwhile are each at most one strip: they never overlap in two strips.
    const unsigned ring_bytes = 2 * XB;
    // window state of a stage, kept in step by loader and compute waves: start of its window and end of the data, in ring bytes
    auto advance = [&](int pos, unsigned& wstart, unsigned& wend, int& first_col, int& n_new) {
        const int oh = pos / OW, ow0 = pos - oh * OW, cnt = cnt_of(pos);
        const bool cont = pos != p0 && ow0 != 0;  // same input row as the stage before
        const int lo = ow0 * a.S, hi = (ow0 + cnt - 1) * a.S + a.KW;
        first_col = cont ? lo - a.S + a.KW : lo;  // the previous stage ended with position ow0 - 1
        n_new = hi - first_col;
        if (cont) {
            wstart += (unsigned)(PG * a.S * a.x_pix);  // the previous stage of a row always holds PG positions
            if (wstart >= ring_bytes) wstart -= ring_bytes;
        } else {
            wstart = wend;
        }
        wend += (unsigned)(n_new * a.x_pix);
        if (wend >= ring_bytes) wend -= ring_bytes;
    };
    unsigned l_wstart = 0, l_wend = 0;  // the loader's copy of the window state
    auto stage = [&](int bb, int pos, unsigned par) {
        const int oh = pos / OW, ow0 = pos - oh * OW, cnt = cnt_of(pos);
        const unsigned long xs = xb + (unsigned long)(a.x_shared ? bb : k * a.nb + bb) * (unsigned long)a.x_slot;
        const unsigned buf = lds0 + par * stage_bytes;
        unsigned dy_dst;
        if (NPX == 3) {
            const unsigned at = l_wend;  // append here
            int first_col, n_new;
            advance(pos, l_wstart, l_wend, first_col, n_new);
            const unsigned long src = xs + (unsigned long)(oh * a.S + it.kh) * (unsigned long)a.x_row +
                                      (unsigned long)first_col * (unsigned long)a.x_pix;
            const int npiece = (n_new * a.x_pix) >> 10;
            for (int i = wave; i < npiece; i += 4) {
                unsigned d = at + i * 1024;
                if (d >= ring_bytes) d -= ring_bytes;
                dma16(voff, src + (unsigned long)i * 1024, lds0 + d);
            }
            dy_dst = lds0 + ring_bytes + par * (unsigned)(PG * dy_pix);
        } else {
            const int npiece = cnt + 1;
            for (int kh = 0; kh < a.KH; ++kh) {
                const unsigned long src = xs + (unsigned long)(oh * a.S + kh) * (unsigned long)a.x_row + (unsigned long)ow0 * 1024;
                for (int i = wave; i < npiece; i += 4) dma16(voff, src + (unsigned long)i * 1024, buf + kh * strip_bytes + i * 1024);
            }
            dy_dst = buf + XB;
        }
        const unsigned long dsrc = dyb + (unsigned long)(k * a.nb + bb) * (unsigned long)a.dy_slot +
                                   ((unsigned long)(oh + a.dy_lo_h) * a.dy_Wp + (ow0 + a.dy_lo_w)) * (unsigned long)dy_pix;
        const int ndp = (cnt * dy_pix) >> 10;
        for (int i = wave; i < ndp; i += 4) dma16(voff, dsrc + (unsigned long)i * 1024, dy_dst + i * 1024);
    };

    if (loader) {
        // ---- loader waves: copy stage s + 1 while the compute waves work on stage s -----------------------------------
        int ibb = 0, ipos = p0, par = 0;
        stage(ibb, ipos, 0u);
        // bias gradient of this chunk: the per-position sums of dy, added in (batch block, position) order -- by the first
        // loader wave while the first stage travels (on a compute wave these loads sat in front of its MFMA loop)
        if (wave == 0 && a.pb && (NPX == 1 || it.kh == 0)) {
            float bsum = 0.f;
            const int bcol = min(lane, a.CO - 1);
            const int npos = a.OH * a.OW, n = it.np * a.nb;
            for (int e0 = 0; e0 < n; e0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u, n - 1), bb = e / it.np, p = p0 + (e - bb * it.np);
                    v[u] = a.pb[((long)(k * a.nb + bb) * npos + p) * a.CO + bcol];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < n) bsum += v[u];
            }
            if (lane < a.CO) (a.slab + ((long)it.chunk * a.K + k) * a.slab_stride)[(long)a.KH * a.KW * a.CI * a.CO + lane] = bsum;
        }
        ipos += cnt_of(ipos);
        if (ipos >= p_end) { ipos = p0; ++ibb; }
        int cbb = 0, cpos = p0;  // mirrors the compute waves' progress (same barrier count)
        while (cbb < a.nb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ibb < a.nb) {
                stage(ibb, ipos, (unsigned)(par ^ 1));
                ipos += cnt_of(ipos);
                if (ipos >= p_end) { ipos = p0; ++ibb; }
            }
            cpos += cnt_of(cpos);
            if (cpos >= p_end) { cpos = p0; ++cbb; }
            par ^= 1;
        }
        return;
    }

    // ---- compute waves -----------------------------------------------------------------------------------------------
    f32x16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // Tile-steps of a stage, flattened: u = (position pp, k-step ks, tile i); the x fragments of tile-step u + 2 and the dy
    // fragments of the next (pp, ks) are requested in the gaps between the MFMAs of tile-step u (sched_barrier pins that);
    // reads run ahead unconditionally (always inside the stage buffer), only the MFMAs of positions past the stage's
    // count are skipped (wave-uniform).
    constexpr int U = PG * 2 * NTW, DSTEP = NTW >= 2 ? NTW - 2 : 0;
    const unsigned char* zero_blk = lds + 2 * stage_bytes;  // 2 KB of zeros behind the two stages
    *LDS_PTR(u32x4, lds + 2 * stage_bytes + wave * 512 + cl * 16) = (u32x4){0u, 0u, 0u, 0u};  // published by the first barrier
    int cbb = 0, cpos = p0, par = 0;
    unsigned c_wstart = 0, c_wend = 0;  // window state (NPX == 3), advanced exactly like the loader's
    while (cbb < a.nb) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* cur = lds + par * stage_bytes;
        const int cnt = cnt_of(cpos);
        if (NPX == 3) {
            int fc, nn;
            advance(cpos, c_wstart, c_wend, fc, nn);
        }
        auto xaddr = [&](int u) {  // LDS address of the x fragment rows of tile-step u (plane 0)
            const int i = u % NTW, g = u / NTW, ks = g & 1, pp = g >> 1;
            if (NPX == 3) {  // pixel (position pp, tap of tile i) of this stage's window in the ring (all wave-uniform)
                unsigned off = c_wstart + pp * pos_stride + tbase[i];
                if (off >= ring_bytes) off -= ring_bytes;
                return (pp < cnt ? lds + off : zero_blk) + (ks ? rd1 : rd0);
            }
            return (pp < cnt ? cur + pp * pos_stride + tbase[i] : zero_blk) + (ks ? rd1 : rd0);
        };
        auto xplane = [&](int u) { return (u / NTW >> 1) < cnt ? a.x_plane : 0; };
        // positions past the stage's count read dy from a block of zeros and x from the same zeros (stale LDS may hold NaN patterns: 0 x NaN would poison the sum): no branch around
        // the MFMAs -- a conditional accumulator update made hipcc shuffle whole accumulators through v_accvgpr moves
        auto daddr = [&](int g) {
            const int ks = g & 1, pp = g >> 1;
            const unsigned char* dyb_ = NPX == 3 ? lds + ring_bytes + par * (PG * dy_pix) : cur + XB;
            return (pp < cnt ? dyb_ + pp * dy_pix + ct * 2048 : zero_blk) + (ks ? rd1 : rd0);
        };
        auto dplane = [&](int g) { return (g >> 1) < cnt ? a.CO * 64 : 0; };
        bf16x8 xf[3][NPX], df[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) df[0][pl] = frag128(daddr(0) + pl * dplane(0));
#pragma unroll
        for (int u0 = 0; u0 < 2 && u0 < U; ++u0)
#pragma unroll
            for (int pl = 0; pl < NPX; ++pl) xf[u0][pl] = frag128(xaddr(u0) + pl * xplane(u0));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u % NTW, g = u / NTW;
            const bool next_x = u + 2 < U, next_d = (i == DSTEP && g + 1 < 2 * PG);
            const unsigned char* xn = xaddr(next_x ? u + 2 : 0);
            const int xpl = xplane(next_x ? u + 2 : 0);
            const unsigned char* dn = daddr(next_d ? g + 1 : 0);
            const int dpl = dplane(next_d ? g + 1 : 0);
            const bf16x8* X = xf[u % 3];
            const bf16x8* D = df[g & 1];
            __builtin_amdgcn_sched_barrier(0);
#define CW_GAP(m)                                                                                          \
    {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (next_x && (m) < NPX) xf[(u + 2) % 3][(m) < NPX ? (m) : 0] = frag128(xn + (m) * xpl);      \
        if (next_d && (m) < 3) df[(g + 1) & 1][(m) < 3 ? (m) : 0] = frag128(dn + (m) * dpl);        \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
            if (NPX == 3) {  // smallest terms first
                acc[i] = mfma_bf16(X[2], D[0], acc[i]);
                CW_GAP(0)
                acc[i] = mfma_bf16(X[0], D[2], acc[i]);
                CW_GAP(1)
                acc[i] = mfma_bf16(X[1], D[1], acc[i]);
                CW_GAP(2)
                acc[i] = mfma_bf16(X[1], D[0], acc[i]);
                CW_GAP(3)
                acc[i] = mfma_bf16(X[0], D[1], acc[i]);
                CW_GAP(4)
                acc[i] = mfma_bf16(X[0], D[0], acc[i]);
                CW_GAP(5)
            } else {
                acc[i] = mfma_bf16(X[0], D[2], acc[i]);
                CW_GAP(0)
                acc[i] = mfma_bf16(X[0], D[1], acc[i]);
                CW_GAP(1)
                acc[i] = mfma_bf16(X[0], D[0], acc[i]);
                CW_GAP(2)
            }
#undef CW_GAP
        }
        cpos += cnt;
        if (cpos >= p_end) { cpos = p0; ++cbb; }
        par ^= 1;
    }

    // (non-temporal, as k_adam's loads of them: the slabs are written once and read once -- 18 MB per step that need not take room in the
    // memory-side cache from the online Dense_0 kernels; step -1.6 us over four interleaved rounds, profiles/r5_d0_keep_online_ab.txt)
    float* S = a.slab + ((long)it.chunk * a.K + k) * a.slab_stride;
    const long row_base = NPX == 3 ? (long)it.kh * a.KW * a.CI : 0;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (tm[i] < 0) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_nontemporal_store(acc[i][r] / a.out_div, &S[(row_base + tm[i] * 32 + mfma_row(r, h)) * a.CO + ct * 32 + cl]);
    }
}


}  // namespace
