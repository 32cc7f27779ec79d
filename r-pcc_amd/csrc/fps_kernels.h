// fps_kernels.h -- exact tile-pruned farthest point sampling (a6) and the ground-mask kernel that performs
// its first pass (a3+a5 + FPS pass 1); included by rpcc_hip.hip.
//
// Tiles hold 64 * FPS_NH points = FPS_NH per lane of one wavefront ("half" h in 0 .. FPS_NH-1; FPS_NH = 4: 256 points):
//   range image:  tile t = (tr, tc) covers rows 2*FPS_NH*tr .. +2*FPS_NH-1, columns 32*tc .. 32*tc+31 (compact in 3-D);
//                 lane l, half h  ->  row 2*FPS_NH*tr + 2*h + (l >> 5), column 32*tc + (l & 31)
//   point list:   tile t covers indices 64*FPS_NH*t ..; lane l, half h -> 64*FPS_NH*t + 64*h + l
// In both layouts (half, lane) in lexicographic order is increasing point index, which is what the
// lowest-index tie rule of the arg-max needs.  Measured on 64x2048 (FPS alone / step with batches in flight): 128-point
// tiles 583 us / 1.10 ms, 256-point tiles 527 us / 1.06 ms, 512-point tiles 536 us / 1.10 ms -- larger tiles re-read a
// little more per visit (+10 % points) but halve the per-tile work of the test / select phases and of the box and
// arg-max reductions, and the tile table shrinks to 25 KB of LDS.
//
// Per tile the workgroup keeps in LDS (FpsLds, 11 dwords): the bounding box of the tile's candidates, the
// tile's current maximum of temp with its (lowest) index, and that point's coordinates.  For a new centre
// c a tile can only change if some point is closer to c than its temp, i.e. only if
//     bound(c, box) < tile_max,   bound = ((bx*bx)+(by*by))+(bz*bz),  b* = per-axis gap to the box.
// bound is evaluated with the SAME fp32 operation sequence as the point distance on per-axis gaps that
// are <= every candidate's |d*| (rounding is monotone), so bound <= computed distance of every candidate
// and skipping is bit-exact, not approximate (DESIGN.md "FPS").  Everything else -- min with temp, strict
// '>' arg-max with lowest-index ties -- is the brute-force definition.
#pragma once

#define FPS_TAB_ROWS 11  // lo[3], hi[3], tmax, cx[3], targ
#ifndef FPS_NH
#define FPS_NH 4  // wavefront-loads ("halves") per tile: 64 * FPS_NH points, 2 * FPS_NH rows x 32 columns of a range image
#endif
#define FPS_TILE (64 * FPS_NH)
#define FPS_TROWS (2 * FPS_NH)

struct FpsTiling {
    int N;      // points per frame (P for a range image)
    int W, H;   // range image shape (RANGE only)
    int tcols;  // tiles per tile-row (RANGE only)
    int T;      // tiles per frame
};
static inline FpsTiling fps_tiling_range(int H, int W) {
    FpsTiling g;
    g.N = H * W; g.W = W; g.H = H; g.tcols = (W + 31) / 32; g.T = ((H + FPS_TROWS - 1) / FPS_TROWS) * g.tcols;
    return g;
}
static inline FpsTiling fps_tiling_list(int N) {
    FpsTiling g;
    g.N = N; g.W = 0; g.H = 0; g.tcols = 0; g.T = (N + FPS_TILE - 1) / FPS_TILE;
    return g;
}

// index of (tile, half, lane); -1 when outside
template <bool RANGE>
__device__ __forceinline__ int fps_tile_point(const FpsTiling &g, int t, int half, int lane) {
    if (RANGE) {
        const int tr = t / g.tcols, tc = t - tr * g.tcols;
        const int row = FPS_TROWS * tr + 2 * half + (lane >> 5), col = 32 * tc + (lane & 31);
        return (row < g.H && col < g.W) ? row * g.W + col : -1;
    }
    const int p = t * FPS_TILE + half * 64 + lane;
    return p < g.N ? p : -1;
}
template <bool RANGE>
__device__ __forceinline__ int fps_tile_of(const FpsTiling &g, int p) {
    if (RANGE) {
        const int row = p / g.W, col = p - row * g.W;
        return (row / FPS_TROWS) * g.tcols + (col >> 5);
    }
    return p / FPS_TILE;
}

struct FpsLds {
    float *lo[3], *hi[3], *tmax, *cx[3];
    uint32_t *targ;
    uint32_t *torg;   // range image: first pixel of the tile (22 bits) | valid columns - 1 (5 bits) << 22 | valid rows - 1 (5 bits) << 27
    uint16_t *work;
    __device__ FpsLds(unsigned char *base, int T) {
        float *f = reinterpret_cast<float *>(base);
        for (int a = 0; a < 3; a++) { lo[a] = f + (size_t)a * T; hi[a] = f + (size_t)(3 + a) * T; cx[a] = f + (size_t)(7 + a) * T; }
        tmax = f + (size_t)6 * T;
        targ = reinterpret_cast<uint32_t *>(f + (size_t)10 * T);
        torg = reinterpret_cast<uint32_t *>(f + (size_t)11 * T);
        work = reinterpret_cast<uint16_t *>(f + (size_t)12 * T);
    }
};
static inline size_t fps_tiled_lds_bytes(int T) { return (size_t)T * 50 + 64; }
#define FPS_TILED_MAX_TILES 3200  // 50 B/tile must fit the 160 KiB LDS of one CU

// Per-tile reductions shared by the FPS kernel and the ground-mask kernel.  x/y/z/nt: the lane's two
// points; cand: they take part in the bounding box; valid: they exist.  Writes the 11 table values of
// tile t through `put(row, value)` from lanes 0..10 (one dword each).
struct TileStats {
    float v[FPS_TAB_ROWS];
};
__device__ __forceinline__ void fps_tile_argmax(const float (&x)[FPS_NH], const float (&y)[FPS_NH], const float (&z)[FPS_NH],
                                                const float (&nt)[FPS_NH], const bool (&valid)[FPS_NH],
                                                const int (&pidx)[FPS_NH], float &wt, float &wx, float &wy, float &wz,
                                                uint32_t &widx) {
    // largest value, lowest (half, lane) among equals
    uint32_t o[FPS_NH], om = 0u;
#pragma unroll
    for (int h = 0; h < FPS_NH; h++) {
        o[h] = (!valid[h] || nt[h] < 0.0f) ? 0u : f2u(nt[h]) + 1u;
        om = o[h] > om ? o[h] : om;
    }
    const uint32_t vmax = dpp_max_u32(om);
    bool done = false;
    wt = wx = wy = wz = 0.0f;
    widx = 0u;
#pragma unroll
    for (int h = 0; h < FPS_NH; h++) {
        const unsigned long long m = __ballot(o[h] == vmax);
        if (!done && m) {  // wave-uniform: a scalar branch instead of per-lane selects
            const int wl = (int)__ffsll((long long)m) - 1;
            wt = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(nt[h]), wl));
            wx = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(x[h]), wl));
            wy = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(y[h]), wl));
            wz = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(z[h]), wl));
            widx = (uint32_t)__builtin_amdgcn_readlane(pidx[h], wl);
            done = true;
        }
    }
    if (vmax == 0u) wt = -1.0f;
}
__device__ __forceinline__ void fps_tile_box(const float (&x)[FPS_NH], const float (&y)[FPS_NH], const float (&z)[FPS_NH],
                                             const bool (&cand)[FPS_NH], float (&lo)[3], float (&hi)[3]) {
    const float inf = __builtin_inff();
    lo[0] = lo[1] = lo[2] = inf;
    hi[0] = hi[1] = hi[2] = -inf;
#pragma unroll
    for (int h = 0; h < FPS_NH; h++) {
        lo[0] = fminf(lo[0], cand[h] ? x[h] : inf); hi[0] = fmaxf(hi[0], cand[h] ? x[h] : -inf);
        lo[1] = fminf(lo[1], cand[h] ? y[h] : inf); hi[1] = fmaxf(hi[1], cand[h] ? y[h] : -inf);
        lo[2] = fminf(lo[2], cand[h] ? z[h] : inf); hi[2] = fmaxf(hi[2], cand[h] ? z[h] : -inf);
    }
    dpp_box6(lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
}

// RANGE: point k = (ri[k]*tx[k], ri[k]*ty[k], ri[k]*tz[k]) with SoA rays; else AoS xyz[k*3..].
template <bool RANGE>
__device__ __forceinline__ void fps_load_point(const float *__restrict__ src, const float *__restrict__ tx,
                                               const float *__restrict__ ty, const float *__restrict__ tz, int k,
                                               float &x, float &y, float &z) {
    if (RANGE) {
        const uint32_t o = (uint32_t)k * 4u;  // byte offset from the wave-uniform bases (P * 4 < 2^32)
        const float r = ld_f32(src, o);
        x = r * ld_f32(tx, o); y = r * ld_f32(ty, o); z = r * ld_f32(tz, o);
    } else {
        x = src[3 * (int64_t)k]; y = src[3 * (int64_t)k + 1]; z = src[3 * (int64_t)k + 2];
    }
}

#define FPS_THREADS 1024

// Threads of the tile-pruned FPS workgroup (one workgroup per frame).  Measured with 256-point tiles on 64x2048 (kernel
// alone / step with three batches in flight): 1024 threads 452 us / 1.14 ms, 768 threads 474 us / 1.12 ms, 512 threads
// 527 us / 1.07 ms, 384 threads 636 us / 1.10 ms.  Alone the wide workgroup wins; with batches in flight two narrow ones
// leave wave slots to the throughput kernels of the other batches, and the step is what counts.
// Small batches (fewer frames than half the CUs) have nothing to co-schedule with and take the 1024-thread form.
#define FPS_TT_BATCH 512
#define FPS_TT_SMALL 1024
template <bool RANGE, int FPS_TT>
__global__ __launch_bounds__(FPS_TT) void fps_tiled_kernel(const float *__restrict__ src,
                                                                const float *__restrict__ tx,
                                                                const float *__restrict__ ty,
                                                                const float *__restrict__ tz, float *__restrict__ temp,
                                                                const int32_t *__restrict__ info, FpsTiling g, int M,
                                                                int32_t *__restrict__ out_idx,
                                                                float *__restrict__ out_cen,
                                                                const float *__restrict__ tiletab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fps_smem[];
    __shared__ unsigned long long red[FPS_TT / 64];
    __shared__ int redt[FPS_TT / 64];
    __shared__ int wcount;
    const int T = g.T, N = g.N;
    FpsLds L(fps_smem, T);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    src += (int64_t)b * N * (RANGE ? 1 : 3);
    temp += (int64_t)b * N;
    out_idx += (int64_t)b * M;
    if (out_cen) out_cen += (int64_t)b * M * 3;
    if (M <= 0) return;

    int old = 0;
    if (RANGE) { old = info[4 * b + 1]; if (old >= N) old = 0; }
    float c0, c1, c2;
    fps_load_point<RANGE>(src, tx, ty, tz, old, c0, c1, c2);
    if (tid == 0) {
        out_idx[0] = old;
        if (out_cen) { out_cen[0] = c0; out_cen[1] = c1; out_cen[2] = c2; }
        wcount = 0;
    }

    if (RANGE) {
        for (int t = tid; t < T; t += FPS_TT) {
            const int tr = t / g.tcols, tc = t - tr * g.tcols;
            const int ncol = min(32, g.W - 32 * tc), nrow = min(FPS_TROWS, g.H - FPS_TROWS * tr);
            L.torg[t] = (uint32_t)(FPS_TROWS * tr * g.W + 32 * tc) | ((uint32_t)(ncol - 1) << 22) | ((uint32_t)(nrow - 1) << 27);
        }
    }
    __syncthreads();
    // A tile's data in registers (loads are issued for a group of tiles before any is consumed, so the
    // memory latency of a round is paid once per group; all loads are unconditional on clamped indices).
    struct TileRegs { float x[FPS_NH], y[FPS_NH], z[FPS_NH], tp[FPS_NH]; int p[FPS_NH]; };
    const int lrow = lane >> 5, lcol = lane & 31;
    const int loff0 = lrow * g.W + lcol;  // lane's pixel offset inside a tile, half 0 (half h: + 2 * h * W)
    auto load_tile = [&](int t, TileRegs &q) {
        const uint32_t org = RANGE ? L.torg[t] : 0u;
#pragma unroll
        for (int h = 0; h < FPS_NH; h++) {
            if (RANGE) {
                const bool ok = lcol <= (int)((org >> 22) & 31u) && 2 * h + lrow <= (int)(org >> 27);
                q.p[h] = ok ? (int)(org & 0x3FFFFFu) + loff0 + 2 * h * g.W : -1;
            } else {
                q.p[h] = fps_tile_point<RANGE>(g, t, h, lane);
            }
            const int pc = q.p[h] < 0 ? 0 : q.p[h];
            fps_load_point<RANGE>(src, tx, ty, tz, pc, q.x[h], q.y[h], q.z[h]);
            q.tp[h] = ld_f32(temp, (uint32_t)pc * 4u);
            if (q.p[h] < 0) q.tp[h] = -1.0f;  // not a candidate; its (clamped-load) coordinates are never used
        }
    };
    // distance update against the current centre, tile maximum, (optionally) bounding box
    auto compute_tile = [&](int t, const TileRegs &q, bool with_box) {
        bool valid[FPS_NH], cand[FPS_NH];
        float nt[FPS_NH];
#pragma unroll
        for (int h = 0; h < FPS_NH; h++) {
            valid[h] = q.p[h] >= 0;
            cand[h] = q.tp[h] >= 0.0f;
            const float dx = q.x[h] - c0, dy = q.y[h] - c1, dz = q.z[h] - c2;
            const float d = (dx * dx + dy * dy) + dz * dz;  // sampling_gpu.cu:64, un-fused
            nt[h] = d < q.tp[h] ? d : q.tp[h];  // == fminf(d, tp): a NaN distance keeps tp, tp itself is never NaN
            if (valid[h] && nt[h] != q.tp[h]) st_f32(temp, (uint32_t)q.p[h] * 4u, nt[h]);
        }
        // nothing changed in this tile: its table entry (maximum, arg, coordinates) is still exact
        bool changed = false;
#pragma unroll
        for (int h = 0; h < FPS_NH; h++) changed |= valid[h] && nt[h] != q.tp[h];
        if (!with_box && __ballot(changed) == 0ull) return;
        if (with_box) {
            float lo[3], hi[3];
            fps_tile_box(q.x, q.y, q.z, cand, lo, hi);
            if (lane == 0) { L.lo[0][t] = lo[0]; L.lo[1][t] = lo[1]; L.lo[2][t] = lo[2]; L.hi[0][t] = hi[0]; L.hi[1][t] = hi[1]; L.hi[2][t] = hi[2]; }
        }
        float wt, wx, wy, wz;
        uint32_t widx;
        fps_tile_argmax(q.x, q.y, q.z, nt, valid, q.p, wt, wx, wy, wz, widx);
        if (lane == 0) { L.tmax[t] = wt; L.targ[t] = widx; L.cx[0][t] = wx; L.cx[1][t] = wy; L.cx[2][t] = wz; }
    };

    // arg-max over the tile table -> next centre (index and coordinates)
    auto select_next = [&]() {
        uint32_t hi = 0u, ix = 0xFFFFFFFFu;  // orderable value, index
        int bt = 0;
        for (int t = tid; t < T; t += FPS_TT) {
            const float v = L.tmax[t];
            const uint32_t h = (v < 0.0f) ? 0u : f2u(v) + 1u, i = L.targ[t];
            if (h > hi || (h == hi && i < ix)) { hi = h; ix = i; bt = t; }
        }
        uint32_t vmax = dpp_max_u32(hi);
        uint32_t imin = dpp_min_u32(hi == vmax ? ix : 0xFFFFFFFFu);
        {
            const unsigned long long mm = __ballot(hi == vmax && ix == imin);
            const int wl = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)mm) - 1);
            const int wt_ = __builtin_amdgcn_readlane(bt, wl < 0 ? 0 : wl);
            if (lane == 0) { red[wave] = ((unsigned long long)vmax << 32) | imin; redt[wave] = wt_; }
        }
        __syncthreads();
        const unsigned long long k = red[lane % (FPS_TT / 64)];
        const int kt = redt[lane % (FPS_TT / 64)];
        hi = (uint32_t)(k >> 32); ix = (uint32_t)k;
        vmax = dpp_max_u32(hi);
        imin = dpp_min_u32(hi == vmax ? ix : 0xFFFFFFFFu);
        const unsigned long long mm = __ballot(hi == vmax && ix == imin);
        const int wl = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)mm) - 1);
        const int t = __builtin_amdgcn_readlane(kt, wl < 0 ? 0 : wl);
        if (imin == 0xFFFFFFFFu) {  // no candidate anywhere: keep indices defined (the reference would fail)
            old = 0;
            fps_load_point<RANGE>(src, tx, ty, tz, 0, c0, c1, c2);
        } else {
            old = (int)imin;
            c0 = L.cx[0][t]; c1 = L.cx[1][t]; c2 = L.cx[2][t];
        }
    };

    constexpr int NW = FPS_TT / 64, GROUP = FPS_NH >= 8 ? 1 : 8 / FPS_NH;  // tiles per wavefront in flight: 512 points (256 and 1024 measured: no better)
    DBG_STAMP(8);
    // first centre: every tile is visited once (also builds the boxes) -- unless ground_mask already did
    // that pass and left the tile table (info[b][3] == 1)
    const bool have_tab = RANGE && tiletab != nullptr && info[4 * b + 3] == 1;
    if (M > 1 && have_tab) {
        const float *tab = tiletab + (int64_t)b * FPS_TAB_ROWS * T;
        float *dst = reinterpret_cast<float *>(fps_smem);
        if ((T & 3) == 0) {  // 16-byte copies (the table of a frame starts at a multiple of 16 bytes then)
            const float4 *t4 = reinterpret_cast<const float4 *>(tab);
            float4 *d4 = reinterpret_cast<float4 *>(dst);
            for (int i = tid; i < FPS_TAB_ROWS * T / 4; i += FPS_TT) d4[i] = t4[i];
        } else {
            for (int i = tid; i < FPS_TAB_ROWS * T; i += FPS_TT) dst[i] = tab[i];
        }
        __syncthreads();
    } else if (M > 1) {
        for (int t = wave; t < T; t += NW * GROUP) {
            TileRegs q[GROUP];
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) if (t + gi * NW < T) load_tile(t + gi * NW, q[gi]);
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) if (t + gi * NW < T) compute_tile(t + gi * NW, q[gi], true);
        }
        __syncthreads();
    }
    if (M > 1) {
        DBG_STAMP(9);
        select_next();
        DBG_STAMP(10);
        if (tid == 0) { out_idx[1] = old; if (out_cen) { out_cen[3] = c0; out_cen[4] = c1; out_cen[5] = c2; } }
    }
    long long acc_a = 0, acc_b = 0, acc_c = 0, acc_n = 0, tq = 0;
    const bool prof = g_dbg_stamps != nullptr && blockIdx.x == 0 && tid == 0;
    for (int j = 2; j < M; j++) {
        if (prof) tq = (long long)__builtin_readcyclecounter();
        // tile test against the new centre; active tiles go to the work list
        for (int t = tid; t < T; t += FPS_TT) {
            const float g0 = fmaxf(fmaxf(L.lo[0][t] - c0, c0 - L.hi[0][t]), 0.0f);
            const float g1 = fmaxf(fmaxf(L.lo[1][t] - c1, c1 - L.hi[1][t]), 0.0f);
            const float g2 = fmaxf(fmaxf(L.lo[2][t] - c2, c2 - L.hi[2][t]), 0.0f);
            const float bound = (g0 * g0 + g1 * g1) + g2 * g2;
            const bool act = bound < L.tmax[t];
            const unsigned long long m = __ballot(act);
            if (m) {
                int base = 0;
                if (lane == (int)__ffsll((long long)m) - 1) base = atomicAdd(&wcount, __popcll(m));
                base = __shfl(base, (int)__ffsll((long long)m) - 1, 64);
                if (act) L.work[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)t;
            }
        }
        __syncthreads();
        const int n = wcount;
        if (prof) { const long long t1 = (long long)__builtin_readcyclecounter(); acc_a += t1 - tq; tq = t1; acc_n += n; }
        for (int e = wave; e < n; e += NW * GROUP) {
            TileRegs q[GROUP];
            int tt[GROUP];
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) {
                const int ee = e + gi * NW;
                tt[gi] = (int)L.work[ee < n ? ee : n - 1];
                load_tile(tt[gi], q[gi]);   // unconditional (a repeated tile for the tail is harmless and unused)
            }
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) if (e + gi * NW < n) compute_tile(tt[gi], q[gi], false);
        }
        __syncthreads();
        if (prof) { const long long t1 = (long long)__builtin_readcyclecounter(); acc_b += t1 - tq; tq = t1; }
        if (tid == 0) wcount = 0;
        select_next();
        if (prof) { const long long t1 = (long long)__builtin_readcyclecounter(); acc_c += t1 - tq; tq = t1; }
        if (tid == 0) { out_idx[j] = old; if (out_cen) { out_cen[3 * j] = c0; out_cen[3 * j + 1] = c1; out_cen[3 * j + 2] = c2; } }
    }
    DBG_STAMP(16);
    if (prof) { g_dbg_stamps[24] = acc_a; g_dbg_stamps[25] = acc_b; g_dbg_stamps[26] = acc_c; g_dbg_stamps[27] = acc_n; }
}

// ------------------------------------------------------------------------------------------------
// a3 + a5 with the FIRST pass of the farthest point sampling (at full-chip parallelism).
// FPS starts at the first candidate in row-major order.  Every wavefront re-derives it from the first
// 64 pixels of the frame; if one of them is a candidate (the usual case: the first pixels are empty and
// empty pixels are candidates) the kernel writes temp = min(1e10, d(pixel, first centre)) instead of
// 1e10 and fills the FPS tile table (FpsLds layout) so the FPS kernel starts at the second centre.
// Otherwise info[b][3] stays 0, the classic temp = 1e10 / -1 is written and the FPS kernel does its own
// first pass.  Same arithmetic either way.  One wavefront per 4x32 tile, TAB_TPW tiles per wavefront.
// ------------------------------------------------------------------------------------------------
#define TAB_TPW (FPS_NH >= 8 ? 1 : 8 / FPS_NH)
template <bool RAW>
__global__ __launch_bounds__(256) void ground_mask_tab_kernel(float *__restrict__ ri, const float *__restrict__ tm,
                                                              const double *__restrict__ ground, double thr, FpsTiling g,
                                                              float *__restrict__ temp, int32_t *__restrict__ info,
                                                              float *__restrict__ tiletab) {
    __shared__ int s_cnt[4], s_nz[4], s_first[4];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = g.N, T = g.T;
    const double a = ground[4 * b], bb = ground[4 * b + 1], c = ground[4 * b + 2], d = ground[4 * b + 3];
    // np.linalg.norm(plane_param[:, :3]) on a (1,1,4) array: all four components (segment_utils.py:47)
    const double div = sqrt(((a * a + bb * bb) + c * c) + d * d);
    // fabs(s + d) / div > thr without the division for all but the borderline pixels: with T = thr * div (one
    // rounding) a numerator above T * (1 + 1e-15) has a correctly rounded quotient above thr, one below
    // T * (1 - 1e-15) a quotient below thr; everything else (incl. NaN, a degenerate plane or threshold) divides.
    const double thr_div = thr * div;
    const bool screen = thr >= 1e-200 && thr_div >= 1e-200 && thr_div <= 1e200;
    const double t_hi = thr_div * (1.0 + 1e-15), t_lo = thr_div * (1.0 - 1e-15);
    // classify one pixel from its loaded range and ray: back-projection, "is a candidate"
    auto classify = [&](float &r, float tx, float ty, float tz, float &x, float &y, float &z) -> bool {
        if (RAW && f2u(r) == RI_EMPTY) r = 0.0f;
        x = r * tx; y = r * ty; z = r * tz;
        const double s = ((double)x * a + (double)y * bb) + (double)z * c;
        const double num = fabs(s + d);
        if (screen && num > t_hi) return true;
        if (screen && num < t_lo) return false;
        return num / div > thr;
    };
    auto load_px = [&](int p, float &r, float &x, float &y, float &z) -> bool {
        r = ri[(int64_t)b * P + p];
        return classify(r, tm[3 * p], tm[3 * p + 1], tm[3 * p + 2], x, y, z);
    };
    bool fast;
    float c0, c1, c2;
    {
        float r, x, y, z;
        const bool cd = load_px(min(lane, P - 1), r, x, y, z) && lane < P;
        const unsigned long long m = __ballot(cd);
        fast = m != 0ull;
        const int f0 = fast ? (int)__ffsll((long long)m) - 1 : 0;
        c0 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(x), f0));
        c1 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(y), f0));
        c2 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(z), f0));
        if (blockIdx.x == 0 && threadIdx.x == 0) info[4 * b + 3] = fast ? 1 : 0;
    }
    float *tab = tiletab + (int64_t)b * FPS_TAB_ROWS * T;
    float *ri_b = ri + (int64_t)b * P, *temp_b = temp + (int64_t)b * P;
    int cnt = 0, nzc = 0, first = P;
    const int t0 = (blockIdx.x * 4 + wave) * TAB_TPW;
    // the loads of all TAB_TPW tiles of this wavefront first (unconditional, clamped): one memory latency, not TAB_TPW
    float pr[TAB_TPW][FPS_NH], ptx[TAB_TPW][FPS_NH], pty[TAB_TPW][FPS_NH], ptz[TAB_TPW][FPS_NH];
    int ppix[TAB_TPW][FPS_NH];
#pragma unroll
    for (int q = 0; q < TAB_TPW; q++)
#pragma unroll
        for (int h = 0; h < FPS_NH; h++) {
            ppix[q][h] = fps_tile_point<true>(g, min(t0 + q, T - 1), h, lane);
            const uint32_t pc = ppix[q][h] >= 0 ? (uint32_t)ppix[q][h] : 0u;  // byte offsets from wave-uniform bases
            pr[q][h] = ld_at(ri_b, pc * 4u);
            const f32x3 ray = ld_at(reinterpret_cast<const f32x3 *>(tm), pc * 12u);
            ptx[q][h] = ray.x; pty[q][h] = ray.y; ptz[q][h] = ray.z;
        }
#pragma unroll
    for (int q = 0; q < TAB_TPW; q++) {
        const int t = t0 + q;
        if (t >= T) break;
        float x[FPS_NH], y[FPS_NH], z[FPS_NH], nt[FPS_NH];
        bool valid[FPS_NH], cand[FPS_NH];
        int pidx[FPS_NH];
#pragma unroll
        for (int h = 0; h < FPS_NH; h++) {
            pidx[h] = ppix[q][h];
            valid[h] = pidx[h] >= 0;
            float r = pr[q][h];
            cand[h] = classify(r, ptx[q][h], pty[q][h], ptz[q][h], x[h], y[h], z[h]) && valid[h];
            const bool nz = valid[h] && r != 0.0f;
            nt[h] = cand[h] ? 1e10f : -1.0f;
            if (fast) {
                const float dx = x[h] - c0, dy = y[h] - c1, dz = z[h] - c2;
                const float dist = (dx * dx + dy * dy) + dz * dz;  // sampling_gpu.cu:64, un-fused
                nt[h] = cand[h] ? fminf(dist, 1e10f) : -1.0f;
            }
            if (valid[h]) {
                if (RAW) st_at(ri_b, (uint32_t)pidx[h] * 4u, r);
                st_at(temp_b, (uint32_t)pidx[h] * 4u, nt[h]);
            }
            const unsigned long long mc = __ballot(cand[h]), mz = __ballot(nz);
            cnt += __popcll(mc);
            nzc += __popcll(mz);
            if (mc) first = min(first, (int)__builtin_amdgcn_readlane(pidx[h], (int)__ffsll((long long)mc) - 1));
        }
        if (fast) {
            float lo[3], hi[3], wt, wx, wy, wz;
            uint32_t widx;
            fps_tile_box(x, y, z, cand, lo, hi);
            fps_tile_argmax(x, y, z, nt, valid, pidx, wt, wx, wy, wz, widx);
            if (lane < FPS_TAB_ROWS) {
                float v = lo[0];
                v = lane == 1 ? lo[1] : v; v = lane == 2 ? lo[2] : v; v = lane == 3 ? hi[0] : v; v = lane == 4 ? hi[1] : v;
                v = lane == 5 ? hi[2] : v; v = lane == 6 ? wt : v; v = lane == 7 ? wx : v; v = lane == 8 ? wy : v;
                v = lane == 9 ? wz : v; v = lane == 10 ? u2f(widx) : v;
                tab[(int64_t)lane * T + t] = v;
            }
        }
    }
    if (lane == 0) { s_cnt[wave] = cnt; s_nz[wave] = nzc; s_first[wave] = first; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tc = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const int tz = s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3];
        const int tf = min(min(s_first[0], s_first[1]), min(s_first[2], s_first[3]));
        if (tc) { atomicAdd(&info[4 * b + 0], tc); atomicMin(&info[4 * b + 1], tf); }
        if (tz) atomicAdd(&info[4 * b + 2], tz);
    }
}
