// project_ordered.h -- a2 for sweeps in SCANNER ORDER (cpp_modules.cpp:427-467; the order of the points: dataset/dataset.py:48-50): no records,
// one launch; included by rpcc_hip.hip behind the pixel / band kernels.
//
// A .bin sweep holds the scanner's rings one after the other (KITTI: top ring first), so the rows of the range image that consecutive points
// fall into move slowly through the image: 8192 consecutive points of the example sweep span 9 rows, and the rows the data has left are
// not visited again.  project_pix_kernel + project_band_kernel do not use that -- every point becomes a 6-byte record that a second launch
// reads back (357 MB of the batch's 2.6 GB) -- because the synthetic sweeps of the benchmark are shuffled.
//
// Here ONE 1024-thread workgroup owns a frame and keeps a WINDOW of the image in LDS: ORD_WIN_PX pixels = WR whole rows [lo, lo + WR), row r in
// slot r mod WR.  It walks the frame's points in input order, ORD_CHUNK at a time (8 per thread, the next chunk's loads in flight while this
// one is resolved): screened fast pixel as in the pixel kernel, exact fdlibm sequence for the uncertain ones (queued in LDS, drained every
// chunk), ds_min_u32 on the window for the points whose row is inside.  Points outside stay PENDING in their thread's registers; the
// workgroup then moves the window to where most pending points are (a histogram of their rows), which writes the rows that leave to the
// image -- final values: 0 for empty, plus the ground fit's candidate counts and bytes exactly as project_band_kernel leaves them -- and
// RE-OPENS a row that had been written before by loading it back (its candidate counts are taken back).  So the result never depends on
// the order of the points: any input gives the bit pattern of the two-kernel path (min over the bit patterns of the finite, non-zero
// depths; a frame with a depth-0 point is flagged for the fix-up workgroups of the band kernel as before); order only decides how often the
// window moves.  A probe of 16 x 64 points in front (rows of consecutive points close together, their drift through the image close to
// monotone) says whether the frame is worth it: accept[b] = 1 and this kernel projects it, or 0 and the frame takes the records.
#pragma once

#define ORD_THREADS 1024
#define ORD_PPT 8
#define ORD_CHUNK (ORD_THREADS * ORD_PPT)
#define ORD_WIN_PX 32768      // 128 KiB of LDS
#define ORD_MAX_H 128         // rows of an image this kernel takes (the histogram of pending rows)
#define ORD_MIN_POINTS 4096   // smaller frames are not worth a 1024-thread workgroup
#define ORD_NONE 0xFFFFFFFFu
#define ORD_MODE_PROBE 0
#define ORD_MODE_FORCE 1      // test hook: every frame with a point is accepted (the window then thrashes on shuffled input: slow, same result)

struct OrdShared {
    uint16_t queue[ORD_CHUNK];          // chunk-relative indices of the points the fast path is not certain about
    uint32_t hist[ORD_MAX_H];           // pending points per row
    uint32_t ret[ORD_MAX_H / 32];       // rows written to the image so far
    int zc[RS_CHUNKS];                  // the ground fit's candidate counts of this frame
    uint32_t qn, npend, best;
    int lo, sawzero, accept;
    int pmin[ORD_THREADS / 64], pmax[ORD_THREADS / 64];
};

static inline bool ordered_geometry_ok(const rpcc_geom g) {
    const long long P = (long long)g.H * g.W;
    return g.H >= 2 && g.H <= ORD_MAX_H && g.W >= 8 && (g.W & 3) == 0 && g.W <= ORD_WIN_PX / 2 && P < (1ll << 24);
}

struct OrdArgs {
    const float *xyz;
    const int64_t *offs;
    int64_t base;
    int B;
    rpcc_geom g;
    PixFastCfg cfg;
    uint32_t *ri;
    int32_t *flags;
    const int32_t *epoch;
    int32_t *accept;
    int mode;
    const float *tm;      // [P,3] rays: z of a pixel for the candidate test (the planar copy does not exist yet when this kernel runs)
    float zthr;
    int rs_chunk;
    int32_t *zcnt;        // nullptr: no hand-off to the ground fit
};

// every lane of the wavefront calls; pend: the lane holds a point outside the window, in `row`
__device__ __forceinline__ void ord_note_pending(bool pend, int row, OrdShared &S) {
    unsigned long long m = __ballot(pend);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    const uint32_t total = (uint32_t)__popcll(m);
    const int first = (int)__ffsll((long long)m) - 1;
    while (m) {
        const int l = (int)__ffsll((long long)m) - 1;
        const int r = __builtin_amdgcn_readlane(row, l);
        const unsigned long long mm = __ballot(pend && row == r);
        if (lane == l) atomicAdd(&S.hist[r], (uint32_t)__popcll(mm));
        m &= ~mm;
    }
    if (lane == first) atomicAdd(&S.npend, total);
}

template <int PS>
__global__ __launch_bounds__(ORD_THREADS) void project_ordered_kernel(const OrdArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t ord_win[];   // [ORD_WIN_PX]
    __shared__ OrdShared S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, B = A.B;
    const rpcc_geom g = A.g;
    const int H = g.H, W = g.W, P = H * W, Wq = W >> 2;
    const int WR = min(H, ORD_WIN_PX / W);
    const int64_t o0 = A.offs[b] - A.base, o1 = A.offs[b + 1] - A.base;
    const int64_t n64 = o1 - o0;
    const int mark = flag_mark(A.epoch);
    const float *fx = A.xyz + PS * (A.base + o0);   // the frame's points
    // this frame's share of the hand-off: nothing valid yet (the band kernel writes it for a frame this kernel leaves alone)
    if (A.zcnt && tid <= RS_CHUNKS) A.zcnt[b * (RS_CHUNKS + 1) + tid] = 0;
    if (tid < ORD_MAX_H) S.hist[tid] = 0u;
    if (tid < ORD_MAX_H / 32) S.ret[tid] = 0u;
    if (tid < RS_CHUNKS) S.zc[tid] = 0;
    if (tid == 0) { S.qn = 0u; S.npend = 0u; S.best = 0u; S.sawzero = 0; S.accept = 0; S.lo = 0; }
    // ---- probe: 16 runs of 64 consecutive points spread over the frame ----
    const bool sized = n64 >= (A.mode == ORD_MODE_FORCE ? 1 : ORD_MIN_POINTS) && n64 < ((int64_t)1 << 28);   // (32-bit byte offsets into the frame)
    const uint32_t n = sized ? (uint32_t)n64 : 0u;
    if (sized) {
        const uint32_t start = (uint32_t)(((uint64_t)(n > 64u ? n - 64u : 0u) * (uint32_t)wave) / (ORD_THREADS / 64 - 1));
        const uint32_t i = min(start + (uint32_t)lane, n - 1u);
        float x, y, z;
        if (PS == 4) { const float4 p4 = ld_at(reinterpret_cast<const float4 *>(fx), i * 16u); x = p4.x; y = p4.y; z = p4.z; }
        else { const f32x3 p3 = ld_at(reinterpret_cast<const f32x3 *>(fx), i * 12u); x = p3.x; y = p3.y; z = p3.z; }
        int pix, row, col;
        const bool ok = project_point_fast(x, y, z, g, A.cfg, pix, nullptr, nullptr, &row, &col);
        // (a point the fast path does not vouch for -- near a pixel border, a special value -- still has a usable row, or is left out)
        const bool use = ok || (x == x && y == y && z == z && row >= 0 && row < H);
        int rmin = use ? row : H, rmax = use ? row : -1;
        rmin = (int)dpp_min_u32((uint32_t)rmin);
        rmax = (int)dpp_max_u32((uint32_t)(rmax + 1)) - 1;
        if (lane == 0) { S.pmin[wave] = rmin; S.pmax[wave] = rmax; }
    }
    __syncthreads();
    if (tid == 0 && sized) {
        int tv = 0, prev = -1, first = -1, spanmax = 0, seen = 0;
        for (int w = 0; w < ORD_THREADS / 64; w++) {
            if (S.pmax[w] < 0) continue;   // a run without a usable point
            const int c = S.pmin[w] + S.pmax[w];
            spanmax = max(spanmax, S.pmax[w] - S.pmin[w]);
            if (prev >= 0) tv += abs(c - prev);
            if (first < 0) first = c;
            prev = c; seen++;
        }
        // rows of consecutive points close together (a run fits half the window) and the runs' drift through the image close to monotone
        // (a sweep stored ring by ring: total variation = its net travel, 2 H in these units; rings in random order: 5 H and more)
        const bool good = seen >= 8 && (WR >= H || (2 * spanmax <= WR && tv <= 4 * H));
        const int acc = (A.mode == ORD_MODE_FORCE) ? 1 : (good ? 1 : 0);
        S.accept = acc;
        S.lo = first < 0 ? 0 : min(max(first / 2 - WR / 2, 0), H - WR);
    }
    __syncthreads();
    const int accepted = S.accept;
    if (tid == 0) A.accept[b] = accepted;
    if (!accepted) return;

    // ---- the frame ----
    uint32_t *img = A.ri + (int64_t)b * P;
    const bool want_cnt = A.zcnt != nullptr;
    const bool want_bytes = want_cnt && (P & 63) == 0;
    uint8_t *zm = want_cnt ? rs_zmask_of(A.zcnt, B) + (int64_t)b * (P >> 2) : nullptr;
    const UDiv32 by_wq = udiv32_make((uint32_t)Wq);   // (W >= 8: ordered_geometry_ok)
    {
        uint4 *w4 = reinterpret_cast<uint4 *>(ord_win);
        for (uint32_t q = tid; q < ORD_WIN_PX / 4; q += ORD_THREADS) w4[q] = make_uint4(RI_EMPTY, RI_EMPTY, RI_EMPTY, RI_EMPTY);
    }
    // the four candidate tests of a quad of pixels (ray z from the [P,3] table: 48 bytes = three 16-byte loads)
    auto ztests = [&](uint32_t p4, const uint4 v, int &t0, int &t1, int &t2, int &t3) {
        const float4 a = ld_at(reinterpret_cast<const float4 *>(A.tm), p4 * 12u), bq = ld_at(reinterpret_cast<const float4 *>(A.tm), p4 * 12u + 16u),
                     c = ld_at(reinterpret_cast<const float4 *>(A.tm), p4 * 12u + 32u);
        t0 = (int)(u2f(v.x) * a.z < A.zthr); t1 = (int)(u2f(v.y) * bq.y < A.zthr); t2 = (int)(u2f(v.z) * c.x < A.zthr); t3 = (int)(u2f(v.w) * c.w < A.zthr);
    };
    auto zcount = [&](uint32_t p4, int c, bool act) {   // (whole wavefronts call) c candidates of the quad at pixel p4 -> the chunk's counter
        const uint32_t ch = p4 / (uint32_t)A.rs_chunk;
        const uint32_t ch0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch);
        if (__ballot(act && ch != ch0) == 0ull) {
            const int tot = (int)dpp_sum_u32((uint32_t)(act ? c : 0));
            if (tot && lane == 0) atomicAdd(&S.zc[ch0], tot);
        } else if (act && c) {
            atomicAdd(&S.zc[ch], c);
        }
    };
    // rows [e0, e0 + k) enter the window whose low row was lo_old (k <= WR): the row of the old window in the same slot leaves first
    // (k < 0: nothing enters -- the last write-out of all WR rows)
    auto move_rows = [&](int lo_old, int e0, int k) {
        const bool final_pass = k < 0;
        const int rows = final_pass ? WR : k;
        const uint32_t total = (uint32_t)rows * (uint32_t)Wq;
        const int lom_old = lo_old % WR;
        uint4 *w4 = reinterpret_cast<uint4 *>(ord_win);
        for (uint32_t q0 = tid; q0 - (uint32_t)lane < total; q0 += ORD_THREADS) {
            const bool act = q0 < total;
            const uint32_t q = act ? q0 : total - 1u;
            const uint32_t ri_ = udiv32(q, by_wq), quad = q - ri_ * (uint32_t)Wq;
            int lrow, erow, slot;
            if (final_pass) { lrow = lo_old + (int)ri_; erow = -1; int s = (int)ri_ + lom_old; slot = s >= WR ? s - WR : s; }
            else {
                erow = e0 + (int)ri_; slot = erow % WR;
                int d = slot - lom_old; d = d < 0 ? d + WR : d;
                lrow = lo_old + d;
            }
            const uint32_t widx = (uint32_t)slot * (uint32_t)Wq + quad;
            // the leaving row: final values to the image, candidate counts and bytes as project_band_kernel's write-out
            uint4 v = w4[widx];
            v.x = v.x == RI_EMPTY ? 0u : v.x; v.y = v.y == RI_EMPTY ? 0u : v.y; v.z = v.z == RI_EMPTY ? 0u : v.z; v.w = v.w == RI_EMPTY ? 0u : v.w;
            const uint32_t lp = (uint32_t)lrow * (uint32_t)W + 4u * quad;
            if (act) *reinterpret_cast<uint4 *>(img + lp) = v;
            if (want_cnt) {
                int t0, t1, t2, t3;
                ztests(lp, v, t0, t1, t2, t3);
                if (act && want_bytes) zm[lp >> 2] = (uint8_t)(t0 | (t1 << 1) | (t2 << 2) | (t3 << 3));
                zcount(lp, (t0 + t1) + (t2 + t3), act);
            }
            // the entering row: empty, or what was written before (its candidates are counted again when it leaves)
            uint4 e = make_uint4(RI_EMPTY, RI_EMPTY, RI_EMPTY, RI_EMPTY);
            const bool back = erow >= 0 && ((S.ret[erow >> 5] >> (erow & 31)) & 1u);
            if (__ballot(back) != 0ull) {
                const uint32_t ep = (uint32_t)(back ? erow : lrow) * (uint32_t)W + 4u * quad;
                uint4 o = make_uint4(0u, 0u, 0u, 0u);
                if (act && back) {   // (written by this workgroup's own stores, maybe after an earlier load of the line: read past the CU's L1)
                    o.x = __hip_atomic_load(img + ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    o.y = __hip_atomic_load(img + ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    o.z = __hip_atomic_load(img + ep + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    o.w = __hip_atomic_load(img + ep + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (want_cnt) {
                    int t0, t1, t2, t3;
                    ztests(ep, o, t0, t1, t2, t3);
                    zcount(ep, -((t0 + t1) + (t2 + t3)), act && back);
                }
                if (back) { e.x = o.x ? o.x : RI_EMPTY; e.y = o.y ? o.y : RI_EMPTY; e.z = o.z ? o.z : RI_EMPTY; e.w = o.w ? o.w : RI_EMPTY; }
            }
            if (act) w4[widx] = e;
        }
    };
    __syncthreads();

    const uint32_t nchunks = (n + ORD_CHUNK - 1u) / ORD_CHUNK;
    float x[ORD_PPT], y[ORD_PPT], z[ORD_PPT];
    auto load_chunk = [&](uint32_t c) {
        const uint32_t il0 = c * (uint32_t)ORD_CHUNK;
#pragma unroll
        for (int u = 0; u < ORD_PPT; u++) {
            const uint32_t i = min(il0 + (uint32_t)(u * ORD_THREADS) + (uint32_t)tid, n - 1u);   // unconditional (clamped) loads
            if (PS == 4) { const float4 p4 = ld_at(reinterpret_cast<const float4 *>(fx), i * 16u); x[u] = p4.x; y[u] = p4.y; z[u] = p4.z; }
            else { const f32x3 p3 = ld_at(reinterpret_cast<const f32x3 *>(fx), i * 12u); x[u] = p3.x; y[u] = p3.y; z[u] = p3.z; }
        }
    };
    load_chunk(0u);
    TRACE_ORD_DECLS();
    for (uint32_t c = 0; c < nchunks; c++) {
        TRACE_ORD_PHASE(0);
        const uint32_t il0 = c * (uint32_t)ORD_CHUNK;
        const uint32_t room = min((uint32_t)ORD_CHUNK, n - il0);
        int lo = __builtin_amdgcn_readfirstlane(S.lo);
        int lom = lo % WR;
        uint32_t ppix[ORD_PPT], pdep[ORD_PPT];   // pending: row << 24 | pixel (P < 2^24; ORD_NONE: none) and depth bits
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int u = 0; u < ORD_PPT; u++) {
            const uint32_t ci = (uint32_t)(u * ORD_THREADS) + (uint32_t)tid;
            const bool in = ci < room;
            int pix, row, col;
            const bool fast = project_point_fast(x[u], y[u], z[u], g, A.cfg, pix, nullptr, nullptr, &row, &col) && A.cfg.on && in;
            const float depth = sqrt_rn_normal(x[u] * x[u] + y[u] * y[u] + z[u] * z[u]);   // :446 (== sqrtf when `fast`)
            const int d = row - lo;
            const bool inwin = (uint32_t)d < (uint32_t)WR;
            int s = d + lom; s = s >= WR ? s - WR : s;
            if (fast && inwin) atomicMin(&ord_win[s * W + col], f2u(depth));
            ppix[u] = (fast && !inwin) ? ((uint32_t)row << 24) | (uint32_t)pix : ORD_NONE;
            pdep[u] = f2u(depth);
            const bool slow = in && !fast;
            const unsigned long long sm = __ballot(slow);
            if (sm) {
                const int leader = (int)__ffsll((long long)sm) - 1;
                uint32_t q0 = 0u;
                if (lane == leader) q0 = atomicAdd(&S.qn, (uint32_t)__popcll(sm));
                q0 = (uint32_t)__builtin_amdgcn_readlane((int)q0, leader);
                if (slow) S.queue[q0 + __popcll(sm & lt)] = (uint16_t)ci;
            }
        }
        TRACE_ORD_PHASE(1);
        if (c + 1u < nchunks) load_chunk(c + 1u);   // in flight while this chunk is resolved
#pragma unroll
        for (int u = 0; u < ORD_PPT; u++) {
            ord_note_pending(ppix[u] != ORD_NONE, (int)(ppix[u] >> 24), S);
        }
        TRACE_ORD_PHASE(2);
        __syncthreads();
        TRACE_ORD_PHASE(3);
        const uint32_t nq = S.qn;
        uint32_t qpos = 0u;
        do {
            // the exact sequence for up to 1024 queued points (whole wavefronts, the first ones)
            uint32_t epix = ORD_NONE, edep = 0u;
            if (qpos + (uint32_t)(tid & ~63) < nq) {
                const bool have = qpos + (uint32_t)tid < nq;
                if (have) {
                    const uint32_t i = il0 + (uint32_t)S.queue[qpos + (uint32_t)tid];
                    float ex, ey, ez;
                    if (PS == 4) { const float4 p4 = ld_at(reinterpret_cast<const float4 *>(fx), i * 16u); ex = p4.x; ey = p4.y; ez = p4.z; }
                    else { const f32x3 p3 = ld_at(reinterpret_cast<const f32x3 *>(fx), i * 12u); ex = p3.x; ey = p3.y; ez = p3.z; }
                    const RowCol rc = project_point(ex, ey, ez, g);
                    if (fabsf(rc.depth) <= 3.402823466e+38f) {
                        if (rc.depth == 0.0f) {   // resets its pixel in the reference's loop: the frame is redone in input order (project_fixup_frame)
                            A.flags[b] = mark; A.flags[B] = mark; S.sawzero = 1;
                        } else {
                            const int row = rc.pix / W, col = rc.pix - row * W;
                            const int d = row - lo;
                            if ((uint32_t)d < (uint32_t)WR) { int s = d + lom; s = s >= WR ? s - WR : s; atomicMin(&ord_win[s * W + col], f2u(rc.depth)); }
                            else { epix = ((uint32_t)row << 24) | (uint32_t)rc.pix; edep = f2u(rc.depth); }
                        }
                    }
                }
                ord_note_pending(epix != ORD_NONE, (int)(epix >> 24), S);
            }
            qpos += ORD_THREADS;
            TRACE_ORD_PHASE(4);
            __syncthreads();
            TRACE_ORD_PHASE(5);
            // move the window until nothing is pending
            while (S.npend != 0u) {   // (workgroup-uniform: read between barriers)
                if (tid <= H - WR) {
                    uint32_t cover = 0u;
                    for (int j = 0; j < WR; j++) cover += S.hist[tid + j];
                    const uint32_t dist = (uint32_t)abs(tid - lo);
                    atomicMax(&S.best, (cover << 16) | ((255u - dist) << 8) | (uint32_t)tid);   // most pending points, then the shortest move
                }
                TRACE_ORD_COUNT(14);
                __syncthreads();
                const int nlo = (int)(S.best & 255u);
                // (cover > 0 for the best position, and no pending point lies in the current window: nlo != lo)
                int e0, k;
                if (nlo < lo) { e0 = nlo; k = min(lo - nlo, WR); } else { k = min(nlo - lo, WR); e0 = nlo + WR - k; }
                move_rows(lo, e0, k);
                __syncthreads();
                if (tid < ORD_MAX_H) S.hist[tid] = 0u;
                if (tid == 0) {
                    // the rows that left are in the image now
                    for (int r = lo; r < lo + WR; r++)
                        if (r < nlo || r >= nlo + WR) S.ret[r >> 5] |= 1u << (r & 31);
                    S.npend = 0u; S.best = 0u; S.lo = nlo;
                }
                lo = nlo; lom = lo % WR;
                __syncthreads();
#pragma unroll
                for (int u = 0; u < ORD_PPT; u++) {
                    bool pend = ppix[u] != ORD_NONE;
                    const int row = (int)(ppix[u] >> 24);
                    const int d = row - lo;
                    if (pend && (uint32_t)d < (uint32_t)WR) {
                        int s = d + lom; s = s >= WR ? s - WR : s;
                        atomicMin(&ord_win[s * W + (int)((ppix[u] & 0xFFFFFFu) - (uint32_t)row * (uint32_t)W)], pdep[u]);
                        ppix[u] = ORD_NONE; pend = false;
                    }
                    ord_note_pending(pend, row, S);
                }
                {
                    bool pend = epix != ORD_NONE;
                    const int row = (int)(epix >> 24);
                    const int d = row - lo;
                    if (pend && (uint32_t)d < (uint32_t)WR) {
                        int s = d + lom; s = s >= WR ? s - WR : s;
                        atomicMin(&ord_win[s * W + (int)((epix & 0xFFFFFFu) - (uint32_t)row * (uint32_t)W)], edep);
                        epix = ORD_NONE; pend = false;
                    }
                    ord_note_pending(pend, row, S);
                }
                __syncthreads();
            }
            TRACE_ORD_PHASE(6);
        } while (qpos < nq);
        TRACE_ORD_COUNT(15);
        if (tid == 0) S.qn = 0u;
        // (the next chunk's queue pushes come after its compute phase started; a barrier lies between: the one at the top of the drain
        // loop of THIS chunk was the last read of qn)
        __syncthreads();
    }
    // ---- the rows still in the window, then the rows no point ever asked for ----
    TRACE_ORD_PHASE(7);
    const int lo_end = __builtin_amdgcn_readfirstlane(S.lo);
    move_rows(lo_end, 0, -1);
    __syncthreads();
    for (int r = 0; r < H; r++) {   // (workgroup-uniform)
        const bool done = (r >= lo_end && r < lo_end + WR) || ((S.ret[r >> 5] >> (r & 31)) & 1u);
        if (done) continue;
        for (int quad = tid; quad < Wq; quad += ORD_THREADS) {
            const uint32_t lp = (uint32_t)r * (uint32_t)W + 4u * (uint32_t)quad;
            *reinterpret_cast<uint4 *>(img + lp) = make_uint4(0u, 0u, 0u, 0u);
            if (want_bytes) zm[lp >> 2] = 0;
        }
    }
    TRACE_ORD_PHASE(8);
    TRACE_ORD_END();
    if (want_cnt) {
        if (tid < RS_CHUNKS) A.zcnt[b * (RS_CHUNKS + 1) + tid] = S.zc[tid];
        // (a frame with a depth-0 point is projected again by the fix-up workgroup and counts for itself)
        if (tid == 0) A.zcnt[b * (RS_CHUNKS + 1) + RS_CHUNKS] = S.sawzero ? 0 : (want_bytes ? 3 : 1);
    }
}
