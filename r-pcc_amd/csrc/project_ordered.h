// project_ordered.h -- a2 for sweeps in SCANNER ORDER (cpp_modules.cpp:427-467; the order of the points: dataset/dataset.py:48-50): no records,
// one launch; included by rpcc_hip.hip behind the pixel / band kernels.  OPT-IN (RPCC_PROJECT_ORDER_PROBE): exact for every input, but on MI355X
// slower than the two record kernels it replaces -- the numbers are at the end of this comment.
//
// A .bin sweep holds the scanner's rings one after the other (KITTI: top ring first), so the rows of the range image that consecutive points
// fall into move slowly through the image: 8192 consecutive points of the example sweep span 9 rows, and the rows the data has left are
// not visited again.  project_pix_kernel + project_band_kernel do not use that -- every point becomes a 6-byte record that a second launch
// reads back (357 MB of the batch's 2.6 GB) -- because the synthetic sweeps of the benchmark are shuffled.
//
// Here ONE 1024-thread workgroup owns a frame and keeps a WINDOW of the image in LDS: ORD_WIN_PX pixels = WR whole rows [lo, lo + WR), row r in
// slot r mod WR.  It walks the frame's points in input order, ORD_CHUNK at a time (8 per thread, the next chunk's loads in flight while this
// one is resolved): screened fast pixel as in the pixel kernel, exact fdlibm sequence for the uncertain ones (queued in LDS, taken by the first
// wavefronts at the head of the NEXT chunk's compute phase), ds_min_u32 on the window for the points whose row is inside.  Points outside stay
// PENDING in their thread's registers; the workgroup then moves the window to where most pending points are (a histogram of their rows), which
// writes the rows that leave to the image -- final values, 0 for empty -- and RE-OPENS a row that had been written before by loading it back.
// So the result never depends on the order of the points: any input gives the bit pattern of the two-kernel path (min over the bit patterns
// of the finite, non-zero depths; a frame with a depth-0 point, or with more uncertain points in a chunk than the queue holds, is flagged for
// the fix-up workgroups of the band kernel as before); order only decides how often the window moves.  When the image is complete one pass
// over it leaves the ground fit's hand-off (candidate counts per chunk, a byte per quad of pixels) exactly as project_band_kernel does.  A probe
// of 16 x 64 points in front (rows of consecutive points close together, their drift through the image close to monotone) says whether the
// frame is worth it: accept[b] = 1 and this kernel projects it, or 0 and the frame takes the records.
//
// Measured (round 6, 256 copies of the example sweep in stored order, 122 k points each; profiles/HISTORY.md): 290 us per launch against
// 151 + 85 us of the pixel and band kernels; with three batches in flight 0.861 against 0.795 ms per step.  The pixel arithmetic is the same
// ~140 VALU instructions per point in both; the pixel kernel runs it with 7 wavefronts per SIMD in free-running 256-thread workgroups, this
// kernel with the 4 per SIMD that one 1024-thread workgroup per CU gives (its 145 KB of LDS exclude a second one) and a barrier-separated
// chunk loop: 47 % of the VALU against 58 %.  The records it saves (12 % of the step's traffic) do not pay for that.  The probe alone, on the
// shuffled benchmark sweeps, costs 5.6 us per launch and 1.3 % of the pipelined step (a 1024-thread / 145 KB workgroup per frame waits for a
// whole free CU): hence opt-in.
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

#define ORD_QCAP 4096          // entries per queue of uncertain points (a power of two; a chunk with more sends its frame to the fix-up pass)
struct OrdShared {
    uint16_t queue[2][ORD_QCAP];        // chunk-relative indices of the points the fast path is not certain about: this chunk's / the previous one's
    uint32_t hist[ORD_MAX_H];           // pending points per row
    uint32_t ret[ORD_MAX_H / 32];       // rows written to the image so far
    int zc[RS_CHUNKS];                  // the ground fit's candidate counts of this frame
    uint32_t qn[2];                     // running totals of the two queues (never reset)
    uint32_t npend, best;
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

// Barrier of the chunk loop: LDS traffic complete (s_waitcnt lgkmcnt(0)), then s_barrier -- NOT __syncthreads(), whose fence also waits for every
// global access in flight (vmcnt(0)): the next chunk's point loads and the stores of the rows that left the window would be waited for at every
// barrier.  (As it turned out the compiler's __syncthreads() does not wait for them either -- without threadgroup-split mode a workgroup-scope fence is
// s_waitcnt lgkmcnt(0) -- so this is the same barrier, spelled out.)  Global data written by one thread and read by another inside this kernel (a row
// that is re-opened, the hand-off pass) is ordered explicitly at those places: s_waitcnt vmcnt(0) in every wavefront, then the barrier.
__device__ __forceinline__ void ord_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
    if (tid == 0) { S.qn[0] = 0u; S.qn[1] = 0u; S.npend = 0u; S.best = 0u; S.sawzero = 0; S.accept = 0; S.lo = 0; }
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
    const UDiv32 by_wq = udiv32_make((uint32_t)Wq);   // (W >= 8: ordered_geometry_ok)
    {
        uint4 *w4 = reinterpret_cast<uint4 *>(ord_win);
        for (uint32_t q = tid; q < ORD_WIN_PX / 4; q += ORD_THREADS) w4[q] = make_uint4(RI_EMPTY, RI_EMPTY, RI_EMPTY, RI_EMPTY);
    }
    // rows [e0, e0 + k) enter the window whose low row was lo_old (k <= WR): the row of the old window in the same slot leaves first -- its final
    // values (0 = empty) go to the image with 16-byte stores nobody waits for -- and the entering row starts empty, or, if it has been written
    // before, with what was written.  (k < 0: nothing enters -- the last write-out of all WR rows)
    auto move_rows = [&](int lo_old, int e0, int k) {
        const bool final_pass = k < 0;
        const int rows = final_pass ? WR : k;
        const uint32_t total = (uint32_t)rows * (uint32_t)Wq;
        const int lom_old = lo_old % WR;
        uint4 *w4 = reinterpret_cast<uint4 *>(ord_win);
        for (uint32_t q = tid; q < total; q += ORD_THREADS) {
            const uint32_t ri_ = udiv32(q, by_wq), quad = q - ri_ * (uint32_t)Wq;
            int lrow, erow, slot;
            if (final_pass) { lrow = lo_old + (int)ri_; erow = -1; const int s = (int)ri_ + lom_old; slot = s >= WR ? s - WR : s; }
            else {
                erow = e0 + (int)ri_; slot = erow % WR;
                int d = slot - lom_old; d = d < 0 ? d + WR : d;
                lrow = lo_old + d;
            }
            const uint32_t widx = (uint32_t)slot * (uint32_t)Wq + quad;
            uint4 v = w4[widx];
            v.x = v.x == RI_EMPTY ? 0u : v.x; v.y = v.y == RI_EMPTY ? 0u : v.y; v.z = v.z == RI_EMPTY ? 0u : v.z; v.w = v.w == RI_EMPTY ? 0u : v.w;
            *reinterpret_cast<uint4 *>(img + (uint32_t)lrow * (uint32_t)W + 4u * quad) = v;
            uint4 e = make_uint4(RI_EMPTY, RI_EMPTY, RI_EMPTY, RI_EMPTY);
            if (erow >= 0 && ((S.ret[erow >> 5] >> (erow & 31)) & 1u)) {
                // (written by this workgroup's own stores, maybe after an earlier load of the line: read past the CU's L1)
                const uint32_t *ep = img + (uint32_t)erow * (uint32_t)W + 4u * quad;
                const uint32_t o0_ = __hip_atomic_load(ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), o1_ = __hip_atomic_load(ep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               o2_ = __hip_atomic_load(ep + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), o3_ = __hip_atomic_load(ep + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                e.x = o0_ ? o0_ : RI_EMPTY; e.y = o1_ ? o1_ : RI_EMPTY; e.z = o2_ ? o2_ : RI_EMPTY; e.w = o3_ ? o3_ : RI_EMPTY;
            }
            w4[widx] = e;
        }
    };
    __syncthreads();

    TRACE_ORD_DECLS();
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
    int lo = 0, lom = 0;
    uint32_t epix = ORD_NONE, edep = 0u;   // a point of the exact sequence that fell outside the window
    // the exact sequence for queue entry j (j < cnt) of the chunk that started at point il0q and pushed from running total qs0 on into queue qi
    auto exact_entry = [&](int qi, uint32_t qs0, uint32_t cnt, uint32_t j, uint32_t il0q) {
        if (j < cnt) {
            const uint32_t i = il0q + (uint32_t)S.queue[qi][(qs0 + j) & (ORD_QCAP - 1)];
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
    };
    uint32_t ppix[ORD_PPT], pdep[ORD_PPT];   // pending: row << 24 | pixel (P < 2^24; ORD_NONE: none) and depth bits
#pragma unroll
    for (int u = 0; u < ORD_PPT; u++) { ppix[u] = ORD_NONE; pdep[u] = 0u; }
    // every point a thread holds that the window now covers goes in; the others are counted per row (whole wavefronts call)
    auto apply_pending = [&]() {
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
        bool pend = epix != ORD_NONE;
        const int row = (int)(epix >> 24);
        const int d = row - lo;
        if (pend && (uint32_t)d < (uint32_t)WR) {
            int s = d + lom; s = s >= WR ? s - WR : s;
            atomicMin(&ord_win[s * W + (int)((epix & 0xFFFFFFu) - (uint32_t)row * (uint32_t)W)], edep);
            epix = ORD_NONE; pend = false;
        }
        ord_note_pending(pend, row, S);
    };
    // moves the window until no thread holds a pending point (called by all threads between barriers; three barriers per move)
    auto resolve = [&]() {
        while (S.npend != 0u) {   // (workgroup-uniform: read between barriers)
            if (tid <= H - WR) {
                uint32_t cover = 0u;
                for (int j = 0; j < WR; j++) cover += S.hist[tid + j];
                const uint32_t dist = (uint32_t)abs(tid - lo);
                atomicMax(&S.best, (cover << 16) | ((255u - dist) << 8) | (uint32_t)tid);   // most pending points, then the shortest move
            }
            TRACE_ORD_COUNT(14);
            ord_barrier();
            const int nlo = (int)(S.best & 255u);
            // (the best position covers a pending point and none lies in the current window: nlo != lo)
            int e0, k;
            if (nlo < lo) { e0 = nlo; k = min(lo - nlo, WR); } else { k = min(nlo - lo, WR); e0 = nlo + WR - k; }
            if (tid < ORD_MAX_H) S.hist[tid] = 0u;
            if (tid == 0) { S.npend = 0u; S.lo = nlo; }
            // (a row that comes back is read from the image: the stores of the move that wrote it must have landed -- rare, the full barrier)
            bool back = false;
            for (int r = e0; r < e0 + k; r++) back = back || ((S.ret[r >> 5] >> (r & 31)) & 1u);
            if (back) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // (every wavefront's own stores acknowledged, then the barrier)
            move_rows(lo, e0, k);
            ord_barrier();
            if (tid == 0) {
                for (int r = lo; r < lo + WR; r++)   // the rows that left are in the image now
                    if (r < nlo || r >= nlo + WR) S.ret[r >> 5] |= 1u << (r & 31);
                S.best = 0u;
            }
            lo = nlo; lom = lo % WR;
            apply_pending();
            ord_barrier();
        }
    };
    load_chunk(0u);
    // The points the fast path is not certain about (about 3 %) are queued and take the exact sequence ONE CHUNK LATER, at the head of the next
    // chunk's compute phase (whole wavefronts, the first ones, while the others already compute): two queues used in turn, entries at
    // (running total) mod ORD_QCAP, totals never reset.  Iteration nchunks only drains.
    uint32_t qs0 = 0u, qs1 = 0u;                 // running totals of the two queues when their last chunk began
    uint32_t prev_n = 0u, prev_s = 0u, prev_il0 = 0u;
    for (uint32_t c = 0; c <= nchunks; c++) {
        TRACE_ORD_PHASE(0);
        const uint32_t il0 = c * (uint32_t)ORD_CHUNK;
        const int qi = (int)(c & 1u), pqi = qi ^ 1;
        lo = __builtin_amdgcn_readfirstlane(S.lo);
        lom = lo % WR;
        if ((uint32_t)(tid & ~63) < min(prev_n, (uint32_t)ORD_THREADS)) exact_entry(pqi, prev_s, prev_n, (uint32_t)tid, prev_il0);
        TRACE_ORD_PHASE(4);
        if (c < nchunks) {
            const uint32_t room = min((uint32_t)ORD_CHUNK, n - il0);
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
                    if (lane == leader) q0 = atomicAdd(&S.qn[qi], (uint32_t)__popcll(sm));
                    q0 = (uint32_t)__builtin_amdgcn_readlane((int)q0, leader);
                    if (slow) S.queue[qi][(q0 + __popcll(sm & lt)) & (ORD_QCAP - 1)] = (uint16_t)ci;
                }
            }
            TRACE_ORD_PHASE(1);
            if (c + 1u < nchunks) load_chunk(c + 1u);   // in flight while this chunk is resolved
        }
#pragma unroll
        for (int u = 0; u < ORD_PPT; u++) ord_note_pending(ppix[u] != ORD_NONE, (int)(ppix[u] >> 24), S);
        ord_note_pending(epix != ORD_NONE, (int)(epix >> 24), S);
        TRACE_ORD_PHASE(2);
        ord_barrier();
        TRACE_ORD_PHASE(3);
        // this chunk's queue: entries [qs, qend) of queue qi
        const uint32_t qend = S.qn[qi], qs = qi ? qs1 : qs0;
        uint32_t n_c = qend - qs;
        if (n_c > (uint32_t)ORD_QCAP) {   // more uncertain points than the queue holds (points on pixel borders throughout): the exact input-order pass
            if (tid == 0) { A.flags[b] = mark; A.flags[B] = mark; S.sawzero = 1; }
            n_c = 0u;
        }
        // the rest of the previous chunk's queue (more than 1024 entries: rare), 1024 at a time, the window following after every round
        for (uint32_t qpos = ORD_THREADS;; qpos += ORD_THREADS) {
            resolve();
            if (qpos >= prev_n) break;
            if (qpos + (uint32_t)(tid & ~63) < prev_n) {
                exact_entry(pqi, prev_s, prev_n, qpos + (uint32_t)tid, prev_il0);
                ord_note_pending(epix != ORD_NONE, (int)(epix >> 24), S);
            }
            ord_barrier();
        }
        TRACE_ORD_PHASE(6);
        TRACE_ORD_COUNT(15);
        prev_n = n_c; prev_s = qs; prev_il0 = il0;
        if (qi) qs1 = qend; else qs0 = qend;
    }
    // ---- the rows still in the window, then the rows no point ever asked for ----
    TRACE_ORD_PHASE(7);
    move_rows(lo, 0, -1);
    for (int r = 0; r < H; r++) {   // (workgroup-uniform)
        const bool done = (r >= lo && r < lo + WR) || ((S.ret[r >> 5] >> (r & 31)) & 1u);
        if (done) continue;
        for (int quad = tid; quad < Wq; quad += ORD_THREADS)
            *reinterpret_cast<uint4 *>(img + (uint32_t)r * (uint32_t)W + 4u * (uint32_t)quad) = make_uint4(0u, 0u, 0u, 0u);
    }
    TRACE_ORD_PHASE(8);
    if (!want_cnt) { TRACE_ORD_END(); return; }
    // ---- the hand-off to the ground fit, as project_band_kernel leaves it: per chunk of rs_chunk pixels the number of pixels with z = r * tz < zthr,
    // and which ones, a byte per quad.  One pass over the finished image (this CU's own stores: in its L2) and the ray table.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (a workgroup-scope barrier waits for LDS traffic only: the image stores must have landed)
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (the image is read back: nothing stale from the CU's L1)
    if (!S.sawzero) {   // (a frame with a depth-0 point is projected again by the fix-up workgroup and counts for itself)
        uint8_t *zm = rs_zmask_of(A.zcnt, B) + (int64_t)b * (P >> 2);
        const uint32_t nqd = (uint32_t)P >> 2;
        constexpr int FU = 4;   // quads per thread in flight
        for (uint32_t q0 = tid; q0 - (uint32_t)lane < nqd; q0 += ORD_THREADS * FU) {   // (whole wavefronts stay in the loop: the DPP sums need every lane)
            uint4 v[FU];
            float4 ta[FU], tb[FU], tc[FU];
#pragma unroll
            for (int u = 0; u < FU; u++) {
                const uint32_t q = min(q0 + (uint32_t)(u * ORD_THREADS), nqd - 1u);   // unconditional (clamped) loads
                v[u] = ld_at(reinterpret_cast<const uint4 *>(img), q * 16u);
                ta[u] = ld_at(reinterpret_cast<const float4 *>(A.tm), q * 48u); tb[u] = ld_at(reinterpret_cast<const float4 *>(A.tm), q * 48u + 16u);
                tc[u] = ld_at(reinterpret_cast<const float4 *>(A.tm), q * 48u + 32u);
            }
#pragma unroll
            for (int u = 0; u < FU; u++) {
                const uint32_t q = q0 + (uint32_t)(u * ORD_THREADS);
                const bool act = q < nqd;
                const int t0 = (int)(u2f(v[u].x) * ta[u].z < A.zthr), t1 = (int)(u2f(v[u].y) * tb[u].y < A.zthr), t2 = (int)(u2f(v[u].z) * tc[u].x < A.zthr),
                          t3 = (int)(u2f(v[u].w) * tc[u].w < A.zthr);
                const int cnt = act ? (t0 + t1) + (t2 + t3) : 0;
                if (act && want_bytes) zm[q] = (uint8_t)(t0 | (t1 << 1) | (t2 << 2) | (t3 << 3));
                const uint32_t ch = (4u * min(q, nqd - 1u)) / (uint32_t)A.rs_chunk;
                const uint32_t ch0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch);
                if (__ballot(ch != ch0) == 0ull) {
                    const int tot = (int)dpp_sum_u32((uint32_t)cnt);
                    if (tot && lane == 0) atomicAdd(&S.zc[ch0], tot);
                } else if (cnt) {
                    atomicAdd(&S.zc[ch], cnt);
                }
            }
        }
    }
    __syncthreads();
    if (tid < RS_CHUNKS) A.zcnt[b * (RS_CHUNKS + 1) + tid] = S.zc[tid];
    if (tid == 0) A.zcnt[b * (RS_CHUNKS + 1) + RS_CHUNKS] = S.sawzero ? 0 : (want_bytes ? 3 : 1);
    TRACE_ORD_PHASE(9);
    TRACE_ORD_END();
}
