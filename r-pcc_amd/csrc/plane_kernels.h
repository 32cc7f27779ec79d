// plane_kernels.h -- a9: per-segment plane model  (utils/segment_utils.py:188-216, plane_angle_validation
// :84-93); included by rpcc_hip.hip.
//
// Per label k >= 2 of a frame (pixels in row-major order):
//   n == 0            -> [0,0,0,NaN]                      (numpy mean of an empty array)
//   n < 30            -> [0,0,0, fp32 pairwise mean]       (segment_utils.py:203-204)
//   else RANSAC(ransac_n = 4, 10 iterations, 0.1 m)        (:207-209; Open3D in the reference, the build's
//                                                           seeded specification here -- DESIGN.md "RANSAC")
//        angle validation (:84-93): reject when max_px arccos(|n.t| / |n| * |t|) > threshold
//        rejected     -> [0,0,0, fp32 pairwise mean]       (:216)
// The mean is NumPy's fp32 reduction restated (8192-element blocks added sequentially, each block summed
// pairwise with 8-accumulator leaves of <= 128 elements) because its rounding
// differs from the double accumulation of the point model.
#pragma once

// label-ordered pixel list: order[b][pos] = pixel index, labels ascending, row-major inside a label -- the same positions
// the quantiser's ordered scatter uses.  Labels 0 (ground: its model is the ground plane) and 1 (empty) have no plane fit:
// their positions are left unwritten (about half of a frame's pixels).
// pts4 (optional): the same list as points (x, y, z, r) so that the plane model streams a label's points instead of
// gathering them through the pixel index.
template <class L = uint8_t>
__device__ __forceinline__ void label_order_body(const L *__restrict__ seg, const uint32_t *__restrict__ hist,
                                                 int P, int M, int KP, int T, uint32_t *__restrict__ order,
                                                 const float *__restrict__ ri, const float *__restrict__ tm,
                                                 float4 *__restrict__ pts4, const int b, const int t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *segcnt = reinterpret_cast<uint32_t *>(smem_raw);  // [16][KP+1]
    const int SEGP = KP + 1;
    uint32_t *soff = segcnt + 16 * SEGP;                        // [KP] this tile's offsets per label
    const int K = M + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-frame bases + byte offsets; all loads of the tile first (unconditional, clamped)
    seg += (int64_t)b * P;
    order += (int64_t)b * P;
    if (pts4) { ri += (int64_t)b * P; pts4 += (int64_t)b * P; }
    int lraw[4];
    float rr[4];
    f32x3 ray[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t p = (uint32_t)min(t * TILE + j * 256 + (int)threadIdx.x, P - 1);
        lraw[j] = ld_at(seg, p * (uint32_t)sizeof(L));
        if (pts4) { rr[j] = ld_at(ri, p * 4u); ray[j] = ld_at(reinterpret_cast<const f32x3 *>(tm), p * 12u); }
    }
    for (int i = threadIdx.x; i < K; i += 256) soff[i] = hist[((int64_t)b * T + t) * KP + i];
    for (int i = threadIdx.x; i < 16 * SEGP; i += 256) segcnt[i] = 0u;
    __syncthreads();
    int lab[4], rank[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = t * TILE + j * 256 + threadIdx.x;
        lab[j] = (p < P && lraw[j] > 1) ? lraw[j] : -1;
        rank[j] = 0;
        int todo = lab[j];
        while (true) {
            const unsigned long long pending = __ballot(todo >= 0);
            if (!pending) break;
            const int leader = (int)__ffsll((long long)pending) - 1;
            const int cur = __builtin_amdgcn_readlane(todo, leader);
            const unsigned long long same = __ballot(todo == cur);
            if (todo == cur) {
                rank[j] = __popcll(same & ((1ull << lane) - 1ull));
                if (lane == leader) segcnt[(j * 4 + wave) * SEGP + cur] = (uint32_t)__popcll(same);
                todo = -1;
            }
        }
    }
    __syncthreads();
    segment_prefix(segcnt, SEGP, soff, K);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (lab[j] >= 0) {
            const int p = t * TILE + j * 256 + threadIdx.x;
            const uint32_t o = segcnt[(j * 4 + wave) * SEGP + lab[j]] + (uint32_t)rank[j];
            order[o] = (uint32_t)p;
            if (pts4) pts4[o] = make_float4(rr[j] * ray[j].x, rr[j] * ray[j].y, rr[j] * ray[j].z, rr[j]);  // transformer.py:94-101
        }
}
template <class L = uint8_t>
__global__ __launch_bounds__(256) void label_order_kernel(const L *__restrict__ seg, const uint32_t *__restrict__ hist,
                                                          int P, int M, int KP, int T, uint32_t *__restrict__ order,
                                                          const float *__restrict__ ri, const float *__restrict__ tm,
                                                          float4 *__restrict__ pts4) {
    label_order_body<L>(seg, hist, P, M, KP, T, order, ri, tm, pts4, blockIdx.y, blockIdx.x);
}
struct OrderGroup {   // one geometry group of rpcc_compress_batch_mixed
    const uint8_t *seg;
    const uint32_t *hist;
    int P, T;
    uint32_t *order;
    const float *ri, *tm;
    float4 *pts4;
};
__global__ __launch_bounds__(256) void label_order_multi_kernel(const MultiArgs<OrderGroup> m, int M, int KP) {
    int b, t;
    const OrderGroup &a = multi_locate(m, b, t);
    label_order_body<uint8_t>(a.seg, a.hist, a.P, M, KP, a.T, a.order, a.ri, a.tm, a.pts4, b, t);
}

// points of one label, through the ordered pixel list
struct LabelPoints {
    const float4 *pts;  // this label's slice of the ordered point list (x, y, z, r)
    int n;
    __device__ __forceinline__ void getf(int i, float &x, float &y, float &z) const {
        const float4 q = pts[i];
        x = q.x; y = q.y; z = q.z;
    }
    __device__ __forceinline__ void get(int i, double &x, double &y, double &z) const {
        float a, b, c;
        getf(i, a, b, c);
        x = (double)a; y = (double)b; z = (double)c;
    }
};

struct PlaneParams {
    double cos_cut;      // reject when some pixel has v <= cos_cut (v = |n.t|/|n| * |t|); from the host's arccos
    float thr;           // RANSAC inlier distance
    int min_points;      // 30
    int iters;           // 10
    uint32_t seed;
    const int64_t *frame_ids;  // dev i64 [B] or nullptr: the frame's identity for seeding (nullptr: its index in the batch)
    const double *inject;      // dev f64 [B,K,4] or nullptr: planes that replace the RANSAC result (test hook: pins the glue)
};

// ------------------------------------------------------------------------------------------------------------------
// One WAVEFRONT per (label, frame).  A label has ~1000 points: as a 256-thread workgroup its chain of short passes is
// mostly barriers (two 8-level tree sums, the leaves of the mean) and three of the four wavefronts idle during the fits.
// A single wavefront needs no barrier and no LDS, and four times as many labels are in flight per CU.  The arithmetic is
// the workgroup form's (ransac_plane_wg), operation by operation:
//   * ordered fp64 sums: the specification's 256 strided partials, lane l owns partials l, l+64, l+128, l+192 -- the
//     strides 128 and 64 of the halving tree are lane-local additions, 32 .. 1 are lane shifts;
//   * hypotheses: lane h fits hypothesis h, the planes are broadcast with v_readlane (wave-uniform operands of the
//     packed-fp32 scoring);
//   * NumPy's pairwise fp32 mean: eight leaves at a time, eight lanes (= the eight accumulators) per leaf.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_bcast_f64(double v, int src) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)u, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(u >> 32), src, 64);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int NV>
__device__ __forceinline__ void wave_treesum(const double (&a)[4][NV], double (&out)[NV]) {
#pragma unroll
    for (int q = 0; q < NV; q++) {
        double v = (a[0][q] + a[2][q]) + (a[1][q] + a[3][q]);  // strides 128, then 64
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v = v + __shfl_down(v, o, 64);  // lanes below the stride are the specification's
        out[q] = wave_bcast_f64(v, 0);
    }
}

template <int RN, int MAXH>
__device__ int ransac_plane_wave(const float4 *__restrict__ pts, int n, int iters, float thr_f, uint32_t seed, double plane[4]) {
    const int lane = threadIdx.x & 63;
    plane[0] = 0; plane[1] = 0; plane[2] = 1; plane[3] = 0;
    if (n < RN || iters > MAXH) return 0;
    // (1) fits: hypothesis h on lane h
    double pl[4] = {0, 0, 0, 0};
    bool ok = false;
    if (lane < iters) {
        const int h = lane;
        int idx[RN];
#pragma unroll
        for (int k = 0; k < RN; k++) {
            uint32_t a = 0;
            int cand;
            bool dup;
            do {
                cand = (int)(mix32(seed, (uint32_t)(h * 16 + k), a++) % (uint32_t)n);
                dup = false;
#pragma unroll
                for (int j = 0; j < RN; j++) dup |= (j < k) && (idx[j] == cand);
            } while (dup);
            idx[k] = cand;
        }
        double px[RN], py[RN], pz[RN];
        double c[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < RN; k++) {
            const float4 q = pts[idx[k]];
            px[k] = (double)q.x; py[k] = (double)q.y; pz[k] = (double)q.z;
            c[0] += px[k]; c[1] += py[k]; c[2] += pz[k];
        }
        c[0] /= (double)RN; c[1] /= (double)RN; c[2] /= (double)RN;
        double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
#pragma unroll
        for (int k = 0; k < RN; k++) {
            const double rx = px[k] - c[0], ry = py[k] - c[1], rz = pz[k] - c[2];
            xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
        }
        ok = plane_from_moments(c, xx, xy, xz, yy, yz, zz, pl);
    }
    const unsigned long long okmask = __ballot(ok);
    // (2) scoring: every hypothesis in one pass over the points
    const float pfl[4] = {(float)pl[0], (float)pl[1], (float)pl[2], (float)pl[3]};
    // the planes are wave-uniform (scalar registers); the inlier counts are too: a compare writes a lane mask, s_bcnt1 counts
    // it -- one vector instruction per test, no per-lane counters and no reduction afterwards.  Lanes past the end of the
    // list test the point (inf, 0, 0): inf or NaN on every plane, never an inlier.
    float pf[MAXH][4];
    int cnt[MAXH];
#pragma unroll
    for (int q = 0; q < MAXH; q++) {
        const bool v = q < iters && ((okmask >> q) & 1ull);
        pf[q][0] = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(pfl[0]), q));
        pf[q][1] = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(pfl[1]), q));
        pf[q][2] = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(pfl[2]), q));
        const float d = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(pfl[3]), q));
        pf[q][3] = v ? d : __builtin_inff();  // invalid -> never an inlier
        cnt[q] = 0;
    }
    for (int i0 = 0; i0 < n; i0 += 256) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = pts[min(i0 + 64 * u + lane, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float x = i0 + 64 * u + lane < n ? p[u].x : __builtin_inff();
            const rs_v2f xx = {x, x}, yy = {p[u].y, p[u].y}, zz = {p[u].z, p[u].z};
#pragma unroll
            for (int q = 0; q + 1 < MAXH; q += 2) {
                const rs_v2f a = {pf[q][0], pf[q + 1][0]}, b2 = {pf[q][1], pf[q + 1][1]}, c2 = {pf[q][2], pf[q + 1][2]},
                             d2 = {pf[q][3], pf[q + 1][3]};
                const rs_v2f dd = ((a * xx + b2 * yy) + c2 * zz) + d2;
                cnt[q] += (int)__popcll(__ballot(fabsf(dd.x) < thr_f));
                cnt[q + 1] += (int)__popcll(__ballot(fabsf(dd.y) < thr_f));
            }
            if (MAXH & 1) cnt[MAXH - 1] += (int)__popcll(__ballot(plane_inlier(pf[MAXH - 1], x, p[u].y, p[u].z, thr_f)));
        }
    }
    int wcnt = -1, wh = 0;
#pragma unroll
    for (int q = 0; q < MAXH; q++)
        if (q < iters && ((okmask >> q) & 1ull) && cnt[q] > wcnt) { wcnt = cnt[q]; wh = q; }  // most inliers, lowest h among equals
    if (wcnt < 0) return 0;
#pragma unroll
    for (int j = 0; j < 4; j++) plane[j] = wave_bcast_f64(pl[j], wh);
    if (wcnt < 3) return wcnt;
    const float wf[4] = {(float)plane[0], (float)plane[1], (float)plane[2], (float)plane[3]};
    // (3) refit on the winner's inliers: ordered fp64 sums, centroid, then moments
    double a3[4][3];
#pragma unroll
    for (int u = 0; u < 4; u++) { a3[u][0] = 0; a3[u][1] = 0; a3[u][2] = 0; }
    for (int i0 = 0; i0 < n; i0 += 256) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = pts[min(i0 + 64 * u + lane, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i0 + 64 * u + lane < n && plane_inlier(wf, p[u].x, p[u].y, p[u].z, thr_f)) {
                a3[u][0] += (double)p[u].x; a3[u][1] += (double)p[u].y; a3[u][2] += (double)p[u].z;
            }
    }
    double c[3];
    wave_treesum<3>(a3, c);
    c[0] /= (double)wcnt; c[1] /= (double)wcnt; c[2] /= (double)wcnt;
    double a6[4][6];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 6; q++) a6[u][q] = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = pts[min(i0 + 64 * u + lane, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i0 + 64 * u + lane < n && plane_inlier(wf, p[u].x, p[u].y, p[u].z, thr_f)) {
                const double rx = (double)p[u].x - c[0], ry = (double)p[u].y - c[1], rz = (double)p[u].z - c[2];
                a6[u][0] += rx * rx; a6[u][1] += rx * ry; a6[u][2] += rx * rz; a6[u][3] += ry * ry; a6[u][4] += ry * rz; a6[u][5] += rz * rz;
            }
    }
    double m[6];
    wave_treesum<6>(a6, m);
    double r[4];
    if (plane_from_moments(c, m[0], m[1], m[2], m[3], m[4], m[5], r)) { plane[0] = r[0]; plane[1] = r[1]; plane[2] = r[2]; plane[3] = r[3]; }
    return wcnt;
}

// NumPy's fp32 pairwise sum of one block (<= 8192 elements) of a label's ranges, by one wavefront, without walking the
// recursion: its call tree (split at len/2 rounded down to a multiple of 8 until len <= 128; at most 7 splits deep for 8192
// elements) is laid out as a binary heap in LDS -- node h has children 2h, 2h+1 --
//   top-down, one level per step, lanes = nodes:  (offset, length) of every node;
//   the leaves (0 < length <= 128), eight at a time: lane group g = leaf, lane j of the group = accumulator j of NumPy's
//   unrolled-by-8 loop, then ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the tail elements in order;
//   bottom-up, one level per step:  value(h) = value(2h) + value(2h+1).
struct NpwLds {
    int off[256], len[256];
    float val[256];
    unsigned char list[128];
};
__device__ __forceinline__ void npw_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ float np_block_sum_wave(const float4 *__restrict__ pts, int blk_off, int bl, NpwLds &W) {
    const int lane = threadIdx.x & 63, g = lane >> 3, j = lane & 7;
    npw_sync();
    if (lane == 0) { W.off[1] = blk_off; W.len[1] = bl; }
    for (int d = 0; d < 7; d++) {
        npw_sync();
        for (int h = (1 << d) + lane; h < (2 << d); h += 64) {
            const int l = W.len[h], o = W.off[h];
            int n2 = l / 2;
            n2 -= n2 % 8;
            const bool split = l > 128;
            W.off[2 * h] = o; W.len[2 * h] = split ? n2 : 0;
            W.off[2 * h + 1] = o + n2; W.len[2 * h + 1] = split ? l - n2 : 0;
        }
    }
    npw_sync();
    int nl = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int h = c * 64 + lane;
        const int l = h >= 1 ? W.len[h] : 0;
        const bool leaf = l > 0 && l <= 128;
        const unsigned long long m = __ballot(leaf);
        if (leaf) W.list[nl + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned char)h;
        nl += __popcll(m);
    }
    npw_sync();
    for (int l0 = 0; l0 < nl; l0 += 8) {
        const int h = W.list[min(l0 + g, nl - 1)];
        const int o = W.off[h], l = W.len[h];
        float res;
        if (l < 8) {
            res = -0.0f;
            for (int i = 0; i < l; i++) res += pts[o + i].w;
        } else {
            const int body = l - (l % 8);
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = pts[o + min(8 * q + j, l - 1)].w;
            float r = v[0];
#pragma unroll
            for (int q = 1; q < 16; q++) if (8 * q + j < body) r += v[q];
            const float r1 = r + __shfl_xor(r, 1, 64);       // (r0+r1), (r2+r3), ...
            const float r2 = r1 + __shfl_xor(r1, 2, 64);     // (r0+r1)+(r2+r3), ...
            res = r2 + __shfl_xor(r2, 4, 64);                // fp addition commutes bit for bit: every lane of the group holds NumPy's value
            for (int i = body; i < l; i++) res += pts[o + i].w;
        }
        if (j == 0 && l0 + g < nl) W.val[h] = res;
    }
    for (int d = 6; d >= 0; d--) {
        npw_sync();
        for (int h = (1 << d) + lane; h < (2 << d); h += 64)
            if (W.len[h] > 128) W.val[h] = W.val[2 * h] + W.val[2 * h + 1];
    }
    npw_sync();
    return W.val[1];
}
// NumPy fp32 mean of a label's ranges by one wavefront: 8192-element blocks added in sequence.
__device__ float np_mean_wave(const float4 *__restrict__ pts, int n, NpwLds &W) {
    if (n == 0) return u2f(0xFFC00000u);
    float total = 0.0f;
    for (int blk = 0; blk < n; blk += 8192) {
        const float r = np_block_sum_wave(pts, blk, min(8192, n - blk), W);
        total = (blk == 0) ? r : total + r;
    }
    return total / (float)n;
}
// ... by a workgroup: the wavefronts take a block each, the block sums are added in block order.
__device__ float np_mean_wg(const float4 *__restrict__ pts, int n, NpwLds *W, float *part) {
    if (n == 0) return u2f(0xFFC00000u);
    const int wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    float total = 0.0f;
    for (int blk0 = 0; blk0 < n; blk0 += 8192 * NW) {
        const int blk = blk0 + 8192 * wave;
        if (blk < n) {
            const float r = np_block_sum_wave(pts, blk, min(8192, n - blk), W[wave]);
            if ((threadIdx.x & 63) == 0) part[wave] = r;
        }
        __syncthreads();
        for (int w = 0; w < NW && blk0 + 8192 * w < n; w++) total = (blk0 == 0 && w == 0) ? part[0] : total + part[w];
        __syncthreads();
    }
    return total / (float)n;
}

// plane_angle_validation (segment_utils.py:84-93) for one pixel: v = |n.t| / |n| * |t| in the reference's operation order;
// "nan" when arccos(v) is NaN (alpha.max() is NaN then and NaN > threshold is False), "bad" when the angle is above the
// threshold (v <= cos_cut).  The fp64 division is only done for the pixels the multiplication by 1/|n| cannot decide: its
// result is within a few ulp of v, so it decides every pixel that is not within 1e-12 of cos_cut or of 1.
__device__ __forceinline__ void plane_angle_test(float tx, float ty, float tz, double a, double bb, double c, double nrm, double inv_nrm,
                                                 double cos_cut, bool in, bool &nan, bool &bad) {
    const double dot = fabs(((double)tx * a + (double)ty * bb) + (double)tz * c);
    const float tn = sqrtf((tx * tx + ty * ty) + tz * tz);
    const double w = dot * inv_nrm * (double)tn;
    const bool decided = w > cos_cut + 1e-12 && w < 1.0 - 1e-12;  // neither bad nor nan
    if (in && !decided) {
        const double v = dot / nrm * (double)tn;
        nan |= (v != v) || v > 1.0;
        bad |= v <= cos_cut;
    }
}

#define PL_THREADS 256
#define PL_RU 8
#define PL_MAXH 10
// Workgroup form for one label (all PL_THREADS threads call it): ransac_plane_wg with thread t < 256 owning the strided
// partial t of the ordered fp64 sums and the wavefronts sharing the points of the scoring, validation and mean passes.  Used
// for the large labels, where one wavefront alone would be the tail of the launch.
struct PlaneWgLds {
    float part[PL_THREADS / 64];
    double sred[6 * RS_NT];
    double swin[64 + PL_MAXH * 4];
    int sbest[32];
};
__device__ void plane_label_wg(const float *__restrict__ tm, const uint32_t *__restrict__ order, const float4 *__restrict__ pl_pts,
                               int n, uint32_t seed, const PlaneParams &pp, PlaneWgLds &S, NpwLds *NW_, float *__restrict__ row,
                               const double *__restrict__ inject) {
    constexpr int NTH = PL_THREADS;
    const int tid = threadIdx.x;
    bool use_plane = false;
    double plane[4] = {0, 0, 0, 0};
    if (n >= pp.min_points) {
        LabelPoints pts;
        pts.pts = pl_pts; pts.n = n;
        ransac_plane_wg<4, NTH, PL_MAXH, 4, LabelPoints, true, PL_RU>(pts, pp.iters, (double)pp.thr, seed, plane, S.sred, S.swin, S.sbest);
        if (inject) { plane[0] = inject[0]; plane[1] = inject[1]; plane[2] = inject[2]; plane[3] = inject[3]; }
        const double a = plane[0], bb = plane[1], c = plane[2];
        const double nrm = sqrt((a * a + bb * bb) + c * c), inv_nrm = 1.0 / nrm;
        bool bad = false, nan = false;
        uint32_t p[4], pn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) pn[u] = order[min(tid + NTH * u, n - 1)];
        for (int i0 = tid; i0 < n; i0 += NTH * 4) {  // 4 pixels per thread in flight; the next indices load under the ray gather
            float tx[4], ty[4], tz[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { p[u] = pn[u]; tx[u] = tm[3 * p[u]]; ty[u] = tm[3 * p[u] + 1]; tz[u] = tm[3 * p[u] + 2]; }
#pragma unroll
            for (int u = 0; u < 4; u++) pn[u] = order[min(i0 + NTH * (4 + u), n - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) plane_angle_test(tx[u], ty[u], tz[u], a, bb, c, nrm, inv_nrm, pp.cos_cut, i0 + NTH * u < n, nan, bad);
        }
        const int any_nan = __syncthreads_or(nan ? 1 : 0), any_bad = __syncthreads_or(bad ? 1 : 0);
        use_plane = any_nan || !any_bad;
    }
    if (use_plane) {
        if (tid < 4) row[tid] = (float)plane[tid];
    } else {
        const float mean = np_mean_wg(pl_pts, n, NW_, S.part);
        if (tid < 4) row[tid] = tid == 3 ? mean : 0.0f;
    }
}

// One launch, two kinds of workgroups:
//   blockIdx <  B*(K-2): one (frame, label >= 2) each; returns at once unless the label has more than `big` points, which
//                        the whole workgroup then fits.  These start first: the long labels are the critical path.
//   the others:          PL_THREADS/64 consecutive labels of a frame, one wavefront each (labels above `big` skipped).
// Measured on 256 frames of 64x2048 (26 k labels: 48 % below 30 points, median 35, 90 % below 900; per frame five to nine
// labels above 2048 points holding 25 k .. 60 k of its pixels, the largest 15 k): DESIGN.md section 6.
#define PL_BIG 4096
#define PL_WAVES 4
template <int MAXH>
__device__ __forceinline__ void plane_model_body(const float *__restrict__ tm,
                                                 const uint32_t *__restrict__ order_all,
                                                 const float4 *__restrict__ pts_all,
                                                 const uint32_t *__restrict__ hist,
                                                 const int32_t *__restrict__ counts,
                                                 const double *__restrict__ ground, int B, int P, int M, int KP, int T,
                                                 const PlaneParams &pp, int big, float *__restrict__ model, const int wg) {   // wg: workgroup index of the launch layout above
    __shared__ PlaneWgLds S;
    __shared__ NpwLds npw[PL_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = M + 2, nbig = B * (K - 2);
    if (wg < nbig) {  // workgroup-uniform
        const int b = wg / (K - 2), kk = 2 + wg % (K - 2);
        const int nn = counts[(int64_t)b * K + kk];
        if (nn <= big) return;
        const uint32_t fid = pp.frame_ids ? (uint32_t)pp.frame_ids[b] : (uint32_t)b;
        const uint32_t base = hist[((int64_t)b * T) * KP + kk];  // tile 0 offset = start of label kk in the ordered list
        plane_label_wg(tm, order_all + (int64_t)b * P + base, pts_all + (int64_t)b * P + base, nn, mix32(pp.seed, fid, (uint32_t)kk),
                       pp, S, npw, model + ((int64_t)b * K + kk) * 4, pp.inject ? pp.inject + ((int64_t)b * K + kk) * 4 : nullptr);
        return;
    }
    const int groups = (K + PL_THREADS / 64 - 1) / (PL_THREADS / 64), g = wg - nbig;
    const int b = g / groups, k = (g % groups) * (PL_THREADS / 64) + wave;
    const uint32_t fid = pp.frame_ids ? (uint32_t)pp.frame_ids[b] : (uint32_t)b;
    if (k >= K) return;
    float *row = model + ((int64_t)b * K + k) * 4;
    if (k == 0) {
        if (lane < 4) row[lane] = ground ? (float)ground[4 * b + lane] : 0.0f;
        return;
    }
    if (k == 1) {
        if (lane < 4) row[lane] = 0.0f;
        return;
    }
    const int n = counts[(int64_t)b * K + k];
    if (n > big) return;
    const uint32_t base = hist[((int64_t)b * T) * KP + k];
    const uint32_t *order = order_all + (int64_t)b * P + base;
    const float4 *pts = pts_all + (int64_t)b * P + base;
    bool use_plane = false;
    double plane[4] = {0, 0, 0, 0};
    if (n >= pp.min_points) {
        const uint32_t seed = mix32(pp.seed, fid, (uint32_t)k);
        ransac_plane_wave<4, MAXH>(pts, n, pp.iters, pp.thr, seed, plane);
        if (pp.inject) {
            const double *ij = pp.inject + ((int64_t)b * K + k) * 4;
            plane[0] = ij[0]; plane[1] = ij[1]; plane[2] = ij[2]; plane[3] = ij[3];
        }
        const double a = plane[0], bb = plane[1], c = plane[2];
        const double nrm = sqrt((a * a + bb * bb) + c * c), inv_nrm = 1.0 / nrm;
        bool bad = false, nan = false;
        uint32_t p[4], pn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) pn[u] = order[min(64 * u + lane, n - 1)];
        for (int i0 = 0; i0 < n; i0 += 256) {  // the next indices load under the ray gather
            float tx[4], ty[4], tz[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { p[u] = pn[u]; tx[u] = tm[3 * p[u]]; ty[u] = tm[3 * p[u] + 1]; tz[u] = tm[3 * p[u] + 2]; }
#pragma unroll
            for (int u = 0; u < 4; u++) pn[u] = order[min(i0 + 256 + 64 * u + lane, n - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++)
                plane_angle_test(tx[u], ty[u], tz[u], a, bb, c, nrm, inv_nrm, pp.cos_cut, i0 + 64 * u + lane < n, nan, bad);
        }
        use_plane = __ballot(nan) != 0ull || __ballot(bad) == 0ull;
    }
    if (use_plane) {
        if (lane < 4) row[lane] = (float)plane[lane];
    } else {
        const float mean = np_mean_wave(pts, n, npw[wave]);
        if (lane < 4) row[lane] = lane == 3 ? mean : 0.0f;
    }
}
template <int MAXH>
__global__ __launch_bounds__(PL_THREADS) __attribute__((amdgpu_waves_per_eu(PL_WAVES, 8))) void plane_model_kernel(const float *__restrict__ tm,
                                                                 const uint32_t *__restrict__ order_all,
                                                                 const float4 *__restrict__ pts_all,
                                                                 const uint32_t *__restrict__ hist,
                                                                 const int32_t *__restrict__ counts,
                                                                 const double *__restrict__ ground, int B, int P, int M, int KP, int T,
                                                                 PlaneParams pp, int big, float *__restrict__ model) {
    plane_model_body<MAXH>(tm, order_all, pts_all, hist, counts, ground, B, P, M, KP, T, pp, big, model, blockIdx.x);
}
// the labels of several geometry groups in one launch (rpcc_compress_batch_mixed; fps_kernels.h: fps_regtab_planar_multi_kernel)
struct PlaneGroupArgs {
    const float *tm;
    const uint32_t *order_all;
    const float4 *pts_all;
    const uint32_t *hist;
    const int32_t *counts;
    const double *ground;
    int B, P, T;
    PlaneParams pp;
    float *model;
};
struct PlaneMulti {
    int n, first[RPCC_MAX_GROUPS + 1];
    PlaneGroupArgs a[RPCC_MAX_GROUPS];
};
template <int MAXH>
__global__ __launch_bounds__(PL_THREADS) __attribute__((amdgpu_waves_per_eu(PL_WAVES, 8))) void plane_model_multi_kernel(const PlaneMulti m, int M, int KP, int big) {
    const int gi = multi_group_of(m.first, m.n, blockIdx.x);
    const PlaneGroupArgs &a = m.a[gi];
    plane_model_body<MAXH>(a.tm, a.order_all, a.pts_all, a.hist, a.counts, a.ground, a.B, a.P, M, KP, a.T, a.pp, big, a.model, (int)blockIdx.x - m.first[gi]);
}
