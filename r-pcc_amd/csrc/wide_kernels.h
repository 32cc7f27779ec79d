// wide_kernels.h -- the path for cluster_num above RPCC_MAX_CLUSTERS (labels as uint16); included by rpcc_hip.hip.
//
// The reference takes any cluster_num (cfgs/compressor.yaml:22; its labels travel as uint16, utils/compress_utils.py:160).  The batch kernels of
// this library keep a label in one byte and their per-label tables in LDS; this file is the plain form of the same stages for 255 .. 65 533
// clusters: one thread per pixel or per label, per-label totals by global atomics, and the ordered scatter (residuals grouped by label ascending,
// row-major inside a label: cpp_modules.cpp:326-331) through ONE stable radix sort of (frame, label) keys -- the position of every pixel in its
// frame's stream (wide_positions_kernel) then serves the quantiser, the plane model's label-ordered lists and the decoder alike.  Same arithmetic,
// same results as the byte-label kernels (tests: test_gpu_wide.py against the oracle); written for correctness first (89 k frames/s of 64 x 2048 at
// 300 clusters against 360 k of the byte-label kernels at 100).
#pragma once
#include <cstring>
#include <rocprim/rocprim.hpp>

#define RPCC_MAX_CLUSTERS_WIDE 65533

// a7: first minimum over the ground term and the M radii (utils/segment_utils.py:21-23,64-67,127-131,168-169): two smallest squared distances,
// then assign_label's tie window and ground screen.  cen4: float4 [B,M] (x, y, z, 0).
// A wavefront owns 64 consecutive pixels.  It first lists the centres that can matter to any of them (64 centres per step, one per lane): with the box
// of its non-empty pixels, u = min over the centres of the bound ABOVE every pixel's distance (per-axis farthest gap), and a centre stays when its bound
// BELOW every pixel's distance (per-axis nearest gap) is <= u * 1.00001 -- both in the operation order of the distance itself on gaps that bracket every
// pixel's (rounding is monotone), so a dropped centre is farther than the nearest one by more than assign_label's tie window for every pixel: same m1,
// k1, same tie decision.  The list keeps ascending centre order (the first minimum wins).  More than WIDE_ASSIGN_CAP survivors: every centre, as before.
#define WIDE_ASSIGN_CAP 256
__global__ __launch_bounds__(256) void wide_assign_kernel(const float *__restrict__ ri, const float *__restrict__ tm, const double *__restrict__ ground,
                                                          const float4 *__restrict__ cen4, int P, int M, uint16_t *__restrict__ seg) {
    __shared__ uint16_t s_list[4][WIDE_ASSIGN_CAP];
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x, pc = min(p, P - 1), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float r = ri[(int64_t)b * P + pc];
    const float tx = tm[3 * pc], ty = tm[3 * pc + 1], tz = tm[3 * pc + 2];
    const float x = r * tx, y = r * ty, z = r * tz;
    const float4 *cb = cen4 + (int64_t)b * M;
    AssignGround G;
    G.a = ground[4 * b]; G.b = ground[4 * b + 1]; G.c = ground[4 * b + 2]; G.d = ground[4 * b + 3];
    G.af = (float)G.a; G.bf = (float)G.b; G.cf = (float)G.c; G.df = (float)G.d;
    G.S = (float)((fabs(G.a) + fabs(G.b) + fabs(G.c)) * 1.001);
    const float inf = __builtin_inff();
    // the wavefront's box (empty pixels become label 1 whatever their distances: left out)
    const bool in = p < P && r != 0.0f;
    float lo0 = in ? x : inf, lo1 = in ? y : inf, lo2 = in ? z : inf, hi0 = in ? x : -inf, hi1 = in ? y : -inf, hi2 = in ? z : -inf;
    dpp_box6(lo0, lo1, lo2, hi0, hi1, hi2);
    int n = 0;   // listed centres; -1: all of them (a wavefront of empty pixels: none)
    if (lo0 <= hi0) {   // (wave-uniform) some pixel is not empty
        float u = inf;
        for (int k0 = 0; k0 < M; k0 += 64) {
            const float4 cc = cb[min(k0 + lane, M - 1)];
            const float fx = fmaxf(fabsf(lo0 - cc.x), fabsf(hi0 - cc.x)), fy = fmaxf(fabsf(lo1 - cc.y), fabsf(hi1 - cc.y)),
                        fz = fmaxf(fabsf(lo2 - cc.z), fabsf(hi2 - cc.z));
            u = fminf(u, (fx * fx + fy * fy) + fz * fz);
        }
        u = dpp_min_f32_native(u);
        const float cut = u * 1.00001f;
        const unsigned long long lt = (1ull << lane) - 1ull;
        n = 0;
        for (int k0 = 0; k0 < M; k0 += 64) {
            const int k = k0 + lane;
            const float4 cc = cb[min(k, M - 1)];
            const float gx = fmaxf(fmaxf(lo0 - cc.x, cc.x - hi0), 0.0f), gy = fmaxf(fmaxf(lo1 - cc.y, cc.y - hi1), 0.0f),
                        gz = fmaxf(fmaxf(lo2 - cc.z, cc.z - hi2), 0.0f);
            const bool keep = k < M && !((gx * gx + gy * gy) + gz * gz > cut);   // (an unordered compare keeps the centre)
            const unsigned long long m = __ballot(keep);
            const int pos = n + (int)__popcll(m & lt);
            if (keep && pos < WIDE_ASSIGN_CAP) s_list[wave][pos] = (uint16_t)k;
            n += (int)__popcll(m);
        }
        if (n > WIDE_ASSIGN_CAP) n = -1;
    }
    float m1 = inf, m2 = inf;
    int k1 = -1;
    const int cnt = n < 0 ? M : n;
    for (int i = 0; i < cnt; i++) {   // (wave-uniform index: the centre is one scalar load)
        const int k = n < 0 ? i : __builtin_amdgcn_readfirstlane((int)s_list[wave][i]);
        const float4 cc = cb[k];
        const float dx = x - cc.x, dy = y - cc.y, dz = z - cc.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        const bool lt = d2 < m1;
        m2 = lt ? m1 : (d2 < m2 ? d2 : m2);
        k1 = lt ? k : k1;
        m1 = lt ? d2 : m1;
    }
    int label = assign_label(r, tx, ty, tz, x, y, z, m1, m2, k1, cb, G);
    if (r == 0.0f) label = 1;
    if (p < P) seg[(int64_t)b * P + p] = (uint16_t)label;
}
__global__ __launch_bounds__(256) void wide_cen4_kernel(const float *__restrict__ centers, int n, float4 *__restrict__ cen4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) cen4[i] = make_float4(centers[3 * (int64_t)i], centers[3 * (int64_t)i + 1], centers[3 * (int64_t)i + 2], 0.0f);
}

// sort keys (frame << 16 | label; value = pixel) and the per-label totals: pixel counts and, for the point model, the exact fixed-point range
// sums of model_hist_kernel (r * 2^28 as an integer for 2^-5 <= r < 2^8, else the frame's flag: sequential fp64 loop in wide_point_model_kernel)
// (Labels are spatially coherent: the 64 consecutive pixels of a wavefront hold a handful of labels.  Up to WIDE_KEY_ROUNDS of them -- the label of the
// first pixel still pending -- are counted and summed once per wavefront, the rest adds itself; integer sums, any order.  One device atomic per pixel on
// a frame's few hundred counters took 12 ms per 256 frames.)
#define WIDE_KEY_ROUNDS 6
template <class L>
__global__ __launch_bounds__(256) void wide_keys_kernel(const L *__restrict__ seg, const float *__restrict__ ri, int P, int K, uint32_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals, int32_t *__restrict__ counts, unsigned long long *__restrict__ sums,
                                                        int32_t *__restrict__ flags) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const bool valid = p < P;
    const int64_t g = (int64_t)b * P + (valid ? p : P - 1);
    // A label map from a foreign or damaged stream (rpcc_decode_wide) may hold values above K - 1: they count as label K - 1 here and in
    // wide_decode_kernel (memory-safe; flags[4 b + 1] says so), never as an index behind the frame's K counters.
    const uint32_t lraw = seg[g];
    const uint32_t l = min(lraw, (uint32_t)(K - 1));
    if (__any(valid && lraw != l) && lane == 0) flags[4 * b + 1] = 1;
    if (valid) { keys[g] = ((uint32_t)b << 16) | l; vals[g] = (uint32_t)p; }
    uint32_t lo = 0u, hi = 0u;   // r * 2^28 = hi << 18 | lo: the sums of either part over a wavefront stay below 2^32
    bool bad = false;
    if (valid && ri != nullptr && l >= 2u) {
        const uint32_t rb = f2u(ri[g]);
        const float r = u2f(rb);
        if (!(r >= 0.03125f && r < 256.0f)) {
            bad = true;
        } else {   // biased exponent 122 .. 134: r * 2^28 = mantissa << (exponent - 122)
            const uint32_t sh = (rb >> 23) - 122u, m = (rb & 0x7FFFFFu) | 0x800000u;
            lo = (m << sh) & 0x3FFFFu;
            hi = m >> (18u - sh);
        }
    }
    if (__any(bad) && lane == 0) flags[4 * b] = 1;
    unsigned long long pend = __ballot(valid);
#pragma unroll 1
    for (int round = 0; round < WIDE_KEY_ROUNDS && pend != 0ull; round++) {
        const int first = (int)__ffsll((long long)pend) - 1;
        const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)l, first);
        const bool mine = valid && l == cur;
        const unsigned long long m = __ballot(mine);
        pend &= ~m;
        const uint32_t slo = dpp_sum_u32(mine ? lo : 0u), shi = dpp_sum_u32(mine ? hi : 0u);
        if (lane == 0) {
            atomicAdd(&counts[(int64_t)b * K + cur], (int32_t)__popcll(m));
            const unsigned long long sv = (unsigned long long)slo + ((unsigned long long)shi << 18);
            if (sv) atomicAdd(&sums[(int64_t)b * K + cur], sv);
        }
    }
    if ((pend >> lane) & 1ull) {
        atomicAdd(&counts[(int64_t)b * K + l], 1);
        const unsigned long long sv = (unsigned long long)lo + ((unsigned long long)hi << 18);
        if (sv) atomicAdd(&sums[(int64_t)b * K + l], sv);
    }
}
// exclusive prefix of the label counts of a frame without label 1 (the empty pixels have no residual: cpp_modules.cpp:314): base[b][k] = first
// position of label k in the frame's stream; nnz[b] = its length.  One workgroup per frame, a contiguous run of labels per thread.
__global__ __launch_bounds__(256) void wide_bases_kernel(const int32_t *__restrict__ counts, int K, uint32_t *__restrict__ base, int32_t *__restrict__ nnz) {
    __shared__ uint32_t part[256];
    const int b = blockIdx.x, per = (K + 255) / 256, k0 = threadIdx.x * per, k1 = min(k0 + per, K);
    const int32_t *c = counts + (int64_t)b * K;
    uint32_t s = 0;
    for (int k = k0; k < k1; k++) s += k == 1 ? 0u : (uint32_t)c[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; i++) { const uint32_t v = part[i]; part[i] = run; run += v; }
        if (nnz) nnz[b] = (int32_t)run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (int k = k0; k < k1; k++) { base[(int64_t)b * K + k] = run; run += k == 1 ? 0u : (uint32_t)c[k]; }
}
// sorted element i of frame b (keys ascending, the sort is stable: pixels ascending inside a label) -> its pixel's position in the frame's stream,
// the label-ordered pixel list and (optionally) point list.  pos = -1 for label 1.
__global__ __launch_bounds__(256) void wide_positions_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, const int32_t *__restrict__ counts,
                                                             int P, int K, int64_t n, int32_t *__restrict__ pos, uint32_t *__restrict__ order,
                                                             const float *__restrict__ ri, const float *__restrict__ tm, float4 *__restrict__ pts4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys[i], p = vals[i], b = key >> 16, l = key & 0xFFFFu;
    const int64_t j = i - (int64_t)b * P;
    const int64_t o = l == 1u ? -1 : j - (l > 1u ? counts[(int64_t)b * K + 1] : 0);
    pos[(int64_t)b * P + p] = (int32_t)o;
    if (o >= 0) {
        order[(int64_t)b * P + o] = p;
        if (pts4) {
            const float r = ri[(int64_t)b * P + p];
            pts4[(int64_t)b * P + o] = make_float4(r * tm[3 * p], r * tm[3 * p + 1], r * tm[3 * p + 2], r);  // transformer.py:94-101
        }
    }
}
// a8 rows (cpp_modules.cpp:471-518, segment_utils.py:183-185): one thread per (frame, label)
template <class L>
__global__ __launch_bounds__(256) void wide_point_model_kernel(const float *__restrict__ ri, const L *__restrict__ seg, const double *__restrict__ ground,
                                                               const int32_t *__restrict__ counts, const unsigned long long *__restrict__ sums,
                                                               const int32_t *__restrict__ flags, int P, int K, float *__restrict__ model) {
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float *row = model + ((int64_t)b * K + k) * 4;
    if (k == 0) {
        row[0] = (float)ground[4 * b]; row[1] = (float)ground[4 * b + 1]; row[2] = (float)ground[4 * b + 2]; row[3] = (float)ground[4 * b + 3];
    } else if (k == 1) {
        row[0] = row[1] = row[2] = row[3] = 0.0f;
    } else {
        const int total = counts[(int64_t)b * K + k];
        double s;
        if (flags[4 * b]) {   // sequential double accumulation in row-major order (cpp_modules.cpp:514)
            s = 0.0;
            for (int p = 0; p < P; p++)
                if (seg[(int64_t)b * P + p] == (L)k) s += (double)ri[(int64_t)b * P + p];
        } else {
            s = (double)(long long)sums[(int64_t)b * K + k] * (1.0 / 268435456.0);
        }
        row[0] = row[1] = row[2] = 0.0f;
        row[3] = total == 0 ? u2f(0xFFC00000u) : (float)(s / (double)total);
    }
}
// a13 from the per-label totals (salience_levels_kernel for any K)
__global__ __launch_bounds__(256) void wide_salience_levels_kernel(const int32_t *__restrict__ counts, const int32_t *__restrict__ kpn, int K, SalienceParams sp,
                                                                   uint8_t *__restrict__ salience, float *__restrict__ label_acc) {
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int pn = counts[(int64_t)b * K + k], kn = kpn[(int64_t)b * K + k];
    int lv = 0;
    if (k == 0) lv = sp.ground_level;
    else if (k == 1) lv = sp.levels - 1;
    else if (pn < 30) lv = sp.levels - 1;
    else
        for (int l = 0; l < sp.levels; l++)
            if (kn >= sp.level_kp_num[l]) { lv = l; break; }
    salience[(int64_t)b * K + k] = (uint8_t)lv;
    label_acc[(int64_t)b * K + k] = sp.level_acc[lv];
}
// a10 + a11 / a13: prediction, residual, quantisation, the integer to its position (cpp_modules.cpp:248-285,288-334, compress.py:106)
template <class L>
__global__ __launch_bounds__(256) void wide_quantise_kernel(const float *__restrict__ ri, const float *__restrict__ tm, const L *__restrict__ seg,
                                                            const float *__restrict__ model, const int32_t *__restrict__ pos, float acc,
                                                            const float *__restrict__ label_acc, int P, int K, int16_t *__restrict__ q16) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int64_t g = (int64_t)b * P + p;
    const int o = pos[g];
    if (o < 0) return;
    const int l = seg[g];
    const float *m = model + ((int64_t)b * K + l) * 4;
    const float p0 = m[0], p1 = m[1], p2 = m[2], p3 = m[3];
    float pr;
    if (p0 + p1 + p2 == 0.0f) pr = p3;
    else pr = -p3 / (p0 * tm[3 * p] + p1 * tm[3 * p + 1] + p2 * tm[3 * p + 2]);
    const float res = ri[g] - pr;
    const float step = label_acc ? label_acc[(int64_t)b * K + l] : acc;
    q16[(int64_t)b * P + o] = (int16_t)(int)roundf(res / step);   // astype(np.int16): two's-complement truncation
}
// f3: the decoder's body on the positions (decode_kernel for any K)
template <class L>
__global__ __launch_bounds__(256) void wide_decode_kernel(const L *__restrict__ seg, const int16_t *__restrict__ q16, const float *__restrict__ model,
                                                          const float *__restrict__ tm, const int32_t *__restrict__ pos, const uint8_t *__restrict__ salience,
                                                          DecodeSteps steps, int P, int K, float *__restrict__ ri_rec, float *__restrict__ pc_rec) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int64_t g = (int64_t)b * P + p;
    const int l = min((int)seg[g], K - 1), o = pos[g];   // (wide_keys_kernel: labels above K - 1 count as K - 1)
    const float *m = model + ((int64_t)b * K + l) * 4;
    const float p0 = m[0], p1 = m[1], p2 = m[2], p3 = m[3];
    const float tx = tm[3 * p], ty = tm[3 * p + 1], tz = tm[3 * p + 2];
    float pr;
    if (p0 + p1 + p2 == 0.0f) pr = p3;
    else pr = -p3 / (p0 * tx + p1 * ty + p2 * tz);
    float res = 0.0f;
    if (o >= 0) {
        const double st = steps.levels ? steps.acc[min((int)salience[(int64_t)b * K + l], steps.levels - 1)] : steps.acc[0];
        res = (float)((double)q16[(int64_t)b * P + o] * st);
    }
    const float rec = pr + res;
    ri_rec[g] = rec;
    if (pc_rec) { pc_rec[3 * g] = rec * tx; pc_rec[3 * g + 1] = rec * ty; pc_rec[3 * g + 2] = rec * tz; }
}
