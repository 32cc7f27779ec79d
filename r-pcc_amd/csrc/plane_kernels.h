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

// label-ordered pixel list: order[b][pos] = pixel index, labels ascending (label 1 skipped), row-major
// inside a label -- the same positions the quantiser's ordered scatter uses.
__global__ __launch_bounds__(256) void label_order_kernel(const uint8_t *__restrict__ seg, const uint32_t *__restrict__ hist,
                                                          int P, int M, int KP, int T, uint32_t *__restrict__ order) {
    extern __shared__ unsigned char smem_raw[];
    uint32_t *segcnt = reinterpret_cast<uint32_t *>(smem_raw);  // [16][KP]
    const int b = blockIdx.y, t = blockIdx.x, K = M + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * KP; i += 256) segcnt[i] = 0u;
    __syncthreads();
    int lab[4], rank[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = t * TILE + j * 256 + threadIdx.x;
        lab[j] = -1;
        rank[j] = 0;
        if (p < P) {
            const int l = seg[(int64_t)b * P + p];
            lab[j] = (l == 1) ? -1 : l;
        }
        int todo = lab[j];
        while (true) {
            const unsigned long long pending = __ballot(todo >= 0);
            if (!pending) break;
            const int leader = (int)__ffsll((long long)pending) - 1;
            const int cur = __shfl(todo, leader, 64);
            const unsigned long long same = __ballot(todo == cur);
            if (todo == cur) {
                rank[j] = __popcll(same & ((1ull << lane) - 1ull));
                if (lane == leader) segcnt[(j * 4 + wave) * KP + cur] = (uint32_t)__popcll(same);
                todo = -1;
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        uint32_t run = hist[((int64_t)b * T + t) * KP + k];
        for (int s = 0; s < 16; s++) {
            const uint32_t c = segcnt[s * KP + k];
            segcnt[s * KP + k] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (lab[j] >= 0)
            order[(int64_t)b * P + segcnt[(j * 4 + wave) * KP + lab[j]] + rank[j]] = (uint32_t)(t * TILE + j * 256 + threadIdx.x);
}

// points of one label, through the ordered pixel list
struct LabelPoints {
    const uint32_t *order;  // this label's slice
    const float *ri;        // this frame's range image
    const float *tm;
    int n;
    __device__ __forceinline__ void getf(int i, float &x, float &y, float &z) const {
        const uint32_t p = order[i];
        const float r = ri[p];
        x = r * tm[3 * p]; y = r * tm[3 * p + 1]; z = r * tm[3 * p + 2];
    }
    __device__ __forceinline__ void get(int i, double &x, double &y, double &z) const {
        float a, b, c;
        getf(i, a, b, c);
        x = (double)a; y = (double)b; z = (double)c;
    }
};

// hypothesis h of the specification: RN distinct sample indices from the counter-based hash, plane by
// centroid + centred moments (fp64, sequential sums over the sample)
template <int RN, class PTS>
__device__ __forceinline__ bool ransac_fit_hypothesis(const PTS &pts, int n, uint32_t seed, int h, double pl[4]) {
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
    for (int k = 0; k < RN; k++) { pts.get(idx[k], px[k], py[k], pz[k]); c[0] += px[k]; c[1] += py[k]; c[2] += pz[k]; }
    c[0] /= (double)RN; c[1] /= (double)RN; c[2] /= (double)RN;
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
#pragma unroll
    for (int k = 0; k < RN; k++) {
        const double rx = px[k] - c[0], ry = py[k] - c[1], rz = pz[k] - c[2];
        xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
    }
    pl[0] = pl[1] = pl[2] = pl[3] = 0.0;
    return plane_from_moments(c, xx, xy, xz, yy, yz, zz, pl);
}

// ordered fp64 sum of the specification evaluated by ONE wavefront: lane l owns partials l, l+64, l+128,
// l+192 (the 256 strided partials), then the binary tree partial[t] += partial[t+stride].
__device__ __forceinline__ double wave_treesum256(double p0, double p1, double p2, double p3, int lane) {
    p0 += p2;  // stride 128: t = l      += l+128 ; t = l+64 += l+192
    p1 += p3;
    p0 += p1;  // stride 64
    double v = p0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const double o = __shfl_down(v, s, 64);
        if (lane < s) v += o;
    }
    return __shfl(v, 0, 64);
}

// Single-wavefront RANSAC (workgroup = 64 threads).  iters <= 64.  hyp: LDS [iters*5] floats + [iters*4] doubles.
template <int RN, int HMAX, class PTS>
__device__ int ransac_plane_wave(const PTS &pts, int iters, float thr_f, uint32_t seed, double plane[4], float *hypf,
                                 double *hypd) {
    const int lane = threadIdx.x & 63, n = pts.n;
    plane[0] = 0; plane[1] = 0; plane[2] = 1; plane[3] = 0;
    if (n < RN || iters > HMAX) return 0;
    if (lane < iters) {
        double pl[4];
        const bool ok = ransac_fit_hypothesis<RN>(pts, n, seed, lane, pl);
        hypf[5 * lane] = (float)pl[0]; hypf[5 * lane + 1] = (float)pl[1]; hypf[5 * lane + 2] = (float)pl[2];
        hypf[5 * lane + 3] = ok ? (float)pl[3] : __builtin_inff();
        hypf[5 * lane + 4] = ok ? 1.0f : 0.0f;
        hypd[4 * lane] = pl[0]; hypd[4 * lane + 1] = pl[1]; hypd[4 * lane + 2] = pl[2]; hypd[4 * lane + 3] = pl[3];
    }
    __syncthreads();
    float pf[HMAX][4];
    int cnt[HMAX];
#pragma unroll
    for (int q = 0; q < HMAX; q++) {
        const int hh = q < iters ? q : 0;
        pf[q][0] = hypf[5 * hh]; pf[q][1] = hypf[5 * hh + 1]; pf[q][2] = hypf[5 * hh + 2];
        pf[q][3] = q < iters ? hypf[5 * hh + 3] : __builtin_inff();
        cnt[q] = 0;
    }
    for (int i = lane; i < n; i += 64) {
        float x, y, z;
        pts.getf(i, x, y, z);
#pragma unroll
        for (int q = 0; q < HMAX; q++) cnt[q] += plane_inlier(pf[q], x, y, z, thr_f);
    }
    int best_cnt = -1, best_h = 0;
#pragma unroll
    for (int q = 0; q < HMAX; q++) {
        const int c = (int)dpp_sum_u32((uint32_t)cnt[q]);
        if (q < iters && hypf[5 * q + 4] != 0.0f && c > best_cnt) { best_cnt = c; best_h = q; }
    }
    if (best_cnt < 0) return 0;
    plane[0] = hypd[4 * best_h]; plane[1] = hypd[4 * best_h + 1]; plane[2] = hypd[4 * best_h + 2]; plane[3] = hypd[4 * best_h + 3];
    if (best_cnt < 3) return best_cnt;
    const float wf[4] = {(float)plane[0], (float)plane[1], (float)plane[2], (float)plane[3]};
    double c[3];
    {
        double s[3][4];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int j = 0; j < 4; j++) s[a][j] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            for (int i = lane + 64 * j; i < n; i += 256) {
                float x, y, z;
                pts.getf(i, x, y, z);
                if (plane_inlier(wf, x, y, z, thr_f)) { s[0][j] += (double)x; s[1][j] += (double)y; s[2][j] += (double)z; }
            }
#pragma unroll
        for (int a = 0; a < 3; a++) c[a] = wave_treesum256(s[a][0], s[a][1], s[a][2], s[a][3], lane) / (double)best_cnt;
    }
    double m[6];
    {
        double s[6][4];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int j = 0; j < 4; j++) s[a][j] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            for (int i = lane + 64 * j; i < n; i += 256) {
                float x, y, z;
                pts.getf(i, x, y, z);
                if (plane_inlier(wf, x, y, z, thr_f)) {
                    const double rx = (double)x - c[0], ry = (double)y - c[1], rz = (double)z - c[2];
                    s[0][j] += rx * rx; s[1][j] += rx * ry; s[2][j] += rx * rz; s[3][j] += ry * ry; s[4][j] += ry * rz; s[5][j] += rz * rz;
                }
            }
#pragma unroll
        for (int a = 0; a < 6; a++) m[a] = wave_treesum256(s[a][0], s[a][1], s[a][2], s[a][3], lane);
    }
    double pl[4];
    if (plane_from_moments(c, m[0], m[1], m[2], m[3], m[4], m[5], pl)) { plane[0] = pl[0]; plane[1] = pl[1]; plane[2] = pl[2]; plane[3] = pl[3]; }
    return best_cnt;
}

// NumPy fp32 pairwise sum of v[0..len) held in LDS (len <= 128): the leaf of the recursion.
__device__ __forceinline__ float np_leaf_sum(const float *leaf, float *acc8, int len, int lane) {
    float res = 0.0f;
    if (len < 8) {
        if (lane == 0) {
            res = -0.0f;
            for (int i = 0; i < len; i++) res += leaf[i];
            acc8[0] = res;
        }
    } else {
        const int body = len - (len % 8);
        if (lane < 8) {
            float r = leaf[lane];
            for (int i = 8 + lane; i < body; i += 8) r += leaf[i];
            acc8[lane] = r;
        }
        __syncthreads();
        if (lane == 0) {
            res = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
            for (int i = body; i < len; i++) res += leaf[i];
            acc8[0] = res;
        }
    }
    __syncthreads();
    res = acc8[0];
    __syncthreads();
    return res;
}

// NumPy fp32 mean of the label's ranges (row-major order through `order`), one wavefront.
__device__ float np_mean_wave(const uint32_t *order, const float *ri, int n, float *leaf, float *acc8, int lane) {
    if (n == 0) return u2f(0xFFC00000u);
    float total = 0.0f;
    for (int blk = 0; blk < n; blk += 8192) {
        const int bl = min(8192, n - blk);
        // post-order walk of the pairwise recursion (split at len/2 rounded down to a multiple of 8)
        int off[12], len[12], phase[12], sp = 0;
        float lv[12];
        off[0] = blk; len[0] = bl; phase[0] = 0; sp = 1;
        float result = 0.0f;
        while (sp > 0) {
            const int t = sp - 1;
            float v;
            bool done = false;
            if (len[t] <= 128) {
                for (int i = lane; i < len[t]; i += 64) leaf[i] = ri[order[off[t] + i]];
                __syncthreads();
                v = np_leaf_sum(leaf, acc8, len[t], lane);
                done = true;
            } else if (phase[t] == 0) {
                int n2 = len[t] / 2;
                n2 -= n2 % 8;
                phase[t] = 1;
                off[sp] = off[t]; len[sp] = n2; phase[sp] = 0; sp++;
            } else if (phase[t] == 1) {
                int n2 = len[t] / 2;
                n2 -= n2 % 8;
                phase[t] = 2;
                off[sp] = off[t] + n2; len[sp] = len[t] - n2; phase[sp] = 0; sp++;
            } else {
                v = lv[t];  // left + right already combined below
                done = true;
            }
            if (done) {
                sp--;
                if (sp == 0) result = v;
                else {
                    const int par = sp - 1;
                    if (phase[par] == 1) lv[par] = v;       // left child finished
                    else lv[par] = lv[par] + v;             // right child finished: left + right
                }
            }
        }
        total = (blk == 0) ? result : total + result;
    }
    return total / (float)n;
}

struct PlaneParams {
    double cos_cut;      // reject when some pixel has v <= cos_cut (v = |n.t|/|n| * |t|); from the host's arccos
    float thr;           // RANSAC inlier distance
    int min_points;      // 30
    int iters;           // 10
    uint32_t seed;
};

__global__ __launch_bounds__(64) void plane_model_kernel(const float *__restrict__ ri_all, const float *__restrict__ tm,
                                                         const uint32_t *__restrict__ order_all,
                                                         const uint32_t *__restrict__ hist, const int32_t *__restrict__ counts,
                                                         const double *__restrict__ ground, int P, int M, int KP, int T,
                                                         PlaneParams pp, float *__restrict__ model) {
    __shared__ float leaf[128];
    __shared__ float acc8[8];
    __shared__ float hypf[16 * 5];
    __shared__ double hypd[16 * 4];
    const int k = blockIdx.x, b = blockIdx.y, K = M + 2, lane = threadIdx.x;
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
    const uint32_t base = hist[((int64_t)b * T) * KP + k];  // tile 0 offset = start of label k in the ordered list
    const uint32_t *order = order_all + (int64_t)b * P + base;
    const float *ri = ri_all + (int64_t)b * P;
    bool use_plane = false;
    double plane[4] = {0, 0, 0, 0};
    if (n >= pp.min_points) {
        LabelPoints pts;
        pts.order = order; pts.ri = ri; pts.tm = tm; pts.n = n;
        const uint32_t seed = mix32(pp.seed, (uint32_t)b, (uint32_t)k);
        ransac_plane_wave<4, 10>(pts, pp.iters, pp.thr, seed, plane, hypf, hypd);
        // plane_angle_validation (segment_utils.py:84-93)
        const double a = plane[0], bb = plane[1], c = plane[2];
        const double nrm = sqrt((a * a + bb * bb) + c * c);
        bool bad = false, nan = false;
        for (int i = lane; i < n; i += 64) {
            const uint32_t p = order[i];
            const float tx = tm[3 * p], ty = tm[3 * p + 1], tz = tm[3 * p + 2];
            const double dot = fabs(((double)tx * a + (double)ty * bb) + (double)tz * c);
            const float tn = sqrtf((tx * tx + ty * ty) + tz * tz);
            const double v = dot / nrm * (double)tn;
            nan |= (v != v) || v > 1.0;   // arccos gives NaN, alpha.max() is NaN, NaN > threshold is False
            bad |= v <= pp.cos_cut;
        }
        use_plane = __any(nan) || !__any(bad);
    }
    if (use_plane) {
        if (lane < 4) row[lane] = (float)plane[lane];
    } else {
        const float mean = np_mean_wave(order, ri, n, leaf, acc8, lane);
        if (lane < 4) row[lane] = lane == 3 ? mean : 0.0f;
    }
}
