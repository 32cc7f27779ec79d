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
// pts4 (optional): the same list as points (x, y, z, r) so that the plane model streams a label's points instead of
// gathering them through the pixel index.
__global__ __launch_bounds__(256) void label_order_kernel(const uint8_t *__restrict__ seg, const uint32_t *__restrict__ hist,
                                                          int P, int M, int KP, int T, uint32_t *__restrict__ order,
                                                          const float *__restrict__ ri, const float *__restrict__ tm,
                                                          float4 *__restrict__ pts4) {
    extern __shared__ unsigned char smem_raw[];
    uint32_t *segcnt = reinterpret_cast<uint32_t *>(smem_raw);  // [16][KP+1]
    const int SEGP = KP + 1;
    uint32_t *soff = segcnt + 16 * SEGP;                        // [KP] this tile's offsets per label
    const int b = blockIdx.y, t = blockIdx.x, K = M + 2;
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
        lraw[j] = ld_at(seg, p);
        if (pts4) { rr[j] = ld_at(ri, p * 4u); ray[j] = ld_at(reinterpret_cast<const f32x3 *>(tm), p * 12u); }
    }
    for (int i = threadIdx.x; i < K; i += 256) soff[i] = hist[((int64_t)b * T + t) * KP + i];
    for (int i = threadIdx.x; i < 16 * SEGP; i += 256) segcnt[i] = 0u;
    __syncthreads();
    int lab[4], rank[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = t * TILE + j * 256 + threadIdx.x;
        lab[j] = (p < P && lraw[j] != 1) ? lraw[j] : -1;
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

// NumPy fp32 pairwise sum of v[0..len) held in LDS (len <= 128): the leaf of the recursion.
__device__ __forceinline__ float np_leaf_sum(const float *leaf, float *acc8, int len) {
    const int lane = threadIdx.x;  // the first 8 threads of the workgroup do the leaf (every thread takes the barriers)
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

// NumPy fp32 mean of the label's ranges (row-major order through the ordered list); every thread of the workgroup calls it.
__device__ float np_mean_wg(const float4 *pts, int n, float *leaf, float *acc8) {
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
                for (int i = threadIdx.x; i < len[t]; i += blockDim.x) leaf[i] = pts[off[t] + i].w;
                __syncthreads();
                v = np_leaf_sum(leaf, acc8, len[t]);
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

#define PL_THREADS 256
#define PL_MAXH 16
// One 256-thread workgroup per (label, frame): the workgroup form of the RANSAC specification (ransac_plane_wg: thread t
// owns the strided partial t of the ordered fp64 sums), four wavefronts share the point loops.
__global__ __launch_bounds__(PL_THREADS) void plane_model_kernel(const float *__restrict__ tm,
                                                                 const uint32_t *__restrict__ order_all,
                                                                 const float4 *__restrict__ pts_all,
                                                                 const uint32_t *__restrict__ hist,
                                                                 const int32_t *__restrict__ counts,
                                                                 const double *__restrict__ ground, int P, int M, int KP, int T,
                                                                 PlaneParams pp, float *__restrict__ model) {
    __shared__ float leaf[128];
    __shared__ float acc8[8];
    __shared__ double sred[6 * RS_NT];
    __shared__ double swin[64 + PL_MAXH * 4];
    __shared__ int sbest[32];
    const int k = blockIdx.x, b = blockIdx.y, K = M + 2, tid = threadIdx.x;
    float *row = model + ((int64_t)b * K + k) * 4;
    if (k == 0) {
        if (tid < 4) row[tid] = ground ? (float)ground[4 * b + tid] : 0.0f;
        return;
    }
    if (k == 1) {
        if (tid < 4) row[tid] = 0.0f;
        return;
    }
    const int n = counts[(int64_t)b * K + k];
    const uint32_t base = hist[((int64_t)b * T) * KP + k];  // tile 0 offset = start of label k in the ordered list
    const uint32_t *order = order_all + (int64_t)b * P + base;
    const float4 *pl_pts = pts_all + (int64_t)b * P + base;
    bool use_plane = false;
    double plane[4] = {0, 0, 0, 0};
    if (n >= pp.min_points) {
        LabelPoints pts;
        pts.pts = pl_pts; pts.n = n;
        const uint32_t seed = mix32(pp.seed, (uint32_t)b, (uint32_t)k);
        ransac_plane_wg<4, PL_THREADS, PL_MAXH, 4>(pts, pp.iters, (double)pp.thr, seed, plane, sred, swin, sbest);
        // plane_angle_validation (segment_utils.py:84-93)
        const double a = plane[0], bb = plane[1], c = plane[2];
        const double nrm = sqrt((a * a + bb * bb) + c * c);
        bool bad = false, nan = false;
        for (int i0 = tid; i0 < n; i0 += PL_THREADS * 4) {  // 4 pixels per thread in flight (index, then ray gather)
            uint32_t p[4];
            float tx[4], ty[4], tz[4];
#pragma unroll
            for (int u = 0; u < 4; u++) p[u] = order[min(i0 + PL_THREADS * u, n - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) { tx[u] = tm[3 * p[u]]; ty[u] = tm[3 * p[u] + 1]; tz[u] = tm[3 * p[u] + 2]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double dot = fabs(((double)tx[u] * a + (double)ty[u] * bb) + (double)tz[u] * c);
                const float tn = sqrtf((tx[u] * tx[u] + ty[u] * ty[u]) + tz[u] * tz[u]);
                const double v = dot / nrm * (double)tn;
                const bool in = i0 + PL_THREADS * u < n;
                nan |= in && ((v != v) || v > 1.0);   // arccos gives NaN, alpha.max() is NaN, NaN > threshold is False
                bad |= in && v <= pp.cos_cut;
            }
        }
        const int any_nan = __syncthreads_or(nan ? 1 : 0), any_bad = __syncthreads_or(bad ? 1 : 0);
        use_plane = any_nan || !any_bad;
    }
    if (use_plane) {
        if (tid < 4) row[tid] = (float)plane[tid];
    } else {
        const float mean = np_mean_wg(pl_pts, n, leaf, acc8);
        if (tid < 4) row[tid] = tid == 3 ? mean : 0.0f;
    }
}
