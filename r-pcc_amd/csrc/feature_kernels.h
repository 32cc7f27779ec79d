// feature_kernels.h -- non-uniform framework: key-point extraction (a12) and salience levels (a13);
// included by rpcc_hip.hip.
//
//   a12  extract_features_with_segment + mark_as_picked   ops/cpp_modules/src/cpp_modules.cpp:10-25,28-121
//   a13  salience part of nonuniform_quantize             cpp_modules.cpp:355-405
//
// Intended semantics of a12 = zero where the reference leaves its outputs uninitialised
// (cpp_modules.cpp:38-43): feat / key_point_map are zero-filled before the kernel runs.
#pragma once

// One wavefront (= one workgroup of 64 threads) owns one image row:
//   compaction of the row's pixels with label >= 2 (ballot scan), curvature on the compacted sequence,
//   `segments` equal chunks; per chunk a bitonic sort of (curvature, position) keys in LDS, the sequential
//   "largest first" pick loop on lane 0, a second sort and the "smallest first" loop.
// LDS (dynamic): row f32[W] | v f32[W] | cbuf f32[W] | keys u64[NS] | vidx u16[W] | picked u8[W]
struct FeatParams {
    int feature_region, segments, sharp_num, less_sharp_num, flat_num;
};

__device__ __forceinline__ bool feat_mark_picked(const float *row, uint8_t *picked, int w_i, int fr) {
    // cpp_modules.cpp:10-25: always marks the pixel itself (dif = 0 at i = 0); a neighbour more than
    // 0.3 m closer rejects it
    bool ret = true;
    const float r = row[w_i];
    for (int i = -fr; i <= fr; i++) {
        const float dif = r - row[w_i + i];
        if (fabsf(dif) < 0.2f) picked[w_i] = 1;
        if (dif > 0.3f) ret = false;
    }
    return ret;
}

__device__ __forceinline__ void feat_bitonic_sort(unsigned long long *keys, int NS, int lane) {
    for (int k = 2; k <= NS; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < NS; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], b = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                }
            }
            __syncthreads();  // single-wavefront workgroup: orders the LDS traffic of consecutive stages
        }
    }
}

__global__ __launch_bounds__(64) void features_kernel(const float *__restrict__ ri, const uint8_t *__restrict__ seg, int H,
                                                      int W, int NS, FeatParams fp, float *__restrict__ feat,
                                                      uint8_t *__restrict__ kp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    float *row = reinterpret_cast<float *>(fsm);
    float *v = row + W;
    float *cbuf = v + W;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(cbuf + W + (W & 1));
    uint16_t *vidx = reinterpret_cast<uint16_t *>(keys + NS);
    uint8_t *picked = reinterpret_cast<uint8_t *>(vidx + W);
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int64_t base = ((int64_t)b * H + h) * W;
    const int fr = fp.feature_region;
    int vl = 0;
    for (int c0 = 0; c0 < W; c0 += 64) {
        const int col = c0 + lane;
        const bool in = col < W;
        const int cc = in ? col : W - 1;
        const float r = ri[base + cc];
        const int lab = seg[base + cc];
        if (in) { row[col] = r; picked[col] = 0; }
        const bool ok = in && lab != 0 && lab != 1;
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const int pos = vl + __popcll(m & ((1ull << lane) - 1ull));
            v[pos] = r;
            vidx[pos] = (uint16_t)col;
        }
        vl += __popcll(m);
    }
    __syncthreads();
    if (vl < fp.segments + fr * 2 + 1) return;  // cpp_modules.cpp:59
    const int L = vl - 2 * fr;
    for (int s = fr + lane; s < vl - fr; s += 64) {  // cpp_modules.cpp:64-72, fp32 in that operation order
        float f = 0.0f;
        const float vs = v[s];
        for (int k = -fr; k <= fr; k++) f += v[s + k] - vs;
        f = f * f;
        f /= (float)(2 * fr);
        f /= vs;
        feat[base + vidx[s]] = f;
        cbuf[s - fr] = f;
    }
    __syncthreads();
    const int chunk = L / fp.segments;
    for (int j = 0; j < fp.segments; j++) {
        const int sp = chunk * j;
        // keys: (curvature bits, compacted position); curvature >= 0 so its bit pattern orders like the value
        for (int i = lane; i < NS; i += 64)
            keys[i] = i < chunk ? (((unsigned long long)f2u(cbuf[sp + i]) << 32) | (unsigned)(sp + i + fr)) : ~0ull;
        __syncthreads();
        feat_bitonic_sort(keys, NS, lane);
        if (lane == 0) {  // cpp_modules.cpp:79-95
            int n = 0;
            for (int i = chunk - 1; i >= 0; i--) {
                const int s = (int)(unsigned)keys[i];
                keys[i] = (unsigned long long)(unsigned)s;  // first = 0
                const int col = vidx[s];
                if (picked[col] == 0)
                    if (feat_mark_picked(row, picked, col, fr)) {
                        n += 1;
                        if (n < fp.sharp_num) kp[base + col] = 3;
                        else if (n < fp.less_sharp_num) kp[base + col] = 2;
                        else break;
                    }
            }
        }
        __syncthreads();
        feat_bitonic_sort(keys, NS, lane);
        if (lane == 0) {  // cpp_modules.cpp:97-112
            int n = 0;
            for (int i = 0; i < chunk; i++) {
                if ((keys[i] >> 32) == 0ull) continue;  // first == 0 (visited, or a genuinely zero curvature)
                const int s = (int)(unsigned)keys[i];
                keys[i] = (unsigned long long)(unsigned)s;
                const int col = vidx[s];
                if (picked[col] == 0)
                    if (feat_mark_picked(row, picked, col, fr)) {
                        n += 1;
                        if (n < fp.flat_num) kp[base + col] = 1;
                        else break;
                    }
            }
        }
        __syncthreads();
    }
}

// a13 (salience): per label p_num (pixels, label 1 excluded by the caller's quantiser anyway) and kp_num
// (key points > 0); level: label 0 -> ground_level, label 1 -> L-1, p_num < 30 -> L-1, else the first
// level l with kp_num >= level_kp_num[l] (cpp_modules.cpp:388-403).  One workgroup per frame.
struct SalienceParams {
    int level_kp_num[8];
    float level_acc[8];
    int levels, ground_level;
};

__global__ __launch_bounds__(256) void salience_kernel(const uint8_t *__restrict__ seg, const uint8_t *__restrict__ kp, int P,
                                                       int M, SalienceParams sp, uint8_t *__restrict__ salience,
                                                       float *__restrict__ label_acc) {
    __shared__ int kpn[256], pn[256];
    const int b = blockIdx.x, K = M + 2;
    kpn[threadIdx.x] = 0;
    pn[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t *sg = seg + (int64_t)b * P, *kk = kp + (int64_t)b * P;
    for (int p0 = 0; p0 < P; p0 += 256 * 4) {
        int lab[4], key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = min(p0 + u * 256 + (int)threadIdx.x, P - 1);
            lab[u] = sg[p]; key[u] = kk[p];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (p0 + u * 256 + (int)threadIdx.x >= P) lab[u] = -1;
            // pixel counts: aggregate the wavefront's lanes per distinct label
            int todo = lab[u];
            while (true) {
                const unsigned long long pending = __ballot(todo >= 0);
                if (!pending) break;
                const int leader = (int)__ffsll((long long)pending) - 1;
                const int cur = __builtin_amdgcn_readlane(todo, leader);
                const unsigned long long same = __ballot(todo == cur);
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&pn[cur], (int)__popcll(same));
                if (todo == cur) todo = -1;
            }
            if (lab[u] >= 0 && key[u] > 0) atomicAdd(&kpn[lab[u]], 1);  // sparse
        }
    }
    __syncthreads();
    const int k = threadIdx.x;
    if (k < K) {
        int lv = 0;
        if (k == 0) lv = sp.ground_level;
        else if (k == 1) lv = sp.levels - 1;
        else if (pn[k] < 30) lv = sp.levels - 1;
        else
            for (int l = 0; l < sp.levels; l++)
                if (kpn[k] >= sp.level_kp_num[l]) { lv = l; break; }
        salience[(int64_t)b * K + k] = (uint8_t)lv;
        label_acc[(int64_t)b * K + k] = sp.level_acc[lv];
    }
}

// a10 as its own entry: intra_predict (cpp_modules.cpp:248-285)
__global__ __launch_bounds__(256) void intra_predict_kernel(const uint8_t *__restrict__ seg, const float *__restrict__ model,
                                                            const float *__restrict__ tm, int P, int K,
                                                            float *__restrict__ pred) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float *m = model + ((int64_t)b * K + seg[(int64_t)b * P + p]) * 4;
    const float p0 = m[0], p1 = m[1], p2 = m[2], p3 = m[3];
    float pr;
    if (p0 + p1 + p2 == 0.0f) pr = p3;
    else pr = -p3 / (p0 * tm[3 * p] + p1 * tm[3 * p + 1] + p2 * tm[3 * p + 2]);
    pred[(int64_t)b * P + p] = pr;
}
