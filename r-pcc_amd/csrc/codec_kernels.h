// codec_kernels.h -- contour codec (f1, f3) and the decoder's residual gather (f3); included by rpcc_hip.hip.
//
//   f1  extract_contour  ops/cpp_modules/src/cpp_modules.cpp:521-558  +  np.packbits / uint16 casts of
//       compress_point_cloud (utils/compress_utils.py:156-160)
//   f3  recover_map      cpp_modules.cpp:561-593;  dequantize_residual  utils/compress_utils.py:114-132;
//       range_image_rec = pred + residual, range_image_to_point_cloud  (tools/decompress.py:88-112)
#pragma once

// ------------------------------------------------------------------------------------------------
// f1: contour bit of pixel (h,w) = (w == 0) || seg[h,w] != seg[h,w-1]; idx_sequence = labels at the
// contour positions in row-major order (uint16); contour_map = np.packbits (first pixel = MSB).
// Pass 1 counts contour bits per 1024-pixel tile, pass 2 turns the counts into offsets (one workgroup
// per frame), pass 3 writes the packed bits and scatters the labels.
// ------------------------------------------------------------------------------------------------
// (L: the label type -- uint8_t, or uint16_t for cluster_num above RPCC_MAX_CLUSTERS, wide_kernels.h)
template <class L>
__device__ __forceinline__ bool contour_bit(const L *__restrict__ seg, int p, int W) {
    const int col = p % W;
    return col == 0 || seg[p] != seg[p - 1];
}

template <class L>
__global__ __launch_bounds__(256) void contour_count_kernel(const L *__restrict__ seg, int P, int W, int T,
                                                            uint32_t *__restrict__ tile_cnt) {
    __shared__ int s[4];
    const int b = blockIdx.y, t = blockIdx.x;
    const L *sg = seg + (int64_t)b * P;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < TILE / 256; j++) {
        const int p = t * TILE + j * 256 + threadIdx.x;
        const bool c = p < P && contour_bit(sg, min(p, P - 1), W);
        cnt += __popcll(__ballot(c));
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[(int64_t)b * T + t] = (uint32_t)(s[0] + s[1] + s[2] + s[3]);
}

// exclusive scan of the per-tile counts of one frame (T <= 4096 handled in chunks of 256), total -> nseq
__global__ __launch_bounds__(256) void tile_scan_kernel(uint32_t *__restrict__ tile_cnt, int T, int32_t *__restrict__ total) {
    __shared__ uint32_t sh[256];
    const int b = blockIdx.x;
    uint32_t *c = tile_cnt + (int64_t)b * T;
    uint32_t run = 0;
    for (int base = 0; base < T; base += 256) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < T ? c[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan
            const uint32_t add = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < T) c[i] = run + sh[threadIdx.x] - v;
        run += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) total[b] = (int32_t)run;
}

template <class L>
__global__ __launch_bounds__(256) void contour_write_kernel(const L *__restrict__ seg, int P, int W, int T,
                                                            const uint32_t *__restrict__ tile_off,
                                                            uint8_t *__restrict__ bits, uint16_t *__restrict__ seq) {
    __shared__ uint32_t segcnt[16];
    const int b = blockIdx.y, t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const L *sg = seg + (int64_t)b * P;
    const int nbytes = (P + 7) >> 3;
    uint8_t *ob = bits + (int64_t)b * nbytes;
    uint16_t *os = seq + (int64_t)b * P;
    bool c[TILE / 256];
    unsigned long long m[TILE / 256];
#pragma unroll
    for (int j = 0; j < TILE / 256; j++) {
        const int p = t * TILE + j * 256 + threadIdx.x;
        c[j] = p < P && contour_bit(sg, min(p, P - 1), W);
        m[j] = __ballot(c[j]);
        if (lane == 0) segcnt[j * 4 + wave] = (uint32_t)__popcll(m[j]);
        // 64 pixels -> 8 bytes, pixel 8k+i of the segment is bit (7-i) of byte k
        if (lane < 8) {
            const int byte_idx = ((t * TILE + j * 256 + wave * 64) >> 3) + lane;
            if (byte_idx < nbytes) {
                const uint32_t v = (uint32_t)((m[j] >> (8 * lane)) & 0xFFull);
                ob[byte_idx] = (uint8_t)(__brev(v) >> 24);
            }
        }
    }
    __syncthreads();
    uint32_t run = tile_off[(int64_t)b * T + t];
#pragma unroll
    for (int j = 0; j < TILE / 256; j++) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t n = segcnt[j * 4 + w];
            if (w == wave && c[j]) {
                const int p = t * TILE + j * 256 + threadIdx.x;
                os[run + __popcll(m[j] & ((1ull << lane) - 1ull))] = (uint16_t)sg[p];
            }
            run += n;
        }
    }
}

// f3: recover_map.  Label of pixel p = seq[(number of contour bits at positions <= p) - 1].
__global__ __launch_bounds__(256) void contour_bits_count_kernel(const uint8_t *__restrict__ bits, int P, int T,
                                                                 uint32_t *__restrict__ tile_cnt) {
    __shared__ int s[4];
    const int b = blockIdx.y, t = blockIdx.x;
    const int nbytes = (P + 7) >> 3;
    const uint8_t *ib = bits + (int64_t)b * nbytes;
    // a tile = 1024 pixels = 128 bytes; threads 0..127 take one byte each
    int cnt = 0;
    if (threadIdx.x < TILE / 8) {
        const int byte_idx = t * (TILE / 8) + threadIdx.x;
        if (byte_idx < nbytes) {
            uint32_t v = ib[byte_idx];
            const int valid = P - byte_idx * 8;  // pixels of this byte inside the image
            if (valid < 8) v &= 0xFFu << (8 - valid);
            cnt = __popc(v);
        }
    }
    cnt = (int)dpp_sum_u32((uint32_t)cnt);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[(int64_t)b * T + t] = (uint32_t)(s[0] + s[1] + s[2] + s[3]);
}

template <class L>
__global__ __launch_bounds__(256) void recover_map_kernel(const uint8_t *__restrict__ bits, const uint16_t *__restrict__ seq,
                                                          int P, int T, const uint32_t *__restrict__ tile_off,
                                                          L *__restrict__ seg) {
    __shared__ uint32_t segcnt[16];
    const int b = blockIdx.y, t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nbytes = (P + 7) >> 3;
    const uint8_t *ib = bits + (int64_t)b * nbytes;
    const uint16_t *is = seq + (int64_t)b * P;
    bool c[TILE / 256];
    unsigned long long m[TILE / 256];
#pragma unroll
    for (int j = 0; j < TILE / 256; j++) {
        const int p = t * TILE + j * 256 + threadIdx.x;
        const int pc = min(p, P - 1);
        c[j] = p < P && ((ib[pc >> 3] >> (7 - (pc & 7))) & 1);
        m[j] = __ballot(c[j]);
        if (lane == 0) segcnt[j * 4 + wave] = (uint32_t)__popcll(m[j]);
    }
    __syncthreads();
    uint32_t run = tile_off[(int64_t)b * T + t];
#pragma unroll
    for (int j = 0; j < TILE / 256; j++) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w == wave) {
                const int p = t * TILE + j * 256 + threadIdx.x;
                // inclusive count of contour bits up to this pixel
                const uint32_t k = run + (uint32_t)__popcll(m[j] & ((2ull << lane) - 1ull));
                if (p < P) seg[(int64_t)b * P + p] = (L)(k ? is[k - 1] : 0);
            }
            run += segcnt[j * 4 + w];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// f3: decoder body.  For every pixel: pred (intra_predict), residual = (float)((double)q * step) read
// from the label-ordered stream at tile offset + segment prefix + ballot rank (the inverse of the
// encoder's ordered scatter), rec = pred + residual (fp32), optional point cloud rec * tm.
// step: one double (uniform) or per-label through salience (non-uniform).
// ------------------------------------------------------------------------------------------------
struct DecodeSteps {
    double acc[8];  // acc[level]; uniform: acc[0]
    int levels;     // 0 = uniform
};

__global__ __launch_bounds__(256) void decode_kernel(const uint8_t *__restrict__ seg, const int16_t *__restrict__ q16,
                                                     const float *__restrict__ model, const float *__restrict__ tm,
                                                     const uint32_t *__restrict__ hist, const uint8_t *__restrict__ salience,
                                                     DecodeSteps steps, int P, int M, int KP, int T,
                                                     float *__restrict__ ri_rec, float *__restrict__ pc_rec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *smodel = reinterpret_cast<float *>(smem_raw);                 // [KP*4]
    uint32_t *segcnt = reinterpret_cast<uint32_t *>(smodel + 4 * KP);    // [16][KP+1]
    const int SEGP = KP + 1;
    uint32_t *soff = segcnt + 16 * SEGP;                                 // [KP] this tile's input offsets per label
    const int b = blockIdx.y, t = blockIdx.x, K = M + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-frame bases (wave-uniform) + byte offsets; all loads of the tile are issued first (unconditional, clamped)
    seg += (int64_t)b * P;
    q16 += (int64_t)b * P;
    ri_rec += (int64_t)b * P;
    if (pc_rec) pc_rec += (int64_t)b * P * 3;
    int lab[4], rank[4], lraw[4];
    f32x3 ray[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t p = (uint32_t)min(t * TILE + j * 256 + (int)threadIdx.x, P - 1);
        lraw[j] = ld_at(seg, p);
        ray[j] = ld_at(reinterpret_cast<const f32x3 *>(tm), p * 12u);
    }
    for (int i = threadIdx.x; i < 4 * K; i += 256) smodel[i] = model[(int64_t)b * K * 4 + i];
    for (int i = threadIdx.x; i < K; i += 256) soff[i] = hist[((int64_t)b * T + t) * KP + i];
    for (int i = threadIdx.x; i < 16 * SEGP; i += 256) segcnt[i] = 0u;
    __syncthreads();
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
    int16_t qv[4];
#pragma unroll
    for (int j = 0; j < 4; j++)  // gather of the label-ordered integers (clamped: unused for label 1 / outside)
        qv[j] = ld_at(q16, (lab[j] >= 0 ? segcnt[(j * 4 + wave) * SEGP + lab[j]] + (uint32_t)rank[j] : 0u) * 2u);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = t * TILE + j * 256 + threadIdx.x;
        if (p >= P) continue;
        const int l = lraw[j];
        const float p0 = smodel[4 * l], p1 = smodel[4 * l + 1], p2 = smodel[4 * l + 2], p3 = smodel[4 * l + 3];
        float pr;
        if (p0 + p1 + p2 == 0.0f) pr = p3;
        else pr = -p3 / (p0 * ray[j].x + p1 * ray[j].y + p2 * ray[j].z);
        float res = 0.0f;  // label 1 keeps the zero of np.zeros_like (compress_utils.py:115)
        if (lab[j] >= 0) {
            // (a level beyond the configured ones -- a corrupt stream; tools/decompress.py rejects it -- is clamped, never read past acc[])
            const double st = steps.levels ? steps.acc[min((int)salience[(int64_t)b * K + l], steps.levels - 1)] : steps.acc[0];
            res = (float)((double)qv[j] * st);  // int16 * python float -> float64 -> stored into a float32 array
        }
        const float rec = pr + res;          // tools/decompress.py:104
        st_at(ri_rec, (uint32_t)p * 4u, rec);
        if (pc_rec) {
            f32x3 o;
            o.x = rec * ray[j].x; o.y = rec * ray[j].y; o.z = rec * ray[j].z;
            st_at(reinterpret_cast<f32x3 *>(pc_rec), (uint32_t)p * 12u, o);
        }
    }
}

// a3 as its own entry: pc = ri[...,None] * transform_map  (dataset/transformer.py:94-101)
__global__ __launch_bounds__(256) void backproject_kernel(const float *__restrict__ ri, const float *__restrict__ tm, int P,
                                                          float *__restrict__ pc) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) {
        const float r = ri[(int64_t)b * P + p];
        float *o = pc + ((int64_t)b * P + p) * 3;
        o[0] = r * tm[3 * p]; o[1] = r * tm[3 * p + 1]; o[2] = r * tm[3 * p + 2];
    }
}

// f2: the batch's residual stream as the container holds it -- the frames' label-ordered int16 runs back to back
// (`residual_quantized` of compress_point_cloud, utils/compress_utils.py:142,160: one array of nnz entries per
// frame).  packed[prefix(b) + i] = q16[b][i] for i < nnz[b], prefix(b) = nnz[0] + ... + nnz[b-1] computed on the
// device, so neither the D2H copy nor the RCCL gather of a rank's payloads moves the B*P padded array.
// One workgroup copies PACK_EPW entries: 2-byte accesses, lane-consecutive (the destination has no alignment),
// PACK_EPT loads in flight per lane.
#define PACK_EPT 8
#define PACK_EPW (256 * PACK_EPT)
__global__ __launch_bounds__(256) void pack_payload_kernel(const int16_t *__restrict__ q16, const int32_t *__restrict__ nnz,
                                                           int P, int64_t capacity, int16_t *__restrict__ packed,
                                                           int64_t *__restrict__ total) {
    const int b = blockIdx.y, B = gridDim.y;
    const int n = min(max(nnz[b], 0), P);
    const int i0 = blockIdx.x * PACK_EPW;
    const bool last = b == B - 1 && blockIdx.x == 0 && total != nullptr;
    if (i0 >= n && !last) return;
    // prefix over the earlier frames: every wavefront computes it for itself (B is a few hundred values)
    const int lane = threadIdx.x & 63;
    int64_t pre = 0;
    for (int j = lane; j < b; j += 64) pre += min(max(nnz[j], 0), P);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, RPCC_WAVE);
    if (last && threadIdx.x == 0) *total = pre + n;
    const int16_t *src = q16 + (int64_t)b * P;
    int16_t v[PACK_EPT];
#pragma unroll
    for (int k = 0; k < PACK_EPT; k++) {
        const int i = i0 + k * 256 + (int)threadIdx.x;
        v[k] = src[min(i, P - 1)];
    }
#pragma unroll
    for (int k = 0; k < PACK_EPT; k++) {
        const int i = i0 + k * 256 + (int)threadIdx.x;
        if (i < n && pre + i < capacity) packed[pre + i] = v[k];
    }
}
