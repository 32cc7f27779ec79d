// feature_kernels.h -- non-uniform framework: key-point extraction (a12) and salience levels (a13);
// included by rpcc_hip.hip.
//
//   a12  extract_features_with_segment + mark_as_picked   ops/cpp_modules/src/cpp_modules.cpp:10-25,28-121
//   a13  salience part of nonuniform_quantize             cpp_modules.cpp:355-405
//
// Intended semantics of a12 = zero where the reference leaves its outputs uninitialised
// (cpp_modules.cpp:38-43): feat / key_point_map are zero-filled before the kernel runs.
#pragma once

// One workgroup of four wavefronts owns one image row.
//
// What the reference's loops compute (cpp_modules.cpp:74-112), restated without their sequential form:
//   * cloud_neighbors_picked is dead state: mark_as_picked only ever sets the entry of the pixel it is called for, every
//     compacted position belongs to exactly one chunk and is examined at most once (the second loop skips what the first
//     one visited), so "picked == 0" always holds when it is tested.  What remains of mark_as_picked is the pure
//     predicate  A(col) = no neighbour within +-feature_region is more than 0.3 m closer.
//   * both sorts order the chunk by the pair (curvature, position) -- a total order, positions are unique.  The first loop
//     walks it downwards: the n-th accepted entry gets 3 (n < sharp_num) or 2 (n < less_sharp_num) and the loop stops at
//     the less_sharp_num-th accepted entry; everything from the top down to that entry is "visited" (first = 0).  If
//     fewer entries are accepted the whole chunk is visited.
//   * the second sort moves the visited entries (and genuine zero curvatures) to the front, where the second loop skips
//     them, and leaves the others in their old relative order: the loop walks the unvisited non-zero entries upwards, the
//     n-th accepted one gets 1 (n < flat_num) and it stops at the flat_num-th.
// So per chunk: the less_sharp_num largest and then the flat_num smallest accepted keys, found by repeated wave-wide
// arg-max / arg-min over keys held in registers (no sort, no single-lane loop).  The row's outputs are assembled in LDS
// and written once, coalesced (no memset of the output images).
// LDS (dynamic): row f32[W] | v f32[W] (later the feat row) | cbuf f32[W] | vidx u16[W] | accf u8[W] | kprow u8[W]
struct FeatParams {
    int feature_region, segments, sharp_num, less_sharp_num, flat_num;
};

// wave-wide maximum of 64-bit keys (hi, lo); 0 = "no key".  Two 32-bit DPP reductions.
__device__ __forceinline__ void feat_wave_max(uint32_t hi, uint32_t lo, uint32_t &mh, uint32_t &ml) {
    mh = dpp_max_u32(hi);
    ml = dpp_max_u32(hi == mh ? lo : 0u);
}
__device__ __forceinline__ void feat_wave_min(uint32_t hi, uint32_t lo, uint32_t &mh, uint32_t &ml) {
    mh = dpp_min_u32(hi);
    ml = dpp_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
}

#define FEAT_MAX_PER_LANE 8  // chunk <= 512 entries
#define FEAT_CP 2            // chunks a wavefront processes together
#define FEAT_THREADS 256     // one workgroup (4 wavefronts) per image row
#define FEAT_GPW 16          // 64-column groups per wavefront: W <= 4096
typedef unsigned long long feat_key;  // curvature bits << 32 | compacted position + 1; 0 = no key

// ---- row mode: one chunk per 16-lane DPP row, four chunks per wavefront ----------------------------------------------------
// all-reduce inside the 16-lane rows by rotation (row_ror 8, 4, 2, 1): every lane ends with its row's maximum / minimum
__device__ __forceinline__ uint32_t row_allmax_u32(uint32_t v) {
#define STEP_(ctrl_) v = max(v, (uint32_t)RPCC_DPP(0, v, ctrl_, 0xf))
    STEP_(0x128); STEP_(0x124); STEP_(0x122); STEP_(0x121);
#undef STEP_
    return v;
}
__device__ __forceinline__ uint32_t row_allmin_u32(uint32_t v) {
#define STEP_(ctrl_) v = min(v, (uint32_t)RPCC_DPP(-1, v, ctrl_, 0xf))
    STEP_(0x128); STEP_(0x124); STEP_(0x122); STEP_(0x121);
#undef STEP_
    return v;
}
__device__ __forceinline__ feat_key row_allmax_key(feat_key k) {
    const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
    const uint32_t mh = row_allmax_u32(hi), ml = row_allmax_u32(hi == mh ? lo : 0u);
    return ((feat_key)mh << 32) | ml;
}
__device__ __forceinline__ feat_key row_allmin_key(feat_key k) {
    const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
    const uint32_t mh = row_allmin_u32(hi), ml = row_allmin_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((feat_key)mh << 32) | ml;
}
// the three largest (smallest) of a lane's keys, kept sorted: x is inserted
__device__ __forceinline__ void top3_insert(feat_key x, feat_key &c1, feat_key &c2, feat_key &c3) {
    bool g = x > c1;
    feat_key t = g ? c1 : x; c1 = g ? x : c1; x = t;
    g = x > c2;
    t = g ? c2 : x; c2 = g ? x : c2; x = t;
    c3 = x > c3 ? x : c3;
}
__device__ __forceinline__ void bot3_insert(feat_key x, feat_key &c1, feat_key &c2, feat_key &c3) {
    bool g = x < c1;
    feat_key t = g ? c1 : x; c1 = g ? x : c1; x = t;
    g = x < c2;
    t = g ? c2 : x; c2 = g ? x : c2; x = t;
    c3 = x < c3 ? x : c3;
}
#define FEAT_ROWMODE (-1)  // template value of Q: chunk <= 256 entries, 16 keys per lane
#define FEAT_RQ 16
#define FEAT_ROW_SEGS 32  // row mode: most segments ...
#define FEAT_ROW_FLAT 8   // ... and most flat key points per segment (flat_num - 1)
// Q: keys per lane (chunk <= 64 * Q) or FEAT_ROWMODE; GP: 64-column groups per wavefront (W <= 256 * GP);
// FEATOUT = false (row mode only, ranges >= 0): no curvature image is produced, the curvatures take the place of the row in
// LDS with the "accepted" flag in their sign bit -- 11 bytes of LDS per column instead of 16: 7 workgroups per CU, not 5.
// LabelT: the label type (uint16_t for cluster_num above RPCC_MAX_CLUSTERS: the key points per label then go straight to the global counters)
template <int Q, int GP = FEAT_GPW, bool FEATOUT = true, class LabelT = uint8_t>
__global__ __launch_bounds__(FEAT_THREADS) void features_kernel(const float *__restrict__ ri, const LabelT *__restrict__ seg,
                                                                int H, int W, FeatParams fp, float *__restrict__ feat,
                                                                uint8_t *__restrict__ kp, int32_t *__restrict__ kpn = nullptr,
                                                                int K = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    __shared__ int wcnt[FEAT_THREADS / 64];
    __shared__ int kcnt[256];  // key points per label of this row
    __shared__ feat_key sel_thr[Q == FEAT_ROWMODE ? FEAT_ROW_SEGS : 1], sel_small[Q == FEAT_ROWMODE ? FEAT_ROW_SEGS * FEAT_ROW_FLAT : 1];
    static_assert(FEATOUT || Q == FEAT_ROWMODE, "the compact LDS layout is built for the row mode");
    float *row = reinterpret_cast<float *>(fsm);
    float *v = row + W;
    float *cbuf = FEATOUT ? v + W : row;
    uint16_t *vidx = reinterpret_cast<uint16_t *>((FEATOUT ? cbuf : v) + W);
    uint8_t *accf = reinterpret_cast<uint8_t *>(vidx + W);  // FEATOUT only
    uint8_t *kprow = FEATOUT ? accf + W : accf;
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = ((int64_t)b * H + h) * W;
    const int fr = fp.feature_region;
    // compaction of the row's pixels with label >= 2: every wavefront takes a contiguous run of 64-column groups, keeps
    // them in registers (all loads in flight at once), counts, and after the prefix over the wavefronts writes its part
    const int ngroups = (W + 63) >> 6, gpw = (ngroups + FEAT_THREADS / 64 - 1) / (FEAT_THREADS / 64);
    float rr[GP];
    int ll[GP];
    kcnt[tid] = 0;
#pragma unroll
    for (int u = 0; u < GP; u++) {  // unconditional (clamped) loads
        const int cc = min((wave * gpw + u) * 64 + lane, W - 1);
        rr[u] = ri[base + cc];
        ll[u] = seg[base + cc];
    }
    int cnt = 0;
    unsigned long long okm[GP];
#pragma unroll
    for (int u = 0; u < GP; u++) {
        const int col = (wave * gpw + u) * 64 + lane;
        const bool in = u < gpw && col < W;
        if (in) { row[col] = rr[u]; kprow[col] = 0; }
        okm[u] = __ballot(in && ll[u] != 0 && ll[u] != 1);
        cnt += __popcll(okm[u]);
    }
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();
    int vl = 0, run = 0;
#pragma unroll
    for (int w = 0; w < FEAT_THREADS / 64; w++) { run += w < wave ? wcnt[w] : 0; vl += wcnt[w]; }
#pragma unroll
    for (int u = 0; u < GP; u++) {
        if ((okm[u] >> lane) & 1ull) {
            const int pos = run + __popcll(okm[u] & ((1ull << lane) - 1ull));
            v[pos] = rr[u];
            vidx[pos] = (uint16_t)((wave * gpw + u) * 64 + lane);
        }
        run += __popcll(okm[u]);
    }
    __syncthreads();
    const bool active = vl >= fp.segments + fr * 2 + 1;  // cpp_modules.cpp:59
    const int L = vl - 2 * fr;
    // curvature (cpp_modules.cpp:64-72, fp32 in that operation order) and mark_as_picked's return value (:10-25;
    // fr <= s <= col <= W-1-fr, so col +- fr is inside the row) of the compacted position s
    auto curvature = [&](int s, float &f, bool &ok) {
        f = 0.0f;
        const float vs = v[s];
        for (int k = -fr; k <= fr; k++) f += v[s + k] - vs;
        f = f * f;
        f /= (float)(2 * fr);
        f /= vs;
        const int col = vidx[s];
        ok = true;
        for (int k = -fr; k <= fr; k++) ok = ok && !(vs - row[col + k] > 0.3f);  // row[col] is v[s]
    };
    if constexpr (FEATOUT) {
        if (active) {
            for (int s = fr + tid; s < vl - fr; s += FEAT_THREADS) {
                float f;
                bool ok;
                curvature(s, f, ok);
                cbuf[s - fr] = f;
                accf[s - fr] = ok ? 1 : 0;
            }
        }
        __syncthreads();
        // feat row (aliases v, which is dead now): zero, then the curvatures at their columns
        for (int c = tid; c < W; c += FEAT_THREADS) v[c] = 0.0f;
        __syncthreads();
        if (active)
            for (int i = tid; i < L; i += FEAT_THREADS) v[vidx[i + fr]] = cbuf[i];
    } else {
        uint32_t cb[GP];
#pragma unroll
        for (int u = 0; u < GP; u++) {
            const int s = fr + tid + FEAT_THREADS * u;
            cb[u] = 0u;
            if (active && s < vl - fr) {
                float f;
                bool ok;
                curvature(s, f, ok);
                cb[u] = (f2u(f) & 0x7FFFFFFFu) | (ok ? 0x80000000u : 0u);
            }
        }
        __syncthreads();  // the row is dead: the curvatures take its place
#pragma unroll
        for (int u = 0; u < GP; u++) {
            const int s = fr + tid + FEAT_THREADS * u;
            if (active && s < vl - fr) reinterpret_cast<uint32_t *>(cbuf)[s - fr] = cb[u];
        }
        __syncthreads();
    }
    if (active) {
        const int chunk = L / fp.segments;
        // the less_sharp_num largest, then the flat_num smallest accepted keys of every chunk, FEAT_CP chunks interleaved
        // (independent dependency chains hide the latency of the DPP reductions)
        const int stop_n = max(1, max(fp.sharp_num, fp.less_sharp_num));  // the first loop breaks at this accepted entry
        if constexpr (Q == FEAT_ROWMODE) {
            // Every round of the two selections costs two reductions and a pass over the lane's keys in the register form
            // below; here a chunk lives in one 16-lane row (entry i on lane i % 16: neighbours, which tend to be large
            // together, on different lanes), four chunks per wavefront, the reductions are four DPP rotations each, and a lane
            // keeps its three best remaining keys sorted, so that a round is: reduce, the winner shifts its cache.  A lane
            // that wins a fourth time rebuilds the cache from its keys (once per few wavefronts).
            // The two selections run side by side on different wavefronts (0, 1: the largest; 2, 3: the smallest): the
            // smallest are taken among ALL accepted non-zero keys and filtered by the threshold afterwards -- they come out
            // in ascending order, so the ones below the threshold are exactly the first ones of the reference's second loop.
            const int r = lane & 15, row4 = lane >> 4, half = wave & 1;
            const bool smallest = wave >= 2;
            const feat_key none = ~0ull;
            for (int j0 = half * 4; j0 < fp.segments; j0 += 8) {
                const int j = j0 + row4;
                const bool rowok = j < fp.segments;
                const int sp = chunk * j;
                feat_key key[FEAT_RQ];
#pragma unroll
                for (int q = 0; q < FEAT_RQ; q++) {
                    const int i = r + 16 * q;
                    const bool have = rowok && i < chunk;
                    const int ii = have ? sp + i : 0;
                    if constexpr (FEATOUT) {
                        key[q] = (have && accf[ii] != 0) ? ((feat_key)f2u(cbuf[ii]) << 32) | (feat_key)(uint32_t)(ii + fr + 1) : 0ull;
                    } else {
                        const uint32_t cbits = reinterpret_cast<const uint32_t *>(cbuf)[ii];
                        key[q] = (have && (cbits >> 31)) ? ((feat_key)(cbits & 0x7FFFFFFFu) << 32) | (feat_key)(uint32_t)(ii + fr + 1) : 0ull;
                    }
                }
                if (!smallest) {
                    // the stop_n - 1 largest get 3 / 2, the stop_n-th is the threshold of the visited entries
                    feat_key c1 = 0ull, c2 = 0ull, c3 = 0ull, thr = 0ull;
#pragma unroll
                    for (int q = 0; q < FEAT_RQ; q++) top3_insert(key[q], c1, c2, c3);
                    int left = 3;
                    for (int n = 1; n <= stop_n; n++) {
                        const feat_key m = row_allmax_key(c1);
                        if (__ballot(m != 0ull) == 0ull) break;  // every row of the wavefront is exhausted
                        if (n == stop_n) { thr = m; break; }      // 0 in an exhausted row: all visited
                        const bool win = m != 0ull && c1 == m;
                        if (win) {
                            kprow[vidx[(uint32_t)m - 1u]] = n < fp.sharp_num ? 3 : 2;
                            c1 = c2; c2 = c3; c3 = 0ull;
                            left--;
                        }
                        if (__ballot(win && left == 0) != 0ull) {  // rare: rebuild from the keys below the winner
                            if (win && left == 0) {
                                c1 = 0ull; c2 = 0ull; c3 = 0ull;
#pragma unroll
                                for (int q = 0; q < FEAT_RQ; q++) top3_insert(key[q] < m ? key[q] : 0ull, c1, c2, c3);
                                left = 3;
                            }
                        }
                    }
                    if (rowok && r == 0) sel_thr[j] = thr;
                } else {
                    // the flat_num - 1 smallest accepted entries with a non-zero curvature, in ascending order
                    feat_key d1 = none, d2 = none, d3 = none;
#pragma unroll
                    for (int q = 0; q < FEAT_RQ; q++) {
                        key[q] = u2f((uint32_t)(key[q] >> 32)) != 0.0f ? key[q] : none;
                        bot3_insert(key[q], d1, d2, d3);
                    }
                    if (rowok && r < FEAT_ROW_FLAT) sel_small[j * FEAT_ROW_FLAT + r] = none;
                    int left = 3;
                    for (int n = 1; n < fp.flat_num; n++) {
                        const feat_key m = row_allmin_key(d1);
                        if (__ballot(m != none) == 0ull) break;
                        if (rowok && r == 0) sel_small[j * FEAT_ROW_FLAT + n - 1] = m;
                        const bool win = m != none && d1 == m;
                        if (win) {
                            d1 = d2; d2 = d3; d3 = none;
                            left--;
                        }
                        if (__ballot(win && left == 0) != 0ull) {
                            if (win && left == 0) {
                                d1 = none; d2 = none; d3 = none;
#pragma unroll
                                for (int q = 0; q < FEAT_RQ; q++) bot3_insert(key[q] > m ? key[q] : none, d1, d2, d3);
                                left = 3;
                            }
                        }
                    }
                }
            }
            __syncthreads();  // workgroup-uniform: `active` and the template value are
            for (int i = tid; i < fp.segments * FEAT_ROW_FLAT; i += FEAT_THREADS) {
                const feat_key t = sel_thr[i / FEAT_ROW_FLAT], w = sel_small[i];
                if (i % FEAT_ROW_FLAT < fp.flat_num - 1 && w != none && w < t) kprow[vidx[(uint32_t)w - 1u]] = 1;  // t == 0: all visited
            }
        } else
        if constexpr (Q == 0) {
            // chunks of more than 64 * FEAT_MAX_PER_LANE entries (few segments on a wide image): the same selection with the
            // keys rebuilt from LDS in every round instead of held in registers, one chunk per wavefront at a time
            for (int j = wave; j < fp.segments; j += FEAT_THREADS / 64) {
                const int sp = chunk * j;
                auto key_at = [&](int i) -> feat_key {
                    const int ii = sp + i;
                    return accf[ii] != 0 ? ((feat_key)f2u(cbuf[ii]) << 32) | (feat_key)(uint32_t)(ii + fr + 1) : 0ull;
                };
                feat_key prev = ~0ull, thr = 0ull;
                for (int n = 1; n <= stop_n; n++) {
                    feat_key best = 0ull;
                    for (int i = lane; i < chunk; i += 64) {
                        const feat_key k = key_at(i), c = k < prev ? k : 0ull;
                        best = c > best ? c : best;
                    }
                    uint32_t mh, ml;
                    feat_wave_max((uint32_t)(best >> 32), (uint32_t)best, mh, ml);
                    const feat_key m = ((feat_key)mh << 32) | ml;
                    if (ml == 0u) break;                   // fewer accepted entries: all visited
                    if (n == stop_n) { thr = m; break; }
                    if (lane == 0) kprow[vidx[ml - 1u]] = n < fp.sharp_num ? 3 : 2;
                    prev = m;
                }
                if (thr == 0ull) continue;
                prev = 0ull;
                for (int n = 1; n < fp.flat_num; n++) {
                    feat_key best = ~0ull;
                    for (int i = lane; i < chunk; i += 64) {
                        const feat_key k = key_at(i);
                        const feat_key e = (k != 0ull && k < thr && u2f((uint32_t)(k >> 32)) != 0.0f) ? k : ~0ull;
                        const feat_key c = e > prev ? e : ~0ull;
                        best = c < best ? c : best;
                    }
                    uint32_t mh, ml;
                    feat_wave_min((uint32_t)(best >> 32), (uint32_t)best, mh, ml);
                    if (mh == 0xFFFFFFFFu && ml == 0xFFFFFFFFu) break;
                    if (lane == 0) kprow[vidx[ml - 1u]] = 1;
                    prev = ((feat_key)mh << 32) | ml;
                }
            }
        } else
        for (int j0 = wave * FEAT_CP; j0 < fp.segments; j0 += FEAT_CP * (FEAT_THREADS / 64)) {
            feat_key key[FEAT_CP][Q > 0 ? Q : 1], prev[FEAT_CP], thr[FEAT_CP];
            bool alive[FEAT_CP];
#pragma unroll
            for (int cI = 0; cI < FEAT_CP; cI++) {
                const int sp = chunk * (j0 + cI);
                alive[cI] = j0 + cI < fp.segments;
#pragma unroll
                for (int q = 0; q < Q; q++) {  // curvature bits (>= 0: order like the value) << 32 | compacted position + 1
                    const int i = lane + 64 * q;
                    const bool have = alive[cI] && i < chunk;
                    const int ii = have ? sp + i : 0;
                    key[cI][q] = (have && accf[ii] != 0) ? ((feat_key)f2u(cbuf[ii]) << 32) | (feat_key)(uint32_t)(ii + fr + 1) : 0ull;
                }
                prev[cI] = ~0ull;  // previous winner (exclusive upper bound)
                thr[cI] = 0ull;    // keys >= thr are visited; 0 = the whole chunk
            }
            for (int n = 1; n <= stop_n; n++) {  // largest first: labels 3 / 2
#pragma unroll
                for (int cI = 0; cI < FEAT_CP; cI++) {
                    feat_key best = 0ull;
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const feat_key c = key[cI][q] < prev[cI] ? key[cI][q] : 0ull;
                        best = c > best ? c : best;
                    }
                    uint32_t mh, ml;
                    feat_wave_max((uint32_t)(best >> 32), (uint32_t)best, mh, ml);
                    const feat_key m = ((feat_key)mh << 32) | ml;
                    if (!alive[cI] || ml == 0u) { alive[cI] = false; continue; }  // fewer accepted entries: all visited
                    if (n == stop_n) { thr[cI] = m; continue; }
                    if (lane == 0) kprow[vidx[ml - 1u]] = n < fp.sharp_num ? 3 : 2;
                    prev[cI] = m;
                }
            }
            feat_key ek[FEAT_CP][Q > 0 ? Q : 1];
#pragma unroll
            for (int cI = 0; cI < FEAT_CP; cI++) {
                alive[cI] = thr[cI] != 0ull;
                prev[cI] = 0ull;  // previous winner (exclusive lower bound; every key is > 0)
#pragma unroll
                for (int q = 0; q < Q; q++)  // "first == 0" is skipped (cpp_modules.cpp:99): visited, or a zero curvature
                    ek[cI][q] = (key[cI][q] != 0ull && key[cI][q] < thr[cI] && u2f((uint32_t)(key[cI][q] >> 32)) != 0.0f)
                                    ? key[cI][q] : ~0ull;
            }
            for (int n = 1; n < fp.flat_num; n++) {  // smallest first among the unvisited non-zero entries: label 1
#pragma unroll
                for (int cI = 0; cI < FEAT_CP; cI++) {
                    feat_key best = ~0ull;
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const feat_key c = ek[cI][q] > prev[cI] ? ek[cI][q] : ~0ull;
                        best = c < best ? c : best;
                    }
                    uint32_t mh, ml;
                    feat_wave_min((uint32_t)(best >> 32), (uint32_t)best, mh, ml);
                    if (!alive[cI] || (mh == 0xFFFFFFFFu && ml == 0xFFFFFFFFu)) { alive[cI] = false; continue; }
                    if (lane == 0) kprow[vidx[ml - 1u]] = 1;
                    prev[cI] = ((feat_key)mh << 32) | ml;
                }
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < W; c += FEAT_THREADS) {
        if (FEATOUT && feat) feat[base + c] = v[c];
        kp[base + c] = kprow[c];
        // key points per label for the salience levels (sparse: a few dozen per row), tallied in LDS first: one device
        // atomic per label present in the row instead of one per key point (1.6 M atomics per batch cost 0.2 ms)
        if (kpn && kprow[c] > 0) {
            if (sizeof(LabelT) == 1) atomicAdd(&kcnt[seg[base + c]], 1);
            else atomicAdd(&kpn[(int64_t)b * K + seg[base + c]], 1);
        }
    }
    if (kpn && sizeof(LabelT) == 1) {  // kpn is zeroed by the caller
        __syncthreads();
        if (tid < K && kcnt[tid] > 0) atomicAdd(&kpn[(int64_t)b * K + tid], kcnt[tid]);
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

#define SAL_THREADS 1024  // one workgroup per frame: 16 wavefronts hide the latency of the frame-long scan
// L: label type, KMAX: most labels (256 for byte labels, 1024 for uint16 ones up to RPCC_MAX_CLUSTERS_MID clusters)
template <class L = uint8_t, int KMAX = 256>
__global__ __launch_bounds__(SAL_THREADS) void salience_kernel(const L *__restrict__ seg, const uint8_t *__restrict__ kp, int P,
                                                       int M, SalienceParams sp, uint8_t *__restrict__ salience,
                                                       float *__restrict__ label_acc) {
    static_assert(KMAX <= SAL_THREADS, "a thread per label");
    __shared__ int kpn[KMAX], pn[KMAX];
    const int b = blockIdx.x, K = M + 2;
    if (threadIdx.x < KMAX) { kpn[threadIdx.x] = 0; pn[threadIdx.x] = 0; }
    __syncthreads();
    const L *sg = seg + (int64_t)b * P;
    const uint8_t *kk = kp + (int64_t)b * P;
    for (int p0 = 0; p0 < P; p0 += SAL_THREADS * 4) {
        int lab[4], key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = min(p0 + u * SAL_THREADS + (int)threadIdx.x, P - 1);
            lab[u] = sg[p]; key[u] = kk[p];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (p0 + u * SAL_THREADS + (int)threadIdx.x >= P) lab[u] = -1;
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

// a13 (salience) from per-label totals: pixels per label (the model scan's counts) and key points per label (counted by
// features_kernel): the same rule as salience_kernel below without another pass over the frame.
__device__ __forceinline__ void salience_levels_body(const int32_t *__restrict__ counts, const int32_t *__restrict__ kpn,
                                                     int M, const SalienceParams &sp, uint8_t *__restrict__ salience,
                                                     float *__restrict__ label_acc, const int b) {
    const int K = M + 2, k = threadIdx.x;
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
__global__ __launch_bounds__(256) void salience_levels_kernel(const int32_t *__restrict__ counts, const int32_t *__restrict__ kpn,
                                                              int M, SalienceParams sp, uint8_t *__restrict__ salience,
                                                              float *__restrict__ label_acc) {
    salience_levels_body(counts, kpn, M, sp, salience, label_acc, blockIdx.x);
}
struct SalienceGroup {   // one geometry group of rpcc_compress_batch_mixed (the groups may differ in their non-uniform settings)
    const int32_t *counts, *kpn;
    SalienceParams sp;
    uint8_t *salience;
    float *label_acc;
};
__global__ __launch_bounds__(256) void salience_levels_multi_kernel(const MultiArgs<SalienceGroup> m, int M) {
    int b, t;
    const SalienceGroup &a = multi_locate(m, b, t);
    salience_levels_body(a.counts, a.kpn, M, a.sp, a.salience, a.label_acc, b);
}

// a10 as its own entry: intra_predict (cpp_modules.cpp:248-285)
template <class L = uint8_t>
__global__ __launch_bounds__(256) void intra_predict_kernel(const L *__restrict__ seg, const float *__restrict__ model,
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
