// fps_kernels.h -- exact tile-pruned farthest point sampling (a6) and the ground-mask kernel that performs
// its first pass (a3+a5 + FPS pass 1); included by rpcc_hip.hip.
//
// Tiles hold 256 points, four CONSECUTIVE points per lane of one wavefront (16-byte loads):
//   range image:  tile t = (tr, tc) covers rows 8*tr .. 8*tr+7, columns 32*tc .. 32*tc+31 (compact in 3-D);
//                 lane l, element e  ->  row 8*tr + (l >> 3), column 32*tc + 4*(l & 7) + e
//   point list:   tile t covers indices 256*t ..; lane l, element e -> 256*t + 4*l + e
// In both layouts (lane, element) in lexicographic order is increasing point index, which is what the
// lowest-index tie rule of the arg-max needs.  With W % 4 == 0 (N % 4 == 0 for lists) a lane's four points are one
// 16-byte load of the range image, one of temp and three of the [P,3] ray table (VEC).  A range image whose width is no multiple
// of four (Velodyne 32E: 2250 columns) keeps the 16-byte loads at 4-byte alignment (EDGE: global_load_dwordx4 needs dword
// alignment only); its row ends cut a lane's quad short, and those lanes read and write their valid elements one by one.
// Point lists that fit neither are read with scalar loads.
//
// Per tile the workgroup keeps in LDS (FpsLds, three float4): the bounding box of the tile's candidates, the
// tile's current maximum of temp with its (lowest) index, and that point's coordinates.  For a new centre
// c a tile can only change if some point is closer to c than its temp, i.e. only if
//     bound(c, box) < tile_max,   bound = ((bx*bx)+(by*by))+(bz*bz),  b* = per-axis gap to the box.
// bound is evaluated with the SAME fp32 operation sequence as the point distance on per-axis gaps that
// are <= every candidate's |d*| (rounding is monotone), so bound <= computed distance of every candidate
// and skipping is bit-exact, not approximate (DESIGN.md "FPS").  Everything else -- min with temp, strict
// '>' arg-max with lowest-index ties -- is the brute-force definition.
//
// The origin class (range images).  Empty pixels back-project to (0,0,0) and ARE candidates of the reference
// (utils/segment_utils.py:119-120: their vertical residual is |d| / divisor > threshold), about 15-20 % of a sweep,
// scattered over every tile.  Kept in the tiles they would stretch every bounding box to the sensor and defeat the
// pruning.  They all have the same coordinates, hence the same distance to every centre and the same temp: the
// kernel carries them as ONE scalar (t_org) with the index of the first of them, leaves them out of the tile boxes
// and maxima, and lets the scalar compete in the arg-max with its (value, index) key.  Identical selections; the
// empty pixels' temp entries are brought up to date at the end when the caller reads temp (finalize_temp).
#pragma once

#define FPS_TAB_ROWS 12  // three float4 per tile: (lo.xyz, tmax) (hi.xyz, targ) (cx.xyz, tile origin)
#define FPS_TILE 256
#define FPS_TROWS 8
// RPCC_INFO (int32 per frame in `info`) is defined in rpcc_hip.hip

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

struct FpsLds {
    float4 *lo4;   // [T] (lo0, lo1, lo2, tmax)
    float4 *hi4;   // [T] (hi0, hi1, hi2, bits of targ)
    float4 *cx4;   // [T] (cx0, cx1, cx2, bits of torg: first pixel of the tile (22 bits) | valid columns - 1 (5 bits) << 22 | valid rows - 1 (3 bits) << 27)
    uint16_t *work;
    __device__ FpsLds(unsigned char *base, int T) {
        lo4 = reinterpret_cast<float4 *>(base);
        hi4 = lo4 + T;
        cx4 = hi4 + T;
        work = reinterpret_cast<uint16_t *>(cx4 + T);
    }
};
static inline size_t fps_tiled_lds_bytes(int T) { return (size_t)T * 50 + 64; }
#define FPS_TILED_MAX_TILES 3200  // 50 B/tile must fit the 160 KiB LDS of one CU

// One lane's share of a tile: four consecutive points.
struct FpsQuad {
    float r[4];     // ranges (RANGE)
    float tp[4];    // temp as loaded
    float t[12];    // RANGE: rays (tx,ty,tz) x 4; else xyz x 4
    int p0;         // index of element 0 (clamped to 0 for a lane outside the frame)
    int nval;       // valid elements: e < nval
};

__device__ __forceinline__ void fps_quad_xyz(const FpsQuad &q, bool range, float (&x)[4], float (&y)[4], float (&z)[4]) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (range) { x[e] = q.r[e] * q.t[3 * e]; y[e] = q.r[e] * q.t[3 * e + 1]; z[e] = q.r[e] * q.t[3 * e + 2]; }
        else { x[e] = q.t[3 * e]; y[e] = q.t[3 * e + 1]; z[e] = q.t[3 * e + 2]; }
    }
}

// loads of one lane's quad: src = range image (RANGE) or xyz list; rays = [P,3] table (RANGE)
// SOA (RANGE, VEC): rays = the planar copy [3][n_plane] of the table -- the three 16-byte loads share the offset of the range /
// temp loads (no multiplication by 12) and return the four pixels' x, y, z as register pairs (no re-packing for packed fp32)
template <bool RANGE, bool VEC, bool SOA = false, bool EDGE = false>
__device__ __forceinline__ void fps_quad_load(const float *__restrict__ src, const float *__restrict__ rays,
                                              const float *__restrict__ temp, FpsQuad &q, int n_plane = 0) {
    const uint32_t p0 = (uint32_t)q.p0;
    // (EDGE: the same 16-byte loads, 4-byte aligned: ld_quad<true>)
    if (VEC && EDGE && q.nval != 4) {   // a quad cut short by the row end (or a lane outside the frame): element loads, clamped
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t p = p0 + (uint32_t)(e < q.nval ? e : 0);
            q.tp[e] = ld_f32(temp, p * 4u);
            if (SOA) { q.t[3 * e] = ld_f32(rays, p * 4u); q.t[3 * e + 1] = ld_f32(rays + n_plane, p * 4u); q.t[3 * e + 2] = ld_f32(rays + 2 * (size_t)n_plane, p * 4u); }
            else { const float *tb = RANGE ? rays : src; q.t[3 * e] = ld_f32(tb, p * 12u); q.t[3 * e + 1] = ld_f32(tb, p * 12u + 4u); q.t[3 * e + 2] = ld_f32(tb, p * 12u + 8u); }
            if (RANGE) q.r[e] = ld_f32(src, p * 4u);
        }
        return;
    }
    if (VEC && RANGE && SOA) {
        const float4 tp = ld_quad<EDGE>(temp, p0 * 4u);
        q.tp[0] = tp.x; q.tp[1] = tp.y; q.tp[2] = tp.z; q.tp[3] = tp.w;
        const float4 a = ld_quad<EDGE>(rays, p0 * 4u);
        const float4 b = ld_quad<EDGE>(rays + n_plane, p0 * 4u);
        const float4 c = ld_quad<EDGE>(rays + 2 * (size_t)n_plane, p0 * 4u);
        q.t[0] = a.x; q.t[3] = a.y; q.t[6] = a.z; q.t[9] = a.w; q.t[1] = b.x; q.t[4] = b.y; q.t[7] = b.z; q.t[10] = b.w;
        q.t[2] = c.x; q.t[5] = c.y; q.t[8] = c.z; q.t[11] = c.w;
        const float4 r = ld_quad<EDGE>(src, p0 * 4u);
        q.r[0] = r.x; q.r[1] = r.y; q.r[2] = r.z; q.r[3] = r.w;
    } else if (VEC) {  // 16-byte loads at wave-uniform base + 32-bit byte offset
        const float4 tp = ld_quad<EDGE>(temp, p0 * 4u);
        q.tp[0] = tp.x; q.tp[1] = tp.y; q.tp[2] = tp.z; q.tp[3] = tp.w;
        const float *tb = RANGE ? rays : src;
        const float4 a = ld_quad<EDGE>(tb, p0 * 12u);
        const float4 b = ld_quad<EDGE>(tb, p0 * 12u + 16u);
        const float4 c = ld_quad<EDGE>(tb, p0 * 12u + 32u);
        q.t[0] = a.x; q.t[1] = a.y; q.t[2] = a.z; q.t[3] = a.w; q.t[4] = b.x; q.t[5] = b.y; q.t[6] = b.z; q.t[7] = b.w;
        q.t[8] = c.x; q.t[9] = c.y; q.t[10] = c.z; q.t[11] = c.w;
        if (RANGE) {
            const float4 r = ld_quad<EDGE>(src, p0 * 4u);
            q.r[0] = r.x; q.r[1] = r.y; q.r[2] = r.z; q.r[3] = r.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t p = p0 + (uint32_t)(e < q.nval ? e : 0);  // unconditional loads on clamped indices
            q.tp[e] = ld_f32(temp, p * 4u);
            const float *tb = RANGE ? rays : src;
            q.t[3 * e] = ld_f32(tb, p * 12u); q.t[3 * e + 1] = ld_f32(tb, p * 12u + 4u); q.t[3 * e + 2] = ld_f32(tb, p * 12u + 8u);
            if (RANGE) q.r[e] = ld_f32(src, p * 4u);
        }
    }
}

// order-preserving key of a temp value: non-negative floats order as integers, so flipping the sign bit gives an unsigned
// key >= FPS_KEY_MIN for every candidate value and a key below it for the negative "not a candidate" marker
#define FPS_KEY_MIN 0x80000000u
__device__ __forceinline__ uint32_t fps_val_key(float v) { return f2u(v) ^ 0x80000000u; }
__device__ __forceinline__ float fps_key_val(uint32_t k) { return k >= FPS_KEY_MIN ? u2f(k ^ 0x80000000u) : -1.0f; }

// arg-max of a tile: largest key, lowest (lane, element) among equals -> value, coordinates, point index
__device__ __forceinline__ void fps_tile_argmax(const float (&x)[4], const float (&y)[4], const float (&z)[4],
                                                const uint32_t (&key)[4], int p0, float &wt, float &wx, float &wy, float &wz,
                                                uint32_t &widx) {
    uint32_t m = key[0];
    int em = 0;
    float sx = x[0], sy = y[0], sz = z[0];
#pragma unroll
    for (int e = 1; e < 4; e++) {
        const bool gt = key[e] > m;
        m = gt ? key[e] : m; em = gt ? e : em;
        sx = gt ? x[e] : sx; sy = gt ? y[e] : sy; sz = gt ? z[e] : sz;
    }
    const uint32_t vmax = dpp_max_u32(m);
    const unsigned long long bm = __ballot(m == vmax);
    const int wl = (int)__ffsll((long long)bm) - 1;
    wt = fps_key_val(vmax);
    wx = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(sx), wl));
    wy = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(sy), wl));
    wz = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(sz), wl));
    widx = (uint32_t)__builtin_amdgcn_readlane(p0 + em, wl);
    if (vmax < FPS_KEY_MIN) { wx = wy = wz = 0.0f; widx = 0u; }
}
__device__ __forceinline__ void fps_tile_box(const float (&x)[4], const float (&y)[4], const float (&z)[4],
                                             const bool (&cand)[4], float (&lo)[3], float (&hi)[3]) {
    const float inf = __builtin_inff();
    lo[0] = lo[1] = lo[2] = inf;
    hi[0] = hi[1] = hi[2] = -inf;
    // non-candidates enter as quiet NaNs, which v_min3 / v_max3 skip (rpcc_device.h): one select per coordinate, two points per instruction
    const float qnan = u2f(0x7FC00000u);
    float bx[4], by[4], bz[4];
#pragma unroll
    for (int e = 0; e < 4; e++) { bx[e] = cand[e] ? x[e] : qnan; by[e] = cand[e] ? y[e] : qnan; bz[e] = cand[e] ? z[e] : qnan; }
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        lo[0] = fmin3_raw(lo[0], bx[e], bx[e + 1]); hi[0] = fmax3_raw(hi[0], bx[e], bx[e + 1]);
        lo[1] = fmin3_raw(lo[1], by[e], by[e + 1]); hi[1] = fmax3_raw(hi[1], by[e], by[e + 1]);
        lo[2] = fmin3_raw(lo[2], bz[e], bz[e + 1]); hi[2] = fmax3_raw(hi[2], bz[e], bz[e + 1]);
    }
    dpp_box6(lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
}

// RANGE point k = ri[k] * rays[k]; else AoS xyz[k*3..]
template <bool RANGE>
__device__ __forceinline__ void fps_load_point(const float *__restrict__ src, const float *__restrict__ rays, int k,
                                               float &x, float &y, float &z) {
    if (RANGE) {
        const float r = ld_f32(src, (uint32_t)k * 4u);
        x = r * ld_f32(rays, (uint32_t)k * 12u); y = r * ld_f32(rays, (uint32_t)k * 12u + 4u); z = r * ld_f32(rays, (uint32_t)k * 12u + 8u);
    } else {
        x = src[3 * (int64_t)k]; y = src[3 * (int64_t)k + 1]; z = src[3 * (int64_t)k + 2];
    }
}

#define FPS_THREADS 1024

// Threads of the tile-pruned FPS workgroup (one workgroup per frame): FPS_TT_BATCH for batches that fill the chip
// (several batches in flight share the CUs), FPS_TT_SMALL for small batches (nothing to co-schedule with).
// Measurements: DESIGN.md section 5.
#define FPS_TT_BATCH 512
#define FPS_TT_SMALL 1024
#define FPS_GROUP 2
#define FPS_FLAG_FINALIZE_TEMP 1  // write the origin class's value back to the empty pixels' temp entries at the end
#define FPS_FLAG_PROBED 2         // point lists: fps_list_probe_kernel has marked the frames the one-pass-per-centre kernel takes (out_idx[b][0] < 0)

template <bool RANGE, bool VEC, int FPS_TT>
__global__ __launch_bounds__(FPS_TT) void fps_tiled_kernel(const float *__restrict__ src, const float *__restrict__ rays,
                                                           float *__restrict__ temp, const int32_t *__restrict__ info,
                                                           FpsTiling g, int M, int flags, int32_t *__restrict__ out_idx,
                                                           float *__restrict__ out_cen,
                                                           const float *__restrict__ tiletab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fps_smem[];
    __shared__ unsigned long long red[FPS_TT / 64];
    __shared__ int redt[FPS_TT / 64];
    __shared__ int wcount, s_viol;
    const int T = g.T, N = g.N;
    FpsLds L(fps_smem, T);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    src += (int64_t)b * N * (RANGE ? 1 : 3);
    temp += (int64_t)b * N;
    out_idx += (int64_t)b * M;
    if (out_cen) out_cen += (int64_t)b * M * 3;
    if (M <= 0) return;
    if (!RANGE && (flags & FPS_FLAG_PROBED) && out_idx[0] < 0) return;   // (workgroup-uniform) this list goes to fps_xyz_kernel

    int old = 0;
    if (RANGE) { old = info[RPCC_INFO * b + 1]; if (old >= N) old = 0; }
    float c0, c1, c2;
    fps_load_point<RANGE>(src, rays, old, c0, c1, c2);
    if (tid == 0) {
        out_idx[0] = old;
        if (out_cen) { out_cen[0] = c0; out_cen[1] = c1; out_cen[2] = c2; }
        wcount = 0;
        s_viol = 0;
    }
    // origin class: every empty pixel that is a candidate (see the header)
    int org_idx = N;
    if (RANGE) { org_idx = info[RPCC_INFO * b + 4]; if (org_idx < 0 || org_idx > N) org_idx = N; }
    bool org_on = RANGE && org_idx < N;
    float ox = 0.0f, oy = 0.0f, oz = 0.0f, t_org = -1.0f;
    if (org_on) {
        fps_load_point<RANGE>(src, rays, org_idx, ox, oy, oz);   // 0 * ray: signed zeros
        t_org = ld_f32(temp, (uint32_t)org_idx * 4u);
        if (!(t_org >= 0.0f) || ld_f32(src, (uint32_t)org_idx * 4u) != 0.0f) { org_on = false; t_org = -1.0f; }
    }
    const float t_org0 = t_org;

    for (int t = tid; t < T; t += FPS_TT) {
        uint32_t org = 0u;
        if (RANGE) {
            const int tr = t / g.tcols, tc = t - tr * g.tcols;
            const int ncol = min(32, g.W - 32 * tc), nrow = min(FPS_TROWS, g.H - FPS_TROWS * tr);
            org = (uint32_t)(FPS_TROWS * tr * g.W + 32 * tc) | ((uint32_t)(ncol - 1) << 22) | ((uint32_t)(nrow - 1) << 27);
        }
        L.cx4[t].w = u2f(org);
    }
    __syncthreads();
    const int lrow = lane >> 3, lcol = 4 * (lane & 7);
    auto locate = [&](int t, FpsQuad &q) {
        if (RANGE) {
            const uint32_t org = f2u(L.cx4[t].w);
            const int ncol = (int)((org >> 22) & 31u) + 1, nrow = (int)(org >> 27) + 1;
            const int nv = lrow < nrow ? min(max(ncol - lcol, 0), 4) : 0;
            q.nval = nv;
            q.p0 = nv > 0 ? (int)(org & 0x3FFFFFu) + lrow * g.W + lcol : 0;
        } else {
            const int p = t * FPS_TILE + 4 * lane;
            q.nval = min(max(N - p, 0), 4);
            q.p0 = q.nval > 0 ? p : 0;
        }
    };
    // distance update against the current centre, tile maximum, (optionally) bounding box and the check that the
    // origin class is what the header says (all empty pixels candidates with one common temp)
    auto compute_tile = [&](int t, const FpsQuad &q, bool with_box) {
        float x[4], y[4], z[4], nt[4];
        uint32_t key[4];
        bool cand[4], ch = false, viol = false;
        fps_quad_xyz(q, RANGE, x, y, z);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float tp = e < q.nval ? q.tp[e] : -1.0f;
            if (RANGE && org_on && q.r[e] == 0.0f) {   // member of the origin class: carried by t_org
                if (with_box) viol |= e < q.nval && tp != t_org0;
                tp = -1.0f;
            }
            cand[e] = tp >= 0.0f;
            const float dx = x[e] - c0, dy = y[e] - c1, dz = z[e] - c2;
            const float d = (dx * dx + dy * dy) + dz * dz;  // sampling_gpu.cu:64, un-fused
            nt[e] = d < tp ? d : tp;  // == fminf(d, tp): a NaN distance keeps tp, tp itself is never NaN
            key[e] = fps_val_key(nt[e]);
            const bool c = nt[e] != tp;
            ch |= c;
            if (!VEC && c) st_f32(temp, (uint32_t)(q.p0 + e) * 4u, nt[e]);
            if (!c) nt[e] = q.tp[e];   // value to write back for an unchanged element
        }
        if (VEC && ch) st_at(reinterpret_cast<float4 *>(temp), (uint32_t)q.p0 * 4u, make_float4(nt[0], nt[1], nt[2], nt[3]));
        if (with_box && __ballot(viol) != 0ull && lane == 0) s_viol = 1;
        // nothing changed in this tile: its table entry (maximum, arg, coordinates) is still exact
        if (!with_box && __ballot(ch) == 0ull) return;
        if (with_box) {
            float lo[3], hi[3];
            fps_tile_box(x, y, z, cand, lo, hi);
            if (lane == 0) { L.lo4[t].x = lo[0]; L.lo4[t].y = lo[1]; L.lo4[t].z = lo[2]; L.hi4[t].x = hi[0]; L.hi4[t].y = hi[1]; L.hi4[t].z = hi[2]; }
        }
        float wt, wx, wy, wz;
        uint32_t widx;
        fps_tile_argmax(x, y, z, key, q.p0, wt, wx, wy, wz, widx);
        if (lane == 0) { L.lo4[t].w = wt; L.hi4[t].w = u2f(widx); L.cx4[t].x = wx; L.cx4[t].y = wy; L.cx4[t].z = wz; }
    };
    auto update_origin = [&]() {
        if (org_on) {
            const float dx = ox - c0, dy = oy - c1, dz = oz - c2;
            const float d = (dx * dx + dy * dy) + dz * dz;
            t_org = d < t_org ? d : t_org;
        }
    };

    // arg-max over the tile table and the origin class -> next centre (index and coordinates)
    auto select_next = [&]() {
        uint32_t hi = 0u, ix = 0xFFFFFFFFu;  // orderable value, index
        int bt = 0;
        for (int t = tid; t < T; t += FPS_TT) {
            const uint32_t h = fps_val_key(L.lo4[t].w), i = f2u(L.hi4[t].w);
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
        const uint32_t okey = org_on ? fps_val_key(t_org) : 0u;
        if (okey >= FPS_KEY_MIN && (okey > vmax || (okey == vmax && (uint32_t)org_idx < imin))) {
            old = org_idx;
            c0 = ox; c1 = oy; c2 = oz;
        } else if (vmax < FPS_KEY_MIN || imin == 0xFFFFFFFFu) {  // no candidate anywhere: keep indices defined (the reference would fail)
            old = 0;
            fps_load_point<RANGE>(src, rays, 0, c0, c1, c2);
        } else {
            old = (int)imin;
            const float4 cc = L.cx4[t];
            c0 = cc.x; c1 = cc.y; c2 = cc.z;
        }
    };

    constexpr int NW = FPS_TT / 64, GROUP = FPS_GROUP;  // tiles per wavefront in flight
    // first centre: every tile is visited once (also builds the boxes) -- unless ground_mask already did
    // that pass and left the tile table (info[b][3] == 1)
    const bool have_tab = RANGE && tiletab != nullptr && info[RPCC_INFO * b + 3] == 1;
    if (M > 1 && have_tab) {
        const float4 *t4 = reinterpret_cast<const float4 *>(tiletab + (int64_t)b * FPS_TAB_ROWS * T);
        float4 *d4 = reinterpret_cast<float4 *>(fps_smem);
        for (int i = tid; i < 2 * T; i += FPS_TT) d4[i] = t4[i];
        for (int i = tid; i < T; i += FPS_TT) { const float4 v = t4[2 * T + i]; L.cx4[i].x = v.x; L.cx4[i].y = v.y; L.cx4[i].z = v.z; }
        update_origin();  // (idempotent: temp of the empty pixels already holds the first centre's distance)
        __syncthreads();
    } else if (M > 1) {
        for (int pass = 0; pass < 2; pass++) {
            for (int t = wave; t < T; t += NW * GROUP) {
                FpsQuad q[GROUP];
#pragma unroll
                for (int gi = 0; gi < GROUP; gi++) {
                    locate(min(t + gi * NW, T - 1), q[gi]);
                    fps_quad_load<RANGE, VEC>(src, rays, temp, q[gi]);
                }
#pragma unroll
                for (int gi = 0; gi < GROUP; gi++) if (t + gi * NW < T) compute_tile(t + gi * NW, q[gi], true);
            }
            __syncthreads();
            // the origin class must be uniform (it is when temp comes from rpcc_ground_mask); a caller-made temp that
            // treats the empty pixels individually is handled by a second pass without the class
            if (!(org_on && s_viol)) break;
            org_on = false; t_org = -1.0f;
            __threadfence_block();
        }
        update_origin();
    }
    if (M > 1) {
        select_next();
        if (tid == 0) { out_idx[1] = old; if (out_cen) { out_cen[3] = c0; out_cen[4] = c1; out_cen[5] = c2; } }
    }
    for (int j = 2; j < M; j++) {
        // tile test against the new centre; active tiles go to the work list
        for (int t = tid; t < T; t += FPS_TT) {
            const float4 lo = L.lo4[t], hi = L.hi4[t];
            const float g0 = fmaxf(fmaxf(lo.x - c0, c0 - hi.x), 0.0f);
            const float g1 = fmaxf(fmaxf(lo.y - c1, c1 - hi.y), 0.0f);
            const float g2 = fmaxf(fmaxf(lo.z - c2, c2 - hi.z), 0.0f);
            const float bound = (g0 * g0 + g1 * g1) + g2 * g2;
            const bool act = bound < lo.w;
            const unsigned long long m = __ballot(act);
            if (m) {
                int base = 0;
                if (lane == (int)__ffsll((long long)m) - 1) base = atomicAdd(&wcount, __popcll(m));
                base = __shfl(base, (int)__ffsll((long long)m) - 1, 64);
                if (act) L.work[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)t;
            }
        }
        update_origin();
        __syncthreads();
        const int n = wcount;
        for (int e = wave; e < n; e += NW * GROUP) {
            FpsQuad q[GROUP];
            int tt[GROUP];
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) {
                const int ee = e + gi * NW;
                tt[gi] = (int)L.work[ee < n ? ee : n - 1];
                locate(tt[gi], q[gi]);
                fps_quad_load<RANGE, VEC>(src, rays, temp, q[gi]);   // unconditional (a repeated tile for the tail is harmless and unused)
            }
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) if (e + gi * NW < n) compute_tile(tt[gi], q[gi], false);
        }
        __syncthreads();
        if (tid == 0) wcount = 0;
        select_next();
        if (lane == 0 && wave == (j & (NW - 1))) { out_idx[j] = old; if (out_cen) { out_cen[3 * j] = c0; out_cen[3 * j + 1] = c1; out_cen[3 * j + 2] = c2; } }
    }
    if (RANGE && org_on && (flags & FPS_FLAG_FINALIZE_TEMP)) {
        // the empty pixels' temp entries were not touched while the class was carried as a scalar
        for (int p = tid; p < N; p += FPS_TT) {
            const float r = ld_f32(src, (uint32_t)p * 4u), tv = ld_f32(temp, (uint32_t)p * 4u);
            if (r == 0.0f && tv >= 0.0f && tv != t_org) st_f32(temp, (uint32_t)p * 4u, t_org);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Register-table form (the default whenever the frame has at most 64 tiles per wavefront of the workgroup).
// The per-iteration chain of the kernel above is  test -> work list -> barrier -> tile loads -> update -> barrier ->
// table scan -> barrier: three barriers and five dependent LDS round trips around ~1200 cycles of load latency.  Here
// every tile belongs to ONE lane of ONE wavefront for the whole launch: its box, maximum, arg and coordinates live in that
// lane's registers, the wavefront tests its own 64 tiles (no LDS), updates the ones the new centre can change (their
// results are wave-uniform scalars, written back into the owner lane), reduces its own maxima with DPP and publishes one
// candidate; ONE barrier later every wavefront picks the winner among the NW candidates.  The candidates are double
// buffered by iteration parity, so a fast wavefront may run ahead into its next test + loads while the others still read.
// Tiles are dealt to the wavefronts round-robin in an order that is skewed from tile-row to tile-row, so the tiles around
// a new centre (horizontal and vertical neighbours) belong to different wavefronts.  Same arithmetic, same results.
// ------------------------------------------------------------------------------------------------------------------
struct FpsTileOut { float lo[3], hi[3], wt, wx, wy, wz; uint32_t widx; };

// update of one tile against centre (c0,c1,c2): returns true (wave-uniform) when its table entry changed (always with
// with_box); the new entry comes back in `o`.  viol: an empty pixel is not what the origin class assumes (first pass only).
template <bool RANGE, bool VEC, bool EDGE = false>
__device__ __forceinline__ bool fps_tile_update(const FpsQuad &q, bool org_on, float t_org0, float c0, float c1, float c2,
                                                float *__restrict__ temp, bool with_box, FpsTileOut &o, bool &viol) {
    float x[4], y[4], z[4], nt[4];
    uint32_t key[4];
    bool cand[4], ch = false;
    fps_quad_xyz(q, RANGE, x, y, z);
    const bool lane_ok = q.nval > 0;   // VEC without EDGE: a lane's four elements are inside the frame together
    constexpr bool QUAD = VEC && !EDGE;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float tp = (QUAD ? lane_ok : e < q.nval) ? q.tp[e] : -1.0f;
        if (RANGE && org_on && q.r[e] == 0.0f) {   // member of the origin class: carried by t_org
            if (with_box) viol |= (QUAD ? lane_ok : e < q.nval) && tp != t_org0;
            tp = -1.0f;
        }
        cand[e] = tp >= 0.0f;
        const float dx = x[e] - c0, dy = y[e] - c1, dz = z[e] - c2;
        const float d = (dx * dx + dy * dy) + dz * dz;  // sampling_gpu.cu:64, un-fused
        nt[e] = d < tp ? d : tp;  // == fminf(d, tp): a NaN distance keeps tp, tp itself is never NaN
        key[e] = fps_val_key(nt[e]);
        const bool c = nt[e] != tp;
        ch |= c;
        if ((!VEC || (EDGE && q.nval != 4)) && c) st_f32(temp, (uint32_t)(q.p0 + e) * 4u, nt[e]);   // (c implies a valid element)
        if (!c) nt[e] = q.tp[e];   // value to write back for an unchanged element
    }
    if (QUAD && ch) st_at(reinterpret_cast<float4 *>(temp), (uint32_t)q.p0 * 4u, make_float4(nt[0], nt[1], nt[2], nt[3]));
    if (VEC && EDGE && ch && q.nval == 4) st_quad<true>(temp, (uint32_t)q.p0 * 4u, nt[0], nt[1], nt[2], nt[3]);
    if (!with_box && __ballot(ch) == 0ull) return false;
    // (Leaving the arg-max out when the point that holds the tile's maximum did not change -- the entry is then provably what it
    // was -- was measured in round 3: the holder has the largest temp of the tile, so it is the FIRST point a centre in reach
    // lowers; the test fired rarely, cost four compares per visit, and the kernel got 2 % slower.)
    if (with_box) fps_tile_box(x, y, z, cand, o.lo, o.hi);
    fps_tile_argmax(x, y, z, key, q.p0, o.wt, o.wx, o.wy, o.wz, o.widx);
    return true;
}

// TPL: tiles per lane (1; 2 for images with more tiles than the workgroup has lanes -- 80 x 2000, 128 x 2048: slot s of lane l is position (l + 64 s) NW + wave).
// (Round 5 ran the headline with 4 / 2 / 1 wavefronts per frame and 2 / 4 / 8 tiles per lane: every step towards fewer wavefronts is slower, profiles/HISTORY.md.)
template <bool RANGE, bool VEC, int FPS_TT, bool SOA, bool EDGE = false, int TPL = 1>
__device__ __forceinline__ void fps_regtab_body(const float *__restrict__ src, const float *__restrict__ rays,
                                                float *__restrict__ temp, const int32_t *__restrict__ info,
                                                FpsTiling g, int M, int flags, int32_t *__restrict__ out_idx,
                                                float *__restrict__ out_cen, const float *__restrict__ tiletab,
                                                const float *__restrict__ rays_soa, const int b) {   // b: the workgroup's frame
    constexpr int NW = FPS_TT / 64;
    TRACE_FPS_DECLS();      // (developer trace hooks: empty unless the library is built with -DRPCC_DEVTRACE, rpcc_trace.h)
    TRACE_FPS_WG(0);
    __shared__ uint2 slot_k[2][NW];    // candidate of a wavefront: (value key, point index)
    __shared__ float4 slot_c[2][NW];   //                          its coordinates
    __shared__ int s_viol;
    const int T = g.T, N = g.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    src += (int64_t)b * N * (RANGE ? 1 : 3);
    temp += (int64_t)b * N;
    out_idx += (int64_t)b * M;
    if (out_cen) out_cen += (int64_t)b * M * 3;
    if (M <= 0) return;
    if (!RANGE && (flags & FPS_FLAG_PROBED) && out_idx[0] < 0) return;   // (workgroup-uniform) this list goes to fps_xyz_kernel

    int old = 0;
    if (RANGE) { old = info[RPCC_INFO * b + 1]; if (old >= N) old = 0; }
    float c0, c1, c2;
    fps_load_point<RANGE>(src, rays, old, c0, c1, c2);
    if (tid == 0) {
        out_idx[0] = old;
        if (out_cen) { out_cen[0] = c0; out_cen[1] = c1; out_cen[2] = c2; }
        s_viol = 0;
    }
    // origin class: every empty pixel that is a candidate (header of this file)
    int org_idx = N;
    if (RANGE) { org_idx = info[RPCC_INFO * b + 4]; if (org_idx < 0 || org_idx > N) org_idx = N; }
    bool org_on = RANGE && org_idx < N;
    float t_org = -1.0f;   // (the class's coordinates are 0 * ray = signed zeros: its distance to a centre is the centre's squared norm)
    if (org_on) {
        t_org = ld_f32(temp, (uint32_t)org_idx * 4u);
        if (!(t_org >= 0.0f) || ld_f32(src, (uint32_t)org_idx * 4u) != 0.0f) { org_on = false; t_org = -1.0f; }
    }
    const float t_org0 = t_org;

    // this lane's tiles: position pos = (lane + 64 s) * NW + wave in the skewed order -> tile id, packed origin
    bool have[TPL];
    int my_t[TPL];
    uint32_t my_org[TPL];
    const float inf = __builtin_inff();
    float lo0[TPL], lo1[TPL], lo2[TPL], hi0[TPL], hi1[TPL], hi2[TPL], tmax[TPL], cx[TPL], cy[TPL], cz[TPL];
    uint32_t targ[TPL];
    unsigned long long have_m[TPL];
#pragma unroll
    for (int s = 0; s < TPL; s++) {
        const int pos = (lane + 64 * s) * NW + wave;
        have[s] = pos < T;
        my_t[s] = 0; my_org[s] = 0u;
        if (have[s]) {
            if (RANGE) {
                const int tr = pos / g.tcols, q = pos - tr * g.tcols;
                int tc = q - (3 * tr) % g.tcols;
                if (tc < 0) tc += g.tcols;
                my_t[s] = tr * g.tcols + tc;
                const int ncol = min(32, g.W - 32 * tc), nrow = min(FPS_TROWS, g.H - FPS_TROWS * tr);
                my_org[s] = (uint32_t)(FPS_TROWS * tr * g.W + 32 * tc) | ((uint32_t)(ncol - 1) << 22) | ((uint32_t)(nrow - 1) << 27);
            } else {
                my_t[s] = pos;
            }
        }
        lo0[s] = lo1[s] = lo2[s] = inf; hi0[s] = hi1[s] = hi2[s] = -inf; tmax[s] = -1.0f; cx[s] = cy[s] = cz[s] = 0.0f; targ[s] = 0u;
        have_m[s] = __ballot(have[s]);
    }
    const int lrow = lane >> 3, lcol = 4 * (lane & 7);
    // the quad of this lane in the tile owned by slot s of lane `l` (wave-uniform l, s)
    auto locate = [&](int l, int s, FpsQuad &q) {
        if (RANGE) {
            uint32_t org = (uint32_t)__builtin_amdgcn_readlane((int)my_org[0], l);
#pragma unroll
            for (int t = 1; t < TPL; t++) if (s == t) org = (uint32_t)__builtin_amdgcn_readlane((int)my_org[t], l);   // (s is wave-uniform)
            const int ncol = (int)((org >> 22) & 31u) + 1, nrow = (int)(org >> 27) + 1;
            const int nv = lrow < nrow ? min(max(ncol - lcol, 0), 4) : 0;
            q.nval = nv;
            q.p0 = nv > 0 ? (int)(org & 0x3FFFFFu) + lrow * g.W + lcol : 0;
        } else {
            int tl = __builtin_amdgcn_readlane(my_t[0], l);
#pragma unroll
            for (int t = 1; t < TPL; t++) if (s == t) tl = __builtin_amdgcn_readlane(my_t[t], l);
            const int p = tl * FPS_TILE + 4 * lane;
            q.nval = min(max(N - p, 0), 4);
            q.p0 = q.nval > 0 ? p : 0;
        }
    };
    static_assert(TPL >= 1 && TPL <= 8, "tiles per lane");
    auto store_entry = [&](int l, int s, const FpsTileOut &o, bool with_box) {
#pragma unroll
        for (int t = 0; t < TPL; t++) {
            if (lane == l && s == t) {
                if (with_box) { lo0[t] = o.lo[0]; lo1[t] = o.lo[1]; lo2[t] = o.lo[2]; hi0[t] = o.hi[0]; hi1[t] = o.hi[1]; hi2[t] = o.hi[2]; }
                tmax[t] = o.wt; targ[t] = o.widx; cx[t] = o.wx; cy[t] = o.wy; cz[t] = o.wz;
            }
        }
    };
    // visits the tiles of the lanes in mask m, FPS_VISIT = 2 at a time (all loads of a pair are in flight before either is used).
    // REGISTERS: the kernel must stay at or below 96 VGPRs (FPS_VGPR_ATTR).  Round 3 found that every variant that was "faster
    // alone, slower with batches in flight" (the second tile loaded only when present, the planar ray table, round 2's
    // software-pipelined visits) had crossed that line by two registers; capped, they are faster in flight too.
    // Larger groups -- three or four tiles' loads in flight, so that the wavefront that owns most of the changed tiles pays one
    // memory latency instead of two -- were measured in round 3 (with and without loads for absent tiles): 292 / 298 us alone
    // against 273 us (139 instead of 96 VGPRs), and 7-11 % fewer frames/s with batches in flight.  (Round 2 had measured
    // software-pipelined visits: 7 % faster alone, 3 % slower with batches in flight.)
    constexpr int FPS_VISIT = 2;          // tiles per visit round
    constexpr int FPS_VISIT_UNCOND = 1;   // tiles of a round loaded unconditionally; the second tile's loads are issued only when there is one
    // (mv[s]: the lanes whose slot-s tile is to be visited; slot 0's tiles first)
    auto visit = [&](const unsigned long long (&mv)[TPL], bool with_box) {
        bool viol = false;
        // (the slots' masks one after the other: `next` steps to the next slot that has tiles)
        unsigned long long m = mv[0];
        int cur = 0;
        auto next = [&]() {
#pragma unroll
            for (int t = 1; t < TPL; t++) if (m == 0ull && cur < t) { m = mv[t]; cur = t; }
        };
        if (TPL > 1) next();
        while (m) {
            int l[FPS_VISIT], sl[FPS_VISIT];
            bool on[FPS_VISIT];
            FpsQuad q[FPS_VISIT];
            TRACE_FPS_VISIT(0);
#pragma unroll
            for (int u = 0; u < FPS_VISIT; u++) {
                on[u] = m != 0ull;
                l[u] = on[u] ? (int)__ffsll((long long)m) - 1 : l[0];
                sl[u] = on[u] ? cur : sl[0];
                m &= m - 1ull;     // (0 & anything stays 0)
                if (TPL > 1) next();   // this slot's tiles are done: on to the next slot that has any
                if (u < FPS_VISIT_UNCOND || on[u]) {   // (wave-uniform) the first pair unconditionally, the rest only when there is a tile
                    locate(l[u], sl[u], q[u]);
                    fps_quad_load<RANGE, VEC, SOA, EDGE>(src, SOA ? rays_soa : rays, temp, q[u], N);
                }
            }
            TRACE_FPS_VISIT(1);
#pragma unroll
            for (int u = 0; u < FPS_VISIT; u++) {
                FpsTileOut o;
                if (on[u] && fps_tile_update<RANGE, VEC, EDGE>(q[u], org_on, t_org0, c0, c1, c2, temp, with_box, o, viol)) store_entry(l[u], sl[u], o, with_box);
            }
            TRACE_FPS_VISIT(2);
        }
        if (with_box && __ballot(viol) != 0ull && lane == 0) s_viol = 1;
    };
    auto update_origin = [&]() {
        if (org_on) {
            const float d = (c0 * c0 + c1 * c1) + c2 * c2;   // dx = (+-0) - c0 = -c0 exactly (a zero result's sign vanishes in the square)
            t_org = d < t_org ? d : t_org;
        }
    };
    int par = 0;
    // arg-max over all tiles and the origin class -> next centre (index and coordinates); one barrier
    auto select_next = [&]() {
        // (two tiles per lane: the lane's better one first -- larger key, lower index among equals)
        uint32_t key = have[0] ? fps_val_key(tmax[0]) : 0u, targ_b = targ[0];
        float cx_b = cx[0], cy_b = cy[0], cz_b = cz[0];
#pragma unroll
        for (int t = 1; t < TPL; t++) {
            const uint32_t k1 = have[t] ? fps_val_key(tmax[t]) : 0u;
            const bool sec = k1 > key || (k1 == key && targ[t] < targ_b);
            key = sec ? k1 : key; targ_b = sec ? targ[t] : targ_b;
            cx_b = sec ? cx[t] : cx_b; cy_b = sec ? cy[t] : cy_b; cz_b = sec ? cz[t] : cz_b;
        }
        uint32_t vmax = dpp_max_u32(key);
        // lowest index among the lanes that hold the maximum: almost always one lane, whose index is read directly (the second
        // reduction is 7 dependent DPP steps)
        unsigned long long mm = __ballot(key == vmax);
        uint32_t imin;
        if (__popcll(mm) == 1) {
            imin = (uint32_t)__builtin_amdgcn_readlane((int)targ_b, (int)__ffsll((long long)mm) - 1);
        } else {
            imin = dpp_min_u32(key == vmax ? targ_b : 0xFFFFFFFFu);
            mm = __ballot(key == vmax && targ_b == imin);
        }
        {
            const int wl = mm ? (int)__ffsll((long long)mm) - 1 : 0;
            const float wx = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(cx_b), wl));
            const float wy = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(cy_b), wl));
            const float wz = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(cz_b), wl));
            if (lane == 0) { slot_k[par][wave] = make_uint2(vmax, vmax >= FPS_KEY_MIN ? imin : 0xFFFFFFFFu); slot_c[par][wave] = make_float4(wx, wy, wz, 0.0f); }
        }
        TRACE_FPS_PHASE(3);
        __syncthreads();
        TRACE_FPS_PHASE(4);
        const uint2 kv = slot_k[par][lane % NW];
        const float4 cc = slot_c[par][lane % NW];
        par ^= 1;
        vmax = NW == 8 ? dpp_max8_u32(kv.x) : dpp_max_u32(kv.x);
        mm = __ballot(kv.x == vmax) & ((1ull << NW) - 1ull);   // the NW candidates repeat along the lanes: look at the first NW
        if (__popcll(mm) == 1) {
            imin = (uint32_t)__builtin_amdgcn_readlane((int)kv.y, (int)__ffsll((long long)mm) - 1);
        } else {
            imin = dpp_min_u32(kv.x == vmax ? kv.y : 0xFFFFFFFFu);
            mm = __ballot(kv.x == vmax && kv.y == imin);
        }
        const int wl = mm ? (int)__ffsll((long long)mm) - 1 : 0;
        const uint32_t okey = org_on ? fps_val_key(t_org) : 0u;
        if (okey >= FPS_KEY_MIN && (okey > vmax || (okey == vmax && (uint32_t)org_idx < imin))) {
            old = org_idx;
            fps_load_point<RANGE>(src, rays, org_idx, c0, c1, c2);   // 0 * ray: signed zeros (at most once per frame)
        } else if (vmax < FPS_KEY_MIN || imin == 0xFFFFFFFFu) {  // no candidate anywhere: keep indices defined (the reference would fail)
            old = 0;
            fps_load_point<RANGE>(src, rays, 0, c0, c1, c2);
        } else {
            old = (int)imin;
            c0 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(cc.x), wl));
            c1 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(cc.y), wl));
            c2 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(cc.z), wl));
        }
    };

    const bool have_tab = RANGE && tiletab != nullptr && info[RPCC_INFO * b + 3] == 1;
    if (M > 1 && have_tab) {   // the ground-mask kernel ran the first pass: this lane's entry
#pragma unroll
        for (int s = 0; s < TPL; s++) {
            if (have[s]) {
                const float4 *t4 = reinterpret_cast<const float4 *>(tiletab + (int64_t)b * FPS_TAB_ROWS * T);
                const float4 a = t4[my_t[s]], h = t4[T + my_t[s]], c = t4[2 * T + my_t[s]];
                lo0[s] = a.x; lo1[s] = a.y; lo2[s] = a.z; tmax[s] = a.w; hi0[s] = h.x; hi1[s] = h.y; hi2[s] = h.z; targ[s] = f2u(h.w);
                cx[s] = c.x; cy[s] = c.y; cz[s] = c.z;
            }
        }
        update_origin();  // (idempotent: temp of the empty pixels already holds the first centre's distance)
    } else if (M > 1) {
        __syncthreads();   // s_viol = 0 visible
        visit(have_m, true);
        __syncthreads();
        // the origin class must be uniform (it is when temp comes from rpcc_ground_mask); a caller-made temp that
        // treats the empty pixels individually is handled by a second pass without the class
        if (org_on && s_viol) {
            org_on = false; t_org = -1.0f;
            visit(have_m, true);
        }
        update_origin();
    }
    if (M > 1) {
        select_next();
        if (tid == 0) { out_idx[1] = old; if (out_cen) { out_cen[3] = c0; out_cen[4] = c1; out_cen[5] = c2; } }
    }
    TRACE_FPS_PHASE(0);
    for (int j = 2; j < M; j++) {
        // this wavefront's tiles against the new centre
        unsigned long long vm[TPL];
#pragma unroll
        for (int s = 0; s < TPL; s++) {
            const float g0 = fmaxf(fmaxf(lo0[s] - c0, c0 - hi0[s]), 0.0f);
            const float g1 = fmaxf(fmaxf(lo1[s] - c1, c1 - hi1[s]), 0.0f);
            const float g2 = fmaxf(fmaxf(lo2[s] - c2, c2 - hi2[s]), 0.0f);
            const float bound = (g0 * g0 + g1 * g1) + g2 * g2;
            vm[s] = __ballot(have[s] && bound < tmax[s]);
        }
        TRACE_FPS_TILES(vm[0], j);
        TRACE_FPS_PHASE(1);
        visit(vm, false);
        TRACE_FPS_PHASE(2);
        update_origin();
        select_next();
        // (every wavefront knows the winner; they take turns writing it: a wavefront's loads of the next iteration wait behind its stores on the one
        // vmcnt, and one wavefront carrying all of them was 1.7 us of the launch.  Collecting the centres in LDS for one write at the end: the same alone.)
        if (lane == 0 && wave == (j & (NW - 1))) { out_idx[j] = old; if (out_cen) { out_cen[3 * j] = c0; out_cen[3 * j + 1] = c1; out_cen[3 * j + 2] = c2; } }
        TRACE_FPS_PHASE(5);
    }
    TRACE_FPS_WG(1);
    if (RANGE && org_on && (flags & FPS_FLAG_FINALIZE_TEMP)) {
        __syncthreads();
        for (int p = tid; p < N; p += FPS_TT) {
            const float r = ld_f32(src, (uint32_t)p * 4u), tv = ld_f32(temp, (uint32_t)p * 4u);
            if (r == 0.0f && tv >= 0.0f && tv != t_org) st_f32(temp, (uint32_t)p * 4u, t_org);
        }
    }
}

// The fused batch's variant (planar ray table) is the one that shares the CUs with the other batches in flight: at most 96 VGPRs
// (with 98 the kernel is 2 % faster alone and the step 2 % slower: one wavefront per SIMD less beside it).  The stage entry's
// variants (rays as the [P,3] table, point lists) need a few registers more for the re-packing and would spill under that cap;
// they run alone and take the next occupancy step instead (<= 128 VGPRs, no scratch).
template <bool RANGE, bool VEC, int FPS_TT, bool EDGE = false>
__global__ __launch_bounds__(FPS_TT) __attribute__((amdgpu_waves_per_eu(4, 8))) void fps_regtab_kernel(
    const float *__restrict__ src, const float *__restrict__ rays, float *__restrict__ temp, const int32_t *__restrict__ info, FpsTiling g, int M,
    int flags, int32_t *__restrict__ out_idx, float *__restrict__ out_cen, const float *__restrict__ tiletab) {
    fps_regtab_body<RANGE, VEC, FPS_TT, false, EDGE>(src, rays, temp, info, g, M, flags, out_idx, out_cen, tiletab, nullptr, blockIdx.x);
}
template <int FPS_TT, bool EDGE = false>
__global__ __launch_bounds__(FPS_TT) __attribute__((amdgpu_waves_per_eu(5, 8))) void fps_regtab_planar_kernel(
    const float *__restrict__ src, const float *__restrict__ rays, float *__restrict__ temp, const int32_t *__restrict__ info, FpsTiling g, int M,
    int flags, int32_t *__restrict__ out_idx, float *__restrict__ out_cen, const float *__restrict__ tiletab, const float *__restrict__ rays_soa) {
    fps_regtab_body<true, true, FPS_TT, true, EDGE>(src, rays, temp, info, g, M, flags, out_idx, out_cen, tiletab, rays_soa, blockIdx.x);
}
// Two tiles per lane: images with more tiles than the workgroup has lanes (80 x 2000: 630, 128 x 2048: 1024) keep the register table -- the
// LDS-table kernel above pays three barriers per iteration.  The second table entry costs 11 registers: four wavefronts per SIMD instead of five.
template <int FPS_TT, bool EDGE = false>
__global__ __launch_bounds__(FPS_TT) __attribute__((amdgpu_waves_per_eu(4, 8))) void fps_regtab_planar2_kernel(
    const float *__restrict__ src, const float *__restrict__ rays, float *__restrict__ temp, const int32_t *__restrict__ info, FpsTiling g, int M,
    int flags, int32_t *__restrict__ out_idx, float *__restrict__ out_cen, const float *__restrict__ tiletab, const float *__restrict__ rays_soa) {
    fps_regtab_body<true, true, FPS_TT, true, EDGE, 2>(src, rays, temp, info, g, M, flags, out_idx, out_cen, tiletab, rays_soa, blockIdx.x);
}
// The same for the frames of several geometry groups in ONE launch (rpcc_compress_batch_mixed: variable H x W inside one call).  The device runs
// as many kernels side by side as the process has hardware queues -- three or four -- and this kernel keeps one CU per frame busy for 99
// dependent iterations whatever the image size, so the groups' launches side by side on streams leave most of the chip idle
// (tools_dev/launch_rate.hip, profiles/HISTORY.md).  Workgroup -> (group, frame of the group) through the table in the kernel arguments.
struct FpsGroupArgs {
    const float *src, *rays;
    float *temp;
    const int32_t *info;
    FpsTiling g;
    int32_t *out_idx;
    float *out_cen;
    const float *tiletab, *rays_soa;
};
struct FpsMulti {
    int n, first[RPCC_MAX_GROUPS + 1];   // group i owns the workgroups first[i] .. first[i + 1] - 1
    int edge[RPCC_MAX_GROUPS];           // the group's image width is no multiple of four (EDGE accesses)
    FpsGroupArgs a[RPCC_MAX_GROUPS];
};
template <int FPS_TT>
__global__ __launch_bounds__(FPS_TT) __attribute__((amdgpu_waves_per_eu(5, 8))) void fps_regtab_planar_multi_kernel(const FpsMulti m, int M, int flags) {
    const int gi = multi_group_of(m.first, m.n, blockIdx.x);
    const FpsGroupArgs &a = m.a[gi];
    const int b = (int)blockIdx.x - m.first[gi];
    // (both access variants in one kernel, chosen per workgroup: the launch lasts as long as its slowest frame, two launches as long as both)
    if (m.edge[gi]) fps_regtab_body<true, true, FPS_TT, true, true>(a.src, a.rays, a.temp, a.info, a.g, M, flags, a.out_idx, a.out_cen, a.tiletab, a.rays_soa, b);
    else            fps_regtab_body<true, true, FPS_TT, true, false>(a.src, a.rays, a.temp, a.info, a.g, M, flags, a.out_idx, a.out_cen, a.tiletab, a.rays_soa, b);
}

// ------------------------------------------------------------------------------------------------
// a3 + a5 with the FIRST pass of the farthest point sampling (at full-chip parallelism).
// FPS starts at the first candidate in row-major order.  Every wavefront re-derives it from the first
// 64 pixels of the frame; if one of them is a candidate (the usual case: the first pixels are empty and
// empty pixels are candidates) the kernel writes temp = min(1e10, d(pixel, first centre)) instead of
// 1e10 and fills the FPS tile table (FpsLds layout) so the FPS kernel starts at the second centre.
// Otherwise info[b][3] stays 0, the classic temp = 1e10 / -1 is written and the FPS kernel does its own
// first pass.  Same arithmetic either way.  One wavefront per tile, TAB_TPW tiles per wavefront.
// info[b][4] = first empty pixel that is a candidate (the representative of the FPS kernel's origin class).
// ------------------------------------------------------------------------------------------------
#define TAB_TPW 2
#define MASK_WAVES 4   // wavefronts per workgroup (independent of each other but for the frame counters at the end)
#define MASK_VGPR_ATTR
template <bool RAW, bool VEC, bool EDGE = false>   // EDGE (with VEC): the image width is no multiple of four -- 16-byte accesses at 4-byte alignment, row-end quads by element
__global__ __launch_bounds__(64 * MASK_WAVES) MASK_VGPR_ATTR void ground_mask_tab_kernel(float *__restrict__ ri, const float *__restrict__ tm,
                                                              const double *__restrict__ ground, double thr, FpsTiling g,
                                                              float *__restrict__ temp, int32_t *__restrict__ info,
                                                              float *__restrict__ tiletab) {
    __shared__ int s_cnt[MASK_WAVES], s_nz[MASK_WAVES], s_first[MASK_WAVES], s_forg[MASK_WAVES];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = g.N, T = g.T;
    // (all loads of the wavefront go out before the plane's fp64 preamble -- square root, thresholds -- is computed)
    float *ri_b = ri + (int64_t)b * P, *temp_b = temp + (int64_t)b * P;
    float4 *tab4 = reinterpret_cast<float4 *>(tiletab + (int64_t)b * FPS_TAB_ROWS * T);
    int cnt = 0, nzc = 0, first = P, forg = P;
    const int t0 = (blockIdx.x * MASK_WAVES + wave) * TAB_TPW;
    const int lrow = lane >> 3, lcol = 4 * (lane & 7);
    const float pr = ri_b[min(lane, P - 1)];                                                   // the frame's first 64 pixels (first centre)
    const f32x3 pray = ld_at(reinterpret_cast<const f32x3 *>(tm), (uint32_t)min(lane, P - 1) * 12u);
    // the loads of all TAB_TPW tiles of this wavefront first (unconditional, clamped): one memory latency, not TAB_TPW
    FpsQuad q[TAB_TPW];
#pragma unroll
    for (int k = 0; k < TAB_TPW; k++) {
        const int t = min(t0 + k, T - 1);
        const int tr = t / g.tcols, tc = t - tr * g.tcols;
        const int row = FPS_TROWS * tr + lrow, col = 32 * tc + lcol;
        q[k].nval = row < g.H ? min(max(g.W - col, 0), 4) : 0;
        q[k].p0 = q[k].nval > 0 ? row * g.W + col : 0;
        const uint32_t p0 = (uint32_t)q[k].p0;
        if (VEC && (!EDGE || q[k].nval == 4)) {
            const float4 r = ld_quad<EDGE>(ri_b, p0 * 4u);
            q[k].r[0] = r.x; q[k].r[1] = r.y; q[k].r[2] = r.z; q[k].r[3] = r.w;
            const float4 ra = ld_quad<EDGE>(tm, p0 * 12u);
            const float4 rb = ld_quad<EDGE>(tm, p0 * 12u + 16u);
            const float4 rc = ld_quad<EDGE>(tm, p0 * 12u + 32u);
            q[k].t[0] = ra.x; q[k].t[1] = ra.y; q[k].t[2] = ra.z; q[k].t[3] = ra.w; q[k].t[4] = rb.x; q[k].t[5] = rb.y;
            q[k].t[6] = rb.z; q[k].t[7] = rb.w; q[k].t[8] = rc.x; q[k].t[9] = rc.y; q[k].t[10] = rc.z; q[k].t[11] = rc.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t p = p0 + (uint32_t)(e < q[k].nval ? e : 0);
                q[k].r[e] = ld_f32(ri_b, p * 4u);
                q[k].t[3 * e] = ld_f32(tm, p * 12u); q[k].t[3 * e + 1] = ld_f32(tm, p * 12u + 4u); q[k].t[3 * e + 2] = ld_f32(tm, p * 12u + 8u);
            }
        }
    }
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
    // (The fp32 estimate in front of the fp64 dot product -- the pattern of assign_kernel's ground term -- does not lower the
    // instruction COUNT (52.4 M against 50.5 M wave instructions per batch: 14 fp32 instructions per pixel against 11 fp64 ones),
    // but fp64 instructions take two issue passes: 95.6 -> 93.8 us alone, 0.765 -> 0.762 ms per step.)
    // fp32 estimate of the numerator with its error bound in front of the fp64 sequence (fp64 instructions take two issue passes):
    // n32 = fl(fl(fl(x af + y bf) + z cf) + df) differs from the exact x a + y b + z c + d by at most 5 * 2^-24 * (|x a| + |y b| +
    // |z c| + |d|) (narrowing of the plane, three products, three sums); twice that is used.  Pixels the estimate cannot decide
    // (and NaN / infinite values, which fail both compares) take the fp64 sequence.
    const float af = (float)a, bf = (float)bb, cf = (float)c, df = (float)d;
    const float thi32 = (float)t_hi * 1.0000002f, tlo32 = (float)t_lo * 0.9999998f;
    auto classify = [&](float &r, float tx, float ty, float tz, float &x, float &y, float &z) -> bool {
        if (RAW && f2u(r) == RI_EMPTY) r = 0.0f;
        x = r * tx; y = r * ty; z = r * tz;
        if (screen) {
            const float n32 = fabsf(((x * af + y * bf) + z * cf) + df);
            const float e32 = 6.0e-7f * (((fabsf(x) * fabsf(af) + fabsf(y) * fabsf(bf)) + fabsf(z) * fabsf(cf)) + fabsf(df)) + 1.0e-30f;
            if (n32 - e32 > thi32) return true;
            if (n32 + e32 < tlo32) return false;
        }
        const double s = ((double)x * a + (double)y * bb) + (double)z * c;
        const double num = fabs(s + d);
        if (screen && num > t_hi) return true;
        if (screen && num < t_lo) return false;
        return num / div > thr;
    };
    // (the first centre -- the frame's first candidate among its first 64 pixels -- is classified AFTER the tiles' loads are issued: its own
    // load went out before them, so the wavefront pays one trip to memory at its start, not two)
    bool fast;
    float c0, c1, c2;
    int forg0;   // first EMPTY candidate among the frame's first 64 pixels (-1: none there)
    {
        float r = pr, x, y, z;
        const bool cd = classify(r, pray.x, pray.y, pray.z, x, y, z) && lane < P;
        const unsigned long long m = __ballot(cd), mo = __ballot(cd && r == 0.0f);
        fast = m != 0ull;
        const int f0 = fast ? (int)__ffsll((long long)m) - 1 : 0;
        forg0 = mo ? (int)__ffsll((long long)mo) - 1 : -1;
        c0 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(x), f0));
        c1 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(y), f0));
        c2 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(z), f0));
        // Every wavefront of the frame knows these from its own copy of the first 64 pixels: when the frame's first candidate / first
        // empty candidate lies among them (the usual case) it IS the minimum over the frame, so workgroup 0 stores it and nobody
        // runs an atomicMin on the frame's words (two of the four atomics of every workgroup).
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            info[RPCC_INFO * b + 3] = fast ? 1 : 0;
            if (fast) info[RPCC_INFO * b + 1] = f0;
            if (forg0 >= 0) info[RPCC_INFO * b + 4] = forg0;
        }
    }
#pragma unroll
    for (int k = 0; k < TAB_TPW; k++) {
        const int t = t0 + k;
        if (t >= T) break;
        float x[4], y[4], z[4], nt[4], rr[4];
        uint32_t key[4];
        bool boxc[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const bool valid = e < q[k].nval;
            float r = q[k].r[e];
            const bool cand = classify(r, q[k].t[3 * e], q[k].t[3 * e + 1], q[k].t[3 * e + 2], x[e], y[e], z[e]) && valid;
            rr[e] = r;
            const bool nz = valid && r != 0.0f;
            nt[e] = cand ? 1e10f : -1.0f;
            if (fast) {
                const float dx = x[e] - c0, dy = y[e] - c1, dz = z[e] - c2;
                const float dist = (dx * dx + dy * dy) + dz * dz;  // sampling_gpu.cu:64, un-fused
                nt[e] = cand ? fminf(dist, 1e10f) : -1.0f;
            }
            boxc[e] = cand && nz;                 // empty pixels belong to the origin class, not to the tile
            key[e] = boxc[e] ? fps_val_key(nt[e]) : 0u;
            if ((!VEC || (EDGE && q[k].nval != 4)) && valid) {
                if (RAW) st_f32(ri_b, (uint32_t)(q[k].p0 + e) * 4u, r);
                st_f32(temp_b, (uint32_t)(q[k].p0 + e) * 4u, nt[e]);
            }
            const unsigned long long mc = __ballot(cand), mz = __ballot(nz), mo = __ballot(cand && !nz);
            cnt += __popcll(mc);
            nzc += __popcll(mz);
            // inside a tile the lanes are in pixel order (lane = 8 * row + quad of the row), so the first set lane of a slot's
            // mask holds the slot's lowest pixel: wave-uniform minima in scalar registers, no reduction at the end
            if (mc) first = min(first, __builtin_amdgcn_readlane(q[k].p0, (int)__ffsll((long long)mc) - 1) + e);
            if (mo) forg = min(forg, __builtin_amdgcn_readlane(q[k].p0, (int)__ffsll((long long)mo) - 1) + e);
        }
        if (VEC && !EDGE && q[k].nval > 0) {
            if (RAW) st_at(reinterpret_cast<float4 *>(ri_b), (uint32_t)q[k].p0 * 4u, make_float4(rr[0], rr[1], rr[2], rr[3]));
            st_at(reinterpret_cast<float4 *>(temp_b), (uint32_t)q[k].p0 * 4u, make_float4(nt[0], nt[1], nt[2], nt[3]));
        }
        if (VEC && EDGE && q[k].nval == 4) {
            if (RAW) st_quad<true>(ri_b, (uint32_t)q[k].p0 * 4u, rr[0], rr[1], rr[2], rr[3]);
            st_quad<true>(temp_b, (uint32_t)q[k].p0 * 4u, nt[0], nt[1], nt[2], nt[3]);
        }
        if (fast) {
            float lo[3], hi[3], wt, wx, wy, wz;
            uint32_t widx;
            if ((__ballot(boxc[0]) | __ballot(boxc[1]) | __ballot(boxc[2]) | __ballot(boxc[3])) == 0ull) {
                // (wave-uniform) a tile without a candidate -- the ground, two tiles in five: the entry fps_tile_box / fps_tile_argmax give for it, without them
                lo[0] = lo[1] = lo[2] = __builtin_inff(); hi[0] = hi[1] = hi[2] = -__builtin_inff();
                wt = -1.0f; wx = wy = wz = 0.0f; widx = 0u;
            } else {
                fps_tile_box(x, y, z, boxc, lo, hi);
                fps_tile_argmax(x, y, z, key, q[k].p0, wt, wx, wy, wz, widx);
            }
            if (lane < 3) {
                float4 v = make_float4(lo[0], lo[1], lo[2], wt);
                if (lane == 1) v = make_float4(hi[0], hi[1], hi[2], u2f(widx));
                if (lane == 2) v = make_float4(wx, wy, wz, 0.0f);
                tab4[(int64_t)lane * T + t] = v;
            }
        }
    }
    if (lane == 0) { s_cnt[wave] = cnt; s_nz[wave] = nzc; s_first[wave] = first; s_forg[wave] = forg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tc = 0, tz = 0, tf = P, to = P;
#pragma unroll
        for (int w = 0; w < MASK_WAVES; w++) { tc += s_cnt[w]; tz += s_nz[w]; tf = min(tf, s_first[w]); to = min(to, s_forg[w]); }
        if (tc) { atomicAdd(&info[RPCC_INFO * b + 0], tc); if (!fast) atomicMin(&info[RPCC_INFO * b + 1], tf); }
        if (tz) atomicAdd(&info[RPCC_INFO * b + 2], tz);
        if (forg0 < 0 && to < P) atomicMin(&info[RPCC_INFO * b + 4], to);
    }
}
