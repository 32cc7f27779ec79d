/*
 * rpcc_oracle.c -- CPU ORACLE for the R-PCC per-frame compression hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is a plain-C restatement of the reference algorithm,
 * written from the semantics of the reference (file:line cited per function, paths relative to
 * /root/reference) and pinned bit-for-bit against
 *   (1) the reference's own C++ built from its sources (oracle/_ref, `make -C oracle ref`), and
 *   (2) the reference's Python imported in the build container (tests/golden/gen_golden.py),
 * see tests/test_oracle_vs_ref.py and tests/test_oracle_golden.py.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The product (r-pcc_amd/) never does.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  Every reference fp32
 * expression is un-fused x86 SSE arithmetic, so contraction must stay off.
 *
 * Parity status: a2,a8,a10,a11,a12,a13,f1,f3 pinned against the compiled reference; a1,a3,a5,a7
 * pinned against the imported reference Python; a6 (FPS) has no runnable reference here (CUDA only)
 * -> "parity unpinned" for exact-tie order, restated from ops/fps/src/sampling_gpu.cu:44-69,136-138.
 * a4/a9 RANSAC is third-party Open3D in the reference (random, unpinned); orc_ransac_plane is the
 * build's own seeded definition, not a restatement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* glibc-2.35 atan2f/atanf (Sun fdlibm float algorithm), restated.  The reference projection    */
/* calls libm atan2f (cpp_modules.cpp:447,450); glibc's result is not correctly rounded, so the */
/* operation sequence itself is part of the contract.  KAT: tests/test_oracle_vs_ref.py compares */
/* this against the container's libm on >=10^7 inputs.                                          */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static const float ATANHI[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
static const float ATANLO[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
static const float AT[11] = {3.3333334327e-01f,  -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                             9.0908870101e-02f,  -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                             4.9768779427e-02f,  -3.6531571299e-02f, 1.6285819933e-02f};

float orc_atanf(float x) {
    float w, s1, s2, z;
    int32_t hx = (int32_t)f2u(x);
    int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) { /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;
        if (hx > 0) return ATANHI[3] + ATANLO[3];
        return -ATANHI[3] - ATANLO[3];
    }
    if (ix < 0x3ee00000) { /* |x| < 0.4375 */
        if (ix < 0x31000000) return x; /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (AT[0] + w * (AT[2] + w * (AT[4] + w * (AT[6] + w * (AT[8] + w * AT[10])))));
    s2 = w * (AT[1] + w * (AT[3] + w * (AT[5] + w * (AT[7] + w * AT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = ATANHI[id] - ((x * (s1 + s2) - ATANLO[id]) - x);
    return (hx < 0) ? -z : z;
}

float orc_atan2f(float y, float x) {
    static const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                       pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    int32_t k, m;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return orc_atanf(y);
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = orc_atanf(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return u2f(f2u(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

void orc_atan2f_array(const float *y, const float *x, float *out, long n) {
    for (long i = 0; i < n; i++) out[i] = orc_atan2f(y[i], x[i]);
}
/* libm's own atan2f over an array: used by the KAT to compare the restatement with glibc. */
void orc_libm_atan2f_array(const float *y, const float *x, float *out, long n) {
    for (long i = 0; i < n; i++) out[i] = atan2f(y[i], x[i]);
}

/* ------------------------------------------------------------------------------------------ */
/* a2: point_cloud_to_range_image_even, cpp_modules.cpp:427-467.                               */
/* Sequential, in input order, exactly as the reference (so a depth-0 point resets its pixel).  */
/* Row/column of one point are exported separately for kernel unit tests.                       */
/* ------------------------------------------------------------------------------------------ */
void orc_project_rowcol(float x, float y, float z, int H, int W, float hfov, float vmax, float vmin,
                        float *depth, int *row, int *col) {
    float d = sqrtf(x * x + y * y + z * z);
    float az = orc_atan2f(y, x);
    if (az < 0) az = (float)((double)az + 2 * 3.14159265);
    float el = orc_atan2f(z, sqrtf(x * x + y * y));
    int c = (int)roundf(az / hfov * W);
    c = c % W;
    float vres = (vmax - vmin) / (H - 1);
    int r = (int)roundf((el - vmin) / vres);
    if (r >= H) r = H - 1;
    if (r < 0) r = 0;
    *depth = d; *row = r; *col = c;
}

void orc_project(const float *xyz, long n, int H, int W, float hfov, float vmax, float vmin, float *ri) {
    for (long i = 0; i < (long)H * W; i++) ri[i] = 0;
    for (long i = 0; i < n; i++) {
        float d; int r, c;
        orc_project_rowcol(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], H, W, hfov, vmax, vmin, &d, &r, &c);
        float *p = &ri[(long)r * W + c];
        if (*p == 0 || d < *p) *p = d;
    }
}

/* a3: range_image_to_point_cloud, dataset/transformer.py:94-101 (one fp32 multiply per component). */
void orc_backproject(const float *ri, const float *tm, long P, float *pc) {
    for (long i = 0; i < P; i++) {
        pc[3 * i] = ri[i] * tm[3 * i];
        pc[3 * i + 1] = ri[i] * tm[3 * i + 1];
        pc[3 * i + 2] = ri[i] * tm[3 * i + 2];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a5: calc_plane_residual_vertical (cpu branch), utils/segment_utils.py:44-47.                 */
/* fp64: abs(((x*a + y*b) + z*c) + d) / sqrt(((a^2+b^2)+c^2)+d^2)  -- the reference slices       */
/* plane_param[:, :3] on a (1,1,4) array, which keeps all four components in the divisor.       */
/* ------------------------------------------------------------------------------------------ */
double orc_plane_divisor4(const double *pl) {
    return sqrt(((pl[0] * pl[0] + pl[1] * pl[1]) + pl[2] * pl[2]) + pl[3] * pl[3]);
}
void orc_vertical_residual(const float *pc, const double *pl, long P, double *out) {
    double div = orc_plane_divisor4(pl);
    for (long i = 0; i < P; i++) {
        double s = ((double)pc[3 * i] * pl[0] + (double)pc[3 * i + 1] * pl[1]) + (double)pc[3 * i + 2] * pl[2];
        out[i] = fabs(s + pl[3]) / div;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a6: furthest point sampling, ops/fps/src/sampling_gpu.cu:44-69,136-138 restated sequentially: */
/* idx[0]=0, temp=1e10 (fps_utils.py:26), per iteration temp=min(temp,d), strict '>' arg-max     */
/* (lowest index wins ties), squared distance un-fused ((dx*dx)+(dy*dy))+(dz*dz).               */
/* ------------------------------------------------------------------------------------------ */
void orc_fps(const float *xyz, int n, int m, int *idx) {
    if (m <= 0 || n <= 0) return;
    float *temp = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < n; k++) temp[k] = 1e10f;
    int old = 0;
    idx[0] = 0;
    for (int j = 1; j < m; j++) {
        int besti = 0;
        float best = -1;
        float x1 = xyz[3 * old], y1 = xyz[3 * old + 1], z1 = xyz[3 * old + 2];
        for (int k = 0; k < n; k++) {
            float dx = xyz[3 * k] - x1, dy = xyz[3 * k + 1] - y1, dz = xyz[3 * k + 2] - z1;
            float d = (dx * dx + dy * dy) + dz * dz;
            float d2 = d < temp[k] ? d : temp[k];
            temp[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
        old = besti;
        idx[j] = old;
    }
    free(temp);
}

/* ------------------------------------------------------------------------------------------ */
/* a6 with the two things of the CUDA binary that the specification above leaves out, as switches (the  */
/* reference has no golden vector for either; this restates what the SOURCE defines):                    */
/*   fma  0  un-fused ((dx*dx)+(dy*dy))+(dz*dz)         (orc_fps; what -fmad=false would give)            */
/*        1  fma(dz,dz, fma(dx,dx, dy*dy))              nvcc's default --fmad=true contracts :64; the     */
/*        2  fma(dz,dz, fma(dy,dy, dx*dx))              two plausible contractions of (A + B) + C         */
/*   cuda_tie 0  lowest index among equal values (the sequential strict-'>' scan)                          */
/*            1  the winner the kernel's own reduction picks among EXACTLY equal values                    */
/*               (sampling_gpu.cu:9-13,16-21,55-69,74-134): thread tid scans k = tid, tid+bs, ... with a     */
/*               strict '>' (lowest k of its own), the shared-memory tree folds slot t+s into slot t with    */
/*               `v2 > v1 ? i2 : i1` for s = bs/2 .. 1 (ties keep the lower slot), so the survivor is the    */
/*               candidate with the smallest bit-reversed (k mod bs), then the smallest k;                   */
/*               bs = max(min(1 << (int)(log(n)/log(2)), 1024), 1) as opt_n_threads computes it in double.   */
/* ------------------------------------------------------------------------------------------ */
static inline float fps_dist(float dx, float dy, float dz, int fma_mode) {
    if (fma_mode == 1) return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
    if (fma_mode == 2) return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    return (dx * dx + dy * dy) + dz * dz;
}
int orc_fps_block_size(int n) {
    const int pow_2 = (int)(log((double)n) / log(2.0));   /* sampling_gpu.cu:9-13 */
    int b = 1 << pow_2;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return b;
}
static inline uint32_t bitrev_bits(uint32_t v, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}
void orc_fps_modes(const float *xyz, int n, int m, int fma_mode, int cuda_tie, int *idx) {
    if (m <= 0 || n <= 0) return;
    float *temp = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < n; k++) temp[k] = 1e10f;
    const int bs = orc_fps_block_size(n);
    int bits = 0;
    while ((1 << bits) < bs) bits++;
    int old = 0;
    idx[0] = 0;
    for (int j = 1; j < m; j++) {
        int besti = 0;
        float best = -1;
        uint64_t bestkey = 0;
        const float x1 = xyz[3 * old], y1 = xyz[3 * old + 1], z1 = xyz[3 * old + 2];
        for (int k = 0; k < n; k++) {
            const float dx = xyz[3 * k] - x1, dy = xyz[3 * k + 1] - y1, dz = xyz[3 * k + 2] - z1;
            const float d = fps_dist(dx, dy, dz, fma_mode);
            const float d2 = d < temp[k] ? d : temp[k];   /* min(d, temp[k]) */
            temp[k] = d2;
            if (!cuda_tie) {
                if (d2 > best) { best = d2; besti = k; }
            } else {
                /* the CUDA kernel's survivor among equal values: smallest (bitrev(k mod bs), k) */
                const uint64_t key = ((uint64_t)bitrev_bits((uint32_t)(k % bs), bits) << 32) | (uint32_t)k;
                if (d2 > best || (d2 == best && key < bestkey)) { best = d2; besti = k; bestkey = key; }
            }
        }
        old = besti;
        idx[j] = old;
    }
    free(temp);
}
/* The kernel itself, thread by thread (per-thread strided scan + the shared-memory tree), for small n: the check that  */
/* the closed form of cuda_tie = 1 above is what the source's reduction computes.                                        */
void orc_fps_cuda_emulated(const float *xyz, int n, int m, int fma_mode, int *idx) {
    if (m <= 0 || n <= 0) return;
    float *temp = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < n; k++) temp[k] = 1e10f;
    const int bs = orc_fps_block_size(n);
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
    int old = 0;
    idx[0] = 0;
    for (int j = 1; j < m; j++) {
        const float x1 = xyz[3 * old], y1 = xyz[3 * old + 1], z1 = xyz[3 * old + 2];
        for (int tid = 0; tid < bs; tid++) {
            int besti = 0;
            float best = -1;
            for (int k = tid; k < n; k += bs) {
                const float dx = xyz[3 * k] - x1, dy = xyz[3 * k + 1] - y1, dz = xyz[3 * k + 2] - z1;
                const float d = fps_dist(dx, dy, dz, fma_mode);
                const float d2 = d < temp[k] ? d : temp[k];
                temp[k] = d2;
                besti = d2 > best ? k : besti;
                best = d2 > best ? d2 : best;
            }
            dists[tid] = best;
            dists_i[tid] = besti;
        }
        for (int s = bs / 2; s >= 1; s >>= 1)
            for (int tid = 0; tid < s; tid++) {   /* __update(dists, dists_i, tid, tid + s) */
                const float v1 = dists[tid], v2 = dists[tid + s];
                const int i1 = dists_i[tid], i2 = dists_i[tid + s];
                dists[tid] = v1 > v2 ? v1 : v2;
                dists_i[tid] = v2 > v1 ? i2 : i1;
            }
        old = dists_i[0];
        idx[j] = old;
    }
    free(temp); free(dists); free(dists_i);
}

/* ------------------------------------------------------------------------------------------ */
/* a7: assignment, utils/segment_utils.py:127-131 + :168-169.                                   */
/*   ground term (calc_plane_residual_depth :64-67): g = ri - (-d / ((a*tx + b*ty) + c*tz)), fp64 */
/*   cluster term (calc_cluster_residual_radius :21-23): sqrtf(((dx*dx)+(dy*dy))+(dz*dz)), fp32   */
/*   distance = concat(g, radius) as fp64; seg = argmax(-abs(distance)) (first maximum; a NaN     */
/*   anywhere makes numpy return the first NaN position); then seg[seg>0]+=1; seg[ri==0]=1.       */
/* ------------------------------------------------------------------------------------------ */
void orc_assign(const float *ri, const float *pc, const float *tm, const double *pl, const float *cen, int M,
                long P, int *seg) {
    for (long i = 0; i < P; i++) {
        double den = ((double)tm[3 * i] * pl[0] + (double)tm[3 * i + 1] * pl[1]) + (double)tm[3 * i + 2] * pl[2];
        /* numpy evaluates plane[..., :3] * transform_map: fp64 * fp32 -> same products, index order */
        double rp = -pl[3] / den;
        double g = (double)ri[i] - rp;
        double best = -fabs(g);
        int bi = 0;
        int nan_seen = isnan(best);
        float x = pc[3 * i], y = pc[3 * i + 1], z = pc[3 * i + 2];
        for (int k = 0; k < M && !nan_seen; k++) {
            float dx = x - cen[3 * k], dy = y - cen[3 * k + 1], dz = z - cen[3 * k + 2];
            float r = sqrtf((dx * dx + dy * dy) + dz * dz);
            double v = -fabs((double)r);
            if (isnan(v)) { bi = k + 1; nan_seen = 1; break; }
            if (v > best) { best = v; bi = k + 1; }
        }
        if (bi > 0) bi += 1;
        if (ri[i] == 0) bi = 1;
        seg[i] = bi;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a8: point_modeling, cpp_modules.cpp:471-518.  double sequential accumulate, ids 0/1 -> 0,     */
/* empty id -> 0.0/0 = NaN.  Returns cluster_num = max(seg)+1.                                  */
/* ------------------------------------------------------------------------------------------ */
int orc_point_modeling(const float *ri, const int *seg, long P, float *out /* >= max+1 */) {
    int cn = 0;
    for (long i = 0; i < P; i++) if (seg[i] > cn) cn = seg[i];
    cn += 1;
    double *sum = (double *)calloc((size_t)cn, sizeof(double));
    long *cnt = (long *)calloc((size_t)cn, sizeof(long));
    for (long i = 0; i < P; i++) {
        int s = seg[i];
        if (s != 0 && s != 1) { sum[s] += (double)ri[i]; cnt[s]++; }
    }
    for (int k = 0; k < cn; k++) {
        if (k == 0 || k == 1) out[k] = 0.0f;
        else out[k] = (float)(sum[k] / (double)(size_t)cnt[k]);
    }
    free(sum); free(cnt);
    return cn;
}

/* a10: intra_predict, cpp_modules.cpp:248-285 (fp32, IEEE divide, no FMA). */
void orc_intra_predict(const int *seg, const float *mp, const float *tm, long P, float *pred) {
    for (long i = 0; i < P; i++) {
        const float *p = &mp[4 * (long)seg[i]];
        if (p[0] + p[1] + p[2] == 0) pred[i] = p[3];
        else pred[i] = -p[3] / (p[0] * tm[3 * i] + p[1] * tm[3 * i + 1] + p[2] * tm[3 * i + 2]);
    }
}

/* a11: uniform_quantize, cpp_modules.cpp:288-334.  roundf(res/acc), skip label 1, output grouped */
/* by label ascending, row-major inside a label.  Returns nnz.                                   */
long orc_uniform_quantize(const int *seg, const float *res, long P, float acc, int *out) {
    int cn = 0;
    for (long i = 0; i < P; i++) if (seg[i] > cn) cn = seg[i];
    cn += 1;
    long *cnt = (long *)calloc((size_t)cn + 1, sizeof(long));
    for (long i = 0; i < P; i++) if (seg[i] != 1) cnt[seg[i] + 1]++;
    for (int k = 0; k < cn; k++) cnt[k + 1] += cnt[k];
    long total = cnt[cn];
    for (long i = 0; i < P; i++) {
        int s = seg[i];
        if (s != 1) out[cnt[s]++] = (int)roundf(res[i] / acc);
    }
    free(cnt);
    return total;
}

/* a13: nonuniform_quantize, cpp_modules.cpp:337-424. */
long orc_nonuniform_quantize(const int *seg, const float *res, const int *kp, const int *lkpn, const float *lacc,
                             int level_num, int ground_level, long P, int *out, int *salience /* >= max+1 */) {
    int cn = 0;
    for (long i = 0; i < P; i++) if (seg[i] > cn) cn = seg[i];
    cn += 1;
    long *cnt = (long *)calloc((size_t)cn + 1, sizeof(long));
    int *kpn = (int *)calloc((size_t)cn, sizeof(int));
    int *pn = (int *)calloc((size_t)cn, sizeof(int));
    for (long i = 0; i < P; i++) {
        int s = seg[i];
        if (s == 1) continue;
        if (kp[i] > 0) kpn[s]++;
        pn[s]++;
        cnt[s + 1]++;
    }
    for (int k = 0; k < cn; k++) {
        int lv = 0;
        if (k == 0) lv = ground_level;
        else if (k == 1) lv = level_num - 1;
        else if (pn[k] < 30) lv = level_num - 1;
        else for (int l = 0; l < level_num; l++) if (kpn[k] >= lkpn[l]) { lv = l; break; }
        salience[k] = lv;
    }
    for (int k = 0; k < cn; k++) cnt[k + 1] += cnt[k];
    long total = cnt[cn];
    for (long i = 0; i < P; i++) {
        int s = seg[i];
        if (s != 1) out[cnt[s]++] = (int)roundf(res[i] / lacc[salience[s]]);
    }
    free(cnt); free(kpn); free(pn);
    return total;
}

/* ------------------------------------------------------------------------------------------ */
/* a12: extract_features_with_segment + mark_as_picked, cpp_modules.cpp:10-25,28-121.            */
/* Intended semantics: feat / key_point_map are ZERO where the reference leaves its freshly       */
/* allocated arrays uninitialised (cpp_modules.cpp:38-43, a latent reference defect).            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float c; int s; } fpair;
static int fpair_cmp(const void *a, const void *b) {
    const fpair *p = (const fpair *)a, *q = (const fpair *)b;
    if (p->c < q->c) return -1;
    if (q->c < p->c) return 1;
    return (p->s > q->s) - (p->s < q->s);
}
static int mark_picked(const float *rip, unsigned char *picked, int w, int h_i, int w_i, int fr) {
    int ret = 1;
    float r = rip[(long)h_i * w + w_i];
    for (int i = -fr; i <= fr; i++) {
        float dif = r - rip[(long)h_i * w + w_i + i];
        if (fabsf(dif) < 0.2f) picked[(long)h_i * w + w_i] = 1;
        if (dif > 0.3f) ret = 0;
    }
    return ret;
}
void orc_extract_features_with_segment(const float *ri, const int *seg, int h, int w, int fr, int segments,
                                       int sharp_num, int less_sharp_num, int flat_num, float *feat, int *kp) {
    memset(feat, 0, sizeof(float) * (size_t)h * w);
    memset(kp, 0, sizeof(int) * (size_t)h * w);
    unsigned char *picked = (unsigned char *)calloc((size_t)h * w, 1);
    float *vri = (float *)malloc(sizeof(float) * (size_t)w);
    int *vidx = (int *)malloc(sizeof(int) * (size_t)w);
    fpair *fm = (fpair *)malloc(sizeof(fpair) * (size_t)w);
    for (int h_i = 0; h_i < h; h_i++) {
        int vl = 0;
        for (int w_i = 0; w_i < w; w_i++) {
            int s = seg[(long)h_i * w + w_i];
            if (s != 0 && s != 1) { vri[vl] = ri[(long)h_i * w + w_i]; vidx[vl] = w_i; vl++; }
        }
        if (vl < segments + fr * 2 + 1) continue;
        int L = 0;
        for (int s_i = fr; s_i < vl - fr; s_i++) {
            float f = 0;
            for (int k = -fr; k <= fr; k++) f += vri[s_i + k] - vri[s_i];
            f = f * f;
            f /= (float)(2 * fr);
            f /= vri[s_i];
            feat[(long)h_i * w + vidx[s_i]] = f;
            fm[L].c = f; fm[L].s = s_i; L++;
        }
        int chunk = L / segments;
        for (int j = 0; j < segments; j++) {
            int sp = chunk * j, ep = chunk * (j + 1);
            int n = 0;
            qsort(fm + sp, (size_t)(ep - sp), sizeof(fpair), fpair_cmp);
            for (int i = ep - 1; i >= sp; i--) {
                int idx = fm[i].s;
                fm[i].c = 0;
                if (picked[(long)h_i * w + vidx[idx]] == 0)
                    if (mark_picked(ri, picked, w, h_i, vidx[idx], fr)) {
                        n += 1;
                        if (n < sharp_num) kp[(long)h_i * w + vidx[idx]] = 3;
                        else if (n < less_sharp_num) kp[(long)h_i * w + vidx[idx]] = 2;
                        else break;
                    }
            }
            n = 0;
            qsort(fm + sp, (size_t)(ep - sp), sizeof(fpair), fpair_cmp);
            for (int i = sp; i < ep; i++) {
                if (fm[i].c == 0) continue;
                int idx = fm[i].s;
                fm[i].c = 0;
                if (picked[(long)h_i * w + vidx[idx]] == 0)
                    if (mark_picked(ri, picked, w, h_i, vidx[idx], fr)) {
                        n += 1;
                        if (n < flat_num) kp[(long)h_i * w + vidx[idx]] = 1;
                        else break;
                    }
            }
        }
    }
    free(picked); free(vri); free(vidx); free(fm);
}

/* f1: extract_contour, cpp_modules.cpp:521-558.  Returns the sequence length. */
long orc_extract_contour(const int *im, int h, int w, int *cm, int *seq) {
    long n = 0;
    for (int h_i = 0; h_i < h; h_i++) {
        seq[n++] = im[(long)h_i * w];
        cm[(long)h_i * w] = 1;
        for (int w_i = 1; w_i < w; w_i++) {
            long p = (long)h_i * w + w_i;
            if (im[p] - im[p - 1] != 0) { seq[n++] = im[p]; cm[p] = 1; }
            else cm[p] = 0;
        }
    }
    return n;
}

/* f3: recover_map, cpp_modules.cpp:561-593. */
void orc_recover_map(const int *cm, const int *seq, long l, int h, int w, int *im) {
    long P = (long)h * w, ptr = 0;
    for (long i = 0; i < l; i++) {
        int index = seq[i];
        im[ptr++] = index;
        if (ptr >= P) break;
        while (cm[ptr] == 0) {
            im[ptr++] = index;
            if (ptr >= P) break;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* NumPy 2.2 fp32 pairwise mean (plane-mode fallbacks, utils/segment_utils.py:204,216):          */
/* 8192-element blocks added sequentially in fp32; each block summed pairwise (n<8 plain loop    */
/* from -0.0; n<=128 eight interleaved accumulators + tail; else split at n/2 rounded down to a  */
/* multiple of 8).  mean = sum / n in fp32.  NumPy-version dependent; fixtures record 2.2.6.     */
/* ------------------------------------------------------------------------------------------ */
static float np_pairwise_f32(const float *a, long n) {
    if (n < 8) {
        float res = -0.0f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_f32(a, n2) + np_pairwise_f32(a + n2, n - n2);
    }
}
float orc_np_mean_f32(const float *a, long n) {
    if (n == 0) { volatile float z = 0.0f; return z / z; } /* numpy: 0/0 -> x86 default NaN 0xFFC00000 */
    float total = 0.0f;
    int first = 1;
    for (long off = 0; off < n; off += 8192) {
        long m = n - off < 8192 ? n - off : 8192;
        float s = np_pairwise_f32(a + off, m);
        if (first) { total = s; first = 0; } else total += s;
    }
    return total / (float)n;
}

/* ------------------------------------------------------------------------------------------ */
/* Build-defined seeded RANSAC plane fit (a4/a9).  NOT a restatement of Open3D (absent, random,  */
/* unpinned): this is the specification the HIP kernel implements, kept here so tests can check   */
/* the kernel against a sequential version of the same definition.  See DESIGN.md "RANSAC".       */
/*   rng: counter-based 32-bit mix hash (seed, iteration, draw).                                 */
/*   fit: centroid + covariance, largest-determinant closed form (fp64).                         */
/*   score: inlier count, then smaller sum of squared distance; refit on the best inlier set.    */
/* ------------------------------------------------------------------------------------------ */
uint32_t orc_mix32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u + 0x7F4A7C15u;
    h ^= b + 0x85EBCA6Bu + (h << 6) + (h >> 2);
    h *= 0xC2B2AE35u;
    h ^= c + 0x27D4EB2Fu + (h << 6) + (h >> 2);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

/* ------------------------------------------------------------------------------------------ */
/* RANSAC plane specification (sequential form).  Points are fp32 xyz promoted to fp64.          */
/* ------------------------------------------------------------------------------------------ */
#define RS_NT 256 /* number of strided partial sums of the ordered fp64 reduction */

/* ordered fp64 sum of v[0..n): partial[t] = sum_{i = t (mod 256), ascending} v[i]; then a fixed binary
 * tree partial[t] += partial[t+stride], stride = 128..1.  The HIP kernel uses the same order. */
static double rs_treesum(const double *v, long n) {
    double part[RS_NT];
    for (int t = 0; t < RS_NT; t++) {
        double s = 0.0;
        for (long i = t; i < n; i += RS_NT) s += v[i];
        part[t] = s;
    }
    for (int stride = RS_NT / 2; stride >= 1; stride >>= 1)
        for (int t = 0; t < stride; t++) part[t] += part[t + stride];
    return part[0];
}

/* plane through a point set given its centroid and centred second moments: the closed form that
 * solves for the axis with the largest 2x2 determinant.  Returns 0 when degenerate. */
static int rs_plane_from_moments(const double c[3], double xx, double xy, double xz, double yy, double yz, double zz,
                                 double out[4]) {
    const double det_x = yy * zz - yz * yz, det_y = xx * zz - xz * xz, det_z = xx * yy - xy * xy;
    double a, b, cc;
    if (det_x >= det_y && det_x >= det_z) { a = det_x; b = xz * yz - xy * zz; cc = xy * yz - xz * yy; }
    else if (det_y >= det_z) { a = xz * yz - xy * zz; b = det_y; cc = xy * xz - yz * xx; }
    else { a = xy * yz - xz * yy; b = xy * xz - yz * xx; cc = det_z; }
    const double nrm = sqrt((a * a + b * b) + cc * cc);
    if (!(nrm > 0.0)) return 0;
    a /= nrm; b /= nrm; cc /= nrm;
    out[0] = a; out[1] = b; out[2] = cc;
    out[3] = -((a * c[0] + b * c[1]) + cc * c[2]);
    return 1;
}

/* inlier test: fp32, un-fused, plane narrowed to fp32 */
static int rs_inlier(const float pl[4], const float *q, float thr) {
    return fabsf(((pl[0] * q[0] + pl[1] * q[1]) + pl[2] * q[2]) + pl[3]) < thr;
}

/* pts: fp32 [n,3].  ransac_n sample size, iters hypotheses, thr inlier distance, seed.
 * Returns the inlier count of the winning hypothesis; plane[4] fp64 out. */
long orc_ransac_plane(const float *pts, long n, int ransac_n, int iters, double thr, uint32_t seed, double *plane) {
    plane[0] = 0; plane[1] = 0; plane[2] = 1; plane[3] = 0;
    if (n < ransac_n || ransac_n < 3 || ransac_n > 16) return 0;
    long best_cnt = -1;
    double best[4] = {0, 0, 1, 0};
    for (int h = 0; h < iters; h++) {
        long idx[16];
        for (int k = 0; k < ransac_n; k++) {
            uint32_t a = 0;
            long cand;
            int dup;
            do {
                cand = (long)(orc_mix32(seed, (uint32_t)(h * 16 + k), a++) % (uint32_t)n);
                dup = 0;
                for (int j = 0; j < k; j++) dup |= (idx[j] == cand);
            } while (dup);
            idx[k] = cand;
        }
        double c[3] = {0, 0, 0};
        for (int k = 0; k < ransac_n; k++) for (int a = 0; a < 3; a++) c[a] += (double)pts[3 * idx[k] + a];
        for (int a = 0; a < 3; a++) c[a] /= (double)ransac_n;
        double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
        for (int k = 0; k < ransac_n; k++) {
            const double rx = (double)pts[3 * idx[k]] - c[0], ry = (double)pts[3 * idx[k] + 1] - c[1],
                         rz = (double)pts[3 * idx[k] + 2] - c[2];
            xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
        }
        double pl[4];
        if (!rs_plane_from_moments(c, xx, xy, xz, yy, yz, zz, pl)) continue;
        long cnt = 0;
        const float pf[4] = {(float)pl[0], (float)pl[1], (float)pl[2], (float)pl[3]};
        for (long i = 0; i < n; i++) cnt += rs_inlier(pf, &pts[3 * i], (float)thr);
        if (cnt > best_cnt) { best_cnt = cnt; memcpy(best, pl, sizeof(best)); }
    }
    if (best_cnt < 0) return 0;
    memcpy(plane, best, sizeof(best));
    if (best_cnt < 3) return best_cnt;
    /* refit on the inliers of the winner, ordered reductions */
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    unsigned char *in = (unsigned char *)malloc((size_t)n);
    const float bf[4] = {(float)best[0], (float)best[1], (float)best[2], (float)best[3]};
    for (long i = 0; i < n; i++) in[i] = (unsigned char)rs_inlier(bf, &pts[3 * i], (float)thr);
    double c[3];
    for (int a = 0; a < 3; a++) {
        for (long i = 0; i < n; i++) v[i] = in[i] ? (double)pts[3 * i + a] : 0.0;
        c[a] = rs_treesum(v, n) / (double)best_cnt;
    }
    double m[6];
    static const int A[6] = {0, 0, 0, 1, 1, 2}, Bx[6] = {0, 1, 2, 1, 2, 2};
    for (int q = 0; q < 6; q++) {
        for (long i = 0; i < n; i++)
            v[i] = in[i] ? ((double)pts[3 * i + A[q]] - c[A[q]]) * ((double)pts[3 * i + Bx[q]] - c[Bx[q]]) : 0.0;
        m[q] = rs_treesum(v, n);
    }
    double pl[4];
    if (rs_plane_from_moments(c, m[0], m[1], m[2], m[3], m[4], m[5], pl)) memcpy(plane, pl, sizeof(pl));
    free(v); free(in);
    return best_cnt;
}

/* Ground candidate selection (utils/segment_utils.py:101-106) with the build's deterministic
 * stand-in for np.random.choice: pixels with z < -1.5 in row-major order; more than max_pts ->
 * systematic subsample of exactly max_pts (candidate i kept iff floor((i+1)*max/n) > floor(i*max/n));
 * fewer than min_pts -> every pixel (zeros included).  Writes fp32 xyz, returns the count. */
long orc_ground_candidates(const float *ri, const float *tm, long P, float zthr, long max_pts, long min_pts, float *out) {
    long nc = 0;
    for (long p = 0; p < P; p++) nc += (ri[p] * tm[3 * p + 2] < zthr);
    long m = 0;
    if (nc < min_pts) {
        for (long p = 0; p < P; p++) { out[3 * m] = ri[p] * tm[3 * p]; out[3 * m + 1] = ri[p] * tm[3 * p + 1]; out[3 * m + 2] = ri[p] * tm[3 * p + 2]; m++; }
        return m;
    }
    long i = 0;
    for (long p = 0; p < P; p++) {
        if (!(ri[p] * tm[3 * p + 2] < zthr)) continue;
        int keep = 1;
        if (nc > max_pts) keep = ((i + 1) * max_pts) / nc > (i * max_pts) / nc;
        if (keep) { out[3 * m] = ri[p] * tm[3 * p]; out[3 * m + 1] = ri[p] * tm[3 * p + 1]; out[3 * m + 2] = ri[p] * tm[3 * p + 2]; m++; }
        i++;
    }
    return m;
}
