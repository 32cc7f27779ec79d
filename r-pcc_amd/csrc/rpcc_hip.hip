// rpcc_hip.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of librpcc_hip.so.
//
// One translation unit, built by `hipcc --offload-arch=gfx950 -O3 -ffp-contract=off` (see
// __graft_entry__.build / r-pcc_amd/build.py).  Interface: include/rpcc_hip.h.  Design, data layout
// and the roofline of every kernel: DESIGN.md.  Reference citations are relative to the reference
// repository root.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>
#include "../../include/rpcc_hip.h"
#include "rpcc_device.h"

using namespace rpcc;

// ------------------------------------------------------------------------------------------------
// host-side error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int set_err(int code, const char *fmt, const char *a = "", const char *b = "") {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return set_err(RPCC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define ARG_TRY(cond)                                                         \
    do {                                                                      \
        if (!(cond)) return set_err(RPCC_ERR_ARG, "bad argument: %s%s", #cond); \
    } while (0)
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

extern "C" int rpcc_version(void) { return RPCC_ABI_VERSION; }
extern "C" const char *rpcc_last_error(void) { return g_err; }

// Developer trace: the shipped library carries none.  Only a build with -DRPCC_DEVTRACE includes rpcc_trace.h (cycle stamps at
// phase boundaries of the instrumented kernels); here its hooks are empty and rpcc_debug_stamps() refuses a buffer.
#ifdef RPCC_DEVTRACE
#include "rpcc_trace.h"
#else
extern "C" int rpcc_debug_stamps(void *dev_i64_buffer) {
    if (dev_i64_buffer == nullptr) return RPCC_OK;
    return set_err(RPCC_ERR_ARG, "rpcc_debug_stamps: this library was built without -DRPCC_DEVTRACE%s%s");
}
#define DBG_STAMP(slot_) do { } while (0)
#define TRACE_FPS_DECLS() do { } while (0)
#define TRACE_FPS_PHASE(i_) do { } while (0)
#define TRACE_FPS_VISIT(k_) do { } while (0)
#define TRACE_FPS_TILES(vm_, j_) do { } while (0)
#define TRACE_FPS_WG(end_) do { } while (0)
#define TRACE_ORD_DECLS() do { } while (0)
#define TRACE_ORD_PHASE(i_) do { } while (0)
#define TRACE_ORD_COUNT(i_) do { } while (0)
#define TRACE_ORD_END() do { } while (0)
#endif

// Kernel attributes (dynamic LDS size) are set once per (device, kernel), not per launch.
static std::mutex g_attr_mu;
struct AttrDone { int dev; const void *fn; int bytes; };
static std::vector<AttrDone> g_attr_done;
static hipError_t ensure_dyn_lds(const void *fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_attr_mu);
    AttrDone *slot = nullptr;
    for (auto &d : g_attr_done)
        if (d.dev == dev && d.fn == fn) slot = &d;
    if (slot && slot->bytes >= bytes) return hipSuccess;   // the largest size asked for so far is in force
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    if (slot) { slot->bytes = bytes; return hipSuccess; }
    // (nothing throws across the C ABI: a failed growth of the table only means the attribute is set again next time)
    try { g_attr_done.push_back({dev, fn, bytes}); } catch (...) { }
    return hipSuccess;
}

// Timer object (bench.py): hipEvents around the FPS launches of the calls it is handed to, on the stream the kernel is
// launched on.  State lives in the object (created by the caller, one per measuring thread), not in the library.
struct rpcc_timer {
    std::mutex mu;
    std::vector<hipEvent_t> ev0, ev1;
    size_t used = 0;
};
extern "C" void *rpcc_timer_create(void) { return new (std::nothrow) rpcc_timer(); }
extern "C" void rpcc_timer_destroy(void *t) {
    rpcc_timer *tm = reinterpret_cast<rpcc_timer *>(t);
    if (!tm) return;
    for (size_t i = 0; i < tm->ev0.size(); i++) { (void)hipEventDestroy(tm->ev0[i]); (void)hipEventDestroy(tm->ev1[i]); }
    delete tm;
}
// creates the events of the next `launches` timed launches now (hipEventCreate is not cheap: without this the first use of
// every slot pays for it inside the caller's timed region)
extern "C" int rpcc_timer_reserve(void *t, int launches) {
    rpcc_timer *tm = reinterpret_cast<rpcc_timer *>(t);
    ARG_TRY(tm != nullptr && launches >= 0 && launches <= (1 << 20));
    std::lock_guard<std::mutex> lk(tm->mu);
    try {   // (nothing throws across the C ABI: the vectors' growth is the only thing here that can)
        tm->ev0.reserve((size_t)launches);
        tm->ev1.reserve((size_t)launches);
    } catch (...) {
        return set_err(RPCC_ERR_HIP, "rpcc_timer_reserve: out of host memory%s%s");
    }
    while (tm->ev0.size() < (size_t)launches) {   // (capacity reserved: push_back cannot throw)
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return set_err(RPCC_ERR_HIP, "rpcc_timer_reserve: hipEventCreate failed%s%s"); }
        tm->ev0.push_back(a);
        tm->ev1.push_back(b);
    }
    return RPCC_OK;
}
extern "C" int rpcc_timer_read(void *t, double *ms, int *launches) {
    rpcc_timer *tm = reinterpret_cast<rpcc_timer *>(t);
    ARG_TRY(tm != nullptr);
    std::lock_guard<std::mutex> lk(tm->mu);
    double tot = 0;
    for (size_t i = 0; i < tm->used; i++) {
        float v = 0;
        HIP_TRY(hipEventSynchronize(tm->ev1[i]));
        HIP_TRY(hipEventElapsedTime(&v, tm->ev0[i], tm->ev1[i]));
        tot += v;
    }
    if (ms) *ms = tot;
    if (launches) *launches = (int)tm->used;
    tm->used = 0;
    return RPCC_OK;
}
struct FpsTimer {
    hipStream_t s;
    rpcc_timer *tm;
    size_t slot = 0;
    FpsTimer(hipStream_t st, void *timer) : s(st), tm(reinterpret_cast<rpcc_timer *>(timer)) {
        if (!tm) return;
        std::lock_guard<std::mutex> lk(tm->mu);
        slot = tm->used;
        // a launch that cannot get its pair of events (host memory, event creation) goes untimed: nothing throws across the C ABI
        try {
            while (tm->ev0.size() <= slot) {
                tm->ev0.reserve(tm->ev0.size() + 1);
                tm->ev1.reserve(tm->ev1.size() + 1);
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess) { tm = nullptr; return; }
                if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); tm = nullptr; return; }
                tm->ev0.push_back(a);
                tm->ev1.push_back(b);
            }
        } catch (...) { tm = nullptr; return; }
        tm->used = slot + 1;
        (void)hipEventRecord(tm->ev0[slot], s);
    }
    ~FpsTimer() {
        if (!tm) return;
        std::lock_guard<std::mutex> lk(tm->mu);
        (void)hipEventRecord(tm->ev1[slot], s);
    }
};

// a / d for 32-bit unsigned a and a divisor d >= 2 that is the same for a whole frame: multiply-high by a 33-bit reciprocal
// prepared once (the "branch-free" form of invariant division: q = mulhi(a, magic); ((a - q) >> 1) + q) >> shift; exact for
// every a < 2^32) -- five VALU instructions instead of the ~20 of a hardware-assisted 32-bit division per use.
struct UDiv32 { uint32_t magic, shift; };
__device__ __forceinline__ UDiv32 udiv32_make(uint32_t d) {
    UDiv32 u;
    const uint32_t L = 31u - (uint32_t)__builtin_clz(d);
    if ((d & (d - 1u)) == 0u) { u.magic = 0u; u.shift = L - 1u; return u; }
    const unsigned long long num = (1ull << L) << 32;
    uint32_t m = (uint32_t)(num / d);
    const uint32_t rem = (uint32_t)(num - (unsigned long long)m * d);
    m += m;
    const uint32_t tw = rem + rem;
    if (tw >= d || tw < rem) m += 1u;
    u.magic = 1u + m; u.shift = L;
    return u;
}
__device__ __forceinline__ uint32_t udiv32(uint32_t a, const UDiv32 u) {
    const uint32_t q = __umulhi(a, u.magic);
    return (((a - q) >> 1) + q) >> u.shift;
}

// ================================================================================================
// a2  spherical projection  (cpp_modules.cpp:427-467)
// ================================================================================================
#define RI_EMPTY 0xFFFFFFFFu
#define RPCC_INFO 8      // int32 per frame in `info`: n_left, first candidate, nnz, table valid, first empty candidate, 3 spare

struct RowCol {
    float depth;
    int pix;
    float colf, rowf;  // the values handed to roundf (test hook of the fast path)
};

__device__ __forceinline__ RowCol project_point(float x, float y, float z, const rpcc_geom g) {
    RowCol o;
    o.depth = sqrtf(x * x + y * y + z * z);                       // :446
    float az = atan2f_fdlibm(y, x);                               // :447
    if (az < 0) az = (float)((double)az + 2 * 3.14159265);        // :448-449 (double literal)
    const float el = atan2f_fdlibm(z, sqrtf(x * x + y * y));      // :450
    o.colf = az / g.horizontal_fov * (float)g.W;
    int col = (int)roundf(o.colf);                                // :451
    if (col < 0 || col >= g.W) col = col % g.W;                   // :452 (identity inside [0, W): skips the integer division)
    const float vres = (g.vertical_max - g.vertical_min) / (float)(g.H - 1);  // :453
    o.rowf = (el - g.vertical_min) / vres;
    int row = (int)roundf(o.rowf);                                // :454
    row = row >= g.H ? g.H - 1 : row;                             // :455-458
    row = row < 0 ? 0 : row;
    o.pix = row * g.W + col;
    return o;
}

__device__ __forceinline__ int find_frame(const int64_t *__restrict__ offs, int B, int64_t i) {
    int lo = 0, hi = B;  // offs[lo] <= i < offs[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// MODE 0: min-depth pass over every frame, records frames that hold a depth-0 point.
// MODE 3: MODE 0 restricted to flagged frames.
// MODE 1: for flagged frames, last input position of a depth-0 point per pixel.
// MODE 2: for flagged frames, min-depth pass restricted to points after that position.
// Grid-stride; the flagged-only modes leave at once when no frame of the batch is flagged (flags[B]).
template <int MODE>
__global__ __launch_bounds__(256) void project_kernel(const float *__restrict__ xyz, const int64_t *__restrict__ offs,
                                                      int64_t total, int64_t base, int B, rpcc_geom g,
                                                      uint32_t *__restrict__ ri, int32_t *__restrict__ lastz,
                                                      int32_t *__restrict__ flags, int ps) {
    // ps: floats per point (3: packed xyz; 4: the .bin rows x, y, z, intensity as stored -- dataset/dataset.py:48-50,62)
    // offs[] holds absolute point indices; this launch covers points base .. base+total (frames offs[0..B])
    if (MODE != 0 && flags[B] == 0) return;  // no frame of this batch holds a depth-0 point
    const int64_t P = (int64_t)g.H * g.W;
    for (int64_t i = base + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < base + total;
         i += (int64_t)gridDim.x * blockDim.x) {
        // one binary search per wavefront (first active lane); lanes beyond that frame's end search again
        int b = __builtin_amdgcn_readfirstlane(find_frame(offs, B, __shfl(i, __ffsll((long long)__ballot(1)) - 1, 64)));
        if (i >= offs[b + 1]) b = find_frame(offs, B, i);
        if (MODE != 0 && flags[b] == 0) continue;
        const float x = xyz[ps * i], y = xyz[ps * i + 1], z = xyz[ps * i + 2];
        const RowCol rc = project_point(x, y, z, g);
        if (!(fabsf(rc.depth) <= 3.402823466e+38f)) continue;  // NaN / inf depth: skipped (reference: UB)
        const int32_t pos = (int32_t)(i - offs[b]) + 1;
        if (MODE == 0) {
            if (rc.depth == 0.0f) { flags[b] = 1; flags[B] = 1; continue; }
            atomicMin(&ri[b * P + rc.pix], f2u(rc.depth));
        } else if (MODE == 3) {
            if (rc.depth != 0.0f) atomicMin(&ri[b * P + rc.pix], f2u(rc.depth));
        } else if (MODE == 1) {
            if (rc.depth == 0.0f) atomicMax(&lastz[b * P + rc.pix], pos);
        } else {
            if (rc.depth != 0.0f && pos > lastz[b * P + rc.pix]) atomicMin(&ri[b * P + rc.pix], f2u(rc.depth));
        }
    }
}

// fills ri with the "untouched" pattern; with FIXUP only for flagged frames (and zeroes lastz there)
template <bool FIXUP>
__global__ __launch_bounds__(256) void project_fill_kernel(uint32_t *__restrict__ ri, int32_t *__restrict__ lastz,
                                                           int32_t *__restrict__ flags, int P) {
    const int b = blockIdx.y;
    const int B = gridDim.y;
    if (FIXUP && flags[b] == 0) return;
    const int64_t base = (int64_t)b * P;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        ri[base + p] = RI_EMPTY;
        if (FIXUP) lastz[base + p] = 0;
    }
    if (!FIXUP && blockIdx.x == 0 && threadIdx.x == 0) {
        flags[b] = 0;
        if (b == 0) flags[B] = 0;
    }
}

// RI_EMPTY -> 0.  FLAGGED: only frames with flags[b] set (the others are already final).
template <bool FLAGGED>
__global__ __launch_bounds__(256) void project_finalize_kernel(uint32_t *__restrict__ ri, int P,
                                                               const int32_t *__restrict__ flags) {
    const int b = blockIdx.y;
    if (FLAGGED && flags[b] == 0) return;
    const int64_t base = (int64_t)b * P;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x)
        if (ri[base + p] == RI_EMPTY) ri[base + p] = 0u;
}

// ---- fast path: (pixel, depth) list + per-band minimum in LDS ----------------------------------------
// Device-scope atomics on a 134 MB image cost one fabric transaction per point.  Instead every point
// is projected once into an 8-byte (pixel, depth bits) record, and one workgroup per (frame, band of
// BAND_PX pixels) takes the minimum of its band in LDS (ds_min_u32) while streaming the frame's records
// from L2 / Infinity Cache.  Frames that contain a depth-0 point are left to the exact input-order
// passes above.
#define BAND_PX 32768  // 128 KiB of LDS
#define BAND_WG_PER_XCD 32   // persistent workgroups per XCD (32 CUs): one per CU with 128 KiB bands
#define BAND_THREADS 1024
#define BAND_INFLIGHT 4   // 16-byte record pairs per thread in flight

// ---- screened fast path of the pixel computation -----------------------------------------------------------
// project_point() costs ~330 VALU instructions per wavefront, almost all of it the two fdlibm atan2f
// sequences and seven correctly rounded divisions -- but only the INTEGER pixel (row, col) is kept.  The fast
// path computes the two angles with ~1e-6 rad accuracy (hardware rcp/sqrt, fma Horner polynomial), and
// accepts its pixel only when both pre-rounding coordinates are farther from a rounding boundary than a
// margin of PIX_MARGIN x the worst-case discrepancy to the reference arithmetic (error budget: DESIGN.md "Projection").
// Points that are not certain (about 3 % for 64x2048; every point with a special value, a zero depth or
// column 0 / W) are queued in LDS and recomputed by the exact sequence.  The depth itself -- the payload --
// is always the reference's sqrtf(x*x + y*y + z*z).  rpcc_project_fastpath_check() counts disagreements
// between "certain" fast pixels and the exact ones (tests: must be 0).
struct PixFastCfg {
    float kcol, krow, vmin;  // W / hfov, 1 / vres
    float ccol, crow;        // 0.5 - margin
    int on;
};
#define PIX_ANGLE_ERR 2.0e-6f
#define PIX_REL_ERR 6.0e-7f
// Safety factor on the analytic budget.  8 until round 5; the largest discrepancy ever observed (8e8 points per geometry, the adversarial sets of
// test_projection_fast_path_never_disagrees) is 0.13 of the budget in the column coordinate and 0.17 in the row, so 4 still keeps the margin 24-30 x
// above it -- and halves the points that take the exact sequence.  That matters for REAL sweeps only: the synthetic generator jitters a ray inside
// +-0.45 of its cell, 0.05 % of its points are uncertain at any factor, while a stored sweep's points are spread evenly over the cell (3.4 % at 8:
// the exact sequence was a quarter of the pixel kernel's instructions on the example sweep, 68.7 M against 52.6 M on synthetic ones).
#define PIX_MARGIN 4.0f
static PixFastCfg pix_fast_cfg(const rpcc_geom g) {
    PixFastCfg c;
    const float vres = (g.vertical_max - g.vertical_min) / (float)(g.H - 1);
    c.kcol = (float)g.W / g.horizontal_fov;
    c.krow = 1.0f / vres;
    c.vmin = g.vertical_min;
    const float mcol = PIX_MARGIN * (PIX_ANGLE_ERR * fabsf(c.kcol) + PIX_REL_ERR * (float)g.W);
    const float mrow = PIX_MARGIN * ((PIX_ANGLE_ERR + 1.0e-7f) * fabsf(c.krow) + PIX_REL_ERR * ((float)g.H + fabsf(c.vmin * c.krow)));
    c.ccol = 0.5f - mcol;
    c.crow = 0.5f - mrow;
    c.on = (g.H >= 2 && g.W >= 2 && g.horizontal_fov > 0.0f && vres > 0.0f && mcol < 0.2f && mrow < 0.2f &&
            mcol == mcol && mrow == mrow) ? 1 : 0;
    return c;
}

// atan(mn / mx) for 0 <= mn <= mx, mx > 0 (result in [0, pi/4]), absolute error < 5e-7 (polynomial 2e-8; rcp and the reduction the rest)
__device__ __forceinline__ float fast_atan_pos(float mn, float mx) {
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const bool red = t > 0.41421356f;  // tan(pi/8): atan(t) = pi/4 + atan((t-1)/(t+1))
    const float u = red ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t;
    const float z = u * u;
    // (1 - atan(u) / u) / z on z <= tan(pi/8)^2 = 0.1716 as a degree-5 polynomial (Chebyshev fit, coefficients rounded to fp32):
    // evaluated in fp32 the result is within 1.8e-8 of atan(u) -- the rounding of the result itself (half an ulp at 0.39 is
    // 1.5e-8); fdlibm's eleven coefficients, used here until round 3, give the same 1.8e-8 with five more fmas.
    float p = -5.0542026758e-02f;
    p = __builtin_fmaf(p, z, 8.6272865534e-02f);
    p = __builtin_fmaf(p, z, -1.1071758717e-01f);
    p = __builtin_fmaf(p, z, 1.4284175634e-01f);
    p = __builtin_fmaf(p, z, -1.9999977946e-01f);
    p = __builtin_fmaf(p, z, 3.3333334327e-01f);
    const float r = __builtin_fmaf(-u, z * p, u);
    return red ? 0.78539816f + r : r;
}

// Correctly rounded sqrtf for 2^-96 <= s < 2^127 (finite, normal result): the compiler's own expansion of sqrtf -- v_sqrt_f32 (1 ulp)
// followed by the two residual tests s - (r -+ 1 ulp) r that pick the neighbour when it is the better rounding -- without the
// rescaling of tiny arguments and the zero / inf selection in front and behind it (seven of its sixteen instructions), which
// the caller's range test makes unnecessary.
__device__ __forceinline__ float sqrt_rn_normal(float s) {
    const float r = __builtin_amdgcn_sqrtf(s);
    const float rd = u2f(f2u(r) - 1u), ru = u2f(f2u(r) + 1u);
    const float ed = __builtin_fmaf(-rd, r, s), eu = __builtin_fmaf(-ru, r, s);
    float q = ed <= 0.0f ? rd : r;
    q = eu > 0.0f ? ru : q;
    return q;
}

// -> true when pix is certainly the reference's pixel
__device__ __forceinline__ bool project_point_fast(float x, float y, float z, const rpcc_geom g, const PixFastCfg c, int &pix,
                                                   float *colf_out = nullptr, float *rowf_out = nullptr, int *row_out = nullptr, int *col_out = nullptr) {
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    // max / min of non-negative floats as unsigned integers (one instruction each; fmaxf / fminf on the loaded values would be
    // preceded by a canonicalising v_max x, x).  A NaN orders above everything: mx is then NaN and `ok` false.
    const float mx = u2f(max(f2u(ax), f2u(ay))), mn = u2f(min(f2u(ax), f2u(ay)));
    // 2^-46 .. 2^60; false for NaN / inf / the origin (every coordinate is tested itself as well).  The
    // lower bound also keeps x*x + y*y + z*z >= 2^-92 for sqrt_rn_normal() (the depth of a point that passes)
    bool ok = mx >= 1.5e-14f && ax <= 1.15e18f && ay <= 1.15e18f && az <= 1.15e18f;
    float a = fast_atan_pos(mn, mx);
    a = ay > ax ? 1.57079633f - a : a;
    a = x < 0.0f ? 3.14159265f - a : a;
    a = y < 0.0f ? 6.28318531f - a : a;  // = az + 2*pi of :448-449
    const float colf = a * c.kcol;
    const float c0 = rintf(colf);
    ok = ok && fabsf(colf - c0) < c.ccol && c0 >= 1.0f && c0 <= (float)(g.W - 1);  // columns 0 / W (wrap) go the exact way
    const float rho = __builtin_amdgcn_sqrtf(__builtin_fmaf(x, x, y * y));
    float e = fast_atan_pos(u2f(min(f2u(az), f2u(rho))), u2f(max(f2u(az), f2u(rho))));   // (both >= 0 -- or NaN, then not `ok`)
    e = az > rho ? 1.57079633f - e : e;
    e = z < 0.0f ? -e : e;
    const float rowf = fminf(fmaxf((e - c.vmin) * c.krow, 0.0f), (float)(g.H - 1));  // the clamp of :455-458 first
    const float r0 = rintf(rowf);
    ok = ok && fabsf(rowf - r0) < c.crow;
    pix = (int)r0 * g.W + (int)c0;
    if (colf_out) { *colf_out = colf; *rowf_out = (e - c.vmin) * c.krow; }
    if (row_out) { *row_out = (int)r0; *col_out = (int)c0; }
    return ok;
}

// A persistent 256-thread workgroup walks over 256-point chunks (small workgroups: the kernel shares the CUs with
// the one-workgroup-per-frame kernels of the neighbouring batches, a 1024-thread workgroup would wait for 16 free
// wave slots).  Uncertain points collect in an LDS queue that is drained by the exact sequence 256 at a time, i.e.
// with full wavefronts.
#define PIX_THREADS 256
// What the first kernel of a fused batch (project_pix_kernel) sets up besides its own work, so that a batch needs no
// separate initialisation launch: the planar copy of the ray table (its z plane is read by the band kernel), the info
// counters of the ground mask, cleared RANSAC candidate counts and label sums (all consumed by LATER launches only).
// The per-frame "holds a depth-0 point" flags are read and written by the projection kernels themselves, so they are
// not cleared but compared with a mark that changes from call to call: a flag is set when it holds *epoch + 1, and the
// last kernel of the batch increments *epoch (a word of the caller's workspace).  Stale or never-initialised flag words
// can at worst equal the mark by chance, which sends a frame through the exact fix-up path: slower, same result.
struct ZeroRange { uint32_t *p; int n; };
struct BatchInit {
    const float *tm; float *soa; int P;
    int32_t *info; int B;
    ZeroRange z0, z1, z2;
    int on;
};
__device__ __forceinline__ int flag_mark(const int32_t *epoch) { return epoch ? *epoch + 1 : 1; }

// Records binned by (frame, band).  project_pix_kernel walks the points in chunks of 2048 that never cross a frame end (chunk ids:
// frame f owns the ids from (offs[f] - base) / 2048 + f on, one more than it can need) and writes every point's (pixel, depth bits)
// record straight into the share of its wavefront and chunk (a wavefront handles 512 points of a chunk): 512 slots per image band
// (BAND_PX pixels), filled from slot 0 in the order the LDS counter of the band hands out -- no reservation in global memory, no
// device atomics and no register copy of the records.  counts[chunk][wavefront][band] says how
// many there are.  A band workgroup then streams only its own records instead of testing the frame's whole list (three of four
// records were another band's).  The few points of the exact path (queued, projected later by whichever lanes drain the queue)
// go to a second set of lists, one per frame and band pair with room for all the frame's points, reserved through cursors.
// The order inside a list is whatever it is -- the band kernel takes minima, which do not care.
#define PIX_PPT 8  // points per thread and chunk: loads in flight, and one pair of barriers per PIX_PPT * 256 points
#define PIX_CH_SHIFT 11
#define PIX_WAVES (PIX_THREADS / 64)
#define SUB_CAP (PIX_PPT * PIX_THREADS)   // slots per chunk and band
#define SHARE_BYTES (SUB_CAP * 6)   // 6-byte records: a depth u32 array + a pixel-in-band u16 array per share
#define PIX_MAX_BANDS 8   // bands per frame the binned path handles (a power of two)
#define BAND_ROUND 256     // shares (chunks) a band workgroup queues at a time
// the records are binned by image bands of BIN_PX pixels; a band workgroup handles BAND_PX <= BIN_PX of them (its LDS band) and
// filters its bin's records when the two differ
#define BIN_SHIFT 15
#define BIN_PX (1 << BIN_SHIFT)
static_assert(BAND_PX <= BIN_PX && BIN_PX % BAND_PX == 0, "a band workgroup's pixels lie in one bin");
static_assert((1 << PIX_CH_SHIFT) == PIX_PPT * PIX_THREADS, "chunk size");
struct BandBins {
    uint32_t *counts;    // [chunk ids][nbe]
    char *lists;         // [chunk ids][nbe] shares of SHARE_BYTES: depth bits u32 [SUB_CAP] | pixel inside the band u16 [SUB_CAP]
    uint32_t *ocursor;   // [B][nbe]   records of the exact path (cleared per launch)
    uint2 *olists;       // frame f: nbe / 2 regions of cap_f = n_f rounded up to even slots from slot (nbe / 2) * (offs[f] - base + f)
    int nbe;             // bands per frame, rounded up to even
};
__device__ __forceinline__ int64_t chunk_first(const int64_t *__restrict__ offs, int64_t base, int f) {
    return ((offs[f] - base) >> PIX_CH_SHIFT) + f;
}

// one record per lane, any mix of bins in the wavefront (the exact path's records): the lanes of one bin elect a leader, which
// reserves the bin's slots with ONE atomic; all leaders' atomics are in flight together.  Every lane of the wavefront calls.
// ostart / cap: the lane's pair region in olists; odd bands fill it from the top.
__device__ __forceinline__ void bin_append_any(bool valid, uint32_t bin, int64_t ostart, uint32_t cap, uint32_t pix, uint32_t dep, const BandBins bb) {
    const int lane = threadIdx.x & 63;
    unsigned long long pend = __ballot(valid);
    uint32_t r = 0u, cnt = 0u;
    int ldr = 0;
    while (pend) {
        const int l = (int)__ffsll((long long)pend) - 1;
        const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)bin, l);
        const bool mine = valid && bin == kb;
        const unsigned long long m = __ballot(mine);
        if (mine) { r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); cnt = (uint32_t)__popcll(m); ldr = l; }
        pend &= ~m;
    }
    uint32_t b0 = 0u;
    if (valid && lane == ldr) b0 = atomicAdd(&bb.ocursor[bin], cnt);
    b0 = (uint32_t)__shfl((int)b0, ldr, 64);
    const uint32_t x = b0 + r;
    if (valid) bb.olists[ostart + ((bin & 1u) ? cap - 1u - x : x)] = make_uint2(pix, dep);   // (nbe is even: bin & 1 == band & 1)
}

// the exact sequence for one point (every lane of the wavefront calls; `active` = the lane holds a queued point)
__device__ __forceinline__ void project_exact_record(const float *__restrict__ xyz, const int64_t *__restrict__ offs, int64_t base,
                                                     int B, const rpcc_geom g, bool active, int64_t il, const BandBins bb,
                                                     int32_t *__restrict__ flags, int mark, int ps) {
    bool valid = false;
    uint32_t bin = 0u, cap = 0u, pix = 0u, dep = 0u;
    int64_t ostart = 0;
    if (active) {
        const int64_t i = base + il;
        const float x = xyz[ps * i], y = xyz[ps * i + 1], z = xyz[ps * i + 2];
        const RowCol rc = project_point(x, y, z, g);
        if (fabsf(rc.depth) <= 3.402823466e+38f) {
            const int b = find_frame(offs, B, i);
            if (rc.depth == 0.0f) {
                flags[b] = mark; flags[B] = mark;
            } else {
                const int band = rc.pix >> BIN_SHIFT;
                const int64_t o0 = offs[b] - base, o1 = offs[b + 1] - base;
                valid = true; pix = (uint32_t)rc.pix; dep = f2u(rc.depth);
                bin = (uint32_t)(b * bb.nbe + band);
                cap = (uint32_t)((o1 - o0 + 1) & ~(int64_t)1);
                ostart = (int64_t)(bb.nbe >> 1) * (o0 + b) + (int64_t)(band >> 1) * cap;
            }
        }
    }
    bin_append_any(valid, bin, ostart, cap, pix, dep, bb);
}

// the batch's small initialisations, grid-stride over nthr threads (this thread: t)
__device__ __forceinline__ void batch_init(const BatchInit &init, int nthr, int t) {
    for (int p = t; p < init.P; p += nthr) {
        init.soa[p] = init.tm[3 * p]; init.soa[init.P + p] = init.tm[3 * p + 1]; init.soa[2 * (int64_t)init.P + p] = init.tm[3 * p + 2];
    }
    for (int p = t; p < init.B; p += nthr) {
        init.info[RPCC_INFO * p] = 0; init.info[RPCC_INFO * p + 1] = init.P; init.info[RPCC_INFO * p + 2] = 0; init.info[RPCC_INFO * p + 3] = 0;
        init.info[RPCC_INFO * p + 4] = init.P; init.info[RPCC_INFO * p + 5] = 0; init.info[RPCC_INFO * p + 6] = 0; init.info[RPCC_INFO * p + 7] = 0;
    }
    for (int p = t; p < init.z0.n; p += nthr) init.z0.p[p] = 0u;
    for (int p = t; p < init.z1.n; p += nthr) init.z1.p[p] = 0u;
    for (int p = t; p < init.z2.n; p += nthr) init.z2.p[p] = 0u;
}
// (the device-atomic projection path has no pixel kernel to carry the initialisations)
__global__ __launch_bounds__(256) void batch_init_kernel(BatchInit init) { batch_init(init, gridDim.x * 256, blockIdx.x * 256 + threadIdx.x); }

#define PIX_WG_PER_CU 16   // grid = 256 x this many persistent workgroups (6 / 8 / 32 measured: kernel alone 126 / 126 / 117 against 120 us, no change in flight)
#define PIX_VGPR_ATTR __attribute__((amdgpu_waves_per_eu(6, 8)))   // 80 VGPRs (81 without: one wavefront per SIMD less)
template <int PS>   // floats per point: 3 (packed xyz) or 4 (x, y, z, intensity rows as stored in a KITTI .bin: one 16-byte load per point)
__global__ __launch_bounds__(PIX_THREADS) PIX_VGPR_ATTR void project_pix_kernel(const float *__restrict__ xyz, const int64_t *__restrict__ offs,
                                                                  int64_t total, int64_t base, int B, rpcc_geom g, PixFastCfg cfg,
                                                                  BandBins bb, int32_t *__restrict__ flags,
                                                                  const int32_t *__restrict__ epoch, BatchInit init,
                                                                  const int32_t *__restrict__ accept) {
    __shared__ uint32_t queue[(PIX_PPT + 1) * PIX_THREADS];   // launch-relative point indices (launch_project: total < 2^32)
    __shared__ uint32_t qn;
    __shared__ uint32_t bcnt[PIX_MAX_BANDS];   // records per band of the chunk so far
    if (threadIdx.x == 0) qn = 0u;
    if (threadIdx.x < PIX_MAX_BANDS) bcnt[threadIdx.x] = 0u;
    const int mark = flag_mark(epoch);
    if (init.on) batch_init(init, gridDim.x * PIX_THREADS, blockIdx.x * PIX_THREADS + threadIdx.x);   // (nothing of it is read by this launch)
    if (accept) {   // every frame taken by project_ordered_kernel (a batch of stored sweeps): nothing to walk
        int rej = 0;    // (every wavefront looks at all B words and finds the same answer: no barrier, no LDS)
        for (int i = threadIdx.x & 63; i < B; i += 64) rej |= accept[i] == 0;
        if (__ballot(rej) == 0ull) return;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int64_t nck = (total >> PIX_CH_SHIFT) + B;   // chunk ids
    // chunk id -> frame, first point (launch-relative), points; the ids of a workgroup ascend, so the search starts at the last frame
    auto locate = [&](int64_t k, int &f, int64_t &first, int64_t &room) {
        int lo = f, hi = B;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (chunk_first(offs, base, mid) <= k) lo = mid; else hi = mid;
        }
        f = lo;
        const int64_t o0 = offs[f] - base, o1 = offs[f + 1] - base;
        first = o0 + ((k - ((o0 >> PIX_CH_SHIFT) + f)) << PIX_CH_SHIFT);
        room = min((int64_t)(PIX_PPT * PIX_THREADS), o1 - first);
        if (accept && accept[f]) room = 0;   // the frame was projected by project_ordered_kernel (the band kernel skips it too)
    };
    int fnext = 0;
    int64_t il_next = 0, room_next = 0;
    if ((int64_t)blockIdx.x < nck) locate(blockIdx.x, fnext, il_next, room_next);
    for (int64_t k = blockIdx.x; k < nck; k += gridDim.x) {
        const int64_t il0 = il_next, room64 = room_next;
        uint32_t *cnt_out = bb.counts + k * bb.nbe;
        if (room64 <= 0) {   // (workgroup-uniform) an id the frame does not need
            if ((int)threadIdx.x < bb.nbe) cnt_out[threadIdx.x] = 0u;
            if (k + gridDim.x < nck) locate(k + gridDim.x, fnext, il_next, room_next);
            continue;
        }
        float x[PIX_PPT], y[PIX_PPT], z[PIX_PPT];
        // unconditional loads, all in flight: wave-uniform chunk base + 32-bit lane offsets; only a frame's last chunk clamps its indices
        const float *cb = xyz + PS * (base + il0);
        const uint32_t room = (uint32_t)room64;   // points of this chunk (>= 1)
#pragma unroll
        for (int u = 0; u < PIX_PPT; u++) {
            uint32_t i = (uint32_t)(u * PIX_THREADS) + threadIdx.x;
            if (room < (uint32_t)(PIX_PPT * PIX_THREADS)) i = min(i, room - 1u);   // (wave-uniform test)
            if (PS == 4) { const float4 p4 = ld_at(reinterpret_cast<const float4 *>(cb), i * 16u); x[u] = p4.x; y[u] = p4.y; z[u] = p4.z; }
            else { const f32x3 p3 = ld_at(reinterpret_cast<const f32x3 *>(cb), i * 12u); x[u] = p3.x; y[u] = p3.y; z[u] = p3.z; }
        }
        if (k + gridDim.x < nck) locate(k + gridDim.x, fnext, il_next, room_next);   // (scalar loads: their latency passes under the point loads)
        char *reg = bb.lists + k * (int64_t)(bb.nbe * SHARE_BYTES);   // the chunk's shares of the lists
#pragma unroll
        for (int u = 0; u < PIX_PPT; u++) {
            const uint32_t ci = (uint32_t)(u * PIX_THREADS) + threadIdx.x;
            // straight-line: the lane masks of `fast` / `slow` stay in scalar registers (a boolean set inside divergent branches is
            // materialised in a VGPR and compared again for the ballot)
            int pix;
            const bool in = ci < room;
            const bool fast = project_point_fast(x[u], y[u], z[u], g, cfg, pix) && cfg.on && in;
            const float depth = sqrt_rn_normal(x[u] * x[u] + y[u] * y[u] + z[u] * z[u]);   // :446 (2^-92 <= the sum < 2^122 when `fast`: == sqrtf)
            // the record's slot in its band's run: the workgroup's LDS counter of the band (ds_add_rtn: the LDS pipe ranks, not the VALU)
            const uint32_t band = (uint32_t)pix >> BIN_SHIFT;
            if (fast) {
                const uint32_t xr = atomicAdd(&bcnt[band & (PIX_MAX_BANDS - 1)], 1u);
                const uint32_t so = band * (uint32_t)SHARE_BYTES;   // (32-bit offsets from the chunk's wave-uniform base: scalar-base stores)
                // 6 bytes per record: the depth bits and the pixel's offset in its band
                st_at(reinterpret_cast<uint32_t *>(reg), so + xr * 4u, f2u(depth));
                st_at(reinterpret_cast<uint16_t *>(reg), so + (uint32_t)(SUB_CAP * 4) + xr * 2u, (uint16_t)((uint32_t)pix & (BIN_PX - 1)));
            }
            const bool slow = in && !fast;
            const unsigned long long sm = __ballot(slow);
            if (sm) {
                const int leader = (int)__ffsll((long long)sm) - 1;
                uint32_t q0 = 0u;
                if (lane == leader) q0 = atomicAdd(&qn, (uint32_t)__popcll(sm));
                q0 = (uint32_t)__builtin_amdgcn_readlane((int)q0, leader);
                if (slow) queue[q0 + __popcll(sm & lt)] = (uint32_t)(il0 + ci);
            }
        }
        __syncthreads();
        uint32_t n = qn;
        if ((int)threadIdx.x < bb.nbe) { cnt_out[threadIdx.x] = bcnt[threadIdx.x]; bcnt[threadIdx.x] = 0u; }
        __syncthreads();
        if (n >= PIX_THREADS) {  // full workgroups of uncertain points: the exact sequence
            while (n >= PIX_THREADS) {
                project_exact_record(xyz, offs, base, B, g, true, (int64_t)queue[n - PIX_THREADS + threadIdx.x], bb, flags, mark, PS);
                n -= PIX_THREADS;
            }
            __syncthreads();
            if (threadIdx.x == 0) qn = n;
            __syncthreads();
        }
    }
    const uint32_t n = qn;
    if ((threadIdx.x & ~63u) < n) project_exact_record(xyz, offs, base, B, g, threadIdx.x < n, (int64_t)queue[min(threadIdx.x, n - 1u)], bb, flags, mark, PS);
}

// test hook: counts[0] = points the fast path is certain about, counts[1] = of those, points whose pixel differs
// from the exact sequence (must be 0), counts[2] = points sent to the exact sequence, counts[3] / counts[4] = largest
// |fast - exact| of the pre-rounding column / row coordinate over the points of ordinary magnitude, in units of 1e-9
// (the error budget the margins are derived from)
__global__ __launch_bounds__(256) void project_fastcheck_kernel(const float *__restrict__ xyz, int64_t total, rpcc_geom g,
                                                                PixFastCfg cfg, unsigned long long *__restrict__ counts) {
    unsigned long long sure = 0ull, bad = 0ull, slow = 0ull;
    float dc = 0.0f, dr = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        int pix;
        float cf, rf;
        const bool ok = project_point_fast(x, y, z, g, cfg, pix, &cf, &rf) && cfg.on;
        const RowCol rc = project_point(x, y, z, g);
        const float mx = fmaxf(fabsf(x), fabsf(y));
        if (mx >= 8.7e-19f && fabsf(x) <= 1.15e18f && fabsf(y) <= 1.15e18f && fabsf(z) <= 1.15e18f) {
            float d = fabsf(cf - rc.colf);
            d = fminf(d, fabsf(d - (float)g.W));  // azimuth 0 == 2*pi
            dc = fmaxf(dc, d);
            if (fabsf(rc.rowf) < 1.0e6f) dr = fmaxf(dr, fabsf(rf - rc.rowf));
        }
        if (ok) {
            sure++;
            if (!(fabsf(rc.depth) <= 3.402823466e+38f) || rc.depth == 0.0f || rc.pix != pix) bad++;
        } else {
            slow++;
        }
    }
    const int s = wave_sum_i32((int)sure), b2 = wave_sum_i32((int)bad), sl = wave_sum_i32((int)slow);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], (unsigned long long)s);
        atomicAdd(&counts[1], (unsigned long long)b2);
        atomicAdd(&counts[2], (unsigned long long)sl);
    }
    atomicMax(&counts[3], (unsigned long long)(dc * 1.0e9f));
    atomicMax(&counts[4], (unsigned long long)(dr * 1.0e9f));
}

// RANSAC hand-off (optional, zcnt != nullptr): while the final image leaves LDS the kernel also counts, per chunk of
// rs_chunk pixels (the pixel range one wavefront of ground_ransac_kernel owns), the pixels with z = r * tz < zthr, i.e.
// the first pass of the ground candidate selection.  zcnt[b][0..15] counts, zcnt[b][16] = 1 when they are valid
// (frames with a depth-0 point are re-projected afterwards and count for themselves).
#define RS_CHUNKS 16
static inline int rs_chunk_px(int P) { return (((P + RS_CHUNKS - 1) / RS_CHUNKS) + 63) & ~63; }
// Beside the counts (P % 64 == 0: the quad write-out, whole 64-pixel words): WHICH pixels have z < zthr, a byte per quad of pixels (bit e = pixel 4 q + e),
// [B][P / 4] bytes behind the B x (RS_CHUNKS + 1) count words (16-byte aligned); zcnt[b][RS_CHUNKS] = 3 when counts and bytes are valid (1: counts only).
// The ground fit compacts its candidates from these bytes instead of testing every pixel again.
__host__ __device__ __forceinline__ size_t rs_count_words(int B) { return (size_t)B * (RS_CHUNKS + 1); }
__host__ __device__ __forceinline__ uint8_t *rs_zmask_of(const int32_t *zcnt, int B) {
    return reinterpret_cast<uint8_t *>(const_cast<int32_t *>(zcnt)) + ((rs_count_words(B) * 4 + 15) & ~(size_t)15);
}
__device__ void project_fixup_frame(const float *__restrict__ xyz, const int64_t *__restrict__ offs, int b, rpcc_geom g,
                                    uint32_t *__restrict__ ri, int32_t *__restrict__ lastz, int ps);

// band_wgs: the workgroups below that id are band workgroups; the B workgroups from there on (present when the launch has
// points) run the exact input-order projection of the frames that hold a depth-0 point (project_fixup_frame) -- a no-op
// for every other frame, and the band workgroups skip those frames -- so the fix-up costs no launch of its own.
#define BAND_VGPR_ATTR __attribute__((amdgpu_waves_per_eu(6, 8)))   // 80 VGPRs instead of 81 (no spill): +0.5 % with batches in flight
__global__ __launch_bounds__(BAND_THREADS) BAND_VGPR_ATTR void project_band_kernel(const BandBins bb,
                                                                    const int64_t *__restrict__ offs, int64_t base,
                                                                    int B, int P, uint32_t *__restrict__ ri,
                                                                    const int32_t *__restrict__ flags,
                                                                    const float *__restrict__ tz, float zthr, int rs_chunk,
                                                                    int32_t *__restrict__ zcnt, const int32_t *__restrict__ epoch,
                                                                    int band_wgs, const float *__restrict__ xyz, rpcc_geom g,
                                                                    int32_t *__restrict__ lastz, int ps, const int32_t *__restrict__ accept) {
    extern __shared__ __attribute__((aligned(16))) uint32_t band[];  // [BAND_PX]
    __shared__ uint16_t ldq[BAND_ROUND * 16];   // queued loads: share of the round << 2 | 64-pair step
    __shared__ uint16_t cntl[BAND_ROUND];          // records per share
    __shared__ uint32_t nslots;
    const int mark = flag_mark(epoch);
    if ((int)blockIdx.x >= band_wgs) {
        const int fb = (int)blockIdx.x - band_wgs;
        if (flags[fb] == mark) project_fixup_frame(xyz, offs, fb, g, ri, lastz, ps);
        return;
    }
    // XCD-aware mapping: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2).  The nbands
    // workgroups of a frame all stream the same record list, so they are placed on ONE XCD (ids x, x+8, x+16, ...):
    // XCD x serves the frames b = x (mod 8), and the list is fetched into that L2 once instead of nbands times.
    // The workgroups are persistent: workgroup (xcd, slot) takes the items slot, slot + slots, ... of its XCD (same band, the
    // next frames), so a 1024-thread / 128 KB workgroup is launched once per CU instead of once per item.
    if (accept) {   // every frame taken by project_ordered_kernel: no band to build
        int rej = 0;
        for (int i = threadIdx.x & 63; i < B; i += 64) rej |= accept[i] == 0;
        if (__ballot(rej) == 0ull) return;
    }
    const int nbands = (P + BAND_PX - 1) / BAND_PX;
    const int xcd = blockIdx.x & 7, slots = band_wgs >> 3;
    // the LDS band is cleared once; every item's write-out leaves it cleared for the next
    {
        uint4 *b4 = reinterpret_cast<uint4 *>(band);
        for (uint32_t q = threadIdx.x; q < BAND_PX / 4; q += BAND_THREADS) b4[q] = make_uint4(RI_EMPTY, RI_EMPTY, RI_EMPTY, RI_EMPTY);
    }
    for (int slot = blockIdx.x >> 3;; slot += slots) {
    const int b = xcd + 8 * (slot / nbands), kband = slot % nbands;
    if (b >= B) break;
    if (accept && accept[b]) continue;   // (workgroup-uniform) image, candidate counts and bytes are project_ordered_kernel's
    const int flagged = flags[b] == mark;
    const int64_t o0 = offs[b] - base, o1 = offs[b + 1] - base;
    const int bin = (int)(((uint32_t)kband * BAND_PX) >> BIN_SHIFT);   // the record bin this band lies in
    const uint32_t sub0 = ((uint32_t)kband * BAND_PX) & (BIN_PX - 1);  // the band's first pixel inside the bin
    const uint32_t ocnt = bb.ocursor[b * bb.nbe + bin];          // records of the exact path
    const int64_t c0 = (o0 >> PIX_CH_SHIFT) + b;                   // the frame's chunk ids: c0 .. (o1 >> 11) + b
    const int S = (int)(((o1 >> PIX_CH_SHIFT) + b + 1) - c0);   // shares (one per chunk of the pixel kernel)
    const int pairb = bin >> 1;
    const uint32_t odd = (uint32_t)bin & 1u;
    const uint32_t band0 = (uint32_t)kband * BAND_PX;
    const uint32_t npx = min((uint32_t)BAND_PX, (uint32_t)P - band0);
    // (issued before the band is cleared: the latency passes under the LDS stores)
    uint32_t mycnt = 0u;
    if ((int)threadIdx.x < min(S, BAND_ROUND) && !flagged) mycnt = min(bb.counts[(c0 + threadIdx.x) * bb.nbe + bin], (uint32_t)SUB_CAP);
    if (threadIdx.x == 0) nslots = 0u;   // (every thread has read the previous item's value: a barrier lies between)
    if (flagged) continue;   // (workgroup-uniform) a frame with a depth-0 point: project_fixup_kernel
    // The band's records of a share are a run of 0 .. 2048 slots, read four at a time (16 bytes of depths + 8 bytes of pixels per lane),
    // 256 records per wavefront and step; every share first queues the loads it needs (`ldq`), then the
    // wavefronts take BAND_INFLIGHT of them at a time.
    for (int g0 = 0; g0 < S; g0 += BAND_ROUND) {   // (one round unless the frame has more than half a million points)
        if (g0) {
            __syncthreads();   // the previous round's queue is consumed
            if (threadIdx.x == 0) nslots = 0u;
            mycnt = (int)threadIdx.x < min(S - g0, BAND_ROUND) ? min(bb.counts[(c0 + g0 + threadIdx.x) * bb.nbe + bin], (uint32_t)SUB_CAP) : 0u;
        }
        __syncthreads();   // the band is cleared (the previous item's write-out is complete), the queue is empty
        if ((int)threadIdx.x < min(S - g0, BAND_ROUND)) {
            cntl[threadIdx.x] = (uint16_t)mycnt;
            const uint32_t nq = (mycnt + 255u) >> 8;   // <= 8
            uint32_t pos = nq ? atomicAdd(&nslots, nq) : 0u;
            for (uint32_t q = 0; q < nq; q++) ldq[pos + q] = (uint16_t)((threadIdx.x << 4) | q);
        }
        __syncthreads();
        const uint32_t ns = nslots;
        const char *gl = bb.lists + ((c0 + g0) * bb.nbe + bin) * (int64_t)SHARE_BYTES;   // share 0 of the round
        const uint32_t share_stride = (uint32_t)bb.nbe * SHARE_BYTES;
        for (uint32_t i = (threadIdx.x >> 6) * BAND_INFLIGHT; i < ns; i += (BAND_THREADS >> 6) * BAND_INFLIGHT) {
            uint4 dv[BAND_INFLIGHT];
            uint2 pv[BAND_INFLIGHT];
            uint32_t left[BAND_INFLIGHT];   // records of the run from this lane's first on (0: none)
#pragma unroll
            for (int u = 0; u < BAND_INFLIGHT; u++) {  // unconditional (clamped) loads; records beyond the run are masked
                const uint32_t e = ldq[min(i + u, ns - 1u)];
                const uint32_t sl = e >> 4;
                const uint32_t cnt = cntl[sl];
                const uint32_t p = ((e & 15u) << 6) + (threadIdx.x & 63u);   // group of four records
                const uint32_t pc = min(p, (cnt - 1u) >> 2);
                dv[u] = ld_at(reinterpret_cast<const uint4 *>(gl), sl * share_stride + pc * 16u);
                pv[u] = ld_at(reinterpret_cast<const uint2 *>(gl), sl * share_stride + (uint32_t)(SUB_CAP * 4) + pc * 8u);
                left[u] = (i + u < ns && 4u * p < cnt) ? cnt - 4u * p : 0u;
            }
#pragma unroll
            for (int u = 0; u < BAND_INFLIGHT; u++) {
                if (BAND_PX == BIN_PX) {
                    if (left[u] > 0u) atomicMin(&band[pv[u].x & 0xFFFFu], dv[u].x);
                    if (left[u] > 1u) atomicMin(&band[pv[u].x >> 16], dv[u].y);
                    if (left[u] > 2u) atomicMin(&band[pv[u].y & 0xFFFFu], dv[u].z);
                    if (left[u] > 3u) atomicMin(&band[pv[u].y >> 16], dv[u].w);
                } else {   // the bin's records of the other bands fall outside [0, npx)
                    const uint32_t r0 = (pv[u].x & 0xFFFFu) - sub0, r1 = (pv[u].x >> 16) - sub0, r2 = (pv[u].y & 0xFFFFu) - sub0, r3 = (pv[u].y >> 16) - sub0;
                    if (left[u] > 0u && r0 < npx) atomicMin(&band[r0], dv[u].x);
                    if (left[u] > 1u && r1 < npx) atomicMin(&band[r1], dv[u].y);
                    if (left[u] > 2u && r2 < npx) atomicMin(&band[r2], dv[u].z);
                    if (left[u] > 3u && r3 < npx) atomicMin(&band[r3], dv[u].w);
                }
            }
        }
    }
    if (ocnt) {   // the exact path's records of this band
        const uint32_t cap = (uint32_t)((o1 - o0 + 1) & ~(int64_t)1);
        const uint2 *ol = bb.olists + (int64_t)(bb.nbe >> 1) * (o0 + b) + (int64_t)pairb * cap + (odd ? cap - min(ocnt, cap) : 0u);
        const uint32_t on = min(ocnt, cap);
        for (uint32_t i = threadIdx.x; i < on; i += BAND_THREADS * 2) {
            const uint2 ra = ol[i], rb2 = ol[min(i + BAND_THREADS, on - 1u)];
            const uint32_t r0 = ra.x - band0, r1 = rb2.x - band0;
            if (r0 < npx) atomicMin(&band[r0], ra.y);
            if (i + BAND_THREADS < on && r1 < npx) atomicMin(&band[r1], rb2.y);
        }
    }
    __syncthreads();
    uint32_t *out = ri + (int64_t)b * P + band0;
    if (zcnt == nullptr) {
        for (uint32_t p = threadIdx.x; p < npx; p += BAND_THREADS) {
            const uint32_t v = band[p];
            band[p] = RI_EMPTY;
            out[p] = (v == RI_EMPTY) ? 0u : v;
        }
        continue;
    }
    if ((P & 3) == 0) {
        // write-out, 16 bytes per lane: four consecutive pixels from LDS (ds_read_b128), their ray z (one 16-byte load), one
        // 16-byte store.  A chunk boundary (multiple of 64 pixels) never splits a lane's four pixels; a wavefront's 256
        // pixels lie in one chunk except at a boundary, where the lanes count for themselves.
        __shared__ int zc[RS_CHUNKS];
        if (threadIdx.x < RS_CHUNKS) zc[threadIdx.x] = 0;
        __syncthreads();
        const uint4 *band4 = reinterpret_cast<const uint4 *>(band);
        uint4 *band4w = reinterpret_cast<uint4 *>(band);
        const float4 *tz4 = reinterpret_cast<const float4 *>(tz + band0);
        uint4 *out4 = reinterpret_cast<uint4 *>(out);
        const uint32_t nq = npx >> 2;
        const bool want_bytes = (P & 63) == 0;
        uint8_t *zm = rs_zmask_of(zcnt, B) + (int64_t)b * (P >> 2) + (band0 >> 2);
        // (whole wavefronts stay in the loop: the DPP sum below needs every lane)
        for (uint32_t q0 = threadIdx.x; q0 - (threadIdx.x & 63u) < nq; q0 += BAND_THREADS * 4) {  // 4 quads per lane in flight
            uint4 rv[4];
            float4 zr[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t q = min(q0 + u * BAND_THREADS, nq - 1);  // unconditional (clamped) loads
                zr[u] = tz4[q];
                uint4 v = band4[q];
                if (q0 + u * BAND_THREADS < nq) band4w[q] = make_uint4(RI_EMPTY, RI_EMPTY, RI_EMPTY, RI_EMPTY);
                v.x = v.x == RI_EMPTY ? 0u : v.x; v.y = v.y == RI_EMPTY ? 0u : v.y;
                v.z = v.z == RI_EMPTY ? 0u : v.z; v.w = v.w == RI_EMPTY ? 0u : v.w;
                rv[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t q = q0 + u * BAND_THREADS;
                const bool in = q < nq;
                if (in) out4[q] = rv[u];
                const int t0 = (int)(u2f(rv[u].x) * zr[u].x < zthr), t1 = (int)(u2f(rv[u].y) * zr[u].y < zthr),
                          t2 = (int)(u2f(rv[u].z) * zr[u].z < zthr), t3 = (int)(u2f(rv[u].w) * zr[u].w < zthr);
                const int c = in ? (t0 + t1) + (t2 + t3) : 0;
                if (in && want_bytes) zm[q] = (uint8_t)(t0 | (t1 << 1) | (t2 << 2) | (t3 << 3));
                const uint32_t ch = (band0 + 4u * min(q, nq - 1)) / (uint32_t)rs_chunk;
                const uint32_t ch0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch);
                if (__ballot(ch != ch0) == 0ull) {
                    const int tot = (int)dpp_sum_u32((uint32_t)c);
                    if (tot && (threadIdx.x & 63) == 0) atomicAdd(&zc[ch0], tot);
                } else if (c) {
                    atomicAdd(&zc[ch], c);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < RS_CHUNKS && zc[threadIdx.x]) atomicAdd(&zcnt[b * (RS_CHUNKS + 1) + threadIdx.x], zc[threadIdx.x]);
        if (kband == 0 && threadIdx.x == 0) zcnt[b * (RS_CHUNKS + 1) + RS_CHUNKS] = want_bytes ? 3 : 1;
        continue;
    }
    // a wavefront covers 64 consecutive pixels per step; rs_chunk is a multiple of 64, so the chunk is wave-uniform
    uint32_t g0 = band0 + (threadIdx.x & ~63u);
    uint32_t ch = g0 / (uint32_t)rs_chunk, nb = (ch + 1u) * (uint32_t)rs_chunk;
    int acc = 0;
    for (uint32_t p0 = threadIdx.x; p0 < npx; p0 += BAND_THREADS * 8) {  // 8 ray loads in flight per lane
        float zr[8];
        uint32_t rv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = min(p0 + u * BAND_THREADS, npx - 1);  // unconditional (clamped) loads
            zr[u] = tz[band0 + p];
            const uint32_t v = band[p];
            if (p0 + u * BAND_THREADS < npx) band[p] = RI_EMPTY;
            rv[u] = (v == RI_EMPTY) ? 0u : v;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t p = p0 + u * BAND_THREADS;
            const bool in = p < npx;
            if (in) out[p] = rv[u];
            if (g0 >= nb) {
                if (acc && (threadIdx.x & 63) == 0) atomicAdd(&zcnt[b * (RS_CHUNKS + 1) + ch], acc);
                ch = g0 / (uint32_t)rs_chunk; nb = (ch + 1u) * (uint32_t)rs_chunk; acc = 0;
            }
            acc += __popcll(__ballot(in && u2f(rv[u]) * zr[u] < zthr));
            g0 += BAND_THREADS;
        }
    }
    if (acc && (threadIdx.x & 63) == 0) atomicAdd(&zcnt[b * (RS_CHUNKS + 1) + ch], acc);
    if (kband == 0 && threadIdx.x == 0) zcnt[b * (RS_CHUNKS + 1) + RS_CHUNKS] = 1;
    }
}

// Exact input-order projection of a frame that holds a depth-0 point (flags): one workgroup (the extra workgroups of
// project_band_kernel) does what project_fill<true> / project_kernel<1> / project_kernel<2> / project_finalize<true> do grid-wide
// (all dependencies are inside a frame).  Frames without a flag -- the normal case -- leave at once, so the fast path
// pays one empty launch instead of four.  Loads that follow this kernel's own atomics bypass the CU's L1 (agent scope).
__device__ void project_fixup_frame(const float *__restrict__ xyz, const int64_t *__restrict__ offs, int b, rpcc_geom g,
                                    uint32_t *__restrict__ ri, int32_t *__restrict__ lastz, int ps) {
    const int FIXUP_THREADS = blockDim.x;
    const int P = g.H * g.W;
    uint32_t *img = ri + (int64_t)b * P;
    int32_t *lz = lastz + (int64_t)b * P;
    for (int p = threadIdx.x; p < P; p += FIXUP_THREADS) { img[p] = RI_EMPTY; lz[p] = 0; }
    __threadfence();
    __syncthreads();
    const int64_t i0 = offs[b], i1 = offs[b + 1];
    for (int64_t i = i0 + threadIdx.x; i < i1; i += FIXUP_THREADS) {  // last input position of a depth-0 point per pixel
        const RowCol rc = project_point(xyz[ps * i], xyz[ps * i + 1], xyz[ps * i + 2], g);
        if (fabsf(rc.depth) <= 3.402823466e+38f && rc.depth == 0.0f) atomicMax(&lz[rc.pix], (int32_t)(i - i0) + 1);
    }
    __threadfence();
    __syncthreads();
    for (int64_t i = i0 + threadIdx.x; i < i1; i += FIXUP_THREADS) {  // minimum over the points after that position
        const RowCol rc = project_point(xyz[ps * i], xyz[ps * i + 1], xyz[ps * i + 2], g);
        if (!(fabsf(rc.depth) <= 3.402823466e+38f) || rc.depth == 0.0f) continue;
        const int32_t last = __hip_atomic_load(&lz[rc.pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(i - i0) + 1 > last) atomicMin(&img[rc.pix], f2u(rc.depth));
    }
    __threadfence();
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += FIXUP_THREADS)
        if (__hip_atomic_load(&img[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == RI_EMPTY) img[p] = 0u;
}

#include "project_ordered.h"

static inline int band_bins_even(int P) { return (((P + BIN_PX - 1) / BIN_PX) + 1) & ~1; }   // record bins per frame, rounded up to even
static inline int64_t pix_chunk_ids(int64_t total, int B) { return ((total > 0 ? total : 0) >> PIX_CH_SHIFT) + B; }
static inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
static inline size_t project_small_bytes(int B, int P) { return align16(((size_t)B * ((size_t)P + 8)) * 4 + 256); }   // lastz + flags
static inline size_t project_cursor_bytes(int B, int P) { return align16((size_t)B * band_bins_even(P) * 4); }
static inline size_t project_counts_bytes(int64_t total, int B, int P) { return align16((size_t)pix_chunk_ids(total, B) * band_bins_even(P) * 4); }
static inline size_t project_lists_bytes(int64_t total, int B, int P) { return (size_t)pix_chunk_ids(total, B) * band_bins_even(P) * SHARE_BYTES; }
static size_t project_scratch_bytes(int64_t total, int B, int P) {   // lastz, flags | cursors | counts | shares | exact path's lists
    return project_small_bytes(B, P) + project_cursor_bytes(B, P) + project_counts_bytes(total, B, P) + project_lists_bytes(total, B, P) +
           (size_t)(band_bins_even(P) / 2) * (size_t)((total > 0 ? total : 0) + B) * 8 + 16;
}
extern "C" size_t rpcc_project_scratch_bytes(int64_t total, int B, int P) { return project_scratch_bytes(total, B, P); }

extern "C" int rpcc_project_fastpath_check(const float *xyz, int64_t total, rpcc_geom g, uint64_t *counts, void *stream) {
    ARG_TRY(xyz && counts && total > 0 && g.H > 0 && g.W > 0);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(counts, 0, 5 * sizeof(uint64_t), st));
    project_fastcheck_kernel<<<4096, 256, 0, st>>>(xyz, total, g, pix_fast_cfg(g), reinterpret_cast<unsigned long long *>(counts));
    LAUNCH_CHECK();
    return RPCC_OK;
}

// On return ri is final (0 = empty pixel).  scratch_bytes < rpcc_project_scratch_bytes() selects the
// atomic path, which only needs B*(P+8)*4 bytes.
// order_mode: ORD_MODE_OFF (the default: pixel + band kernels), ORD_MODE_PROBE (a frame whose points come in scanner order is projected by
// project_ordered_kernel, the others by the pixel + band kernels), ORD_MODE_FORCE (test hook: every frame by the former).  tm: the [P,3] ray table
// (needed by the ordered kernel when zcnt is wanted).  accept_out: dev i32 [B] or nullptr -- which frames the ordered kernel took.
#define ORD_MODE_OFF 2
static int launch_project(const float *xyz, const int64_t *offsets, int64_t total, int64_t base, int B, rpcc_geom g,
                          float *ri, void *scratch, size_t scratch_bytes, hipStream_t st, const float *tz_plane = nullptr,
                          int32_t *zcnt = nullptr, const BatchInit *init = nullptr, const int32_t *epoch = nullptr, int ps = 3,
                          int order_mode = ORD_MODE_OFF, const float *tm = nullptr, int32_t *accept_out = nullptr) {
    const bool cleared = init != nullptr;   // fused batch: the pixel kernel initialises, flags are marked by epoch
    const int P = g.H * g.W;
    uint32_t *rb = reinterpret_cast<uint32_t *>(ri);
    int32_t *lastz = reinterpret_cast<int32_t *>(scratch);
    int32_t *flags = lastz + (int64_t)B * P;
    const dim3 fg((P + 1023) / 1024 < 64 ? (P + 1023) / 1024 : 64, B);
    const unsigned nb = (unsigned)((total + 255) / 256);
    const unsigned nb_small = nb < 2048 ? (nb ? nb : 1) : 2048;
    BandBins bb;
    bb.nbe = band_bins_even(P);
    // the binned path: room for the lists, a counter per band and wavefront in the pixel kernel's LDS
    const bool fast = scratch_bytes >= project_scratch_bytes(total, B, P) && bb.nbe <= PIX_MAX_BANDS && total < ((int64_t)1 << 32);
    if (fast) {
        char *q = reinterpret_cast<char *>(scratch) + project_small_bytes(B, P);
        bb.ocursor = reinterpret_cast<uint32_t *>(q); q += project_cursor_bytes(B, P);
        bb.counts = reinterpret_cast<uint32_t *>(q); q += project_counts_bytes(total, B, P);
        bb.lists = q; q += project_lists_bytes(total, B, P);
        bb.olists = reinterpret_cast<uint2 *>(q);
        HIP_TRY(hipMemsetAsync(bb.ocursor, 0, (size_t)B * bb.nbe * 4, st));
        if (!cleared) HIP_TRY(hipMemsetAsync(flags, 0, (size_t)(B + 1) * 4, st));
        BatchInit bi;
        memset(&bi, 0, sizeof(bi));
        if (init) bi = *init;
        if (zcnt && !cleared) HIP_TRY(hipMemsetAsync(zcnt, 0, rs_count_words(B) * 4, st));
        // Sweeps in scanner order: one workgroup per frame probes the order of its points and, if they move through the image ring by
        // ring, projects the frame without records (project_ordered.h); accept[b] tells the two kernels below to leave that frame alone.
        int32_t *accept = nullptr;
        const bool want_z = tz_plane != nullptr && zcnt != nullptr;
        if (order_mode != ORD_MODE_OFF && total > 0 && ordered_geometry_ok(g) && ((uintptr_t)ri & 15u) == 0 &&
            (!want_z || (tm != nullptr && ((uintptr_t)tm & 15u) == 0))) {
            accept = accept_out ? accept_out : flags + (B + 1);   // (the flag area has 8 B words)
            OrdArgs oa;
            oa.xyz = xyz; oa.offs = offsets; oa.base = base; oa.B = B; oa.g = g; oa.cfg = pix_fast_cfg(g); oa.ri = rb; oa.flags = flags; oa.epoch = epoch;
            oa.accept = accept; oa.mode = order_mode; oa.tm = tm; oa.zthr = -1.5f; oa.rs_chunk = rs_chunk_px(P); oa.zcnt = want_z ? zcnt : nullptr;
            if (ps == 4) {
                HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&project_ordered_kernel<4>), ORD_WIN_PX * 4));
                project_ordered_kernel<4><<<B, ORD_THREADS, ORD_WIN_PX * 4, st>>>(oa);
            } else {
                HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&project_ordered_kernel<3>), ORD_WIN_PX * 4));
                project_ordered_kernel<3><<<B, ORD_THREADS, ORD_WIN_PX * 4, st>>>(oa);
            }
            LAUNCH_CHECK();
            if (want_z) bi.z0.n = 0;   // (every frame's candidate counts were cleared -- and maybe filled -- by its workgroup above)
        } else if (accept_out) {
            HIP_TRY(hipMemsetAsync(accept_out, 0, (size_t)B * 4, st));
        }
        // (also for a batch without points: every chunk id's share counts are written by this kernel, the band kernel reads them)
        const unsigned pix_wgs = (unsigned)std::max<int64_t>(std::min<int64_t>(pix_chunk_ids(total, B), 256 * PIX_WG_PER_CU), 1);
        if (ps == 4) project_pix_kernel<4><<<pix_wgs, PIX_THREADS, 0, st>>>(xyz, offsets, total, base, B, g, pix_fast_cfg(g), bb, flags, epoch, bi, accept);
        else project_pix_kernel<3><<<pix_wgs, PIX_THREADS, 0, st>>>(xyz, offsets, total, base, B, g, pix_fast_cfg(g), bb, flags, epoch, bi, accept);
        HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&project_band_kernel), BAND_PX * 4));
        // persistent: at most one workgroup per CU (8 XCDs x 32), each walking over its XCD's (frame, band) items
        const int nbands = (P + BAND_PX - 1) / BAND_PX;
        const int band_wgs = 8 * std::min(((B + 7) / 8) * nbands, std::max(BAND_WG_PER_XCD / nbands, 1) * nbands);
        // + B workgroups for the exact input-order semantics of frames with depth-0 points (a no-op otherwise)
        project_band_kernel<<<band_wgs + (total > 0 ? B : 0), BAND_THREADS, BAND_PX * 4, st>>>(
            bb, offsets, base, B, P, rb, flags, tz_plane, -1.5f, rs_chunk_px(P), tz_plane ? zcnt : nullptr, epoch, band_wgs, xyz, g, lastz, ps, accept);
        LAUNCH_CHECK();
        return RPCC_OK;
    }
    if (accept_out) HIP_TRY(hipMemsetAsync(accept_out, 0, (size_t)B * 4, st));
    if (zcnt && !cleared) HIP_TRY(hipMemsetAsync(zcnt, 0, rs_count_words(B) * 4, st));
    if (init && init->on) batch_init_kernel<<<256, 256, 0, st>>>(*init);   // (clears zcnt: the RANSAC kernel counts for itself)
    project_fill_kernel<false><<<fg, 256, 0, st>>>(rb, lastz, flags, P);
    LAUNCH_CHECK();
    if (total > 0) {
        project_kernel<0><<<nb, 256, 0, st>>>(xyz, offsets, total, base, B, g, rb, lastz, flags, ps);
        project_fill_kernel<true><<<fg, 256, 0, st>>>(rb, lastz, flags, P);
        project_kernel<1><<<nb_small, 256, 0, st>>>(xyz, offsets, total, base, B, g, rb, lastz, flags, ps);
        project_kernel<2><<<nb_small, 256, 0, st>>>(xyz, offsets, total, base, B, g, rb, lastz, flags, ps);
        LAUNCH_CHECK();
    }
    project_finalize_kernel<false><<<fg, 256, 0, st>>>(rb, P, flags);
    LAUNCH_CHECK();
    return RPCC_OK;
}

// point_stride_bytes -> floats per point (0 = 12); only packed xyz (12) and the stored (x, y, z, intensity) rows (16) exist
static inline int point_floats(int point_stride_bytes) { return point_stride_bytes == 0 || point_stride_bytes == 12 ? 3 : point_stride_bytes == 16 ? 4 : -1; }

// flags of rpcc_batch_io / order argument of rpcc_project_ordered -> launch_project's order_mode
static inline int order_mode_of(int flags) {
    return (flags & RPCC_PROJECT_FORCE_ORDERED) ? ORD_MODE_FORCE : (flags & RPCC_PROJECT_ORDER_PROBE) ? ORD_MODE_PROBE : ORD_MODE_OFF;
}
extern "C" int rpcc_project_ordered(const float *points, int point_stride_bytes, const int64_t *offsets, int64_t total, int B,
                                    rpcc_geom g, float *ri, void *scratch, size_t scratch_bytes, int order_flags, int32_t *accepted,
                                    void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && g.H > 1 && g.W > 0 && total >= 0);
    ARG_TRY(ri != nullptr && scratch != nullptr && offsets != nullptr);
    ARG_TRY(total == 0 || points != nullptr);
    ARG_TRY(point_floats(point_stride_bytes) > 0);
    ARG_TRY(point_stride_bytes != 16 || (reinterpret_cast<uintptr_t>(points) & 15u) == 0);   // rows are read with 16-byte loads
    ARG_TRY(scratch_bytes >= ((size_t)B * ((size_t)g.H * g.W + 8)) * 4);
    ARG_TRY((order_flags & ~(RPCC_PROJECT_ORDER_PROBE | RPCC_PROJECT_FORCE_ORDERED)) == 0);
    return launch_project(points, offsets, total, 0, B, g, ri, scratch, scratch_bytes, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr,
                          point_floats(point_stride_bytes), order_mode_of(order_flags), nullptr, accepted);
}
extern "C" int rpcc_project_strided(const float *points, int point_stride_bytes, const int64_t *offsets, int64_t total, int B,
                                    rpcc_geom g, float *ri, void *scratch, size_t scratch_bytes, void *stream) {
    return rpcc_project_ordered(points, point_stride_bytes, offsets, total, B, g, ri, scratch, scratch_bytes, 0, nullptr, stream);
}
extern "C" int rpcc_project(const float *xyz, const int64_t *offsets, int64_t total, int B, rpcc_geom g, float *ri,
                            void *scratch, size_t scratch_bytes, void *stream) {
    return rpcc_project_strided(xyz, 12, offsets, total, B, g, ri, scratch, scratch_bytes, stream);
}

// ================================================================================================
// a4  ground plane: candidate selection + seeded RANSAC  (utils/segment_utils.py:74-82,101-108)
// ================================================================================================
// The reference calls Open3D's segment_plane (random, not vendored, unpinned) on a random subsample.
// This is the build's own deterministic definition (DESIGN.md "RANSAC"); its sequential form lives in
// the oracle (orc_ransac_plane / orc_ground_candidates) and the two agree bit for bit:
//   candidates  pixels with z = ri*tz < zthr in row-major order; more than max_pts -> systematic
//               subsample of exactly max_pts (candidate i kept iff floor((i+1)*max/n) > floor(i*max/n),
//               its slot is floor(i*max/n)); fewer than min_pts -> every pixel, zeros included
//   hypotheses  `iters` samples of `ransac_n` distinct points drawn with a counter-based hash
//   fit         centroid + centred second moments, largest-determinant closed form, fp64
//   score       inlier count (|n.p+d| < thr), ties -> lower hypothesis id
//   refit       the same closed form on the winner's inliers; fp64 sums in a fixed order (256 strided
//               partials, then a binary tree)
#define RS_THREADS 512   // 1024 is no faster alone and co-schedules worse with the other batches' kernels
#define RS_NT 256
#define RS_MAX_LIST 5120
#define RS_MAX_HYP 128

__device__ __forceinline__ uint32_t mix32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u + 0x7F4A7C15u;
    h ^= b + 0x85EBCA6Bu + (h << 6) + (h >> 2);
    h *= 0xC2B2AE35u;
    h ^= c + 0x27D4EB2Fu + (h << 6) + (h >> 2);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ bool plane_from_moments(const double c[3], double xx, double xy, double xz, double yy,
                                                   double yz, double zz, double out[4]) {
    const double det_x = yy * zz - yz * yz, det_y = xx * zz - xz * xz, det_z = xx * yy - xy * xy;
    double a, b, cc;
    if (det_x >= det_y && det_x >= det_z) { a = det_x; b = xz * yz - xy * zz; cc = xy * yz - xz * yy; }
    else if (det_y >= det_z) { a = xz * yz - xy * zz; b = det_y; cc = xy * xz - yz * xx; }
    else { a = xy * yz - xz * yy; b = xy * xz - yz * xx; cc = det_z; }
    const double nrm = sqrt((a * a + b * b) + cc * cc);
    if (!(nrm > 0.0)) return false;
    a /= nrm; b /= nrm; cc /= nrm;
    out[0] = a; out[1] = b; out[2] = cc;
    out[3] = -((a * c[0] + b * c[1]) + cc * c[2]);
    return true;
}

// Point source of one RANSAC problem: a compacted fp32 list in LDS, or every pixel of a range image.
struct RsPoints {
    const float *lds;   // [n,3] or nullptr
    const float *ri;    // frame's range image
    const float *tm;    // [P,3]
    int n;
    int raw;            // ri still holds projection bit patterns (RI_EMPTY = empty pixel)
    __device__ __forceinline__ void get(int i, double &x, double &y, double &z) const {
        if (lds) { x = (double)lds[3 * i]; y = (double)lds[3 * i + 1]; z = (double)lds[3 * i + 2]; }
        else { float r = ri[i]; if (raw && f2u(r) == RI_EMPTY) r = 0.0f; x = (double)(r * tm[3 * i]); y = (double)(r * tm[3 * i + 1]); z = (double)(r * tm[3 * i + 2]); }
    }
    __device__ __forceinline__ void getf(int i, float &x, float &y, float &z) const {
        if (lds) { x = lds[3 * i]; y = lds[3 * i + 1]; z = lds[3 * i + 2]; }
        else { float r = ri[i]; if (raw && f2u(r) == RI_EMPTY) r = 0.0f; x = r * tm[3 * i]; y = r * tm[3 * i + 1]; z = r * tm[3 * i + 2]; }
    }
};

#define RS_GLOBAL_PU 8   // points per lane / thread in flight when a RANSAC pass reads its points from the range image instead of a list in LDS
__device__ __forceinline__ bool pts_in_memory(const RsPoints &p) { return p.lds == nullptr; }
template <class PTS> __device__ __forceinline__ bool pts_in_memory(const PTS &) { return false; }   // (other sources choose their own RS_PU / RS_RU)
typedef float rs_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rfl_f32(float v) { return u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(v))); }
// inlier test of the specification: fp32, un-fused, plane narrowed to fp32
__device__ __forceinline__ bool plane_inlier(const float pl[4], float x, float y, float z, float thr) {
    return fabsf(((pl[0] * x + pl[1] * y) + pl[2] * z) + pl[3]) < thr;
}

// ordered fp64 reduction of NV values at once: thread t < 256 owns partial t of each; the results are
// valid in every thread after return.  sred: [NV][RS_NT].
template <int NV>
__device__ __forceinline__ void rs_treesum(double (&v)[NV], double *sred) {
    const int t = threadIdx.x;
    __syncthreads();
    if (t < RS_NT) {
#pragma unroll
        for (int q = 0; q < NV; q++) sred[q * RS_NT + t] = v[q];
    }
    __syncthreads();
    for (int stride = RS_NT / 2; stride >= 1; stride >>= 1) {
        if (t < stride) {
#pragma unroll
            for (int q = 0; q < NV; q++) sred[q * RS_NT + t] += sred[q * RS_NT + t + stride];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NV; q++) v[q] = sred[q * RS_NT];
}

// Step (1) of the RANSAC: thread h < iters draws and fits hypothesis h -- RN distinct points by the counter-based hash, centroid + centred
// moments in fp64, largest-determinant closed form -- and leaves the plane narrowed to fp32 with a validity word in hyp[5 h ..] and in
// fp64 in hypd[4 h ..].  (Every thread calls; no barrier inside.)
template <int RN, class PTS>
__device__ __forceinline__ void ransac_hypotheses(const PTS &pts, int iters, uint32_t seed, float *hyp, double *hypd) {
    const int tid = threadIdx.x, n = pts.n;
    if (tid < iters) {
        const int h = tid;
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
        double pl[4] = {0, 0, 0, 0};
        const bool ok = plane_from_moments(c, xx, xy, xz, yy, yz, zz, pl);
        hyp[5 * h] = (float)pl[0]; hyp[5 * h + 1] = (float)pl[1]; hyp[5 * h + 2] = (float)pl[2]; hyp[5 * h + 3] = (float)pl[3];
        hyp[5 * h + 4] = ok ? 1.0f : 0.0f;
        hypd[4 * h] = pl[0]; hypd[4 * h + 1] = pl[1]; hypd[4 * h + 2] = pl[2]; hypd[4 * h + 3] = pl[3];
    }
}

// The end of the RANSAC: plane = the winning hypothesis with wcnt inliers (wcnt < 0: none was valid) -> the refit on its inliers, fp64 moments with
// ordered sums: thread t < RS_NT accumulates the inliers among points t, t + RS_NT, ... in index order (whatever RU, the points in flight), then a
// binary tree.  Every thread of the workgroup calls; sred: [6 * RS_NT] doubles of LDS.  Returns the inlier count reported for the frame.
template <int RS_RU, class PTS>
__device__ __forceinline__ int ransac_refit(const PTS &pts, int wcnt, float thr_f, double plane[4], double *sred) {
    const int tid = threadIdx.x, n = pts.n;
    if (wcnt < 0) return 0;
    if (wcnt < 3) return wcnt;
    const float wf[4] = {(float)plane[0], (float)plane[1], (float)plane[2], (float)plane[3]};
    // refit on the winner's inliers (fp64 moments, ordered sums)
    double c[3] = {0, 0, 0};
    auto centroid = [&](auto ru_tag) {   // RU points per thread in flight; partial tid accumulates its points in index order whatever RU
        constexpr int RU = decltype(ru_tag)::value;
        for (int i0 = tid; i0 < n; i0 += RS_NT * RU) {
            float x[RU], y[RU], z[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) pts.getf(min(i0 + RS_NT * u, n - 1), x[u], y[u], z[u]);
#pragma unroll
            for (int u = 0; u < RU; u++)
                if (i0 + RS_NT * u < n && plane_inlier(wf, x[u], y[u], z[u], thr_f)) { c[0] += (double)x[u]; c[1] += (double)y[u]; c[2] += (double)z[u]; }
        }
    };
    if (tid < RS_NT) {
        if (pts_in_memory(pts)) centroid(std::integral_constant<int, (RS_RU > RS_GLOBAL_PU ? RS_RU : RS_GLOBAL_PU)>{});
        else centroid(std::integral_constant<int, RS_RU>{});
    }
    DBG_STAMP(4);
    rs_treesum<3>(c, sred);
    DBG_STAMP(5);
    c[0] /= (double)wcnt; c[1] /= (double)wcnt; c[2] /= (double)wcnt;
    double m[6] = {0, 0, 0, 0, 0, 0};
    auto moments = [&](auto ru_tag) {
        constexpr int RU = decltype(ru_tag)::value;
        for (int i0 = tid; i0 < n; i0 += RS_NT * RU) {
            float x[RU], y[RU], z[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) pts.getf(min(i0 + RS_NT * u, n - 1), x[u], y[u], z[u]);
#pragma unroll
            for (int u = 0; u < RU; u++)
                if (i0 + RS_NT * u < n && plane_inlier(wf, x[u], y[u], z[u], thr_f)) {
                    const double rx = (double)x[u] - c[0], ry = (double)y[u] - c[1], rz = (double)z[u] - c[2];
                    m[0] += rx * rx; m[1] += rx * ry; m[2] += rx * rz; m[3] += ry * ry; m[4] += ry * rz; m[5] += rz * rz;
                }
        }
    };
    if (tid < RS_NT) {
        if (pts_in_memory(pts)) moments(std::integral_constant<int, (RS_RU > RS_GLOBAL_PU ? RS_RU : RS_GLOBAL_PU)>{});
        else moments(std::integral_constant<int, RS_RU>{});
    }
    rs_treesum<6>(m, sred);
    double pl[4];
    if (plane_from_moments(c, m[0], m[1], m[2], m[3], m[4], m[5], pl)) { plane[0] = pl[0]; plane[1] = pl[1]; plane[2] = pl[2]; plane[3] = pl[3]; }
    return wcnt;
}

// Workgroup-wide RANSAC on `pts` (RN = sample size, NTH = threads of the workgroup (>= RS_NT), MAXH = most
// hypotheses); every thread calls it.  Returns the
// winner's inlier count.  sred [6*RS_NT] doubles, swin [16*4] doubles, sbest [16*2] ints.
// RS_PU: points per lane in flight in the scoring / refit loops (1 for points in LDS, more for points in global memory)
// BYPTS: the wavefronts share the POINTS instead of the hypotheses (every wavefront scores all MAXH <= 32 planes on its
// part): for a long list in global memory and few hypotheses, where a pass over the list costs more than the tests.
// RS_RU: points per thread in flight in the two ordered refit passes (only 256 threads walk those).
template <int RN, int NTH, int MAXH, int RS_PU, class PTS, bool BYPTS = false, int RS_RU = RS_PU>
__device__ int ransac_plane_wg(const PTS &pts, int iters, double thr, uint32_t seed, double plane[4], double *sred,
                               double *swin, int *sbest) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = pts.n;
    const float thr_f = (float)thr;
    plane[0] = 0; plane[1] = 0; plane[2] = 1; plane[3] = 0;
    if (n < RN || iters > MAXH) return 0;
    // (1) fits: one hypothesis per lane (iters <= RS_MAX_HYP), results narrowed to fp32 in LDS
    float *hyp = reinterpret_cast<float *>(sred);  // [iters][4] fp32 planes + validity, reused before the sums
    double *hypd = swin + 64;                      // [iters][4] fp64 planes (winner is read back from here)
    ransac_hypotheses<RN>(pts, iters, seed, hyp, hypd);
    if (BYPTS && tid < 32) sbest[tid] = 0;
    __syncthreads();
    DBG_STAMP(7);
    int wcnt = -1;
    if constexpr (BYPTS) {
        static_assert(!BYPTS || MAXH <= 32, "sbest holds one count per hypothesis");
        float pf[MAXH][4];
        int cnt[MAXH];
#pragma unroll
        for (int q = 0; q < MAXH; q++) {
            const int hh = q < iters ? q : 0;
            const bool ok = q < iters && hyp[5 * hh + 4] != 0.0f;
            pf[q][0] = rfl_f32(hyp[5 * hh]); pf[q][1] = rfl_f32(hyp[5 * hh + 1]); pf[q][2] = rfl_f32(hyp[5 * hh + 2]);
            pf[q][3] = rfl_f32(ok ? hyp[5 * hh + 3] : __builtin_inff());  // workgroup-uniform: scalar registers
            cnt[q] = 0;
        }
        for (int ib = 0; ib < n; ib += NTH * RS_PU) {   // (workgroup-uniform trip count)
            const int i0 = ib + tid;
            float x[RS_PU], y[RS_PU], z[RS_PU];
#pragma unroll
            for (int u = 0; u < RS_PU; u++) pts.getf(min(i0 + NTH * u, n - 1), x[u], y[u], z[u]);
#pragma unroll
            for (int u = 0; u < RS_PU; u++) {
                // past the end: the point (inf, y, z) -- inf or NaN on every plane, never an inlier.  The counts are
                // wave-uniform: a compare writes a lane mask, s_bcnt1 counts it.
                const float xu = i0 + NTH * u < n ? x[u] : __builtin_inff();
                const rs_v2f xx = {xu, xu}, yy = {y[u], y[u]}, zz = {z[u], z[u]};
#pragma unroll
                for (int q = 0; q + 1 < MAXH; q += 2) {
                    const rs_v2f a = {pf[q][0], pf[q + 1][0]}, b2 = {pf[q][1], pf[q + 1][1]}, c2 = {pf[q][2], pf[q + 1][2]},
                                 d2 = {pf[q][3], pf[q + 1][3]};
                    const rs_v2f dd = ((a * xx + b2 * yy) + c2 * zz) + d2;
                    cnt[q] += (int)__popcll(__ballot(fabsf(dd.x) < thr_f));
                    cnt[q + 1] += (int)__popcll(__ballot(fabsf(dd.y) < thr_f));
                }
                if (MAXH & 1) cnt[MAXH - 1] += (int)__popcll(__ballot(plane_inlier(pf[MAXH - 1], xu, y[u], z[u], thr_f)));
            }
        }
#pragma unroll
        for (int q = 0; q < MAXH; q++)
            if (lane == 0 && cnt[q]) atomicAdd(&sbest[q], cnt[q]);
        __syncthreads();
        int wh = 0;
        for (int q = 0; q < MAXH; q++)  // most inliers, lowest hypothesis among equals
            if (q < iters && hyp[5 * q + 4] != 0.0f && sbest[q] > wcnt) { wcnt = sbest[q]; wh = q; }
        if (wcnt >= 0) { plane[0] = hypd[4 * wh]; plane[1] = hypd[4 * wh + 1]; plane[2] = hypd[4 * wh + 2]; plane[3] = hypd[4 * wh + 3]; }
    } else {
        // (2) scoring: wavefront w counts the inliers of hypotheses w, w+16, ... -- all of them in one pass
        // over the points (a point is read from LDS once and tested against up to RS_HPW planes)
        constexpr int RS_HPW = (MAXH + NTH / 64 - 1) / (NTH / 64);
        float pf[RS_HPW][4];
        int cnt[RS_HPW];
#pragma unroll
        for (int q = 0; q < RS_HPW; q++) {
            const int h = wave + q * (NTH / 64);
            const bool ok = h < iters && hyp[5 * (h < iters ? h : 0) + 4] != 0.0f;
            const int hh = h < iters ? h : 0;
            pf[q][0] = hyp[5 * hh]; pf[q][1] = hyp[5 * hh + 1]; pf[q][2] = hyp[5 * hh + 2];
            pf[q][3] = ok ? hyp[5 * hh + 3] : __builtin_inff();  // invalid -> never an inlier
            cnt[q] = 0;
        }
        // (wave-uniform trip count: with a per-lane bound the loop is divergent and the scalar counters are copied to VGPRs at every exit test)
        // PU points per lane in flight (unconditional, clamped loads): RS_PU for a list in LDS, RS_GLOBAL_PU when the points are read from the
        // range image (the ground fit on the whole cloud -- a frame with fewer than 800 candidates: every wavefront walks all P pixels, and with
        // four loads in flight that walk was the launch: 242 us for 16 x 1800 images against 86-113 us for the frames that fit from their list)
        auto score = [&](auto pu_tag) {
        constexpr int PU = decltype(pu_tag)::value;
        for (int ib = 0; ib < n; ib += 64 * PU) {
            const int i0 = ib + lane;
            float x[PU], y[PU], z[PU];
#pragma unroll
            for (int u = 0; u < PU; u++) pts.getf(min(i0 + 64 * u, n - 1), x[u], y[u], z[u]);
#pragma unroll
            for (int u = 0; u < PU; u++) {
                // past the end: the point (inf, y, z) -- inf or NaN on every plane, never an inlier (no per-lane `in` mask)
                const float xq = i0 + 64 * u < n ? x[u] : __builtin_inff();
                // two hypotheses per packed-fp32 instruction (v_pk_mul_f32 / v_pk_add_f32: each half rounds like the scalar
                // operation, so this is plane_inlier() twice)
                const rs_v2f xx = {xq, xq}, yy = {y[u], y[u]}, zz = {z[u], z[u]};
#pragma unroll
                for (int q = 0; q + 1 < RS_HPW; q += 2) {
                    const rs_v2f a = {pf[q][0], pf[q + 1][0]}, b2 = {pf[q][1], pf[q + 1][1]}, c2 = {pf[q][2], pf[q + 1][2]},
                                 d2 = {pf[q][3], pf[q + 1][3]};
                    const rs_v2f dd = ((a * xx + b2 * yy) + c2 * zz) + d2;
                    cnt[q] += (int)__popcll(__ballot(fabsf(dd.x) < thr_f));       // wave-uniform: v_cmp writes a lane mask, s_bcnt1 counts it
                    cnt[q + 1] += (int)__popcll(__ballot(fabsf(dd.y) < thr_f));
                }
                if (RS_HPW & 1) cnt[RS_HPW - 1] += (int)__popcll(__ballot(plane_inlier(pf[RS_HPW - 1], xq, y[u], z[u], thr_f)));
            }
        }
        };
        // The fit on the whole cloud (points in memory: a frame with fewer than 800 ground candidates) shares the POINTS among the wavefronts instead:
        // every wavefront walking all P pixels for its 13 planes is eight walks with two wavefronts per SIMD to hide them behind (1.22 ms per launch of 64 x 2048
        // sweeps, 0.25 ms of 16 x 1800 ones).  A wavefront now loads its eighth of the points once -- RS_GLOBAL_PU per lane in flight -- and runs all the planes
        // over them, two per packed instruction, the planes read from LDS (broadcast); the counts meet in LDS (integers: any order).  Same tests, same counts.
        int *cnt_all = reinterpret_cast<int *>(sred) + 1024;   // [MAXH] behind the fp32 planes in the (not yet used) sum area
        const bool shared_pts = pts_in_memory(pts);             // (workgroup-uniform)
        if (shared_pts) {
            for (int q = tid; q < MAXH; q += NTH) cnt_all[q] = 0;
            __syncthreads();
            constexpr int PU = RS_GLOBAL_PU;
            for (int ib = wave * 64 * PU; ib < n; ib += NTH * PU) {   // (wave-uniform trip count)
                const int i0 = ib + lane;
                float x[PU], y[PU], z[PU];
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    pts.getf(min(i0 + 64 * u, n - 1), x[u], y[u], z[u]);
                    if (i0 + 64 * u >= n) x[u] = __builtin_inff();   // past the end: inf or NaN on every plane, never an inlier
                }
                for (int q = 0; q < iters; q += 2) {
                    const int q1 = min(q + 1, iters - 1);
                    const bool ok0 = hyp[5 * q + 4] != 0.0f, ok1 = q + 1 < iters && hyp[5 * q1 + 4] != 0.0f;
                    const rs_v2f a = {hyp[5 * q], hyp[5 * q1]}, b2 = {hyp[5 * q + 1], hyp[5 * q1 + 1]}, c2 = {hyp[5 * q + 2], hyp[5 * q1 + 2]},
                                 d2 = {ok0 ? hyp[5 * q + 3] : __builtin_inff(), ok1 ? hyp[5 * q1 + 3] : __builtin_inff()};
                    int c0 = 0, c1 = 0;
#pragma unroll
                    for (int u = 0; u < PU; u++) {
                        const rs_v2f xx = {x[u], x[u]}, yy = {y[u], y[u]}, zz = {z[u], z[u]};
                        const rs_v2f dd = ((a * xx + b2 * yy) + c2 * zz) + d2;   // plane_inlier() twice (each half rounds like the scalar operation)
                        c0 += (int)__popcll(__ballot(fabsf(dd.x) < thr_f));
                        c1 += (int)__popcll(__ballot(fabsf(dd.y) < thr_f));
                    }
                    if (lane == 0) { if (c0) atomicAdd(&cnt_all[q], c0); if (c1 && q + 1 < iters) atomicAdd(&cnt_all[q + 1], c1); }
                }
            }
            __syncthreads();
        } else {
            score(std::integral_constant<int, RS_PU>{});
        }
        int best_cnt = -1, best_h = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < RS_HPW; q++) {
            const int h = wave + q * (NTH / 64);
            const int c = shared_pts ? cnt_all[h < iters ? h : 0] : cnt[q];
            if (h < iters && hyp[5 * h + 4] != 0.0f && c > best_cnt) { best_cnt = c; best_h = h; }
        }
        double best[4] = {0, 0, 1, 0};
        if (best_cnt >= 0) { best[0] = hypd[4 * best_h]; best[1] = hypd[4 * best_h + 1]; best[2] = hypd[4 * best_h + 2]; best[3] = hypd[4 * best_h + 3]; }
        __syncthreads();
        DBG_STAMP(3);
        if (lane == 0) {
            sbest[2 * wave] = best_cnt; sbest[2 * wave + 1] = best_h;
            swin[4 * wave] = best[0]; swin[4 * wave + 1] = best[1]; swin[4 * wave + 2] = best[2]; swin[4 * wave + 3] = best[3];
        }
        __syncthreads();
        int wbest = -1, wh = 0x7fffffff;
        for (int w = 0; w < NTH / 64; w++) {
            const int c = sbest[2 * w], hh = sbest[2 * w + 1];
            if (c > wcnt || (c == wcnt && c >= 0 && hh < wh)) { wcnt = c; wh = hh; wbest = w; }
        }
        if (wcnt >= 0) { plane[0] = swin[4 * wbest]; plane[1] = swin[4 * wbest + 1]; plane[2] = swin[4 * wbest + 2]; plane[3] = swin[4 * wbest + 3]; }
    }
    return ransac_refit<RS_RU>(pts, wcnt, thr_f, plane, sred);
}

// Scratch of the deferred whole-cloud fits of a batch of nframes frames: flag i32 [nframes] (1: the frame's fit is deferred; written for EVERY frame by
// every ground-fit launch, so nothing needs clearing), then per frame cnt i32 [RS_WC_H] inliers per hypothesis | hyp f32 [RS_WC_H * 5] | hypd f64 [RS_WC_H * 4].
#define RS_WC_H 104   // = RS_GROUND_MAXH
#define RS_WC_SLOT_BYTES (RS_WC_H * 4 + RS_WC_H * 20 + RS_WC_H * 32)
static_assert(RS_WC_SLOT_BYTES % 8 == 0 && (RS_WC_H * 24) % 8 == 0, "the fp64 planes of a slot are 8-byte aligned");
struct WcSlot { int32_t *cnt; float *hyp; double *hypd; };
__host__ __device__ __forceinline__ int32_t *wc_flags_of(char *wc) { return reinterpret_cast<int32_t *>(wc); }
__host__ __device__ __forceinline__ WcSlot wc_slot_of(char *wc, int nframes, int b) {
    char *p = wc + (((size_t)nframes * 4 + 63) & ~(size_t)63) + (size_t)b * RS_WC_SLOT_BYTES;
    WcSlot w;
    w.cnt = reinterpret_cast<int32_t *>(p); w.hyp = reinterpret_cast<float *>(w.cnt + RS_WC_H);
    w.hypd = reinterpret_cast<double *>(p + RS_WC_H * 24);
    return w;
}
// where the scratch lies: behind the candidate bytes of the band kernel's hand-off (all of it inside the FPS tile table's area, which is written later)
__host__ __device__ __forceinline__ char *rs_wc_of(const int32_t *zcnt, int B, int P) {
    return reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(rs_zmask_of(zcnt, B)) + (size_t)B * (size_t)(P >> 2) + 63) & ~(uintptr_t)63);
}

#define RS_CU 16  // pixels per lane in flight during the compaction pass
// hypotheses the ground fit's scoring is compiled for: 100 iterations over 8 wavefronts = 13 slots per wavefront (with RS_MAX_HYP = 128
// every wavefront evaluated 16, three of them always invalid)
#define RS_GROUND_MAXH 104
#define RS_GROUND_PU 4   // candidates per lane in flight in the scoring loop (LDS reads; 1 / 2 / 4: 108.5 / 106.9 / 105.3 us)
#define RS_VGPR_ATTR
__device__ __forceinline__ void ground_ransac_body(const float *__restrict__ ri_all,
                                                   const float *__restrict__ tm, int P, float zthr,
                                                   int max_pts, int min_pts, int ransac_n, int iters,
                                                   double thr, uint32_t seed0, int raw,
                                                   double *__restrict__ ground,
                                                   int32_t *__restrict__ ninl,
                                                   const int32_t *__restrict__ zcnt,
                                                   const int64_t *__restrict__ frame_ids, const int b, const int nframes,   // b: the workgroup's frame of nframes
                                                   char *__restrict__ wc = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    double *sred = reinterpret_cast<double *>(rs_smem);          // [6*256]
    double *swin = sred + 6 * RS_NT;                             // [64] + [RS_MAX_HYP*4] fp64 hypotheses
    int *sbest = reinterpret_cast<int *>(swin + 64 + RS_MAX_HYP * 4);  // [32]
    int *swave = sbest + 32;                                     // [16]
    float *list = reinterpret_cast<float *>(swave + 16);         // [max_pts*3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *ri = ri_all + (int64_t)b * P;
    DBG_STAMP(0);
    // each wave owns a contiguous run of pixels (rounded up to whole 64-pixel steps)
    constexpr int CPW = RS_CHUNKS / (RS_THREADS / 64);  // chunks (of the band kernel's hand-off) per wavefront
    const int per_wave = CPW * ((((P + RS_CHUNKS - 1) / RS_CHUNKS) + 63) & ~63);  // CPW * rs_chunk_px(P)
    const int w0 = wave * per_wave, w1 = min(P, w0 + per_wave);
    const bool have_cnt = zcnt != nullptr && zcnt[b * (RS_CHUNKS + 1) + RS_CHUNKS] != 0;  // counted by project_band_kernel
    auto zval = [&](int p) -> float {
        float r = ri[p];
        if (raw && f2u(r) == RI_EMPTY) r = 0.0f;  // projection bits not finalised yet
        return r * tm[3 * p + 2];
    };
    int cnt = 0;
    if (have_cnt)
        for (int q = 0; q < CPW; q++) cnt += zcnt[b * (RS_CHUNKS + 1) + wave * CPW + q];
    for (int p0 = w0; p0 < w1 && !have_cnt; p0 += 64 * 8) {  // 8 independent loads in flight per lane
        float zv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) zv[u] = zval(min(p0 + u * 64 + lane, P - 1));  // unconditional loads (clamped)
#pragma unroll
        for (int u = 0; u < 8; u++) cnt += __popcll(__ballot(p0 + u * 64 + lane < w1 && zv[u] < zthr));
    }
    if (lane == 0) swave[wave] = cnt;
    __syncthreads();
    DBG_STAMP(1);
    int nc = 0, base = 0;
    for (int w = 0; w < RS_THREADS / 64; w++) { if (w < wave) base += swave[w]; nc += swave[w]; }
    RsPoints pts;
    pts.ri = ri; pts.tm = tm; pts.lds = nullptr; pts.n = P; pts.raw = raw;
    const bool small_prod = (unsigned long long)nc * (unsigned long long)max_pts < (1ull << 32);
    UDiv32 by_nc = {0u, 0u};
    if (nc > max_pts && small_prod) {   // (workgroup-uniform; one 64-bit division per frame, kept in scalar registers)
        by_nc = udiv32_make((uint32_t)__builtin_amdgcn_readfirstlane(nc));
        by_nc.magic = (uint32_t)__builtin_amdgcn_readfirstlane((int)by_nc.magic);
        by_nc.shift = (uint32_t)__builtin_amdgcn_readfirstlane((int)by_nc.shift);
    }
    const bool have_bytes = have_cnt && zcnt[b * (RS_CHUNKS + 1) + RS_CHUNKS] == 3 && (P & 63) == 0 && !raw;
    if (nc >= min_pts && have_bytes) {
        // The band kernel left a byte per quad of pixels saying which of them are candidates: a lane takes a word of 64 pixels (16 bytes), ranks its
        // candidates with one scan over the wavefront and walks only the SET bits; candidate i goes to slot floor(i * max_pts / nc) when the next one
        // goes to another -- carried along as a = (i * max_pts) mod nc: kept iff a + max_pts >= nc -- which is the rule of the loop below, integer for
        // integer.  Kept pixels are noted in the list first; their coordinates follow in one round of loads (<= 10 per thread, all in flight).
        const uint8_t *zm = rs_zmask_of(zcnt, nframes) + (int64_t)b * (P >> 2);
        uint32_t *listp = reinterpret_cast<uint32_t *>(list);
        const bool sub = nc > max_pts;
        const uint32_t ncu = (uint32_t)nc, maxu = (uint32_t)max_pts;
        // (the words are dealt to the threads round-robin, not a contiguous eighth of the image per wavefront: the ground fills the lower rows, and the
        // wavefronts that owned those walked 64 set bits per word while the others had none; the ranks then need a scan over the workgroup per round)
        const int nwords = P >> 6;
        uint32_t run = 0u;
        __syncthreads();   // (every thread has read the chunk counts in swave: the rounds write it again)
        for (int wi0 = 0; wi0 < nwords; wi0 += RS_THREADS) {   // (workgroup-uniform trip count)
            const int wi = wi0 + tid;
            const int pw = wi << 6;                    // this thread's word: pixels pw .. pw + 63
            unsigned long long word = 0ull;
            if (wi < nwords) {
                const uint4 v = ld_at(reinterpret_cast<const uint4 *>(zm), (uint32_t)(pw >> 2));
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {   // 4 bytes of nibbles -> 16 bits
                    const uint32_t x = d[j] & 0x0F0F0F0Fu;
                    const uint32_t h = (x | (x >> 4)) & 0x00FF00FFu;
                    word |= (unsigned long long)((h | (h >> 8)) & 0xFFFFu) << (16 * j);
                }
            }
            const uint32_t c = (uint32_t)__popcll(word);
            const uint32_t incl = dpp_scan_incl_u32(c);
            if (lane == 63) swave[wave] = (int)incl;
            __syncthreads();
            uint32_t before = 0u, all = 0u;
            for (int w = 0; w < RS_THREADS / 64; w++) { const uint32_t t = (uint32_t)swave[w]; if (w < wave) before += t; all += t; }
            __syncthreads();                           // (swave is written again in the next round)
            uint32_t i = run + before + incl - c;      // rank of this thread's first candidate in the frame
            run += all;
            if (c) {
                if (!sub) {
                    while (word) { listp[3 * i] = (uint32_t)pw + (uint32_t)__builtin_ctzll(word); i++; word &= word - 1ull; }
                } else {
                    uint32_t slot, a;
                    if (small_prod) { slot = udiv32(i * maxu, by_nc); a = i * maxu - slot * ncu; }
                    else { const unsigned long long pr = (unsigned long long)i * maxu; slot = (uint32_t)(pr / ncu); a = (uint32_t)(pr - (unsigned long long)slot * ncu); }
                    while (word) {
                        const uint32_t t = a + maxu;
                        const bool keep = t >= ncu;
                        if (keep) listp[3 * slot] = (uint32_t)pw + (uint32_t)__builtin_ctzll(word);
                        a = keep ? t - ncu : t;
                        slot += keep ? 1u : 0u;
                        word &= word - 1ull;
                    }
                }
            }
        }
        __syncthreads();
        DBG_STAMP(8);
        const int nl = sub ? max_pts : nc;
        constexpr int LU = 10;   // (5000 slots over 512 threads)
        for (int s0 = tid; s0 < nl; s0 += RS_THREADS * LU) {
            float r[LU];
            f32x3 ray[LU];
#pragma unroll
            for (int u = 0; u < LU; u++) {
                // (past the end: the thread's OWN first slot again -- slot nl - 1 belongs to another thread, which may already have
                // replaced its pixel index by coordinates: float bits taken as an index would be an address anywhere in 4 GiB)
                const int sl = s0 + u * RS_THREADS;
                const uint32_t pp = listp[3 * (sl < nl ? sl : s0)];
                r[u] = ld_at(ri, pp * 4u);
                ray[u] = ld_at(reinterpret_cast<const f32x3 *>(tm), pp * 12u);
            }
#pragma unroll
            for (int u = 0; u < LU; u++) {
                const int sl = s0 + u * RS_THREADS;
                if (sl < nl) { list[3 * sl] = r[u] * ray[u].x; list[3 * sl + 1] = r[u] * ray[u].y; list[3 * sl + 2] = r[u] * ray[u].z; }
            }
        }
        pts.lds = list;
        pts.n = nl;
    } else if (nc >= min_pts) {
        int run = base;
        for (int p00 = w0; p00 < w1; p00 += 64 * RS_CU) {  // all loads of RS_CU steps are issued before any is used
            float xv[RS_CU], yv[RS_CU], zv[RS_CU];
#pragma unroll
            for (int u = 0; u < RS_CU; u++) {  // unconditional (clamped) loads: a guarded load would be waited for at once
                const uint32_t p = (uint32_t)min(p00 + u * 64 + lane, P - 1);  // byte offsets from wave-uniform bases
                float r = ld_at(ri, p * 4u);
                if (raw && f2u(r) == RI_EMPTY) r = 0.0f;
                const f32x3 ray = ld_at(reinterpret_cast<const f32x3 *>(tm), p * 12u);
                xv[u] = r * ray.x; yv[u] = r * ray.y; zv[u] = r * ray.z;
            }
#pragma unroll
            for (int u = 0; u < RS_CU; u++) {
                const int p = p00 + u * 64 + lane;
                const bool c = p < w1 && zv[u] < zthr;
                const unsigned long long m = __ballot(c);
                if (c) {
                    const long long i = run + __popcll(m & ((1ull << lane) - 1ull));
                    bool keep = true;
                    long long slot = i;
                    if (nc > max_pts) {
                        if (small_prod) {  // i * max_pts < nc * max_pts < 2^32: 32-bit quotients by the frame's invariant divisor
                            const uint32_t a = (uint32_t)i * (uint32_t)max_pts;
                            const uint32_t sl = udiv32(a, by_nc);
                            slot = sl;
                            keep = udiv32(a + (uint32_t)max_pts, by_nc) > sl;
                        } else {
                            slot = (i * max_pts) / nc;
                            keep = ((i + 1) * max_pts) / nc > slot;
                        }
                    }
                    if (keep) { list[3 * slot] = xv[u]; list[3 * slot + 1] = yv[u]; list[3 * slot + 2] = zv[u]; }
                }
                run += __popcll(m);
            }
        }
        pts.lds = list;
        pts.n = nc > max_pts ? max_pts : nc;
    }
    __syncthreads();
    DBG_STAMP(2);
    double plane[4];
    (void)ransac_n;  // the ground fit samples 10 points (utils/segment_utils.py:75)
    // the frame's seed follows its identity (datalist index), not its position in the batch
    const uint32_t fid = frame_ids ? (uint32_t)frame_ids[b] : (uint32_t)b;
    const bool defer = pts.lds == nullptr && wc != nullptr && P >= 10 && iters <= RS_GROUND_MAXH;   // (workgroup-uniform)
    if (wc != nullptr && tid == 0) wc_flags_of(wc)[b] = defer ? 1 : 0;
    if (defer) {
        // Fewer than min_pts candidates: the fit runs on EVERY pixel (segment_utils.py:105-106), 100 planes x P pixels of scoring -- 1.06 ms for the one CU
        // this workgroup sits on, and a launch lasts as long as its slowest frame.  Here only the hypotheses are drawn and fitted; the frame is marked,
        // ground_wc_score_kernel scores its planes with workgroups all over the chip, ground_wc_refit_kernel picks the winner and refits (same tests,
        // same ordered sums: the same plane to the bit).
        float *hyp = reinterpret_cast<float *>(sred);
        double *hypd = swin + 64;
        ransac_hypotheses<10>(pts, iters, seed0 + fid, hyp, hypd);
        __syncthreads();
        const WcSlot w = wc_slot_of(wc, nframes, b);
        for (int q = tid; q < RS_WC_H; q += RS_THREADS) w.cnt[q] = 0;
        for (int q = tid; q < 5 * RS_WC_H; q += RS_THREADS) w.hyp[q] = q < 5 * iters ? hyp[q] : 0.0f;   // (validity word 0 beyond the iterations)
        for (int q = tid; q < 4 * iters; q += RS_THREADS) w.hypd[q] = hypd[q];
        return;
    }
    const int inl = ransac_plane_wg<10, RS_THREADS, RS_GROUND_MAXH, RS_GROUND_PU, RsPoints, false, 1>(pts, iters, thr, seed0 + fid, plane, sred, swin, sbest);
    DBG_STAMP(6);
    if (tid == 0) {
        ground[4 * b] = plane[0]; ground[4 * b + 1] = plane[1]; ground[4 * b + 2] = plane[2]; ground[4 * b + 3] = plane[3];
        if (ninl) ninl[b] = inl;
    }
}
__global__ __launch_bounds__(RS_THREADS) RS_VGPR_ATTR void ground_ransac_kernel(const float *__restrict__ ri_all,
                                                                   const float *__restrict__ tm, int P, float zthr,
                                                                   int max_pts, int min_pts, int ransac_n, int iters,
                                                                   double thr, uint32_t seed0, int raw,
                                                                   double *__restrict__ ground,
                                                                   int32_t *__restrict__ ninl,
                                                                   const int32_t *__restrict__ zcnt,
                                                                   const int64_t *__restrict__ frame_ids, char *__restrict__ wc) {
    ground_ransac_body(ri_all, tm, P, zthr, max_pts, min_pts, ransac_n, iters, thr, seed0, raw, ground, ninl, zcnt, frame_ids, blockIdx.x, gridDim.x, wc);
}
// the frames of several geometry groups in one launch (rpcc_compress_batch_mixed; fps_kernels.h: fps_regtab_planar_multi_kernel)
struct RansacGroupArgs {
    const float *ri_all, *tm;
    int P;
    uint32_t seed0;
    double *ground;
    const int32_t *zcnt;
    const int64_t *frame_ids;
    char *wc;        // the group's scratch for deferred whole-cloud fits (flags, slots) or nullptr: such frames are fitted by their own workgroup
    int iters;       // hypotheses per fit
};
struct RansacMulti {
    int n, first[RPCC_MAX_GROUPS + 1];
    RansacGroupArgs a[RPCC_MAX_GROUPS];
};
__global__ __launch_bounds__(RS_THREADS) RS_VGPR_ATTR void ground_ransac_multi_kernel(const RansacMulti m, float zthr, int max_pts, int min_pts,
                                                                                      int ransac_n, int iters, double thr) {
    const int gi = multi_group_of(m.first, m.n, blockIdx.x);
    const RansacGroupArgs &a = m.a[gi];
    ground_ransac_body(a.ri_all, a.tm, a.P, zthr, max_pts, min_pts, ransac_n, iters, thr, a.seed0, 0, a.ground, nullptr, a.zcnt, a.frame_ids,
                       (int)blockIdx.x - m.first[gi], m.first[gi + 1] - m.first[gi], a.wc);
}

// ---- the deferred whole-cloud fits (frames with fewer than min_pts candidates; ground_ransac_body) -------------------------------------------
// Scoring: every listed frame's P pixels against its <= 104 planes, spread over the chip -- a workgroup takes 2048 pixels of one frame, a wavefront
// loads 8 per lane and runs all planes over them, two per packed instruction (plane_inlier() twice: each half rounds like the scalar operation),
// counts per plane meet in LDS and then in the slot (integers: any order).  The same tests as ransac_plane_wg's walk over a cloud in memory.
#define WC_SCORE_WGS 1024
#define WC_SCORE_THREADS 256
#define WC_PU 8
#define WC_LIST 1024      // frames whose marks a workgroup lists at a time
__global__ __launch_bounds__(WC_SCORE_THREADS) void ground_wc_score_kernel(const RansacMulti m, double thr) {
    __shared__ float s_hyp[RS_WC_H * 5];
    __shared__ int s_cnt[RS_WC_H];
    __shared__ uint16_t s_list[WC_LIST];
    __shared__ int s_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float thr_f = (float)thr;
    for (int gi = 0; gi < m.n; gi++) {
        const RansacGroupArgs &a = m.a[gi];
        if (a.wc == nullptr) continue;
        const int nframes = m.first[gi + 1] - m.first[gi];
        for (int fb = 0; fb < nframes; fb += WC_LIST) {   // WC_LIST frames at a time
        // the marked frames in ascending order (every workgroup finds the same list: the first wavefront, 64 flags per step)
        __syncthreads();
        if (wave == 0) {
            int nl = 0;
            for (int f0 = fb; f0 < min(nframes, fb + WC_LIST); f0 += 64) {
                const bool on = f0 + lane < nframes && wc_flags_of(a.wc)[f0 + lane] != 0;
                const unsigned long long mk = __ballot(on);
                if (on) s_list[nl + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)(f0 + lane - fb);
                nl += (int)__popcll(mk);
            }
            if (lane == 0) s_n = nl;
        }
        __syncthreads();
        const int nl = s_n;
        if (nl == 0) continue;   // (the common case: every frame of the batch had its ground candidates)
        const int P = a.P, tiles = (P + WC_SCORE_THREADS * WC_PU - 1) / (WC_SCORE_THREADS * WC_PU), iters = a.iters;
        for (int item = blockIdx.x; item < nl * tiles; item += gridDim.x) {
            const int li = item / tiles, t = item - li * tiles, b = fb + (int)s_list[li];
            const WcSlot w = wc_slot_of(a.wc, nframes, b);
            __syncthreads();
            for (int q = tid; q < 5 * iters; q += WC_SCORE_THREADS) s_hyp[q] = w.hyp[q];
            if (tid < RS_WC_H) s_cnt[tid] = 0;
            __syncthreads();
            const float *ri = a.ri_all + (int64_t)b * P;
            const int i0 = t * (WC_SCORE_THREADS * WC_PU) + wave * 64 * WC_PU + lane;
            float x[WC_PU], y[WC_PU], z[WC_PU];
#pragma unroll
            for (int u = 0; u < WC_PU; u++) {
                const int i = min(i0 + 64 * u, P - 1);
                const float r = ri[i];
                x[u] = r * a.tm[3 * i]; y[u] = r * a.tm[3 * i + 1]; z[u] = r * a.tm[3 * i + 2];   // RsPoints::getf
                if (i0 + 64 * u >= P) x[u] = __builtin_inff();   // past the end: inf or NaN on every plane, never an inlier
            }
            for (int q = 0; q < iters; q += 2) {
                const int q1 = min(q + 1, iters - 1);
                const bool ok0 = s_hyp[5 * q + 4] != 0.0f, ok1 = q + 1 < iters && s_hyp[5 * q1 + 4] != 0.0f;
                const rs_v2f pa = {s_hyp[5 * q], s_hyp[5 * q1]}, pb = {s_hyp[5 * q + 1], s_hyp[5 * q1 + 1]}, pc = {s_hyp[5 * q + 2], s_hyp[5 * q1 + 2]},
                             pd = {ok0 ? s_hyp[5 * q + 3] : __builtin_inff(), ok1 ? s_hyp[5 * q1 + 3] : __builtin_inff()};
                int c0 = 0, c1 = 0;
#pragma unroll
                for (int u = 0; u < WC_PU; u++) {
                    const rs_v2f xx = {x[u], x[u]}, yy = {y[u], y[u]}, zz = {z[u], z[u]};
                    const rs_v2f dd = ((pa * xx + pb * yy) + pc * zz) + pd;
                    c0 += (int)__popcll(__ballot(fabsf(dd.x) < thr_f));
                    c1 += (int)__popcll(__ballot(fabsf(dd.y) < thr_f));
                }
                if (lane == 0) { if (c0) atomicAdd(&s_cnt[q], c0); if (c1 && q + 1 < iters) atomicAdd(&s_cnt[q + 1], c1); }
            }
            __syncthreads();
            if (tid < iters && s_cnt[tid]) atomicAdd(&w.cnt[tid], s_cnt[tid]);
        }
        }
    }
}
// Winner and refit: one workgroup per frame, at work only for a marked one; most inliers among the valid hypotheses, the lower one among
// equals, then ransac_refit on every pixel -- the ordered sums of the specification (8 points per thread in flight: the passes are VALU-bound on the four
// wavefronts that own the 256 partials -- 16 or 32 in flight change nothing -- and a launch that finds no marked frame should not ask for 232 registers).
__global__ __launch_bounds__(RS_NT) void ground_wc_refit_kernel(const RansacMulti m, double thr) {
    __shared__ double sred[6 * RS_NT];
    const int gi = multi_group_of(m.first, m.n, blockIdx.x);
    const RansacGroupArgs &a = m.a[gi];
    const int b = (int)blockIdx.x - m.first[gi], nframes = m.first[gi + 1] - m.first[gi];
    if (a.wc == nullptr || wc_flags_of(a.wc)[b] == 0) return;
    const WcSlot w = wc_slot_of(a.wc, nframes, b);
    const int iters = a.iters;
    int wcnt = -1, wh = 0;
    for (int q = 0; q < iters; q++) {
        const int c = w.cnt[q];
        if (w.hyp[5 * q + 4] != 0.0f && c > wcnt) { wcnt = c; wh = q; }
    }
    double plane[4] = {0, 0, 1, 0};
    if (wcnt >= 0) { plane[0] = w.hypd[4 * wh]; plane[1] = w.hypd[4 * wh + 1]; plane[2] = w.hypd[4 * wh + 2]; plane[3] = w.hypd[4 * wh + 3]; }
    RsPoints pts;
    pts.ri = a.ri_all + (int64_t)b * a.P; pts.tm = a.tm; pts.lds = nullptr; pts.n = a.P; pts.raw = 0;
    ransac_refit<8>(pts, wcnt, (float)thr, plane, sred);
    if (threadIdx.x == 0) { a.ground[4 * b] = plane[0]; a.ground[4 * b + 1] = plane[1]; a.ground[4 * b + 2] = plane[2]; a.ground[4 * b + 3] = plane[3]; }
}

#define RS_GROUND_MAX_PTS 5000
#define RS_GROUND_MIN_PTS 800
static inline size_t ground_ransac_lds_bytes() {
    return (size_t)6 * RS_NT * 8 + (64 + RS_MAX_HYP * 4) * 8 + 32 * 4 + 16 * 4 + (size_t)RS_GROUND_MAX_PTS * 3 * 4;
}
// the two launches behind the per-frame fits: nothing to do (a read of the list's length per workgroup) unless a frame of the batch lacked ground candidates
static int launch_ground_wc(const RansacMulti &m, hipStream_t st) {
    bool any = false;
    for (int i = 0; i < m.n; i++) any = any || m.a[i].wc != nullptr;
    if (!any) return RPCC_OK;
    ground_wc_score_kernel<<<WC_SCORE_WGS, WC_SCORE_THREADS, 0, st>>>(m, 0.1);
    ground_wc_refit_kernel<<<m.first[m.n], RS_NT, 0, st>>>(m, 0.1);
    LAUNCH_CHECK();
    return RPCC_OK;
}
static int launch_ground_ransac(const float *ri, const float *tm, int B, int P, uint32_t seed0, bool raw, double *ground,
                                int32_t *ninl, hipStream_t st, const int32_t *zcnt = nullptr,
                                const int64_t *frame_ids = nullptr) {
    const int max_pts = RS_GROUND_MAX_PTS, min_pts = RS_GROUND_MIN_PTS;
    const size_t sh = ground_ransac_lds_bytes();
    HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&ground_ransac_kernel), (int)sh));
    // (with the projection's hand-off comes room for the deferred whole-cloud fits; the stand-alone entry has none: such a frame is fitted by its workgroup)
    const bool defer = zcnt != nullptr && !raw && ninl == nullptr;
    char *wc = defer ? rs_wc_of(zcnt, B, P) : nullptr;
    ground_ransac_kernel<<<B, RS_THREADS, sh, st>>>(ri, tm, P, -1.5f, max_pts, min_pts, 10, 100, 0.1, seed0, raw ? 1 : 0,
                                                    ground, ninl, zcnt, frame_ids, wc);
    LAUNCH_CHECK();
    if (defer) {
        RansacMulti m;
        m.n = 1; m.first[0] = 0; m.first[1] = B;
        m.a[0] = {ri, tm, P, seed0, ground, zcnt, frame_ids, wc, 100};
        return launch_ground_wc(m, st);
    }
    return RPCC_OK;
}
static int launch_ground_ransac_multi(const RansacMulti &m, hipStream_t st) {
    const size_t sh = ground_ransac_lds_bytes();
    HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&ground_ransac_multi_kernel), (int)sh));
    ground_ransac_multi_kernel<<<m.first[m.n], RS_THREADS, sh, st>>>(m, -1.5f, RS_GROUND_MAX_PTS, RS_GROUND_MIN_PTS, 10, 100, 0.1);
    LAUNCH_CHECK();
    return launch_ground_wc(m, st);
}

extern "C" int rpcc_ground_ransac(const float *ri, const float *tm, int B, int P, uint32_t seed, const int64_t *frame_ids,
                                  double *ground, int32_t *inliers, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && ri && tm && ground);
    return launch_ground_ransac(ri, tm, B, P, seed, false, ground, inliers, (hipStream_t)stream, nullptr, frame_ids);
}

// ================================================================================================
// a3 + a5  back-projection, vertical ground residual, candidate mask, FPS state init
//          (dataset/transformer.py:94-101, utils/segment_utils.py:44-47,119-120)
// ================================================================================================
#include "fps_kernels.h"

__global__ void info_init_kernel(int32_t *__restrict__ info, int B, int P) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        info[RPCC_INFO * b + 0] = 0;
        info[RPCC_INFO * b + 1] = P;
        info[RPCC_INFO * b + 2] = 0;
        info[RPCC_INFO * b + 3] = 0;
        info[RPCC_INFO * b + 4] = P;
        info[RPCC_INFO * b + 5] = 0; info[RPCC_INFO * b + 6] = 0; info[RPCC_INFO * b + 7] = 0;
    }
}

// RAW: ri still holds the projection's bit patterns (RI_EMPTY = untouched) and is finalised here.
// A 256-thread workgroup owns GM_PIX consecutive pixels of one frame; counts are reduced per wave
// (ballot), then per workgroup (LDS), then one atomic per counter per workgroup.
// (The variant that also runs the first FPS pass is ground_mask_tab_kernel in fps_kernels.h.)
#define GM_PIX 4096
template <bool RAW>
__global__ __launch_bounds__(256) void ground_mask_kernel(float *__restrict__ ri, const float *__restrict__ tm,
                                                          const double *__restrict__ ground, double thr, int P,
                                                          float *__restrict__ temp, int32_t *__restrict__ info) {
    __shared__ int s_cnt[4], s_nz[4], s_first[4], s_forg[4];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double a = ground[4 * b], bb = ground[4 * b + 1], c = ground[4 * b + 2], d = ground[4 * b + 3];
    // np.linalg.norm(plane_param[:, :3]) on a (1,1,4) array: all four components (segment_utils.py:47)
    const double div = sqrt(((a * a + bb * bb) + c * c) + d * d);
    int cnt = 0, nzc = 0, first = P, forg = P;
    for (int it = 0; it < GM_PIX / 256; it++) {
        const int p = blockIdx.x * GM_PIX + it * 256 + threadIdx.x;
        if ((int)(blockIdx.x * GM_PIX) + it * 256 >= P) break;  // whole workgroup past the image (uniform)
        const bool valid = p < P;
        const int pc = valid ? p : P - 1;
        float r = ri[(int64_t)b * P + pc];
        if (RAW && f2u(r) == RI_EMPTY) r = 0.0f;
        const float x = r * tm[3 * pc], y = r * tm[3 * pc + 1], z = r * tm[3 * pc + 2];
        const double s = ((double)x * a + (double)y * bb) + (double)z * c;
        const bool cand = valid && fabs(s + d) / div > thr;
        const bool nz = valid && r != 0.0f;
        if (valid) {
            if (RAW) ri[(int64_t)b * P + p] = r;
            temp[(int64_t)b * P + p] = cand ? 1e10f : -1.0f;
        }
        const unsigned long long mc = __ballot(cand), mz = __ballot(nz), mo = __ballot(cand && !nz);
        cnt += __popcll(mc);
        nzc += __popcll(mz);
        if (mc && first == P) first = (p - lane) + (int)__ffsll((long long)mc) - 1;
        if (mo && forg == P) forg = (p - lane) + (int)__ffsll((long long)mo) - 1;  // first empty pixel that is a candidate
    }
    if (lane == 0) { s_cnt[wave] = cnt; s_nz[wave] = nzc; s_first[wave] = first; s_forg[wave] = forg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tc = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const int tz = s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3];
        const int tf = min(min(s_first[0], s_first[1]), min(s_first[2], s_first[3]));
        const int to = min(min(s_forg[0], s_forg[1]), min(s_forg[2], s_forg[3]));
        if (tc) { atomicAdd(&info[RPCC_INFO * b + 0], tc); atomicMin(&info[RPCC_INFO * b + 1], tf); }
        if (tz) atomicAdd(&info[RPCC_INFO * b + 2], tz);
        if (to < P) atomicMin(&info[RPCC_INFO * b + 4], to);
    }
}

// tiletab: dev float4 [B][3][T] (T = tiles of fps_tiling_range(H,W)) or NULL
static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }
static int launch_ground_mask(float *ri, const float *tm, const double *ground, double thr, int B, int H, int W,
                              float *temp, int32_t *info, float *tiletab, hipStream_t st, bool raw,
                              bool info_ready = false) {
    const int P = H * W;
    if (!info_ready) info_init_kernel<<<(B + 255) / 256, 256, 0, st>>>(info, B, P);
    if (tiletab) {
        const FpsTiling g = fps_tiling_range(H, W);
        const dim3 grid((g.T + MASK_WAVES * TAB_TPW - 1) / (MASK_WAVES * TAB_TPW), B);
        const bool vec = (W % 4 == 0) && aligned16(ri) && aligned16(temp) && aligned16(tm);
#define GM_LAUNCH(RAW_, EDGE_) ground_mask_tab_kernel<RAW_, true, EDGE_><<<grid, 64 * MASK_WAVES, 0, st>>>(ri, tm, ground, thr, g, temp, info, tiletab)
        // (a width that is no multiple of four, or buffers that are not 16-byte aligned: the same quad layout at 4-byte alignment)
        if (raw) { if (vec) GM_LAUNCH(true, false); else GM_LAUNCH(true, true); }
        else     { if (vec) GM_LAUNCH(false, false); else GM_LAUNCH(false, true); }
#undef GM_LAUNCH
    } else {
        const dim3 grid((P + GM_PIX - 1) / GM_PIX, B);
        if (raw) ground_mask_kernel<true><<<grid, 256, 0, st>>>(ri, tm, ground, thr, P, temp, info);
        else     ground_mask_kernel<false><<<grid, 256, 0, st>>>(ri, tm, ground, thr, P, temp, info);
    }
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" size_t rpcc_fps_table_bytes(int B, int H, int W) {
    return (size_t)B * FPS_TAB_ROWS * fps_tiling_range(H, W).T * 4;
}

extern "C" int rpcc_ground_mask(const float *ri, const float *tm, const double *ground, double threshold, int B, int H,
                                int W, float *temp, int32_t *info, void *fps_table, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && ri && tm && ground && temp && info);
    ARG_TRY(fps_table == nullptr || fps_tiling_range(H, W).T <= FPS_TILED_MAX_TILES);
    return launch_ground_mask(const_cast<float *>(ri), tm, ground, threshold, B, H, W, temp, info,
                              reinterpret_cast<float *>(fps_table), (hipStream_t)stream, false);
}

// ================================================================================================
// a6  farthest point sampling  (ops/fps/src/sampling_gpu.cu:24-140, ops/fps/fps_utils.py:10-36)
//     v1: one 1024-thread workgroup per frame, brute-force pass per centre.
// ================================================================================================
__device__ __forceinline__ uint32_t block_argmax(unsigned long long key, unsigned long long *sm) {
    key = wave_max_u64(key);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // previous readers of sm are done
    if ((threadIdx.x & 63) == 0) sm[wave] = key;
    __syncthreads();
    unsigned long long k = sm[threadIdx.x & 15];  // FPS_THREADS / 64 = 16 partials
    k = wave_max_u64(k);
    return fps_key_index(k);
}

// probed: only the lists fps_list_probe_kernel has marked (idxs[b][0] < 0) -- the others ran in the tile-pruned kernel before this launch.
// VEC (n % 4 == 0, 16-byte aligned arrays): a thread owns four consecutive points per 4096-point stride -- three 16-byte loads of coordinates, one of
// temp, two strides in flight; element order inside a thread is ascending, so the strict '>' keeps the lowest index as the scalar form does.
template <bool VEC>
__global__ __launch_bounds__(FPS_THREADS) void fps_xyz_kernel(int n, int m, const float *__restrict__ dataset,
                                                              float *__restrict__ temp, int32_t *__restrict__ idxs, int probed) {
    __shared__ unsigned long long sm[16];
    if (m <= 0) return;
    const int b = blockIdx.x;
    dataset += (int64_t)b * n * 3;
    temp += (int64_t)b * n;
    idxs += (int64_t)b * m;
    if (probed && idxs[0] >= 0) return;   // (workgroup-uniform)
    const int tid = threadIdx.x;
    int old = 0;
    if (tid == 0) idxs[0] = 0;
    for (int j = 1; j < m; j++) {
        const float x1 = dataset[old * 3 + 0], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
        float best = -1.0f;
        int besti = 0;
        if (VEC) {
            const float4 *d4 = reinterpret_cast<const float4 *>(dataset);
            float4 *t4 = reinterpret_cast<float4 *>(temp);
            const int nq = n >> 2;
#pragma unroll 2
            for (int q = tid; q < nq; q += FPS_THREADS) {
                const float4 a = d4[3 * q], bq = d4[3 * q + 1], c = d4[3 * q + 2];
                float4 t = t4[q];
                const float px[4] = {a.x, a.w, bq.z, c.y}, py[4] = {a.y, bq.x, bq.w, c.z}, pz[4] = {a.z, bq.y, c.x, c.w};
                float tv[4] = {t.x, t.y, t.z, t.w};
                bool ch = false;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float dx = px[e] - x1, dy = py[e] - y1, dz = pz[e] - z1;
                    const float d = (dx * dx + dy * dy) + dz * dz;  // sampling_gpu.cu:64, un-fused
                    const float d2 = fminf(d, tv[e]);
                    ch = ch || d2 != tv[e];
                    tv[e] = d2;
                    if (d2 > best) { best = d2; besti = 4 * q + e; }
                }
                if (ch) t4[q] = make_float4(tv[0], tv[1], tv[2], tv[3]);
            }
        } else {
            for (int k = tid; k < n; k += FPS_THREADS) {
                const float dx = dataset[k * 3 + 0] - x1, dy = dataset[k * 3 + 1] - y1, dz = dataset[k * 3 + 2] - z1;
                const float d = (dx * dx + dy * dy) + dz * dz;  // sampling_gpu.cu:64, un-fused
                const float t = temp[k];
                const float d2 = fminf(d, t);
                if (d2 != t) temp[k] = d2;
                if (d2 > best) { best = d2; besti = k; }
            }
        }
        old = (int)block_argmax(fps_key(best, (uint32_t)besti), sm);
        if (tid == 0) idxs[j] = old;
    }
}

// Which kernel a point list takes.  The tile-pruned kernel wins when consecutive points are neighbours in space (a tile of 256 consecutive points is
// compact: the reference's row-major candidate list, a sweep in its stored order -- 6-12 % of the tiles touched per centre), the one-pass-per-centre kernel
// when they are not (a shuffled cloud: 80 % touched, and a touched tile costs 3.6 x a streamed one).  Measure of the order: mean over 16 sampled tiles of
// the squared box diagonal / squared diagonal of the sample's box (row-major lists 0.02-0.05, shuffled 0.44; break-even about 0.07: profiles/HISTORY.md).
// Marks idxs[b][0] = -1 (one pass per centre) or 0.  Any answer is correct -- both kernels return the same indices.
#define FPS_PROBE_TILES 16
__global__ __launch_bounds__(256) void fps_list_probe_kernel(const float *__restrict__ pts, int n, int m, int T, int32_t *__restrict__ idxs) {
    __shared__ float s_box[4][6], s_sum[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    pts += (int64_t)b * n * 3;
    const float inf = __builtin_inff();
    float g0 = inf, g1 = inf, g2 = inf, h0 = -inf, h1 = -inf, h2 = -inf, sum = 0.0f;
    float x[FPS_PROBE_TILES / 4][4], y[FPS_PROBE_TILES / 4][4], z[FPS_PROBE_TILES / 4][4];
#pragma unroll
    for (int k = 0; k < FPS_PROBE_TILES / 4; k++) {   // all loads first
        const int t = (int)(((int64_t)(4 * k + wave) * T) / FPS_PROBE_TILES);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int p = min(t * FPS_TILE + 64 * e + lane, n - 1);
            x[k][e] = pts[3 * (int64_t)p]; y[k][e] = pts[3 * (int64_t)p + 1]; z[k][e] = pts[3 * (int64_t)p + 2];
        }
    }
#pragma unroll
    for (int k = 0; k < FPS_PROBE_TILES / 4; k++) {
        float l0 = fminf(fminf(x[k][0], x[k][1]), fminf(x[k][2], x[k][3])), u0 = fmaxf(fmaxf(x[k][0], x[k][1]), fmaxf(x[k][2], x[k][3]));
        float l1 = fminf(fminf(y[k][0], y[k][1]), fminf(y[k][2], y[k][3])), u1 = fmaxf(fmaxf(y[k][0], y[k][1]), fmaxf(y[k][2], y[k][3]));
        float l2 = fminf(fminf(z[k][0], z[k][1]), fminf(z[k][2], z[k][3])), u2 = fmaxf(fmaxf(z[k][0], z[k][1]), fmaxf(z[k][2], z[k][3]));
        dpp_box6(l0, l1, l2, u0, u1, u2);
        sum += ((u0 - l0) * (u0 - l0) + (u1 - l1) * (u1 - l1)) + (u2 - l2) * (u2 - l2);
        g0 = fminf(g0, l0); g1 = fminf(g1, l1); g2 = fminf(g2, l2); h0 = fmaxf(h0, u0); h1 = fmaxf(h1, u1); h2 = fmaxf(h2, u2);
    }
    if (lane == 0) { s_box[wave][0] = g0; s_box[wave][1] = g1; s_box[wave][2] = g2; s_box[wave][3] = h0; s_box[wave][4] = h1; s_box[wave][5] = h2; s_sum[wave] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
        for (int w = 0; w < 4; w++) {
            tot += s_sum[w];
            g0 = fminf(g0, s_box[w][0]); g1 = fminf(g1, s_box[w][1]); g2 = fminf(g2, s_box[w][2]);
            h0 = fmaxf(h0, s_box[w][3]); h1 = fmaxf(h1, s_box[w][4]); h2 = fmaxf(h2, s_box[w][5]);
        }
        const float all = ((h0 - g0) * (h0 - g0) + (h1 - g1) * (h1 - g1)) + (h2 - g2) * (h2 - g2);
        const bool streamed = !(tot < 0.07f * (float)FPS_PROBE_TILES * all);   // (NaN / infinite coordinates: one pass per centre)
        idxs[(int64_t)b * m] = streamed ? -1 : 0;
    }
}

// Range-image form: point k is pixel k, xyz = ri*tm in registers, temp < 0 = not a candidate.
// Each thread owns 4 consecutive pixels per 4096-pixel stride (16-byte loads of ri/temp, 48 of tm).
__global__ __launch_bounds__(FPS_THREADS) void fps_range_kernel(const float *__restrict__ ri, const float *__restrict__ tm,
                                                                float *__restrict__ temp, const int32_t *__restrict__ info,
                                                                int P, int M, int32_t *__restrict__ cen_pix,
                                                                float *__restrict__ centers) {
    __shared__ unsigned long long sm[16];
    const int b = blockIdx.x;
    ri += (int64_t)b * P;
    temp += (int64_t)b * P;
    cen_pix += (int64_t)b * M;
    centers += (int64_t)b * M * 3;
    const int tid = threadIdx.x;
    int old = info[RPCC_INFO * b + 1];
    if (old >= P) old = 0;  // no candidate at all: the reference would fail; keep indices defined
    const int P4 = P & ~3;
    for (int j = 0; j < M; j++) {
        const float r0 = ri[old];
        const float x1 = r0 * tm[3 * old], y1 = r0 * tm[3 * old + 1], z1 = r0 * tm[3 * old + 2];
        if (tid == 0) {
            cen_pix[j] = old;
            centers[3 * j] = x1; centers[3 * j + 1] = y1; centers[3 * j + 2] = z1;
        }
        if (j == M - 1) break;
        float best = -1.0f;
        int besti = 0;
        for (int k = tid * 4; k < P4; k += FPS_THREADS * 4) {
            const float4 r = *reinterpret_cast<const float4 *>(ri + k);
            const float4 t = *reinterpret_cast<const float4 *>(temp + k);
            const float4 ta = *reinterpret_cast<const float4 *>(tm + 3 * k);
            const float4 tb = *reinterpret_cast<const float4 *>(tm + 3 * k + 4);
            const float4 tc = *reinterpret_cast<const float4 *>(tm + 3 * k + 8);
            const float rr[4] = {r.x, r.y, r.z, r.w};
            const float tt[4] = {t.x, t.y, t.z, t.w};
            const float ray[12] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w, tc.x, tc.y, tc.z, tc.w};
            float o[4];
            bool changed = false;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float dx = rr[q] * ray[3 * q] - x1, dy = rr[q] * ray[3 * q + 1] - y1, dz = rr[q] * ray[3 * q + 2] - z1;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const float d2 = fminf(d, tt[q]);
                o[q] = d2;
                changed |= d2 != tt[q];
                if (d2 > best) { best = d2; besti = k + q; }
            }
            if (changed) *reinterpret_cast<float4 *>(temp + k) = make_float4(o[0], o[1], o[2], o[3]);
        }
        for (int k = P4 + tid; k < P; k += FPS_THREADS) {  // tail when P is not a multiple of 4
            const float rr = ri[k];
            const float dx = rr * tm[3 * k] - x1, dy = rr * tm[3 * k + 1] - y1, dz = rr * tm[3 * k + 2] - z1;
            const float d = (dx * dx + dy * dy) + dz * dz;
            const float t = temp[k];
            const float d2 = fminf(d, t);
            if (d2 != t) temp[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
        old = (int)block_argmax(fps_key(best, (uint32_t)besti), sm);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The CUDA binary's two unobservable degrees of freedom as a selectable mode (measured sensitivity:
// profiles/r03_fps_mode_sensitivity.md).  One 1024-thread workgroup per frame, one full pass per centre:
//   fma      0 un-fused (the specification), 1 fma(dz,dz,fma(dx,dx,dy*dy)), 2 fma(dz,dz,fma(dy,dy,dx*dx)):
//            nvcc's --fmad=true contractions of sampling_gpu.cu:64
//   ctie     winner among EXACTLY equal values as the kernel's own reduction picks it (sampling_gpu.cu:16-21,55-69,74-134):
//            the candidate with the smallest bit-reversed (k mod bs), then the smallest k, where k is the position in the
//            COMPACTED candidate list the reference hands to the kernel (range image: rank among the pixels with temp >= 0;
//            point list: the index) and bs = opt_n_threads(number of candidates) (:9-13; pow2_down: see fps_cuda_block)
// A pass tracks per thread the best value, its lowest index and how many of the thread's points reach it; only when the
// maximum is reached by more than one point does the tie pass run (ranks by a workgroup-wide prefix count in index order).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fps_dist_mode(float dx, float dy, float dz, int fma) {
    if (fma == 1) return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
    if (fma == 2) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    return (dx * dx + dy * dy) + dz * dz;
}
// opt_n_threads (sampling_gpu.cu:9-13): max(min(1 << (int)(log(n) / log(2)), 1024), 1), evaluated in double on the HOST by the
// reference.  For n that is no power of two the truncation is floor(log2 n); for n = 2^p (p <= 10 is all that matters below
// the cap) the quotient of the two logarithms may fall just below p -- pow2_down has bit p set when the host's libm does that.
static uint32_t fps_pow2_down_mask() {
    uint32_t m = 0;
    for (int p = 1; p <= 10; p++)
        if ((int)(log((double)(1 << p)) / log(2.0)) < p) m |= 1u << p;
    return m;
}
__device__ __forceinline__ int fps_cuda_block(int n, uint32_t pow2_down) {
    if (n < 1) return 1;
    int p = 31 - __builtin_clz((unsigned)n);
    if (n == (1 << p) && p <= 10 && ((pow2_down >> p) & 1u)) p -= 1;
    const int b = 1 << p;
    return b > 1024 ? 1024 : (b < 1 ? 1 : b);
}
template <bool RANGE>
__global__ __launch_bounds__(FPS_THREADS) void fps_modes_kernel(const float *__restrict__ src, const float *__restrict__ rays,
                                                                float *__restrict__ temp, const int32_t *__restrict__ info, int N,
                                                                int M, int fma, int ctie, uint32_t pow2_down,
                                                                int32_t *__restrict__ out_idx, float *__restrict__ out_cen) {
    __shared__ unsigned long long sm[16];
    __shared__ int scnt[16];
    __shared__ int swin;
    if (M <= 0) return;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    src += (int64_t)b * N * (RANGE ? 1 : 3);
    temp += (int64_t)b * N;
    out_idx += (int64_t)b * M;
    if (out_cen) out_cen += (int64_t)b * M * 3;
    int old = 0, ncand = N;
    if (RANGE) { old = info[RPCC_INFO * b + 1]; if (old >= N) old = 0; ncand = info[RPCC_INFO * b + 0]; }
    const int bs = fps_cuda_block(ncand, pow2_down);
    int bits = 0;
    while ((1 << bits) < bs) bits++;
    for (int j = 0; j < M; j++) {
        float x1, y1, z1;
        fps_load_point<RANGE>(src, rays, old, x1, y1, z1);
        if (tid == 0) {
            out_idx[j] = old;
            if (out_cen) { out_cen[3 * j] = x1; out_cen[3 * j + 1] = y1; out_cen[3 * j + 2] = z1; }
        }
        if (j == M - 1) break;
        float best = -1.0f;
        int besti = 0, cnt_eq = 0;
        for (int k = tid; k < N; k += FPS_THREADS) {   // (a reference mode: scalar loads, one point per thread and step)
            const float t = temp[k];
            float x, y, z;
            fps_load_point<RANGE>(src, rays, k, x, y, z);
            const float d = fps_dist_mode(x - x1, y - y1, z - z1, fma);
            const float d2 = fminf(d, t);      // t < 0 (not a candidate) stays negative: never above best = -1
            if (d2 != t) temp[k] = d2;
            if (d2 > best) { best = d2; besti = k; cnt_eq = 1; }
            else if (d2 == best && t >= 0.0f) cnt_eq++;
        }
        // maximum, lowest index among the points that reach it, and how many reach it
        const unsigned long long key = fps_key(best, (uint32_t)besti);
        unsigned long long kmax = wave_max_u64(key);
        __syncthreads();   // previous readers of sm / scnt are done; every temp store of this pass is visible afterwards
        if (lane == 0) sm[wave] = kmax;
        __syncthreads();
        kmax = wave_max_u64(sm[lane & 15]);
        const uint32_t vmax_hi = (uint32_t)(kmax >> 32);
        int winner = (int)fps_key_index(kmax);
        if (ctie && vmax_hi != 0u) {
            const float vmax = u2f(vmax_hi - 1u);
            int tot = wave_sum_i32(best == vmax ? cnt_eq : 0);
            if (lane == 0) scnt[wave] = tot;
            __syncthreads();
            tot = wave_sum_i32(lane < 16 ? scnt[lane] : 0);
            if (tot > 1) {   // (workgroup-uniform) exact ties at the maximum: the CUDA tree's survivor
                unsigned long long bkey = ~0ull;
                int bp = 0, running = 0;
                for (int base = 0; base < N; base += FPS_THREADS) {
                    const int p = base + tid;
                    const float t = p < N ? temp[p] : -1.0f;
                    const bool cand = RANGE ? t >= 0.0f : p < N;
                    const unsigned long long m = __ballot(cand);
                    const int below = __popcll(m & ((1ull << lane) - 1ull));
                    __syncthreads();
                    if (lane == 0) scnt[wave] = __popcll(m);
                    __syncthreads();
                    int wbase = 0, total = 0;
                    for (int w = 0; w < 16; w++) { const int c = scnt[w]; if (w < wave) wbase += c; total += c; }
                    const int rank = running + wbase + below;
                    if (cand && t == vmax) {
                        const uint32_t r = (uint32_t)rank % (uint32_t)bs;
                        const uint32_t br = bits ? (__brev(r) >> (32 - bits)) : 0u;
                        const unsigned long long k2 = ((unsigned long long)br << 32) | (uint32_t)rank;
                        if (k2 < bkey) { bkey = k2; bp = p; }
                    }
                    running += total;
                }
                // minimum key of the workgroup and the pixel that holds it
                unsigned long long kk = bkey;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const unsigned long long t2 = __shfl_xor(kk, o, RPCC_WAVE); kk = t2 < kk ? t2 : kk; }
                __syncthreads();
                if (lane == 0) sm[wave] = kk;
                __syncthreads();
                unsigned long long kmin = sm[lane & 15];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { const unsigned long long t2 = __shfl_xor(kmin, o, RPCC_WAVE); kmin = t2 < kmin ? t2 : kmin; }
                if (bkey == kmin && bkey != ~0ull) swin = bp;   // ranks are distinct: exactly one thread
                __syncthreads();
                winner = swin;
            }
        }
        old = winner;
    }
}

template <bool RANGE>
static int launch_fps_tiled(const float *src, const float *rays, float *temp, const int32_t *info, int B, const FpsTiling &g,
                            int M, int kflags, int32_t *idx, float *cen, const float *tiletab, bool vec, hipStream_t st,
                            const float *rays_soa = nullptr, bool edge = false) {
    // edge (range images only): the width is no multiple of four -- the quad kernels run with 4-byte aligned 16-byte accesses
    const size_t sh = fps_tiled_lds_bytes(g.T);
    // register-table form: every tile owned by one lane (at most 64 tiles per wavefront)
#define FPS_RT_LAUNCH(VEC_, TT_) fps_regtab_kernel<RANGE, VEC_, TT_><<<B, TT_, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab)
    {
        const int tt = B <= 128 ? FPS_TT_SMALL : FPS_TT_BATCH;
        if constexpr (RANGE) if (g.T <= tt && (vec || edge) && rays_soa != nullptr) {   // planar copy of the ray table (the fused batch has one)
            if (edge) {
                if (B <= 128) fps_regtab_planar_kernel<FPS_TT_SMALL, true><<<B, FPS_TT_SMALL, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab, rays_soa);
                else          fps_regtab_planar_kernel<FPS_TT_BATCH, true><<<B, FPS_TT_BATCH, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab, rays_soa);
            } else {
                if (B <= 128) fps_regtab_planar_kernel<FPS_TT_SMALL><<<B, FPS_TT_SMALL, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab, rays_soa);
                else          fps_regtab_planar_kernel<FPS_TT_BATCH><<<B, FPS_TT_BATCH, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab, rays_soa);
            }
            LAUNCH_CHECK();
            return RPCC_OK;
        }
        if constexpr (RANGE) if (g.T > tt && g.T <= 2 * tt && (vec || edge) && rays_soa != nullptr) {   // two tiles per lane
            if (edge) {
                if (B <= 128) fps_regtab_planar2_kernel<FPS_TT_SMALL, true><<<B, FPS_TT_SMALL, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab, rays_soa);
                else          fps_regtab_planar2_kernel<FPS_TT_BATCH, true><<<B, FPS_TT_BATCH, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab, rays_soa);
            } else {
                if (B <= 128) fps_regtab_planar2_kernel<FPS_TT_SMALL><<<B, FPS_TT_SMALL, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab, rays_soa);
                else          fps_regtab_planar2_kernel<FPS_TT_BATCH><<<B, FPS_TT_BATCH, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab, rays_soa);
            }
            LAUNCH_CHECK();
            return RPCC_OK;
        }
        if constexpr (RANGE) if (g.T <= tt && edge) {
            if (B <= 128) fps_regtab_kernel<true, true, FPS_TT_SMALL, true><<<B, FPS_TT_SMALL, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab);
            else          fps_regtab_kernel<true, true, FPS_TT_BATCH, true><<<B, FPS_TT_BATCH, 0, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab);
            LAUNCH_CHECK();
            return RPCC_OK;
        }
        if (g.T <= tt) {
            if (B <= 128) { if (vec) FPS_RT_LAUNCH(true, FPS_TT_SMALL); else FPS_RT_LAUNCH(false, FPS_TT_SMALL); }
            else          { if (vec) FPS_RT_LAUNCH(true, FPS_TT_BATCH); else FPS_RT_LAUNCH(false, FPS_TT_BATCH); }
            LAUNCH_CHECK();
            return RPCC_OK;
        }
    }
#undef FPS_RT_LAUNCH
#define FPS_LAUNCH(VEC_, TT_)                                                                                        \
    do {                                                                                                             \
        HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&fps_tiled_kernel<RANGE, VEC_, TT_>), (int)sh));       \
        fps_tiled_kernel<RANGE, VEC_, TT_><<<B, TT_, sh, st>>>(src, rays, temp, info, g, M, kflags, idx, cen, tiletab); \
    } while (0)
    if (B <= 128) { if (vec) FPS_LAUNCH(true, FPS_TT_SMALL); else FPS_LAUNCH(false, FPS_TT_SMALL); }
    else          { if (vec) FPS_LAUNCH(true, FPS_TT_BATCH); else FPS_LAUNCH(false, FPS_TT_BATCH); }
#undef FPS_LAUNCH
    LAUNCH_CHECK();
    return RPCC_OK;
}

#define FPS_SOA 1   // the fused batch hands the planar copy of the ray table to the FPS kernel
#define RPCC_FPS_MODE_BITS (RPCC_FPS_FMA1 | RPCC_FPS_FMA2 | RPCC_FPS_TIE_CUDA)
static inline int fps_fma_of(int flags) { return (flags & RPCC_FPS_FMA1) ? 1 : (flags & RPCC_FPS_FMA2) ? 2 : 0; }

static int fps_xyz_impl(int B, int N, int M, const float *points, float *temp, int32_t *idx, bool brute, hipStream_t st, int flags = 0) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && N > 0 && M >= 0 && points && temp && idx);
    ARG_TRY(!((flags & RPCC_FPS_FMA1) && (flags & RPCC_FPS_FMA2)));
    if (M == 0) return RPCC_OK;
    if (flags & RPCC_FPS_MODE_BITS) {   // CUDA-binary modes: the reference kernel, one pass per centre
        fps_modes_kernel<false><<<B, FPS_THREADS, 0, st>>>(points, nullptr, temp, nullptr, N, M, fps_fma_of(flags),
                                                           (flags & RPCC_FPS_TIE_CUDA) ? 1 : 0, fps_pow2_down_mask(), idx, nullptr);
        LAUNCH_CHECK();
        return RPCC_OK;
    }
    const FpsTiling g = fps_tiling_list(N);
    const bool vec = (N % 4 == 0) && aligned16(points) && aligned16(temp);
    int probed = 0;
    if (!brute && g.T <= FPS_TILED_MAX_TILES && N < (1 << 30) / 3) {
        // every list takes the kernel its order suits: the probe marks idx[b][0], the pruned kernel skips the marked lists, the streamed one the others
        fps_list_probe_kernel<<<B, 256, 0, st>>>(points, N, M, g.T, idx);
        LAUNCH_CHECK();
        const int rc = launch_fps_tiled<false>(points, nullptr, temp, nullptr, B, g, M, FPS_FLAG_PROBED, idx, nullptr, nullptr, vec, st);
        if (rc != RPCC_OK) return rc;
        probed = 1;
    }
    if (vec && N < (1 << 29)) fps_xyz_kernel<true><<<B, FPS_THREADS, 0, st>>>(N, M, points, temp, idx, probed);
    else fps_xyz_kernel<false><<<B, FPS_THREADS, 0, st>>>(N, M, points, temp, idx, probed);
    LAUNCH_CHECK();
    return RPCC_OK;
}
extern "C" int rpcc_fps_xyz(int B, int N, int M, const float *points, float *temp, int32_t *idx, void *stream) {
    return fps_xyz_impl(B, N, M, points, temp, idx, false, (hipStream_t)stream);
}
extern "C" int rpcc_fps_xyz_bruteforce(int B, int N, int M, const float *points, float *temp, int32_t *idx, void *stream) {
    return fps_xyz_impl(B, N, M, points, temp, idx, true, (hipStream_t)stream);
}
extern "C" int rpcc_fps_xyz_probe(int B, int N, const float *points, int32_t *marks, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && N > 0 && points && marks);
    hipStream_t st = (hipStream_t)stream;
    fps_list_probe_kernel<<<B, 256, 0, st>>>(points, N, 1, fps_tiling_list(N).T, marks);
    LAUNCH_CHECK();
    return RPCC_OK;
}
extern "C" int rpcc_fps_xyz_mode(int B, int N, int M, const float *points, float *temp, int32_t *idx, int flags, void *stream) {
    return fps_xyz_impl(B, N, M, points, temp, idx, (flags & RPCC_FPS_BRUTEFORCE) != 0, (hipStream_t)stream, flags);
}

// flags: RPCC_FPS_BRUTEFORCE -> the one-pass-per-centre kernel; finalize_temp: temp is read by the caller afterwards
static int launch_fps_range(const float *ri, const float *tm, float *temp, const int32_t *info, int B, int H, int W, int M,
                            int32_t *cen_pix, float *centers, int flags, bool finalize_temp, const float *tiletab,
                            void *timer, hipStream_t st, const float *rays_soa = nullptr) {
    const int P = H * W;
    const FpsTiling g = fps_tiling_range(H, W);
    const bool brute = (flags & RPCC_FPS_BRUTEFORCE) != 0;
    if (flags & RPCC_FPS_MODE_BITS) {   // CUDA-binary modes (contraction / tree tie rule): the reference kernel, one pass per centre
        if ((flags & RPCC_FPS_FMA1) && (flags & RPCC_FPS_FMA2)) return set_err(RPCC_ERR_ARG, "fps: RPCC_FPS_FMA1 and RPCC_FPS_FMA2 exclude each other%s%s");
        if (tiletab != nullptr) return set_err(RPCC_ERR_ARG, "fps: the FPS table of rpcc_ground_mask holds un-fused first-pass distances; pass NULL with a mode flag%s%s");
        FpsTimer tmr(st, timer);
        fps_modes_kernel<true><<<B, FPS_THREADS, 0, st>>>(ri, tm, temp, info, P, M, fps_fma_of(flags), (flags & RPCC_FPS_TIE_CUDA) ? 1 : 0,
                                                          fps_pow2_down_mask(), cen_pix, centers);
        LAUNCH_CHECK();
        return RPCC_OK;
    }
    if (!brute && g.T <= FPS_TILED_MAX_TILES && P < (1 << 22)) {
        const bool vec = (W % 4 == 0) && aligned16(ri) && aligned16(temp) && aligned16(tm);
        FpsTimer tmr(st, timer);
        return launch_fps_tiled<true>(ri, tm, temp, info, B, g, M, finalize_temp ? FPS_FLAG_FINALIZE_TEMP : 0, cen_pix, centers,
                                      tiletab, vec, st, rays_soa, !vec);
    }
    if (tiletab != nullptr && !brute)
        return set_err(RPCC_ERR_ARG, "fps_range: an FPS table was produced but the tiled kernel cannot run (image too large)%s%s");
    if (P % 4 != 0 || !aligned16(ri) || !aligned16(temp) || !aligned16(tm))
        return set_err(RPCC_ERR_ARG, "fps_range brute-force path needs 16-byte aligned buffers and P %% 4 == 0%s%s");
    FpsTimer tmr(st, timer);
    fps_range_kernel<<<B, FPS_THREADS, 0, st>>>(ri, tm, temp, info, P, M, cen_pix, centers);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_fps_range(const float *ri, const float *tm, float *temp, const int32_t *info, int B, int H, int W,
                              int M, int32_t *cen_pix, float *centers, int flags, const void *fps_table, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && M > 0 && ri && tm && temp && info && cen_pix && centers);
    return launch_fps_range(ri, tm, temp, info, B, H, W, M, cen_pix, centers, flags, true,
                            reinterpret_cast<const float *>(fps_table), nullptr, (hipStream_t)stream);
}

// ================================================================================================
// a7  assignment + relabel  (utils/segment_utils.py:21-23,64-67,127-131,168-169)
// ================================================================================================
// distance = concat(ground fp64, radius fp32->fp64); seg = argmax(-abs(distance)) = first minimum of
// |distance|.  The fp32 radius is sqrtf((dx*dx+dy*dy)+dz*dz); sqrtf is monotone, so the running minimum
// is tracked on the squared distance and sqrtf is evaluated only when the squared distance strictly
// improves -- the selected index is identical to evaluating all M square roots (DESIGN.md "assign").
// v3: per-tile centre pruning.  A wavefront owns a 4-row x 16-column tile of the range image (compact in
// 3-D).  With [lo,hi] the bounding box of the tile's non-empty pixels, for every centre k
//     dmin_k = ((gx*gx)+(gy*gy))+(gz*gz)   g* = per-axis gap to the box      <= computed d2_k(p)
//     dmax_k = ((fx*fx)+(fy*fy))+(fz*fz)   f* = per-axis farthest distance   >= computed d2_k(p)
// for every pixel p of the tile (same fp32 operation order as the distance; rounding is monotone).
// A centre with dmin_k > 1.000002 * min_j dmax_j is farther than the best centre by more than the
// sqrtf tie window for every pixel, so it can neither win nor tie and is skipped.  The survivors are
// visited in ascending index order with the two-smallest tracking below -- identical labels.
__device__ __forceinline__ float next_up_pos(float v) { return u2f(f2u(v) + 1u); }  // v >= 0, finite

// v4: a wavefront owns a 4-row x 32-column tile, two horizontally adjacent pixels per lane (the box reductions and
// the screening of the M centres are paid once per 128 pixels), and the two rarely decisive parts are screened:
//   * sqrt-level ties: sqrtf(m2) can equal sqrtf(m1) only if m2 <= m1 * (1 + 2^-22 + ...) -- the window U is computed
//     only behind  m2 <= m1 * 1.0000005f;
//   * ground versus cluster: the fp64 ground term |r - (-d / den)| is first estimated in fp32 together with a bound of
//     its error (representation of the plane, the three roundings of den, rcp, the subtraction); the fp64 sequence
//     runs only for pixels whose radius lies inside that error band (NaN / inf / a ray parallel to the plane fall
//     through to it automatically because every comparison with NaN is false).
// v5 (round 4): what bounds the survivors is no longer the box bound min_j dmax_j but the largest REACH of the tile's pixels (assign_reach):
// per pixel the smaller of (a) the squared ground term -- a centre farther than |r - (-d / den)| loses against the ground label -- and (b)
// the pixel's nearest-centre distance, which the FPS leaves in `temp` for every candidate pixel.  1.9 instead of 6.4 centres survive per
// tile; 32-57 % of the tiles (ground far from every centre) have none and are labelled without a distance.  The box bound remains for
// tiles whose reach is unknown (no `temp`, a ray parallel to the plane).
struct AssignGround {
    double a, b, c, d;
    float af, bf, cf, df, S;  // S >= |a| + |b| + |c|
};
// fp32 estimate of the ground term |r - (-d / den)| with a bound of its distance to the fp64 value (valid when rel < 0.01)
struct GroundEstimate { float agf, err, rel; };
__device__ __forceinline__ GroundEstimate ground_estimate(float r, float tx, float ty, float tz, const AssignGround &G) {
    const float denf = __builtin_fmaf(tx, G.af, __builtin_fmaf(ty, G.bf, tz * G.cf));
    const float rinv = __builtin_amdgcn_rcpf(denf);
    const float qf = -G.df * rinv;
    GroundEstimate e;
    e.agf = fabsf(r - qf);
    e.rel = 3.0e-7f * G.S * fabsf(rinv);  // |den error| / |den|
    e.err = fabsf(qf) * (e.rel + 4.0e-7f) + fabsf(r) * 1.0e-7f + e.agf * 1.5e-7f;
    return e;
}
// Upper bound of the squared distance beyond which a centre cannot matter for this pixel: a centre farther than the ground term loses
// against the ground label (index 0 wins ties), and a centre farther than the pixel's nearest centre -- which the FPS left in `temp`
// for every candidate pixel: the minimum over the first M - 1 centres of the same un-fused fp32 distance -- is not the nearest.  inf
// when neither is known (a ray parallel to the plane, NaN).  The factor covers the roundings of the square and of sqrtf.
__device__ __forceinline__ float assign_reach(float r, float tx, float ty, float tz, float tp, const AssignGround &G) {
    const GroundEstimate e = ground_estimate(r, tx, ty, tz, G);
    const float au = e.agf + e.err;
    float u = (e.rel < 0.01f && au < 1.0e18f) ? au * au * 1.000001f : __builtin_inff();   // (a NaN fails both compares)
    if (tp >= 0.0f) u = tp < u ? tp : u;
    return u;
}
__device__ __forceinline__ int assign_label(float r, float tx, float ty, float tz, float x, float y, float z, float m1, float m2,
                                            int k1, const float4 *cen4, const AssignGround &G) {
    if (k1 < 0) return 0;
    // radius = sqrtf(min d2).  Every squared distance that rounds to the same radius ties with it, and numpy's
    // argmax keeps the lowest index: U = largest float whose sqrtf equals the radius.
    int kk = k1;
    if (m2 <= m1 * 1.0000005f) {  // another centre may tie after the square root (rare)
        const float s = sqrtf(m1);
        float U = m1;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float n = next_up_pos(U);
            if (U < 3.0e38f && sqrtf(n) == s) U = n;
        }
        if (m2 <= U) {  // first index with d2 <= U
            for (int k = 0; k < kk; k++) {
                const float4 cc = cen4[k];
                const float dx = x - cc.x, dy = y - cc.y, dz = z - cc.z;
                if ((dx * dx + dy * dy) + dz * dz <= U) { kk = k; break; }
            }
        }
    }
    const GroundEstimate ge = ground_estimate(r, tx, ty, tz, G);
    const float agf = ge.agf, rel = ge.rel;
    // the radius enters the screen as the hardware square root (1 ulp) with its error added to the band; the correctly
    // rounded sqrtf of the reference is evaluated only inside the band
    const float sa = __builtin_amdgcn_sqrtf(m1);
    const float err = ge.err + sa * 2.5e-7f;
    bool cluster;
    if (rel < 0.01f && sa < agf - err) {
        cluster = true;
    } else if (rel < 0.01f && sa > agf + err) {
        cluster = false;
    } else {  // inside the error band (or not finite): the reference's fp64 sequence
        const double den = ((double)tx * G.a + (double)ty * G.b) + (double)tz * G.c;
        const double ag = fabs((double)r - (-G.d / den));
        cluster = !(ag != ag) && (double)sqrtf(m1) < ag;  // ground (index 0) wins ties and NaN
    }
    return cluster ? kk + 2 : 0;
}

#define ASSIGN_PX 4     // pixels per lane: a tile is (2 * ASSIGN_PX) rows x 32 columns, lane l pixel e -> row (l >> 4) + 4 * (e >> 1), column 2 * (l & 15) + (e & 1)
#define ASSIGN_ROWS (2 * ASSIGN_PX)
#define ASSIGN_TILES_PER_WAVE (4 / ASSIGN_PX)
#define ASSIGN_WAVES 4   // wavefronts per workgroup (they share the centre table in LDS, nothing else)
#define ASSIGN_VGPR_ATTR
// L: label type (uint8_t: cluster_num <= 254; uint16_t: up to RPCC_MAX_CLUSTERS_MID), NR: screening rounds of 64 centres (4 / 16)
template <class L = uint8_t, int NR = 4>
__global__ __launch_bounds__(64 * ASSIGN_WAVES) ASSIGN_VGPR_ATTR void assign_kernel(const float *__restrict__ ri, const float *__restrict__ tm,
                                                     const double *__restrict__ ground,
                                                     const float *__restrict__ centers, int H, int W, int M,
                                                     L *__restrict__ seg, const float *__restrict__ temp) {   // temp: the FPS state after its last iteration, or NULL
    extern __shared__ __attribute__((aligned(16))) float4 cen4[];  // [M] (x,y,z,0)
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = H * W;
    static_assert(ASSIGN_TILES_PER_WAVE == 1, "one tile per wavefront: its loads are issued before the centre table is filled");
    const int tcols = (W + 31) >> 5, ntile = ((H + ASSIGN_ROWS - 1) / ASSIGN_ROWS) * tcols;
    const int t0 = (blockIdx.x * ASSIGN_WAVES + wave) * ASSIGN_TILES_PER_WAVE;
    const float *ri_b = ri + (int64_t)b * P;
    L *seg_b = seg + (int64_t)b * P;
    // the tile's loads go out first (unconditional, clamped), then the centres: one trip to memory before the barrier, not two
    bool valid[ASSIGN_PX];
    int p[ASSIGN_PX];
    float r[ASSIGN_PX], tx[ASSIGN_PX], ty[ASSIGN_PX], tz[ASSIGN_PX], tp[ASSIGN_PX];
    const float *temp_b = temp ? temp + (int64_t)b * P : nullptr;
    {
        const int t = min(t0, ntile - 1);
        const int row0 = (t / tcols) * ASSIGN_ROWS + (lane >> 4), col0 = (t % tcols) * 32 + 2 * (lane & 15);
#pragma unroll
        for (int e = 0; e < ASSIGN_PX; e++) {
            const int row = row0 + 4 * (e >> 1), col = col0 + (e & 1);
            valid[e] = row < H && col < W && t0 < ntile;
            p[e] = valid[e] ? row * W + col : 0;
            r[e] = ld_at(ri_b, (uint32_t)p[e] * 4u);  // byte offsets from the frame's bases (scalar-base addressing)
            const f32x3 ray = ld_at(reinterpret_cast<const f32x3 *>(tm), (uint32_t)p[e] * 12u);
            tx[e] = ray.x; ty[e] = ray.y; tz[e] = ray.z;
            tp[e] = temp_b ? ld_at(temp_b, (uint32_t)p[e] * 4u) : -1.0f;
        }
    }
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const float *c = centers + ((int64_t)b * M + i) * 3;
        cen4[i] = make_float4(c[0], c[1], c[2], 0.0f);
    }
    __syncthreads();
    AssignGround G;
    G.a = ground[4 * b]; G.b = ground[4 * b + 1]; G.c = ground[4 * b + 2]; G.d = ground[4 * b + 3];
    G.af = (float)G.a; G.bf = (float)G.b; G.cf = (float)G.c; G.df = (float)G.d;
    G.S = (float)((fabs(G.a) + fabs(G.b) + fabs(G.c)) * 1.001);
    const float inf = __builtin_inff();
    for (int t = t0; t < min(t0 + ASSIGN_TILES_PER_WAVE, ntile); t++) {
        bool live[ASSIGN_PX];
        float x[ASSIGN_PX], y[ASSIGN_PX], z[ASSIGN_PX];
        bool any_live = false;
#pragma unroll
        for (int e = 0; e < ASSIGN_PX; e++) {
            if (!valid[e]) r[e] = 0.0f;
            x[e] = r[e] * tx[e]; y[e] = r[e] * ty[e]; z[e] = r[e] * tz[e];
            live[e] = valid[e] && r[e] != 0.0f;
            any_live |= live[e];
        }
        if (__ballot(any_live) == 0ull) {  // nothing but empty pixels
#pragma unroll
            for (int e = 0; e < ASSIGN_PX; e++)
                if (valid[e]) st_at(seg_b, (uint32_t)p[e] * (uint32_t)sizeof(L), (L)1);
            continue;
        }
        float lo0 = inf, lo1 = inf, lo2 = inf, hi0 = -inf, hi1 = -inf, hi2 = -inf;
        {   // empty pixels enter as quiet NaNs, which v_min3 / v_max3 skip: one select per coordinate, two pixels per instruction
            const float qnan = u2f(0x7FC00000u);
            float bx[ASSIGN_PX], by[ASSIGN_PX], bz[ASSIGN_PX];
#pragma unroll
            for (int e = 0; e < ASSIGN_PX; e++) { bx[e] = live[e] ? x[e] : qnan; by[e] = live[e] ? y[e] : qnan; bz[e] = live[e] ? z[e] : qnan; }
            static_assert(ASSIGN_PX % 2 == 0, "pixels in pairs");
#pragma unroll
            for (int e = 0; e < ASSIGN_PX; e += 2) {
                lo0 = fmin3_raw(lo0, bx[e], bx[e + 1]); hi0 = fmax3_raw(hi0, bx[e], bx[e + 1]);
                lo1 = fmin3_raw(lo1, by[e], by[e + 1]); hi1 = fmax3_raw(hi1, by[e], by[e + 1]);
                lo2 = fmin3_raw(lo2, bz[e], bz[e + 1]); hi2 = fmax3_raw(hi2, bz[e], bz[e + 1]);
            }
        }
        dpp_box6(lo0, lo1, lo2, hi0, hi1, hi2);
        // No centre beyond the largest reach of the tile's pixels (assign_reach) matters: on ground every centre is farther than the ground
        // term -- no survivor at all, the tile is labelled without a single distance --, on objects the nearest centre is known from the FPS.
        float reach = 0.0f;
        bool unknown = false;   // a live pixel the FPS knows nothing about (not a candidate, or no `temp`)
#pragma unroll
        for (int e = 0; e < ASSIGN_PX; e++) unknown |= live[e] && !(tp[e] >= 0.0f);
        if (__ballot(unknown) == 0ull) {   // (wave-uniform) candidates only -- objects: their nearest distances bound the tile, the ground term adds nothing
#pragma unroll
            for (int e = 0; e < ASSIGN_PX; e++) reach = fmaxf(reach, live[e] ? (tp[e] < 1e10f ? tp[e] : inf) : 0.0f);   // (a temp still at its initial 1e10 bounds nothing)
        } else {
#pragma unroll
            for (int e = 0; e < ASSIGN_PX; e++) reach = fmaxf(reach, live[e] ? assign_reach(r[e], tx[e], ty[e], tz[e], tp[e] < 1e10f ? tp[e] : -1.0f, G) : 0.0f);
        }
        reach = dpp_max_f32_native(reach);
        const bool bounded = reach < inf;   // (wave-uniform) otherwise: the box bound min_j dmax_j of the header
        // screen the centres: lane handles centres lane, lane+64, ...
        float my_dmin[NR], upper = inf;
#pragma unroll
        for (int rd = 0; rd < NR; rd++) {
            const int k = rd * 64 + lane;
            my_dmin[rd] = inf;
            if (rd * 64 < M && k < M) {
                const float4 cc = cen4[k];
                const float a0 = lo0 - cc.x, b0 = cc.x - hi0, a1 = lo1 - cc.y, b1 = cc.y - hi1, a2 = lo2 - cc.z, b2 = cc.z - hi2;
                const float g0 = fmaxf(fmaxf(a0, b0), 0.0f), g1 = fmaxf(fmaxf(a1, b1), 0.0f), g2 = fmaxf(fmaxf(a2, b2), 0.0f);
                my_dmin[rd] = (g0 * g0 + g1 * g1) + g2 * g2;
                if (!bounded) {
                    const float f0 = fmaxf(fabsf(a0), fabsf(b0)), f1 = fmaxf(fabsf(a1), fabsf(b1)), f2 = fmaxf(fabsf(a2), fabsf(b2));
                    upper = fminf(upper, (f0 * f0 + f1 * f1) + f2 * f2);
                }
            }
        }
        if (!bounded) upper = dpp_min_f32_native(upper);
        const float cut = (bounded ? reach : upper) * 1.000002f;
        unsigned long long surv_m[NR], surv_any = 0ull;
#pragma unroll
        for (int rd = 0; rd < NR; rd++) { surv_m[rd] = rd * 64 < M ? __ballot(my_dmin[rd] <= cut) : 0ull; surv_any |= surv_m[rd]; }
        if (surv_any == 0ull) {   // no centre within reach of any pixel: ground (0), empty pixels 1
#pragma unroll
            for (int e = 0; e < ASSIGN_PX; e++)
                if (valid[e]) st_at(seg_b, (uint32_t)p[e] * (uint32_t)sizeof(L), (L)(r[e] == 0.0f ? 1 : 0));
            continue;
        }
        float m1[ASSIGN_PX], m2[ASSIGN_PX];
        int k1[ASSIGN_PX];
        rs_v2f xv[ASSIGN_PX / 2], yv[ASSIGN_PX / 2], zv[ASSIGN_PX / 2];
#pragma unroll
        for (int e = 0; e < ASSIGN_PX; e++) { m1[e] = inf; m2[e] = inf; k1[e] = -1; }
#pragma unroll
        for (int e2 = 0; e2 < ASSIGN_PX / 2; e2++) {
            xv[e2] = rs_v2f{x[2 * e2], x[2 * e2 + 1]}; yv[e2] = rs_v2f{y[2 * e2], y[2 * e2 + 1]}; zv[e2] = rs_v2f{z[2 * e2], z[2 * e2 + 1]};
        }
#pragma unroll
        for (int rd = 0; rd < NR; rd++) {
            if (rd * 64 >= M) break;
            unsigned long long surv = surv_m[rd];
            while (surv) {
                const int k = rd * 64 + (int)__ffsll((long long)surv) - 1;
                surv &= surv - 1ull;
                const float4 cc = cen4[k];
                // the lane's pixels in pairs of packed fp32 (v_pk_add / v_pk_mul: each half rounds like the scalar operation)
#pragma unroll
                for (int e2 = 0; e2 < ASSIGN_PX / 2; e2++) {
                    const rs_v2f dx = xv[e2] - rs_v2f{cc.x, cc.x}, dy = yv[e2] - rs_v2f{cc.y, cc.y}, dz = zv[e2] - rs_v2f{cc.z, cc.z};
                    const rs_v2f dd = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
                    for (int e1 = 0; e1 < 2; e1++) {
                        const int e = 2 * e2 + e1;
                        const float d2 = e1 ? dd.y : dd.x;
                        // two smallest squared distances: m1 <= m2 always, so the new runner-up is the median of (m1, m2, d2)
                        const bool lt = d2 < m1[e];
                        m2[e] = __builtin_amdgcn_fmed3f(m1[e], m2[e], d2);
                        k1[e] = lt ? k : k1[e];
                        m1[e] = lt ? d2 : m1[e];   // (the compare's mask again: fminf would add a canonicalising v_max)
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < ASSIGN_PX; e++) {
            int label = assign_label(r[e], tx[e], ty[e], tz[e], x[e], y[e], z[e], m1[e], m2[e], k1[e], cen4, G);
            if (r[e] == 0.0f) label = 1;
            if (valid[e]) st_at(seg_b, (uint32_t)p[e] * (uint32_t)sizeof(L), (L)label);
        }
    }
}

static int launch_assign(const float *ri, const float *tm, const double *ground, const float *centers, int B, int H,
                         int W, int M, uint8_t *seg, hipStream_t st, const float *temp = nullptr) {
    const int ntile = ((H + ASSIGN_ROWS - 1) / ASSIGN_ROWS) * ((W + 31) / 32);
    const dim3 grid((ntile + ASSIGN_WAVES * ASSIGN_TILES_PER_WAVE - 1) / (ASSIGN_WAVES * ASSIGN_TILES_PER_WAVE), B);
    assign_kernel<uint8_t, 4><<<grid, 64 * ASSIGN_WAVES, (size_t)M * sizeof(float4), st>>>(ri, tm, ground, centers, H, W, M, seg, temp);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_assign(const float *ri, const float *tm, const double *ground, const float *centers, int B, int H,
                           int W, int M, uint8_t *seg, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS && ri && tm && ground && centers && seg);
    return launch_assign(ri, tm, ground, centers, B, H, W, M, seg, (hipStream_t)stream);
}

// ================================================================================================
// a8  point model  (cpp_modules.cpp:471-518)  +  per-tile label histograms for the ordered scatter
// ================================================================================================
// Workspace layout (rpcc_workspace_bytes):  [ sums i64 B*KP | flags i32 B*4 (pad to 16 B) | hist u32 B*T*KP ]
// KP = K rounded up to 64, T = ceil(P / TILE).  hist[b][t][k] is first the pixel count of label k in
// tile t, then (after model_scan_kernel) the output offset of that tile's first label-k pixel.
#define TILE 1024
#define FX_UNIT_LOG2 28  // fixed-point unit 2^-28 m: exact for 2^-5 <= r < 2^8 (DESIGN.md "point model")

static inline int kpad(int M) { return ((M + 2) + 63) & ~63; }
static inline int ntiles(int P) { return (P + TILE - 1) / TILE; }
struct WsLayout {
    int64_t *sums;
    int32_t *flags;
    uint32_t *hist;
    size_t bytes;
};
static WsLayout ws_layout(void *ws, int B, int P, int M) {
    WsLayout L;
    const size_t KP = (size_t)kpad(M), T = (size_t)ntiles(P);
    char *p = reinterpret_cast<char *>(ws);
    L.sums = reinterpret_cast<int64_t *>(p);
    size_t off = (size_t)B * KP * 8;
    L.flags = reinterpret_cast<int32_t *>(p + off);
    off += (((size_t)B * 4 * 4) + 255) & ~(size_t)255;
    L.hist = reinterpret_cast<uint32_t *>(p + off);
    off += (size_t)B * T * KP * 4;
    L.bytes = off;
    return L;
}
// Workspace of one batch: [ model part | projection scratch | FPS temp | planar rays | FPS tile table ]
static size_t plane_extra_bytes(int B, int P, int M) {   // label-ordered pixel list u32 [B,P] | points float4 [B,P] | key points per label i32 [B,K] | label steps f32 [B,K]
    return (((size_t)B * P * 4 + 255) & ~(size_t)255) + (size_t)B * P * 16 + 2 * (((size_t)B * (M + 2) * 4 + 255) & ~(size_t)255) + 256;
}
static size_t slice_workspace_bytes(int B, int P, int M, int64_t total_points) {
    const size_t model_ws = ws_layout(nullptr, B, P, M).bytes;
    const size_t proj_ws = (project_scratch_bytes(total_points, B, P) + 255) & ~(size_t)255;
    return model_ws + 256 + proj_ws + (size_t)B * P * 4        // + FPS temp [B,P] f32
           + (size_t)3 * P * 4 + 256                            // + SoA copy of the ray table
           + (size_t)B * FPS_TAB_ROWS * ((P + 31) / 32 + 4096) * 4 + 256;  // + FPS tile table (generous bound)
}
extern "C" size_t rpcc_workspace_bytes(int B, int P, int M, int64_t total_points) {
    if (B <= 0 || P <= 0 || M <= 0) return 0;
    return slice_workspace_bytes(B, P, M, total_points) + 4096;
}
extern "C" size_t rpcc_workspace_bytes_general(int B, int P, int M, int64_t total_points) {
    if (B <= 0 || P <= 0 || M <= 0) return 0;
    return slice_workspace_bytes(B, P, M, total_points) + 4096 + plane_extra_bytes(B, P, M);
}

// Round 3: thread k walks label k's column of the tile x label table straight from global memory (the loads of consecutive
// threads are consecutive words; SCAN_U tiles' loads in flight per thread) in two passes -- totals, then offsets with the label's
// base added -- instead of staging the table in LDS (64 KB per workgroup for 64x2048: its footprint kept everything else off
// the CU, and the 128 dependent LDS round trips of the in-LDS scan were most of the kernel's time).
#define SCAN_U 16
#define SCAN_THREADS 256
// Thread (g, k): label k of tile group g.  The SCAN_THREADS threads form NG = SCAN_THREADS / KP2 groups (KP2 = K rounded up to a
// power of two: 2 groups for 102 labels), a group owns T / NG consecutive tiles; with at most SCAN_U tiles per group a thread's
// counts stay in registers between the totals and the offsets (one trip to memory per thread), otherwise SCAN_U loads are in flight
// per trip.  This kernel sits on every batch's chain between the histogram and the quantiser, and what counts there is how soon its
// one workgroup per frame finds room on CUs that other batches' kernels fill: 512 threads x 144 registers holding 32 tiles each were
// the fastest alone (10.2 us) and took 65 us in flight; 256 threads with 16 loads in flight (four trips for 64x2048) are slower alone
// and 1.5 % faster per step with three batches in flight (profiles/raw/r06_ab_scan_footprint.txt: 512/32, 512/16, 512/8, 256/32, 256/24, 256/16, 256/8).
struct ScanArgs {
    const float *ri;
    const uint8_t *seg;
    const double *ground;
    int P, M, KP, T, KP2;
    const int64_t *sums;
    const int32_t *flags;
    uint32_t *hist;
    float *model;
    int32_t *counts, *nnz;
};
__device__ __forceinline__ void model_scan_body(const ScanArgs &A, const int b) {
    const float *__restrict__ ri = A.ri;
    const uint8_t *__restrict__ seg = A.seg;
    const double *__restrict__ ground = A.ground;
    const int P = A.P, M = A.M, KP = A.KP, T = A.T, KP2 = A.KP2;
    const int64_t *__restrict__ sums = A.sums;
    const int32_t *__restrict__ flags = A.flags;
    uint32_t *__restrict__ hist = A.hist;
    float *__restrict__ model = A.model;
    int32_t *__restrict__ counts = A.counts, *__restrict__ nnz = A.nnz;
    __shared__ uint32_t gtot[SCAN_THREADS];   // [group][KP2] counts of a label in a tile group
    __shared__ uint32_t base[256];
    __shared__ uint32_t wsum[4];
    const int K = M + 2;
    const int NG = SCAN_THREADS / KP2, k = threadIdx.x & (KP2 - 1), grp = threadIdx.x / KP2;
    const int TG = (T + NG - 1) / NG, tbeg = grp * TG, tend = min(T, tbeg + TG);
    uint32_t *gh = hist + (int64_t)b * T * KP;
    const uint32_t kp4 = (uint32_t)KP * 4u, k4 = (uint32_t)k * 4u;
    const bool held = TG <= SCAN_U;   // (workgroup-uniform)
    uint32_t c[SCAN_U];
    uint32_t part = 0;
    if (k < K) {
        if (held) {
#pragma unroll
            for (int j = 0; j < SCAN_U; j++) c[j] = ld_at(gh, (uint32_t)min(tbeg + j, T - 1) * kp4 + k4);   // unconditional (clamped) loads
#pragma unroll
            for (int j = 0; j < SCAN_U; j++) { if (tbeg + j >= tend) c[j] = 0u; part += c[j]; }
        } else {
            for (int t0 = tbeg; t0 < tend; t0 += SCAN_U) {
                uint32_t d[SCAN_U];
#pragma unroll
                for (int j = 0; j < SCAN_U; j++) d[j] = ld_at(gh, (uint32_t)min(t0 + j, T - 1) * kp4 + k4);
#pragma unroll
                for (int j = 0; j < SCAN_U; j++) part += t0 + j < tend ? d[j] : 0u;
            }
        }
    }
    gtot[grp * KP2 + k] = k < K ? part : 0u;
    __syncthreads();
    // label totals, then their exclusive prefix (label 1 = empty pixels has no residuals): DPP scans of the first 256 threads
    uint32_t total = 0;
    for (int g2 = 0; g2 < NG; g2++) total += gtot[g2 * KP2 + k];
    {
        const int kk = threadIdx.x;   // label kk for the prefix (threads 0 .. 255)
        uint32_t v = 0u;
        if (kk < 256 && kk < K && kk != 1) { for (int g2 = 0; g2 < NG; g2++) v += gtot[g2 * KP2 + kk]; }
        const uint32_t incl = dpp_scan_incl_u32(v);
        if (kk < 256 && (kk & 63) == 63) wsum[kk >> 6] = incl;
        __syncthreads();
        if (kk < 256) {
            uint32_t off = 0u;
            for (int w = 0; w < (kk >> 6); w++) off += wsum[w];
            base[kk] = off + incl - v;
            if (kk == 255 && nnz) nnz[b] = (int32_t)(off + incl);
        }
    }
    __syncthreads();
    if (k < K) {
        uint32_t run = base[k];   // exclusive prefix over tiles, seeded with the label's base and the groups before
        for (int g2 = 0; g2 < grp; g2++) run += gtot[g2 * KP2 + k];
        if (held) {
#pragma unroll
            for (int j = 0; j < SCAN_U; j++) {
                if (tbeg + j < tend) { st_at(gh, (uint32_t)(tbeg + j) * kp4 + k4, run); run += c[j]; }
            }
        } else {
            for (int t0 = tbeg; t0 < tend; t0 += SCAN_U) {
                uint32_t d[SCAN_U];
#pragma unroll
                for (int j = 0; j < SCAN_U; j++) d[j] = ld_at(gh, (uint32_t)min(t0 + j, T - 1) * kp4 + k4);
#pragma unroll
                for (int j = 0; j < SCAN_U; j++) {
                    if (t0 + j < tend) { st_at(gh, (uint32_t)(t0 + j) * kp4 + k4, run); run += d[j]; }
                }
            }
        }
        if (counts && grp == 0) counts[(int64_t)b * K + k] = (int32_t)total;
    }
    if (k < K && grp == 0 && model != nullptr) {
        float *row = model + ((int64_t)b * K + k) * 4;
        if (k == 0) {
            row[0] = (float)ground[4 * b]; row[1] = (float)ground[4 * b + 1];
            row[2] = (float)ground[4 * b + 2]; row[3] = (float)ground[4 * b + 3];
        } else if (k == 1) {
            row[0] = row[1] = row[2] = row[3] = 0.0f;
        } else {
            double s;
            if (flags[4 * b]) {
                // exact sequential double accumulation in row-major order (cpp_modules.cpp:514) for frames
                // whose ranges fall outside the fixed-point window; one thread per label, rare.
                s = 0.0;
                const uint8_t *sg = seg + (int64_t)b * P;
                const float *rr = ri + (int64_t)b * P;
                for (int p = 0; p < P; p++)
                    if (sg[p] == k) s += (double)rr[p];
            } else {
                s = (double)sums[(int64_t)b * KP + k] * (1.0 / 268435456.0);  // exact: sum < 2^53 units
            }
            const double n = (double)total;
            row[0] = row[1] = row[2] = 0.0f;
            row[3] = (total == 0) ? u2f(0xFFC00000u) : (float)(s / n);  // 0.0/0 on x86 = default NaN
        }
    }
}
__global__ __launch_bounds__(SCAN_THREADS) void model_scan_kernel(const ScanArgs A) { model_scan_body(A, blockIdx.x); }
// (the frames of several geometry groups in one launch, rpcc_compress_batch_mixed: one workgroup per frame, per[] = 1)
__global__ __launch_bounds__(SCAN_THREADS) void model_scan_multi_kernel(const MultiArgs<ScanArgs> m) {
    int b, t;
    const ScanArgs &A = multi_locate(m, b, t);
    model_scan_body(A, b);
}
// A lane owns FOUR CONSECUTIVE pixels of the tile (VEC: one 4-byte load of labels, one 16-byte load of ranges), a wavefront
// 256 consecutive pixels.  Labels are spatially coherent, so the pixels that carry the label of the wavefront's first pixel
// -- usually most of the 256 -- are aggregated once per wavefront (four compare masks counted in scalar registers, two DPP
// sums for the range total), the others add themselves to LDS directly: integer sums, any order.  The fixed-point value of a
// range r in [2^-5, 2^8) is read off its bit pattern: r * 2^28 = mantissa << (exponent + 5), an integer below 2^36, kept as
// an 18-bit low part and a high part whose sums over a wavefront stay below 2^32.
template <bool VEC, class L = uint8_t>
__device__ __forceinline__ void model_hist_body(const float *__restrict__ ri, const L *__restrict__ seg,
                                                int P, int KP, int T, int64_t *__restrict__ sums,
                                                int32_t *__restrict__ flags, uint32_t *__restrict__ hist, const int b, const int t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long *ssum = reinterpret_cast<unsigned long long *>(smem_raw);  // [KP]
    uint32_t *scnt = reinterpret_cast<uint32_t *>(ssum + KP);                      // [KP]
    const int lane = threadIdx.x & 63;
    const L *seg_b = seg + (int64_t)b * P;
    const float *ri_b = ri != nullptr ? ri + (int64_t)b * P : nullptr;
    const bool want_sum = ri != nullptr;
    const int p0 = t * TILE + 4 * (int)threadIdx.x;
    const int nval = min(max(P - p0, 0), 4);
    int lab[4];
    uint32_t rb[4];
    // all loads first (unconditional, clamped)
    if (VEC) {
        const uint32_t q = (uint32_t)(nval > 0 ? p0 : 0);
        if (sizeof(L) == 1) {
            const uint32_t l4 = ld_at(reinterpret_cast<const uint32_t *>(seg_b), q);
            lab[0] = (int)(l4 & 255u); lab[1] = (int)((l4 >> 8) & 255u); lab[2] = (int)((l4 >> 16) & 255u); lab[3] = (int)(l4 >> 24);
        } else {   // four 16-bit labels: one 8-byte load
            const uint2 l8 = ld_at(reinterpret_cast<const uint2 *>(seg_b), q * 2u);
            lab[0] = (int)(l8.x & 0xFFFFu); lab[1] = (int)(l8.x >> 16); lab[2] = (int)(l8.y & 0xFFFFu); lab[3] = (int)(l8.y >> 16);
        }
        if (want_sum) {
            const uint4 r4 = ld_at(reinterpret_cast<const uint4 *>(ri_b), q * 4u);
            rb[0] = r4.x; rb[1] = r4.y; rb[2] = r4.z; rb[3] = r4.w;
        } else {
            rb[0] = rb[1] = rb[2] = rb[3] = 0x3F800000u;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t gp = (uint32_t)min(p0 + e, P - 1);
            lab[e] = ld_at(seg_b, gp * (uint32_t)sizeof(L));
            rb[e] = want_sum ? f2u(ld_at(ri_b, gp * 4u)) : 0x3F800000u;
        }
    }
    for (int k = threadIdx.x; k < KP; k += blockDim.x) { ssum[k] = 0ull; scnt[k] = 0u; }  // while the loads are in flight
    __syncthreads();
    bool inexact = false;
    int todo[4];
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        todo[e] = e < nval ? lab[e] : -1;
        lo[e] = hi[e] = 0u;
        if (want_sum && todo[e] >= 2) {
            const float r = u2f(rb[e]);
            if (!(r >= 0.03125f && r < 256.0f)) {
                inexact = true;
            } else {   // r * 2^28, exact (see above): biased exponent 122 .. 134
                const uint32_t sh = (rb[e] >> 23) - 122u, m = (rb[e] & 0x7FFFFFu) | 0x800000u;
                lo[e] = (m << sh) & 0x3FFFFu;
                hi[e] = m >> (18u - sh);
            }
        }
    }
    // up to HIST_ROUNDS labels of the wavefront are aggregated (the label of the first pixel still pending: 256 consecutive
    // pixels hold four labels on average); what is left adds itself to LDS pixel by pixel.  The pending pixels are four lane
    // masks in scalar registers (one per pixel slot), as in the quantiser.
#define HIST_ROUNDS 3
    unsigned long long pend[4];
#pragma unroll
    for (int e = 0; e < 4; e++) pend[e] = __ballot(todo[e] >= 0);
#pragma unroll 1
    for (int round = 0; round < HIST_ROUNDS && (pend[0] | pend[1] | pend[2] | pend[3]) != 0ull; round++) {
        int cur;
        if (pend[0])      cur = __builtin_amdgcn_readlane(todo[0], (int)__ffsll((long long)pend[0]) - 1);
        else if (pend[1]) cur = __builtin_amdgcn_readlane(todo[1], (int)__ffsll((long long)pend[1]) - 1);
        else if (pend[2]) cur = __builtin_amdgcn_readlane(todo[2], (int)__ffsll((long long)pend[2]) - 1);
        else              cur = __builtin_amdgcn_readlane(todo[3], (int)__ffsll((long long)pend[3]) - 1);
        int ctot = 0;
        uint32_t slo = 0u, shi = 0u;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const bool mine = todo[e] == cur;
            const unsigned long long m = __ballot(mine);
            ctot += (int)__popcll(m);
            pend[e] &= ~m;
            slo += mine ? lo[e] : 0u;
            shi += mine ? hi[e] : 0u;
        }
        const bool sum = want_sum && cur >= 2;
        if (sum) { slo = dpp_sum_u32(slo); shi = dpp_sum_u32(shi); }
        if (lane == 0) {
            atomicAdd(&scnt[cur], (uint32_t)ctot);
            if (sum) atomicAdd(&ssum[cur], (unsigned long long)slo + ((unsigned long long)shi << 18));
        }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (pend[e] != 0ull && ((pend[e] >> lane) & 1ull)) {   // (first test wave-uniform: usually nothing is left)
            atomicAdd(&scnt[todo[e]], 1u);
            if (want_sum && todo[e] >= 2) atomicAdd(&ssum[todo[e]], (unsigned long long)lo[e] + ((unsigned long long)hi[e] << 18));
        }
    }
    if (__any(inexact) && lane == 0) flags[4 * b] = 1;
    __syncthreads();
    for (int k = threadIdx.x; k < KP; k += blockDim.x) {
        hist[((int64_t)b * T + t) * KP + k] = scnt[k];
        if (ssum[k]) atomicAdd(reinterpret_cast<unsigned long long *>(&sums[(int64_t)b * KP + k]), ssum[k]);
    }
}
template <bool VEC, class L = uint8_t>
__global__ __launch_bounds__(256) void model_hist_kernel(const float *__restrict__ ri, const L *__restrict__ seg,
                                                         int P, int KP, int T, int64_t *__restrict__ sums,
                                                         int32_t *__restrict__ flags, uint32_t *__restrict__ hist) {
    model_hist_body<VEC, L>(ri, seg, P, KP, T, sums, flags, hist, blockIdx.y, blockIdx.x);
}
struct HistGroup {
    const float *ri;
    const uint8_t *seg;
    int P, T, vec;
    int64_t *sums;
    int32_t *flags;
    uint32_t *hist;
};
__global__ __launch_bounds__(256) void model_hist_multi_kernel(const MultiArgs<HistGroup> m, int KP) {
    int b, t;
    const HistGroup &a = multi_locate(m, b, t);
    if (a.vec) model_hist_body<true>(a.ri, a.seg, a.P, KP, a.T, a.sums, a.flags, a.hist, b, t);
    else       model_hist_body<false>(a.ri, a.seg, a.P, KP, a.T, a.sums, a.flags, a.hist, b, t);
}
static inline int scan_kp2(int M) { int v = 1; while (v < M + 2) v <<= 1; return v; }   // labels rounded up to a power of two (<= 256)
// histogram + scan: tile x label counts (and the labels' range sums when ri is given), then the offsets of the ordered scatter, counts, nnz and
// -- with `model` -- the point model's rows.  The sums / flags block must be zero (memset or BatchInit).
static int launch_hist_scan(const float *ri, const uint8_t *seg, const double *ground, int B, int P, int M, const WsLayout &L, float *model,
                            int32_t *counts, int32_t *nnz, hipStream_t st) {
    const int KP = kpad(M), T = ntiles(P);
    const bool vec = (P & 3) == 0 && ((uintptr_t)seg & 3u) == 0 && (ri == nullptr || ((uintptr_t)ri & 15u) == 0);
    const ScanArgs sa = {ri, seg, ground, P, M, KP, T, scan_kp2(M), L.sums, L.flags, L.hist, model, counts, nnz};
    // (The scan as the tail of the histogram kernel -- the frame's last workgroup to finish runs it -- was measured in round 4: a workgroup must
    // release its rows device-wide before it draws its ticket, and an agent-scope fence writes the XCD's L2 back: 62 us -> 2.9 ms.)
    if (vec) model_hist_kernel<true><<<dim3(T, B), 256, (size_t)KP * 12, st>>>(ri, seg, P, KP, T, L.sums, L.flags, L.hist);
    else     model_hist_kernel<false><<<dim3(T, B), 256, (size_t)KP * 12, st>>>(ri, seg, P, KP, T, L.sums, L.flags, L.hist);
    LAUNCH_CHECK();
    model_scan_kernel<<<B, SCAN_THREADS, 0, st>>>(sa);
    LAUNCH_CHECK();
    return RPCC_OK;
}

// One workgroup per frame: label totals, tile offsets, label bases, means, model rows.
static int launch_point_model(const float *ri, const uint8_t *seg, const double *ground, int B, int P, int M,
                              float *model, int32_t *counts, int32_t *nnz, void *ws, hipStream_t st, bool cleared = false) {
    const int KP = kpad(M), T = ntiles(P);
    WsLayout L = ws_layout(ws, B, P, M);
    if (!cleared) HIP_TRY(hipMemsetAsync(L.sums, 0, (size_t)((char *)L.hist - (char *)L.sums), st));
    return launch_hist_scan(ri, seg, ground, B, P, M, L, model, counts, nnz, st);
}

extern "C" int rpcc_point_model(const float *ri, const uint8_t *seg, const double *ground, int B, int P, int M,
                                float *model, int32_t *counts, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS && ri && seg && ground && model && ws);
    return launch_point_model(ri, seg, ground, B, P, M, model, counts, nullptr, ws, (hipStream_t)stream);
}

// ================================================================================================
// a10 + a11  intra-prediction, residual, uniform quantisation, ordered scatter
//            (cpp_modules.cpp:248-285, tools/compress.py:106, cpp_modules.cpp:288-334)
// ================================================================================================
// A 256-thread workgroup owns one 1024-pixel tile as 16 segments (4 passes x 4 waves) of 64
// consecutive pixels.  Rank of a pixel inside its label = tile offset (model_scan_kernel) + pixels of
// that label in earlier segments + earlier lanes of its own segment (ballot + popcount).
// exclusive prefix over the 16 segments of every label, seeded with the tile's offset: one 16-lane DPP row per
// label (row_shr scan), 16 labels per pass of the workgroup.  segcnt is [16][SEGP] with SEGP = KP + 1 (odd
// stride: the 16 lanes of a row hit 16 different banks).
__device__ __forceinline__ void segment_prefix(uint32_t *segcnt, int SEGP, const uint32_t *tile_off, int K) {
    const int row = threadIdx.x >> 4, sgi = threadIdx.x & 15;  // 16 rows of 16 lanes in a 256-thread workgroup
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int k = k0 + row;
        const bool ok = k < K;
        const uint32_t c = ok ? segcnt[sgi * SEGP + k] : 0u;
        uint32_t v = c;  // inclusive scan inside the 16-lane row
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
        if (ok) segcnt[sgi * SEGP + k] = tile_off[k] + v - c;
    }
}

// RESIDUAL_ONLY: the quantiser's own seam (uniform_quantize(seg_idx, residual, acc)): no prediction, so ri, tm and
// model are not read at all (they may be NULL).
//
// Round 3 layout: a lane owns FOUR CONSECUTIVE pixels of the tile (VEC: one 4-byte load of labels, 16-byte loads of range,
// rays and residual), a wavefront 256 consecutive pixels, the workgroup's four wavefronts the four quarters of the tile in
// order.  Output position of a pixel = tile offset of its label (model_scan_kernel) + pixels of that label in the wavefronts
// before + in the lanes before (four compare masks per label of the wavefront, counted with mbcnt) + in the lane's own
// earlier pixels.  The cross-wavefront prefix is four counters per label, one thread per label.  (Until round 3 the tile
// was 16 segments of 64 pixels, one pixel per lane and pass: a 16 x K counter matrix, cleared and prefix-scanned per tile.)
template <bool RESIDUAL_ONLY, bool VEC, class L = uint8_t>
__device__ __forceinline__ void predict_quantize_body(const float *__restrict__ ri, const float *__restrict__ tm,
                                                      const L *__restrict__ seg,
                                                      const float *__restrict__ model,
                                                      const uint32_t *__restrict__ hist, float acc,
                                                      const float *__restrict__ label_acc,
                                                      const float *__restrict__ residual_in, int P, int M,
                                                      int KP, int T, int16_t *__restrict__ q16,
                                                      int32_t *__restrict__ q32, float *__restrict__ pred_out,
                                                      int32_t *__restrict__ epoch_inc, const int b, const int t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // last kernel of a fused batch: the next call's projection flags get a new mark (BatchInit)
    if (epoch_inc && b == 0 && t == 0 && threadIdx.x == 0) *epoch_inc += 1;
    float4 *smodel = reinterpret_cast<float4 *>(smem_raw);               // [KP] model rows
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smodel + KP);          // [4][KP]: pixels of label k in wavefront w -> its first output slot
    uint32_t *soff = wcnt + 4 * KP;                                      // [KP] this tile's output offsets per label
    float *sacc = reinterpret_cast<float *>(soff + KP);                  // [KP] quantisation step per label (non-uniform)
    const int K = M + 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-frame bases (wave-uniform): scalar base + 32-bit lane offset instead of 64-bit address arithmetic per access
    seg += (int64_t)b * P;
    if (!RESIDUAL_ONLY) ri += (int64_t)b * P;
    if (residual_in) residual_in += (int64_t)b * P;
    if (pred_out) pred_out += (int64_t)b * P;
    if (q16) q16 += (int64_t)b * P;
    if (q32) q32 += (int64_t)b * P;
    if (label_acc) label_acc += (int64_t)b * K;
    if (!RESIDUAL_ONLY) model += (int64_t)b * K * 4;
    hist += ((int64_t)b * T + t) * KP;
    const int p0 = t * TILE + 4 * (int)threadIdx.x;
    const int nval = min(max(P - p0, 0), 4);
    // all global loads first, unconditional on clamped indices (a guarded load is waited for on the spot)
    int lab[4];
    float rv[4], ray[12], rin[4];
    if (VEC) {
        const uint32_t q = (uint32_t)(nval > 0 ? p0 : 0);
        if (sizeof(L) == 1) {
            const uint32_t l4 = ld_at(reinterpret_cast<const uint32_t *>(seg), q);
            lab[0] = (int)(l4 & 255u); lab[1] = (int)((l4 >> 8) & 255u); lab[2] = (int)((l4 >> 16) & 255u); lab[3] = (int)(l4 >> 24);
        } else {   // four 16-bit labels: one 8-byte load
            const uint2 l8 = ld_at(reinterpret_cast<const uint2 *>(seg), q * 2u);
            lab[0] = (int)(l8.x & 0xFFFFu); lab[1] = (int)(l8.x >> 16); lab[2] = (int)(l8.y & 0xFFFFu); lab[3] = (int)(l8.y >> 16);
        }
        if (RESIDUAL_ONLY) {
#pragma unroll
            for (int e = 0; e < 4; e++) rv[e] = 0.0f;
#pragma unroll
            for (int i = 0; i < 12; i++) ray[i] = 0.0f;
        } else {
            const float4 r4 = ld_at(reinterpret_cast<const float4 *>(ri), q * 4u);
            rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
            const float4 a = ld_at(reinterpret_cast<const float4 *>(tm), q * 12u), bq = ld_at(reinterpret_cast<const float4 *>(tm), q * 12u + 16u),
                         c = ld_at(reinterpret_cast<const float4 *>(tm), q * 12u + 32u);
            ray[0] = a.x; ray[1] = a.y; ray[2] = a.z; ray[3] = a.w; ray[4] = bq.x; ray[5] = bq.y; ray[6] = bq.z; ray[7] = bq.w;
            ray[8] = c.x; ray[9] = c.y; ray[10] = c.z; ray[11] = c.w;
        }
        if (residual_in) {
            const float4 x4 = ld_at(reinterpret_cast<const float4 *>(residual_in), q * 4u);
            rin[0] = x4.x; rin[1] = x4.y; rin[2] = x4.z; rin[3] = x4.w;
        } else {
            rin[0] = rin[1] = rin[2] = rin[3] = 0.0f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t up = (uint32_t)min(p0 + e, P - 1);
            lab[e] = ld_at(seg, up * (uint32_t)sizeof(L));
            rv[e] = RESIDUAL_ONLY ? 0.0f : ld_at(ri, up * 4u);
            if (RESIDUAL_ONLY) { ray[3 * e] = ray[3 * e + 1] = ray[3 * e + 2] = 0.0f; }
            else { ray[3 * e] = ld_f32(tm, up * 12u); ray[3 * e + 1] = ld_f32(tm, up * 12u + 4u); ray[3 * e + 2] = ld_f32(tm, up * 12u + 8u); }
            rin[e] = residual_in ? ld_at(residual_in, up * 4u) : 0.0f;
        }
    }
    // staging (byte labels: K <= 256, one step each); model rows as 16-byte copies
    for (uint32_t i = threadIdx.x; i < (uint32_t)K; i += 256u) {   // (one trip for byte labels)
        smodel[i] = RESIDUAL_ONLY ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(model[4 * i], model[4 * i + 1], model[4 * i + 2], model[4 * i + 3]);
        soff[i] = hist[i];
        if (label_acc) sacc[i] = label_acc[i];
        wcnt[i] = 0u; wcnt[KP + i] = 0u; wcnt[2 * KP + i] = 0u; wcnt[3 * KP + i] = 0u;
        if (sizeof(L) == 1) break;
    }
    __syncthreads();
    int qv[4], rank[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int l = lab[e];
        int keep = -1;
        qv[e] = 0;
        rank[e] = 0;
        if (e < nval) {
            const float4 pm = smodel[l];
            float pr;
            if (pm.x + pm.y + pm.z == 0.0f) pr = pm.w;                                          // cpp_modules.cpp:271-272
            else pr = -pm.w / (pm.x * ray[3 * e] + pm.y * ray[3 * e + 1] + pm.z * ray[3 * e + 2]);  // :275-277
            if (!RESIDUAL_ONLY && pred_out) st_at(pred_out, (uint32_t)(p0 + e) * 4u, pr);
            const float res = (RESIDUAL_ONLY || residual_in) ? rin[e] : rv[e] - pr;            // compress.py:106
            const float step = label_acc ? sacc[l] : acc;                                      // cpp_modules.cpp:404,419
            qv[e] = (int)roundf(res / step);                                                    // cpp_modules.cpp:315
            keep = (l == 1) ? -1 : l;                                                           // label 1 is skipped (:314)
        }
        lab[e] = keep;
    }
    // rank among equal labels inside the wavefront's 256 pixels, one label of the wavefront per round.  Which pixels are still
    // unranked is kept as four lane masks in SCALAR registers (one per pixel slot): a round takes the label of the first
    // pending pixel of the lowest pending slot, and every pixel that carries it is ranked in that round, so a label compare
    // alone finds the round's pixels -- no per-lane bookkeeping in vector registers.
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long pend[4];
#pragma unroll
    for (int e = 0; e < 4; e++) pend[e] = __ballot(lab[e] >= 0);
    while ((pend[0] | pend[1] | pend[2] | pend[3]) != 0ull) {
        int cur;
        if (pend[0])      cur = __builtin_amdgcn_readlane(lab[0], (int)__ffsll((long long)pend[0]) - 1);
        else if (pend[1]) cur = __builtin_amdgcn_readlane(lab[1], (int)__ffsll((long long)pend[1]) - 1);
        else if (pend[2]) cur = __builtin_amdgcn_readlane(lab[2], (int)__ffsll((long long)pend[2]) - 1);
        else              cur = __builtin_amdgcn_readlane(lab[3], (int)__ffsll((long long)pend[3]) - 1);
        unsigned long long m[4];
        int below = 0, total = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            m[e] = __ballot(lab[e] == cur);
            below += (int)__popcll(m[e] & lt);     // matching pixels in the lanes before (mbcnt)
            total += (int)__popcll(m[e]);          // (scalar)
            pend[e] &= ~m[e];
        }
        int own = below;                            // + matching pixels of this lane before slot e
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const bool mine = lab[e] == cur;
            rank[e] = mine ? own : rank[e];
            own += mine ? 1 : 0;
        }
        if (lane == 0) wcnt[wave * KP + cur] = (uint32_t)total;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)K; i += 256u) {   // exclusive prefix over the four wavefronts, seeded with the tile's offset
        const uint32_t c0 = wcnt[i], c1 = wcnt[KP + i], c2 = wcnt[2 * KP + i], base = soff[i];
        wcnt[i] = base; wcnt[KP + i] = base + c0; wcnt[2 * KP + i] = base + c0 + c1; wcnt[3 * KP + i] = base + c0 + c1 + c2;
        if (sizeof(L) == 1) break;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (lab[e] >= 0) {
            const uint32_t o = wcnt[wave * KP + lab[e]] + (uint32_t)rank[e];
            if (q16) st_at(q16, o * 2u, (int16_t)qv[e]);  // astype(np.int16): two's-complement truncation
            if (q32) st_at(q32, o * 4u, (int32_t)qv[e]);
        }
    }
}

template <bool RESIDUAL_ONLY, bool VEC, class L = uint8_t>
__global__ __launch_bounds__(256) void predict_quantize_kernel(const float *__restrict__ ri, const float *__restrict__ tm,
                                                               const L *__restrict__ seg,
                                                               const float *__restrict__ model,
                                                               const uint32_t *__restrict__ hist, float acc,
                                                               const float *__restrict__ label_acc,
                                                               const float *__restrict__ residual_in, int P, int M,
                                                               int KP, int T, int16_t *__restrict__ q16,
                                                               int32_t *__restrict__ q32, float *__restrict__ pred_out,
                                                               int32_t *__restrict__ epoch_inc) {
    predict_quantize_body<RESIDUAL_ONLY, VEC, L>(ri, tm, seg, model, hist, acc, label_acc, residual_in, P, M, KP, T, q16, q32, pred_out, epoch_inc, blockIdx.y, blockIdx.x);
}
struct QuantGroup {   // one geometry group of rpcc_compress_batch_mixed (the fused batch's form: prediction from model rows, int16 output)
    const float *ri, *tm, *model, *label_acc;
    const uint8_t *seg;
    const uint32_t *hist;
    int P, T, vec;
    int16_t *q16;
    int32_t *epoch_inc;
};
__global__ __launch_bounds__(256) void predict_quantize_multi_kernel(const MultiArgs<QuantGroup> m, float acc, int M, int KP) {
    int b, t;
    const QuantGroup &a = multi_locate(m, b, t);
    if (a.vec) predict_quantize_body<false, true>(a.ri, a.tm, a.seg, a.model, a.hist, acc, a.label_acc, nullptr, a.P, M, KP, a.T, a.q16, nullptr, nullptr, a.epoch_inc, b, t);
    else       predict_quantize_body<false, false>(a.ri, a.tm, a.seg, a.model, a.hist, acc, a.label_acc, nullptr, a.P, M, KP, a.T, a.q16, nullptr, nullptr, a.epoch_inc, b, t);
}

static int launch_predict_quantize(const float *ri, const float *tm, const uint8_t *seg, const float *model, float acc,
                                   const float *label_acc, const float *residual_in, int B, int P, int M, int16_t *q16,
                                   int32_t *q32, float *pred, void *ws, hipStream_t st, int32_t *epoch_inc = nullptr) {
    const int KP = kpad(M), T = ntiles(P);
    WsLayout L = ws_layout(ws, B, P, M);
    const size_t sh = (size_t)KP * 16 + (size_t)4 * KP * 4 + (size_t)KP * 4 + (size_t)KP * 4;
    const bool resid = residual_in && !pred;
    const bool vec = (P & 3) == 0 && ((uintptr_t)seg & 3u) == 0 && (resid || (aligned16(ri) && aligned16(tm))) &&
                     (!residual_in || aligned16(residual_in));
#define PQ_LAUNCH(R_, V_) predict_quantize_kernel<R_, V_><<<dim3(T, B), 256, sh, st>>>(ri, tm, seg, model, L.hist, acc, label_acc, residual_in, P, M, KP, T, q16, q32, pred, epoch_inc)
    if (resid) { if (vec) PQ_LAUNCH(true, true); else PQ_LAUNCH(true, false); }
    else       { if (vec) PQ_LAUNCH(false, true); else PQ_LAUNCH(false, false); }
#undef PQ_LAUNCH
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_predict_quantize(const float *ri, const float *tm, const uint8_t *seg, const float *model,
                                     const float *label_acc, const float *residual_in, float acc, int B, int P, int M,
                                     int16_t *q16, int32_t *q32, int32_t *nnz, float *pred, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS && seg && ws);
    ARG_TRY((residual_in && !pred) || (ri && tm && model));   // residual handed in, no prediction wanted: seg only
    ARG_TRY(q16 || q32);
    hipStream_t st = (hipStream_t)stream;
    // Self-contained entry: the tile offsets are rebuilt from this segmentation (histogram + scan with
    // no model output); the model rows are the caller's (point or plane models).
    const int KP = kpad(M), T = ntiles(P);
    WsLayout L = ws_layout(ws, B, P, M);
    HIP_TRY(hipMemsetAsync(L.sums, 0, (size_t)((char *)L.hist - (char *)L.sums), st));
    int rc;
    if ((rc = launch_hist_scan(ri, seg, nullptr, B, P, M, L, nullptr, nullptr, nnz, st))) return rc;
    return launch_predict_quantize(ri, tm, seg, model, acc, label_acc, residual_in, B, P, M, q16, q32, pred, ws, st);
}

// ================================================================================================
// f1 / f3  contour codec, decoder body, a3 as a stand-alone entry   (kernels: codec_kernels.h)
// ================================================================================================
#include "codec_kernels.h"

extern "C" size_t rpcc_codec_workspace_bytes(int B, int P, int M) {
    if (B <= 0 || P <= 0 || M <= 0) return 0;
    return ws_layout(nullptr, B, P, M).bytes + 256 + (size_t)B * ntiles(P) * 4;
}

extern "C" int rpcc_contour_encode(const uint8_t *seg, int B, int H, int W, uint8_t *contour_bits, uint16_t *idx_sequence,
                                   int32_t *nseq, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && seg && contour_bits && idx_sequence && nseq && ws);
    hipStream_t st = (hipStream_t)stream;
    const int P = H * W, T = ntiles(P);
    uint32_t *tile_cnt = reinterpret_cast<uint32_t *>(ws);
    contour_count_kernel<uint8_t><<<dim3(T, B), 256, 0, st>>>(seg, P, W, T, tile_cnt);
    tile_scan_kernel<<<B, 256, 0, st>>>(tile_cnt, T, nseq);
    contour_write_kernel<uint8_t><<<dim3(T, B), 256, 0, st>>>(seg, P, W, T, tile_cnt, contour_bits, idx_sequence);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_contour_decode(const uint8_t *contour_bits, const uint16_t *idx_sequence, int B, int H, int W,
                                   uint8_t *seg, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && seg && contour_bits && idx_sequence && ws);
    hipStream_t st = (hipStream_t)stream;
    const int P = H * W, T = ntiles(P);
    uint32_t *tile_cnt = reinterpret_cast<uint32_t *>(ws);
    contour_bits_count_kernel<<<dim3(T, B), 256, 0, st>>>(contour_bits, P, T, tile_cnt);
    tile_scan_kernel<<<B, 256, 0, st>>>(tile_cnt, T, nullptr);
    recover_map_kernel<uint8_t><<<dim3(T, B), 256, 0, st>>>(contour_bits, idx_sequence, P, T, tile_cnt, seg);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_decode(const uint8_t *seg, const int16_t *q16, const float *model, const float *tm,
                           const double *level_acc, int levels, const uint8_t *salience, int B, int P, int M,
                           float *ri_rec, float *pc_rec, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS && seg && q16 && model && tm && level_acc && ri_rec && ws);
    ARG_TRY(levels >= 0 && levels <= 8 && (levels == 0 || salience != nullptr));
    hipStream_t st = (hipStream_t)stream;
    const int KP = kpad(M), T = ntiles(P);
    WsLayout L = ws_layout(ws, B, P, M);
    HIP_TRY(hipMemsetAsync(L.sums, 0, (size_t)((char *)L.hist - (char *)L.sums), st));
    int rc;
    if ((rc = launch_hist_scan(nullptr, seg, nullptr, B, P, M, L, nullptr, nullptr, nullptr, st))) return rc;
    DecodeSteps steps;
    steps.levels = levels;
    for (int i = 0; i < 8; i++) steps.acc[i] = i < (levels ? levels : 1) ? level_acc[i] : 0.0;
    const size_t sh = (size_t)KP * 4 * 4 + (size_t)16 * (KP + 1) * 4 + (size_t)KP * 4;
    decode_kernel<<<dim3(T, B), 256, sh, st>>>(seg, q16, model, tm, L.hist, salience, steps, P, M, KP, T, ri_rec, pc_rec);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_pack_payload(const int16_t *q16, const int32_t *nnz, int B, int P, int16_t *packed, int64_t capacity,
                                 int64_t *total, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && q16 && nnz && packed && capacity >= 0);
    pack_payload_kernel<<<dim3((P + PACK_EPW - 1) / PACK_EPW, B), 256, 0, (hipStream_t)stream>>>(q16, nnz, P, capacity, packed,
                                                                                              total);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_backproject(const float *ri, const float *tm, int B, int P, float *pc, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && ri && tm && pc);
    backproject_kernel<<<dim3((P + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(ri, tm, P, pc);
    LAUNCH_CHECK();
    return RPCC_OK;
}

// ================================================================================================
// a12 / a13  non-uniform framework: key points and salience levels   (kernels: feature_kernels.h)
// ================================================================================================
#include "feature_kernels.h"

// feat may be NULL (the fused entry only needs the key-point map)
template <class L = uint8_t>
static int launch_features(const float *ri, const L *seg, int B, int H, int W, int feature_region, int segments,
                           int sharp_num, int less_sharp_num, int flat_num, float *feat, uint8_t *key_point_map,
                           hipStream_t st, int32_t *kpn = nullptr, int K = 0) {
    ARG_TRY(feature_region >= 1 && feature_region <= 16 && segments >= 1 && W < 65536);
    const bool rowmode = (W - 2 * feature_region) / segments <= 16 * FEAT_RQ && segments <= FEAT_ROW_SEGS && flat_num - 1 <= FEAT_ROW_FLAT;
    const bool compact = rowmode && !feat;  // no curvature image: the fused entry, whose ranges come from the projection (>= 0)
    const size_t sh = compact ? (size_t)2 * W * 4 + (size_t)W * 2 + (size_t)W + 16 : (size_t)3 * W * 4 + (size_t)W * 2 + (size_t)W * 2 + 16;
    ARG_TRY(sh <= 160 * 1024);
    ARG_TRY(W <= 64 * FEAT_GPW * (FEAT_THREADS / 64));                        // and so does a wavefront's part of the row
    FeatParams fp = {feature_region, segments, sharp_num, less_sharp_num, flat_num};
    const int need = ((W - 2 * feature_region) / segments + 63) / 64;  // keys per lane
#define FEAT_LAUNCH_G(Q_, G_, F_)                                                                                         \
    do {                                                                                                                  \
        HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&features_kernel<Q_, G_, F_, L>), (int)sh));                   \
        features_kernel<Q_, G_, F_, L><<<dim3(H, B), FEAT_THREADS, sh, st>>>(ri, seg, H, W, fp, feat, key_point_map, kpn, K); \
    } while (0)
#define FEAT_LAUNCH(Q_, F_)                                                \
    do {                                                                   \
        if (W <= 64 * 8 * (FEAT_THREADS / 64)) FEAT_LAUNCH_G(Q_, 8, F_);   \
        else FEAT_LAUNCH_G(Q_, FEAT_GPW, F_);                              \
    } while (0)
    if (compact) FEAT_LAUNCH(FEAT_ROWMODE, false);
    else if (rowmode) FEAT_LAUNCH(FEAT_ROWMODE, true);  // a chunk fits a 16-lane row
    else if (need <= 2) FEAT_LAUNCH(2, true);
    else if (need <= 4) FEAT_LAUNCH(4, true);
    else if (need <= 6) FEAT_LAUNCH(6, true);
    else if (need <= 8) FEAT_LAUNCH(8, true);
    else FEAT_LAUNCH(0, true);  // long chunks: keys stay in LDS
#undef FEAT_LAUNCH
#undef FEAT_LAUNCH_G
    LAUNCH_CHECK();
    return RPCC_OK;
}
extern "C" int rpcc_extract_features(const float *ri, const uint8_t *seg, int B, int H, int W, int feature_region,
                                     int segments, int sharp_num, int less_sharp_num, int flat_num, float *feat,
                                     uint8_t *key_point_map, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && ri && seg && feat && key_point_map);
    return launch_features(ri, seg, B, H, W, feature_region, segments, sharp_num, less_sharp_num, flat_num, feat,
                           key_point_map, (hipStream_t)stream);
}

extern "C" int rpcc_salience(const uint8_t *seg, const uint8_t *key_point_map, const int32_t *level_kp_num,
                             const float *level_acc, int levels, int ground_level, int B, int P, int M,
                             uint8_t *salience, float *label_acc, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS && seg && key_point_map && salience && label_acc);
    ARG_TRY(level_kp_num && level_acc && levels >= 1 && levels <= 8 && ground_level >= 0 && ground_level < levels);
    SalienceParams sp;
    for (int i = 0; i < 8; i++) { sp.level_kp_num[i] = i < levels ? level_kp_num[i] : 0; sp.level_acc[i] = i < levels ? level_acc[i] : 0.f; }
    sp.levels = levels;
    sp.ground_level = ground_level;
    salience_kernel<uint8_t, 256><<<B, SAL_THREADS, 0, (hipStream_t)stream>>>(seg, key_point_map, P, M, sp, salience, label_acc);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_intra_predict(const uint8_t *seg, const float *model, const float *tm, int B, int P, int M, float *pred,
                                  void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && seg && model && tm && pred);
    intra_predict_kernel<uint8_t><<<dim3((P + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(seg, model, tm, P, M + 2, pred);
    LAUNCH_CHECK();
    return RPCC_OK;
}

// ================================================================================================
// a9  plane model   (kernels: plane_kernels.h)
// ================================================================================================
#include "plane_kernels.h"

extern "C" size_t rpcc_plane_workspace_bytes(int B, int P, int M) {
    if (B <= 0 || P <= 0 || M <= 0) return 0;
    // model part | label-ordered pixel list u32 [B,P] | the same list as points (x, y, z, r) float4 [B,P]
    return ws_layout(nullptr, B, P, M).bytes + 256 + (((size_t)B * P * 4 + 255) & ~(size_t)255) + (size_t)B * P * 16;
}

// hist / scan for a segmentation without the point sums: tile offsets (ordered scatter), counts, nnz
static int launch_label_scan(const uint8_t *seg, int B, int P, int M, int32_t *counts, int32_t *nnz, void *ws, hipStream_t st,
                             bool cleared) {
    const int KP = kpad(M), T = ntiles(P);
    WsLayout L = ws_layout(ws, B, P, M);
    if (!cleared) HIP_TRY(hipMemsetAsync(L.sums, 0, (size_t)((char *)L.hist - (char *)L.sums), st));
    return launch_hist_scan(nullptr, seg, nullptr, B, P, M, L, nullptr, counts, nnz, st);
}
// plane rows from a segmentation whose tile offsets (launch_label_scan) are in ws; extra = order | pts4 scratch.  Two launches: the
// label-ordered lists, then the fits.
static inline float4 *plane_pts4(void *extra, int B, int P) {
    return reinterpret_cast<float4 *>(reinterpret_cast<char *>(extra) + (((size_t)B * P * 4 + 255) & ~(size_t)255));
}
static int launch_label_order(const float *ri, const float *tm, const uint8_t *seg, int B, int P, int M, void *ws, void *extra, hipStream_t st) {
    const int KP = kpad(M), T = ntiles(P);
    WsLayout L = ws_layout(ws, B, P, M);
    // (the quantiser's round-3 layout -- four consecutive pixels per lane -- was tried here as well: 121 us against 97 us, because a
    // lane's four 16-byte point stores then lie 64 bytes apart from the next lane's; this kernel is bound by its 255 MB of stores)
    label_order_kernel<uint8_t><<<dim3(T, B), 256, (size_t)16 * (KP + 1) * 4 + (size_t)KP * 4, st>>>(seg, L.hist, P, M, KP, T, reinterpret_cast<uint32_t *>(extra), ri, tm,
                                                                                                     plane_pts4(extra, B, P));
    LAUNCH_CHECK();
    return RPCC_OK;
}
static PlaneGroupArgs plane_group_args(const float *tm, const double *ground, int B, int P, int M, double cos_cut, uint32_t seed,
                                       const int64_t *frame_ids, float *model, const int32_t *counts, void *ws, void *extra, const double *inject) {
    WsLayout L = ws_layout(ws, B, P, M);
    PlaneGroupArgs a;
    a.tm = tm; a.order_all = reinterpret_cast<const uint32_t *>(extra); a.pts_all = plane_pts4(extra, B, P); a.hist = L.hist; a.counts = counts;
    a.ground = ground; a.B = B; a.P = P; a.T = ntiles(P); a.model = model;
    a.pp.cos_cut = cos_cut; a.pp.thr = 0.1f; a.pp.min_points = 30; a.pp.iters = 10; a.pp.seed = seed; a.pp.frame_ids = frame_ids; a.pp.inject = inject;
    return a;
}
static int launch_plane_fits(const float *tm, const double *ground, int B, int P, int M, double cos_cut, uint32_t seed, const int64_t *frame_ids,
                             float *model, const int32_t *counts, void *ws, void *extra, hipStream_t st, const double *inject = nullptr) {
    const int K = M + 2;
    const PlaneGroupArgs a = plane_group_args(tm, ground, B, P, M, cos_cut, seed, frame_ids, model, counts, ws, extra, inject);
    // B x (K-2) workgroups that exist for the labels above PL_BIG points (they start first: the long ones are the tail of
    // the launch), then one wavefront per label for the rest
    const int wpg = PL_THREADS / 64, groups = (K + wpg - 1) / wpg;
    plane_model_kernel<10><<<B * (K - 2) + B * groups, PL_THREADS, 0, st>>>(a.tm, a.order_all, a.pts_all, a.hist, a.counts, a.ground, B, P, M, kpad(M), a.T,
                                                                            a.pp, PL_BIG, a.model);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_plane_model(const float *ri, const float *tm, const uint8_t *seg, const double *ground, int B, int P,
                                int M, double cos_cut, uint32_t seed, const int64_t *frame_ids, const double *inject_planes,
                                float *model, int32_t *counts, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS && ri && tm && seg && model && counts && ws);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = launch_label_scan(seg, B, P, M, counts, nullptr, ws, st, false))) return rc;
    void *extra = reinterpret_cast<char *>(ws) + ws_layout(nullptr, B, P, M).bytes + 256;
    if ((rc = launch_label_order(ri, tm, seg, B, P, M, ws, extra, st))) return rc;
    return launch_plane_fits(tm, ground, B, P, M, cos_cut, seed, frame_ids, model, counts, ws, extra, st, inject_planes);
}

// ================================================================================================
// fused batch entry (uniform framework, FPS segmentation, point model): a2 .. a11
// ================================================================================================
// Everything a call derives from its arguments for ONE geometry group (one io, one rpcc_geom): the carve-up of the workspace and the
// choices that follow from the flags.  rpcc_compress_batch has one plan, rpcc_compress_batch_mixed one per group.
struct BatchPlan {
    const rpcc_batch_io *io;
    int Bs, M, P;
    int64_t npts;
    rpcc_geom g;
    double ground_threshold;
    float acc;
    char *ws;
    WsLayout L;
    bool fit_ground, brute, tiled;
    char *proj_scratch, *extra;
    size_t proj_bytes;
    float *temp, *rays_soa, *tiletab, *label_acc;
    int32_t *zcnt, *epoch, *kpn;
    BatchInit bi;
};
static BatchPlan plan_batch(const rpcc_batch_io *io, int Bs, int64_t npts, rpcc_geom g, int M, double ground_threshold, float acc, char *ws) {
    BatchPlan p;
    p.io = io; p.Bs = Bs; p.M = M; p.P = g.H * g.W; p.npts = npts; p.g = g; p.ground_threshold = ground_threshold; p.acc = acc; p.ws = ws;
    const int P = p.P;
    p.fit_ground = io->ground_seed >= 0;  // >= 0: fit the ground plane here (seed + frame identity)
    p.L = ws_layout(ws, Bs, P, M);
    p.proj_scratch = ws + p.L.bytes + 256;
    p.proj_bytes = (project_scratch_bytes(npts, Bs, P) + 255) & ~(size_t)255;
    p.temp = reinterpret_cast<float *>(p.proj_scratch + p.proj_bytes);
    p.rays_soa = p.temp + (size_t)Bs * P;
    p.tiletab = p.rays_soa + (size_t)3 * P + 64;
    p.zcnt = p.fit_ground ? reinterpret_cast<int32_t *>(p.tiletab) : nullptr;  // the tile table is written later
    // The first kernel of the batch (the pixel kernel) also writes the planar ray table (band kernel's z), initialises the info
    // counters of the ground mask and clears the RANSAC candidate counts and the label sums; the projection's per-frame
    // flags are marked with an epoch kept in the workspace (BatchInit), so the batch has no initialisation launch.
    p.epoch = reinterpret_cast<int32_t *>(ws + p.L.bytes);   // the 256 bytes between the model part and the projection scratch
    p.bi.tm = io->tm; p.bi.soa = p.rays_soa; p.bi.P = P; p.bi.info = io->info; p.bi.B = Bs; p.bi.on = 1;
    p.bi.z0 = {reinterpret_cast<uint32_t *>(p.zcnt), p.zcnt ? (int)rs_count_words(Bs) : 0};
    p.bi.z1 = {reinterpret_cast<uint32_t *>(p.L.sums), (int)(((char *)p.L.hist - (char *)p.L.sums) / 4)};
    // model rows + the tile offsets of the ordered scatter (built once, used by the plane list and by the quantiser)
    p.extra = reinterpret_cast<char *>(p.tiletab) + (((size_t)Bs * FPS_TAB_ROWS * ((P + 31) / 32 + 4096) * 4 + 255) & ~(size_t)255) + 256;
    const size_t ksz = (((size_t)Bs * (M + 2) * 4 + 255) & ~(size_t)255);
    p.label_acc = io->nonuniform ? reinterpret_cast<float *>(p.extra + plane_extra_bytes(Bs, P, M) - 256 - ksz) : nullptr;
    p.kpn = io->nonuniform ? reinterpret_cast<int32_t *>(p.extra + plane_extra_bytes(Bs, P, M) - 256 - 2 * ksz) : nullptr;
    p.bi.z2 = {reinterpret_cast<uint32_t *>(p.kpn), p.kpn ? Bs * (M + 2) : 0};  // key points per label
    p.brute = (io->flags & (RPCC_FPS_BRUTEFORCE | RPCC_FPS_MODE_BITS)) != 0;   // a mode flag selects the reference kernel too
    p.tiled = !p.brute && fps_tiling_range(g.H, g.W).T <= FPS_TILED_MAX_TILES;
    return p;
}
// The launches of a batch in order, as stages: the three marked (*) are the kernels with one workgroup per frame or per label, which
// rpcc_compress_batch_mixed runs as one launch over all groups (the others it runs group after group).
enum { ST_PROJECT, ST_GROUND /* (*) */, ST_MASK, ST_FPS /* (*) */, ST_ASSIGN_LABELS, ST_PLANES /* (*) */, ST_QUANTISE, ST_COUNT };
static int run_stage(const BatchPlan &p, int stage, hipStream_t st) {
    const rpcc_batch_io *io = p.io;
    const int Bs = p.Bs, M = p.M, P = p.P;
    int rc;
    switch (stage) {
    case ST_PROJECT:
        return launch_project(io->xyz, io->offsets, p.npts, 0, Bs, p.g, io->ri, p.proj_scratch, p.proj_bytes, st,
                              p.rays_soa + 2 * (int64_t)P, p.zcnt, &p.bi, p.epoch, point_floats(io->point_stride_bytes), order_mode_of(io->flags), io->tm);
    case ST_GROUND:
        if (!p.fit_ground) return RPCC_OK;
        return launch_ground_ransac(io->ri, io->tm, Bs, P, (uint32_t)io->ground_seed, false, io->ground, nullptr, st, p.zcnt, io->frame_ids);
    case ST_MASK:
        return launch_ground_mask(io->ri, io->tm, io->ground, p.ground_threshold, Bs, p.g.H, p.g.W, p.temp, io->info,
                                  p.tiled ? p.tiletab : nullptr, st, false, true);
    case ST_FPS:
        return launch_fps_range(io->ri, io->tm, p.temp, io->info, Bs, p.g.H, p.g.W, M, io->cen_pix, io->centers, io->flags, false,
                                p.tiled ? p.tiletab : nullptr, io->timer, st, FPS_SOA ? p.rays_soa : nullptr);
    case ST_ASSIGN_LABELS:
        // (the FPS state is the un-fused minimum the assignment's bound needs; a CUDA-binary mode contracts it)
        if ((rc = launch_assign(io->ri, io->tm, io->ground, io->centers, Bs, p.g.H, p.g.W, M, io->seg, st,
                                (io->flags & RPCC_FPS_MODE_BITS) ? nullptr : p.temp))) return rc;
        if (io->model_method == 0) return launch_point_model(io->ri, io->seg, io->ground, Bs, P, M, io->model, io->counts, io->nnz, p.ws, st, true);
        if ((rc = launch_label_scan(io->seg, Bs, P, M, io->counts, io->nnz, p.ws, st, true))) return rc;
        return launch_label_order(io->ri, io->tm, io->seg, Bs, P, M, p.ws, p.extra, st);
    case ST_PLANES:
        if (io->model_method == 0) return RPCC_OK;
        return launch_plane_fits(io->tm, io->ground, Bs, P, M, io->plane_cos_cut, (uint32_t)io->plane_seed, io->frame_ids, io->model,
                                 io->counts, p.ws, p.extra, st);
    case ST_QUANTISE:
        if (io->nonuniform) {   // key points -> salience level and quantisation step per label
            const rpcc_nonuniform_cfg *nu = io->nonuniform;
            if ((rc = launch_features(io->ri, io->seg, Bs, p.g.H, p.g.W, nu->feature_region, nu->segments, nu->sharp_num, nu->less_sharp_num,
                                      nu->flat_num, nullptr, io->key_point_map, st, p.kpn, M + 2)))
                return rc;
            SalienceParams sp;
            for (int i = 0; i < 8; i++) { sp.level_kp_num[i] = i < nu->levels ? nu->level_kp_num[i] : 0; sp.level_acc[i] = i < nu->levels ? nu->level_acc[i] : 0.f; }
            sp.levels = nu->levels;
            sp.ground_level = nu->ground_level;
            // levels from the per-label totals (pixels: the scan's counts; key points: counted by the key-point kernel)
            salience_levels_kernel<<<Bs, 256, 0, st>>>(io->counts, p.kpn, M, sp, io->salience, p.label_acc);
            LAUNCH_CHECK();
        }
        return launch_predict_quantize(io->ri, io->tm, io->seg, io->model, p.acc, p.label_acc, nullptr, Bs, P, M, io->q16,
                                       nullptr, nullptr, p.ws, st, p.epoch);
    }
    return RPCC_ERR_ARG;
}

static int check_batch_io(const rpcc_batch_io *io, int B, rpcc_geom g, int M, const void *ws) {
    ARG_TRY(io != nullptr && ws != nullptr && B > 0 && B <= RPCC_MAX_BATCH && M > 0 && M <= RPCC_MAX_CLUSTERS && g.H > 1 && g.W > 0);
    ARG_TRY(io->offsets && io->tm && io->ground && io->ri && io->seg && io->cen_pix && io->centers && io->model &&
            io->counts && io->q16 && io->nnz && io->info);
    const int P = g.H * g.W;
    // only the brute-force FPS kernel (16-byte loads at frame bases) needs P % 4 == 0; the tile-pruned one does not
    ARG_TRY(P % 4 == 0 || (io->flags & RPCC_FPS_MODE_BITS) || (!(io->flags & RPCC_FPS_BRUTEFORCE) && fps_tiling_range(g.H, g.W).T <= FPS_TILED_MAX_TILES));
    ARG_TRY(io->model_method == 0 || io->model_method == 1);
    ARG_TRY(point_floats(io->point_stride_bytes) > 0 && (io->point_stride_bytes != 16 || (reinterpret_cast<uintptr_t>(io->xyz) & 15u) == 0));
    if (io->nonuniform) {
        const rpcc_nonuniform_cfg *nu = io->nonuniform;
        ARG_TRY(io->salience && io->key_point_map && nu->levels >= 1 && nu->levels <= 8 && nu->ground_level >= 0 && nu->ground_level < nu->levels);
    }
    return RPCC_OK;
}

extern "C" int rpcc_compress_batch(const rpcc_batch_io *io, int B, rpcc_geom g, int M, double ground_threshold,
                                   float acc, void *ws, void *stream) {
    int rc;
    if ((rc = check_batch_io(io, B, g, M, ws))) return rc;
    const BatchPlan p = plan_batch(io, B, io->total, g, M, ground_threshold, acc, reinterpret_cast<char *>(ws));
    for (int stage = 0; stage < ST_COUNT; stage++)
        if ((rc = run_stage(p, stage, (hipStream_t)stream))) return rc;
    return RPCC_OK;
}

// A subset of the batch's stages (bit i of stage_mask = stage i of the enum above: projection, ground fit, mask, FPS, assignment + label
// histograms + scan (+ label order), plane fits, key points + quantiser), in order, on `stream`.  For callers that schedule the stages of several batches
// themselves (tools_dev/tick_bench.py: the one-workgroup-per-frame kernels of different batches back to back on one stream, the pixel-parallel ones on
// others); running all bits is rpcc_compress_batch.
extern "C" int rpcc_compress_batch_stages(const rpcc_batch_io *io, int B, rpcc_geom g, int M, double ground_threshold, float acc, void *ws,
                                          int stage_mask, void *stream) {
    int rc;
    if ((rc = check_batch_io(io, B, g, M, ws))) return rc;
    const BatchPlan p = plan_batch(io, B, io->total, g, M, ground_threshold, acc, reinterpret_cast<char *>(ws));
    for (int stage = 0; stage < ST_COUNT; stage++)
        if (((stage_mask >> stage) & 1) && (rc = run_stage(p, stage, (hipStream_t)stream))) return rc;
    return RPCC_OK;
}

// ---- several geometry groups in one call ----------------------------------------------------------------------------------------
// (*) stages as one launch over the groups that take the common kernel; a group that does not (injected ground, brute-force FPS or a
// CUDA-binary mode, an image too large for the register-table FPS, the point model) runs that stage by itself.
static int mixed_ground(const BatchPlan *pl, int G, hipStream_t st) {
    RansacMulti m;
    m.n = 0; m.first[0] = 0;
    for (int i = 0; i < G; i++) {
        if (!pl[i].fit_ground) continue;
        const rpcc_batch_io *io = pl[i].io;
        m.a[m.n] = {io->ri, io->tm, pl[i].P, (uint32_t)io->ground_seed, io->ground, pl[i].zcnt, io->frame_ids,
                    pl[i].zcnt ? rs_wc_of(pl[i].zcnt, pl[i].Bs, pl[i].P) : nullptr, 100};
        m.first[m.n + 1] = m.first[m.n] + pl[i].Bs;
        m.n++;
    }
    if (m.n == 0) return RPCC_OK;
    return launch_ground_ransac_multi(m, st);
}
static int mixed_fps(const BatchPlan *pl, int G, hipStream_t st) {
    int rc, total = 0;
    bool common[RPCC_MAX_GROUPS], edge[RPCC_MAX_GROUPS];
    for (int i = 0; i < G; i++) {   // the planar register-table kernel (launch_fps_tiled's first branch)
        const BatchPlan &p = pl[i];
        const bool vec = (p.g.W % 4 == 0) && aligned16(p.io->ri) && aligned16(p.temp) && aligned16(p.io->tm);
        edge[i] = !vec;
        common[i] = p.tiled && p.P < (1 << 22) && FPS_SOA && p.io->timer == nullptr;
        if (common[i]) total += p.Bs;
    }
    const int tt = total <= 128 ? FPS_TT_SMALL : FPS_TT_BATCH;
    for (int i = 0; i < G; i++) common[i] = common[i] && fps_tiling_range(pl[i].g.H, pl[i].g.W).T <= tt;
    FpsMulti m;
    m.n = 0; m.first[0] = 0;
    for (int i = 0; i < G; i++) {
        if (!common[i]) continue;
        const BatchPlan &p = pl[i];
        m.a[m.n] = {p.io->ri, p.io->tm, p.temp, p.io->info, fps_tiling_range(p.g.H, p.g.W), p.io->cen_pix, p.io->centers, p.tiletab, p.rays_soa};
        m.edge[m.n] = edge[i] ? 1 : 0;
        m.first[m.n + 1] = m.first[m.n] + p.Bs;
        m.n++;
    }
    if (m.n > 0) {
        if (tt == FPS_TT_SMALL) fps_regtab_planar_multi_kernel<FPS_TT_SMALL><<<m.first[m.n], FPS_TT_SMALL, 0, st>>>(m, pl[0].M, 0);
        else                    fps_regtab_planar_multi_kernel<FPS_TT_BATCH><<<m.first[m.n], FPS_TT_BATCH, 0, st>>>(m, pl[0].M, 0);
        LAUNCH_CHECK();
    }
    for (int i = 0; i < G; i++)
        if (!common[i] && (rc = run_stage(pl[i], ST_FPS, st))) return rc;
    return RPCC_OK;
}
static int mixed_planes(const BatchPlan *pl, int G, hipStream_t st) {
    PlaneMulti m;
    m.n = 0; m.first[0] = 0;
    const int M = pl[0].M, K = M + 2, wpg = PL_THREADS / 64, groups = (K + wpg - 1) / wpg;
    for (int i = 0; i < G; i++) {
        const BatchPlan &p = pl[i];
        if (p.io->model_method == 0) continue;
        m.a[m.n] = plane_group_args(p.io->tm, p.io->ground, p.Bs, p.P, M, p.io->plane_cos_cut, (uint32_t)p.io->plane_seed, p.io->frame_ids,
                                    p.io->model, p.io->counts, p.ws, p.extra, nullptr);
        m.first[m.n + 1] = m.first[m.n] + p.Bs * (K - 2) + p.Bs * groups;
        m.n++;
    }
    if (m.n == 0) return RPCC_OK;
    plane_model_multi_kernel<10><<<m.first[m.n], PL_THREADS, 0, st>>>(m, M, kpad(M), PL_BIG);
    LAUNCH_CHECK();
    return RPCC_OK;
}

// The pixel-parallel stages as ONE launch over the groups as well (round 5): a workgroup finds its group, frame and tile through the table in the
// kernel arguments (MultiArgs, rpcc_device.h); every group keeps its own buffers, image size and access variant.  Per group the results are
// what its own launch gives (same kernel bodies).
template <class A>
static void multi_add(MultiArgs<A> &m, const A &a, int frames, int per_frame) {
    m.a[m.n] = a; m.per[m.n] = per_frame; m.first[m.n + 1] = m.first[m.n] + frames * per_frame; m.n++;
}
static int mixed_assign(const BatchPlan *pl, int G, hipStream_t st) {
    int rc;
    for (int i = 0; i < G; i++) {
        const rpcc_batch_io *io = pl[i].io;
        if ((rc = launch_assign(io->ri, io->tm, io->ground, io->centers, pl[i].Bs, pl[i].g.H, pl[i].g.W, pl[i].M, io->seg, st,
                                (io->flags & RPCC_FPS_MODE_BITS) ? nullptr : pl[i].temp))) return rc;
    }
    return RPCC_OK;
}
// label histograms + scan (+ the point model's rows, or the plane model's label-ordered lists)
static int mixed_labels(const BatchPlan *pl, int G, hipStream_t st) {
    const int M = pl[0].M, KP = kpad(M);
    MultiArgs<HistGroup> mh; MultiArgs<ScanArgs> ms; MultiArgs<OrderGroup> mo;
    mh.n = ms.n = mo.n = 0; mh.first[0] = ms.first[0] = mo.first[0] = 0;
    for (int i = 0; i < G; i++) {
        const BatchPlan &p = pl[i];
        const rpcc_batch_io *io = p.io;
        const int T = ntiles(p.P);
        const bool point = io->model_method == 0;
        const float *ri = point ? io->ri : nullptr;   // (the plane model needs no range sums)
        const bool vec = (p.P & 3) == 0 && ((uintptr_t)io->seg & 3u) == 0 && (ri == nullptr || ((uintptr_t)ri & 15u) == 0);
        multi_add(mh, HistGroup{ri, io->seg, p.P, T, vec ? 1 : 0, p.L.sums, p.L.flags, p.L.hist}, p.Bs, T);
        multi_add(ms, ScanArgs{ri, io->seg, point ? io->ground : nullptr, p.P, M, KP, T, scan_kp2(M), p.L.sums, p.L.flags, p.L.hist,
                               point ? io->model : nullptr, io->counts, io->nnz}, p.Bs, 1);
        if (!point) multi_add(mo, OrderGroup{io->seg, p.L.hist, p.P, T, reinterpret_cast<uint32_t *>(p.extra), io->ri, io->tm, plane_pts4(p.extra, p.Bs, p.P)}, p.Bs, T);
    }
    model_hist_multi_kernel<<<mh.first[mh.n], 256, (size_t)KP * 12, st>>>(mh, KP);
    LAUNCH_CHECK();
    model_scan_multi_kernel<<<ms.first[ms.n], SCAN_THREADS, 0, st>>>(ms);
    LAUNCH_CHECK();
    if (mo.n > 0) {
        label_order_multi_kernel<<<mo.first[mo.n], 256, (size_t)16 * (KP + 1) * 4 + (size_t)KP * 4, st>>>(mo, M, KP);
        LAUNCH_CHECK();
    }
    return RPCC_OK;
}
// key points per group (the kernel's shape follows the image width), then salience levels and the quantiser once over the groups
static int mixed_quantise(const BatchPlan *pl, int G, hipStream_t st) {
    int rc;
    const int M = pl[0].M, KP = kpad(M);
    MultiArgs<SalienceGroup> msal; MultiArgs<QuantGroup> mq;
    msal.n = mq.n = 0; msal.first[0] = mq.first[0] = 0;
    for (int i = 0; i < G; i++) {
        const BatchPlan &p = pl[i];
        const rpcc_batch_io *io = p.io;
        if (io->nonuniform) {
            const rpcc_nonuniform_cfg *nu = io->nonuniform;
            if ((rc = launch_features(io->ri, io->seg, p.Bs, p.g.H, p.g.W, nu->feature_region, nu->segments, nu->sharp_num, nu->less_sharp_num,
                                      nu->flat_num, nullptr, io->key_point_map, st, p.kpn, M + 2)))
                return rc;
            SalienceGroup sg;
            sg.counts = io->counts; sg.kpn = p.kpn; sg.salience = io->salience; sg.label_acc = p.label_acc;
            for (int l = 0; l < 8; l++) { sg.sp.level_kp_num[l] = l < nu->levels ? nu->level_kp_num[l] : 0; sg.sp.level_acc[l] = l < nu->levels ? nu->level_acc[l] : 0.f; }
            sg.sp.levels = nu->levels; sg.sp.ground_level = nu->ground_level;
            multi_add(msal, sg, p.Bs, 1);
        }
        const int T = ntiles(p.P);
        const bool vec = (p.P & 3) == 0 && ((uintptr_t)io->seg & 3u) == 0 && aligned16(io->ri) && aligned16(io->tm);
        // (every group's kernel advances its own workspace's epoch: its frame 0 / tile 0 workgroup)
        multi_add(mq, QuantGroup{io->ri, io->tm, io->model, p.label_acc, io->seg, p.L.hist, p.P, T, vec ? 1 : 0, io->q16, p.epoch}, p.Bs, T);
    }
    if (msal.n > 0) {
        salience_levels_multi_kernel<<<msal.first[msal.n], 256, 0, st>>>(msal, M);
        LAUNCH_CHECK();
    }
    const size_t sh = (size_t)KP * 16 + (size_t)4 * KP * 4 + (size_t)KP * 4 + (size_t)KP * 4;
    predict_quantize_multi_kernel<<<mq.first[mq.n], 256, sh, st>>>(mq, pl[0].acc, M, KP);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_compress_batch_mixed(const rpcc_batch_io *ios, const int *Bs, const rpcc_geom *geoms, int G, int M,
                                         double ground_threshold, float acc, void *const *wss, void *stream) {
    ARG_TRY(ios != nullptr && Bs != nullptr && geoms != nullptr && wss != nullptr && G >= 1 && G <= RPCC_MAX_GROUPS);
    int rc;
    BatchPlan pl[RPCC_MAX_GROUPS];
    for (int i = 0; i < G; i++) {
        if ((rc = check_batch_io(&ios[i], Bs[i], geoms[i], M, wss[i]))) return rc;
        pl[i] = plan_batch(&ios[i], Bs[i], ios[i].total, geoms[i], M, ground_threshold, acc, reinterpret_cast<char *>(wss[i]));
    }
    hipStream_t st = (hipStream_t)stream;
    for (int stage = 0; stage < ST_COUNT; stage++) {
        if (stage == ST_GROUND) rc = mixed_ground(pl, G, st);
        else if (stage == ST_FPS) rc = mixed_fps(pl, G, st);
        else if (stage == ST_PLANES) rc = mixed_planes(pl, G, st);
        else if (stage == ST_ASSIGN_LABELS) { if (!(rc = mixed_assign(pl, G, st))) rc = mixed_labels(pl, G, st); }
        else if (stage == ST_QUANTISE) rc = mixed_quantise(pl, G, st);
        else
            for (int i = 0; i < G && !(rc = run_stage(pl[i], stage, st)); i++) {}
        if (rc) return rc;
    }
    return RPCC_OK;
}

// ================================================================================================
// cluster_num above RPCC_MAX_CLUSTERS: uint16 labels, the plain kernels of wide_kernels.h
// ================================================================================================
#include "wide_kernels.h"

// ---- 255 .. RPCC_MAX_CLUSTERS_MID clusters with the point model: the tuned assignment / histogram / quantiser kernels on uint16 labels ----------------
// The label tables of those kernels live in LDS (16 B per centre, 12 and 40 B per label): up to 1022 clusters they still fit, so the kernels are
// instantiated for uint16_t labels (assign_kernel<uint16_t, 16>: sixteen screening rounds of 64 centres) instead of going through the radix sort of
// wide_kernels.h (89 k frames/s at 300 clusters against 367 k at 100: a 4 x cliff at 254 -> 255).  The scan between histogram and quantiser is
// model_scan_kernel's job with a thread per label (K <= 1024): label totals, their exclusive prefix without label 1, the tiles' offsets, counts, nnz and the
// point model's rows (cpp_modules.cpp:471-518).
#define SCANW_THREADS 1024
template <class L>
__global__ __launch_bounds__(SCANW_THREADS) void model_scan_wide_kernel(const float *__restrict__ ri, const L *__restrict__ seg, const double *__restrict__ ground,
                                                                        int P, int M, int KP, int T, const int64_t *__restrict__ sums,
                                                                        const int32_t *__restrict__ flags, uint32_t *__restrict__ hist,
                                                                        float *__restrict__ model, int32_t *__restrict__ counts, int32_t *__restrict__ nnz) {
    __shared__ uint32_t wsum[SCANW_THREADS / 64];
    const int b = blockIdx.x, k = threadIdx.x, K = M + 2, lane = k & 63, wave = k >> 6;
    uint32_t *gh = hist + (int64_t)b * T * KP;
    const uint32_t kp4 = (uint32_t)KP * 4u, k4 = (uint32_t)min(k, KP - 1) * 4u;
    uint32_t total = 0u;
    for (int t0 = 0; t0 < T; t0 += SCAN_U) {
        uint32_t d[SCAN_U];
#pragma unroll
        for (int j = 0; j < SCAN_U; j++) d[j] = ld_at(gh, (uint32_t)min(t0 + j, T - 1) * kp4 + k4);   // unconditional (clamped) loads
#pragma unroll
        for (int j = 0; j < SCAN_U; j++) total += (t0 + j < T && k < K) ? d[j] : 0u;
    }
    const uint32_t v = (k < K && k != 1) ? total : 0u;   // label 1 = empty pixels: no residuals (cpp_modules.cpp:314)
    const uint32_t incl = dpp_scan_incl_u32(v);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t off = 0u;
    for (int w = 0; w < wave; w++) off += wsum[w];
    if (k == SCANW_THREADS - 1 && nnz) nnz[b] = (int32_t)(off + incl);
    if (k >= K) return;
    uint32_t run = off + incl - v;
    for (int t0 = 0; t0 < T; t0 += SCAN_U) {
        uint32_t d[SCAN_U];
#pragma unroll
        for (int j = 0; j < SCAN_U; j++) d[j] = ld_at(gh, (uint32_t)min(t0 + j, T - 1) * kp4 + k4);
#pragma unroll
        for (int j = 0; j < SCAN_U; j++)
            if (t0 + j < T) { st_at(gh, (uint32_t)(t0 + j) * kp4 + k4, run); run += d[j]; }
    }
    if (counts) counts[(int64_t)b * K + k] = (int32_t)total;
    if (model != nullptr) {
        float *row = model + ((int64_t)b * K + k) * 4;
        if (k == 0) {
            row[0] = (float)ground[4 * b]; row[1] = (float)ground[4 * b + 1]; row[2] = (float)ground[4 * b + 2]; row[3] = (float)ground[4 * b + 3];
        } else if (k == 1) {
            row[0] = row[1] = row[2] = row[3] = 0.0f;
        } else {
            double sm;
            if (flags[4 * b]) {   // sequential double accumulation in row-major order (cpp_modules.cpp:514): ranges outside the fixed-point window, rare
                sm = 0.0;
                for (int p = 0; p < P; p++)
                    if (seg[(int64_t)b * P + p] == (L)k) sm += (double)ri[(int64_t)b * P + p];
            } else {
                sm = (double)sums[(int64_t)b * KP + k] * (1.0 / 268435456.0);
            }
            row[0] = row[1] = row[2] = 0.0f;
            row[3] = total == 0 ? u2f(0xFFC00000u) : (float)(sm / (double)total);
        }
    }
}
// The whole batch for such a cluster count: the fused batch's own plan and stages up to the FPS (projection with its hand-over to the ground fit, planar
// ray table, the batch's initialisations inside the first kernel), then the label kernels on uint16_t.  ws: laid out as for rpcc_compress_batch
// (rpcc_wide_workspace_bytes covers rpcc_workspace_bytes_general for these counts).
static bool mid_clusters_ok(const rpcc_batch_io *io, rpcc_geom g, int M) {
    const int P = g.H * g.W;
    (void)io;
    return M <= RPCC_MAX_CLUSTERS_MID && P < (1 << 22) && fps_tiling_range(g.H, g.W).T <= FPS_TILED_MAX_TILES;
}
static int compress_batch_mid(const rpcc_batch_io *io, int B, rpcc_geom g, int M, double ground_threshold, float acc, void *ws, hipStream_t st) {
    const BatchPlan p = plan_batch(io, B, io->total, g, M, ground_threshold, acc, reinterpret_cast<char *>(ws));
    int rc;
    for (int stage = ST_PROJECT; stage <= ST_FPS; stage++)
        if ((rc = run_stage(p, stage, st))) return rc;
    const int P = p.P, K = M + 2, KP = kpad(M), T = ntiles(P);
    uint16_t *seg = reinterpret_cast<uint16_t *>(io->seg);
    const int ntile = ((g.H + ASSIGN_ROWS - 1) / ASSIGN_ROWS) * ((g.W + 31) / 32);
    const dim3 agrid((ntile + ASSIGN_WAVES * ASSIGN_TILES_PER_WAVE - 1) / (ASSIGN_WAVES * ASSIGN_TILES_PER_WAVE), B);
    if (M <= 510) assign_kernel<uint16_t, 8><<<agrid, 64 * ASSIGN_WAVES, (size_t)M * sizeof(float4), st>>>(io->ri, io->tm, io->ground, io->centers, g.H, g.W, M, seg, p.temp);
    else          assign_kernel<uint16_t, 16><<<agrid, 64 * ASSIGN_WAVES, (size_t)M * sizeof(float4), st>>>(io->ri, io->tm, io->ground, io->centers, g.H, g.W, M, seg, p.temp);
    LAUNCH_CHECK();
    // (sums / flags were cleared by the batch's first kernel: BatchInit)
    const bool vec = (P & 3) == 0 && ((uintptr_t)seg & 7u) == 0 && ((uintptr_t)io->ri & 15u) == 0 && ((uintptr_t)io->tm & 15u) == 0;
    const bool point = io->model_method == 0;
    const float *ri_sums = point ? io->ri : nullptr;      // (the plane model needs the label counts and offsets only)
    if (vec) model_hist_kernel<true, uint16_t><<<dim3(T, B), 256, (size_t)KP * 12, st>>>(ri_sums, seg, P, KP, T, p.L.sums, p.L.flags, p.L.hist);
    else     model_hist_kernel<false, uint16_t><<<dim3(T, B), 256, (size_t)KP * 12, st>>>(ri_sums, seg, P, KP, T, p.L.sums, p.L.flags, p.L.hist);
    model_scan_wide_kernel<uint16_t><<<B, SCANW_THREADS, 0, st>>>(ri_sums, seg, io->ground, P, M, KP, T, p.L.sums, p.L.flags, p.L.hist, point ? io->model : nullptr,
                                                                  io->counts, io->nnz);
    LAUNCH_CHECK();
    if (!point) {   // a9 on the label-ordered lists (label_order_kernel on uint16 labels; the fits never see a label)
        const size_t osh = (size_t)16 * (KP + 1) * 4 + (size_t)KP * 4;
        HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&label_order_kernel<uint16_t>), (int)osh));
        label_order_kernel<uint16_t><<<dim3(T, B), 256, osh, st>>>(seg, p.L.hist, P, M, KP, T, reinterpret_cast<uint32_t *>(p.extra), io->ri, io->tm, plane_pts4(p.extra, B, P));
        LAUNCH_CHECK();
        if ((rc = launch_plane_fits(io->tm, io->ground, B, P, M, io->plane_cos_cut, (uint32_t)io->plane_seed, io->frame_ids, io->model, io->counts, p.ws, p.extra, st))) return rc;
    }
    if (io->nonuniform) {
        const rpcc_nonuniform_cfg *nu = io->nonuniform;
        if ((rc = launch_features<uint16_t>(io->ri, seg, B, g.H, g.W, nu->feature_region, nu->segments, nu->sharp_num, nu->less_sharp_num, nu->flat_num,
                                            nullptr, io->key_point_map, st, p.kpn, K))) return rc;
        SalienceParams sp;
        for (int i = 0; i < 8; i++) { sp.level_kp_num[i] = i < nu->levels ? nu->level_kp_num[i] : 0; sp.level_acc[i] = i < nu->levels ? nu->level_acc[i] : 0.f; }
        sp.levels = nu->levels; sp.ground_level = nu->ground_level;
        wide_salience_levels_kernel<<<dim3((K + 255) / 256, B), 256, 0, st>>>(io->counts, p.kpn, K, sp, io->salience, p.label_acc);
        LAUNCH_CHECK();
    }
    const size_t sh = (size_t)KP * 40;
    // (the last kernel of the batch gives the next call's projection flags a new mark: p.epoch)
    if (vec) predict_quantize_kernel<false, true, uint16_t><<<dim3(T, B), 256, sh, st>>>(io->ri, io->tm, seg, io->model, p.L.hist, acc, p.label_acc, nullptr, P, M, KP, T, io->q16, nullptr, nullptr, p.epoch);
    else     predict_quantize_kernel<false, false, uint16_t><<<dim3(T, B), 256, sh, st>>>(io->ri, io->tm, seg, io->model, p.L.hist, acc, p.label_acc, nullptr, P, M, KP, T, io->q16, nullptr, nullptr, p.epoch);
    LAUNCH_CHECK();
    return RPCC_OK;
}

// ---- the stage entries on uint16 labels, 255 .. RPCC_MAX_CLUSTERS_MID clusters (the reference-side classes call the stages one by one:
// PointCloudSegment.segment / cluster_modeling('point') / intra_predict, QuantizationModule.quantize_residual with the uniform framework) ------------
static int hist_scan_u16(const float *ri, const uint16_t *seg, const double *ground, int B, int P, int M, const WsLayout &L, float *model,
                         int32_t *counts, int32_t *nnz, hipStream_t st) {
    const int KP = kpad(M), T = ntiles(P);
    const bool vec = (P & 3) == 0 && ((uintptr_t)seg & 7u) == 0 && (ri == nullptr || ((uintptr_t)ri & 15u) == 0);
    HIP_TRY(hipMemsetAsync(L.sums, 0, (size_t)((char *)L.hist - (char *)L.sums), st));
    if (vec) model_hist_kernel<true, uint16_t><<<dim3(T, B), 256, (size_t)KP * 12, st>>>(ri, seg, P, KP, T, L.sums, L.flags, L.hist);
    else     model_hist_kernel<false, uint16_t><<<dim3(T, B), 256, (size_t)KP * 12, st>>>(ri, seg, P, KP, T, L.sums, L.flags, L.hist);
    model_scan_wide_kernel<uint16_t><<<B, SCANW_THREADS, 0, st>>>(ri, seg, ground, P, M, KP, T, L.sums, L.flags, L.hist, model, counts, nnz);
    LAUNCH_CHECK();
    return RPCC_OK;
}
extern "C" int rpcc_assign_wide(const float *ri, const float *tm, const double *ground, const float *centers, int B, int H, int W, int M,
                                uint16_t *seg, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS_MID && ri && tm && ground && centers && seg);
    hipStream_t st = (hipStream_t)stream;
    const int ntile = ((H + ASSIGN_ROWS - 1) / ASSIGN_ROWS) * ((W + 31) / 32);
    const dim3 grid((ntile + ASSIGN_WAVES * ASSIGN_TILES_PER_WAVE - 1) / (ASSIGN_WAVES * ASSIGN_TILES_PER_WAVE), B);
    if (M <= 510) assign_kernel<uint16_t, 8><<<grid, 64 * ASSIGN_WAVES, (size_t)M * sizeof(float4), st>>>(ri, tm, ground, centers, H, W, M, seg, nullptr);
    else          assign_kernel<uint16_t, 16><<<grid, 64 * ASSIGN_WAVES, (size_t)M * sizeof(float4), st>>>(ri, tm, ground, centers, H, W, M, seg, nullptr);
    LAUNCH_CHECK();
    return RPCC_OK;
}
extern "C" int rpcc_point_model_wide(const float *ri, const uint16_t *seg, const double *ground, int B, int P, int M, float *model, int32_t *counts,
                                     void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS_MID && ri && seg && ground && model && ws);
    return hist_scan_u16(ri, seg, ground, B, P, M, ws_layout(ws, B, P, M), model, counts, nullptr, (hipStream_t)stream);
}
// rpcc_extract_features / rpcc_salience on uint16 labels
extern "C" int rpcc_extract_features_wide(const float *ri, const uint16_t *seg, int B, int H, int W, int feature_region, int segments, int sharp_num,
                                          int less_sharp_num, int flat_num, float *feat, uint8_t *key_point_map, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && ri && seg && feat && key_point_map);
    return launch_features<uint16_t>(ri, seg, B, H, W, feature_region, segments, sharp_num, less_sharp_num, flat_num, feat, key_point_map, (hipStream_t)stream);
}
extern "C" int rpcc_salience_wide(const uint16_t *seg, const uint8_t *key_point_map, const int32_t *level_kp_num, const float *level_acc, int levels,
                                  int ground_level, int B, int P, int M, uint8_t *salience, float *label_acc, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS_MID && seg && key_point_map && salience && label_acc);
    ARG_TRY(level_kp_num && level_acc && levels >= 1 && levels <= 8 && ground_level >= 0 && ground_level < levels);
    SalienceParams sp;
    for (int i = 0; i < 8; i++) { sp.level_kp_num[i] = i < levels ? level_kp_num[i] : 0; sp.level_acc[i] = i < levels ? level_acc[i] : 0.f; }
    sp.levels = levels;
    sp.ground_level = ground_level;
    salience_kernel<uint16_t, 1024><<<B, SAL_THREADS, 0, (hipStream_t)stream>>>(seg, key_point_map, P, M, sp, salience, label_acc);
    LAUNCH_CHECK();
    return RPCC_OK;
}
// rpcc_plane_model on uint16 labels (ws: rpcc_plane_workspace_bytes(B, P, M) bytes)
extern "C" int rpcc_plane_model_wide(const float *ri, const float *tm, const uint16_t *seg, const double *ground, int B, int P, int M, double cos_cut,
                                     uint32_t seed, const int64_t *frame_ids, const double *inject_planes, float *model, int32_t *counts, void *ws,
                                     void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS_MID && ri && tm && seg && model && counts && ws);
    hipStream_t st = (hipStream_t)stream;
    const WsLayout L = ws_layout(ws, B, P, M);
    const int KP = kpad(M), T = ntiles(P);
    int rc;
    if ((rc = hist_scan_u16(nullptr, seg, nullptr, B, P, M, L, nullptr, counts, nullptr, st))) return rc;
    void *extra = reinterpret_cast<char *>(ws) + L.bytes + 256;
    const size_t osh = (size_t)16 * (KP + 1) * 4 + (size_t)KP * 4;
    HIP_TRY(ensure_dyn_lds(reinterpret_cast<const void *>(&label_order_kernel<uint16_t>), (int)osh));
    label_order_kernel<uint16_t><<<dim3(T, B), 256, osh, st>>>(seg, L.hist, P, M, KP, T, reinterpret_cast<uint32_t *>(extra), ri, tm, plane_pts4(extra, B, P));
    LAUNCH_CHECK();
    return launch_plane_fits(tm, ground, B, P, M, cos_cut, seed, frame_ids, model, counts, ws, extra, st, inject_planes);
}
extern "C" int rpcc_intra_predict_wide(const uint16_t *seg, const float *model, const float *tm, int B, int P, int M, float *pred, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS_WIDE && seg && model && tm && pred);
    intra_predict_kernel<uint16_t><<<dim3((P + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(seg, model, tm, P, M + 2, pred);
    LAUNCH_CHECK();
    return RPCC_OK;
}
extern "C" int rpcc_predict_quantize_wide(const float *ri, const float *tm, const uint16_t *seg, const float *model, const float *label_acc,
                                          const float *residual_in, float acc, int B, int P, int M, int16_t *q16, int32_t *q32, int32_t *nnz,
                                          float *pred, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS_MID && seg && ws);
    ARG_TRY((residual_in && !pred) || (ri && tm && model));   // residual handed in, no prediction wanted: seg only
    ARG_TRY(q16 || q32);
    hipStream_t st = (hipStream_t)stream;
    const int KP = kpad(M), T = ntiles(P);
    const WsLayout L = ws_layout(ws, B, P, M);
    int rc;
    if ((rc = hist_scan_u16(ri, seg, nullptr, B, P, M, L, nullptr, nullptr, nnz, st))) return rc;
    const size_t sh = (size_t)KP * 40;
    const bool resid = residual_in && !pred;
    const bool vec = (P & 3) == 0 && ((uintptr_t)seg & 7u) == 0 && (resid || (aligned16(ri) && aligned16(tm))) && (!residual_in || aligned16(residual_in));
#define PQW_LAUNCH(R_, V_) predict_quantize_kernel<R_, V_, uint16_t><<<dim3(T, B), 256, sh, st>>>(ri, tm, seg, model, L.hist, acc, label_acc, residual_in, P, M, KP, T, q16, q32, pred, nullptr)
    if (resid) { if (vec) PQW_LAUNCH(true, true); else PQW_LAUNCH(true, false); }
    else       { if (vec) PQW_LAUNCH(false, true); else PQW_LAUNCH(false, false); }
#undef PQW_LAUNCH
    LAUNCH_CHECK();
    return RPCC_OK;
}

// carve-up of a wide workspace: [ keys, vals (in / out) u32 4 x [B,P] | pos i32 [B,P] | order u32 [B,P] | pts4 float4 [B,P] | sums u64 [B,K] |
//   base u32 [B,K] | kpn i32 [B,K] | label_acc f32 [B,K] | flags i32 [B,4] | cen4 float4 [B,M] | FPS temp f32 [B,P] | FPS tile table |
//   projection scratch | radix sort scratch ]
struct WideWs {
    uint32_t *keys_in, *vals_in, *keys_out, *vals_out, *order, *base;
    int32_t *pos, *kpn, *flags;
    float4 *pts4, *cen4;
    unsigned long long *sums;
    float *label_acc, *temp, *tiletab;
    char *proj;
    size_t proj_bytes;
    void *sort_tmp;
    size_t sort_bytes, bytes;
};
static size_t wide_sort_bytes(int64_t n, int end_bit) {
    size_t b = 0;
    uint32_t *z = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, b, z, z, z, z, (size_t)n, 0u, (unsigned)end_bit, (hipStream_t)0);
    return b;
}
static inline int wide_key_bits(int B) { int v = 16; while ((1 << (v - 16)) < B) v++; return v; }
static WideWs wide_layout(void *ws, int B, int P, int M, int64_t total_points) {
    WideWs w;
    const size_t BP = (size_t)B * P, K = (size_t)M + 2, a = 255;
    char *p = reinterpret_cast<char *>(ws);
    size_t off = 0;
    auto take = [&](size_t n) { char *r = p + off; off += (n + a) & ~a; return r; };
    w.keys_in = (uint32_t *)take(BP * 4); w.vals_in = (uint32_t *)take(BP * 4); w.keys_out = (uint32_t *)take(BP * 4); w.vals_out = (uint32_t *)take(BP * 4);
    w.pos = (int32_t *)take(BP * 4); w.order = (uint32_t *)take(BP * 4); w.pts4 = (float4 *)take(BP * 16);
    w.sums = (unsigned long long *)take((size_t)B * K * 8); w.base = (uint32_t *)take((size_t)B * K * 4); w.kpn = (int32_t *)take((size_t)B * K * 4);
    w.label_acc = (float *)take((size_t)B * K * 4); w.flags = (int32_t *)take((size_t)B * 16); w.cen4 = (float4 *)take((size_t)B * M * 16);
    w.temp = (float *)take(BP * 4);
    w.tiletab = (float *)take((size_t)B * FPS_TAB_ROWS * ((P + 31) / 32 + 4096) * 4);
    w.proj_bytes = (project_scratch_bytes(total_points, B, P) + 255) & ~(size_t)255;
    w.proj = take(w.proj_bytes);
    w.sort_bytes = wide_sort_bytes((int64_t)BP, wide_key_bits(B));
    w.sort_tmp = take(w.sort_bytes + 256);
    w.bytes = off;
    return w;
}
extern "C" size_t rpcc_wide_workspace_bytes(int B, int P, int M, int64_t total_points) {
    if (B <= 0 || P <= 0 || M <= 0) return 0;
    const size_t sorted = wide_layout(nullptr, B, P, M, total_points).bytes + 4096;
    // (up to RPCC_MAX_CLUSTERS_MID clusters the point model runs on the fused batch's own layout: compress_batch_mid)
    return M <= RPCC_MAX_CLUSTERS_MID ? std::max(sorted, rpcc_workspace_bytes_general(B, P, M, total_points)) : sorted;
}
// counts, sums, the sort and the positions of a segmentation (encoder and decoder)
template <class L>
static int wide_order(const L *seg, const float *ri_for_sums, const float *ri, const float *tm, int B, int P, int M, const WideWs &w, int32_t *counts,
                      int32_t *nnz, bool want_pts, hipStream_t st) {
    const int K = M + 2;
    const int64_t n = (int64_t)B * P;
    HIP_TRY(hipMemsetAsync(counts, 0, (size_t)B * K * 4, st));
    HIP_TRY(hipMemsetAsync(w.sums, 0, (size_t)B * K * 8, st));
    HIP_TRY(hipMemsetAsync(w.flags, 0, (size_t)B * 16, st));
    wide_keys_kernel<L><<<dim3((P + 255) / 256, B), 256, 0, st>>>(seg, ri_for_sums, P, K, w.keys_in, w.vals_in, counts, w.sums, w.flags);
    LAUNCH_CHECK();
    size_t sb = w.sort_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(w.sort_tmp, sb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)n, 0u, (unsigned)wide_key_bits(B), st));
    wide_bases_kernel<<<B, 256, 0, st>>>(counts, K, w.base, nnz);
    wide_positions_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(w.keys_out, w.vals_out, counts, P, K, n, w.pos, w.order, ri, tm, want_pts ? w.pts4 : nullptr);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_compress_batch_wide(const rpcc_batch_io *io, int B, rpcc_geom g, int M, double ground_threshold, float acc, void *ws, void *stream) {
    ARG_TRY(io != nullptr && ws != nullptr && B > 0 && B <= RPCC_MAX_BATCH && M > 0 && M <= RPCC_MAX_CLUSTERS_WIDE && g.H > 1 && g.W > 0);
    ARG_TRY(io->offsets && io->tm && io->ground && io->ri && io->seg && io->cen_pix && io->centers && io->model && io->counts && io->q16 && io->nnz && io->info);
    ARG_TRY(io->model_method == 0 || io->model_method == 1);
    ARG_TRY(!(io->flags & RPCC_FPS_MODE_BITS));
    ARG_TRY(point_floats(io->point_stride_bytes) > 0 && (io->point_stride_bytes != 16 || (reinterpret_cast<uintptr_t>(io->xyz) & 15u) == 0));
    if (io->nonuniform) ARG_TRY(io->salience && io->key_point_map && io->nonuniform->levels >= 1 && io->nonuniform->levels <= 8 &&
                                io->nonuniform->ground_level >= 0 && io->nonuniform->ground_level < io->nonuniform->levels);
    hipStream_t st = (hipStream_t)stream;
    const int P = g.H * g.W, K = M + 2;
    ARG_TRY(fps_tiling_range(g.H, g.W).T <= FPS_TILED_MAX_TILES && P < (1 << 22));
    if (mid_clusters_ok(io, g, M)) return compress_batch_mid(io, B, g, M, ground_threshold, acc, ws, st);
    const WideWs w = wide_layout(ws, B, P, M, io->total);
    uint16_t *seg = reinterpret_cast<uint16_t *>(io->seg);
    int rc;
    if ((rc = launch_project(io->xyz, io->offsets, io->total, 0, B, g, io->ri, w.proj, w.proj_bytes, st, nullptr, nullptr, nullptr, nullptr,
                             point_floats(io->point_stride_bytes), order_mode_of(io->flags), io->tm))) return rc;
    if (io->ground_seed >= 0 && (rc = launch_ground_ransac(io->ri, io->tm, B, P, (uint32_t)io->ground_seed, false, io->ground, nullptr, st, nullptr, io->frame_ids))) return rc;
    const bool brute = (io->flags & RPCC_FPS_BRUTEFORCE) != 0;
    if ((rc = launch_ground_mask(io->ri, io->tm, io->ground, ground_threshold, B, g.H, g.W, w.temp, io->info, brute ? nullptr : w.tiletab, st, false))) return rc;
    if ((rc = launch_fps_range(io->ri, io->tm, w.temp, io->info, B, g.H, g.W, M, io->cen_pix, io->centers, io->flags, false, brute ? nullptr : w.tiletab, nullptr, st))) return rc;
    const bool point = io->model_method == 0;
    wide_cen4_kernel<<<(B * M + 255) / 256, 256, 0, st>>>(io->centers, B * M, w.cen4);
    wide_assign_kernel<<<dim3((P + 255) / 256, B), 256, 0, st>>>(io->ri, io->tm, io->ground, w.cen4, P, M, seg);
    LAUNCH_CHECK();
    if ((rc = wide_order<uint16_t>(seg, point ? io->ri : nullptr, io->ri, io->tm, B, P, M, w, io->counts, io->nnz, !point, st))) return rc;
    if (point) {
        wide_point_model_kernel<uint16_t><<<dim3((K + 255) / 256, B), 256, 0, st>>>(io->ri, seg, io->ground, io->counts, w.sums, w.flags, P, K, io->model);
        LAUNCH_CHECK();
    } else {   // the plane fits read a label's slice of the ordered lists through "tile 0" of a one-tile offset table: base[b][k]
        PlaneParams pp;   // (as plane_group_args sets them)
        pp.cos_cut = io->plane_cos_cut; pp.thr = 0.1f; pp.min_points = 30; pp.iters = 10; pp.seed = (uint32_t)io->plane_seed; pp.frame_ids = io->frame_ids; pp.inject = nullptr;
        const int wpg = PL_THREADS / 64, groups = (K + wpg - 1) / wpg;
        plane_model_kernel<10><<<B * (K - 2) + B * groups, PL_THREADS, 0, st>>>(io->tm, w.order, w.pts4, w.base, io->counts, io->ground, B, P, M, K, 1, pp, PL_BIG, io->model);
        LAUNCH_CHECK();
    }
    const float *label_acc = nullptr;
    if (io->nonuniform) {
        const rpcc_nonuniform_cfg *nu = io->nonuniform;
        HIP_TRY(hipMemsetAsync(w.kpn, 0, (size_t)B * K * 4, st));
        if ((rc = launch_features<uint16_t>(io->ri, seg, B, g.H, g.W, nu->feature_region, nu->segments, nu->sharp_num, nu->less_sharp_num, nu->flat_num,
                                            nullptr, io->key_point_map, st, w.kpn, K))) return rc;
        SalienceParams sp;
        for (int i = 0; i < 8; i++) { sp.level_kp_num[i] = i < nu->levels ? nu->level_kp_num[i] : 0; sp.level_acc[i] = i < nu->levels ? nu->level_acc[i] : 0.f; }
        sp.levels = nu->levels; sp.ground_level = nu->ground_level;
        wide_salience_levels_kernel<<<dim3((K + 255) / 256, B), 256, 0, st>>>(io->counts, w.kpn, K, sp, io->salience, w.label_acc);
        LAUNCH_CHECK();
        label_acc = w.label_acc;
    }
    wide_quantise_kernel<uint16_t><<<dim3((P + 255) / 256, B), 256, 0, st>>>(io->ri, io->tm, seg, io->model, w.pos, acc, label_acc, P, K, io->q16);
    LAUNCH_CHECK();
    return RPCC_OK;
}

extern "C" int rpcc_contour_encode_wide(const uint16_t *seg, int B, int H, int W, uint8_t *contour_bits, uint16_t *idx_sequence, int32_t *nseq, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && seg && contour_bits && idx_sequence && nseq && ws);
    hipStream_t st = (hipStream_t)stream;
    const int P = H * W, T = ntiles(P);
    uint32_t *tile_cnt = reinterpret_cast<uint32_t *>(ws);
    contour_count_kernel<uint16_t><<<dim3(T, B), 256, 0, st>>>(seg, P, W, T, tile_cnt);
    tile_scan_kernel<<<B, 256, 0, st>>>(tile_cnt, T, nseq);
    contour_write_kernel<uint16_t><<<dim3(T, B), 256, 0, st>>>(seg, P, W, T, tile_cnt, contour_bits, idx_sequence);
    LAUNCH_CHECK();
    return RPCC_OK;
}
extern "C" int rpcc_contour_decode_wide(const uint8_t *contour_bits, const uint16_t *idx_sequence, int B, int H, int W, uint16_t *seg, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && H > 0 && W > 0 && seg && contour_bits && idx_sequence && ws);
    hipStream_t st = (hipStream_t)stream;
    const int P = H * W, T = ntiles(P);
    uint32_t *tile_cnt = reinterpret_cast<uint32_t *>(ws);
    contour_bits_count_kernel<<<dim3(T, B), 256, 0, st>>>(contour_bits, P, T, tile_cnt);
    tile_scan_kernel<<<B, 256, 0, st>>>(tile_cnt, T, nullptr);
    recover_map_kernel<uint16_t><<<dim3(T, B), 256, 0, st>>>(contour_bits, idx_sequence, P, T, tile_cnt, seg);
    LAUNCH_CHECK();
    return RPCC_OK;
}
// ws: rpcc_wide_workspace_bytes(B, P, M, 0) bytes
extern "C" int rpcc_decode_wide(const uint16_t *seg, const int16_t *q16, const float *model, const float *tm, const double *level_acc, int levels,
                                const uint8_t *salience, int B, int P, int M, float *ri_rec, float *pc_rec, void *ws, void *stream) {
    ARG_TRY(B > 0 && B <= RPCC_MAX_BATCH && P > 0 && M > 0 && M <= RPCC_MAX_CLUSTERS_WIDE && seg && q16 && model && tm && level_acc && ri_rec && ws);
    ARG_TRY(levels >= 0 && levels <= 8 && (levels == 0 || salience != nullptr));
    hipStream_t st = (hipStream_t)stream;
    const WideWs w = wide_layout(ws, B, P, M, 0);
    int rc;
    // (the per-label counts land in the workspace's key-point counters: the decoder has no use for either)
    if ((rc = wide_order<uint16_t>(seg, nullptr, nullptr, nullptr, B, P, M, w, w.kpn, nullptr, false, st))) return rc;
    DecodeSteps steps;
    steps.levels = levels;
    for (int i = 0; i < 8; i++) steps.acc[i] = i < (levels ? levels : 1) ? level_acc[i] : 0.0;
    wide_decode_kernel<uint16_t><<<dim3((P + 255) / 256, B), 256, 0, st>>>(seg, q16, model, tm, w.pos, salience, steps, P, M + 2, ri_rec, pc_rec);
    LAUNCH_CHECK();
    return RPCC_OK;
}
