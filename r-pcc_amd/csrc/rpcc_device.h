// rpcc_device.h -- device-side helpers shared by the gfx950 kernels of librpcc_hip.so.
//
// Numerics contract (DESIGN.md "Arithmetic"): every fp32/fp64 expression reproduces the reference's
// un-fused x86 SSE arithmetic, so this translation unit is compiled with -ffp-contract=off, IEEE
// divide/sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt) and denormals preserved.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RPCC_WAVE 64

namespace rpcc {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// ---------------------------------------------------------------------------------------------
// glibc 2.35 atanf / atan2f (Sun fdlibm float algorithm) operation sequence.  The reference
// projection calls libm atan2f (cpp_modules.cpp:447,450) whose result is not correctly rounded, so
// the sequence itself is reproduced: plain fp32 + - * /, fabsf and integer tests only.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float atanf_fdlibm(float x) {
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f, a3 = -1.1111110449e-01f,
                a4 = 9.0908870101e-02f, a5 = -7.6918758452e-02f, a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f,
                a8 = 4.9768779427e-02f, a9 = -3.6531571299e-02f, a10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {
        if (ix > 0x7f800000) return x + x;
        return (hx > 0) ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x31000000) return x;
    // The four argument reductions differ only in (numerator, denominator, hi, lo): all four are formed with
    // the reference's own operations and selected per lane, so a wavefront whose lanes fall into different
    // ranges executes ONE correctly rounded division instead of four divergent ones.  Without reduction
    // (|x| < 0.4375) the division is x / 1 = x exactly.
    const float ax = fabsf(x);
    const bool reduced = ix >= 0x3ee00000, r0 = ix < 0x3f300000, r1 = ix < 0x3f980000, r2 = ix < 0x401c0000;
    const float n0 = 2.0f * ax - 1.0f, d0 = 2.0f + ax;       // [0.4375, 0.6875)
    const float n1 = ax - 1.0f, d1 = ax + 1.0f;              // [0.6875, 1.1875)
    const float n2 = ax - 1.5f, d2 = 1.0f + 1.5f * ax;       // [1.1875, 2.4375)
    float num = r2 ? (r1 ? (r0 ? n0 : n1) : n2) : -1.0f;     // else -1 / x
    float den = r2 ? (r1 ? (r0 ? d0 : d1) : d2) : ax;
    const float hi = r2 ? (r1 ? (r0 ? hi0 : hi1) : hi2) : hi3;
    const float lo = r2 ? (r1 ? (r0 ? lo0 : lo1) : lo2) : lo3;
    num = reduced ? num : x;
    den = reduced ? den : 1.0f;
    x = num / den;
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    const float t = x * (s1 + s2);
    const float r = hi - ((t - lo) - x);
    return reduced ? ((hx < 0) ? -r : r) : x - t;
}

__device__ __forceinline__ float atan2f_fdlibm(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_fdlibm(y);
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m < 2) return y;
        return (m == 2) ? pi + tiny : -pi - tiny;
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
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_fdlibm(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return u2f(f2u(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// v_min_f32 / v_max_f32 as single instructions.  fminf / fmaxf on a value the compiler cannot prove to be no signalling
// NaN (a select, a loop-carried value) are lowered to a canonicalising v_max_f32 x, x, x in front of the v_min / v_max; the
// callers here feed coordinates and +-inf only (never NaN), for which the bare instruction gives the same result.
// CAUTION: the compiler's hazard recogniser does not look into inline assembly -- an operand produced by a transcendental
// instruction (v_sqrt / v_rcp / ...) just before needs a wait state the compiler will not insert (found the hard way in the
// pixel kernel, round 3).  Only use these on values that come from plain VALU arithmetic, selects or loads.
__device__ __forceinline__ float fmin_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// three operands at once; a quiet-NaN operand is skipped (IEEE mode: v_min / v_max return the other operand), so "value or NaN" masks an element
// with one select instead of one per bound
__device__ __forceinline__ float fmin3_raw(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fmax3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// ---------------------------------------------------------------------------------------------
// Loads / stores at "wave-uniform base + 32-bit byte offset".  Written with an explicit unsigned byte offset so that the
// compiler can use the scalar-base addressing mode (global_load v, v_offset, s[base]) instead of building a 64-bit
// address in VGPRs for every access (index * 4 as a 64-bit shift-add): the caller guarantees offset < 2^32.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ld_f32(const float *base, uint32_t byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void st_f32(float *base, uint32_t byte_off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
struct f32x3 { float x, y, z; };  // one ray of the transform map: a single 12-byte load
// four floats at a 4-byte aligned address: one global_load_dwordx4 (dword alignment is all the hardware asks).  A VECTOR type on purpose: a struct of four
// floats is four scalar loads to the optimiser, which then folds the element-wise row-end branch and the quad branch of the EDGE kernels into one set of
// sixteen scalar loads with selected addresses (rounds 3-5 shipped that: 156 global_load_dword in the EDGE FPS kernel, no dwordx4).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
// A quad of floats at base + byte_off: U (unaligned) -> the 4-byte aligned vector type named in the access itself (a template parameter would drop the
// typedef's alignment and type the access as 16-byte aligned), else one aligned 16-byte access.
template <bool U>
__device__ __forceinline__ float4 ld_quad(const float *base, uint32_t byte_off) {
    const char *a = reinterpret_cast<const char *>(base) + byte_off;
    if (U) { const f32x4u v = *reinterpret_cast<const f32x4u *>(a); return make_float4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const float4 *>(a);
}
template <bool U>
__device__ __forceinline__ void st_quad(float *base, uint32_t byte_off, float x, float y, float z, float w) {
    char *a = reinterpret_cast<char *>(base) + byte_off;
    if (U) { f32x4u v; v.x = x; v.y = y; v.z = z; v.w = w; *reinterpret_cast<f32x4u *>(a) = v; }
    else *reinterpret_cast<float4 *>(a) = make_float4(x, y, z, w);
}
template <class T>
__device__ __forceinline__ T ld_at(const T *base, uint32_t byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void st_at(T *base, uint32_t byte_off, T v) {
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// ---------------------------------------------------------------------------------------------
// Wavefront (64 lanes) reductions.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, RPCC_WAVE);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RPCC_WAVE);
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, RPCC_WAVE));
    return v;
}

// ---------------------------------------------------------------------------------------------
// DPP reductions (no LDS crossbar): inclusive scan inside each 16-lane row with row_shr 1/2/4/8, then
// row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3); lane 63 holds the total, read with readlane.
// `id` is the identity of the operation (lanes without a source keep it).
// ---------------------------------------------------------------------------------------------
#define RPCC_DPP(old_, src_, ctrl_, rmask_) __builtin_amdgcn_update_dpp((int)(old_), (int)(src_), ctrl_, rmask_, 0xf, false)

__device__ __forceinline__ uint32_t dpp_max_u32(uint32_t v) {
#define STEP_(ctrl_, rm_) v = max(v, (uint32_t)RPCC_DPP(0, v, ctrl_, rm_))
    STEP_(0x111, 0xf); STEP_(0x112, 0xf); STEP_(0x114, 0xf); STEP_(0x118, 0xf); STEP_(0x142, 0xa); STEP_(0x143, 0xc);
#undef STEP_
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// maximum of a value that repeats with period 8 along the lanes (8 candidates): three xor steps inside every group of 8
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror) instead of six row steps
__device__ __forceinline__ uint32_t dpp_max8_u32(uint32_t v) {
    v = max(v, (uint32_t)RPCC_DPP(0, v, 0xB1, 0xf));   // quad_perm [1,0,3,2]
    v = max(v, (uint32_t)RPCC_DPP(0, v, 0x4E, 0xf));   // quad_perm [2,3,0,1]
    v = max(v, (uint32_t)RPCC_DPP(0, v, 0x141, 0xf));  // row_half_mirror
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// float min / max as ONE v_max_u32_dpp per step: floats are mapped to order-preserving unsigned keys
// (sign bit flipped for non-negative values, all bits flipped for negative ones).  -0 orders below +0;
// NaNs are not expected here (the callers feed coordinates and +-inf identities).
__device__ __forceinline__ uint32_t f32_order_key(float f) {
    const uint32_t b = f2u(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t k) {
    return u2f(k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu));
}
__device__ __forceinline__ float dpp_max_f32(float v) { return f32_from_order_key(dpp_max_u32(f32_order_key(v))); }
__device__ __forceinline__ float dpp_min_f32(float v) { return f32_from_order_key(~dpp_max_u32(~f32_order_key(v))); }
__device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
#define STEP_(ctrl_, rm_) v = min(v, (uint32_t)RPCC_DPP(-1, v, ctrl_, rm_))
    STEP_(0x111, 0xf); STEP_(0x112, 0xf); STEP_(0x114, 0xf); STEP_(0x118, 0xf); STEP_(0x142, 0xa); STEP_(0x143, 0xc);
#undef STEP_
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t dpp_sum_u32(uint32_t v) {
#define STEP_(ctrl_, rm_) v = v + (uint32_t)RPCC_DPP(0, v, ctrl_, rm_)
    STEP_(0x111, 0xf); STEP_(0x112, 0xf); STEP_(0x114, 0xf); STEP_(0x118, 0xf); STEP_(0x142, 0xa); STEP_(0x143, 0xc);
#undef STEP_
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// inclusive prefix sum over the 64 lanes (six v_add_u32_dpp); all lanes must be active
__device__ __forceinline__ uint32_t dpp_scan_incl_u32(uint32_t v) {
#define STEP_(ctrl_, rm_) v = v + (uint32_t)RPCC_DPP(0, v, ctrl_, rm_)
    STEP_(0x111, 0xf); STEP_(0x112, 0xf); STEP_(0x114, 0xf); STEP_(0x118, 0xf); STEP_(0x142, 0xa); STEP_(0x143, 0xc);
#undef STEP_
    return v;
}

// Bounding box of a wavefront in ONE block of native float DPP instructions: three minima and three maxima reduced
// together, the six chains interleaved step by step (five independent instructions between two dependent ones cover
// the DPP read-after-write wait states, so no s_nop and no order-key conversion: 36 + 6 instructions instead of 6 x 13).
// Lanes without a DPP source keep their value (bound_ctrl off), which is the identity of min / max.  All 64 lanes must be
// active; the inputs are coordinates or +-inf, never NaN.
#define RPCC_BOX6_STEP_(ctrl_)                                  \
    "v_min_f32_dpp %0, %0, %0 " ctrl_ "\n\t"                     \
    "v_min_f32_dpp %1, %1, %1 " ctrl_ "\n\t"                     \
    "v_min_f32_dpp %2, %2, %2 " ctrl_ "\n\t"                     \
    "v_max_f32_dpp %3, %3, %3 " ctrl_ "\n\t"                     \
    "v_max_f32_dpp %4, %4, %4 " ctrl_ "\n\t"                     \
    "v_max_f32_dpp %5, %5, %5 " ctrl_ "\n\t"
__device__ __forceinline__ void dpp_box6(float &lo0, float &lo1, float &lo2, float &hi0, float &hi1, float &hi2) {
    asm volatile("s_nop 1\n\t"
                 RPCC_BOX6_STEP_("row_shr:1 row_mask:0xf bank_mask:0xf")
                 RPCC_BOX6_STEP_("row_shr:2 row_mask:0xf bank_mask:0xf")
                 RPCC_BOX6_STEP_("row_shr:4 row_mask:0xf bank_mask:0xf")
                 RPCC_BOX6_STEP_("row_shr:8 row_mask:0xf bank_mask:0xf")
                 RPCC_BOX6_STEP_("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 RPCC_BOX6_STEP_("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(lo0), "+v"(lo1), "+v"(lo2), "+v"(hi0), "+v"(hi1), "+v"(hi2));
    lo0 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(lo0), 63));
    lo1 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(lo1), 63));
    lo2 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(lo2), 63));
    hi0 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(hi0), 63));
    hi1 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(hi1), 63));
    hi2 = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(hi2), 63));
}

// single minimum, native float DPP (a dependent chain: s_nop covers the wait states; still 6 VALU instructions
// instead of 13 with the order keys).  Same conditions as dpp_box6.
__device__ __forceinline__ float dpp_min_f32_native(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(v), 63));
}

__device__ __forceinline__ float dpp_max_f32_native(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(v), 63));
}

// FPS arg-max key: larger squared distance first, then LOWER index (the sequential strict-'>' scan of
// ops/fps/src/sampling_gpu.cu:67-68 restated as a total order).  value < 0 means "not a candidate".
__device__ __forceinline__ unsigned long long fps_key(float v, uint32_t idx) {
    const uint32_t hi = (v < 0.0f) ? 0u : f2u(v) + 1u;
    return ((unsigned long long)hi << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ uint32_t fps_key_index(unsigned long long k) { return 0xFFFFFFFFu - (uint32_t)k; }

}  // namespace rpcc

// launches over the frames of several geometry groups (rpcc_compress_batch_mixed): group i owns the workgroups first[i] .. first[i + 1] - 1
__device__ __forceinline__ int multi_group_of(const int *first, int n, int wg) {   // (workgroup-uniform)
    int gi = 0;
    for (int i = 1; i < n; i++) gi += wg >= first[i] ? 1 : 0;
    return gi;
}
// The same for the pixel-parallel kernels, whose workgroups are (frame, tile workgroup of the frame) pairs: group i owns the workgroups
// first[i] .. first[i + 1] - 1 = its frames x per[i] workgroups per frame.  A: the kernel's arguments for one group.
#ifndef RPCC_MAX_GROUPS
#define RPCC_MAX_GROUPS 4
#endif
template <class A>
struct MultiArgs {
    int n, first[RPCC_MAX_GROUPS + 1], per[RPCC_MAX_GROUPS];
    A a[RPCC_MAX_GROUPS];
};
template <class A>
__device__ __forceinline__ const A &multi_locate(const MultiArgs<A> &m, int &b, int &t) {   // (workgroup-uniform)
    const int gi = multi_group_of(m.first, m.n, (int)blockIdx.x);
    const int w = (int)blockIdx.x - m.first[gi];
    b = w / m.per[gi];
    t = w - b * m.per[gi];
    return m.a[gi];
}
