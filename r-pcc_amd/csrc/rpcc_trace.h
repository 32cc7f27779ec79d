// rpcc_trace.h -- developer trace hooks.  NOT part of the shipped library: included by rpcc_hip.hip only under -DRPCC_DEVTRACE
// (RPCC_EXTRA_FLAGS=-DRPCC_DEVTRACE python r-pcc_amd/build.py; tools_dev/fps_phases.sh, fps_balance.sh, wg_starts.py,
// ransac_phases.sh read the buffer).  rpcc_debug_stamps() registers a device int64 buffer; instrumented kernels store cycle
// counters into it.  Without the flag every hook below is an empty macro (rpcc_hip.hip) and rpcc_debug_stamps() refuses a buffer.
#pragma once

__device__ long long *g_dbg_stamps = nullptr;
extern "C" int rpcc_debug_stamps(void *dev_i64_buffer) {
    long long *p = reinterpret_cast<long long *>(dev_i64_buffer);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps), &p, sizeof(p)));
    return RPCC_OK;
}
// shader clock of block 0 / thread 0 at a phase boundary -> stamps[slot]
#define DBG_STAMP(slot_)                                                                     \
    do {                                                                                     \
        if (g_dbg_stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) \
            g_dbg_stamps[slot_] = (long long)__builtin_readcyclecounter();                   \
    } while (0)

// ---- fps_regtab_kernel ---------------------------------------------------------------------------------------------------
// per wavefront of block 0: cycles per phase of the iteration chain summed over the iterations -> stamps[64 + wave * 8 + phase]
// (phase 6: tiles visited, 7: iterations with a visit); visit rounds: cycles issuing the loads / waiting for the data /
// updating, rounds -> stamps[3000 + wave * 4 ..]; wall clock (100 MHz) at the start and the end of every workgroup ->
// stamps[2048 + 2 b ..]; tiles to visit per iteration and wavefront of blocks 0..15 -> stamps[4096 + ((block * 128 + j) * 8 + wave)]
#define TRACE_FPS_DECLS() long long tr_p[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last = (long long)__builtin_readcyclecounter(), tr_v[4] = {0, 0, 0, 0}, tr_t[3] = {0, 0, 0}
#define TRACE_FPS_PHASE(i_) do { const long long t_ = (long long)__builtin_readcyclecounter(); tr_p[i_] += t_ - tr_last; tr_last = t_; } while (0)
#define TRACE_FPS_VISIT(k_)                                                                                                   \
    do {                                                                                                                      \
        if ((k_) == 1) { tr_t[1] = (long long)__builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
        tr_t[(k_) == 0 ? 0 : 2] = (long long)__builtin_readcyclecounter();                                                  \
        if ((k_) == 1) { tr_v[0] += tr_t[1] - tr_t[0]; tr_v[1] += tr_t[2] - tr_t[1]; }                                        \
        if ((k_) == 2) { tr_v[2] += (long long)__builtin_readcyclecounter() - tr_t[2]; tr_v[3] += 1; }                        \
    } while (0)
#define TRACE_FPS_TILES(vm_, j_)                                                                                              \
    do {                                                                                                                      \
        tr_p[6] += __popcll(vm_); tr_p[7] += (vm_) != 0ull;                                                                   \
        if (g_dbg_stamps != nullptr && blockIdx.x < 16 && (threadIdx.x & 63) == 0 && (j_) < 128 && (threadIdx.x >> 6) < 8)    \
            g_dbg_stamps[4096 + ((blockIdx.x * 128 + (j_)) * 8 + (threadIdx.x >> 6))] = __popcll(vm_);                        \
    } while (0)
#define TRACE_FPS_WG(end_)                                                                                                    \
    do {                                                                                                                      \
        if (g_dbg_stamps != nullptr && threadIdx.x == 0) g_dbg_stamps[2048 + 2 * blockIdx.x + (end_)] = (long long)wall_clock64(); \
        if ((end_) && g_dbg_stamps != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                               \
            for (int i_ = 0; i_ < 8; i_++) g_dbg_stamps[64 + (threadIdx.x >> 6) * 8 + i_] = tr_p[i_];                         \
            for (int i_ = 0; i_ < 4; i_++) g_dbg_stamps[3000 + (threadIdx.x >> 6) * 4 + i_] = tr_v[i_];                       \
        }                                                                                                                     \
    } while (0)

// ---- project_ordered_kernel: cycles per phase of the chunk loop, summed over the chunks (thread 0 of block 0) -> stamps[900 + phase]; phase 15: chunks
#define TRACE_ORD_DECLS() long long to_p[16] = {0}, to_last = (long long)__builtin_readcyclecounter()
#define TRACE_ORD_PHASE(i_) do { const long long t_ = (long long)__builtin_readcyclecounter(); to_p[i_] += t_ - to_last; to_last = t_; } while (0)
#define TRACE_ORD_COUNT(i_) do { to_p[i_] += 1; } while (0)
#define TRACE_ORD_END() do { if (g_dbg_stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 16; i_++) g_dbg_stamps[900 + i_] = to_p[i_]; } while (0)
