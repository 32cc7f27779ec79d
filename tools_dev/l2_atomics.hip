// developer microbenchmark (GPU box): can the projection's per-pixel minimum be taken with atomics that STAY in one XCD's L2?
// Device-scope atomics on the batch's 134 MB image cost a fabric transaction per point (round 1: 1.06 ms per batch), which is why the
// projection became two kernels (6-byte records + a band workgroup with 128 KB of LDS).  On this eight-XCD part an atomic of a scope
// below "agent" is performed by the L2 of the XCD that issues it; if every point of a frame is handled by workgroups of ONE XCD (the
// workgroup reads its XCC_ID and takes frames from that XCD's queue), the frame's 512 KB image lives in that L2 for the launch and the
// kernel boundary writes it back.  This program measures it on the headline's shape (256 frames x 131072 pixels, 29 M points with random
// pixels) and checks the result against a reference minimum:
//   agent        device-scope atomicMin, any workgroup any frame (the round-1 form)
//   wg-scope     workgroup-scope atomicMin, frames bound to the XCD that the workgroup runs on
//   hipcc --offload-arch=gfx950 -O3 tools_dev/l2_atomics.hip -o /tmp/l2_atomics && /tmp/l2_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return (int)(v & 0xF); }

// point i of frame f -> (pixel, depth bits): a stand-in for the pixel computation (cheap, so the atomics are what is timed)
__device__ __forceinline__ void point_of(int f, int i, int P, uint32_t &pix, uint32_t &dep) {
    const uint32_t h = hash32((uint32_t)f * 0x9E3779B9u + (uint32_t)i);
    pix = h % (uint32_t)P;
    dep = 0x3F800000u + (hash32(h) >> 9);
}

__global__ __launch_bounds__(256) void fill_kernel(uint32_t *img, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) img[i] = 0xFFFFFFFFu;
}

template <int MODE>   // 0: agent scope, chunks dealt to any workgroup; 1: workgroup scope, frames bound to the workgroup's XCD
__global__ __launch_bounds__(256) void min_kernel(uint32_t *img, int B, int P, int N, int *queues, int *xcc_seen) {
    __shared__ int s_chunk;
    const int chunks_per_frame = (N + 2047) / 2048;
    const int xcc = xcc_id();
    if (threadIdx.x == 0 && blockIdx.x < 4096) xcc_seen[blockIdx.x] = xcc;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_chunk = atomicAdd(&queues[MODE == 1 ? xcc * 32 : 0], 1);   // (one device atomic per 2048 points)
        __syncthreads();
        int f, c;
        if (MODE == 1) {   // this XCD's frames: xcc, xcc + 8, ...
            const int per = (B + 7 - xcc) / 8;
            if (s_chunk >= per * chunks_per_frame) return;
            f = xcc + 8 * (s_chunk / chunks_per_frame); c = s_chunk % chunks_per_frame;
        } else {
            if (s_chunk >= B * chunks_per_frame) return;
            f = s_chunk / chunks_per_frame; c = s_chunk % chunks_per_frame;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = c * 2048 + k * 256 + threadIdx.x;
            if (i < N) {
                uint32_t pix, dep;
                point_of(f, i, P, pix, dep);
                uint32_t *a = img + (size_t)f * P + pix;
                if (MODE == 1) __hip_atomic_fetch_min(a, dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_min(a, dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main() {
    const int B = 256, P = 131072, N = 113000;   // 29 M points per batch
    uint32_t *img[2];
    int *queues, *seen;
    for (int m = 0; m < 2; m++) CHECK(hipMalloc(&img[m], (size_t)B * P * 4));
    CHECK(hipMalloc(&queues, 8 * 32 * 4));
    CHECK(hipMalloc(&seen, 4096 * 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) for (int m = 0; m < 2; m++) {
        fill_kernel<<<4096, 256>>>(img[m], (size_t)B * P);
        CHECK(hipMemset(queues, 0, 8 * 32 * 4));
        CHECK(hipDeviceSynchronize());
        hipEventRecord(e0);
        if (m == 0) min_kernel<0><<<4096, 256>>>(img[m], B, P, N, queues, seen);
        else min_kernel<1><<<4096, 256>>>(img[m], B, P, N, queues, seen);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-9s %8.1f us for %.1f M atomics\n", m == 0 ? "agent" : "wg-scope", ms * 1e3, (double)B * N / 1e6);
    }
    std::vector<uint32_t> a((size_t)B * P), b((size_t)B * P);
    CHECK(hipMemcpy(a.data(), img[0], a.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), img[1], b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, touched = 0;
    for (size_t i = 0; i < a.size(); i++) { bad += a[i] != b[i]; touched += a[i] != 0xFFFFFFFFu; }
    std::vector<int> sx(4096);
    CHECK(hipMemcpy(sx.data(), seen, 4096 * 4, hipMemcpyDeviceToHost));
    int rr = 0; for (int i = 0; i < 4096; i++) rr += sx[i] == i % 8;
    printf("pixels touched %zu, mismatches wg-scope vs agent: %zu; workgroups whose XCC_ID == blockIdx %% 8: %d of 4096\n", touched, bad, rr);
    return bad != 0;
}
