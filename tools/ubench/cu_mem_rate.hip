// Micro-benchmark 3: what one CU's vector-memory path moves per clock on gfx950, by instruction kind and source:
//   load   global_load_dwordx4 -> VGPRs            (register staging)
//   dma    global_load_lds_dwordx4 -> LDS          (LDS-DMA, what conv_qp / conv_w2d stage their operands with)
//   store  global_store_dwordx4
// One workgroup per CU (256 workgroups, W waves each), every wave walks its own window of `win` KiB `iters` times (window small:
// the data stays in L2 -- or in the 32 KiB L1 when it fits; large: HBM stream), `inflight` instructions between waits.
// Output: bytes per clock per CU from the wall time at the reported shader clock, and from s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { K_LOAD, K_DMA, K_STORE };

template <int KIND, int INF>
__global__ __launch_bounds__(512) void k_rate(const f32x4 *src, f32x4 *dst, int win_pieces, int iters, unsigned long long *cyc, float *sink, int share, int rot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    // a piece = 64 lanes x 16 B = 1 KiB; wave w of workgroup b owns pieces [(b * nw + w) * win_pieces, +win_pieces)
    // share > 1: groups of `share` workgroups that land on the same XCD (ids 8 apart) read the SAME windows (the operand tiles
    // 32 CUs of an XCD share in the conv kernels); rot: each workgroup starts its walk at a different piece of the window
    const int grp = share > 1 ? (blockIdx.x % 8) + 8 * ((blockIdx.x / 8) / share) : blockIdx.x;
    const size_t base = ((size_t)grp * nw + wave) * (size_t)win_pieces * 64 + lane;
    const int p_rot = rot ? ((blockIdx.x / 8) * rot) % win_pieces : 0;
    const f32x4 *s = src + base;
    f32x4 *d = dst + base;
    char *l = smem + wave * INF * 1024;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[INF];
#pragma unroll
    for (int k = 0; k < INF; ++k) v[k] = f32x4{(float)lane, 1.f, 2.f, (float)k};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it)
        for (int p = 0; p + INF <= win_pieces; p += INF) {
            if (KIND == K_LOAD) {
#pragma unroll
                for (int k = 0; k < INF; ++k) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(v[k]) : "v"(s + (size_t)((p + k + p_rot) % win_pieces) * 64));
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[INF - 1])::"memory");
#pragma unroll
                for (int k = 0; k < INF; ++k) asm volatile("" : "+v"(v[k]));
            } else if (KIND == K_DMA) {
#pragma unroll
                for (int k = 0; k < INF; ++k)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s + (size_t)((p + k + p_rot) % win_pieces) * 64),
                                                     (__attribute__((address_space(3))) void *)(l + k * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int k = 0; k < INF; ++k) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(d + (size_t)(p + k) * 64), "v"(v[k]) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
#pragma unroll
    for (int k = 0; k < INF; ++k) acc += v[k];
    if (KIND == K_DMA) acc += *(const f32x4 *)(l + lane * 16);
    if (lane == 0) cyc[blockIdx.x * nw + wave] = t1 - t0;
    if (acc[0] == 12345.678f) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int KIND, int INF> double run(const char *name, int waves, int win_kib, int iters, const f32x4 *src, f32x4 *dst, unsigned long long *cyc, float *sink, int cus, double mhz, int share = 1, int rot = 0) {
    const int win_pieces = win_kib;
    const size_t lds = (size_t)waves * INF * 1024;
    CHECK(hipFuncSetAttribute((const void *)k_rate<KIND, INF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    k_rate<KIND, INF><<<cus, waves * 64, lds>>>(src, dst, win_pieces, 1, cyc, sink, share, rot);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_rate<KIND, INF><<<cus, waves * 64, lds>>>(src, dst, win_pieces, iters, cyc, sink, share, rot);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)cus * waves);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= h.size();
    const double bytes_cu = (double)waves * (win_pieces / INF * INF) * 1024.0 * iters;
    const double wall_clk = ms * 1e-3 * mhz * 1e6;
    printf("%-5s share %2d rot %2d waves %d inflight %2d window %6d KiB/wave (%7.1f MiB total): %7.3f ms  %6.1f B/clk/CU (wall)  %6.1f B/clk/CU (s_memtime, %0.0f ticks)  %7.2f TB/s aggregate\n",
           name, share, rot, waves, INF, win_kib, (double)cus * waves * win_kib / 1024.0, ms, bytes_cu / wall_clk, bytes_cu / mean, mean, bytes_cu * cus / (ms * 1e-3) / 1e12);
    return ms;
}

int main(int argc, char **argv) {
    int dev = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("%s: %d CUs, %.0f MHz\n", prop.gcnArchName, cus, mhz);
    const size_t max_bytes = (size_t)cus * 8 * 4096 * 1024;   // 8 waves x 4 MiB windows
    f32x4 *src, *dst;
    unsigned long long *cyc;
    float *sink;
    CHECK(hipMalloc(&src, max_bytes));
    CHECK(hipMalloc(&dst, max_bytes));
    CHECK(hipMalloc(&cyc, (size_t)cus * 8 * 8));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(src, 0, max_bytes));
    CHECK(hipMemset(dst, 0, max_bytes));
    for (int waves : {4, 8}) {
        // L2-resident: 16 KiB per wave (up to 32 MiB over the chip is more than the L2s hold: use 8 KiB/wave = 16 MiB at 8 waves)
        for (int win : {8, 16, 32, 64, 4096}) {
            const int iters = win == 4096 ? 2 : (win >= 32 ? 64 : 512);
            run<K_LOAD, 8>("load", waves, win, iters, src, dst, cyc, sink, cus, mhz);
            run<K_DMA, 8>("dma", waves, win, iters, src, dst, cyc, sink, cus, mhz);
            run<K_STORE, 8>("store", waves, win, iters, src, dst, cyc, sink, cus, mhz);
        }
    }
    printf("-- few CUs active: is the store / HBM-stream rate a per-CU limit or the chip's?\n");
    for (int ncu : {8, 32, 64, 128, 256}) {
        run<K_STORE, 8>("store", 8, 64, 64, src, dst, cyc, sink, ncu, mhz);
        run<K_LOAD, 8>("load", 8, 4096, 2, src, dst, cyc, sink, ncu, mhz);
    }
    printf("-- operand sharing inside an XCD (one window set per `share` workgroups of the same XCD)\n");
    for (int share : {1, 4, 32})
        for (int rot : {0, 5})
            for (int win : {32, 64}) {
                run<K_LOAD, 8>("load", 4, win, 64, src, dst, cyc, sink, cus, mhz, share, rot);
                run<K_DMA, 8>("dma", 4, win, 64, src, dst, cyc, sink, cus, mhz, share, rot);
            }
    return 0;
}
