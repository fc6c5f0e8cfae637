// Micro-benchmark: what v_mfma_f32_32x32x2_f32 sustains on gfx950 under the issue patterns of the conv kernels.
// Prints, per pattern, cycles per MFMA per SIMD (s_memtime), wall TFLOP/s and the in-kernel clock.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f32_rate mfma_f32_rate.hip && ./mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: NACC independent accumulators, round robin (one MFMA per accumulator per turn)
// MODE 1: chains of 4 dependent MFMAs per accumulator, NACC accumulators in turn (the conv kernels' pattern)
// VALU : extra independent v_fma per MFMA;  LDS: ds_read_b128 per 4 MFMAs (conflict-free, lane*16)
template <int MODE, int NACC, int VALU, int LDS>
__global__ __launch_bounds__(512) void k_rate(float *out, unsigned long long *stamps, int iters, float seed) {
    __shared__ f32x4 sm[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = f32x4{seed * i, 1.f, 2.f, 3.f};
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float va = seed + lane, vb = seed * 0.5f + lane;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed * i;
    f32x4 ld = sm[lane];
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if (LDS) ld = sm[(lane + 64 * ((it + a) & 31)) & 4095];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ai = MODE == 0 ? (a * 4 + q) % NACC : a;
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(va + (LDS ? ld[q] : 0.f), vb, acc[ai], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VALU; ++v) f[v & 7] = __builtin_fmaf(f[v & 7], 1.0001f, vb);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    float s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int MODE, int NACC, int VALU, int LDS>
void run(const char *name, int threads, float *out, unsigned long long *stamps, int nblk) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_rate<MODE, NACC, VALU, LDS>), dim3(nblk), dim3(threads), 0, 0, out, stamps, iters, 0.001f);
    CHECK(hipDeviceSynchronize());
    // hold the load for ~1 s so that the clock settles, then time
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL((k_rate<MODE, NACC, VALU, LDS>), dim3(nblk), dim3(threads), 0, 0, out, stamps, iters, 0.001f);
    CHECK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_rate<MODE, NACC, VALU, LDS>), dim3(nblk), dim3(threads), 0, 0, out, stamps, iters, 0.001f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * nblk);
    CHECK(hipMemcpy(st.data(), stamps, sizeof(unsigned long long) * 2 * nblk, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < nblk; ++i) { cyc += st[2 * i]; real += st[2 * i + 1]; }
    cyc /= nblk; real /= nblk;
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_wave = (double)iters * NACC * 4;
    const double cyc_per_mfma_simd = cyc / (mfma_per_wave * waves_per_simd);
    const double flop = 2.0 * 32 * 32 * 2 * mfma_per_wave * (threads / 64) * nblk * reps;
    printf("%-44s %3d thr  cyc/MFMA/SIMD %6.1f  clock %5.2f GHz  wall %6.1f TFLOP/s\n", name, threads, cyc_per_mfma_simd,
           cyc / real * 0.1, flop / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    int dev = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int nblk = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.gcnArchName, nblk);
    float *out;
    unsigned long long *stamps;
    CHECK(hipMalloc(&out, (size_t)nblk * 512 * 4));
    CHECK(hipMalloc(&stamps, (size_t)nblk * 16));
    run<0, 4, 0, 0>("independent x4, bare", 256, out, stamps, nblk);
    run<0, 4, 0, 0>("independent x4, bare", 512, out, stamps, nblk);
    run<1, 6, 0, 0>("chains of 4 over 6 acc, bare", 256, out, stamps, nblk);
    run<1, 6, 0, 0>("chains of 4 over 6 acc, bare", 512, out, stamps, nblk);
    run<1, 6, 3, 0>("chains of 4 over 6 acc, 3 v_fma per MFMA", 256, out, stamps, nblk);
    run<1, 6, 3, 0>("chains of 4 over 6 acc, 3 v_fma per MFMA", 512, out, stamps, nblk);
    run<1, 6, 0, 1>("chains of 4 over 6 acc, A from ds_read_b128", 512, out, stamps, nblk);
    run<1, 6, 3, 1>("chains of 4 over 6 acc, ds_read + 3 v_fma", 512, out, stamps, nblk);
    run<1, 6, 8, 1>("chains of 4 over 6 acc, ds_read + 8 v_fma", 512, out, stamps, nblk);
    run<1, 4, 0, 0>("chains of 4 over 4 acc (2x2 tiles), bare", 512, out, stamps, nblk);
    return 0;
}
