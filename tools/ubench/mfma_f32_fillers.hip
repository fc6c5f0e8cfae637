// Micro-benchmark 2: what one filler instruction costs next to v_mfma_f32_32x32x2_f32 (gfx950), by instruction kind.
// Pattern per wave: 24 MFMAs (6 accumulators x chains of 4) per iteration with NF fillers after every MFMA (inline asm, so the
// compiler neither packs nor moves them).  Reports wall TFLOP/s and the cycles each filler adds per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { F_NONE, F_FMA, F_PKFMA, F_PKADD, F_ADDU32, F_CNDMASK, F_MOV, F_DSREAD, F_DSREAD_C4, F_SALU, F_XOR, F_MUL, F_GLOAD, F_GLDS, F_GSTORE, F_DSWRITE };

template <int KIND> __device__ __forceinline__ void filler(float &a, float &b, f32x2 &p, f32x2 &q, int &i, f32x4 &ld, int addr, const f32x4 *gsrc, f32x4 *gdst, char *lds) {
    // "+v": the destination stays live across the asynchronous load (a pure output would let the compiler reuse its registers for
    // the next address while the data is still in flight)
    if (KIND == F_GLOAD) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(ld) : "v"(gsrc));
    if (KIND == F_GLDS) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    if (KIND == F_GSTORE) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(gdst), "v"(ld) : "memory");
    if (KIND == F_DSWRITE) asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(ld) : "memory");
    if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    if (KIND == F_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (KIND == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));
    if (KIND == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    if (KIND == F_ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i) : "v"(addr));
    if (KIND == F_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i) : "v"(addr));
    if (KIND == F_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
    if (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
    if (KIND == F_DSREAD || KIND == F_DSREAD_C4) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(addr));
    if (KIND == F_SALU) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
}

template <int KIND, int NF, int EVERY>   // NF fillers after every EVERY-th MFMA
__global__ __launch_bounds__(512) void k_fill(float *out, int iters, float seed, const f32x4 *gsrc, f32x4 *gdst) {
    __shared__ f32x4 sm[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = f32x4{seed * i, 1.f, 2.f, 3.f};
    __syncthreads();
    f32x16 acc[6];
    for (int a = 0; a < 6; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float va = seed + lane, vb = seed * 0.5f + lane;
    float fa = seed, fb = 1.0001f;
    f32x2 p = {seed, seed}, q = {1.0001f, 0.9999f};
    int ii = lane;
    f32x4 ld = {0, 0, 0, 0};
    // conflict-free: lane*16; 4-way conflict: 64-byte stride between lanes (the group reads of conv_w1d)
    const int addr = (int)(size_t)sm + (KIND == F_DSREAD_C4 ? (lane & 31) * 64 + (lane >> 5) * 16 : lane * 16);
    const int wave = threadIdx.x >> 6;
    const f32x4 *gs = gsrc + ((size_t)blockIdx.x * 8 + wave) * 4096 + lane;     // 64 KB window per wave, L2-resident after the first pass
    f32x4 *gd = gdst + ((size_t)blockIdx.x * 8 + wave) * 4096 + lane;
    char *ldsw = (char *)sm + wave * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, acc[a], 0, 0, 0);
                if ((a * 4 + qq) % EVERY == 0) {
#pragma unroll
                    for (int v = 0; v < NF; ++v) filler<KIND>(fa, fb, p, q, ii, ld, addr, gs + 64 * ((it * 24 + a * 4 + qq) & 63), gd + 64 * ((it * 24 + a * 4 + qq) & 63), ldsw);
                }
            }
        if (KIND == F_DSREAD || KIND == F_DSREAD_C4 || KIND == F_DSWRITE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == F_GLOAD || KIND == F_GLDS || KIND == F_GSTORE) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld) :: "memory");
    }
    float s = fa + p[0] + p[1] + ii + ld[0] + ld[3];
    for (int a = 0; a < 6; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_base[2] = {0, 0};
template <int KIND, int NF, int EVERY>
void run(const char *name, int threads, float *out, int nblk, const f32x4 *gsrc = nullptr, f32x4 *gdst = nullptr) {
    static f32x4 *g_src = nullptr, *g_dst = nullptr;
    if (!g_src) { CHECK(hipMalloc(&g_src, (size_t)nblk * 8 * 4096 * 16 + 65536)); CHECK(hipMalloc(&g_dst, (size_t)nblk * 8 * 4096 * 16 + 65536)); CHECK(hipMemset(g_src, 0, (size_t)nblk * 8 * 4096 * 16)); }
    gsrc = g_src; gdst = g_dst;
    const int iters = 3000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k_fill<KIND, NF, EVERY>), dim3(nblk), dim3(threads), 0, 0, out, iters, 0.001f, gsrc, gdst);
    CHECK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_fill<KIND, NF, EVERY>), dim3(nblk), dim3(threads), 0, 0, out, iters, 0.001f, gsrc, gdst);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)iters * 24 * (threads / 256) * reps;
    const double ns_per_mfma = ms * 1e6 / mfma_per_simd;          // per SIMD
    const double flop = 2.0 * 32 * 32 * 2 * (double)iters * 24 * (threads / 64) * nblk * reps;
    const int w2 = threads == 512;
    if (KIND == F_NONE) g_base[w2] = ns_per_mfma;
    const double fill_per_mfma = (double)NF / EVERY;
    printf("%-34s %d w/SIMD  %6.1f TFLOP/s  %6.2f ns/MFMA/SIMD", name, threads / 256, flop / (ms * 1e-3) / 1e12, ns_per_mfma);
    if (KIND != F_NONE) printf("  +%.2f ns per filler (= %.1f cyc at 2.4 GHz; %.2f fillers/MFMA)", (ns_per_mfma - g_base[w2]) / fill_per_mfma,
                               (ns_per_mfma - g_base[w2]) / fill_per_mfma * 2.4, fill_per_mfma);
    printf("\n");
    fflush(stdout);
}

#define BOTH(K, NF, EV, NAME) run<K, NF, EV>(NAME, 256, out, nblk); run<K, NF, EV>(NAME, 512, out, nblk);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int nblk = prop.multiProcessorCount;
    float *out;
    CHECK(hipMalloc(&out, (size_t)nblk * 512 * 4));
    BOTH(F_NONE, 0, 1, "bare");
    BOTH(F_FMA, 1, 1, "v_fma_f32 x1 per MFMA");
    BOTH(F_FMA, 3, 1, "v_fma_f32 x3 per MFMA");
    BOTH(F_FMA, 8, 1, "v_fma_f32 x8 per MFMA");
    BOTH(F_MUL, 3, 1, "v_mul_f32 x3 per MFMA");
    BOTH(F_PKFMA, 3, 1, "v_pk_fma_f32 x3 per MFMA");
    BOTH(F_PKADD, 3, 1, "v_pk_add_f32 x3 per MFMA");
    BOTH(F_ADDU32, 3, 1, "v_add_u32 x3 per MFMA");
    BOTH(F_XOR, 3, 1, "v_xor_b32 x3 per MFMA");
    BOTH(F_CNDMASK, 3, 1, "v_cndmask_b32 x3 per MFMA");
    BOTH(F_MOV, 3, 1, "v_mov_b32 x3 per MFMA");
    BOTH(F_SALU, 3, 1, "s_add_u32 x3 per MFMA");
    BOTH(F_DSREAD, 1, 2, "ds_read_b128 1 per 2 MFMA");
    BOTH(F_DSREAD, 1, 1, "ds_read_b128 1 per MFMA");
    BOTH(F_DSREAD_C4, 1, 2, "ds_read_b128 4-way cfl 1 per 2");
    BOTH(F_DSREAD_C4, 1, 1, "ds_read_b128 4-way cfl 1 per 1");
    BOTH(F_GLOAD, 1, 4, "global_load_dwordx4 1 per 4 MFMA");
    BOTH(F_GLOAD, 1, 8, "global_load_dwordx4 1 per 8 MFMA");
    BOTH(F_GLDS, 1, 4, "global_load_lds_dwordx4 1 per 4");
    BOTH(F_GLDS, 1, 8, "global_load_lds_dwordx4 1 per 8");
    BOTH(F_GSTORE, 1, 4, "global_store_dwordx4 1 per 4 MFMA");
    BOTH(F_GSTORE, 1, 8, "global_store_dwordx4 1 per 8 MFMA");
    BOTH(F_DSWRITE, 1, 4, "ds_write_b128 1 per 4 MFMA");
    BOTH(F_FMA, 12, 4, "v_fma_f32 x12 after every 4th MFMA");
    BOTH(F_FMA, 24, 8, "v_fma_f32 x24 after every 8th MFMA");
    return 0;
}
