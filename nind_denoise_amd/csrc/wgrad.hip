// Weight gradient of the conv layers on the fp32 matrix cores (training step; reference: nn_common.py:201-218
// `loss.backward()` -- autograd's conv backward-weight).
//
//   G[m][n][t] = sum_p A[m][p] * B[n][p + off(t)]        p = linear pixel of the COMMON grid of A and B
//
//   Conv2d(3)           dW[co][ci][ky][kx] : A = dY (co) re-pitched onto X's grid, B = X (ci),  off = ky*Wb + kx
//   ConvTranspose2d(3)  dW[ci][co][ky][kx] : A = X (ci) re-pitched onto dY's grid, B = dY (co), off = ky*Wb + kx
//   ConvTranspose2d(2,s=2) dW[ci][co][a][b]: four 1-tap problems, B = the (a,b) phase of dY gathered onto X's grid
//   Conv2d(1)           1 tap
// "Re-pitched" = copied onto the other operand's rows x cols with zeros everywhere else (k_repitch below), so K is one
// contiguous pixel range for both operands, every tap is a constant offset, and padding pixels contribute exact zeros.
//
// MFMA: v_mfma_f32_32x32x2_f32 with M = A channels, N = B channels, K = pixels (2 per instruction: lanes 0-31 pixel
// 2s, lanes 32-63 pixel 2s+1).  Operands are single floats per lane read with ds_read_b32 from the quad-planar LDS
// images; the plane stride is padded by 16 B so the 32 channels of a half wave hit 32 different banks.
// Workgroup: tile 64 (M) x 64 (N) x all taps, one K slice.  3x3: 12 waves = (2 x 2 quadrants) x 3 kernel rows, each wave
// owning the three kx accumulator tiles of its row (three waves per SIMD cover each other's barrier / LDS bubbles);
// 1 tap: 4 waves.  K chunks of 62 pixels so that the B image of one kernel row (62 + 2 pixels) is exactly one 64-lane
// LDS-DMA piece per plane.  Partial sums go to [kslice][tap][m][n]; k_wgrad_reduce adds the slices in a fixed order
// (deterministic) and writes the torch weight layout.
#include <stdlib.h>

#include "nd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
    const f32x4 *A, *B;     // plane 0 of the operands (same grid)
    long a_plane, b_plane;  // 16-byte elements per plane
    float *partial;         // [ksplit][taps][Mp][Np]
    int M, N;               // logical channels
    int Mp, Np;             // padded to 64
    int Wb;                 // row pitch of the common grid
    long K;                 // pixels to contract over
    int chunks_per_slice;   // K chunks per workgroup slice
    int nblk;               // N tiles
};

__device__ __forceinline__ void glds16w(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

template <int TAPS>
__global__ __launch_bounds__(TAPS == 9 ? 768 : 256) void k_wgrad(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KY = TAPS == 9 ? 3 : 1;          // kernel rows: one B image each
    constexpr int PC = TAPS == 9 ? 62 : 64;        // pixels per chunk
    constexpr int PLANE = 1024 + 16;               // one 64-pixel piece per plane, padded against bank conflicts
    constexpr int APL = 16, BPL = 16;              // planes of the 64-channel tiles
    constexpr int A_BYTES = APL * PLANE;
    constexpr int STAGE = A_BYTES + KY * BPL * PLANE;
    constexpr int NW = 4 * KY;                     // waves: (2 x 2 quadrants) x kernel rows
    constexpr int TPW = TAPS == 9 ? 3 : 1;         // accumulator tiles per wave: the kx taps of its kernel row

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int ky = wave >> 2;                      // kernel row of this wave (3x3 only)
    const int i = lane & 31, h = lane >> 5;
    const int mb = blockIdx.x / p.nblk, nb = blockIdx.x - mb * p.nblk;
    const int ks = blockIdx.y;

    const long k_begin = (long)ks * p.chunks_per_slice * PC;
    long k_end = k_begin + (long)p.chunks_per_slice * PC;
    if (k_end > p.K) k_end = p.K;
    const int nchunks = k_begin < p.K ? (int)((k_end - k_begin + PC - 1) / PC) : 0;

    // planes of this tile (clamped: rows / columns past M / N are masked at the store)
    const int a_planes = (p.M + 3) / 4, b_planes = (p.N + 3) / 4;
    auto fill = [&](int c, int s) {
        char *sb = smem + s * STAGE;
        const long p0 = k_begin + (long)c * PC;
        for (int q = wave; q < APL + KY * BPL; q += NW) {
            if (q < APL) {
                int pl = mb * 16 + q;
                pl = pl < a_planes ? pl : a_planes - 1;
                glds16w(p.A + (long)pl * p.a_plane + p0 + lane, sb + q * PLANE);
            } else {
                const int r = q - APL, kr = r / BPL;
                int pl = nb * 16 + (r - kr * BPL);
                pl = pl < b_planes ? pl : b_planes - 1;
                glds16w(p.B + (long)pl * p.b_plane + p0 + (long)kr * p.Wb + lane, sb + A_BYTES + r * PLANE);
            }
        }
    };

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // per-lane LDS byte offsets of "my" channel in the A image and in the B image of my kernel row
    const int am = wm * 32 + i, bn = wn * 32 + i;
    const int aoff = (am >> 2) * PLANE + (am & 3) * 4 + h * 16;
    const int boff = A_BYTES + ky * BPL * PLANE + (bn >> 2) * PLANE + (bn & 3) * 4 + h * 16;

    if (nchunks > 0) fill(0, 0);
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + 1 < nchunks) fill(c + 1, (c + 1) & 1);
        const char *sb = smem + (c & 1) * STAGE;
        const long left = k_end - (k_begin + (long)c * PC);
        const int npix = left < PC ? (int)left : PC;     // the last chunk of a slice may be short
        // operands of step s+1 are read while the MFMAs of step s run (explicit double buffer: the LDS latency of a
        // read-then-use schedule would leave the matrix pipe idle a quarter of the time)
        float av[2], bv[2][TPW];
        auto ld = [&](int buf, int s) {
            av[buf] = *(const float *)(sb + aoff + 2 * s * 16);
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) bv[buf][tt] = *(const float *)(sb + boff + (2 * s + tt) * 16);
        };
        ld(0, 0);
#pragma unroll
        for (int s = 0; s < PC / 2; ++s) {
            if (s + 1 < PC / 2) ld((s + 1) & 1, s + 1);
            __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of this step's MFMAs (the scheduler sinks them otherwise)
            const float a = 2 * s + h < npix ? av[s & 1] : 0.f;
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[s & 1][tt], acc[tt], 0, 0, 0);
        }
    }

    // partial[ks][t][m][n]: accumulator register r of lane (col n, half h) is row (r&3) + 8*(r>>2) + 4h
    const int n = nb * 64 + wn * 32 + i;
    float *out = p.partial + (long)ks * TAPS * p.Mp * p.Np;
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int t = ky * TPW + tt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[((long)t * p.Mp + m) * p.Np + n] = acc[tt][r];
        }
    }
}

// dW[(m*N + n)*taps_total + tap0 + t] = sum_ks partial[ks][t][m][n]
// One workgroup per 64 consecutive (t, m, n) outputs; its four waves each add every fourth slice (in order), then the
// four sums are added in wave order: a fixed summation tree, so the result is deterministic.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int ksplit, int taps, int Mp, int Np,
                                                      int M, int N, int taps_total, int tap0, float *__restrict__ dw) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + lane;
    const long total = (long)taps * M * N;
    const bool ok = idx < total;
    const long ic = ok ? idx : total - 1;
    const int n = (int)(ic % N);
    const int m = (int)((ic / N) % M);
    const int t = (int)(ic / ((long)N * M));
    const long slice = (long)taps * Mp * Np;
    const float *src = partial + ((long)t * Mp + m) * Np + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = w;
    for (; k + 12 < ksplit; k += 16) {
        const float v0 = src[(long)k * slice], v1 = src[(long)(k + 4) * slice], v2 = src[(long)(k + 8) * slice],
                    v3 = src[(long)(k + 12) * slice];
        s0 += v0;
        s1 += v1;
        s2 += v2;
        s3 += v3;
    }
    for (; k < ksplit; k += 4) s0 += src[(long)k * slice];
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && ok) dw[((long)m * N + n) * taps_total + tap0 + t] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// dst (C planes on its own grid) = zeros, except  dst[c][img][y + oy][x + ox] = src[c][img][y*ss + sy][x*ss + sx]
// for 0 <= y < h, 0 <= x < w   (coordinates without borders; the kernel adds each buffer's own pad)
__global__ void k_repitch(const f32x4 *__restrict__ src, long snp, int sHb, int sWb, int spad, int ss, int sy, int sx,
                          f32x4 *__restrict__ dst, long dnp, int dHb, int dWb, int oy, int ox, int h, int w, int B) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z % B, q = blockIdx.z / B;
    if (x >= dWb) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int yy = y - oy, xx = x - ox;
    if (yy >= 0 && yy < h && xx >= 0 && xx < w)
        v = src[(long)q * snp + ((long)b * sHb + yy * ss + sy + spad) * sWb + xx * ss + sx + spad];
    dst[(long)q * dnp + ((long)b * dHb + y) * dWb + x] = v;
}

int nd_launch_repitch(const QpBuf &src, int src_plane0, int planes, int ss, int sy, int sx, const QpBuf &dst, int oy,
                      int ox, int h, int w, hipStream_t s) {
    if (src.dt != ND_F32 || dst.dt != ND_F32) ND_FAIL(ND_EINVAL, "repitch: fp32 buffers only");
    if (dst.B != src.B || dst.planes < planes) ND_FAIL(ND_EINVAL, "repitch: destination too small");
    dim3 grid((dst.Wb + 127) / 128, dst.Hb, src.B * planes);
    hipLaunchKernelGGL(k_repitch, grid, dim3(128), 0, s, (const f32x4 *)src.base + (long)src_plane0 * src.np(), src.np(),
                       src.Hb, src.Wb, src.pad, ss, sy, sx, (f32x4 *)dst.base, dst.np(), dst.Hb, dst.Wb, oy, ox, h, w, src.B);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

size_t nd_wgrad_partial_floats(int taps, int M, int N, long K, int *ksplit_out, int *cps_out) {
    const int PC = taps == 9 ? 62 : 64;
    const int Mp = (M + 63) / 64 * 64, Np = (N + 63) / 64 * 64;
    const long chunks = (K + PC - 1) / PC;
    const long tiles = (long)(Mp / 64) * (Np / 64);
    long ksplit = 512 / tiles;                          // <= 2 whole rounds of one workgroup per CU (LDS: one fits)
    if (ksplit > chunks) ksplit = chunks;
    if (ksplit < 1) ksplit = 1;
    const long cps = (chunks + ksplit - 1) / ksplit;
    ksplit = (chunks + cps - 1) / cps;
    if (ksplit_out) *ksplit_out = (int)ksplit;
    if (cps_out) *cps_out = (int)cps;
    return (size_t)ksplit * taps * Mp * Np;
}

// A, B: same grid (B, Hb, Wb, pad ignored: the caller re-pitched);  dw: torch layout, taps_total entries per (m, n)
int nd_launch_wgrad(const QpBuf &A, int a_plane0, int M, const QpBuf &Bq, int b_plane0, int N, int taps, int taps_total,
                    int tap0, float *partial, size_t partial_floats, float *dw, hipStream_t s) {
    if (taps != 9 && taps != 1) ND_FAIL(ND_EINVAL, "wgrad: taps must be 9 or 1");
    if (A.dt != ND_F32 || Bq.dt != ND_F32) ND_FAIL(ND_EINVAL, "wgrad: fp32 buffers only");
    if (A.B != Bq.B || A.Hb != Bq.Hb || A.Wb != Bq.Wb) ND_FAIL(ND_EINVAL, "wgrad: operands are not on one grid");
    const long K = Bq.used();
    int ksplit, cps;
    const size_t need = nd_wgrad_partial_floats(taps, M, N, K, &ksplit, &cps);
    if (partial_floats < need) ND_FAIL(ND_ENOMEM, "wgrad: partial buffer %zu floats given, %zu needed", partial_floats, need);
    WgradParams p;
    p.A = (const f32x4 *)A.base + (long)a_plane0 * A.np();
    p.B = (const f32x4 *)Bq.base + (long)b_plane0 * Bq.np();
    p.a_plane = A.np();
    p.b_plane = Bq.np();
    p.partial = partial;
    p.M = M;
    p.N = N;
    p.Mp = (M + 63) / 64 * 64;
    p.Np = (N + 63) / 64 * 64;
    p.Wb = Bq.Wb;
    p.K = K;
    p.chunks_per_slice = cps;
    p.nblk = p.Np / 64;
    const int lds = 2 * ((16 + (taps == 9 ? 3 : 1) * 16) * (1024 + 16));
    dim3 grid((p.Mp / 64) * p.nblk, ksplit);
    static std::atomic<bool> lds_set[16][2];   // per device: function attributes belong to the device's copy of the code object
    int dev = 0;
    ND_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) ND_FAIL(ND_EINVAL, "wgrad: device index %d", dev);
    if (!lds_set[dev][taps == 9].load(std::memory_order_relaxed)) {
        const void *fn = taps == 9 ? (const void *)k_wgrad<9> : (const void *)k_wgrad<1>;
        ND_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        lds_set[dev][taps == 9].store(true, std::memory_order_relaxed);
    }
    if (taps == 9)
        hipLaunchKernelGGL(k_wgrad<9>, grid, dim3(768), lds, s, p);
    else
        hipLaunchKernelGGL(k_wgrad<1>, grid, dim3(256), lds, s, p);
    ND_HIP(hipGetLastError());
    const long total = (long)taps * M * N;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, partial, ksplit, taps, p.Mp,
                       p.Np, M, N, taps_total, tap0, dw);
    ND_HIP(hipGetLastError());
    return ND_OK;
}
