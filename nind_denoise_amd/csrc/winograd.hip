// Winograd F(T x T, 3 x 3) for the wide 3x3 layers (inference, fp32): T = 6 in the network (64 positions, 5.06x fewer MACs than the
// direct form; V / M are 1.78x the size of X / Y); T = 2 (16 positions, 2.25x) and T = 4 (36 positions, 4x) remain for the layer entry
// point and the tests.  Reference op: the Conv2d(3) / ConvTranspose2d(3) layers of networks/UtNet.py:27-88; same math, other association:
//     Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        per (T+2) x (T+2) input tile d -> T x T output tile Y
// Three passes over HBM, because a fused kernel does not fit the LDS / register budget of one CU:
//   1. input transform   X (quad-planar, bordered)         -> V[pos][Cin/4][tile]   (HBM-bound: reads X once, writes (T+2)^2/T^2 x |X|)
//   2. conv_qp 1-tap     P = (T+2)^2 independent GEMMs in ONE launch: M[pos] = U[pos] (Cout x Cin) * V[pos]   (MFMA-bound)
//   3. output transform  M[pos][Cout/4][tile] -> A^T m A + bias, activation (+ fused 2x2 max pool) -> destination buffer (HBM-bound)
// T = 6: k_wino_in2 / k_wino_out2 (a tile shared by 8 threads through LDS); T = 2 | 4: k_wino_input / k_wino_output (one thread per tile).
// A layer may be restricted to a region of its output (ConvDesc::roi_*: the same passes on shifted base pointers).
// ConvTranspose2d(3) is the same valid correlation on its zero-bordered input with flipped / transposed weights (as in pack.hip).
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "nd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// tile index -> (column, row, image); the launcher keeps the tile count below 2^31, so one 32-bit division pair instead of three
// 64-bit ones (a 64-bit division is ~100 VALU instructions per thread)
__device__ __forceinline__ void tile_pos(long t, int TX, int TY, int &tx, int &ty, int &b) {
    const unsigned u = (unsigned)t, row = u / (unsigned)TX;
    tx = (int)(u - row * (unsigned)TX);
    b = (int)(row / (unsigned)TY);
    ty = (int)(row - (unsigned)b * (unsigned)TY);
}

template <int T> struct Wino;
template <> struct Wino<2> {
    static constexpr int A = 4;
    template <typename V> __host__ __device__ static void bt(const V *d, V *o) {   // B^T d
        o[0] = d[0] - d[2];
        o[1] = d[1] + d[2];
        o[2] = d[2] - d[1];
        o[3] = d[1] - d[3];
    }
    template <typename V> __host__ __device__ static void at(const V *m, V *o) {   // A^T m
        o[0] = m[0] + m[1] + m[2];
        o[1] = m[1] - m[2] - m[3];
    }
    static void g(const double *w, double *o) {   // G g
        o[0] = w[0];
        o[1] = 0.5 * (w[0] + w[1] + w[2]);
        o[2] = 0.5 * (w[0] - w[1] + w[2]);
        o[3] = w[2];
    }
};
template <> struct Wino<4> {
    static constexpr int A = 6;
    template <typename V> __host__ __device__ static void bt(const V *d, V *o) {
        o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
        o[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
        o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
        o[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
        o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
        o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
    }
    template <typename V> __host__ __device__ static void at(const V *m, V *o) {
        o[0] = m[0] + m[1] + m[2] + m[3] + m[4];
        o[1] = m[1] - m[2] + 2.f * (m[3] - m[4]);
        o[2] = m[1] + m[2] + 4.f * (m[3] + m[4]);
        o[3] = m[1] - m[2] + 8.f * (m[3] - m[4]) + m[5];
    }
    static void g(const double *w, double *o) {
        o[0] = w[0] / 4;
        o[1] = -(w[0] + w[1] + w[2]) / 6;
        o[2] = -(w[0] - w[1] + w[2]) / 6;
        o[3] = w[0] / 24 + w[1] / 12 + w[2] / 6;
        o[4] = w[0] / 24 - w[1] / 12 + w[2] / 6;
        o[5] = w[2];
    }
};

// F(6,3): points 0, +-1, +-2, +-1/2, inf.  (8x8)/(6x6) = 1.78x the bytes of X in V (F(4,3): 2.25x) and 64 MACs per 36 outputs
// (5.06x fewer than direct; F(4x4): 4x).  fp32 error on UtNet-scale data: ~2x that of F(4x4) (9e-6 against 4.6e-6 on O(1)
// outputs of a 128-channel layer) -- the larger constants (21/4, 32) cost one more bit, not an order of magnitude.
template <> struct Wino<6> {
    static constexpr int A = 8;
    template <typename V> __host__ __device__ static void bt(const V *d, V *o) {
        o[0] = d[0] - d[6] + 5.25f * (d[4] - d[2]);
        o[7] = d[7] - d[1] + 5.25f * (d[3] - d[5]);
        V p = d[2] + d[6] - 4.25f * d[4], q = d[1] + d[5] - 4.25f * d[3];
        o[1] = p + q;
        o[2] = p - q;
        p = 0.25f * d[2] - 1.25f * d[4] + d[6];
        q = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
        o[3] = p + q;
        o[4] = p - q;
        p = 4.f * d[2] - 5.f * d[4] + d[6];
        q = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
        o[5] = p + q;
        o[6] = p - q;
    }
    template <typename V> __host__ __device__ static void at(const V *m, V *o) {
        const V s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4], s3 = m[5] + m[6], d3 = m[5] - m[6];
        o[0] = m[0] + s1 + s2 + s3;
        o[1] = d1 + 2.f * d2 + 0.5f * d3;
        o[2] = s1 + 4.f * s2 + 0.25f * s3;
        o[3] = d1 + 8.f * d2 + 0.125f * d3;
        o[4] = s1 + 16.f * s2 + 0.0625f * s3;
        o[5] = d1 + 32.f * d2 + 0.03125f * d3 + m[7];
    }
    static void g(const double *w, double *o) {
        o[0] = w[0];
        o[1] = -2.0 / 9 * (w[0] + w[1] + w[2]);
        o[2] = -2.0 / 9 * (w[0] - w[1] + w[2]);
        o[3] = w[0] / 90 + w[1] / 45 + w[2] * 2 / 45;
        o[4] = w[0] / 90 - w[1] / 45 + w[2] * 2 / 45;
        o[5] = w[0] * 32 / 45 + w[1] * 16 / 45 + w[2] * 8 / 45;
        o[6] = w[0] * 32 / 45 - w[1] * 16 / 45 + w[2] * 8 / 45;
        o[7] = w[2];
    }
};

// V[pos][plane][tile] = B^T d B.  grid (ceil(tiles / 256), planes)
template <int T>
__global__ __launch_bounds__(256) void k_wino_input(const f32x4 *__restrict__ x, long xnp, int Hb, int Wb, int B, int TY, int TX,
                                                    f32x4 *__restrict__ v, long vnp, long vbs) {
    constexpr int A = Wino<T>::A;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long tiles = (long)B * TY * TX;
    if (t >= tiles) return;
    const int q = blockIdx.y;
    int tx, ty, b;
    tile_pos(t, TX, TY, tx, ty, b);
    const f32x4 *src = x + (long)q * xnp + ((long)b * Hb + T * ty) * Wb + T * tx;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[A][A], r[A][A];
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j) d[i][j] = (T * ty + i < Hb && T * tx + j < Wb) ? src[(long)i * Wb + j] : zero;
    // columns: r[:, j] = B^T d[:, j]
#pragma unroll
    for (int j = 0; j < A; ++j) {
        f32x4 c[A], o[A];
#pragma unroll
        for (int i = 0; i < A; ++i) c[i] = d[i][j];
        Wino<T>::bt(c, o);
#pragma unroll
        for (int i = 0; i < A; ++i) r[i][j] = o[i];
    }
    // rows: V[i, :] = B^T r[i, :]   (= r B)
    f32x4 *dst = v + (long)q * vnp + t;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        f32x4 o[A];
        Wino<T>::bt(r[i], o);
#pragma unroll
        for (int j = 0; j < A; ++j) dst[(long)(i * A + j) * vbs] = o[j];
    }
}

// out = act(A^T m A + bias).  grid (ceil(tiles / 128), Cout / 4), 128 threads.
// A thread owns one tile (T x T pixels), but stores go out PIXEL-contiguous across the lanes: the workgroup's 128 tiles are
// consecutive in x, so row r of all of them is staged in LDS and written as whole 1 KiB wave stores instead of 16-byte
// pieces at a 64-byte stride (partial cache lines that the L2 has to merge: measured 3.3 TB/s before)
template <int T>
__global__ __launch_bounds__(128) void k_wino_output(const f32x4 *__restrict__ m, long mnp, long mbs, int B, int TY, int TX, int Hv,
                                                     int Wv, const float *__restrict__ bias, int act, float slope_imm,
                                                     const float *__restrict__ slope_dev, f32x4 *__restrict__ out, long onp,
                                                     int out_plane0, int Ho, int Wo, int opad, f32x4 *__restrict__ pool, long pnp,
                                                     int Hp, int Wp, int ppad) {
    constexpr int A = Wino<T>::A;
    constexpr int NT = 128;
    __shared__ f32x4 sm[T][NT * T];
    const long t0 = (long)blockIdx.x * NT;
    const long t = t0 + threadIdx.x;
    const long tiles = (long)B * TY * TX;
    const int q = blockIdx.y;
    const f32x4 bv = *(const f32x4 *)(bias + 4 * q);
    const float slope = act == ND_ACT_NONE ? 1.f : (slope_dev ? *slope_dev : slope_imm);
    if (t < tiles) {
        const f32x4 *src = m + (long)q * mnp + t;
        f32x4 r[A][T], yv[T][T];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            f32x4 row[A];
#pragma unroll
            for (int j = 0; j < A; ++j) row[j] = src[(long)(i * A + j) * mbs];
            Wino<T>::at(row, r[i]);   // m A  (row-wise A^T)
        }
#pragma unroll
        for (int j = 0; j < T; ++j) {
            f32x4 c[A], o[T];
#pragma unroll
            for (int i = 0; i < A; ++i) c[i] = r[i][j];
            Wino<T>::at(c, o);
#pragma unroll
            for (int i = 0; i < T; ++i) {
                f32x4 y = o[i] + bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = y[e];
                    switch (act) {
                        case ND_ACT_ELU: y[e] = u > 0.f ? u : expm1f(u); break;
                        case ND_ACT_HARDSWISH: y[e] = u * fminf(fmaxf(u + 3.f, 0.f), 6.f) / 6.f; break;
                        default: y[e] = u > 0.f ? u : u * slope; break;   // PReLU; "none" is slope 1
                    }
                }
                sm[i][threadIdx.x * T + j] = y;
                yv[i][j] = y;
            }
        }
        if (pool) {   // fused MaxPool2d(2): a tile's T x T outputs hold (T/2)^2 whole 2x2 blocks (T even, tiles start on even pixels)
            int tx, ty, b;
    tile_pos(t, TX, TY, tx, ty, b);
            f32x4 *pd = pool + (long)q * pnp + (long)b * Hp * Wp;
#pragma unroll
            for (int a = 0; a < T / 2; ++a)
#pragma unroll
                for (int c = 0; c < T / 2; ++c) {
                    const int py = ty * (T / 2) + a, px = tx * (T / 2) + c;
                    if (2 * py + 1 < Hv && 2 * px + 1 < Wv) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = fmaxf(fmaxf(yv[2 * a][2 * c][e], yv[2 * a][2 * c + 1][e]), fmaxf(yv[2 * a + 1][2 * c][e], yv[2 * a + 1][2 * c + 1][e]));
                        pd[(long)(py + ppad) * Wp + px + ppad] = v;
                    }
                }
        }
    }
    __syncthreads();
    f32x4 *dst = out + (long)(out_plane0 + q) * onp;
    for (int p = threadIdx.x; p < NT * T; p += NT) {
        const long tt = t0 + p / T;
        if (tt >= tiles) break;
        const int j = p % T;
        int tx, ty, b;
        tile_pos(tt, TX, TY, tx, ty, b);
        const int x = T * tx + j;
        if (x >= Wv) continue;
#pragma unroll
        for (int i = 0; i < T; ++i)
            if (T * ty + i < Hv) dst[((long)b * Ho + T * ty + i + opad) * Wo + x + opad] = sm[i][p];
    }
}

// The same two transforms with the tile shared by A threads through LDS (T = 6: a thread of the kernels above would hold 64 + 64
// float4 = 512 VGPRs).  32 tiles per workgroup, 32 * A threads.
//   input : pass 1 thread (tile, column j)  loads d[0..A)[j] (A lanes = 16 A contiguous bytes per tile row), column transform -> LDS
//           pass 2 thread (row i, tile)     row transform of r[i][0..A), A position stores, each 512 B contiguous per half wave
template <int T, int NTL>
__global__ __launch_bounds__(NTL * (T + 2)) void k_wino_in2(const f32x4 *__restrict__ x, long xnp, int Hb, int Wb, long img_stride, int row_stride, int B,
                                                            int TY, int TX, f32x4 *__restrict__ v, long vnp, long vbs) {
    // Hb x Wb: extent of the (view of the) bordered input a tile may read; img_stride / row_stride: of the buffer it lives in
    constexpr int A = Wino<T>::A, RS = A * A + 1;   // (+1: pass 2 reads a tile per lane, 16 (A*A+1) B apart: all banks)
    __shared__ f32x4 sm[NTL * RS];
    const long tiles = (long)B * TY * TX;
    const long t0 = (long)blockIdx.x * NTL;
    const int q = blockIdx.y;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {
        const int tl = threadIdx.x / A, j = threadIdx.x % A;
        const long t = t0 + tl;
        if (t < tiles) {
            int tx, ty, b;
    tile_pos(t, TX, TY, tx, ty, b);
            const f32x4 *src = x + (long)q * xnp + (long)b * img_stride + (long)(T * ty) * row_stride + T * tx + j;
            f32x4 d[A], o[A];
            const bool okj = T * tx + j < Wb;
#pragma unroll
            for (int i = 0; i < A; ++i) d[i] = (okj && T * ty + i < Hb) ? src[(long)i * row_stride] : zero;
            Wino<T>::bt(d, o);
#pragma unroll
            for (int i = 0; i < A; ++i) sm[tl * RS + i * A + j] = o[i];
        }
    }
    __syncthreads();
    {
        const int i = threadIdx.x / NTL, tl = threadIdx.x % NTL;
        const long t = t0 + tl;
        if (t < tiles) {
            f32x4 r[A], o[A];
#pragma unroll
            for (int j = 0; j < A; ++j) r[j] = sm[tl * RS + i * A + j];
            Wino<T>::bt(r, o);
            f32x4 *dst = v + (long)q * vnp + t;
#pragma unroll
            for (int j = 0; j < A; ++j) dst[(long)(i * A + j) * vbs] = o[j];
        }
    }
}

//   output: pass 1 thread (column j, tile)  loads m[0..A)[j] (tile-contiguous), column transform (A -> T) -> LDS
//           pass 2 thread (row i < T, tile) row transform, bias, activation -> LDS row image [T][32 tiles x T pixels]
//           pass 3 all threads              pixel-contiguous stores of the T rows (+ the 2x2 max pool of the tile rows, T even)
template <int T, int NTL>
__global__ __launch_bounds__(NTL * (T + 2)) void k_wino_out2(const f32x4 *__restrict__ m, long mnp, long mbs, int B, int TY, int TX, int Hv,
                                                             int Wv, const float *__restrict__ bias, int act, float slope_imm,
                                                             const float *__restrict__ slope_dev, f32x4 *__restrict__ out, long onp,
                                                             int out_plane0, long out_img_stride, int Wo, int opad, f32x4 *__restrict__ pool, long pnp,
                                                             int Hp, int Wp, int ppad) {
    constexpr int A = Wino<T>::A, RS = T * A + 1, NTH = NTL * A;
    __shared__ f32x4 sr[NTL * RS];
    __shared__ f32x4 so[T][NTL * T];
    const long tiles = (long)B * TY * TX;
    const long t0 = (long)blockIdx.x * NTL;
    const int q = blockIdx.y;
    const f32x4 bv = *(const f32x4 *)(bias + 4 * q);
    const float slope = act == ND_ACT_NONE ? 1.f : (slope_dev ? *slope_dev : slope_imm);
    {
        const int j = threadIdx.x / NTL, tl = threadIdx.x % NTL;
        const long t = t0 + tl;
        if (t < tiles) {
            const f32x4 *src = m + (long)q * mnp + t;
            f32x4 c[A], o[T];
#pragma unroll
            for (int i = 0; i < A; ++i) c[i] = src[(long)(i * A + j) * mbs];
            Wino<T>::at(c, o);
#pragma unroll
            for (int i = 0; i < T; ++i) sr[tl * RS + i * A + j] = o[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < NTL * T) {
        const int i = threadIdx.x / NTL, tl = threadIdx.x % NTL;
        if (t0 + tl < tiles) {
            f32x4 r[A], o[T];
#pragma unroll
            for (int j = 0; j < A; ++j) r[j] = sr[tl * RS + i * A + j];
            Wino<T>::at(r, o);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                f32x4 y = o[j] + bv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = y[e];
                    switch (act) {
                        case ND_ACT_ELU: y[e] = u > 0.f ? u : expm1f(u); break;
                        case ND_ACT_HARDSWISH: y[e] = u * fminf(fmaxf(u + 3.f, 0.f), 6.f) / 6.f; break;
                        default: y[e] = u > 0.f ? u : u * slope; break;   // PReLU; "none" is slope 1
                    }
                }
                so[i][tl * T + j] = y;
            }
        }
    }
    __syncthreads();
    f32x4 *dst = out + (long)(out_plane0 + q) * onp;
    for (int idx = threadIdx.x; idx < T * NTL * T; idx += NTH) {
        const int i = idx / (NTL * T), pp = idx - i * (NTL * T);
        const long tt = t0 + pp / T;
        if (tt >= tiles) continue;
        int tx, ty, b;
        tile_pos(tt, TX, TY, tx, ty, b);
        const int xx = T * tx + pp % T, yy = T * ty + i;
        if (xx < Wv && yy < Hv) dst[(long)b * out_img_stride + (long)(yy + opad) * Wo + xx + opad] = so[i][pp];
    }
    if (pool) {   // fused MaxPool2d(2): tiles start on even pixels and T is even, so every 2x2 block lies inside one tile
        f32x4 *pd = pool + (long)q * pnp;
        for (int idx = threadIdx.x; idx < (T / 2) * NTL * (T / 2); idx += NTH) {
            const int a = idx / (NTL * (T / 2)), pp = idx - a * (NTL * (T / 2));
            const int tl = pp / (T / 2), c = pp % (T / 2);
            const long tt = t0 + tl;
            if (tt >= tiles) continue;
            int tx, ty, b;
        tile_pos(tt, TX, TY, tx, ty, b);
            const int px = tx * (T / 2) + c, py = ty * (T / 2) + a;
            if (2 * px + 1 < Wv && 2 * py + 1 < Hv) {
                const f32x4 p00 = so[2 * a][tl * T + 2 * c], p01 = so[2 * a][tl * T + 2 * c + 1];
                const f32x4 p10 = so[2 * a + 1][tl * T + 2 * c], p11 = so[2 * a + 1][tl * T + 2 * c + 1];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(p00[e], p01[e]), fmaxf(p10[e], p11[e]));
                pd[((long)b * Hp + py + ppad) * Wp + px + ppad] = v;
            }
        }
    }
}

constexpr int kOutNtl = 32;   // tiles per workgroup of the F(6x6) output transform (16: 21.8 KB of LDS, 32: 43.5 KB; measured equal within 1 %)
int positions(int T) { return (T + 2) * (T + 2); }
size_t gemm_floats(int cin, int cout) { return nd_packed_floats(ND_CONV1, cin, cout, ND_F32); }

struct WinoGeo {
    int TY, TX, Hv, Wv;
    long tiles;          // B * TY * TX
    long vnp, mnp;       // plane strides (16-byte elements, with slack for the GEMM's DMA over-read)
    long vbs, mbs;       // position strides
    size_t v_bytes, m_bytes;
};
WinoGeo wino_geo(int T, const QpBuf &in, int cin, int cout, int roi_rows = 0, int roi_cols = 0) {
    WinoGeo g;
    g.Hv = roi_rows > 0 ? roi_rows : in.Hb - 2;
    g.Wv = roi_rows > 0 ? roi_cols : in.Wb - 2;
    g.TY = (g.Hv + T - 1) / T;
    g.TX = (g.Wv + T - 1) / T;
    g.tiles = (long)in.B * g.TY * g.TX;
    g.vnp = g.mnp = g.tiles;
    const long slack = 4096;
    g.vbs = (long)(2 * nd_kblocks(cin)) * g.vnp + slack;
    g.mbs = (long)((cout + 3) / 4) * g.mnp + slack;
    g.v_bytes = (size_t)positions(T) * g.vbs * 16;
    g.m_bytes = (size_t)positions(T) * g.mbs * 16;
    return g;
}
}  // namespace

// ------------------------------------------------------------------ packing (host)
// layout: P x [1-tap packed GEMM weights U_pos (pack.hip layout, zero bias)] + bias[cout rounded up to 4]
size_t nd_wino_packed_floats(int T, int cin, int cout) {
    return (size_t)positions(T) * gemm_floats(cin, cout) + (size_t)(cout + 3) / 4 * 4;
}

// w: torch layout of the layer (Conv2d: [cout][cin][3][3]; ConvTranspose2d: [cin][cout][3][3]); bias may be null
int nd_wino_pack(int T, int kind, int cin, int cout, const float *w, const float *bias, float *packed) {
    if (T != 2 && T != 4 && T != 6) ND_FAIL(ND_EINVAL, "winograd: tile must be 2, 4 or 6");
    if (kind != ND_CONV3 && kind != ND_CONVT3) ND_FAIL(ND_EINVAL, "winograd: 3x3 layers only");
    const int A = T + 2, P = A * A;
    std::vector<float> u((size_t)P * cout * cin);
    // host threads: the transform + repack of UtNet(64)'s 13 Winograd layers is ~2 s on one core (model load time)
    auto parallel_for = [](int n, auto &&fn) {
        const int nt = std::max(1, std::min(n, std::min(16, (int)std::thread::hardware_concurrency())));
        std::atomic<int> next(0);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&] {
                for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
            });
        for (auto &x : th) x.join();
    };
    parallel_for(cout, [&](int co) {
        for (int ci = 0; ci < cin; ++ci) {
            double g[3][3], tmp[8][3], U[8][8];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    g[ky][kx] = kind == ND_CONV3 ? w[(((size_t)co * cin + ci) * 3 + ky) * 3 + kx]
                                                 : w[(((size_t)ci * cout + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
            for (int kx = 0; kx < 3; ++kx) {   // columns: tmp[:, kx] = G g[:, kx]
                double c[3] = {g[0][kx], g[1][kx], g[2][kx]}, o[8];
                if (T == 2) Wino<2>::g(c, o); else if (T == 4) Wino<4>::g(c, o); else Wino<6>::g(c, o);
                for (int i = 0; i < A; ++i) tmp[i][kx] = o[i];
            }
            for (int i = 0; i < A; ++i) {      // rows: U[i, :] = G tmp[i, :]
                double o[8];
                if (T == 2) Wino<2>::g(tmp[i], o); else if (T == 4) Wino<4>::g(tmp[i], o); else Wino<6>::g(tmp[i], o);
                for (int j = 0; j < A; ++j) U[i][j] = o[j];
            }
            for (int i = 0; i < A; ++i)
                for (int j = 0; j < A; ++j) u[((size_t)(i * A + j) * cout + co) * cin + ci] = (float)U[i][j];
        }
    });
    const size_t gf = gemm_floats(cin, cout);
    parallel_for(P, [&](int p) { nd_pack_layer(ND_CONV1, cin, cout, ND_F32, u.data() + (size_t)p * cout * cin, nullptr, packed + p * gf); });
    float *b = packed + (size_t)P * gf;
    for (int co = 0; co < (cout + 3) / 4 * 4; ++co) b[co] = (bias && co < cout) ? bias[co] : 0.f;
    return ND_OK;
}

// ------------------------------------------------------------------ launch
size_t nd_wino_scratch_bytes(int T, const QpBuf &in, int cin, int cout) {
    const WinoGeo g = wino_geo(T, in, cin, cout);
    return ((g.v_bytes + 255) & ~(size_t)255) + ((g.m_bytes + 255) & ~(size_t)255);
}

// d: the layer as for nd_launch_conv (kind CONV3 / CONVT3, fp32); d.wpk = nd_wino_pack blob.  scratch: nd_wino_scratch_bytes
void nd_wino_xform_bytes(int T, const QpBuf &in, int cin, int cout, double *bytes_in, double *bytes_out) {
    const WinoGeo g = wino_geo(T, in, cin, cout);
    const double P = positions(T), px_in = (double)in.B * in.Hb * in.Wb, px_out = (double)in.B * g.Hv * g.Wv;
    *bytes_in = 4.0 * cin * px_in + 4.0 * cin * P * (double)g.tiles;
    *bytes_out = 4.0 * cout * P * (double)g.tiles + 4.0 * cout * px_out;
}

int nd_launch_conv_wino(int T, const ConvDesc &d, void *scratch, size_t scratch_bytes, hipStream_t s, hipEvent_t *ev2) {
    if (T != 2 && T != 4 && T != 6) ND_FAIL(ND_EINVAL, "winograd: tile must be 2, 4 or 6");
    if ((d.kind != ND_CONV3 && d.kind != ND_CONVT3) || d.in.dt != ND_F32 || d.out.dt != ND_F32)
        ND_FAIL(ND_EINVAL, "winograd: fp32 3x3 layers only");
    if (d.cin % 16 || d.cout % 4) ND_FAIL(ND_EINVAL, "winograd: Cin must be a multiple of 16, Cout of 4 (got %d, %d)", d.cin, d.cout);
    if (d.pre) ND_FAIL(ND_EINVAL, "winograd: inference only (no pre-activation copy)");
    const bool roi = d.roi_rows > 0;
    if (roi && (T != 6 || d.pool || d.roi_r0 < 0 || d.roi_c0 < 0 || d.roi_cols < 1 || d.roi_r0 + d.roi_rows > d.in.Hb - 2 || d.roi_c0 + d.roi_cols > d.in.Wb - 2))
        ND_FAIL(ND_EINVAL, "winograd: region [%d,+%d) x [%d,+%d) outside the output (F(6x6), unpooled layers only)", d.roi_r0, d.roi_rows, d.roi_c0, d.roi_cols);
    const WinoGeo g = wino_geo(T, d.in, d.cin, d.cout, d.roi_rows, d.roi_cols);
    if (d.out.Hb != d.in.Hb - 2 + 2 * d.out.pad || d.out.Wb != d.in.Wb - 2 + 2 * d.out.pad || d.out.B != d.in.B)
        ND_FAIL(ND_EINVAL, "winograd: destination does not fit the result");
    // a region is the same three passes on shifted base pointers (input view: rows [r0, r0 + rows + 2) of the bordered buffer)
    const long roi_in = roi ? (long)d.roi_r0 * d.in.Wb + d.roi_c0 : 0, roi_out = roi ? (long)d.roi_r0 * d.out.Wb + d.roi_c0 : 0;
    const int vHb = g.Hv + 2, vWb = g.Wv + 2;
    const size_t need = nd_wino_scratch_bytes(T, d.in, d.cin, d.cout);
    if (!scratch || scratch_bytes < need) ND_FAIL(ND_ENOMEM, "winograd: scratch %zu B given, %zu B needed", scratch_bytes, need);
    if (g.tiles >= (1L << 31)) ND_FAIL(ND_EINVAL, "winograd: %ld tiles exceed the 32-bit tile index", g.tiles);
    const int P = positions(T);
    f32x4 *v = (f32x4 *)scratch;
    f32x4 *m = (f32x4 *)((char *)scratch + ((g.v_bytes + 255) & ~(size_t)255));
    const int in_planes = 2 * nd_kblocks(d.cin), out_planes = d.cout / 4;
    const f32x4 *x = (const f32x4 *)d.in.base + (long)d.in_plane0 * d.in.np() + roi_in;
    dim3 gi((unsigned)((g.tiles + 255) / 256), in_planes);
    // tiles per workgroup of the LDS-shared input transform: 16 (measured on UtNet(64) at cs = 264: 16 and 64 equal on every layer;
    // 32 is 50 % slower on the 132 x 132 input of tconvs3.0 -- 3.45 against 2.27 ms -- and equal elsewhere)
    constexpr int kNtl = 16;
    dim3 gi2((unsigned)((g.tiles + kNtl - 1) / kNtl), in_planes);
    if (T == 2)
        hipLaunchKernelGGL(k_wino_input<2>, gi, dim3(256), 0, s, x, d.in.np(), d.in.Hb, d.in.Wb, d.in.B, g.TY, g.TX, v, g.vnp, g.vbs);
    else if (T == 6)
        hipLaunchKernelGGL((k_wino_in2<6, kNtl>), gi2, dim3(kNtl * 8), 0, s, x, d.in.np(), vHb, vWb, (long)d.in.Hb * d.in.Wb, d.in.Wb, d.in.B, g.TY, g.TX, v, g.vnp, g.vbs);
    else
        hipLaunchKernelGGL(k_wino_input<4>, gi, dim3(256), 0, s, x, d.in.np(), d.in.Hb, d.in.Wb, d.in.B, g.TY, g.TX, v, g.vnp, g.vbs);
    ND_HIP(hipGetLastError());
    if (ev2) ND_HIP(hipEventRecord(ev2[0], s));

    ConvDesc e;
    e.kind = ND_CONV1;
    e.act = ND_ACT_NONE;
    e.slope = 1.f;
    e.slope_dev = nullptr;
    e.cin = d.cin;
    e.cout = d.cout;
    e.wpk = d.wpk;
    e.bias = d.wpk + (size_t)nd_mtiles(ND_CONV1, d.cout) * nd_kblocks(d.cin) * 256;   // the zero bias of position 0
    e.in.base = (float *)v;
    e.in.planes = in_planes;
    e.in.B = d.in.B;
    e.in.Hb = g.TY;
    e.in.Wb = g.TX;
    e.in.pad = 0;
    e.in.pstride = g.vnp;
    e.in.dt = ND_F32;
    e.out = e.in;
    e.out.base = (float *)m;
    e.out.planes = out_planes;
    e.out.pstride = g.mnp;
    e.out_plane0 = 0;
    e.variant = nd_conv_variant_gemm(d.cin, d.cout);
    e.part = d.part;
    e.part_bytes = d.part_bytes;
    e.nosplit = d.nosplit;
    e.nbatch = P;
    e.in_bs = g.vbs;
    e.out_bs = g.mbs;
    e.w_bs = gemm_floats(d.cin, d.cout);
    ND_TRY(nd_launch_conv(e, s));
    if (ev2) ND_HIP(hipEventRecord(ev2[1], s));

    const float *bias = d.wpk + (size_t)P * gemm_floats(d.cin, d.cout);
    dim3 go((unsigned)((g.tiles + 127) / 128), out_planes);
    f32x4 *out = (f32x4 *)d.out.base + roi_out;
    f32x4 *pool = nullptr;
    long pnp = 0;
    int Hp = 0, Wp = 0, ppad = 0;
    if (d.pool) {
        const QpBuf &q = *d.pool;
        if (q.dt != ND_F32 || q.B != d.in.B || q.Hb - 2 * q.pad != g.Hv / 2 || q.Wb - 2 * q.pad != g.Wv / 2 || q.planes < out_planes)
            ND_FAIL(ND_EINVAL, "winograd: pooled destination does not fit %dx%dx%d", d.cout, g.Hv / 2, g.Wv / 2);
        pool = (f32x4 *)q.base;
        pnp = q.np();
        Hp = q.Hb;
        Wp = q.Wb;
        ppad = q.pad;
    }
    if (T == 6 && kOutNtl == 16) {
        dim3 go2((unsigned)((g.tiles + 15) / 16), out_planes);
        hipLaunchKernelGGL((k_wino_out2<6, 16>), go2, dim3(128), 0, s, (const f32x4 *)m, g.mnp, g.mbs, d.in.B, g.TY, g.TX, g.Hv, g.Wv, bias,
                           d.act, d.slope, d.slope_dev, out, d.out.np(), d.out_plane0, (long)d.out.Hb * d.out.Wb, d.out.Wb, d.out.pad, pool, pnp, Hp, Wp, ppad);
    } else if (T == 6) {
        dim3 go2((unsigned)((g.tiles + 31) / 32), out_planes);
        hipLaunchKernelGGL((k_wino_out2<6, 32>), go2, dim3(256), 0, s, (const f32x4 *)m, g.mnp, g.mbs, d.in.B, g.TY, g.TX, g.Hv, g.Wv, bias,
                           d.act, d.slope, d.slope_dev, out, d.out.np(), d.out_plane0, (long)d.out.Hb * d.out.Wb, d.out.Wb, d.out.pad, pool, pnp, Hp, Wp, ppad);
    }
    else if (T == 2)
        hipLaunchKernelGGL(k_wino_output<2>, go, dim3(128), 0, s, (const f32x4 *)m, g.mnp, g.mbs, d.in.B, g.TY, g.TX, g.Hv, g.Wv, bias,
                           d.act, d.slope, d.slope_dev, out, d.out.np(), d.out_plane0, d.out.Hb, d.out.Wb, d.out.pad, pool, pnp, Hp, Wp, ppad);
    else
        hipLaunchKernelGGL(k_wino_output<4>, go, dim3(128), 0, s, (const f32x4 *)m, g.mnp, g.mbs, d.in.B, g.TY, g.TX, g.Hv, g.Wv, bias,
                           d.act, d.slope, d.slope_dev, out, d.out.np(), d.out_plane0, d.out.Hb, d.out.Wb, d.out.pad, pool, pnp, Hp, Wp, ppad);
    ND_HIP(hipGetLastError());
    return ND_OK;
}
