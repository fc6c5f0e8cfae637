// Device-side weight packing (fp32): the host packers of pack.hip / conv_w1d.hip / winograd.hip as kernels, for weights that
// live in HBM -- the training step re-packs every step (the weights change every step), and a model load packs here instead of
// on host threads (the Winograd transforms of UtNet(64) take ~1 s there).  Same layouts, same maps; the Winograd transforms are
// evaluated in fp32 here (the host packers use double): the packed values agree to ~1e-7 relative.
#include "nd_common.h"

namespace {

// direct form (pack.hip: nd_pack_layer): nw packed weights followed by nb biases
__global__ void k_pack_dev(int kind, int cin, int cout, int M, int KB, int taps, const float *__restrict__ w,
                           const float *__restrict__ bias, float *__restrict__ packed, long nw, int nb) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nw) {
        const int s = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        long rest = idx >> 8;
        const int t = (int)(rest % taps);
        rest /= taps;
        const int kb = (int)(rest % KB), mt = (int)(rest / KB);
        const int m = 32 * mt + (lane & 31), ci = 8 * kb + 4 * (lane >> 5) + s;
        float v = 0.f;
        if (m < M && ci < cin) {
            switch (kind) {
                case ND_CONV3: v = w[((long)m * cin + ci) * 9 + t]; break;
                case ND_CONVT3: v = w[((long)ci * cout + m) * 9 + (8 - t)]; break;
                case ND_CONVT2S2: {
                    const NdUpRow r = nd_up_row(m, cout, ND_F32);
                    v = w[((long)ci * cout + r.co) * 4 + 2 * r.a + r.b];
                    break;
                }
                case ND_CONV2S2: v = w[((long)m * cin + ci) * 4 + t]; break;
                default: v = w[(long)m * cin + ci]; break;
            }
        }
        packed[idx] = v;
    } else if (idx < nw + nb) {
        const int m = (int)(idx - nw);
        packed[idx] = (m < M && bias) ? bias[kind == ND_CONVT2S2 ? nd_up_row(m, cout, ND_F32).co : m] : 0.f;
    }
}

// one row of G g for F(T,3): u[0 .. T+2)
__device__ __forceinline__ void wino_g(int T, const float *g, float *u) {
    if (T == 2) {
        u[0] = g[0];
        u[1] = 0.5f * (g[0] + g[1] + g[2]);
        u[2] = 0.5f * (g[0] - g[1] + g[2]);
        u[3] = g[2];
    } else if (T == 6) {
        u[0] = g[0];
        u[1] = -2.f / 9.f * (g[0] + g[1] + g[2]);
        u[2] = -2.f / 9.f * (g[0] - g[1] + g[2]);
        u[3] = g[0] / 90.f + g[1] / 45.f + g[2] * 2.f / 45.f;
        u[4] = g[0] / 90.f - g[1] / 45.f + g[2] * 2.f / 45.f;
        u[5] = g[0] * 32.f / 45.f + g[1] * 16.f / 45.f + g[2] * 8.f / 45.f;
        u[6] = g[0] * 32.f / 45.f - g[1] * 16.f / 45.f + g[2] * 8.f / 45.f;
        u[7] = g[2];
    } else {
        u[0] = g[0] / 4.f;
        u[1] = -(g[0] + g[1] + g[2]) / 6.f;
        u[2] = -(g[0] - g[1] + g[2]) / 6.f;
        u[3] = g[0] / 24.f + g[1] / 12.f + g[2] / 6.f;
        u[4] = g[0] / 24.f - g[1] / 12.f + g[2] / 6.f;
        u[5] = g[2];
    }
}
__device__ __forceinline__ float tap3(int kind, const float *w, int cin, int cout, int co, int ci, int ky, int kx) {
    return kind == ND_CONV3 ? w[((long)co * cin + ci) * 9 + ky * 3 + kx] : w[((long)ci * cout + co) * 9 + (8 - (ky * 3 + kx))];
}

// fused 1-D Winograd form (conv_w1d.hip: nd_w1d_pack): [mt][kb][ky*(T+2) + xi][lane][4] + bias
__global__ void k_pack_w1d_dev(int T, int kind, int cin, int cout, int KB, const float *__restrict__ w, const float *__restrict__ bias,
                               float *__restrict__ packed, long nw, int nb) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int NP = T + 2;
    if (idx < nw) {
        const int s = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        long rest = idx >> 8;
        const int plane = (int)(rest % (3 * NP));
        rest /= 3 * NP;
        const int kb = (int)(rest % KB), mt = (int)(rest / KB);
        const int ky = plane / NP, xi = plane - NP * ky;
        const int m = 32 * mt + (lane & 31), ci = 8 * kb + 4 * (lane >> 5) + s;
        float v = 0.f;
        if (m < cout && ci < cin) {
            float g[3], u[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] = tap3(kind, w, cin, cout, m, ci, ky, k);
            wino_g(T, g, u);
            v = u[xi];
        }
        packed[idx] = v;
    } else if (idx < nw + nb) {
        const int m = (int)(idx - nw);
        packed[idx] = (m < cout && bias) ? bias[m] : 0.f;
    }
}

// three-pass Winograd form (winograd.hip: nd_wino_pack): P = (T+2)^2 1-tap GEMM blobs [mt][kb][lane][4] + zero bias each,
// then bias[cout rounded up to 4].  gf = floats per position blob, nwp = packed weights per position
__global__ void k_pack_wino_dev(int T, int kind, int cin, int cout, int KB, const float *__restrict__ w, const float *__restrict__ bias,
                                float *__restrict__ packed, long gf, long nwp, int P) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int A = T + 2;
    if (idx < (long)P * gf) {
        const int pos = (int)(idx / gf);
        const long r = idx - (long)pos * gf;
        float v = 0.f;
        if (r < nwp) {
            const int s = (int)(r & 3), lane = (int)((r >> 2) & 63);
            const long rest = r >> 8;
            const int kb = (int)(rest % KB), mt = (int)(rest / KB);
            const int m = 32 * mt + (lane & 31), ci = 8 * kb + 4 * (lane >> 5) + s;
            if (m < cout && ci < cin) {
                const int i = pos / A, j = pos - A * i;
                float col[3];   // (G g)[i][kx] for kx = 0..2
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float g[3], u[8];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) g[ky] = tap3(kind, w, cin, cout, m, ci, ky, kx);
                    wino_g(T, g, u);
                    col[kx] = u[i];
                }
                float u[8];
                wino_g(T, col, u);
                v = u[j];
            }
        }
        packed[idx] = v;   // (the tail of a position blob is its zero bias)
    } else {
        const long m = idx - (long)P * gf;
        if (m < (cout + 3) / 4 * 4) packed[idx] = (bias && m < cout) ? bias[m] : 0.f;
    }
}

}  // namespace

int nd_pack_layer_device(int kind, int cin, int cout, const float *w, const float *bias, float *packed, hipStream_t s) {
    const int MT = nd_mtiles(kind, cout), KB = nd_kblocks(cin), taps = nd_taps(kind);
    const int M = kind == ND_CONVT2S2 ? 4 * cout : cout;
    const long nw = (long)MT * KB * taps * 256;
    hipLaunchKernelGGL(k_pack_dev, dim3((unsigned)((nw + MT * 32 + 255) / 256)), dim3(256), 0, s, kind, cin, cout, M, KB, taps, w, bias,
                       packed, nw, MT * 32);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

int nd_pack_w1d_device(int T, int kind, int cin, int cout, const float *w, const float *bias, float *packed, hipStream_t s) {
    if ((T != 2 && T != 4) || (kind != ND_CONV3 && kind != ND_CONVT3)) ND_FAIL(ND_EINVAL, "device w1d packing: T = 2 | 4, 3x3 layers");
    const int MT = nd_mtiles(ND_CONV3, cout), KB = nd_kblocks(cin);
    const long nw = (long)MT * KB * 3 * (T + 2) * 256;
    hipLaunchKernelGGL(k_pack_w1d_dev, dim3((unsigned)((nw + MT * 32 + 255) / 256)), dim3(256), 0, s, T, kind, cin, cout, KB, w, bias,
                       packed, nw, MT * 32);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

int nd_pack_wino_device(int T, int kind, int cin, int cout, const float *w, const float *bias, float *packed, hipStream_t s) {
    if ((T != 2 && T != 4 && T != 6) || (kind != ND_CONV3 && kind != ND_CONVT3)) ND_FAIL(ND_EINVAL, "device Winograd packing: T = 2 | 4 | 6, 3x3 layers");
    const int P = (T + 2) * (T + 2), KB = nd_kblocks(cin);
    const long gf = (long)nd_packed_floats(ND_CONV1, cin, cout, ND_F32);
    const long nwp = (long)nd_mtiles(ND_CONV1, cout) * KB * 256;
    const long total = (long)P * gf + (cout + 3) / 4 * 4;
    hipLaunchKernelGGL(k_pack_wino_dev, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, T, kind, cin, cout, KB, w, bias, packed,
                       gf, nwp, P);
    ND_HIP(hipGetLastError());
    return ND_OK;
}
