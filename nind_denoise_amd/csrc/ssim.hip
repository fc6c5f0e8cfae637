// Image-quality scores of the eval harness: MSE, SSIM and MS-SSIM on the GPU.
// Reference: common/libs/pt_helpers.py:40-48 (get_losses), common/libs/pt_losses.py:6-18 (1 - piqa.SSIM / piqa.MS_SSIM with
// piqa's defaults), nn_common.py:170-177 (the same two classes as training losses).  piqa (~=1.3.2) is a third-party
// dependency that is neither vendored by the reference nor installed here: the algorithm below restates its published
// definition (see oracle/losses.py for the statement and for what pins it).
//
//   window: 11-tap Gaussian (sigma 1.5, normalised), separable, per channel, VALID;  c1 = 0.01^2, c2 = 0.03^2
//   cs = (2 s_xy + c2) / (s_xx + s_yy + c2),  ss = (2 mu_x mu_y + c1) / (mu_x^2 + mu_y^2 + c1) * cs
//   SSIM = mean_{c,h,w} ss;   MS-SSIM = mean_c prod_i relu(cs_i or ss_5)^w_i over 5 scales of avg_pool2d(2, ceil_mode)
//
// HBM-bound: every scale reads x and y once.  One workgroup computes a 32 x 32 patch of the score maps from a 42 x 42 patch
// of x and y staged in LDS (row filter of the five moment maps into LDS, then the column filter per output pixel) and
// reduces it to one partial (sum ss, sum cs); partials are added in a fixed order (deterministic, no atomics).
// Algorithmic bytes: 2 * 4 B per input pixel per scale (x 1.72 for the 10-pixel halo of a 32 x 32 patch, mostly served by L2).
#include <math.h>

#include "nd_common.h"

namespace {
constexpr int kWin = 11;
constexpr int kTile = 32;
constexpr int kIn = kTile + kWin - 1;   // 42
constexpr int kScales = 5;
__constant__ float c_gauss[kWin];
const float kMsWeights[kScales] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};

// grid (tiles_x, tiles_y, planes); x, y: [planes][H][W]
__global__ __launch_bounds__(256) void k_ssim_tile(const float *__restrict__ x, const float *__restrict__ y, int H, int W,
                                                   float2 *__restrict__ partial) {
    __shared__ float sx[kIn][kIn + 1], sy[kIn][kIn + 1];
    __shared__ float hm[5][kIn][kTile + 1];
    __shared__ float2 red[256];
    const int plane = blockIdx.z;
    const int ox0 = blockIdx.x * kTile, oy0 = blockIdx.y * kTile;
    const float *xp = x + (size_t)plane * H * W, *yp = y + (size_t)plane * H * W;
    for (int i = threadIdx.x; i < kIn * kIn; i += 256) {
        const int r = i / kIn, c = i - r * kIn;
        const int gy = oy0 + r, gx = ox0 + c;
        const bool in = gy < H && gx < W;
        sx[r][c] = in ? xp[(size_t)gy * W + gx] : 0.f;
        sy[r][c] = in ? yp[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kIn * kTile; i += 256) {
        const int r = i / kTile, c = i - r * kTile;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const float g = c_gauss[k], u = sx[r][c + k], v = sy[r][c + k];
            a += g * u;
            b += g * v;
            aa += g * (u * u);
            bb += g * (v * v);
            ab += g * (u * v);
        }
        hm[0][r][c] = a;
        hm[1][r][c] = b;
        hm[2][r][c] = aa;
        hm[3][r][c] = bb;
        hm[4][r][c] = ab;
    }
    __syncthreads();
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    const int Hv = H - kWin + 1, Wv = W - kWin + 1;
    float sum_ss = 0.f, sum_cs = 0.f;
    for (int i = threadIdx.x; i < kTile * kTile; i += 256) {
        const int r = i / kTile, c = i - r * kTile;
        if (oy0 + r >= Hv || ox0 + c >= Wv) continue;
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const float g = c_gauss[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += g * hm[q][r + k][c];
        }
        const float mxx = m[0] * m[0], myy = m[1] * m[1], mxy = m[0] * m[1];
        const float sxx = m[2] - mxx, syy = m[3] - myy, sxy = m[4] - mxy;
        const float cs = (2.f * sxy + c2) / (sxx + syy + c2);
        const float ss = (2.f * mxy + c1) / (mxx + myy + c1) * cs;
        sum_ss += ss;
        sum_cs += cs;
    }
    red[threadIdx.x] = make_float2(sum_ss, sum_cs);
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            red[threadIdx.x].x += red[threadIdx.x + k].x;
            red[threadIdx.x].y += red[threadIdx.x + k].y;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0];
}

// stats[plane] = (mean ss, mean cs): one workgroup per plane, fixed summation order
__global__ __launch_bounds__(256) void k_ssim_reduce(const float2 *__restrict__ partial, int nblocks, float inv_count,
                                                     float2 *__restrict__ stats) {
    __shared__ float2 red[256];
    const float2 *p = partial + (size_t)blockIdx.x * nblocks;
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        a += p[i].x;
        b += p[i].y;
    }
    red[threadIdx.x] = make_float2(a, b);
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            red[threadIdx.x].x += red[threadIdx.x + k].x;
            red[threadIdx.x].y += red[threadIdx.x + k].y;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) stats[blockIdx.x] = make_float2(red[0].x * inv_count, red[0].y * inv_count);
}

// avg_pool2d(kernel 2, ceil_mode=True): a window hanging over the edge averages the pixels it has.  grid (.., H2, planes*2)
__global__ void k_pool2_ceil(const float *__restrict__ x, const float *__restrict__ y, int H, int W, float *__restrict__ xo,
                             float *__restrict__ yo, int H2, int W2, int planes) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
    if (ox >= W2) return;
    const int z = blockIdx.z;
    const float *src = (z < planes ? x : y) + (size_t)(z % planes) * H * W;
    float *dst = (z < planes ? xo : yo) + (size_t)(z % planes) * H2 * W2;
    const int y0 = 2 * oy, x0 = 2 * ox;
    const int ny = y0 + 1 < H ? 2 : 1, nx = x0 + 1 < W ? 2 : 1;
    float s = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) s += src[(size_t)(y0 + a) * W + x0 + b];
    dst[(size_t)oy * W2 + ox] = s / (float)(ny * nx);
}

// out[n] = mean_c stats[n*c + ch].x
__global__ void k_ssim_final(const float2 *__restrict__ stats, int n, int c, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int ch = 0; ch < c; ++ch) s += stats[i * c + ch].x;
    out[i] = s / (float)c;
}

// out[n] = mean_c prod_i relu(v_i)^w_i, v_i = cs of scale i (ss at the last scale);  stats: [scale][planes]
__global__ void k_msssim_final(const float2 *__restrict__ stats, int n, int c, float w0, float w1, float w2, float w3, float w4,
                               float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float w[kScales] = {w0, w1, w2, w3, w4};
    const int planes = n * c;
    float s = 0.f;
    for (int ch = 0; ch < c; ++ch) {
        float prod = 1.f;
        for (int k = 0; k < kScales; ++k) {
            const float2 st = stats[(size_t)k * planes + i * c + ch];
            const float v = fmaxf(k + 1 < kScales ? st.y : st.x, 0.f);
            prod *= powf(v, w[k]);
        }
        s += prod;
    }
    out[i] = s / (float)c;
}

__global__ __launch_bounds__(256) void k_sqdiff_partial(const float *__restrict__ x, const float *__restrict__ y, size_t n,
                                                        float *__restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = x[i] - y[i];
        acc += d * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_sum_scale(const float *__restrict__ partial, int n, float scale, float *__restrict__ out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// ------------------------------------------------------------------ backward (SSIM / MS-SSIM as training losses)
// F = mean_q phi(q) per plane, phi = ss (SSIM; last MS-SSIM scale) or cs (other MS-SSIM scales):
//   dF/dx(u) = 1/Nq * [ (G^T A)(u) + 2 x(u) (G^T B)(u) + y(u) (G^T C)(u) ],   A = dphi/dmu_x, B = dphi/dE[xx], C = dphi/dE[xy]
// (G^T = the same separable window applied as a FULL correlation over the valid score positions q).
// One workgroup: 16 x 16 gradient pixels <- derivative maps at 26 x 26 score positions <- a 36 x 36 patch of x and y.
// gx(u) (+)= coef[plane] * dF/dx(u) + gcoarse(u/2) / (pixels of that pooling window)      [avg_pool2d(2, ceil) backward]
constexpr int kBT = 16;
constexpr int kBQ = kBT + kWin - 1;    // 26 score positions per side
constexpr int kBI = kBQ + kWin - 1;    // 36 input pixels per side
__global__ __launch_bounds__(256) void k_ssim_bwd_tile(const float *__restrict__ x, const float *__restrict__ y, int H, int W,
                                                       const float *__restrict__ coef, int use_ss,
                                                       const float *__restrict__ gcoarse, int H2, int W2,
                                                       float *__restrict__ gx, int accumulate) {
    __shared__ float sx[kBI][kBI + 1], sy[kBI][kBI + 1];
    __shared__ float hm[5][kBI][kBQ + 1];
    __shared__ float dm[3][kBQ][kBQ + 1];
    __shared__ float rm[3][kBQ][kBT + 1];
    const int plane = blockIdx.z;
    const int ux0 = blockIdx.x * kBT, uy0 = blockIdx.y * kBT;   // first gradient pixel of the tile
    const int qx0 = ux0 - (kWin - 1), qy0 = uy0 - (kWin - 1);   // first score position it depends on
    const float *xp = x + (size_t)plane * H * W, *yp = y + (size_t)plane * H * W;
    for (int i = threadIdx.x; i < kBI * kBI; i += 256) {
        const int r = i / kBI, c = i - r * kBI;
        const int gy_ = qy0 + r, gx_ = qx0 + c;
        const bool in = gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;
        sx[r][c] = in ? xp[(size_t)gy_ * W + gx_] : 0.f;
        sy[r][c] = in ? yp[(size_t)gy_ * W + gx_] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBI * kBQ; i += 256) {
        const int r = i / kBQ, c = i - r * kBQ;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const float g = c_gauss[k], u = sx[r][c + k], v = sy[r][c + k];
            a += g * u;
            b += g * v;
            aa += g * (u * u);
            bb += g * (v * v);
            ab += g * (u * v);
        }
        hm[0][r][c] = a;
        hm[1][r][c] = b;
        hm[2][r][c] = aa;
        hm[3][r][c] = bb;
        hm[4][r][c] = ab;
    }
    __syncthreads();
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    const int Hv = H - kWin + 1, Wv = W - kWin + 1;
    for (int i = threadIdx.x; i < kBQ * kBQ; i += 256) {
        const int r = i / kBQ, c = i - r * kBQ;
        const int qy = qy0 + r, qx = qx0 + c;
        float A = 0.f, B = 0.f, C = 0.f;
        if (qy >= 0 && qy < Hv && qx >= 0 && qx < Wv) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < kWin; ++k) {
                const float g = c_gauss[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] += g * hm[q][r + k][c];
            }
            const float mx = m[0], my = m[1];
            const float a2 = 2.f * (m[4] - mx * my) + c2, b2 = (m[2] - mx * mx) + (m[3] - my * my) + c2;
            const float cs = a2 / b2;
            const float dcs_mx = 2.f / b2 * (mx * cs - my), dcs_xx = -cs / b2, dcs_xy = 2.f / b2;
            if (use_ss) {
                const float b1 = mx * mx + my * my + c1, l = (2.f * mx * my + c1) / b1;
                A = 2.f / b1 * (my - mx * l) * cs + l * dcs_mx;
                B = l * dcs_xx;
                C = l * dcs_xy;
            } else {
                A = dcs_mx;
                B = dcs_xx;
                C = dcs_xy;
            }
        }
        dm[0][r][c] = A;
        dm[1][r][c] = B;
        dm[2][r][c] = C;
    }
    __syncthreads();
    // full correlation: out(u) = sum_{k} g[k] * d(u - k) ; local index of q = u - k is (u_local + 10 - k)
    for (int i = threadIdx.x; i < kBQ * kBT; i += 256) {
        const int r = i / kBT, c = i - r * kBT;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const float g = c_gauss[k];
            a += g * dm[0][r][c + kWin - 1 - k];
            b += g * dm[1][r][c + kWin - 1 - k];
            d += g * dm[2][r][c + kWin - 1 - k];
        }
        rm[0][r][c] = a;
        rm[1][r][c] = b;
        rm[2][r][c] = d;
    }
    __syncthreads();
    const float cf = coef[plane];
    {
        const int r = threadIdx.x / kBT, c = threadIdx.x - r * kBT;
        const int uy = uy0 + r, ux = ux0 + c;
        if (uy < H && ux < W) {
            float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
            for (int k = 0; k < kWin; ++k) {
                const float g = c_gauss[k];
                a += g * rm[0][r + kWin - 1 - k][c];
                b += g * rm[1][r + kWin - 1 - k][c];
                d += g * rm[2][r + kWin - 1 - k][c];
            }
            const float xv = sx[r + kWin - 1][c + kWin - 1], yv = sy[r + kWin - 1][c + kWin - 1];
            float g = cf * (a + 2.f * xv * b + yv * d);
            if (gcoarse) {
                const int oy = uy >> 1, ox = ux >> 1;
                const int ny = 2 * oy + 1 < H ? 2 : 1, nx = 2 * ox + 1 < W ? 2 : 1;
                g += gcoarse[((size_t)plane * H2 + oy) * W2 + ox] / (float)(ny * nx);
            }
            float *dst = gx + ((size_t)plane * H + uy) * W + ux;
            *dst = accumulate ? *dst + g : g;
        }
    }
}

// per-plane coefficients d loss / d F_scale (including 1/Nq) and the loss value.
//   SSIM  (scales = 1): loss = weight * mean_n (1 - mean_c ss)
//   MS-SSIM           : loss = weight * mean_n (1 - mean_c prod_s relu(v_s)^w_s)
// one thread per sample; loss_acc[0] += sum (single thread, fixed order)
struct Five { float v[kScales]; };
__global__ void k_ssim_loss_coef(const float2 *__restrict__ stats, int n, int c, int scales, Five wts, float weight, Five inv_nq_,
                                 float *__restrict__ coef, float *__restrict__ loss_acc) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float *w = wts.v, *inv_nq = inv_nq_.v;
    const int planes = n * c;
    float total = 0.f;
    for (int i = 0; i < n; ++i) {
        float score = 0.f;
        for (int ch = 0; ch < c; ++ch) {
            const int p = i * c + ch;
            if (scales == 1) {
                score += stats[p].x;
                coef[p] = -weight / (float)(n * c) * inv_nq[0];
            } else {
                float v[kScales], prod = 1.f;
                for (int k = 0; k < kScales; ++k) {
                    const float2 st = stats[(size_t)k * planes + p];
                    v[k] = fmaxf(k + 1 < kScales ? st.y : st.x, 0.f);
                    prod *= powf(v[k], w[k]);
                }
                score += prod;
                for (int k = 0; k < kScales; ++k)
                    coef[(size_t)k * planes + p] = v[k] > 0.f ? -weight / (float)(n * c) * w[k] * prod / v[k] * inv_nq[k] : 0.f;
            }
        }
        total += 1.f - score / (float)c;
    }
    loss_acc[0] += weight * total / (float)n;
}

struct SsimPlan {
    int h[kScales], w[kScales];
    float *px[kScales], *py[kScales];   // pyramid levels 1.. (level 0 = the caller's tensors)
    float2 *partial, *stats;
    float *pg[kScales];                 // training: gradient wrt pyramid levels 1..
    float *coef;
    size_t bytes;
};
SsimPlan ssim_plan(int n, int c, int h, int w, char *base, bool with_grad = false) {
    SsimPlan p;
    size_t off = 0;
    const size_t planes = (size_t)n * c;
    p.h[0] = h;
    p.w[0] = w;
    p.px[0] = p.py[0] = nullptr;
    for (int i = 1; i < kScales; ++i) {
        p.h[i] = (p.h[i - 1] + 1) / 2;
        p.w[i] = (p.w[i - 1] + 1) / 2;
        const size_t sz = (planes * p.h[i] * p.w[i] * 4 + 255) & ~(size_t)255;
        p.px[i] = (float *)(base + off);
        off += sz;
        p.py[i] = (float *)(base + off);
        off += sz;
    }
    const size_t nblk = (size_t)((h + kTile - 1) / kTile) * ((w + kTile - 1) / kTile);
    p.partial = (float2 *)(base + off);
    off += (planes * nblk * 8 + 255) & ~(size_t)255;
    p.stats = (float2 *)(base + off);
    off += (kScales * planes * 8 + 255) & ~(size_t)255;
    p.pg[0] = nullptr;
    if (with_grad) {
        for (int i = 1; i < kScales; ++i) {
            p.pg[i] = (float *)(base + off);
            off += (planes * p.h[i] * p.w[i] * 4 + 255) & ~(size_t)255;
        }
        p.coef = (float *)(base + off);
        off += (kScales * planes * 4 + 255) & ~(size_t)255;
    }
    p.bytes = off;
    return p;
}

int upload_window() {
    static bool done[64] = {false};   // __constant__ memory is per device
    int dev = 0;
    ND_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) ND_FAIL(ND_EINVAL, "ssim: device index %d", dev);
    if (done[dev]) return ND_OK;
    float g[kWin];
    float s = 0.f;   // fp32 throughout, like torch's kernel / kernel.sum()
    for (int k = 0; k < kWin; ++k) {
        const float d = (float)k - (kWin - 1) / 2.f;
        g[k] = expf(-(d * d) / (2.f * 1.5f * 1.5f));
        s += g[k];
    }
    for (int k = 0; k < kWin; ++k) g[k] /= s;
    ND_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), g, sizeof(g)));
    done[dev] = true;
    return ND_OK;
}

// score maps of `scales` pyramid levels -> plan.stats[scale][plane]
int run_scales(const float *x, const float *y, int n, int c, const SsimPlan &p, int scales, hipStream_t s) {
    ND_TRY(upload_window());
    const int planes = n * c;
    const float *cx = x, *cy = y;
    for (int i = 0; i < scales; ++i) {
        if (i > 0) {
            dim3 g((p.w[i] + 127) / 128, p.h[i], 2 * planes);
            hipLaunchKernelGGL(k_pool2_ceil, g, dim3(128), 0, s, cx, cy, p.h[i - 1], p.w[i - 1], p.px[i], p.py[i], p.h[i], p.w[i], planes);
            cx = p.px[i];
            cy = p.py[i];
        }
        const int Hv = p.h[i] - kWin + 1, Wv = p.w[i] - kWin + 1;
        dim3 g((Wv + kTile - 1) / kTile, (Hv + kTile - 1) / kTile, planes);
        hipLaunchKernelGGL(k_ssim_tile, g, dim3(256), 0, s, cx, cy, p.h[i], p.w[i], p.partial);
        hipLaunchKernelGGL(k_ssim_reduce, dim3(planes), dim3(256), 0, s, (const float2 *)p.partial, (int)(g.x * g.y),
                           1.f / ((float)Hv * (float)Wv), p.stats + (size_t)i * planes);
    }
    ND_HIP(hipGetLastError());
    return ND_OK;
}

int check_shape(const char *who, int n, int c, int h, int w, int min_side) {
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) ND_FAIL(ND_EINVAL, "%s: bad shape [%d,%d,%d,%d]", who, n, c, h, w);
    if (h < min_side || w < min_side)
        ND_FAIL(ND_EINVAL, "%s: %dx%d image is too small (needs at least %d pixels per side: an 11-tap window must fit%s)", who, h, w,
                min_side, min_side > kWin ? " the fifth scale" : "");
    if ((long)n * c > 32767) ND_FAIL(ND_EINVAL, "%s: more than 32767 image planes", who);
    return ND_OK;
}
}  // namespace

extern "C" size_t nd_ssim_workspace_bytes(int n, int c, int h, int w) {
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return 0;
    return ssim_plan(n, c, h, w, nullptr).bytes;
}

extern "C" int nd_ssim(const float *x, const float *y, int n, int c, int h, int w, float *out, void *ws, size_t ws_bytes,
                       void *stream) {
    ND_TRY(check_shape("nd_ssim", n, c, h, w, kWin));
    const SsimPlan p = ssim_plan(n, c, h, w, (char *)ws);
    if (!ws || ws_bytes < p.bytes) ND_FAIL(ND_ENOMEM, "nd_ssim: workspace %zu B given, %zu B needed", ws_bytes, p.bytes);
    hipStream_t s = (hipStream_t)stream;
    ND_TRY(run_scales(x, y, n, c, p, 1, s));
    hipLaunchKernelGGL(k_ssim_final, dim3((n + 63) / 64), dim3(64), 0, s, (const float2 *)p.stats, n, c, out);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

extern "C" int nd_ms_ssim(const float *x, const float *y, int n, int c, int h, int w, float *out, void *ws, size_t ws_bytes,
                          void *stream) {
    ND_TRY(check_shape("nd_ms_ssim", n, c, h, w, (kWin - 1) * 16 + 1));   // 161: ceil-halved four times it is still >= 11
    const SsimPlan p = ssim_plan(n, c, h, w, (char *)ws);
    if (!ws || ws_bytes < p.bytes) ND_FAIL(ND_ENOMEM, "nd_ms_ssim: workspace %zu B given, %zu B needed", ws_bytes, p.bytes);
    hipStream_t s = (hipStream_t)stream;
    ND_TRY(run_scales(x, y, n, c, p, kScales, s));
    hipLaunchKernelGGL(k_msssim_final, dim3((n + 63) / 64), dim3(64), 0, s, (const float2 *)p.stats, n, c, kMsWeights[0],
                       kMsWeights[1], kMsWeights[2], kMsWeights[3], kMsWeights[4], out);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// out[0] = mean (x - y)^2 over `count` floats (F.mse_loss).  ws: >= 4 KiB
extern "C" int nd_mse(const float *x, const float *y, size_t count, float *out, void *ws, size_t ws_bytes, void *stream) {
    if (!count) ND_FAIL(ND_EINVAL, "nd_mse: empty input");
    if (!ws || ws_bytes < 4096) ND_FAIL(ND_ENOMEM, "nd_mse: workspace %zu B given, 4096 B needed", ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sqdiff_partial, dim3(1024), dim3(256), 0, s, x, y, count, (float *)ws);
    hipLaunchKernelGGL(k_sum_scale, dim3(1), dim3(256), 0, s, (const float *)ws, 1024, 1.f / (float)count, out);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// ------------------------------------------------------------------ SSIM / MS-SSIM as differentiable losses
// loss_acc[0] += weight * mean_n (1 - score_n);   gx (+)= d(that) / dx      (x = generated batch, y = target; both [n,c,h,w])
// Reference: nn_common.py:170-177, 226-241 (criterions['SSIM' | 'MSSSIM'] = pt_losses.*_loss, weighted sum, .backward()).
extern "C" size_t nd_ssim_loss_workspace_bytes(int n, int c, int h, int w) {
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return 0;
    return ssim_plan(n, c, h, w, nullptr, true).bytes;
}

extern "C" int nd_ssim_loss_grad(const float *x, const float *y, int n, int c, int h, int w, int multiscale, float weight,
                                 float *loss_acc, float *gx, int accumulate, void *ws, size_t ws_bytes, void *stream) {
    ND_TRY(check_shape(multiscale ? "nd_ssim_loss_grad (MS-SSIM)" : "nd_ssim_loss_grad", n, c, h, w,
                       multiscale ? (kWin - 1) * 16 + 1 : kWin));
    const SsimPlan p = ssim_plan(n, c, h, w, (char *)ws, true);
    if (!ws || ws_bytes < p.bytes) ND_FAIL(ND_ENOMEM, "nd_ssim_loss_grad: workspace %zu B given, %zu B needed", ws_bytes, p.bytes);
    hipStream_t s = (hipStream_t)stream;
    const int scales = multiscale ? kScales : 1, planes = n * c;
    ND_TRY(run_scales(x, y, n, c, p, scales, s));
    Five wts, inv;
    for (int i = 0; i < kScales; ++i) {
        wts.v[i] = kMsWeights[i];
        inv.v[i] = 1.f / ((float)(p.h[i] - kWin + 1) * (float)(p.w[i] - kWin + 1));
    }
    hipLaunchKernelGGL(k_ssim_loss_coef, dim3(1), dim3(64), 0, s, (const float2 *)p.stats, n, c, scales, wts, weight, inv, p.coef,
                       loss_acc);
    // coarsest scale first; every finer level adds the pooled-back gradient of the level below it
    for (int i = scales - 1; i >= 0; --i) {
        const float *cx = i ? p.px[i] : x, *cy = i ? p.py[i] : y;
        float *dst = i ? p.pg[i] : gx;
        const float *coarse = i + 1 < scales ? p.pg[i + 1] : nullptr;
        dim3 g((p.w[i] + kBT - 1) / kBT, (p.h[i] + kBT - 1) / kBT, planes);
        hipLaunchKernelGGL(k_ssim_bwd_tile, g, dim3(256), 0, s, cx, cy, p.h[i], p.w[i], p.coef + (size_t)i * planes,
                           (multiscale && i + 1 < scales) ? 0 : 1, coarse, coarse ? p.h[i + 1] : 0, coarse ? p.w[i + 1] : 0, dst,
                           i == 0 ? accumulate : 0);
    }
    ND_HIP(hipGetLastError());
    return ND_OK;
}
