// Training-step building blocks and their single-layer C-ABI entry points (parity tests vs torch autograd).
// Reference: Generator.learn / compute_loss (nn_common.py:201-255) -- loss.backward() + Adam(amsgrad).
#include "nd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int nd_launch_repitch(const QpBuf &src, int src_plane0, int planes, int ss, int sy, int sx, const QpBuf &dst, int oy,
                      int ox, int h, int w, hipStream_t s);
size_t nd_wgrad_partial_floats(int taps, int M, int N, long K, int *ksplit_out, int *cps_out);
int nd_launch_wgrad(const QpBuf &A, int a_plane0, int M, const QpBuf &Bq, int b_plane0, int N, int taps, int taps_total,
                    int tap0, float *partial, size_t partial_floats, float *dw, hipStream_t s);

// ------------------------------------------------------------------ per-channel sum over all pixels (bias gradient)
// two stages, fixed summation order (deterministic): one workgroup per (plane, image), then one per plane
__global__ __launch_bounds__(256) void k_channel_sum1(const f32x4 *__restrict__ src, long np, int Hb, int Wb, int pad, int H,
                                                      int W, f32x4 *__restrict__ partial) {
    __shared__ f32x4 red[256];
    const int q = blockIdx.x, b = blockIdx.y;
    const f32x4 *s = src + (long)q * np + (long)b * Hb * Wb;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int y = i / W, x = i - y * W;
        acc += s[(long)(y + pad) * Wb + x + pad];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(long)q * gridDim.y + b] = red[0];
}
__global__ __launch_bounds__(64) void k_channel_sum2(const f32x4 *__restrict__ partial, int B, int C, float *__restrict__ out) {
    const int q = blockIdx.x;
    if (threadIdx.x >= 4 || 4 * q + (int)threadIdx.x >= C) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += partial[(long)q * B + b][threadIdx.x];
    out[4 * q + threadIdx.x] = acc;
}

// scratch: at least 4 * ceil(C/4) * B floats
int nd_launch_channel_sum(const QpBuf &src, int plane0, int C, float *out, float *scratch, hipStream_t s) {
    const int H = src.Hb - 2 * src.pad, W = src.Wb - 2 * src.pad, planes = (C + 3) / 4;
    hipLaunchKernelGGL(k_channel_sum1, dim3(planes, src.B), dim3(256), 0, s, (const f32x4 *)src.base + (long)plane0 * src.np(),
                       src.np(), src.Hb, src.Wb, src.pad, H, W, (f32x4 *)scratch);
    hipLaunchKernelGGL(k_channel_sum2, dim3(planes), dim3(64), 0, s, (const f32x4 *)scratch, src.B, C, out);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// ------------------------------------------------------------------ single-layer weight / bias gradient (tests)
namespace {
QpBuf make_buf(char *base, size_t *off, int planes, int B, int H, int W) {
    QpBuf q;
    q.planes = planes;
    q.B = B;
    q.Hb = H;
    q.Wb = W;
    q.pad = 0;
    q.pstride = (long)B * H * W;
    q.dt = ND_F32;
    q.base = (float *)(base ? base + *off : nullptr);
    *off += ((size_t)planes * q.pstride + nd_buf_slack(W)) * 16;
    *off = (*off + 255) & ~(size_t)255;
    return q;
}
struct WgPlan {
    QpBuf x, dy, a, b;   // inputs as given; a / b: the operands on the common grid
    float *partial;
    size_t partial_floats, bytes;
    int M, N, taps, gh, gw;
};
WgPlan wg_plan(int kind, int B, int cin, int cout, int h, int w, char *base) {
    WgPlan p;
    size_t off = 0;
    int oh, ow;
    switch (kind) {
        case ND_CONV3: oh = h - 2; ow = w - 2; break;
        case ND_CONVT3: oh = h + 2; ow = w + 2; break;
        case ND_CONVT2S2: oh = 2 * h; ow = 2 * w; break;
        default: oh = h; ow = w; break;
    }
    const int xp = (cin + 3) / 4, yp = (cout + 3) / 4;
    p.x = make_buf(base, &off, xp, B, h, w);
    p.dy = make_buf(base, &off, yp, B, oh, ow);
    p.taps = (kind == ND_CONV3 || kind == ND_CONVT3) ? 9 : 1;
    if (kind == ND_CONV3 || kind == ND_CONV1) {   // grid of x: A = dy re-pitched, B = x
        p.gh = h; p.gw = w; p.M = cout; p.N = cin;
        p.a = make_buf(base, &off, yp, B, h, w);
        p.b = p.x;
    } else if (kind == ND_CONVT3) {               // grid of dy: A = x re-pitched, B = dy
        p.gh = oh; p.gw = ow; p.M = cin; p.N = cout;
        p.a = make_buf(base, &off, xp, B, oh, ow);
        p.b = p.dy;
    } else {                                      // grid of x: A = x, B = one phase of dy
        p.gh = h; p.gw = w; p.M = cin; p.N = cout;
        p.a = p.x;
        p.b = make_buf(base, &off, yp, B, h, w);
    }
    p.partial_floats = nd_wgrad_partial_floats(p.taps, p.M, p.N, (long)B * p.gh * p.gw, nullptr, nullptr);
    p.partial = (float *)(base ? base + off : nullptr);
    off += p.partial_floats * 4 + (size_t)4 * yp * B * 4 + 256;   // + channel-sum scratch behind the partial sums
    p.bytes = off;
    return p;
}
}  // namespace

extern "C" size_t nd_layer_wgrad_workspace_bytes(int kind, int batch, int cin, int cout, int h, int w) {
    if (kind < 0 || kind > 3 || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return 0;
    if (kind == ND_CONV3 && (h < 3 || w < 3)) return 0;
    return wg_plan(kind, batch, cin, cout, h, w, nullptr).bytes;
}

// x [B,cin,h,w] (layer input), dy [B,cout,oh,ow] (gradient of the layer's pre-activation output), both NCHW fp32 in HBM
// -> dw (torch weight layout of the layer) and db [cout]
extern "C" int nd_layer_wgrad(int kind, const float *x, const float *dy, int batch, int cin, int h, int w, int cout,
                              float *dw, float *db, void *ws, size_t ws_bytes, void *stream) {
    const size_t need = nd_layer_wgrad_workspace_bytes(kind, batch, cin, cout, h, w);
    if (!need) ND_FAIL(ND_EINVAL, "nd_layer_wgrad: bad shape");
    if (!ws || ws_bytes < need) ND_FAIL(ND_ENOMEM, "nd_layer_wgrad: workspace %zu B given, %zu B needed", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    WgPlan p = wg_plan(kind, batch, cin, cout, h, w, (char *)ws);
    ND_HIP(hipMemsetAsync(ws, 0, need, s));
    ND_TRY(nd_launch_nchw_to_qp(x, cin, p.x, 0, s));
    ND_TRY(nd_launch_nchw_to_qp(dy, cout, p.dy, 0, s));
    if (db) ND_TRY(nd_launch_channel_sum(p.dy, 0, cout, db, p.partial + p.partial_floats, s));
    if (kind == ND_CONV3 || kind == ND_CONV1) {
        ND_TRY(nd_launch_repitch(p.dy, 0, p.dy.planes, 1, 0, 0, p.a, 0, 0, p.dy.Hb, p.dy.Wb, s));
        ND_TRY(nd_launch_wgrad(p.a, 0, p.M, p.b, 0, p.N, p.taps, p.taps, 0, p.partial, p.partial_floats, dw, s));
    } else if (kind == ND_CONVT3) {
        ND_TRY(nd_launch_repitch(p.x, 0, p.x.planes, 1, 0, 0, p.a, 0, 0, h, w, s));
        ND_TRY(nd_launch_wgrad(p.a, 0, p.M, p.b, 0, p.N, p.taps, p.taps, 0, p.partial, p.partial_floats, dw, s));
    } else {
        for (int ab = 0; ab < 4; ++ab) {
            ND_TRY(nd_launch_repitch(p.dy, 0, p.dy.planes, 2, ab >> 1, ab & 1, p.b, 0, 0, h, w, s));
            ND_TRY(nd_launch_wgrad(p.a, 0, p.M, p.b, 0, p.N, 1, 4, ab, p.partial, p.partial_floats, dw, s));
        }
    }
    return ND_OK;
}
