// UtNet training step (forward + loss + backward) -- BASELINE config 5 / SURVEY.md section 8(f) rank 3.
// Reference: nn_train.py:308-380 (loop), nn_common.py:198-255 (denoise_batch = model(x).clip(0,1); compute_loss;
// loss.backward(); Adam(amsgrad).step()).
//
// Everything heavy reuses the inference machinery:
//   forward        the inference launch sequence (run_stack) with the conv epilogue also storing acc + bias ("pre")
//   data gradient  the SAME conv kernel on the weight tensor read in its transposed role:
//                  dgrad(Conv2d) = ConvTranspose2d, dgrad(ConvTranspose2d) = Conv2d, dgrad(ConvT 2x2 s2) = Conv 2x2 s2
//   weight gradient k_wgrad (wgrad.hip): MFMA contraction over pixels on a common grid (k_repitch)
// Small HBM-bound kernels below: PReLU backward (+ slope gradient), max-pool backward, final 1x1 backward, loss, bias
// sums, device-side weight packing, Adam(amsgrad).
// Gradients come out in ONE flat fp32 buffer in state-dict order, so the data-parallel all-reduce (RCCL) and the
// optimizer are single flat operations.
#include <math.h>

#include <algorithm>

#include "utnet_net.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int nd_launch_repitch(const QpBuf &src, int src_plane0, int planes, int ss, int sy, int sx, const QpBuf &dst, int oy,
                      int ox, int h, int w, hipStream_t s);
size_t nd_wgrad_partial_floats(int taps, int M, int N, long K, int *ksplit_out, int *cps_out);
int nd_launch_wgrad(const QpBuf &A, int a_plane0, int M, const QpBuf &Bq, int b_plane0, int N, int taps, int taps_total,
                    int tap0, float *partial, size_t partial_floats, float *dw, hipStream_t s);
int nd_launch_channel_sum(const QpBuf &src, int plane0, int C, float *out, float *scratch, hipStream_t s);

// ------------------------------------------------------------------ elementwise kernels
// g (interior of a bordered gradient buffer, planes [plane0, plane0+planes)) *= act'(pre), pre = conv output + bias;
// PReLU also: partial[block] = sum g*pre over pre <= 0 (the slope's gradient).  ELU: act' = 1 | exp(pre); Hardswish: 0 | (2 pre + 3) / 6 | 1
// (torch's derivative at the break points: ELU'(0) = 1 from the exp branch, Hardswish' = 0 at -3 and 1 at +3 are not reached by (2x+3)/6 -- see below)
template <int ACT>
__global__ __launch_bounds__(256) void k_act_bwd(f32x4 *__restrict__ g, long gnp, int gHb, int gWb, int gpad,
                                                 const f32x4 *__restrict__ pre, long pnp, int H, int W,
                                                 const float *__restrict__ slope, float *__restrict__ partial) {
    __shared__ float red[256];
    const int b = blockIdx.x, q = blockIdx.y;
    const float a = ACT == ND_ACT_PRELU ? *slope : 0.f;
    float acc = 0.f;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int y = i / W, x = i - y * W;
        f32x4 *gp = g + (long)q * gnp + ((long)b * gHb + y + gpad) * gWb + x + gpad;
        const f32x4 pv = pre[(long)q * pnp + (long)b * H * W + i];
        f32x4 gv = *gp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ACT == ND_ACT_PRELU) {
                if (!(pv[e] > 0.f)) {
                    acc += gv[e] * pv[e];
                    gv[e] *= a;
                }
            } else if (ACT == ND_ACT_ELU) {
                if (!(pv[e] > 0.f)) gv[e] *= expf(pv[e]);                     // torch: grad * (x <= 0 ? alpha * exp(x) : 1), alpha = 1
            } else {
                // torch hardswish_backward: x < -3 -> 0;  x <= 3 -> grad * (x / 3 + 0.5);  else grad
                gv[e] = pv[e] < -3.f ? 0.f : (pv[e] <= 3.f ? gv[e] * (pv[e] / 3.f + 0.5f) : gv[e]);
            }
        }
        *gp = gv;
    }
    if (ACT != ND_ACT_PRELU) return;
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(long)q * gridDim.x + b] = red[0];
}

// out[0] = scale * sum partial[0..n)   (one workgroup, fixed order)
__global__ __launch_bounds__(256) void k_sum_partials(const float *__restrict__ partial, int n, float scale, float *__restrict__ out) {
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

// gfine[argmax of each 2x2 window] += gpool   (fwd: the forward values that were pooled; first maximum in row-major order)
__global__ void k_maxpool_bwd_add(const f32x4 *__restrict__ gpool, long pnp, int pHb, int pWb, int ppad,
                                  const f32x4 *__restrict__ fwd, long fnp, int fHb, int fWb, int fpad,
                                  f32x4 *__restrict__ gfine, long gnp, int gHb, int gWb, int gpad, int Ho, int Wo, int B) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int b = blockIdx.z % B, q = blockIdx.z / B;
    if (x >= Wo) return;
    const f32x4 gv = gpool[(long)q * pnp + ((long)b * pHb + y + ppad) * pWb + x + ppad];
    const f32x4 *f = fwd + (long)q * fnp + ((long)b * fHb + 2 * y + fpad) * fWb + 2 * x + fpad;
    f32x4 *g = gfine + (long)q * gnp + ((long)b * gHb + 2 * y + gpad) * gWb + 2 * x + gpad;
    const f32x4 v[4] = {f[0], f[1], f[fWb], f[fWb + 1]};
    f32x4 o[4] = {g[0], g[1], g[gWb], g[gWb + 1]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int best = 0;
        float m = v[0][e];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (v[k][e] > m) {
                m = v[k][e];
                best = k;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k == best) o[k][e] += gv[e];
    }
    g[0] = o[0];
    g[1] = o[1];
    g[gWb] = o[2];
    g[gWb + 1] = o[3];
}

// loss = w_l1 * mean|clip(y) - t| + w_mse * mean (clip(y) - t)^2 ;  gy = d loss / d y   (nn_common.py:198-199, 236-255)
__global__ __launch_bounds__(256) void k_loss_grad(const float *__restrict__ y, const float *__restrict__ t, long n, float w_l1,
                                                   float w_mse, float *__restrict__ gy, float *__restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    const float inv = 1.f / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = y[i];
        const float c = fminf(fmaxf(v, 0.f), 1.f);
        const float d = c - t[i];
        acc += w_l1 * fabsf(d) + w_mse * d * d;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float pass = (v >= 0.f && v <= 1.f) ? 1.f : 0.f;   // clamp passes the gradient on [min, max]
        gy[i] = pass * (w_l1 * sgn + w_mse * 2.f * d) * inv;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// SSIM / MS-SSIM act on clip(y, 0, 1) (nn_common.py:198-199): yc = clip(y);  later gy += [0 <= y <= 1] * g(yc)
__global__ void k_clip01(const float *__restrict__ y, long n, float *__restrict__ yc) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        yc[i] = fminf(fmaxf(y[i], 0.f), 1.f);
}
__global__ void k_add_clip_grad(const float *__restrict__ y, const float *__restrict__ gc, long n, float *__restrict__ gy) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = y[i];
        if (v >= 0.f && v <= 1.f) gy[i] += gc[i];
    }
}

// pt_ops.pt_crop_batch (common/libs/pt_ops.py:1-8; nn_train.py:319-323): centre crop [N,S,S] -> [N,L,L], x0 = y0 = (S - L) / 2
__global__ void k_center_crop(const float *__restrict__ src, int S, int L, long n_out, float *__restrict__ dst) {
    const int o = (S - L) / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % L), y = (int)((i / L) % L);
        const long img = i / ((long)L * L);
        dst[i] = src[(img * S + y + o) * S + x + o];
    }
}
// the gradient on the crop back onto the full output: zero outside the crop
__global__ void k_center_uncrop(const float *__restrict__ g, int S, int L, long n_full, float *__restrict__ dst) {
    const int o = (S - L) / 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_full; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S) - o, y = (int)((i / S) % S) - o;
        const long img = i / ((long)S * S);
        dst[i] = (x >= 0 && x < L && y >= 0 && y < L) ? g[(img * L + y) * L + x] : 0.f;
    }
}

// data gradient of the final Conv2d(f,3,1) + crop: g[c][b][Y][X] = sum_co gy[co][b][Y-crop][X-crop] * w[co][c] (0 outside)
__global__ void k_final_bwd_data(const float *__restrict__ gy, int S, const float *__restrict__ w, int cin, int crop,
                                 f32x4 *__restrict__ g, long gnp, int Hb, int Wb, int B) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = blockIdx.y;
    const int b = blockIdx.z % B, q = blockIdx.z / B;
    if (X >= Wb) return;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    const int yy = Y - crop, xx = X - crop;
    if (yy >= 0 && yy < S && xx >= 0 && xx < S) {
        const float *p = gy + ((long)b * 3 * S + yy) * S + xx;
        const float g0 = p[0], g1 = p[(long)S * S], g2 = p[2 * (long)S * S];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * q + e;
            if (c < cin) o[e] = g0 * w[c] + g1 * w[cin + c] + g2 * w[2 * cin + c];
        }
    }
    g[(long)q * gnp + ((long)b * Hb + Y) * Wb + X] = o;
}

// weight / bias gradient of the final 1x1, stage 1: one workgroup per (co, plane, image) -> partial[(co*planes+q)*B + b] (x,y,z,w = dw, then db)
__global__ __launch_bounds__(256) void k_final_wgrad1(const float *__restrict__ gy, int S, const f32x4 *__restrict__ act, long anp,
                                                      int Hb, int Wb, int crop, f32x4 *__restrict__ pw, float *__restrict__ pb) {
    __shared__ f32x4 red[256];
    __shared__ float redb[256];
    const int co = blockIdx.x, q = blockIdx.y, b = blockIdx.z;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    for (int i = threadIdx.x; i < S * S; i += 256) {
        const int y = i / S, x = i - y * S;
        const float gv = gy[(((long)b * 3 + co) * S + y) * S + x];
        acc += act[(long)q * anp + ((long)b * Hb + y + crop) * Wb + x + crop] * gv;
        accb += gv;
    }
    red[threadIdx.x] = acc;
    redb[threadIdx.x] = accb;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            red[threadIdx.x] += red[threadIdx.x + k];
            redb[threadIdx.x] += redb[threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const long o = ((long)co * gridDim.y + q) * gridDim.z + b;
        pw[o] = red[0];
        pb[o] = redb[0];
    }
}
__global__ __launch_bounds__(64) void k_final_wgrad2(const f32x4 *__restrict__ pw, const float *__restrict__ pb, int planes, int B,
                                                     int cin, float *__restrict__ dw, float *__restrict__ db) {
    const int co = blockIdx.x, q = blockIdx.y;
    const long o = ((long)co * planes + q) * B;
    if (threadIdx.x < 4 && 4 * q + (int)threadIdx.x < cin) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += pw[o + b][threadIdx.x];
        dw[(long)co * cin + 4 * q + threadIdx.x] = acc;
    }
    if (threadIdx.x == 4 && q == 0) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += pb[o + b];
        db[co] = acc;
    }
}

// Adam with amsgrad, torch.optim.Adam semantics (nn_common.py:185): no weight decay
__global__ void k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                       float *__restrict__ vmax, long n, float lr, float b1, float b2, float eps, float bc1, float bc2,
                       int amsgrad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float vd = vi;
    if (amsgrad) {
        vd = fmaxf(vmax[i], vi);
        vmax[i] = vd;
    }
    const float denom = sqrtf(vd) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * mi / denom;
}

namespace {

inline int transposed_kind(int kind) {
    return kind == ND_CONV3 ? ND_CONVT3 : (kind == ND_CONVT3 ? ND_CONV3 : (kind == ND_CONVT2S2 ? ND_CONV2S2 : -1));
}

// flat parameter layout = state-dict order (nd_utnet_tensor_name)
struct ParamLayout {
    std::vector<size_t> off, cnt;
    size_t total;
};
ParamLayout param_layout(int f) {
    ParamLayout pl;
    size_t o = 0;
    for (const std::string &name : tensor_names()) {
        size_t n = 1;   // PReLU slope
        for (int i = 0; i < kNumLayers; ++i) {
            const LayerSpec &l = kLayers[i];
            const int k = l.kind == ND_CONV1 ? 1 : (l.kind == ND_CONVT2S2 ? 2 : 3);
            if (name == std::string(l.key) + ".weight") n = (size_t)lcin(l, f) * lcout(l, f) * k * k;
            if (name == std::string(l.key) + ".bias") n = (size_t)lcout(l, f);
        }
        pl.off.push_back(o);
        pl.cnt.push_back(n);
        o += n;
    }
    pl.total = o;
    return pl;
}
// Gradient buckets for a data-parallel reduction that overlaps the backward pass: one per decoder / encoder level, numbered in
// the order the backward pass COMPLETES them (0: up4 + tconvs4, 1: up3 + tconvs3, 2: up2 + tconvs2, 3: up1 + tconvs1, 4: bottom,
// 5: convs4, 6: convs3, 7: convs2, 8: convs1).  Each is one contiguous range of the flat state-dict-order buffer.
constexpr int kNumBuckets = 9;
int bucket_of_key(const std::string &key) {
    static const char *const prefix[kNumBuckets][2] = {{"tconvs4", "up4"}, {"tconvs3", "up3"}, {"tconvs2", "up2"}, {"tconvs1", "up1"},
                                                        {"bottom", "bottom"}, {"convs4", "convs4"}, {"convs3", "convs3"},
                                                        {"convs2", "convs2"}, {"convs1", "convs1"}};
    for (int k = 0; k < kNumBuckets; ++k)
        for (int j = 0; j < 2; ++j) {
            const std::string p = prefix[k][j];
            if (key.compare(0, p.size(), p) == 0 && (key.size() == p.size() || key[p.size()] == '.')) return k;
        }
    return -1;
}
// the layer whose gradients finish a bucket: the bucket's first layer in forward order
int bucket_tail_layer(int k) {
    for (int i = 0; i < kNumLayers; ++i)
        if (bucket_of_key(kLayers[i].key) == k) return i;
    return -1;
}
// name of the PReLU tensor that follows layer `key` in its Sequential
std::string prelu_name(const char *key) {
    std::string k(key);
    const size_t dot = k.rfind('.');
    return k.substr(0, dot + 1) + std::to_string(atoi(k.c_str() + dot + 1) + 1) + ".weight";
}

struct BwdBlob {   // dgrad weights: per layer the transposed-kind packing (none for layer 0 and the final 1x1)
    size_t off[kNumLayers];
    size_t total;
};
BwdBlob bwd_blob_layout(int f) {
    BwdBlob b;
    size_t o = 0;
    for (int i = 0; i < kNumLayers; ++i) {
        b.off[i] = o;
        const LayerSpec &l = kLayers[i];
        if (i == 0 || l.kind == ND_CONV1) continue;
        const int kt = transposed_kind(l.kind);
        size_t n = nd_packed_floats(kt, lcout(l, f), lcin(l, f), ND_F32);
        if (kt == ND_CONV3 || kt == ND_CONVT3) n = std::max(n, nd_w1d_packed_floats(kW1dTile, lcout(l, f), lcin(l, f)));   // either packing
        o += n;
    }
    b.total = o;
    return b;
}

struct TrainPlan {
    Plan fwd;
    QpBuf pre[kNumSlopes];
    QpBuf g[NBUF];
    QpBuf scratch;          // re-pitched wgrad operand (largest over the layers)
    float *partial;         // wgrad K-slice partial sums
    size_t partial_floats;
    float *red;             // reduction scratch
    float *gy;              // d loss / d output  [B,3,S,S]
    float *yclip, *gssim;   // SSIM / MS-SSIM terms: clip(y, 0, 1) and the gradient with respect to it  [B,3,S,S]
    float *ycrop, *tcrop, *gcrop;   // loss_cs < cs: centre crops of output / target and the gradient on the crop  [B,3,L,L]
    char *ssim_ws;          // nd_ssim_loss_workspace_bytes(B, 3, S, S)
    size_t ssim_ws_bytes;
    size_t bytes;
};
constexpr int kRedFloats = 1 << 19;   // reduction scratch: >= 4 floats x planes x batch

int step_of_layer(int layer) {
    for (int i = 0; i < kNumSteps; ++i)
        if (kSteps[i].layer == layer) return i;
    return -1;
}
// pad of a gradient buffer: 2 when the tensor is produced by a Conv2d(3) layer (its data gradient is a transpose conv,
// which reads a zero-bordered input), else 0
int grad_pad(Buf id) {
    switch (id) {
        case A1: case A2: case A3: case A4: case BT0: case CAT1: case CAT2: case CAT3: case CAT4: return 2;
        default: return 0;
    }
}

TrainPlan make_train_plan(int f, int cs, int B, char *base) {
    TrainPlan t;
    t.fwd = make_plan(f, cs, cs, B, B, base, ND_F32);
    size_t off = t.fwd.bytes;
    auto alloc = [&](QpBuf &q, int planes, int Hb, int Wb, int pad) {
        q.planes = planes;
        q.B = B;
        q.Hb = Hb;
        q.Wb = Wb;
        q.pad = pad;
        q.dt = ND_F32;
        q.pstride = (long)B * Hb * Wb;
        q.base = (float *)(base ? base + off : nullptr);
        off += ((size_t)planes * q.pstride + nd_buf_slack(Wb)) * 16;
        off = (off + 255) & ~(size_t)255;
    };
    // pre-activation copies: compact, the layer's output size
    for (int i = 0; i < kNumLayers; ++i) {
        const LayerSpec &l = kLayers[i];
        if (l.prelu < 0) continue;
        const QpBuf &o = t.fwd.buf[kSteps[step_of_layer(i)].dst];
        alloc(t.pre[l.prelu], lcout(l, f) / 4, o.Hb - 2 * o.pad, o.Wb - 2 * o.pad, 0);
    }
    // gradient buffers
    size_t scratch_elems = 0;
    for (int id = 1; id < NBUF; ++id) {
        const QpBuf &o = t.fwd.buf[id];
        const int pad = grad_pad((Buf)id);
        alloc(t.g[id], o.planes, o.Hb - 2 * o.pad + 2 * pad, o.Wb - 2 * o.pad + 2 * pad, pad);
    }
    t.g[X0] = QpBuf();
    // scratch + wgrad partials: maxima over the layers
    size_t pf = 0;
    for (int i = 0; i < kNumLayers - 1; ++i) {
        const LayerSpec &l = kLayers[i];
        const Step &st = kSteps[step_of_layer(i)];
        const QpBuf &in = t.fwd.buf[st.src], &out = t.fwd.buf[st.dst];
        const int ih = in.Hb - 2 * in.pad, iw = in.Wb - 2 * in.pad, oh = out.Hb - 2 * out.pad, ow = out.Wb - 2 * out.pad;
        const int ci = lcin(l, f), co = lcout(l, f);
        size_t e = 0, p = 0;
        if (l.kind == ND_CONV3) {          // A = g(out) on the input grid
            e = (size_t)(co / 4) * B * in.Hb * in.Wb;
            p = nd_wgrad_partial_floats(9, co, ci, (long)B * in.Hb * in.Wb, nullptr, nullptr);
        } else if (l.kind == ND_CONVT3) {  // A = input interior on the output grid
            e = (size_t)((ci + 3) / 4) * B * oh * ow;
            p = nd_wgrad_partial_floats(9, ci, co, (long)B * oh * ow, nullptr, nullptr);
        } else {                           // up: B = one phase of g(out) on the input grid
            e = (size_t)(co / 4) * B * ih * iw;
            p = nd_wgrad_partial_floats(1, ci, co, (long)B * ih * iw, nullptr, nullptr);
        }
        if (e > scratch_elems) scratch_elems = e;
        if (p > pf) pf = p;
    }
    t.scratch = QpBuf();
    t.scratch.base = (float *)(base ? base + off : nullptr);
    off += (scratch_elems + 4096) * 16;
    off = (off + 255) & ~(size_t)255;
    t.partial = (float *)(base ? base + off : nullptr);
    t.partial_floats = pf;
    off += pf * 4;
    off = (off + 255) & ~(size_t)255;
    t.red = (float *)(base ? base + off : nullptr);
    off += kRedFloats * 4;
    t.gy = (float *)(base ? base + off : nullptr);
    off += (size_t)B * 3 * cs * cs * 4;
    off = (off + 255) & ~(size_t)255;
    t.yclip = (float *)(base ? base + off : nullptr);
    off += ((size_t)B * 3 * cs * cs * 4 + 255) & ~(size_t)255;
    t.gssim = (float *)(base ? base + off : nullptr);
    off += ((size_t)B * 3 * cs * cs * 4 + 255) & ~(size_t)255;
    for (float **pp : {&t.ycrop, &t.tcrop, &t.gcrop}) {
        *pp = (float *)(base ? base + off : nullptr);
        off += ((size_t)B * 3 * cs * cs * 4 + 255) & ~(size_t)255;
    }
    t.ssim_ws = base ? base + off : nullptr;
    t.ssim_ws_bytes = nd_ssim_loss_workspace_bytes(B, 3, cs, cs);
    off += (t.ssim_ws_bytes + 255) & ~(size_t)255;
    t.bytes = off;
    return t;
}

QpBuf scratch_view(const TrainPlan &t, int planes, int B, int H, int W) {
    QpBuf q = t.scratch;
    q.planes = planes;
    q.B = B;
    q.Hb = H;
    q.Wb = W;
    q.pad = 0;
    q.dt = ND_F32;
    q.pstride = (long)B * H * W;
    return q;
}

int check_train(int funit, int cs, int batch) {
    if (funit < 8 || funit % 8) ND_FAIL(ND_EINVAL, "UtNet training: funit=%d must be a positive multiple of 8", funit);
    if (!valid_cs(cs)) ND_FAIL(ND_EINVAL, "UtNet training: crop size %d is not of the form 16k+56 (e.g. 136, 184)", cs);
    if (batch <= 0) ND_FAIL(ND_EINVAL, "UtNet training: batch=%d", batch);
    return ND_OK;
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" size_t nd_utnet_param_count(int funit) { return (funit < 8 || funit % 8) ? 0 : param_layout(funit).total; }
extern "C" int nd_utnet_param_range(int funit, int tensor_idx, size_t *offset, size_t *count) {
    if (funit < 8 || funit % 8) ND_FAIL(ND_EINVAL, "bad funit");
    const ParamLayout pl = param_layout(funit);
    if (tensor_idx < 0 || tensor_idx >= (int)pl.off.size()) ND_FAIL(ND_EINVAL, "tensor index %d out of range", tensor_idx);
    if (offset) *offset = pl.off[tensor_idx];
    if (count) *count = pl.cnt[tensor_idx];
    return ND_OK;
}
extern "C" size_t nd_utnet_train_blob_bytes(int funit) {
    if (funit < 8 || funit % 8) return 0;
    return (blob_layout(funit, ND_F32, false, true).total + bwd_blob_layout(funit).total) * sizeof(float);
}
extern "C" size_t nd_utnet_train_workspace_bytes(int funit, int cs, int batch) {
    if (check_train(funit, cs, batch) != ND_OK) return 0;
    return make_train_plan(funit, cs, batch, nullptr).bytes;
}
extern "C" int nd_utnet_train_workspace_init(void *ws, size_t ws_bytes, int funit, int cs, int batch, void *stream) {
    ND_TRY(check_train(funit, cs, batch));
    const size_t need = make_train_plan(funit, cs, batch, nullptr).bytes;
    if (!ws || ws_bytes < need) ND_FAIL(ND_ENOMEM, "UtNet training workspace: %zu B given, %zu B needed", ws_bytes, need);
    ND_HIP(hipMemsetAsync(ws, 0, need, (hipStream_t)stream));   // zero borders of activations AND gradients, slack
    return ND_OK;
}

// ---- the step in two halves: (1) weight packing + forward with the pre-activations kept, (2) backward from d loss / d output.
// nd_utnet_train_step runs both with the loss between them; nd_utnet_train_forward / nd_utnet_train_backward expose the halves
// to torch.autograd (networks/UtNet.py: model(x).clip(0, 1), loss.backward() of nn_common.py:198-218 then work unchanged).
struct TrainCtx {
    int f, B, cs, flags, act;
    TrainPlan t;
    ParamLayout pl;
    BlobLayout bl;
    BwdBlob bb;
    float *fblob, *bblob;
    unsigned char fwd_w1[kNumLayers], bwd_w1[kNumLayers];
    const float *slopes[kNumSlopes];
    hipStream_t s;
};

static int train_ctx(TrainCtx &c, int funit, int flags, int act, const float *params, void *blobs, int batch, int cs, void *ws,
                     size_t ws_bytes, void *stream) {
    ND_TRY(nd_check_flags(flags));
    ND_TRY(check_train(funit, cs, batch));
    if (act < ND_ACT_PRELU || act > ND_ACT_HARDSWISH) ND_FAIL(ND_EINVAL, "UtNet training: unknown activation %d", act);
    if (!params || !blobs || !ws) ND_FAIL(ND_EINVAL, "train step: null pointer");
    c.f = funit;
    c.B = batch;
    c.cs = cs;
    c.flags = flags;
    c.act = act;
    c.t = make_train_plan(funit, cs, batch, (char *)ws);
    if (ws_bytes < c.t.bytes) ND_FAIL(ND_ENOMEM, "UtNet training workspace: %zu B given, %zu B needed", ws_bytes, c.t.bytes);
    c.s = (hipStream_t)stream;
    c.pl = param_layout(funit);
    c.bl = blob_layout(funit, ND_F32, false, true);
    // 3x3 layers whose rows fit its LDS images run the fused 1-D Winograd kernel, forward and data gradient (else the direct one)
    memset(c.fwd_w1, 0, sizeof(c.fwd_w1));
    memset(c.bwd_w1, 0, sizeof(c.bwd_w1));
    for (const Step &st : kSteps) {
        if (st.layer < 0) continue;
        const LayerSpec &l = kLayers[st.layer];
        if (l.kind != ND_CONV3 && l.kind != ND_CONVT3) continue;
        if (!(flags & ND_FLAG_DIRECT_CONV)) {
            c.fwd_w1[st.layer] = nd_w1d_fits(kW1dTile, c.t.fwd.buf[st.src]);
            c.bwd_w1[st.layer] = st.layer > 0 && nd_w1d_fits(kW1dTile, c.t.g[st.dst]);
        }
    }
    c.bb = bwd_blob_layout(funit);
    c.fblob = (float *)blobs;
    c.bblob = c.fblob + c.bl.total;
    for (int k = 0; k < kNumSlopes; ++k) c.slopes[k] = nullptr;
    for (int i = 0; i < kNumLayers; ++i)
        if (kLayers[i].prelu >= 0) c.slopes[kLayers[i].prelu] = params + c.pl.off[tensor_index(prelu_name(kLayers[i].key))];
    return ND_OK;
}

// (1) pack the weights on the device (forward roles, and transposed roles for the data gradients), forward with the
// pre-activations kept; y_out: [batch,3,cs,cs]
static int train_forward(TrainCtx &c, const float *params, const float *x, float *y_out) {
    const int f = c.f, B = c.B, cs = c.cs;
    hipStream_t s = c.s;
    const BlobLayout &bl = c.bl;
    const BwdBlob &bb = c.bb;
    float *fblob = c.fblob, *bblob = c.bblob;
    auto P = [&](const std::string &name) -> const float * { return params + c.pl.off[tensor_index(name)]; };
    for (int i = 0; i < kNumLayers; ++i) {
        const LayerSpec &l = kLayers[i];
        const int ci = lcin(l, f), co = lcout(l, f);
        const float *w = P(std::string(l.key) + ".weight"), *b = P(std::string(l.key) + ".bias");
        if (l.kind == ND_CONV1) {
            ND_HIP(hipMemcpyAsync(fblob + bl.off[i], w, sizeof(float) * 3 * ci, hipMemcpyDeviceToDevice, s));
            ND_HIP(hipMemcpyAsync(fblob + bl.off[i] + 3 * ci, b, sizeof(float) * 3, hipMemcpyDeviceToDevice, s));
            continue;
        }
        if (c.fwd_w1[i])
            nd_pack_w1d_device(kW1dTile, l.kind, ci, co, w, b, fblob + bl.off[i], s);
        else
            nd_pack_layer_device(l.kind, ci, co, w, b, fblob + bl.off[i], s);
        if (i > 0) {   // transposed role: cin' = co, cout' = ci, no bias
            const int kt = transposed_kind(l.kind);
            if (c.bwd_w1[i])
                nd_pack_w1d_device(kW1dTile, kt, co, ci, w, nullptr, bblob + bb.off[i], s);
            else
                nd_pack_layer_device(kt, co, ci, w, nullptr, bblob + bb.off[i], s);
        }
    }
    ND_HIP(hipGetLastError());
    // slope table of the forward blob header (PReLU only: the other activations have no parameters)
    if (c.act == ND_ACT_PRELU)
        for (int k = 0; k < kNumSlopes; ++k)
            ND_HIP(hipMemcpyAsync(fblob + k, c.slopes[k], sizeof(float), hipMemcpyDeviceToDevice, s));
    ND_TRY(nd_launch_reflect_pack(x, B, cs, cs, c.t.fwd.buf[X0], s));
    ND_TRY(run_stack(f, c.act, ND_F32, fblob, c.t.fwd, s, c.flags, nullptr, c.t.pre, nullptr, c.fwd_w1));
    const float *fw = fblob + bl.off[kNumLayers - 1];
    ND_TRY(nd_launch_final1x1(c.t.fwd.buf[T4B], f, fw, fw + 3 * f, 2, y_out, cs, cs, s));
    return ND_OK;
}

// (2) backward: gy = d loss / d output [batch,3,cs,cs]; every parameter gradient into the flat buffer `grads`
static int train_backward(TrainCtx &c, float *grads, const float *gy, void *const *bucket_ev = nullptr) {
    const int f = c.f, B = c.B, cs = c.cs, flags = c.flags;
    hipStream_t s = c.s;
    TrainPlan &t = c.t;
    const BlobLayout &bl = c.bl;
    const BwdBlob &bb = c.bb;
    float *fblob = c.fblob, *bblob = c.bblob;
    const float *const *slopes = c.slopes;
    auto G = [&](const std::string &name) -> float * { return grads + c.pl.off[tensor_index(name)]; };
    const float *fw = fblob + bl.off[kNumLayers - 1];
    // ---- 4. backward
    // final 1x1
    {
        const QpBuf &a = t.fwd.buf[T4B], &g = t.g[T4B];
        f32x4 *pw = (f32x4 *)t.red;
        float *pb = t.red + (size_t)4 * 3 * (f / 4) * B;
        hipLaunchKernelGGL(k_final_wgrad1, dim3(3, f / 4, B), dim3(256), 0, s, gy, cs, (const f32x4 *)a.base,
                           a.np(), a.Hb, a.Wb, 2, pw, pb);
        hipLaunchKernelGGL(k_final_wgrad2, dim3(3, f / 4), dim3(64), 0, s, (const f32x4 *)pw, (const float *)pb, f / 4, B, f,
                           G("tconvs4.4.weight"), G("tconvs4.4.bias"));
        dim3 grid((g.Wb + 127) / 128, g.Hb, B * (f / 4));
        hipLaunchKernelGGL(k_final_bwd_data, grid, dim3(128), 0, s, gy, cs, fw, f, 2, (f32x4 *)g.base, g.np(), g.Hb,
                           g.Wb, B);
        ND_HIP(hipGetLastError());
    }
    for (int si = kNumSteps - 1; si >= 0; --si) {
        const Step &st = kSteps[si];
        if (st.layer < 0) {
            // pool: the skip half of the concat buffer was pooled into P; route g(P) back and ADD it to g(skip)
            const int planes = st.dst_plane0_mul * f / 4, plane0 = planes;
            const QpBuf &gp = t.g[st.dst], &fw_ = t.fwd.buf[st.src], &gc = t.g[st.src];
            const int Ho = gp.Hb - 2 * gp.pad, Wo = gp.Wb - 2 * gp.pad;
            dim3 grid((Wo + 127) / 128, Ho, B * planes);
            hipLaunchKernelGGL(k_maxpool_bwd_add, grid, dim3(128), 0, s, (const f32x4 *)gp.base, gp.np(), gp.Hb, gp.Wb, gp.pad,
                               (const f32x4 *)fw_.base + (long)plane0 * fw_.np(), fw_.np(), fw_.Hb, fw_.Wb, fw_.pad,
                               (f32x4 *)gc.base + (long)plane0 * gc.np(), gc.np(), gc.Hb, gc.Wb, gc.pad, Ho, Wo, B);
            ND_HIP(hipGetLastError());
            continue;
        }
        const LayerSpec &l = kLayers[st.layer];
        const int ci = lcin(l, f), co = lcout(l, f);
        const QpBuf &in = t.fwd.buf[st.src];    // layer input (forward values)
        const QpBuf &go = t.g[st.dst];          // gradient of the layer's output buffer
        const int oplane0 = st.dst_plane0_mul * f / 4, oplanes = co / 4;
        const int oh = go.Hb - 2 * go.pad, ow = go.Wb - 2 * go.pad;
        const int ih = in.Hb - 2 * in.pad, iw = in.Wb - 2 * in.pad;
        // activation backward (in place) + slope gradient
        if (l.prelu >= 0) {
            const QpBuf &pr = t.pre[l.prelu];
            f32x4 *gq = (f32x4 *)go.base + (long)oplane0 * go.np();
            if (c.act == ND_ACT_PRELU) {
                hipLaunchKernelGGL(k_act_bwd<ND_ACT_PRELU>, dim3(B, oplanes), dim3(256), 0, s, gq, go.np(), go.Hb, go.Wb, go.pad,
                                   (const f32x4 *)pr.base, pr.np(), oh, ow, slopes[l.prelu], t.red);
                hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, (const float *)t.red, B * oplanes, 1.f, G(prelu_name(l.key)));
            } else if (c.act == ND_ACT_ELU) {
                hipLaunchKernelGGL(k_act_bwd<ND_ACT_ELU>, dim3(B, oplanes), dim3(256), 0, s, gq, go.np(), go.Hb, go.Wb, go.pad,
                                   (const f32x4 *)pr.base, pr.np(), oh, ow, (const float *)nullptr, t.red);
            } else {
                hipLaunchKernelGGL(k_act_bwd<ND_ACT_HARDSWISH>, dim3(B, oplanes), dim3(256), 0, s, gq, go.np(), go.Hb, go.Wb, go.pad,
                                   (const f32x4 *)pr.base, pr.np(), oh, ow, (const float *)nullptr, t.red);
            }
            ND_HIP(hipGetLastError());
        }
        // bias gradient
        ND_TRY(nd_launch_channel_sum(go, oplane0, co, G(std::string(l.key) + ".bias"), t.red, s));
        // weight gradient
        float *dw = G(std::string(l.key) + ".weight");
        if (l.kind == ND_CONV3) {
            QpBuf a = scratch_view(t, oplanes, B, in.Hb, in.Wb);
            ND_TRY(nd_launch_repitch(go, oplane0, oplanes, 1, 0, 0, a, 0, 0, oh, ow, s));
            ND_TRY(nd_launch_wgrad(a, 0, co, in, 0, ci, 9, 9, 0, t.partial, t.partial_floats, dw, s));
        } else if (l.kind == ND_CONVT3) {
            QpBuf a = scratch_view(t, (ci + 3) / 4, B, oh, ow);
            ND_TRY(nd_launch_repitch(in, 0, (ci + 3) / 4, 1, 0, 0, a, 0, 0, ih, iw, s));
            QpBuf gb = go;   // pad 0 by construction: the whole buffer is the grid
            ND_TRY(nd_launch_wgrad(a, 0, ci, gb, oplane0, co, 9, 9, 0, t.partial, t.partial_floats, dw, s));
        } else {   // ConvTranspose2d(2, s=2): four 1-tap problems on the input grid
            for (int ab = 0; ab < 4; ++ab) {
                QpBuf bq = scratch_view(t, oplanes, B, ih, iw);
                ND_TRY(nd_launch_repitch(go, oplane0, oplanes, 2, ab >> 1, ab & 1, bq, 0, 0, ih, iw, s));
                ND_TRY(nd_launch_wgrad(in, 0, ci, bq, 0, co, 1, 4, ab, t.partial, t.partial_floats, dw, s));
            }
        }
        // data gradient into g(input buffer): a forward launch of the transposed kind
        if (st.layer > 0) {
            ConvDesc d;
            d.kind = transposed_kind(l.kind);
            d.act = ND_ACT_NONE;
            d.slope = 1.f;
            d.slope_dev = nullptr;
            d.cin = co;
            d.cout = ci;
            d.wpk = bblob + bb.off[st.layer];
            d.bias = d.wpk + (size_t)nd_mtiles(d.kind, ci) * nd_kblocks(co) * nd_taps(d.kind) * 256;
            d.in = go;
            d.in_plane0 = oplane0;
            d.out = t.g[st.src];
            d.out_plane0 = 0;
            d.variant = -1;
            d.part = t.fwd.split;
            d.part_bytes = kSplitScratchBytes;
            d.nosplit = (flags & ND_FLAG_NO_SPLITK) != 0;
            if (c.bwd_w1[st.layer]) {
                d.bias = d.wpk + (size_t)nd_mtiles(ND_CONV3, ci) * nd_kblocks(co) * 18 * 256;
                // (a data gradient keeps no pre-activation copy: the inference form with the LDS-shared transform applies)
                if (nd_w2d_ok(d.in) && !(flags & ND_FLAG_W1D_REGS))
                    ND_TRY(nd_launch_conv_w2d(d, s));
                else
                    ND_TRY(nd_launch_conv_w1d(kW1dTile, d, s));
            } else {
                ND_TRY(nd_launch_conv(d, s));
            }
        }
        // every parameter gradient of this layer's level is final once the level's first layer is done: tell the reducer
        if (bucket_ev) {
            const int k = bucket_of_key(l.key);
            if (k >= 0 && bucket_tail_layer(k) == st.layer) ND_HIP(hipEventRecord((hipEvent_t)bucket_ev[k], s));
        }
    }
    return ND_OK;
}

// One training step without the optimizer: packs the weights on the device, runs forward (PReLU only), the loss
//   loss = w_l1 * mean|g - target| + w_mse * mean (g - target)^2 + w_ssim * mean_n(1 - SSIM_n(g, target))
//          + w_msssim * mean_n(1 - MS-SSIM_n(g, target)),      g = clip(y, 0, 1)          (nn_common.py:198-199, 226-241)
// and the backward pass.  params / grads: flat fp32 buffers in state-dict order (nd_utnet_param_range);
// x, target, y_out: [batch,3,cs,cs] NCHW fp32; loss_out: one float in HBM; blobs: nd_utnet_train_blob_bytes scratch.
static int train_step_impl(int funit, int flags, const float *params, float *grads, void *blobs, const float *x,
                           const float *target, float *y_out, float w_l1, float w_mse, float w_ssim, float w_msssim,
                           float *loss_out, int batch, int cs, int loss_cs, void *ws, size_t ws_bytes, void *stream,
                           void *const *bucket_ev) {
    const int L = loss_cs > 0 ? loss_cs : cs;   // the criteria see the centre crop of this size (nn_train.py:319-323)
    if (L > cs) ND_FAIL(ND_EINVAL, "UtNet training: loss_cs=%d exceeds the crop size %d", L, cs);
    if (w_msssim != 0.f && L < 161)
        ND_FAIL(ND_EINVAL, "UtNet training: the MS-SSIM loss needs crops of at least 161 pixels (five scales of an 11-tap window), "
                           "got %d; the reference fails on them too (pt_losses.py:20-28)", L);
    if (w_ssim != 0.f && L < 11) ND_FAIL(ND_EINVAL, "UtNet training: the SSIM loss needs at least 11 pixels, got %d", L);
    if (!grads || !x || !target || !y_out || !loss_out) ND_FAIL(ND_EINVAL, "train step: null pointer");
    TrainCtx c;
    ND_TRY(train_ctx(c, funit, flags, ND_ACT_PRELU, params, blobs, batch, cs, ws, ws_bytes, stream));
    ND_TRY(train_forward(c, params, x, y_out));
    TrainPlan &t = c.t;
    hipStream_t s = c.s;
    const int B = batch;
    // ---- 3. loss and its gradient, on the centre crop of loss_cs pixels (the whole output when loss_cs == cs)
    const long nfull = (long)B * 3 * cs * cs, nout = (long)B * 3 * L * L;
    const int lblocks = 1024;
    const float *yl = y_out, *tl = target;
    float *gl = t.gy;
    if (L != cs) {
        hipLaunchKernelGGL(k_center_crop, dim3(1024), dim3(256), 0, s, (const float *)y_out, cs, L, nout, t.ycrop);
        hipLaunchKernelGGL(k_center_crop, dim3(1024), dim3(256), 0, s, target, cs, L, nout, t.tcrop);
        yl = t.ycrop;
        tl = t.tcrop;
        gl = t.gcrop;
    }
    hipLaunchKernelGGL(k_loss_grad, dim3(lblocks), dim3(256), 0, s, yl, tl, nout, w_l1, w_mse, gl, t.red);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, (const float *)t.red, lblocks, 1.f / (float)nout, loss_out);
    ND_HIP(hipGetLastError());
    if (w_ssim != 0.f || w_msssim != 0.f) {
        hipLaunchKernelGGL(k_clip01, dim3(1024), dim3(256), 0, s, yl, nout, t.yclip);
        int acc = 0;
        if (w_ssim != 0.f) {
            ND_TRY(nd_ssim_loss_grad(t.yclip, tl, B, 3, L, L, 0, w_ssim, loss_out, t.gssim, acc, t.ssim_ws, t.ssim_ws_bytes, s));
            acc = 1;
        }
        if (w_msssim != 0.f)
            ND_TRY(nd_ssim_loss_grad(t.yclip, tl, B, 3, L, L, 1, w_msssim, loss_out, t.gssim, acc, t.ssim_ws, t.ssim_ws_bytes, s));
        hipLaunchKernelGGL(k_add_clip_grad, dim3(1024), dim3(256), 0, s, yl, (const float *)t.gssim, nout, gl);
        ND_HIP(hipGetLastError());
    }
    if (L != cs) {
        hipLaunchKernelGGL(k_center_uncrop, dim3(1024), dim3(256), 0, s, (const float *)t.gcrop, cs, L, nfull, t.gy);
        ND_HIP(hipGetLastError());
    }

    return train_backward(c, grads, t.gy, bucket_ev);
}
extern "C" int nd_utnet_train_step(int funit, int flags, const float *params, float *grads, void *blobs, const float *x,
                                   const float *target, float *y_out, float w_l1, float w_mse, float w_ssim, float w_msssim,
                                   float *loss_out, int batch, int cs, int loss_cs, void *ws, size_t ws_bytes, void *stream) {
    return train_step_impl(funit, flags, params, grads, blobs, x, target, y_out, w_l1, w_mse, w_ssim, w_msssim, loss_out, batch, cs,
                           loss_cs, ws, ws_bytes, stream, nullptr);
}
// The same step for a data-parallel run that overlaps the gradient reduction with the backward pass: bucket_events[k]
// (hipEvent_t, nd_utnet_grad_buckets of them) is recorded on `stream` as soon as every gradient of bucket k is final.
extern "C" int nd_utnet_train_step_ev(int funit, int flags, const float *params, float *grads, void *blobs, const float *x,
                                      const float *target, float *y_out, float w_l1, float w_mse, float w_ssim, float w_msssim,
                                      float *loss_out, int batch, int cs, int loss_cs, void *ws, size_t ws_bytes, void *stream,
                                      void *const *bucket_events, int n_events) {
    if (!bucket_events || n_events != kNumBuckets) ND_FAIL(ND_EINVAL, "train step: %d bucket events expected", kNumBuckets);
    return train_step_impl(funit, flags, params, grads, blobs, x, target, y_out, w_l1, w_mse, w_ssim, w_msssim, loss_out, batch, cs,
                           loss_cs, ws, ws_bytes, stream, bucket_events);
}
// Buckets of the flat gradient buffer in the order the backward pass completes them (one per decoder / encoder level):
// offsets / counts in floats.  Returns the number of buckets (9); fills at most `max` entries.
extern "C" int nd_utnet_grad_buckets(int funit, size_t *offsets, size_t *counts, int max) {
    if (funit < 8 || funit % 8) ND_FAIL(ND_EINVAL, "bad funit");
    const ParamLayout pl = param_layout(funit);
    const auto &names = tensor_names();
    for (int k = 0; k < kNumBuckets && k < max; ++k) {
        size_t lo = (size_t)-1, hi = 0;
        for (size_t i = 0; i < names.size(); ++i) {
            const std::string key = names[i].substr(0, names[i].find('.') == std::string::npos ? names[i].size() : names[i].find('.'));
            if (bucket_of_key(key) != k) continue;
            if (pl.off[i] < lo) lo = pl.off[i];
            if (pl.off[i] + pl.cnt[i] > hi) hi = pl.off[i] + pl.cnt[i];
        }
        if (offsets) offsets[k] = lo;
        if (counts) counts[k] = hi - lo;
    }
    return kNumBuckets;
}

// The two halves for torch.autograd (act: ND_ACT_PRELU | ND_ACT_ELU | ND_ACT_HARDSWISH, the reference constructor's choices,
// networks/UtNet.py:17-26).  `ws` and `blobs` carry the forward's activations, pre-activations and packed weights to the
// backward call: the caller keeps both untouched in between.  The input's own gradient is not produced (the first layer's data
// gradient is skipped, as in the fused step): images are not trained.
extern "C" int nd_utnet_train_forward(int funit, int act, int flags, const float *params, void *blobs, const float *x, float *y_out,
                                      int batch, int cs, void *ws, size_t ws_bytes, void *stream) {
    if (!x || !y_out) ND_FAIL(ND_EINVAL, "train forward: null pointer");
    TrainCtx c;
    ND_TRY(train_ctx(c, funit, flags, act, params, blobs, batch, cs, ws, ws_bytes, stream));
    return train_forward(c, params, x, y_out);
}
extern "C" int nd_utnet_train_backward(int funit, int act, int flags, const float *params, float *grads, void *blobs, const float *gy,
                                       int batch, int cs, void *ws, size_t ws_bytes, void *stream, void *const *bucket_events,
                                       int n_events) {
    if (!grads || !gy) ND_FAIL(ND_EINVAL, "train backward: null pointer");
    if (bucket_events && n_events != kNumBuckets) ND_FAIL(ND_EINVAL, "train backward: %d bucket events expected", kNumBuckets);
    TrainCtx c;
    ND_TRY(train_ctx(c, funit, flags, act, params, blobs, batch, cs, ws, ws_bytes, stream));
    return train_backward(c, grads, gy, bucket_events);
}

// torch.optim.Adam(params, lr, betas=(b1,b2), eps, amsgrad) on flat buffers; step = 1, 2, ...
extern "C" int nd_adam_step(float *params, const float *grads, float *m, float *v, float *vmax, size_t n, float lr, float b1,
                            float b2, float eps, int step, int amsgrad, void *stream) {
    if (!params || !grads || !m || !v || (amsgrad && !vmax) || step < 1) ND_FAIL(ND_EINVAL, "nd_adam_step: bad arguments");
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, vmax,
                       (long)n, lr, b1, b2, eps, bc1, bc2, amsgrad);
    ND_HIP(hipGetLastError());
    return ND_OK;
}
