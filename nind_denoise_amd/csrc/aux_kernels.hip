// HBM-bound helpers around the conv stack: tile geometry, tile gather (+mirror), stitch, layout conversion,
// max-pool, the final 1x1 convolution.  All are pure copies / max / short dot products: the roofline that bounds them
// is HBM bandwidth, so they read and write 16 B (or one full pixel row segment) per lane, coalesced along x.
#include "nd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// One 16-byte plane element = the channels of one pixel in one plane: 4 x fp32, 8 x bf16 or 8 x fp16.
template <int DT> struct Elem { typedef f32x4 vec; static constexpr int N = 4; typedef float scalar; };
template <> struct Elem<ND_BF16> { typedef bf16x8 vec; static constexpr int N = 8; typedef __bf16 scalar; };
template <> struct Elem<ND_F16> { typedef f16x8 vec; static constexpr int N = 8; typedef _Float16 scalar; };
#define ND_DISPATCH_DT(dt, CALL)                     \
    switch (dt) {                                    \
        case ND_F32: { constexpr int DT = ND_F32; CALL; break; }   \
        case ND_BF16: { constexpr int DT = ND_BF16; CALL; break; } \
        case ND_F16: { constexpr int DT = ND_F16; CALL; break; }   \
        default: ND_FAIL(ND_EINVAL, "unsupported storage type %d", dt); \
    }

// ------------------------------------------------------------------ tile geometry (OneImageDS, denoise_image.py:100-143)
struct TileGeo {
    int W, H, cs, ucs, ol, pad, stride, cols, rows;
};

static __host__ __device__ inline int ceil_div_py(int a, int b) {  // math.ceil(a / b) for b > 0, any sign of a
    return a >= 0 ? (a + b - 1) / b : -((-a) / b);
}

static int make_geo(int W, int H, int cs, int ucs, int ol, TileGeo *g) {
    if (W <= 0 || H <= 0 || cs <= 0 || ucs <= 0 || ol < 0) ND_FAIL(ND_EINVAL, "tile grid: non-positive size");
    if (ucs - ol <= 0) ND_FAIL(ND_EINVAL, "tile grid: ucs (%d) must exceed the overlap (%d)", ucs, ol);
    if (cs < ucs) ND_FAIL(ND_EINVAL, "tile grid: cs (%d) < ucs (%d)", cs, ucs);
    g->W = W; g->H = H; g->cs = cs; g->ucs = ucs; g->ol = ol;
    g->stride = ucs - ol;
    g->pad = (cs - ucs) / 2;                                 // int((cs-ucs)/2), denoise_image.py:102
    g->cols = ceil_div_py(W - ucs, g->stride) + 1;           // iperhl + 1, :101
    g->rows = ceil_div_py(H - ucs, g->stride) + 1;           // ipervl + 1, :103
    if (g->cols <= 0 || g->rows <= 0) ND_FAIL(ND_EINVAL, "tile grid: image %dx%d smaller than ucs=%d (undefined in the reference)", W, H, ucs);
    // mirrored strips must come from inside the image (the reference's numpy slices fail otherwise)
    const int x1pad = (g->cols - 1) * g->stride - g->pad + cs - W;
    const int y1pad = (g->rows - 1) * g->stride - g->pad + cs - H;
    if (g->pad > W || g->pad > H || x1pad > W || y1pad > H)
        ND_FAIL(ND_EINVAL, "tile grid: mirror padding (%d,%d,%d) exceeds the image %dx%d", g->pad, x1pad, y1pad, W, H);
    return ND_OK;
}

extern "C" int nd_tile_grid(int W, int H, int cs, int ucs, int ol, int *cols, int *rows, int *pad) {
    TileGeo g;
    ND_TRY(make_geo(W, H, cs, ucs, ol, &g));
    if (cols) *cols = g.cols;
    if (rows) *rows = g.rows;
    if (pad) *pad = g.pad;
    return ND_OK;
}

extern "C" int nd_tile_geom(int i, int W, int H, int cs, int ucs, int ol, int *x0, int *y0, int ud[4], int us[2]) {
    TileGeo g;
    ND_TRY(make_geo(W, H, cs, ucs, ol, &g));
    if (i < 0 || i >= g.cols * g.rows) ND_FAIL(ND_EINVAL, "tile index %d outside [0,%d)", i, g.cols * g.rows);
    const int yi = i / g.cols, xi = i - yi * g.cols;  // == int(ceil((i+1)/(iperhl+1) - 1)), :131-132
    const int tx0 = xi * g.stride - g.pad, ty0 = yi * g.stride - g.pad;
    const int x1pad = tx0 + cs - W > 0 ? tx0 + cs - W : 0;
    const int y1pad = ty0 + cs - H > 0 ? ty0 + cs - H : 0;
    if (x0) *x0 = tx0;
    if (y0) *y0 = ty0;
    if (ud) {
        ud[0] = g.pad;
        ud[1] = g.pad;
        ud[2] = cs - (g.pad > x1pad ? g.pad : x1pad);
        ud[3] = cs - (g.pad > y1pad ? g.pad : y1pad);
    }
    if (us) {
        us[0] = tx0 + g.pad;
        us[1] = ty0 + g.pad;
    }
    return ND_OK;
}

__device__ __forceinline__ int mirror_sym(int v, int n) {  // edge pixel repeated (np.flip of the adjacent band)
    return v < 0 ? -1 - v : (v >= n ? 2 * n - 1 - v : v);
}
__device__ __forceinline__ int reflect_nr(int v, int n) {  // nn.ReflectionPad2d: edge pixel NOT repeated
    return v < 0 ? -v : (v >= n ? 2 * (n - 1) - v : v);
}

// ------------------------------------------------------------------ gather: image CHW -> tiles NCHW
__global__ void k_tile_gather(const float *__restrict__ img, TileGeo g, int tile_begin, float *__restrict__ out) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    const int t = blockIdx.z;
    if (xx >= g.cs) return;
    const int i = tile_begin + t;
    const int yi = i / g.cols, xi = i - yi * g.cols;
    const int sx = mirror_sym(xi * g.stride - g.pad + xx, g.W);
    const int sy = mirror_sym(yi * g.stride - g.pad + yy, g.H);
    const size_t plane = (size_t)g.W * g.H;
    const size_t tplane = (size_t)g.cs * g.cs;
    const float *s = img + (size_t)sy * g.W + sx;
    float *d = out + ((size_t)t * 3) * tplane + (size_t)yy * g.cs + xx;
    d[0] = s[0];
    d[tplane] = s[plane];
    d[2 * tplane] = s[2 * plane];
}

extern "C" int nd_tile_gather(const float *img, int W, int H, int cs, int ucs, int ol, int tile_begin, int tile_count,
                              float *tiles, void *stream) {
    TileGeo g;
    ND_TRY(make_geo(W, H, cs, ucs, ol, &g));
    if (tile_count == 0) return ND_OK;
    if (!img || !tiles || tile_begin < 0 || tile_count < 0 || tile_begin + tile_count > g.cols * g.rows)
        ND_FAIL(ND_EINVAL, "nd_tile_gather: tiles [%d,%d) outside the grid of %d", tile_begin, tile_begin + tile_count, g.cols * g.rows);
    dim3 grid((cs + 255) / 256, cs, tile_count);
    hipLaunchKernelGGL(k_tile_gather, grid, dim3(256), 0, (hipStream_t)stream, img, g, tile_begin, tiles);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// fused: gather(+symmetric mirror) -> ReflectionPad2d(2) -> quad-planar first-layer input (plane 0 = r,g,b,0)
template <int DT>
__global__ void k_gather_pack(const float *__restrict__ img, TileGeo g, int tile_begin, typename Elem<DT>::vec *__restrict__ dst, int Sb) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;  // column in the reflect-padded tile
    const int v = blockIdx.y;
    const int t = blockIdx.z;
    if (u >= Sb) return;
    const int i = tile_begin + t;
    const int yi = i / g.cols, xi = i - yi * g.cols;
    const int qx = reflect_nr(u - 2, g.cs), qy = reflect_nr(v - 2, g.cs);
    const int sx = mirror_sym(xi * g.stride - g.pad + qx, g.W);
    const int sy = mirror_sym(yi * g.stride - g.pad + qy, g.H);
    const size_t plane = (size_t)g.W * g.H;
    const float *s = img + (size_t)sy * g.W + sx;
    typename Elem<DT>::vec o = {};
    o[0] = (typename Elem<DT>::scalar)s[0];
    o[1] = (typename Elem<DT>::scalar)s[plane];
    o[2] = (typename Elem<DT>::scalar)s[2 * plane];
    dst[((size_t)t * Sb + v) * Sb + u] = o;
}

int nd_launch_gather_pack(const float *img, int W, int H, int cs, int ucs, int ol, int tile_begin, int tile_count,
                          const QpBuf &dst, hipStream_t s) {
    TileGeo g;
    ND_TRY(make_geo(W, H, cs, ucs, ol, &g));
    if (tile_begin < 0 || tile_count <= 0 || tile_begin + tile_count > g.cols * g.rows || tile_count > dst.B)
        ND_FAIL(ND_EINVAL, "gather_pack: bad tile range [%d,+%d)", tile_begin, tile_count);
    if (dst.Hb != cs + 4 || dst.Wb != cs + 4 || dst.pad != 0) ND_FAIL(ND_EINVAL, "gather_pack: destination is not (cs+4)^2");
    dim3 grid((cs + 4 + 255) / 256, cs + 4, tile_count);
    ND_DISPATCH_DT(dst.dt, hipLaunchKernelGGL(k_gather_pack<DT>, grid, dim3(256), 0, s, img, g, tile_begin,
                                              (typename Elem<DT>::vec *)dst.base, cs + 4));
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// ------------------------------------------------------------------ stitch (denoise_image.py:204-213, 249-267)
// One thread per canvas pixel; contributions of the tiles [tile_begin, tile_begin+count) that cover it are added in
// ascending tile index order on top of the current canvas value: the fp32 sum order of the reference's loop.
struct Cover {
    int t;      // tile index relative to tile_begin
    int iy, ix; // position inside the tile
    float f;    // 1, .5 or .25 (seamless edges)
};

template <typename F>
__device__ __forceinline__ void for_each_cover(const TileGeo &g, int X, int Y, int tile_begin, int tile_count, F &&fn) {
    const int uwmax = g.cs - 2 * g.pad;
    int yi_lo = ceil_div_py(Y - uwmax + 1, g.stride);
    if (yi_lo < 0) yi_lo = 0;
    int yi_hi = Y / g.stride;
    if (yi_hi > g.rows - 1) yi_hi = g.rows - 1;
    int xi_lo = ceil_div_py(X - uwmax + 1, g.stride);
    if (xi_lo < 0) xi_lo = 0;
    int xi_hi = X / g.stride;
    if (xi_hi > g.cols - 1) xi_hi = g.cols - 1;
    for (int yi = yi_lo; yi <= yi_hi; ++yi) {
        const int ay = yi * g.stride;
        const int y1pad = max(0, ay - g.pad + g.cs - g.H);
        const int uh = g.cs - max(g.pad, y1pad) - g.pad;
        const int dy = Y - ay;
        if (dy >= uh) continue;
        float fy = 1.f;
        if (ay != 0 && dy < g.ol) fy *= 0.5f;
        if (ay + g.ucs < g.H && g.ol && dy >= uh - g.ol) fy *= 0.5f;
        for (int xi = xi_lo; xi <= xi_hi; ++xi) {
            const int i = yi * g.cols + xi;
            if (i < tile_begin || i >= tile_begin + tile_count) continue;
            const int ax = xi * g.stride;
            const int x1pad = max(0, ax - g.pad + g.cs - g.W);
            const int uw = g.cs - max(g.pad, x1pad) - g.pad;
            const int dx = X - ax;
            if (dx >= uw) continue;
            float f = fy;
            if (ax != 0 && dx < g.ol) f *= 0.5f;
            if (ax + g.ucs < g.W && g.ol && dx >= uw - g.ol) f *= 0.5f;
            fn(i - tile_begin, g.pad + dy, g.pad + dx, f);
        }
    }
}

__global__ void k_stitch_add(float *__restrict__ canvas, TileGeo g, const float *__restrict__ tiles, int tile_begin,
                             int tile_count, int y_first) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = y_first + blockIdx.y;
    if (X >= g.W || Y >= g.H) return;
    const size_t plane = (size_t)g.W * g.H;
    const size_t tplane = (size_t)g.cs * g.cs;
    float *c = canvas + (size_t)Y * g.W + X;
    float v0 = c[0], v1 = c[plane], v2 = c[2 * plane];
    bool any = false;
    for_each_cover(g, X, Y, tile_begin, tile_count, [&](int t, int iy, int ix, float f) {
        const float *s = tiles + (size_t)t * 3 * tplane + (size_t)iy * g.cs + ix;
        v0 += s[0] * f;
        v1 += s[tplane] * f;
        v2 += s[2 * tplane] * f;
        any = true;
    });
    if (any) {
        c[0] = v0;
        c[plane] = v1;
        c[2 * plane] = v2;
    }
}

static void stitch_band(const TileGeo &g, int tile_begin, int tile_count, int *y_first, int *y_rows) {
    const int yi0 = tile_begin / g.cols, yi1 = (tile_begin + tile_count - 1) / g.cols;
    const int uwmax = g.cs - 2 * g.pad;
    int y0 = yi0 * g.stride, y1 = yi1 * g.stride + uwmax;
    if (y1 > g.H) y1 = g.H;
    *y_first = y0;
    *y_rows = y1 - y0;
}

extern "C" int nd_stitch_add(float *canvas, int W, int H, int cs, int ucs, int ol, const float *tiles, int tile_begin,
                             int tile_count, void *stream) {
    TileGeo g;
    ND_TRY(make_geo(W, H, cs, ucs, ol, &g));
    if (tile_count == 0) return ND_OK;
    if (!canvas || !tiles || tile_begin < 0 || tile_count < 0 || tile_begin + tile_count > g.cols * g.rows)
        ND_FAIL(ND_EINVAL, "nd_stitch_add: tiles [%d,%d) outside the grid of %d", tile_begin, tile_begin + tile_count, g.cols * g.rows);
    int yf, yr;
    stitch_band(g, tile_begin, tile_count, &yf, &yr);
    if (yr <= 0) return ND_OK;
    dim3 grid((W + 255) / 256, yr);
    hipLaunchKernelGGL(k_stitch_add, grid, dim3(256), 0, (hipStream_t)stream, canvas, g, tiles, tile_begin, tile_count, yf);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// ------------------------------------------------------------------ layout conversion NCHW <-> quad-planar
template <int DT>
__global__ void k_nchw_to_qp(const float *__restrict__ x, int C, int H, int W, typename Elem<DT>::vec *__restrict__ dst,
                             long np, int Hb, int Wb, int pad, int plane0) {
    constexpr int N = Elem<DT>::N;
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    const int q = blockIdx.z % ((C + N - 1) / N), b = blockIdx.z / ((C + N - 1) / N);
    if (xx >= W) return;
    typename Elem<DT>::vec o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int c = N * q + e;
        o[e] = (typename Elem<DT>::scalar)(c < C ? x[(((size_t)b * C + c) * H + yy) * W + xx] : 0.f);
    }
    dst[(size_t)(plane0 + q) * np + ((size_t)b * Hb + yy + pad) * Wb + xx + pad] = o;
}

int nd_launch_nchw_to_qp(const float *x, int C, const QpBuf &dst, int plane0, hipStream_t s) {
    const int H = dst.Hb - 2 * dst.pad, W = dst.Wb - 2 * dst.pad;
    const int n = nd_cpp(dst.dt);
    dim3 grid((W + 255) / 256, H, dst.B * ((C + n - 1) / n));
    ND_DISPATCH_DT(dst.dt, hipLaunchKernelGGL(k_nchw_to_qp<DT>, grid, dim3(256), 0, s, x, C, H, W,
                                              (typename Elem<DT>::vec *)dst.base, dst.np(), dst.Hb, dst.Wb, dst.pad, plane0));
    ND_HIP(hipGetLastError());
    return ND_OK;
}

template <int DT>
__global__ void k_qp_to_nchw(const typename Elem<DT>::vec *__restrict__ src, long np, int Hb, int Wb, int pad, int plane0,
                             float *__restrict__ y, int C, int H, int W) {
    constexpr int N = Elem<DT>::N;
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    const int q = blockIdx.z % ((C + N - 1) / N), b = blockIdx.z / ((C + N - 1) / N);
    if (xx >= W) return;
    const typename Elem<DT>::vec v = src[(size_t)(plane0 + q) * np + ((size_t)b * Hb + yy + pad) * Wb + xx + pad];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int c = N * q + e;
        if (c < C) y[(((size_t)b * C + c) * H + yy) * W + xx] = (float)v[e];
    }
}

int nd_launch_qp_to_nchw(const QpBuf &src, int plane0, float *y, int C, hipStream_t s) {
    const int H = src.Hb - 2 * src.pad, W = src.Wb - 2 * src.pad;
    const int n = nd_cpp(src.dt);
    dim3 grid((W + 255) / 256, H, src.B * ((C + n - 1) / n));
    ND_DISPATCH_DT(src.dt, hipLaunchKernelGGL(k_qp_to_nchw<DT>, grid, dim3(256), 0, s, (const typename Elem<DT>::vec *)src.base,
                                              src.np(), src.Hb, src.Wb, src.pad, plane0, y, C, H, W));
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// x [B,3,H,W] -> ReflectionPad2d(2) (UtNet.py:27,98) -> plane 0 of the first-layer input [(H+4) x (W+4)]
template <int DT>
__global__ void k_reflect_pack(const float *__restrict__ x, int H, int W, typename Elem<DT>::vec *__restrict__ dst) {
    const int Hb = H + 4, Wb = W + 4;
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y, b = blockIdx.z;
    if (u >= Wb) return;
    const int qx = reflect_nr(u - 2, W), qy = reflect_nr(v - 2, H);
    const size_t plane = (size_t)H * W;
    const float *s = x + ((size_t)b * 3 * H + qy) * W + qx;
    typename Elem<DT>::vec o = {};
    o[0] = (typename Elem<DT>::scalar)s[0];
    o[1] = (typename Elem<DT>::scalar)s[plane];
    o[2] = (typename Elem<DT>::scalar)s[2 * plane];
    dst[((size_t)b * Hb + v) * Wb + u] = o;
}

int nd_launch_reflect_pack(const float *x, int B, int H, int W, const QpBuf &dst, hipStream_t s) {
    if (dst.Hb != H + 4 || dst.Wb != W + 4 || dst.pad != 0 || B > dst.B) ND_FAIL(ND_EINVAL, "reflect_pack: bad destination");
    dim3 grid((W + 4 + 255) / 256, H + 4, B);
    ND_DISPATCH_DT(dst.dt, hipLaunchKernelGGL(k_reflect_pack<DT>, grid, dim3(256), 0, s, x, H, W, (typename Elem<DT>::vec *)dst.base));
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// ------------------------------------------------------------------ MaxPool2d(2) (UtNet.py:34), quad-planar
template <int DT>
__global__ void k_maxpool2(const typename Elem<DT>::vec *__restrict__ src, long snp, int sHb, int sWb, int spad, int splane0,
                           typename Elem<DT>::vec *__restrict__ dst, long dnp, int dHb, int dWb, int dpad, int Ho, int Wo, int B) {
    typedef typename Elem<DT>::vec V;
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    const int b = blockIdx.z % B, q = blockIdx.z / B;
    if (xx >= Wo) return;
    const V *s = src + (size_t)(splane0 + q) * snp + ((size_t)b * sHb + 2 * yy + spad) * sWb + 2 * xx + spad;
    const V a = s[0], c = s[1], d = s[sWb], e = s[sWb + 1];
    V o;
#pragma unroll
    for (int k = 0; k < Elem<DT>::N; ++k)
        o[k] = (typename Elem<DT>::scalar)fmaxf(fmaxf((float)a[k], (float)c[k]), fmaxf((float)d[k], (float)e[k]));
    dst[(size_t)q * dnp + ((size_t)b * dHb + yy + dpad) * dWb + xx + dpad] = o;
}

int nd_launch_maxpool2(const QpBuf &src, int src_plane0, int planes, const QpBuf &dst, hipStream_t s) {
    const int Hi = src.Hb - 2 * src.pad, Wi = src.Wb - 2 * src.pad;
    const int Ho = Hi / 2, Wo = Wi / 2;
    if (dst.Hb - 2 * dst.pad != Ho || dst.Wb - 2 * dst.pad != Wo || dst.B != src.B || dst.planes < planes || dst.dt != src.dt)
        ND_FAIL(ND_EINVAL, "maxpool2: destination does not fit %dx%d", Ho, Wo);
    dim3 grid((Wo + 127) / 128, Ho, src.B * planes);
    ND_DISPATCH_DT(src.dt, hipLaunchKernelGGL(k_maxpool2<DT>, grid, dim3(128), 0, s, (const typename Elem<DT>::vec *)src.base,
                                              src.np(), src.Hb, src.Wb, src.pad, src_plane0, (typename Elem<DT>::vec *)dst.base,
                                              dst.np(), dst.Hb, dst.Wb, dst.pad, Ho, Wo, src.B));
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// ------------------------------------------------------------------ final Conv2d(funit,3,1) + ZeroPad2d(-2) (UtNet.py:86,88)
// w: [3][cin] (torch layout), bias [3].  One thread per output pixel; each plane read is a coalesced float4 stream.
template <int DT>
__device__ __forceinline__ void dot3(const typename Elem<DT>::vec *__restrict__ s, long np, int planes,
                                     const float *__restrict__ w, int cin, float &o0, float &o1, float &o2) {
    constexpr int N = Elem<DT>::N;
    for (int q = 0; q < planes; ++q) {
        const typename Elem<DT>::vec v = s[(size_t)q * np];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const int c = N * q + e;
            if (c < cin) {
                const float f = (float)v[e];
                o0 = fmaf(f, w[c], o0);
                o1 = fmaf(f, w[cin + c], o1);
                o2 = fmaf(f, w[2 * cin + c], o2);
            }
        }
    }
}

template <int DT>
__global__ void k_final1x1(const typename Elem<DT>::vec *__restrict__ src, long np, int Hb, int Wb, int planes, int cin,
                           const float *__restrict__ w, const float *__restrict__ bias, int crop, float *__restrict__ y,
                           int H, int W, int sigmoid) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y, b = blockIdx.z;
    if (xx >= W) return;
    float o0 = bias[0], o1 = bias[1], o2 = bias[2];
    dot3<DT>(src + ((size_t)b * Hb + yy + crop) * Wb + xx + crop, np, planes, w, cin, o0, o1, o2);
    if (sigmoid) {   // UNet head (ThirdPartyNets.py:169)
        o0 = 1.f / (1.f + expf(-o0));
        o1 = 1.f / (1.f + expf(-o1));
        o2 = 1.f / (1.f + expf(-o2));
    }
    float *d = y + ((size_t)b * 3 * H + yy) * W + xx;
    d[0] = o0;
    d[(size_t)H * W] = o1;
    d[2 * (size_t)H * W] = o2;
}

int nd_launch_final1x1(const QpBuf &src, int cin, const float *w, const float *bias, int crop, float *y, int H, int W,
                       hipStream_t s, int sigmoid) {
    if (src.pad != 0 || src.Hb != H + 2 * crop || src.Wb != W + 2 * crop) ND_FAIL(ND_EINVAL, "final1x1: bad source geometry");
    dim3 grid((W + 255) / 256, H, src.B);
    const int n = nd_cpp(src.dt);
    ND_DISPATCH_DT(src.dt, hipLaunchKernelGGL(k_final1x1<DT>, grid, dim3(256), 0, s, (const typename Elem<DT>::vec *)src.base,
                                              src.np(), src.Hb, src.Wb, (cin + n - 1) / n, cin, w, bias, crop, y, H, W, sigmoid));
    ND_HIP(hipGetLastError());
    return ND_OK;
}

// fused: final 1x1 + crop + useful crop + seamless edges + canvas += (no NCHW tile batch in HBM)
template <int DT>
__global__ void k_final1x1_stitch(const typename Elem<DT>::vec *__restrict__ src, long np, int Hb, int Wb, int planes, int cin,
                                  const float *__restrict__ w, const float *__restrict__ bias, int crop,
                                  float *__restrict__ canvas, TileGeo g, int tile_begin, int tile_count, int y_first) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = y_first + blockIdx.y;
    if (X >= g.W || Y >= g.H) return;
    const size_t plane = (size_t)g.W * g.H;
    float *c = canvas + (size_t)Y * g.W + X;
    float v0 = c[0], v1 = c[plane], v2 = c[2 * plane];
    bool any = false;
    for_each_cover(g, X, Y, tile_begin, tile_count, [&](int t, int iy, int ix, float f) {
        float o0 = bias[0], o1 = bias[1], o2 = bias[2];
        dot3<DT>(src + ((size_t)t * Hb + iy + crop) * Wb + ix + crop, np, planes, w, cin, o0, o1, o2);
        v0 += o0 * f;
        v1 += o1 * f;
        v2 += o2 * f;
        any = true;
    });
    if (any) {
        c[0] = v0;
        c[plane] = v1;
        c[2 * plane] = v2;
    }
}

int nd_launch_final1x1_stitch(const QpBuf &src, int cin, const float *w, const float *bias, int crop, float *canvas,
                              int W, int H, int cs, int ucs, int ol, int tile_begin, int tile_count, hipStream_t s) {
    TileGeo g;
    ND_TRY(make_geo(W, H, cs, ucs, ol, &g));
    if (src.pad != 0 || src.Hb != cs + 2 * crop || src.Wb != cs + 2 * crop || tile_count > src.B)
        ND_FAIL(ND_EINVAL, "final1x1_stitch: bad source geometry");
    int yf, yr;
    stitch_band(g, tile_begin, tile_count, &yf, &yr);
    if (yr <= 0) return ND_OK;
    dim3 grid((W + 255) / 256, yr);
    const int n = nd_cpp(src.dt);
    ND_DISPATCH_DT(src.dt, hipLaunchKernelGGL(k_final1x1_stitch<DT>, grid, dim3(256), 0, s, (const typename Elem<DT>::vec *)src.base,
                                              src.np(), src.Hb, src.Wb, (cin + n - 1) / n, cin, w, bias, crop, canvas, g,
                                              tile_begin, tile_count, yf));
    ND_HIP(hipGetLastError());
    return ND_OK;
}
