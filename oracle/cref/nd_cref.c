/* Plain-C restatement of the integer / byte-exact part of the hot path and of the layer arithmetic
 * (ORACLE: test infrastructure only -- never linked into or called by the product path).
 *
 *   cref_tile_geom / cref_gather_tile   OneImageDS      /root/reference/src/nind_denoise/denoise_image.py:100-174
 *   cref_stitch_add                     main-loop body  denoise_image.py:204-213, 249-267
 *   cref_conv2d / cref_conv_transpose2d / cref_maxpool2 / cref_prelu
 *                                       the torch.nn layers UtNet is built from (networks/UtNet.py:27-88), written as
 *                                       naive direct loops with double accumulation -- an independent check of the
 *                                       torch-functional oracle (oracle/networks.py) on small shapes.
 * Built by oracle/cref/Makefile into libnd_cref.so; tests load it with ctypes.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

static int ceil_div(int a, int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }

int cref_tile_grid(int W, int H, int cs, int ucs, int ol, int *cols, int *rows, int *pad) {
    if (ucs - ol <= 0) return -1;
    *cols = ceil_div(W - ucs, ucs - ol) + 1;
    *rows = ceil_div(H - ucs, ucs - ol) + 1;
    *pad = (cs - ucs) / 2;
    return 0;
}

void cref_tile_geom(int i, int W, int H, int cs, int ucs, int ol, int *x0, int *y0, int ud[4], int us[2]) {
    int cols, rows, pad;
    cref_tile_grid(W, H, cs, ucs, ol, &cols, &rows, &pad);
    int yi = i / cols, xi = i - yi * cols;
    *x0 = ucs * xi - ol * xi - pad;
    *y0 = ucs * yi - ol * yi - pad;
    int x1pad = *x0 + cs - W > 0 ? *x0 + cs - W : 0;
    int y1pad = *y0 + cs - H > 0 ? *y0 + cs - H : 0;
    ud[0] = pad; ud[1] = pad;
    ud[2] = cs - (pad > x1pad ? pad : x1pad);
    ud[3] = cs - (pad > y1pad ? pad : y1pad);
    us[0] = *x0 + pad; us[1] = *y0 + pad;
}

/* strip by strip, exactly the slices of denoise_image.py:145-170 */
void cref_gather_tile(const float *img, int W, int H, int cs, int ucs, int ol, int i, float *ret) {
    int x0, y0, ud[4], us[2];
    cref_tile_geom(i, W, H, cs, ucs, ol, &x0, &y0, ud, us);
    int x1 = x0 + cs, y1 = y0 + cs;
    int x0pad = x0 < 0 ? -x0 : 0, x1pad = x1 > W ? x1 - W : 0;
    int y0pad = y0 < 0 ? -y0 : 0, y1pad = y1 > H ? y1 - H : 0;
    int ys = y0 + y0pad, ye = y1 - y1pad, xs = x0 + x0pad, xe = x1 - x1pad;
    for (int c = 0; c < 3; ++c) {
        const float *im = img + (size_t)c * W * H;
        float *rt = ret + (size_t)c * cs * cs;
#define IM(y, x) im[(size_t)(y) * W + (x)]
#define RT(y, x) rt[(size_t)(y) * cs + (x)]
        for (int y = ys; y < ye; ++y)
            for (int x = xs; x < xe; ++x) RT(y0pad + y - ys, x0pad + x - xs) = IM(y, x);
        if (x0pad > 0) {
            for (int y = ys; y < ye; ++y)
                for (int q = 0; q < x0pad; ++q) RT(y0pad + y - ys, q) = IM(y, xs + x0pad - 1 - q);
            for (int r = 0; r < y0pad; ++r)
                for (int q = 0; q < x0pad; ++q) RT(r, q) = IM(y0pad - 1 - r, x0pad - 1 - q);
            for (int r = 0; r < y1pad; ++r)
                for (int q = 0; q < x0pad; ++q) RT(cs - y1pad + r, q) = IM(H - 1 - r, x0pad - 1 - q);
        }
        if (x1pad > 0) {
            for (int y = ys; y < ye; ++y)
                for (int q = 0; q < x1pad; ++q) RT(y0pad + y - ys, cs - x1pad + q) = IM(y, xe - 1 - q);
            for (int r = 0; r < y0pad; ++r)
                for (int q = 0; q < x1pad; ++q) RT(r, cs - x1pad + q) = IM(y0pad - 1 - r, W - 1 - q);
            for (int r = 0; r < y1pad; ++r)
                for (int q = 0; q < x1pad; ++q) RT(cs - y1pad + r, cs - x1pad + q) = IM(H - 1 - r, W - 1 - q);
        }
        for (int r = 0; r < y0pad; ++r)
            for (int x = xs; x < xe; ++x) RT(r, x0pad + x - xs) = IM(ys + y0pad - 1 - r, x);
        for (int r = 0; r < y1pad; ++r)
            for (int x = xs; x < xe; ++x) RT(cs - y1pad + r, x0pad + x - xs) = IM(ye - 1 - r, x);
#undef IM
#undef RT
    }
}

/* crop the useful region, halve the overlap strips, add into the canvas (tile i) */
void cref_stitch_add(float *canvas, int W, int H, int cs, int ucs, int ol, int i, const float *tile) {
    int x0, y0, ud[4], us[2];
    cref_tile_geom(i, W, H, cs, ucs, ol, &x0, &y0, ud, us);
    int w = ud[2] - ud[0], h = ud[3] - ud[1], ax = us[0], ay = us[1];
    for (int c = 0; c < 3; ++c)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float v = tile[((size_t)c * cs + ud[1] + y) * cs + ud[0] + x];
                if (ax != 0 && x < ol) v = v / 2;
                if (ay != 0 && y < ol) v = v / 2;
                if (ax + ucs < W && ol && x >= w - ol) v = v / 2;
                if (ay + ucs < H && ol && y >= h - ol) v = v / 2;
                float *d = &canvas[((size_t)c * H + ay + y) * W + ax + x];
                *d = *d + v;
            }
}

/* x [B,Ci,H,W], w [Co,Ci,k,k] -> y [B,Co,H-k+1,W-k+1] (valid cross-correlation, as torch Conv2d) */
void cref_conv2d(const float *x, int B, int Ci, int H, int W, const float *w, const float *bias, int Co, int k, float *y) {
    int Ho = H - k + 1, Wo = W - k + 1;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? bias[co] : 0.0;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int ky = 0; ky < k; ++ky)
                            for (int kx = 0; kx < k; ++kx)
                                acc += (double)x[(((size_t)b * Ci + ci) * H + oy + ky) * W + ox + kx] *
                                       w[(((size_t)co * Ci + ci) * k + ky) * k + kx];
                    y[(((size_t)b * Co + co) * Ho + oy) * Wo + ox] = (float)acc;
                }
}

/* x [B,Ci,H,W], w [Ci,Co,k,k], stride s -> y [B,Co,(H-1)s+k,(W-1)s+k] (scatter form, as torch ConvTranspose2d) */
void cref_conv_transpose2d(const float *x, int B, int Ci, int H, int W, const float *w, const float *bias, int Co, int k,
                           int s, double *scratch, float *y) {
    int Ho = (H - 1) * s + k, Wo = (W - 1) * s + k;
    size_t n = (size_t)B * Co * Ho * Wo;
    for (size_t t = 0; t < n; ++t) scratch[t] = 0.0;
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Ci; ++ci)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    double v = x[(((size_t)b * Ci + ci) * H + iy) * W + ix];
                    for (int co = 0; co < Co; ++co)
                        for (int ky = 0; ky < k; ++ky)
                            for (int kx = 0; kx < k; ++kx)
                                scratch[(((size_t)b * Co + co) * Ho + iy * s + ky) * Wo + ix * s + kx] +=
                                    v * w[(((size_t)ci * Co + co) * k + ky) * k + kx];
                }
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (size_t t = 0; t < (size_t)Ho * Wo; ++t) {
                size_t o = ((size_t)b * Co + co) * Ho * Wo + t;
                y[o] = (float)(scratch[o] + (bias ? bias[co] : 0.0));
            }
}

void cref_maxpool2(const float *x, int BC, int H, int W, float *y) {
    int Ho = H / 2, Wo = W / 2;
    for (int c = 0; c < BC; ++c)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const float *p = x + ((size_t)c * H + 2 * oy) * W + 2 * ox;
                float m = p[0];
                if (p[1] > m) m = p[1];
                if (p[W] > m) m = p[W];
                if (p[W + 1] > m) m = p[W + 1];
                y[((size_t)c * Ho + oy) * Wo + ox] = m;
            }
}

void cref_prelu(float *x, size_t n, float slope) {
    for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}
