// Internal declarations shared by the translation units of libnind_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "nind_hip.h"

// ------------------------------------------------------------------ errors
void nd_set_error(const char *fmt, ...);
#define ND_FAIL(code, ...)        \
    do {                          \
        nd_set_error(__VA_ARGS__); \
        return (code);            \
    } while (0)
#define ND_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) ND_FAIL(ND_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define ND_TRY(call)        \
    do {                    \
        int r_ = (call);    \
        if (r_ != 0) return r_; \
    } while (0)

// ------------------------------------------------------------------ quad-planar activation buffers
// An activation tensor [B, C, H, W] lives in HBM as C/4 planes of float4 "channel quads":
//     plane q, image b, row y, col x  ->  float4 at  ((q * B + b) * Hb + y + pad) * Wb + x + pad
// where Hb = H + 2*pad, Wb = W + 2*pad and pad is the zero border the CONSUMER needs
// (2 when the consumer is a ConvTranspose2d(3): it then is a plain valid 3x3 correlation on the bordered buffer).
// Within a plane all images are contiguous, so a pixel has ONE linear index p = (b*Hb + y)*Wb + x and the 3x3
// neighbour (ky,kx) is p + ky*Wb + kx: an implicit-GEMM N tile is a contiguous pixel range and its LDS halo
// image is a contiguous copy.
struct QpBuf {
    float *base;    // first plane
    int planes;     // C/4 (channels padded to a multiple of 4)
    int B, Hb, Wb;  // images in use, bordered rows / cols
    int pad;        // zero border width
    long pstride;   // float4 per plane (capacity: batch * Hb * Wb); B may be smaller for a partial batch
    long np() const { return pstride; }
    long used() const { return (long)B * Hb * Wb; }
};

// ------------------------------------------------------------------ one conv launch
struct ConvDesc {
    int kind;           // nd_layer_kind
    int act;            // nd_act
    float slope;        // PReLU slope (used when slope_dev is null)
    const float *slope_dev;  // PReLU slope in HBM (the packed blob), or null
    const float *wpk;   // packed weights [mtile][kb][tap][64 lanes][4]
    const float *bias;  // [mtiles*32]
    int cin, cout;      // logical channels (cin is padded to 8 in the buffers)
    QpBuf in;           // bordered input (for CONVT3 its border supplies the implicit zero padding)
    QpBuf out;          // destination buffer (possibly a concat buffer)
    int out_plane0;     // first destination plane (channel offset / 4) inside `out`
    int variant;        // -1: pick automatically
};
int nd_launch_conv_f32(const ConvDesc &d, hipStream_t stream);
int nd_conv_variant_count();
const char *nd_conv_variant_label(int v);

// packed size helpers (host)
// 32-row MFMA tiles, padded so that every workgroup shape (M_blk <= 128, 256 for the 2x2 stride-2 layers) reads packed rows only
static inline int nd_mtiles(int kind, int cout) {
    return kind == ND_CONVT2S2 ? (4 * cout + 255) / 256 * 8 : (cout + 127) / 128 * 4;   // up layers use 256-row workgroup tiles
}
static inline int nd_taps(int kind) { return (kind == ND_CONV3 || kind == ND_CONVT3) ? 9 : 1; }
static inline int nd_kblocks(int cin) { return (cin + 7) / 8; }
static inline size_t nd_packed_floats(int kind, int cin, int cout) {
    return (size_t)nd_mtiles(kind, cout) * nd_kblocks(cin) * nd_taps(kind) * 256 + (size_t)nd_mtiles(kind, cout) * 32;
}
void nd_pack_layer_f32(int kind, int cin, int cout, const float *w, const float *bias, float *packed);

// ------------------------------------------------------------------ auxiliary kernels (aux_kernels.hip)
int nd_launch_nchw_to_qp(const float *x, int C, const QpBuf &dst, int plane0, hipStream_t s);
int nd_launch_qp_to_nchw(const QpBuf &src, int plane0, float *y, int C, hipStream_t s);
int nd_launch_reflect_pack(const float *x_nchw, int B, int H, int W, const QpBuf &dst, hipStream_t s);
int nd_launch_maxpool2(const QpBuf &src, int src_plane0, int planes, const QpBuf &dst, hipStream_t s);
int nd_launch_final1x1(const QpBuf &src, int cin, const float *w, const float *bias, int crop, float *y_nchw, int H,
                       int W, hipStream_t s, int sigmoid = 0);
int nd_launch_final1x1_stitch(const QpBuf &src, int cin, const float *w, const float *bias, int crop, float *canvas,
                              int width, int height, int cs, int ucs, int ol, int tile_begin, int tile_count,
                              hipStream_t s);
int nd_launch_gather_pack(const float *img, int width, int height, int cs, int ucs, int ol, int tile_begin,
                          int tile_count, const QpBuf &dst, hipStream_t s);
