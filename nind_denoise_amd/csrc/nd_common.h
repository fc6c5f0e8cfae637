// Internal declarations shared by the translation units of libnind_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
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
// A plane element is ALWAYS 16 bytes per pixel: 4 fp32 channels (ND_F32) or 8 bf16 / fp16 channels (ND_BF16 / ND_F16),
// so the pixel indexing, the LDS-DMA images and the ds_read_b128 fragment reads are identical for every storage type.
static inline int nd_cpp(int dt) { return dt == ND_F32 ? 4 : 8; }   // channels per plane
struct QpBuf {
    float *base;    // first plane
    int planes;     // C / nd_cpp(dt) (channels padded to a whole plane)
    int B, Hb, Wb;  // images in use, bordered rows / cols
    int pad;        // zero border width
    long pstride;   // 16-byte elements per plane (capacity: batch * Hb * Wb); B may be smaller for a partial batch
    int dt = ND_F32;  // storage type (nd_dtype)
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
    int in_plane0 = 0;  // first input plane inside `in` (a channel slice of a concat / gradient buffer)
    float *pre = nullptr;   // training: also store the pre-activation (acc + bias), compact [C/4][B][Hv][Wv] float4 planes
    long pre_plane = 0;     // 16-byte elements per plane of `pre`
    float *part = nullptr;  // scratch for the split-K tail (raw accumulators of K slices); null: never split
    size_t part_bytes = 0;
    bool nosplit = false;   // ND_FLAG_NO_SPLITK of the call: keep every tile whole (bits independent of the launch composition)
    int nbatch = 1;         // independent problems of this shape in one launch (Winograd positions)
    long in_bs = 0, out_bs = 0;   // 16-byte elements between consecutive problems' input / output buffers
    size_t w_bs = 0;        // floats between consecutive problems' packed weights (the bias is shared)
    // fused MaxPool2d(2) (fp32 inference: conv_w2d.hip / winograd.hip; 16-bit storage: conv_qp): the layer also writes max over 2x2 blocks of its activated
    // output into planes [0, cout/4) of `pool` (UtNet.py:99-105: every pooled tensor is a conv output that is also a skip)
    const QpBuf *pool = nullptr;
    // region of interest (rows == 0: the whole layer).  3x3 layers (conv_w2d, three-pass F(6x6)): rectangle [r0, r0 + rows) x
    // [c0, c0 + cols) of the valid OUTPUT grid -- only these outputs are computed and stored, from input rows [r0, r0 + rows + 2);
    // 2x2 stride-2 transpose: rectangle of the INPUT grid (each input pixel makes its 2x2 outputs).  Used by the fused denoise
    // loop: the last decoder levels only compute what the useful crop of a tile can reach (utnet_net.h: plan_rois)
    int roi_r0 = 0, roi_c0 = 0, roi_rows = 0, roi_cols = 0;
};
// scratch that lets every layer split its partial round: 512 work items of 64 x 1024 accumulators
static const size_t kSplitScratchBytes = (size_t)512 * 64 * 1024 * 4;
int nd_launch_conv(const ConvDesc &d, hipStream_t stream);
bool nd_conv_pool_fits(const ConvDesc &d);  // a 16-bit 3x3 launch with d.pool set can pool in its epilogue (else: nd_launch_maxpool2 after it)
bool nd_conv_roi_fits(const ConvDesc &d);   // a launch restricted to d.roi_* finds a workgroup shape that fits the LDS
static inline int nd_launch_conv_f32(const ConvDesc &d, hipStream_t stream) { return nd_launch_conv(d, stream); }
int nd_conv_variant_count();
int nd_conv_variant_gemm(int cin, int cout);   // 1-tap fp32 variant (256- / 128-row workgroup tiles) for a Winograd GEMM
// Winograd F(T x T, 3 x 3), T = 2 | 4 (winograd.hip): fp32 inference path of the wide 3x3 layers
size_t nd_wino_packed_floats(int T, int cin, int cout);
int nd_wino_pack(int T, int kind, int cin, int cout, const float *w, const float *bias, float *packed);
size_t nd_wino_scratch_bytes(int T, const QpBuf &in, int cin, int cout);
// ev2 (optional, profiling): two events, recorded after the input transform pass and after the GEMM launch
int nd_launch_conv_wino(int T, const ConvDesc &d, void *scratch, size_t scratch_bytes, hipStream_t s, hipEvent_t *ev2 = nullptr);
// algorithmic HBM bytes of the two transform passes of a three-pass layer (X read + V written; M read + Y written)
void nd_wino_xform_bytes(int T, const QpBuf &in, int cin, int cout, double *bytes_in, double *bytes_out);
// 1-D Winograd F(2,3) along x inside the implicit-GEMM kernel (conv_w1d.hip): fp32 inference form of the narrow 3x3 layers
// (T = 2: F(2,3), 2/3 of the MFMAs;  T = 4: F(4,3), 1/2)
size_t nd_w1d_packed_floats(int T, int cin, int cout);
int nd_w1d_pack(int T, int kind, int cin, int cout, const float *w, const float *bias, float *packed);
bool nd_w1d_fits(int T, const QpBuf &in);
int nd_launch_conv_w1d(int T, const ConvDesc &d, hipStream_t stream);
// the same F(4,3) layer with the input transform shared by the workgroup through LDS (conv_w2d.hip; same packed weights as T = 4)
bool nd_w2d_ok(const QpBuf &in);
long nd_w2d_tiles(const QpBuf &in, int cout);   // workgroup tiles of the whole layer
int nd_launch_conv_w2d(const ConvDesc &d, hipStream_t stream);
// slack (16-byte elements) behind the last plane of an activation buffer: an N tile of the conv kernels may read a 3x3 halo past
// the last pixel, a strip of conv_w2d up to 9 rows + 5 pixels
static inline size_t nd_buf_slack(int Wb) { return (size_t)10 * Wb + 8 + 2048; }
// Per-device launch caches (CU count, "dynamic LDS size already raised for this kernel") are std::atomic: entry points may be
// called from several host threads (include/nind_hip.h), and two threads that both miss write the same value.
// CU count of device `dev` (0 <= dev < 16), queried once per device.
static inline int nd_num_cus(int dev, int *out) {
    static std::atomic<int> cus[16];
    int n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        hipDeviceProp_t prop;
        ND_HIP(hipGetDeviceProperties(&prop, dev));
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    *out = n;
    return ND_OK;
}
// the arithmetic switches every flags-taking entry point accepts (include/nind_hip.h: nd_flags); unknown bits are an error
static inline int nd_check_flags(int flags) {
    if (flags & ~(ND_FLAG_NO_SPLITK | ND_FLAG_DIRECT_CONV | ND_FLAG_W1D_REGS | ND_FLAG_FULL_TILES | ND_FLAG_UNFUSED_POOL)) ND_FAIL(ND_EINVAL, "unknown flag bits 0x%x", flags);
    return ND_OK;
}
const char *nd_conv_variant_label(int v);

// packed size helpers (host)
// 32-row MFMA tiles, padded so that every workgroup shape (M_blk <= 128, 256 for the 2x2 stride-2 layers) reads packed rows only
static inline int nd_mtiles(int kind, int cout) {
    return kind == ND_CONVT2S2 ? (4 * cout + 255) / 256 * 8 : (cout + 127) / 128 * 4;   // up layers use 256-row workgroup tiles
}
// Row order of the GEMM of a 2x2 stride-2 transpose (M = 4 * Cout rows; weights and bias are packed in it, the conv epilogue
// and k_split_finish decode it).  An MFMA accumulator hands lane (pixel j, half h) the rows 8g + 4h + e (e = 0..3) of a 32-row
// tile.  The order puts the SAME channels of the two horizontally adjacent output pixels (2x, 2x + 1) into the two lane
// halves, so that a wave's store instruction writes 64 consecutive 16-byte plane elements (1 KiB contiguous) instead of 16-byte
// pieces at a 32-byte stride:
//   fp32   (4 channels per plane element):  m = 8 * (a * Cout/4 + quad) + 4 * b + e          co = 4 * quad + e
//   16-bit (8 channels per plane element):  m = 16 * (a * Cout/8 + oct) + 8 * b + e8         co = 8 * oct + e8
//          (there the epilogue first exchanges the halves of two 8-row groups, v_permlane32_swap, so that a lane owns all 8
//           channels of one pixel)
// (a, b) = output sub-position (row, column) of ConvTranspose2d(2, stride 2): out[2y + a][2x + b].
struct NdUpRow { int a, b, co; };
__host__ __device__ static inline NdUpRow nd_up_row(int m, int cout, int dt) {
    NdUpRow r;
    const int cpp = dt == ND_F32 ? 4 : 8, q = cout / cpp;
    const int G = m / (2 * cpp), e = m % cpp;
    r.b = (m / cpp) & 1;
    r.a = G / q;
    r.co = cpp * (G % q) + e;
    return r;
}
static inline int nd_taps(int kind) { return (kind == ND_CONV3 || kind == ND_CONVT3) ? 9 : (kind == ND_CONV2S2 ? 4 : 1); }
// K block = two planes = the K extent of one ds_read_b128 per operand: 8 fp32 channels or 16 bf16/fp16 channels
static inline int nd_kblocks(int cin, int dt = ND_F32) { return (cin + 2 * nd_cpp(dt) - 1) / (2 * nd_cpp(dt)); }
// packed layer size in 4-byte units: 1 KiB fragment pieces [mtile][kb][tap] (any dtype) + fp32 bias[mtiles*32]
static inline size_t nd_packed_floats(int kind, int cin, int cout, int dt = ND_F32) {
    return (size_t)nd_mtiles(kind, cout) * nd_kblocks(cin, dt) * nd_taps(kind) * 256 + (size_t)nd_mtiles(kind, cout) * 32;
}
void nd_pack_layer(int kind, int cin, int cout, int dt, const float *w, const float *bias, float *packed);
static inline void nd_pack_layer_f32(int kind, int cin, int cout, const float *w, const float *bias, float *packed) {
    nd_pack_layer(kind, cin, cout, ND_F32, w, bias, packed);
}

// device-side packers (pack_dev.hip): the same layouts from weights in HBM (fp32)
int nd_pack_layer_device(int kind, int cin, int cout, const float *w, const float *bias, float *packed, hipStream_t s);
int nd_pack_w1d_device(int T, int kind, int cin, int cout, const float *w, const float *bias, float *packed, hipStream_t s);
int nd_pack_wino_device(int T, int kind, int cin, int cout, const float *w, const float *bias, float *packed, hipStream_t s);

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
