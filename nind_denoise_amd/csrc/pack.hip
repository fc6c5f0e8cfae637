// Host-side weight repacking: torch state-dict tensors -> MFMA "A"-fragment order (done once at model load;
// reference load path: nn_common.py:127-132 `load_state_dict(torch.load(path))`).
//
// Packed layer = [mtile][kb][tap][lane 0..63][4 floats]  followed by  bias[mtiles*32]
//   lane = 32*h + i supplies rows m = 32*mtile + i and channels ci = 8*kb + 4*h + s (s = 0..3)
//   value = Weff[m][ci][tap]:
//     CONV3    Weff[co][ci][(ky,kx)] = w[co][ci][ky][kx]                         (w: [Cout,Cin,3,3])
//     CONVT3   Weff[co][ci][(a,b)]   = w[ci][co][2-a][2-b]                       (w: [Cin,Cout,3,3]; spatial flip +
//                                                                                  channel transpose turn the transpose
//                                                                                  conv into a correlation on the
//                                                                                  zero-bordered input)
//     CONVT2S2 Weff[m][ci]           = w[ci][co][a][b],  (a, b, co) = nd_up_row(m) (w: [Cin,Cout,2,2]; row order chosen for
//                                                                                  contiguous pixel-shuffle stores, nd_common.h)
//     CONV1    Weff[co][ci]          = w[co][ci]
//     CONV2S2  Weff[co][ci][(a,b)]   = w[co][ci][a][b]                           (w: [Cout,Cin,2,2], stride 2)
// bf16 / fp16: the same order with 8 channels per lane (ci = 16*kb + 8*h + s), values rounded to nearest even.
#include <string.h>

#include "nd_common.h"

static thread_local char g_err[512] = "";

void nd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *nd_last_error(void) { return g_err; }
extern "C" int nd_version(void) { return 103; }   // 103: round-3 ABI (train forward / backward halves, gradient bucket events, ND_FLAG_UNFUSED_POOL, GEMM row order of the 2x2 stride-2 transposes)

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline uint16_t f32_to_f16_rne(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

// 16-bit types: one 1 KiB piece holds, for lane = 32*h + i, the 8 channels ci = 16*kb + 8*h + s (s = 0..7) of row m:
// exactly the A operand of v_mfma_f32_32x32x16_{bf16,f16} (lane l supplies A[l&31][k = 8*(l>>5) + s]).
void nd_pack_layer(int kind, int cin, int cout, int dt, const float *w, const float *bias, float *packed) {
    const int taps = nd_taps(kind);
    const int KB = nd_kblocks(cin, dt);
    const int MT = nd_mtiles(kind, cout);
    const int M = kind == ND_CONVT2S2 ? 4 * cout : cout;
    const int cpp = nd_cpp(dt);   // channels per lane per piece
    float *bp = packed + (size_t)MT * KB * taps * 256;
    auto weff = [&](int m, int ci, int t) -> float {
        if (m >= M || ci >= cin) return 0.f;
        switch (kind) {
            case ND_CONV3: return w[((size_t)m * cin + ci) * 9 + t];
            case ND_CONVT3: return w[((size_t)ci * cout + m) * 9 + (8 - t)];
            case ND_CONVT2S2: {
                const NdUpRow r = nd_up_row(m, cout, dt);
                return w[((size_t)ci * cout + r.co) * 4 + 2 * r.a + r.b];
            }
            case ND_CONV2S2: return w[((size_t)m * cin + ci) * 4 + t];
            default: return w[(size_t)m * cin + ci];
        }
    };
    for (int mt = 0; mt < MT; ++mt)
        for (int kb = 0; kb < KB; ++kb)
            for (int t = 0; t < taps; ++t) {
                float *dst = packed + (((size_t)mt * KB + kb) * taps + t) * 256;
                uint16_t *dst16 = (uint16_t *)dst;
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 31, h = lane >> 5;
                    const int m = 32 * mt + i;
                    for (int s = 0; s < cpp; ++s) {
                        const float v = weff(m, 2 * cpp * kb + cpp * h + s, t);
                        if (dt == ND_F32)
                            dst[lane * 4 + s] = v;
                        else
                            dst16[lane * 8 + s] = dt == ND_BF16 ? f32_to_bf16_rne(v) : f32_to_f16_rne(v);
                    }
                }
            }
    for (int m = 0; m < MT * 32; ++m) {
        float v = 0.f;
        if (m < M && bias) v = bias[kind == ND_CONVT2S2 ? nd_up_row(m, cout, dt).co : m];
        bp[m] = v;
    }
}

extern "C" size_t nd_layer_packed_bytes(int kind, int cin, int cout, int dtype) {
    if (dtype < ND_F32 || dtype > ND_F16 || kind < 0 || kind > 4 || cin <= 0 || cout <= 0) return 0;
    return nd_packed_floats(kind, cin, cout, dtype) * sizeof(float);
}

extern "C" int nd_layer_pack(int kind, int cin, int cout, int dtype, const float *weight, const float *bias,
                             void *packed_host, size_t packed_bytes) {
    if (dtype < ND_F32 || dtype > ND_F16) ND_FAIL(ND_EINVAL, "nd_layer_pack: unsupported dtype %d", dtype);
    if (kind < 0 || kind > 4 || cin <= 0 || cout <= 0 || !weight || !packed_host)
        ND_FAIL(ND_EINVAL, "nd_layer_pack: bad arguments");
    if (packed_bytes < nd_layer_packed_bytes(kind, cin, cout, dtype))
        ND_FAIL(ND_ENOMEM, "nd_layer_pack: packed buffer too small");
    nd_pack_layer(kind, cin, cout, dtype, weight, bias, (float *)packed_host);
    return ND_OK;
}
