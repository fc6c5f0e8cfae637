// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32), quad-planar layout.
//
// One kernel serves every weighted 3x3 / 1x1-shaped layer of UtNet (reference: networks/UtNet.py:27-88):
//   Conv2d(3, valid)            -> 9 taps on the input buffer
//   ConvTranspose2d(3, s=1)     -> the same 9-tap valid correlation on a buffer that carries a 2-pixel ZERO border,
//                                  with weights flipped / channel-transposed at pack time (pack.hip)
//   ConvTranspose2d(2, s=2)     -> 1 tap, M = 4*Cout rows ordered (a,b,co); pixel-shuffle in the store
//   Conv2d(1)                   -> 1 tap
//
// GEMM view (per launch):  D[m][p] = sum_{tap,ci} Wp[m][tap,ci] * X[ci][p + off(tap)]
//   M = output channels (MFMA "A" operand = packed weights), N = linear pixels p of the bordered input buffer
//   (MFMA "B" operand), K = taps * Cin.  Junk columns (x >= Wvalid, y >= Hvalid) are computed and masked at the store:
//   that is what makes every N tile a contiguous range and every LDS halo image one contiguous copy.
//
// MFMA fragment use (f32 32x32x2: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]):
//   one ds_read_b128 per operand fetches 4 consecutive channels c..c+3 of channel-quad (2*kb + h), h = lane>>5;
//   MFMA step s (0..3) therefore contracts channels {8kb+s, 8kb+4+s}: the K order is permuted identically for A and B.
//   The accumulator holds, for lane (j,h), rows (r&3) + 8*(r>>2) + 4h: registers 4g..4g+3 are four CONSECUTIVE output
//   channels 8g+4h.. of pixel j  => one float4 store per (g) lands in the quad-planar output, 512 B contiguous per
//   half wave.
//
// Pipeline: 2 LDS stages filled by LDS-DMA (global_load_lds_dwordx4); per K chunk ONE workgroup barrier:
//   wait vmcnt(0) -> s_barrier -> issue the DMA of chunk c+1 into the other stage -> MFMAs of chunk c.
#include <type_traits>

#include "nd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const f32x4 *in;     // plane 0 of the input buffer
    const float *wpk;    // packed weights
    const float *bias;   // [mtiles*32]
    f32x4 *out;          // plane 0 of the destination buffer
    long in_plane;       // float4 per input plane
    long out_plane;      // float4 per output plane
    int NP;              // linear pixels in the input buffer (B*Hb*Wb)
    int P, Wb;           // Hb*Wb, Wb of the input buffer
    int Hv, Wv;          // valid output rows / cols per image
    int KB;              // Cin / 8
    int M;               // GEMM rows (Cout, or 4*Cout for the 2x2 stride-2 transpose)
    int cout;            // output channels
    int Po, Wo, opad;    // destination buffer: Hb*Wb, Wb, border
    int out_plane0;      // first destination plane
    int act;
    float slope;
    const float *slope_dev;
};

__device__ __forceinline__ void glds16(const void *g, void *l) {
    // 64 lanes x 16 B: per-lane global source, LDS destination = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case ND_ACT_PRELU: return v > 0.f ? v : v * slope;
        case ND_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case ND_ACT_HARDSWISH: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f;
        default: return v;
    }
}

// MR x NR : 32x32 MFMA tiles per wave;  WM x WN : waves per workgroup;  TAPS in {9,1};  KBC : 8-channel blocks per chunk
template <int MR, int NR, int WM, int WN, int TAPS, int KBC, bool UP>
__global__ __launch_bounds__(64 * WM * WN) void conv_qp_f32(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN;
    constexpr int MTB = MR * WM;                    // 32-row tiles per workgroup
    constexpr int NBLK = 32 * NR * WN;              // pixels per workgroup
    constexpr int WBYTES = MTB * KBC * TAPS * 1024; // weight bytes per stage

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, h = lane >> 5;

    const int mb = blockIdx.y;
    const long n0 = (long)blockIdx.x * NBLK;

    const int halo = (TAPS == 9) ? 2 * p.Wb + 2 : 0;
    const int G = (NBLK + halo + 63) >> 6;   // 64-pixel DMA pieces per channel-quad plane
    const int planeB = G * 1024;
    const int stageB = WBYTES + 2 * KBC * planeB;

    int toff[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) toff[t] = (TAPS == 9) ? ((t / 3) * p.Wb + (t % 3)) * 16 : 0;

    const float *wsrc = p.wpk + (size_t)mb * MTB * p.KB * TAPS * 256 + lane * 4;
    const f32x4 *asrc = p.in + n0 + lane;

    auto fill = [&](int c, int s) {
        char *sb = smem + s * stageB;
        // weights: for each of the workgroup's m-tiles, KBC*TAPS consecutive 1 KiB pieces
#pragma unroll
        for (int mt = 0; mt < MTB; ++mt) {
            const float *src = wsrc + ((size_t)mt * p.KB + (size_t)c * KBC) * TAPS * 256;
            char *dst = sb + mt * KBC * TAPS * 1024;
            for (int q = wave; q < KBC * TAPS; q += NW) glds16(src + q * 256, dst + q * 1024);
        }
        // activations: 2*KBC channel-quad planes, G pieces each
#pragma unroll
        for (int pl = 0; pl < 2 * KBC; ++pl) {
            const f32x4 *src = asrc + (size_t)(c * 2 * KBC + pl) * p.in_plane;
            char *dst = sb + WBYTES + pl * planeB;
            for (int g = wave; g < G; g += NW) glds16(src + g * 64, dst + g * 1024);
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int a = 0; a < MR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nchunks = p.KB / KBC;
    fill(0, 0);
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + 1 < nchunks) fill(c + 1, (c + 1) & 1);
        const char *sb = smem + (c & 1) * stageB;
        const char *aB = sb + (wm * MR) * KBC * TAPS * 1024 + lane * 16;
        const char *bB = sb + WBYTES + h * planeB + (wn * NR * 32 + j) * 16;
        // fragments of step i+1 are fetched before the MFMAs of step i (one wave per SIMD has nobody else to hide
        // the LDS latency); every index below is a compile-time constant after unrolling
        constexpr int STEPS = KBC * TAPS;
        f32x4 a[2][MR], b[2][NR];
        auto load_frags = [&](int buf, int step) {
            const int kbl = step / TAPS, t = step % TAPS;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
                a[buf][mr] = *(const f32x4 *)(aB + ((mr * KBC + kbl) * TAPS + t) * 1024);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
                b[buf][nr] = *(const f32x4 *)(bB + kbl * 2 * planeB + toff[t] + nr * 512);
        };
        load_frags(0, 0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 1 < STEPS) load_frags((st + 1) & 1, st + 1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][mr][s], b[st & 1][nr][s], acc[mr][nr], 0, 0, 0);
        }
    }

    // ---- epilogue: bias + activation, float4 stores into the (bordered, possibly concatenated) destination
    const float slope = p.act == ND_ACT_NONE ? 1.f : (p.slope_dev ? *p.slope_dev : p.slope);
    auto epilogue = [&](auto generic_act) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const long pix = n0 + (wn * NR + nr) * 32 + j;
            const int b = (int)(pix / p.P);
            const int r = (int)(pix - (long)b * p.P);
            const int y = r / p.Wb;
            const int x = r - y * p.Wb;
            const bool valid = pix < p.NP && y < p.Hv && x < p.Wv;
            const long pbase = UP ? (long)b * p.Po + (long)(2 * y + p.opad) * p.Wo + (2 * x + p.opad)
                                  : (long)b * p.Po + (long)(y + p.opad) * p.Wo + (x + p.opad);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m4 = ((mb * MTB + wm * MR + mr) * 32) + 8 * g + 4 * h;
                    if (valid && m4 < p.M) {
                        const f32x4 bv = *(const f32x4 *)(p.bias + m4);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = acc[mr][nr][4 * g + e] + bv[e];
                            if constexpr (decltype(generic_act)::value)
                                v[e] = apply_act(t, p.act, slope);
                            else
                                v[e] = t > 0.f ? t : t * slope;  // PReLU; "no activation" is slope 1
                        }
                        long off;
                        if (UP) {
                            const int ab = m4 / p.cout;
                            const int co = m4 - ab * p.cout;
                            off = (long)(p.out_plane0 + (co >> 2)) * p.out_plane + pbase + (long)(ab >> 1) * p.Wo + (ab & 1);
                        } else {
                            off = (long)(p.out_plane0 + (m4 >> 2)) * p.out_plane + pbase;
                        }
                        p.out[off] = v;
                    }
                }
            }
        }
    };
    if (p.act <= ND_ACT_PRELU)
        epilogue(std::false_type{});
    else
        epilogue(std::true_type{});
}

// ------------------------------------------------------------------ variants and dispatch
struct Variant {
    const char *name;
    int mblk, nblk, threads, taps, kbc;
    bool up;
    void (*fn)(ConvParams);
};

#define ND_VARIANT(MR, NR, WM, WN, TAPS, KBC, UP)                                                            \
    {                                                                                                        \
        "f32_m" #MR "x" #WM "_n" #NR "x" #WN "_t" #TAPS "_k" #KBC "_up" #UP, 32 * MR * WM, 32 * NR * WN, 64 * WM * WN, TAPS, KBC, UP, \
            conv_qp_f32<MR, NR, WM, WN, TAPS, KBC, UP>                                                       \
    }

static const Variant g_variants[] = {
    ND_VARIANT(2, 2, 1, 4, 9, 1, false),  // 0: M64  x N256, 4 waves
    ND_VARIANT(2, 2, 2, 2, 9, 1, false),  // 1: M128 x N128, 4 waves
    ND_VARIANT(2, 2, 2, 4, 9, 1, false),  // 2: M128 x N256, 8 waves
    ND_VARIANT(2, 2, 1, 8, 9, 1, false),  // 3: M64  x N512, 8 waves
    ND_VARIANT(1, 2, 1, 4, 9, 1, false),  // 4: M32  x N256, 4 waves (narrow nets / tests)
    ND_VARIANT(2, 2, 1, 4, 1, 2, false),  // 5: 1x1, M64 x N256
    ND_VARIANT(2, 2, 2, 2, 1, 2, false),  // 6: 1x1, M128 x N128
    ND_VARIANT(2, 2, 1, 4, 1, 2, true),   // 7: up (2x2 s2), M64 x N256
    ND_VARIANT(2, 2, 2, 2, 1, 2, true),   // 8: up, M128 x N128
    ND_VARIANT(2, 2, 2, 4, 1, 2, true),   // 9: up, M128 x N256, 8 waves
    ND_VARIANT(2, 2, 1, 4, 1, 1, false),  // 10: 1x1, KBC=1 (Cin == 8)
    ND_VARIANT(2, 2, 1, 4, 1, 1, true),   // 11: up, KBC=1
};
static const int g_nvariants = (int)(sizeof(g_variants) / sizeof(g_variants[0]));

int nd_conv_variant_count() { return g_nvariants; }
const char *nd_conv_variant_label(int v) { return (v >= 0 && v < g_nvariants) ? g_variants[v].name : ""; }

static int pick_variant(const ConvDesc &d, int M, long NP) {
    const int taps = nd_taps(d.kind);
    const bool up = d.kind == ND_CONVT2S2;
    const int KB = nd_kblocks(d.cin);
    if (taps == 9) {
        if (M <= 32) return 4;
        if (M <= 64) return 0;
        return 1;
    }
    if (KB % 2) return up ? 11 : 10;
    if (up) return M <= 64 ? 7 : 8;
    return M <= 64 ? 5 : 6;
}

int nd_launch_conv_f32(const ConvDesc &d, hipStream_t stream) {
    const int taps = nd_taps(d.kind);
    const bool up = d.kind == ND_CONVT2S2;
    const int KB = nd_kblocks(d.cin);
    const int M = up ? 4 * d.cout : d.cout;
    if (d.cout % 4) ND_FAIL(ND_EINVAL, "conv: cout=%d must be a multiple of 4", d.cout);
    if (d.in.planes < 2 * KB) ND_FAIL(ND_EINVAL, "conv: input buffer has %d planes, needs %d", d.in.planes, 2 * KB);
    const long NP = d.in.used();
    if (NP >= (1L << 31)) ND_FAIL(ND_EINVAL, "conv: %ld linear pixels exceed the int32 index range", NP);

    int v = d.variant >= 0 ? d.variant : pick_variant(d, M, NP);
    if (v < 0 || v >= g_nvariants) ND_FAIL(ND_EINVAL, "conv: unknown variant %d", v);
    const Variant &V = g_variants[v];
    if (V.taps != taps || V.up != up) ND_FAIL(ND_EINVAL, "conv: variant %s does not match layer kind %d", V.name, d.kind);
    if (KB % V.kbc) ND_FAIL(ND_EINVAL, "conv: Cin/8=%d not a multiple of the variant's K chunk %d", KB, V.kbc);

    ConvParams p;
    p.in = (const f32x4 *)d.in.base;
    p.wpk = d.wpk;
    p.bias = d.bias;
    p.out = (f32x4 *)d.out.base;
    p.in_plane = d.in.np();
    p.out_plane = d.out.np();
    p.NP = (int)NP;
    p.P = d.in.Hb * d.in.Wb;
    p.Wb = d.in.Wb;
    p.Hv = taps == 9 ? d.in.Hb - 2 : d.in.Hb;
    p.Wv = taps == 9 ? d.in.Wb - 2 : d.in.Wb;
    p.KB = KB;
    p.M = M;
    p.cout = d.cout;
    p.Po = d.out.Hb * d.out.Wb;
    p.Wo = d.out.Wb;
    p.opad = d.out.pad;
    p.out_plane0 = d.out_plane0;
    p.act = d.act;
    p.slope = d.slope;
    p.slope_dev = d.slope_dev;

    // destination geometry must hold the result
    const int oh = up ? 2 * p.Hv : p.Hv, ow = up ? 2 * p.Wv : p.Wv;
    if (taps == 1 && d.in.pad != 0) ND_FAIL(ND_EINVAL, "conv: 1-tap layers read unbordered buffers only");
    if (d.out.Hb != oh + 2 * d.out.pad || d.out.Wb != ow + 2 * d.out.pad || d.out.B != d.in.B)
        ND_FAIL(ND_EINVAL, "conv: destination %dx%dx%d(pad %d) does not fit result %dx%dx%d", d.out.B, d.out.Hb, d.out.Wb,
                d.out.pad, d.in.B, oh, ow);
    if (d.out_plane0 + d.cout / 4 > d.out.planes) ND_FAIL(ND_EINVAL, "conv: destination planes overflow");

    const int halo = taps == 9 ? 2 * p.Wb + 2 : 0;
    const int G = (V.nblk + halo + 63) / 64;
    const size_t lds = 2 * ((size_t)(V.mblk / 32) * V.kbc * taps * 1024 + (size_t)2 * V.kbc * G * 1024);
    if (lds > 160 * 1024) ND_FAIL(ND_EINVAL, "conv: %zu B of LDS needed (row width %d too large for variant %s)", lds, p.Wb, V.name);
    ND_HIP(hipFuncSetAttribute((const void *)V.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    // only pixels up to the last valid output need a workgroup
    const long last = NP - (taps == 9 ? 2L * p.Wb + 2 : 0);
    dim3 grid((unsigned)((last + V.nblk - 1) / V.nblk), (unsigned)((M + V.mblk - 1) / V.mblk));
    hipLaunchKernelGGL(V.fn, grid, dim3(V.threads), lds, stream, p);
    ND_HIP(hipGetLastError());
    return ND_OK;
}
