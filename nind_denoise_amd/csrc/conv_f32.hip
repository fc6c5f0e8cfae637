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
//   (MFMA "B" operand), K = taps * Cin.  An N tile enumerates VALID output pixels only (compact index
//   r = y*Wv + x inside an image); the input pixels they read still form ONE contiguous range of the linear input index,
//   so the LDS halo image stays one contiguous copy per plane and only the per-lane fragment offset (computed once per
//   tile) knows about the row / image gaps.
//
// MFMA fragment use (f32 32x32x2: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]):
//   one ds_read_b128 per operand fetches 4 consecutive channels c..c+3 of channel-quad (2*kb + h), h = lane>>5;
//   MFMA step s (0..3) therefore contracts channels {8kb+s, 8kb+4+s}: the K order is permuted identically for A and B.
//   The accumulator holds, for lane (j,h), rows (r&3) + 8*(r>>2) + 4h: registers 4g..4g+3 are four CONSECUTIVE output
//   channels 8g+4h.. of pixel j  => one float4 store per (g) lands in the quad-planar output, 512 B contiguous per
//   half wave.
//
// Schedule: PERSISTENT workgroups (one per CU) walk a stream of (output tile, K chunk) steps.  LDS holds NSTAGE stage
// images (weights + activation halo image of one K chunk) filled by LDS-DMA (global_load_lds_dwordx4):
//   NSTAGE = 3: the DMA of step s+2 is issued at the top of step s; one barrier per step proves that step s+1 has landed
//               for every wave, so the first fragments of step s+1 are read BEFORE the next barrier and the matrix pipe
//               never drains at a step boundary -- nor at a tile boundary: the next tile's first chunks are already
//               in flight while the epilogue stores of the finished tile are issued.
//   NSTAGE = 2: (rows too wide for three stage images) DMA of step s+1 issued at the top of step s, fragments read after
//               the barrier.
// Workgroup ids are remapped so that the 32 workgroups sharing an XCD (and its L2) walk adjacent tiles.
#include <stdlib.h>

#include <type_traits>

#include "nd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// One 16-byte MFMA operand fragment per lane and K block:
//   ND_F32  : 4 fp32 channels  -> 4 x v_mfma_f32_32x32x2_f32   (K block = 8 channels over the two lane halves)
//   ND_BF16 : 8 bf16 channels  -> 1 x v_mfma_f32_32x32x16_bf16 (K block = 16 channels)
//   ND_F16  : 8 fp16 channels  -> 1 x v_mfma_f32_32x32x16_f16
template <int DT> struct FragOf { typedef f32x4 type; };
template <> struct FragOf<ND_BF16> { typedef bf16x8 type; };
template <> struct FragOf<ND_F16> { typedef f16x8 type; };

struct ConvParams {
    const f32x4 *in;     // plane 0 of the input buffer
    const float *wpk;    // packed weights
    const float *bias;   // [mtiles*32]
    f32x4 *out;          // plane 0 of the destination buffer
    long in_plane;       // float4 per input plane
    long out_plane;      // float4 per output plane
    int nimg;            // images in the batch
    int P, Wb;           // Hb*Wb, Wb of the input buffer
    int Hv, Wv, PV;      // valid output rows / cols per image, PV = Hv*Wv
    int tpi;             // > 0: tiles never cross an image, tpi tiles per image (wide rows); 0: tiles run over the whole batch
    int G;               // 64-pixel DMA pieces per plane and stage (covers the largest input span of a tile + halo)
    int KB;              // Cin / 8
    int M;               // GEMM rows (Cout, or 4*Cout for the 2x2 stride-2 transpose)
    int cout;            // output channels
    int Po, Wo, opad;    // destination buffer: Hb*Wb, Wb, border
    int out_plane0;      // first destination plane
    int act;
    float slope;
    const float *slope_dev;
    int n_tiles_n;       // N tiles (pixels / NBLK, rounded up)
    int n_tiles_m;       // M tiles (rows / MBLK, rounded up)
    int ablate;          // diagnostics only (NIND_ABLATE): 1 no DMA, 2 no barrier, 4 no stores, 8 no LDS fragment reads,
                         // 16 no vmcnt wait, 32 DMA of weights only, 64 DMA of activations only
};

__device__ __forceinline__ void glds16(const void *g, void *l) {
    // 64 lanes x 16 B: per-lane global source, LDS destination = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// 8 consecutive floats at a wave-uniform address through the scalar cache (lgkmcnt, not vmcnt)
__device__ __forceinline__ f32x8 sload8(const float *p) {
    f32x8 v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
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
template <int DT, int MR, int NR, int WM, int WN, int TAPS, int KBC, bool UP, int NSTAGE>
__global__ __launch_bounds__(64 * WM * WN) void conv_qp(ConvParams p) {
    typedef typename FragOf<DT>::type Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN;
    constexpr int MTB = MR * WM;                    // 32-row tiles per workgroup
    constexpr int NBLK = 32 * NR * WN;              // pixels per workgroup tile
    constexpr int WBYTES = MTB * KBC * TAPS * 1024; // weight bytes per stage
    constexpr int STEPS = KBC * TAPS;               // fragment sub-steps per K chunk

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, h = lane >> 5;

    // XCD-aware id: hardware deals workgroups round-robin over the 8 XCDs; give each XCD a contiguous id range
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int vb = (nwg % 8 == 0) ? (bid % 8) * (nwg / 8) + bid / 8 : bid;

    const int G = p.G;
    const int planeB = G * 1024;
    const int stageB = WBYTES + 2 * KBC * planeB;

    int toff[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) toff[t] = (TAPS == 9) ? ((t / 3) * p.Wb + (t % 3)) * 16 : 0;

    const int nchunks = p.KB / KBC;
    const int ntiles = p.n_tiles_n * p.n_tiles_m;
    const int my_tiles = vb < ntiles ? (ntiles - vb + nwg - 1) / nwg : 0;
    const int nsteps = my_tiles * nchunks;
    if (nsteps == 0) return;

    // ---- tile geometry: N tile nb covers NBLK consecutive valid pixels (of one image if p.tpi, else of the batch)
    // (image, compact index) of the l-th pixel of N tile nb; pixels past the end are clamped to the last valid one
    auto pixel_of = [&](int nb, int l, int &img, int &r) -> bool {
        bool ok;
        if (p.tpi) {
            img = nb / p.tpi;
            r = (nb - img * p.tpi) * NBLK + l;
            ok = r < p.PV;
            r = ok ? r : p.PV - 1;
        } else {
            const long g = (long)nb * NBLK + l;
            const long tot = (long)p.nimg * p.PV;
            ok = g < tot;
            const long gc = ok ? g : tot - 1;
            img = (int)(gc / p.PV);
            r = (int)(gc - (long)img * p.PV);
        }
        return ok;
    };
    // linear input index of (image, compact index)
    auto q_of = [&](int img, int r) -> long {
        const int y = r / p.Wv;
        return (long)img * p.P + (long)y * p.Wb + (r - y * p.Wv);
    };
    auto tile_q0 = [&](int nb) -> long {   // wave-uniform: input index of the tile's first pixel
        int img, r;
        pixel_of(nb, 0, img, r);
        return q_of(img, r);
    };

    // ---- DMA cursor: walks the same (tile, chunk) stream as the compute loop, NSTAGE-1 steps ahead
    int f_id = vb, f_c = 0, f_stage = 0, issued = 0;
    const float *f_w;
    const f32x4 *f_a;
    auto set_fill_tile = [&](int id) {
        const int nb = id / p.n_tiles_m, mb = id - nb * p.n_tiles_m;   // M tiles of one N tile run back to back
        f_w = p.wpk + (size_t)mb * MTB * p.KB * TAPS * 256 + lane * 4;
        f_a = p.in + tile_q0(nb) + lane;
    };
    set_fill_tile(f_id);
    // With two waves per SIMD (NW == 8) only waves 0..3 -- one per SIMD -- issue the DMA: their SIMD partners (waves
    // 4..7) go straight to their MFMAs, so the matrix pipe is fed while the DMA instructions are being issued.
    constexpr int NFILL = NW > 4 ? 4 : NW;
    auto fill_next = [&]() {
        if (issued >= nsteps) return;
        char *sb = smem + f_stage * stageB;
        if (wave < NFILL && !(p.ablate & 1)) {
            if (!(p.ablate & 64))
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) {
                const float *src = f_w + ((size_t)mt * p.KB + (size_t)f_c * KBC) * TAPS * 256;
                char *dst = sb + mt * KBC * TAPS * 1024;
                for (int q = wave; q < KBC * TAPS; q += NFILL) glds16(src + q * 256, dst + q * 1024);
            }
            if (!(p.ablate & 32))
#pragma unroll
            for (int pl = 0; pl < 2 * KBC; ++pl) {
                const f32x4 *src = f_a + (size_t)(f_c * 2 * KBC + pl) * p.in_plane;
                char *dst = sb + WBYTES + pl * planeB;
                for (int g = wave; g < G; g += NFILL) glds16(src + g * 64, dst + g * 1024);
            }
        }
        ++issued;
        f_stage = (f_stage + 1 == NSTAGE) ? 0 : f_stage + 1;
        if (++f_c == nchunks) {
            f_c = 0;
            f_id += nwg;
            if (f_id < ntiles) set_fill_tile(f_id);
        }
    };

    // ---- fragments
    const int aOff = (wm * MR) * KBC * TAPS * 1024 + lane * 16;
    // byte offset of this lane's pixel inside a stage's activation image, for each of the wave's NR pixel groups
    struct BOff { int v[NR]; };
    auto lane_offsets = [&](int id) -> BOff {
        BOff o;
        const int nb = id / p.n_tiles_m;
        const long q0 = tile_q0(nb);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            int img, r;
            pixel_of(nb, (wn * NR + nr) * 32 + j, img, r);
            o.v[nr] = WBYTES + h * planeB + (int)(q_of(img, r) - q0) * 16;
        }
        return o;
    };
    BOff bOff = lane_offsets(vb), bOffN = bOff;
    Frag a[2][MR] = {}, b[2][NR] = {};
    auto load_frags = [&](int buf, const char *sb, int step, const BOff &bo) {
        if (p.ablate & 8) return;
        const int kbl = step / TAPS, t = step % TAPS;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
            a[buf][mr] = *(const Frag *)(sb + aOff + ((mr * KBC + kbl) * TAPS + t) * 1024);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
            b[buf][nr] = *(const Frag *)(sb + bo.v[nr] + kbl * 2 * planeB + toff[t]);
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int x = 0; x < MR; ++x)
#pragma unroll
        for (int y = 0; y < NR; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

    const float slope = p.act == ND_ACT_NONE ? 1.f : (p.slope_dev ? *p.slope_dev : p.slope);

    // ---- epilogue of one finished tile: bias + activation, float4 stores into the (bordered, concatenated) destination
    auto epilogue = [&](int id, auto generic_act) {
        const int nb = id / p.n_tiles_m, mb = id - nb * p.n_tiles_m;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            int bi, r;
            const bool valid = pixel_of(nb, (wn * NR + nr) * 32 + j, bi, r);
            const int y = r / p.Wv;
            const int x = r - y * p.Wv;
            const long pbase = UP ? (long)bi * p.Po + (long)(2 * y + p.opad) * p.Wo + (2 * x + p.opad)
                                  : (long)bi * p.Po + (long)(y + p.opad) * p.Wo + (x + p.opad);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // bias through the scalar cache (a vector load here would make the compiler wait vmcnt(0), i.e. for
                    // the LDS-DMA of the next steps that is in flight during the epilogue)
                    const int m8 = ((mb * MTB + wm * MR + mr) * 32) + 8 * g;   // wave-uniform
                    const f32x8 b8 = sload8(p.bias + m8);
                    const int m4 = m8 + 4 * h;
                    if (valid && m4 < p.M) {
                        f32x4 bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = h ? b8[4 + e] : b8[e];
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = acc[mr][nr][4 * g + e] + bv[e];
                            if constexpr (decltype(generic_act)::value)
                                v[e] = apply_act(t, p.act, slope);
                            else
                                v[e] = t > 0.f ? t : t * slope;  // PReLU; "no activation" is slope 1
                        }
                        // destination channel (a multiple of 4) and pixel
                        int co = m4;
                        long pix = pbase;
                        if (UP) {
                            const int ab = m4 / p.cout;
                            co = m4 - ab * p.cout;
                            pix += (long)(ab >> 1) * p.Wo + (ab & 1);
                        }
                        if (!(p.ablate & 4)) {
                            if constexpr (DT == ND_F32) {
                                p.out[(long)(p.out_plane0 + (co >> 2)) * p.out_plane + pix] = v;
                            } else {
                                // a 16-bit plane element holds 8 channels: this lane owns its lower or upper half (8 bytes)
                                char *dst = (char *)p.out + (((long)(p.out_plane0 + (co >> 3)) * p.out_plane + pix) << 4) + ((co >> 2) & 1) * 8;
                                if constexpr (DT == ND_BF16) {
                                    bf16x4 o;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                                    *(bf16x4 *)dst = o;
                                } else {
                                    f16x4 o;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                                    *(f16x4 *)dst = o;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mr][nr][4 * g + e] = 0.f;
                }
            }
        }
    };

    // ---- prologue
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) fill_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (NSTAGE == 3) load_frags(0, smem, 0, bOff);

    int c_id = vb, c_c = 0, c_stage = 0;
    for (int s = 0; s < nsteps; ++s) {
        // my share of the youngest outstanding DMA was issued one whole step ago
        if (!(p.ablate & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(p.ablate & 2)) __builtin_amdgcn_s_barrier();
        fill_next();
        const char *sb = smem + c_stage * stageB;
        const int n_stage = (c_stage + 1 == NSTAGE) ? 0 : c_stage + 1;
        const bool last_chunk = c_c + 1 == nchunks;
        // the fragments prefetched at the end of a tile's last chunk belong to the NEXT tile: its lane offsets
        if (last_chunk && c_id + nwg < ntiles) bOffN = lane_offsets(c_id + nwg);
        if (NSTAGE == 2) load_frags(0, sb, 0, bOff);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 1 < STEPS)
                load_frags((st + 1) & 1, sb, st + 1, bOff);
            else if (NSTAGE == 3 && s + 1 < nsteps)   // next step's first fragments, before its barrier
                load_frags((st + 1) & 1, smem + n_stage * stageB, 0, last_chunk ? bOffN : bOff);
            if constexpr (DT == ND_F32) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr)
                            acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][mr][q], b[st & 1][nr][q], acc[mr][nr], 0, 0, 0);
            } else {
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        if constexpr (DT == ND_BF16)
                            acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[st & 1][mr], b[st & 1][nr], acc[mr][nr], 0, 0, 0);
                        else
                            acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[st & 1][mr], b[st & 1][nr], acc[mr][nr], 0, 0, 0);
                    }
            }
        }
        if (NSTAGE == 3 && (STEPS & 1)) {
            // an odd number of sub-steps leaves the prefetched fragments in buffer 1: the next step starts from buffer 0
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) a[0][mr] = a[1][mr];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) b[0][nr] = b[1][nr];
        }
        c_stage = n_stage;
        if (++c_c == nchunks) {
            if (p.act <= ND_ACT_PRELU)
                epilogue(c_id, std::false_type{});
            else
                epilogue(c_id, std::true_type{});
            c_c = 0;
            c_id += nwg;
            bOff = bOffN;
        }
    }
}

// ------------------------------------------------------------------ variants and dispatch
struct Variant {
    const char *name;
    int dt, mblk, nblk, threads, taps, kbc, nstage;
    bool up;
    void (*fn)(ConvParams);
};

#define ND_VARIANT(DT, DTN, MR, NR, WM, WN, TAPS, KBC, UP, NS)                                                         \
    {                                                                                                                  \
        DTN "_m" #MR "x" #WM "_n" #NR "x" #WN "_t" #TAPS "_k" #KBC "_up" #UP "_s" #NS, DT, 32 * MR * WM, 32 * NR * WN,  \
            64 * WM * WN, TAPS, KBC, NS, UP, conv_qp<DT, MR, NR, WM, WN, TAPS, KBC, UP, NS>                            \
    }
// the shapes every storage type gets (index inside a dtype group)
#define ND_VARIANT_GROUP(DT, DTN)                                                                                      \
    ND_VARIANT(DT, DTN, 2, 2, 1, 8, 9, 1, false, 3),  /* 0: M64  x N512, 8 waves, 3 stages  (default 3x3)           */ \
    ND_VARIANT(DT, DTN, 2, 2, 1, 8, 9, 1, false, 2),  /* 1: M64  x N512, 8 waves, 2 stages  (wide rows: cs >= ~400) */ \
    ND_VARIANT(DT, DTN, 2, 2, 1, 4, 9, 1, false, 2),  /* 2: M64  x N256, 4 waves, 2 stages  (widest rows)           */ \
    ND_VARIANT(DT, DTN, 1, 2, 1, 8, 9, 1, false, 3),  /* 3: M32  x N512, 8 waves (narrow nets / tests)              */ \
    ND_VARIANT(DT, DTN, 2, 2, 1, 8, 1, 2, false, 3),  /* 4: 1x1, M64 x N512                                         */ \
    ND_VARIANT(DT, DTN, 2, 2, 1, 8, 1, 1, false, 3),  /* 5: 1x1, single-K-block chunks                              */ \
    ND_VARIANT(DT, DTN, 2, 2, 1, 8, 1, 2, true, 3),   /* 6: up (2x2 s2), M64 x N512                                 */ \
    ND_VARIANT(DT, DTN, 2, 2, 1, 8, 1, 1, true, 3),   /* 7: up, single-K-block chunks                               */ \
    ND_VARIANT(DT, DTN, 4, 2, 2, 4, 1, 2, true, 3)    /* 8: up, M256 x N256, 8 waves x (128x64)                     */
constexpr int kGroup = 9;

static const Variant g_variants[] = {
    ND_VARIANT_GROUP(ND_F32, "f32"),
    ND_VARIANT_GROUP(ND_BF16, "bf16"),
    ND_VARIANT_GROUP(ND_F16, "f16"),
    // fp32-only experiments
    ND_VARIANT(ND_F32, "f32", 2, 2, 2, 4, 9, 1, false, 3),  // M128 x N256, 8 waves, 3 stages
    ND_VARIANT(ND_F32, "f32", 2, 2, 1, 4, 9, 1, false, 3),  // M64  x N256, 4 waves, 3 stages
    ND_VARIANT(ND_F32, "f32", 2, 2, 2, 4, 1, 2, true, 3),   // up, M128 x N256
};
static const int g_nvariants = (int)(sizeof(g_variants) / sizeof(g_variants[0]));

int nd_conv_variant_count() { return g_nvariants; }
const char *nd_conv_variant_label(int v) { return (v >= 0 && v < g_nvariants) ? g_variants[v].name : ""; }

// Largest input span (pixels) of one N tile + 3x3 halo.  cross = tiles may run across image boundaries.
static int tile_span(const Variant &V, const QpBuf &in, bool cross) {
    const int taps = V.taps;
    const int Hv = taps == 9 ? in.Hb - 2 : in.Hb, Wv = taps == 9 ? in.Wb - 2 : in.Wb;
    const int n = V.nblk;
    int span = n + (in.Wb - Wv) * ((n - 1) / Wv + 1);
    if (cross) span += (in.Hb - Hv) * in.Wb * ((n - 1) / (Hv * Wv) + 1);
    if (taps == 9) span += 2 * in.Wb + 2;
    return span;
}
static size_t lds_for(const Variant &V, int G) {
    return (size_t)V.nstage * ((size_t)(V.mblk / 32) * V.kbc * V.taps * 1024 + (size_t)2 * V.kbc * G * 1024);
}
// tiles may cross images when that still fits the LDS (small images: no padding of every image to a tile multiple)
static size_t variant_lds(const Variant &V, const QpBuf &in, bool *cross_out = nullptr, int *G_out = nullptr) {
    const size_t kMax = 160 * 1024;
    int G = (tile_span(V, in, true) + 63) / 64;
    bool cross = true;
    if (lds_for(V, G) > kMax) {
        cross = false;
        G = (tile_span(V, in, false) + 63) / 64;
    }
    if (cross_out) *cross_out = cross;
    if (G_out) *G_out = G;
    return lds_for(V, G);
}

static const size_t kMaxLds = 160 * 1024;

static int pick_variant(const ConvDesc &d, int M) {
    const int taps = nd_taps(d.kind);
    const bool up = d.kind == ND_CONVT2S2;
    const int dt = d.in.dt;
    const int KB = nd_kblocks(d.cin, dt);
    const int g0 = dt * kGroup;
    if (taps == 9) {
        const int order[] = {M <= 32 ? 3 : 0, 0, 1, 2};
        for (int v : order)
            if (variant_lds(g_variants[g0 + v], d.in) <= kMaxLds) return g0 + v;
        return g0 + 2;
    }
    if (KB % 2) return g0 + (up ? 7 : 5);
    if (up) return g0 + (M >= 256 ? 8 : 6);
    return g0 + 4;
}

static int g_num_cus = 0;
static int g_lds_set[64] = {0};

int nd_launch_conv(const ConvDesc &d, hipStream_t stream) {
    const int taps = nd_taps(d.kind);
    const bool up = d.kind == ND_CONVT2S2;
    const int dt = d.in.dt;
    if (dt < ND_F32 || dt > ND_F16 || d.out.dt != dt) ND_FAIL(ND_EINVAL, "conv: input / output storage types %d / %d", d.in.dt, d.out.dt);
    const int KB = nd_kblocks(d.cin, dt);
    const int M = up ? 4 * d.cout : d.cout;
    if (d.cout % nd_cpp(dt)) ND_FAIL(ND_EINVAL, "conv: cout=%d must be a multiple of %d", d.cout, nd_cpp(dt));
    if (d.in.planes < 2 * KB) ND_FAIL(ND_EINVAL, "conv: input buffer has %d planes, needs %d", d.in.planes, 2 * KB);
    const long NP = d.in.used();
    if (NP >= (1L << 31)) ND_FAIL(ND_EINVAL, "conv: %ld linear pixels exceed the int32 index range", NP);

    int v = d.variant >= 0 ? d.variant : pick_variant(d, M);
    if (v < 0 || v >= g_nvariants) ND_FAIL(ND_EINVAL, "conv: unknown variant %d", v);
    const Variant &V = g_variants[v];
    if (V.taps != taps || V.up != up || V.dt != dt) ND_FAIL(ND_EINVAL, "conv: variant %s does not match layer kind %d / dtype %d", V.name, d.kind, dt);
    if (KB % V.kbc) ND_FAIL(ND_EINVAL, "conv: Cin/8=%d not a multiple of the variant's K chunk %d", KB, V.kbc);

    ConvParams p;
    p.in = (const f32x4 *)d.in.base;
    p.wpk = d.wpk;
    p.bias = d.bias;
    p.out = (f32x4 *)d.out.base;
    p.in_plane = d.in.np();
    p.out_plane = d.out.np();
    p.nimg = d.in.B;
    p.P = d.in.Hb * d.in.Wb;
    p.Wb = d.in.Wb;
    p.Hv = taps == 9 ? d.in.Hb - 2 : d.in.Hb;
    p.Wv = taps == 9 ? d.in.Wb - 2 : d.in.Wb;
    p.PV = p.Hv * p.Wv;
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
    static const int ablate = getenv("NIND_ABLATE") ? atoi(getenv("NIND_ABLATE")) : 0;
    p.ablate = ablate;

    // destination geometry must hold the result
    const int oh = up ? 2 * p.Hv : p.Hv, ow = up ? 2 * p.Wv : p.Wv;
    if (taps == 1 && d.in.pad != 0) ND_FAIL(ND_EINVAL, "conv: 1-tap layers read unbordered buffers only");
    // (a 2x2 stride-2 result may be smaller than its destination: UNet's F.pad fix-up for odd sizes, ThirdPartyNets.py:110-118)
    const bool fits = up ? (d.out.Hb >= oh + 2 * d.out.pad && d.out.Wb >= ow + 2 * d.out.pad)
                         : (d.out.Hb == oh + 2 * d.out.pad && d.out.Wb == ow + 2 * d.out.pad);
    if (!fits || d.out.B != d.in.B)
        ND_FAIL(ND_EINVAL, "conv: destination %dx%dx%d(pad %d) does not fit result %dx%dx%d", d.out.B, d.out.Hb, d.out.Wb,
                d.out.pad, d.in.B, oh, ow);
    if (d.out_plane0 + d.cout / nd_cpp(dt) > d.out.planes) ND_FAIL(ND_EINVAL, "conv: destination planes overflow");

    bool cross = true;
    const size_t lds = variant_lds(V, d.in, &cross, &p.G);
    if (lds > kMaxLds) ND_FAIL(ND_EINVAL, "conv: %zu B of LDS needed (row width %d too large for variant %s)", lds, p.Wb, V.name);
    if ((int)lds > g_lds_set[v]) {
        ND_HIP(hipFuncSetAttribute((const void *)V.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        g_lds_set[v] = (int)lds;
    }
    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        ND_HIP(hipGetDevice(&dev));
        ND_HIP(hipGetDeviceProperties(&prop, dev));
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }

    if (cross) {
        p.tpi = 0;
        p.n_tiles_n = (int)(((long)p.nimg * p.PV + V.nblk - 1) / V.nblk);
    } else {
        p.tpi = (p.PV + V.nblk - 1) / V.nblk;
        p.n_tiles_n = p.tpi * p.nimg;
    }
    p.n_tiles_m = (M + V.mblk - 1) / V.mblk;
    const long ntiles = (long)p.n_tiles_n * p.n_tiles_m;
    const int per_cu = lds * 2 <= kMaxLds && V.threads <= 256 ? 2 : 1;
    const long grid = ntiles < (long)g_num_cus * per_cu ? ntiles : (long)g_num_cus * per_cu;
    hipLaunchKernelGGL(V.fn, dim3((unsigned)grid), dim3(V.threads), lds, stream, p);
    ND_HIP(hipGetLastError());
    return ND_OK;
}
