// 3x3 convolution with a 1-D Winograd F(2,3) | F(4,3) along x inside the implicit-GEMM kernel (fp32 inference, narrow full-resolution layers).
//
// The 64-channel layers of UtNet are HBM-bound in any multi-pass Winograd form (winograd.hip) and MFMA-bound in the direct
// form (conv_qp.inc).  This kernel keeps the direct kernel's structure -- persistent workgroups, LDS-DMA halo images of the RAW
// input, split-K tail -- and shortens only the matrix work:  for an output pixel PAIR (x0, x0+1) and kernel row ky
//     d0..d3 = X[y+ky][x0 .. x0+3]          v = (d0-d2, d1+d2, d2-d1, d1-d3)                  (in registers, 4 VALU ops per float4)
//     m_xi  += U[ky][xi] (Cout x Cin) * v_xi,   U[ky] = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) of that kernel row      (MFMA)
//     Y[x0] = m0+m1+m2,  Y[x0+1] = m1-m2-m3                                                   (in registers, epilogue)
// 12 weight planes and 4 accumulator sets per pair instead of 9 taps x 2 pixels: 2/3 of the MFMAs, the same LDS-DMA bytes.
// T = 4 is the same with F(4,3): groups of 4 pixels, 6 positions, 18 weight planes, 6 accumulator sets (one 32-row tile per
// wave instead of two): 1/2 of the MFMAs.  An N tile enumerates pixel GROUPS row by row; a group hanging over the row end takes
// zeros for the inputs only its out-of-row outputs need (so a tile's bits never depend on what lies behind the row: the next
// image of the batch or slack) and does not store those outputs.  Result differs from the direct kernel by fp32
// re-association (~1e-6 / ~5e-6).
#include "conv_qp.inc"

namespace {

template <int T> struct W1;
template <> struct W1<2> {
    __device__ static void in(const f32x4 *d, f32x4 *v) {
        v[0] = d[0] - d[2];
        v[1] = d[1] + d[2];
        v[2] = d[2] - d[1];
        v[3] = d[1] - d[3];
    }
    __device__ static void out(const float *m, float *y) {
        y[0] = m[0] + m[1] + m[2];
        y[1] = m[1] - m[2] - m[3];
    }
    static void g(const double *w, double *u) {
        u[0] = w[0];
        u[1] = 0.5 * (w[0] + w[1] + w[2]);
        u[2] = 0.5 * (w[0] - w[1] + w[2]);
        u[3] = w[2];
    }
};
template <> struct W1<4> {
    __device__ static void in(const f32x4 *d, f32x4 *v) {
        v[0] = 4.f * d[0] - 5.f * d[2] + d[4];
        v[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
        v[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
        v[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
        v[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
        v[5] = 4.f * d[1] - 5.f * d[3] + d[5];
    }
    __device__ static void out(const float *m, float *y) {
        y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
        y[1] = m[1] - m[2] + 2.f * (m[3] - m[4]);
        y[2] = m[1] + m[2] + 4.f * (m[3] + m[4]);
        y[3] = m[1] - m[2] + 8.f * (m[3] - m[4]) + m[5];
    }
    static void g(const double *w, double *u) {
        u[0] = w[0] / 4;
        u[1] = -(w[0] + w[1] + w[2]) / 6;
        u[2] = -(w[0] - w[1] + w[2]) / 6;
        u[3] = w[0] / 24 + w[1] / 12 + w[2] / 6;
        u[4] = w[0] / 24 - w[1] / 12 + w[2] / 6;
        u[5] = w[2];
    }
};

// T : output pixels per group (2 | 4);  MR : 32-row tiles per wave (x T+2 position sets);  one 32-group tile per wave;  WM x WN waves
template <int T, int MR, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(64 * WM * WN) void conv_w1d(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = T + 2;               // positions
    constexpr int NW = WM * WN;
    constexpr int MTB = MR * WM;
    constexpr int NBLK = 32 * WN;           // pixel groups per workgroup tile
    constexpr int TAPS = 3 * NP;            // weight planes per K block: 3 kernel rows x positions
    constexpr int WBYTES = MTB * TAPS * 1024;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, h = lane >> 5;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int vb = (nwg % 8 == 0) ? (bid % 8) * (nwg / 8) + bid / 8 : bid;

    const int G = p.G;
    const int planeB = G * 1024;
    const int stageB = WBYTES + 2 * planeB;

    const int nchunks = p.KB;
    const int nitems = p.nitems;
    const int my_items = vb < nitems ? (nitems - vb + nwg - 1) / nwg : 0;
    struct Item { int tile, c0, c1, slice; };
    auto decode = [&](int w) -> Item {
        Item it;
        if (w < p.split_first) {
            it.tile = w; it.c0 = 0; it.c1 = nchunks; it.slice = -1;
        } else {
            const int u = w - p.split_first;
            const int t = u / p.S, ks = u - t * p.S;
            it.tile = p.split_first + t;
            it.c0 = ks * p.cps;
            it.c1 = it.c0 + p.cps < nchunks ? it.c0 + p.cps : nchunks;
            it.slice = u;
        }
        return it;
    };
    int nsteps = 0;
    if (p.split_first >= nitems)
        nsteps = my_items * nchunks;
    else
        for (int i = 0; i < my_items; ++i) {
            const Item it = decode(vb + i * nwg);
            nsteps += it.c1 - it.c0;
        }
    if (nsteps == 0) return;

    // (image, compact GROUP index) of the l-th pixel group of N tile nb  (p.Wv = groups per row, p.PV = groups per image)
    auto group_of = [&](int nb, int l, int &img, int &r) -> bool {
        bool ok;
        if (p.tpi) {
            img = nb / p.tpi;
            r = (nb - img * p.tpi) * NBLK + l;
            ok = r < p.PV;
            r = ok ? r : p.PV - 1;
        } else {
            const unsigned g = (unsigned)nb * NBLK + l;      // < 2^31: the launcher checks the pixel count
            const unsigned tot = (unsigned)p.nimg * p.PV;
            ok = g < tot;
            const unsigned gc = ok ? g : tot - 1;
            img = (int)(gc / (unsigned)p.PV);
            r = (int)(gc - (unsigned)img * p.PV);
        }
        return ok;
    };
    auto q_of = [&](int img, int r) -> long {   // linear input index of the group's first pixel
        const int y = r / p.Wv;
        return (long)img * p.P + (long)y * p.Wb + T * (r - y * p.Wv);
    };
    auto tile_q0 = [&](int nb) -> long {
        int img, r;
        group_of(nb, 0, img, r);
        return q_of(img, r);
    };

    // ---- DMA cursor (as in conv_qp)
    int f_id = vb, f_c = 0, f_end = 0, f_stage = 0, issued = 0;
    const float *f_w;
    const f32x4 *f_a;
    auto set_fill_tile = [&](int w) {
        const Item it = decode(w);
        f_c = it.c0;
        f_end = it.c1;
        const int nb = it.tile / p.n_tiles_m, mb = it.tile - nb * p.n_tiles_m;
        f_w = p.wpk + (size_t)mb * MTB * p.KB * TAPS * 256 + lane * 4;
        f_a = p.in + tile_q0(nb) + lane;
    };
    set_fill_tile(f_id);
    constexpr int NFILL = NW > 4 ? 4 : NW;
    auto fill_next = [&]() {
        if (issued >= nsteps) return;
        char *sb = smem + f_stage * stageB;
        if (wave < NFILL) {
#pragma unroll
            for (int mt = 0; mt < MTB; ++mt) {
                const float *src = f_w + ((size_t)mt * p.KB + (size_t)f_c) * TAPS * 256;
                char *dst = sb + mt * TAPS * 1024;
                for (int q = wave; q < TAPS; q += NFILL) glds16(src + q * 256, dst + q * 1024);
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const f32x4 *src = f_a + (size_t)(f_c * 2 + pl) * p.in_plane;
                char *dst = sb + WBYTES + pl * planeB;
                for (int g = wave; g < G; g += NFILL) glds16(src + g * 64, dst + g * 1024);
            }
        }
        ++issued;
        f_stage = (f_stage + 1 == NSTAGE) ? 0 : f_stage + 1;
        if (++f_c == f_end) {
            f_id += nwg;
            if (f_id < nitems) set_fill_tile(f_id);
        }
    };

    const int aOff = (wm * MR) * TAPS * 1024 + lane * 16;
    int keep = NP;   // inputs d[0 .. keep) of this lane's group feed in-row outputs (NP unless the group hangs over the row end)
    auto lane_offset = [&](int w) -> int {
        const int nb = decode(w).tile / p.n_tiles_m;
        int img, r;
        group_of(nb, wn * 32 + j, img, r);
        const int left = p.wpx - T * (r % p.Wv);
        keep = left + 2 < NP ? left + 2 : NP;
        return WBYTES + h * planeB + (int)(q_of(img, r) - tile_q0(nb)) * 16;
    };
    int bOff = lane_offset(vb);
    const int rowB = p.Wb * 16;

    f32x16 acc[NP][MR];
#pragma unroll
    for (int x = 0; x < NP; ++x)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;

    const float slope = p.act == ND_ACT_NONE ? 1.f : (p.slope_dev ? *p.slope_dev : p.slope);

    // combine the position accumulators of (mr, g) into the T pixels' sums (clears them)
    auto combine = [&](int mr, int g, f32x4 *y) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m[NP], o[T];
#pragma unroll
            for (int x = 0; x < NP; ++x) {
                m[x] = acc[x][mr][4 * g + e];
                acc[x][mr][4 * g + e] = 0.f;
            }
            W1<T>::out(m, o);
#pragma unroll
            for (int i = 0; i < T; ++i) y[i][e] = o[i];
        }
    };

    auto epilogue = [&](int id) {
        const int nb = id / p.n_tiles_m, mb = id - nb * p.n_tiles_m;
        int bi, r;
        const bool valid = group_of(nb, wn * 32 + j, bi, r);
        const int y = r / p.Wv, xp = r - y * p.Wv;
        const long pix = (long)bi * p.Po + (long)(y + p.opad) * p.Wo + T * xp + p.opad;
        const int left = p.wpx - T * xp;   // valid pixels of this group (the last group of a row may hang over its end)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m8 = ((mb * MTB + wm * MR + mr) * 32) + 8 * g;
                const f32x4 bq = sload_bias4(p.bias + m8, h);
                const int m4 = m8 + 4 * h;
                f32x4 yy[T];
                combine(mr, g, yy);
                if (valid && m4 < p.M) {
                    f32x4 *dst = p.out + (long)(p.out_plane0 + (m4 >> 2)) * p.out_plane + pix;
                    // training forward: keep acc + bias for the activation's backward pass (compact [C/4][B][Hv][wpx] planes)
                    f32x4 *pre = p.pre ? p.pre + (long)(m4 >> 2) * p.pre_plane + ((long)bi * p.Hv + y) * p.wpx + T * xp : nullptr;
#pragma unroll
                    for (int i = 0; i < T; ++i) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) yy[i][e] += bq[e];
                        if (pre && i < left) pre[i] = yy[i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) yy[i][e] = apply_act(yy[i][e], p.act, slope);
                        if (i < left) dst[i] = yy[i];
                    }
                }
            }
    };
    // a K slice of a split tile: combined raw sums in the tile-local layout of k_split_finish,
    // [channel quad][T * NBLK "pixels"] with pixel index = T * group + i (groups hanging over a row end keep their slots)
    auto epilogue_partial = [&](int slice) {
        f32x4 *dst = p.part + (size_t)slice * (MTB * 8) * (T * NBLK);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 yy[T];
                combine(mr, g, yy);
                f32x4 *d2 = dst + (size_t)((wm * MR + mr) * 8 + 2 * g + h) * (T * NBLK) + T * (wn * 32 + j);
#pragma unroll
                for (int i = 0; i < T; ++i) d2[i] = yy[i];
            }
    };

    // ---- prologue
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) fill_next();

    int c_id = vb, c_stage = 0;
    Item c_it = decode(vb);
    int c_c = c_it.c0;
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        fill_next();
        const char *sb = smem + c_stage * stageB;
        const char *bp = sb + bOff;
        f32x4 d[2][NP], a[2][MR];
#pragma unroll
        for (int i = 0; i < NP; ++i) d[0][i] = *(const f32x4 *)(bp + i * 16);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) a[0][mr] = *(const f32x4 *)(sb + aOff + (mr * TAPS) * 1024);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            if (ky + 1 < 3) {
#pragma unroll
                for (int i = 0; i < NP; ++i) d[(ky + 1) & 1][i] = *(const f32x4 *)(bp + (ky + 1) * rowB + i * 16);
            }
            if constexpr (T > 2) {   // (T = 2: the one over-read input only reaches the output that is not stored)
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 3; i < NP; ++i)
                    if (i >= keep) d[ky & 1][i] = zero;
            }
            f32x4 v[NP];
            W1<T>::in(d[ky & 1], v);
#pragma unroll
            for (int xi = 0; xi < NP; ++xi) {
                const int st = ky * NP + xi;
                if (st + 1 < TAPS) {   // weight fragments of the next (ky, position) ahead of this one's MFMAs
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr) a[(st + 1) & 1][mr] = *(const f32x4 *)(sb + aOff + (mr * TAPS + st + 1) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
                        acc[xi][mr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][mr][q], v[xi][q], acc[xi][mr], 0, 0, 0);
            }
        }
        c_stage = (c_stage + 1 == NSTAGE) ? 0 : c_stage + 1;
        if (++c_c == c_it.c1) {
            if (c_it.slice >= 0)
                epilogue_partial(c_it.slice);
            else
                epilogue(c_it.tile);
            c_id += nwg;
            if (c_id < nitems) {
                c_it = decode(c_id);
                bOff = lane_offset(c_id);
            }
            c_c = c_it.c0;
        }
    }
}

// workgroup shapes: 64 output channels x 512 pixels either way
struct W1Shape { int T, mblk, groups, threads, taps; };
constexpr int kW1dStages = 2;
//   T = 2: MR 2, WM 1, WN 8 -> 64 rows x 256 pairs (512 pixels), 8 accumulator tiles per wave
//   T = 4: MR 1, WM 2, WN 4 -> 64 rows x 128 quads (512 pixels), 6 accumulator tiles per wave
constexpr W1Shape kShape2 = {2, 64, 256, 512, 12}, kShape4 = {4, 64, 128, 512, 18};
const W1Shape &shape_of(int T) { return T == 4 ? kShape4 : kShape2; }

// input pixels spanned by one N tile (+ halo): n groups of T pixels row by row, row and image gaps, 2 rows + T+2 pixels of halo
int w1d_span(const W1Shape &sh, const QpBuf &in, bool cross) {
    const int Hv = in.Hb - 2, Wv = in.Wb - 2, Wg = (Wv + sh.T - 1) / sh.T, n = sh.groups;
    const long rows = (n - 1) / Wg + 2;                 // rows an N tile can touch
    long span = rows * in.Wb;
    if ((long)sh.T * n + in.Wb < span) span = (long)sh.T * n + (long)(in.Wb - sh.T * Wg > 0 ? in.Wb - sh.T * Wg : 0) * rows + in.Wb;
    if (cross) span += (long)(in.Hb * in.Wb - Hv * in.Wb) * ((n - 1) / (Hv * Wg) + 1);
    const long whole = (long)in.B * in.Hb * in.Wb;
    if (cross && span > whole) span = whole;
    return (int)(span + 2 * in.Wb + sh.T + 2);
}
size_t w1d_lds(const W1Shape &sh, int G) { return (size_t)kW1dStages * ((size_t)(sh.mblk / 32) * sh.taps * 1024 + (size_t)2 * G * 1024); }

}  // namespace

// packed layout: [32-row tile][K block][3 * (T+2) planes = ky*(T+2) + xi][lane][4] + bias[mtiles*32]   (as pack.hip)
size_t nd_w1d_packed_floats(int T, int cin, int cout) {
    return (size_t)nd_mtiles(ND_CONV3, cout) * nd_kblocks(cin) * 3 * (T + 2) * 256 + (size_t)nd_mtiles(ND_CONV3, cout) * 32;
}

int nd_w1d_pack(int T, int kind, int cin, int cout, const float *w, const float *bias, float *packed) {
    if (T != 2 && T != 4) ND_FAIL(ND_EINVAL, "w1d: group size must be 2 or 4");
    if (kind != ND_CONV3 && kind != ND_CONVT3) ND_FAIL(ND_EINVAL, "w1d: 3x3 layers only");
    const int MT = nd_mtiles(ND_CONV3, cout), KB = nd_kblocks(cin), NP = T + 2;
    auto tap = [&](int co, int ci, int ky, int kx) -> double {
        if (co >= cout || ci >= cin) return 0.0;
        return kind == ND_CONV3 ? w[(((size_t)co * cin + ci) * 3 + ky) * 3 + kx] : w[(((size_t)ci * cout + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
    };
    for (int mt = 0; mt < MT; ++mt)
        for (int kb = 0; kb < KB; ++kb)
            for (int ky = 0; ky < 3; ++ky)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int co = mt * 32 + (lane & 31), ci = 8 * kb + 4 * (lane >> 5) + e;   // A fragment map of conv_qp (pack.hip)
                        const double g[3] = {tap(co, ci, ky, 0), tap(co, ci, ky, 1), tap(co, ci, ky, 2)};
                        double u[6];
                        if (T == 2) W1<2>::g(g, u); else W1<4>::g(g, u);
                        for (int xi = 0; xi < NP; ++xi)
                            packed[(((size_t)mt * KB + kb) * 3 * NP + ky * NP + xi) * 256 + lane * 4 + e] = (float)u[xi];
                    }
    float *b = packed + (size_t)MT * KB * 3 * NP * 256;
    for (int i = 0; i < MT * 32; ++i) b[i] = (bias && i < cout) ? bias[i] : 0.f;
    return ND_OK;
}

static int w1d_geometry(const W1Shape &sh, const QpBuf &in, bool *cross, size_t *lds) {
    *cross = true;
    int G = (w1d_span(sh, in, true) + 63) / 64;
    if (w1d_lds(sh, G) > 160 * 1024) {
        *cross = false;
        G = (w1d_span(sh, in, false) + 63) / 64;
    }
    *lds = w1d_lds(sh, G);
    return G;
}
// the stage images of a row this wide fit the LDS (cs >= ~400 at full resolution does not for T = 4: fall back to T = 2 / direct)
bool nd_w1d_fits(int T, const QpBuf &in) {
    bool cross;
    size_t lds;
    w1d_geometry(shape_of(T), in, &cross, &lds);
    return (T == 2 || T == 4) && in.dt == ND_F32 && in.Hb >= 3 && in.Wb >= 3 && lds <= 160 * 1024;
}

// d: the layer as for nd_launch_conv (CONV3 / CONVT3, fp32, no pre-activation copy); d.wpk = nd_w1d_pack blob of the same T
int nd_launch_conv_w1d(int T, const ConvDesc &d, hipStream_t stream) {
    if (T != 2 && T != 4) ND_FAIL(ND_EINVAL, "w1d: group size must be 2 or 4");
    if ((d.kind != ND_CONV3 && d.kind != ND_CONVT3) || d.in.dt != ND_F32 || d.out.dt != ND_F32) ND_FAIL(ND_EINVAL, "w1d: fp32 3x3 layers only");
    if (d.pre && d.pre_plane < (long)d.in.B * (d.in.Hb - 2) * (d.in.Wb - 2)) ND_FAIL(ND_EINVAL, "w1d: pre-activation planes too small");
    if (d.cout % 4) ND_FAIL(ND_EINVAL, "w1d: cout must be a multiple of 4");
    const W1Shape &sh = shape_of(T);
    const int KB = nd_kblocks(d.cin);
    if (d.in.planes < d.in_plane0 + 2 * KB) ND_FAIL(ND_EINVAL, "w1d: input buffer has %d planes, needs %d", d.in.planes, d.in_plane0 + 2 * KB);
    const int Hv = d.in.Hb - 2, Wpx = d.in.Wb - 2, Wg = (Wpx + T - 1) / T;
    if (Hv < 1 || Wpx < 1) ND_FAIL(ND_EINVAL, "w1d: input smaller than the kernel");
    if (d.out.Hb != Hv + 2 * d.out.pad || d.out.Wb != Wpx + 2 * d.out.pad || d.out.B != d.in.B) ND_FAIL(ND_EINVAL, "w1d: destination does not fit the result");
    if (d.out_plane0 + d.cout / 4 > d.out.planes) ND_FAIL(ND_EINVAL, "w1d: destination planes overflow");
    if (d.in.used() >= (1L << 31)) ND_FAIL(ND_EINVAL, "w1d: input too large for int32 indexing");

    static std::atomic<int> lds_set[16][2];
    int dev = 0, ncus = 0;
    ND_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) ND_FAIL(ND_EINVAL, "w1d: device index %d", dev);
    ND_TRY(nd_num_cus(dev, &ncus));
    bool cross;
    size_t lds;
    const int G = w1d_geometry(sh, d.in, &cross, &lds);
    if (lds > 160 * 1024) ND_FAIL(ND_EINVAL, "w1d: %zu B of LDS needed (row width %d too large)", lds, d.in.Wb);
    void (*fn)(ConvParams) = T == 4 ? conv_w1d<4, 1, 2, 4, kW1dStages> : conv_w1d<2, 2, 1, 8, kW1dStages>;
    if ((int)lds > lds_set[dev][T == 4].load(std::memory_order_relaxed)) {
        ND_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set[dev][T == 4].store((int)lds, std::memory_order_relaxed);
    }

    ConvParams p = {};
    p.in = (const f32x4 *)d.in.base + (long)d.in_plane0 * d.in.np();
    p.wpk = d.wpk;
    p.bias = d.bias;
    p.out = (f32x4 *)d.out.base;
    p.in_plane = d.in.np();
    p.out_plane = d.out.np();
    p.nimg = d.in.B;
    p.P = d.in.Hb * d.in.Wb;
    p.Wb = d.in.Wb;
    p.Hv = Hv;
    p.Wv = Wg;              // groups per row
    p.PV = Hv * Wg;         // groups per image
    p.wpx = Wpx;            // valid pixels per row
    p.G = G;
    p.stride = 1;
    p.ioff = 0;
    p.pre = (f32x4 *)d.pre;
    p.pre_plane = d.pre_plane;
    p.KB = KB;
    p.M = d.cout;
    p.cout = d.cout;
    p.Po = d.out.Hb * d.out.Wb;
    p.Wo = d.out.Wb;
    p.opad = d.out.pad;
    p.out_plane0 = d.out_plane0;
    p.act = d.act;
    p.slope = d.slope;
    p.slope_dev = d.slope_dev;
    if (cross) {
        p.tpi = 0;
        p.n_tiles_n = (int)(((long)p.nimg * p.PV + sh.groups - 1) / sh.groups);
    } else {
        p.tpi = (p.PV + sh.groups - 1) / sh.groups;
        p.n_tiles_n = p.tpi * p.nimg;
    }
    p.n_tiles_m = (d.cout + sh.mblk - 1) / sh.mblk;
    p.tiles_per_problem = p.n_tiles_n * p.n_tiles_m;
    const long ntiles = p.tiles_per_problem;
    const long slots = ncus;
    const long cap = d.part && !d.nosplit ? (long)(d.part_bytes / ((size_t)sh.mblk * T * sh.groups * 4)) : 0;
    int first, S, cps;
    nd_plan_split(ntiles, KB, slots, cap, &first, &S, &cps);
    p.split_first = first;
    p.S = S;
    p.cps = cps;
    p.nitems = (int)(first + (ntiles - first) * S);
    p.part = (f32x4 *)d.part;
    const long grid = p.nitems < slots ? p.nitems : slots;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(sh.threads), lds, stream, p);
    if (first < ntiles) {
        // the finish kernel works in pixel slots: a tile of n groups is T*n consecutive slots of rows T*Wg slots wide
        // (slots past the real row width are skipped: ConvParams::wpx)
        ConvParams f = p;
        f.Wv = T * Wg;
        f.PV = Hv * T * Wg;
        ND_TRY(nd_launch_split_finish(f, (int)(ntiles - first), sh.mblk, T * sh.groups, 0, ND_F32, stream));
    }
    ND_HIP(hipGetLastError());
    return ND_OK;
}
