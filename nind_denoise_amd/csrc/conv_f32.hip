// fp32 instantiations of the convolution kernel + variant table and launch code for every storage type
#include <vector>

#include "conv_qp.inc"

// ------------------------------------------------------------------ variants and dispatch
extern const Variant g_variants_bf16[kGroup];
extern const Variant g_variants_f16[kGroup];
static const Variant g_f32_group[kGroup] = {ND_VARIANT_GROUP(ND_F32, "f32")};
static const Variant g_f32_extra[] = {
    ND_VARIANT(ND_F32, "f32", 2, 2, 2, 4, 9, 1, false, 3),  // M128 x N256, 8 waves, 3 stages
    ND_VARIANT(ND_F32, "f32", 2, 2, 1, 4, 9, 1, false, 3),  // M64  x N256, 4 waves, 3 stages
    ND_VARIANT(ND_F32, "f32", 2, 4, 4, 2, 1, 2, false, 3),  // 1 tap, M256 x N256 (64x128 per wave): the Winograd GEMMs (Cout % 256 == 0)
    ND_VARIANT(ND_F32, "f32", 2, 2, 2, 4, 1, 2, false, 3),  // 1 tap, M128 x N256
    ND_VARIANT(ND_F32, "f32", 2, 4, 4, 2, 1, 4, false, 2),  // 1 tap, M256 x N256, 4 K blocks per step on 2 stages: 3-6 % faster (any Cin % 32 == 0)
    ND_VARIANT(ND_F32, "f32", 2, 4, 2, 4, 1, 4, false, 2),  // 1 tap, M128 x N512, 4 K blocks per step: 2-4 % over M128 x N256
    ND_VARIANT(ND_F32, "f32", 2, 4, 4, 2, 1, 4, true, 2),   // up (2x2 s2), M256 x N256, 4 K blocks per step: 5-9 % over M128 x N256
};
constexpr int kExtra = (int)(sizeof(g_f32_extra) / sizeof(g_f32_extra[0]));
static const int g_nvariants = 3 * kGroup + kExtra;
// variant index: [f32 group][bf16 group][f16 group][fp32-only experiments]
static const Variant &variant_at(int v) {
    if (v < kGroup) return g_f32_group[v];
    if (v < 2 * kGroup) return g_variants_bf16[v - kGroup];
    if (v < 3 * kGroup) return g_variants_f16[v - 2 * kGroup];
    return g_f32_extra[v - 3 * kGroup];
}


int nd_conv_variant_count() { return g_nvariants; }
// 1-tap fp32 variant for a Winograd GEMM of this shape (measured on the UtNet(64) layers at 256 tiles per pass), -1: automatic
int nd_conv_variant_gemm(int cin, int cout) {
    const int KB = nd_kblocks(cin), x = 3 * kGroup;
    if (KB % 2) return -1;
    // (4 K blocks per step on two stages: re-measured with the 64-position GEMMs of F(6x6): 3 - 6 % faster than 2 blocks on three
    //  stages from Cin = 128 up -- fewer barriers and epilogue-adjacent steps per tile)
    if (cout % 256 == 0) return KB % 4 == 0 ? x + 4 : x + 2;
    if (cout % 128 == 0) return KB % 4 == 0 ? x + 5 : x + 3;
    return -1;
}
const char *nd_conv_variant_label(int v) { return (v >= 0 && v < g_nvariants) ? variant_at(v).name : ""; }

// Largest input span (pixels) of one N tile + 3x3 halo.  cross = tiles may run across image boundaries.
// 3x3 layers consume the border of their input buffer as padding (valid grid = buffer - 2); 1- and 4-tap layers read the
// interior of a possibly bordered buffer
static void valid_grid(int taps, const QpBuf &in, int *Hv, int *Wv, int *stride) {
    *stride = taps == 4 ? 2 : 1;
    *Hv = taps == 9 ? in.Hb - 2 : (in.Hb - 2 * in.pad) / *stride;
    *Wv = taps == 9 ? in.Wb - 2 : (in.Wb - 2 * in.pad) / *stride;
}
// (Hv_o x Wv_o > 0: the valid grid is a region of the buffer's -- same strides, more row / image gaps per tile)
// band2: the pixels of an image are enumerated over 2-row bands (ConvParams::band2) -- a tile of n pixels starting at column x0 of a
// band ends on the lower row of a later band; the stage image is the linear range between its first and last input pixel
static int tile_span(const Variant &V, const QpBuf &in, bool cross, int Hv_o = 0, int Wv_o = 0, bool band2 = false) {
    const int taps = V.taps;
    int Hv, Wv, s;
    valid_grid(taps, in, &Hv, &Wv, &s);
    if (Hv_o > 0) {
        Hv = Hv_o;
        Wv = Wv_o;
    }
    const int n = V.nblk;
    // consecutive valid pixels are s input pixels apart; every row end adds s*(Wb - Wv), every image end the rest of the image
    int span = s * n + s * (in.Wb - Wv) * ((n - 1) / Wv + 1);
    if (band2) {
        span = 0;
        for (int x0 = 0; x0 < Wv; ++x0) {
            const int last = 2 * x0 + n - 1, row1 = 2 * (last / (2 * Wv)) + 1, x1 = (last % (2 * Wv)) >> 1;
            const int sp = row1 * in.Wb + x1 - x0 + 1;
            if (sp > span) span = sp;
        }
    }
    if (cross) span += (in.Hb * in.Wb - Hv * s * in.Wb) * ((n - 1) / (Hv * Wv) + 1);
    const long whole = (long)in.B * in.Hb * in.Wb;   // a tile never reads past the pixels in use (tiny images: 3x3 at the bottom)
    if (cross && span > whole) span = (int)whole;
    span += taps == 9 ? 2 * in.Wb + 2 : (taps == 4 ? in.Wb + 1 : 0);
    return span;
}
static size_t lds_for(const Variant &V, int G) {
    return (size_t)V.nstage * ((size_t)(V.mblk / 32) * V.kbc * V.taps * 1024 + (size_t)2 * V.kbc * G * 1024);
}
// tiles may cross images when that still fits the LDS (small images: no padding of every image to a tile multiple)
static size_t variant_lds(const Variant &V, const QpBuf &in, bool *cross_out = nullptr, int *G_out = nullptr, int Hv_o = 0, int Wv_o = 0,
                          bool band2 = false) {
    const size_t kMax = 160 * 1024;
    int G = (tile_span(V, in, true, Hv_o, Wv_o, band2) + 63) / 64;
    bool cross = true;
    if (lds_for(V, G) > kMax) {
        cross = false;
        G = (tile_span(V, in, false, Hv_o, Wv_o, band2) + 63) / 64;
    }
    if (cross_out) *cross_out = cross;
    if (G_out) *G_out = G;
    return lds_for(V, G);
}

static const size_t kMaxLds = 160 * 1024;

// ------------------------------------------------------------------ split-K tail
// Persistent workgroups finish whole rounds of tiles at full rate, but the last, partial round leaves CUs idle (and a layer
// with fewer tiles than CUs -- the deep levels of a small batch -- is nothing but a partial round).  The tiles of that
// round are therefore cut along K into S slices that fill the idle CUs; slices store raw accumulators and this kernel adds
// them in slice order (deterministic), then applies bias / activation exactly like the conv epilogue.
// grid: (split tiles, MBLK/4 channel quads)
__global__ __launch_bounds__(256) void k_split_finish(ConvParams p, int mblk, int nblk, int up, int dt) {
    const int t = blockIdx.x, quad = blockIdx.y;
    const int gid = p.split_first + t;
    const int z = gid / p.tiles_per_problem, id = gid - z * p.tiles_per_problem;
    const int nb = id / p.n_tiles_m, mb = id - nb * p.n_tiles_m;
    const int m4 = mb * mblk + quad * 4;
    if (m4 >= p.M) return;
    p.out += z * p.out_bs;
    const float slope = p.act == ND_ACT_NONE ? 1.f : (p.slope_dev ? *p.slope_dev : p.slope);
    const f32x4 bv = *(const f32x4 *)(p.bias + m4);
    const int nq = mblk / 4;
    for (int l = threadIdx.x; l < nblk; l += 256) {
        int bi, r;
        if (p.tpi) {
            bi = nb / p.tpi;
            r = (nb - bi * p.tpi) * nblk + l;
            if (r >= p.PV) continue;
        } else {
            const long g = (long)nb * nblk + l;
            if (g >= (long)p.nimg * p.PV) continue;
            bi = (int)(g / p.PV);
            r = (int)(g - (long)bi * p.PV);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < p.S; ++ks) acc += p.part[((size_t)(t * p.S + ks) * nq + quad) * nblk + l];
        acc += bv;
        if (p.pre) {
            const long pr = p.wpx ? ((long)bi * p.Hv + r / p.Wv) * p.wpx + (r - (r / p.Wv) * p.Wv) : (long)bi * p.PV + r;
            if (!p.wpx || r - (r / p.Wv) * p.Wv < p.wpx) p.pre[(long)(m4 >> 2) * p.pre_plane + pr] = acc;
        }
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p.act <= ND_ACT_PRELU ? (acc[e] > 0.f ? acc[e] : acc[e] * slope) : apply_act(acc[e], p.act, slope);
        const int y = r / p.Wv, x = r - y * p.Wv;
        if (p.wpx && x >= p.wpx) continue;
        int co = m4;
        long pix;
        if (up) {
            const NdUpRow ur = nd_up_row(m4, p.cout, dt);
            co = ur.co;
            pix = (long)bi * p.Po + (long)(2 * y + p.opad + ur.a) * p.Wo + (2 * x + p.opad + ur.b);
        } else {
            pix = (long)bi * p.Po + (long)(y + p.opad) * p.Wo + (x + p.opad);
        }
        if (dt == ND_F32) {
            p.out[(long)(p.out_plane0 + (co >> 2)) * p.out_plane + pix] = v;
        } else {
            char *dst = (char *)p.out + (((long)(p.out_plane0 + (co >> 3)) * p.out_plane + pix) << 4) + ((co >> 2) & 1) * 8;
            if (dt == ND_BF16) {
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

// Time of a launch in units of "one workgroup runs one K chunk": `slots` workgroups run concurrently, `ntiles` tiles of
// `nchunks` chunks each.  Picks the split of the partial round (S slices of cps chunks) that minimises it; kOver chunks
// of fixed cost per work item (pipeline prologue, epilogue / partial store + its share of k_split_finish).
struct SplitPlan { int first, S, cps; double time; };
static SplitPlan plan_split(long ntiles, int nchunks, long slots, long max_items, double over) {
    SplitPlan b;
    const long full = ntiles / slots * slots, R = ntiles - full;
    b.first = (int)ntiles;
    b.S = 1;
    b.cps = nchunks;
    b.time = (double)((ntiles + slots - 1) / slots) * (nchunks + over);
    if (R == 0 || max_items <= 0) return b;
    const double base = (double)(full / slots) * (nchunks + over);
    for (int S = 2; S <= nchunks && S <= 64; ++S) {
        const int cps = (nchunks + S - 1) / S, Se = (nchunks + cps - 1) / cps;
        if (R * Se > max_items) break;
        const double t = base + (double)((R * Se + slots - 1) / slots) * (cps + 1.5 * over);
        if (t < 0.93 * b.time) {
            b.first = (int)full;
            b.S = Se;
            b.cps = cps;
            b.time = t;
        }
    }
    return b;
}

static std::atomic<int> g_num_cus{0};   // CU count of the device in use (every GPU of a node is the same part)
static const double kSplitOver = 3.0;

// max_items = 0 (no scratch, or ND_FLAG_NO_SPLITK on the call): never split
void nd_plan_split(long ntiles, int nchunks, long slots, long max_items, int *first, int *S, int *cps) {
    const SplitPlan sp = plan_split(ntiles, nchunks, slots, max_items, kSplitOver);
    *first = sp.first;
    *S = sp.S;
    *cps = sp.cps;
}
int nd_launch_split_finish(const ConvParams &p, int n_split_tiles, int mblk, int nblk, int up, int dt, hipStream_t s) {
    hipLaunchKernelGGL(k_split_finish, dim3((unsigned)n_split_tiles, mblk / 4), dim3(256), 0, s, p, mblk, nblk, up, dt);
    ND_HIP(hipGetLastError());
    return ND_OK;
}

static int pick_variant(const ConvDesc &d, int M) {
    // (a region of interest shortens the rows of valid pixels: more row gaps per tile, a longer LDS halo image)
    const int Hr = d.roi_rows > 0 ? d.roi_rows : 0, Wr = d.roi_rows > 0 ? d.roi_cols : 0;
    const bool band2 = d.pool != nullptr;   // fused pool: 2-row band enumeration, a longer stage image
    const int taps = nd_taps(d.kind);
    const bool up = d.kind == ND_CONVT2S2;
    const int dt = d.in.dt;
    const int KB = nd_kblocks(d.cin, dt);
    const int g0 = dt * kGroup;
    if (taps == 9) {
        if (M > 32) {
            // two shapes compete: 64x512 tiles on three LDS stages and 64x1024 tiles (64x128 per wave: a third less LDS-DMA
            // per FLOP, measured ~4.5 % cheaper per pixel) on two.  The bigger tile loses when the tile count quantises badly
            // against the CU count, so compare whole rounds.
            int Hv, Wv, st_;
            valid_grid(9, d.in, &Hv, &Wv, &st_);
            const int cus = g_num_cus.load(std::memory_order_relaxed) > 0 ? g_num_cus.load(std::memory_order_relaxed) : 256;
            double best = 0;
            int best_v = -1;
            // (16-bit units re-measured in round 3, after the 16-byte epilogue stores, on every 3x3 layer of UtNet(64) at 160 tiles of 264:
            //  the 64 x 1024 shape is now 2 - 4 % ahead of 128 x 512 from 128 channels up -- the wider tile's advantage was its
            //  cheaper store epilogue -- and 5 - 10 % ahead of 64 x 512; the 128 x 512 shape keeps the 13-pixel bottom, where its
            //  tiles quantise better: tools/sweep_variants.sh)
            const struct { int v; double unit; } cand[] = {{0, dt == ND_F32 ? 1.0 : 0.92}, {11, dt == ND_F32 ? 0.955 : 0.85}, {14, dt == ND_F32 ? 2.0 : 1.75}};
            // (the cost below is rounds x chunks x N-tile pixels x unit: a 128 x 512 tile does the work of a 64 x 1024 one, so equal
            //  cost per tile is unit(14) = 2 x unit(11); 1.75 / 0.85 = 2.06 says "3 % slower per tile")
            for (const auto &c : cand) {
                const Variant &V = variant_at(g0 + c.v);
                if (c.v == 14 && (M < 128 || dt == ND_F32)) continue;
                bool cross = true;
                if (variant_lds(V, d.in, &cross, nullptr, Hr, Wr, band2) > kMaxLds) continue;
                const long pv = (long)Hv * Wv;
                const long tn = cross ? ((long)d.in.B * pv + V.nblk - 1) / V.nblk : ((pv + V.nblk - 1) / V.nblk) * d.in.B;
                const long tiles = tn * ((M + V.mblk - 1) / V.mblk);
                const long cap = d.part && !d.nosplit && !band2 ? (long)(d.part_bytes / ((size_t)V.mblk * V.nblk * 4)) : 0;
                const double cost = plan_split(tiles, KB / V.kbc, cus, cap, kSplitOver).time * V.nblk * c.unit;
                if (best_v < 0 || cost < best) {
                    best = cost;
                    best_v = c.v;
                }
            }
            if (best_v >= 0) return g0 + best_v;
        }
        const int order[] = {M <= 32 ? 3 : 0, 0, 1, 2};
        for (int v : order)
            if (variant_lds(variant_at(g0 + v), d.in, nullptr, nullptr, Hr, Wr, band2) <= kMaxLds) return g0 + v;
        return g0 + 2;
    }
    if (taps == 4) return g0 + (variant_lds(variant_at(g0 + 12), d.in, nullptr, nullptr, Hr, Wr) <= kMaxLds ? 12 : 13);
    if (KB % 2) return g0 + (up ? 7 : 5);
    if (up) {
        // preferred shape first, then the ones with shorter K chunks / more stages whose stage images still fit the LDS
        int cand[8], n = 0;
        if (dt == ND_F32 && M % 256 == 0 && KB % 4 == 0) cand[n++] = 3 * kGroup + 6;
        if (M >= 128) {
            if (KB % 4 == 0) cand[n++] = g0 + 10;
            if (KB % 2 == 0) cand[n++] = g0 + 8;
        }
        if (KB % 4 == 0) cand[n++] = g0 + 9;
        if (KB % 2 == 0) cand[n++] = g0 + 6;
        cand[n++] = g0 + 7;
        for (int i = 0; i < n; ++i)
            if (variant_lds(variant_at(cand[i]), d.in, nullptr, nullptr, Hr, Wr) <= kMaxLds) return cand[i];
        return cand[n - 1];
    }
    return g0 + 4;
}

// can this layer write its 2x2-pooled tensor itself (conv_qp, 16-bit storage)?  Needs even output sizes and a workgroup shape whose
// stage images hold the longer pixel range of the 2-row band enumeration
bool nd_conv_pool_fits(const ConvDesc &d) {
    if (!d.pool || d.in.dt == ND_F32 || nd_taps(d.kind) != 9 || d.roi_rows > 0 || d.pre || d.nbatch > 1) return false;
    if (((d.in.Hb - 2) | (d.in.Wb - 2)) & 1) return false;
    // measured at 160 tiles of 264 (bf16; same box, flags 16 against 0): conv + pool kernel 1.240 -> 1.05 ms on the 264-pixel level,
    // 0.842 -> 0.78 on the 128-pixel level, 0.621 -> 0.609 and 0.479 -> 0.480 below: the quad maxima, second rounding and pooled
    // stores cost the epilogue about what the separate kernel costs once the tensor is small -- fuse the large levels only
    if ((long)(d.in.Hb - 2) * (d.in.Wb - 2) < 100L * 100) return false;
    const int v = pick_variant(d, d.cout);
    if (v < 0 || v >= g_nvariants) return false;
    return variant_lds(variant_at(v), d.in, nullptr, nullptr, 0, 0, true) <= kMaxLds;
}

// does a layer restricted to the region d.roi_* have a workgroup shape whose LDS stage images fit?  (a region much narrower than the
// buffer's rows puts many row gaps into the linear pixel range of an N tile)
bool nd_conv_roi_fits(const ConvDesc &d) {
    const bool up = d.kind == ND_CONVT2S2;
    const int v = pick_variant(d, up ? 4 * d.cout : d.cout);
    if (v < 0 || v >= g_nvariants) return false;
    return variant_lds(variant_at(v), d.in, nullptr, nullptr, d.roi_rows > 0 ? d.roi_rows : 0, d.roi_rows > 0 ? d.roi_cols : 0) <= kMaxLds;
}

static std::atomic<int> g_lds_set[16][64];   // per device: function attributes belong to the device's copy of the code object

int nd_launch_conv(const ConvDesc &d, hipStream_t stream) {
    const int taps = nd_taps(d.kind);
    const bool up = d.kind == ND_CONVT2S2;
    const int dt = d.in.dt;
    if (dt < ND_F32 || dt > ND_F16 || d.out.dt != dt) ND_FAIL(ND_EINVAL, "conv: input / output storage types %d / %d", d.in.dt, d.out.dt);
    const int KB = nd_kblocks(d.cin, dt);
    const int M = up ? 4 * d.cout : d.cout;
    if (d.cout % nd_cpp(dt)) ND_FAIL(ND_EINVAL, "conv: cout=%d must be a multiple of %d", d.cout, nd_cpp(dt));
    if (d.in.planes < d.in_plane0 + 2 * KB) ND_FAIL(ND_EINVAL, "conv: input buffer has %d planes, needs %d", d.in.planes, d.in_plane0 + 2 * KB);
    const long NP = d.in.used();
    if (NP >= (1L << 31)) ND_FAIL(ND_EINVAL, "conv: %ld linear pixels exceed the int32 index range", NP);

    int dev = 0, ncus = 0;
    ND_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) ND_FAIL(ND_EINVAL, "conv: device index %d", dev);
    ND_TRY(nd_num_cus(dev, &ncus));
    g_num_cus.store(ncus, std::memory_order_relaxed);
    int v = d.variant >= 0 ? d.variant : pick_variant(d, M);
    if (v < 0 || v >= g_nvariants) ND_FAIL(ND_EINVAL, "conv: unknown variant %d", v);
    const Variant &V = variant_at(v);
    if (V.taps != taps || V.up != up || V.dt != dt) ND_FAIL(ND_EINVAL, "conv: variant %s does not match layer kind %d / dtype %d", V.name, d.kind, dt);
    if (KB % V.kbc) ND_FAIL(ND_EINVAL, "conv: Cin/8=%d not a multiple of the variant's K chunk %d", KB, V.kbc);

    ConvParams p = {};
    p.in = (const f32x4 *)d.in.base + (long)d.in_plane0 * d.in.np();
    p.wpk = d.wpk;
    p.bias = d.bias;
    p.out = (f32x4 *)d.out.base + (d.roi_rows > 0 ? (long)((up ? 2 : 1) * d.roi_r0) * d.out.Wb + (up ? 2 : 1) * d.roi_c0 : 0);
    p.in_plane = d.in.np();
    p.out_plane = d.out.np();
    p.nimg = d.in.B;
    p.P = d.in.Hb * d.in.Wb;
    p.Wb = d.in.Wb;
    valid_grid(taps, d.in, &p.Hv, &p.Wv, &p.stride);
    const bool roi = d.roi_rows > 0;
    if (roi) {
        // region of the valid grid (of the INPUT grid for a 2x2 stride-2 transpose): same launch, shifted first pixel, smaller valid
        // extents (the kernel takes the row / image strides from the buffer and the extents from Hv / Wv)
        if (taps == 4 || d.roi_r0 < 0 || d.roi_c0 < 0 || d.roi_cols < 1 || d.roi_r0 + d.roi_rows > p.Hv || d.roi_c0 + d.roi_cols > p.Wv || d.nbatch > 1 || d.pre)
            ND_FAIL(ND_EINVAL, "conv: region [%d,+%d) x [%d,+%d) outside the %d x %d grid (or a batched / training launch)", d.roi_r0, d.roi_rows, d.roi_c0, d.roi_cols, p.Hv, p.Wv);
        p.Hv = d.roi_rows;
        p.Wv = d.roi_cols;
    }
    p.PV = p.Hv * p.Wv;
    if (taps == 4 && ((d.in.Hb | d.in.Wb) & 1)) ND_FAIL(ND_EINVAL, "conv: the stride-2 layer reads even-sized buffers only");
    p.ioff = (taps == 9 ? 0 : d.in.pad * d.in.Wb + d.in.pad) + (roi ? d.roi_r0 * d.in.Wb + d.roi_c0 : 0);
    p.pre = (f32x4 *)d.pre;
    p.pre_plane = d.pre_plane;
    if (d.pre && (dt != ND_F32 || up)) ND_FAIL(ND_EINVAL, "conv: the pre-activation copy exists for fp32 non-upsampling layers only");
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
    // (a 2x2 stride-2 result may be smaller than its destination: UNet's F.pad fix-up for odd sizes, ThirdPartyNets.py:110-118)
    const bool fits = (up || roi) ? (d.out.Hb >= oh + 2 * d.out.pad && d.out.Wb >= ow + 2 * d.out.pad)
                                  : (d.out.Hb == oh + 2 * d.out.pad && d.out.Wb == ow + 2 * d.out.pad);
    if (!fits || d.out.B != d.in.B)
        ND_FAIL(ND_EINVAL, "conv: destination %dx%dx%d(pad %d) does not fit result %dx%dx%d", d.out.B, d.out.Hb, d.out.Wb,
                d.out.pad, d.in.B, oh, ow);
    if (d.out_plane0 + d.cout / nd_cpp(dt) > d.out.planes) ND_FAIL(ND_EINVAL, "conv: destination planes overflow");

    // fused MaxPool2d(2) (16-bit storage): 2-row band enumeration of the pixels, pooled tensor written from the epilogue
    if (d.pool) {
        const QpBuf &q = *d.pool;
        if (dt == ND_F32 || taps != 9 || up || roi || d.pre || d.nbatch > 1 || (p.Hv & 1) || (p.Wv & 1))
            ND_FAIL(ND_EINVAL, "conv: a fused pool needs a whole 16-bit 3x3 layer with even output sizes (got %d x %d, dtype %d)", p.Hv, p.Wv, dt);
        if (q.dt != dt || q.B != d.in.B || q.Hb - 2 * q.pad != p.Hv / 2 || q.Wb - 2 * q.pad != p.Wv / 2 || q.planes < d.cout / nd_cpp(dt))
            ND_FAIL(ND_EINVAL, "conv: pooled destination does not fit %d x %d x %d", d.cout, p.Hv / 2, p.Wv / 2);
        p.pool = (f32x4 *)q.base;
        p.pool_plane = q.np();
        p.pool_P = q.Hb * q.Wb;
        p.pool_W = q.Wb;
        p.pool_pad = q.pad;
        p.band2 = 1;
    }
    bool cross = true;
    const size_t lds = variant_lds(V, d.in, &cross, &p.G, roi ? p.Hv : 0, roi ? p.Wv : 0, p.band2 != 0);
    if (lds > kMaxLds) ND_FAIL(ND_EINVAL, "conv: %zu B of LDS needed (row width %d too large for variant %s)", lds, p.Wb, V.name);
    if ((int)lds > g_lds_set[dev][v].load(std::memory_order_relaxed)) {
        ND_HIP(hipFuncSetAttribute((const void *)V.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        g_lds_set[dev][v].store((int)lds, std::memory_order_relaxed);
    }

    if (cross) {
        p.tpi = 0;
        p.n_tiles_n = (int)(((long)p.nimg * p.PV + V.nblk - 1) / V.nblk);
    } else {
        p.tpi = (p.PV + V.nblk - 1) / V.nblk;
        p.n_tiles_n = p.tpi * p.nimg;
    }
    p.n_tiles_m = (M + V.mblk - 1) / V.mblk;
    p.tiles_per_problem = p.n_tiles_n * p.n_tiles_m;
    p.wpx = 0;
    // gap-free 1-tap launch with raw output (the Winograd GEMMs): see ConvParams::linear
    p.linear = taps == 1 && !up && dt == ND_F32 && d.in.pad == 0 && d.out.pad == 0 && d.out.Hb == d.in.Hb && d.out.Wb == d.in.Wb && cross &&
               d.act == ND_ACT_NONE && d.nbatch > 1 && !d.pre && !roi;
    p.in_bs = d.in_bs;
    p.out_bs = d.out_bs;
    p.w_bs = (long)d.w_bs;
    if (d.nbatch > 1 && d.pre) ND_FAIL(ND_EINVAL, "conv: a batched launch keeps no pre-activation copy");
    nd_conv_fastdivs(p);
    if (dt != ND_F32) p.fd_upq = nd_fastdiv(p.cout / 8 > 0 ? p.cout / 8 : 1);   // 8 channels per plane element
    const long ntiles = (long)p.tiles_per_problem * (d.nbatch > 1 ? d.nbatch : 1);
    const int per_cu = lds * 2 <= kMaxLds && V.threads <= 256 ? 2 : 1;
    const long slots = (long)ncus * per_cu;
    // (a layer that pools keeps its tiles whole: the split-K finish kernel sees no neighbours)
    const long cap = d.part && !d.nosplit && !p.band2 ? (long)(d.part_bytes / ((size_t)V.mblk * V.nblk * 4)) : 0;
    const SplitPlan sp = plan_split(ntiles, KB / V.kbc, slots, cap, kSplitOver);
    p.split_first = sp.first;
    p.S = sp.S;
    p.cps = sp.cps;
    p.nitems = (int)(sp.first + (ntiles - sp.first) * sp.S);
    p.part = (f32x4 *)d.part;
    const long grid = p.nitems < slots ? p.nitems : slots;
#ifdef ND_QP_ABLATE
    static const int abl_env = getenv("ND_QP_ABL") ? atoi(getenv("ND_QP_ABL")) : 0;   // make ABLATE=1: see fill_slice / epilogue
    p.dbg = abl_env;
#endif
#ifdef ND_QP_STAMPS
    static const int dbg_env = getenv("ND_QP_DBG") ? atoi(getenv("ND_QP_DBG")) : 0;
    if (dbg_env & 128) {
        // stamped diagnostic launch: no split-K (p.part carries the stamp buffer), synchronous, prints the phase split of two waves
        static unsigned long long *buf = nullptr;
        const size_t n = (size_t)slots * 8 * 8;
        if (!buf) ND_HIP(hipMalloc(&buf, n * 8));
        ND_HIP(hipMemsetAsync(buf, 0, n * 8, stream));
        p.dbg = dbg_env;
        p.split_first = (int)ntiles; p.S = 1; p.cps = KB / V.kbc; p.nitems = (int)ntiles; p.part = (f32x4 *)buf;
        const long g2 = ntiles < slots ? ntiles : slots;
        hipLaunchKernelGGL(V.fn, dim3((unsigned)g2), dim3(V.threads), lds, stream, p);
        ND_HIP(hipStreamSynchronize(stream));
        static int printed = 0;
        if (printed++ < 40) {
            std::vector<unsigned long long> h(n);
            ND_HIP(hipMemcpy(h.data(), buf, n * 8, hipMemcpyDeviceToHost));
            const char *names[5] = {"barrier", "issue DMA", "MFMA loop", "wait vmcnt", "epilogue+next"};
            fprintf(stderr, "[qp stamps] %s cin %d M %d tiles %ld chunks %d G %d\n", V.name, d.cin, M, ntiles, KB / V.kbc, p.G);
            for (int w : {0, V.threads / 64 - 1}) {
                double tot[5] = {0}, steps = 0;
                for (long b = 0; b < g2; ++b) {
                    for (int k = 0; k < 5; ++k) tot[k] += (double)h[((size_t)b * 8 + w) * 8 + k];
                    steps += (double)h[((size_t)b * 8 + w) * 8 + 6];
                }
                double sum = 0;
                for (int k = 0; k < 5; ++k) sum += tot[k];
                fprintf(stderr, "   wave %d: %.0f cycles/step:", w, sum / steps);
                for (int k = 0; k < 5; ++k) fprintf(stderr, "  %s %.0f (%.1f%%)", names[k], tot[k] / steps, 100 * tot[k] / sum);
                fprintf(stderr, "\n");
            }
        }
        return ND_OK;
    }
#endif
    hipLaunchKernelGGL(V.fn, dim3((unsigned)grid), dim3(V.threads), lds, stream, p);
    if (sp.first < ntiles)
        hipLaunchKernelGGL(k_split_finish, dim3((unsigned)(ntiles - sp.first), V.mblk / 4), dim3(256), 0, stream, p, V.mblk,
                           V.nblk, up ? 1 : 0, dt);
    ND_HIP(hipGetLastError());
    return ND_OK;
}
