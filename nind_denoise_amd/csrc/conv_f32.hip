// fp32 instantiations of the convolution kernel + variant table and launch code for every storage type
#include "conv_qp.inc"

// ------------------------------------------------------------------ variants and dispatch
extern const Variant g_variants_bf16[kGroup];
extern const Variant g_variants_f16[kGroup];
static const Variant g_f32_group[kGroup] = {ND_VARIANT_GROUP(ND_F32, "f32")};
static const Variant g_f32_extra[] = {
    ND_VARIANT(ND_F32, "f32", 2, 2, 2, 4, 9, 1, false, 3),  // M128 x N256, 8 waves, 3 stages
    ND_VARIANT(ND_F32, "f32", 2, 2, 1, 4, 9, 1, false, 3),  // M64  x N256, 4 waves, 3 stages
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
const char *nd_conv_variant_label(int v) { return (v >= 0 && v < g_nvariants) ? variant_at(v).name : ""; }

// Largest input span (pixels) of one N tile + 3x3 halo.  cross = tiles may run across image boundaries.
// 3x3 layers consume the border of their input buffer as padding (valid grid = buffer - 2); 1- and 4-tap layers read the
// interior of a possibly bordered buffer
static void valid_grid(int taps, const QpBuf &in, int *Hv, int *Wv, int *stride) {
    *stride = taps == 4 ? 2 : 1;
    *Hv = taps == 9 ? in.Hb - 2 : (in.Hb - 2 * in.pad) / *stride;
    *Wv = taps == 9 ? in.Wb - 2 : (in.Wb - 2 * in.pad) / *stride;
}
static int tile_span(const Variant &V, const QpBuf &in, bool cross) {
    const int taps = V.taps;
    int Hv, Wv, s;
    valid_grid(taps, in, &Hv, &Wv, &s);
    const int n = V.nblk;
    // consecutive valid pixels are s input pixels apart; every row end adds s*(Wb - Wv), every image end the rest of the image
    int span = s * n + s * (in.Wb - Wv) * ((n - 1) / Wv + 1);
    if (cross) span += (in.Hb * in.Wb - Hv * s * in.Wb) * ((n - 1) / (Hv * Wv) + 1);
    span += taps == 9 ? 2 * in.Wb + 2 : (taps == 4 ? in.Wb + 1 : 0);
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

static int g_num_cus = 0;

static int pick_variant(const ConvDesc &d, int M) {
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
            const int cus = g_num_cus > 0 ? g_num_cus : 256;
            double best = 0;
            int best_v = -1;
            const struct { int v; double unit; } cand[] = {{0, 1.0}, {11, 0.955}};
            for (const auto &c : cand) {
                const Variant &V = variant_at(g0 + c.v);
                bool cross = true;
                if (variant_lds(V, d.in, &cross) > kMaxLds) continue;
                const long pv = (long)Hv * Wv;
                const long tn = cross ? ((long)d.in.B * pv + V.nblk - 1) / V.nblk : ((pv + V.nblk - 1) / V.nblk) * d.in.B;
                const long tiles = tn * ((M + V.mblk - 1) / V.mblk);
                const double cost = (double)((tiles + cus - 1) / cus) * V.nblk * c.unit;
                if (best_v < 0 || cost < best) {
                    best = cost;
                    best_v = c.v;
                }
            }
            if (best_v >= 0) return g0 + best_v;
        }
        const int order[] = {M <= 32 ? 3 : 0, 0, 1, 2};
        for (int v : order)
            if (variant_lds(variant_at(g0 + v), d.in) <= kMaxLds) return g0 + v;
        return g0 + 2;
    }
    if (taps == 4) return g0 + (variant_lds(variant_at(g0 + 12), d.in) <= kMaxLds ? 12 : 13);
    if (KB % 2) return g0 + (up ? 7 : 5);
    if (up) return g0 + (M >= 128 ? (KB % 4 == 0 ? 10 : 8) : (KB % 4 == 0 ? 9 : 6));
    return g0 + 4;
}

static int g_lds_set[64] = {0};

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

    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        ND_HIP(hipGetDevice(&dev));
        ND_HIP(hipGetDeviceProperties(&prop, dev));
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    int v = d.variant >= 0 ? d.variant : pick_variant(d, M);
    if (v < 0 || v >= g_nvariants) ND_FAIL(ND_EINVAL, "conv: unknown variant %d", v);
    const Variant &V = variant_at(v);
    if (V.taps != taps || V.up != up || V.dt != dt) ND_FAIL(ND_EINVAL, "conv: variant %s does not match layer kind %d / dtype %d", V.name, d.kind, dt);
    if (KB % V.kbc) ND_FAIL(ND_EINVAL, "conv: Cin/8=%d not a multiple of the variant's K chunk %d", KB, V.kbc);

    ConvParams p;
    p.in = (const f32x4 *)d.in.base + (long)d.in_plane0 * d.in.np();
    p.wpk = d.wpk;
    p.bias = d.bias;
    p.out = (f32x4 *)d.out.base;
    p.in_plane = d.in.np();
    p.out_plane = d.out.np();
    p.nimg = d.in.B;
    p.P = d.in.Hb * d.in.Wb;
    p.Wb = d.in.Wb;
    valid_grid(taps, d.in, &p.Hv, &p.Wv, &p.stride);
    p.PV = p.Hv * p.Wv;
    if (taps == 4 && ((d.in.Hb | d.in.Wb) & 1)) ND_FAIL(ND_EINVAL, "conv: the stride-2 layer reads even-sized buffers only");
    p.ioff = taps == 9 ? 0 : d.in.pad * d.in.Wb + d.in.pad;
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
