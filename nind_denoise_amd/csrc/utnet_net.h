// UtNet description shared by the inference executor (utnet.hip) and the training step (utnet_train.hip):
// layer table, state-dict tensor order, packed-blob layout, activation-buffer plan and the forward launch sequence.
// Everything lives in an unnamed namespace: each translation unit gets its own copy.
#pragma once
#include <string.h>

#include <string>
#include <vector>

#include "nd_common.h"

namespace {

struct LayerSpec {
    const char *key;
    int kind;
    int cin_mul, cout_mul;  // channels = mul * funit (cin_mul 0 -> 3 input channels, cout_mul 0 -> 3 output channels)
    int prelu;              // index into the slope table, -1: no activation
};

// forward order; prelu indices follow the state-dict order of the activation modules
const LayerSpec kLayers[] = {
    {"convs1.0", ND_CONV3, 0, 1, 0},    {"convs1.2", ND_CONV3, 1, 1, 1},    {"convs2.0", ND_CONV3, 1, 2, 2},
    {"convs2.2", ND_CONV3, 2, 2, 3},    {"convs3.0", ND_CONV3, 2, 4, 4},    {"convs3.2", ND_CONV3, 4, 4, 5},
    {"convs4.0", ND_CONV3, 4, 8, 6},    {"convs4.2", ND_CONV3, 8, 8, 7},    {"bottom.0", ND_CONV3, 8, 16, 8},
    {"bottom.2", ND_CONVT3, 16, 16, 9}, {"up1", ND_CONVT2S2, 16, 8, -1},    {"tconvs1.0", ND_CONVT3, 16, 8, 10},
    {"tconvs1.2", ND_CONVT3, 8, 8, 11}, {"up2", ND_CONVT2S2, 8, 4, -1},     {"tconvs2.0", ND_CONVT3, 8, 4, 12},
    {"tconvs2.2", ND_CONVT3, 4, 4, 13}, {"up3", ND_CONVT2S2, 4, 2, -1},     {"tconvs3.0", ND_CONVT3, 4, 2, 14},
    {"tconvs3.2", ND_CONVT3, 2, 2, 15}, {"up4", ND_CONVT2S2, 2, 1, -1},     {"tconvs4.0", ND_CONVT3, 2, 1, 16},
    {"tconvs4.2", ND_CONVT3, 1, 1, 17}, {"tconvs4.4", ND_CONV1, 1, 0, -1},
};
constexpr int kNumLayers = (int)(sizeof(kLayers) / sizeof(kLayers[0]));
constexpr int kNumSlopes = 18;
constexpr int kHeaderFloats = 32;  // slope table (18 used)

// state-dict order of the reference module (UtNet.py:27-88): weight, bias of every layer, PReLU weights interleaved
std::vector<std::string> build_tensor_names() {
    std::vector<std::string> n;
    auto seq = [&](const std::string &p, int n_act_pairs, bool final1x1) {
        for (int k = 0; k < n_act_pairs; ++k) {
            n.push_back(p + "." + std::to_string(2 * k) + ".weight");
            n.push_back(p + "." + std::to_string(2 * k) + ".bias");
            n.push_back(p + "." + std::to_string(2 * k + 1) + ".weight");
        }
        if (final1x1) {
            n.push_back(p + ".4.weight");
            n.push_back(p + ".4.bias");
        }
    };
    for (int i = 1; i <= 4; ++i) seq("convs" + std::to_string(i), 2, false);
    seq("bottom", 2, false);
    for (int i = 1; i <= 4; ++i) {
        n.push_back("up" + std::to_string(i) + ".weight");
        n.push_back("up" + std::to_string(i) + ".bias");
        seq("tconvs" + std::to_string(i), 2, i == 4);
    }
    return n;
}
const std::vector<std::string> &tensor_names() {
    static const std::vector<std::string> n = build_tensor_names();
    return n;
}
int tensor_index(const std::string &name) {
    const auto &n = tensor_names();
    for (size_t i = 0; i < n.size(); ++i)
        if (n[i] == name) return (int)i;
    return -1;
}

inline int lcin(const LayerSpec &l, int f) { return l.cin_mul ? l.cin_mul * f : 3; }
inline int lcout(const LayerSpec &l, int f) { return l.cout_mul ? l.cout_mul * f : 3; }

// fp32 inference form of the 3x3 layers, chosen from measurements on the UtNet(64) shapes at 256 tiles per launch:
//   Cin * Cout >= 128 * 256 : three-pass Winograd F(6x6, 3x3) (winograd.hip; F(4x4) until late round 2): 1.65x (128 -> 256) ... 2.6x (1024 -> 512) the direct kernel
//   below                   : 1-D Winograd F(4, 3) along x inside the implicit-GEMM kernel (conv_w1d.hip): 1.43 - 1.5x the direct
//                             kernel; the three-pass form is HBM-bound on its transform passes there (64 -> 64: 0.87x, 128 -> 128: 1.4x)
constexpr int kWinoTile = 6;   // F(6x6,3x3): 64 MACs per 36 outputs and 1.78x |X| of transform traffic (F(4x4): 36 per 16, 2.25x); every
                             // three-pass layer of UtNet(64) measured 1 - 27 % faster than with F(4x4) at cs = 264 (bottom.2, a 13x13 output, the least)
constexpr int kW1dTile = 4;
constexpr int kWinoChunk = 256;  // images per three-pass Winograd pass: bounds the V / M scratch (60 MB per 264-pixel tile for the
                                 // largest layer); G24 at 256 tiles per launch: 46.1 MP/s with 64, 47.4 with 128, 47.8 with 256
inline bool wino_layer(const LayerSpec &l, int f, int dt) {
    return dt == ND_F32 && (l.kind == ND_CONV3 || l.kind == ND_CONVT3) && l.cin_mul * f >= 128 && l.cout_mul * f >= 128 &&
           ((long)l.cin_mul * f * l.cout_mul * f >= 128L * 256 || l.kind == ND_CONVT3) && (l.cin_mul * f) % 16 == 0;
    // (128 -> 128: the transposed layer tconvs3.2 is 0.47 ms faster in the three-pass form at 256 tiles of 264, the valid layer
    //  convs2.2 -- which also writes its pooled tensor -- 0.16 ms slower)
}

// float offsets of every layer inside the packed blob
struct BlobLayout {
    size_t off[kNumLayers];
    size_t woff[kNumLayers];   // three-pass Winograd F(6x6,3x3) form of the layer (0: none)
    size_t w1off[kNumLayers];  // 1-D F(4,3) form fused into the implicit-GEMM kernel (conv_w1d.hip): the other fp32 3x3 layers
    size_t w1off2[kNumLayers]; // ... and its F(2,3) form, for rows whose stage images do not fit the LDS with 18 weight planes
    size_t total;
};
// train: the layout of the training step's forward blob -- every 3x3 region can hold either the direct or the fused 1-D
// Winograd packing (the step picks per layer), no separate Winograd regions
inline size_t layer_floats(const LayerSpec &l, int f, int dt, bool train) {
    const size_t direct = nd_packed_floats(l.kind, lcin(l, f), lcout(l, f), dt);
    if (!train || dt != ND_F32 || (l.kind != ND_CONV3 && l.kind != ND_CONVT3)) return direct;
    const size_t w1 = nd_w1d_packed_floats(kW1dTile, lcin(l, f), lcout(l, f));
    return w1 > direct ? w1 : direct;
}
BlobLayout blob_layout(int f, int dt, bool with_wino = true, bool train = false) {
    BlobLayout b;
    size_t o = kHeaderFloats;
    for (int i = 0; i < kNumLayers; ++i) {
        b.off[i] = o;
        const LayerSpec &l = kLayers[i];
        if (i == kNumLayers - 1)
            o += ((size_t)3 * lcin(l, f) + 3 + 3) / 4 * 4;  // raw [3][cin] + bias[3] for the VALU 1x1 kernel
        else
            o += layer_floats(l, f, dt, train);
    }
    for (int i = 0; i < kNumLayers; ++i) {
        b.woff[i] = 0;
        if (with_wino && wino_layer(kLayers[i], f, dt)) {
            b.woff[i] = o;
            o += (nd_wino_packed_floats(kWinoTile, lcin(kLayers[i], f), lcout(kLayers[i], f)) + 63) / 64 * 64;
        }
        b.w1off[i] = b.w1off2[i] = 0;
        const LayerSpec &l = kLayers[i];
        if (with_wino && dt == ND_F32 && (l.kind == ND_CONV3 || l.kind == ND_CONVT3) && !b.woff[i]) {
            b.w1off[i] = o;
            o += (nd_w1d_packed_floats(kW1dTile, lcin(l, f), lcout(l, f)) + 63) / 64 * 64;
            b.w1off2[i] = o;
            o += (nd_w1d_packed_floats(2, lcin(l, f), lcout(l, f)) + 63) / 64 * 64;
        }
    }
    b.total = o;
    return b;
}

bool valid_cs(int cs) { return cs >= 104 && (cs - 56) % 16 == 0; }

int check_funit(int funit, int dtype) {
    if (dtype < ND_F32 || dtype > ND_F16) ND_FAIL(ND_EINVAL, "UtNet: unsupported dtype %d", dtype);
    const int q = 2 * nd_cpp(dtype);   // every conv input must be whole K blocks: 8 (fp32) / 16 (bf16, fp16) channels
    if (funit < q || funit % q) ND_FAIL(ND_EINVAL, "UtNet: funit=%d must be a positive multiple of %d for dtype %d", funit, q, dtype);
    return ND_OK;
}

int check_net(int funit, int h, int w, int batch, int dtype) {
    ND_TRY(check_funit(funit, dtype));
    for (int cs : {h, w})
        if (!valid_cs(cs))
            ND_FAIL(ND_EINVAL, "UtNet: tile size %d is not of the form 16k+56 (104, 120, ..., 248, 264, ..., 504, 520); "
                               "the reference network rejects it too (sizes of the skip concats do not match)", cs);
    if (batch <= 0) ND_FAIL(ND_EINVAL, "UtNet: batch=%d", batch);
    return ND_OK;
}

// ---------------------------------------------------------------- workspace plan
enum Buf { X0, A1, CAT4, P1, A2, CAT3, P2, A3, CAT2, P3, A4, CAT1, P4, BT0, BT1, T1A, T1B, T2A, T2B, T3A, T3B, T4A, T4B, NBUF };

struct Step {
    int layer;  // index into kLayers, or -1 for a pool
    Buf src, dst;
    int dst_plane0_mul;  // destination plane offset = mul * funit / 4
};
// the conv stack between the input pack and the final 1x1 (UtNet.py:99-107)
constexpr int kNumSteps = 26;
const Step kSteps[kNumSteps] = {
    {0, X0, A1, 0},     {1, A1, CAT4, 1},   {-1, CAT4, P1, 1},  {2, P1, A2, 0},    {3, A2, CAT3, 2},  {-1, CAT3, P2, 2},
    {4, P2, A3, 0},     {5, A3, CAT2, 4},   {-1, CAT2, P3, 4},  {6, P3, A4, 0},    {7, A4, CAT1, 8},  {-1, CAT1, P4, 8},
    {8, P4, BT0, 0},    {9, BT0, BT1, 0},   {10, BT1, CAT1, 0}, {11, CAT1, T1A, 0}, {12, T1A, T1B, 0}, {13, T1B, CAT2, 0},
    {14, CAT2, T2A, 0}, {15, T2A, T2B, 0},  {16, T2B, CAT3, 0}, {17, CAT3, T3A, 0}, {18, T3A, T3B, 0}, {19, T3B, CAT4, 0},
    {20, CAT4, T4A, 0}, {21, T4A, T4B, 0},
};

struct Plan {
    QpBuf buf[NBUF];
    float *split;   // split-K scratch shared by every conv launch of the stream (kSplitScratchBytes)
    char *wino;     // Winograd V / M scratch (largest layer at kWinoChunk images)
    size_t wino_bytes;
    size_t bytes;
};

// cap = batch the workspace was sized for; nimg = images in use (<= cap)
Plan make_plan(int f, int ch_, int cw_, int cap, int nimg, char *base, int dt) {
    Plan p;
    size_t off = 0;
    // `size` is the extent of the tensor for a SQUARE cs x cs input; the other dimension follows the same chain
    auto chain = [](int cs, int which) {
        const int l1 = cs, l2 = cs / 2 - 4, l3 = l2 / 2 - 4, l4 = l3 / 2 - 4, p4 = l4 / 2;
        const int v[] = {cs + 4, cs + 2, l1, l1 / 2, l1 / 2 - 2, l2, l2 / 2, l2 / 2 - 2, l3, l3 / 2, l3 / 2 - 2, l4, p4,
                         p4 - 2, p4, l4 + 2, l4 + 4, l3 + 2, l3 + 4, l2 + 2, l2 + 4, l1 + 2, l1 + 4};
        return v[which];
    };
    auto add = [&](Buf id, int ch, int /*size*/, int pad) {
        QpBuf &q = p.buf[id];
        q.planes = (ch + nd_cpp(dt) - 1) / nd_cpp(dt);
        if (id == X0) q.planes = 2;   // one K block: plane 0 = (r,g,b,0..), plane 1 = zeros
        q.dt = dt;
        q.B = nimg;
        q.Hb = chain(ch_, (int)id) + 2 * pad;
        q.Wb = chain(cw_, (int)id) + 2 * pad;
        q.pad = pad;
        q.pstride = (long)cap * q.Hb * q.Wb;
        q.base = (float *)(base + off);
        const size_t slack = nd_buf_slack(q.Wb);   // reads past the last plane (halo of the last tile / strip)
        off += ((size_t)q.planes * q.pstride + slack) * 16;
        off = (off + 255) & ~(size_t)255;
    };
    const int cs = ch_;
    const int l1 = cs, l2 = cs / 2 - 4, l3 = l2 / 2 - 4, l4 = l3 / 2 - 4, p4 = l4 / 2;
    add(X0, 8, cs + 4, 0);
    add(A1, f, cs + 2, 0);
    add(CAT4, 2 * f, l1, 2);
    add(P1, f, l1 / 2, 0);
    add(A2, 2 * f, l1 / 2 - 2, 0);
    add(CAT3, 4 * f, l2, 2);
    add(P2, 2 * f, l2 / 2, 0);
    add(A3, 4 * f, l2 / 2 - 2, 0);
    add(CAT2, 8 * f, l3, 2);
    add(P3, 4 * f, l3 / 2, 0);
    add(A4, 8 * f, l3 / 2 - 2, 0);
    add(CAT1, 16 * f, l4, 2);
    add(P4, 8 * f, p4, 0);
    add(BT0, 16 * f, p4 - 2, 2);
    add(BT1, 16 * f, p4, 0);
    add(T1A, 8 * f, l4 + 2, 2);
    add(T1B, 8 * f, l4 + 4, 0);
    add(T2A, 4 * f, l3 + 2, 2);
    add(T2B, 4 * f, l3 + 4, 0);
    add(T3A, 2 * f, l2 + 2, 2);
    add(T3B, 2 * f, l2 + 4, 0);
    add(T4A, f, l1 + 2, 2);
    add(T4B, f, l1 + 4, 0);
    p.split = (float *)(base + off);
    off += kSplitScratchBytes;
    p.wino = base + off;
    p.wino_bytes = 0;
    for (const Step &st : kSteps) {
        if (st.layer < 0 || !wino_layer(kLayers[st.layer], f, dt)) continue;
        QpBuf v = p.buf[st.src];
        v.B = cap < kWinoChunk ? cap : kWinoChunk;
        const size_t need = nd_wino_scratch_bytes(kWinoTile, v, lcin(kLayers[st.layer], f), lcout(kLayers[st.layer], f));
        if (need > p.wino_bytes) p.wino_bytes = need;
    }
    off += (p.wino_bytes + 255) & ~(size_t)255;
    p.bytes = off;
    return p;
}

// which kernel family runs step `st` of the stack (pl = the plan of the call: the fused 1-D form needs the row's LDS images to fit)
enum Form { FORM_POOL = -1, FORM_DIRECT = 0, FORM_W1D4 = 1, FORM_W1D2 = 2, FORM_WINO3P = 3 };
inline Form step_form(const Step &st, int f, int dt, int flags, const Plan &pl, const BlobLayout &bl, bool train,
                      const unsigned char *train_w1) {
    if (st.layer < 0) return FORM_POOL;
    const LayerSpec &l = kLayers[st.layer];
    if (l.kind != ND_CONV3 && l.kind != ND_CONVT3) return FORM_DIRECT;
    if (train) return (train_w1 && train_w1[st.layer]) ? FORM_W1D4 : FORM_DIRECT;
    if (flags & ND_FLAG_DIRECT_CONV) return FORM_DIRECT;
    if (bl.w1off[st.layer]) {
        if (!(flags & ND_FLAG_W1D_REGS) && nd_w2d_ok(pl.buf[st.src])) return FORM_W1D4;   // conv_w2d: any row width
        if (nd_w1d_fits(kW1dTile, pl.buf[st.src])) return FORM_W1D4;
        if (nd_w1d_fits(2, pl.buf[st.src])) return FORM_W1D2;
        return FORM_DIRECT;
    }
    return bl.woff[st.layer] ? FORM_WINO3P : FORM_DIRECT;
}

// Regions of interest of the decoder layers when only the centre [crop_h, H - crop_h) x [crop_w, W - crop_w) of the network
// output is used -- the fused denoise loop (denoise_image.py:249-258 crops every tile to its useful part before it is added to
// the canvas: pixels outside [pad, cs - pad) never reach it).  Walking the stack backwards from the final 1x1: a transposed 3x3
// layer needs input rows [lo - 2, hi) for output rows [lo, hi), a 2x2 stride-2 transpose input rows [lo / 2, (hi + 1) / 2).  With
// cs = 264 / ucs = 200 the last decoder level computes 204^2 of its 266^2 pixels, the level below 104^2 of 130^2; from the
// third level down everything is needed.  The encoder always runs whole (its pooled outputs feed every level).
struct Roi { int r0, c0, rows, cols; };
inline bool plan_rois(const Plan &pl, int crop_h, int crop_w, Roi *roi) {
    for (int i = 0; i < kNumSteps; ++i) roi[i] = Roi{0, 0, 0, 0};
    if (crop_h <= 0 && crop_w <= 0) return false;
    bool any = false;
    int lo[2][NBUF], hi[2][NBUF];
    for (int dim = 0; dim < 2; ++dim) {
        for (int b = 0; b < NBUF; ++b) {
            lo[dim][b] = 1 << 30;
            hi[dim][b] = -1;
        }
        const QpBuf &last = pl.buf[T4B];
        const int size = (dim ? last.Wb : last.Hb) - 2 * last.pad, crop = dim ? crop_w : crop_h;
        lo[dim][T4B] = crop + 2;          // final 1x1 + ZeroPad2d(-2): output pixel y is T4B pixel y + 2
        hi[dim][T4B] = size - 2 - crop;
        for (int i = kNumSteps - 1; i >= 0; --i) {
            const Step &st = kSteps[i];
            if (st.layer < 0) continue;
            const int kind = kLayers[st.layer].kind;
            if (kind != ND_CONVT3 && kind != ND_CONVT2S2) continue;
            if (hi[dim][st.dst] < 0) continue;   // nobody restricted this output
            const QpBuf &src = pl.buf[st.src];
            const int si = (dim ? src.Wb : src.Hb) - 2 * src.pad;
            int a = lo[dim][st.dst], b = hi[dim][st.dst];
            if (kind == ND_CONVT3) {
                a = a - 2 < 0 ? 0 : a - 2;
                b = b > si ? si : b;
            } else {
                a = a >> 1;
                b = (b + 1) >> 1;
                b = b > si ? si : b;
            }
            if (a < lo[dim][st.src]) lo[dim][st.src] = a;
            if (b > hi[dim][st.src]) hi[dim][st.src] = b;
        }
    }
    for (int i = 0; i < kNumSteps; ++i) {
        const Step &st = kSteps[i];
        if (st.layer < 0) continue;
        const int kind = kLayers[st.layer].kind;
        if (kind != ND_CONVT3 && kind != ND_CONVT2S2) continue;
        // the region lives on the layer's output grid (3x3) or input grid (2x2 stride-2)
        const Buf b = kind == ND_CONVT3 ? st.dst : st.src;
        const QpBuf &q = pl.buf[b];
        const int H = q.Hb - 2 * q.pad, W = q.Wb - 2 * q.pad;
        int r0 = 0, r1 = H, c0 = 0, c1 = W;
        if (kind == ND_CONVT3) {
            if (hi[0][b] >= 0) { r0 = lo[0][b]; r1 = hi[0][b]; }
            if (hi[1][b] >= 0) { c0 = lo[1][b]; c1 = hi[1][b]; }
        } else {
            if (hi[0][st.dst] >= 0) { r0 = lo[0][st.dst] >> 1; r1 = (hi[0][st.dst] + 1) >> 1; }
            if (hi[1][st.dst] >= 0) { c0 = lo[1][st.dst] >> 1; c1 = (hi[1][st.dst] + 1) >> 1; }
        }
        r0 = r0 < 0 ? 0 : r0; c0 = c0 < 0 ? 0 : c0;
        r1 = r1 > H ? H : r1; c1 = c1 > W ? W : c1;
        if (r0 == 0 && c0 == 0 && r1 == H && c1 == W) continue;
        if (r1 <= r0 || c1 <= c0) continue;
        roi[i] = Roi{r0, c0, r1 - r0, c1 - c0};
        any = true;
    }
    return any;
}

// every restricted layer must run in a kernel that takes a region (conv_qp, conv_w2d, three-pass F(6x6)): one that does not would
// compute its whole output from a producer that only wrote its region
inline bool rois_supported(int f, int dt, int flags, const Plan &pl, const BlobLayout &bl, const Roi *rois) {
    for (int i = 0; i < kNumSteps; ++i) {
        if (rois[i].rows <= 0) continue;
        const Form form = step_form(kSteps[i], f, dt, flags, pl, bl, false, nullptr);
        if (form == FORM_DIRECT) {
            // conv_qp walks linear pixel ranges: a region much narrower than its buffer may not fit any stage image -- then no
            // layer is restricted (a whole-tile layer needs whole-tile producers)
            const LayerSpec &l = kLayers[kSteps[i].layer];
            ConvDesc d;
            d.kind = l.kind;
            d.cin = lcin(l, f);
            d.cout = lcout(l, f);
            d.in = pl.buf[kSteps[i].src];
            d.out = pl.buf[kSteps[i].dst];
            d.variant = -1;
            d.roi_r0 = rois[i].r0;
            d.roi_c0 = rois[i].c0;
            d.roi_rows = rois[i].rows;
            d.roi_cols = rois[i].cols;
            if (!nd_conv_roi_fits(d)) return false;
            continue;
        }
        if (form == FORM_WINO3P) continue;
        if (form == FORM_W1D4 && !(flags & ND_FLAG_W1D_REGS)) continue;
        return false;
    }
    return true;
}

// ev (optional): kNumSteps+1 events, ev[i] recorded before step i, ev[kNumSteps] after the last one;
// ev_x (optional, with ev): 2 events per step, recorded after the input transform and after the GEMMs of a three-pass layer
// pre (optional, training): kNumSlopes compact buffers that receive acc + bias of every activated layer
// train_w1 (training forward, with pre): per layer, 1 = the layer's blob region holds the fused 1-D Winograd packing
// flags: nd_flags of the call (ND_FLAG_NO_SPLITK: every tile whole; ND_FLAG_DIRECT_CONV: no Winograd form on any layer)
int run_stack(int f, int act, int dt, const float *blob, const Plan &pl, hipStream_t s, int flags = 0, hipEvent_t *ev = nullptr,
              const QpBuf *pre = nullptr, const float *slopes = nullptr, const unsigned char *train_w1 = nullptr,
              hipEvent_t *ev_x = nullptr, const Roi *rois = nullptr) {
    const BlobLayout bl = blob_layout(f, dt, pre == nullptr, pre != nullptr);
    const int cpp = nd_cpp(dt);
    int si = 0;
    bool pool_done = false;   // the previous layer wrote the pooled tensor itself
    QpBuf pool_view;
    for (const Step &st : kSteps) {
        if (ev) ND_HIP(hipEventRecord(ev[si], s));
        const int this_step = si++;
        if (st.layer < 0) {
            if (pool_done) {
                pool_done = false;
                continue;
            }
            // pool reads the skip half of the concat buffer: planes [mul*f/4, 2*mul*f/4)
            ND_TRY(nd_launch_maxpool2(pl.buf[st.src], st.dst_plane0_mul * f / cpp, st.dst_plane0_mul * f / cpp, pl.buf[st.dst], s));
            continue;
        }
        const LayerSpec &l = kLayers[st.layer];
        ConvDesc d;
        d.kind = l.kind;
        d.act = l.prelu >= 0 ? act : ND_ACT_NONE;
        d.slope = 0.25f;
        d.slope_dev = (l.prelu >= 0 && act == ND_ACT_PRELU) ? (slopes ? slopes + l.prelu : blob + l.prelu) : nullptr;
        if (pre && l.prelu >= 0) {
            d.pre = pre[l.prelu].base;
            d.pre_plane = pre[l.prelu].np();
        }
        d.cin = lcin(l, f);
        d.cout = lcout(l, f);
        d.wpk = blob + bl.off[st.layer];
        d.bias = d.wpk + (size_t)nd_mtiles(l.kind, d.cout) * nd_kblocks(d.cin, dt) * nd_taps(l.kind) * 256;
        d.in = pl.buf[st.src];
        d.out = pl.buf[st.dst];
        d.out_plane0 = st.dst_plane0_mul * f / cpp;
        d.variant = -1;
        d.part = pl.split;
        d.part_bytes = kSplitScratchBytes;
        d.nosplit = (flags & ND_FLAG_NO_SPLITK) != 0;
        const Form form = step_form(st, f, dt, flags, pl, bl, pre != nullptr, train_w1);
        if (rois && rois[this_step].rows > 0) {
            d.roi_r0 = rois[this_step].r0;
            d.roi_c0 = rois[this_step].c0;
            d.roi_rows = rois[this_step].rows;
            d.roi_cols = rois[this_step].cols;
        }
        // MaxPool2d(2) fused into the producing layer's epilogue where its kernel can (conv_w2d, three-pass output transform):
        // the pool kernel re-read the whole skip tensor from HBM (2.4 % of the fp32 conv stack)
        const bool next_is_pool = this_step + 1 < kNumSteps && kSteps[this_step + 1].layer < 0;
        const bool w2d = form == FORM_W1D4 && !pre && !(flags & ND_FLAG_W1D_REGS);
        if (next_is_pool && !pre && !(flags & ND_FLAG_UNFUSED_POOL)) {
            pool_view = pl.buf[kSteps[this_step + 1].dst];
            d.pool = &pool_view;
            // fp32: conv_w2d and the three-pass output transform pool; 16-bit storage: conv_qp over 2-row bands of pixels, where a
            // workgroup shape fits the longer stage image
            pool_done = w2d || form == FORM_WINO3P || (form == FORM_DIRECT && dt != ND_F32 && nd_conv_pool_fits(d));
            if (!pool_done) d.pool = nullptr;
        }
        if (form == FORM_W1D4 || form == FORM_W1D2) {
            // narrow layer: 1-D Winograd along x inside the implicit-GEMM kernel; F(4,3), or F(2,3) on rows too wide for it
            const int T = form == FORM_W1D4 ? kW1dTile : 2;
            if (!pre) d.wpk = blob + (T == kW1dTile ? bl.w1off[st.layer] : bl.w1off2[st.layer]);
            d.bias = d.wpk + (size_t)nd_mtiles(ND_CONV3, d.cout) * nd_kblocks(d.cin) * 3 * (T + 2) * 256;
            // transform shared through LDS (conv_w2d.hip): inference always; training forward (it keeps the pre-activation copy, so its
            // tiles are never split along K) where the layer has tiles for two rounds of workgroups -- else conv_w1d, which splits
            const bool w2d_ok = T == 4 && !(flags & ND_FLAG_W1D_REGS) && nd_w2d_ok(d.in) && (!pre || nd_w2d_tiles(d.in, d.cout) >= 512);
            if (w2d_ok)
                ND_TRY(nd_launch_conv_w2d(d, s));
            else
                ND_TRY(nd_launch_conv_w1d(T, d, s));
            continue;
        }
        if (form == FORM_WINO3P) {
            // Winograd form, kWinoChunk images per pass (views of the same buffers)
            d.wpk = blob + bl.woff[st.layer];
            d.bias = nullptr;
            const int nimg = d.in.B;
            for (int b0 = 0; b0 < nimg; b0 += kWinoChunk) {
                ConvDesc c = d;
                c.in.B = c.out.B = nimg - b0 < kWinoChunk ? nimg - b0 : kWinoChunk;
                c.in.base = d.in.base + (size_t)b0 * d.in.Hb * d.in.Wb * 4;
                c.out.base = d.out.base + (size_t)b0 * d.out.Hb * d.out.Wb * 4;
                QpBuf pv;
                if (d.pool) {
                    pv = *d.pool;
                    pv.B = c.in.B;
                    pv.base = d.pool->base + (size_t)b0 * pv.Hb * pv.Wb * 4;
                    c.pool = &pv;
                }
                // (profiling: the split of a layer's time into its passes is recorded for a single-chunk layer only)
                ND_TRY(nd_launch_conv_wino(kWinoTile, c, pl.wino, pl.wino_bytes, s, (ev_x && nimg <= kWinoChunk) ? ev_x + 2 * this_step : nullptr));
            }
            continue;
        }
        ND_TRY(nd_launch_conv(d, s));
    }
    if (ev) ND_HIP(hipEventRecord(ev[si], s));
    return ND_OK;
}

}  // namespace
