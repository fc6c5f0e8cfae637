// UNet executor (reference: networks/ThirdPartyNets.py:62-169, eval mode) on the same quad-planar conv kernel:
//   Conv2d(3, padding=1)  -> valid 3x3 correlation on a buffer with a 1-pixel zero border
//   BatchNorm2d (eval)    -> folded into the conv weights / bias at pack time (running stats, eps 1e-5)
//   ReLU                  -> the PReLU epilogue with slope 0
//   cat([skip, up])       -> zero-copy: skip FIRST, up-sampled second (the opposite of UtNet; ThirdPartyNets.py:124)
//   F.pad fix-up (:110-118) for odd sizes -> the 2x2 stride-2 result is written at offset 0 of a destination that is one
//                                            row / column larger; the remainder stays zero (never written)
//   outc + Sigmoid        -> k_final1x1 with the sigmoid flag
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "nd_common.h"

namespace {

struct ULayer {
    std::string key;   // conv / convT module path
    std::string bn;    // BatchNorm module path ("" = none)
    int kind, cin, cout;
};

std::vector<ULayer> build_layers() {
    std::vector<ULayer> L;
    auto dconv = [&](const std::string &p, int ci, int co) {
        L.push_back({p + ".0", p + ".1", ND_CONV3, ci, co});
        L.push_back({p + ".3", p + ".4", ND_CONV3, co, co});
    };
    dconv("inc.conv.conv", 3, 64);
    const int dc[4][2] = {{64, 128}, {128, 256}, {256, 512}, {512, 512}};
    for (int n = 0; n < 4; ++n) dconv("down" + std::to_string(n + 1) + ".mpconv.1.conv", dc[n][0], dc[n][1]);
    const int uc[4][2] = {{1024, 256}, {512, 128}, {256, 64}, {128, 64}};
    for (int n = 0; n < 4; ++n) {
        const std::string u = "up" + std::to_string(n + 1);
        L.push_back({u + ".up", "", ND_CONVT2S2, uc[n][0] / 2, uc[n][0] / 2});
        dconv(u + ".conv.conv", uc[n][0], uc[n][1]);
    }
    L.push_back({"outc.conv", "", ND_CONV1, 64, 3});
    return L;
}
const std::vector<ULayer> &layers() {
    static const std::vector<ULayer> L = build_layers();
    return L;
}
std::vector<std::string> build_names() {
    std::vector<std::string> n;
    for (const ULayer &l : layers()) {
        n.push_back(l.key + ".weight");
        n.push_back(l.key + ".bias");
        if (!l.bn.empty())
            for (const char *s : {".weight", ".bias", ".running_mean", ".running_var"}) n.push_back(l.bn + s);
    }
    return n;
}
const std::vector<std::string> &names() {
    static const std::vector<std::string> n = build_names();
    return n;
}
int name_index(const std::string &s) {
    const auto &n = names();
    for (size_t i = 0; i < n.size(); ++i)
        if (n[i] == s) return (int)i;
    return -1;
}

struct Blob {
    std::vector<size_t> off;
    size_t total;
};
Blob blob_layout() {
    Blob b;
    size_t o = 0;
    for (const ULayer &l : layers()) {
        b.off.push_back(o);
        o += l.kind == ND_CONV1 ? (size_t)(3 * l.cin + 3 + 3) / 4 * 4 : nd_packed_floats(l.kind, l.cin, l.cout);
    }
    b.total = o;
    return b;
}

enum UB { XIN, I1, CAT4, Q1, D1, CAT3, Q2, D2, CAT2, Q3, D3, CAT1, Q4, D4, X5, U1A, U1B, U2A, U2B, U3A, U3B, U4A, U4B, NUB };
struct UPlan {
    QpBuf buf[NUB];
    float *split;
    size_t bytes;
};
UPlan make_plan(int h, int w, int B, char *base) {
    UPlan p;
    size_t off = 0;
    int hs[5] = {h}, ws[5] = {w};
    for (int i = 1; i < 5; ++i) {
        hs[i] = hs[i - 1] / 2;
        ws[i] = ws[i - 1] / 2;
    }
    auto add = [&](UB id, int ch, int lvl, int pad) {
        QpBuf &q = p.buf[id];
        q.planes = id == XIN ? 2 : ch / 4;
        q.B = B;
        q.Hb = hs[lvl] + 2 * pad;
        q.Wb = ws[lvl] + 2 * pad;
        q.pad = pad;
        q.pstride = (long)B * q.Hb * q.Wb;
        q.base = (float *)(base + off);
        off += ((size_t)q.planes * q.pstride + nd_buf_slack(q.Wb)) * 16;
        off = (off + 255) & ~(size_t)255;
    };
    add(XIN, 8, 0, 1); add(I1, 64, 0, 1); add(CAT4, 128, 0, 1);
    add(Q1, 64, 1, 1); add(D1, 128, 1, 1); add(CAT3, 256, 1, 1);
    add(Q2, 128, 2, 1); add(D2, 256, 2, 1); add(CAT2, 512, 2, 1);
    add(Q3, 256, 3, 1); add(D3, 512, 3, 1); add(CAT1, 1024, 3, 1);
    add(Q4, 512, 4, 1); add(D4, 512, 4, 1); add(X5, 512, 4, 0);
    add(U1A, 256, 3, 1); add(U1B, 256, 3, 0);
    add(U2A, 128, 2, 1); add(U2B, 128, 2, 0);
    add(U3A, 64, 1, 1); add(U3B, 64, 1, 0);
    add(U4A, 64, 0, 1); add(U4B, 64, 0, 0);
    p.split = (float *)(base + off);
    off += kSplitScratchBytes;
    p.bytes = off;
    return p;
}

struct UStep {
    int layer;  // index into layers(), -1: pool
    UB src, dst;
    int dst_plane0;  // destination plane offset (channels / 4); for pools: number of planes pooled from plane 0
};
const UStep kSteps[] = {
    {0, XIN, I1, 0},    {1, I1, CAT4, 0},   {-1, CAT4, Q1, 16},  {2, Q1, D1, 0},     {3, D1, CAT3, 0},   {-1, CAT3, Q2, 32},
    {4, Q2, D2, 0},     {5, D2, CAT2, 0},   {-1, CAT2, Q3, 64},  {6, Q3, D3, 0},     {7, D3, CAT1, 0},   {-1, CAT1, Q4, 128},
    {8, Q4, D4, 0},     {9, D4, X5, 0},     {10, X5, CAT1, 128}, {11, CAT1, U1A, 0}, {12, U1A, U1B, 0},  {13, U1B, CAT2, 64},
    {14, CAT2, U2A, 0}, {15, U2A, U2B, 0},  {16, U2B, CAT3, 32}, {17, CAT3, U3A, 0}, {18, U3A, U3B, 0},  {19, U3B, CAT4, 16},
    {20, CAT4, U4A, 0}, {21, U4A, U4B, 0},
};

int check(int h, int w, int batch, int dtype) {
    if (dtype != ND_F32) ND_FAIL(ND_EINVAL, "UNet: unsupported dtype %d", dtype);
    if (h < 16 || w < 16 || batch <= 0) ND_FAIL(ND_EINVAL, "UNet: input %dx%dx%d too small (four 2x2 pools)", batch, h, w);
    return ND_OK;
}

}  // namespace

extern "C" int nd_unet_num_tensors(void) { return (int)names().size(); }
extern "C" const char *nd_unet_tensor_name(int i) {
    return (i >= 0 && i < (int)names().size()) ? names()[i].c_str() : nullptr;
}
extern "C" size_t nd_unet_packed_bytes(int dtype) { return dtype == ND_F32 ? blob_layout().total * sizeof(float) : 0; }

extern "C" int nd_unet_pack_weights(int dtype, const float *const *tensors, int n_tensors, void *packed_host,
                                    size_t packed_bytes) {
    if (dtype != ND_F32) ND_FAIL(ND_EINVAL, "nd_unet_pack_weights: unsupported dtype %d", dtype);
    if (n_tensors != nd_unet_num_tensors()) ND_FAIL(ND_EINVAL, "nd_unet_pack_weights: expected %d tensors", nd_unet_num_tensors());
    const Blob bl = blob_layout();
    if (packed_bytes < bl.total * sizeof(float)) ND_FAIL(ND_ENOMEM, "nd_unet_pack_weights: packed buffer too small");
    float *blob = (float *)packed_host;
    memset(blob, 0, bl.total * sizeof(float));
    const auto &L = layers();
    for (size_t i = 0; i < L.size(); ++i) {
        const ULayer &l = L[i];
        const float *w = tensors[name_index(l.key + ".weight")], *b = tensors[name_index(l.key + ".bias")];
        if (!w || !b) ND_FAIL(ND_EINVAL, "nd_unet_pack_weights: missing %s", l.key.c_str());
        if (l.kind == ND_CONV1) {
            memcpy(blob + bl.off[i], w, sizeof(float) * 3 * l.cin);
            memcpy(blob + bl.off[i] + 3 * l.cin, b, sizeof(float) * 3);
            continue;
        }
        if (l.bn.empty()) {
            nd_pack_layer_f32(l.kind, l.cin, l.cout, w, b, blob + bl.off[i]);
            continue;
        }
        // fold eval-mode BatchNorm2d: y = (conv + b - mean) * gamma / sqrt(var + eps) + beta
        const float *g = tensors[name_index(l.bn + ".weight")], *be = tensors[name_index(l.bn + ".bias")];
        const float *rm = tensors[name_index(l.bn + ".running_mean")], *rv = tensors[name_index(l.bn + ".running_var")];
        if (!g || !be || !rm || !rv) ND_FAIL(ND_EINVAL, "nd_unet_pack_weights: missing BatchNorm tensors of %s", l.bn.c_str());
        std::vector<float> wf((size_t)l.cout * l.cin * 9), bf(l.cout);
        for (int co = 0; co < l.cout; ++co) {
            const float sc = g[co] / sqrtf(rv[co] + 1e-5f);
            for (int k = 0; k < l.cin * 9; ++k) wf[(size_t)co * l.cin * 9 + k] = w[(size_t)co * l.cin * 9 + k] * sc;
            bf[co] = (b[co] - rm[co]) * sc + be[co];
        }
        nd_pack_layer_f32(l.kind, l.cin, l.cout, wf.data(), bf.data(), blob + bl.off[i]);
    }
    return ND_OK;
}

extern "C" size_t nd_unet_workspace_bytes(int h, int w, int batch, int dtype) {
    if (check(h, w, batch, dtype) != ND_OK) return 0;
    return make_plan(h, w, batch, nullptr).bytes;
}

extern "C" int nd_unet_workspace_init(void *ws, size_t ws_bytes, int h, int w, int batch, int dtype, void *stream) {
    ND_TRY(check(h, w, batch, dtype));
    const size_t need = make_plan(h, w, batch, nullptr).bytes;
    if (!ws || ws_bytes < need) ND_FAIL(ND_ENOMEM, "UNet workspace: %zu B given, %zu B needed", ws_bytes, need);
    ND_HIP(hipMemsetAsync(ws, 0, need, (hipStream_t)stream));
    return ND_OK;
}

// UNet.forward (ThirdPartyNets.py:153-169, find_noise handled by the caller): x [B,3,H,W] -> sigmoid(outc(...)) [B,3,H,W]
extern "C" int nd_unet_forward(int dtype, const void *packed, const float *x, float *y, int batch, int h, int w, void *ws,
                               size_t ws_bytes, void *stream) {
    ND_TRY(check(h, w, batch, dtype));
    if (!packed || !x || !y || !ws) ND_FAIL(ND_EINVAL, "UNet: null pointer");
    UPlan pl = make_plan(h, w, batch, (char *)ws);
    if (ws_bytes < pl.bytes) ND_FAIL(ND_ENOMEM, "UNet workspace: %zu B given, %zu B needed", ws_bytes, pl.bytes);
    hipStream_t s = (hipStream_t)stream;
    const float *blob = (const float *)packed;
    const Blob bl = blob_layout();
    const auto &L = layers();
    ND_TRY(nd_launch_nchw_to_qp(x, 3, pl.buf[XIN], 0, s));
    for (const UStep &st : kSteps) {
        if (st.layer < 0) {
            ND_TRY(nd_launch_maxpool2(pl.buf[st.src], 0, st.dst_plane0, pl.buf[st.dst], s));
            continue;
        }
        const ULayer &l = L[st.layer];
        ConvDesc d;
        d.kind = l.kind;
        d.act = l.kind == ND_CONV3 ? ND_ACT_PRELU : ND_ACT_NONE;   // ReLU = PReLU with slope 0
        d.slope = 0.f;
        d.slope_dev = nullptr;
        d.cin = l.cin;
        d.cout = l.cout;
        d.wpk = blob + bl.off[st.layer];
        d.bias = d.wpk + (size_t)nd_mtiles(l.kind, l.cout) * nd_kblocks(l.cin) * nd_taps(l.kind) * 256;
        d.in = pl.buf[st.src];
        d.out = pl.buf[st.dst];
        d.out_plane0 = st.dst_plane0;
        d.variant = -1;
        d.part = pl.split;
        d.part_bytes = kSplitScratchBytes;
        ND_TRY(nd_launch_conv_f32(d, s));
    }
    const float *fw = blob + bl.off[L.size() - 1];
    ND_TRY(nd_launch_final1x1(pl.buf[U4B], 64, fw, fw + 3 * 64, 0, y, h, w, s, 1));
    return ND_OK;
}
