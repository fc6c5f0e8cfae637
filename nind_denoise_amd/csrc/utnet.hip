// UtNet executor: static buffer plan per (funit, cs, batch), every layer enqueued on one stream.
// Reference: networks/UtNet.py:27-88 (layers), :97-109 (forward).  The concats of forward() are zero-copy: the
// up-sampling layer and the encoder skip both write straight into their halves of one bordered buffer
// (up-sampled channels FIRST, skip second -- UtNet.py:103-106).
#include <string.h>

#include <string>
#include <vector>

#include "nd_common.h"

#include "utnet_net.h"

// ------------------------------------------------------------------ C ABI
extern "C" int nd_utnet_num_tensors(void) { return (int)tensor_names().size(); }
extern "C" const char *nd_utnet_tensor_name(int idx) {
    const auto &n = tensor_names();
    return (idx >= 0 && idx < (int)n.size()) ? n[idx].c_str() : nullptr;
}

extern "C" size_t nd_utnet_packed_bytes(int funit, int dtype) {
    if (check_funit(funit, dtype) != ND_OK) return 0;
    return blob_layout(funit, dtype).total * sizeof(float);
}

extern "C" int nd_utnet_pack_weights(int funit, int dtype, const float *const *tensors, int n_tensors, void *packed_host,
                                     size_t packed_bytes) {
    ND_TRY(check_funit(funit, dtype));
    if (n_tensors != nd_utnet_num_tensors()) ND_FAIL(ND_EINVAL, "nd_utnet_pack_weights: expected %d tensors, got %d", nd_utnet_num_tensors(), n_tensors);
    const BlobLayout bl = blob_layout(funit, dtype);
    if (packed_bytes < bl.total * sizeof(float)) ND_FAIL(ND_ENOMEM, "nd_utnet_pack_weights: packed buffer too small");
    float *blob = (float *)packed_host;
    memset(blob, 0, bl.total * sizeof(float));
    for (int i = 0; i < kNumLayers; ++i) {
        const LayerSpec &l = kLayers[i];
        const int wi = tensor_index(std::string(l.key) + ".weight"), bi = tensor_index(std::string(l.key) + ".bias");
        if (wi < 0 || bi < 0 || !tensors[wi] || !tensors[bi]) ND_FAIL(ND_EINVAL, "nd_utnet_pack_weights: missing tensor %s.{weight,bias}", l.key);
        const int ci = lcin(l, funit), co = lcout(l, funit);
        if (i == kNumLayers - 1) {
            memcpy(blob + bl.off[i], tensors[wi], sizeof(float) * 3 * ci);
            memcpy(blob + bl.off[i] + 3 * ci, tensors[bi], sizeof(float) * 3);
        } else {
            nd_pack_layer(l.kind, ci, co, dtype, tensors[wi], tensors[bi], blob + bl.off[i]);
            if (bl.woff[i]) ND_TRY(nd_wino_pack(kWinoTile, l.kind, ci, co, tensors[wi], tensors[bi], blob + bl.woff[i]));
            if (bl.w1off[i]) ND_TRY(nd_w1d_pack(kW1dTile, l.kind, ci, co, tensors[wi], tensors[bi], blob + bl.w1off[i]));
            if (bl.w1off2[i]) ND_TRY(nd_w1d_pack(2, l.kind, ci, co, tensors[wi], tensors[bi], blob + bl.w1off2[i]));
        }
        if (l.prelu >= 0) {
            // activation module sits right after the layer in its Sequential: "<seq>.<k+1>.weight"
            std::string k(l.key);
            const size_t dot = k.rfind('.');
            const std::string an = k.substr(0, dot + 1) + std::to_string(atoi(k.c_str() + dot + 1) + 1) + ".weight";
            const int ai = tensor_index(an);
            blob[l.prelu] = (ai >= 0 && tensors[ai]) ? tensors[ai][0] : 0.25f;
        }
    }
    return ND_OK;
}

// Same blob from tensors that already live in HBM (fp32 storage only): device-side packers, nothing touches the host.
extern "C" int nd_utnet_pack_weights_device(int funit, int dtype, const float *const *tensors, int n_tensors, void *packed_dev,
                                            size_t packed_bytes, void *stream) {
    ND_TRY(check_funit(funit, dtype));
    if (dtype != ND_F32) ND_FAIL(ND_EINVAL, "nd_utnet_pack_weights_device: fp32 storage only (16-bit blobs are packed on the host)");
    if (n_tensors != nd_utnet_num_tensors()) ND_FAIL(ND_EINVAL, "nd_utnet_pack_weights_device: expected %d tensors, got %d", nd_utnet_num_tensors(), n_tensors);
    const BlobLayout bl = blob_layout(funit, dtype);
    if (!packed_dev || packed_bytes < bl.total * sizeof(float)) ND_FAIL(ND_ENOMEM, "nd_utnet_pack_weights_device: packed buffer too small");
    hipStream_t s = (hipStream_t)stream;
    float *blob = (float *)packed_dev;
    ND_HIP(hipMemsetAsync(blob, 0, bl.total * sizeof(float), s));
    for (int i = 0; i < kNumLayers; ++i) {
        const LayerSpec &l = kLayers[i];
        const int wi = tensor_index(std::string(l.key) + ".weight"), bi = tensor_index(std::string(l.key) + ".bias");
        if (wi < 0 || bi < 0 || !tensors[wi] || !tensors[bi]) ND_FAIL(ND_EINVAL, "nd_utnet_pack_weights_device: missing tensor %s.{weight,bias}", l.key);
        const int ci = lcin(l, funit), co = lcout(l, funit);
        if (i == kNumLayers - 1) {
            ND_HIP(hipMemcpyAsync(blob + bl.off[i], tensors[wi], sizeof(float) * 3 * ci, hipMemcpyDeviceToDevice, s));
            ND_HIP(hipMemcpyAsync(blob + bl.off[i] + 3 * ci, tensors[bi], sizeof(float) * 3, hipMemcpyDeviceToDevice, s));
        } else {
            ND_TRY(nd_pack_layer_device(l.kind, ci, co, tensors[wi], tensors[bi], blob + bl.off[i], s));
            if (bl.woff[i]) ND_TRY(nd_pack_wino_device(kWinoTile, l.kind, ci, co, tensors[wi], tensors[bi], blob + bl.woff[i], s));
            if (bl.w1off[i]) ND_TRY(nd_pack_w1d_device(kW1dTile, l.kind, ci, co, tensors[wi], tensors[bi], blob + bl.w1off[i], s));
            if (bl.w1off2[i]) ND_TRY(nd_pack_w1d_device(2, l.kind, ci, co, tensors[wi], tensors[bi], blob + bl.w1off2[i], s));
        }
        if (l.prelu >= 0) {
            std::string k(l.key);
            const size_t dot = k.rfind('.');
            const std::string an = k.substr(0, dot + 1) + std::to_string(atoi(k.c_str() + dot + 1) + 1) + ".weight";
            const int ai = tensor_index(an);
            if (ai >= 0 && tensors[ai]) {
                ND_HIP(hipMemcpyAsync(blob + l.prelu, tensors[ai], sizeof(float), hipMemcpyDeviceToDevice, s));
            } else {
                const float dflt = 0.25f;
                ND_HIP(hipMemcpyAsync(blob + l.prelu, &dflt, sizeof(float), hipMemcpyHostToDevice, s));
                ND_HIP(hipStreamSynchronize(s));   // (dflt lives on this stack frame)
            }
        }
    }
    return ND_OK;
}

extern "C" size_t nd_utnet_workspace_bytes_hw(int funit, int h, int w, int batch, int dtype) {
    if (check_net(funit, h, w, batch, dtype) != ND_OK) return 0;
    return make_plan(funit, h, w, batch, batch, nullptr, dtype).bytes;
}
extern "C" size_t nd_utnet_workspace_bytes(int funit, int cs, int batch, int dtype) {
    return nd_utnet_workspace_bytes_hw(funit, cs, cs, batch, dtype);
}

extern "C" int nd_utnet_workspace_init_hw(void *ws, size_t ws_bytes, int funit, int h, int w, int batch, int dtype,
                                          void *stream);
extern "C" int nd_utnet_workspace_init(void *ws, size_t ws_bytes, int funit, int cs, int batch, int dtype, void *stream) {
    return nd_utnet_workspace_init_hw(ws, ws_bytes, funit, cs, cs, batch, dtype, stream);
}
extern "C" int nd_utnet_workspace_init_hw(void *ws, size_t ws_bytes, int funit, int h, int w, int batch, int dtype,
                                          void *stream) {
    ND_TRY(check_net(funit, h, w, batch, dtype));
    const size_t need = make_plan(funit, h, w, batch, batch, nullptr, dtype).bytes;
    if (!ws || ws_bytes < need) ND_FAIL(ND_ENOMEM, "UtNet workspace: %zu B given, %zu B needed", ws_bytes, need);
    // zero borders (the implicit padding of the transpose convolutions), the unused input channel plane and the slack
    ND_HIP(hipMemsetAsync(ws, 0, need, (hipStream_t)stream));
    return ND_OK;
}

static int forward_common(int funit, int act, int dtype, const void *packed, int batch_cap, int nimg, int h, int w,
                          void *ws, size_t ws_bytes, Plan *out_plan) {
    ND_TRY(check_net(funit, h, w, batch_cap, dtype));
    if (act < ND_ACT_PRELU || act > ND_ACT_HARDSWISH) ND_FAIL(ND_EINVAL, "UtNet: unknown activation %d", act);
    if (nimg <= 0 || nimg > batch_cap) ND_FAIL(ND_EINVAL, "UtNet: %d images with a workspace batch of %d", nimg, batch_cap);
    if (!packed || !ws) ND_FAIL(ND_EINVAL, "UtNet: null pointer");
    if (((uintptr_t)ws & 15) || ((uintptr_t)packed & 15)) ND_FAIL(ND_EINVAL, "UtNet: workspace / weights must be 16-byte aligned");
    *out_plan = make_plan(funit, h, w, batch_cap, nimg, (char *)ws, dtype);
    if (ws_bytes < out_plan->bytes) ND_FAIL(ND_ENOMEM, "UtNet workspace: %zu B given, %zu B needed", ws_bytes, out_plan->bytes);
    return ND_OK;
}

extern "C" int nd_utnet_forward_hw(int funit, int act, int dtype, int flags, const void *packed, const float *x, float *y,
                                   int batch, int h, int w, void *ws, size_t ws_bytes, void *stream);
extern "C" int nd_utnet_forward(int funit, int act, int dtype, int flags, const void *packed, const float *x, float *y,
                                int batch, int cs, void *ws, size_t ws_bytes, void *stream) {
    return nd_utnet_forward_hw(funit, act, dtype, flags, packed, x, y, batch, cs, cs, ws, ws_bytes, stream);
}
extern "C" int nd_utnet_forward_hw(int funit, int act, int dtype, int flags, const void *packed, const float *x, float *y,
                                   int batch, int h, int w, void *ws, size_t ws_bytes, void *stream) {
    ND_TRY(nd_check_flags(flags));
    Plan pl;
    ND_TRY(forward_common(funit, act, dtype, packed, batch, batch, h, w, ws, ws_bytes, &pl));
    if (!x || !y) ND_FAIL(ND_EINVAL, "UtNet: null tensor");
    hipStream_t s = (hipStream_t)stream;
    const float *blob = (const float *)packed;
    ND_TRY(nd_launch_reflect_pack(x, batch, h, w, pl.buf[X0], s));
    ND_TRY(run_stack(funit, act, dtype, blob, pl, s, flags));
    const BlobLayout bl = blob_layout(funit, dtype);
    const float *fw = blob + bl.off[kNumLayers - 1];
    ND_TRY(nd_launch_final1x1(pl.buf[T4B], funit, fw, fw + 3 * funit, 2, y, h, w, s));
    return ND_OK;
}

extern "C" int nd_utnet_denoise_tiles(int funit, int act, int dtype, int flags, const void *packed, const float *img,
                                      float *canvas, int width, int height, int cs, int ucs, int ol, int tile_begin,
                                      int tile_count, int batch, void *ws, size_t ws_bytes, void *stream) {
    ND_TRY(nd_check_flags(flags));
    Plan pl;
    ND_TRY(forward_common(funit, act, dtype, packed, batch, tile_count, cs, cs, ws, ws_bytes, &pl));
    if (!img || !canvas) ND_FAIL(ND_EINVAL, "UtNet: null image");
    hipStream_t s = (hipStream_t)stream;
    const float *blob = (const float *)packed;
    ND_TRY(nd_launch_gather_pack(img, width, height, cs, ucs, ol, tile_begin, tile_count, pl.buf[X0], s));
    const BlobLayout bl = blob_layout(funit, dtype);
    // only the useful centre [pad, cs - pad) of a tile reaches the canvas (k_final1x1_stitch reads nothing else): the last decoder
    // levels compute just what that centre depends on
    Roi rois[kNumSteps];
    const Roi *use = nullptr;
    const int crop = (cs - ucs) / 2;
    if (!(flags & ND_FLAG_FULL_TILES) && plan_rois(pl, crop, crop, rois) && rois_supported(funit, dtype, flags, pl, bl, rois)) use = rois;
    ND_TRY(run_stack(funit, act, dtype, blob, pl, s, flags, nullptr, nullptr, nullptr, nullptr, nullptr, use));
    const float *fw = blob + bl.off[kNumLayers - 1];
    ND_TRY(nd_launch_final1x1_stitch(pl.buf[T4B], funit, fw, fw + 3 * funit, 2, canvas, width, height, cs, ucs, ol,
                                     tile_begin, tile_count, s));
    return ND_OK;
}

// Profiling entry point (bench.py roofline leg): one forward of the conv stack with a HIP event between every launch on
// `stream` (and around the GEMM launch of a three-pass Winograd layer).  Synchronises the stream.  26 entries, forward order:
// 22 MFMA conv layers and 4 pools.  FLOP conventions: `flops` = algorithmic (torch FlopCounterMode, no
// padding-zero MACs); `mfma_flops` = what the matrix cores execute in the form the layer ran in (MFMAs issued x 4096).
extern "C" int nd_utnet_profile_stack(int funit, int act, int dtype, int flags, const void *packed, int batch, int cs, int crop,
                                      void *ws, size_t ws_bytes, void *stream, nd_step_profile *steps, int max_steps) {
    ND_TRY(nd_check_flags(flags));
    Plan pl;
    ND_TRY(forward_common(funit, act, dtype, packed, batch, batch, cs, cs, ws, ws_bytes, &pl));
    if (max_steps < kNumSteps || !steps) ND_FAIL(ND_EINVAL, "nd_utnet_profile_stack: need room for %d steps", kNumSteps);
    if (crop < 0 || 2 * crop >= cs) ND_FAIL(ND_EINVAL, "nd_utnet_profile_stack: crop %d", crop);
    hipStream_t s = (hipStream_t)stream;
    // every argument is validated above: from here on the events are destroyed on every exit path
    struct Events {
        hipEvent_t ev[kNumSteps + 1 + 2 * kNumSteps];
        int n = 0;
        ~Events() { for (int i = 0; i < n; ++i) (void)hipEventDestroy(ev[i]); }
    } evs;
    for (; evs.n < kNumSteps + 1 + 2 * kNumSteps; ++evs.n) ND_HIP(hipEventCreate(&evs.ev[evs.n]));
    hipEvent_t *const ev = evs.ev, *const evx = evs.ev + kNumSteps + 1;
    const BlobLayout bl = blob_layout(funit, dtype);
    Roi rois[kNumSteps];
    const Roi *use = nullptr;
    if (!(flags & ND_FLAG_FULL_TILES) && plan_rois(pl, crop, crop, rois) && rois_supported(funit, dtype, flags, pl, bl, rois)) use = rois;
    int rc = run_stack(funit, act, dtype, (const float *)packed, pl, s, flags, ev, nullptr, nullptr, nullptr, evx, use);
    if (rc == ND_OK) {
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            nd_set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e));
            rc = ND_EHIP;
        }
    }
    for (int i = 0; i < kNumSteps && rc == ND_OK; ++i) {
        nd_step_profile &o = steps[i];
        memset(&o, 0, sizeof(o));
        if (hipEventElapsedTime(&o.ms, ev[i], ev[i + 1]) != hipSuccess) {
            nd_set_error("hipEventElapsedTime failed");
            rc = ND_EHIP;
            break;
        }
        const Step &st = kSteps[i];
        const Form form = step_form(st, funit, dtype, flags, pl, bl, false, nullptr);
        o.form = (int)form;
        o.kind = st.layer >= 0 ? kLayers[st.layer].kind : -1;
        const QpBuf &in = pl.buf[st.src], &out = pl.buf[st.dst];
        const double B = batch, esz = 16.0 / nd_cpp(dtype);      // bytes per stored channel value
        double hin = in.Hb - 2 * in.pad, win = in.Wb - 2 * in.pad;
        if (use && use[i].rows > 0) {   // the layer ran on a region: count what it computed
            const bool t3 = kLayers[st.layer].kind == ND_CONVT3;
            hin = use[i].rows - (t3 ? 2 : 0);   // (a transposed 3x3 layer's region is on its output grid = input + 2)
            win = use[i].cols - (t3 ? 2 : 0);
        }
        if (st.layer < 0) {
            const double c = st.dst_plane0_mul * funit;
            o.bytes = B * c * esz * (hin * win + (hin / 2) * (win / 2));
            continue;
        }
        const LayerSpec &l = kLayers[st.layer];
        const double ci = lcin(l, funit), co = lcout(l, funit);
        const double cip = nd_kblocks((int)ci, dtype) * 2.0 * nd_cpp(dtype);   // input channels padded to whole K blocks
        double hout, wout;
        switch (l.kind) {
            case ND_CONV3: hout = hin - 2; wout = win - 2; o.flops = 2.0 * hout * wout * ci * co * 9; break;
            case ND_CONVT3: hout = hin + 2; wout = win + 2; o.flops = 2.0 * hin * win * ci * co * 9; break;
            case ND_CONVT2S2: hout = 2 * hin; wout = 2 * win; o.flops = 2.0 * hin * win * ci * co * 4; break;
            default: hout = hin; wout = win; o.flops = 2.0 * hin * win * ci * co; break;
        }
        o.flops *= B;
        o.bytes = B * esz * (ci * hin * win + co * hout * wout) + 4.0 * ci * co * nd_taps(l.kind);
        auto cdiv = [](double a, double b) { return (double)(long)((a + b - 1) / b); };
        switch (form) {
            case FORM_W1D4: o.mfma_flops = 2.0 * 3 * 6 * cip * co * hout * cdiv(wout, 4) * B; break;
            case FORM_W1D2: o.mfma_flops = 2.0 * 3 * 4 * cip * co * hout * cdiv(wout, 2) * B; break;
            case FORM_WINO3P: o.mfma_flops = 2.0 * (kWinoTile + 2) * (kWinoTile + 2) * cip * co * cdiv(hout, kWinoTile) * cdiv(wout, kWinoTile) * B; break;
            default:
                o.mfma_flops = l.kind == ND_CONVT2S2 ? 2.0 * 4 * cip * co * hin * win * B
                                                     : 2.0 * nd_taps(l.kind) * cip * co * hout * wout * B;   // (zero-border MACs of a transposed layer included)
                break;
        }
        if (form == FORM_WINO3P && batch <= kWinoChunk) {
            float a = 0, b = 0, c = 0;
            if (hipEventElapsedTime(&a, ev[i], evx[2 * i]) == hipSuccess && hipEventElapsedTime(&b, evx[2 * i], evx[2 * i + 1]) == hipSuccess &&
                hipEventElapsedTime(&c, evx[2 * i + 1], ev[i + 1]) == hipSuccess) {
                o.ms_xform_in = a;
                o.ms_gemm = b;
                o.ms_xform_out = c;
            }
            QpBuf v = in;
            if (use && use[i].rows > 0) {   // the passes ran on a view of rows + 2 x cols + 2 bordered input pixels
                v.Hb = use[i].rows + 2;
                v.Wb = use[i].cols + 2;
            }
            nd_wino_xform_bytes(kWinoTile, v, (int)ci, (int)co, &o.xform_bytes_in, &o.xform_bytes_out);
        }
    }
    return rc;
}

// Host-only query (no GPU call): the region of step i that nd_utnet_denoise_tiles computes when only the centre
// [crop, cs - crop) of a tile is kept -- rect = {r0, c0, rows, cols} on the layer's output grid (3x3 layers) or input grid
// (2x2 stride-2 transposes), all zero where the step computes its whole tensor.  Returns the number of restricted steps.
extern "C" int nd_utnet_useful_region(int funit, int cs, int crop, int step, int *rect) {
    if (!valid_cs(cs) || funit <= 0 || step < 0 || step >= kNumSteps || !rect || crop < 0 || 2 * crop >= cs)
        ND_FAIL(ND_EINVAL, "nd_utnet_useful_region: bad arguments");
    const Plan pl = make_plan(funit, cs, cs, 1, 1, nullptr, ND_F32);
    Roi rois[kNumSteps];
    plan_rois(pl, crop, crop, rois);
    int n = 0;
    for (int i = 0; i < kNumSteps; ++i) n += rois[i].rows > 0;
    rect[0] = rois[step].r0;
    rect[1] = rois[step].c0;
    rect[2] = rois[step].rows;
    rect[3] = rois[step].cols;
    return n;
}

extern "C" const char *nd_utnet_step_name(int i) {
    if (i < 0 || i >= kNumSteps) return nullptr;
    return kSteps[i].layer >= 0 ? kLayers[kSteps[i].layer].key : "maxpool";
}

extern "C" double nd_utnet_flops(int funit, int cs) {
    if (!valid_cs(cs) || funit <= 0) return 0.0;
    const double f = funit;
    double mac = 0;
    int h = cs + 4;
    const double ch[4][2] = {{3, f}, {f, 2 * f}, {2 * f, 4 * f}, {4 * f, 8 * f}};
    for (int i = 0; i < 4; ++i) {
        mac += (double)(h - 2) * (h - 2) * ch[i][0] * ch[i][1] * 9;
        mac += (double)(h - 4) * (h - 4) * ch[i][1] * ch[i][1] * 9;
        h = (h - 4) / 2;
    }
    mac += (double)(h - 2) * (h - 2) * 8 * f * 16 * f * 9;
    mac += (double)(h - 2) * (h - 2) * 16 * f * 16 * f * 9;
    double c = 16 * f;
    for (int i = 0; i < 4; ++i) {
        mac += (double)h * h * c * (c / 2) * 4;
        h *= 2;
        mac += (double)h * h * c * (c / 2) * 9;
        mac += (double)(h + 2) * (h + 2) * (c / 2) * (c / 2) * 9;
        h += 4;
        c /= 2;
    }
    mac += (double)h * h * f * 3;
    return 2 * mac;
}

// ------------------------------------------------------------------ single-layer entry points (parity tests)
namespace {
struct LayerPlan {
    QpBuf in, out;
    float *split;
    size_t bytes;
};
LayerPlan layer_plan(int kind, int B, int cin, int cout, int h, int w, char *base, int dt = ND_F32) {
    LayerPlan p;
    const int ipad = kind == ND_CONVT3 ? 2 : 0;
    p.in.dt = p.out.dt = dt;
    p.in.planes = 2 * nd_kblocks(cin, dt);
    p.in.B = B;
    p.in.Hb = h + 2 * ipad;
    p.in.Wb = w + 2 * ipad;
    p.in.pad = ipad;
    p.in.pstride = (long)B * p.in.Hb * p.in.Wb;
    p.in.base = (float *)base;
    size_t off = ((size_t)p.in.planes * p.in.pstride + nd_buf_slack(p.in.Wb)) * 16;
    off = (off + 255) & ~(size_t)255;
    int oh, ow;
    switch (kind) {
        case ND_CONV3: oh = h - 2; ow = w - 2; break;
        case ND_CONVT3: oh = h + 2; ow = w + 2; break;
        case ND_CONVT2S2: oh = 2 * h; ow = 2 * w; break;
        case ND_CONV2S2: oh = h / 2; ow = w / 2; break;
        default: oh = h; ow = w; break;
    }
    p.out.planes = (cout + nd_cpp(dt) - 1) / nd_cpp(dt);
    p.out.B = B;
    p.out.Hb = oh;
    p.out.Wb = ow;
    p.out.pad = 0;
    p.out.pstride = (long)B * oh * ow;
    p.out.base = (float *)(base + off);
    off += ((size_t)p.out.planes * p.out.pstride + 64) * 16;
    off = (off + 255) & ~(size_t)255;
    p.split = (float *)(base + off);
    off += kSplitScratchBytes;
    p.bytes = off;
    return p;
}
}  // namespace

extern "C" size_t nd_layer_workspace_bytes(int kind, int batch, int cin, int cout, int h, int w, int dtype) {
    if (dtype < ND_F32 || dtype > ND_F16 || kind < 0 || kind > 4 || batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return 0;
    if (kind == ND_CONV3 && (h < 3 || w < 3)) return 0;
    return layer_plan(kind, batch, cin, cout, h, w, nullptr, dtype).bytes;
}

extern "C" int nd_layer_forward(int kind, int act, float slope, int dtype, const void *packed, const float *x, int batch,
                                int cin, int h, int w, int cout, float *y, void *ws, size_t ws_bytes, int variant, int flags,
                                void *stream) {
    ND_TRY(nd_check_flags(flags));
    if (dtype < ND_F32 || dtype > ND_F16) ND_FAIL(ND_EINVAL, "nd_layer_forward: unsupported dtype %d", dtype);
    const size_t need = nd_layer_workspace_bytes(kind, batch, cin, cout, h, w, dtype);
    if (!need) ND_FAIL(ND_EINVAL, "nd_layer_forward: bad shape");
    if (!ws || ws_bytes < need) ND_FAIL(ND_ENOMEM, "nd_layer_forward: workspace %zu B given, %zu B needed", ws_bytes, need);
    if (cout % nd_cpp(dtype)) ND_FAIL(ND_EINVAL, "nd_layer_forward: cout must be a multiple of %d", nd_cpp(dtype));
    hipStream_t s = (hipStream_t)stream;
    LayerPlan pl = layer_plan(kind, batch, cin, cout, h, w, (char *)ws, dtype);
    ND_HIP(hipMemsetAsync(ws, 0, need, s));
    ND_TRY(nd_launch_nchw_to_qp(x, cin, pl.in, 0, s));
    ConvDesc d;
    d.kind = kind;
    d.act = act;
    d.slope = slope;
    d.slope_dev = nullptr;
    d.cin = cin;
    d.cout = cout;
    d.wpk = (const float *)packed;
    d.bias = d.wpk + (size_t)nd_mtiles(kind, cout) * nd_kblocks(cin, dtype) * nd_taps(kind) * 256;
    d.in = pl.in;
    d.out = pl.out;
    d.out_plane0 = 0;
    d.variant = variant;
    d.part = pl.split;
    d.part_bytes = kSplitScratchBytes;
    d.nosplit = (flags & ND_FLAG_NO_SPLITK) != 0;
    ND_TRY(nd_launch_conv(d, s));
    ND_TRY(nd_launch_qp_to_nchw(pl.out, 0, y, cout, s));
    return ND_OK;
}

// Winograd form of a 3x3 layer (tile = 2 | 4): same interface as nd_layer_forward with a blob from nd_winograd_pack
// (tile = 1 | 3: the 1-D F(2,3) | F(4,3) form fused into the implicit-GEMM kernel, conv_w1d.hip)
static bool wino_tile_ok(int tile) { return tile >= 1 && tile <= 6; }   // 5: the F(4,3) form of tile 3 through conv_w2d; 6: three-pass F(6x6,3x3)
extern "C" size_t nd_winograd_packed_bytes(int tile, int cin, int cout) {
    if (!wino_tile_ok(tile) || cin <= 0 || cout <= 0) return 0;
    if (tile == 5) tile = 3;
    return ((tile & 1) ? nd_w1d_packed_floats(tile + 1, cin, cout) : nd_wino_packed_floats(tile, cin, cout)) * sizeof(float);
}
extern "C" int nd_winograd_pack(int tile, int kind, int cin, int cout, const float *w, const float *bias, void *packed,
                                size_t packed_bytes) {
    const size_t need = nd_winograd_packed_bytes(tile, cin, cout);
    if (!need) ND_FAIL(ND_EINVAL, "nd_winograd_pack: bad shape");
    if (!packed || packed_bytes < need) ND_FAIL(ND_ENOMEM, "nd_winograd_pack: %zu B given, %zu B needed", packed_bytes, need);
    if (tile == 5) tile = 3;
    if (tile & 1) return nd_w1d_pack(tile + 1, kind, cin, cout, w, bias, (float *)packed);
    return nd_wino_pack(tile, kind, cin, cout, w, bias, (float *)packed);
}
extern "C" size_t nd_layer_winograd_workspace_bytes(int tile, int kind, int batch, int cin, int cout, int h, int w) {
    if (!wino_tile_ok(tile) || (kind != ND_CONV3 && kind != ND_CONVT3)) return 0;
    const size_t base = nd_layer_workspace_bytes(kind, batch, cin, cout, h, w, ND_F32);
    if (!base) return 0;
    if (tile & 1) return base;
    const LayerPlan pl = layer_plan(kind, batch, cin, cout, h, w, nullptr, ND_F32);
    return base + nd_wino_scratch_bytes(tile, pl.in, cin, cout);
}
extern "C" int nd_layer_forward_winograd(int tile, int kind, int act, float slope, const void *packed, const float *x, int batch,
                                         int cin, int h, int w, int cout, float *y, void *ws, size_t ws_bytes, int flags,
                                         void *stream) {
    ND_TRY(nd_check_flags(flags));
    const size_t need = nd_layer_winograd_workspace_bytes(tile, kind, batch, cin, cout, h, w);
    if (!need) ND_FAIL(ND_EINVAL, "nd_layer_forward_winograd: bad shape / kind / tile");
    if (!ws || ws_bytes < need) ND_FAIL(ND_ENOMEM, "nd_layer_forward_winograd: workspace %zu B given, %zu B needed", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    LayerPlan pl = layer_plan(kind, batch, cin, cout, h, w, (char *)ws, ND_F32);
    ND_HIP(hipMemsetAsync(ws, 0, pl.bytes, s));
    ND_TRY(nd_launch_nchw_to_qp(x, cin, pl.in, 0, s));
    ConvDesc d;
    d.kind = kind;
    d.act = act;
    d.slope = slope;
    d.slope_dev = nullptr;
    d.cin = cin;
    d.cout = cout;
    d.wpk = (const float *)packed;
    d.bias = nullptr;
    d.in = pl.in;
    d.out = pl.out;
    d.out_plane0 = 0;
    d.variant = -1;
    d.part = pl.split;
    d.part_bytes = kSplitScratchBytes;
    d.nosplit = (flags & ND_FLAG_NO_SPLITK) != 0;
    if (tile == 5) {
        d.bias = d.wpk + (size_t)nd_mtiles(ND_CONV3, cout) * nd_kblocks(cin) * 3 * 6 * 256;
        ND_TRY(nd_launch_conv_w2d(d, s));
    } else if (tile & 1) {
        d.bias = d.wpk + (size_t)nd_mtiles(ND_CONV3, cout) * nd_kblocks(cin) * 3 * (tile + 3) * 256;
        ND_TRY(nd_launch_conv_w1d(tile + 1, d, s));
    } else {
        ND_TRY(nd_launch_conv_wino(tile, d, (char *)ws + pl.bytes, ws_bytes - pl.bytes, s));
    }
    ND_TRY(nd_launch_qp_to_nchw(pl.out, 0, y, cout, s));
    return ND_OK;
}

extern "C" int nd_maxpool2_forward(const float *x, int batch, int c, int h, int w, float *y, void *ws, size_t ws_bytes,
                                   void *stream) {
    if (batch <= 0 || c <= 0 || h < 2 || w < 2) ND_FAIL(ND_EINVAL, "nd_maxpool2_forward: bad shape");
    QpBuf in, out;
    in.planes = out.planes = (c + 3) / 4;
    in.B = out.B = batch;
    in.Hb = h; in.Wb = w; in.pad = 0; in.pstride = (long)batch * h * w;
    out.Hb = h / 2; out.Wb = w / 2; out.pad = 0; out.pstride = (long)batch * (h / 2) * (w / 2);
    const size_t ib = ((size_t)in.planes * in.pstride * 16 + 255) & ~(size_t)255;
    const size_t need = ib + (size_t)out.planes * out.pstride * 16;
    if (!ws || ws_bytes < need) ND_FAIL(ND_ENOMEM, "nd_maxpool2_forward: workspace %zu B given, %zu B needed", ws_bytes, need);
    in.base = (float *)ws;
    out.base = (float *)((char *)ws + ib);
    hipStream_t s = (hipStream_t)stream;
    ND_TRY(nd_launch_nchw_to_qp(x, c, in, 0, s));
    ND_TRY(nd_launch_maxpool2(in, 0, in.planes, out, s));
    ND_TRY(nd_launch_qp_to_nchw(out, 0, y, c, s));
    return ND_OK;
}

// Kernel micro-benchmark (tools/bench_layers.py): `iters` launches of one conv layer on pseudo-random data already
// in the quad-planar layout; reports the mean launch duration from HIP events on `stream`.  Synchronises.
__global__ void k_fill_random(float *p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xFFFF) - 32768.f) * (1.f / 65536.f);
    }
}

__global__ void k_fill_random16(unsigned short *p, size_t n, unsigned seed, int dt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float f = ((h & 0xFFFF) - 32768.f) * (1.f / 65536.f);
        if (dt == ND_BF16) {
            const __bf16 v = (__bf16)f;
            p[i] = *(const unsigned short *)&v;
        } else {
            const _Float16 v = (_Float16)f;
            p[i] = *(const unsigned short *)&v;
        }
    }
}

extern "C" int nd_conv_bench(int kind, int dtype, int batch, int cin, int cout, int h, int w, int variant, int iters, void *ws,
                             size_t ws_bytes, void *stream, float *mean_ms) {
    const size_t need = nd_layer_workspace_bytes(kind, batch, cin, cout, h, w, dtype);
    const size_t wfloats = need ? nd_packed_floats(kind, cin, cout, dtype) : 0;
    if (!need) ND_FAIL(ND_EINVAL, "nd_conv_bench: bad shape");
    const size_t total = need + ((wfloats * 4 + 255) & ~(size_t)255);
    if (!ws || ws_bytes < total) ND_FAIL(ND_ENOMEM, "nd_conv_bench: workspace %zu B given, %zu B needed", ws_bytes, total);
    hipStream_t s = (hipStream_t)stream;
    LayerPlan pl = layer_plan(kind, batch, cin, cout, h, w, (char *)ws, dtype);
    float *wpk = (float *)((char *)ws + need);
    if (dtype == ND_F32) {
        hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, s, (float *)ws, need / 4, 12345u);
        hipLaunchKernelGGL(k_fill_random, dim3(1024), dim3(256), 0, s, wpk, wfloats, 777u);
    } else {
        // pseudo-random finite 16-bit patterns: fill as bf16/fp16 values in (-0.5, 0.5) through the fp32 generator + convert
        hipLaunchKernelGGL(k_fill_random16, dim3(2048), dim3(256), 0, s, (unsigned short *)ws, need / 2, 12345u, dtype);
        hipLaunchKernelGGL(k_fill_random16, dim3(1024), dim3(256), 0, s, (unsigned short *)wpk, wfloats * 2, 777u, dtype);
        hipLaunchKernelGGL(k_fill_random, dim3(64), dim3(256), 0, s, wpk + (size_t)nd_mtiles(kind, cout) * nd_kblocks(cin, dtype) * nd_taps(kind) * 256,
                           (size_t)nd_mtiles(kind, cout) * 32, 99u);
    }
    ConvDesc d;
    d.kind = kind;
    d.act = ND_ACT_PRELU;
    d.slope = 0.2f;
    d.slope_dev = nullptr;
    d.cin = cin;
    d.cout = cout;
    d.wpk = wpk;
    d.bias = wpk + (size_t)nd_mtiles(kind, cout) * nd_kblocks(cin, dtype) * nd_taps(kind) * 256;
    d.in = pl.in;
    d.out = pl.out;
    d.out_plane0 = 0;
    d.variant = variant;
    d.part = pl.split;
    d.part_bytes = kSplitScratchBytes;
    ND_TRY(nd_launch_conv(d, s));  // warm-up (also validates the variant)
    hipEvent_t e0, e1;
    ND_HIP(hipEventCreate(&e0));
    ND_HIP(hipEventCreate(&e1));
    ND_HIP(hipEventRecord(e0, s));
    int rc = ND_OK;
    for (int i = 0; i < iters && rc == ND_OK; ++i) rc = nd_launch_conv(d, s);
    (void)hipEventRecord(e1, s);
    if (hipStreamSynchronize(s) != hipSuccess && rc == ND_OK) {
        nd_set_error("nd_conv_bench: stream failed");
        rc = ND_EHIP;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (mean_ms) *mean_ms = ms / (iters > 0 ? iters : 1);
    return rc;
}

extern "C" int nd_num_conv_variants(void) { return nd_conv_variant_count(); }
extern "C" const char *nd_conv_variant_name(int v) { return nd_conv_variant_label(v); }

// Same measurement for the Winograd form (tile = 2 | 4) of a 3x3 layer: the three passes per iteration.
// workspace: nd_layer_winograd_workspace_bytes + nd_winograd_packed_bytes + 256 B
extern "C" int nd_winograd_bench(int tile, int kind, int batch, int cin, int cout, int h, int w, int iters, void *ws,
                                 size_t ws_bytes, void *stream, float *mean_ms) {
    const size_t need = nd_layer_winograd_workspace_bytes(tile, kind, batch, cin, cout, h, w);
    if (!need) ND_FAIL(ND_EINVAL, "nd_winograd_bench: bad shape / kind / tile");
    const size_t wfloats = (tile & 1) ? nd_w1d_packed_floats(tile == 5 ? 4 : tile + 1, cin, cout) : nd_wino_packed_floats(tile, cin, cout);
    const size_t total = ((need + 255) & ~(size_t)255) + wfloats * 4;
    if (!ws || ws_bytes < total) ND_FAIL(ND_ENOMEM, "nd_winograd_bench: workspace %zu B given, %zu B needed", ws_bytes, total);
    hipStream_t s = (hipStream_t)stream;
    LayerPlan pl = layer_plan(kind, batch, cin, cout, h, w, (char *)ws, ND_F32);
    float *wpk = (float *)((char *)ws + ((need + 255) & ~(size_t)255));
    hipLaunchKernelGGL(k_fill_random, dim3(2048), dim3(256), 0, s, (float *)ws, pl.bytes / 4, 12345u);
    hipLaunchKernelGGL(k_fill_random, dim3(1024), dim3(256), 0, s, wpk, wfloats, 777u);
    ConvDesc d;
    d.kind = kind;
    d.act = ND_ACT_PRELU;
    d.slope = 0.2f;
    d.slope_dev = nullptr;
    d.cin = cin;
    d.cout = cout;
    d.wpk = wpk;
    d.bias = nullptr;
    d.in = pl.in;
    d.out = pl.out;
    d.out_plane0 = 0;
    d.variant = -1;
    d.part = pl.split;
    d.part_bytes = kSplitScratchBytes;
    void *scratch = (char *)ws + pl.bytes;
    const size_t scratch_bytes = need - pl.bytes;
    if (tile & 1) d.bias = d.wpk + (size_t)nd_mtiles(ND_CONV3, cout) * nd_kblocks(cin) * 3 * ((tile == 5 ? 3 : tile) + 3) * 256;
    auto run = [&]() {
        return tile == 5 ? nd_launch_conv_w2d(d, s) : ((tile & 1) ? nd_launch_conv_w1d(tile + 1, d, s) : nd_launch_conv_wino(tile, d, scratch, scratch_bytes, s));
    };
    ND_TRY(run());
    hipEvent_t e0, e1;
    ND_HIP(hipEventCreate(&e0));
    ND_HIP(hipEventCreate(&e1));
    ND_HIP(hipEventRecord(e0, s));
    int rc = ND_OK;
    for (int i = 0; i < iters && rc == ND_OK; ++i) rc = run();
    (void)hipEventRecord(e1, s);
    if (hipStreamSynchronize(s) != hipSuccess && rc == ND_OK) {
        nd_set_error("nd_winograd_bench: stream failed");
        rc = ND_EHIP;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (mean_ms) *mean_ms = ms / (iters > 0 ? iters : 1);
    return rc;
}
