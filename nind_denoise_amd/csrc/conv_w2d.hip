// 3x3 convolution, 1-D Winograd F(4,3) along x, with the input transform SHARED by the workgroup through LDS (fp32 inference).
//
// Why a second form of conv_w1d.hip: on gfx950 v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 datapath -- every VALU instruction a
// wave issues next to it costs its full issue time in matrix throughput (tools/ubench/mfma_f32_fillers.hip: +4.5 cycles per
// VALU instruction of any kind at two waves per SIMD, +3.5 per ds_read_b128, against 64 cycles per MFMA; nothing hides).  conv_w1d
// computes v = B^T d in registers per (lane, kernel row): every input pixel group is transformed 3x (once per kernel row ky) and
// again by each 32-row M tile, ~225 VALU instructions per 72 MFMAs -> 65 % MFMA busy.  Here:
//   * an N tile is 16 STRIPS of 8 output rows x 4 pixels (one Winograd group wide); a strip reads 10 input rows x 6 pixels.  Strips
//     are enumerated linearly over (image, 8-row band, group), so a tile is not tied to the row width (no LDS-size dependence on
//     cs, no padding of rows to a tile width);
//   * per 8-channel K step the 320 (strip, input row, channel-quad half) pixel groups of the tile are loaded ONCE from HBM/L2 into
//     registers (6 x global_load_dwordx4 per lane, issued a whole step ahead), transformed ONCE (v = B^T d, ~30 packed VALU
//     instructions) and written to an LDS image V[position][half][row][strip]; the three kernel rows and both 32-row M tiles read
//     the same V rows (row r + ky);
//   * the MFMA loop is then VALU-free: per (ky, position) one ds_read_b128 of the weight fragment and one of the V fragment (both
//     conflict-free: V rows are padded to 20 slots so that the 16 lanes of a read group hit 16 different bank quads), 4 MFMAs.
// Weights: the conv_w1d packing [32-row tile][K block][ky*6 + position][lane][4], staged by LDS-DMA, two stages; V: two stages;
// one barrier per K step.  Split-K tail and persistent XCD-aware workgroups as in conv_qp.inc.
#include <vector>

#include "conv_qp.inc"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kR = 8;                         // output rows per strip
constexpr int kS = 16;                        // strips per workgroup tile
constexpr int kRI = kR + 2;                   // input rows per strip
constexpr int kVPlane = kRI * kS * 16;        // bytes per (position, channel-quad half): [row][strip slot], 16-byte slots
constexpr int kVPos = 2 * kVPlane;            // bytes per position
constexpr int kVBytes = 6 * kVPos;            // V image of one K step (30,720 B)
constexpr int kTaps = 18;                     // weight planes per K block: 3 kernel rows x 6 positions
constexpr int kMTB = 2;                       // 32-row M tiles per workgroup
constexpr int kWBytes = kMTB * kTaps * 1024;  // 36,864 B
constexpr int kStage = kWBytes + kVBytes;
constexpr int kStages = 2;
constexpr int kScratch = 2048;                // per wave: epilogue transpose (8 rows x 16 pixels of one channel quad)
constexpr int kLds = kStages * kStage + 8 * kScratch;   // 151,552 B
constexpr int kDmaAt = 4;                     // waves 0..3 issue the next step's weight DMA after this many groups
constexpr int kXfAt = 12;                     // the transform of the next step's pixels sits after this many (ky, position) groups
constexpr int kSlots = 4 * 32 * 4;            // pixel slots of a tile in the split-K scratch: (wave column, lane, pixel)

// LDS-DMA piece with an immediate offset (applied to the global AND the LDS address): 8 pieces of a contiguous run share one
// address register and one M0
template <int OFF> __device__ __forceinline__ void glds16o(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, OFF, 0);
}

// slot of strip s in row r of a V plane: XOR-swizzled so that the 16 lanes of a ds_read_b128 group (4 strips x 4 rows of one wave)
// hit 16 different bank quads without padding the rows
__device__ __forceinline__ int vslot(int r, int s) { return r * kS + (s ^ ((r & 3) << 2)); }

__device__ __forceinline__ void w4_in(const f32x4 *d, f32x4 *v) {
    // B^T d for F(4,3); shared sub-expressions written out so that the compiler emits ~30 packed instructions
    const f32x4 s34 = d[3] + d[4], d43 = d[4] - d[3], s12 = d[1] + d[2], d12 = d[1] - d[2], d42 = d[4] - d[2], d31 = d[3] - d[1];
    v[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    v[1] = s34 - 4.f * s12;
    v[2] = d43 + 4.f * d12;
    v[3] = d42 + 2.f * d31;
    v[4] = d42 - 2.f * d31;
    v[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
__device__ __forceinline__ void w4_out(const float *m, float *y) {
    const float a = m[1] + m[2], b = m[1] - m[2], c = m[3] + m[4], e = m[3] - m[4];
    y[0] = m[0] + a + c;
    y[1] = b + 2.f * e;
    y[2] = a + 4.f * c;
    y[3] = b + 8.f * e + m[5];
}

// diagnostic builds only (DBG & 128): one s_memtime stamp with the scheduler pinned around it
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// decoded strip: image, first output row, first output column
struct Strip { int img, y0, x0; bool ok; };

template <int DBG>   // ablation bits for timing experiments (tools/w2d_ablate.sh); 0 = the production kernel
__global__ __launch_bounds__(512) void conv_w2d(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int j = lane & 31, h = lane >> 5;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int vb = (nwg % 8 == 0) ? (bid % 8) * (nwg / 8) + bid / 8 : bid;

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

    // strip s of N tile nb  (p.Wv = groups per row, p.PV = strips per image, p.Hv / p.wpx = valid output rows / pixels per row)
    const unsigned tot = (unsigned)p.nimg * (unsigned)p.PV;
    auto strip_of = [&](int nb, int s) -> Strip {
        Strip r;
        const unsigned u = (unsigned)nb * kS + s;
        r.ok = u < tot;
        const unsigned uc = r.ok ? u : tot - 1;
        r.img = (int)fdiv(uc, p.fd_PV);
        const int rem = (int)(uc - (unsigned)r.img * p.PV);
        const int band = (int)fdiv((unsigned)rem, p.fd_Wv);
        r.y0 = band * kR;
        r.x0 = 4 * (rem - band * p.Wv);
        return r;
    };

    // ---- fill cursor: the (tile, K chunk) stream one step ahead of the MFMA loop.
    // Who does the memory work: of the two waves of a SIMD the one that starts its MFMAs first wins the matrix pipe, finishes its
    // 72 MFMAs early and then waits ~4,800 cycles at the step's barrier; the other one is the step's critical path (in-kernel
    // stamps).  One half of the waves (p.dbg bit 0: which) loads and transforms the pixels (pass A: 4 x 64 = 256
    // of the 320 lane-transforms, pass B: the other 64 on one more wave), three waves issue the 36 weight pieces by LDS-DMA.
    int f_id = vb, f_c = 0, f_end = 0, f_stage = 0, issued = 0;
    const float *f_w;
    // Roles (tools/w2d_roles.sh tried twelve assignments: all within 3 % of each other): waves 4..7 load + transform the pixels
    // (pass A), wave 0 pass B; waves 0..3 issue the weight DMA, 9 contiguous pieces each.
    const unsigned lane16 = (unsigned)lane * 16u;
    const int kXfHalf = 1;   // which half of the waves transforms (1: waves 4..7 -- measured 5 % faster than the older half); the other half issues the DMA
    const bool xfA = (wave >> 2) == kXfHalf, xfB = wave == 4 * (1 - kXfHalf);
    const int dw = wave - 4 * (1 - kXfHalf);   // DMA wave index 0..3 (negative / >= 4: none)
    int vwOff[2];
    int x_h[2], x_i[2], x_s[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int xt = ps == 0 ? (wave & 3) * 64 + lane : 256 + lane;      // lane-transform id
        x_h[ps] = xt / (kS * kRI);
        x_i[ps] = (xt / kS) % kRI;
        x_s[ps] = xt % kS;
        vwOff[ps] = kWBytes + x_h[ps] * kVPlane + vslot(x_i[ps], x_s[ps]) * 16;   // where this lane writes v[position 0]
    }
    unsigned x_off[2] = {0, 0};   // byte offset of the lane's 6 input pixels inside the two-plane window of a K block (< 2^32)
    int x_keep[2] = {6, 6};       // inputs [0, keep) feed in-row outputs (6 unless the group hangs over the row end)
    auto set_fill_tile = [&](int w) {
        const Item it = decode(w);
        f_c = it.c0;
        f_end = it.c1;
        const int nb = (int)fdiv((unsigned)it.tile, p.fd_ntm), mb = it.tile - nb * p.n_tiles_m;
        f_w = p.wpk + (size_t)mb * kMTB * p.KB * kTaps * 256;     // wave-uniform: the lane's 16 bytes are added as a 32-bit offset
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            if (ps == 0 ? xfA : xfB) {
                const Strip st = strip_of(nb, x_s[ps]);
                x_off[ps] = 16u * ((unsigned)x_h[ps] * (unsigned)p.in_plane + (unsigned)st.img * (unsigned)p.P + (unsigned)(st.y0 + x_i[ps]) * p.Wb + st.x0);
                const int left = p.wpx - st.x0;
                x_keep[ps] = left + 2 < 6 ? left + 2 : 6;
            }
    };
    set_fill_tile(f_id);
    f32x4 dn[2][6];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int k = 0; k < 6; ++k) dn[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The next step's loads, issued at the top of the step: the raw pixels into registers (fill_loads), then the weight DMA
    // (fill_dma) -- in this order: hipcc puts an s_waitcnt vmcnt(0) in front of a register load that follows an LDS-DMA (measured:
    // ~2,000 cycles per step with the other order).  fill_xform, in the middle of the wave's own MFMA stream: transform + publish.
    // (Also measured, and not better: the loads dealt out one per MFMA group; transform + DMA after the wave's MFMAs at raised
    // priority -- DESIGN.md section 4.)
    auto fill_loads = [&]() {
        if (issued >= nsteps || (DBG & 8)) return;
        const char *kbase = (const char *)(p.in + (size_t)(2 * f_c) * p.in_plane);   // scalar; x_off is a 32-bit BYTE offset
        if (xfA) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dn[0][k] = *(const f32x4 *)(kbase + x_off[0] + 16 * k);
        }
        if (xfB) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dn[1][k] = *(const f32x4 *)(kbase + x_off[1] + 16 * k);
        }
    };
    auto fill_xform = [&]() {
        if (issued >= nsteps || (DBG & 16)) return;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            if (ps == 0 ? xfA : xfB) {
                if (__builtin_amdgcn_ballot_w64(x_keep[ps] < 6)) {   // some lane's group hangs over its row end (1 group in ~66)
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 3; k < 6; ++k)
                        if (k >= x_keep[ps]) dn[ps][k] = zero;
                }
                f32x4 v[6];
                w4_in(dn[ps], v);
                char *dst = smem + f_stage * kStage + vwOff[ps];
#pragma unroll
                for (int x = 0; x < 6; ++x) *(f32x4 *)(dst + x * kVPos) = v[x];
            }
    };
    auto fill_dma = [&]() {
        if (issued >= nsteps || (DBG & 4)) return;
        if (dw >= 0 && dw < 4) {
            // stage image of the weights = [M tile mt][18 planes] of 1 KiB pieces; wave w copies planes [9 * (w & 1), + 9) of M tile
            // w >> 1 of K block f_c: source and destination are both contiguous, so the 9 pieces need two address registers (one
            // 64-bit add each) and immediate offsets -- with a 64-bit pointer computed per piece every piece cost ~4 VALU-slot
            // instructions, each of which waits for a 64-cycle MFMA slot of the SIMD partner (12 pieces took 5,200 cycles)
            const int mt = dw >> 1, pl0 = (dw & 1) * 9;
            const char *src = (const char *)(f_w + (((size_t)mt * p.KB + (size_t)f_c) * kTaps + pl0) * 256) + lane16;
            char *dst = smem + f_stage * kStage + (mt * kTaps + pl0) * 1024;
            glds16o<-4096>(src + 4096, dst + 4096);
            glds16o<-3072>(src + 4096, dst + 4096);
            glds16o<-2048>(src + 4096, dst + 4096);
            glds16o<-1024>(src + 4096, dst + 4096);
            glds16o<0>(src + 4096, dst + 4096);
            glds16o<1024>(src + 4096, dst + 4096);
            glds16o<2048>(src + 4096, dst + 4096);
            glds16o<3072>(src + 4096, dst + 4096);
            glds16o<0>(src + 8192, dst + 8192);
        }
    };
    auto fill_advance = [&]() {
        if (issued >= nsteps) return;
        ++issued;
        f_stage ^= 1;
        if (++f_c == f_end) {
            f_id += nwg;
            if (f_id < nitems) set_fill_tile(f_id);
        }
    };

    // ---- fragments: lane (j, h) of wave (wm, wn) owns output row j % 8 of strip 4 * wn + j / 8
    const int sl = 4 * wn + (j >> 3), rr = j & 7;
    const int aOff = (wm * kTaps) * 1024 + lane * 16;
    // V fragment of (position x, kernel row ky): plane h, row rr + ky, strip sl; the XOR swizzle depends on (rr + ky) & 3
    int bOffK[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) bOffK[ky] = kWBytes + h * kVPlane + vslot(rr + ky, sl) * 16;

    f32x16 acc[6];
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    const float slope = p.act == ND_ACT_NONE ? 1.f : (p.slope_dev ? *p.slope_dev : p.slope);
    const bool slope01 = p.act <= ND_ACT_PRELU && slope >= 0.f && slope <= 1.f;   // PReLU(t) == max(t, slope * t)

    // combine the position accumulators of channel group g into the 4 pixels' sums (clears them)
    auto combine = [&](int g, f32x4 *y) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m[6], o[4];
#pragma unroll
            for (int x = 0; x < 6; ++x) {
                m[x] = acc[x][4 * g + e];
                acc[x][4 * g + e] = 0.f;
            }
            w4_out(m, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i][e] = o[i];
        }
    };

    // Epilogue of a finished tile.  A lane's 4 pixels x 4 channels go through a 2 KiB per-wave LDS scratch, one channel quad
    // at a time ([8 rows][16 pixels] of 16 bytes), so that the stores leave as 256-byte row segments (16 lanes x 16 B contiguous,
    // 4 segments per wave instruction) instead of 64 separate 16-byte pieces at a 64-byte stride.  Every VALU instruction here
    // is matrix-pipe time (see the header), so: packed fp32 arithmetic on adjacent accumulator registers, the bias in registers
    // (loaded when the tile starts), PReLU as max(t, slope * t) when 0 <= slope <= 1, store offsets computed once per tile.
    char *const scratch = smem + kStages * kStage + wave * kScratch;
    const int e_row = lane >> 4, e_px = lane & 15;                  // reader side: rows e_row and e_row + 4, pixel e_px of the wave's 16
    f32x4 biasq[4];                                                 // bias of channels 8g + 4h .. + 3 of this wave's M tile
    auto load_bias = [&](int id) {
        const int mb = id - (int)fdiv((unsigned)id, p.fd_ntm) * p.n_tiles_m;
#pragma unroll
        for (int g = 0; g < 4; ++g) biasq[g] = *(const f32x4 *)(p.bias + (mb * kMTB + wm) * 32 + 8 * g + 4 * h);
    };
    load_bias(decode(vb).tile);
    auto epilogue = [&](int id, auto fast) {
        const int nb = (int)fdiv((unsigned)id, p.fd_ntm), mb = id - nb * p.n_tiles_m;
        const Strip st = strip_of(nb, 4 * wn + (e_px >> 2));        // the strip this lane STORES for
        const int ex = st.x0 + (e_px & 3);
        const unsigned offA = (unsigned)((st.img * p.Po + (st.y0 + e_row + p.opad) * p.Wo + ex + p.opad) * 16);   // bytes inside a plane (< 2^32)
        const unsigned offB = offA + (unsigned)(4 * p.Wo * 16);
        const bool okx = st.ok && ex < p.wpx;
        const bool okA = okx && st.y0 + e_row < p.Hv, okB = okx && st.y0 + e_row + 4 < p.Hv;
        const int q0 = (mb * kMTB + wm) * 8;                        // first channel quad of this wave's M tile
        const f32x2 slope2 = {slope, slope};
        // fused MaxPool2d(2): the lanes with (lane & 2) == 0 each own one pooled pixel of the wave's 8 rows x 16 pixels -- pooled
        // row e_row, pooled pixel e_px & 1 of the strip the lane already decoded (strips start on rows = 0 mod 8, pixels = 0 mod 4)
        const int pp_i = (e_px & ~3) + 2 * (e_px & 1);              // scratch column of the 2x2 block's first pixel
        const int pxp = (st.x0 >> 1) + (e_px & 1), pyp = (st.y0 >> 1) + e_row;
        const bool okP = p.pool && !(lane & 2) && st.ok && 2 * pxp + 1 < p.wpx && 2 * pyp + 1 < p.Hv;
        const unsigned offP = (unsigned)((st.img * p.pool_P + (pyp + p.pool_pad) * p.pool_W + pxp + p.pool_pad) * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 yy[4], ypre[4];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x2 m[6];
#pragma unroll
                for (int x = 0; x < 6; ++x) {
                    m[x] = f32x2{acc[x][4 * g + 2 * pr], acc[x][4 * g + 2 * pr + 1]};
                    acc[x][4 * g + 2 * pr] = 0.f;
                    acc[x][4 * g + 2 * pr + 1] = 0.f;
                }
                const f32x2 bq = {biasq[g][2 * pr], biasq[g][2 * pr + 1]};
                const f32x2 a = m[1] + m[2] + bq, b = m[1] - m[2] + bq, c = m[3] + m[4], e = m[3] - m[4];   // A^T m, bias folded in once
                f32x2 y[4];
                y[0] = m[0] + a + c;
                y[1] = b + 2.f * e;
                y[2] = a + 4.f * c;
                y[3] = b + 8.f * e + m[5];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ypre[i][2 * pr] = y[i][0];
                    ypre[i][2 * pr + 1] = y[i][1];
                    if constexpr (decltype(fast)::value) {
                        const f32x2 t = y[i] * slope2;
                        y[i] = f32x2{fmaxf(y[i][0], t[0]), fmaxf(y[i][1], t[1])};
                    } else {
                        y[i] = f32x2{apply_act(y[i][0], p.act, slope), apply_act(y[i][1], p.act, slope)};
                    }
                    yy[i][2 * pr] = y[i][0];
                    yy[i][2 * pr + 1] = y[i][1];
                }
            }
            if (p.pre) {   // training forward: acc + bias for the activation's backward pass, compact planes [C/4][B][Hv][wpx]
                const Strip so = strip_of(nb, sl);
                const int yo = so.y0 + rr, m4 = (q0 + 2 * g + h) * 4, left = p.wpx - so.x0;
                if (so.ok && yo < p.Hv && m4 < p.M) {
                    f32x4 *dst = p.pre + (long)(m4 >> 2) * p.pre_plane + ((long)so.img * p.Hv + yo) * p.wpx + so.x0;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < left) dst[i] = ypre[i];
                }
            }
            if (DBG & 256) {   // diagnostic: the untransposed stores (each lane its own 4 pixels, 16-byte pieces at a 64-byte stride)
                const Strip so = strip_of(nb, sl);
                const int yo = so.y0 + rr, m4 = (q0 + 2 * g + h) * 4, left = p.wpx - so.x0;
                if (so.ok && yo < p.Hv && m4 < p.M) {
                    f32x4 *dst = p.out + (long)(p.out_plane0 + (m4 >> 2)) * p.out_plane + (long)so.img * p.Po + (long)(yo + p.opad) * p.Wo + so.x0 + p.opad;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < left) dst[i] = yy[i];
                }
                continue;
            }
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                if (h == hq) {
                    f32x4 *w = (f32x4 *)scratch + rr * 16 + (j >> 3) * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = yy[i];
                }
                // lanes exchange data through LDS inside one wave: the hardware completes a wave's LDS operations in order, but the
                // compiler must be told that other lanes wrote this memory (without the fences it reuses the values a lane read in
                // the previous pass for every lane that did not store itself)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const f32x4 va = *((const f32x4 *)scratch + e_row * 16 + e_px);
                const f32x4 vb2 = *((const f32x4 *)scratch + (e_row + 4) * 16 + e_px);
                f32x4 vp = va;
                if (p.pool) {   // wave-uniform
                    const f32x4 *blk = (const f32x4 *)scratch + (2 * e_row) * 16 + pp_i;
                    const f32x4 p00 = blk[0], p01 = blk[1], p10 = blk[16], p11 = blk[17];
#pragma unroll
                    for (int e = 0; e < 4; ++e) vp[e] = fmaxf(fmaxf(p00[e], p01[e]), fmaxf(p10[e], p11[e]));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int quad = q0 + 2 * g + hq;
                if (quad * 4 < p.M && !(DBG & 1)) {
                    char *pb = (char *)(p.out + (size_t)(p.out_plane0 + quad) * p.out_plane);   // wave-uniform plane base
                    if (okA) *(f32x4 *)(pb + offA) = va;
                    if (okB) *(f32x4 *)(pb + offB) = vb2;
                    if (okP) *(f32x4 *)((char *)(p.pool + (size_t)quad * p.pool_plane) + offP) = vp;
                }
            }
        }
    };
    // a K slice of a split tile: combined raw sums, tile-local layout [channel quad][pixel slot = 4 * (32 * wn + j) + i]
    auto epilogue_partial = [&](int slice) {
        f32x4 *dst = p.part + (size_t)slice * (kMTB * 8) * kSlots;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 yy[4];
            combine(g, yy);
            f32x4 *d2 = dst + (size_t)(wm * 8 + 2 * g + h) * kSlots + 4 * (wn * 32 + j);
#pragma unroll
            for (int i = 0; i < 4; ++i) d2[i] = yy[i];
        }
    };

    // ---- prologue: step 0's weights and V image
    fill_loads();
    fill_xform();
    fill_dma();
    fill_advance();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int c_id = vb, c_stage = 0;
    Item c_it = decode(vb);
    int c_c = c_it.c0;
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;   // DBG & 128: cycles per phase, summed over the steps
    for (int s = 0; s < nsteps; ++s) {
        if (DBG & 128) t0 = stamp();
        fill_loads();                       // next step's pixels fly under this step's MFMAs (their weights: kDmaAt)
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 128) { t1 = stamp(); tph[0] += t1 - t0; t0 = t1; }
        const char *wa = smem + c_stage * kStage + aOff;
        const char *vq = smem + c_stage * kStage;
        f32x4 a[2], b[2];
        a[0] = *(const f32x4 *)wa;
        b[0] = *(const f32x4 *)(vq + bOffK[0]);
#pragma unroll
        for (int st = 0; st < kTaps; ++st) {
            if (st + 1 < kTaps) {
                const int ky = (st + 1) / 6, x = (st + 1) % 6;
                a[(st + 1) & 1] = *(const f32x4 *)(wa + (st + 1) * 1024);
                b[(st + 1) & 1] = *(const f32x4 *)(vq + bOffK[ky] + x * kVPos);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int x = st % 6;
            if (!(DBG & 32)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][q], b[st & 1][q], acc[x], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[x][q] += a[st & 1][q] * b[st & 1][q];
            }
            if (st + 1 == kDmaAt) {
                // the weight DMA of the next step goes out here, not at the top of the step: the issuing waves get one issue slot
                // per MFMA of their SIMD partners, so the step's 36 pieces take thousands of cycles to issue -- the pixel loads, which
                // gate the transforming waves, must not queue behind them
                __builtin_amdgcn_sched_barrier(0);
                fill_dma();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (st + 1 == kXfAt) {
                // the raw pixels of the next step were requested at the top of this step: transform them here, inside this wave's
                // own MFMA stream (the SIMD partner's MFMAs fill the matrix pipe meanwhile), and publish them in the OTHER stage
                __builtin_amdgcn_sched_barrier(0);
                if (DBG & 128) { t1 = stamp(); tph[1] += t1 - t0; t0 = t1; }
                fill_xform();
                fill_advance();
                if (DBG & 128) { t1 = stamp(); tph[3] += t1 - t0; t0 = t1; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DBG & 128) { t1 = stamp(); tph[1] += t1 - t0; t0 = t1; }
        // next step's weights (DMA) have landed and this wave's V rows are written.  This is the only full vmcnt wait of the step
        // and it comes BEFORE the epilogue: the stores of a finished tile are never waited for here -- they drain under the next
        // step's MFMAs and are a step old when the next wait comes
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (DBG & 128) { t1 = stamp(); tph[2] += t1 - t0; t0 = t1; }
        c_stage ^= 1;
        if (++c_c == c_it.c1) {
            if (DBG & 2) {
            } else if (c_it.slice >= 0)
                epilogue_partial(c_it.slice);
            else if (slope01)
                epilogue(c_it.tile, std::true_type{});
            else
                epilogue(c_it.tile, std::false_type{});
            c_id += nwg;
            if (c_id < nitems) {
                c_it = decode(c_id);
                if (p.n_tiles_m > 1) load_bias(c_it.tile);   // (one M tile: the registers already hold it)
            }
            c_c = c_it.c0;
        }
        if (DBG & 128) { t1 = stamp(); tph[4] += t1 - t0; t0 = t1; }
        if (!(DBG & 64)) __builtin_amdgcn_s_barrier();
        if (DBG & 128) { t1 = stamp(); tph[5] += t1 - t0; }
    }
    if ((DBG & 128) && lane == 0 && p.part) {
        // stamps go to a buffer of their own (the launcher passes it in p.part in this build): [workgroup][wave][6 phases + steps]
        unsigned long long *o = (unsigned long long *)p.part + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int k = 0; k < 6; ++k) o[k] = tph[k];
        o[6] = (unsigned long long)nsteps;
    }
}

// adds the K slices of a split tile in slice order and applies bias / activation like the epilogue.  grid (split tiles, 16 quads)
__global__ __launch_bounds__(256) void k_w2d_split_finish(ConvParams p) {
    const int t = blockIdx.x, quad = blockIdx.y;
    const int gid = p.split_first + t;
    const int nb = gid / p.n_tiles_m, mb = gid - nb * p.n_tiles_m;
    const int m4 = mb * (kMTB * 32) + quad * 4;
    if (m4 >= p.M) return;
    const float slope = p.act == ND_ACT_NONE ? 1.f : (p.slope_dev ? *p.slope_dev : p.slope);
    const f32x4 bv = *(const f32x4 *)(p.bias + m4);
    const unsigned tot = (unsigned)p.nimg * (unsigned)p.PV;
    for (int l = threadIdx.x; l < kSlots; l += 256) {
        const int i = l & 3, jj = l >> 2, wn = jj >> 5, j = jj & 31;
        const int s = 4 * wn + (j >> 3), rr = j & 7;
        const unsigned u = (unsigned)nb * kS + s;
        if (u >= tot) continue;
        const int img = (int)(u / (unsigned)p.PV);
        const int rem = (int)(u - (unsigned)img * p.PV);
        const int band = rem / p.Wv;
        const int y = band * kR + rr, x = 4 * (rem - band * p.Wv) + i;
        if (y >= p.Hv || x >= p.wpx) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < p.S; ++ks) acc += p.part[((size_t)(t * p.S + ks) * (kMTB * 8) + quad) * kSlots + l];
        acc += bv;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p.act <= ND_ACT_PRELU ? (acc[e] > 0.f ? acc[e] : acc[e] * slope) : apply_act(acc[e], p.act, slope);
        p.out[(long)(p.out_plane0 + (m4 >> 2)) * p.out_plane + (long)img * p.Po + (long)(y + p.opad) * p.Wo + x + p.opad] = v;
    }
}

}  // namespace

bool nd_w2d_ok(const QpBuf &in) { return in.dt == ND_F32 && in.Hb >= 3 && in.Wb >= 3; }
// workgroup tiles of a whole layer (16 strips of 8 rows x 4 pixels x 64 output channels each)
long nd_w2d_tiles(const QpBuf &in, int cout) {
    const long strips = (long)in.B * ((in.Hb - 2 + kR - 1) / kR) * ((in.Wb - 2 + 3) / 4);
    return ((strips + kS - 1) / kS) * ((cout + kMTB * 32 - 1) / (kMTB * 32));
}

// d: the layer as for nd_launch_conv (CONV3 / CONVT3, fp32); d.wpk = nd_w1d_pack blob with T = 4 (the same packing as conv_w1d)
int nd_launch_conv_w2d(const ConvDesc &d, hipStream_t stream) {
    if ((d.kind != ND_CONV3 && d.kind != ND_CONVT3) || d.in.dt != ND_F32 || d.out.dt != ND_F32) ND_FAIL(ND_EINVAL, "w2d: fp32 3x3 layers only");
    const int KB = nd_kblocks(d.cin);
    const int Hfull = d.in.Hb - 2, Wfull = d.in.Wb - 2;
    if (Hfull < 1 || Wfull < 1) ND_FAIL(ND_EINVAL, "w2d: input smaller than the kernel");
    const bool roi = d.roi_rows > 0;
    if (roi && (d.roi_r0 < 0 || d.roi_c0 < 0 || d.roi_cols < 1 || d.roi_r0 + d.roi_rows > Hfull || d.roi_c0 + d.roi_cols > Wfull || d.pool))
        ND_FAIL(ND_EINVAL, "w2d: region [%d,+%d) x [%d,+%d) outside the %d x %d output (or a pooled layer)", d.roi_r0, d.roi_rows, d.roi_c0, d.roi_cols, Hfull, Wfull);
    // (a region is the same launch on shifted base pointers: the kernel knows row / image strides and valid extents separately)
    const int Hv = roi ? d.roi_rows : Hfull, Wpx = roi ? d.roi_cols : Wfull, Wg = (Wpx + 3) / 4, NB = (Hv + kR - 1) / kR;
    const long roi_in = roi ? (long)d.roi_r0 * d.in.Wb + d.roi_c0 : 0, roi_out = roi ? (long)d.roi_r0 * d.out.Wb + d.roi_c0 : 0;
    if (d.pre && (d.roi_rows > 0 || d.pool)) ND_FAIL(ND_EINVAL, "w2d: a pre-activation copy goes with whole, unpooled layers only");
    if (d.cout % 4) ND_FAIL(ND_EINVAL, "w2d: cout must be a multiple of 4");
    if (d.in.planes < d.in_plane0 + 2 * KB) ND_FAIL(ND_EINVAL, "w2d: input buffer has %d planes, needs %d", d.in.planes, d.in_plane0 + 2 * KB);
    if (d.out.Hb != Hfull + 2 * d.out.pad || d.out.Wb != Wfull + 2 * d.out.pad || d.out.B != d.in.B) ND_FAIL(ND_EINVAL, "w2d: destination does not fit the result");
    if (d.out_plane0 + d.cout / 4 > d.out.planes) ND_FAIL(ND_EINVAL, "w2d: destination planes overflow");
    // a lane's float4 index inside a K block's two-plane window is 32 bits; strips of the last band / group read up to 9 rows + 5
    // pixels past the last valid pixel (the buffers carry that slack)
    if (2 * d.in.np() * 16 + 65536 >= (1L << 32) || d.in.used() >= (1L << 31)) ND_FAIL(ND_EINVAL, "w2d: input too large for 32-bit byte offsets");

    static std::atomic<int> lds_set[16];
    int dev = 0, ncus = 0;
    ND_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) ND_FAIL(ND_EINVAL, "w2d: device index %d", dev);
    ND_TRY(nd_num_cus(dev, &ncus));
    const int lds = kLds;
    void (*fn)(ConvParams) = conv_w2d<0>;
#ifdef ND_QP_STAMPS
    // diagnostic build only (make STAMPS=1; tools/w2d_ablate.sh): ND_W2D_DBG names one of the ablation masks / the stamped kernel
    static const int dbg_env = getenv("ND_W2D_DBG") ? atoi(getenv("ND_W2D_DBG")) : 0;
    switch (dbg_env) {
        case 1: fn = conv_w2d<1>; break;
        case 2: fn = conv_w2d<2>; break;
        case 4: fn = conv_w2d<4>; break;
        case 8: fn = conv_w2d<8>; break;
        case 16: fn = conv_w2d<16>; break;
        case 24: fn = conv_w2d<24>; break;
        case 28: fn = conv_w2d<28>; break;
        case 30: fn = conv_w2d<30>; break;
        case 32: fn = conv_w2d<32>; break;
        case 62: fn = conv_w2d<62>; break;
        case 128: fn = conv_w2d<128>; break;
        case 256: fn = conv_w2d<256>; break;
        default: break;
    }
#else
    constexpr int dbg_env = 0;
#endif
    if (lds_set[dev].load(std::memory_order_relaxed) != dbg_env + 1) {
        ND_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        lds_set[dev].store(dbg_env + 1, std::memory_order_relaxed);
    }

    ConvParams p = {};
    p.in = (const f32x4 *)d.in.base + (long)d.in_plane0 * d.in.np() + roi_in;
    p.wpk = d.wpk;
    p.bias = d.bias;
    p.out = (f32x4 *)d.out.base + roi_out;
    p.in_plane = d.in.np();
    p.out_plane = d.out.np();
    p.nimg = d.in.B;
    p.P = d.in.Hb * d.in.Wb;
    p.Wb = d.in.Wb;
    p.Hv = Hv;
    p.Wv = Wg;               // groups per row
    p.PV = NB * Wg;          // strips per image
    p.wpx = Wpx;             // valid pixels per row
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
    if (d.pool) {
        const QpBuf &q = *d.pool;
        if (q.dt != ND_F32 || q.B != d.in.B || q.Hb - 2 * q.pad != Hv / 2 || q.Wb - 2 * q.pad != Wpx / 2 || q.planes < d.cout / 4)
            ND_FAIL(ND_EINVAL, "w2d: pooled destination does not fit %dx%dx%d", d.cout, Hv / 2, Wpx / 2);
        if (q.used() * 16 >= (1L << 32)) ND_FAIL(ND_EINVAL, "w2d: pooled destination too large for 32-bit byte offsets");
        p.pool = (f32x4 *)q.base;
        p.pool_plane = q.np();
        p.pool_P = q.Hb * q.Wb;
        p.pool_W = q.Wb;
        p.pool_pad = q.pad;
    }
    p.n_tiles_n = (int)(((long)p.nimg * p.PV + kS - 1) / kS);
    p.n_tiles_m = (d.cout + kMTB * 32 - 1) / (kMTB * 32);
    p.tiles_per_problem = p.n_tiles_n * p.n_tiles_m;
    nd_conv_fastdivs(p);
    const long ntiles = p.tiles_per_problem;
    const long slots = ncus;
    // (a layer with a fused pool or a pre-activation copy keeps every tile whole: the split-K finish kernel has no view of a tile's
    //  2x2 neighbours and writes no copy -- the training step sends layers with few tiles through conv_w1d, which splits)
    const long cap = d.part && !d.nosplit && !d.pool && !d.pre ? (long)(d.part_bytes / ((size_t)kMTB * 32 * kSlots * 4)) : 0;
    int first, S, cps;
    nd_plan_split(ntiles, KB, slots, cap, &first, &S, &cps);
    p.split_first = first;
    p.S = S;
    p.cps = cps;
    p.nitems = (int)(first + (ntiles - first) * S);
    p.part = (f32x4 *)d.part;
    const long grid = p.nitems < slots ? p.nitems : slots;
#ifdef ND_QP_STAMPS
    if (dbg_env == 128) {
        // stamped diagnostic launch: no split-K (p.part carries the stamp buffer), synchronous, prints the phase split per wave role
        static unsigned long long *buf = nullptr;
        const size_t n = (size_t)slots * 8 * 8;
        if (!buf) ND_HIP(hipMalloc(&buf, n * 8));
        ND_HIP(hipMemsetAsync(buf, 0, n * 8, stream));
        p.split_first = (int)ntiles;
        p.S = 1;
        p.cps = KB;
        p.nitems = (int)ntiles;
        p.part = (f32x4 *)buf;
        const long g2 = ntiles < slots ? ntiles : slots;
        hipLaunchKernelGGL(fn, dim3((unsigned)g2), dim3(512), lds, stream, p);
        ND_HIP(hipStreamSynchronize(stream));
        static int printed = 0;
        if (printed++ < 2) {
            std::vector<unsigned long long> h(n);
            ND_HIP(hipMemcpy(h.data(), buf, n * 8, hipMemcpyDeviceToHost));
            const char *names[6] = {"issue loads + DMA", "MFMA loop", "wait vmcnt", "transform+publish (inside the loop)", "epilogue", "barrier"};
            for (int w : {0, 1, 4}) {
                double tot[6] = {0}, steps = 0;
                for (long b = 0; b < g2; ++b) {
                    for (int k = 0; k < 6; ++k) tot[k] += (double)h[((size_t)b * 8 + w) * 8 + k];
                    steps += (double)h[((size_t)b * 8 + w) * 8 + 6];
                }
                double sum = 0;
                for (int k = 0; k < 6; ++k) sum += tot[k];
                fprintf(stderr, "[w2d stamps] wave %d: %.0f cycles/step:", w, sum / steps);
                for (int k = 0; k < 6; ++k) fprintf(stderr, "  %s %.0f (%.1f%%)", names[k], tot[k] / steps, 100 * tot[k] / sum);
                fprintf(stderr, "\n");
            }
        }
        return ND_OK;
    }
#endif
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(512), lds, stream, p);
    if (first < ntiles) hipLaunchKernelGGL(k_w2d_split_finish, dim3((unsigned)(ntiles - first), kMTB * 8), dim3(256), 0, stream, p);
    ND_HIP(hipGetLastError());
    return ND_OK;
}
