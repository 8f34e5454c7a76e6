// Native RAFT executor: owns the packed weights on the device and sequences the HIP kernels of one
// `RAFT.forward(test_mode=True)` (RAFT/core/raft.py:86-144) for a batch of frame pairs.
//
// The reference drives ~60 PyTorch ops per refinement iteration from Python; here the whole forward is
// one C call that enqueues ~15 launches per iteration on the caller's stream, with every activation in
// a caller-provided HBM workspace (NHWC fp32):
//
//   encoders   preprocess -> 7x7/s2 conv -> 3 stages x 2 residual blocks -> 1x1 conv
//              fnet: instance-norm statistics kernels + norm/ReLU fused into the next conv's A-load
//              cnet: BatchNorm folded into the conv epilogue (scale/shift), residual add in the epilogue
//   volume     batched fp32-MFMA GEMM + one pooling kernel (4-level pyramid stays resident in HBM)
//   iteration  lookup -> convc1 -> convc2 -> convf1 -> convf2 -> conv -> [zr, q] x2 -> flow head (x2)
//              cat([h, x]) / cat([r*h, x]) / cat([cor, flo]) are channel slices of shared buffers;
//              GRU gates, h update and coords1 += delta live in conv epilogues
//   tail       mask head only on the last iteration (the reference evaluates it on every iteration and
//              discards 19 of 20 results, raft.py:128-139) -> convex upsample -> flow f32[B,H,W,2]
//
// Frame<->keyframe batches share one image (OFX_RAFT_SHARED_IMG2): its feature map is computed once
// and broadcast through a zero batch stride of the correlation GEMM.
#include "ofx_internal.h"

#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

struct ConvW {
    float* w = nullptr;       // [Cout][Kpad]
    float* wsplit = nullptr;  // the same matrix pre-split into bf16 (hi, lo) quads for the bf16x3 mode
    float* wsplit3 = nullptr; // ... and into (hi, mid | lo) for the bf16x6 mode (ofx_split_conv_weight3: 1.5 x the floats)
    float* scale = nullptr;   // [Cout] or null
    float* shift = nullptr;   // [Cout] or null
    int cout = 0, cin = 0, cin_pad = 0, kh = 0, kw = 0;
    long kpad = 0;
    std::string name;         // layer label for the per-layer profile (ofx_prof_enable(2))
};

struct HostTensor {
    const float* data;
    int ndim;
    long shape[4];
    long numel() const {
        long n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        return n;
    }
};

constexpr int HD = 128;      // hidden dim (raft.py:38)
constexpr int CD = 128;      // context dim (raft.py:39)
constexpr int FD = 256;      // feature dim (raft.py:54)
constexpr int LEVELS = 4, RADIUS = 4, CORR_CH = LEVELS * (2 * RADIUS + 1) * (2 * RADIUS + 1);   // 324
// row of the lookup's output = convc1's operand: 324 correlation features + 12 zeros.  336 = 21 whole 16-wide K chunks, so every
// chunk of the 1x1 convolution lies inside the row (the scalar-coordinate schedule of conv.hip applies: 116 against 111 TF at the
// same 21 chunks, tools/experiments/README.md) and rows start on 64-byte boundaries; the lookup writes the zeros itself
constexpr int CORR_LD = 336;
// hx row = [h(128) | motion(126) flow(2) | inp(128)] = 384 floats.  The recurrent part (h, motion) is
// contiguous so the per-iteration GRU convolutions read 256 channels; `inp` (the context features) is
// loop-invariant: its contribution to the six GRU convolutions (+ their biases) is computed ONCE per
// forward into `gadd` and enters the per-iteration convolutions as an epilogue addend -- one third of
// the GRU FLOPs leaves the 20-iteration loop.
constexpr int HX_LD = HD + 128 + CD;
constexpr int MOT_OFF = HD;
constexpr int FLOW_OFF = MOT_OFF + 126;
constexpr int INP_OFF = HD + 128;
constexpr int GADD_LD = 2 * (2 * HD + HD);   // [zr1(256) | q1(128) | zr2(256) | q2(128)]
constexpr int FROW = 16;         // floats per flow row: 7 row neighbours x (fx, fy) + 2 zeros -- convf1's operand (flow_head.hip)
constexpr int ENC_CHUNK = 64;          // images per encoder pass (bounds the activation workspace)
// ... fewer for large frames: the widest encoder activation (64 channels at half resolution) must stay
// below the 2 GiB reach of the convolution kernel's 32-bit byte offsets
static inline int enc_chunk(int H, int W) {
    const long per_image = (long)(H / 2) * (W / 2) * 64 * 4;
    const long fit = ((1L << 31) - 4096) / per_image;
    return (int)std::max<long>(1, std::min<long>(ENC_CHUNK, fit));
}

// A single pair at 512x768 gives the convolution kernels 100-400 workgroups per launch on a 256-CU part: the
// machine is under-filled and each launch is latency-bound.  Below this many 1/8-resolution pixels the
// executor therefore runs the independent chains of a forward side by side on separate HIP streams (the
// three encoders; the flow branch of the motion encoder next to the correlation branch).  Large batches fill
// the machine with every launch and stay on one stream.
static inline bool overlap_pays(long M) {
    // (round 6 sweep at 512x768, M = 6144 per frame: four frames 23.9 ms with the side streams / 24.4 without, five 30.9 / 30.3, six
    // 38.0 / 37.7; OFX_OVERLAP_MAX_M overrides)
    static const char* e = getenv("OFX_OVERLAP_MAX_M");
    static const long lim = e ? atol(e) : 28672;
    return M <= lim;
}

struct Carver {
    char* base;
    size_t off = 0, cap;
    Carver(void* b, size_t c) : base((char*)b), cap(c) {}
    float* take(size_t nfloats) {
        size_t bytes = ((nfloats * sizeof(float) + 255) / 256) * 256;
        char* p = base ? base + off : nullptr;
        off += bytes;
        return (float*)p;
    }
};

}  // namespace

struct ofx_raft {
    std::map<std::string, ConvW> convs;
    // gamma / beta of the context encoder's BatchNorm layers for the batch-statistics mode (OFX_RAFT_BN_BATCH), by norm name
    std::map<std::string, std::pair<float*, float*>> affine;
    std::vector<void*> allocs;
    std::map<std::string, std::pair<void*, size_t>> bufs;
    // side streams for the small-batch schedule (see overlap_pays): independent chains of a forward run
    // concurrently and are joined back into the caller's stream with events
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
};

namespace {

int upload(ofx_raft* r, const std::vector<float>& h, float** d) {
    OFX_HIP_CHECK(hipMalloc((void**)d, h.size() * sizeof(float)));
    r->allocs.push_back(*d);
    OFX_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// both operand formats of a packed weight matrix: fp32, and the bf16x3 pre-split copy
int upload_weight(ofx_raft* r, const std::vector<float>& packed, ConvW* c) {
    int st = upload(r, packed, &c->w);
    if (st) return st;
    std::vector<float> sp(packed.size());
    st = ofx_split_conv_weight(packed.data(), (long)packed.size(), sp.data());
    if (st) return st;
    st = upload(r, sp, &c->wsplit);
    if (st) return st;
    std::vector<float> sp3(packed.size() * 3 / 2);
    st = ofx_split_conv_weight3(packed.data(), (long)packed.size(), sp3.data());
    if (st) return st;
    return upload(r, sp3, &c->wsplit3);
}

const HostTensor* find(const std::map<std::string, HostTensor>& sd, const std::string& k) {
    auto it = sd.find(k);
    return it == sd.end() ? nullptr : &it->second;
}

// pack one conv (optionally with an eval-mode BatchNorm folded into scale/shift, optionally an
// extra output scale as in `.25 * self.mask(net)`, update.py:135)
int add_conv(ofx_raft* r, const std::map<std::string, HostTensor>& sd, const std::string& name,
             const std::string& store_as, int cin_pad, const std::string& bn, float out_scale,
             std::vector<float>* append_w = nullptr, std::vector<float>* append_shift = nullptr,
             const std::vector<int>* chan_sel = nullptr, bool with_bias = true) {
    const HostTensor* w = find(sd, name + ".weight");
    const HostTensor* b = find(sd, name + ".bias");
    if (!w || !b || w->ndim != 4 || b->ndim != 1 || b->shape[0] != w->shape[0]) return OFX_EKEY;
    ConvW c;
    c.cout = (int)w->shape[0];
    c.cin = (int)w->shape[1];
    c.kh = (int)w->shape[2];
    c.kw = (int)w->shape[3];
    const float* wdata = w->data;
    std::vector<float> wsel;
    if (chan_sel) {   // keep a subset of the input channels, in the given order
        const int full = c.cin, taps = c.kh * c.kw;
        c.cin = (int)chan_sel->size();
        wsel.resize((size_t)c.cout * c.cin * taps);
        for (int o = 0; o < c.cout; ++o)
            for (int ci = 0; ci < c.cin; ++ci) {
                const int src = (*chan_sel)[ci];
                if (src < 0 || src >= full) return OFX_EKEY;
                for (int t = 0; t < taps; ++t)
                    wsel[((size_t)o * c.cin + ci) * taps + t] = w->data[((size_t)o * full + src) * taps + t];
            }
        wdata = wsel.data();
    }
    c.cin_pad = cin_pad > 0 ? cin_pad : ((c.cin + 3) / 4) * 4;
    if (c.cin_pad < c.cin) return OFX_EKEY;
    c.kpad = ofx_pack_conv_weight(nullptr, c.cout, c.cin, c.kh, c.kw, c.cin_pad, nullptr);
    if (c.kpad < 0) return (int)c.kpad;
    std::vector<float> pw((size_t)c.cout * c.kpad);
    ofx_pack_conv_weight(wdata, c.cout, c.cin, c.kh, c.kw, c.cin_pad, pw.data());
    std::vector<float> zero_bias;
    if (!with_bias) zero_bias.assign(c.cout, 0.f);
    HostTensor bz;
    if (!with_bias) {
        bz.data = zero_bias.data();
        bz.ndim = 1;
        bz.shape[0] = c.cout;
        b = &bz;
    }
    std::vector<float> scale, shift(c.cout);
    if (!bn.empty()) {
        const HostTensor* g = find(sd, bn + ".weight");
        const HostTensor* be = find(sd, bn + ".bias");
        const HostTensor* rm = find(sd, bn + ".running_mean");
        const HostTensor* rv = find(sd, bn + ".running_var");
        if (!g || !be || !rm || !rv) return OFX_EKEY;
        if (g->numel() != c.cout || be->numel() != c.cout || rm->numel() != c.cout || rv->numel() != c.cout) return OFX_EKEY;
        scale.resize(c.cout);
        for (int i = 0; i < c.cout; ++i) {
            // F.batch_norm(eval): (x - mean) / sqrt(var + eps) * gamma + beta with x = conv + bias
            const double sc = (double)g->data[i] / std::sqrt((double)rv->data[i] + 1e-5);
            scale[i] = (float)sc;
            shift[i] = (float)(((double)b->data[i] - (double)rm->data[i]) * sc + (double)be->data[i]);
        }
    } else if (out_scale != 1.0f) {
        scale.assign(c.cout, out_scale);
        for (int i = 0; i < c.cout; ++i) shift[i] = b->data[i] * out_scale;
    } else {
        for (int i = 0; i < c.cout; ++i) shift[i] = b->data[i];
    }
    if (append_w) {   // caller concatenates several convs along Cout (GRU z|r)
        append_w->insert(append_w->end(), pw.begin(), pw.end());
        append_shift->insert(append_shift->end(), shift.begin(), shift.end());
        c.w = nullptr;
        c.name = store_as;
        r->convs[store_as] = c;
        return 0;
    }
    int st = upload_weight(r, pw, &c);
    if (st) return st;
    st = upload(r, shift, &c.shift);
    if (st) return st;
    if (!scale.empty()) {
        st = upload(r, scale, &c.scale);
        if (st) return st;
    }
    c.name = store_as;
    r->convs[store_as] = c;
    return 0;
}

// `as`: name the packed layers are stored under.  "cnetb" = the context encoder's convolutions WITHOUT the folded running
// statistics, plus gamma / beta of every BatchNorm: the layers of the batch-statistics mode (see ofx_raft_forward).
int build_encoder(ofx_raft* r, const std::map<std::string, HostTensor>& sd, const std::string& enc, bool bn,
                  const std::string& as_ = std::string()) {
    const std::string as = as_.empty() ? enc : as_;
    const bool affine = !as_.empty();
    auto B = [&](const std::string& n) { return bn && !affine ? enc + "." + n : std::string(); };
    auto A = [&](const std::string& n) -> int {   // upload gamma / beta of norm layer `n`
        if (!affine) return 0;
        const HostTensor* g = find(sd, enc + "." + n + ".weight");
        const HostTensor* be = find(sd, enc + "." + n + ".bias");
        if (!g || !be || g->numel() != be->numel() || g->numel() % 4) return OFX_EKEY;
        std::vector<float> hg(g->data, g->data + g->numel()), hb(be->data, be->data + be->numel());
        float *dg = nullptr, *db = nullptr;
        int st = upload(r, hg, &dg);
        if (!st) st = upload(r, hb, &db);
        r->affine[as + "." + n] = std::make_pair(dg, db);
        return st;
    };
    int st = add_conv(r, sd, enc + ".conv1", as + ".conv1", 4, B("norm1"), 1.f);
    if (!st) st = A("norm1");
    if (st) return st;
    for (int li = 1; li <= 3; ++li) {
        for (int bi = 0; bi < 2; ++bi) {
            const std::string p = enc + ".layer" + std::to_string(li) + "." + std::to_string(bi);
            const std::string pa = as + ".layer" + std::to_string(li) + "." + std::to_string(bi);
            const std::string pb = "layer" + std::to_string(li) + "." + std::to_string(bi);
            st = add_conv(r, sd, p + ".conv1", pa + ".conv1", 0, B(pb + ".norm1"), 1.f);
            if (!st) st = A(pb + ".norm1");
            if (st) return st;
            st = add_conv(r, sd, p + ".conv2", pa + ".conv2", 0, B(pb + ".norm2"), 1.f);
            if (!st) st = A(pb + ".norm2");
            if (st) return st;
            if (li > 1 && bi == 0) {
                st = add_conv(r, sd, p + ".downsample.0", pa + ".down", 0, B(pb + ".norm3"), 1.f);
                if (!st) st = A(pb + ".norm3");
                if (st) return st;
            }
        }
    }
    return add_conv(r, sd, enc + ".conv2", as + ".conv2", 0, "", 1.f);
}

int build_gru(ofx_raft* r, const std::map<std::string, HostTensor>& sd, const std::string& tag) {
    // The reference's GRU input is cat([h, x]) with x = cat([inp, motion]) (update.py:131-133): input
    // channels 0..127 = h, 128..255 = inp, 256..383 = motion.  Split every GRU conv into
    //   * a recurrent part over [h | motion] (256 channels, no bias) evaluated every iteration, and
    //   * a loop-invariant part over [inp] (128 channels, with the bias) evaluated once per forward.
    // z and r share their input: one conv with Cout = 256 ([convz ; convr] rows).
    std::vector<int> rec, inv;
    for (int c = 0; c < HD; ++c) rec.push_back(c);
    for (int c = 0; c < 128; ++c) rec.push_back(HD + CD + c);
    for (int c = 0; c < CD; ++c) inv.push_back(HD + c);
    for (int part = 0; part < 2; ++part) {
        const std::vector<int>* sel = part == 0 ? &rec : &inv;
        const bool bias = part == 1;
        const std::string sfx = part == 0 ? "" : ".inp";
        std::vector<float> w, sh;
        int st = add_conv(r, sd, "update_block.gru.convz" + tag, "gru.z" + tag + sfx, 0, "", 1.f, &w, &sh, sel, bias);
        if (st) return st;
        st = add_conv(r, sd, "update_block.gru.convr" + tag, "gru.r" + tag + sfx, 0, "", 1.f, &w, &sh, sel, bias);
        if (st) return st;
        ConvW c = r->convs["gru.z" + tag + sfx];
        c.cout *= 2;
        st = upload_weight(r, w, &c);
        if (st) return st;
        if (bias) {
            st = upload(r, sh, &c.shift);
            if (st) return st;
        }
        c.name = "gru.zr" + tag + sfx;
        r->convs["gru.zr" + tag + sfx] = c;
        st = add_conv(r, sd, "update_block.gru.convq" + tag, "gru.q" + tag + sfx, 0, "", 1.f, nullptr, nullptr, sel, bias);
        if (st) return st;
    }
    return 0;
}

struct Launcher {
    hipStream_t s;
    int st = 0;
    const float* addend = nullptr;   // consumed (and cleared) by the next conv() call
    int ldadd = 0;
    int precision = OFX_PREC_FP32;   // sticky: OFX_PREC_BF16X3 when the forward was asked for the split-bf16 mode
    void* sk_ws = nullptr;           // split-K scratch of the stream this launcher feeds (null: never split)
    size_t sk_bytes = 0;
    float* stats_part = nullptr;     // consumed by the next conv(): instance-norm partial sums out of its epilogue (conv.hip)
    size_t stats_floats = 0;
    int stats_rows = 0;              // set by that conv(): rows per image in stats_part, 0 = not produced
    // generic conv launch; all pointer plumbing in one place
    void conv(const ConvW& c, const float* in0, int ld0, int c0, const float* in1, int ld1, int c1, float* out, int ldo,
              int B, int Hin, int Win, int stride, int act, int epi = OFX_EPI_PLAIN, const float* res = nullptr,
              int ldres = 0, const float* nmean = nullptr, const float* nrstd = nullptr, float* aux_z = nullptr,
              float* aux_rh = nullptr, float* aux_h = nullptr, int ldh = 0, float* aux_coords = nullptr,
              float* aux_flow4 = nullptr, int row_off = 0, int rows = 0) {
        if (st) return;
        ofx_conv_desc d{};
        d.in0 = in0; d.ld0 = ld0; d.c0 = c0;
        d.in1 = in1; d.ld1 = ld1; d.c1 = c1;
        const int cout = rows ? rows : c.cout;
        const bool wsplit = precision == OFX_PREC_BF16X3 && c.wsplit != nullptr;
        // (the three-piece format addresses its lo groups from the end of the whole matrix: row slices keep the on-the-fly split)
        const bool wsplit3 = precision == OFX_PREC_BF16X6 && c.wsplit3 != nullptr && row_off == 0 && rows == 0;
        d.w = wsplit3 ? c.wsplit3 : (wsplit ? c.wsplit : c.w) + (long)row_off * c.kpad;
        d.scale = c.scale ? c.scale + row_off : nullptr;
        d.shift = c.shift ? c.shift + row_off : nullptr;
        d.out = out; d.ldo = ldo;
        d.res = res; d.ldres = ldres;
        d.nmean = nmean; d.nrstd = nrstd;
        d.addend = addend; d.ldadd = ldadd;
        addend = nullptr; ldadd = 0;
        d.aux_z = aux_z; d.aux_rh = aux_rh; d.aux_h = aux_h; d.ldh = ldh;
        d.aux_coords = aux_coords; d.aux_flow4 = aux_flow4;
        d.B = B; d.Hin = Hin; d.Win = Win;
        const int padH = c.kh / 2, padW = c.kw / 2;
        d.Hout = (Hin + 2 * padH - c.kh) / stride + 1;
        d.Wout = (Win + 2 * padW - c.kw) / stride + 1;
        d.Cout = cout; d.KH = c.kh; d.KW = c.kw; d.stride = stride; d.padH = padH; d.padW = padW;
        d.act = act; d.epi = epi;
        d.precision = wsplit3 ? OFX_PREC_BF16X6_W : wsplit ? OFX_PREC_BF16X3_W : precision;
        d.splitk_ws = sk_ws; d.splitk_ws_bytes = sk_bytes;
        if (c0 + c1 != c.cin_pad) { st = OFX_EKEY; return; }
        ofx_prof_set_tag(c.name.c_str());
        stats_rows = 0;
        if (stats_part) st = ofx_conv2d_stats(&d, stats_part, stats_floats, &stats_rows, s);
        else st = ofx_conv2d(&d, s);
        stats_part = nullptr;
        ofx_prof_set_tag(nullptr);
    }
};

struct Streams {
    ofx_raft* r;
    hipStream_t main;
    bool on;
    hipStream_t get(int i) const { return on ? r->aux[i] : main; }
    int fork(int i) const {   // side stream i waits for everything enqueued on the caller's stream so far
        if (!on) return 0;
        OFX_HIP_CHECK(hipEventRecord(r->ev_fork, main));
        OFX_HIP_CHECK(hipStreamWaitEvent(r->aux[i], r->ev_fork, 0));
        return 0;
    }
    int join(int i) const {   // the caller's stream waits for side stream i
        if (!on) return 0;
        OFX_HIP_CHECK(hipEventRecord(r->ev_join[i], r->aux[i]));
        OFX_HIP_CHECK(hipStreamWaitEvent(main, r->ev_join[i], 0));
        return 0;
    }
    // The same two edges with the event riding on a kernel's own dispatch packet (ofx_tl_stop_event, ofx_internal.h) instead of a
    // marker packet behind it: arm_*() before the launch that ends the producing chain, *_armed() where fork() / join() would stand.
    // On a single 512x768 pair the marker of fork() held the caller's stream for ~7 us per iteration (profiles/r05_single_pair_gap_pairs.txt).
    static bool stop_events() { static const bool off = getenv("OFX_NO_STOP_EVENT") != nullptr; return !off; }
    bool arm_fork() const { if (!on || !stop_events()) return false; ofx_tl_stop_event = r->ev_fork; return true; }
    bool arm_join(int i) const { if (!on || !stop_events()) return false; ofx_tl_stop_event = r->ev_join[i]; return true; }
    static bool taken() {   // did the launcher hand the armed event to its kernel?  (if not: disarm, the caller takes the plain edge)
        const bool t = ofx_tl_stop_event == nullptr;
        ofx_tl_stop_event = nullptr;
        return t;
    }
    int fork_armed(int i) const { OFX_HIP_CHECK(hipStreamWaitEvent(r->aux[i], r->ev_fork, 0)); return 0; }
    int join_armed(int i) const { OFX_HIP_CHECK(hipStreamWaitEvent(main, r->ev_join[i], 0)); return 0; }
};

constexpr size_t SK_BYTES = 65536 + (size_t)1024 * 4 * 64 * 64 * sizeof(float);   // counters + 1024 tiles x 4 splits (conv.hip)

struct EncBufs {
    void* sk = nullptr;      // split-K scratch of the stream this encoder chain runs on
    size_t sk_bytes = 0;
    float *x0, *X, *Y, *R1, *R2, *R3;
    float* stats;     // 6 x [chunk][128] floats (mean/rstd for up to 3 norms)
    float* scratch;   // inorm partial sums
    float* spart;     // instance-norm partial sums written by the convolution epilogues: [chunk][rows][C][2]
    size_t spart_floats;
};

}  // namespace

// --------------------------------------------------------------------------------------------
static int run_encoder(ofx_raft* r, const std::string& enc, bool bn, const uint8_t* imgs, int n, int H, int W,
                       int bgr, const EncBufs& eb, float* out, int out_ld, bool split_tanh_relu, int relu_off, hipStream_t s,
                       int precision = OFX_PREC_FP32, bool separate_stats = false, const uint8_t* extra = nullptr) {
    // `extra`: one more image appended to the chunk (the shared key frame riding with the frames: see raft_forward_impl)
    // `out`: [n*h*w][out_ld]; fnet writes 256 channels; cnet writes tanh(0:128) | relu(128:256)
    Launcher L{s};
    L.precision = precision;
    L.sk_ws = eb.sk; L.sk_bytes = eb.sk_bytes;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    const int SC = (ENC_CHUNK + 1) * 128;   // floats per stats vector slot
    float* m1 = eb.stats + 0 * SC; float* s1 = eb.stats + 1 * SC;
    float* m2 = eb.stats + 2 * SC; float* s2 = eb.stats + 3 * SC;
    float* m3 = eb.stats + 4 * SC; float* s3 = eb.stats + 5 * SC;
    auto C = [&](const std::string& k) -> const ConvW& { return r->convs[enc + "." + k]; };
    int st = ofx_preprocess_u8(imgs, eb.x0, (long)n * H * W, bgr, s);
    if (st) return st;
    if (extra) {
        st = ofx_preprocess_u8(extra, eb.x0 + (long)n * H * W * 4, (long)H * W, bgr, s);
        if (st) return st;
        ++n;
    }
    // statistics of the tensor the conv() just before has written: from its epilogue's partial sums when it produced them.
    // `norm`: the layer's name -- for "cnetb" (BatchNorm on per-image statistics) its gamma / beta are folded into (mean, rstd)
    auto stats = [&](const float* x, long HW, int ch, float* mean, float* rstd, const std::string& norm) {
        if (L.st) return;
        const float *gamma = nullptr, *beta = nullptr;
        auto af = r->affine.find(enc + "." + norm);
        if (af != r->affine.end()) { gamma = af->second.first; beta = af->second.second; }
        if (L.stats_rows > 0) L.st = ofx_inorm_finalize_part(eb.spart, mean, rstd, n, L.stats_rows, HW, ch, 1e-5f, s, gamma, beta);
        else L.st = ofx_inorm_stats_affine(x, ch, mean, rstd, eb.scratch, n, HW, ch, 1e-5f, gamma, beta, s);
        L.stats_rows = 0;
    };
    static const bool no_epi_stats = getenv("OFX_NO_EPI_STATS") != nullptr;     // diagnostic: always the separate statistics pass
    auto want_stats = [&]() {
        if (no_epi_stats || separate_stats) return;
        L.stats_part = eb.spart;
        L.stats_floats = eb.spart_floats;
    };
    float* X = eb.X;
    float* Y = eb.Y;
    // instance norm: the stem's normalised output is never materialised.  Its raw output stays in X with its statistics in (m3, s3)
    // -- free until the first strided block -- and both of its consumers apply relu(norm(.)) on the fly: layer1.0.conv1 in its
    // operand staging (like every conv2), layer1.0's residual merge inside its inorm_apply (relu bit 1).  One full-resolution
    // read + write pass less per encoder.
    bool stem_raw = false;
    if (!bn) {
        want_stats();
        L.conv(C("conv1"), eb.x0, 4, 4, nullptr, 0, 0, X, 64, n, H, W, 2, OFX_ACT_NONE);
        stats(X, (long)H2 * W2, 64, m3, s3, "norm1");
        stem_raw = true;
    } else {
        L.conv(C("conv1"), eb.x0, 4, 4, nullptr, 0, 0, X, 64, n, H, W, 2, OFX_ACT_RELU);
    }
    int hin = H2, win = W2, cin = 64;
    const int dims[3] = {64, 96, 128};
    for (int li = 1; li <= 3; ++li) {
        const int dim = dims[li - 1];
        for (int bi = 0; bi < 2; ++bi) {
            const int stride = (li > 1 && bi == 0) ? 2 : 1;
            const int ho = hin / stride, wo = win / stride;
            const std::string p = "layer" + std::to_string(li) + "." + std::to_string(bi);
            if (!bn) {
                want_stats();
                L.conv(C(p + ".conv1"), X, cin, cin, nullptr, 0, 0, eb.R1, dim, n, hin, win, stride, OFX_ACT_NONE, OFX_EPI_PLAIN, nullptr, 0,
                       stem_raw ? m3 : nullptr, stem_raw ? s3 : nullptr);
                stats(eb.R1, (long)ho * wo, dim, m1, s1, p + ".norm1");
                // norm1 + ReLU fused into conv2's operand load
                want_stats();
                L.conv(C(p + ".conv2"), eb.R1, dim, dim, nullptr, 0, 0, eb.R2, dim, n, ho, wo, 1, OFX_ACT_NONE,
                       OFX_EPI_PLAIN, nullptr, 0, m1, s1);
                stats(eb.R2, (long)ho * wo, dim, m2, s2, p + ".norm2");
                if (stride == 2) {
                    want_stats();
                    L.conv(C(p + ".down"), X, cin, cin, nullptr, 0, 0, eb.R3, dim, n, hin, win, 2, OFX_ACT_NONE);
                    stats(eb.R3, (long)ho * wo, dim, m3, s3, p + ".norm3");
                    if (!L.st) L.st = ofx_inorm_apply(eb.R2, m2, s2, eb.R3, m3, s3, Y, n, (long)ho * wo, dim, 1, s);
                } else {
                    if (!L.st)
                        L.st = ofx_inorm_apply(eb.R2, m2, s2, X, stem_raw ? m3 : nullptr, stem_raw ? s3 : nullptr, Y, n, (long)ho * wo, dim,
                                               stem_raw ? 3 : 1, s);
                }
                stem_raw = false;
            } else {
                L.conv(C(p + ".conv1"), X, cin, cin, nullptr, 0, 0, eb.R1, dim, n, hin, win, stride, OFX_ACT_RELU);
                const float* res = X;
                if (stride == 2) {
                    L.conv(C(p + ".down"), X, cin, cin, nullptr, 0, 0, eb.R3, dim, n, hin, win, 2, OFX_ACT_NONE);
                    res = eb.R3;
                }
                L.conv(C(p + ".conv2"), eb.R1, dim, dim, nullptr, 0, 0, Y, dim, n, ho, wo, 1, OFX_ACT_RELU, OFX_EPI_PLAIN,
                       res, dim);
            }
            std::swap(X, Y);
            hin = ho; win = wo; cin = dim;
        }
    }
    (void)H4; (void)W4; (void)H8; (void)W8;
    if (!split_tanh_relu) {
        L.conv(C("conv2"), X, 128, 128, nullptr, 0, 0, out, out_ld, n, hin, win, 1, OFX_ACT_NONE);
    } else {
        // net = tanh(cnet[:, :128]), inp = relu(cnet[:, 128:256])  (raft.py:111-114)
        L.conv(C("conv2"), X, 128, 128, nullptr, 0, 0, out, out_ld, n, hin, win, 1, OFX_ACT_TANH, OFX_EPI_PLAIN, nullptr, 0,
               nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, HD);
        L.conv(C("conv2"), X, 128, 128, nullptr, 0, 0, out + relu_off, out_ld, n, hin, win, 1, OFX_ACT_RELU, OFX_EPI_PLAIN, nullptr,
               0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, HD, CD);
    }
    return L.st;
}

// workspace layout shared by the size query and the forward
struct RaftWs {
    EncBufs eb[3];    // [1], [2] only exist (else alias [0]) when the encoders may run concurrently
    void* sk[3];      // split-K scratch per stream (small batches only; null otherwise)
    float *fmap1, *fmap2, *f2l[LEVELS];
    float* fmap2b;    // image-2 feature maps with rows in the blocked order of the volume's columns (corr.hip)
    float* ctx;       // indexed-pairs mode: per-image context features [n][N][256] (tanh | relu halves)
    int* idx_dev;     // indexed-pairs mode: image1 index per pair
    float* pyr[LEVELS];
    float *hx, *gadd, *coords1, *frows, *corr, *c1, *corflo, *f1, *z, *rh, *mask;
    void* warp_pad;   // zero-bordered RGBX copy of the frame the tail warps (ofx_raft_forward_warp)
    size_t bytes;
};

// n_images > 0 selects the indexed-pairs layout: one feature map + one context map per unique image
static RaftWs carve(void* base, size_t cap, int B, int H, int W, int flags, int n_images = 0) {
    RaftWs w{};
    Carver c(base, cap);
    const int h = H / 8, wd = W / 8;
    const long N = (long)h * wd;
    const long M = (long)B * N;
    const int most = n_images > 0 ? n_images : 2 * B;
    const int nch = std::min(enc_chunk(H, W) + 1, most);   // + 1: the shared key frame rides in the last chunk of the frames (fold_key)
    const long half = (long)(H / 2) * (W / 2) * 64;
    for (int e = 0; e < 3; ++e) w.sk[e] = overlap_pays(M) ? (void*)c.take(SK_BYTES / sizeof(float)) : nullptr;
    for (int e = 0; e < 3; ++e) {
        if (e > 0 && !overlap_pays(M)) {
            w.eb[e] = w.eb[0];
            continue;
        }
        EncBufs& eb = w.eb[e];
        eb.sk = w.sk[e];
        eb.sk_bytes = w.sk[e] ? SK_BYTES : 0;
        eb.x0 = c.take((size_t)nch * H * W * 4);
        eb.X = c.take((size_t)nch * half);
        eb.Y = c.take((size_t)nch * half);
        eb.R1 = c.take((size_t)nch * half);
        eb.R2 = c.take((size_t)nch * half);
        eb.R3 = c.take((size_t)nch * (H / 4) * (W / 4) * 96);
        eb.stats = c.take((size_t)6 * (ENC_CHUNK + 1) * 128);
        eb.scratch = c.take((size_t)(ENC_CHUNK + 1) * 64 * 128 * 2 * 2);   // doubles: chunk x slices x C x {sum, sumsq}
        // one (sum, sumsq) pair per channel and 32 tile rows at most: the half-resolution 64-channel stage bounds it
        eb.spart_floats = (size_t)nch * (H / 2) * (W / 2) * 4 + 4096;
        eb.spart = c.take(eb.spart_floats);
    }
    long n1 = (flags & OFX_RAFT_SHARED_IMG1) ? 1 : B, n2 = (flags & OFX_RAFT_SHARED_IMG2) ? 1 : B;
    if (n_images > 0) {
        n1 = n_images;
        n2 = 0;
        w.ctx = c.take((size_t)n_images * N * (HD + CD));
        w.idx_dev = (int*)c.take((size_t)2 * B);   // image1 indices, then image2 indices
    }
    w.fmap1 = c.take((size_t)n1 * N * FD);
    w.fmap2 = n2 ? c.take((size_t)n2 * N * FD) : w.fmap1;
    if (flags & OFX_RAFT_ALT_CORR) {
        w.f2l[0] = w.fmap2;
        for (int l = 1; l < LEVELS; ++l) w.f2l[l] = c.take((size_t)n2 * (h >> l) * (wd >> l) * FD);
    } else {
        w.fmap2b = c.take((size_t)(n_images > 0 ? n_images : n2) * ofx_corr_slice_floats_l(h, wd) * FD);
        for (int l = 0; l < LEVELS; ++l) w.pyr[l] = c.take((size_t)M * ofx_corr_slice_floats_l(h >> l, wd >> l));
    }
    w.hx = c.take((size_t)M * HX_LD);
    w.gadd = c.take((size_t)M * GADD_LD);
    w.coords1 = c.take((size_t)M * 2);
    w.frows = c.take((size_t)M * FROW);
    w.corr = c.take((size_t)M * CORR_LD);
    w.c1 = c.take((size_t)M * 256);
    w.corflo = c.take((size_t)M * 256);
    w.f1 = c.take((size_t)M * 128);
    w.z = c.take((size_t)M * HD);
    w.rh = c.take((size_t)M * HD);
    w.mask = c.take((size_t)M * 576);
    w.warp_pad = c.take(ofx_warp_pad_bytes(H, W) / sizeof(float) + 4);
    w.bytes = c.off;
    return w;
}

// everything after the feature / context encoders and the correlation volume: state init, the loop-invariant
// GRU terms, `iters` refinement iterations, mask head, convex upsample
static int run_recurrence(ofx_raft* r, const RaftWs& ws, int B, int h, int w, int iters, bool alt, bool shared,
                          float* flow_up, float* flow_low, hipStream_t s, int precision, bool overlap,
                          uint8_t* warped = nullptr, float warp_sign = 1.0f, int n_warp = -1) {
    const long N = (long)h * w;
    const bool sh1 = shared, sh2 = shared;
    int st = 0;
    st = ofx_init_state(ws.coords1, ws.frows, ws.hx, HX_LD, FLOW_OFF, B, h, w, s);
    if (st) return st;
    {   // loop-invariant GRU terms: conv(W[:, inp], inp) + bias for z|r and q of both passes
        Launcher G{s};
        G.precision = precision;
        G.sk_ws = ws.sk[0]; G.sk_bytes = ws.sk[0] ? SK_BYTES : 0;
        const char* names[4] = {"gru.zr1.inp", "gru.q1.inp", "gru.zr2.inp", "gru.q2.inp"};
        const int offs[4] = {0, 2 * HD, 3 * HD, 5 * HD};
        for (int i = 0; i < 4; ++i)
            G.conv(r->convs[names[i]], ws.hx + INP_OFF, HX_LD, CD, nullptr, 0, 0, ws.gadd + offs[i], GADD_LD, B, h, w, 1,
                   OFX_ACT_NONE);
        if (G.st) return G.st;
    }

    if (alt) OFX_HIP_CHECK(hipMemsetAsync(ws.corr, 0, (size_t)B * N * CORR_LD * sizeof(float), s));   // (the pad columns: the volume-free kernel writes 324 of 336)
    Launcher L{s};
    L.precision = precision;
    L.sk_ws = ws.sk[0]; L.sk_bytes = ws.sk[0] ? SK_BYTES : 0;
    const Streams S{r, s, overlap};
    Launcher LF{S.get(0)};   // flow branch of the motion encoder: independent of the correlation branch
    LF.precision = precision;
    LF.sk_ws = overlap ? ws.sk[1] : ws.sk[0]; LF.sk_bytes = LF.sk_ws ? SK_BYTES : 0;   // its own scratch when it runs concurrently
    auto C = [&](const char* k) -> const ConvW& { return r->convs[k]; };
    const float* pyr_c[LEVELS] = {ws.pyr[0], ws.pyr[1], ws.pyr[2], ws.pyr[3]};
    const int rd2 = (2 * RADIUS + 1) * (2 * RADIUS + 1);
    bool fork_on_kernel = false;   // the previous iteration's flow head carries ev_fork
    for (int it = 0; it < iters && !L.st; ++it) {
        // flow features (update.py:93-94) on the side stream, from the flow the previous iteration left
        if ((L.st = fork_on_kernel ? S.fork_armed(0) : S.fork(0))) break;
        LF.conv(C("convf1"), ws.frows, FROW, FROW, nullptr, 0, 0, ws.f1, 128, B, h, w, 1, OFX_ACT_RELU);   // 7x1 over the flow rows
        const bool join_armed = S.arm_join(0);
        LF.conv(C("convf2"), ws.f1, 128, 128, nullptr, 0, 0, ws.corflo + 192, 256, B, h, w, 1, OFX_ACT_RELU);
        const bool join_on_kernel = join_armed && Streams::taken();
        if ((L.st = LF.st)) break;
        // correlation features at the current estimate
        if (!alt) {
            L.st = ofx_corr_lookup_pad(pyr_c, ws.coords1, ws.corr, CORR_LD, CORR_LD - CORR_CH, B, h, w, LEVELS, RADIUS, s);
        } else {
            for (int l = 0; l < LEVELS && !L.st; ++l) {
                if (sh1 || sh2) { L.st = OFX_EINVAL; break; }   // alt-corr path: per-pair feature maps only
                L.st = ofx_local_corr_launch(ws.fmap1, ws.f2l[l], ws.coords1, ws.corr + (long)l * rd2, N * CORR_LD, 0, 1,
                                             CORR_LD, B, h, w, h >> l, w >> l, FD, 1, RADIUS, 1.0f / std::sqrt((float)FD),
                                             1.0f / (float)(1 << l), s);
            }
        }
        // motion encoder (update.py:88-97)
        L.conv(C("convc1"), ws.corr, CORR_LD, CORR_LD, nullptr, 0, 0, ws.c1, 256, B, h, w, 1, OFX_ACT_RELU);
        L.conv(C("convc2"), ws.c1, 256, 256, nullptr, 0, 0, ws.corflo, 256, B, h, w, 1, OFX_ACT_RELU);
        if (!L.st) L.st = join_on_kernel ? S.join_armed(0) : S.join(0);
        L.conv(C("conv"), ws.corflo, 256, 256, nullptr, 0, 0, ws.hx + MOT_OFF, HX_LD, B, h, w, 1, OFX_ACT_RELU);
        // SepConvGRU (update.py:44-60): horizontal then vertical pass
        for (int pass = 1; pass <= 2; ++pass) {
            const char* zr = pass == 1 ? "gru.zr1" : "gru.zr2";
            const char* q = pass == 1 ? "gru.q1" : "gru.q2";
            const float* g = ws.gadd + (pass - 1) * (3 * HD);
            L.addend = g; L.ldadd = GADD_LD;
            L.conv(C(zr), ws.hx, HX_LD, 2 * HD, nullptr, 0, 0, nullptr, 0, B, h, w, 1, OFX_ACT_NONE, OFX_EPI_GRU_ZR, nullptr, 0,
                   nullptr, nullptr, ws.z, ws.rh, ws.hx, HX_LD);
            L.addend = g + 2 * HD; L.ldadd = GADD_LD;
            L.conv(C(q), ws.rh, HD, HD, ws.hx + MOT_OFF, HX_LD, 128, nullptr, 0, B, h, w, 1, OFX_ACT_NONE, OFX_EPI_GRU_Q,
                   nullptr, 0, nullptr, nullptr, ws.z, nullptr, ws.hx, HX_LD);
        }
        // flow head (update.py:6-14) + coords1 += delta (raft.py:131) in the epilogue
        L.conv(C("fh1"), ws.hx, HX_LD, HD, nullptr, 0, 0, ws.c1, 256, B, h, w, 1, OFX_ACT_RELU);
        if (!L.st) {   // 256 -> 2 channels: dedicated reduction kernel instead of a 1/16-utilised GEMM tile
            const ConvW& f2 = C("fh2");
            const bool fork_armed = it + 1 < iters && S.arm_fork();
            L.st = ofx_flow_head_launch(ws.c1, 256, f2.w, (int)f2.kpad, f2.shift, ws.coords1, ws.hx + FLOW_OFF, HX_LD, ws.frows, B, h,
                                        w, s);
            fork_on_kernel = fork_armed && Streams::taken();
        }
    }
    // mask head (update.py:122-125,135) on the final hidden state, then convex upsample
    L.conv(C("mask0"), ws.hx, HX_LD, HD, nullptr, 0, 0, ws.c1, 256, B, h, w, 1, OFX_ACT_RELU);
    L.conv(C("mask2"), ws.c1, 256, 256, nullptr, 0, 0, ws.mask, 576, B, h, w, 1, OFX_ACT_NONE);
    if (L.st) return L.st;
    // convex upsample -- with the backward warp of the AI key frame in the same pass when the caller asked for it: the flow is in
    // registers right there, flow_up is then written only if wanted
    // (n_warp: only the first n_warp pairs are warped -- the indexed-pairs call of the forward-backward confidence, whose reverse
    // flows need no warp; the rest take the plain upsample)
    const int nw = warped ? (n_warp < 0 ? B : std::min(n_warp, B)) : 0;
    if (nw > 0) st = ofx_upsample_warp_launch(ws.coords1, ws.mask, flow_up, ws.warp_pad, warped, nw, h, w, warp_sign, s);
    if (!st && nw < B) {
        if (!flow_up) return OFX_EINVAL;
        st = ofx_upsample_flow(ws.coords1 + (long)nw * N * 2, ws.mask + (long)nw * N * 576, flow_up + (long)nw * N * 128, B - nw, h, w, s);
    }
    if (st) return st;
    if (flow_low) {
        st = ofx_coords_to_flow(ws.coords1, flow_low, B, h, w, s);
        if (st) return st;
    }

    return 0;
}

// The volume GEMM on the A-stationary kernel (corr_split.hip).  OFX_RAFT_VOL_BF16X3 / _BF16X6 ask for its split-bf16 forms alone (every
// convolution stays fp32), the all-layer split modes take them along.  Its operands in fragment order live in the recurrence's scratch
// (corr ... mask: dead until the first iteration).  0 = the generic batched GEMM of conv.hip (shapes the kernel does not take).
static int volume_planes(int flags, int h, int w, long n_planes_images, const RaftWs& ws, long M, int nz) {
    // 1: the exact-fp32 form of the same A-stationary kernel (round 6: bit-identical to the generic batched GEMM it replaces, whole-line
    // stores, no operand re-fetch) -- the default wherever the shape allows; OFX_VOL_GENERIC=1 keeps the generic GEMM (diagnostic)
    const int planes = (flags & (OFX_RAFT_VOL_BF16X6 | OFX_RAFT_BF16X6)) ? 3 : (flags & (OFX_RAFT_VOL_BF16X3 | OFX_RAFT_BF16X3)) ? 2 : 1;
    static const bool generic = getenv("OFX_VOL_GENERIC") != nullptr;
    if ((planes == 1 && generic) || !ofx_corr_volsplit_ok(h, w, FD) || !ofx_corr_volsplit_pays(nz, h, w, planes) || getenv("OFX_NO_VOLSPLIT")) return 0;
    const size_t room = (size_t)((const char*)(ws.mask + M * 576) - (const char*)ws.corr);
    return (size_t)n_planes_images * ofx_corr_planes_bytes(h, w, planes) <= room ? planes : 0;
}

static int raft_forward_impl(ofx_raft* r, const uint8_t* image1, const uint8_t* image2, int B, int H, int W, int iters, int flags,
                             float* flow_up, float* flow_low, const uint8_t* warp_frame, float warp_sign, uint8_t* warped, void* workspace,
                             size_t workspace_bytes, void* stream);

extern "C" {

int ofx_raft_create(const ofx_tensor* tensors, int n, ofx_raft** out) {
    OFX_REQUIRE(tensors && n > 0 && out, OFX_EINVAL);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return OFX_ENODEV;
    std::map<std::string, HostTensor> sd;
    for (int i = 0; i < n; ++i) {
        if (!tensors[i].name || !tensors[i].data) continue;
        std::string k = tensors[i].name;
        if (k.rfind("module.", 0) == 0) k = k.substr(7);
        HostTensor t;
        t.data = tensors[i].data;
        t.ndim = tensors[i].ndim;
        for (int j = 0; j < 4; ++j) t.shape[j] = j < t.ndim ? tensors[i].shape[j] : 1;
        sd[k] = t;
    }
    ofx_raft* r = new ofx_raft();
    int st = build_encoder(r, sd, "fnet", false);
    if (!st) st = build_encoder(r, sd, "cnet", true);
    if (!st) st = build_encoder(r, sd, "cnet", true, "cnetb");
    const char* ub = "update_block.";
    if (!st) st = add_conv(r, sd, std::string(ub) + "encoder.convc1", "convc1", CORR_LD, "", 1.f);   // input rows padded to 336 (zero weights)
    if (!st) st = add_conv(r, sd, std::string(ub) + "encoder.convc2", "convc2", 0, "", 1.f);
    std::vector<float> wf1;   // must outlive add_conv below
    if (!st) {
        // convf1 (7x7 on the 2-channel flow, update.py:93) as a 7x1 convolution over the 16-float flow rows the flow head leaves
        // (flow_head.hip): row channel kx * 2 + c = tap kx of input channel c, channels 14 and 15 are zero.  K = 112 instead of 196.
        const HostTensor* wf = find(sd, std::string(ub) + "encoder.convf1.weight");
        const HostTensor* bf = find(sd, std::string(ub) + "encoder.convf1.bias");
        if (!wf || !bf || wf->ndim != 4 || wf->shape[1] != 2 || wf->shape[2] != 7 || wf->shape[3] != 7) st = OFX_EKEY;
        if (!st) {
            const int co = (int)wf->shape[0];
            wf1.assign((size_t)co * 14 * 7, 0.f);
            for (int o = 0; o < co; ++o)
                for (int c = 0; c < 2; ++c)
                    for (int ky = 0; ky < 7; ++ky)
                        for (int kx = 0; kx < 7; ++kx)
                            wf1[((size_t)o * 14 + kx * 2 + c) * 7 + ky] = wf->data[(((size_t)o * 2 + c) * 7 + ky) * 7 + kx];
            HostTensor t;
            t.data = wf1.data(); t.ndim = 4;
            t.shape[0] = co; t.shape[1] = 14; t.shape[2] = 7; t.shape[3] = 1;
            sd["convf1_rows.weight"] = t;
            sd["convf1_rows.bias"] = *bf;
            st = add_conv(r, sd, "convf1_rows", "convf1", FROW, "", 1.f);
        }
    }
    if (!st) st = add_conv(r, sd, std::string(ub) + "encoder.convf2", "convf2", 0, "", 1.f);
    if (!st) st = add_conv(r, sd, std::string(ub) + "encoder.conv", "conv", 0, "", 1.f);
    if (!st) st = build_gru(r, sd, "1");
    if (!st) st = build_gru(r, sd, "2");
    if (!st) st = add_conv(r, sd, std::string(ub) + "flow_head.conv1", "fh1", 0, "", 1.f);
    if (!st) st = add_conv(r, sd, std::string(ub) + "flow_head.conv2", "fh2", 0, "", 1.f);
    if (!st) st = add_conv(r, sd, std::string(ub) + "mask.0", "mask0", 0, "", 1.f);
    if (!st) st = add_conv(r, sd, std::string(ub) + "mask.2", "mask2", 0, "", 0.25f);
    for (int i = 0; i < 2 && !st; ++i) {
        if (hipStreamCreateWithFlags(&r->aux[i], hipStreamNonBlocking) != hipSuccess) st = OFX_ENODEV;
        if (!st && hipEventCreateWithFlags(&r->ev_join[i], hipEventDisableTiming) != hipSuccess) st = OFX_ENODEV;
    }
    if (!st && hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming) != hipSuccess) st = OFX_ENODEV;
    if (st) {
        ofx_raft_destroy(r);
        return st;
    }
    *out = r;
    return 0;
}

int ofx_raft_destroy(ofx_raft* r) {
    if (!r) return 0;
    for (void* p : r->allocs) (void)hipFree(p);
    for (int i = 0; i < 2; ++i) {
        if (r->aux[i]) (void)hipStreamDestroy(r->aux[i]);
        if (r->ev_join[i]) (void)hipEventDestroy(r->ev_join[i]);
    }
    if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
    delete r;
    return 0;
}

size_t ofx_raft_workspace_bytes(const ofx_raft* r, int B, int H, int W) {
    (void)r;
    if (B <= 0 || H <= 0 || W <= 0 || (H % 8) || (W % 8)) return 0;
    // the volume layout is the larger of the two correlation modes
    size_t a = carve(nullptr, 0, B, H, W, 0).bytes;
    size_t b = carve(nullptr, 0, B, H, W, OFX_RAFT_ALT_CORR).bytes;
    return a > b ? a : b;
}

int ofx_raft_forward(ofx_raft* r, const uint8_t* image1, const uint8_t* image2, int B, int H, int W, int iters,
                     int flags, float* flow_up, float* flow_low, void* workspace, size_t workspace_bytes, void* stream) {
    OFX_REQUIRE(flow_up, OFX_EINVAL);
    return raft_forward_impl(r, image1, image2, B, H, W, iters, flags, flow_up, flow_low, nullptr, 1.0f, nullptr, workspace, workspace_bytes, stream);
}

int ofx_raft_forward_warp(ofx_raft* r, const uint8_t* image1, const uint8_t* image2, int B, int H, int W, int iters, int flags,
                          float* flow_up, float* flow_low, const uint8_t* warp_frame, float warp_sign, uint8_t* warped, void* workspace,
                          size_t workspace_bytes, void* stream) {
    OFX_REQUIRE(warp_frame && warped && (warp_sign == 1.0f || warp_sign == -1.0f), OFX_EINVAL);
    OFX_REQUIRE(ofx_upsample_warp_ok(B, H, W) && (((uintptr_t)warped) & 3u) == 0, OFX_EINVAL);
    return raft_forward_impl(r, image1, image2, B, H, W, iters, flags, flow_up, flow_low, warp_frame, warp_sign, warped, workspace, workspace_bytes,
                             stream);
}

}  // extern "C"

static int raft_forward_impl(ofx_raft* r, const uint8_t* image1, const uint8_t* image2, int B, int H, int W, int iters, int flags,
                             float* flow_up, float* flow_low, const uint8_t* warp_frame, float warp_sign, uint8_t* warped, void* workspace,
                             size_t workspace_bytes, void* stream) {
    OFX_REQUIRE(r && image1 && image2 && (flow_up || warped) && workspace, OFX_EINVAL);
    OFX_REQUIRE(B > 0 && H >= 64 && W >= 64 && (H % 8) == 0 && (W % 8) == 0 && iters >= 1, OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)workspace) & 255u) == 0, OFX_EALIGN);
    RaftWs ws = carve(workspace, workspace_bytes, B, H, W, flags);
    OFX_REQUIRE(ws.bytes <= workspace_bytes, OFX_ENOMEM);
    hipStream_t s = (hipStream_t)stream;
    for (int e = 0; e < 3; ++e)   // split-K arrival counters start at zero (the kernels leave them at zero)
        if (ws.sk[e]) OFX_HIP_CHECK(hipMemsetAsync(ws.sk[e], 0, 65536, s));
    const int h = H / 8, w = W / 8;
    const long N = (long)h * w;
    const long M = (long)B * N;
    const int bgr = (flags & OFX_RAFT_BGR) ? 1 : 0;
    const int prec = (flags & OFX_RAFT_BF16X6) ? OFX_PREC_BF16X6 : (flags & OFX_RAFT_BF16X3) ? OFX_PREC_BF16X3 : OFX_PREC_FP32;
    const bool sh1 = flags & OFX_RAFT_SHARED_IMG1, sh2 = flags & OFX_RAFT_SHARED_IMG2;
    const bool alt = flags & OFX_RAFT_ALT_CORR;
    const long img_bytes = (long)H * W * 3;
    // Context-encoder BatchNorm.  Default: the running statistics, folded into the convolutions (a model in .eval(), as PDCNet's
    // own code and canonical RAFT inference run it).  OFX_RAFT_BN_BATCH: the statistics of the image itself -- what
    // `RAFT_2` computes AS WRITTEN: the reference never calls .eval() (ofgen_keyframe_inpaint.py:47-60), so its
    // BatchNorm2d layers (RAFT/core/raft.py:55, extractor.py:118-192) normalise with the statistics of the call's batch,
    // and every call is one image.  Each image of a batch is normalised by itself here (= B reference calls).
    const bool bnb = flags & OFX_RAFT_BN_BATCH;
    const bool sepst = flags & OFX_RAFT_SEPARATE_STATS;
    const char* cnet = bnb ? "cnetb" : "cnet";

    // ---- encoders (instance norm => per-image statistics, so chunking is exact).  Three independent chains:
    // fnet(image1) on the caller's stream, fnet(image2) and cnet(image1) on the side streams when the batch is
    // too small to fill the machine by itself.
    int st = 0;
    const int n1 = sh1 ? 1 : B, n2 = sh2 ? 1 : B;
    const bool overlap = overlap_pays(M) && !(flags & OFX_RAFT_SERIAL);
    const Streams S{r, s, overlap};
    const int ech = enc_chunk(H, W);
    if ((st = S.fork(0))) return st;
    if ((st = S.fork(1))) return st;
    // A shared key frame encoded by itself is a single-image launch sequence on a 256-CU part (1.3 ms where 1/64 of the frames' pass is
    // 0.3): large batches append it to the last chunk of the frames instead -- instance norm is per image, so the result is the same
    // network; its feature map lands right behind the frames' (fmap2 follows fmap1 in the workspace).
    const long half_bytes = (long)(H / 2) * (W / 2) * 64 * 4;
    const bool fold_key = sh2 && !sh1 && !overlap_pays(M) && ws.fmap2 == ws.fmap1 + (long)n1 * N * FD &&   // (by batch size, not by schedule: OFX_RAFT_SERIAL must not change the arithmetic)
                          (long)(std::min(ech, (n1 - 1) % ech + 1) + 1) * half_bytes < (1L << 31) - 4096;
    for (int i0 = 0; i0 < n1 && !st; i0 += ech) {
        const int n = std::min(ech, n1 - i0);
        const bool last = i0 + n == n1;
        st = run_encoder(r, "fnet", false, image1 + i0 * img_bytes, n, H, W, bgr, ws.eb[0], ws.fmap1 + (long)i0 * N * FD, FD,
                         false, 0, s, prec, sepst, (fold_key && last) ? image2 : nullptr);
    }
    for (int i0 = 0; i0 < n2 && !st && !fold_key; i0 += ech) {
        const int n = std::min(ech, n2 - i0);
        st = run_encoder(r, "fnet", false, image2 + i0 * img_bytes, n, H, W, bgr, ws.eb[1], ws.fmap2 + (long)i0 * N * FD, FD,
                         false, 0, S.get(0), prec, sepst);
    }
    // context encoder on image1 -> hx[:, 0:128] = tanh (net), hx[:, 256:384] = relu (inp)
    if (sh1) {
        if (!st) st = run_encoder(r, cnet, !bnb, image1, 1, H, W, bgr, ws.eb[2], ws.hx, HX_LD, true, INP_OFF, S.get(1), prec, sepst);
        for (int k = 1; k < B && !st; ++k)   // one shared image1: replicate its context rows
            OFX_HIP_CHECK(hipMemcpyAsync(ws.hx + (long)k * N * HX_LD, ws.hx, (size_t)N * HX_LD * sizeof(float),
                                         hipMemcpyDeviceToDevice, S.get(1)));
    } else {
        for (int i0 = 0; i0 < B && !st; i0 += ech) {
            const int n = std::min(ech, B - i0);
            st = run_encoder(r, cnet, !bnb, image1 + i0 * img_bytes, n, H, W, bgr, ws.eb[2], ws.hx + (long)i0 * N * HX_LD,
                             HX_LD, true, INP_OFF, S.get(1), prec, sepst);
        }
    }
    if (!st) st = S.join(0);
    if (!st) st = S.join(1);
    if (st) return st;

    // ---- correlation
    const int vplanes = alt ? 0 : volume_planes(flags, h, w, (long)n1 + n2, ws, M, B);
    if (vplanes) {
        char* pl = (char*)ws.corr;
        const long ib = (long)ofx_corr_planes_bytes(h, w, vplanes);
        st = ofx_corr_split_planes(ws.fmap1, pl, n1, h, w, vplanes, 0, 1.0f / std::sqrt((float)FD), s);
        if (!st) st = ofx_corr_split_planes(ws.fmap2, pl + ib * n1, n2, h, w, vplanes, 1, 1.0f, s);
        if (!st)
            st = ofx_corr_vol_split_launch(pl, pl + ib * n1, nullptr, nullptr, sh1 ? 0 : ib, sh2 ? 0 : ib, ws.pyr[0], ws.pyr[1], B, h, w, vplanes, s);
        if (!st) st = ofx_corr_pool_launch(ws.pyr[0], ws.pyr[1], ws.pyr[2], ws.pyr[3], B, h, w, LEVELS, s, true);
        if (st) return st;
    } else if (!alt) {
        // the volume's columns in blocked order: permute the rows of the B operand once (6.3 MB per image)
        const long Nb = ofx_corr_slice_floats_l(h, w);
        st = ofx_corr_block_rows(ws.fmap2, ws.fmap2b, (int)n2, h, w, FD, s);
        if (st) return st;
        ofx_conv_desc d{};
        d.in0 = ws.fmap1; d.ld0 = FD; d.c0 = FD;
        d.w = ws.fmap2b;
        d.out = ws.pyr[0]; d.ldo = (int)Nb;
        d.nz = B; d.a_zs = sh1 ? 0 : N * FD; d.w_zs = sh2 ? 0 : Nb * FD; d.o_zs = N * Nb;
        d.B = 1; d.Hin = h; d.Win = w; d.Hout = h; d.Wout = w; d.Cout = (int)Nb;
        d.KH = 1; d.KW = 1; d.stride = 1;
        d.act = OFX_ACT_NONE; d.epi = OFX_EPI_PLAIN;
        d.precision = prec;
        // zero batch strides broadcast a shared key-frame feature map across the batch.  Level 1 of the pyramid comes out
        // of the GEMM's accumulators when it can (whole-block level 1): level 0 is then never read back
        const long slice1 = ofx_corr_slice_floats_l(h >> 1, w >> 1);
        const bool fused = ofx_corr_volpool_ok(h, w) && Nb % 128 == 0 && N * slice1 * 4 < (1L << 31) - 64;   // (every arithmetic since round 4)
        if (fused)
            st = ofx_conv2d_volpool(&d, 1.0f / std::sqrt((float)FD), ws.pyr[1], N * slice1, (w + 7) >> 3, ((w >> 1) + 7) >> 3, (int)slice1, s);
        else
            st = ofx_conv2d_alpha(&d, 1.0f / std::sqrt((float)FD), s);
        if (!st) st = ofx_corr_pool_launch(ws.pyr[0], ws.pyr[1], ws.pyr[2], ws.pyr[3], B, h, w, LEVELS, s, fused);
        if (st) return st;
    } else {
        for (int l = 1; l < LEVELS && !st; ++l)
            st = ofx_avgpool2_nhwc(ws.f2l[l - 1], ws.f2l[l], n2, h >> (l - 1), w >> (l - 1), FD, s);
        if (st) return st;
    }

    if (warped) {   // the zero-bordered RGBX copy the warp samples (1.6 MB at 512x768: stays in L2)
        st = ofx_warp_pad_launch(warp_frame, ws.warp_pad, H, W, s);
        if (st) return st;
    }
    st = run_recurrence(r, ws, B, h, w, iters, alt, sh1 || sh2, flow_up, flow_low, s, prec, overlap, warped,
                        warp_sign);
    if (st) return st;

    r->bufs.clear();
    auto reg = [&](const char* k, float* p, size_t nf) { r->bufs[k] = std::make_pair((void*)p, nf); };
    reg("fmap1", ws.fmap1, (size_t)n1 * N * FD);
    reg("fmap2", ws.fmap2, (size_t)n2 * N * FD);
    reg("hx", ws.hx, (size_t)M * HX_LD);
    reg("coords1", ws.coords1, (size_t)M * 2);
    reg("corr", ws.corr, (size_t)M * CORR_LD);   // rows of 336: 324 features + 12 zeros
    reg("mask", ws.mask, (size_t)M * 576);
    if (!alt)
        for (int l = 0; l < LEVELS; ++l) {
            char nm[8];
            snprintf(nm, sizeof nm, "pyr%d", l);
            reg(nm, ws.pyr[l], (size_t)M * ofx_corr_slice_floats_l(h >> l, w >> l));   // blocked layout (ofx.h)
        }
    return 0;
}

extern "C" {

size_t ofx_raft_workspace_bytes_pairs(const ofx_raft* r, int n_images, int B, int H, int W) {
    (void)r;
    if (n_images <= 0 || B <= 0 || H <= 0 || W <= 0 || (H % 8) || (W % 8)) return 0;
    return carve(nullptr, 0, B, H, W, 0, n_images).bytes;
}

int ofx_raft_forward_pairs(ofx_raft* r, const uint8_t* images, int n_images, const int* idx1, const int* idx2, int B, int H,
                           int W, int iters, int flags, float* flow_up, float* flow_low, void* workspace,
                           size_t workspace_bytes, void* stream) {
    return ofx_raft_forward_pairs_warp(r, images, n_images, idx1, idx2, B, H, W, iters, flags, flow_up, flow_low, nullptr, 1.0f, 0, nullptr,
                                       workspace, workspace_bytes, stream);
}

int ofx_raft_forward_pairs_warp(ofx_raft* r, const uint8_t* images, int n_images, const int* idx1, const int* idx2, int B, int H,
                                int W, int iters, int flags, float* flow_up, float* flow_low, const uint8_t* warp_frame, float warp_sign,
                                int n_warp, uint8_t* warped, void* workspace, size_t workspace_bytes, void* stream) {
    OFX_REQUIRE(r && images && idx1 && idx2 && flow_up && workspace, OFX_EINVAL);
    if (warped || warp_frame || n_warp) {
        OFX_REQUIRE(warp_frame && warped && n_warp > 0 && n_warp <= B && (warp_sign == 1.0f || warp_sign == -1.0f), OFX_EINVAL);
        OFX_REQUIRE(ofx_upsample_warp_ok(n_warp, H, W) && (((uintptr_t)warped) & 3u) == 0, OFX_EINVAL);
    }
    OFX_REQUIRE(n_images > 0 && B > 0 && H >= 64 && W >= 64 && (H % 8) == 0 && (W % 8) == 0 && iters >= 1, OFX_EINVAL);
    OFX_REQUIRE(!(flags & (OFX_RAFT_ALT_CORR | OFX_RAFT_SHARED_IMG1 | OFX_RAFT_SHARED_IMG2)), OFX_EINVAL);
    OFX_REQUIRE((((uintptr_t)workspace) & 255u) == 0, OFX_EALIGN);
    for (int b = 0; b < B; ++b)
        OFX_REQUIRE(idx1[b] >= 0 && idx1[b] < n_images && idx2[b] >= 0 && idx2[b] < n_images, OFX_EINVAL);
    RaftWs ws = carve(workspace, workspace_bytes, B, H, W, 0, n_images);
    OFX_REQUIRE(ws.bytes <= workspace_bytes, OFX_ENOMEM);
    hipStream_t s = (hipStream_t)stream;
    for (int e = 0; e < 3; ++e)
        if (ws.sk[e]) OFX_HIP_CHECK(hipMemsetAsync(ws.sk[e], 0, 65536, s));
    const int h = H / 8, w = W / 8;
    const long N = (long)h * w;
    const int bgr = (flags & OFX_RAFT_BGR) ? 1 : 0;
    const int prec = (flags & OFX_RAFT_BF16X6) ? OFX_PREC_BF16X6 : (flags & OFX_RAFT_BF16X3) ? OFX_PREC_BF16X3 : OFX_PREC_FP32;
    const long img_bytes = (long)H * W * 3;
    // Context-encoder BatchNorm.  Default: the running statistics, folded into the convolutions (a model in .eval(), as PDCNet's
    // own code and canonical RAFT inference run it).  OFX_RAFT_BN_BATCH: the statistics of the image itself -- what
    // `RAFT_2` computes AS WRITTEN: the reference never calls .eval() (ofgen_keyframe_inpaint.py:47-60), so its
    // BatchNorm2d layers (RAFT/core/raft.py:55, extractor.py:118-192) normalise with the statistics of the call's batch,
    // and every call is one image.  Each image of a batch is normalised by itself here (= B reference calls).
    const bool bnb = flags & OFX_RAFT_BN_BATCH;
    const bool sepst = flags & OFX_RAFT_SEPARATE_STATS;
    const char* cnet = bnb ? "cnetb" : "cnet";
    int st = 0;
    // every image is encoded ONCE (feature + context), however many pairs it takes part in: a 15-frame
    // KeyframeConv window has 210 ordered pairs but only 15 images (ofgen_keyframe_inpaint.py:627-668)
    const bool overlap = overlap_pays((long)B * N) && !(flags & OFX_RAFT_SERIAL);
    const Streams S{r, s, overlap};
    if ((st = S.fork(1))) return st;
    for (int i0 = 0; i0 < n_images && !st; i0 += enc_chunk(H, W)) {
        const int n = std::min(enc_chunk(H, W), n_images - i0);
        st = run_encoder(r, "fnet", false, images + i0 * img_bytes, n, H, W, bgr, ws.eb[0], ws.fmap1 + (long)i0 * N * FD, FD,
                         false, 0, s, prec, sepst);
        if (!st)
            st = run_encoder(r, cnet, !bnb, images + i0 * img_bytes, n, H, W, bgr, ws.eb[2], ws.ctx + (long)i0 * N * (HD + CD),
                             HD + CD, true, HD, S.get(1), prec, sepst);
    }
    if (!st) st = S.join(1);
    if (st) return st;
    OFX_HIP_CHECK(hipMemcpyAsync(ws.idx_dev, idx1, sizeof(int) * B, hipMemcpyHostToDevice, s));
    st = ofx_ctx_gather(ws.ctx, ws.idx_dev, ws.hx, HX_LD, INP_OFF, HD, B, N, s);
    if (st) return st;
    const long Nb = ofx_corr_slice_floats_l(h, w);
    const long slice1p = ofx_corr_slice_floats_l(h >> 1, w >> 1);
    const bool fused_pairs = ofx_corr_volpool_ok(h, w) && Nb % 128 == 0 && N * slice1p * 4 < (1L << 31) - 64;
    const int vplanes = volume_planes(flags, h, w, 2L * n_images, ws, (long)B * N, B);
    if (vplanes) {
        // split-bf16 volume: every image split once per role (rows scaled in pixel order / columns in quad order), ONE launch over the
        // pair list through the device-side index arrays
        char* pl = (char*)ws.corr;
        const long ib = (long)ofx_corr_planes_bytes(h, w, vplanes);
        OFX_HIP_CHECK(hipMemcpyAsync(ws.idx_dev + B, idx2, sizeof(int) * B, hipMemcpyHostToDevice, s));
        st = ofx_corr_split_planes(ws.fmap1, pl, n_images, h, w, vplanes, 0, 1.0f / std::sqrt((float)FD), s);
        if (!st) st = ofx_corr_split_planes(ws.fmap1, pl + ib * n_images, n_images, h, w, vplanes, 1, 1.0f, s);
        if (!st) st = ofx_corr_vol_split_launch(pl, pl + ib * n_images, ws.idx_dev, ws.idx_dev + B, 0, 0, ws.pyr[0], ws.pyr[1], B, h, w, vplanes, s);
    } else {
        st = ofx_corr_block_rows(ws.fmap1, ws.fmap2b, n_images, h, w, FD, s);   // every image can be an image2: blocked copy of all
    }
    if (st) return st;
    for (int b = 0; b < B && !st && !vplanes; ++b) {   // one N x Nb correlation GEMM per pair, straight from the shared feature maps
        ofx_conv_desc d{};
        d.in0 = ws.fmap1 + (long)idx1[b] * N * FD; d.ld0 = FD; d.c0 = FD;
        d.w = ws.fmap2b + (long)idx2[b] * Nb * FD;
        d.out = ws.pyr[0] + (long)b * N * Nb; d.ldo = (int)Nb;
        d.B = 1; d.Hin = h; d.Win = w; d.Hout = h; d.Wout = w; d.Cout = (int)Nb;
        d.KH = 1; d.KW = 1; d.stride = 1;
        d.act = OFX_ACT_NONE; d.epi = OFX_EPI_PLAIN;
        d.precision = prec;
        if (fused_pairs)
            st = ofx_conv2d_volpool(&d, 1.0f / std::sqrt((float)FD), ws.pyr[1] + (long)b * N * slice1p, 0, (w + 7) >> 3, ((w >> 1) + 7) >> 3,
                                    (int)slice1p, s);
        else
            st = ofx_conv2d_alpha(&d, 1.0f / std::sqrt((float)FD), s);
    }
    if (!st) st = ofx_corr_pool_launch(ws.pyr[0], ws.pyr[1], ws.pyr[2], ws.pyr[3], B, h, w, LEVELS, s, fused_pairs || vplanes);
    if (st) return st;
    if (warped) {   // the zero-bordered RGBX copy the warp samples
        st = ofx_warp_pad_launch(warp_frame, ws.warp_pad, H, W, s);
        if (st) return st;
    }
    st = run_recurrence(r, ws, B, h, w, iters, false, false, flow_up, flow_low, s, prec, overlap, warped, warp_sign,
                        n_warp);
    if (st) return st;
    r->bufs.clear();
    return 0;
}

int ofx_raft_buffer(const ofx_raft* r, const char* name, void** ptr, size_t* nfloats) {
    OFX_REQUIRE(r && name && ptr && nfloats, OFX_EINVAL);
    auto it = r->bufs.find(name);
    if (it == r->bufs.end()) return OFX_EKEY;
    *ptr = it->second.first;
    *nfloats = it->second.second;
    return 0;
}

}  // extern "C"
