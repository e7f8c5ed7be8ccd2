// td_engine: weights, workspace plan and the forward schedule of the Mask R-CNN R50/R101-FPN tile predictor.
//
// Stands behind `self.model(batch_tensors)` (TreeDetection/prediction.py:182-183) with the model that
// TreeDetection/config.py:25-66 configures. Layer order and hyper-parameters follow SURVEY.md Appendix A; every
// contraction runs through conv2d_launch (MFMA implicit GEMM), everything else through the kernels of
// stem.hip / rpn.hip / roi.hip. The whole forward is asynchronous on one HIP stream: dynamic counts
// (proposals, detections) stay on the device and bound the later kernels, so there is no host sync inside.
#include "common.h"
#include "detect.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

namespace {

struct HostTensor {
    const float* data = nullptr;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct ConvLayer {
    int cout = 0, cin = 0, kh = 1, kw = 1;
    void* w = nullptr;        // [cout][kh][kw][cin], float32 or float16 (engine precision)
    float* scale = nullptr;   // [cout] or null
    float* bias = nullptr;    // [cout] or null
    bool out_f32 = false;     // fp16 engine: this layer still writes float32 (feeds the fp32 selection kernels)
    void* w_frag = nullptr;   // fp16 engine: the filters in MFMA fragment order (conv_bdirect.hip), layers with cin % 64 == 0
    float* wino_u = nullptr;  // fp32 engine, 3x3 layers: Winograd-transformed filters U [16][cout][cin] (winograd.hip)
    float* wino_u43 = nullptr;  // the same for F(4x4,3x3): U [36][cout][cin] (layers with >= 128 channels on both sides)
};

struct Block {
    ConvLayer c1, c2, c3, sc;
    bool has_sc = false;
    int stride = 1;
};

struct NamedTensor {
    void* p = nullptr;
    int64_t dims[4] = {0, 0, 0, 0};
    int elem = 4;
};

}  // namespace

struct td_engine {
    td_model_desc desc{};
    int device = 0;
    bool loaded = false;
    std::vector<void*> weight_allocs;
    std::vector<void*> ws_allocs;

    // weights
    float* stem_w = nullptr;  // [147][stem_c]
    float* stem_scale = nullptr;
    float* stem_bias = nullptr;
    unsigned short* stem_w16 = nullptr;   // fp16 engine: filter image (half bit patterns) of the MFMA stem [64][192] (stem.hip stem_mfma_kernel)
    float* stem_bias16 = nullptr;   //              bias with the mean subtraction folded in
    int stem_c = 64;
    std::vector<Block> stages[4];
    ConvLayer lateral[4], fpn_out[4];
    ConvLayer rpn_conv, rpn_head;   // head: fused 3 + 12 rows
    ConvLayer fc1, fc2, pred;       // pred: fused 2 + 4 rows
    ConvLayer mask_fcn[4], deconv;  // deconv packed [4*C][C]
    float* mask_pred_w = nullptr;
    float mask_pred_b = 0.f;
    int fpn_c = 256, fc_dim = 1024;

    // workspace (sized by reserve)
    int rB = 0, rHp = 0, rWp = 0;
    // activations (element type = engine precision)
    void *stem_out = nullptr, *pool_out = nullptr;
    void *t1[4] = {}, *t2[4] = {}, *scb[4] = {}, *res[4] = {}, *xtmp[4] = {};
    void *inner[4] = {}, *pfeat[5] = {};
    void* rpn_t = nullptr;
    float* rpn_headbuf[5] = {};
    uint32_t* key_ws = nullptr;
    float *cand_boxes = nullptr, *cand_scores = nullptr;
    int *cand_valid = nullptr, *cand_idx = nullptr;
    unsigned long long* nms_mask = nullptr;
    int *rpn_keep = nullptr, *rpn_keep_count = nullptr;
    float *props = nullptr, *prop_scores = nullptr;
    int* prop_count = nullptr;
    void *pooled7 = nullptr, *fc1_out = nullptr, *fc2_out = nullptr;
    float* pred_out = nullptr;
    float *dboxes = nullptr, *dscores = nullptr;
    int* dflags = nullptr;
    float *sboxes = nullptr, *sscores = nullptr;
    int *sidx = nullptr, *scount = nullptr, *det_keep = nullptr, *det_keep_count = nullptr;
    float* det_boxes_net = nullptr;
    void *pooled14 = nullptr, *mbuf0 = nullptr, *mbuf1 = nullptr, *deconv_out = nullptr;
    float *mask_logits = nullptr, *mask_probs_compact = nullptr;
    int* total_rows = nullptr;
    // fallback outputs when the caller passes NULL fields
    float *o_boxes = nullptr, *o_scores = nullptr, *o_mask_probs = nullptr;
    int *o_classes = nullptr, *o_count = nullptr;

    std::map<std::string, NamedTensor> named;

    // measured block-tile choice per (layer weights, rows, stride): filled lazily by the first forward of a shape
    bool autotune = true;
    int backbone_subbatch = 0;    // 0 = whole batch; else images per backbone pass (TD_BACKBONE_SUBBATCH)
    // key: (cout, cin, kh*16+kw, rows, stride*4 + out_mode*2 + has_residual) — layers of identical shape share it
    std::map<std::tuple<int, int, int, int, int>, int> tuned;
    bool winograd = true;         // TD_WINOGRAD=0 disables the Winograd path (diagnostics)
    float *wino_v = nullptr, *wino_m = nullptr;
    size_t wino_elems = 0;
    bool group_levels = true;     // TD_GROUP_LEVELS=0: the FPN output convs / the RPN conv + head as one launch per pyramid level (fp16 engine)
    bool fuse_head = true;        // TD_FUSE_HEAD=0: the RPN's 1x1 head as a launch of its own at every level (diagnostics, tests)
    bool fuse_tail = true;        // TD_FUSE_TAIL=0: conv2 / conv3 of res2 as two launches (diagnostics, tests); 2: fuse the fp16 engine's res3 too
    bool fuse_tail_fp16 = false;
    bool wino_fused = true;       // TD_WINO_FUSED=0: separate input-transform kernel + batched conv_igemm launch (diagnostics)
    int wino_slab = 0;            // TD_WINO_SLAB: tiles per slab (diagnostics); 0 = sized for the Infinity Cache
    int wino_minc = 128;          // fewest channels (both sides) of a 3x3 layer on the Winograd path (TD_WINO_MINC: experiments)
    int wino43_min = 12;          // maps at least this large on both sides take F(4x4,3x3) (TD_WINO43_MIN; 0 = never)
    // F(4x4) layers whose plane contractions and output transform run as ONE launch (wino43_fused_kernel: the M planes never reach
    // memory). A FIXED rule on the layer's own shape — per-image map size and channels, never the batch and never a timing: the
    // fold associates the transform sums differently from the three-launch form, and a batch of 8 must equal eight batches of 1
    // bit for bit. Measured (profiles/r04_wino_fold.txt, batch 8): 256 -> 256 on the 200 x 200 maps 1 007 -> 895 us, 128 -> 128
    // on 100 x 100 (res3 conv2) 103 -> 95 us; it loses where its 64-tile x 64-channel blocks cannot fill the chip (res4 / res5,
    // p3 - p6: 88 - 316 blocks on 256 CUs) and on the RPN layers, whose head rides in the three-launch form's output transform.
    // TD_WINO_FOLD (bit mask): 1 = 256 channels from 160 x 160 and 128 channels from 80 x 80 (isolated winners), 2 = the mask head
    // (device-side RoI count), 4 = 256-channel maps from 80 x 80 (FPN output 3) — the default 7: with three forwards in flight the
    // fold also pays where its isolated time ties, it frees HBM bandwidth for the other streams (bench, fp32 tiles/s: 0 → 638,
    // 1 → 650, 3 → 655, 7 → 656) —, 8 = every map from 40 x 40 and 16 = every layer the kernel can run: 660 - 661 under the
    // stream schedule but 574 → 545 one forward at a time (88-block launches on 256 CUs), not taken.
    // 32 = the RPN conv on the 160 x 160+ maps folds too (its head becomes a launch of its own).
    int wino_fold = 39;
    std::string tune_cache;       // TD_TUNE_CACHE: load / append measured choices (keeps profiled runs free of tuning launches)
    // The tuner times every candidate tile with the L2 (8 x 4 MB) emptied before each launch (a fill of this scratch on the same
    // stream): in the forward a layer's filters and most of its input are NOT in L2 — 50-odd other launches ran since — and a loop of
    // back-to-back launches of one layer otherwise favours tiles that re-read their filters (res5 conv2, fp16: the hot loop picked
    // a tile that runs 57 us in the forward against 45 us for the cold loop's choice). TD_TUNE_EVICT=0 restores the hot loop.
    void* tune_evict = nullptr;
    size_t tune_evict_bytes = 0;

    // forward context (kept between the phases of td_engine_forward_phase) and the per-phase completion events
    struct FwdCtx {
        const void* images = nullptr;
        int input_format = 0, B = 0, Hp = 0, Wp = 0;
        ImgSizes valid{}, outsz{};
        td_detections out{};
        bool valid_ctx = false;
    } ctx;
    hipEvent_t phase_ev[7] = {};           // [6] = the optional stem pre-phase (TD_PHASE_STEM)
    bool phase_ev_recorded[7] = {};
    hipEvent_t prev5_ev = nullptr;         // phase-5 event of the previous batch (the next batch's phase 3 waits for it)
    bool prev5_recorded = false;
    bool stem_done = false;                // stem + pool of the current batch already ran in the pre-phase

    // optional per-category device timing (td_engine_profile_*)
    bool prof = false;
    bool prof_group_open = false;
    struct ProfRec { hipEvent_t a, b; int cat; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_free;
    double prof_ms[TD_PROF_CATEGORIES] = {};
    int64_t prof_launches[TD_PROF_CATEGORIES] = {};
    double prof_flops[TD_PROF_CATEGORIES] = {};
    double prof_bytes[TD_PROF_CATEGORIES] = {};
    // per-class speed-of-light accounting of the contraction family (td_engine_profile_classes): what each launch EXECUTES
    // (Winograd planes: the plane products, not the direct convolution's multiplies) and the bytes its algorithm moves
    // (V / M planes included), with t_min = max(executed FLOPs / MFMA peak, bytes / achievable HBM rate) per launch;
    // measured time per class only in detail mode (td_engine_profile_enable(e, 2): an event pair per launch)
    bool prof_detail = false;
    double cls_ms[TD_PROF_CLASSES] = {};
    int64_t cls_launches[TD_PROF_CLASSES] = {};
    double cls_flops[TD_PROF_CLASSES] = {};
    double cls_bytes[TD_PROF_CLASSES] = {};
    double cls_tmin_ms[TD_PROF_CLASSES] = {};
    struct ClsRec { hipEvent_t a, b; int cls; };
    std::vector<ClsRec> cls_recs;
};

namespace {

td_status dev_alloc(std::vector<void*>& pool, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    TD_HIP_CHECK(hipMalloc(p, bytes));
    pool.push_back(*p);
    return TD_OK;
}

template <typename T>
td_status upload(td_engine* e, const std::vector<T>& h, T** d) {
    void* p = nullptr;
    td_status st = dev_alloc(e->weight_allocs, &p, h.size() * sizeof(T));
    if (st < 0) return st;
    TD_HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *d = static_cast<T*>(p);
    return TD_OK;
}

// conv / fc weight matrix in the engine's precision (float16: round-to-nearest-even on the host)
td_status upload_w(td_engine* e, const std::vector<float>& h, void** d) {
    if (e->desc.precision == TD_PRECISION_FP16) {
        std::vector<_Float16> hh(h.size());
        for (size_t i = 0; i < h.size(); ++i) hh[i] = (_Float16)h[i];
        _Float16* p = nullptr;
        td_status st = upload(e, hh, &p);
        *d = p;
        return st;
    }
    float* p = nullptr;
    td_status st = upload(e, h, &p);
    *d = p;
    return st;
}

// second copy of a filter bank in MFMA fragment order for conv_bd_kernel (tile ids 23 - 27), both precisions
td_status upload_frag(td_engine* e, const std::vector<float>& h, ConvLayer& L) {
    L.w_frag = nullptr;
    static const bool off = getenv("TD_BDIRECT") && atoi(getenv("TD_BDIRECT")) == 0;
    const bool f16 = e->desc.precision == TD_PRECISION_FP16;
    const int ke = f16 ? 64 : 32;
    if (off || L.cin % ke != 0 || L.kh * L.kw > 32 || (size_t)L.cout * L.kh * L.kw * L.cin != h.size()) return TD_OK;
    std::vector<unsigned char> packed;
    if (f16) {
        std::vector<unsigned short> bits(h.size());
        for (size_t i = 0; i < h.size(); ++i) bits[i] = __builtin_bit_cast(unsigned short, (_Float16)h[i]);
        conv_bd_pack(bits.data(), 2, L.cout, L.kh, L.kw, L.cin, packed);
    } else {
        conv_bd_pack(h.data(), 4, L.cout, L.kh, L.kw, L.cin, packed);
    }
    unsigned char* d = nullptr;
    td_status st = upload(e, packed, &d);
    L.w_frag = d;
    return st;
}

using TensorMap = std::map<std::string, HostTensor>;

td_status need(const TensorMap& tm, const std::string& name, int ndim, const HostTensor** out) {
    auto it = tm.find(name);
    if (it == tm.end()) {
        td_set_error("load_weights: tensor '%s' missing", name.c_str());
        return TD_ERR_WEIGHTS;
    }
    if ((int)it->second.shape.size() != ndim) {
        td_set_error("load_weights: tensor '%s' has %d dims, expected %d", name.c_str(), (int)it->second.shape.size(), ndim);
        return TD_ERR_WEIGHTS;
    }
    *out = &it->second;
    return TD_OK;
}

// [Cout,Cin,KH,KW] → [Cout][KH][KW][Cin]
std::vector<float> pack_ohwi(const HostTensor& t) {
    const int co = (int)t.shape[0], ci = (int)t.shape[1], kh = (int)t.shape[2], kw = (int)t.shape[3];
    std::vector<float> out((size_t)co * ci * kh * kw);
    for (int o = 0; o < co; ++o)
        for (int c = 0; c < ci; ++c)
            for (int y = 0; y < kh; ++y)
                for (int x = 0; x < kw; ++x)
                    out[(((size_t)o * kh + y) * kw + x) * ci + c] = t.data[(((size_t)o * ci + c) * kh + y) * kw + x];
    return out;
}

// FrozenBN fold: scale = gamma * (1/sqrt(var + eps)), bias = beta - mean * scale (float32, like the oracle)
td_status bn_fold(const TensorMap& tm, const std::string& p, int c, std::vector<float>& scale, std::vector<float>& bias) {
    const HostTensor *g, *b, *m, *v;
    td_status st;
    if ((st = need(tm, p + ".norm.weight", 1, &g)) < 0) return st;
    if ((st = need(tm, p + ".norm.bias", 1, &b)) < 0) return st;
    if ((st = need(tm, p + ".norm.running_mean", 1, &m)) < 0) return st;
    if ((st = need(tm, p + ".norm.running_var", 1, &v)) < 0) return st;
    if (g->shape[0] != c || b->shape[0] != c || m->shape[0] != c || v->shape[0] != c) {
        td_set_error("load_weights: norm tensors of '%s' do not have %d channels", p.c_str(), c);
        return TD_ERR_WEIGHTS;
    }
    scale.resize(c);
    bias.resize(c);
    for (int i = 0; i < c; ++i) {
        const float r = 1.0f / std::sqrt(v->data[i] + 1e-5f);
        const float s = g->data[i] * r;
        const float ms = m->data[i] * s;
        scale[i] = s;
        bias[i] = b->data[i] - ms;
    }
    return TD_OK;
}

// Tiles per Winograd slab. A layer can be walked in slabs of tiles whose V and M planes would stay inside the 256 MiB
// Infinity Cache (TD_WINO_SLAB = tiles per slab); measured on the bench (round 2, fp32, 3 batches in flight): 4096-tile
// slabs 425 tiles/s, 8192 441, whole layers 456 — the shorter launches lose more to their tails and boundaries than the
// on-die hand-over of V / M saves, so the default is one slab = the whole layer.
int wino_slab_tiles(const td_engine* e, const ConvLayer& L) {
    (void)L;
    return e->wino_slab > 0 ? e->wino_slab : (1 << 30);
}

// fp32 engine: Winograd F(2x2,3x3) filter bank of a 3x3 layer (used where the engine measures it faster, run_conv)
td_status upload_wino(td_engine* e, const std::vector<float>& w_ohwi, ConvLayer& L) {
    L.wino_u = nullptr;
    if (e->desc.precision != TD_PRECISION_FP32 || !e->winograd || L.kh != 3 || L.kw != 3 || L.cin % 32 != 0 || L.cout % 4 != 0)
        return TD_OK;
    std::vector<float> u((size_t)16 * L.cout * L.cin);
    wino_filter_transform(w_ohwi.data(), L.cout, L.cin, u.data());
    td_status st = upload(e, u, &L.wino_u);
    if (st < 0 || e->wino43_min <= 0 || L.cin < e->wino_minc || L.cout < e->wino_minc) return st;
    std::vector<float> u43((size_t)36 * L.cout * L.cin);
    wino43_filter_transform(w_ohwi.data(), L.cout, L.cin, u43.data());
    return upload(e, u43, &L.wino_u43);
}

td_status load_conv_bn(td_engine* e, const TensorMap& tm, const std::string& p, ConvLayer& L) {
    const HostTensor* w;
    td_status st = need(tm, p + ".weight", 4, &w);
    if (st < 0) return st;
    L.cout = (int)w->shape[0];
    L.cin = (int)w->shape[1];
    L.kh = (int)w->shape[2];
    L.kw = (int)w->shape[3];
    const std::vector<float> packed = pack_ohwi(*w);
    if ((st = upload_w(e, packed, &L.w)) < 0) return st;
    if ((st = upload_frag(e, packed, L)) < 0) return st;
    if ((st = upload_wino(e, packed, L)) < 0) return st;
    std::vector<float> s, b;
    if ((st = bn_fold(tm, p, L.cout, s, b)) < 0) return st;
    if ((st = upload(e, s, &L.scale)) < 0) return st;
    return upload(e, b, &L.bias);
}

td_status load_conv_bias(td_engine* e, const TensorMap& tm, const std::string& p, ConvLayer& L) {
    const HostTensor *w, *b;
    td_status st = need(tm, p + ".weight", 4, &w);
    if (st < 0) return st;
    if ((st = need(tm, p + ".bias", 1, &b)) < 0) return st;
    L.cout = (int)w->shape[0];
    L.cin = (int)w->shape[1];
    L.kh = (int)w->shape[2];
    L.kw = (int)w->shape[3];
    if (b->shape[0] != L.cout) {
        td_set_error("load_weights: bias of '%s' has wrong length", p.c_str());
        return TD_ERR_WEIGHTS;
    }
    const std::vector<float> packed = pack_ohwi(*w);
    if ((st = upload_w(e, packed, &L.w)) < 0) return st;
    if ((st = upload_frag(e, packed, L)) < 0) return st;
    if ((st = upload_wino(e, packed, L)) < 0) return st;
    L.scale = nullptr;
    return upload(e, std::vector<float>(b->data, b->data + L.cout), &L.bias);
}

td_status run_conv_raw(const ConvLayer& L, const void* x, int B, int H, int W, int stride, int pad, bool relu, void* y,
                       const void* res, int res_shift, hipStream_t s, int precision, const int* m_dyn, int m_mul,
                       int out_mode, int tile_cfg = -1, const ConvLayer* head = nullptr, float* head_y = nullptr) {
    ConvArgs a{};
    if (head) {          // fused 1x1 head (ConvArgs::head_w): the caller checked conv_head_capable(tile_cfg)
        a.head_w = head->w; a.head_b = head->bias; a.head_y = head_y; a.head_n = head->cout;
    }
    a.x = x; a.w = L.w; a.w_frag = L.w_frag; a.scale = L.scale; a.bias = L.bias; a.res = res; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = L.cin; a.Cout = L.cout; a.KH = L.kh; a.KW = L.kw;
    a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - L.kh) / stride + 1;
    a.Wo = (W + 2 * pad - L.kw) / stride + 1;
    a.res_shift = res_shift; a.relu = relu ? 1 : 0; a.out_mode = out_mode;
    a.M = B * a.Ho * a.Wo; a.m_dyn = m_dyn; a.m_mul = m_mul; a.tile_cfg = tile_cfg; a.out_f32 = L.out_f32 ? 1 : 0;
    return conv2d_launch(a, precision, s);
}

void set_named(td_engine* e, const char* name, void* p, int64_t d0, int64_t d1 = 0, int64_t d2 = 0, int64_t d3 = 0,
               int elem = 4) {
    NamedTensor t;
    t.p = p;
    t.dims[0] = d0; t.dims[1] = d1; t.dims[2] = d2; t.dims[3] = d3;
    t.elem = elem;
    e->named[name] = t;
}

hipEvent_t prof_event(td_engine* e) {
    hipEvent_t ev = nullptr;
    if (!e->prof_free.empty()) {
        ev = e->prof_free.back();
        e->prof_free.pop_back();
    } else {
        (void)hipEventCreate(&ev);
    }
    return ev;
}

// RAII bracket: records start on construction and stop on destruction (only when profiling is on). A GROUP scope
// brackets a run of back-to-back launches of one category with ONE event pair (an event per launch costs ≈3 % of
// the fp32 step and ≈10 % of the fp16 step in extra kernel boundaries); launches inside it only add their counts.
struct ProfScope {
    td_engine* e;
    hipStream_t s;
    hipEvent_t a = nullptr;
    int cat;
    bool group;
    ProfScope(td_engine* e_, hipStream_t s_, int cat_, double flops = 0.0, double bytes = 0.0, bool group_ = false)
        : e(e_), s(s_), cat(cat_), group(group_) {
        if (!e->prof) return;
        if (!group) {
            e->prof_flops[cat] += flops;
            e->prof_bytes[cat] += bytes;
            e->prof_launches[cat] += 1;
        }
        if (e->prof_group_open) return;                // counted; the enclosing group owns the events (groups do not nest)
        a = prof_event(e);
        (void)hipEventRecord(a, s);
        if (group) e->prof_group_open = true;
    }
    ~ProfScope() {
        if (!a) return;
        hipEvent_t b = prof_event(e);
        (void)hipEventRecord(b, s);
        e->prof_recs.push_back({a, b, cat});
        if (group) e->prof_group_open = false;
    }
};

// One launch of the contraction family under its speed-of-light class: always adds the model side (executed FLOPs,
// bytes the chosen algorithm moves, t_min) while profiling is on; in detail mode also an event pair of its own.
struct ClassScope {
    td_engine* e;
    hipStream_t s;
    hipEvent_t a = nullptr;
    int cls;
    ClassScope(td_engine* e_, hipStream_t s_, int cls_, double exec_flops, double bytes) : e(e_), s(s_), cls(cls_) {
        if (!e->prof) return;
        const double peak = (e->desc.precision == TD_PRECISION_FP16 ? TD_PEAK_F16_MFMA_TFLOPS : TD_PEAK_F32_MFMA_TFLOPS) * 1e12;
        e->cls_launches[cls] += 1;
        e->cls_flops[cls] += exec_flops;
        e->cls_bytes[cls] += bytes;
        e->cls_tmin_ms[cls] += 1e3 * std::max(exec_flops / peak, bytes / (TD_HBM_ACHIEVABLE_TBS * 1e12));
        if (!e->prof_detail) return;
        a = prof_event(e);
        (void)hipEventRecord(a, s);
    }
    ~ClassScope() {
        if (!a) return;
        hipEvent_t b = prof_event(e);
        (void)hipEventRecord(b, s);
        e->cls_recs.push_back({a, b, cls});
    }
};

// Measured block-tile choices shared between engines (and runs) through the TD_TUNE_CACHE file: read at creation and
// again whenever a layer shape is missing, so engines created together pick up what the first one measured.
void load_tune_cache(td_engine* e) {
    if (e->tune_cache.empty()) return;
    if (FILE* f = fopen(e->tune_cache.c_str(), "r")) {
        int pr, a0, a1, a2, a3, a4, cfg;
        while (fscanf(f, "%d %d %d %d %d %d %d", &pr, &a0, &a1, &a2, &a3, &a4, &cfg) == 7) {
            if (pr == e->desc.precision && cfg >= 0 && cfg <= TD_CONV_TILE_CFG_MAX) e->tuned[std::make_tuple(a0, a1, a2, a3, a4)] = cfg;
        }
        fclose(f);
    }
}

void free_pool(std::vector<void*>& pool) {
    for (void* p : pool) (void)hipFree(p);
    pool.clear();
}

}  // namespace

extern "C" {

td_status td_engine_create(const td_model_desc* desc, int device, td_engine** out) {
    TD_REQUIRE(out, "td_engine_create: out is NULL");
    td_model_desc d;
    if (desc) d = *desc; else td_model_desc_default(&d);
    TD_REQUIRE(d.num_classes == 1, "td_engine_create: num_classes=%d (the reference configures exactly 1, config.py:35)", d.num_classes);
    TD_REQUIRE(d.precision == TD_PRECISION_FP32 || d.precision == TD_PRECISION_FP16, "td_engine_create: bad precision %d", d.precision);
    TD_REQUIRE(d.pre_nms_topk >= 1 && d.pre_nms_topk <= RPN_CAND, "td_engine_create: pre_nms_topk must be in [1,%d]", RPN_CAND);
    TD_REQUIRE(d.post_nms_topk >= 1 && d.post_nms_topk <= 1024, "td_engine_create: post_nms_topk must be in [1,1024]");
    TD_REQUIRE(d.detections_per_image >= 1 && d.detections_per_image <= 1024, "td_engine_create: detections_per_image must be in [1,1024]");
    int ndev = 0;
    TD_HIP_CHECK(hipGetDeviceCount(&ndev));
    TD_REQUIRE(device >= 0 && device < ndev, "td_engine_create: device %d of %d", device, ndev);
    TD_HIP_CHECK(hipSetDevice(device));
    td_engine* e = new td_engine();
    if (const char* sbenv = getenv("TD_BACKBONE_SUBBATCH")) e->backbone_subbatch = atoi(sbenv);
    if (const char* tc = getenv("TD_TUNE_CACHE")) e->tune_cache = tc;
    if (const char* wg = getenv("TD_WINOGRAD")) e->winograd = atoi(wg) != 0;
    if (const char* ws = getenv("TD_WINO_SLAB")) e->wino_slab = atoi(ws);
    if (const char* wf = getenv("TD_WINO_FUSED")) e->wino_fused = atoi(wf) != 0;
    if (const char* fh = getenv("TD_FUSE_HEAD")) e->fuse_head = atoi(fh) != 0;
    if (const char* gl = getenv("TD_GROUP_LEVELS")) e->group_levels = atoi(gl) != 0;
    if (const char* ft = getenv("TD_FUSE_TAIL")) { e->fuse_tail = atoi(ft) != 0; e->fuse_tail_fp16 = atoi(ft) == 2; }
    if (const char* w4 = getenv("TD_WINO43_MIN")) e->wino43_min = atoi(w4);
    if (const char* wf2 = getenv("TD_WINO_FOLD")) e->wino_fold = atoi(wf2);
    if (const char* wc = getenv("TD_WINO_MINC")) e->wino_minc = atoi(wc);
    e->desc = d;
    load_tune_cache(e);
    e->device = device;
    *out = e;
    return TD_OK;
}

void td_engine_destroy(td_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    free_pool(e->weight_allocs);
    free_pool(e->ws_allocs);
    if (e->tune_evict) (void)hipFree(e->tune_evict);
    for (auto& r : e->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto& r : e->cls_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto ev : e->prof_free) (void)hipEventDestroy(ev);
    for (auto ev : e->phase_ev) if (ev) (void)hipEventDestroy(ev);
    if (e->prev5_ev) (void)hipEventDestroy(e->prev5_ev);
    delete e;
}

td_status td_engine_load_weights(td_engine* e, const td_tensor_desc* tensors, size_t n) {
    TD_REQUIRE(e && tensors, "td_engine_load_weights: null argument");
    TD_HIP_CHECK(hipSetDevice(e->device));
    TensorMap tm;
    for (size_t i = 0; i < n; ++i) {
        TD_REQUIRE(tensors[i].name && tensors[i].data && tensors[i].ndim >= 1 && tensors[i].ndim <= 4,
                   "td_engine_load_weights: bad tensor descriptor %zu", i);
        HostTensor t;
        t.data = tensors[i].data;
        t.shape.assign(tensors[i].shape, tensors[i].shape + tensors[i].ndim);
        tm[tensors[i].name] = t;
    }
    free_pool(e->weight_allocs);
    e->loaded = false;
    td_status st;
    // stem: [C,3,7,7] → [k = (ky*7+kx)*3 + c][C]
    {
        const HostTensor* w;
        if ((st = need(tm, "backbone.bottom_up.stem.conv1.weight", 4, &w)) < 0) return st;
        if (w->shape[1] != 3 || w->shape[2] != 7 || w->shape[3] != 7) {
            td_set_error("load_weights: stem conv must be [C,3,7,7] (RGB+nDSM tiles still feed 3 channels, prediction.py:166)");
            return TD_ERR_WEIGHTS;
        }
        e->stem_c = (int)w->shape[0];
        std::vector<float> p((size_t)147 * e->stem_c);
        for (int co = 0; co < e->stem_c; ++co)
            for (int c = 0; c < 3; ++c)
                for (int ky = 0; ky < 7; ++ky)
                    for (int kx = 0; kx < 7; ++kx)
                        p[(size_t)((ky * 7 + kx) * 3 + c) * e->stem_c + co] = w->data[(((size_t)co * 3 + c) * 7 + ky) * 7 + kx];
        if ((st = upload(e, p, &e->stem_w)) < 0) return st;
        std::vector<float> s, b;
        if ((st = bn_fold(tm, "backbone.bottom_up.stem.conv1", e->stem_c, s, b)) < 0) return st;
        if ((st = upload(e, s, &e->stem_scale)) < 0) return st;
        if ((st = upload(e, b, &e->stem_bias)) < 0) return st;
        const char* mfma_stem = getenv("TD_STEM_MFMA");              // 0 = the VALU stem for uint8 inputs too (diagnostics, tests)
        if (e->desc.precision == TD_PRECISION_FP16 && e->stem_c == 64 && !(mfma_stem && atoi(mfma_stem) == 0)) {
            std::vector<unsigned short> w16;
            std::vector<float> b16;
            stem_mfma_prepare(p.data(), s.data(), b.data(), e->stem_c, w16, b16);
            if ((st = upload(e, w16, &e->stem_w16)) < 0) return st;
            if ((st = upload(e, b16, &e->stem_bias16)) < 0) return st;
        }
    }
    for (int si = 0; si < 4; ++si) {
        e->stages[si].clear();
        for (int bi = 0;; ++bi) {
            const std::string p = "backbone.bottom_up.res" + std::to_string(si + 2) + "." + std::to_string(bi);
            if (tm.find(p + ".conv1.weight") == tm.end()) break;
            Block blk;
            blk.stride = (bi == 0 && si > 0) ? 2 : 1;
            if ((st = load_conv_bn(e, tm, p + ".conv1", blk.c1)) < 0) return st;
            if ((st = load_conv_bn(e, tm, p + ".conv2", blk.c2)) < 0) return st;
            if ((st = load_conv_bn(e, tm, p + ".conv3", blk.c3)) < 0) return st;
            blk.has_sc = tm.find(p + ".shortcut.weight") != tm.end();
            if (blk.has_sc && (st = load_conv_bn(e, tm, p + ".shortcut", blk.sc)) < 0) return st;
            e->stages[si].push_back(blk);
        }
        if (e->stages[si].empty()) {
            td_set_error("load_weights: no blocks found for res%d", si + 2);
            return TD_ERR_WEIGHTS;
        }
    }
    for (int l = 0; l < 4; ++l) {
        if ((st = load_conv_bias(e, tm, "backbone.fpn_lateral" + std::to_string(l + 2), e->lateral[l])) < 0) return st;
        if ((st = load_conv_bias(e, tm, "backbone.fpn_output" + std::to_string(l + 2), e->fpn_out[l])) < 0) return st;
    }
    e->fpn_c = e->lateral[0].cout;
    if ((st = load_conv_bias(e, tm, "proposal_generator.rpn_head.conv", e->rpn_conv)) < 0) return st;
    {
        const HostTensor *wo, *bo, *wd, *bd;
        if ((st = need(tm, "proposal_generator.rpn_head.objectness_logits.weight", 4, &wo)) < 0) return st;
        if ((st = need(tm, "proposal_generator.rpn_head.objectness_logits.bias", 1, &bo)) < 0) return st;
        if ((st = need(tm, "proposal_generator.rpn_head.anchor_deltas.weight", 4, &wd)) < 0) return st;
        if ((st = need(tm, "proposal_generator.rpn_head.anchor_deltas.bias", 1, &bd)) < 0) return st;
        if (wo->shape[0] != RPN_A || wd->shape[0] != 4 * RPN_A || wo->shape[1] != e->fpn_c || wd->shape[1] != e->fpn_c) {
            td_set_error("load_weights: RPN head must have %d anchors per position", RPN_A);
            return TD_ERR_WEIGHTS;
        }
        const int c = e->fpn_c;
        std::vector<float> w((size_t)RPN_HEAD_C * c), b(RPN_HEAD_C);
        std::memcpy(w.data(), wo->data, sizeof(float) * RPN_A * c);
        std::memcpy(w.data() + (size_t)RPN_A * c, wd->data, sizeof(float) * 4 * RPN_A * c);
        std::memcpy(b.data(), bo->data, sizeof(float) * RPN_A);
        std::memcpy(b.data() + RPN_A, bd->data, sizeof(float) * 4 * RPN_A);
        e->rpn_head.cout = RPN_HEAD_C; e->rpn_head.cin = c; e->rpn_head.kh = e->rpn_head.kw = 1;
        e->rpn_head.out_f32 = true;
        if ((st = upload_w(e, w, &e->rpn_head.w)) < 0) return st;
        if ((st = upload_frag(e, w, e->rpn_head)) < 0) return st;
        if ((st = upload(e, b, &e->rpn_head.bias)) < 0) return st;
    }
    {   // fc1: input index c*49 + y*7 + x → (y*7 + x)*C + c (RoIAlign writes NHWC rows)
        const HostTensor *w, *b;
        if ((st = need(tm, "roi_heads.box_head.fc1.weight", 2, &w)) < 0) return st;
        if ((st = need(tm, "roi_heads.box_head.fc1.bias", 1, &b)) < 0) return st;
        const int c = e->fpn_c, o = (int)w->shape[0];
        if (w->shape[1] != (int64_t)c * 49) {
            td_set_error("load_weights: fc1 expects %d inputs", c * 49);
            return TD_ERR_WEIGHTS;
        }
        std::vector<float> p((size_t)o * c * 49);
        for (int r = 0; r < o; ++r)
            for (int ch = 0; ch < c; ++ch)
                for (int q = 0; q < 49; ++q) p[(size_t)r * c * 49 + (size_t)q * c + ch] = w->data[(size_t)r * c * 49 + (size_t)ch * 49 + q];
        e->fc1.cout = o; e->fc1.cin = c * 49; e->fc1.kh = e->fc1.kw = 1;
        e->fc_dim = o;
        if ((st = upload_w(e, p, &e->fc1.w)) < 0) return st;
        if ((st = upload_frag(e, p, e->fc1)) < 0) return st;
        if ((st = upload(e, std::vector<float>(b->data, b->data + o), &e->fc1.bias)) < 0) return st;
    }
    {
        const HostTensor *w, *b;
        if ((st = need(tm, "roi_heads.box_head.fc2.weight", 2, &w)) < 0) return st;
        if ((st = need(tm, "roi_heads.box_head.fc2.bias", 1, &b)) < 0) return st;
        e->fc2.cout = (int)w->shape[0]; e->fc2.cin = (int)w->shape[1]; e->fc2.kh = e->fc2.kw = 1;
        const std::vector<float> w2v(w->data, w->data + w->numel());
        if ((st = upload_w(e, w2v, &e->fc2.w)) < 0) return st;
        if ((st = upload_frag(e, w2v, e->fc2)) < 0) return st;
        if ((st = upload(e, std::vector<float>(b->data, b->data + e->fc2.cout), &e->fc2.bias)) < 0) return st;
    }
    {
        const HostTensor *wc, *bc, *wr, *br;
        if ((st = need(tm, "roi_heads.box_predictor.cls_score.weight", 2, &wc)) < 0) return st;
        if ((st = need(tm, "roi_heads.box_predictor.cls_score.bias", 1, &bc)) < 0) return st;
        if ((st = need(tm, "roi_heads.box_predictor.bbox_pred.weight", 2, &wr)) < 0) return st;
        if ((st = need(tm, "roi_heads.box_predictor.bbox_pred.bias", 1, &br)) < 0) return st;
        if (wc->shape[0] != 2 || wr->shape[0] != 4) {
            td_set_error("load_weights: box predictor must be 1 class (+background), got cls %lld / reg %lld rows",
                         (long long)wc->shape[0], (long long)wr->shape[0]);
            return TD_ERR_WEIGHTS;
        }
        const int k = (int)wc->shape[1];
        std::vector<float> w((size_t)6 * k), b(6);
        std::memcpy(w.data(), wc->data, sizeof(float) * 2 * k);
        std::memcpy(w.data() + (size_t)2 * k, wr->data, sizeof(float) * 4 * k);
        std::memcpy(b.data(), bc->data, sizeof(float) * 2);
        std::memcpy(b.data() + 2, br->data, sizeof(float) * 4);
        e->pred.cout = 6; e->pred.cin = k; e->pred.kh = e->pred.kw = 1;
        e->pred.out_f32 = true;
        if ((st = upload_w(e, w, &e->pred.w)) < 0) return st;
        if ((st = upload_frag(e, w, e->pred)) < 0) return st;
        if ((st = upload(e, b, &e->pred.bias)) < 0) return st;
    }
    for (int i = 0; i < 4; ++i)
        if ((st = load_conv_bias(e, tm, "roi_heads.mask_head.mask_fcn" + std::to_string(i + 1), e->mask_fcn[i])) < 0) return st;
    {   // ConvTranspose2d weight [Cin,Cout,2,2] → rows n = (dy*2+dx)*Cout + co over Cin
        const HostTensor *w, *b;
        if ((st = need(tm, "roi_heads.mask_head.deconv.weight", 4, &w)) < 0) return st;
        if ((st = need(tm, "roi_heads.mask_head.deconv.bias", 1, &b)) < 0) return st;
        const int ci = (int)w->shape[0], co = (int)w->shape[1];
        if (w->shape[2] != 2 || w->shape[3] != 2) {
            td_set_error("load_weights: mask deconv must be 2x2");
            return TD_ERR_WEIGHTS;
        }
        std::vector<float> p((size_t)4 * co * ci);
        for (int i = 0; i < ci; ++i)
            for (int o = 0; o < co; ++o)
                for (int q = 0; q < 4; ++q) p[((size_t)q * co + o) * ci + i] = w->data[((size_t)i * co + o) * 4 + q];
        e->deconv.cout = 4 * co; e->deconv.cin = ci; e->deconv.kh = e->deconv.kw = 1;
        if ((st = upload_w(e, p, &e->deconv.w)) < 0) return st;
        if ((st = upload(e, std::vector<float>(b->data, b->data + co), &e->deconv.bias)) < 0) return st;
    }
    {
        const HostTensor *w, *b;
        if ((st = need(tm, "roi_heads.mask_head.predictor.weight", 4, &w)) < 0) return st;
        if ((st = need(tm, "roi_heads.mask_head.predictor.bias", 1, &b)) < 0) return st;
        if (w->shape[0] != 1) {
            td_set_error("load_weights: mask predictor must have 1 output channel");
            return TD_ERR_WEIGHTS;
        }
        if ((st = upload(e, std::vector<float>(w->data, w->data + w->shape[1]), &e->mask_pred_w)) < 0) return st;
        e->mask_pred_b = b->data[0];
    }
    e->loaded = true;
    return TD_OK;
}

td_status td_engine_reserve(td_engine* e, int B, int Hp, int Wp) {
    TD_REQUIRE(e, "td_engine_reserve: null engine");
    TD_REQUIRE(e->loaded, "td_engine_reserve: load weights first");
    TD_REQUIRE(B >= 1 && B <= TD_MAX_BATCH, "td_engine_reserve: batch %d not in [1,%d]", B, TD_MAX_BATCH);
    TD_REQUIRE(Hp >= 64 && Wp >= 64 && Hp % 32 == 0 && Wp % 32 == 0, "td_engine_reserve: padded size %dx%d must be multiples of 32 (>= 64)", Hp, Wp);
    if (B <= e->rB && Hp <= e->rHp && Wp <= e->rWp) return TD_OK;   // current plan already covers it
    B = B > e->rB ? B : e->rB;
    Hp = Hp > e->rHp ? Hp : e->rHp;
    Wp = Wp > e->rWp ? Wp : e->rWp;
    TD_HIP_CHECK(hipSetDevice(e->device));
    free_pool(e->ws_allocs);
    e->rB = e->rHp = e->rWp = 0;
    auto A = [&](auto** p, size_t elems) -> td_status {
        void* q = nullptr;
        td_status st = dev_alloc(e->ws_allocs, &q, elems * sizeof(**p));
        *p = static_cast<std::remove_reference_t<decltype(*p)>>(q);
        return st;
    };
    const size_t es = e->desc.precision == TD_PRECISION_FP16 ? 2 : 4;
    auto AA = [&](void** p, size_t elems) -> td_status { return dev_alloc(e->ws_allocs, p, elems * es); };
    td_status st;
    const size_t b = B;
    const int H2 = Hp / 4, W2 = Wp / 4;
    if ((st = AA(&e->stem_out, b * (Hp / 2) * (Wp / 2) * e->stem_c)) < 0) return st;
    if ((st = AA(&e->pool_out, b * H2 * W2 * e->stem_c)) < 0) return st;
    int hs[5], wsz[5];
    for (int l = 0; l < 4; ++l) {
        hs[l] = Hp >> (l + 2);
        wsz[l] = Wp >> (l + 2);
    }
    hs[4] = (hs[3] - 1) / 2 + 1;
    wsz[4] = (wsz[3] - 1) / 2 + 1;
    for (int s = 0; s < 4; ++s) {
        const size_t px = b * hs[s] * wsz[s];
        const int mid = e->stages[s][0].c1.cout, co = e->stages[s][0].c3.cout;
        if ((st = AA(&e->t1[s], px * mid)) < 0) return st;
        if ((st = AA(&e->t2[s], px * mid)) < 0) return st;
        if ((st = AA(&e->scb[s], px * co)) < 0) return st;
        if ((st = AA(&e->res[s], px * co)) < 0) return st;
        if ((st = AA(&e->xtmp[s], px * co)) < 0) return st;
        if ((st = AA(&e->inner[s], px * e->fpn_c)) < 0) return st;
    }
    size_t total_anchors = 0;
    for (int l = 0; l < 5; ++l) {
        const size_t px = b * hs[l] * wsz[l];
        if ((st = AA(&e->pfeat[l], px * e->fpn_c)) < 0) return st;
        if ((st = A(&e->rpn_headbuf[l], px * RPN_HEAD_C)) < 0) return st;
        total_anchors += (size_t)hs[l] * wsz[l] * RPN_A;
    }
    if ((st = AA(&e->rpn_t, b * hs[0] * wsz[0] * e->fpn_c)) < 0) return st;
    if ((st = A(&e->key_ws, rpn_topk_ws_elems((int)b, (int)total_anchors))) < 0) return st;
    const size_t nc = b * RPN_LEVELS * RPN_CAND;
    if ((st = A(&e->cand_boxes, nc * 4)) < 0) return st;
    if ((st = A(&e->cand_scores, nc)) < 0) return st;
    if ((st = A(&e->cand_valid, nc)) < 0) return st;
    if ((st = A(&e->cand_idx, nc)) < 0) return st;
    if ((st = A(&e->nms_mask, nc * (RPN_CAND / 64))) < 0) return st;
    if ((st = A(&e->rpn_keep, nc)) < 0) return st;
    if ((st = A(&e->rpn_keep_count, b * RPN_LEVELS)) < 0) return st;
    const int P = e->desc.post_nms_topk, D = e->desc.detections_per_image;
    if ((st = A(&e->props, b * P * 4)) < 0) return st;
    if ((st = A(&e->prop_scores, b * P)) < 0) return st;
    if ((st = A(&e->prop_count, b)) < 0) return st;
    if ((st = AA(&e->pooled7, b * P * 49 * e->fpn_c)) < 0) return st;
    if ((st = AA(&e->fc1_out, b * P * e->fc_dim)) < 0) return st;
    if ((st = AA(&e->fc2_out, b * P * e->fc_dim)) < 0) return st;
    if ((st = A(&e->pred_out, b * P * 6)) < 0) return st;
    if ((st = A(&e->dboxes, b * P * 4)) < 0) return st;
    if ((st = A(&e->dscores, b * P)) < 0) return st;
    if ((st = A(&e->dflags, b * P)) < 0) return st;
    if ((st = A(&e->sboxes, b * P * 4)) < 0) return st;
    if ((st = A(&e->sscores, b * P)) < 0) return st;
    if ((st = A(&e->sidx, b * P)) < 0) return st;
    if ((st = A(&e->scount, b)) < 0) return st;
    if ((st = A(&e->det_keep, b * D)) < 0) return st;
    if ((st = A(&e->det_keep_count, b)) < 0) return st;
    if ((st = A(&e->det_boxes_net, b * D * 4)) < 0) return st;
    const size_t mrows = b * D;
    if ((st = AA(&e->pooled14, mrows * 196 * e->fpn_c)) < 0) return st;
    if ((st = AA(&e->mbuf0, mrows * 196 * e->fpn_c)) < 0) return st;
    if ((st = AA(&e->mbuf1, mrows * 196 * e->fpn_c)) < 0) return st;
    if ((st = AA(&e->deconv_out, mrows * 784 * (e->deconv.cout / 4))) < 0) return st;
    if ((st = A(&e->mask_logits, mrows * 784)) < 0) return st;
    if ((st = A(&e->mask_probs_compact, mrows * 784)) < 0) return st;
    if ((st = A(&e->total_rows, 4)) < 0) return st;
    if ((st = A(&e->o_boxes, b * D * 4)) < 0) return st;
    if ((st = A(&e->o_scores, b * D)) < 0) return st;
    if ((st = A(&e->o_classes, b * D)) < 0) return st;
    if ((st = A(&e->o_count, b)) < 0) return st;
    if ((st = A(&e->o_mask_probs, mrows * 784)) < 0) return st;
    e->wino_v = e->wino_m = nullptr;
    e->wino_elems = 0;
    if (e->desc.precision == TD_PRECISION_FP32 && e->winograd) {
        // Winograd workspace: 16 planes of [tiles][channels] for V and for M of the largest 3x3 layer that takes the path
        // (FPN / RPN at p2, the bottleneck conv2 of every stage, the mask head over B*D RoIs of 14x14)
        auto tiles = [](int h, int w) { return (size_t)((h + 1) / 2) * ((w + 1) / 2); };
        size_t need = 0;
        for (int l = 0; l < 5; ++l) need = std::max(need, b * tiles(hs[l], wsz[l]) * (size_t)e->fpn_c);
        for (int s = 0; s < 4; ++s) need = std::max(need, b * tiles(hs[s], wsz[s]) * (size_t)e->stages[s][0].c2.cout);
        need = std::max(need, mrows * tiles(14, 14) * (size_t)e->fpn_c);
        e->wino_elems = 16 * need;
        if ((st = A(&e->wino_v, e->wino_elems)) < 0) return st;
        if ((st = A(&e->wino_m, e->wino_elems)) < 0) return st;
    }
    e->rB = B;
    e->rHp = Hp;
    e->rWp = Wp;
    return TD_OK;
}

}  // extern "C"

namespace {

// The forward in six phases. Contractions (0 trunk: stem..RPN heads, 2 box-head FCs, 4 mask-head convs) and the
// low-occupancy selection work (1 RPN top-k/NMS/merge + RoIAlign 7x7, 3 detections + RoIAlign 14x14, 5 mask
// predictor/scatter/paste) alternate, so a caller can keep three batches in flight: contraction phases of successive
// batches back to back on a main stream, selection phases on a side stream where they overlap the next contractions.
td_status forward_impl(td_engine* e, unsigned phase_mask, hipStream_t s) {
    const td_engine::FwdCtx& c = e->ctx;
    const void* images = c.images;
    const int input_format = c.input_format, B = c.B, Hp = c.Hp, Wp = c.Wp;
    const ImgSizes& valid = c.valid;
    const ImgSizes& outsz = c.outsz;
    const td_detections* out = &c.out;
    const int prec = e->desc.precision;
    auto PH = [&](int k) { return (phase_mask >> k) & 1u; };
    if (PH(0)) e->named.clear();
    td_status st;
    // Times `launch(cfg)` for every candidate block tile on the live buffers (idempotent: same inputs, same outputs) and
    // returns the fastest; launches shorter than ~100 us are timed again over a longer run (event granularity and clock
    // ramps otherwise pick the wrong tile for them).
    auto time_launch = [&](hipStream_t s_, hipEvent_t ea, hipEvent_t eb, auto&& launch, float* ms_out) -> td_status {
        td_status st2 = launch();
        if (st2 < 0) return st2;
        static const bool evict = !(getenv("TD_TUNE_EVICT") && atoi(getenv("TD_TUNE_EVICT")) == 0);
        if (evict) {
            // one launch per timed interval, L2 emptied before it; the fastest of five
            if (!e->tune_evict) {
                e->tune_evict_bytes = (size_t)64 << 20;
                if (hipMalloc(&e->tune_evict, e->tune_evict_bytes) != hipSuccess) {      // no room for the scratch: time the hot loop instead
                    (void)hipGetLastError();
                    e->tune_evict = nullptr;
                    e->tune_evict_bytes = 0;
                }
            }
        }
        if (evict && e->tune_evict) {
            float ms = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                TD_HIP_CHECK(hipMemsetAsync(e->tune_evict, rep, e->tune_evict_bytes, s_));
                TD_HIP_CHECK(hipEventRecord(ea, s_));
                if ((st2 = launch()) < 0) return st2;
                TD_HIP_CHECK(hipEventRecord(eb, s_));
                TD_HIP_CHECK(hipEventSynchronize(eb));
                float t = 0.f;
                TD_HIP_CHECK(hipEventElapsedTime(&t, ea, eb));
                if (t < ms) ms = t;
            }
            *ms_out = ms;
            return TD_OK;
        }
        float ms = 1e30f;
        for (int round = 0, reps = 2; round < 2; ++round, reps = 8) {
            TD_HIP_CHECK(hipEventRecord(ea, s_));
            for (int rep = 0; rep < reps; ++rep)
                if ((st2 = launch()) < 0) return st2;
            TD_HIP_CHECK(hipEventRecord(eb, s_));
            TD_HIP_CHECK(hipEventSynchronize(eb));
            float t = 0.f;
            TD_HIP_CHECK(hipEventElapsedTime(&t, ea, eb));
            t /= (float)reps;
            if (t < ms) ms = t;
            if (ms > 0.1f) break;
        }
        *ms_out = ms;
        return TD_OK;
    };
    // measured block tile of one launch shape (cached per engine, shared through the TD_TUNE_CACHE file)
    // bd_ok: the launch being tuned has a fragment-ordered filter copy and a plain output (filter-direct tiles 23-27); the
    // Winograd plane contractions pass false (their ConvArgs carry no w_frag: conv2d_launch would remap the id to the heuristic tile)
    auto tuned_cfg = [&](const std::tuple<int, int, int, int, int>& key, int prec_, int ksteps, bool pp8_ok, bool plane_ok, bool bd_ok, hipStream_t s_,
                         auto&& launch_cfg, int* cfg_out, float* best_ms) -> td_status {
        static const int forced = getenv("TD_FORCE_CFG") ? atoi(getenv("TD_FORCE_CFG")) : -1;      // diagnostics: one block tile everywhere it applies
        if (forced >= 0 && forced <= TD_CONV_TILE_CFG_MAX && !best_ms && !(forced >= 14 && forced <= 16 && ksteps > 4) && !(forced == 17 && !pp8_ok) &&
            !(forced >= 18 && forced <= 20 && !plane_ok) && !(conv_cfg_is_bd(forced) && !bd_ok)) {
            *cfg_out = forced;
            return TD_OK;
        }
        auto it = e->tuned.find(key);
        if (it == e->tuned.end()) {
            load_tune_cache(e);                   // another engine of this process may have measured it meanwhile
            it = e->tuned.find(key);
        }
        if (it != e->tuned.end() && !best_ms) {
            *cfg_out = it->second;
            return TD_OK;
        }
        float best = 1e30f;
        int best_cfg = -1;
        hipEvent_t ea, eb;
        TD_HIP_CHECK(hipEventCreate(&ea));
        TD_HIP_CHECK(hipEventCreate(&eb));
        for (int c : TD_CONV_TUNE_CANDIDATES) {
            if (it != e->tuned.end() && c != it->second) continue;       // known choice: only its time is wanted
            if (c >= 14 && c <= 16 && ksteps > 4) continue;      // single-stage tiles only pay on the thin 1x1 layers
            if (c == 17 && !pp8_ok) continue;                    // fp16 256x256 ping-pong tile
            if (c >= 18 && c <= 20 && !plane_ok) continue;       // persistent tile walk: Winograd plane contractions only
            if (conv_cfg_is_bd(c) && !bd_ok) continue;          // filter-direct tiles: layers with a fragment-ordered filter copy
            if (c == 33 && !(std::get<2>(key) == 17 && (std::get<4>(key) & ~1) == 4 && std::get<0>(key) >= 128 && ksteps <= 4 && ksteps != 3)) continue;      // filter-stationary: 1x1, stride 1, plain output, <= 4 k-chunks
            if ((c == 31 || c == 32) && std::get<0>(key) > 32) continue;      // 32-column tiles: the thin heads only
            float ms = 1e30f;
            td_status st2 = time_launch(s_, ea, eb, [&]() { return launch_cfg(c); }, &ms);
            if (st2 < 0) {
                (void)hipEventDestroy(ea);
                (void)hipEventDestroy(eb);
                return st2;
            }
            static const float hyst = 1.f - 0.01f * (getenv("TD_TUNE_HYST") ? (float)atof(getenv("TD_TUNE_HYST")) : 2.f);
            if (best_cfg < 0 || ms < best * hyst) { best = ms; best_cfg = c; }
        }
        (void)hipEventDestroy(ea);
        (void)hipEventDestroy(eb);
        if (it == e->tuned.end()) {
            e->tuned.emplace(key, best_cfg);
            if (!e->tune_cache.empty()) {
                if (FILE* f = fopen(e->tune_cache.c_str(), "a")) {
                    fprintf(f, "%d %d %d %d %d %d %d\n", prec_, std::get<0>(key), std::get<1>(key), std::get<2>(key), std::get<3>(key), std::get<4>(key), best_cfg);
                    fclose(f);
                }
            }
        }
        *cfg_out = best_cfg;
        if (best_ms) *best_ms = best;
        return TD_OK;
    };
    // Winograd F(2x2,3x3) path of a stride-1 3x3 layer (fp32 engine): input transform → ONE batched launch of the 16
    // plane contractions through conv_igemm_kernel → output transform with the layer's scale / bias / ReLU (winograd.hip)
    auto run_wino = [&](const ConvLayer& L, const void* x_, int B_, int H_, int W_, bool relu, void* y_, hipStream_t s_,
                        const int* m_dyn, int m_mul, int gemm_cfg) -> td_status {
        const long long T = (long long)B_ * ((H_ + 1) / 2) * ((W_ + 1) / 2);
        const int Ts = wino_slab_tiles(e, L);                 // tiles per slab: V + M of a slab stay in the Infinity Cache
        for (long long t0 = 0; t0 < T; t0 += Ts) {
            const int n = (int)std::min<long long>(Ts, T - t0);
            td_status st2;
            ConvArgs a{};
            a.w = L.wino_u; a.y = e->wino_m;
            a.Cin = L.cin; a.Cout = L.cout; a.KH = a.KW = 1; a.stride = 1; a.pad = 0;
            a.M = n; a.m_dyn = m_dyn; a.m_mul = m_dyn ? m_mul / 4 : 1;    // even H, W with a device row count: tiles = rows / 4
            a.m_off = (int)t0;
            a.w_bs = (long long)L.cout * L.cin; a.y_bs = (long long)n * L.cout;
            // speed-of-light model (static row counts only): 16 plane products of [n x cin] x [cin x cout]; V = 4x the input,
            // M = 4x the output, each crossing HBM once per direction
            const double xb = 4.0 * B_ * H_ * W_ * L.cin, yb = 4.0 * B_ * H_ * W_ * L.cout, ub = 4.0 * 16 * L.cout * L.cin;
            const double vb = 4.0 * 16 * n * L.cin, mb = 4.0 * 16 * n * L.cout, pf = 2.0 * 16 * n * (double)L.cin * L.cout;
            const bool dyn = m_dyn != nullptr;
            if (e->wino_fused) {
                // input transform fused into the contraction's A staging (wino_gemm_kernel): V never exists in memory
                ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_GEMM, dyn ? 0.0 : pf, dyn ? 0.0 : xb + ub + mb);
                a.x = x_; a.B = B_; a.H = H_; a.W = W_;
                if ((st2 = wino_gemm_launch(a, s_)) < 0) return st2;
            } else {
                { ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_XFORM, 0.0, dyn ? 0.0 : xb + vb);
                if ((st2 = wino_input_launch(static_cast<const float*>(x_), B_, H_, W_, L.cin, e->wino_v, m_dyn, m_mul, t0, n, s_)) < 0) return st2; }
                a.x = e->wino_v; a.B = 1; a.H = 1; a.W = n; a.Ho = 1; a.Wo = n;
                a.batch_count = 16; a.x_bs = (long long)n * L.cin;
                a.tile_cfg = gemm_cfg;
                ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_GEMM, dyn ? 0.0 : pf, dyn ? 0.0 : vb + ub + mb);
                if ((st2 = conv2d_launch(a, TD_PRECISION_FP32, s_)) < 0) return st2;
            }
            ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_XFORM, 0.0, dyn ? 0.0 : mb + yb);
            if ((st2 = wino_output_launch(e->wino_m, B_, H_, W_, L.cout, L.scale, L.bias, relu ? 1 : 0, static_cast<float*>(y_), m_dyn,
                                          m_mul, t0, n, s_)) < 0) return st2;
        }
        return TD_OK;
    };
    // Winograd F(4x4,3x3) of a whole layer: x → V [36][T][cin] → 36 batched plane contractions → M → y. m_dyn = device-side
    // image count (the mask head's live RoIs): the planes keep the stride of the full batch, only the live tiles are computed.
    auto run_wino43 = [&](const ConvLayer& L, const void* x_, int B_, int H_, int W_, bool relu, void* y_, hipStream_t s_,
                          const int* m_dyn, int gemm_cfg, const ConvLayer* head = nullptr, float* head_y = nullptr, bool fold = false) -> td_status {
        const int tiles_img = ((H_ + 3) / 4) * ((W_ + 3) / 4);
        const long long T = (long long)B_ * tiles_img;
        td_status st2;
        // speed-of-light model: 36 plane products of [T x cin] x [cin x cout]; V / M = 36 T rows (2.25x the padded input / output)
        const bool dyn = m_dyn != nullptr;
        const double xb = 4.0 * B_ * H_ * W_ * L.cin, yb = 4.0 * B_ * H_ * W_ * L.cout, ub = 4.0 * 36 * L.cout * L.cin;
        const double vb = 4.0 * 36 * T * L.cin, mb = 4.0 * 36 * T * L.cout, pf = 2.0 * 36 * T * (double)L.cin * L.cout;
        if (fold) {
            // contraction + output transform in one launch: V and U in, y out (wino_fused.hip). Its 32-bit plane offsets hold
            // 4 GB of V: a larger batch runs in image groups, so the layer takes the SAME form at every batch size (batch invariance)
            const unsigned long long per_img = 36ull * tiles_img * (unsigned long long)std::max(L.cin, L.cout) * 4ull;
            const int gmax = (int)std::max<unsigned long long>(1ull, std::min<unsigned long long>((0xfffffff0ull - (2u << 20)) / per_img,
                                                                                                 (0xfffffff0ull - (2u << 20)) / (16ull * tiles_img * L.cout * 4ull)));
            for (int b0 = 0; b0 < B_; b0 += gmax) {
                const int Bg = dyn ? B_ : std::min(gmax, B_ - b0);       // (device-side RoI count: one group — 800 RoIs of 14 x 14 are 0.5 GB)
                const float* xg = static_cast<const float*>(x_) + (size_t)b0 * H_ * W_ * L.cin;
                float* yg = static_cast<float*>(y_) + (size_t)b0 * H_ * W_ * L.cout;
                const double share = (double)Bg / B_;
                { ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_XFORM, 0.0, dyn ? 0.0 : (xb + vb) * share);
                if ((st2 = wino43_input_launch(xg, Bg, H_, W_, L.cin, e->wino_v, m_dyn, s_)) < 0) return st2; }
                ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_GEMM, dyn ? 0.0 : pf * share, dyn ? 0.0 : (vb + yb) * share + ub);
                if ((st2 = wino43_fused_launch(e->wino_v, static_cast<const float*>(L.wino_u43), Bg, H_, W_, L.cin, L.cout, L.scale, L.bias, relu ? 1 : 0,
                                               yg, m_dyn, s_)) < 0) return st2;
                if (dyn) break;
            }
            return TD_OK;
        }
        { ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_XFORM, 0.0, dyn ? 0.0 : xb + vb);
        if ((st2 = wino43_input_launch(static_cast<const float*>(x_), B_, H_, W_, L.cin, e->wino_v, m_dyn, s_)) < 0) return st2; }
        ConvArgs a{};
        a.x = e->wino_v; a.w = L.wino_u43; a.y = e->wino_m;
        a.Cin = L.cin; a.Cout = L.cout; a.KH = a.KW = 1; a.stride = 1; a.pad = 0;
        a.B = 1; a.H = 1; a.W = (int)T; a.Ho = 1; a.Wo = (int)T;
        a.M = (int)T; a.m_dyn = m_dyn; a.m_mul = m_dyn ? tiles_img : 1;
        a.batch_count = 36; a.x_bs = T * L.cin; a.w_bs = (long long)L.cout * L.cin; a.y_bs = T * L.cout;
        a.tile_cfg = gemm_cfg;
        { ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_GEMM, dyn ? 0.0 : pf, dyn ? 0.0 : vb + ub + mb);
        if ((st2 = conv2d_launch(a, TD_PRECISION_FP32, s_)) < 0) return st2; }
        if (head) {        // the 1x1 head contracted inside the output transform (wino43_output_head_kernel): y_ is not written
            const double hflops = 2.0 * B_ * H_ * W_ * (double)head->cout * L.cout, hb = 4.0 * B_ * H_ * W_ * head->cout + 4.0 * head->cout * L.cout;
            if (e->prof) {
                e->prof_flops[0] += hflops;
                e->prof_flops[8] += hflops;
                e->prof_launches[0] += 1;          // two layers of the reference in one launch: "launches" keeps counting layers
                e->prof_bytes[0] += hb + yb;       // the unfused pair's algorithmic bytes
            }
            ClassScope cs(e, s_, TD_CLS_WINO_XFORM, hflops, mb + hb);
            return wino43_output_head_launch(e->wino_m, B_, H_, W_, L.cout, L.scale, L.bias, relu ? 1 : 0, static_cast<const float*>(head->w),
                                             head->bias, head_y, head->cout, s_);
        }
        ClassScope cs(e, s_, dyn ? TD_CLS_MASK_HEAD : TD_CLS_WINO_XFORM, 0.0, dyn ? 0.0 : mb + yb);
        return wino43_output_launch(e->wino_m, B_, H_, W_, L.cout, L.scale, L.bias, relu ? 1 : 0, static_cast<float*>(y_), m_dyn, s_);
    };
    auto run_conv = [&](const ConvLayer& L, const void* x_, int B_, int H_, int W_, int stride, int pad, bool relu,
                        void* y_, const void* res_, int res_shift, hipStream_t s_, int prec_,
                        const int* m_dyn = nullptr, int m_mul = 1, int out_mode = 0,
                        const ConvLayer* head = nullptr, float* head_y = nullptr, bool* head_fused = nullptr) -> td_status {
        const int Ho = (H_ + 2 * pad - L.kh) / stride + 1, Wo = (W_ + 2 * pad - L.kw) / stride + 1;
        const double M = (double)B_ * Ho * Wo, K = (double)L.kh * L.kw * L.cin;
        const double flops = 2.0 * M * L.cout * K;
        const double es = prec_ == TD_PRECISION_FP16 ? 2.0 : 4.0;
        const double bytes = es * ((double)B_ * H_ * W_ * L.cin / (stride * stride) + M * L.cout * (res_ ? 2.0 : 1.0) + L.cout * K);
        int cfg = -1, wino_cfg = -1;
        bool use_wino = false, use_43 = false, use_fold = false;
        td_status st2;
        if (e->autotune) {
            const auto key = std::make_tuple(L.cout, L.cin, L.kh * 16 + L.kw, B_ * Ho * Wo, stride * 4 + out_mode * 2 + (res_ ? 1 : 0));
            const int ksteps = L.kh * L.kw * L.cin / (prec_ == TD_PRECISION_FP16 ? 64 : 32);
            const bool pp8_ok = prec_ == TD_PRECISION_FP16 && out_mode == 0 && L.cout >= 128;
            const bool bd_ok = out_mode == 0 && L.w_frag != nullptr;
            // persistent tile walk (tile ids 18-20): fp32 1x1 / stride-1 layers, same-size residual at most
            const bool plane_ok = prec_ == TD_PRECISION_FP32 && L.kh == 1 && L.kw == 1 && stride == 1 && pad == 0 && out_mode == 0 && res_shift == 0 &&
                                  !L.out_f32 && L.cin >= 32 && L.cin % 32 == 0;
            auto direct = [&](int c) { return run_conv_raw(L, x_, B_, H_, W_, stride, pad, relu, y_, res_, res_shift, s_, prec_, m_dyn, m_mul, out_mode, c); };
            const bool wino_ok = prec_ == TD_PRECISION_FP32 && L.wino_u && e->wino_v && stride == 1 && pad == 1 && !res_ && out_mode == 0 &&
                                 (!m_dyn || ((H_ | W_) & 1) == 0) &&
                                 (size_t)16 * std::min<long long>((long long)B_ * ((H_ + 1) / 2) * ((W_ + 1) / 2), wino_slab_tiles(e, L)) *
                                         (size_t)std::max(L.cin, L.cout) <= e->wino_elems;
            if (wino_ok) {
                // the batched plane contraction is a launch shape of its own (1x1, rows = tiles, flag 32 = batched x16)
                const int T = (int)std::min<long long>((long long)B_ * ((H_ + 1) / 2) * ((W_ + 1) / 2), wino_slab_tiles(e, L));
                const auto wkey = std::make_tuple(L.cout, L.cin, 1 * 16 + 1, T, 4 + 32);
                auto wino = [&](int c) { return run_wino(L, x_, B_, H_, W_, relu, y_, s_, m_dyn, m_mul, c); };
                // Which path a layer takes must not depend on a measurement: the two differ in rounding, and a timing
                // decision would make the results vary between processes (sharded ranks, resumed runs). The rule is the
                // measured outcome on the R50/R101 layer set (profiles/r02_tile_choices.txt): every 3x3 layer with at least
                // 128 channels on both sides is faster through Winograd at every map size from 13x13 to 200x200
                // (1.3-1.9x); 64 -> 64 (res2) is HBM-bound on the transforms and stays direct.
                use_wino = L.cin >= e->wino_minc && L.cout >= e->wino_minc;
                // F(4x4,3x3) on the large maps (another fixed rule: map size and channels only): 2.25 multiplies per output
                // instead of 4; the 36 plane contractions are one batched launch whose block tile is measured (flag 64)
                const long long T43 = (long long)B_ * ((H_ + 3) / 4) * ((W_ + 3) / 4);
                use_43 = use_wino && L.wino_u43 && e->wino43_min > 0 && H_ >= e->wino43_min && W_ >= e->wino43_min &&
                         (size_t)36 * T43 * (size_t)std::max(L.cin, L.cout) <= e->wino_elems && T43 * std::max(L.cin, L.cout) < (1ll << 31);
                // the fold rule: layer shape only (see td_engine::wino_fold)
                const int hw_ = H_ * W_, wf_ = e->wino_fold;
                // (a layer with a 1x1 head — the RPN conv — folds only on the 160 x 160+ maps, bit 32: its head then runs as a launch
                // of its own on the stored map, 1 000 + ~70 us against 1 280 us for the three-launch form with the head in its
                // output transform; on the smaller maps that form wins)
                // (device-side RoI count: ALL B_ reserved RoIs are one group below, so its 32-bit plane offsets must hold them — about
                // 7 279 RoIs of 14 x 14 x 256; past that the layer keeps the three-launch form. Still a rule on reserved shapes only.)
                use_fold = use_43 && wf_ > 0 && wino43_fused_ok(m_dyn ? B_ : 1, H_, W_, L.cin, L.cout) &&
                           (((wf_ & 1) && !m_dyn && (!head || (wf_ & 32)) && ((L.cin == 256 && hw_ >= 160 * 160) || (L.cin == 128 && hw_ >= 80 * 80))) ||
                            ((wf_ & 2) && m_dyn && !head) || ((wf_ & 4) && !m_dyn && !head && L.cin == 256 && hw_ >= 80 * 80) ||
                            ((wf_ & 8) && !m_dyn && !head && hw_ >= 40 * 40) || ((wf_ & 16) && !m_dyn && !head));
                if (use_43 && !use_fold) {
                    const auto key43 = std::make_tuple(L.cout, L.cin, 1 * 16 + 1, (int)T43, 4 + 64);
                    auto w43 = [&](int c) { return run_wino43(L, x_, B_, H_, W_, relu, y_, s_, m_dyn, c); };
                    if ((st2 = tuned_cfg(key43, prec_, L.cin / 32, false, true, false, s_, w43, &wino_cfg, nullptr)) < 0) return st2;
                }
                if (use_wino && !use_43 && !e->wino_fused && (st2 = tuned_cfg(wkey, prec_, L.cin / 32, false, true, false, s_, wino, &wino_cfg, nullptr)) < 0) return st2;
            }
            if (!use_wino && (st2 = tuned_cfg(key, prec_, ksteps, pp8_ok, plane_ok, bd_ok, s_, direct, &cfg, nullptr)) < 0) return st2;
        }
        ProfScope ps(e, s_, m_dyn ? 7 : 0, m_dyn ? 0.0 : flops, m_dyn ? 0.0 : bytes);
        if (e->prof && !m_dyn) {          // category 8: FLOPs the MFMA pipe really executes
            const double padded = use_43 ? (double)(((H_ + 3) / 4) * 4) * (((W_ + 3) / 4) * 4) / ((double)H_ * W_) : 1.0;
            e->prof_flops[8] += use_43 ? flops * padded * 0.25 : (use_wino ? flops * 4.0 / 9.0 : flops);
            e->prof_launches[8] += use_wino ? 1 : 0;
        }
        if (use_43) {
            // fp32 engine: the head rides in the F(4x4) output transform — a FIXED rule (layer shapes only), bit-identical to the
            // separate launch anyway
            const bool fuse43 = head && !use_fold && e->fuse_head && !m_dyn && relu && L.cout == 256 && head->cin == 256 && head->kh == 1 && head->kw == 1 &&
                                head->cout <= 32 && !head->scale;
            if (head_fused) *head_fused = fuse43;
            return run_wino43(L, x_, B_, H_, W_, relu, y_, s_, m_dyn, wino_cfg, fuse43 ? head : nullptr, fuse43 ? head_y : nullptr, use_fold);
        }
        if (use_wino) return run_wino(L, x_, B_, H_, W_, relu, y_, s_, m_dyn, m_mul, wino_cfg);
        const int cls = m_dyn ? TD_CLS_MASK_HEAD : (H_ == 1 && W_ == 1 ? TD_CLS_FC : (L.kh == 1 && L.kw == 1 ? TD_CLS_CONV1X1 : TD_CLS_CONV3X3));
        // A 1x1 head contracted from the finished tile inside the same launch (the RPN's 15-row head after its 3x3 conv): only
        // when the measured tile owns all 256 output channels — bit-identical to the separate launch either way, so the tile
        // choice (a timing) may decide it. The head's FLOPs / bytes join this launch's; its own input read and this layer's
        // output write disappear.
        const bool fuse_head = head && e->fuse_head && !m_dyn && out_mode == 0 && !res_ && relu && L.cout == 256 && head->cin == 256 &&
                               head->kh == 1 && head->kw == 1 && head->cout <= 32 && head->out_f32 && !head->scale && conv_head_capable(cfg, prec_);
        if (head_fused) *head_fused = fuse_head;
        if (fuse_head) {
            const double hflops = 2.0 * M * head->cout * L.cout, hbytes = 4.0 * M * head->cout + es * head->cout * L.cout;
            const double bytes_f = bytes - es * M * L.cout + hbytes;
            if (e->prof) {
                e->prof_flops[0] += hflops;
                e->prof_flops[8] += hflops;
                e->prof_launches[0] += 1;          // two layers of the reference in one launch: "launches" keeps counting layers
                e->prof_bytes[0] += hbytes + es * M * L.cout;       // the unfused pair's algorithmic bytes (every tensor of every layer once)
            }
            ClassScope cs(e, s_, cls, flops + hflops, bytes_f);
            return run_conv_raw(L, x_, B_, H_, W_, stride, pad, relu, y_, res_, res_shift, s_, prec_, m_dyn, m_mul, out_mode, cfg, head, head_y);
        }
        ClassScope cs(e, s_, cls, m_dyn ? 0.0 : flops, m_dyn ? 0.0 : bytes);
        return run_conv_raw(L, x_, B_, H_, W_, stride, pad, relu, y_, res_, res_shift, s_, prec_, m_dyn, m_mul, out_mode, cfg);
    };
    // The same 3x3 layer shape on several pyramid levels as ONE launch (conv_pp8_grouped_launch, fp16 engine): the FPN's output
    // convs, the RPN conv with its head. The small levels (p4-p6: 79 / 20 / 6 tiles of 256 rows) cannot fill 256 CUs on their own;
    // in one grid their tiles run beside the big levels' last wave. A fixed rule (layer shapes only); bit-identical per level.
    auto groupable = [&](const ConvLayer& L) {
        return L.kh == 3 && L.kw == 3 && L.cout == 256 && L.cin % 64 == 0 && !L.scale && L.bias && !L.out_f32;
    };
    auto run_grouped = [&](const ConvLayer* const* Ls, const void* const* xs, void* const* ys, float* const* head_ys, int nl, const int* Hs,
                           const int* Ws, bool relu, const ConvLayer* head, hipStream_t s_) -> td_status {
        ConvArgs a{};
        a.B = B; a.Cin = Ls[0]->cin; a.Cout = 256; a.KH = a.KW = 3; a.stride = 1; a.pad = 1; a.relu = relu ? 1 : 0; a.m_mul = 1; a.nlev = nl;
        a.tile_cfg = 17;
        double flops = 0.0, bytes = 0.0, unfused_bytes = 0.0;
        for (int l = 0; l < nl; ++l) {
            a.lev[l].x = xs[l]; a.lev[l].w = Ls[l]->w; a.lev[l].bias = Ls[l]->bias;
            a.lev[l].y = ys ? ys[l] : nullptr; a.lev[l].head_y = head_ys ? head_ys[l] : nullptr;
            a.lev[l].H = Hs[l]; a.lev[l].W = Ws[l];
            const double M = (double)B * Hs[l] * Ws[l], K = 9.0 * a.Cin;
            flops += 2.0 * M * 256 * K + (head ? 2.0 * M * head->cout * 256 : 0.0);
            bytes += 2.0 * (M * a.Cin + (head ? 0.0 : M * 256) + 256 * K) + (head ? 4.0 * M * head->cout + 2.0 * head->cout * 256 : 0.0);
            unfused_bytes += head ? 2.0 * 2.0 * M * 256 : 0.0;          // the 256-channel map written and read back by a separate head
        }
        if (head) { a.head_w = head->w; a.head_b = head->bias; a.head_n = head->cout; }
        ProfScope ps(e, s_, 0, flops, bytes + unfused_bytes);
        if (e->prof) {
            e->prof_flops[8] += flops;
            e->prof_launches[0] += nl * (head ? 2 : 1) - 1;      // "launches" keeps counting the reference's layers
        }
        ClassScope cs(e, s_, TD_CLS_CONV3X3, flops, bytes);
        return conv_pp8_grouped_launch(a, s_);
    };
    // conv2 (3x3) → conv3 (1x1) + shortcut + ReLU of a bottleneck block as one launch (bottleneck_tail_kernel)
    auto run_tail = [&](const Block& blk, const void* t1_, int B_, int H_, int W_, void* y_, const void* shortcut_, hipStream_t s_) -> td_status {
        TailArgs a{};
        a.x = t1_; a.w2 = blk.c2.w; a.scale2 = blk.c2.scale; a.bias2 = blk.c2.bias;
        a.w3 = blk.c3.w; a.scale3 = blk.c3.scale; a.bias3 = blk.c3.bias; a.res = shortcut_; a.y = y_;
        a.B = B_; a.H = H_; a.W = W_; a.MID = blk.c2.cout; a.COUT = blk.c3.cout; a.M = B_ * H_ * W_;
        const double M = (double)a.M, mid = a.MID, co = a.COUT, es = prec == TD_PRECISION_FP16 ? 2.0 : 4.0;
        const double flops = 2.0 * M * mid * (9.0 * mid) + 2.0 * M * co * mid;
        const double bytes = es * (M * mid + 2.0 * M * co + 9.0 * mid * mid + co * mid);       // t1 in, shortcut in, y out, both filter banks
        ProfScope ps(e, s_, 0, flops, bytes);
        if (e->prof) {
            e->prof_flops[8] += flops;
            e->prof_launches[0] += 1;          // two layers of the reference in one launch: keep "launches" = layers
            // the unfused pair's algorithmic bytes (what `bytes` of category 0 means: every tensor of every LAYER once)
            e->prof_bytes[0] += es * 2.0 * M * mid;
        }
        ClassScope cs(e, s_, TD_CLS_TAIL, flops, bytes);
        return bottleneck_tail_launch(a, prec, s_);
    };
    // ---- backbone ------------------------------------------------------------------------------------------
    // Runs in sub-batches of `sb` images (stem → res5 per sub-batch): the stage-2/3 activations of a sub-batch
    // (≈ 80 MB per 256-channel tensor and image at fp32) then hand over from layer to layer through the 256 MiB
    // Infinity Cache instead of HBM. sb comes from TD_BACKBONE_SUBBATCH (default: the whole batch at once).
    int hs[5], wsz[5];
    for (int l = 0; l < 4; ++l) {
        hs[l] = Hp >> (l + 2);
        wsz[l] = Wp >> (l + 2);
    }
    hs[4] = (hs[3] - 1) / 2 + 1;
    wsz[4] = (wsz[3] - 1) / 2 + 1;
    const size_t esz = prec == TD_PRECISION_FP16 ? 2 : 4;
    auto off = [&](void* p, size_t elems) -> void* { return static_cast<char*>(p) + elems * esz; };
    int sb = e->backbone_subbatch > 0 ? e->backbone_subbatch : B;
    if (sb > B) sb = B;
    const bool do_stem = PH(6) || (PH(0) && !e->stem_done);
    // ONE event pair around the whole trunk (backbone, FPN, RPN convs): every hipEventRecord is a kernel boundary with a
    // signal on the main stream, and the fp16 trunk is short enough to notice each of them (un-profiled 4.87 ms per step,
    // 5.11 ms with a pair per section). The p6 subsample (5 us) falls inside the bracket and is timed with the convs.
    std::optional<ProfScope> trunk_group;
    for (int b0 = 0; (PH(0) || PH(6)) && b0 < B; b0 += sb) {
        const int nb_img = (B - b0) < sb ? (B - b0) : sb;
        ImgSizes vsub{};
        for (int i = 0; i < nb_img; ++i) {
            vsub.h[i] = valid.h[b0 + i];
            vsub.w[i] = valid.w[b0 + i];
        }
        const size_t in_img = input_format == TD_INPUT_U8_HWC ? (size_t)Hp * Wp * 3 : (size_t)Hp * Wp * 3 * 4;
        const void* img_sub = static_cast<const char*>(images) + (size_t)b0 * in_img;
        void* stem_sub = off(e->stem_out, (size_t)b0 * (Hp / 2) * (Wp / 2) * e->stem_c);
        void* pool_sub = off(e->pool_out, (size_t)b0 * hs[0] * wsz[0] * e->stem_c);
        if (do_stem) {
            { ProfScope ps(e, s, 1);
            if ((st = stem_launch(img_sub, input_format, vsub, nb_img, Hp, Wp, e->stem_w, e->stem_scale, e->stem_bias, stem_sub,
                                  e->stem_c, prec, s, e->stem_w16, e->stem_bias16)) < 0) return st; }
            { ProfScope ps(e, s, 2);
            if ((st = maxpool3x3s2_launch(stem_sub, pool_sub, nb_img, Hp / 2, Wp / 2, e->stem_c, prec, s)) < 0) return st; }
        }
        if (!PH(0)) continue;
        if (!trunk_group && sb >= B) trunk_group.emplace(e, s, 0, 0.0, 0.0, true);
        const void* x = pool_sub;
        int xh = hs[0], xw = wsz[0];
        ProfScope backbone_group(e, s, 0, 0.0, 0.0, true);
        for (int si = 0; si < 4; ++si) {
            const int nb = (int)e->stages[si].size();
            const int oh = hs[si], ow = wsz[si];
            const size_t px = (size_t)b0 * oh * ow;
            const int mid = e->stages[si][0].c1.cout, co = e->stages[si][0].c3.cout;
            void* t1 = off(e->t1[si], px * mid);
            void* t2 = off(e->t2[si], px * mid);
            void* scb = off(e->scb[si], px * co);
            void* resb = off(e->res[si], px * co);
            void* xtmp = off(e->xtmp[si], px * co);
            for (int bi = 0; bi < nb; ++bi) {
                const Block& blk = e->stages[si][bi];
                // the last block must land in res[si]: alternate so that block nb-1 writes res
                void* y = ((nb - 1 - bi) % 2 == 0) ? resb : xtmp;
                const void* shortcut = x;
                if (blk.has_sc) {
                    if ((st = run_conv(blk.sc, x, nb_img, xh, xw, blk.stride, 0, false, scb, nullptr, 0, s, prec)) < 0) return st;
                    shortcut = scb;
                }
                if ((st = run_conv(blk.c1, x, nb_img, xh, xw, blk.stride, 0, true, t1, nullptr, 0, s, prec)) < 0) return st;
                // Measured (round 3, plain loop, batch 8, 64-row blocks): fp32 res2 345 us fused against 215 + 185 us as two launches;
                // fp16 res2 130 us against 54 + 87; fp16 res3 111 us against 41 + 47 — slower (a fused block is a chain of short
                // latency-bound steps, and with 128 mid channels only two blocks fit a CU), so res3 fuses only on request
                // (TD_FUSE_TAIL=2). Either way the result is bit-identical, so this IS allowed to follow a measurement.
                if (e->fuse_tail && (blk.c2.cout == 64 || e->fuse_tail_fp16) && blk.c2.kh == 3 && blk.c2.kw == 3 &&
                    blk.c2.cin == blk.c2.cout && blk.c3.cin == blk.c2.cout && bottleneck_tail_ok(prec, blk.c2.cout, blk.c3.cout)) {
                    // conv2 + conv3 + shortcut add in one launch, bit-identical to the two launches below (bottleneck.hip); the mid
                    // tensor t2 never reaches HBM
                    if ((st = run_tail(blk, t1, nb_img, oh, ow, y, shortcut, s)) < 0) return st;
                } else {
                    if ((st = run_conv(blk.c2, t1, nb_img, oh, ow, 1, 1, true, t2, nullptr, 0, s, prec)) < 0) return st;
                    if ((st = run_conv(blk.c3, t2, nb_img, oh, ow, 1, 0, true, y, shortcut, 0, s, prec)) < 0) return st;
                }
                x = y;
                xh = oh;
                xw = ow;
            }
        }
    }
    if (PH(0)) {
    set_named(e, "stem", e->stem_out, B, Hp / 2, Wp / 2, e->stem_c, (int)esz);
    set_named(e, "pool", e->pool_out, B, Hp / 4, Wp / 4, e->stem_c, (int)esz);
    for (int si = 0; si < 4; ++si) {
        const std::string nm = "res" + std::to_string(si + 2);
        set_named(e, nm.c_str(), e->res[si], B, hs[si], wsz[si], e->stages[si][0].c3.cout, (int)esz);
    }
    // ---- FPN (top-down; the nearest-2x upsampled add rides in the lateral conv's epilogue) ---------------------
    {
    ProfScope fpn_group(e, s, 0, 0.0, 0.0, true);
    // fp16: the four output convs do not depend on each other (only the lateral sums chain top-down), so the laterals run first
    // and the outputs follow as one grouped launch
    const bool group_fpn = prec == TD_PRECISION_FP16 && e->group_levels && groupable(e->fpn_out[0]) && groupable(e->fpn_out[1]) &&
                           groupable(e->fpn_out[2]) && groupable(e->fpn_out[3]);
    for (int l = 3; l >= 0; --l) {
        const void* td_res = l == 3 ? nullptr : e->inner[l + 1];
        if ((st = run_conv(e->lateral[l], e->res[l], B, hs[l], wsz[l], 1, 0, false, e->inner[l], td_res, td_res ? 1 : 0, s, prec)) < 0) return st;
        if (!group_fpn && (st = run_conv(e->fpn_out[l], e->inner[l], B, hs[l], wsz[l], 1, 1, false, e->pfeat[l], nullptr, 0, s, prec)) < 0) return st;
    }
    if (group_fpn) {
        const ConvLayer* Ls[4] = {&e->fpn_out[0], &e->fpn_out[1], &e->fpn_out[2], &e->fpn_out[3]};
        const void* xs[4] = {e->inner[0], e->inner[1], e->inner[2], e->inner[3]};
        void* ys[4] = {e->pfeat[0], e->pfeat[1], e->pfeat[2], e->pfeat[3]};
        if ((st = run_grouped(Ls, xs, ys, nullptr, 4, hs, wsz, false, nullptr, s)) < 0) return st;
    }
    }
    { ProfScope ps(e, s, 2);
    if ((st = subsample2_launch(e->pfeat[3], e->pfeat[4], B, hs[3], wsz[3], e->fpn_c, prec, s)) < 0) return st; }
    for (int l = 0; l < 5; ++l) {
        const std::string nm = "p" + std::to_string(l + 2);
        set_named(e, nm.c_str(), e->pfeat[l], B, hs[l], wsz[l], e->fpn_c, (e->desc.precision == TD_PRECISION_FP16 ? 2 : 4));
    }
    // ---- RPN -----------------------------------------------------------------------------------------------------
    {
        ProfScope rpn_group(e, s, 0, 0.0, 0.0, true);
        const bool group_rpn = prec == TD_PRECISION_FP16 && e->group_levels && e->fuse_head && groupable(e->rpn_conv) && e->rpn_head.cin == 256 &&
                               e->rpn_head.kh == 1 && e->rpn_head.kw == 1 && e->rpn_head.cout <= 32 && e->rpn_head.out_f32 && !e->rpn_head.scale;
        if (group_rpn) {
            const ConvLayer* Ls[5] = {&e->rpn_conv, &e->rpn_conv, &e->rpn_conv, &e->rpn_conv, &e->rpn_conv};
            const void* xs[5] = {e->pfeat[0], e->pfeat[1], e->pfeat[2], e->pfeat[3], e->pfeat[4]};
            float* hy[5] = {e->rpn_headbuf[0], e->rpn_headbuf[1], e->rpn_headbuf[2], e->rpn_headbuf[3], e->rpn_headbuf[4]};
            if ((st = run_grouped(Ls, xs, nullptr, hy, 5, hs, wsz, true, &e->rpn_head, s)) < 0) return st;
        }
        for (int l = 0; l < 5; ++l) {
            if (group_rpn) {
                const std::string nm = "rpn_head" + std::to_string(l + 2);
                set_named(e, nm.c_str(), e->rpn_headbuf[l], B, hs[l], wsz[l], RPN_HEAD_C);
                continue;
            }
            bool fused = false;
            if ((st = run_conv(e->rpn_conv, e->pfeat[l], B, hs[l], wsz[l], 1, 1, true, e->rpn_t, nullptr, 0, s, prec, nullptr, 1, 0,
                               &e->rpn_head, e->rpn_headbuf[l], &fused)) < 0) return st;
            if (!fused && (st = run_conv(e->rpn_head, e->rpn_t, B, hs[l], wsz[l], 1, 0, false, e->rpn_headbuf[l], nullptr, 0, s, prec)) < 0) return st;
            const std::string nm = "rpn_head" + std::to_string(l + 2);
            set_named(e, nm.c_str(), e->rpn_headbuf[l], B, hs[l], wsz[l], RPN_HEAD_C);
        }
    }
    trunk_group.reset();
    }   // phase 0
    RpnLevels lv{};
    {
        static const int sizes[5] = {32, 64, 128, 256, 512};
        static const double ratios[3] = {0.5, 1.0, 2.0};
        int off = 0;
        for (int l = 0; l < 5; ++l) {
            lv.head[l] = e->rpn_headbuf[l];
            lv.h[l] = hs[l];
            lv.w[l] = wsz[l];
            lv.stride[l] = 4 << l;
            lv.anchor_off[l] = off;
            off += hs[l] * wsz[l] * RPN_A;
            const double area = (double)sizes[l] * sizes[l];
            for (int a = 0; a < 3; ++a) {
                const double w = std::sqrt(area / ratios[a]);
                const double h = ratios[a] * w;
                lv.base[l][a][0] = (float)(-w / 2.0);
                lv.base[l][a][1] = (float)(-h / 2.0);
                lv.base[l][a][2] = (float)(w / 2.0);
                lv.base[l][a][3] = (float)(h / 2.0);
            }
        }
        lv.total_anchors = off;
    }
    const int P = e->desc.post_nms_topk, D = e->desc.detections_per_image;
    FeatLevels fl{};
    for (int l = 0; l < 4; ++l) {
        fl.feat[l] = e->pfeat[l];
        fl.h[l] = hs[l];
        fl.w[l] = wsz[l];
        fl.scale[l] = 1.0f / (float)(4 << l);
    }
    fl.C = e->fpn_c;
    float* o_boxes = out->boxes ? out->boxes : e->o_boxes;
    float* o_scores = out->scores ? out->scores : e->o_scores;
    int* o_classes = out->classes ? out->classes : e->o_classes;
    int* o_count = out->count ? out->count : e->o_count;
    float* o_probs = out->mask_probs ? out->mask_probs : e->o_mask_probs;
    const int mrows = B * D;
    if (PH(1)) {
    { ProfScope ps(e, s, 3);
    if ((st = rpn_topk_decode_launch(lv, valid, B, e->desc.pre_nms_topk, e->key_ws, e->cand_boxes, e->cand_scores,
                                     e->cand_valid, e->cand_idx, s)) < 0) return st; }
    set_named(e, "rpn_cand_boxes", e->cand_boxes, B, RPN_LEVELS, RPN_CAND, 4);
    set_named(e, "rpn_cand_scores", e->cand_scores, B, RPN_LEVELS, RPN_CAND);
    set_named(e, "rpn_cand_valid", e->cand_valid, B, RPN_LEVELS, RPN_CAND);
    set_named(e, "rpn_cand_idx", e->cand_idx, B, RPN_LEVELS, RPN_CAND);
    { ProfScope ps(e, s, 3);
    if ((st = nms_launch(e->cand_boxes, nullptr, e->cand_valid, B * RPN_LEVELS, RPN_CAND, e->desc.rpn_nms_thresh,
                         e->nms_mask, e->rpn_keep, e->rpn_keep_count, RPN_CAND, s)) < 0) return st; }
    set_named(e, "rpn_keep", e->rpn_keep, B, RPN_LEVELS, RPN_CAND);
    set_named(e, "rpn_keep_count", e->rpn_keep_count, B, RPN_LEVELS);
    { ProfScope ps(e, s, 3);
    if ((st = rpn_merge_launch(e->cand_boxes, e->cand_scores, e->rpn_keep, e->rpn_keep_count, B, P, e->props,
                               e->prop_scores, e->prop_count, P, s)) < 0) return st; }
    set_named(e, "proposals", e->props, B, P, 4);
    set_named(e, "proposal_scores", e->prop_scores, B, P);
    set_named(e, "proposal_count", e->prop_count, B);
    // ---- box head -----------------------------------------------------------------------------------------------
    { ProfScope ps(e, s, 4);
    if ((st = roi_align_launch(fl, e->props, e->prop_count, B, P, 7, 0, e->pooled7, nullptr, prec, s)) < 0) return st; }
    set_named(e, "pooled7", e->pooled7, (int64_t)B * P, 7, 7, e->fpn_c, (e->desc.precision == TD_PRECISION_FP16 ? 2 : 4));
    }   // phase 1
    if (PH(2)) {
    {
    ProfScope box_group(e, s, 0, 0.0, 0.0, true);
    if ((st = run_conv(e->fc1, e->pooled7, B * P, 1, 1, 1, 0, true, e->fc1_out, nullptr, 0, s, prec)) < 0) return st;
    if ((st = run_conv(e->fc2, e->fc1_out, B * P, 1, 1, 1, 0, true, e->fc2_out, nullptr, 0, s, prec)) < 0) return st;
    if ((st = run_conv(e->pred, e->fc2_out, B * P, 1, 1, 1, 0, false, e->pred_out, nullptr, 0, s, prec)) < 0) return st;
    }
    set_named(e, "box_pred", e->pred_out, (int64_t)B * P, 6);
    }   // phase 2
    if (PH(3)) {
    { ProfScope ps(e, s, 5);
    if ((st = det_decode_launch(e->pred_out, 6, e->props, e->prop_count, valid, B, P, e->desc.score_thresh, e->dboxes,
                                e->dscores, e->dflags, s)) < 0) return st; }
    set_named(e, "det_all_boxes", e->dboxes, B, P, 4);
    set_named(e, "det_all_scores", e->dscores, B, P);
    set_named(e, "det_flags", e->dflags, B, P);
    { ProfScope ps(e, s, 5);
    if ((st = sort_boxes_launch(e->dboxes, e->dscores, e->dflags, e->prop_count, B, P, e->sboxes, e->sscores, e->sidx,
                                e->scount, s)) < 0) return st; }
    { ProfScope ps(e, s, 5);
    if ((st = nms_launch(e->sboxes, e->scount, nullptr, B, P, e->desc.nms_thresh, e->nms_mask, e->det_keep,
                         e->det_keep_count, D, s)) < 0) return st; }
    { ProfScope ps(e, s, 5);
    if ((st = det_finalize_launch(e->sboxes, e->sscores, e->det_keep, e->det_keep_count, valid, outsz, B, P, D,
                                  e->det_boxes_net, o_boxes, o_scores, o_classes, o_count, s)) < 0) return st; }
    set_named(e, "det_boxes_net", e->det_boxes_net, B, D, 4);
    // ---- mask head (compact rows: only live detections are computed) ------------------------------------------
    { ProfScope ps(e, s, 4);
    if ((st = roi_align_launch(fl, e->det_boxes_net, o_count, B, D, 14, 1, e->pooled14, e->total_rows, prec, s)) < 0) return st; }
    set_named(e, "pooled14", e->pooled14, mrows, 14, 14, e->fpn_c, (e->desc.precision == TD_PRECISION_FP16 ? 2 : 4));
    }   // phase 3
    if (PH(4)) {
    const void* mx = e->pooled14;
    void* mbuf[2] = {e->mbuf0, e->mbuf1};
    {
    ProfScope mask_group(e, s, 7, 0.0, 0.0, true);
    for (int i = 0; i < 4; ++i) {
        if ((st = run_conv(e->mask_fcn[i], mx, mrows, 14, 14, 1, 1, true, mbuf[i & 1], nullptr, 0, s, prec, e->total_rows, 196)) < 0) return st;
        mx = mbuf[i & 1];
    }
    if ((st = run_conv(e->deconv, mx, mrows, 14, 14, 1, 0, true, e->deconv_out, nullptr, 0, s, prec, e->total_rows, 196, 1)) < 0) return st;
    }
    }   // phase 4
    if (PH(5)) {
    { ProfScope ps(e, s, 6);
    if ((st = mask_predict_launch(e->deconv_out, e->mask_pred_w, e->mask_pred_b, e->deconv.cout / 4, mrows * 784,
                                  e->total_rows, 784, e->mask_logits, e->mask_probs_compact, prec, s)) < 0) return st; }
    set_named(e, "mask_logits", e->mask_logits, mrows, 28, 28);
    { ProfScope ps(e, s, 6);
    if ((st = mask_scatter_launch(e->mask_probs_compact, o_count, B, D, o_probs, s)) < 0) return st; }
    if (out->mask_bits) {
        { ProfScope ps(e, s, 6);
        if ((st = paste_masks_launch(o_probs, o_boxes, o_count, outsz, B, D, e->desc.mask_thresh, out->mask_region,
                                     reinterpret_cast<long long*>(out->mask_offset), out->mask_bits,
                                     out->mask_words_per_image, s)) < 0) return st; }
    }
    }   // phase 5
    return TD_OK;
}

td_status set_forward_ctx(td_engine* e, const void* images, int input_format, const int32_t* hw_valid,
                          const int32_t* hw_out, int B, int Hp, int Wp, const td_detections* out) {
    TD_REQUIRE(e && images && hw_valid && hw_out && out, "td_engine_forward: null argument");
    TD_REQUIRE(e->loaded, "td_engine_forward: load weights first");
    TD_REQUIRE(input_format == TD_INPUT_F32_CHW || input_format == TD_INPUT_U8_HWC, "td_engine_forward: bad input format %d", input_format);
    TD_REQUIRE(B >= 1 && Hp % 32 == 0 && Wp % 32 == 0 && Hp >= 64 && Wp >= 64, "td_engine_forward: bad batch geometry B=%d %dx%d", B, Hp, Wp);
    if (B > e->rB || Hp > e->rHp || Wp > e->rWp) {
        td_set_error("td_engine_forward: B=%d %dx%d exceeds the reserved B=%d %dx%d", B, Hp, Wp, e->rB, e->rHp, e->rWp);
        return TD_ERR_CAPACITY;
    }
    td_engine::FwdCtx& c = e->ctx;
    c.valid_ctx = false;
    for (int i = 0; i < B; ++i) {
        c.valid.h[i] = hw_valid[2 * i];
        c.valid.w[i] = hw_valid[2 * i + 1];
        c.outsz.h[i] = hw_out[2 * i];
        c.outsz.w[i] = hw_out[2 * i + 1];
        TD_REQUIRE(c.valid.h[i] >= 1 && c.valid.h[i] <= Hp && c.valid.w[i] >= 1 && c.valid.w[i] <= Wp, "td_engine_forward: image %d valid size %dx%d outside %dx%d", i, c.valid.h[i], c.valid.w[i], Hp, Wp);
        TD_REQUIRE(c.outsz.h[i] >= 1 && c.outsz.w[i] >= 1, "td_engine_forward: image %d has an empty output size", i);
    }
    if (out->mask_bits)
        TD_REQUIRE(out->mask_region && out->mask_offset && out->mask_words_per_image > 0, "td_engine_forward: mask_bits needs mask_region, mask_offset and mask_words_per_image");
    c.images = images;
    c.input_format = input_format;
    c.B = B;
    c.Hp = Hp;
    c.Wp = Wp;
    c.out = *out;
    c.valid_ctx = true;
    return TD_OK;
}

}  // namespace

extern "C" {

td_status td_engine_forward(td_engine* e, const void* images, int input_format, const int32_t* hw_valid,
                            const int32_t* hw_out, int B, int Hp, int Wp, void* stream_v, td_detections* out) {
    td_status st = set_forward_ctx(e, images, input_format, hw_valid, hw_out, B, Hp, Wp, out);
    if (st < 0) return st;
    e->stem_done = false;
    return forward_impl(e, 0x3fu, static_cast<hipStream_t>(stream_v));
}

td_status td_engine_forward_phase(td_engine* e, int phase, const void* images, int input_format, const int32_t* hw_valid,
                                  const int32_t* hw_out, int B, int Hp, int Wp, void* stream_v, td_detections* out) {
    TD_REQUIRE(e && phase >= 0 && phase <= TD_PHASE_STEM, "td_engine_forward_phase: bad phase %d", phase);
    hipStream_t s = static_cast<hipStream_t>(stream_v);
    td_status st;
    if (phase == TD_PHASE_STEM) {
        // pre-phase of the NEXT batch: needs the engine's stem / pool buffers, which the previous batch's trunk read
        if ((st = set_forward_ctx(e, images, input_format, hw_valid, hw_out, B, Hp, Wp, out)) < 0) return st;
        // (the previous batch's LATER phases do not touch stem_out / pool_out: waiting for its phase 5 here put the
        // pre-phase behind the whole selection tail and stalled the next trunk 0.8 ms per fp16 step)
        if (e->phase_ev_recorded[0]) TD_HIP_CHECK(hipStreamWaitEvent(s, e->phase_ev[0], 0));
        e->stem_done = false;
    } else if (phase == 0) {
        if (e->stem_done) {
            TD_REQUIRE(e->ctx.valid_ctx && (!images || images == e->ctx.images),
                       "td_engine_forward_phase: phase 0 after the stem pre-phase must continue the same batch");
            TD_HIP_CHECK(hipStreamWaitEvent(s, e->phase_ev[TD_PHASE_STEM], 0));
        } else if ((st = set_forward_ctx(e, images, input_format, hw_valid, hw_out, B, Hp, Wp, out)) < 0) {
            return st;
        }
        // The engine's previous batch must have left the buffers the trunk writes. The FPN levels and RPN head maps are
        // read last by RoIAlign 14x14 in phase 3, but phase 4 (the mask-head 3x3 convs) goes through run_conv like any
        // other layer: in the fp32 engine its Winograd transforms use e->wino_v / e->wino_m, the workspace the next
        // trunk's Winograd layers write. So the trunk waits for the previous batch's phase 4 (which implies 0-3). Its
        // phase 5 (predictor, scatter, paste) only reads deconv_out / total_rows / the mask buffers, which the next
        // batch rewrites in phases 3 and 4: that dependency is kept in its own slot (prev5_ev) and taken by the next
        // batch's phase 3 — waiting for phase 5 HERE chained every trunk behind the previous-but-two batch's mask tail
        // (0.5-0.8 ms of idle main stream per fp16 step, rocprof trace of round 2). With the even phases on one
        // stream, as the Predictor and bench.py enqueue them, the phase-4 wait is already satisfied by stream order.
        for (int k = 4; k >= 0; --k)
            if (e->phase_ev_recorded[k]) {
                TD_HIP_CHECK(hipStreamWaitEvent(s, e->phase_ev[k], 0));
                break;
            }
        if (e->phase_ev_recorded[5]) {            // hand the previous batch's phase-5 event to the slot phase 3 waits on
            std::swap(e->phase_ev[5], e->prev5_ev);
            e->prev5_recorded = true;
        }
    } else {
        TD_REQUIRE(e->ctx.valid_ctx, "td_engine_forward_phase: phase %d before phase 0", phase);
        TD_REQUIRE(e->phase_ev_recorded[phase - 1], "td_engine_forward_phase: phase %d before phase %d", phase, phase - 1);
        TD_HIP_CHECK(hipStreamWaitEvent(s, e->phase_ev[phase - 1], 0));
        if (phase == 3 && e->prev5_recorded) {    // the previous batch's predictor / paste still read what phases 3-4 rewrite
            TD_HIP_CHECK(hipStreamWaitEvent(s, e->prev5_ev, 0));
            e->prev5_recorded = false;
        }
    }
    if ((st = forward_impl(e, 1u << phase, s)) < 0) return st;
    if (!e->phase_ev[phase]) TD_HIP_CHECK(hipEventCreateWithFlags(&e->phase_ev[phase], hipEventDisableTiming));
    TD_HIP_CHECK(hipEventRecord(e->phase_ev[phase], s));
    e->phase_ev_recorded[phase] = true;
    if (phase == TD_PHASE_STEM) e->stem_done = true;
    if (phase == 0) {
        e->stem_done = false;
        for (int k = 1; k < 6; ++k) e->phase_ev_recorded[k] = false;
    }
    return TD_OK;
}

td_status td_engine_tensor(td_engine* e, const char* name, void** dev_ptr, int64_t dims[4], int* elem_size) {
    TD_REQUIRE(e && name && dev_ptr && dims, "td_engine_tensor: null argument");
    auto it = e->named.find(name);
    if (it == e->named.end()) {
        td_set_error("td_engine_tensor: no tensor named '%s' (run forward first)", name);
        return TD_ERR_INVALID;
    }
    *dev_ptr = it->second.p;
    for (int i = 0; i < 4; ++i) dims[i] = it->second.dims[i];
    if (elem_size) *elem_size = it->second.elem;
    return TD_OK;
}

td_status td_engine_profile_enable(td_engine* e, int enable) {
    TD_REQUIRE(e, "td_engine_profile_enable: null engine");
    e->prof = enable != 0;
    e->prof_detail = enable == 2;
    return TD_OK;
}

td_status td_engine_profile_classes(td_engine* e, double* ms, int64_t* launches, double* exec_flops, double* bytes, double* tmin_ms, int reset) {
    TD_REQUIRE(e, "td_engine_profile_classes: null engine");
    for (auto& r : e->cls_recs) {
        TD_HIP_CHECK(hipEventSynchronize(r.b));
        float t = 0.f;
        TD_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        e->cls_ms[r.cls] += t;
        e->prof_free.push_back(r.a);
        e->prof_free.push_back(r.b);
    }
    e->cls_recs.clear();
    for (int c = 0; c < TD_PROF_CLASSES; ++c) {
        if (ms) ms[c] = e->cls_ms[c];
        if (launches) launches[c] = e->cls_launches[c];
        if (exec_flops) exec_flops[c] = e->cls_flops[c];
        if (bytes) bytes[c] = e->cls_bytes[c];
        if (tmin_ms) tmin_ms[c] = e->cls_tmin_ms[c];
        if (reset) {
            e->cls_ms[c] = 0.0;
            e->cls_launches[c] = 0;
            e->cls_flops[c] = e->cls_bytes[c] = e->cls_tmin_ms[c] = 0.0;
        }
    }
    return TD_OK;
}

td_status td_engine_profile_read(td_engine* e, double* ms, int64_t* launches, double* flops, double* bytes, int reset) {
    TD_REQUIRE(e, "td_engine_profile_read: null engine");
    for (auto& r : e->prof_recs) {
        TD_HIP_CHECK(hipEventSynchronize(r.b));
        float t = 0.f;
        TD_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        e->prof_ms[r.cat] += t;
        e->prof_free.push_back(r.a);
        e->prof_free.push_back(r.b);
    }
    e->prof_recs.clear();
    for (int c = 0; c < TD_PROF_CATEGORIES; ++c) {
        if (ms) ms[c] = e->prof_ms[c];
        if (launches) launches[c] = e->prof_launches[c];
        if (flops) flops[c] = e->prof_flops[c];
        if (bytes) bytes[c] = e->prof_bytes[c];
        if (reset) {
            e->prof_ms[c] = 0.0;
            e->prof_launches[c] = 0;
            e->prof_flops[c] = 0.0;
            e->prof_bytes[c] = 0.0;
        }
    }
    return TD_OK;
}

td_status td_engine_read_tensor(td_engine* e, const char* name, void* dst_dev, int64_t bytes, void* stream) {
    TD_REQUIRE(e && name && dst_dev, "td_engine_read_tensor: null argument");
    auto it = e->named.find(name);
    if (it == e->named.end()) {
        td_set_error("td_engine_read_tensor: no tensor named '%s' (run forward first)", name);
        return TD_ERR_INVALID;
    }
    int64_t n = it->second.elem;
    for (int i = 0; i < 4; ++i)
        if (it->second.dims[i] > 0) n *= it->second.dims[i];
    TD_REQUIRE(bytes == n, "td_engine_read_tensor: '%s' is %lld bytes, caller passed %lld", name, (long long)n, (long long)bytes);
    TD_HIP_CHECK(hipMemcpyAsync(dst_dev, it->second.p, (size_t)n, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return TD_OK;
}

}  // extern "C"
