// Host-side runtime of the MI355X page-inference engine: weight store (safetensors image), folded
// parameter block, static memory planner, op list ("plan") per input shape, executor with optional
// per-op HIP-event profiling.  One Engine = one network on one device; not thread-safe.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "rd_kernels.h"

namespace rd {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define RD_CHECK(cond, msg)                                                          \
    do {                                                                             \
        if (!(cond)) throw ::rd::Error(std::string(msg) + " [" #cond "]");           \
    } while (0)
#define RD_HIP(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) throw ::rd::Error(std::string(#expr ": ") + hipGetErrorString(_e));          \
    } while (0)

// ------------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<int64_t> shape;
    std::string dtype;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    size_t numel() const {
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        return n;
    }
    const float* f32() const { return reinterpret_cast<const float*>(data); }
};

class WeightStore {
   public:
    void load_safetensors(const void* blob, size_t nbytes);  // copies the image; strips a leading "model."
    bool has(const std::string& name) const { return map_.count(name) != 0; }
    const HostTensor& get(const std::string& name) const;
    size_t size() const { return map_.size(); }
    void clear() { map_.clear(); blob_.clear(); blob_.shrink_to_fit(); }

   private:
    std::vector<uint8_t> blob_;
    std::unordered_map<std::string, HostTensor> map_;
};

// Folded, re-laid-out parameters in one device allocation.
class ParamBlock {
   public:
    ~ParamBlock();
    size_t add(const std::string& key, const std::vector<float>& v);
    size_t add_u16(const std::string& key, const std::vector<uint16_t>& v);  // packed into the same block
    bool has(const std::string& key) const { return off_.count(key) != 0; }
    void upload();
    const float* ptr(const std::string& key) const;
    const float* host_ptr(const std::string& key) const;   // before upload(): the folded host copy
    size_t bytes() const { return host_.size() * sizeof(float); }

   private:
    std::vector<float> host_;
    std::unordered_map<std::string, size_t> off_;
    float* dev_ = nullptr;
};

// ------------------------------------------------------------------------------------------------
struct Buf {
    int n = 0, h = 0, w = 0, c = 0;
    size_t off = 0, bytes = 0;
    int external = -1;  // >= 0: slot in the run-time external pointer table
    bool live = false;
};
struct TView {  // channel slice [coff, coff+c) of an NHWC buffer
    int buf = -1, coff = 0;
    int n = 0, h = 0, w = 0, c = 0;
    long pixels() const { return (long)n * h * w; }
};

struct RunCtx {
    uint8_t* arena = nullptr;
    std::vector<void*> ext;
    hipStream_t stream = nullptr;
};

struct OpRecord {
    std::string name, kind, cfg, shape;
    double flops = 0, bytes = 0;
    std::function<void(const struct Plan&, const RunCtx&)> run;
};

// One captured replay of a plan: valid for exactly these external pointers, this workspace and this stream (the kernel
// arguments inside the graph are raw addresses).
struct GraphSlot {
    std::vector<void*> ext;
    uint8_t* arena = nullptr;
    hipStream_t stream = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t last_use = 0;
};

struct Plan {
    ~Plan();
    std::vector<Buf> bufs;
    std::vector<OpRecord> ops;
    size_t arena_bytes = 0;
    mutable uint64_t last_use = 0;   // Engine::plan_for's LRU clock
    mutable std::vector<GraphSlot> graphs;   // Engine::run: hipGraph replays of this plan (at most kMaxGraphSlots)
    mutable bool graph_broken = false;       // a capture of this plan failed once: launch directly from then on
    mutable unsigned graph_evictions = 0;    // slots dropped for newer pointer sets: past kMaxGraphEvictions the plan stops capturing
    std::vector<TView> outputs;  // model specific
    float* vptr(const TView& v, const RunCtx& c) const {
        const Buf& b = bufs[v.buf];
        uint8_t* base = b.external >= 0 ? (uint8_t*)c.ext[b.external] : c.arena + b.off;
        return reinterpret_cast<float*>(base) + v.coff;
    }
    int ld(const TView& v) const { return bufs[v.buf].c; }
};

struct ProfileEntry {
    std::string name, kind, cfg, shape;
    double flops, bytes;
    float ms;
};

enum class Mode { PREPARE, PLAN };

// Network description helper shared by every model builder.  In PREPARE mode it folds weights into the
// parameter block; in PLAN mode it allocates buffers and records kernel launches for one input shape.
class Builder {
   public:
    Builder(Mode m, const WeightStore* ws, ParamBlock* pb, Plan* plan, bool h3 = false, bool mixer_h3 = false,
            unsigned* range_flag = nullptr)
        : mode_(m), h3_(h3), mixer_h3_(mixer_h3), range_flag_(range_flag), ws_(ws), pb_(pb), plan_(plan) {}
    bool h3() const { return h3_; }
    Mode mode() const { return mode_; }
    bool planning() const { return mode_ == Mode::PLAN; }

    // --- memory
    TView alloc(int n, int h, int w, int c);
    TView external(int slot, int n, int h, int w, int c);
    TView slice(const TView& v, int coff, int c) const;
    TView reshape(const TView& v, int n, int h, int w) const;  // same pixel count, full-width view only
    void release(const TView& v);
    TView alloc_raw(size_t nfloats);  // scratch as [1,1,1,n]

    // The recogniser's per-line width table (rd_kernels.h LineTab, an int32 external [B][4]): from here on stem_front, dwconv,
    // se_gate and avgpool3x2 treat image n as line_tab[n]-wide (REC_LINE_WIDTHS plans only).
    void set_line_table(const TView& t) { lt_ = t; has_lt_ = true; }
    // zero the columns >= line_tab[n][col] of v (no-op without a line table): the separate-kernel stem of the fp32 mode
    void mask_cols(const TView& v, int col);

    // --- layers.  `key` = reference state-dict prefix of the layer.
    struct ConvGeom {
        int kh = 1, kw = 1, sh = 1, sw = 1, pt = 0, pl = 0, pb = 0, pr = 0;
    };
    // conv weight `wname` [Cout,Cin,kh,kw] (+ optional bias `bname`) (+ optional BatchNorm `bn` prefix, folded)
    TView conv(const std::string& wname, const std::string& bname, const std::string& bn, const TView& x,
               const ConvGeom& g, int act, const TView* out = nullptr, const TView* res = nullptr,
               const TView* ascale = nullptr);
    // fused PPLCNetV4 residual channel mixer (prefix.channel_conv1/2); gate = optional SE gate [N,1,1,C]
    // `dw_key`: (round 6) x is the block's INPUT and the mixer computes the block's depthwise 3x3 / stride 1 itself from the weights
    // Builder::dwconv folded under that key (only where mixer_takes_dw says so)
    TView mixer_fused(const std::string& prefix, const TView& x, const TView* gate, const std::string* dw_key = nullptr);
    // true when, in THIS plan, the channel mixer of `prefix` will run on the kernel that can take the block's depthwise conv along
    bool mixer_takes_dw(const std::string& prefix, int C) const;
    static std::string dw_key(const std::string& wname, const std::string& bn) { return wname + "|" + bn + "|dw"; }
    void fold_conv(const std::string& wname, const std::string& bname, const std::string& bn);
    TView linear(const std::string& prefix, const TView& x, int act, const TView* out = nullptr,
                 const TView* res = nullptr);
    // ConvTranspose2d(2, 2) + BN + ReLU -> ConvTranspose2d(2, 2) to ONE channel -> sigmoid, fused when the widths allow (DB head tail)
    bool deconv_pair_to_prob(const std::string& w1n, const std::string& b1n, const std::string& bn1, const std::string& w2n,
                             const std::string& b2n, const TView& x, const TView& out);
    TView deconv2x2(const std::string& wname, const std::string& bname, const std::string& bn, const TView& x, int act,
                    const TView* out = nullptr);
    TView stem3x3s2(const std::string& wname, const std::string& bn, const TView& x_nchw, int act);
    // stem1 -> [max-pool | stem2a -> stem2b] -> the 2 c1-channel concat buffer at half resolution (StemBlock front).  One fused
    // kernel in the split-fp16 precision modes (kernels_stem_fused.hip), the four separate kernels otherwise.
    TView stem_front(const std::string& w1, const std::string& bn1, const std::string& w2a, const std::string& bn2a,
                     const std::string& w2b, const std::string& bn2b, const TView& x_nchw);
    // stem3 (3x3 / stride 2 / pad 1 + BN + act) -> stem4 (1x1 + BN + act): one fused kernel in the split-fp16 precision modes
    // (kernels_stem34.hip, round 6), the two convolutions otherwise.  `out`: where stem4's output goes (a channel slot of a concat buffer)
    TView stem_tail(const std::string& w3, const std::string& bn3, const std::string& w4, const std::string& bn4, const TView& x, int act3,
                    int act4, const TView* out = nullptr);
    struct GapOut { TView partial; int chunks = 0; };  // per-image partial sums of a layer's output (SE pooling)
    TView dwconv(const std::string& wname, const std::string& bname, const std::string& bn, const TView& x,
                 const ConvGeom& g, int act, const TView* out = nullptr, const TView* res = nullptr, GapOut* gap = nullptr,
                 const TView* tokinfo = nullptr);   // tokinfo: ragged rows (DwParams::tokinfo), an int32 external of x.pixels() entries
    void maxpool2x2s1(const TView& x, const TView& out);
    TView avgpool3x2(const TView& x, const TView* out = nullptr);
    // squeeze-excite gate s[n][c]; `w1/b1/w2/b2` full tensor names
    TView se_gate(const std::string& w1, const std::string& b1, const std::string& w2, const std::string& b2,
                  const TView& x, int gate_act, const GapOut* pre = nullptr);
    void scale(const TView& x, const TView& gate, float alpha, const TView& out);
    void upsample(const TView& x, const TView& out, int f, bool accumulate);
    TView layernorm(const std::string& prefix, const TView& x, float eps);
    // seg: ragged batch - int32 external [B][2] = (first token, tokens) of every sequence; T = the longest sequence
    TView attention(const TView& qkv, int B, int T, int heads, int hd, const TView* seg = nullptr);
    TView add(const TView& a, const TView& b);
    void to_nchw(const TView& x, const TView& out_ext);
    void copy(const TView& x, const TView& out);   // same geometry, possibly different channel strides
    void ctc_stats(const TView& logits, const TView& idx_ext, const TView& prob_ext);
    void softmax_rows(const TView& logits, const TView& out_ext, const TView* idx_ext = nullptr, const TView* prob_ext = nullptr);
    void ctc_head(const std::string& prefix, const TView& x, const TView& idx_ext, const TView& prob_ext);

    const HostTensor& weight(const std::string& name) const { return ws_->get(name); }
    bool has_weight(const std::string& name) const { return ws_ && ws_->has(name); }
    int weight_dim(const std::string& name, int d) const;

   private:
    void emit(OpRecord&& r) { plan_->ops.push_back(std::move(r)); }
    std::vector<float> bn_scale_shift(const std::string& bn, int c, std::vector<float>& shift) const;
    Mode mode_;
    bool h3_ = false;        // every dense layer on the fp16 matrix cores with hi/lo operand splitting (fp32-accurate)
    bool mixer_h3_ = false;  // only the fused channel mixers (the default "auto" precision)
    unsigned* range_flag_ = nullptr;   // device word the split kernels raise when an operand leaves the fp16 range
    const WeightStore* ws_;
    ParamBlock* pb_;
    Plan* plan_;
    struct Block { size_t off, size; bool free; };
    std::vector<Block> blocks_;
    TView lt_;
    bool has_lt_ = false;
};

// ------------------------------------------------------------------------------------------------
class Engine {
   public:
    explicit Engine(int device, const std::string& kind);
    ~Engine();
    const std::string& kind() const { return kind_; }
    void load_weights(const void* blob, size_t nbytes);
    bool loaded() const { return loaded_; }

    // shape key -> plan (built lazily, cached)
    // flags: model-specific plan variants (RD_REC_* bits)
    const Plan& plan_for(int B, int H, int W, int flags = 0);
    size_t workspace_bytes(int B, int H, int W, int flags = 0) { return plan_for(B, H, W, flags).arena_bytes; }
    // run with caller-provided externals (model specific order); ws may be null (internal arena)
    void run(int B, int H, int W, int flags, const std::vector<void*>& ext, void* ws, size_t ws_bytes, hipStream_t s);

    // plans built / hipGraph captures / hipGraph replays since the handle was created (a document stream keeps meeting new shapes:
    // bench.py reports how many of them fell into its timed region)
    uint64_t plans_built() const { return plans_built_; }
    uint64_t graph_captures() const { return graph_captures_; }
    uint64_t graph_replays() const { return graph_replays_; }
    void set_profiling(bool on) { profiling_ = on; }
    const std::vector<ProfileEntry>& last_profile() const { return profile_; }
    std::string profile_json() const;
    int n_classes() const { return n_classes_; }
    int rec_token_dim() const { return rec_token_dim_; }   // channels of the rec backbone's pooled tokens (two-stage form)
    bool h3() const { return precision_ == PREC_H3; }
    int device() const { return device_; }
    // Arithmetic of the dense layers.  Every mode returns fp32 results with fp32-level error:
    //   PREC_AUTO  fp32 MFMA everywhere except the fused PPLCNetV4 channel mixers, which run on the fp16 matrix cores with
    //              (hi, lo) operand splitting (3 MFMAs per product, fp32 accumulate; measured error vs fp64 is BELOW the
    //              fp32 MFMA path's).  Split operands must stay inside the fp16 range (|v| < 65504): the kernels raise
    //              the range flag otherwise and the caller re-runs in PREC_FP32 (take_range_flag()).
    //   PREC_FP32  native fp32 MFMA only.      PREC_H3  every dense layer split (experimental; needs RD_PRECISION=h3 at load)
    enum Precision : int { PREC_AUTO = 0, PREC_FP32 = 1, PREC_H3 = 2 };
    void set_precision(int p);
    int precision() const { return precision_; }
    // 1 if a split kernel saw an out-of-range operand since the last call (synchronises the stream); clears the flag
    int take_range_flag(hipStream_t s);

    std::string last_error;

   private:
    void build(Builder& b, int B, int H, int W, int flags);
    int device_;
    std::string kind_;
    bool loaded_ = false;
    bool profiling_ = false;
    int precision_ = PREC_AUTO;
    bool h3_prepared_ = false;
    unsigned* range_flag_ = nullptr;          // device address of ...
    unsigned* range_flag_host_ = nullptr;     // ... this word of pinned, mapped host memory
    int n_classes_ = 0;
    int rec_token_dim_ = 0;
    ParamBlock params_;
    WeightStore store_;
    std::map<std::tuple<int, int, int, int>, std::unique_ptr<Plan>> plans_;
    uint64_t plan_clock_ = 0;
    uint64_t plans_built_ = 0, graph_captures_ = 0, graph_replays_ = 0;
    static constexpr size_t kMaxPlans = 512;   // least recently used plans are dropped beyond this
    static constexpr size_t kMaxGraphSlots = 8;
    static constexpr unsigned kMaxGraphEvictions = 32;   // a plan whose external pointers keep changing never replays: stop capturing it
    // Captured execs that lost their slot.  An exec may have been launched microseconds ago on an asynchronous stream, so it is
    // parked with an event recorded on that stream and destroyed only once the event has completed (sweep_retired_graphs).
    struct RetiredGraph { hipGraphExec_t exec; hipEvent_t done; };
    std::vector<RetiredGraph> retired_graphs_;
    void retire_graph(hipGraphExec_t exec, hipStream_t s);
    void sweep_retired_graphs(bool wait);
    bool graphs_ = true;                       // RD_GRAPHS=0 switches the hipGraph replays off
    uint8_t* arena_ = nullptr;
    size_t arena_bytes_ = 0;
    std::vector<ProfileEntry> profile_;
    std::vector<hipEvent_t> events_;
};

// model builders (models.cpp)
void build_ppocrv6_det(Builder& b, int B, int H, int W);
// REC_STAGE_BACKBONE: image -> pooled tokens only (ext[1] = [B][T][C] out).  REC_STAGE_TAIL: the LightSVTR neck + CTC head over
// the tokens of MANY batches at once (B = text lines, H = longest line in tokens, W = all tokens; ext: 0 tokens, 1 idx,
// 2 prob, 4 seg int32 [B][2], 5 tokinfo int32 [W]) - see build_ppocrv6_rec
// REC_LINE_WIDTHS (with REC_STAGE_BACKBONE): ext[2] = LineTab int32 [B][4] (rd_kernels.h) - every line is computed at ITS OWN padded
// width inside the shared [B,3,48,W] tensor and writes its own number of tokens at its own offset of ext[1]
enum RecFlags : int { REC_UNFUSED_CTC = 1, REC_WANT_SOFTMAX = 2, REC_WANT_LOGITS = 4, REC_STAGE_BACKBONE = 8, REC_STAGE_TAIL = 16, REC_LINE_WIDTHS = 32 };
void build_ppocrv6_rec(Builder& b, int B, int H, int W, int flags);
void build_pphgnetv2_b4(Builder& b, int B, int H, int W);
// PP-FormulaNet_plus encoder; flags bit 0: the caller's image has 1 channel (replicated to 3 like the reference)
void build_pphgnetv2_b6_formula(Builder& b, int B, int H, int W, int flags);

}  // namespace rd
