#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace rd {

// =================================================================================================
// safetensors image -> WeightStore.  Format: u64 LE header length, JSON header
// {"name": {"dtype": "F32", "shape": [...], "data_offsets": [b, e]}, "__metadata__": {...}}, raw data.
// (what the reference loads with safetensors.torch.load_file, rapid_doc/model/ocr/torch.py:93-110)
// =================================================================================================
namespace {
struct JsonCur {
    const char* p;
    const char* end;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool eat(char c) {
        ws();
        if (p < end && *p == c) { ++p; return true; }
        return false;
    }
    void expect(char c) {
        if (!eat(c)) throw Error(std::string("safetensors header: expected '") + c + "'");
    }
    std::string str() {
        ws();
        if (p >= end || *p != '"') throw Error("safetensors header: expected string");
        ++p;
        std::string out;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'u': out += '?'; p += 4; break;
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) throw Error("safetensors header: unterminated string");
        ++p;
        return out;
    }
    int64_t integer() {
        ws();
        bool neg = false;
        if (p < end && *p == '-') { neg = true; ++p; }
        if (p >= end || *p < '0' || *p > '9') throw Error("safetensors header: expected integer");
        int64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
        return neg ? -v : v;
    }
    void skip_value() {
        ws();
        if (p >= end) throw Error("safetensors header: truncated");
        if (*p == '"') { str(); return; }
        if (*p == '{') {
            ++p;
            if (eat('}')) return;
            do { str(); expect(':'); skip_value(); } while (eat(','));
            expect('}');
            return;
        }
        if (*p == '[') {
            ++p;
            if (eat(']')) return;
            do { skip_value(); } while (eat(','));
            expect(']');
            return;
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']') ++p;
    }
};
size_t dtype_size(const std::string& d) {
    if (d == "F32" || d == "I32" || d == "U32") return 4;
    if (d == "F64" || d == "I64" || d == "U64") return 8;
    if (d == "F16" || d == "BF16" || d == "I16" || d == "U16") return 2;
    if (d == "U8" || d == "I8" || d == "BOOL") return 1;
    throw Error("safetensors: unsupported dtype " + d);
}
}  // namespace

void WeightStore::load_safetensors(const void* blob, size_t nbytes) {
    RD_CHECK(blob && nbytes >= 8, "weights: empty image");
    uint64_t hlen = 0;
    std::memcpy(&hlen, blob, 8);
    RD_CHECK(hlen > 0 && 8 + hlen <= nbytes, "weights: bad safetensors header length");
    blob_.assign((const uint8_t*)blob, (const uint8_t*)blob + nbytes);
    map_.clear();
    const uint8_t* base = blob_.data() + 8 + hlen;
    const size_t data_bytes = nbytes - 8 - hlen;
    JsonCur c{(const char*)blob_.data() + 8, (const char*)blob_.data() + 8 + hlen};
    c.expect('{');
    if (!c.eat('}')) {
        do {
            std::string name = c.str();
            c.expect(':');
            if (name == "__metadata__") { c.skip_value(); continue; }
            HostTensor t;
            int64_t b = -1, e = -1;
            c.expect('{');
            do {
                std::string k = c.str();
                c.expect(':');
                if (k == "dtype") t.dtype = c.str();
                else if (k == "shape") {
                    c.expect('[');
                    if (!c.eat(']')) {
                        do { t.shape.push_back(c.integer()); } while (c.eat(','));
                        c.expect(']');
                    }
                } else if (k == "data_offsets") {
                    c.expect('[');
                    b = c.integer();
                    c.expect(',');
                    e = c.integer();
                    c.expect(']');
                } else c.skip_value();
            } while (c.eat(','));
            c.expect('}');
            RD_CHECK(b >= 0 && e >= b && (size_t)e <= data_bytes, "weights: tensor offsets out of range: " + name);
            RD_CHECK((size_t)(e - b) == t.numel() * dtype_size(t.dtype), "weights: tensor byte size mismatch: " + name);
            t.data = base + b;
            t.nbytes = (size_t)(e - b);
            if (name.rfind("model.", 0) == 0) name = name.substr(6);  // reference torch.py:105-110
            map_[name] = std::move(t);
        } while (c.eat(','));
        c.expect('}');
    }
    RD_CHECK(!map_.empty(), "weights: no tensors in image");
}

const HostTensor& WeightStore::get(const std::string& name) const {
    auto it = map_.find(name);
    if (it == map_.end()) throw Error("weights: missing tensor '" + name + "'");
    if (it->second.dtype != "F32") throw Error("weights: tensor '" + name + "' is " + it->second.dtype + ", expected F32");
    return it->second;
}

// =================================================================================================
ParamBlock::~ParamBlock() {
    if (dev_) (void)hipFree(dev_);
}
size_t ParamBlock::add(const std::string& key, const std::vector<float>& v) {
    RD_CHECK(!dev_, "ParamBlock: add after upload");
    auto it = off_.find(key);
    if (it != off_.end()) return it->second;
    size_t off = (host_.size() + 63) & ~size_t(63);  // 256-byte aligned
    host_.resize(off + v.size(), 0.f);
    std::copy(v.begin(), v.end(), host_.begin() + off);
    off_[key] = off;
    return off;
}
size_t ParamBlock::add_u16(const std::string& key, const std::vector<uint16_t>& v) {
    std::vector<float> packed((v.size() + 1) / 2, 0.f);
    std::memcpy(packed.data(), v.data(), v.size() * sizeof(uint16_t));
    return add(key, packed);
}
void ParamBlock::upload() {
    if (dev_) { (void)hipFree(dev_); dev_ = nullptr; }
    size_t n = std::max<size_t>(host_.size(), 64) + 64;
    RD_HIP(hipMalloc((void**)&dev_, n * sizeof(float)));
    RD_HIP(hipMemcpy(dev_, host_.data(), host_.size() * sizeof(float), hipMemcpyHostToDevice));
}
const float* ParamBlock::ptr(const std::string& key) const {
    auto it = off_.find(key);
    if (it == off_.end()) throw Error("params: missing folded tensor '" + key + "'");
    RD_CHECK(dev_, "params: not uploaded");
    return dev_ + it->second;
}

const float* ParamBlock::host_ptr(const std::string& key) const {
    auto it = off_.find(key);
    if (it == off_.end()) throw Error("params: missing folded tensor '" + key + "'");
    return host_.data() + it->second;
}

// =================================================================================================
// Builder: memory
// =================================================================================================
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

TView Builder::alloc(int n, int h, int w, int c) {
    RD_CHECK(n > 0 && h > 0 && w > 0 && c > 0, "alloc: empty tensor");
    Buf b;
    b.n = n; b.h = h; b.w = w; b.c = c;
    b.bytes = align_up((size_t)n * h * w * c * sizeof(float), 256);
    b.live = true;
    // first fit
    size_t found = blocks_.size();
    for (size_t i = 0; i < blocks_.size(); ++i)
        if (blocks_[i].free && blocks_[i].size >= b.bytes) { found = i; break; }
    if (found == blocks_.size()) {
        // grow: extend a trailing free block or append
        if (!blocks_.empty() && blocks_.back().free) {
            blocks_.back().size = b.bytes;
            found = blocks_.size() - 1;
        } else {
            size_t off = blocks_.empty() ? 0 : blocks_.back().off + blocks_.back().size;
            blocks_.push_back({off, b.bytes, true});
            found = blocks_.size() - 1;
        }
    }
    Block& blk = blocks_[found];
    if (blk.size > b.bytes) {
        Block rest{blk.off + b.bytes, blk.size - b.bytes, true};
        blk.size = b.bytes;
        blocks_.insert(blocks_.begin() + found + 1, rest);
    }
    blocks_[found].free = false;
    b.off = blocks_[found].off;
    plan_->arena_bytes = std::max(plan_->arena_bytes, blocks_.back().off + blocks_.back().size);
    plan_->bufs.push_back(b);
    TView v;
    v.buf = (int)plan_->bufs.size() - 1;
    v.coff = 0; v.n = n; v.h = h; v.w = w; v.c = c;
    return v;
}
TView Builder::alloc_raw(size_t nfloats) { return alloc(1, 1, 1, (int)align_up(nfloats, 4)); }

TView Builder::external(int slot, int n, int h, int w, int c) {
    Buf b;
    b.n = n; b.h = h; b.w = w; b.c = c;
    b.external = slot;
    b.live = true;
    plan_->bufs.push_back(b);
    TView v;
    v.buf = (int)plan_->bufs.size() - 1;
    v.n = n; v.h = h; v.w = w; v.c = c;
    return v;
}
TView Builder::slice(const TView& v, int coff, int c) const {
    RD_CHECK(coff >= 0 && coff + c <= v.c, "slice out of range");
    TView s = v;
    s.coff = v.coff + coff;
    s.c = c;
    return s;
}
TView Builder::reshape(const TView& v, int n, int h, int w) const {
    RD_CHECK((long)n * h * w == v.pixels(), "reshape: pixel count mismatch");
    TView s = v;
    s.n = n; s.h = h; s.w = w;
    return s;
}
void Builder::release(const TView& v) {
    Buf& b = plan_->bufs[v.buf];
    if (b.external >= 0 || !b.live) return;
    b.live = false;
    for (size_t i = 0; i < blocks_.size(); ++i) {
        if (blocks_[i].off == b.off && !blocks_[i].free) {
            blocks_[i].free = true;
            if (i + 1 < blocks_.size() && blocks_[i + 1].free) {
                blocks_[i].size += blocks_[i + 1].size;
                blocks_.erase(blocks_.begin() + i + 1);
            }
            if (i > 0 && blocks_[i - 1].free) {
                blocks_[i - 1].size += blocks_[i].size;
                blocks_.erase(blocks_.begin() + i);
            }
            return;
        }
    }
}

int Builder::weight_dim(const std::string& name, int d) const {
    const HostTensor& t = ws_->get(name);
    RD_CHECK(d < (int)t.shape.size(), "weight_dim: rank");
    return (int)t.shape[d];
}

// BatchNorm (eval) as y = x*scale + shift, eps = 1e-5 (nn.BatchNorm2d default)
std::vector<float> Builder::bn_scale_shift(const std::string& bn, int c, std::vector<float>& shift) const {
    std::vector<float> scale(c, 1.f);
    shift.assign(c, 0.f);
    if (bn.empty()) return scale;
    const float* g = ws_->get(bn + ".weight").f32();
    const float* b = ws_->get(bn + ".bias").f32();
    const float* m = ws_->get(bn + ".running_mean").f32();
    const float* v = ws_->get(bn + ".running_var").f32();
    RD_CHECK((int)ws_->get(bn + ".weight").numel() == c, "BN channel mismatch: " + bn);
    for (int i = 0; i < c; ++i) {
        const double s = (double)g[i] / std::sqrt((double)v[i] + 1e-5);
        scale[i] = (float)s;
        shift[i] = (float)((double)b[i] - (double)m[i] * s);
    }
    return scale;
}

// weights beyond the fp16 range cannot be split: such a layer stays on the fp32 MFMA kernel in every precision mode
static bool fits_fp16_range(const std::vector<float>& w) {
    for (float v : w)
        if (!(std::fabs(v) < 65504.f)) return false;
    return true;
}
static inline int out_dim(int in, int k, int s, int p0, int p1) { return (in + p0 + p1 - k) / s + 1; }
// "auto" precision: wide pointwise convolutions also run on the split-fp16 matrix-core path (256x128 tile; measured
// 1.2x the fp32 MFMA kernel at K = 192, 2x at K >= 768 - table in kernels_conv_h3.hip)
static inline bool auto_split_conv(int kh, int kw, int K, int cout) {
    static const int kxk_k = std::getenv("RD_H3_KXK_MIN_K") ? std::atoi(std::getenv("RD_H3_KXK_MIN_K")) : 96;   // round 2: 96 / 24 (round 1: 288 / 48) measured +2.8 % pages/s
    static const int kxk_n = std::getenv("RD_H3_KXK_MIN_N") ? std::atoi(std::getenv("RD_H3_KXK_MIN_N")) : 24;
    static const int pw_k = std::getenv("RD_H3_1X1_MIN_K") ? std::atoi(std::getenv("RD_H3_1X1_MIN_K")) : 96;
    static const int pw_n = std::getenv("RD_H3_1X1_MIN_N") ? std::atoi(std::getenv("RD_H3_1X1_MIN_N")) : 96;
    static const bool stream_off = std::getenv("RD_CONV_STREAM") && std::getenv("RD_CONV_STREAM")[0] == '0';
    // small-K layers: the streaming split kernel (kernels_conv_stream_h3.hip) takes them whatever the thresholds below say
    if (!stream_off && cout >= 12 && conv_stream_h3_shape_ok(kh, kw, K / (kh * kw), cout)) return true;
    return (kh == 1 && kw == 1) ? (K >= pw_k && cout >= pw_n) : (K >= kxk_k && cout >= kxk_n);
}

// =================================================================================================
// Builder: layers
// =================================================================================================
TView Builder::conv(const std::string& wname, const std::string& bname, const std::string& bn, const TView& x,
                    const ConvGeom& g, int act, const TView* out, const TView* res, const TView* ascale) {
    const HostTensor& w = ws_->get(wname);
    RD_CHECK(w.shape.size() == 4 || w.shape.size() == 2, "conv weight rank: " + wname);
    const int cout = (int)w.shape[0], cin = (int)w.shape[1];
    const int kh = w.shape.size() == 4 ? (int)w.shape[2] : 1, kw = w.shape.size() == 4 ? (int)w.shape[3] : 1;
    RD_CHECK(kh == g.kh && kw == g.kw, "conv kernel size mismatch: " + wname);
    RD_CHECK(cin == x.c, "conv Cin mismatch: " + wname + " expects " + std::to_string(cin) + " got " + std::to_string(x.c));
    RD_CHECK(cin % 4 == 0, "conv_igemm needs Cin % 4 == 0: " + wname);
    const int oh = out_dim(x.h, kh, g.sh, g.pt, g.pb), ow = out_dim(x.w, kw, g.sw, g.pl, g.pr);
    TView y = out ? *out : alloc(x.n, oh, ow, cout);
    RD_CHECK(y.n == x.n && y.h == oh && y.w == ow && y.c == cout, "conv output view mismatch: " + wname);
    if (res) RD_CHECK(res->n == y.n && res->h == oh && res->w == ow && res->c == cout, "conv residual mismatch: " + wname);
    const int K = kh * kw * cin;
    const std::string key = wname + "|" + bn;
    if (!planning()) {
        fold_conv(wname, bname, bn);
        return y;
    }
    ConvParams p{};
    p.xld = plan_->ld(x);
    p.N = x.n; p.H = x.h; p.W = x.w; p.Cin = cin;
    p.w = pb_->ptr(key + "#w");
    p.bias = pb_->has(key + "#b") ? pb_->ptr(key + "#b") : nullptr;
    p.yld = plan_->ld(y);
    p.OH = oh; p.OW = ow; p.Cout = cout;
    p.KH = kh; p.KW = kw; p.SH = g.sh; p.SW = g.sw; p.PT = g.pt; p.PL = g.pl;
    p.rld = res ? plan_->ld(*res) : 0;
    p.act = act;
    p.out_mode = OUT_NHWC;
    p.M = x.n * oh * ow; p.K = K; p.Ng = cout;
    OpRecord r;
    r.name = wname;
    r.kind = (kh == 1 && kw == 1) ? "conv1x1" : "conv" + std::to_string(kh) + "x" + std::to_string(kw);
    r.cfg = conv_igemm_config_name(p);
    r.shape = "M" + std::to_string(p.M) + "_K" + std::to_string(K) + "_N" + std::to_string(cout);
    r.flops = 2.0 * p.M * (double)K * cout;
    r.bytes = 4.0 * ((double)x.pixels() * cin + (double)p.M * cout * (res ? 2 : 1) + (double)cout * K);
    const TView xv = x, yv = y;
    const bool has_res = res != nullptr, has_as = ascale != nullptr;
    const TView rv = res ? *res : TView{}, av = ascale ? *ascale : TView{};
    // (routed by LAYER - kernel size, K, N - never by the row count M: an image's bits must not depend on the launch it rides in)
    const bool h3 = (h3_ || (mixer_h3_ && !ascale && auto_split_conv(kh, kw, K, cout))) &&
                    pb_->has(key + "#wh");
    if (h3) {
        p.wh = reinterpret_cast<const uint16_t*>(pb_->ptr(key + "#wh"));
        p.wl = reinterpret_cast<const uint16_t*>(pb_->ptr(key + "#wl"));
        if (pb_->has(key + "#w1")) {
            p.w1 = reinterpret_cast<const uint16_t*>(pb_->ptr(key + "#w1"));
            p.w1_inv = *pb_->host_ptr(key + "#w1s");
        }
        if (pb_->has(key + "#w3")) {
            p.w3 = reinterpret_cast<const uint16_t*>(pb_->ptr(key + "#w3"));
            p.w3_inv = *pb_->host_ptr(key + "#w3s");
        }
        p.range_flag = range_flag_;
        r.cfg = std::string(conv_stream_h3_applies(p) ? "stream" : conv3x3_h1_applies(p) ? "c3h1" : conv_direct_h3_applies(p) ? "direct" : gemm_h1_applies(p) ? "h1w256x128" : gemm_h3_dma_applies(p) ? (gemm_h3_dma_uses16(p) ? "dma16w256x128" : "dma256x128") : cout > 96 ? "256x128" : cout > 64 ? "128x96"
                            : (cout > 32 && p.M >= 65536) ? "256x64" : K <= 256 || cout <= 32 ? "128x32" : "128x64") + "/h3";
    }
    r.run = [p, xv, yv, rv, av, has_res, has_as, h3](const Plan& pl, const RunCtx& c) mutable {
        ConvParams q = p;
        q.x = pl.vptr(xv, c);
        q.y = pl.vptr(yv, c);
        q.res = has_res ? pl.vptr(rv, c) : nullptr;
        q.ascale = has_as ? pl.vptr(av, c) : nullptr;
        if (h3) launch_conv_igemm_h3(q, c.stream);
        else launch_conv_igemm(q, c.stream);
    };
    emit(std::move(r));
    return y;
}

void Builder::fold_conv(const std::string& wname, const std::string& bname, const std::string& bn) {
    const std::string key = wname + "|" + bn;
    if (pb_->has(key + "#w")) return;
    const HostTensor& w = ws_->get(wname);
    const int cout = (int)w.shape[0], cin = (int)w.shape[1];
    const int kh = w.shape.size() == 4 ? (int)w.shape[2] : 1, kw = w.shape.size() == 4 ? (int)w.shape[3] : 1;
    const int K = kh * kw * cin;
    std::vector<float> shift;
    std::vector<float> scale = bn_scale_shift(bn, cout, shift);
    std::vector<float> wf((size_t)cout * K);
    const float* src = w.f32();
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int a = 0; a < kh; ++a)
                for (int b = 0; b < kw; ++b)
                    wf[(size_t)co * K + (a * kw + b) * cin + ci] = src[(((size_t)co * cin + ci) * kh + a) * kw + b] * scale[co];
    std::vector<float> bias(cout, 0.f);
    bool any_bias = !bn.empty();
    if (!bname.empty()) {
        const float* bs = ws_->get(bname).f32();
        for (int co = 0; co < cout; ++co) bias[co] = bs[co] * scale[co];
        any_bias = true;
    }
    for (int co = 0; co < cout; ++co) bias[co] += shift[co];
    pb_->add(key + "#w", wf);
    if (any_bias) pb_->add(key + "#b", bias);
    if ((h3_ || (mixer_h3_ && auto_split_conv(kh, kw, K, cout))) && fits_fp16_range(wf)) {
        std::vector<uint16_t> hi, lo;
        split_weights_h3(wf.data(), cout, K, hi, lo);
        pb_->add_u16(key + "#wh", hi);
        pb_->add_u16(key + "#wl", lo);
        if (conv3x3_h1_shape_ok(kh, kw, cin, cout)) {               // one-accumulator direct 3x3: slab-ordered fragment image
            std::vector<uint16_t> img;
            const float inv = prepare_conv3x3_h1_weights(wf.data(), cout, cin, img);
            pb_->add_u16(key + "#w3", img);
            pb_->add(key + "#w3s", std::vector<float>{inv});
        }
        if (kh == 1 && kw == 1 && gemm_h1_shape_ok(K, cout)) {      // single-accumulator GEMM: fragment-ordered, per-channel pre-scaled image
            std::vector<uint16_t> img;
            const float inv = prepare_gemm_h1_weights(wf.data(), cout, K, img);
            pb_->add_u16(key + "#w1", img);
            pb_->add(key + "#w1s", std::vector<float>{inv});
        }
    }
}

bool Builder::mixer_takes_dw(const std::string& prefix, int C) const {
    if (!planning() || !(h3_ || mixer_h3_) || !mixer_res_fuses_dw(C)) return false;
    const std::string hkey = prefix + ".mixer#h3";
    static const bool ws_off = std::getenv("RD_MIXER_WS") && std::string(std::getenv("RD_MIXER_WS")) == "0";
    const bool split = pb_->has(hkey + ".w1h");
    const bool ws = split && !ws_off && pb_->has(hkey + ".ws");
    return split && !ws && pb_->has(hkey + ".res");       // the route Builder::mixer_fused takes to the resident-weights kernel
}

TView Builder::mixer_fused(const std::string& prefix, const TView& x, const TView* gate, const std::string* dw_key) {
    // prefix.channel_conv1 / channel_conv2 (+ normalization), residual block: out channels == in channels
    const std::string w1 = prefix + ".channel_conv1.convolution.weight", bn1 = prefix + ".channel_conv1.normalization";
    const std::string w2 = prefix + ".channel_conv2.convolution.weight", bn2 = prefix + ".channel_conv2.normalization";
    const int C = x.c;
    RD_CHECK(mixer_fused_supported(C), "mixer_fused: unsupported width");
    RD_CHECK(weight_dim(w1, 0) == 2 * C && weight_dim(w1, 1) == C && weight_dim(w2, 0) == C && weight_dim(w2, 1) == 2 * C,
             "mixer_fused: weight shapes: " + prefix);
    TView y = alloc(x.n, x.h, x.w, C);
    const std::string hkey = prefix + ".mixer#h3";
    if (!planning()) {
        fold_conv(w1, "", bn1);
        fold_conv(w2, "", bn2);
        const float* f1 = pb_->host_ptr(w1 + "|" + bn1 + "#w");
        const float* f2 = pb_->host_ptr(w2 + "|" + bn2 + "#w");
        const bool fits = fits_fp16_range(std::vector<float>(f1, f1 + (size_t)2 * C * C)) && fits_fp16_range(std::vector<float>(f2, f2 + (size_t)2 * C * C));
        if (!pb_->has(hkey + ".w1h") && fits) {   // split-fp16 copies for the default (auto) precision
            std::vector<uint16_t> v[4];
            prepare_mixer_weights_h3(f1, f2, C, v[0], v[1], v[2], v[3]);
            pb_->add_u16(hkey + ".w1h", v[0]); pb_->add_u16(hkey + ".w1l", v[1]);
            pb_->add_u16(hkey + ".w2h", v[2]); pb_->add_u16(hkey + ".w2l", v[3]);
        }
        if (!pb_->has(hkey + ".res") && fits && mixer_res_supported(C)) {   // fragment image of the resident-weights kernel
            std::vector<uint16_t> img;
            float inv[2];
            prepare_mixer_weights_res(f1, f2, C, img, inv);
            pb_->add_u16(hkey + ".res", img);
            pb_->add(hkey + ".resinv", std::vector<float>{inv[0], inv[1]});
        }
        if (!pb_->has(hkey + ".ws") && fits && mixer_ws_preferred(C)) {   // weight stream image of the ws kernel
            std::vector<uint16_t> img;
            float inv[2];
            prepare_mixer_weights_ws(f1, f2, C, img, inv);
            pb_->add_u16(hkey + ".ws", img);
            pb_->add(hkey + ".wsinv", std::vector<float>{inv[0], inv[1]});
        }
        return y;
    }
    const bool split = (h3_ || mixer_h3_) && pb_->has(hkey + ".w1h");
    static const bool ws_off = std::getenv("RD_MIXER_WS") && std::string(std::getenv("RD_MIXER_WS")) == "0";   // A/B switch
    const bool ws = split && !ws_off && pb_->has(hkey + ".ws");
    const bool res_k = split && !ws && mixer_res_supported(C) && pb_->has(hkey + ".res");
    MixerParams p{};
    if (res_k) {
        p.w1h = reinterpret_cast<const uint16_t*>(pb_->ptr(hkey + ".res"));
        p.ws_inv1 = pb_->host_ptr(hkey + ".resinv")[0];
        p.ws_inv2 = pb_->host_ptr(hkey + ".resinv")[1];
        p.range_flag = range_flag_;
    } else if (ws) {
        p.w1h = reinterpret_cast<const uint16_t*>(pb_->ptr(hkey + ".ws"));
        p.ws_inv1 = pb_->host_ptr(hkey + ".wsinv")[0];
        p.ws_inv2 = pb_->host_ptr(hkey + ".wsinv")[1];
        p.ws_pf = mixer_ws_prefetch() && C == 192;
        p.range_flag = range_flag_;
    } else if (split) {
        p.w1h = reinterpret_cast<const uint16_t*>(pb_->ptr(hkey + ".w1h")); p.w1l = reinterpret_cast<const uint16_t*>(pb_->ptr(hkey + ".w1l"));
        p.w2h = reinterpret_cast<const uint16_t*>(pb_->ptr(hkey + ".w2h")); p.w2l = reinterpret_cast<const uint16_t*>(pb_->ptr(hkey + ".w2l"));
        p.range_flag = range_flag_;
    }
    p.xld = plan_->ld(x);
    p.yld = plan_->ld(y);
    p.M = (int)x.pixels(); p.HW = x.h * x.w; p.C = C;
    const bool with_dw = dw_key != nullptr;
    if (with_dw) {
        RD_CHECK(res_k && !gate, "mixer_fused: the depthwise conv only rides in the resident-weights kernel, without a gate: " + prefix);
        RD_CHECK((size_t)x.pixels() * plan_->ld(x) < ((size_t)1 << 29), "mixer_fused + depthwise: the input tensor exceeds the 32-bit tap offsets");
        p.dw_w = pb_->ptr(*dw_key + "#w");
        p.dw_b = pb_->ptr(*dw_key + "#b");
        p.dwH = x.h; p.dwW = x.w;
    }
    p.w1 = pb_->ptr(w1 + "|" + bn1 + "#w"); p.b1 = pb_->ptr(w1 + "|" + bn1 + "#b");
    p.w2 = pb_->ptr(w2 + "|" + bn2 + "#w"); p.b2 = pb_->ptr(w2 + "|" + bn2 + "#b");
    OpRecord r;
    r.name = prefix + ".mixer";
    r.kind = res_k ? (with_dw ? "mixer_fused_res_dw" : "mixer_fused_res") : ws ? "mixer_fused_ws" : split ? "mixer_fused_h3" : "mixer_fused";
    r.cfg = "C" + std::to_string(C);
    r.shape = "M" + std::to_string(p.M) + "_C" + std::to_string(C);
    r.flops = 8.0 * p.M * (double)C * C + (with_dw ? 18.0 * p.M * C : 0.0);
    r.bytes = 8.0 * p.M * C;
    const TView xv = x, yv = y;
    const bool has_gate = gate != nullptr;
    const TView gv = gate ? *gate : TView{};
    const bool has_lt = with_dw && has_lt_;
    const TView ltv = lt_;
    r.run = [p, xv, yv, gv, has_gate, split, ws, res_k, has_lt, ltv](const Plan& pl, const RunCtx& c) {
        MixerParams q = p;
        q.x = pl.vptr(xv, c);
        q.y = pl.vptr(yv, c);
        q.gate = has_gate ? pl.vptr(gv, c) : nullptr;
        if (has_lt) {        // (as Builder::dwconv: column 2 of the line table = the line's width at this resolution)
            q.dw_line_w = reinterpret_cast<const int32_t*>(pl.vptr(ltv, c)) + 2;
            q.dw_line_w_stride = kLineTabStride;
        }
        if (res_k) launch_mixer_fused_res(q, c.stream);
        else if (ws) launch_mixer_fused_ws(q, c.stream);
        else if (split) launch_mixer_fused_h3(q, c.stream);
        else launch_mixer_fused(q, c.stream);
    };
    emit(std::move(r));
    return y;
}

TView Builder::linear(const std::string& prefix, const TView& x, int act, const TView* out, const TView* res) {
    return conv(prefix + ".weight", has_weight(prefix + ".bias") ? prefix + ".bias" : "", "", x, ConvGeom{}, act, out, res);
}

bool Builder::deconv_pair_to_prob(const std::string& w1n, const std::string& b1n, const std::string& bn1, const std::string& w2n,
                                  const std::string& b2n, const TView& x, const TView& out) {
    static const bool off = [] { const char* e = getenv("RD_DET_HEAD_FUSED"); return e && e[0] == '0'; }();
    const HostTensor& w1 = ws_->get(w1n);
    const HostTensor& w2 = ws_->get(w2n);
    if (off || w1.shape.size() != 4 || w2.shape.size() != 4 || w1.shape[2] != 2 || w1.shape[3] != 2 || w2.shape[2] != 2 || w2.shape[3] != 2)
        return false;
    const int cin = (int)w1.shape[0], cmid = (int)w1.shape[1], cout = (int)w2.shape[1];
    if (!det_head_tail_supported(cin, cmid, cout) || (int)w2.shape[0] != cmid || cin != x.c) return false;
    RD_CHECK(out.h == 4 * x.h && out.w == 4 * x.w && out.c == 1 && out.n == x.n, "det head tail: output view mismatch");
    const std::string key = w1n + "|" + bn1 + "|" + w2n + "|pair";
    if (!planning()) {
        if (!pb_->has(key + "#w1")) {
            std::vector<float> shift;
            std::vector<float> scale = bn_scale_shift(bn1, cmid, shift);
            std::vector<float> wa((size_t)4 * cmid * cin), ba(cmid, 0.f), wb((size_t)4 * cmid), bb(1, 0.f);
            const float* s1 = w1.f32();          // [cin][cmid][2][2]
            for (int ci = 0; ci < cin; ++ci)
                for (int co = 0; co < cmid; ++co)
                    for (int tap = 0; tap < 4; ++tap)
                        wa[((size_t)tap * cmid + co) * cin + ci] = s1[((size_t)ci * cmid + co) * 4 + tap] * scale[co];
            if (!b1n.empty()) {
                const float* bs = ws_->get(b1n).f32();
                for (int co = 0; co < cmid; ++co) ba[co] = bs[co] * scale[co];
            }
            for (int co = 0; co < cmid; ++co) ba[co] += shift[co];
            const float* s2 = w2.f32();          // [cmid][1][2][2]
            for (int co = 0; co < cmid; ++co)
                for (int tap = 0; tap < 4; ++tap) wb[(size_t)tap * cmid + co] = s2[(size_t)co * 4 + tap];
            if (!b2n.empty()) bb[0] = ws_->get(b2n).f32()[0];
            pb_->add(key + "#w1", wa);
            pb_->add(key + "#b1", ba);
            pb_->add(key + "#w2", wb);
            pb_->add(key + "#b2", bb);
        }
        return true;
    }
    const float* pw1 = pb_->ptr(key + "#w1");
    const float* pb1 = pb_->ptr(key + "#b1");
    const float* pw2 = pb_->ptr(key + "#w2");
    const float* pb2 = pb_->ptr(key + "#b2");
    OpRecord r;
    r.name = w1n + "+" + w2n;
    r.kind = "det_head_tail";
    r.cfg = "fused";
    const double m = (double)x.n * x.h * x.w;
    r.flops = 2.0 * m * (4.0 * cmid * cin + 16.0 * cmid);
    r.bytes = 4.0 * (m * cin + m * 16);
    const TView xv = x, yv = out;
    r.run = [xv, yv, pw1, pb1, pw2, pb2](const Plan& pl, const RunCtx& c) {
        launch_det_head_tail(pl.vptr(xv, c), pl.ld(xv), xv.n, xv.h, xv.w, pw1, pb1, pw2, pb2, pl.vptr(yv, c), c.stream);
    };
    emit(std::move(r));
    return true;
}

TView Builder::deconv2x2(const std::string& wname, const std::string& bname, const std::string& bn, const TView& x,
                         int act, const TView* out) {
    // nn.ConvTranspose2d(k=2, stride=2): weight [Cin, Cout, 2, 2]  (det_db_head.py:66-72,124-129)
    const HostTensor& w = ws_->get(wname);
    RD_CHECK(w.shape.size() == 4 && w.shape[2] == 2 && w.shape[3] == 2, "deconv2x2 weight shape: " + wname);
    const int cin = (int)w.shape[0], cout = (int)w.shape[1];
    RD_CHECK(cin == x.c && cin % 4 == 0, "deconv Cin mismatch: " + wname);
    TView y = out ? *out : alloc(x.n, 2 * x.h, 2 * x.w, cout);
    RD_CHECK(y.h == 2 * x.h && y.w == 2 * x.w && y.c == cout, "deconv output view mismatch");
    const std::string key = wname + "|" + bn + "|T";
    if (!planning()) {
        if (!pb_->has(key + "#w")) {
            std::vector<float> shift;
            std::vector<float> scale = bn_scale_shift(bn, cout, shift);
            std::vector<float> wf((size_t)4 * cout * cin);
            const float* src = w.f32();
            for (int ci = 0; ci < cin; ++ci)
                for (int co = 0; co < cout; ++co)
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx)
                            wf[((size_t)(dy * 2 + dx) * cout + co) * cin + ci] =
                                src[(((size_t)ci * cout + co) * 2 + dy) * 2 + dx] * scale[co];
            std::vector<float> bias(cout, 0.f);
            if (!bname.empty()) {
                const float* bs = ws_->get(bname).f32();
                for (int co = 0; co < cout; ++co) bias[co] = bs[co] * scale[co];
            }
            for (int co = 0; co < cout; ++co) bias[co] += shift[co];
            pb_->add(key + "#w", wf);
            pb_->add(key + "#b", bias);
            if (h3_ && fits_fp16_range(wf)) {
                std::vector<uint16_t> hi, lo;
                split_weights_h3(wf.data(), 4 * cout, cin, hi, lo);
                pb_->add_u16(key + "#wh", hi);
                pb_->add_u16(key + "#wl", lo);
            }
        }
        return y;
    }
    ConvParams p{};
    p.xld = plan_->ld(x);
    p.N = x.n; p.H = x.h; p.W = x.w; p.Cin = cin;
    p.w = pb_->ptr(key + "#w");
    p.bias = pb_->ptr(key + "#b");
    if (h3_ && pb_->has(key + "#wh")) {
        p.wh = reinterpret_cast<const uint16_t*>(pb_->ptr(key + "#wh"));
        p.wl = reinterpret_cast<const uint16_t*>(pb_->ptr(key + "#wl"));
        p.range_flag = range_flag_;
    }
    p.yld = plan_->ld(y);
    p.OH = x.h; p.OW = x.w; p.Cout = cout;
    p.KH = p.KW = p.SH = p.SW = 1;
    p.act = act;
    p.out_mode = OUT_DECONV2X2;
    p.M = x.n * x.h * x.w; p.K = cin; p.Ng = 4 * cout;
    OpRecord r;
    r.name = wname;
    r.kind = "deconv2x2";
    r.cfg = conv_igemm_config_name(p);
    r.flops = 2.0 * p.M * (double)cin * 4 * cout;
    r.bytes = 4.0 * ((double)p.M * cin + (double)p.M * 4 * cout);
    const TView xv = x, yv = y;
    const bool h3 = h3_ && p.wh != nullptr;
    r.run = [p, xv, yv, h3](const Plan& pl, const RunCtx& c) {
        ConvParams q = p;
        q.x = pl.vptr(xv, c);
        q.y = pl.vptr(yv, c);
        if (h3) launch_conv_igemm_h3(q, c.stream);
        else launch_conv_igemm(q, c.stream);
    };
    emit(std::move(r));
    return y;
}

TView Builder::stem3x3s2(const std::string& wname, const std::string& bn, const TView& xin, int act) {
    // xin: external NCHW image described as n, h, w with c = 3 (addressed as planes by the kernel)
    const HostTensor& w = ws_->get(wname);
    RD_CHECK(w.shape.size() == 4 && w.shape[1] == 3 && w.shape[2] == 3 && w.shape[3] == 3, "stem weight shape");
    const int cout = (int)w.shape[0];
    RD_CHECK(cout % 8 == 0, "stem Cout % 8");
    const int oh = out_dim(xin.h, 3, 2, 1, 1), ow = out_dim(xin.w, 3, 2, 1, 1);
    TView y = alloc(xin.n, oh, ow, cout);
    const std::string key = wname + "|" + bn + "|stem";
    if (!planning()) {
        if (!pb_->has(key + "#w")) {
            std::vector<float> shift;
            std::vector<float> scale = bn_scale_shift(bn, cout, shift);
            std::vector<float> wf((size_t)27 * cout);
            const float* src = w.f32();
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < 3; ++ci)
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b)
                            wf[(size_t)((a * 3 + b) * 3 + ci) * cout + co] = src[(((size_t)co * 3 + ci) * 3 + a) * 3 + b] * scale[co];
            pb_->add(key + "#w", wf);
            pb_->add(key + "#b", shift);
        }
        return y;
    }
    StemParams p{};
    p.N = xin.n; p.H = xin.h; p.W = xin.w; p.in_ch = xin.c;   // c == 1: grey image replicated to 3 channels
    p.w = pb_->ptr(key + "#w");
    p.bias = pb_->ptr(key + "#b");
    p.yld = plan_->ld(y);
    p.OH = oh; p.OW = ow; p.Cout = cout;
    p.act = act;
    OpRecord r;
    r.name = wname;
    r.kind = "stem3x3s2";
    r.flops = 2.0 * xin.n * oh * ow * 27.0 * cout;
    r.bytes = 4.0 * ((double)xin.n * xin.c * xin.h * xin.w + (double)xin.n * oh * ow * cout);
    const TView xv = xin, yv = y;
    r.run = [p, xv, yv](const Plan& pl, const RunCtx& c) {
        StemParams q = p;
        q.x = pl.vptr(xv, c);
        q.y = pl.vptr(yv, c);
        launch_stem_conv3x3s2(q, c.stream);
    };
    emit(std::move(r));
    return y;
}

TView Builder::stem_front(const std::string& w1, const std::string& bn1, const std::string& w2a, const std::string& bn2a,
                          const std::string& w2b, const std::string& bn2b, const TView& xin) {
    const int c1 = weight_dim(w1, 0);
    ConvGeom g2;            // padding='same' with an even kernel / F.pad(0,1,0,1): pad right + bottom only
    g2.kh = g2.kw = 2;
    g2.pb = g2.pr = 1;
    const std::string fkey = w1 + "|" + bn1 + "|stemfront";
    const bool off = std::getenv("RD_STEM_FUSED") && std::getenv("RD_STEM_FUSED")[0] == '0';   // A/B switch, read per plan
    const bool fused = planning() && !off && (h3_ || mixer_h3_) && pb_->has(fkey + "#img");
    if (!fused) {
        // the four separate kernels (fp32 precision mode, unsupported widths, and - in PREPARE mode - the weight folding of both forms)
        TView e = stem3x3s2(w1, bn1, xin, ACT_RELU);
        mask_cols(e, 1);                 // (line table only: beyond a line's own width e / a / cat are the next layer's zero padding)
        TView a = conv(w2a, "", bn2a, e, g2, ACT_RELU);
        mask_cols(a, 1);
        TView cat = alloc(e.n, e.h, e.w, 2 * c1);
        TView cat_pool = slice(cat, 0, c1), cat_b = slice(cat, c1, c1);
        maxpool2x2s1(e, cat_pool);
        conv(w2b, "", bn2b, a, g2, ACT_RELU, &cat_b);
        mask_cols(cat_b, 1);             // (the pool half is zero there already: e >= 0 is)
        release(e);
        release(a);
        if (!planning() && stem_fused_supported(c1) && !pb_->has(fkey + "#img")) {
            const std::string k1 = w1 + "|" + bn1 + "|stem", k2a = w2a + "|" + bn2a, k2b = w2b + "|" + bn2b;
            const float* f1 = pb_->host_ptr(k1 + "#w");
            const float* f2a = pb_->host_ptr(k2a + "#w");
            const float* f2b = pb_->host_ptr(k2b + "#w");
            const int na = c1 / 2;
            const bool fits = fits_fp16_range(std::vector<float>(f1, f1 + (size_t)27 * c1)) &&
                              fits_fp16_range(std::vector<float>(f2a, f2a + (size_t)na * 4 * c1)) &&
                              fits_fp16_range(std::vector<float>(f2b, f2b + (size_t)c1 * 4 * na));
            if (fits && weight_dim(w2a, 0) == na && weight_dim(w2a, 1) == c1 && weight_dim(w2b, 0) == c1 && weight_dim(w2b, 1) == na) {
                std::vector<uint16_t> img;
                prepare_stem_fused_weights(c1, f1, f2a, f2b, img);
                pb_->add_u16(fkey + "#img", img);
                std::vector<float> bias;
                const float* b1 = pb_->host_ptr(k1 + "#b");
                bias.insert(bias.end(), b1, b1 + c1);
                const float* b2a = pb_->host_ptr(k2a + "#b");
                bias.insert(bias.end(), b2a, b2a + na);
                const float* b2b = pb_->host_ptr(k2b + "#b");
                bias.insert(bias.end(), b2b, b2b + c1);
                pb_->add(fkey + "#bias", bias);
            }
        }
        return cat;
    }
    const int oh = out_dim(xin.h, 3, 2, 1, 1), ow = out_dim(xin.w, 3, 2, 1, 1);
    TView cat = alloc(xin.n, oh, ow, 2 * c1);
    const uint16_t* img = reinterpret_cast<const uint16_t*>(pb_->ptr(fkey + "#img"));
    const float* bias = pb_->ptr(fkey + "#bias");
    OpRecord r;
    r.name = w1 + ":front";
    r.kind = "stem_fused";
    r.cfg = "C" + std::to_string(c1);
    r.shape = "N" + std::to_string(xin.n) + "_" + std::to_string(oh) + "x" + std::to_string(ow);
    const int na = c1 / 2;
    r.flops = 2.0 * xin.n * oh * ow * (27.0 * c1 + 4.0 * c1 * na + 4.0 * na * c1);
    r.bytes = 4.0 * ((double)xin.n * xin.c * xin.h * xin.w + (double)xin.n * oh * ow * 2 * c1);
    const TView xv = xin, yv = cat;
    unsigned* flag = range_flag_;
    const bool has_lt = has_lt_;
    const TView ltv = lt_;
    r.run = [xv, yv, img, bias, c1, flag, has_lt, ltv](const Plan& pl, const RunCtx& c) {
        launch_stem_fused(c1, pl.vptr(xv, c), xv.n, xv.h, xv.w, xv.c, img, bias, pl.vptr(yv, c), pl.ld(yv), flag, c.stream,
                          has_lt ? reinterpret_cast<const int32_t*>(pl.vptr(ltv, c)) : nullptr);
    };
    emit(std::move(r));
    return cat;
}

TView Builder::stem_tail(const std::string& w3, const std::string& bn3, const std::string& w4, const std::string& bn4, const TView& x, int act3,
                         int act4, const TView* out) {
    const int n1 = weight_dim(w3, 0), cin = weight_dim(w3, 1), n2 = weight_dim(w4, 0);
    ConvGeom g3;
    g3.kh = g3.kw = 3;
    g3.sh = g3.sw = 2;
    g3.pt = g3.pl = g3.pb = g3.pr = 1;
    ConvGeom g1;
    const std::string fkey = w3 + "|" + bn3 + "|" + w4 + "|" + bn4 + "|stemtail";
    const int oh = out_dim(x.h, 3, 2, 1, 1), ow = out_dim(x.w, 3, 2, 1, 1);
    // (fused: 16-byte float4 accesses on both sides - channel strides and the output view's channel offset are multiples of four floats;
    //  an image of < 2^30 elements for the 32-bit buffer offsets of the patch loads)
    const bool fused = planning() && stem34_enabled() && (h3_ || mixer_h3_) && pb_->has(fkey + "#w3") && plan_->ld(x) % 4 == 0 &&
                       (size_t)x.h * x.w * plan_->ld(x) < ((size_t)1 << 30) && (!out || (plan_->ld(*out) % 4 == 0 && out->coff % 4 == 0)) &&
                       (size_t)oh * ow * (out ? plan_->ld(*out) : n2) < ((size_t)1 << 29);
    if (!fused) {
        TView s3 = conv(w3, "", bn3, x, g3, act3);
        TView s4 = conv(w4, "", bn4, s3, g1, act4, out);
        release(s3);
        if (!planning() && stem34_enabled() && !pb_->has(fkey + "#w3") && stem34_shape_ok(cin, n1, n2) && weight_dim(w3, 2) == 3 && weight_dim(w3, 3) == 3 &&
            weight_dim(w4, 1) == n1) {
            const std::string k3 = w3 + "|" + bn3, k4 = w4 + "|" + bn4;
            const float* f3 = pb_->host_ptr(k3 + "#w");
            const float* f4 = pb_->host_ptr(k4 + "#w");
            if (fits_fp16_range(std::vector<float>(f3, f3 + (size_t)n1 * 9 * cin)) && fits_fp16_range(std::vector<float>(f4, f4 + (size_t)n2 * n1))) {
                std::vector<uint16_t> img3, img4;
                float inv[2];
                prepare_stem34_weights(f3, f4, cin, n1, n2, img3, img4, inv);
                pb_->add_u16(fkey + "#w3", img3);
                pb_->add_u16(fkey + "#w4", img4);
                std::vector<float> b3((size_t)((n1 + 31) / 32) * 32, 0.f), b4((size_t)n2, 0.f);
                if (pb_->has(k3 + "#b")) std::copy(pb_->host_ptr(k3 + "#b"), pb_->host_ptr(k3 + "#b") + n1, b3.begin());
                if (pb_->has(k4 + "#b")) std::copy(pb_->host_ptr(k4 + "#b"), pb_->host_ptr(k4 + "#b") + n2, b4.begin());
                pb_->add(fkey + "#b3", b3);
                pb_->add(fkey + "#b4", b4);
                pb_->add(fkey + "#inv", std::vector<float>{inv[0], inv[1]});
            }
        }
        return s4;
    }
    TView y = out ? *out : alloc(x.n, oh, ow, n2);
    RD_CHECK(y.n == x.n && y.h == oh && y.w == ow && y.c == n2, "stem_tail output view mismatch: " + w4);
    Stem34Params p{};
    p.xld = plan_->ld(x);
    p.N = x.n; p.H = x.h; p.W = x.w; p.Cin = cin;
    p.yld = plan_->ld(y);
    p.OH = oh; p.OW = ow; p.N1 = n1; p.N2 = n2;
    p.w3 = reinterpret_cast<const uint16_t*>(pb_->ptr(fkey + "#w3"));
    p.w4 = reinterpret_cast<const uint16_t*>(pb_->ptr(fkey + "#w4"));
    p.b3 = pb_->ptr(fkey + "#b3");
    p.b4 = pb_->ptr(fkey + "#b4");
    p.w3_inv = pb_->host_ptr(fkey + "#inv")[0];
    p.w4_inv = pb_->host_ptr(fkey + "#inv")[1];
    p.act3 = act3; p.act4 = act4;
    p.range_flag = range_flag_;
    OpRecord r;
    r.name = w3 + ":tail";
    r.kind = "stem_tail";
    r.cfg = "C" + std::to_string(cin) + "_" + std::to_string(n1) + "_" + std::to_string(n2);
    r.shape = "M" + std::to_string(x.n * oh * ow) + "_K" + std::to_string(9 * cin) + "_N" + std::to_string(n1) + "_N" + std::to_string(n2);
    r.flops = 2.0 * x.n * oh * ow * (9.0 * cin * n1 + (double)n1 * n2);
    r.bytes = 4.0 * ((double)x.pixels() * cin + (double)x.n * oh * ow * n2 + 9.0 * cin * n1 + (double)n1 * n2);
    const TView xv = x, yv = y;
    r.run = [p, xv, yv](const Plan& pl, const RunCtx& c) mutable {
        Stem34Params q = p;
        q.x = pl.vptr(xv, c);
        q.y = pl.vptr(yv, c);
        launch_stem34(q, c.stream);
    };
    emit(std::move(r));
    return y;
}

void Builder::mask_cols(const TView& v, int col) {
    if (!has_lt_ || !planning()) return;
    OpRecord r;
    r.name = "mask_cols";
    r.kind = "mask";
    r.bytes = 4.0 * v.pixels() * v.c;
    const TView yv = v, ltv = lt_;
    r.run = [yv, ltv, col](const Plan& pl, const RunCtx& c) {
        launch_mask_cols(pl.vptr(yv, c), pl.ld(yv), yv.n, yv.h, yv.w, yv.c, reinterpret_cast<const int32_t*>(pl.vptr(ltv, c)) + col,
                         kLineTabStride, c.stream);
    };
    emit(std::move(r));
}

TView Builder::dwconv(const std::string& wname, const std::string& bname, const std::string& bn, const TView& x,
                      const ConvGeom& g, int act, const TView* out, const TView* res, GapOut* gap, const TView* tokinfo) {
    const HostTensor& w = ws_->get(wname);
    RD_CHECK(w.shape.size() == 4 && w.shape[1] == 1, "depthwise weight shape: " + wname);
    const int c = (int)w.shape[0], kh = (int)w.shape[2], kw = (int)w.shape[3];
    RD_CHECK(c == x.c && c % 4 == 0, "depthwise channel mismatch: " + wname);
    RD_CHECK(kh == g.kh && kw == g.kw, "depthwise kernel mismatch: " + wname);
    const int oh = out_dim(x.h, kh, g.sh, g.pt, g.pb), ow = out_dim(x.w, kw, g.sw, g.pl, g.pr);
    TView y = out ? *out : alloc(x.n, oh, ow, c);
    RD_CHECK(y.h == oh && y.w == ow && y.c == c, "depthwise output view mismatch: " + wname);
    const std::string key = wname + "|" + bn + "|dw";
    if (gap) {
        DwParams gp{};
        gp.N = x.n; gp.H = x.h; gp.W = x.w; gp.C = c; gp.OH = oh; gp.OW = ow;
        gp.KH = kh; gp.KW = kw; gp.SH = g.sh; gp.SW = g.sw; gp.PT = g.pt; gp.PL = g.pl;
        gap->chunks = dwconv_gap_chunks(gp);
        if (gap->chunks > 0) gap->partial = alloc_raw((size_t)x.n * gap->chunks * c);
    }
    if (!planning()) {
        if (!pb_->has(key + "#w")) {
            std::vector<float> shift;
            std::vector<float> scale = bn_scale_shift(bn, c, shift);
            std::vector<float> wf((size_t)kh * kw * c);
            const float* src = w.f32();
            for (int ch = 0; ch < c; ++ch)
                for (int t = 0; t < kh * kw; ++t) wf[(size_t)t * c + ch] = src[(size_t)ch * kh * kw + t] * scale[ch];
            std::vector<float> bias(c, 0.f);
            if (!bname.empty()) {
                const float* bs = ws_->get(bname).f32();
                for (int ch = 0; ch < c; ++ch) bias[ch] = bs[ch] * scale[ch];
            }
            for (int ch = 0; ch < c; ++ch) bias[ch] += shift[ch];
            pb_->add(key + "#w", wf);
            pb_->add(key + "#b", bias);
        }
        return y;
    }
    DwParams p{};
    p.xld = plan_->ld(x);
    p.N = x.n; p.H = x.h; p.W = x.w; p.C = c;
    p.w = pb_->ptr(key + "#w");
    p.bias = pb_->ptr(key + "#b");
    p.yld = plan_->ld(y);
    p.OH = oh; p.OW = ow; p.KH = kh; p.KW = kw; p.SH = g.sh; p.SW = g.sw; p.PT = g.pt; p.PL = g.pl;
    p.act = act;
    p.rld = res ? plan_->ld(*res) : 0;
    OpRecord r;
    r.name = wname;
    r.kind = "dwconv" + std::to_string(kh) + "x" + std::to_string(kw);
    r.flops = 2.0 * x.n * oh * ow * (double)c * kh * kw;
    r.bytes = 4.0 * ((double)x.pixels() * c + (double)x.n * oh * ow * c * (res ? 2 : 1));
    const TView xv = x, yv = y;
    const bool has_res = res != nullptr;
    const TView rv = res ? *res : TView{};
    const bool has_gap = gap && gap->chunks > 0;
    const TView gpv = has_gap ? gap->partial : TView{};
    const bool ragged = tokinfo != nullptr;
    RD_CHECK(!ragged || (kh == 1 && g.sw == 1 && x.n == 1 && x.h == 1 && x.w < (1 << 30)), "ragged depthwise conv: 1 x k over one token row");
    const TView tiv = ragged ? *tokinfo : TView{};
    const bool has_lt = has_lt_ && !ragged;
    RD_CHECK(!has_lt || (g.sw == 1 && ow == x.w), "line table: depthwise convs keep the width (stride (s, 1), 'same' padding)");
    const TView ltv = lt_;
    r.run = [p, xv, yv, rv, has_res, has_gap, gpv, ragged, tiv, has_lt, ltv](const Plan& pl, const RunCtx& cx) {
        DwParams q = p;
        q.x = pl.vptr(xv, cx);
        q.y = pl.vptr(yv, cx);
        q.res = has_res ? pl.vptr(rv, cx) : nullptr;
        q.gap_partial = has_gap ? pl.vptr(gpv, cx) : nullptr;
        q.tokinfo = ragged ? reinterpret_cast<const int32_t*>(pl.vptr(tiv, cx)) : nullptr;
        if (has_lt) {
            q.line_w = reinterpret_cast<const int32_t*>(pl.vptr(ltv, cx)) + 2;
            q.line_w_stride = kLineTabStride;
        }
        launch_dwconv(q, cx.stream);
    };
    emit(std::move(r));
    return y;
}

void Builder::maxpool2x2s1(const TView& x, const TView& out) {
    RD_CHECK(out.h == x.h && out.w == x.w && out.c == x.c && x.c % 4 == 0, "maxpool view mismatch");
    if (!planning()) return;
    OpRecord r;
    r.name = "maxpool2x2s1";
    r.kind = "pool";
    r.bytes = 8.0 * x.pixels() * x.c;
    const TView xv = x, yv = out;
    r.run = [xv, yv](const Plan& pl, const RunCtx& c) {
        launch_maxpool2x2s1(pl.vptr(xv, c), pl.ld(xv), pl.vptr(yv, c), pl.ld(yv), xv.n, xv.h, xv.w, xv.c, c.stream);
    };
    emit(std::move(r));
}

TView Builder::avgpool3x2(const TView& x, const TView* out) {
    RD_CHECK(x.h >= 3 && x.w >= 2, "avg_pool2d([3,2]): feature map too small");  // rec_lcnetv4.py:309-310
    const int oh = (x.h - 3) / 3 + 1, ow = (x.w - 2) / 2 + 1;
    TView y = out ? *out : alloc(x.n, oh, ow, x.c);
    RD_CHECK(y.n == x.n && y.h == oh && y.w == ow && y.c == x.c, "avgpool3x2: output view mismatch");
    if (!planning()) return y;
    OpRecord r;
    r.name = "avgpool3x2";
    r.kind = "pool";
    r.bytes = 4.0 * (x.pixels() * x.c + y.pixels() * y.c);
    const TView xv = x, yv = y;
    const bool has_lt = has_lt_;
    RD_CHECK(!has_lt || oh == 1, "line table: the pooled map is one row of tokens per line");
    const TView ltv = lt_;
    r.run = [xv, yv, has_lt, ltv](const Plan& pl, const RunCtx& c) {
        launch_avgpool3x2(pl.vptr(xv, c), pl.ld(xv), pl.vptr(yv, c), pl.ld(yv), xv.n, xv.h, xv.w, xv.c, c.stream,
                          has_lt ? reinterpret_cast<const int32_t*>(pl.vptr(ltv, c)) : nullptr);
    };
    emit(std::move(r));
    return y;
}

TView Builder::se_gate(const std::string& w1n, const std::string& b1n, const std::string& w2n, const std::string& b2n,
                       const TView& x, int gate_act, const GapOut* pre) {
    const HostTensor& w1 = ws_->get(w1n);
    const int cr = (int)w1.shape[0], c = (int)w1.shape[1];
    RD_CHECK(c == x.c && c % 4 == 0, "SE channel mismatch: " + w1n);
    RD_CHECK(c <= 512, "SE width above 512 channels (se_fc_kernel reads a row of W1 with two 16-byte loads per lane): " + w1n);
    const int hw = x.h * x.w;
    const bool fused_gap = pre && pre->chunks > 0;  // the producing depthwise conv already wrote the partial sums
    RD_CHECK(!has_lt_ || fused_gap, "line table: the SE pooling sums must come from the depthwise kernel (it leaves out the columns beyond a line)");
    const int chunks = fused_gap ? pre->chunks : std::max(1, std::min(64, hw / 256));
    TView partial = fused_gap ? pre->partial : alloc_raw((size_t)x.n * chunks * c);
    TView gate = alloc(x.n, 1, 1, c);
    if (!planning()) {
        if (!pb_->has(w1n)) {
            const HostTensor& w2 = ws_->get(w2n);
            pb_->add(w1n, std::vector<float>(w1.f32(), w1.f32() + w1.numel()));
            pb_->add(b1n, std::vector<float>(ws_->get(b1n).f32(), ws_->get(b1n).f32() + cr));
            pb_->add(w2n, std::vector<float>(w2.f32(), w2.f32() + w2.numel()));
            pb_->add(b2n, std::vector<float>(ws_->get(b2n).f32(), ws_->get(b2n).f32() + c));
        }
        release(partial);
        return gate;
    }
    if (!fused_gap) {
        OpRecord r;
        r.name = w1n + ":gap";
        r.kind = "gap";
        r.bytes = 4.0 * x.pixels() * c;
        const TView xv = x, pv = partial;
        r.run = [xv, pv, hw, c, chunks](const Plan& pl, const RunCtx& cx) {
            launch_gap_partial(pl.vptr(xv, cx), pl.ld(xv), xv.n, hw, c, pl.vptr(pv, cx), chunks, cx.stream);
        };
        emit(std::move(r));
    }
    {
        SeFcParams p{};
        p.chunks = chunks; p.N = x.n; p.C = c; p.Cr = cr; p.inv_hw = 1.f / (float)hw;
        p.w1 = pb_->ptr(w1n); p.b1 = pb_->ptr(b1n); p.w2 = pb_->ptr(w2n); p.b2 = pb_->ptr(b2n);
        p.gate = gate_act;
        OpRecord r;
        r.name = w1n + ":fc";
        r.kind = "se_fc";
        r.flops = 4.0 * x.n * c * cr;
        const TView pv = partial, gv = gate;
        const bool has_lt = has_lt_;
        const TView ltv = lt_;
        p.H = x.h;
        r.run = [p, pv, gv, has_lt, ltv](const Plan& pl, const RunCtx& cx) {
            SeFcParams q = p;
            q.partial = pl.vptr(pv, cx);
            q.scale = pl.vptr(gv, cx);
            if (has_lt) {
                q.line_w = reinterpret_cast<const int32_t*>(pl.vptr(ltv, cx)) + 2;
                q.line_w_stride = kLineTabStride;
            }
            launch_se_fc(q, cx.stream);
        };
        emit(std::move(r));
    }
    release(partial);
    return gate;
}

void Builder::scale(const TView& x, const TView& gate, float alpha, const TView& out) {
    RD_CHECK(out.h == x.h && out.w == x.w && out.c == x.c && gate.c == x.c, "scale view mismatch");
    if (!planning()) return;
    OpRecord r;
    r.name = "se_scale";
    r.kind = "scale";
    r.bytes = 8.0 * x.pixels() * x.c;
    const TView xv = x, gv = gate, yv = out;
    r.run = [xv, gv, yv, alpha](const Plan& pl, const RunCtx& c) {
        launch_scale_channels(pl.vptr(xv, c), pl.ld(xv), pl.vptr(yv, c), pl.ld(yv), pl.vptr(gv, c), alpha, xv.n,
                              xv.h * xv.w, xv.c, c.stream);
    };
    emit(std::move(r));
}

void Builder::upsample(const TView& x, const TView& out, int f, bool accumulate) {
    RD_CHECK(out.h == x.h * f && out.w == x.w * f && out.c == x.c && out.n == x.n, "upsample view mismatch");
    if (!planning()) return;
    OpRecord r;
    r.name = accumulate ? "upsample_add" : "upsample_copy";
    r.kind = "upsample";
    r.bytes = 4.0 * (x.pixels() * x.c + out.pixels() * out.c * (accumulate ? 2 : 1));
    const TView xv = x, yv = out;
    r.run = [xv, yv, f, accumulate](const Plan& pl, const RunCtx& c) {
        launch_upsample(pl.vptr(xv, c), pl.ld(xv), pl.vptr(yv, c), pl.ld(yv), yv.n, yv.h, yv.w, yv.c, f,
                        accumulate ? 1 : 0, c.stream);
    };
    emit(std::move(r));
}

TView Builder::layernorm(const std::string& prefix, const TView& x, float eps) {
    RD_CHECK(x.c <= 512, "layernorm: C <= 512");
    TView y = alloc(x.n, x.h, x.w, x.c);
    if (!planning()) {
        if (!pb_->has(prefix + ".weight")) {
            const HostTensor& g = ws_->get(prefix + ".weight");
            const HostTensor& b = ws_->get(prefix + ".bias");
            RD_CHECK((int)g.numel() == x.c, "layernorm width: " + prefix);
            pb_->add(prefix + ".weight", std::vector<float>(g.f32(), g.f32() + g.numel()));
            pb_->add(prefix + ".bias", std::vector<float>(b.f32(), b.f32() + b.numel()));
        }
        return y;
    }
    const float* g = pb_->ptr(prefix + ".weight");
    const float* b = pb_->ptr(prefix + ".bias");
    OpRecord r;
    r.name = prefix;
    r.kind = "layernorm";
    r.bytes = 8.0 * x.pixels() * x.c;
    const TView xv = x, yv = y;
    r.run = [xv, yv, g, b, eps](const Plan& pl, const RunCtx& c) {
        launch_layernorm(pl.vptr(xv, c), pl.ld(xv), pl.vptr(yv, c), pl.ld(yv), g, b, (int)xv.pixels(), xv.c, eps, c.stream);
    };
    emit(std::move(r));
    return y;
}

TView Builder::attention(const TView& qkv, int B, int T, int heads, int hd, const TView* seg) {
    RD_CHECK(qkv.c == 3 * heads * hd && qkv.coff == 0 && plan_->ld(qkv) == qkv.c, "attention: packed qkv expected");
    RD_CHECK(hd == 15 || hd == 16 || hd == 32, "attention: head_dim 15/16/32");
    // (any T: sequences longer than the LDS holds run over key tiles, kernels_misc.hip attention_kernel)
    TView o = alloc(qkv.n, qkv.h, qkv.w, heads * hd);
    if (!planning()) return o;
    const float sc = 1.0f / std::sqrt((float)hd);
    OpRecord r;
    r.name = "self_attn";
    r.kind = "attention";
    r.flops = 4.0 * B * heads * (double)T * T * hd;
    const TView qv = qkv, ov = o;
    const bool ragged = seg != nullptr;
    const TView sv = ragged ? *seg : TView{};
    r.run = [qv, ov, B, T, heads, hd, sc, ragged, sv](const Plan& pl, const RunCtx& c) {
        launch_attention(pl.vptr(qv, c), pl.vptr(ov, c), B, T, heads, hd, sc, c.stream,
                         ragged ? reinterpret_cast<const int32_t*>(pl.vptr(sv, c)) : nullptr);
    };
    emit(std::move(r));
    return o;
}

TView Builder::add(const TView& a, const TView& b) {
    RD_CHECK(a.pixels() == b.pixels() && a.c == b.c, "add: shape mismatch");
    TView y = alloc(a.n, a.h, a.w, a.c);
    if (!planning()) return y;
    OpRecord r;
    r.name = "add";
    r.kind = "add";
    r.bytes = 12.0 * a.pixels() * a.c;
    const TView av = a, bv = b, yv = y;
    r.run = [av, bv, yv](const Plan& pl, const RunCtx& c) {
        launch_add(pl.vptr(av, c), pl.ld(av), pl.vptr(bv, c), pl.ld(bv), pl.vptr(yv, c), pl.ld(yv), (int)av.pixels(), av.c, c.stream);
    };
    emit(std::move(r));
    return y;
}

void Builder::to_nchw(const TView& x, const TView& out_ext) {
    if (!planning()) return;
    OpRecord r;
    r.name = "to_nchw";
    r.kind = "layout";
    r.bytes = 8.0 * x.pixels() * x.c;
    const TView xv = x, yv = out_ext;
    r.run = [xv, yv](const Plan& pl, const RunCtx& c) {
        launch_nhwc_to_nchw(pl.vptr(xv, c), pl.ld(xv), pl.vptr(yv, c), xv.n, xv.h, xv.w, xv.c, c.stream);
    };
    emit(std::move(r));
}

void Builder::copy(const TView& x, const TView& out) {
    RD_CHECK(x.pixels() == out.pixels() && x.c == out.c && x.c % 4 == 0, "copy: shape mismatch");
    if (!planning()) return;
    OpRecord r;
    r.name = "copy";
    r.kind = "scale";
    r.bytes = 8.0 * x.pixels() * x.c;
    const TView xv = x, yv = out;
    r.run = [xv, yv](const Plan& pl, const RunCtx& c) {
        launch_upsample(pl.vptr(xv, c), pl.ld(xv), pl.vptr(yv, c), pl.ld(yv), yv.n, yv.h, yv.w, yv.c, 1, 0, c.stream);
    };
    emit(std::move(r));
}

void Builder::ctc_stats(const TView& logits, const TView& idx_ext, const TView& prob_ext) {
    if (!planning()) return;
    OpRecord r;
    r.name = "ctc_rowstats";
    r.kind = "ctc_stats";
    r.bytes = 8.0 * logits.pixels() * logits.c;
    const TView lv = logits, iv = idx_ext, pv = prob_ext;
    r.run = [lv, iv, pv](const Plan& pl, const RunCtx& c) {
        launch_rowmax_softmax(pl.vptr(lv, c), pl.ld(lv), (int)lv.pixels(), lv.c, (int32_t*)pl.vptr(iv, c), pl.vptr(pv, c), c.stream);
    };
    emit(std::move(r));
}

void Builder::softmax_rows(const TView& logits, const TView& out_ext, const TView* idx_ext, const TView* prob_ext) {
    if (!planning()) return;
    OpRecord r;
    r.name = "softmax";
    r.kind = "softmax";
    r.bytes = 12.0 * logits.pixels() * logits.c;
    const TView lv = logits, ov = out_ext;
    const bool stats = idx_ext != nullptr && prob_ext != nullptr;
    const TView iv = stats ? *idx_ext : TView{}, pv = stats ? *prob_ext : TView{};
    r.run = [lv, ov, stats, iv, pv](const Plan& pl, const RunCtx& c) {
        launch_row_softmax(pl.vptr(lv, c), pl.ld(lv), pl.vptr(ov, c), (int)lv.pixels(), lv.c, c.stream,
                           stats ? (int32_t*)pl.vptr(iv, c) : nullptr, stats ? pl.vptr(pv, c) : nullptr);
    };
    emit(std::move(r));
}

void Builder::ctc_head(const std::string& prefix, const TView& x, const TView& idx_ext, const TView& prob_ext) {
    const HostTensor& w = ws_->get(prefix + ".weight");
    const int C = (int)w.shape[0], K = (int)w.shape[1];
    RD_CHECK(K == x.c && K % 4 == 0, "ctc head width");
    const int M = (int)x.pixels();
    // the two kernels tile differently, so their class-split counts differ: the workspace is sized for either
    const int ns_f32 = ctc_head_nsplit(M, C, false), ns_h3 = ctc_head_nsplit(M, C, true);
    TView part = alloc_raw((size_t)M * std::max(ns_f32, ns_h3) * 4);
    if (!planning()) {
        if (!pb_->has(prefix + "|ctc#w")) {
            // W' [C][128]: columns 0..K-1 = W, column K = bias (the kernel feeds X[:,K] = 1), rest 0
            RD_CHECK(K < 128, "ctc head: K < 128");
            const float* bs = ws_->get(prefix + ".bias").f32();
            std::vector<float> wp((size_t)C * 128, 0.f);
            for (int c = 0; c < C; ++c) {
                std::copy(w.f32() + (size_t)c * K, w.f32() + (size_t)(c + 1) * K, wp.begin() + (size_t)c * 128);
                wp[(size_t)c * 128 + K] = bs[c];
            }
            pb_->add(prefix + "|ctc#w", wp);
            if (fits_fp16_range(wp)) {   // split-fp16 copy (K is already padded to 128)
                std::vector<uint16_t> hi, lo;
                split_weights_h3(wp.data(), C, 128, hi, lo);
                pb_->add_u16(prefix + "|ctc#wh", hi);
                pb_->add_u16(prefix + "|ctc#wl", lo);
            }
        }
        release(part);
        return;
    }
    CtcParams p{};
    p.xld = plan_->ld(x);
    p.w = pb_->ptr(prefix + "|ctc#w");
    p.bias = nullptr;
    p.M = M; p.K = K; p.C = C;
    const bool split = (h3_ || mixer_h3_) && pb_->has(prefix + "|ctc#wh");
    p.nsplit = split ? ns_h3 : ns_f32;
    if (split) {
        p.wh = reinterpret_cast<const uint16_t*>(pb_->ptr(prefix + "|ctc#wh"));
        p.wl = reinterpret_cast<const uint16_t*>(pb_->ptr(prefix + "|ctc#wl"));
        p.range_flag = range_flag_;
    }
    OpRecord r;
    r.name = prefix;
    r.kind = split ? "ctc_head_fused_h3" : "ctc_head_fused";
    r.flops = 2.0 * M * (double)K * C;
    r.bytes = 4.0 * ((double)M * K + (double)C * K);
    const TView xv = x, pv = part, iv = idx_ext, prv = prob_ext;
    r.run = [p, xv, pv, iv, prv](const Plan& pl, const RunCtx& c) {
        CtcParams q = p;
        q.x = pl.vptr(xv, c);
        q.part = pl.vptr(pv, c);
        q.idx = (int32_t*)pl.vptr(iv, c);
        q.prob = pl.vptr(prv, c);
        launch_ctc_head(q, c.stream);
    };
    emit(std::move(r));
    release(part);
}

Plan::~Plan() {
    // a plan is dropped only by the LRU of Engine::plan_for (rare): wait for the stream an exec was last launched on before
    // freeing it - its kernel arguments live in the exec
    for (auto& g : graphs)
        if (g.exec) {
            (void)hipStreamSynchronize(g.stream);
            (void)hipGraphExecDestroy(g.exec);
        }
    (void)hipGetLastError();
}

// =================================================================================================
// Engine
// =================================================================================================
extern bool g_disable_fused_mixer;
Engine::Engine(int device, const std::string& kind) : device_(device), kind_(kind) {
    if (const char* e = getenv("RD_DISABLE_FUSED_MIXER")) g_disable_fused_mixer = e[0] == '1';
    if (const char* e = getenv("RD_GRAPHS")) graphs_ = e[0] != '0';
    if (const char* e = getenv("RD_PRECISION")) {
        const std::string v(e);
        RD_CHECK(v == "auto" || v == "fp32" || v == "h3", "RD_PRECISION must be auto, fp32 or h3");
        precision_ = v == "h3" ? PREC_H3 : v == "fp32" ? PREC_FP32 : PREC_AUTO;
    }
    RD_CHECK(kind == "ppocrv6_det" || kind == "ppocrv6_rec" || kind == "pphgnetv2_b4" || kind == "pphgnetv2_b6_formula",
             "unknown model kind '" + kind + "'");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) throw Error("no HIP device available (MI355X required; there is no CPU fallback)");
    RD_CHECK(device >= 0 && device < count, "device id out of range");
}

void Engine::retire_graph(hipGraphExec_t exec, hipStream_t s) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, s) == hipSuccess) {
        retired_graphs_.push_back({exec, ev});
        return;
    }
    if (ev) (void)hipEventDestroy(ev);
    (void)hipGetLastError();
    (void)hipStreamSynchronize(s);            // no event: wait, then free
    (void)hipGraphExecDestroy(exec);
}

void Engine::sweep_retired_graphs(bool wait) {
    size_t keep = 0;
    for (auto& r : retired_graphs_) {
        if (wait) (void)hipEventSynchronize(r.done);
        if (wait || hipEventQuery(r.done) == hipSuccess) {
            (void)hipGraphExecDestroy(r.exec);
            (void)hipEventDestroy(r.done);
        } else {
            retired_graphs_[keep++] = r;
        }
    }
    retired_graphs_.resize(keep);
    (void)hipGetLastError();                  // hipErrorNotReady of the queries
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    sweep_retired_graphs(true);
    for (auto ev : events_) (void)hipEventDestroy(ev);
    if (arena_) (void)hipFree(arena_);
    if (range_flag_host_) (void)hipHostFree(range_flag_host_);
}

void Engine::build(Builder& b, int B, int H, int W, int flags) {
    if (kind_ == "ppocrv6_det") build_ppocrv6_det(b, B, H, W);
    else if (kind_ == "ppocrv6_rec") build_ppocrv6_rec(b, B, H, W, flags);
    else if (kind_ == "pphgnetv2_b6_formula") build_pphgnetv2_b6_formula(b, B, H, W, flags);
    else build_pphgnetv2_b4(b, B, H, W);
}

void Engine::load_weights(const void* blob, size_t nbytes) {
    RD_HIP(hipSetDevice(device_));
    RD_CHECK(!loaded_, "weights already loaded for this handle");
    store_.load_safetensors(blob, nbytes);
    if (kind_ == "ppocrv6_rec") {
        n_classes_ = (int)store_.get("head.head.weight").shape[0];  // torch.py:112-116
        rec_token_dim_ = (int)store_.get("head.encoder.conv_block.0.convolution.weight").shape[1];
    }
    Plan dummy;
    h3_prepared_ = precision_ == PREC_H3;
    // the range flag: one word of pinned host memory mapped into the device's address space (rd_device.h rd_raise_flag) - reading
    // it costs a stream synchronise and a host load, not a device-to-host copy that queues behind whatever shares its hardware queue
    RD_HIP(hipHostMalloc((void**)&range_flag_host_, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *range_flag_host_ = 0u;
    RD_HIP(hipHostGetDevicePointer((void**)&range_flag_, (void*)range_flag_host_, 0));
    Builder b(Mode::PREPARE, &store_, &params_, &dummy, h3_prepared_, true);
    // smallest legal geometry; only weight names/shapes matter in PREPARE mode
    if (kind_ == "ppocrv6_rec") build(b, 1, 48, 64, 0), build(b, 1, 48, 64, REC_UNFUSED_CTC);
    else build(b, 1, 64, 64, 0);
    params_.upload();
    loaded_ = true;
}

void Engine::set_precision(int p) {
    RD_CHECK(p == PREC_AUTO || p == PREC_FP32 || p == PREC_H3, "unknown precision mode");
    RD_CHECK(p != PREC_H3 || h3_prepared_ || !loaded_, "precision h3 needs RD_PRECISION=h3 when the weights are loaded");
    precision_ = p;
}

int Engine::take_range_flag(hipStream_t s) {
    if (!range_flag_host_) return 0;
    RD_HIP(hipSetDevice(device_));
    RD_HIP(hipStreamSynchronize(s));            // the forwards to be judged were launched on s, or s waits on their events
    const unsigned v = __atomic_load_n(range_flag_host_, __ATOMIC_ACQUIRE);
    if (v) __atomic_store_n(range_flag_host_, 0u, __ATOMIC_RELEASE);
    return v ? 1 : 0;
}

const Plan& Engine::plan_for(int B, int H, int W, int flags) {
    RD_CHECK(loaded_, "weights not loaded");
    auto key = std::make_tuple(B, H, W, flags | (precision_ << 24));
    auto it = plans_.find(key);
    if (it != plans_.end()) {
        it->second->last_use = ++plan_clock_;
        return *it->second;
    }
    // LRU: one cold shape (a long line, an odd page size) must not throw away the hot plans of this handle
    while (plans_.size() >= kMaxPlans) {
        auto victim = plans_.begin();
        for (auto j = plans_.begin(); j != plans_.end(); ++j)
            if (j->second->last_use < victim->second->last_use) victim = j;
        plans_.erase(victim);
    }
    ++plans_built_;
    auto plan = std::make_unique<Plan>();
    Builder b(Mode::PLAN, &store_, &params_, plan.get(), precision_ == PREC_H3, precision_ == PREC_AUTO, range_flag_);
    build(b, B, H, W, flags);
    plan->arena_bytes = (plan->arena_bytes + 255) / 256 * 256;
    plan->last_use = ++plan_clock_;
    auto& ref = *plan;
    plans_[key] = std::move(plan);
    return ref;
}

void Engine::run(int B, int H, int W, int flags, const std::vector<void*>& ext, void* ws, size_t ws_bytes, hipStream_t s) {
    RD_HIP(hipSetDevice(device_));
    const Plan& plan = plan_for(B, H, W, flags);
    RunCtx ctx;
    ctx.ext = ext;
    ctx.stream = s;
    if (ws) {
        RD_CHECK(ws_bytes >= plan.arena_bytes, "workspace too small");
        ctx.arena = (uint8_t*)ws;
    } else {
        if (arena_bytes_ < plan.arena_bytes) {
            RD_HIP(hipStreamSynchronize(s));
            if (arena_) RD_HIP(hipFree(arena_));
            arena_ = nullptr;
            arena_bytes_ = 0;
            RD_HIP(hipMalloc((void**)&arena_, plan.arena_bytes));
            arena_bytes_ = plan.arena_bytes;
            // developer check (RD_POISON_ARENA=1): fresh workspace filled with NaN bit patterns - no kernel may let workspace bytes
            // it did not write reach a result or the range guard
            static const bool poison = std::getenv("RD_POISON_ARENA") && std::getenv("RD_POISON_ARENA")[0] == '1';
            if (poison) {     // on the launch stream: a null-stream memset is not ordered against a non-blocking stream
                RD_HIP(hipMemsetAsync(arena_, 0xFF, arena_bytes_, s));
                RD_HIP(hipStreamSynchronize(s));
            }
        }
        ctx.arena = arena_;
    }
    for (const Buf& b : plan.bufs)
        if (b.external >= 0) RD_CHECK(b.external < (int)ext.size() && ext[b.external], "missing external buffer");
    if (!profiling_) {
        // developer trace (RD_RANGE_TRACE=1): which op raises the split-fp16 range flag (synchronises after every op)
        static const bool range_trace = std::getenv("RD_RANGE_TRACE") && std::getenv("RD_RANGE_TRACE")[0] == '1';
        if (range_trace && range_flag_) {
            for (const OpRecord& op : plan.ops) {
                op.run(plan, ctx);
                RD_HIP(hipStreamSynchronize(s));
                unsigned v = __atomic_load_n(range_flag_host_, __ATOMIC_ACQUIRE);
                if (v) {
                    fprintf(stderr, "[range] %s: op '%s' kind %s cfg %s shape %s raised the flag (B=%d H=%d W=%d flags=%d)\n", kind_.c_str(),
                            op.name.c_str(), op.kind.c_str(), op.cfg.c_str(), op.shape.c_str(), B, H, W, flags);
                    break;                       // (the flag stays raised for the caller)
                }
            }
            return;
        }
        // developer what-if (timing only, results are garbage): RD_SKIP_KIND=<op kind> leaves those launches out
        static const char* skip_kind = std::getenv("RD_SKIP_KIND");
        if (skip_kind) {
            for (const OpRecord& op : plan.ops)
                if (op.kind != skip_kind) op.run(plan, ctx);
            RD_HIP(hipGetLastError());
            return;
        }
        // hipGraph replay.  A forward is 30-90 dependent launches whose arguments are fixed by (plan, external pointers, workspace):
        // the second time the same triple shows up on a stream the launches are captured into a graph, from the third on the
        // host issues ONE hipGraphLaunch (a page batch is ~900 launches: 6-10 ms of a single host thread per 95-ms step, and a
        // slow host thread was measured to cost 10 % of the throughput).  Callers keep their buffers stable (PagePipeline).
        if (graphs_ && !plan.graph_broken && plan.ops.size() >= 4) {
            GraphSlot* slot = nullptr;
            for (auto& g : plan.graphs)
                if (g.arena == ctx.arena && g.stream == s && g.ext == ext) { slot = &g; break; }
            if (slot && slot->exec) {
                slot->last_use = ++plan_clock_;
                plan.graph_evictions = 0;     // replays do happen for this plan
                ++graph_replays_;
                RD_HIP(hipGraphLaunch(slot->exec, s));
                return;
            }
            if (slot) {                       // seen once before: worth a capture
                hipGraph_t graph = nullptr;
                bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    for (const OpRecord& op : plan.ops) op.run(plan, ctx);
                    ok = hipStreamEndCapture(s, &graph) == hipSuccess && graph != nullptr;
                }
                hipGraphExec_t exec = nullptr;
                if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph) (void)hipGraphDestroy(graph);
                if (ok) {
                    ++graph_captures_;
                    slot->exec = exec;
                    slot->last_use = ++plan_clock_;
                    RD_HIP(hipGraphLaunch(exec, s));
                    return;
                }
                (void)hipGetLastError();
                plan.graph_broken = true;     // this plan does not capture: direct launches from now on
            } else {
                if (!retired_graphs_.empty()) sweep_retired_graphs(false);
                if (plan.graphs.size() >= kMaxGraphSlots) {
                    auto victim = plan.graphs.begin();
                    for (auto j = plan.graphs.begin(); j != plan.graphs.end(); ++j)
                        if (j->last_use < victim->last_use) victim = j;
                    if (victim->exec) retire_graph(victim->exec, victim->stream);   // possibly still in flight: freed after its event
                    plan.graphs.erase(victim);
                    if (++plan.graph_evictions > kMaxGraphEvictions) {
                        // the callers' pointers do not repeat for this plan (more live pointer sets than slots): every slot is
                        // evicted before it is seen again, so capturing only costs - launch directly from now on
                        for (auto& g : plan.graphs)
                            if (g.exec) retire_graph(g.exec, g.stream);
                        plan.graphs.clear();
                        plan.graph_broken = true;
                    }
                }
                if (plan.graph_broken) {
                    for (const OpRecord& op : plan.ops) op.run(plan, ctx);
                    RD_HIP(hipGetLastError());
                    return;
                }
                GraphSlot g;
                g.ext = ext; g.arena = ctx.arena; g.stream = s; g.last_use = ++plan_clock_;
                plan.graphs.push_back(std::move(g));
            }
        }
        for (const OpRecord& op : plan.ops) op.run(plan, ctx);
        RD_HIP(hipGetLastError());
        return;
    }
    const size_t need = plan.ops.size() + 1;
    while (events_.size() < need) {
        hipEvent_t ev;
        RD_HIP(hipEventCreate(&ev));
        events_.push_back(ev);
    }
    RD_HIP(hipEventRecord(events_[0], s));
    for (size_t i = 0; i < plan.ops.size(); ++i) {
        plan.ops[i].run(plan, ctx);
        RD_HIP(hipEventRecord(events_[i + 1], s));
    }
    RD_HIP(hipEventSynchronize(events_[plan.ops.size()]));
    RD_HIP(hipGetLastError());
    profile_.clear();
    for (size_t i = 0; i < plan.ops.size(); ++i) {
        float ms = 0.f;
        RD_HIP(hipEventElapsedTime(&ms, events_[i], events_[i + 1]));
        const OpRecord& op = plan.ops[i];
        profile_.push_back({op.name, op.kind, op.cfg, op.shape, op.flops, op.bytes, ms});
    }
}

std::string Engine::profile_json() const {
    std::ostringstream os;
    os << "[";
    for (size_t i = 0; i < profile_.size(); ++i) {
        const ProfileEntry& e = profile_[i];
        if (i) os << ",";
        os << "{\"name\":\"" << e.name << "\",\"kind\":\"" << e.kind << "\",\"cfg\":\"" << e.cfg << "\",\"shape\":\"" << e.shape
           << "\",\"flops\":" << e.flops
           << ",\"bytes\":" << e.bytes << ",\"ms\":" << e.ms << "}";
    }
    os << "]";
    return os.str();
}

}  // namespace rd
