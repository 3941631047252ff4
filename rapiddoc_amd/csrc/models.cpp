// Network topologies of the page hot path, written against the reference's state-dict names so the
// shipped .safetensors load unchanged.  Reference definitions (file:line, relative to /root/reference):
//   PPLCNetV4 ............ rapid_doc/model/ocr/ppocrv6_pytorch/modeling/backbones/rec_lcnetv4.py:7-311
//   RepLKFPN ............. .../necks/db_fpn.py:288-415
//   DBHead (ppocrv6) ..... .../heads/det_db_head.py:52-149
//   LightSVTR ............ .../necks/rnn.py:225-379
//   MultiHead CTC branch . .../heads/rec_multi_head.py:43-75
//   PPHGNetV2-B4 (det) ... rapid_doc/model/formula/rapid_formula_self/networks/backbones/rec_pphgnetv2.py:860-1477
#include "engine.h"

namespace rd {

using G = Builder::ConvGeom;

static G geom(int k, int s = 1) {
    G g;
    g.kh = g.kw = k;
    g.sh = g.sw = s;
    g.pt = g.pl = g.pb = g.pr = (k - 1) / 2;
    return g;
}
static G geom_same_even(int k) {  // padding='same' with an even kernel / explicit F.pad(0,1,0,1): pad right+bottom only
    G g;
    g.kh = g.kw = k;
    g.pt = g.pl = (k - 1) / 2;
    g.pb = g.pr = (k - 1) - (k - 1) / 2;
    return g;
}

// ---------------------------------------------------------------------------------------------------
// PPLCNetV4
// ---------------------------------------------------------------------------------------------------
struct LcBlockCfg { int k, cin, cout, sh, sw; bool se; };
bool g_disable_fused_mixer = false;  // RD_DISABLE_FUSED_MIXER=1: A/B switch for parity tests and profiling

static TView lc_stem(Builder& b, const std::string& p, const TView& x_nchw, int c1, int c2) {
    auto cw = [&](const char* n) { return p + "." + n + ".convolution.weight"; };
    auto bn = [&](const char* n) { return p + "." + n + ".normalization"; };
    // stem1 (H/2, c1) -> [max-pool | stem2a (c1/2) -> stem2b (c1)] -> cat (2 c1): one fused kernel in the split-fp16 modes
    (void)c1;
    TView cat = b.stem_front(cw("stem1"), bn("stem1"), cw("stem2a"), bn("stem2a"), cw("stem2b"), bn("stem2b"), x_nchw);
    TView s4 = b.stem_tail(cw("stem3"), bn("stem3"), cw("stem4"), bn("stem4"), cat, ACT_RELU, ACT_RELU);
    b.release(cat);
    (void)c2;
    return s4;
}

static TView lc_block(Builder& b, const std::string& p, const TView& x, const LcBlockCfg& c) {
    const bool rep = c.sh == 1 && c.sw == 1 && c.cin == c.cout;
    G g = geom(c.k);
    g.sh = c.sh;
    g.sw = c.sw;
    TView t;
    Builder::GapOut gap;  // the depthwise kernel also emits the SE pooling partial sums of its output
    Builder::GapOut* gp = c.se ? &gap : nullptr;
    // (round 6) a no-SE 3x3 block whose mixer runs on the resident-weights kernel: the depthwise conv is computed in that kernel's tile load
    // and its output never written (the PREPARE pass still folds the depthwise weights through dwconv)
    const bool fuse_dw = rep && !c.se && c.k == 3 && mixer_fused_supported(c.cin) && !g_disable_fused_mixer && b.mixer_takes_dw(p, c.cin);
    if (fuse_dw) {
        const std::string key = Builder::dw_key(p + ".token_conv.weight", "");
        return b.mixer_fused(p, x, nullptr, &key);
    }
    if (rep) t = b.dwconv(p + ".token_conv.weight", p + ".token_conv.bias", "", x, g, ACT_NONE, nullptr, nullptr, gp);
    else t = b.dwconv(p + ".token_conv.convolution.weight", "", p + ".token_conv.normalization", x, g, ACT_NONE, nullptr, nullptr, gp);
    TView gate;
    bool has_gate = false;
    if (c.se) {
        const std::string s = p + ".token_squeeze_excitation.convolutions.";
        gate = b.se_gate(s + "0.weight", s + "0.bias", s + "2.weight", s + "2.bias", t, ACT_HSIG, &gap);
        has_gate = true;
    }
    if (rep && mixer_fused_supported(c.cin) && !g_disable_fused_mixer) {
        // SE gate, expand, GELU, project and the residual add in one kernel; the gated tensor is never written
        TView o = b.mixer_fused(p, t, has_gate ? &gate : nullptr);
        if (has_gate) b.release(gate);
        b.release(t);
        return o;
    }
    if (has_gate) {
        b.scale(t, gate, 0.f, t);
        b.release(gate);
    }
    TView m = b.conv(p + ".channel_conv1.convolution.weight", "", p + ".channel_conv1.normalization", t, geom(1), ACT_GELU);
    TView o = b.conv(p + ".channel_conv2.convolution.weight", "", p + ".channel_conv2.normalization", m, geom(1), ACT_NONE,
                     nullptr, rep ? &t : nullptr);
    b.release(m);
    b.release(t);
    return o;
}

static const std::vector<std::vector<LcBlockCfg>> kDetSmall = {
    {{3, 48, 48, 1, 1, true}, {3, 48, 48, 1, 1, false}},
    {{3, 48, 96, 2, 2, false}, {3, 96, 96, 1, 1, true}, {3, 96, 96, 1, 1, false}},
    {{3, 96, 192, 2, 2, false}, {3, 192, 192, 1, 1, true}, {3, 192, 192, 1, 1, false}, {3, 192, 192, 1, 1, true},
     {3, 192, 192, 1, 1, false}},
    {{3, 192, 384, 2, 2, false}, {3, 384, 384, 1, 1, true}, {3, 384, 384, 1, 1, false}},
};
static const std::vector<std::vector<LcBlockCfg>> kRecSmall = {
    {{3, 96, 96, 1, 1, true}},
    {{3, 96, 96, 1, 1, false}, {3, 96, 96, 1, 1, false}},
    {{3, 96, 192, 2, 1, false}, {3, 192, 192, 1, 1, true}, {3, 192, 192, 1, 1, false}, {3, 192, 192, 1, 1, true},
     {3, 192, 192, 1, 1, false}, {3, 192, 192, 1, 1, true}, {3, 192, 192, 1, 1, false}},
    {{3, 192, 384, 2, 1, false}, {3, 384, 384, 1, 1, true}, {3, 384, 384, 1, 1, false}},
};

// returns the 4 stage outputs; the caller releases them
static std::vector<TView> lcnetv4(Builder& b, const TView& x_nchw, const std::vector<std::vector<LcBlockCfg>>& cfg,
                                  int c1, int c2, bool keep_all) {
    const std::string enc = "backbone.encoder";
    TView h = lc_stem(b, enc + ".convolution", x_nchw, c1, c2);
    std::vector<TView> feats;
    for (size_t si = 0; si < cfg.size(); ++si) {
        for (size_t bi = 0; bi < cfg[si].size(); ++bi) {
            TView o = lc_block(b, enc + ".blocks." + std::to_string(si) + ".blocks." + std::to_string(bi), h, cfg[si][bi]);
            const bool is_feat = !feats.empty() && feats.back().buf == h.buf;
            if (!is_feat) b.release(h);
            h = o;
        }
        if (keep_all || si + 1 == cfg.size()) feats.push_back(h);
    }
    return feats;
}

// ---------------------------------------------------------------------------------------------------
// PP-OCRv6 det small.   ext[0] = x NCHW [B,3,H,W];  ext[1] = prob map [B,1,H,W]
// ---------------------------------------------------------------------------------------------------
void build_ppocrv6_det(Builder& b, int B, int H, int W) {
    RD_CHECK(H % 32 == 0 && W % 32 == 0 && H >= 32 && W >= 32, "det input H, W must be multiples of 32");
    TView x = b.external(0, B, H, W, 3);
    TView out = b.external(1, B, H, W, 1);
    std::vector<TView> f = lcnetv4(b, x, kDetSmall, 24, 48, true);

    // RepLKFPN
    std::vector<TView> fused(4);
    for (int i = 0; i < 4; ++i) {
        const std::string p = "neck.insert_conv." + std::to_string(i);
        TView y = b.conv(p + ".in_conv.weight", "", "", f[i], geom(1), ACT_NONE);
        b.release(f[i]);
        const std::string s = p + ".squeeze_excitation_block.";
        TView gate = b.se_gate(s + "conv1.weight", s + "conv1.bias", s + "conv2.weight", s + "conv2.bias", y, ACT_HSIG_PADDLE);
        b.scale(y, gate, 1.f, y);  // y + y*s
        b.release(gate);
        fused[i] = y;
    }
    for (int i = 2; i >= 0; --i) b.upsample(fused[i + 1], fused[i], 2, true);
    TView cat = b.alloc(B, H / 4, W / 4, 96);
    for (int i = 0; i < 4; ++i) {
        const std::string p = "neck.input_conv." + std::to_string(i);
        TView d = b.dwconv(p + ".depthwise_convolution.weight", p + ".depthwise_convolution.bias", "", fused[i], geom(7), ACT_NONE);
        b.release(fused[i]);
        TView z = b.conv(p + ".pointwise_convolution.weight", "", "", d, geom(1), ACT_NONE);
        b.release(d);
        const std::string s = p + ".squeeze_excitation_module.";
        TView gate = b.se_gate(s + "conv1.weight", s + "conv1.bias", s + "conv2.weight", s + "conv2.bias", z, ACT_HSIG_PADDLE);
        TView slot = b.slice(cat, 24 * (3 - i), 24);  // cat(processed[::-1]) : deepest level first (db_fpn.py:415)
        if (i == 0) {
            b.scale(z, gate, 1.f, slot);
        } else {
            b.scale(z, gate, 1.f, z);
            b.upsample(z, slot, 1 << i, false);
        }
        b.release(gate);
        b.release(z);
    }
    // DBHead v6
    TView c = b.conv("head.conv_down.convolution.weight", "", "head.conv_down.norm", cat, geom(3), ACT_RELU);
    b.release(cat);
    if (b.deconv_pair_to_prob("head.conv_up.convolution.weight", "head.conv_up.convolution.bias", "head.conv_up.norm", "head.conv_final.weight",
                              "head.conv_final.bias", c, out)) {
        b.release(c);
    } else {
        TView u = b.deconv2x2("head.conv_up.convolution.weight", "head.conv_up.convolution.bias", "head.conv_up.norm", c, ACT_RELU);
        b.release(c);
        b.deconv2x2("head.conv_final.weight", "head.conv_final.bias", "", u, ACT_SIGMOID, &out);
        b.release(u);
    }
}

// ---------------------------------------------------------------------------------------------------
// PP-OCRv6 rec small.  ext[0] = x NCHW [B,3,48,W]; ext[1] = idx i32 [B*T]; ext[2] = prob f32 [B*T];
// ext[3] = optional [B,T,C] softmax probabilities or raw logits (flags)
//
// Two-stage form for the page pipeline (the neck + head of one 64-line batch is ~25 launches of 17-30 us on 4352 tokens:
// pure launch latency, ~45 % of a batch's kernel count for ~3 % of its FLOPs):
//   REC_STAGE_BACKBONE   x -> avg-pooled backbone tokens only: ext[1] = [B][T][384] f32 out (the caller points it INTO one
//                        token buffer shared by all batches of the page group)
//   REC_STAGE_TAIL       LightSVTR neck + CTC head ONCE over the tokens of all batches: B = text lines, H = the longest
//                        line in tokens, W = all tokens; ext[0] = tokens [W][384], ext[1] / ext[2] = idx / prob [W],
//                        ext[4] = seg i32 [B][2] (first token, tokens per line), ext[5] = tokinfo i32 [W] (position in the
//                        line | tokens of the line << 16).  Every layer of the neck is row-wise except the 1x7 depthwise
//                        conv and the attention, which take the line structure from those two tables; lines of different
//                        batches (different widths) simply have different lengths.
// ---------------------------------------------------------------------------------------------------
void build_ppocrv6_rec(Builder& b, int B, int H, int W, int flags) {
    const bool tail_only = (flags & REC_STAGE_TAIL) != 0, backbone_only = (flags & REC_STAGE_BACKBONE) != 0;
    RD_CHECK(!(tail_only && backbone_only), "rec: choose one stage");
    const std::string e = "head.encoder";
    auto cw = [&](int i) { return e + ".conv_block." + std::to_string(i) + ".convolution.weight"; };
    auto cbn = [&](int i) { return e + ".conv_block." + std::to_string(i) + ".normalization"; };
    TView pooled, seg, tokinfo;
    int T, n_seq = B;
    if (tail_only) {
        RD_CHECK((flags & ~REC_STAGE_TAIL) == 0, "rec tail: fused CTC only");
        RD_CHECK(B >= 1 && H >= 1 && H < 32768 && W >= B, "rec tail: B lines, H = longest line (tokens), W = all tokens");
        pooled = b.external(0, 1, 1, W, b.weight_dim(cw(0), 1));
        seg = b.external(4, B, 1, 1, 2);
        tokinfo = b.external(5, 1, 1, W, 1);
        T = H;
    } else {
        RD_CHECK(H == 48, "rec input height must be 48");
        RD_CHECK(W >= 16, "rec input width must be >= 16");
        TView x = b.external(0, B, H, W, 3);
        if (flags & REC_LINE_WIDTHS) {
            RD_CHECK(backbone_only, "rec: per-line widths belong to the backbone stage");
            b.set_line_table(b.external(2, B, 1, 1, kLineTabStride));
        }
        std::vector<TView> f = lcnetv4(b, x, kRecSmall, 48, 96, false);
        if (backbone_only) {
            RD_CHECK((flags & ~(REC_STAGE_BACKBONE | REC_LINE_WIDTHS)) == 0, "rec backbone stage takes no other flag");
            TView out = b.external(1, B, 1, (f[0].w - 2) / 2 + 1, f[0].c);
            b.avgpool3x2(f[0], &out);
            b.release(f[0]);
            return;
        }
        pooled = b.avgpool3x2(f[0]);  // [B,1,W/8,384]
        b.release(f[0]);
        T = pooled.w;
        RD_CHECK(pooled.h == 1, "rec: pooled height");
    }
    const TView* ti = tail_only ? &tokinfo : nullptr;
    const TView* sg = tail_only ? &seg : nullptr;

    TView res = b.conv(cw(0), "", cbn(0), pooled, geom(1), ACT_SILU);
    TView h = b.conv(cw(1), "", cbn(1), pooled, geom(1), ACT_SILU);
    b.release(pooled);
    G g17;
    g17.kh = 1;
    g17.kw = b.weight_dim(cw(2), 3);
    g17.pl = g17.pr = g17.kw / 2;
    TView t = b.dwconv(cw(2), "", cbn(2), h, g17, ACT_SILU, nullptr, &h, nullptr, ti);  // h + silu(bn(dw(h)))
    b.release(h);
    const int C = t.c, heads = 8, hd = C / heads;
    int depth = 0;
    while (b.has_weight(e + ".svtr_block." + std::to_string(depth) + ".layer_norm1.weight")) ++depth;
    for (int d = 0; d < depth; ++d) {
        const std::string p = e + ".svtr_block." + std::to_string(d);
        TView y = b.layernorm(p + ".layer_norm1", t, 1e-6f);
        TView qkv = b.linear(p + ".self_attn.qkv", y, ACT_NONE);
        b.release(y);
        TView a = b.attention(qkv, n_seq, T, heads, hd, sg);
        b.release(qkv);
        TView t2 = b.linear(p + ".self_attn.projection", a, ACT_NONE, nullptr, &t);
        b.release(a);
        b.release(t);
        TView y2 = b.layernorm(p + ".layer_norm2", t2, 1e-6f);
        TView m = b.linear(p + ".mlp.fc1", y2, ACT_SILU);
        b.release(y2);
        t = b.linear(p + ".mlp.fc2", m, ACT_NONE, nullptr, &t2);
        b.release(m);
        b.release(t2);
    }
    TView n = b.layernorm(e + ".norm", t, 1e-6f);
    b.release(t);
    TView seq = b.add(n, res);  // [B,1,T,120]
    b.release(n);
    b.release(res);

    const int ncls = b.weight_dim("head.head.weight", 0);
    TView idx = tail_only ? b.external(1, 1, 1, W, 1) : b.external(1, B, 1, T, 1);
    TView prob = tail_only ? b.external(2, 1, 1, W, 1) : b.external(2, B, 1, T, 1);
    const bool want_full = (flags & (REC_WANT_SOFTMAX | REC_WANT_LOGITS)) != 0;
    if ((flags & REC_UNFUSED_CTC) || want_full) {
        if (flags & REC_WANT_LOGITS) {
            TView lg = b.external(3, B, 1, T, ncls);
            b.linear("head.head", seq, ACT_NONE, &lg);
            b.ctc_stats(lg, idx, prob);
        } else {
            TView lg = b.linear("head.head", seq, ACT_NONE);
            if (flags & REC_WANT_SOFTMAX) {
                // (idx, prob) = numpy's argmax / max of the softmax tensor as written: what the host's CTC decode would compute from it
                TView sm = b.external(3, B, 1, T, ncls);
                b.softmax_rows(lg, sm, &idx, &prob);
            } else {
                b.ctc_stats(lg, idx, prob);
            }
            b.release(lg);
        }
    } else {
        b.ctc_head("head.head", seq, idx, prob);
    }
    b.release(seq);
}

// ---------------------------------------------------------------------------------------------------
// PPHGNetV2-B4 (det=True): the PP-DocLayout-L / plus-L / V2 / V3 backbone.
// ext[0] = x NCHW [B,3,H,W]; ext[1..4] = stage outputs NCHW (strides 4/8/16/32; 128/512/1024/2048 ch)
// ---------------------------------------------------------------------------------------------------
struct HgStageCfg { int cin, mid, cout, blocks; bool down, light; int k, layers; };
static const HgStageCfg kB4Det[4] = {
    {48, 48, 128, 1, false, false, 3, 6},
    {128, 96, 512, 1, true, false, 3, 6},
    {512, 192, 1024, 3, true, true, 5, 6},
    {1024, 384, 2048, 1, true, true, 5, 6},
};

static const HgStageCfg kB6Formula[4] = {   // rec_pphgnetv2.py:1601-1607
    {96, 96, 192, 2, false, false, 3, 6},
    {192, 192, 512, 3, true, false, 3, 6},
    {512, 384, 1024, 6, true, true, 5, 6},
    {1024, 768, 2048, 3, true, true, 5, 6},
};

// Shared PPHGNetV2 body.  `pre` = state-dict prefix ("" or "backbone.pphgnet_b6.").  Every stage output listed in
// `want` is handed to `emit(stage, view)`.
template <typename Emit>
static void build_pphgnetv2(Builder& b, const TView& x, const HgStageCfg (&cfg)[4], const std::string& pre, Emit emit) {
    const int B = x.n;
    auto cw = [&](const std::string& p) { return pre + p + ".conv.weight"; };
    auto bn = [&](const std::string& p) { return pre + p + ".bn"; };
    // stem (StemBlock, rec_pphgnetv2.py:979-1056)
    TView cat = b.stem_front(cw("stem.stem1"), bn("stem.stem1"), cw("stem.stem2a"), bn("stem.stem2a"), cw("stem.stem2b"),
                             bn("stem.stem2b"), x);

    // stage inputs are produced straight into channel slot 0 of the stage's dense-concat buffer
    TView cur, cur_cat;
    auto new_cat = [&](int n, int h, int w, const HgStageCfg& c, int cin) { return b.alloc(n, h, w, cin + c.layers * c.mid); };
    {
        const HgStageCfg& c = cfg[0];
        RD_CHECK(!c.down, "PPHGNetV2: the first stage does not downsample");
        const int s3h = (cat.h + 2 - 3) / 2 + 1, s3w = (cat.w + 2 - 3) / 2 + 1;      // stem3: 3x3 / stride 2 / pad 1
        cur_cat = new_cat(B, s3h, s3w, c, c.cin);
        cur = b.slice(cur_cat, 0, c.cin);
        b.stem_tail(cw("stem.stem3"), bn("stem.stem3"), cw("stem.stem4"), bn("stem.stem4"), cat, ACT_RELU, ACT_RELU, &cur);
        b.release(cat);
    }
    for (int si = 0; si < 4; ++si) {
        const HgStageCfg& c = cfg[si];
        const std::string sp = "stages." + std::to_string(si);
        RD_CHECK(si == 0 || c.down, "PPHGNetV2: stages 2-4 downsample");
        if (c.down) {  // depthwise 3x3 stride 2 + BN, no activation (HGV2_Stage.downsample)
            const int oh = (cur.h + 2 - 3) / 2 + 1, ow = (cur.w + 2 - 3) / 2 + 1;
            TView ncat = new_cat(B, oh, ow, c, c.cin);
            TView nin = b.slice(ncat, 0, c.cin);
            b.dwconv(cw(sp + ".downsample"), "", bn(sp + ".downsample"), cur, geom(3, 2), ACT_NONE, &nin);
            b.release(cur_cat);
            cur_cat = ncat;
            cur = nin;
        }
        for (int bi = 0; bi < c.blocks; ++bi) {
            const std::string bp = sp + ".blocks." + std::to_string(bi);
            const int cin = bi == 0 ? c.cin : c.cout;
            TView prev = cur;
            for (int li = 0; li < c.layers; ++li) {
                const std::string lp = bp + ".layers." + std::to_string(li);
                TView slot = b.slice(cur_cat, cin + li * c.mid, c.mid);
                if (c.light) {
                    TView t = b.conv(cw(lp + ".conv1"), "", bn(lp + ".conv1"), prev, geom(1), ACT_NONE);
                    b.dwconv(cw(lp + ".conv2"), "", bn(lp + ".conv2"), t, geom(c.k), ACT_RELU, &slot);
                    b.release(t);
                } else {
                    b.conv(cw(lp), "", bn(lp), prev, geom(c.k), ACT_RELU, &slot);
                }
                prev = slot;
            }
            TView full = b.slice(cur_cat, 0, cin + c.layers * c.mid);
            TView sq = b.conv(cw(bp + ".aggregation_squeeze_conv"), "", bn(bp + ".aggregation_squeeze_conv"), full, geom(1), ACT_RELU);
            // destination of the block output: slot 0 of the next block's concat buffer, or a plain buffer at a stage end
            const bool last_block = bi + 1 == c.blocks;
            TView ncat, nout;
            if (!last_block) {
                ncat = new_cat(B, cur.h, cur.w, c, c.cout);
                nout = b.slice(ncat, 0, c.cout);
            } else {
                ncat = b.alloc(B, cur.h, cur.w, c.cout);
                nout = ncat;
            }
            const bool identity = bi > 0;
            b.conv(cw(bp + ".aggregation_excitation_conv"), "", bn(bp + ".aggregation_excitation_conv"), sq, geom(1), ACT_RELU,
                   &nout, identity ? &cur : nullptr);
            b.release(sq);
            b.release(cur_cat);
            cur_cat = ncat;
            cur = nout;
        }
        emit(si, cur);
    }
    b.release(cur_cat);
}

void build_pphgnetv2_b4(Builder& b, int B, int H, int W) {
    RD_CHECK(H % 32 == 0 && W % 32 == 0 && H >= 64 && W >= 64, "backbone input H, W must be multiples of 32 (>= 64)");
    TView x = b.external(0, B, H, W, 3);
    build_pphgnetv2(b, x, kB4Det, "", [&](int si, const TView& v) {
        TView o = b.external(1 + si, B, v.h, v.w, v.c);
        b.to_nchw(v, o);
    });
}

// ---------------------------------------------------------------------------------------------------
// PP-FormulaNet_plus-M encoder = PPHGNetV2_B6_Formula (rec_pphgnetv2.py:1587-1642): x [B,1|3,H,W] -> [B, H/32*W/32, 2048].
// NHWC [B,h,w,2048] IS the reference's reshape(b,c,h*w).permute(0,2,1), so the result is written without a transpose.
// ext[0] = x NCHW; ext[1] = encoder states [B, h*w, 2048]
// ---------------------------------------------------------------------------------------------------
void build_pphgnetv2_b6_formula(Builder& b, int B, int H, int W, int flags) {
    RD_CHECK(H % 32 == 0 && W % 32 == 0 && H >= 64 && W >= 64, "formula encoder input H, W must be multiples of 32 (>= 64)");
    TView x = b.external(0, B, H, W, (flags & 1) ? 1 : 3);
    build_pphgnetv2(b, x, kB6Formula, "backbone.pphgnet_b6.", [&](int si, const TView& v) {
        if (si != 3) return;
        TView o = b.external(1, B, v.h, v.w, v.c);
        b.copy(v, o);
    });
}

}  // namespace rd
