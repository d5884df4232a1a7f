// Onsets & Frames inference engine for gfx950: weight packing + the launch sequence of one
// `run_on_batch` forward (amt_tools/models/onsetsframes.py:94-136 forward, :138-196 post_proc without the
// loss; TranscriptionModel.run_on_batch amt_tools/models/common.py:151-184).
//
// Tensors are handed over under the reference's own state_dict names (e.g. "onset_head.0.layer2.0.weight"),
// eval-mode BatchNorm is folded, fc1's columns are permuted from the reference's (channel, freq) flatten
// order to the channels-last (freq, channel) order the conv kernels write, LSTM biases b_ih + b_hh are merged
// into the input-projection GEMM, everything is packed into MFMA fragment order once.
//
// Launch sequence (acoustic heads batched as kernel groups):
//   conv1 -> conv2+pool -> conv3+pool -> fc1            [groups = heads: onset, (offset), pitch]
//   onset(/offset) x-proj GEMM -> BiLSTM -> head GEMM -> joint[:, 0:88(:176)]
//   pitch head GEMM -> joint[:, 88:176]
//   adjoin x-proj GEMM (A = joint logits, fp32) -> BiLSTM -> head GEMM -> frame logits
//   piano-roll finalize (sigmoid, threshold 0.5, transpose) x2

#include "amtx_kernels.h"
#include "amtx_kernels_f16.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Tensor { std::vector<float> v; };

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int upload(const void* host, size_t n) {
        if (!p || bytes != n) {        // a weight re-sync (validate() inside train()) keeps its allocations: same model, same sizes
            if (p) (void)hipFree(p);
            p = nullptr; bytes = n;
            AMTX_CHECK_HIP(hipMalloc(&p, n));
        }
        AMTX_CHECK_HIP(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
        return AMTX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
};

struct LinearPack { DevBuf w, b; int N = 0, K = 0, n_pad = 0, k_pad = 0; };

}  // namespace

struct amtx_of_model {
    int dim_in, in_channels, mc, n_out, has_offsets, precision;
    int planes, act_type;
    bool f16 = false;                          // AMTX_PREC_F16: the half-operand builds of the conv / GEMM / BiLSTM kernels, weights packed as half
    int nf1, nf2, nf3, dim_am, dim_lm, fq, kfc;
    int kfc_pad;                               // fc1's K rounded up to the DMA GEMM's 64-deep k-tile (rows of a3 are this long)
    int hid, xw;                               // LSTM hidden size per direction, width of an x-projection row (2 dirs x 4 gates x hid)
    bool gen_conv = false;                     // channel counts other than 32/32/64: convg.hip (weights in LDS) instead of conv.hip
    bool gen_conv2 = false;                    // conv2 (+ fused first conv) through convg.hip: gen_conv, or 32 -> 32 with a multi-channel input
    int n_heads;                               // acoustic heads: onset, (offset), pitch
    int n_rec;                                 // recurrent heads feeding the joint: onset, (offset)
    std::vector<std::string> head_names;       // state_dict prefixes of the acoustic models, group order
    std::map<std::string, Tensor> tensors;
    std::map<std::string, std::pair<const float*, int64_t>> dev_tensors;   // amtx_of_model_set_tensor_device: borrowed device pointers
    DevBuf pack_scratch;                       // device re-sync: BatchNorm scale / shift, the folded pitch head, the unused backward LSTM fragments
    bool finalized = false;
    bool packed_once = false;                  // the packed buffers hold a weight version that forwards may still be reading
    // packed device weights (group-major)
    DevBuf conv1_w, conv1_s, conv1_frag, conv2_w, conv2_s, conv3_w, conv3_s;
    DevBuf conv2_wx;                           // x12m: layer2's weights a second time, in conv.hip's fragment order (convx12_kernel<true> reads them; conv2_w is convg.hip's)
    bool x12m = false;                         // round 6: two planes, 2 .. 8 input channels, 32 / 32 channels: conv1 + conv2 from two-plane 16-bit features on convx.hip
    bool fuse_conv1 = false;                   // first conv computed inside the conv2 kernel (9*C_in <= 64)
    bool fuse_stack = false;                   // layer1 -> layer2 -> layer3 in one kernel (convf.hip) for batches that fill the chip with strips
    bool split_acts = false;                   // x3 (round 5): the dense layers' activations live in HBM as two 16-bit planes (AMTX_T_SPLIT), written by
                                               // the producing kernel's epilogue; the GEMMs DMA them straight into LDS (gemm_split_kernel)
    LinearPack fc1;                            // groups = n_heads
    LinearPack rec_ih;                         // groups = n_rec, N = 1024
    DevBuf rec_hh;                             // groups = n_rec
    LinearPack rec_out;                        // groups = n_rec, N = n_out, K = dim_lm
    LinearPack pitch_out;                      // the pitch head's fc1 and LogisticBank folded into one layer: K = kfc_pad, N = n_out
    LinearPack adj_ih;                         // K = dim_aj, N = 1024
    DevBuf adj_hh;
    LinearPack adj_out;
    int dim_aj;
    // optional per-stage timing with HIP events recorded on the launch stream (bench.py roofline)
    mutable bool prof = false;
    mutable std::vector<std::vector<hipEvent_t>> prof_events;
};

enum { ST_CONV1 = 0, ST_CONV2, ST_CONV3, ST_FC1, ST_REC_XPROJ, ST_REC_LSTM, ST_REC_HEAD, ST_PITCH_HEAD, ST_ADJ_XPROJ, ST_ADJ_LSTM,
       ST_ADJ_HEAD, ST_PIANOROLL, ST_COUNT };
static const char* kStageNames[ST_COUNT] = {"conv1", "conv2_pool", "conv3_pool", "fc1_gemm", "rec_xproj_gemm", "rec_bilstm", "rec_head_gemm",
                                            "pitch_head_gemm", "adj_xproj_gemm", "adj_bilstm", "adj_head_gemm", "pianoroll"};

namespace {

int need(const amtx_of_model* m, const std::string& name, size_t numel, const float** out) {
    auto it = m->tensors.find(name);
    if (it == m->tensors.end()) {
        amtx_set_error("of_model: tensor '%s' was not provided", name.c_str());
        return AMTX_ERR_ARG;
    }
    if (it->second.v.size() != numel) {
        amtx_set_error("of_model: tensor '%s' has %zu elements, expected %zu", name.c_str(), it->second.v.size(), numel);
        return AMTX_ERR_ARG;
    }
    *out = it->second.v.data();
    return AMTX_OK;
}

#define NEED(name, numel, ptr)                                         \
    do {                                                               \
        int _rc = need(m, name, numel, &(ptr));                        \
        if (_rc != AMTX_OK) return _rc;                                \
    } while (0)

// fold eval-mode BatchNorm2d behind a conv: scale[c] and shift[c]
int fold_bn(const amtx_of_model* m, const std::string& conv, const std::string& bn, int c_out, std::vector<float>& scale,
            std::vector<float>& shift) {
    const float *cb, *g, *be, *mu, *var;
    NEED(conv + ".bias", (size_t)c_out, cb);
    NEED(bn + ".weight", (size_t)c_out, g);
    NEED(bn + ".bias", (size_t)c_out, be);
    NEED(bn + ".running_mean", (size_t)c_out, mu);
    NEED(bn + ".running_var", (size_t)c_out, var);
    scale.resize(c_out); shift.resize(c_out);
    for (int c = 0; c < c_out; ++c) {
        const double s = (double)g[c] / std::sqrt((double)var[c] + 1e-5);
        scale[c] = (float)s;
        shift[c] = (float)((double)be[c] + ((double)cb[c] - (double)mu[c]) * s);
    }
    return AMTX_OK;
}

int pack_linear_groups(amtx_of_model* m, LinearPack& lp, const std::vector<std::vector<float>>& Ws, const std::vector<std::vector<float>>& bs,
                       int N, int K) {
    lp.N = N; lp.K = K;
    amtx_gemm_pack_dims(N, K, &lp.n_pad, &lp.k_pad);
    const size_t per = (size_t)lp.n_pad * lp.k_pad * m->planes;
    std::vector<bf16_t> packed(per * Ws.size());
    std::vector<float> bias((size_t)N * Ws.size());
    for (size_t g = 0; g < Ws.size(); ++g) {
        (m->f16 ? amtx_gemm_pack_host_f16 : amtx_gemm_pack_host)(Ws[g].data(), K, N, K, m->planes, packed.data() + g * per);
        memcpy(bias.data() + g * N, bs[g].data(), sizeof(float) * N);
    }
    int rc = lp.w.upload(packed.data(), packed.size() * sizeof(bf16_t));
    if (rc != AMTX_OK) return rc;
    return lp.b.upload(bias.data(), bias.size() * sizeof(float));
}

int pack_lstm(amtx_of_model* m, const std::string& prefix, int dim_in, std::vector<float>& w_ih, std::vector<float>& b, std::vector<bf16_t>& hh) {
    const int H = m->dim_lm / 2, G = 4 * H;
    const float *wif, *wib, *whf, *whb, *bif, *bib, *bhf, *bhb;
    NEED(prefix + ".mlm.weight_ih_l0", (size_t)G * dim_in, wif);
    NEED(prefix + ".mlm.weight_ih_l0_reverse", (size_t)G * dim_in, wib);
    NEED(prefix + ".mlm.weight_hh_l0", (size_t)G * H, whf);
    NEED(prefix + ".mlm.weight_hh_l0_reverse", (size_t)G * H, whb);
    NEED(prefix + ".mlm.bias_ih_l0", (size_t)G, bif);
    NEED(prefix + ".mlm.bias_ih_l0_reverse", (size_t)G, bib);
    NEED(prefix + ".mlm.bias_hh_l0", (size_t)G, bhf);
    NEED(prefix + ".mlm.bias_hh_l0_reverse", (size_t)G, bhb);
    w_ih.resize((size_t)2 * G * dim_in);
    memcpy(w_ih.data(), wif, sizeof(float) * G * dim_in);
    memcpy(w_ih.data() + (size_t)G * dim_in, wib, sizeof(float) * G * dim_in);
    b.resize(2 * G);
    for (int i = 0; i < G; ++i) { b[i] = bif[i] + bhf[i]; b[G + i] = bib[i] + bhb[i]; }
    hh.resize(amtx_bilstm_wfrag_elems_h(H, m->planes));
    (m->f16 ? amtx_bilstm_pack_host_h_f16 : amtx_bilstm_pack_host_h)(whf, whb, H, m->planes, hh.data());
    return AMTX_OK;
}

struct Workspace {
    char *a1, *a2, *a3, *e, *xp, *l1, *joint, *joint16, *xp2, *l2, *mp;
    size_t total;
};

Workspace carve(const amtx_of_model* m, int B, int T, char* base) {
    Workspace w;
    const size_t es = amtx_tsize(m->act_type);
    const size_t BT = (size_t)B * T;
    const int F = m->dim_in, F2 = F / 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    w.a1 = take(m->fuse_conv1 ? 256 : BT * F * m->nf1 * es * m->n_heads);
    // the 32-channel map behind layer2 only exists in HBM on the two-kernel path
    w.a2 = take(m->fuse_stack && amtx_conv_stack_fused_ok(B, T, F, m->n_heads) ? 256 : BT * F2 * m->nf2 * es * m->n_heads);
    w.a3 = take(BT * m->kfc_pad * es * m->n_heads);
    w.e = take(BT * m->dim_am * es * m->n_heads);
    w.xp = take(BT * m->xw * es * m->n_rec);
    w.l1 = take(BT * m->dim_lm * es * m->n_rec);
    w.joint = take(BT * m->dim_aj * sizeof(float));
    w.joint16 = take(BT * (size_t)((m->dim_aj + 63) / 64 * 64) * 2 * (m->split_acts ? 2 : 1));   // bf16 copy (two planes with split_acts), K padded to the GEMM's 64-deep k-tile
    w.xp2 = take(BT * m->xw * es);
    w.l2 = take(BT * m->dim_lm * es);
    w.mp = take(BT * m->n_out * sizeof(float));
    w.total = off;
    return w;
}

}  // namespace

// 1 when the half-operand twins of conv / convf / convg / gemm / lstm / pack.hip are linked in (precision AMTX_PREC_F16 available)
extern "C" int amtx_has_f16(void) {
#ifdef AMTX_WITH_F16
    return 1;
#else
    return 0;
#endif
}

extern "C" int amtx_of_model_create(amtx_of_model** out, int dim_in, int in_channels, int model_complexity, int n_out,
                                    int has_offsets, int precision) {
    AMTX_REQUIRE(out, "amtx_of_model_create: null model pointer");
    *out = nullptr;
    AMTX_REQUIRE(precision == AMTX_PREC_BF16 || precision == AMTX_PREC_X3 || precision == AMTX_PREC_F16, "amtx_of_model_create: bad precision");
    AMTX_REQUIRE(dim_in >= 4 && in_channels >= 1 && n_out > 0 && n_out % 4 == 0, "amtx_of_model_create: bad dims");
    if (precision == AMTX_PREC_F16 && !amtx_has_f16()) {
        amtx_set_error("amtx_of_model_create: precision f16 needs a library built with the half-operand kernel twins (AMTX_BUILD_F16=1 python -m amt_tools_amd.build)");
        return AMTX_ERR_UNSUPPORTED;
    }
    if (model_complexity < 2 || model_complexity > 5) {
        amtx_set_error("amtx_of_model_create: model_complexity 2 (32/32/64-channel convolutions, LSTM hidden 128), 3 (48/48/96, hidden 256) and "
                       "4 (64/64/128, hidden 384), 5 (80/80/160, hidden 512) are implemented, with or without the OnsetsFrames2 offset head (got model_complexity=%d)", model_complexity);
        return AMTX_ERR_UNSUPPORTED;
    }
    amtx_of_model* m = new amtx_of_model();
    m->dim_in = dim_in; m->in_channels = in_channels; m->mc = model_complexity; m->n_out = n_out;
    m->has_offsets = has_offsets; m->precision = precision;
    m->planes = precision == AMTX_PREC_X3 ? 2 : 1;
    m->act_type = precision == AMTX_PREC_X3 ? AMTX_T_F32 : AMTX_T_BF16;      // AMTX_T_BF16 = "16-bit operand format": half in the f16 mode
    m->f16 = precision == AMTX_PREC_F16;
    m->nf1 = 16 * model_complexity; m->nf2 = m->nf1; m->nf3 = 32 * model_complexity;
    m->dim_am = 256 * model_complexity; m->dim_lm = 256 * (model_complexity - 1);
    m->fq = dim_in / 4;                      // two MaxPool(1,2): floor(floor(F/2)/2) == F//4
    m->kfc = m->nf3 * m->fq;
    m->kfc_pad = (m->kfc + 63) / 64 * 64;
    m->hid = m->dim_lm / 2; m->xw = 8 * m->hid;
    m->gen_conv = m->nf1 != 32;
    // 32/32/64 channels with more than one input channel (HCQT): conv.hip's fused first conv runs its 9 c_in taps as four legacy
    // 16-deep MFMA steps and stages c_in x 20 x (columns + 4) feature values per tile through a register-starved loop (3.6 ms per
    // 512 HCQT clips); convg.hip's fused first conv (two 32-deep steps, weights in LDS) is the faster one there (2.7 ms).  With one
    // input channel conv.hip stays far ahead: 5.6 vs 9.7 ms per 1024 mel clips (weights stationary in registers, 18 x 46 tiles).
    m->gen_conv2 = m->gen_conv || (in_channels > 1 && getenv("AMTX_NO_CONVG_MC2") == nullptr && amtx_conv3x3_gen_can_fuse1(in_channels, m->nf1, m->nf2, m->planes));
    m->head_names = {"onset_head"};
    if (has_offsets) m->head_names.push_back("offset_head");
    m->n_rec = (int)m->head_names.size();
    m->head_names.push_back("pitch_head");
    m->n_heads = (int)m->head_names.size();
    m->dim_aj = (m->n_rec + 1) * n_out;
    m->fuse_conv1 = m->gen_conv2 ? amtx_conv3x3_gen_can_fuse1(in_channels, m->nf1, m->nf2, m->planes) : (9 * in_channels <= 64);
    // A/B switch: AMTX_NO_CONV_FUSE=1 keeps conv.hip's two kernels (conv1+conv2, conv3) at every batch size
    m->fuse_stack = !m->gen_conv && !m->gen_conv2 && m->fuse_conv1 && in_channels == 1 && m->planes == 1 && getenv("AMTX_NO_CONV_FUSE") == nullptr;
    // A/B switch: AMTX_X3_NO_SPLIT=1 keeps fp32 activations between the two-plane kernels (round 4's data path)
    // (one input channel: conv.hip / convx.hip write the planes; 2 .. 7 input channels at 32 / 32 / 64 channels -- HCQT --: convg.hip's two-plane
    // kernel writes a2 as planes, convx.hip's conv3 and the GEMMs behind it are the same)
    m->split_acts = m->planes == 2 && !m->gen_conv && m->fuse_conv1 && getenv("AMTX_X3_NO_SPLIT") == nullptr &&
                    ((!m->gen_conv2 && in_channels == 1) || (m->gen_conv2 && in_channels > 1 && (9 * in_channels + 31) / 32 <= 2));
    // A/B switch: AMTX_NO_CONVX12M=1 keeps convg.hip's two-plane kernel (fp32 features) for the multi-channel first conv
    m->x12m = m->split_acts && m->gen_conv2 && in_channels >= 2 && in_channels <= 8 && m->nf1 == 32 && m->nf2 == 32 && amtx_conv1g_tapk(in_channels, 2) &&
              getenv("AMTX_NO_CONVX12M") == nullptr;
    *out = m;
    return AMTX_OK;
}

extern "C" int amtx_of_model_destroy(amtx_of_model* m) {
    if (!m) return AMTX_OK;
    DevBuf* bufs[] = {&m->conv1_w, &m->conv1_s, &m->conv1_frag, &m->conv2_w, &m->conv2_wx, &m->conv2_s, &m->conv3_w, &m->conv3_s, &m->fc1.w, &m->fc1.b,
                      &m->rec_ih.w, &m->rec_ih.b, &m->rec_hh, &m->rec_out.w, &m->rec_out.b, &m->pitch_out.w, &m->pitch_out.b,
                      &m->adj_ih.w, &m->adj_ih.b, &m->adj_hh, &m->adj_out.w, &m->adj_out.b, &m->pack_scratch};
    for (DevBuf* b : bufs) b->release();
    delete m;
    return AMTX_OK;
}

extern "C" int amtx_of_model_set_tensor(amtx_of_model* m, const char* name, const float* host_data, int64_t numel) {
    AMTX_REQUIRE(m && name && host_data && numel > 0, "amtx_of_model_set_tensor: bad argument");
    Tensor& t = m->tensors[name];
    t.v.assign(host_data, host_data + numel);
    m->finalized = false;
    return AMTX_OK;
}

// A RE-sync overwrites the packed buffers in place (DevBuf::upload keeps its allocation; the device packers write on the caller's
// stream).  A forward pass of the previous weight version may still be in flight on ANOTHER stream (PyTorch side streams do not
// synchronise with the null stream), so both re-sync entry points first wait for everything the device has been given.
static int quiesce_before_resync(const amtx_of_model* m) {
    if (m->packed_once) AMTX_CHECK_HIP(hipDeviceSynchronize());
    return AMTX_OK;
}

extern "C" int amtx_of_model_finalize(amtx_of_model* m) {
    AMTX_REQUIRE(m, "amtx_of_model_finalize: null model");
    {
        int qrc = quiesce_before_resync(m);
        if (qrc != AMTX_OK) return qrc;
    }
    const int nh = m->n_heads;
    // ---- acoustic heads
    std::vector<float> c1w((size_t)nh * m->nf1 * m->in_channels * 9), c1s((size_t)nh * m->nf1);
    m->fuse_conv1 = m->gen_conv2 ? amtx_conv3x3_gen_can_fuse1(m->in_channels, m->nf1, m->nf2, m->planes) : (9 * m->in_channels <= 64);
    const size_t c1f_per = m->gen_conv2 ? amtx_conv1g_wfrag_elems(m->in_channels, m->nf1, m->planes) : amtx_conv1_wfrag_elems(m->in_channels, m->planes);
    std::vector<bf16_t> c1f(m->fuse_conv1 ? c1f_per * nh : 0);
    const size_t c2w_per = m->gen_conv2 ? amtx_conv3x3_gen_wfrag_elems(m->nf1, m->nf2, m->planes) : amtx_conv3x3_wfrag_elems(m->nf2, m->planes);
    const size_t c3w_per = m->gen_conv ? amtx_conv3x3_gen_wfrag_elems(m->nf2, m->nf3, m->planes) : amtx_conv3x3_wfrag_elems(m->nf3, m->planes);
    if (m->gen_conv && (!amtx_conv3x3_gen_ntc(m->nf1, m->nf2) || !amtx_conv3x3_gen_ntc(m->nf2, m->nf3))) {
        amtx_set_error("of_model: no convolution kernel for %d -> %d -> %d channels", m->nf1, m->nf2, m->nf3);
        return AMTX_ERR_UNSUPPORTED;
    }
    std::vector<bf16_t> c2w(c2w_per * nh), c3w(c3w_per * nh);
    const size_t c2x_per = amtx_conv3x3_wfrag_elems(m->nf2, m->planes);
    std::vector<bf16_t> c2x(m->x12m ? c2x_per * nh : 0);
    std::vector<float> c2s((size_t)nh * m->nf2), c3s((size_t)nh * m->nf3);
    std::vector<std::vector<float>> fcw(nh), fcb(nh);
    for (int h = 0; h < nh; ++h) {
        const std::string am = m->head_names[h] + ".0";
        std::vector<float> scale, shift;
        const float* w;
        int rc = fold_bn(m, am + ".layer1.0", am + ".layer1.1", m->nf1, scale, shift);
        if (rc != AMTX_OK) return rc;
        NEED(am + ".layer1.0.weight", (size_t)m->nf1 * m->in_channels * 9, w);
        for (int co = 0; co < m->nf1; ++co)
            for (int i = 0; i < m->in_channels * 9; ++i)
                c1w[((size_t)h * m->nf1 + co) * m->in_channels * 9 + i] = w[(size_t)co * m->in_channels * 9 + i] * scale[co];
        memcpy(c1s.data() + (size_t)h * m->nf1, shift.data(), sizeof(float) * m->nf1);
        if (m->fuse_conv1 && m->gen_conv2) (m->f16 ? amtx_conv1g_pack_host_f16 : amtx_conv1g_pack_host)(w, scale.data(), m->in_channels, m->nf1, m->planes, c1f.data() + c1f_per * h);
        else if (m->fuse_conv1) (m->f16 ? amtx_conv1_pack_host_f16 : amtx_conv1_pack_host)(w, scale.data(), m->in_channels, m->planes, c1f.data() + c1f_per * h);

        rc = fold_bn(m, am + ".layer2.0", am + ".layer2.1", m->nf2, scale, shift);
        if (rc != AMTX_OK) return rc;
        NEED(am + ".layer2.0.weight", (size_t)m->nf2 * m->nf1 * 9, w);
        if (m->gen_conv2) (m->f16 ? amtx_conv3x3_gen_pack_host_f16 : amtx_conv3x3_gen_pack_host)(w, scale.data(), m->nf1, m->nf2, m->planes, c2w.data() + c2w_per * h);
        else (m->f16 ? amtx_conv3x3_pack_host_f16 : amtx_conv3x3_pack_host)(w, scale.data(), m->nf2, m->planes, c2w.data() + c2w_per * h);
        if (m->x12m) amtx_conv3x3_pack_host(w, scale.data(), m->nf2, m->planes, c2x.data() + c2x_per * h);
        memcpy(c2s.data() + (size_t)h * m->nf2, shift.data(), sizeof(float) * m->nf2);

        rc = fold_bn(m, am + ".layer3.0", am + ".layer3.1", m->nf3, scale, shift);
        if (rc != AMTX_OK) return rc;
        NEED(am + ".layer3.0.weight", (size_t)m->nf3 * m->nf2 * 9, w);
        if (m->gen_conv) (m->f16 ? amtx_conv3x3_gen_pack_host_f16 : amtx_conv3x3_gen_pack_host)(w, scale.data(), m->nf2, m->nf3, m->planes, c3w.data() + c3w_per * h);
        else (m->f16 ? amtx_conv3x3_pack_host_f16 : amtx_conv3x3_pack_host)(w, scale.data(), m->nf3, m->planes, c3w.data() + c3w_per * h);
        memcpy(c3s.data() + (size_t)h * m->nf3, shift.data(), sizeof(float) * m->nf3);

        // fc1: reference column index c*fq + f  ->  ours f*nf3 + c
        const float* fb;
        NEED(am + ".fc1.0.weight", (size_t)m->dim_am * m->kfc, w);
        NEED(am + ".fc1.0.bias", (size_t)m->dim_am, fb);
        fcw[h].assign((size_t)m->dim_am * m->kfc_pad, 0.0f);     // columns kfc .. kfc_pad stay zero
        for (int n = 0; n < m->dim_am; ++n)
            for (int c = 0; c < m->nf3; ++c)
                for (int f = 0; f < m->fq; ++f)
                    fcw[h][(size_t)n * m->kfc_pad + (size_t)f * m->nf3 + c] = w[(size_t)n * m->kfc + (size_t)c * m->fq + f];
        fcb[h].assign(fb, fb + m->dim_am);
    }
    int rc;
    if ((rc = m->conv1_w.upload(c1w.data(), c1w.size() * 4)) != AMTX_OK) return rc;
    if ((rc = m->conv1_s.upload(c1s.data(), c1s.size() * 4)) != AMTX_OK) return rc;
    if (m->fuse_conv1 && (rc = m->conv1_frag.upload(c1f.data(), c1f.size() * 2)) != AMTX_OK) return rc;
    if ((rc = m->conv2_w.upload(c2w.data(), c2w.size() * 2)) != AMTX_OK) return rc;
    if (m->x12m && (rc = m->conv2_wx.upload(c2x.data(), c2x.size() * 2)) != AMTX_OK) return rc;
    if ((rc = m->conv2_s.upload(c2s.data(), c2s.size() * 4)) != AMTX_OK) return rc;
    if ((rc = m->conv3_w.upload(c3w.data(), c3w.size() * 2)) != AMTX_OK) return rc;
    if ((rc = m->conv3_s.upload(c3s.data(), c3s.size() * 4)) != AMTX_OK) return rc;
    {
        // fc1 of the RECURRENT heads only.  The pitch head's fc1 feeds its LogisticBank directly (AcousticModel.fc1 is Linear +
        // Dropout, no activation: onsetsframes.py:422-427, then models/common.py:539), so in eval mode the two Linear layers are one:
        //     logits = W_out (W_fc1 a + b_fc1) + b_out = (W_out W_fc1) a + (W_out b_fc1 + b_out)
        // folded here in double precision (the same kind of weight folding as the BatchNorms above): a K = kfc, N = n_out GEMM
        // replaces a K = kfc, N = dim_am one plus a K = dim_am, N = n_out one, and the dim_am-wide activation never exists.
        std::vector<std::vector<float>> fcw_rec(fcw.begin(), fcw.begin() + m->n_rec), fcb_rec(fcb.begin(), fcb.begin() + m->n_rec);
        if (m->n_rec > 0 && (rc = pack_linear_groups(m, m->fc1, fcw_rec, fcb_rec, m->dim_am, m->kfc_pad)) != AMTX_OK) return rc;
    }

    // ---- recurrent heads (onset, offset): LSTM + LogisticBank
    {
        std::vector<std::vector<float>> wih(m->n_rec), bih(m->n_rec), wo(m->n_rec), bo(m->n_rec);
        std::vector<bf16_t> hh_all;
        for (int r = 0; r < m->n_rec; ++r) {
            std::vector<bf16_t> hh;
            if ((rc = pack_lstm(m, m->head_names[r] + ".1", m->dim_am, wih[r], bih[r], hh)) != AMTX_OK) return rc;
            hh_all.insert(hh_all.end(), hh.begin(), hh.end());
            const float *w, *b;
            NEED(m->head_names[r] + ".2.output_layer.weight", (size_t)m->n_out * m->dim_lm, w);
            NEED(m->head_names[r] + ".2.output_layer.bias", (size_t)m->n_out, b);
            wo[r].assign(w, w + (size_t)m->n_out * m->dim_lm);
            bo[r].assign(b, b + m->n_out);
        }
        if ((rc = pack_linear_groups(m, m->rec_ih, wih, bih, m->xw, m->dim_am)) != AMTX_OK) return rc;
        if ((rc = m->rec_hh.upload(hh_all.data(), hh_all.size() * 2)) != AMTX_OK) return rc;
        if ((rc = pack_linear_groups(m, m->rec_out, wo, bo, m->n_out, m->dim_lm)) != AMTX_OK) return rc;
    }
    // ---- pitch head LogisticBank
    {
        const float *w, *b;
        NEED("pitch_head.1.output_layer.weight", (size_t)m->n_out * m->dim_am, w);
        NEED("pitch_head.1.output_layer.bias", (size_t)m->n_out, b);
        const std::vector<float>& fw = fcw[nh - 1];   // (dim_am, kfc_pad), columns already in the engine's (freq, channel) order
        const std::vector<float>& fb1 = fcb[nh - 1];
        std::vector<std::vector<float>> W(1), Bv(1);
        W[0].assign((size_t)m->n_out * m->kfc_pad, 0.0f);
        Bv[0].assign(m->n_out, 0.0f);
        // n_out x dim_am x kfc double-precision multiply-adds (164 M at model_complexity 2): output rows dealt to a few host threads, each
        // row summed in the same order as before (the result does not depend on the thread count)
        auto fold_rows = [&](int o0, int o1) {
            std::vector<double> rowacc(m->kfc_pad);
            for (int o = o0; o < o1; ++o) {
                std::fill(rowacc.begin(), rowacc.end(), 0.0);
                double bacc = b[o];
                for (int j = 0; j < m->dim_am; ++j) {
                    const double wo = w[(size_t)o * m->dim_am + j];
                    const float* frow = fw.data() + (size_t)j * m->kfc_pad;
                    for (int k = 0; k < m->kfc_pad; ++k) rowacc[k] += wo * frow[k];
                    bacc += wo * fb1[j];
                }
                for (int k = 0; k < m->kfc_pad; ++k) W[0][(size_t)o * m->kfc_pad + k] = (float)rowacc[k];
                Bv[0][o] = (float)bacc;
            }
        };
        const int nthreads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        const int per = (m->n_out + nthreads - 1) / nthreads;
        for (int t = 1; t < nthreads; ++t)
            if (t * per < m->n_out) pool.emplace_back(fold_rows, t * per, std::min(m->n_out, (t + 1) * per));
        fold_rows(0, std::min(m->n_out, per));
        for (auto& th : pool) th.join();
        if ((rc = pack_linear_groups(m, m->pitch_out, W, Bv, m->n_out, m->kfc_pad)) != AMTX_OK) return rc;
    }
    // ---- adjoin: LSTM over the joint logits + LogisticBank
    {
        std::vector<std::vector<float>> wih(1), bih(1);
        std::vector<bf16_t> hh;
        if ((rc = pack_lstm(m, "adjoin.0", m->dim_aj, wih[0], bih[0], hh)) != AMTX_OK) return rc;
        if ((rc = pack_linear_groups(m, m->adj_ih, wih, bih, m->xw, m->dim_aj)) != AMTX_OK) return rc;
        if ((rc = m->adj_hh.upload(hh.data(), hh.size() * 2)) != AMTX_OK) return rc;
        const float *w, *b;
        NEED("adjoin.1.output_layer.weight", (size_t)m->n_out * m->dim_lm, w);
        NEED("adjoin.1.output_layer.bias", (size_t)m->n_out, b);
        std::vector<std::vector<float>> W{std::vector<float>(w, w + (size_t)m->n_out * m->dim_lm)}, Bv{std::vector<float>(b, b + m->n_out)};
        if ((rc = pack_linear_groups(m, m->adj_out, W, Bv, m->n_out, m->dim_lm)) != AMTX_OK) return rc;
    }
    m->tensors.clear();
    m->finalized = true;
    m->packed_once = true;
    return AMTX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight RE-SYNC without leaving the GPU.  After one host-side amtx_of_model_finalize (which sizes and allocates every packed
// buffer), later weight versions can be handed over as device pointers under the same state_dict names and packed by the kernels of
// pack.hip -- the host packers' arithmetic and layouts, bit for bit.  Built for every configuration the engine runs (the engine
// validates at every checkpoint of train.py, amt_tools/train.py:183-189): model_complexity 2, 3 and 4, one or several input channels,
// any precision -- except a multi-channel first conv on conv.hip's kernel (only reachable with AMTX_NO_CONVG_MC2), which answers
// AMTX_ERR_UNSUPPORTED and keeps the host path.
extern "C" int amtx_of_model_set_tensor_device(amtx_of_model* m, const char* name, const float* device_data, int64_t numel) {
    AMTX_REQUIRE(m && name && device_data && numel > 0, "amtx_of_model_set_tensor_device: bad argument");
    m->dev_tensors[name] = std::make_pair(device_data, numel);
    return AMTX_OK;
}

namespace {
int need_dev(const amtx_of_model* m, const std::string& name, size_t numel, const float** out) {
    auto it = m->dev_tensors.find(name);
    if (it == m->dev_tensors.end()) {
        amtx_set_error("of_model: device tensor '%s' was not provided", name.c_str());
        return AMTX_ERR_ARG;
    }
    if ((size_t)it->second.second != numel) {
        amtx_set_error("of_model: device tensor '%s' has %lld elements, expected %zu", name.c_str(), (long long)it->second.second, numel);
        return AMTX_ERR_ARG;
    }
    *out = it->second.first;
    return AMTX_OK;
}
#define NEED_DEV(name, numel, ptr)                                     \
    do {                                                               \
        int _rc = need_dev(m, name, numel, &(ptr));                    \
        if (_rc != AMTX_OK) return _rc;                                \
    } while (0)
#define PACK_TRY(expr)                                                 \
    do {                                                               \
        int _rc = (expr);                                              \
        if (_rc != AMTX_OK) return _rc;                                \
    } while (0)
}  // namespace

// `dry`: look up and size-check every tensor, launch nothing -- amtx_of_model_finalize_device runs this pass first, so a missing or
// mis-sized tensor is reported before a single packed buffer has been touched (the buffers never end up half new, half old).
static int finalize_device_pass(amtx_of_model* m, hipStream_t s, const bool dry);

extern "C" int amtx_of_model_finalize_device(amtx_of_model* m, void* stream_) {
    AMTX_REQUIRE(m, "amtx_of_model_finalize_device: null model");
    struct Clear {                             // the borrowed device pointers are dropped on EVERY exit: the caller may free them afterwards
        amtx_of_model* m;
        ~Clear() { m->dev_tensors.clear(); }
    } clear{m};
    AMTX_REQUIRE(m->finalized, "amtx_of_model_finalize_device: the first sync goes through amtx_of_model_finalize (it allocates the packed buffers)");
    if (!m->gen_conv2 && m->in_channels != 1) {
        amtx_set_error("amtx_of_model_finalize_device: no device packer for conv.hip's multi-channel first conv; use amtx_of_model_finalize");
        return AMTX_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream_;
    int rc = finalize_device_pass(m, s, true);
    if (rc != AMTX_OK) return rc;
    if ((rc = quiesce_before_resync(m)) != AMTX_OK) return rc;
    rc = finalize_device_pass(m, s, false);
    if (rc != AMTX_OK) m->finalized = false;   // a launch failed half-way: the packed weights are no version at all, refuse to run on them
    return rc;
}

#undef PACK_TRY
#define PACK_TRY(expr)                                                 \
    do {                                                               \
        if (!dry) {                                                    \
            int _rc = (expr);                                          \
            if (_rc != AMTX_OK) return _rc;                            \
        }                                                              \
    } while (0)
#define PACK_COPY(dst, src, bytes)                                                                         \
    do {                                                                                                   \
        if (!dry) AMTX_CHECK_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToDevice, s));       \
    } while (0)

static int finalize_device_pass(amtx_of_model* m, hipStream_t s, const bool dry) {
    const int nh = m->n_heads, pl = m->planes, H = m->hid, G = 4 * H;
    const bool f16 = m->f16;
    auto pack_conv = [f16](const float* w, const float* sc, int c_out, int planes, bf16_t* out, hipStream_t st) {
        return f16 ? amtx_pack_conv3x3_dev_f16(w, sc, c_out, planes, out, st) : amtx_pack_conv3x3_dev(w, sc, c_out, planes, out, st);
    };
    auto pack_lin = [f16](const float* W, int64_t ldw, int N, int K, int planes, int n_pad, int k_pad, int row0, int rows, int pc, int pf, bf16_t* out,
                          hipStream_t st) {
        return f16 ? amtx_pack_linear_dev_f16(W, ldw, N, K, planes, n_pad, k_pad, row0, rows, pc, pf, out, st)
                   : amtx_pack_linear_dev(W, ldw, N, K, planes, n_pad, k_pad, row0, rows, pc, pf, out, st);
    };
    // scratch: scale[256] | folded pitch head (n_out x kfc_pad) | folded bias | backward LSTM fragments (written by the shared pack
    // kernel, not used by inference)
    const size_t hh_elems = amtx_bilstm_wfrag_elems_h(H, pl);
    const size_t sc_bytes = (size_t)(256 + (size_t)m->n_out * m->kfc_pad + m->n_out) * sizeof(float) + hh_elems * sizeof(bf16_t) + 256;
    AMTX_REQUIRE(m->nf3 <= 256, "amtx_of_model_finalize_device: internal: scale scratch");
    if (!m->pack_scratch.p || m->pack_scratch.bytes < sc_bytes) {
        m->pack_scratch.release();
        AMTX_CHECK_HIP(hipMalloc(&m->pack_scratch.p, sc_bytes));
        m->pack_scratch.bytes = sc_bytes;
    }
    float* scale = (float*)m->pack_scratch.p;
    float* wfold = scale + 256;
    float* bfold = wfold + (size_t)m->n_out * m->kfc_pad;
    bf16_t* hh_bwd = (bf16_t*)(((uintptr_t)(bfold + m->n_out) + 255) & ~(uintptr_t)255);

    const int ic = m->in_channels;
    const size_t c1f_per = m->gen_conv2 ? amtx_conv1g_wfrag_elems(ic, m->nf1, pl) : amtx_conv1_wfrag_elems(ic, pl);
    const size_t c2w_per = m->gen_conv2 ? amtx_conv3x3_gen_wfrag_elems(m->nf1, m->nf2, pl) : amtx_conv3x3_wfrag_elems(m->nf2, pl);
    const size_t c3w_per = m->gen_conv ? amtx_conv3x3_gen_wfrag_elems(m->nf2, m->nf3, pl) : amtx_conv3x3_wfrag_elems(m->nf3, pl);
    auto pack_conv_gen = [f16](const float* w, const float* sc, int c_in, int c_out, int planes, bf16_t* out, hipStream_t st) {
        const int ntc = amtx_conv3x3_gen_ntc(c_in, c_out);
        return f16 ? amtx_pack_conv_gen_dev_f16(w, sc, c_in, c_out, ntc, planes, out, st) : amtx_pack_conv_gen_dev(w, sc, c_in, c_out, ntc, planes, out, st);
    };
    const size_t fc_per = (size_t)m->fc1.n_pad * m->fc1.k_pad * pl;
    for (int h = 0; h < nh; ++h) {
        const std::string am = m->head_names[h] + ".0";
        const float *w, *cb, *g, *be, *mu, *var;
        // layer1: scale folded into the Toeplitz fragments, shift kept fp32 (conv1_s); the fp32 copy conv1_w feeds only the unfused first conv
        NEED_DEV(am + ".layer1.0.weight", (size_t)m->nf1 * ic * 9, w);
        NEED_DEV(am + ".layer1.0.bias", (size_t)m->nf1, cb); NEED_DEV(am + ".layer1.1.weight", (size_t)m->nf1, g); NEED_DEV(am + ".layer1.1.bias", (size_t)m->nf1, be);
        NEED_DEV(am + ".layer1.1.running_mean", (size_t)m->nf1, mu); NEED_DEV(am + ".layer1.1.running_var", (size_t)m->nf1, var);
        PACK_TRY(amtx_pack_bn_fold_dev(cb, g, be, mu, var, m->nf1, scale, (float*)m->conv1_s.p + (size_t)h * m->nf1, s));
        PACK_TRY(amtx_pack_scale_rows_dev(w, scale, m->nf1, ic * 9, (float*)m->conv1_w.p + (size_t)h * m->nf1 * ic * 9, s));
        if (m->fuse_conv1 && m->gen_conv2) PACK_TRY((f16 ? amtx_pack_conv1g_dev_f16 : amtx_pack_conv1g_dev)(w, scale, ic, m->nf1, pl, (bf16_t*)m->conv1_frag.p + c1f_per * h, s));
        else if (m->fuse_conv1) PACK_TRY((f16 ? amtx_pack_conv1_dev_f16 : amtx_pack_conv1_dev)(w, scale, pl, (bf16_t*)m->conv1_frag.p + c1f_per * h, s));
        NEED_DEV(am + ".layer2.0.weight", (size_t)m->nf2 * m->nf1 * 9, w);
        NEED_DEV(am + ".layer2.0.bias", (size_t)m->nf2, cb); NEED_DEV(am + ".layer2.1.weight", (size_t)m->nf2, g); NEED_DEV(am + ".layer2.1.bias", (size_t)m->nf2, be);
        NEED_DEV(am + ".layer2.1.running_mean", (size_t)m->nf2, mu); NEED_DEV(am + ".layer2.1.running_var", (size_t)m->nf2, var);
        PACK_TRY(amtx_pack_bn_fold_dev(cb, g, be, mu, var, m->nf2, scale, (float*)m->conv2_s.p + (size_t)h * m->nf2, s));
        if (m->gen_conv2) PACK_TRY(pack_conv_gen(w, scale, m->nf1, m->nf2, pl, (bf16_t*)m->conv2_w.p + c2w_per * h, s));
        else PACK_TRY(pack_conv(w, scale, m->nf2, pl, (bf16_t*)m->conv2_w.p + c2w_per * h, s));
        if (m->x12m) PACK_TRY(pack_conv(w, scale, m->nf2, pl, (bf16_t*)m->conv2_wx.p + (size_t)amtx_conv3x3_wfrag_elems(m->nf2, pl) * h, s));
        NEED_DEV(am + ".layer3.0.weight", (size_t)m->nf3 * m->nf2 * 9, w);
        NEED_DEV(am + ".layer3.0.bias", (size_t)m->nf3, cb); NEED_DEV(am + ".layer3.1.weight", (size_t)m->nf3, g); NEED_DEV(am + ".layer3.1.bias", (size_t)m->nf3, be);
        NEED_DEV(am + ".layer3.1.running_mean", (size_t)m->nf3, mu); NEED_DEV(am + ".layer3.1.running_var", (size_t)m->nf3, var);
        PACK_TRY(amtx_pack_bn_fold_dev(cb, g, be, mu, var, m->nf3, scale, (float*)m->conv3_s.p + (size_t)h * m->nf3, s));
        if (m->gen_conv) PACK_TRY(pack_conv_gen(w, scale, m->nf2, m->nf3, pl, (bf16_t*)m->conv3_w.p + c3w_per * h, s));
        else PACK_TRY(pack_conv(w, scale, m->nf3, pl, (bf16_t*)m->conv3_w.p + c3w_per * h, s));
        // fc1 of the recurrent heads, columns permuted (channel, freq) -> (freq, channel)
        if (h < m->n_rec) {
            const float* fb;
            NEED_DEV(am + ".fc1.0.weight", (size_t)m->dim_am * m->kfc, w);
            NEED_DEV(am + ".fc1.0.bias", (size_t)m->dim_am, fb);
            PACK_TRY(pack_lin(w, m->kfc, m->dim_am, m->kfc, pl, m->fc1.n_pad, m->fc1.k_pad, 0, m->fc1.n_pad, m->nf3, m->fq, (bf16_t*)m->fc1.w.p + fc_per * h, s));
            PACK_COPY((float*)m->fc1.b.p + (size_t)h * m->dim_am, fb, sizeof(float) * m->dim_am);
        }
    }
    // LSTM + LogisticBank of a recurrent stage: input projection rows [fwd | reverse], merged biases, W_hh fragments, output layer
    auto pack_rec = [&](const std::string& lstm, const std::string& bank, int dim_in, LinearPack& ih, DevBuf& hh, LinearPack& outp, int grp) -> int {
        const float *wif, *wib, *whf, *whb, *bif, *bib, *bhf, *bhb, *wo, *bo;
        const std::string p = lstm + ".mlm.";
        NEED_DEV(p + "weight_ih_l0", (size_t)G * dim_in, wif); NEED_DEV(p + "weight_ih_l0_reverse", (size_t)G * dim_in, wib);
        NEED_DEV(p + "weight_hh_l0", (size_t)G * H, whf); NEED_DEV(p + "weight_hh_l0_reverse", (size_t)G * H, whb);
        NEED_DEV(p + "bias_ih_l0", (size_t)G, bif); NEED_DEV(p + "bias_ih_l0_reverse", (size_t)G, bib);
        NEED_DEV(p + "bias_hh_l0", (size_t)G, bhf); NEED_DEV(p + "bias_hh_l0_reverse", (size_t)G, bhb);
        bf16_t* ihw = (bf16_t*)ih.w.p + (size_t)ih.n_pad * ih.k_pad * pl * grp;
        PACK_TRY(pack_lin(wif, dim_in, G, dim_in, pl, ih.n_pad, ih.k_pad, 0, G, 0, 0, ihw, s));
        PACK_TRY(pack_lin(wib, dim_in, G, dim_in, pl, ih.n_pad, ih.k_pad, G, ih.n_pad - G, 0, 0, ihw, s));
        float* ihb = (float*)ih.b.p + (size_t)ih.N * grp;
        PACK_TRY(amtx_pack_vec_add_dev(bif, bhf, G, ihb, s));
        PACK_TRY(amtx_pack_vec_add_dev(bib, bhb, G, ihb + G, s));
        PACK_TRY((f16 ? amtx_launch_bilstm_pack_dev_h_f16 : amtx_launch_bilstm_pack_dev_h)(whf, whb, H, pl, (bf16_t*)hh.p + hh_elems * grp, hh_bwd, s));
        NEED_DEV(bank + ".output_layer.weight", (size_t)m->n_out * m->dim_lm, wo);
        NEED_DEV(bank + ".output_layer.bias", (size_t)m->n_out, bo);
        PACK_TRY(pack_lin(wo, m->dim_lm, m->n_out, m->dim_lm, pl, outp.n_pad, outp.k_pad, 0, outp.n_pad, 0, 0,
                          (bf16_t*)outp.w.p + (size_t)outp.n_pad * outp.k_pad * pl * grp, s));
        PACK_COPY((float*)outp.b.p + (size_t)outp.N * grp, bo, sizeof(float) * m->n_out);
        return AMTX_OK;
    };
    for (int r = 0; r < m->n_rec; ++r) {
        const int prc = pack_rec(m->head_names[r] + ".1", m->head_names[r] + ".2", m->dim_am, m->rec_ih, m->rec_hh, m->rec_out, r);
        if (prc != AMTX_OK) return prc;
    }
    {
        const int prc = pack_rec("adjoin.0", "adjoin.1", m->dim_aj, m->adj_ih, m->adj_hh, m->adj_out, 0);
        if (prc != AMTX_OK) return prc;
    }
    // pitch head: LogisticBank folded into fc1 in double precision, then packed like any Linear layer
    {
        const float *wo, *bo, *w1, *b1;
        NEED_DEV("pitch_head.1.output_layer.weight", (size_t)m->n_out * m->dim_am, wo);
        NEED_DEV("pitch_head.1.output_layer.bias", (size_t)m->n_out, bo);
        NEED_DEV("pitch_head.0.fc1.0.weight", (size_t)m->dim_am * m->kfc, w1);
        NEED_DEV("pitch_head.0.fc1.0.bias", (size_t)m->dim_am, b1);
        PACK_TRY(amtx_pack_head_fold_dev(wo, w1, b1, bo, m->n_out, m->dim_am, m->kfc, m->kfc_pad, m->nf3, m->fq, wfold, bfold, s));
        PACK_TRY(pack_lin(wfold, m->kfc_pad, m->n_out, m->kfc_pad, pl, m->pitch_out.n_pad, m->pitch_out.k_pad, 0, m->pitch_out.n_pad, 0, 0,
                          (bf16_t*)m->pitch_out.w.p, s));
        PACK_COPY(m->pitch_out.b.p, bfold, sizeof(float) * m->n_out);
    }
    return AMTX_OK;
}

extern "C" size_t amtx_of_workspace_bytes(const amtx_of_model* m, int batch, int num_frames) {
    if (!m || batch <= 0 || num_frames <= 0) return 0;
    return carve(m, batch, num_frames, nullptr).total;
}

static GemmArgs gemm_args(const void* A, int64_t lda, int a_type, const LinearPack& lp, int planes, void* C, int64_t ldc, int c_type,
                          int64_t M, int groups, int64_t a_gs, int64_t c_gs) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.a_type = a_type;
    g.W = (const bf16_t*)lp.w.p; g.n_pad = lp.n_pad; g.k_pad = lp.k_pad; g.planes = planes;
    g.bias = (const float*)lp.b.p;
    g.C = C; g.ldc = ldc; g.c_type = c_type;
    g.M = M; g.N = lp.N; g.K = lp.K;
    g.groups = groups; g.a_gs = a_gs; g.w_gs = (int64_t)lp.n_pad * lp.k_pad * planes; g.bias_gs = lp.N; g.c_gs = c_gs;
    return g;
}

extern "C" int amtx_of_fuses_db_scale(const amtx_of_model* m);

extern "C" int amtx_of_takes_feats16(const amtx_of_model* m);

// clip_max != null: `feats` are raw power values, dB-scaled by the conv kernel while it stages them (amtx_of_forward_power)
// feats16 != null: the features as [B][T][F][8] 16-bit channels-last instead of `feats` (amtx_of_forward_feats16)
static int of_forward_impl(const amtx_of_model* m, const float* feats, const void* feats16, int64_t stride_b, int64_t stride_c, int64_t stride_t,
                           int64_t stride_f, const float* clip_max, const float* ref, int batch, int num_frames, void* workspace,
                           size_t workspace_bytes, float* out_onsets, float* out_multi_pitch, float* logits_onsets,
                           float* logits_multi_pitch, float* logits_pitch_head, void* stream_) {
    AMTX_REQUIRE(m && m->finalized, "amtx_of_forward: model not finalized");
    AMTX_REQUIRE(!clip_max || amtx_of_fuses_db_scale(m), "amtx_of_forward_power: this model does not stage its features in the conv kernel");
    AMTX_REQUIRE(!feats16 || amtx_of_takes_feats16(m), "amtx_of_forward_feats16: this model does not stage 16-bit channels-last features");
    AMTX_REQUIRE((feats || feats16) && workspace, "amtx_of_forward: null pointer");
    AMTX_REQUIRE(batch > 0 && num_frames > 0, "amtx_of_forward: bad batch/num_frames");
    const int B = batch, T = num_frames;
    Workspace w = carve(m, B, T, (char*)workspace);
    AMTX_REQUIRE(workspace_bytes >= w.total, "amtx_of_forward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    AMTX_REQUIRE(((uintptr_t)workspace % 256) == 0, "amtx_of_forward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream_;
    const int64_t BT = (int64_t)B * T;
    const int F = m->dim_in, F2 = F / 2, at = m->act_type, pl = m->planes;
    int rc;
    // the kernels of this model's 16-bit operand format (bf16, or IEEE half: the second build of conv / convf / gemm / lstm.hip)
    const bool f16 = m->f16;
    auto launch_gemm = [f16](const GemmArgs& ga, hipStream_t st) { return f16 ? amtx_launch_gemm_f16(ga, st) : amtx_launch_gemm(ga, st); };
    auto launch_conv = [f16](const ConvArgs& ca, hipStream_t st) { return f16 ? amtx_launch_conv3x3_f16(ca, st) : amtx_launch_conv3x3(ca, st); };
    auto launch_convg = [f16](const ConvArgs& ca, int ci, hipStream_t st) { return f16 ? amtx_launch_conv3x3_gen_f16(ca, ci, st) : amtx_launch_conv3x3_gen(ca, ci, st); };
    auto launch_bilstm = [f16](const LstmArgs& la, hipStream_t st) { return f16 ? amtx_launch_bilstm_f16(la, st) : amtx_launch_bilstm(la, st); };
    std::vector<hipEvent_t>* evs = nullptr;
    if (m->prof) {
        m->prof_events.emplace_back();
        evs = &m->prof_events.back();
    }
    auto mark = [&]() {
        if (!evs) return;
        hipEvent_t e;
        if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); evs->push_back(e); }
    };
    mark();

    Conv1Args c1;
    c1.in = feats; c1.stride_b = stride_b; c1.stride_c = stride_c; c1.stride_t = stride_t; c1.stride_f = stride_f;
    c1.w = (const float*)m->conv1_w.p; c1.shift = (const float*)m->conv1_s.p; c1.out = w.a1; c1.out_type = at;
    c1.B = B; c1.T = T; c1.F = F; c1.c_in = m->in_channels; c1.c_out = m->nf1;
    c1.groups = m->n_heads; c1.w_gs = (int64_t)m->nf1 * m->in_channels * 9; c1.shift_gs = m->nf1; c1.out_gs = BT * F * m->nf1;
    if (!m->fuse_conv1 && (rc = (f16 ? amtx_launch_conv1_f16 : amtx_launch_conv1)(c1, s)) != AMTX_OK) return rc;
    mark();

    ConvArgs c2;
    c2.in = w.a1; c2.in_type = at; c2.wfrag = (const bf16_t*)m->conv2_w.p; c2.planes = pl; c2.shift = (const float*)m->conv2_s.p;
    c2.out = w.a2; c2.out_type = at; c2.B = B; c2.T = T; c2.F = F; c2.c_out = m->nf2;
    c2.groups = m->n_heads; c2.in_gs = BT * F * m->nf1; c2.shift_gs = m->nf2;
    c2.w_gs = (int64_t)(m->gen_conv2 ? amtx_conv3x3_gen_wfrag_elems(m->nf1, m->nf2, pl) : amtx_conv3x3_wfrag_elems(m->nf2, pl));
    c2.out_gs = BT * F2 * m->nf2;
    const bool sp = m->split_acts;
    // two-plane maps: plane stride = all groups of one plane
    const int64_t a2_split = BT * F2 * m->nf2 * m->n_heads, a3_split = BT * m->kfc_pad * m->n_heads, e_split = BT * m->dim_am * m->n_heads;
    if (sp) { c2.out_type = AMTX_T_SPLIT; c2.out_split = a2_split; }
    if (m->fuse_conv1) {   // Conv(c_in->32)+BN+ReLU computed inside the conv2 kernel; a1 is never materialised
        c2.in = nullptr;
        c2.feats = feats; c2.f_stride_b = stride_b; c2.f_stride_c = stride_c; c2.f_stride_t = stride_t; c2.f_stride_f = stride_f;
        c2.c_in = m->in_channels; c2.w1frag = (const bf16_t*)m->conv1_frag.p; c2.shift1 = (const float*)m->conv1_s.p;
        c2.w1_gs = (int64_t)(m->gen_conv2 ? amtx_conv1g_wfrag_elems(m->in_channels, m->nf1, pl) : amtx_conv1_wfrag_elems(m->in_channels, pl));
        c2.f_clip_max = clip_max; c2.f_ref = ref;
        if (feats16) { c2.feats = nullptr; c2.feats16 = feats16; }
    }
    // two-plane 16-bit features into a two-plane model: conv1 + conv2 on convx.hip's layer-specialised kernel (layer2's weights in conv.hip's order)
    const bool x12m_now = feats16 && m->x12m;
    if (x12m_now) {
        c2.in_split = BT * F * 8;                              // the lo plane of the features: right behind the hi plane
        c2.wfrag = (const bf16_t*)m->conv2_wx.p;
        c2.w_gs = (int64_t)amtx_conv3x3_wfrag_elems(m->nf2, pl);
    }
    const bool fused_stack = m->fuse_stack && amtx_conv_stack_fused_ok(B, T, F, m->n_heads);
    // the fused stack writes its output in planes of 64 channels per pooled frequency column ([F / 4][B T][64]): a k-tile of the two GEMMs
    // that read it (fc1, the folded pitch head) is then contiguous memory.  A/B switch: AMTX_OF_ROWMAJOR_A3=1
    static const bool rowmajor_a3 = getenv("AMTX_OF_ROWMAJOR_A3") != nullptr;
    const int64_t a3_plane = (fused_stack && !rowmajor_a3 && m->nf3 == 64 && m->kfc_pad == m->kfc) ? BT * 64 : 0;
    if (fused_stack) {
        // layer1 -> layer2 -> layer3 in one kernel: neither intermediate map reaches HBM (stage timer: all of it under conv2_pool)
        if ((rc = (f16 ? amtx_launch_conv_stack_f16 : amtx_launch_conv_stack)(c2, (const bf16_t*)m->conv3_w.p, (int64_t)amtx_conv3x3_wfrag_elems(m->nf3, pl), (const float*)m->conv3_s.p,
                                         w.a3, BT * m->kfc_pad, a3_plane, s)) != AMTX_OK) return rc;
    } else if (x12m_now) {
        if ((rc = amtx_launch_convx12(c2, s)) != AMTX_OK) return rc;
    } else if ((rc = m->gen_conv2 ? launch_convg(c2, m->nf1, s) : launch_conv(c2, s)) != AMTX_OK) return rc;
    mark();

    ConvArgs c3 = c2;
    c3.feats = nullptr; c3.feats16 = nullptr; c3.w1frag = nullptr; c3.shift1 = nullptr; c3.c_in = 0;
    c3.in = w.a2; c3.wfrag = (const bf16_t*)m->conv3_w.p; c3.shift = (const float*)m->conv3_s.p; c3.out = w.a3;
    c3.F = F2; c3.c_out = m->nf3; c3.in_gs = BT * F2 * m->nf2;
    c3.w_gs = (int64_t)(m->gen_conv ? amtx_conv3x3_gen_wfrag_elems(m->nf2, m->nf3, pl) : amtx_conv3x3_wfrag_elems(m->nf3, pl));
    c3.shift_gs = m->nf3; c3.out_gs = BT * m->kfc_pad;
    if (sp) { c3.in_type = AMTX_T_SPLIT; c3.in_split = a2_split; c3.out_type = AMTX_T_SPLIT; c3.out_split = a3_split; }
    if (m->kfc_pad != m->kfc) {
        // rows of a3 are padded to the DMA GEMM's k-tile: the pad columns meet zero weights, they only have to be finite
        AMTX_REQUIRE(m->gen_conv, "amtx_of_forward: internal: padded fc1 rows need the general conv kernel");
        c3.out_ts = m->kfc_pad;
        const size_t es = amtx_tsize(at);
        if ((rc = amtx_launch_zero_cols(w.a3 + (size_t)m->kfc * es, (int64_t)m->kfc_pad * es, (int)((m->kfc_pad - m->kfc) * es),
                                        BT * m->n_heads, s)) != AMTX_OK) return rc;
    }
    if (!fused_stack && (rc = m->gen_conv ? launch_convg(c3, m->nf2, s) : launch_conv(c3, s)) != AMTX_OK) return rc;
    mark();

    // fc1 of the recurrent heads (heads 0..n_rec-1 of a3); the pitch head's fc1 is folded into its output layer below
    const int at_d = sp ? AMTX_T_SPLIT : at;   // element type of the dense layers' activations
    GemmArgs g = gemm_args(w.a3, m->kfc_pad, at_d, m->fc1, pl, w.e, m->dim_am, at_d, BT, m->n_rec, BT * m->kfc_pad, BT * m->dim_am);
    g.a_plane = a3_plane; g.a_split = a3_split; g.c_split = e_split;
    if ((rc = launch_gemm(g, s)) != AMTX_OK) return rc;
    mark();

    // recurrent heads: heads 0..n_rec-1 of `e`
    g = gemm_args(w.e, m->dim_am, at_d, m->rec_ih, pl, w.xp, m->xw, at, BT, m->n_rec, BT * m->dim_am, BT * m->xw);
    g.a_split = e_split;
    if ((rc = launch_gemm(g, s)) != AMTX_OK) return rc;
    mark();
    LstmArgs l;
    l.xproj = w.xp; l.x_type = at; l.whh = (const bf16_t*)m->rec_hh.p; l.planes = pl; l.out = w.l1; l.out_type = at;
    l.B = B; l.T = T; l.groups = m->n_rec; l.x_gs = BT * m->xw; l.w_gs = (int64_t)amtx_bilstm_wfrag_elems_h(m->hid, pl); l.out_gs = BT * m->dim_lm;
    l.hidden = m->hid;
    if ((rc = launch_bilstm(l, s)) != AMTX_OK) return rc;
    mark();
    // LogisticBank of each recurrent head -> joint[:, r*n_out : (r+1)*n_out]; group stride of C = n_out columns
    g = gemm_args(w.l1, m->dim_lm, at, m->rec_out, pl, w.joint, m->dim_aj, AMTX_T_F32, BT, m->n_rec, BT * m->dim_lm, m->n_out);
    // piano rolls (LogisticBank.finalize_output with threshold 0.5) come out of the LogisticBank GEMMs' epilogues where that kernel has
    // one (bf16 mode); otherwise amtx_launch_pianoroll below reads the logits back
    static const bool no_roll_epi = getenv("AMTX_OF_NO_ROLL_EPILOGUE") != nullptr;     // A/B switch: separate pianoroll launches
    const bool roll_on = out_onsets && !no_roll_epi && amtx_gemm_has_roll_epilogue(g);
    if (roll_on) { g.roll_out = out_onsets; g.roll_T = T; g.roll_thr = 0.5f; g.roll_group = 0; }
    // bf16 mode: the refinement stage's input (the joint logits rounded to bf16, K zero-padded to the DMA GEMM's 64-deep k-tile) is
    // written by the same two epilogues instead of a conversion pass over the fp32 joint buffer; the fp32 joint logits themselves are
    // only written when something reads them (logit outputs, the offset head's probabilities, the modes without these epilogues)
    const int kp = (m->dim_aj + 63) / 64 * 64;
    GemmArgs gp = gemm_args(w.a3 + (size_t)(m->n_heads - 1) * BT * m->kfc_pad * (sp ? 2 : amtx_tsize(at)), m->kfc_pad, at_d, m->pitch_out, pl,
                            w.joint + (size_t)m->n_rec * m->n_out * sizeof(float), m->dim_aj, AMTX_T_F32, BT, 1, 0, 0);
    gp.a_plane = a3_plane; gp.a_split = a3_split;
    const bool copy_on = pl == 1 && !no_roll_epi && amtx_gemm_has_roll_epilogue(g) && amtx_gemm_has_roll_epilogue(gp) && m->n_out % 4 == 0 &&
                         (kp - m->dim_aj) % 4 == 0 && gp.N + (kp - m->dim_aj) <= gp.n_pad;
    if (copy_on) {
        g.copy16 = (bf16_t*)w.joint16; g.copy16_ld = kp; g.copy16_col0 = 0; g.copy16_gs = m->n_out; g.copy16_pad = 0;
        gp.copy16 = (bf16_t*)w.joint16; gp.copy16_ld = kp; gp.copy16_col0 = m->n_rec * m->n_out; gp.copy16_gs = 0; gp.copy16_pad = kp - m->dim_aj;
        if (roll_on && !logits_onsets && !logits_pitch_head && !m->has_offsets) { g.C = nullptr; gp.C = nullptr; }
    }
    if ((rc = launch_gemm(g, s)) != AMTX_OK) return rc;
    mark();
    // pitch head: (fc1 . LogisticBank) folded, straight from its conv3 map -> last n_out columns of joint
    if ((rc = launch_gemm(gp, s)) != AMTX_OK) return rc;
    mark();

    // adjoin
    if (pl == 1) {
        // bf16 mode: the joint logits rounded to bf16 (zero-padded to a 64-multiple K) feed the direct-to-LDS GEMM
        if (!copy_on && (rc = amtx_launch_cvt_pad_bf16((const float*)w.joint, m->dim_aj, m->dim_aj, (bf16_t*)w.joint16, kp, BT, s, f16)) != AMTX_OK) return rc;
        g = gemm_args(w.joint16, kp, AMTX_T_BF16, m->adj_ih, pl, w.xp2, m->xw, at, BT, 1, 0, 0);
        g.K = kp;
    } else if (sp) {
        // two-plane mode: the joint logits as two 16-bit planes, K zero-padded to whole 32-deep stages, feed the direct-to-LDS two-plane GEMM
        if ((rc = amtx_launch_cvt_split((const float*)w.joint, m->dim_aj, m->dim_aj, (bf16_t*)w.joint16, kp, BT * kp, BT, s)) != AMTX_OK) return rc;
        g = gemm_args(w.joint16, kp, AMTX_T_SPLIT, m->adj_ih, pl, w.xp2, m->xw, at, BT, 1, 0, 0);
        g.K = kp; g.a_split = BT * kp;
    } else {
        g = gemm_args(w.joint, m->dim_aj, AMTX_T_F32, m->adj_ih, pl, w.xp2, m->xw, at, BT, 1, 0, 0);
    }
    if ((rc = launch_gemm(g, s)) != AMTX_OK) return rc;
    mark();
    l.xproj = w.xp2; l.whh = (const bf16_t*)m->adj_hh.p; l.out = w.l2; l.groups = 1;
    if ((rc = launch_bilstm(l, s)) != AMTX_OK) return rc;
    mark();
    g = gemm_args(w.l2, m->dim_lm, at, m->adj_out, pl, w.mp, m->n_out, AMTX_T_F32, BT, 1, 0, 0);
    const bool roll_mp = out_multi_pitch && !no_roll_epi && amtx_gemm_has_roll_epilogue(g);
    if (roll_mp) {
        g.roll_out = out_multi_pitch; g.roll_T = T; g.roll_thr = 0.5f; g.roll_group = 0;
        if (!logits_multi_pitch) g.C = nullptr;            // nobody reads the refined logits then: only the roll is written
    }
    if ((rc = launch_gemm(g, s)) != AMTX_OK) return rc;
    mark();

    // piano rolls of the modes whose LogisticBank GEMM has no roll epilogue (x3)
    if (out_onsets && !roll_on && (rc = amtx_launch_pianoroll((const float*)w.joint, m->dim_aj, 0, B, T, m->n_out, 0.5f, out_onsets, s)) != AMTX_OK) return rc;
    if (out_multi_pitch && !roll_mp && (rc = amtx_launch_pianoroll((const float*)w.mp, m->n_out, 0, B, T, m->n_out, 0.5f, out_multi_pitch, s)) != AMTX_OK) return rc;

    mark();
    // optional raw logits, contiguous (B, T, n_out)
    const size_t row = (size_t)m->n_out * sizeof(float);
    if (logits_onsets)
        AMTX_CHECK_HIP(hipMemcpy2DAsync(logits_onsets, row, w.joint, (size_t)m->dim_aj * 4, row, BT, hipMemcpyDeviceToDevice, s));
    if (logits_pitch_head)
        AMTX_CHECK_HIP(hipMemcpy2DAsync(logits_pitch_head, row, w.joint + (size_t)m->n_rec * row, (size_t)m->dim_aj * 4, row, BT,
                                        hipMemcpyDeviceToDevice, s));
    if (logits_multi_pitch)
        AMTX_CHECK_HIP(hipMemcpyAsync(logits_multi_pitch, w.mp, row * BT, hipMemcpyDeviceToDevice, s));
    return AMTX_OK;
}

// 1 when amtx_of_forward_power applies: one input channel and the first conv fused into the C_out = 32 conv kernel (conv.hip, KS = 1)
extern "C" int amtx_of_fuses_db_scale(const amtx_of_model* m) {
    return m && m->finalized && m->fuse_conv1 && !m->gen_conv2 && m->in_channels == 1;
}

extern "C" int amtx_of_conv_stack_fused(const amtx_of_model* m, int batch, int num_frames) {
    return m && m->fuse_stack && amtx_conv_stack_fused_ok(batch, num_frames, m->dim_in, m->n_heads);
}

extern "C" int amtx_of_forward(const amtx_of_model* m, const float* feats, int64_t stride_b, int64_t stride_c, int64_t stride_t,
                               int64_t stride_f, int batch, int num_frames, void* workspace, size_t workspace_bytes,
                               float* out_onsets, float* out_multi_pitch, float* logits_onsets, float* logits_multi_pitch,
                               float* logits_pitch_head, void* stream_) {
    return of_forward_impl(m, feats, nullptr, stride_b, stride_c, stride_t, stride_f, nullptr, nullptr, batch, num_frames, workspace, workspace_bytes,
                           out_onsets, out_multi_pitch, logits_onsets, logits_multi_pitch, logits_pitch_head, stream_);
}

extern "C" int amtx_of_forward_power(const amtx_of_model* m, const float* power, int64_t stride_b, int64_t stride_t, int64_t stride_f,
                                     const float* clip_max, const float* ref, int batch, int num_frames, void* workspace,
                                     size_t workspace_bytes, float* out_onsets, float* out_multi_pitch, float* logits_onsets,
                                     float* logits_multi_pitch, float* logits_pitch_head, void* stream_) {
    AMTX_REQUIRE(clip_max, "amtx_of_forward_power: clip_max is null");
    return of_forward_impl(m, power, nullptr, stride_b, 0, stride_t, stride_f, clip_max, ref, batch, num_frames, workspace, workspace_bytes,
                           out_onsets, out_multi_pitch, logits_onsets, logits_multi_pitch, logits_pitch_head, stream_);
}

// 1 when amtx_of_forward_feats16 applies: 2 .. 8 input channels and the first conv fused tap-major into the general conv kernel's 32-channel
// pipelined variant (convg.hip: one-plane bf16 mode, model_complexity 2) -- the HCQT configuration (BASELINE config 3)
extern "C" int amtx_of_takes_feats16(const amtx_of_model* m) {
    if (!(m && m->finalized && m->fuse_conv1 && m->gen_conv2 && !m->f16)) return 0;
    if (m->x12m) return 2;     // two planes: [2][B][T][F][8], the lo plane B T F 8 elements behind the hi plane (amtx_cqt_forward16_split)
    return m->planes == 1 && m->nf1 == 32 && amtx_conv1g_tapk(m->in_channels, m->planes) && m->act_type == AMTX_T_BF16;
}

extern "C" int amtx_of_forward_feats16(const amtx_of_model* m, const void* feats16, int batch, int num_frames, void* workspace, size_t workspace_bytes,
                                       float* out_onsets, float* out_multi_pitch, float* logits_onsets, float* logits_multi_pitch,
                                       float* logits_pitch_head, void* stream_) {
    AMTX_REQUIRE(feats16, "amtx_of_forward_feats16: feats16 is null");
    return of_forward_impl(m, nullptr, feats16, 0, 0, 0, 0, nullptr, nullptr, batch, num_frames, workspace, workspace_bytes, out_onsets, out_multi_pitch,
                           logits_onsets, logits_multi_pitch, logits_pitch_head, stream_);
}

// OnsetsFrames2: the offset head's LogisticBank output of the LAST amtx_of_forward on this workspace
// (onsetsframes.py:256-261,323-325: finalize_output without a threshold = sigmoid probabilities, (B, n_out, T)).
extern "C" int amtx_of_offsets(const amtx_of_model* m, void* workspace, size_t workspace_bytes, int batch, int num_frames, float* out_offsets,
                               float* logits_offsets, void* stream_) {
    AMTX_REQUIRE(m && m->finalized, "amtx_of_offsets: model not finalized");
    AMTX_REQUIRE(m->has_offsets, "amtx_of_offsets: the model has no offset head");
    AMTX_REQUIRE(workspace && batch > 0 && num_frames > 0, "amtx_of_offsets: bad argument");
    Workspace w = carve(m, batch, num_frames, (char*)workspace);
    AMTX_REQUIRE(workspace_bytes >= w.total, "amtx_of_offsets: workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    const int64_t BT = (int64_t)batch * num_frames;
    int rc;
    if (out_offsets && (rc = amtx_launch_pianoroll((const float*)w.joint, m->dim_aj, m->n_out, batch, num_frames, m->n_out, -1.0f, out_offsets, s)) != AMTX_OK)
        return rc;
    const size_t row = (size_t)m->n_out * sizeof(float);
    if (logits_offsets)
        AMTX_CHECK_HIP(hipMemcpy2DAsync(logits_offsets, row, w.joint + row, (size_t)m->dim_aj * 4, row, BT, hipMemcpyDeviceToDevice, s));
    return AMTX_OK;
}

extern "C" int amtx_of_num_stages(void) { return ST_COUNT; }
extern "C" const char* amtx_of_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

extern "C" int amtx_of_profile_enable(amtx_of_model* m, int enable) {
    AMTX_REQUIRE(m, "amtx_of_profile_enable: null model");
    for (auto& v : m->prof_events)
        for (hipEvent_t e : v) (void)hipEventDestroy(e);
    m->prof_events.clear();
    m->prof = enable != 0;
    return AMTX_OK;
}

// Sum of per-stage durations (ms) over all forwards since profiling was enabled; waits for them to finish.
extern "C" int amtx_of_profile_read(amtx_of_model* m, double* stage_ms, int* num_forwards) {
    AMTX_REQUIRE(m && stage_ms && num_forwards, "amtx_of_profile_read: null argument");
    for (int i = 0; i < ST_COUNT; ++i) stage_ms[i] = 0.0;
    *num_forwards = 0;
    for (auto& v : m->prof_events) {
        if ((int)v.size() != ST_COUNT + 1) continue;
        AMTX_CHECK_HIP(hipEventSynchronize(v.back()));
        for (int i = 0; i < ST_COUNT; ++i) {
            float ms = 0.f;
            AMTX_CHECK_HIP(hipEventElapsedTime(&ms, v[i], v[i + 1]));
            stage_ms[i] += ms;
        }
        ++*num_forwards;
    }
    return AMTX_OK;
}
