// Op-level C ABI wrappers around the kernel launchers (include/amtx.h, "Op-level entry points").
#include "amtx_kernels.h"

extern "C" int64_t amtx_linear_packed_elems(int n, int k, int planes) {
    int n_pad, k_pad;
    amtx_gemm_pack_dims(n, k, &n_pad, &k_pad);
    return (int64_t)n_pad * k_pad * planes;
}

extern "C" int amtx_linear_pack(const float* host_w, int n, int k, int planes, uint16_t* host_out) {
    AMTX_REQUIRE(host_w && host_out && n > 0 && k > 0 && (planes == 1 || planes == 2), "amtx_linear_pack: bad argument");
    amtx_gemm_pack_host(host_w, k, n, k, planes, host_out);
    return AMTX_OK;
}

extern "C" int amtx_linear_fwd(const void* a, int64_t lda, int a_type, const uint16_t* w_packed, int planes, const float* bias, void* c,
                               int64_t ldc, int c_type, int64_t m, int n, int k, void* stream) {
    GemmArgs g;
    g.A = a; g.lda = lda; g.a_type = a_type; g.W = w_packed; g.planes = planes;
    amtx_gemm_pack_dims(n, k, &g.n_pad, &g.k_pad);
    g.bias = bias; g.C = c; g.ldc = ldc; g.c_type = c_type; g.M = m; g.N = n; g.K = k;
    g.groups = 1; g.a_gs = g.w_gs = g.bias_gs = g.c_gs = 0;
    return amtx_launch_gemm(g, (hipStream_t)stream);
}

extern "C" int amtx_split_planes(const float* src, int64_t ld_src, int n, uint16_t* dst, int ld_dst, int64_t split, int64_t rows, void* stream) {
    return amtx_launch_cvt_split(src, ld_src, n, dst, ld_dst, split, rows, (hipStream_t)stream);
}

extern "C" int amtx_linear_fwd_split(const void* a, int64_t lda, int64_t a_split, const uint16_t* w_packed, const float* bias, void* c, int64_t ldc,
                                     int c_type, int64_t c_split, int64_t m, int n, int k, void* stream) {
    GemmArgs g;
    g.A = a; g.lda = lda; g.a_type = AMTX_T_SPLIT; g.a_split = a_split; g.W = w_packed; g.planes = 2;
    amtx_gemm_pack_dims(n, k, &g.n_pad, &g.k_pad);
    g.bias = bias; g.C = c; g.ldc = ldc; g.c_type = c_type; g.c_split = c_split; g.M = m; g.N = n; g.K = k;
    g.groups = 1; g.a_gs = g.w_gs = g.bias_gs = g.c_gs = 0;
    return amtx_launch_gemm(g, (hipStream_t)stream);
}

extern "C" int64_t amtx_conv3x3_packed_elems(int c_out, int planes) { return (int64_t)amtx_conv3x3_wfrag_elems(c_out, planes); }

extern "C" int amtx_conv3x3_pack(const float* host_w, const float* host_scale, int c_out, int planes, uint16_t* host_out) {
    AMTX_REQUIRE(host_w && host_out && c_out % 16 == 0 && (planes == 1 || planes == 2), "amtx_conv3x3_pack: bad argument");
    amtx_conv3x3_pack_host(host_w, host_scale, c_out, planes, host_out);
    return AMTX_OK;
}

extern "C" int amtx_conv3x3_fwd(const void* in, int elem_type, const uint16_t* w_packed, int planes, const float* shift, void* out,
                                int batch, int num_frames, int num_bins, int c_out, void* stream) {
    ConvArgs a;
    a.in = in; a.in_type = elem_type; a.wfrag = w_packed; a.planes = planes; a.shift = shift; a.out = out; a.out_type = elem_type;
    a.B = batch; a.T = num_frames; a.F = num_bins; a.c_out = c_out;
    a.groups = 1; a.in_gs = a.w_gs = a.shift_gs = a.out_gs = 0;
    return amtx_launch_conv3x3(a, (hipStream_t)stream);
}

extern "C" int64_t amtx_conv3x3g_packed_elems(int c_in, int c_out, int planes) {
    if (c_in <= 0 || c_out <= 0 || c_in % 16 || c_out % 16 || !amtx_conv3x3_gen_ntc(c_in, c_out)) return 0;
    return (int64_t)amtx_conv3x3_gen_wfrag_elems(c_in, c_out, planes);
}

extern "C" int amtx_conv3x3g_pack(const float* host_w, const float* host_scale, int c_in, int c_out, int planes, uint16_t* host_out) {
    AMTX_REQUIRE(host_w && host_out && (planes == 1 || planes == 2), "amtx_conv3x3g_pack: bad argument");
    AMTX_REQUIRE(amtx_conv3x3g_packed_elems(c_in, c_out, planes) > 0, "amtx_conv3x3g_pack: channel counts %d -> %d are not built", c_in, c_out);
    amtx_conv3x3_gen_pack_host(host_w, host_scale, c_in, c_out, planes, host_out);
    return AMTX_OK;
}

extern "C" int amtx_conv3x3g_fwd(const void* in, int elem_type, const uint16_t* w_packed, int planes, const float* shift, void* out,
                                 int batch, int num_frames, int num_bins, int c_in, int c_out, void* stream) {
    ConvArgs a;
    a.in = in; a.in_type = elem_type; a.wfrag = w_packed; a.planes = planes; a.shift = shift; a.out = out; a.out_type = elem_type;
    a.B = batch; a.T = num_frames; a.F = num_bins; a.c_out = c_out;
    a.groups = 1; a.in_gs = a.w_gs = a.shift_gs = a.out_gs = 0;
    return amtx_launch_conv3x3_gen(a, c_in, (hipStream_t)stream);
}

extern "C" int amtx_conv1_fwd(const float* feats, int64_t stride_b, int64_t stride_c, int64_t stride_t, int64_t stride_f, const float* w,
                              const float* shift, void* out, int out_type, int batch, int num_frames, int num_bins, int c_in, int c_out,
                              void* stream) {
    Conv1Args a;
    a.in = feats; a.stride_b = stride_b; a.stride_c = stride_c; a.stride_t = stride_t; a.stride_f = stride_f;
    a.w = w; a.shift = shift; a.out = out; a.out_type = out_type;
    a.B = batch; a.T = num_frames; a.F = num_bins; a.c_in = c_in; a.c_out = c_out;
    a.groups = 1; a.w_gs = a.shift_gs = a.out_gs = 0;
    return amtx_launch_conv1(a, (hipStream_t)stream);
}

extern "C" int64_t amtx_bilstm_packed_elems(int planes) { return (int64_t)amtx_bilstm_wfrag_elems(planes); }

extern "C" int amtx_bilstm_pack(const float* host_whh_fwd, const float* host_whh_bwd, int planes, uint16_t* host_out) {
    AMTX_REQUIRE(host_whh_fwd && host_whh_bwd && host_out && (planes == 1 || planes == 2), "amtx_bilstm_pack: bad argument");
    amtx_bilstm_pack_host(host_whh_fwd, host_whh_bwd, planes, host_out);
    return AMTX_OK;
}

extern "C" int amtx_bilstm_fwd(const void* xproj, const uint16_t* whh_packed, int planes, int elem_type, void* out, int batch,
                               int num_frames, void* stream) {
    LstmArgs l;
    l.xproj = xproj; l.x_type = elem_type; l.whh = whh_packed; l.planes = planes; l.out = out; l.out_type = elem_type;
    l.B = batch; l.T = num_frames; l.groups = 1; l.x_gs = l.w_gs = l.out_gs = 0;
    return amtx_launch_bilstm(l, (hipStream_t)stream);
}

extern "C" int64_t amtx_bilstm_h_packed_elems(int hidden, int planes) {
    if (hidden != 128 && hidden != 256 && hidden != 384 && hidden != 512) return 0;
    return (int64_t)amtx_bilstm_wfrag_elems_h(hidden, planes);
}

extern "C" int amtx_bilstm_h_pack(const float* host_whh_fwd, const float* host_whh_bwd, int hidden, int planes, uint16_t* host_out) {
    AMTX_REQUIRE(host_whh_fwd && host_whh_bwd && host_out && (planes == 1 || planes == 2), "amtx_bilstm_h_pack: bad argument");
    AMTX_REQUIRE(hidden == 128 || hidden == 256 || hidden == 384 || hidden == 512, "amtx_bilstm_h_pack: hidden size %d is not built (128, 256, 384, 512)", hidden);
    amtx_bilstm_pack_host_h(host_whh_fwd, host_whh_bwd, hidden, planes, host_out);
    return AMTX_OK;
}

extern "C" int amtx_bilstm_h_fwd(const void* xproj, const uint16_t* whh_packed, int hidden, int planes, int elem_type, void* out, int batch,
                                 int num_frames, void* stream) {
    LstmArgs l;
    l.xproj = xproj; l.x_type = elem_type; l.whh = whh_packed; l.planes = planes; l.out = out; l.out_type = elem_type;
    l.B = batch; l.T = num_frames; l.groups = 1; l.x_gs = l.w_gs = l.out_gs = 0; l.hidden = hidden;
    return amtx_launch_bilstm(l, (hipStream_t)stream);
}

extern "C" int amtx_pianoroll_fwd(const float* logits, int64_t ld, int col0, int batch, int num_frames, int keys, float threshold,
                                  float* out, void* stream) {
    return amtx_launch_pianoroll(logits, ld, col0, batch, num_frames, keys, threshold, out, (hipStream_t)stream);
}

extern "C" size_t amtx_bce_logits_loss_workspace_bytes(int batch, int num_frames, int keys) {
    if (batch <= 0 || num_frames <= 0 || keys <= 0) return 0;
    return amtx_bce_loss_partials(batch, num_frames, keys) * sizeof(float);
}

extern "C" int amtx_bce_logits_loss(const float* logits, int64_t ld, const float* labels, const float* weight, int batch, int num_frames,
                                    int keys, float* loss, float* grad, void* workspace, size_t workspace_bytes, void* stream) {
    AMTX_REQUIRE(workspace && workspace_bytes >= amtx_bce_logits_loss_workspace_bytes(batch, num_frames, keys),
                 "amtx_bce_logits_loss: workspace too small");
    return amtx_launch_bce_loss(logits, ld, labels, weight, batch, num_frames, keys, loss, grad, (float*)workspace, (hipStream_t)stream);
}

// ---- training entry points of the BiLSTM (amt_tools_amd/autograd.py)
extern "C" int amtx_bilstm_pack_device(const float* whh_fwd, const float* whh_bwd, int planes, uint16_t* frag_fwd, uint16_t* frag_bwd, void* stream) {
    return amtx_launch_bilstm_pack_dev(whh_fwd, whh_bwd, planes, (bf16_t*)frag_fwd, (bf16_t*)frag_bwd, (hipStream_t)stream);
}

extern "C" int amtx_bilstm_train_fwd(const float* xproj, const uint16_t* whh_packed, int planes, float* out, float* save, int batch, int num_frames,
                                     void* stream) {
    AMTX_REQUIRE(xproj && whh_packed && out && save, "amtx_bilstm_train_fwd: null pointer");
    LstmArgs l;
    l.xproj = xproj; l.x_type = AMTX_T_F32; l.whh = (const bf16_t*)whh_packed; l.planes = planes; l.out = out; l.out_type = AMTX_T_F32;
    l.B = batch; l.T = num_frames; l.groups = 1; l.x_gs = l.w_gs = l.out_gs = 0; l.save = save;
    AMTX_REQUIRE(planes == 2, "amtx_bilstm_train_fwd: training runs in the fp32-class (two-plane) precision");
    return amtx_launch_bilstm(l, (hipStream_t)stream);
}

extern "C" int amtx_bilstm_train_bwd(const float* dout, const float* save, const uint16_t* whh_t_packed, int planes, float* dxproj, int batch,
                                     int num_frames, void* stream) {
    return amtx_launch_bilstm_bwd(dout, save, (const bf16_t*)whh_t_packed, planes, dxproj, batch, num_frames, (hipStream_t)stream);
}

// ---- training recurrences for any built hidden size (128: the register-stationary kernels, 256 / 384: the streaming ones)
extern "C" int amtx_bilstm_h_pack_device(const float* whh_fwd, const float* whh_bwd, int hidden, int planes, uint16_t* frag_fwd, uint16_t* frag_bwd,
                                         void* stream) {
    return amtx_launch_bilstm_pack_dev_h(whh_fwd, whh_bwd, hidden, planes, frag_fwd, frag_bwd, (hipStream_t)stream);
}

extern "C" int amtx_bilstm_h_train_fwd(const float* xproj, const uint16_t* whh_packed, int hidden, int planes, float* out, float* save, int batch,
                                       int num_frames, int groups, void* stream) {
    AMTX_REQUIRE(save && groups >= 1, "amtx_bilstm_h_train_fwd: null save buffer / bad group count");
    const int64_t bt = (int64_t)batch * num_frames;
    LstmArgs l;
    l.xproj = xproj; l.x_type = AMTX_T_F32; l.whh = whh_packed; l.planes = planes; l.out = out; l.out_type = AMTX_T_F32;
    l.B = batch; l.T = num_frames; l.hidden = hidden; l.save = save;
    l.groups = groups; l.x_gs = bt * 8 * hidden; l.w_gs = (int64_t)amtx_bilstm_wfrag_elems_h(hidden, planes); l.out_gs = bt * 2 * hidden;
    return amtx_launch_bilstm(l, (hipStream_t)stream);
}

extern "C" int amtx_bilstm_h_train_bwd(const float* dout, const float* save, const uint16_t* whh_t_packed, int hidden, int planes, float* dxproj,
                                       int batch, int num_frames, int groups, void* stream) {
    return amtx_launch_bilstm_bwd_h(dout, save, whh_t_packed, hidden, planes, dxproj, batch, num_frames, groups, (hipStream_t)stream);
}
