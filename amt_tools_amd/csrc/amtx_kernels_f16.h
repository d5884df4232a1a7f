// Prototypes of the OPTIONAL half-precision-operand builds of conv.hip / convf.hip / gemm.hip / lstm.hip (amtx_f16_names.h) that the engine calls
// for models created with AMTX_PREC_F16.  Same arguments and layouts as their bf16 namesakes in amtx_kernels.h; packed weights and
// 16-bit activations hold IEEE half instead of bf16.
#pragma once
#include "amtx_kernels.h"

#ifdef AMTX_WITH_F16
void amtx_conv3x3_pack_host_f16(const float* w, const float* scale, int c_out, int planes, bf16_t* out);
void amtx_conv1_pack_host_f16(const float* w, const float* scale, int c_in, int planes, bf16_t* out);
int amtx_launch_conv3x3_f16(const ConvArgs& c, hipStream_t stream);
int amtx_launch_conv1_f16(const Conv1Args& c, hipStream_t stream);
int amtx_launch_conv_stack_f16(const ConvArgs& c2, const bf16_t* w3frag, int64_t w3_gs, const float* shift3, void* out, int64_t out_gs,
                               int64_t out_plane, hipStream_t stream);
void amtx_conv3x3_gen_pack_host_f16(const float* w, const float* scale, int c_in, int c_out, int planes, bf16_t* out);
void amtx_conv1g_pack_host_f16(const float* w, const float* scale, int c_in, int c_mid, int planes, bf16_t* out);
int amtx_launch_conv3x3_gen_f16(const ConvArgs& c, int c_in, hipStream_t stream);
void amtx_gemm_pack_host_f16(const float* W, int64_t ldw, int N, int K, int planes, bf16_t* out);
int amtx_launch_gemm_f16(const GemmArgs& g, hipStream_t stream);
void amtx_bilstm_pack_host_h_f16(const float* whh_fwd, const float* whh_bwd, int hidden, int planes, bf16_t* out);
int amtx_launch_bilstm_f16(const LstmArgs& l, hipStream_t stream);
int amtx_pack_conv3x3_dev_f16(const float* w, const float* scale, int c_out, int planes, bf16_t* out, hipStream_t s);
int amtx_pack_conv1_dev_f16(const float* w, const float* scale, int planes, bf16_t* out, hipStream_t s);
int amtx_pack_conv_gen_dev_f16(const float* w, const float* scale, int c_in, int c_out, int ntc, int planes, bf16_t* out, hipStream_t s);
int amtx_pack_conv1g_dev_f16(const float* w, const float* scale, int c_in, int c_mid, int planes, bf16_t* out, hipStream_t s);
int amtx_pack_linear_dev_f16(const float* W, int64_t ldw, int N, int K, int planes, int n_pad, int k_pad, int row0, int rows_owned, int perm_c, int perm_f,
                             bf16_t* out, hipStream_t s);
int amtx_launch_bilstm_pack_dev_f16(const float* whh_fwd, const float* whh_bwd, int planes, bf16_t* frag_fwd, bf16_t* frag_bwd, hipStream_t stream);
int amtx_launch_bilstm_pack_dev_h_f16(const float* whh_fwd, const float* whh_bwd, int hidden, int planes, bf16_t* frag_fwd, bf16_t* frag_bwd, hipStream_t stream);
#else
// Library built without the half-operand twin objects (the default: amt_tools_amd/build.py, AMTX_BUILD_F16=1 adds them): amtx_of_model_create
// refuses AMTX_PREC_F16, so every `f16 ? x_f16 : x` in the engine is dead code -- the names alias their bf16 namesakes to keep it compiling.
#define amtx_conv3x3_pack_host_f16 amtx_conv3x3_pack_host
#define amtx_conv1_pack_host_f16 amtx_conv1_pack_host
#define amtx_launch_conv3x3_f16 amtx_launch_conv3x3
#define amtx_launch_conv1_f16 amtx_launch_conv1
#define amtx_launch_conv_stack_f16 amtx_launch_conv_stack
#define amtx_conv3x3_gen_pack_host_f16 amtx_conv3x3_gen_pack_host
#define amtx_conv1g_pack_host_f16 amtx_conv1g_pack_host
#define amtx_launch_conv3x3_gen_f16 amtx_launch_conv3x3_gen
#define amtx_gemm_pack_host_f16 amtx_gemm_pack_host
#define amtx_launch_gemm_f16 amtx_launch_gemm
#define amtx_bilstm_pack_host_h_f16 amtx_bilstm_pack_host_h
#define amtx_launch_bilstm_f16 amtx_launch_bilstm
#define amtx_pack_conv3x3_dev_f16 amtx_pack_conv3x3_dev
#define amtx_pack_conv1_dev_f16 amtx_pack_conv1_dev
#define amtx_pack_conv_gen_dev_f16 amtx_pack_conv_gen_dev
#define amtx_pack_conv1g_dev_f16 amtx_pack_conv1g_dev
#define amtx_pack_linear_dev_f16 amtx_pack_linear_dev
#define amtx_launch_bilstm_pack_dev_f16 amtx_launch_bilstm_pack_dev
#define amtx_launch_bilstm_pack_dev_h_f16 amtx_launch_bilstm_pack_dev_h
#endif
