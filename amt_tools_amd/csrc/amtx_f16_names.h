// Included FIRST by the files that amt_tools_amd/build.py compiles twice (conv.hip, convf.hip, convg.hip, gemm.hip, lstm.hip, pack.hip).  In the second
// build (-DAMTX_F16: IEEE half operands instead of bf16, see amtx_common.h) every public function of those files gets the suffix _f16,
// declarations in amtx_kernels.h included, so that both objects link into one libamtx.so.  The engine (ofmodel.hip) picks the variant by
// the model's precision; the prototypes it needs are in amtx_kernels_f16.h.
#pragma once
#ifdef AMTX_F16
#define amtx_conv3x3_wfrag_elems amtx_conv3x3_wfrag_elems_f16
#define amtx_conv3x3_pack_host amtx_conv3x3_pack_host_f16
#define amtx_conv1_wfrag_elems amtx_conv1_wfrag_elems_f16
#define amtx_conv1_pack_host amtx_conv1_pack_host_f16
#define amtx_launch_conv3x3 amtx_launch_conv3x3_f16
#define amtx_launch_conv1 amtx_launch_conv1_f16
#define amtxdbg_conv_prof amtxdbg_conv_prof_f16
#define amtx_conv_stack_fused_ok amtx_conv_stack_fused_ok_f16
#define amtx_launch_conv_stack amtx_launch_conv_stack_f16
#define amtxdbg_convf_prof amtxdbg_convf_prof_f16
#define amtx_conv3x3_gen_ntc amtx_conv3x3_gen_ntc_f16
#define amtx_conv3x3_gen_wfrag_elems amtx_conv3x3_gen_wfrag_elems_f16
#define amtx_conv3x3_gen_pack_host amtx_conv3x3_gen_pack_host_f16
#define amtx_conv1g_wfrag_elems amtx_conv1g_wfrag_elems_f16
#define amtx_conv1g_pack_host amtx_conv1g_pack_host_f16
#define amtx_conv3x3_gen_can_fuse1 amtx_conv3x3_gen_can_fuse1_f16
#define amtx_launch_conv3x3_gen amtx_launch_conv3x3_gen_f16
#define amtxdbg_convg_prof amtxdbg_convg_prof_f16
#define amtx_pack_bn_fold_dev amtx_pack_bn_fold_dev_f16
#define amtx_pack_conv3x3_dev amtx_pack_conv3x3_dev_f16
#define amtx_pack_conv1_dev amtx_pack_conv1_dev_f16
#define amtx_pack_linear_dev amtx_pack_linear_dev_f16
#define amtx_pack_head_fold_dev amtx_pack_head_fold_dev_f16
#define amtx_pack_vec_add_dev amtx_pack_vec_add_dev_f16
#define amtx_gemm_pack_dims amtx_gemm_pack_dims_f16
#define amtx_gemm_pack_host amtx_gemm_pack_host_f16
#define amtx_launch_gemm_multi amtx_launch_gemm_multi_f16
#define amtx_gemm_has_roll_epilogue amtx_gemm_has_roll_epilogue_f16
#define amtx_launch_gemm amtx_launch_gemm_f16
#define amtx_bilstm_wfrag_elems amtx_bilstm_wfrag_elems_f16
#define amtx_bilstm_wfrag_elems_h amtx_bilstm_wfrag_elems_h_f16
#define amtx_bilstm_pack_host_h amtx_bilstm_pack_host_h_f16
#define amtx_bilstm_pack_host amtx_bilstm_pack_host_f16
#define amtx_launch_bilstm amtx_launch_bilstm_f16
#define amtx_launch_bilstm_pack_dev amtx_launch_bilstm_pack_dev_f16
#define amtx_launch_bilstm_bwd amtx_launch_bilstm_bwd_f16
#define amtx_launch_bilstm_pack_dev_h amtx_launch_bilstm_pack_dev_h_f16
#define amtx_launch_bilstm_bwd_h amtx_launch_bilstm_bwd_h_f16
#endif
