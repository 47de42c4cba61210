/*
 * snvc_hip.h -- C ABI of libsnvc_hip.so, the MI355X (gfx950) implementation of SNVC's
 * cost-volume / voxel-resampling / 3D-CNN hot path.
 *
 * The reference has no C FFI: its native boundary is three pybind11 modules taking
 * at::Tensor.  The entry points below are what those modules'
 * bodies would call once tensors have been checked and unwrapped, in the reference's own
 * raw-pointer launcher style (snvc/extension/roiaware_pool3d/src/roiaware_pool3d.cpp:19-27):
 * plain device pointers + sizes + a stream, no allocation (except where stated), no
 * synchronisation, int status return (0 = ok).  Each declaration cites the reference
 * interface it replaces.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - tensors are contiguous in the layout given (NCDHW family, W fastest), exactly the
 *     layout the reference's .contiguous().data<T>() hands its kernels;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is
 *     asynchronous on that stream;
 *   - sizes are int64_t, indices inside the kernels are 64-bit (the reference overflows
 *     32-bit indices at N*2C*D*H*W >= 2^31, BuildCostVolume_cuda.cu:70-82);
 *   - status codes: see snvc_status; snvc_last_error_string() gives the text of the last
 *     failing call on the calling thread.
 */
#ifndef SNVC_HIP_H
#define SNVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNVC_API __attribute__((visibility("default")))

typedef enum {
    SNVC_OK = 0,
    SNVC_ERR_INVALID_ARGUMENT = 1, /* shapes / flags the kernels cannot honour */
    SNVC_ERR_UNSUPPORTED = 2,      /* valid request outside the instantiated kernel set */
    SNVC_ERR_HIP = 3               /* a HIP runtime call or launch failed */
} snvc_status;

typedef enum { SNVC_F32 = 0, SNVC_F64 = 1 } snvc_dtype;

SNVC_API const char *snvc_last_error_string(void);
/* Version of this ABI; bumped on any signature change. */
SNVC_API int snvc_abi_version(void);

/* ------------------------------------------------------------------------------------
 * a1 / a2  build_cost_volume
 * replaces: build_cost_volume_cuda.build_cost_volume_forward / _backward
 *           (snvc/extension/build_cost_volume/src/BuildCostVolume.cpp:13-48,
 *            launchers BuildCostVolume_cuda.cu:208-303)
 *   left,right [N,C,Hi,Wi]   shift [N,D] (>= 0, same dtype)   out [N,2C,D,Hi/ds,Wi/ds]
 *   Hi and Wi must be multiples of ds (the reference silently mis-strides otherwise).
 * backward: grad [N,2C,D,H,W] -> grad_left, grad_right [N,C,H*ds,W*ds]; both outputs are
 *   fully written (no pre-zeroing needed); deterministic gather, no atomics.
 * ---------------------------------------------------------------------------------- */
SNVC_API int snvc_cost_volume_forward(const void *left, const void *right, const void *shift,
                                      void *out, int64_t N, int64_t C, int64_t Hi, int64_t Wi,
                                      int64_t D, int64_t downsample, int dtype, void *stream);
/* Right (warped) half only: out [N,C,D,Hi/ds,Wi/ds] = out_full[:, C:].  fp32. */
SNVC_API int snvc_cost_volume_forward_right(const float *right, const float *shift, float *out,
                                            int64_t N, int64_t C, int64_t Hi, int64_t Wi, int64_t D,
                                            int64_t downsample, void *stream);
SNVC_API int snvc_cost_volume_backward(const void *grad, const void *shift, void *grad_left,
                                       void *grad_right, int64_t N, int64_t C, int64_t H,
                                       int64_t W, int64_t D, int64_t downsample, int dtype,
                                       void *stream);

/* Training through the FACTORED first convolution (see snvc_conv3d_forward_ex): the two adjoints that have no
 * counterpart in the materialised path.
 *   snvc_cost_volume_backward_right : the warped half of snvc_cost_volume_backward alone -- grad_right_half [N,C,D,H,W]
 *       (gradient of snvc_cost_volume_forward_right's output) -> grad_right [N,C,H,W]; fp32, downsample 1; same sum in the
 *       same order as the full backward's right half (bit-identical).
 *   snvc_depth_class_sums : adjoint of the depth-class planes: g [NC,D,HW] -> out [NC,3,HW] = (g[0], sum_{d=1..D-2} g[d],
 *       g[D-1]), d ascending. */
SNVC_API int snvc_cost_volume_backward_right(const float *grad_right_half, const float *shift, float *grad_right, int64_t N,
                                             int64_t C, int64_t H, int64_t W, int64_t D, void *stream);
SNVC_API int snvc_depth_class_sums(const float *g, float *out, int64_t NC, int64_t D, int64_t HW, void *stream);

/* ------------------------------------------------------------------------------------
 * a3  feature -> voxel resampling
 * replaces: VernierScale._sample_2d_feat(aggregate="concat")  (snvc/models/vernier.py:323-349):
 *   coordinate normalisation (:335-338) + 2x F.grid_sample(bilinear, zeros, align_corners=False)
 *   (:339-340) + torch.cat (:346), in one pass.
 *   left,right [N,F,Hf,Wf]   l_pts,r_pts [N,2,V] RoI-pixel (x row, y row)
 *   res_x = cfg.resolution[1], res_y = cfg.resolution[0]   out [N,2F,V] (V = nh*nw*nl)
 *   The coordinate tensors are NOT modified (the reference normalises them in place).
 * ---------------------------------------------------------------------------------- */
SNVC_API int snvc_voxel_gather_forward(const float *left, const float *right, const float *l_pts,
                                       const float *r_pts, float *out, int64_t N, int64_t F,
                                       int64_t Hf, int64_t Wf, int64_t V, float res_x,
                                       float res_y, void *stream);
/* adjoint of the above w.r.t. the feature maps (bilinear scatter); grad_left/right [N,F,Hf,Wf]
 * are fully written; deterministic order is NOT guaranteed (float atomics), like
 * grid_sampler_2d_backward on GPUs. */
/* Same result (bit for bit), faster: `workspace` holds snvc_voxel_gather_workspace_floats(N,F,Hf,Wf) floats
 * (16-byte aligned) into which the two feature maps are first re-laid channels-last, so that one 16-byte load
 * fetches 4 channels of a bilinear tap.  Needs F % 4 == 0; otherwise (or with workspace == NULL) it is
 * snvc_voxel_gather_forward. */
SNVC_API int64_t snvc_voxel_gather_workspace_floats(int64_t N, int64_t F, int64_t Hf, int64_t Wf);
SNVC_API int snvc_voxel_gather_forward_ws(const float *left, const float *right, const float *l_pts,
                                          const float *r_pts, float *out, float *workspace, int64_t N,
                                          int64_t F, int64_t Hf, int64_t Wf, int64_t V, float res_x,
                                          float res_y, void *stream);
/* Deterministic form of the adjoint (training; VERDICT r1: the atomics above are order-dependent and collapse when many
 * voxels project onto one pixel): voxels are grouped by their base pixel with a stable radix sort, every sum has one
 * fixed order, no atomics -> bit-reproducible run to run.  workspace: snvc_voxel_gather_backward_workspace_bytes()
 * bytes, 256-byte aligned.  Same result as snvc_voxel_gather_backward up to summation order. */
SNVC_API int64_t snvc_voxel_gather_backward_workspace_bytes(int64_t N, int64_t F, int64_t Hf, int64_t Wf, int64_t V);
SNVC_API int snvc_voxel_gather_backward_det(const float *grad_out, const float *l_pts, const float *r_pts,
                                            float *grad_left, float *grad_right, void *workspace, int64_t N, int64_t F,
                                            int64_t Hf, int64_t Wf, int64_t V, float res_x, float res_y, void *stream);
/* replaces: the aggregate="concat-atten" branch of _sample_2d_feat (vernier.py:341-344), applied in place to the gather's
 * result vox [N,2F,V]: every channel of voxel v is multiplied by max(cosine_similarity(vox[:, :F, v], vox[:, F:, v]), 0). */
SNVC_API int snvc_voxel_atten_scale(float *vox, int64_t N, int64_t F, int64_t V, void *stream);
SNVC_API int snvc_voxel_gather_backward(const float *grad_out, const float *l_pts,
                                        const float *r_pts, float *grad_left, float *grad_right,
                                        int64_t N, int64_t F, int64_t Hf, int64_t Wf, int64_t V,
                                        float res_x, float res_y, void *stream);

/* ------------------------------------------------------------------------------------
 * a11  projection of the 3D sampling grid into the RoI crops (producer of a3's coordinates)
 * replaces: refinementDataset._init_3d_grid / _to_cam / _generate_grid_proj
 *   (snvc/dataset/KITTIRefinement_dataset.py:267-282,828-868) + Calibration.project_rect_to_image
 *   (snvc/dataset/kitti_util.py:282-293) + affine_transform (snvc/utils/img_proc.py:71-74),
 *   which run in numpy float64 on the host and are then copied to the device.
 *   samples [N,7] (h,w,l,x,y,z,ry) f64, P_left / P_right [3,4] f64, trans_l / trans_r [N,2,3] f64
 *   (all DEVICE pointers), ranges_host = {x_min,x_max,y_min,y_max,z_min,z_max} (HOST pointer),
 *   grid (nh,nw,nl) -> out_l, out_r [N,2,V] f32 (x row, y row; V = nh*nw*nl, index (ih*nw+iw)*nl+il),
 *   grid_cam [N,V,3] f64 camera-frame points (may be NULL).
 * ---------------------------------------------------------------------------------- */
SNVC_API int snvc_grid_projection(const double *samples, const double *P_left, const double *P_right,
                                  const double *trans_l, const double *trans_r,
                                  const double *ranges_host, int nh, int nw, int nl, float *out_l,
                                  float *out_r, double *grid_cam, int N, void *stream);

/* ------------------------------------------------------------------------------------
 * a4-a7  3D convolution / transposed convolution with fused epilogue
 * replaces: nn.Conv3d / nn.ConvTranspose3d (+ eval-mode BatchNorm3d or a per-channel affine)
 *   (+ residual add) (+ ReLU / Sigmoid) as composed by convbn_3d, hourglass,
 *   get_hg_down_sample / get_hg_up_sample, hourglass_downsample_16
 *   (snvc/models/submodule.py:32-50,85-268) and VernierScale._init_3d_net /
 *   predict_3d_heatmaps (snvc/models/vernier.py:249-295,414-438).
 *
 *   y = epilogue( conv(x, w) ),   epilogue(v) for output channel c:
 *       v = v * scale[c] + bias[c]           (scale/bias may be NULL: identity)
 *       if (flags & SNVC_EPI_ADD_PRE)  v += residual
 *       if (flags & SNVC_EPI_RELU)     v = max(v, 0)
 *       if (flags & SNVC_EPI_SIGMOID)  v = 1 / (1 + exp(-v))
 *       if (flags & SNVC_EPI_ADD_POST) v += residual
 *
 *   x [N,Cin,Din,Hin,Win]  y [N,Cout,Dout,Hout,Wout]  residual like y.
 *   Batch strides (in elements) let x / y / residual be channel slices of larger buffers
 *   (the torch.cat of vernier.py:433 is built in place that way); 0 = dense.
 *   Cubic kernels only, as on the path: ksize in {1,3,5,7}, stride in {1,2}, dilation in {1,2}
 *   with pad = dilation*(ksize-1)/2 ("same" for stride 1); transposed: ksize 3, stride 2,
 *   pad 1, output_padding 1 (Dout = 2*Din).
 *   Arithmetic: fp32 in, fp32 products and accumulation on v_mfma_f32_32x32x2_f32; see desc.algo.
 * ---------------------------------------------------------------------------------- */
enum {
    SNVC_EPI_RELU = 1,
    SNVC_EPI_ADD_PRE = 2,
    SNVC_EPI_ADD_POST = 4,
    SNVC_EPI_SIGMOID = 8,
    /* y = AvgPool3d((4,1,1),(4,1,1))(epilogue(conv(x))) as a [N,Cout,Dout/4,Hout,Wout] tensor, the full-resolution
     * result never written (snvc/models/vernier.py:289,436: the pool between the trunk's conv4 and the BEV reshape).
     * Built for 3x3x3 / stride-1 Conv3d layers on the default Winograd form with Dout % 4 == 0 and no residual;
     * SNVC_ERR_UNSUPPORTED otherwise (the caller then pools in a launch of its own: snvc_avgpool_depth4). */
    SNVC_EPI_AVGPOOL_D4 = 16,
    /* snvc_sheared_expand_split / snvc_warped_expand_split: write the result with non-temporal stores (a 0.74 GB tensor that is read
     * once, by the next layer, long after the L2 has turned over: 4.0 -> 4.5 TB/s at cfg2, r5) */
    SNVC_EPI_STREAM_OUT = 32,
    /* snvc_warped_expand only: the r3 kernel form (four 16-byte LDS reads per kd and plane) instead of the register-window
     * form; same values up to fp32 rounding, kept selectable for the parity tests and A/B timing. */
    SNVC_WARPED_EXPAND_R3 = 256
};

/* desc.algo: which arithmetic / kernel form a layer may use.  Low byte = the arithmetic:
 *   SNVC_ALGO_AUTO   : fastest form inside the 1e-3 contract -- Winograd F(4,k) along W where the layer
 *                      qualifies (fp32 products and sums; measured 2e-6 (k3) .. 1e-4 (k7) of the output range)
 *   SNVC_ALGO_DIRECT : the direct kernels only (an exact fp32 FMA chain per output)
 * ORed on top, kernel-form selectors.  The dispatcher's own choice (none of them set) is the measured-fastest
 * form; the selectors exist so that every instantiated form can be reached -- and parity-tested -- through the
 * ABI rather than through process environment:
 *   SNVC_ALGO_WINO_TILE_*      : k3 / stride-1 Winograd tile: 4x4x64 LDS-DMA staged (BIG), 2x4x64
 *                                register staged (STD), 4x4x32 register staged (NARROW_REG); default is
 *                                the 4x4x32 LDS-DMA staged tile at three workgroups per CU.  On a k3 /
 *                                stride-2 layer STD selects the per-chunk image refill form; the default there
 *                                is the slice-pipelined refill (counted vmcnt waits, conv3d.hip).  On a depth-1 k3 /
 *                                stride-1 layer BIG forces the 1x16x32 Winograd form that large calls take by themselves
 *   SNVC_ALGO_GENERIC_EPILOGUE : scalar predicated epilogue instead of the 16-byte-store ones
 *   SNVC_ALGO_SCALAR_STAGING   : snvc_conv3d_wgrad only: element-wise staging instead of the float4 form */
enum {
    SNVC_ALGO_AUTO = 0,
    SNVC_ALGO_DIRECT = 1,
    SNVC_ALGO_ARITH_MASK = 0xff,
    SNVC_ALGO_WINO_TILE_BIG = 0x100,
    SNVC_ALGO_WINO_TILE_STD = 0x200,
    SNVC_ALGO_WINO_TILE_NARROW_REG = 0x300,
    SNVC_ALGO_WINO_TILE_MASK = 0x300,
    SNVC_ALGO_GENERIC_EPILOGUE = 0x400,
    SNVC_ALGO_SCALAR_STAGING = 0x800,
    /* snvc_f16x3_* stride-1 layers only: the image's hi / lo planes taken one after the other (double-buffered 19.6 KB image)
     * instead of side by side (single-buffered 39 KB image); same values, kept selectable for A/B timing.  Pack and forward
     * must agree on it. */
    SNVC_ALGO_X3_SERIAL = 0x1000,
    /* snvc_f16x3_* stride-1 layers: one 32-channel block per workgroup even for Cout = 64 (NARROW), and additionally 2x4x32
     * tiles (SMALL) -- more, smaller workgroups for layers that would not fill the chip otherwise.  Pack and forward must agree. */
    SNVC_ALGO_X3_NARROW = 0x2000,
    SNVC_ALGO_X3_SMALL = 0x4000,
    /* snvc_f16x3_* / snvc_f16_* stride-1 layers with Cin % 8 == 0 (one channel group per k-block), Cout % 32 == 0 and a C8 output:
     * the kernel forms on v_mfma_f32_16x16x32_f16 (that instruction shape sustains ~20 % more under the chip's power limit).
     * 3x3x3: split mode only, no residual (an optional side head); 5^3, dilated 5^3, 7^3: split and fp16-storage mode, residual
     * allowed.  Same values as the 32x32x16 forms up to fp32 summation order.  Pack and forward must agree;
     * SNVC_ERR_UNSUPPORTED otherwise. */
    SNVC_ALGO_X3_Q16 = 0x8000,
    /* snvc_conv3d_wgrad only (r6): keep the fp32-MFMA forms.  By default a 3x3x3 / stride-1 layer with 16-byte aligned rows takes the
     * split-operand form: x and g are scaled by a power of two from their own maxima and split into (hi, lo) pairs of halves on the
     * way into LDS, every fp32 product is three v_mfma_f32_16x16x32_f16 with fp32 accumulation (22 significant bits per operand;
     * deterministic: fixed-order partial sums) -- conv3d_wgrad_x3_kernel, csrc/conv3d_bwd.hip. */
    SNVC_ALGO_WGRAD_FP32 = 0x10000
};

typedef struct {
    int32_t N, Cin, Din, Hin, Win;
    int32_t Cout, Dout, Hout, Wout;
    int32_t ksize, stride, dilation, pad;
    int32_t transposed; /* 0: Conv3d, 1: ConvTranspose3d(k3,s2,p1,op1) */
    int32_t flags;      /* SNVC_EPI_* */
    int32_t algo;       /* SNVC_ALGO_* */
    int32_t ksize_d;    /* kernel extent along D: 0 = cubic (= ksize); 1 = depth-1 layer, i.e. an nn.Conv2d(k, stride,
                           padding=(k-1)/2) of the 2D BEV neck (snvc/models/vernier.py:296-313, submodule.py:11-29,
                           270-361) run on the [N,C,1,H,W] view of its NCHW tensors: Din = Dout = 1, k in {1,3}, stride
                           in {1,2} applied to H and W only, weight [Cout,Cin,k,k] */
    int32_t ksize_h;    /* kernel extent along H of a depth-1 layer: 0 = ksize; 3 with ksize = 7 is the 3 x 7 layer of the
                           sheared first convolution (weight [Cout,Cin,3,7], padding (1,3)); anything else unsupported */
    int64_t x_batch_stride, y_batch_stride, res_batch_stride; /* elements; 0 = dense */
} snvc_conv3d_desc;

/* Number of floats of the packed-weight buffer for this layer. */
SNVC_API int64_t snvc_conv3d_packed_weight_count(const snvc_conv3d_desc *desc_host);
/* Re-lays nn.Conv3d weights [Cout,Cin,k,k,k] (or nn.ConvTranspose3d weights [Cin,Cout,3,3,3]
 * when desc.transposed) into the kernel's streaming order.  Run once per layer. */
SNVC_API int snvc_conv3d_pack_weights(const snvc_conv3d_desc *desc_host, const float *weight,
                                      float *packed, void *stream);
SNVC_API int snvc_conv3d_forward(const snvc_conv3d_desc *desc_host, const float *x,
                                 const float *packed_weight, const float *scale,
                                 const float *bias, const float *residual, float *y,
                                 void *stream);
/* Same, plus `depth_planes` [N,Cout,3,Hout,Wout] (may be NULL) added to the convolution result BEFORE
 * the affine: plane 0 for output depth 0, plane 2 for the last output depth, plane 1 in between.
 * This is the factored first convolution over a concat cost volume: its left half repeats the left
 * feature for every disparity plane (BuildCostVolume_cuda.cu:86), so that half of
 * conv(volume) is independent of d except at the two zero-padded ends -- three 2D planes, computed
 * by this same kernel on the left feature stacked 3 deep -- and only the right half of the volume
 * needs the full 3D convolution (and needs to exist at all). */
SNVC_API int snvc_conv3d_forward_ex(const snvc_conv3d_desc *desc_host, const float *x,
                                    const float *packed_weight, const float *scale,
                                    const float *bias, const float *residual,
                                    const float *depth_planes, float *y, void *stream);

/* ------------------------------------------------------------------------------------
 * Normalisation statistics for GroupNorm / train-mode BatchNorm3d
 * replaces: nn.GroupNorm(32, C) / nn.BatchNorm3d(C).train() inside convbn_3d
 *           (snvc/models/submodule.py:49,135,146,206)
 *   x [N,C,S] (S = D*H*W).  Statistics are taken over `groups` contiguous channel groups per
 *   sample (GroupNorm: per_sample = 1) or over the whole batch per channel (BatchNorm:
 *   per_sample = 0, groups = C).  Outputs: scale[N or 1][C], shift[N or 1][C] such that
 *   y = x * scale + shift reproduces (x - mean) * rsqrt(var + eps) * gamma + beta (biased var);
 *   mean_out / var_out (may be NULL) receive the statistics [N or 1][groups].
 *   workspace: at least snvc_norm_workspace_bytes() bytes.
 * ---------------------------------------------------------------------------------- */
SNVC_API int64_t snvc_norm_workspace_bytes(int64_t N, int64_t C, int64_t groups);
SNVC_API int snvc_norm_stats(const float *x, const float *gamma, const float *beta,
                             float *scale, float *shift, float *mean_out, float *var_out,
                             void *workspace, int64_t N, int64_t C, int64_t S,
                             int64_t x_batch_stride, int64_t groups, int per_sample, float eps,
                             void *stream);
/* y = epilogue(x * scale[n?][c] + shift[n?][c]) with the SNVC_EPI_* flags; in-place allowed. */
SNVC_API int snvc_affine_act(const float *x, const float *scale, const float *shift,
                             const float *residual, float *y, int64_t N, int64_t C, int64_t S,
                             int64_t x_batch_stride, int64_t y_batch_stride,
                             int64_t res_batch_stride, int per_sample, int flags, void *stream);
#define SNVC_AMAX_SLOTS 64
/* r6: the same pass; additionally atomicMax-es the bit pattern of max|y| into amax[0 .. SNVC_AMAX_SLOTS) (zeroed device words, a workgroup
 * picks one slot -- same-address atomics serialise; the maximum over the slots is the tensor's; may be NULL): the
 * split-operand weight gradient (snvc_conv3d_wgrad_amax) scales its operands by their maxima, and the pass that writes a tensor gets
 * its maximum for free where a separate reduction costs a read of the whole tensor. */
SNVC_API int snvc_affine_act_amax(const float *x, const float *scale, const float *shift,
                                  const float *residual, float *y, int64_t N, int64_t C, int64_t S,
                                  int64_t x_batch_stride, int64_t y_batch_stride,
                                  int64_t res_batch_stride, int per_sample, int flags, uint32_t *amax, void *stream);

/* ------------------------------------------------------------------------------------
 * Training backward of the 3D stack (BASELINE.json configs[3]; reference: torch autograd through
 * nn.Conv3d / nn.ConvTranspose3d / nn.BatchNorm3d / ReLU as composed in snvc/models/submodule.py).
 *   data gradient  : the forward kernels themselves -- a stride-1 Conv3d's dgrad is a Conv3d with the
 *                    taps flipped and channels transposed, a stride-2 Conv3d's dgrad is a
 *                    ConvTranspose3d(k3,s2,p1,op1) with the SAME weight memory, and vice versa.
 *   weight gradient: snvc_conv3d_wgrad (deterministic: per-partition slabs + fixed-order sum).
 *   epilogue       : forward was v = raw*scale + shift; [v += res]; y = act(v); [y += res].
 *                    snvc_act_backward_reduce gives, per (n, c), sum(g) and sum(g*raw) in fp64 with
 *                    g = gy*act'(v) (what BatchNorm / GroupNorm backward need);
 *                    snvc_act_backward_apply writes draw = coef_g*g + coef_raw*raw + coef_const
 *                    (per channel, or per (n, c) when per_sample) and optionally g itself.
 * ---------------------------------------------------------------------------------- */
/* Transposed layer + fused 1x1x1 head: y_head[n,0,:] = sum_c head_weight[c] * epilogue(conv(x, w))[n,c,:]; the
 * Cout-channel tensor itself is never written.  This is the tail of the global model
 * (stereo_volume.py: `v + hourglass(v)[0]` -> classifier Conv3d(C,1,1), the composition of vernier.py:366-371)
 * when the layer's output has no other consumer.  Needs desc.transposed, Cout == 32, an even input width, no
 * Sigmoid, a 16-byte aligned residual; returns SNVC_ERR_UNSUPPORTED otherwise (the caller then runs the two
 * layers separately). */
SNVC_API int snvc_conv3d_forward_head(const snvc_conv3d_desc *desc_host, const float *x,
                                      const float *packed_weight, const float *scale, const float *bias,
                                      const float *residual, const float *head_weight, float *y_head,
                                      void *stream);
/* A layer AND the 1x1x1 projection of its own result in one launch: y as snvc_conv3d_forward writes it, and
 * y_head[n,0,:] = sum_c head_weight[c] * y[n,c,:] ([N,1,D,H,W], 16-byte aligned).  Used for the `classifier(v)` term of
 * the global model's folded tail (stereo_volume.py): with no activation behind the hourglass's last transposed layer
 * (snvc/models/submodule.py:166), classifier(bn(deconv(post)) + v) = deconv'(post) + b' + classifier(v), where deconv' is
 * a ConvTranspose3d to ONE channel (run by snvc_conv3d_forward: desc.transposed, Cout = 1) and classifier(v) comes out
 * of the launch that produces v.  Built for 3x3x3 / stride-1 layers with 32 output channels on the default Winograd
 * form, no residual; returns SNVC_ERR_UNSUPPORTED otherwise (the caller then projects y in a launch of its own). */
SNVC_API int snvc_conv3d_forward_side_head(const snvc_conv3d_desc *desc_host, const float *x,
                                           const float *packed_weight, const float *scale, const float *bias,
                                           const float *residual, float *y, const float *head_weight,
                                           float *y_head, void *stream);
/* Convolution + the batch statistics of its own result in one launch (train-mode BatchNorm, cfg4): the 3x3x3 Winograd
 * kernels (stride 1 and 2, their default forms) leave per-workgroup fp64 (sum, sum of squares) of the raw result beside y, a
 * fold adds them in a fixed order and scale / shift [Cout] (for y' = scale*y + shift; gamma / beta may be NULL) and mean / var
 * [Cout] (may be NULL) come back as from snvc_norm_stats -- which would read the tensor once more.  desc.flags must be 0.
 * SNVC_ERR_UNSUPPORTED (nothing launched) when the layer does not take one of those kernel forms. */
SNVC_API int64_t snvc_conv3d_stats_workspace_bytes(const snvc_conv3d_desc *desc_host);
SNVC_API int snvc_conv3d_forward_stats(const snvc_conv3d_desc *desc_host, const float *x, const float *packed_weight, float *y,
                                       const float *gamma, const float *beta, float *scale, float *shift, float *mean,
                                       float *var, void *workspace, float eps, void *stream);
/* ------------------------------------------------------------------------------------
 * Sheared first convolution: the 3x3x3 convolution over the WARPED half of build_cost_volume's result when the
 * disparity planes are uniformly spaced, shift[n][d] = (m0 + d) / q with q in {1, 2} (whole- / half-pixel steps; the
 * plane-sweep volume of BASELINE.json configs[1] is q = 2, m0 = 0).  Then V[c][d][h][w] = Rq[c][h][q*w - d - m0] with
 * Rq the reference's own interpolation of the right feature on the 1/q-pixel grid (BuildCostVolume_cuda.cu:15-98), and
 * conv(V)[co][d][h][w] = G[co][h][q*w - d - m0] for a 2D convolution G of Rq (csrc/sheared_conv.hip has the algebra and
 * the three borders that are not sheared).  The warped volume is never built and the layer's 318 GFLOP become 3.4.
 *   snvc_sheared_upsample : out[n][c][h][i] = Rq[i - off] on a row of WU floats (zero outside 0 <= u <= q*(W-1); off may
 *                           be negative: a window of Rq)
 *   snvc_conv3d_forward   : G = the depth-1 3 x 7 convolution of that image (desc.ksize_d = 1, ksize_h = 3, ksize = 7); G' =
 *                           the same with the kernel stripped of its kw = +1 taps, over the window the last column reads;
 *                           both for the three depth classes (first plane d = 0: kernel without its kd = -1 taps; interior;
 *                           last plane d = D-1: without kd = +1), stacked class-major as 3*C output channels
 *   snvc_sheared_expand   : y[n][co][d][h][w] = epilogue(scale*G[n][cls(d)][co][h][q*w - d - m0 + off] + planes[n][co][cls(d)][h][w]),
 *                           G'[..][q*(W-1) - d - m0 + off2] in place of G at w = W-1; g [N][3][C][H][WG], gcol [N][3][C][H][WG2],
 *                           planes = the depth-class planes of snvc_conv3d_forward_ex (or NULL).
 * r6: snvc_sheared_expand / snvc_sheared_expand_split also take q = 4 (index 4*w - d - m0: four phases): the layer at
 * `downsample` ds > 1 (BuildCostVolume_cuda.cu:224-225: x = ds*w - shift on row ds*h) is the same shear with q*ds phases of the
 * row-subsampled right feature, upsampled by q alone -- ds = 2 with whole-pixel planes = two phases (the q = 2 kernels, Rq = the
 * feature itself), ds = 2 with half-pixel planes and ds = 4 with whole-pixel planes = four.  With four phases the 3 x 11 kernel of G
 * splits into three 3 x 3 layers whose results are added 4 elements apart (host side: models/stereo_volume.py). */
/* The same first convolution for an ARBITRARY shift array (shift >= 0): linear interpolation along w commutes with the
 * convolution, so conv3d(warped half)[co][d][h][w] = sum_kd lerp(P_kd[co][h][:], w - shift[d+kd-1]) with P_kd = the depth-1 3x3
 * convolution of the right feature with the layer's kd slice -- three 2D convolutions computed once and three
 * interpolations per output voxel; the two places where the reference is not "interpolate the zero-extended signal" are
 * corrected exactly (csrc/sheared_conv.hip: q = the kw = +1 taps alone, for the zero padding at w = W-1; e = the (c, kh)
 * contraction of the image's first column per (kd, kw), for the gate x >= 0).  The warped volume is never built.
 *   y[n][co][d][h][w] = epilogue(scale * (sum_kd ...) + planes[n][co][cls(d)][h][w] ...)
 *   p, q [N][3][C][H][W], e [N][3][3][C][H][4] (column 0 used), planes [N][C][3][H][W] or NULL, shift [N][D] float32. */
SNVC_API int snvc_warped_expand(const float *p, const float *q, const float *e, const float *planes, const float *shift,
                                const float *scale, const float *bias, float *y, int64_t N, int64_t C, int64_t D, int64_t H,
                                int64_t W, int flags, void *stream);
/* Adjoint of snvc_warped_expand w.r.t. the right feature and the weights, in one pass over the output gradient dy [N][C][D][H][W]
 * (scale = 1, no activation: the caller applies the epilogue's backward first), replacing the reference's Conv3d backward over
 * the built volume + BuildCostVolume_cuda.cu:152-205 for the right half:
 *   a[n][kd][kw][co][h][j] = the gradient of plane e = d+kd-1's warp, taken back through tap (kd, kw):
 *       sum_e f_e [j+m_e+1 <= W-1] dy[co][e-kd+1][h][j+m_e+2-kw] + g_e [j >= 1, j+m_e <= W-1] dy[co][e-kd+1][h][j+m_e+1-kw]
 *   (shift[e] = m_e + f_e, g_e = 1 - f_e; a whole-pixel shift counts as (m-1, 1, 0); planes with a negative or NaN shift add 0),
 *   so that dWt[co][c][kd][kh][kw] = sum_{n,h,j} a * right[n][c][h+kh-1][j] and dRight = the depth-1 transposed convolution of a.
 *   dplanes [N][C][3][H][W] or NULL: the depth-class sums of dy (snvc_depth_class_sums) from the same pass.
 * Deterministic.  W <= 1024. */
SNVC_API int snvc_warped_expand_backward(const float *dy, const float *shift, float *a, float *dplanes, int64_t N, int64_t C,
                                         int64_t D, int64_t H, int64_t W, void *stream);
/* snvc_warped_expand with the result written as a split C8 pair (y_hi, y_lo), like snvc_sheared_expand_split. */
SNVC_API int snvc_warped_expand_split(const float *p, const float *q, const float *e, const float *planes, const float *shift,
                                      const float *scale, const float *bias, void *y_hi, void *y_lo, int *overflow,
                                      int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int64_t y_batch_stride, int flags,
                                      void *stream);
/* Structure of a float32 shift array [N][D] in one launch: out4 = { all >= 0, every row == shift[0][0] + d (q = 1),
 * every row == shift[0][0] + d/2 (q = 2), shift[0][0] } (flags as 1.0 / 0.0, exact fp32 comparisons).  The caller reads the four
 * floats back: the same single device -> host sync as the reference wrapper's `assert torch.all(shift >= 0)`
 * (snvc/extension/build_cost_volume/__init__.py:12), which it replaces. */
SNVC_API int snvc_shift_structure(const float *shift, float *out4, int64_t N, int64_t D, void *stream);
SNVC_API int snvc_sheared_upsample(const float *right, float *out, int64_t N, int64_t C, int64_t H, int64_t W, int q,
                                   int64_t WU, int off, void *stream);
SNVC_API int snvc_sheared_expand(const float *g, const float *gcol, const float *planes, const float *scale,
                                 const float *bias, float *y, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int q,
                                 int m0, int64_t WG, int off, int64_t WG2, int off2, int flags, void *stream);
/* r6: + the bit pattern of max|y| into SNVC_AMAX_SLOTS zeroed words (see snvc_affine_act_amax): the training step's next layer scales its
 * split operands by it -- exact, and without the dozen small launches a bound formed from max|G|, max|G'|, max|planes| took. */
SNVC_API int snvc_sheared_expand_amax(const float *g, const float *gcol, const float *planes, const float *scale, const float *bias,
                                      float *y, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int q, int m0, int64_t WG,
                                      int off, int64_t WG2, int off2, int flags, uint32_t *amax, void *stream);
/* snvc_sheared_expand with the result written as a split C8 pair (y_hi, y_lo: half [N][2][C/8][D][H][W][8] planes, see the
 * split-mode section below) for a consumer on the snvc_f16x3_* kernels; scale / bias carry the tensor's exponent (the host
 * folds 2^e in: exact).  y_batch_stride in halves (0: dense pair).  overflow (device int, may be NULL): set to 1 if a value had
 * to be clamped to half's range. */
SNVC_API int snvc_sheared_expand_split(const float *g, const float *gcol, const float *planes, const float *scale,
                                       const float *bias, void *y_hi, void *y_lo, int *overflow, int64_t N, int64_t C,
                                       int64_t D, int64_t H, int64_t W, int q, int m0, int64_t WG, int off, int64_t WG2,
                                       int off2, int64_t y_batch_stride, int flags, void *stream);
/* Backward of the sheared first convolution (training, BASELINE.json configs[3]); dy = the gradient of the layer's RAW
 * result (the caller applies the norm / activation backward first, snvc_act_backward_*).
 *   snvc_sheared_reduce            : adjoint of snvc_sheared_expand: dg [N][3][C][H][WG] / dgcol [N][3][C][H][WG2] = the
 *                                    sums of dy along each shear line, per depth class; every element written; deterministic
 *   snvc_sheared_wgrad             : dK[co][c][kh][t] = sum_{n,h,i} dy[n][co][h][i] * x[n][c][h+kh-1][i+t-3] of the depth-1
 *                                    3 x 7 layer (x [N][C][H][WU], dy [N][CO][H][WU], C <= 32; dk [CO][C][3][7]) on the matrix
 *                                    pipe, partial sums in `workspace` (snvc_sheared_wgrad_workspace_bytes) reduced in a
 *                                    fixed order.  The 3D gradient is dW[kd][kh][kw] = sum over classes holding kd of
 *                                    dK_cls[kh][q*kw - kd + 3] (+ the last-column window's, kw <= 0)
 *   snvc_sheared_upsample_backward : adjoint of snvc_sheared_upsample: dright[j] = dRq[q*j] (+ (dRq[2j-1] + dRq[2j+1]) / 2 for
 *                                    q = 2), drq [N][C][H][WU] with element i = dRq[i - off]
 * The input gradient dRq itself is snvc_conv3d_forward of dg with the flipped / transposed 3 x 7 kernel. */
/* Train-mode BatchNorm around the sheared layer WITHOUT storing the layer's raw result (the 736 MB tensor of cfg4):
 *   snvc_sheared_expand_stats    : batch statistics of raw = expand(g, gcol) + planes, computed while walking the same values
 *                                  snvc_sheared_expand writes: scale / shift [C] for y = relu(scale*raw + shift) (gamma / beta
 *                                  may be NULL), mean / var [C] (may be NULL); fp64 sums, deterministic
 *   snvc_sheared_backward_reduce : everything the layer's backward needs from gy = dL/dy, in ONE pass over it; raw is recomputed
 *                                  from g / gcol / planes.  With g' = gy where scale*raw + shift > 0 (else 0):
 *                                    sums   [N][C][2] fp64 = (sum g', sum g'*raw)     -> snvc_bn_backward_coefs
 *                                    line   [2][N][3][C][H][WG]  sums of g' (0) and of raw (1) along every shear line, per depth
 *                                                                class, columns w <= W-2 (the layout of snvc_sheared_reduce's dg)
 *                                    lastc  [2][N][3][C][H][WG2] g' and raw of column W-1 by slot (dgcol's layout)
 *                                    colsum [2][N][C][3][H][W]   sums of g' and of raw over each depth class (the left half's planes)
 *                                  The BatchNorm backward is linear, draw = A*g' + B*raw + Cc per channel, so
 *                                  dg = A*line[0] + B*line[1] + Cc*(terms per line), and likewise dgcol and the planes: the
 *                                  caller combines 8 MB of sums; draw itself (736 MB) is never formed.  W % 8 == 0, W <= 512. */
SNVC_API int64_t snvc_sheared_stats_workspace_bytes(int64_t N, int64_t C, int64_t H);
SNVC_API int snvc_sheared_expand_stats(const float *g, const float *gcol, const float *planes, const float *gamma,
                                       const float *beta, float *scale, float *shift, float *mean, float *var,
                                       void *workspace, int64_t N, int64_t C, int64_t D, int64_t H, int64_t W, int q, int m0,
                                       int64_t WG, int off, int64_t WG2, int off2, float eps, void *stream);
SNVC_API int64_t snvc_sheared_backward_workspace_bytes(int64_t N, int64_t C, int64_t H);
SNVC_API int snvc_sheared_backward_reduce(const float *g, const float *gcol, const float *planes, const float *scale,
                                          const float *shift, const float *gy, float *line, float *colsum, float *lastc,
                                          double *sums, void *workspace, int64_t N, int64_t C, int64_t D, int64_t H,
                                          int64_t W, int q, int m0, int64_t WG, int off, int64_t WG2, int off2, void *stream);
SNVC_API int snvc_sheared_reduce(const float *dy, float *dg, float *dgcol, int64_t N, int64_t C, int64_t D, int64_t H,
                                 int64_t W, int q, int m0, int64_t WG, int off, int64_t WG2, int off2, void *stream);
SNVC_API int64_t snvc_sheared_wgrad_workspace_bytes(int64_t N, int64_t CO, int64_t H, int64_t WU);
SNVC_API int snvc_sheared_wgrad(const float *x, const float *dy, float *dk, void *workspace, int64_t N, int64_t C,
                                int64_t CO, int64_t H, int64_t WU, void *stream);
SNVC_API int snvc_sheared_upsample_backward(const float *drq, float *dright, int64_t N, int64_t C, int64_t H, int64_t W,
                                            int q, int64_t WU, int off, void *stream);
SNVC_API int64_t snvc_conv3d_wgrad_workspace_bytes(const snvc_conv3d_desc *desc_host);
SNVC_API int snvc_conv3d_wgrad(const snvc_conv3d_desc *desc_host, const float *x, const float *g, float *dw,
                               void *workspace, void *stream);
/* r6: snvc_conv3d_wgrad with the operands' maxima supplied: amax_x / amax_g point at SNVC_AMAX_SLOTS device words whose maximum is the
 * bit pattern of max|x| / max|g| (
 * e.g. written by snvc_affine_act_amax / snvc_act_backward_apply_amax when the tensors were produced); either may be NULL, the
 * split-operand form then finds that maximum itself with one more pass over the tensor (0.2 ms per 736 MB).  A supplied value must
 * be >= the true maximum (a larger one only costs precision: 2^k too large = k of the 39 bits below the maximum); the fp32
 * forms ignore both. */
SNVC_API int snvc_conv3d_wgrad_amax(const snvc_conv3d_desc *desc_host, const float *x, const float *g, float *dw,
                                    void *workspace, const uint32_t *amax_x, const uint32_t *amax_g, void *stream);
SNVC_API int64_t snvc_act_backward_workspace_bytes(int64_t N, int64_t C);
SNVC_API int snvc_act_backward_reduce(const float *raw, const float *gy, const float *residual,
                                      const float *scale, const float *shift, double *sums,
                                      void *workspace, int64_t N, int64_t C, int64_t S,
                                      int64_t raw_batch_stride, int64_t gy_batch_stride,
                                      int64_t res_batch_stride, int per_sample, int flags, void *stream);
SNVC_API int snvc_act_backward_apply(const float *raw, const float *gy, const float *residual,
                                     const float *scale, const float *shift, const float *coef_g,
                                     const float *coef_raw, const float *coef_const, float *draw,
                                     float *g_out, int64_t N, int64_t C, int64_t S,
                                     int64_t raw_batch_stride, int64_t gy_batch_stride,
                                     int64_t res_batch_stride, int per_sample, int flags, void *stream);
/* r6: + the bit pattern of max|draw| into *amax (see snvc_affine_act_amax). */
SNVC_API int snvc_act_backward_apply_amax(const float *raw, const float *gy, const float *residual,
                                          const float *scale, const float *shift, const float *coef_g,
                                          const float *coef_raw, const float *coef_const, float *draw,
                                          float *g_out, int64_t N, int64_t C, int64_t S,
                                          int64_t raw_batch_stride, int64_t gy_batch_stride,
                                          int64_t res_batch_stride, int per_sample, int flags, uint32_t *amax, void *stream);

/* r6 -- passes that also write the tensor's split C8 TWIN.  The training step's half- / quarter-resolution and transposed layers run
 * their forward and data-gradient convolutions on the split kernels (snvc_f16x3_conv3d_forward with y_f32), which read
 * [N][2 (hi | lo)][C/8][S][8] half pairs; the pass that produces the float32 NCDHW tensor (which the weight gradient, the statistics
 * and autograd keep using) writes the pair too, +4 bytes per element instead of a layout pass (snvc_f16x3_from_ncdhw, 8 bytes per
 * element).  twin value = v * twin_mul[0], hi = half(that), lo = half(that - hi); twin_mul: one device float, a power of two from
 * snvc_split_scale_bound (so no clamp and no overflow flag); twin_lo is the lo plane's address, twin_batch_stride in halves
 * (0 = 2*C*S).  C % 8 == 0, S % 4 == 0, 16-byte aligned tensors; everything else as snvc_affine_act_amax /
 * snvc_act_backward_apply_amax / snvc_act_backward_reduce (+ amax_gy: the bit pattern of max|gy| into SNVC_AMAX_SLOTS words). */
SNVC_API int snvc_affine_act_twin(const float *x, const float *scale, const float *shift, const float *residual, float *y,
                                  void *twin_hi, void *twin_lo, const float *twin_mul, int64_t N, int64_t C, int64_t S,
                                  int64_t x_batch_stride, int64_t y_batch_stride, int64_t res_batch_stride,
                                  int64_t twin_batch_stride, int per_sample, int flags, uint32_t *amax, void *stream);
SNVC_API int snvc_act_backward_apply_twin(const float *raw, const float *gy, const float *residual, const float *scale,
                                          const float *shift, const float *coef_g, const float *coef_raw, const float *coef_const,
                                          float *draw, float *g_out, void *twin_hi, void *twin_lo, const float *twin_mul, int64_t N,
                                          int64_t C, int64_t S, int64_t raw_batch_stride, int64_t gy_batch_stride,
                                          int64_t res_batch_stride, int64_t twin_batch_stride, int per_sample, int flags,
                                          uint32_t *amax, void *stream);
SNVC_API int snvc_act_backward_reduce_amax(const float *raw, const float *gy, const float *residual, const float *scale,
                                           const float *shift, double *sums, void *workspace, int64_t N, int64_t C, int64_t S,
                                           int64_t raw_batch_stride, int64_t gy_batch_stride, int64_t res_batch_stride,
                                           int per_sample, int flags, uint32_t *amax_gy, void *stream);
/* The scale of a twin from an upper bound of max|v| that is known BEFORE the pass runs (one small launch, no host round trip):
 *   bound = max over rows r of ( |a[r]| * P + |b[r]| * l1[r % C] * X + |c[r]| ) + R,   mul_out[0] = 2^k with bound * 2^k in [2^13, 2^14)
 * (1 for a zero or non-finite bound).  P / X / R: the maxima of the SNVC_AMAX_SLOTS-word arrays amax_p / amax_x / amax_r (NULL: 0; X: 1);
 * a / c NULL count 0, b / l1 NULL count 1; a, b, c hold `rows` floats (C, or N*C per-sample), l1 holds C.
 *   forward  y = act(scale*raw + shift [+res]) [+res]: b = scale, l1 = L1 norms of the layer's filters and X = max|x| (l1*X bounds |raw|),
 *            c = shift, R = max|res|;     backward draw = A*g + B*raw + Cc: a = A, P = max|gy|, b = B, l1, X as forward, c = Cc. */
SNVC_API int snvc_split_scale_bound(const float *a, const uint32_t *amax_p, const float *b, const float *l1, const uint32_t *amax_x,
                                    const float *c, const uint32_t *amax_r, int64_t rows, int64_t C, float *mul_out, void *stream);

/* replaces: nn.BatchNorm3d's train-mode bookkeeping (torch/nn/modules/batchnorm.py, reached from snvc/models/submodule.py:32-50's
 * BatchNorm3d layers): num_batches_tracked += 1 (may be NULL), running_mean / running_var <- lerp(running, batch, momentum) with the batch
 * variance multiplied by `unbias` = n / (n - 1) first -- one launch instead of four small ones per layer and step (r6).  float32 [C]. */
SNVC_API int snvc_bn_track(float *running_mean, float *running_var, int64_t *num_batches_tracked, const float *mean, const float *var,
                           int64_t C, float momentum, float unbias, void *stream);

/* Train-mode BatchNorm backward coefficients from snvc_act_backward_reduce's sums [N, C, 2] (fp64), in fp64, one launch
 * (replaces the ~15 per-channel tensor operations of torch autograd's native_batch_norm_backward on this path):
 *   sg = sum_n sums[n,c,0], sgr = sum_n sums[n,c,1], rstd = 1/sqrt(var[c] + eps), sgx = rstd * (sgr - mean[c] * sg)
 *   coef_g = gamma*rstd, coef_raw = -gamma*rstd^2*sgx/m, coef_const = -gamma*rstd*sg/m - coef_raw*mean   (m = count)
 *   dgamma = sgx, dbeta = sg.    gamma may be NULL (= 1); mean / var are the batch statistics of the forward. */
SNVC_API int snvc_bn_backward_coefs(const double *sums, const float *mean, const float *var, const float *gamma,
                                    float *coef_g, float *coef_raw, float *coef_const, float *dgamma, float *dbeta,
                                    int64_t N, int64_t C, double count, double eps, void *stream);

/* ------------------------------------------------------------------------------------
 * Small fused element-wise steps of predict_3d_heatmaps
 * ---------------------------------------------------------------------------------- */
/* replaces: torch.cat([voxel, voxel_img_feat * occupancy], dim=1)'s second half
 *   (snvc/models/vernier.py:433): out[n,c,s] = feat[n,c,s] * occ[n,0,s]; out may be a channel
 *   slice of the concat buffer (out_batch_stride). */
SNVC_API int snvc_mul_broadcast(const float *feat, const float *occ, float *out, int64_t N,
                                int64_t C, int64_t S, int64_t out_batch_stride, void *stream);
/* replaces: AvgPool3d((4,1,1),(4,1,1)) + reshape to BEV (vernier.py:289,436-438):
 *   x [N,C,D,H*W] -> y [N,C,D/4,H*W] (== [N, C*D/4, H, W] after a free reshape). */
SNVC_API int snvc_avgpool_depth4(const float *x, float *y, int64_t N, int64_t C, int64_t D,
                                 int64_t HW, void *stream);
/* its adjoint (training: the pool in front of the BEV reshape under autograd, vernier.py:436): grad_x[n,c,4q+j,i] = grad_y[n,c,q,i] / 4,
 * zero on the depth planes AvgPool3d's floor drops.  grad_y [N,C,D/4,HW], grad_x [N,C,D,HW], both dense fp32. */
SNVC_API int snvc_avgpool_depth4_backward(const float *grad_y, float *grad_x, int64_t N, int64_t C, int64_t D, int64_t HW,
                                          void *stream);
/* Zero-stuffing pass of the 2D up-sampling layers: x [R,H,W] -> y [R,2H,2W], y[r,2i,2j] = x[r,i,j], 0 elsewhere.
 * nn.ConvTranspose2d(k3,s2,p1,op1) of the BEV neck (snvc/models/submodule.py:291-314) == the depth-1 k3 / stride-1
 * convolution (desc.ksize_d = 1) of y with the flipped, channel-transposed kernel. */
SNVC_API int snvc_zero_stuff2x(const float *x, float *y, int64_t R, int64_t H, int64_t W, void *stream);
/* replaces: disparityregression.forward (snvc/models/submodule.py:81-83):
 *   out[n,h,w] = sum_d x[n,d,h,w] * depth[d]. */
SNVC_API int snvc_disparity_regression(const float *x, const float *depth, float *out, int64_t N,
                                       int64_t D, int64_t HW, void *stream);
/* replaces: np.argmax over the flattened heatmap (snvc/models/vernier.py:683-693, :570-572):
 *   x [R, L] -> idx [R] int64 (first maximum, numpy semantics; NaN counts as maximal like numpy),
 *   val [R] (may be NULL). */
SNVC_API int snvc_argmax_rows(const float *x, int64_t *idx, float *val, int64_t R, int64_t L,
                              void *stream);

/* ------------------------------------------------------------------------------------
 * fp16-storage mode of the local 3D trunk (BASELINE.json configs[4]: "High-res local model: ... 64ch, fp16
 * with MFMA").  The reference has no half-precision path: its kernels dispatch float / double only
 * (BuildCostVolume_cuda.cu:240) and VernierScale never calls .half().  These entry points run the same layers
 * -- snvc/models/vernier.py:249-264,414-438 and snvc/models/submodule.py:223-268 -- with activations and weights
 * stored as IEEE half, fp32 accumulation (v_mfma_f32_32x32x16_f16), fp32 epilogue, one rounding on the way out.
 * Parity target: this library's own fp32 path (tolerances in tests/test_gpu_f16.py).
 *
 * Tensor layout "C8": [N][C/8][D][H][W][8] half -- the 8 consecutive channels of a voxel are one 16-byte piece
 * (C a multiple of 8; pointers 16-byte aligned; batch strides in ELEMENTS, multiples of 8, 0 = dense).
 * Channel slices at multiples of 8 channels are contiguous, so torch.cat on channels (vernier.py:433) is free.
 * ---------------------------------------------------------------------------------- */
/* NCDHW fp32 <-> C8 half (S = D*H*W).  C need not be a multiple of 8 here: missing channels are zero / dropped. */
SNVC_API int snvc_f16_from_ncdhw(const float *x, void *y_c8, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                                 int64_t y_batch_stride, void *stream);
SNVC_API int snvc_f16_to_ncdhw(const void *x_c8, float *y, int64_t N, int64_t C, int64_t S, int64_t x_batch_stride,
                               int64_t y_batch_stride, void *stream);
/* replaces: _sample_2d_feat (vernier.py:323-349) with a C8 half result [N][2F/8][V][8]: same taps and the same
 * separately rounded fp32 bilinear sum as snvc_voxel_gather_forward_ws, rounded to half once.  F % 8 == 0;
 * workspace as for snvc_voxel_gather_forward_ws (required). */
SNVC_API int snvc_voxel_gather_forward_f16(const float *left, const float *right, const float *l_pts,
                                           const float *r_pts, void *out_c8, float *workspace, int64_t N, int64_t F,
                                           int64_t Hf, int64_t Wf, int64_t V, float res_x, float res_y, void *stream);
/* The same gather with the result written as a split C8 pair (out_hi, out_lo: [N][2F/8][V][8] half planes, batch stride in
 * halves, 0 = a dense [N][2][2F/8][V][8] pair) for a consumer on the split-mode kernels (snvc_f16x3_*): value * *mul_dev = hi + lo,
 * mul_dev a device scalar (a power of two with max|feature| * mul < 2^15: a bilinear sample never exceeds the features' maximum).
 * Bit-identical to snvc_voxel_gather_forward_ws followed by snvc_f16x3_from_ncdhw(mul_dev) without the fp32 tensor in between.
 * SNVC_ERR_UNSUPPORTED unless Hf * Wf <= 4608 and V >= 4096 (the LDS-staged form). */
SNVC_API int snvc_voxel_gather_forward_split(const float *left, const float *right, const float *l_pts, const float *r_pts,
                                             void *out_hi, void *out_lo, const float *mul_dev, float *workspace, int64_t N,
                                             int64_t F, int64_t Hf, int64_t Wf, int64_t V, int64_t out_batch_stride, float res_x,
                                             float res_y, void *stream);
/* replaces: nn.Conv3d / nn.ConvTranspose3d (+ folded eval BatchNorm) (+ residual) (+ ReLU) as snvc_conv3d_forward
 * does, on C8 half tensors.  desc as for snvc_conv3d_forward (Cin % 8 == 0; Cout % 32 == 0; the SNVC_EPI_* flags
 * except SIGMOID; desc.algo ignored; batch strides in half elements).  scale / bias / the epilogue are fp32.
 * Cout == 1 (k3, stride 1: the occupancy head, vernier.py:269-278): the single channel is written as an fp32
 * plane y_f32 [N][1][D][H][W] and SNVC_EPI_SIGMOID is honoured; y_c8 is then unused. */
SNVC_API int64_t snvc_f16_conv3d_packed_weight_bytes(const snvc_conv3d_desc *desc_host);
/* weight: the fp32 nn.Conv3d [Cout,Cin,k,k,k] / nn.ConvTranspose3d [Cin,Cout,3,3,3] parameter; rounded to half. */
SNVC_API int snvc_f16_conv3d_pack_weights(const snvc_conv3d_desc *desc_host, const float *weight, void *packed,
                                          void *stream);
SNVC_API int snvc_f16_conv3d_forward(const snvc_conv3d_desc *desc_host, const void *x_c8, const void *packed_weight,
                                     const float *scale, const float *bias, const void *residual_c8, void *y_c8,
                                     float *y_f32, void *stream);
/* ------------------------------------------------------------------------------------
 * r4 split mode ("f16x3"): the fp32 layers' contraction at fp32 accuracy on the half-precision matrix pipe.
 * A value travels as a PAIR of C8 half tensors, v * 2^e = hi + lo (hi = half(v * 2^e), lo = half(v * 2^e - hi): 22
 * significant bits; the exponent e is the caller's, a power of two keeps the split exact), weights likewise, and a
 * product is evaluated as hi_w*hi_x + lo_w*hi_x + hi_w*lo_x on three v_mfma_f32_32x32x16_f16 with fp32 accumulation
 * (the dropped lo*lo term is 2^-22 of the product: measured 5e-7 of the output range on an 864-term contraction, the
 * fp32 FMA chain's own figure).  Same layers, same epilogue and flags as snvc_conv3d_forward for
 * nn.Conv3d(k3, pad 1, stride 1 or 2) and nn.ConvTranspose3d(k3, s2, p1, op1) (+ folded eval BatchNorm) (+ residual) (+ ReLU)
 * -- snvc/models/submodule.py:32-50,127-146,170-208 -- with Cin % 8 == 0 and Cout % 32 == 0 (stride 2 / transposed: % 64).  The caller folds 2^-(e_x + e_w) (and 2^e_y for a split output) into
 * scale / bias.  Output: a split C8 pair (y_hi, y_lo; residual then a split pair too), or, with y_f32 != NULL, a plain
 * fp32 NCDHW tensor [N][Cout][D][H][W] for a consumer on the fp32 kernels (a residual stays a split pair).
 * Values beyond half's range after scaling (|v * 2^e| > 65504) overflow: the caller picks e from what it knows of the
 * tensor (folded BatchNorm statistics) -- see snvc_amd/ops.py.
 * ---------------------------------------------------------------------------------- */
/* mul_dev (device pointer, may be NULL): when given, the scale is read from device memory instead of `mul` -- a caller that
 * derives the exponent from the data (max |x| of a small tensor) does so without a host round trip.  Batch strides in elements
 * (0: dense; a dense split tensor is [N][2][C/8][S][8] halves, i.e. 2*C*S per sample). */
SNVC_API int snvc_f16x3_from_ncdhw(const float *x, void *y_hi, void *y_lo, int64_t N, int64_t C, int64_t S,
                                   int64_t x_batch_stride, int64_t y_batch_stride, float mul, const float *mul_dev,
                                   void *stream);
/* replaces: the norm + residual + activation of a convbn_3d(..., gn=True) layer (submodule.py:41-49, GroupNorm(32, C)) behind a
 * split-mode convolution with an fp32 result: x = the raw fp32 NCDHW conv result, scale / shift [N or 1][C] from its statistics
 * (snvc_norm_stats; per_sample = 1 for GroupNorm), res a split pair in units 2^-e_res (res_mul = 2^-e_res) or NULL,
 *   y = split( out_mul * ( act(scale * x + shift [+ res]) [+ res] ) ),   flags: SNVC_EPI_RELU, SNVC_EPI_ADD_PRE | SNVC_EPI_ADD_POST.
 * A value beyond half's range is clamped and sets *overflow (may be NULL).  Batch strides in elements, 0 = dense. */
SNVC_API int snvc_f16x3_affine_from_ncdhw(const float *x, const float *scale, const float *shift, const void *res_hi,
                                          const void *res_lo, void *y_hi, void *y_lo, int *overflow, int64_t N, int64_t C,
                                          int64_t S, int64_t x_batch_stride, int64_t y_batch_stride, int64_t res_batch_stride,
                                          int per_sample, int flags, float out_mul, float res_mul, void *stream);
/* replaces: torch.cat([voxel, voxel_img_feat * occupancy], dim=1)'s second half (vernier.py:433) on split pairs:
 *   out = split((hi + lo) * occ), occ an fp32 plane [N][S]; C % 8 == 0; the pair's exponent is unchanged. */
SNVC_API int snvc_f16x3_mul_broadcast(const void *feat_hi, const void *feat_lo, const float *occ, void *out_hi, void *out_lo,
                                      int64_t N, int64_t C, int64_t S, int64_t feat_batch_stride, int64_t out_batch_stride,
                                      void *stream);
SNVC_API int snvc_f16x3_to_ncdhw(const void *x_hi, const void *x_lo, float *y, int64_t N, int64_t C, int64_t S,
                                 int64_t x_batch_stride, int64_t y_batch_stride, float mul, void *stream);
SNVC_API int64_t snvc_f16x3_conv3d_packed_weight_bytes(const snvc_conv3d_desc *desc_host);
/* weight: the fp32 nn.Conv3d [Cout,Cin,3,3,3] parameter; packed as (hi, lo) of weight * wmul (wmul a power of two). */
SNVC_API int snvc_f16x3_conv3d_pack_weights(const snvc_conv3d_desc *desc_host, const float *weight, void *packed,
                                            float wmul, void *stream);
/* head / y_head (both or neither; 32-channel stride-1 layers with a split output): y_head[n][voxel] = head_mul * sum_c head[c] *
 * (the value stored for channel c) -- the classifier's projection of the layer's own result (the r3 side head of
 * snvc_conv3d_forward_side_head); head_mul = 2^-e_y (with y_f32 the stored result is multiplied by head_mul as well: the epilogue
 * then works in the residual's units 2^e_y).  res_mul = 2^(e_y - e_res): the residual pair's stored units relative to the result's (1 if they
 * share an exponent).  overflow (device int, may be NULL): set to 1 if a value had to be
 * clamped to half's range on the way out (the exponent the caller chose was too large for this input).
 * r6: with y_f32 and SNVC_EPI_ADD_POST, res_hi may be a FLOAT32 NCDHW tensor of the result's shape and batch stride (res_lo = NULL):
 * y_f32 = head_mul * act(scale * conv + bias) + res -- the training step's data gradients take a skip connection's gradient this way. */
SNVC_API int snvc_f16x3_conv3d_forward(const snvc_conv3d_desc *desc_host, const void *x_hi, const void *x_lo,
                                       const void *packed_weight, const float *scale, const float *bias,
                                       const void *res_hi, const void *res_lo, void *y_hi, void *y_lo, float *y_f32,
                                       const float *head, float *y_head, float head_mul, float res_mul, int *overflow,
                                       void *stream);
/* r6: the split layer with a float32 result + the batch statistics of that result in the same launch (train-mode BatchNorm behind a
 * layer of the training step that runs on the split kernels): as snvc_conv3d_forward_stats, for the kernel forms that write
 * float32 through the shared epilogue (3x3x3 stride 1 / 2 and the transposed layer, 256-thread forms).  scale / bias / head_mul as
 * snvc_f16x3_conv3d_forward (the statistics are those of the STORED values); desc.flags == 0, Cout % 32 == 0.
 * SNVC_ERR_UNSUPPORTED (nothing launched) when the layer does not take such a form.
 * x_mul (may be NULL): ONE device float, the power of two the input pair was multiplied by (snvc_split_scale_bound /
 * snvc_f16x3_split_scale); the epilogue takes it out of the channel scale exactly -- no host read, no launch to fold it into `scale`. */
SNVC_API int64_t snvc_f16x3_conv3d_stats_workspace_bytes(const snvc_conv3d_desc *desc_host);
SNVC_API int snvc_f16x3_conv3d_forward_stats(const snvc_conv3d_desc *desc_host, const void *x_hi, const void *x_lo,
                                             const void *packed_weight, const float *scale, const float *bias, const float *x_mul,
                                             float *y_f32, float head_mul, const float *gamma, const float *beta, float *bn_scale,
                                             float *bn_shift, float *mean, float *var, void *workspace, float eps, void *stream);
/* The same layer without the statistics: y_f32 = out_mul * act(scale / x_mul[0] * conv + bias) [+ res_f32] -- the float32-output form of
 * snvc_f16x3_conv3d_forward with its arguments named (res_f32: a float32 NCDHW tensor of the result's shape and batch stride, or NULL;
 * desc.flags: SNVC_EPI_RELU or 0).  The training step's data gradients run through this. */
SNVC_API int snvc_f16x3_conv3d_forward_f32(const snvc_conv3d_desc *desc_host, const void *x_hi, const void *x_lo,
                                           const void *packed_weight, const float *scale, const float *bias, const float *x_mul,
                                           const float *res_f32, float *y_f32, float out_mul, void *stream);
/* The global stack's tail -- classifier(bn(conv6(post)) + v), reference snvc/models/submodule.py:127-146,166 with the composition
 * of snvc/models/vernier.py:366-371 -- is linear in `post` (conv6 has no activation): a transposed layer (k3, s2, p1, op1) to ONE
 * channel with folded weights W'[c][kd][kh][kw] (c over post's channels) and a scalar bias.  Three entry points (r5) evaluate it
 * WITHOUT storing `post` = relu(bn(conv5(x)) + pre) (reference submodule.py:161-164):
 *   snvc_f16x3_tail_pack_weights:      W' ([Cin][27] fp32, Cin % 32 == 0) as split MFMA A fragments, values * wmul (a power of two);
 *   snvc_f16x3_deconv3d_tail_forward:  the split transposed layer conv5 (same arguments as snvc_f16x3_conv3d_forward: desc of a
 *       ConvTranspose3d(k3,s2,p1,op1) with Cout = 32 or 64, folded scale / bias, residual pair, flags) whose epilogue forms the result in
 *       units 2^e_y (clamped to half's range and flagged like every split output), splits it and contracts it per voxel with W':
 *       t_out[n][tap = (kd*3+kh)*3+kw][class][pd][ph][pw] = tail_mul * sum_c W'[c][tap] * post[n][c][2pd+rd][2ph+rh][2pw+rw],
 *       class = rd*4 + rh*2 + rw, (pd, ph, pw) over the layer's INPUT grid; fp32, dense; tail_mul = 2^-(e_y + w_exp);
 *   snvc_deconv_tail_gather:           y[n][o] = bias[0] + residual[n][o] + sum over (i, k) with o = 2 i - 1 + k per dimension of
 *       t[n][k][i]; (nd, nh, nw) = the grid of t's classes, y / residual [N][4nd][4nh][4nw] fp32 (residual, bias may be NULL). */
/* r5: the depth-1 (2D) layers of the sheared first convolution in split mode -- G / G' = a 3x7 convolution of the upsampled right
 * feature, the left half's depth-class planes = a 3x3 one (reference: the composition BuildCostVolume_cuda.cu:63-98 +
 * submodule.py:32-50) -- three v_mfma_f32_16x16x32_f16 per fp32 product instead of the fp32 matrix pipe.
 *   snvc_sheared_upsample_split: snvc_sheared_upsample writing the split pair [N][2][C/8][H][WU][8], value * *mul_dev = hi + lo;
 *   snvc_f16x3_conv2d_pack_weights: weight [Cout][Cin][kh][kw] fp32 ((kh, kw) in {(3,7), (3,3)}, Cin % 8 == 0, Cout % 32 == 0) * wmul;
 *   snvc_f16x3_conv2d_forward: y[n][co][h][w] = act(scale[co] * out_mul / *x_mul_dev * conv(x)[..] + bias[co]), fp32, stride 1, "same"
 *     zero padding; x = the split pair (hi, lo planes), x_mul_dev (device scalar, may be NULL = 1) what it was multiplied by. */
/* out_mul[0] = the power of two that puts max|x| (n floats, 16-byte aligned) into [2^13, 2^14); 1 for an all-zero or non-finite tensor.
 * One launch, no host round trip; scratch8 = 8 bytes of device memory, ZERO before the first call (the kernel leaves it zero). */
SNVC_API int snvc_f16x3_split_scale(const float *x, int64_t n, void *scratch8, float *out_mul, void *stream);
SNVC_API int snvc_sheared_upsample_split(const float *right, void *y_hi, void *y_lo, const float *mul_dev, int64_t N, int64_t C,
                                         int64_t H, int64_t W, int q, int64_t WU, int off, void *stream);
/* all of it in one host call (snvc_f16x3_split_scale, two snvc_sheared_upsample_split, two 3x7 snvc_f16x3_conv2d_forward): rq_split /
 * rq2_split = workspaces of N*2*C*H*WU / N*2*C*H*WU2 halves, g / gcol [N][Cout3][H][WU | WU2] fp32, out_mul_* = 2^-w_exp of the packings */
SNVC_API int snvc_sheared_prep_x3(const float *right, int64_t N, int64_t C, int64_t H, int64_t W, int q, int64_t WU, int off,
                                  int64_t WU2, int off2, const void *packed_g, const void *packed_col, int64_t Cout3, float out_mul_g,
                                  float out_mul_col, void *rq_split, void *rq2_split, void *scratch8, float *mul_dev, float *g,
                                  float *gcol, void *stream);
/* n_layers (1..8) depth-1 split layers of ONE fp32 [N][C][H][W] input in one host call: scale, split pair (ws_split: N*2*C*H*W halves),
 * then layer i: y[i] [N][cout[i]][H][W] = out_mul[i] / *mul_dev * conv(x, packed[i]).  packed / cout / out_mul / y are HOST arrays. */
SNVC_API int snvc_f16x3_conv2d_from_f32(const float *x, int64_t N, int64_t C, int64_t H, int64_t W, int kh, int kw, int n_layers,
                                        const void *const *packed, const int64_t *cout, const float *out_mul, float *const *y,
                                        void *ws_split, void *scratch8, float *mul_dev, void *stream);
SNVC_API int64_t snvc_f16x3_conv2d_packed_weight_bytes(int cout, int cin, int kh, int kw);
SNVC_API int snvc_f16x3_conv2d_pack_weights(const float *weight, int cout, int cin, int kh, int kw, void *packed, float wmul,
                                            void *stream);
SNVC_API int snvc_f16x3_conv2d_forward(const void *x_hi, const void *x_lo, const void *packed_weight, const float *scale,
                                       const float *bias, float *y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                                       int kh, int kw, float out_mul, const float *x_mul_dev, int flags, void *stream);
SNVC_API int64_t snvc_f16x3_tail_packed_weight_bytes(int cin);
SNVC_API int snvc_f16x3_tail_pack_weights(const float *weight, int cin, void *packed, float wmul, void *stream);
SNVC_API int snvc_f16x3_deconv3d_tail_forward(const snvc_conv3d_desc *desc_host, const void *x_hi, const void *x_lo,
                                              const void *packed_weight, const float *scale, const float *bias,
                                              const void *res_hi, const void *res_lo, float res_mul, const void *tail_packed,
                                              float *t_out, float tail_mul, int *overflow, void *stream);
SNVC_API int snvc_deconv_tail_gather(const float *t, const float *bias, const float *residual, float *y, int64_t N,
                                     int64_t nd, int64_t nh, int64_t nw, void *stream);
/* replaces: torch.cat([voxel, voxel_img_feat * occupancy], dim=1)'s second half (vernier.py:433) on C8 tensors:
 *   out[n,c,s] = half(float(feat[n,c,s]) * occ[n,0,s]), occ an fp32 plane [N][S]; C % 8 == 0. */
SNVC_API int snvc_f16_mul_broadcast(const void *feat_c8, const float *occ, void *out_c8, int64_t N, int64_t C,
                                    int64_t S, int64_t feat_batch_stride, int64_t out_batch_stride, void *stream);
/* replaces: AvgPool3d((4,1,1),(4,1,1)) + reshape to BEV (vernier.py:289,436-438): C8 [N][C/8][D][HW][8] ->
 *   fp32 [N,C,D/4,HW] (== [N, C*D/4, H, W] after a free reshape), the layout the 2D neck reads. */
SNVC_API int snvc_f16_avgpool_depth4(const void *x_c8, float *y, int64_t N, int64_t C, int64_t D, int64_t HW,
                                     int64_t x_batch_stride, void *stream);

/* ------------------------------------------------------------------------------------
 * N3  plane-sweep volume -> 3D grid resampling (global model)
 * The consumer of build_cost_volume is not in the public reference (snvc/models/__init__.py:1-2); its helpers are:
 * project_rect_to_image / project_image_to_rect / project_disp_to_depth_new (snvc/utils/torch_utils.py:5-45) and
 * disparityregression (snvc/models/submodule.py:76-83, snvc_disparity_regression above).  The resampling step is
 * defined against 5-D F.grid_sample(mode="bilinear" (= trilinear), padding_mode="zeros"); PARITY UNPINNED by the
 * reference -- the oracle is torch's own grid_sample.
 *   x [N,C,D,H,W]   grid [N,V,3] = (gx -> W, gy -> H, gz -> D) in [-1,1]   out [N,C,V]
 * ---------------------------------------------------------------------------------- */
SNVC_API int snvc_volume_resample(const float *x, const float *grid, float *out, int64_t N, int64_t C, int64_t D,
                                  int64_t H, int64_t W, int64_t V, int align_corners, int64_t x_batch_stride,
                                  int64_t out_batch_stride, void *stream);
/* replaces: project_rect_to_image (snvc/utils/torch_utils.py:37-45) + the normalisation of (u, v, z) to grid_sample's
 *   cube: pts_rect [V,3] -> grid [V,3]; g = (value - origin) / span * 2 - 1 per axis.  P_host: HOST pointer, 12 floats
 *   (the 3x4 projection matrix, row major). */
SNVC_API int snvc_rect_to_psv_grid(const float *pts_rect, const float *P_host, float *grid, int64_t V, float u0,
                                   float u_span, float v0, float v_span, float z0, float z_span, void *stream);

/* ------------------------------------------------------------------------------------
 * a10  roiaware_pool3d
 * replaces: roiaware_pool3d_cuda.forward / backward / points_in_boxes_gpu / points_in_boxes_cpu
 *   (snvc/extension/roiaware_pool3d/src/roiaware_pool3d.cpp:29-177; kernels
 *    roiaware_pool3d_kernel.cu:39-336).  Same contract: the caller pre-zeroes argmax,
 *   pts_idx_of_voxels, pooled (roiaware_pool3d_utils.py:124-126) and grad_in (:142).
 *   rois [B,7]  pts [P,3]  feat [P,C]  argmax/pooled [B,ox,oy,oz,C]
 *   pts_idx_of_voxels [B,ox,oy,oz,max_pts] (slot 0 = count).  pool_method 0 max / 1 avg.
 *   workspace: B*P int32 (the reference cudaMalloc's it per call, :203-205,228).
 * ---------------------------------------------------------------------------------- */
SNVC_API int snvc_roiaware_pool3d_forward(const float *rois, const float *pts, const float *feat,
                                          int32_t *argmax, int32_t *pts_idx_of_voxels,
                                          float *pooled, int32_t *workspace, int B, int P, int C,
                                          int max_pts, int ox, int oy, int oz, int pool_method,
                                          void *stream);
SNVC_API int snvc_roiaware_pool3d_backward(const int32_t *pts_idx_of_voxels, const int32_t *argmax,
                                           const float *grad_out, float *grad_in, int B, int C,
                                           int max_pts, int ox, int oy, int oz, int pool_method,
                                           void *stream);
/* boxes [Bs,T,7], pts [Bs,M,3], out [Bs,M] pre-filled with -1 by the caller. */
SNVC_API int snvc_points_in_boxes_gpu(const float *boxes, const float *pts, int32_t *out, int Bs,
                                      int T, int M, void *stream);
/* HOST pointers (the reference op is a CPU function): boxes [T,7], pts [M,3], out [T,M]. */
SNVC_API int snvc_points_in_boxes_cpu(const float *boxes_host, const float *pts_host,
                                      int32_t *out_host, int T, int M);

/* ------------------------------------------------------------------------------------
 * N4  KITTI object AP / AOS evaluator core (HOST function: every pointer is a host pointer, no device work, no stream)
 * replaces: tools/kitti-eval/evaluate_object_3d_offline.cpp / evaluate_object_3d_offline_r40.cpp -- imageBoxOverlap,
 *   groundBoxOverlap, box3DOverlap (:227-344; the rotated-box overlaps without Boost.Geometry), getThresholds (:346-379),
 *   cleanData (:381-454), computeStatistics (:456-616), eval_class (:622-706) and the class / metric loop of eval
 *   (:853-911).  File parsing, the "which tables" flags of loadDetections (:159-170) and the AP read-out (:719-723,
 *   R40: mean of recall points 1..40; R11: points 0,4,..,40) live in snvc_amd/evaluate.py.
 *   gt  [G][14] doubles: truncation, occlusion, alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry   (label_2 columns 2..15)
 *   det [M][13] doubles: alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry, score                    (result columns 4..16)
 *   gt_type / det_type: SNVC_KITTI_* codes; gt_offsets / det_offsets [frames + 1]: rows of frame k are [off[k], off[k+1])
 *   min_overlap [3 metrics: image, ground, 3D][3 classes: car, pedestrian, cyclist]   (the tool's MIN_OVERLAP, :55)
 *   evaluate    [3 metrics][3 classes] != 0: fill that table (the tool's eval_image / eval_ground / eval_3d)
 *   precision   [3 metrics][3 classes][3 difficulties: easy, moderate, hard][41 recall points]   (tables not asked for: 0)
 *   aos         [3 classes][3 difficulties][41] or NULL; written when compute_aos != 0 (image metric only, :876-877)
 *   threads     worker threads (0: one per hardware thread, at most 32).
 * ---------------------------------------------------------------------------------- */
enum {
    SNVC_KITTI_CAR = 0, SNVC_KITTI_PEDESTRIAN = 1, SNVC_KITTI_CYCLIST = 2, SNVC_KITTI_VAN = 3,
    SNVC_KITTI_PERSON_SITTING = 4, SNVC_KITTI_DONTCARE = 5, SNVC_KITTI_OTHER = 6
};
SNVC_API int snvc_kitti_eval(const double *gt, const int32_t *gt_type, const int64_t *gt_offsets, const double *det,
                             const int32_t *det_type, const int64_t *det_offsets, int64_t frames,
                             const double *min_overlap, const int32_t *evaluate, int compute_aos, double *precision,
                             double *aos, int threads);

#ifdef __cplusplus
}
#endif
#endif /* SNVC_HIP_H */
