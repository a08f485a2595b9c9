/*
 * voxelnet_hip.h — C ABI of libvoxelnet_hip.so (gfx950 / MI355X).
 *
 * The reference (johanngerberding/voxelnet-pytorch) has no FFI or operator
 * registry: its hot path is Python calling ATen/NumPy.  This header is the
 * boundary a maintainer would bind (ctypes stub in INTEGRATION.md) to replace
 * those calls; every entry point cites the reference code it stands in for
 * (paths relative to /root/reference/voxelnet/).
 *
 * Conventions:
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - the library never allocates or frees caller-visible memory: outputs and
 *     workspaces are caller-allocated; *_workspace_bytes() say how much.
 *   - asynchronous: work is enqueued on `stream` (a hipStream_t passed as
 *     void*) and ordered only by it.  No entry point synchronises.
 *   - return value: 0 ok, <0 invalid argument (VN_E*), >0 a hipError_t.
 *   - no global mutable state; re-entrant.
 *   - activations are channels-last: (B, D, H, W, C) / (B, H, W, C).
 */
#ifndef VOXELNET_HIP_H
#define VOXELNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VN_OK 0
#define VN_EINVAL (-1)     /* bad argument (null pointer, negative size, ...) */
#define VN_EUNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define VN_EWORKSPACE (-3) /* workspace too small */

typedef enum { VN_F32 = 0, VN_BF16 = 1,
               /* OPERAND dtype of the conv / weight-gradient entry points only (vnConv.dtype): fp32 storage — sources and
                * rows exactly as VN_F32 — with every product evaluated as three bf16 MFMAs on hi / lo splits (a.b ~=
                * ah.bh + al.bh + ah.bl, ~2^-16 per product): the "fp32x3" mode, ~1e-4 on the RPN maps at a fraction of the
                * exact fp32 MFMA cost (round 4).  Sources / rows are split in registers; PACKED WEIGHTS of a convolution
                * launched with this dtype must come from vn_pack_weight(s_batch) with packed_dtype VN_F32X3, which splits
                * them once (same bytes as fp32; the layout is described there).  Tensor dtypes (outputs, BatchNorm, ...)
                * never take this value. */
               VN_F32X3 = 2,
               /* "split fp32" STORAGE (round 5): a tensor of C % 8 == 0 channels at 4 bytes per element, every aligned group of 8
                * channels = 32 B holding the eight hi bf16 parts (hi = bf16(x)) followed by the eight lo parts (lo = bf16(x -
                * hi)) — x to ~2^-17, the precision the VN_F32X3 products use of an fp32 operand anyway.  Written by the
                * BatchNorm passes (vn_bn_apply / vn_bn_bwd_apply and their BEV forms: a_dtype / dy_dtype), read as vnConv.dtype
                * by the conv entry points (source rows; the packed weights are the VN_F32X3 ones, same format) and the
                * weight-gradient entry points (BOTH operands): the fp32x3 products without any split work in the kernels.
                * Element strides count 4-byte elements and must be multiples of 8; Cs % 32 == 0 for a convolution source. */
               VN_F32X3S = 3 } vnDtype;

typedef void *vnStream; /* hipStream_t */

int vn_abi_version(void); /* bumps when a signature, a data layout or a call protocol changes (4: round 5) */
const char *vn_build_info(void); /* "gfx950 <date> ..." static string */
const char *vn_build_id(void);   /* 12 hex digits: SHA-256 over the library's sources (changes with ANY source change) */

/* ------------------------------------------------------------------------
 * Voxelizer — utils.py:10-100 (pcl_to_voxels), minus the host-side shuffle
 * (utils.py:35), which the Python wrapper performs before upload.
 * Grid literals: utils.py:24-33.
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t D, H, W;   /* grid (z,y,x) */
    float vz, vy, vx;  /* voxel size (z,y,x), float32 as in the reference */
    float ox, oy, oz;  /* lidar_coord added to (x,y,z) */
    int32_t T;         /* max points per voxel (<= 64) */
} vnGrid;

size_t vn_voxelize_workspace_bytes(int64_t n_points, const vnGrid *grid);

/* Phase 1 (utils.py:37-63): per-point voxel key (exact fp32 add / IEEE divide /
 * floor), occupancy, ordered compaction.  Writes K (number of non-empty voxels,
 * rows sorted by z,y,x) to *k_out (device int32).  The caller reads K back
 * (one 4-byte copy) to size phase 2's outputs — the reference returns
 * (K,T,7)/(K,3)/(K,) arrays, so K is part of the boundary format. */
int vn_voxelize_index(const float *points /*[N,4]*/, int64_t n_points,
                      const vnGrid *grid, void *workspace, size_t workspace_bytes,
                      int32_t *k_out, vnStream stream);

/* Phase 2 (utils.py:69-88 + dataset.py:110-117): first-T points per voxel in
 * input order, centroid offsets on all T slots (float64 divide/subtract as
 * numpy promotes), int64 coordinates with `coord_cols` = 3 (z,y,x) or
 * 4 (batch_index,z,y,x; prepare_voxel's padding), int64 counts.
 * Deterministic: no order-dependent atomics reach the outputs.
 * k_dev == NULL: K is the exact row count (read back from phase 1).  k_dev != NULL
 * (phase 1's k_out): K is only the CAPACITY of the output buffers (any K <= n_points
 * fits) and rows >= *k_dev are left untouched — no host read-back between the phases. */
int vn_voxelize_gather(const float *points, int64_t n_points, const vnGrid *grid,
                       void *workspace, size_t workspace_bytes, int64_t K,
                       int64_t batch_index, int32_t coord_cols,
                       float *feature /*[K,T,7]*/, int64_t *coord /*[K,coord_cols]*/,
                       int64_t *number /*[K]*/, const int32_t *k_dev, vnStream stream);

/* Host (CPU) variant of the two phases for the reference's own call site — pcl_to_voxels inside forked DataLoader
 * worker processes (dataset.py:58, train.py:77-84), which cannot touch the GPU.  HOST pointers, no stream, synchronous;
 * same arithmetic, row order and output formats as the device pair above, bit for bit.  The index phase leaves its
 * row table in `workspace`; the gather phase must be given the K it returned. */
size_t vn_voxelize_host_workspace_bytes(int64_t n_points, const vnGrid *grid);
int vn_voxelize_host_index(const float *points /*host [N,4]*/, int64_t n_points, const vnGrid *grid,
                           void *workspace /*host*/, size_t workspace_bytes, int64_t *k_out);
int vn_voxelize_host_gather(const float *points, int64_t n_points, const vnGrid *grid, const void *workspace,
                            size_t workspace_bytes, int64_t K, int64_t batch_index, int32_t coord_cols,
                            float *feature /*host [K,T,7]*/, int64_t *coord /*host [K,coord_cols]*/,
                            int64_t *number /*host [K]*/);

/* ------------------------------------------------------------------------
 * Voxel feature encoder — FeatureLearningNet.forward up to the scatter
 * (model.py:93-100) incl. both VFELayer.forward calls (model.py:74-82):
 *   mask = max_c(x) != 0 ; h1 = relu(x W1^T + b1) ; p1 = BN1d(h1) over all K*T rows
 *   out1 = [p1, max_T p1] * mask ; h2 = relu(out1 W2^T + b2) ; p2 = BN1d(h2)
 *   out2 = [p2, max_T p2] * mask ; voxelwise = max_T out2            (K,128)
 * Shapes are the reference's: W1 (16,7), W2 (64,32), T <= 64.
 * Train mode uses batch statistics (biased variance) and updates the running
 * statistics (momentum, unbiased variance) exactly like nn.BatchNorm1d.
 * stats (320 floats: [mean|invstd|gamma*invstd|beta] of BN1 then BN2) is saved
 * for the backward.  Deterministic (fixed-order slab reductions, no float atomics).
 * ---------------------------------------------------------------------- */
typedef struct {
    const float *w1, *b1, *g1, *be1; float *rm1, *rv1;   /* vfe_1: fcn.0.weight/bias, bn.weight/bias, running stats */
    const float *w2, *b2, *g2, *be2; float *rm2, *rv2;   /* vfe_2 */
} vnVfeWeights;
typedef struct {
    float *dw1, *db1, *dg1, *dbe1, *dw2, *db2, *dg2, *dbe2;   /* overwritten */
} vnVfeGrads;

size_t vn_vfe_workspace_bytes(int64_t K, int32_t T);
int vn_vfe_fwd(const float *feature /*[K,T,7]*/, int64_t K, int32_t T, const vnVfeWeights *w,
               int32_t training, float momentum, float eps, float *voxelwise /*[K,128]*/,
               float *stats /*[320]*/, void *workspace, size_t workspace_bytes, vnStream stream);
/* vn_vfe_fwd that ALSO writes the voxel features as bf16 rows (K, 128) — bf16(x) of every element of `voxelwise`, what
 * vn_cast_rows makes of it — for a caller whose next consumer reads bf16 (vn_net_step in the bf16 mode: the first Conv3d's
 * rulebook GEMM; one launch less at the start of the step's dependency chain). */
int vn_vfe_fwd_rows(const float *feature, int64_t K, int32_t T, const vnVfeWeights *w, int32_t training, float momentum,
                    float eps, float *voxelwise, void *rows_bf16, float *stats, void *workspace, size_t workspace_bytes,
                    vnStream stream);
/* gradients of all eight parameter tensors for upstream d_voxelwise (K,128); the input
 * features are leaf data (no d_feature).  `workspace` need not be the forward's; when it IS — the buffer vn_vfe_fwd
 * was given for the same feature / K / T, untouched since — pass workspace_is_forwards = 1 (bit 0) and the effective-row
 * work list found there is reused instead of rebuilt (two launches less).  Bit 1 of workspace_is_forwards: the forward
 * ran with training = 0 (stats = the running statistics): the eval-mode BatchNorm backward, whose batch-statistic terms
 * vanish (the reference's autograd through `model.eval()`; model.py:76 with self.training False). */
int vn_vfe_bwd(const float *feature, int64_t K, int32_t T, const vnVfeWeights *w, const float *stats,
               const float *d_voxelwise, const vnVfeGrads *g, void *workspace,
               size_t workspace_bytes, int32_t workspace_is_forwards, vnStream stream);

/* Stand-alone VFELayer.forward(inputs, mask) (model.py:60-82) for callers that compose the layers themselves:
 * out (K,T,2*units) = concat(p, max_t p) * mask with p = BatchNorm1d(relu(inputs W^T + b)) over the K*T rows
 * (training: batch statistics + running-stat update with `momentum`; else the running statistics).  mask: one
 * byte per (voxel, slot) row.  cin, units <= 64.  The backward reads the forward's workspace (same pointer, untouched
 * in between) and returns the gradients of inputs (may be NULL), weight (units,cin), bias, gamma, beta. */
size_t vn_vfe_layer_workspace_bytes(int64_t K, int32_t T, int32_t cin, int32_t units);
int vn_vfe_layer_fwd(const float *inputs /*[K,T,cin]*/, const uint8_t *mask /*[K,T]*/, int64_t K, int32_t T,
                     int32_t cin, int32_t units, const float *weight, const float *bias, const float *gamma,
                     const float *beta, float *running_mean, float *running_var, int32_t training, float momentum,
                     float eps, float *out /*[K,T,2*units]*/, void *workspace, size_t workspace_bytes, vnStream stream);
int vn_vfe_layer_bwd(const float *inputs, const uint8_t *mask, const float *d_out /*[K,T,2*units]*/, int64_t K,
                     int32_t T, int32_t cin, int32_t units, const float *weight, int32_t training, float *d_inputs,
                     float *d_weight, float *d_bias, float *d_gamma, float *d_beta, void *workspace,
                     size_t workspace_bytes, vnStream stream);

/* ------------------------------------------------------------------------
 * Sparse -> dense scatter — model.py:102-106 (sparse COO .to_dense()).
 * dense[b,z,y,x,:] = voxelwise[k,:]; zero elsewhere.  Coordinates unique.
 * dense is f32 (B,D,H,W,C), bf16, or (split != 0) bf16 [hi|lo] with dense_channels = 2C.
 * Backward = row gather.  Deterministic.
 * ---------------------------------------------------------------------- */
int vn_scatter_dense_fwd(const float *voxelwise /*[K,C]*/, const int64_t *coord /*[K,4]*/,
                         int64_t K, int32_t C, int32_t B, int32_t D, int32_t H, int32_t W,
                         void *dense, vnDtype dense_dtype, int32_t dense_channels,
                         int32_t split, vnStream stream);
/* The same scatter WITHOUT the zero fill, for a dense buffer the caller keeps all-zero between steps: writes the K
 * voxel rows, or — voxelwise == NULL — zeros at those rows (undoing the previous call).  A persistent grid then costs
 * 2 x K rows of writes per step instead of a 721 MB fill. */
int vn_scatter_dense_update(const float *voxelwise /* (K,C) or NULL */, const int64_t *coord, int64_t K,
                            int32_t C, int32_t B, int32_t D, int32_t H, int32_t W, void *dense,
                            vnDtype dense_dtype, int32_t dense_channels, int32_t split, vnStream stream);

int vn_scatter_dense_bwd(const void *d_dense, vnDtype dtype, const int64_t *coord,
                         int64_t K, int32_t C, int32_t B, int32_t D, int32_t H, int32_t W,
                         float *d_voxelwise /*[K,C]*/, vnStream stream);

/* ------------------------------------------------------------------------
 * Convolutions — model.py:111-199 (ConvMD: Conv2d/Conv3d; DeConv2d:
 * ConvTranspose2d) as ONE gather-GEMM on the matrix cores.
 *
 *   out[m, n] = bias[n] + sum_{tap, k} src[site(m, tap), k] * w[tap][n][k]
 *
 * m runs over the "row" sites (B, Dr, Hr, Wr); site(m, tap) is, per axis,
 *   s = (row * mul + tap * tmul - pad) / div   (valid iff divisible, 0 <= s < S)
 * and an invalid site contributes zero.  Forward conv (mul = stride, tmul = 1,
 * div = 1), the data-gradient of a strided conv and ConvTranspose2d forward
 * (mul = 1, tmul = -1, pad = -p, div = stride), and the data-gradient of a
 * ConvTranspose2d are all instances; the host code picks the geometry and the
 * packed weight orientation.  div > 1 is executed as div^3 (div^2) residue
 * classes of a stride-1 gather, all inside one launch.
 *
 * Operands are bf16 with fp32 accumulation on v_mfma_f32_16x16x32_bf16.
 * (The "split" bf16x3 mode of rounds 1-4 — [hi | lo] 2C-wide rows, weights [hi ; hi ; lo], src_wrap = 2C — was retired in
 * round 5: vnConv.src_wrap, the `split` argument of the weight-gradient entry points and vn_pack_weight's split3 must be 0
 * (VN_EUNSUPPORTED otherwise).  The fp32-accurate products are VN_F32X3 / VN_F32X3S: same error, one launch.)
 * Addresses are explicit strides in ELEMENTS, so channel slices of a wider
 * buffer (the 768-channel concat, model.py:271-273) and the BEV fold
 * (model.py:262) are views, not copies.
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t dtype;        /* operand type of src / w_packed (/ rows in wgrad): VN_BF16, or VN_F32 =
                             exact fp32 products on v_mfma_f32_16x16x4_f32 (1/16 of the bf16 rate) */
    int32_t B;
    int32_t Ds, Hs, Ws;   /* source (gathered) tensor sites */
    int32_t Dr, Hr, Wr;   /* row (produced) tensor sites */
    int32_t Cs;           /* GEMM K per tap (multiple of 8 bf16 / 4 fp32 elements; 3C in split mode) */
    int32_t src_wrap;     /* must be 0 (the K wrap of the retired bf16x3 mode) */
    int32_t Cr;           /* GEMM N = row channels (multiple of 4) */
    int32_t kD, kH, kW;   /* taps */
    int32_t mulD, mulH, mulW;
    int32_t tmulD, tmulH, tmulW;
    int32_t padD, padH, padW;
    int32_t divD, divH, divW;
    int64_t src_sB, src_sD, src_sH, src_sW; /* source strides, elements */
    int64_t out_sB, out_sD, out_sH, out_sW; /* row/output strides, elements */
} vnConv;

/* stats_slab: NULL, or float[vn_conv_stats_slab_rows(geom)][2][Cr]: every workgroup
 * writes the per-channel sum / sum of squares of (out - bias) over its valid rows
 * (plain stores, deterministic) — the train-mode BatchNorm reduction fused into the
 * epilogue; vn_bn_finalize_slab reduces it.  Only for div == 1 geometries. */
int64_t vn_conv_stats_slab_rows(const vnConv *geom);
int vn_conv_gather_gemm(const void *src, const void *w_packed /*[taps][Cr][Cs]*/,
                        const float *bias /*[Cr] or NULL*/, void *out, vnDtype out_dtype,
                        const vnConv *geom, int32_t accumulate, float *stats_slab,
                        vnStream stream);
/* Introspection for the parity tests (no launch): which kernel / tile vn_conv_gather_gemm picks for `geom`:
 * 100 + k_conv_patch tile id (0 = 10x16, 1 = 8x32, 2 = 4x16, 3 = 6x32, 4 = 6x16, 5 = 8x16 pixels), else the
 * k_gather_gemm tile id (0 = 256x64, 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 160x128 rows x channels); < 0: bad geom. */
int32_t vn_conv_plan_id(const vnConv *geom);

/* Weight-gradient: dw[tap][n][k] += sum_m src[site(m,tap), k] * rows[m, n]
 * fp32, packed [taps][Cr][C] orientation, C = real source channels — the caller zeroes dw first.
 * The sum over m is split into row chunks whose partial tiles go to `workspace`
 * (vn_conv_wgrad_workspace_bytes; n_rows = 0 for the dense form) and are then added in a fixed order: no
 * atomics, bit-reproducible.  A NULL / smaller workspace only means fewer chunks (less parallelism).
 * rows_* strides address the (B,Dr,Hr,Wr,Cr) gradient.  split must be 0 (retired, see above).
 * geom->dtype VN_F32X3S: BOTH operands are stored split; VN_F32X3: both are fp32 and split in registers. */
size_t vn_conv_wgrad_workspace_bytes(const vnConv *geom, int32_t split, int64_t n_rows);
int vn_conv_wgrad(const void *src /*bf16*/, const void *rows /*bf16*/, float *dw_packed,
                  const vnConv *geom, int32_t split, void *workspace, size_t workspace_bytes,
                  vnStream stream);
/* Introspection (no launch): 200 = k_wgrad_patch, else 1000 * three-tap + 10 * TN + TK of k_wgrad<TN,TK> (tile =
 * 32 TN x 32 TK channels) for this geometry. */
int32_t vn_conv_wgrad_plan_id(const vnConv *geom, int32_t split, int64_t n_rows);
/* Same products, but the row-chunk partials are left in the workspace: chunk c at workspace + c * (taps*Cr*C) floats,
 * *chunks (host) = their number.  vn_unpack_wgrads_batch sums them (vnUnpackJob.chunks) while it converts the layout,
 * so a backward pass needs no per-layer reduction launch and no zeroed accumulator.  row_list != NULL: the
 * row-list form (vn_conv_wgrad_rows' operands).  The workspace must hold vn_conv_wgrad_workspace_bytes. */
int vn_conv_wgrad_partials(const void *src, const void *rows, const vnConv *geom, int32_t split,
                           const int64_t *row_list, int64_t n_rows, void *workspace,
                           size_t workspace_bytes, int32_t *chunks, vnStream stream);
/* One of the three bf16 products of an fp32x3 weight gradient over operands in split storage (geom->dtype VN_F32X3S):
 * pass 0 = hi(src).hi(rows), 1 = lo(src).hi(rows), 2 = hi(src).lo(rows) on the bf16 nine-tap patch kernel, reading the
 * halves in place.  Only the geometries of that kernel (the 64-channel Conv3d layers, model.py:207-209), else
 * VN_EUNSUPPORTED.  Lay the three passes' partial slabs one after the other; the unpack sums them like row chunks. */
int vn_conv_wgrad_partials_split_pass(const void *src, const void *rows, const vnConv *geom, int32_t pass, void *workspace,
                                      size_t workspace_bytes, int32_t *chunks, vnStream stream);

/* A data-gradient launch (ConvMD backward, model.py:111-167) that also leaves the BatchNorm-backward sums of the layer
 * BELOW in its epilogue: out = the gradient w.r.t. that layer's activation a = relu(BN(y)); bn_y = that layer's conv
 * output (same site strides as out), bn_stats its forward statistics [4][Cr] (vn_bn_finalize_slab);
 * slab[vn_conv_stats_slab_rows(geom)][2][Cr] receives per workgroup sum dz and sum dz*xhat, the rows
 * vn_bn_bwd_finalize_slab reads — the vn_bn_bwd_reduce_slab launch is saved.  Only the geometries with
 * vn_conv_plan_id == 123 (small-image 3x3 kernel); VN_EUNSUPPORTED otherwise. */
int vn_conv_dgrad_bn_bwd(const void *src, const void *w_packed, void *out, vnDtype out_dtype, const vnConv *geom,
                         const void *bn_y, vnDtype bn_y_dtype, const float *bn_stats, float *slab, vnStream stream);
/* Row-list ("sparse rows") variants for the first middle layer, whose input grid is ~99 % empty:
 * the produced rows are an explicit list of (b,d,h,w) int64 coordinates instead of the dense
 * (B,Dr,Hr,Wr) grid.  row_count (device int32, may be NULL = row_cap) lets the launch be sized by a
 * capacity known on the host while the true count stays on the device.  out_linear != 0: output row i is
 * written at out + i*out_sW (e.g. the (K,128) voxel gradient); else at the site given by the coordinates.
 * Any div (strided transposed gather) is allowed.  In vn_conv_wgrad_rows the rows operand is the
 * [n_rows][Cr] matrix of the list rows (stride out_sW) and dw is [taps][Cr][Cs]. */
int vn_conv_gather_gemm_rows(const void *src, const void *w_packed, const float *bias, void *out,
                             vnDtype out_dtype, const vnConv *geom, const int64_t *row_list,
                             int64_t row_cap, const int32_t *row_count, int32_t out_linear,
                             float *stats_slab, vnStream stream);
int vn_conv_wgrad_rows(const void *src, const void *rows, float *dw_packed, const vnConv *geom,
                       const int64_t *row_list, int64_t n_rows, void *workspace, size_t workspace_bytes,
                       vnStream stream);
/* vn_conv_wgrad_partials over a row list whose length is only known on the device: row_cap = the list's capacity (it
 * sizes the launch and the chunking), *row_count (device) the valid rows; chunks past the count store zero partials. */
int vn_conv_wgrad_partials_counted(const void *src, const void *rows, const vnConv *geom, const int64_t *row_list,
                                   int64_t row_cap, const int32_t *row_count, void *workspace, size_t workspace_bytes,
                                   int32_t *chunks, vnStream stream);
/* Active output sites of a forward conv over a sparse input: the ordered (b,d,h,w) list of the sites
 * whose receptive field contains at least one of the K occupied voxel coordinates (coord (K,4) int64
 * [b,z,y,x]).  geom = the conv's forward geometry.  list holds up to cap rows; *count = min(n, cap).
 * Deterministic order (ascending linear site index). */
size_t vn_active_sites_workspace_bytes(const vnConv *geom);
int vn_active_sites(const int64_t *coord, int64_t K, const vnConv *geom, void *workspace,
                    size_t workspace_bytes, int64_t *list, int64_t cap, int32_t *count, vnStream stream);
/* "Rulebook" evaluation of a conv over a sparse input (the first middle layer, model.py:207 over model.py:102-106),
 * in three calls: (1) vn_voxel_index_grid: int32 grid over the INPUT cells, voxel row index or -1;
 * (2) P[v][t][:] = W[t] . x[v] for every voxel and tap = ONE dense vn_conv_gather_gemm of the (K,Cin) voxel rows
 * against the packed [taps*Cout][Cin] weights (fp32 output, row stride taps*Cout); (3) vn_rulebook_combine: every
 * listed (active) output site adds the P rows of the taps whose source cell is occupied, in tap order, + bias ->
 * y at the site (dense addressing) and, optionally, per-workgroup sum / sum of squares (stats_slab
 * [vn_rulebook_slab_rows(cap)][2][Cout], the layout vn_bn_finalize_slab reads).  Work ~ K*taps instead of sites*taps.  Cout must be 64. */
int vn_voxel_index_grid(const int64_t *coord, int64_t K, int32_t B, int32_t D, int32_t H, int32_t W,
                        int32_t *grid /* (B,D,H,W) */, vnStream stream);
int64_t vn_rulebook_slab_rows(int64_t cap);   /* rows of vn_rulebook_combine's stats_slab for a list capacity */
int vn_rulebook_combine(const float *P, const int32_t *index_grid, const int64_t *list, int64_t cap,
                        const int32_t *count, const vnConv *geom, const float *bias, void *y,
                        vnDtype y_dtype, float *stats_slab, vnStream stream);
/* y[m][0:C] = values[0:C] for M rows (the conv output at inactive sites is the bias) */
int vn_fill_rows(void *y, vnDtype dtype, int64_t M, int32_t C, int64_t stride, const float *values,
                 vnStream stream);

/* Packing between torch parameter layouts and the kernels' [taps][N][K] bf16.
 * mode 0: Conv weight (Cout,Cin,k...) -> forward operand   [tap][Cout][Cin]
 * mode 1: Conv weight                 -> data-grad operand [tap][Cin][Cout]
 * mode 2: ConvTranspose weight (Cin,Cout,kh,kw) -> forward operand [tap][Cout][Cin]
 * mode 3: ConvTranspose weight        -> data-grad operand [tap][Cin][Cout]
 * split3: must be 0 (the [hi;hi;lo] K expansion of the retired bf16x3 mode).
 * packed_dtype VN_F32X3 (operand of a convolution launched with vnConv.dtype VN_F32X3): rows of K fp32-sized slots
 * in the "split fp32" format of VN_F32X3S: every aligned group of 8 input channels (32 B) = the eight hi bf16 parts, then
 * the eight lo parts (hi = bf16(w), lo = bf16(w - hi)) — a lane of the kernels reads the group 8 fq .. 8 fq + 7 of each
 * 128-B chunk as its eight k values.  Needs K % 32 == 0; for other K the packed operand is plain fp32 and the kernels split
 * it in registers (the same values either way).
 * cin_fold f: the packed Cin index p stands for torch channel (p % (Cin/f))*f + p/(Cin/f)
 * (f = 2 implements the BEV reshape of model.py:262, channel = c*2 + d; else 1). */
int vn_pack_weight(const float *w, int32_t c_out, int32_t c_in, int32_t taps, int32_t mode,
                   int32_t split3, int32_t cin_fold, void *packed, vnDtype packed_dtype,
                   vnStream stream);
/* inverse for gradients: packed fp32 [tap][N][K] (mode 0 or 2 orientation) -> torch layout */
int vn_unpack_wgrad(const float *dw_packed, int32_t c_out, int32_t c_in, int32_t taps,
                    int32_t mode, int32_t cin_fold, float *dw, vnStream stream);
/* Batched forms of the two calls above: all layers of a step in one launch (the single-layer launches are a few
 * microseconds of dispatch each, 72 per train step).  Fields as the arguments of vn_pack_weight / vn_unpack_wgrad. */
typedef struct vnPackJob {
    const float *w;
    void *packed;
    int32_t c_out, c_in, taps, mode, split3, cin_fold;
    int32_t packed_dtype;   /* vnDtype */
    int32_t pad_;
} vnPackJob;
typedef struct vnUnpackJob {
    const float *dw_packed;
    float *dw;
    int32_t c_out, c_in, taps, mode, cin_fold;
    int32_t chunks;          /* > 1: dw = sum over c < chunks (in order) of dw_packed[c * chunk_stride + i] */
    int64_t chunk_stride;    /* elements; the chunk partials of vn_conv_wgrad_partials */
} vnUnpackJob;
int vn_pack_weights_batch(const vnPackJob *jobs /* host array */, int32_t n, vnStream stream);
int vn_unpack_wgrads_batch(const vnUnpackJob *jobs /* host array */, int32_t n, vnStream stream);


/* ------------------------------------------------------------------------
 * Camera field-of-view crop of a raw Velodyne sweep — voxelnet/preprocess_data.py:42-103 (prepare_velo_points,
 * project_velo_to_img and the in-image test of align_img_and_velo; main() rewrites the .bin files with the survivors,
 * :151-154), SURVEY.md 8(f)-3.  points (n,4) fp32 [x,y,z,reflectance] on the device; P_3x4 / Tr_velo_to_cam_4x4 /
 * R_rect_4x4: HOST pointers to the float32 matrices load_calib returns (row-major).  A point survives when reflectance
 * > 0, its rectified camera z >= 0 and its rounded pixel satisfies 0 < col < image_cols, 0 < row < image_rows (float32
 * arithmetic, np.round = round half to even).  out_points (capacity n rows) receives the survivors in input order,
 * out_index (n int32, may be NULL) their input row numbers, *out_count (device int32) their number; rows
 * [*out_count, n) of out_points are filled with NaN points, which every range test drops (vn_voxelize_index on all n
 * rows gives the result of the first *out_count rows bit for bit), so the input pipeline never reads the count back.
 * Asynchronous.
 * ---------------------------------------------------------------------- */
size_t vn_fov_crop_workspace_bytes(int64_t n);
int vn_fov_crop(const float *points, int64_t n, const float *P_3x4, const float *Tr_velo_to_cam_4x4,
                const float *R_rect_4x4, int32_t image_rows, int32_t image_cols, float *out_points,
                int32_t *out_index, int32_t *out_count, void *workspace, size_t workspace_bytes, vnStream stream);

/* ------------------------------------------------------------------------
 * Native step executor — MiddleConvNet.forward (model.py:257-281) and its backward as ONE call
 * each (csrc/runtime.hip): layer table, launch geometry and workspace arena live in C++, so the
 * ~450 launches of a step cost microseconds of host time instead of a Python round trip each.
 * The workspace (vn_net_workspace_bytes, caller allocated) also keeps the activations between
 * the two calls.  mode: 0 = bf16 operands, 1 = fp32 operands (parity mode).  Layers are indexed
 * in execution order: middle_layer.0-2, block1.0-4, deconv1, block2.0-5, deconv2, block3.0-5,
 * deconv3 (23 entries); the two 1x1 heads are passed concatenated (prob rows first).
 * sparse_first: the first Conv3d runs in its row-list form (needs coord, K; backward returns
 * the (K,128) fp32 voxel gradient in d_input and needs vw_rows (K,128) in the operand dtype);
 * else d_input receives the dense (B,D,H,W,128) gradient in the operand dtype (may be NULL).
 * The backward can be issued in segments of its 24 steps (0 = heads, 1..23 = layers in backward
 * order: deconv3, block3.5..0, deconv2, block2.5..0, deconv1, block1.4..0, middle_layer.2..0) so
 * a caller can start the gradient all-reduce of finished parameter groups in between.
 * ---------------------------------------------------------------------- */
typedef struct {
    int32_t B, D, H, W;      /* dense voxel grid; D = 9 .. 12 (every depth whose three Conv3d layers fold to 2: model.py:207-209, 262) */
    int32_t block1_stride;   /* 2: Car, 1: Pedestrian/Cyclist (model.py:212-227) */
    int32_t mode;            /* 0 bf16, 1 fp32, 2 fp32x3 (fp32 storage, conv / weight-gradient products as three bf16 MFMAs: VN_F32X3) */
    int32_t training;        /* BatchNorm: batch statistics + running-stat update, or running statistics */
    int32_t sparse_first;
    int32_t prepared;        /* forward: vn_net_prepare has already been issued for this step on the same vnNet (weights packed,
                              * first layer's site list / index grid / bias fill); vn_net_forward orders itself behind it
                              * (two events: the first layer's needs, then the rest of the packing) */
    int32_t bucket_events;   /* single-call backward with a side stream: unpack each parameter group's weight gradients on the
                              * side stream at the group's end and record "group final" events (vn_net_wait_bucket);
                              * implies that the caller joins the side stream (as defer_join) */
    int32_t defer_join;      /* backward with a side stream: the last segment does NOT wait for the side stream; the
                              * weight-gradient unpack runs there and the CALLER joins it (stream wait) before anything
                              * reads the weight gradients — lets e.g. the VFE backward run beside the last weight gradients */
    int32_t grad_storage;    /* bf16 mode only: which tensors of the step are kept in fp32 instead of bf16 (bit set), DESIGN.md §4:
                              * 1 = the heads' data gradient (the 768-channel concat gradient the three deconvs' BatchNorm
                              *     backward reads), 2 = every layer's data gradient (the input of the BatchNorm backward below),
                              * 4 = every conv output y (forward), 8 = the heads' data gradient is computed from the fp32
                              * (B*S,16) logit gradient and the fp32 head weights (vn_heads_dgrad_f32) instead of their bf16
                              * roundings.  0 = everything bf16 (the shipped configuration: tools/grad_attribution.py measures
                              * that none of the four moves a gradient).  fp32 mode: 16 = diagnostic, every activation is rounded
                              * to the nearest bf16 VALUE (stored fp32; the sparse first-layer backward shortcuts are off) — the
                              * bf16 forward function with an exact fp32 backward. */
} vnNetConfig;
typedef struct {
    const float *weight, *bias, *gamma, *beta;
    float *running_mean, *running_var;
} vnLayerParams;
typedef struct {
    float *weight, *bias, *gamma, *beta;   /* overwritten */
} vnLayerGrads;
size_t vn_net_workspace_bytes(const vnNetConfig *cfg, int64_t K);
/* Executor context: the HIP events of the two-stream schedule (fork / join ring, "parameter group final" events), created
 * on the current device.  Caller-owned (one per executor instance, e.g. one RPN3D module on one device) — the library
 * keeps no mutable state of its own.  Calls sharing a context must not overlap on the host. */
typedef struct vnNet vnNet;
int vn_net_create(vnNet **out);
int vn_net_destroy(vnNet *net);
/* Optional per-launch timing of the executor (bench.py's live roofline measurement, taken on the SAME native path it
 * times): after vn_net_timing_begin every launch of vn_net_prepare / vn_net_forward / vn_net_backward on this context
 * is bracketed by two HIP timing events on the stream the kernel is launched on; vn_net_timing_read stops the timing,
 * waits for the events and returns one record per launch in issue order (*count = launches recorded; at most `cap`
 * are written).  flops / bytes are the ALGORITHMIC work of the launch (2 x MACs of the layer definition
 * for forward, data gradient and weight gradient alike; tensor bytes read + written once), 0 where not defined. */
enum { VN_T_CONV_FWD = 0, VN_T_CONV_DGRAD = 1, VN_T_WGRAD = 2, VN_T_BN_APPLY = 3, VN_T_BN_BWD_REDUCE = 4,
       VN_T_BN_BWD_APPLY = 5, VN_T_BN_FINALIZE = 6, VN_T_UNPACK = 7, VN_T_PACK = 8, VN_T_FIRST = 9, VN_T_MISC = 10 };
typedef struct {
    int32_t kind;    /* VN_T_* */
    int32_t layer;   /* execution-order index 0..22, 23 = heads, -1 = all layers */
    float ms;        /* event-to-event duration of the launch */
    float start_ms;  /* start of the launch's event bracket relative to the first record of this vn_net_timing_read */
    double flops, bytes;
} vnTimingRecord;
int vn_net_timing_begin(vnNet *net, int32_t max_records);
int vn_net_timing_read(vnNet *net, vnTimingRecord *out, int32_t cap, int32_t *count);
/* Make `stream` wait until parameter group `bucket` (0: heads+deconv3+block3, 1: deconv2+block2+deconv1, 2: block1,
 * 3: middle_layer) of the most recent vn_net_backward issued with cfg->bucket_events has its final gradients. */
int vn_net_wait_bucket(vnNet *net, int32_t bucket, vnStream stream);
/* The part of the forward that does not depend on the voxel features (weight packing; sparse first layer: active
 * sites, voxel index grid, bias fill): may be issued on another stream while the VFE forward runs; then set
 * cfg->prepared for vn_net_forward.  The forward's stream need not wait for this one: vn_net_prepare records, on the
 * vnNet, one event behind what the first layer needs (issued first) and one behind the rest of the weight packing, and
 * vn_net_forward waits for the first at its start and for the second in front of the second layer.
 * cfg->prepared names the PHASE of a vn_net_prepare call: 0 = everything in one call; 1 = only the first layer's needs
 * (event 0; heads_w may be NULL) and return; 2 = only the rest (event 1), valid only right after a phase-1 call for the
 * same workspace, coord, K and grid (else VN_EINVAL) — for a caller whose heads_w is itself produced on that stream (a
 * concatenation of the two heads' parameters) and should not sit in front of the site list.  An error return clears the
 * protocol state; vn_net_forward with cfg->prepared != 0 returns VN_EINVAL unless the prepare was issued for ITS
 * workspace / coord / K / grid, and consumes it. */
int vn_net_prepare(vnNet *net, const vnNetConfig *cfg, const vnLayerParams *layers, const float *heads_w,
                   const int64_t *coord, int64_t K, void *workspace, size_t workspace_bytes, vnStream stream);
int vn_net_forward(vnNet *net, const vnNetConfig *cfg, const vnLayerParams *layers /*[23]*/,
                   const float *heads_w /*[16,768]*/, const float *heads_b /*[16]*/,
                   const void *dense, const int64_t *coord, const void *vw_rows /* (K,128) voxel rows in the operand dtype, sparse_first only */, int64_t K, void *workspace,
                   size_t workspace_bytes, float *prob /*(B,2,h,w)*/, float *reg /*(B,14,h,w)*/,
                   vnStream stream,
                   vnStream side_stream /* NULL, or a second stream deconv1 / deconv2 run on beside block2 / block3 */);
/* VN_EUNSUPPORTED when cfg->training == 0: only the train-mode BatchNorm backward exists. */
int vn_net_backward(vnNet *net, const vnNetConfig *cfg, const vnLayerParams *layers, const float *heads_w,
                    const float *d_prob, const float *d_reg, const float *prob, const void *dense,
                    const int64_t *coord, const void *vw_rows, int64_t K, void *workspace,
                    size_t workspace_bytes, const vnLayerGrads *grads /*[23]*/, float *d_heads_w,
                    float *d_heads_b, void *d_input, int32_t seg_begin, int32_t seg_end,
                    vnStream stream, vnStream side_stream /* NULL, or a second stream the weight-gradient
                    launches run on beside the data-gradient ones; joined back before the call returns */);

/* ONE call per train step (voxelnet/model.py:298-362 + voxelnet/train.py:151-154): the voxel feature encoder, the
 * middle layers + RPN, the loss, the whole backward and (n_chunks > 0) clip_grad_norm_ + SGD, issued in the order and on
 * the streams the separate calls are issued by a host that overlaps them by hand:
 *   side:  [wait stream]  counters += 1 | vn_net_prepare phase 1 | heads' parameters -> heads_w / heads_b | vn_net_prepare phase 2 |
 *          [wait targets_stream] vn_rpn_loss_norm | ... | [wait the pass] vn_rpn_loss_finalize -> loss5
 *   main:  vn_vfe_fwd (bf16 mode: vn_vfe_fwd_rows, whose last pass writes vw_rows too: no vn_cast_rows launch) |
 *          vn_net_forward | [wait the normalisers] vn_rpn_loss_fwd_bwd_rows (g_loss): ONE launch between the heads and their
 *          backward, which also leaves the heads' gradient rows in the arena (no vn_heads_bwd launch) |
 *          vn_net_backward(0..24, defer_join) | vn_vfe_bwd | [wait side] | vn_clip_sgd
 * Same arithmetic and same results as those calls (tests/test_gpu_step.py: bit-identical); what it saves is the host's
 * work between them and four of the five loss launches on the chain between the network's forward and backward.  Every buffer is the caller's; scratch ones (voxelwise, vfe_stats, vw_rows, d_voxelwise,
 * heads_w / heads_b, d_prob / d_reg, the three workspaces) need only live until the step has run.
 * Requires cfg->sparse_first, cfg->training and a side stream (VN_EUNSUPPORTED / VN_EINVAL otherwise).
 * cfg->bucket_events is honoured (vn_net_wait_bucket after the call; pass n_chunks = 0 and update after the exchange). */
struct vnParamChunk;
typedef struct {
    const float *feature;          /* (K,T,7) */
    const int64_t *coord;          /* (K,4) */
    int64_t K;
    int32_t T;
    float bn_momentum, bn_eps;     /* the VFE's BatchNorm1d (the executor's layers use 0.1 / 1e-5 as vn_net_forward does) */
    vnVfeWeights vfe;
    vnVfeGrads vfe_grads;
    void *vfe_ws; size_t vfe_ws_bytes;            /* vn_vfe_workspace_bytes(K, T) */
    float *voxelwise;              /* (K,128) */
    float *vfe_stats;              /* 320 */
    void *vw_rows;                 /* (K,128) in the operand dtype: bf16 mode = a bf16 scratch buffer; fp32 / fp32x3 = voxelwise itself */
    float *d_voxelwise;            /* (K,128) */
    const float *prob_w, *prob_b, *reg_w, *reg_b;   /* prob_conv / reg_conv parameters: (2,768) (2) (14,768) (14) */
    float *heads_w, *heads_b;      /* scratch (16,768) (16): their concatenation, prob rows first */
    float *d_heads_w, *d_heads_b;  /* gradients of the concatenation (16,768) (16) */
    const vnLayerParams *layers;   /* [23] */
    const vnLayerGrads *grads;     /* [23] */
    void *ws; size_t ws_bytes;     /* vn_net_workspace_bytes */
    float *prob, *reg;             /* outputs (B,2,h,w) (B,14,h,w) */
    float *d_prob, *d_reg;         /* scratch, same shapes */
    const float *pos, *neg, *targets;   /* (B,h,w,2) (B,h,w,2) (B,h,w,14) */
    vnStream targets_stream;       /* NULL, or the stream those three are produced on: the loss waits for what is queued there now */
    float alpha, beta, sigma;
    void *loss_ws; size_t loss_ws_bytes;          /* vn_rpn_loss_workspace_bytes */
    float *loss5;                  /* output: loss, cls_loss, reg_loss, cls_pos_loss_rec, cls_neg_loss_rec */
    const float *g_loss;           /* device scalar: d(objective) / d(loss), normally 1 */
    const struct vnParamChunk *chunks; int32_t n_chunks;   /* vn_clip_sgd's table; n_chunks = 0: no parameter update */
    float max_norm, lr;
    int32_t scale_grads;
    void *opt_ws; size_t opt_ws_bytes;
    float *total_norm;             /* may be NULL */
    int64_t *const *bn_counters;   /* NULL, or a DEVICE array of n_bn_counters pointers: every BatchNorm's num_batches_tracked, */
    int32_t n_bn_counters;         /* += 1 each (nn.BatchNorm*.forward in train mode) — one launch on the side stream */
    vnStream stream, side_stream;
} vnStep;
int vn_net_step(vnNet *net, const vnNetConfig *cfg, const vnStep *step);

/* ------------------------------------------------------------------------
 * Data-parallel gradient exchange over RCCL / xGMI (the reference has no distributed code):
 * thin wrappers, RCCL bound at run time (dlopen; inside a PyTorch process torch's own librccl is reused).
 * Status: VN_EUNSUPPORTED when no librccl can be found, 1000 + ncclResult_t for RCCL errors.
 * vn_comm_create is collective (every rank, on its own current device, with rank 0's 128-byte id);
 * `nccl_comm` is a plain ncclComm_t — a communicator made elsewhere (e.g. by an MPI launcher) works as well.
 * vn_allreduce_bucket: bucket[i] = sum over ranks of scale * bucket[i], in place, asynchronous on `stream`
 * (scale = 1/world: the mean stock DDP takes; scaled before the sum, so all ranks end bit-identical).
 * ---------------------------------------------------------------------- */
int vn_comm_rccl_version(void); /* ncclGetVersion() code of the librccl bound at run time (e.g. 22703), 0 if none;
                                  * every vn_comm_* / vn_allreduce_bucket call returns VN_EUNSUPPORTED unless it is 2.x:
                                  * the RCCL entry points are declared by hand (ncclUniqueId by value, ncclFloat32 = 7) */
int vn_comm_unique_id(void *id128 /* out: 128 bytes (ncclUniqueId) */);
int vn_comm_create(void **nccl_comm, const void *id128, int32_t world, int32_t rank);
int vn_comm_destroy(void *nccl_comm);
int vn_allreduce_bucket(void *nccl_comm, float *bucket, int64_t count, float scale, vnStream stream);

/* ------------------------------------------------------------------------
 * BatchNorm(+ReLU) over channels-last rows — nn.BatchNorm{1,2,3}d defaults
 * (momentum 0.1, eps 1e-5) as used at model.py:72,142,153,193.
 * Rows are M x C with `stride` elements between rows.  `fold` (1 or 2): the C
 * columns are `fold` copies of C/fold real channels (BEV view of the last
 * Conv3d, model.py:262); parameter vectors have C/fold entries, the sums /
 * stats / coef work vectors are C wide.
 * ---------------------------------------------------------------------- */
/* sums[0:C] += sum_m (y - shift), sums[C:2C] += sum_m (y - shift)^2   (double; caller zeroes)
 * shift: float[C/fold] or NULL (conditioning only; finalize takes the same shift) */
int vn_bn_stats(const void *y, vnDtype dtype, int64_t M, int32_t C, int64_t stride, int32_t fold,
                const float *shift, double *sums, vnStream stream);
/* sums -> stats[4C] = mean | invstd | S = gamma*invstd | beta  with a = S*(y-mean) + beta;
 * training: batch statistics (biased var) + running-stat update (unbiased var);
 * eval (training == 0): running statistics, sums ignored. */
int vn_bn_finalize(const double *sums, int64_t M, int32_t C, int32_t fold, const float *shift,
                   const float *gamma, const float *beta, float *running_mean, float *running_var,
                   int32_t training, float momentum, float eps, float *stats, vnStream stream);
/* same, from a vn_conv_gather_gemm stats slab: float[rows][2][C] partial sums of (y - shift) */
int vn_bn_finalize_slab(const float *slab, int64_t slab_rows, int64_t M, int32_t C, const float *shift,
                        const float *gamma, const float *beta, float *running_mean, float *running_var,
                        float momentum, float eps, float *stats, vnStream stream);
/* a = relu?(S*(y-mean) + beta) -> f32 or bf16 rows.  lo_off != 0 (split mode): the
 * bf16 residual lo = bf16(a - hi) is also written, lo_off elements after hi. */
int vn_bn_apply(const void *y, vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C,
                const float *stats, int32_t relu, void *a, vnDtype a_dtype, int64_t a_stride,
                int64_t lo_off, vnStream stream);
/* backward step 1: dz = da * (relu ? S*y+T > 0 : 1);
 * sums[0:C] += sum dz, sums[C:2C] += sum dz*xhat   (double; caller zeroes) */
int vn_bn_bwd_reduce(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y,
                     vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                     int32_t relu, double *sums, vnStream stream);
/* step 2: folds the sums; coef[3C] with dy = c0*dz + c1*(y-mean) + c2;
 * d_gamma[C/fold] = sum dz*xhat, d_beta[C/fold] = sum dz */
int vn_bn_bwd_finalize(const double *sums, int64_t M, int32_t C, int32_t fold, const float *gamma,
                       const float *stats, float *coef, float *d_gamma, float *d_beta,
                       vnStream stream);
/* steps 1+2 without atomics: every workgroup writes one slab row float[2][C] (vn_bn_bwd_slab_rows(M,C)
 * rows), reduced in double by the finalize — deterministic and ~2x faster than thousands of
 * workgroups adding into the same 2C addresses */
int64_t vn_bn_bwd_slab_rows(int64_t M, int32_t C);
int vn_bn_bwd_reduce_slab(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y,
                          vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                          int32_t relu, float *slab, vnStream stream);
int vn_bn_bwd_finalize_slab(const float *slab, int64_t slab_rows, int64_t M, int32_t C,
                            const float *gamma, const float *stats, float *coef, float *d_gamma,
                            float *d_beta, vnStream stream);
/* step 3: dy = c0*dz + c1*(y-mean) + c2 -> f32 or bf16 (+ residual at lo_off, as vn_bn_apply) */
int vn_bn_bwd_apply(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y,
                    vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                    const float *coef, int32_t relu, void *dy, vnDtype dy_dtype, int64_t dy_stride,
                    int64_t lo_off, vnStream stream);
/* BEV-fold variants for the last Conv3d (model.py:262): y / dy are its plain (B*2*H*W, C=64) rows, a / da the
 * (B,1,H,W,2C) tensor block1 sees (row stride wide_stride, channel = d*C + c).  hw = H*W.  One launch each. */
int vn_bn_apply_bev(const void *y, vnDtype y_dtype, int64_t M, int32_t C, int64_t hw, const float *stats,
                    int32_t relu, void *a, vnDtype a_dtype, int64_t wide_stride, vnStream stream);
int vn_bn_bwd_reduce_slab_bev(const void *da, vnDtype da_dtype, int64_t wide_stride, const void *y,
                              vnDtype y_dtype, int64_t M, int32_t C, int64_t hw, const float *stats,
                              int32_t relu, float *slab, vnStream stream);
int vn_bn_bwd_apply_bev(const void *da, vnDtype da_dtype, int64_t wide_stride, const void *y,
                        vnDtype y_dtype, int64_t M, int32_t C, int64_t hw, const float *stats,
                        const float *coef, int32_t relu, void *dy, vnDtype dy_dtype, vnStream stream);
/* Sparse-aware BatchNorm passes of the FIRST middle layer (model.py:207 on the ~99 % empty grid): every output site
 * without an occupied voxel in its receptive field holds exactly the conv bias (`inactive`, float[C]); row_flags are the
 * uint8 site flags vn_active_sites leaves at the head of its workspace.  vn_bn_apply_flagged / _reduce_slab_flagged do
 * not read y at rows with flag 0 (same results as the dense calls); vn_bn_bwd_apply_list writes dy only at the sites of
 * vn_active_sites' list ((b,d,h,w) int64 rows, *count valid, dense contiguous (B,D,H,W,C) tensors). */
int vn_bn_apply_flagged(const void *y, vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                        int32_t relu, void *a, vnDtype a_dtype, int64_t a_stride, const uint8_t *row_flags,
                        const float *inactive, vnStream stream);
int vn_bn_bwd_reduce_slab_flagged(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y,
                                  vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                                  int32_t relu, float *slab, const uint8_t *row_flags, const float *inactive,
                                  vnStream stream);
int vn_bn_bwd_apply_list(const void *da, vnDtype da_dtype, const void *y, vnDtype y_dtype, int32_t C, int32_t D,
                         int32_t H, int32_t W, const float *stats, const float *coef, int32_t relu, void *dy,
                         vnDtype dy_dtype, const int64_t *list, const int32_t *count, int64_t cap, vnStream stream);

/* BatchNorm backward of a layer whose output y is the constant inactive[c] outside a site list (the first middle layer,
 * model.py:207 + 142 over the sparse grid of model.py:102-106) WITHOUT the dense gradient of its activation: da_rows is
 * the next layer's data gradient at the listed sites only ([cap][C], list order: vn_conv_gather_gemm_rows with
 * out_linear) and total[c] its sum over ALL sites (vn_dgrad_total).  reduce_list: slab [vn_bn_bwd_list_slab_rows][3][C];
 * finalize_list: coef / d_gamma / d_beta as vn_bn_bwd_finalize_slab; apply_list_rows: dy at the listed sites (dense
 * addressing) as vn_bn_bwd_apply_list.  Same sums as the dense passes (tests/test_gpu_layers.py). */
int64_t vn_bn_bwd_list_slab_rows(int64_t cap, int32_t C);
int vn_bn_bwd_reduce_list(const void *da_rows, vnDtype da_dtype, const void *y, vnDtype y_dtype, int32_t C, int32_t D,
                          int32_t H, int32_t W, const float *stats, int32_t relu, float *slab, const int64_t *list,
                          const int32_t *count, int64_t cap, vnStream stream);
int vn_bn_bwd_finalize_list(const float *slab, int64_t slab_rows, int64_t M, int32_t C, const float *gamma,
                            const float *stats, const float *total, const float *inactive, vnDtype y_dtype,
                            int32_t relu, float *coef, float *d_gamma, float *d_beta, vnStream stream);
int vn_bn_bwd_apply_list_rows(const void *da_rows, vnDtype da_dtype, const void *y, vnDtype y_dtype, int32_t C, int32_t D,
                              int32_t H, int32_t W, const float *stats, const float *coef, int32_t relu, void *dy,
                              vnDtype dy_dtype, const int64_t *list, const int32_t *count, int64_t cap, vnStream stream);
/* total[ci] = the sum over ALL input sites of the data gradient of a 3x3x(kD) convolution (stride 1 and padding 1 in
 * H/W, stride 1 and no padding in D; weight w = the torch (Co,Ci,kD,3,3) fp32 tensor, rounded to bf16 when dy is bf16
 * as the data-gradient kernels read it) — from nine box sums of its dense output gradient dy (B,D,H,W,Co) instead of
 * the dense data gradient (ConvMD backward, model.py:111-167).  dy_sums_to_zero != 0: dy comes out of a train-mode
 * BatchNorm backward (model.py:142), whose per-channel sum over all sites is zero in exact arithmetic: it is taken as
 * zero and only the edge rows / columns of dy are read. */
size_t vn_dgrad_total_workspace_bytes(int32_t Co);
int vn_dgrad_total(const void *dy, vnDtype dy_dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co, int32_t Ci,
                   int32_t kD, const float *w, int32_t dy_sums_to_zero, void *workspace, size_t workspace_bytes,
                   float *total, vnStream stream);
/* The sparse route for the WEIGHT gradient of the layer above the first middle layer (ConvMD backward, model.py:111-167,
 * for middle_layer.1): its input a = relu(BN(y)) is the constant cvec outside the first layer's site list, so
 *   dW = [weight gradient against the rows a(site) - cvec of the listed sites: vn_act_delta_rows +
 *         vn_conv_wgrad_partials_counted]  +  cvec (x) box sums of dy  [vn_wgrad_const_add, after the unpack].
 * vn_box_col_sums: the box-sum half of vn_dgrad_total alone (same workspace layout). */
int vn_box_col_sums(const void *dy, vnDtype dy_dtype, int32_t B, int32_t D, int32_t H, int32_t W, int32_t Co,
                    void *workspace, size_t workspace_bytes, vnStream stream);
int vn_act_delta_rows(const void *a, vnDtype a_dtype, int32_t C, int32_t D, int32_t H, int32_t W, const float *stats,
                      const float *inactive, vnDtype y_dtype, int32_t relu, const int64_t *list, const int32_t *count,
                      int64_t cap, void *delta_rows, vnDtype delta_dtype, vnStream stream);
int vn_wgrad_const_add(float *dw, const void *workspace, size_t workspace_bytes, int32_t Co, int32_t Ci, int32_t kD,
                       const float *stats, const float *inactive, vnDtype y_dtype, vnDtype a_dtype, int32_t relu,
                       vnStream stream);
/* Row-flag variant for the first middle layer: row_flags = the uint8 site flags vn_active_sites leaves at the head
 * of its workspace ((B,Dr,Hr,Wr) order, 1 = some occupied voxel in the receptive field).  Rows with flag 0 are
 * skipped: that layer's weight- and data-gradient (the row-list kernels) only gather dy at flagged sites. */
int vn_bn_bwd_apply_flagged(const void *da, vnDtype da_dtype, int64_t da_stride, const void *y,
                            vnDtype y_dtype, int64_t y_stride, int64_t M, int32_t C, const float *stats,
                            const float *coef, int32_t relu, void *dy, vnDtype dy_dtype, int64_t dy_stride,
                            const uint8_t *row_flags, vnStream stream);


/* ------------------------------------------------------------------------
 * Layout / dtype helpers at the nn.Module boundary (the reference's modules
 * speak NC(D)HW fp32; model.py:259,262,281).
 * ---------------------------------------------------------------------- */
/* (B,C,S) fp32 -> (B,S,C) rows (f32 / bf16 [+ residual at lo_off]); S = prod(spatial) */
int vn_nchw_to_rows(const float *src, int32_t B, int32_t C, int64_t S, void *dst,
                    vnDtype dst_dtype, int64_t dst_stride, int64_t lo_off, vnStream stream);
/* (B,S,C) rows -> (B,C,S) fp32; sigmoid applied to the first `sigmoid_first_n` channels */
int vn_rows_to_nchw(const void *src, vnDtype src_dtype, int64_t src_stride, int32_t B, int32_t C,
                    int64_t S, float *dst, int32_t sigmoid_first_n, vnStream stream);
/* row-wise cast/copy of M rows of C channels between strided buffers (+ residual at lo_off) */
int vn_cast_rows(const void *src, vnDtype src_dtype, int64_t src_stride, int64_t M, int32_t C,
                 void *dst, vnDtype dst_dtype, int64_t dst_stride, int64_t lo_off, vnStream stream);
/* per-channel column sums of M rows (bias gradients): out[C] = sum_m rows[m,:] (float), two passes through per-
 * workgroup partial rows in the workspace — no atomics: bit-reproducible */
size_t vn_col_sums_workspace_bytes(int64_t M, int32_t C);
int vn_col_sums(const void *rows, vnDtype dtype, int64_t stride, int64_t M, int32_t C,
                float *out, void *workspace, size_t workspace_bytes, vnStream stream);
/* heads epilogue (model.py:281): rows (M,16) = [2 prob logits | 14 reg] ->
 * NCHW prob = sigmoid (B,2,S), reg (B,14,S); and its backward:
 * d_rows = [d_prob * p * (1-p) | d_reg] as f32, bf16 or split bf16 rows */
/* forward side in one launch: rows16 (B*S,16) fp32 (stride 16) -> prob = sigmoid(rows[:, :2]) (B,2,S), reg (B,14,S) */
int vn_heads_to_nchw(const float *rows16, int32_t B, int64_t S, float *prob, float *reg, vnStream stream);
/* The two heads (model.py:276-281) as streaming kernels over the (B*S, 768) bf16 concat rows: forward = product with the
 * packed [16][768] weight (vn_pack_weight mode 0) + bias + sigmoid on channels 0-1, written straight to prob (B,2,S) /
 * reg (B,14,S); data gradient = d_rows (M,16) bf16 times the packed [768][16] weight (mode 1) -> d_cat (M,768) bf16.
 * bf16 only (the fp32 mode keeps vn_conv_gather_gemm + vn_heads_to_nchw). */
int vn_heads_fwd(const void *cat_rows, int64_t cat_stride, const void *w_packed, const float *bias, int32_t B, int64_t S,
                 float *prob, float *reg, vnStream stream);
int vn_heads_dgrad(const void *d_rows, int64_t d_rows_stride, const void *w_packed_dgrad, void *d_cat, int64_t d_cat_stride,
                   int64_t M, vnStream stream);
/* d_cat (M, 768) rows (fp32 or bf16) = d_rows (M,16) fp32 . w (16,768) fp32 (the torch layout of the two heads' weights,
 * score rows first), all products and sums in fp32: no operand rounding in front of the three deconvs' BatchNorm backward */
int vn_heads_dgrad_f32(const float *d_rows, int64_t d_rows_stride, const float *w /*[16][768]*/, void *d_cat,
                       vnDtype d_cat_dtype, int64_t d_cat_stride, int64_t M, vnStream stream);
int vn_heads_bwd(const float *d_prob /*(B,2,S)*/, const float *d_reg /*(B,14,S)*/,
                 const float *prob /*(B,2,S)*/, int32_t B, int64_t S, void *d_rows, vnDtype d_dtype,
                 int64_t d_stride, int32_t split, vnStream stream);

/* ---- RPN loss (voxelnet/model.py:309-352, voxelnet/loss.py:3-13) -----------------------------------------
 * One pass over the (B,h,w) anchor sites for the forward sums and one for the gradients.
 *   prob (B,2,h,w), delta (B,14,h,w): the module's fp32 NCHW outputs (prob after the sigmoid, model.py:281)
 *   pos, neg (B,h,w,2), targets (B,h,w,14): utils.generate_targets' arrays as fp32 channels-last (model.py:309)
 *   out5 = [loss, cls_loss, reg_loss, cls_pos_loss_rec, cls_neg_loss_rec]  (model.py:342-351)
 * vn_rpn_loss_fwd leaves the per-sample normalisers max(1, sum pos), max(1, sum neg) (model.py:313-322) at the
 * head of the workspace; vn_rpn_loss_bwd reads them and the upstream gradients of the five outputs — five device
 * scalars, NULL = 0 (what autograd hands a five-output function; typically only g_loss is set) — and writes
 * d loss / d prob and d loss / d delta. */
size_t vn_rpn_loss_workspace_bytes(int32_t B, int32_t H, int32_t W);
int vn_rpn_loss_fwd(const float *prob, const float *delta, const float *pos, const float *neg,
                    const float *targets, int32_t B, int32_t H, int32_t W, float alpha, float beta, float sigma,
                    void *workspace, size_t workspace_bytes, float *out5, vnStream stream);
int vn_rpn_loss_bwd(const float *prob, const float *delta, const float *pos, const float *neg,
                    const float *targets, int32_t B, int32_t H, int32_t W, float alpha, float beta, float sigma,
                    const void *workspace, const float *g_loss, const float *g_cls, const float *g_reg,
                    const float *g_cls_pos, const float *g_cls_neg, float *d_prob, float *d_delta,
                    vnStream stream);

/* The same loss in three pieces for a caller that schedules them itself (vn_net_step does): the normalisers depend on
 * the target maps only (vn_rpn_loss_norm: any stream, any time before the pass); vn_rpn_loss_fwd_bwd is ONE pass over the
 * sites that leaves the forward partial sums in the workspace AND writes both gradients; vn_rpn_loss_finalize turns the
 * partial sums into out5 (nobody's input in a train step: it need not sit between the network's forward and backward).
 * Same workspace throughout; results bit-identical to vn_rpn_loss_fwd + vn_rpn_loss_bwd. */
int vn_rpn_loss_norm(const float *pos, const float *neg, int32_t B, int32_t H, int32_t W, void *workspace,
                     size_t workspace_bytes, vnStream stream);
int vn_rpn_loss_fwd_bwd(const float *prob, const float *delta, const float *pos, const float *neg,
                        const float *targets, int32_t B, int32_t H, int32_t W, float alpha, float beta, float sigma,
                        void *workspace, size_t workspace_bytes, const float *g_loss, const float *g_cls,
                        const float *g_reg, const float *g_cls_pos, const float *g_cls_neg, float *d_prob,
                        float *d_delta, vnStream stream);
int vn_rpn_loss_finalize(const void *workspace, size_t workspace_bytes, int32_t B, int32_t H, int32_t W, float alpha,
                         float beta, float *out5, vnStream stream);
/* vn_rpn_loss_fwd_bwd with the loss's own upstream gradient only, which ALSO writes the (B*H*W, 16) gradient rows the heads'
 * backward reads — d_logit = d_prob * p * (1 - p) for the two probability columns, then the 14 regression gradients: what
 * vn_heads_bwd makes of d_prob / d_delta / prob, bit for bit (a thread of the pass holds exactly one such row).  vn_net_step
 * uses it: the heads_bwd launch leaves the chain between the loss and the heads' data gradient (model.py:303-304 backward). */
int vn_rpn_loss_fwd_bwd_rows(const float *prob, const float *delta, const float *pos, const float *neg,
                             const float *targets, int32_t B, int32_t H, int32_t W, float alpha, float beta, float sigma,
                             void *workspace, size_t workspace_bytes, const float *g_loss, float *d_prob, float *d_delta,
                             void *d_rows, vnDtype d_dtype, int64_t d_stride, int32_t split, vnStream stream);

/* ---- optimizer tail (voxelnet/train.py:153-154 with the optimizer of train.py:130-132) ---------------------
 * torch.nn.utils.clip_grad_norm_(parameters, max_norm) followed by SGD(lr) without momentum / weight decay:
 *   total = sqrt(sum |g|^2) over every chunk;  coef = min(1, max_norm / (total + 1e-6));  p -= lr * coef * g
 * in two launches.  `chunks` is a DEVICE array the caller builds once for a fixed set of fp32 tensors: every
 * (parameter, gradient) pair cut into pieces of at most VN_OPT_CHUNK elements.  scale_grads != 0 also stores
 * g *= coef (what clip_grad_norm_ leaves behind); total_norm (device, may be NULL) receives the norm before
 * clipping, clip_grad_norm_'s return value. */
#define VN_OPT_CHUNK 4096
typedef struct vnParamChunk {
    float *param;
    float *grad;
    int32_t n;          /* elements in this piece, 1..VN_OPT_CHUNK */
    int32_t reserved;
} vnParamChunk;
size_t vn_clip_sgd_workspace_bytes(int32_t n_chunks);
int vn_clip_sgd(const vnParamChunk *chunks, int32_t n_chunks, float max_norm, float lr, int32_t scale_grads,
                void *workspace, size_t workspace_bytes, float *total_norm, vnStream stream);

/* ---- RPN training targets (voxelnet/utils.py:376-473 generate_targets, :344-373 bbox_iou, :213-227
 * anchor_to_standup_box2d; called by RPN3D.forward, voxelnet/model.py:309) -------------------------------
 * Per sample b: gt_count[b] ground-truth boxes in lidar coordinates, gt [B,max_gt,7] = (x,y,z,h,w,l,r) float64 and
 * their stand-up rectangles gt_standup [B,max_gt,4] = (x1,y1,x2,y2) float32 (utils.label_to_gt_box_3d and
 * corner_to_standup_box2d(center_to_corner_box_2d(.)) on the host: O(boxes) work).  anchors [n_anchors,7] float64 is
 * utils.generate_anchors().reshape(-1,7): anchor n = (iy*W + ix)*2 + rotation.
 * Outputs in the reference's channels-last layouts: pos, neg [B,n_anchors] == (B,h,w,2); targets [B,n_anchors,7] ==
 * (B,h,w,14); float32 (model.py:327-329 converts the float64 arrays).  Which anchors are positive / negative is
 * bit-exact with the reference, its IoU quirks included; regression targets are float64 arithmetic rounded to fp32. */
#define VN_TARGETS_MAX_GT 128
size_t vn_rpn_targets_workspace_bytes(int32_t B, int32_t n_anchors, int32_t max_gt);
int vn_rpn_targets(const double *anchors, int32_t n_anchors, const double *gt, const float *gt_standup,
                   const int32_t *gt_count /*device [B]*/, int32_t B, int32_t max_gt, float pos_iou, float neg_iou,
                   double anchor_h, float *pos, float *neg, float *targets, void *workspace, size_t workspace_bytes,
                   vnStream stream);

/* ---- inference tail (voxelnet/model.py:364-395 RPN3D.predict: utils.deltas_to_boxes_3d utils.py:476-489,
 * filter_boxes model.py:28-57, utils.nms utils.py:492-553) ----------------------------------------------
 * probs (B,2,h,w) and deltas (B,14,h,w): the module's fp32 NCHW outputs, read as the flat (B, n_anchors) and
 * (B, n_anchors, 7) views the reference's reshapes take (no permute: utils.py:478, model.py:384); anchors
 * [n_anchors,7] float64.  Per sample: boxes [top_k,7] (x,y,z,h,w,l,r) and scores [top_k] of the kept detections in
 * descending score, counts[b] of them valid.  score_thres = cfg.RPN.SCORE_THRES (>= keeps), nms_thres =
 * cfg.RPN.NMS_THRES, top_k = cfg.RPN.NMS_POST_TOPK <= VN_PREDICT_MAX_TOPK. */
#define VN_PREDICT_MAX_TOPK 64
size_t vn_rpn_predict_workspace_bytes(int32_t B, int32_t n_anchors);
int vn_rpn_predict(const float *probs, const float *deltas, const double *anchors, int32_t B, int32_t n_anchors,
                   float score_thres, double nms_thres, int32_t top_k, double anchor_h, float *boxes, float *scores,
                   int32_t *counts, void *workspace, size_t workspace_bytes, vnStream stream);

#ifdef __cplusplus
}
#endif
#endif /* VOXELNET_HIP_H */
