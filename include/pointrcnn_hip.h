/*
 * pointrcnn_hip.h -- C ABI of libpointrcnn_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the PointRCNN point-cloud hot path: the native launcher
 * functions that the reference's TF custom-op shims call (tf_sampling.cpp,
 * tf_grouping.cpp, tf_interpolate.cpp) and the rotated NMS of utils/nms_rotate.py.
 * Each entry point cites the reference interface it replaces (file:line relative to
 * the reference tree).  Parameter lists are the reference launchers' parameter lists
 * plus a trailing stream and an int status return.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd or torch-ROCm storage) unless
 *     named host_*; tensors are dense row-major fp32 / int32 as in the reference.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work
 *     is enqueued asynchronously on it; no entry point synchronises the device,
 *     allocates or frees memory, or keeps state between calls.
 *   - the caller owns every buffer including workspaces (as TF's allocate_output /
 *     allocate_temp did, tf_sampling.cpp:108-117); *_workspace_bytes() size them.
 *   - return value: PRCNN_OK or a negative prcnn_status; nothing throws.  The shape /
 *     attribute checks mirror the reference's OP_REQUIRES list (tf_sampling.cpp:99,105;
 *     tf_grouping.cpp:71,74,79,84; tf_interpolate.cpp:163,168,203,206).
 *   - re-entrant and thread-safe: kernels are stateless.
 */
#ifndef POINTRCNN_HIP_H
#define POINTRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *prcnn_stream_t; /* hipStream_t */

typedef enum {
    PRCNN_OK = 0,
    PRCNN_BAD_ARG = -1,      /* non-positive attr: npoint, radius, nsample, ...      */
    PRCNN_BAD_SHAPE = -2,    /* negative / inconsistent dimension, NULL tensor       */
    PRCNN_HIP_ERROR = -3,    /* a HIP runtime call or kernel launch failed           */
    PRCNN_WORKSPACE = -4,    /* workspace missing or smaller than *_workspace_bytes  */
    PRCNN_UNSUPPORTED = -5   /* size beyond what the kernels are built for           */
} prcnn_status;

const char *prcnn_status_string(int status);
int prcnn_version(void);
/* Text of the last HIP error seen by the calling thread (diagnostics for PRCNN_HIP_ERROR). */
const char *prcnn_last_hip_error_string(void);

/* ---- sampling ---------------------------------------------------------------------- */

/* farthestpointsamplingLauncher(int b,int n,int m,const float* inp,float* temp,int* out)
 *   tf_sampling.cpp:94, tf_sampling_g.cu:105-170,203-205.
 * inp (b,n,3) -> out (b,m) int32.  out[:,0]=0; m-1 greedy iterations with the
 * reference's fp32 distance (no FMA contraction) and tie-break (max value, then
 * smallest k mod 512, then smallest k).  `temp` is the reference's scratch argument
 * (32*n floats there).  Here clouds of up to 16384 points never touch it (it may be NULL);
 * larger clouds are split over several cooperating workgroups that exchange one 8-byte
 * record per iteration through `temp`, which must then hold prcnn_fps_temp_bytes(b,n)
 * bytes (it is zeroed by the call).  Its first int is an ERROR WORD, valid once the stream has
 * reached the end of the call: 0 = every index was sampled; 1 = a workgroup waited longer than
 * the bound (~seconds) for its partners -- they were not co-resident, e.g. because another
 * process filled the GPU -- and the cloud's remaining indices were set to -1.  Callers that
 * read `out` must check the word (pointrcnn_amd.ops does: `check=True`, or the `fps_error`
 * tensor attached to the result).  Cooperative launches of one process are ordered one after
 * the other on the device (two running side by side could starve each other of CUs), except
 * under stream capture, where the graph's own edges must order them.  With temp == NULL
 * clouds up to 32768 points run on one CU with the min-distance array in LDS; beyond that
 * PRCNN_WORKSPACE is returned. */
int prcnn_farthest_point_sample(int b, int n, int m, const float *inp, float *temp,
                                int *out, prcnn_stream_t stream);
size_t prcnn_fps_temp_bytes(int b, int n);

/* The same sampling (same launcher, tf_sampling.cpp:94 / tf_sampling_g.cu:105-170: identical
 * indices) for callers that keep OTHER kernels in flight beside it.  prcnn_farthest_point_sample
 * holds a cloud in one CU's registers and LDS for the whole chain, so that CU serves nothing else
 * meanwhile; this variant keeps the cloud in `temp` (L2-resident, sorted into buckets of
 * neighbouring points by a short first launch) and samples with four waves and ~50 registers per
 * cloud, which fit beside any other kernel's waves.  Alone it is slower per iteration (one more
 * L2 round trip); n <= 16384.  `temp`: prcnn_fps_shared_temp_bytes(b, n) bytes, 16-byte aligned,
 * overwritten by the call (no state is kept in it between calls). */
size_t prcnn_fps_shared_temp_bytes(int b, int n);
int prcnn_farthest_point_sample_shared(int b, int n, int m, const float *inp, void *temp,
                                       size_t temp_bytes, int *out, prcnn_stream_t stream);

/* gatherpointLauncher(int b,int n,int m,const float* inp,const int* idx,float* out)
 *   tf_sampling.cpp:125, tf_sampling_g.cu:172-181.  out[b,j,:] = inp[b,idx[b,j],:], C=3 */
int prcnn_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out,
                       prcnn_stream_t stream);

/* ---- grouping ---------------------------------------------------------------------- */

/* queryBallPointLauncher(int b,int n,int m,float radius,int nsample,const float* xyz1,
 *                        const float* xyz2,int* idx,int* pts_cnt)
 *   tf_grouping.cpp:66, tf_grouping_g.cu:3-36,125-128.
 * xyz1 (b,n,3) points, xyz2 (b,m,3) centroids -> idx (b,m,nsample), pts_cnt (b,m).
 * First `nsample` points in INDEX ORDER with max(sqrtf(d2),1e-20f) < radius; rows are
 * padded with the first hit.  Rows without any hit get pts_cnt 0 and are ZERO-FILLED (the
 * reference leaves them uninitialised, tf_grouping_g.cu:26-34; zero is what its callers
 * read from freshly cleared memory).  pts_cnt may be NULL. */
int prcnn_query_ball_point(int b, int n, int m, float radius, int nsample,
                           const float *xyz1, const float *xyz2, int *idx, int *pts_cnt,
                           prcnn_stream_t stream);

/* Multi-scale form: the `nscales` (<= 4) ball queries of one MSG set-abstraction level
 * (models/pointnet_util.py:174-177 calls query_ball_point once per radius on the same
 * xyz1 / xyz2) in ONE pass over the points: each squared distance is evaluated once and
 * tested against every radius.  host_* arrays live in HOST memory and are read before the
 * call returns; host_idx[s] / host_pts_cnt[s] are device pointers (b,m,nsample[s]) / (b,m).
 * Results are identical to nscales separate prcnn_query_ball_point calls. */
int prcnn_query_ball_point_msg(int b, int n, int m, int nscales, const float *host_radii,
                               const int *host_nsamples, const float *xyz1, const float *xyz2,
                               int *const *host_idx, int *const *host_pts_cnt,
                               prcnn_stream_t stream);

/* groupPointLauncher(int b,int n,int c,int m,int nsample,const float* points,
 *                    const int* idx,float* out)   tf_grouping.cpp:142, tf_grouping_g.cu:40-57
 * out[b,j,k,:] = points[b,idx[b,j,k],:]  -> (b,m,nsample,c) */
int prcnn_group_point(int b, int n, int c, int m, int nsample, const float *points,
                      const int *idx, float *out, prcnn_stream_t stream);

/* ---- 3d_interpolation -------------------------------------------------------------- */

/* threenn_cpu(int b,int n,int m,const float* xyz1,const float* xyz2,float* dist,int* idx)
 *   tf_interpolate.cpp:60-103.  For each of the n points of xyz1 the 3 nearest of the m
 * points of xyz2: SQUARED fp32 distances ascending + indices; ties keep the earlier
 * index; with m<3 the missing slots are (+inf, 0). */
int prcnn_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                   int *idx, prcnn_stream_t stream);

/* threeinterpolate_cpu(int b,int m,int c,int n,const float* points,const int* idx,
 *                      const float* weight,float* out)    tf_interpolate.cpp:107-127
 * NOTE the (b, m, c, n) order.  out[b,j,:] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3, evaluated
 * left to right without FMA contraction (bit-exact with the CPU reference). */
int prcnn_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                            const float *weight, float *out, prcnn_stream_t stream);

/* Inverse-distance weights of pointnet_fp_module, models/pointnet_util.py:211-214:
 *   dist=max(dist,1e-10); norm=sum(1/dist); weight=(1/dist)/norm      rows = b*n */
int prcnn_fp_weights(int64_t rows, const float *dist, float *weight, prcnn_stream_t stream);

/* Fused three_nn + weights + three_interpolate (pointnet_util.py:210-215 in one launch;
 * additive, the three primitives above stay available).  xyz1 (b,n,3), xyz2 (b,m,3),
 * points2 (b,m,c) -> out rows of c floats written at out + row*ldo (ldo >= c lets the
 * caller write straight into the left part of the FP concat buffer,
 * pointnet_util.py:218).  dist / idx (b,n,3) are optional outputs (may be NULL). */
int prcnn_three_nn_interpolate(int b, int n, int m, int c, const float *xyz1,
                               const float *xyz2, const float *points2, float *out,
                               int64_t ldo, float *dist, int *idx, prcnn_stream_t stream);

/* ---- rotated BEV NMS --------------------------------------------------------------- */

/* nms_rotate_cpu(boxes, scores, iou_threshold, max_output_size)  utils/nms_rotate.py:39-73
 * boxes (n,5) float64 [x_c, y_c, w, h, theta_deg] exactly as the reference callers build
 * them (dataset/frustum_dataset.py:126-128), scores (n,) float64.  Greedy NMS in
 * descending score order (equal scores: HIGHER index first = scores.argsort(kind='stable')[::-1],
 * the reference's line :41 with a stable sort; its default sort leaves ties unspecified), rectangles as OpenCV RotatedRect
 * ((b1,b0),(b3,b2),b4), iou = I/(a1+a2-I+1e-7), suppress when iou >= threshold, stop at
 * max_output_size.  keep receives up to min(n,max_output_size) int64 indices into the
 * ORIGINAL arrays, score-descending; *n_keep (device int) the count.  All arithmetic is
 * float64 on the device.  workspace: prcnn_nms_rotate_workspace_bytes(n) bytes. */
size_t prcnn_nms_rotate_workspace_bytes(int n);
int prcnn_nms_rotate(int n, const double *boxes, const double *scores, double iou_threshold,
                     int max_output_size, int64_t *keep, int *n_keep, void *workspace,
                     size_t workspace_bytes, prcnn_stream_t stream);

/* The same NMS on `batch` independent box sets of equal size n in one set of launches
 * (one frame per set: dataset/frustum_dataset.py:117-137 runs it once per frame; a batch
 * of frames is what one GPU holds).  boxes (batch,n,5), scores (batch,n), keep
 * (batch,max_output_size), n_keep (batch).  No counterpart in the reference beyond the
 * per-frame loop; prcnn_nms_rotate is this call with batch = 1. */
size_t prcnn_nms_rotate_batched_workspace_bytes(int batch, int n);
int prcnn_nms_rotate_batched(int batch, int n, const double *boxes, const double *scores,
                             double iou_threshold, int max_output_size, int64_t *keep,
                             int *n_keep, void *workspace, size_t workspace_bytes,
                             prcnn_stream_t stream);

/* prcnn_nms_rotate without a workspace argument, for callers that CANNOT ALLOCATE device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no
 * counterpart in the reference.  The workspace is a static device array inside the library
 * (about 11.3 MB per GPU the library runs on, held until the process ends), so
 * n <= PRCNN_NMS_NOALLOC_MAX_N; larger n returns PRCNN_UNSUPPORTED (use prcnn_nms_rotate).  The call
 * allocates and frees nothing and gives the same keep / n_keep as prcnn_nms_rotate.  The stream's
 * device must be the current device.  All calls on one GPU share the array: calls that overlap
 * on one GPU (different streams or host threads) race; calls on one stream are ordered. */
#define PRCNN_NMS_NOALLOC_MAX_N 9216
int prcnn_nms_rotate_noalloc(int n, const double *boxes, const double *scores, double iou_threshold,
                             int max_output_size, int64_t *keep, int *n_keep, prcnn_stream_t stream);

/* ---- shared MLP (1x1 conv + folded BN + ReLU) and pooling --------------------------- */

/* One shared-MLP layer, models/tf_util.py:120-185 (conv2d 1x1 + bias + batch_norm +
 * ReLU) with the inference-time BN folded by the caller into scale/shift:
 *   out[r, o] = act( scale[o] * sum_k x[r,k] * w[k,o] + shift[o] )
 * x rows at x + r*ldx (cin floats), w (cin,cout) row-major, out rows at out + r*ldo.
 * If pool > 1, consecutive groups of `pool` rows are max-reduced on the fly
 * (pointnet_util.py:126,192) and out has rows/pool rows.  precision: 0 = fp32 MFMA
 * (exact fp32 FMA chain), 1 = bf16 MFMA with fp32 accumulation. */
int prcnn_mlp_layer(int64_t rows, int cin, int cout, const float *x, int64_t ldx,
                    const float *w, const float *scale, const float *shift, int relu,
                    int pool, int precision, float *out, int64_t ldo, prcnn_stream_t stream);

/* First layer of a set-abstraction scale fused with the grouping
 * (pointnet_util.py:177-190 / :43-53 in one launch): the (b,m,nsample,C+3) grouped
 * tensor is never materialised.  Row (b,j,k) of the implicit input is
 *   xyz_first == 0 (MSG, :183): [ feats[b,idx,:] , xyz[b,idx,:]-new_xyz[b,j,:] ]
 *   xyz_first == 1 (SSG, :49):  [ xyz[b,idx,:]-new_xyz[b,j,:] , feats[b,idx,:] ]
 * feats may be NULL (c=0: points=None, :52/:185).  Followed by the same epilogue as
 * prcnn_mlp_layer (w is (c+3, cout)). */
int prcnn_sa_group_mlp_layer(int b, int n, int m, int nsample, int c, const float *xyz,
                             const float *feats, const float *new_xyz, const int *idx,
                             int xyz_first, int cout, const float *w, const float *scale,
                             const float *shift, int relu, int pool, int precision,
                             float *out, int64_t ldo, prcnn_stream_t stream);

/* The same first layer with its feature half hoisted out of the grouping (additive; exact algebra,
 * fp32 re-association): the convolution is linear, so  [feats[idx], xyz[idx]-new_xyz] . W  =
 * (feats . W_feat)[idx] + (xyz[idx]-new_xyz) . W_xyz.  `table` (b*n rows of ldt floats, cout used)
 * holds feats . W_feat per POINT (one prcnn_mlp_layer over b*n rows instead of b*m*nsample
 * grouped rows); this call gathers table rows by idx, adds the 3-column xyz product (w_xyz is
 * (3, cout)) and applies the epilogue of prcnn_mlp_layer. */
int prcnn_sa_group_add_layer(int b, int n, int m, int nsample, const float *xyz,
                             const float *new_xyz, const int *idx, const float *table,
                             int64_t ldt, int cout, const float *w_xyz, const float *scale,
                             const float *shift, int relu, int pool, int precision, float *out,
                             int64_t ldo, prcnn_stream_t stream);

/* A whole hoisted scale in one launch (additive): layer 1 as in prcnn_sa_group_add_layer (table
 * rows gathered by idx + the 3-column product with the relative coordinates, folded BN, ReLU),
 * then two more 1x1 convolutions (+ folded BN + ReLU) on the matrix cores and the max over
 * nsample -- the activations between the layers never leave the CU (models/pointnet_util.py:
 * 177-192 for one radius).  c1 <= 64, c2 <= 96, c3 <= 128, nsample 16 or 32, fp32; other shapes:
 * PRCNN_UNSUPPORTED (use the per-layer entry points).  table (b*n, ldt) holds this scale's c1
 * columns at `table`; w_xyz (3,c1); out rows (b*m) at out + r*ldo. */
int prcnn_sa_group_add_mlp2_pool(int b, int n, int m, int nsample, const float *xyz,
                                 const float *new_xyz, const int *idx, const float *table,
                                 int64_t ldt, int c1, int c2, int c3, const float *w_xyz,
                                 const float *s1, const float *t1, const float *w2,
                                 const float *s2, const float *t2, const float *w3,
                                 const float *s3, const float *t3, float *out, int64_t ldo,
                                 prcnn_stream_t stream);

/* A whole set-abstraction scale in one launch for NARROW layers: grouping + three shared-MLP
 * layers + max over nsample (pointnet_util.py:177-192 for one radius; :43-53,116-126 for
 * SSG), fp32 matrix cores, no intermediate activation in HBM.  Serves c+3 <= 8, c1,c2 <= 32,
 * c3 <= 128, nsample 16 or 32 (e.g. the first RPN level, models/rpn.py:166-169); anything
 * else returns PRCNN_UNSUPPORTED and the caller uses prcnn_sa_group_mlp_layer +
 * prcnn_mlp_layer.  w_i are (cin_i, c_i) row-major; out (b*m, c3) rows at out + g*ldo. */
int prcnn_sa_group_mlp3_pool(int b, int n, int m, int nsample, int c, const float *xyz,
                             const float *feats, const float *new_xyz, const int *idx,
                             int xyz_first, int c1, int c2, int c3, const float *w1,
                             const float *s1, const float *t1, const float *w2, const float *s2,
                             const float *t2, const float *w3, const float *s3, const float *t3,
                             float *out, int64_t ldo, prcnn_stream_t stream);

/* ---- duplicate-free groups (additive; identical results) -----------------------------------
 * query_ball_point pads a ball of cnt < nsample points with copies of its first hit
 * (tf_grouping_g.cu:26-29): the grouped tensor repeats that row and the max over the group does
 * not need the copies.  prcnn_ball_count_classes sorts the `groups` = b*m centroids of one scale
 * by pts_cnt into six classes (cnt <= 1, 2, 4, 8, 16, 32): cls_list (6, groups) int32 holds the
 * centroid numbers of each class (order inside a class unspecified), cls_count (8 int32, the
 * first six used) their sizes.  The *_classes forms of the chain kernels below then evaluate only
 * the first 1 / 2 / 4 / 8 / 16 / 32 slots of each centroid of a class: bit-identical output, work
 * proportional to the distinct rows.  nsample must be 16 or 32 as for the plain forms. */
int prcnn_ball_count_classes(int64_t groups, const int *pts_cnt, int *cls_list, int *cls_count,
                             prcnn_stream_t stream);
int prcnn_sa_group_mlp3_pool_classes(int b, int n, int m, int nsample, int c, const float *xyz,
                                     const float *feats, const float *new_xyz, const int *idx,
                                     int xyz_first, int c1, int c2, int c3, const float *w1,
                                     const float *s1, const float *t1, const float *w2,
                                     const float *s2, const float *t2, const float *w3,
                                     const float *s3, const float *t3, float *out, int64_t ldo,
                                     const int *cls_list, const int *cls_count,
                                     prcnn_stream_t stream);
int prcnn_sa_group_add_mlp2_pool_classes(int b, int n, int m, int nsample, const float *xyz,
                                         const float *new_xyz, const int *idx, const float *table,
                                         int64_t ldt, int c1, int c2, int c3, const float *w_xyz,
                                         const float *s1, const float *t1, const float *w2,
                                         const float *s2, const float *t2, const float *w3,
                                         const float *s3, const float *t3, float *out, int64_t ldo,
                                         const int *cls_list, const int *cls_count,
                                         prcnn_stream_t stream);

/* The same idea for scales too wide for the chain kernels, as a compacted row list: centroid g
 * contributes its max(cnt,1) distinct slots as rows off[g] .. off[g+1]-1 (rowpi: point row b*n
 * space, rowci: centroid number; all int32, capacity b*m*nsample rows; off has b*m+1 entries;
 * *rows_dev = number of rows).  prcnn_sa_rows_add_layer is prcnn_sa_group_add_layer over such rows,
 * prcnn_mlp_layer_n is prcnn_mlp_layer with the row count read from device memory (max_rows sizes
 * the grid; tiles past *rows_dev leave at once), and prcnn_segment_max takes the maximum over
 * each centroid's rows: together they give the pooled features of the layer-by-layer path, bit
 * for bit, with work proportional to the distinct rows. */
int prcnn_ball_compact_rows(int b, int n, int m, int nsample, const int *idx, const int *pts_cnt,
                            int *rowpi, int *rowci, int *off, int *rows_dev, prcnn_stream_t stream);
int prcnn_sa_rows_add_layer(int64_t max_rows, const int *rows_dev, const int *rowpi, const int *rowci,
                            const float *xyz, const float *new_xyz, const float *table, int64_t ldt,
                            int cout, const float *w_xyz, const float *scale, const float *shift,
                            int relu, int precision, float *out, int64_t ldo, prcnn_stream_t stream);
int prcnn_mlp_layer_n(int64_t max_rows, const int *rows_dev, int cin, int cout, const float *x,
                      int64_t ldx, const float *w, const float *scale, const float *shift, int relu,
                      int precision, float *out, int64_t ldo, prcnn_stream_t stream);
int prcnn_segment_max(int64_t groups, int c, const float *x, int64_t ldx, const int *off, float *out,
                      int64_t ldo, prcnn_stream_t stream);

/* Max over consecutive groups of `nsample` rows, models/pointnet_util.py:126,192
 * (tf.reduce_max over the nsample axis) for group sizes the fused epilogue of
 * prcnn_mlp_layer does not cover (pool must there be 16, 32, 64 or 128).
 * x rows at x + r*ldx (c floats), out (groups, c) rows at out + g*ldo. */
int prcnn_max_pool(int64_t groups, int nsample, int c, const float *x, int64_t ldx, float *out,
                   int64_t ldo, prcnn_stream_t stream);

/* ---- spatially culled three_nn (additive; same results as prcnn_three_nn) -----------------
 * For large n*m: queries and known points are bucketed into a Morton-ordered grid, known
 * points are cut into boxed tiles of 64, and each wave of 64 neighbouring queries visits
 * tiles nearest-first until the nearest unvisited tile is farther than its worst
 * third-best distance.  The bound uses the same unfused fp32 operations as the distance
 * (monotone under rounding), and insertion orders (distance, index) pairs, so dist/idx
 * are bit-identical to threenn_cpu (tf_interpolate.cpp:60-103) for every input.
 * workspace: *_workspace_bytes(), 256-byte aligned, contents irrelevant on entry. */
size_t prcnn_three_nn_sorted_workspace_bytes(int b, int n, int m);
int prcnn_three_nn_sorted(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                          int *idx, void *workspace, size_t workspace_bytes, prcnn_stream_t stream);
/* prcnn_three_nn_interpolate with the culled search (dist / idx may be NULL). */
int prcnn_three_nn_interpolate_sorted(int b, int n, int m, int c, const float *xyz1,
                                      const float *xyz2, const float *points2, float *out,
                                      int64_t ldo, float *dist, int *idx, void *workspace,
                                      size_t workspace_bytes, prcnn_stream_t stream);

/* out[:, :c] = out[:, :c] + interpolate(points2) (then ReLU if relu != 0): the epilogue of the
 * split FP layer  relu(BN([interp, points1] W)) = relu(interp3((points2 W_a) s) + (points1 W_b) s + t),
 * which runs the wide half of the first FP convolution on the m known rows instead of the
 * n = 4m interpolated rows (models/pointnet_util.py:215-226).  workspace: NULL = brute-force
 * search, else prcnn_three_nn_sorted_workspace_bytes() bytes for the culled search. */
int prcnn_three_nn_interpolate_add(int b, int n, int m, int c, const float *xyz1,
                                   const float *xyz2, const float *points2, float *out,
                                   int64_t ldo, int relu, void *workspace, size_t workspace_bytes,
                                   prcnn_stream_t stream);

/* The same with the added term evaluated in the epilogue instead of being read from `out`
 * (additive): out[q][:c] = act(interpolate(points2)[q] + p1[q][:c1] . u + t) with p1 (b,n,c1),
 * u (c1,c), t (c), c1 <= 4 (c1 == 0: t only).  It is the points1 half of the split first FP
 * convolution (pointnet_util.py:218-226) when points1 is narrow (the RPN's last FP level
 * concatenates the 1-channel intensity): no (b,n,c) tensor is written and read back. */
int prcnn_three_nn_interpolate_affine(int b, int n, int m, int c, const float *xyz1,
                                      const float *xyz2, const float *points2, const float *p1,
                                      int c1, const float *u, const float *t, float *out,
                                      int64_t ldo, int relu, void *workspace,
                                      size_t workspace_bytes, prcnn_stream_t stream);

/* out[:, :c] = out[:, :c] + three_interpolate(points, idx, weight) (then ReLU if relu != 0),
 * rows of out at out + r*ldo: the element-parallel form of the epilogue above, for levels
 * whose search is small (idx / weight from prcnn_three_nn + prcnn_fp_weights). */
int prcnn_three_interpolate_add(int b, int m, int c, int n, const float *points, const int *idx,
                                const float *weight, float *out, int64_t ldo, int relu,
                                prcnn_stream_t stream);

/* prcnn_three_nn_interpolate_affine without the search (additive): idx / weight (b,n,3) come
 * from prcnn_three_nn[_sorted] + prcnn_fp_weights, e.g. computed ahead of the feature phase on
 * another stream (they depend on coordinates only).  out[q][:c] = act(three_interpolate(points,
 * idx, weight)[q] + p1[q][:c1] . u + t); same arithmetic, same bits as the fused entry point. */
int prcnn_three_interpolate_affine(int b, int m, int c, int n, const float *points, const int *idx,
                                   const float *weight, const float *p1, int c1, const float *u,
                                   const float *t, float *out, int64_t ldo, int relu,
                                   prcnn_stream_t stream);

/* prcnn_query_ball_point_msg with the spatially culled search (additive; identical idx /
 * pts_cnt): points are bucketed and tiled, a wave of four neighbouring centroids visits only
 * tiles within the largest radius, and the per-ball hits are reduced to the nsample smallest
 * indices in ascending order.  nsample <= 64 per scale.  workspace as above. */
size_t prcnn_query_ball_point_sorted_workspace_bytes(int b, int n, int m);
int prcnn_query_ball_point_msg_sorted(int b, int n, int m, int nscales, const float *host_radii,
                                      const int *host_nsamples, const float *xyz1,
                                      const float *xyz2, int *const *host_idx,
                                      int *const *host_pts_cnt, void *workspace,
                                      size_t workspace_bytes, prcnn_stream_t stream);

/* ---- one grid order for several searches over the same two point sets (additive) ---------------
 * The culled searches above start by ordering both point sets along the cells of a grid over their
 * JOINT bounding box.  The order depends on the two sets only, not on which is the query side, so
 * searches over the same pair -- in the RPN stack the level-1 ball query (points = the cloud,
 * centroids = level-1 new_xyz, models/pointnet_util.py:141-160 via rpn.py:166) and the last FP
 * level's three_nn (unknown = the cloud, known = level-1 new_xyz, pointnet_util.py:210 via
 * rpn.py:192) -- can share it: prcnn_grid_order once, then the *_ordered entry points, which skip
 * their own ordering pass.  order1 (b,n) / order2 (b,m): point indices of xyz1 / xyz2 in grid
 * order (the order inside a cell is unspecified; no result depends on it).  The *_ordered entry
 * points take (order1, order2) for THEIR (xyz1, xyz2): swap the two when the roles are swapped.
 * Both NULL = same as the entry point without the suffix.  Results are bit-identical either way.
 * The arrays must be what prcnn_grid_order wrote for this (b, n, m) and these coordinates, on the
 * same stream or ordered after it: their entries are used as gather indices into xyz1 / xyz2 and are
 * NOT range-checked (an array from elsewhere is a caller error like a wrong pointer). */
int prcnn_grid_order(int b, int n, int m, const float *xyz1, const float *xyz2, int *order1,
                     int *order2, prcnn_stream_t stream);
int prcnn_three_nn_ordered(int b, int n, int m, const float *xyz1, const float *xyz2,
                           const int *order1, const int *order2, float *dist, int *idx,
                           void *workspace, size_t workspace_bytes, prcnn_stream_t stream);
int prcnn_three_nn_interpolate_affine_ordered(int b, int n, int m, int c, const float *xyz1,
                                              const float *xyz2, const int *order1,
                                              const int *order2, const float *points2,
                                              const float *p1, int c1, const float *u,
                                              const float *t, float *out, int64_t ldo, int relu,
                                              void *workspace, size_t workspace_bytes,
                                              prcnn_stream_t stream);
int prcnn_query_ball_point_msg_ordered(int b, int n, int m, int nscales, const float *host_radii,
                                       const int *host_nsamples, const float *xyz1,
                                       const float *xyz2, const int *order1, const int *order2,
                                       int *const *host_idx, int *const *host_pts_cnt,
                                       void *workspace, size_t workspace_bytes,
                                       prcnn_stream_t stream);

/* ---- backward ops ------------------------------------------------
 * All three are ordered segmented reductions: every output element is summed in exactly
 * the order of the reference's serial CPU loops, so results are deterministic and equal
 * to those loops bit for bit (the reference's GPU kernels use float atomicAdd, whose
 * order varies).  The output needs no prior memset (the reference ops clear it first:
 * tf_sampling.cpp:174, tf_grouping.cpp:204, tf_interpolate.cpp:258); every row is written.
 * workspace: device memory of at least *_workspace_bytes(), 4-byte aligned, contents
 * irrelevant on entry.  Indices outside [0, rows) are ignored. */

/* scatteraddpointLauncher(b,n,m,out_g,idx,inp_g)  models/tf_ops/sampling/tf_sampling_g.cu:212-214
 * (kernel :183-192; op GatherPointGrad tf_sampling.cpp:151-177).
 * out_g (b,m,3), idx (b,m) -> inp_g (b,n,3):  inp_g[b,idx[b,j],:] += out_g[b,j,:]. */
size_t prcnn_gather_point_grad_workspace_bytes(int b, int n, int m);
int prcnn_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                            void *workspace, size_t workspace_bytes, prcnn_stream_t stream);

/* groupPointGradLauncher(b,n,c,m,nsample,grad_out,idx,grad_points)
 * models/tf_ops/grouping/tf_grouping.cpp:173 (kernel tf_grouping_g.cu:61-78, CPU twin
 * group_point_grad_cpu query_ball_point.cpp:70-84).
 * grad_out (b,m,nsample,c), idx (b,m,nsample) -> grad_points (b,n,c). */
size_t prcnn_group_point_grad_workspace_bytes(int b, int n, int m, int nsample);
int prcnn_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                           const int *idx, float *grad_points, void *workspace,
                           size_t workspace_bytes, prcnn_stream_t stream);

/* threeinterpolate_grad_cpu(b,n,c,m,grad_out,idx,weight,grad_points)
 * models/tf_ops/3d_interpolation/tf_interpolate.cpp:131-153 (standalone twin
 * interpolate_grad_cpu interpolate.cpp:108-130).
 * grad_out (b,n,c), idx (b,n,3), weight (b,n,3) -> grad_points (b,m,c):
 * grad_points[b,idx[b,j,t],l] += grad_out[b,j,l]*weight[b,j,t], t = 0,1,2 in that order. */
size_t prcnn_three_interpolate_grad_workspace_bytes(int b, int n, int m);
int prcnn_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                 const float *weight, float *grad_points, void *workspace,
                                 size_t workspace_bytes, prcnn_stream_t stream);

/* ---- kNN grouping -------------------------------------------------------
 * selectionSortLauncher(b,n,m,k,dist,outi,out)  models/tf_ops/grouping/tf_grouping.cpp:106
 * (kernel selection_sort_gpu tf_grouping_g.cu:83-123, CPU twin selection_sort_cpu
 * selection_sort.cpp:20-63): dist (b,m,n) -> outi (b,m,n) int32, out (b,m,n) f32, the rows
 * after k steps of "swap the first minimum of positions s.. to position s" (the first k entries
 * are the k smallest, ascending; ties in the reference's swap order, reproduced exactly).
 * 0 < k <= min(n, 128). */
int prcnn_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out,
                         prcnn_stream_t stream);

/* knn_point(k, xyz1, xyz2)  models/tf_ops/grouping/tf_grouping.py:49-74, fused: the (b,m,n)
 * matrix reduce_sum((xyz1-xyz2)**2,-1) is never materialised.  xyz1 (b,n,3) dataset, xyz2
 * (b,m,3) queries -> val (b,m,k) squared distances ascending, idx (b,m,k) int32; identical to
 * prcnn_selection_sort on that matrix, sliced to k. */
int prcnn_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val,
                    int *idx, prcnn_stream_t stream);

/* ---- RPN proposal post-processing (TensorFlow graph ops in the
 * reference, no native kernel) ----------------------------------------------------------------
 * BoxEncoder.tf_decode (utils/box_encoder.py:109-179) + "box_center + fg_points_xyz"
 * (models/rpn.py:109) + get_box3d_corners_helper (models/model_util.py:55-82) + the confidence
 * product and BEV boxes of RPN.reduce_proposals (models/rpn.py:213-227), one pass per proposal.
 * head_output (b,n,4*NCB+1+2*NHB+4*NSC) laid out as models/rpn.py:76-104; fg_logits (b,n,2) or
 * NULL (segmentation score 1); host_mean_sizes NSC x (l,h,w).  Any output may be NULL.
 * center (b,n,3), angle (b,n), size (b,n,3), corners (b,n,8,3), confidence (b,n),
 * bev_boxes (b,n,4) = [min x, min z, max x, max z]. */
int prcnn_rpn_decode_proposals(int b, int n, int num_center_bin, int num_heading_bin,
                               int num_size_cluster, float center_search_range,
                               float heading_search_range, const float *host_mean_sizes,
                               const float *head_output, const float *fg_points_xyz,
                               const float *fg_logits, float *center, float *angle, float *size,
                               float *corners, float *confidence, float *bev_boxes,
                               prcnn_stream_t stream);

/* tf.image.non_max_suppression(boxes, scores, max_output_size, iou_threshold) per frame
 * (models/rpn.py:236; TensorFlow NonMaxSuppression semantics: decreasing score, drop when IoU
 * with a selected box is > iou_threshold, (y1,x1,y2,x2) in any corner order, boxes of
 * non-positive area neither suppress nor are suppressed).  boxes (batch,n,4) f32, scores
 * (batch,n) f64 (exact widening of the f32 scores), keep (batch,max_output_size) int64,
 * n_keep (batch).  Equal scores: lower index first (TensorFlow's heap order is unspecified).
 * workspace: prcnn_nms_rotate_batched_workspace_bytes(batch, n). */
int prcnn_nms_aabb_batched(int batch, int n, const float *boxes, const double *scores,
                           float iou_threshold, int max_output_size, int64_t *keep, int *n_keep,
                           void *workspace, size_t workspace_bytes, prcnn_stream_t stream);

/* nms_on_bev / group_overlaps (test/test_frustum.py:191-218, get_iou :185-189): per frame, in index
 * order, the first remaining detection groups every later remaining detection whose rotated BEV
 * IoU with it is >= iou_threshold (> 0); the member with the highest score (earliest of equals)
 * is kept.  boxes (frames,n,5) f64 rows [x, z, l, w, ry] (obj.t[0], obj.t[2], obj.l, obj.w,
 * obj.ry of compute_box_3d, kitti/kitti_util.py:322-339), scores (frames,n) f64, counts
 * (frames) = detections present per frame (NULL: n).  group_id (frames,n): group of each
 * detection (-1 past counts); keep (frames,n): kept detection of group g at [g], -1 past n_keep.
 * workspace: prcnn_nms_rotate_batched_workspace_bytes(frames, n); n <= 65536. */
int prcnn_bev_group_keep(int frames, int n, const int *counts, const double *boxes,
                         const double *scores, double iou_threshold, int *group_id, int *keep,
                         int *n_keep, void *workspace, size_t workspace_bytes, prcnn_stream_t stream);

/* ---- scene-sharded harness (no counterpart in the reference, which is
 * single-GPU) ---------------------------------------------------------------------------------
 * One float32 record per scene for the per-step all-gather:
 *   rec[s] = [scene_ids[s], n_keep[s], max_det x (boxes3d[s, keep[s,d], 0:7], scores[s, keep[s,d]])]
 * with zeros past n_keep[s].  boxes3d (scenes,nprop,7) f64, scores (scenes,nprop) f64,
 * keep (scenes,max_det) int64 and n_keep (scenes) int32 as produced by
 * prcnn_nms_rotate_batched; rec (scenes, 2 + 8*max_det) f32. */
int prcnn_pack_detections(int scenes, int nprop, int max_det, const float *scene_ids,
                          const double *boxes3d, const double *scores, const int64_t *keep,
                          const int *n_keep, float *rec, prcnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* POINTRCNN_HIP_H */
