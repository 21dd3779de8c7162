/*
 * geom_hip.h -- C ABI of libgeom_hip.so: the MI355X (gfx950) implementation of the
 * GEOMetrics per-step hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller
 *     allocates every output and every workspace (the library never allocates
 *     or frees device memory, so all calls are hipGraph-capture safe);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); kernels
 *     are enqueued on it and the call returns without synchronising;
 *   - return value: 0 on success, a positive hipError_t if a launch failed,
 *     a negative GEOM_E* code for a rejected argument.  Nothing is printed.
 *     (The reference swallows launch errors with printf:
 *      chamfer_distance/chamfer_distance.cu:70-72, tri_distance/tri_distance.cu:225-227.)
 *   - layouts are the reference's: contiguous AoS float32 [B,N,3], int32 index
 *     outputs (chamfer_distance/chamfer_distance.py:16-26, tri_distance/tri_distance.py:22-30).
 *
 * Paths in the comments are relative to the reference checkout (EdwardSmith1884/GEOMetrics).
 */
#ifndef GEOM_HIP_H
#define GEOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEOM_ABI_VERSION 14

/* argument errors */
#define GEOM_EINVAL   (-1) /* bad size / null pointer */
#define GEOM_ETOOBIG  (-2) /* a dimension exceeds what the index encoding supports */
#define GEOM_EUNSUPPORTED (-3) /* the shape is outside a fast path; call the general entry point instead */

/* flags (bit set) */
#define GEOM_FLAG_REF_TAIL_TRUNC 1u /* reproduce the shipped CUDA kernels' tail truncation:
                                       chamfer_distance.cu:31-33, tri_distance.cu:129-134 */
#define GEOM_FLAG_FIX_REGION6    2u /* tri: walk option 6 along CA instead of the reference's AB (tri_distance.cu:180, Q2) */
#define GEOM_FLAG_TRI_BRUTE_FORCE 4u /* tri: evaluate the full decision tree for every pair (no sphere culling);
                                       same result, kept as the in-library cross-check of the culled scan */
#define GEOM_FLAG_NN_FMA 8u          /* chamfer: the FMA-contracted distance fma(dz,dz,fma(dx,dx,dy*dy)) -- what a contracting
                                      * compiler (gcc -mfma -ffp-contract=fast; nvcc by default) makes of the reference's source
                                      * line.  Second pinned arithmetic: bit-identical to the reference nnsearch built
                                      * that way; distances differ from the default un-fused form by <= 1 ulp-level round-off */

#define GEOM_FLAG_TRI_WS_READY 16u    /* geom_surface_scan_f32: `workspace` still holds the triangle records a previous call
                                      * wrote for the SAME verts / faces / order / flags: the prep launch is skipped */

int geom_abi_version(void);
/* static string for a code returned by any entry point */
const char *geom_strerror(int code);

/* ---- Chamfer nearest neighbour, both directions in one launch --------------------
 * Replaces ChamferDistanceKernelLauncher (chamfer_distance/chamfer_distance.cpp:4-12,
 * chamfer_distance/chamfer_distance.cu:57-73) as bound by cd.forward_cuda
 * (chamfer_distance.cpp:15-27,36-38).
 *   xyz  [b,n,3], xyz2 [b,m,3]
 *   result  [b,n] / result_i  [b,n]: squared distance / index of the nearest xyz2 point of each xyz point
 *   result2 [b,m] / result2_i [b,m]: the other direction
 * Arg-min is the sequential-scan one: first target seeds, strict '<', lowest index wins ties. */
int geom_chamfer_nn_f32(int b, int n, const float *xyz, int m, const float *xyz2,
                        float *result, int *result_i, float *result2, int *result2_i,
                        unsigned flags, void *stream);

/* The same result -- bit for bit, for ANY orders -- from the culled scan: order1 / order2 (int32 [b,n] / [b,m] permutations,
 * visiting position -> original index; NULL = the clouds' own order) list each cloud so that neighbours in space are
 * neighbours in the list (e.g. a Morton / k-d order); the scan then skips every run of 16 targets whose bounding sphere
 * proves it irrelevant to all 64 queries of a tile.  A bad order only costs speed.  workspace:
 * geom_chamfer_nn_culled_workspace_floats(b, n, m) floats, 16-byte aligned.  flags: 0 or GEOM_FLAG_NN_FMA. */
int64_t geom_chamfer_nn_culled_workspace_floats(int b, int n, int m);
int geom_chamfer_nn_culled_f32(int b, int n, const float *xyz, int m, const float *xyz2, const int *order1, const int *order2,
                               float *result, int *result_i, float *result2, int *result2_i, unsigned flags, float *workspace,
                               void *stream);

/* ---- point -> triangle distance ------------------------------------------------------
 * Replaces TriDistanceKernelLauncher (tri_distance/tri_distance.cpp:4-13,
 * tri_distance/tri_distance.cu:213-228) as bound by tri.forward_cuda (tri_distance.cpp:16-30,34-36).
 *   xyz [b,n,3]; tri1,tri2,tri3 [b,m,3] = corner A,B,C of every face
 *   dist [b,n] squared distance to the chosen closest point, point [b,n] region code 0..6,
 *   index [b,n] winning triangle. */
int geom_tri_distance_f32(int b, int n, const float *xyz, int m,
                          const float *tri1, const float *tri2, const float *tri3,
                          float *dist, int *point, int *index, unsigned flags, void *stream);

/* Same result with the corners gathered in-kernel: verts [b,nv,3], faces [nf,3] int64
 * shared by the batch (what utils.py:467-469 materialises with three index_selects). */
int geom_tri_distance_indexed_f32(int b, int n, const float *xyz, int nv, const float *verts,
                                  int nf, const int64_t *faces,
                                  float *dist, int *point, int *index, unsigned flags, void *stream);

/* Workspace variants (what the python operators call): identical results; the scan reads
 * per-triangle {bounding sphere, corners} records that a prep kernel writes into `workspace`
 * (device memory, 16-byte aligned, at least geom_tri_distance_workspace_bytes(b, n, m) bytes,
 * contents undefined on entry and exit), which removes all per-block staging work; with few query
 * tiles the triangle range of a tile is split over several workgroups whose partial results are
 * merged through 64-bit atomic-min words that also live in the workspace.
 *
 * `order` (device int32 [m], may be NULL) is a permutation of the triangle indices that makes
 * consecutive entries spatially close (e.g. a Morton order of the centroids; the identity when the
 * triangle list is already coherent).  When given, triangles are visited in that order in groups of 16
 * under a group bounding sphere (two-level scan: ~10x fewer sphere tests).  It NEVER changes the result --
 * `index` is always the original triangle index, ties still go to the lowest one; an incoherent order only
 * costs time (then pass NULL: flat scan).  Entries outside [0, m) drop their slot. */
size_t geom_tri_distance_workspace_bytes(int b, int n, int m);
int geom_tri_distance_ws_f32(int b, int n, const float *xyz, int m,
                             const float *tri1, const float *tri2, const float *tri3, const int *order,
                             float *dist, int *point, int *index, unsigned flags,
                             void *workspace, size_t workspace_bytes, void *stream);
int geom_tri_distance_indexed_ws_f32(int b, int n, const float *xyz, int nv, const float *verts,
                                     int nf, const int64_t *faces, const int *order,
                                     float *dist, int *point, int *index, unsigned flags,
                                     void *workspace, size_t workspace_bytes, void *stream);

/* geom_tri_distance_indexed_ws_f32 followed by geom_p2tri_loss_fwd_f32 on its result (the gt-side half of
 * batch_point_to_surface, utils.py:464-481) as ONE call: with a coherent `order` the two-level scan writes sqdist /
 * closest / weights from its own epilogue and no second launch happens. */
int geom_tri_surface_fwd_f32(int b, int n, const float *xyz, int nv, const float *verts, int nf,
                             const int64_t *faces, const int *order, float *dist, int *point, int *index,
                             float *sqdist, float *closest, float *weights, unsigned flags, void *workspace,
                             size_t workspace_bytes, void *stream);

/* ---- differentiable face sampling (utils.py:590-633) -------------------------------------
 * areas[b,nf] = 0.5*|(v0-v1) x (v1-v2)|, the un-normalised multinomial weights (utils.py:596-602). */
int geom_face_areas_f32(int b, int nv, const float *verts, int nf, const int64_t *faces,
                        float *areas, void *stream);
/* The random part of batch_sample (utils.py:604-612, 627-628) from caller-supplied uniforms
 * uniforms[3][b][num] in [0,1):  choices[b,num] = area-weighted face id (inverse CDF of the face areas
 * at uniforms[0], i.e. multinomial with replacement), u = sqrt(uniforms[1]), v = uniforms[2].
 * nf <= 16384 (the CDF lives in LDS), otherwise GEOM_EUNSUPPORTED. */
int geom_draw_samples_f32(int b, int nv, const float *verts, int nf, const int64_t *faces, int num,
                          const float *uniforms, int64_t *choices, float *u, float *v, void *stream);
/* Same draws from an in-kernel counter-based generator (Philox4x32-10): rng_state = 4 device uint64
 * {seed, stream position, 0, first global mesh index}; the last workgroup of every call advances the position on
 * the device (the third word is its arrival counter and is 0 between calls), so a captured HIP graph draws fresh
 * numbers on each replay and no generator bookkeeping is launched.  A sample's uniforms are a function of (seed,
 * position, first global mesh index + mesh, sample): data-parallel shards that share a seed reproduce the draws of
 * one process holding the whole batch.  One HIP stream per rng_state (the position/arrival words are not
 * safe under concurrent calls).  points (may be NULL):
 * [b,num,3], the sampled points themselves -- what geom_sample_faces_fwd_f32 would compute from the draws. */
int geom_draw_samples_rng_f32(int b, int nv, const float *verts, int nf, const int64_t *faces, int num,
                              uint64_t *rng_state, int64_t *choices, float *u, float *v, float *points,
                              void *stream);
/* points[b,num,3] = (1-u)*x + (u*(1-v))*y + (u*v)*z with x,y,z the corners of face
 * choices[b,num] (int64 face ids), u already sqrt'ed (utils.py:615-631). */
int geom_sample_faces_fwd_f32(int b, int nv, const float *verts, int nf, const int64_t *faces,
                              int num, const int64_t *choices, const float *u, const float *v,
                              float *points, void *stream);
/* grad_verts[b,nv,3] += barycentric weights * grad_points[b,num,3]; grad_verts must be
 * zero-initialised (or hold a running sum) by the caller; fp32 atomics. */
int geom_sample_faces_bwd_f32(int b, int nv, int nf, const int64_t *faces,
                              int num, const int64_t *choices, const float *u, const float *v,
                              const float *grad_points, float *grad_verts, void *stream);

/* ---- loss stages (utils.py:393-502, 506-587) -------------------------------------------------
 * Forward of the Chamfer gather loss needs no kernel of its own: sum_j |dst[idx[j]] - src[j]|^2
 * (utils.py:416-417, 462) is the sum of the squared distances geom_chamfer_nn_f32 already wrote.
 *
 * Gradient of  coef * sum_j |dst[b, idx[b,j]] - src[b,j]|^2  with coef = coef_host * coef_dev[0]
 * (coef_dev may be NULL = 1; it is the upstream autograd scalar, kept on the device so no
 * host sync is needed):
 *   grad_src[b,j]        = (or +=, when accumulate != 0)  2*coef*(src - dst[idx])   (may be NULL)
 *   grad_dst[b,idx[b,j]] -= the same, with fp32 atomics; caller zero-initialises      (may be NULL) */
int geom_chamfer_grad_f32(int b, int n, const float *src, int m, const float *dst,
                          const int *idx, const float *coef_dev, float coef_host,
                          float *grad_src, int accumulate, float *grad_dst, void *stream);

/* Point-to-surface loss for the winning triangle of each point (calc_point_to_line,
 * utils.py:506-550 with edge/Plane utils.py:553-587): q = the candidate selected by option[b,j]
 * on triangle index[b,j] of (verts [b,nv,3], faces [nf,3] int64).
 *   sqdist[b,n]    = |q - p|^2
 *   closest[b,n,3] = q                     (optional, needed by the backward)
 *   weights[b,n,3] = affine weights of q on the corners A,B,C (optional, needed by the backward) */
int geom_p2tri_loss_fwd_f32(int b, int n, const float *xyz, int nv, const float *verts,
                            int nf, const int64_t *faces, const int *option, const int *index,
                            float *sqdist, float *closest, float *weights, void *stream);
/* grad_verts[b, faces[index,k]] += 2*coef * w_k * (q - p)  -- the closed form of the reference's
 * autograd through calc_point_to_line (q minimises over every free parameter, so the
 * parameter derivatives vanish).  fp32 atomics; caller zero-initialises grad_verts. */
int geom_p2tri_loss_bwd_f32(int b, int n, const float *xyz, int nv, int nf, const int64_t *faces,
                            const int *index, const float *closest, const float *weights,
                            const float *coef_dev, float coef_host, float *grad_verts, void *stream);

/* out[0] = scale * sum(x[0..n)) with a fixed reduction tree (bit-reproducible run to run). */
int geom_sum_f32(int64_t n, const float *x, float scale, float *out, void *stream);
/* out[0] = scale1*sum(x1) + scale2*sum(x2): the whole (dist_1 + dist_2) * 3000 of utils.py:420/484.
 * clear/clear_count (may be NULL/0): floats the same launch sets to zero -- the buffer the backward scatters
 * into, so that it needs no fill launch of its own (one workgroup: meant for a few hundred KB). */
int geom_sum2_f32(int64_t n1, const float *x1, float scale1, int64_t n2, const float *x2, float scale2,
                  float *out, float *clear, int64_t clear_count, void *stream);

/* Fused backward of the Chamfer term THROUGH the sampling (utils.py:454-462 + 615-631 under autograd):
 * the gradient 2*coef*(point - other) of a sampled point is scattered straight into grad_verts with
 * the point's barycentric weights (fp32 atomics, caller zero-initialises), never materialised.
 *   points [b,num,3] sampled points with their draws (choices,u,v) on faces; other [b,n_other,3].
 *   via_nn == 0: one thread per sampled point t,  pair (t, idx[b,t])      idx [b,num]      -> into other
 *   via_nn == 1: one thread per other point t,    pair (idx[b,t], t)      idx [b,n_other]  -> into points */
int geom_sample_chamfer_bwd_f32(int b, int nv, int nf, const int64_t *faces, int num,
                                const int64_t *choices, const float *u, const float *v,
                                const float *points, int n_other, const float *other,
                                const int *idx, int via_nn, const float *coef_dev, float coef_host,
                                float *grad_verts, void *stream);
/* Backward of batch_point_to_surface (utils.py:441-502) in one launch: the Chamfer term of the sampled points
 * (idx_g [b,num] = nearest gt point; coefficient coef_sample) and the point-to-triangle term of the gt points
 * (index/closest/weights from geom_p2tri_loss_fwd_f32; coefficient coef_tri), both scattered into the same
 * ZEROED grad_verts [b,nv,3]; coef_dev (device scalar, may be NULL) multiplies both. */
int geom_surface_loss_bwd_f32(int b, int nv, int nf, const int64_t *faces, int num, const int64_t *choices,
                              const float *u, const float *v, const float *points, int n_gt, const float *gt,
                              const int *idx_g, const int *index, const float *closest, const float *weights,
                              const float *coef_dev, float coef_sample, float coef_tri, float *grad_verts,
                              void *stream);
/* The same backward (and that of batch_point_to_point) as a GATHER: no float atomics, no zero-fill of grad_verts,
 * bit-reproducible.  vf_ptr [nv+1] / vf_item [3*nf] = static CSR vertex -> incident (face << 2 | corner), ascending
 * per vertex.  Exactly one of idx_p ([b,n_gt] nearest sampled point of each gt point: the two-sided Chamfer term)
 * and index (+ closest, weights: the point-to-triangle term) may be given, or neither (sampled-point term only).
 * counts: geom_surface_bin_count_words(b, nf) int32, MUST BE ZERO on entry and is left zero again;
 * lists: geom_surface_bin_list_words(b, nf, num, n_gt) int32 scratch, 16-byte aligned (face offsets, per-point slot /
 * face / ordered id lists and two float4 records per point: the points are counting-sorted by face and every
 * face's points summed in ascending id order -- exact for any distribution of points over faces).  Returns
 * GEOM_EUNSUPPORTED when nf + num + n_gt exceeds what one workgroup can order in LDS (~38 000): use the scatter
 * entry points then.  Every element of grad_verts [b,nv,3] is written. */
int64_t geom_surface_bin_count_words(int b, int nf);
int64_t geom_surface_bin_list_words(int b, int nf, int num, int n_gt);
int geom_surface_loss_bwd_gather_f32(int b, int nv, int nf, const int *vf_ptr, const int *vf_item, int num,
                                     const int64_t *choices, const float *u, const float *v, const float *points,
                                     int n_gt, const float *gt, const int *idx_g, const int *idx_p, const int *index,
                                     const float *closest, const float *weights, const float *coef_dev,
                                     float coef_sample, float coef_other, int *counts, int *lists, float *grad_verts,
                                     void *stream);

/* ---- 0N-GCN aggregation (layers.py:34-41, 107-116, 143-152) -----------------------------------
 * out[r,:k] = sum_j val[j]*support[col[j],:k] over CSR row r (rowptr int32 [nv+1], col int32, val f32),
 * out[r,k:] = support[r,k:], then + bias[c] when bias != NULL.  support/out are [b,nv,c] row-major.
 * act: 0 none, 1 relu, 2 elu(alpha=1) fused into the epilogue. */
int geom_zn_gcn_aggregate_fwd_f32(int b, int nv, int c, int k, const int *rowptr, const int *col,
                                  const float *val, const float *support, const float *bias,
                                  int act, float *out, void *stream);
/* grad_support[r,:k] = sum over CSR^T row r of valT*g[colT,:k]; grad_support[r,k:] = g[r,k:], where
 * g = grad_out * act'(out) when act != 0 (out = the saved forward output, may be NULL when act == 0).
 * grad_bias[c] (optional) = column sums of g, reduced in a fixed order (bit-reproducible); it needs
 * `scratch` of at least geom_zn_gcn_bwd_scratch_floats(b, nv, c) floats.
 * scratch != NULL with grad_bias == NULL: the launch leaves its per-workgroup partial column sums in scratch --
 * geom_zn_gcn_bwd_partial_rows(...) rows of c floats -- and the caller finishes them later with geom_colsum_batch_f32
 * (same fixed-order reduction, same bits), ONE launch for the pending bias gradients of a whole backward pass
 * instead of one per layer. */
int64_t geom_zn_gcn_bwd_scratch_floats(int b, int nv, int c);
/* ell_w = the table width passed to geom_zn_gcn_aggregate_ell_bwd_f32 (8 / 16); 0 = geom_zn_gcn_aggregate_bwd_f32. */
int64_t geom_zn_gcn_bwd_partial_rows(int b, int nv, int c, int k, int ell_w);
/* outs[i][0..cols[i]) = column sums of partials[i] (rows[i] x cols[i], row-major), i < count <= GEOM_COLSUM_MAX_JOBS;
 * the pointer / size arrays are HOST arrays. */
#define GEOM_COLSUM_MAX_JOBS 32
int geom_colsum_batch_f32(int count, const float *const *partials, const int *rows, const int *cols,
                          float *const *outs, void *stream);
int geom_zn_gcn_aggregate_bwd_f32(int b, int nv, int c, int k, const int *rowptrT, const int *colT,
                                  const float *valT, const float *grad_out, const float *out,
                                  int act, float *grad_support, float *grad_bias, float *scratch,
                                  void *stream);

/* ELL fast path of the same two operations (identical results: neighbours are summed in the same
 * order).  ell_col/ell_val are [nv][w] row-major, w = 8 or 16, unused slots col = -1; supported when
 * k % 4 == 0 and c == 3k (split 3) or c == 10k (split 10) -- every hidden layer of the reference
 * models on a triangle mesh; anything else returns GEOM_EUNSUPPORTED and the CSR entry points above
 * apply.  Rows longer than w (the 33-entry poles of the reference's 482.obj, GEOMetrics.py:44) keep their first w
 * entries in the table and continue in a CSR tail: over_ptr int32 [nv+1], over_col / over_val = entries w.. of
 * every row in row order (all three NULL when no row is longer than w).  All pointers 16-byte aligned.  Scratch as
 * for the CSR backward.
 * relu_mask (may be NULL; act == ReLU and c == 3k only, else GEOM_EINVAL): geom_zn_gcn_relu_mask_words(b,nv,c,k)
 * uint16 words.  The forward stores the sign of every output element in it (one bit each); a backward that is
 * given the mask takes relu' from it and does not read `out` (which may then be NULL): 1/3 less traffic. */
int64_t geom_zn_gcn_relu_mask_words(int b, int nv, int c, int k);
int geom_zn_gcn_aggregate_ell_fwd_f32(int b, int nv, int c, int k, int w, const int *ell_col,
                                      const float *ell_val, const int *over_ptr, const int *over_col,
                                      const float *over_val, const float *support, const float *bias,
                                      int act, float *out, uint16_t *relu_mask, void *stream);
int geom_zn_gcn_aggregate_ell_bwd_f32(int b, int nv, int c, int k, int w, const int *ell_colT,
                                      const float *ell_valT, const int *over_ptrT, const int *over_colT,
                                      const float *over_valT, const float *grad_out, const float *out,
                                      const uint16_t *relu_mask, int act, float *grad_support,
                                      float *grad_bias, float *scratch, void *stream);

/* ---- dense products of a 0N-GCN layer on the fp32 matrix cores (layers.py:30, 107, 140: `support = input . W`) -------
 * Exact fp32 (v_mfma_f32_16x16x4_f32: a k-ordered fma chain per output element, no reduced precision).  Row-major
 * operands as the reference holds them: x [rows, cin] (rows = b*V), w [cin, c], c <= 192 and c % 16 == 0 for the forward,
 * anything else GEOM_EUNSUPPORTED (callers then use the library product).  x may be only 4-byte aligned per row (cin = 963).
 *
 * geom_dense_fwd_f32: support = x . w.
 *   ksplit == 0: out [rows, c] = support.
 *   ksplit  > 0 (% 16 == 0, < c; layers.py:108-116 with k = ksplit, activation ReLU): the aggregated columns go RAW to
 *     sup [rows, ksplit] (compact), the pass-through columns are finished: out[r, j] = relu(support[r, j] + bias[j]) for
 *     j >= ksplit (bias may be NULL), and mask (may be NULL) [rows, c/16] uint16 receives their sign bits (bit j % 16 of word
 *     (r, j / 16) = out[r, j] > 0; the words of the aggregated columns are left for the aggregation kernel).
 * geom_dense_bwd_input_f32: grad_x [rows, cin] = g [rows, c] . w^T.
 * geom_dense_bwd_weight_f32 (c % 12 == 0, else GEOM_EUNSUPPORTED): per-split partial tiles of grad_w = x^T . g (and, want_colsum != 0, of the column sums of g =
 *   the bias gradient) into `workspace` (geom_dense_bwd_weight_workspace_floats floats, 16-byte aligned);
 * geom_dense_reduce_f32 adds them up in a fixed order -- ONE launch for the pending weight / bias gradients of up to
 *   GEOM_DENSE_MAX_LAYERS layers (host arrays of per-layer sizes and pointers; grad_bias or its entries may be NULL). */
#define GEOM_DENSE_MAX_LAYERS 8
#define GEOM_DENSE_MAX_REDUCE_JOBS 32
int geom_dense_fwd_f32(int rows, int cin, int c, const float *x, const float *w, int ksplit, const float *bias,
                       float *out, float *sup, uint16_t *mask, void *stream);
int geom_dense_bwd_input_f32(int rows, int cin, int c, const float *g, const float *w, float *grad_x, void *stream);
int64_t geom_dense_bwd_weight_workspace_floats(int rows, int cin, int c);
int geom_dense_bwd_weight_f32(int rows, int cin, int c, const float *x, const float *g, float *workspace,
                              int want_colsum, void *stream);
/* both gradients of one layer; one launch (two workgroups per CU: one on grad_x, one on the grad_w partials) when
 * cin <= 192 and x rows are 16-byte aligned, the two launches above otherwise.  workspace / reduction as above. */
int geom_dense_bwd_f32(int rows, int cin, int c, const float *x, const float *g, const float *w, float *grad_x,
                       float *workspace, int want_colsum, void *stream);
/* geom_dense_reduce_f32 + `ncs` column-sum jobs (cs_outs[i] = column sums of the cs_rows[i] x cs_cols[i] row-major cs_partials[i],
 * cs_cols % 4 == 0) in the same launch: the bias gradients of the aggregation backward join the weight gradients. */
int geom_dense_reduce2_f32(int count, const int *rows, const int *cin, const int *c, const float *const *workspaces,
                           float *const *grad_w, float *const *grad_bias, int ncs, const float *const *cs_partials,
                           const int *cs_rows, const int *cs_cols, float *const *cs_outs, void *stream);
/* geom_dense_reduce2_f32 + the Adam step (torch.optim.Adam's rule, as geom_adam_step_f32) of the parameters whose gradients
 * the launch finishes: w_p / w_m / w_v[l] = parameter and its two moments for layer l's weight, b_p / b_m / b_v[i] for column-sum
 * job i (entries may be NULL: gradient only); `state` = the optimiser's device-side step state, advanced once by this launch;
 * grad_scale = 1.  Single-process steps only: a data-parallel step reduces across ranks first. */
int geom_dense_reduce_adam_f32(int count, const int *rows, const int *cin, const int *c, const float *const *workspaces,
                               float *const *grad_w, float *const *w_p, float *const *w_m, float *const *w_v, int ncs,
                               const float *const *cs_partials, const int *cs_rows, const int *cs_cols, float *const *cs_outs,
                               float *const *b_p, float *const *b_m, float *const *b_v, float lr, float beta1, float beta2,
                               float eps, float *state, void *stream);
int geom_dense_reduce_f32(int count, const int *rows, const int *cin, const int *c, const float *const *workspaces,
                          float *const *grad_w, float *const *grad_bias, void *stream);

/* HEAD mode of the ELL fast path: the layer whose three leading output channels are the coordinate update of a deformation
 * stage (GEOMetrics.py:121,126,131: positions + block output[..., :3]).  Split 3 with ReLU + relu_mask or act == 0 only
 * (else GEOM_EUNSUPPORTED: use the plain entry points + geom_vertex_head_*).
 *   forward : as geom_zn_gcn_aggregate_ell_fwd_f32, and pos[r,:] = base[r,:] + scale * out[r,:3]  (base / pos [b*nv, 3]);
 *   backward: the upstream gradient is [scale * grad_pos | 0 ...] by construction, so it is never materialised or read:
 *             grad_support / grad_bias exactly as geom_zn_gcn_aggregate_ell_bwd_f32 would give for that grad_out. */
int geom_zn_gcn_aggregate_ell_head_fwd_f32(int b, int nv, int c, int k, int w, const int *ell_col,
                                           const float *ell_val, const int *over_ptr, const int *over_col,
                                           const float *over_val, const float *support, const float *bias,
                                           int act, float *out, uint16_t *relu_mask, const float *base,
                                           float scale, float *pos, void *stream);
int geom_zn_gcn_aggregate_ell_head_bwd_f32(int b, int nv, int c, int k, int w, const int *ell_colT,
                                           const float *ell_valT, const int *over_ptrT, const int *over_colT,
                                           const float *over_valT, const float *grad_pos, float scale,
                                           const uint16_t *relu_mask, int act, float *grad_support,
                                           float *grad_bias, float *scratch, void *stream);

/* ---- a whole layer boundary in ONE launch (csrc/zn_stack.hip; layers.py:107-116 across two consecutive layers) --------
 * The aggregation of one layer is fused into the operand load of the product that follows it (forward) / precedes it
 * (backward); the weight slice of every wave stays in registers, the gathered operand goes through a 13 KB LDS panel.
 * Shapes: c == 192, k == 64 (split 3), ell_w == 8 with NO row longer than the table, n_out / n_in <= 192 and % 12 == 0;
 * anything else GEOM_EUNSUPPORTED (callers then use geom_zn_gcn_aggregate_ell_* + a product).
 *
 * geom_zn_layer_fwd_f32:  x_out = act([A . s_prev[:, :k] | s_prev[:, k:]] + bias_prev)   [b*nv, 192]  (the bits of
 *   geom_zn_gcn_aggregate_ell_fwd_f32; relu_mask in that entry point's layout, optional, act == 1 only)
 *                         s_out = x_out . w    (w [192, n_out] row-major, s_out [b*nv, n_out])
 *   wt_out (optional) [n_out, 192] receives w transposed -- what geom_zn_layer_bwd_f32 takes as `wt`.
 * geom_zn_layer_bwd_f32:  g_out = [A^T . g'[:, :k] | g'[:, k:]], g' = grad_out * act'(out)    [b*nv, 192]  (the bits of
 *   geom_zn_gcn_aggregate_ell_bwd_f32; act' from relu_mask (act 1) / `out` (act 2))
 *                         grad_in = g_out . wt   (wt [192, n_in] = the layer's weight transposed, grad_in [b*nv, n_in])
 *   colsum_partial (optional) [geom_zn_layer_partial_rows(b, nv)][192]: per-workgroup column sums of g' (bias-gradient
 *   partials; finished by geom_dense_reduce2_f32 / geom_colsum_batch_f32).  grad_pos != NULL: head mode -- grad_out is
 *   [head_scale * grad_pos | 0 ...] by construction and is not read (act 0 or 1 only). */
int64_t geom_zn_layer_partial_rows(int b, int nv);
int geom_zn_layer_fwd_f32(int b, int nv, int c, int k, int ell_w, const int *ell_col, const float *ell_val,
                          const float *s_prev, const float *bias_prev, int act, const float *w, int n_out,
                          float *x_out, uint16_t *relu_mask, float *s_out, float *wt_out, void *stream);
int geom_zn_layer_bwd_f32(int b, int nv, int c, int k, int ell_w, const int *ell_col_t, const float *ell_val_t,
                          const float *grad_out, const float *out, const uint16_t *relu_mask, int act,
                          const float *grad_pos, float head_scale, const float *wt, int n_in, float *g_out,
                          float *grad_in, float *colsum_partial, void *stream);

/* ---- a hidden layer of the mesh deformation block in ONE launch per direction (csrc/deform_block.hip) -------------------
 * reference: models.py:237-297 (BatchMeshDeformationBlock.forward: gcK -> F.relu(bnK(.)) with bnK = nn.BatchNorm1d(verts),
 * `features = features + x; features /= 2` after every second layer) on top of layers.py:107-116.  One workgroup per vertex
 * (its b <= 16 batch rows are one 16-row tile of the fp32 matrix core; BatchNorm1d(verts) statistics are tile-local).
 * Shapes: c == 192, k == 64 (split 3), ell_w == 8 (rows longer than the table continue in the tail table: tail_col / tail_val
 * [nv, GEOM_DEFORM_TAIL], entries 8, 9, ... of the row in CSR order, col -1 = padding; NULL: no row is longer), b <= 16;
 * anything else GEOM_EUNSUPPORTED (callers then run the separate operators: product, geom_zn_gcn_aggregate_ell_*,
 * geom_vertex_bn_*).  All arrays [b, nv, 192] row-major fp32, 16-byte aligned, caller-allocated.
 *
 * geom_deform_layer_fwd_f32:
 *     z_out = [A . s_in[:, :k] | s_in[:, k:]] + bias                 (the bits of geom_zn_gcn_aggregate_ell_fwd_f32, act 0)
 *     x_out = relu?(BatchNorm_v(z_out)) ; with res: (res + .) * scale  (nn.BatchNorm1d(verts) semantics of geom_vertex_bn_fwd_f32:
 *             training: batch statistics, biased variance, running stats updated with the unbiased one; else running stats)
 *     s_out = x_out . w_next            (w_next: the next layer's weight in the register-slice order
 *                                        geom_deform_pack_weights_f32 writes as fwd[l]; NULL: no product -- the last hidden layer)
 * geom_deform_layer_bwd_f32:  (dz_up != NULL)
 *     ds_up = [A^T . dz_up[:, :k] | dz_up[:, k:]]                    (aggregation backward of the layer ABOVE; its weight
 *                                                                     gradient is x^T . ds_up)
 *     g     = ds_up . W_up^T  (+ g2 if given)                         (gradient of THIS layer's output x_out; wt_up = the packed
 *                                                                     bwd[l] copy of W_up from geom_deform_pack_weights_f32)
 *   (dz_up == NULL: g is read from memory instead, + g2, + the coordinate head's input gradient when ds_head is given)
 *     with has_res: g *= scale, grad_res = g (the residual's gradient);  relu: g masked where BatchNorm_v(z) <= 0
 *     dz = BatchNorm_v backward(g; z, save_mean, save_invstd, bn_w);  grad_bn_w[v] / grad_bn_b[v] = the vertex's sums
 *     colsum (optional) [nv,192]: the vertex's column sums of dz over its meshes (the layer's bias gradient = their sum over
 *     the vertices, in vertex order).
 * `vpx` is filled in by the library. */
typedef struct geom_deform_fwd {
    int b, nv, c, k, ell_w;
    const float *s_in, *bias;
    const int *ell_col; const float *ell_val;
    const int *tail_col; const float *tail_val;                 /* [nv, GEOM_DEFORM_TAIL] or NULL */
    const float *bn_w, *bn_b;                                   /* [nv] or NULL (1 / 0) */
    float *run_mean, *run_var;                                  /* [nv] */
    int training; float momentum, eps;
    int relu;
    const float *res; int res_ld; float scale;                  /* optional residual [b,nv,res_ld >= 192] */
    float *z_out, *x_out, *save_mean, *save_invstd;
    const float *w_next; float *s_out;
    const float *w_head; float *s_head;                         /* optional, only with w_next == NULL: s_head [b,nv,3] = x_out . w_head
                                                                 * (w_head [192,3] row-major: the block's coordinate head gc15) */
    int vpx;
} geom_deform_fwd;
typedef struct geom_deform_bwd {
    int b, nv, c, k, ell_w;
    const float *dz_up;
    const int *ell_col_t; const float *ell_val_t;
    const int *tail_col_t; const float *tail_val_t;
    float *ds_up; const float *wt_up;
    const float *g, *g2;
    int g_ld, g2_ld;            /* floats between two rows of g / g2 (0 = 192): column slices of wider buffers are read in place */
    const float *z, *bn_w, *bn_b, *save_mean, *save_invstd;
    int relu, has_res; float scale;
    float *grad_res, *dz, *grad_bn_w, *grad_bn_b, *colsum;
    const float *ds_head, *w_head, *x_top; float *dw_head;      /* optional, only with dz_up == NULL: g += ds_head [b,nv,3] . w_head^T
                                                                 * (g may then be NULL); dw_head [nv,192,3] = per-vertex partials of
                                                                 * x_top^T . ds_head (x_top [b,nv,192]: the layer's output) */
    int vpx;
} geom_deform_bwd;
/* The hidden layers of a block as ONE launch: `count` <= GEOM_DEFORM_CHAIN_MAX consecutive geom_deform_layer_fwd_f32 layers
 * (layers[l + 1].s_in == layers[l].s_out; only the last may lack a product), HOST array of their argument structs; results
 * bit for bit those of the separate calls.  A vertex's workgroup runs layer after layer and waits, in front of a layer's
 * gathers, for the workgroups of its neighbours to publish the previous layer's rows (the only cross-tile dependency of a
 * layer: models.py:237-297 + layers.py:107-116) -- twelve kernel boundaries less per block and direction.
 * done: nv * 32 ints of device scratch, 128-byte aligned, ZERO on entry (geom_deform_pack_weights_zero_f32 clears it in
 * the step's packing launch).  Needs every workgroup of the launch resident at once: geom_deform_chain_fits(nv) != 0,
 * otherwise GEOM_EUNSUPPORTED (issue the layers one by one).  A wait that gives up (seconds) turns the layer's outputs
 * into NaN. */
#define GEOM_DEFORM_CHAIN_MAX 13
int geom_deform_chain_fits(int nv);
int geom_deform_chain_fwd_f32(int count, const geom_deform_fwd *layers, int *done, void *stream);
/* ... and the backward layers, in execution order: layers[0] = the top layer (dz_up == NULL: its gradient comes from memory),
 * layers[t].dz_up == layers[t - 1].dz; every step its own dz array; `done` = its own nv * 32 zeroed ints.  ds_first (may be
 * NULL; needs count >= 2): [b,nv,192] = [A^T . dz[:, :k] | dz[:, k:]] of the LAST step's dz -- the aggregation backward of the
 * chain's first layer (geom_zn_gcn_aggregate_ell_bwd_f32 without activation, same bits) as one more step of the launch. */
int geom_deform_chain_bwd_f32(int count, const geom_deform_bwd *layers, int *done, float *ds_first, void *stream);
int geom_deform_pack_weights_zero_f32(int count, const float *const *w, float *fwd, float *bwd, int *zero, int zero_words,
                                      void *stream);
/* EXPERIMENT (csrc/dense_split_bf16.hip; on no default route): c [m, 192] = a [m, k] . w [k, 192] on the BF16 matrix cores with
 * exact fp32 products -- every fp32 operand is the exact sum of three bf16 numbers, the products a_i b_j are exact in fp32,
 * terms = 6 keeps those with i + j <= 2 (the rest is below 2^-24 of |a b|), 9 keeps all; fp32 accumulation, the leading
 * term and the small terms in separate accumulators.  w comes pre-split: planes [3][192][kpad] bf16, kpad =
 * geom_split_bf16_kpad(k), from geom_split_bf16_planes_f32 (once per weight update).  n != 192: GEOM_EUNSUPPORTED. */
int geom_split_bf16_kpad(int k);
int geom_split_bf16_planes_f32(int k, int n, const float *w, uint16_t *planes, void *stream);
int geom_gemm_split_bf16_f32(int m, int k, int n, const float *a, const uint16_t *planes, float *c, int terms, void *stream);

/* The regularisers of ONE deformation stage (GEOMetrics.py:147-161: edge term of the new positions + squared difference of
 * the Laplacian coordinates of the previous and the new positions + their squared displacement) in one launch per direction
 * (csrc/regularizers.hip).  forward: partial[geom_stage_regularisers_blocks(b, nv, nf)] = per-workgroup sums of
 *     c_edge * sum_faces(|e1|^2 + |e2|^2 + |e3|^2)(cur) + c_lap * sum_v |lap(prev - cur)|^2 + c_move * sum_v |prev - cur|^2
 * (finish with geom_sum_f32; fold the means' 1 / counts into the c_*), lapd [b,nv,3] = lap(prev - cur) for the backward.
 * prev [b,nv,3] (prev_batched != 0) or one [nv,3] mesh for the whole batch; (rowptr, col, inv_deg) as geom_laplacian_f32.
 * backward: grad_cur (and grad_prev, batched prev only) of gout[0] * that sum; (vf_ptr [nv+1], vf_item) = every vertex's
 * incident corners as face * 4 + corner, ascending: the edge gradient is a gather, no float atomics. */
int64_t geom_stage_regularisers_blocks(int b, int nv, int nf);
int geom_stage_regularisers_fwd_f32(int b, int nv, const float *prev, int prev_batched, const float *cur, int nf,
                                    const int64_t *faces, const int *rowptr, const int *col, const float *inv_deg,
                                    float c_lap, float c_move, float c_edge, float *lapd, float *partial, void *stream);
int geom_stage_regularisers_bwd_f32(int b, int nv, const float *prev, int prev_batched, const float *cur, int nf,
                                    const int64_t *faces, const int *rowptr, const int *col, const float *inv_deg,
                                    const int *vf_ptr, const int *vf_item, float c_lap, float c_move, float c_edge,
                                    const float *lapd, const float *gout, float *grad_prev, float *grad_cur, void *stream);

/* out = ((t[0] + t[1]) + t[2]) + ... element by element, count <= GEOM_SUM_MAX_TENSORS contiguous fp32 tensors of n elements (HOST
 * array of device pointers): the gradients that reach one tensor through several consumers, summed by ONE launch in a fixed order
 * instead of one accumulation launch per consumer (python: utils.fan_out). */
#define GEOM_SUM_MAX_TENSORS 8
int geom_sum_tensors_f32(int count, const float *const *tensors, int64_t n, float *out, void *stream);
/* ... for [rows, width] operands some of which are column slices of wider row-major buffers: lds[k] = floats between two rows
 * of tensors[k] (>= width; == width: contiguous); out contiguous.  (The gradient of a block's coordinate input is the
 * leading three columns of the 1155-wide input gradient: read in place.) */
int geom_sum_tensors_rows_f32(int count, const float *const *tensors, const int64_t *lds, int64_t rows, int width, float *out,
                              void *stream);

/* Camera of every image from param [b,3] = (azimuth deg, elevation deg, distance) (reference utils.py:286-313: ~35 tiny
 * torch launches per call, three calls per training step): cam_mat [b,3,3] (rows = the camera's normalised x, y, z axes),
 * cam_pos [b,3] -- the operands of geom_pool_features_*; the same fp32 expressions in the same order, one launch. */
int geom_camera_info_f32(int b, const float *param, float *cam_mat, float *cam_pos, void *stream);

#define GEOM_DEFORM_MAX_PACK 16
#define GEOM_DEFORM_TAIL 32
/* fwd[l] / bwd[l] ([count, 36864] floats each; either may be NULL): w[l] / w[l]^T ([192,192] row-major device matrices, `w` a
 * HOST array of count <= GEOM_DEFORM_MAX_PACK pointers) in the order a wave of the layer launches keeps its weight slice in
 * registers; one launch for all layers of a block, once per step (the weights change with every optimiser step). */
int geom_deform_pack_weights_f32(int count, const float *const *w, float *fwd, float *bwd, void *stream);
int geom_deform_layer_fwd_f32(const geom_deform_fwd *args, void *stream);
int geom_deform_layer_bwd_f32(const geom_deform_bwd *args, void *stream);

/* Any-shape product on the fp32 matrix cores (csrc/dense_any.hip): c [m, n] (row pitch ldc) = op(a) . op(b), exact fp32
 * (v_mfma_f32_16x16x4_f32), any sizes / pitches / 4-byte alignments -- `torch.mm(input, weight)` of the layers whose widths
 * the 192-column kernels above do not take (the mesh encoder's ZERON_GCN layers, reference layers.py:30, models.py:299-348)
 * and the two gradients autograd derives from it:
 *   a_km == 0: a is [m, k] (row pitch lda);  a_km != 0: a is [k, m] (the transposed operand of x^T . g)
 *   b_kn != 0: b is [k, n] (row pitch ldb);  b_kn == 0: b is [n, k] (the transposed operand of g . w^T)
 *   forward  x . w   : (rows, c, cin,  x, cin, 0,  w, c, 1)      input gradient  g . w^T : (rows, cin, c,  g, c, 0,  w, c, 0)
 *   weight gradient x^T . g : (cin, c, rows,  x, cin, 1,  g, c, 1)
 * A long sum against few output tiles is split over the summed index into partial tiles in `workspace`
 * (geom_gemm_workspace_floats(m, n, k) floats; NULL / too small: one pass) and added up in split order by a second launch:
 * bit-reproducible.  Operands and result must each stay below 2 GiB (GEOM_ETOOBIG). */
int64_t geom_gemm_workspace_floats(int m, int n, int k);
int geom_gemm_f32(int m, int n, int k, const float *a, int64_t lda, int a_km, const float *b, int64_t ldb, int b_kn,
                  float *c, int64_t ldc, float *workspace, int64_t workspace_floats, void *stream);

/* Coordinate update of a deformation stage (GEOMetrics.py:121,126,131) when the predicted offsets are the
 * three leading channels of a wider feature tensor: pos[r,:] = base[r,:] + scale*feat[r,:3] for `rows`
 * vertices (feat row length c), and its adjoint grad_feat[r,:] = [scale*grad_pos[r,:] | 0 ...] (c % 4 == 0). */
int geom_vertex_head_fwd_f32(int64_t rows, int c, const float *base, const float *feat, float scale,
                             float *pos, void *stream);
int geom_vertex_head_bwd_f32(int64_t rows, int c, const float *grad_pos, float scale, float *grad_feat,
                             void *stream);

/* ---- mesh regularisers (SURVEY 8f "next" row 1; utils.py:636-662, GEOMetrics.py:147-161) ----------------
 * Laplacian coordinates over the CSR of the BINARY adjacency with self loops (adj_info['adj_orig']):
 *   transpose == 0:  out[b,v] = x[b,v] - (sum_{j in row v} x[b,j] - x[b,v]) * inv_deg[v]     (batch_get_lap_info)
 *   transpose != 0:  out[b,v] = g[b,v] - (sum_{j in row v} g[b,j]*inv_deg[j] - g[b,v]*inv_deg[v])   (its adjoint)
 * x/out [b,nv,3]; inv_deg[v] = 1 / (row length - 1). */
int geom_laplacian_f32(int b, int nv, const int *rowptr, const int *col, const float *inv_deg,
                       const float *x, int transpose, float *out, void *stream);
/* per_face[b,f] = |p2-p1|^2 + |p3-p1|^2 + |p2-p3|^2 (batch_calc_edge = sum / (3*b*nf)); and the gradient of
 * coef * sum(per_face) scattered into grad_verts (fp32 atomics, caller zero-initialises). */
int geom_edge_sqlen_fwd_f32(int b, int nv, const float *verts, int nf, const int64_t *faces,
                            float *per_face, void *stream);
int geom_edge_sqlen_bwd_f32(int b, int nv, const float *verts, int nf, const int64_t *faces,
                            const float *coef_dev, float coef_host, float *grad_verts, void *stream);

/* ---- per-vertex BatchNorm + ReLU + residual average (SURVEY 8f "next" row 2; models.py:222-297) --------
 * nn.BatchNorm1d(verts) applied to x [b,nv,c]: one statistic per vertex over its b*c values, then
 *     out = (residual + act(bn(x))) * scale      (residual == NULL: out = act(bn(x)))
 * training != 0: batch statistics (biased variance), running stats updated in place with `momentum` and
 * the unbiased variance, save_mean/save_invstd [nv] written for the backward; training == 0: running stats.
 * residual may be a column slice of a wider tensor: row stride residual_ld >= c.  b*c <= 4096, else
 * GEOM_EUNSUPPORTED (callers then use the library ops). */
int geom_vertex_bn_fwd_f32(int b, int nv, int c, const float *x, const float *weight, const float *bias,
                           float *running_mean, float *running_var, int training, float momentum, float eps,
                           int relu, const float *residual, int residual_ld, float scale,
                           float *out, float *save_mean, float *save_invstd, void *stream);
/* grad_x, (optional) grad_residual [b,nv,c] = grad_out*scale, grad_weight / grad_bias [nv] (optional).
 * grad_out2 (may be NULL): a second upstream gradient of the same output -- in the deformation block every second
 * BatchNorm output feeds the next layer AND a later residual average (models.py:249-287) -- added to grad_out on the
 * fly (one fp32 add per element, as autograd's own accumulation would do in a separate pass). */
int geom_vertex_bn_bwd_f32(int b, int nv, int c, const float *x, const float *grad_out, const float *weight,
                           const float *bias, const float *save_mean, const float *save_invstd, int relu,
                           int has_residual, float scale, float *grad_x, float *grad_residual,
                           float *grad_weight, float *grad_bias, const float *grad_out2, void *stream);

/* ---- image-feature pooling (SURVEY 8f "next" row 3; utils.py:316-389 batched_pooling) -----------------
 * out[b,v,:] = concat over `levels` feature maps blocks[l] [b, channels[l], dims[l], dims[l]] (NCHW) of the
 * reference's bilinear pooling at the pixel vertex v projects to: p' = cam_mat[b] . (0.57*verts[b,v] -
 * cam_pos[b]), h = (-Y)/(-Z)*248 + 112, w = X/(-Z)*248 + 112, (xs, ys) = (h, w)/223, per map clamp(xs*dim,
 * 0, dim-1) with weights (ceil-x, x-floor) (utils.py:321-350).  blocks/channels/dims (and grad_blocks) are
 * HOST arrays of `levels` <= GEOM_POOL_MAX_LEVELS entries.  Backward: grad_blocks[l] (entries or the array
 * may be NULL) and grad_verts [b,nv,3] (may be NULL) are fully overwritten (nothing to zero-initialise).  The
 * map gradient is computed as a gather over texel -> (vertex, weight) lists built in `workspace` (device,
 * 16-byte aligned, >= geom_pool_features_bwd_workspace_bytes(...) bytes), which also holds the per-channel-chunk
 * partial sums of the vertex gradient (added up in chunk order by a finishing launch): required whenever a
 * gradient is requested (ABI 9; GEOM_EINVAL otherwise); dims[l] <= 64, else GEOM_EUNSUPPORTED. */
size_t geom_pool_features_bwd_workspace_bytes(int b, int nv, int levels, const int *dims);
#define GEOM_POOL_MAX_LEVELS 8
int geom_pool_features_fwd_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos,
                               int levels, const float *const *blocks, const int *channels, const int *dims,
                               float *out, void *stream);
int geom_pool_features_bwd_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos,
                               int levels, const float *const *blocks, const int *channels, const int *dims,
                               const float *grad_out, float *const *grad_blocks, float *grad_verts,
                               void *workspace, size_t workspace_bytes, void *stream);
/* The same two with a ROW PITCH: out_ld / grad_ld = floats between two vertex rows (0 = the pooled width, i.e. the calls
 * above).  Larger: the features are written into / their gradient is read out of a column slice of a wider row-major buffer --
 * the deformation block's concatenated input [positions | previous features | pooled] and the input gradient of its first
 * product -- so that neither `torch.cat` (utils.py / models.py:241) nor the slicing copies of its backward are launched. */
int geom_pool_features_fwd_ld_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos, int levels,
                                  const float *const *blocks, const int *channels, const int *dims, float *out, int64_t out_ld,
                                  void *stream);
/* ... with up to two tensors fronts[f] [b, nv, widths[f]] (contiguous) copied into columns [cols[f], cols[f] + widths[f]) of the
 * wide buffer buf (row pitch out_ld; out = buf + the pooled features' first column) by workgroups of the same launch: what the
 * caller concatenates in front of the pooled features (GEOMetrics.py:123,128: the previous features; models.py:241: the
 * coordinates) -- the narrow copies were launches of their own. */
int geom_pool_features_fwd_fronts_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos, int levels,
                                      const float *const *blocks, const int *channels, const int *dims, float *out,
                                      int64_t out_ld, int n_fronts, const float *const *fronts, const int *widths,
                                      const int *cols, float *buf, void *stream);
int geom_pool_features_bwd_ld_f32(int b, int nv, const float *verts, const float *cam_mat, const float *cam_pos, int levels,
                                  const float *const *blocks, const int *channels, const int *dims, const float *grad_out,
                                  int64_t grad_ld, float *const *grad_blocks, float *grad_verts, void *workspace,
                                  size_t workspace_bytes, void *stream);

/* ---- culled Chamfer scan inside the surface step (optional; NULL everywhere = the brute-force tiles) ------------------
 * The culled scan (geom_chamfer_nn_culled_f32) needs both clouds listed in a spatially coherent order, as an INDEX
 * (geom_nn_cull_index_floats(b, n) floats, 16-byte aligned: the run spheres, then the cloud in that order):
 *   gt cloud:      static per batch -- gt_order [b,n_gt] (any coherent order: a Morton or k-d order made by the data
 *                  loader) and gt_index written ONCE by geom_nn_cull_index_f32(b, n_gt, gt, gt_order, gt_index);
 *   sampled cloud: new every step -- given the struct, geom_surface_prepare_f32 GENERATES the samples in the order of their
 *                  faces' positions in tri_order (sorted uniforms from exponential spacings: no sorting pass; the multiset
 *                  of samples has the law of independent draws, only their order is no longer random -- which is why the
 *                  stand-alone draw entry points never do this) and writes sample_index; *prepared gets bit 1 (value 2)
 *                  when it did (num between 64 and 4095).  geom_surface_scan_f32 given the same struct then runs the
 *                  culled tiles: the same outputs as the brute-force tiles for the same samples, bit for bit. */
typedef struct geom_surface_cull {
    const int *gt_order;     /* [b,n_gt] visiting position -> gt point (NULL: the cloud's own order) */
    const float *gt_index;   /* geom_nn_cull_index_floats(b, n_gt) floats */
    float *sample_index;     /* geom_nn_cull_index_floats(b, num) floats of scratch, written by the prepare call */
    const int *faces_in_order; /* optional [nf,3] int32: the corners of face tri_order[j] at position j (static per face list
                                * and order): saves the prepare launch a dependent load in each of its two gather chains */
} geom_surface_cull;
int64_t geom_nn_cull_index_floats(int b, int n);
int geom_nn_cull_index_f32(int b, int n, const float *xyz, const int *order, float *index, void *stream);

/* ---- per-step preparation of the surface loss in one launch ------------------------------------------------------
 * Everything that depends only on the vertex positions: the random face draws (+ sampled points) of batch_sample,
 * exactly geom_draw_samples_rng_f32, AND -- when the scan of the same step will take the fused route (coherent
 * tri_order, no truncation / brute-force flag, >= 256 query tiles of n_gt points) -- the triangle records of
 * geom_surface_scan_f32 in `workspace` (geom_tri_distance_workspace_bytes(b, n_gt, nf)).  *prepared = 1 then: pass
 * GEOM_FLAG_TRI_WS_READY and the same workspace to the scan.  Otherwise only the draws are made (*prepared = 0). */
int geom_surface_prepare_f32(int b, int nv, const float *verts, int nf, const int64_t *faces, int num,
                             uint64_t *rng_state, int64_t *choices, float *u, float *v, float *points, int n_gt,
                             const int *tri_order, unsigned flags, void *workspace, size_t workspace_bytes,
                             int *prepared, const geom_surface_cull *cull, void *stream);

/* ---- the two arg-min scans of the surface loss in one call (utils.py:451 + 470) ----------------------------------
 * gt [b,n_gt,3] against the sampled points [b,num,3]: nearest neighbours both ways, exactly geom_chamfer_nn_f32(gt,
 * points) -> (sq_gt, idx_p) for the gt points, (sq_pred, idx_g) for the sampled points; and, when verts != NULL, gt
 * against the mesh, exactly geom_tri_surface_fwd_f32 -> tri_dist / option / index + sq / closest / weights.  With a
 * coherent tri_order, default tri flags and at least 256 query tiles both scans run as ONE heterogeneous launch
 * (workgroups [0, tri tiles) scan triangles, the rest scan points) behind the triangle-record prep; otherwise as the
 * separate launches -- same results either way.  flags: GEOM_FLAG_FIX_REGION6 / _TRI_BRUTE_FORCE / _REF_TAIL_TRUNC /
 * _NN_FMA as for the separate entry points.  workspace: as geom_tri_distance_workspace_bytes(b, n_gt, nf).
 * order_scratch (may be NULL): the `order` buffer of geom_surface_finalize_f32; the scans then also write every
 * point's gradient record into it (u, v [b,num]: the draws of the sampled points; coef_sample / coef_other as for the
 * finalize) and *records_written = 1; the variants that cannot (split query tiles, brute force, tail truncation)
 * leave *records_written = 0 and the finalize pass forms the records itself.
 * tail (may be NULL): also run geom_surface_finalize_f32's work -- loss = scale_sample * sum(sq_pred) + scale_other *
 * sum(sq) into *loss and, with want_order, the points ordered by face in order_scratch -- as extra (role) workgroups of the
 * fused launch (each mesh is ordered as soon as ITS triangle tiles are through, the loss is summed behind the last tile)
 * instead of a launch of its own: same outputs, bit for bit.  tail->finalized = 1 when the launch did it (fused route
 * with the culled Chamfer tiles, nf + num + n_gt <= ~11 700 per mesh, at most (CUs / 4) meshes); 0: call
 * geom_surface_finalize_f32 as usual.
 * FAIL-SAFE: the roles wait inside the launch for tiles of the same launch (HIP promises no forward progress between
 * workgroups; the host admits the tail only where every workgroup of the launch is resident).  A role that waits in vain
 * gives up after ~2 s: *loss comes out as NaN, the role reads NOTHING of the incomplete results, and its status word at the
 * end of order_scratch (int32 [b + 1] at word geom_surface_order_words(b, nf, num, n_gt) - ((b + 4) & ~3): 0 = complete)
 * is set -- geom_surface_gather_f32 then writes NaN gradients for that mesh (for every mesh when the loss role gave up)
 * instead of walking a half-built order.  After such a launch the completion counters in `workspace` are undefined: run
 * geom_surface_prepare_f32 (which zeroes them) or zero them before the next tail launch on the same workspace. */
size_t geom_surface_tail_counters_offset(int b, int n_gt, int nf); /* byte offset of the tail's completion counters in the
                                                                     * scan workspace (tests of the give-up path); 0: none */
typedef struct geom_surface_tail {
    const int64_t *choices;         /* [b,num] faces of the sampled points */
    float scale_sample, scale_other;
    int want_order;                 /* needs order_scratch */
    float *loss;                    /* one element */
    int finalized;                  /* out */
} geom_surface_tail;
int geom_surface_scan_f32(int b, int n_gt, const float *gt, int num, const float *points, float *sq_gt, int *idx_p,
                          float *sq_pred, int *idx_g, int nv, const float *verts, int nf, const int64_t *faces,
                          const int *tri_order, float *tri_dist, int *option, int *index, float *sq, float *closest,
                          float *weights, const float *u, const float *v, float coef_sample, float coef_other,
                          int *order_scratch, unsigned flags, void *workspace, size_t workspace_bytes,
                          int *records_written, const geom_surface_cull *cull, geom_surface_tail *tail, void *stream);

/* ---- surface loss: forward-side finalize + single-launch backward ------------------------------------------
 * geom_surface_finalize_f32 runs once after the two scans of batch_point_to_surface / batch_point_to_point
 * (utils.py:393-502).  It (a) writes the scalar loss = scale_sample * sum(sq_sample[b,num]) + scale_other *
 * sum(sq_other[b,n_gt]) (one workgroup, fixed tree: bit-reproducible) and, when want_order,
 * (b) prepares the backward in `order` (geom_surface_order_words(b, nf, num, n_gt) int32 words, 16-byte aligned,
 * uninitialised scratch that must stay alive until the backward).  Every sampled point (face choices[b,num], draws u / v,
 * partner gt[idx_g]) and every gt point -- partner of the sampled point idx_p[b,n_gt] (two-sided Chamfer) or of the
 * closest point on triangle index[b,n_gt] with corner weights (point-to-surface); give one of idx_p / index or neither
 * -- gets a gradient record (point - partner) * coef_{sample,other} + corner weights, and the points are
 * counting-sorted by face with ascending ids inside a face.  One workgroup per mesh, counts / offsets / ids in LDS:
 * GEOM_EUNSUPPORTED when nf + num + n_gt exceeds ~38 000 (call again with want_order = 0 and use the scatter
 * backward).  geom_surface_gather_f32 is then the whole backward: grad_verts[b,nv,3] = 2 * grad[0] * sum over the
 * points on the vertex's incident faces (vf_ptr / vf_item: static CSR vertex -> (face << 2 | corner)), every element
 * written once, no float atomics, bit-reproducible.  has_other = whether idx_p or index was given to the finalize.
 * records_ready != 0: the gradient records were already written into `order` by geom_surface_scan_f32. */
int64_t geom_surface_order_words(int b, int nf, int num, int n_gt);
int geom_surface_finalize_f32(int b, int nf, int num, const int64_t *choices, const float *u, const float *v,
                              const float *points, int n_gt, const float *gt, const int *idx_g, const int *idx_p,
                              const int *index, const float *closest, const float *weights, const float *sq_sample,
                              const float *sq_other, float scale_sample, float scale_other, float coef_sample,
                              float coef_other, int want_order, int records_ready, int *order, float *loss,
                              void *stream);
int geom_surface_gather_f32(int b, int nv, int nf, const int *vf_ptr, const int *vf_item, int num, int n_gt,
                            int has_other, const int *order, const float *grad, float *grad_verts, void *stream);
/* The same two calls with PER-MESH factors mesh_weight[b] (device floats; NULL = the calls above): loss = sum_m w[m] *
 * (scale_sample * sum(sq_sample[m]) + scale_other * sum(sq_other[m])), mesh m's gradient = w[m] * 2 * grad[0] * (...).
 * The stages of the reference's cascade (GEOMetrics.py:134-138: three surface losses on meshes of one size against one
 * ground truth, weights .2 / .2 / 2) stacked into ONE batch: one draw / scan / finalize / gather launch instead of
 * three each.  The gradient records do not carry the weights (both calls must be given the same array). */
int geom_surface_finalize_w_f32(int b, int nf, int num, const int64_t *choices, const float *u, const float *v,
                                const float *points, int n_gt, const float *gt, const int *idx_g, const int *idx_p,
                                const int *index, const float *closest, const float *weights, const float *sq_sample,
                                const float *sq_other, float scale_sample, float scale_other, float coef_sample,
                                float coef_other, int want_order, int records_ready, int *order, float *loss,
                                const float *mesh_weight, void *stream);
int geom_surface_gather_w_f32(int b, int nv, int nf, const int *vf_ptr, const int *vf_item, int num, int n_gt,
                              int has_other, const int *order, const float *grad, const float *mesh_weight,
                              float *grad_verts, void *stream);

/* ---- ragged mesh batches (SURVEY 8f "next" row 4; auto_encoder.py:71-76, layers.py:78) ---------------
 * Meshes of different sizes are concatenated along the vertex axis; segment s owns rows
 * [offsets[s], offsets[s+1]) of x [offsets[nseg], c] (offsets: nseg+1 device int64).  out[s,col] =
 * max over the segment's rows (GCNMax's torch.max(..., dim=0)), arg[s,col] = the winning row RELATIVE to
 * the segment start (lowest row on ties, first NaN wins; -inf / -1 for an empty segment).  max_len = the
 * longest segment (host-known; sizes the row split).  The backward writes every element of grad_x
 * [total_rows, c] once: grad_out[s,col] at the arg-max row, 0 elsewhere. */
int64_t geom_segment_max_workspace_bytes(int nseg, int c, int64_t max_len);
int geom_segment_max_fwd_f32(int nseg, const int64_t *offsets, int64_t max_len, int c, const float *x,
                             float *out, int *arg, void *workspace, int64_t workspace_bytes, void *stream);
int geom_segment_max_bwd_f32(int nseg, const int64_t *offsets, int64_t total_rows, int c,
                             const float *grad_out, const int *arg, float *grad_x, void *stream);

/* ---- optimiser step for the replicated layer parameters (GEOMetrics.py:73: Adam, lr 1e-4) -----------
 * torch.optim.Adam's update (no weight decay / amsgrad) for up to GEOM_ADAM_MAX_TENSORS tensors in ONE
 * launch.  params/grads/exp_avg/exp_avg_sq/sizes are HOST arrays of `count` device pointers / lengths;
 * grads are multiplied by grad_scale first (1/world after a SUM all-reduce).  `state` is
 * GEOM_ADAM_STATE_WORDS 4-byte device words, zero-initialised by the caller once: {t, beta1^t, beta2^t}
 * as floats followed by the arrival counters of the in-kernel step advance (root at word 3, leaf l at word 32 * (1 + l): one
 * cache line each).  Every call applies the bias
 * corrections of step t+1; with advance != 0 the last workgroup to finish moves the state to t+1 (so a
 * captured HIP graph replays the correct correction, with no separate tick launch).  An optimiser holding
 * more than GEOM_ADAM_MAX_TENSORS tensors issues several calls on one stream for one step: advance = 0 on
 * all but the last. */
#define GEOM_ADAM_MAX_TENSORS 64
#define GEOM_ADAM_STATE_WORDS 2112 /* {t, b1^t, b2^t}, the root counter, 64 leaf counters one 128-byte line apart */
int geom_adam_step_f32(int count, float *const *params, const float *const *grads, float *const *exp_avg,
                       float *const *exp_avg_sq, const int64_t *sizes, float lr, float beta1, float beta2,
                       float eps, float grad_scale, float *state, int advance, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOM_HIP_H */
