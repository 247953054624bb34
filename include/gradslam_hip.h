/*
 * gradslam_hip.h — C-ABI of the MI355X (gfx950) dense-SLAM hot path.
 *
 * One shared library (gradslam_amd/csrc/libgradslam_hip.so) exports exactly these entry
 * points.  They replace the *bodies* of the reference's Python functions listed beside
 * each declaration (file:line are relative to the gradslam reference checkout).  The
 * reference has no FFI layer (it is pure Python on PyTorch); INTEGRATION.md shows the
 * ctypes stub a gradslam maintainer would add at each call site.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - plain pointers + explicit sizes, no torch types, no C++ types, no exceptions;
 *   - return value: GS_OK (0) or a GS_ERR_* code; gs_last_error() gives a message;
 *   - nothing here allocates: the caller (torch) owns every buffer, scratch included;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), no host sync
 *     unless the function name ends in _sync;
 *   - one call handles ONE sequence of the batch (the reference's batch index b is a host-side loop) except
 *     the *_batch_* entry points at the end, which run the frame loop of B sequences per launch;
 *   - images are channels-last row-major (H, W, C) float32; map attributes are (N, 3)
 *     or (N, 1) float32 row-major; index tables are int64 (rows, 4) = [b, n, h, w];
 *   - 4x4 matrices are 16 contiguous float32, row-major, on the device.
 *
 * Arithmetic is specified operation-by-operation in DESIGN.md §arithmetic; the CPU oracle
 * (oracle/gs_oracle.c, same signatures with prefix gs_or_) restates it in plain C.
 */
#ifndef GRADSLAM_HIP_H
#define GRADSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported from libgradslam_hip.so (the library is built with -fvisibility=hidden) */
#define GS_API __attribute__((visibility("default")))

#define GS_OK 0
#define GS_ERR_INVALID 1 /* bad argument (null pointer, non-positive size, ...) */
#define GS_ERR_HIP 2     /* a HIP runtime call failed; see gs_last_error() */
#define GS_ERR_CAPACITY 3 /* an output would exceed the caller-provided capacity */

/* ABI version; bumped on any signature change. */
GS_API int gs_abi_version(void);
/* Message of the last non-OK return on this thread (never NULL). */
GS_API const char* gs_last_error(void);
/* Bytes of device scratch the functions below need for a map of `n_map` points and an
 * image of `n_pix` pixels (caller allocates once, reuses every frame). */
GS_API int64_t gs_scratch_bytes(int64_t n_map, int64_t n_pix);

/* Optional in-library kernel timing with HIP events recorded on the launch stream (used by
 * bench.py for the roofline line; off by default).  gs_profile_begin arms up to max_records
 * timed launches, gs_profile_end synchronises the device and aggregates,
 * gs_profile_read returns per kernel group: total milliseconds, launches, and work units
 * (kind 0 brute-force KNN: pair distances; other kinds: algorithmic bytes).
 * kinds: 0 brute-force KNN, 1 ICP linearise, 2 frame maps, 3 map projection, 4 association,
 * 5 fuse/append, 6 compaction / grid build, 7 ICP solve/update, 8 fused ICP search+linearise. */
GS_API int gs_profile_begin(int max_records);
GS_API int gs_profile_end(void);
GS_API int gs_profile_read(int kind, double* ms_total, int64_t* launches, double* work_total);

/* ---------------------------------------------------------------- K1: frame maps ------
 * depth -> local vertex map, local normal map, confidence alpha, validity mask.
 * Replaces RGBDImages._compute_vertex_map (structures/rgbdimages.py:643-679) incl.
 * inverse_intrinsics (geometry/projutils.py:405-450) and the pixel meshgrid,
 * RGBDImages._compute_normal_map (structures/rgbdimages.py:710-743),
 * RGBDImages.valid_depth_mask (:320-332) and get_alpha as called by fuse_with_map
 * (slam/fusionutils.py:69-72, :657).  Any of vertex/normal/alpha/valid may be NULL.
 * two_sigma_sq = (float)(2*sigma*sigma) evaluated in double by the caller. */
GS_API int gs_frame_maps_f32(const float* depth, const float* K16, int H, int W, float two_sigma_sq,
                      float* vertex, float* normal, float* alpha, uint8_t* valid, void* stream);

/* local maps + pose -> global vertex / normal maps.
 * Replaces RGBDImages._compute_global_vertex_map (structures/rgbdimages.py:681-708) and
 * _compute_global_normal_map (:745-762).  gnormal may be NULL.  pose16 == NULL means the
 * reference's "no poses" branch (plain copy). */
GS_API int gs_global_maps_f32(const float* vertex, const float* normal, const float* depth,
                       const float* pose16, int H, int W, float* gvertex, float* gnormal,
                       void* stream);

/* get_alpha on an arbitrary (n,3) point array (slam/fusionutils.py:16-73). */
GS_API int gs_alpha_f32(const float* points, int64_t n, float two_sigma_sq, float eps, float* alpha,
                 void* stream);

/* Reverse mode of gs_alpha_f32 = what PyTorch autograd computes through get_alpha (slam/fusionutils.py:69-72; the
 * reference's own gradient check: tests/slam/test_fusionutils.py:56-75): points_bar[i] = alpha_bar[i] * d alpha_i / d p_i
 * (zero where the clamp is active), sigma_terms[i] (may be NULL) = alpha_bar[i] * alpha_i * |p_i|^2, so that
 * d loss / d sigma = sum_i sigma_terms[i] / sigma^3. */
GS_API int gs_alpha_backward_f32(const float* points, int64_t n, float two_sigma_sq, float eps,
                          const float* alpha_bar, float* points_bar, float* sigma_terms, void* stream);

/* ------------------------------------------------------- K2: ICP source / target sets --
 * Valid pixels of the [::ds, ::ds] lattice, raster order -> compact point list.
 * Replaces downsample_rgbdimages (odometry/icputils.py:623-669).  out_* have room for
 * ceil(H/ds)*ceil(W/ds) rows; out_nrm/out_rgb (and gnormal/rgb) may be NULL.
 * count_out: device int64[1]. */
GS_API int gs_downsample_frame_f32(const float* gvertex, const float* gnormal, const float* rgb,
                            const float* depth, int H, int W, int ds, float* out_pts,
                            float* out_nrm, float* out_rgb, int64_t* count_out, void* scratch,
                            void* stream);

/* Project every map point into a frame: pix[n] = h*W + w of the pixel it lands on, or -1
 * when it is not "active".  Replaces the arithmetic of find_active_map_points
 * (slam/fusionutils.py:249-274): inverse_transformation, Pointclouds.transform
 * (structures/pointclouds.py:466-573), pinhole_projection_ (:575-614, geometry/projutils.py
 * :225-238), the in-frame test and round/clamp.  pose16 is camera-to-world. */
GS_API int gs_project_map_f32(const float* points, int64_t n_map, const float* pose16,
                       const float* K16, int H, int W, int32_t* pix, void* stream);

/* pix[] -> ordered table rows [b, n, h, w] (find_active_map_points' return value,
 * slam/fusionutils.py:276-282).  rows_out has room for n_map rows; count_out device int64. */
GS_API int gs_active_table_i64(const int32_t* pix, int64_t n_map, int W, int64_t b, int64_t* rows_out,
                        int64_t* count_out, void* scratch, void* stream);

/* Active map points whose pixel lies on the [::ds, ::ds] lattice, in map order, gathered.
 * Replaces downsample_pointclouds (odometry/icputils.py:596-620) fed by
 * find_active_map_points.  out_* have room for `cap` rows: the kernels never write past cap, count_out[0]
 * always receives the TRUE number of selected points (the caller compares it with cap).
 * normals/colors optional. */
GS_API int gs_select_targets_f32(const int32_t* pix, int64_t n_map, int W, int ds, const float* points,
                          const float* normals, const float* colors, float* out_pts,
                          float* out_nrm, float* out_rgb, int64_t cap, int64_t* count_out,
                          void* scratch, void* stream);

/* downsample_pointclouds on an explicit table (rows of one sequence): keeps rows with
 * h % ds == 0 and w % ds == 0, gathers map attributes by n.  (odometry/icputils.py:596-620) */
GS_API int gs_downsample_table_f32(const int64_t* rows, int64_t n_rows, int ds, const float* points,
                            const float* normals, const float* colors, float* out_pts,
                            float* out_nrm, float* out_rgb, int64_t* count_out, void* scratch,
                            void* stream);

/* ------------------------------------------------------------------ K3: exact 1-NN -----
 * For every src point the index of the nearest tgt point (squared L2, lowest index on
 * ties) and that squared distance.  Replaces chamferdist.chamfer.knn_points as called at
 * odometry/icputils.py:200 (third-party, chamferdist==1.0.0, requirements.txt:2).
 * best_scratch: device uint64[n_src].  out_idx int64[n_src], out_d2 float[n_src] (may be NULL). */
GS_API int gs_knn1_f32(const float* src, int64_t n_src, const float* tgt, int64_t n_tgt,
                int64_t* out_idx, float* out_d2, uint64_t* best_scratch, void* stream);

/* Same result through the uniform-grid engine the ICP loop uses (targets counting-sorted into
 * cells once, Chebyshev-shell search with an exactness bound, brute-force pass for queries it
 * cannot resolve): bit-identical to gs_knn1_f32.  scratch: gs_knn1_grid_scratch_bytes().
 * unresolved_out (HOST pointer, may be NULL; forces a stream sync): queries finished by brute force. */
GS_API int64_t gs_knn1_grid_scratch_bytes(int64_t n_src, int64_t n_tgt);
GS_API int gs_knn1_grid_f32(const float* src, int64_t n_src, const float* tgt, int64_t n_tgt,
                            int64_t* out_idx, float* out_d2, void* scratch, int64_t* unresolved_out_host,
                            void* stream);

/* --------------------------------------------------------- K4: Gauss-Newton system -----
 * gauss_newton_solve (odometry/icputils.py:93-232): KNN + rows A_i = [n, s x n],
 * b_i = n.(d - s).  A (n_src,6) and b (n_src) are written densely for ALL src rows plus a
 * keep mask (dist filter, :203-208); idx int64[n_src].  dist_thresh < 0 means None. */
GS_API int gs_gauss_newton_rows_f32(const float* src, int64_t n_src, const float* tgt,
                             const float* tgt_normals, int64_t n_tgt, float dist_thresh,
                             float* A, float* b, int64_t* idx, uint8_t* keep,
                             uint64_t* best_scratch, void* stream);

/* solve_linear_system (odometry/icputils.py:22-90): x = (A^T A + damp I)^-1 A^T b for
 * A (n_rows, ncols), 1 <= ncols <= 8 (the SLAM path uses 6).  keep may be NULL.
 * x: device float[ncols]. */
GS_API int gs_solve_normal_eq_f32(const float* A, const float* b, const uint8_t* keep, int64_t n_rows,
                           int ncols, float damp, float* x, void* stream);

/* se3_exp (geometry/se3utils.py:77-115): xi(6) -> T(4x4). */
GS_API int gs_se3_exp_f32(const float* xi6, float* T16, void* stream);

/* Reverse mode of gs_se3_exp_f32 (PyTorch autograd through geometry/se3utils.py:77-115): xi_bar(6) from T_bar(4x4). */
GS_API int gs_se3_exp_backward_f32(const float* xi6, const float* Tbar16, float* xi_bar6, void* stream);

/* relative_transformation(T01, T02, orthogonal_rotations=False) (geometry/geometryutils.py:413-478; the
 * arithmetic of GroundTruthOdometryProvider.provide, odometry/groundtruth.py:74-78, and of the dataset
 * loaders' pose preprocessing, datasets/tum.py:497-499): out[m] = compose(inv(T01[m]), T02[m]) for n
 * pairs of 4x4 float32 matrices. */
GS_API int gs_relative_pose_f32(const float* T01, const float* T02, int64_t n, float* out, void* stream);

/* transform_pointcloud (geometry/geometryutils.py:737-794): out = R p + t. */
GS_API int gs_transform_points_f32(const float* pts, int64_t n, const float* T16, float* out,
                            void* stream);

/* Whole LM loop on the device, no host sync.
 * mode 0: point_to_plane_ICP      (odometry/icputils.py:235-367)
 * mode 1: point_to_plane_gradICP  (odometry/icputils.py:370-545)
 * src is NOT modified (a working copy lives in scratch).  init16: initial transform.
 * compose16 (may be NULL): if given, T_out = T_icp * compose16, i.e. the caller's
 * compose_transformations(transform, prev_pose) (slam/icpslam.py:245-247) fused in.
 * out_T16: device float[16]; out_idx (may be NULL): int64[n_src] neighbour indices of the
 * last iteration's first linearisation (:328,:367).  icp_scratch: gs_icp_scratch_bytes(). */
typedef struct gs_icp_params {
  int mode;          /* 0 = ICP, 1 = gradICP */
  int numiters;      /* default 20 */
  float damp;        /* initial damping, default 1e-8 */
  float dist_thresh; /* < 0: None */
  float lambda_max;  /* gradICP, default 2.0 */
  float B;           /* gradICP, default 1.0 */
  float B2;          /* gradICP, default 1.0 */
  float nu;          /* gradICP, default 200.0 */
} gs_icp_params;

GS_API int64_t gs_icp_scratch_bytes(int64_t n_src, int64_t n_tgt);
GS_API int gs_icp_f32(const float* src, int64_t n_src, const float* tgt, const float* tgt_normals,
               int64_t n_tgt, const float* init16, const float* compose16,
               const gs_icp_params* params_host, float* out_T16, int64_t* out_idx,
               void* icp_scratch, void* stream);

/* ------------------------------------------------- K7: backward of the frame maps ------
 * Reverse mode of gs_frame_maps_f32 w.r.t. depth: depth_bar (H,W) from the adjoints of the local
 * vertex map, normal map and alpha (any may be NULL), as PyTorch autograd does through
 * structures/rgbdimages.py:643-743 and slam/fusionutils.py:69-72.  scratch_6hw: 6*H*W floats,
 * needed when normal_bar is given.  K_bar16 (may be NULL): the gradient w.r.t. the 4x4 intrinsics as autograd
 * produces it through inverse_intrinsics (geometry/projutils.py:437-449): entries [0,0] (fx), [0,2] (cx), [1,1] (fy),
 * [1,2] (cy), zero elsewhere; needs kbar_scratch of gs_frame_maps_backward_kbar_scratch_bytes(H, W) bytes (fixed-order
 * float64 block sums).  depth_bar may be NULL when only K_bar16 is wanted. */
GS_API int64_t gs_frame_maps_backward_kbar_scratch_bytes(int H, int W);
GS_API int gs_frame_maps_backward_f32(const float* depth, const float* K16, int H, int W, float two_sigma_sq,
                                      const float* vertex_bar, const float* normal_bar, const float* alpha_bar,
                                      float* depth_bar, float* scratch_6hw, float* K_bar16, void* kbar_scratch,
                                      void* stream);
/* Reverse mode of gs_global_maps_f32 w.r.t. the local maps (rgbdimages.py:681-762):
 * vertex_bar = R^T (gvertex_bar * valid), normal_bar = R^T gnormal_bar.  (No pose gradient.) */
GS_API int gs_global_maps_backward_f32(const float* gvertex_bar, const float* gnormal_bar, const float* depth,
                                       const float* pose16, int H, int W, float* vertex_bar, float* normal_bar,
                                       void* stream);
/* ... and w.r.t. the pose: pose_bar (4x4, last row zero) with R_bar = sum_p valid_p gvertex_bar_p (x) v_p +
 * sum_p gnormal_bar_p (x) n_p and t_bar = sum_p valid_p gvertex_bar_p (the recovered pose of a frame feeds the
 * global maps that are fused into the map: this is the link map -> pose -> ICP of the reference's graph). */
GS_API int64_t gs_global_maps_pose_backward_scratch_bytes(int H, int W);
GS_API int gs_global_maps_pose_backward_f32(const float* vertex, const float* normal, const float* depth,
                                            const float* gvertex_bar, const float* gnormal_bar, int H, int W,
                                            float* pose_bar16, void* scratch, void* stream);
/* Reverse mode of gs_downsample_frame_f32 (points): gvertex_bar (H,W,3) = zeros with the compact
 * adjoints pts_bar scattered back to the valid lattice pixels (odometry/icputils.py:654-660). */
GS_API int gs_downsample_frame_backward_f32(const float* pts_bar, const float* depth, int H, int W, int ds,
                                            float* gvertex_bar, void* scratch, void* stream);

/* ------------------------------------------------------------- K7: (grad)ICP backward ----
 * Both solvers are differentiable: mode 1 (gradICP, odometry/icputils.py:479-545) through the soft accept and
 * damping functions, mode 0 (hard LM, :310-367) through the accepted steps only (the accept test and the
 * damping schedule are constants, as in the reference's autograd graph).
 * Differentiable gradICP (config C3): gs_icp_tape_f32 is gs_icp_f32 that additionally records the
 * forward tape (caller-owned, gs_icp_tape_bytes(n_src, numiters) bytes: per-iteration source cloud,
 * neighbour indices of both searches, float32 normal equations, trace).  gs_icp_backward_f32 then
 * turns dL/dT (T_bar16, the gradient w.r.t. the UN-composed transform returned with compose16 ==
 * NULL) into dL/d(src), dL/d(tgt), dL/d(tgt_normals), dL/d(init) exactly as PyTorch autograd does
 * through odometry/icputils.py:479-545 (indices and the dist_thresh filter are constants).  Any of
 * the four outputs may be NULL.  mode must be 1 (gradICP).  scratch: gs_icp_backward_scratch_bytes. */
GS_API int64_t gs_icp_tape_bytes(int64_t n_src, int numiters);
GS_API int gs_icp_tape_f32(const float* src, int64_t n_src, const float* tgt, const float* tgt_normals,
                           int64_t n_tgt, const float* init16, const float* compose16,
                           const gs_icp_params* params_host, float* out_T16, int64_t* out_idx,
                           void* icp_scratch, void* tape, void* stream);
GS_API int64_t gs_icp_backward_scratch_bytes(int64_t n_src, int64_t n_tgt);
GS_API int gs_icp_backward_f32(const void* tape, const float* src, int64_t n_src, const float* tgt,
                               const float* tgt_normals, int64_t n_tgt, const float* init16,
                               const gs_icp_params* params_host, const float* T_bar16, float* src_bar,
                               float* tgt_bar, float* normals_bar, float* init_bar16, void* scratch,
                               void* stream);

/* gs_icp_f32 with DEVICE-SIDE point counts: n_src_bound / n_tgt_bound are upper bounds used for
 * launch geometry and scratch sizing (src / tgt buffers must hold that many rows); the actual
 * counts are read on the device from n_src_dev / n_tgt_dev (e.g. the count_out of
 * gs_downsample_frame_f32 / gs_select_targets_f32), so the host never reads them back: no sync
 * between selecting the ICP point sets and solving.  Either pointer may be NULL (= the bound is exact). */
GS_API int gs_icp_dc_f32(const float* src, int64_t n_src_bound, const int64_t* n_src_dev, const float* tgt,
                         const float* tgt_normals, int64_t n_tgt_bound, const int64_t* n_tgt_dev,
                         const float* init16, const float* compose16, const gs_icp_params* params_host,
                         float* out_T16, int64_t* out_idx, void* icp_scratch, void* stream);

/* The SLAM loop's solve without gathering the target set: the targets are the rows of the surfel map
 * (map_points / map_normals, n_map_bound rows, actual count in n_map_dev or NULL) whose projection `pix`
 * (gs_project_map_*_f32 with the previous pose) lies on the [::ds, ::ds] pixel lattice -- the set
 * find_active_map_points + downsample_pointclouds select (slam/icpslam.py:241-242, odometry/icputils.py:
 * 596-597) -- binned straight from the map.  Same transform, bit for bit, as gs_icp_dc_f32 on the output of
 * gs_select_targets_f32 (candidate order and ties follow the map row order, which is the order of the
 * compacted set).  icp_scratch: gs_icp_scratch_bytes(n_src_bound, n_map_bound). */
GS_API int gs_icp_map_dc_f32(const float* src, int64_t n_src_bound, const int64_t* n_src_dev,
                             const float* map_points, const float* map_normals, const int32_t* pix,
                             int64_t n_map_bound, const int64_t* n_map_dev, int W, int ds, const float* init16,
                             const float* compose16, const gs_icp_params* params_host, float* out_T16,
                             void* icp_scratch, void* stream);

/* Per-iteration record kept in icp_scratch for the backward pass / diagnostics:
 * gs_icp_trace_f32 copies (numiters, 12) floats = [err, new_err, damp_after, sigmoid, xi(6), pad(2)]. */
GS_API int gs_icp_trace_f32(const void* icp_scratch, int numiters, float* trace_out, void* stream);

/* ------------------------------------------------- K5: surfel association (PointFusion) -
 * find_similar_map_points on an explicit table (slam/fusionutils.py:381-401):
 * mask[r] = |f - p| < dist_th  &&  f_n . p_n > dot_th  for row r = [b, n, h, w]. */
GS_API int gs_similar_rows_f32(const int64_t* rows, int64_t n_rows, const float* points,
                        const float* normals, const float* gvertex, const float* gnormal, int W,
                        float dist_th, float dot_th, uint8_t* mask, void* stream);

/* find_best_unique_correspondences on an explicit table (slam/fusionutils.py:489-544):
 * per pixel the row minimising (1/(ccount+1e-20), |p-f|^2, n); output rows sorted by (h, w)
 * as [b, n, h, w].  best_pix: device int32[H*W] work array (also returned: winner n or -1). */
GS_API int gs_best_unique_rows_f32(const int64_t* rows, int64_t n_rows, const float* points,
                            const float* ccounts, const float* gvertex, int H, int W, int64_t b,
                            int32_t* best_pix, int64_t* rows_out, int64_t* count_out,
                            void* scratch, void* stream);

/* Fused find_correspondences (slam/fusionutils.py:549-577) without tables:
 * pix[] (from gs_project_map_f32) + map + frame -> best_pix[H*W] = winning map index or -1.
 * similar (may be NULL): uint8[n_map] = is_similar_mask scattered by map index. */
GS_API int gs_associate_f32(const int32_t* pix, int64_t n_map, const float* points,
                     const float* normals, const float* ccounts, const float* gvertex,
                     const float* gnormal, int H, int W, float dist_th, float dot_th,
                     int32_t* best_pix, uint8_t* similar, void* scratch, void* stream);

/* best_pix[] -> ordered table rows [b, n, h, w] sorted by (h, w). */
GS_API int gs_best_table_i64(const int32_t* best_pix, int H, int W, int64_t b, int64_t* rows_out,
                      int64_t* count_out, void* scratch, void* stream);
/* table rows -> best_pix[] (for fuse_with_map called with an explicit table). */
GS_API int gs_rows_to_best_pix(const int64_t* rows, int64_t n_rows, int H, int W, int32_t* best_pix,
                        void* stream);

/* ------------------------------------------------------ K6: merge + append (PointFusion) -
 * fuse_with_map (slam/fusionutils.py:653-720) + Pointclouds.append_points
 * (structures/pointclouds.py:1117-1237) on a capacity-backed surfel store:
 *   matched rows n = best_pix[p]:  cc' = cc + a;  x' = (cc*x + a*f) * (1/cc')  for points,
 *   normals, colours;  every other row is rewritten as (cc*x)*(1/cc) exactly as the
 *   reference does (renorm_all != 0; 0 skips that and leaves unmatched rows untouched).  The reference skips
 *   the merge when the correspondence table OF THE WHOLE BATCH is empty (:659); this call sees one sequence, so
 *   renorm_all = 1 skips it when this sequence's table is empty and renorm_all = 2 never skips it (the caller knows
 *   that another sequence of the batch has matches);
 *   unmatched valid pixels are appended in raster order at rows n_map .. n_map+n_new-1.
 * The arrays must have room for n_map + H*W rows.  n_map_host is the current size;
 * new_count_out: device int64[1] receives n_map + n_new. */
GS_API int gs_fuse_append_f32(float* points, float* normals, float* colors, float* ccounts,
                       int64_t n_map_host, int64_t capacity, const int32_t* best_pix,
                       const float* gvertex, const float* gnormal, const float* rgb,
                       const float* alpha, const float* depth, int H, int W, int renorm_all,
                       int64_t* new_count_out, void* scratch, void* stream);

/* update_map_aggregate (slam/fusionutils.py:725-758) / pointclouds_from_rgbdimages
 * (structures/utils.py:7-57): append every valid pixel, raster order.  normals / colors /
 * ccounts (with gnormal / rgb / alpha) may be NULL. */
GS_API int gs_append_valid_f32(float* points, float* normals, float* colors, float* ccounts,
                        int64_t n_map_host, int64_t capacity, const float* gvertex,
                        const float* gnormal, const float* rgb, const float* alpha,
                        const float* depth, int H, int W, int64_t* new_count_out, void* scratch,
                        void* stream);

/* ICP source set of a frame without compaction (the points of downsample_rgbdimages,
 * odometry/icputils.py:654-668, computed from the LOCAL vertex map and the pose as in
 * structures/rgbdimages.py:681-708): out_pts (ceil(H/ds) * ceil(W/ds), 3), raster order, NaN where the
 * lattice pixel has no depth.  The grid path of gs_icp_f32 / gs_icp_dc_f32 ignores NaN source points
 * (no search, no row, out_idx = -1), so the lattice is a valid `src` as is. */
GS_API int gs_lattice_source_f32(const float* vertex, const float* depth, const float* pose16, int H, int W, int ds,
                                 float* out_pts, void* stream);

/* ---- dataset -> device ingest (datasets/tum.py:448-477 _preprocess_color / _preprocess_depth; the same
 * two steps in datasets/icl.py and datasets/scannet.py).  raw: the decoded PNG on the device (depth
 * uint16 (H0, W0); colour uint8 (H0, W0, 3)); out: float32 (H, W) / (H, W, 3).  Depth: cv2.INTER_NEAREST
 * to (H, W), then / scale_div (TUM 5000, ICL 5000, ScanNet 1000) in float64, cast last.  Colour:
 * cv2.INTER_LINEAR in float64 with float32 weights, optional / 255.  Same size: exact copy. */
GS_API int gs_ingest_depth_u16_f32(const uint16_t* raw, int H0, int W0, float* out, int H, int W,
                                   double scale_div, void* stream);
GS_API int gs_ingest_color_u8_f32(const uint8_t* raw, int H0, int W0, float* out, int H, int W, int normalize,
                                  void* stream);
/* Streaming ingest (round 4; datasets/tum.py:352-434 per frame, datasets/datautils.py:73-118): n_frames native-size frames
 * in one launch -- depth uint16 (n, H, W) -> float32 metres (raw / scale_div in float64, then float32), colour uint8
 * (n, H, W, 3) -> float32 (optionally / 255): the arithmetic of the two entry points above without a resize.  Either pair
 * may be NULL.  n_frames * H * W must be a multiple of 4. */
GS_API int gs_ingest_frames_native_f32(const uint16_t* depth_raw, const uint8_t* color_raw, int64_t n_frames, int H, int W,
                                       double scale_div, int normalize, float* depth_out, float* color_out, void* stream);
/* Device-side address of a pinned host allocation (hipHostGetDevicePointer), for kernels that read raw frames straight
 * from host memory; GS_ERR_INVALID when the range is not device-mapped. */
GS_API int gs_host_device_pointer(const void* host_ptr, void** dev_ptr_out);

/* ---- the per-frame map pipeline with the surfel count kept ON THE DEVICE -------------------
 * Same kernels and results as the functions they are named after; `n_map_bound` (host) is an
 * upper bound of the surfel count used for launch geometry / scratch sizing, the actual count is
 * read by the kernels from `n_map_dev` (int64[1], typically the new_count_out of the previous
 * frame's gs_fuse_append_dc_f32).  With these, one PointFusion frame (slam/icpslam.py:137-161 +
 * slam/fusionutils.py:761-789) never returns to the host: no read-back, no stream sync.
 * new_count_out must not alias n_map_dev.  Capacity must cover n_map_bound + H*W rows. */
GS_API int gs_project_map_dc_f32(const float* points, int64_t n_map_bound, const int64_t* n_map_dev,
                                 const float* pose16, const float* K16, int H, int W, int32_t* pix,
                                 void* stream);
GS_API int gs_select_targets_dc_f32(const int32_t* pix, int64_t n_map_bound, const int64_t* n_map_dev, int W,
                                    int ds, const float* points, const float* normals, const float* colors,
                                    float* out_pts, float* out_nrm, float* out_rgb, int64_t cap,
                                    int64_t* count_out, void* scratch, void* stream);
GS_API int gs_associate_dc_f32(const int32_t* pix, int64_t n_map_bound, const int64_t* n_map_dev,
                               const float* points, const float* normals, const float* ccounts,
                               const float* gvertex, const float* gnormal, int H, int W, float dist_th,
                               float dot_th, int32_t* best_pix, uint8_t* similar, void* scratch, void* stream);
GS_API int gs_fuse_append_dc_f32(float* points, float* normals, float* colors, float* ccounts,
                                 int64_t n_map_bound, const int64_t* n_map_dev, int64_t capacity,
                                 const int32_t* best_pix, const float* gvertex, const float* gnormal,
                                 const float* rgb, const float* alpha, const float* depth, int H, int W,
                                 int renorm_all, int64_t* new_count_out, void* scratch, void* stream);
GS_API int gs_append_valid_dc_f32(float* points, float* normals, float* colors, float* ccounts,
                                  int64_t n_map_bound, const int64_t* n_map_dev, int64_t capacity,
                                  const float* gvertex, const float* gnormal, const float* rgb,
                                  const float* alpha, const float* depth, int H, int W,
                                  int64_t* new_count_out, void* scratch, void* stream);

/* Backward of gs_fuse_append_f32 = reverse mode of fuse_with_map (slam/fusionutils.py:653-720): adjoints of
 * the fused map (n_new rows; the first n_old are the merged old rows, the rest the appended pixels in raster
 * order) -> adjoints of the old map rows (values BEFORE the merge are passed in) and of the frame's global
 * vertex / normal maps, colours (H, W, 3) and alpha (H, W).  best_pix: the correspondence table used in the
 * forward.  scratch: gs_scratch_bytes(n_old, H * W). */
GS_API int gs_fuse_append_backward_f32(const float* points, const float* normals, const float* colors,
                                       const float* ccounts, int64_t n_old, const int32_t* best_pix,
                                       const float* gvertex, const float* gnormal, const float* rgb,
                                       const float* alpha, const float* depth, int H, int W, int renorm_all,
                                       const float* P_bar, const float* N_bar, const float* C_bar,
                                       const float* F_bar, int64_t n_new, float* old_points_bar,
                                       float* old_normals_bar, float* old_colors_bar, float* old_ccounts_bar,
                                       float* gvertex_bar, float* gnormal_bar, float* rgb_bar, float* alpha_bar,
                                       void* scratch, void* stream);

/* update_map_fusion (slam/fusionutils.py:761-789) of one sequence as ONE call: global maps of the frame under
 * `pose16` (structures/rgbdimages.py:681-762), projection + association of the map (fusionutils.py:198-577) and
 * the confidence-weighted merge + ordered append (fusionutils.py:580-722), regrouped into 4 launches (round 5: no pick pass over the map).  Same
 * results, bit for bit, as gs_global_maps_f32 + gs_project_map_dc_f32 + gs_associate_dc_f32 +
 * gs_fuse_append_dc_f32.  vertex / normal: LOCAL maps (H, W, 3); alpha (H, W); outputs gvertex / gnormal
 * (H, W, 3), best_pix (H*W) (the correspondence table, -1 = none), new_count_out (must not alias n_map_dev;
 * n_map_dev may be NULL = the bound is exact).  capacity >= n_map_bound + H*W. */
GS_API int64_t gs_update_map_scratch_bytes(int64_t n_map_bound, int H, int W);
GS_API int gs_update_map_fusion_dc_f32(float* points, float* normals, float* colors, float* ccounts,
                                       int64_t n_map_bound, const int64_t* n_map_dev, int64_t capacity,
                                       const float* vertex, const float* normal, const float* depth,
                                       const float* rgb, const float* alpha, const float* pose16, const float* K16,
                                       int H, int W, float dist_th, float dot_th, int renorm_all, float* gvertex,
                                       float* gnormal, int32_t* best_pix, int64_t* new_count_out, void* scratch,
                                       void* stream);

/* ---- the frame loop for B INDEPENDENT sequences per call -----------------------------------------------
 * The reference loops `for b in range(B)` over the sequences of a batch (odometry/gradicp.py:105,
 * odometry/icp.py:84, slam/fusionutils.py:710-713); one 640x480 sequence cannot fill an MI355X (its frame is a
 * chain of ~50 dependent launches).  These entry points run EVERY kernel of a frame for all B
 * sequences at once (workgroup b serves sequence b mod B, so with B = 8 every sequence lives on one XCD), which
 * pays the dependent-launch floor once per B frames.  Per sequence the results are bit-identical to the
 * one-sequence entry points they are named after (same device functions).  The descriptor arrays are HOST arrays
 * of device pointers; nothing is read back. */
typedef struct gs_map_view {
  float* points;        /* (capacity, 3) */
  float* normals;       /* (capacity, 3) */
  float* colors;        /* (capacity, 3); may be NULL where only points / normals are read */
  float* ccounts;       /* (capacity, 1); idem */
  int64_t capacity;     /* rows the buffers hold */
  int64_t n_bound;      /* host-side upper bound of the surfel count (launch geometry, scratch sizing) */
  const int64_t* n_dev; /* device int64[1]: the actual count, or NULL when n_bound is exact */
} gs_map_view;

/* K1 for n_frames = B x frames_per_K frames in one launch (an RGBDImages of B sequences x L frames: frames_per_K =
 * L): frame f = b * L + l reads the (H, W) depth image at depth + b * depth_stride_seq + l * depth_stride_frame
 * (strides in floats: a contiguous stack has L*H*W and H*W; a one-frame slice frames[:, s] of a longer stack keeps
 * the stack's strides, so no copy is needed) and the intrinsics K16 + 16 * b; outputs are dense: vertex / normal
 * (n_frames, H, W, 3), alpha (n_frames, H, W); normal / alpha may be NULL.  Same arithmetic as gs_frame_maps_f32. */
GS_API int gs_frame_maps_batch_f32(const float* depth, int64_t depth_stride_seq, int64_t depth_stride_frame,
                                   const float* K16, int n_frames, int frames_per_K, int H, int W, float two_sigma_sq,
                                   float* vertex, float* normal, float* alpha, void* stream);

/* ICPSLAM._localize (slam/icpslam.py:238-247) for B sequences: ICP source = the live frame's [::ds, ::ds] lattice
 * under the previous pose (gs_lattice_source_f32), targets = the map rows that project onto that lattice in the
 * previous frame (gs_project_map_dc_f32 + the selection of gs_select_targets_f32, binned straight from the map),
 * numiters (grad)LM iterations, result composed with the previous pose: out_pose16 = T_icp * prev_pose16
 * (the transform gs_icp_map_dc_f32 returns).  scratch: gs_localize_scratch_bytes(H, W, ds, rows) bytes per sequence
 * with rows = max(map.capacity, map.n_bound): the layout follows the CAPACITY of the map buffers, so that it does not
 * move from frame to frame while the map grows inside them (after the call the scratch holds the solver state / trace
 * at the offset gs_icp_trace_f32 expects, the projection table and the candidate lists of far queries). */
typedef struct gs_localize_seq {
  const float* vertex;      /* live frame, LOCAL vertex map (H, W, 3) */
  const float* depth;       /* live frame depth (H, W) */
  const float* K16;         /* intrinsics of the previous frame */
  const float* prev_pose16; /* pose of the previous frame (initial guess and composition) */
  gs_map_view map;          /* points + normals are read; n_bound > 0 */
  float* out_pose16;        /* recovered pose of the live frame */
  void* scratch;
} gs_localize_seq;
GS_API int64_t gs_localize_scratch_bytes(int H, int W, int ds, int64_t n_map_bound);
GS_API int gs_localize_batch_f32(const gs_localize_seq* seqs_host, int B, int H, int W, int ds,
                                 const gs_icp_params* params_host, void* stream);
/* Diagnostics of the last solve a scratch was used for (tests, tools; synchronises the stream): out4[0] = number of
 * source points the first search found far from every target and handed to the candidate-list builder (0 when the
 * lists are disabled), out4[1] = how many of them hold a proven list after the last search, out4[2] = source points
 * handed to the second builder pass, out4[3] = how many of out4[0] have a list that fits (exactness radius > 0).
 * H, W, ds and map_rows (= max(map.capacity, map.n_bound) of that call) locate the counters in the scratch. */
GS_API int gs_localize_far_stats_i64(const void* scratch, int H, int W, int ds, int64_t map_rows, int64_t* out4_host,
                                     void* stream);
/* Diagnostics of the candidate lists of ordinary source points (round 4; tests, tools; synchronises the stream) for the
 * last solve a scratch was used for: out[h] = source points whose list gave no proof in launch h of the solve (h = 2 x
 * iteration + half; they were re-searched by the 2x2x2 scan and got a new list), out[64 + h] = source points whose list
 * was empty (no radius fitted the slots: re-searched as well), out[128 + h] = source points that had no list in launch h
 * because neither the 2x2x2 stage nor the cubes could prove them (as without lists).  All zero when the solve kept no
 * lists (GRADSLAM_HIP_ICP_LISTS=0, far-candidate lists in use, blocks walking several groups of row units) and for the
 * launches before the lists start (GRADSLAM_HIP_ICP_LISTS_FROM).  out_host holds 192 values. */
GS_API int gs_localize_list_stats_i64(const void* scratch, int H, int W, int ds, int64_t map_rows, int64_t* out192_host,
                                      void* stream);

/* update_map_fusion (slam/fusionutils.py:761-789) for B sequences: gs_update_map_fusion_dc_f32 per sequence, 4
 * launches for the whole batch.  scratch: gs_update_map_scratch_bytes(map.n_bound, H, W) per sequence. */
typedef struct gs_update_seq {
  gs_map_view map;          /* all four attributes, capacity >= n_bound + H*W */
  const float* vertex;      /* LOCAL vertex / normal maps (H, W, 3) */
  const float* normal;
  const float* depth;       /* (H, W) */
  const float* rgb;         /* (H, W, 3) */
  const float* alpha;       /* (H, W) */
  const float* pose16;      /* the frame's (recovered) pose */
  const float* K16;
  float* gvertex;           /* out: global maps (H, W, 3) */
  float* gnormal;
  int32_t* best_pix;        /* out: correspondence table (H*W), -1 = none */
  int64_t* new_count_out;   /* out: device int64[1]; must not alias map.n_dev */
  void* scratch;
} gs_update_seq;
GS_API int gs_update_map_fusion_batch_f32(const gs_update_seq* seqs_host, int B, int H, int W, float dist_th,
                                          float dot_th, int renorm_all, void* stream);

/* One frame of PointFusion.step (slam/icpslam.py:140-178 with the _map override of slam/pointfusion.py:107-112) for B
 * sequences in ONE call: the live frames' local maps (gs_frame_maps_batch_f32), the poses (gs_localize_batch_f32) and the
 * map update under those poses (gs_update_map_fusion_batch_f32) -- the same kernels in the same order, enqueued from
 * C so that the host cost of a frame is one foreign call (49 launches; the host runs frames ahead of the device, so the
 * launches wait in the queue: a graph replay of the ICP launches, which round 3 had, bought nothing and went with that engine).
 * vertex / normal / alpha of all sequences must be dense ((B, H, W, 3) / (B, H, W): seqs[b].vertex = seqs[0].vertex +
 * b * 3 * H * W, ...) and the depth images equally strided (seqs[b].depth = seqs[0].depth + b * stride). */
typedef struct gs_step_seq {
  const float* depth;        /* live frame (H, W) */
  const float* rgb;          /* (H, W, 3) */
  const float* K16;          /* intrinsics (of the previous = live frame) */
  const float* prev_pose16;  /* pose of the previous frame */
  float* out_pose16;         /* out: recovered pose of the live frame (must not alias prev_pose16) */
  float* vertex;             /* out: LOCAL vertex / normal maps (H, W, 3), sample confidences (H, W) */
  float* normal;
  float* alpha;
  float* gvertex;            /* out, optional: global maps under the recovered pose.  Both NULL (for every sequence of the */
  float* gnormal;            /* call): not written -- the update computes the global vertex / normal of a pixel where it   */
                             /* uses them (same bits) and skips its per-pixel pass; gs_global_maps_f32 gives them on demand */
  int32_t* best_pix;         /* out: correspondence table (H*W) */
  int64_t* new_count_out;    /* out: device int64[1], must not alias map.n_dev */
  gs_map_view map;           /* all four attributes; n_bound > 0; capacity >= n_bound + H*W */
  void* loc_scratch;         /* gs_localize_scratch_bytes(H, W, ds, map.capacity) */
  void* upd_scratch;         /* gs_update_map_scratch_bytes(map.capacity, H, W) */
} gs_step_seq;
GS_API int gs_pointfusion_step_batch_f32(const gs_step_seq* seqs_host, int B, int H, int W, int ds,
                                         const gs_icp_params* params_host, float two_sigma_sq, float dist_th,
                                         float dot_th, int renorm_all, void* stream);

/* API-level projective / Lie helpers (the SLAM kernels fuse their own copies; these are the drop-in names).
 * gs_project_points_f32: project_points (geometry/projutils.py:92-238) on n points of cdim (3 | 4) coordinates; point i
 *   uses the 4x4 matrix proj16 + 16 * (i / pts_per_mat); out_uv (n, 2) = (x'/z', y'/z'), z' = 1 where it is 0.
 * gs_unproject_points_f32: unproject_points (geometry/projutils.py:241-402): (K^-1 [u v w]) * depth, pdim (2 | 3).
 * gs_lie_small_f32: op 0 so3_hat (3 -> 3x3), 1 se3_hat (6 -> 4x4), 2 so3_exp (3 -> 3x3) (geometry/se3utils.py:11-74). */
GS_API int gs_project_points_f32(const float* cam_coords, int cdim, int64_t n, const float* proj16, int64_t pts_per_mat,
                                 float* out_uv, void* stream);
GS_API int gs_unproject_points_f32(const float* pixel_coords, int pdim, int64_t n, const float* kinv9,
                                   int64_t pts_per_mat, const float* depths, float* out_xyz, void* stream);
GS_API int gs_lie_small_f32(int op, const float* in, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRADSLAM_HIP_H */
