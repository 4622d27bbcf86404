/*
 * pof_abi.h -- C ABI of libpof_hip.so, the MI355X (gfx950) implementation of the
 * per-point planar-flow hot path of huzjkevin/planar_optical_flow.
 *
 * Conventions
 *   - every entry point returns int: POF_OK (0) or a negative POF_E_* code;
 *   - all array arguments are DEVICE pointers owned by the caller; the library
 *     never allocates or frees user-visible memory and keeps no global state
 *     (re-entrant, thread-safe); scratch space is passed in explicitly;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream);
 *   - shapes are C-contiguous unless a stride argument says otherwise.
 *
 * Each declaration cites the reference interface it replaces (paths relative
 * to the reference checkout).
 */
#ifndef POF_ABI_H
#define POF_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POF_ABI_VERSION 1

enum {
    POF_OK = 0,
    POF_E_BADARG = -1, /* null pointer / illegal enum / negative size          */
    POF_E_SHAPE = -2,  /* sizes inconsistent with each other or with a limit   */
    POF_E_LAUNCH = -3, /* the HIP runtime reported an error on launch          */
    POF_E_WORKSPACE = -4 /* caller workspace too small                         */
};

typedef void *pof_stream_t; /* hipStream_t */

int pof_abi_version(void);
const char *pof_error_string(int code);
/* Every entry point starts from a clean HIP error state (so that its own launch check is meaningful).  A sticky
 * error left by an EARLIER call of the calling thread -- the caller's own previous launch, say -- is not
 * swallowed by that: it is parked, and this function returns it (a hipError_t value; 0 = none) and clears the
 * slot.  Thread-local. */
int pof_take_stale_error(void);

/* ------------------------------------------------------------------------
 * A1  get_laser_phi(angle_inc, num_pts)            src/utils/utils.py:25-29
 * Fills tab[0..N) = phi, tab[N..3N) = (cos phi_i, sin phi_i) interleaved, all
 * float64.  phi is bit-identical to numpy.linspace(-fov/2, fov/2, N).
 * ---------------------------------------------------------------------- */
int pof_laser_phi(double angle_inc, int num_pts, double *tab /* [3*N] */, pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A2-A7 fused per-sample preprocessing of the CURRENT scan of each window:
 *   rphi_to_xy                          src/utils/utils.py:47-48
 *   get_displacement_from_odometry      src/utils/utils.py:639-662  (kind 0)
 *   get_flow_target                     src/utils/utils.py:204-229  (kind 1)
 *   get_velocity_from_odometry          src/utils/utils.py:609-636  (kind 2)
 *   data_prepare.get_flow_target        bin/data_prepare.py:29-47   (kind 3; odom0 = odometry
 *                                       difference (dx,dy,dphi), odom1[0] = dt)
 *   scan-pair alignment                 src/utils/dataset.py:76-93  (kind 4; `flow` receives the
 *                                       transformed points; odom0 = (dx,dy,dphi), odom1[0] = scan_dir)
 *   global_to_canonical_flow            src/utils/utils.py:62-75    (canonical != 0)
 *   closest_detection / get_regression_target   src/utils/utils.py:147-185, 232-256
 *   _get_dynamic_mask / _get_valid_point_mask   src/utils/dataset_dr_spaam.py:511-529
 * i.e. the arithmetic of DROWDataset2.__getitem__ (dataset_dr_spaam.py:384-409)
 * for a whole batch in one launch; the collate (dataset_dr_spaam.py:464-471) is
 * implicit because outputs are written in batched layout.
 *
 * ranges        current-scan rows, float32; row b starts at ranges + b*sample_stride
 * tab           output of pof_laser_phi for this N
 * odom0/odom1   [B][3] float64 (x, y, phi); may be NULL when flow and xy are NULL
 * out_f64       0: xy/flow are float32, 1: float64
 * xy, flow      [B][N][2], either may be NULL
 * det_offsets   [B+1] int32 CSR offsets into det_rphi/det_cls, or NULL to skip
 *               association and the dynamic mask
 * det_rphi      [D][2] float64 (r, phi); det_cls [D] uint8 in {0,1,2} = wc, wa, wp
 * assoc_radius  [3] radius per class for closest_detection (reference 0.6,0.4,0.35)
 * labels        [3] class label per class (reference 1,2,3; pedestrian_only: x,x,1)
 * dyn_radius    [3] radius per class for the dynamic mask (reference 2.5,2.0,2.0)
 * closest       [B][N] int64, 1-based detection index within the sample, 0 = none
 * target_cls    [B][N] int64;  target_reg [B][N][2] float32
 * dyn_mask, valid_mask, exclude_mask   [B][N] float32 in {0,1}
 * Any output pointer may be NULL.
 * D             number of rows of det_rphi / det_cls (= det_offsets[B])
 * workspace     device scratch of at least pof_scan_preprocess_workspace_bytes(B, D)
 *               bytes: the per-sample rigid motion and per-detection cartesian
 *               centres (a first, tiny launch) that the streaming launch reads
 *               as wave-uniform scalars.
 * ---------------------------------------------------------------------- */
size_t pof_scan_preprocess_workspace_bytes(int B, int D);
int pof_scan_preprocess(const float *ranges, long long sample_stride, int B, int N,
                        const double *tab, const double *odom0, const double *odom1,
                        int flow_kind, int canonical, int out_f64, void *xy, void *flow,
                        const int32_t *det_offsets, const double *det_rphi, const uint8_t *det_cls,
                        int D, const double *assoc_radius, const int32_t *labels,
                        const double *dyn_radius, int64_t *closest, int64_t *target_cls,
                        float *target_reg, float *dyn_mask, float *valid_mask, float *exclude_mask,
                        void *workspace, size_t workspace_bytes, pof_stream_t stream);

/* Same call split in its two launches, for callers that pipeline independent
 * batches on two streams (params of batch i+1 under the streaming launch of
 * batch i): phases = 1 runs only the per-sample params launch (fills the
 * workspace), phases = 2 only the streaming launch (workspace must have been
 * filled for these inputs), phases = 3 both (= pof_scan_preprocess). */
int pof_scan_preprocess_phase(const float *ranges, long long sample_stride, int B, int N,
                              const double *tab, const double *odom0, const double *odom1,
                              int flow_kind, int canonical, int out_f64, void *xy, void *flow,
                              const int32_t *det_offsets, const double *det_rphi, const uint8_t *det_cls,
                              int D, const double *assoc_radius, const int32_t *labels,
                              const double *dyn_radius, int64_t *closest, int64_t *target_cls,
                              float *target_reg, float *dyn_mask, float *valid_mask, float *exclude_mask,
                              void *workspace, size_t workspace_bytes, int phases, pof_stream_t stream);

/* Chained form for a stream of batches (a data loader knows the next batch):
 * ONE launch streams the current batch -- whose workspace must already hold its
 * params (from the previous chained call, or a phases = 1 call for the first
 * batch) -- and, on extra workgroups of the same grid, evaluates the params of
 * the NEXT batch into next->workspace.  The streaming rows are HBM-bound and
 * leave the ALUs idle, so the next batch's sincos work hides under them and
 * the steady state is one launch per batch.  next == NULL: plain phases = 2. */
typedef struct pof_scan_inputs {
    const double *odom0, *odom1;       /* [B][3], may be NULL when want_flow == 0 */
    const int32_t *det_offsets;        /* [B+1] or NULL */
    const double *det_rphi;            /* [D][2] */
    const uint8_t *det_cls;            /* [D] */
    int32_t B, D, flow_kind, want_flow;
    double assoc_radius[3];
    int32_t labels[3];
    int32_t pad_;
    double dyn_radius[3];
    void *workspace;
    size_t workspace_bytes;
} pof_scan_inputs;

int pof_scan_preprocess_chained(const float *ranges, long long sample_stride, int B, int N,
                                const double *tab, const double *odom0, const double *odom1,
                                int flow_kind, int canonical, int out_f64, void *xy, void *flow,
                                const int32_t *det_offsets, const double *det_rphi, const uint8_t *det_cls,
                                int D, const double *assoc_radius, const int32_t *labels,
                                const double *dyn_radius, int64_t *closest, int64_t *target_cls,
                                float *target_reg, float *dyn_mask, float *valid_mask, float *exclude_mask,
                                void *workspace, size_t workspace_bytes, const pof_scan_inputs *next,
                                pof_stream_t stream);

/* Several batches per launch (round 3).  A data loader that runs ahead hands over up to
 * POF_SCAN_MAX_SLOTS ring slots at once: ONE launch streams the n_cur batches `cur` (their workspaces must hold
 * their params) and evaluates the params of the n_next batches `next` on extra workgroups, as the chained form
 * does for one.  All batches of a call share N, the angle table, flow_kind / canonical / out_f64, the class
 * constants and the SET of non-NULL outputs (POF_E_BADARG otherwise); B may differ.  n_cur == 0: params only.
 * tab_cs_f32: optional [N][2] float32 copy of the table's (cos, sin) pairs, each rounded once from the float64
 * entry (what the float32-output kernel would otherwise convert per point); NULL is allowed.
 * Shapes: N even, N >= 128 (POF_E_SHAPE otherwise: use pof_scan_preprocess_chained), sample_stride even.
 * Replaces the same reference calls as pof_scan_preprocess (dataset_dr_spaam.py:384-409), for a window of
 * consecutive DataLoader batches (dataset_dr_spaam.py:26-28, prefetching workers). */
#define POF_SCAN_MAX_SLOTS 8
typedef struct pof_scan_batch {
    const float *ranges;               /* row b at ranges + b * sample_stride */
    long long sample_stride;
    int32_t B, D;
    const int32_t *det_offsets;        /* [B+1] or NULL */
    void *xy, *flow;                   /* [B][N][2] float32 / float64 (out_f64), or NULL */
    int64_t *closest, *target_cls;     /* [B][N] or NULL */
    float *target_reg;                 /* [B][N][2] or NULL */
    float *dyn_mask, *valid_mask, *exclude_mask;   /* [B][N] or NULL */
    void *workspace;
    size_t workspace_bytes;
} pof_scan_batch;

int pof_scan_preprocess_multi(const pof_scan_batch *cur, int n_cur, const pof_scan_inputs *next, int n_next,
                              int N, const double *tab, const float *tab_cs_f32, int flow_kind, int canonical,
                              int out_f64, const double *assoc_radius, const int32_t *labels,
                              const double *dyn_radius, pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A3 on caller-supplied scanner-frame points (the reference's own signature):
 *   get_displacement_from_odometry(scan1_xy, odom0, odom1)   src/utils/utils.py:639-662
 *   get_velocity_from_odometry(scan1_xy, odom0, odom1)       src/utils/utils.py:609-636
 * xy, flow [B][N][2] float64; odom [B][3]; tab only needed when canonical != 0.
 * ---------------------------------------------------------------------- */
int pof_flow_from_xy(const double *xy, const double *odom0, const double *odom1, int flow_kind,
                     int canonical, const double *tab, double *flow, int B, int N,
                     pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A2 inverse  xy_to_rphi(x, y) -> (hypot, atan2(y, x))   src/utils/utils.py:39-43
 * float64, n elements.
 * ---------------------------------------------------------------------- */
int pof_xy_to_rphi(const double *x, const double *y, double *r, double *phi, long long n,
                   pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A4 stand-alone frame rotation of a flow field
 *   global_to_canonical_flow / canonical_to_global_flow(_torch)
 *                                             src/utils/utils.py:62-105
 * flow_in/out [B][N][2], float32 (is_f64=0) or float64; may alias.
 * ---------------------------------------------------------------------- */
int pof_rotate_flow(const void *flow_in, void *flow_out, const double *tab, int B, int N,
                    int to_canonical, int is_f64, pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A5 global_to_canonical / canonical_to_global   src/utils/utils.py:55-59,109-126
 * Batched over [B][N]; det_* and d* are per point.  float64.
 * ---------------------------------------------------------------------- */
int pof_det_to_canonical(const float *ranges, const double *tab, const double *det_r,
                         const double *det_phi, double *dx, double *dy, int B, int N,
                         pof_stream_t stream);
int pof_canonical_to_det(const float *ranges, const double *tab, const double *dx, const double *dy,
                         double *det_r, double *det_phi, int B, int N, pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A8 scans_to_cutout                            src/utils/utils.py:259-334
 * scans [B][T][N] float32 -> out [B][N/stride][T][P] float32.
 * workspace: int32[B] (per-sample area factor), caller provided.
 * half-angle arctangent: correctly rounded float32.
 * dbg_lo (optional, may be NULL): [B][P][T][N/stride] int32 copy of
 * inds_ct_low for the bit-exact index tests.
 * ---------------------------------------------------------------------- */
int pof_cutout(const float *scans, int B, int T, int N, const double *tab, int stride, int centered,
               int fixed, double window_width, double window_depth, int num_cutout_pts,
               double padding_val, int area_mode, float *out, int32_t *workspace,
               int32_t *dbg_lo, pof_stream_t stream);

/* Same with a value-path selector: value_mode 0 = float64 value path (bit-exact, what
 * pof_cutout runs); 1 = float32 value path: the index math stays float64 and exact
 * (same inds_ct_low / out-of-range / area indices), only lerp, clip and centring run in
 * float32 (|error| <= 1e-5 in the normalised output).  Compare: the reference's own
 * scans_to_cutout_torch does its INDEX math in float32 (src/utils/utils.py:337-420). */
int pof_cutout_ex(const float *scans, int B, int T, int N, const double *tab, int stride, int centered,
                  int fixed, double window_width, double window_depth, int num_cutout_pts,
                  double padding_val, int area_mode, int value_mode, float *out, int32_t *workspace,
                  int32_t *dbg_lo, pof_stream_t stream);

/* BASELINE config 5 storage (reference src/utils/utils.py:259-334 returns float32): the same cutout
 * written as IEEE float16 (the float32 result
 * rounded to nearest even once more), out_f16 [B][ceil(N/stride)][T][P] half.  Halves the
 * dominant write traffic (3600*11*(4 + 56*2) bytes per dense sample). */
int pof_cutout_f16(const float *scans, int B, int T, int N, const double *tab, int stride, int centered,
                   int fixed, double window_width, double window_depth, int num_cutout_pts,
                   double padding_val, int area_mode, int value_mode, void *out_f16, int32_t *workspace,
                   int32_t *dbg_lo, pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A11 nms_predicted_center                      src/utils/utils.py:535-571
 * One scan per batch entry.  pred_cls [B][N] float64 scores, pred_reg [B][N][2].
 * Outputs: det_xy [B][N][2] float64 and det_cls [B][N] float64 compacted to the
 * first num_det[b] rows, instance_mask [B][N] int32.  Scores must be distinct
 * (the reference's argsort is unstable on ties).
 * workspace: at least pof_nms_workspace_bytes(B, N) bytes.
 * ---------------------------------------------------------------------- */
size_t pof_nms_workspace_bytes(int B, int N);
int pof_nms_predicted_center(const float *ranges, const double *tab, const double *pred_cls,
                             const double *pred_reg, double min_dist, int B, int N, double *det_xy,
                             double *det_cls, int32_t *num_det, int32_t *instance_mask,
                             void *workspace, size_t workspace_bytes, pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A12 flow_loss / loss_fn_eval
 *   src/depracted/model/prototype.py:27-32, src/depracted/model/dr_spaam.py:22-27,
 *   src/utils/eval_utils.py:129-134
 * pred/target [B][N][2] float32, mask [B][N] float32 or NULL.
 * epe_sum[b] = sum_i |pred-target| (masked), cnt[b] = number of points counted,
 * aae_sum[b] = sum_i |atan2(p0,p1) - atan2(t0,t1)| (radians); all float64 [B].
 * ---------------------------------------------------------------------- */
int pof_flow_errors(const float *pred, const float *target, const float *mask, int B, int N,
                    double *epe_sum, double *aae_sum, double *cnt, pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A9 Prototype._fusion               src/depracted/model/prototype.py:118-156
 * feat1/feat2 [B][C][n] float32 -> out [B][2*max_disp+1][n] float32.
 * ---------------------------------------------------------------------- */
int pof_band_correlation(const float *feat1, const float *feat2, float *out, int B, int C, int n,
                         int kernel_size, int max_disp, pof_stream_t stream);

/* BASELINE config 5 ("fp16 correlation"; no counterpart in the reference, which is float32
 * throughout prototype.py:118-156): the same with float16 feature storage.  Products and
 * accumulation are float32 (a float16 converts exactly), the output stays float32. */
int pof_band_correlation_f16(const void *feat1_f16, const void *feat2_f16, float *out, int B, int C, int n,
                             int kernel_size, int max_disp, pof_stream_t stream);

/* Backward of pof_band_correlation (training Prototype end to end; in the reference this is torch
 * autograd through the unfold / matmul / gather of src/depracted/model/prototype.py:118-156): given
 * g_out = dL/d out [B][D][n] returns dL/d feat1, dL/d feat2 [B][C][n].  n <= 512. */
int pof_band_correlation_backward(const float *feat1, const float *feat2, const float *g_out,
                                  float *d_feat1, float *d_feat2, int B, int C, int n,
                                  int kernel_size, int max_disp, pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A10 _SpatialAttention.forward (everything after the embedding conv)
 *                                   src/depracted/model/dr_spaam.py:163-217
 * emb_x/emb_t [B][N][E] float32; x/tmpl [B][N][F] float32 (F = channels*pts).
 * band [B][N][w] pre-softmax similarities (clamped duplicates kept),
 * prob [B][N][w] softmax weights with duplicates zeroed (scratch, also useful
 * for the backward pass), out [B][N][F] = alpha*x + (1-alpha)*sum_k prob*tmpl.
 * ---------------------------------------------------------------------- */
int pof_spatial_attention(const float *emb_x, const float *emb_t, const float *x, const float *tmpl,
                          int B, int N, int E, int F, int window, double alpha, float *band,
                          float *prob, float *out, pof_stream_t stream);

/* BASELINE config 5 storage (the reference's dr_spaam.py:163-217 is float32 throughout): the same
 * with the large tensors (x, tmpl, out: [B][N][F]) stored as float16 and float32 arithmetic;
 * the embeddings, band and prob stay float32.  Halves the traffic of the merge kernel. */
int pof_spatial_attention_f16(const float *emb_x, const float *emb_t, const void *x_f16, const void *tmpl_f16,
                              int B, int N, int E, int F, int window, double alpha, float *band, float *prob,
                              void *out_f16, pof_stream_t stream);

/* Backward of pof_spatial_attention (training SpatialDROW through the gate; in the reference torch
 * autograd through src/depracted/model/dr_spaam.py:183-215).
 * g_out = dL/d out [B][N][F]; g_band = dL/d band [B][N][w] or NULL.
 * dsim [B][N][w] is scratch.  Outputs: d_emb_x, d_emb_t [B][N][E]; d_x, d_tmpl [B][N][F]. */
int pof_spatial_attention_backward(const float *emb_x, const float *emb_t, const float *tmpl,
                                   const float *prob, const float *g_out, const float *g_band,
                                   int B, int N, int E, int F, int window, double alpha,
                                   float *dsim, float *d_emb_x, float *d_emb_t, float *d_x,
                                   float *d_tmpl, pof_stream_t stream);

/* The same gradients with the two large passes fused (one walk over g and tmpl: the algorithmic
 * 4 * N * F * 4 bytes instead of reading g and tmpl twice).  workspace:
 * pof_spatial_attention_backward_workspace_bytes(B, N, F, window) bytes of per-column-block partial
 * band products (deterministic: summed in a fixed order by a finishing pass). */
size_t pof_spatial_attention_backward_workspace_bytes(int B, int N, int F, int window);
int pof_spatial_attention_backward_fused(const float *emb_x, const float *emb_t, const float *tmpl,
                                         const float *prob, const float *g_out, const float *g_band,
                                         int B, int N, int E, int F, int window, double alpha,
                                         float *dsim, float *d_emb_x, float *d_emb_t, float *d_x,
                                         float *d_tmpl, void *workspace, size_t workspace_bytes,
                                         pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A13 jump-distance segmentation + per-segment least squares
 *   src/depracted/model/adaboost_person_det.py:71-90 (cuts), :102-210 (features)
 * ranges [B][N] float32.  seg_id [B][N] int32 (segment index of every point),
 * num_seg [B] int32, feat [B][max_seg][16] float64 (columns: see DESIGN.md).
 * ---------------------------------------------------------------------- */
int pof_segment_features(const float *ranges, const double *tab, int B, int N, double jump_dist,
                         int max_seg, int32_t *seg_id, int32_t *num_seg, double *feat,
                         pof_stream_t stream);

/* The reference's own feature rows, Dataset.scan_to_segments + compute_feature
 *   src/depracted/model/adaboost_person_det.py:71-90, :102-210
 * for the segments it keeps (more than two points, :53-55), in its column order and with its data-set
 * coupled definitions: ref_feat [B][max_seg][15] float64 =
 *   0 n, 1 sigma, 2 ||segment - median||_F / n (:127-130), 3 jump to the previous kept segment,
 *   4 jump to kept[min(q+1, 3)] (:133-138; NaN where the reference raises IndexError), 5 width,
 *   6 line residual, 7 circle criterion, 8 radius, 9 boundary length, 10 boundary regularity,
 *   11 summed curvature, 12 mean angular difference,
 *   13 mean((next_ranges - ranges)[piece q of the unfiltered split] / (odom_dt + 1e-3)) (:196-203),
 *   14 label (+1 when the segment centre lies within radius_wp of an annotation, else -1; :84-88).
 * next_ranges [B][N] (NULL: column 13 = NaN), odom_dt [B] = next_odom - odom (NULL: 0),
 * annotations as CSR wp_offsets [B+1] / wp_xy [W][2] (NULL: every label -1), num_kept [B] (may be NULL).
 * feat (the 16-column table of pof_segment_features) and ref_feat may each be NULL, not both. */
int pof_segment_features_ex(const float *ranges, const float *next_ranges, const double *tab, int B, int N,
                            double jump_dist, const double *odom_dt, const int32_t *wp_offsets,
                            const double *wp_xy, double radius_wp, int max_seg, int32_t *seg_id,
                            int32_t *num_seg, int32_t *num_kept, double *feat, double *ref_feat,
                            pof_stream_t stream);

/* ------------------------------------------------------------------------
 * A16 rotate_iou_gpu_eval                     src/utils/rotate_iou.py:297-404
 * boxes [N][5|7], query [K][5|7] float32 (already permuted to the kernel's
 * x,y,l,w,rot,z,h order for 3-D) -> iou [N][K] float32.
 * Batched form: boxes [G][N][s], query [G][K][s], iou [G][N][K], with optional
 * per-group valid counts n_valid[G], k_valid[G] (NULL = all).
 * ---------------------------------------------------------------------- */
int pof_rotate_iou(const float *boxes, const float *query, float *iou, int G, int N, int K,
                   const int32_t *n_valid, const int32_t *k_valid, int criterion, int is_3d,
                   pof_stream_t stream);

/* ------------------------------------------------------------------------
 * N1: device-resident scan store -> batch of windows.
 *   DROWDataset2.__getitem__ window gather     src/utils/dataset_dr_spaam.py:357-366
 *   scan <-> odometry time association          src/utils/dataset_dr_spaam.py:370-378
 * scans_all [S_total][N] float32: all sequences concatenated.  Per sample b:
 * seq_first[b] = global row of its sequence's first scan, scan_idx[b] = index of the
 * current scan inside the sequence.  out [B][num_scans+1][N]: rows
 * max(0, scan_idx - (num_scans+distance-1-j)*stride), j < num_scans, then scan_idx.
 * row_cur / row_prev [B]: global rows of the current scan and of the last template
 * row (whose time stamps select odom1 / odom0).
 * ---------------------------------------------------------------------- */
int pof_gather_windows(const float *scans_all, const int32_t *seq_first, const int32_t *scan_idx,
                       int B, int num_scans, int distance, int stride, int N, float *out,
                       int32_t *row_cur, int32_t *row_prev, pof_stream_t stream);

/* Scan <-> odometry time association of __getitem__ (src/utils/dataset_dr_spaam.py:369-378):
 * odom{0,1}[b] = odoms[argmin_k |odoms_t[k] - scans_t[row_{prev,cur}[b]]|], k in
 * [odom_lo[b], odom_hi[b]) (the sample's sequence), float32 differences, first
 * minimum wins (np.argmin).  odoms [O_total][3] float32 -> odom0/odom1 [B][3] float64;
 * idx0/idx1 (optional) receive the indices relative to odom_lo. */
int pof_associate_odometry(const float *scans_t, const float *odoms_t, const float *odoms,
                           const int32_t *odom_lo, const int32_t *odom_hi, const int32_t *row_cur,
                           const int32_t *row_prev, int B, double *odom0, double *odom1,
                           int32_t *idx0, int32_t *idx1, pof_stream_t stream);

/* ----------------------------------------------------------------------
 * N2 DROW / DR-SPAAM trunk layer, inference     src/depracted/model/dr_spaam.py:8-19, :86-92
 * y = max_pool1d?(LeakyReLU(BatchNorm_eval(Conv1d(k=3, pad=1)(x))), 2) on S sequences:
 * x [S][Ci][L] float32 -> out [S][Co][pool ? L/2 : L].  wt = conv weight transposed to
 * [3][Ci][Co]; scale[Co] = gamma / sqrt(running_var + eps), shift[Co] = beta +
 * (conv_bias - running_mean) * scale (the caller folds them once per checkpoint).
 * Implicit GEMM on the float32 MFMA (exact float32 products, k-ordered accumulation).
 * ---------------------------------------------------------------------- */
int pof_conv3_bn_lrelu(const float *x, const float *wt, const float *scale, const float *shift,
                       int S, int Ci, int Co, int L, int pool, double negative_slope, float *out,
                       pof_stream_t stream);

/* The same layer with kernel_size 1 or 3 (padding kernel_size / 2) and stride 1 or 2 (round 3): the units of the
 * Prototype flow network -- Conv1d(k = 3, stride 2 | 1) / Conv1d(k = 1) + BatchNorm(eval) + LeakyReLU --
 * src/depracted/model/prototype.py:6-25, 38-45.  wt [kernel_size][Ci][Co]; out [S][Co][Lc] with
 * Lc = L (stride 1) or (L + 1) / 2 (stride 2), halved again when pool != 0 (stride 1 only, Lc even).
 * Supported: (3, 1), (3, 2), (1, 1); anything else returns POF_E_SHAPE. */
int pof_conv1d_bn_lrelu(const float *x, const float *wt, const float *scale, const float *shift,
                        int S, int Ci, int Co, int L, int kernel_size, int stride, int pool,
                        double negative_slope, float *out, pof_stream_t stream);

/* ----------------------------------------------------------------------
 * N2 detector heads, inference                  src/depracted/model/dr_spaam.py:104-121
 * pred_cls[s][o] = b_cls[o] + sum_c w_cls[o][c] * mean_l feat[s][c][l]   (o < n_cls <= 6)
 * pred_reg[s][o] = b_reg[o] + sum_c w_reg[o][c] * mean_l feat[s][c][l]   (o < 2)
 * -- the average pool over the last block's positions and the two 1x1 convolutions
 * (conv_cls [n_cls][C][1], conv_reg [2][C][1]) in one launch.  feat [S][C][L] float32.
 * ---------------------------------------------------------------------- */
int pof_drow_heads(const float *feat, int S, int C, int L, const float *w_cls, const float *b_cls,
                   int n_cls, const float *w_reg, const float *b_reg, float *pred_cls,
                   float *pred_reg, pof_stream_t stream);

/* ----------------------------------------------------------------------
 * N2 trunk unit tail, training                  src/depracted/model/dr_spaam.py:8-19, :86-92
 * z = max_pool1d?(LeakyReLU(BatchNorm1d_train(y)), 2) and its backward pass, for the
 * convolution output y [S][C][L] float32 (L <= 256, C*L % 4 == 0, L even when pooled):
 * out [S][C][pool ? L/2 : L].  Batch statistics over (S, L) in float64; running_mean /
 * running_var (may be NULL) are updated with `momentum` as torch.nn.BatchNorm1d does
 * (unbiased variance); save_mean / save_invstd [C] are what the backward pass needs
 * besides y.  backward: dz [S][C][L or L/2] -> dy [S][C][L], dgamma [C], dbeta [C]; the
 * pool routes a gradient to the first maximum of its pair (torch.max_pool1d).
 * dbias_in [C] (may be NULL) = sum of dy over (S, L): the gradient of a per-channel bias
 * added in front of the BatchNorm (the convolution's), from the same pass that writes dy.
 * groups >= 1 (S % groups == 0): the sequences form `groups` equal contiguous ranges, each
 * normalised with its OWN batch statistics -- the five scans of a DR-SPAAM window, which the
 * reference sends through the trunk one after the other (dr_spaam.py:246-262), in one launch;
 * save_mean / save_invstd are then [groups][C] and the running statistics receive the groups'
 * updates in order, as `groups` separate calls would give.
 * workspace: pof_bn_lrelu_pool_workspace_bytes(S, C, L, groups) bytes (0 = unsupported shape).
 * ---------------------------------------------------------------------- */
size_t pof_bn_lrelu_pool_workspace_bytes(long long S, int C, int L, int groups);
int pof_bn_lrelu_pool_forward(const float *y, long long S, int C, int L, int groups, const float *gamma,
                              const float *beta, float *running_mean, float *running_var,
                              double momentum, double eps, double negative_slope, int pool,
                              float *out, float *save_mean, float *save_invstd, void *workspace,
                              size_t workspace_bytes, pof_stream_t stream);
int pof_bn_lrelu_pool_backward(const float *y, const float *dz, long long S, int C, int L, int groups,
                               const float *gamma, const float *beta, const float *save_mean,
                               const float *save_invstd, double negative_slope, int pool, float *dy,
                               float *dgamma, float *dbeta, float *dbias_in, void *workspace,
                               size_t workspace_bytes, pof_stream_t stream);

/* The trunk's first TWO units in one launch (inference): x [S][L] float32 is the single-channel
 * cutout (dr_spaam.py:86-92, conv_block_1[0] and [1]); the C1 channels of the first unit,
 * lrelu_slope1(a0 x[q-1] + a1 x[q] + a2 x[q+1] + b) with l1[c] = {a0, a1, a2, b} (its taps times its
 * folded BatchNorm scale, and its shift; zero padding at the sequence borders), are computed inside
 * the second unit's kernel instead of being written and read back (1 GB each way at B = 32).  wt
 * [3][C1][Co], scale / shift [Co], pool, negative_slope: the second unit, as in pof_conv3_bn_lrelu.
 * C1 <= 128.  The first unit's sums are FMA chains here and MFMA accumulations in the two-launch
 * form: results agree to float32 round-off, not bit for bit. */
int pof_conv3_first_two(const float *x, const float *l1, double slope1, const float *wt, const float *scale,
                        const float *shift, int S, int C1, int Co, int L, int pool, double negative_slope,
                        float *out, pof_stream_t stream);

/* ----------------------------------------------------------------------
 * N2 trunk convolution, weight gradient         src/depracted/model/dr_spaam.py:8-19
 * dw[co][ci][t] = sum_{s,l} dy[s][co][l] * x[s][ci][l + t - 1]  (Conv1d k = 3, pad = 1):
 * x [S][Ci][L], dy [S][Co][L] float32 -> dw [Co][Ci][3] float32 (torch's weight layout).
 * Split-K float32-MFMA GEMM over (sequence, position), deterministic (partial tiles in
 * the workspace, one reduction pass).  The forward and the data gradient of the same
 * convolution are pof_conv3_bn_lrelu with unit scale / slope 1 (data gradient: dy as
 * input, taps reversed, channel roles swapped).
 * workspace: pof_conv3_wgrad_workspace_bytes(S, Ci, Co, L) bytes (0 = unsupported shape: rows longer
 * than 64 positions, or odd rows longer than 32 -- callers keep their library path for those).
 * ---------------------------------------------------------------------- */
size_t pof_conv3_wgrad_workspace_bytes(int S, int Ci, int Co, int L);
int pof_conv3_wgrad(const float *x, const float *dy, int S, int Ci, int Co, int L, float *dw,
                    void *workspace, size_t workspace_bytes, pof_stream_t stream);
/* The same pass for kernel_size 1 | 3 (1: the point-wise convolutions of the box-regression PointNet,
 * src/model/box_regression.py:8-17, and of the Prototype head, src/depracted/model/prototype.py:52-58):
 * dw [Co][Ci][kernel_size]; kernel_size 3 is pof_conv3_wgrad. */
size_t pof_conv1d_wgrad_workspace_bytes(int S, int Ci, int Co, int L, int kernel_size);
int pof_conv1d_wgrad(const float *x, const float *dy, int S, int Ci, int Co, int L, int kernel_size, float *dw,
                     void *workspace, size_t workspace_bytes, pof_stream_t stream);

/* ----------------------------------------------------------------------
 * configs[3] box-regression head, dense layers    src/model/box_regression.py:26-45 (_fc), :139-141
 * out[b][n] = sum_k x[b][k] * w[n][k] + bias[n] (torch.nn.Linear's forward; x [B][K], w [N][K],
 * bias [N] or NULL, out [B][N], float32, K a multiple of 4, x and w 16-byte aligned).  For the
 * head's batch (a few hundred rows): one workgroup per 32 x 32 output tile, K split over its
 * four waves, float32 MFMA, deterministic.  The backward GEMMs stay with the BLAS library.
 * ---------------------------------------------------------------------- */
int pof_linear_bias(const float *x, const float *w, const float *bias, int B, int K, int N, float *out,
                    pof_stream_t stream);

/* ----------------------------------------------------------------------
 * configs[3] box-regression loss                   src/model/box_regression.py:52-67 (regression_loss2)
 * pred, target [B][T] float32, T = 3 (dims, dims, orientation) or 5 (z, 3 dims, orientation):
 * loss[0] = mean_b sum_j c_j |pred - target|, c_j = 1 except c_{T-1} = alpha; dpred [B][T] (or
 * NULL) = d loss / d pred = c_j sign(pred - target) / B.  One launch instead of the ~27 the
 * composed form takes forward and backward.
 * ---------------------------------------------------------------------- */
int pof_regression_loss2(const float *pred, const float *target, long long B, int T, double alpha, float *loss,
                         float *dpred, pof_stream_t stream);

/* ----------------------------------------------------------------------
 * N3 BoxRegressor input preparation, batched     box_regressor.py:43-75, :94-105
 *                                                src/data_handle/jrdb_handle.py:178-256
 * points [Np][D] float64 (D = 2 or 3), centers [S][D], oris [S] -> per detection the
 * radius query norm(points - centre) <= radius (float64), count[S] = segment size,
 * and x [S][input_size][D+1] float32 = the reference's fixed-size resampling
 * (random subset when larger, repeat + pad when smaller) of (point - centre, ori).
 * Segments with fewer than min_segment_size points get zero rows (the caller skips
 * them, as the reference returns None).  Randomness: a counter-based hash of
 * (seed, detection index, point index); rows come out in hash order (the consumer is
 * order invariant; the reference's order depends on the global NumPy RNG).
 * mask (optional, [S][Np] uint8) receives the radius-query result itself
 * (generate_segment / anns_to_segments return the variable-size segment).
 * ---------------------------------------------------------------------- */
int pof_segment_inputs(const double *points, int Np, int D, const double *centers, const double *oris,
                       int S, double radius, int input_size, int min_segment_size, uint32_t seed,
                       float *x, int32_t *count, uint8_t *mask, pof_stream_t stream);

/* Training-side twin (src/data_handle/jrdb_dataset.py:99-156): segments already cut (CSR
 * seg_offsets [S+1] over points [P][D]); per sample: subtract centers[s], optionally append
 * extra[s] as a column (the random input angle), optionally drop int(n * random_drop) random
 * points first, then the same fixed-size resampling -> x [S][input_size][D + (extra ? 1 : 0)],
 * count[s] = points kept.  max_segment = longest segment (<= 4096). */
int pof_segment_resample(const double *points, int D, const int32_t *seg_offsets, int S, int max_segment,
                         const double *centers, const double *extra, double random_drop, int input_size,
                         uint32_t seed, float *x, int32_t *count, pof_stream_t stream);

/* ----------------------------------------------------------------------
 * N4 scans_to_polar_grid                        src/utils/utils.py:492-531
 * scans [B][T][N] float32 -> out [B][T][R][N] float32, R = int((max-min)/bin) + 1:
 * the truncated-signed-distance column of every beam (the "fc2d" network input,
 * src/utils/dataset_dr_spaam.py:455-458).  float32 arithmetic as NumPy >= 2 evaluates
 * the reference (bit-exact against it); tsdf_clip <= 0 disables the distance ramp.
 * ---------------------------------------------------------------------- */
int pof_polar_grid(const float *scans, int B, int T, int N, double min_range, double max_range,
                   double range_bin_size, double tsdf_clip, int normalize, float *out,
                   pof_stream_t stream);

/* ----------------------------------------------------------------------
 * N1, host side: numeric CSV -> float64 matrix (no device work, no stream).
 * Replaces np.genfromtxt(path, delimiter=",") for the DROW sequence files
 * (src/utils/dataset_dr_spaam.py:473-478 `.csv`, :504-509 `.odom2`, :497-502
 * `.difodom`; bin/data_prepare.py:70-72 `.flow`).  Fields are converted with
 * strtod (correctly rounded, = Python float()); blank lines and `#` lines are
 * skipped, empty / unparsable fields become NaN.  pof_csv_shape reports the
 * row count and the column count of the first row; pof_csv_read_f64 fills a
 * caller-owned [rows][cols] buffer (POF_E_SHAPE if the file does not have
 * exactly that shape on every row); threads <= 0: one per hardware thread (max 16).
 * ---------------------------------------------------------------------- */
int pof_csv_shape(const char *path, long long *rows, int *cols);
int pof_csv_read_f64(const char *path, long long rows, int cols, double *out, int threads);

/* ----------------------------------------------------------------------
 * N4: boosted decision stumps of the legacy person-detection baseline.
 * pof_stump_search replaces BoostedFeatureDetector.simple_classifier
 * (src/depracted/model/adaboost_person_det.py:283-347) for all D feature
 * dimensions in one launch: X [rows][D] float64, Y [rows] (+1 / -1), the n
 * samples are rows index[0..n) (index NULL: rows 0..n-1; 2 <= n <= 2048,
 * duplicates allowed -- the boosting loop samples with replacement).  Per
 * dimension d: n_thresh[d] = number of threshold candidates (midpoints of
 * sorted neighbours of opposite class), min_err[d] / max_err[d] = smallest /
 * largest count of samples misclassified by "x > theta -> +1" over the
 * candidates, theta_min[d] / theta_max[d] = the FIRST candidate (ascending)
 * reaching it (-1 / 0.0 when there is no candidate).  Equal feature values
 * keep their sample order (the reference's np.argsort leaves that order to the
 * sort implementation).  X must be finite.
 * pof_stump_vote replaces BoostedFeatureDetector.eval (:349-378): result[i] =
 * sum_k alpha[k] * (X[i][dim[k]-1] > theta[k] ? +1 : -1) accumulated in k order
 * in float64, label[i] = sign(result[i]) (label may be NULL); dim is 1-based,
 * 0 addresses the last column (an unused round of the reference's K x 2
 * parameter table).
 * ---------------------------------------------------------------------- */
int pof_stump_search(const double *X, const double *Y, long long rows, const int *index, int n, int D,
                     int *min_err, double *theta_min, int *max_err, double *theta_max, int *n_thresh,
                     pof_stream_t stream);
int pof_stump_vote(const double *X, long long N, int D, const int *dim, const double *theta,
                   const double *alpha, int K, double *result, double *label, pof_stream_t stream);

/* ----------------------------------------------------------------------
 * N4, host side: LZF block decoder for `DATA binary_compressed` .pcd files.
 * Replaces lzf.decompress(compressed_data, uncompressed_size) in the vendored
 * pypcd (src/data_handle/_pypcd.py:249-264), which JRDBHandle._load_pointcloud
 * (src/data_handle/jrdb_handle.py:293-305) goes through.  Returns the number of
 * bytes written to out, or -1 when the stream is malformed, refers before the
 * start of the output or would exceed out_cap (nothing outside [out, out+out_cap)
 * is touched).
 * ---------------------------------------------------------------------- */
long long pof_lzf_decompress(const void *in, long long in_len, void *out, long long out_cap);

#ifdef __cplusplus
}
#endif
#endif /* POF_ABI_H */
