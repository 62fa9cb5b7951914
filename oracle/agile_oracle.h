/*
 * agile_oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a dependency-free CPU restatement of the
 * agile_grasp per-sample hot path (reference files cited per function in
 * agile_oracle.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (agile_grasp_amd/, include/) never
 * includes, links or calls anything in oracle/.
 *
 * PARITY PINNING: the reference ships no golden vectors for this path
 *  and cannot be built here (PCL/Eigen/LAPACK/OpenCV are
 * absent).  The oracle is pinned by (1) LAPACK dggev goldens generated with
 * scipy in this container (tests/golden/), (2) an independent numpy
 * transcription of the finger/hand logic, (3) structural HOG known-answer
 * tests.  At the third-party seams (FLANN order, dggev, Eigen::EigenSolver,
 * OpenCV HOG/SVM float order) parity with a 2015 reference binary is
 * UNPINNED; the interpretation chosen is stated at each function.
 */
#ifndef AGILE_ORACLE_H
#define AGILE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NORMALS_DETERMINISTIC 0 /* Quadric(is_deterministic=true): all neighbours */
#define ORC_NORMALS_RAND50 1        /* HandSearch default: 50 draws of glibc rand() % n, single thread */

typedef struct orc_params
{
  double finger_width;        /* find_grasps.cpp:13 */
  double hand_outer_diameter; /* find_grasps.cpp:14 */
  double hand_depth;          /* find_grasps.cpp:15 */
  double hand_height;         /* find_grasps.cpp:17 */
  double init_bite;           /* find_grasps.cpp:16 */
  double nn_radius_taubin;    /* hand_search.h:85  (0.03) */
  double nn_radius_hands;     /* hand_search.h:85  (0.08) */
  double nn_radius_normals;   /* hand_search.cpp:20 (0.01), all-points pass */
  double cam_origin[2][3];    /* translation of cam_tf_left_/right_ (hand_search.cpp:72-74) */
  int32_t normals_mode;       /* ORC_NORMALS_* */
  uint32_t rand_seed;         /* srand() seed for ORC_NORMALS_RAND50 (glibc default 1) */
  int32_t num_threads;        /* OpenMP threads (hand_search.cpp:78,136) */
  int32_t pow6_libm;          /* 1: std::pow(v,6.0) as Eigen .pow(6) does; 0: ((v*v)*(v*v))*(v*v) */
} orc_params;

/* Same layout as agh_hypothesis (include/agh.h). */
typedef struct orc_hypothesis
{
  double axis[3];
  double approach[3];
  double binormal[3];
  double bottom[3];
  double surface[3];
  double width;
  int32_t sample;      /* position in the sample-index list */
  int32_t orientation; /* 0..7 */
  int32_t cam_source;
  int32_t n_in_box;
  uint8_t half_antipodal, full_antipodal, svm_keep, valid;
  int32_t finger_index; /* eroded hand index e (finger_hand.cpp:190) */
  int32_t depth_index;  /* number of successful deepen steps (finger_hand.cpp:204-225) */
  int32_t epoch;  /* the product stamps its calls here; the oracle leaves 0 */
} orc_hypothesis;

typedef struct orc_frame
{
  double sample[3];
  double normal[3];
  double axis[3];     /* curvature axis (quadric.cpp:304) */
  double binormal[3];
  double params[10];  /* quadric parameters after the 0.5 scaling (quadric.cpp:152-153) */
  double eigenvalue;  /* smallest finite generalized eigenvalue */
  int32_t n_nb;
  int32_t majority_cam;
  int32_t max_index;  /* argmax column of quadric.cpp:283-284 */
  int32_t valid;      /* 0 only for an empty neighbourhood (rank-deficient pencils are deflated, not dropped) */
} orc_frame;

/* a2: exact radius search; output sorted ascending by (float d2, index). Returns count (may exceed cap). */
int64_t orc_radius_search(const float* xyz, int64_t stride_floats, int64_t n, const float q[3], double radius,
  int32_t* idx_out, float* d2_out, int64_t cap);

/* a3-a6: fit frames for the listed samples with the given radius. */
int orc_fit_frames(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, double radius, orc_frame* frames_out);

/* a1: the whole HandSearch::findHands.  images_out (optional) receives cap*8000 bytes (80x100 grasp images,
 * learning.cpp:320-365, cam_pos = cam_origin), nh_out (optional) the r=nn_radius_hands neighbour counts,
 * frames_out (optional) the per-sample frames.  Hypotheses are written sample-major, orientation-ascending. */
int orc_find_hands(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal, orc_hypothesis* out, int64_t cap,
  int64_t* n_out, orc_frame* frames_out, int32_t* nh_out, uint8_t* images_out);

/* a7-a13 only: hand search with frames supplied by the caller (stage-wise parity).
 * normals (optional, 3*n doubles, per cloud point) feed the antipodal test. */
int orc_hands_from_frames(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam,
  int64_t n, const int32_t* sample_idx, int64_t n_samples, const orc_frame* frames, const double* normals,
  orc_hypothesis* out, int64_t cap, int64_t* n_out, int32_t* nh_out, uint8_t* images_out);

/* a18: OpenCV-2.4 HOGDescriptor(winSize 64x64).compute(image 80 rows x 100 cols, winStride 32x32) -> 3528 floats */
int orc_hog(const uint8_t* image, float* desc_out);

/* a19: CvSVM::predict for the linear 1-SV model: returns 1 if the reference keeps the hand. */
int orc_svm_keep(const float* desc, const float* weights, int32_t n_w, double rho, double* sum_out);

/* a15: images (H x 8000) -> keep flags (and optionally the decision sums). */
int orc_classify(const uint8_t* images, int64_t n_hyp, const float* weights, int32_t n_w, double rho,
  uint8_t* keep_out, double* sum_out, int num_threads);

/* a20: parse the OpenCV YAML linear SVM (one support vector). Returns n_w or <0. */
int orc_load_svm(const char* path, float* weights_out, int32_t cap, double* rho_out);

/* glibc rand() restatement (for ORC_NORMALS_RAND50 tests). Fills out[0..count). */
void orc_glibc_rand(uint32_t seed, int32_t* out, int64_t count);

/* generalized eigen reduction exposed for the LAPACK goldens: M,N 10x10 row-major; v_out 10. */
int orc_solve_taubin(const double* M, const double* N, double* v_out, double* lambda_out);
/* The unit eigenvector of the smallest eigenvalue of a symmetric 3 x 3 (row-major), as fit_frame uses it (quadric.cpp:268-280). */
void orc_smallest_eigvec3(const double* M3, double* axis_out);

/* f2: HandleSearch::findHandles + Handle (handle_search.cpp:4-128, handle.cpp:3-74). */
typedef struct orc_handle
{
  double axis[3];         /* eigenvector of the largest eigenvalue of sum(axis axis^T) (handle.cpp:12-26), sign made
                             to agree with the first inlier's axis (Eigen::EigenSolver's sign is arbitrary) */
  double center[3];       /* grasp bottom of the inlier nearest the middle of the handle (handle.cpp:39-56) */
  double approach[3];
  double binormal[3];     /* approach x axis */
  double hands_center[3]; /* grasp surface of that inlier */
  double width;           /* mean grasp width of the inliers (handle.cpp:66-74) */
  int32_t n_inliers;
  int32_t first_inlier;   /* offset of this handle's inlier list in the index array */
} orc_handle;

/* Returns the number of handles (<0 on error); inlier_idx receives the concatenated inlier lists (indices into hands,
 * in the order handle.cpp sees them: ascending distance along the seed hand's axis). */
int64_t orc_find_handles(const orc_hypothesis* hands, int64_t n_hands, int32_t min_inliers, double min_length,
  orc_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx, int64_t idx_cap);

/* f1: NaN removal + workspace box + per-camera voxelisation (localization.cpp:17-45,216-355).  xyz_out 3*cap floats,
 * cam_out cap ints; returns the number of voxels (may exceed cap: then only cap are written). */
int64_t orc_preprocess(const float* xyz, int64_t stride_floats, int64_t n, int64_t size_left, int dense,
  const double workspace[6], double cell_size, float* xyz_out, int32_t* cam_out, int64_t cap);

/* a14: points_for_learning + the camera id of each column for every hypothesis of a (non-antipodal) search
 * (rotating_hand.cpp:125-151); ofs_out has n_out + 1 entries.  Returns the total column count. */
int64_t orc_find_hands_points(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, orc_hypothesis* out, int64_t cap, int64_t* n_out, double* pts_out,
  int32_t* cam_out, int64_t pts_cap, int64_t* ofs_out);

/* f4 (training side): orc_find_hands(calculates_antipodal = 1) that also returns, per hypothesis, the images of
 * createInstance(h, cam_pos, cam = 0) and (cam = 1) (learning.cpp:389-397): cam_images_out is cap x 2 x 8000 bytes. */
int orc_find_hands_training(const orc_params* p, const float* xyz, int64_t stride_floats, const int32_t* cam, int64_t n,
  const int32_t* sample_idx, int64_t n_samples, orc_hypothesis* out, int64_t cap, int64_t* n_out, uint8_t* images_out,
  uint8_t* cam_images_out);

/* f4: CvSVM::train(C_SVC, LINEAR) + optimize_linear_svm as Learning::convertData runs it (learning.cpp:296-313).
 * features n x var_count row-major, labels > 0 = positive.  OpenCV's solver restated; parity with OpenCV UNPINNED
 * (see the .cpp).  info_out = {iterations, n_sv, n_class0, n_class1}.  Returns 0, -3 if a class is missing. */
int orc_train_svm(const float* features, const int8_t* labels, int64_t n, int32_t var_count, int32_t kernel /* 0 LINEAR,
  1 POLY degree 2 (convertData's uses_linear_kernel = false) */, double C, int32_t max_iter, double eps,
  float* weights_out /* LINEAR: the compacted vector; may be NULL */, double* rho_out, int32_t* info_out,
  double* alpha_out, int32_t* sv_order_out, int num_threads);

/* f4: CvSVM::save / load / predict for both model shapes (one compacted vector, or n_sv support vectors + alphas). */
int orc_save_svm_model(const char* path, int32_t kernel, const float* sv, int32_t n_sv, int32_t n_w, const double* alpha,
  double rho);
int orc_load_svm_model(const char* path, int32_t* kernel_out, float* sv_out, int32_t sv_cap, int32_t n_w,
  double* alpha_out, double* rho_out);
int orc_classify_model(const uint8_t* images, int64_t n_hyp, int32_t kernel, const float* sv, int32_t n_sv, int32_t n_w,
  const double* alpha, double rho, uint8_t* keep_out, double* sum_out, int num_threads);

/* f4: CvSVM::save of the compacted linear model (pinned: regenerates the shipped model file byte for byte). */
int orc_save_svm(const char* path, const float* weights, int32_t n_w, double rho);

#ifdef __cplusplus
}
#endif
#endif
