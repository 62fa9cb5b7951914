/*
 * agh.h -- C ABI of libagile_grasp_hip.so, the MI355X-native (gfx950) grasp-hypothesis search.
 *
 * This is the drop-in boundary for agile_grasp's per-sample hot path.  The reference has no FFI; the seam is
 * a set of C++ methods (paths relative to the reference repository):
 *
 *   agh_create / agh_destroy        <- HandSearch::HandSearch(...)               include/agile_grasp/hand_search.h:77-85
 *   agh_set_cloud[_device]          <- kd-tree build inside HandSearch::findHands src/agile_grasp/hand_search.cpp:10-11
 *   agh_find_hands[_device]         <- HandSearch::findHands(cloud, pts_cam_source, indices, ...)
 *                                                                                 include/agile_grasp/hand_search.h:101-104,
 *                                                                                 src/agile_grasp/hand_search.cpp:4-62
 *   agh_load_svm[_file]             <- CvSVM::load in Learning::classify          src/agile_grasp/learning.cpp:185
 *   agh_classify                    <- Learning::classify(hands, svm, cam_pos)    include/agile_grasp/learning.h:122-123,
 *                                                                                 src/agile_grasp/learning.cpp:165-247
 *   agh_hypothesis                  <- GraspHypothesis                            include/agile_grasp/grasp_hypothesis.h:46-231
 *
 * The header-only C++ adapter in include/agile_grasp_amd/ keeps the reference's class and method names on top of
 * this ABI.  Conventions: every function returns AGH_OK (0) or a negative agh_status; no
 * exception crosses the ABI; all buffers are caller-owned; one context per host thread; a context owns its HIP
 * stream unless a stream is passed in.  There is NO CPU fallback: without a usable HIP device agh_create fails.
 */
#ifndef AGH_H
#define AGH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGH_VERSION 1

typedef enum agh_status
{
  AGH_OK = 0,
  AGH_ERR_INVALID_ARGUMENT = -1,
  AGH_ERR_NO_DEVICE = -2,       /* no HIP device / wrong architecture: the product path never falls back to the CPU */
  AGH_ERR_HIP = -3,             /* a HIP runtime call failed; see agh_last_error */
  AGH_ERR_CAPACITY = -4,        /* output buffer too small, or a neighbourhood exceeds the kernels' LDS capacity */
  AGH_ERR_NO_CLOUD = -5,
  AGH_ERR_NO_SVM = -6,
  AGH_ERR_IO = -7,
  AGH_ERR_STATE = -8,
  AGH_ERR_RETRY = -9            /* the call's result is incomplete because the context ran in a cheaper configuration than
                                 * this input needs; it has switched itself (for good): repeat the call.  Two cases: a Taubin
                                 * neighbourhood beyond the first capacity class of the kernels (the launches of the larger
                                 * classes are skipped until a cloud needs them), and a rank that overflowed its exchange
                                 * segment in a sharded search (both conditions reach every rank through the segment
                                 * headers, so all ranks repeat together).  Host-buffer entry points repeat by themselves; the
                                 * asynchronous device variants report it at the next agh_synchronize. */
} agh_status;

#define AGH_NORMALS_DETERMINISTIC 0 /* Quadric(is_deterministic = true): all neighbours (quadric.cpp:194-212) */
#define AGH_NORMALS_RAND50 1        /* HandSearch default: 50 draws of glibc rand() % n in sample order (quadric.cpp:177-193) */

typedef struct agh_params
{
  double finger_width;        /* find_grasps.cpp:13 */
  double hand_outer_diameter; /* find_grasps.cpp:14 */
  double hand_depth;          /* find_grasps.cpp:15 */
  double hand_height;         /* find_grasps.cpp:17 */
  double init_bite;           /* find_grasps.cpp:16 */
  double nn_radius_taubin;    /* hand_search.h:85 (0.03) */
  double nn_radius_hands;     /* hand_search.h:85 (0.08) */
  double nn_radius_normals;   /* hand_search.cpp:20 (0.01) */
  double cam_origin[2][3];    /* translations of cam_tf_left / cam_tf_right (hand_search.cpp:72-74) */
  int32_t normals_mode;       /* AGH_NORMALS_* */
  uint32_t rand_seed;         /* srand() seed for AGH_NORMALS_RAND50 */
  int32_t device;             /* HIP device ordinal */
  int32_t profile;            /* 1: time every kernel with HIP events (agh_get_timing); 2: only k_hand_sweep (start / stop events
                                 attached to its dispatch); 3: the same on every fourth call (a timed launch costs ~6 us) */
} agh_params;

/* One grasp hypothesis, fixed size (160 B).  The variable-size points_for_learning_ of the reference
 * (grasp_hypothesis.h:220) stays on the device as an 80x100 occupancy image consumed by agh_classify. */
typedef struct agh_hypothesis
{
  double axis[3];      /* getAxis() */
  double approach[3];  /* getApproach() */
  double binormal[3];  /* getBinormal() */
  double bottom[3];    /* getGraspBottom() */
  double surface[3];   /* getGraspSurface() */
  double width;        /* getGraspWidth() */
  int32_t sample;      /* position in the sample-index list */
  int32_t orientation; /* 0..7, angle = -pi + k*pi/4 (rotating_hand.cpp:13-15) */
  int32_t cam_source;  /* getCamSource() */
  int32_t n_in_box;    /* columns of points_for_learning_ */
  uint8_t half_antipodal, full_antipodal; /* isHalfAntipodal(), isFullAntipodal() */
  uint8_t svm_keep;    /* set by agh_classify: 1 iff CvSVM::predict == 1 (learning.cpp:225-227) */
  uint8_t valid;
  int32_t finger_index; /* eroded hand index (finger_hand.cpp:190) */
  int32_t depth_index;  /* successful deepen steps (finger_hand.cpp:204-225) */
  int32_t epoch;        /* stamp of the agh_find_hands* call that produced the record (process-wide counter, never 0):
                           the device-side state behind agh_classify / agh_get_learning_points / agh_get_packed_images
                           belongs to ONE call, and a record with another stamp must not be matched against it */
} agh_hypothesis;

/* Per-sample local frame (Quadric's results: quadric.h getters) -- for stage-wise parity tests and plotting. */
typedef struct agh_frame
{
  double sample[3];
  double normal[3];
  double axis[3];
  double binormal[3];
  double params[10];
  double eigenvalue;
  int32_t n_nb;
  int32_t majority_cam;
  int32_t max_index;
  int32_t valid;
} agh_frame;

/* Kernel times of the last find_hands call, milliseconds, measured with HIP events on the context's stream. */
#define AGH_TIMING_SLOTS 16
typedef struct agh_timing
{
  float ms[AGH_TIMING_SLOTS];
  const char* name[AGH_TIMING_SLOTS];
  int32_t n;
  float total_ms;
} agh_timing; /* (this layout is frozen: a caller built against an earlier header passes a buffer of exactly this size;
                 what was added later has its own getter, agh_get_timing_counts) */

typedef struct agh_ctx agh_ctx;

void agh_default_params(agh_params* p);
int agh_create(const agh_params* p, agh_ctx** out);
void agh_destroy(agh_ctx* ctx);
const char* agh_last_error(const agh_ctx* ctx); /* ctx may be NULL: error of the last failed agh_create */

/* Upload (host pointers) or adopt (device pointers) a cloud and build the uniform search grid on the GPU.
 * xyz: x,y,z float32 at byte offset 0 of each point; stride_bytes = 12 (packed) or 32 (pcl::PointXYZRGBA).
 * cam_source: 0/1 per point (Eigen::VectorXi pts_cam_source), may be NULL (all 0).
 * The host variant returns when xyz and cam_source have been READ (the caller may free or overwrite them at once); the grid
 * build is then still queued on the context's stream, in front of whatever uses the cloud next -- a search on another stream
 * waits for it first -- and an asynchronous failure of the build surfaces at that call's synchronisation.  That promise holds
 * for pageable AND for page-locked sources (hipHostMalloc / hipHostRegister): a copy from page-locked memory is truly
 * asynchronous, so the call then waits for the two copies (not for the build) before it returns.  Upload from page-locked
 * memory if the caller can: a pageable upload goes through the runtime's staging buffers. */
int agh_set_cloud(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, const int32_t* cam_source, int64_t n);
int agh_set_cloud_device(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, const int32_t* d_cam_source,
  int64_t n, void* hip_stream);

/* A BATCH of clouds in one context (BASELINE config C5: a batch of 8 x 300k-point clouds): the n_clouds clouds lie end to
 * end in one point array, cloud k = points [offsets[k], offsets[k + 1]) (offsets: n_clouds + 1 host integers, offsets[0] =
 * 0; at most 64 clouds, 2^30 points in total).  Every cloud gets its own search grid, so a radius search only sees the
 * points of its own cloud; point and sample indices of all later calls are positions in the common array.  One
 * agh_find_hands* call then searches samples of ALL clouds in one launch set -- thousands of independent work-groups more
 * per kernel, which is what fills the GPU when a single cloud's 2000 samples do not -- and agh_hypothesis::sample is the
 * position in that call's sample list, as always.  All clouds share the context's camera origins and hand geometry.
 * agh_set_cloud* is the batch of one. */
int agh_set_cloud_batch(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, const int32_t* cam_source, const int64_t* offsets,
  int32_t n_clouds);
int agh_set_cloud_batch_device(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, const int32_t* d_cam_source,
  const int64_t* offsets, int32_t n_clouds, void* hip_stream);

/* The head of Localization::localizeHands (localization.cpp:17-45) on the GPU, followed by the grid build: camera id of
 * raw point i = (i >= size_left); removal of points with a non-finite coordinate (skipped when dense != 0, like
 * pcl::removeNaNFromPointCloud on an is_dense cloud) WITHOUT re-indexing the camera ids (the reference's behaviour);
 * workspace box {xmin,xmax,ymin,ymax,zmin,zmax} (filterWorkspace, :216-245); per-camera voxelisation with cell_size
 * (voxelizeCloud, :247-355; the reference passes 0.003) in lexicographic voxel order, camera 0 block first.  The
 * voxelised cloud becomes the context's cloud (as after agh_set_cloud) and can be read back with agh_get_cloud.
 * The device variant synchronises hip_stream once per cloud for the voxel count (which sizes the search grid); the lattice
 * size, which sizes the voxel bitmap, costs a second synchronisation only for the first cloud of a context or when a cloud's
 * lattice outgrew the bitmap kept from the previous one.  The host variant returns when the caller's buffer has been read;
 * the grid build is still queued on the context's stream (as after agh_set_cloud).  AGH_ERR_CAPACITY if the kept points
 * span more than 2^33 lattice cells. */
int agh_preprocess(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, int64_t n, int64_t size_left, int dense,
  const double workspace[6], double cell_size, int64_t* n_voxels_out);
int agh_preprocess_device(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, int64_t n, int64_t size_left,
  int dense, const double workspace[6], double cell_size, int64_t* n_voxels_out, void* hip_stream);
/* HandleSearch::findHandles + Handle (handle_search.cpp:4-128, handle.cpp:3-74) on a list of hypotheses (normally the
 * ones Learning::classify kept; grasp_localizer.cpp:103 passes min_inliers from the launch file and min_length 0.005).
 * handles_out receives up to handle_cap records, inlier_idx_out the concatenated inlier lists (indices into hands, in
 * the order handle.cpp sees them).  At most 8192 hands and 2048 inliers per seed (AGH_ERR_CAPACITY beyond). */
typedef struct agh_handle
{
  double axis[3];         /* Handle::getAxis: principal direction of the inliers' axes (sign: that of the first inlier) */
  double center[3];       /* getCenter: grasp bottom of the inlier nearest the middle of the handle */
  double approach[3];     /* getApproach */
  double binormal[3];     /* approach x axis */
  double hands_center[3]; /* getHandsCenter: grasp surface of that inlier */
  double width;           /* getWidth: mean grasp width of the inliers */
  int32_t n_inliers;
  int32_t first_inlier;   /* offset of this handle's inliers in inlier_idx_out */
} agh_handle;
int agh_find_handles(agh_ctx* ctx, const agh_hypothesis* hands, int64_t n_hands, int32_t min_inliers, double min_length,
  agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap, int64_t* n_handles_out);

/* The chain the reference's online caller runs per cloud (GraspLocalizer::localizeGrasps, grasp_localizer.cpp:95-103:
 * localizeHands -> predictAntipodalHands -> findHandles) as ONE call with ONE synchronisation: raw capture in, handles out.
 * Preprocessing (as agh_preprocess), grid build, sample selection, hand search, classification (as agh_classify), compaction of
 * the kept hands and the handle search (as agh_find_handles) are queued on the context's stream without a host round trip in
 * between: the voxel count, the hypothesis count and the number of kept hands stay on the device, launches are sized for their
 * bounds (n, 8 x n_samples, min(8 x n_samples, 8192)).  The results equal those of the four separate calls on the same
 * samples, bit for bit.  The first cloud of a context (no voxel bitmap to speculate with yet) and a cloud whose lattice
 * outgrew the bitmap take the preprocessing's own synchronisations once.
 *
 * sample_idx: n_samples indices into the VOXELISED cloud (localizeHands' `indices`; validated on the device:
 * AGH_ERR_INVALID_ARGUMENT), or NULL: n_samples indices are drawn on the device -- one per stratum
 * [k N / S, (k + 1) N / S) of the N voxels, offset splitmix64(sample_seed ^ k * 0x9E3779B97F4A7C15) % width, i.e. sorted,
 * distinct, every point equally likely (hand_search.cpp:36-39 draws a uniform subset with pcl::RandomSample seeded by the
 * clock: not reproducible, never part of parity); with N < S every point is a sample.  The list can be read back (samples_out).
 * classify != 0: Learning::classify between the search and the handle search (needs agh_load_svm*); 0: every hypothesis
 * is handed to the handle search (at most 8192).
 * hands_out (optional, hands_cap records): the hands the handle search ran on, in list order -- inlier_idx_out indexes them.
 * After the call the context holds the voxelised cloud and the search's results like after the separate calls
 * (agh_get_cloud, agh_get_frames, agh_get_images ...). */
typedef struct agh_localize_params
{
  int64_t size_left;       /* camera id of raw point i = (i >= size_left) */
  int32_t dense;           /* as agh_preprocess */
  int32_t classify;
  double workspace[6];
  double cell_size;        /* the reference passes 0.003 */
  const int32_t* sample_idx;
  int64_t n_samples;
  uint64_t sample_seed;
  int32_t min_inliers;     /* grasp_localizer.cpp:103: from the launch file */
  int32_t reserved;
  double min_length;       /* 0.005 */
} agh_localize_params;
typedef struct agh_localize_result
{
  int64_t n_voxels, n_hypotheses, n_hands, n_handles, n_inlier_idx;
} agh_localize_result;
int agh_localize(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, int64_t n, const agh_localize_params* lp,
  agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap, agh_hypothesis* hands_out,
  int64_t hands_cap, int32_t* samples_out, agh_localize_result* result);
/* The same with the raw capture already in device memory (a depth pipeline that runs on the GPU): d_xyz is read in place with
 * the caller's stride and must stay valid until the call returns; the results still come back into host buffers.  Without the
 * 8 MB upload of a 700k-point capture the chain is 0.16 ms shorter. */
int agh_localize_device(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, int64_t n, const agh_localize_params* lp,
  agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap, agh_hypothesis* hands_out,
  int64_t hands_cap, int32_t* samples_out, agh_localize_result* result);
/* The same chain as two calls, so that the NEXT capture goes up while this one is searched (the caller of
 * grasp_localizer.cpp:95-103 is handed one cloud after the other; the 8 MB upload of a 700k-point capture is 0.16 ms of a
 * 0.8 ms call, serial in front of everything):
 *   agh_localize_begin(ctx, cloud k)      queues the whole chain of cloud k and returns without waiting
 *   agh_localize_stage(ctx, cloud k + 1)  copies capture k + 1 into the context's second raw buffer on a stream of its own,
 *                                         beside cloud k's kernels (a pageable source: the call lasts as long as the copy)
 *   agh_localize_end(ctx, outputs)        the one synchronisation; cloud k's results, exactly agh_localize's
 *   agh_localize_begin(ctx, cloud k + 1)  recognises the staged capture (same pointer, stride and count): no upload
 * agh_localize(...) is begin + end.  One chain may be in flight (AGH_ERR_STATE for a second begin, or an end without a
 * begin); between begin and end only agh_localize_stage may be called on the context.  The capture handed to begin must stay valid
 * and unchanged until the agh_localize_end of its chain has returned, the one handed to stage until the agh_localize_end of the
 * chain that adopts it has (a pageable source has been read when agh_localize_stage returns; a pinned one is read asynchronously);
 * sample_idx is copied by begin.  A staged capture that the next begin does not name is dropped (its copy may still be running:
 * keep the source until the next agh_localize_end or agh_synchronize). */
int agh_localize_begin(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, int64_t n, const agh_localize_params* lp);
int agh_localize_stage(agh_ctx* ctx, const float* xyz, int64_t stride_bytes, int64_t n);
int agh_localize_end(agh_ctx* ctx, agh_handle* handles_out, int64_t handle_cap, int32_t* inlier_idx_out, int64_t idx_cap,
  agh_hypothesis* hands_out, int64_t hands_cap, int32_t* samples_out, agh_localize_result* result);

/* The context's current cloud: packed xyz (3 floats per point) and camera ids; returns the number of points. */
int agh_get_cloud(agh_ctx* ctx, float* xyz_out, int32_t* cam_out, int64_t cap);

/* HandSearch::findHands for explicit sample indices.  out receives <= 8*n_samples records, sample-major and
 * orientation-ascending (the reference's concatenation order, hand_search.cpp:194-200).  The host variant stages the sample
 * list in pinned memory of the context, the concatenation kernel writes count, flags and records into pinned memory as well,
 * and the call waits for ONE stream synchronisation (no read-back copies). */
int agh_find_hands(agh_ctx* ctx, const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* out, int64_t cap, int64_t* n_out);
/* Same, everything device-resident and asynchronous on hip_stream (NULL = the context's stream):
 * d_out has room for cap records, *d_n_out (device int64) receives the count.  Device-side errors (capacity, a sample
 * index outside the cloud, AGH_ERR_RETRY) are reported by the next agh_synchronize. */
int agh_find_hands_device(agh_ctx* ctx, const int32_t* d_sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* d_out, int64_t cap, int64_t* d_n_out, void* hip_stream);

/* Linear SVM (one weight vector of 3528 floats + rho) from memory, or any supported model (see agh_load_svm_model)
 * from an OpenCV YAML file. */
int agh_load_svm(agh_ctx* ctx, const float* weights, int32_t n_weights, double rho);
int agh_load_svm_file(agh_ctx* ctx, const char* path);
/* Learning::classify on the hypotheses of the last agh_find_hands* call: keep[i] = 1 iff kept.  Also sets
 * svm_keep in the device-side records; keep may be NULL for the device variant. */
int agh_classify(agh_ctx* ctx, uint8_t* keep, int64_t cap, int64_t* n_kept);
int agh_classify_device(agh_ctx* ctx, uint8_t* d_keep, void* hip_stream);
/* Stamp (agh_hypothesis::epoch) and hypothesis count of the last completed agh_find_hands* call of this context
 * (*n_hyp = -1 while the count of an asynchronous agh_find_hands_device call is not known to the host). */
int agh_get_epoch(agh_ctx* ctx, int32_t* epoch, int64_t* n_hyp);
/* The 80x100 occupancy images (Learning::convertToImage, learning.cpp:320-365) of the hypotheses of the last
 * agh_find_hands* call, packed (250 words each, layout below): what a hypothesis must carry to be classified later, by
 * any context, without the device state of its search.  Returns the number of images written. */
int agh_get_packed_images(agh_ctx* ctx, uint32_t* images, int64_t cap_hyp);
/* Learning::classify on n such images, independent of any earlier call (the reference's classify is stateless and takes
 * any list, learning.cpp:165-247): keep[i] = 1 iff CvSVM::predict == 1; sums (optional) receives the decision values. */
int agh_classify_images(agh_ctx* ctx, const uint32_t* images, int64_t n, uint8_t* keep, double* sums);

/* ---- training side: Learning::train / trainBalanced / convertData, learning.cpp:3-163, 249-318
 * The reference keeps, in every GraspHypothesis, the points of its hand box and their split by camera
 * (rotating_hand.cpp:143-151) so that Learning::train can later rasterise three instances per hand: all points, camera
 * 0's and camera 1's (createInstance, learning.cpp:375-400; same source_to_center for the three).  Here the three
 * 80x100 occupancy images are produced by the hand sweep itself and stand for the instance.
 * Packed image: 250 uint32 words, bit (b & 31) of word (b >> 5) is pixel b = row * 100 + col (set = 255). */
/* on != 0: every following agh_find_hands*(calculates_antipodal = 1) also rasterises the per-camera images. */
int agh_set_training_images(agh_ctx* ctx, int on);
/* The three images (cam = -1, 0, 1) of each hypothesis of the last such call: cap_hyp x 3 x 250 words.  Returns the
 * number of hypotheses written. */
int agh_get_training_images(agh_ctx* ctx, uint32_t* images, int64_t cap_hyp);
/* cv::HOGDescriptor(winSize 64x64).compute(image, winStride 32x32) as convertData calls it (learning.cpp:253-281):
 * n packed images -> n x 3528 floats. */
int agh_hog_images(agh_ctx* ctx, const uint32_t* images, int64_t n, float* desc);
/* convertData's CvSVM::train (C_SVC) on the images' descriptors.  kernel_type AGH_SVM_LINEAR is convertData's
 * uses_linear_kernel = true (the shipped model's shape): CvSVM::optimize_linear_svm compacts the result to one vector,
 * returned as sv_out[0..3527] with alpha_out[0] = 1 and *n_sv_out = 1.  AGH_SVM_POLY2 is uses_linear_kernel = false,
 * the default Learning::train* pass (learning.h:180-182): kernel (x.y)^2; sv_out receives the *n_sv_out support vectors
 * (3528 floats each, room for sv_cap of them: AGH_ERR_CAPACITY with *n_sv_out set if there are more), alpha_out their
 * signed coefficients.  labels[k] > 0 marks a positive (label 1), anything else label -1.  The reference's CvSVMParams
 * defaults are C = 1, max_iter = 1000, eps = FLT_EPSILON.  info_out (optional, 6 ints): solver steps taken, support
 * vectors of the solve, instances of label -1, instances of label +1, kernel rows computed, kernel rows served by the
 * row cache.
 * OpenCV's solver is third-party code restated from its published algorithm: see DESIGN.md for what is pinned. */
#define AGH_SVM_LINEAR 0
#define AGH_SVM_POLY2 1
int agh_train_svm(agh_ctx* ctx, const uint32_t* images, const int8_t* labels, int64_t n, int32_t kernel_type, double C,
  int32_t max_iter, double eps, float* sv_out, int64_t sv_cap, double* alpha_out, int32_t* n_sv_out, double* rho_out,
  int32_t* info_out);
/* CvSVM::save (learning.cpp:312) of such a model in OpenCV's YAML layout (what agh_load_svm_file and CvSVM::load read). */
int agh_save_svm_file(const char* path, int32_t kernel_type, const float* sv, int32_t n_sv, int32_t n_weights,
  const double* alpha, double rho);
/* The same with the training parameters the file's header records (C, term_criteria); agh_save_svm_file writes
 * CvSVMParams' defaults (C = 1, 1000 iterations, FLT_EPSILON), which is what Learning::convertData trains with. */
int agh_save_svm_file_ex(const char* path, int32_t kernel_type, const float* sv, int32_t n_sv, int32_t n_weights,
  const double* alpha, double rho, double C, int32_t max_iter, double eps);
/* Load a model for agh_classify from memory: the compacted linear vector (same as agh_load_svm) or support vectors +
 * alphas with either kernel (CvSVM::predict: sum = -rho + sum_k alpha[k] K(sv_k, x), kept iff sum <= 0). */
int agh_load_svm_model(agh_ctx* ctx, int32_t kernel_type, const float* sv, int32_t n_sv, int32_t n_weights,
  const double* alpha, double rho);

/* ---- multi-GPU: the sample set of ONE cloud sharded over the GPUs of a node --------------------------
 * The reference's two OpenMP loops run over independent samples (hand_search.cpp:77-80, 135-138); here rank g of G takes
 * the contiguous slice [g*S/G, (g+1)*S/G) of the sample list, so the concatenation of the ranks' results in rank order IS
 * the reference's sample-major list.  One process per GPU, one context per process; every rank sets the SAME cloud
 * (agh_set_cloud*) and passes the SAME sample list.  The only data-path communication is RCCL all-gathers over xGMI,
 * issued from this library on the search's stream:
 *   - the ranks' compacted hypothesis lists (one all-gather; fixed-size segments, see agh_find_hands_sharded_device);
 *   - with calculates_antipodal: the all-points normals pass is sharded by point range and cloud_normals_ (3 x N doubles,
 *     hand_search.cpp:13-26) is all-gathered before the hand search, and so are the samples' own normals
 *     (hand_search.cpp:102); with AGH_NORMALS_RAND50: the ranks' rand() draw counts (one int each).
 * The merged list is byte-identical to the single-GPU list apart from the per-call stamp.
 *
 * agh_comm_unique_id: ncclGetUniqueId; call on ONE rank and hand the 128 bytes to the others out of band (MPI, a
 * torch.distributed broadcast, a file).  agh_comm_init: ncclCommInitRank on the context's device (collective: every rank
 * calls it).  agh_comm_init_local: n contexts of THIS process (one host thread each; they may share a device) exchange
 * through device copies instead of RCCL -- RCCL refuses two ranks on one GPU, and this is how the sharded schedule is
 * validated on a single-GPU machine.  A context belongs to at most one communicator. */
#define AGH_COMM_ID_BYTES 128
int agh_comm_unique_id(uint8_t id[AGH_COMM_ID_BYTES]);
int agh_comm_init(agh_ctx* ctx, int32_t rank, int32_t n_ranks, const uint8_t id[AGH_COMM_ID_BYTES]);
int agh_comm_init_local(agh_ctx* const* ctxs, int32_t n_ranks);
int agh_comm_destroy(agh_ctx* ctx);
int agh_comm_rank(const agh_ctx* ctx, int32_t* rank, int32_t* n_ranks); /* 0 / 1 without a communicator */
/* Which RCCL image the library bound ("already mapped: <path>" -- the copy the process had loaded, e.g. PyTorch's --, or
 * "loaded: <name>"), or an error text if none was found.  Calling it BINDS RCCL if that has not happened yet (the same one-time
 * dlopen agh_comm_unique_id / agh_comm_init perform).  RCCL is bound at run time: building the library needs neither its headers
 * nor the library itself. */
const char* agh_comm_rccl_origin(void);
/* Length of the merged list of the last agh_find_hands_sharded (host variant) of this context; AGH_ERR_STATE if the context
 * has no communicator or has not run a sharded search. */
int agh_comm_last_count(const agh_ctx* ctx, int64_t* n_hyp);
/* What the hypothesis all-gather of the last agh_find_hands_sharded* call of this context moved: *segment_bytes = bytes every
 * rank contributed (header + record slots), *n_ranks = ranks of the communicator it ran on (so n_ranks x segment_bytes land in
 * every rank's exchange buffer), *via_rccl = 1 for ncclAllGather, 0 for the in-process communicator's device copies.  Any of
 * the three may be NULL.  AGH_ERR_STATE without a communicator or before the first sharded search. */
int agh_comm_last_exchange(const agh_ctx* ctx, int64_t* segment_bytes, int32_t* n_ranks, int32_t* via_rccl);
/* Testing aid (tests/test_gpu_sharding.py): make this rank fail ON ITS OWN at the named sites of its next sharded call, as an
 * out-of-memory or a launch error would -- 1: per-call buffers (device variant), 2: the Taubin launch, 4: growth of the exchange
 * buffer, 8: the HOG / SVM launch of agh_classify_sharded*, 16: per-call buffers (host variant); one shot per bit.  What the tests
 * then check is the contract of the sharded calls: such a rank still takes part in every collective (an empty segment whose header
 * says so), every rank returns an error for the call (the failing rank its own, the others AGH_ERR_STATE, or AGH_ERR_HIP together
 * when an exchange buffer could not grow), and the communicator stays usable. */
int agh_comm_inject_fault(agh_ctx* ctx, int32_t sites);

/* Tuning: record slots of one rank's exchange segment (0 = the default described at agh_find_hands_sharded_device; values
 * above 8 per sample are clipped).  Every rank must use the same value. */
int agh_comm_set_segment_records(agh_ctx* ctx, int64_t records);
/* The slice of an n-item list that rank `rank` of `n_ranks` takes: [*lo, *hi). */
void agh_shard_slice(int64_t n, int32_t rank, int32_t n_ranks, int64_t* lo, int64_t* hi);
/* HandSearch::findHands with the samples sharded over the communicator's ranks (collective).  Arguments as for
 * agh_find_hands_device; on return (asynchronously on hip_stream) every rank's d_out holds the complete list and
 * *d_n_out its length.  Each rank contributes a segment of max(2 ceil(S/G), 1024) records (never more than 8 ceil(S/G)): scenes
 * yield well under one hypothesis per sample, and xGMI all-gathers of this size are latency bound.  If a rank found more,
 * the call reports AGH_ERR_RETRY at the next agh_synchronize and switches the context to full-size segments (8 per
 * sample) for the following calls; the host variant retries by itself. */
int agh_find_hands_sharded_device(agh_ctx* ctx, const int32_t* d_sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* d_out, int64_t cap, int64_t* d_n_out, void* hip_stream);
int agh_find_hands_sharded(agh_ctx* ctx, const int32_t* sample_idx, int64_t n_samples, int calculates_antipodal,
  agh_hypothesis* out, int64_t cap, int64_t* n_out);
/* Learning::classify after a sharded search (collective): every rank classifies the hypotheses of its own samples (their
 * images are local), the labels travel with a second all-gather of the segments; d_out of the search is updated in
 * place (svm_keep), d_keep (optional, room for the search's cap) receives the flags in list order. */
int agh_classify_sharded_device(agh_ctx* ctx, uint8_t* d_keep, void* hip_stream);
int agh_classify_sharded(agh_ctx* ctx, agh_hypothesis* out, uint8_t* keep, int64_t cap, int64_t* n_kept);

/* Introspection for parity tests / plotting (host buffers). */
int agh_get_frames(agh_ctx* ctx, agh_frame* out, int64_t cap);
/* Neighbour counts of the samples of the last call: n_taubin = points in the Taubin ball (hand_search.cpp:85), n_hands =
 * points radiusSearch(sample, nn_radius_hands) returns (hand_search.cpp:147).  Either may be NULL.  n_hands is counted
 * on demand (one extra kernel per call of this getter; the cloud of the search must still be set): the hand sweep itself
 * only visits the slab of that ball the hand can occupy. */
int agh_get_neighbor_counts(agh_ctx* ctx, int32_t* n_taubin, int32_t* n_hands, int64_t cap);
int agh_get_images(agh_ctx* ctx, uint8_t* images, int64_t cap_hyp); /* cap_hyp x 8000 bytes, 80 rows x 100 cols */
int agh_get_hog(agh_ctx* ctx, float* desc, double* sums, int64_t cap_hyp); /* cap_hyp x 3528 floats (+ SVM sums) */
int agh_get_normals(agh_ctx* ctx, double* normals, int64_t cap_points);  /* cloud_normals_ (3 doubles per point) */
/* GraspHypothesis::getPointsForLearning and the split of its columns by camera (grasp_hypothesis.h:149-170; filled at
 * rotating_hand.cpp:125-157), recomputed on demand for hypothesis `hyp` of the last agh_find_hands* call: `points`
 * receives the 3 x n_b matrix column by column (Eigen's Matrix3Xd layout) in the reference's column order, cam_source[k]
 * the camera id of column k (indices_cam1 = the k with 0, indices_cam2 = the k with 1).  *n_out = n_b in any case;
 * AGH_ERR_CAPACITY if n_b > cap.  A lazy getter (one pass over the cloud per call), not part of the hot path. */
int agh_get_learning_points(agh_ctx* ctx, int64_t hyp, double* points, int32_t* cam_source, int64_t cap, int64_t* n_out);
int agh_get_timing(agh_ctx* ctx, agh_timing* out);
/* Timed launches behind ms[i] of the LAST agh_get_timing call of this context (profile 3 times a sample of the calls):
 * counts[0 .. min(cap, AGH_TIMING_SLOTS) - 1]. */
int agh_get_timing_counts(agh_ctx* ctx, int32_t* counts, int32_t cap);
/* Change agh_params::profile of a live context (0 .. 3); pending timings are dropped. */
int agh_set_profile(agh_ctx* ctx, int32_t level);
int agh_synchronize(agh_ctx* ctx);
/* Device self-test of the IEEE assumptions the parity contract rests on (fp64 div/sqrt, fp32 div/sqrt correctly
 * rounded, no FMA contraction): returns the number of mismatches against host arithmetic on n random inputs. */
int64_t agh_selftest_math(agh_ctx* ctx, int64_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* AGH_H */
