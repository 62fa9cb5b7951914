// agh_internal.h -- shared declarations of the HIP implementation (gfx950 only, no CPU path).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "agh.h"

namespace agh
{

constexpr int kBboxBlocks = 128;   // work-groups of k_bbox per cloud (one slot of six extrema each)
constexpr int kCellCap = 1 << 21;  // cells in the uniform grid table (8 MiB of int32)
constexpr int kCellStride = kCellCap + 4;  // entries of one cloud's cell table (kCellCap + 1 used; 16-byte aligned tables)
constexpr int kMaxRows = 128;      // (y,z) cell rows one ball query may touch
constexpr int kNumSums = 37;       // distinct sequential sums behind M and N (quadric.cpp:40-131)
constexpr int kSumStride = 40;     // doubles per sample in the sums buffer
constexpr int kSlots = 8;          // hypothesis slots per sample = hand orientations (rotating_hand.cpp:13)
constexpr int kImageWords = 250;   // 80x100 occupancy bitmap, one bit per pixel
constexpr int kBigListGrid = 512;  // work-groups of the 4096-class Taubin launches, which walk the list of the samples beyond 1152 neighbours
constexpr int kSweepWg4MinSamples = 4096;  // k_hand_sweep launches beyond this many samples run four work-groups per CU (hand_sweep.hip, WG4)
constexpr int64_t kNormalsChunk = 16384;  // points per batch of the all-points normals pass

// Uniform grid over the cloud's bounding box (stands in for the kd-tree of hand_search.cpp:10-11).
struct GridDesc
{
  double mn[3];
  double cell;
  double inv_cell;
  int dim[3];
  int ncell;
  unsigned done;     // (unused since k_cell_count reduces k_bbox's slots itself)
  unsigned ticket;   // tile tickets of k_cell_scan (reset by the holder of the last one)
};

// Everything a search kernel needs to walk the grid.
// A context holds a BATCH of n_clouds >= 1 clouds laid end to end in one point array (cloud k = points [cloud_off[k],
// cloud_off[k + 1])); point and sample indices are positions in that array.  Every cloud has its own grid descriptor and
// its own cell table (kCellStride entries each, holding positions in the common cell-sorted array), so a ball query only
// sees its own cloud.  n_clouds == 1 is the plain HandSearch::findHands case.
constexpr int kMaxClouds = 64;
struct GridView
{
  const GridDesc* desc;    // n_clouds descriptors (or, after grid_of_*, the one of the query's cloud)
  const int* cell_start;   // n_clouds x kCellStride
  const float4* sorted;    // x, y, z, bits((idx << 1) | cam), cloud-major, then cell-major
  const int* cloud_off;    // n_clouds + 1
  int n_clouds;
};

struct HandGeom
{
  double finger_width, hand_outer_diameter, hand_depth, hand_height, init_bite;
  double cam_origin[2][3];
  double cos_a[8], sin_a[8];   // host libm cos/sin of the 8 hand angles (rotating_hand.cpp:13-15, 89-90)
  double fs[20];               // finger_spacing_ (finger_hand.cpp:8-15)
  double thr[40];              // sorted unique thresholds {fs_i, fs_i + fw}
  int n_thr;
  int lo_idx[20], hi_idx[20];  // index of fs_i and of fs_i + fw in thr
  double depths[16];           // init_bite, then += 0.005 while <= hand_depth (finger_hand.cpp:204)
  double backs[16];            // -1.0 * (hand_depth - depth_k)
  double boxy[16];             // backs[k] + hand_depth (rotating_hand.cpp:127)
  int n_depths;
  double cos_antipodal;        // cos(20 deg) (antipodal.cpp:16)
  // Uniform-cell look-up tables over the two sorted tables above: lut[cell(v)] = number of table entries in EARLIER
  // cells; the entries of v's own cell (at most kLutProbe of them, checked on the host) are compared explicitly, so
  // rank(v) = #{entries < v} (or <= v) is exact for every v.
  double xlut_lo, xlut_scale;
  double ylut_lo, ylut_scale;
  unsigned char xlut[1024];
  unsigned char ylut[64];
  int x_probes, y_probes;      // most entries any one cell holds: the probes a kernel instantiation must make
};
constexpr int kLutProbe = 4;

struct HogTablesDev
{
  // Gradient LUT for binary images: index = (sx+1)*3 + (sy+1), sx/sy = sign of dx/dy
  float mag0[9], mag1[9];
  int bin0[9], bin1[9];
  // coef[code][bin] = mag0 for bin0, mag1 for bin1, 0 elsewhere: a pixel's vote is hist[bin] += coef[code][bin] * w for
  // all nine bins (adding the +0.0f products is exact), i.e. 9 mul + 9 add instead of 18 compares and 18 selects
  alignas(16) float coef[9][12];
  // pixData of HOGCache::init in its accumulation order (count1 | count2 | count4 groups): position of entry k inside
  // the 16x16 block and its weight gradWeight * histWeights for each of the four cells (0 where it does not vote)
  int pix_x[256], pix_y[256];
  float pix_wcell[256][4];
};

enum SampleStatus : int
{
  kStatusOk = 0,
  kStatusOverflow = 1,    // neighbourhood larger than the kernel's LDS capacity
  kStatusDegenerate = 2,  // no frame (an empty neighbourhood; rank-deficient pencils DO get one since round 3): no hypotheses
  kStatusRows = 3,
  kStatusBadIndex = 4,    // sample index outside the cloud (device-resident sample lists are validated on the device)
  kStatusSkipped = 5      // kSampleSkip: an unused slot of a device-drawn sample list
};
constexpr int32_t kSampleSkip = INT32_MIN;
constexpr int kBigCap = 4096;    // per-sample neighbour-list scratch (and the largest LDS-resident class of K1a) without ...
constexpr int kHugeCap = 6144;   // ... and with the 6144 class (Ctx::huge_classes)
constexpr int64_t kHugeEntries = 1ll << 21;  // pool of the neighbourhoods beyond 4096 points of ONE launch: points in all (AGH_ERR_CAPACITY beyond)

// K-1 (voxelize.hip): per-camera voxel lattice of the preprocessing step
constexpr unsigned long long kVoxMaxWords = 1ull << 28;  // 1 GiB of bitmap (a 6 m x 6 m x 3 m lattice at 3 mm)
struct VoxDesc
{
  unsigned mn_enc[2][3], mx_enc[2][3];  // order-preserving encodings of the float minima / maxima per camera
  double mn[2][3];                      // the minima as the reference holds them (10000 if no smaller coordinate)
  int dim[2][3];
  unsigned long long bits[2], word_ofs[2], n_words;
  long long n_kept[2], n_vox[2];
  int error;
};
struct VoxWorkspace
{
  double lo[3], hi[3];
};

struct Comm;  // shard.hip

// Host-buffer entry points: the concatenation kernel (K4) writes every record a second time, straight into pinned host
// memory of the context, and the count and the error word into a header there -- the list is on the host when the stream
// drains, with no read-back copies behind the kernels.  rec == nullptr: no mirror (device-resident callers).
struct HostMirror
{
  agh_hypothesis* rec;  // room for `cap` records (list positions beyond it stay on the device only)
  int64_t cap;
  int64_t* hdr;         // [0] hypotheses found, [1] flags[0] as the kernel saw it
};

// agh_localize_begin / _stage / _end (api.hip): the one chain in flight and the capture staged for the next one
struct LocalizeState
{
  bool active = false;        // a chain is queued and agh_localize_end has not collected it
  bool repeated = false;      // inside the one repeat of a whole call (the voxel lattice outgrew the speculative bitmap)
  bool deferred = false;      // the voxel count is still on the device
  bool classify = false, with_sequential = false, explicit_samples = false;
  int64_t S = 0, nv = 0;
  int32_t min_inliers = 0;
  double min_length = 0.0, x1 = 0.0, x2 = 0.0;
  agh_localize_params lp{};   // (sample_idx cleared: the list lives in the pinned staging)
  const float* d_raw = nullptr;  // where the chain read the raw capture (the context's raw buffer or the caller's device memory)
  int64_t dev_stride = 0, n_raw = 0;
  bool staged = false;        // agh_localize_stage: a capture is (being) copied into d_stage_xyz
  const float* staged_src = nullptr;
  int64_t staged_stride = 0, staged_n = 0;
};

struct Ctx
{
  agh_params p;
  HandGeom geom;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t copy_done = nullptr;
  hipEvent_t copy_gate = nullptr;   // recorded on the build's stream before its kernels: the camera-id copy waits for earlier work only
  hipEvent_t xyz_copied = nullptr;  // host-buffer agh_set_cloud from PINNED memory: recorded behind the coordinates' copy
  bool cam_copy_on_copy_stream = false;  // the last grid_build put the camera ids on copy_stream (copy_done was recorded)
  hipStream_t copy_stream = nullptr;  // host-buffer agh_set_cloud: the camera ids go up here while the grid build's first kernels run
  bool copy_stream_failed = false;    // its creation failed once: not retried (the ids go up on the build's stream)
  std::string err;

  // cloud
  int64_t n = 0;
  const float* d_xyz = nullptr;  // borrowed or owned
  int64_t stride_floats = 3;
  const int32_t* d_cam = nullptr;
  float* own_xyz = nullptr;
  int32_t* own_cam = nullptr;
  int64_t own_cap = 0;
  int64_t own_cap_floats = 0;
  int32_t* d_idx_own = nullptr;  // device copy of a host sample list
  int64_t idx_cap = 0;
  // pinned staging of the host-buffer entry points (hipHostMalloc, device-visible): [256-byte header | sample indices |
  // records]; the caller's pageable buffers are copied from / to it by the CPU, the GPU reads and writes it directly
  uint8_t* h_pin = nullptr;
  int64_t h_pin_samples = 0, h_pin_records = 0;
  HostMirror mirror{ nullptr, 0, nullptr };  // set by agh_find_hands around its device call
  uint8_t* h_pin_handles = nullptr;  // agh_find_handles: [256-byte header | hands in | handles out | inlier indices out]
  int64_t h_pin_handles_cap = 0;     // in hands
  bool handles_sequential = false;   // the previous agh_find_handles needed k_handle_greedy: launch it along
  uint8_t* h_pin_keep = nullptr;     // agh_classify: the keep flags, written by K3 itself
  int64_t h_pin_keep_cap = 0;

  // preprocessing (K-1)
  VoxDesc* d_vox_desc = nullptr;
  uint8_t* d_vox_code = nullptr;   // per raw point: 0 dropped, 1 | cam << 1 kept
  int* d_vox_blk = nullptr;        // finite-point counts per 1024 raw points
  int* d_vox_blk2 = nullptr;       // popcounts per 4096 bitmap words
  long long* d_vox_total = nullptr;
  unsigned* d_vox_bitmap = nullptr;
  int64_t vox_bitmap_cap = 0;      // words (a multiple of the popcount block)
  int64_t vox_last_words = 0;   // lattice words of the last preprocessed cloud (a much larger kept bitmap is dropped)
  VoxDesc* h_vox_desc = nullptr;   // pinned host mirror of the descriptor, written by k_vox_lattice / k_vox_totals
  float* d_vox_xyz = nullptr;      // voxelised cloud (packed xyz) and camera ids
  int32_t* d_vox_cam = nullptr;
  int64_t vox_cap = 0;
  float* d_raw_xyz = nullptr;      // device copy of a raw host cloud
  float* d_stage_xyz = nullptr;    // ... and of the NEXT one (agh_localize_stage); the two change places when it is adopted
  int64_t stage_cap = 0;           // floats
  hipStream_t stage_stream = nullptr;
  hipEvent_t stage_done = nullptr;
  LocalizeState loc;
  int64_t raw_cap = 0;             // floats

  // handle search (K5)
  agh_hypothesis* d_h_hands = nullptr;
  unsigned long long* d_h_bits = nullptr;
  int* d_h_rowcnt = nullptr;
  int* d_h_first = nullptr;
  int* d_h_n = nullptr;
  int* d_h_idx = nullptr;
  int* d_h_counts = nullptr;  // HandleCounts
  int* d_h_tmp = nullptr;     // h_cap: scratch of k_handle_batch (inlier lists in commit order)
  agh_handle* d_h_handles = nullptr;
  int64_t h_cap = 0;

  // grid (per cloud of the batch)
  int n_clouds = 1;
  int clouds_cap = 0;               // clouds the tables below are allocated for
  std::vector<int64_t> cloud_off;   // host copy, n_clouds + 1
  std::vector<int32_t> cloud_off_i32;
  bool cloud_off_on_device = false;
  bool defer_cloud_count = false;  // set by the deferred preprocessing for the agh_set_cloud_device call it makes next
  bool n_is_bound = false;  // agh_localize, between its launches and its synchronisation: `n` is the RAW point count, an upper bound of
                            // the voxelised cloud's; the true count is d_cloud_off[1], written by the voxeliser
  int* d_cloud_off = nullptr;       // kMaxClouds + 1
  int32_t* d_scloud = nullptr;      // s_cap: cloud of every sample of the last call (written by k_taubin_moments)
  GridDesc* d_desc = nullptr;       // clouds_cap
  float* d_bbox_part = nullptr;     // clouds_cap x kBboxBlocks x 6 partial extrema of k_bbox
  int* d_cell_start = nullptr;      // clouds_cap x kCellStride
  int* d_cell_count = nullptr;      // clouds_cap x kCellCap
  int* d_block_sums = nullptr;
  unsigned long long* d_tile_state = nullptr;  // clouds_cap x kCellCap / 1024 look-back descriptors of k_cell_scan, tagged with the build number
  unsigned build_gen = 0;
  bool grid_clean = false;      // cell counts, bbox and counters are in their reset state (self-cleaning kernels)
  int* d_cell_of = nullptr;     // n
  int* d_rank_of = nullptr;     // n: rank of a point inside its cell
  float4* d_sorted = nullptr;   // n
  int64_t grid_cap = 0;
  bool has_cloud = false;
  const int32_t* pending_cam_host = nullptr;  // host-buffer agh_set_cloud: camera ids still to upload (grid_build does it while
  int64_t pending_cam_n = 0;                  // its first three kernels run: only k_scatter reads them)
  bool cloud_async = false;     // agh_set_cloud (host buffers) left the grid build running on `stream`; a search on ANOTHER
                                // stream must first wait for it (order_after_cloud)

  // per-call buffers (sized for s_cap samples)
  int64_t s_cap = 0;
  int32_t* d_samples = nullptr;
  double* d_sums = nullptr;        // s_cap * kSumStride
  int32_t* d_nt = nullptr;         // n_taubin
  int32_t* d_ovf = nullptr;        // {count, samples...}: the samples k_taubin_moments<1152> hands on to the 4096 class (s_cap + 16)
  int32_t* d_nh = nullptr;         // n_hands
  int32_t* d_status = nullptr;
  int* d_weight = nullptr;         // candidate count of each sample's hand-search ball (scheduling weight)
  uint8_t* d_vmask = nullptr;      // per sample: orientations with a hypothesis (k_hand_sweep -> concatenation)
  int* d_order = nullptr;          // samples by descending n_t: blockIdx -> sample of k_taubin_frame
  int* d_order_sweep = nullptr;    // blocks of 32 samples, heaviest first: blockIdx -> sample of k_hand_sweep (made by K1b)
  int order_frame_s = 0;           // ... and the same for k_taubin_frame's longest-first order (d_order)
  const int32_t* order_frame_samples = nullptr;
  int order_sweep_s = 0;           // the sample count / list that order was made for (0: none)
  const int32_t* order_sweep_samples = nullptr;
  float4* d_nbr = nullptr;         // s_cap * nbr_stride sorted neighbour lists
  int64_t nbr_stride = 0;
  double* d_eig = nullptr;         // s_cap * 12 : params[10], eigenvalue, valid
  agh_frame* d_frames = nullptr;
  agh_hypothesis* d_slots = nullptr;   // s_cap * 8
  uint32_t* d_images = nullptr;        // s_cap * 8 * kImageWords
  int32_t* d_slot_index = nullptr;     // compacted position of each slot (or -1)
  int32_t* d_scan_tmp = nullptr;
  agh_hypothesis* d_out_own = nullptr;  // compacted hypotheses when the caller gave host memory
  int64_t* d_nout = nullptr;
  uint32_t* d_out_images = nullptr;     // compacted images (s_cap * 8 * kImageWords)
  uint32_t* d_images_cam = nullptr;     // training side: per slot the camera-0 and camera-1 images (s_cap * 16 * kImageWords)
  int64_t images_cam_cap = 0;           // in samples
  bool training_images = false;         // agh_set_training_images
  bool last_has_cam_images = false;     // the last hand sweep filled d_images_cam
  int32_t* d_draw_ofs = nullptr;        // RAND50: offset of each sample's 50 draws
  int32_t* d_draws = nullptr;
  int64_t draws_cap = 0;
  int64_t last_s = 0;
  int32_t epoch = 0;         // stamp of the last agh_find_hands* call (agh_hypothesis::epoch; process-wide counter)
  int64_t last_nout = -1;
  int64_t last_cap = 0;
  agh_hypothesis* d_out_last = nullptr;  // where the last call's compacted records live
  int64_t* d_nout_last = nullptr;
  int32_t* d_flags = nullptr;  // [0] any overflow, [1] ...
  bool huge_classes = false;  // ... and the 6144 class behind those (K1a through global scratch, K1c in LDS): neighbourhoods of up to
                              // kHugeCap points; enabled like big_classes, by the first call that needs it (AGH_ERR_RETRY once)
  float4* d_huge_stage = nullptr;            // kHugeEntries points of the neighbourhoods beyond 4096, unsorted
  unsigned long long* d_huge_key = nullptr;  // ... their (d2, index) keys
  float4* d_huge_sorted = nullptr;           // ... the sorted lists of those beyond the per-sample scratch (> kHugeCap)
  double* d_huge_normals = nullptr;          // ... and their normals (3 x kHugeEntries: x | y | z), K1c beyond the LDS classes
  long long* d_huge_base = nullptr;          // per sample: first pool entry of its neighbourhood (s_cap)
  unsigned long long* d_huge_count = nullptr;  // entries handed out in this launch
  bool big_classes = false;  // launch the larger capacity classes of K1a / K1c too (sticky; set by the first call that
                             // met a neighbourhood beyond the first class, see AGH_ERR_RETRY)
  bool zero_flags_pending = false;  // the next k_taubin_moments launch clears d_flags first

  // normals for the antipodal test (cloud_normals_, hand_search.cpp:13-14)
  double* d_normals = nullptr;  // 3 * n
  int64_t normals_cap = 0;
  bool has_normals = false;

  // svm / hog
  float* d_svm_w = nullptr;
  double svm_rho = 0.0;
  bool has_svm = false;
  // a model that is not the compacted linear vector (several support vectors and/or the POLY degree-2 kernel)
  bool svm_general = false;
  int svm_kernel = 0;               // AGH_SVM_*
  int svm_n_sv = 1;
  float* d_svm_svT = nullptr;       // support vectors, tiles of 64 (train.hip: xt_index)
  double* d_svm_alpha = nullptr;
  float* d_cls_desc = nullptr;      // descriptors of the hypotheses being classified (general models)
  int64_t cls_desc_cap = 0;         // in hypotheses
  float* d_cls_kbuf = nullptr;      // kernel values, hypotheses x support vectors
  int64_t cls_kbuf_cap = 0;         // in floats
  HogTablesDev* d_hog = nullptr;
  HandGeom* d_geom = nullptr;
  float* d_desc_out = nullptr;  // optional descriptor dump
  double* d_svm_sums = nullptr;
  uint8_t* d_keep = nullptr;
  int64_t keep_cap = 0;
  uint32_t* d_cls_images = nullptr;  // agh_classify_images: uploaded packed images, keep flags, decision values
  uint8_t* d_cls_keep = nullptr;
  double* d_cls_sums = nullptr;
  int64_t cls_images_cap = 0;

  // multi-GPU: the sample set of one cloud sharded over the ranks of a communicator (shard.hip)
  Comm* comm = nullptr;
  uint8_t* d_xbuf = nullptr;      // n_ranks segments of [160-byte header | seg_records records], all-gathered in place
  int64_t xbuf_bytes = 0;
  double* d_nbuf = nullptr;       // n_ranks segments of per-sample normals (antipodal mode), all-gathered in place
  int64_t nbuf_doubles = 0;
  int64_t* d_xcnt = nullptr;      // n_ranks RAND50 draw counts + scratch
  bool shard_symmetric_error = false;  // the last sharded device call failed on something every rank saw alike, after the ranks
                                       // had met in a collective: nobody is left waiting, the communicator stays usable
  int64_t shard_seg_override = 0;    // agh_comm_set_segment_records
  int shard_inject = 0;              // agh_comm_inject_fault (testing aid): sites that fail on this rank, one shot each
  bool shard_full_exchange = false;  // exchange all 8 slots per sample instead of the 2-per-sample prefix
  int64_t shard_seg_records = 0, shard_seg_bytes = 0, shard_S = 0;
  agh_hypothesis* shard_out = nullptr;  // the caller's merged list of the last sharded search
  int64_t shard_cap = 0;
  int64_t* shard_nout = nullptr;
  int64_t shard_last_n = -1;            // length of the merged list, once a host entry point has read it

  long long* d_dbg = nullptr;  // AGH_DEBUG_CLOCKS: per-sample phase timestamps of k_hand_sweep (dumped to a file)
  int debug_stop_sweep = 0;    // AGH_DEBUG_STOP_SWEEP: phase-timing aid, see k_hand_sweep
  int debug_stop_moments = 0;  // AGH_DEBUG_STOP_MOMENTS
  int debug_stop_frame = 0;    // AGH_DEBUG_STOP_FRAME
  int debug_stop_hog = 0;      // AGH_DEBUG_STOP_HOG  (all four only in builds with -DAGH_DEBUG_HOOKS)

  // timing
  std::vector<hipEvent_t> ev;
  std::vector<const char*> ev_name;
  int ev_used = 0;
  unsigned prof_calls = 0;     // profile 3: calls seen; every fourth one is timed
  agh_timing timing;
  int32_t timing_counts[AGH_TIMING_SLOTS] = { 0 };  // timed launches per slot of the last agh_get_timing (agh_get_timing_counts)
};

inline int class_level(const Ctx* c) { return c->huge_classes ? 2 : (c->big_classes ? 1 : 0); }

// a search about to run on `st`: if the host-buffer agh_set_cloud left its grid build running on the context's stream and `st`
// is another stream, wait for the build first (streams the caller brings may be non-blocking ones)
inline hipError_t order_after_cloud(Ctx* c, hipStream_t st)
{
  if (!c->cloud_async || st == c->stream)
    return hipSuccess;  // (in order behind the build on the context's own stream: the flag stays up for a LATER search on another
                        // stream, which is not -- ADVICE r4)
  c->cloud_async = false;
  return hipStreamSynchronize(c->stream);
}

// Bits of d_flags[0]: 1 = a Taubin neighbourhood beyond the capacity classes this context launched, 2 = output buffer too small,
// 4 = a sample index outside the cloud, 16 = a rank's exchange segment overflowed (sharded search).  The sharded search decides
// on what EVERY rank reported, never on a rank's own state -- two ranks that disagreed would part ways before the next collective,
// and raw RCCL has no watchdog -- so a rank's findings travel in word 1 of its segment header and the merge kernel turns the
// gathered headers into three more bits, identical on every rank: kFlagShardRetry (some rank that had NOT launched the larger
// classes needs them), kFlagShardHard (some rank that had launched them still overflowed), kFlagSharded (a merge ran: the
// decision comes from these bits, the rank's own bit 0 is ignored).
constexpr int kFlagShardRetry = 32, kFlagShardHard = 64, kFlagSharded = 128, kFlagShardRetryHuge = 256;
// ... and two for a rank that took part in the call's collectives without doing its share (shard.hip, "no rank leaves alone"):
// kFlagShardPeerFailed (a rank's own failure: memory, a launch), kFlagShardPeerNoCloud (a rank holds no cloud) -- from the header
// bits kHdrRankFailed / kHdrRankNoCloud of its segment
constexpr int kFlagShardPeerFailed = 512, kFlagShardPeerNoCloud = 1024;
constexpr int kHdrRankFailed = 32, kHdrRankNoCloud = 64;
// word 1 of a segment header: an overflow of the capacity classes the rank had launched -- 1 with the larger classes off (level
// 0: a retry with them helps), 8 with them on (level 1: a retry with the 6144 class helps), 16 with that on too (level 2: hard)
// -- and 4 = bad sample index
__host__ __device__ inline int shard_header_word(int flags0, int class_level)
{
  return ((flags0 & 1) ? (class_level >= 2 ? 16 : (class_level == 1 ? 8 : 1)) : 0) | (flags0 & 4);
}

// ---- kernel launchers (defined in the .hip files) ----
int vox_stage1(Ctx* c, const float* d_xyz, int64_t stride_floats, int64_t n, int64_t size_left, int dense,
  const double workspace[6], double cell, hipStream_t st, int64_t cap_words, VoxDesc* host_desc, bool with_lattice);
int vox_stage2(Ctx* c, const float* d_xyz, int64_t stride_floats, int64_t n, double cell, int64_t n_words, hipStream_t st,
  VoxDesc* host_desc, bool with_lattice, int* cloud_off_out = nullptr);
// host mirror of the handle search's results (pinned memory of the context; all nullptr / 0: none)
struct HandleMirror
{
  agh_handle* handles;
  int handle_cap;
  int* idx;
  int idx_cap;
  int* counts;  // [0] handles, [1] inlier indices, [2] error, [3] the batched walk declined and no sequential kernel was launched
};
int handle_search(Ctx* c, int64_t H, double x1, double x2, int min_inliers, double min_length, hipStream_t st,
  const HandleMirror& hm, bool with_sequential, const int* d_H = nullptr);
int grid_build(Ctx* c, hipStream_t st);
int taubin_frames(Ctx* c, const int32_t* d_samples, int64_t S, double radius, agh_frame* d_frames, int32_t* d_nt,
  bool write_normals, hipStream_t st);
int taubin_moments_eigen(Ctx* c, const int32_t* d_samples, int64_t S, double radius, int32_t* d_nt, hipStream_t st,
  bool with_draw_offsets = false);
int taubin_frame_stage(Ctx* c, const int32_t* d_samples, int64_t S, double radius, agh_frame* d_frames, int32_t* d_nt,
  bool write_normals, hipStream_t st, bool draw_offsets_done = false);
int hand_sweep(Ctx* c, const int32_t* d_samples, int64_t S, bool use_normals, hipStream_t st);
int ball_counts(Ctx* c, int64_t S, hipStream_t st);  // d_nh of the last call's samples (agh_get_neighbor_counts)
int compact_hypotheses(Ctx* c, int64_t S, agh_hypothesis* d_out, int64_t cap, int64_t* d_nout, hipStream_t st,
  int64_t* d_hdr_flags = nullptr);  // d_hdr_flags: sharded search, slices of at most 65536 samples (see shard.hip)
int hog_svm(Ctx* c, int64_t n_hyp_cap, uint8_t* d_keep, hipStream_t st);
int hog_images(Ctx* c, const uint32_t* d_images, const int32_t* d_order, int64_t n, float* d_desc, hipStream_t st);
int svm_predict_general(Ctx* c, const float* d_desc, int64_t cap, uint8_t* d_keep, hipStream_t st);
int svm_predict_images(Ctx* c, const float* d_desc, int64_t n, uint8_t* d_keep, double* d_sums, hipStream_t st);
int svm_predict_desc(Ctx* c, const float* d_desc, int64_t cap, const int64_t* d_nhyp, agh_hypothesis* d_out, uint8_t* d_keep,
  double* d_sums, hipStream_t st);
int svm_load_general(Ctx* c, int kernel_type, const float* sv, int n_sv, const double* alpha, double rho);
void hog_tables_host(HogTablesDev* t);
int64_t selftest_math(Ctx* c, int64_t n, uint64_t seed);

void timing_mark(Ctx* c, const char* name, hipStream_t st);
hipEvent_t timing_next_event(Ctx* c, const char* name);
bool timing_launch_events(Ctx* c, const char* name, hipEvent_t* start, hipEvent_t* stop);
void timing_begin(Ctx* c, hipStream_t st);
int32_t next_epoch();
int ensure_call_buffers(Ctx* c, int64_t S);
int ensure_host_staging(Ctx* c, int64_t samples, int64_t records);
constexpr int64_t kPinHeaderBytes = 256;  // pinned staging of the host-buffer entry points: [header | sample indices | records]
// (re)allocation of one of the context's device buffers
template <typename T>
inline int dev_alloc(Ctx* c, T** p, size_t count)
{
  if (*p)
  {
    (void) hipFree(*p);
    *p = nullptr;
  }
  const hipError_t e = hipMalloc((void**) p, (count > 0 ? count : 1) * sizeof(T));
  if (e != hipSuccess)
  {
    c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
    return AGH_ERR_HIP;
  }
  return AGH_OK;
}
int ensure_clouds(Ctx* c, int C);
int ensure_draws(Ctx* c, int64_t count, hipStream_t st);
int normals_pass(Ctx* c, int64_t p0, int64_t p1, hipStream_t st);
void comm_release(Ctx* c);

// ---- device helpers ----
#if defined(__HIPCC__)

__device__ __forceinline__ unsigned enc_float(float f)
{
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_float(unsigned u)
{
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ int cell_coord(const GridDesc& g, double v, int a)
{
  int c = (int) floor((v - g.mn[a]) * g.inv_cell);
  return min(max(c, 0), g.dim[a] - 1);
}

// The view of the grid of cloud k / of the cloud that holds point p (uniform per work-group: scalar loads).
__device__ __forceinline__ GridView grid_of_cloud(GridView gv, int k)
{
  gv.desc += k;
  gv.cell_start += (int64_t) k * kCellStride;
  return gv;
}
__device__ __forceinline__ int cloud_of_point(const GridView& gv, int p)
{
  // (p wave-uniform; the offsets of the <= 64 clouds are compared a lane each: one load and a ballot, where a loop over the clouds
  // was a chain of dependent scalar loads -- 0.6 us per work-group with a batch of eight)
  if (gv.n_clouds == 1)
    return 0;
  const int lane = threadIdx.x & 63;
  const int off = (lane >= 1 && lane < gv.n_clouds) ? gv.cloud_off[lane] : 0x7fffffff;
  return __popcll(__ballot(p >= off));
}

// Squared distance exactly as FLANN's L2_Simple<float> accumulates it (see oracle a2): ((0+dx*dx)+dy*dy)+dz*dz.
__device__ __forceinline__ float flann_d2(float qx, float qy, float qz, float px, float py, float pz)
{
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  float d2 = __fmul_rn(dx, dx);
  d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
  d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
  return d2;
}

// Inclusive sum scan over the 64 lanes of a wave by DPP: row_shr 1, 2, 4, 8 inside each 16-lane row (a lane without a source
// adds zero), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 -- six v_add_u32_dpp, where a
// __shfl_up loop is six dependent ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ int wave_incl_scan_i32(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// The value lane (l ^ O) holds, for a double, without the LDS crossbar: v_permlane32_swap / v_permlane16_swap (gfx950) for
// O = 32 / 16, DPP moves inside a 16-lane row for O = 8 (row_ror:8), 4 (row_shl:4 into banks 0 and 2, row_shr:4 into banks 1 and
// 3), 2 and 1 (quad_perm).  A __shfl_xor of a double is two ds_bpermute round trips (~130 cycles of latency in a dependent
// butterfly); these are two to four VALU moves.
template <int O>
__device__ __forceinline__ int xor_partner_i32(int x)
{
  static_assert(O == 32 || O == 16 || O == 8 || O == 4 || O == 2 || O == 1, "a single bit of the lane index");
  if (O == 32)
  {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // r[0] = {x[0..31], x[0..31]}, r[1] = {x[32..63], x[32..63]}
    return (threadIdx.x & 32) ? (int) r[0] : (int) r[1];
  }
  if (O == 16)
  {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // r[0] = rows {0, 0, 2, 2}, r[1] = rows {1, 1, 3, 3}
    return (threadIdx.x & 16) ? (int) r[0] : (int) r[1];
  }
  if (O == 8)
    return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);  // row_ror:8
  if (O == 4)
  {
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);  // row_shl:4 -> lanes 0-3, 8-11 of a row take l + 4
    return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);         // row_shr:4 -> lanes 4-7, 12-15 take l - 4
  }
  if (O == 2)
    return __builtin_amdgcn_update_dpp(0, x, 0x4e, 0xf, 0xf, false);  // quad_perm [2, 3, 0, 1]
  return __builtin_amdgcn_update_dpp(0, x, 0xb1, 0xf, 0xf, false);    // quad_perm [1, 0, 3, 2]
}
template <int O>
__device__ __forceinline__ double xor_partner_f64(double x)
{
  return __hiloint2double(xor_partner_i32<O>(__double2hiint(x)), xor_partner_i32<O>(__double2loint(x)));
}
// The butterfly of the oracle's LaneSum64: v + v(l ^ 32), then ^ 16, 8, 4, 2, 1; every lane ends with the same total.
__device__ __forceinline__ double wave_allsum_f64(double v)
{
  v = v + xor_partner_f64<32>(v);
  v = v + xor_partner_f64<16>(v);
  v = v + xor_partner_f64<8>(v);
  v = v + xor_partner_f64<4>(v);
  v = v + xor_partner_f64<2>(v);
  v = v + xor_partner_f64<1>(v);
  return v;
}

// Wave-wide integer sum, the same on every lane.
__device__ __forceinline__ int wave_allsum_i32(int v)
{
  v += xor_partner_i32<32>(v);
  v += xor_partner_i32<16>(v);
  v += xor_partner_i32<8>(v);
  v += xor_partner_i32<4>(v);
  v += xor_partner_i32<2>(v);
  v += xor_partner_i32<1>(v);
  return v;
}

// Row table of one ball query, built cooperatively in LDS.
struct RowTable
{
  int begin[kMaxRows];
  int prefix[kMaxRows + 1];
  int seg_row[64];  // the row that holds flat candidate 128 j, for the first 64 segments of 128 candidates (FlatRows)
  int nrows;
  int total;
  int bad;
};

// Fill rt for the ball (q, rpad).  Must be called by all threads of the block; ends with a barrier.
// SLAB: the caller only wants the points p of the ball with |a . (p - q)| < hh (the hand sweep's crop along the hand axis,
// rotating_hand.cpp:37-51, keeps a 2 hh thick slab of a 2 r ball: about a fifth of its points).  Every row is then clipped
// to the part of its chord that can hold such a point -- by interval arithmetic over the row's (y, z) cell, widened by
// kSlabMargin, which is orders of magnitude above the rounding of the caller's own test (the float32 subtraction it
// starts from is off by < 1e-8 m) -- and rows the slab misses are dropped: half the candidates and half the row segments
// of the plain ball at C2.  The caller's exact test still decides every point; this only prunes what it never accepts.
constexpr double kSlabMargin = 1e-6;
// `gpre`: the query's grid descriptor if the caller has loaded it already (so that the load could be issued before other
// loads the caller had to wait for).
template <bool SLAB = false>
__device__ __forceinline__ void build_rows(const GridView& gv, float qx, float qy, float qz, double rpad, RowTable& rt,
  const double* slab_axis = nullptr, double slab_hh = 0.0, const GridDesc* gpre = nullptr)
{
  const GridDesc& g = gpre ? *gpre : *gv.desc;
  const int tid = threadIdx.x;
  const int lx = cell_coord(g, (double) qx - rpad, 0), hx = cell_coord(g, (double) qx + rpad, 0);
  const int ly = cell_coord(g, (double) qy - rpad, 1), hy = cell_coord(g, (double) qy + rpad, 1);
  const int lz = cell_coord(g, (double) qz - rpad, 2), hz = cell_coord(g, (double) qz + rpad, 2);
  const int ny = hy - ly + 1, nz = hz - lz + 1;
  const int nrows = ny * nz;
  if (tid == 0)
  {
    rt.nrows = nrows;
    rt.bad = nrows > kMaxRows;
  }
  if (nrows <= kMaxRows)
  {
    for (int t = tid; t < nrows; t += blockDim.x)
    {
      // (t / ny for ny <= 9, t < 128, without the ~40-instruction integer division: exact in float)
      const int tq = (int) (((float) t + 0.5f) * (1.0f / (float) ny));
      const int cy = ly + (t - tq * ny), cz = lz + tq;
      // distance from q to the row's (y,z) slab; rows that cannot touch the ball are skipped
      const double y0 = g.mn[1] + cy * g.cell, z0 = g.mn[2] + cz * g.cell;
      const double dy = fmax(fmax(y0 - (double) qy, (double) qy - (y0 + g.cell)), 0.0);
      const double dz = fmax(fmax(z0 - (double) qz, (double) qz - (z0 + g.cell)), 0.0);
      int b = 0, len = 0;
      const double rem = rpad * rpad - (dy * dy + dz * dz);
      if (rem >= 0.0)
      {
        // the ball's chord along x at this row: a point of the row within rpad of q has |x - qx| <= sqrt(rem)
        // (dy, dz are the smallest possible offsets), so only the cells under the chord are candidates -- about a
        // third fewer than the bounding box's
        const double xr = sqrt(rem);
        double xlo = (double) qx - xr, xhi = (double) qx + xr;
        bool keep = true;
        if (SLAB)
        {
          // a_y (y - qy) + a_z (z - qz) over the row's cell lies in [L, U]; a point passes only if a_x (x - qx) + that is
          // inside (-hh, hh), i.e. a_x (x - qx) in (-hh - U, hh - L)
          const double ya = (y0 - kSlabMargin) - (double) qy, yb = (y0 + g.cell + kSlabMargin) - (double) qy;
          const double za = (z0 - kSlabMargin) - (double) qz, zb = (z0 + g.cell + kSlabMargin) - (double) qz;
          const double ey0 = slab_axis[1] * ya, ey1 = slab_axis[1] * yb, ez0 = slab_axis[2] * za, ez1 = slab_axis[2] * zb;
          const double L = fmin(ey0, ey1) + fmin(ez0, ez1), U = fmax(ey0, ey1) + fmax(ez0, ez1);
          const double lo_n = (-1.0 * (slab_hh + kSlabMargin)) - U, hi_n = (slab_hh + kSlabMargin) - L;
          const double ax = slab_axis[0];
          if (fabs(ax) > 1e-9)
          {
            const double inv_ax = 1.0 / ax;  // (one division instead of two: these bounds only prune, with kSlabMargin to spare)
            const double b1 = lo_n * inv_ax, b2 = hi_n * inv_ax;
            xlo = fmax(xlo, ((double) qx + fmin(b1, b2)) - kSlabMargin);
            xhi = fmin(xhi, ((double) qx + fmax(b1, b2)) + kSlabMargin);
            keep = xlo <= xhi;
          }
          else  // the slab contains the x direction (|a_x (x - qx)| < 1e-9 * 2 r): the row is inside or outside as a whole
            keep = (lo_n < 0.0) & (hi_n > 0.0);
          keep = keep | !(L == L && U == U);  // a NaN axis prunes nothing
        }
        if (keep)
        {
          const int lxr = max(lx, cell_coord(g, xlo, 0)), hxr = min(hx, cell_coord(g, xhi, 0));
          const int base = (cz * g.dim[1] + cy) * g.dim[0];
          b = gv.cell_start[base + lxr];
          len = hxr >= lxr ? gv.cell_start[base + hxr + 1] - b : 0;
        }
      }
      rt.begin[t] = b;
      rt.prefix[t + 1] = len;
    }
  }
  __syncthreads();
  if (tid < 64)  // wave 0: inclusive scan of the (<= 128) row lengths, two per lane -- and the rows without candidates are
  {              // squeezed out of the table (the walks then never meet an empty row): a second scan, of the non-empty flags
    const int nr = rt.bad ? 0 : nrows;
    const int v0 = tid < nr ? rt.prefix[tid + 1] : 0;
    const int v1 = (64 + tid) < nr ? rt.prefix[64 + tid + 1] : 0;
    const int b0 = tid < nr ? rt.begin[tid] : 0;
    const int b1 = (64 + tid) < nr ? rt.begin[64 + tid] : 0;
    const int f0 = v0 > 0 ? 1 : 0, f1 = v1 > 0 ? 1 : 0;
    int i0 = wave_incl_scan_i32(v0), i1 = wave_incl_scan_i32(v1), c0 = wave_incl_scan_i32(f0), c1 = wave_incl_scan_i32(f1);
    i1 += __builtin_amdgcn_readlane(i0, 63);  // (a v_readlane, where __shfl is a ds_bpermute round trip)
    c1 += __builtin_amdgcn_readlane(c0, 63);
    // (all reads of the table above precede these writes in the wave's program order; a row only moves down)
    if (f0)
    {
      rt.begin[c0 - 1] = b0;
      rt.prefix[c0] = i0;
      for (int jj = (i0 - v0 + 127) >> 7; (jj << 7) < i0 && jj < 64; jj++)  // the segment starts inside this row (mostly none or one)
        rt.seg_row[jj] = c0 - 1;
    }
    if (f1)
    {
      rt.begin[c1 - 1] = b1;
      rt.prefix[c1] = i1;
      for (int jj = (i1 - v1 + 127) >> 7; (jj << 7) < i1 && jj < 64; jj++)
        rt.seg_row[jj] = c1 - 1;
    }
    const int tot = __builtin_amdgcn_readlane(i1, 63), nkept = __builtin_amdgcn_readlane(c1, 63);  // lanes past nr hold zeros, so lane 63 of the second half has the totals
    if (tid == 0)
    {
      rt.prefix[0] = 0;
      rt.total = tot;
      rt.nrows = nkept;
    }
  }
  __syncthreads();
}

// Candidate j of the query -> position in the sorted array, for a thread whose j only grows: r is its row cursor.
__device__ __forceinline__ int row_advance(const RowTable& rt, int j, int& r)
{
  while (r + 1 < rt.nrows && j >= rt.prefix[r + 1])  // (bounded: a cursor can never leave the table)
    r++;
  return rt.begin[r] + (j - rt.prefix[r]);
}

// Candidate j of the query -> position in the sorted array (binary search).
__device__ __forceinline__ int row_lookup(const RowTable& rt, int j)
{
  int lo = 0, hi = rt.nrows;  // find the largest r with prefix[r] <= j
  while (hi - lo > 1)
  {
    const int mid = (lo + hi) >> 1;
    if (rt.prefix[mid] <= j)
      lo = mid;
    else
      hi = mid;
  }
  return rt.begin[lo] + (j - rt.prefix[lo]);
}

// The candidates of a ball query as ONE run of `total` flat indices, walked in segments of 128 (lane: flat index
// 128 j + lane and 128 j + 64 + lane).  A row-per-wave walk leaves a third of the lanes without a candidate (a slab row of
// the hand search holds 87 candidates on average, a Taubin row 83) and gives the waves unequal numbers of rows; flat
// segments are full except the last, and wave w takes segments w, w + 4, ... .  The row table lives in lane registers
// (lane k: rows k and k + 64) and is read with v_readlane at wave-uniform indices, so finding a segment's rows costs no LDS
// round trip: a scalar loop over the row boundaries inside the segment and two selects per boundary and lane.
struct FlatRows
{
  int pre_lo, pre_hi;  // prefix[k + 1], prefix[k + 65]: the flat index at which row k (k + 64) ENDS
  int adj_lo, adj_hi;  // begin[k] - prefix[k] (rows k, k + 64): position in the sorted array = flat index + adj
  int seg_v;           // lane j: the first row of segment j (j < 64)
  int nrows, total;

  __device__ __forceinline__ void init(const RowTable& rt, int lane)
  {
    nrows = rt.nrows;
    total = rt.total;
    pre_lo = lane < nrows ? rt.prefix[lane + 1] : total;
    pre_hi = lane + 64 < nrows ? rt.prefix[lane + 65] : total;
    adj_lo = lane < nrows ? rt.begin[lane] - rt.prefix[lane] : 0;
    adj_hi = lane + 64 < nrows ? rt.begin[lane + 64] - rt.prefix[lane + 64] : 0;
    seg_v = (lane << 7) < total ? rt.seg_row[lane] : 0;
  }
  __device__ __forceinline__ int row_end(int r) const  // r wave-uniform; (two reads and a scalar select: no branch)
  {
    const int a = __builtin_amdgcn_readlane(pre_lo, r & 63), b = __builtin_amdgcn_readlane(pre_hi, r & 63);
    return r < 64 ? a : b;
  }
  __device__ __forceinline__ int row_adj(int r) const
  {
    const int a = __builtin_amdgcn_readlane(adj_lo, r & 63), b = __builtin_amdgcn_readlane(adj_hi, r & 63);
    return r < 64 ? a : b;
  }
  __device__ __forceinline__ int segments() const { return (total + 127) >> 7; }
  // Segment j (wave-uniform; r0 = the wave's row cursor, which only grows with j): positions of the lane's two candidates
  // in the sorted array and whether it has them.
  __device__ __forceinline__ void locate(int j, int& r0, int lane, int& a0, int& a1, bool& h0, bool& h1) const
  {
    // (j and r0 derive from the wave's index, i.e. from threadIdx: tell the compiler they are wave-uniform, or the two
    // loops below become exec-masked vector loops)
    j = __builtin_amdgcn_readfirstlane(j);
    r0 = __builtin_amdgcn_readfirstlane(r0);
    const int tot = __builtin_amdgcn_readfirstlane(total), nrows = __builtin_amdgcn_readfirstlane(this->nrows);
    const int start = j << 7, end = min(start + 128, tot);
    const int i0 = start + lane, i1 = i0 + 64;
    h0 = i0 < tot;
    h1 = i1 < tot;
    if (start >= tot)  // (the read-ahead past the last segment)
    {
      a0 = a1 = 0;
      return;
    }
    if (j < 64)  // the segment's first row comes from the table build_rows left; later segments walk on from the cursor
      r0 = __builtin_amdgcn_readlane(seg_v, j);
    else
      while (r0 + 1 < nrows && row_end(r0) <= start)
        r0++;
    int adj = row_adj(r0);
    a0 = adj;
    a1 = adj;
    for (int r = r0; r + 1 < nrows;)  // the row boundaries inside the segment (build_rows leaves no empty rows)
    {
      const int b = row_end(r);
      if (b >= end)
        break;
      r++;
      adj = row_adj(r);
      a0 = i0 >= b ? adj : a0;
      a1 = i1 >= b ? adj : a1;
    }
    a0 += i0;
    a1 += i1;
  }
};

// Cyclic Jacobi of a symmetric 3x3 (oracle jacobi_sym<3>): one lane, the oracle's operation order.
__device__ inline void jacobi3_serial(double A[3][3], double V[3][3], double d[3])
{
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; sweep++)
  {
    double off = 0.0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++)
        off += A[p][q] * A[p][q];
    if (off == 0.0)
      break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++)
      {
        const double apq = A[p][q];
        if (apq == 0.0)
          continue;
        const double app = A[p][p], aqq = A[q][q];
        const double aabs = fabs(apq);
        if (sweep > 3 && (fabs(app) + aabs == fabs(app)) && (fabs(aqq) + aabs == fabs(aqq)))
        {
          A[p][q] = 0.0;
          A[q][p] = 0.0;
          continue;
        }
        const double theta = (aqq - app) / (2.0 * apq);
        double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
        if (theta < 0.0)
          t = -t;
        const double c = 1.0 / sqrt(t * t + 1.0);
        const double s = t * c;
        A[p][p] = app - t * apq;
        A[q][q] = aqq + t * apq;
        A[p][q] = 0.0;
        A[q][p] = 0.0;
        for (int k = 0; k < 3; k++)
        {
          if (k == p || k == q)
            continue;
          const double akp = A[k][p], akq = A[k][q];
          const double np_ = c * akp - s * akq;
          const double nq_ = s * akp + c * akq;
          A[k][p] = np_;
          A[p][k] = np_;
          A[k][q] = nq_;
          A[q][k] = nq_;
        }
        for (int k = 0; k < 3; k++)
        {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; i++)
    d[i] = A[i][i];
}

#endif  // __HIPCC__

}  // namespace agh

struct agh_ctx
{
  agh::Ctx c;
};

// api.hip's stages as localize.hip queues them
int preprocess_device_impl(agh_ctx* ctx, const float* d_xyz, int64_t stride_bytes, int64_t n, int64_t size_left, int dense,
  const double workspace[6], double cell_size, int64_t* n_voxels_out, void* hip_stream, bool defer_count, bool* deferred);
bool handle_thresholds(double* x1, double* x2);
int ensure_handle_buffers(agh::Ctx* c, int64_t n_hands);
int flags_to_status(agh::Ctx* c, const int32_t* flags);
