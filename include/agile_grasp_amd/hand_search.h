// hand_search.h -- HandSearch with the reference's constructor and findHands signature
// (include/agile_grasp/hand_search.h:77-85, 101-104) running on the MI355X through the agh C ABI.
//
// Differences a maintainer should know (all documented in DESIGN.md / INTEGRATION.md):
//  * no Plot member (hand_search.h:145 pulls in ROS + PCL visualisation); cloud_plot and plots_hands are accepted and
//    ignored;
//  * both camera poses are needed (the reference's HandSearch never initialises cam_tf_right_, hand_search.h:77-85):
//    pass it with setCamTfRight(), as Localization holds it (localization.h:343);
//  * uses_determinstic_normal_estimation_ (hand_search.h:84, hard-wired false) is exposed by
//    setDeterministicNormalEstimation(); `false` reproduces the reference's 50 x rand() % n subsample for ONE thread;
//  * explicit `indices` define hands_cam_source(i) = pts_cam_source(indices[i]) (the reference reads an empty vector
//    there, hand_search.cpp:166); an empty `indices` draws num_samples indices like pcl::RandomSample (hand_search.cpp:
//    36-39; PCL 1.7's algorithm restated, see randomSample) -- time-seeded like PCL unless setSampleSeed() is called.
//  * errors follow the reference's convention: a message on std::cout and an empty vector.
#ifndef AGILE_GRASP_AMD_HAND_SEARCH_H
#define AGILE_GRASP_AMD_HAND_SEARCH_H

#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../agh.h"
#include "grasp_hypothesis.h"
#include "types.h"

namespace agile_grasp_amd
{

class HandSearch
{
public:
  HandSearch(double finger_width, double hand_outer_diameter, double hand_depth, double hand_height, double init_bite,
    int num_threads, int num_samples, const Matrix4d& cam_tf_left, bool plots_hands)
    : ctx_(nullptr), cam_tf_left_(cam_tf_left), cam_tf_right_(cam_tf_left), num_threads_(num_threads),
      num_samples_(num_samples), plots_hands_(plots_hands), deterministic_(false), sample_seed_(1), device_(0),
      dirty_(true), link_(new detail::SearchLink)
  {
    agh_default_params(&params_);
    params_.finger_width = finger_width;
    params_.hand_outer_diameter = hand_outer_diameter;
    params_.hand_depth = hand_depth;
    params_.hand_height = hand_height;
    params_.init_bite = init_bite;
    (void) num_threads_;
    (void) plots_hands_;
  }
  ~HandSearch()
  {
    link_->ctx = nullptr;  // hypotheses that outlive the search see that their device state is gone
    agh_destroy(ctx_);
  }
  HandSearch(const HandSearch&) = delete;
  HandSearch& operator=(const HandSearch&) = delete;

  void setCamTfRight(const Matrix4d& cam_tf_right)
  {
    cam_tf_right_ = cam_tf_right;
    dirty_ = true;
  }
  void setDeterministicNormalEstimation(bool b)
  {
    deterministic_ = b;
    dirty_ = true;
  }
  void setRandSeed(unsigned seed)  // srand() seed of the 50-point normal subsample (glibc default: 1)
  {
    params_.rand_seed = seed;
    dirty_ = true;
  }
  void setSampleSeed(std::uint64_t seed)  // pcl::RandomSample::setSeed; default: time(NULL) like PCL
  {
    sample_seed_ = seed;
    sample_seed_set_ = true;
  }
  void setDevice(int device)
  {
    device_ = device;
    dirty_ = true;
  }
  agh_ctx* context() { return ctx_; }
  /** GraspHypothesis::getPointsForLearning / getIndicesPointsForLearningCam1 / ...Cam2 (grasp_hypothesis.h:149-170) for
   *  a hypothesis of the most recent findHands of this search, into caller-owned containers (the hypothesis' own lazy
   *  getters do the same and cache the result).  Returns false (after printing why) on error, e.g. for a hypothesis of
   *  an earlier call or of another search. */
  bool getPointsForLearning(const GraspHypothesis& h, Matrix3Xd& points_for_learning, std::vector<int>& indices_cam1,
    std::vector<int>& indices_cam2)
  {
    indices_cam1.clear();
    indices_cam2.clear();
    resize_3xn(points_for_learning, 0);
    if (!ctx_ || h.getLiveContext() != ctx_)
    {
      std::cout << " Error: the hypothesis does not come from the most recent findHands of this search\n";
      return false;
    }
    const std::size_t n_b = (std::size_t) h.getNumPointsForLearning();
    std::vector<std::int32_t> cam(n_b + 1);
    std::int64_t n = 0;
    double* dst = resize_3xn(points_for_learning, n_b);
    std::vector<double> dummy(3);
    if (agh_get_learning_points(ctx_, h.getLocalIndex(), n_b ? dst : dummy.data(), cam.data(), (std::int64_t) n_b, &n) != AGH_OK)
    {
      std::cout << " Error in agh_get_learning_points: " << agh_last_error(ctx_) << "\n";
      resize_3xn(points_for_learning, 0);
      return false;
    }
    for (std::size_t k = 0; k < n_b; k++)  // rotating_hand.cpp:143-151
      if (cam[k] == 0)
        indices_cam1.push_back((int) k);
      else if (cam[k] == 1)
        indices_cam2.push_back((int) k);
    return true;
  }
  // pcl::RandomSample<PointT>::applyFilter(std::vector<int>&) as hand_search.cpp:36-39 uses it (setSample(num_samples_),
  // no input indices => all points).  PCL is THIRD PARTY and absent from /root/reference; this restates PCL 1.7's
  // filters/impl/random_sample.hpp: every index if num_samples >= N, else std::srand(seed) and Vitter's "Algorithm A"
  // (J. S. Vitter, ACM TOMS 13(1), 1987) with unifRand() = (float) (rand() / double(RAND_MAX)); the result is ascending.
  // PCL seeds with time(NULL) by default -- so does this unless setSampleSeed() was called -- hence the reference's own
  // samples are never reproducible and explicit `indices` are what parity tests use.  The srand() side effect on the
  // process-wide rand() stream is kept (it is the host's libc, as in the reference).
  static std::vector<std::int32_t> randomSample(std::int64_t n_points, int num_samples, unsigned seed)
  {
    std::vector<std::int32_t> idx;
    unsigned N = (unsigned) n_points;
    const unsigned sample = num_samples < 0 ? 0u : (unsigned) num_samples;
    if (sample >= N)
    {
      idx.resize((std::size_t) N);
      for (unsigned i = 0; i < N; i++)
        idx[i] = (std::int32_t) i;
      return idx;
    }
    if (sample == 0)
      return idx;
    idx.resize((std::size_t) sample);
    std::srand(seed);
    unsigned top = N - sample, i = 0, index = 0;
    for (std::size_t n = sample; n >= 2; n--)
    {
      const float V = (float) (std::rand() / double(RAND_MAX));
      unsigned S = 0;
      float quot = (float) top / (float) N;
      while (quot > V)
      {
        S++;
        top--;
        N--;
        quot = quot * (float) top / (float) N;
      }
      index += S;
      idx[i++] = (std::int32_t) index++;
      N--;
    }
    index += N * (unsigned) (float) (std::rand() / double(RAND_MAX));  // (PCL's cast truncates the variate, not the product)
    idx[i++] = (std::int32_t) index++;
    return idx;
  }

  /** The sample indices of the most recent findHands (the given ones, or the ones drawn for an empty list). */
  const std::vector<std::int32_t>& getLastSampleIndices() const { return last_samples_; }

  /** Multi-GPU (one process -- or, with joinLocalCommunicator, one host thread -- per GPU): after this call findHands is a
   *  COLLECTIVE: every rank passes the same cloud and the same `indices`, searches the slice [rank S / n, (rank + 1) S / n)
   *  of the samples (the reference's OpenMP fan-out over samples, hand_search.cpp:78,136, across GPUs) and receives the
   *  complete list through one RCCL all-gather issued by the library (agh_find_hands_sharded).  `id`: the bytes of
   *  agh_comm_unique_id() from ONE rank, handed to the others out of band.  Returns false (after printing why) on error. */
  bool joinCommunicator(int rank, int n_ranks, const std::uint8_t id[AGH_COMM_ID_BYTES])
  {
    if (!ensureContext())
      return false;
    if (agh_comm_init(ctx_, rank, n_ranks, id) != AGH_OK)
    {
      std::cout << " Error in agh_comm_init: " << agh_last_error(ctx_) << "\n";
      return false;
    }
    sharded_ = true;
    return true;
  }
  /** The same for searches that live in ONE process (one host thread each; they may share a GPU): the exchange is device
   *  copies instead of RCCL -- how the sharded schedule is validated on a single-GPU machine. */
  static bool joinLocalCommunicator(const std::vector<HandSearch*>& searches)
  {
    std::vector<agh_ctx*> ctxs;
    for (std::size_t i = 0; i < searches.size(); i++)
    {
      if (!searches[i] || !searches[i]->ensureContext())
        return false;
      ctxs.push_back(searches[i]->ctx_);
    }
    if (agh_comm_init_local(ctxs.data(), (std::int32_t) ctxs.size()) != AGH_OK)
    {
      std::cout << " Error in agh_comm_init_local\n";
      return false;
    }
    for (std::size_t i = 0; i < searches.size(); i++)
      searches[i]->sharded_ = true;
    return true;
  }
  bool isSharded() const { return sharded_; }

  /** Training runs (src/nodes/train.cpp): every findHands(calculates_antipodal = true) also attaches the three
   *  instance images to its hypotheses (GraspHypothesis::getTrainingImage), the input of Learning::train*. */
  void setKeepsTrainingImages(bool b) { keeps_training_images_ = b; }
  /** Default true: every hypothesis carries its packed 80x100 occupancy image (1000 bytes), which makes
   *  Learning::classify independent of this search's device state (any list, any time, like the reference's).  With
   *  false, only hypotheses of the most recent findHands can be classified (no image download). */
  void setKeepsImages(bool b) { keeps_images_ = b; }

  std::vector<GraspHypothesis> findHands(const PointCloud::Ptr cloud, const VectorXi& pts_cam_source,
    const std::vector<int>& indices, const PointCloud::Ptr cloud_plot, bool calculates_antipodal, bool uses_clustering)
  {
    (void) cloud_plot;
    (void) uses_clustering;
    std::vector<GraspHypothesis> hand_list;
    if (!cloud || cloud->size() == 0)
    {
      std::cout << "Input cloud is empty!\n";
      return hand_list;
    }
    if (!ensureContext())
      return hand_list;
    const std::int64_t n = (std::int64_t) cloud->size();
    std::vector<std::int32_t> cam((std::size_t) n, 0);
    for (std::int64_t i = 0; i < n && i < (std::int64_t) pts_cam_source.size(); i++)
      cam[(std::size_t) i] = pts_cam_source((std::size_t) i);
    int rc = agh_set_cloud(ctx_, &cloud->points[0].x, (std::int64_t) sizeof(cloud->points[0]), cam.data(), n);
    if (rc != AGH_OK)
      return fail("agh_set_cloud");
    return findHandsInSearchedCloud(indices, calculates_antipodal);
  }

  /** The head of Localization::localizeHands on the GPU (localization.cpp:17-45: camera ids, NaN removal, workspace
   *  box, per-camera voxelisation) followed by the search-structure build.  The voxelised cloud stays on the device as
   *  the cloud findHandsInSearchedCloud works on; a host copy is returned for the caller (plots, sample indices).
   *  @return false (after printing) on error */
  bool preprocess(const PointCloud::Ptr& cloud_in, int size_left, const VectorXd& workspace, double cell_size,
    PointCloud::Ptr& voxels_out, VectorXi& pts_cam_source_out)
  {
    if (!ensureContext())
      return false;
    double ws[6];
    for (int i = 0; i < 6; i++)
      ws[i] = workspace(i);
    std::int64_t nv = 0;
    const std::int64_t n = (std::int64_t) cloud_in->size();
    int rc = agh_preprocess(ctx_, n > 0 ? &cloud_in->points[0].x : nullptr, (std::int64_t) sizeof(cloud_in->points[0]), n,
      (std::int64_t) size_left, cloud_is_dense(*cloud_in) ? 1 : 0, ws, cell_size, &nv);
    if (rc != AGH_OK)
    {
      fail("agh_preprocess");
      return false;
    }
    std::vector<float> xyz(3 * (std::size_t) nv + 3);
    std::vector<std::int32_t> cam((std::size_t) nv + 1);
    if (agh_get_cloud(ctx_, xyz.data(), cam.data(), nv) < 0)
    {
      fail("agh_get_cloud");
      return false;
    }
    voxels_out.reset(new PointCloud);
    voxels_out->points.resize((std::size_t) nv);
    pts_cam_source_out = VectorXi((std::size_t) nv);
    for (std::int64_t i = 0; i < nv; i++)
    {
      voxels_out->points[(std::size_t) i].x = xyz[3 * (std::size_t) i];
      voxels_out->points[(std::size_t) i].y = xyz[3 * (std::size_t) i + 1];
      voxels_out->points[(std::size_t) i].z = xyz[3 * (std::size_t) i + 2];
      pts_cam_source_out((std::size_t) i) = cam[(std::size_t) i];
    }
    searched_n_ = nv;
    return true;
  }

  /** The online chain of grasp_localizer.cpp:95-103 -- preprocessing, search, Learning::classify, HandleSearch -- as ONE device
   *  call with one synchronisation (agh_localize): the raw capture goes in, the hands the classifier kept, the handles and
   *  their inlier lists come out as records.  indices empty: num_samples indices are drawn ON THE DEVICE (one per stratum of
   *  the voxelised cloud, seeded like randomSample: setSampleSeed or the clock).
   *  @return false (after printing) on error */
  bool localize(const PointCloud::Ptr& cloud_in, int size_left, const VectorXd& workspace, double cell_size,
    const std::vector<int>& indices, const std::string& svm_filename, int min_inliers, double min_length,
    std::vector<agh_hypothesis>& hands_out, std::vector<agh_handle>& handles_out, std::vector<std::int32_t>& inliers_out)
  {
    hands_out.clear();
    handles_out.clear();
    inliers_out.clear();
    return localizeBegin(cloud_in, size_left, workspace, cell_size, indices, svm_filename, min_inliers, min_length) &&
           localizeEnd(hands_out, handles_out, inliers_out);
  }

  /** The same chain as two calls (agh_localize_begin / agh_localize_end), for a caller that holds the NEXT capture while this
   *  one is searched: between the two, localizeStage(next) uploads it on a second stream under this capture's kernels, and the
   *  localizeBegin that is later handed the same cloud object finds it on the device.  One chain may be in flight; `cloud_in`
   *  must stay alive and unchanged until localizeEnd has returned. */
  bool localizeBegin(const PointCloud::Ptr& cloud_in, int size_left, const VectorXd& workspace, double cell_size,
    const std::vector<int>& indices, const std::string& svm_filename, int min_inliers, double min_length)
  {
    if (!ensureContext())
      return false;
    if (agh_load_svm_file(ctx_, svm_filename.c_str()) != AGH_OK)
    {
      std::cout << " Exception: " << agh_last_error(ctx_) << "\n";  // learning.cpp:187-191
      return false;
    }
    agh_localize_params lp;
    lp.size_left = (std::int64_t) size_left;
    lp.dense = cloud_is_dense(*cloud_in) ? 1 : 0;
    lp.classify = 1;
    for (int i = 0; i < 6; i++)
      lp.workspace[i] = workspace(i);
    lp.cell_size = cell_size;
    std::vector<std::int32_t> idx(indices.begin(), indices.end());  // (copied by agh_localize_begin)
    lp.sample_idx = idx.empty() ? nullptr : idx.data();
    lp.n_samples = idx.empty() ? (std::int64_t) (num_samples_ < 0 ? 0 : num_samples_) : (std::int64_t) idx.size();
    lp.sample_seed = sample_seed_set_ ? (std::uint64_t) sample_seed_ : (std::uint64_t) std::time(nullptr);
    lp.min_inliers = min_inliers;
    lp.reserved = 0;
    lp.min_length = min_length;
    loc_cap_ = lp.n_samples * 8 < 8192 ? lp.n_samples * 8 + 1 : 8193;
    last_samples_.assign((std::size_t) lp.n_samples, 0);
    const std::int64_t n = (std::int64_t) cloud_in->size();
    if (agh_localize_begin(ctx_, n > 0 ? &cloud_in->points[0].x : nullptr, (std::int64_t) sizeof(cloud_in->points[0]), n, &lp) != AGH_OK)
    {
      fail("agh_localize_begin");
      return false;
    }
    return true;
  }

  /** agh_localize_stage: the NEXT capture up, beside the chain in flight (keep `next` alive and unchanged until the
   *  localizeEnd of the chain that searches it has returned). */
  bool localizeStage(const PointCloud::Ptr& next)
  {
    if (!ensureContext() || !next)
      return false;
    const std::int64_t n = (std::int64_t) next->size();
    if (agh_localize_stage(ctx_, n > 0 ? &next->points[0].x : nullptr, (std::int64_t) sizeof(next->points[0]), n) != AGH_OK)
    {
      fail("agh_localize_stage");
      return false;
    }
    return true;
  }

  bool localizeEnd(std::vector<agh_hypothesis>& hands_out, std::vector<agh_handle>& handles_out, std::vector<std::int32_t>& inliers_out)
  {
    hands_out.clear();
    handles_out.clear();
    inliers_out.clear();
    if (!ctx_)
      return false;
    const std::int64_t cap = loc_cap_;
    hands_out.resize((std::size_t) cap);
    handles_out.resize((std::size_t) cap);
    inliers_out.resize((std::size_t) cap);
    agh_localize_result res;
    const int rc = agh_localize_end(ctx_, handles_out.data(), cap, inliers_out.data(), cap, hands_out.data(), cap,
      last_samples_.empty() ? nullptr : last_samples_.data(), &res);
    if (rc != AGH_OK)
    {
      hands_out.clear();
      handles_out.clear();
      inliers_out.clear();
      fail("agh_localize");
      return false;
    }
    hands_out.resize((std::size_t) res.n_hands);
    handles_out.resize((std::size_t) res.n_handles);
    inliers_out.resize((std::size_t) res.n_inlier_idx);
    searched_n_ = res.n_voxels;
    return true;
  }

  /** hand_search.cpp:31-62 on the cloud the context already holds (after findHands' upload or preprocess). */
  std::vector<GraspHypothesis> findHandsInSearchedCloud(const std::vector<int>& indices, bool calculates_antipodal)
  {
    std::vector<GraspHypothesis> hand_list;
    if (!ctx_)
      return hand_list;
    const std::int64_t n = (std::int64_t) agh_get_cloud(ctx_, nullptr, nullptr, 0);
    if (n <= 0)
    {
      std::cout << "Input cloud is empty!\n";
      return hand_list;
    }
    std::vector<std::int32_t> idx;
    if (indices.empty())
    {
      std::cout << "Generating uniform random indices ...\n";  // hand_search.cpp:34
      idx = randomSample(n, num_samples_, sample_seed_set_ ? (unsigned) sample_seed_ : (unsigned) std::time(nullptr));
    }
    else
      idx.assign(indices.begin(), indices.end());
    last_samples_ = idx;
    if (calculates_antipodal)
      std::cout << "Calculating normals for all points\n";  // hand_search.cpp:19
    std::cout << "Estimating local axes ...\nFinding hand poses ...\n";  // hand_search.cpp:52,58
    const bool training = keeps_training_images_ && calculates_antipodal;
    if (agh_set_training_images(ctx_, training ? 1 : 0) != AGH_OK)
      return fail("agh_set_training_images");
    std::vector<agh_hypothesis> out(8 * idx.size() + 1);
    std::int64_t n_out = 0;
    const int rc = sharded_
      ? agh_find_hands_sharded(ctx_, idx.data(), (std::int64_t) idx.size(), calculates_antipodal ? 1 : 0, out.data(),
          (std::int64_t) out.size(), &n_out)
      : agh_find_hands(ctx_, idx.data(), (std::int64_t) idx.size(), calculates_antipodal ? 1 : 0, out.data(),
          (std::int64_t) out.size(), &n_out);
    if (rc != AGH_OK)
      return fail(sharded_ ? "agh_find_hands_sharded" : "agh_find_hands");
    if (sharded_)
    {
      // every rank holds the complete list; the device-side state (images, points) of a hypothesis lives on the rank that
      // searched its sample, so the hypotheses carry no image here and Learning::classify goes through the collective
      // agh_classify_sharded (see learning.h)
      // The records of THIS rank's samples (a contiguous run of the merged list: slices are contiguous and the list is
      // sample-major) keep their position in this rank's own device-side list, so getPointsForLearning() and the index
      // getters work for them; the others say that their points live on another rank.
      std::int32_t rank = 0, n_ranks = 1;
      std::int64_t lo = 0, hi = 0;
      agh_comm_rank(ctx_, &rank, &n_ranks);
      agh_shard_slice((std::int64_t) idx.size(), rank, n_ranks, &lo, &hi);
      std::int64_t first = 0;
      while (first < n_out && out[(std::size_t) first].sample < lo)
        first++;
      hand_list.reserve((std::size_t) n_out);
      for (std::int64_t i = 0; i < n_out; i++)
      {
        const std::int64_t smp = out[(std::size_t) i].sample;
        hand_list.push_back(GraspHypothesis(out[(std::size_t) i], (long) i, link_, (smp >= lo && smp < hi) ? (long) (i - first) : -1L));
      }
      std::cout << " Found " << hand_list.size() << " robot hand poses\n";
      return hand_list;
    }
    hand_list.reserve((std::size_t) n_out);
    for (std::int64_t i = 0; i < n_out; i++)
      hand_list.push_back(GraspHypothesis(out[(std::size_t) i], (long) i, link_));
    if (keeps_images_ && n_out > 0)
    {
      std::shared_ptr<std::vector<std::uint32_t> > block(new std::vector<std::uint32_t>((std::size_t) n_out * 250));
      if (agh_get_packed_images(ctx_, block->data(), n_out) != (int) n_out)
        return fail("agh_get_packed_images");
      for (std::int64_t i = 0; i < n_out; i++)
        hand_list[(std::size_t) i].setImage(block, (std::size_t) i * 250);
    }
    if (training && n_out > 0)
    {
      std::shared_ptr<std::vector<std::uint32_t> > block(new std::vector<std::uint32_t>((std::size_t) n_out * 750));
      if (agh_get_training_images(ctx_, block->data(), n_out) != (int) n_out)
        return fail("agh_get_training_images");
      for (std::int64_t i = 0; i < n_out; i++)
        hand_list[(std::size_t) i].setTrainingImages(block, (std::size_t) i * 750);
    }
    std::cout << " Found " << hand_list.size() << " robot hand poses\n";  // hand_search.cpp:203
    return hand_list;
  }

private:
  std::vector<GraspHypothesis> fail(const char* what)
  {
    std::cout << " Error in " << what << ": " << agh_last_error(ctx_) << "\n";
    return std::vector<GraspHypothesis>();
  }

  bool ensureContext()
  {
    if (ctx_ && !dirty_)
      return true;
    link_->ctx = nullptr;  // hypotheses of the old context are cut loose
    link_.reset(new detail::SearchLink);
    agh_destroy(ctx_);
    ctx_ = nullptr;
    for (int r = 0; r < 3; r++)
    {
      params_.cam_origin[0][r] = mat4(cam_tf_left_, r, 3);   // hand_search.cpp:72-74 -> quadric.cpp:8-11
      params_.cam_origin[1][r] = mat4(cam_tf_right_, r, 3);
    }
    params_.normals_mode = deterministic_ ? AGH_NORMALS_DETERMINISTIC : AGH_NORMALS_RAND50;
    params_.device = device_;
    const int rc = agh_create(&params_, &ctx_);
    if (rc != AGH_OK)
    {
      std::cout << " Error: cannot create the MI355X grasp-search context: " << agh_last_error(nullptr) << "\n";
      ctx_ = nullptr;
      return false;
    }
    dirty_ = false;
    link_->ctx = ctx_;
    return true;
  }

  agh_ctx* ctx_;
  std::int64_t searched_n_ = 0;
  std::int64_t loc_cap_ = 1;  // room for the results of the chain localizeBegin queued
  agh_params params_;
  Matrix4d cam_tf_left_, cam_tf_right_;
  int num_threads_, num_samples_;
  bool plots_hands_, deterministic_;
  std::uint64_t sample_seed_;
  bool sample_seed_set_ = false;
  std::vector<std::int32_t> last_samples_;
  int device_;
  bool dirty_;
  bool keeps_training_images_ = false;
  bool keeps_images_ = true;
  bool sharded_ = false;
  std::shared_ptr<detail::SearchLink> link_;
};

namespace detail
{
/** A device context for work that follows a search (classification, training, handle search) when the caller -- like the
 *  reference's Learning(int) and HandleSearch() -- names none: the given search's, else one a hypothesis of the list
 *  still links to, else a context of the finder's own (created on first use with default parameters). */
class ContextFinder
{
public:
  agh_ctx* find(HandSearch* search, const std::vector<GraspHypothesis>& hands_list)
  {
    if (search && search->context())
      return search->context();
    for (std::size_t i = 0; i < hands_list.size(); i++)
      if (hands_list[i].getAnyContext())
        return hands_list[i].getAnyContext();
    if (!own_)
    {
      agh_params p;
      agh_default_params(&p);
      agh_ctx* c = nullptr;
      if (agh_create(&p, &c) != AGH_OK)
      {
        std::cout << " Error: cannot create the MI355X grasp-search context: " << agh_last_error(nullptr) << "\n";
        return nullptr;
      }
      own_.reset(c, agh_destroy);
    }
    return own_.get();
  }

private:
  std::shared_ptr<agh_ctx> own_;
};
}  // namespace detail

}  // namespace agile_grasp_amd
#endif
