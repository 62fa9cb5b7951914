// localization.h -- the Localization facade (include/agile_grasp/localization.h:64-351) over the MI355X hand search.
//
// Kept: constructors, every setter, localizeHands(cloud, size_left, indices, calculates_antipodal, uses_clustering) and
// its two PCD-filename overloads (pcd_io.h reads PCD files when PCL is absent), predictAntipodalHands(hand_list,
// svm_filename), findHandles(hand_list, min_inliers, min_length), filterHands.  The preprocessing that precedes the hot
// path (NaN removal, workspace box, per-camera 3 mm voxelisation: localization.cpp:25-45, 216-355; its output ORDER
// defines the point indices the search works on) runs on the GPU as well (agh_preprocess).
// Not carried over: the RANSAC table-plane removal behind uses_clustering (localization.cpp:51-98,
// pcl::SACSegmentation; training path only) -- localizeHands(..., uses_clustering = true) prints an error and returns an
// empty list -- and the Plot members.
#ifndef AGILE_GRASP_AMD_LOCALIZATION_H
#define AGILE_GRASP_AMD_LOCALIZATION_H

#include <cmath>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "hand_search.h"
#include "handle_search.h"
#include "learning.h"
#include "pcd_io.h"

namespace agile_grasp_amd
{

class Localization
{
public:
  Localization() : num_threads_(1), num_samples_(2000), filters_boundaries_(false), plotting_mode_(0) { init(); }
  Localization(int num_threads, bool filters_boundaries, int plotting_mode)
    : num_threads_(num_threads), num_samples_(2000), filters_boundaries_(filters_boundaries), plotting_mode_(plotting_mode)
  {
    init();
  }

  // ---- setters (localization.h:148-259) ----
  void setCameraTransforms(const Matrix4d& cam_tf_left, const Matrix4d& cam_tf_right)
  {
    cam_tf_left_ = cam_tf_left;
    cam_tf_right_ = cam_tf_right;
    search_.reset();
  }
  const Matrix4d& getCameraTransform(bool is_left) { return is_left ? cam_tf_left_ : cam_tf_right_; }
  void setWorkspace(const VectorXd& workspace) { workspace_ = workspace; }
  void setNumSamples(int num_samples)
  {
    num_samples_ = num_samples;
    search_.reset();
  }
  // stored but never forwarded, exactly like the reference (localization.h:189-201 vs localization.cpp:111-112)
  void setNeighborhoodRadiusHands(double r) { nn_radius_hands_ = r; }
  void setNeighborhoodRadiusTaubin(double r) { nn_radius_taubin_ = r; }
  void setFingerWidth(double v)
  {
    finger_width_ = v;
    search_.reset();
  }
  void setHandDepth(double v)
  {
    hand_depth_ = v;
    search_.reset();
  }
  void setHandOuterDiameter(double v)
  {
    hand_outer_diameter_ = v;
    search_.reset();
  }
  void setInitBite(double v)
  {
    init_bite_ = v;
    search_.reset();
  }
  void setHandHeight(double v)
  {
    hand_height_ = v;
    search_.reset();
  }
  // additions: the reference hard-wires these inside HandSearch
  void setDeterministicNormalEstimation(bool b)
  {
    deterministic_ = b;
    search_.reset();
  }
  void setDevice(int device)
  {
    device_ = device;
    search_.reset();
  }
  /** Training runs (src/nodes/train.cpp): localizeHands(..., calculates_antipodal = true, ...) then returns hypotheses
   *  that carry their three instance images, the input of Learning::train / trainBalanced. */
  void setKeepsTrainingImages(bool b)
  {
    keeps_training_images_ = b;
    if (search_)
      search_->setKeepsTrainingImages(b);
  }
  /** The search the hypotheses came from: what this adapter's Learning is constructed on (train.cpp:122). */
  HandSearch& getHandSearch()
  {
    ensureSearch();
    return *search_;
  }

  /** localization.cpp:3-140 */
  std::vector<GraspHypothesis> localizeHands(const PointCloud::Ptr& cloud_in, int size_left,
    const std::vector<int>& indices, bool calculates_antipodal, bool uses_clustering)
  {
    std::vector<GraspHypothesis> hand_list;
    if (size_left == 0 || !cloud_in || cloud_in->size() == 0)
    {
      std::cout << "Input cloud is empty!\n";
      std::cout << size_left << std::endl;
      return hand_list;
    }
    // localization.cpp:17-45 on the GPU (agh_preprocess): camera id = (position >= size_left), removal of non-finite
    // points WITHOUT re-indexing the camera ids (the reference's behaviour), workspace box, per-camera 3 mm voxels in
    // lexicographic order.
    if (uses_clustering)
    {
      // localization.cpp:51-98 removes the table plane with pcl::SACSegmentation (RANSAC) before the search; that step
      // is not part of this build.  Searching the unsegmented cloud instead would label hands on the table and build a
      // different training set while returning normally, so the request fails like the reference's other errors do.
      std::cout << " Error: uses_clustering (table-plane removal, pcl::SACSegmentation) is not available in this build; "
                   "remove the plane before localizeHands or pass uses_clustering = false\n";
      return hand_list;
    }
    std::cout << "Generating camera sources for " << cloud_in->size() << " points ...\n";
    std::cout << "Filtering workspace ...\nVoxelizing point cloud\n";
    ensureSearch();
    PointCloud::Ptr voxels;
    VectorXi pts_cam_source;
    if (!search_->preprocess(cloud_in, size_left, workspace_, 0.003, voxels, pts_cam_source))
      return hand_list;
    std::cout << " Created " << voxels->points.size() << " voxels\n";
    remove_nan_in_place(*cloud_in);  // localization.cpp:27 filters the caller's cloud in place
    hand_list = search_->findHandsInSearchedCloud(indices, calculates_antipodal);
    if (filters_boundaries_)
    {
      std::cout << "Filtering out hands close to workspace boundaries ...\n";
      hand_list = filterHands(hand_list);
      std::cout << " # hands left: " << hand_list.size() << "\n";
    }
    last_cloud_ = voxels;
    last_cam_ = pts_cam_source;
    return hand_list;
  }

  /** localization.cpp:168-173 */
  std::vector<GraspHypothesis> localizeHands(const std::string& pcd_filename_left, const std::string& pcd_filename_right,
    bool calculates_antipodal = false, bool uses_clustering = false)
  {
    return localizeHands(pcd_filename_left, pcd_filename_right, std::vector<int>(), calculates_antipodal, uses_clustering);
  }

  /** localization.cpp:175-212 */
  std::vector<GraspHypothesis> localizeHands(const std::string& pcd_filename_left, const std::string& pcd_filename_right,
    const std::vector<int>& indices, bool calculates_antipodal = false, bool uses_clustering = false)
  {
    PointCloud::Ptr cloud_left(new PointCloud);
    if (loadPCDFile(pcd_filename_left, *cloud_left) == -1)
    {
      std::cout << "Couldn't read pcd_filename_left file: " << pcd_filename_left << " \n";
      return std::vector<GraspHypothesis>();
    }
    if (pcd_filename_right.length() > 0)
      std::cout << "Loaded left point cloud with " << cloud_left->size() << " data points.\n";
    else
      std::cout << "Loaded point cloud with " << cloud_left->size() << " data points.\n";
    PointCloud::Ptr cloud_right(new PointCloud);
    if (pcd_filename_right.length() > 0)
    {
      if (loadPCDFile(pcd_filename_right, *cloud_right) == -1)
      {
        std::cout << "Couldn't read pcd_filename_left file: " << pcd_filename_right << " \n";
        return std::vector<GraspHypothesis>();
      }
      std::cout << "Loaded right point cloud with " << cloud_right->size() << " data points.\n";
    }
    std::cout << "Concatenating point clouds ...\n";
    PointCloud::Ptr cloud(new PointCloud);
    *cloud = *cloud_left;  // *cloud_left + *cloud_right (pcl::PointCloud::operator+ also ANDs is_dense)
    cloud->points.insert(cloud->points.end(), cloud_right->points.begin(), cloud_right->points.end());
    cloud->is_dense = cloud_left->is_dense && cloud_right->is_dense;
    return localizeHands(cloud, (int) cloud_left->size(), indices, calculates_antipodal, uses_clustering);
  }

  /** localization.cpp:142-167 */
  std::vector<GraspHypothesis> predictAntipodalHands(const std::vector<GraspHypothesis>& hand_list,
    const std::string& svm_filename)
  {
    Learning learn(num_threads_);  // localization.cpp:146
    Matrix3Xd cams_mat;  // the images were rasterised with both camera origins already (localization.cpp:147-150)
    std::vector<GraspHypothesis> antipodal_hands = learn.classify(hand_list, svm_filename, cams_mat);
    std::cout << antipodal_hands.size() << " antipodal hand configurations found\n";
    return antipodal_hands;
  }

  /** localization.cpp:390-409 (the plotting branches aside) */
  std::vector<Handle> findHandles(const std::vector<GraspHypothesis>& hand_list, int min_inliers, double min_length)
  {
    HandleSearch handle_search;  // localization.cpp:392
    return handle_search.findHandles(hand_list, min_inliers, min_length);
  }

  /** Additional: what GraspLocalizer::localizeGrasps runs per cloud (grasp_localizer.cpp:95-103) --
   *      hands = localizeHands(cloud, size_left, indices, false, false);
   *      antipodal_hands = predictAntipodalHands(hands, svm_file_name);
   *      handles = findHandles(antipodal_hands, min_inliers, 0.005);
   *  -- as ONE call into the device library with one synchronisation (agh_localize: no host round trip of the hypotheses
   *  between the stages).  Same handles as the three calls on the same sample indices; with `indices` empty the samples are
   *  drawn on the device (see HandSearch::localize).  The hands the classifier kept come back through `antipodal_hands`
   *  (Handle::getHandList of every handle is that list).  With setFiltersBoundaries(true) -- a host-side filter BETWEEN the
   *  search and the classifier -- the three separate calls are made. */
  std::vector<Handle> localizeHandles(const PointCloud::Ptr& cloud_in, int size_left, const std::vector<int>& indices,
    const std::string& svm_filename, int min_inliers, double min_length, std::vector<GraspHypothesis>* antipodal_hands = nullptr)
  {
    if (antipodal_hands)
      antipodal_hands->clear();
    if (!localizeHandlesBegin(cloud_in, size_left, indices, svm_filename, min_inliers, min_length))
      return std::vector<Handle>();
    return localizeHandlesEnd(antipodal_hands);
  }

  /** Additional: the same chain for a node that already holds the NEXT capture while this one is searched
   *  (GraspLocalizer::cloud_callback queues clouds, grasp_localizer.cpp:55-75):
   *      loc.localizeHandlesBegin(cloud_k, size_left, indices, svm, min_inliers, 0.005);   // queued, not waited for
   *      loc.stageNextCloud(cloud_k1);                                                     // its upload runs under cloud k's kernels
   *      handles = loc.localizeHandlesEnd(&antipodal_hands);                               // the one synchronisation
   *      loc.localizeHandlesBegin(cloud_k1, ...);                                          // finds cloud k + 1 on the device
   *  Same results as localizeHandles.  The clouds must stay alive and unchanged until the localizeHandlesEnd of their chain
   *  has returned (which then filters NaNs out of the searched cloud in place, as localization.cpp:27 does). */
  bool localizeHandlesBegin(const PointCloud::Ptr& cloud_in, int size_left, const std::vector<int>& indices,
    const std::string& svm_filename, int min_inliers, double min_length)
  {
    pending_cloud_ = PointCloud::Ptr();
    pending_three_calls_ = false;
    if (filters_boundaries_)  // (a host-side filter between the search and the classifier: the three calls, at End)
    {
      pending_cloud_ = cloud_in;
      pending_three_calls_ = true;
      pending_size_left_ = size_left;
      pending_indices_ = indices;
      pending_svm_ = svm_filename;
      pending_min_inliers_ = min_inliers;
      pending_min_length_ = min_length;
      return true;
    }
    if (size_left == 0 || !cloud_in || cloud_in->size() == 0)
    {
      std::cout << "Input cloud is empty!\n";
      std::cout << size_left << std::endl;
      return false;
    }
    std::ifstream f(svm_filename.c_str());
    if (!f.good())
    {
      std::cout << " Error: File " << svm_filename << " does not exist!\n";  // learning.cpp:172-178
      return false;
    }
    ensureSearch();
    if (!search_->localizeBegin(cloud_in, size_left, workspace_, 0.003, indices, svm_filename, min_inliers, min_length))
      return false;
    pending_cloud_ = cloud_in;
    return true;
  }

  /** agh_localize_stage through the adapter: the next capture up, beside the chain in flight */
  bool stageNextCloud(const PointCloud::Ptr& next)
  {
    if (filters_boundaries_ || !next || next->size() == 0)
      return false;
    ensureSearch();
    return search_->localizeStage(next);
  }

  std::vector<Handle> localizeHandlesEnd(std::vector<GraspHypothesis>* antipodal_hands = nullptr)
  {
    std::vector<Handle> handle_list;
    if (antipodal_hands)
      antipodal_hands->clear();
    if (!pending_cloud_)
      return handle_list;
    PointCloud::Ptr cloud_in = pending_cloud_;
    pending_cloud_ = PointCloud::Ptr();
    if (pending_three_calls_)
    {
      std::vector<GraspHypothesis> kept = predictAntipodalHands(localizeHands(cloud_in, pending_size_left_, pending_indices_, false, false), pending_svm_);
      if (antipodal_hands)
        *antipodal_hands = kept;
      return findHandles(kept, pending_min_inliers_, pending_min_length_);
    }
    std::vector<agh_hypothesis> hands;
    std::vector<agh_handle> handles;
    std::vector<std::int32_t> idx;
    if (!search_->localizeEnd(hands, handles, idx))
      return handle_list;
    remove_nan_in_place(*cloud_in);  // localization.cpp:27 filters the caller's cloud in place
    std::shared_ptr<std::vector<GraspHypothesis> > kept(new std::vector<GraspHypothesis>());
    kept->reserve(hands.size());
    for (std::size_t i = 0; i < hands.size(); i++)
    {
      kept->push_back(GraspHypothesis(hands[i], -1));
      kept->back().setFullAntipodal(true);  // learning.cpp:240
    }
    std::cout << kept->size() << " antipodal hand configurations found\n";  // localization.cpp:153
    if (antipodal_hands)
      *antipodal_hands = *kept;
    const std::shared_ptr<const std::vector<GraspHypothesis> > shared = kept;
    for (std::size_t h = 0; h < handles.size(); h++)
    {
      const agh_handle& r = handles[h];
      std::vector<int> in(idx.begin() + r.first_inlier, idx.begin() + r.first_inlier + r.n_inliers);
      handle_list.push_back(Handle(r, shared, in));
      std::cout << "handle found with " << in.size() << " inliers\n";  // handle_search.cpp:73
    }
    std::cout << "Handle Search\n " << handle_list.size() << " handles found\n";  // handle_search.cpp:82-84
    return handle_list;
  }

  /** the voxelised cloud and camera ids the last localizeHands searched (what the reference plots) */
  const PointCloud::Ptr& getSearchedCloud() const { return last_cloud_; }
  const VectorXi& getSearchedCamSource() const { return last_cam_; }

  /** localization.cpp:364-388 */
  std::vector<GraspHypothesis> filterHands(const std::vector<GraspHypothesis>& hand_list) const
  {
    const double MIN_DIST = 0.02;
    std::vector<GraspHypothesis> filtered;
    for (std::size_t i = 0; i < hand_list.size(); i++)
    {
      const Vector3d& center = hand_list[i].getGraspSurface();
      const int n_ws = (int) workspace_.size();  // (Eigen::Index is signed, the stand-in's size() is not)
      int k;
      for (k = 0; k < n_ws; k++)
        if (std::fabs(center((int) std::floor(k / 2.0)) - workspace_(k)) < MIN_DIST)
          break;
      if (k == n_ws)
        filtered.push_back(hand_list[i]);
    }
    return filtered;
  }

private:
  void init()
  {
    workspace_ = VectorXd(6);
    const double ws[6] = { -1.0, 1.0, -1.0, 1.0, -1.0, 1.0 };  // find_grasps.cpp:19
    for (int i = 0; i < 6; i++)
      workspace_(i) = ws[i];
    finger_width_ = 0.01;  // find_grasps.cpp:13-17
    hand_outer_diameter_ = 0.09;
    hand_depth_ = 0.06;
    init_bite_ = 0.01;
    hand_height_ = 0.02;
    nn_radius_taubin_ = 0.03;
    nn_radius_hands_ = 0.08;
    deterministic_ = false;
    device_ = 0;
  }
  void ensureSearch()
  {
    if (search_)
      return;
    search_.reset(new HandSearch(finger_width_, hand_outer_diameter_, hand_depth_, hand_height_, init_bite_,
      num_threads_, num_samples_, cam_tf_left_, false));
    search_->setCamTfRight(cam_tf_right_);
    search_->setDeterministicNormalEstimation(deterministic_);
    search_->setDevice(device_);
    search_->setKeepsTrainingImages(keeps_training_images_);
  }

  int num_threads_, num_samples_;
  bool filters_boundaries_;
  int plotting_mode_;
  Matrix4d cam_tf_left_, cam_tf_right_;
  VectorXd workspace_;
  double finger_width_, hand_outer_diameter_, hand_depth_, init_bite_, hand_height_, nn_radius_taubin_, nn_radius_hands_;
  bool deterministic_;
  int device_;
  bool keeps_training_images_ = false;
  // localizeHandlesBegin -> localizeHandlesEnd
  PointCloud::Ptr pending_cloud_;
  bool pending_three_calls_ = false;
  int pending_size_left_ = 0, pending_min_inliers_ = 0;
  double pending_min_length_ = 0.0;
  std::vector<int> pending_indices_;
  std::string pending_svm_;
  std::unique_ptr<HandSearch> search_;
  PointCloud::Ptr last_cloud_;
  VectorXi last_cam_;
};

}  // namespace agile_grasp_amd
#endif
