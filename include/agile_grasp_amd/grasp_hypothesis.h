// grasp_hypothesis.h -- GraspHypothesis with the reference's accessors (include/agile_grasp/grasp_hypothesis.h:46-231)
// on top of the fixed-size agh_hypothesis record of the C ABI.
#ifndef AGILE_GRASP_AMD_GRASP_HYPOTHESIS_H
#define AGILE_GRASP_AMD_GRASP_HYPOTHESIS_H

#include <cstdint>
#include <iostream>
#include <memory>
#include <vector>

#include "../agh.h"
#include "types.h"

namespace agile_grasp_amd
{

namespace detail
{
/** What a hypothesis knows about the search that produced it: the device context, or nullptr once that HandSearch is
 *  gone or has been re-created.  Shared by the HandSearch and every hypothesis it returned. */
struct SearchLink
{
  agh_ctx* ctx;
  SearchLink() : ctx(nullptr) {}
};
}  // namespace detail

class GraspHypothesis
{
public:
  GraspHypothesis() : cam_source_(-1), grasp_width_(0), full_antipodal_(false), half_antipodal_(false), device_index_(-1),
    local_index_(-1), epoch_(0), n_points_for_learning_(0), points_fetched_(false)
  {
  }

  /** The reference's constructor (grasp_hypothesis.h:67-75): a self-contained hypothesis with its points.  Such a
   *  hypothesis carries no occupancy image, so Learning::classify refuses it (the image is rasterised by the search). */
  GraspHypothesis(const Vector3d& axis, const Vector3d& approach, const Vector3d& binormal, const Vector3d& bottom,
    const Vector3d& surface, double width, const Matrix3Xd& points_for_learning, const std::vector<int>& indices_cam1,
    const std::vector<int>& indices_cam2, int cam_source)
    : axis_(axis), approach_(approach), binormal_(binormal), grasp_bottom_(bottom), grasp_surface_(surface),
      cam_source_(cam_source), grasp_width_(width), full_antipodal_(false), half_antipodal_(false), device_index_(-1),
      local_index_(-1), epoch_(0), n_points_for_learning_((int) points_for_learning.cols()), points_fetched_(true),
      points_for_learning_(points_for_learning), indices_cam1_(indices_cam1), indices_cam2_(indices_cam2)
  {
  }

  /** Built from one record of agh_find_hands; device_index = its position in that call's result list.  A SHARDED search
   *  returns the merged list of all ranks, but a rank's device keeps only the results (points, images) of its own samples:
   *  local_index = the record's position in THIS rank's device-side list, or -1 if another rank searched its sample
   *  (default: the same as device_index, the single-GPU case). */
  GraspHypothesis(const agh_hypothesis& h, long device_index,
    const std::shared_ptr<detail::SearchLink>& link = std::shared_ptr<detail::SearchLink>(), long local_index = -2)
    : axis_(make_vec3(h.axis[0], h.axis[1], h.axis[2])), approach_(make_vec3(h.approach[0], h.approach[1], h.approach[2])),
      binormal_(make_vec3(h.binormal[0], h.binormal[1], h.binormal[2])),
      grasp_bottom_(make_vec3(h.bottom[0], h.bottom[1], h.bottom[2])),
      grasp_surface_(make_vec3(h.surface[0], h.surface[1], h.surface[2])), cam_source_(h.cam_source),
      grasp_width_(h.width), full_antipodal_(h.full_antipodal != 0), half_antipodal_(h.half_antipodal != 0),
      device_index_(device_index), local_index_(local_index == -2 ? device_index : local_index), epoch_(h.epoch),
      n_points_for_learning_(h.n_in_box), points_fetched_(false), link_(link)
  {
  }

  void print()  // grasp_hypothesis.cpp:3-11
  {
    std::cout << "axis: " << axis_(0) << " " << axis_(1) << " " << axis_(2) << std::endl;
    std::cout << "approach: " << approach_(0) << " " << approach_(1) << " " << approach_(2) << std::endl;
    std::cout << "binormal: " << binormal_(0) << " " << binormal_(1) << " " << binormal_(2) << std::endl;
    std::cout << "grasp width: " << grasp_width_ << std::endl;
    std::cout << "grasp surface: " << grasp_surface_(0) << " " << grasp_surface_(1) << " " << grasp_surface_(2) << std::endl;
    std::cout << "grasp bottom: " << grasp_bottom_(0) << " " << grasp_bottom_(1) << " " << grasp_bottom_(2) << std::endl;
  }

  const Vector3d& getApproach() const { return approach_; }
  const Vector3d& getAxis() const { return axis_; }
  const Vector3d& getBinormal() const { return binormal_; }
  bool isFullAntipodal() const { return full_antipodal_; }
  const Vector3d& getGraspBottom() const { return grasp_bottom_; }
  const Vector3d& getGraspSurface() const { return grasp_surface_; }
  double getGraspWidth() const { return grasp_width_; }
  bool isHalfAntipodal() const { return half_antipodal_; }
  int getCamSource() const { return cam_source_; }
  void setFullAntipodal(bool b) { full_antipodal_ = b; }
  void setHalfAntipodal(bool b) { half_antipodal_ = b; }
  void setGraspWidth(double w) { grasp_width_ = w; }

  /** grasp_hypothesis.h:149-170.  The reference stores the 3 x n_b matrix and the two index lists in every hypothesis
   *  (hundreds of MB per cloud); here they are fetched from the GPU the first time one of the three getters
   *  is called (agh_get_learning_points) and cached in the hypothesis.  That works while the search that produced the
   *  hypothesis still holds the same cloud and results; afterwards the getters print why and return empty containers. */
  const Matrix3Xd& getPointsForLearning() const
  {
    fetchPoints();
    return points_for_learning_;
  }
  const std::vector<int>& getIndicesPointsForLearningCam1() const
  {
    fetchPoints();
    return indices_cam1_;
  }
  const std::vector<int>& getIndicesPointsForLearningCam2() const
  {
    fetchPoints();
    return indices_cam2_;
  }

  /** The geometric fields as an ABI record (input of agh_find_handles). */
  void toRecord(agh_hypothesis& h) const
  {
    h = agh_hypothesis();
    for (int r = 0; r < 3; r++)
    {
      h.axis[r] = axis_(r);
      h.approach[r] = approach_(r);
      h.binormal[r] = binormal_(r);
      h.bottom[r] = grasp_bottom_(r);
      h.surface[r] = grasp_surface_(r);
    }
    h.width = grasp_width_;
    h.cam_source = cam_source_;
    h.n_in_box = n_points_for_learning_;
    h.half_antipodal = half_antipodal_ ? 1 : 0;
    h.full_antipodal = full_antipodal_ ? 1 : 0;
    h.valid = 1;
    h.epoch = epoch_;
  }

  /** Number of columns of points_for_learning_ (grasp_hypothesis.h:220) without fetching them. */
  int getNumPointsForLearning() const { return n_points_for_learning_; }
  /** Position of this hypothesis in the device-side result list of the HandSearch call that produced it, and that
   *  call's stamp (agh_hypothesis::epoch). */
  long getDeviceIndex() const { return device_index_; }
  /** Position in the device-side list of the context that searched this hypothesis' sample (what agh_get_learning_points
   *  takes); -1 if that was another rank of a sharded search. */
  long getLocalIndex() const { return local_index_; }
  int getEpoch() const { return epoch_; }
  /** The device context that still holds this hypothesis' search results, or nullptr (search gone, re-created, it has run
   *  another findHands since, or -- sharded search -- the results live on another rank). */
  agh_ctx* getLiveContext() const
  {
    if (!link_ || !link_->ctx || local_index_ < 0)
      return nullptr;
    std::int32_t e = 0;
    std::int64_t n = 0;
    if (agh_get_epoch(link_->ctx, &e, &n) != AGH_OK || e != epoch_ || local_index_ >= n)
      return nullptr;
    return link_->ctx;
  }
  /** The context if it is still at this hypothesis' call (epoch), whatever the position: what a sharded search's list, whose
   *  positions count over all ranks, is matched by. */
  agh_ctx* getContextAtEpoch() const
  {
    if (!link_ || !link_->ctx || device_index_ < 0)
      return nullptr;
    std::int32_t e = 0;
    if (agh_get_epoch(link_->ctx, &e, nullptr) != AGH_OK || e != epoch_)
      return nullptr;
    return link_->ctx;
  }
  /** Any context this hypothesis' search still owns (for work that needs a device but none of the search's state). */
  agh_ctx* getAnyContext() const { return link_ ? link_->ctx : nullptr; }

  /** The 80 x 100 occupancy image Learning::convertToImage would build from the points (learning.cpp:320-365), packed
   *  (250 words, see agh.h): the hand sweep rasterises it and HandSearch attaches it, so that Learning::classify works on
   *  any list of hypotheses at any later time, like the reference's.  `block` is shared by the hypotheses of one call. */
  void setImage(const std::shared_ptr<const std::vector<std::uint32_t> >& block, std::size_t first_word)
  {
    image_block_ = block;
    image_first_ = first_word;
  }
  const std::uint32_t* getImage() const { return image_block_ ? image_block_->data() + image_first_ : nullptr; }

  /** Training side.  The reference keeps points_for_learning_ and their split by camera in every hypothesis
   *  (grasp_hypothesis.h:220-222) so that Learning::train can rasterise three instances later; here the hand sweep
   *  rasterises them and the hypothesis carries the three packed 80x100 images (250 words each; cam = -1, 0, 1) when
   *  HandSearch::setKeepsTrainingImages(true) was set.  `block` is shared by the hypotheses of one findHands call. */
  void setTrainingImages(const std::shared_ptr<const std::vector<std::uint32_t> >& block, std::size_t first_word)
  {
    training_block_ = block;
    training_first_ = first_word;
  }
  bool hasTrainingImages() const { return (bool) training_block_; }
  /** cam = -1: all points of the hand box (createInstance's default); 0 / 1: that camera's points (learning.cpp:385-397). */
  const std::uint32_t* getTrainingImage(int cam) const
  {
    return training_block_ ? training_block_->data() + training_first_ + (std::size_t) (cam + 1) * 250 : nullptr;
  }

private:
  void fetchPoints() const
  {
    if (points_fetched_)
      return;
    points_fetched_ = true;  // one attempt: a stale hypothesis stays empty
    resize_3xn(points_for_learning_, 0);
    indices_cam1_.clear();
    indices_cam2_.clear();
    if (device_index_ < 0)
      return;  // default-constructed
    agh_ctx* ctx = getLiveContext();
    if (!ctx)
    {
      if (local_index_ < 0)
        std::cout << " Error: getPointsForLearning: this hypothesis of a sharded search was found by another rank; its points "
                     "live on that rank's GPU\n";
      else
        std::cout << " Error: getPointsForLearning: the search that produced this hypothesis no longer holds its cloud and "
                     "results (it ran another findHands or was destroyed); fetch the points before the next search\n";
      return;
    }
    const std::size_t n_b = (std::size_t) n_points_for_learning_;
    std::vector<std::int32_t> cam(n_b + 1);
    std::vector<double> dummy(3);
    std::int64_t n = 0;
    double* dst = resize_3xn(points_for_learning_, n_b);
    if (agh_get_learning_points(ctx, local_index_, n_b ? dst : dummy.data(), cam.data(), (std::int64_t) n_b, &n) != AGH_OK)
    {
      std::cout << " Error in agh_get_learning_points: " << agh_last_error(ctx) << "\n";
      resize_3xn(points_for_learning_, 0);
      return;
    }
    for (std::size_t k = 0; k < n_b; k++)  // rotating_hand.cpp:143-151
      if (cam[k] == 0)
        indices_cam1_.push_back((int) k);
      else if (cam[k] == 1)
        indices_cam2_.push_back((int) k);
  }

  Vector3d axis_, approach_, binormal_, grasp_bottom_, grasp_surface_;
  int cam_source_;
  double grasp_width_;
  bool full_antipodal_, half_antipodal_;
  long device_index_, local_index_;
  int epoch_;
  int n_points_for_learning_;
  mutable bool points_fetched_;
  mutable Matrix3Xd points_for_learning_;
  mutable std::vector<int> indices_cam1_, indices_cam2_;
  std::shared_ptr<detail::SearchLink> link_;
  std::shared_ptr<const std::vector<std::uint32_t> > image_block_;
  std::size_t image_first_ = 0;
  std::shared_ptr<const std::vector<std::uint32_t> > training_block_;
  std::size_t training_first_ = 0;
};

}  // namespace agile_grasp_amd
#endif
