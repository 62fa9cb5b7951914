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

class GraspHypothesis
{
public:
  GraspHypothesis() : cam_source_(-1), grasp_width_(0), full_antipodal_(false), half_antipodal_(false), device_index_(-1),
    n_points_for_learning_(0)
  {
  }

  /** Built from one record of agh_find_hands; device_index = its position in that call's result list. */
  GraspHypothesis(const agh_hypothesis& h, long device_index)
    : axis_(make_vec3(h.axis[0], h.axis[1], h.axis[2])), approach_(make_vec3(h.approach[0], h.approach[1], h.approach[2])),
      binormal_(make_vec3(h.binormal[0], h.binormal[1], h.binormal[2])),
      grasp_bottom_(make_vec3(h.bottom[0], h.bottom[1], h.bottom[2])),
      grasp_surface_(make_vec3(h.surface[0], h.surface[1], h.surface[2])), cam_source_(h.cam_source),
      grasp_width_(h.width), full_antipodal_(h.full_antipodal != 0), half_antipodal_(h.half_antipodal != 0),
      device_index_(device_index), n_points_for_learning_(h.n_in_box)
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
  }

  /** Number of columns the reference's points_for_learning_ would have (grasp_hypothesis.h:220).  The points
   *  themselves stay on the GPU as the 80x100 occupancy image that Learning::classify consumes. */
  int getNumPointsForLearning() const { return n_points_for_learning_; }
  /** Position of this hypothesis in the device-side result list of the HandSearch call that produced it. */
  long getDeviceIndex() const { return device_index_; }

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
  Vector3d axis_, approach_, binormal_, grasp_bottom_, grasp_surface_;
  int cam_source_;
  double grasp_width_;
  bool full_antipodal_, half_antipodal_;
  long device_index_;
  int n_points_for_learning_;
  std::shared_ptr<const std::vector<std::uint32_t> > training_block_;
  std::size_t training_first_ = 0;
};

}  // namespace agile_grasp_amd
#endif
