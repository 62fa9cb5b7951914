// handle_search.h -- Handle (include/agile_grasp/handle.h:49-113) and HandleSearch
// (include/agile_grasp/handle_search.h:45-140) over the MI355X C ABI (agh_find_handles).
//
// Kept: Handle's getters (getAxis/getCenter/getApproach/getBinormal/getHandsCenter/getWidth/getInliers/getHandList) and
// HandleSearch::findHandles(hand_list, min_inliers, min_length) with its prints.  A Handle is built from the record
// the GPU returns (handle.cpp's constructor work happens there); it shares one copy of the hand list with the other
// handles of the same search instead of deep-copying it per handle (handle.cpp:4).
#ifndef AGILE_GRASP_AMD_HANDLE_SEARCH_H
#define AGILE_GRASP_AMD_HANDLE_SEARCH_H

#include <cstdint>
#include <iostream>
#include <memory>
#include <vector>

#include "agh.h"
#include "grasp_hypothesis.h"
#include "hand_search.h"

namespace agile_grasp_amd
{

class Handle
{
public:
  Handle(const agh_handle& h, const std::shared_ptr<const std::vector<GraspHypothesis> >& hand_list,
    const std::vector<int>& inliers)
    : axis_(make_vec3(h.axis[0], h.axis[1], h.axis[2])), center_(make_vec3(h.center[0], h.center[1], h.center[2])),
      approach_(make_vec3(h.approach[0], h.approach[1], h.approach[2])),
      binormal_(make_vec3(h.binormal[0], h.binormal[1], h.binormal[2])),
      hands_center_(make_vec3(h.hands_center[0], h.hands_center[1], h.hands_center[2])), width_(h.width),
      hand_list_(hand_list), inliers_(inliers)
  {
  }
  const Vector3d& getApproach() const { return approach_; }
  const Vector3d& getAxis() const { return axis_; }
  const Vector3d& getCenter() const { return center_; }
  const Vector3d& getHandsCenter() const { return hands_center_; }
  const Vector3d& getBinormal() const { return binormal_; }
  double getWidth() const { return width_; }
  const std::vector<GraspHypothesis>& getHandList() const { return *hand_list_; }
  const std::vector<int>& getInliers() const { return inliers_; }

private:
  Vector3d axis_, center_, approach_, binormal_, hands_center_;
  double width_;
  std::shared_ptr<const std::vector<GraspHypothesis> > hand_list_;
  std::vector<int> inliers_;
};

class HandleSearch
{
public:
  HandleSearch() : search_(nullptr) {}  // handle_search.h:57-68: the context comes from the hands
  /** Additional: work on the context of `search`. */
  explicit HandleSearch(HandSearch& search) : search_(&search) {}

  /** handle_search.cpp:4-85 */
  std::vector<Handle> findHandles(const std::vector<GraspHypothesis>& hand_list, int min_inliers, double min_length)
  {
    std::vector<Handle> handle_list;
    agh_ctx* ctx = finder_.find(search_, hand_list);
    if (!ctx)
      return handle_list;
    std::vector<agh_hypothesis> recs(hand_list.size());
    for (std::size_t i = 0; i < hand_list.size(); i++)
      hand_list[i].toRecord(recs[i]);
    std::vector<agh_handle> handles(hand_list.size() + 1);
    std::vector<std::int32_t> idx(hand_list.size() + 1);
    std::int64_t n = 0;
    const int rc = agh_find_handles(ctx, recs.data(), (std::int64_t) recs.size(), min_inliers, min_length, handles.data(),
      (std::int64_t) handles.size(), idx.data(), (std::int64_t) idx.size(), &n);
    if (rc != AGH_OK)
    {
      std::cout << " Error in agh_find_handles: " << agh_last_error(ctx) << "\n";
      return handle_list;
    }
    std::shared_ptr<const std::vector<GraspHypothesis> > shared(new std::vector<GraspHypothesis>(hand_list));
    for (std::int64_t h = 0; h < n; h++)
    {
      const agh_handle& r = handles[(std::size_t) h];
      std::vector<int> in(idx.begin() + r.first_inlier, idx.begin() + r.first_inlier + r.n_inliers);
      handle_list.push_back(Handle(r, shared, in));
      std::cout << "handle found with " << in.size() << " inliers\n";  // handle_search.cpp:73
    }
    std::cout << "Handle Search\n " << handle_list.size() << " handles found\n";  // :82-84
    return handle_list;
  }

private:
  HandSearch* search_;
  detail::ContextFinder finder_;
};

}  // namespace agile_grasp_amd
#endif
