// learning.h -- the prediction side of Learning (include/agile_grasp/learning.h:122-123, learning.cpp:165-247):
// classify(hands_list, svm_filename, cam_pos) keeps the hypotheses the linear SVM labels antipodal.
//
// The HOG descriptor and the SVM score are computed on the GPU from the occupancy images the hand search left there,
// so `hands_list` must come from the most recent HandSearch::findHands of the HandSearch passed to the constructor
// (they are matched by GraspHypothesis::getDeviceIndex()).  cam_pos is the pair of camera origins the search
// already holds (localization.cpp:147-150); it is accepted for signature compatibility.
#ifndef AGILE_GRASP_AMD_LEARNING_H
#define AGILE_GRASP_AMD_LEARNING_H

#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../agh.h"
#include "grasp_hypothesis.h"
#include "hand_search.h"

namespace agile_grasp_amd
{

class Learning
{
public:
  explicit Learning(HandSearch& search, int num_threads = 1) : search_(search), num_threads_(num_threads)
  {
    (void) num_threads_;
  }

  std::vector<GraspHypothesis> classify(const std::vector<GraspHypothesis>& hands_list, const std::string& svm_filename,
    const Matrix3Xd& cam_pos, bool is_plotting = false)
  {
    (void) cam_pos;
    (void) is_plotting;
    std::cout << "Predicting ...\n";
    std::vector<GraspHypothesis> antipodal_hands;
    std::ifstream f(svm_filename.c_str());
    if (!f.good())
    {
      std::cout << " Error: File " << svm_filename << " does not exist!\n";  // learning.cpp:172-178
      return antipodal_hands;
    }
    agh_ctx* ctx = search_.context();
    if (!ctx)
    {
      std::cout << " Error: no hand search has run on this device context\n";
      return antipodal_hands;
    }
    if (agh_load_svm_file(ctx, svm_filename.c_str()) != AGH_OK)
    {
      std::cout << " Exception: " << agh_last_error(ctx) << "\n";  // learning.cpp:187-191
      return antipodal_hands;
    }
    std::vector<unsigned char> keep(hands_list.size() + 1, 0);
    std::int64_t n_kept = 0;
    if (agh_classify(ctx, keep.data(), (std::int64_t) keep.size(), &n_kept) != AGH_OK)
    {
      std::cout << " Error: " << agh_last_error(ctx) << "\n";
      return antipodal_hands;
    }
    for (std::size_t i = 0; i < hands_list.size(); i++)  // input order preserved (learning.cpp:236-243)
    {
      const long k = hands_list[i].getDeviceIndex();
      if (k >= 0 && (std::size_t) k < keep.size() && keep[(std::size_t) k])
      {
        antipodal_hands.push_back(hands_list[i]);
        antipodal_hands.back().setFullAntipodal(true);
      }
    }
    std::cout << " " << antipodal_hands.size() << " antipodal grasps found.\n";
    return antipodal_hands;
  }

private:
  HandSearch& search_;
  int num_threads_;
};

}  // namespace agile_grasp_amd
#endif
