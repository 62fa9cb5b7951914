// adapter_test.cpp -- exercises the header-only C++ adapter (include/agile_grasp_amd/) the way the reference's
// src/tests/hands_test.cpp / learning_test.cpp drive HandSearch and Learning, on a cloud dumped by the Python test.
//   adapter_test <cloud.bin> <svm file> <deterministic 0|1>
// cloud.bin: int64 n, int64 n_samples, double cam_left[3], double cam_right[3], then n*(3 float) xyz, n*int32 cam,
// n_samples*int32 indices.  Prints one line per hypothesis: sample-less record fields in full precision.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "agile_grasp_amd/hand_search.h"
#include "agile_grasp_amd/learning.h"

using namespace agile_grasp_amd;

int main(int argc, char** argv)
{
  if (argc < 4)
    return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f)
    return 2;
  long long n = 0, ns = 0;
  double cl[3], cr[3];
  if (std::fread(&n, 8, 1, f) != 1 || std::fread(&ns, 8, 1, f) != 1 || std::fread(cl, 8, 3, f) != 3 || std::fread(cr, 8, 3, f) != 3)
    return 2;
  std::vector<float> xyz(3 * (size_t) n);
  std::vector<int> cam((size_t) n), idx((size_t) ns);
  if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || std::fread(cam.data(), 4, cam.size(), f) != cam.size() ||
      std::fread(idx.data(), 4, idx.size(), f) != idx.size())
    return 2;
  std::fclose(f);
  PointCloud::Ptr cloud(new PointCloud);
  cloud->points.resize((size_t) n);
  VectorXi src((size_t) n);
  for (long long i = 0; i < n; i++)
  {
    cloud->points[(size_t) i].x = xyz[3 * i];
    cloud->points[(size_t) i].y = xyz[3 * i + 1];
    cloud->points[(size_t) i].z = xyz[3 * i + 2];
    src((size_t) i) = cam[(size_t) i];
  }
  Matrix4d tl, tr;
  for (int r = 0; r < 3; r++)
  {
    tl(r, 3) = cl[r];
    tr(r, 3) = cr[r];
  }
  // the reference node's hand geometry (find_grasps.cpp:13-17), 1 thread, 2000 samples
  HandSearch hand_search(0.01, 0.09, 0.06, 0.02, 0.01, 1, 2000, tl, false);
  hand_search.setCamTfRight(tr);
  hand_search.setDeterministicNormalEstimation(std::atoi(argv[3]) != 0);
  std::vector<GraspHypothesis> hands = hand_search.findHands(cloud, src, idx, cloud, false, false);
  Learning learn(hand_search, 1);
  Matrix3Xd cam_pos;
  std::vector<GraspHypothesis> antipodal = learn.classify(hands, argv[2], cam_pos);
  std::printf("RESULT %zu %zu\n", hands.size(), antipodal.size());
  for (size_t i = 0; i < hands.size(); i++)
  {
    const GraspHypothesis& h = hands[i];
    std::printf("H %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %d\n", h.getGraspSurface()(0), h.getGraspSurface()(1),
      h.getGraspSurface()(2), h.getApproach()(0), h.getAxis()(1), h.getBinormal()(2), h.getGraspWidth(), h.getCamSource(),
      h.getNumPointsForLearning());
  }
  for (size_t i = 0; i < antipodal.size(); i++)
    std::printf("A %ld\n", antipodal[i].getDeviceIndex());
  // the variable part of the first and the last hypothesis, fetched on demand (grasp_hypothesis.h:149-170)
  for (size_t pick = 0; pick < 2 && !hands.empty(); pick++)
  {
    const GraspHypothesis& h = pick == 0 ? hands.front() : hands.back();
    Matrix3Xd pts;
    std::vector<int> cam1, cam2;
    if (!hand_search.getPointsForLearning(h, pts, cam1, cam2))
      return 3;
    // the reference's accessors on the hypothesis itself (learning.cpp:387-395) give the same
    if (h.getPointsForLearning().cols() != pts.cols() || h.getIndicesPointsForLearningCam1() != cam1 ||
        h.getIndicesPointsForLearningCam2() != cam2)
      return 4;
    for (size_t k = 0; k < pts.cols(); k++)
      for (int r = 0; r < 3; r++)
        if (h.getPointsForLearning()(r, k) != pts(r, k))
          return 4;
    double sum[3] = { 0, 0, 0 };
    for (size_t k = 0; k < pts.cols(); k++)
      for (int r = 0; r < 3; r++)
        sum[r] += pts(r, k);
    std::printf("P %ld %zu %zu %zu %.17g %.17g %.17g\n", h.getDeviceIndex(), (size_t) pts.cols(), cam1.size(), cam2.size(),
      sum[0], sum[1], sum[2]);
  }
  return 0;
}
