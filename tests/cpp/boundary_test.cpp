// boundary_test.cpp -- the drop-in boundary as the reference's ROS node uses it (src/agile_grasp/grasp_localizer.cpp:21,
// 95-103; src/nodes/test.cpp:72, 95-97): Localization with filters_boundaries = true, predictAntipodalHands on the
// FILTERED list, lists that outlive their search, the lazy GraspHypothesis::getPointsForLearning accessors
// (grasp_hypothesis.h:149-170), the empty-`indices` sampling path (hand_search.cpp:31-44).
//   boundary_test sampler <n_points> <num_samples> <seed>                 (needs no GPU)
//   boundary_test filtered|stale|drawn <raw.bin> <svm file>
// raw.bin as localization_test.cpp reads it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "agile_grasp_amd/localization.h"

using namespace agile_grasp_amd;

static void print_hands(const char* tag, const std::vector<GraspHypothesis>& hands)
{
  for (size_t i = 0; i < hands.size(); i++)
    std::printf("%s %.17g %.17g %.17g %.17g %d\n", tag, hands[i].getGraspSurface()(0), hands[i].getGraspSurface()(1),
      hands[i].getGraspSurface()(2), hands[i].getGraspWidth(), hands[i].isFullAntipodal() ? 1 : 0);
}

static void print_points(const char* tag, const GraspHypothesis& h)
{
  const Matrix3Xd& pts = h.getPointsForLearning();  // learning.cpp:387
  const std::vector<int>& c1 = h.getIndicesPointsForLearningCam1();
  const std::vector<int>& c2 = h.getIndicesPointsForLearningCam2();
  double sum[3] = { 0, 0, 0 };
  for (size_t k = 0; k < (size_t) pts.cols(); k++)
    for (int r = 0; r < 3; r++)
      sum[r] += pts(r, k);
  std::printf("%s %ld %zu %zu %zu %.17g %.17g %.17g\n", tag, h.getDeviceIndex(), (size_t) pts.cols(), c1.size(), c2.size(),
    sum[0], sum[1], sum[2]);
}

int main(int argc, char** argv)
{
  if (argc >= 5 && std::strcmp(argv[1], "sampler") == 0)
  {
    const std::vector<std::int32_t> idx = HandSearch::randomSample(std::atoll(argv[2]), std::atoi(argv[3]),
      (unsigned) std::strtoul(argv[4], nullptr, 10));
    std::printf("SAMPLES %zu\n", idx.size());
    for (size_t i = 0; i < idx.size(); i++)
      std::printf("S %d\n", (int) idx[i]);
    return 0;
  }
  if (argc < 4)
    return 2;
  FILE* f = std::fopen(argv[2], "rb");
  if (!f)
    return 2;
  long long n = 0, size_left = 0, n_idx = 0;
  double ws[6], cl[3], cr[3];
  if (std::fread(&n, 8, 1, f) != 1 || std::fread(&size_left, 8, 1, f) != 1 || std::fread(&n_idx, 8, 1, f) != 1 ||
      std::fread(ws, 8, 6, f) != 6 || std::fread(cl, 8, 3, f) != 3 || std::fread(cr, 8, 3, f) != 3)
    return 2;
  std::vector<float> xyz(3 * (size_t) n);
  std::vector<int> idx((size_t) n_idx);
  if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || std::fread(idx.data(), 4, idx.size(), f) != idx.size())
    return 2;
  std::fclose(f);
  PointCloud::Ptr cloud(new PointCloud);
  cloud->points.resize((size_t) n);
  for (long long i = 0; i < n; i++)
  {
    cloud->points[(size_t) i].x = xyz[3 * i];
    cloud->points[(size_t) i].y = xyz[3 * i + 1];
    cloud->points[(size_t) i].z = xyz[3 * i + 2];
  }
  Matrix4d tl, tr;
  for (int r = 0; r < 3; r++)
  {
    tl(r, 3) = cl[r];
    tr(r, 3) = cr[r];
  }
  VectorXd w(6);
  for (int i = 0; i < 6; i++)
    w(i) = ws[i];

  if (std::strcmp(argv[1], "filtered") == 0)
  {
    // exactly the node's construction: Localization(num_threads, filters_boundaries = true, plotting mode)
    Localization loc(4, true, 0);
    loc.setCameraTransforms(tl, tr);
    loc.setWorkspace(w);
    loc.setDeterministicNormalEstimation(true);
    const size_t n_before = cloud->size();
    std::vector<GraspHypothesis> hands = loc.localizeHands(cloud, (int) size_left, idx, false, false);
    std::printf("CLOUD %zu %zu\n", n_before, cloud->size());  // the caller's cloud loses its NaN points (localization.cpp:27)
    std::vector<GraspHypothesis> kept = loc.predictAntipodalHands(hands, argv[3]);
    std::printf("RESULT %zu %zu\n", hands.size(), kept.size());
    print_hands("H", hands);
    print_hands("K", kept);
    std::vector<Handle> handles = loc.findHandles(kept, 3, 0.005);
    std::printf("HANDLES %zu\n", handles.size());
    // uses_clustering is not available: an error and an empty list, never a silent search of the unsegmented cloud
    std::vector<GraspHypothesis> none = loc.localizeHands(cloud, (int) cloud->size(), idx, false, true);
    std::printf("CLUSTERING %zu\n", none.size());
    return 0;
  }
  if (std::strcmp(argv[1], "stale") == 0)
  {
    std::unique_ptr<Localization> locp(new Localization(1, false, 0));
    Localization& loc = *locp;
    loc.setCameraTransforms(tl, tr);
    loc.setWorkspace(w);
    loc.setDeterministicNormalEstimation(true);
    const std::vector<int> first(idx.begin(), idx.begin() + (long) (idx.size() / 2));
    const std::vector<int> second(idx.begin() + (long) (idx.size() / 2), idx.end());
    std::vector<GraspHypothesis> hands1 = loc.localizeHands(cloud, (int) size_left, first, false, false);
    if (hands1.size() < 2)
      return 3;
    print_points("P1", hands1[0]);  // fetched while the search still holds this result: cached in the hypothesis
    std::vector<GraspHypothesis> hands2 = loc.localizeHands(cloud, (int) size_left, second, false, false);
    // the first list is now stale on the device; it still classifies (its images travel with it), in any mixture
    std::vector<GraspHypothesis> k1 = loc.predictAntipodalHands(hands1, argv[3]);
    std::vector<GraspHypothesis> k2 = loc.predictAntipodalHands(hands2, argv[3]);
    std::vector<GraspHypothesis> both(hands2);
    both.insert(both.end(), hands1.begin(), hands1.end());
    std::vector<GraspHypothesis> k12 = loc.predictAntipodalHands(both, argv[3]);
    std::printf("RESULT %zu %zu %zu %zu %zu\n", hands1.size(), hands2.size(), k1.size(), k2.size(), k12.size());
    print_hands("H1", hands1);
    print_hands("H2", hands2);
    print_hands("K1", k1);
    print_hands("K2", k2);
    print_hands("K12", k12);
    print_points("P1AGAIN", hands1[0]);  // the cached copy
    print_points("P1STALE", hands1[1]);  // never fetched in time: an error message and empty containers
    print_points("P2", hands2[0]);
    // a Learning of its own, after every search is gone, on copies of the hypotheses (learning.h:73-76)
    std::vector<GraspHypothesis> copies(hands1);
    locp.reset();
    Learning learn(2);
    Matrix3Xd cam_pos;
    std::vector<GraspHypothesis> k1b = learn.classify(copies, argv[3], cam_pos);
    std::printf("AGAIN %zu\n", k1b.size());
    return 0;
  }
  if (std::strcmp(argv[1], "drawn") == 0)
  {
    // hand_search.cpp:31-44: no indices given => num_samples random ones
    HandSearch hs(0.01, 0.09, 0.06, 0.02, 0.01, 1, 40, tl, false);
    hs.setCamTfRight(tr);
    hs.setDeterministicNormalEstimation(true);
    hs.setSampleSeed(7);
    VectorXi src((size_t) n);
    for (long long i = 0; i < n; i++)
      src((size_t) i) = i >= size_left ? 1 : 0;
    std::vector<GraspHypothesis> hands = hs.findHands(cloud, src, std::vector<int>(), cloud, false, false);
    std::printf("RESULT %zu\n", hands.size());
    for (size_t i = 0; i < hs.getLastSampleIndices().size(); i++)
      std::printf("S %d\n", (int) hs.getLastSampleIndices()[i]);
    print_hands("H", hands);
    return 0;
  }
  return 2;
}
