// sharded_test.cpp -- the C++ host side of the multi-GPU path: G HandSearch objects (one host thread each, standing for one
// process per GPU) join a communicator, every one calls findHands with the same cloud and indices, searches its slice of the
// samples and receives the complete list; Learning::classify on that list is the matching collective.
//   sharded_test <cloud.bin> <svm file> <G>        (cloud.bin as adapter_test.cpp reads it)
// The communicator here is the in-process one (device copies); with one process per GPU the only difference is
// HandSearch::joinCommunicator(rank, n_ranks, id) with the bytes of agh_comm_unique_id().
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
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
  const int G = std::atoi(argv[3]);
  std::vector<std::unique_ptr<HandSearch> > searches;
  std::vector<HandSearch*> raw;
  for (int g = 0; g < G; g++)
  {
    searches.emplace_back(new HandSearch(0.01, 0.09, 0.06, 0.02, 0.01, 1, 2000, tl, false));
    searches.back()->setCamTfRight(tr);
    searches.back()->setDeterministicNormalEstimation(true);
    raw.push_back(searches.back().get());
  }
  if (!HandSearch::joinLocalCommunicator(raw))
  {
    std::printf("NO_COMMUNICATOR\n");
    return 0;
  }
  std::vector<std::vector<GraspHypothesis> > hands((size_t) G), kept((size_t) G);
  std::vector<std::thread> th;
  for (int g = 0; g < G; g++)
    th.emplace_back([&, g] {
      hands[(size_t) g] = raw[(size_t) g]->findHands(cloud, src, idx, cloud, false, false);  // collective
      Learning learn(1);
      Matrix3Xd cam_pos;
      kept[(size_t) g] = learn.classify(hands[(size_t) g], argv[2], cam_pos);                  // collective
    });
  for (size_t g = 0; g < th.size(); g++)
    th[g].join();
  for (int g = 0; g < G; g++)
  {
    std::printf("RANK %d %zu %zu\n", g, hands[(size_t) g].size(), kept[(size_t) g].size());
    for (size_t i = 0; i < hands[(size_t) g].size(); i++)
      std::printf("H%d %.17g %.17g %.17g %.17g\n", g, hands[(size_t) g][i].getGraspSurface()(0), hands[(size_t) g][i].getGraspBottom()(1),
        hands[(size_t) g][i].getApproach()(2), hands[(size_t) g][i].getGraspWidth());
    for (size_t i = 0; i < kept[(size_t) g].size(); i++)
      std::printf("K%d %ld\n", g, kept[(size_t) g][i].getDeviceIndex());
  }
  // GraspHypothesis::getPointsForLearning after a sharded search: the points live on the rank that searched the sample;
  // exactly one rank serves each hypothesis, the others say so (the merged index used to be taken for a local one)
  for (size_t i = 0; i < hands[0].size(); i += 5)
  {
    int owners = 0, owner = -1;
    for (int g = 0; g < G; g++)
      if (hands[(size_t) g][i].getLocalIndex() >= 0)
      {
        owners++;
        owner = g;
      }
    double sum = 0.0;
    long cols = -1, foreign_cols = 0;
    if (owners == 1)
    {
      const Matrix3Xd& pts = hands[(size_t) owner][i].getPointsForLearning();
      cols = (long) pts.cols();
      for (long c = 0; c < cols; c++)
        for (int r = 0; r < 3; r++)
          sum += pts(r, (size_t) c);
      foreign_cols = (long) hands[(size_t) ((owner + 1) % G)][i].getPointsForLearning().cols();  // prints why, returns empty
    }
    std::printf("P %zu %d %ld %d %ld %.17g\n", i, owners, cols, hands[0][i].getNumPointsForLearning(), G > 1 ? foreign_cols : 0L, sum);
  }
  return 0;
}
