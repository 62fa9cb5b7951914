// train_test.cpp -- the training run of src/nodes/train.cpp (lines 103-130) through the adapter: hands with antipodal
// labels from one or more clouds, Learning::train / trainBalanced, then Learning::classify with the model just written.
//   train_test <model out> <mode: all|sizes|balanced|linear> <max_positive> <cloud.bin>...
// cloud.bin as in adapter_test.cpp.  Prints "TRAINED <hands> <kept by the new model>".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "agile_grasp_amd/hand_search.h"
#include "agile_grasp_amd/learning.h"

using namespace agile_grasp_amd;

int main(int argc, char** argv)
{
  if (argc < 5)
    return 2;
  const std::string model = argv[1], mode = argv[2];
  const int max_positive = std::atoi(argv[3]);
  std::vector<GraspHypothesis> hand_list, last_hands;
  std::vector<int> hand_list_sizes;
  HandSearch* search = nullptr;
  for (int a = 4; a < argc; a++)
  {
    FILE* f = std::fopen(argv[a], "rb");
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
    if (!search)
    {
      Matrix4d tl, tr;
      for (int r = 0; r < 3; r++)
      {
        tl(r, 3) = cl[r];
        tr(r, 3) = cr[r];
      }
      search = new HandSearch(0.01, 0.09, 0.06, 0.02, 0.01, 1, 2000, tl, false);
      search->setCamTfRight(tr);
      search->setDeterministicNormalEstimation(true);
      search->setKeepsTrainingImages(true);
    }
    std::vector<GraspHypothesis> hands = search->findHands(cloud, src, idx, cloud, true, false);  // train.cpp:115
    hand_list.insert(hand_list.end(), hands.begin(), hands.end());
    hand_list_sizes.push_back((int) hand_list.size());  // train.cpp:117: cumulative
    last_hands = hands;
  }
  Learning learn(*search, 1);
  Matrix3Xd cam_pos;
  std::srand(1);  // the draws of learning.cpp:28,50,119 come from std::rand(): seeded here so that the test can follow them
  if (mode == "all")
    learn.train(hand_list, model, cam_pos, false);  // train.cpp:130
  else if (mode == "sizes")
    learn.train(hand_list, hand_list_sizes, model, cam_pos, max_positive);  // train.cpp:129
  else if (mode == "balanced")
    learn.trainBalanced(hand_list, hand_list_sizes, model, cam_pos, max_positive);  // train.cpp:128
  else if (mode == "linear")
  {
    std::vector<Learning::Instance> ins;
    for (size_t i = 0; i < hand_list.size(); i++)
      if (!hand_list[i].isHalfAntipodal() || hand_list[i].isFullAntipodal())
        for (int c = -1; c <= 1; c++)
          ins.push_back(learn.createInstance(hand_list[i], cam_pos, c));
    if (!learn.convertData(ins, model, false, true))
      return 1;
  }
  else
    return 2;
  std::vector<GraspHypothesis> kept = learn.classify(last_hands, model, cam_pos);
  std::printf("TRAINED %zu %zu\n", hand_list.size(), kept.size());
  delete search;
  return 0;
}
