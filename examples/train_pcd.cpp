// train_pcd.cpp -- DEMO of the kept API, not part of the hot-path scope (the reference's CLI nodes are out
// of scope; this file only shows that a caller written against Localization / Learning compiles and runs unchanged).
// What src/nodes/train.cpp does, without ROS/boost: collect hands with antipodal labels from a set of two-view PCD captures,
// train the SVM on their grasp images, write the OpenCV model file.  With the reference's default `uses_clustering = true`
// the search returns nothing (the RANSAC table-plane removal is not built: INTEGRATION.md); this demo passes false.
//
//   g++ -std=c++11 -O2 -Iinclude examples/train_pcd.cpp -o train_pcd -Lagile_grasp_amd/lib -lagile_grasp_hip
//       -Wl,-rpath,$PWD/agile_grasp_amd/lib -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
//   ./train_pcd <num_files> <pcd_dir/> <svm file out> [plots_hands] [num_samples] [num_threads]
//
// As in the node: num_files == 0 reads the capture names from <pcd_dir>files.txt, otherwise the captures are named
// 0, 1, ...; capture X consists of <pcd_dir>X + "l_reg.pcd" and "r_reg.pcd"; <pcd_dir>workspace.txt (six numbers per line)
// overrides the standard workspace [0.65 0.9 -0.1 0.1 -0.2 1.0]; hand geometry of train.cpp:95-103 (init bite 0.015);
// 1000 samples; max_positives = 20; Learning::train(hand_list, sizes, file, cam_pos, max_positives), which trains the
// quadratic-kernel model (convertData's default).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "agile_grasp_amd/learning.h"
#include "agile_grasp_amd/localization.h"

using namespace agile_grasp_amd;

int main(int argc, char** argv)
{
  if (argc <= 3)
  {
    std::cout << "No PCD filenames given!\n";  // train.cpp:135
    std::cout << "usage: " << argv[0] << " <num_files> <pcd_dir/> <svm file out> [plots_hands] [num_samples] [num_threads]\n";
    return -1;
  }
  const int num_files = std::atoi(argv[1]);
  const std::string pcd_dir = argv[2], svm_file_name = argv[3];
  std::vector<std::string> files;
  if (num_files == 0)
  {
    std::ifstream file((pcd_dir + "files.txt").c_str());
    std::string str;
    while (std::getline(file, str))
      if (!str.empty())
      {
        files.push_back(pcd_dir + str);
        std::cout << files.back() << "\n";
      }
  }
  else
    for (int i = 0; i < num_files; i++)
    {
      std::ostringstream name;
      name << pcd_dir << i;
      files.push_back(name.str());
    }
  std::vector<std::vector<double> > workspace_mat(files.size(), std::vector<double>{ 0.65, 0.9, -0.1, 0.1, -0.2, 1.0 });
  std::ifstream file_ws((pcd_dir + "workspace.txt").c_str());
  if (!file_ws.good())
    std::cout << "No workspace.txt file found in pcd directory\n Using standard workspace limits\n";
  else
  {
    std::string str;
    for (std::size_t t = 0; t < files.size() && std::getline(file_ws, str); t++)
    {
      std::istringstream line(str);
      for (int i = 0; i < 6; i++)
        line >> workspace_mat[t][(std::size_t) i];
    }
  }
  const int num_samples = argc > 5 ? std::atoi(argv[5]) : 1000;
  const int num_threads = argc > 6 ? std::atoi(argv[6]) : 4;
  // camera poses of the two-camera Baxter setup (train.cpp:80-93): base_tf * sqrt_tf^-1 and base_tf * sqrt_tf; only
  // the translations enter the search
  Matrix4d cam_left, cam_right;
  const double tl[3] = { 0.2535951756826822, 0.2724534249381174, 0.19915903314998992 };
  const double tr[3] = { 0.2671843128, -0.3013, 0.2105167716 };
  for (int r = 0; r < 3; r++)
  {
    cam_left(r, 3) = tl[r];
    cam_right(r, 3) = tr[r];
  }
  Localization loc(num_threads, false, 0);
  loc.setCameraTransforms(cam_left, cam_right);
  loc.setNumSamples(num_samples);
  loc.setNeighborhoodRadiusTaubin(0.03);
  loc.setNeighborhoodRadiusHands(0.08);
  loc.setFingerWidth(0.01);
  loc.setHandOuterDiameter(0.09);
  loc.setHandDepth(0.06);
  loc.setInitBite(0.015);
  loc.setHandHeight(0.02);
  loc.setKeepsTrainingImages(true);  // the hypotheses carry their three instance images

  std::cout << "Acquiring training data ...\n";
  std::vector<GraspHypothesis> hand_list;
  std::vector<int> hand_list_sizes(files.size());
  for (std::size_t i = 0; i < files.size(); i++)
  {
    std::cout << " Creating training data from file " << files[i] << " ...\n";
    VectorXd ws(6);
    for (int k = 0; k < 6; k++)
      ws((std::size_t) k) = workspace_mat[i][(std::size_t) k];
    loc.setWorkspace(ws);
    // src/nodes/train.cpp:115 passes uses_clustering = true (RANSAC table-plane removal, pcl::SACSegmentation).  That step
    // is not part of this build -- asking for it returns an empty list with an error -- so this example expects clouds
    // whose table plane has been removed already (or workspaces that exclude it) and passes false.
    std::vector<GraspHypothesis> hands = loc.localizeHands(files[i] + "l_reg.pcd", files[i] + "r_reg.pcd", true, false);
    hand_list.insert(hand_list.end(), hands.begin(), hands.end());
    hand_list_sizes[i] = (int) hand_list.size();
    std::cout << i << ") # hands: " << hands.size() << std::endl;
  }
  if (hand_list.empty())
  {
    std::cout << "No hands found: nothing to train on\n";
    return 1;
  }
  std::cout << "Training the SVM ...\n";
  Learning learn(4);   // train.cpp:122
  Matrix3Xd cam_pos;  // the instance images were rasterised with the two camera origins the search holds
  const int max_positives = 20;
  learn.train(hand_list, hand_list_sizes, svm_file_name, cam_pos, max_positives);
  return 0;
}
