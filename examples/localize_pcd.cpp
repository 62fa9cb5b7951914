// localize_pcd.cpp -- what src/nodes/find_grasps.cpp + grasp_localizer.cpp do per cloud, without ROS: read one or two
// PCD files, localise hands, keep the ones the SVM calls antipodal, search handles, print the Grasp message fields.
//
//   g++ -std=c++11 -O2 -Iinclude examples/localize_pcd.cpp -o localize_pcd -Lagile_grasp_amd/lib -lagile_grasp_hip
//       -Wl,-rpath,$PWD/agile_grasp_amd/lib -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
//   ./localize_pcd <svm file> <left.pcd> [right.pcd] [num_samples] [min_inliers]
//
// Parameters are the node's defaults (find_grasps.cpp:10-21): finger width 0.01, outer diameter 0.09, hand depth 0.06,
// hand height 0.02, init bite 0.01, 2000 samples, workspace [0.65 0.9 -0.1 0.1 -0.2 1.0], min_inliers 3, camera poses
// of find_grasps.cpp:35-45.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "agile_grasp_amd/localization.h"

using namespace agile_grasp_amd;

int main(int argc, char** argv)
{
  if (argc < 3)
  {
    std::printf("usage: %s <svm file> <left.pcd> [right.pcd] [num_samples] [min_inliers]\n", argv[0]);
    return 2;
  }
  const std::string svm = argv[1], left = argv[2], right = argc > 3 ? argv[3] : "";
  const int num_samples = argc > 4 ? std::atoi(argv[4]) : 2000;
  const int min_inliers = argc > 5 ? std::atoi(argv[5]) : 3;
  // camera poses: base_tf * sqrt_tf^-1 and base_tf * sqrt_tf; only the translations enter the search
  Matrix4d cam_left, cam_right;
  const double tl[3] = { 0.2535951756826822, 0.2724534249381174, 0.19915903314998992 };
  const double tr[3] = { 0.2671843128, -0.3013, 0.2105167716 };
  for (int r = 0; r < 3; r++)
  {
    cam_left(r, 3) = tl[r];
    cam_right(r, 3) = tr[r];
  }
  Localization loc(4, false, 0);  // num_threads (unused on the GPU), filters_boundaries, plotting mode
  loc.setCameraTransforms(cam_left, cam_right);
  VectorXd ws(6);
  const double w[6] = { 0.65, 0.9, -0.1, 0.1, -0.2, 1.0 };
  for (int i = 0; i < 6; i++)
    ws(i) = w[i];
  loc.setWorkspace(ws);
  loc.setNumSamples(num_samples);
  loc.setFingerWidth(0.01);
  loc.setHandOuterDiameter(0.09);
  loc.setHandDepth(0.06);
  loc.setInitBite(0.01);
  loc.setHandHeight(0.02);

  std::vector<GraspHypothesis> hands = loc.localizeHands(left, right, false, false);      // grasp_localizer.cpp:95
  std::vector<GraspHypothesis> antipodal = loc.predictAntipodalHands(hands, svm);          // :102
  std::vector<Handle> handles = loc.findHandles(antipodal, min_inliers, 0.005);            // :103
  const Grasps msg = createGraspsMsg(handles);                                             // :106
  std::printf("%zu hands, %zu antipodal, %zu handles\n", hands.size(), antipodal.size(), handles.size());
  for (std::size_t i = 0; i < msg.grasps.size(); i++)
  {
    const Grasp& g = msg.grasps[i];
    std::printf("grasp %zu: center %.4f %.4f %.4f  axis %.4f %.4f %.4f  approach %.4f %.4f %.4f  width %.4f\n", i,
      g.center(0), g.center(1), g.center(2), g.axis(0), g.axis(1), g.axis(2), g.approach(0), g.approach(1), g.approach(2),
      (double) g.width);
  }
  return 0;
}
