// Compile-only translation unit for the branch of include/agile_grasp_amd/types.h that a ROS / PCL / Eigen build takes
// (AGILE_GRASP_AMD_HAVE_PCL_EIGEN), against the API-shape declarations under tests/cpp/stubs (see the README there).
// The functions below restate, with the adapter's classes, the call patterns of the reference's callers:
//   localization.cpp:111-113 (HandSearch on the stack, findHands), 142-151 (predictAntipodalHands -> Learning::classify with a
//   3 x 2 camera matrix), 390-394 (findHandles); grasp_localizer.cpp:95-103 (the online chain), 137-146 (message fields from the
//   getters); learning.cpp:375-400 (createInstance: column access into points_for_learning_); hands_test.cpp-style mains.
// It is never linked or run: `g++ -fsyntax-only`.
#include <iostream>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include <agile_grasp_amd/grasp_hypothesis.h>
#include <agile_grasp_amd/hand_search.h>
#include <agile_grasp_amd/handle_search.h>
#include <agile_grasp_amd/learning.h>
#include <agile_grasp_amd/localization.h>

#ifndef AGILE_GRASP_AMD_HAVE_PCL_EIGEN
#error "this translation unit is for the real-type branch"
#endif

using namespace agile_grasp_amd;

// the aliases are the library's types, not the stand-ins of the other branch
static_assert(std::is_same<PointCloud, pcl::PointCloud<pcl::PointXYZRGBA> >::value, "hand_search.h:49");
static_assert(std::is_same<PointCloud::Ptr, boost::shared_ptr<pcl::PointCloud<pcl::PointXYZRGBA> > >::value, "PCL 1.7: a boost pointer");
static_assert(!std::is_same<PointCloud::Ptr, std::shared_ptr<PointCloud> >::value, "nothing may assume std::shared_ptr");
static_assert(std::is_same<Vector3d, Eigen::Vector3d>::value && std::is_same<Matrix3Xd, Eigen::Matrix3Xd>::value, "Eigen types");
static_assert(std::is_same<VectorXi, Eigen::VectorXi>::value && std::is_same<Matrix4d, Eigen::Matrix4d>::value, "Eigen types");

// tf::vectorEigenToMsg(const Eigen::Vector3d&, geometry_msgs::Vector3&) as grasp_localizer.cpp:140-143 calls it
struct MsgVector3
{
  double x, y, z;
};
void vectorEigenToMsg(const Eigen::Vector3d& e, MsgVector3& m);

struct GraspMsg
{
  MsgVector3 center, axis, approach, surface_center;
  double width;
};

// grasp_localizer.cpp:137-146
GraspMsg site_create_grasp_msg(const GraspHypothesis& hand)
{
  GraspMsg msg;
  vectorEigenToMsg(hand.getGraspBottom(), msg.center);
  vectorEigenToMsg(hand.getAxis(), msg.axis);
  vectorEigenToMsg(hand.getApproach(), msg.approach);
  vectorEigenToMsg(hand.getGraspSurface(), msg.surface_center);
  msg.width = hand.getGraspWidth();
  return msg;
}

// grasp_localizer.cpp:160-180 (a message per handle)
GraspMsg site_create_handle_msg(const Handle& handle)
{
  GraspMsg msg;
  vectorEigenToMsg(handle.getCenter(), msg.center);
  vectorEigenToMsg(handle.getAxis(), msg.axis);
  vectorEigenToMsg(handle.getApproach(), msg.approach);
  vectorEigenToMsg(handle.getHandsCenter(), msg.surface_center);
  msg.width = handle.getWidth();
  return msg;
}

// localization.cpp:111-113: the searcher lives on the caller's stack, is built from nine scalars and a 4 x 4 pose, and is
// handed a PCL pointer, an Eigen integer vector and an index list
std::vector<GraspHypothesis> site_localization_111(const PointCloud::Ptr& cloud, const Eigen::VectorXi& pts_cam_source,
  const std::vector<int>& indices, const Eigen::Matrix4d& cam_tf_left, const Eigen::Matrix4d& cam_tf_right)
{
  double finger_width = 0.01, hand_outer_diameter = 0.09, hand_depth = 0.06, hand_height = 0.02, init_bite = 0.01;
  int num_threads = 4, num_samples = 2000;
  bool plots_hands = false, calculates_antipodal = false, uses_clustering = false;
  PointCloud::Ptr cloud_plot(new PointCloud);
  cloud_plot = cloud;
  std::vector<GraspHypothesis> hand_list;
  HandSearch hand_search(finger_width, hand_outer_diameter, hand_depth, hand_height, init_bite, num_threads, num_samples,
    cam_tf_left, plots_hands);
  hand_search.setCamTfRight(cam_tf_right);  // (the one call the adapter adds: INTEGRATION.md section 2)
  hand_list = hand_search.findHands(cloud, pts_cam_source, indices, cloud_plot, calculates_antipodal, uses_clustering);
  return hand_list;
}

// localization.cpp:142-151: a fixed 3 x 2 matrix of camera origins goes where a const Matrix3Xd& is expected
std::vector<GraspHypothesis> site_localization_142(const std::vector<GraspHypothesis>& hand_list, const std::string& svm_filename,
  const Eigen::Matrix4d& cam_tf_left, const Eigen::Matrix4d& cam_tf_right)
{
  std::vector<GraspHypothesis> antipodal_hands;
  Learning learn(4);
  Eigen::Matrix<double, 3, 2> cams_mat;
  cams_mat.col(0) = cam_tf_left.block<3, 1>(0, 3);
  cams_mat.col(1) = cam_tf_right.block<3, 1>(0, 3);
  antipodal_hands = learn.classify(hand_list, svm_filename, cams_mat);
  std::cout << antipodal_hands.size() << " antipodal hand configurations found\n";
  return antipodal_hands;
}

// grasp_localizer.cpp:95-103 through the facade
std::vector<Handle> site_grasp_localizer_95(Localization& localization, const PointCloud::Ptr& cloud_left,
  const PointCloud::Ptr& cloud_right, const std::string& svm_file_name, int min_inliers)
{
  std::vector<int> indices(0);
  PointCloud::Ptr cloud(new PointCloud());
  *cloud = *cloud_left + *cloud_right;
  std::vector<GraspHypothesis> hands = localization.localizeHands(cloud, (int) cloud_left->size(), indices, false, false);
  std::vector<GraspHypothesis> antipodal_hands = localization.predictAntipodalHands(hands, svm_file_name);
  std::vector<Handle> handles = localization.findHandles(antipodal_hands, min_inliers, 0.005);
  return handles;
}

// find_grasps.cpp:60-75: the facade's setters take Eigen types
void site_find_grasps_setup(Localization& loc, const Eigen::Matrix4d& base_tf, const Eigen::Matrix4d& sqrt_tf)
{
  Eigen::VectorXd workspace(6);
  workspace(0) = -10.0;
  workspace(1) = 10.0;
  loc.setCameraTransforms(base_tf, sqrt_tf);
  loc.setWorkspace(workspace);
  loc.setNumSamples(2000);
  loc.setFingerWidth(0.01);
  loc.setHandOuterDiameter(0.09);
  loc.setHandDepth(0.06);
  loc.setInitBite(0.01);
  loc.setHandHeight(0.02);
  const Eigen::Matrix4d& left = loc.getCameraTransform(true);
  (void) left;
}

// learning.cpp:375-400: createInstance reads the hypothesis through Eigen expressions
struct Instance
{
  Eigen::Matrix3Xd pts;
  Eigen::Vector3d binormal;
  Eigen::Vector3d source_to_center;
  bool label;
};
Instance site_learning_375(const GraspHypothesis& h, const Eigen::Matrix3Xd& cam_pos, int cam)
{
  Instance ins;
  ins.binormal = h.getBinormal();
  ins.label = h.isFullAntipodal();
  const Eigen::Vector3d& source = cam_pos.col(h.getCamSource());
  ins.source_to_center = h.getGraspSurface() - source;
  if (cam == -1)
    ins.pts = h.getPointsForLearning();
  else
  {
    const std::vector<int>& indices_cam = (cam == 0) ? h.getIndicesPointsForLearningCam1() : h.getIndicesPointsForLearningCam2();
    ins.pts.resize(3, indices_cam.size());
    for (int i = 0; i < (int) indices_cam.size(); i++)
      ins.pts.col(i) = h.getPointsForLearning().col(indices_cam[i]);
  }
  return ins;
}

// handle_search.cpp:13-28 reads these getters of every hand
double site_handle_search_13(const std::vector<GraspHypothesis>& hand_list)
{
  double acc = 0.0;
  for (std::size_t i = 0; i < hand_list.size(); i++)
  {
    const Eigen::Vector3d& axis = hand_list[i].getAxis();
    const Eigen::Vector3d& bottom = hand_list[i].getGraspBottom();
    const Eigen::Vector3d& approach = hand_list[i].getApproach();
    acc += axis(0) + bottom(1) + approach(2) + hand_list[i].getGraspWidth();
  }
  return acc;
}

// a point of a PCL cloud is filled the way pcl::fromROSMsg / the test mains do
void site_fill_cloud(PointCloud& cloud)
{
  pcl::PointXYZRGBA p;
  p.x = 0.1f;
  p.y = 0.2f;
  p.z = 0.3f;
  p.rgba = 0u;
  cloud.points.push_back(p);
  cloud.width = (std::uint32_t) cloud.points.size();
  cloud.height = 1;
  cloud.is_dense = true;
}
