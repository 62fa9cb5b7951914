// types.h -- the PCL / Eigen types that appear in the reference's hot-path signatures.
//
// With PCL and Eigen installed (the ROS build of the reference) the real types are used and the adapter classes are
// source-compatible with include/agile_grasp/{hand_search,grasp_hypothesis,learning}.h.  This repository's build image
// has neither, so minimal stand-ins with the members the adapter touches are provided for the tests; they are NOT a
// PCL/Eigen replacement.
#ifndef AGILE_GRASP_AMD_TYPES_H
#define AGILE_GRASP_AMD_TYPES_H

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>)
#define AGILE_GRASP_AMD_HAVE_PCL_EIGEN 1
#endif
#endif

#ifdef AGILE_GRASP_AMD_HAVE_PCL_EIGEN
#include <Eigen/Dense>
#include <pcl/filters/filter.h>
#include <pcl/io/pcd_io.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

namespace agile_grasp_amd
{
typedef pcl::PointCloud<pcl::PointXYZRGBA> PointCloud;  // hand_search.h:49
typedef Eigen::Vector3d Vector3d;
typedef Eigen::Matrix4d Matrix4d;
typedef Eigen::VectorXi VectorXi;
typedef Eigen::Matrix3Xd Matrix3Xd;
typedef Eigen::VectorXd VectorXd;
inline double mat4(const Matrix4d& m, int r, int c) { return m(r, c); }
inline Vector3d make_vec3(double x, double y, double z) { return Vector3d(x, y, z); }
inline bool cloud_is_dense(const PointCloud& c) { return c.is_dense; }
inline double* resize_3xn(Matrix3Xd& m, std::size_t n)  // column-major storage of a 3 x n matrix
{
  m.resize(3, (Eigen::Index) n);
  return m.data();
}
inline int loadPCDFile(const std::string& f, PointCloud& c) { return pcl::io::loadPCDFile<pcl::PointXYZRGBA>(f, c); }
inline void remove_nan_in_place(PointCloud& c)  // localization.cpp:26-27
{
  std::vector<int> nan_indices;
  pcl::removeNaNFromPointCloud(c, c, nan_indices);
}
}  // namespace agile_grasp_amd

#else  // stand-ins

namespace agile_grasp_amd
{
struct Vector3d
{
  double v[3];
  Vector3d() : v{ 0, 0, 0 } {}
  Vector3d(double x, double y, double z) : v{ x, y, z } {}
  double operator()(int i) const { return v[i]; }
  double& operator()(int i) { return v[i]; }
};
struct Matrix4d
{
  double m[4][4];
  Matrix4d() : m{ { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } } {}
  double operator()(int r, int c) const { return m[r][c]; }
  double& operator()(int r, int c) { return m[r][c]; }
};
struct VectorXi
{
  std::vector<int> d;
  VectorXi() {}
  explicit VectorXi(std::size_t n) : d(n, 0) {}
  int operator()(std::size_t i) const { return d[i]; }
  int& operator()(std::size_t i) { return d[i]; }
  std::size_t size() const { return d.size(); }
  const int* data() const { return d.data(); }
};
struct VectorXd
{
  std::vector<double> d;
  VectorXd() {}
  explicit VectorXd(std::size_t n) : d(n, 0.0) {}
  double operator()(std::size_t i) const { return d[i]; }
  double& operator()(std::size_t i) { return d[i]; }
  std::size_t size() const { return d.size(); }
};
struct Matrix3Xd  // 3 x n, column-major like Eigen
{
  std::vector<double> d;
  double operator()(int r, std::size_t c) const { return d[3 * c + r]; }
  std::size_t cols() const { return d.size() / 3; }
};
inline double* resize_3xn(Matrix3Xd& m, std::size_t n)
{
  m.d.assign(3 * n, 0.0);
  return m.d.data();
}
struct PointXYZRGBA  // pcl::PointXYZRGBA: 32 bytes, xyz at offset 0
{
  float x, y, z, pad0;
  std::uint32_t rgba;
  float pad1[3];
};
struct PointCloud
{
  typedef std::shared_ptr<PointCloud> Ptr;
  typedef PointXYZRGBA PointType;
  std::vector<PointXYZRGBA> points;
  bool is_dense = false;  // pcl::PointCloud::is_dense: "no point has a non-finite coordinate"
  std::size_t size() const { return points.size(); }
};
inline double mat4(const Matrix4d& m, int r, int c) { return m(r, c); }
inline Vector3d make_vec3(double x, double y, double z) { return Vector3d(x, y, z); }
inline bool cloud_is_dense(const PointCloud& c) { return c.is_dense; }
// pcl::removeNaNFromPointCloud(c, c, idx) (localization.cpp:26-27): a dense cloud is passed through, otherwise the points
// with a non-finite coordinate are dropped in place and the cloud is marked dense
inline void remove_nan_in_place(PointCloud& c)
{
  if (c.is_dense)
    return;
  std::size_t j = 0;
  for (std::size_t i = 0; i < c.points.size(); i++)
  {
    const PointXYZRGBA& p = c.points[i];
    if (p.x - p.x != 0.0f || p.y - p.y != 0.0f || p.z - p.z != 0.0f)  // NaN or infinity
      continue;
    c.points[j++] = p;
  }
  c.points.resize(j);
  c.is_dense = true;
}
}  // namespace agile_grasp_amd
#endif

static_assert(sizeof(float) == 4, "float32 expected");

#endif
