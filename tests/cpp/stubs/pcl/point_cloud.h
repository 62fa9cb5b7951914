// API-shape declaration of pcl::PointCloud<PointT> (PCL 1.7 point_cloud.h).  TEST-ONLY.
#ifndef AGH_TEST_STUB_PCL_POINT_CLOUD
#define AGH_TEST_STUB_PCL_POINT_CLOUD
#include <Eigen/Dense>
#include <boost/shared_ptr.hpp>
#include <cstdint>
#include <vector>
namespace pcl
{
template <typename PointT>
class PointCloud
{
public:
  typedef PointT PointType;
  typedef std::vector<PointT, Eigen::aligned_allocator<PointT> > VectorType;
  typedef boost::shared_ptr<PointCloud<PointT> > Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT> > ConstPtr;
  PointCloud();
  PointCloud(const PointCloud&);
  PointCloud& operator=(const PointCloud&);
  PointCloud& operator+=(const PointCloud& rhs);
  const PointCloud operator+(const PointCloud& rhs);
  std::size_t size() const;
  VectorType points;
  std::uint32_t width;
  std::uint32_t height;
  bool is_dense;
};
}  // namespace pcl
#endif
