// API-shape declaration of pcl::removeNaNFromPointCloud (PCL 1.7 filters/filter.h).  TEST-ONLY.
#ifndef AGH_TEST_STUB_PCL_FILTER
#define AGH_TEST_STUB_PCL_FILTER
#include <pcl/point_cloud.h>
#include <vector>
namespace pcl
{
template <typename PointT>
void removeNaNFromPointCloud(const pcl::PointCloud<PointT>& cloud_in, pcl::PointCloud<PointT>& cloud_out, std::vector<int>& index);
}
#endif
