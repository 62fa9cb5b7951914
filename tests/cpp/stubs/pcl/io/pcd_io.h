// API-shape declaration of pcl::io::loadPCDFile (PCL 1.7 io/pcd_io.h).  TEST-ONLY.
#ifndef AGH_TEST_STUB_PCL_PCD_IO
#define AGH_TEST_STUB_PCL_PCD_IO
#include <pcl/point_cloud.h>
#include <string>
namespace pcl
{
namespace io
{
template <typename PointT>
int loadPCDFile(const std::string& file_name, pcl::PointCloud<PointT>& cloud);
}
}  // namespace pcl
#endif
