// API-shape declaration of pcl::PointXYZRGBA (PCL 1.7 point_types.h).  TEST-ONLY.
#ifndef AGH_TEST_STUB_PCL_POINT_TYPES
#define AGH_TEST_STUB_PCL_POINT_TYPES
#include <cstdint>
namespace pcl
{
struct alignas(16) PointXYZRGBA  // 32 bytes: float data[4] (x, y, z, pad) then the colour union, padded to 16
{
  union
  {
    float data[4];
    struct
    {
      float x, y, z;
    };
  };
  union
  {
    struct
    {
      std::uint8_t b, g, r, a;
    };
    std::uint32_t rgba;
  };
  PointXYZRGBA();
};
static_assert(sizeof(PointXYZRGBA) == 32, "pcl::PointXYZRGBA is 32 bytes");
}  // namespace pcl
#endif
