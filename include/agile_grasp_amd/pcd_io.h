// pcd_io.h -- PCD files without PCL, and the agile_grasp/Grasp(s) message fields without ROS.
//
// loadPCDFile stands in for pcl::io::loadPCDFile<pcl::PointXYZRGBA> as Localization::localizeHands calls it
// (localization.cpp:183-200): PCD v0.7 headers, DATA ascii and DATA binary, fields x y z (float32/float64) and an
// optional 4-byte rgb / rgba field; binary_compressed is refused with a message.  is_dense is set the way PCL's reader
// leaves it: true unless a non-finite coordinate was read.
// Grasp / Grasps mirror msg/Grasp.msg and msg/Grasps.msg; createGraspMsg etc. fill them exactly as
// GraspLocalizer::createGraspMsg / createGraspsMsg / createGraspsMsgFromHands do (grasp_localizer.cpp:107-180).
#ifndef AGILE_GRASP_AMD_PCD_IO_H
#define AGILE_GRASP_AMD_PCD_IO_H

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "grasp_hypothesis.h"
#include "handle_search.h"
#include "types.h"

namespace agile_grasp_amd
{

#ifndef AGILE_GRASP_AMD_HAVE_PCL_EIGEN
/** @return 0 on success, -1 on failure (pcl::io::loadPCDFile's convention) */
inline int loadPCDFile(const std::string& file_name, PointCloud& cloud)
{
  std::ifstream in(file_name.c_str(), std::ios::binary);
  if (!in)
    return -1;
  std::vector<std::string> fields, types;
  std::vector<int> sizes, counts;
  long long width = 0, height = 1, points = -1;
  bool has_points = false;
  std::string data_kind, line;
  while (std::getline(in, line))
  {
    if (!line.empty() && line[line.size() - 1] == '\r')
      line.erase(line.size() - 1);
    if (line.empty() || line[0] == '#')
      continue;
    std::istringstream ls(line);
    std::string key;
    ls >> key;
    if (key == "FIELDS" || key == "COLUMNS")
      for (std::string f; ls >> f;)
        fields.push_back(f);
    else if (key == "SIZE")
      for (int v; ls >> v;)
        sizes.push_back(v);
    else if (key == "TYPE")
      for (std::string t; ls >> t;)
        types.push_back(t);
    else if (key == "COUNT")
      for (int v; ls >> v;)
        counts.push_back(v);
    else if (key == "WIDTH")
      ls >> width;
    else if (key == "HEIGHT")
      ls >> height;
    else if (key == "POINTS")
    {
      if (!(ls >> points))
        return -1;
      has_points = true;
    }
    else if (key == "DATA")
    {
      ls >> data_kind;
      break;
    }
  }
  if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size())
    return -1;
  if (counts.empty())
    counts.assign(fields.size(), 1);
  if (counts.size() != fields.size())
    return -1;
  // the header is untrusted input: every number is checked before it sizes a buffer or an offset
  const long long kMaxPoints = 1ll << 30;  // agh_set_cloud's limit
  if (width < 0 || height < 0 || width > kMaxPoints || height > kMaxPoints)
    return -1;
  if (!has_points)
    points = width * height;
  if (points < 0 || points > kMaxPoints)
    return -1;
  for (std::size_t f = 0; f < fields.size(); f++)
  {
    if (sizes[f] != 1 && sizes[f] != 2 && sizes[f] != 4 && sizes[f] != 8)
      return -1;
    if (counts[f] < 1 || counts[f] > 65536)
      return -1;
    if (types[f] != "F" && types[f] != "U" && types[f] != "I")
      return -1;
  }
  std::vector<int> offset(fields.size(), 0);
  int ix = -1, iy = -1, iz = -1, irgb = -1, point_bytes = 0, n_cols = 0;
  for (std::size_t f = 0; f < fields.size(); f++)
  {
    offset[f] = point_bytes;
    point_bytes += sizes[f] * counts[f];
    n_cols += counts[f];
    if (fields[f] == "x")
      ix = (int) f;
    else if (fields[f] == "y")
      iy = (int) f;
    else if (fields[f] == "z")
      iz = (int) f;
    else if ((fields[f] == "rgb" || fields[f] == "rgba") && sizes[f] == 4)
      irgb = (int) f;
  }
  if (ix < 0 || iy < 0 || iz < 0 || point_bytes <= 0 || point_bytes > (1 << 24))
    return -1;
  try
  {
    cloud.points.assign((std::size_t) points, PointXYZRGBA());
  }
  catch (const std::exception&)  // a header that promises more points than memory holds: the documented -1, no throw
  {
    return -1;
  }
  bool dense = true;
  if (data_kind == "ascii")
  {
    for (long long p = 0; p < points; p++)
    {
      if (!std::getline(in, line))
        return -1;
      std::istringstream ls(line);
      PointXYZRGBA& q = cloud.points[(std::size_t) p];
      for (std::size_t f = 0; f < fields.size(); f++)
        for (int c = 0; c < counts[f]; c++)
        {
          std::string tok;
          if (!(ls >> tok))
            return -1;
          if (c != 0)
            continue;
          if ((int) f == ix || (int) f == iy || (int) f == iz)
          {
            const float v = (tok == "nan" || tok == "NaN") ? NAN : (float) std::strtod(tok.c_str(), nullptr);
            ((int) f == ix ? q.x : (int) f == iy ? q.y : q.z) = v;
          }
          else if ((int) f == irgb)
          {
            // PCL writes the packed colour as the float (TYPE F) or integer (TYPE U) whose bits are the colour
            if (types[f] == "F")
            {
              const float v = (float) std::strtod(tok.c_str(), nullptr);
              std::memcpy(&q.rgba, &v, 4);
            }
            else
              q.rgba = (std::uint32_t) std::strtoul(tok.c_str(), nullptr, 10);
          }
        }
    }
  }
  else if (data_kind == "binary")
  {
    std::vector<char> buf((std::size_t) point_bytes);
    for (long long p = 0; p < points; p++)
    {
      in.read(buf.data(), point_bytes);
      if (in.gcount() != point_bytes)
        return -1;
      PointXYZRGBA& q = cloud.points[(std::size_t) p];
      const int idx3[3] = { ix, iy, iz };
      float* dst[3] = { &q.x, &q.y, &q.z };
      for (int a = 0; a < 3; a++)
      {
        const int f = idx3[a];
        if (sizes[(std::size_t) f] == 4 && types[(std::size_t) f] == "F")
          std::memcpy(dst[a], buf.data() + offset[(std::size_t) f], 4);
        else if (sizes[(std::size_t) f] == 8 && types[(std::size_t) f] == "F")
        {
          double v;
          std::memcpy(&v, buf.data() + offset[(std::size_t) f], 8);
          *dst[a] = (float) v;
        }
        else
          return -1;
      }
      if (irgb >= 0)
        std::memcpy(&q.rgba, buf.data() + offset[(std::size_t) irgb], 4);
    }
  }
  else
  {
    std::cout << " PCD DATA " << data_kind << " is not supported (ascii and binary are)\n";
    return -1;
  }
  for (std::size_t p = 0; p < cloud.points.size(); p++)
    if (!std::isfinite(cloud.points[p].x) || !std::isfinite(cloud.points[p].y) || !std::isfinite(cloud.points[p].z))
      dense = false;
  cloud.is_dense = dense;
  return 0;
}
#endif

// ---- sensor_msgs/PointCloud2 and agile_grasp/CloudSized -> the cloud localizeHands takes --------------
// GraspLocalizer::cloud_callback / cloud_sized_callback (grasp_localizer.cpp:40-78) call pcl::fromROSMsg on the incoming
// message; CloudSized (msg/CloudSized.msg) is a PointCloud2 plus the number of points of the left camera.  The structs
// below carry the wire fields of the two messages for builds without ROS; fromROSMsg is a template on the message type,
// so it reads a real sensor_msgs::PointCloud2 just as well (same member names).
struct PointField  // sensor_msgs/PointField
{
  enum { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name;
  std::uint32_t offset;
  std::uint8_t datatype;
  std::uint32_t count;
  PointField() : offset(0), datatype(0), count(1) {}
  PointField(const std::string& n, std::uint32_t o, std::uint8_t d, std::uint32_t c = 1) : name(n), offset(o), datatype(d), count(c) {}
};
struct MsgHeader  // std_msgs/Header (the part the node reads: grasp_localizer.cpp:44-51)
{
  std::uint32_t seq;
  std::string frame_id;
  MsgHeader() : seq(0) {}
};
struct PointCloud2  // sensor_msgs/PointCloud2
{
  MsgHeader header;
  std::uint32_t height, width;
  std::vector<PointField> fields;
  bool is_bigendian;
  std::uint32_t point_step, row_step;
  std::vector<std::uint8_t> data;
  bool is_dense;
  PointCloud2() : height(0), width(0), is_bigendian(false), point_step(0), row_step(0), is_dense(false) {}
};
struct Int64Msg  // std_msgs/Int64
{
  std::int64_t data;
  Int64Msg() : data(0) {}
};
struct CloudSized  // msg/CloudSized.msg:1-2
{
  PointCloud2 cloud;
  Int64Msg size_left;
};

/** pcl::fromROSMsg(msg, cloud) for pcl::PointXYZRGBA as grasp_localizer.cpp:55-57,73 uses it: x, y, z are read at their
 *  field offsets (FLOAT32, or FLOAT64 narrowed), the packed colour from a 4-byte "rgba" or "rgb" field if there is one,
 *  rows may be padded (row_step), is_dense is taken from the message.  The message is untrusted input: offsets and sizes
 *  are checked against point_step / data.size() first.  @return 0, or -1 (after printing why) on a malformed message */
template <typename CloudMsg>
inline int fromROSMsg(const CloudMsg& msg, PointCloud& cloud)
{
  cloud.points.clear();
  cloud.is_dense = msg.is_dense;
  int fx = -1, fy = -1, fz = -1, fc = -1;
  for (std::size_t f = 0; f < msg.fields.size(); f++)
  {
    const std::string& nm = msg.fields[f].name;
    if (nm == "x")
      fx = (int) f;
    else if (nm == "y")
      fy = (int) f;
    else if (nm == "z")
      fz = (int) f;
    else if ((nm == "rgba" || nm == "rgb") && fc < 0)
      fc = (int) f;
  }
  if (fx < 0 || fy < 0 || fz < 0)
  {
    std::cout << " Error: the PointCloud2 has no x / y / z fields\n";
    return -1;
  }
  if (msg.is_bigendian)
  {
    std::cout << " Error: big-endian PointCloud2 data is not supported\n";
    return -1;
  }
  const std::uint64_t step = msg.point_step, row = msg.row_step, w = msg.width, h = msg.height;
  const int idx[3] = { fx, fy, fz };
  for (int a = 0; a < 3; a++)
  {
    const unsigned dt = msg.fields[(std::size_t) idx[a]].datatype;
    const std::uint64_t sz = dt == PointField::FLOAT32 ? 4 : dt == PointField::FLOAT64 ? 8 : 0;
    if (sz == 0 || (std::uint64_t) msg.fields[(std::size_t) idx[a]].offset + sz > step)
    {
      std::cout << " Error: PointCloud2 field " << msg.fields[(std::size_t) idx[a]].name << " is not a float inside point_step\n";
      return -1;
    }
  }
  if (fc >= 0)
  {
    const unsigned dt = msg.fields[(std::size_t) fc].datatype;
    const bool four = dt == PointField::FLOAT32 || dt == PointField::UINT32 || dt == PointField::INT32;
    if (!four || (std::uint64_t) msg.fields[(std::size_t) fc].offset + 4 > step)
      fc = -1;  // like PCL: a colour field that does not match is skipped, the points are still read
  }
  if (w * h > (1ull << 30) || step == 0 || (w > 0 && row < w * step) || (h > 0 && (h - 1) * row + w * step > msg.data.size()))
  {
    std::cout << " Error: PointCloud2 width / height / point_step / row_step do not fit its data\n";
    return -1;
  }
  cloud.points.resize((std::size_t) (w * h));
  for (std::uint64_t r = 0; r < h; r++)
    for (std::uint64_t c = 0; c < w; c++)
    {
      const std::uint8_t* p = msg.data.data() + r * row + c * step;
      PointCloud::PointType& q = cloud.points[(std::size_t) (r * w + c)];
      float* dst[3] = { &q.x, &q.y, &q.z };
      for (int a = 0; a < 3; a++)
      {
        const std::uint8_t* src = p + msg.fields[(std::size_t) idx[a]].offset;
        if (msg.fields[(std::size_t) idx[a]].datatype == PointField::FLOAT32)
          std::memcpy(dst[a], src, 4);
        else
        {
          double v;
          std::memcpy(&v, src, 8);
          *dst[a] = (float) v;
        }
      }
      if (fc >= 0)
        std::memcpy(&q.rgba, p + msg.fields[(std::size_t) fc].offset, 4);
    }
  return 0;
}

/** cloud_sized_callback (grasp_localizer.cpp:63-78): the cloud and size_left of a CloudSized message. */
template <typename CloudSizedMsg>
inline int fromCloudSized(const CloudSizedMsg& msg, PointCloud& cloud, int& size_left)
{
  size_left = (int) msg.size_left.data;
  return fromROSMsg(msg.cloud, cloud);
}

/** msg/Grasp.msg:1-5 */
struct Grasp
{
  Vector3d center, axis, approach, surface_center;
  float width;  // std_msgs/Float32
};
/** msg/Grasps.msg (the header's stamp is the publisher's business) */
struct Grasps
{
  std::vector<Grasp> grasps;
};

/** grasp_localizer.cpp:137-146 */
inline Grasp createGraspMsg(const GraspHypothesis& hand)
{
  Grasp msg;
  msg.center = hand.getGraspBottom();
  msg.axis = hand.getAxis();
  msg.approach = hand.getApproach();
  msg.surface_center = hand.getGraspSurface();
  msg.width = (float) hand.getGraspWidth();
  return msg;
}
/** grasp_localizer.cpp:171-180 */
inline Grasp createGraspMsg(const Handle& handle)
{
  Grasp msg;
  msg.center = handle.getCenter();
  msg.axis = handle.getAxis();
  msg.approach = handle.getApproach();
  msg.surface_center = handle.getHandsCenter();
  msg.width = (float) handle.getWidth();
  return msg;
}
/** grasp_localizer.cpp:123-134 */
inline Grasps createGraspsMsg(const std::vector<GraspHypothesis>& hands)
{
  Grasps msg;
  for (std::size_t i = 0; i < hands.size(); i++)
    msg.grasps.push_back(createGraspMsg(hands[i]));
  return msg;
}
/** grasp_localizer.cpp:107-120 */
inline Grasps createGraspsMsg(const std::vector<Handle>& handles)
{
  Grasps msg;
  for (std::size_t i = 0; i < handles.size(); i++)
    msg.grasps.push_back(createGraspMsg(handles[i]));
  return msg;
}
/** grasp_localizer.cpp:149-168: every hand of every handle */
inline Grasps createGraspsMsgFromHands(const std::vector<Handle>& handles)
{
  Grasps msg;
  for (std::size_t i = 0; i < handles.size(); i++)
  {
    const std::vector<GraspHypothesis>& hands = handles[i].getHandList();
    const std::vector<int>& inliers = handles[i].getInliers();
    for (std::size_t j = 0; j < inliers.size(); j++)
      msg.grasps.push_back(createGraspMsg(hands[(std::size_t) inliers[j]]));
  }
  return msg;
}

}  // namespace agile_grasp_amd
#endif
