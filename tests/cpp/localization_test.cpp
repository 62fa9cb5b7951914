// localization_test.cpp -- drives the Localization facade (include/agile_grasp_amd/localization.h) on a RAW cloud
// (NaNs, points outside the workspace, no voxelisation), like src/tests/test_local_axes.cpp drives the reference's.
//   localization_test <raw.bin> <svm file> <mode: voxels|hands|antipodal>   (antipodal: src/tests/antipodal_test.cpp)
// raw.bin: int64 n, int64 size_left, int64 n_idx, double ws[6], double cam_left[3], double cam_right[3], n*3 float xyz,
// n_idx int32 indices (into the voxelised cloud).
#include <cstdio>
#include <cstring>
#include <vector>

#include "agile_grasp_amd/localization.h"

using namespace agile_grasp_amd;

int main(int argc, char** argv)
{
  if (argc < 4)
    return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f)
    return 2;
  long long n = 0, size_left = 0, n_idx = 0;
  double ws[6], cl[3], cr[3];
  if (std::fread(&n, 8, 1, f) != 1 || std::fread(&size_left, 8, 1, f) != 1 || std::fread(&n_idx, 8, 1, f) != 1 ||
      std::fread(ws, 8, 6, f) != 6 || std::fread(cl, 8, 3, f) != 3 || std::fread(cr, 8, 3, f) != 3)
    return 2;
  std::vector<float> xyz(3 * (size_t) n);
  std::vector<int> idx((size_t) n_idx);
  if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || std::fread(idx.data(), 4, idx.size(), f) != idx.size())
    return 2;
  std::fclose(f);
  PointCloud::Ptr cloud(new PointCloud);
  cloud->points.resize((size_t) n);
  for (long long i = 0; i < n; i++)
  {
    cloud->points[(size_t) i].x = xyz[3 * i];
    cloud->points[(size_t) i].y = xyz[3 * i + 1];
    cloud->points[(size_t) i].z = xyz[3 * i + 2];
  }
  Matrix4d tl, tr;
  for (int r = 0; r < 3; r++)
  {
    tl(r, 3) = cl[r];
    tr(r, 3) = cr[r];
  }
  Localization loc(1, false, 0);
  loc.setCameraTransforms(tl, tr);
  VectorXd w(6);
  for (int i = 0; i < 6; i++)
    w(i) = ws[i];
  loc.setWorkspace(w);
  loc.setDeterministicNormalEstimation(true);
  if (std::strcmp(argv[3], "voxels") == 0)
  {
    // preprocessing as localizeHands runs it (on the GPU): print the cloud the search worked on
    std::vector<GraspHypothesis> hands = loc.localizeHands(cloud, (int) size_left, idx, false, false);
    const PointCloud::Ptr& vox = loc.getSearchedCloud();
    if (!vox)
    {
      std::printf("NO_CLOUD\n");
      return 0;
    }
    std::printf("VOXELS %zu\n", vox->points.size());
    for (size_t i = 0; i < vox->points.size(); i++)
      std::printf("V %.9g %.9g %.9g %d\n", vox->points[i].x, vox->points[i].y, vox->points[i].z,
        (int) loc.getSearchedCamSource()((int) i));
    return 0;
  }
  if (std::strcmp(argv[3], "chain") == 0)
  {
    // grasp_localizer.cpp:95-103 twice: the three calls of the reference's caller, then the one-call form; the two must print
    // the same kept hands and the same handles
    std::vector<GraspHypothesis> hands3 = loc.localizeHands(cloud, (int) size_left, idx, false, false);
    std::vector<GraspHypothesis> kept3 = loc.predictAntipodalHands(hands3, argv[2]);
    std::vector<Handle> handles3 = loc.findHandles(kept3, 2, 0.005);
    std::vector<GraspHypothesis> kept1;
    std::vector<Handle> handles1 = loc.localizeHandles(cloud, (int) size_left, idx, argv[2], 2, 0.005, &kept1);
    for (int pass = 0; pass < 2; pass++)
    {
      const std::vector<GraspHypothesis>& kept = pass == 0 ? kept3 : kept1;
      const std::vector<Handle>& handles = pass == 0 ? handles3 : handles1;
      std::printf("CHAIN%d %zu %zu\n", pass == 0 ? 3 : 1, kept.size(), handles.size());
      for (size_t i = 0; i < kept.size(); i++)
        std::printf("K%d %.17g %.17g %.17g %.17g %d\n", pass == 0 ? 3 : 1, kept[i].getGraspSurface()(0), kept[i].getGraspBottom()(1),
          kept[i].getApproach()(2), kept[i].getGraspWidth(), kept[i].isFullAntipodal() ? 1 : 0);
      for (size_t i = 0; i < handles.size(); i++)
      {
        std::printf("G%d %zu %.17g %.17g %.17g %.17g %zu", pass == 0 ? 3 : 1, handles[i].getInliers().size(), handles[i].getAxis()(0),
          handles[i].getCenter()(1), handles[i].getBinormal()(2), handles[i].getWidth(), handles[i].getHandList().size());
        for (size_t k = 0; k < handles[i].getInliers().size(); k++)
          std::printf(" %d", handles[i].getInliers()[k]);
        std::printf("\n");
      }
    }
    return 0;
  }
  if (std::strcmp(argv[3], "stream") == 0)
  {
    // a node that holds the next capture while this one is searched: localizeHandlesBegin / stageNextCloud / localizeHandlesEnd
    // over three captures (copies of the cloud: distinct objects) against localizeHandles on the first
    PointCloud::Ptr clouds[3];
    for (int k = 0; k < 3; k++)
      clouds[k] = PointCloud::Ptr(new PointCloud(*cloud));
    std::vector<GraspHypothesis> kept1;
    PointCloud::Ptr ref_cloud(new PointCloud(*cloud));
    std::vector<Handle> handles1 = loc.localizeHandles(ref_cloud, (int) size_left, idx, argv[2], 2, 0.005, &kept1);
    std::printf("CHAIN1 %zu %zu\n", kept1.size(), handles1.size());
    if (!loc.localizeHandlesBegin(clouds[0], (int) size_left, idx, argv[2], 2, 0.005))
      return 3;
    for (int k = 0; k < 3; k++)
    {
      if (k + 1 < 3 && !loc.stageNextCloud(clouds[k + 1]))
        return 4;
      std::vector<GraspHypothesis> kept;
      std::vector<Handle> handles = loc.localizeHandlesEnd(&kept);
      if (k + 1 < 3 && !loc.localizeHandlesBegin(clouds[k + 1], (int) size_left, idx, argv[2], 2, 0.005))
        return 5;
      bool same = kept.size() == kept1.size() && handles.size() == handles1.size();
      for (size_t i = 0; same && i < kept.size(); i++)
        same = kept[i].getGraspSurface()(0) == kept1[i].getGraspSurface()(0) && kept[i].getGraspBottom()(1) == kept1[i].getGraspBottom()(1) &&
               kept[i].getGraspWidth() == kept1[i].getGraspWidth();
      for (size_t i = 0; same && i < handles.size(); i++)
        same = handles[i].getInliers() == handles1[i].getInliers() && handles[i].getAxis()(0) == handles1[i].getAxis()(0) &&
               handles[i].getCenter()(1) == handles1[i].getCenter()(1) && handles[i].getWidth() == handles1[i].getWidth();
      std::printf("STREAM %d %zu %zu %d\n", k, kept.size(), handles.size(), same ? 1 : 0);
    }
    return 0;
  }
  const bool antipodal = std::strcmp(argv[3], "antipodal") == 0;  // calculates_antipodal (antipodal_test.cpp:61)
  std::vector<GraspHypothesis> hands = loc.localizeHands(cloud, (int) size_left, idx, antipodal, false);
  if (antipodal)
  {
    for (size_t i = 0; i < hands.size(); i++)
      std::printf("A %d %d\n", hands[i].isHalfAntipodal() ? 1 : 0, hands[i].isFullAntipodal() ? 1 : 0);
    return 0;
  }
  std::vector<GraspHypothesis> kept = loc.predictAntipodalHands(hands, argv[2]);
  std::printf("RESULT %zu %zu %zu\n", loc.getSearchedCloud() ? loc.getSearchedCloud()->size() : (size_t) 0, hands.size(),
    kept.size());
  for (size_t i = 0; i < hands.size(); i++)
    std::printf("H %.17g %.17g %.17g %.17g\n", hands[i].getGraspSurface()(0), hands[i].getGraspBottom()(1),
      hands[i].getApproach()(2), hands[i].getGraspWidth());
  // grasp_localizer.cpp:103 runs the handle search on the SVM-positive hands; all hands give the test more material
  std::vector<Handle> handles = loc.findHandles(hands, 3, 0.005);
  for (size_t i = 0; i < handles.size(); i++)
    std::printf("HANDLE %zu %d %.17g %.17g %.17g %.17g\n", handles[i].getInliers().size(), handles[i].getInliers()[0],
      handles[i].getAxis()(0), handles[i].getCenter()(1), handles[i].getBinormal()(2), handles[i].getWidth());
  return 0;
}
