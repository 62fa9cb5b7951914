// points.hip -- GraspHypothesis::getPointsForLearning / getIndicesPointsForLearningCam1/2 on demand
// (grasp_hypothesis.h:149-170; filled at rotating_hand.cpp:125-157).
//
// The reference stores, in every hypothesis, the 3 x n_b matrix of the hand-box points (hand frame, offset by the grasp
// surface) and their split by camera: hundreds of MB per cloud.  The hot path keeps only the rasterised images; this
// entry point recomputes the matrix of ONE hypothesis when a caller (plotting, inspection) asks for it.
// One work-group scans the whole cell-sorted cloud twice -- a lazy getter, not a hot path:
//   pass 1: min over the cropped neighbourhood of the rotated y (FingerHand::evaluateGraspParameters' grasp surface,
//           finger_hand.cpp:135-136);
//   pass 2: emit the columns  points_rot.col(j) - surface  for the points with y < back_of_hand + hand_depth
//           (rotating_hand.cpp:125-139), each with its FLANN distance and cloud index; the host orders them as
//           radiusSearch returned them (ascending (d2, index)), which is the column order of the reference.
#include "agh_internal.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

namespace agh
{

struct LpEntry
{
  double x, y, z;
  float d2;
  int id;  // (cloud index << 1) | camera
};

__global__ __launch_bounds__(1024) void k_learning_points(GridView gv, const int32_t* __restrict__ scloud, const HandGeom* __restrict__ geom_p,
  const agh_frame* __restrict__ frames, const agh_hypothesis* __restrict__ hyps, int64_t hyp, float r2f,
  LpEntry* __restrict__ out, int cap, int* __restrict__ n_out)
{
  __shared__ double red[16];
  __shared__ int cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const agh_hypothesis H = hyps[hyp];
  const agh_frame F = frames[H.sample];
  // the points of the hypothesis' own cloud of the batch: [cloud_off[k], cloud_off[k + 1]) of the sorted array
  const int64_t k_first = gv.cloud_off[scloud[H.sample]], n_points = gv.cloud_off[scloud[H.sample] + 1];
  const HandGeom& G = *geom_p;
  const int o = H.orientation, e = H.finger_index < 0 ? 0 : H.finger_index, last = H.depth_index;
  double fr[3][3];
  {
    const double* nm = F.normal;
    const double* ax = F.axis;
    const double nxa[3] = { nm[1] * ax[2] - nm[2] * ax[1], nm[2] * ax[0] - nm[0] * ax[2], nm[0] * ax[1] - nm[1] * ax[0] };
    for (int r = 0; r < 3; r++)
    {
      fr[r][0] = nm[r];
      fr[r][1] = nxa[r];
      fr[r][2] = ax[r];
    }
  }
  const float sx = (float) F.sample[0], sy = (float) F.sample[1], sz = (float) F.sample[2];  // hand_search.cpp:141-144
  const double hh = G.hand_height;
  const double cs = G.cos_a[o], sn = G.sin_a[o], ms = -1.0 * sn;
  auto transform = [&](const float4& p, bool& keep, double& xr, double& yr, double& tz, float& d2) {
    d2 = flann_d2(sx, sy, sz, p.x, p.y, p.z);
    const double cx = (double) (p.x - sx), cy = (double) (p.y - sy), cz = (double) (p.z - sz);  // 157-158
    tz = (fr[0][2] * cx + fr[1][2] * cy) + fr[2][2] * cz;
    keep = d2 < r2f && (tz > -1.0 * hh) && (tz < hh);  // radius search, then the crop of rotating_hand.cpp:37-51
    const double tx = (fr[0][0] * cx + fr[1][0] * cy) + fr[2][0] * cz;
    const double ty = (fr[0][1] * cx + fr[1][1] * cy) + fr[2][1] * cz;
    xr = cs * tx + ms * ty;  // rot * points_ (rotating_hand.cpp:91)
    yr = sn * tx + cs * ty;
  };
  // pass 1
  double ymin = INFINITY;
  for (int64_t k = k_first + tid; k < n_points; k += 1024)
  {
    bool keep;
    double xr, yr, tz;
    float d2;
    transform(gv.sorted[k], keep, xr, yr, tz, d2);
    if (keep)
      ymin = fmin(ymin, yr);
  }
  for (int off = 32; off >= 1; off >>= 1)
    ymin = fmin(ymin, __shfl_xor(ymin, off));
  if (lane == 0)
    red[wave] = ymin;
  if (tid == 0)
    cnt = 0;
  __syncthreads();
  ymin = red[0];
  for (int w = 1; w < 16; w++)
    ymin = fmin(ymin, red[w]);
  // surface = frame_ * rot^T * [hor_pos, min_y, 0] (rotating_hand.cpp:118-121), hand-local, before `+= sample`
  const double rot[3][3] = { { cs, -1.0 * sn, 0.0 }, { sn, cs, 0.0 }, { 0.0, 0.0, 1.0 } };
  const double hor_pos = (G.hand_outer_diameter / 2.0) + (G.fs[e] / 1);
  double surf[3];
  for (int i = 0; i < 3; i++)
  {
    double T[3];
    for (int j = 0; j < 3; j++)
      T[j] = (fr[i][0] * rot[j][0] + fr[i][1] * rot[j][1]) + fr[i][2] * rot[j][2];
    surf[i] = (T[0] * hor_pos + T[1] * ymin) + T[2] * 0.0;
  }
  const double box_y = G.boxy[last];
  // pass 2
  for (int64_t k = k_first + tid; k < n_points; k += 1024)
  {
    bool keep;
    double xr, yr, tz;
    float d2;
    const float4 p = gv.sorted[k];
    transform(p, keep, xr, yr, tz, d2);
    if (keep && yr < box_y)
    {
      const int pos = atomicAdd(&cnt, 1);
      if (pos < cap)
      {
        LpEntry en;
        en.x = xr - surf[0];  // rotating_hand.cpp:138: a world-frame offset from hand-frame points, as in the reference
        en.y = yr - surf[1];
        en.z = tz - surf[2];
        en.d2 = d2;
        en.id = (int) __float_as_uint(p.w);
        out[pos] = en;
      }
    }
  }
  __syncthreads();
  if (tid == 0)
    *n_out = cnt;
}

}  // namespace agh

using namespace agh;

extern "C" int agh_get_learning_points(agh_ctx* ctx, int64_t hyp, double* points, int32_t* cam_source, int64_t cap,
  int64_t* n_out)
{
  if (!ctx || !n_out)
    return AGH_ERR_INVALID_ARGUMENT;
  Ctx* c = &ctx->c;
  *n_out = 0;
  if (c->last_nout < 0 || !c->d_out_last || !c->has_cloud)
  {
    c->err = "agh_get_learning_points: needs the hypotheses of a completed agh_find_hands call";
    return AGH_ERR_STATE;
  }
  if (hyp < 0 || hyp >= c->last_nout)
  {
    c->err = "agh_get_learning_points: hypothesis index out of range";
    return AGH_ERR_INVALID_ARGUMENT;
  }
  if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
    return AGH_ERR_HIP;
  agh_hypothesis h;
  if (hipMemcpy(&h, c->d_out_last + hyp, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
    return AGH_ERR_HIP;
  const int n_b = h.n_in_box;
  *n_out = n_b;
  if (n_b > cap || (n_b > 0 && (!points || !cam_source)))
  {
    c->err = "agh_get_learning_points: " + std::to_string(n_b) + " points, room for " + std::to_string(cap);
    return AGH_ERR_CAPACITY;
  }
  if (n_b == 0)
    return AGH_OK;
  LpEntry* d_out = nullptr;
  int* d_n = nullptr;
  int rc = AGH_OK;
  std::vector<LpEntry> ent((size_t) n_b);
  int n_dev = 0;
  if (hipMalloc((void**) &d_out, sizeof(LpEntry) * (size_t) n_b) != hipSuccess || hipMalloc((void**) &d_n, sizeof(int)) != hipSuccess)
    rc = AGH_ERR_HIP;
  if (rc == AGH_OK)
  {
    GridView gv{ c->d_desc, c->d_cell_start, c->d_sorted, c->d_cloud_off, c->n_clouds };
    const double radius = c->p.nn_radius_hands;
    hipLaunchKernelGGL(k_learning_points, dim3(1), dim3(1024), 0, c->stream, gv, (const int32_t*) c->d_scloud, (const HandGeom*) c->d_geom,
      (const agh_frame*) c->d_frames, (const agh_hypothesis*) c->d_out_last, hyp, static_cast<float>(radius * radius), d_out,
      n_b, d_n);
    if (hipMemcpyAsync(ent.data(), d_out, sizeof(LpEntry) * (size_t) n_b, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(&n_dev, d_n, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      rc = AGH_ERR_HIP;
  }
  if (d_out)
    (void) hipFree(d_out);
  if (d_n)
    (void) hipFree(d_n);
  if (rc != AGH_OK)
  {
    c->err = "agh_get_learning_points: device allocation, launch or copy failed";
    return rc;
  }
  if (n_dev != n_b)  // the sweep counted n_b with the same arithmetic: a mismatch is a bug, not a capacity problem
  {
    c->err = "agh_get_learning_points: recount differs from the hypothesis' n_in_box";
    return AGH_ERR_STATE;
  }
  std::vector<int> perm((size_t) n_b);
  std::iota(perm.begin(), perm.end(), 0);
  std::sort(perm.begin(), perm.end(), [&](int a, int b) {  // radiusSearch order: ascending (distance, index)
    return ent[(size_t) a].d2 != ent[(size_t) b].d2 ? ent[(size_t) a].d2 < ent[(size_t) b].d2
                                                    : (ent[(size_t) a].id >> 1) < (ent[(size_t) b].id >> 1);
  });
  for (int k = 0; k < n_b; k++)
  {
    const LpEntry& en = ent[(size_t) perm[(size_t) k]];
    points[3 * (size_t) k + 0] = en.x;
    points[3 * (size_t) k + 1] = en.y;
    points[3 * (size_t) k + 2] = en.z;
    cam_source[k] = en.id & 1;
  }
  return AGH_OK;
}
