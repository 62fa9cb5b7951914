// The HOST-buffer entry points timed from plain C++ (no Python between the calls): what a ROS node calling the adapter's
// HandSearch::findHands pays per cloud.  Input: a cloud dump (tests/test_cpp_adapter.py::_dump layout:
// int64 n, int64 S, 6 doubles cam origins, n x 3 floats, n int32 cam ids, S int32 samples).
//   g++ -O2 -std=c++17 -Iinclude scripts/micro/host_api_c.cpp -o /tmp/host_api_c -Lagile_grasp_amd/lib -lagile_grasp_hip \
//       -Wl,-rpath,$PWD/agile_grasp_amd/lib -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
//   /tmp/host_api_c cloud.bin [calls] [svm_file]
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "agh.h"

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double median(std::vector<double> v)
{
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv)
{
  if (argc < 2)
    return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f)
    return 2;
  const int K = argc > 2 ? std::atoi(argv[2]) : 50;
  int64_t n = 0, S = 0;
  double cam[6];
  if (std::fread(&n, 8, 1, f) != 1 || std::fread(&S, 8, 1, f) != 1 || std::fread(cam, 8, 6, f) != 6)
    return 2;
  std::vector<float> xyz((size_t) n * 3);
  std::vector<int32_t> cs((size_t) n), idx((size_t) S);
  if (std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() || std::fread(cs.data(), 4, cs.size(), f) != cs.size() ||
      std::fread(idx.data(), 4, idx.size(), f) != idx.size())
    return 2;
  std::fclose(f);
  agh_params p;
  agh_default_params(&p);
  for (int k = 0; k < 6; k++)
    p.cam_origin[k / 3][k % 3] = cam[k];
  p.normals_mode = AGH_NORMALS_DETERMINISTIC;
  agh_ctx* ctx = nullptr;
  if (agh_create(&p, &ctx) != AGH_OK)
  {
    std::printf("{\"error\": \"%s\"}\n", agh_last_error(nullptr));
    return 1;
  }
  const bool classify = argc > 3 && agh_load_svm_file(ctx, argv[3]) == AGH_OK;
  std::vector<agh_hypothesis> out((size_t) S * 8);
  std::vector<uint8_t> keep((size_t) S * 8);
  int64_t n_out = 0, n_kept = 0;
  std::vector<double> t_set, t_find, t_cls, t_all;
  for (int it = 0; it < K + 5; it++)
  {
    const double t0 = now_us();
    int rc = agh_set_cloud(ctx, xyz.data(), 12, cs.data(), n);
    const double t1 = now_us();
    if (rc == AGH_OK)
      rc = agh_find_hands(ctx, idx.data(), S, 0, out.data(), (int64_t) out.size(), &n_out);
    const double t2 = now_us();
    if (rc == AGH_OK && classify)
      rc = agh_classify(ctx, keep.data(), (int64_t) keep.size(), &n_kept);
    const double t3 = now_us();
    if (rc != AGH_OK)
    {
      std::printf("{\"error\": \"%s\", \"rc\": %d}\n", agh_last_error(ctx), rc);
      return 1;
    }
    if (it >= 5)
    {
      t_set.push_back(t1 - t0);
      t_find.push_back(t2 - t1);
      t_cls.push_back(t3 - t2);
      t_all.push_back(t3 - t0);
    }
  }
  double mean_all = 0;
  for (double v : t_all)
    mean_all += v / t_all.size();
  std::printf("{\"what\": \"C ABI from C++: agh_set_cloud + agh_find_hands%s, host buffers\", \"calls\": %d, \"points\": %lld, "
              "\"samples\": %lld, \"hypotheses\": %lld, \"kept\": %lld, \"us_set_cloud_median\": %.1f, \"us_find_hands_median\": %.1f, "
              "\"us_classify_median\": %.1f, \"us_per_call_median\": %.1f, \"us_per_call_mean\": %.1f}\n",
    classify ? " + agh_classify" : "", K, (long long) n, (long long) S, (long long) n_out, (long long) n_kept, median(t_set),
    median(t_find), median(t_cls), median(t_all), mean_all);
  agh_destroy(ctx);
  return 0;
}
