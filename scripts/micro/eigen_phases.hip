// Where one lane's chain of k_taubin_eigen spends its cycles: shader-clock stamps at the phase boundaries of
// taubin_smallest_eigenpair (agile_grasp_amd/csrc/taubin_eigen.h), one wave, and the same with W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I agile_grasp_amd/csrc -o scripts/micro/eigen_phases \
//     scripts/micro/eigen_phases.hip && scripts/micro/eigen_phases
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

// a truncation point: everything the phases so far produced is folded into the result, so nothing before it is dead
// code and nothing after it is computed
#define AGH_SUM_LOWER(A) ([&] { double t_ = 0; for (int i_ = 0; i_ < 9; i_++) for (int j_ = 0; j_ <= i_; j_++) t_ += A[i_][j_]; return t_; }())
#define AGH_SUM_STRICT(A) ([&] { double t_ = 0; for (int i_ = 1; i_ < 9; i_++) for (int j_ = 0; j_ < i_; j_++) t_ += A[i_][j_]; return t_; }())
#define AGH_SUM_VEC(A, N_) ([&] { double t_ = 0; for (int i_ = 0; i_ < N_; i_++) t_ += A[i_]; return t_; }())
#define AGH_EIG_STAMP0 if (STOP == 0) return AGH_SUM_LOWER(C) + AGH_SUM_VEC(b, 9)
#define AGH_EIG_STAMP1 if (STOP == 1) return AGH_SUM_LOWER(C) + AGH_SUM_VEC(b, 9) + AGH_SUM_STRICT(L) + AGH_SUM_VEC(rinv, 9)
#define AGH_EIG_STAMP2 if (STOP == 2) return AGH_SUM_LOWER(C) + AGH_SUM_VEC(b, 9) + AGH_SUM_STRICT(L) + AGH_SUM_VEC(rinv, 9)
#define AGH_EIG_STAMP3 if (STOP == 3) return AGH_SUM_LOWER(C) + AGH_SUM_VEC(b, 9) + AGH_SUM_STRICT(L) + AGH_SUM_VEC(rinv, 9) + AGH_SUM_VEC(RH, 7) + AGH_SUM_VEC(e, 8)
#define AGH_EIG_STAMP4 if (STOP == 4) return AGH_SUM_LOWER(C) + AGH_SUM_VEC(b, 9) + AGH_SUM_STRICT(L) + AGH_SUM_VEC(rinv, 9) + AGH_SUM_VEC(RH, 7) + AGH_SUM_VEC(e, 8) + lo + hi
#define AGH_EIG_STAMP5 if (STOP == 5) return AGH_SUM_LOWER(C) + AGH_SUM_VEC(b, 9) + AGH_SUM_STRICT(L) + AGH_SUM_VEC(rinv, 9) + AGH_SUM_VEC(RH, 7) + AGH_SUM_VEC(z, 9)
#define AGH_EIG_STAMP6 if (STOP == 6) return AGH_SUM_VEC(b, 9) + AGH_SUM_VEC(v, 9)
#define AGH_EIG_TEMPLATE template <int STOP, int LPS = 1>
#include "taubin_eigen.h"

using namespace agh;

template <int STOP>
__global__ __launch_bounds__(64) void k_phase(const double* sums, const double* ns, double* out)
{
  double sv[kNumSums];
  const int s = blockIdx.x * 64 + threadIdx.x;
  for (int k = 0; k < kNumSums; k++)
    sv[k] = sums[(size_t) s * kNumSums + k];
  double v[10];
  for (int k = 0; k < 10; k++)
    v[k] = 0.0;
  const double lam = taubin_smallest_eigenpair<STOP>(sv, ns[s], v);
  for (int k = 0; k < 10; k++)
    out[(size_t) s * 12 + k] = v[k];
  out[(size_t) s * 12 + 10] = lam;
}

template <int STOP>
static float run(int w, const double* d_s, const double* d_n, double* d_o)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++)
  {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_phase<STOP>, dim3(w), dim3(64), 0, 0, d_s, d_n, d_o);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  hipError_t err = hipDeviceSynchronize();
  double lam = 0;
  hipMemcpy(&lam, d_o + 10, 8, hipMemcpyDeviceToHost);
  if (w == 1)
    printf("  stop %d: %s, out[10] = %.6e\n", STOP, hipGetErrorString(err), lam);
  return best * 1000.0f;
}

int main()
{
  const int waves = 2048, S = waves * 64;
  std::vector<double> sums((size_t) S * kNumSums), ns(S);
  unsigned long long rs = 12345;
  auto rnd = [&]() { rs = rs * 6364136223846793005ull + 1442695040888963407ull; return (double) (rs >> 11) / 9007199254740992.0; };
  for (int s = 0; s < S; s++)
  {
    // a lattice patch of a gently curved surface around (0.7, 0.05, -0.05), 3 mm voxels, ~600 points
    double acc[kNumSums] = { 0 };
    const double cx = 0.7 + 0.1 * rnd(), cy = 0.05 + 0.1 * rnd(), cz = -0.05, k1 = 20.0 * rnd(), amp = 0.02 * rnd();
    int n = 0;
    for (int i = -10; i <= 10; i++)
      for (int j = -10; j <= 10; j++)
      {
        if (i * i + j * j > 100 || rnd() < 0.2)
          continue;
        const float xf = (float) (cx + 0.003 * i), yf = (float) (cy + 0.003 * j);
        const float zf = (float) (cz + std::floor(amp * std::sin(k1 * 0.003 * i) / 0.003) * 0.003);
        const double x = xf, y = yf, z = zf, x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, yz = y * z, xz = x * z;
        const double t[kNumSums] = { x2 * x2, x2 * y2, x2 * z2, x2 * xy, x2 * yz, x2 * xz, x2 * x, x2 * y, x2 * z, x2, y2 * y2, y2 * z2,
          y2 * xy, y2 * yz, y2 * xz, y2 * x, y2 * y, y2 * z, y2, z2 * z2, z2 * xy, z2 * yz, z2 * xz, z2 * x, z2 * y, z2 * z, z2, x * yz,
          xy, yz, xz, x, y, z, x2 + y2, y2 + z2, x2 + z2 };
        for (int k = 0; k < kNumSums; k++)
          acc[k] += t[k];
        n++;
      }
    for (int k = 0; k < kNumSums; k++)
      sums[(size_t) s * kNumSums + k] = acc[k];
    ns[s] = n;
  }
  double *d_s, *d_n, *d_o;
  hipMalloc(&d_s, sums.size() * 8);
  hipMalloc(&d_n, ns.size() * 8);
  hipMalloc(&d_o, (size_t) S * 12 * 8);
  hipMemcpy(d_s, sums.data(), sums.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_n, ns.data(), ns.size() * 8, hipMemcpyHostToDevice);
  for (int w : { 1, 32, 256, 1024, 2048 })
  {
    const float t[8] = { run<0>(w, d_s, d_n, d_o), run<1>(w, d_s, d_n, d_o), run<2>(w, d_s, d_n, d_o), run<3>(w, d_s, d_n, d_o),
      run<4>(w, d_s, d_n, d_o), run<5>(w, d_s, d_n, d_o), run<6>(w, d_s, d_n, d_o), run<7>(w, d_s, d_n, d_o) };
    printf("%4d waves: total %.1f us | launch+setup %.1f chol %.1f reduce %.1f tridiag %.1f bisect %.1f twisted %.1f backtf %.1f tail %.1f\n", w,
      t[7], t[0], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6]);
  }
  run<7>(1, d_s, d_n, d_o);
  std::vector<double> out((size_t) 12 * 4);
  hipMemcpy(out.data(), d_o, out.size() * 8, hipMemcpyDeviceToHost);
  printf("sample 0: lambda %.6e v = %.6f %.6f %.6f ... %.6f\n", out[10], out[0], out[1], out[2], out[9]);
  return 0;
}
