// What a grid-wide barrier costs inside ONE kernel on the MI355X (8 XCDs, one L2 each): an arrival counter and a
// generation word in device memory, every work-group resident.  With the fences that make the other groups' plain stores
// visible (release before arriving, acquire after leaving: agent scope = L2 write-back / invalidate across the XCDs), and
// without them (the raw latency of the atomics).  This is what decides whether stages of the search can be fused across
// a launch boundary: a launch boundary costs ~5 us.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/grid_barrier scripts/micro/grid_barrier.hip && scripts/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

struct Sync
{
  unsigned arrive, gen;
};

template <bool FENCES, bool TOUCH>
__global__ __launch_bounds__(256) void k(Sync* gs, int rounds, float* data, int n_data)
{
  for (int r = 0; r < rounds; r++)
  {
    if (TOUCH)  // every group dirties some lines the others will read after the barrier (what a real stage does)
      for (int i = blockIdx.x * 256 + threadIdx.x; i < n_data; i += gridDim.x * 256)
        data[i] += 1.0f;
    __syncthreads();
    if (threadIdx.x == 0)
    {
      const unsigned g = __hip_atomic_load(&gs->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (FENCES)
        __threadfence();
      if (__hip_atomic_fetch_add(&gs->arrive, 1u, FENCES ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
      {
        __hip_atomic_store(&gs->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&gs->gen, 1u, FENCES ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      else
        while (__hip_atomic_load(&gs->gen, FENCES ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g)
          __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    if (FENCES)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

template <bool FENCES, bool TOUCH>
static void run(const char* name, int groups, Sync* gs, float* data, int n_data)
{
  const int rounds = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++)
  {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FENCES, TOUCH>), dim3(groups), dim3(256), 0, 0, gs, rounds, data, n_data);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-44s %4d groups: %7.2f us per barrier\n", name, groups, best * 1000.0f / rounds);
}

int main()
{
  Sync* gs;
  float* data;
  const int n_data = 1 << 20;  // 4 MB
  hipMalloc(&gs, 64);
  hipMemset(gs, 0, 64);
  hipMalloc(&data, n_data * 4);
  hipMemset(data, 0, n_data * 4);
  for (int groups : { 32, 256, 1024 })
  {
    run<false, false>("atomics only (relaxed)", groups, gs, data, n_data);
    run<true, false>("release / acquire fences, nothing dirty", groups, gs, data, n_data);
    run<true, true>("release / acquire fences, 4 MB dirtied per round", groups, gs, data, n_data);
    run<false, true>("no fences, 4 MB touched per round (work only)", groups, gs, data, n_data);
  }
  return 0;
}
