// Does gfx950 execute scalar atomics (s_atomic_add ... glc)?  Every workgroup draws one ticket per round from one counter; the tickets
// of a launch must be a permutation of 0 .. blocks*rounds-1.   hipcc --offload-arch=gfx950 satomic_test.hip -o satomic_test && ./satomic_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void draw(int* counter, int* out, int rounds) {
  for (int r = 0; r < rounds; ++r) {
    int t = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(counter) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x * rounds + r] = t;
  }
}
int main() {
  const int blocks = 1024, rounds = 8;
  int *c, *o;
  hipMalloc(&c, 4); hipMalloc(&o, blocks * rounds * 4);
  hipMemset(c, 0, 4);
  hipLaunchKernelGGL(draw, dim3(blocks), dim3(512), 0, 0, c, o, rounds);
  std::vector<int> h(blocks * rounds);
  if (hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: memcpy\n"); return 1; }
  int final_c; hipMemcpy(&final_c, c, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = final_c == blocks * rounds;
  for (int i = 0; i < (int)h.size(); ++i) ok = ok && h[i] == i;
  printf("%s: counter %d (want %d), tickets %d..%d\n", ok ? "OK permutation" : "FAIL", final_c, blocks * rounds, h.front(), h.back());
  return ok ? 0 : 1;
}
