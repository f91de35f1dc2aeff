#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const unsigned char* p, uint32_t* out, int n) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  unsigned off = 5u * (unsigned)(e >> 1) + 2u * (unsigned)(e & 1);
  uint32_t w;
  asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(off), "s"(p) : "memory");
  out[e] = (w >> (4 * (e & 1))) & 0xFFFFF;
}
int main() {
  const int n = 1 << 20;
  std::vector<unsigned char> h(5 * (n / 2) + 8, 0);
  std::vector<uint32_t> ref(n);
  for (int e = 0; e < n; ++e) {
    uint32_t v = (uint32_t)(e * 2654435761u) & 0xFFFFF; ref[e] = v;
    size_t bit = (size_t)20 * e;
    for (int b = 0; b < 20; ++b) if (v >> b & 1) h[(bit + b) >> 3] |= 1 << ((bit + b) & 7);
  }
  unsigned char* d; uint32_t* o;
  hipMalloc(&d, h.size()); hipMalloc(&o, 4 * n);
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
  std::vector<uint32_t> g(n);
  hipMemcpy(g.data(), o, 4 * n, hipMemcpyDeviceToHost);
  int bad = 0; for (int e = 0; e < n; ++e) bad += g[e] != ref[e];
  printf("mismatches: %d of %d (err %s)\n", bad, n, hipGetErrorString(hipGetLastError()));
}
