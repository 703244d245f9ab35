// Diagnostic (not product): the operand / result lane layout of v_mfma_f64_4x4x4_f64, found by one-hot probing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long* out) {      // out[la * 64 + lb] = mask of result lanes that see a[la] * b[lb]
  const int lane = threadIdx.x, la = blockIdx.x >> 6, lb = blockIdx.x & 63;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  const unsigned long long m = __ballot(d != 0.0);
  if (lane == 0) out[blockIdx.x] = m;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 4096 * 8);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d);
  std::vector<unsigned long long> h(4096);
  hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("a lane %2d pairs with:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb]) {
        printf(" b%d->d", lb);
        for (int l = 0; l < 64; ++l) if (h[la * 64 + lb] >> l & 1) printf("%d,", l);
      }
    printf("\n");
  }
  return 0;
}
