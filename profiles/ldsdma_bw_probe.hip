// Hardware probe: the rate at which every CU can pull operand tiles global -> LDS with `buffer_load_dwordx4 ... lds` when all
// 256 CUs stream at once, (a) from an L2-resident window shared by the workgroups of an XCD (what the weight tiles and the
// re-read activation rows of the implicit-GEMM kernels are) and (b) from a 2 GiB buffer (HBM).  One 512-thread workgroup
// per CU (150 KB of LDS keeps a second one off), every wave keeps DEPTH 1-KB requests in flight into a private ring.
//   hipcc --offload-arch=gfx950 -O3 ldsdma_bw_probe.hip -o ldsdma_bw_probe && ./ldsdma_bw_probe
// Result of round 2 is recorded in DESIGN.md section 6.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int DEPTH = 8;
__global__ __launch_bounds__(512) void probe(const char* src, unsigned long long window_bytes, unsigned long long span_per_wg,
                                             int iters, unsigned long long* cycles) {
  __shared__ __attribute__((aligned(1024))) char lds[150 * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // this workgroup's region: window_bytes == 0 -> private span of the big buffer; else a window shared by its XCD
  const unsigned long long base = window_bytes ? (unsigned long long)(blockIdx.x & 7) * window_bytes
                                               : (unsigned long long)blockIdx.x * span_per_wg;
  const unsigned long long limit = window_bytes ? window_bytes : span_per_wg;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + base), (short)0, (int)limit, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* ring = lds + wave * (DEPTH * 1024);
  unsigned off = (unsigned)(wave * 1024 + lane * 16);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(ring + d * 1024), 16, off, 0, 0, 0);
      off += 8 * 1024;
      if (off >= (unsigned)limit) off -= (unsigned)limit;
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const unsigned long long big = 2ull << 30;
  char* src;
  unsigned long long* cyc;
  hipMalloc(&src, big);
  hipMemset(src, 1, big);
  hipMalloc(&cyc, 256 * 8);
  const int iters = 400;
  const double bytes_per_wg = (double)iters * DEPTH * 8 * 1024;
  struct { const char* name; unsigned long long window; } cases[] = {
      {"L2-resident window (512 KB per XCD)", 512ull << 10}, {"L2-resident window (2 MB per XCD)", 2ull << 20}, {"HBM stream (8 MB per workgroup)", 0}};
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      probe<<<256, 512>>>(src, c.window, 8ull << 20, iters, cyc);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      unsigned long long h[256];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double sum = 0;
      for (int i = 0; i < 256; ++i) sum += (double)h[i];
      if (rep == 1)
        printf("%-40s %7.3f ms  %6.2f TB/s chip-wide  %5.1f B/clk/CU (in-kernel cycles)\n", c.name, ms,
               256 * bytes_per_wg / (ms * 1e-3) / 1e12, bytes_per_wg / (sum / 256));
    }
  }
  return 0;
}
