// Hardware probe (hipcc --offload-arch=gfx950 -O3 glds_probe.hip; result recorded in DESIGN.md, section 4): does `buffer_load_dwordx4 ... lds` write zeros for out-of-range lanes, and how do EXEC-masked lanes behave?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, float* o, unsigned nbytes) {
  __shared__ __attribute__((aligned(1024))) float lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, (int)nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  if (threadIdx.x < 2)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 1024), 16, threadIdx.x * 16 + 64, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) o[i] = lds[i];
}
int main() {
  float *p, *o, h[2048], hp[1024];
  for (int i = 0; i < 1024; ++i) hp[i] = (float)(i + 1);
  hipMalloc(&p, 4096); hipMalloc(&o, 8192);
  hipMemcpy(p, hp, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(p, o, 4096);
  hipMemcpy(h, o, 8192, hipMemcpyDeviceToHost);
  printf("main:"); for (int i = 0; i < 24; ++i) printf(" %g", h[i]); printf("\n");
  printf("masked:"); for (int i = 1024; i < 1024 + 16; ++i) printf(" %g", h[i]); printf("\n");
  return 0;
}
