import torch, subprocess, os, ctypes
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 1000;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
extern "C" void run(unsigned* o) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); hipDeviceSynchronize(); }
'''
open('/tmp/pl.hip','w').write(src)
subprocess.check_call(['hipcc','--offload-arch=gfx950','-O3','-shared','-fPIC','-o','/tmp/pl.so','/tmp/pl.hip'])
lib = ctypes.CDLL('/tmp/pl.so')
o = torch.zeros(128, dtype=torch.int32, device='cuda')
lib.run(ctypes.c_void_p(o.data_ptr()))
print('r0', o[:64].tolist()); print('r1', o[64:].tolist())
