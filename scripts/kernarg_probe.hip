// kernarg_probe.hip - development aid (round 5): how a NON-INLINED device function can read the kernel's arguments with scalar loads.
// -DUSE_KERNARG: the callee reads them through a pointer to the kernel-argument segment that the KERNEL passes down (works);
// asking for that pointer inside the callee (__builtin_amdgcn_kernarg_segment_ptr() there) faults on this toolchain (ROCm 7.2, gfx950).
// -DUSE_BID / -DUSE_LDS: blockIdx and dynamic LDS inside the callee (both work).  genomicsbench_amd/csrc/poa_kernels.hip: poa_serial_call.
//   hipcc -O3 --offload-arch=gfx950 -DUSE_KERNARG -DUSE_BID -DUSE_LDS scripts/kernarg_probe.hip -o build_tmp/kernarg_probe && build_tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct Inner { long long a; int *out; int k[5]; };
struct KArgs { Inner A; long long layout[4]; };
struct St { int x, y, z; };
template <int W>
__device__ __attribute__((noinline)) St callee(St st, long long w, int flags, int *outp, unsigned long long kp)
{
    st.x = __builtin_amdgcn_readfirstlane(st.x); st.y = __builtin_amdgcn_readfirstlane(st.y); st.z = __builtin_amdgcn_readfirstlane(st.z);
    int v = 0;
#ifdef USE_KERNARG
    typedef const __attribute__((address_space(4))) int kw_t;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kp), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(kp >> 32));
    kw_t *kw = (kw_t *)(((unsigned long long)hi << 32) | lo);
    KArgs K;
    { int words[sizeof(KArgs) / 4];
#pragma unroll
      for (unsigned k = 0; k < sizeof(KArgs) / 4; ++k) words[k] = kw[k];
      __builtin_memcpy(&K, words, sizeof(KArgs)); }
    v += (int)K.A.a + K.A.k[3] + (int)K.layout[2];
    outp = K.A.out;
#endif
#ifdef USE_LDS
    extern __shared__ int lds[];
    if ((threadIdx.x & 63) == 0) { lds[0] = 5; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    v += lds[0];
#endif
    int b = 0;
#ifdef USE_BID
    b = blockIdx.x;
#endif
    if (threadIdx.x == 0) outp[b * 4 + 0] = v + flags + (int)w + st.x;
    St r = {st.x + 1, st.y + 2, b};
    return r;
}
template <int W>
__global__ void __launch_bounds__(64, W) kern(KArgs K)
{
    St st = {1, 2, 3};
    for (int i = 0; i < 3; ++i) { st = callee<W>(st, 100 + i, i, K.A.out, (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr()); st.x = __builtin_amdgcn_readfirstlane(st.x); }
    if (threadIdx.x == 0) { K.A.out[blockIdx.x * 4 + 1] = st.x; K.A.out[blockIdx.x * 4 + 2] = st.y; K.A.out[blockIdx.x * 4 + 3] = st.z; }
}
int main()
{
    int *d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0, 64);
    KArgs K; K.A.a = 7; K.A.out = d; for (int i = 0; i < 5; ++i) K.A.k[i] = 10 * i; for (int i = 0; i < 4; ++i) K.layout[i] = 1000 * i;
    hipLaunchKernelGGL(kern<3>, dim3(2), dim3(64), 256, 0, K);
    hipError_t e = hipDeviceSynchronize();
    int h[8]; (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("err %d: %d %d %d %d | %d %d %d %d\n", (int)e, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
