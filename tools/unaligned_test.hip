// Diagnostic: do dword-aligned (not 16-B aligned) buffer_load_dwordx4 ... lds, buffer_load_dwordx2 ... lds
// and buffer_store_dwordx4 / global dwordx4 stores behave on gfx950?  (The padded-activation conv kernel
// gathers 4 consecutive positions whose byte address is only 4-B aligned.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void k(const float* x, float* y, float* z, int shift, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds), 16, (lane * 4 + shift) * 4, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 4, (lane + shift) * 4, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 320; i += 64) y[i] = lds[i];
    // misaligned vector stores: z[shift + 4*lane .. +3] = lane*4 + {0,1,2,3}
    v4f v = {lane * 4.f, lane * 4.f + 1, lane * 4.f + 2, lane * 4.f + 3};
    __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(z, 0, n * 4, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rz, (lane * 4 + shift) * 4, 0, 0);
    // plain pointer store of a 4-B aligned float4
    float* zp = z + 512 + shift + 4 * lane;
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(zp));
}

int main() {
    const int n = 2048;
    std::vector<float> hx(n), hy(320), hz(n);
    for (int i = 0; i < n; ++i) hx[i] = (float)i;
    float *x, *y, *z;
    hipMalloc(&x, n * 4); hipMalloc(&y, 320 * 4); hipMalloc(&z, n * 4);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(z, 0, n * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, x, y, z, shift, n);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("shift %d: %s\n", shift, hipGetErrorString(e)); return 1; }
        hipMemcpy(hy.data(), y, 320 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hz.data(), z, n * 4, hipMemcpyDeviceToHost);
        int b1 = 0, b2 = 0, b3 = 0, b4 = 0;
        for (int i = 0; i < 256; ++i) b1 += hy[i] != (float)(i + shift);
        for (int i = 0; i < 64; ++i) b2 += hy[256 + i] != (float)(i + shift);
        for (int i = 0; i < 256; ++i) b3 += hz[shift + i] != (float)i;
        for (int i = 0; i < 256; ++i) b4 += hz[512 + shift + i] != (float)i;
        printf("shift %d: dma_x4 bad=%d dma_x1 bad=%d buffer_store_x4 bad=%d ptr_store_x4 bad=%d\n", shift, b1, b2, b3, b4);
        bad += b1 + b2 + b3 + b4;
    }
    printf(bad ? "FAIL\n" : "OK\n");
    return bad != 0;
}
