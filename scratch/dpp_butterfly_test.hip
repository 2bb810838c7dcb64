#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
__device__ __forceinline__ float xor_add_32(float v) {
    // v_permlane32_swap: swaps the upper 32 lanes of the first operand with the lower 32 lanes of the second
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor_add_16(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ float dpp(float old, float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(v), CTRL, 0xF, BANK, false));
}
__device__ __forceinline__ float xor_add_8(float v) { return v + dpp<0x128>(v, v); }   // row_ror:8
__device__ __forceinline__ float xor_add_4(float v) {
    float t = dpp<0x104, 0x5>(v, v);   // row_shl:4 -> lane i reads lane i+4, banks 0 and 2
    t = dpp<0x114, 0xA>(t, v);         // row_shr:4 -> lane i reads lane i-4, banks 1 and 3
    return v + t;
}
__device__ __forceinline__ float xor_add_2(float v) { return v + dpp<0x4E>(v, v); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ float xor_add_1(float v) { return v + dpp<0xB1>(v, v); }    // quad_perm [1,0,3,2]
__global__ void k(const float *in, float *out, float *ref) {
    float v = in[threadIdx.x];
    float a = xor_add_1(xor_add_2(xor_add_4(xor_add_8(xor_add_16(xor_add_32(v))))));
    out[threadIdx.x] = a;
    float b = v;
    for (int off = 32; off >= 1; off >>= 1) b = b + __shfl_xor(b, off, 64);
    ref[threadIdx.x] = b;
    // per-step check values
    out[64 + threadIdx.x] = xor_add_32(v);  ref[64 + threadIdx.x] = v + __shfl_xor(v, 32, 64);
    out[128 + threadIdx.x] = xor_add_16(v); ref[128 + threadIdx.x] = v + __shfl_xor(v, 16, 64);
    out[192 + threadIdx.x] = xor_add_8(v);  ref[192 + threadIdx.x] = v + __shfl_xor(v, 8, 64);
    out[256 + threadIdx.x] = xor_add_4(v);  ref[256 + threadIdx.x] = v + __shfl_xor(v, 4, 64);
    out[320 + threadIdx.x] = xor_add_2(v);  ref[320 + threadIdx.x] = v + __shfl_xor(v, 2, 64);
    out[384 + threadIdx.x] = xor_add_1(v);  ref[384 + threadIdx.x] = v + __shfl_xor(v, 1, 64);
}
int main() {
    float h[64], *din, *dout, *dref, o[448], r[448];
    for (int i = 0; i < 64; i++) h[i] = 1.0f / (i + 3) * (i % 3 ? 1.f : -7.3f);
    hipMalloc(&din, 256); hipMalloc(&dout, 448 * 4); hipMalloc(&dref, 448 * 4);
    hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, dref);
    hipMemcpy(o, dout, 448 * 4, hipMemcpyDeviceToHost); hipMemcpy(r, dref, 448 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 448; i++) if (memcmp(&o[i], &r[i], 4)) { bad++; if (bad < 10) printf("mismatch %d: %a vs %a\n", i, o[i], r[i]); }
    printf("dpp butterfly: %d mismatches of 448\n", bad);
    return bad != 0;
}
