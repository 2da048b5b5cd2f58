#include <hip/hip_runtime.h>
#include <cstdio>
template <int MASK>
__device__ __forceinline__ void lane_pair(float x, float& a, float& b) {
    const unsigned ux = __builtin_bit_cast(unsigned, x);
    if constexpr (MASK == 8) {
        a = x;
        b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, (int)ux, 0x128, 0xF, 0xF, true));
    } else if constexpr (MASK == 16) {
        // inline asm: with both inputs equal, hipcc (ROCm 7.2) folds the builtin's two results into one register
        unsigned ua = ux, ub = ux;
        asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ua), "+v"(ub));
        a = __builtin_bit_cast(float, ua); b = __builtin_bit_cast(float, ub);
    } else {
        unsigned ua = ux, ub = ux;
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ua), "+v"(ub));
        a = __builtin_bit_cast(float, ua); b = __builtin_bit_cast(float, ub);
    }
}
__global__ void k(float* o) {
    float x = (float)threadIdx.x;
    float a, b;
    lane_pair<8>(x, a, b);  o[threadIdx.x] = a; o[64 + threadIdx.x] = b;
    lane_pair<16>(x, a, b); o[128 + threadIdx.x] = a; o[192 + threadIdx.x] = b;
    lane_pair<32>(x, a, b); o[256 + threadIdx.x] = a; o[320 + threadIdx.x] = b;
}
int main() {
    float* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"8a", "8b", "16a", "16b", "32a", "32b"};
    for (int r = 0; r < 6; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %d", (int)h[r * 64 + i]); printf("\n"); }
    return 0;
}
