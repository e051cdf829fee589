// which way do the GFX9 wave-wide DPP shifts move data, and what does a lane get from a disabled / missing source?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    int v = 100 + lane;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    out[lane] = a;
    out[64 + lane] = b;
    if (lane % 3 != 0) {  // divergent: sources with lane % 3 == 0 are disabled
        int c = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);
        int d = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
        out[128 + lane] = c;
        out[192 + lane] = d;
    }
}
int main() {
    int* d;
    hipMalloc(&d, 256 * sizeof(int));
    hipMemset(d, 0, 256 * sizeof(int));
    probe<<<1, 64>>>(d);
    int h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) {
        printf("row %d:", r);
        for (int l = 0; l < 64; ++l) if (l < 6 || l > 60) printf(" [%d]=%d", l, h[r * 64 + l]);
        printf("\n");
    }
    return 0;
}
