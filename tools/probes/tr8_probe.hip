// Layout probe for ds_read_b64_tr_b8 (gfx950): LDS holds byte value (row * 16 + col) of a [16 rows][16 cols] block (row stride 16 B);
// every lane passes an address, the 8 returned bytes are dumped.  Two address patterns are tried.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) int lds_i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned char* out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * 16];
    for (int i = threadIdx.x; i < 64 * 16; i += 64) img[i] = (unsigned char)i;       // value = row*16 + col (rows 0..63 wrap at 256)
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i16 = lane & 15;
    int row, col;
    if (pattern == 0) { row = 8 * g + (i16 >> 1); col = 8 * (i16 & 1); }          // guess: lane 2q+p -> row q, cols 8p..8p+7
    else { row = 8 * g + (i16 & 7); col = 8 * (i16 >> 3); }                          // alternative: lane 8p+q
    const unsigned addr = (unsigned)(row * 16 + col);
    i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(img + addr));
    reinterpret_cast<i32x2*>(out)[lane] = v;
}
int main() {
    unsigned char* d; hipMalloc(&d, 64 * 8);
    unsigned char h[64 * 8];
    for (int pat = 0; pat < 2; ++pat) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pat);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 20; ++l) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) printf(" (%d,%2d)", h[l * 8 + b] >> 4, h[l * 8 + b] & 15);
            printf("\n");
        }
    }
    return 0;
}
