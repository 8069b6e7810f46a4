// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): which LDS elements does lane l receive, given per-lane addresses?
// LDS holds element index e at bf16 slot e (value = e).  Case A: lane i supplies address of elements [4*(i&15) + 64*(i>>4) .. +3]
// (the canonical contiguous 4x16 block per 16-lane group).  Case B: row-major rows with a 144-byte stride: lane i supplies
// &row[(i&15)>>2][4*(i&3)] of block (i>>4).  Prints the 4 values every lane received.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read_probe.hip -o tools/micro/tr_read_probe && tools/micro/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(u16* out, int mode) {
    __shared__ __attribute__((aligned(16))) u16 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (u16)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = (unsigned)(size_t)lds + (4 * (l & 15) + 64 * (l >> 4)) * 2;
    else addr = (unsigned)(size_t)lds + (((l & 15) >> 2) * 72 + 4 * (l & 3) + (l >> 4) * 16) * 2;     // 4 rows of a [*][72] image, 16-col block (l>>4)
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (u16)(v[0] & 0xffff); out[l * 4 + 1] = (u16)(v[0] >> 16);
    out[l * 4 + 2] = (u16)(v[1] & 0xffff); out[l * 4 + 3] = (u16)(v[1] >> 16);
}

int main() {
    u16* d; hipMalloc(&d, 64 * 4 * 2);
    u16 h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
