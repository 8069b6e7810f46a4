// Probe: does a 16-byte LDS-DMA piece (`buffer_load_dwordx4 ... lds`, the GEMM's operand loader) accept a global address that is only
// 4-byte aligned?  The im2col-free patch-embedding loader reads 16-pixel runs of NCHW frames at byte offset 28 * px (patch = 14 bf16
// pixels), i.e. 4-byte-aligned 16-byte pieces.  Source buffer: u16 element e holds value e.  Every lane fetches 16 bytes starting at
// element (lane * 14 + shift) for shift = 0, 2, 4, 6 (byte alignments 0/4/8/12 mod 16 across lanes) and the LDS image is copied out.
// Prints "OK" per shift when every lane received elements [start, start + 8).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_dma_align_probe.hip -o tools/micro/lds_dma_align_probe && tools/micro/lds_dma_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16;

__global__ void probe(const u16* src, u16* out, int shift, int nbytes) {
    __shared__ __attribute__((aligned(16))) u16 lds[64 * 8];
    const int l = threadIdx.x;
    __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const unsigned voff = (unsigned)(l * 14 + shift) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 8; ++i) out[l * 8 + i] = lds[l * 8 + i];
}

int main() {
    const int N = 4096;
    u16 h[N]; for (int i = 0; i < N; ++i) h[i] = (u16)i;
    u16 *s, *d; hipMalloc(&s, N * 2); hipMalloc(&d, 64 * 8 * 2);
    hipMemcpy(s, h, N * 2, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int shift = 0; shift < 8; shift += 2) {
        hipMemset(d, 0xff, 64 * 8 * 2);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, d, shift, N * 2);
        u16 o[512];
        hipError_t e = hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 8; ++i) if (o[l * 8 + i] != (u16)(l * 14 + shift + i)) ++bad;
        printf("shift %d (lane byte offsets %d + 28 l): %s (%d wrong elements, hip %d); lane 1 got %d %d %d %d %d %d %d %d (want %d..)\n", shift, shift * 2,
               bad ? "MISMATCH" : "OK", bad, (int)e, o[8], o[9], o[10], o[11], o[12], o[13], o[14], o[15], 14 + shift);
        bad_total += bad;
    }
    printf(bad_total ? "RESULT: unaligned 16-byte LDS-DMA pieces are NOT delivered as addressed\n" : "RESULT: 4-byte-aligned 16-byte LDS-DMA pieces work\n");
    return 0;
}
