// Probe (gfx950): does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
// One wave; source buffer word i holds i; M0 = 4096; offset:256.  Prints the LDS dword index where source word 0x40 (byte 256) landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(const uint32_t* src, uint32_t* out) {
    __shared__ uint32_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(void __attribute__((address_space(3)))*)lds + 4096;
    const uint32_t voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:256\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(src), "s"(__builtin_amdgcn_readfirstlane(base)) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<uint32_t> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;
    uint32_t *src, *out;
    hipMalloc(&src, 8192 * 4); hipMalloc(&out, 8192 * 4);
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, out);
    std::vector<uint32_t> o(8192);
    hipMemcpy(o.data(), out, 8192 * 4, hipMemcpyDeviceToHost);
    int first = -1, n = 0;
    for (int i = 0; i < 8192; ++i) if (o[i] != 0xFFFFFFFFu) { if (first < 0) first = i; ++n; }
    printf("glds_offset: %d dwords written, first at LDS dword %d (byte %d), value 0x%x (source byte %u); M0 pointed at byte 4096\n",
           n, first, first * 4, first >= 0 ? o[first] : 0, first >= 0 ? o[first] * 4 : 0);
    printf("  -> offset:256 %s the LDS destination and %s the global source\n", first * 4 == 4096 ? "does NOT move" : (first * 4 == 4096 + 256 ? "MOVES" : "??"),
           (first >= 0 && o[first] == 64) ? "moves" : "does not move");
    return 0;
}
