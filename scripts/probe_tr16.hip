// What ds_read_b64_tr_b16 returns (gfx950): every lane supplies the address of 4 contiguous 16-bit values; the 16 lanes of a group
// exchange them.  Prints, for two address patterns, which LDS element index each (lane, element) received, and checks the rule the
// attention kernel relies on: lane c of a group, element j  <-  element (c % 4) of the run addressed by lane 4 j + c / 4 of the group.
//   hipcc -O2 --offload-arch=gfx950 scripts/probe_tr16.hip -o build/probe_tr16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 h4 __attribute__((__vector_size__(8)));
__global__ void k(unsigned short* out, const int* addr_elems)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short) i;
    __syncthreads();
    const int l = threadIdx.x;
    for (int pat = 0; pat < 2; ++pat) {
        const int e0 = addr_elems[pat * 64 + l];
        __attribute__((address_space(3))) h4* p = (__attribute__((address_space(3))) h4*) (lds + e0);
        h4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
        unsigned short v[4];
        __builtin_memcpy(v, &a, 8);
        for (int j = 0; j < 4; ++j) out[(pat * 64 + l) * 4 + j] = v[j];
    }
}
int main()
{
    int h_addr[128];
    for (int l = 0; l < 64; ++l) h_addr[l] = l * 4;                                              // lane-linear
    for (int l = 0; l < 64; ++l) h_addr[64 + l] = (l >> 4) * 1024 + ((l & 15) >> 2) * 128 + (l & 3) * 4 + 8;   // rows 256 B apart, runs inside a row
    int* d_addr; unsigned short* d_out; unsigned short h_out[512];
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_out, d_addr);
    if (hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
    int bad = 0;
    for (int pat = 0; pat < 2; ++pat)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int grp = l & ~15, c = l & 15;
                const int want = h_addr[pat * 64 + grp + 4 * j + c / 4] + (c % 4);
                if (h_out[(pat * 64 + l) * 4 + j] != want) ++bad;
            }
    for (int pat = 0; pat < 2; ++pat) {
        printf("pattern %d\n", pat);
        for (int l = 0; l < 20; ++l) printf("  lane %2d (addr elem %4d): %4d %4d %4d %4d\n", l, h_addr[pat * 64 + l], h_out[(pat * 64 + l) * 4], h_out[(pat * 64 + l) * 4 + 1],
                                            h_out[(pat * 64 + l) * 4 + 2], h_out[(pat * 64 + l) * 4 + 3]);
    }
    printf("rule 'lane c elem j <- elem c%%4 of lane 4j + c/4': %s (%d mismatches)\n", bad ? "WRONG" : "holds", bad);
    return bad ? 2 : 0;
}
