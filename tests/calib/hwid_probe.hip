// Developer probe (GPU box): which hardware CU does a workgroup run on?  HW_REG_HW_ID (4) and HW_REG_XCC_ID (20) per block of a
// 1024 x 256-thread launch with 27 KB of LDS per block (the frame kernel's footprint: four blocks per CU), all blocks resident.
// usage: hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid_probe tests/calib/hwid_probe.hip && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) probe(unsigned *out, unsigned *arrived, unsigned n)
{
    __shared__ float pad[26800 / 4];
    pad[threadIdx.x] = (float) threadIdx.x;
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        atomicAdd(arrived, 1u);
        unsigned long long t = wall_clock64() + 20000000ull;            // everybody resident, or 0.2 s
        while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n && wall_clock64() < t) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    if (pad[threadIdx.x] < 0) out[0] = 0;
}
int main()
{
    const unsigned n = 1024;
    unsigned *d, *a;
    hipMalloc(&d, n * 8); hipMalloc(&a, 4); hipMemset(a, 0, 4);
    probe<<<n, 256>>>(d, a, n);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<unsigned>> byk;
    for (unsigned b = 0; b < n; b++) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        byk[(xcc << 8) | (se << 5) | (sh << 4) | cu].push_back(b);
    }
    std::map<size_t, unsigned> hist;
    for (auto &kv : byk) hist[kv.second.size()]++;
    printf("%zu distinct (xcc, se, sh, cu) keys for %u blocks; blocks per key:", byk.size(), n);
    for (auto &kv : hist) printf(" %zu x%u", kv.first, kv.second);
    printf("\nfirst keys:");
    int k = 0;
    for (auto &kv : byk) { if (k++ >= 6) break; printf(" [%03x:", kv.first); for (unsigned b : kv.second) printf(" %u", b); printf("]"); }
    printf("\nblock 0..7: ");
    for (unsigned b = 0; b < 8; b++) printf("hw %08x xcc %x  ", h[2 * b], h[2 * b + 1]);
    printf("\n");
    return 0;
}
