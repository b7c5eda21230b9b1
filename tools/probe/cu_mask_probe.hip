// cu_mask_probe.hip -- which XCD (and shader engine / CU) the workgroups of a stream created with hipExtStreamCreateWithCUMask land
// on, for masks that set every 8th bit (bit i -> XCD i % 8 is what the driver's symmetric mapping suggests) and for a contiguous run.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/cu_mask_probe.hip -o tools/probe/cu_mask_probe && tools/probe/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <map>
#define CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

__global__ void k_where(unsigned* out, int spin)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;           // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | 4);                  // HW_REG_HW_ID: [11:8] CU, [12] SH, [15:13] SE
    volatile int sink = 0;
    for (int i = 0; i < spin; i++) sink += i;                                        // keep the block resident so that others spread out
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xFFFFu);
}

static int run(const char* name, const std::vector<uint32_t>& mask, unsigned* d_out, int nblocks)
{
    hipStream_t s;
    if (mask.empty()) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else CHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    hipLaunchKernelGGL(k_where, dim3(nblocks), dim3(64), 0, s, d_out, 20000);
    CHK(hipStreamSynchronize(s));
    std::vector<unsigned> h(nblocks);
    CHK(hipMemcpy(h.data(), d_out, sizeof(unsigned) * nblocks, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_xcc; std::map<unsigned, int> cus;
    for (unsigned v : h) { per_xcc[v >> 16]++; cus[((v >> 16) << 16) | ((v >> 8) & 0xFF)]++; }
    std::printf("%-28s blocks per XCD:", name);
    for (auto& kv : per_xcc) std::printf(" %u:%d", kv.first, kv.second);
    std::printf("   distinct (XCD, SE/SH/CU): %zu\n", cus.size());
    CHK(hipStreamDestroy(s));
    return 0;
}

int main()
{
    unsigned* d_out = nullptr;
    const int nblocks = 4096;
    CHK(hipMalloc(&d_out, sizeof(unsigned) * nblocks));
    if (run("no mask", {}, d_out, nblocks)) return 1;
    for (int x = 0; x < 8; x++) {
        std::vector<uint32_t> m(8, 0);
        for (int b = x; b < 256; b += 8) m[b / 32] |= 1u << (b % 32);
        char name[64]; std::snprintf(name, sizeof name, "bits %d, %d, %d, ...", x, x + 8, x + 16);
        if (run(name, m, d_out, nblocks)) return 1;
    }
    for (int x = 0; x < 2; x++) {
        std::vector<uint32_t> m(8, 0);
        m[x] = 0xFFFFFFFFu;
        char name[64]; std::snprintf(name, sizeof name, "bits %d .. %d", 32 * x, 32 * x + 31);
        if (run(name, m, d_out, nblocks)) return 1;
    }
    { std::vector<uint32_t> m(8, 0); for (int b = 0; b < 256; b++) if ((b % 8) < 4) m[b / 32] |= 1u << (b % 32); if (run("bits with (i % 8) < 4", m, d_out, nblocks)) return 1; }
    return 0;
}
