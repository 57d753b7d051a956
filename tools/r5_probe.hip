// Developer probe (round 5): facts the resident apply's host interface is designed on.  Built by `make -C tools r5_probe`
// (hipcc --offload-arch=gfx950), run on the GPU box; prints plain text.
//  (1) CU masks: which physical CUs (XCC, SE, CU) the workgroups of a kernel launched on a hipExtStreamCreateWithCUMask stream land on
//  (2) does a DMA (hipMemcpyAsync H2D / D2H, linear and 2-D) make progress while a kernel occupies every workgroup slot of the chip?
//  (3) the doorbell: host -> device -> host round trip with the command word (a) in pinned host memory polled over PCIe by the device
//      (round 4's ring) and (b) in device memory written by the host through the PCIe BAR (fine-grained allocation), polled locally
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <chrono>
#include <thread>
#include <immintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void census_kernel(unsigned *out, int spin)
{
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 0xF;      // HW_REG_XCC_ID
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11));            // HW_REG_HW_ID
        out[blockIdx.x] = (xcc << 16) | (hw & 0xFFFFu);
    }
    // stay a while so that the grid spreads over every CU the mask allows instead of re-using the first ones
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

// holds every workgroup slot it is given until *stop != 0 (or 3 s)
__global__ __launch_bounds__(256) void hog_kernel(volatile unsigned *stop, unsigned *up)
{
    if (threadIdx.x == 0) atomicAdd(up, 1u);
    const long long t0 = wall_clock64();
    while (!__hip_atomic_load(const_cast<unsigned *>(stop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
        if (wall_clock64() - t0 > 300000000ll) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

// one wave: waits for cmd == seq (seq = 1, 2, ...), answers by writing seq to *ack (pinned host memory); leaves at seq == last or 2 s
__global__ void echo_kernel(volatile unsigned *cmd, volatile unsigned *ack, unsigned last, int system_scope_poll)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    for (unsigned seq = 1; seq <= last; ++seq) {
        for (;;) {
            const unsigned v = system_scope_poll ? __hip_atomic_load(const_cast<unsigned *>(cmd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                 : __hip_atomic_load(const_cast<unsigned *>(cmd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= seq) break;
            if (wall_clock64() - t0 > 200000000ll) return;
        }
        __hip_atomic_store(const_cast<unsigned *>(ack), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static void census(const char *what, hipStream_t st, int grid)
{
    unsigned *d = nullptr;
    CK(hipMalloc(&d, grid * sizeof(unsigned)));
    CK(hipMemset(d, 0xFF, grid * sizeof(unsigned)));
    hipLaunchKernelGGL(census_kernel, dim3(grid), dim3(256), 0, st, d, 20000);      // 200 us each
    CK(hipStreamSynchronize(st));
    std::vector<unsigned> h(grid);
    CK(hipMemcpy(h.data(), d, grid * sizeof(unsigned), hipMemcpyDeviceToHost));
    CK(hipFree(d));
    std::map<unsigned, int> cus;                      // (xcc, se, sh, cu) -> workgroups
    int per_xcc[16] = {0};
    for (unsigned v : h) {
        const unsigned xcc = v >> 16, hw = v & 0xFFFF, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        per_xcc[xcc & 15]++;
    }
    int cu_per_xcc[16] = {0};
    for (auto &kv : cus) cu_per_xcc[(kv.first >> 12) & 15]++;
    printf("CENSUS %-34s grid %5d: %3zu distinct CUs; CUs per XCC:", what, grid, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", cu_per_xcc[x]);
    printf("; workgroups per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    // the first workgroups' placement: does block b still run on XCC b % 8?
    printf("       first 16 workgroups' XCC:");
    for (int i = 0; i < 16 && i < grid; ++i) printf(" %u", h[i] >> 16);
    printf("\n");
}

int main(int argc, char **argv)
{
    const char *what = argc > 1 ? argv[1] : "all";
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s: %d CUs, isLargeBar %d, canMapHostMemory %d, asyncEngineCount %d\n", prop.name, prop.multiProcessorCount, prop.isLargeBar,
           prop.canMapHostMemory, prop.asyncEngineCount);
    const int ncu = prop.multiProcessorCount;

    if (!strcmp(what, "all") || !strcmp(what, "census")) {
        hipStream_t s0;
        CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
        census("no mask", s0, 2048);
        struct M { const char *name; std::vector<uint32_t> w; };
        std::vector<M> masks;
        { M m{"low 128 bits", std::vector<uint32_t>(8, 0)}; for (int i = 0; i < 4; ++i) m.w[i] = 0xFFFFFFFFu; masks.push_back(m); }
        { M m{"high 128 bits", std::vector<uint32_t>(8, 0)}; for (int i = 4; i < 8; ++i) m.w[i] = 0xFFFFFFFFu; masks.push_back(m); }
        { M m{"all but bits 0..7", std::vector<uint32_t>(8, 0xFFFFFFFFu)}; m.w[0] = 0xFFFFFF00u; masks.push_back(m); }
        { M m{"all but bits 248..255", std::vector<uint32_t>(8, 0xFFFFFFFFu)}; m.w[7] = 0x00FFFFFFu; masks.push_back(m); }
        { M m{"all but bit 0 of every 32", std::vector<uint32_t>(8, 0xFFFFFFFEu)}; masks.push_back(m); }
        { M m{"even bits", std::vector<uint32_t>(8, 0x55555555u)}; masks.push_back(m); }
        { M m{"bits 0..31", std::vector<uint32_t>(8, 0)}; m.w[0] = 0xFFFFFFFFu; masks.push_back(m); }
        { M m{"bits 0..7", std::vector<uint32_t>(8, 0)}; m.w[0] = 0xFFu; masks.push_back(m); }
        { M m{"bits 0,8,16,...,248 (every 8th)", std::vector<uint32_t>(8, 0x01010101u)}; masks.push_back(m); }
        for (auto &m : masks) {
            hipStream_t s;
            hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.w.size(), m.w.data());
            if (e != hipSuccess) { printf("CENSUS %-34s hipExtStreamCreateWithCUMask: %s\n", m.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            census(m.name, s, 2048);
            CK(hipStreamDestroy(s));
        }
        // two masked streams side by side: a hog on the low half, a census on the high half - do they coexist?
        {
            std::vector<uint32_t> lo(8, 0), hi(8, 0);
            for (int i = 0; i < 4; ++i) { lo[i] = 0xFFFFFFFFu; hi[4 + i] = 0xFFFFFFFFu; }
            hipStream_t sl, sh;
            if (hipExtStreamCreateWithCUMask(&sl, 8, lo.data()) == hipSuccess && hipExtStreamCreateWithCUMask(&sh, 8, hi.data()) == hipSuccess) {
                unsigned *stop, *up;
                CK(hipHostMalloc(&stop, 64, hipHostMallocCoherent | hipHostMallocMapped));
                CK(hipMalloc(&up, 4)); CK(hipMemset(up, 0, 4));
                *stop = 0;
                hipLaunchKernelGGL(hog_kernel, dim3(8 * 128), dim3(256), 0, sl, stop, up);
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
                const double t0 = now_us();
                census("high half beside a hog on the low", sh, 1024);
                printf("       ... took %.0f us (a census alone: ~200-400 us; seconds = it waited for the hog)\n", now_us() - t0);
                *stop = 1;
                CK(hipStreamSynchronize(sl));
                unsigned hup = 0; CK(hipMemcpy(&hup, up, 4, hipMemcpyDeviceToHost));
                printf("       hog workgroups that ran: %u\n", hup);
                CK(hipStreamDestroy(sl)); CK(hipStreamDestroy(sh));
            } else { printf("masked stream pair: create failed\n"); (void)hipGetLastError(); }
        }
    }

    if (!strcmp(what, "all") || !strcmp(what, "dma")) {
        // a hog that fills the chip (8 x 256-thread workgroups per CU), then copies on another stream
        unsigned *stop, *up;
        CK(hipHostMalloc(&stop, 64, hipHostMallocCoherent | hipHostMallocMapped));
        CK(hipMalloc(&up, 4));
        const size_t n = 28u << 20;
        uint8_t *hp, *dp, *dq;
        CK(hipHostMalloc(&hp, n, hipHostMallocDefault));
        CK(hipMalloc(&dp, n)); CK(hipMalloc(&dq, n));
        memset(hp, 7, n);
        hipStream_t sk, sc;
        CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
        for (int hog = 0; hog < 2; ++hog) {
            *stop = 0; CK(hipMemset(up, 0, 4));
            if (hog) { hipLaunchKernelGGL(hog_kernel, dim3(8 * ncu), dim3(256), 0, sk, stop, up); std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
            double t0 = now_us();
            CK(hipMemcpyAsync(dp, hp, n, hipMemcpyHostToDevice, sc)); CK(hipStreamSynchronize(sc));
            const double h2d = now_us() - t0;
            t0 = now_us();
            CK(hipMemcpyAsync(hp, dp, n, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc));
            const double d2h = now_us() - t0;
            t0 = now_us();
            CK(hipMemcpy2DAsync(hp, 4096, dp, 3840, 3840, 2160, hipMemcpyDeviceToHost, sc)); CK(hipStreamSynchronize(sc));
            const double d2h2 = now_us() - t0;
            t0 = now_us();
            CK(hipMemcpy2DAsync(dp, 2176, hp, 2200, 2160, 2160, hipMemcpyHostToDevice, sc)); CK(hipStreamSynchronize(sc));
            const double h2d2 = now_us() - t0;
            t0 = now_us();
            CK(hipMemcpyAsync(dq, dp, n, hipMemcpyDeviceToDevice, sc)); CK(hipStreamSynchronize(sc));
            const double d2d = now_us() - t0;
            t0 = now_us();
            CK(hipMemsetAsync(dq, 0, n, sc)); CK(hipStreamSynchronize(sc));
            const double mset = now_us() - t0;
            *stop = 1;
            if (hog) CK(hipStreamSynchronize(sk));
            unsigned hup = 0; CK(hipMemcpy(&hup, up, 4, hipMemcpyDeviceToHost));
            printf("DMA %s (hog workgroups up: %u of %d): 28 MB H2D %.0f us, D2H %.0f us, 2-D D2H (4K frame, pitch 4096) %.0f us, 2-D H2D (plate rows) %.0f us, D2D %.0f us, memset %.0f us\n",
                   hog ? "while a kernel holds every slot" : "on an idle device            ", hup, 8 * ncu, h2d, d2h, d2h2, h2d2, d2d, mset);
        }
    }

    if (!strcmp(what, "all") || !strcmp(what, "doorbell")) {
        const unsigned N = 2000;
        unsigned *ack;
        CK(hipHostMalloc(&ack, 64, hipHostMallocCoherent | hipHostMallocMapped));
        hipStream_t sk;
        CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
        for (int mode = 0; mode < 3; ++mode) {
            unsigned *cmd = nullptr;
            const char *name = mode == 0 ? "command in pinned HOST memory, device polls over PCIe" : mode == 1 ? "command in fine-grained DEVICE memory, host writes through the BAR"
                                                                                                               : "command in plain hipMalloc DEVICE memory, host writes through the BAR";
            if (mode == 0) CK(hipHostMalloc(&cmd, 64, hipHostMallocCoherent | hipHostMallocMapped));
            else if (mode == 1) { if (hipExtMallocWithFlags((void **)&cmd, 4096, hipDeviceMallocFinegrained) != hipSuccess) { printf("DOORBELL %s: allocation failed\n", name); (void)hipGetLastError(); continue; } }
            else CK(hipMalloc(&cmd, 4096));
            if (mode != 0 && !prop.isLargeBar) { printf("DOORBELL %s: skipped (no large BAR)\n", name); continue; }
            if (mode != 0) {
                hipPointerAttribute_t at;
                if (hipPointerGetAttributes(&at, cmd) == hipSuccess) printf("   (pointer attributes: type %d, device %d, host pointer %p, device pointer %p, managed %d)\n", (int)at.type, at.device, at.hostPointer, at.devicePointer, at.isManaged);
            }
            if (mode == 0) *cmd = 0; else CK(hipMemset(cmd, 0, 64));
            *ack = 0;
            CK(hipDeviceSynchronize());
            fflush(stdout);
            hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, sk, cmd, ack, N, mode == 0 ? 1 : 1);
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            std::vector<double> rt;
            bool ok = true;
            for (unsigned seq = 1; seq <= N && ok; ++seq) {
                const double t0 = now_us();
                *(volatile unsigned *)cmd = seq;
                _mm_sfence();
                while (*(volatile unsigned *)ack < seq) { _mm_pause(); if (now_us() - t0 > 500000) { ok = false; break; } }
                rt.push_back(now_us() - t0);
            }
            CK(hipStreamSynchronize(sk));
            if (!ok) { printf("DOORBELL %s: no answer (the device never saw the host's write)\n", name); continue; }
            std::sort(rt.begin(), rt.end());
            printf("DOORBELL %-72s round trip us: min %.2f median %.2f p90 %.2f\n", name, rt.front(), rt[rt.size() / 2], rt[rt.size() * 9 / 10]);
        }
    }
    printf("done\n");
    return 0;
}
