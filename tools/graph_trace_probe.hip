// Discriminating probe for the rocprofv3 --kernel-trace crash inside hipGraphLaunch (profiles/r02_rocprofv3_graph_crash.txt).
// Builds hipGraphs by stream capture the way gmres.hip does and relaunches them; which variant dies under the tracer says
// whether a node of OURS (the by-value kernel argument of ~400 bytes, the captured device-to-host copy into pinned memory)
// is at fault or the tool's graph-launch interception.  A SIGSEGV handler prints the faulting address, the frames with
// their modules (dladdr) and the /proc/self/maps line that holds each frame, so the backtrace is symbolised on the box.
//
//   hipcc --offload-arch=gfx950 -O2 tools/graph_trace_probe.hip -o gpurun_out/graph_probe -ldl
//   rocprofv3 --kernel-trace -d gpurun_out/probe_X -- gpurun_out/graph_probe <variant> [relaunches]
// variants: memcpy | kernel_small | kernel_big | both | cycle (62 kernels + the copy, two ping-pong execs as gmres.hip)
//           [relaunches] [kernels per graph]: enough relaunches x packets to wrap the 1 MiB AQL ring (16384 packets) several times,
//           with a packet count per launch that does not divide the ring, reproduces the fault without any library code
#include <dlfcn.h>
#include <execinfo.h>
#include <hip/hip_runtime.h>
#include <signal.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

struct Big {
    double a[48];       // 384 bytes + the tail below: the size class of gmres.hip's GDev
    double *out;
    int n, pad;
};
struct Small {
    double *out;
    int n;
};

__global__ void k_big(Big b, int j) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.n) b.out[i] = b.a[j % 48] + i;
}
__global__ void k_small(Small s, int j) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < s.n) s.out[i] = j + i;
}

static void print_maps_line(void *addr) {
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f) return;
    char line[512];
    while (fgets(line, sizeof line, f)) {
        unsigned long lo, hi;
        if (sscanf(line, "%lx-%lx", &lo, &hi) == 2 && (unsigned long)addr >= lo && (unsigned long)addr < hi) {
            fprintf(stderr, "      maps: %s", line);
            break;
        }
    }
    fclose(f);
}

static void on_segv(int, siginfo_t *si, void *) {
    fprintf(stderr, "\n*** probe SIGSEGV at address %p\n", si->si_addr);
    fprintf(stderr, "    mapping that ends at / holds the faulting address:\n");
    print_maps_line((char *)si->si_addr - 1);
    print_maps_line(si->si_addr);
    void *fr[48];
    const int n = backtrace(fr, 48);
    for (int i = 0; i < n; ++i) {
        Dl_info di;
        if (dladdr(fr[i], &di) && di.dli_fname)
            fprintf(stderr, "  #%d %p  %s + 0x%lx  (%s)\n", i, fr[i], di.dli_fname,
                    (unsigned long)((char *)fr[i] - (char *)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
        else
            fprintf(stderr, "  #%d %p  ?\n", i, fr[i]);
    }
    _exit(139);
}

int main(int argc, char **argv) {
    const char *variant = argc > 1 ? argv[1] : "both";
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, nullptr);

    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 1 << 16;
    double *d = nullptr, *h = nullptr;
    CK(hipMalloc((void **)&d, n * sizeof(double)));
    CK(hipHostMalloc((void **)&h, 2 * 64, hipHostMallocDefault));
    Big b;
    for (int i = 0; i < 48; ++i) b.a[i] = i;
    b.out = d;
    b.n = n;
    b.pad = 0;
    Small s{d, n};
    const bool want_copy = !strcmp(variant, "memcpy") || !strcmp(variant, "both") || !strcmp(variant, "cycle");
    const bool want_big = !strcmp(variant, "kernel_big") || !strcmp(variant, "both") || !strcmp(variant, "cycle");
    const bool want_small = !strcmp(variant, "kernel_small");
    const int nk = argc > 3 ? atoi(argv[3]) : (!strcmp(variant, "cycle") ? 62 : 1);

    hipGraph_t g[2];
    hipGraphExec_t ex[2];
    hipEvent_t ev[2];
    for (int k = 0; k < 2; ++k) {
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int j = 0; j < nk; ++j) {
            if (want_big) hipLaunchKernelGGL(k_big, dim3(n / 256), dim3(256), 0, st, b, j);
            if (want_small) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, st, s, j);
        }
        if (want_copy) CK(hipMemcpyAsync((char *)h + 64 * k, d, 56, hipMemcpyDeviceToHost, st));
        CK(hipStreamEndCapture(st, &g[k]));
        CK(hipGraphInstantiate(&ex[k], g[k], nullptr, nullptr, 0));
        CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    }
    fprintf(stderr, "probe %s: graphs built, relaunching %d times (ping-pong, one launch ahead)\n", variant, reps);
    CK(hipGraphLaunch(ex[0], st));
    CK(hipEventRecord(ev[0], st));
    for (int c = 0; c < reps; ++c) {
        const int cur = c & 1, nxt = cur ^ 1;
        CK(hipGraphLaunch(ex[nxt], st));
        CK(hipEventRecord(ev[nxt], st));
        CK(hipEventSynchronize(ev[cur]));
    }
    CK(hipStreamSynchronize(st));
    fprintf(stderr, "probe %s: OK (%d relaunches), h[0] = %g\n", variant, reps, h[0]);
    return 0;
}
