// What does a small dependent kernel cost when several host threads drive the GPU at once?  (round 4: why 8 sequences on one
// MI355X run at 2.5-3x the frame latency of one -- every kernel of the 8-sequence trace is ~20 us longer than alone.)
//   hipcc --offload-arch=gfx950 -O2 tools/multiq.hip -o xrslam_amd/bin/xr-multiq -pthread
// T host threads, a stream each; every thread runs N rounds of: launch a chain of C tiny kernels, the last one publishes a
// sequence number into pinned host memory (system-scope fence, like the library's mailboxes); spin on it.  Variants:
//   mode 0  kernels of one workgroup that touch nothing but the mailbox
//   mode 1  every kernel of the chain writes 64 KB to HBM (a dirty L2 to write back at the kernel boundary)
//   mode 2  one workgroup per kernel that spins ~20 us on the shader clock (a "kb_chain"-like occupant)
//   mode 3  256 workgroups per kernel (a wide, short kernel)
// Prints one JSON line per (T, C, mode): rounds/s per thread, kernels/s over all threads, mean round latency in us.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_step(int mode, int *host_flag, int seq, int last, char *scratch) {
    if (mode == 1 || mode == 3) {
        const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        reinterpret_cast<int *>(scratch)[i & 16383] = seq;
    }
    if (mode == 2) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 2000) {   // 100 MHz ticks: 20 us
        }
    }
    if (last && blockIdx.x == 0 && threadIdx.x == 0) {
        __threadfence_system();
        *reinterpret_cast<volatile int *>(host_flag) = seq;
    }
}

struct Result {
    double seconds = 0;
};

static void worker(int mode, int chain, int rounds, Result *res, std::atomic<int> *go) {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *h_flag = nullptr, *d_flag = nullptr;
    CK(hipHostMalloc(&h_flag, 64, hipHostMallocDefault));
    *h_flag = 0;
    CK(hipHostGetDevicePointer((void **)&d_flag, h_flag, 0));
    char *scratch = nullptr;
    CK(hipMalloc(&scratch, 65536));
    const dim3 grid(mode == 3 ? 256 : (mode == 1 ? 64 : 1)), block(256);
    auto round = [&](int seq) {
        for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(k_step, grid, block, 0, s, mode, d_flag, seq, c + 1 == chain ? 1 : 0, scratch);
        volatile int *f = h_flag;
        while (*f != seq) {
        }
    };
    for (int i = 1; i <= 50; ++i) round(i);
    go->fetch_add(1);
    while (go->load() < 0) {
    }
    // all threads started: wait for the common start
    while (go->load() < 1000000) {
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < rounds; ++i) round(51 + i);
    res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CK(hipStreamSynchronize(s));
    hipFree(scratch);
    hipHostFree(h_flag);
    hipStreamDestroy(s);
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    CK(hipSetDevice(0));
    const int Ts[] = {1, 2, 4, 8, 16};
    const int Cs[] = {1, 3, 6};
    for (int mode = 0; mode < 4; ++mode)
        for (int chain : Cs)
            for (int T : Ts) {
                std::vector<Result> res(T);
                std::vector<std::thread> th;
                std::atomic<int> go{0};
                for (int t = 0; t < T; ++t)
                    th.emplace_back([&, t] {
                        CK(hipSetDevice(0));
                        worker(mode, chain, rounds, &res[t], &go);
                    });
                while (go.load() < T) std::this_thread::yield();
                go.store(1000000);
                for (auto &x : th) x.join();
                double worst = 0, sum_rate = 0;
                for (auto &r : res) {
                    worst = r.seconds > worst ? r.seconds : worst;
                    sum_rate += rounds / r.seconds;
                }
                printf("{\"mode\": %d, \"chain\": %d, \"threads\": %d, \"round_us\": %.2f, \"rounds_per_s_all\": %.0f, \"kernels_per_s_all\": %.0f}\n",
                       mode, chain, T, 1e6 * worst / rounds, sum_rate, sum_rate * chain);
                fflush(stdout);
            }
    return 0;
}
