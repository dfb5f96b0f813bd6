// host_probe.hip -- what the host side of a deep ObserveImage queue costs on the GPU box: memcpy pageable -> pinned
// (1..4 threads), one H2D copy command per batch, event query / record, an empty launch.
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp/host_probe tools/exp/host_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
using Clock = std::chrono::steady_clock;
static double sec(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1000) *p = 1; }
int main() {
  const size_t frame = 2 * 640 * 480, nsrc = 64, nslot = 64;
  std::vector<uint8_t> src(frame * nsrc, 7);
  uint8_t* pin = nullptr;
  hipHostMalloc((void**)&pin, frame * nslot, hipHostMallocDefault);
  memset(pin, 1, frame * nslot);
  uint8_t* dev = nullptr;
  hipMalloc((void**)&dev, frame * nslot);
  std::printf("hw threads %u\n", std::thread::hardware_concurrency());
  for (int nt = 1; nt <= 4; nt *= 2) {
    const int reps = 2000;
    auto t0 = Clock::now();
    for (int r = 0; r < reps; r++) {
      const uint8_t* s = src.data() + (size_t)(r % nsrc) * frame;
      uint8_t* d = pin + (size_t)(r % nslot) * frame;
      if (nt == 1) {
        memcpy(d, s, frame);
      } else {
        std::vector<std::thread> th;
        const size_t part = frame / nt;
        for (int t = 1; t < nt; t++) th.emplace_back([=] { memcpy(d + t * part, s + t * part, part); });
        memcpy(d, s, part);
        for (auto& x : th) x.join();
      }
    }
    const double dt = sec(t0, Clock::now());
    std::printf("memcpy %d thread(s) (spawned per frame): %.1f us per frame, %.2f GB/s\n", nt, 1e6 * dt / reps, frame * reps / dt / 1e9);
  }
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (size_t nb : {1, 4, 16, 32, 64}) {
    const int reps = 50;
    hipStreamSynchronize(st);
    auto t0 = Clock::now();
    for (int r = 0; r < reps; r++) hipMemcpyAsync(dev, pin, frame * nb, hipMemcpyHostToDevice, st);
    hipStreamSynchronize(st);
    const double dt = sec(t0, Clock::now());
    std::printf("H2D %zu frames per command: %.1f us per command, %.2f GB/s\n", nb, 1e6 * dt / reps, frame * nb * reps / dt / 1e9);
  }
  for (size_t nb : {1, 16, 64}) {  // D2H of results (100 KB per frame)
    const int reps = 50;
    auto t0 = Clock::now();
    for (int r = 0; r < reps; r++) hipMemcpyAsync(pin, dev, 100000 * nb, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    const double dt = sec(t0, Clock::now());
    std::printf("D2H %zu x 100 KB per command: %.1f us per command, %.2f GB/s\n", nb, 1e6 * dt / reps, 100000.0 * nb * reps / dt / 1e9);
  }
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  {
    const int reps = 20000;
    hipEventRecord(ev, st);
    hipStreamSynchronize(st);
    auto t0 = Clock::now();
    int ok = 0;
    for (int r = 0; r < reps; r++) ok += hipEventQuery(ev) == hipSuccess;
    double dt = sec(t0, Clock::now());
    std::printf("hipEventQuery: %.2f us (%d)\n", 1e6 * dt / reps, ok);
    t0 = Clock::now();
    for (int r = 0; r < 2000; r++) hipEventRecord(ev, st);
    dt = sec(t0, Clock::now());
    hipStreamSynchronize(st);
    std::printf("hipEventRecord: %.2f us\n", 1e6 * dt / 2000);
    t0 = Clock::now();
    for (int r = 0; r < 2000; r++) empty_kernel<<<1, 64, 0, st>>>(nullptr);
    dt = sec(t0, Clock::now());
    hipStreamSynchronize(st);
    std::printf("launch (host side): %.2f us\n", 1e6 * dt / 2000);
    t0 = Clock::now();
    for (int r = 0; r < 200; r++) {
      empty_kernel<<<1, 64, 0, st>>>(nullptr);
      hipStreamSynchronize(st);
    }
    dt = sec(t0, Clock::now());
    std::printf("launch + sync round trip: %.2f us\n", 1e6 * dt / 200);
    // pinned-flag polling: kernel writes a flag into mapped host memory, host spins on it
    int* flag = nullptr;
    hipHostMalloc((void**)&flag, 64, hipHostMallocMapped);
    *flag = 0;
    t0 = Clock::now();
    for (int r = 0; r < 200; r++) {
      *(volatile int*)flag = 0;
      empty_kernel<<<1, 1024, 0, st>>>(flag);
      while (*(volatile int*)flag == 0) {}
    }
    dt = sec(t0, Clock::now());
    hipStreamSynchronize(st);
    std::printf("launch + pinned-flag poll round trip: %.2f us\n", 1e6 * dt / 200);
    // a 614 KB upload followed by a kernel and a flag
    t0 = Clock::now();
    for (int r = 0; r < 200; r++) {
      *(volatile int*)flag = 0;
      hipMemcpyAsync(dev, pin, frame, hipMemcpyHostToDevice, st);
      empty_kernel<<<1, 1024, 0, st>>>(flag);
      while (*(volatile int*)flag == 0) {}
    }
    dt = sec(t0, Clock::now());
    std::printf("upload of one frame + kernel + flag: %.2f us\n", 1e6 * dt / 200);
  }
  return 0;
}
