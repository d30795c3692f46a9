// Developer aid (GPU box): how fast does a 100+ MB file get from the page cache into HBM?
//   a) hipMemcpy straight from the mmap (pageable)            b) hipHostRegister the mapping, then hipMemcpy
//   c) read() into a pinned buffer, then hipMemcpy            d) mmap + MAP_POPULATE, then a)
//   e) chunked: memcpy into two pinned staging buffers on the calling thread, async copies behind it
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/h2d.hip -o tools/micro/h2d ; run: tools/micro/h2d <file>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    if (argc < 2) return 1;
    int fd = open(argv[1], O_RDONLY);
    struct stat st; fstat(fd, &st);
    const size_t N = st.st_size;
    void* d = nullptr;
    hipMalloc(&d, N + 4096);
    hipMemset(d, 0, N);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        { double t0 = now(); void* m = mmap(nullptr, N, PROT_READ, MAP_PRIVATE, fd, 0); hipMemcpy(d, m, N, hipMemcpyHostToDevice); double t1 = now(); munmap(m, N);
          printf("a) mmap + pageable hipMemcpy          %7.2f ms  %5.1f GB/s\n", t1 - t0, N / (t1 - t0) / 1e6); }
        { double t0 = now(); void* m = mmap(nullptr, N, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0); double tm = now(); hipMemcpy(d, m, N, hipMemcpyHostToDevice); double t1 = now(); munmap(m, N);
          printf("d) mmap POPULATE %.2f + hipMemcpy     %7.2f ms  %5.1f GB/s\n", tm - t0, t1 - t0, N / (t1 - t0) / 1e6); }
        { double t0 = now(); void* m = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_POPULATE, fd, 0); double tm = now();
          hipError_t e = hipHostRegister(m, N, hipHostRegisterDefault); double tr = now();
          if (e == hipSuccess) { hipMemcpy(d, m, N, hipMemcpyHostToDevice); double tc = now(); hipHostUnregister(m); double t1 = now();
            printf("b) map %.2f register %.2f copy %.2f unregister %.2f = %7.2f ms  %5.1f GB/s\n", tm - t0, tr - tm, tc - tr, t1 - tc, t1 - t0, N / (t1 - t0) / 1e6); }
          else printf("b) hipHostRegister failed: %s\n", hipGetErrorString(e));
          munmap(m, N); (void)hipGetLastError(); }
        { static void* pin = nullptr; if (!pin) hipHostMalloc(&pin, N, hipHostMallocDefault);
          double t0 = now(); size_t got = 0; while (got < N) { ssize_t r = pread(fd, (char*)pin + got, N - got, got); if (r <= 0) break; got += r; } double tr = now();
          hipMemcpy(d, pin, N, hipMemcpyHostToDevice); double t1 = now();
          printf("c) pread into pinned %.2f + copy %.2f = %7.2f ms  %5.1f GB/s\n", tr - t0, t1 - tr, t1 - t0, N / (t1 - t0) / 1e6); }
        { static void* stg[2] = {nullptr, nullptr}; const size_t CH = 8u << 20; if (!stg[0]) { hipHostMalloc(&stg[0], CH, 0); hipHostMalloc(&stg[1], CH, 0); }
          hipStream_t s; hipStreamCreate(&s); hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
          double t0 = now(); void* m = mmap(nullptr, N, PROT_READ, MAP_PRIVATE, fd, 0);
          int k = 0; for (size_t off = 0; off < N; off += CH, k ^= 1) { size_t n = N - off < CH ? N - off : CH; if (off >= 2 * CH) hipEventSynchronize(ev[k]);
              memcpy(stg[k], (char*)m + off, n); hipMemcpyAsync((char*)d + off, stg[k], n, hipMemcpyHostToDevice, s); hipEventRecord(ev[k], s); }
          hipStreamSynchronize(s); double t1 = now(); munmap(m, N); hipStreamDestroy(s);
          printf("e) chunked memcpy -> 2 pinned buffers    %7.2f ms  %5.1f GB/s\n", t1 - t0, N / (t1 - t0) / 1e6); }
        printf("\n");
    }
    return 0;
}
