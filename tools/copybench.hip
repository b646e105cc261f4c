// tools/copybench.hip -- how fast can THIS box stream N bytes in + N bytes out, and with which
// access shape?  Calibrates the HBM ceiling the IMDCT kernel is judged against.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/copybench tools/copybench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float __attribute__((ext_vector_type(4))) v4;

// A: classic grid-stride float4 copy
template <int NT_LD, int NT_ST>
__global__ void copy_gs(const v4 *__restrict__ in, v4 *__restrict__ out, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        v4 v = NT_LD ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT_ST) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

// B: each wave owns contiguous chunks of CH float4 per lane-row (like the IMDCT: 15 KB = 15 x 1 KB),
// loads the whole chunk into registers, then stores it.
template <int CH, int NT_LD, int NT_ST>
__global__ void copy_chunk(const v4 *__restrict__ in, v4 *__restrict__ out, size_t nchunks) {
    const int lane = threadIdx.x & 63;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t c = wave; c < nchunks; c += nw) {
        const v4 *p = in + c * (CH * 64) + lane;
        v4 *q = out + c * (CH * 64) + lane;
        v4 r[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) r[k] = NT_LD ? __builtin_nontemporal_load(p + 64 * k) : p[64 * k];
#pragma unroll
        for (int k = 0; k < CH; k++) { if (NT_ST) __builtin_nontemporal_store(r[k], q + 64 * k); else q[64 * k] = r[k]; }
    }
}

// C: read-only (sum to defeat DCE) and write-only streams
__global__ void read_only(const v4 *__restrict__ in, float *__restrict__ sink, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    v4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc += in[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
__global__ void write_only(v4 *__restrict__ out, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    v4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = v;
}
// D: grid-stride copy, U float4 per thread per iteration (all loads first, then all stores)
template <int U>
__global__ void copy_gs_u(const v4 *__restrict__ in, v4 *__restrict__ out, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
        v4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++) r[k] = in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) out[i + k * stride] = r[k];
    }
}

// E: like B but every 3840-byte row is written 240 bytes further on (the IMDCT's fin rows: raw[p] lands at
// fin[60 + p]); rows of 60 float4 (= 960 floats), 4 rows per wave, lane-contiguous 1 KB stores that straddle lines
template <int SHIFT4>
__global__ void copy_rows_shifted(const v4 *__restrict__ in, v4 *__restrict__ out, size_t nrows) {
    const int lane = threadIdx.x & 63;
    size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t g = wave; g * 4 < nrows; g += nw) {
        v4 r[16];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int h = 0; h < 4; h++) {
                int j = (h & 1) * 64 + lane;                       // 120 float4 per half row pair
                const v4 *row = in + (g * 4 + k) * 240;
                r[k * 4 + h] = (j < 120) ? row[(h < 2) ? j : 239 - j] : v4{0, 0, 0, 0};
            }
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int h = 0; h < 4; h++) {
                int j = (h & 1) * 64 + lane;
                v4 *row = out + (g * 4 + k) * 240;
                if (j < 120) {
                    int pos = (h < 2) ? j : 239 - j;
                    pos = (pos + SHIFT4) % 240;                    // SHIFT4 = 15: +240 bytes with wrap (head region)
                    row[pos] = r[k * 4 + h];
                }
            }
    }
}

template <typename F>
static double timeit(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    size_t bytes = (size_t)1 << 32;   // 4 GiB in, 4 GiB out
    if (argc > 1) bytes = (size_t)atoll(argv[1]) << 20;
    v4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    size_t n4 = bytes / 16;
    auto rep = [&](const char *name, double ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
    rep("hipMemcpyDtoD", timeit([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }, 5));
    for (int bpc : {4, 8, 16, 32}) {
        char nm[128];
        int grid = 256 * bpc;
        snprintf(nm, sizeof nm, "grid-stride f4, 256thr, %d blk/CU", bpc);
        rep(nm, timeit([&] { copy_gs<0, 0><<<grid, 256>>>(in, out, n4); }, 5));
        snprintf(nm, sizeof nm, "grid-stride f4 nt-ld nt-st, %d blk/CU", bpc);
        rep(nm, timeit([&] { copy_gs<1, 1><<<grid, 256>>>(in, out, n4); }, 5));
        snprintf(nm, sizeof nm, "grid-stride f4 nt-st only, %d blk/CU", bpc);
        rep(nm, timeit([&] { copy_gs<0, 1><<<grid, 256>>>(in, out, n4); }, 5));
    }
    {
        float *sink; CK(hipMalloc(&sink, 4));
        for (int bpc : {2, 4, 8}) {
            char nm[128];
            snprintf(nm, sizeof nm, "READ-ONLY f4, %d blk/CU (GB/s x0.5)", bpc);
            rep(nm, timeit([&] { read_only<<<256 * bpc, 256>>>(in, sink, n4); }, 5));
            snprintf(nm, sizeof nm, "WRITE-ONLY f4, %d blk/CU (GB/s x0.5)", bpc);
            rep(nm, timeit([&] { write_only<<<256 * bpc, 256>>>(out, n4); }, 5));
        }
        for (int bpc : {1, 2, 3, 4, 5, 6}) {
            char nm[128];
            snprintf(nm, sizeof nm, "grid-stride f4 x1, %d blk/CU", bpc);
            rep(nm, timeit([&] { copy_gs<0, 0><<<256 * bpc, 256>>>(in, out, n4); }, 5));
            snprintf(nm, sizeof nm, "grid-stride f4 x4, %d blk/CU", bpc);
            rep(nm, timeit([&] { copy_gs_u<4><<<256 * bpc, 256>>>(in, out, n4); }, 5));
        }
    }
    for (int wpc : {6}) {
        char nm[128];
        int grid = 256 * wpc;
        size_t nrows = (bytes / 3840) & ~(size_t)3;   // whole groups of 4 rows only: the kernel has no tail guard
        snprintf(nm, sizeof nm, "rows 3840B aligned stores, %d waves/CU", wpc);
        rep(nm, timeit([&] { copy_rows_shifted<0><<<grid, 64>>>(in, out, nrows); }, 8));
        snprintf(nm, sizeof nm, "rows 3840B stores shifted +240B, %d waves/CU", wpc);
        rep(nm, timeit([&] { copy_rows_shifted<15><<<grid, 64>>>(in, out, nrows); }, 8));
        snprintf(nm, sizeof nm, "rows 3840B aligned stores (again), %d waves/CU", wpc);
        rep(nm, timeit([&] { copy_rows_shifted<0><<<grid, 64>>>(in, out, nrows); }, 8));
        snprintf(nm, sizeof nm, "rows 3840B stores shifted +240B (again), %d waves/CU", wpc);
        rep(nm, timeit([&] { copy_rows_shifted<15><<<grid, 64>>>(in, out, nrows); }, 8));
    }
    for (int wpc : {2, 3, 4, 6}) {
        char nm[128];
        int grid = 256 * wpc / 2;   // 128-thread blocks
        snprintf(nm, sizeof nm, "chunk 15KB/wave, %d waves/CU", wpc);
        rep(nm, timeit([&] { copy_chunk<15, 0, 0><<<grid, 128>>>(in, out, n4 / (15 * 64)); }, 5));
        snprintf(nm, sizeof nm, "chunk 15KB/wave nt both, %d waves/CU", wpc);
        rep(nm, timeit([&] { copy_chunk<15, 1, 1><<<grid, 128>>>(in, out, n4 / (15 * 64)); }, 5));
        snprintf(nm, sizeof nm, "chunk 4KB/wave, %d waves/CU", wpc);
        rep(nm, timeit([&] { copy_chunk<4, 0, 0><<<grid, 128>>>(in, out, n4 / (4 * 64)); }, 5));
    }
    return 0;
}
