// Calibration of the TCC FETCH_SIZE / WRITE_SIZE counters on gfx950 for the access shapes the frame kernel
// uses (MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports wide streaming reads by 2x; nothing is
// said about 4-byte-per-lane rows and 4-byte gathers).  Each kernel moves a KNOWN number of bytes from / to
// a buffer far larger than the caches; tests/gpu_fetch_calib.sh runs every kernel under
// rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE and prints counter x 1024 / bytes moved.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void read4(const float *p, float *out, size_t n)          // 4 B per lane, coalesced rows
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    float acc = 0;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void read16(const float4 *p, float *out, size_t n)       // 16 B per lane
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    float acc = 0;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void gather4(const float *p, float *out, size_t n, size_t lines)   // one 4-byte word per 128-byte line
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    float acc = 0;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) acc += p[((i * 2654435761ull) % lines) * 32];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void write4(float *p, size_t n)
{
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = (float) i;
}

int main(int argc, char **argv)
{
    const char *which = argc > 1 ? argv[1] : "read4";
    const size_t bytes = (size_t) 8 << 30;                    // 8 GiB: 32 x the 256 MB of last-level cache
    float *buf, *out;
    if (hipMalloc((void **) &buf, bytes) != hipSuccess || hipMalloc((void **) &out, 256) != hipSuccess) return 1;
    (void) hipMemset(buf, 0, bytes);
    (void) hipDeviceSynchronize();
    const size_t n4 = bytes / 4, n16 = bytes / 16, lines = bytes / 128;
    double moved = 0;
    if (!strcmp(which, "read4"))        { hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, buf, out, n4); moved = (double) bytes; }
    else if (!strcmp(which, "read16"))  { hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const float4 *) buf, out, n16); moved = (double) bytes; }
    else if (!strcmp(which, "gather4")) { hipLaunchKernelGGL(gather4, dim3(4096), dim3(256), 0, 0, buf, out, lines, lines); moved = (double) lines * 4; }
    else if (!strcmp(which, "write4"))  { hipLaunchKernelGGL(write4, dim3(4096), dim3(256), 0, 0, buf, n4); moved = (double) bytes; }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("%s useful_bytes %.0f lines_touched_bytes %.0f\n", which, moved, !strcmp(which, "gather4") ? (double) lines * 128 : moved);
    return 0;
}
