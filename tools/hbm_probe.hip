// Read-bandwidth probe for the slab access pattern of the half-transform (tools only, not product).
//   mode 0: flat    - a wave reads its slab as contiguous 16-byte-per-lane loads (1 KiB / instruction)
//   mode 1: pattern - the half-transform's operand-layout loads (4 rows x 256 B pairs + 4 rows x 88 B)
//   mode 2: flat, persistent waves (58 slabs per wave, next slab issued before the sum of this one)
//   mode 3: pattern with rows padded to a multiple of 128 B (what an aligned layout would give)
//   mode 4: pattern + the kernel's T2 stores (81 doubles per slab, 4 scattered 8-byte stores per lane)
//   mode 5: pattern + 45 dependent-chain fp64 MFMAs per slab (the kernel's MFMA count), no stores
//   mode 6: pattern + MFMAs + stores
//   mode 7: as 6, but one fully coalesced 16-byte-per-lane store (1 KiB per wave) instead
//   mode 8: as 6, stores issued BEFORE the MFMAs (of load sums), MFMA result unused
//   mode 9: as 6 with one wave per workgroup
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

template <int MODE, int WPB = 8>
__global__ __launch_bounds__(WPB * 64) void probe(const double* __restrict__ g, double* out, int N, long nslabs,
                                             int ld)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const long slab_elems = (long)N * ld;
    double acc = 0.0;
    if (MODE == 0) {
        long slab = (long)blockIdx.x * 8 + wave;
        if (slab >= nslabs) return;
        const double* gs = g + slab * slab_elems;
        d2u v[15];
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            long e = 2 * (lane + 64 * j);
            if (e + 1 >= slab_elems) e = slab_elems - 2;
            v[j] = *reinterpret_cast<const d2u*>(gs + e);
        }
#pragma unroll
        for (int j = 0; j < 15; ++j) acc += v[j].x + v[j].y;
        if (acc == 12345.678) out[slab] = acc;
    } else if (MODE == 14) {
        const long stride = (long)gridDim.x * 8;
        long slab = (long)blockIdx.x * 8 + wave;
        if (slab >= nslabs) return;
        d2u p0[11], p1[11];
        double s0[11], s1[11];
        auto issue = [&](long s, d2u (&ap)[11], double (&as)[11]) {
            if (s >= nslabs) s = nslabs - 1;
            const double* gs = g + s * slab_elems;
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                int r = 4 * i + lq;
                if (r >= N) r = N - 1;
                ap[i] = *reinterpret_cast<const d2u*>(gs + (long)r * ld + 2 * lr);
            }
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                int r = 4 * i + lq;
                if (r >= N) r = N - 1;
                int c = 32 + lr;
                if (c >= N) c = N - 1;
                as[i] = gs[(long)r * ld + c];
            }
        };
        issue(slab, p0, s0);
        const long n_mine = (nslabs - slab + stride - 1) / stride;
        for (long k = 0; k < n_mine; k += 2) {
            issue(slab + stride, p1, s1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 11; ++j) acc += p0[j].x + p0[j].y + s0[j];
            __builtin_amdgcn_sched_barrier(0);
            issue(slab + 2 * stride, p0, s0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 11; ++j) acc += p1[j].x + p1[j].y + s1[j];
            __builtin_amdgcn_sched_barrier(0);
            slab += 2 * stride;
        }
        if (acc == 12345.678) out[blockIdx.x] = acc;
    } else if (MODE == 1 || MODE >= 3) {
        long slab = (long)blockIdx.x * WPB + wave;
        if (slab >= nslabs) return;
        const double* gs = g + slab * slab_elems;
        d2u ap[11];
        double as[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            int r = 4 * i + lq;
            if (r >= N) r = N - 1;
            ap[i] = *reinterpret_cast<const d2u*>(gs + (long)r * ld + 2 * lr);
            int c = 32 + lr;
            if (c >= N) c = N - 1;
            as[i] = gs[(long)r * ld + c];
        }
        __builtin_amdgcn_sched_barrier(0);   // all loads in flight before any use, as in the kernel
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 11; ++i) acc += ap[i].x + ap[i].y + as[i];
            if (acc == 12345.678) out[slab] = acc;
        } else {
            typedef double d4 __attribute__((ext_vector_type(4)));
            d4 jt = {0, 0, 0, 0};
            if (MODE == 8) {
                double* dst = out + slab * 81;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int zz = lq + 4 * i;
                    if (lr < 9 && zz < 9) dst[lr * 9 + zz] = ap[i].x + ap[i + 4].y + as[i];
                }
            }
            if (MODE >= 5) {
                d4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0}, x2 = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[i].x, 1.0 + lane, x0, 0, 0, 0);
                    x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[i].y, 1.0 + lane, x1, 0, 0, 0);
                    x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(as[i], 1.0 + lane, x2, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    jt = __builtin_amdgcn_mfma_f64_16x16x4f64(2.0 + lane, x0[i], jt, 0, 0, 0);
                    jt = __builtin_amdgcn_mfma_f64_16x16x4f64(2.0 + lane, x1[i], jt, 0, 0, 0);
                    jt = __builtin_amdgcn_mfma_f64_16x16x4f64(2.0 + lane, x2[i], jt, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 11; ++i) jt[i & 3] += ap[i].x + ap[i].y + as[i];
            }
            if (MODE == 5 || MODE == 8) {
                if (jt[0] + jt[1] + jt[2] + jt[3] == 12345.678) out[slab] = jt[0];
            } else if (MODE == 7) {
                d2u v = {jt[0] + jt[1], jt[2] + jt[3]};
                *reinterpret_cast<d2u*>(out + slab * 128 + 2 * lane) = v;
            } else {
                double* dst = out + slab * 81;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int zz = lq + 4 * i;
                    if (lr < 9 && zz < 9) dst[lr * 9 + zz] = jt[i];
                }
            }
        }
    } else {
        const long stride = (long)gridDim.x * 8;
        long slab = (long)blockIdx.x * 8 + wave;
        if (slab >= nslabs) return;
        d2u v0[15], v1[15];
        auto issue = [&](long s, d2u (&v)[15]) {
            if (s >= nslabs) s = nslabs - 1;
            const double* gs = g + s * slab_elems;
#pragma unroll
            for (int j = 0; j < 15; ++j) {
                long e = 2 * (lane + 64 * j);
                if (e + 1 >= slab_elems) e = slab_elems - 2;
                v[j] = *reinterpret_cast<const d2u*>(gs + e);
            }
        };
        issue(slab, v0);
        const long n_mine = (nslabs - slab + stride - 1) / stride;
        for (long k = 0; k < n_mine; k += 2) {
            issue(slab + stride, v1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 15; ++j) acc += v0[j].x + v0[j].y;
            __builtin_amdgcn_sched_barrier(0);
            issue(slab + 2 * stride, v0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 15; ++j) acc += v1[j].x + v1[j].y;
            __builtin_amdgcn_sched_barrier(0);
            slab += 2 * stride;
        }
        if (acc == 12345.678) out[blockIdx.x] = acc;
    }
}

int main()
{
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int N = 43, G = 64;
    const long nslabs = (long)G * N * N;
    for (int mode = 0; mode < 15; ++mode) {
        const int ld = mode == 3 ? 48 : N;
        const size_t bytes = (size_t)nslabs * N * ld * 8;
        double *g, *out;
        hipMalloc(&g, bytes + 4096);
        hipMalloc(&out, nslabs * 128 * 8);
        hipMemset(g, 0, bytes + 4096);
        if (getenv("PROBE_RANDOM")) {
            std::vector<double> h(1 << 20);
            for (auto& x : h) x = rand() / (double)RAND_MAX - 0.5;
            for (size_t off = 0; off + h.size() * 8 <= bytes; off += h.size() * 8)
                hipMemcpy((char*)g + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        }
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e30f, tot = 0;
        const int reps = getenv("PROBE_REPS") ? atoi(getenv("PROBE_REPS")) : 20;
        float last = 0; int nlast = 0;
        if (getenv("PROBE_MODE") && atoi(getenv("PROBE_MODE")) != mode) continue;
        for (int r = 0; r < reps + 3; ++r) {
            hipEventRecord(e0);
            const unsigned grid = (unsigned)((nslabs + 7) / 8);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(grid), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(grid), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 7) hipLaunchKernelGGL(probe<7>, dim3(grid), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 8) hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(512), 0, 0, g, out, N, nslabs, ld);
            // occupancy sweeps of mode 1 / mode 0 through dynamic LDS: 1 or 2 workgroups (8 / 16 waves) per CU
            if (mode == 10) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 100 * 1024, 0, g, out, N, nslabs, ld);
            if (mode == 11) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 70 * 1024, 0, g, out, N, nslabs, ld);
            if (mode == 12) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 100 * 1024, 0, g, out, N, nslabs, ld);
            if (mode == 13) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 70 * 1024, 0, g, out, N, nslabs, ld);
            if (mode == 14) hipLaunchKernelGGL(probe<14>, dim3(256), dim3(512), 0, 0, g, out, N, nslabs, ld);
            if (mode == 9) hipLaunchKernelGGL((probe<6, 1>), dim3((unsigned)nslabs), dim3(64), 0, 0, g, out, N, nslabs, ld);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r >= 3) { tot += ms; if (ms < best) best = ms; }
            if (r >= reps + 3 - 20) { last += ms; ++nlast; }
        }
        const double useful = (double)nslabs * N * N * 8;
        printf("(last 20: %.1f us) ", last / nlast * 1e3);
        printf("mode %d: avg %.1f us best %.1f us -> %.2f TB/s useful (%.2f TB/s incl. padding)\n", mode,
               tot / reps * 1e3, best * 1e3, useful / (tot / reps * 1e-3) / 1e12,
               (double)bytes / (tot / reps * 1e-3) / 1e12);
        hipFree(g);
        hipFree(out);
    }
    return 0;
}
