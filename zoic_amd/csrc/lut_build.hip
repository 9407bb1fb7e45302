// lut_build.hip -- exitPupilLUT(&ld, 32, 100000) (zoic.cpp:1391-1452) entirely on the GPU.
//
// The reference draws 2 x 100000 numbers per film position from ONE sequential xorshift128 stream (zoic.cpp:1411-1412),
// traces the probe rays and grows a bounding box in draw order, restarting it whenever min.x + min.y == 0 (zoic.cpp:1423-
// 1428; true for the empty box, i.e. at the first accepted probe -- and, in principle, later).  Round 1 moved the traces to
// the GPU and left the draws (6.4 M numbers), 64 host-to-device copies and the order-dependent replay (3.2 M iterations) on
// the host: 10 ms.  Here:
//   * xorshift128 is linear over GF(2): thread b of film position i starts at state  J^b E^i s0  (xorshift_jump.hpp: E = the
//     matrix of 200000 draws, applied on the host; J = the matrix of one thread's 256 draws, J^(2^k) in device memory) and
//     draws, traces and boxes its 128 probes on its own (probe kernel);
//   * without a second restart the reference's box is simply min / max over the accepted probes.  The check kernel re-walks
//     every thread's accepted probes against the TRUE running minima (exclusive prefix over the threads before it) and
//     raises a flag if min.x + min.y == 0 ever holds again; the host then falls back to the sequential replay for the whole
//     table (never seen; the test forces it).
// Same strict arithmetic as the host tracer: tables bit-identical (tests/test_parity_gpu.py).
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>

#include "kernels.hpp"
#include "lens_system.hpp"
#include "optics.hpp"
#include "xorshift_jump.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

namespace {

constexpr int kProbesPerEntry = 100000;                                   // exitPupilLUT(&ld, 32, 100000)
constexpr int kProbesPerThread = 128;
constexpr int kThreadsPerEntry = (kProbesPerEntry + kProbesPerThread - 1) / kProbesPerThread;   // 782
constexpr int kJumpLevels = 10;                                           // 782 < 2^10
constexpr int kProbeBlock = 128;

struct LutJumps { BitMat128 level[kJumpLevels]; };                         // J^(2^k), J = 256 draws

struct ThreadBox { float minX, minY, maxX, maxY; uint32_t accepted[4]; };  // one thread's 128 probes: plain box + accept bits

__device__ __forceinline__ Rng thread_start(const LutJumps *__restrict__ jumps, Rng s, uint32_t b)
{
#pragma unroll 1
    for (int k = 0; k < kJumpLevels; ++k) {
        if (((b >> k) & 1u) == 0u) continue;
        const uint32_t v[4] = {s.x, s.y, s.z, s.w};
        uint32_t o[4];
        bitmat_apply(jumps->level[k], v, o);
        s = Rng{o[0], o[1], o[2], o[3]};
    }
    return s;
}

struct EntryBases { Rng s[kLutEntries]; };

__global__ __launch_bounds__(kProbeBlock) void lut_build_probe_kernel(const KolbTable T, const EntryBases bases, const LutJumps *__restrict__ jumps,
                                                                      ThreadBox *__restrict__ boxes, unsigned int *tirOut)
{
    const uint32_t entry = blockIdx.y, b = blockIdx.x * kProbeBlock + threadIdx.x;
    uint32_t tir = 0;
    if (b < static_cast<uint32_t>(kThreadsPerEntry)) {
        Rng rng = thread_start(jumps, bases.s[entry], b);
        const float spacing = 4.0f / static_cast<float>(kLutEntries);
        const float ox = static_cast<float>(spacing * static_cast<float>(entry));   // zoic.cpp:1401
        const float ap0 = T.rearAperture;
        ThreadBox tb{0.0f, 0.0f, 0.0f, 0.0f, {0u, 0u, 0u, 0u}};
        bool any = false;
        const int first = static_cast<int>(b) * kProbesPerThread;
        const int count = (kProbesPerEntry - first) < kProbesPerThread ? (kProbesPerEntry - first) : kProbesPerThread;
        for (int p = 0; p < count; ++p) {
            const float U = (rng_unit(xor128(rng)) * 2.0f) - 1.0f;          // zoic.cpp:1411-1412
            const float V = (rng_unit(xor128(rng)) * 2.0f) - 1.0f;
            V3 o{ox, 0.0f, T.originShift};
            V3 d{(U * ap0) - ox, (V * ap0) - 0.0f, T.dirZ};
            if (trace_lens_strict(T, o, d, tir)) {
                const float px = U * ap0, py = V * ap0;
                if (!any) { tb.minX = tb.maxX = px; tb.minY = tb.maxY = py; any = true; }
                else {
                    if (px > tb.maxX) tb.maxX = px;
                    if (py > tb.maxY) tb.maxY = py;
                    if (px < tb.minX) tb.minX = px;
                    if (py < tb.minY) tb.minY = py;
                }
                tb.accepted[p >> 5] |= 1u << (p & 31);
            }
        }
        boxes[entry * kThreadsPerEntry + b] = tb;
    }
    for (int off = 32; off > 0; off >>= 1) tir += __shfl_down(tir, off, 64);
    if ((threadIdx.x & 63) == 0 && tir) atomicAdd(tirOut, tir);
}

// one workgroup per film position: exclusive prefix of the running minima over the threads of the probe kernel, the
// restart check, the final box
__global__ __launch_bounds__(1024) void lut_build_check_kernel(const KolbTable T, const EntryBases bases, const LutJumps *__restrict__ jumps,
                                                               const ThreadBox *__restrict__ boxes, LutBox *__restrict__ out, unsigned int *restartFlags)
{
    __shared__ float sMinX[1024], sMinY[1024], sMaxX[1024], sMaxY[1024];
    __shared__ int sAny[1024];
    __shared__ int sRestart;
    const uint32_t entry = blockIdx.x, b = threadIdx.x;
    const float inf = __builtin_inff();
    ThreadBox tb{inf, inf, -inf, -inf, {0u, 0u, 0u, 0u}};
    bool any = false;
    if (b < static_cast<uint32_t>(kThreadsPerEntry)) {
        tb = boxes[entry * kThreadsPerEntry + b];
        any = (tb.accepted[0] | tb.accepted[1] | tb.accepted[2] | tb.accepted[3]) != 0u;
        if (!any) { tb.minX = tb.minY = inf; tb.maxX = tb.maxY = -inf; }
    }
    if (b == 0) sRestart = 0;
    // inclusive scan of (min x, min y, any) over the threads, Hillis-Steele; max only needs the total
    sMinX[b] = tb.minX; sMinY[b] = tb.minY; sMaxX[b] = tb.maxX; sMaxY[b] = tb.maxY; sAny[b] = any ? 1 : 0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        float ax = sMinX[b], ay = sMinY[b], bx = sMaxX[b], by = sMaxY[b];
        int aa = sAny[b];
        if (b >= static_cast<uint32_t>(off)) {
            ax = fminf(ax, sMinX[b - off]); ay = fminf(ay, sMinY[b - off]);
            bx = fmaxf(bx, sMaxX[b - off]); by = fmaxf(by, sMaxY[b - off]);
            aa |= sAny[b - off];
        }
        __syncthreads();
        sMinX[b] = ax; sMinY[b] = ay; sMaxX[b] = bx; sMaxY[b] = by; sAny[b] = aa;
        __syncthreads();
    }
    // re-walk this thread's accepted probes with the running minima as the reference has them at that point
    if (b < static_cast<uint32_t>(kThreadsPerEntry) && any) {
        bool have = b > 0 && sAny[b - 1] != 0;                 // an accepted probe exists before this thread's first
        float mx = have ? sMinX[b - 1] : 0.0f, my = have ? sMinY[b - 1] : 0.0f;
        Rng rng = thread_start(jumps, bases.s[entry], b);
        const float ap0 = T.rearAperture;
        bool restart = false;
        for (int p = 0; p < kProbesPerThread; ++p) {
            const float U = (rng_unit(xor128(rng)) * 2.0f) - 1.0f;
            const float V = (rng_unit(xor128(rng)) * 2.0f) - 1.0f;
            if (((tb.accepted[p >> 5] >> (p & 31)) & 1u) == 0u) continue;
            const float px = U * ap0, py = V * ap0;
            if (!have) { mx = px; my = py; have = true; }       // the empty box: min.x + min.y == 0, the reference's one expected restart
            else {
                if ((mx + my) == 0.0f) restart = true;          // zoic.cpp:1423: the box would restart HERE, dropping what it held
                if (px < mx) mx = px;
                if (py < my) my = py;
            }
        }
        if (restart) sRestart = 1;
    }
    __syncthreads();
    if (b == 0) {
        LutBox box;                                             // no accepted probe at all: the reference's zero box
        if (sAny[1023]) { box.minX = sMinX[1023]; box.minY = sMinY[1023]; box.maxX = sMaxX[1023]; box.maxY = sMaxY[1023]; }
        out[entry] = box;
        if (sRestart) atomicAdd(restartFlags, 1u);
    }
}

struct LutDeviceState {
    std::once_flag once;
    BitMat128 entryJump;          // 200000 draws
    LutJumps jumps;               // host copy
    LutJumps *dJumps[64] = {};    // per device
    std::mutex m;
};
LutDeviceState g_lut;

}  // namespace

// 0 = boxes / tir / the stream's final state are the reference's; 1 = a second box restart was detected (caller falls back to
// the sequential replay; nothing written); otherwise a hipError_t
int build_lut_device(const KolbTable &table, Rng &rng, LutBox boxes[kLutEntries], uint32_t *tirCount)
{
    std::call_once(g_lut.once, [] {
        const BitMat128 step = xor128_step_matrix();
        g_lut.entryJump = bitmat_pow(step, 2ull * kProbesPerEntry);
        g_lut.jumps.level[0] = bitmat_pow(step, 2ull * kProbesPerThread);
        for (int k = 1; k < kJumpLevels; ++k) g_lut.jumps.level[k] = bitmat_mul(g_lut.jumps.level[k - 1], g_lut.jumps.level[k - 1]);
    });
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) return static_cast<int>(e != hipSuccess ? e : hipErrorInvalidDevice);
    LutJumps *dJ = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_lut.m);
        if (!g_lut.dJumps[dev]) {
            e = hipMalloc(reinterpret_cast<void **>(&g_lut.dJumps[dev]), sizeof(LutJumps));
            if (e == hipSuccess) e = hipMemcpy(g_lut.dJumps[dev], &g_lut.jumps, sizeof(LutJumps), hipMemcpyHostToDevice);
            if (e != hipSuccess) { g_lut.dJumps[dev] = nullptr; return static_cast<int>(e); }
        }
        dJ = g_lut.dJumps[dev];
    }
    EntryBases bases;
    Rng s = rng;
    for (int i = 0; i < kLutEntries; ++i) { bases.s[i] = s; s = bitmat_apply(g_lut.entryJump, s); }   // s ends as the state after all 6.4 M draws
    ThreadBox *dBoxes = nullptr;
    LutBox *dOut = nullptr;
    unsigned int *dWords = nullptr;   // [0] TIR bumps, [1] restart flags
    e = hipMalloc(reinterpret_cast<void **>(&dBoxes), sizeof(ThreadBox) * kLutEntries * kThreadsPerEntry);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dOut), sizeof(LutBox) * kLutEntries);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dWords), 2 * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(dWords, 0, 2 * sizeof(unsigned int));
    LutBox host[kLutEntries];
    unsigned int words[2] = {0, 0};
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lut_build_probe_kernel, dim3((kThreadsPerEntry + kProbeBlock - 1) / kProbeBlock, kLutEntries), dim3(kProbeBlock), 0, nullptr,
                           table, bases, dJ, dBoxes, dWords);
        hipLaunchKernelGGL(lut_build_check_kernel, dim3(kLutEntries), dim3(1024), 0, nullptr, table, bases, dJ, dBoxes, dOut, dWords + 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(host, dOut, sizeof(host), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(words, dWords, sizeof(words), hipMemcpyDeviceToHost);
    for (void *p : {static_cast<void *>(dBoxes), static_cast<void *>(dOut), static_cast<void *>(dWords)}) if (p) (void)hipFree(p);
    if (e != hipSuccess) return static_cast<int>(e);
    if (words[1] != 0u) return 1;
    std::memcpy(boxes, host, sizeof(host));
    *tirCount += words[0];
    rng = s;
    return 0;
}

}  // namespace zoic
