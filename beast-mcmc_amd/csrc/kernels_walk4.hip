// kernels_walk4.hip — 4-state (nucleotide) pruning as ONE launch per operation list: the "pattern walk".
//
// A site pattern never needs another pattern's data, so a thread that owns (pattern p, rate category c) can execute a
// whole dependency-ordered operation list by itself, one operation after the other, with no grid-wide synchronisation:
// what it reads was either there before the launch or written by the same thread earlier in the same launch (program
// order makes a thread's own stores visible to its later loads).  The host (engine.cpp, "walk planner") turns an
// updatePartials list into a post-order program of micro-operations; the result of a micro-operation stays in the
// thread's registers (ACC, and two hold registers H0/H1 for a value that has to wait for its sibling's subtree), so a
// child that was computed by the previous micro-operation is not read back from HBM, and a node whose subtree is a few
// compact tips ("virtual" buffer, engine.cpp) is never written at all.
//
// Mapping:  workgroup = 64 consecutive patterns x all C categories; wave w = category w; lane l = pattern p0 + l.
//           A wave's loads/stores of a partials buffer ([C][P][4] doubles) are 64 x 32 B = 2 KiB contiguous.
//           All C*P/64 waves of a 1e5-pattern alignment are resident at once (6 waves per SIMD at C = 4).
// Branch matrices are wave-uniform (one category per wave): the 4x4 mat-vec takes its matrix from SGPRs (scalar loads
// through the constant address space), no LDS.  A compact tip child needs column `state` of the matrix per lane: the
// wave loads the 16 entries once (lane j <- M[j], lanes >= 16 hold 1.0 for a missing state) and every lane picks its
// four entries with ds_bpermute (crossbar only, no LDS storage).
// Rescaling in write mode needs the per-pattern maximum over all categories: the C waves exchange their maxima
// through 2 x C x 64 doubles of LDS and one barrier (double-buffered); read mode multiplies by the stored reciprocal.
//
// Arithmetic restated from src/dr/oldevomodel/treelikelihood/NucleotideLikelihoodCore.java:54-270 /
// GeneralLikelihoodCore.java:52-203; rescaling AbstractLikelihoodCore.java:406-440 applied unconditionally.
#include "kernels.h"

namespace mi355 {

typedef double v4d __attribute__((ext_vector_type(4)));
#define MI355_CONST __attribute__((address_space(4)))

__device__ __forceinline__ v4d ldg4(const void* base, size_t elemOff) {
    return *gptr(reinterpret_cast<const v4d*>(reinterpret_cast<const double*>(base) + elemOff));
}

// y = M x with the (wave-uniform) matrix read through the scalar cache; the element order of every sum is the one
// NucleotideLikelihoodCore uses (j = 0..3), shared by every path of this file so that a value is bitwise the same
// whichever micro-operation produced it
__device__ __forceinline__ v4d matvec4s(const double MI355_CONST* __restrict__ m, v4d x) {
    v4d y;      // explicit FMA chains: the rounding sequence is fixed by the source, not by the optimiser
    y.x = __builtin_fma(m[3], x.w, __builtin_fma(m[2], x.z, __builtin_fma(m[1], x.y, m[0] * x.x)));
    y.y = __builtin_fma(m[7], x.w, __builtin_fma(m[6], x.z, __builtin_fma(m[5], x.y, m[4] * x.x)));
    y.z = __builtin_fma(m[11], x.w, __builtin_fma(m[10], x.z, __builtin_fma(m[9], x.y, m[8] * x.x)));
    y.w = __builtin_fma(m[15], x.w, __builtin_fma(m[14], x.z, __builtin_fma(m[13], x.y, m[12] * x.x)));
    return y;
}

__device__ __forceinline__ double bperm(double v, int byteAddr) {
    const int lo = __builtin_amdgcn_ds_bpermute(byteAddr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byteAddr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// column `s` of the matrix spread over lanes 0..15 (lane 16.. = 1.0): y[i] = M[i][s], or 1 for a missing state (s >= 4)
__device__ __forceinline__ v4d column4(double spread, int s) {
    const int base = s < 4 ? s * 4 : 64;            // byte address of lane s (or lane 16)
    const int step = s < 4 ? 16 : 0;                // next row = 4 lanes further
    v4d y;
    y.x = bperm(spread, base);
    y.y = bperm(spread, base + step);
    y.z = bperm(spread, base + 2 * step);
    y.w = bperm(spread, base + 3 * step);
    return y;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ const void* mkptr(unsigned lo, unsigned hi) { return (const void*)(((unsigned long long)hi << 32) | lo); }

// LDS of one workgroup: two hold slots per thread (a v4d as two 16-byte halves, each half contiguous over the lanes of
// a wave: conflict-free ds_read/write_b128) and the write-mode exchange area
typedef double v2d __attribute__((ext_vector_type(2)));

template <int MAXT>
__global__ __launch_bounds__(MAXT) void k_walk4(const u32x4 MI355_CONST* __restrict__ prog, const WalkSeg MI355_CONST* __restrict__ segs,
                                                const double* __restrict__ matrices, int P, int C, long recipOff) {
    extern __shared__ v2d lds[];                      // hold[2][C][2][64] (v2d), then exch[2][C][64] (double)
    const WalkSeg MI355_CONST& sg = segs[blockIdx.y];
    const int progStart = sg.progStart, progCount = sg.progCount, pStart = sg.pStart, pEnd = sg.pEnd;
    const int p0 = pStart + (int)blockIdx.x * 64;
    if (p0 >= pEnd) return;                           // the whole workgroup
    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool valid = p0 + lane < pEnd;
    const int p = valid ? p0 + lane : pEnd - 1;       // lanes past the end recompute the last pattern and store nothing
    const size_t off = ((size_t)c * P + p) * 4;
    const double MI355_CONST* mats = (const double MI355_CONST*)matrices + (size_t)c * 16;
    const double* matsG = matrices + (size_t)c * 16 + (lane & 15);
    v2d* holdBase = lds + (size_t)c * 128 + lane;     // + slot * C * 128 (+ 64 for the second half)
    double* exch = reinterpret_cast<double*>(lds + (size_t)2 * C * 128);
    int buf = 0;

    v4d ACC = v4d{1.0, 1.0, 1.0, 1.0};
    const u32x4 MI355_CONST* dp = prog + (size_t)progStart * 3;
    u32x4 n0 = dp[0], n1 = dp[1], n2 = dp[2];         // descriptor of the first micro-operation
    for (int k = 0; k < progCount; k++) {
        const u32x4 d0 = n0, d1 = n1, d2 = n2;
        dp += 3;
        if (k + 1 < progCount) { n0 = dp[0]; n1 = dp[1]; n2 = dp[2]; }     // next descriptor: in flight during this one
        const unsigned fl = d2.z;
        const int k1 = fl & 7, k2 = (fl >> 3) & 7, hold = (fl >> 6) & 3, smode = (fl >> 8) & 3;
        const unsigned mat1 = d2.x, mat2 = d2.y;            // element offsets of the two matrices (category 0)
        const void* src1 = mkptr(d0.z, d0.w);
        const void* src2 = mkptr(d1.x, d1.y);
        double* scale = (double*)mkptr(d1.z, d1.w);
        double* store = (double*)mkptr(d0.x, d0.y);

        // ---- everything that comes from memory is requested first
        v4d x1, x2;
        int s1 = 4, s2 = 4;
        double spread1 = 1.0, spread2 = 1.0, inv = 1.0;
        if (k1 == WK_MEM) x1 = ldg4(src1, off);
        else if (k1 == WK_TIPS) {
            s1 = gptr(reinterpret_cast<const uint8_t*>(src1))[p];
            spread1 = gptr(matsG + mat1)[0];
        } else {                                       // WK_H0 / WK_H1: the thread's own hold slot
            const v2d* h = holdBase + (size_t)(k1 - WK_H0) * C * 128;
            const v2d lo = h[0], hi = h[64];
            x1 = v4d{lo.x, lo.y, hi.x, hi.y};
        }
        if (k2 == WK_MEM) x2 = ldg4(src2, off);
        else if (k2 == WK_TIPS) {
            s2 = gptr(reinterpret_cast<const uint8_t*>(src2))[p];
            spread2 = gptr(matsG + mat2)[0];
        }
        if (smode == WS_READ) inv = gptr(scale)[recipOff + p];

        // ---- the two child factors
        v4d f1, f2;
        if (k1 == WK_TIPS) f1 = column4(lane < 16 ? spread1 : 1.0, s1);
        else f1 = matvec4s(mats + mat1, x1);
        if (k2 == WK_TIPS) f2 = column4(lane < 16 ? spread2 : 1.0, s2);
        else if (k2 == WK_MEM) f2 = matvec4s(mats + mat2, x2);
        else f2 = matvec4s(mats + mat2, ACC);
        v4d r = f1 * f2;
        if (smode == WS_READ) r = r * inv;
        else if (smode == WS_WRITE) {
            double m = fmax(fmax(fmax(0.0, r.x), fmax(r.y, r.z)), r.w);
            double* e = exch + (size_t)buf * C * 64;
            e[c * 64 + lane] = m;
            __syncthreads();
            m = 0.0;
            for (int cc = 0; cc < C; cc++) m = fmax(m, e[cc * 64 + lane]);
            buf ^= 1;
            if (!(m > 0.0)) m = 1.0;
            const double im = 1.0 / m;
            r = r * im;
            if (c == 0 && valid) { gptr(scale)[p] = m; gptr(scale)[recipOff + p] = im; }
        }
        if (store != nullptr && valid) *gptr(reinterpret_cast<v4d*>(store + off)) = r;
        if (hold) {                                    // this value waits for its sibling's subtree
            v2d* h = holdBase + (size_t)(hold - 1) * C * 128;
            h[0] = v2d{r.x, r.y}; h[64] = v2d{r.z, r.w};
        }
        ACC = r;
    }
}

void launchWalk4(hipStream_t stream, const WalkOp* dProg, const WalkSeg* dSegs, int nSegs, int maxRange,
                 const double* matrices, int P, int C, long recipOff) {
    if (nSegs <= 0 || maxRange <= 0) return;
    const dim3 grid((maxRange + 63) / 64, nSegs), block(64 * C);
    const size_t lds = (size_t)2 * C * 128 * sizeof(v2d) + (size_t)2 * C * 64 * sizeof(double);
    if (C <= 4)
        hipLaunchKernelGGL((k_walk4<256>), grid, block, lds, stream, (const u32x4 MI355_CONST*)dProg, (const WalkSeg MI355_CONST*)dSegs,
                           matrices, P, C, recipOff);
    else
        hipLaunchKernelGGL((k_walk4<1024>), grid, block, lds, stream, (const u32x4 MI355_CONST*)dProg, (const WalkSeg MI355_CONST*)dSegs,
                           matrices, P, C, recipOff);
}

}  // namespace mi355
