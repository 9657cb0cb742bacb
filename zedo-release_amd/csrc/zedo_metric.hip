// Hypothesis selection for gfx950: per-(hypothesis,pose) MPJPE / Procrustes-aligned MPJPE in fp64 and the
// per-pose minimum over hypotheses (reference lib/dataset/h36m.py:394-417, lib/dataset/pw3d.py:302-331,
// lib/utils/transforms.py:42-127).  One lane per row for the errors, the rows of a wave staged through the LDS
// with coalesced loads; the arg-min with one wavefront per pose (few poses) or one lane per pose (many: coalesced
// reads of one hypothesis' errors at a time).
#include "zedo_internal.h"

#include <algorithm>

namespace zedo {

// One-sided Jacobi on the 3x3 matrix M (columns rotated until orthogonal): M V = U diag(s).
// Returns R = V U^T (the rotation/reflection of procrustes(..., reflection='best'), transforms.py:93-96)
// and the sum of singular values.
__device__ void polar_from_svd(double M[3][3], double R[3][3], double &strace) {
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) { al += M[i][p] * M[i][p]; be += M[i][q] * M[i][q]; ga += M[i][p] * M[i][q]; }
                if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                off = fmax(off, fabs(ga) / sqrt(al * be));
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double mp = M[i][p], mq = M[i][q];
                    M[i][p] = c * mp - s * mq; M[i][q] = s * mp + c * mq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double s[3], U[3][3];
    double smax = 0;
    for (int c = 0; c < 3; ++c) {
        s[c] = sqrt(M[0][c] * M[0][c] + M[1][c] * M[1][c] + M[2][c] * M[2][c]);
        smax = fmax(smax, s[c]);
    }
    int nbad = 0, bad = -1;
    for (int c = 0; c < 3; ++c) {
        if (s[c] > 1e-14 * smax && s[c] > 0) {
            for (int i = 0; i < 3; ++i) U[i][c] = M[i][c] / s[c];
        } else { ++nbad; bad = c; }
    }
    if (nbad == 1) {
        // rank-2 input (planar pose): complete U with the cross product of the other two columns and pick
        // the sign that makes R a proper rotation.  numpy's LAPACK picks an arbitrary sign here.
        const int a = (bad + 1) % 3, b = (bad + 2) % 3;
        U[0][bad] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
        U[1][bad] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
        U[2][bad] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
        const double detV = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) - V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) +
                            V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
        if (detV < 0) for (int i = 0; i < 3; ++i) U[i][bad] = -U[i][bad];
        s[bad] = 0;
    } else if (nbad > 1) {
        for (int c = 0; c < 3; ++c) for (int i = 0; i < 3; ++i) U[i][c] = (i == c);
        for (int c = 0; c < 3; ++c) for (int i = 0; i < 3; ++i) V[i][c] = (i == c);
    }
    strace = s[0] + s[1] + s[2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + V[i][2] * U[j][2];
}

// Error of one row from its staged operands: p = the row's J*3 fp32 coordinates, g = its pose's J*3 fp64 ground-truth coordinates
// (both register / LDS resident: every access below is a plain indexed read).  The statements - and therefore every rounding - are the
// ones of rounds 1-5's one-lane-per-row kernel; only where the operands come from has changed.
template <class P, class G>
__device__ __forceinline__ double row_error(const P &p, const G &g, int J, int procrustes) {
    double e = 0.0;
    if (!procrustes) {
        for (int j = 0; j < J; ++j) {
            const double dx = (double)p[3 * j] - g[3 * j], dy = (double)p[3 * j + 1] - g[3 * j + 1],
                         dz = (double)p[3 * j + 2] - g[3 * j + 2];
            e += sqrt(dx * dx + dy * dy + dz * dz);
        }
        return e / J;
    }
    // procrustes(A = gt, B = pred, scaling=True, reflection='best').Z  (transforms.py:42-127)
    double ab[3] = {0, 0, 0}, bb[3] = {0, 0, 0};
    for (int j = 0; j < J; ++j)
        for (int c = 0; c < 3; ++c) { ab[c] += g[3 * j + c]; bb[c] += (double)p[3 * j + c]; }
    for (int c = 0; c < 3; ++c) { ab[c] /= J; bb[c] /= J; }
    double ssA = 0, ssB = 0, M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < J; ++j) {
        double a0[3], b0[3];
        for (int c = 0; c < 3; ++c) { a0[c] = g[3 * j + c] - ab[c]; b0[c] = (double)p[3 * j + c] - bb[c]; }
        for (int c = 0; c < 3; ++c) { ssA += a0[c] * a0[c]; ssB += b0[c] * b0[c]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) M[r][c] += a0[r] * b0[c];   // A0^T B0 (un-normalised)
    }
    const double An = sqrt(ssA), Bn = sqrt(ssB);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r][c] /= (An * Bn);
    double R[3][3], st;
    polar_from_svd(M, R, st);
    const double scale = An * st / Bn;   // Z = A_norm * S_trace * (B0/B_norm) R + A_bar
    for (int j = 0; j < J; ++j) {
        double b0[3], z[3];
        for (int c = 0; c < 3; ++c) b0[c] = (double)p[3 * j + c] - bb[c];
        for (int c = 0; c < 3; ++c) z[c] = scale * (b0[0] * R[0][c] + b0[1] * R[1][c] + b0[2] * R[2][c]) + ab[c];
        const double dx = z[0] - g[3 * j], dy = z[1] - g[3 * j + 1], dz = z[2] - g[3 * j + 2];
        e += sqrt(dx * dx + dy * dy + dz * dz);
    }
    return e / J;
}

// Any joint count: one lane per row straight from global memory (rows are J*12 bytes apart: every load instruction touches 64 cache lines).
__global__ void row_error_kernel(const float *__restrict__ pred, const double *__restrict__ gt, int B, int N, int J,
                                 long long row_offset, int procrustes, double *__restrict__ err) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n = (int)((row_offset + b) % N);
    err[b] = row_error(pred + (size_t)b * J * 3, gt + (size_t)n * J * 3, J, procrustes);
}

// J = 17 (every dataset of the path), round 6: the 64 rows of a wave arrive as ONE contiguous 13 KB piece of the pose tensor, fetched with
// coalesced 16-byte loads (a lane's own row would be 204 bytes from its neighbour's: 64 cache lines per load instruction, 427 GB/s
// at 3.5 M rows), and the rows' ground-truth poses - consecutive poses n = (row_offset + b) mod N, 408 bytes each - with coalesced
// 8-byte loads; both land in the LDS ([row][51], odd row stride: a lane reading its own row is bank-conflict free), all loads of the
// tile in flight at once; then one lane per row as before.
constexpr int RE_ROWS = 64, RE_D = 17 * 3;
__global__ __launch_bounds__(RE_ROWS) void row_error17_kernel(const float *__restrict__ pred, const double *__restrict__ gt, int B, int N,
                                                               long long row_offset, int procrustes, double *__restrict__ err) {
    __shared__ __attribute__((aligned(16))) float sp[RE_ROWS * RE_D];
    __shared__ double sg[RE_ROWS * RE_D];
    const int tid = threadIdx.x, b0 = blockIdx.x * RE_ROWS;
    const int rows = min(RE_ROWS, B - b0);
    const int nf = rows * RE_D;                                   // floats of this tile (the last tile of a batch is short)
    const float *src = pred + (size_t)b0 * RE_D;                  // 16-byte aligned: b0 * 204 bytes, b0 a multiple of 64
    for (int c = tid; c * 4 + 3 < nf; c += RE_ROWS) *reinterpret_cast<f32x4 *>(sp + c * 4) = *reinterpret_cast<const f32x4 *>(src + c * 4);
    if (tid < (nf & 3)) sp[(nf & ~3) + tid] = src[(nf & ~3) + tid];
    // ground truth of row r: pose n_r = (row_offset + b0 + r) mod N - consecutive poses, wrapping to 0 behind N - 1 (several times in one
    // tile when N < 64): the pose index is wave-uniform and advances by scalar increment / compare, no division per element; lanes 0..50
    // fetch the pose's 51 doubles (408 contiguous bytes) per row
    int n = (int)((row_offset + b0) % N);
    if (tid < RE_D) {
#pragma unroll 16
        for (int r = 0; r < rows; ++r) {       // (unrolled: sixteen independent loads in flight per lane)
            sg[r * RE_D + tid] = gt[(size_t)n * RE_D + tid];
            n = (n + 1 == N) ? 0 : n + 1;
        }
    }
    __syncthreads();
    if (tid < rows) err[b0 + tid] = row_error(sp + tid * RE_D, sg + tid * RE_D, 17, procrustes);
}

// np.amin / np.argmin order: NaN is smaller than everything (a diverged hypothesis poisons the pose, and the first
// NaN index is reported), otherwise the smaller value, ties to the lower hypothesis index.
__device__ __forceinline__ bool min_takes(double ov, int oh, double v, int h) {
    if (oh < 0) return false;
    if (h < 0) return true;
    const bool on = ov != ov, vn = v != v;
    if (on || vn) return on && (!vn || oh < h);
#ifdef ZEDO_MUT_ARGMIN_TIE  // tools/mutation_check.py only: ties to the HIGHER hypothesis index
    return ov < v || (ov == v && oh > h);
#else
    return ov < v || (ov == v && oh < h);
#endif
}

// min / first arg-min (np.argmin tie rule) over the hypotheses held locally.  min_takes is a strict total order on (value, hypothesis):
// the minimum does not depend on the order in which the candidates are visited, so both kernels below return the same bits.
// Few poses (N below POSE_MIN_LANE_N): one WAVEFRONT per pose, lanes stride over the hypotheses (N * 8 bytes apart: one cache line per
// lane), butterfly reduction - the parallelism is across hypotheses, a pass is one load deep.
__global__ void pose_min_wave_kernel(const double *__restrict__ err, int B, int N, long long row_offset,
                                     double *__restrict__ best, int *__restrict__ best_h) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    // global rows of pose n: n, n+N, n+2N, ... ; local index = global - row_offset
    long long h0 = (row_offset - n + N - 1) / N;     // first hypothesis with h*N + n >= row_offset
    if (row_offset <= n) h0 = 0;
    double e = __builtin_huge_val();
    int hi = -1;
    for (long long h = h0 + lane;; h += 64) {
        const long long loc = h * N + n - row_offset;
        if (loc >= B) break;
        const double v = err[loc];
        if (min_takes(v, (int)h, e, hi)) { e = v; hi = (int)h; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oe = __shfl_xor(e, off);
        const int oh = __shfl_xor(hi, off);
        if (min_takes(oe, oh, e, hi)) { e = oe; hi = oh; }
    }
    if (lane == 0) { best[n] = (hi >= 0) ? e : __builtin_huge_val(); best_h[n] = hi; }
}

// Many poses (round 6): one LANE per pose, hypotheses walked in ascending order - the 64 lanes of a wave read 64 consecutive poses of one
// hypothesis, 512 contiguous bytes per load, where the wave-per-pose kernel touches H cache lines per pose.
constexpr int POSE_MIN_LANE_N = 8192;
__global__ void pose_min_kernel(const double *__restrict__ err, int B, int N, long long row_offset,
                                double *__restrict__ best, int *__restrict__ best_h) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    long long h0 = (row_offset - n + N - 1) / N;
    if (row_offset <= n) h0 = 0;
    double e = __builtin_huge_val();
    int hi = -1;
    for (long long h = h0;; ++h) {
        const long long loc = h * N + n - row_offset;
        if (loc >= B) break;
        const double v = err[loc];
        if (min_takes(v, (int)h, e, hi)) { e = v; hi = (int)h; }
    }
    best[n] = (hi >= 0) ? e : __builtin_huge_val();
    best_h[n] = hi;
}

// Many poses (N >= POSE_MIN_LANE_N), J = 17, round 6: the row errors POSE-MAJOR.  One wave per (64 consecutive poses, chunk of the local
// hypotheses): the poses' ground truth is staged once (26 KB, coalesced) and stays in the LDS while the wave walks its hypotheses in
// ascending order; the 64 rows of hypothesis h are 13 KB of contiguous pose tensor, fetched with 16-byte loads into registers one
// hypothesis AHEAD of the arithmetic and dropped into the LDS behind it (a lane then reads its own row at an odd word stride: conflict
// free).  Where the row-major kernel reads a pose's ground truth once per ROW (from the L2), this reads it once per chunk.  The hypotheses
// are cut into chunks only to have enough workgroups to balance over the chip (1 108 pose tiles alone are 1.08 rounds of 1 024 resident
// waves: the second round would run 8 % full); the arg-min over hypotheses follows in pose_min_kernel (28 MB of errors, coalesced).
// Same row_error statements as every other kernel of this file: the same bits
// (tests/test_hip_parity.py::test_selection_pose_major_kernel_is_bitwise_the_row_major_pair).
constexpr int SEL_CHUNKS = (RE_ROWS * RE_D + 3) / 4 + 1;          // 16-byte chunks that cover 64 rows at any 4-byte alignment of their first element
constexpr int SEL_T = (SEL_CHUNKS + RE_ROWS - 1) / RE_ROWS;       // per lane
__global__ __launch_bounds__(RE_ROWS) void row_error17_pose_major_kernel(const float *__restrict__ pred, const double *__restrict__ gt, int B, int N,
                                                                          long long row_offset, int procrustes, int h_per_chunk,
                                                                          double *__restrict__ err) {
    __shared__ double sg[RE_ROWS * RE_D];
    __shared__ __attribute__((aligned(16))) float sp[SEL_CHUNKS * 4];
    const int lane = threadIdx.x, n0 = blockIdx.x * RE_ROWS, n = n0 + lane;
    const int poses = min(RE_ROWS, N - n0);
    const long long total = (long long)B * RE_D;                   // floats of the local pose tensor
    const long long h_lo = row_offset / N + (long long)blockIdx.y * h_per_chunk;
    const long long h_hi = min((row_offset + B - 1) / N, h_lo + h_per_chunk - 1);
    if (h_lo > h_hi) return;
    for (int q = lane; q < poses * RE_D; q += RE_ROWS) sg[q] = gt[(size_t)n0 * RE_D + q];
    // chunk c of hypothesis h: floats [a + 4 c, a + 4 c + 4) of the tensor, a = the first element of the tile rounded down to a multiple of 4
    f32x4 nxt[SEL_T];
    auto fetch = [&](long long h) {
        const long long e0 = (h * N + n0 - row_offset) * RE_D, a = e0 & ~3LL;
#pragma unroll
        for (int t = 0; t < SEL_T; ++t) {
            const int c = lane + t * RE_ROWS;
            const long long lo = a + 4LL * c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < SEL_CHUNKS) {
                if (lo >= 0 && lo + 3 < total) v = *reinterpret_cast<const f32x4 *>(pred + lo);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (lo + e >= 0 && lo + e < total) v[e] = pred[lo + e];
            }
            nxt[t] = v;
        }
    };
    auto drop = [&]() {
#pragma unroll
        for (int t = 0; t < SEL_T; ++t) {
            const int c = lane + t * RE_ROWS;
            if (c < SEL_CHUNKS) *reinterpret_cast<f32x4 *>(sp + 4 * c) = nxt[t];
        }
    };
    fetch(h_lo);
    for (long long h = h_lo; h <= h_hi; ++h) {
        drop();                                                    // tile h: registers -> LDS (one wave: LDS operations execute in order)
        if (h < h_hi) fetch(h + 1);                                // tile h + 1 on its way while tile h is worked on
        const long long loc = h * N + n - row_offset;
        if (n < N && loc >= 0 && loc < B) {
            const int shift = (int)(((h * N + n0 - row_offset) * RE_D) & 3LL);
            err[loc] = row_error(sp + shift + lane * RE_D, sg + lane * RE_D, 17, procrustes);
        }
    }
}

hipError_t launch_pose_min(const double *err, int B, int N, long long row_offset, double *best, int *best_h, hipStream_t st) {
    if (N >= POSE_MIN_LANE_N) hipLaunchKernelGGL(pose_min_kernel, dim3((N + 127) / 128), dim3(128), 0, st, err, B, N, row_offset, best, best_h);
    else hipLaunchKernelGGL(pose_min_wave_kernel, dim3((N + 3) / 4), dim3(256), 0, st, err, B, N, row_offset, best, best_h);
    return hipGetLastError();
}

hipError_t launch_min_mpjpe(const float *pred, const double *gt, int B, int N, int J, long long row_offset,
                            int procrustes, double *err, double *best, int *best_h, hipStream_t st) {
    // (the staged kernels fetch the pose tensor with 16-byte loads: a row pointer that is not 16-byte aligned takes the generic kernel)
    const bool aligned = (reinterpret_cast<uintptr_t>(pred) & 15) == 0;
    if (J == 17 && aligned && N >= POSE_MIN_LANE_N) {              // many poses: the row errors pose-major
        const int tiles = (N + RE_ROWS - 1) / RE_ROWS;
        const long long h_local = (row_offset + B - 1) / N - row_offset / N + 1;              // hypotheses this shard touches
        // enough workgroups for ~8 rounds of the resident waves (4 per CU: 39 KB of LDS each), at least 2 hypotheses per chunk
        long long chunks = (8LL * 4 * num_cus() + tiles - 1) / tiles;
        chunks = std::max(1LL, std::min(chunks, (h_local + 1) / 2));
        const int per = (int)((h_local + chunks - 1) / chunks);
        chunks = (h_local + per - 1) / per;
        hipLaunchKernelGGL(row_error17_pose_major_kernel, dim3(tiles, (unsigned)chunks), dim3(RE_ROWS), 0, st, pred, gt, B, N, row_offset, procrustes, per, err);
    } else
    if (J == 17 && aligned)
        hipLaunchKernelGGL(row_error17_kernel, dim3((B + RE_ROWS - 1) / RE_ROWS), dim3(RE_ROWS), 0, st, pred, gt, B, N, row_offset, procrustes, err);
    else
        hipLaunchKernelGGL(row_error_kernel, dim3((B + 127) / 128), dim3(128), 0, st, pred, gt, B, N, J, row_offset, procrustes, err);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_pose_min(err, B, N, row_offset, best, best_h, st);
}

}  // namespace zedo
