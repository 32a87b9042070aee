// Rotation of per-node 3-vectors by the node's local frame, for the factored form of the vector operator
// (Q = D Pt D^T, D = blockdiag(O_i); see OpCsrV3 in escat.hip and esc_gnn_amd/operators.py::FactoredVectorOperator;
// the frames are the O_i of /root/reference/data_processing/process_pyg_data.py:238-286):
//   Y[n][a][f] = sum_c O[n][a][c] * X[n][c][f]        (transpose: O[n][c][a])
// One thread per (node, f); HBM-bound: 36 B of frame + 24 F bytes of vectors per node.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/escat.h"
#include "escat_internal.h"

template <bool TR>
__global__ void __launch_bounds__(256)
k_rotate_frames(const float *__restrict__ O, const float *__restrict__ X, float *__restrict__ Y, int64_t n_nodes, int F)
{
    const int64_t total = n_nodes * F;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int64_t n = g / F; const int f = (int)(g - n * F);
        const float *o = O + n * 9, *x = X + n * 3 * F + f;
        const float x0 = x[0], x1 = x[F], x2 = x[2 * F];
        float *y = Y + n * 3 * F + f;
        if (TR) {
            y[0]     = fmaf(o[6], x2, fmaf(o[3], x1, o[0] * x0));
            y[F]     = fmaf(o[7], x2, fmaf(o[4], x1, o[1] * x0));
            y[2 * F] = fmaf(o[8], x2, fmaf(o[5], x1, o[2] * x0));
        } else {
            y[0]     = fmaf(o[2], x2, fmaf(o[1], x1, o[0] * x0));
            y[F]     = fmaf(o[5], x2, fmaf(o[4], x1, o[3] * x0));
            y[2 * F] = fmaf(o[8], x2, fmaf(o[7], x1, o[6] * x0));
        }
    }
}

extern "C" int escat_rotate_frames_f32(const float *frames, const float *X, float *Y, int64_t n_nodes, int F, int transpose,
                                       void *stream)
{
    if (n_nodes < 0 || F < 1 || F > 4096) return escat_fail(ESCAT_EINVAL, "bad n_nodes / F");
    if (n_nodes == 0) return ESCAT_OK;
    if (!frames || !X || !Y) return escat_fail(ESCAT_EINVAL, "NULL pointer");
    if (X == Y) return escat_fail(ESCAT_EINVAL, "in-place rotation is not supported");
    const int64_t total = n_nodes * F;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (transpose) hipLaunchKernelGGL(k_rotate_frames<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames, X, Y, n_nodes, F);
    else           hipLaunchKernelGGL(k_rotate_frames<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames, X, Y, n_nodes, F);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "k_rotate_frames: %s", hipGetErrorString(e));
    return ESCAT_OK;
}
