// escat_knn_grid.hip -- kNN graph + lazy-walk P for ONE point cloud of any size (cfg3: 2 M points)
// (reference: data_processing/ellipsoid_data_classes.py:209-211 -- knn_graph + to_undirected on the whole cloud --
//  and get_P, data_processing/process_pyg_data.py:499-600).  escat_build.hip's builder keeps a graph's n x n adjacency
// bitmap in LDS and stops at ~1090 nodes; this one bins the points into a uniform grid in HBM instead:
//
//   G0  k_bbox           bounding box of the cloud (ordered-int atomics)
//   G1  k_grid_insert    per-cell linked lists (head[cell], next[point]) + the number of occupied cells; the host
//                        picks the cell edge h from that count (~4 points per occupied cell: right for a surface, a
//                        volume or a curve alike) and rebuilds, at most 4 times
//   G2  k_knn_grid<K>    one thread per point: rings of cells around its own, the k best (fp64 squared distance, ties
//                        to the smaller index -- the order of escat_build.hip's brute force, so both give the same
//                        lists) kept sorted in registers; stops when the k-th best lies strictly inside the scanned cube
//   G3  k_rev_count / k_scan_* / k_sym_fill_grid / k_row_sort_P
//                        symmetrise without a bitmap: a directed edge i -> j whose reverse is missing adds i to row j;
//                        rows = diagonal + own list + those, sorted; P = 1/2 (I + D^-1 A) with get_P's fp32 operations
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include "../../include/escat.h"
#include "escat_internal.h"

#define fail escat_fail
#define CHECK_LAUNCH(what)                                                              \
    do { hipError_t e_ = hipGetLastError();                                             \
         if (e_ != hipSuccess) return fail(ESCAT_ELAUNCH, "%s: %s", what, hipGetErrorString(e_)); } while (0)
#define GRID_KMAX 32

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

// ------------------------------------------------------------------ G0: bounding box
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static inline float ord2f(unsigned o) { const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; memcpy(&f, &u, 4); return f; }

__global__ void __launch_bounds__(256)
k_bbox(const float *__restrict__ pos, int64_t n, unsigned *__restrict__ box /* min xyz, max xyz (ordered ints) */)
{
    __shared__ unsigned smin[3][4], smax[3][4];
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const unsigned o = f2ord(pos[3 * i + a]); lo[a] = o < lo[a] ? o : lo[a]; hi[a] = o > hi[a] ? o : hi[a]; }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned l2 = __shfl_xor(lo[a], m, 64), h2 = __shfl_xor(hi[a], m, 64);
            lo[a] = l2 < lo[a] ? l2 : lo[a]; hi[a] = h2 > hi[a] ? h2 : hi[a];
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { smin[a][threadIdx.x >> 6] = lo[a]; smax[a][threadIdx.x >> 6] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        unsigned l = smin[a][0], h = smax[a][0];
        for (int w = 1; w < 4; ++w) { l = smin[a][w] < l ? smin[a][w] : l; h = smax[a][w] > h ? smax[a][w] : h; }
        atomicMin(&box[a], l); atomicMax(&box[3 + a], h);
    }
}

// ------------------------------------------------------------------ G1: cell lists
struct GridDesc { double ox, oy, oz, inv_h, h; int dx, dy, dz; };

__device__ __forceinline__ void cell_of(const GridDesc &g, float x, float y, float z, int &cx, int &cy, int &cz)
{
    cx = (int)floor(((double)x - g.ox) * g.inv_h); cy = (int)floor(((double)y - g.oy) * g.inv_h); cz = (int)floor(((double)z - g.oz) * g.inv_h);
    cx = cx < 0 ? 0 : cx >= g.dx ? g.dx - 1 : cx; cy = cy < 0 ? 0 : cy >= g.dy ? g.dy - 1 : cy; cz = cz < 0 ? 0 : cz >= g.dz ? g.dz - 1 : cz;
}

__global__ void __launch_bounds__(256)
k_grid_insert(const float *__restrict__ pos, int64_t n, GridDesc g, int *__restrict__ head, int *__restrict__ next, int *__restrict__ occupied)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cx, cy, cz;
    cell_of(g, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], cx, cy, cz);
    const int64_t cell = ((int64_t)cz * g.dy + cy) * g.dx + cx;
    const int old = atomicExch(&head[cell], (int)i);
    next[i] = old;
    if (old < 0) atomicAdd(occupied, 1);
}

// ------------------------------------------------------------------ G2: kNN over rings of cells
template <int K>
__global__ void __launch_bounds__(256)
k_knn_grid(const float *__restrict__ pos, int64_t n, int k, GridDesc g, const int *__restrict__ head, const int *__restrict__ next,
           int *__restrict__ knn)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double bd[K]; int bi[K];
#pragma unroll
    for (int s = 0; s < K; ++s) { bd[s] = 1e300; bi[s] = -1; }
    const float xf = pos[3 * i], yf = pos[3 * i + 1], zf = pos[3 * i + 2];
    const double xi = xf, yi = yf, zi = zf;
    int cx, cy, cz;
    cell_of(g, xf, yf, zf, cx, cy, cz);
    auto visit = [&](int x, int y, int z) {
        if (x < 0 || x >= g.dx || y < 0 || y >= g.dy || z < 0 || z >= g.dz) return;
        for (int j = head[((int64_t)z * g.dy + y) * g.dx + x]; j >= 0; j = next[j]) {
            if (j == (int)i) continue;
            const double dx = xi - (double)pos[3 * (int64_t)j], dy = yi - (double)pos[3 * (int64_t)j + 1], dz = zi - (double)pos[3 * (int64_t)j + 2];
            const double d = dx * dx + dy * dy + dz * dz;
            if (d < bd[K - 1] || (d == bd[K - 1] && j < bi[K - 1])) {
                bd[K - 1] = d; bi[K - 1] = j;
#pragma unroll
                for (int s = K - 1; s > 0; --s) {
                    if (bd[s] < bd[s - 1] || (bd[s] == bd[s - 1] && bi[s] < bi[s - 1])) {
                        const double td = bd[s]; bd[s] = bd[s - 1]; bd[s - 1] = td;
                        const int ti = bi[s]; bi[s] = bi[s - 1]; bi[s - 1] = ti;
                    }
                }
            }
        }
    };
    // the cube of rings 0..R covers the whole grid once R reaches the farthest face
    int far = cx > g.dx - 1 - cx ? cx : g.dx - 1 - cx;
    far = cy > far ? cy : far; far = g.dy - 1 - cy > far ? g.dy - 1 - cy : far;
    far = cz > far ? cz : far; far = g.dz - 1 - cz > far ? g.dz - 1 - cz : far;
    // bd[K-1] is the K-th best; the caller's k-th (k <= K) is bd[k-1] -- found by a scan, K is a small constant
    for (int R = 0; R <= far; ++R) {
        if (R == 0) visit(cx, cy, cz);
        else
            for (int dz = -R; dz <= R; ++dz)
                for (int dy = -R; dy <= R; ++dy) {
                    const bool face = dz == -R || dz == R || dy == -R || dy == R;
                    if (face) for (int dx = -R; dx <= R; ++dx) visit(cx + dx, cy + dy, cz + dz);
                    else { visit(cx - R, cy + dy, cz + dz); visit(cx + R, cy + dy, cz + dz); }
                }
        // every point not yet seen differs by more than R cells along some axis: farther than R*h
        double kth = 1e300;
#pragma unroll
        for (int s = 0; s < K; ++s) if (s == k - 1) kth = bi[s] >= 0 ? bd[s] : 1e300;
        const double bound = (double)R * g.h;
        if (kth < bound * bound * (1.0 - 1e-9)) break;
    }
#pragma unroll
    for (int s = 0; s < K; ++s)
        if (s < k) knn[i * k + s] = bi[s];
}

// ------------------------------------------------------------------ G3: symmetrise + P
// extra[j] += 1 for every directed edge i -> j whose reverse j -> i is not in j's list
__global__ void __launch_bounds__(256)
k_rev_count(const int *__restrict__ knn, int64_t n, int k, int *__restrict__ extra)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * k) return;
    const int64_t i = t / k;
    const int j = knn[t];
    if (j < 0) return;
    bool found = false;
    for (int s = 0; s < k; ++s) found = found || knn[(int64_t)j * k + s] == (int)i;
    if (!found) atomicAdd(&extra[j], 1);
}

// rowlen[i] = 1 + own list + extra[i]; chunk sums for the scan (1024 rows per workgroup)
__global__ void __launch_bounds__(1024)
k_scan_local(const int *__restrict__ knn, const int *__restrict__ extra, int64_t n, int k, int *__restrict__ excl, int *__restrict__ chunk_sum)
{
    __shared__ int sh[1024];
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    int len = 0;
    if (i < n) {
        int own = 0;
        for (int s = 0; s < k; ++s) own += knn[i * k + s] >= 0;
        len = 1 + own + extra[i];
    }
    sh[threadIdx.x] = len;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    if (i < n) excl[i] = sh[threadIdx.x] - len;
    if (threadIdx.x == 1023) chunk_sum[blockIdx.x] = sh[1023];
}

// single workgroup: exclusive scan of the chunk sums in place; total -> *nnz_out (-1 if it does not fit int32)
__global__ void __launch_bounds__(1024)
k_scan_chunks(int *__restrict__ chunk_sum, int nchunks, int *__restrict__ nnz_out)
{
    __shared__ long long part[1024];
    const int t = threadIdx.x, per = (nchunks + 1023) / 1024;
    const int b0 = t * per < nchunks ? t * per : nchunks, b1 = b0 + per < nchunks ? b0 + per : nchunks;
    long long s = 0;
    for (int b = b0; b < b1; ++b) s += chunk_sum[b];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long long run = t ? part[t - 1] : 0;
    for (int b = b0; b < b1; ++b) { const int c = chunk_sum[b]; chunk_sum[b] = run > 0x7fffffffLL ? 0x7fffffff : (int)run; run += c; }
    if (t == 1023) *nnz_out = part[1023] > 0x7fffffffLL ? -1 : (int)part[1023];
}

// rowptr, the diagonal and the own list of every row; cursor[i] = where the reverse-only entries of row i start
__global__ void __launch_bounds__(256)
k_sym_fill_grid(const int *__restrict__ knn, int64_t n, int k, const int *__restrict__ excl, const int *__restrict__ chunk_off,
                const int *__restrict__ nnz, int *__restrict__ rowptr, int *__restrict__ col, int *__restrict__ cursor, int64_t cap)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    if (i == n) { rowptr[n] = *nnz; return; }
    int e = excl[i] + chunk_off[i >> 10];
    rowptr[i] = e;
    if (e < cap) col[e] = (int)i;
    ++e;
    for (int s = 0; s < k; ++s) {
        const int j = knn[i * k + s];
        if (j < 0) continue;
        if (e < cap) col[e] = j;
        ++e;
    }
    cursor[i] = e;
}

__global__ void __launch_bounds__(256)
k_rev_fill(const int *__restrict__ knn, int64_t n, int k, int *__restrict__ cursor, int *__restrict__ col, int64_t cap)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * k) return;
    const int64_t i = t / k;
    const int j = knn[t];
    if (j < 0) return;
    bool found = false;
    for (int s = 0; s < k; ++s) found = found || knn[(int64_t)j * k + s] == (int)i;
    if (!found) { const int e = atomicAdd(&cursor[j], 1); if (e < cap) col[e] = (int)i; }
}

// sort every row's columns ascending (<= 2k+1 and a few more entries: insertion sort in place), then the values of
// get_P (:586-598): 0.5 on the diagonal, 0.5 * (1/deg) elsewhere
__global__ void __launch_bounds__(256)
k_row_sort_P(const int *__restrict__ rowptr, int64_t n, int *__restrict__ col, float *__restrict__ pval, int64_t cap)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    if (e1 > cap) return;
    for (int a = e0 + 1; a < e1; ++a) {
        const int v = col[a];
        int b = a - 1;
        while (b >= e0 && col[b] > v) { col[b + 1] = col[b]; --b; }
        col[b + 1] = v;
    }
    const float D = (float)(e1 - e0 - 1);
    const float dinv = D > 0.f ? 1.0f / D : 0.f;
    const float off_val = 0.5f * (dinv * 1.0f);
    for (int e = e0; e < e1; ++e) pval[e] = col[e] == (int)i ? 0.5f : off_val;
}

// ------------------------------------------------------------------ host
static int64_t grid_max_cells(int64_t n)
{
    int64_t c = 16 * n;
    if (c < ((int64_t)1 << 20)) c = (int64_t)1 << 20;
    if (c > ((int64_t)1 << 28)) c = (int64_t)1 << 28;
    return c;
}

extern "C" size_t escat_build_grid_workspace_bytes(int64_t n_nodes, int k)
{
    if (n_nodes < 1 || k < 1) return 0;
    // knn | extra | excl | cursor | next | chunk sums | head | box (6) + occupied (1) + pad
    const int64_t chunks = (n_nodes + 1023) / 1024;
    return (size_t)(n_nodes * k + 4 * n_nodes + chunks + grid_max_cells(n_nodes) + 16) * sizeof(int32_t);
}

extern "C" int escat_build_knn_P_grid(const float *pos, int64_t n_nodes, int k, int32_t *rowptr, int32_t *col, float *pval,
                                      int64_t cap, int32_t *nnz_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!pos || !rowptr || !col || !pval || !nnz_out) return fail(ESCAT_EINVAL, "NULL pointer");
    if (k < 1 || k > GRID_KMAX) return fail(ESCAT_ERANGE, "k=%d outside [1,%d]", k, GRID_KMAX);
    if (n_nodes < 1) return fail(ESCAT_EINVAL, "empty input");
    if (n_nodes * (2 * (int64_t)k + 1) >= (int64_t)1 << 31)      // the most nonzeros the symmetrised graph can have: int32 offsets
        return fail(ESCAT_ERANGE, "n_nodes * (2k + 1) >= 2^31");
    const size_t need = escat_build_grid_workspace_bytes(n_nodes, k);
    if (!workspace || workspace_bytes < need) return fail(ESCAT_EWORKSPACE, "workspace %zu B < %zu B needed", workspace_bytes, need);
    if (cap < n_nodes) return fail(ESCAT_EINVAL, "cap smaller than the number of nodes");
    hipStream_t st = S(stream);
    const int64_t n = n_nodes, chunks = (n + 1023) / 1024, max_cells = grid_max_cells(n);
    int32_t *knn = (int32_t *)workspace, *extra = knn + n * k, *excl = extra + n, *cursor = excl + n, *next = cursor + n,
            *chunk = next + n, *head = chunk + chunks, *misc = head + max_cells;
    unsigned *box = (unsigned *)misc; int *occupied = misc + 6;

    // G0: bounding box
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if (hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess) return fail(ESCAT_ELAUNCH, "hipMemcpyAsync(box)");
    const unsigned nblk = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_bbox, dim3(nblk < 2048 ? nblk : 2048), dim3(256), 0, st, pos, n, box);
    unsigned hbox[6];
    if (hipMemcpyAsync(hbox, box, sizeof(hbox), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(ESCAT_ELAUNCH, "bounding box: %s", hipGetErrorString(hipGetLastError()));
    double lo[3], ext[3];
    for (int a = 0; a < 3; ++a) {
        const float l = ord2f(hbox[a]), h = ord2f(hbox[3 + a]);
        if (!(l <= h) || isinf(l) || isinf(h)) return fail(ESCAT_EINVAL, "positions are not finite");
        lo[a] = l; ext[a] = (double)h - (double)l;
    }
    const double emax = fmax(ext[0], fmax(ext[1], ext[2]));

    // G1: pick the cell edge from the measured occupancy
    GridDesc g;
    auto set_h = [&](double h) {
        // smallest h' >= h whose grid fits max_cells
        for (;;) {
            int64_t d[3]; int64_t prod = 1;
            for (int a = 0; a < 3; ++a) { d[a] = (int64_t)floor(ext[a] / h) + 1; if (d[a] > (1 << 20)) d[a] = (int64_t)1 << 21; prod *= d[a]; if (prod > max_cells) break; }
            if (prod <= max_cells) { g.dx = (int)d[0]; g.dy = (int)d[1]; g.dz = (int)d[2]; break; }
            h *= 1.26;
        }
        g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2]; g.h = h; g.inv_h = 1.0 / h;
        return h;
    };
    double h = emax > 0.0 ? emax / fmax(1.0, cbrt((double)n / 4.0)) : 1.0;     // ~4 points per cell if they filled the box
    h = set_h(h);
    for (int it = 0; it < 4; ++it) {
        const int64_t cells = (int64_t)g.dx * g.dy * g.dz;
        if (hipMemsetAsync(head, 0xff, (size_t)cells * 4, st) != hipSuccess || hipMemsetAsync(occupied, 0, 4, st) != hipSuccess)
            return fail(ESCAT_ELAUNCH, "hipMemsetAsync(grid)");
        hipLaunchKernelGGL(k_grid_insert, dim3(nblk), dim3(256), 0, st, pos, n, g, head, next, occupied);
        int occ = 0;
        if (hipMemcpyAsync(&occ, occupied, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(ESCAT_ELAUNCH, "grid build: %s", hipGetErrorString(hipGetLastError()));
        if (occ < 1) return fail(ESCAT_ELAUNCH, "grid build found no occupied cell");
        const double ppc = (double)n / occ;
        if (it == 3 || emax == 0.0 || (ppc >= 2.0 && ppc <= 8.0)) break;
        // points per occupied cell scale like h^D, D = the cloud's intrinsic dimension (unknown: 2.5 splits surface / volume)
        const double got = set_h(h * pow(4.0 / ppc, 1.0 / 2.5));
        if (fabs(got - h) <= 1e-9 * h) break;          // the cell cap (or a single cell) does not let it move: lists stand
        h = got;
    }

    // G2
#define KNN(KK) hipLaunchKernelGGL((k_knn_grid<KK>), dim3(nblk), dim3(256), 0, st, pos, n, k, g, head, next, knn)
    if (k <= 4) KNN(4); else if (k <= 8) KNN(8); else if (k <= 16) KNN(16); else KNN(32);
#undef KNN
    // G3
    const unsigned eblk = (unsigned)((n * k + 255) / 256);
    if (hipMemsetAsync(extra, 0, (size_t)n * 4, st) != hipSuccess) return fail(ESCAT_ELAUNCH, "hipMemsetAsync(extra)");
    hipLaunchKernelGGL(k_rev_count, dim3(eblk), dim3(256), 0, st, knn, n, k, extra);
    hipLaunchKernelGGL(k_scan_local, dim3((unsigned)chunks), dim3(1024), 0, st, knn, extra, n, k, excl, chunk);
    hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, st, chunk, (int)chunks, nnz_out);
    hipLaunchKernelGGL(k_sym_fill_grid, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, knn, n, k, excl, chunk, nnz_out, rowptr, col, cursor, cap);
    hipLaunchKernelGGL(k_rev_fill, dim3(eblk), dim3(256), 0, st, knn, n, k, cursor, col, cap);
    hipLaunchKernelGGL(k_row_sort_P, dim3(nblk), dim3(256), 0, st, rowptr, n, col, pval, cap);
    CHECK_LAUNCH("escat_build_knn_P_grid");
    return ESCAT_OK;
}
