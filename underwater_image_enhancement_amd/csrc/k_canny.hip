// cv2.Canny(gray, low, high) with aperture 3 and the L1 gradient norm (six_stadigy.py:150,
// enhancement_strategies.py:181), evaluated on a list of rectangular REGIONS of the gray plane.  compute_Q
// runs Canny on every quadrant of the quadtree block separately (each quadrant is its own image: Sobel
// replicates ITS border and the magnitude outside it is 0), so a region here is one quadrant; the standalone
// entry point uses one region per frame.
//
// Stages (all integer arithmetic, bit-exact by construction):
//   k_canny_gradnms  16x64 tiles through LDS: 3x3 Sobel with BORDER_REPLICATE at the region edge -> |dx|+|dy| and the
//                  direction class; non-maximum suppression with OpenCV's fixed-point tan(22.5 deg) -> map {weak, none, strong}
//   k_canny_union  8-connected components of (weak | strong) pixels: lock-free union-find on pixel indices
//                  (links always point to the smaller index; agent-scope atomics, so XCD placement is irrelevant)
//   k_canny_flat   pointer jumping: every candidate points at its root
//   k_canny_mark   roots that own a strong pixel are flagged
//   k_canny_emit   a candidate is an edge iff its root is flagged (= hysteresis); per-region edge counts
// Hysteresis is order independent (an edge pixel is a weak-or-strong pixel whose 8-connected component holds a
// strong one), so the component formulation equals OpenCV's stack-based flood fill.
#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

struct CannyBufs {
    uint16_t *magdir;  // |dx|+|dy| (<= 2040) | dir << 12, per pixel of the frame
    uint8_t *cmap;     // 0 weak, 1 none, 2 strong
    int32_t *label;    // union-find parent (index inside the frame) or -1
    uint8_t *flag;     // root owns a strong pixel
};

__device__ __forceinline__ bool region_px(const Region &r, int lp, int &y, int &x)
{
    if (lp >= r.rows * r.cols) return false;
    y = r.y0 + lp / r.cols;
    x = r.x0 + lp % r.cols;
    return true;
}

// Sobel + direction class + non-maximum suppression for one 16x64 tile of a region, staged through LDS:
// gray tile with a 2-pixel halo (replicated at the REGION border, like Sobel's BORDER_REPLICATE on the quadrant),
// magnitude tile with a 1-pixel halo (0 outside the region, like OpenCV's zero-padded magnitude rows/columns).
// Writes the map byte of every pixel, and label/flag only for candidates (nobody reads them elsewhere).
constexpr int kCT_H = 16, kCT_W = 64;

__global__ void __launch_bounds__(256) k_canny_gradnms(const uint8_t *__restrict__ gray, const Region *__restrict__ regs,
                                                       int H, int W, int tiles_x, int low, int high, CannyBufs bufs)
{
    __shared__ uint8_t sg[kCT_H + 4][kCT_W + 4];
    __shared__ uint16_t sm[kCT_H + 2][kCT_W + 2];
    const Region r = regs[blockIdx.y];
    const int ty0 = (blockIdx.x / tiles_x) * kCT_H, tx0 = (blockIdx.x % tiles_x) * kCT_W;  // tile origin inside the region
    if (ty0 >= r.rows || tx0 >= r.cols) return;
    const int tid = threadIdx.x;
    const size_t base = (size_t)r.img * H * W;
    const uint8_t *g = gray + base;
    for (int i = tid; i < (kCT_H + 4) * (kCT_W + 4); i += 256) {
        const int ly = i / (kCT_W + 4), lx = i % (kCT_W + 4);
        const int ry = min(max(ty0 + ly - 2, 0), r.rows - 1), rx = min(max(tx0 + lx - 2, 0), r.cols - 1);
        sg[ly][lx] = g[(size_t)(r.y0 + ry) * W + r.x0 + rx];
    }
    __syncthreads();
    for (int i = tid; i < (kCT_H + 2) * (kCT_W + 2); i += 256) {
        const int ly = i / (kCT_W + 2), lx = i % (kCT_W + 2);
        const int ry = ty0 + ly - 1, rx = tx0 + lx - 1;  // region coordinates of this magnitude sample
        uint16_t v = 0;
        if (ry >= 0 && ry < r.rows && rx >= 0 && rx < r.cols) {
            const int a = sg[ly][lx], b = sg[ly][lx + 1], c = sg[ly][lx + 2];
            const int d = sg[ly + 1][lx], f = sg[ly + 1][lx + 2];
            const int p = sg[ly + 2][lx], q = sg[ly + 2][lx + 1], s = sg[ly + 2][lx + 2];
            const int dx = (c - a) + 2 * (f - d) + (s - p);
            const int dy = (p - a) + 2 * (q - b) + (s - c);
            const int ax = abs(dx), ay = abs(dy) << 15;
            const int tg22x = ax * 13573;  // (int)(tan(22.5deg) * 2^15 + 0.5)
            int dir;
            if (ay < tg22x) dir = 0;
            else if (ay > tg22x + (ax << 16)) dir = 1;
            else dir = ((dx ^ dy) < 0) ? 3 : 2;
            v = (uint16_t)((abs(dx) + abs(dy)) | (dir << 12));
        }
        sm[ly][lx] = v;
    }
    __syncthreads();
    for (int i = tid; i < kCT_H * kCT_W; i += 256) {
        const int ly = i / kCT_W, lx = i % kCT_W;
        const int ry = ty0 + ly, rx = tx0 + lx;
        if (ry >= r.rows || rx >= r.cols) continue;
        const int v = sm[ly + 1][lx + 1];
        const int m = v & 0xfff, dir = v >> 12;
        auto M = [&](int dy, int dx) -> int { return sm[ly + 1 + dy][lx + 1 + dx] & 0xfff; };
        bool keep = false;
        if (m > low) {
            if (dir == 0) keep = m > M(0, -1) && m >= M(0, 1);
            else if (dir == 1) keep = m > M(-1, 0) && m >= M(1, 0);
            else if (dir == 2) keep = m > M(-1, -1) && m > M(1, 1);
            else keep = m > M(-1, 1) && m > M(1, -1);
        }
        const int p = (r.y0 + ry) * W + r.x0 + rx;
        bufs.cmap[base + p] = keep ? (m > high ? 2 : 0) : 1;
        if (keep) {
            bufs.label[base + p] = p;
            bufs.flag[base + p] = 0;
        }
    }
}

// The component kernels walk a region four pixels of a row per thread and skip groups without candidates
// (map byte 1), which is almost every group on smooth frames.
__device__ __forceinline__ bool region_quad(const Region &r, int lq, int &y, int &x, int &n)
{
    const int qpr = (r.cols + 3) / 4;
    if (lq >= r.rows * qpr) return false;
    y = r.y0 + lq / qpr;
    const int lx = (lq % qpr) * 4;
    x = r.x0 + lx;
    n = min(4, r.cols - lx);
    return true;
}

__device__ __forceinline__ int ld_label(const int32_t *L, int i)
{
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Root of i with path halving.  Parent links only ever decrease, so a stale read is still an ancestor.
__device__ int uf_find(int32_t *L, int i)
{
    for (;;) {
        const int p = ld_label(L, i);
        if (p == i) return i;
        const int gp = ld_label(L, p);
        if (gp == p) return p;
        atomicMin(L + i, gp);
        i = gp;
    }
}

__device__ void uf_union(int32_t *L, int a, int b)
{
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicCAS(L + a, a, b);  // link the larger root under the smaller one
        if (old == a) return;
        a = old;  // somebody re-parented `a` first: continue from its new parent
    }
}

__global__ void __launch_bounds__(256) k_canny_union(const Region *__restrict__ regs, int H, int W, CannyBufs bufs)
{
    const Region r = regs[blockIdx.y];
    int y, x0, n;
    if (!region_quad(r, blockIdx.x * 256 + threadIdx.x, y, x0, n)) return;
    const size_t base = (size_t)r.img * H * W;
    const uint8_t *cm = bufs.cmap + base;
    int32_t *L = bufs.label + base;
    const bool down = y + 1 < r.y0 + r.rows;
    for (int i = 0; i < n; ++i) {
        const int x = x0 + i, p = y * W + x;
        if (cm[p] == 1) continue;
        const bool right = x + 1 < r.x0 + r.cols, left = x - 1 >= r.x0;
        if (right && cm[p + 1] != 1) uf_union(L, p, p + 1);
        if (down) {
            if (left && cm[p + W - 1] != 1) uf_union(L, p, p + W - 1);
            if (cm[p + W] != 1) uf_union(L, p, p + W);
            if (right && cm[p + W + 1] != 1) uf_union(L, p, p + W + 1);
        }
    }
}

__global__ void __launch_bounds__(256) k_canny_flat(const Region *__restrict__ regs, int H, int W, CannyBufs bufs)
{
    const Region r = regs[blockIdx.y];
    int y, x0, n;
    if (!region_quad(r, blockIdx.x * 256 + threadIdx.x, y, x0, n)) return;
    const size_t base = (size_t)r.img * H * W;
    const uint8_t *cm = bufs.cmap + base;
    int32_t *L = bufs.label + base;
    for (int i = 0; i < n; ++i) {
        const int p = y * W + x0 + i;
        if (cm[p] == 1) continue;
        int root = p;
        for (;;) {
            const int q = ld_label(L, root);
            if (q == root) break;
            root = q;
        }
        if (root != p) atomicMin(L + p, root);
    }
}

__global__ void __launch_bounds__(256) k_canny_mark(const Region *__restrict__ regs, int H, int W, CannyBufs bufs)
{
    const Region r = regs[blockIdx.y];
    int y, x0, n;
    if (!region_quad(r, blockIdx.x * 256 + threadIdx.x, y, x0, n)) return;
    const size_t base = (size_t)r.img * H * W;
    const int32_t *L = bufs.label + base;
    for (int i = 0; i < n; ++i) {
        const int p = y * W + x0 + i;
        if (bufs.cmap[base + p] != 2) continue;
        int root = p;
        while (L[root] != root) root = L[root];
        bufs.flag[base + root] = 1;
    }
}

__global__ void __launch_bounds__(256) k_canny_emit(const Region *__restrict__ regs, int H, int W, CannyBufs bufs,
                                                    uint32_t *__restrict__ count, uint8_t *__restrict__ edges)
{
    const Region r = regs[blockIdx.y];
    int y, x0, n;
    uint32_t mine = 0;
    if (region_quad(r, blockIdx.x * 256 + threadIdx.x, y, x0, n)) {
        const size_t base = (size_t)r.img * H * W;
        const int32_t *L = bufs.label + base;
        for (int i = 0; i < n; ++i) {
            const int p = y * W + x0 + i;
            bool edge = false;
            if (bufs.cmap[base + p] != 1) {
                int root = p;
                while (L[root] != root) root = L[root];
                edge = bufs.flag[base + root] != 0;
            }
            if (edges) edges[base + p] = edge ? 255 : 0;
            mine += edge;
        }
    }
    if (count) {
        const uint32_t tot = wave_sum_u32(mine);
        if ((threadIdx.x & 63) == 0 && tot) atomicAdd(count + blockIdx.y, tot);
    }
}

__global__ void k_full_regions(Region *regs, int B, int H, int W)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) regs[b] = Region{b, 0, 0, H, W};
}

CannyBufs carve_canny(Carver &c, Shape s)
{
    CannyBufs b;
    const size_t n = (size_t)s.B * s.npx();
    b.magdir = c.take<uint16_t>(n);
    b.cmap = c.take<uint8_t>(n);
    b.label = c.take<int32_t>(n);
    b.flag = c.take<uint8_t>(n);
    return b;
}

}  // namespace

size_t canny_ws_bytes(Shape s)
{
    Carver c(nullptr);
    carve_canny(c, s);
    return c.total();
}

int launch_make_full_regions(Region *d_regions, Shape s, hipStream_t st)
{
    UWIE_LAUNCH(k_full_regions, dim3(cdiv(s.B, 64)), dim3(64), 0, st, d_regions, s.B, s.H, s.W);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_canny(const uint8_t *d_gray, Shape s, const Region *d_regions, int nreg, int max_rows, int max_cols, int low,
                 int high, uint32_t *d_count, uint8_t *d_edges, void *ws, hipStream_t st)
{
    Carver c(ws);
    CannyBufs bufs = carve_canny(c, s);
    const int tiles_x = cdiv(max_cols, kCT_W), tiles_y = cdiv(max_rows, kCT_H);
    const dim3 tgrid(tiles_x * tiles_y, nreg), block(256);
    const dim3 grid(cdiv((long long)max_rows * cdiv(max_cols, 4), 256), nreg);
    if (d_count) UWIE_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t) * nreg, st));
    UWIE_LAUNCH(k_canny_gradnms, tgrid, block, 0, st, d_gray, d_regions, s.H, s.W, tiles_x, low, high, bufs);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_canny_union, grid, block, 0, st, d_regions, s.H, s.W, bufs);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_canny_flat, grid, block, 0, st, d_regions, s.H, s.W, bufs);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_canny_mark, grid, block, 0, st, d_regions, s.H, s.W, bufs);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_canny_emit, grid, block, 0, st, d_regions, s.H, s.W, bufs, d_count, d_edges);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
