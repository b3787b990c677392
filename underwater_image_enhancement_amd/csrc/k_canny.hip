// cv2.Canny(gray, low, high) with aperture 3 and the L1 gradient norm (six_stadigy.py:150,
// enhancement_strategies.py:181), evaluated on a list of rectangular REGIONS of the gray plane.  compute_Q
// runs Canny on every quadrant of the quadtree block separately (each quadrant is its own image: Sobel
// replicates ITS border and the magnitude outside it is 0), so a region here is one quadrant; the standalone
// entry point uses one region per frame.
//
// Stages (all integer arithmetic, bit-exact by construction):
//   k_canny_gradnms  32x64 tiles: the gray tile (2-pixel halo, replicated at the REGION border like Sobel's
//                  BORDER_REPLICATE on the quadrant) goes through LDS once; a thread then owns 4x2 pixels and keeps its
//                  6x8 gray neighbourhood in registers: separable Sobel on packed 16-bit pairs (v_pk_*), |dx|+|dy| for
//                  the 4x6 magnitudes it needs (0 outside the region, like OpenCV's zero-padded magnitude rows/columns),
//                  direction class and non-maximum suppression with OpenCV's fixed-point tan(22.5 deg) for its 8 pixels.
//                  Writes the map byte {weak, none, strong} of every pixel and APPENDS the candidates (weak | strong) to
//                  a list (one atomic per wavefront).
//   k_canny_union  8-connected components of the candidates: lock-free union-find on pixel indices (links always point
//                  to the smaller index; agent-scope atomics, so XCD placement is irrelevant)
//   k_canny_flat   pointer jumping: every candidate points at its root
//   k_canny_mark   roots that own a strong pixel are flagged
//   k_canny_emit   a candidate is an edge iff its root is flagged (= hysteresis); per-region edge counts, edge map
// The component kernels walk the region's candidate list (grid-stride, length read on the device), not the frame:
// smooth frames have few candidates.
// Hysteresis is order independent (an edge pixel is a weak-or-strong pixel whose 8-connected component holds a
// strong one), so the component formulation equals OpenCV's stack-based flood fill.
#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

struct CannyBufs {
    uint8_t *cmap;     // 0 weak, 1 none, 2 strong
    int32_t *label;    // union-find parent (index inside the frame), candidates only
    uint8_t *flag;     // root owns a strong pixel, candidates only
    uint32_t *cand;    // candidate lists, one segment of `seg` entries per region: pixel index inside the frame
    uint32_t *ncand;   // their lengths
    size_t seg;
};

typedef short v2s __attribute__((ext_vector_type(2)));
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

constexpr int kCT_H = 32, kCT_W = 64;      // output tile
constexpr int kSG_H = kCT_H + 4, kSG_W = 72;  // gray tile: 2-pixel halo, rows padded to a multiple of 8 bytes

__device__ __forceinline__ v2s as_v2s(uint32_t u)
{
    union { uint32_t u; v2s v; } c;
    c.u = u;
    return c.v;
}
// bytes i and j of the 64-bit value hi:lo, zero-extended into the two halves of a dword
template <int I, int J>
__device__ __forceinline__ v2s byte_pair(uint32_t lo, uint32_t hi)
{
    return as_v2s(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)I | ((uint32_t)J << 16)));
}

__global__ void __launch_bounds__(256) k_canny_gradnms(const uint8_t *__restrict__ gray, const Region *__restrict__ regs,
                                                       int H, int W, int tiles_x, int low, int high, CannyBufs bufs)
{
    __shared__ __attribute__((aligned(8))) uint8_t sg[kSG_H][kSG_W];
    const Region r = regs[blockIdx.y];
    const int ty0 = (blockIdx.x / tiles_x) * kCT_H, tx0 = (blockIdx.x % tiles_x) * kCT_W;  // tile origin inside the region
    if (ty0 >= r.rows || tx0 >= r.cols) return;
    const int tid = threadIdx.x;
    const size_t base = (size_t)r.img * H * W;
    const uint8_t *g = gray + base;
    for (int i = tid; i < kSG_H * (kCT_W + 4); i += 256) {
        const int ly = i / (kCT_W + 4), lx = i % (kCT_W + 4);
        const int ry = min(max(ty0 + ly - 2, 0), r.rows - 1), rx = min(max(tx0 + lx - 2, 0), r.cols - 1);
        sg[ly][lx] = g[(size_t)(r.y0 + ry) * W + r.x0 + rx];
    }
    __syncthreads();
    const int cg = tid & 15, rp = tid >> 4;            // column group (4 pixels), row pair
    const int ry0 = ty0 + 2 * rp, rx0 = tx0 + 4 * cg;  // region coordinates of this thread's first pixel
    const bool inside = ry0 < r.rows && rx0 < r.cols;  // no early exit: the wavefront scan below needs every lane
    // separable Sobel on column pairs: pair p holds magnitude columns rx0-1+2p, rx0+2p
    v2s hd[6][3], vs[6][3];
#pragma unroll
    for (int gr = 0; gr < 6; ++gr) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(&sg[2 * rp + gr][4 * cg]);
        const uint32_t w0 = row[0], w1 = row[1];
        const v2s A0 = byte_pair<0, 1>(w0, w1), A1 = byte_pair<2, 3>(w0, w1), A2 = byte_pair<4, 5>(w0, w1),
                  A3 = byte_pair<6, 7>(w0, w1);
        const v2s B0 = byte_pair<1, 2>(w0, w1), B1 = byte_pair<3, 4>(w0, w1), B2 = byte_pair<5, 6>(w0, w1);
        hd[gr][0] = A1 - A0; hd[gr][1] = A2 - A1; hd[gr][2] = A3 - A2;
        vs[gr][0] = A0 + B0 + B0 + A1; vs[gr][1] = A1 + B1 + B1 + A2; vs[gr][2] = A2 + B2 + B2 + A3;
    }
    int mag[4][6], cdx[2][4], cdy[2][4];
#pragma unroll
    for (int mr = 0; mr < 4; ++mr) {
        const int my = ry0 - 1 + mr;
        const bool rowin = my >= 0 && my < r.rows;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            const v2s dx = hd[mr][pp] + hd[mr + 1][pp] + hd[mr + 1][pp] + hd[mr + 2][pp];
            const v2s dy = vs[mr + 2][pp] - vs[mr][pp];
            const v2s m = __builtin_elementwise_abs(dx) + __builtin_elementwise_abs(dy);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mc = 2 * pp + h, mx = rx0 - 1 + mc;
                mag[mr][mc] = (rowin && mx >= 0 && mx < r.cols) ? (int)(h ? m.y : m.x) : 0;
                if ((mr == 1 || mr == 2) && mc >= 1 && mc <= 4) {
                    cdx[mr - 1][mc - 1] = h ? dx.y : dx.x;
                    cdy[mr - 1][mc - 1] = h ? dy.y : dy.x;
                }
            }
        }
    }
    uint32_t ncand = 0, cls[2] = {0x01010101u, 0x01010101u};
    uint32_t cpix[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ry = ry0 + i, rx = rx0 + j;
            if (!inside || ry >= r.rows || rx >= r.cols) continue;
            const int m = mag[i + 1][j + 1];
            if (m <= low) continue;
            const int dx = cdx[i][j], dy = cdy[i][j];
            const int ax = abs(dx), ay = abs(dy) << 15;
            const int tg22x = ax * 13573;  // (int)(tan(22.5deg) * 2^15 + 0.5)
            bool keep;
            if (ay < tg22x) keep = m > mag[i + 1][j] && m >= mag[i + 1][j + 2];
            else if (ay > tg22x + (ax << 16)) keep = m > mag[i][j + 1] && m >= mag[i + 2][j + 1];
            else if ((dx ^ dy) < 0) keep = m > mag[i][j + 2] && m > mag[i + 2][j];
            else keep = m > mag[i][j] && m > mag[i + 2][j + 2];
            if (!keep) continue;
            const uint32_t c = m > high ? 2u : 0u;
            cls[i] = (cls[i] & ~(0xffu << (8 * j))) | (c << (8 * j));
            const int p = (r.y0 + ry) * W + r.x0 + rx;
            bufs.label[base + p] = p;
            bufs.flag[base + p] = 0;
            cpix[ncand++] = (uint32_t)p;
        }
    }
    // map bytes: one (unaligned) dword per row when the 4 pixels exist, else byte by byte
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!inside || ry0 + i >= r.rows) continue;
        uint8_t *dst = bufs.cmap + base + (size_t)(r.y0 + ry0 + i) * W + r.x0 + rx0;
        if (rx0 + 3 < r.cols) *reinterpret_cast<u32_unaligned *>(dst) = cls[i];
        else
            for (int j = 0; j < r.cols - rx0; ++j) dst[j] = (uint8_t)(cls[i] >> (8 * j));
    }
    // candidate list of this region: one global atomic per workgroup (a single list-wide counter serialises in L2)
    __shared__ uint32_t blk_n, blk_base;
    if (tid == 0) blk_n = 0;
    __syncthreads();
    const uint32_t incl = wave_incl_scan_u32(ncand), total = __shfl(incl, 63);
    uint32_t wbase = 0;
    if ((tid & 63) == 63 && total) wbase = atomicAdd(&blk_n, total);
    wbase = __shfl(wbase, 63);
    __syncthreads();
    if (tid == 0 && blk_n) blk_base = atomicAdd(bufs.ncand + blockIdx.y, blk_n);
    __syncthreads();
    if (ncand) {
        uint32_t *dst = bufs.cand + (size_t)blockIdx.y * bufs.seg + blk_base + wbase + (incl - ncand);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < (int)ncand) dst[k] = cpix[k];
    }
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

// grid-stride walk of the candidate list of region blockIdx.y
#define UWIE_FOR_CANDIDATES(p)                                                                          \
    const uint32_t n_cand = bufs.ncand[blockIdx.y];                                                     \
    const uint32_t *cand = bufs.cand + (size_t)blockIdx.y * bufs.seg;                                   \
    for (uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x; ci < n_cand; ci += gridDim.x * blockDim.x) \
        if (const int p = (int)cand[ci]; true)

__global__ void __launch_bounds__(256) k_canny_union(const Region *__restrict__ regs, int H, int W, CannyBufs bufs)
{
    const Region r = regs[blockIdx.y];
    const size_t base = (size_t)r.img * H * W;
    const uint8_t *cm = bufs.cmap + base;
    int32_t *L = bufs.label + base;
    UWIE_FOR_CANDIDATES(p)
    {
        const int y = p / W, x = p - y * W;
        const bool right = x + 1 < r.x0 + r.cols, left = x - 1 >= r.x0, down = y + 1 < r.y0 + r.rows;
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
    int32_t *L = bufs.label + (size_t)regs[blockIdx.y].img * H * W;
    UWIE_FOR_CANDIDATES(p)
    {
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
    const size_t base = (size_t)regs[blockIdx.y].img * H * W;
    const int32_t *L = bufs.label + base;
    UWIE_FOR_CANDIDATES(p)
    {
        if (bufs.cmap[base + p] != 2) continue;
        int root = p;
        while (L[root] != root) root = L[root];
        bufs.flag[base + root] = 1;
    }
}

__global__ void __launch_bounds__(256) k_canny_emit(const Region *__restrict__ regs, int H, int W, CannyBufs bufs,
                                                    uint32_t *__restrict__ count, uint8_t *__restrict__ edges)
{
    const size_t base = (size_t)regs[blockIdx.y].img * H * W;
    const int32_t *L = bufs.label + base;
    uint32_t mine = 0;
    UWIE_FOR_CANDIDATES(p)
    {
        int root = p;
        while (L[root] != root) root = L[root];
        if (bufs.flag[base + root]) {
            ++mine;
            if (edges) edges[base + p] = 255;
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

// The largest launches are one region per frame (H x W) and four quadrants per frame (ceil(H/2) x ceil(W/2) each).
size_t canny_list_entries(Shape s) { return (size_t)4 * s.B * ((s.H + 1) / 2) * ((s.W + 1) / 2); }

CannyBufs carve_canny(Carver &c, Shape s)
{
    CannyBufs b;
    const size_t n = (size_t)s.B * s.npx();
    b.cmap = c.take<uint8_t>(n);
    b.label = c.take<int32_t>(n);
    b.flag = c.take<uint8_t>(n);
    b.seg = 0;  // set per launch
    b.cand = c.take<uint32_t>(canny_list_entries(s));
    b.ncand = c.take<uint32_t>((size_t)4 * s.B);
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
    bufs.seg = (size_t)max_rows * max_cols;
    if (nreg > 4 * s.B || (size_t)nreg * bufs.seg > canny_list_entries(s)) {
        set_error("canny: %d regions of %d x %d exceed the candidate-list workspace", nreg, max_rows, max_cols);
        return UWIE_E_INVALID;
    }
    // list walkers: a few blocks per region (grid-stride), at most one thread per possible candidate
    const dim3 lgrid((unsigned)std::min<size_t>(cdiv(bufs.seg, (size_t)256), (size_t)std::max(8, 4096 / nreg)), nreg);
    if (d_count) UWIE_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t) * nreg, st));
    if (d_edges) UWIE_HIP_CHECK(hipMemsetAsync(d_edges, 0, (size_t)s.B * s.npx(), st));
    UWIE_HIP_CHECK(hipMemsetAsync(bufs.ncand, 0, sizeof(uint32_t) * nreg, st));
    UWIE_LAUNCH(k_canny_gradnms, tgrid, block, 0, st, d_gray, d_regions, s.H, s.W, tiles_x, low, high, bufs);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_canny_union, lgrid, block, 0, st, d_regions, s.H, s.W, bufs);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_canny_flat, lgrid, block, 0, st, d_regions, s.H, s.W, bufs);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_canny_mark, lgrid, block, 0, st, d_regions, s.H, s.W, bufs);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_canny_emit, lgrid, block, 0, st, d_regions, s.H, s.W, bufs, d_count, d_edges);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
