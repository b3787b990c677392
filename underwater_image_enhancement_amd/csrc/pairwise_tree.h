// NumPy's float32 summation order, shared by the quadtree statistics (k_airlight.hip) and the feature statistics
// (k_features.hip).  np.add.reduce walks its input in buffers of 8192 elements (np.getbufsize()); every buffer is summed
// by pairwise_sum -- n < 8: sequential; n <= 128: eight interleaved accumulators, combined as a 3-level tree, then the
// tail sequentially; n > 128: split at n2 = n/2 - (n/2)%8 and add the two halves -- and the buffer results are added to
// the running total one after the other.
#pragma once
#include "devutil.h"

namespace uwie {

constexpr int kNpChunk = 8192;     // NumPy's reduction buffer size
constexpr int kMaxLeaves = 192;    // a leaf of the pairwise recursion holds 64..128 elements: at most 127 per chunk
constexpr int kTreeLevels = 10;    // depth of that recursion for n < 8192 is at most 8

// LEAVES / LEVELS: 127 leaves and 7 levels are the most a chunk (< 8192 elements) needs; the defaults are generous
template <class T, int LEAVES = kMaxLeaves, int LEVELS = kTreeLevels>
struct PairwiseTreeT {  // LDS scratch of one wavefront
    uint16_t off[LEVELS + 1][LEAVES], len[LEVELS + 1][LEAVES], child[LEVELS][LEAVES];
    int cnt[LEVELS + 1];
    T val[2][3][LEAVES];
};
using PairwiseTree = PairwiseTreeT<float>;  // float32 data (the u8-derived frame); float64 images use PairwiseTreeT<double>

// WAVE_SYNC: several wavefronts of one workgroup each walk their own tree (no workgroup barrier: the tree is private to
// the wavefront and LDS executes a wavefront's accesses in order; only the compiler has to be told)
template <bool WAVE_SYNC>
__device__ __forceinline__ void pairwise_sync()
{
    if constexpr (WAVE_SYNC) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Enumerates the recursion's nodes level by level; returns the level that holds only leaves (its off/len/cnt entries
// describe them).  The caller then stores every leaf's sums into t.val[level & 1][c][leaf] and calls pairwise_combine.
template <class T, int LEAVES, int LEVELS, bool WAVE_SYNC>
__device__ int pairwise_build(int len, int lane, PairwiseTreeT<T, LEAVES, LEVELS> &t)
{
    int nlev = 0;
    if (lane == 0) { t.off[0][0] = 0; t.len[0][0] = (uint16_t)len; t.cnt[0] = 1; }
    pairwise_sync<WAVE_SYNC>();
    for (;;) {
        const int cnt = t.cnt[nlev];
        int carry = 0;
        bool any = false;
        for (int b0 = 0; b0 < cnt; b0 += 64) {
            const int i = b0 + lane;
            const int l = i < cnt ? t.len[nlev][i] : 0, off = i < cnt ? t.off[nlev][i] : 0;
            const bool split = l > 128;
            const uint32_t kids = i < cnt ? (split ? 2u : 1u) : 0u;
            const uint32_t incl = wave_incl_scan_u32(kids);
            const int pos = carry + (int)(incl - kids);
            if (i < cnt) {
                t.child[nlev][i] = (uint16_t)pos;
                if (split) {
                    int n2 = l / 2;
                    n2 -= n2 % 8;
                    t.off[nlev + 1][pos] = (uint16_t)off; t.len[nlev + 1][pos] = (uint16_t)n2;
                    t.off[nlev + 1][pos + 1] = (uint16_t)(off + n2); t.len[nlev + 1][pos + 1] = (uint16_t)(l - n2);
                } else {
                    t.off[nlev + 1][pos] = (uint16_t)off; t.len[nlev + 1][pos] = (uint16_t)l;
                }
            }
            carry += (int)__shfl(incl, 63);
            any = any || __any(split);
        }
        if (!any) break;  // level `nlev` holds only leaves
        if (lane == 0) t.cnt[nlev + 1] = carry;
        ++nlev;
        pairwise_sync<WAVE_SYNC>();
    }
    return nlev;
}

// Post-order combination of the leaf sums stored at level nlev; the caller synchronises between storing them and this.
template <class T, int LEAVES, int LEVELS, bool WAVE_SYNC>
__device__ void pairwise_combine(int nlev, int lane, PairwiseTreeT<T, LEAVES, LEVELS> &t, T res[3])
{
    for (int lv = nlev - 1; lv >= 0; --lv) {
        const int cnt = t.cnt[lv];
        for (int i = lane; i < cnt; i += 64) {
            const int ch = t.child[lv][i];
            const bool split = t.len[lv][i] > 128;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T a = t.val[(lv + 1) & 1][c][ch];
                t.val[lv & 1][c][i] = split ? a + t.val[(lv + 1) & 1][c][ch + 1] : a;
            }
        }
        pairwise_sync<WAVE_SYNC>();
    }
    res[0] = t.val[0][0][0]; res[1] = t.val[0][1][0]; res[2] = t.val[0][2][0];
}

// Three sums at once over a ragged chunk of `len` < 8192 elements, one wavefront (blockDim.x == 64, all lanes call).
// The tree is expanded level by level with every lane working (a node list per level, kept in left-to-right order: a
// split node is replaced by its two children in place), leaves are summed one per lane by `leaf(off, len, out3)`, and
// the sums are folded back level by level (value = left + right, as the recursion returns them).  Result in lane 0.
template <class T, class Leaf, int LEAVES = kMaxLeaves, int LEVELS = kTreeLevels, bool WAVE_SYNC = false>
__device__ void pairwise_ragged(int len, int lane, PairwiseTreeT<T, LEAVES, LEVELS> &t, Leaf leaf, T res[3])
{
    const int nlev = pairwise_build<T, LEAVES, LEVELS, WAVE_SYNC>(len, lane, t);
    const int nLeaf = t.cnt[nlev];
    for (int i = lane; i < nLeaf; i += 64) {
        T s[3];
        leaf(t.off[nlev][i], t.len[nlev][i], s);
        t.val[nlev & 1][0][i] = s[0]; t.val[nlev & 1][1][i] = s[1]; t.val[nlev & 1][2][i] = s[2];
    }
    pairwise_sync<WAVE_SYNC>();
    pairwise_combine<T, LEAVES, LEVELS, WAVE_SYNC>(nlev, lane, t, res);
}

// the same with the tree private to the calling wavefront (several wavefronts of a workgroup, one tree each)
template <class T, class Leaf, int LEAVES, int LEVELS>
__device__ void pairwise_ragged_wave(int len, int lane, PairwiseTreeT<T, LEAVES, LEVELS> &t, Leaf leaf, T res[3])
{
    pairwise_ragged<T, Leaf, LEAVES, LEVELS, true>(len, lane, t, leaf, res);
}

template <class T>
__device__ __forceinline__ T tree8(const T *r)
{
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

}  // namespace uwie
