// FP64 MFMA GEMM with LDS-DMA staging and the hand-placed software pipeline of the sweep
// contraction (sweep_gemm.hip, variant 3), generalised to the products of the factor chain:
//   NT  C -= A B^T        Cholesky trailing update (SYRK, lower tiles only)
//   NN  C  = -+ A B       V = L^-1 recursion levels
//   TN  C  = A^T B        K^-1 = V^T V
// The register-staged engine (gemm_f64.hip) keeps one workgroup per CU only ~45 % busy (60 us for
// a 128x128x256 tile whose MFMAs take 27 us): most launches of the factor chain have fewer tiles
// than the GPU has workgroup slots, so nothing hides its per-slab global -> VGPR -> LDS round trip.
// Here each wave issues its eight 1-KiB DMA pieces of slab s+2 in the shadow of the MFMAs of slab s.
//
// Operand images in LDS (per slab of 16 k):
//   KC (k contiguous in memory: A of NN/NT, B of NT): [128 rows][8 pieces of 16 B], unpadded, bank
//      conflicts removed by an XOR swizzle on the SOURCE side (LDS piece p' of row r holds global
//      piece p' ^ ((r >> 1) & 7));
//   MC (m/n contiguous: B of NN/TN, A of TN): [16 k-rows][144 doubles] (128 + pad).
// Requirements (checked by gemm_dma_usable): M, N multiples of 128, every k-range a multiple of 32
// and >= 32, leading dimensions and offsets even (16-byte pieces).  Everything else takes the
// register-staged kernel.
#include "common.h"
#include <algorithm>

#include "gemm_dma_body.h"

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(GemmArgs g) {
    constexpr int A_SZ = !AT ? KC_DOUBLES : MC_DOUBLES, B_SZ = BT ? KC_DOUBLES : MC_DOUBLES;
    __shared__ __attribute__((aligned(16))) double smem[2 * (A_SZ + B_SZ)];
    int bx, tb, bz;
    if (!gemm_block_order(g, &bx, &tb, &bz)) return;
    gemm_dma_tile_body<AT, BT, EPI>(g, smem, bx, (int)blockIdx.y, bz, nullptr, tb);
}

// can this product go through the DMA kernel?  (host side; batch items are checked by the caller
// that builds them: see chol.hip)
bool gemm_dma_usable(const GemmArgs& g, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 32) return false;
    if (M % BM || N % BN || K % 32) return false;
    if ((g.lda | g.ldb | g.ldc) & 1) return false;
    // triangular k-ranges start / end on tile boundaries (multiples of 128); split-K works on slab pairs
    return true;
}

template <bool AT, bool BT>
static int gd_launch_epi(gpry_ctx* ctx, const GemmArgs& g, int epi, dim3 grid) {
    hipStream_t st = g.stream ? g.stream : ctx->stream;
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_dma_kernel<AT, BT, EPI_STORE>), grid, dim3(256), 0, st, g); break;
        case EPI_STORE_NEG: hipLaunchKernelGGL((gemm_dma_kernel<AT, BT, EPI_STORE_NEG>), grid, dim3(256), 0, st, g); break;
        case EPI_SUB: hipLaunchKernelGGL((gemm_dma_kernel<AT, BT, EPI_SUB>), grid, dim3(256), 0, st, g); break;
        default: return gpry_fail(ctx, -1, "gemm_dma: bad epilogue %d", epi);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// the product only (same grid as gemm_f64_launch); the caller runs the split-K reduce
int gemm_dma_launch_product(gpry_ctx* ctx, const GemmArgs& g0, bool a_trans, bool b_trans, int epi, dim3 grid) {
    GemmArgs g = g0;
    gemm_fill_batch(ctx, &g);
    grid.z = gemm_grid_z(g);
    if ((g.tile_map & 15) == TM_ROWMAJOR) {
        g.tile_map = TM_BALANCED;
        const int tm = g.M / BM, tn = g.N / BN;      // batched launches: the largest item
        if (g.lower_only && tm == tn && g.batch == nullptr) grid.x = (unsigned)(tm * (tm + 1) / 2);
    }
    if (!a_trans && !b_trans) return gd_launch_epi<false, false>(ctx, g, epi, grid);
    if (!a_trans && b_trans) return gd_launch_epi<false, true>(ctx, g, epi, grid);
    if (a_trans && !b_trans) return gd_launch_epi<true, false>(ctx, g, epi, grid);
    return gpry_fail(ctx, -1, "gemm_dma: the TT layout is not built");
}

// ------------------------------------------------------------------------------------
// Stream-K launches.  The triangular products of the factor chain (V = L^-1 levels, K^-1 = V^T V) have tiles
// whose k-ranges differ 16x to 32x, and at N <= 6144 fewer tiles than a few rounds of the GPU's 512 workgroup
// slots: with one workgroup per tile (or per equal share of a tile, the uniform split-K above) the launch lasts
// as long as its longest piece while most slots idle -- 460 us for K^-1 at N = 4096 whose MFMA work is 318 us
// on a full GPU.  Here the (tile, k) space of a launch -- tiles of an item in order of descending k-length, their
// slab pairs consecutive -- is cut into segments of equal length, one per workgroup; a workgroup walks the parts
// of its segment (the rest of one tile, whole tiles, the start of another).  Tiles that end up in one part are
// stored with the final epilogue, the others as partial slices that gemm_parts_reduce_kernel adds in k order.
// The segmentation of an item depends only on its own shape and on the segment length, which the caller fixes
// per product (not per launch): a launch over a subset of the items (pipelined factor chain) gives the same bits.
template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_dma_parts_kernel(GemmArgs g, const GemmPart* __restrict__ parts,
                                                                const int* __restrict__ first) {
    constexpr int A_SZ = !AT ? KC_DOUBLES : MC_DOUBLES, B_SZ = BT ? KC_DOUBLES : MC_DOUBLES;
    __shared__ __attribute__((aligned(16))) double smem[2 * (A_SZ + B_SZ)];
    const int p1 = first[blockIdx.x + 1];
    for (int p = first[blockIdx.x]; p < p1; p++) {
        const GemmPart pt = parts[p];
        gemm_dma_tile_body<AT, BT, EPI>(g, smem, 0, 0, 0, &pt, (int)blockIdx.z);      // grid.z = thetas of a batched launch
        // the body counts its own DMA pieces with s_waitcnt: nothing of this part may be in flight, and no wave may
        // still read the LDS images, when the next one starts
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// C = sign * (slice 0 + slice 1 + ...) for the tiles that were computed in several parts; 8 row slices per tile
__global__ __launch_bounds__(256) void gemm_parts_reduce_kernel(GemmArgs g, const GemmRedTile* __restrict__ tiles, double sign) {
    const int tb = (int)blockIdx.z;
    if (g.info != nullptr && *bset(g.info, tb, g.bstride) != 0) return;
    const GemmRedTile t = tiles[blockIdx.x];
    const double* split_buf = bset(g.split_buf, tb, g.bstride);
    double* Cb = bset(g.C, tb, g.bstride);
    int64_t coff = 0;
    if (g.batch != nullptr) coff = g.batch[t.bz].c_off;
    const int row0 = t.ti * BM, col0 = t.tj * BN;
    const int rbeg = (int)blockIdx.y * (BM / 8);
    for (int e = threadIdx.x; e < (BM / 8) * (BN / 2); e += 256) {
        const int rr = rbeg + e / (BN / 2), cc = (e % (BN / 2)) * 2;
        const int64_t off = coff + (int64_t)(row0 + rr) * g.ldc + col0 + cc;
        double2 acc = make_double2(0.0, 0.0);
        for (int sidx = 0; sidx < t.nslice; sidx++) {
            const double2 v = *reinterpret_cast<const double2*>(split_buf + (int64_t)sidx * g.split_stride + off);
            acc.x += v.x; acc.y += v.y;
        }
        *reinterpret_cast<double2*>(Cb + off) = make_double2(sign * acc.x, sign * acc.y);
    }
}

namespace {
struct HostTile { int ti, tj, pairs; };
// tiles of one item, longest k-range first (ties: row-major), with their slab-pair counts
void enumerate_tiles(int kmode, int lower_only, const GemmShape& it, std::vector<HostTile>& out) {
    out.clear();
    const int tm = it.M / BM, tn = it.N / BN;
    for (int ti = 0; ti < tm; ti++)
        for (int tj = 0; tj < tn; tj++) {
            if (lower_only && tj > ti) continue;
            const int row0 = ti * BM, col0 = tj * BN;
            int kbeg = 0, kend = it.K;
            if (kmode == KM_A_LOWER) kend = std::min(it.K, row0 + BM);
            else if (kmode == KM_B_LOWER) kbeg = std::min(it.K, col0);
            else if (kmode == KM_AT_LOWER_B_LOWER) kbeg = std::min(it.K, std::max(row0, col0));
            else if (kmode == KM_B_UPPER) kend = std::min(it.K, col0 + BN);
            else if (kmode == KM_AT_LOWER) kbeg = std::min(it.K, row0);
            out.push_back({ti, tj, (kend - kbeg) / (2 * BK)});
        }
    std::stable_sort(out.begin(), out.end(), [](const HostTile& a, const HostTile& b) { return a.pairs > b.pairs; });
}
}  // namespace

int gemm_parts_segment(int kmode, int lower_only, const std::vector<GemmShape>& shapes, int slots) {
    int64_t total = 0;
    std::vector<HostTile> tl;
    for (const GemmShape& it : shapes) {
        enumerate_tiles(kmode, lower_only, it, tl);
        for (const HostTile& t : tl) total += t.pairs;
    }
    int64_t seg = (total + slots - 1) / slots;
    if (seg < 4) seg = 4;       // below 128 k a part is mostly prologue and epilogue
    return (int)seg;
}

void gemm_parts_plan_free(GemmPartsPlan* pl) {
    if (pl->d_parts) (void)hipFree(pl->d_parts);
    if (pl->d_first) (void)hipFree(pl->d_first);
    if (pl->d_red) (void)hipFree(pl->d_red);
    *pl = GemmPartsPlan();
}

int gemm_parts_plan_build(gpry_ctx* ctx, int kmode, int lower_only, const std::vector<GemmShape>& items, int seg_pairs,
                          GemmPartsPlan* out) {
    *out = GemmPartsPlan();
    std::vector<GemmPart> parts;
    std::vector<int> first;
    std::vector<GemmRedTile> red;
    std::vector<HostTile> tl;
    int max_slices = 0;
    for (size_t z = 0; z < items.size(); z++) {
        if (items[z].M % BM || items[z].N % BN || items[z].K % (2 * BK))
            return gpry_fail(ctx, -1, "gemm parts: item %zu is not 128-aligned", z);
        enumerate_tiles(kmode, lower_only, items[z], tl);
        int room = 0;                      // slab pairs left in the current segment
        for (const HostTile& t : tl) {
            if (t.pairs == 0) {            // an empty k-range still has to store its zeros
                if (room == 0) { first.push_back((int)parts.size()); room = seg_pairs; }
                parts.push_back({(int)z, t.ti, t.tj, 0, 0, -1});
                continue;
            }
            int done = 0, nslice = 0;
            const size_t p0 = parts.size();
            while (done < t.pairs) {
                if (room == 0) { first.push_back((int)parts.size()); room = seg_pairs; }
                const int take = std::min(room, t.pairs - done);
                parts.push_back({(int)z, t.ti, t.tj, done, done + take, nslice});
                done += take; room -= take; nslice++;
            }
            if (nslice == 1) parts[p0].slice = -1;
            else { red.push_back({(int)z, t.ti, t.tj, nslice}); if (nslice > max_slices) max_slices = nslice; }
        }
        // the next item starts a segment of its own (its segmentation must not depend on its neighbours)
        room = 0;
    }
    first.push_back((int)parts.size());
    out->nwg = (int)first.size() - 1;
    out->nred = (int)red.size();
    out->max_slices = max_slices;
    if (out->nwg == 0) return 0;
    hipError_t e = hipMalloc((void**)&out->d_parts, parts.size() * sizeof(GemmPart));
    if (e == hipSuccess) e = hipMalloc((void**)&out->d_first, first.size() * sizeof(int));
    if (e == hipSuccess && !red.empty()) e = hipMalloc((void**)&out->d_red, red.size() * sizeof(GemmRedTile));
    if (e == hipSuccess) e = hipMemcpy(out->d_parts, parts.data(), parts.size() * sizeof(GemmPart), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(out->d_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess && !red.empty()) e = hipMemcpy(out->d_red, red.data(), red.size() * sizeof(GemmRedTile), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        gemm_parts_plan_free(out);
        return gpry_fail(ctx, -2, "gemm parts plan: %s", hipGetErrorString(e));
    }
    return 0;
}

template <bool AT, bool BT>
static int gd_parts_launch_epi(gpry_ctx* ctx, const GemmArgs& g, int epi, const GemmPartsPlan& pl, hipStream_t st) {
    const dim3 grid((unsigned)pl.nwg, 1, (unsigned)(g.bn > 1 ? g.bn : 1));
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_dma_parts_kernel<AT, BT, EPI_STORE>), grid, dim3(256), 0, st, g, pl.d_parts, pl.d_first); break;
        case EPI_STORE_NEG: hipLaunchKernelGGL((gemm_dma_parts_kernel<AT, BT, EPI_STORE_NEG>), grid, dim3(256), 0, st, g, pl.d_parts, pl.d_first); break;
        default: return gpry_fail(ctx, -1, "gemm parts: store epilogues only (got %d)", epi);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// g: the product as for gemm_f64_launch (batch pointer = the device items the plan was built for); the slices go to
// the context's split-K scratch, `slice_stride` doubles apart (>= the extent of C)
int gemm_dma_parts_launch(gpry_ctx* ctx, const GemmArgs& g0, bool a_trans, bool b_trans, int epi, const GemmPartsPlan& pl,
                          int64_t slice_stride) {
    if (pl.nwg == 0) return 0;
    GemmArgs g = g0;
    gemm_fill_batch(ctx, &g);
    g.nsplit = 1; g.split_buf = nullptr; g.split_stride = slice_stride;
    if (pl.max_slices > 0) GPRY_TRY(gemm_split_scratch(ctx, pl.max_slices, slice_stride, &g.split_buf));
    hipStream_t st = g.stream ? g.stream : ctx->stream;
    int rc;
    if (!a_trans && !b_trans) rc = gd_parts_launch_epi<false, false>(ctx, g, epi, pl, st);
    else if (!a_trans && b_trans) rc = gd_parts_launch_epi<false, true>(ctx, g, epi, pl, st);
    else if (a_trans && !b_trans) rc = gd_parts_launch_epi<true, false>(ctx, g, epi, pl, st);
    else return gpry_fail(ctx, -1, "gemm parts: the TT layout is not built");
    if (rc || pl.nred == 0) return rc;
    hipLaunchKernelGGL(gemm_parts_reduce_kernel, dim3((unsigned)pl.nred, 8, (unsigned)(g.bn > 1 ? g.bn : 1)), dim3(256), 0, st, g, pl.d_red,
                       epi == EPI_STORE_NEG ? -1.0 : 1.0);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
