// Device-wide primitives of the DenseCRF lattice build (crf.hip), hand-written for gfx950: a STABLE least-significant-digit
// radix sort of (64-bit key, 32-bit value) pairs and an int32 prefix sum.  Stability is part of the numerical contract: the
// contributor list of a lattice point must stay in ascending pixel order so that the splat adds in the order of the sequential
// CPU algorithm (oracle/densecrf_ref.c) -- a sort that is only "a" sort changes Q in the last bit.
//
// Sort: 8-bit digits, one pass per digit over [begin_bit, end_bit), optionally segmented (items of a segment stay in its range:
// the images of a batch -- the image index never goes through a pass); per pass three launches --
//   count   : every workgroup histograms the digit of its tile of 4096 items (one LDS atomic per item) -> hist[segment][digit][tile]
//   scan    : exclusive prefix sum over hist in (segment, digit, tile) order = the first output slot of each run
//   scatter : the workgroup re-reads its tile; every wave ranks a contiguous quarter of it, 64 items per round, among the equal
//             digits before them (eight ballots; the wave's running digit counts in LDS -- no workgroup barrier inside the 16
//             rounds), the wave totals give every wave its first slot per digit, the tile is laid out sorted by digit in LDS and
//             written out run by run -- consecutive lanes write consecutive addresses, ~16 items (128 B of keys) per run
// HBM traffic per pass and item: keys read twice, value read once, both written once (36 B); 24 M entries x 8 passes ~ 7 GB.
// The passes ping-pong between the caller's input and output arrays (the input is destroyed); an even pass count ends with
// one copy.
//
// Scan: tiles of 8192 items (256 threads x 32), three phases -- tile sums, the scan of the tile sums (recursively; a single
// workgroup below 8192 items), tile scan with its offset.
#include <stdint.h>

#include "common.h"
#include "kernels.h"

namespace pnp {

constexpr int kSortThreads = 256, kSortRounds = 16, kSortTile = kSortThreads * kSortRounds;   // 4096 items per workgroup
constexpr int kScanThreads = 256, kScanIpt = 32, kScanTile = kScanThreads * kScanIpt;       // 8192 items per workgroup
constexpr int kSortMaxSegs = 64;

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over a 256-thread workgroup; returns the exclusive prefix, *total = workgroup sum
__device__ __forceinline__ int block_excl_scan(int v, int* sh /* >= 4 ints */, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = wave_incl_scan(v, lane);
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; w++) {
        const int s = sh[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---- scan ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(const int* __restrict__ in, size_t n, int* __restrict__ sums) {
    __shared__ int sh[4];
    const size_t t0 = (size_t)blockIdx.x * kScanTile;
    int acc = 0;
#pragma unroll 4
    for (int i = 0; i < kScanIpt; i++) {
        const size_t idx = t0 + (size_t)i * kScanThreads + threadIdx.x;
        if (idx < n) acc += in[idx];
    }
    int tot;
    (void)block_excl_scan(acc, sh, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// one tile: out = scan(in) + offset of the tile (offsets == nullptr: a single tile, offset 0).  A thread owns kScanIpt
// CONSECUTIVE items (four 16-byte loads when the tile is full); the order of the additions does not matter for integers.
template <bool INCLUSIVE>
__global__ __launch_bounds__(kScanThreads) void scan_tile_kernel(const int* __restrict__ in, size_t n, const int* __restrict__ offsets,
                                                                 int* __restrict__ out, int aligned16) {
    __shared__ int sh[4];
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const size_t t0 = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanIpt;
    const bool vec = aligned16 && t0 + kScanIpt <= n;          // 16-byte accesses on full, aligned runs
    int v[kScanIpt];
    int acc = 0;
    if (vec) {
#pragma unroll
        for (int i = 0; i < kScanIpt / 4; i++) {
            const i32x4 q = reinterpret_cast<const i32x4*>(in + t0)[i];
#pragma unroll
            for (int e = 0; e < 4; e++) v[i * 4 + e] = q[e];
        }
    } else {
#pragma unroll
        for (int i = 0; i < kScanIpt; i++) v[i] = t0 + i < n ? in[t0 + i] : 0;
    }
#pragma unroll
    for (int i = 0; i < kScanIpt; i++) acc += v[i];
    int tot;
    int run = block_excl_scan(acc, sh, &tot) + (offsets ? offsets[blockIdx.x] : 0);
#pragma unroll
    for (int i = 0; i < kScanIpt; i++) {
        const int x = v[i];
        v[i] = INCLUSIVE ? run + x : run;
        run += x;
    }
    if (vec) {
#pragma unroll
        for (int i = 0; i < kScanIpt / 4; i++)
            reinterpret_cast<i32x4*>(out + t0)[i] = i32x4{v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
    } else {
#pragma unroll
        for (int i = 0; i < kScanIpt; i++)
            if (t0 + i < n) out[t0 + i] = v[i];
    }
}

static size_t scan_temp_ints(size_t n) {                   // tile sums of every level
    size_t tot = 0;
    while (n > (size_t)kScanTile) {
        n = (n + kScanTile - 1) / kScanTile;
        tot += (n + 63) / 64 * 64;
    }
    return tot;
}

static int scan_rec(const int* in, int* out, size_t n, bool inclusive, int* temp, hipStream_t s) {
    const int al = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (n <= (size_t)kScanTile) {
        if (inclusive) hipLaunchKernelGGL((scan_tile_kernel<true>), dim3(1), dim3(kScanThreads), 0, s, in, n, (const int*)nullptr, out, al);
        else hipLaunchKernelGGL((scan_tile_kernel<false>), dim3(1), dim3(kScanThreads), 0, s, in, n, (const int*)nullptr, out, al);
        return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
    }
    const size_t nt = (n + kScanTile - 1) / kScanTile;
    int* sums = temp;
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nt), dim3(kScanThreads), 0, s, in, n, sums);
    const int r = scan_rec(sums, sums, nt, false, temp + (nt + 63) / 64 * 64, s);      // in place: tile offsets
    if (r != PNP_OK) return r;
    if (inclusive) hipLaunchKernelGGL((scan_tile_kernel<true>), dim3((unsigned)nt), dim3(kScanThreads), 0, s, in, n, (const int*)sums, out, al);
    else hipLaunchKernelGGL((scan_tile_kernel<false>), dim3((unsigned)nt), dim3(kScanThreads), 0, s, in, n, (const int*)sums, out, al);
    return hipGetLastError() == hipSuccess ? PNP_OK : PNP_ERR_HIP;
}

// ---- radix sort ------------------------------------------------------------------------------------------------------
// Segments: items [off[s], off[s+1]) are sorted among themselves and stay where they are (the images of a batch: their
// entries are generated image by image, so the image index never has to go through a radix pass).  Tiles do not straddle
// segments; the histogram is laid out (segment, digit, tile of the segment), whose prefix sum IS the segmented output order.
struct SortSegs {
    int nseg;
    unsigned off[kSortMaxSegs + 1];
    int tile_off[kSortMaxSegs + 1];                 // first tile of each segment; tile_off[nseg] = all tiles
};
// tile `blk` -> its items [t0, t0 + tile_n) and the index of its digit-0 histogram cell / the cell stride of a digit
__device__ __forceinline__ void sort_tile(const SortSegs& sg, int blk, unsigned& t0, int& tile_n, int& h0, int& hstride) {
    int sgi = 0;
    while (sgi + 1 < sg.nseg && blk >= sg.tile_off[sgi + 1]) sgi++;
    const int tis = blk - sg.tile_off[sgi];
    hstride = sg.tile_off[sgi + 1] - sg.tile_off[sgi];
    h0 = 256 * sg.tile_off[sgi] + tis;
    t0 = sg.off[sgi] + (unsigned)tis * kSortTile;
    const unsigned left = sg.off[sgi + 1] - t0;
    tile_n = (int)(left < (unsigned)kSortTile ? left : (unsigned)kSortTile);
}

__global__ __launch_bounds__(kSortThreads) void sort_count_kernel(const uint64_t* __restrict__ keys, const SortSegs sg, int shift,
                                                                  int dmask, int* __restrict__ hist) {
    __shared__ int cnt[4][256];                     // one histogram per wave (a quarter of the atomic traffic per address)
#pragma unroll
    for (int w = 0; w < 4; w++) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    unsigned t0;
    int tile_n, h0, hstride;
    sort_tile(sg, blockIdx.x, t0, tile_n, h0, hstride);
    int* const mine = cnt[threadIdx.x >> 6];
    uint64_t k[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {         // all loads first, then the atomics
        const int i = r * kSortThreads + threadIdx.x;
        k[r] = i < tile_n ? keys[t0 + i] : 0;
    }
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int i = r * kSortThreads + threadIdx.x;
        if (i < tile_n) atomicAdd(&mine[(int)(k[r] >> shift) & dmask], 1);
    }
    __syncthreads();
    hist[(size_t)h0 + (size_t)threadIdx.x * hstride] =
        cnt[0][threadIdx.x] + cnt[1][threadIdx.x] + cnt[2][threadIdx.x] + cnt[3][threadIdx.x];       // (segment, digit, tile): the scan order of the pass
}

__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                    const SortSegs sg, int shift, int dmask, const int* __restrict__ hist_scanned,
                                                                    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint64_t skey[kSortTile];            // 32 KB: the tile sorted by digit
    __shared__ uint32_t sval[kSortTile];            // 16 KB
    __shared__ int wrun[4][256];                    // per wave: items of each digit seen so far (after the ranking pass: the wave's totals,
                                                    // then the wave's first slot of each digit in the sorted tile)
    __shared__ int gbase[256];                      // output slot of the sorted tile's slot j of digit d: gbase[d] + j
    __shared__ int sh[4];
    constexpr int kWaveItems = kSortTile / 4;       // a wave ranks a CONTIGUOUS quarter of the tile, 64 items per round: item order =
                                                    // (wave, round, lane), so the four waves rank without a workgroup barrier per round
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned t0;
    int tile_n, h0, hstride;
    sort_tile(sg, blockIdx.x, t0, tile_n, h0, hstride);
#pragma unroll
    for (int w = 0; w < 4; w++) wrun[w][tid] = 0;
    uint64_t k[kSortRounds];
    uint32_t v[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int i = wave * kWaveItems + r * 64 + lane;
        k[r] = 0;
        v[r] = 0;
        if (i < tile_n) {
            k[r] = keys[t0 + i];
            v[r] = vals[t0 + i];
        }
    }
    __syncthreads();

    // ranking pass: rank[r] = items of the same digit before this one IN THE WAVE's quarter
    int rank[kSortRounds];
    int* const myrun = wrun[wave];
    const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int i = wave * kWaveItems + r * 64 + lane;
        const bool on = i < tile_n;
        const int d = on ? (int)(k[r] >> shift) & dmask : 0;
        uint64_t peers = __ballot(on);              // lanes of this wave holding the same digit
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t m = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        const int before = __popcll(peers & lt);
        const int seen = myrun[d];                  // the wave's LDS accesses complete in program order: read, then the leaders' update
        rank[r] = seen + before;
        __builtin_amdgcn_wave_barrier();
        if (on && before == 0) myrun[d] = seen + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // thread d: totals of digit d -> first slot of the digit in the sorted tile, of each wave's run inside it, and in the output
        const int c0 = wrun[0][tid], c1 = wrun[1][tid], c2 = wrun[2][tid], c3 = wrun[3][tid];
        int tot;
        const int ex = block_excl_scan(c0 + c1 + c2 + c3, sh, &tot);
        wrun[0][tid] = ex;
        wrun[1][tid] = ex + c0;
        wrun[2][tid] = ex + c0 + c1;
        wrun[3][tid] = ex + c0 + c1 + c2;
        gbase[tid] = hist_scanned[(size_t)h0 + (size_t)tid * hstride] - ex;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRounds; r++) {
        const int i = wave * kWaveItems + r * 64 + lane;
        if (i < tile_n) {
            const int slot = myrun[(int)(k[r] >> shift) & dmask] + rank[r];
            skey[slot] = k[r];
            sval[slot] = v[r];
        }
    }
    __syncthreads();
    // the sorted tile leaves run by run: slot j belongs to digit (key >> shift) & dmask, its output slot is gbase[digit] + j
#pragma unroll 4
    for (int r = 0; r < kSortRounds; r++) {
        const int j = r * kSortThreads + tid;
        if (j < tile_n) {
            const uint64_t kk = skey[j];
            const size_t g = (size_t)(gbase[(int)(kk >> shift) & dmask] + j);
            keys_out[g] = kk;
            vals_out[g] = sval[j];
        }
    }
}

size_t sort_temp_bytes(size_t n) {
    const size_t ntiles = (n + kSortTile - 1) / kSortTile + kSortMaxSegs;      // (a segmented sort rounds every segment up to a tile)
    const size_t hist = (256 * ntiles + 63) / 64 * 64;
    const size_t a = hist + scan_temp_ints(256 * ntiles);       // sort: histogram + its scan's tile sums
    const size_t b = scan_temp_ints(n);                         // a scan over n items
    return ((a > b ? a : b) + 64) * sizeof(int);
}

int device_scan_i32(const int* in, int* out, size_t n, bool inclusive, void* temp, size_t temp_bytes, hipStream_t s) {
    if (!n) return PNP_OK;
    if (n >= ((size_t)1 << 31) || scan_temp_ints(n) * sizeof(int) > temp_bytes) return PNP_ERR_ARG;
    return scan_rec(in, out, n, inclusive, reinterpret_cast<int*>(temp), s);
}

// Sorts (keys_in, vals_in)[0, n) by the key bits [begin_bit, end_bit) into (keys_out, vals_out); stable; destroys the inputs.
// seg_off (host array of nseg + 1 ascending item offsets, seg_off[0] = 0, seg_off[nseg] = n; nullptr = one segment): the items
// of a segment are sorted among themselves and stay in the segment's range.
int radix_sort_pairs(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, size_t n, int begin_bit, int end_bit,
                     const size_t* seg_off, int nseg, void* temp, size_t temp_bytes, hipStream_t s) {
    if (!n) return PNP_OK;
    if (n >= ((size_t)1 << 31) || end_bit <= begin_bit || end_bit > 64 || sort_temp_bytes(n) > temp_bytes) return PNP_ERR_ARG;
    SortSegs sg;
    if (!seg_off) nseg = 1;
    if (nseg < 1 || nseg > kSortMaxSegs) return PNP_ERR_ARG;
    int ntiles = 0, ns = 0;
    for (int i = 0; i < nseg; i++) {
        const size_t a = seg_off ? seg_off[i] : 0, b = seg_off ? seg_off[i + 1] : n;
        if (b < a || b > n || (i == 0 && a != 0)) return PNP_ERR_ARG;
        if (b == a) continue;                                   // empty segments own no tiles
        sg.off[ns] = (unsigned)a;
        sg.tile_off[ns] = ntiles;
        ntiles += (int)((b - a + kSortTile - 1) / kSortTile);
        sg.off[++ns] = (unsigned)b;
    }
    if (seg_off && seg_off[nseg] != n) return PNP_ERR_ARG;
    sg.nseg = ns;
    sg.tile_off[ns] = ntiles;
    const size_t nh = (size_t)256 * ntiles;
    int* hist = reinterpret_cast<int*>(temp);
    int* scan_tmp = hist + (nh + 63) / 64 * 64;
    uint64_t* kin = keys_in;
    uint64_t* kout = keys_out;
    uint32_t* vin = vals_in;
    uint32_t* vout = vals_out;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        const int dmask = end_bit - shift >= 8 ? 255 : (1 << (end_bit - shift)) - 1;      // the last digit stops at end_bit
        hipLaunchKernelGGL(sort_count_kernel, dim3(ntiles), dim3(kSortThreads), 0, s, (const uint64_t*)kin, sg, shift, dmask, hist);
        const int r = scan_rec(hist, hist, nh, false, scan_tmp, s);
        if (r != PNP_OK) return r;
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(ntiles), dim3(kSortThreads), 0, s, (const uint64_t*)kin, (const uint32_t*)vin, sg, shift,
                           dmask, (const int*)hist, kout, vout);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    if (hipGetLastError() != hipSuccess) return PNP_ERR_HIP;
    if (kin != keys_out) {                          // even number of passes: the result sits in the input arrays
        if (hipMemcpyAsync(keys_out, kin, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return PNP_ERR_HIP;
        if (hipMemcpyAsync(vals_out, vin, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess) return PNP_ERR_HIP;
    }
    return PNP_OK;
}

}  // namespace pnp
