// pi_tile_kernels.hip — tile-staged evaluation sweeps (second half of the kernel template;
// appended after pi_sweep_kernels.hip in the same translation unit).
//
// Why: on MI355X the flat one-state-per-lane sweep is bound by the REAL traffic of its
// scattered V gather (4.85 GB past L2 per 80^4 sweep against 0.53 GB
// compulsory).  The successor cells of neighbouring states are neighbours too, but not along
// the memory-fastest dimension, so a wave's 64 lanes hit ~20 different rows.  Here a
// workgroup owns a compact D-dimensional TILE of states (PI_TILE_INIT, <= 256 states); the
// successor cells of a tile fall in a small D-dimensional BOX of V (PI_BOX_INIT).  The box is
// copied into LDS with wide coalesced loads — one pass over each cache line instead of one
// access per lane per corner — and the 2^D-corner interpolation reads LDS.  A state whose cell
// is not inside the box (the box extents are compile-time constants measured by
// pi_reach_kernel; dynamics are arbitrary user code) falls back to the global gather, per
// thread, so results never depend on the box.  Arithmetic is the flat kernels' (same
// pi_locate / pi_corner_weights, same fmaf chain order): bit-identical V.
//
// Required macros (generated): PI_TILE_INIT {T0..}, PI_BOX_INIT {E0..} with E[d] <= g[d] and
// E[D-1] % 4 == 0.  PI_TILED must be defined to compile this half.
#ifdef PI_TILED

struct PiTiling {
    int t[PI_D];          // tile extents
    int nt[PI_D];         // tiles per dimension
    int e[PI_D];          // staged box extents
    int lstride[PI_D];    // LDS strides of the box (row-major, last dimension fastest)
    int n_in_tile;        // states per tile (<= PI_BLOCK)
    int box_vol;          // floats in the box
    int tiles_inner;      // tiles per unit of the dim-0 tile coordinate
};
__host__ __device__ constexpr PiTiling pi_make_tiling() {
    PiTiling r = {};
    const int t[PI_D] = PI_TILE_INIT;
    const int e[PI_D] = PI_BOX_INIT;
    r.n_in_tile = 1;
    r.box_vol = 1;
    r.tiles_inner = 1;
    for (int d = 0; d < PI_D; ++d) {
        r.t[d] = t[d];
        r.e[d] = e[d];
        r.nt[d] = (PI_GRID.g[d] + t[d] - 1) / t[d];
        r.n_in_tile *= t[d];
        r.box_vol *= e[d];
        if (d > 0) r.tiles_inner *= r.nt[d];
    }
    r.lstride[PI_D - 1] = 1;
    for (int d = PI_D - 2; d >= 0; --d) r.lstride[d] = r.lstride[d + 1] * r.e[d + 1];
    return r;
}
constexpr PiTiling PI_TL = pi_make_tiling();
static_assert(PI_TL.n_in_tile <= PI_BLOCK, "tile larger than the workgroup");
static_assert(PI_TL.e[PI_D - 1] % 4 == 0, "box rows must be whole float4s");
static_assert(PI_TL.box_vol * 4 <= 64 * 1024, "staged box exceeds 64 KiB of LDS");

__device__ __forceinline__ constexpr int pi_box_corner_offset(int c) {
    int m = pi_corner_mask(c), off = 0;
    for (int d = 0; d < PI_D; ++d) off += ((m >> d) & 1) * PI_TL.lstride[d];
    return off;
}

// Cell coordinates (per dimension) instead of the flat base of pi_locate; same arithmetic.
__device__ __forceinline__ void pi_locate_cell(const float (&ns)[PI_D], const float* __restrict__ tab,
                                               int (&cell)[PI_D], float (&fr)[PI_D]) {
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
        const float lo = tab[PI_TAB_LO + d];
        const float hi = tab[PI_TAB_HI + d];
        const float top = (float)(PI_GRID.g[d] - 1);
        float n = (ns[d] - lo) / (hi - lo) * top;
        n = fmaxf(0.0f, fminf(n, top));
        int i = min((int)n, PI_GRID.g[d] - 2);
        fr[d] = n - (float)i;
        cell[d] = i;
    }
}

// Interpolation from the LDS copy of the box: same weights, same chain order as pi_interpolate.
__device__ __forceinline__ float pi_interpolate_lds(const float* box, int lbase, const float (&fr)[PI_D]) {
    float w[PI_C];
    pi_corner_weights(fr, w);
    float v[PI_C];
#pragma unroll
    for (int c = 0; c < PI_C; ++c) v[c] = box[lbase + pi_box_corner_offset(c)];
    float e = 0.0f;
#pragma unroll
    for (int c = 0; c < PI_C; ++c) e = fmaf(w[pi_corner_mask(c)], v[c], e);
    return e;
}

// Block-wide min and max of D integers (invalid lanes pass INT_MAX / INT_MIN).
__device__ __forceinline__ void pi_block_minmax(int (&mn)[PI_D], int (&mx)[PI_D], int* lds_mm) {
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[d] = min(mn[d], __shfl_xor(mn[d], o, 64));
            mx[d] = max(mx[d], __shfl_xor(mx[d], o, 64));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < PI_D; ++d) {
            lds_mm[wave * 2 * PI_D + d] = mn[d];
            lds_mm[wave * 2 * PI_D + PI_D + d] = mx[d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
        int a = lds_mm[d], b = lds_mm[PI_D + d];
#pragma unroll
        for (int w = 1; w < PI_BLOCK / 64; ++w) {
            a = min(a, lds_mm[w * 2 * PI_D + d]);
            b = max(b, lds_mm[w * 2 * PI_D + PI_D + d]);
        }
        mn[d] = a;
        mx[d] = b;
    }
}

// Box origin for a tile whose cells span [mn, mx] per dimension: the cells' range centred in
// the box when it does not fit, clamped so the box lies inside the grid.
__device__ __forceinline__ void pi_box_origin(const int (&mn)[PI_D], const int (&mx)[PI_D],
                                              int (&lo)[PI_D]) {
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
        const int need = mx[d] - mn[d] + 2;                 // cells + the upper corner
        int o = mn[d] - max(0, (PI_TL.e[d] - need) / 2);
        if (need > PI_TL.e[d]) o = mn[d] + (need - PI_TL.e[d]) / 2;
        lo[d] = max(0, min(o, PI_GRID.g[d] - PI_TL.e[d]));
    }
}

typedef float pi_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float pi_f4 __attribute__((ext_vector_type(4)));

// Copy V[lo .. lo+E) into LDS, row-major, as float4s (rows are E[D-1] contiguous floats of V).
__device__ __forceinline__ void pi_fill_box(float* box, const float* __restrict__ V,
                                            const int (&lo)[PI_D]) {
    long long origin = 0;
#pragma unroll
    for (int d = 0; d < PI_D; ++d) origin += (long long)lo[d] * PI_GRID.stride[d];
    const float* __restrict__ Vo = V + origin;
    constexpr int n4 = PI_TL.box_vol / 4;
    constexpr int row4 = PI_TL.e[PI_D - 1] / 4;
#pragma unroll 4
    for (int e4 = threadIdx.x; e4 < n4; e4 += PI_BLOCK) {
        int r = e4 / row4;
        int goff = (e4 - r * row4) * 4;
#pragma unroll
        for (int d = PI_D - 2; d >= 0; --d) {
            const int q = r / PI_TL.e[d];
            goff += (r - q * PI_TL.e[d]) * PI_GRID.stride[d];
            r = q;
        }
        const pi_f4u v = *reinterpret_cast<const pi_f4u*>(Vo + goff);
        *reinterpret_cast<pi_f4*>(box + e4 * 4) = v;
    }
}

// Tile id (last dimension fastest; dim-0 tile coordinate offset by tile0_lo) and thread id ->
// grid coordinates and flat index.  Returns false for lanes outside the tile / the grid.
__device__ __forceinline__ bool pi_tile_state(long long tile, int tile0_lo, int tid,
                                              int (&coord)[PI_D], long long& s) {
    int tc[PI_D];
    long long r = tile;
#pragma unroll
    for (int d = PI_D - 1; d > 0; --d) {
        const long long q = r / PI_TL.nt[d];
        tc[d] = (int)(r - q * PI_TL.nt[d]);
        r = q;
    }
    tc[0] = (int)r + tile0_lo;
    int u = tid;
    bool ok = tid < PI_TL.n_in_tile;
    s = 0;
#pragma unroll
    for (int d = PI_D - 1; d >= 0; --d) {
        const int q = u / PI_TL.t[d];
        coord[d] = tc[d] * PI_TL.t[d] + (u - q * PI_TL.t[d]);
        u = q;
        ok = ok && coord[d] < PI_GRID.g[d];
        s += (long long)coord[d] * PI_GRID.stride[d];
    }
    return ok;
}

// Transition-record layout of the tiled kernels (struct of arrays over k = tile_local*256 + tid;
// `cap` = n_tiles * 256 entries per array): reward | code | frac_0 .. frac_{D-1}, then
// tile_lo[n_tiles][D] ints.  code >= 0: LDS offset of the cell inside the tile's box;
// -1: transition terminates; -2: terminal node; <= -3: cell outside the box, flat index = -3 - code.
#define PI_TREC_DONE (-1)
#define PI_TREC_TERMINAL (-2)

// MODE 0: recompute (plain sweep); 1: recompute + write records; 2: replay records.
template <int MODE>
__device__ __forceinline__ void pi_tile_eval_body(const float* __restrict__ V, float* __restrict__ Vn,
                                                  const int* __restrict__ policy,
                                                  const unsigned char* __restrict__ term,
                                                  const float* __restrict__ tab, long long s_begin,
                                                  long long s_end, float gamma,
                                                  unsigned int* __restrict__ delta_bits,
                                                  float* __restrict__ rec, long long cap,
                                                  int tile0_lo, long long n_tiles) {
    __shared__ __attribute__((aligned(16))) float box[PI_TL.box_vol];
    __shared__ float lds_tab[MODE == 2 ? 1 : PI_GRID.tab_len];
    __shared__ int lds_mm[(PI_BLOCK / 64) * 2 * PI_D];
    __shared__ float lds_red[PI_BLOCK / 64];
    if (MODE != 2) {
        for (int i = threadIdx.x; i < PI_GRID.tab_len; i += PI_BLOCK) lds_tab[i] = tab[i];
        __syncthreads();
    }
    int* __restrict__ tile_lo = reinterpret_cast<int*>(rec + (2 + PI_D) * cap);
    const PiChunks ck = pi_chunks_of(n_tiles);
    float dmax = 0.0f;
    for (long long cl = ck.j; cl < ck.span; cl += ck.step) {
        const long long tile = ck.x * ck.span + cl;
        if (tile >= ck.n_chunks) break;
        int coord[PI_D];
        long long s;
        bool ok = pi_tile_state(tile, tile0_lo, threadIdx.x, coord, s);
        ok = ok && s >= s_begin && s < s_end;
        const long long k = tile * PI_BLOCK + threadIdx.x;

        float reward = 0.0f, fr[PI_D], v_old = 0.0f;
        int cell[PI_D], lo[PI_D];
        int code = PI_TREC_TERMINAL;       // what this lane has: a cell (code 0), done, terminal
        bool has_cell = false;
#pragma unroll
        for (int d = 0; d < PI_D; ++d) { fr[d] = 0.0f; cell[d] = 0; }

        if (MODE == 2) {
            // ---- replay: records + the tile's box origin ----
#pragma unroll
            for (int d = 0; d < PI_D; ++d) lo[d] = tile_lo[tile * PI_D + d];
            if (ok) {
                reward = __builtin_nontemporal_load(rec + k);
                code = __builtin_nontemporal_load(reinterpret_cast<const int*>(rec) + cap + k);
#pragma unroll
                for (int d = 0; d < PI_D; ++d) fr[d] = __builtin_nontemporal_load(rec + (2 + d) * cap + k);
                v_old = V[s];
            }
        } else {
            // ---- phase A: the transition of this lane's state ----
            if (ok) {
                v_old = V[s];
                if (!term[s]) {
                    float x[PI_D], ns[PI_D];
#pragma unroll
                    for (int d = 0; d < PI_D; ++d) x[d] = lds_tab[PI_GRID.bins_off[d] + coord[d]];
                    const float a = lds_tab[PI_TAB_ACT + policy[s]];
                    bool done;
                    pi_dynamics(x, a, ns, &reward, &done);
                    code = PI_TREC_DONE;
                    if (!done) {
                        pi_locate_cell(ns, tab, cell, fr);
                        has_cell = true;
                        code = 0;
                    }
                }
            }
            // ---- phase B: where the tile's cells are ----
            int mn[PI_D], mx[PI_D];
#pragma unroll
            for (int d = 0; d < PI_D; ++d) {
                mn[d] = has_cell ? cell[d] : 0x7fffffff;
                mx[d] = has_cell ? cell[d] : (int)0x80000000;
            }
            pi_block_minmax(mn, mx, lds_mm);
            if (mn[0] == 0x7fffffff) {
#pragma unroll
                for (int d = 0; d < PI_D; ++d) { mn[d] = 0; mx[d] = 0; }
            }
            pi_box_origin(mn, mx, lo);
        }

        // ---- phase C: stage the box ----
        __syncthreads();                    // previous tile's readers are done with `box`
        pi_fill_box(box, V, lo);
        __syncthreads();

        // ---- phase D: interpolate ----
        float nv = v_old;
        if (MODE == 2) {
            if (ok && code != PI_TREC_TERMINAL) {
                float e = 0.0f;
                if (code >= 0) e = pi_interpolate_lds(box, code, fr);
                else if (code <= -3) e = pi_interpolate(V, -3 - code, fr);
                nv = reward + gamma * e;
            }
        } else {
            if (ok && code != PI_TREC_TERMINAL) {
                float e = 0.0f;
                if (has_cell) {
                    bool inside = true;
                    int lbase = 0, gbase = 0;
#pragma unroll
                    for (int d = 0; d < PI_D; ++d) {
                        const int rel = cell[d] - lo[d];
                        inside = inside && rel >= 0 && rel <= PI_TL.e[d] - 2;
                        lbase += rel * PI_TL.lstride[d];
                        gbase += cell[d] * PI_GRID.stride[d];
                    }
                    if (inside) { e = pi_interpolate_lds(box, lbase, fr); code = lbase; }
                    else { e = pi_interpolate(V, gbase, fr); code = -3 - gbase; }
                }
                nv = reward + gamma * e;
            }
            if (MODE == 1) {
                if (threadIdx.x < PI_D) tile_lo[tile * PI_D + threadIdx.x] = lo[threadIdx.x];
                rec[k] = reward;
                reinterpret_cast<int*>(rec)[cap + k] = code;
#pragma unroll
                for (int d = 0; d < PI_D; ++d) rec[(2 + d) * cap + k] = fr[d];
            }
        }
        if (ok) {
            Vn[s] = nv;
            const float dlt = fabsf(nv - v_old);
            dmax = dlt > dmax ? dlt : dmax;
        }
    }
    if (delta_bits != nullptr) pi_block_max_to(dmax, lds_red, delta_bits);
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_tile_eval_kernel(const float* __restrict__ V, float* __restrict__ Vn,
                    const int* __restrict__ policy, const unsigned char* __restrict__ term,
                    const float* __restrict__ tab, long long s_begin, long long s_end, float gamma,
                    unsigned int* __restrict__ delta_bits, int tile0_lo, long long n_tiles) {
    pi_tile_eval_body<0>(V, Vn, policy, term, tab, s_begin, s_end, gamma, delta_bits, nullptr, 0,
                         tile0_lo, n_tiles);
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_tile_build_kernel(const float* __restrict__ V, float* __restrict__ Vn,
                     const int* __restrict__ policy, const unsigned char* __restrict__ term,
                     const float* __restrict__ tab, long long s_begin, long long s_end, float gamma,
                     unsigned int* __restrict__ delta_bits, float* __restrict__ rec, long long cap,
                     int tile0_lo, long long n_tiles) {
    pi_tile_eval_body<1>(V, Vn, policy, term, tab, s_begin, s_end, gamma, delta_bits, rec, cap,
                         tile0_lo, n_tiles);
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_tile_replay_kernel(const float* __restrict__ V, float* __restrict__ Vn,
                      float* __restrict__ rec, long long cap, long long s_begin, long long s_end,
                      float gamma, unsigned int* __restrict__ delta_bits, int tile0_lo,
                      long long n_tiles) {
    pi_tile_eval_body<2>(V, Vn, nullptr, nullptr, nullptr, s_begin, s_end, gamma, delta_bits, rec, cap,
                         tile0_lo, n_tiles);
}

#endif  // PI_TILED

// ---- reach probe: how large a box do the successor cells of a tile need? ----------------
// For every sampled tile (PI_TILE_INIT) and EVERY action, the extent per dimension of the cells
// hit by the tile's states (+1 for the upper corner); reach[d] = max over tiles (atomicMax) and
// hist[d][x] = number of tiles with extent x (x < 64), so the host can pick a percentile.
#ifdef PI_TILE_INIT
#ifndef PI_TILED
struct PiTilingLite { int t[PI_D]; int nt[PI_D]; int n_in_tile; };
__host__ __device__ constexpr PiTilingLite pi_make_tiling_lite() {
    PiTilingLite r = {};
    const int t[PI_D] = PI_TILE_INIT;
    r.n_in_tile = 1;
    for (int d = 0; d < PI_D; ++d) {
        r.t[d] = t[d];
        r.nt[d] = (PI_GRID.g[d] + t[d] - 1) / t[d];
        r.n_in_tile *= t[d];
    }
    return r;
}
constexpr PiTilingLite PI_TLL = pi_make_tiling_lite();
#define PI_TLX PI_TLL
#else
#define PI_TLX PI_TL
#endif

extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_reach_kernel(const unsigned char* __restrict__ term, const float* __restrict__ tab,
                long long n_tiles_total, long long tile_step, int* __restrict__ reach,
                int* __restrict__ hist) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ int lds_mm[(PI_BLOCK / 64) * 2 * PI_D];
    for (int i = threadIdx.x; i < PI_GRID.tab_len; i += PI_BLOCK) lds_tab[i] = tab[i];
    __syncthreads();
    for (long long tile = (long long)blockIdx.x * tile_step; tile < n_tiles_total;
         tile += (long long)gridDim.x * tile_step) {
        // tile id -> coordinates (same order as pi_tile_state with tile0_lo = 0)
        int tc[PI_D], coord[PI_D];
        long long r = tile;
#pragma unroll
        for (int d = PI_D - 1; d > 0; --d) {
            const long long q = r / PI_TLX.nt[d];
            tc[d] = (int)(r - q * PI_TLX.nt[d]);
            r = q;
        }
        tc[0] = (int)r;
        int u = threadIdx.x;
        bool ok = threadIdx.x < PI_TLX.n_in_tile;
        long long s = 0;
#pragma unroll
        for (int d = PI_D - 1; d >= 0; --d) {
            const int q = u / PI_TLX.t[d];
            coord[d] = tc[d] * PI_TLX.t[d] + (u - q * PI_TLX.t[d]);
            u = q;
            ok = ok && coord[d] < PI_GRID.g[d];
            s += (long long)coord[d] * PI_GRID.stride[d];
        }
        int mn[PI_D], mx[PI_D];
#pragma unroll
        for (int d = 0; d < PI_D; ++d) { mn[d] = 0x7fffffff; mx[d] = (int)0x80000000; }
        if (ok && (term == nullptr || !term[s])) {
            float x[PI_D];
#pragma unroll
            for (int d = 0; d < PI_D; ++d) x[d] = lds_tab[PI_GRID.bins_off[d] + coord[d]];
            for (int a = 0; a < PI_NA; ++a) {
                float ns[PI_D], reward;
                bool done;
                pi_dynamics(x, tab[PI_TAB_ACT + a], ns, &reward, &done);
                if (!done) {
                    int base;
                    float fr[PI_D];
                    pi_locate(ns, tab, base, fr);
                    int rem = base;
#pragma unroll
                    for (int d = 0; d < PI_D; ++d) {
                        const int c = rem / PI_GRID.stride[d];
                        rem -= c * PI_GRID.stride[d];
                        mn[d] = min(mn[d], c);
                        mx[d] = max(mx[d], c);
                    }
                }
            }
        }
#pragma unroll
        for (int d = 0; d < PI_D; ++d) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn[d] = min(mn[d], __shfl_xor(mn[d], o, 64));
                mx[d] = max(mx[d], __shfl_xor(mx[d], o, 64));
            }
        }
        const int wave = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int d = 0; d < PI_D; ++d) {
                lds_mm[wave * 2 * PI_D + d] = mn[d];
                lds_mm[wave * 2 * PI_D + PI_D + d] = mx[d];
            }
        }
        __syncthreads();
        if (threadIdx.x < PI_D) {
            const int d = threadIdx.x;
            int a = lds_mm[d], b = lds_mm[PI_D + d];
            for (int w = 1; w < PI_BLOCK / 64; ++w) {
                a = min(a, lds_mm[w * 2 * PI_D + d]);
                b = max(b, lds_mm[w * 2 * PI_D + PI_D + d]);
            }
            if (a != 0x7fffffff) {
                const int ext = b - a + 2;
                atomicMax(&reach[d], ext);
                atomicAdd(&hist[d * 64 + min(ext, 63)], 1);
            }
        }
    }
}
#endif  // PI_TILE_INIT
